// CPU harness for the per-thread stage bodies of csrc/kernels_fft3.h (ASX_HOST_TEST): runs the three Stockham passes, the
// forward split and the inverse merge thread by thread, exactly as a 256-thread workgroup would between barriers, and
// writes the results for tests/test_fft3_host.py to compare with numpy's FFT.
//   fft3_host <in.bin> <out.bin>
//   in : 6144 floats (windowed frame), then 3072 x 2 floats (a half spectrum X[k], k < 3072)
//   out: 3072 x 2 floats forward X[k]; then 6144 floats = inverse frame / 3072 (un-windowed)
#define ASX_HOST_TEST 1
#include "../../python-audio-separator_amd/csrc/kernels_fft3.h"
#include <cstdio>
#include <vector>
using namespace asx::f3;

int main(int argc, char **argv) {
  if (argc < 3) return 2;
  std::vector<float> x(NFFT), X(2 * NH);
  FILE *f = fopen(argv[1], "rb");
  if (!f || fread(x.data(), 4, NFFT, f) != (size_t)NFFT || fread(X.data(), 4, 2 * NH, f) != (size_t)(2 * NH)) return 3;
  fclose(f);
  const double PI = 3.14159265358979323846;
  std::vector<float2> tw(NFFT), twB(16 * 12), twC(16 * NB);
  for (int i = 0; i < NFFT; ++i) tw[i] = make_float2((float)cos(2 * PI * i / NFFT), (float)-sin(2 * PI * i / NFFT));
  for (int r = 0; r < 16; ++r)
    for (int k = 0; k < 12; ++k) twB[r * 12 + k] = make_float2((float)cos(2 * PI * k * r / 192.0), (float)-sin(2 * PI * k * r / 192.0));
  for (int r = 0; r < 16; ++r)
    for (int j = 0; j < NB; ++j) twC[r * NB + j] = make_float2((float)cos(2 * PI * j * r / 3072.0), (float)-sin(2 * PI * j * r / 3072.0));
  std::vector<float2> buf(LDS_X);
  std::vector<float2> regs(NB * 16);   // the registers a thread keeps across a barrier
  std::vector<float> out(2 * NH + NFFT);
  // ---- forward ----
  for (int j = 0; j < 256; ++j) {
    float2 v[12];
    for (int r = 0; r < 12; ++r) v[r] = make_float2(x[2 * (j + 256 * r)], x[2 * (j + 256 * r) + 1]);
    pass_a<-1>(j, v, buf.data());
  }
  for (int j = 0; j < NB; ++j) pass_b_load(j, buf.data(), &regs[j * 16]);                       // barrier
  for (int j = 0; j < NB; ++j) pass_b_store<-1>(j, &regs[j * 16], buf.data(), twB.data());       // barrier
  for (int j = 0; j < NB; ++j) pass_c_load(j, buf.data(), &regs[j * 16]);                       // barrier
  std::vector<float2> Z(NH);
  for (int j = 0; j < NB; ++j) {
    pass_c_compute<-1>(j, &regs[j * 16], twC.data());
    for (int r = 0; r < 16; ++r) Z[j + NB * r] = regs[j * 16 + r];
  }
  for (int k = 0; k < NH; ++k) {
    const float2 v = split_bin(k, Z.data(), tw[k]);
    out[2 * k] = v.x;
    out[2 * k + 1] = v.y;
  }
  // ---- inverse ----
  auto bin = [&](int k) { return k >= NH ? make_float2(0.f, 0.f) : make_float2(X[2 * k], k == 0 ? 0.f : X[2 * k + 1]); };
  for (int j = 0; j < 256; ++j) {
    float2 v[12];
    for (int r = 0; r < 12; ++r) {
      const int k = j + 256 * r;
      v[r] = merge_bin(bin(k), bin(NH - k), tw[k]);
    }
    pass_a<+1>(j, v, buf.data());
  }
  for (int j = 0; j < NB; ++j) pass_b_load(j, buf.data(), &regs[j * 16]);
  for (int j = 0; j < NB; ++j) pass_b_store<+1>(j, &regs[j * 16], buf.data(), twB.data());
  for (int j = 0; j < NB; ++j) {
    float2 *c = &regs[j * 16];
    pass_c_load(j, buf.data(), c);
    pass_c_compute<+1>(j, c, twC.data());
    for (int r = 0; r < 16; ++r) {
      const int m = j + NB * r;
      out[2 * NH + 2 * m] = c[r].x / (float)NH;
      out[2 * NH + 2 * m + 1] = c[r].y / (float)NH;
    }
  }
  f = fopen(argv[2], "wb");
  fwrite(out.data(), 4, out.size(), f);
  fclose(f);
  return 0;
}
