// Stand-alone harness of the bf16x6 attention kernel (csrc/kernels_rof.h: attention6_kernel) against attention2_kernel and a
// float64 softmax attention on sampled (sequence, head, query) items; time per launch on the BS-Roformer shapes.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/proto_attn6 tools/proto_attn6.hip && tools/proto_attn6
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#include "../python-audio-separator_amd/csrc/kernels_net.h"
#include "../python-audio-separator_amd/csrc/kernels_gemm3.h"
#include "../python-audio-separator_amd/csrc/kernels_rof.h"

using namespace asx;

#define CK(x)                                                                  \
  do {                                                                         \
    hipError_t e_ = (x);                                                       \
    if (e_ != hipSuccess) {                                                    \
      fprintf(stderr, "HIP %s at line %d\n", hipGetErrorString(e_), __LINE__); \
      exit(2);                                                                 \
    }                                                                          \
  } while (0)

struct Shape {
  const char *name;
  int nseq, len, heads;
  float amp;   // logit scale: larger = peakier softmax
};

static void run(const Shape &sh, int reps) {
  const int H = sh.heads, L = sh.len, inner = H * 64;
  const int64_t M = (int64_t)sh.nseq * L;
  std::mt19937 rng(11 + L);
  std::normal_distribution<float> nd(0.f, 1.f);
  // host data for the first two sequences; the rest repeat them
  const int hs = std::min(sh.nseq, 2);
  std::vector<float> hq((size_t)hs * L * 3 * inner), hg((size_t)hs * L * H);
  for (auto &v : hq) v = nd(rng) * sh.amp;
  for (auto &v : hg) v = nd(rng);
  float *dq, *dg, *o2, *o6, *o3;
  CK(hipMalloc(&dq, (size_t)M * 3 * inner * 4));
  CK(hipMalloc(&dg, (size_t)M * H * 4));
  CK(hipMalloc(&o2, (size_t)M * inner * 4));
  CK(hipMalloc(&o6, (size_t)M * inner * 4));
  CK(hipMalloc(&o3, (size_t)M * inner * 4));
  CK(hipMemset(o3, 0xff, (size_t)M * inner * 4));
  for (int s = 0; s < sh.nseq; s += hs) {
    const int n = std::min(hs, sh.nseq - s);
    CK(hipMemcpy(dq + (size_t)s * L * 3 * inner, hq.data(), (size_t)n * L * 3 * inner * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dg + (size_t)s * L * H, hg.data(), (size_t)n * L * H * 4, hipMemcpyHostToDevice));
  }
  CK(hipMemset(o2, 0xff, (size_t)M * inner * 4));
  CK(hipMemset(o6, 0xff, (size_t)M * inner * 4));
  AttnArgs a{};
  a.qkv = dq;
  a.gate = dg;
  a.heads = H;
  a.gate_ld = H;
  a.len = L;
  a.row_stride = 1;
  a.inner_cnt = 1;
  a.outer_stride = L;
  a.inner_stride = 0;
  a.scale = 0.125f;
  a.exact = 0;
  a.nqt = (L + 63) / 64;
  const dim3 grid((unsigned)((int64_t)a.nqt * H * sh.nseq));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  auto time_it = [&](auto &&fn) {
    fn();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < reps; ++i) fn();
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    return (double)ms / reps;
  };
  a.out = o2;
  const double t2 = time_it([&]() { hipLaunchKernelGGL(attention2_kernel<1>, grid, dim3(256), 0, 0, a); });
  a.out = o6;
  const double t61 = time_it([&]() { hipLaunchKernelGGL(attention6_kernel<1>, grid, dim3(256), 0, 0, a); });
  AttnArgs b = a;
  b.nqt = (L + 127) / 128;
  const dim3 grid2((unsigned)((int64_t)b.nqt * H * sh.nseq));
  CK(hipMemset(o6, 0xff, (size_t)M * inner * 4));
  const double t6 = time_it([&]() { hipLaunchKernelGGL(attention6_kernel<2>, grid2, dim3(256), 0, 0, b); });
  // fp16 x 3 arithmetic (template parameter H)
  b.out = o3;
  const double t3 = time_it([&]() { hipLaunchKernelGGL((attention6_kernel<2, true>), grid2, dim3(256), 0, 0, b); });
  AttnArgs c1 = a;
  c1.out = o3;
  const double t31 = time_it([&]() { hipLaunchKernelGGL((attention6_kernel<1, true>), grid, dim3(256), 0, 0, c1); });
  CK(hipMemset(o3, 0xff, (size_t)M * inner * 4));
  hipLaunchKernelGGL((attention6_kernel<2, true>), grid2, dim3(256), 0, 0, b);
  CK(hipDeviceSynchronize());
  CK(hipGetLastError());
  const double flops = 4.0 * sh.nseq * H * (double)L * L * 64;

  // float64 reference on the LAST sequence (same data as sequence (nseq - 1) % hs), a few heads / queries
  const int sl = sh.nseq - 1, hsq = sl % hs;
  std::vector<float> y2((size_t)L * inner), y6((size_t)L * inner), y3((size_t)L * inner);
  CK(hipMemcpy(y3.data(), o3 + (size_t)sl * L * inner, y3.size() * 4, hipMemcpyDeviceToHost));
  CK(hipMemcpy(y2.data(), o2 + (size_t)sl * L * inner, y2.size() * 4, hipMemcpyDeviceToHost));
  CK(hipMemcpy(y6.data(), o6 + (size_t)sl * L * inner, y6.size() * 4, hipMemcpyDeviceToHost));
  double e2 = 0, e6 = 0, e3 = 0, nrm = 0, mx2 = 0, mx6 = 0, mx3 = 0;
  long bad = 0;
  const int qstep = std::max(1, L / 37);
  for (int h = 0; h < H; h += std::max(1, H / 3))
    for (int q = 0; q < L; q += qstep) {
      const float *qr = &hq[((size_t)hsq * L + q) * 3 * inner + h * 64];
      std::vector<double> sc(L);
      double m = -1e300;
      for (int k = 0; k < L; ++k) {
        const float *kr = &hq[((size_t)hsq * L + k) * 3 * inner + inner + h * 64];
        double d = 0;
        for (int i = 0; i < 64; ++i) d += (double)qr[i] * kr[i];
        sc[k] = d * 0.125;
        m = std::max(m, sc[k]);
      }
      double den = 0;
      for (int k = 0; k < L; ++k) den += (sc[k] = std::exp(sc[k] - m));
      const double g = 1.0 / (1.0 + std::exp(-(double)hg[((size_t)hsq * L + q) * H + h]));
      for (int i = 0; i < 64; ++i) {
        double o = 0;
        for (int k = 0; k < L; ++k) o += sc[k] * hq[((size_t)hsq * L + k) * 3 * inner + 2 * inner + h * 64 + i];
        o = o / den * g;
        const double a2 = y2[(size_t)q * inner + h * 64 + i], a6 = y6[(size_t)q * inner + h * 64 + i];
        const double a3 = y3[(size_t)q * inner + h * 64 + i];
        if (!std::isfinite(a6) || !std::isfinite(a3)) ++bad;
        e3 += (a3 - o) * (a3 - o);
        mx3 = std::max(mx3, std::fabs(a3 - o));
        e2 += (a2 - o) * (a2 - o);
        e6 += (a6 - o) * (a6 - o);
        nrm += o * o;
        mx2 = std::max(mx2, std::fabs(a2 - o));
        mx6 = std::max(mx6, std::fabs(a6 - o));
      }
    }
  printf("%-12s nseq=%-6d len=%-5d heads=%d  attn2 %8.3f ms %6.1f TF | attn6<1> %8.3f ms | attn6<2> %8.3f ms %6.1f TF-eq (x%.2f) | relrms vs f64: attn2 %.2e attn6 %.2e maxabs %.2e / %.2e nonfinite %ld\n",
         sh.name, sh.nseq, L, H, t2, flops / t2 * 1e-9, t61, t6, flops / t6 * 1e-9, t2 / t6, std::sqrt(e2 / nrm), std::sqrt(e6 / nrm), mx2, mx6, bad);
  printf("             fp16 x 3: attn6h<1> %8.3f ms | attn6h<2> %8.3f ms %6.1f TF-eq (x%.2f vs bf16 x 6) | relrms vs f64 %.2e maxabs %.2e\n", t31, t3,
         flops / t3 * 1e-9, t6 / t3, std::sqrt(e3 / nrm), mx3);
  fflush(stdout);
  CK(hipFree(dq));
  CK(hipFree(dg));
  CK(hipFree(o2));
  CK(hipFree(o6));
  CK(hipFree(o3));
}

int main() {
  std::vector<Shape> shapes = {
      {"tiny", 3, 62, 2, 1.0f},   {"ragged", 2, 100, 3, 1.0f},    {"peaky", 2, 257, 2, 2.5f},
      {"rof time", 496, 801, 8, 1.0f}, {"rof band", 6408, 62, 8, 1.0f}, {"long", 16, 4096, 8, 1.0f},
      {"small amp", 2, 300, 2, 1e-3f}, {"large amp", 2, 300, 2, 6.0f},
  };
  for (auto &s : shapes) run(s, 3);
  return 0;
}
