// Stand-alone harness of the bf16x6 row GEMM (csrc/kernels_gemm3.h) against the fp32-MFMA kernel it replaces (kernels_gemm2.h)
// and a float64 host GEMM on sampled rows: error of both kernels, time per launch on the TDF / Roformer / Demucs shapes.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/proto_gemm3 tools/proto_gemm3.hip && tools/proto_gemm3 [abl] [first] [last] [tile_map]
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

#include "../python-audio-separator_amd/csrc/kernels_net.h"
#include "../python-audio-separator_amd/csrc/kernels_gemm2.h"
#include "../python-audio-separator_amd/csrc/kernels_gemm3.h"

using namespace asx;

#define CK(x)                                                                  \
  do {                                                                         \
    hipError_t e_ = (x);                                                       \
    if (e_ != hipSuccess) {                                                    \
      fprintf(stderr, "HIP %s at line %d\n", hipGetErrorString(e_), __LINE__); \
      exit(2);                                                                 \
    }                                                                          \
  } while (0)

template <int NREP, int MREP, int ABL>
static void launch3(const TdfDmaArgs &a, const u32x4 *w3, hipStream_t s) {
  constexpr int BM = 16 * MREP, BN = 64 * NREP, LDS = 2 * 3 * BM * 64;
  const int64_t nbm = (a.M + BM - 1) / BM;
  const int nbn = (a.N + BN - 1) / BN;
  hipLaunchKernelGGL((tdf3_kernel<NREP, MREP, ABL>), dim3((unsigned)(nbm * nbn)), dim3(256), LDS, s, a, w3, RowGather{});
}
template <int NREP, int MREP>
static void launch2(const TdfDmaArgs &a, hipStream_t s) {
  constexpr int BM = 16 * MREP, BN = 64 * NREP, LDS = 2 * (BM + BN) * 32 * 4;
  static bool done = false;
  if (!done) {
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(&tdf2_kernel<NREP, MREP, 0, 32>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
    done = true;
  }
  const int64_t nbm = (a.M + BM - 1) / BM;
  const int nbn = (a.N + BN - 1) / BN;
  hipLaunchKernelGGL((tdf2_kernel<NREP, MREP, 0, 32>), dim3((unsigned)(nbm * nbn)), dim3(256), LDS, s, a, 1, -1);
}

static int g_map = 0;   // TdfDmaArgs::tile_map of the tdf3 launches (argv[4])
struct Shape {
  const char *name;
  int64_t M;
  int N, K, C, T, relu, res, bias;
};

static double run_shape(const Shape &sh, int abl, int reps) {
  const int64_t M = sh.M;
  const int N = sh.N, K = sh.K;
  std::mt19937 rng(1234 + N + K);
  std::normal_distribution<float> nd(0.f, 1.f);
  // host data only for the sampled rows: x is generated on the host in full when small, else tiled from a 4096-row block
  const int64_t HB = std::min<int64_t>(M, 4096);
  std::vector<float> hx((size_t)HB * K), hw((size_t)N * K), hb(N), hsc(sh.C), hsh(sh.C), hr((size_t)HB * N);
  for (auto &v : hx) v = nd(rng) * 3.0f;
  for (auto &v : hw) v = nd(rng) / std::sqrt((float)K);
  for (auto &v : hb) v = nd(rng);
  for (auto &v : hsc) v = 0.5f + 0.5f * std::fabs(nd(rng));
  for (auto &v : hsh) v = 0.2f * nd(rng);
  for (auto &v : hr) v = nd(rng);
  float *dx, *dw, *db, *dsc, *dsh, *dr, *dy2, *dy3;
  u32x4 *dw3;
  CK(hipMalloc(&dx, (size_t)M * K * 4));
  CK(hipMalloc(&dw, (size_t)N * K * 4));
  CK(hipMalloc(&db, N * 4));
  CK(hipMalloc(&dsc, sh.C * 4));
  CK(hipMalloc(&dsh, sh.C * 4));
  CK(hipMalloc(&dr, (size_t)M * N * 4));
  CK(hipMalloc(&dy2, (size_t)M * N * 4));
  CK(hipMalloc(&dy3, (size_t)M * N * 4));
  const int ntiles = (N + 15) / 16, nk = ((K + 63) / 64) * 2;
  CK(hipMalloc(&dw3, (size_t)ntiles * nk * 3 * 1024));
  for (int64_t r0 = 0; r0 < M; r0 += HB) {
    const int64_t n = std::min<int64_t>(HB, M - r0);
    CK(hipMemcpy(dx + r0 * K, hx.data(), (size_t)n * K * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dr + r0 * N, hr.data(), (size_t)n * N * 4, hipMemcpyHostToDevice));
  }
  CK(hipMemcpy(dw, hw.data(), hw.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(db, hb.data(), N * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dsc, hsc.data(), sh.C * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dsh, hsh.data(), sh.C * 4, hipMemcpyHostToDevice));
  CK(hipMemset(dy2, 0xff, (size_t)M * N * 4));
  CK(hipMemset(dy3, 0xff, (size_t)M * N * 4));

  TdfDmaArgs a{};
  a.x = dx;
  a.w = dw;
  a.bias = sh.bias ? db : nullptr;
  a.scale = dsc;
  a.shift = dsh;
  a.res = sh.res ? dr : nullptr;
  a.y = dy2;
  a.M = M;
  a.N = N;
  a.K = K;
  a.C = sh.C;
  a.T = sh.T;
  a.relu = sh.relu;
  a.tile_map = g_map;

  const int64_t total = (int64_t)ntiles * nk * 64;
  hipLaunchKernelGGL(w3_split_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, 0, dw, dw3, N, K, total);
  CK(hipDeviceSynchronize());

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
  a.y = dy2;
  const double t2 = time_it([&]() { launch2<3, 8>(a, 0); });
  a.y = dy3;
  double t3 = 0;
  switch (abl) {
    case 0: t3 = time_it([&]() { launch3<3, 8, 0>(a, dw3, 0); }); break;
    case 1: t3 = time_it([&]() { launch3<3, 8, 1>(a, dw3, 0); }); break;
    case 2: t3 = time_it([&]() { launch3<3, 8, 2>(a, dw3, 0); }); break;
    case 4: t3 = time_it([&]() { launch3<3, 8, 4>(a, dw3, 0); }); break;
    case 8: t3 = time_it([&]() { launch3<3, 8, 8>(a, dw3, 0); }); break;
    case 9: t3 = time_it([&]() { launch3<3, 8, 9>(a, dw3, 0); }); break;
    case 13: t3 = time_it([&]() { launch3<3, 8, 13>(a, dw3, 0); }); break;
    default: fprintf(stderr, "abl?\n"); exit(2);
  }
  CK(hipGetLastError());
  const double flops = 2.0 * M * N * K;

  // float64 reference on sampled rows (first HB rows suffice: the data repeats), both kernels
  const int nsamp = 64;
  std::vector<float> y2((size_t)N), y3((size_t)N);
  double e2 = 0, e3 = 0, d23 = 0, nrm = 0, mx2 = 0, mx3 = 0;
  int nan3 = 0;
  for (int si = 0; si < nsamp; ++si) {
    int64_t row = (int64_t)((double)si / nsamp * M);
    if (si == nsamp - 1) row = M - 1;
    const int64_t hrow = row % HB;
    CK(hipMemcpy(y2.data(), dy2 + row * N, N * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(y3.data(), dy3 + row * N, N * 4, hipMemcpyDeviceToHost));
    const int c = (int)((row / sh.T) % sh.C);
    for (int n = 0; n < N; ++n) {
      double acc = 0;
      for (int k = 0; k < K; ++k) acc += (double)hx[hrow * K + k] * (double)hw[(size_t)n * K + k];
      double v = hsc[c] * (acc + (sh.bias ? hb[n] : 0.0)) + hsh[c];
      if (sh.relu == 1) v = v > 0 ? v : 0;
      if (sh.res) v += hr[hrow * N + n];
      if (!std::isfinite(y3[n])) ++nan3;
      e2 += (y2[n] - v) * (y2[n] - v);
      e3 += (y3[n] - v) * (y3[n] - v);
      d23 += ((double)y2[n] - y3[n]) * ((double)y2[n] - y3[n]);
      nrm += v * v;
      mx2 = std::max(mx2, std::fabs(y2[n] - v));
      mx3 = std::max(mx3, std::fabs(y3[n] - v));
    }
  }
  printf("%-22s M=%-8lld N=%-5d K=%-5d  tdf2 %8.3f ms %6.1f TF | tdf3 %8.3f ms %6.1f TF-eq (x%.2f) | relrms vs f64: tdf2 %.2e tdf3 %.2e  "
         "maxabs %.2e / %.2e  tdf2-vs-tdf3 %.2e  nonfinite %d\n",
         sh.name, (long long)M, N, K, t2, flops / t2 * 1e-9, t3, flops / t3 * 1e-9, t2 / t3, std::sqrt(e2 / nrm), std::sqrt(e3 / nrm), mx2, mx3,
         std::sqrt(d23 / nrm), nan3);
  fflush(stdout);
  CK(hipFree(dx));
  CK(hipFree(dw));
  CK(hipFree(db));
  CK(hipFree(dsc));
  CK(hipFree(dsh));
  CK(hipFree(dr));
  CK(hipFree(dy2));
  CK(hipFree(dy3));
  CK(hipFree(dw3));
  return t3;
}

int main(int argc, char **argv) {
  const int abl = argc > 1 ? atoi(argv[1]) : 0;
  const int first = argc > 2 ? atoi(argv[2]) : 0;
  const int last = argc > 3 ? atoi(argv[3]) : 99;
  g_map = argc > 4 ? atoi(argv[4]) : 0;
  std::vector<Shape> shapes = {
      {"small ragged", 1000, 200, 192, 3, 8, 1, 1, 1},
      {"small gelu", 4096 + 64, 512, 256, 1, 1, 2, 1, 1},
      {"odd stages", 1000, 200, 96, 3, 8, 1, 1, 1},
      {"tdf L2 gemm2", 506880, 768, 96, 144, 64, 1, 1, 0},
      {"tdf L0 gemm1", 675840, 384, 3072, 48, 256, 1, 0, 0},
      {"tdf L0 gemm2", 675840, 3072, 384, 48, 256, 1, 1, 0},
      {"tdf L1 gemm1", 675840, 192, 1536, 96, 128, 1, 0, 0},
      {"tdf L1 gemm2", 675840, 1536, 192, 96, 128, 1, 1, 0},
      {"rof ff1", 480000, 2048, 512, 1, 1, 2, 0, 1},
      {"rof ff2", 480000, 512, 2048, 1, 1, 0, 1, 1},
      {"rof qkv", 480000, 1536, 512, 1, 1, 0, 0, 0},
      {"ht lin", 43008, 1536, 384, 1, 1, 2, 0, 1},
  };
  for (int i = first; i < (int)shapes.size() && i <= last; ++i) run_shape(shapes[i], abl, 5);
  return 0;
}
