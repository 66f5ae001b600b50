// Stand-alone harness of the bf16x6 Winograd kernel (python-audio-separator_amd/csrc/kernels_wino6.h): a float64 direct convolution on small shapes
// (borders, ragged sizes, channel padding), then time per launch on the HQ_3 level shapes against conv_wino3_kernel.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/experimental/proto_wino6 tools/experimental/proto_wino6.hip
//   tools/experimental/proto_wino6 [abl] [first shape] [last shape] [nt switches] [grid] [one: 1 = one workgroup per (tile, channel group)]
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

#include "../../python-audio-separator_amd/csrc/kernels_net.h"
#include "../../python-audio-separator_amd/csrc/kernels_wino.h"
#include "../../python-audio-separator_amd/csrc/kernels_wino6.h"

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
  int B, Cin, Cout, T, F, act, res, check;
};

static void pack_wu3(const std::vector<float> &w, int cout, int cin, std::vector<float> &wu3, int *cg, int *nci) {
  static const double G[4][3] = {{1, 0, 0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0, 0, 1}};
  *cg = (cout + 47) / 48;
  *nci = ((cin + 7) / 8) * 2;
  wu3.assign((size_t)*cg * *nci * Wino3Cfg::USTAGE, 0.f);
  for (int co = 0; co < cout; ++co)
    for (int c = 0; c < cin; ++c) {
      const float *g = &w[((size_t)co * cin + c) * 9];
      double t[4][3];
      for (int a = 0; a < 4; ++a)
        for (int j = 0; j < 3; ++j) t[a][j] = G[a][0] * g[0 * 3 + j] + G[a][1] * g[1 * 3 + j] + G[a][2] * g[2 * 3 + j];
      const int cgi = co / 48, col = co % 48;
      float *dst3 = &wu3[((size_t)cgi * *nci + c / 4) * Wino3Cfg::USTAGE + ((size_t)(c % 4) * 16 + col % 16) * Wino3Cfg::ULS];
      for (int a = 0; a < 4; ++a)
        for (int bb = 0; bb < 4; ++bb) dst3[(a * 4 + bb) * 3 + col / 16] = (float)(t[a][0] * G[bb][0] + t[a][1] * G[bb][1] + t[a][2] * G[bb][2]);
    }
}

static int g_one = 0;
static int g_h = 0;   // 1: the fp16 x 3 arithmetic (argv[7]; one-workgroup-per-item form only)
template <int ABL>
static void launch6(const ConvArgs &a, int nb) {
  static bool done = false;
  if (g_h) {
    static bool hdone = false;
    if (!hdone) {
      CK(hipFuncSetAttribute(reinterpret_cast<const void *>(&conv_wino6_kernel<0, 1, true>), hipFuncAttributeMaxDynamicSharedMemorySize, Wino6Cfg::LDS_BYTES));
      hdone = true;
    }
    const int S = a.tilesT * a.tilesF * a.B;
    hipLaunchKernelGGL((conv_wino6_kernel<0, 1, true>), dim3(((S + 7) / 8) * 8 * a.CG), dim3(512), Wino6Cfg::LDS_BYTES, 0, a);
    return;
  }
  if (!done) {
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(&conv_wino6_kernel<ABL, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, Wino6Cfg::LDS_BYTES));
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(&conv_wino6_kernel<ABL, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, Wino6Cfg::LDS_BYTES));
    done = true;
  }
  if (g_one) {
    const int S = a.tilesT * a.tilesF * a.B;
    hipLaunchKernelGGL((conv_wino6_kernel<ABL, 1>), dim3(((S + 7) / 8) * 8 * a.CG), dim3(512), Wino6Cfg::LDS_BYTES, 0, a);
  } else {
    hipLaunchKernelGGL((conv_wino6_kernel<ABL, 0>), dim3(nb), dim3(512), Wino6Cfg::LDS_BYTES, 0, a);
  }
}

static int g_nt = 0, g_grid = 256;
static void run_shape(const Shape &sh, int abl, int reps) {
  const int B = sh.B, Cin = sh.Cin, Cout = sh.Cout, T = sh.T, F = sh.F;
  std::mt19937 rng(77 + Cin * 3 + Cout);
  std::normal_distribution<float> nd(0.f, 1.f);
  const size_t plane = (size_t)T * F;
  const size_t nx = (size_t)B * Cin * plane, ny = (size_t)B * Cout * plane;
  // host data: one batch item, replicated
  std::vector<float> hx((size_t)Cin * plane), hw((size_t)Cout * Cin * 9), hb(((Cout + 47) / 48) * 48, 0.f), hr((size_t)Cout * plane);
  for (auto &v : hx) v = nd(rng) * 2.0f;
  for (auto &v : hw) v = nd(rng) / std::sqrt(9.f * Cin);
  for (int i = 0; i < Cout; ++i) hb[i] = 0.3f * nd(rng);
  for (auto &v : hr) v = nd(rng);
  float *dx, *dy3, *dy6, *db, *dr, *dz, *dwu3;
  uint32_t *dw6;
  CK(hipMalloc(&dx, nx * 4));
  CK(hipMalloc(&dy3, ny * 4));
  CK(hipMalloc(&dy6, ny * 4));
  CK(hipMalloc(&dr, ny * 4));
  CK(hipMalloc(&db, hb.size() * 4));
  CK(hipMalloc(&dz, 4096));
  CK(hipMemset(dz, 0, 4096));
  for (int b = 0; b < B; ++b) {
    CK(hipMemcpy(dx + (size_t)b * Cin * plane, hx.data(), hx.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dr + (size_t)b * Cout * plane, hr.data(), hr.size() * 4, hipMemcpyHostToDevice));
  }
  CK(hipMemcpy(db, hb.data(), hb.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemset(dy3, 0xff, ny * 4));
  CK(hipMemset(dy6, 0xff, ny * 4));
  std::vector<float> wu3;
  int cg3, nci3;
  pack_wu3(hw, Cout, Cin, wu3, &cg3, &nci3);
  CK(hipMalloc(&dwu3, wu3.size() * 4));
  CK(hipMemcpy(dwu3, wu3.data(), wu3.size() * 4, hipMemcpyHostToDevice));
  std::vector<uint32_t> w6;
  int cg6, nci6;
  if (g_h) wino6_pack_h(hw.data(), Cout, Cin, w6, &cg6, &nci6);
  else wino6_pack(hw.data(), Cout, Cin, w6, &cg6, &nci6);
  CK(hipMalloc(&dw6, w6.size() * 4));
  CK(hipMemcpy(dw6, w6.data(), w6.size() * 4, hipMemcpyHostToDevice));

  ConvArgs a{};
  a.x = dx;
  a.bias = db;
  a.res = sh.res ? dr : nullptr;
  a.zeros = dz;
  a.B = B;
  a.Cin = Cin;
  a.Cout = Cout;
  a.T = T;
  a.F = F;
  a.To = T;
  a.Fo = F;
  a.act = sh.act;
  a.nt = g_nt;
  a.x_bstride = (int64_t)Cin * plane;
  a.y_bstride = (int64_t)Cout * plane;
  a.aux_bstride = (int64_t)Cout * plane;
  a.tilesT = (T + 7) / 8;
  a.tilesF = (F + 31) / 32;

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
  ConvArgs a3 = a;
  a3.wp = dwu3;
  a3.CG = cg3;
  a3.NCI = nci3;
  a3.y = dy3;
  const int nb3 = cg3 * a.tilesT * a.tilesF * B;
  constexpr int lds3 = Wino3CfgT<4, 2, 1>::LDS_BYTES;
  CK(hipFuncSetAttribute(reinterpret_cast<const void *>(&conv_wino3_kernel<0, 4, 2, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, lds3));
  const double t3 = time_it([&]() { hipLaunchKernelGGL((conv_wino3_kernel<0, 4, 2, 1>), dim3(nb3), dim3(256), lds3, 0, a3); });
  ConvArgs a6 = a;
  a6.wp = reinterpret_cast<const float *>(dw6);
  a6.CG = cg6;
  a6.NCI = nci6;
  a6.y = dy6;
  const int nsp6 = a.tilesT * a.tilesF * B;
  const int nb6 = nsp6 < g_grid ? nsp6 : g_grid;   // persistent: one workgroup per CU (or fewer)
  double t6 = 0;
  switch (abl) {
    case 0: t6 = time_it([&]() { launch6<0>(a6, nb6); }); break;
    case 1: t6 = time_it([&]() { launch6<1>(a6, nb6); }); break;
    case 2: t6 = time_it([&]() { launch6<2>(a6, nb6); }); break;
    case 4: t6 = time_it([&]() { launch6<4>(a6, nb6); }); break;
    case 8: t6 = time_it([&]() { launch6<8>(a6, nb6); }); break;
    case 16: t6 = time_it([&]() { launch6<16>(a6, nb6); }); break;
    case 15: t6 = time_it([&]() { launch6<15>(a6, nb6); }); break;
    default: fprintf(stderr, "abl?\n"); exit(2);
  }
  CK(hipGetLastError());
  CK(hipDeviceSynchronize());
  const double flops = 2.0 * B * Cout * Cin * 9.0 * plane;

  double e3 = 0, e6 = 0, nrm = 0, mx3 = 0, mx6 = 0;
  long bad6 = 0, nbad = 0;
  std::vector<long> badco(Cout, 0), badt(T, 0), badf(64, 0);
  if (sh.check) {
    // float64 direct convolution of the LAST batch item (all items hold the same data)
    std::vector<float> y3((size_t)Cout * plane), y6((size_t)Cout * plane);
    CK(hipMemcpy(y3.data(), dy3 + (size_t)(B - 1) * Cout * plane, y3.size() * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(y6.data(), dy6 + (size_t)(B - 1) * Cout * plane, y6.size() * 4, hipMemcpyDeviceToHost));
    for (int co = 0; co < Cout; ++co)
      for (int t = 0; t < T; ++t)
        for (int f = 0; f < F; ++f) {
          double acc = hb[co];
          for (int c = 0; c < Cin; ++c)
            for (int dy = 0; dy < 3; ++dy) {
              const int tt = t + dy - 1;
              if (tt < 0 || tt >= T) continue;
              for (int dxx = 0; dxx < 3; ++dxx) {
                const int ff = f + dxx - 1;
                if (ff < 0 || ff >= F) continue;
                acc += (double)hx[(size_t)c * plane + (size_t)tt * F + ff] * (double)hw[((size_t)co * Cin + c) * 9 + dy * 3 + dxx];
              }
            }
          if (sh.act == ACT_RELU) acc = acc > 0 ? acc : 0;
          if (sh.res) acc += hr[(size_t)co * plane + (size_t)t * F + f];
          const size_t i = (size_t)co * plane + (size_t)t * F + f;
          if (!std::isfinite(y6[i])) ++bad6;
          if (std::fabs(y6[i] - acc) > 1e-3 * (1.0 + std::fabs(acc))) {
            if (nbad < 6) printf("   bad co=%d t=%d f=%d got %g want %g (wino3 %g)\n", co, t, f, y6[i], acc, y3[i]);
            ++nbad;
            badco[co]++;
            badt[t]++;
            badf[f % 64]++;
          }
          e3 += (y3[i] - acc) * (y3[i] - acc);
          e6 += (y6[i] - acc) * (y6[i] - acc);
          nrm += acc * acc;
          mx3 = std::max(mx3, std::fabs(y3[i] - acc));
          mx6 = std::max(mx6, std::fabs(y6[i] - acc));
        }
  }
  printf("%-14s B=%-3d Cin=%-4d Cout=%-4d T=%-4d F=%-5d wino3 %8.3f ms %6.1f TF-alg | wino6 %8.3f ms %6.1f TF-alg (x%.2f)", sh.name, B, Cin, Cout, T, F,
         t3, flops / t3 * 1e-9, t6, flops / t6 * 1e-9, t3 / t6);
  if (sh.check)
    printf(" | relrms vs f64: wino3 %.2e wino6 %.2e maxabs %.2e / %.2e nonfinite %ld bad %ld", std::sqrt(e3 / nrm), std::sqrt(e6 / nrm), mx3, mx6, bad6, nbad);
  printf("\n");
  if (nbad) {
    printf("   bad by cout:");
    for (int i = 0; i < Cout; ++i) if (badco[i]) printf(" %d:%ld", i, badco[i]);
    printf("\n   bad by t:");
    for (int i = 0; i < T; ++i) if (badt[i]) printf(" %d:%ld", i, badt[i]);
    printf("\n   bad by f%%64:");
    for (int i = 0; i < 64; ++i) if (badf[i]) printf(" %d:%ld", i, badf[i]);
    printf("\n");
  }
  fflush(stdout);
  CK(hipFree(dx));
  CK(hipFree(dy3));
  CK(hipFree(dy6));
  CK(hipFree(dr));
  CK(hipFree(db));
  CK(hipFree(dz));
  CK(hipFree(dwu3));
  CK(hipFree(dw6));
}

int main(int argc, char **argv) {
  const int abl = argc > 1 ? atoi(argv[1]) : 0;
  const int first = argc > 2 ? atoi(argv[2]) : 0;
  const int last = argc > 3 ? atoi(argv[3]) : 99;
  g_nt = argc > 4 ? atoi(argv[4]) : 0;
  g_grid = argc > 5 ? atoi(argv[5]) : 256;
  g_one = argc > 6 ? atoi(argv[6]) : 0;
  g_h = argc > 7 ? atoi(argv[7]) : 0;
  std::vector<Shape> shapes = {
      {"one wg 48", 1, 48, 48, 8, 32, ACT_NONE, 0, 1},
      {"multi wg 40", 1, 40, 20, 16, 64, ACT_NONE, 0, 1},
      {"one wg 32", 1, 32, 48, 8, 32, ACT_NONE, 0, 1},
      {"two wg 32", 1, 32, 48, 8, 64, ACT_NONE, 0, 1},
      {"small", 2, 48, 48, 16, 64, ACT_RELU, 0, 1},
      {"small pad", 1, 40, 20, 8, 32, ACT_NONE, 1, 1},
      {"small ragged", 2, 96, 96, 10, 44, ACT_RELU, 1, 1},
      {"small 144", 1, 144, 144, 16, 96, ACT_RELU, 0, 1},
      {"L0 48", 55, 48, 48, 256, 3072, ACT_RELU, 0, 0},
      {"L1 96", 55, 96, 96, 128, 1536, ACT_RELU, 0, 0},
      {"L2 144", 55, 144, 144, 64, 768, ACT_RELU, 0, 0},
      {"L3 192", 55, 192, 192, 32, 384, ACT_RELU, 0, 0},
      {"L4 240", 55, 240, 240, 16, 192, ACT_RELU, 0, 0},
      {"L5 288", 55, 288, 288, 8, 96, ACT_RELU, 0, 0},
  };
  for (int i = first; i < (int)shapes.size() && i <= last; ++i) run_shape(shapes[i], abl, 3);
  return 0;
}
