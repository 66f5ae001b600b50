// Stand-alone harness of the direct fp16 x 3 convolution kernel (python-audio-separator_amd/csrc/kernels_conv3h.h): a float64 direct
// convolution on small shapes (borders, ragged sizes) and on sampled outputs of the HQ_3 level-0 shape, then time per launch.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/proto_conv3h tools/proto_conv3h.hip
//   tools/proto_conv3h [abl] [order] [B of the timing shape] [reps]
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

#include "../python-audio-separator_amd/csrc/kernels_conv3h.h"

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
  int B, T, F, act, full_check;
  float spread;   // decades of magnitude spread between tiles (block-exponent stress)
};

template <int ABL>
static void launch(const Conv3hArgs &a, hipStream_t s) {
  static bool done = false;
  if (!done) {
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(&conv3h_kernel<ABL, false>), hipFuncAttributeMaxDynamicSharedMemorySize, Conv3hCfg::LDS_BYTES));
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(&conv3h_kernel<ABL, true>), hipFuncAttributeMaxDynamicSharedMemorySize, Conv3hCfg::LDS_BYTES));
    done = true;
  }
  if (a.prev) hipLaunchKernelGGL((conv3h_kernel<ABL, true>), dim3(256), dim3(512), Conv3hCfg::LDS_BYTES, s, a);
  else hipLaunchKernelGGL((conv3h_kernel<ABL, false>), dim3(256), dim3(512), Conv3hCfg::LDS_BYTES, s, a);
}
static void launch_abl(int abl, const Conv3hArgs &a, hipStream_t s) {
  switch (abl) {
    case 1: return launch<1>(a, s);
    case 2: return launch<2>(a, s);
    case 4: return launch<4>(a, s);
    case 8: return launch<8>(a, s);
    case 6: return launch<6>(a, s);
    case 7: return launch<7>(a, s);
    case 14: return launch<14>(a, s);
    case 32: return launch<32>(a, s);
    case 64: return launch<64>(a, s);
    case 128: return launch<128>(a, s);
    case 160: return launch<160>(a, s);
    case 288: return launch<288>(a, s);
    case 416: return launch<416>(a, s);
    case 512: return launch<512>(a, s);
    case 1024: return launch<1024>(a, s);
    case 2048: return launch<2048>(a, s);
    case 2080: return launch<2080>(a, s);
    case 34: return launch<34>(a, s);
    case 36: return launch<36>(a, s);
    case 38: return launch<38>(a, s);
    case 40: return launch<40>(a, s);
    case 96: return launch<96>(a, s);
    default: return launch<0>(a, s);
  }
}

static double ref_at(const std::vector<float> &x, const std::vector<float> &w, const std::vector<float> &bias, int T, int F, int b, int co, int t, int f, int act, double *mag) {
  const int C = 48;
  double s = bias[co], m = std::fabs((double)bias[co]);
  for (int ci = 0; ci < C; ++ci)
    for (int ky = 0; ky < 3; ++ky)
      for (int kx = 0; kx < 3; ++kx) {
        const int tt = t + ky - 1, ff = f + kx - 1;
        if (tt < 0 || tt >= T || ff < 0 || ff >= F) continue;
        const double p = (double)w[((size_t)co * C + ci) * 9 + ky * 3 + kx] * (double)x[(((size_t)b * C + ci) * T + tt) * F + ff];
        s += p;
        m += std::fabs(p);
      }
  *mag = m;
  if (act == ACT_RELU && s < 0) s = 0;
  return s;
}

static int g_bw = 32;      // strips per band (argv[7]: 32, 16, 8)
static int g_alias = 0;   // 1: every batch item reads item 0's planes, 2: and writes item 0's output (cache-resident traffic: is the launch DRAM- or CU-bound?)
static int run_shape(const Shape &sh, int abl, int order, int reps) {
  const int C = 48;
  const size_t n = (size_t)sh.B * C * sh.T * sh.F;
  std::mt19937 rng(1234);
  std::normal_distribution<float> nd(0.f, 1.f);
  std::uniform_real_distribution<float> ud(0.f, 1.f);
  std::vector<float> w((size_t)C * C * 9), bias(C);
  for (int co = 0; co < C; ++co) {
    const float cs = std::pow(10.f, 2.f * (ud(rng) - 0.5f)) * 0.05f;   // per-output-channel scale (a folded BatchNorm)
    for (int i = 0; i < C * 9; ++i) w[(size_t)co * C * 9 + i] = nd(rng) * cs;
    bias[co] = nd(rng) * 0.1f;
  }
  std::vector<float> x;
  const bool big = n > ((size_t)1 << 28);
  float *dx = nullptr, *dy = nullptr;
  CK(hipMalloc(&dx, n * 4));
  CK(hipMalloc(&dy, n * 4));
  CK(hipMemset(dy, 0xff, n * 4));
  // host data for ONE batch item, repeated over the batch (the big shape would be 8 GB of host memory otherwise)
  const size_t nb = (size_t)C * sh.T * sh.F;
  x.resize(nb);
  for (int c = 0; c < C; ++c)
    for (int t = 0; t < sh.T; ++t)
      for (int f = 0; f < sh.F; ++f) {
        const float amp = std::pow(10.f, sh.spread * (std::sin(0.013f * f) * std::cos(0.21f * t)));
        x[((size_t)c * sh.T + t) * sh.F + f] = nd(rng) * amp;
      }
  for (int b = 0; b < sh.B; ++b) CK(hipMemcpy(dx + (size_t)b * nb, x.data(), nb * 4, hipMemcpyHostToDevice));
  std::vector<uint32_t> img;
  conv3h_pack(w.data(), img);
  uint32_t *dimg = nullptr;
  float *dbias = nullptr;
  CK(hipMalloc(&dimg, img.size() * 4));
  CK(hipMalloc(&dbias, C * 4));
  CK(hipMemcpy(dimg, img.data(), img.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dbias, bias.data(), C * 4, hipMemcpyHostToDevice));

  Conv3hArgs a{};
  a.x = dx;
  a.y = dy;
  a.wimg = reinterpret_cast<const u32x4 *>(dimg);
  a.bias = dbias;
  a.B = sh.B;
  a.T = sh.T;
  a.F = sh.F;
  a.x_bstride = a.y_bstride = (int64_t)nb;
  a.act = sh.act;
  a.tilesT = (sh.T + 3) / 4;
  a.tilesF = (sh.F + 31) / 32;
  a.bw = g_bw;
  a.tps = a.tilesT * g_bw / 32;
  if ((a.tilesT * g_bw) % 32 != 0) {
    a.bw = 32;
    a.tps = a.tilesT;
  }
  long long *ddbg = nullptr;
  CK(hipMalloc(&ddbg, (512 + 4 * 2048) * 8));
  CK(hipMemset(ddbg, 0, (512 + 4 * 2048) * 8));
  a.dbg = ddbg;
  launch_abl(abl, a, 0);
  CK(hipDeviceSynchronize());

  int bad = 0;
  if (abl == 0) {
    // batch items are copies: compare item 0 (all of it or a sample) and the last item against item 0 bit for bit
    std::vector<float> y0(nb), yl(nb);
    CK(hipMemcpy(y0.data(), dy, nb * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(yl.data(), dy + (size_t)(sh.B - 1) * nb, nb * 4, hipMemcpyDeviceToHost));
    if (memcmp(y0.data(), yl.data(), nb * 4) != 0) {
      printf("  %s: batch item %d differs from item 0\n", sh.name, sh.B - 1);
      ++bad;
    }
    double num = 0, den = 0, worst = 0;
    size_t cnt = 0;
    auto check = [&](int co, int t, int f) {
      double mag = 0;
      const double r = ref_at(x, w, bias, sh.T, sh.F, 0, co, t, f, sh.act, &mag);
      const double g = y0[((size_t)co * sh.T + t) * sh.F + f];
      const double d = g - r;
      // the bar is relative to sum |w x| over the taps (what an fp32 chain's rounding is relative to): 2e-6 of it
      num += d * d;
      den += r * r;
      worst = std::max(worst, std::fabs(d) / (mag + 1e-30));
      if (!(std::fabs(d) <= 2e-6 * mag + 1e-30)) {
        if (bad < 8) printf("  %s: mismatch at co %d t %d f %d: got %.9g want %.9g\n", sh.name, co, t, f, g, r);
        ++bad;
      }
      ++cnt;
    };
    if (sh.full_check) {
      for (int co = 0; co < C; ++co)
        for (int t = 0; t < sh.T; ++t)
          for (int f = 0; f < sh.F; ++f) check(co, t, f);
    } else {
      std::mt19937 r2(99);
      for (int i = 0; i < 20000; ++i) check((int)(r2() % C), (int)(r2() % sh.T), (int)(r2() % sh.F));
      for (int co = 0; co < C; co += 7)
        for (int t : {0, 1, 3, 4, sh.T - 1})
          for (int f : {0, 1, 31, 32, 33, sh.F - 33, sh.F - 1}) check(co, t, f);
    }
    printf("  %s: %zu outputs checked, rel RMS %.3e, worst |d| / sum |w x| %.3e, %s\n", sh.name, cnt, std::sqrt(num / std::max(den, 1e-300)), worst, bad ? "FAIL" : "ok");
  }
  if (abl & 32) {
    launch_abl(abl, a, 0);
    CK(hipDeviceSynchronize());
    std::vector<long long> d(512 + 4 * 2048);
    CK(hipMemcpy(d.data(), ddbg, d.size() * 8, hipMemcpyDeviceToHost));
    for (int w = 0; w < 4; ++w) {
      const long long *q = &d[512 + w * 2048];
      int n = 0;
      while (n < 2048 && q[n] != 0) ++n;
      printf("  workgroup slot %d: %d steps, %lld ticks in all; ticks per step by item (64 steps):", w, n, n ? q[n - 1] - q[0] : 0);
      for (int i = 0; i + 64 < n; i += 64) printf(" %lld", (q[i + 64] - q[i]) / 64);
      printf("\n");
    }
    printf("  timeline of workgroup 0 (s_memtime ticks at 100 MHz? printed raw, relative to step 8's start): producer wave 4: start, split done, loads issued, max done; consumer wave 0: start, MFMAs done, stores issued\n");
    for (int st = 8; st < 40; ++st) {
      printf("  step %2d:", st);
      for (int k = 0; k < 7; ++k) printf(" %8lld", d[st * 8 + k] - d[8 * 8]);
      printf("\n");
    }
  }
  if (g_alias >= 1) a.x_bstride = 0;
  if (g_alias >= 2) a.y_bstride = 0;
  if (reps > 0) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int i = 0; i < 2; ++i) launch_abl(abl, a, 0);
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < reps; ++i) launch_abl(abl, a, 0);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    ms /= reps;
    const double flops = 2.0 * 9 * C * C * (double)sh.B * sh.T * sh.F;
    printf("  %s: abl %d order %d: %.3f ms per launch, %.1f TFLOP/s algorithmic (x3 products: %.3f of 2.5 PF), %.2f TB/s algorithmic traffic\n", sh.name, abl, order, ms,
           flops / ms * 1e-9, 3 * flops / ms * 1e-9 / 2500.0, 2.0 * n * 4 / ms * 1e-9);
  }
  CK(hipFree(dx));
  CK(hipFree(dy));
  CK(hipFree(dimg));
  CK(hipFree(dbias));
  return bad;
}

// A 96 -> 96 layer as four launches of the 48-channel kernel: for each half of the output channels, the first half of the input channels
// (bias, no activation), then the second half ADDED to it (a.prev = the output itself) with the activation.
static int run_shape96(const Shape &sh, int reps) {
  const int C = 96;
  const size_t nb = (size_t)C * sh.T * sh.F, n = nb * sh.B;
  std::mt19937 rng(4321);
  std::normal_distribution<float> nd(0.f, 1.f);
  std::uniform_real_distribution<float> ud(0.f, 1.f);
  std::vector<float> w((size_t)C * C * 9), bias(C), x(nb);
  for (int co = 0; co < C; ++co) {
    const float cs = std::pow(10.f, 2.f * (ud(rng) - 0.5f)) * 0.035f;
    for (int i = 0; i < C * 9; ++i) w[(size_t)co * C * 9 + i] = nd(rng) * cs;
    bias[co] = nd(rng) * 0.1f;
  }
  for (int c = 0; c < C; ++c)
    for (int t = 0; t < sh.T; ++t)
      for (int f = 0; f < sh.F; ++f) x[((size_t)c * sh.T + t) * sh.F + f] = nd(rng) * std::pow(10.f, sh.spread * (std::sin(0.013f * f) * std::cos(0.21f * t)));
  float *dx, *dy, *dbias;
  CK(hipMalloc(&dx, n * 4));
  CK(hipMalloc(&dy, n * 4));
  CK(hipMalloc(&dbias, C * 4));
  CK(hipMemset(dy, 0xff, n * 4));
  for (int b = 0; b < sh.B; ++b) CK(hipMemcpy(dx + (size_t)b * nb, x.data(), nb * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dbias, bias.data(), C * 4, hipMemcpyHostToDevice));
  uint32_t *dimg[2][2];
  for (int ob = 0; ob < 2; ++ob)
    for (int ib = 0; ib < 2; ++ib) {
      std::vector<float> ws((size_t)48 * 48 * 9);
      for (int co = 0; co < 48; ++co)
        for (int ci = 0; ci < 48; ++ci)
          for (int k = 0; k < 9; ++k) ws[((size_t)co * 48 + ci) * 9 + k] = w[((size_t)(ob * 48 + co) * C + ib * 48 + ci) * 9 + k];
      std::vector<uint32_t> img;
      conv3h_pack(ws.data(), img);
      CK(hipMalloc(&dimg[ob][ib], img.size() * 4));
      CK(hipMemcpy(dimg[ob][ib], img.data(), img.size() * 4, hipMemcpyHostToDevice));
    }
  long long *ddbg = nullptr;
  CK(hipMalloc(&ddbg, (512 + 4 * 2048) * 8));
  auto go = [&]() {
    for (int ob = 0; ob < 2; ++ob)
      for (int ib = 0; ib < 2; ++ib) {
        Conv3hArgs a{};
        a.x = dx + (size_t)ib * 48 * sh.T * sh.F;
        a.y = dy + (size_t)ob * 48 * sh.T * sh.F;
        a.prev = ib ? a.y : nullptr;
        a.wimg = reinterpret_cast<const u32x4 *>(dimg[ob][ib]);
        a.bias = ib ? nullptr : dbias + ob * 48;
        a.B = sh.B;
        a.T = sh.T;
        a.F = sh.F;
        a.x_bstride = a.y_bstride = (int64_t)nb;
        a.act = ib ? sh.act : ACT_NONE;
        a.tilesT = (sh.T + 3) / 4;
        a.tilesF = sh.F / 32;
        a.bw = g_bw;
        a.tps = a.tilesT * g_bw / 32;
        if ((a.tilesT * g_bw) % 32 != 0) {
          a.bw = 32;
          a.tps = a.tilesT;
        }
        a.dbg = ddbg;
        launch<0>(a, 0);
      }
  };
  go();
  CK(hipDeviceSynchronize());
  int bad = 0;
  std::vector<float> y0(nb), yl(nb);
  CK(hipMemcpy(y0.data(), dy, nb * 4, hipMemcpyDeviceToHost));
  CK(hipMemcpy(yl.data(), dy + (size_t)(sh.B - 1) * nb, nb * 4, hipMemcpyDeviceToHost));
  if (memcmp(y0.data(), yl.data(), nb * 4) != 0) {
    printf("  %s: batch item %d differs from item 0\n", sh.name, sh.B - 1);
    ++bad;
  }
  double num = 0, den = 0, worst = 0;
  size_t cnt = 0;
  auto check = [&](int co, int t, int f) {
    double s = bias[co], mag = std::fabs((double)bias[co]);
    for (int ci = 0; ci < C; ++ci)
      for (int ky = 0; ky < 3; ++ky)
        for (int kx = 0; kx < 3; ++kx) {
          const int tt = t + ky - 1, ff = f + kx - 1;
          if (tt < 0 || tt >= sh.T || ff < 0 || ff >= sh.F) continue;
          const double p = (double)w[((size_t)co * C + ci) * 9 + ky * 3 + kx] * (double)x[((size_t)ci * sh.T + tt) * sh.F + ff];
          s += p;
          mag += std::fabs(p);
        }
    if (sh.act == ACT_RELU && s < 0) s = 0;
    const double g = y0[((size_t)co * sh.T + t) * sh.F + f], d = g - s;
    num += d * d;
    den += s * s;
    worst = std::max(worst, std::fabs(d) / (mag + 1e-30));
    if (!(std::fabs(d) <= 2e-6 * mag + 1e-30)) {
      if (bad < 8) printf("  %s: mismatch at co %d t %d f %d: got %.9g want %.9g\n", sh.name, co, t, f, g, s);
      ++bad;
    }
    ++cnt;
  };
  if (sh.full_check) {
    for (int co = 0; co < C; ++co)
      for (int t = 0; t < sh.T; ++t)
        for (int f = 0; f < sh.F; ++f) check(co, t, f);
  } else {
    std::mt19937 r2(99);
    for (int i = 0; i < 20000; ++i) check((int)(r2() % C), (int)(r2() % sh.T), (int)(r2() % sh.F));
  }
  printf("  %s (96 channels, four launches): %zu outputs checked, rel RMS %.3e, worst |d| / sum |w x| %.3e, %s\n", sh.name, cnt, std::sqrt(num / std::max(den, 1e-300)), worst,
         bad ? "FAIL" : "ok");
  if (reps > 0) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    go();
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < reps; ++i) go();
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("  %s: %.3f ms per 96 -> 96 layer (four launches), bw %d\n", sh.name, ms / reps, g_bw);
  }
  CK(hipFree(dx));
  CK(hipFree(dy));
  return bad;
}

int main(int argc, char **argv) {
  const int abl = argc > 1 ? atoi(argv[1]) : 0;
  const int order = argc > 2 ? atoi(argv[2]) : 0;
  const int Bt = argc > 3 ? atoi(argv[3]) : 55;
  const int reps = argc > 4 ? atoi(argv[4]) : 5;
  const int quick = argc > 5 ? atoi(argv[5]) : 0;      // 1: the timing shape only (profiling runs)
  g_alias = argc > 6 ? atoi(argv[6]) : 0;
  g_bw = argc > 7 ? atoi(argv[7]) : 32;
  int bad = 0;
  if (abl == 0 && !quick) {
    const Shape small[] = {
        {"small 2x8x64", 2, 8, 64, ACT_RELU, 1, 0.f},
        {"ragged 3x10x96", 3, 10, 96, ACT_NONE, 1, 0.f},
        {"one tile row 1x4x32", 1, 4, 32, ACT_RELU, 1, 0.f},
        {"tall 1x37x96 spread", 1, 37, 96, ACT_NONE, 1, 3.f},
        {"tall 2x67x64 wide spread", 2, 67, 64, ACT_NONE, 1, 8.f},
        {"wide 2x5x1088", 2, 5, 1088, ACT_RELU, 1, 2.f},
    };
    for (const Shape &s : small)
      bad += run_shape(s, 0, 2, 0);
  }
  if (quick == 2) {   // level 1 as four launches
    const Shape s96[] = {{"small96 2x8x64", 2, 8, 64, ACT_RELU, 1, 0.f}, {"ragged96 1x18x96", 1, 18, 96, ACT_NONE, 1, 2.f}};
    for (const Shape &s : s96) bad += run_shape96(s, 0);
    const Shape l1 = {"level 1", Bt, 128, 1536, ACT_RELU, 0, 1.f};
    bad += run_shape96(l1, reps);
    printf(bad ? "FAILED (%d)\n" : "all ok\n", bad);
    return bad ? 1 : 0;
  }
  const Shape l0 = {"level 0", Bt, 256, 3072, ACT_RELU, 0, 1.f};
  bad += run_shape(l0, abl, order, reps);
  printf(bad ? "FAILED (%d)\n" : "all ok\n", bad);
  return bad ? 1 : 0;
}
