// libasx.so -- engine + C ABI (include/asx.h) of the MI355X demix path.
//
// One engine = one GPU.  The engine owns: FFT tables, the packed ConvTDFNet
// weights, and a device workspace sized for `max_batch` chunks.  The chunk loop
// of MDXSeparator.demix (mdx_separator.py:348-392) is executed as batches of
// independent chunks: STFT -> net -> iSTFT -> fold+window per batch, then one
// gather pass folds all windowed chunks into the song (result / divider).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <map>
#include <string>
#include <mutex>
#include <set>
#include <vector>

#include "../../include/asx.h"
#include "kernels_fft.h"
#include "kernels_fft3.h"
#include "kernels_net.h"
#include "kernels_gemm2.h"
#include "kernels_gemm3.h"
#include "kernels_wino.h"
#include "kernels_winos.h"
#include "kernels_wino6.h"
#ifndef ASX_TDF2_DEFAULT
#define ASX_TDF2_DEFAULT 1
#endif
#include "kernels_rof.h"
#include "kernels_ht.h"
#include "kernels_halo.h"
#include "kernels_hd.h"
#include "kernels_vr.h"
#include "kernels_ens.h"

using namespace asx;

// ----------------------------------------------------------------------------
// errors
// ----------------------------------------------------------------------------
static thread_local char g_err[512] = "";
static void set_err(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
#define HIPCHK(x)                                                                       \
  do {                                                                                  \
    hipError_t e_ = (x);                                                                \
    if (e_ != hipSuccess) {                                                             \
      set_err("%s:%d %s -> %s", __FILE__, __LINE__, #x, hipGetErrorString(e_));         \
      return ASX_ERR_HIP;                                                               \
    }                                                                                   \
  } while (0)
#define CHK(x)                    \
  do {                            \
    int r_ = (x);                 \
    if (r_ != ASX_OK) return r_;  \
  } while (0)
#define REQUIRE(cond, ...)        \
  do {                            \
    if (!(cond)) {                \
      set_err(__VA_ARGS__);       \
      return ASX_ERR_INVALID;     \
    }                             \
  } while (0)

// ----------------------------------------------------------------------------
// device buffer helper
// ----------------------------------------------------------------------------
struct DevBuf {
  void *p = nullptr;
  size_t bytes = 0;
  int ensure(size_t n) {
    if (n <= bytes) return ASX_OK;
    if (p) (void)hipFree(p);
    p = nullptr;
    bytes = 0;
    HIPCHK(hipMalloc(&p, n));
    bytes = n;
    // ASX_POISON=<byte 0..255>: fill every fresh allocation with that byte (255 = NaN, 127 = 3.39e38 floats) -- a debugging aid that
    // makes any read of memory the engine never wrote show up in the results, instead of depending on what the allocation held
    // before (round 5: a first forward on the bf16 x 6 kernels differed from later ones on some boxes and not on others)
    static const int poison = getenv("ASX_POISON") ? atoi(getenv("ASX_POISON")) : -1;
    if (poison >= 0) HIPCHK(hipMemset(p, poison & 255, n));
    return ASX_OK;
  }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    bytes = 0;
  }
  float *f() const { return reinterpret_cast<float *>(p); }
};

// ----------------------------------------------------------------------------
// packed layers
// ----------------------------------------------------------------------------
enum ConvKind { CK_3X3 = 0, CK_DOWN = 1, CK_1X1 = 2, CK_UP = 3 };

struct ConvLayer {
  int kind = CK_3X3;
  int cin = 0, cout = 0;
  int nrep = 0, kc = 0, cg = 0, nci = 0;
  int relu = 1;
  DevBuf w, b;
  DevBuf wu;       // CK_3X3 only: Winograd F(2x2,3x3) transformed weights [CG48][NCI8][xi][pair][48][2]
  DevBuf wu2;      // the same values as [CG48][NCI8][xi][channel 8][48] (conv_wino2_kernel)
  DevBuf wu3;      // and as [CG48][NCI4][channel 4][cout % 16][52: (xi, cout / 16) in MFMA order, 4 pad] (conv_wino3_kernel)
  int wu_cg = 0, wu_nci = 0, wu3_nci = 0;
  DevBuf wus;      // weight-stationary image [CG48][wave 8][6 KS / 4][lane 64][4] (conv_winos_kernel<KS>), Cin <= 96 only
  int wus_ks = 0;  // 12 / 24 (k-steps of four channels the image was packed for), 0 = none
  DevBuf gn_w, gn_b;  // GroupNorm(2, cout) affine behind this conv (asx_net_config.norm == 1), else empty
  DevBuf wu6;      // conv_wino6_kernel (kernels_wino6.h): U split three ways into bf16, MFMA-fragment order [CG48][NCI32][wave 8][18][lane 64][4 x u32]
  int wu6_nci = 0; // 32-channel stages of that image (0 = not packed: Cin < 64)
};

struct TdfLayer {
  int n = 0, k = 0, c = 0;
  bool has_bias = false;
  DevBuf w, bias, scale, shift;
  DevBuf gn_w, gn_b;  // GroupNorm(2, c) affine behind this linear (asx_net_config.norm == 1), else empty
};

struct Block {
  std::vector<ConvLayer> tfc;
  TdfLayer tdf0, tdf1;
  int c = 0, t = 0, f = 0;
};

struct ProfRec {
  int cls;
  hipEvent_t a, b;
  double flops, bytes;
};

struct V3Net;
struct RofNet;
struct HtNet;
struct HdNet;
struct VrNet;
struct EnsCtx;

// split image of one weight matrix for the bf16 x 6 kernels (kernels_gemm3.h); owned by the engine (asx_engine::w3)
struct W3Entry {
  const float *w;
  int N, K, cin;                                       // cin > 0: the (tap, chunk)-padded image of the GATHER mode, else 0
  void *img;
};

// what the fold's divider table was built for (asx_finalize_dev)
struct DivKey {
  int64_t N = -1;
  int n_chunks = 0;
  int64_t C = 0, step = 0, L = 0;
  int trim = 0, win = 0;
  int hann_tab = 0;    // 1: the window came from the fast-FFT path's float64 table (d_hann3), 0: computed in place
  bool operator==(const DivKey &o) const {
    return N == o.N && n_chunks == o.n_chunks && C == o.C && step == o.step && L == o.L && trim == o.trim && win == o.win &&
           hann_tab == o.hann_tab;
  }
};

struct asx_engine {
  int device = 0;
  V3Net *v3 = nullptr;
  RofNet *rof = nullptr;
  HtNet *ht = nullptr;
  HdNet *hd = nullptr;   // Demucs v3: owns the inner levels, e->ht the strided ones
  // workspaces of further chunk groups over the same weights (engine_hd.h) and the shared BLSTM scratch
  std::vector<HtNet *> ht_cl;
  std::vector<HdNet *> hd_cl;
  DevBuf hd_lstm_ws;
  VrNet *vr = nullptr;
  EnsCtx *ens = nullptr;
  asx_mdx_config cfg{};
  FftPlan plan{};
  DevBuf d_window, d_tw, d_env;  // env for T = segment_size
  DevBuf d_hann3;                // np.hanning(chunk_size) in float64 for the fused inverse's chunk window
  DevBuf d_tw3, seam3;           // fast FFT path (kernels_fft3.h): twiddles [16][12] + [16][192]; seam partial sums
  bool fft3 = false;             // n_fft == 6144 && hop == 1024 (and ASX_FFT3 != 0)
  bool fft3p = false;            // inverse with the LDS-DMA spectrum prefetch (ASX_FFT3P != 0)
  DevBuf d_zeros;                // zero page: source of out-of-range DMA slots
  // net
  bool net_begun = false, net_ready = false;
  asx_net_config net{};
  std::map<std::string, std::vector<float>> host_tensors;
  ConvLayer first, final_;
  std::vector<Block> enc, dec;
  Block mid;
  std::vector<ConvLayer> ds, us;
  // workspace
  int ws_batch = 0;  // chunks the workspace is sized for
  DevBuf spec_in, spec_out, R[3], H, frames, chunk_out, d_starts, d_nact, d_peak, d_demixed;
  DevBuf gn_part;    // per-plane float64 (sum, sum of squares) of the GroupNorm variant of the net (asx_net_config.norm == 1)
  DevBuf d_div;      // divider of the chunk fold for div_key's plan (input-independent: built once, asx_finalize_dev)
  DivKey div_key;
  DevBuf sinc_tab;   // coefficient table of asx_resample_sinc (built on first use)
  hipEvent_t div_ev = nullptr;       // recorded behind the kernel that built d_div; a call on ANOTHER stream waits for it
  hipStream_t div_stream = nullptr;
  std::vector<DevBuf> skip;
  // 3x3 / pad-1 convs of the ConvTDFNet and TFC-TDF-v3 nets: 3 = Winograd F(2x2,3x3) (conv_wino3_kernel, the default), 0 = the
  // direct kernel (conv_dma_kernel), 1 / 2 = the earlier Winograd generations (kept for A/B runs).  ASX_WINOGRAD or
  // asx_set_option("winograd", n).
  int winograd = getenv("ASX_WINOGRAD") ? std::max(0, atoi(getenv("ASX_WINOGRAD"))) : 3;
  // 1: layers with Cin <= 96 run the weight-stationary Winograd kernel (conv_winos_kernel, kernels_winos.h) when the option above
  // is 3; 0 (default -- the stationary form measured 3-8 % slower, profiles/NOTES.md round 4): conv_wino3_kernel everywhere.
  // ASX_WINOS or asx_set_option("winograd_stationary", n).
  int winos = getenv("ASX_WINOS") ? std::max(0, atoi(getenv("ASX_WINOS"))) : 0;
  // 1 (default): row GEMMs, channels-last convolutions (GATHER mode) and attention of THIS engine run the bf16 x 6 kernels when their
  // shapes allow (kernels_gemm3.h); 0: the fp32-MFMA kernels.  ASX_GEMM_BF16X6 or asx_set_option("gemm_bf16x6", n).
  int gemm_bf16x6 = getenv("ASX_GEMM_BF16X6") ? atoi(getenv("ASX_GEMM_BF16X6")) : 1;
  // 3x3 TFC convs with at least this many input channels run Winograd F(2x2,3x3) on the bf16 pipe (conv_wino6_kernel, kernels_wino6.h)
  // when "winograd" is 3 and "gemm_bf16x6" is on; 0 = never.  Default 144: measured faster than conv_wino3_kernel from level 2 of the
  // HQ_3 net down, equal on level 1, slower on level 0 (profiles/r05_wino6_forms.txt).  ASX_WINO6 or asx_set_option("winograd_bf16x6", n).
  int wino6 = getenv("ASX_WINO6") ? std::max(0, atoi(getenv("ASX_WINO6"))) : 144;
  // split (bf16 x 3) images of this engine's weight matrices, built on first use and freed only with the engine or when the engine's
  // own weights are re-loaded: another engine of the process can never invalidate a pointer a captured graph of this one holds
  std::vector<W3Entry> w3;
  std::mutex w3_mu;
  // profiling
  bool prof = false;
  std::vector<ProfRec> recs;
};

// default: a whole 4-minute song (55 chunks, ~45 GB of the 288 GB) in one batch -- deep U-Net levels then
// launch enough workgroups to fill 256 CUs (measured 328 vs 337 ms per song against batches of 8)
static int pick_batch(const asx_engine *e) { return e->cfg.max_batch > 0 ? e->cfg.max_batch : 64; }

// ----------------------------------------------------------------------------
// profiling wrapper
// ----------------------------------------------------------------------------
template <class F>
static int timed(asx_engine *e, int cls, double flops, double bytes, hipStream_t s, F &&launch) {
  if (!e->prof) {
    launch();
    HIPCHK(hipGetLastError());
    return ASX_OK;
  }
  ProfRec r;
  r.cls = cls;
  r.flops = flops;
  r.bytes = bytes;
  HIPCHK(hipEventCreate(&r.a));
  HIPCHK(hipEventCreate(&r.b));
  HIPCHK(hipEventRecord(r.a, s));
  launch();
  HIPCHK(hipGetLastError());
  HIPCHK(hipEventRecord(r.b, s));
  e->recs.push_back(r);
  return ASX_OK;
}

// ----------------------------------------------------------------------------
// FFT plan / tables
// ----------------------------------------------------------------------------
static bool make_plan(int n_fft, FftPlan *p) {
  if (n_fft < 8 || (n_fft & 1)) return false;
  p->n_fft = n_fft;
  p->nh = n_fft / 2;
  p->n_stage = 0;
  int n = p->nh;
  // large radices first: a 2048-point transform is 16 x 16 x 8 -- three barrier-separated LDS passes instead of six radix-4 / 2
  // ones (the generic kernels are bound by their passes, not by memory: profiles/NOTES.md).  ASX_FFT_RADIX4=1: the old plans.
  static const bool r4only = getenv("ASX_FFT_RADIX4") && atoi(getenv("ASX_FFT_RADIX4")) != 0;
  if (!r4only) {
    while (n % 16 == 0) {
      p->radix[p->n_stage++] = 16;
      n /= 16;
    }
    if (n % 8 == 0) {
      p->radix[p->n_stage++] = 8;
      n /= 8;
    }
  }
  while (n % 4 == 0) {
    p->radix[p->n_stage++] = 4;
    n /= 4;
  }
  const int primes[3] = {2, 3, 5};
  for (int q : primes)
    while (n % q == 0) {
      if (p->n_stage >= 16) return false;
      p->radix[p->n_stage++] = q;
      n /= q;
    }
  return n == 1;
}

// torch.hann_window(win_length) (periodic), zero padded to n at both ends like torch.stft does for win_length < n_fft
static void host_window(int n, std::vector<float> &w, int win_length = 0) {
  const int wl = (win_length > 0 && win_length < n) ? win_length : n;
  const int off = (n - wl) / 2;
  w.assign(n, 0.f);
  for (int k = 0; k < wl; ++k) w[off + k] = (float)(0.5 - 0.5 * cos(2.0 * M_PI * (double)k / (double)wl));
}

// sum of squared windows, accumulated in f32 in increasing frame order like torch.istft
static void host_env(int n, int hop, int T, std::vector<float> &env, int win_length = 0) {
  std::vector<float> w;
  host_window(n, w, win_length);
  env.assign((size_t)n + (size_t)hop * (T - 1), 0.f);
  for (int t = 0; t < T; ++t)
    for (int k = 0; k < n; ++k) env[(size_t)t * hop + k] += w[k] * w[k];
}

static size_t stft_lds(const FftPlan &p) { return (size_t)p.nh * 2 * sizeof(float2); }
static size_t istft_lds(const FftPlan &p) { return ((size_t)p.nh * 3 + 1) * sizeof(float2); }
static size_t ht_istft_lds(const FftPlan &p) { return ((size_t)p.nh * 2 + 1) * sizeof(float2); }   // ht_istft_kernel stages X in bufB

// ----------------------------------------------------------------------------
// kernel launchers
// ----------------------------------------------------------------------------
template <class CFG>
static void launch_conv_t(const ConvArgs &a, int nblk, hipStream_t s) {
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&conv_mfma_kernel<CFG>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, CFG::LDS_BYTES);
    attr_done = true;
  }
  hipLaunchKernelGGL(conv_mfma_kernel<CFG>, dim3(nblk), dim3(256), CFG::LDS_BYTES, s, a);
}

static int conv_tile_h(int kind) { return kind == CK_UP ? 4 : 8; }

static int pick_nrep_conv(int cout) {
  const int ct = (cout + 15) / 16;
  int best = 1, best_pad = 1 << 30;
  for (int n = 3; n >= 1; --n) {
    const int pad = ((ct + n - 1) / n) * n - ct;
    if (pad < best_pad) {
      best_pad = pad;
      best = n;
    }
  }
  return best;
}
static int pick_nrep_up(int cout) {
  const int ct = (cout + 15) / 16;  // virtual tiles = 4*ct, pairs must stay together
  if ((4 * ct) % 6 == 0) return 6;
  if ((4 * ct) % 4 == 0) return 4;
  return 2;
}

static int conv_setup(ConvLayer &L, int kind, int cin, int cout, int relu) {
  L.kind = kind;
  L.cin = cin;
  L.cout = cout;
  L.relu = relu;
  if (kind == CK_UP) {
    L.nrep = pick_nrep_up(cout);
    L.kc = 8;
    const int vt = 4 * ((cout + 15) / 16);
    L.cg = (vt + L.nrep - 1) / L.nrep;
  } else {
    L.nrep = pick_nrep_conv(cout);
    L.kc = (kind == CK_3X3) ? 8 : (kind == CK_DOWN ? 4 : (cin >= 16 ? 16 : 4));
    // 3x3 convs stage FOUR channels at a time (37 KB of LDS instead of 75 KB at eight): a third co-resident workgroup
    // per CU covers the prologue / epilogue / barrier gaps of the other two.  Measured on the bench configuration: 3x3 class
    // 234.9 -> 224.0 ms (85.1 -> 89.4 % of the fp32-MFMA peak), every level gains.  ASX_CONV_KC4=<cin threshold> for the
    // A/B (0 = eight channels everywhere).
    static const int kc4 = getenv("ASX_CONV_KC4") ? atoi(getenv("ASX_CONV_KC4")) : (1 << 30);
    // (also for the two-tile channel groups of the MDX23C widths: ASX_CONV_KC4_N2=0 keeps their eight-channel stages)
    static const int kc4_n2 = getenv("ASX_CONV_KC4_N2") ? atoi(getenv("ASX_CONV_KC4_N2")) : 1;
    if (kind == CK_3X3 && cin <= kc4 && cin % 4 == 0 && (pick_nrep_conv(cout) == 3 || (kc4_n2 && pick_nrep_conv(cout) == 2))) L.kc = 4;
    // 2x2 / stride-2 conv: two-channel stages (four workgroups per CU); ASX_DOWN_KC2=0 keeps the four-channel stages
    static const int down_kc2 = getenv("ASX_DOWN_KC2") ? atoi(getenv("ASX_DOWN_KC2")) : 1;
    if (kind == CK_DOWN && down_kc2 && cin % 2 == 0 && pick_nrep_conv(cout) == 3) L.kc = 2;
    L.cg = ((cout + 15) / 16 + L.nrep - 1) / L.nrep;
  }
  L.nci = (cin + L.kc - 1) / L.kc;
  return ASX_OK;
}

// w layouts: 3x3 [cout,cin,3,3]; down [cout,cin,2,2]; 1x1 [cout,cin]; up [cin,cout,2,2]
// packed: [CG][NCI][WSTAGE] with WSTAGE = roundup(ntap*kc*NWP, 256) floats; row (tap, kc) holds NWP floats
// wino_mode: the engine's "winograd" option at load time.  The image of the default kernel (3) is always built -- the option may be
// switched between 0 and 3 on a loaded net (bench.py's direct_kernel leg does) -- the images of the earlier generations (1, 2) only
// when the option selects them before the weights arrive.
static int conv_pack(ConvLayer &L, const float *w, const float *b, int wino_mode = 3) {
  const int ntap = L.kind == CK_3X3 ? 9 : (L.kind == CK_DOWN ? 4 : 1);
  const int NW = 16 * L.nrep;
  const int NWP = (L.nrep % 2 == 0) ? NW + 16 : NW;
  const size_t wstage = (((size_t)ntap * L.kc * NWP + 255) / 256) * 256;
  const size_t per_cg = (size_t)L.nci * wstage;
  std::vector<float> wp(per_cg * L.cg, 0.f);
  const int CT = (L.cout + 15) / 16;
  for (int cg = 0; cg < L.cg; ++cg)
    for (int ci = 0; ci < L.nci; ++ci)
      for (int tap = 0; tap < ntap; ++tap)
        for (int kc = 0; kc < L.kc; ++kc) {
          const int c = ci * L.kc + kc;
          if (c >= L.cin) continue;
          float *dst = &wp[cg * per_cg + ci * wstage + ((size_t)tap * L.kc + kc) * NWP];
          for (int n = 0; n < NW; ++n) {
            float v = 0.f;
            if (L.kind == CK_UP) {
              const int nt = cg * L.nrep + n / 16;
              const int pair = nt / 2, dx = nt & 1;
              const int dy = pair / CT, ct = pair % CT;
              const int co = ct * 16 + (n & 15);
              if (dy < 2 && co < L.cout) v = w[(((size_t)c * L.cout + co) * 2 + dy) * 2 + dx];
            } else {
              const int co = cg * NW + n;
              if (co < L.cout) v = w[((size_t)co * L.cin + c) * ntap + tap];
            }
            dst[n] = v;
          }
        }
  if (L.kind == CK_3X3) {
    // U = G g G^T in float64, laid out [cg48][ci8][xi][pair][cout 48][2]
    static const double G[4][3] = {{1, 0, 0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0, 0, 1}};
    L.wu_cg = (L.cout + 47) / 48;
    L.wu_nci = (L.cin + 7) / 8;
    L.wu3_nci = ((L.cin + 7) / 8) * 2;     // four-channel stage images, an even number of them (zero padded): the kernel's 8-channel build reads pairs
    std::vector<float> wu((size_t)L.wu_cg * L.wu_nci * WinoCfg::USTAGE, 0.f), wu2(wu.size(), 0.f);
    std::vector<float> wu3((size_t)L.wu_cg * L.wu3_nci * Wino3Cfg::USTAGE, 0.f);
    for (int co = 0; co < L.cout; ++co)
      for (int c = 0; c < L.cin; ++c) {
        const float *g = &w[((size_t)co * L.cin + c) * 9];
        double t[4][3], U[4][4];
        for (int a = 0; a < 4; ++a)
          for (int j = 0; j < 3; ++j) t[a][j] = G[a][0] * g[0 * 3 + j] + G[a][1] * g[1 * 3 + j] + G[a][2] * g[2 * 3 + j];
        for (int a = 0; a < 4; ++a)
          for (int bb = 0; bb < 4; ++bb) U[a][bb] = t[a][0] * G[bb][0] + t[a][1] * G[bb][1] + t[a][2] * G[bb][2];
        const int cgi = co / 48, col = co % 48, ci = c / 8, pair = (c % 8) / 2, e = c & 1;
        float *dst = &wu[((size_t)cgi * L.wu_nci + ci) * WinoCfg::USTAGE];
        float *dst2 = &wu2[((size_t)cgi * L.wu_nci + ci) * WinoCfg::USTAGE];
        float *dst3 = &wu3[((size_t)cgi * L.wu3_nci + c / 4) * Wino3Cfg::USTAGE + ((size_t)(c % 4) * 16 + col % 16) * Wino3Cfg::ULS];
        for (int a = 0; a < 4; ++a)
          for (int bb = 0; bb < 4; ++bb) {
            dst[(((a * 4 + bb) * 4 + pair) * 48 + col) * 2 + e] = (float)U[a][bb];
            dst2[((a * 4 + bb) * 8 + (c % 8)) * 48 + col] = (float)U[a][bb];
            dst3[(a * 4 + bb) * 3 + col / 16] = (float)U[a][bb];
          }
      }
    if (wino_mode == 1) {
      CHK(L.wu.ensure(wu.size() * 4));
      HIPCHK(hipMemcpy(L.wu.p, wu.data(), wu.size() * 4, hipMemcpyHostToDevice));
    }
    if (wino_mode == 2) {
      CHK(L.wu2.ensure(wu2.size() * 4));
      HIPCHK(hipMemcpy(L.wu2.p, wu2.data(), wu2.size() * 4, hipMemcpyHostToDevice));
    }
    CHK(L.wu3.ensure(wu3.size() * 4));
    HIPCHK(hipMemcpy(L.wu3.p, wu3.data(), wu3.size() * 4, hipMemcpyHostToDevice));
    // bf16 x 6 image (kernels_wino6.h) for every layer wide enough to be worth a 32-channel stage; which layers RUN on it is the
    // engine's "winograd_bf16x6" option at launch time
    L.wu6_nci = 0;
    if (L.cin >= 64) {
      std::vector<uint32_t> w6;
      int cg6 = 0, nci6 = 0;
      wino6_pack(w, L.cout, L.cin, w6, &cg6, &nci6);
      if (cg6 == L.wu_cg) {
        CHK(L.wu6.ensure(w6.size() * 4));
        HIPCHK(hipMemcpy(L.wu6.p, w6.data(), w6.size() * 4, hipMemcpyHostToDevice));
        L.wu6_nci = nci6;
      }
    }
    // weight-stationary image: only where all of U fits the registers of eight waves (Cin <= 96) without much zero padding
    L.wus_ks = (L.cin > 40 && L.cin <= 48) ? 12 : ((L.cin > 88 && L.cin <= 96) ? 24 : 0);
    if (L.wus_ks) {
      const size_t per = (size_t)8 * (6 * L.wus_ks / 4) * 64 * 4;
      std::vector<float> wus((size_t)L.wu_cg * per, 0.f);
      for (int co = 0; co < L.cout; ++co)
        for (int c = 0; c < L.cin; ++c) {
          const float *g = &w[((size_t)co * L.cin + c) * 9];
          double t[4][3];
          for (int a = 0; a < 4; ++a)
            for (int j = 0; j < 3; ++j) t[a][j] = G[a][0] * g[0 * 3 + j] + G[a][1] * g[1 * 3 + j] + G[a][2] * g[2 * 3 + j];
          float *dst = &wus[(size_t)(co / 48) * per];
          for (int a = 0; a < 4; ++a)
            for (int bb = 0; bb < 4; ++bb) {
              const float u = (float)(t[a][0] * G[bb][0] + t[a][1] * G[bb][1] + t[a][2] * G[bb][2]);
              dst[L.wus_ks == 12 ? winos_u_index<12>(a, bb, c, co % 48) : winos_u_index<24>(a, bb, c, co % 48)] = u;
            }
        }
      CHK(L.wus.ensure(wus.size() * 4));
      HIPCHK(hipMemcpy(L.wus.p, wus.data(), wus.size() * 4, hipMemcpyHostToDevice));
    }
  }
  const int nb = (L.kind == CK_UP) ? CT * 16 : std::max(L.cg * NW, ((L.cout + 47) / 48) * 48);
  std::vector<float> bp(nb, 0.f);
  for (int i = 0; i < L.cout; ++i) bp[i] = b ? b[i] : 0.f;
  CHK(L.w.ensure(wp.size() * 4));
  CHK(L.b.ensure(bp.size() * 4));
  HIPCHK(hipMemcpy(L.w.p, wp.data(), wp.size() * 4, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(L.b.p, bp.data(), bp.size() * 4, hipMemcpyHostToDevice));
  return ASX_OK;
}

template <class CFG>
static void launch_conv_dma_t(const ConvArgs &a, int nblk, hipStream_t s) {
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&conv_dma_kernel<CFG>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, CFG::LDS_BYTES);
    attr_done = true;
  }
  hipLaunchKernelGGL(conv_dma_kernel<CFG>, dim3(nblk), dim3(256), CFG::LDS_BYTES, s, a);
}

static std::atomic<long long> g_wino6_launches{0};   // launches of conv_wino6_kernel (kernels_wino6.h) since the process started

// Optional view description of a conv's operands (channel slices of larger buffers).
struct ConvView {
  int64_t x_bstride = 0;    // 0 = dense [B, cin, T, F]
  int64_t y_bstride = 0;    // 0 = dense [B, cout, To, Fo]
  const float *res = nullptr;
  int64_t aux_bstride = 0;  // of skip / res; 0 = dense
  int act = -1;             // -1 = the layer's own activation
};

// x [B,cin,T,F] -> y
static int conv_launch(asx_engine *e, const ConvLayer &L, const float *x, const float *skip, float *y, int B, int T,
                       int F, hipStream_t s, const ConvView &v = ConvView()) {
  ConvArgs a{};
  a.x = x;
  a.wp = L.w.f();
  a.bias = L.b.f();
  a.skip = skip;
  a.res = v.res;
  a.zeros = e->d_zeros.f();
  a.y = y;
  a.B = B;
  a.Cin = L.cin;
  a.Cout = L.cout;
  a.T = T;
  a.F = F;
  a.act = v.act >= 0 ? v.act : (L.relu ? ACT_RELU : ACT_NONE);
  a.CG = L.cg;
  a.NCI = L.nci;
  static const int nt_mode = getenv("ASX_NT") ? atoi(getenv("ASX_NT")) : 0;   // bit 0: conv stores, bit 1: TDF stores, bit 2: TDF residual loads
  a.nt = nt_mode & 1;
  int cls = ASX_PROF_CONV3X3;
  double taps = 9;
  int64_t out_plane;
  if (L.kind == CK_DOWN) {
    a.To = T / 2;
    a.Fo = F / 2;
    cls = ASX_PROF_DOWN;
    taps = 4;
    out_plane = (int64_t)a.To * a.Fo;
  } else {
    a.To = T;
    a.Fo = F;
    out_plane = (int64_t)T * F;
    if (L.kind == CK_1X1) {
      cls = ASX_PROF_CONV1X1;
      taps = 1;
    } else if (L.kind == CK_UP) {
      cls = ASX_PROF_UP;
      taps = 4;
      out_plane = (int64_t)4 * T * F;
    }
  }
  a.x_bstride = v.x_bstride ? v.x_bstride : (int64_t)L.cin * T * F;
  a.y_bstride = v.y_bstride ? v.y_bstride : (int64_t)L.cout * out_plane;
  a.aux_bstride = v.aux_bstride ? v.aux_bstride : (int64_t)L.cout * out_plane;
  const int th = conv_tile_h(L.kind);
  a.tilesT = (a.To + th - 1) / th;
  a.tilesF = (a.Fo + 63) / 64;
  const int nblk = a.CG * a.tilesT * a.tilesF * B;
  if (nblk <= 0) return ASX_OK;
  const double outpix = (double)B * a.To * a.Fo;
  const double flops = 2.0 * taps * L.cin * L.cout * outpix;
  double bytes = 4.0 * ((double)B * L.cin * T * F + (double)L.cout * outpix * (L.kind == CK_UP ? 4 : 1));
  if (L.kind == CK_UP && skip) bytes += 4.0 * (double)L.cout * outpix * 4;  // skip read
  if (v.res) bytes += 4.0 * (double)L.cout * outpix;
  int bad = 0;
  const bool dma = (F % 4 == 0) && ((reinterpret_cast<uintptr_t>(x) & 15) == 0) && (a.x_bstride % 4 == 0) &&
                   getenv("ASX_NO_DMA") == nullptr;
  const ConvArgs &d = a;
  if (L.kind == CK_3X3 && e->winograd == 3 && e->winos == 1 && dma && L.wus.p != nullptr && a.Fo % 32 == 0 && v.res == nullptr &&
      (a.act == ACT_RELU || a.act == ACT_NONE) && (int64_t)L.cin * T * F < ((int64_t)1 << 30)) {
    // weight-stationary Winograd (kernels_winos.h): a workgroup walks a 32-pixel-wide column strip, one tile row per step
    ConvArgs wa = a;
    wa.wp = L.wus.f();
    wa.CG = L.wu_cg;
    wa.tilesF = a.Fo / 32;
    const int NS = (a.To + 1) / 2;
    const int base = wa.CG * wa.tilesF * B;
    // row blocks: enough workgroups to fill the chip a few times over, but at least 8 tile rows each (prologue: weights + 4 rows)
    int RB = std::max(1, std::min((NS + 7) / 8, (2048 + base - 1) / base));
    const int SPB = (NS + RB - 1) / RB;
    RB = (NS + SPB - 1) / SPB;
    wa.tilesT = RB;
    wa.NCI = SPB;
    const int nb = base * RB;
    auto gos = [&](auto kern, int lds) {
      {
        static std::mutex attr_mutex;
        static std::set<const void *> attr_done;
        std::lock_guard<std::mutex> lock(attr_mutex);
        if (attr_done.insert(reinterpret_cast<const void *>(kern)).second)
          (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
      }
      return timed(e, cls, flops, bytes, s, [&]() { hipLaunchKernelGGL(kern, dim3(nb), dim3(512), lds, s, wa); });
    };
    static const int abls = getenv("ASX_WINOS_ABL") ? atoi(getenv("ASX_WINOS_ABL")) : 0;   // timing probes (results invalid)
    const bool k12 = L.wus_ks == 12;
    const bool ragged = (a.To & 1) || (L.cout % 48) != 0;
    if (abls && !ragged) {
      switch (abls) {
        case 1: return k12 ? gos(&conv_winos_kernel<12, 1>, WinoSCfg<12>::LDS_BYTES) : gos(&conv_winos_kernel<24, 1>, WinoSCfg<24>::LDS_BYTES);
        case 2: return k12 ? gos(&conv_winos_kernel<12, 2>, WinoSCfg<12>::LDS_BYTES) : gos(&conv_winos_kernel<24, 2>, WinoSCfg<24>::LDS_BYTES);
        case 4: return k12 ? gos(&conv_winos_kernel<12, 4>, WinoSCfg<12>::LDS_BYTES) : gos(&conv_winos_kernel<24, 4>, WinoSCfg<24>::LDS_BYTES);
        case 5: return k12 ? gos(&conv_winos_kernel<12, 5>, WinoSCfg<12>::LDS_BYTES) : gos(&conv_winos_kernel<24, 5>, WinoSCfg<24>::LDS_BYTES);
        case 13: return k12 ? gos(&conv_winos_kernel<12, 13>, WinoSCfg<12>::LDS_BYTES) : gos(&conv_winos_kernel<24, 13>, WinoSCfg<24>::LDS_BYTES);
        default: break;
      }
    }
    if (ragged) return k12 ? gos(&conv_winos_kernel<12, 0, true>, WinoSCfg<12>::LDS_BYTES) : gos(&conv_winos_kernel<24, 0, true>, WinoSCfg<24>::LDS_BYTES);
    return k12 ? gos(&conv_winos_kernel<12, 0, false>, WinoSCfg<12>::LDS_BYTES) : gos(&conv_winos_kernel<24, 0, false>, WinoSCfg<24>::LDS_BYTES);
  }
  if (L.kind == CK_3X3 && e->winograd == 3 && e->gemm_bf16x6 > 0 && e->wino6 > 0 && L.cin >= e->wino6 && dma && L.wu6_nci > 0 &&
      (int64_t)T * F < ((int64_t)1 << 24)) {
    // Winograd on the bf16 pipe, one workgroup per (8 x 32 tile, 48-channel group); the groups of a tile are consecutive on one XCD
    ConvArgs wa = a;
    wa.wp = reinterpret_cast<const float *>(L.wu6.p);
    wa.CG = L.wu_cg;
    wa.NCI = L.wu6_nci;
    wa.tilesT = (a.To + Wino6Cfg::TH - 1) / Wino6Cfg::TH;
    wa.tilesF = (a.Fo + Wino6Cfg::TW - 1) / Wino6Cfg::TW;
    const int64_t S = (int64_t)wa.tilesT * wa.tilesF * B;
    const int64_t nb = ((S + 7) / 8) * 8 * wa.CG;
    if (nb < ((int64_t)1 << 31)) {
      {
        static std::mutex attr_mutex;
        static bool attr_done = false;
        std::lock_guard<std::mutex> lock(attr_mutex);
        if (!attr_done) {
          (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&conv_wino6_kernel<0, 1>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    Wino6Cfg::LDS_BYTES);
          attr_done = true;
        }
      }
      g_wino6_launches.fetch_add(1);
      return timed(e, cls, flops, bytes, s, [&]() {
        hipLaunchKernelGGL((conv_wino6_kernel<0, 1>), dim3((unsigned)nb), dim3(512), Wino6Cfg::LDS_BYTES, s, wa);
      });
    }
  }
  if (L.kind == CK_3X3 && e->winograd == 3 && dma && L.wu3.p != nullptr) {
    ConvArgs wa = a;
    wa.wp = L.wu3.f();
    wa.CG = L.wu_cg;
    wa.NCI = L.wu3_nci;
    wa.tilesT = (a.To + Wino3Cfg::TH - 1) / Wino3Cfg::TH;
    wa.tilesF = (a.Fo + Wino3Cfg::TW - 1) / Wino3Cfg::TW;
    const int nb = wa.CG * wa.tilesT * wa.tilesF * B;
    static const int abl3 = getenv("ASX_WINO_ABL") ? atoi(getenv("ASX_WINO_ABL")) : 0;   // timing probes (results invalid)
    // A/B builds: 6 (default): 4-channel stages x 2 LDS buffers, raw planes at an odd float stride; 5 / 4: rings of 3 / 4 buffers
    // (two / three stages of DMA in flight, counted vmcnt); 0 / 2: 4 / 3 buffers at the even stride; 1: 8-channel stages x 2 buffers
    static const int wcfg = getenv("ASX_WINO_CFG") ? atoi(getenv("ASX_WINO_CFG")) : 6;
    auto go = [&](auto kern, int lds, int stages) {
      {
        static std::mutex attr_mutex;            // engines may be driven from several host threads (one per bag member / rank)
        static std::set<const void *> attr_done;
        std::lock_guard<std::mutex> lock(attr_mutex);
        if (attr_done.insert(reinterpret_cast<const void *>(kern)).second)
          (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
      }
      wa.NCI = stages;
      return timed(e, cls, flops, bytes, s, [&]() { hipLaunchKernelGGL(kern, dim3(nb), dim3(256), lds, s, wa); });
    };
    if (abl3) {
      switch (abl3) {
        case 1: return go(&conv_wino3_kernel<1>, Wino3Cfg::LDS_BYTES, L.wu3_nci);
        case 2: return go(&conv_wino3_kernel<2>, Wino3Cfg::LDS_BYTES, L.wu3_nci);
        case 4: return go(&conv_wino3_kernel<4>, Wino3Cfg::LDS_BYTES, L.wu3_nci);
        case 16: return go(&conv_wino3_kernel<16>, Wino3Cfg::LDS_BYTES, L.wu3_nci);
        default: break;
      }
    }
    if (wcfg == 1) return go(&conv_wino3_kernel<0, 8, 2>, Wino3CfgT<8, 2>::LDS_BYTES, L.wu3_nci / 2);
    if (wcfg == 2) return go(&conv_wino3_kernel<0, 4, 3>, Wino3CfgT<4, 3>::LDS_BYTES, L.wu3_nci);
    if (wcfg == 0) return go(&conv_wino3_kernel<0>, Wino3Cfg::LDS_BYTES, L.wu3_nci);
    if (wcfg == 4) return go(&conv_wino3_kernel<0, 4, 4, 1>, Wino3CfgT<4, 4, 1>::LDS_BYTES, L.wu3_nci);
    if (wcfg == 5) return go(&conv_wino3_kernel<0, 4, 3, 1>, Wino3CfgT<4, 3, 1>::LDS_BYTES, L.wu3_nci);
    return go(&conv_wino3_kernel<0, 4, 2, 1>, Wino3CfgT<4, 2, 1>::LDS_BYTES, L.wu3_nci);
  }
  if (L.kind == CK_3X3 && e->winograd >= 2 && dma && L.wu2.p != nullptr) {
    ConvArgs wa = a;
    wa.wp = L.wu2.f();
    wa.CG = L.wu_cg;
    wa.NCI = L.wu_nci;
    wa.tilesT = (a.To + 7) / 8;
    wa.tilesF = (a.Fo + 31) / 32;
    const int nb = wa.CG * wa.tilesT * wa.tilesF * B;
    static bool attr_done = false;
    if (!attr_done) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&conv_wino2_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                Wino2Cfg<4>::LDS_BYTES);
      attr_done = true;
    }
    return timed(e, cls, flops, bytes, s, [&]() {
      hipLaunchKernelGGL(conv_wino2_kernel<4>, dim3(nb), dim3(256), Wino2Cfg<4>::LDS_BYTES, s, wa);
    });
  }
  if (L.kind == CK_3X3 && e->winograd == 1 && dma && L.wu.p != nullptr) {
    ConvArgs wa = a;
    wa.wp = L.wu.f();
    wa.CG = L.wu_cg;
    wa.NCI = L.wu_nci;
    wa.tilesT = (a.To + WinoCfg::TH - 1) / WinoCfg::TH;
    wa.tilesF = (a.Fo + WinoCfg::TW - 1) / WinoCfg::TW;
    const int nb = wa.CG * wa.tilesT * wa.tilesF * B;
    static bool attr_done = false;
    if (!attr_done) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&conv_wino_kernel),
                                hipFuncAttributeMaxDynamicSharedMemorySize, WinoCfg::LDS_BYTES);
      attr_done = true;
    }
    return timed(e, cls, flops, bytes, s, [&]() {
      hipLaunchKernelGGL(conv_wino_kernel, dim3(nb), dim3(256), WinoCfg::LDS_BYTES, s, wa);
    });
  }
#define ASX_CONV_CASE(KH, KW, S, PAD, NR, KC, RPW, EPI)                                      \
  do {                                                                                       \
    if (dma) launch_conv_dma_t<ConvDmaCfg<KH, KW, S, PAD, NR, KC, RPW, EPI>>(d, nblk, s);    \
    else launch_conv_t<ConvCfg<KH, KW, S, PAD, NR, KC, RPW, EPI>>(a, nblk, s);               \
  } while (0)
  CHK(timed(e, cls, flops, bytes, s, [&]() {
    switch (L.kind) {
      case CK_3X3:
        if (L.nrep == 3 && L.kc == 4) ASX_CONV_CASE(3, 3, 1, 1, 3, 4, 2, EPI_BIAS_ACT);
        else if (L.nrep == 3) ASX_CONV_CASE(3, 3, 1, 1, 3, 8, 2, EPI_BIAS_ACT);
        else if (L.nrep == 2 && L.kc == 4) ASX_CONV_CASE(3, 3, 1, 1, 2, 4, 2, EPI_BIAS_ACT);
        else if (L.nrep == 2) ASX_CONV_CASE(3, 3, 1, 1, 2, 8, 2, EPI_BIAS_ACT);
        else ASX_CONV_CASE(3, 3, 1, 1, 1, 8, 2, EPI_BIAS_ACT);
        break;
      case CK_DOWN:
        if (L.nrep == 3 && L.kc == 2) ASX_CONV_CASE(2, 2, 2, 0, 3, 2, 2, EPI_BIAS_ACT);
        else if (L.nrep == 3) ASX_CONV_CASE(2, 2, 2, 0, 3, 4, 2, EPI_BIAS_ACT);
        else if (L.nrep == 2) ASX_CONV_CASE(2, 2, 2, 0, 2, 4, 2, EPI_BIAS_ACT);
        else ASX_CONV_CASE(2, 2, 2, 0, 1, 4, 2, EPI_BIAS_ACT);
        break;
      case CK_1X1:
        if (L.kc == 16) {
          if (L.nrep == 3) ASX_CONV_CASE(1, 1, 1, 0, 3, 16, 2, EPI_BIAS_ACT);
          else if (L.nrep == 2) ASX_CONV_CASE(1, 1, 1, 0, 2, 16, 2, EPI_BIAS_ACT);
          else ASX_CONV_CASE(1, 1, 1, 0, 1, 16, 2, EPI_BIAS_ACT);
        } else {
          if (L.nrep == 3) ASX_CONV_CASE(1, 1, 1, 0, 3, 4, 2, EPI_BIAS_ACT);
          else if (L.nrep == 2) ASX_CONV_CASE(1, 1, 1, 0, 2, 4, 2, EPI_BIAS_ACT);
          else ASX_CONV_CASE(1, 1, 1, 0, 1, 4, 2, EPI_BIAS_ACT);
        }
        break;
      case CK_UP:
        if (L.nrep == 6) ASX_CONV_CASE(1, 1, 1, 0, 6, 8, 1, EPI_UP_MULSKIP);
        else if (L.nrep == 4) ASX_CONV_CASE(1, 1, 1, 0, 4, 8, 1, EPI_UP_MULSKIP);
        else ASX_CONV_CASE(1, 1, 1, 0, 2, 8, 1, EPI_UP_MULSKIP);
        break;
      default: bad = 1;
    }
  }));
#undef ASX_CONV_CASE
  if (bad) {
    set_err("conv_launch: bad kind");
    return ASX_ERR_INVALID;
  }
  return ASX_OK;
}

template <int NREP, int MREP, bool KVEC>
static void launch_tdf_tt(const TdfArgs &a, hipStream_t s) {
  using CFG = TdfCfg<NREP, MREP>;
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&tdf_mfma_kernel<NREP, MREP, KVEC>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, CFG::LDS_BYTES);
    attr_done = true;
  }
  const int64_t nbm = (a.M + CFG::BM - 1) / CFG::BM;
  const int nbn = (a.N + CFG::BN - 1) / CFG::BN;
  hipLaunchKernelGGL((tdf_mfma_kernel<NREP, MREP, KVEC>), dim3((unsigned)(nbm * nbn)), dim3(256), CFG::LDS_BYTES, s,
                     a);
}
template <int NREP, int MREP>
static void launch_tdf_t(const TdfArgs &a, hipStream_t s) {
  if ((a.K & 3) == 0 && a.K >= 4) launch_tdf_tt<NREP, MREP, true>(a, s);
  else launch_tdf_tt<NREP, MREP, false>(a, s);
}

template <int NREP, int MREP, int BK>
static void launch_tdf_dma_tt(const TdfDmaArgs &a, hipStream_t s) {
  using CFG = TdfDmaCfg<NREP, MREP, BK>;
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&tdf_dma_kernel<NREP, MREP, BK>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, CFG::LDS_BYTES);
    attr_done = true;
  }
  const int64_t nbm = (a.M + CFG::BM - 1) / CFG::BM;
  const int nbn = (a.N + CFG::BN - 1) / CFG::BN;
  hipLaunchKernelGGL((tdf_dma_kernel<NREP, MREP, BK>), dim3((unsigned)(nbm * nbn)), dim3(256), CFG::LDS_BYTES, s, a);
}
template <int NREP, int MREP>
static void launch_tdf_dma_t(const TdfDmaArgs &a, hipStream_t s) {
  static const bool bk64 = getenv("ASX_GEMM_BK64") != nullptr;
  if (bk64 && NREP == 3 && MREP == 8 && a.K >= 128) launch_tdf_dma_tt<3, 8, 64>(a, s);
  else launch_tdf_dma_tt<NREP, MREP, 32>(a, s);
}

// tile choice for the LDS-DMA row GEMM: 128 x 192 or 128 x 128 for wide outputs, by a two-term cost model -- padding of
// the last column tile and wave quantisation over the 512 workgroup slots (2 per CU); the narrower tile moves 20 % more
// bytes per flop, charged as 4 %.  Measured: N = 512 out-proj / FF2 of BS-Roformer 103 -> 111 TFLOP/s (a third 192-wide tile
// would be 1/3 empty); HTDemucs transformer linears (43 k rows, N = 384 .. 1536) 73 -> 92 TFLOP/s.
// ASX_GEMM_T128=0 forces 128 x 192, =2 forces 128 x 128 (tuning aid).
// second-generation row GEMM (kernels_gemm2.h).  ASX_TDF2: 0 = tdf_dma_kernel only, 1 = tdf2 one tile per workgroup,
// 2 = + persistent over the column tiles of a row tile on short-K layers, 3 = + start stagger (ASX_TDF2_SBIT: block-id bit).
static int tdf2_mode() {
  static const int m = getenv("ASX_TDF2") ? atoi(getenv("ASX_TDF2")) : ASX_TDF2_DEFAULT;
  return m;
}
static bool tdf2_ok(const TdfDmaArgs &d) {
  auto a16 = [](const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
  const int64_t lda = d.lda ? d.lda : d.K, ldy = d.ldy ? d.ldy : d.N, ldr = d.ldr ? d.ldr : d.N;
  if (d.relu == 1 && (d.rscale != nullptr || d.rot_tab != nullptr)) return false;   // its ReLU ring epilogue applies neither
  return tdf2_mode() > 0 && d.K % 32 == 0 && d.K >= 32 && d.M % 8 == 0 && d.M >= 8 && d.M < (1ll << 31) && d.T > 0 && d.C > 0 && d.N % 8 == 0 && d.N >= 8 && lda % 4 == 0 &&
         ldy % 4 == 0 && ldr % 4 == 0 && a16(d.x) && a16(d.w) && a16(d.y) && (!d.res || a16(d.res)) && (!d.bias || a16(d.bias)) &&
         (uint64_t)8 * (uint64_t)lda * 4 < (1ull << 31) && (uint64_t)8 * (uint64_t)d.K * 4 < (1ull << 31) &&
         (uint64_t)16 * (uint64_t)ldy * 4 < (1ull << 31) && (uint64_t)16 * (uint64_t)ldr * 4 < (1ull << 31);
}
template <int NREP, int MREP, int ABL, int BK = 32>
static void launch_tdf2_abl(const TdfDmaArgs &a, hipStream_t s) {
  constexpr int BM = 16 * MREP, BN = 64 * NREP, LDS_BYTES = 2 * (BM + BN) * BK * 4;
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&tdf2_kernel<NREP, MREP, ABL, BK>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    attr_done = true;
  }
  static const int sbit = getenv("ASX_TDF2_SBIT") ? atoi(getenv("ASX_TDF2_SBIT")) : 8;
  const int64_t nbm = (a.M + BM - 1) / BM;
  const int nbn = (a.N + BN - 1) / BN;
  const int mode = tdf2_mode();
  // persistent over the column tiles when the K loop is short (prologue / epilogue are a visible share of a tile) and the
  // row tiles alone fill the 512 workgroup slots several times over
  const bool persist = mode >= 2 && nbn >= 2 && a.K <= 768 && nbm >= 2048;
  const int tiles = persist ? nbn : 1;
  hipLaunchKernelGGL((tdf2_kernel<NREP, MREP, ABL, BK>), dim3((unsigned)(nbm * (nbn / tiles))), dim3(256), LDS_BYTES, s, a, tiles,
                     (mode >= 3 && persist) ? sbit : -1);
}
template <int NREP, int MREP>
static void launch_tdf2(const TdfDmaArgs &a, hipStream_t s) {
  static const int abl = getenv("ASX_TDF2_ABL") ? atoi(getenv("ASX_TDF2_ABL")) : 0;   // ablation builds exist for the 128 x 192 tile only
  if constexpr (NREP == 3 && MREP == 8) {
    switch (abl) {
      case 1: return launch_tdf2_abl<3, 8, 1>(a, s);
      case 2: return launch_tdf2_abl<3, 8, 2>(a, s);
      case 3: return launch_tdf2_abl<3, 8, 3>(a, s);
      case 4: return launch_tdf2_abl<3, 8, 4>(a, s);
      case 5: return launch_tdf2_abl<3, 8, 5>(a, s);
      case 7: return launch_tdf2_abl<3, 8, 7>(a, s);
      case 8: return launch_tdf2_abl<3, 8, 8>(a, s);
      case 16: return launch_tdf2_abl<3, 8, 16>(a, s);
      default: break;
    }
  }
  // ASX_TDF2_BK16: 16-float stages (40 KB of LDS, launch bound 3) -- 1: on the 128 x 192 tile, 2: on a 64 x 192 tile
  static const int bk16 = getenv("ASX_TDF2_BK16") ? atoi(getenv("ASX_TDF2_BK16")) : 0;
  if constexpr (NREP == 3 && MREP == 8) {
    if (a.M % 16 == 0 && a.N % 16 == 0 && a.relu == 1 && a.K % 16 == 0) {
      if (bk16 == 1) return launch_tdf2_abl<3, 8, 0, 16>(a, s);
      if (bk16 == 2) return launch_tdf2_abl<3, 4, 0, 16>(a, s);
    }
  }
  launch_tdf2_abl<NREP, MREP, 0>(a, s);
}

// ---- third-generation row GEMM (kernels_gemm3.h): fp32 results from six bf16 MFMA products on exactly split operands ----------
// Per-engine switch (asx_engine::gemm_bf16x6): ASX_GEMM_BF16X6 (default 1) or asx_set_option(e, "gemm_bf16x6", n); 0 = the fp32-MFMA kernels only.
// launches of tdf3_kernel since the process started (tests assert that the path under test is the one that ran)
static std::atomic<long long> g_tdf3_launches{0};
static std::atomic<long long> g_attn6_launches{0};   // launches of attention6_kernel (kernels_rof.h)

// The split image of a weight matrix is built on first use and cached PER ENGINE by (pointer, N, K, cin) (asx_engine::w3).  Every
// entry point that uploads or frees weights of an engine flushes that engine's images (w3_flush) -- an address reused by another tensor
// of the same shape can therefore never meet a stale image, and no other engine's load / destroy touches them (round 5: the cache
// was process-wide, so a second engine's commit freed images a captured hipGraph of the first still pointed at).
static void w3_flush(asx_engine *e) {
  std::lock_guard<std::mutex> lk(e->w3_mu);
  for (auto &en : e->w3) (void)hipFree(en.img);
  e->w3.clear();
}

// drop the images of ONE weight buffer (a temporary layer of the single-op test hooks, about to be freed)
static void w3_drop(asx_engine *e, const void *w) {
  std::lock_guard<std::mutex> lk(e->w3_mu);
  for (size_t i = 0; i < e->w3.size();)
    if (e->w3[i].w == w) {
      (void)hipFree(e->w3[i].img);
      e->w3.erase(e->w3.begin() + (long)i);
    } else {
      ++i;
    }
}

static const u32x4 *w3_image(asx_engine *e, const float *w, int N, int K, hipStream_t s, int cin = 0) {
  std::lock_guard<std::mutex> lk(e->w3_mu);
  for (auto &en : e->w3)
    if (en.w == w && en.N == N && en.K == K && en.cin == cin) return reinterpret_cast<const u32x4 *>(en.img);
  const int nst = cin > 0 ? (K / cin) * ((cin + 31) / 32) : 0;
  const int ntiles = (N + 15) / 16, nk = cin > 0 ? ((nst + 1) & ~1) : ((K + 63) / 64) * 2;   // an even number of 32-wide stages (zero padded)
  W3Entry en{w, N, K, cin, nullptr};
  if (hipMalloc(&en.img, (size_t)ntiles * nk * 3 * 1024) != hipSuccess) return nullptr;
  const int64_t total = (int64_t)ntiles * nk * 64;
  hipLaunchKernelGGL(w3_split_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, w, reinterpret_cast<u32x4 *>(en.img), N, K,
                     total, cin, nst);
  // built once per weight tensor (the first forward after a load): make the image visible to every stream before it is published
  if (hipStreamSynchronize(s) != hipSuccess) {
    (void)hipFree(en.img);
    return nullptr;
  }
  e->w3.push_back(en);
  return reinterpret_cast<const u32x4 *>(en.img);
}

static bool tdf3_ok(const asx_engine *e, const TdfDmaArgs &d) {
  auto a16 = [](const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
  const int64_t lda = d.lda ? d.lda : d.K, ldy = d.ldy ? d.ldy : d.N, ldr = d.ldr ? d.ldr : d.N;
  if (d.relu == 1 && (d.rscale != nullptr || d.rot_tab != nullptr)) return false;   // as tdf2_ok: the ReLU ring epilogue applies neither
  return e->gemm_bf16x6 > 0 && d.K % 32 == 0 && d.K >= 64 && d.M >= 1 && d.M < (1ll << 31) && d.T > 0 && d.C > 0 && d.N % 8 == 0 &&
         d.N > 64 && lda % 4 == 0 && ldy % 4 == 0 && ldr % 4 == 0 && a16(d.x) && a16(d.w) && a16(d.y) && (!d.res || a16(d.res)) &&
         (!d.bias || a16(d.bias)) && (uint64_t)16 * (uint64_t)ldy * 4 < (1ull << 31) && (uint64_t)16 * (uint64_t)ldr * 4 < (1ull << 31);
}
template <int NREP, int MREP, int ABL>
static void launch_tdf3_abl(const TdfDmaArgs &a0, const u32x4 *w3, hipStream_t s) {
  constexpr int BM = 16 * MREP, BN = 64 * NREP, LDS_BYTES = 2 * 3 * BM * 64;
  TdfDmaArgs a = a0;
  const int64_t nbm = (a.M + BM - 1) / BM;
  const int nbn = (a.N + BN - 1) / BN;
  // tile -> XCD map (kernels_gemm3.h); ASX_TDF3_MAP = "<narrow><wide>" digits for N < 8 tiles / N >= 8 tiles (A/B switch)
  // Default (round 5, profiles/r05_tdf3_tile_map_ab.txt + r05_pmc_tdf3.json): fewer than 8 column tiles -> map 1 (the column tiles of a
  // row block are neighbours on one XCD, so x crosses the fabric once instead of twice: level-0 first TDF linear 6.86 -> 6.69 ms,
  // fabric traffic 2.0x -> ~1.0x algorithmic); 8 or more -> map 0 (column tiles partitioned over the XCDs; maps 1 / 2 measure the same).
  static const int map_env = getenv("ASX_TDF3_MAP") ? atoi(getenv("ASX_TDF3_MAP")) : -1;
  a.tile_map = map_env >= 0 ? (nbn >= 8 ? map_env % 10 : map_env / 10) : (nbn < 8 ? 1 : 0);
  hipLaunchKernelGGL((tdf3_kernel<NREP, MREP, ABL>), dim3((unsigned)(nbm * nbn)), dim3(256), LDS_BYTES, s, a, w3, RowGather{});
  g_tdf3_launches.fetch_add(1);
}
// GATHER mode (kernels_gemm3.h): stride-1 convolutions of the channels-last nets as implicit GEMMs on the same kernel
static std::atomic<long long> g_tdf3_gather_launches{0};
template <int NREP, int MREP>
static bool launch_tdf3_gather(asx_engine *e, const TdfDmaArgs &a, const RowGather &gq, hipStream_t s) {
  const u32x4 *w3 = w3_image(e, a.w, a.N, a.K, s, (gq.cin & 31) ? gq.cin : 0);   // channel counts off the 32-grid get their own padded image
  if (!w3) return false;
  constexpr int BM = 16 * MREP, BN = 64 * NREP, LDS_BYTES = 2 * 3 * BM * 64;
  const int64_t nbm = (a.M + BM - 1) / BM;
  const int nbn = (a.N + BN - 1) / BN;
  hipLaunchKernelGGL((tdf3_kernel<NREP, MREP, 0, true>), dim3((unsigned)(nbm * nbn)), dim3(256), LDS_BYTES, s, a, w3, gq);
  g_tdf3_gather_launches.fetch_add(1);
  return true;
}
static bool launch_tdf3_gather_auto(asx_engine *e, const TdfDmaArgs &d, const RowGather &gq, hipStream_t s) {
  if (d.glu_cout > 0) {                                // value / gate fragment pairs: 128-column tiles (two fragments per wave)
    const bool small = (d.M + 127) / 128 * ((d.N + 127) / 128) < 1024;   // grid too small for 128-row tiles to fill the chip twice
    return small ? launch_tdf3_gather<2, 4>(e, d, gq, s) : launch_tdf3_gather<2, 8>(e, d, gq, s);
  }
  if (d.N <= 64) return launch_tdf3_gather<1, 8>(e, d, gq, s);   // narrow layers (48 .. 64 columns): one fragment per wave
  if (d.N > 128) {
    const double rows = (double)((d.M + 127) / 128);
    auto cost = [&](int bn, double eff) { return ceil(rows * (double)((d.N + bn - 1) / bn) / 512.0) * bn / eff; };
    return cost(128, 0.96) < cost(192, 1.0) ? launch_tdf3_gather<2, 8>(e, d, gq, s) : launch_tdf3_gather<3, 8>(e, d, gq, s);
  }
  return launch_tdf3_gather<2, 4>(e, d, gq, s);
}
template <int NREP, int MREP>
static bool launch_tdf3(asx_engine *e, const TdfDmaArgs &a, hipStream_t s) {
  const u32x4 *w3 = w3_image(e, a.w, a.N, a.K, s);
  if (!w3) return false;                               // out of memory for the image: the caller falls back to the fp32 kernels
  static const int abl = getenv("ASX_TDF3_ABL") ? atoi(getenv("ASX_TDF3_ABL")) : 0;   // ablation builds exist for the 128 x 192 tile only
  if constexpr (NREP == 3 && MREP == 8) {
    switch (abl) {
      case 1: launch_tdf3_abl<3, 8, 1>(a, w3, s); return true;
      case 2: launch_tdf3_abl<3, 8, 2>(a, w3, s); return true;
      case 4: launch_tdf3_abl<3, 8, 4>(a, w3, s); return true;
      case 8: launch_tdf3_abl<3, 8, 8>(a, w3, s); return true;
      case 13: launch_tdf3_abl<3, 8, 13>(a, w3, s); return true;
      default: break;
    }
  }
  launch_tdf3_abl<NREP, MREP, 0>(a, w3, s);
  return true;
}

static void launch_tdf_dma_auto(asx_engine *e, const TdfDmaArgs &d, hipStream_t s) {
  static const int t128 = getenv("ASX_GEMM_T128") ? atoi(getenv("ASX_GEMM_T128")) : 1;
  const bool v2 = tdf2_ok(d);
  static const int small = getenv("ASX_TDF2_SMALL") ? atoi(getenv("ASX_TDF2_SMALL")) : 0;   // A/B: 64 x 128 tiles (3+ workgroups per CU) on short-K layers
  const bool v3 = tdf3_ok(e, d);
  if (v3 && ((small && d.K <= small) || d.prefer_small) && d.N > 128 && launch_tdf3<2, 4>(e, d, s)) return;
  if (v2 && ((small && d.K <= small) || d.prefer_small) && d.N > 128) return launch_tdf2<2, 4>(d, s);
  if (d.N > 128) {
    const double rows = (double)((d.M + 127) / 128);
    auto cost = [&](int bn, double eff) {
      const double blocks = rows * (double)((d.N + bn - 1) / bn);
      return ceil(blocks / 512.0) * bn / eff;
    };
    const bool narrow = t128 == 2 || (t128 == 1 && cost(128, 0.96) < cost(192, 1.0));
    // ASX_TDF3_EFF128: relative efficiency charged to the 128-column tile of the bf16 x 6 kernel.  Measured on the BS-Roformer and
    // HTDemucs linears: 0.96 / 0.85 / 0.75 -> 1251 / 1262 / 1262 ms and 30.5 / 30.4 / 30.5 ms per song -- no reason to move off
    // the fp32 kernel's figure (N = 512 stays on four 128-column tiles).
    static const double eff128 = getenv("ASX_TDF3_EFF128") ? atof(getenv("ASX_TDF3_EFF128")) : 0.96;
    const bool narrow3 = t128 == 2 || (t128 == 1 && cost(128, eff128) < cost(192, 1.0));
    if (v3 && (narrow3 ? launch_tdf3<2, 8>(e, d, s) : launch_tdf3<3, 8>(e, d, s))) return;
    if (narrow) v2 ? launch_tdf2<2, 8>(d, s) : launch_tdf_dma_t<2, 8>(d, s);
    else v2 ? launch_tdf2<3, 8>(d, s) : launch_tdf_dma_t<3, 8>(d, s);
  } else if (d.N > 64) {
    if (v3 && launch_tdf3<2, 4>(e, d, s)) return;
    v2 ? launch_tdf2<2, 4>(d, s) : launch_tdf_dma_t<2, 4>(d, s);
  } else {
    launch_tdf_dma_t<1, 4>(d, s);
  }
}

static int tdf_launch(asx_engine *e, const TdfLayer &L, const float *x, const float *res, float *y, int64_t M,
                      int T, hipStream_t s, int relu = 1) {
  TdfArgs a{};
  a.x = x;
  a.w = L.w.f();
  a.bias = L.has_bias ? L.bias.f() : nullptr;
  a.scale = L.scale.p ? L.scale.f() : nullptr;
  a.shift = L.shift.p ? L.shift.f() : nullptr;
  a.relu = relu;
  a.res = res;
  a.y = y;
  a.M = M;
  a.N = L.n;
  a.K = L.k;
  a.C = L.c;
  a.T = T;
  if (M <= 0) return ASX_OK;
  const double flops = 2.0 * (double)M * L.n * L.k;
  const double bytes = 4.0 * ((double)M * L.k + (double)M * L.n * (res ? 2 : 1) + (double)L.n * L.k);
  const bool dma = (L.k % 4 == 0) && ((reinterpret_cast<uintptr_t>(x) & 15) == 0) && getenv("ASX_NO_DMA") == nullptr;
  TdfDmaArgs d{};
  d.x = a.x;
  d.w = a.w;
  d.bias = a.bias;
  d.scale = a.scale;
  d.shift = a.shift;
  d.res = a.res;
  d.zeros = e->d_zeros.f();
  d.y = a.y;
  d.M = a.M;
  d.N = a.N;
  d.K = a.K;
  d.C = a.C;
  d.T = a.T;
  d.relu = a.relu;
  static const int nt_mode = getenv("ASX_NT") ? atoi(getenv("ASX_NT")) : 0;
  d.nt = ((nt_mode >> 1) & 1) | ((nt_mode >> 2) & 1) << 1;
  return timed(e, ASX_PROF_TDF, flops, bytes, s, [&]() {
    if (dma) {
      launch_tdf_dma_auto(e, d, s);
    } else {
      if (L.n > 128) launch_tdf_t<3, 8>(a, s);
      else if (L.n > 64) launch_tdf_t<2, 4>(a, s);
      else launch_tdf_t<1, 4>(a, s);
    }
  });
}

static int tdf_pack(TdfLayer &L, int n, int k, int c, const float *w, const float *bias, const float *scale,
                    const float *shift) {
  L.n = n;
  L.k = k;
  L.c = c;
  L.has_bias = bias != nullptr;
  CHK(L.w.ensure((size_t)n * k * 4));
  HIPCHK(hipMemcpy(L.w.p, w, (size_t)n * k * 4, hipMemcpyHostToDevice));
  if (bias) {
    CHK(L.bias.ensure((size_t)n * 4));
    HIPCHK(hipMemcpy(L.bias.p, bias, (size_t)n * 4, hipMemcpyHostToDevice));
  }
  if (scale && shift) {
    CHK(L.scale.ensure((size_t)c * 4));
    CHK(L.shift.ensure((size_t)c * 4));
    HIPCHK(hipMemcpy(L.scale.p, scale, (size_t)c * 4, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(L.shift.p, shift, (size_t)c * 4, hipMemcpyHostToDevice));
  }
  return ASX_OK;
}

// ----------------------------------------------------------------------------
// STFT / iSTFT launch helpers (device buffers)
// ----------------------------------------------------------------------------
static int stft_launch(asx_engine *e, const float *wave, const int64_t *d_starts, int64_t n_song, int B, int64_t C,
                       int T, float *spec, int tf_layout, int zero_low, float sign, hipStream_t s) {
  StftArgs a{};
  a.wave = wave;
  a.chunk_start = d_starts;
  a.n_song = n_song;
  a.trim = e->cfg.n_fft / 2;
  a.C = C;
  a.hop = e->cfg.hop_length;
  a.T = T;
  a.dim_f = e->cfg.dim_f;
  a.zero_low = zero_low;
  a.tf_layout = tf_layout;
  a.spec = spec;
  a.window = e->d_window.f();
  a.tw = reinterpret_cast<const float2 *>(e->d_tw.p);
  a.sign = sign;
  const double bytes = 4.0 * ((double)B * 2 * C + (double)B * 4 * T * e->cfg.dim_f);
  FftPlan p = e->plan;
  if (e->fft3 && tf_layout == 1 && e->cfg.dim_f <= f3::NH) {
    // n_fft 6144 / hop 1024, engine-internal [B, 4, T, F] layout: three-pass register FFT (kernels_fft3.h)
    f3::Stft3Args f{};
    f.wave = wave;
    f.chunk_start = d_starts;
    f.n_song = n_song;
    f.trim = a.trim;
    f.C = C;
    f.T = T;
    f.dim_f = a.dim_f;
    f.zero_low = zero_low;
    f.spec = spec;
    f.window = a.window;
    f.tw = reinterpret_cast<const f3::cplx *>(a.tw);
    f.twB = reinterpret_cast<const f3::cplx *>(e->d_tw3.p);
    f.twC = f.twB + 16 * 12;
    f.sign = sign;
    // ~16 frames per workgroup, in even shares: many short workgroups balance better over the CUs than two or three rounds
    // of long ones (measured: 8 / 16 frames 0.257 / 0.253 ms, 20 frames in exactly two rounds 0.368 ms)
    static const int GS = getenv("ASX_FFT3_GS") ? std::max(1, atoi(getenv("ASX_FFT3_GS"))) : 16;
    f.n_groups = std::max(1, T / GS);
    return timed(e, ASX_PROF_STFT, 0.0, bytes, s, [&]() {
      if (e->fft3p)
        hipLaunchKernelGGL(f3::stft3p_kernel, dim3(f.n_groups, 2, B), dim3(256), f3::STFT3P_LDS_BYTES, s, f);
      else
        hipLaunchKernelGGL(f3::stft3_kernel, dim3(T, 2, B), dim3(256), f3::STFT3_LDS_BYTES, s, f);
    });
  }
  return timed(e, ASX_PROF_STFT, 0.0, bytes, s, [&]() {
    hipLaunchKernelGGL(stft_kernel, dim3(T, 2, B), dim3(256), stft_lds(p), s, a, p);
  });
}

static int istft_launch(asx_engine *e, const float *spec, int B, int T, int tf_layout, int combine, float *frames,
                        hipStream_t s) {
  IstftArgs a{};
  a.spec = spec;
  a.T = T;
  a.dim_f = e->cfg.dim_f;
  a.tf_layout = tf_layout;
  a.combine = combine;
  a.frames = frames;
  a.window = e->d_window.f();
  a.tw = reinterpret_cast<const float2 *>(e->d_tw.p);
  const double bytes = 4.0 * ((double)B * 4 * T * e->cfg.dim_f * (combine ? 2 : 1) + (double)B * 2 * T * e->cfg.n_fft);
  FftPlan p = e->plan;
  return timed(e, ASX_PROF_ISTFT, 0.0, bytes, s, [&]() {
    hipLaunchKernelGGL(istft_kernel, dim3(T, 2, B), dim3(256), istft_lds(p), s, a, p);
  });
}

static int ola_launch(asx_engine *e, const float *frames, const float *env, const int64_t *d_nact, int B, int T,
                      int64_t C, float *out, hipStream_t s) {
  const double bytes = 4.0 * ((double)B * 2 * T * e->cfg.n_fft + (double)B * 2 * C);
  const int n_fft = e->cfg.n_fft, hop = e->cfg.hop_length;
  return timed(e, ASX_PROF_OLA, 0.0, bytes, s, [&]() {
    hipLaunchKernelGGL(ola_kernel, dim3((unsigned)((C + 255) / 256), 2, B), dim3(256), 0, s, frames, env, d_nact,
                       n_fft, hop, T, C, out);
  });
}

// STFT.inverse + the chunk's Hann window: the fast path (n_fft 6144 / hop 1024) inverse-transforms, overlap-adds in an LDS
// ring and writes the chunk directly (kernels_fft3.h); every other geometry runs istft_kernel -> frames -> ola_kernel.
static int istft_ola_launch(asx_engine *e, const float *spec, int B, int T, int combine, const int64_t *d_nact, int64_t C,
                            float *out, hipStream_t s) {
  // emit_hop / emit_finish / seam3_kernel store float4: `out` (the caller's chunk buffer + a chunk offset) must be 16-byte aligned
  if (!(e->fft3 && e->cfg.dim_f <= f3::NH && C == (int64_t)f3::HOP * (T - 1) && (reinterpret_cast<uintptr_t>(out) & 15) == 0)) {
    CHK(istft_launch(e, spec, B, T, 1, combine, e->frames.f(), s));
    return ola_launch(e, e->frames.f(), e->d_env.f(), d_nact, B, T, C, out, s);
  }
  // >= 16 frames per workgroup, in even shares.  The grouping depends on T only: a hop on a seam is summed as
  // (tail partial) + (head partial), so results stay bit-identical whatever the batch size.
  static const int G = getenv("ASX_FFT3_G") ? std::max(5, atoi(getenv("ASX_FFT3_G"))) : 16;
  const int ng = std::max(1, T / G);
  CHK(e->seam3.ensure((size_t)B * 2 * ng * 2 * 5 * f3::HOP * 4));
  f3::Istft3Args f{};
  f.spec = spec;
  f.T = T;
  f.dim_f = e->cfg.dim_f;
  f.combine = combine;
  f.window = e->d_window.f();
  f.tw = reinterpret_cast<const f3::cplx *>(e->d_tw.p);
  f.twB = reinterpret_cast<const f3::cplx *>(e->d_tw3.p);
  f.twC = f.twB + 16 * 12;
  f.env = e->d_env.f();
  f.n_act = d_nact;
  f.C = C;
  f.out = out;
  f.seam = e->seam3.f();
  f.n_groups = ng;
  f.hann = (C == (int64_t)e->cfg.hop_length * (e->cfg.segment_size - 1) && e->d_hann3.p) ? reinterpret_cast<const double *>(e->d_hann3.p) : nullptr;
  // algorithmic bytes: the spectrogram read once, the chunk written once (the seam buffer's round trip is overhead, not counted)
  const double bytes = 4.0 * ((double)B * 4 * T * e->cfg.dim_f * (combine ? 2 : 1) + (double)B * 2 * C);
  return timed(e, ASX_PROF_ISTFT, 0.0, bytes, s, [&]() {
    const bool aligned = e->cfg.dim_f % 4 == 0 && (reinterpret_cast<uintptr_t>(spec) & 15) == 0;
    if (e->fft3p && combine == 0 && aligned)
    {
      static const int abl = getenv("ASX_ISTFT_ABL") ? atoi(getenv("ASX_ISTFT_ABL")) : 0;   // timing probes (results invalid)
      if (abl == 1) hipLaunchKernelGGL(f3::istft3p_kernel<1>, dim3(ng, 2, B), dim3(256), f3::ISTFT3P_LDS_BYTES, s, f);
      else if (abl == 2) hipLaunchKernelGGL(f3::istft3p_kernel<2>, dim3(ng, 2, B), dim3(256), f3::ISTFT3P_LDS_BYTES, s, f);
      else if (abl == 3) hipLaunchKernelGGL(f3::istft3p_kernel<3>, dim3(ng, 2, B), dim3(256), f3::ISTFT3P_LDS_BYTES, s, f);
      else hipLaunchKernelGGL(f3::istft3p_kernel<0>, dim3(ng, 2, B), dim3(256), f3::ISTFT3P_LDS_BYTES, s, f);
    }
    else
      hipLaunchKernelGGL(f3::istft3_kernel, dim3(ng, 2, B), dim3(256), f3::ISTFT3_LDS_BYTES, s, f);
    if (ng > 1) hipLaunchKernelGGL(f3::seam3_kernel, dim3(5, ng - 1, B * 2), dim3(256), 0, s, f);
  });
}

// ----------------------------------------------------------------------------
// net forward on device buffers: spec_in [B,4,T,F] (TF layout) -> spec_out
// ----------------------------------------------------------------------------
// GroupNorm(2, C) + ReLU (+ res) (* mul) over a dense [B, C, P] tensor, in place or into y (kernels_fft.h: gn_*_kernel); the ConvTDFNet
// variant built with optimizer == 'adamw' (uvr_lib_v5/mdxnet.py:48-49) normalises with batch-dependent statistics, so the norm cannot
// be folded into the weights: every conv / linear runs bare (bias only) and this pass follows it.
static int gn_launch(asx_engine *e, const float *x, int B, int C, int64_t P, const DevBuf &gw, const DevBuf &gb, const float *res,
                     const float *mul, float *y, hipStream_t s) {
  if (B <= 0) return ASX_OK;
  REQUIRE(C % 2 == 0 && gw.p && gb.p && e->gn_part.bytes >= (size_t)B * C * 16, "GroupNorm(2, %d): channels / workspace", C);
  double2 *part = reinterpret_cast<double2 *>(e->gn_part.p);
  const double bytes = 4.0 * (double)B * C * (double)P;
  CHK(timed(e, ASX_PROF_MISC, 0.0, bytes, s,
            [&]() { hipLaunchKernelGGL(gn_partial_kernel, dim3((unsigned)C, (unsigned)B), dim3(256), 0, s, x, C, P, part); }));
  const int nx = (int)std::max<int64_t>(1, std::min<int64_t>(32, (P / 4 + 1023) / 1024));
  return timed(e, ASX_PROF_MISC, 0.0, bytes * (2 + (res ? 1 : 0) + (mul ? 1 : 0)), s, [&]() {
    hipLaunchKernelGGL(gn_apply_kernel, dim3((unsigned)nx, (unsigned)C, (unsigned)B), dim3(256), 0, s, x, C, 2, P, part, gw.f(), gb.f(), 1e-5f, 1, res,
                       mul, y);
  });
}

static int block_forward(asx_engine *e, const Block &blk, float *&cur, float *dest, int B, hipStream_t s) {
  // TFC convs rotate through e->R; the TDF output goes to `dest` (or a free R buffer when dest == nullptr)
  auto next_free = [&](const float *a, const float *b) -> float * {
    for (int i = 0; i < 3; ++i)
      if (e->R[i].f() != a && e->R[i].f() != b) return e->R[i].f();
    return nullptr;
  };
  const bool gn = e->net.norm == 1;
  const int64_t plane = (int64_t)blk.t * blk.f;
  for (size_t j = 0; j < blk.tfc.size(); ++j) {
    float *out = next_free(cur, nullptr);
    CHK(conv_launch(e, blk.tfc[j], cur, nullptr, out, B, blk.t, blk.f, s));
    if (gn) CHK(gn_launch(e, out, B, blk.c, plane, blk.tfc[j].gn_w, blk.tfc[j].gn_b, nullptr, nullptr, out, s));
    cur = out;
  }
  const int64_t M = (int64_t)B * blk.c * blk.t;
  const int bnk = e->net.bn;
  if (bnk < 0) {                                       // bn is None: TFC only (modules.py:52, 74)
    if (dest) {
      HIPCHK(hipMemcpyAsync(dest, cur, (size_t)M * blk.f * 4, hipMemcpyDeviceToDevice, s));
      cur = dest;
    }
    return ASX_OK;
  }
  if (bnk == 0) {                                      // one Linear(f, f) + norm + ReLU (modules.py:55-60), x + tdf(x)
    float *out = dest ? dest : next_free(cur, nullptr);
    if (gn) {
      float *tmp = next_free(cur, out);
      CHK(tdf_launch(e, blk.tdf0, cur, nullptr, tmp, M, blk.t, s, 0));
      CHK(gn_launch(e, tmp, B, blk.c, plane, blk.tdf0.gn_w, blk.tdf0.gn_b, cur, nullptr, out, s));
    } else {
      CHK(tdf_launch(e, blk.tdf0, cur, cur, out, M, blk.t, s));
    }
    cur = out;
    return ASX_OK;
  }
  if (gn) {
    CHK(tdf_launch(e, blk.tdf0, cur, nullptr, e->H.f(), M, blk.t, s, 0));
    CHK(gn_launch(e, e->H.f(), B, blk.c, (int64_t)blk.t * (blk.f / bnk), blk.tdf0.gn_w, blk.tdf0.gn_b, nullptr, nullptr, e->H.f(), s));
    float *out = dest ? dest : next_free(cur, nullptr);
    float *tmp = dest ? next_free(cur, nullptr) : out;
    CHK(tdf_launch(e, blk.tdf1, e->H.f(), nullptr, tmp, M, blk.t, s, 0));
    CHK(gn_launch(e, tmp, B, blk.c, plane, blk.tdf1.gn_w, blk.tdf1.gn_b, cur, nullptr, out, s));
    cur = out;
    return ASX_OK;
  }
  CHK(tdf_launch(e, blk.tdf0, cur, nullptr, e->H.f(), M, blk.t, s));
  // (A/B ASX_TDF_INPLACE=1: x + tdf(x) written over x where no skip copy is needed -- every output element depends on the
  // same element of x only)
  static const bool inplace = getenv("ASX_TDF_INPLACE") && atoi(getenv("ASX_TDF_INPLACE")) != 0;
  float *out = dest ? dest : (inplace ? cur : next_free(cur, nullptr));
  CHK(tdf_launch(e, blk.tdf1, e->H.f(), cur, out, M, blk.t, s));
  cur = out;
  return ASX_OK;
}

static int net_forward_dev(asx_engine *e, const float *spec_in, float *spec_out, int B, hipStream_t s) {
  if (!e->net_ready) {
    set_err("net weights not committed");
    return ASX_ERR_STATE;
  }
  const int T = e->net.dim_t, F = e->net.dim_f;
  const int n = e->net.num_blocks / 2;
  const bool gn = e->net.norm == 1;
  float *cur = e->R[0].f();
  CHK(conv_launch(e, e->first, spec_in, nullptr, cur, B, T, F, s));
  if (gn) CHK(gn_launch(e, cur, B, e->net.g, (int64_t)T * F, e->first.gn_w, e->first.gn_b, nullptr, nullptr, cur, s));
  for (int i = 0; i < n; ++i) {
    CHK(block_forward(e, e->enc[i], cur, e->skip[i].f(), B, s));
    float *out = e->R[0].f();
    CHK(conv_launch(e, e->ds[i], cur, nullptr, out, B, e->enc[i].t, e->enc[i].f, s));
    if (gn)
      CHK(gn_launch(e, out, B, e->ds[i].cout, (int64_t)(e->enc[i].t / 2) * (e->enc[i].f / 2), e->ds[i].gn_w, e->ds[i].gn_b, nullptr, nullptr, out, s));
    cur = out;
  }
  CHK(block_forward(e, e->mid, cur, nullptr, B, s));
  for (int i = 0; i < n; ++i) {
    const Block &blk = e->dec[i];
    float *out = nullptr;
    for (int r = 0; r < 3; ++r)
      if (e->R[r].f() != cur) {
        out = e->R[r].f();
        break;
      }
    // input at (t/2, f/2) -> (t, f), multiplied by the matching encoder output (mdxnet.py:113)
    CHK(conv_launch(e, e->us[i], cur, gn ? nullptr : e->skip[n - 1 - i].f(), out, B, blk.t / 2, blk.f / 2, s));
    if (gn)   // norm + ReLU first, then `x *= ds_outputs[-i - 1]` (mdxnet.py:111-113)
      CHK(gn_launch(e, out, B, e->us[i].cout, (int64_t)blk.t * blk.f, e->us[i].gn_w, e->us[i].gn_b, nullptr, e->skip[n - 1 - i].f(), out, s));
    cur = out;
    CHK(block_forward(e, blk, cur, nullptr, B, s));
  }
  CHK(conv_launch(e, e->final_, cur, nullptr, spec_out, B, T, F, s));
  return ASX_OK;
}

// ----------------------------------------------------------------------------
// workspace
// ----------------------------------------------------------------------------
static int ensure_workspace(asx_engine *e, int Bchunks, bool need_net) {
  const int T = e->cfg.segment_size, Fq = e->cfg.dim_f;
  const int mult = (need_net && e->cfg.enable_denoise) ? 2 : 1;
  const size_t Bn = (size_t)Bchunks * mult;
  CHK(e->spec_in.ensure(Bn * 4 * T * Fq * 4));
  CHK(e->frames.ensure((size_t)Bchunks * 2 * T * e->cfg.n_fft * 4));
  if (need_net) {
    CHK(e->spec_out.ensure(Bn * 4 * T * Fq * 4));
    const size_t lvl0 = Bn * e->net.g * T * Fq * 4;
    for (int i = 0; i < 3; ++i) CHK(e->R[i].ensure(lvl0));
    CHK(e->H.ensure(Bn * e->net.g * T * (Fq / std::max(1, e->net.bn)) * 4 + 256));
    if (e->net.norm == 1) CHK(e->gn_part.ensure(Bn * (size_t)e->net.g * (e->net.num_blocks / 2 + 1) * 16));
    const int n = e->net.num_blocks / 2;
    e->skip.resize(n);
    for (int i = 0; i < n; ++i) {
      const size_t c = (size_t)e->net.g * (i + 1), t = T >> i, f = Fq >> i;
      CHK(e->skip[i].ensure(Bn * c * t * f * 4));
    }
  }
  e->ws_batch = std::max(e->ws_batch, Bchunks);
  return ASX_OK;
}

// ----------------------------------------------------------------------------
// C ABI
// ----------------------------------------------------------------------------
extern "C" {

int asx_abi_version(void) { return ASX_ABI_VERSION; }
const char *asx_last_error(void) { return g_err; }

int asx_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

int asx_engine_create(int device, const asx_mdx_config *cfg, asx_engine **out) {
  REQUIRE(cfg && out, "asx_engine_create: null argument");
  REQUIRE(cfg->n_fft >= 8 && cfg->n_fft % 2 == 0, "n_fft must be even and >= 8 (got %d)", cfg->n_fft);
  REQUIRE(cfg->hop_length > 0 && cfg->segment_size >= 2, "bad hop_length/segment_size");
  REQUIRE(cfg->dim_f > 0 && cfg->dim_f <= cfg->n_fft / 2 + 1, "dim_f %d out of range for n_fft %d", cfg->dim_f,
          cfg->n_fft);
  REQUIRE(cfg->overlap >= 0.0 && cfg->overlap < 1.0, "overlap must be in [0,1)");
  REQUIRE(cfg->win_length >= 0 && cfg->win_length <= cfg->n_fft, "win_length %d out of range for n_fft %d", cfg->win_length, cfg->n_fft);
  const int64_t C = (int64_t)cfg->hop_length * (cfg->segment_size - 1);
  REQUIRE(C > cfg->n_fft / 2, "chunk_size %lld must exceed n_fft/2 (reflect padding)", (long long)C);
  REQUIRE(C - cfg->n_fft > 0, "chunk_size %lld must exceed n_fft (gen_size > 0)", (long long)C);
  FftPlan plan{};
  REQUIRE(make_plan(cfg->n_fft, &plan), "n_fft/2 = %d must factor into {2,3,5}", cfg->n_fft / 2);
  int ndev = 0;
  HIPCHK(hipGetDeviceCount(&ndev));
  REQUIRE(device >= 0 && device < ndev, "device %d not available (%d visible)", device, ndev);
  HIPCHK(hipSetDevice(device));
  asx_engine *e = new asx_engine();
  e->device = device;
  e->cfg = *cfg;
  e->plan = plan;
  std::vector<float> w;
  host_window(cfg->n_fft, w, cfg->win_length);
  std::vector<float> tw((size_t)cfg->n_fft * 2);
  for (int j = 0; j < cfg->n_fft; ++j) {
    const double ang = -2.0 * M_PI * (double)j / (double)cfg->n_fft;
    tw[2 * j] = (float)cos(ang);
    tw[2 * j + 1] = (float)sin(ang);
  }
  std::vector<float> env;
  host_env(cfg->n_fft, cfg->hop_length, cfg->segment_size, env, cfg->win_length);
  int rc = ASX_OK;
  if ((rc = e->d_window.ensure(w.size() * 4)) == ASX_OK && (rc = e->d_tw.ensure(tw.size() * 4)) == ASX_OK &&
      (rc = e->d_env.ensure(env.size() * 4)) == ASX_OK) {
    if (hipMemcpy(e->d_window.p, w.data(), w.size() * 4, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(e->d_tw.p, tw.data(), tw.size() * 4, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(e->d_env.p, env.data(), env.size() * 4, hipMemcpyHostToDevice) != hipSuccess) {
      set_err("table upload failed");
      rc = ASX_ERR_HIP;
    }
  }
  if (rc == ASX_OK && cfg->n_fft == f3::NFFT && cfg->hop_length == f3::HOP && !(getenv("ASX_FFT3") && atoi(getenv("ASX_FFT3")) == 0)) {
    std::vector<float> t3((size_t)(16 * 12 + 16 * f3::NB) * 2);
    for (int r = 0; r < 16; ++r)
      for (int k = 0; k < 12; ++k) {
        const double ang = -2.0 * M_PI * (double)(k * r) / 192.0;
        t3[(size_t)(r * 12 + k) * 2] = (float)cos(ang);
        t3[(size_t)(r * 12 + k) * 2 + 1] = (float)sin(ang);
      }
    for (int r = 0; r < 16; ++r)
      for (int j = 0; j < f3::NB; ++j) {
        const double ang = -2.0 * M_PI * (double)(j * r) / (double)f3::NH;
        t3[(size_t)(16 * 12 + r * f3::NB + j) * 2] = (float)cos(ang);
        t3[(size_t)(16 * 12 + r * f3::NB + j) * 2 + 1] = (float)sin(ang);
      }
    if ((rc = e->d_tw3.ensure(t3.size() * 4)) == ASX_OK) {
      if (hipMemcpy(e->d_tw3.p, t3.data(), t3.size() * 4, hipMemcpyHostToDevice) != hipSuccess) {
        set_err("table upload failed");
        rc = ASX_ERR_HIP;
      } else {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&f3::stft3_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  f3::STFT3_LDS_BYTES);
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&f3::istft3_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  f3::ISTFT3_LDS_BYTES);
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&f3::istft3p_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  f3::ISTFT3P_LDS_BYTES);
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&f3::istft3p_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  f3::ISTFT3P_LDS_BYTES);
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&f3::istft3p_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  f3::ISTFT3P_LDS_BYTES);
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&f3::istft3p_kernel<3>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  f3::ISTFT3P_LDS_BYTES);
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&f3::stft3p_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  f3::STFT3P_LDS_BYTES);
        e->fft3 = true;
        e->fft3p = !(getenv("ASX_FFT3P") && atoi(getenv("ASX_FFT3P")) == 0);
        if ((rc = e->d_hann3.ensure((size_t)C * 8)) == ASX_OK) {
          hipLaunchKernelGGL(f3::hann3_table_kernel, dim3((unsigned)((C + 255) / 256)), dim3(256), 0, nullptr, C,
                             reinterpret_cast<double *>(e->d_hann3.p));
          if (hipDeviceSynchronize() != hipSuccess) {
            set_err("hann table kernel failed");
            rc = ASX_ERR_HIP;
          }
        }
      }
    }
  }
  if (rc == ASX_OK) rc = e->d_zeros.ensure(256);
  if (rc == ASX_OK && hipMemset(e->d_zeros.p, 0, 256) != hipSuccess) {
    set_err("zero page init failed");
    rc = ASX_ERR_HIP;
  }
  if (rc == ASX_OK) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&stft_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)stft_lds(plan));
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&istft_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)istft_lds(plan));
  }
  if (rc != ASX_OK) {
    asx_engine_destroy(e);
    return rc;
  }
  *out = e;
  return ASX_OK;
}

static void v3_destroy(V3Net *n);
static void rof_destroy(RofNet *n);
static void ht_destroy(HtNet *n);
static void hd_destroy(HdNet *n);
static void hd_drop_clones(asx_engine *e);
static void vr_destroy(VrNet *n);
static void ens_destroy(EnsCtx *c);
static void free_conv(ConvLayer &L) {
  L.w.release();
  L.b.release();
  L.wu.release();
  L.wu2.release();
  L.wu3.release();
  L.wus.release();
  L.wu6.release();
  L.gn_w.release();
  L.gn_b.release();
}
static void free_tdf(TdfLayer &L) {
  L.w.release();
  L.bias.release();
  L.scale.release();
  L.shift.release();
  L.gn_w.release();
  L.gn_b.release();
}
static void free_block(Block &b) {
  for (auto &c : b.tfc) free_conv(c);
  free_tdf(b.tdf0);
  free_tdf(b.tdf1);
}

void asx_engine_destroy(asx_engine *e) {
  w3_flush(e);
  if (!e) return;
  (void)hipSetDevice(e->device);
  for (auto &r : e->recs) {
    (void)hipEventDestroy(r.a);
    (void)hipEventDestroy(r.b);
  }
  if (e->div_ev) (void)hipEventDestroy(e->div_ev);
  e->sinc_tab.release();
  e->d_window.release();
  e->d_tw.release();
  e->d_env.release();
  e->d_tw3.release();
  e->d_hann3.release();
  e->seam3.release();
  e->d_zeros.release();
  free_conv(e->first);
  free_conv(e->final_);
  for (auto &b : e->enc) free_block(b);
  for (auto &b : e->dec) free_block(b);
  free_block(e->mid);
  for (auto &c : e->ds) free_conv(c);
  for (auto &c : e->us) free_conv(c);
  e->spec_in.release();
  e->spec_out.release();
  for (auto &r : e->R) r.release();
  e->H.release();
  e->frames.release();
  e->chunk_out.release();
  e->d_starts.release();
  e->d_nact.release();
  e->d_peak.release();
  e->d_demixed.release();
  e->d_div.release();
  for (auto &sk : e->skip) sk.release();
  if (e->v3) v3_destroy(e->v3);
  if (e->rof) rof_destroy(e->rof);
  hd_drop_clones(e);
  if (e->ht) ht_destroy(e->ht);
  if (e->hd) hd_destroy(e->hd);
  if (e->vr) vr_destroy(e->vr);
  if (e->ens) ens_destroy(e->ens);
  delete e;
}

// ---- weights ---------------------------------------------------------------
int asx_net_begin(asx_engine *e, const asx_net_config *cfg) {
  w3_flush(e);
  REQUIRE(e && cfg, "asx_net_begin: null argument");
  REQUIRE(cfg->dim_f == e->cfg.dim_f, "net dim_f %d != engine dim_f %d", cfg->dim_f, e->cfg.dim_f);
  REQUIRE(cfg->dim_t == e->cfg.segment_size, "net dim_t %d != segment_size %d", cfg->dim_t, e->cfg.segment_size);
  REQUIRE(cfg->k == 3, "only k=3 TFC kernels are supported (got %d)", cfg->k);
  REQUIRE(cfg->dim_c > 0 && cfg->g > 0 && cfg->l > 0 && cfg->bn >= -1, "bad net hyper-parameters");
  REQUIRE(cfg->norm == 0 || (cfg->norm == 1 && cfg->g % 2 == 0), "norm must be 0 (BatchNorm, folded by the host) or 1 (GroupNorm(2, c): even g)");
  REQUIRE(cfg->num_blocks >= 1 && cfg->num_blocks % 2 == 1, "num_blocks must be odd");
  const int n = cfg->num_blocks / 2;
  REQUIRE((cfg->dim_f % (1 << n)) == 0 && (cfg->dim_t % (1 << n)) == 0,
          "dim_f and dim_t must be divisible by 2^%d", n);
  REQUIRE(cfg->bn <= 0 || ((cfg->dim_f >> n) % cfg->bn) == 0, "dim_f / 2^n must be divisible by bn");
  e->net = *cfg;
  e->host_tensors.clear();
  e->net_begun = true;
  e->net_ready = false;
  return ASX_OK;
}

int asx_net_set_tensor(asx_engine *e, const char *name, const float *host, int64_t numel) {
  REQUIRE(e && name && host && numel > 0, "asx_net_set_tensor: bad argument");
  if (!e->net_begun) {
    set_err("asx_net_set_tensor before asx_net_begin");
    return ASX_ERR_STATE;
  }
  e->host_tensors[name].assign(host, host + numel);
  return ASX_OK;
}

static int get_tensor(asx_engine *e, const std::string &name, int64_t numel, const float **out, bool optional = false) {
  auto it = e->host_tensors.find(name);
  if (it == e->host_tensors.end()) {
    if (optional) {
      *out = nullptr;
      return ASX_OK;
    }
    set_err("missing tensor '%s'", name.c_str());
    return ASX_ERR_INVALID;
  }
  if ((int64_t)it->second.size() != numel) {
    set_err("tensor '%s': expected %lld elements, got %zu", name.c_str(), (long long)numel, it->second.size());
    return ASX_ERR_INVALID;
  }
  *out = it->second.data();
  return ASX_OK;
}

// GroupNorm affine of a layer: "<name>.gn_w" / "<name>.gn_b" [c] (asx_net_config.norm == 1)
static int load_gn(asx_engine *e, const std::string &name, int c, DevBuf &gw, DevBuf &gb) {
  const float *w, *b;
  CHK(get_tensor(e, name + ".gn_w", c, &w));
  CHK(get_tensor(e, name + ".gn_b", c, &b));
  CHK(gw.ensure((size_t)c * 4));
  CHK(gb.ensure((size_t)c * 4));
  HIPCHK(hipMemcpy(gw.p, w, (size_t)c * 4, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(gb.p, b, (size_t)c * 4, hipMemcpyHostToDevice));
  return ASX_OK;
}

static int build_block(asx_engine *e, Block &blk, const std::string &pre, int c, int t, int f) {
  const asx_net_config &n = e->net;
  const bool gn = n.norm == 1;
  blk.c = c;
  blk.t = t;
  blk.f = f;
  blk.tfc.resize(n.l);
  for (int j = 0; j < n.l; ++j) {
    const float *w, *b;
    const std::string nm = pre + ".tfc" + std::to_string(j);
    CHK(get_tensor(e, nm + ".w", (int64_t)c * c * 9, &w));
    CHK(get_tensor(e, nm + ".b", c, &b));
    CHK(conv_setup(blk.tfc[j], CK_3X3, c, c, gn ? 0 : 1));
    CHK(conv_pack(blk.tfc[j], w, b, e->winograd));
    if (gn) CHK(load_gn(e, nm, c, blk.tfc[j].gn_w, blk.tfc[j].gn_b));
  }
  if (n.bn < 0) return ASX_OK;                         // bn is None: no TDF branch (modules.py:52)
  const int fb = n.bn == 0 ? f : f / n.bn;             // bn == 0: ONE Linear(f, f) (modules.py:55-60)
  const float *w, *bias, *sc = nullptr, *sh = nullptr;
  CHK(get_tensor(e, pre + ".tdf0.w", (int64_t)fb * f, &w));
  CHK(get_tensor(e, pre + ".tdf0.bias", fb, &bias, !n.tdf_bias));
  if (!gn) {
    CHK(get_tensor(e, pre + ".tdf0.scale", c, &sc));
    CHK(get_tensor(e, pre + ".tdf0.shift", c, &sh));
  }
  CHK(tdf_pack(blk.tdf0, fb, f, c, w, bias, sc, sh));
  if (gn) CHK(load_gn(e, pre + ".tdf0", c, blk.tdf0.gn_w, blk.tdf0.gn_b));
  if (n.bn == 0) return ASX_OK;
  CHK(get_tensor(e, pre + ".tdf1.w", (int64_t)f * fb, &w));
  CHK(get_tensor(e, pre + ".tdf1.bias", f, &bias, !n.tdf_bias));
  if (!gn) {
    CHK(get_tensor(e, pre + ".tdf1.scale", c, &sc));
    CHK(get_tensor(e, pre + ".tdf1.shift", c, &sh));
  }
  CHK(tdf_pack(blk.tdf1, f, fb, c, w, bias, sc, sh));
  if (gn) CHK(load_gn(e, pre + ".tdf1", c, blk.tdf1.gn_w, blk.tdf1.gn_b));
  return ASX_OK;
}

int asx_net_commit(asx_engine *e) {
  w3_flush(e);
  REQUIRE(e, "asx_net_commit: null engine");
  if (!e->net_begun) {
    set_err("asx_net_commit before asx_net_begin");
    return ASX_ERR_STATE;
  }
  HIPCHK(hipSetDevice(e->device));
  const asx_net_config &n = e->net;
  const int nn = n.num_blocks / 2;
  const float *w, *b;
  CHK(get_tensor(e, "first.w", (int64_t)n.g * n.dim_c, &w));
  CHK(get_tensor(e, "first.b", n.g, &b));
  const bool gn = n.norm == 1;
  CHK(conv_setup(e->first, CK_1X1, n.dim_c, n.g, gn ? 0 : 1));
  CHK(conv_pack(e->first, w, b, e->winograd));
  if (gn) CHK(load_gn(e, "first", n.g, e->first.gn_w, e->first.gn_b));
  e->enc.assign(nn, Block());
  e->dec.assign(nn, Block());
  e->ds.assign(nn, ConvLayer());
  e->us.assign(nn, ConvLayer());
  int c = n.g, t = n.dim_t, f = n.dim_f;
  for (int i = 0; i < nn; ++i) {
    CHK(build_block(e, e->enc[i], "enc" + std::to_string(i), c, t, f));
    CHK(get_tensor(e, "ds" + std::to_string(i) + ".w", (int64_t)(c + n.g) * c * 4, &w));
    CHK(get_tensor(e, "ds" + std::to_string(i) + ".b", c + n.g, &b));
    CHK(conv_setup(e->ds[i], CK_DOWN, c, c + n.g, gn ? 0 : 1));
    CHK(conv_pack(e->ds[i], w, b, e->winograd));
    if (gn) CHK(load_gn(e, "ds" + std::to_string(i), c + n.g, e->ds[i].gn_w, e->ds[i].gn_b));
    c += n.g;
    t /= 2;
    f /= 2;
  }
  CHK(build_block(e, e->mid, "mid", c, t, f));
  for (int i = 0; i < nn; ++i) {
    CHK(get_tensor(e, "us" + std::to_string(i) + ".w", (int64_t)c * (c - n.g) * 4, &w));
    CHK(get_tensor(e, "us" + std::to_string(i) + ".b", c - n.g, &b));
    CHK(conv_setup(e->us[i], CK_UP, c, c - n.g, gn ? 0 : 1));
    CHK(conv_pack(e->us[i], w, b, e->winograd));
    if (gn) CHK(load_gn(e, "us" + std::to_string(i), c - n.g, e->us[i].gn_w, e->us[i].gn_b));
    c -= n.g;
    t *= 2;
    f *= 2;
    CHK(build_block(e, e->dec[i], "dec" + std::to_string(i), c, t, f));
  }
  CHK(get_tensor(e, "final.w", (int64_t)n.dim_c * n.g, &w));
  CHK(get_tensor(e, "final.b", n.dim_c, &b));
  CHK(conv_setup(e->final_, CK_1X1, n.g, n.dim_c, 0));
  CHK(conv_pack(e->final_, w, b, e->winograd));
  e->host_tensors.clear();
  e->net_ready = true;
  return ASX_OK;
}

double asx_net_flops(const asx_engine *e, int32_t batch) {
  if (!e || !e->net_begun) return 0.0;
  const asx_net_config &d = e->net;
  const int n = d.num_blocks / 2;
  double fl = 2.0 * d.dim_c * d.g * (double)d.dim_t * d.dim_f;
  auto block = [&](double c, double t, double f) {
    const double tdf = d.bn > 0 ? 2.0 * 2.0 * c * t * f * (f / d.bn) : (d.bn == 0 ? 2.0 * c * t * f * f : 0.0);
    return d.l * 2.0 * 9.0 * c * c * t * f + tdf;
  };
  double c = d.g, t = d.dim_t, f = d.dim_f;
  for (int i = 0; i < n; ++i) {
    fl += block(c, t, f);
    fl += 2.0 * 4.0 * c * (c + d.g) * (t / 2) * (f / 2);
    c += d.g;
    t /= 2;
    f /= 2;
  }
  fl += block(c, t, f);
  for (int i = 0; i < n; ++i) {
    fl += 2.0 * 4.0 * c * (c - d.g) * t * f;
    c -= d.g;
    t *= 2;
    f *= 2;
    fl += block(c, t, f);
  }
  fl += 2.0 * d.g * d.dim_c * (double)d.dim_t * d.dim_f;
  return fl * batch;
}

}  // extern "C" (the engine headers below define templates)
#include "engine_v3.h"
#include "engine_rof.h"
#include "engine_ht.h"
#include "engine_hd.h"
#include "engine_vr.h"
#include "engine_ens.h"
extern "C" {

// ---- plan ------------------------------------------------------------------
int asx_plan_query(const asx_engine *e, int64_t N, uint32_t flags, asx_plan *out) {
  REQUIRE(e && out, "asx_plan_query: null argument");
  REQUIRE(N >= 1, "n_samples must be >= 1 (an empty mix raises in the reference, common_separator.py:267)");
  const bool match = (flags & ASX_FLAG_MATCH_MIX) != 0;
  // python: overlap is a float (double); 0.02 for the match-mix pass (mdx_separator.py:311)
  const double overlap = match ? 0.02 : e->cfg.overlap;
  asx_plan p{};
  p.n_samples = N;
  p.trim = e->cfg.n_fft / 2;
  p.chunk_size = (int64_t)e->cfg.hop_length * (e->cfg.segment_size - 1);
  p.gen_size = p.chunk_size - 2 * p.trim;
  p.pad = p.gen_size + p.trim - (N % p.gen_size);
  p.padded_len = p.trim + N + p.pad;
  p.step = (int64_t)((1.0 - overlap) * (double)p.chunk_size);  // int() truncation (mdx_separator.py:335)
  REQUIRE(p.step >= 1, "overlap too close to 1: step == 0");
  p.n_chunks = (int32_t)((p.padded_len + p.step - 1) / p.step);
  p.n_frames = e->cfg.segment_size;
  *out = p;
  return ASX_OK;
}

// ---- chunk batches -----------------------------------------------------------
__global__ void chunk_table_kernel(int k0, int nk, int64_t step, int64_t chunk, int64_t padded_len, int win,
                                   int64_t *__restrict__ starts, int64_t *__restrict__ nact) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nk) return;
  const int64_t st = (int64_t)(k0 + i) * step;
  const int64_t na = chunk < padded_len - st ? chunk : padded_len - st;
  starts[i] = st;
  nact[i] = win ? na : -1;
}

static bool windowed_mode(const asx_engine *e, uint32_t flags) {
  return (flags & ASX_FLAG_MATCH_MIX) ? true : (e->cfg.overlap != 0.0);
}

int asx_demix_chunks_dev(asx_engine *e, const float *mix_dev, int64_t N, int32_t k0, int32_t k1,
                         float *chunk_out_dev, uint32_t flags, void *stream) {
  REQUIRE(e && mix_dev && chunk_out_dev, "asx_demix_chunks_dev: null argument");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  HIPCHK(hipSetDevice(e->device));
  asx_plan p;
  CHK(asx_plan_query(e, N, flags, &p));
  REQUIRE(k0 >= 0 && k1 <= p.n_chunks && k0 <= k1, "chunk range [%d,%d) outside [0,%d)", k0, k1, p.n_chunks);
  const bool match = (flags & ASX_FLAG_MATCH_MIX) != 0;
  const bool need_net = !match;
  if (need_net && !e->net_ready) {
    set_err("asx_demix: net weights not committed");
    return ASX_ERR_STATE;
  }
  const int nk = k1 - k0;
  if (nk == 0) return ASX_OK;
  const int maxB = pick_batch(e);
  const int nbatch = (nk + maxB - 1) / maxB;
  const int per = (nk + nbatch - 1) / nbatch;
  CHK(ensure_workspace(e, per, need_net));
  // chunk tables, built on the device: start of chunk k = k * step, active length = min(chunk, L - start), or negative
  // for "no chunk window" (overlap == 0, mdx_separator.py:389-390).  No host copy, no host synchronisation: the call
  // only enqueues work on `stream` (hipGraph-capturable, and a gather of step k can overlap the compute of step k + 1).
  const bool win = windowed_mode(e, flags);
  CHK(e->d_starts.ensure((size_t)nk * 8));
  CHK(e->d_nact.ensure((size_t)nk * 8));
  hipLaunchKernelGGL(chunk_table_kernel, dim3((unsigned)((nk + 255) / 256)), dim3(256), 0, s, k0, nk, p.step, p.chunk_size,
                     p.padded_len, win ? 1 : 0, reinterpret_cast<int64_t *>(e->d_starts.p), reinterpret_cast<int64_t *>(e->d_nact.p));
  HIPCHK(hipGetLastError());
  const int T = e->cfg.segment_size;
  const int64_t C = p.chunk_size;
  const size_t spec_elems = (size_t)4 * T * e->cfg.dim_f;
  for (int b0 = 0; b0 < nk; b0 += per) {
    const int B = std::min(per, nk - b0);
    const int64_t *ds = reinterpret_cast<const int64_t *>(e->d_starts.p) + b0;
    const int64_t *dn = reinterpret_cast<const int64_t *>(e->d_nact.p) + b0;
    CHK(stft_launch(e, mix_dev, ds, N, B, C, T, e->spec_in.f(), 1, 3, 1.0f, s));
    const float *spec_final = e->spec_in.f();
    int combine = 0;
    if (need_net) {
      int Bn = B;
      if (e->cfg.enable_denoise) {
        // second half of the batch: the negated spectrum (mdx_separator.py:437)
        CHK(stft_launch(e, mix_dev, ds, N, B, C, T, e->spec_in.f() + (size_t)B * spec_elems, 1, 3, -1.0f, s));
        Bn = 2 * B;
        combine = B;
      }
      CHK(net_forward_dev(e, e->spec_in.f(), e->spec_out.f(), Bn, s));
      spec_final = e->spec_out.f();
    }
    CHK(istft_ola_launch(e, spec_final, B, T, combine, dn, C, chunk_out_dev + (size_t)b0 * 2 * C, s));
  }
  return ASX_OK;
}

int asx_finalize_dev(asx_engine *e, const float *chunk_out_dev, int64_t N, float *out_dev, uint32_t flags,
                     void *stream) {
  REQUIRE(e && chunk_out_dev && out_dev, "asx_finalize_dev: null argument");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  HIPCHK(hipSetDevice(e->device));
  asx_plan p;
  CHK(asx_plan_query(e, N, flags, &p));
  const int win = windowed_mode(e, flags) ? 1 : 0;
  const double bytes = 4.0 * ((double)p.n_chunks * 2 * p.chunk_size + 2.0 * N);
  const double *hann = (e->fft3 && e->d_hann3.p) ? reinterpret_cast<const double *>(e->d_hann3.p) : nullptr;
  static const bool fin4 = !(getenv("ASX_FINALIZE4") && atoi(getenv("ASX_FINALIZE4")) == 0);
  // vector path: the divider (input-independent) comes from a table built once per plan; needs 4-sample alignment of the
  // chunk geometry and 16-byte aligned buffers
  if (fin4 && p.chunk_size % 4 == 0 && p.step % 4 == 0 && p.trim % 4 == 0 && N % 4 == 0 && p.chunk_size + (int64_t)p.trim >= 0 &&
      (((uintptr_t)chunk_out_dev | (uintptr_t)out_dev) & 15) == 0) {
    const DivKey key{N, p.n_chunks, p.chunk_size, p.step, p.padded_len, p.trim, win, hann != nullptr ? 1 : 0};
    if (!(e->div_key == key) || !e->d_div.p) {
      CHK(e->d_div.ensure((size_t)N * 4));
      CHK(timed(e, ASX_PROF_MISC, 0.0, 4.0 * N, s, [&]() {
        hipLaunchKernelGGL(finalize_div_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, s, p.n_chunks, p.chunk_size, p.step,
                           p.padded_len, p.trim, N, win, e->d_div.f(), hann);
      }));
      e->div_key = key;
      // the table is shared by later calls: a call on a different stream must be ordered behind this build
      if (e->div_ev == nullptr) HIPCHK(hipEventCreateWithFlags(&e->div_ev, hipEventDisableTiming));
      hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
      (void)hipStreamIsCapturing(s, &cap);
      if (cap == hipStreamCaptureStatusNone) {
        HIPCHK(hipEventRecord(e->div_ev, s));
        e->div_stream = s;
      } else {
        e->div_stream = s;          // built inside a capture: the graph owns the ordering, nothing to wait for outside it
      }
    } else if (s != e->div_stream && e->div_ev != nullptr && hipEventQuery(e->div_ev) == hipErrorNotReady) {
      HIPCHK(hipStreamWaitEvent(s, e->div_ev, 0));
    }
    return timed(e, ASX_PROF_FINALIZE, 0.0, bytes, s, [&]() {
      hipLaunchKernelGGL(finalize4_kernel, dim3((unsigned)((N / 4 + 255) / 256), 2), dim3(256), 0, s, chunk_out_dev, p.n_chunks,
                         p.chunk_size, p.step, p.padded_len, p.trim, N, e->d_div.f(), out_dev);
    });
  }
  return timed(e, ASX_PROF_FINALIZE, 0.0, bytes, s, [&]() {
    hipLaunchKernelGGL(finalize_kernel, dim3((unsigned)((N + 255) / 256), 2), dim3(256), 0, s, chunk_out_dev,
                       p.n_chunks, p.chunk_size, p.step, p.padded_len, p.trim, N, win, out_dev, hann);
  });
}

int asx_demix_dev(asx_engine *e, const float *mix_dev, int64_t N, float *out_dev, uint32_t flags, void *stream) {
  REQUIRE(e && mix_dev && out_dev, "asx_demix_dev: null argument");
  HIPCHK(hipSetDevice(e->device));
  asx_plan p;
  CHK(asx_plan_query(e, N, flags, &p));
  CHK(e->chunk_out.ensure((size_t)p.n_chunks * 2 * p.chunk_size * 4));
  CHK(asx_demix_chunks_dev(e, mix_dev, N, 0, p.n_chunks, e->chunk_out.f(), flags, stream));
  CHK(asx_finalize_dev(e, e->chunk_out.f(), N, out_dev, flags, stream));
  return ASX_OK;
}

int asx_demix(asx_engine *e, const float *mix_host, int64_t N, float *out_host, uint32_t flags) {
  REQUIRE(e && mix_host && out_host, "asx_demix: null argument");
  REQUIRE(N >= 1, "n_samples must be >= 1");
  HIPCHK(hipSetDevice(e->device));
  DevBuf dmix, dout;
  int rc = dmix.ensure((size_t)2 * N * 4);
  if (rc == ASX_OK) rc = dout.ensure((size_t)2 * N * 4);
  if (rc == ASX_OK && hipMemcpy(dmix.p, mix_host, (size_t)2 * N * 4, hipMemcpyHostToDevice) != hipSuccess) {
    set_err("asx_demix: H2D copy failed");
    rc = ASX_ERR_HIP;
  }
  if (rc == ASX_OK) rc = asx_demix_dev(e, dmix.f(), N, dout.f(), flags, nullptr);
  if (rc == ASX_OK && hipStreamSynchronize(nullptr) != hipSuccess) {
    set_err("asx_demix: device execution failed: %s", hipGetErrorString(hipGetLastError()));
    rc = ASX_ERR_HIP;
  }
  if (rc == ASX_OK && hipMemcpy(out_host, dout.p, (size_t)2 * N * 4, hipMemcpyDeviceToHost) != hipSuccess) {
    set_err("asx_demix: D2H copy failed");
    rc = ASX_ERR_HIP;
  }
  dmix.release();
  dout.release();
  return rc;
}

// ---- stem algebra ------------------------------------------------------------
int asx_separate_dev(asx_engine *e, float *mix_dev, int64_t N, float max_peak, float min_peak, int32_t has_min,
                     float compensate, float *primary_dev, float *secondary_dev, void *stream) {
  REQUIRE(e && mix_dev && primary_dev && secondary_dev, "asx_separate_dev: null argument");
  REQUIRE(N >= 1, "n_samples must be >= 1");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  HIPCHK(hipSetDevice(e->device));
  CHK(e->d_peak.ensure(256));
  CHK(e->d_demixed.ensure((size_t)2 * N * 4));
  unsigned int *pk = reinterpret_cast<unsigned int *>(e->d_peak.p);
  HIPCHK(hipMemsetAsync(pk, 0, 4, s));
  const int64_t n2 = 2 * N;
  const unsigned nb = (unsigned)std::min<int64_t>((n2 + 255) / 256, 2048);
  CHK(timed(e, ASX_PROF_MISC, 0.0, 4.0 * n2, s, [&]() {
    hipLaunchKernelGGL(absmax_kernel, dim3(nb), dim3(256), 0, s, mix_dev, n2, pk);
  }));
  CHK(timed(e, ASX_PROF_MISC, 0.0, 8.0 * n2, s, [&]() {
    hipLaunchKernelGGL(normalize_kernel, dim3(nb), dim3(256), 0, s, mix_dev, n2, pk, max_peak, min_peak, has_min);
  }));
  CHK(asx_demix_dev(e, mix_dev, N, e->d_demixed.f(), 0, stream));
  CHK(timed(e, ASX_PROF_MISC, 0.0, 4.0 * 4 * n2, s, [&]() {
    hipLaunchKernelGGL(stems_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, s, e->d_demixed.f(), mix_dev, N,
                       pk, compensate, primary_dev, secondary_dev);
  }));
  return ASX_OK;
}

int asx_separate(asx_engine *e, float *mix_host, int64_t N, float max_peak, float min_peak, int32_t has_min,
                 float compensate, float *primary_host, float *secondary_host) {
  REQUIRE(e && mix_host && primary_host && secondary_host, "asx_separate: null argument");
  REQUIRE(N >= 1, "n_samples must be >= 1");
  HIPCHK(hipSetDevice(e->device));
  DevBuf dmix, dp, ds;
  const size_t bytes = (size_t)2 * N * 4;
  int rc = dmix.ensure(bytes);
  if (rc == ASX_OK) rc = dp.ensure(bytes);
  if (rc == ASX_OK) rc = ds.ensure(bytes);
  if (rc == ASX_OK && hipMemcpy(dmix.p, mix_host, bytes, hipMemcpyHostToDevice) != hipSuccess) {
    set_err("asx_separate: H2D copy failed");
    rc = ASX_ERR_HIP;
  }
  if (rc == ASX_OK)
    rc = asx_separate_dev(e, dmix.f(), N, max_peak, min_peak, has_min, compensate, dp.f(), ds.f(), nullptr);
  if (rc == ASX_OK && hipStreamSynchronize(nullptr) != hipSuccess) {
    set_err("asx_separate: device execution failed: %s", hipGetErrorString(hipGetLastError()));
    rc = ASX_ERR_HIP;
  }
  if (rc == ASX_OK && (hipMemcpy(mix_host, dmix.p, bytes, hipMemcpyDeviceToHost) != hipSuccess ||
                       hipMemcpy(primary_host, dp.p, bytes, hipMemcpyDeviceToHost) != hipSuccess ||
                       hipMemcpy(secondary_host, ds.p, bytes, hipMemcpyDeviceToHost) != hipSuccess)) {
    set_err("asx_separate: D2H copy failed");
    rc = ASX_ERR_HIP;
  }
  dmix.release();
  dp.release();
  ds.release();
  return rc;
}

// spec_utils.normalize(wave, max_peak, min_peak) (uvr_lib_v5/spec_utils.py:99-115) on any float32 array, in place:
// maxv = |wave|.max(); > max_peak: *= max_peak / maxv; else (min_peak given and maxv < min_peak): *= min_peak / maxv.
int asx_normalize(asx_engine *e, float *wave_host, int64_t numel, float max_peak, float min_peak, int32_t has_min, float *peak_before) {
  REQUIRE(e && wave_host && numel >= 1, "asx_normalize: bad argument");
  HIPCHK(hipSetDevice(e->device));
  CHK(e->d_peak.ensure(256));
  DevBuf d;
  int rc = d.ensure((size_t)numel * 4);
  unsigned int *pk = reinterpret_cast<unsigned int *>(e->d_peak.p);
  float maxv = 0.f;
  const unsigned nb = (unsigned)std::min<int64_t>((numel + 255) / 256, 2048);
  if (rc == ASX_OK && (hipMemcpy(d.p, wave_host, (size_t)numel * 4, hipMemcpyHostToDevice) != hipSuccess ||
                       hipMemsetAsync(pk, 0, 4, nullptr) != hipSuccess)) {
    set_err("asx_normalize: H2D copy failed");
    rc = ASX_ERR_HIP;
  }
  if (rc == ASX_OK) {
    hipLaunchKernelGGL(absmax_kernel, dim3(nb), dim3(256), 0, nullptr, d.f(), numel, pk);
    hipLaunchKernelGGL(normalize_kernel, dim3(nb), dim3(256), 0, nullptr, d.f(), numel, pk, max_peak, min_peak, has_min);
    if (hipGetLastError() != hipSuccess || hipMemcpy(&maxv, pk, 4, hipMemcpyDeviceToHost) != hipSuccess ||
        hipMemcpy(wave_host, d.p, (size_t)numel * 4, hipMemcpyDeviceToHost) != hipSuccess) {
      set_err("asx_normalize: device execution failed");
      rc = ASX_ERR_HIP;
    }
  }
  if (peak_before) *peak_before = maxv;
  d.release();
  return rc;
}

// the same on an array that is already in HBM (MDXCSeparator.separate normalises the mix and every stem, mdxc_separator.py:147,
// 170-190): in place, only enqueues work
int asx_normalize_dev(asx_engine *e, float *wave_dev, int64_t numel, float max_peak, float min_peak, int32_t has_min, void *stream) {
  REQUIRE(e && wave_dev && numel >= 1, "asx_normalize_dev: bad argument");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  HIPCHK(hipSetDevice(e->device));
  CHK(e->d_peak.ensure(256));
  unsigned int *pk = reinterpret_cast<unsigned int *>(e->d_peak.p) + 2;   // its own word (0: separate / pcm16, 1: decode)
  HIPCHK(hipMemsetAsync(pk, 0, 4, s));
  const unsigned nb = (unsigned)std::min<int64_t>((numel + 255) / 256, 2048);
  CHK(timed(e, ASX_PROF_MISC, 0.0, 4.0 * numel, s, [&]() { hipLaunchKernelGGL(absmax_kernel, dim3(nb), dim3(256), 0, s, wave_dev, numel, pk); }));
  return timed(e, ASX_PROF_MISC, 0.0, 8.0 * numel, s, [&]() {
    hipLaunchKernelGGL(normalize_kernel, dim3(nb), dim3(256), 0, s, wave_dev, numel, pk, max_peak, min_peak, has_min);
  });
}

// out = mix - stem (the residual stem of a single-target MDXC / Roformer model, mdxc_separator.py:406-468), float32
int asx_residual_dev(asx_engine *e, const float *mix_dev, const float *stem_dev, float *out_dev, int64_t numel, void *stream) {
  REQUIRE(e && mix_dev && stem_dev && out_dev && numel >= 1, "asx_residual_dev: bad argument");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  HIPCHK(hipSetDevice(e->device));
  const unsigned nb = (unsigned)std::min<int64_t>((numel + 255) / 256, 4096);
  return timed(e, ASX_PROF_MISC, 0.0, 12.0 * numel, s, [&]() {
    hipLaunchKernelGGL(residual_kernel, dim3(nb), dim3(256), 0, s, mix_dev, stem_dev, numel, out_dev);
  });
}

// librosa.resample(res_type="sinc_fastest") for an arbitrary ratio (engine_vr.h resample_sinc_dev)
int asx_resample_sinc_dev(asx_engine *e, const float *x_dev, int32_t channels, int64_t n_in, double ratio, int32_t mono_calls,
                          float *y_dev, int64_t n_out, void *stream) {
  REQUIRE(e && x_dev && y_dev && channels >= 1 && channels <= 65535 && n_in >= 1 && n_out >= 1, "asx_resample_sinc_dev: bad argument");
  REQUIRE(ratio > 1.0 / 256 && ratio < 256.0, "asx_resample_sinc_dev: ratio %g outside libsamplerate's (1/256, 256)", ratio);
  HIPCHK(hipSetDevice(e->device));
  return resample_sinc_dev(e, e->sinc_tab, x_dev, channels, n_in, ratio, mono_calls, y_dev, n_out, reinterpret_cast<hipStream_t>(stream));
}

// ---- stage hooks -------------------------------------------------------------
static int to_dev(DevBuf &d, const float *h, size_t n) {
  CHK(d.ensure(n * 4));
  HIPCHK(hipMemcpy(d.p, h, n * 4, hipMemcpyHostToDevice));
  return ASX_OK;
}
static int to_host(float *h, const DevBuf &d, size_t n) {
  HIPCHK(hipDeviceSynchronize());
  HIPCHK(hipMemcpy(h, d.p, n * 4, hipMemcpyDeviceToHost));
  return ASX_OK;
}
struct BufGuard {
  std::vector<DevBuf *> v;
  ~BufGuard() {
    for (auto *b : v) b->release();
  }
};

static int transpose_launch(const float *in, float *out, int planes, int rows, int cols, hipStream_t s) {
  hipLaunchKernelGGL(transpose_last2_kernel, dim3((cols + 31) / 32, (rows + 31) / 32, planes), dim3(32, 8), 0, s, in,
                     out, rows, cols);
  HIPCHK(hipGetLastError());
  return ASX_OK;
}

// write_audio_pydub's array work (common_separator.py:309-337): stem [2, N] planar float32 -> normalised int16 [N, 2];
// *peak_after = max |stem| after normalisation (the caller skips near-silent stems: < 1e-6, :312-315)
int asx_pcm16_dev(asx_engine *e, const float *stem_dev, int64_t N, float max_peak, float min_peak, int32_t has_min, int16_t *pcm_dev,
                  float *peak_after, void *stream) {
  REQUIRE(e && stem_dev && pcm_dev && N >= 1, "asx_pcm16_dev: bad argument");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  HIPCHK(hipSetDevice(e->device));
  CHK(e->d_peak.ensure(256));
  unsigned int *pk = reinterpret_cast<unsigned int *>(e->d_peak.p);
  HIPCHK(hipMemsetAsync(pk, 0, 4, s));
  const int64_t n2 = 2 * N;
  const unsigned nb = (unsigned)std::min<int64_t>((n2 + 255) / 256, 2048);
  CHK(timed(e, ASX_PROF_MISC, 0.0, 4.0 * n2, s, [&]() { hipLaunchKernelGGL(absmax_kernel, dim3(nb), dim3(256), 0, s, stem_dev, n2, pk); }));
  CHK(timed(e, ASX_PROF_MISC, 0.0, 6.0 * n2, s, [&]() {
    hipLaunchKernelGGL(pcm16_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, s, stem_dev, N, pk, max_peak, min_peak, has_min,
                       reinterpret_cast<short *>(pcm_dev));
  }));
  if (peak_after) {
    float maxv = 0.f;
    HIPCHK(hipMemcpyAsync(&maxv, pk, 4, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    float scale = 1.0f;
    if (maxv > max_peak) scale = max_peak / maxv;
    else if (has_min && maxv < min_peak) scale = min_peak / maxv;
    *peak_after = maxv * scale;
  }
  return ASX_OK;
}

int asx_pcm16_rows_dev(asx_engine *e, const float *stem_rows_dev, int64_t N, float max_peak, float min_peak, int32_t has_min,
                       int16_t *pcm_dev, float *peak_after, void *stream) {
  REQUIRE(e && stem_rows_dev && pcm_dev && N >= 1, "asx_pcm16_rows_dev: bad argument");
  REQUIRE((((uintptr_t)stem_rows_dev) & 15) == 0 && (((uintptr_t)pcm_dev) & 7) == 0, "asx_pcm16_rows_dev: stem must be 16-byte and pcm 8-byte aligned");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  HIPCHK(hipSetDevice(e->device));
  CHK(e->d_peak.ensure(256));
  unsigned int *pk = reinterpret_cast<unsigned int *>(e->d_peak.p);
  HIPCHK(hipMemsetAsync(pk, 0, 4, s));
  const int64_t n2 = 2 * N;
  const unsigned nb = (unsigned)std::min<int64_t>((n2 + 255) / 256, 2048);
  CHK(timed(e, ASX_PROF_MISC, 0.0, 4.0 * n2, s, [&]() { hipLaunchKernelGGL(absmax_kernel, dim3(nb), dim3(256), 0, s, stem_rows_dev, n2, pk); }));
  CHK(timed(e, ASX_PROF_MISC, 0.0, 6.0 * n2, s, [&]() {
    hipLaunchKernelGGL(pcm16_rows_kernel, dim3((unsigned)((n2 / 4 + 256) / 256)), dim3(256), 0, s, stem_rows_dev, n2, pk, max_peak, min_peak,
                       has_min, reinterpret_cast<short *>(pcm_dev));
  }));
  if (peak_after) {
    float maxv = 0.f;
    HIPCHK(hipMemcpyAsync(&maxv, pk, 4, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    float scale = 1.0f;
    if (maxv > max_peak) scale = max_peak / maxv;
    else if (has_min && maxv < min_peak) scale = min_peak / maxv;
    *peak_after = maxv * scale;
  }
  return ASX_OK;
}

int asx_pcm_decode_dev(asx_engine *e, const void *raw_dev, int64_t frames, int32_t channels, int32_t sample_format, float *mix_dev,
                       float *peak, void *stream) {
  REQUIRE(e && raw_dev && mix_dev && frames >= 1, "asx_pcm_decode_dev: bad argument");
  REQUIRE(channels >= 1 && channels <= 64, "asx_pcm_decode_dev: %d channels", channels);
  REQUIRE(sample_format == 16 || sample_format == 24 || sample_format == 32 || sample_format == 0x120,
          "asx_pcm_decode_dev: sample format %d (16, 24, 32 = integer PCM bits, 0x120 = IEEE float32)", sample_format);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  HIPCHK(hipSetDevice(e->device));
  CHK(e->d_peak.ensure(256));
  unsigned int *pk = reinterpret_cast<unsigned int *>(e->d_peak.p) + 1;   // its own word: asx_separate_dev uses word 0
  HIPCHK(hipMemsetAsync(pk, 0, 4, s));
  const unsigned nb = (unsigned)std::min<int64_t>((frames + 255) / 256, 4096);
  CHK(timed(e, ASX_PROF_MISC, 0.0, (double)frames * (channels * (sample_format & 0xff) / 8 + 8.0), s, [&]() {
    hipLaunchKernelGGL(pcm_decode_kernel, dim3(nb), dim3(256), 0, s, reinterpret_cast<const unsigned char *>(raw_dev), frames, (int)channels,
                       (int)sample_format, mix_dev, pk);
  }));
  if (peak) {
    HIPCHK(hipMemcpyAsync(peak, pk, 4, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
  }
  return ASX_OK;
}

int asx_pcm16(asx_engine *e, const float *stem_host, int64_t N, float max_peak, float min_peak, int32_t has_min, int16_t *pcm_host,
              float *peak_after) {
  REQUIRE(e && stem_host && pcm_host && N >= 1, "asx_pcm16: bad argument");
  HIPCHK(hipSetDevice(e->device));
  DevBuf ds, dp;
  BufGuard g{{&ds, &dp}};
  CHK(to_dev(ds, stem_host, (size_t)2 * N));
  CHK(dp.ensure((size_t)2 * N * 2));
  CHK(asx_pcm16_dev(e, ds.f(), N, max_peak, min_peak, has_min, reinterpret_cast<int16_t *>(dp.p), peak_after, nullptr));
  HIPCHK(hipDeviceSynchronize());
  HIPCHK(hipMemcpy(pcm_host, dp.p, (size_t)2 * N * 2, hipMemcpyDeviceToHost));
  return ASX_OK;
}

int asx_resample_sinc(asx_engine *e, const float *x_host, int32_t channels, int64_t n_in, double ratio, int32_t mono_calls,
                      float *y_host, int64_t n_out) {
  REQUIRE(e && x_host && y_host && channels >= 1 && n_in >= 1 && n_out >= 1, "asx_resample_sinc: bad argument");
  HIPCHK(hipSetDevice(e->device));
  DevBuf dx, dy;
  BufGuard g{{&dx, &dy}};
  CHK(to_dev(dx, x_host, (size_t)channels * n_in));
  CHK(dy.ensure((size_t)channels * n_out * 4));
  CHK(asx_resample_sinc_dev(e, dx.f(), channels, n_in, ratio, mono_calls, dy.f(), n_out, nullptr));
  return to_host(y_host, dy, (size_t)channels * n_out);
}

int asx_stft(asx_engine *e, const float *wave_host, int32_t B, int64_t C, float *spec_host) {
  REQUIRE(e && wave_host && spec_host && B > 0, "asx_stft: bad argument");
  REQUIRE(C > e->cfg.n_fft / 2, "asx_stft: n_time %lld must exceed n_fft/2 (reflect padding)", (long long)C);
  HIPCHK(hipSetDevice(e->device));
  const int T = (int)(C / e->cfg.hop_length) + 1;
  const size_t ns = (size_t)B * 4 * e->cfg.dim_f * T;
  DevBuf dw, ds;
  BufGuard g{{&dw, &ds}};
  CHK(to_dev(dw, wave_host, (size_t)B * 2 * C));
  CHK(ds.ensure(ns * 4));
  CHK(stft_launch(e, dw.f(), nullptr, -1, B, C, T, ds.f(), 0, 0, 1.0f, nullptr));
  CHK(to_host(spec_host, ds, ns));
  return ASX_OK;
}

int asx_istft(asx_engine *e, const float *spec_host, int32_t B, int32_t T, float *wave_host) {
  REQUIRE(e && spec_host && wave_host && B > 0 && T >= 1, "asx_istft: bad argument");
  HIPCHK(hipSetDevice(e->device));
  const int n = e->cfg.n_fft, hop = e->cfg.hop_length;
  const int64_t C = (int64_t)hop * (T - 1);
  REQUIRE(C > 0, "asx_istft: need at least 2 frames");
  const size_t ns = (size_t)B * 4 * e->cfg.dim_f * T;
  DevBuf dsp, dfr, denv, dout;
  BufGuard g{{&dsp, &dfr, &denv, &dout}};
  CHK(to_dev(dsp, spec_host, ns));
  CHK(dfr.ensure((size_t)B * 2 * T * n * 4));
  CHK(dout.ensure((size_t)B * 2 * C * 4));
  std::vector<float> env;
  host_env(n, hop, T, env, e->cfg.win_length);
  CHK(to_dev(denv, env.data(), env.size()));
  CHK(istft_launch(e, dsp.f(), B, T, 0, 0, dfr.f(), nullptr));
  CHK(ola_launch(e, dfr.f(), denv.f(), nullptr, B, T, C, dout.f(), nullptr));
  CHK(to_host(wave_host, dout, (size_t)B * 2 * C));
  return ASX_OK;
}

int asx_net_forward(asx_engine *e, const float *spec_host, int32_t B, float *out_host) {
  REQUIRE(e && spec_host && out_host && B > 0, "asx_net_forward: bad argument");
  if (!e->net_ready) {
    set_err("asx_net_forward: net weights not committed");
    return ASX_ERR_STATE;
  }
  HIPCHK(hipSetDevice(e->device));
  const int T = e->net.dim_t, Fq = e->net.dim_f, dc = e->net.dim_c;
  REQUIRE(dc == 4, "asx_net_forward: dim_c must be 4");
  const bool den = e->cfg.enable_denoise;
  e->cfg.enable_denoise = 0;  // size the workspace for exactly B
  int rc = ensure_workspace(e, B, true);
  e->cfg.enable_denoise = den;
  CHK(rc);
  const size_t ns = (size_t)B * dc * Fq * T;
  DevBuf din, dout;
  BufGuard g{{&din, &dout}};
  CHK(to_dev(din, spec_host, ns));
  CHK(dout.ensure(ns * 4));
  // reference layout [B,4,F,T] -> engine layout [B,4,T,F]
  CHK(transpose_launch(din.f(), e->spec_in.f(), B * dc, Fq, T, nullptr));
  CHK(net_forward_dev(e, e->spec_in.f(), e->spec_out.f(), B, nullptr));
  CHK(transpose_launch(e->spec_out.f(), dout.f(), B * dc, T, Fq, nullptr));
  CHK(to_host(out_host, dout, ns));
  return ASX_OK;
}

int asx_run_model(asx_engine *e, const float *wave_host, int32_t B, float *out_host, uint32_t flags) {
  REQUIRE(e && wave_host && out_host && B > 0, "asx_run_model: bad argument");
  HIPCHK(hipSetDevice(e->device));
  const bool match = (flags & ASX_FLAG_MATCH_MIX) != 0;
  if (!match && !e->net_ready) {
    set_err("asx_run_model: net weights not committed");
    return ASX_ERR_STATE;
  }
  const int T = e->cfg.segment_size;
  const int64_t C = (int64_t)e->cfg.hop_length * (T - 1);
  CHK(ensure_workspace(e, B, !match));
  DevBuf dw, dout;
  BufGuard g{{&dw, &dout}};
  CHK(to_dev(dw, wave_host, (size_t)B * 2 * C));
  CHK(dout.ensure((size_t)B * 2 * C * 4));
  const size_t spec_elems = (size_t)4 * T * e->cfg.dim_f;
  CHK(stft_launch(e, dw.f(), nullptr, -1, B, C, T, e->spec_in.f(), 1, 3, 1.0f, nullptr));
  const float *spec_final = e->spec_in.f();
  int combine = 0;
  if (!match) {
    int Bn = B;
    if (e->cfg.enable_denoise) {
      CHK(stft_launch(e, dw.f(), nullptr, -1, B, C, T, e->spec_in.f() + (size_t)B * spec_elems, 1, 3, -1.0f, nullptr));
      Bn = 2 * B;
      combine = B;
    }
    CHK(net_forward_dev(e, e->spec_in.f(), e->spec_out.f(), Bn, nullptr));
    spec_final = e->spec_out.f();
  }
  CHK(istft_ola_launch(e, spec_final, B, T, combine, nullptr, C, dout.f(), nullptr));
  CHK(to_host(out_host, dout, (size_t)B * 2 * C));
  return ASX_OK;
}

// ---- single-layer hooks --------------------------------------------------------
int asx_op_conv(asx_engine *e, const char *op, const float *x_host, int32_t B, int32_t cin, int32_t t, int32_t f,
                const float *w_host, const float *b_host, int32_t cout, const float *aux_host, int32_t relu,
                float *y_host) {
  REQUIRE(e && op && x_host && w_host && y_host && B > 0 && cin > 0 && cout > 0 && t > 0 && f > 0,
          "asx_op_conv: bad argument");
  HIPCHK(hipSetDevice(e->device));
  int kind;
  int to = t, fo = f;
  if (!strcmp(op, "conv3x3")) kind = CK_3X3;
  else if (!strcmp(op, "down")) {
    kind = CK_DOWN;
    to = t / 2;
    fo = f / 2;
  } else if (!strcmp(op, "conv1x1")) kind = CK_1X1;
  else if (!strcmp(op, "up")) {
    kind = CK_UP;
    to = 2 * t;
    fo = 2 * f;
    REQUIRE(aux_host, "asx_op_conv(up): skip tensor required");
  } else {
    set_err("asx_op_conv: unknown op '%s'", op);
    return ASX_ERR_INVALID;
  }
  ConvLayer L;
  DevBuf dx, dy, dskip;
  BufGuard g{{&dx, &dy, &dskip, &L.w, &L.b, &L.wu, &L.wu2, &L.wu3, &L.wus, &L.wu6}};
  CHK(conv_setup(L, kind, cin, cout, relu ? 1 : 0));
  CHK(conv_pack(L, w_host, b_host, e->winograd));
  CHK(to_dev(dx, x_host, (size_t)B * cin * t * f));
  const size_t ny = (size_t)B * cout * to * fo;
  CHK(dy.ensure(ny * 4));
  HIPCHK(hipMemset(dy.p, 0xff, ny * 4));  // NaN canary: every output element must be written
  if (kind == CK_UP) CHK(to_dev(dskip, aux_host, ny));
  CHK(conv_launch(e, L, dx.f(), dskip.f(), dy.f(), B, t, f, nullptr));
  CHK(to_host(y_host, dy, ny));
  return ASX_OK;
}

int asx_op_tdf(asx_engine *e, const float *x_host, int32_t B, int32_t c, int32_t t, int32_t k, const float *w_host,
               const float *bias_host, int32_t n, const float *scale_host, const float *shift_host,
               const float *res_host, float *y_host) {
  REQUIRE(e && x_host && w_host && scale_host && shift_host && y_host && B > 0 && c > 0 && t > 0 && k > 0 && n > 0,
          "asx_op_tdf: bad argument");
  HIPCHK(hipSetDevice(e->device));
  TdfLayer L;
  DevBuf dx, dy, dres;
  BufGuard g{{&dx, &dy, &dres, &L.w, &L.bias, &L.scale, &L.shift}};
  CHK(tdf_pack(L, n, k, c, w_host, bias_host, scale_host, shift_host));
  const int64_t M = (int64_t)B * c * t;
  CHK(to_dev(dx, x_host, (size_t)M * k));
  CHK(dy.ensure((size_t)M * n * 4));
  HIPCHK(hipMemset(dy.p, 0xff, (size_t)M * n * 4));
  if (res_host) CHK(to_dev(dres, res_host, (size_t)M * n));
  const int rc = tdf_launch(e, L, dx.f(), res_host ? dres.f() : nullptr, dy.f(), M, t, nullptr);
  const int rc2 = rc == ASX_OK ? to_host(y_host, dy, (size_t)M * n) : rc;   // synchronises: the launch is done with the image
  w3_drop(e, L.w.p);                                   // the temporary layer's split image goes with its weight buffer
  return rc2;
}

// ---- MDXC / TFC-TDF v3 --------------------------------------------------------------
int asx_v3_begin(asx_engine *e, const asx_v3_config *cfg) {
  w3_flush(e);
  REQUIRE(e && cfg, "asx_v3_begin: null argument");
  REQUIRE(cfg->num_channels == 2, "only stereo (num_channels = 2) is supported");
  REQUIRE(cfg->num_subbands >= 1 && e->cfg.dim_f % cfg->num_subbands == 0, "dim_f must be divisible by num_subbands");
  REQUIRE(cfg->num_scales >= 1 && cfg->num_blocks_per_scale >= 1 && cfg->num_channels_model > 0 && cfg->growth >= 0 &&
              cfg->bottleneck_factor >= 1 && cfg->num_targets >= 1,
          "bad TFC-TDF v3 hyper-parameters");
  REQUIRE(cfg->norm == 0 || cfg->norm == 1, "norm must be None (0) or InstanceNorm (1)");
  REQUIRE(cfg->act == 0 || cfg->act == 1, "act must be relu (0) or gelu (1)");
  const int fs = e->cfg.dim_f / cfg->num_subbands;
  REQUIRE(fs % (1 << cfg->num_scales) == 0 && e->cfg.segment_size % (1 << cfg->num_scales) == 0,
          "dim_f / num_subbands and segment_size must be divisible by 2^num_scales");
  REQUIRE((fs >> cfg->num_scales) % cfg->bottleneck_factor == 0, "bottleneck_factor must divide the deepest dim_f");
  if (!e->v3) e->v3 = new V3Net();
  v3_free(*e->v3);
  e->v3->cfg = *cfg;
  e->v3->begun = true;
  e->v3->ws_batch = 0;
  e->host_tensors.clear();
  e->net_begun = true;   // asx_net_set_tensor() is shared
  return ASX_OK;
}

int asx_v3_commit(asx_engine *e) {
  w3_flush(e);
  REQUIRE(e, "asx_v3_commit: null engine");
  if (!e->v3 || !e->v3->begun) {
    set_err("asx_v3_commit before asx_v3_begin");
    return ASX_ERR_STATE;
  }
  HIPCHK(hipSetDevice(e->device));
  V3Net &n = *e->v3;
  const asx_v3_config &cf = n.cfg;
  const int k = cf.num_subbands, dim_c = k * cf.num_channels * 2;
  int c = cf.num_channels_model, f = e->cfg.dim_f / k;
  CHK(v3_load_conv(e, n.first, CK_1X1, "first_conv.weight", dim_c, c, 1));
  n.enc.assign(cf.num_scales, {});
  n.dec.assign(cf.num_scales, {});
  n.ds.assign(cf.num_scales, ConvLayer());
  n.us.assign(cf.num_scales, ConvLayer());
  n.ds_n.assign(cf.num_scales, V3Norm());
  n.us_n.assign(cf.num_scales, V3Norm());
  for (int i = 0; i < cf.num_scales; ++i) {
    const std::string p = "encoder_blocks." + std::to_string(i);
    CHK(v3_load_tfc_tdf(e, n.enc[i], p + ".tfc_tdf", c, c, f));
    CHK(v3_load_norm(e, n.ds_n[i], p + ".downscale.conv.0", c));
    CHK(v3_load_conv(e, n.ds[i], CK_DOWN, p + ".downscale.conv.2.weight", c, c + cf.growth, 4));
    c += cf.growth;
    f /= 2;
  }
  CHK(v3_load_tfc_tdf(e, n.mid, "bottleneck_block", c, c, f));
  for (int i = 0; i < cf.num_scales; ++i) {
    const std::string p = "decoder_blocks." + std::to_string(i);
    CHK(v3_load_norm(e, n.us_n[i], p + ".upscale.conv.0", c));
    CHK(v3_load_conv(e, n.us[i], CK_UP, p + ".upscale.conv.2.weight", c, c - cf.growth, 4));
    c -= cf.growth;
    f *= 2;
    CHK(v3_load_tfc_tdf(e, n.dec[i], p + ".tfc_tdf", 2 * c, c, f));
  }
  CHK(v3_load_conv(e, n.final0, CK_1X1, "final_conv.0.weight", c + dim_c, c, 1));
  CHK(v3_load_conv(e, n.final1, CK_1X1, "final_conv.2.weight", c, cf.num_targets * dim_c, 1));
  e->host_tensors.clear();
  n.ready = true;
  return ASX_OK;
}

double asx_v3_flops(const asx_engine *e, int32_t batch) { return e ? v3_flops(e, batch) : 0.0; }

int asx_v3_forward(asx_engine *e, const float *wave_host, int32_t B, float *out_host) {
  REQUIRE(e && wave_host && out_host && B > 0, "asx_v3_forward: bad argument");
  if (!e->v3 || !e->v3->ready) {
    set_err("asx_v3_forward: weights not committed");
    return ASX_ERR_STATE;
  }
  HIPCHK(hipSetDevice(e->device));
  const int64_t C = (int64_t)e->cfg.hop_length * (e->cfg.segment_size - 1);
  const int S = e->v3->cfg.num_targets;
  DevBuf dw, dout;
  BufGuard g{{&dw, &dout}};
  CHK(to_dev(dw, wave_host, (size_t)B * 2 * C));
  CHK(dout.ensure((size_t)B * S * 2 * C * 4));
  CHK(v3_chunks_dev(e, dw.f(), nullptr, -1, 0, B, dout.f(), nullptr));
  CHK(to_host(out_host, dout, (size_t)B * S * 2 * C));
  return ASX_OK;
}

int asx_mdxc_plan(const asx_engine *e, int64_t N, int32_t overlap, asx_plan *out) {
  REQUIRE(e && out, "asx_mdxc_plan: null argument");
  REQUIRE(N >= 1 && overlap >= 1, "n_samples and overlap must be >= 1");
  asx_plan p{};
  p.n_samples = N;
  p.chunk_size = (int64_t)e->cfg.hop_length * (e->cfg.segment_size - 1);
  p.step = p.chunk_size / overlap;                       // hop_size (mdxc_separator.py:364)
  REQUIRE(p.step >= 1, "overlap larger than chunk_size");
  int64_t r = (N - p.chunk_size) % p.step;               // Python floor-mod (:368)
  if (r < 0) r += p.step;
  p.pad = p.step - r;
  p.trim = (int32_t)(p.chunk_size - p.step);             // zeros in front (:371)
  p.gen_size = p.step;
  p.padded_len = p.trim + N + p.pad + p.chunk_size - p.step;
  p.n_chunks = (int32_t)((p.padded_len - p.chunk_size) / p.step + 1);   // Tensor.unfold (:374)
  p.n_frames = e->cfg.segment_size;
  *out = p;
  return ASX_OK;
}

// chunks [k0, k1) of the unfold loop (mdxc_separator.py:374-392) -> chunk_out [k1-k0, S, 2, chunk]
int asx_mdxc_chunks_dev(asx_engine *e, const float *mix_dev, int64_t N, int32_t overlap, int32_t k0, int32_t k1, float *chunk_out_dev,
                        void *stream) {
  REQUIRE(e && mix_dev && chunk_out_dev, "asx_mdxc_chunks_dev: null argument");
  if (!e->v3 || !e->v3->ready) {
    set_err("asx_mdxc_demix: weights not committed");
    return ASX_ERR_STATE;
  }
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  HIPCHK(hipSetDevice(e->device));
  asx_plan p;
  CHK(asx_mdxc_plan(e, N, overlap, &p));
  REQUIRE(k0 >= 0 && k0 <= k1 && k1 <= p.n_chunks, "chunk range [%d, %d) outside [0, %d)", k0, k1, p.n_chunks);
  if (k1 == k0) return ASX_OK;
  V3Net &n = *e->v3;
  const int S = n.cfg.num_targets;
  const int64_t C = p.chunk_size;
  const int nk = k1 - k0;
  CHK(n.d_starts.ensure((size_t)nk * 16));   // starts | (unused) active lengths, built on the device: no host sync
  hipLaunchKernelGGL(chunk_table_kernel, dim3((unsigned)((nk + 255) / 256)), dim3(256), 0, s, k0, nk, p.step, p.chunk_size,
                     p.padded_len, 0, reinterpret_cast<int64_t *>(n.d_starts.p), reinterpret_cast<int64_t *>(n.d_starts.p) + nk);
  HIPCHK(hipGetLastError());
  const int maxB = e->cfg.max_batch > 0 ? e->cfg.max_batch : 8;
  const int nbatch = (nk + maxB - 1) / maxB;
  const int per = (nk + nbatch - 1) / nbatch;
  for (int j = 0; j < nk; j += per) {
    const int B = std::min(per, nk - j);
    CHK(v3_chunks_dev(e, mix_dev, reinterpret_cast<const int64_t *>(n.d_starts.p) + j, N, p.trim, B,
                      chunk_out_dev + (size_t)j * S * 2 * C, s));
  }
  return ASX_OK;
}

// uniform fold of ALL chunks, / overlap (mdxc_separator.py:246-255, 394-404): chunk_out [n_chunks, S, 2, chunk] -> out [S, 2, N]
int asx_mdxc_finalize_dev(asx_engine *e, const float *chunk_out_dev, int64_t N, int32_t overlap, float *out_dev, void *stream) {
  REQUIRE(e && chunk_out_dev && out_dev, "asx_mdxc_finalize_dev: null argument");
  if (!e->v3 || !e->v3->ready) {
    set_err("asx_mdxc_demix: weights not committed");
    return ASX_ERR_STATE;
  }
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  HIPCHK(hipSetDevice(e->device));
  asx_plan p;
  CHK(asx_mdxc_plan(e, N, overlap, &p));
  const int S = e->v3->cfg.num_targets;
  const int64_t C = p.chunk_size;
  const double bytes = 4.0 * ((double)p.n_chunks * S * 2 * C + (double)S * 2 * N);
  return timed(e, ASX_PROF_FINALIZE, 0.0, bytes, s, [&]() {
    hipLaunchKernelGGL(mdxc_finalize_kernel, dim3((unsigned)((N + 255) / 256), S * 2), dim3(256), 0, s, chunk_out_dev, p.n_chunks, S, C,
                       p.step, (int64_t)p.trim, N, (float)overlap, out_dev);
  });
}

int asx_mdxc_demix_dev(asx_engine *e, const float *mix_dev, int64_t N, int32_t overlap, float *out_dev, void *stream) {
  REQUIRE(e && mix_dev && out_dev, "asx_mdxc_demix_dev: null argument");
  if (!e->v3 || !e->v3->ready) {
    set_err("asx_mdxc_demix: weights not committed");
    return ASX_ERR_STATE;
  }
  asx_plan p;
  CHK(asx_mdxc_plan(e, N, overlap, &p));
  V3Net &n = *e->v3;
  CHK(n.chunk_out.ensure((size_t)p.n_chunks * n.cfg.num_targets * 2 * p.chunk_size * 4));
  CHK(asx_mdxc_chunks_dev(e, mix_dev, N, overlap, 0, p.n_chunks, n.chunk_out.f(), stream));
  return asx_mdxc_finalize_dev(e, n.chunk_out.f(), N, overlap, out_dev, stream);
}

int asx_mdxc_demix(asx_engine *e, const float *mix_host, int64_t N, int32_t overlap, float *out_host) {
  REQUIRE(e && mix_host && out_host, "asx_mdxc_demix: null argument");
  REQUIRE(N >= 1, "n_samples must be >= 1");
  if (!e->v3 || !e->v3->ready) {
    set_err("asx_mdxc_demix: weights not committed");
    return ASX_ERR_STATE;
  }
  HIPCHK(hipSetDevice(e->device));
  const int S = e->v3->cfg.num_targets;
  DevBuf dmix, dout;
  BufGuard g{{&dmix, &dout}};
  CHK(to_dev(dmix, mix_host, (size_t)2 * N));
  CHK(dout.ensure((size_t)S * 2 * N * 4));
  CHK(asx_mdxc_demix_dev(e, dmix.f(), N, overlap, dout.f(), nullptr));
  CHK(to_host(out_host, dout, (size_t)S * 2 * N));
  return ASX_OK;
}

// ---- BS-Roformer ------------------------------------------------------------------
int asx_rof_begin(asx_engine *e, const asx_rof_config *cfg) {
  w3_flush(e);
  REQUIRE(e && cfg, "asx_rof_begin: null argument");
  REQUIRE(cfg->dim_head == 64, "dim_head must be 64 (got %d)", cfg->dim_head);
  REQUIRE(cfg->dim > 0 && cfg->dim % 4 == 0 && cfg->depth >= 1 && cfg->heads >= 1 && cfg->num_stems >= 1 &&
              cfg->time_depth >= 1 && cfg->freq_depth >= 1 && cfg->mlp_expansion_factor >= 1 &&
              cfg->mask_estimator_depth >= 1 && cfg->n_out >= 1,
          "bad BS-Roformer hyper-parameters");
  REQUIRE(cfg->n_bands >= 2 && cfg->n_bands <= 128, "n_bands must be in [2, 128]");
  int sum = 0;
  for (int j = 0; j < cfg->n_bands; ++j) {
    REQUIRE(cfg->freqs_per_bands[j] >= 1, "freqs_per_bands must be positive");
    sum += cfg->freqs_per_bands[j];
  }
  REQUIRE(e->cfg.dim_f == e->cfg.n_fft / 2 + 1, "dim_f must be n_fft/2 + 1 = %d", e->cfg.n_fft / 2 + 1);
  if (cfg->mel) {
    for (int j = 0; j < cfg->n_bands; ++j) {
      REQUIRE(cfg->band_start[j] >= 0 && cfg->band_start[j] + cfg->freqs_per_bands[j] <= e->cfg.dim_f, "mel band %d out of range", j);
      REQUIRE(j == 0 || (cfg->band_start[j] >= cfg->band_start[j - 1] &&
                         cfg->band_start[j] + cfg->freqs_per_bands[j] >= cfg->band_start[j - 1] + cfg->freqs_per_bands[j - 1]),
              "mel bands must be ordered");
    }
  } else {
    REQUIRE(sum == e->cfg.dim_f, "sum(freqs_per_bands) = %d must equal dim_f = n_fft/2 + 1 = %d", sum, e->cfg.dim_f);
  }
  if (!e->rof) e->rof = new RofNet();
  rof_free(*e->rof);
  RofNet &n = *e->rof;
  n.cfg = *cfg;
  n.band_dim.clear();
  n.band_off.clear();
  n.mask_off.clear();
  int off = 0;
  for (int j = 0; j < cfg->n_bands; ++j) {
    n.band_dim.push_back(4 * cfg->freqs_per_bands[j]);   // 2 (complex) * 2 (stereo) * freqs
    n.band_off.push_back(cfg->mel ? 4 * cfg->band_start[j] : off);
    n.mask_off.push_back(off);
    off += 4 * cfg->freqs_per_bands[j];
  }
  n.W = 4 * e->cfg.dim_f;   // width of the per-frame spectrum vector (f s c)
  n.MW = off;               // width of the concatenated band masks (= W unless the bands overlap)
  n.begun = true;
  e->host_tensors.clear();
  e->net_begun = true;
  return ASX_OK;
}

int asx_rof_commit(asx_engine *e) {
  w3_flush(e);
  REQUIRE(e, "asx_rof_commit: null engine");
  if (!e->rof || !e->rof->begun) {
    set_err("asx_rof_commit before asx_rof_begin");
    return ASX_ERR_STATE;
  }
  HIPCHK(hipSetDevice(e->device));
  RofNet &n = *e->rof;
  const asx_rof_config &c = n.cfg;
  const int T = e->cfg.segment_size, Fb = c.n_bands, D = c.dim;
  n.bs_gamma.assign(Fb, DevBuf());
  n.bs_lin.assign(Fb, RofLin());
  for (int j = 0; j < Fb; ++j) {
    const std::string p = "band_split.to_features." + std::to_string(j);
    CHK(rof_upload(e, n.bs_gamma[j], p + ".0.gamma", n.band_dim[j]));
    CHK(rof_load_lin(e, n.bs_lin[j], p + ".1", D, n.band_dim[j], true));
  }
  n.time_l.assign(c.depth, {});
  n.freq_l.assign(c.depth, {});
  for (int i = 0; i < c.depth; ++i) {
    n.time_l[i].assign(c.time_depth, RofLayer());
    n.freq_l[i].assign(c.freq_depth, RofLayer());
    for (int j = 0; j < c.time_depth; ++j)
      CHK(rof_load_layer(e, n.time_l[i][j], "layers." + std::to_string(i) + ".0.layers." + std::to_string(j), T));
    for (int j = 0; j < c.freq_depth; ++j)
      CHK(rof_load_layer(e, n.freq_l[i][j], "layers." + std::to_string(i) + ".1.layers." + std::to_string(j), Fb));
  }
  if (c.mel) {
    n.tnorm_t.assign(c.depth, DevBuf());
    n.tnorm_f.assign(c.depth, DevBuf());
    for (int i = 0; i < c.depth; ++i) {
      CHK(rof_upload(e, n.tnorm_t[i], "layers." + std::to_string(i) + ".0.norm.gamma", D));
      CHK(rof_upload(e, n.tnorm_f[i], "layers." + std::to_string(i) + ".1.norm.gamma", D));
    }
    // bins -> covering band range (bands are ordered, so the covering set is one run of bands)
    const int F = e->cfg.dim_f;
    std::vector<int> jlo(F, Fb), jhi(F, -1), bst(Fb), mo(Fb);
    for (int j = 0; j < Fb; ++j) {
      bst[j] = c.band_start[j];
      mo[j] = n.mask_off[j];
      for (int f = c.band_start[j]; f < c.band_start[j] + c.freqs_per_bands[j]; ++f) {
        jlo[f] = std::min(jlo[f], j);
        jhi[f] = std::max(jhi[f], j);
      }
    }
    for (int f = 0; f < F; ++f) {
      REQUIRE(jhi[f] >= jlo[f], "all frequencies need to be covered by all bands for now (bin %d is not)", f);
      for (int j = jlo[f]; j <= jhi[f]; ++j)
        REQUIRE(f >= c.band_start[j] && f < c.band_start[j] + c.freqs_per_bands[j], "mel bands covering bin %d are not one run", f);
    }
    auto upi = [&](DevBuf &d, const std::vector<int> &v) -> int {
      CHK(d.ensure(v.size() * 4));
      HIPCHK(hipMemcpy(d.p, v.data(), v.size() * 4, hipMemcpyHostToDevice));
      return ASX_OK;
    };
    CHK(upi(n.d_bstart, bst));
    CHK(upi(n.d_moff, mo));
    CHK(upi(n.d_jlo, jlo));
    CHK(upi(n.d_jhi, jhi));
  } else {
    CHK(rof_upload(e, n.final_g, "final_norm.gamma", D));
  }
  const int hid = D * (c.mel ? 4 : c.mlp_expansion_factor);   // mel: MaskEstimator is built with the default factor 4
  const int n_lin = c.mask_estimator_depth + (c.mel ? 1 : 0);  // mel MLP: dims = (in, hidden * depth, out)
  n.mask.assign(c.num_stems, {});
  for (int st = 0; st < c.num_stems; ++st) {
    n.mask[st].assign(Fb, {});
    for (int j = 0; j < Fb; ++j) {
      auto &mlp = n.mask[st][j];
      mlp.assign(n_lin, RofLin());
      int in = D;
      for (int li = 0; li < n_lin; ++li) {
        const int out = (li + 1 == n_lin) ? 2 * n.band_dim[j] : hid;
        CHK(rof_load_lin(e, mlp[li],
                         "mask_estimators." + std::to_string(st) + ".to_freqs." + std::to_string(j) + ".0." +
                             std::to_string(2 * li),
                         out, in, true));
        in = out;
      }
    }
  }
  // Hamming fold window (scipy.signal.windows.hamming(chunk), float64 -> float32, mdxc_separator.py:310)
  const int64_t C = (int64_t)e->cfg.hop_length * (T - 1);
  std::vector<float> w((size_t)C);
  for (int64_t i = 0; i < C; ++i)
    w[i] = (float)(C == 1 ? 1.0 : 0.54 - 0.46 * cos(2.0 * M_PI * (double)i / (double)(C - 1)));
  CHK(n.d_window.ensure((size_t)C * 4));
  HIPCHK(hipMemcpy(n.d_window.p, w.data(), (size_t)C * 4, hipMemcpyHostToDevice));
  e->host_tensors.clear();
  n.ready = true;
  return ASX_OK;
}

double asx_rof_flops(const asx_engine *e, int32_t batch) { return e ? rof_flops(e, batch) : 0.0; }

int asx_rof_forward(asx_engine *e, const float *wave_host, int32_t B, float *out_host) {
  REQUIRE(e && wave_host && out_host && B > 0, "asx_rof_forward: bad argument");
  if (!e->rof || !e->rof->ready) {
    set_err("asx_rof_forward: weights not committed");
    return ASX_ERR_STATE;
  }
  HIPCHK(hipSetDevice(e->device));
  const int64_t C = (int64_t)e->cfg.hop_length * (e->cfg.segment_size - 1);
  const int S = e->rof->cfg.num_stems;
  DevBuf dw, dout;
  BufGuard g{{&dw, &dout}};
  CHK(to_dev(dw, wave_host, (size_t)B * 2 * C));
  CHK(dout.ensure((size_t)B * S * 2 * C * 4));
  CHK(rof_chunks_dev(e, dw.f(), nullptr, -1, B, dout.f(), nullptr));
  CHK(to_host(out_host, dout, (size_t)B * S * 2 * C));
  return ASX_OK;
}

// start of Roformer chunk k0 + t: t * step, the last one re-anchored to N - C (mdxc_separator.py:323-336)
__global__ void rof_starts_kernel(int k0, int nk, int64_t step, int64_t N, int64_t C, int64_t *__restrict__ starts) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= nk) return;
  const int64_t i = (int64_t)(k0 + t) * step;
  starts[t] = i + C > N ? N - C : i;
}

static int rof_starts(asx_engine *e, int64_t N, int64_t step, std::vector<int64_t> &starts) {
  const int64_t C = (int64_t)e->cfg.hop_length * (e->cfg.segment_size - 1);
  REQUIRE(N >= C, "mix (%lld samples) shorter than one chunk (%lld): not supported on the Roformer path", (long long)N, (long long)C);
  REQUIRE(step >= 1 && step <= C, "step must be in [1, chunk_size]");
  starts.clear();
  for (int64_t i = 0; i < N; i += step) starts.push_back(i + C > N ? N - C : i);   // tail re-anchored (:323-336)
  return ASX_OK;
}

int asx_rof_plan(const asx_engine *e, int64_t N, int64_t step, int32_t *n_chunks, int64_t *chunk_size) {
  REQUIRE(e && n_chunks && chunk_size, "asx_rof_plan: null argument");
  std::vector<int64_t> starts;
  CHK(rof_starts(const_cast<asx_engine *>(e), N, step, starts));
  *n_chunks = (int32_t)starts.size();
  *chunk_size = (int64_t)e->cfg.hop_length * (e->cfg.segment_size - 1);
  return ASX_OK;
}

// chunks [k0, k1) of the Roformer loop (mdxc_separator.py:318-336) -> chunk_out [k1-k0, S, 2, chunk]
int asx_rof_chunks_dev(asx_engine *e, const float *mix_dev, int64_t N, int64_t step, int32_t k0, int32_t k1, float *chunk_out_dev,
                       void *stream) {
  REQUIRE(e && mix_dev && chunk_out_dev, "asx_rof_chunks_dev: null argument");
  if (!e->rof || !e->rof->ready) {
    set_err("asx_rof_demix: weights not committed");
    return ASX_ERR_STATE;
  }
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  HIPCHK(hipSetDevice(e->device));
  RofNet &n = *e->rof;
  const int S = n.cfg.num_stems;
  const int64_t C = (int64_t)e->cfg.hop_length * (e->cfg.segment_size - 1);
  std::vector<int64_t> starts;
  CHK(rof_starts(e, N, step, starts));
  REQUIRE(k0 >= 0 && k0 <= k1 && k1 <= (int)starts.size(), "chunk range [%d, %d) outside [0, %d)", k0, k1, (int)starts.size());
  if (k1 == k0) return ASX_OK;
  const int nk = k1 - k0;
  CHK(n.d_starts.ensure((size_t)starts.size() * 8));
  // chunk starts built on the device (like chunk_table_kernel for MDX): the call only enqueues work
  hipLaunchKernelGGL(rof_starts_kernel, dim3((unsigned)((nk + 255) / 256)), dim3(256), 0, s, k0, nk, step, N, C,
                     reinterpret_cast<int64_t *>(n.d_starts.p));
  HIPCHK(hipGetLastError());
  const int maxB = e->cfg.max_batch > 0 ? e->cfg.max_batch : 8;
  const int nbatch = (nk + maxB - 1) / maxB;
  const int per = (nk + nbatch - 1) / nbatch;
  for (int j = 0; j < nk; j += per) {
    const int B = std::min(per, nk - j);
    CHK(rof_chunks_dev(e, mix_dev, reinterpret_cast<const int64_t *>(n.d_starts.p) + j, N, B, chunk_out_dev + (size_t)j * S * 2 * C, s));
  }
  return ASX_OK;
}

// Hamming-weighted fold of ALL chunks / counter.clamp(1e-10) (mdxc_separator.py:310-343)
int asx_rof_finalize_dev(asx_engine *e, const float *chunk_out_dev, int64_t N, int64_t step, float *out_dev, void *stream) {
  REQUIRE(e && chunk_out_dev && out_dev, "asx_rof_finalize_dev: null argument");
  if (!e->rof || !e->rof->ready) {
    set_err("asx_rof_demix: weights not committed");
    return ASX_ERR_STATE;
  }
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  HIPCHK(hipSetDevice(e->device));
  RofNet &n = *e->rof;
  const int S = n.cfg.num_stems;
  const int64_t C = (int64_t)e->cfg.hop_length * (e->cfg.segment_size - 1);
  std::vector<int64_t> starts;
  CHK(rof_starts(e, N, step, starts));
  const int nk = (int)starts.size();
  CHK(n.d_starts.ensure((size_t)nk * 8));
  hipLaunchKernelGGL(rof_starts_kernel, dim3((unsigned)((nk + 255) / 256)), dim3(256), 0, s, 0, nk, step, N, C,
                     reinterpret_cast<int64_t *>(n.d_starts.p));
  HIPCHK(hipGetLastError());
  const int n_out = n.cfg.n_out;
  return timed(e, ASX_PROF_FINALIZE, 0.0, 4.0 * ((double)nk * S * 2 * C + 2.0 * n_out * N), s, [&]() {
    hipLaunchKernelGGL(roformer_finalize_kernel, dim3((unsigned)((N + 255) / 256), n_out * 2), dim3(256), 0, s, chunk_out_dev,
                       reinterpret_cast<const int64_t *>(n.d_starts.p), nk, S, C, n.d_window.f(), N, out_dev);
  });
}

int asx_rof_demix_dev(asx_engine *e, const float *mix_dev, int64_t N, int64_t step, float *out_dev, void *stream) {
  REQUIRE(e && mix_dev && out_dev, "asx_rof_demix_dev: null argument");
  if (!e->rof || !e->rof->ready) {
    set_err("asx_rof_demix: weights not committed");
    return ASX_ERR_STATE;
  }
  RofNet &n = *e->rof;
  std::vector<int64_t> starts;
  CHK(rof_starts(e, N, step, starts));
  const int64_t C = (int64_t)e->cfg.hop_length * (e->cfg.segment_size - 1);
  CHK(n.chunk_out.ensure(starts.size() * n.cfg.num_stems * 2 * C * 4));
  CHK(asx_rof_chunks_dev(e, mix_dev, N, step, 0, (int32_t)starts.size(), n.chunk_out.f(), stream));
  return asx_rof_finalize_dev(e, n.chunk_out.f(), N, step, out_dev, stream);
}

int asx_rof_demix(asx_engine *e, const float *mix_host, int64_t N, int64_t step, float *out_host) {
  REQUIRE(e && mix_host && out_host, "asx_rof_demix: null argument");
  if (!e->rof || !e->rof->ready) {
    set_err("asx_rof_demix: weights not committed");
    return ASX_ERR_STATE;
  }
  HIPCHK(hipSetDevice(e->device));
  const int n_out = e->rof->cfg.n_out;
  DevBuf dmix, dout;
  BufGuard g{{&dmix, &dout}};
  CHK(to_dev(dmix, mix_host, (size_t)2 * N));
  CHK(dout.ensure((size_t)n_out * 2 * N * 4));
  CHK(asx_rof_demix_dev(e, dmix.f(), N, step, dout.f(), nullptr));
  CHK(to_host(out_host, dout, (size_t)n_out * 2 * N));
  return ASX_OK;
}

// ---- options -----------------------------------------------------------------------
// ---- Demucs v4 ---------------------------------------------------------------------
int asx_ht_begin(asx_engine *e, const asx_ht_config *cfg) {
  w3_flush(e);
  REQUIRE(e && cfg, "asx_ht_begin: null argument");
  REQUIRE(cfg->n_sources >= 1 && cfg->channels >= 4 && cfg->growth >= 1 && cfg->depth >= 1 && cfg->depth <= 8,
          "bad HTDemucs hyper-parameters");
  REQUIRE(cfg->kernel_size == 8 && cfg->stride == 4, "only kernel_size 8 / stride 4 is built (got %d / %d)",
          cfg->kernel_size, cfg->stride);
  REQUIRE(cfg->dconv_depth >= 0 && cfg->dconv_depth <= 4 && cfg->dconv_comp >= 1 && cfg->channels % cfg->dconv_comp == 0,
          "bad DConv hyper-parameters");
  REQUIRE(cfg->nfft >= 64 && cfg->nfft % 8 == 0, "bad nfft %d", cfg->nfft);
  REQUIRE(cfg->t_layers >= 0 && (cfg->t_layers == 0 || (cfg->t_heads >= 1 && cfg->t_hidden >= 4 && cfg->t_hidden % 4 == 0)),
          "bad transformer hyper-parameters");
  REQUIRE(cfg->samplerate > 0 && cfg->segment_samples > 0, "bad samplerate / segment");
  hd_drop_clones(e);
  if (e->hd) hd_free(*e->hd);   // a Demucs v3 net shares e->ht: it is gone with these weights
  if (!e->ht) e->ht = new HtNet();
  ht_free(*e->ht);
  e->ht->cfg = *cfg;
  e->ht->begun = true;
  e->host_tensors.clear();
  e->net_begun = true;
  return ASX_OK;
}

int asx_ht_commit(asx_engine *e) {
  w3_flush(e);
  REQUIRE(e, "asx_ht_commit: null engine");
  if (!e->ht || !e->ht->begun) {
    set_err("asx_ht_commit before asx_ht_begin");
    return ASX_ERR_STATE;
  }
  HIPCHK(hipSetDevice(e->device));
  const int rc = ht_commit(e);
  e->host_tensors.clear();
  return rc;
}

double asx_ht_flops(const asx_engine *e) { return (e && e->ht && e->ht->ready) ? ht_flops(e) : 0.0; }

int asx_ht_forward(asx_engine *e, const float *mix_host, int32_t B, int64_t length, float *out_host) {
  REQUIRE(e && mix_host && out_host && B > 0, "asx_ht_forward: bad argument");
  if (!e->ht || !e->ht->ready) {
    set_err("asx_ht_forward: weights not committed");
    return ASX_ERR_STATE;
  }
  HIPCHK(hipSetDevice(e->device));
  const int64_t TL = e->ht->L[0];
  const int S = e->ht->cfg.n_sources;
  REQUIRE(length >= 1 && length <= TL, "length %lld outside [1, %lld]", (long long)length, (long long)TL);
  DevBuf din, dout;
  BufGuard g{{&din, &dout}};
  CHK(din.ensure((size_t)B * 2 * TL * 4));
  CHK(dout.ensure((size_t)B * S * 2 * TL * 4));
  HIPCHK(hipMemset(din.p, 0, (size_t)B * 2 * TL * 4));
  HIPCHK(hipMemcpy2D(din.p, (size_t)TL * 4, mix_host, (size_t)length * 4, (size_t)length * 4, (size_t)B * 2,
                     hipMemcpyHostToDevice));
  CHK(ht_forward_dev(e, din.f(), B, dout.f(), nullptr));
  HIPCHK(hipDeviceSynchronize());
  HIPCHK(hipMemcpy2D(out_host, (size_t)length * 4, dout.p, (size_t)TL * 4, (size_t)length * 4, (size_t)B * S * 2,
                     hipMemcpyDeviceToHost));
  return ASX_OK;
}

int asx_ht_demix_dev(asx_engine *e, const float *mix_dev, int64_t N, int32_t shifts, const int64_t *offsets, double overlap,
                     uint32_t flags, float *out_dev, void *stream) {
  REQUIRE(e && mix_dev && out_dev && N >= 2, "asx_ht_demix_dev: bad argument");
  REQUIRE(shifts >= 0 && (shifts == 0 || offsets), "shifts > 0 needs the offsets array");
  REQUIRE(overlap >= 0.0 && overlap < 1.0, "overlap must be in [0, 1)");
  if (!e->ht || !e->ht->ready) {
    set_err("asx_ht_demix: weights not committed");
    return ASX_ERR_STATE;
  }
  HIPCHK(hipSetDevice(e->device));
  return ht_demix_dev(e, mix_dev, N, shifts, offsets, overlap, flags, out_dev, reinterpret_cast<hipStream_t>(stream));
}

#define HT_READY(fn)                                   \
  do {                                                 \
    if (!e->ht || !e->ht->ready) {                     \
      set_err(fn ": weights not committed");           \
      return ASX_ERR_STATE;                            \
    }                                                  \
  } while (0)

int asx_ht_plan(const asx_engine *e, int64_t N, int32_t shifts, const int64_t *offsets, double overlap, int32_t *n_segments,
                int64_t *segment_samples) {
  REQUIRE(e && n_segments && segment_samples && N >= 2 && shifts >= 0 && (shifts == 0 || offsets), "asx_ht_plan: bad argument");
  HT_READY("asx_ht_plan");
  HtPlan p;
  CHK(ht_plan(e, N, shifts, offsets, overlap, p));
  *n_segments = (int32_t)p.starts.size();
  *segment_samples = p.segment;
  return ASX_OK;
}

int asx_ht_segments_dev(asx_engine *e, const float *mix_dev, int64_t N, int32_t shifts, const int64_t *offsets, double overlap,
                        uint32_t flags, int32_t k0, int32_t k1, float *chunk_out_dev, void *stream) {
  REQUIRE(e && mix_dev && chunk_out_dev && N >= 2 && shifts >= 0 && (shifts == 0 || offsets), "asx_ht_segments_dev: bad argument");
  HT_READY("asx_ht_segments_dev");
  HIPCHK(hipSetDevice(e->device));
  HtPlan p;
  CHK(ht_plan(e, N, shifts, offsets, overlap, p));
  REQUIRE(k0 >= 0 && k0 <= k1 && k1 <= (int)p.starts.size(), "segment range [%d, %d) outside [0, %d)", k0, k1, (int)p.starts.size());
  return ht_segments_dev(e, mix_dev, N, p, flags, k0, k1, chunk_out_dev, reinterpret_cast<hipStream_t>(stream));
}

int asx_ht_fold_dev(asx_engine *e, const float *mix_dev, int64_t N, int32_t shifts, const int64_t *offsets, double overlap, uint32_t flags,
                    const float *chunk_out_dev, float *out_dev, void *stream) {
  REQUIRE(e && mix_dev && chunk_out_dev && out_dev && N >= 2 && shifts >= 0 && (shifts == 0 || offsets), "asx_ht_fold_dev: bad argument");
  HT_READY("asx_ht_fold_dev");
  HIPCHK(hipSetDevice(e->device));
  HtPlan p;
  CHK(ht_plan(e, N, shifts, offsets, overlap, p));
  return ht_fold_dev(e, mix_dev, N, p, flags, chunk_out_dev, out_dev, reinterpret_cast<hipStream_t>(stream));
}

int asx_ht_demix(asx_engine *e, const float *mix_host, int64_t N, int32_t shifts, const int64_t *offsets, double overlap,
                 uint32_t flags, float *out_host) {
  REQUIRE(e && mix_host && out_host && N >= 2, "asx_ht_demix: bad argument");
  if (!e->ht || !e->ht->ready) {
    set_err("asx_ht_demix: weights not committed");
    return ASX_ERR_STATE;
  }
  HIPCHK(hipSetDevice(e->device));
  const int S = e->ht->cfg.n_sources;
  DevBuf dm, dout;
  BufGuard g{{&dm, &dout}};
  CHK(to_dev(dm, mix_host, (size_t)2 * N));
  CHK(dout.ensure((size_t)S * 2 * N * 4));
  CHK(asx_ht_demix_dev(e, dm.f(), N, shifts, offsets, overlap, flags, dout.f(), nullptr));
  CHK(to_host(out_host, dout, (size_t)S * 2 * N));
  return ASX_OK;
}

// ---- Demucs v3 (HDemucs) -------------------------------------------------------------
int asx_hd_begin(asx_engine *e, const asx_hd_config *cfg) {
  w3_flush(e);
  REQUIRE(e && cfg, "asx_hd_begin: null argument");
  REQUIRE(cfg->n_sources >= 1 && cfg->channels >= 4 && cfg->growth >= 1 && cfg->depth >= 3 && cfg->depth <= 8, "bad HDemucs hyper-parameters");
  REQUIRE(cfg->kernel_size == 8 && cfg->stride == 4 && cfg->time_stride == 2,
          "only kernel_size 8 / stride 4 / time_stride 2 is built (got %d / %d / %d)", cfg->kernel_size, cfg->stride, cfg->time_stride);
  REQUIRE(cfg->dconv_depth >= 1 && cfg->dconv_depth <= 4 && cfg->dconv_comp >= 1 && cfg->channels % cfg->dconv_comp == 0, "bad DConv hyper-parameters");
  REQUIRE(cfg->norm_starts == cfg->depth - 2 && cfg->dconv_lstm == cfg->depth - 2 && cfg->dconv_attn == cfg->depth - 2,
          "only the default structure (GroupNorm, BLSTM and LocalState on the two innermost levels: norm_starts = dconv_lstm = dconv_attn = depth - 2) "
          "is built");
  REQUIRE(cfg->norm_groups >= 1, "bad norm_groups");
  REQUIRE(cfg->nfft >= 64 && cfg->nfft % 8 == 0, "bad nfft %d", cfg->nfft);
  REQUIRE(cfg->samplerate > 0 && cfg->segment_samples >= 1, "bad samplerate / segment");
  hd_drop_clones(e);
  if (!e->ht) e->ht = new HtNet();
  ht_free(*e->ht);
  if (!e->hd) e->hd = new HdNet();
  hd_free(*e->hd);
  e->hd->cfg = *cfg;
  e->hd->begun = true;
  asx_ht_config &h = e->ht->cfg;
  h = asx_ht_config{};
  h.n_sources = cfg->n_sources;
  h.channels = cfg->channels;
  h.growth = cfg->growth;
  h.nfft = cfg->nfft;
  h.depth = cfg->depth - 2;
  h.kernel_size = cfg->kernel_size;
  h.stride = cfg->stride;
  h.dconv_depth = cfg->dconv_depth;
  h.dconv_comp = cfg->dconv_comp;
  h.samplerate = cfg->samplerate;
  h.segment_samples = cfg->segment_samples;
  h.freq_emb_scale = cfg->freq_emb_scale;
  h.max_batch = cfg->max_batch;
  e->host_tensors.clear();
  e->net_begun = true;
  return ASX_OK;
}

int asx_hd_commit(asx_engine *e) {
  w3_flush(e);
  REQUIRE(e, "asx_hd_commit: null engine");
  if (!e->hd || !e->hd->begun) {
    set_err("asx_hd_commit before asx_hd_begin");
    return ASX_ERR_STATE;
  }
  HIPCHK(hipSetDevice(e->device));
  const int rc = hd_commit(e);
  e->host_tensors.clear();
  return rc;
}

#define HD_READY(fn)                                   \
  do {                                                 \
    if (!e->hd || !e->hd->ready) {                     \
      set_err(fn ": weights not committed");           \
      return ASX_ERR_STATE;                            \
    }                                                  \
  } while (0)

double asx_hd_flops(const asx_engine *e, int64_t length) { return (e && e->hd && e->hd->ready && length > 0) ? hd_flops(e, length) : 0.0; }

int asx_hd_forward(asx_engine *e, const float *mix_host, int32_t B, int64_t length, float *out_host) {
  REQUIRE(e && mix_host && out_host && B > 0 && length > 0, "asx_hd_forward: bad argument");
  HD_READY("asx_hd_forward");
  HIPCHK(hipSetDevice(e->device));
  const int S = e->hd->cfg.n_sources;
  DevBuf din, dout;
  BufGuard g{{&din, &dout}};
  CHK(to_dev(din, mix_host, (size_t)B * 2 * length));
  CHK(dout.ensure((size_t)B * S * 2 * length * 4));
  CHK(hd_forward_dev(e, din.f(), B, length, dout.f(), nullptr));
  HIPCHK(hipDeviceSynchronize());
  return to_host(out_host, dout, (size_t)B * S * 2 * length);
}

int asx_hd_demix_dev(asx_engine *e, const float *mix_dev, int64_t N, int32_t shifts, const int64_t *offsets, double overlap, uint32_t flags,
                     float *out_dev, void *stream) {
  REQUIRE(e && mix_dev && out_dev && N >= 2, "asx_hd_demix_dev: bad argument");
  REQUIRE(shifts >= 0 && (shifts == 0 || offsets), "shifts > 0 needs the offsets array");
  REQUIRE(overlap >= 0.0 && overlap < 1.0, "overlap must be in [0, 1)");
  HD_READY("asx_hd_demix_dev");
  HIPCHK(hipSetDevice(e->device));
  return hd_demix_dev(e, mix_dev, N, shifts, offsets, overlap, flags, out_dev, reinterpret_cast<hipStream_t>(stream));
}

int asx_hd_demix(asx_engine *e, const float *mix_host, int64_t N, int32_t shifts, const int64_t *offsets, double overlap, uint32_t flags,
                 float *out_host) {
  REQUIRE(e && mix_host && out_host && N >= 2, "asx_hd_demix: bad argument");
  HD_READY("asx_hd_demix");
  HIPCHK(hipSetDevice(e->device));
  const int S = e->hd->cfg.n_sources;
  DevBuf dm, dout;
  BufGuard g{{&dm, &dout}};
  CHK(to_dev(dm, mix_host, (size_t)2 * N));
  CHK(dout.ensure((size_t)S * 2 * N * 4));
  CHK(asx_hd_demix_dev(e, dm.f(), N, shifts, offsets, overlap, flags, dout.f(), nullptr));
  return to_host(out_host, dout, (size_t)S * 2 * N);
}

int asx_hd_plan(const asx_engine *e, int64_t N, int32_t shifts, const int64_t *offsets, double overlap, int32_t *n_segments,
                int64_t *segment_samples) {
  REQUIRE(e && n_segments && segment_samples && N >= 2 && shifts >= 0 && (shifts == 0 || offsets), "asx_hd_plan: bad argument");
  HD_READY("asx_hd_plan");
  HdPlan p;
  CHK(hd_plan(e, N, shifts, offsets, overlap, p));
  *n_segments = (int32_t)p.starts.size();
  *segment_samples = p.segment;
  return ASX_OK;
}

int asx_hd_segments_dev(asx_engine *e, const float *mix_dev, int64_t N, int32_t shifts, const int64_t *offsets, double overlap, uint32_t flags,
                        int32_t k0, int32_t k1, float *chunk_out_dev, void *stream) {
  REQUIRE(e && mix_dev && chunk_out_dev && N >= 2 && shifts >= 0 && (shifts == 0 || offsets), "asx_hd_segments_dev: bad argument");
  HD_READY("asx_hd_segments_dev");
  HIPCHK(hipSetDevice(e->device));
  HdPlan p;
  CHK(hd_plan(e, N, shifts, offsets, overlap, p));
  REQUIRE(k0 >= 0 && k0 <= k1 && k1 <= (int)p.starts.size(), "segment range [%d, %d) outside [0, %d)", k0, k1, (int)p.starts.size());
  return hd_segments_dev(e, mix_dev, N, p, flags, k0, k1, chunk_out_dev, reinterpret_cast<hipStream_t>(stream));
}

int asx_hd_fold_dev(asx_engine *e, const float *mix_dev, int64_t N, int32_t shifts, const int64_t *offsets, double overlap, uint32_t flags,
                    const float *chunk_out_dev, float *out_dev, void *stream) {
  REQUIRE(e && mix_dev && chunk_out_dev && out_dev && N >= 2 && shifts >= 0 && (shifts == 0 || offsets), "asx_hd_fold_dev: bad argument");
  HD_READY("asx_hd_fold_dev");
  HIPCHK(hipSetDevice(e->device));
  HdPlan p;
  CHK(hd_plan(e, N, shifts, offsets, overlap, p));
  return hd_fold_dev(e, mix_dev, N, p, flags, chunk_out_dev, out_dev, reinterpret_cast<hipStream_t>(stream));
}

// ---- VR ------------------------------------------------------------------------------
int asx_vr_begin(asx_engine *e, const asx_vr_config *cfg) {
  w3_flush(e);
  REQUIRE(e && cfg, "asx_vr_begin: null argument");
  REQUIRE(cfg->n_bands >= 1 && cfg->n_bands <= 8 && cfg->bins >= 32, "bad band layout");
  REQUIRE(cfg->channel_mode >= 0 && cfg->channel_mode <= 3, "bad channel_mode");
  for (int i = 0; i < (cfg->v51 ? 2 : 5); ++i) REQUIRE(cfg->cap[i] >= 4 && cfg->cap[i] % 4 == 0, "net widths must be multiples of 4");
  REQUIRE(cfg->window_size >= 16 && cfg->offset >= 0, "bad window_size / offset");
  for (int d = 0; d < cfg->n_bands; ++d)
    REQUIRE(cfg->band[d].sr > 0 && cfg->band[d].hl > 0 && cfg->band[d].n_fft >= 8 && cfg->band[d].n_fft % 2 == 0, "band %d: bad sr / hl / n_fft",
            d + 1);
  if (!e->vr) e->vr = new VrNet();
  vr_free(*e->vr);
  e->vr->cfg = *cfg;
  e->vr->begun = true;
  e->host_tensors.clear();
  e->net_begun = true;
  return ASX_OK;
}

int asx_vr_commit(asx_engine *e) {
  w3_flush(e);
  REQUIRE(e, "asx_vr_commit: null engine");
  if (!e->vr || !e->vr->begun) {
    set_err("asx_vr_commit before asx_vr_begin");
    return ASX_ERR_STATE;
  }
  HIPCHK(hipSetDevice(e->device));
  const int rc = vr_commit(e);
  e->host_tensors.clear();
  return rc;
}

double asx_vr_flops(const asx_engine *e) { return (e && e->vr && e->vr->ready) ? vr_flops_patch(e) : 0.0; }

#define VR_READY(fn)                                   \
  do {                                                 \
    if (!e->vr || !e->vr->ready) {                     \
      set_err(fn ": weights not committed");           \
      return ASX_ERR_STATE;                            \
    }                                                  \
  } while (0)

int asx_vr_plan(const asx_engine *e, int64_t n_samples, int32_t *n_frames, int64_t *n_out) {
  REQUIRE(e && n_frames && n_out && n_samples > 0, "asx_vr_plan: bad argument");
  VR_READY("asx_vr_plan");
  int T;
  CHK(vr_plan(*e->vr, n_samples, &T, n_out));
  *n_frames = T;
  return ASX_OK;
}

int asx_vr_forward(asx_engine *e, const float *x_host, int32_t B, float *out_host) {
  REQUIRE(e && x_host && out_host && B > 0, "asx_vr_forward: bad argument");
  VR_READY("asx_vr_forward");
  HIPCHK(hipSetDevice(e->device));
  VrNet &n = *e->vr;
  const int W = n.cfg.window_size;
  const size_t numel = (size_t)B * 2 * n.nb1 * W;
  DevBuf dx, dy;
  BufGuard g{{&dx, &dy}};
  CHK(to_dev(dx, x_host, numel));
  CHK(dy.ensure(numel * 4));
  CHK(n.cfg.v51 ? vr51_ensure_workspace(e, B) : vr_ensure_workspace(e, B));
  const int64_t P = (int64_t)B * n.max_bin * W;
  hipLaunchKernelGGL(vr_from_nchw_kernel, dim3((unsigned)((P + 255) / 256)), dim3(256), 0, nullptr, dx.f(), n.nb1, n.max_bin, W, n.ctot,
                     n.b.hc, P);
  HIPCHK(hipGetLastError());
  CHK(n.cfg.v51 ? vr51_net_dev(e, B, nullptr) : vr_net_dev(e, B, nullptr));
  hipLaunchKernelGGL(vr_to_nchw_kernel, dim3((unsigned)((numel + 255) / 256)), dim3(256), 0, nullptr, n.b.mk, n.nb1, n.max_bin, W, dy.f(),
                     (int64_t)numel);
  HIPCHK(hipGetLastError());
  CHK(to_host(out_host, dy, numel));
  return ASX_OK;
}

int asx_vr_analysis(asx_engine *e, const float *wave_host, int64_t n_samples, float *spec_host) {
  REQUIRE(e && wave_host && spec_host && n_samples > 0, "asx_vr_analysis: bad argument");
  VR_READY("asx_vr_analysis");
  HIPCHK(hipSetDevice(e->device));
  VrNet &n = *e->vr;
  int T;
  int64_t n_out;
  CHK(vr_plan(n, n_samples, &T, &n_out));
  DevBuf dw;
  BufGuard g{{&dw}};
  CHK(to_dev(dw, wave_host, (size_t)2 * n_samples));
  n.he_n = 0;
  CHK(vr_analysis_dev(e, dw.f(), n_samples, T, nullptr));
  CHK(to_host(spec_host, n.X, (size_t)2 * T * n.nb1 * 2));
  return ASX_OK;
}

int asx_vr_separate_dev(asx_engine *e, const float *wave_dev, int64_t n_samples, const asx_vr_params *params, float *primary_dev,
                        float *secondary_dev, void *stream) {
  REQUIRE(e && wave_dev && params && n_samples > 0, "asx_vr_separate_dev: bad argument");
  VR_READY("asx_vr_separate");
  HIPCHK(hipSetDevice(e->device));
  return vr_separate_dev(e, wave_dev, n_samples, params, primary_dev, secondary_dev, reinterpret_cast<hipStream_t>(stream));
}

int asx_vr_separate(asx_engine *e, const float *wave_host, int64_t n_samples, const asx_vr_params *params, float *primary_host,
                    float *secondary_host) {
  REQUIRE(e && wave_host && params && n_samples > 0, "asx_vr_separate: bad argument");
  VR_READY("asx_vr_separate");
  HIPCHK(hipSetDevice(e->device));
  int T;
  int64_t n_out;
  CHK(vr_plan(*e->vr, n_samples, &T, &n_out));
  DevBuf dw, dp, ds;
  BufGuard g{{&dw, &dp, &ds}};
  CHK(to_dev(dw, wave_host, (size_t)2 * n_samples));
  if (primary_host) CHK(dp.ensure((size_t)2 * n_out * 4));
  if (secondary_host) CHK(ds.ensure((size_t)2 * n_out * 4));
  CHK(vr_separate_dev(e, dw.f(), n_samples, params, primary_host ? dp.f() : nullptr, secondary_host ? ds.f() : nullptr, nullptr));
  if (primary_host) CHK(to_host(primary_host, dp, (size_t)2 * n_out));
  if (secondary_host) CHK(to_host(secondary_host, ds, (size_t)2 * n_out));
  return ASX_OK;
}

// launch counters of the bf16 x 6 kernels since the process started (tests assert that the path under test is the one that ran)
int asx_counter(const asx_engine *e, const char *name, int64_t *out) {
  REQUIRE(e && name && out, "asx_counter: null argument");
  const std::string nm(name);
  if (nm == "tdf3_launches") *out = (int64_t)g_tdf3_launches.load();
  else if (nm == "attn6_launches") *out = (int64_t)g_attn6_launches.load();
  else if (nm == "tdf3_gather_launches") *out = (int64_t)g_tdf3_gather_launches.load();
  else if (nm == "wino6_launches") *out = (int64_t)g_wino6_launches.load();
  else {
    set_err("asx_counter: unknown counter '%s'", name);
    return ASX_ERR_INVALID;
  }
  return ASX_OK;
}

// debug hook: copy `numel` floats of a named engine workspace buffer to the host (tests / bring-up only)
int asx_debug_trace(uint64_t *host, int64_t n_u64) {
  REQUIRE(host && n_u64 > 0 && n_u64 <= 4096 * 8, "asx_debug_trace: bad argument");
  HIPCHK(hipDeviceSynchronize());
  HIPCHK(hipMemcpyFromSymbol(host, HIP_SYMBOL(asx_dbg_trace), (size_t)n_u64 * 8));
  return ASX_OK;
}

int asx_debug_fetch(asx_engine *e, const char *name, float *host, int64_t numel) {
  REQUIRE(e && name && host && numel > 0, "asx_debug_fetch: bad argument");
  HIPCHK(hipSetDevice(e->device));
  const float *src = nullptr;
  const std::string nm(name);
  if (e->vr && e->vr->ws_batch > 0) {
    auto &b = e->vr->b;
    if (nm == "vr.hc") src = b.hc;
    else if (nm == "vr.y2") src = b.y2;
    else if (nm == "vr.y3") src = b.y3;
    else if (nm == "vr.h3") src = b.h3;
    else if (nm == "vr.mk") src = b.mk;
    else if (nm == "vr.cat") src = b.cat;
    else if (nm == "vr.bn") src = b.bn;
    else if (nm == "vr.pool") src = b.pool;
    else if (nm == "vr.pool2") src = b.pool2;
    else if (nm == "vr.tmp") src = b.tmp;
    else if (nm.size() == 5 && nm.compare(0, 3, "vr.") == 0 && nm[4] >= '0' && nm[4] < '0' + (int)b.D.size()) {
      const int i = nm[4] - '0';
      src = nm[3] == 'D' ? b.D[i] : (nm[3] == 'E' ? b.E[i] : (nm[3] == 'O' ? b.O[i] : nullptr));
    }
  }
  if (e->hd && e->hd->ws_batch > 0 && e->ht && nm.compare(0, 3, "hd.") == 0) {
    auto &w = e->hd->b;
    auto &b = e->ht->b;
    const int D = e->hd->D;
    const int i = nm.size() == 7 ? nm[6] - '0' : -1;
    if (nm == "hd.inj") src = w.inj;
    else if (nm == "hd.ya") src = w.ya;
    else if (nm == "hd.yz") src = w.yz;
    else if (nm == "hd.skA") src = w.skA;
    else if (nm == "hd.skZ") src = w.skZ;
    else if (nm == "hd.dAin") src = w.dAin;
    else if (nm == "hd.pre") src = w.pre;
    else if (nm.compare(0, 6, "hd.skf") == 0 && i >= 0 && i < D) src = b.skf[i];
    else if (nm.compare(0, 6, "hd.skt") == 0 && i >= 0 && i < D) src = b.skt[i];
    else if (nm.compare(0, 6, "hd.df_") == 0 && i >= 0 && i <= D) src = b.df[i];
    else if (nm.compare(0, 6, "hd.dt_") == 0 && i >= 0 && i <= D) src = b.dt[i];
  }
  if (!src) {
    set_err("asx_debug_fetch: unknown buffer '%s'", name);
    return ASX_ERR_INVALID;
  }
  HIPCHK(hipDeviceSynchronize());
  HIPCHK(hipMemcpy(host, src, (size_t)numel * 4, hipMemcpyDeviceToHost));
  return ASX_OK;
}

// ---- spectral edges ------------------------------------------------------------------
int asx_ensemble(asx_engine *e, const float *waves_host, int32_t K, int64_t N, int32_t algorithm, const double *weights,
                 float *out_host, int64_t *n_out) {
  REQUIRE(e && waves_host && out_host && n_out && N >= 1, "asx_ensemble: bad argument");
  HIPCHK(hipSetDevice(e->device));
  CHK(ens_ctx(e));
  EnsCtx &c = *e->ens;
  CHK(c.din.ensure((size_t)K * 2 * N * 4));
  CHK(c.dout.ensure((size_t)2 * N * 4));
  HIPCHK(hipMemcpy(c.din.p, waves_host, (size_t)K * 2 * N * 4, hipMemcpyHostToDevice));
  CHK(ens_ensemble_dev(e, c.din.f(), K, N, algorithm, weights, c.dout.f(), n_out, nullptr));
  HIPCHK(hipDeviceSynchronize());
  HIPCHK(hipMemcpy(out_host, c.dout.p, (size_t)2 * (*n_out) * 4, hipMemcpyDeviceToHost));
  return ASX_OK;
}

int asx_ensemble_dev(asx_engine *e, const float *waves_dev, int32_t K, int64_t N, int32_t algorithm, const double *weights,
                     float *out_dev, int64_t *n_out, void *stream) {
  REQUIRE(e && waves_dev && out_dev && n_out && N >= 1, "asx_ensemble_dev: bad argument");
  HIPCHK(hipSetDevice(e->device));
  return ens_ensemble_dev(e, waves_dev, K, N, algorithm, weights, out_dev, n_out, reinterpret_cast<hipStream_t>(stream));
}

int asx_invert_stem(asx_engine *e, const float *mix_host, const float *stem_host, int64_t N, float *out_host, int64_t *n_out) {
  REQUIRE(e && mix_host && stem_host && out_host && n_out && N >= 1, "asx_invert_stem: bad argument");
  HIPCHK(hipSetDevice(e->device));
  CHK(ens_ctx(e));
  EnsCtx &c = *e->ens;
  CHK(c.din.ensure((size_t)2 * N * 4));
  CHK(c.din2.ensure((size_t)2 * N * 4));
  CHK(c.dout.ensure((size_t)2 * N * 4));
  HIPCHK(hipMemcpy(c.din.p, mix_host, (size_t)2 * N * 4, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(c.din2.p, stem_host, (size_t)2 * N * 4, hipMemcpyHostToDevice));
  CHK(ens_invert_dev(e, c.din.f(), c.din2.f(), N, c.dout.f(), n_out, nullptr));
  HIPCHK(hipDeviceSynchronize());
  HIPCHK(hipMemcpy(out_host, c.dout.p, (size_t)2 * (*n_out) * 4, hipMemcpyDeviceToHost));
  return ASX_OK;
}

// ---- BagOfModels combine on the device (every member keeps its own engine / weights; no float stem visits the host) -------
int asx_ht_standardize_dev(asx_engine *e, const float *mix_dev, int64_t N, float *out_dev, void *stream) {
  REQUIRE(e && mix_dev && out_dev && N >= 2, "asx_ht_standardize_dev: bad argument");
  REQUIRE(e->ht != nullptr, "asx_ht_standardize_dev: no Demucs model committed on this engine");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  HIPCHK(hipSetDevice(e->device));
  CHK(ht_ref_stats(e, mix_dev, N, s));
  return timed(e, ASX_PROF_MISC, 0.0, 16.0 * N, s, [&]() {
    hipLaunchKernelGGL(ht_standardize_kernel, dim3((unsigned)((2 * N + 255) / 256)), dim3(256), 0, s, mix_dev, 2 * N, N,
                       reinterpret_cast<const double *>(e->ht->ref_acc.p), out_dev);
  });
}

int asx_ht_bag_accumulate_dev(asx_engine *e, float *est_dev, const float *member_dev, const float *weights, int32_t S, int64_t N,
                              int32_t first, void *stream) {
  REQUIRE(e && est_dev && member_dev && weights && S >= 1 && S <= 16 && N >= 1, "asx_ht_bag_accumulate_dev: bad argument");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  HIPCHK(hipSetDevice(e->device));
  BagWeights w{};
  for (int i = 0; i < S; ++i) w.v[i] = weights[i];
  const int64_t per = 2 * N;
  return timed(e, ASX_PROF_MISC, 0.0, 12.0 * S * per, s, [&]() {
    hipLaunchKernelGGL(ht_bag_accumulate_kernel, dim3((unsigned)((per + 255) / 256), (unsigned)S), dim3(256), 0, s, est_dev, member_dev, per, w,
                       (int)first);
  });
}

int asx_ht_bag_finish_dev(asx_engine *e, const float *est_dev, const float *totals, int32_t S, const float *mix_dev, int64_t N,
                          uint32_t flags, float *out_dev, void *stream) {
  REQUIRE(e && est_dev && totals && out_dev && S >= 1 && S <= 16 && N >= 2, "asx_ht_bag_finish_dev: bad argument");
  REQUIRE(est_dev != out_dev, "asx_ht_bag_finish_dev: est and out must be different buffers (the stem swap is not in place)");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  HIPCHK(hipSetDevice(e->device));
  const int standardize = (flags & ASX_HT_STANDARDIZE) ? 1 : 0;
  if (standardize) {
    REQUIRE(e->ht != nullptr && mix_dev != nullptr, "asx_ht_bag_finish_dev: de-standardising needs the mix and a committed Demucs model");
    CHK(ht_ref_stats(e, mix_dev, N, s));
  }
  BagWeights t{};
  for (int i = 0; i < S; ++i) t.v[i] = totals[i];
  const int64_t per = 2 * N;
  return timed(e, ASX_PROF_MISC, 0.0, 8.0 * S * per, s, [&]() {
    hipLaunchKernelGGL(ht_bag_finish_kernel, dim3((unsigned)((per + 255) / 256), (unsigned)S), dim3(256), 0, s, est_dev, per, N, t,
                       standardize ? reinterpret_cast<const double *>(e->ht->ref_acc.p) : nullptr, standardize,
                       (flags & ASX_HT_SWAP01) ? 1 : 0, out_dev);
  });
}

int asx_set_option(asx_engine *e, const char *key, int32_t value) {
  REQUIRE(e && key, "asx_set_option: null argument");
  if (!strcmp(key, "winograd")) {
    e->winograd = value < 0 ? 0 : (int)value;
    return ASX_OK;
  }
  if (!strcmp(key, "winograd_stationary")) {
    e->winos = value < 0 ? 0 : (int)value;
    return ASX_OK;
  }
  if (!strcmp(key, "winograd_bf16x6")) {
    e->wino6 = value < 0 ? 0 : (int)value;
    return ASX_OK;
  }
  if (!strcmp(key, "gemm_bf16x6")) {                 // this engine only (round 5; it was process-wide before)
    e->gemm_bf16x6 = value > 0 ? 1 : 0;
    return ASX_OK;
  }
  set_err("asx_set_option: unknown option '%s'", key);
  return ASX_ERR_INVALID;
}

// ---- profiling -------------------------------------------------------------------
int asx_profile_enable(asx_engine *e, int32_t on) {
  REQUIRE(e, "asx_profile_enable: null engine");
  for (auto &r : e->recs) {
    (void)hipEventDestroy(r.a);
    (void)hipEventDestroy(r.b);
  }
  e->recs.clear();
  e->prof = on != 0;
  return ASX_OK;
}

int asx_profile_read(asx_engine *e, asx_profile *out) {
  REQUIRE(e && out, "asx_profile_read: null argument");
  HIPCHK(hipSetDevice(e->device));
  HIPCHK(hipDeviceSynchronize());
  memset(out, 0, sizeof(*out));
  static const bool dump = getenv("ASX_PROF_DUMP") != nullptr;   // one line per launch on stderr (tuning aid)
  for (auto &r : e->recs) {
    float ms = 0.f;
    HIPCHK(hipEventElapsedTime(&ms, r.a, r.b));
    if (dump) fprintf(stderr, "asx_prof cls=%d ms=%.4f gflop=%.3f mbytes=%.2f\n", r.cls, ms, r.flops * 1e-9, r.bytes * 1e-6);
    out->launches[r.cls] += 1;
    out->ms[r.cls] += ms;
    out->flops[r.cls] += r.flops;
    out->bytes[r.cls] += r.bytes;
  }
  return ASX_OK;
}

int asx_profile_launches(asx_engine *e, asx_launch_rec *out, int32_t cap, int32_t *n) {
  REQUIRE(e && n && (out || cap == 0), "asx_profile_launches: null argument");
  HIPCHK(hipSetDevice(e->device));
  HIPCHK(hipDeviceSynchronize());
  *n = (int32_t)e->recs.size();
  for (int32_t i = 0; i < *n && i < cap; ++i) {
    float ms = 0.f;
    HIPCHK(hipEventElapsedTime(&ms, e->recs[i].a, e->recs[i].b));
    out[i].cls = e->recs[i].cls;
    out[i].ms = ms;
    out[i].flops = e->recs[i].flops;
    out[i].bytes = e->recs[i].bytes;
  }
  return ASX_OK;
}

}  // extern "C"
