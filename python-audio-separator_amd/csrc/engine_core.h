// Engine core of libasx.so (included by asx.hip only, one translation unit): error reporting, device buffers, the packed layer
// structures of the ConvTDFNet path, the engine object, the profiling wrapper and the FFT plan / table builders.  Split out of
// asx.hip in round 5 (it had grown to 3,400 lines); nothing here is visible outside the library -- the boundary is include/asx.h.
#pragma once

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
  DevBuf wu6h;     // the same for the fp16 x 3 arithmetic (wino6_pack_h): two fp16 parts per U, scaled per (position, cout); exponents behind the fragments
  DevBuf w3h;      // conv3h_kernel (kernels_conv3h.h): the 48 -> 48 layers' two-part fp16 weights in fragment order, one exponent per output channel
  DevBuf wd6;      // CK_DOWN: conv_down6_kernel (kernels_updown6.h): the weights split three ways into bf16, fragment order [CG48][stage of 8 channels][part][n][lane][8]
  int wd6_cg = 0, wd6_nst = 0, wd6_nrep = 3;
  DevBuf wup6;     // CK_UP: conv_up6_kernel (kernels_updown6.h): [CG][stage of 32 channels][part][n][lane][8 bf16] over the 4 Cout virtual channels
  int wup6_cg = 0, wup6_nst = 0;
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
  int nprod;        // 16-bit MFMA products per multiply-add the launch executed: 6 (bf16 x 6), 3 (fp16 x 3), 0 = fp32 MFMA / VALU
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
  int kind;                                            // 0: bf16 x 3 parts; 1: fp16 x 2 parts + one exponent per group of four output columns
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
  std::vector<float> custom_window;  // asx_set_stft_window: analysis / synthesis window [n_fft] replacing the periodic Hann (empty = Hann)
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
  DevBuf spec_in, spec_out, R[3], H, HE, frames, chunk_out, d_starts, d_nact, d_peak, d_demixed;
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
#ifdef ASX_EXPERIMENTAL_KERNELS
  int winograd = getenv("ASX_WINOGRAD") ? std::max(0, atoi(getenv("ASX_WINOGRAD"))) : 3;
#else   // generations 1 / 2 are not in this build: anything but 0 means the default
  int winograd = getenv("ASX_WINOGRAD") ? (atoi(getenv("ASX_WINOGRAD")) <= 0 ? 0 : 3) : 3;
#endif
  // 1: layers with Cin <= 96 run the weight-stationary Winograd kernel (conv_winos_kernel, kernels_winos.h) when the option above
  // is 3; 0 (default -- the stationary form measured 3-8 % slower, profiles/NOTES.md round 4): conv_wino3_kernel everywhere.
  // ASX_WINOS or asx_set_option("winograd_stationary", n).
#ifdef ASX_EXPERIMENTAL_KERNELS
  int winos = getenv("ASX_WINOS") ? std::max(0, atoi(getenv("ASX_WINOS"))) : 0;
#else
  int winos = 0;
#endif
  // 1 (default): row GEMMs, channels-last convolutions (GATHER mode) and attention of THIS engine run the bf16 x 6 kernels when their
  // shapes allow (kernels_gemm3.h); 0: the fp32-MFMA kernels.  ASX_GEMM_BF16X6 or asx_set_option("gemm_bf16x6", n).
  int gemm_bf16x6 = getenv("ASX_GEMM_BF16X6") ? atoi(getenv("ASX_GEMM_BF16X6")) : 1;
  // 1 (default): every kernel "gemm_bf16x6" sends to the 16-bit matrix pipe -- tdf3_kernel (row GEMMs, GATHER-mode convolutions),
  // attention6_kernel / mha6_kernel, conv_wino6_kernel -- runs the fp16 x 3 arithmetic: operands scaled by power-of-two block exponents and
  // split into two fp16 parts, three MFMAs per product instead of six (kernels_gemm3.h); 4-30 % faster per launch, as close to float64 as the
  // bf16 x 6 form on finite data (profiles/r05_gemm_f16x3.txt, r05_attention_f16x3.txt, r05_wino6_f16x3.txt).  0: bf16 x 6 (exact
  // three-way split) everywhere.  ASX_GEMM_F16X3 or asx_set_option("gemm_f16x3", n).
  int gemm_f16x3 = getenv("ASX_GEMM_F16X3") ? atoi(getenv("ASX_GEMM_F16X3")) : 1;
  // 3x3 TFC convs with at least this many input channels run Winograd F(2x2,3x3) on the bf16 pipe (conv_wino6_kernel, kernels_wino6.h)
  // when "winograd" is 3 and "gemm_bf16x6" is on; 0 = never.  Default 144: measured faster than conv_wino3_kernel from level 2 of the
  // HQ_3 net down, equal on level 1, slower on level 0 (profiles/r05_wino6_forms.txt).  ASX_WINO6 or asx_set_option("winograd_bf16x6", n).
  int wino6 = getenv("ASX_WINO6") ? std::max(0, atoi(getenv("ASX_WINO6"))) : 144;
  // 3x3 TFC convs of 48 n -> 48 n channels with at most this many channels run the DIRECT implicit GEMM on the fp16 x 3 arithmetic (conv3h_kernel,
  // kernels_conv3h.h: 48 x 48 weight slices resident in LDS, no Winograd transforms; n x n launches per layer, the input slices summed through
  // the output) while "winograd" is 3 and "gemm_bf16x6" / "gemm_f16x3" are on; 0: never.  Default 144: levels 0, 1 and 2 of the HQ_3 geometry --
  // 5.2-5.5 ms per launch of 55 chunks against 8.5-8.9 on conv_wino3_kernel at 48 channels, 6.0 against 8.0-8.2 at 96, 3.76 against 4.0-4.1
  // (conv_wino6_kernel) at 144; the image is packed up to 144 channels.
  // ASX_CONV3H or asx_set_option("conv_direct_f16x3", n).
  int conv3h = getenv("ASX_CONV3H") ? std::max(0, atoi(getenv("ASX_CONV3H"))) : 144;
  // 1 (default): the 2 x 2 / stride-2 convolutions between the levels run conv_down6_kernel (kernels_updown6.h: bf16 x 6 -- exact three-way split
  // operands on the 16-bit matrix pipe, fp32 accumulation) while "gemm_bf16x6" is on; 0: the fp32-MFMA kernel conv_dma_kernel<2, 2, 2, 0, ...>.
  // ASX_DOWN6 or asx_set_option("conv_down_bf16x6", n).
  int down6 = getenv("ASX_DOWN6") ? atoi(getenv("ASX_DOWN6")) : 1;
  // the same for the transposed 2 x 2 / stride-2 convolutions of the decoder (conv_up6_kernel).  ASX_UP6 or asx_set_option("conv_up_bf16x6", n).
  int up6 = getenv("ASX_UP6") ? atoi(getenv("ASX_UP6")) : 1;
  // EXPERIMENTAL builds only (`python build.py --experimental`; the default library refuses the option).  1: a matrix whose only reader is a row
  // GEMM on the fp16 x 3 arithmetic is written by its producer as a PAIR IMAGE -- the two fp16 parts the GEMM multiplies, in the bytes of the
  // fp32 values, one exponent per (row, column tile) beside it (kernels_net.h TdfDmaArgs::xexp; kernels_gemm3.h) -- so the reader splits
  // nothing: the bottleneck activations of the TDF blocks, the Roformer feed-forward's hidden activations.  Measured round 6: the reader alone
  // is 3-17 % faster (profiles/r06_tdf3h_abl.txt, r06_tdf_pair_image_chain.txt), the whole nets are not (MDX TDF class 33.3 vs 33.2 ms,
  // BS-Roformer row GEMMs 940 vs 932 ms per step: profiles/NOTES.md) -- retired with the other measured-and-lost variants.  Default 0.
  // ASX_PAIR_IMAGES or asx_set_option("gemm_pair_images", n).
#ifdef ASX_EXPERIMENTAL_KERNELS
  int pair_images = getenv("ASX_PAIR_IMAGES") ? atoi(getenv("ASX_PAIR_IMAGES")) : 0;
#else
  int pair_images = 0;
#endif
  // split (bf16 x 3) images of this engine's weight matrices, built on first use and freed only with the engine or when the engine's
  // own weights are re-loaded: another engine of the process can never invalidate a pointer a captured graph of this one holds
  std::vector<W3Entry> w3;
  std::mutex w3_mu;
  // profiling
  bool prof = false;
  int prof_nprod = 0;   // set by the launchers of the split-operand kernels inside a timed() launch (see ProfRec::nprod)
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
  e->prof_nprod = 0;
  launch();
  r.nprod = e->prof_nprod;
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
static void host_env(int n, int hop, int T, std::vector<float> &env, int win_length = 0, const std::vector<float> *custom = nullptr) {
  std::vector<float> w;
  if (custom && (int)custom->size() == n) w = *custom;   // asx_set_stft_window: the caller's table instead of the Hann default
  else host_window(n, w, win_length);
  env.assign((size_t)n + (size_t)hop * (T - 1), 0.f);
  for (int t = 0; t < T; ++t)
    for (int k = 0; k < n; ++k) env[(size_t)t * hop + k] += w[k] * w[k];
}

static size_t stft_lds(const FftPlan &p) { return (size_t)p.nh * 2 * sizeof(float2); }
static size_t istft_lds(const FftPlan &p) { return ((size_t)p.nh * 3 + 1) * sizeof(float2); }
static size_t ht_istft_lds(const FftPlan &p) { return ((size_t)p.nh * 2 + 1) * sizeof(float2); }   // ht_istft_kernel stages X in bufB

