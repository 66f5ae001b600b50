// The 2 x 2 / stride-2 convolution between the levels of ConvTDFNet (uvr_lib_v5/mdxnet.py:66-72: Conv2d(c, c + g, (2, 2), stride 2) + BN + ReLU) on the
// 16-bit matrix pipe with fp32 results (round 6).  The fp32-MFMA kernel it replaces (conv_dma_kernel<2, 2, 2, 0, ...>, kernels_net.h) is bound
// by the fp32 matrix pipe: with half of its MFMAs removed it runs 33 % faster (profiles/r06_updown_half_mfma_probe.txt).
//
// Arithmetic: the bf16 x 6 form of kernels_gemm3.h -- every fp32 operand an EXACT sum of three bf16 numbers, six `v_mfma_f32_16x16x32_bf16` products per
// 32-deep k step, fp32 accumulation, dropped cross terms <= 2^-24 of a product -- no block exponents, so nothing to manage for a layer whose input is read once.
//
// Implicit GEMM per output row: M = output pixels, N = output channels, K = 4 Cin with k = (channel, dy, dx).  One MFMA k step = EIGHT channels x four taps; a
// lane's eight k values are channels (2 lk, 2 lk + 1) x (dy, dx) of its output pixel -- four `ds_read_b64` (the two dx taps of a pixel are adjacent floats).
// Workgroup = 2 output rows x 64 output pixels x 48 (or 96) output channels, four waves x (2 pixel tiles x NREP channel tiles); the input planes arrive by
// LDS-DMA exactly as in conv_dma_kernel (a stage = 8 planes of 4 rows x 128 floats = 16 KB), the weights as a pre-split fragment-ordered bf16 image (3 NREP KB
// per stage: 3 parts x NREP channel tiles x 64 lanes x 16 B) by the same DMA.  ONE stage buffer (26 KB at NREP = 3): a workgroup waits for its own DMA, four
// workgroups per CU cover each other's waits -- with two buffers and two or three workgroups the kernels measured slower (profiles/r06_up6_ab.txt).  The x
// fragments are split in registers (22 VALU per fragment, shared by the 6 NREP MFMAs that use it); the weight parts are read from LDS one at a time.
#pragma once
#include "kernels_gemm3.h"

namespace asx {

// NREP: 16-channel tiles of the output per workgroup: 3 (48 channels, 26 KB of LDS) or 6 (96 channels: the input is fetched once for all of them where Cout is
// a multiple of 96 -- 35 KB); four workgroups per CU either way
// TH_: output rows per workgroup, 2 or 4 (the taller tile halves the weight traffic per pixel: the deeper levels, where the weights of a tile outweigh its input)
template <int NREP_, int TH_ = 2>
struct Down6CfgT {
  static constexpr int TH = TH_, TW = 64, KC = 8, NREP = NREP_, NW = 16 * NREP_;
  static constexpr int IH = 2 * TH, IWA = 2 * TW;      // staged input rows / floats per row of a plane
  static constexpr int PLANE = IH * IWA;               // 512 / 1024 floats
  static constexpr int NJ = PLANE / 256;               // 1-KB DMA pieces per plane
  static constexpr int PAIRS = TH / 2;                 // pixel-tile pairs per wave
  static constexpr int PS = PLANE + 16;                // plane stride (floats): channel pairs (2 lk, 2 lk + 1) of neighbouring lane groups 32 banks apart
  static constexpr int XBYTES = KC * PS * 4;           // 16,896
  static constexpr int WSTAGE_U32 = 3 * NREP * 64 * 4; // 2304 uint32 = 9216 B per (channel group, stage): [part][n][lane][4 uint32]
  static constexpr int BUF = XBYTES + WSTAGE_U32 * 4;  // 26,112 B
  static constexpr int LDS_BYTES = BUF;                // ONE stage buffer: four workgroups per CU cover each other's DMA waits (see Up6CfgT)
};
typedef Down6CfgT<3> Down6Cfg;

// host: float -> bf16 (round to nearest even; NaN stays NaN)
static inline uint16_t down6_bf16_rne(float v) {
  uint32_t u;
  memcpy(&u, &v, 4);
  if ((u & 0x7f800000u) == 0x7f800000u) return (uint16_t)((u >> 16) | ((u & 0xffffu) ? 0x40u : 0u));
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}
static inline float down6_bf16_f(uint16_t h) {
  const uint32_t u = (uint32_t)h << 16;
  float f;
  memcpy(&f, &u, 4);
  return f;
}

// host: w [cout, cin, 2, 2] fp32 -> image [cg][stage][part][n][lane][8 bf16]; lane (li = lane & 15, lk = lane >> 4) holds output channel cg * 16 NREP + n * 16 + li,
// k values e * 4 + dy * 2 + dx for channels stage * 8 + 2 lk + e (zeros past cout / cin).  v = h + m + l exactly (the two differences are exact in fp32).
template <int NREP>
static inline void down6_pack(const float *w, int cout, int cin, std::vector<uint32_t> &img, int *cg_out, int *nst_out) {
  using Down6Cfg = Down6CfgT<NREP>;
  const int cg = (cout + Down6Cfg::NW - 1) / Down6Cfg::NW, nst = (cin + Down6Cfg::KC - 1) / Down6Cfg::KC;
  img.assign((size_t)cg * nst * Down6Cfg::WSTAGE_U32, 0u);
  uint16_t *o = reinterpret_cast<uint16_t *>(img.data());
  for (int g = 0; g < cg; ++g)
    for (int st = 0; st < nst; ++st)
      for (int n = 0; n < Down6Cfg::NREP; ++n)
        for (int lane = 0; lane < 64; ++lane) {
          const int co = g * Down6Cfg::NW + n * 16 + (lane & 15), lk = lane >> 4;
          for (int e = 0; e < 2; ++e)
            for (int tap = 0; tap < 4; ++tap) {
              const int c = st * Down6Cfg::KC + 2 * lk + e;
              const float v = (co < cout && c < cin) ? w[((size_t)co * cin + c) * 4 + tap] : 0.f;
              const uint16_t h = down6_bf16_rne(v);
              const float r1 = v - down6_bf16_f(h);
              const uint16_t m = down6_bf16_rne(r1);
              const uint16_t l = down6_bf16_rne(r1 - down6_bf16_f(m));
              const uint16_t parts[3] = {h, m, l};
              for (int p = 0; p < 3; ++p)
                o[((((size_t)(g * nst + st) * 3 + p) * Down6Cfg::NREP + n) * 64 + lane) * 8 + e * 4 + tap] = parts[p];
            }
        }
  if (cg_out) *cg_out = cg;
  if (nst_out) *nst_out = nst;
}

// the six products of one 32-deep k step for two pixel tiles x NREP channel tiles; weight fragments [part][n][lane] in LDS, read one part at a time
// (l, m, h: the smallest terms first, as in kernels_gemm3.h)
template <int NREP>
__device__ __forceinline__ void updown6_products(const u32x4 *w_s, int lane, const bf16x8 (&xh)[2], const bf16x8 (&xm)[2], const bf16x8 (&xl)[2],
                                                 f32x4 (&acc)[2][NREP]) {
  bf16x8 bw[NREP];
#pragma unroll
  for (int n = 0; n < NREP; ++n) bw[n] = __builtin_bit_cast(bf16x8, w_s[(2 * NREP + n) * 64 + lane]);   // w_l
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int n = 0; n < NREP; ++n) acc[m][n] = ASX_MFMA_BF16(xh[m], bw[n], acc[m][n]);
#pragma unroll
  for (int n = 0; n < NREP; ++n) bw[n] = __builtin_bit_cast(bf16x8, w_s[(NREP + n) * 64 + lane]);       // w_m
#pragma unroll
  for (int m = 0; m < 2; ++m) {
#pragma unroll
    for (int n = 0; n < NREP; ++n) acc[m][n] = ASX_MFMA_BF16(xm[m], bw[n], acc[m][n]);
#pragma unroll
    for (int n = 0; n < NREP; ++n) acc[m][n] = ASX_MFMA_BF16(xh[m], bw[n], acc[m][n]);
  }
#pragma unroll
  for (int n = 0; n < NREP; ++n) bw[n] = __builtin_bit_cast(bf16x8, w_s[n * 64 + lane]);                // w_h
#pragma unroll
  for (int m = 0; m < 2; ++m) {
#pragma unroll
    for (int n = 0; n < NREP; ++n) acc[m][n] = ASX_MFMA_BF16(xl[m], bw[n], acc[m][n]);
#pragma unroll
    for (int n = 0; n < NREP; ++n) acc[m][n] = ASX_MFMA_BF16(xm[m], bw[n], acc[m][n]);
#pragma unroll
    for (int n = 0; n < NREP; ++n) acc[m][n] = ASX_MFMA_BF16(xh[m], bw[n], acc[m][n]);
  }
}

// a.wp: the image above; a.CG / a.NCI: its channel groups / stages; a.tilesT / a.tilesF: tiles of 2 output rows x 64 output pixels.  F % 4 == 0 (launcher).
template <int NREP, int TH = 2>
__global__ __launch_bounds__(256, TH == 2 ? 4 : 3) void conv_down6_kernel(ConvArgs a) {
  using C = Down6CfgT<NREP, TH>;
  extern __shared__ float lds_f[];
  char *lds = reinterpret_cast<char *>(lds_f);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, lk = lane >> 4;

  int lid = xcd_remap(blockIdx.x, gridDim.x);
  const int cg = lid % a.CG;
  lid /= a.CG;
  const int tf = lid % a.tilesF;
  lid /= a.tilesF;
  const int tt = lid % a.tilesT;
  const int b = lid / a.tilesT;
  const int to0 = tt * C::TH, fo0 = tf * C::TW;
  const int ti0 = 2 * to0, fa0 = 2 * fo0;

  const float *xb = a.x + (int64_t)b * a.x_bstride;
  const uint32_t *wg = reinterpret_cast<const uint32_t *>(a.wp) + (int64_t)cg * a.NCI * C::WSTAGE_U32;
  const int64_t plane_sz = (int64_t)a.T * a.F;

  // a plane = 2 TH rows x 32 sixteen-byte slots: this lane's NJ slots, the same for every plane
  int sp_off[C::NJ];
  bool sp_ok[C::NJ];
#pragma unroll
  for (int j = 0; j < C::NJ; ++j) {
    const int sidx = j * 64 + lane;
    const int row = sidx >> 5, c4 = sidx & 31;
    const int t = ti0 + row, f = fa0 + c4 * 4;
    sp_ok[j] = t < a.T && f < a.F;
    sp_off[j] = t * a.F + f;
  }
  auto issue = [&](int st, int buf) {
    float *in_s = reinterpret_cast<float *>(lds + buf * C::BUF);
    uint32_t *w_s = reinterpret_cast<uint32_t *>(lds + buf * C::BUF + C::XBYTES);
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      const int pl = wave + 4 * p;
      const int c = st * C::KC + pl;
      const float *xc = xb + (int64_t)c * plane_sz;
      const bool cok = c < a.Cin;
#pragma unroll
      for (int j = 0; j < C::NJ; ++j) {
        const float *src = (cok && sp_ok[j]) ? xc + sp_off[j] : a.zeros;
        ASX_GLDS16(src, in_s + pl * C::PS + j * 256);
      }
    }
    const uint32_t *ws = wg + (int64_t)st * C::WSTAGE_U32;
#pragma unroll
    for (int i = 0; i < (3 * C::NREP + 3) / 4; ++i) {
      const int q = wave + 4 * i;                      // 3 NREP pieces of 1 KB
      if (q < 3 * C::NREP) ASX_GLDS16(ws + q * 256 + lane * 4, w_s + q * 256);
    }
  };

  f32x4 acc[C::PAIRS][2][C::NREP];
#pragma unroll
  for (int pi = 0; pi < C::PAIRS; ++pi)
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int n = 0; n < C::NREP; ++n) acc[pi][m][n] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // this wave, TH = 2: output row wave >> 1 of the tile, pixel tiles (wave & 1) * 2 + {0, 1}; TH = 4: output row `wave`, pixel-tile pairs {0, 1} and {2, 3}
  const int row = TH == 2 ? (wave >> 1) : wave;
  for (int st = 0; st < a.NCI; ++st) {
    if (st > 0) __syncthreads();                       // every wave is done with the buffer
    issue(st, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const float *in_s = reinterpret_cast<const float *>(lds);
    const u32x4 *w_s = reinterpret_cast<const u32x4 *>(lds + C::XBYTES);
    // the x fragments of a pair of pixel tiles first, then the weight parts one at a time (l, m, h: smallest terms first) -- 4 NREP live weight registers
    // instead of 12 NREP
#pragma unroll
    for (int pi = 0; pi < C::PAIRS; ++pi) {
      const int col0 = TH == 2 ? (wave & 1) * 2 : 2 * pi;
      bf16x8 xh[2], xm[2], xl[2];
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        // eight k values of pixel (col0 + m) * 16 + li: channels 2 lk, 2 lk + 1 x (dy, dx)
        const float *p0 = in_s + (2 * lk) * C::PS + (2 * row) * C::IWA + 2 * ((col0 + m) * 16 + li);
        const f32x2 c0d0 = *reinterpret_cast<const f32x2 *>(p0);
        const f32x2 c0d1 = *reinterpret_cast<const f32x2 *>(p0 + C::IWA);
        const f32x2 c1d0 = *reinterpret_cast<const f32x2 *>(p0 + C::PS);
        const f32x2 c1d1 = *reinterpret_cast<const f32x2 *>(p0 + C::PS + C::IWA);
        unsigned hh[4], mm[4], ll[4];
        split3_pair(c0d0.x, c0d0.y, hh[0], mm[0], ll[0]);
        split3_pair(c0d1.x, c0d1.y, hh[1], mm[1], ll[1]);
        split3_pair(c1d0.x, c1d0.y, hh[2], mm[2], ll[2]);
        split3_pair(c1d1.x, c1d1.y, hh[3], mm[3], ll[3]);
        xh[m] = __builtin_bit_cast(bf16x8, (u32x4){hh[0], hh[1], hh[2], hh[3]});
        xm[m] = __builtin_bit_cast(bf16x8, (u32x4){mm[0], mm[1], mm[2], mm[3]});
        xl[m] = __builtin_bit_cast(bf16x8, (u32x4){ll[0], ll[1], ll[2], ll[3]});
      }
      updown6_products<C::NREP>(w_s, lane, xh, xm, xl, acc[pi]);
    }
  }

  // ---- epilogue: bias + activation (+ residual); a lane holds four consecutive pixels of output channel li of each channel tile
  const int t = to0 + row;
  if (t >= a.To) return;
  float *yb = a.y + (int64_t)b * a.y_bstride;
  const float *rb = a.res ? a.res + (int64_t)b * a.aux_bstride : nullptr;
#pragma unroll
  for (int n = 0; n < C::NREP; ++n) {
    const int co = cg * C::NW + n * 16 + li;
    if (co >= a.Cout) continue;
    const float bv = a.bias[co];
#pragma unroll
    for (int pi = 0; pi < C::PAIRS; ++pi)
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        const int col0 = TH == 2 ? (wave & 1) * 2 : 2 * pi;
        const int f = fo0 + (col0 + m) * 16 + lk * 4;
        const int64_t off = ((int64_t)co * a.To + t) * a.Fo + f;
        const f32x4 v = acc[pi][m][n];
        if ((a.Fo & 3) == 0 && f + 4 <= a.Fo) {
          f32x4 o;
          o.x = act_fn(v.x + bv, a.act);
          o.y = act_fn(v.y + bv, a.act);
          o.z = act_fn(v.z + bv, a.act);
          o.w = act_fn(v.w + bv, a.act);
          if (rb != nullptr) o += *reinterpret_cast<const f32x4 *>(rb + off);
          *reinterpret_cast<f32x4 *>(yb + off) = o;
        } else {
          const float ov[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (f + r < a.Fo) yb[off + r] = act_fn(ov[r] + bv, a.act) + (rb != nullptr ? rb[off + r] : 0.f);
        }
      }
  }
}

// ---- the transposed 2 x 2 / stride-2 convolution of the decoder (uvr_lib_v5/mdxnet.py:80-86: ConvTranspose2d(c, c - g, (2, 2), stride 2) + BN + ReLU, then
// `x *= skip`, mdxnet.py:113) in the same arithmetic.  As in conv_dma_kernel<1, 1, 1, 0, ..., EPI_UP> it is a 1 x 1 conv onto 4 Cout virtual channels: n-tile
// nt -> pair = nt / 2 -> (dy = pair / CT, channel tile ct = pair % CT), dx = nt & 1, so that the two dx tiles of a pair interleave into 32-byte stores.
// One MFMA k step = 32 input channels; a lane's eight k values are channels 2 (lk + 4 jj) + e of its pixel (jj = 0 .. 3, e = 0, 1): planes are stored in
// PAIRS of 1 KB (one LDS-DMA instruction each) with 64 bytes of padding between pairs, which puts the four lane groups of a `ds_read_b32` on four
// different 16-bank windows.  Workgroup = 2 input rows x 64 pixels x NREP virtual tiles; LDS 17 KB + 3 NREP KB (one stage buffer, as above).
template <int NREP_>
struct Up6CfgT {
  static constexpr int TH = 2, TW = 64, KC = 32, NREP = NREP_;
  static constexpr int PAIR = 2 * TH * TW + 16;        // floats between plane pairs (1 KB of data + 64 B)
  static constexpr int XBYTES = (KC / 2) * PAIR * 4;   // 17,408
  static constexpr int WSTAGE_U32 = 3 * NREP * 64 * 4;
  static constexpr int BUF = XBYTES + WSTAGE_U32 * 4;
  static constexpr int NBUF = 1;                       // one stage buffer (36 KB at NREP = 6): three / four workgroups per CU cover each other's DMA waits; two buffers
  static constexpr int LDS_BYTES = NBUF * BUF;         // (72 KB, two workgroups) measured slower than the fp32 kernel, profiles/r06_up6_ab.txt
};

// host: w [cin, cout, 2, 2] fp32 -> image [cg][stage of 32 channels][part][n][lane][8 bf16]; virtual tile nt = cg * NREP + n as above; lane (li, lk), k value
// jj * 2 + e <-> channel stage * 32 + 2 (lk + 4 jj) + e
template <int NREP>
static inline void up6_pack(const float *w, int cout, int cin, std::vector<uint32_t> &img, int *cg_out, int *nst_out) {
  using C = Up6CfgT<NREP>;
  const int CT = (cout + 15) / 16, vt = 4 * CT;
  const int cg = (vt + NREP - 1) / NREP, nst = (cin + C::KC - 1) / C::KC;
  img.assign((size_t)cg * nst * C::WSTAGE_U32, 0u);
  uint16_t *o = reinterpret_cast<uint16_t *>(img.data());
  for (int g = 0; g < cg; ++g)
    for (int st = 0; st < nst; ++st)
      for (int n = 0; n < NREP; ++n) {
        const int nt = g * NREP + n, pair = nt / 2, dx = nt & 1, dy = pair / CT, ct = pair % CT;
        for (int lane = 0; lane < 64; ++lane) {
          const int co = ct * 16 + (lane & 15), lk = lane >> 4;
          for (int kk = 0; kk < 8; ++kk) {
            const int c = st * C::KC + 2 * (lk + 4 * (kk >> 1)) + (kk & 1);
            const float v = (dy < 2 && co < cout && c < cin) ? w[(((size_t)c * cout + co) * 2 + dy) * 2 + dx] : 0.f;
            const uint16_t h = down6_bf16_rne(v);
            const float r1 = v - down6_bf16_f(h);
            const uint16_t m = down6_bf16_rne(r1);
            const uint16_t l = down6_bf16_rne(r1 - down6_bf16_f(m));
            const uint16_t parts[3] = {h, m, l};
            for (int p = 0; p < 3; ++p) o[((((size_t)(g * nst + st) * 3 + p) * NREP + n) * 64 + lane) * 8 + kk] = parts[p];
          }
        }
      }
  if (cg_out) *cg_out = cg;
  if (nst_out) *nst_out = nst;
}

// a.T / a.F: INPUT plane; output [B, Cout, 2 T, 2 F]; a.tilesT / a.tilesF: tiles of 2 x 64 input pixels; F % 4 == 0 (launcher)
template <int NREP>
__global__ __launch_bounds__(256, 4) void conv_up6_kernel(ConvArgs a) {
  using C = Up6CfgT<NREP>;
  static_assert(NREP % 2 == 0, "the dx tiles of a pair stay together");
  extern __shared__ float lds_f[];
  char *lds = reinterpret_cast<char *>(lds_f);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, lk = lane >> 4;

  int lid = xcd_remap(blockIdx.x, gridDim.x);
  const int cg = lid % a.CG;
  lid /= a.CG;
  const int tf = lid % a.tilesF;
  lid /= a.tilesF;
  const int tt = lid % a.tilesT;
  const int b = lid / a.tilesT;
  const int to0 = tt * C::TH, fo0 = tf * C::TW;

  const float *xb = a.x + (int64_t)b * a.x_bstride;
  const uint32_t *wg = reinterpret_cast<const uint32_t *>(a.wp) + (int64_t)cg * a.NCI * C::WSTAGE_U32;
  const int64_t plane_sz = (int64_t)a.T * a.F;

  // one DMA instruction = a plane PAIR: lanes 0 .. 31 the 32 slots (2 rows x 16) of the even plane, lanes 32 .. 63 of the odd one
  const int slot = lane & 31, srow = slot >> 4, sc4 = slot & 15;
  const bool sp_ok = (to0 + srow) < a.T && (fo0 + sc4 * 4) < a.F;
  const int sp_off = (to0 + srow) * a.F + fo0 + sc4 * 4;
  auto issue = [&](int st, int buf) {
    float *in_s = reinterpret_cast<float *>(lds + buf * C::BUF);
    uint32_t *w_s = reinterpret_cast<uint32_t *>(lds + buf * C::BUF + C::XBYTES);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int pr = wave + 4 * i;                     // 16 pairs per stage
      const int c = st * C::KC + 2 * pr + (lane >> 5);
      const float *src = (c < a.Cin && sp_ok) ? xb + (int64_t)c * plane_sz + sp_off : a.zeros;
      ASX_GLDS16(src, in_s + pr * C::PAIR);
    }
    const uint32_t *ws = wg + (int64_t)st * C::WSTAGE_U32;
#pragma unroll
    for (int i = 0; i < (3 * NREP + 3) / 4; ++i) {
      const int q = wave + 4 * i;
      if (q < 3 * NREP) ASX_GLDS16(ws + q * 256 + lane * 4, w_s + q * 256);
    }
  };

  f32x4 acc[2][NREP];
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int n = 0; n < NREP; ++n) acc[m][n] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int row = wave >> 1, col0 = (wave & 1) * 2;   // this wave: input row `row` of the tile, pixel tiles col0, col0 + 1
  for (int st = 0; st < a.NCI; ++st) {
    if (st > 0) __syncthreads();                       // every wave is done with the buffer
    issue(st, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const float *in_s = reinterpret_cast<const float *>(lds);
    const u32x4 *w_s = reinterpret_cast<const u32x4 *>(lds + C::XBYTES);
    bf16x8 xh[2], xm[2], xl[2];
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      const float *p0 = in_s + lk * C::PAIR + row * C::TW + (col0 + m) * 16 + li;
      unsigned hh[4], mm[4], ll[4];
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) {
        const float v0 = p0[jj * 4 * C::PAIR], v1 = p0[jj * 4 * C::PAIR + C::TH * C::TW];   // channels 2 (lk + 4 jj), + 1
        split3_pair(v0, v1, hh[jj], mm[jj], ll[jj]);
      }
      xh[m] = __builtin_bit_cast(bf16x8, (u32x4){hh[0], hh[1], hh[2], hh[3]});
      xm[m] = __builtin_bit_cast(bf16x8, (u32x4){mm[0], mm[1], mm[2], mm[3]});
      xl[m] = __builtin_bit_cast(bf16x8, (u32x4){ll[0], ll[1], ll[2], ll[3]});
    }
    updown6_products<NREP>(w_s, lane, xh, xm, xl, acc);
  }

  // ---- epilogue (the arithmetic of conv_epilogue's EPI_UP path): bias + activation, the two dx tiles of a pair interleaved, times the skip tensor
  const int t = to0 + row;
  if (t >= a.T) return;
  const int CT = (a.Cout + 15) / 16;
  const int To2 = a.T * 2, Fo2 = a.F * 2;
  float *yb = a.y + (int64_t)b * a.y_bstride;
  const float *sb = a.skip ? a.skip + (int64_t)b * a.aux_bstride : nullptr;
#pragma unroll
  for (int np = 0; np < NREP / 2; ++np) {
    const int pair = (cg * NREP) / 2 + np;
    const int dy = pair / CT, ct = pair - dy * CT;
    if (dy >= 2) continue;
    const int co = ct * 16 + li;
    if (co >= a.Cout) continue;
    const float bv = a.bias[co];
    f32x4 s0[2], s1[2];                                // the pair's skip values first: four loads in flight
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      const int f = fo0 + (col0 + m) * 16 + lk * 4;
      const int64_t off = ((int64_t)co * To2 + (2 * t + dy)) * Fo2 + 2 * f;
      const bool in = sb != nullptr && f + 4 <= a.F;
      s0[m] = in ? *reinterpret_cast<const f32x4 *>(sb + off) : (f32x4){1.f, 1.f, 1.f, 1.f};
      s1[m] = in ? *reinterpret_cast<const f32x4 *>(sb + off + 4) : (f32x4){1.f, 1.f, 1.f, 1.f};
    }
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      const int f = fo0 + (col0 + m) * 16 + lk * 4;
      if (f >= a.F) continue;
      const int64_t off = ((int64_t)co * To2 + (2 * t + dy)) * Fo2 + 2 * f;
      const f32x4 v0 = acc[m][2 * np], v1 = acc[m][2 * np + 1];
      if (f + 4 <= a.F) {
        f32x4 r0, r1;
        r0.x = act_fn(v0.x + bv, a.act);
        r0.y = act_fn(v1.x + bv, a.act);
        r0.z = act_fn(v0.y + bv, a.act);
        r0.w = act_fn(v1.y + bv, a.act);
        r1.x = act_fn(v0.z + bv, a.act);
        r1.y = act_fn(v1.z + bv, a.act);
        r1.z = act_fn(v0.w + bv, a.act);
        r1.w = act_fn(v1.w + bv, a.act);
        r0 *= s0[m];
        r1 *= s1[m];
        *reinterpret_cast<f32x4 *>(yb + off) = r0;
        *reinterpret_cast<f32x4 *>(yb + off + 4) = r1;
      } else {
        const float o[8] = {v0.x, v1.x, v0.y, v1.y, v0.z, v1.z, v0.w, v1.w};
#pragma unroll
        for (int q = 0; q < 8; ++q)
          if (f + (q >> 1) < a.F) yb[off + q] = act_fn(o[q] + bv, a.act) * (sb != nullptr ? sb[off + q] : 1.f);
      }
    }
  }
}

}  // namespace asx
