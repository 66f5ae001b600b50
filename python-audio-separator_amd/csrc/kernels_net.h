// ConvTDFNet kernels for gfx950: fp32-exact MFMA (v_mfma_f32_16x16x4_f32).
//
// Architecture spec: uvr_lib_v5/mdxnet.py:30-120, uvr_lib_v5/modules.py:5-74 (the
// graph the reference runs through onnxruntime, mdx_separator.py:122-123).
// Activations are [B, C, T, F] with F fastest (the layout the reference's convs
// see after x.transpose(-1,-2), mdxnet.py:101).  BatchNorm is folded into the
// weights by the host; ReLU, the residual add of TFC_TDF (modules.py:74) and the
// skip multiply of the decoder (mdxnet.py:113) are fused into epilogues.
//
// Two kernel families:
//   conv_mfma : K runs over channels (planes), M over 16 consecutive F positions.
//               3x3/pad1 (TFC), 2x2/stride2 (ds), 1x1 (first/final) and the 2x2
//               stride-2 transposed conv (us) as a 1x1 conv onto 4*Cout virtual
//               channels with a pixel-shuffle epilogue.
//   tdf_mfma  : K runs along F (contiguous), i.e. a row GEMM x[M,K] @ W[N,K]^T.
//
// MFMA 16x16x4 f32 operand maps (lane l, li = l & 15, lk = l >> 4):
//   A[i=li][k=lk], B[k=lk][j=li], D[i=4*lk+r][j=li] for r in 0..3.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace asx {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

#define ASX_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

// XCD-aware bijective remap of a 1-D grid: consecutive logical ids run on one XCD
// (dispatcher places block b on XCD b % 8), so neighbouring tiles share an L2.
__device__ __forceinline__ int xcd_remap(int bid, int nblk) {
  const int q = nblk >> 3, r = nblk & 7;
  const int xcd = bid & 7, slot = bid >> 3;
  const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + slot;
}

enum { EPI_BIAS_ACT = 0, EPI_UP_MULSKIP = 1 };

template <int KH_, int KW_, int S_, int PAD_, int NREP_, int KC_, int RPW_, int EPI_>
struct ConvCfg {
  static constexpr int KH = KH_, KW = KW_, S = S_, PAD = PAD_, NREP = NREP_, KC = KC_, RPW = RPW_, EPI = EPI_;
  static constexpr int TH = 4 * RPW, TW = 64;
  static constexpr int IH = (TH - 1) * S + KH, IW = (TW - 1) * S + KW;
  static constexpr int PS0 = IH * IW;
  // plane stride: two planes read by one 32-lane group must hit disjoint banks
  static constexpr int PS = (S == 1) ? (PS0 + ((16 - PS0 % 32) + 32) % 32) : (PS0 | 1);
  static constexpr int NW = 16 * NREP;
  static constexpr int NWP = (NREP % 2 == 0) ? NW + 16 : NW;  // row stride == 16 (mod 32) words
  static constexpr int NTAP = KH * KW;
  static constexpr int IN_ELEMS = KC * PS;
  static constexpr int W_ROWS = NTAP * KC;
  static constexpr int W_ELEMS = W_ROWS * NWP;
  static constexpr int WSTAGE = ((W_ROWS * NWP + 255) / 256) * 256;  // packed weight floats per stage (global)
  static constexpr int LDS_BYTES = (IN_ELEMS + W_ELEMS) * 4;
  static constexpr int MREP = RPW * 4;
  static constexpr int NPL = (PS0 + 255) / 256;                // staged elements per thread per plane
  static constexpr int NIN = KC * NPL;                         // input elements per thread per stage
  static constexpr int NWV = (W_ROWS * NWP / 4 + 255) / 256;   // weight float4 per thread per stage
};

enum { ACT_NONE = 0, ACT_RELU = 1, ACT_GELU = 2 };

__device__ __forceinline__ float act_fn(float v, int act) {
  if (act == ACT_RELU) return fmaxf(v, 0.f);
  if (act == ACT_GELU) return 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f));  // nn.GELU() (erf form)
  return v;
}

// x / y / aux may be channel-slice views of larger [B, Ctot, T, F] buffers: the pointers are
// pre-offset to the first channel of the view and the *_bstride fields give the distance (in
// floats) between consecutive batch items.
struct ConvArgs {
  const float *x;      // [B, Cin, T, F] view
  const float *wp;     // packed [CG][NCI][WSTAGE]: rows (tap, kc) of NWP floats, zero padded
  const float *bias;   // padded per (virtual) output channel
  const float *skip;   // EPI_UP: optional [B, Cout, 2T, 2F] multiplier (mdxnet.py:113) or nullptr
  const float *res;    // EPI_BIAS_ACT: optional residual [B, Cout, To, Fo] added after the activation, or nullptr
  const float *zeros;  // >= 16 B of zeros (DMA source of out-of-range slots)
  float *y;            // output view
  int B, Cin, Cout, T, F;    // input geometry; Cout = real output channels
  int To, Fo;                // output spatial size handled by tiles (= T,F; T/2,F/2 for stride 2)
  int tilesT, tilesF, CG, NCI;
  int act;
  int nt;              // 1: non-temporal output stores of the full-tile epilogue (A/B switch ASX_NT)
  int64_t x_bstride, y_bstride, aux_bstride;
};

// Shared epilogue of both conv kernel families.
template <class CFG>
__device__ __forceinline__ void conv_epilogue(const ConvArgs &a, f32x4 (&acc)[CFG::MREP][CFG::NREP], int b, int cg,
                                              int to0, int fo0, int wave, int li, int lk) {
  constexpr int NREP = CFG::NREP, MREP = CFG::MREP, RPW = CFG::RPW, NW = CFG::NW, TH = CFG::TH, TW = CFG::TW;
  if constexpr (CFG::EPI == EPI_BIAS_ACT) {
    const bool full = ((a.Fo & 3) == 0) && (to0 + TH <= a.To) && (fo0 + TW <= a.Fo);
    float *yb = a.y + (int64_t)b * a.y_bstride;
    const float *rb = a.res ? a.res + (int64_t)b * a.aux_bstride : nullptr;
#pragma unroll
    for (int n = 0; n < NREP; ++n) {
      const int co = cg * NW + n * 16 + li;
      const float bv = a.bias[co];
      if (co >= a.Cout) continue;
      if (full) {
        f32x4 rs[MREP];
        if (rb != nullptr) {
#pragma unroll
          for (int m = 0; m < MREP; ++m) {
            const int t = to0 + wave * RPW + (m >> 2), f = fo0 + (m & 3) * 16 + lk * 4;
            rs[m] = *reinterpret_cast<const f32x4 *>(rb + ((int64_t)co * a.To + t) * a.Fo + f);
          }
        }
#pragma unroll
        for (int m = 0; m < MREP; ++m) {
          const int t = to0 + wave * RPW + (m >> 2), f = fo0 + (m & 3) * 16 + lk * 4;
          f32x4 v = acc[m][n];
          v.x = act_fn(v.x + bv, a.act);
          v.y = act_fn(v.y + bv, a.act);
          v.z = act_fn(v.z + bv, a.act);
          v.w = act_fn(v.w + bv, a.act);
          if (rb != nullptr) v += rs[m];
          if (a.nt) __builtin_nontemporal_store(v, reinterpret_cast<f32x4 *>(yb + ((int64_t)co * a.To + t) * a.Fo + f));
          else *reinterpret_cast<f32x4 *>(yb + ((int64_t)co * a.To + t) * a.Fo + f) = v;
        }
      } else {
#pragma unroll
        for (int m = 0; m < MREP; ++m) {
          const int t = to0 + wave * RPW + (m >> 2), f = fo0 + (m & 3) * 16 + lk * 4;
          if (t >= a.To) continue;
          const f32x4 v = acc[m][n];
          const float o[4] = {v.x, v.y, v.z, v.w};
          const int64_t off = ((int64_t)co * a.To + t) * a.Fo + f;
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (f + r < a.Fo) yb[off + r] = act_fn(o[r] + bv, a.act) + (rb != nullptr ? rb[off + r] : 0.f);
        }
      }
    }
  } else {
    // transposed 2x2/stride-2 conv as a 1x1 conv onto 4*Cout virtual channels:
    // virtual n-tile nt = cg*NREP + n;  pair = nt/2 -> (dy = pair / CT, ct = pair % CT), dx = nt & 1
    const int CT = (a.Cout + 15) / 16;
    const int To2 = a.T * 2, Fo2 = a.F * 2;
    const bool full = (to0 + TH <= a.T) && (fo0 + TW <= a.F);
    float *yb = a.y + (int64_t)b * a.y_bstride;
    const float *sb = a.skip ? a.skip + (int64_t)b * a.aux_bstride : nullptr;
#pragma unroll
    for (int np = 0; np < NREP / 2; ++np) {
      const int pair = (cg * NREP) / 2 + np;
      const int dy = pair / CT, ct = pair - dy * CT;
      if (dy >= 2) continue;
      const int co = ct * 16 + li;
      const float bv = a.bias[co];
      if (co >= a.Cout) continue;
      if (full) {
        f32x4 s0[MREP], s1[MREP];
        if (sb != nullptr) {
#pragma unroll
          for (int m = 0; m < MREP; ++m) {
            const int t = to0 + wave * RPW + (m >> 2), f = fo0 + (m & 3) * 16 + lk * 4;
            const int64_t off = ((int64_t)co * To2 + (2 * t + dy)) * Fo2 + 2 * f;
            s0[m] = *reinterpret_cast<const f32x4 *>(sb + off);
            s1[m] = *reinterpret_cast<const f32x4 *>(sb + off + 4);
          }
        }
#pragma unroll
        for (int m = 0; m < MREP; ++m) {
          const int t = to0 + wave * RPW + (m >> 2), f = fo0 + (m & 3) * 16 + lk * 4;
          const int64_t off = ((int64_t)co * To2 + (2 * t + dy)) * Fo2 + 2 * f;
          const f32x4 v0 = acc[m][2 * np], v1 = acc[m][2 * np + 1];
          f32x4 r0, r1;
          r0.x = act_fn(v0.x + bv, a.act);
          r0.y = act_fn(v1.x + bv, a.act);
          r0.z = act_fn(v0.y + bv, a.act);
          r0.w = act_fn(v1.y + bv, a.act);
          r1.x = act_fn(v0.z + bv, a.act);
          r1.y = act_fn(v1.z + bv, a.act);
          r1.z = act_fn(v0.w + bv, a.act);
          r1.w = act_fn(v1.w + bv, a.act);
          if (sb != nullptr) {
            r0 *= s0[m];
            r1 *= s1[m];
          }
          *reinterpret_cast<f32x4 *>(yb + off) = r0;
          *reinterpret_cast<f32x4 *>(yb + off + 4) = r1;
        }
      } else {
#pragma unroll
        for (int m = 0; m < MREP; ++m) {
          const int t = to0 + wave * RPW + (m >> 2), f = fo0 + (m & 3) * 16 + lk * 4;
          if (t >= a.T || f >= a.F) continue;
          const f32x4 v0 = acc[m][2 * np], v1 = acc[m][2 * np + 1];
          const float o[8] = {v0.x, v1.x, v0.y, v1.y, v0.z, v1.z, v0.w, v1.w};
          const int64_t off = ((int64_t)co * To2 + (2 * t + dy)) * Fo2 + 2 * f;
#pragma unroll
          for (int q = 0; q < 8; ++q)
            if (f + (q >> 1) < a.F) yb[off + q] = act_fn(o[q] + bv, a.act) * (sb != nullptr ? sb[off + q] : 1.f);
        }
      }
    }
  }
}

template <class CFG>
__global__ __launch_bounds__(256) void conv_mfma_kernel(ConvArgs a) {
  extern __shared__ float lds_f[];
  float *in_s = lds_f;
  float *w_s = lds_f + CFG::IN_ELEMS;
  constexpr int KH = CFG::KH, KW = CFG::KW, S = CFG::S, PAD = CFG::PAD, NREP = CFG::NREP, KC = CFG::KC;
  constexpr int RPW = CFG::RPW, MREP = CFG::MREP, IW = CFG::IW, PS = CFG::PS, PS0 = CFG::PS0;
  constexpr int NWP = CFG::NWP, NIN = CFG::NIN, NWV = CFG::NWV, TH = CFG::TH, TW = CFG::TW;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, lk = lane >> 4;

  const int nblk = gridDim.x;
  int lid = xcd_remap(blockIdx.x, nblk);
  const int cg = lid % a.CG;
  lid /= a.CG;
  const int tf = lid % a.tilesF;
  lid /= a.tilesF;
  const int tt = lid % a.tilesT;
  const int b = lid / a.tilesT;
  const int to0 = tt * TH, fo0 = tf * TW;          // output tile origin
  const int ti0 = to0 * S - PAD, fi0 = fo0 * S - PAD;  // input tile origin

  const float *xb = a.x + (int64_t)b * a.x_bstride;
  const float *wg = a.wp + (int64_t)cg * a.NCI * CFG::WSTAGE;

  f32x4 acc[MREP][NREP];
#pragma unroll
  for (int m = 0; m < MREP; ++m)
#pragma unroll
    for (int n = 0; n < NREP; ++n) acc[m][n] = (f32x4){0.f, 0.f, 0.f, 0.f};

  float rin[NIN];
  float4 rw[NWV];

  // Each thread stages the same NPL in-plane positions of every plane, so the spatial part of
  // the address (and its validity) is computed once; loads are unconditional (clamped address +
  // select) so that all of a stage's loads are in flight together.
  constexpr int NPL = CFG::NPL;
  int sp_off[NPL];
  bool sp_ok[NPL];
#pragma unroll
  for (int j = 0; j < NPL; ++j) {
    const int r = tid + j * 256;
    const int row = r / IW, col = r - row * IW;
    const int t = ti0 + row, f = fi0 + col;
    sp_ok[j] = (r < PS0) && t >= 0 && t < a.T && f >= 0 && f < a.F;
    const int tc = min(max(t, 0), a.T - 1), fc = min(max(f, 0), a.F - 1);
    sp_off[j] = tc * a.F + fc;
  }
  const int64_t plane_sz = (int64_t)a.T * a.F;
  constexpr int WV_TOTAL = CFG::W_ROWS * NWP / 4;

  auto fetch = [&](int ci) {
#pragma unroll
    for (int pl = 0; pl < KC; ++pl) {
      const int c = ci * KC + pl;
      const float *xp = xb + (int64_t)min(c, a.Cin - 1) * plane_sz;
#pragma unroll
      for (int j = 0; j < NPL; ++j) {
        rin[pl * NPL + j] = xp[sp_off[j]];  // raw; masked at commit (keeps the load unconditional)
      }
    }
    const float4 *w4 = reinterpret_cast<const float4 *>(wg + (int64_t)ci * CFG::WSTAGE);
#pragma unroll
    for (int it = 0; it < NWV; ++it) {
      const int e = tid + it * 256;
      rw[it] = w4[min(e, WV_TOTAL - 1)];
    }
  };
  auto commit = [&](int ci) {
#pragma unroll
    for (int pl = 0; pl < KC; ++pl) {
      const bool cok = ci * KC + pl < a.Cin;
#pragma unroll
      for (int j = 0; j < NPL; ++j) {
        const int r = tid + j * 256;
        if (r < PS0) in_s[pl * PS + r] = (cok && sp_ok[j]) ? rin[pl * NPL + j] : 0.f;
      }
    }
#pragma unroll
    for (int it = 0; it < NWV; ++it) {
      const int e = tid + it * 256;
      if (e < WV_TOTAL) *reinterpret_cast<float4 *>(&w_s[e * 4]) = rw[it];
    }
  };

  fetch(0);
  for (int ci = 0; ci < a.NCI; ++ci) {
    __syncthreads();
    commit(ci);
    __syncthreads();
    if (ci + 1 < a.NCI) fetch(ci + 1);
    if constexpr (KC == 2) {
      // two-channel stages of the 2x2 / stride-2 conv (same k-step layout as conv_dma_kernel: channel lk >> 1, tap dx = lk & 1)
      static_assert(KC != 2 || (S == 2 && KW == 2), "two-channel stages: only the 2x2 / stride-2 conv");
      const int c2 = lk >> 1, dx2 = lk & 1;
#pragma unroll
      for (int dy = 0; dy < KH; ++dy) {
        float bf[NREP];
#pragma unroll
        for (int n = 0; n < NREP; ++n) bf[n] = w_s[((dy * KW + dx2) * KC + c2) * NWP + n * 16 + li];
        float af[MREP];
#pragma unroll
        for (int m = 0; m < MREP; ++m) {
          const int rr = m >> 2, cc = m & 3;
          af[m] = in_s[c2 * PS + ((wave * RPW + rr) * S + dy) * IW + (cc * 16 + li) * S + dx2];
        }
#pragma unroll
        for (int m = 0; m < MREP; ++m)
#pragma unroll
          for (int n = 0; n < NREP; ++n) acc[m][n] = ASX_MFMA(af[m], bf[n], acc[m][n]);
      }
    } else {
#pragma unroll
    for (int tap = 0; tap < KH * KW; ++tap) {
      const int dy = tap / KW, dx = tap % KW;
#pragma unroll
      for (int kq = 0; kq < KC / 4; ++kq) {
        float bf[NREP];
#pragma unroll
        for (int n = 0; n < NREP; ++n) bf[n] = w_s[(tap * KC + kq * 4 + lk) * NWP + n * 16 + li];
        float af[MREP];
#pragma unroll
        for (int m = 0; m < MREP; ++m) {
          const int rr = m >> 2, cc = m & 3;
          af[m] = in_s[(kq * 4 + lk) * PS + ((wave * RPW + rr) * S + dy) * IW + (cc * 16 + li) * S + dx];
        }
#pragma unroll
        for (int m = 0; m < MREP; ++m)
#pragma unroll
          for (int n = 0; n < NREP; ++n) acc[m][n] = ASX_MFMA(af[m], bf[n], acc[m][n]);
      }
    }
    }
  }

  conv_epilogue<CFG>(a, acc, b, cg, to0, fo0, wave, li, lk);
}

// ---------------------------------------------------------------------------
// TDF row GEMM:  y[m, n] = relu(scale[c(m)] * (sum_k x[m,k] w[n,k] + bias[n]) + shift[c(m)]) (+ res[m,n])
// rows m = ((b*C + c)*T + t), so c(m) = (m / T) % C.     (modules.py:57-74)
// Block tile: BM = 16*MREP rows x BN = 64*NREP cols, 4 waves side by side along N.
// The weight tile is the MFMA A operand (i -> n), the activation tile the B
// operand (j -> m), so every lane ends up with 4 consecutive n of one row m.
// Fragments are fetched with ds_read_b128 (4 consecutive k per lane); the k
// order inside a 16-wide step is permuted identically for both operands.
// ---------------------------------------------------------------------------
// erf by Abramowitz & Stegun 7.1.26 (|error| <= 1.5e-7 absolute), branch-free: one v_rcp, one v_exp, seven fma -- against libm's
// erff (two polynomial branches + an exp branch, ~3x the VALU work).  GELU's absolute error is then <= 0.75e-7 |v|.
__device__ __forceinline__ float fast_erf(float x) {
  const float ax = fabsf(x);
  const float t = __builtin_amdgcn_rcpf(__fmaf_rn(0.3275911f, ax, 1.0f));
  float p = __fmaf_rn(1.061405429f, t, -1.453152027f);
  p = __fmaf_rn(p, t, 1.421413741f);
  p = __fmaf_rn(p, t, -0.284496736f);
  p = __fmaf_rn(p, t, 0.254829592f);
  const float r = 1.0f - p * t * __expf(-ax * ax);
  return copysignf(r, x);
}

// row-GEMM activations: 0 none, 1 relu, 2 gelu (erf), 3 tanh, 4 leaky relu (slope 0.01), 5 sigmoid, 6 gelu on fast_erf
__device__ __forceinline__ float tdf_act(float v, int act) {
  if (act == 1) return fmaxf(v, 0.f);
  if (act == 6) return 0.5f * v * (1.0f + fast_erf(v * 0.70710678118654752440f));
  if (act == 2) return 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f));
  if (act == 3) return tanhf(v);
  if (act == 4) return v > 0.f ? v : 0.01f * v;
  if (act == 5) return 1.0f / (1.0f + expf(-v));
  return v;
}

struct TdfArgs {
  const float *x;      // [M, K]
  const float *w;      // [N, K]
  const float *bias;   // [N] or nullptr
  const float *scale;  // [C]
  const float *shift;  // [C]
  const float *res;    // [M, N] or nullptr
  float *y;            // [M, N]
  int64_t M;
  int N, K, C, T;
  int relu;            // 0: no activation (TFC-TDF v3 applies norm/act before the linear)
};

template <int NREP, int MREP>
struct TdfCfg {
  static constexpr int BK = 32, RSW = BK + 8;
  static constexpr int BM = 16 * MREP, BN = 64 * NREP;
  static constexpr int LDS_BYTES = (BM + BN) * RSW * 4;
  static constexpr int NXV = (BM * BK / 4 + 255) / 256;
  static constexpr int NWV = (BN * BK / 4 + 255) / 256;
};

template <int NREP, int MREP, bool KVEC>
__global__ __launch_bounds__(256, 2) void tdf_mfma_kernel(TdfArgs a) {
  using CFG = TdfCfg<NREP, MREP>;
  constexpr int BK = CFG::BK, RSW = CFG::RSW, BM = CFG::BM, BN = CFG::BN, NXV = CFG::NXV, NWV = CFG::NWV;
  extern __shared__ float lds_f[];
  float *xs = lds_f;             // [BM][RSW]
  float *ws = lds_f + BM * RSW;  // [BN][RSW]

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, lk = lane >> 4;

  const int nbn = (a.N + BN - 1) / BN;
  int lid = xcd_remap(blockIdx.x, gridDim.x);
  const int bn = lid % nbn;
  const int64_t bm = lid / nbn;
  const int64_t m0 = bm * BM;
  const int n0 = bn * BN;

  f32x4 acc[NREP][MREP];
#pragma unroll
  for (int n = 0; n < NREP; ++n)
#pragma unroll
    for (int m = 0; m < MREP; ++m) acc[n][m] = (f32x4){0.f, 0.f, 0.f, 0.f};

  float4 rx[NXV], rw[NWV];

  // Loads are unconditional (clamped addresses); out-of-range rows / the K tail are zeroed
  // when the registers are committed to LDS, one barrier later, so the compiler cannot sink a
  // load under its mask and every load of a stage is in flight at once.
  auto load4 = [&](const float *base, int64_t row, int64_t nrows, int k) -> float4 {
    const int64_t rc = row < nrows ? row : nrows - 1;
    float4 v;
    if constexpr (KVEC) {
      v = *reinterpret_cast<const float4 *>(base + rc * a.K + min(k, a.K - 4));
    } else {
      const float *p = base + rc * a.K;
      const int km = a.K - 1;
      v.x = p[min(k, km)];
      v.y = p[min(k + 1, km)];
      v.z = p[min(k + 2, km)];
      v.w = p[min(k + 3, km)];
    }
    return v;
  };
  auto mask4 = [&](float4 v, bool rok, int k) -> float4 {
    v.x = (rok && k < a.K) ? v.x : 0.f;
    v.y = (rok && k + 1 < a.K) ? v.y : 0.f;
    v.z = (rok && k + 2 < a.K) ? v.z : 0.f;
    v.w = (rok && k + 3 < a.K) ? v.w : 0.f;
    return v;
  };
  auto fetch = [&](int k0) {
#pragma unroll
    for (int it = 0; it < NXV; ++it) {
      const int e = min(tid + it * 256, BM * BK / 4 - 1);
      const int row = e / (BK / 4), c4 = e % (BK / 4);
      rx[it] = load4(a.x, m0 + row, a.M, k0 + c4 * 4);
    }
#pragma unroll
    for (int it = 0; it < NWV; ++it) {
      const int e = min(tid + it * 256, BN * BK / 4 - 1);
      const int row = e / (BK / 4), c4 = e % (BK / 4);
      rw[it] = load4(a.w, n0 + row, a.N, k0 + c4 * 4);
    }
  };
  auto commit = [&](int k0) {
#pragma unroll
    for (int it = 0; it < NXV; ++it) {
      const int e = tid + it * 256;
      if (e < BM * BK / 4) {
        const int row = e / (BK / 4), c4 = e % (BK / 4);
        *reinterpret_cast<float4 *>(&xs[row * RSW + c4 * 4]) = mask4(rx[it], m0 + row < a.M, k0 + c4 * 4);
      }
    }
#pragma unroll
    for (int it = 0; it < NWV; ++it) {
      const int e = tid + it * 256;
      if (e < BN * BK / 4) {
        const int row = e / (BK / 4), c4 = e % (BK / 4);
        *reinterpret_cast<float4 *>(&ws[row * RSW + c4 * 4]) = mask4(rw[it], n0 + row < a.N, k0 + c4 * 4);
      }
    }
  };

  const int nk = (a.K + BK - 1) / BK;
  fetch(0);
  for (int ks = 0; ks < nk; ++ks) {
    __syncthreads();
    commit(ks * BK);
    __syncthreads();
    if (ks + 1 < nk) fetch((ks + 1) * BK);
#pragma unroll
    for (int kk = 0; kk < BK / 16; ++kk) {
      f32x4 wa[NREP];
#pragma unroll
      for (int n = 0; n < NREP; ++n)
        wa[n] = *reinterpret_cast<const f32x4 *>(&ws[(wave * 16 * NREP + n * 16 + li) * RSW + kk * 16 + lk * 4]);
      // activation fragments in groups of 4 row-tiles to bound the live registers
#pragma unroll
      for (int mg = 0; mg < MREP; mg += 4) {
        f32x4 xb[4];
#pragma unroll
        for (int m = 0; m < 4; ++m)
          xb[m] = *reinterpret_cast<const f32x4 *>(&xs[((mg + m) * 16 + li) * RSW + kk * 16 + lk * 4]);
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int n = 0; n < NREP; ++n)
#pragma unroll
            for (int m = 0; m < 4; ++m) acc[n][mg + m] = ASX_MFMA(wa[n][j], xb[m][j], acc[n][mg + m]);
      }
    }
  }

  const bool nvec = (a.N & 3) == 0;
  // Fast path (block-uniform test): full tile, 16-byte aligned rows -> every load is
  // unconditional, so the residual / bias / scale loads of the whole tile pipeline.
  const bool full = nvec && (m0 + BM <= a.M) && (n0 + BN <= a.N);
  if (full) {
    f32x4 bz[NREP];
#pragma unroll
    for (int n = 0; n < NREP; ++n) {
      const int col = n0 + wave * 16 * NREP + n * 16 + lk * 4;
      bz[n] = (a.bias != nullptr) ? *reinterpret_cast<const f32x4 *>(a.bias + col) : (f32x4){0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int mg = 0; mg < MREP; mg += 4) {
      float sc[4], sh[4];
      f32x4 rs[4][NREP];
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        const int64_t row = m0 + (mg + m) * 16 + li;
        const int c = (int)((row / a.T) % a.C);
        sc[m] = a.scale ? a.scale[c] : 1.f;
        sh[m] = a.shift ? a.shift[c] : 0.f;
#pragma unroll
        for (int n = 0; n < NREP; ++n) {
          const int col = n0 + wave * 16 * NREP + n * 16 + lk * 4;
          rs[m][n] = (a.res != nullptr) ? *reinterpret_cast<const f32x4 *>(a.res + row * a.N + col)
                                        : (f32x4){0.f, 0.f, 0.f, 0.f};
        }
      }
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        const int64_t row = m0 + (mg + m) * 16 + li;
#pragma unroll
        for (int n = 0; n < NREP; ++n) {
          const int col = n0 + wave * 16 * NREP + n * 16 + lk * 4;
          const f32x4 v = acc[n][mg + m];
          f32x4 o;
          o.x = tdf_act(sc[m] * (v.x + bz[n].x) + sh[m], a.relu) + rs[m][n].x;
          o.y = tdf_act(sc[m] * (v.y + bz[n].y) + sh[m], a.relu) + rs[m][n].y;
          o.z = tdf_act(sc[m] * (v.z + bz[n].z) + sh[m], a.relu) + rs[m][n].z;
          o.w = tdf_act(sc[m] * (v.w + bz[n].w) + sh[m], a.relu) + rs[m][n].w;
          *reinterpret_cast<f32x4 *>(a.y + row * a.N + col) = o;
        }
      }
    }
    return;
  }
  // generic (ragged) path
#pragma unroll
  for (int m = 0; m < MREP; ++m) {
    const int64_t row = m0 + m * 16 + li;
    if (row >= a.M) continue;
    const int c = (int)((row / a.T) % a.C);
    const float sc = a.scale ? a.scale[c] : 1.f, sh = a.shift ? a.shift[c] : 0.f;
#pragma unroll
    for (int n = 0; n < NREP; ++n) {
      const int col = n0 + wave * 16 * NREP + n * 16 + lk * 4;
      if (col >= a.N) continue;
      f32x4 v = acc[n][m];
      float o[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float bzz = (a.bias != nullptr && col + r < a.N) ? a.bias[col + r] : 0.f;
        o[r] = tdf_act(sc * (o[r] + bzz) + sh, a.relu);
      }
      float *dst = a.y + row * a.N + col;
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (col + r < a.N) dst[r] = o[r] + (a.res != nullptr ? a.res[row * a.N + col + r] : 0.f);
    }
  }
}


// ===========================================================================
// LDS-DMA variants (the production path).  Both operands of a stage are copied
// global -> LDS with `global_load_lds_dwordx4` (no staging registers, no commit
// pass); LDS is double buffered and a stage's DMA is issued one whole compute
// phase before it is consumed, behind a single barrier per stage:
//
//     issue(stage 0)
//     for ci: wait vmcnt(0); barrier; issue(stage ci+1 -> other buffer); compute(stage ci)
//
// The LDS image of a DMA is lane-linear (wave-uniform base + lane*16 B), so
// padding is expressed through which lanes take part (EXEC-masked partial
// issues) and, for the row GEMM, through an XOR swizzle applied to the SOURCE
// address and again to the fragment read.  Out-of-range float4s are sourced from
// a zero page in global memory.  Requirements (checked by the launcher, which
// otherwise falls back to the register-staged kernels above): F % 4 == 0 for the
// conv, K % 4 == 0 for the row GEMM, 16-byte aligned base pointers.
// ===========================================================================
#define ASX_GLDS16(gptr, lptr)                                                               \
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(gptr),  \
                                   (__attribute__((address_space(3))) void *)(lptr), 16, 0, 0)

template <int KH_, int KW_, int S_, int PAD_, int NREP_, int KC_, int RPW_, int EPI_>
struct ConvDmaCfg {
  static constexpr int KH = KH_, KW = KW_, S = S_, PAD = PAD_, NREP = NREP_, KC = KC_, RPW = RPW_, EPI = EPI_;
  static constexpr int TH = 4 * RPW, TW = 64;
  static constexpr int IH = (TH - 1) * S + KH;
  static constexpr int LP = (PAD == 0) ? 0 : 4 - PAD;                    // left shift so rows start 16-B aligned
  static constexpr int IWA = ((LP + (TW - 1) * S + KW + 3) / 4) * 4;     // staged row width (floats)
  static constexpr int C4 = IWA / 4;
  static constexpr int SLOTS = IH * C4;                                  // float4 per plane
  static constexpr int NI = (SLOTS + 63) / 64;                           // wave-issues per plane
  static constexpr int PS0 = IH * IWA;
  // plane stride: stride-1 kernels fetch fragments with ds_read_b32 (32 banks: planes lk, lk+1 must sit 16 banks apart);
  // the stride-2 kernel fetches both dx taps of a pixel with one ds_read_b64 (64 banks per 32-lane group: the 16 pixels
  // of a fragment row cover banks 0..31, so the second plane of the group must start 32 banks further)
  static constexpr int PS = (S == 1) ? (PS0 + ((16 - PS0 % 32) + 32) % 32) : (PS0 + ((32 - PS0 % 64) + 64) % 64);
  static constexpr int NW = 16 * NREP;
  static constexpr int NWP = (NREP % 2 == 0) ? NW + 16 : NW;
  static constexpr int NTAP = KH * KW;
  static constexpr int W_ROWS = NTAP * KC;
  static constexpr int WSTAGE = ((W_ROWS * NWP + 255) / 256) * 256;      // packed weight floats per stage
  static constexpr int NWI = WSTAGE / 256;                               // weight wave-issues per stage
  static constexpr int BUF = KC * PS + WSTAGE;                           // floats per LDS buffer
  static constexpr int LDS_BYTES = 2 * BUF * 4;
  static constexpr int MREP = RPW * 4;
  static_assert(KC % 4 == 0 || (KC == 2 && S == 2 && KW == 2 && PAD == 0), "KC must be a multiple of 4 (2: only the 2x2 / stride-2 conv)");
  static_assert(PS % 4 == 0, "plane stride must keep 16-B alignment");
};

template <class CFG>
__global__ __launch_bounds__(256, 2) void conv_dma_kernel(ConvArgs a) {
  extern __shared__ float lds_f[];
  constexpr int KH = CFG::KH, KW = CFG::KW, S = CFG::S, PAD = CFG::PAD, NREP = CFG::NREP, KC = CFG::KC;
  constexpr int RPW = CFG::RPW, MREP = CFG::MREP, IWA = CFG::IWA, C4 = CFG::C4, PS = CFG::PS, LP = CFG::LP;
  constexpr int NWP = CFG::NWP, TH = CFG::TH, TW = CFG::TW, NI = CFG::NI, SLOTS = CFG::SLOTS;
  constexpr int BUF = CFG::BUF, WSTAGE = CFG::WSTAGE, NWI = CFG::NWI;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, lk = lane >> 4;

  int lid = xcd_remap(blockIdx.x, gridDim.x);
  const int cg = lid % a.CG;
  lid /= a.CG;
  const int tf = lid % a.tilesF;
  lid /= a.tilesF;
  const int tt = lid % a.tilesT;
  const int b = lid / a.tilesT;
  const int to0 = tt * TH, fo0 = tf * TW;
  const int ti0 = to0 * S - PAD, fa0 = fo0 * S - PAD - LP;   // aligned input origin

  const float *xb = a.x + (int64_t)b * a.x_bstride;
  const float *wg = a.wp + (int64_t)cg * a.NCI * WSTAGE;
  const int64_t plane_sz = (int64_t)a.T * a.F;

  // per-lane source offsets of this lane's NI slots inside a plane (same for every plane / stage)
  int sp_off[NI];
  bool sp_ok[NI];
#pragma unroll
  for (int j = 0; j < NI; ++j) {
    const int sidx = j * 64 + lane;
    const int row = sidx / C4, c4 = sidx - row * C4;
    const int t = ti0 + row, f = fa0 + c4 * 4;
    sp_ok[j] = (sidx < SLOTS) && t >= 0 && t < a.T && f >= 0 && f < a.F;
    sp_off[j] = t * a.F + f;
  }

  auto issue = [&](int ci, int buf) {
    float *in_s = lds_f + buf * BUF;
    float *w_s = in_s + KC * PS;
#pragma unroll
    for (int p = 0; p < (KC + 3) / 4; ++p) {
      const int pl = wave + 4 * p;
      if (KC % 4 != 0 && pl >= KC) continue;                 // two-channel stages: waves 0 and 1 carry the planes
      const int c = ci * KC + pl;
      const float *xc = xb + (int64_t)c * plane_sz;
      const bool cok = c < a.Cin;
#pragma unroll
      for (int j = 0; j < NI; ++j) {
        const float *src = (cok && sp_ok[j]) ? xc + sp_off[j] : a.zeros;
        if (j * 64 + lane < SLOTS) ASX_GLDS16(src, in_s + pl * PS + j * 256);
      }
    }
    const float *ws = wg + (int64_t)ci * WSTAGE;
#pragma unroll
    for (int i = 0; i < (NWI + 3) / 4; ++i) {
      const int q = wave + 4 * i;
      if (q < NWI) ASX_GLDS16(ws + q * 256 + lane * 4, w_s + q * 256);
    }
  };

  f32x4 acc[MREP][NREP];
#pragma unroll
  for (int m = 0; m < MREP; ++m)
#pragma unroll
    for (int n = 0; n < NREP; ++n) acc[m][n] = (f32x4){0.f, 0.f, 0.f, 0.f};

  issue(0, 0);
  for (int ci = 0; ci < a.NCI; ++ci) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (ci + 1 < a.NCI) issue(ci + 1, (ci + 1) & 1);
    const float *in_s = lds_f + (ci & 1) * BUF;
    const float *w_s = in_s + KC * PS;
    if constexpr (S == 2 && KW == 2 && KC == 2 && LP == 0) {
      // 2x2 / stride-2 conv, TWO-channel stages (18.7 KB per LDS buffer: four workgroups per CU instead of two -- the kernel
      // reads its input exactly once, so it needs HBM latency hidden, not LDS reuse).  One MFMA k-step = (2 channels) x
      // (2 dx taps): lane group lk reads channel lk >> 1, tap dx = lk & 1.  The 16 pixels x 2 taps of a channel are 32
      // consecutive floats and the second plane starts 32 banks further: conflict-free ds_read_b32.
      const int c2 = lk >> 1, dx2 = lk & 1;
#pragma unroll
      for (int dy = 0; dy < KH; ++dy) {
        float bf[NREP];
#pragma unroll
        for (int n = 0; n < NREP; ++n) bf[n] = w_s[((dy * KW + dx2) * KC + c2) * NWP + n * 16 + li];
        float af[MREP];
#pragma unroll
        for (int m = 0; m < MREP; ++m) {
          const int rr = m >> 2, cc = m & 3;
          af[m] = in_s[c2 * PS + ((wave * RPW + rr) * S + dy) * IWA + (cc * 16 + li) * S + dx2];
        }
#pragma unroll
        for (int m = 0; m < MREP; ++m) {
#ifdef ASX_ABL_UPDOWN   // timing probe (wrong results): half of the MFMAs of the down / up convs
          if (m & 1) continue;
#endif
#pragma unroll
          for (int n = 0; n < NREP; ++n) acc[m][n] = ASX_MFMA(af[m], bf[n], acc[m][n]);
        }
      }
    } else if constexpr (S == 2 && KW == 2 && KC == 4 && LP == 0) {
      // 2x2 / stride-2 conv: the two dx taps of an output pixel are adjacent floats -> one conflict-free ds_read_b64 per
      // (row tap, pixel) instead of two 2-way conflicting ds_read_b32 (SQ_LDS_BANK_CONFLICT was 42 % of the LDS cycles);
      // tap order (dy, dx) and therefore the accumulation order are unchanged
#pragma unroll
      for (int dy = 0; dy < KH; ++dy) {
        f32x2 a2[MREP];
#pragma unroll
        for (int m = 0; m < MREP; ++m) {
          const int rr = m >> 2, cc = m & 3;
          a2[m] = *reinterpret_cast<const f32x2 *>(&in_s[lk * PS + ((wave * RPW + rr) * S + dy) * IWA + (cc * 16 + li) * S]);   // ds_read_b64
        }
#pragma unroll
        for (int dx = 0; dx < 2; ++dx) {
          float bf[NREP];
#pragma unroll
          for (int n = 0; n < NREP; ++n) bf[n] = w_s[((dy * KW + dx) * KC + lk) * NWP + n * 16 + li];
#pragma unroll
          for (int m = 0; m < MREP; ++m)
#pragma unroll
            for (int n = 0; n < NREP; ++n) acc[m][n] = ASX_MFMA(dx ? a2[m].y : a2[m].x, bf[n], acc[m][n]);
        }
      }
    } else {
#pragma unroll
      for (int tap = 0; tap < KH * KW; ++tap) {
        const int dy = tap / KW, dx = tap % KW;
#pragma unroll
        for (int kq = 0; kq < KC / 4; ++kq) {
          float bf[NREP];
#pragma unroll
          for (int n = 0; n < NREP; ++n) bf[n] = w_s[(tap * KC + kq * 4 + lk) * NWP + n * 16 + li];
          float af[MREP];
#pragma unroll
          for (int m = 0; m < MREP; ++m) {
            const int rr = m >> 2, cc = m & 3;
            af[m] = in_s[(kq * 4 + lk) * PS + ((wave * RPW + rr) * S + dy) * IWA + LP + (cc * 16 + li) * S + dx];
          }
#pragma unroll
          for (int m = 0; m < MREP; ++m) {
#ifdef ASX_ABL_UPDOWN
            if (CFG::EPI == 1 && (m & 1)) continue;
#endif
#pragma unroll
            for (int n = 0; n < NREP; ++n) acc[m][n] = ASX_MFMA(af[m], bf[n], acc[m][n]);
          }
        }
      }
    }
  }

  conv_epilogue<CFG>(a, acc, b, cg, to0, fo0, wave, li, lk);
}

// ---------------------------------------------------------------------------
// Row GEMM with LDS-DMA staging.  Tiles are stored UNPADDED, [rows][32 floats],
// with the 16-byte chunk index XOR-swizzled by g(row) = (row >> 1) & 7 (applied to
// the source address of the DMA and to the fragment read), which makes every
// ds_read_b128 lane group hit 16 distinct 4-bank slots.
// ---------------------------------------------------------------------------
struct TdfDmaArgs {
  const float *x, *w, *bias, *scale, *shift, *res, *zeros;   // scale/shift may be nullptr (identity)
  float *y;
  int64_t M;
  int N, K, C, T;
  int relu;                 // activation enum of tdf_act()
  int64_t lda, ldy, ldr;    // row strides (floats) of x, y, res; 0 = dense (K, N, N)
  // optional rotary embedding on output columns [0, rot_cols) (the q | k thirds of a Roformer qkv projection, applied in the
  // epilogue instead of a separate in-place pass): interleaved pairs (2i, 2i + 1) of every rot_half * 2 wide head are rotated
  // by rot_tab[pos(row)][i] = (cos, sin), pos(row) = (row / rot_pos_div) % rot_pos_mod.  rot_tab == nullptr: none.
  const float2 *rot_tab;
  int rot_cols, rot_half, rot_pos_mod;
  int64_t rot_pos_div;
  // optional per-ROW factor applied to the accumulator before the bias: y = act(acc * rscale[row] + bias) -- an RMSNorm in
  // front of the projection folded into it (gamma is folded into W by the loader, rscale = sqrt(d) / max(|x_row|, eps)).
  // Honoured by the generic-activation epilogues (not by the ReLU + residual path of the TDF layers).
  const float *rscale;
  int nt;                   // bit 0: non-temporal output stores, bit 1: non-temporal residual loads (ReLU path; A/B switch ASX_NT)
  int prefer_small;         // launcher hint: 64-row tiles (three workgroups per CU) -- the Demucs transformer linears (M ~ 1e5 rows)
  int tile_map;             // tdf3_kernel tile -> workgroup map: 0 = column tiles partitioned over the XCDs (kernels_gemm2.h), 1 = the column
                            // tiles of a row block consecutive on ONE XCD (x fetched once, W streamed by every XCD), 2 = 8 x 8 super-tiles
                            // of (row block, column tile) per XCD
  int glu_cout;             // tdf3_kernel GATHER mode, 128-column tiles only: > 0 = GLU epilogue over value / gate fragment pairs (the rows of W are in
                            // ht_glu_perm order), glu_cout output channels: y[row][c / 2 + ...] = (acc_v + b) * sigmoid(acc_g + b)
  // PAIR IMAGES (tdf3_kernel, fp16 x 3 arithmetic; kernels_gemm3.h "operands split by their producer"): a matrix whose only reader is a
  // row GEMM is stored by its producer as the two fp16 parts the GEMM multiplies -- every 16 bytes hold four consecutive elements as
  // h0 h1 h2 h3 l0 l1 l2 l3 (x 2^e = h + l), same row pitch as the fp32 matrix it replaces -- with one exponent per (row, column span)
  // in an int table beside it.
  const int *xexp;          // != nullptr: x IS a pair image; xexp[row * xexp_n + span] = e of columns [span * 32 xexp_gs, ...)
  int xexp_n, xexp_gs;      // spans per row, 32-column stages per span
  int xexp_inv;             // ceil(65536 / xexp_gs): stage -> span as (stage * xexp_inv) >> 16 (the launcher checks every stage of K)
  int *yexp;                // != nullptr: y is WRITTEN as a pair image, one span per column tile of the launch: yexp[row * yexp_n + tile]
  int yexp_n;               // (no residual, no rotary epilogue; N % 32 == 0)
};

// rotary step of the row-GEMM epilogues on the float4 (row, col .. col + 3), col % 4 == 0.  Every product and sum is rounded
// explicitly (no compiler-chosen fma contraction), so the full-tile and the ragged-tile epilogues of every row-GEMM kernel
// agree bit for bit and results stay independent of the batch size.
__device__ __forceinline__ f32x4 tdf_rot4(const TdfDmaArgs &a, f32x4 o, int64_t row, int col) {
  if (a.rot_tab == nullptr || col >= a.rot_cols) return o;
  const uint32_t pos = ((uint32_t)row / (uint32_t)a.rot_pos_div) % (uint32_t)a.rot_pos_mod;   // launchers: M < 2^31 with a rotary epilogue
  const int i = (col % (2 * a.rot_half)) >> 1;
  const float2 *t = a.rot_tab + (int64_t)pos * a.rot_half + i;
  const float2 c0 = t[0], c1 = t[1];
  // Every product and sum is its own VALU instruction, kept apart by empty asm statements.  Left alone, hipcc packs the eight operations
  // into v_pk_mul_f32 / v_pk_fma_f32 pairs that read the just-loaded table registers; in tdf3_kernel's fp16 x 3 build that sequence
  // returned a wrong FIRST component in lanes 48-63 of a wave about once per 4e4 stores -- different elements on every run (whole
  // output compared bit for bit, tools/proto_gemm3.hip "rof qkv rotary"; with the table replaced by constants, or with the
  // operations kept scalar as here, two runs agree in all 76 M elements).  The same kernel source in other builds had shown the
  // symptom once before (round 5: BS-Roformer chunk test 3e-5 .. 7e-5 off on two boxes with one build, unexplained then).
#ifdef ASX_ROT_PACKED   // the form that misbehaves, for tools/proto_gemm3.hip only (hipcc -DASX_ROT_PACKED ...; shape "rof qkv rotary", full compare)
  f32x4 rp;
  rp.x = __fmaf_rn(o.x, c0.x, -__fmul_rn(o.y, c0.y));
  rp.y = __fmaf_rn(o.y, c0.x, __fmul_rn(o.x, c0.y));
  rp.z = __fmaf_rn(o.z, c1.x, -__fmul_rn(o.w, c1.y));
  rp.w = __fmaf_rn(o.w, c1.x, __fmul_rn(o.z, c1.y));
  return rp;
#endif
  float p0 = __fmul_rn(o.y, c0.y), p1 = __fmul_rn(o.x, c0.y), p2 = __fmul_rn(o.w, c1.y), p3 = __fmul_rn(o.z, c1.y);
  asm volatile("" : "+v"(p0));
  asm volatile("" : "+v"(p1));
  asm volatile("" : "+v"(p2));
  asm volatile("" : "+v"(p3));
  f32x4 r;
  r.x = __fmaf_rn(o.x, c0.x, -p0);
  asm volatile("" : "+v"(r.x));
  r.y = __fmaf_rn(o.y, c0.x, p1);
  asm volatile("" : "+v"(r.y));
  r.z = __fmaf_rn(o.z, c1.x, -p2);
  asm volatile("" : "+v"(r.z));
  r.w = __fmaf_rn(o.w, c1.x, p3);
  return r;
}

template <int NREP, int MREP, int BK_ = 32>
struct TdfDmaCfg {
  static constexpr int BK = BK_;                       // 32 (two workgroups per CU) or 64 (one, longer stages)
  static constexpr int CPR = BK / 4;                   // 16-byte chunks per row
  static constexpr int RPI = 64 / CPR;                 // rows per wave-issue
  static constexpr int BM = 16 * MREP, BN = 64 * NREP;
  static constexpr int BUF = (BM + BN) * BK;           // floats per LDS buffer
  static constexpr int LDS_BYTES = 2 * BUF * 4;
  static constexpr int NXI = BM / RPI, NWI = BN / RPI; // wave-issues per stage
  // chunk swizzle g(row): BK=32 rows are 128 B (two rows per bank sweep), BK=64 rows are a full 256 B sweep
  __device__ static constexpr int g(int row) { return BK == 32 ? ((row >> 1) & 7) : (row & 15); }
};

template <int NREP, int MREP, int BK_ = 32>
__global__ __launch_bounds__(256, (BK_ == 32 ? 2 : 1)) void tdf_dma_kernel(TdfDmaArgs a) {
  using CFG = TdfDmaCfg<NREP, MREP, BK_>;
  constexpr int BK = CFG::BK, BM = CFG::BM, BN = CFG::BN, BUF = CFG::BUF, NXI = CFG::NXI, NWI = CFG::NWI;
  constexpr int CPR = CFG::CPR, RPI = CFG::RPI;
  extern __shared__ float lds_f[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, lk = lane >> 4;

  const int nbn = (a.N + BN - 1) / BN;
  int lid = xcd_remap(blockIdx.x, gridDim.x);
  const int bn = lid % nbn;
  const int64_t bm = lid / nbn;
  const int64_t m0 = bm * BM;
  const int n0 = bn * BN;
  const int64_t lda = a.lda ? a.lda : a.K, ldy = a.ldy ? a.ldy : a.N, ldr = a.ldr ? a.ldr : a.N;

  // DMA lane role: row-in-issue = lane / CPR, physical chunk p = lane % CPR
  const int lr = lane / CPR, lp = lane % CPR;

  auto issue = [&](int k0, int buf) {
    float *xs = lds_f + buf * BUF;
    float *ws = xs + BM * BK;
#pragma unroll
    for (int i = 0; i < (NXI + 3) / 4; ++i) {
      const int q = wave + 4 * i;             // issue index -> rows RPI*q .. RPI*q + RPI - 1
      if (q < NXI) {
        const int row = q * RPI + lr;
        const int c = lp ^ CFG::g(row);        // logical chunk fetched into physical slot lp
        const int k = k0 + c * 4;
        const bool ok = (m0 + row < a.M) && (k < a.K);
        const float *src = ok ? a.x + (m0 + row) * lda + k : a.zeros;
        ASX_GLDS16(src, xs + q * 256);
      }
    }
#pragma unroll
    for (int i = 0; i < (NWI + 3) / 4; ++i) {
      const int q = wave + 4 * i;
      if (q < NWI) {
        const int row = q * RPI + lr;
        const int c = lp ^ CFG::g(row);
        const int k = k0 + c * 4;
        const bool ok = (n0 + row < a.N) && (k < a.K);
        const float *src = ok ? a.w + (int64_t)(n0 + row) * a.K + k : a.zeros;
        ASX_GLDS16(src, ws + q * 256);
      }
    }
  };

  f32x4 acc[NREP][MREP];
#pragma unroll
  for (int n = 0; n < NREP; ++n)
#pragma unroll
    for (int m = 0; m < MREP; ++m) acc[n][m] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int nk = (a.K + BK - 1) / BK;
  const int sw = CFG::g(li);      // fragment rows are (16*tile + li) and g only looks at the low 4 bits
  issue(0, 0);
  for (int ks = 0; ks < nk; ++ks) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (ks + 1 < nk) issue((ks + 1) * BK, (ks + 1) & 1);
    const float *xs = lds_f + (ks & 1) * BUF;
    const float *ws = xs + BM * BK;
#pragma unroll
    for (int kk = 0; kk < BK / 16; ++kk) {
      const int pc = ((kk * 4 + lk) ^ sw) * 4;   // swizzled chunk -> float offset inside the row
      f32x4 wa[NREP];
#pragma unroll
      for (int n = 0; n < NREP; ++n)
        wa[n] = *reinterpret_cast<const f32x4 *>(&ws[(wave * 16 * NREP + n * 16 + li) * BK + pc]);
#pragma unroll
      for (int mg = 0; mg < MREP; mg += 4) {
        f32x4 xb[4];
#pragma unroll
        for (int m = 0; m < 4; ++m)
          xb[m] = *reinterpret_cast<const f32x4 *>(&xs[((mg + m) * 16 + li) * BK + pc]);
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int n = 0; n < NREP; ++n)
#pragma unroll
            for (int m = 0; m < 4; ++m) acc[n][mg + m] = ASX_MFMA(wa[n][j], xb[m][j], acc[n][mg + m]);
      }
    }
  }

  const bool nvec = (a.N & 3) == 0;
  const bool full = nvec && (m0 + BM <= a.M) && (n0 + BN <= a.N);
  if (full) {
    f32x4 bz[NREP];
#pragma unroll
    for (int n = 0; n < NREP; ++n) {
      const int col = n0 + wave * 16 * NREP + n * 16 + lk * 4;
      bz[n] = (a.bias != nullptr) ? *reinterpret_cast<const f32x4 *>(a.bias + col) : (f32x4){0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int mg = 0; mg < MREP; mg += 4) {
      float sc[4], sh[4], rw[4];
      f32x4 rs[4][NREP];
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        const int64_t row = m0 + (mg + m) * 16 + li;
        const int c = (int)((row / a.T) % a.C);
        sc[m] = a.scale ? a.scale[c] : 1.f;
        sh[m] = a.shift ? a.shift[c] : 0.f;
        rw[m] = a.rscale ? a.rscale[row] : 1.f;
#pragma unroll
        for (int n = 0; n < NREP; ++n) {
          const int col = n0 + wave * 16 * NREP + n * 16 + lk * 4;
          rs[m][n] = (a.res != nullptr) ? *reinterpret_cast<const f32x4 *>(a.res + row * ldr + col)
                                        : (f32x4){0.f, 0.f, 0.f, 0.f};
        }
      }
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        const int64_t row = m0 + (mg + m) * 16 + li;
#pragma unroll
        for (int n = 0; n < NREP; ++n) {
          const int col = n0 + wave * 16 * NREP + n * 16 + lk * 4;
          const f32x4 v = acc[n][mg + m];
          f32x4 o;
          // acc * rw + bias as ONE explicitly rounded fma in every epilogue variant (rw = 1: exactly acc + bias)
          o.x = tdf_act(sc[m] * __fmaf_rn(v.x, rw[m], bz[n].x) + sh[m], a.relu) + rs[m][n].x;
          o.y = tdf_act(sc[m] * __fmaf_rn(v.y, rw[m], bz[n].y) + sh[m], a.relu) + rs[m][n].y;
          o.z = tdf_act(sc[m] * __fmaf_rn(v.z, rw[m], bz[n].z) + sh[m], a.relu) + rs[m][n].z;
          o.w = tdf_act(sc[m] * __fmaf_rn(v.w, rw[m], bz[n].w) + sh[m], a.relu) + rs[m][n].w;
          *reinterpret_cast<f32x4 *>(a.y + row * ldy + col) = tdf_rot4(a, o, row, col);
        }
      }
    }
    return;
  }
#pragma unroll
  for (int m = 0; m < MREP; ++m) {
    const int64_t row = m0 + m * 16 + li;
    if (row >= a.M) continue;
    const int c = (int)((row / a.T) % a.C);
    const float sc = a.scale ? a.scale[c] : 1.f, sh = a.shift ? a.shift[c] : 0.f;
    const float rw = a.rscale ? a.rscale[row] : 1.f;
#pragma unroll
    for (int n = 0; n < NREP; ++n) {
      const int col = n0 + wave * 16 * NREP + n * 16 + lk * 4;
      if (col >= a.N) continue;
      f32x4 v = acc[n][m];
      float o[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float bzz = (a.bias != nullptr && col + r < a.N) ? a.bias[col + r] : 0.f;
        o[r] = tdf_act(sc * __fmaf_rn(o[r], rw, bzz) + sh, a.relu);
      }
      float *dst = a.y + row * ldy + col;
      if (a.rot_tab != nullptr) {                     // launcher: N % 4 == 0 and no residual with a rotary epilogue
        const f32x4 q = tdf_rot4(a, (f32x4){o[0], o[1], o[2], o[3]}, row, col);
        o[0] = q.x;
        o[1] = q.y;
        o[2] = q.z;
        o[3] = q.w;
      }
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (col + r < a.N) dst[r] = o[r] + (a.res != nullptr ? a.res[row * ldr + col + r] : 0.f);
    }
  }
}


// ===========================================================================
// Winograd F(2x2, 3x3) variant of the 3x3 / pad-1 convolution (optional, fp32).
//
//   Y = A^T [ sum_cin (G g G^T) .* (B^T d B) ] A        (Lavin & Gray 2016)
//
// 2.25x fewer multiply-accumulates than the direct form; the sixteen transform-domain products are
// sixteen independent [tiles x cin] x [cin x cout] GEMMs on v_mfma_f32_16x16x4_f32.
// Workgroup: 64 output tiles (4 tile-rows x 16 tile-cols = 8 x 32 pixels) x 48 output channels, 4 waves;
// wave w owns tile-row w (one MFMA M tile) for ALL 16 transform positions x 3 N tiles = 48 accumulator
// tiles (192 VGPRs), so the output transform A^T m A is lane-local.
// Per 8-channel stage: the haloed raw input tile (10 x 40, rows 16-B aligned) and the pre-transformed
// weights U (host, fp64) arrive by LDS-DMA (double buffered); each thread transforms one
// (tile, channel pair) into V; fragments are fetched with ds_read_b64, the two k slots a lane
// supplies being channels (2*lk, 2*lk + 1) for both operands.
// LDS: 2 x (raw 12.8 KB + U 24 KB) + V 40 KB = 113.6 KB -> one workgroup per CU.
// ===========================================================================
struct WinoCfg {
  static constexpr int KC = 8, NREP = 3, NW = 48;
  static constexpr int TR = 4, TC = 16;              // tile rows / cols per workgroup
  static constexpr int TH = 2 * TR, TW = 2 * TC;     // 8 x 32 output pixels
  static constexpr int IH = TH + 2, LP = 3, IWA = 40, C4 = IWA / 4;
  static constexpr int SLOTS = IH * C4;              // 100 float4 per plane
  static constexpr int NI = (SLOTS + 63) / 64;       // 2 wave-issues per plane
  static constexpr int PS = IH * IWA;                // 400 floats per raw plane
  static constexpr int RAW = KC * PS;                // 3200
  static constexpr int USTAGE = 16 * 4 * NW * 2;     // [xi][pair][cout][2] = 6144 floats
  static constexpr int UWI = USTAGE / 256;           // 24 wave-issues
  static constexpr int VPS = 64 * 2 + 32;            // pair-plane stride (floats), +32 keeps lk = 0/1 on different bank halves
  static constexpr int VSZ = 16 * 4 * VPS;           // 10240 floats
  static constexpr int BUF = RAW + USTAGE;           // per DMA buffer
  static constexpr int LDS_BYTES = (2 * BUF + VSZ) * 4;
};

__device__ __forceinline__ void wino_bt(float a, float b, float c, float d, float *o) {
  o[0] = a - c;
  o[1] = b + c;
  o[2] = c - b;
  o[3] = b - d;
}

#ifdef ASX_EXPERIMENTAL_KERNELS   // first Winograd generation: measured, superseded (profiles/NOTES.md); not in the default build
__global__ __launch_bounds__(256) void conv_wino_kernel(ConvArgs a) {
  using CFG = WinoCfg;
  extern __shared__ float lds_f[];
  float *Vs = lds_f + 2 * CFG::BUF;
  constexpr int KC = CFG::KC, NREP = CFG::NREP, NW = CFG::NW, IWA = CFG::IWA, C4 = CFG::C4, PS = CFG::PS, LP = CFG::LP;
  constexpr int NI = CFG::NI, SLOTS = CFG::SLOTS, VPS = CFG::VPS;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, lk = lane >> 4;

  int lid = xcd_remap(blockIdx.x, gridDim.x);
  const int cg = lid % a.CG;
  lid /= a.CG;
  const int tf = lid % a.tilesF;
  lid /= a.tilesF;
  const int tt = lid % a.tilesT;
  const int b = lid / a.tilesT;
  const int to0 = tt * CFG::TH, fo0 = tf * CFG::TW;
  const int ti0 = to0 - 1, fa0 = fo0 - 1 - LP;   // aligned input origin (fo0 % 32 == 0 -> fa0 % 4 == 0)

  const float *xb = a.x + (int64_t)b * a.x_bstride;
  const float *ug = a.wp + (int64_t)cg * a.NCI * CFG::USTAGE;
  const int64_t plane_sz = (int64_t)a.T * a.F;

  int sp_off[NI];
  bool sp_ok[NI];
#pragma unroll
  for (int j = 0; j < NI; ++j) {
    const int sidx = j * 64 + lane;
    const int row = sidx / C4, c4 = sidx - row * C4;
    const int t = ti0 + row, f = fa0 + c4 * 4;
    sp_ok[j] = (sidx < SLOTS) && t >= 0 && t < a.T && f >= 0 && f < a.F;
    sp_off[j] = t * a.F + f;
  }

  auto issue = [&](int ci, int buf) {
    float *raw = lds_f + buf * CFG::BUF;
    float *us = raw + CFG::RAW;
#pragma unroll
    for (int p = 0; p < KC / 4; ++p) {
      const int pl = wave + 4 * p;
      const int c = ci * KC + pl;
      const float *xc = xb + (int64_t)c * plane_sz;
      const bool cok = c < a.Cin;
#pragma unroll
      for (int j = 0; j < NI; ++j) {
        const float *src = (cok && sp_ok[j]) ? xc + sp_off[j] : a.zeros;
        if (j * 64 + lane < SLOTS) ASX_GLDS16(src, raw + pl * PS + j * 256);
      }
    }
    const float *ws = ug + (int64_t)ci * CFG::USTAGE;
#pragma unroll
    for (int i = 0; i < CFG::UWI / 4; ++i) {
      const int q = wave + 4 * i;
      ASX_GLDS16(ws + q * 256 + lane * 4, us + q * 256);
    }
  };

  f32x4 acc[16][NREP];
#pragma unroll
  for (int x = 0; x < 16; ++x)
#pragma unroll
    for (int n = 0; n < NREP; ++n) acc[x][n] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // input-transform role: one (tile, channel pair) per thread; tile = lane, pair = wave
  const int ty = lane >> 4, tx = lane & 15;

  issue(0, 0);
  for (int ci = 0; ci < a.NCI; ++ci) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();   // stage ci landed; every wave is done with V of stage ci-1
    if (ci + 1 < a.NCI) issue(ci + 1, (ci + 1) & 1);
    const float *raw = lds_f + (ci & 1) * CFG::BUF;
    const float *us = raw + CFG::RAW;
    // ---- V = B^T d B for this thread's (tile, 2 channels) ----
    {
      float v[2][16];
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const float *pl = raw + (2 * wave + q) * PS + (2 * ty) * IWA + LP + 2 * tx;
        float r[4][4];   // r[col][a] = (B^T d)[a][col]
#pragma unroll
        for (int j = 0; j < 4; ++j) wino_bt(pl[j], pl[IWA + j], pl[2 * IWA + j], pl[3 * IWA + j], r[j]);
#pragma unroll
        for (int x = 0; x < 4; ++x) wino_bt(r[0][x], r[1][x], r[2][x], r[3][x], &v[q][4 * x]);
      }
#pragma unroll
      for (int x = 0; x < 16; ++x)
        *reinterpret_cast<float2 *>(&Vs[(x * 4 + wave) * VPS + lane * 2]) = make_float2(v[0][x], v[1][x]);
    }
    __syncthreads();
    // ---- sixteen GEMMs: M_xi[tile, cout] += V_xi[tile, cin] U_xi[cin, cout] ----
#pragma unroll
    for (int x = 0; x < 16; ++x) {
      const float2 av = *reinterpret_cast<const float2 *>(&Vs[(x * 4 + lk) * VPS + (wave * 16 + li) * 2]);
      float2 bv[NREP];
#pragma unroll
      for (int n = 0; n < NREP; ++n)
        bv[n] = *reinterpret_cast<const float2 *>(&us[((x * 4 + lk) * NW + n * 16 + li) * 2]);
#pragma unroll
      for (int n = 0; n < NREP; ++n) {
        acc[x][n] = ASX_MFMA(av.x, bv[n].x, acc[x][n]);
        acc[x][n] = ASX_MFMA(av.y, bv[n].y, acc[x][n]);
      }
    }
  }

  // ---- Y = A^T m A, bias, activation, store: lane holds tiles (tile-row = wave, tile-col = 4*lk + r) of cout li ----
  float *yb = a.y + (int64_t)b * a.y_bstride;
  const float *rb = a.res ? a.res + (int64_t)b * a.aux_bstride : nullptr;
  const int t0 = to0 + 2 * wave;
  const int f0 = fo0 + 8 * lk;
  const bool full = ((a.Fo & 3) == 0) && (to0 + CFG::TH <= a.To) && (fo0 + CFG::TW <= a.Fo);
#pragma unroll
  for (int n = 0; n < NREP; ++n) {
    const int co = cg * NW + n * 16 + li;
    const float bv = a.bias[co];
    if (co >= a.Cout) continue;
    float o[2][8];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float c[4][2];   // c[col][p] = (A^T m)[p][col]
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float m0 = acc[j][n][r], m1 = acc[4 + j][n][r], m2 = acc[8 + j][n][r], m3 = acc[12 + j][n][r];
        c[j][0] = m0 + m1 + m2;
        c[j][1] = m1 - m2 - m3;
      }
#pragma unroll
      for (int pq = 0; pq < 2; ++pq) {
        o[pq][2 * r] = c[0][pq] + c[1][pq] + c[2][pq];
        o[pq][2 * r + 1] = c[1][pq] - c[2][pq] - c[3][pq];
      }
    }
#pragma unroll
    for (int pq = 0; pq < 2; ++pq) {
      const int t = t0 + pq;
      const int64_t off = ((int64_t)co * a.To + t) * a.Fo + f0;
#pragma unroll
      for (int q = 0; q < 8; ++q) o[pq][q] = act_fn(o[pq][q] + bv, a.act);
      if (full) {
        f32x4 v0 = {o[pq][0], o[pq][1], o[pq][2], o[pq][3]}, v1 = {o[pq][4], o[pq][5], o[pq][6], o[pq][7]};
        if (rb != nullptr) {
          v0 += *reinterpret_cast<const f32x4 *>(rb + off);
          v1 += *reinterpret_cast<const f32x4 *>(rb + off + 4);
        }
        *reinterpret_cast<f32x4 *>(yb + off) = v0;
        *reinterpret_cast<f32x4 *>(yb + off + 4) = v1;
      } else if (t < a.To) {
#pragma unroll
        for (int q = 0; q < 8; ++q)
          if (f0 + q < a.Fo) yb[off + q] = o[pq][q] + (rb != nullptr ? rb[off + q] : 0.f);
      }
    }
  }
}
#endif

}  // namespace asx
