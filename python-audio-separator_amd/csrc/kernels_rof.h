// BS-Roformer kernels for gfx950 (uvr_lib_v5/roformer/bs_roformer.py, attend.py).
//
// Tokens live as rows of a [M, D] matrix, M = B * T * Fb ordered (b, t, band); every linear is
// the row GEMM of kernels_net.h.  The axial attention never permutes the token matrix: a
// sequence is (base row, row stride, length) -- time attention walks rows with stride Fb,
// frequency attention walks consecutive rows.
#pragma once
#include "kernels_gemm3.h"   // split3_pair / split3_oct, bf16x8, ASX_MFMA_BF16 (the bf16 x 6 attention kernels)
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace asx {

// ---------------------------------------------------------------------------
// RMSNorm (bs_roformer.py:42-52): y = x / max(||x||_2, 1e-12) * sqrt(d) * gamma, one wave per row.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void rmsnorm_kernel(const float *__restrict__ x, int64_t lda, int d,
                                                      const float *__restrict__ gamma, float *__restrict__ y,
                                                      int64_t ldy, int64_t M) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  const float *xp = x + row * lda;
  float ss = 0.f;
  for (int i = lane; i < d; i += 64) {
    const float v = xp[i];
    ss += v * v;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) ss += __shfl_xor(ss, off);
  const float denom = fmaxf(sqrtf(ss), 1e-12f);
  const float scale = sqrtf((float)d);
  float *yp = y + row * ldy;
  for (int i = lane; i < d; i += 64) yp[i] = xp[i] / denom * scale * gamma[i];
}

// ---------------------------------------------------------------------------
// The row factor of an RMSNorm folded into the projections that consume it (TdfDmaArgs::rscale):
// r[row] = sqrt(d) / max(||x_row||_2, 1e-12); the sum of squares is accumulated exactly as in rmsnorm_kernel.  One wave per
// row: the row is read once and nothing but one float is written.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void rownorm_kernel(const float *__restrict__ x, int64_t lda, int d, float *__restrict__ r,
                                                      int64_t M) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  const float *xp = x + row * lda;
  float ss = 0.f;
  for (int i = lane; i < d; i += 64) {
    const float v = xp[i];
    ss += v * v;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) ss += __shfl_xor(ss, off);
  if (lane == 0) r[row] = sqrtf((float)d) / fmaxf(sqrtf(ss), 1e-12f);
}

// ---------------------------------------------------------------------------
// Rotary embedding on the q and k thirds of the qkv matrix, in place
// (rotary_embedding_torch.apply_rotary_emb: t*cos + rotate_half(t)*sin, interleaved pairs).
// tab[pos][i] = (cos, sin) of float32(pos) * freqs[i];  pos(row) = (row / pos_div) % pos_mod.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void rotary_kernel(float *__restrict__ qkv, int64_t ld, int64_t M, int heads, int dh,
                                                     const float2 *__restrict__ tab, int64_t pos_div, int pos_mod) {
  const int half = dh / 2;
  const int per_row = 2 * heads * half;  // q and k pairs
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= M * per_row) return;
  const int64_t row = idx / per_row;
  const int r = (int)(idx - row * per_row);
  const int which = r / (heads * half);          // 0 = q, 1 = k
  const int hp = r - which * heads * half;
  const int h = hp / half, i = hp - h * half;
  const int pos = (int)((row / pos_div) % pos_mod);
  const float2 cs = tab[(int64_t)pos * half + i];
  float2 *p = reinterpret_cast<float2 *>(qkv + row * ld + (int64_t)which * heads * dh + h * dh + 2 * i);
  const float2 v = *p;
  *p = make_float2(v.x * cs.x - v.y * cs.y, v.y * cs.x + v.x * cs.y);
}

// ---------------------------------------------------------------------------
// Softmax attention with per-head sigmoid gates (bs_roformer.py:86-103, attend.py:100-112):
//   out[i, h, :] = softmax_j(q_i . k_j * dh^-0.5) v_j * sigmoid(gate[i, h])
// dh = 64, fp32 MFMA 16x16x4.  Workgroup = 64 queries of one (sequence, head); 4 waves x 16 queries.
// Per 64-key tile:  S^T = K Q^T  (keys on the MFMA M axis, queries on N), online softmax per
// query column (registers + 2 cross-lane steps), then O^T += V^T P^T with P^T taken straight from
// the S^T accumulator registers: lane (li, lk) holds P[query li][key 4*lk + r], which is exactly the
// B operand of step r when the key order inside a 16-key tile is permuted the same way for V.
// ---------------------------------------------------------------------------
struct AttnArgs {
  const float *qkv;   // [M, 3*heads*64]  (q | k | v), rotary already applied
  const float *gate;  // [M, heads] pre-sigmoid
  float *out;         // [M, heads*64]
  int heads;
  int gate_ld;        // row stride of gate
  int len;            // sequence length
  int64_t row_stride; // rows between consecutive sequence positions
  int64_t inner_cnt;  // sequence s -> base row = (s / inner_cnt) * outer_stride + (s % inner_cnt) * inner_stride
  int64_t outer_stride, inner_stride;
  float scale;
  int exact;          // 1: libm expf in the softmax; 0: the hardware exp unit (2 ulp, ASX_ATTN_EXACT unset)
  int nqt;            // attention2_kernel: query tiles per sequence (its grid is nqt * heads * sequences workgroups, 1-D)
};

constexpr int ATT_QS = 66;  // LDS row strides (floats): Q/K == 2 (mod 32), V == 4 (mod 8)
constexpr int ATT_VS = 68;
constexpr int ATT_LDS_BYTES = (64 * ATT_QS * 2 + 64 * ATT_VS) * 4;

#ifdef ASX_EXPERIMENTAL_KERNELS   // first attention generation (ASX_ATTN_V1): superseded by attention2_kernel / attention6_kernel; not in the default build
__global__ __launch_bounds__(256) void attention_kernel(AttnArgs a) {
  __shared__ float lds[64 * ATT_QS * 2 + 64 * ATT_VS];
  float *Qs = lds, *Ks = lds + 64 * ATT_QS, *Vs = lds + 128 * ATT_QS;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, lk = lane >> 4;
  const int qt = blockIdx.x, h = blockIdx.y;
  const int64_t sq = blockIdx.z;
  const int64_t base = (sq / a.inner_cnt) * a.outer_stride + (sq % a.inner_cnt) * a.inner_stride;
  const int inner = a.heads * 64;
  const int64_t ld = 3 * (int64_t)inner;
  const int q0 = qt * 64;

  // stage the Q tile (rows beyond len are zero)
  for (int e = tid; e < 64 * 16; e += 256) {
    const int r = e >> 4, c4 = e & 15;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (q0 + r < a.len)
      v = *reinterpret_cast<const float4 *>(a.qkv + (base + (int64_t)(q0 + r) * a.row_stride) * ld + h * 64 + c4 * 4);
    float *d = &Qs[r * ATT_QS + c4 * 4];
    d[0] = v.x;
    d[1] = v.y;
    d[2] = v.z;
    d[3] = v.w;
  }

  f32x4 acc_o[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) acc_o[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float m_run = -INFINITY, l_run = 0.f;

  const int nkt = (a.len + 63) / 64;
  // K / V tiles are prefetched into registers one tile ahead; the wave's Q fragments stay in registers
  float4 kreg[4], vreg[4];
  auto fetch = [&](int kt) {
    const int k0 = kt * 64;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int e = tid + i * 256;
      const int r = e >> 4, c4 = e & 15;
      kreg[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      vreg[i] = kreg[i];
      if (k0 + r < a.len) {
        const float *rowp = a.qkv + (base + (int64_t)(k0 + r) * a.row_stride) * ld + h * 64 + c4 * 4;
        kreg[i] = *reinterpret_cast<const float4 *>(rowp + inner);
        vreg[i] = *reinterpret_cast<const float4 *>(rowp + 2 * inner);
      }
    }
  };
  fetch(0);
  __syncthreads();   // Q staged
  float bqr[16];
#pragma unroll
  for (int kk = 0; kk < 16; ++kk) bqr[kk] = Qs[(wave * 16 + li) * ATT_QS + 4 * kk + lk];
  for (int kt = 0; kt < nkt; ++kt) {
    const int k0 = kt * 64;
    __syncthreads();  // previous tile fully consumed
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int e = tid + i * 256;
      const int r = e >> 4, c4 = e & 15;
      float *dk = &Ks[r * ATT_QS + c4 * 4];
      dk[0] = kreg[i].x;
      dk[1] = kreg[i].y;
      dk[2] = kreg[i].z;
      dk[3] = kreg[i].w;
      *reinterpret_cast<float4 *>(&Vs[r * ATT_VS + c4 * 4]) = vreg[i];
    }
    __syncthreads();
    if (kt + 1 < nkt) fetch(kt + 1);

    // S^T[key, query] for this wave's 16 queries
    f32x4 st[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) st[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kk = 0; kk < 16; ++kk) {
      const float b = bqr[kk];
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) {
        const float av = Ks[(mt * 16 + li) * ATT_QS + 4 * kk + lk];
        st[mt] = ASX_MFMA(av, b, st[mt]);
      }
    }
    // scale, mask keys beyond len, online softmax for query (wave*16 + li)
    float mx = -INFINITY;
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int key = k0 + mt * 16 + 4 * lk + r;
        const float s = (key < a.len) ? st[mt][r] * a.scale : -INFINITY;
        st[mt][r] = s;
        mx = fmaxf(mx, s);
      }
    }
    mx = fmaxf(mx, __shfl_xor(mx, 16));
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    const float m_new = fmaxf(m_run, mx);
    const float corr = (m_run == -INFINITY) ? 0.f : (a.exact ? expf(m_run - m_new) : __expf(m_run - m_new));
    float psum = 0.f;
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float p = (st[mt][r] == -INFINITY) ? 0.f : (a.exact ? expf(st[mt][r] - m_new) : __expf(st[mt][r] - m_new));
        st[mt][r] = p;
        psum += p;
      }
    }
    psum += __shfl_xor(psum, 16);
    psum += __shfl_xor(psum, 32);
    l_run = l_run * corr + psum;
    m_run = m_new;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) acc_o[dt] *= corr;
    // O^T[d, query] += V^T[d, key] P^T[key, query]
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float pb = st[mt][r];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
          const float av = Vs[(mt * 16 + 4 * lk + r) * ATT_VS + dt * 16 + li];
          acc_o[dt] = ASX_MFMA(av, pb, acc_o[dt]);
        }
      }
    }
  }

  const int q = q0 + wave * 16 + li;
  if (q < a.len) {
    const int64_t row = base + (int64_t)q * a.row_stride;
    const float g = a.gate[row * a.gate_ld + h];
    const float gs = 1.0f / (1.0f + expf(-g));
    const float inv = gs / l_run;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
      f32x4 o = acc_o[dt];
      o *= inv;
      *reinterpret_cast<f32x4 *>(a.out + row * inner + h * 64 + dt * 16 + 4 * lk) = o;
    }
  }
}
#endif

// ---------------------------------------------------------------------------
// attention2_kernel: the same flash-style algorithm with 16-byte LDS fragment reads.  The MFMA k-steps are re-assigned so
// that a lane's operands are contiguous in LDS:
//   S^T = K Q^T : k-step kk of lane group lk multiplies head dim lk * 16 + kk (attention_kernel: 4 kk + lk), so the sixteen
//                 values a lane needs from a K row (or its Q row) are four ds_read_b128 instead of sixteen ds_read_b32;
//   O^T += V^T P^T : V is stored TRANSPOSED (Vt[d][key]), so the four keys 4 lk .. 4 lk + 3 of k-steps r = 0..3 are one
//                 ds_read_b128; the key index is XOR-swizzled by 4 * (d >> 4) so that the transposing ds_write_b32 of a
//                 32-lane group hit 32 different banks (the 16-byte read stays contiguous: 4 (lk ^ dt) .. + 3).
// 128 fragment reads per 64-key tile and wave become 32; every row stride is 68 floats (16-byte aligned rows, conflict-free
// for the 8-lane groups of a b128 access).  The sum over the head dimension runs in a different order than in
// attention_kernel (same terms, fp32 MFMA accumulation).
// ---------------------------------------------------------------------------
constexpr int ATT2_S = 68;

// QW = 16-query groups per wave: a workgroup owns 64 * QW queries of one (sequence, head).  Every K / V tile that is
// streamed in (32 KB per 64 keys) serves QW times the MFMA work: with QW = 1 the kernel needs ~8 B/clk per CU at the full
// MFMA rate; QW = 2 halves that (and halves the LDS fragment reads per MFMA, the K / V fragments being shared by both query
// groups of a wave) at the price of the third co-resident workgroup (69 KB LDS, 212 registers).  Measured on the ep_317
// layout: QW = 2 357 ms, QW = 1 328 ms (the 4-byte-fragment attention_kernel: 335 ms) -- neither the fragment reads nor the
// K / V stream bind this kernel; the two workgroup barriers per key tile between three co-resident workgroups do.  QW = 1 is
// the default.
// DB: two K / V tile buffers (Q staged through the second one), tile t + 1 written while tile t is consumed -> ONE barrier
// per key tile; 69.6 KB of LDS, two workgroups per CU (A/B against three workgroups with two barriers: ASX_ATTN_DB).
template <int QW, bool DB = false>
__global__ __launch_bounds__(256, (QW == 1 && !DB) ? 3 : 2) void attention2_kernel(AttnArgs a) {
  constexpr int NQ = 64 * QW;                               // queries per workgroup
  constexpr int TILE = 128 * ATT2_S;
  static_assert(!DB || NQ * ATT2_S <= TILE, "Q must fit a K / V tile buffer");
  __shared__ float lds[DB ? 2 * TILE : (NQ + 128) * ATT2_S];
  float *Qs = DB ? lds + TILE : lds;
  float *const kv0 = DB ? lds : lds + NQ * ATT2_S;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, lk = lane >> 4;
  // 1-D grid, XCD-aware: the query tiles of one (sequence, head) run on one XCD and share its L2 copy of that head's K / V
  int lid = xcd_remap(blockIdx.x, gridDim.x);
  const int qt = lid % a.nqt;
  lid /= a.nqt;
  const int h = lid % a.heads;
  const int64_t sq = lid / a.heads;
  const int64_t base = (sq / a.inner_cnt) * a.outer_stride + (sq % a.inner_cnt) * a.inner_stride;
  const int inner = a.heads * 64;
  const int64_t ld = 3 * (int64_t)inner;
  const int q0 = qt * NQ;

  for (int e = tid; e < NQ * 16; e += 256) {
    const int r = e >> 4, c4 = e & 15;
    f32x4 v = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (q0 + r < a.len)
      v = *reinterpret_cast<const f32x4 *>(a.qkv + (base + (int64_t)(q0 + r) * a.row_stride) * ld + h * 64 + c4 * 4);
    *reinterpret_cast<f32x4 *>(&Qs[r * ATT2_S + c4 * 4]) = v;
  }

  f32x4 acc_o[QW][4];
  float m_run[QW], l_run[QW];
#pragma unroll
  for (int g = 0; g < QW; ++g) {
#pragma unroll
    for (int i = 0; i < 4; ++i) acc_o[g][i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    m_run[g] = -INFINITY;
    l_run[g] = 0.f;
  }

  const int nkt = (a.len + 63) / 64;
  f32x4 kreg[4], vreg[4];
  auto fetch = [&](int kt) {
    const int k0 = kt * 64;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int e = tid + i * 256;
      const int r = e >> 4, c4 = e & 15;
      kreg[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
      vreg[i] = kreg[i];
      if (k0 + r < a.len) {
        const float *rowp = a.qkv + (base + (int64_t)(k0 + r) * a.row_stride) * ld + h * 64 + c4 * 4;
        kreg[i] = *reinterpret_cast<const f32x4 *>(rowp + inner);
        vreg[i] = *reinterpret_cast<const f32x4 *>(rowp + 2 * inner);
      }
    }
  };
  auto stage = [&](int buf) {          // registers -> K / V^T tile `buf`
    float *Kb = kv0 + buf * TILE, *Vb = Kb + 64 * ATT2_S;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int e = tid + i * 256;
      const int r = e >> 4, c4 = e & 15;
      *reinterpret_cast<f32x4 *>(&Kb[r * ATT2_S + c4 * 4]) = kreg[i];
      const int rs = r ^ (4 * (c4 >> 2));                 // key swizzle of the transposed V: 4 * (d >> 4), d = 4 c4 + j
#pragma unroll
      for (int j = 0; j < 4; ++j) Vb[(c4 * 4 + j) * ATT2_S + rs] = vreg[i][j];
    }
  };
  fetch(0);
  __syncthreads();   // Q staged
  f32x4 bq[QW][4];   // head dims lk * 16 .. + 15 of query (wave * QW + g) * 16 + li
#pragma unroll
  for (int g = 0; g < QW; ++g)
#pragma unroll
    for (int q = 0; q < 4; ++q)
      bq[g][q] = *reinterpret_cast<const f32x4 *>(&Qs[((wave * QW + g) * 16 + li) * ATT2_S + lk * 16 + 4 * q]);
  if (DB) {
    stage(0);                          // buffer 0 is untouched so far (Q sits in buffer 1)
    if (nkt > 1) fetch(1);
  }
  for (int kt = 0; kt < nkt; ++kt) {
    const int k0 = kt * 64;
    const float *Ks, *Vt;
    if (DB) {
      __syncthreads();   // tile kt visible in buffer kt & 1; buffer (kt + 1) & 1 (tile kt - 1 or the Q stage) is free
      if (kt + 1 < nkt) {
        stage((kt + 1) & 1);
        if (kt + 2 < nkt) fetch(kt + 2);
      }
      Ks = kv0 + (kt & 1) * TILE;
    } else {
      __syncthreads();  // previous tile fully consumed
      stage(0);
      __syncthreads();
      if (kt + 1 < nkt) fetch(kt + 1);
      Ks = kv0;
    }
    Vt = Ks + 64 * ATT2_S;

    f32x4 st[QW][4];
#pragma unroll
    for (int g = 0; g < QW; ++g)
#pragma unroll
      for (int i = 0; i < 4; ++i) st[g][i] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      f32x4 kf[2][4];
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int q = 0; q < 4; ++q)
          kf[m][q] = *reinterpret_cast<const f32x4 *>(&Ks[((2 * p + m) * 16 + li) * ATT2_S + lk * 16 + 4 * q]);
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int g = 0; g < QW; ++g) st[g][2 * p + m] = ASX_MFMA(kf[m][q][j], bq[g][q][j], st[g][2 * p + m]);
    }
#pragma unroll
    for (int g = 0; g < QW; ++g) {
      float mx = -INFINITY;
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int key = k0 + mt * 16 + 4 * lk + r;
          const float sv = (key < a.len) ? st[g][mt][r] * a.scale : -INFINITY;
          st[g][mt][r] = sv;
          mx = fmaxf(mx, sv);
        }
      }
      mx = fmaxf(mx, __shfl_xor(mx, 16));
      mx = fmaxf(mx, __shfl_xor(mx, 32));
      const float m_new = fmaxf(m_run[g], mx);
      const float corr = (m_run[g] == -INFINITY) ? 0.f : (a.exact ? expf(m_run[g] - m_new) : __expf(m_run[g] - m_new));
      float psum = 0.f;
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float p = (st[g][mt][r] == -INFINITY) ? 0.f : (a.exact ? expf(st[g][mt][r] - m_new) : __expf(st[g][mt][r] - m_new));
          st[g][mt][r] = p;
          psum += p;
        }
      }
      psum += __shfl_xor(psum, 16);
      psum += __shfl_xor(psum, 32);
      l_run[g] = l_run[g] * corr + psum;
      m_run[g] = m_new;
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) acc_o[g][dt] *= corr;
    }
    // O^T[d, query] += V^T[d, key] P^T[key, query]
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
      f32x4 vf[4];
#pragma unroll
      for (int dt = 0; dt < 4; ++dt)
        vf[dt] = *reinterpret_cast<const f32x4 *>(&Vt[(dt * 16 + li) * ATT2_S + mt * 16 + 4 * (lk ^ dt)]);
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int dt = 0; dt < 4; ++dt)
#pragma unroll
          for (int g = 0; g < QW; ++g) acc_o[g][dt] = ASX_MFMA(vf[dt][r], st[g][mt][r], acc_o[g][dt]);
    }
  }

#pragma unroll
  for (int g = 0; g < QW; ++g) {
    const int q = q0 + (wave * QW + g) * 16 + li;
    if (q < a.len) {
      const int64_t row = base + (int64_t)q * a.row_stride;
      const float gt = a.gate[row * a.gate_ld + h];
      const float gs = 1.0f / (1.0f + expf(-gt));
      const float inv = gs / l_run[g];
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        f32x4 o = acc_o[g][dt];
        o *= inv;
        *reinterpret_cast<f32x4 *>(a.out + row * inner + h * 64 + dt * 16 + 4 * lk) = o;
      }
    }
  }
}

// ---------------------------------------------------------------------------
// attention6_kernel: the same flash-style algorithm on the bf16 matrix pipe with fp32 results -- both matmuls as six bf16 MFMA
// products on EXACTLY split fp32 operands (kernels_gemm3.h: v = h + m + l, dropped cross terms <= 2^-24 of a product).
//   S^T = K Q^T   : k = head dim in natural order (lane group lk: dims 32 ks + 8 lk .. + 7).  Q is split once per workgroup in
//                   registers (a lane's 16 values are 64 contiguous bytes of its query row); K is split by the staging threads and
//                   written as three bf16 images Kp[part][key][64 dims] (128-byte rows).
//   O^T += V^T P^T: k = key.  The S^T accumulators of a lane hold keys 16 t + 4 lk + r (t = 16-key tile, r = 0..3); the k order
//                   of a 32-key step kp is DEFINED as 8 lk + e <-> key (2 kp + (e >> 2)) * 16 + 4 lk + (e & 3), so that a lane's
//                   P operand is its own eight probabilities of tiles 2 kp, 2 kp + 1 (split in registers, no cross-lane move),
//                   and V is staged TRANSPOSED in that key order: Vp[part][dim][pos(key)], pos = (t >> 1) * 32 + g * 8 +
//                   (t & 1) * 4 + r for key = 16 t + 4 g + r -- a staging thread owns 4 consecutive keys x 4 dims, transposes
//                   them in registers and writes 8 bytes per (dim, part).
// Both images have 128-byte rows; the 16-byte slot index is XOR-ed with (row >> 1) & 7 (K) / ((row >> 1) ^ (row >> 3)) & 7 (V^T), which
// puts the sixteen lanes of every (non-contiguous) lane group of `ds_read_b128` on sixteen distinct slots of the 256-byte bank row;
// the extra term of the V^T image halves the conflicts of its transposing 8-byte staging writes (sixteen rows, one logical slot:
// 4-way -> 2-way, the minimum; first PMC pass: 31 % of the LDS cycles were conflict cycles).
// LDS 48 KB (three workgroups per CU), two barriers per 64-key tile, K / V of the next tile in registers meanwhile.
// ---------------------------------------------------------------------------
// QW = 16-query groups per wave (64 QW queries per workgroup): the K / V split of a tile (176 VALU per thread, the kernel is VALU-bound)
// serves QW times the MFMA work.
// H: the fp16 x 3 arithmetic of kernels_gemm3.h (two fp16 parts per operand, three MFMAs per product).  Block exponents: one per QUERY for Q
// (its 64 dims live in one lane quadruple: two shuffles), one per 64-key TILE for K (the S^T accumulators are fresh every tile: the factor
// 2^-(e_k + e_q) rides on the logit scale), a running one per tile for V (it only drops, with two bits of headroom; the drop rides on the
// online-softmax correction the O accumulators are multiplied by anyway), none for P (probabilities are in [0, 1]: absolute error 2^-25).
// The tile maxima cross the workgroup through eight LDS floats written in front of the barrier that already separates two tiles.
template <int QW, bool H = false>
__global__ __launch_bounds__(256, (QW == 1 ? 3 : 2)) void attention6_kernel(AttnArgs a) {
  constexpr int NP = H ? 2 : 3;                        // parts per operand
  constexpr int PARTB = 64 * 128;                      // bytes of one part image (64 rows x 64 16-bit values)
  __shared__ __attribute__((aligned(16))) char lds6[2 * NP * PARTB];
  __shared__ __attribute__((aligned(16))) float tile_mx[8];   // H: per-wave largest |K|, |V| of the tile about to be staged
  char *Kp = lds6, *Vp = lds6 + NP * PARTB;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, lk = lane >> 4;
  int lid = xcd_remap(blockIdx.x, gridDim.x);
  const int qt = lid % a.nqt;
  lid /= a.nqt;
  const int h = lid % a.heads;
  const int64_t sq = lid / a.heads;
  const int64_t base = (sq / a.inner_cnt) * a.outer_stride + (sq % a.inner_cnt) * a.inner_stride;
  const int inner = a.heads * 64;
  const int64_t ld = 3 * (int64_t)inner;
  const int q0 = qt * 64 * QW;

  // ---- Q operand of this lane: query q0 + 16 wave + li, dims 32 ks + 8 lk .. + 7, split into three bf16 fragments per k step
  u32x4 qf[QW][2][NP];
  int eq[QW];                                          // H: exponent of this lane's query
#pragma unroll
  for (int g = 0; g < QW; ++g) {
    const int q = q0 + (wave * QW + g) * 16 + li;
    const bool ok = q < a.len;
    const float *qr = a.qkv + (base + (int64_t)(ok ? q : 0) * a.row_stride) * ld + h * 64 + 8 * lk;
    f32x4 qv[2][2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      qv[ks][0] = (f32x4){0.f, 0.f, 0.f, 0.f};
      qv[ks][1] = qv[ks][0];
      if (ok) {
        qv[ks][0] = *reinterpret_cast<const f32x4 *>(qr + 32 * ks);
        qv[ks][1] = *reinterpret_cast<const f32x4 *>(qr + 32 * ks + 4);
      }
    }
    eq[g] = 0;
    if constexpr (H) {
      float m = fmaxf(absmax_oct(qv[0][0], qv[0][1]), absmax_oct(qv[1][0], qv[1][1]));
      m = fmaxf(m, __shfl_xor(m, 16));
      m = fmaxf(m, __shfl_xor(m, 32));
      eq[g] = f16_scale_exp(m);
    }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      if constexpr (H) split2h_oct(qv[ks][0], qv[ks][1], eq[g], qf[g][ks][0], qf[g][ks][1]);
      else split3_oct(qv[ks][0], qv[ks][1], qf[g][ks][0], qf[g][ks][1], qf[g][ks][NP - 1]);
    }
  }

  f32x4 acc_o[QW][4];
  float m_run[QW], l_run[QW];
#pragma unroll
  for (int g = 0; g < QW; ++g) {
#pragma unroll
    for (int i = 0; i < 4; ++i) acc_o[g][i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    m_run[g] = -INFINITY;
    l_run[g] = 0.f;
  }

  // ---- staging: thread (kb = tid >> 4, c4 = tid & 15) owns keys 4 kb .. + 3 x dims 4 c4 .. + 3 of a tile
  const int kb = tid >> 4, c4 = tid & 15;
  const int nkt = (a.len + 63) / 64;
  f32x4 kreg[4], vreg[4];
  auto fetch = [&](int kt) {
    const int k0 = kt * 64 + 4 * kb;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      kreg[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
      vreg[j] = kreg[j];
      if (k0 + j < a.len) {
        const float *rowp = a.qkv + (base + (int64_t)(k0 + j) * a.row_stride) * ld + h * 64 + c4 * 4;
        kreg[j] = *reinterpret_cast<const f32x4 *>(rowp + inner);
        vreg[j] = *reinterpret_cast<const f32x4 *>(rowp + 2 * inner);
      }
    }
  };
  // K image: row = key, 8-byte unit c4 of the row (dims 4 c4 .. + 3): slot c4 >> 1, half c4 & 1
  // V image: row = dim, keys 4 kb .. + 3 at pos (t >> 1) * 32 + g * 8 + (t & 1) * 4 (t = kb >> 2, g = kb & 3): slot (t >> 1) * 4 + g, half t & 1
  const int vslot = ((kb >> 3) << 2) | (kb & 3), vhalf = (kb >> 2) & 1;
  int ek = 0, ev_run = 200;                            // H: exponent of the K tile in LDS; running exponent of V (and of the O accumulators)
  float fdev = 1.0f;                                   // H: 2^(drop of ev_run at this tile), applied with the softmax correction
  auto publish_tile_max = [&]() {                      // H: largest |K|, |V| of the fetched tile, per wave
    float mk = 0.f, mv = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      mk = fmaxf(mk, fmaxf(fmaxf(fabsf(kreg[j].x), fabsf(kreg[j].y)), fmaxf(fabsf(kreg[j].z), fabsf(kreg[j].w))));
      mv = fmaxf(mv, fmaxf(fmaxf(fabsf(vreg[j].x), fabsf(vreg[j].y)), fmaxf(fabsf(vreg[j].z), fabsf(vreg[j].w))));
    }
    mk = wave_max64(mk);
    mv = wave_max64(mv);
    if (lane == 0) {
      tile_mx[wave] = mk;
      tile_mx[4 + wave] = mv;
    }
  };
  auto stage = [&]() {
    if constexpr (H) {
      const f32x4 a4 = *reinterpret_cast<const f32x4 *>(tile_mx), b4 = *reinterpret_cast<const f32x4 *>(tile_mx + 4);
      ek = __builtin_amdgcn_readfirstlane(f16_scale_exp(fmaxf(fmaxf(a4.x, a4.y), fmaxf(a4.z, a4.w))));
      const int need = __builtin_amdgcn_readfirstlane(f16_scale_exp(fmaxf(fmaxf(b4.x, b4.y), fmaxf(b4.z, b4.w))));
      const int ev_new = need < ev_run ? need - 2 : ev_run;
      fdev = __builtin_ldexpf(1.0f, ev_new - ev_run);
      ev_run = ev_new;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int row = 4 * kb + j;
      unsigned hh[2], mm[2], ll[2];
      if constexpr (H) {
        split2h_pair(kreg[j].x, kreg[j].y, ek, hh[0], ll[0]);
        split2h_pair(kreg[j].z, kreg[j].w, ek, hh[1], ll[1]);
      } else {
        split3_pair(kreg[j].x, kreg[j].y, hh[0], mm[0], ll[0]);
        split3_pair(kreg[j].z, kreg[j].w, hh[1], mm[1], ll[1]);
      }
      char *d = Kp + row * 128 + ((((c4 >> 1) ^ ((row >> 1) & 7)) << 4) | ((c4 & 1) << 3));
      *reinterpret_cast<uint2 *>(d) = make_uint2(hh[0], hh[1]);
      if constexpr (!H) *reinterpret_cast<uint2 *>(d + PARTB) = make_uint2(mm[0], mm[1]);
      *reinterpret_cast<uint2 *>(d + (NP - 1) * PARTB) = make_uint2(ll[0], ll[1]);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = 4 * c4 + i;
      unsigned hh[2], mm[2], ll[2];
      if constexpr (H) {
        split2h_pair(vreg[0][i], vreg[1][i], ev_run, hh[0], ll[0]);
        split2h_pair(vreg[2][i], vreg[3][i], ev_run, hh[1], ll[1]);
      } else {
        split3_pair(vreg[0][i], vreg[1][i], hh[0], mm[0], ll[0]);
        split3_pair(vreg[2][i], vreg[3][i], hh[1], mm[1], ll[1]);
      }
      char *d = Vp + row * 128 + (((vslot ^ (((row >> 1) ^ (row >> 3)) & 7)) << 4) | (vhalf << 3));   // V^T swizzle: see the fragment read
      *reinterpret_cast<uint2 *>(d) = make_uint2(hh[0], hh[1]);
      if constexpr (!H) *reinterpret_cast<uint2 *>(d + PARTB) = make_uint2(mm[0], mm[1]);
      *reinterpret_cast<uint2 *>(d + (NP - 1) * PARTB) = make_uint2(ll[0], ll[1]);
    }
  };
  // fragment reads: row 16 t + li, logical slot s -> physical s ^ ((li >> 1) & 7)
  const int frow = li * 128, fsw = (li >> 1) & 7;

  fetch(0);
  for (int kt = 0; kt < nkt; ++kt) {
    const int k0 = kt * 64;
    if constexpr (H) publish_tile_max();
    __syncthreads();   // previous tile fully consumed
    stage();
    __syncthreads();
    if (kt + 1 < nkt) fetch(kt + 1);

    // ---- S^T[key, query] for this wave's 16 QW queries (a K fragment serves every query group) ----
    f32x4 st[QW][4];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
#pragma unroll
      for (int g = 0; g < QW; ++g) st[g][mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const char *kp = Kp + mt * 2048 + frow + (((ks * 4 + lk) ^ fsw) << 4);
        if constexpr (H) {
          const f16x8 kh = *reinterpret_cast<const f16x8 *>(kp), kl = *reinterpret_cast<const f16x8 *>(kp + PARTB);
#pragma unroll
          for (int g = 0; g < QW; ++g) {
            const f16x8 qh = __builtin_bit_cast(f16x8, qf[g][ks][0]), ql = __builtin_bit_cast(f16x8, qf[g][ks][1]);
            st[g][mt] = ASX_MFMA_F16(kl, qh, st[g][mt]);
            st[g][mt] = ASX_MFMA_F16(kh, ql, st[g][mt]);
            st[g][mt] = ASX_MFMA_F16(kh, qh, st[g][mt]);
          }
          continue;
        }
        const bf16x8 kh = *reinterpret_cast<const bf16x8 *>(kp);
        const bf16x8 km = *reinterpret_cast<const bf16x8 *>(kp + PARTB);
        const bf16x8 kl = *reinterpret_cast<const bf16x8 *>(kp + (NP - 1) * PARTB);
#pragma unroll
        for (int g = 0; g < QW; ++g) {
          const bf16x8 qh = __builtin_bit_cast(bf16x8, qf[g][ks][0]), qm = __builtin_bit_cast(bf16x8, qf[g][ks][1]),
                       ql = __builtin_bit_cast(bf16x8, qf[g][ks][NP - 1]);
          st[g][mt] = ASX_MFMA_BF16(kl, qh, st[g][mt]);
          st[g][mt] = ASX_MFMA_BF16(kh, ql, st[g][mt]);
          st[g][mt] = ASX_MFMA_BF16(km, qm, st[g][mt]);
          st[g][mt] = ASX_MFMA_BF16(km, qh, st[g][mt]);
          st[g][mt] = ASX_MFMA_BF16(kh, qm, st[g][mt]);
          st[g][mt] = ASX_MFMA_BF16(kh, qh, st[g][mt]);
        }
      }
    }
    // ---- scale, mask keys beyond len, online softmax per query group ----
#pragma unroll
    for (int g = 0; g < QW; ++g) {
      float mx = -INFINITY;
      const float sfac = H ? __builtin_ldexpf(a.scale, -(ek + eq[g])) : a.scale;   // H: back to the operands' own scale, exact
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int key = k0 + mt * 16 + 4 * lk + r;
          const float sv = (key < a.len) ? st[g][mt][r] * sfac : -INFINITY;
          st[g][mt][r] = sv;
          mx = fmaxf(mx, sv);
        }
      }
      mx = fmaxf(mx, __shfl_xor(mx, 16));
      mx = fmaxf(mx, __shfl_xor(mx, 32));
      const float m_new = fmaxf(m_run[g], mx);
      const float corr = (m_run[g] == -INFINITY) ? 0.f : (a.exact ? expf(m_run[g] - m_new) : __expf(m_run[g] - m_new));
      float psum = 0.f;
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float p = (st[g][mt][r] == -INFINITY) ? 0.f : (a.exact ? expf(st[g][mt][r] - m_new) : __expf(st[g][mt][r] - m_new));
          st[g][mt][r] = p;
          psum += p;
        }
      }
      psum += __shfl_xor(psum, 16);
      psum += __shfl_xor(psum, 32);
      l_run[g] = l_run[g] * corr + psum;
      m_run[g] = m_new;
      const float corr_o = H ? corr * fdev : corr;     // H: the O accumulators follow V's running exponent
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) acc_o[g][dt] *= corr_o;
    }
    // ---- O^T[d, query] += V^T[d, key] P^T[key, query], 32 keys per k step (a V fragment serves every query group) ----
#pragma unroll
    for (int kp = 0; kp < 2; ++kp) {
      if constexpr (H) {
        f16x8 p_h[QW], p_l[QW];
#pragma unroll
        for (int g = 0; g < QW; ++g) {
          u32x4 ph, pl;
          split2h_oct(st[g][2 * kp], st[g][2 * kp + 1], 14, ph, pl);  // probabilities (<= 1 behind the running maximum) x 2^14: the small ones stay clear of fp16's subnormals; undone with V's exponent
          p_h[g] = __builtin_bit_cast(f16x8, ph);
          p_l[g] = __builtin_bit_cast(f16x8, pl);
        }
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
          const char *vp = Vp + dt * 2048 + frow + (((kp * 4 + lk) ^ (fsw ^ ((2 * dt + (li >> 3)) & 7))) << 4);
          const f16x8 vh = *reinterpret_cast<const f16x8 *>(vp), vl = *reinterpret_cast<const f16x8 *>(vp + PARTB);
#pragma unroll
          for (int g = 0; g < QW; ++g) {
            acc_o[g][dt] = ASX_MFMA_F16(vl, p_h[g], acc_o[g][dt]);
            acc_o[g][dt] = ASX_MFMA_F16(vh, p_l[g], acc_o[g][dt]);
            acc_o[g][dt] = ASX_MFMA_F16(vh, p_h[g], acc_o[g][dt]);
          }
        }
        continue;
      }
      bf16x8 p_h[QW], p_m[QW], p_l[QW];
#pragma unroll
      for (int g = 0; g < QW; ++g) {
        u32x4 ph, pm, pl;
        split3_oct(st[g][2 * kp], st[g][2 * kp + 1], ph, pm, pl);
        p_h[g] = __builtin_bit_cast(bf16x8, ph);
        p_m[g] = __builtin_bit_cast(bf16x8, pm);
        p_l[g] = __builtin_bit_cast(bf16x8, pl);
      }
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        const char *vp = Vp + dt * 2048 + frow + (((kp * 4 + lk) ^ (fsw ^ ((2 * dt + (li >> 3)) & 7))) << 4);   // row = 16 dt + li: ((row >> 1) ^ (row >> 3)) & 7
        const bf16x8 vh = *reinterpret_cast<const bf16x8 *>(vp);
        const bf16x8 vm = *reinterpret_cast<const bf16x8 *>(vp + PARTB);
        const bf16x8 vl = *reinterpret_cast<const bf16x8 *>(vp + (NP - 1) * PARTB);
#pragma unroll
        for (int g = 0; g < QW; ++g) {
          acc_o[g][dt] = ASX_MFMA_BF16(vl, p_h[g], acc_o[g][dt]);
          acc_o[g][dt] = ASX_MFMA_BF16(vh, p_l[g], acc_o[g][dt]);
          acc_o[g][dt] = ASX_MFMA_BF16(vm, p_m[g], acc_o[g][dt]);
          acc_o[g][dt] = ASX_MFMA_BF16(vm, p_h[g], acc_o[g][dt]);
          acc_o[g][dt] = ASX_MFMA_BF16(vh, p_m[g], acc_o[g][dt]);
          acc_o[g][dt] = ASX_MFMA_BF16(vh, p_h[g], acc_o[g][dt]);
        }
      }
    }
  }

#pragma unroll
  for (int g = 0; g < QW; ++g) {
    const int q = q0 + (wave * QW + g) * 16 + li;
    if (q < a.len) {
      const int64_t row = base + (int64_t)q * a.row_stride;
      const float gt = a.gate[row * a.gate_ld + h];
      const float gs = 1.0f / (1.0f + expf(-gt));
      const float inv = gs / l_run[g];
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        f32x4 o = acc_o[g][dt];
        o *= inv;
        if constexpr (H) {                             // V's exponent (exact)
          o.x = __builtin_ldexpf(o.x, -ev_run - 14);
          o.y = __builtin_ldexpf(o.y, -ev_run - 14);
          o.z = __builtin_ldexpf(o.z, -ev_run - 14);
          o.w = __builtin_ldexpf(o.w, -ev_run - 14);
        }
        *reinterpret_cast<f32x4 *>(a.out + row * inner + h * 64 + dt * 16 + 4 * lk) = o;
      }
    }
  }
}

// ---------------------------------------------------------------------------
// GLU of the mask MLP output (bs_roformer.py:216) scattered into the mask tensor [B, S, T, W]:
//   mask[b, st, t, off + j] = a[m, j] * sigmoid(a[m, din + j]),   m = b*T + t
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void glu_kernel(const float *__restrict__ a, int din, int64_t M, int T, int S, int st,
                                                  float *__restrict__ y, int64_t W) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= M * din) return;
  const int64_t m = idx / din;
  const int j = (int)(idx - m * din);
  const int64_t b = m / T, t = m - b * T;
  const float x = a[m * 2 * din + j], g = a[m * 2 * din + din + j];
  y[((b * S + st) * T + t) * W + j] = x * (1.0f / (1.0f + expf(-g)));
}

// ---------------------------------------------------------------------------
// Mel-Band Roformer mask merge (mel_band_roformer.py:404-416): band masks maskb [R, MW] (R = B*S*T rows, band j at
// moff[j], (f s c) order inside) -> per-bin masks mask [R, W]: sum over the bands jlo[f] .. jhi[f] that cover bin f,
// divided by their number (scatter_add_ then / num_bands_per_freq.clamp(1e-8)).
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void mel_mask_merge_kernel(const float *__restrict__ maskb, int MW, int W,
                                                             const int *__restrict__ bstart, const int *__restrict__ moff,
                                                             const int *__restrict__ jlo, const int *__restrict__ jhi,
                                                             float *__restrict__ mask, int64_t total) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int64_t row = idx / W;
  const int e = (int)(idx - row * W);
  const int f = e >> 2, sc = e & 3;
  const float *src = maskb + row * MW;
  float acc = 0.f;
  const int j0 = jlo[f], j1 = jhi[f];
  for (int j = j0; j <= j1; ++j) acc += src[moff[j] + (f - bstart[j]) * 4 + sc];
  mask[idx] = acc / fmaxf((float)(j1 - j0 + 1), 1e-8f);
}

// ---------------------------------------------------------------------------
// Roformer chunk fold (mdxc_separator.py:320-343): result += x * w, counter += w,
// out = result / clamp(counter, 1e-10); chunk k covers [starts[k], starts[k] + C).
// chunk_out [n_chunks, S, 2, C];  out [n_out, 2, N] where stem row o reads chunk stem (o % S)
// (the reference broadcasts a single-stem output over len(instruments) rows).
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void roformer_finalize_kernel(const float *__restrict__ chunk_out,
                                                                const int64_t *__restrict__ starts, int n_chunks, int S,
                                                                int64_t C, const float *__restrict__ window, int64_t N,
                                                                float *__restrict__ out) {
  const int oc = blockIdx.y;  // o*2 + ch
  const int o = oc >> 1, ch = oc & 1;
  const int s = o % S;
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  float acc = 0.f, cnt = 0.f;
  for (int k = 0; k < n_chunks; ++k) {
    const int64_t j = i - starts[k];
    if (j < 0 || j >= C) continue;
    const float w = window[j];
    acc += chunk_out[(((int64_t)k * S + s) * 2 + ch) * C + j] * w;
    cnt += w;
  }
  out[(int64_t)oc * N + i] = acc / fmaxf(cnt, 1e-10f);
}

}  // namespace asx
