// Winograd F(2x2, 3x3), WEIGHT-STATIONARY form (round 4) for the 3x3 / pad-1 convolutions of the TFC blocks
// (uvr_lib_v5/modules.py:11-17, 22-54) with few input channels (Cin <= 96: the two outer levels of the HQ_3 net, 68 % of its
// 3x3 multiply-adds).
//
// What conv_wino3_kernel (kernels_wino.h) pays for, per 192 MFMAs of a workgroup: 13 LDS-DMA pieces of transformed weights
// U = G g G^T, 48 ds_read_b128 of them, 8 pieces of raw planes -- the weights are 16/9 larger than the direct kernel's and the
// stage does 2.25x fewer MFMAs, so the operand stream, not the matrix pipe, sets the pace (MFMA busy 63 %).  Here the weights
// never move after the prologue:
//
//   * the sixteen transform positions (a, b) are SPLIT OVER THE EIGHT WAVES of a workgroup: wave w owns row a = w >> 1 and the
//     column pair b in {2 (w & 1), 2 (w & 1) + 1}.  Its share of U for ALL input channels and 48 output channels --
//     2 positions x Cin/4 k-steps x 3 cout fragments = 6 Cin/4 floats per lane, 144 registers at Cin = 96 -- is loaded once
//     and stays in VGPRs as the MFMA B operand; the accumulators are only 2 x 3 fragments = 24 registers;
//   * the workgroup walks DOWN a column strip of 16 tiles (32 output pixels wide), one tile row (2 output rows) per step.  A
//     step needs input rows 2s - 1 .. 2s + 2 of every channel; they live in a six-row LDS ring ([row][channel][40 floats]),
//     and only the two NEW rows of the next step arrive per step -- Cin x 320 B, 15 / 30 LDS-DMA pieces per 576 / 1152 MFMAs,
//     a quarter of the resident kernel's rate;
//   * a wave's input transform is tiny: row a of B^T d is ONE signed sum of two raw rows (d0 - d2, d1 + d2, d2 - d1, d1 - d3),
//     its two columns two more sums: 2 x 2 ds_read_b64 and ~9 VALU per 6 MFMAs, no weight reads at all;
//   * the output transform needs all sixteen positions of a tile, i.e. all eight waves: each wave folds its two columns into
//     Z_w[q] = sum_b m[a][b] A[b][q] (lane-local), the eight partial Z meet in LDS (48 KB), and wave (r', p) finishes output
//     row p of the tiles r' of every lane quad: Y[p][q] = sum_a A^T[p][a] (Z_{a,0}[q] + Z_{a,1}[q]), + bias, activation, store.
//
// LDS images are laid out for the reads: a channel row is 40 floats (160 B = 32 banks mod 64), and the four channels of an
// MFMA k-step are NOT consecutive -- lanes lk = 0 / 1 of a 32-lane ds_read_b64 group read channels four apart (c, c + 4), lanes
// lk = 2 / 3 channels c + 1, c + 5 -- so the two halves of a group sit on opposite bank halves; the whole ring is shifted by one
// float so that a tile's four patch columns start 8-byte aligned (LDS-DMA accepts the 4-byte aligned destination).
// Results do not depend on batch size or sharding (the tiling is a function of the plane only).
#pragma once
#include <type_traits>
#include "kernels_wino.h"

namespace asx {

template <int KS_>
struct WinoSCfg {
  static constexpr int KS = KS_, CIN = 4 * KS, IWA = 40, TW = 32;
  static constexpr int ROW = CIN * IWA;          // floats per ring row: every channel's 40-float segment of one input row
  static constexpr int RING = 6 * ROW;
  static constexpr int SLOTS2 = 2 * CIN * 10;    // float4 slots of a row PAIR (the unit that arrives per step)
  static constexpr int PIECES = SLOTS2 / 64;     // 1-KiB LDS-DMA pieces per row pair
  static constexpr int PPW = (PIECES + 7) / 8;   // pieces per wave
  static constexpr int UREG = 6 * KS;            // stationary weight floats per lane: [k-step][b'][cout fragment]
  static constexpr int UV4 = UREG / 4;
  static constexpr int ZOFF = RING + 8;          // the ring starts at float 1; Z stays 32-byte aligned
  static constexpr int ZF = 8 * 3 * 4 * 64 * 2;  // [wave][cout fragment][r][lane][q]
  static constexpr int LDS_BYTES = (ZOFF + ZF) * 4;
  static_assert(SLOTS2 % 64 == 0 && UREG % 4 == 0, "row pairs are whole DMA pieces, weights whole float4");
};

// host side: float offset of U[a][bb][c][co % 48] inside one cout group's image [wave 8][UV4][lane 64][4]
template <int KS>
static inline size_t winos_u_index(int a, int bb, int c, int col) {
  const int wave = 2 * a + (bb >> 1), bp = bb & 1, n = col >> 4, li = col & 15;
  const int j = c >> 3, x = c & 7;
  const int kodd = (x >> 1) & 1, lk = ((x >> 2) & 1) + 2 * (x & 1);
  const int k = 2 * j + kodd;
  const int e = (k * 2 + bp) * 3 + n;
  return (((size_t)wave * WinoSCfg<KS>::UV4 + (e >> 2)) * 64 + (li + 16 * lk)) * 4 + (e & 3);
}

// ConvArgs: wp = stationary image [CG][8][UV4][64][4]; tilesF = strips (Fo / 32); tilesT = row blocks; NCI = tile rows per block.
// The activation is ReLU or none (a.act), no residual operand (the caller keeps conv_wino3_kernel for anything else).
// RAGGED: To odd or Cout not a multiple of 48 -- stores are predicated (a branch per store); the common case stores unconditionally
// so that a step is ONE basic block the scheduler can interleave freely.
// ABL (ASX_WINOS_ABL, measurement-only builds whose results are garbage): 1 = no LDS-DMA after the prologue, 2 = no global stores,
// 4 = no Z exchange / output transform at all, 8 = no patch reads in the k-loop
//
// Schedule of a step (every wave, all eight in lock step at the two barriers):
//   DMA issue (next step's two rows)  |  k-loop: patch reads run PF k-steps ahead of their MFMAs (sched_barrier pins the order),
//   and the OUTPUT TRANSFORM OF THE PREVIOUS STEP is spread through it -- the Z reads of cout fragment n at one k-step, the sums /
//   bias / activation / store two k-steps later -- so that the exchange costs issue slots beside MFMAs, not a serial phase
//   barrier B (Z of the previous step fully consumed)  |  fold this step's accumulators into Z, write  |  barrier A (Z and the new
//   rows visible).  Only the short Z write sits between the barriers.
template <int KS, int ABL = 0, bool RAGGED = false>
__global__ __launch_bounds__(512, 2) void conv_winos_kernel(ConvArgs a) {
  using CFG = WinoSCfg<KS>;
  extern __shared__ float lds_f[];
  float *ring = lds_f + 1;
  float *zb = lds_f + CFG::ZOFF;
  constexpr int ROW = CFG::ROW, PIECES = CFG::PIECES, PPW = CFG::PPW, UV4 = CFG::UV4, CIN = CFG::CIN;
  constexpr int PF = 2;                          // patch prefetch distance in k-steps

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, lk = lane >> 4;
  const int pa = wave >> 1, bh = wave & 1;

  int lid = xcd_remap(blockIdx.x, gridDim.x);
  const int cg = lid % a.CG;
  lid /= a.CG;
  const int sf = lid % a.tilesF;
  lid /= a.tilesF;
  const int rb = lid % a.tilesT;
  const int b = lid / a.tilesT;
  const int fo0 = sf * CFG::TW;
  const int NS = (a.To + 1) >> 1;
  const int s_begin = rb * a.NCI;
  const int s_end = min(NS, s_begin + a.NCI);
  if (s_begin >= s_end) return;

  const float *xb = a.x + (int64_t)b * a.x_bstride;
  const int64_t plane_sz = (int64_t)a.T * a.F;

  // ---- stationary weights: this wave's two positions, every k-step, three cout fragments ----
  f32x4 uq[UV4];
  {
    const f32x4 *ug = reinterpret_cast<const f32x4 *>(a.wp) + ((int64_t)(cg * 8 + wave) * UV4) * 64 + lane;
#pragma unroll
    for (int i = 0; i < UV4; ++i) uq[i] = ug[i * 64];
  }

  // ---- LDS-DMA of a row pair (ring rows R0, R0 + 1 = input rows R0 - 1, R0): this wave's pieces ----
  uint32_t goff[PPW];
  bool cok[PPW];
  int prow[PPW];
#pragma unroll
  for (int i = 0; i < PPW; ++i) {
    const int p = wave + 8 * i;
    const int f = p * 64 + lane;
    const int row = f / (CIN * 10), rem = f - row * (CIN * 10);
    const int ch = rem / 10, c4 = rem - ch * 10;
    const int col = fo0 - 4 + 4 * c4;
    cok[i] = (p < PIECES) && col >= 0 && col < a.F && ch < a.Cin;
    goff[i] = (uint32_t)(ch * plane_sz + col);
    prow[i] = row;
  }
  auto issue_pair = [&](int R0) {
    float *dst = ring + (R0 % 6) * ROW;
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
      const int p = wave + 8 * i;
      if (p < PIECES) {
        const int t = R0 - 1 + prow[i];
        const bool ok = cok[i] && t >= 0 && t < a.T;
        const float *src = ok ? xb + goff[i] + (int64_t)t * a.F : a.zeros;
        ASX_GLDS16(src, dst + p * 256);
      }
    }
  };

  // row a of B^T d = dA + sB * dB with (A, B, sB) = (0, 2, -), (1, 2, +), (2, 1, -), (1, 3, -)
  const int rowA = pa == 0 ? 0 : (pa == 2 ? 2 : 1);
  const int rowB = pa == 2 ? 1 : (pa == 3 ? 3 : 2);
  const float sB = pa == 1 ? 1.0f : -1.0f;
  // lane part of the patch address: channel (4 (lk & 1) + (lk >> 1)) of the k-step's group of eight, tile li
  // (element i of a channel segment is input column fo0 - 4 + i; tile li needs columns fo0 + 2 li - 1 .. + 2, i.e. i = 2 li + 3:
  // LDS float 2 li + 4 counting the ring's one-float shift, an 8-byte aligned address -- a misaligned ds_read_b64 costs 3x)
  const int lane_off = (4 * (lk & 1) + (lk >> 1)) * CFG::IWA + 2 * li + 3;

  float *yb = a.y + (int64_t)b * a.y_bstride;
  // output transform of wave (rq, pp): accumulator component rq (tile 4 lk + rq), output row pp of the tile row:
  // Y[pp][q] = Zs[pp][q] + sg (Zs[pp + 1][q] + Zs[pp + 2][q]), Zs[a] = Z_{a,0} + Z_{a,1}, sg = +1 / -1 (A^T = [[1,1,1,0],[0,1,-1,-1]])
  const int rq = wave & 3, pp = wave >> 2;
  const float sg = pp ? -1.0f : 1.0f;
  const float lo = a.act == ACT_RELU ? 0.0f : -__builtin_inff();
  const f32x2 *zr = reinterpret_cast<const f32x2 *>(zb) + ((2 * pp) * 3 * 4 + rq) * 64 + lane;
  f32x2 *zw = reinterpret_cast<f32x2 *>(zb) + (wave * 3 * 4) * 64 + lane;
  const int fcol = fo0 + 2 * (4 * lk + rq);
  float bv[3];
#pragma unroll
  for (int n = 0; n < 3; ++n) bv[n] = a.bias[cg * 48 + n * 16 + li];
  float *yrow[3];                 // &y[cout fragment n][row 0][fcol]
#pragma unroll
  for (int n = 0; n < 3; ++n) yrow[n] = yb + (int64_t)(cg * 48 + n * 16 + li) * a.To * a.Fo + fcol;

  f32x2 zin[6];                   // Z reads of one cout fragment in flight
  auto fin_read = [&](int n) {
#pragma unroll
    for (int j = 0; j < 6; ++j) zin[j] = zr[(j * 3 + n) * 4 * 64];       // waves 2 pp .. 2 pp + 5 = (a, bh) pairs of rows pp .. pp + 2
  };
  auto fin_store = [&](int n, int sp) {                                   // sp: the step the Z buffer belongs to
    f32x2 y = (zin[0] + zin[1]) + sg * ((zin[2] + zin[3]) + (zin[4] + zin[5]));
    y.x = fmaxf(y.x + bv[n], lo);
    y.y = fmaxf(y.y + bv[n], lo);
    const int t = 2 * sp + pp;
    if constexpr (ABL & 2) {
      if (y.x + y.y != 1.2345e-30f) return;
    }
    if constexpr (RAGGED) {
      if (t < a.To && cg * 48 + n * 16 + li < a.Cout) *reinterpret_cast<f32x2 *>(yrow[n] + (int64_t)t * a.Fo) = y;
    } else {
      *reinterpret_cast<f32x2 *>(yrow[n] + (int64_t)t * a.Fo) = y;
    }
  };
  // k-steps at which the previous step's cout fragment n is read / finished (spread over the loop, two k-steps apart)
  constexpr int FR0 = 1, FR1 = KS / 3 + 1, FR2 = 2 * (KS / 3) + 1;

  issue_pair(2 * s_begin);
  issue_pair(2 * s_begin + 2);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  auto step = [&](int s, auto fin_prev) {
    constexpr bool FIN = decltype(fin_prev)::value && !(ABL & 4);
    // rows 2 s + 4, 2 s + 5 (for step s + 1) replace rows 2 s - 2, 2 s - 1, which every wave left behind the last barrier
    if constexpr (!(ABL & 1)) {
      if (s + 1 < s_end) issue_pair(2 * s + 4);
    }
    f32x4 acc[2][3];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int n = 0; n < 3; ++n) acc[j][n] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const float *pA = ring + ((2 * s + rowA) % 6) * ROW + lane_off;
    const float *pB = ring + ((2 * s + rowB) % 6) * ROW + lane_off;
    f32x2 d[PF + 1][4];
    auto load = [&](int k) {
      const int koff = (8 * (k >> 1) + 2 * (k & 1)) * CFG::IWA;
      f32x2 *q = d[k % (PF + 1)];
      if constexpr (ABL & 8) {
        q[0] = (f32x2){(float)s, (float)k}, q[1] = (f32x2){(float)lane, 1.f}, q[2] = (f32x2){2.f, (float)(s + k)}, q[3] = (f32x2){(float)(lane - s), 3.f};
      } else {
        q[0] = *reinterpret_cast<const f32x2 *>(pA + koff), q[1] = *reinterpret_cast<const f32x2 *>(pA + koff + 2);
        q[2] = *reinterpret_cast<const f32x2 *>(pB + koff), q[3] = *reinterpret_cast<const f32x2 *>(pB + koff + 2);
      }
    };
#pragma unroll
    for (int k = 0; k < PF; ++k) load(k);
#pragma unroll
    for (int k = 0; k < KS; ++k) {
      if (k + PF < KS) load(k + PF);
      if constexpr (FIN) {
        if (k == FR0) fin_read(0);
        if (k == FR1) fin_read(1);
        if (k == FR2) fin_read(2);
      }
      __builtin_amdgcn_sched_barrier(0);
      const f32x2 *q = d[k % (PF + 1)];
      const float r0 = __fmaf_rn(sB, q[2].x, q[0].x), r1 = __fmaf_rn(sB, q[2].y, q[0].y);
      const float r2 = __fmaf_rn(sB, q[3].x, q[1].x), r3 = __fmaf_rn(sB, q[3].y, q[1].y);
      // columns: b = 0: r0 - r2, 1: r1 + r2, 2: r2 - r1, 3: r1 - r3
      const float v0 = bh ? r2 - r1 : r0 - r2;
      const float v1 = bh ? r1 - r3 : r1 + r2;
#pragma unroll
      for (int n = 0; n < 3; ++n) {
        const int e0 = (k * 2 + 0) * 3 + n, e1 = (k * 2 + 1) * 3 + n;
        acc[0][n] = ASX_MFMA(v0, uq[e0 >> 2][e0 & 3], acc[0][n]);
        acc[1][n] = ASX_MFMA(v1, uq[e1 >> 2][e1 & 3], acc[1][n]);
      }
      if constexpr (FIN) {
        if (k == FR0 + 2) fin_store(0, s - 1);
        if (k == FR1 + 2) fin_store(1, s - 1);
        if (k == FR2 + 2) fin_store(2, s - 1);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (ABL & 4) {
      float chk = 0.f;
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int n = 0; n < 3; ++n) chk += acc[j][n][0] + acc[j][n][1] + acc[j][n][2] + acc[j][n][3];
      if (chk == 1.2345e-30f) a.y[0] = chk;
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      return;
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");      // barrier B: every wave has consumed the previous Z
    // ---- Z_w[q] = sum over this wave's two columns of m[a][b] A[b][q]; A = [[1,0],[1,1],[1,-1],[0,-1]] ----
#pragma unroll
    for (int n = 0; n < 3; ++n) {
      const f32x4 m0 = acc[0][n], m1 = acc[1][n];
      const f32x4 z0 = bh ? m0 : m0 + m1;
      const f32x4 z1 = bh ? -m0 - m1 : m1;
#pragma unroll
      for (int r = 0; r < 4; ++r) zw[(n * 4 + r) * 64] = (f32x2){z0[r], z1[r]};
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's share of the next step's rows has landed (and its stores)
    __syncthreads();                                   // barrier A: Z complete, every wave is done with this step's ring rows
  };

  step(s_begin, std::false_type());
  for (int s = s_begin + 1; s < s_end; ++s) step(s, std::true_type());
  if constexpr (!(ABL & 4)) {
#pragma unroll
    for (int n = 0; n < 3; ++n) {
      fin_read(n);
      fin_store(n, s_end - 1);
    }
  }
}

}  // namespace asx
