// Winograd F(2x2, 3x3) for the 3x3 / pad-1 convolutions of the TFC blocks (uvr_lib_v5/modules.py:22-54) on the bf16 matrix pipe
// with fp32 results -- the arithmetic of kernels_gemm3.h applied to the sixteen transform-domain GEMMs of kernels_wino.h:
//
//   Y = A^T [ sum_cin (G g G^T) .* (B^T d B) ] A ,   every fp32 factor v = h + m + l split EXACTLY into three bf16 numbers,
//   u v = uh vh + (uh vm + um vh) + (um vm + uh vl + ul vh)   [dropped: <= 2^-24 |u v|]  -- six `v_mfma_f32_16x16x32_bf16`
//
// (2.67x the fp32 MFMA rate per multiply-add at fp32 rounding).  The transformed weights U = G g G^T are split once on the host
// (fp64 -> fp32 -> three bf16) into MFMA-fragment order; the input transform V = B^T d B happens in fp32 registers and is split
// there (5.5 VALU per value).
//
// What bounds this kernel is operand delivery, not MFMA rate, and the mapping is chosen for that:
//   * a bf16 MFMA wants EIGHT channels per lane (k = 8 lk .. 8 lk + 7), so a stage is 32 input channels;
//   * a weight fragment (1 KiB) must feed >= 4 MFMAs from registers or LDS / L2 bandwidth runs out: a wave therefore owns all
//     64 tiles of the workgroup (4 tile rows x 16) and only TWO of the sixteen positions -- wave w: row a = w >> 1 of the 4 x 4
//     position grid, columns bx = 2 (w & 1), 2 (w & 1) + 1 -- for 48 output channels: 4 x 2 x 3 accumulator tiles = 96 VGPRs;
//   * its 18 weight fragments per stage (2 positions x 3 column tiles x 3 parts) are DISTINCT from every other wave's, so they
//     go straight from L2 into registers (one coalesced 1-KiB load each, no LDS image, no replication inside the workgroup);
//   * the haloed raw planes (10 x 40 floats x 32 channels = 51 KB) arrive by LDS-DMA, double buffered; a wave reads patch rows
//     (ra, rb) of ITS grid row a only: r = d[ra] + sa d[rb] (one fma per value), then its two columns -- no transform work is
//     repeated across the eight waves of the workgroup;
//   * sign convention: grid row / column 2 of B^T is used negated (d1 - d2 instead of d2 - d1) so that every transform step is
//     "first minus second" or "first plus second" with a wave-uniform sign; the host negates the matching positions of U.
//   * the sixteen partial products M_xi of a tile live in eight different waves: the output transform meets in LDS (the raw
//     buffers are free by then), one 16-channel column tile per round, and is finished by (tile column, tile row) lanes so that
//     stores are 128-byte row segments.
// Workgroup = 512 threads (two waves per SIMD, 256 registers each), one per CU; tile = 8 x 32 output pixels; border tiles are masked.
// Round 5: part of libasx.so in its ONE-workgroup-per-item form (template parameter ONE = 1: grid = items, the channel groups of a
// tile consecutive on one XCD), which is what measures faster than conv_wino3_kernel from 144 channels up (per launch, 55 chunks:
// L2 4.84 -> 4.48 ms, L3 2.29 -> 1.84, L4 0.89 -> 0.79, L5 0.32 -> 0.27; L1 8.31 = 8.31, L0 9.0 -> 13.8: profiles/r05_wino6_forms.txt).
// The launcher (asx.hip: conv_launch) therefore routes a 3x3 layer here only when Cin >= the engine's "winograd_bf16x6" option
// (default 144) and keeps conv_wino3_kernel for the wide planes of levels 0 / 1.  The persistent forms (ONE = 0: grid = CUs, the
// next item's first stage prefetched under the epilogue) stay in the header for the harness tools/experimental/proto_wino6.hip.
#pragma once
#include <vector>
#include <cstring>
#include "kernels_net.h"
#include "kernels_gemm3.h"

namespace asx {

struct Wino6Cfg {
  static constexpr int TR = 4, TC = 16, KC = 32, NW = 48, NREP = 3;
  static constexpr int TH = 2 * TR, TW = 2 * TC;
  static constexpr int IH = TH + 2, LP = 3, IWA = 40, C4 = IWA / 4;
  static constexpr int SLOTS = IH * C4;                // 100 float4 slots per plane ...
  static constexpr int PSLOTS = SLOTS + 1;             // ... + one dummy slot: plane stride 404 floats, 8 PS * 4 B = 128 (mod 256)
  static constexpr int PS = 4 * PSLOTS;
  static constexpr int NPIECE = (KC * PSLOTS + 63) / 64;   // 51 wave-issues of 64 slots cover a stage as ONE linear slot array
  static constexpr int RAWF = NPIECE * 256 + 4;        // floats per stage buffer (planes start one float in: 8-byte aligned patches)
  static constexpr int ZCS = 132;                      // exchange image: floats per (wave, cout) row = 64 tiles x 2 + 4
  static constexpr int ZF = 8 * 16 * ZCS;              // one 16-channel round
  // LDS (floats): [stage buffer 0][spare][stage buffer 1][slot table].  The exchange image of an item lives in the buffer its LAST
  // stage used plus the spare (buffer 0 + spare, or spare + buffer 1: contiguous either way), so the other buffer can already
  // receive the first stage of the workgroup's next item.
  static constexpr int SPARE = ((ZF - RAWF + 3) / 4) * 4;
  static constexpr int BUF1 = RAWF + SPARE;
  static constexpr int LDS_FLOATS = 2 * RAWF + SPARE;
  static constexpr int NPW = (NPIECE + 7) / 8;         // pieces per wave (7; the last exists for waves 0 .. 2 only)
  static constexpr int LDS_BYTES = LDS_FLOATS * 4 + NPW * 512 * 4;   // + the per-thread slot table of the DMA
  static_assert(SPARE > 0 && RAWF + SPARE >= ZF, "exchange image");
  static constexpr int WFRAGS = 18;                    // per wave and stage: (q 2)(n 3)(part 3)
  static constexpr int WSTAGE_U4 = 8 * WFRAGS * 64;    // u32x4 per (cg, stage)
};

// ---- host: U = G g G^T (float64 -> float32), signs of grid row / column 2 flipped, three-way bf16 split, fragment order ----
// image[((cg * nci + ci) * 8 + wave) * 18 + (q * 3 + n) * 3 + part][lane][4 x u32]: lane (li, lk) holds cout cg * 48 + n * 16 + li,
// channels ci * 32 + 8 lk .. + 7 of position (a = wave >> 1, bx = 2 (wave & 1) + q).
inline uint32_t wino6_bf16_rne(float v) {
  uint32_t u;
  memcpy(&u, &v, 4);
  const uint32_t r = ((u >> 16) & 1u) + 0x7FFFu;
  return (u + r) >> 16;
}
inline float wino6_bf16_f(uint32_t b) {
  const uint32_t u = b << 16;
  float f;
  memcpy(&f, &u, 4);
  return f;
}
inline void wino6_pack(const float *w, int cout, int cin, std::vector<uint32_t> &img, int *cg_out, int *nci_out) {
  static const double G[4][3] = {{1, 0, 0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0, 0, 1}};
  const int CG = (cout + 47) / 48, NCI = (cin + 31) / 32;
  img.assign((size_t)CG * NCI * Wino6Cfg::WSTAGE_U4 * 4, 0u);
  std::vector<uint16_t> parts((size_t)3 * 16 * CG * 48 * NCI * 32, 0);   // [part][pos][cout_pad][cin_pad]
  const size_t cinp = (size_t)NCI * 32, coutp = (size_t)CG * 48;
  auto P = [&](int part, int pos, size_t co, size_t c) -> uint16_t & { return parts[(((size_t)part * 16 + pos) * coutp + co) * cinp + c]; };
  for (int co = 0; co < cout; ++co)
    for (int c = 0; c < cin; ++c) {
      const float *g = &w[((size_t)co * cin + c) * 9];
      double t[4][3];
      for (int a = 0; a < 4; ++a)
        for (int j = 0; j < 3; ++j) t[a][j] = G[a][0] * g[0 * 3 + j] + G[a][1] * g[1 * 3 + j] + G[a][2] * g[2 * 3 + j];
      for (int a = 0; a < 4; ++a)
        for (int bb = 0; bb < 4; ++bb) {
          double U = t[a][0] * G[bb][0] + t[a][1] * G[bb][1] + t[a][2] * G[bb][2];
          if ((a == 2) != (bb == 2)) U = -U;
          const float u = (float)U;
          const uint32_t h = wino6_bf16_rne(u);
          const float r1 = u - wino6_bf16_f(h);
          const uint32_t m = wino6_bf16_rne(r1);
          const float r2 = r1 - wino6_bf16_f(m);
          const uint32_t l = wino6_bf16_rne(r2);
          P(0, a * 4 + bb, co, c) = (uint16_t)h;
          P(1, a * 4 + bb, co, c) = (uint16_t)m;
          P(2, a * 4 + bb, co, c) = (uint16_t)l;
        }
    }
  for (int cg = 0; cg < CG; ++cg)
    for (int ci = 0; ci < NCI; ++ci)
      for (int wave = 0; wave < 8; ++wave)
        for (int q = 0; q < 2; ++q)
          for (int n = 0; n < 3; ++n)
            for (int part = 0; part < 3; ++part) {
              const int pos = (wave >> 1) * 4 + 2 * (wave & 1) + q;
              const size_t f = (((size_t)cg * NCI + ci) * 8 + wave) * 18 + (q * 3 + n) * 3 + part;
              for (int lane = 0; lane < 64; ++lane) {
                const size_t co = (size_t)cg * 48 + n * 16 + (lane & 15);
                const size_t c0 = (size_t)ci * 32 + (lane >> 4) * 8;
                uint32_t *dst = &img[(f * 64 + lane) * 4];
                for (int e = 0; e < 4; ++e) dst[e] = (uint32_t)P(part, pos, co, c0 + 2 * e) | ((uint32_t)P(part, pos, co, c0 + 2 * e + 1) << 16);
              }
            }
  *cg_out = CG;
  *nci_out = NCI;
}

// ---- fp16 x 3 arithmetic (template parameter H of the kernel; kernels_gemm3.h has the scheme): U scaled per (position, OUTPUT CHANNEL) by a
// power of two (largest |U| over the input channels in [2^14, 2^15): the BatchNorm folded into these weights can make neighbouring channels
// differ by orders of magnitude) and split into TWO fp16 parts; 12 fragments per wave and stage instead of 18; the exponents -- int32
// [cg][wave][q][48 couts] -- follow the fragments (a lane of the accumulator layout owns one cout: applying them is free).
inline uint16_t wino6_f16_rne(float f) {               // float -> IEEE half, round to nearest even (finite inputs below 65520)
  uint32_t x;
  memcpy(&x, &f, 4);
  const uint32_t sign = (x >> 16) & 0x8000u;
  const int32_t ex = (int32_t)((x >> 23) & 0xffu) - 127 + 15;
  uint32_t mant = x & 0x7fffffu;
  if (((x >> 23) & 0xffu) == 0xffu) return (uint16_t)(sign | 0x7c00u | (mant ? 0x200u : 0u));
  if (ex >= 31) return (uint16_t)(sign | 0x7c00u);
  if (ex <= 0) {                                       // subnormal half (or zero)
    if (ex < -10) return (uint16_t)sign;
    mant |= 0x800000u;
    const int shift = 14 - ex;                         // 14 .. 24
    uint32_t m = mant >> shift;
    const uint32_t rem = mant & ((1u << shift) - 1u), half = 1u << (shift - 1);
    if (rem > half || (rem == half && (m & 1u))) ++m;
    return (uint16_t)(sign | m);
  }
  uint32_t m = mant >> 13;
  const uint32_t rem = mant & 0x1fffu;
  uint32_t e2 = (uint32_t)ex;
  if (rem > 0x1000u || (rem == 0x1000u && (m & 1u))) {
    if (++m == 0x400u) {
      m = 0;
      if (++e2 >= 31u) return (uint16_t)(sign | 0x7c00u);
    }
  }
  return (uint16_t)(sign | (e2 << 10) | m);
}
inline float wino6_f16_f(uint16_t h) {
  const uint32_t sign = ((uint32_t)h & 0x8000u) << 16;
  const uint32_t ex = (h >> 10) & 0x1fu, m = h & 0x3ffu;
  float f;
  uint32_t u;
  if (ex == 0) {
    f = ldexpf((float)m, -24);                         // subnormal: m * 2^-24
    memcpy(&u, &f, 4);
    u |= sign;
  } else if (ex == 31) {
    u = sign | 0x7f800000u | (m << 13);
  } else {
    u = sign | ((ex - 15 + 127) << 23) | (m << 13);
  }
  memcpy(&f, &u, 4);
  return f;
}
inline void wino6_pack_h(const float *w, int cout, int cin, std::vector<uint32_t> &img, int *cg_out, int *nci_out) {
  static const double G[4][3] = {{1, 0, 0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0, 0, 1}};
  const int CG = (cout + 47) / 48, NCI = (cin + 31) / 32;
  const size_t cinp = (size_t)NCI * 32, coutp = (size_t)CG * 48;
  std::vector<float> U((size_t)16 * coutp * cinp, 0.f);   // [pos][cout_pad][cin_pad]
  auto Uat = [&](int pos, size_t co, size_t c) -> float & { return U[((size_t)pos * coutp + co) * cinp + c]; };
  for (int co = 0; co < cout; ++co)
    for (int c = 0; c < cin; ++c) {
      const float *g = &w[((size_t)co * cin + c) * 9];
      double t[4][3];
      for (int a = 0; a < 4; ++a)
        for (int j = 0; j < 3; ++j) t[a][j] = G[a][0] * g[0 * 3 + j] + G[a][1] * g[1 * 3 + j] + G[a][2] * g[2 * 3 + j];
      for (int a = 0; a < 4; ++a)
        for (int bb = 0; bb < 4; ++bb) {
          double u = t[a][0] * G[bb][0] + t[a][1] * G[bb][1] + t[a][2] * G[bb][2];
          if ((a == 2) != (bb == 2)) u = -u;
          Uat(a * 4 + bb, co, c) = (float)u;
        }
    }
  const size_t frag_u32 = (size_t)CG * NCI * 8 * 12 * 64 * 4;
  img.assign(frag_u32 + (size_t)CG * 8 * 2 * 48, 0u);
  for (int cg = 0; cg < CG; ++cg)
    for (int wave = 0; wave < 8; ++wave)
      for (int q = 0; q < 2; ++q)
        for (int n = 0; n < 3; ++n) {
          const int pos = (wave >> 1) * 4 + 2 * (wave & 1) + q;
          int exs[16];
          for (int li = 0; li < 16; ++li) {
            float mx = 0.f;
            for (size_t c = 0; c < cinp; ++c) mx = std::max(mx, std::fabs(Uat(pos, (size_t)cg * 48 + n * 16 + li, c)));
            exs[li] = 0;
            if (mx > 0.f) {
              int fe;
              (void)frexpf(mx, &fe);
              exs[li] = 15 - fe;                       // mx 2^ex in [2^14, 2^15)
            }
            img[frag_u32 + (((size_t)cg * 8 + wave) * 2 + q) * 48 + n * 16 + li] = (uint32_t)exs[li];
          }
          for (int ci = 0; ci < NCI; ++ci)
            for (int lane = 0; lane < 64; ++lane) {
              const size_t co = (size_t)cg * 48 + n * 16 + (lane & 15);
              const size_t c0 = (size_t)ci * 32 + (lane >> 4) * 8;
              const int ex = exs[lane & 15];
              uint16_t hh[8], ll[8];
              for (int e = 0; e < 8; ++e) {
                const float us = ldexpf(Uat(pos, co, c0 + e), ex);
                hh[e] = wino6_f16_rne(us);
                ll[e] = wino6_f16_rne(us - wino6_f16_f(hh[e]));
              }
              const size_t f = (((size_t)cg * NCI + ci) * 8 + wave) * 12 + (q * 3 + n) * 2;
              uint32_t *dh = &img[(f * 64 + lane) * 4], *dl = &img[((f + 1) * 64 + lane) * 4];
              for (int e = 0; e < 4; ++e) {
                dh[e] = (uint32_t)hh[2 * e] | ((uint32_t)hh[2 * e + 1] << 16);
                dl[e] = (uint32_t)ll[2 * e] | ((uint32_t)ll[2 * e + 1] << 16);
              }
            }
        }
  *cg_out = CG;
  *nci_out = NCI;
}

// ABL (ASX_WINO6_ABL, measurement-only builds whose results are garbage): 1 = no LDS-DMA after the first stage, 2 = no patch
// reads / input transform / split (operands from registers), 4 = no output exchange / stores, 8 = no weight loads after the first
// stage, 16 = no MFMA
// ONE = 1: one workgroup per ITEM (spatial tile, channel group), grid = 8 * ceil(S / 8) * CG, the CG groups of a tile consecutive on
// one XCD (block b runs on XCD b % 8) so that the re-read of the planes hits that XCD's L2; no cross-item prefetch.
// H: fp16 x 3.  V gets ONE running exponent per (wave, tile row m): the wave-wide largest |V| of the 16 tiles x 2 positions x 32 channels
// a tile row contributes to a stage (DPP reduction), dropping with two bits of headroom; the accumulators of that tile row are multiplied
// by the exact power of two in front of the stage's MFMAs (1.0 almost always); the output exchange reads them back through
// 2^-(e_V[m] + e_U[q][cout of the lane]).
template <int ABL, int PR, int ONE = 0, bool H = false>
__device__ __forceinline__ void wino6_body(const ConvArgs &a, float *lds_f) {
  using CFG = Wino6Cfg;
  static_assert(!H || ABL == 0, "the ablation builds exist for the bf16 x 6 arithmetic only");
  constexpr int NPART = H ? 2 : 3;
  constexpr int WF = 6 * NPART;                        // weight fragments per wave and stage
  constexpr int WST = 8 * WF * 64;                     // u32x4 per (cg, stage)
  constexpr int IWA = CFG::IWA, C4 = CFG::C4, PS = CFG::PS, LP = CFG::LP, SLOTS = CFG::SLOTS, RAWF = CFG::RAWF;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, lk = lane >> 4;
  const int ga = wave >> 1;                            // grid row of this wave's two positions
  constexpr int pr = PR;                               // their column pair (compile time: the caller branches on wave & 1)
  const int ra = ga == 0 ? 0 : 1, rb = ga == 3 ? 3 : 2;
  const float sa = ga == 1 ? 1.f : -1.f;               // r = d[ra] + sa d[rb]   (row 2 negated: d1 - d2)

  // ---- PERSISTENT workgroup: spatial tiles blockIdx.x, blockIdx.x + gridDim.x, ... and, for each, the CG channel groups in turn
  // (the planes of a tile are re-read from L2 by the next group).  The item sequence is one continuous stream of stages: the
  // first stage of item i + 1 (planes and weights) is requested during the last stage of item i and lands under i's epilogue.
  // With one 512-thread workgroup per CU nothing else would cover a workgroup's launch, first-stage latency and epilogue.
  const int S = a.tilesT * a.tilesF * a.B;
  const int nsp = ONE ? 1 : (S - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
  const int one_j = (int)blockIdx.x >> 3, one_sidx = (one_j / a.CG) * 8 + ((int)blockIdx.x & 7);
  const int nwork = ONE ? (one_sidx < S ? 1 : 0) : nsp * a.CG;
  const int64_t plane_sz = (int64_t)a.T * a.F;
  auto decode = [&](int w, int &b, int &to0, int &fo0, int &cg) {
    int sidx = ONE ? one_sidx : (int)blockIdx.x + (int)gridDim.x * (w / a.CG);
    cg = ONE ? one_j % a.CG : w % a.CG;
    const int tf = sidx % a.tilesF;
    sidx /= a.tilesF;
    const int tt = sidx % a.tilesT;
    b = sidx / a.tilesT;
    to0 = tt * CFG::TH;
    fo0 = tf * CFG::TW;
  };

  // ---- LDS-DMA of the raw planes.  A stage buffer is ONE linear array of 16-byte slots, [plane 32][100 slots + 1 dummy], filled
  // by 51 wave-issues of 64 consecutive slots (piece k -> wave k % 8): every lane of every issue is active (a partially masked
  // `global_load_lds` inside divergent control flow was miscompiled into issues with the wrong lane set), out-of-plane and dummy
  // slots read the zero page.  Per lane and piece one packed word: plane << 24 | (t * F + f), or -1 -- in LDS (seven registers per
  // lane were seven too many: spilled, and every reload came with a `vmcnt(0)` that waited for the piece just issued).
  constexpr int NPW = CFG::NPW;
  int *slot_tab = reinterpret_cast<int *>(lds_f + CFG::LDS_FLOATS) + tid;
  auto build_table = [&](int to0, int fo0) {           // read back by the same thread only: no barrier needed
    const int ti0 = to0 - 1, fa0 = fo0 - 1 - LP;
#pragma unroll
    for (int i = 0; i < NPW; ++i) {
      const int sl = (wave + 8 * i) * 64 + lane;
      const int pl = sl / CFG::PSLOTS, rem = sl - pl * CFG::PSLOTS;
      const int row = rem / C4, c4 = rem - row * C4;
      const int t = ti0 + row, f = fa0 + c4 * 4;
      const bool ok = pl < 32 && rem < SLOTS && t >= 0 && t < a.T && f >= 0 && f < a.F;
      slot_tab[i * 512] = ok ? ((pl << 24) | (t * a.F + f)) : -1;
    }
  };
  auto bufbase = [](int p) { return p ? CFG::BUF1 : 0; };
  auto issue = [&](const float *xb, int ci, int bufp) {
    float *raw = lds_f + bufbase(bufp) + 1;
#pragma unroll
    for (int i = 0; i < NPW; ++i) {
      const int k = wave + 8 * i;
      if (k < CFG::NPIECE) {                           // wave-uniform
        const int pk = slot_tab[i * 512];
        const int c = ci * 32 + (pk >> 24);
        const float *src = (pk >= 0 && c < a.Cin) ? xb + (int64_t)c * plane_sz + (pk & 0xffffff) : a.zeros;
        ASX_GLDS16(src, raw + k * 256);
      }
    }
  };

  // ---- weight fragments: L2 -> registers, 18 per stage ----
  const u32x4 *wimg = reinterpret_cast<const u32x4 *>(a.wp) + (int64_t)wave * (WF * 64) + lane;
  u32x4 wr[2][3][NPART];                               // [q][n][part]
  auto load_w = [&](int cg, int ci, int q) {
    const u32x4 *src = wimg + ((int64_t)cg * a.NCI + ci) * WST + q * (3 * NPART * 64);
#pragma unroll
    for (int n = 0; n < 3; ++n)
#pragma unroll
      for (int p = 0; p < NPART; ++p) wr[q][n][p] = src[(n * NPART + p) * 64];
  };

  // patch base of this lane: plane 8 lk, patch row ra / rb of tile row 0, column LP + 1 + 2 li (8-byte aligned)
  const int pbase_a = (8 * lk) * PS + ra * IWA + LP + 1 + 2 * li;
  const int pbase_b = (8 * lk) * PS + rb * IWA + LP + 1 + 2 * li;

  if (nwork <= 0) return;
  int b, to0, fo0, cg;
  decode(0, b, to0, fo0, cg);
  build_table(to0, fo0);
  const float *xb = a.x + (int64_t)b * a.x_bstride;
  issue(xb, 0, 0);
  load_w(cg, 0, 0);
  load_w(cg, 0, 1);
  int gs = 0;                                          // stages run so far: parity = stage buffer

  for (int w = 0; w < nwork; ++w) {
    const bool has_next = w + 1 < nwork;
    int nb = 0, nto0 = 0, nfo0 = 0, ncg = 0;
    if (has_next) decode(w + 1, nb, nto0, nfo0, ncg);

    f32x4 acc[4][2][3];                                // [tile row m][q][n]
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
      for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int n = 0; n < 3; ++n) acc[m][q][n] = (f32x4){0.f, 0.f, 0.f, 0.f};
    int em[4] = {200, 200, 200, 200};                  // H: running exponent of tile row m (wave-uniform)
    int em_lo[4] = {200, 200, 200, 200};               // H: the lowest it has been (rise cap)

    for (int ci = 0; ci < a.NCI; ++ci) {
      // this stage's planes were requested a stage ago, followed (in order) only by the 18 (H: 12) weight loads of this stage and, across
      // an item boundary, the previous item's output stores: at most that many younger requests may be outstanding
      if constexpr (H) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(18)" ::: "memory");
      asm volatile("s_barrier" ::: "memory");
      const float *raw = lds_f + ((ABL & 1) ? 0 : bufbase(gs & 1));
      const bool last = ci + 1 == a.NCI;
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        // ---- V = B^T d B for positions (ga, 2 pr), (ga, 2 pr + 1) of this lane's tile (row m, column li), channels 8 lk .. + 7 ----
        unsigned vh[2][4], vm[2][4], vl[2][4];
        float hv0[4][2], hv1[4][2];                    // H: the tile row's sixteen transformed values, kept until its exponent is known
#pragma unroll
        for (int cp = 0; cp < 4; ++cp) {
          float v0[2], v1[2];
#pragma unroll
          for (int e = 0; e < 2; ++e) {
            f32x2 da0, da1, db0, db1;
            if constexpr (ABL & 2) {
              da0 = (f32x2){(float)(ci + cp), (float)lane};
              da1 = (f32x2){(float)(m + e), (float)(lane - ci)};
              db0 = (f32x2){(float)(ci * cp), 1.f};
              db1 = (f32x2){(float)(lane + m), 2.f};
            } else {
              const float *pa = raw + pbase_a + (2 * cp + e) * PS + (2 * m) * IWA;
              const float *pb = raw + pbase_b + (2 * cp + e) * PS + (2 * m) * IWA;
              da0 = *reinterpret_cast<const f32x2 *>(pa);
              da1 = *reinterpret_cast<const f32x2 *>(pa + 2);
              db0 = *reinterpret_cast<const f32x2 *>(pb);
              db1 = *reinterpret_cast<const f32x2 *>(pb + 2);
            }
            const float x0 = __fmaf_rn(sa, db0.x, da0.x), x1 = __fmaf_rn(sa, db0.y, da0.y);
            const float x2 = __fmaf_rn(sa, db1.x, da1.x), x3 = __fmaf_rn(sa, db1.y, da1.y);
            // columns: pr 0 -> (x0 - x2, x1 + x2); pr 1 -> (x1 - x2 [column 2 negated], x1 - x3)
            v0[e] = pr ? __fsub_rn(x1, x2) : __fsub_rn(x0, x2);
            v1[e] = pr ? __fsub_rn(x1, x3) : __fadd_rn(x1, x2);
          }
          if constexpr (ABL & 2) {
            vh[0][cp] = __float_as_uint(v0[0]); vm[0][cp] = __float_as_uint(v0[1]); vl[0][cp] = __float_as_uint(v1[0]);
            vh[1][cp] = __float_as_uint(v1[1]); vm[1][cp] = __float_as_uint(v0[0] + v1[1]); vl[1][cp] = __float_as_uint(v0[1] + v1[0]);
          } else if constexpr (H) {
            hv0[cp][0] = v0[0];
            hv0[cp][1] = v0[1];
            hv1[cp][0] = v1[0];
            hv1[cp][1] = v1[1];
          } else {
            split3_pair(v0[0], v0[1], vh[0][cp], vm[0][cp], vl[0][cp]);
            split3_pair(v1[0], v1[1], vh[1][cp], vm[1][cp], vl[1][cp]);
          }
        }
        if constexpr (H) {
          float mx = 0.f;
#pragma unroll
          for (int cp = 0; cp < 4; ++cp)
            mx = fmaxf(mx, fmaxf(fmaxf(fabsf(hv0[cp][0]), fabsf(hv0[cp][1])), fmaxf(fabsf(hv1[cp][0]), fabsf(hv1[cp][1]))));
          mx = wave_max64(mx);
          const int need = __builtin_amdgcn_readfirstlane(f16_scale_exp(mx));
          // down at once, up again when the chunk's largest element is F16X3_RISE bits below the range (kernels_gemm3.h split_chunk_h), at most
          // F16X3_RISE_CAP bits above the tile row's lowest exponent so far
          int e_new = em[m];
          if (need < em[m]) e_new = need - 2;
          else if (need > em[m] + F16X3_RISE && mx > 0.f) e_new = min(need - 2, em_lo[m] + F16X3_RISE_CAP);
          em_lo[m] = min(em_lo[m], e_new);
          // the exponent drops in few stages of a tile: the 24 multiplies by 2^(e_new - e_old) sit behind the wave-uniform test (in-place
          // inline assembly -- as C++ inside a branch the compiler may keep a second accumulator set, kernels_gemm3.h stage_h)
#ifdef ASX_WINO6_RESCALE_ALWAYS
          const float fdev = __builtin_ldexpf(1.0f, e_new - em[m]);
#pragma unroll
          for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int n = 0; n < 3; ++n) acc[m][q][n] *= fdev;
#else
          if (e_new != em[m]) {
            const float fdev = __builtin_ldexpf(1.0f, e_new - em[m]);
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
              for (int n = 0; n < 3; ++n)
                asm volatile("v_mul_f32 %0, %4, %0\n\tv_mul_f32 %1, %4, %1\n\tv_mul_f32 %2, %4, %2\n\tv_mul_f32 %3, %4, %3"
                             : "+v"(acc[m][q][n].x), "+v"(acc[m][q][n].y), "+v"(acc[m][q][n].z), "+v"(acc[m][q][n].w)
                             : "v"(fdev));
          }
#endif
          em[m] = e_new;
#pragma unroll
          for (int cp = 0; cp < 4; ++cp) {
            split2h_pair(hv0[cp][0], hv0[cp][1], e_new, vh[0][cp], vl[0][cp]);
            split2h_pair(hv1[cp][0], hv1[cp][1], e_new, vh[1][cp], vl[1][cp]);
          }
        }
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          if constexpr (H) {
            const f16x8 fh = __builtin_bit_cast(f16x8, (u32x4){vh[q][0], vh[q][1], vh[q][2], vh[q][3]});
            const f16x8 fl = __builtin_bit_cast(f16x8, (u32x4){vl[q][0], vl[q][1], vl[q][2], vl[q][3]});
#pragma unroll
            for (int n = 0; n < 3; ++n) acc[m][q][n] = ASX_MFMA_F16(fl, __builtin_bit_cast(f16x8, wr[q][n][0]), acc[m][q][n]);
#pragma unroll
            for (int n = 0; n < 3; ++n) acc[m][q][n] = ASX_MFMA_F16(fh, __builtin_bit_cast(f16x8, wr[q][n][NPART - 1]), acc[m][q][n]);
#pragma unroll
            for (int n = 0; n < 3; ++n) acc[m][q][n] = ASX_MFMA_F16(fh, __builtin_bit_cast(f16x8, wr[q][n][0]), acc[m][q][n]);
          }
          if constexpr (!H) {
          const bf16x8 ah = __builtin_bit_cast(bf16x8, (u32x4){vh[q][0], vh[q][1], vh[q][2], vh[q][3]});
          const bf16x8 am = __builtin_bit_cast(bf16x8, (u32x4){vm[q][0], vm[q][1], vm[q][2], vm[q][3]});
          const bf16x8 al = __builtin_bit_cast(bf16x8, (u32x4){vl[q][0], vl[q][1], vl[q][2], vl[q][3]});
          if constexpr (ABL & 16) {
#pragma unroll
            for (int n = 0; n < 3; ++n)
              acc[m][q][n].x += (float)ah[0] + (float)am[1] + (float)al[2] + __uint_as_float(wr[q][n][0].x) + __uint_as_float(wr[q][n][1].y) +
                                __uint_as_float(wr[q][n][2].z);
          } else {
#pragma unroll
            for (int n = 0; n < 3; ++n) acc[m][q][n] = ASX_MFMA_BF16(al, __builtin_bit_cast(bf16x8, wr[q][n][0]), acc[m][q][n]);
#pragma unroll
            for (int n = 0; n < 3; ++n) acc[m][q][n] = ASX_MFMA_BF16(ah, __builtin_bit_cast(bf16x8, wr[q][n][NPART - 1]), acc[m][q][n]);
#pragma unroll
            for (int n = 0; n < 3; ++n) acc[m][q][n] = ASX_MFMA_BF16(am, __builtin_bit_cast(bf16x8, wr[q][n][1]), acc[m][q][n]);
#pragma unroll
            for (int n = 0; n < 3; ++n) acc[m][q][n] = ASX_MFMA_BF16(am, __builtin_bit_cast(bf16x8, wr[q][n][0]), acc[m][q][n]);
#pragma unroll
            for (int n = 0; n < 3; ++n) acc[m][q][n] = ASX_MFMA_BF16(ah, __builtin_bit_cast(bf16x8, wr[q][n][1]), acc[m][q][n]);
#pragma unroll
            for (int n = 0; n < 3; ++n) acc[m][q][n] = ASX_MFMA_BF16(ah, __builtin_bit_cast(bf16x8, wr[q][n][0]), acc[m][q][n]);
          }
          }
          if (m == 3 && (!last || has_next)) {
            // the fragments of position q are dead: fetch the next stage's (of this item, or the first of the next item)
            asm volatile("" ::: "memory");
            if constexpr (!(ABL & 8)) {
              if (!last) load_w(cg, ci + 1, q);
              else load_w(ncg, 0, q);
            }
            asm volatile("" ::: "memory");
          }
        }
        if (m == 0 && (!last || has_next)) {
          // next stage's planes: issued once every weight fragment of this stage has been waited for (a counter wait the compiler
          // places in front of an MFMA must never cover them), with three quarters of the stage left to land.  The loop-top
          // `vmcnt(18)` relies on the order [plane pieces][18 weight loads] in the queue.
          asm volatile("" ::: "memory");
          if constexpr (!(ABL & 1)) {
            if (!last) {
              issue(xb, ci + 1, (gs + 1) & 1);
            } else {
              if (ncg == 0) build_table(nto0, nfo0);   // a new spatial tile (this item's own issues are all done)
              issue(a.x + (int64_t)nb * a.x_bstride, 0, (gs + 1) & 1);
            }
          }
          asm volatile("" ::: "memory");
        }
      }
      ++gs;
    }

    // ---- output transform across the eight waves: Y = A^T M A, A^T = [[1 1 1 0], [0 1 -1 -1]] ----------------------------------
    // this wave: z[q'] = sum_bx M[ga][bx] A[bx][q'] over ITS two columns: pr 0: (M0 + M1, M1); pr 1: (M2, -M2 - M3)
    // Raw barriers + LDS-counter waits only: a `__syncthreads()` would also drain the next item's requests.
    if constexpr ((ABL & 4) != 0) {
      float chk = 0.f;
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
          for (int n = 0; n < 3; ++n) chk += acc[m][q][n][0] + acc[m][q][n][1] + acc[m][q][n][2] + acc[m][q][n][3];
      if (chk == 1.2345e-30f) a.y[0] = chk;
    } else {
      float *yb = a.y + (int64_t)b * a.y_bstride;
      const float *rbp = a.res ? a.res + (int64_t)b * a.aux_bstride : nullptr;
      float *zx = lds_f + (((gs - 1) & 1) ? RAWF : 0);   // last stage's buffer + the spare
      int ew[2][3] = {{0, 0, 0}, {0, 0, 0}};             // H: exponents of this LANE's couts (li) under the wave's two positions
      if constexpr (H) {
        // vector loads: fine in the one-workgroup-per-item form (nothing of a next item is in flight); the persistent forms of the
        // harness would see the `vmcnt(0)` these bring drain their prefetch
        const int *ep = reinterpret_cast<const int *>(a.wp) + (int64_t)a.CG * a.NCI * WST * 4 + ((int64_t)cg * 8 + wave) * 96 + li;
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
          for (int n = 0; n < 3; ++n) ew[q][n] = ep[q * 48 + n * 16];
      }
#pragma unroll
      for (int n = 0; n < 3; ++n) {
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // raw planes / the previous round's exchange image are dead
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float m0 = acc[m][0][n][r], m1 = acc[m][1][n][r];
            if constexpr (H) {                         // back to the operands' own scale (exact)
              m0 = __builtin_ldexpf(m0, -(em[m] + ew[0][n]));
              m1 = __builtin_ldexpf(m1, -(em[m] + ew[1][n]));
            }
            f32x2 z;
            z.x = pr ? m0 : m0 + m1;
            z.y = pr ? -m0 - m1 : m1;
            *reinterpret_cast<f32x2 *>(&zx[(wave * 16 + li) * CFG::ZCS + ((m * 4 + r) * 4 + lk) * 2]) = z;
          }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        // reducer: wave -> couts 2 wave, 2 wave + 1 of this column tile; lane -> (tile column tc = lane & 15, tile row m = lane >> 4)
        const int tc = lane & 15, tm = lane >> 4;
        const int toff = ((tm * 4 + (tc & 3)) * 4 + (tc >> 2)) * 2;
#pragma unroll
        for (int cc = 0; cc < 2; ++cc) {
          const int cl = 2 * wave + cc;
          const int co = cg * 48 + n * 16 + cl;
          float y00 = 0.f, y01 = 0.f, y10 = 0.f, y11 = 0.f;
#pragma unroll
          for (int ww = 0; ww < 8; ++ww) {
            const f32x2 z = *reinterpret_cast<const f32x2 *>(&zx[(ww * 16 + cl) * CFG::ZCS + toff]);
            const int g = ww >> 1;
            if (g != 3) { y00 += z.x; y01 += z.y; }
            if (g == 1) { y10 += z.x; y11 += z.y; }
            if (g >= 2) { y10 -= z.x; y11 -= z.y; }
          }
          if (co >= a.Cout) continue;
          // scalar load: a vector load here would bring a `vmcnt(0)` that also waits for the next item's planes and weights
          float bv;
          {
            const uint64_t bpv = reinterpret_cast<uint64_t>(a.bias + co);   // wave-uniform: make that explicit
            const uint32_t blo = __builtin_amdgcn_readfirstlane((uint32_t)bpv), bhi = __builtin_amdgcn_readfirstlane((uint32_t)(bpv >> 32));
            const uint64_t bps = ((uint64_t)bhi << 32) | blo;
            asm volatile("s_load_dword %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(bv) : "s"(bps) : "memory");
          }
          const int t = to0 + 2 * tm, f = fo0 + 2 * tc;
#pragma unroll
          for (int pq = 0; pq < 2; ++pq) {
            if (t + pq >= a.To || f >= a.Fo) continue;
            const int64_t off = ((int64_t)co * a.To + t + pq) * a.Fo + f;
            float o0 = act_fn((pq ? y10 : y00) + bv, a.act), o1 = act_fn((pq ? y11 : y01) + bv, a.act);
            if (f + 1 < a.Fo && (a.Fo & 1) == 0) {
              f32x2 v = {o0, o1};
              if (rbp != nullptr) v += *reinterpret_cast<const f32x2 *>(rbp + off);
              *reinterpret_cast<f32x2 *>(yb + off) = v;
            } else {
              yb[off] = o0 + (rbp != nullptr ? rbp[off] : 0.f);
              if (f + 1 < a.Fo) yb[off + 1] = o1 + (rbp != nullptr ? rbp[off + 1] : 0.f);
            }
          }
        }
      }
    }
    if (has_next) {
      b = nb;
      to0 = nto0;
      fo0 = nfo0;
      cg = ncg;
      xb = a.x + (int64_t)b * a.x_bstride;
    }
  }
}

template <int ABL = 0, int ONE = 0, bool H = false>
__global__ __launch_bounds__(512, 1) void conv_wino6_kernel(ConvArgs a) {
  extern __shared__ float lds_f[];
  // the column pair of a wave's positions decides which transform columns it forms: two instantiations of the body, chosen by a
  // wave-uniform branch (both run the same barrier sequence)
  if (__builtin_amdgcn_readfirstlane(threadIdx.x >> 6) & 1) wino6_body<ABL, 1, ONE, H>(a, lds_f);
  else wino6_body<ABL, 0, ONE, H>(a, lds_f);
}

}  // namespace asx
