// 3x3 / pad 1 convolution of the TFC blocks (uvr_lib_v5/modules.py:11-17, mdxnet.py:97-113) as a DIRECT implicit GEMM on the fp16
// matrix pipe with the fp16 x 3 arithmetic of kernels_gemm3.h (two-part operands, three products, fp32 accumulation) -- no Winograd
// transforms, no cross-wave exchange.  One kernel for 48 -> 48 channels; layers of 96 / 144 channels run as 4 / 9 launches over 48-channel
// slices (accumulate mode).  DESIGN.md 6k has the design record and the measurements behind every choice below.
//
//   y[b, co, t, f] = act(bias[co] + prev[b, co, t, f] + sum_{ci, ky, kx} w[co, ci, ky, kx] x[b, ci, t + ky - 1, f + kx - 1])        [B, C, T, F] fp32, F fastest
//
// GEMM view.  M = output pixels (16 consecutive f per MFMA tile), N = output channels (3 tiles of 16), K = (ky, kx, ci): a kernel row is
// 3 x 48 = 144 = 18 groups of eight channels = 4.5 stages of `v_mfma_f32_16x16x32_f16`, padded to 5 (a stage never straddles input rows: rows of
// different blocks carry different exponents) -- 15 stages per tile.  A lane's eight K values are eight consecutive input channels of ONE
// tap at ONE pixel, so the x operand is kept channels-LAST in LDS ([ring row][34 pixels][48 halves], 96 bytes per pixel and part).
//
// One persistent 512-thread workgroup per CU (grid 256), two kinds of waves, ONE `s_barrier` per 4 x 32 output tile:
//   * waves 4-7 PRODUCE a block of four input rows x 40 columns x 48 channels per step: 240 (row, 8-channel group, column quad) items, one per
//     lane, eight `buffer_load_dwordx4` each (out-of-image rows / columns through the buffer bounds check = 0).  Four register sets: block
//     s + 4 is fetched, block s + 2's largest |x| reduced (v_max3 + DPP in a wave, a four-entry LDS table across waves, published by the step
//     barrier), block s + 1 scaled by the walk's running power of two, split h + l (`v_fma_mixlo / hi_f16`) and written to its slot of a
//     12-row ring (`ds_write_b128`, 2-way at worst);
//   * waves 0-3 CONSUME block s: wave r owns output row r of the tile (two 16-pixel MFMA tiles x three 16-channel tiles = six accumulators);
//     per stage it reads six weight fragments and four x fragments (`ds_read_b128`, a stage ahead) and issues 18 MFMAs (w_l x_h, w_h x_l,
//     w_h x_h); where the exponent moved between the tile's two blocks, the accumulators are rescaled at the kernel-row boundary.  The finished
//     tile's epilogue and its six whole-line nontemporal stores run among the MFMAs of the next tile.
//   The two-part weight image (432 x 48 x 2 x 2 B = 81 KB, fragment order, one exponent per OUTPUT channel) is copied into LDS once per
//   workgroup and stays for the whole launch.  LDS: 82944 (W) + 78336 (ring) + tables = 161.3 KB of the 160 KiB.
// Bank conflicts: a 16-pixel fragment read at the 96-byte pixel stride puts the sixteen lanes of every `ds_read_b128` service group on
// sixteen distinct 16-byte slots (pixel stride 6 slots: p and p + 8 collide, and a group holds p mod 8 all different for each k group;
// the two k groups of a service group are an odd number of slots apart) -- checked in tests/test_conv3h_host.py.
#pragma once
#include <cstring>
#include <vector>

#include "kernels_gemm3.h"

namespace asx {

struct Conv3hArgs {
  const float *x;            // [B, 48, T, F] view (x_bstride floats between batch items)
  float *y;                  // [B, 48, T, F] view
  const u32x4 *wimg;         // conv3h_pack image: fragments, then int32 exponents [48]
  const float *bias;         // [48]
  int B, T, F;
  int64_t x_bstride, y_bstride;
  int act;                   // ACT_NONE / ACT_RELU
  int tilesT, tilesF;        // ceil(T / 4), F / 32
  int bw, tps;               // walk geometry: strips per band (32, 16 or 8: an XCD's 32 workgroups take bw strips x 32 / bw segments of T), tiles per segment (tilesT * bw / 32)
  const float *prev;         // nullptr, or a [B, 48, T, F] view (y's strides) ADDED in front of the activation: the partial sum of another 48-channel slice of the input
  long long *dbg;            // ABL & 32 (timeline build): [64 steps][8] s_memtime stamps of workgroup 0, then [4 workgroups][2048] step starts
};

struct Conv3hCfg {
  static constexpr int C = 48, CG8 = C / 8;                       // channels, groups of eight
  static constexpr int TH = 4, TW = 32, IW = TW + 2;
  static constexpr int PSTR = C * 2;                              // bytes per pixel and part
  static constexpr int ROWB = IW * PSTR;                          // 3264 bytes per ring row and part
  static constexpr int RING = 12;                                 // ring rows: three blocks of four
  static constexpr int PART = RING * ROWB;                        // 39168 bytes per part
  static constexpr int KGY = 3 * CG8;                             // 18 k groups per kernel row
  static constexpr int SPK = (KGY + 3) / 4;                       // 5 stages per kernel row, the last one half empty
  static constexpr int NST = 3 * SPK;                             // 15 stages
  static constexpr int WKY = (KGY / 4) * 3 * 2 * 1024 + 3 * 2 * 512;   // 27648 bytes of weight fragments per kernel row
  static constexpr int WBYTES = 3 * WKY;                          // 82944
  static constexpr int W_OFF = 0, X_OFF = WBYTES, TAB_OFF = X_OFF + 2 * PART;
  static constexpr int LDS_BYTES = TAB_OFF + 64;                  // maxima [2][4] floats, exponents [3] ints, 16 bytes of zeros at + 48
  static constexpr size_t IMG_U32 = WBYTES / 4 + 48;              // image size in 32-bit words
  static constexpr int EUP = 8;                                   // the running exponent of a walk follows a quieter block only when it is more than this many bits quieter
};

// host: w [48, 48, 3, 3] fp32 -> the LDS image.  Stage (ky, sg) = k groups 4 sg .. 4 sg + 3 of kernel row ky, k group kk = (kx = kk / 6, eight
// input channels kk % 6); fragment (channel tile n, part p) of a full stage at ky * WKY + sg * 6144 + (n * 2 + p) * 1024 bytes, of the half stage
// (sg = 4, k groups 16 and 17) at ky * WKY + 24576 + (n * 2 + p) * 512; lane l of a fragment: output channel n * 16 + (l & 15), k group 4 sg + (l >> 4).
inline uint16_t conv3h_f16_rne(float f) {
  uint32_t x;
  memcpy(&x, &f, 4);
  const uint32_t sign = (x >> 16) & 0x8000u;
  const int32_t ex = (int32_t)((x >> 23) & 0xffu) - 127 + 15;
  uint32_t mant = x & 0x7fffffu;
  if (((x >> 23) & 0xffu) == 0xffu) return (uint16_t)(sign | 0x7c00u | (mant ? 0x200u : 0u));
  if (ex >= 31) return (uint16_t)(sign | 0x7c00u);
  if (ex <= 0) {
    if (ex < -10) return (uint16_t)sign;
    mant |= 0x800000u;
    const int shift = 14 - ex;
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
inline float conv3h_f16_f(uint16_t h) {
  const uint32_t sign = ((uint32_t)h & 0x8000u) << 16;
  const uint32_t ex = (h >> 10) & 0x1fu, m = h & 0x3ffu;
  uint32_t u;
  if (ex == 0) {
    if (m == 0) {
      u = sign;
    } else {
      int sh = 0;
      uint32_t mm = m;
      while (!(mm & 0x400u)) {
        mm <<= 1;
        ++sh;
      }
      u = sign | ((uint32_t)(127 - 15 - sh + 1) << 23) | ((mm & 0x3ffu) << 13);
    }
  } else if (ex == 31) {
    u = sign | 0x7f800000u | (m << 13);
  } else {
    u = sign | ((ex - 15 + 127) << 23) | (m << 13);
  }
  float f;
  memcpy(&f, &u, 4);
  return f;
}

inline void conv3h_pack(const float *w, std::vector<uint32_t> &img) {
  using CFG = Conv3hCfg;
  constexpr int C = CFG::C;
  img.assign(CFG::IMG_U32, 0u);
  int ex[C];
  for (int co = 0; co < C; ++co) {
    float mx = 0.f;
    for (int i = 0; i < C * 9; ++i) {
      const float v = std::fabs(w[(size_t)co * C * 9 + i]);
      if (v <= 3.4028234663852886e38f) mx = std::max(mx, v);
    }
    ex[co] = 0;
    if (mx > 0.f) {
      int fe;
      (void)frexpf(mx, &fe);
      ex[co] = 15 - fe;                                // mx 2^ex in [2^14, 2^15)
    }
    img[CFG::WBYTES / 4 + co] = (uint32_t)ex[co];
  }
  for (int ky = 0; ky < 3; ++ky)
    for (int sg = 0; sg < CFG::SPK; ++sg) {
      const bool half = sg == CFG::SPK - 1;
      const int nl = half ? 32 : 64;
      for (int n = 0; n < 3; ++n)
        for (int lane = 0; lane < nl; ++lane) {
          const int co = n * 16 + (lane & 15);
          const int kk = 4 * sg + (lane >> 4);
          const int tap = ky * 3 + kk / CFG::CG8, c0 = (kk % CFG::CG8) * 8;
          uint16_t hh[8], ll[8];
          for (int e = 0; e < 8; ++e) {
            const float us = ldexpf(w[((size_t)co * C + c0 + e) * 9 + tap], ex[co]);
            hh[e] = conv3h_f16_rne(us);
            ll[e] = conv3h_f16_rne(us - conv3h_f16_f(hh[e]));
          }
          const size_t pstep = half ? 512 : 1024;
          const size_t fb = (size_t)ky * CFG::WKY + (half ? (size_t)(CFG::SPK - 1) * 6144 + (size_t)(n * 2) * 512 : (size_t)sg * 6144 + (size_t)(n * 2) * 1024);
          uint32_t *dh = &img[(fb + (size_t)lane * 16) / 4], *dl = &img[(fb + pstep + (size_t)lane * 16) / 4];
          for (int e = 0; e < 4; ++e) {
            dh[e] = (uint32_t)hh[2 * e] | ((uint32_t)hh[2 * e + 1] << 16);
            dl[e] = (uint32_t)ll[2 * e] | ((uint32_t)ll[2 * e + 1] << 16);
          }
        }
    }
}

// Block walk.  XCD x (block id & 7: where the dispatcher puts the workgroup, for speed only) takes the (b, band of bw strips) items x, x + 8, ...;
// its 32 workgroups (block id >> 3) are bw adjacent strips x 32 / bw segments of T and walk their segment down T together, so the halo
// columns a strip shares with its neighbours are re-read from that XCD's L2 (bw = 32: one segment, planes at least 1024 wide; 16 / 8 for the
// narrower planes of the deeper levels, where 32 strips would leave workgroups without work).  A segment is tps + 1 blocks of four input rows:
// block j = rows r0 + 4 j + 1 .. r0 + 4 j + 4, j = -1 .. tps - 1 (r0 = the segment's first row); output tile j (rows r0 + 4 j .. + 3) needs rows
// r0 + 4 j - 1 .. r0 + 4 j + 4 = the last two rows of block j - 1 and block j.  Strips past tilesF are all-padding (never stored): they keep the
// walk in step.  The grid is 256 workgroups.
struct Conv3hWalk {
  int item, j, b, strip;
};
__device__ __forceinline__ void conv3h_walk_item(const Conv3hArgs &a, int wg, Conv3hWalk &w) {
  const int nbands = (a.tilesF + a.bw - 1) / a.bw;
  w.b = w.item / nbands;
  w.strip = (w.item - w.b * nbands) * a.bw + (wg >> 3) % a.bw;
}
__device__ __forceinline__ int conv3h_row0(const Conv3hArgs &a, int wg) { return ((wg >> 3) / a.bw) * a.tps * 4; }   // first row of the workgroup's segment
__device__ __forceinline__ int conv3h_nblocks(const Conv3hArgs &a, int wg) {
  const int items = a.B * ((a.tilesF + a.bw - 1) / a.bw), x = wg & 7;
  return x < items ? ((items - x + 7) >> 3) * (a.tps + 1) : 0;
}
__device__ __forceinline__ void conv3h_walk_init(const Conv3hArgs &a, int wg, Conv3hWalk &w) {
  w.item = wg & 7;
  w.j = -1;
  conv3h_walk_item(a, wg, w);
}
__device__ __forceinline__ void conv3h_walk_next(const Conv3hArgs &a, int wg, Conv3hWalk &w) {
  if (++w.j == a.tps) {
    w.j = -1;
    w.item += 8;
    conv3h_walk_item(a, wg, w);
  }
}

// h + l fp16 parts of x0 s, x1 s (s a power of two): four single-issue VALU instructions for two elements -- `v_fma_mix{lo,hi}_f16` computes
// the fp32 fma and rounds it to a half of the destination: h = RNE_f16(x s + 0), l = RNE_f16(x s - h) (x s and the difference are exact in
// fp32: the same values as kernels_gemm3.h's split2h_pair, which multiplies, converts, converts back, subtracts and converts).
__device__ __forceinline__ void conv3h_split_pair(float x0, float x1, float s, unsigned &h, unsigned &l) {
  unsigned hh = 0, ll = 0;
  asm("v_fma_mixlo_f16 %0, %1, %2, 0 op_sel_hi:[0,0,0]" : "+v"(hh) : "v"(x0), "v"(s));
  asm("v_fma_mixhi_f16 %0, %1, %2, 0 op_sel_hi:[0,0,0]" : "+v"(hh) : "v"(x1), "v"(s));
  asm("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel:[0,0,0] op_sel_hi:[0,0,1]" : "+v"(ll) : "v"(x0), "v"(s), "v"(hh));
  asm("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(ll) : "v"(x1), "v"(s), "v"(hh));
  h = hh;
  l = ll;
}

// `keep` in the lanes outside bank mask BM (bit k: lanes 4 k .. 4 k + 3 of every 16), lane l ^ 8's `from` inside it
template <int BM>
__device__ __forceinline__ float ror8m(float keep, float from) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, keep), __builtin_bit_cast(int, from), 0x128, 0xf, BM, false));   // row_ror:8
}

// ABL (measurement-only builds, results are garbage): 2 = no global loads, 4 = no stores, 8 = no split / LDS writes, 32 = timeline stamps
// ACC: accumulate mode (a.prev != nullptr): its own instantiation -- a run-time test in front of the six loads ended the MFMA scheduling region
// and cost the plain launches a dozen register moves per tile
template <int ABL = 0, bool ACC = false>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void conv3h_kernel(Conv3hArgs a) {
  using CFG = Conv3hCfg;
  extern __shared__ float lds_f[];
  char *lds = reinterpret_cast<char *>(lds_f);
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wg = blockIdx.x;
  float *maxtab = reinterpret_cast<float *>(lds + CFG::TAB_OFF);   // [2][4]: block parity, producer wave
  int *exps = reinterpret_cast<int *>(lds + CFG::TAB_OFF + 32);    // [3]: block % 3

  // ---- weights: global image -> LDS, once
  {
    const u32x4 *src = a.wimg;
    u32x4 *dst = reinterpret_cast<u32x4 *>(lds + CFG::W_OFF);
    for (int i = tid; i < CFG::WBYTES / 16; i += 512) dst[i] = src[i];
  }
  const int64_t TF = (int64_t)a.T * a.F;
  const int NB = conv3h_nblocks(a, wg);
  const int row0 = conv3h_row0(a, wg);
  const unsigned plane_bytes = (unsigned)(CFG::C * TF * 4);

  if (wave >= 4) {
    // =================================================================== producer ===================================================
    // Items of a block: (row 0..3, 8-channel group 0..5, aligned quad of columns 0..9 = image columns f0 - 4 + 4 q .. + 3; the 34-column
    // window is columns f0 - 1 .. f0 + 32: quad 0 gives its last pixel, quad 9 its first, the others all four) -- 240 items, ONE per
    // producer lane, eight `buffer_load_dwordx4` each (one per channel plane).  Lane order inside an 8-lane group: bit 0 = quad parity
    // (lane pairs fetch 32 contiguous bytes: a vector-memory instruction costs ~45 cycles of the CU's address unit that way against
    // ~70 with every lane on its own line, tools/experimental/micro_ta.hip), bits 1-2 = channel groups 0..3 (items 0..159) or channel
    // group 4 / 5 x row parity (items 160..239): the eight lanes of a `ds_write_b128` service group land on four 16-byte slots twice
    // each (2-way: 16 LDS cycles, the instruction costs 13 anyway) instead of on one slot eight times.
    const int ptid = tid - 256, pw = wave - 4;
    int row, cig, q;
    if (ptid < 160) {
      q = (ptid & 1) + 2 * ((ptid >> 3) % 5);
      cig = (ptid >> 1) & 3;
      row = ptid / 40;
    } else {
      const int u = ptid - 160;
      q = (u & 1) + 2 * ((u >> 3) % 5);
      cig = 4 + ((u >> 1) & 1);
      row = ((u >> 2) & 1) + 2 * (u / 40);
    }
    const bool item_ok = ptid < 240;
    const int loff = (row * CFG::IW + 4 * q - 3) * CFG::PSTR + cig * 16;   // pixel 0 of the quad inside a block slot (window column 4 q - 3: never written for quad 0)
    const int goff = (int)(((int64_t)cig * 8 * a.T + row) * a.F + 4 * q);  // floats from (channel 0, row 4 j + 1, column f0 - 4); < 2^29 (launcher)
    f32x4 raw[4][8];                                   // four register sets: block m lives in set m % 4 (fetched in step m - 4, its maximum taken in step m - 2, split in step m - 1)

    Conv3hWalk wk;
    conv3h_walk_init(a, wg, wk);
    int nf = 0;                                        // blocks fetched so far
    auto fetch = [&](auto setc) {                      // the next block of the walk (nothing past the last one: every offset out of range)
      constexpr int S = decltype(setc)::value;
      const int tb = nf < NB ? wk.b : 0, f0 = wk.strip * 32;
      const int t1 = nf < NB ? row0 + wk.j * 4 + 1 : -(1 << 20);
      ++nf;
      conv3h_walk_next(a, wg, wk);
      __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.x + (int64_t)tb * a.x_bstride), 0, plane_bytes, 0x00020000);
      const int org = t1 * a.F + (f0 - 4);
      // F % 4 == 0 and the quad is aligned: inside the row or outside as a whole
      const bool ok = item_ok && (unsigned)(t1 + row) < (unsigned)a.T && (unsigned)(f0 - 4 + 4 * q) < (unsigned)a.F;
      const unsigned vo = ok ? (unsigned)(goff + org) * 4u : 0xfffffff0u;   // past num_records: the load returns 0
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        if constexpr ((ABL & 2) != 0) raw[S][j] = (f32x4){0.25f, 0.5f, 0.75f, 1.f};
        else raw[S][j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)vo, (int)(j * TF * 4), 0));
      }
    };
    auto tile_max = [&](auto setc, int slot) {         // largest finite |x| of the block held in set S -> maxtab[slot][pw]
      constexpr int S = decltype(setc)::value;
      float m = 0.f;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        m = __builtin_fmaxf(__builtin_fmaxf(m, __builtin_fabsf(raw[S][j].x)), __builtin_fabsf(raw[S][j].y));   // v_max3_f32 with |.| modifiers
        m = __builtin_fmaxf(__builtin_fmaxf(m, __builtin_fabsf(raw[S][j].z)), __builtin_fabsf(raw[S][j].w));
      }
      m = wave_max64(m);
      if (!(m <= 3.4028234663852886e38f)) {            // an Inf in the block (v_max never returns a NaN): the largest FINITE |x| sets the exponent --
        m = 0.f;                                       // the Inf / NaN elements poison the outputs they reach and nothing else
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          m = fmaxf(m, fmaxf(fmaxf(finite_or_zero(raw[S][j].x), finite_or_zero(raw[S][j].y)), fmaxf(finite_or_zero(raw[S][j].z), finite_or_zero(raw[S][j].w))));
        }
        m = wave_max64(m);
      }
      if (lane == 0) maxtab[slot * 4 + pw] = m;
    };
    int e_prev = 0, js = -1;                           // exponent of the block split last, position of the next block to split inside its item
    auto split_store = [&](auto setc, int par, int slot3) {   // set S -> ring rows of block slot `slot3` under the block's exponent
      constexpr int S = decltype(setc)::value;
      const f32x4 mv = *reinterpret_cast<const f32x4 *>(maxtab + par * 4);
      const float m = fmaxf(fmaxf(mv.x, mv.y), fmaxf(mv.z, mv.w));
      // Running exponent of the walk down T: it follows a block's need (largest |x| 2^need in [2^14, 2^15)) only when the block would
      // overflow (need < e: one more bit of headroom is taken, so the next few louder blocks pass) or is more than EUP bits quieter --
      // an element keeps its 22 significand bits down to 2^-12 of a block maximum that sits EUP = 8 bits under the range, and an
      // absolute error of 2^-34 of that maximum below: the consumers rescale accumulators only where the exponent moves (rarely).
      int need = f16_scale_exp(m) - 1;
      need = need < 100 ? need : 100;                  // 2^e and the epilogue's 2^-(e + weight exponent) stay normal floats
      int e = e_prev;
      if (js < 0 || need < e_prev) e = need;
      else if (need > e_prev + CFG::EUP) e = need < e_prev + 40 ? need : e_prev + 40;   // (a rise multiplies accumulators by 2^rise: bounded)
      e_prev = e;
      if (++js == a.tps) js = -1;
      if (ptid == 0) exps[slot3] = e;
      const float sc = __builtin_ldexpf(1.0f, e);
      char *dst = lds + CFG::X_OFF + slot3 * 4 * CFG::ROWB + loff;
      if constexpr ((ABL & 8) != 0) return;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        unsigned hh[4], ll[4];
#pragma unroll
        for (int c2 = 0; c2 < 4; ++c2) conv3h_split_pair(raw[S][2 * c2][i], raw[S][2 * c2 + 1][i], sc, hh[c2], ll[c2]);
        const bool wok = item_ok && (i == 3 ? q <= 8 : (i == 0 ? q >= 1 : (q >= 1 && q <= 8)));   // window column 4 q - 3 + i in 0..33
        if (wok) {
          *reinterpret_cast<u32x4 *>(dst + i * CFG::PSTR) = (u32x4){hh[0], hh[1], hh[2], hh[3]};
          *reinterpret_cast<u32x4 *>(dst + CFG::PART + i * CFG::PSTR) = (u32x4){ll[0], ll[1], ll[2], ll[3]};
        }
      }
    };

    // step s (the consumers work on block s): block s + 4 is FETCHED first (its loads start the step: the memory pipe must not idle while
    // the split runs), then block s + 2's maximum is published (loads of the previous step), then block s + 1 is split into its ring slot
    // (maximum of the previous step).  Steps are identical whatever blocks exist.
    int stepno = 0;
    auto stamp = [&](int slot) {
      if constexpr ((ABL & 32) != 0) {
        if (wg == 0 && pw == 0 && stepno < 64) {
          const long long t = __builtin_amdgcn_s_memtime();
          if (lane == 0) a.dbg[stepno * 8 + slot] = t;
        }
      }
    };
    fetch(IntC<0>{});
    fetch(IntC<1>{});
    fetch(IntC<2>{});
    fetch(IntC<3>{});
    tile_max(IntC<0>{}, 0);
    __syncthreads();                                   // (P0) weights in LDS, maxima of block 0
    tile_max(IntC<1>{}, 1);
    split_store(IntC<0>{}, 0, 0);
    __syncthreads();                                   // (P1) ring slot 0, maxima of block 1
    int s3 = 1;                                        // (s + 1) % 3
    auto step = [&](auto r4, int s) {                  // R = s % 4
      constexpr int R = decltype(r4)::value;
      stamp(0);
      if constexpr ((ABL & 32) != 0) {
        if ((wg == 0 || wg == 9 || wg == 100 || wg == 255) && pw == 0 && stepno < 2048) {
          const long long t = __builtin_amdgcn_s_memtime();
          if (lane == 0) a.dbg[512 + (wg == 0 ? 0 : (wg == 9 ? 1 : (wg == 100 ? 2 : 3))) * 2048 + stepno] = t;
        }
      }
      // the three phases touch different register sets and tables: any order is valid.  Waves 4 / 5 fetch first, waves 6 / 7 last, so
      // that the CU's address unit sees half of the loads at either end of the step instead of all 32 at once
      if ((ABL & 1024) != 0 || pw < 2) {
        fetch(IntC<R>{});                              // block s + 4 (set R: block s was split in step s - 1)
        stamp(1);
        tile_max(IntC<(R + 2) % 4>{}, s & 1);          // block s + 2 (its loads are two steps old)
        stamp(2);
        split_store(IntC<(R + 1) % 4>{}, (s + 1) & 1, s3);   // block s + 1
        stamp(3);
      } else {
        split_store(IntC<(R + 1) % 4>{}, (s + 1) & 1, s3);
        tile_max(IntC<(R + 2) % 4>{}, s & 1);
        fetch(IntC<R>{});
      }
      __syncthreads();
      ++stepno;
      s3 = s3 == 2 ? 0 : s3 + 1;
    };
    for (int s = 0; s < NB; s += 4) {
      step(IntC<0>{}, s);
      if (s + 1 < NB) step(IntC<1>{}, s + 1);
      if (s + 2 < NB) step(IntC<2>{}, s + 2);
      if (s + 3 < NB) step(IntC<3>{}, s + 3);
    }
  } else {
    // =================================================================== consumer ===================================================
    const int li = lane & 15, g = lane >> 4;
    int vl[CFG::SPK];                                  // lane part of the x fragment address per stage of a kernel row: (kx + pixel) * 96 + channel group * 16
#pragma unroll
    for (int sg = 0; sg < CFG::SPK; ++sg) {
      int kk = 4 * sg + g;
      kk = kk < CFG::KGY ? kk : CFG::KGY - 2 + (g & 1);   // half stage: lanes past the last k group re-read groups 16 / 17 (an odd slot distance: no bank conflict; their weight fragment is zero)
      vl[sg] = (kk / CFG::CG8 + li) * CFG::PSTR + (kk % CFG::CG8) * 16 + CFG::X_OFF;
    }
    const char *wl = lds + CFG::W_OFF + lane * 16;
    // half stage: lanes without a k group behind them (k groups 18, 19) read a 16-byte slot of zeros as their WEIGHT fragment (their x
    // fragment is the real data of k group 17 -- finite whenever the tile's legitimate operands are)
    const bool pad_lane = g >= (CFG::KGY % 4);
    int wh[3][3][2];
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
      for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int p = 0; p < 2; ++p)
          wh[ky][c][p] = pad_lane ? CFG::TAB_OFF + 48 : CFG::W_OFF + ky * CFG::WKY + (CFG::SPK - 1) * 6144 + (c * 2 + p) * 512 + (lane & 31) * 16;
    if (tid < 4) reinterpret_cast<unsigned *>(lds + CFG::TAB_OFF + 48)[tid] = 0u;   // (published by barrier P0)
    const int *wexp = reinterpret_cast<const int *>(a.wimg) + CFG::WBYTES / 4;
    int ew[3];
    float bz[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      ew[c] = wexp[c * 16 + li];
      bz[c] = a.bias ? a.bias[c * 16 + li] : 0.f;
    }
    // finished tile waiting for its stores (issued in the middle of the next step's MFMAs, when the producers' fetch burst has drained:
    // the address unit takes ~45 cycles per 1-KB store, 24 of them per tile in a row held every consumer wave for ~1250 cycles)
    // A store instruction writes eight channels x one whole 128-byte row (lane: channel li & 7 of the eight, 16-byte chunk (li >> 3) * 4 + g)
    // instead of sixteen channels x half a row: the two pixel tiles of a channel sit in ONE lane (acc[0][c], acc[1][c]), so lanes li and
    // li ^ 8 swap one of them (DPP row_ror:8).  Whole lines, nontemporal: the fetch + store skeleton of this kernel (no arithmetic at all,
    // tools/experimental/micro_fetch.hip) runs 3.97 ms per level-0 launch that way against 4.52 with half-line stores.
    f32x4 pend[6];                                     // store k = (channel tile k / 2, channel half k % 2)
    int pend_vo = 0;                                   // byte offset of (channel li & 7, row, column f0 + chunk * 4) in the batch item
    __amdgpu_buffer_rsrc_t pend_rs = __builtin_amdgcn_make_buffer_rsrc(a.y, 0, 0, 0x00020000);   // num_records 0: every store dropped
    int soff[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) soff[k] = (int)(((k / 2) * 16 + (k % 2) * 8) * TF * 4);
    const bool hi8 = (li & 8) != 0;
    const float act_lo = a.act == ACT_RELU ? 0.f : -__builtin_inff();

    auto flush_one = [&](auto kc) {
      constexpr int K = decltype(kc)::value;
      if constexpr ((ABL & 4) == 0) {
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, pend[K]), pend_rs, pend_vo, soff[K], (ABL & 512) ? 0 : 2);
      } else {
        if (pend[K].x == 1.2345e-30f) a.y[0] = pend[K].y;   // keeps the arithmetic alive in the no-store build
      }
    };
    // The finished tile's accumulators wait in `pacc`; its epilogue (scale, bias, activation, lane swap: ~50 VALU instructions per channel
    // tile) runs among the MFMAs of the NEXT tile's stages 1 / 3 / 5 and its six stores behind stages 7..12 -- off the wave's critical path.
    f32x4 pacc[2][3];
    int pend_e = 0;
    // accumulate mode (a.prev): the six lines of the other input slice's partial sums, fetched in the stores' own lane arrangement behind stage 6
    // of the tile they belong to (the previous tile's epilogue has used the registers by then) and added in front of the activation
    f32x4 pprev[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) pprev[k] = (f32x4){0.f, 0.f, 0.f, 0.f};
    auto epi_chunk = [&](auto kc) {                    // store k of the pending tile: channels (k / 2) * 16 + (k % 2) * 8 .. + 7, one whole 128-byte row each
      constexpr int K = decltype(kc)::value, c = K / 2;
      const float sc = __builtin_ldexpf(1.0f, -(pend_e + ew[c]));   // exact power of two (|exponent| well inside the float range: weights and activations are)
      f32x4 v[2];
#pragma unroll
      for (int qq = 0; qq < 2; ++qq) {
        v[qq].x = __builtin_fmaf(pacc[qq][c].x, sc, bz[c]);
        v[qq].y = __builtin_fmaf(pacc[qq][c].y, sc, bz[c]);
        v[qq].z = __builtin_fmaf(pacc[qq][c].z, sc, bz[c]);
        v[qq].w = __builtin_fmaf(pacc[qq][c].w, sc, bz[c]);
      }
      // even k: lanes li < 8 keep their own pixel tile 0, lanes li >= 8 take pixel tile 1 of lane li - 8; odd k: lanes li >= 8 keep their own
      // pixel tile 1, lanes li < 8 take pixel tile 0 of lane li + 8 (`v_mov_b32_dpp row_ror:8` written under a bank mask: one instruction per
      // register).  Either half recomputes the eight scaled values it needs: sixteen VALU instructions per store, one store's worth per stage.
      f32x4 o;
      if constexpr (K % 2 == 0) o = (f32x4){ror8m<0xC>(v[0].x, v[1].x), ror8m<0xC>(v[0].y, v[1].y), ror8m<0xC>(v[0].z, v[1].z), ror8m<0xC>(v[0].w, v[1].w)};
      else o = (f32x4){ror8m<0x3>(v[1].x, v[0].x), ror8m<0x3>(v[1].y, v[0].y), ror8m<0x3>(v[1].z, v[0].z), ror8m<0x3>(v[1].w, v[0].w)};
      if constexpr (ACC) o += pprev[K];
      // ReLU or nothing, without a branch (a branch would end the MFMA scheduling region)
      pend[K] = (f32x4){fmaxf(o.x, act_lo), fmaxf(o.y, act_lo), fmaxf(o.z, act_lo), fmaxf(o.w, act_lo)};
    };
    auto flush = [&]() {
      epi_chunk(IntC<0>{});
      epi_chunk(IntC<1>{});
      epi_chunk(IntC<2>{});
      epi_chunk(IntC<3>{});
      epi_chunk(IntC<4>{});
      epi_chunk(IntC<5>{});
      flush_one(IntC<0>{});
      flush_one(IntC<1>{});
      flush_one(IntC<2>{});
      flush_one(IntC<3>{});
      flush_one(IntC<4>{});
      flush_one(IntC<5>{});
      pend_rs = __builtin_amdgcn_make_buffer_rsrc(a.y, 0, 0, 0x00020000);
    };

    __syncthreads();                                   // (P0)
    __syncthreads();                                   // (P1)
    Conv3hWalk wk;
    conv3h_walk_init(a, wg, wk);
    int s3 = 0;                                        // s % 3
    for (int s = 0; s < NB; ++s) {
      const int tb = wk.b, j = wk.j, f0 = wk.strip * 32;
      conv3h_walk_next(a, wg, wk);
      if constexpr ((ABL & 32) != 0) {
        if (wg == 0 && wave == 0 && s < 64) {
          const long long t = __builtin_amdgcn_s_memtime();
          if (lane == 0) a.dbg[s * 8 + 4] = t;
        }
      }
      if (j >= 0) {
        // F % 32 == 0 (launcher): a strip is inside the image or outside as a whole; rows past T and padding strips get num_records 0
        const int tt = row0 + j * 4 + wave;
        const bool cur_ok = tt < a.T && f0 < a.F;
        const int cur_vo = (((li & 7) * a.T + tt) * a.F + f0 + ((li >> 3) * 4 + g) * 4) * 4;
        // rows of the tile: u = 0, 1 = the last two rows of the previous block, u = 2..5 = this block; wave r, kernel row ky reads u = r + ky
        const int sp = s3 == 0 ? 2 : s3 - 1;
        const int e_cur = __builtin_amdgcn_readfirstlane(exps[s3]), e_old = __builtin_amdgcn_readfirstlane(exps[sp]);
        int rowoff[3], eky[3];
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
          const int u = wave + ky;
          rowoff[ky] = (u < 2 ? sp * 4 + 2 + u : s3 * 4 + u - 2) * CFG::ROWB;
          eky[ky] = u < 2 ? e_old : e_cur;
        }
        f32x4 acc[2][3];
#pragma unroll
        for (int p = 0; p < 2; ++p)
#pragma unroll
          for (int c = 0; c < 3; ++c) acc[p][c] = (f32x4){0.f, 0.f, 0.f, 0.f};
        constexpr int PF = (ABL & 2048) ? 2 : 1;     // stages of fragment reads in flight ahead of the MFMAs (register sets: PF + 1); 2 measured 2 % slower (224 registers)
        f16x8 wf[PF + 1][3][2], xf[PF + 1][2][2];
        auto load_stage = [&](auto bufc, auto sc) {
          constexpr int BUF = decltype(bufc)::value, S = decltype(sc)::value;
          constexpr int KY = S / CFG::SPK, SG = S % CFG::SPK;
          constexpr bool half = SG == CFG::SPK - 1;
#pragma unroll
          for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int p = 0; p < 2; ++p) {
              if constexpr (half) wf[BUF][c][p] = *reinterpret_cast<const f16x8 *>(lds + wh[KY][c][p]);
              else wf[BUF][c][p] = *reinterpret_cast<const f16x8 *>(wl + KY * CFG::WKY + SG * 6144 + (c * 2 + p) * 1024);
            }
          const char *xs = lds + (rowoff[KY] + vl[SG]);
#pragma unroll
          for (int qq = 0; qq < 2; ++qq)
#pragma unroll
            for (int p = 0; p < 2; ++p) {
              xf[BUF][qq][p] = *reinterpret_cast<const f16x8 *>(xs + qq * 16 * CFG::PSTR + p * CFG::PART);
            }
        };
        auto mfma_stage = [&](auto bufc) {
          constexpr int BUF = decltype(bufc)::value;
          // smallest terms first; the six accumulators of a product are independent
#pragma unroll
          for (int qq = 0; qq < 2; ++qq)
#pragma unroll
            for (int c = 0; c < 3; ++c) acc[qq][c] = ASX_MFMA_F16(xf[BUF][qq][0], wf[BUF][c][1], acc[qq][c]);
#pragma unroll
          for (int qq = 0; qq < 2; ++qq)
#pragma unroll
            for (int c = 0; c < 3; ++c) acc[qq][c] = ASX_MFMA_F16(xf[BUF][qq][1], wf[BUF][c][0], acc[qq][c]);
#pragma unroll
          for (int qq = 0; qq < 2; ++qq)
#pragma unroll
            for (int c = 0; c < 3; ++c) acc[qq][c] = ASX_MFMA_F16(xf[BUF][qq][0], wf[BUF][c][0], acc[qq][c]);
        };
        // The fragments of stage s + 1 are read while the MFMAs of stage s issue: one `ds_read_b128` in front of every second MFMA
        // (scheduling groups; left alone, hipcc reads each fragment right before its first use and waits for it)
        __builtin_amdgcn_sched_barrier(0);
        load_stage(IntC<0>{}, IntC<0>{});
        if constexpr (PF == 2) load_stage(IntC<1>{}, IntC<1>{});
        __builtin_amdgcn_sched_barrier(0);
        auto run = [&](auto sc, auto &&self) {
          constexpr int S = decltype(sc)::value;
          if constexpr (S + PF < CFG::NST) load_stage(IntC<(S + PF) % (PF + 1)>{}, IntC<S + PF>{});
          mfma_stage(IntC<S % (PF + 1)>{});
          constexpr bool EPI = (ABL & 128) == 0 && S >= 1 && S <= 6;
          if constexpr (EPI) epi_chunk(IntC<S - 1>{});   // the previous tile's epilogue, one store's worth per stage, among this stage's MFMAs
          if constexpr (S + PF < CFG::NST) {
#pragma unroll
            for (int i = 0; i < 9; ++i) {
              __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // one DS read
              __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);   // two MFMA
              if constexpr (EPI) __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);   // two VALU of the epilogue chunk
            }
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
          }
          __builtin_amdgcn_sched_barrier(0);
          if constexpr (ACC && S == 7) {
            __amdgpu_buffer_rsrc_t prs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.prev + (int64_t)tb * a.y_bstride), 0, cur_ok ? plane_bytes : 0u, 0x00020000);
#pragma unroll
            for (int k = 0; k < 6; ++k) pprev[k] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(prs, cur_vo, soff[k], 0));
            __builtin_amdgcn_sched_barrier(0);
          }
          if constexpr ((ABL & 128) == 0 && S >= 8 && S <= 13) {
            flush_one(IntC<S - 8>{});                  // the previous tile's stores, one per stage: behind the producers' fetch burst, never two in a row
            if constexpr (S == 13) pend_rs = __builtin_amdgcn_make_buffer_rsrc(a.y, 0, 0, 0x00020000);
            __builtin_amdgcn_sched_barrier(0);
          }
          if constexpr ((ABL & 256) == 0 && S + 1 < CFG::NST && (S + 1) % CFG::SPK == 0) {
            // next kernel row: its input row may belong to the other block -- bring the accumulators to that block's scale (exact;
            // rises are bounded by EUP, a fall flushes what is negligible against what follows)
            constexpr int KY = S / CFG::SPK;
            const int d = eky[KY + 1] - eky[KY];
            if (d != 0) {
#pragma unroll
              for (int qq = 0; qq < 2; ++qq)
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                  acc[qq][c].x = __builtin_ldexpf(acc[qq][c].x, d);
                  acc[qq][c].y = __builtin_ldexpf(acc[qq][c].y, d);
                  acc[qq][c].z = __builtin_ldexpf(acc[qq][c].z, d);
                  acc[qq][c].w = __builtin_ldexpf(acc[qq][c].w, d);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
          }
          if constexpr (S + 1 < CFG::NST) self(IntC<S + 1>{}, self);
        };
        run(IntC<0>{}, run);
        if constexpr ((ABL & 32) != 0) {
          if (wg == 0 && wave == 0 && s < 64) {
            const long long t = __builtin_amdgcn_s_memtime();
            if (lane == 0) a.dbg[s * 8 + 5] = t;
          }
        }
        // ---- the accumulators wait for the next step (epi_chunk)
#pragma unroll
        for (int qq = 0; qq < 2; ++qq)
#pragma unroll
          for (int c = 0; c < 3; ++c) pacc[qq][c] = acc[qq][c];
        pend_e = e_cur;
        pend_rs = __builtin_amdgcn_make_buffer_rsrc(a.y + (int64_t)tb * a.y_bstride, 0, cur_ok ? plane_bytes : 0u, 0x00020000);
        pend_vo = cur_vo;
        if constexpr ((ABL & 128) != 0) flush();
      } else {
        flush();                                       // first block of an item: nothing to compute yet
      }
      if constexpr ((ABL & 32) != 0) {
        if (wg == 0 && wave == 0 && s < 64) {
          const long long t = __builtin_amdgcn_s_memtime();
          if (lane == 0) a.dbg[s * 8 + 6] = t;
        }
      }
      __syncthreads();
      s3 = s3 == 2 ? 0 : s3 + 1;
    }
    flush();
  }
}

}  // namespace asx
