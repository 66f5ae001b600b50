// 3x3 / pad 1 convolution of the TFC blocks (uvr_lib_v5/modules.py:11-17, mdxnet.py:97-113) as a DIRECT implicit GEMM on the fp16
// matrix pipe with the fp16 x 3 arithmetic of kernels_gemm3.h (two-part operands, three products, fp32 accumulation) -- no Winograd
// transforms, no cross-wave exchange.  First form: 48 -> 48 channels (level 0 of UVR-MDX-NET-Inst_HQ_3, the widest planes of the net).
//
//   y[b, co, t, f] = act(bias[co] + sum_{ci, ky, kx} w[co, ci, ky, kx] x[b, ci, t + ky - 1, f + kx - 1])        [B, C, T, F] fp32, F fastest
//
// GEMM view.  M = output pixels (16 consecutive f per MFMA tile), N = output channels (3 tiles of 16), K = (tap, ci) = 9 x 48 = 432 =
// 54 groups of eight channels = 13.5 stages of `v_mfma_f32_16x16x32_f16`.  A lane's eight K values are eight consecutive input channels
// of ONE tap at ONE pixel, so the x operand is kept channels-LAST in LDS ([pixel][48 halves], 96 bytes per pixel and part).
//
// One persistent 512-thread workgroup per CU, two kinds of waves:
//   * waves 4-7 PRODUCE: the haloed 6 x 34-pixel input tile of a 4 x 32 output tile is fetched with `buffer_load_dword` (eight channel
//     planes per item, out-of-image pixels through the buffer bounds check = 0), the tile's largest |x| is reduced (DPP inside a wave,
//     a four-entry LDS table across the waves, published by the step barrier), every element is scaled by the tile's power of two
//     into fp16's range, split h + l (kernels_gemm3.h: split2h_oct) and written as two 16-byte `ds_write_b128` per (pixel, 8 channels);
//   * waves 0-3 CONSUME: wave r owns output row r of the tile (two 16-pixel MFMA tiles x three 16-channel tiles = six accumulators);
//     per stage it reads six weight fragments and four x fragments (`ds_read_b128`) and issues 18 MFMAs (w_l x_h, w_h x_l, w_h x_h).
//   The two-part weight image (432 x 48 x 2 x 2 B = 81 KB, fragment order, one exponent per OUTPUT channel) is copied into LDS once per
//   workgroup and stays for the whole launch; two x buffers (2 x 38.25 KB) alternate, ONE `s_barrier` per tile.  A tile is self
//   contained (its two halo rows are fetched again by the tile below, from L2): one exponent per tile, no accumulator rescale.
//   LDS: 82944 (W) + 2 x 39168 (x) + tables = 161.4 KB of the 160 KiB.
// Bank conflicts: a 16-pixel fragment read at the 96-byte pixel stride puts the sixteen lanes of every `ds_read_b128` service group on
// sixteen distinct 16-byte slots (pixel stride 6 slots: p and p + 8 collide, and a group holds p mod 8 all different for each k group;
// the two k groups of a service group are an odd number of slots apart) -- checked in tests/test_conv3h_layout.py.
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
  int tilesT, tilesF;        // ceil(T / 4), ceil(F / 32)
  long long *dbg;            // ABL & 32 (timeline build): [workgroup 0][step < 64][8] s_memtime stamps, then [4 workgroups][2048] step starts
};

struct Conv3hCfg {
  static constexpr int C = 48, CG8 = C / 8;                       // channels, groups of eight
  static constexpr int TH = 4, TW = 32, IH = TH + 2, IW = TW + 2;
  static constexpr int NPIX = IH * IW;                            // 204 pixels per input tile
  static constexpr int PSTR = C * 2;                              // bytes per pixel and part
  static constexpr int PART = NPIX * PSTR;                        // 19584 bytes per part
  static constexpr int XBUF = 2 * PART;                           // 39168
  static constexpr int KG = 9 * CG8;                              // 54 k groups
  static constexpr int NST = (KG + 3) / 4;                        // 14 stages, the last one half empty
  static constexpr int WFULL = (KG / 4) * 3 * 2 * 1024;           // 13 full stages
  static constexpr int WBYTES = WFULL + 3 * 2 * 512;              // + the half stage (lanes 0-31 of each fragment)
  static constexpr int W_OFF = 0, X_OFF = WBYTES, TAB_OFF = X_OFF + 2 * XBUF;
  static constexpr int LDS_BYTES = TAB_OFF + 64;                  // maxima [2][4] floats, exponents [2] ints
  static constexpr size_t IMG_U32 = WBYTES / 4 + 48;              // image size in 32-bit words
};

// host: w [48, 48, 3, 3] fp32 -> the LDS image.  Fragment (stage s, channel tile n, part p) at ((s * 3 + n) * 2 + p) * 1024 bytes (512
// in the last stage), lane l of it: output channel n * 16 + (l & 15), k group 4 s + (l >> 4) = (tap, eight input channels).
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
  for (int s = 0; s < CFG::NST; ++s) {
    const bool half = (s == CFG::NST - 1) && (CFG::KG % 4 != 0);
    const int nl = half ? 32 : 64;
    for (int n = 0; n < 3; ++n)
      for (int lane = 0; lane < nl; ++lane) {
        const int co = n * 16 + (lane & 15);
        const int kg = 4 * s + (lane >> 4);
        if (kg >= CFG::KG) continue;
        const int tap = kg / CFG::CG8, c0 = (kg % CFG::CG8) * 8;
        uint16_t hh[8], ll[8];
        for (int e = 0; e < 8; ++e) {
          const float us = ldexpf(w[((size_t)co * C + c0 + e) * 9 + tap], ex[co]);
          hh[e] = conv3h_f16_rne(us);
          ll[e] = conv3h_f16_rne(us - conv3h_f16_f(hh[e]));
        }
        const size_t fb = half ? (size_t)CFG::WFULL + (size_t)(n * 2) * 512 : ((size_t)(s * 3 + n) * 2) * 1024;   // byte offset of part 0
        const size_t pstep = half ? 512 : 1024;
        uint32_t *dh = &img[(fb + (size_t)lane * 16) / 4], *dl = &img[(fb + pstep + (size_t)lane * 16) / 4];
        for (int e = 0; e < 4; ++e) {
          dh[e] = (uint32_t)hh[2 * e] | ((uint32_t)hh[2 * e + 1] << 16);
          dl[e] = (uint32_t)ll[2 * e] | ((uint32_t)ll[2 * e + 1] << 16);
        }
      }
  }
}

// Tile walk.  XCD x (block id & 7: where the dispatcher puts the block, for speed only) takes the (b, 32-strip band) items x, x + 8, ...;
// its 32 workgroups (block id >> 3) walk adjacent strips of the band down T together, so the halo columns a strip shares with its
// neighbours and the halo rows a tile shares with the tile below are re-read from that XCD's L2.  Strips past tilesF are all-padding
// tiles (never stored): they keep the walk in step.  The grid is 256 workgroups.
struct Conv3hWalk {
  int item, trow, b, strip;
};
__device__ __forceinline__ void conv3h_walk_item(const Conv3hArgs &a, int wg, Conv3hWalk &w) {
  const int nbands = (a.tilesF + 31) >> 5;
  w.b = w.item / nbands;
  w.strip = (w.item - w.b * nbands) * 32 + (wg >> 3);
}
__device__ __forceinline__ int conv3h_ntiles(const Conv3hArgs &a, int wg) {
  const int items = a.B * ((a.tilesF + 31) >> 5), x = wg & 7;
  return x < items ? ((items - x + 7) >> 3) * a.tilesT : 0;
}
__device__ __forceinline__ void conv3h_walk_init(const Conv3hArgs &a, int wg, Conv3hWalk &w) {
  w.item = wg & 7;
  w.trow = 0;
  conv3h_walk_item(a, wg, w);
}
__device__ __forceinline__ void conv3h_walk_next(const Conv3hArgs &a, int wg, Conv3hWalk &w) {
  if (++w.trow == a.tilesT) {
    w.trow = 0;
    w.item += 8;
    conv3h_walk_item(a, wg, w);
  }
}

// ABL (measurement-only builds, results are garbage): 1 = no MFMA, 2 = no global loads, 4 = no stores, 8 = no split / LDS writes
template <int ABL = 0>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void conv3h_kernel(Conv3hArgs a) {
  using CFG = Conv3hCfg;
  extern __shared__ float lds_f[];
  char *lds = reinterpret_cast<char *>(lds_f);
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wg = blockIdx.x;
  float *maxtab = reinterpret_cast<float *>(lds + CFG::TAB_OFF);   // [2][4]
  int *exps = reinterpret_cast<int *>(lds + CFG::TAB_OFF + 32);    // [2]

  // ---- weights: global image -> LDS, once
  {
    const u32x4 *src = a.wimg;
    u32x4 *dst = reinterpret_cast<u32x4 *>(lds + CFG::W_OFF);
    for (int i = tid; i < CFG::WBYTES / 16; i += 512) dst[i] = src[i];
  }
  const int64_t TF = (int64_t)a.T * a.F;
  const int N = conv3h_ntiles(a, wg);

  if (wave >= 4) {
    // =================================================================== producer ===================================================
    // Items: (input row 0..5, 8-channel group 0..5, aligned quad of columns 0..9 = columns f0 - 4 + 4 q .. + 3 of the image; the 34-column
    // window is columns f0 - 1 .. f0 + 32: quad 0 gives its last pixel, quad 9 its first, the others all four) -- 360 items, eight
    // `buffer_load_dwordx4` each (one per channel plane), two rounds of the 256 producer lanes.  Lane order inside an 8-lane group:
    // bit 0 = quad parity (lane pairs fetch 32 contiguous bytes: a vector-memory instruction costs ~45 cycles of the CU's address
    // unit that way against ~70 with every lane on its own line, tools/experimental/micro_ta.hip), bits 1-2 = channel groups 0..3
    // (items 0..239) or channel group 4 / 5 x row parity (items 240..359): the eight lanes of a `ds_write_b128` service group land
    // on four 16-byte slots twice each (2-way: 16 LDS cycles, the instruction costs 13 anyway) instead of on one slot eight times.
    const int ptid = tid - 256, pw = wave - 4;
    constexpr int NRD = 2;
    int loff[NRD], goff[NRD], prow[NRD], pq[NRD];
#pragma unroll
    for (int r = 0; r < NRD; ++r) {
      const int it = r * 256 + ptid;
      int row, cig, q;
      if (it < 240) {
        q = (it & 1) + 2 * ((it >> 3) % 5);
        cig = (it >> 1) & 3;
        row = it / 40;
      } else {
        const int u = it - 240;
        q = (u & 1) + 2 * ((u >> 3) % 5);
        cig = 4 + ((u >> 1) & 1);
        row = ((u >> 2) & 1) + 2 * (u / 40);
      }
      const bool ok = it < 360;
      loff[r] = (row * CFG::IW + 4 * q - 3) * CFG::PSTR + cig * 16;   // pixel 0 of the quad (window column 4 q - 3: negative for quad 0, never written)
      goff[r] = (int)(((int64_t)cig * 8 * a.T + row) * a.F + 4 * q);   // floats from (channel 0, t0 - 1, f0 - 4); < 2^29 (launcher)
      prow[r] = ok ? row : (1 << 20);
      pq[r] = ok ? q : -1;
    }
    f32x4 raw[3][NRD][8];                              // three register sets: tile m lives in set m % 3 (fetched in step m - 3, its maximum taken in step m - 2, split in step m - 1)
    const unsigned plane_bytes = (unsigned)(CFG::C * TF * 4);

    Conv3hWalk wk;
    conv3h_walk_init(a, wg, wk);
    int nf = 0;                                        // tiles fetched so far
    auto fetch = [&](auto setc) {                      // the next tile of the walk (nothing past the last one: every offset out of range)
      constexpr int S = decltype(setc)::value;
      const int tb = nf < N ? wk.b : 0, f0 = wk.strip * 32;
      const int t0 = nf < N ? wk.trow * 4 : -(1 << 20);
      ++nf;
      conv3h_walk_next(a, wg, wk);
      __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.x + (int64_t)tb * a.x_bstride), 0, plane_bytes, 0x00020000);
      const int org = (t0 - 1) * a.F + (f0 - 4);
#pragma unroll
      for (int r = 0; r < NRD; ++r) {
        // F % 4 == 0 and the quad is aligned: inside the row or outside as a whole
        const bool ok = (unsigned)(t0 - 1 + prow[r]) < (unsigned)a.T && (unsigned)(f0 - 4 + 4 * pq[r]) < (unsigned)a.F && pq[r] >= 0;
        const unsigned vo = ok ? (unsigned)(goff[r] + org) * 4u : 0xfffffff0u;   // past num_records: the load returns 0
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          if constexpr ((ABL & 2) != 0) raw[S][r][j] = (f32x4){0.25f, 0.5f, 0.75f, 1.f};
          else raw[S][r][j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)vo, (int)(j * TF * 4), 0));
        }
      }
    };
    auto tile_max = [&](auto setc, int slot) {         // largest finite |x| of the tile held in set S -> maxtab[slot][pw]
      constexpr int S = decltype(setc)::value;
      float m = 0.f;
#pragma unroll
      for (int r = 0; r < NRD; ++r)
#pragma unroll
        for (int j = 0; j < 8; ++j)
          m = fmaxf(m, fmaxf(fmaxf(fabsf(raw[S][r][j].x), fabsf(raw[S][r][j].y)), fmaxf(fabsf(raw[S][r][j].z), fabsf(raw[S][r][j].w))));
      m = fminf(wave_max64(m), 3.4028234663852886e38f);
      if (lane == 0) maxtab[slot * 4 + pw] = m;
    };
    auto split_store = [&](auto setc, int slot) {      // set S -> x buffer `slot` under the tile's exponent
      constexpr int S = decltype(setc)::value;
      const f32x4 mv = *reinterpret_cast<const f32x4 *>(maxtab + slot * 4);
      const float m = fmaxf(fmaxf(mv.x, mv.y), fmaxf(mv.z, mv.w));
      const int e = f16_scale_exp(m);
      if (ptid == 0) exps[slot] = e;
      char *dst = lds + CFG::X_OFF + slot * CFG::XBUF;
#pragma unroll
      for (int r = 0; r < NRD; ++r) {
        if constexpr ((ABL & 8) != 0) continue;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          u32x4 h, l;
          split2h_oct((f32x4){raw[S][r][0][i], raw[S][r][1][i], raw[S][r][2][i], raw[S][r][3][i]},
                      (f32x4){raw[S][r][4][i], raw[S][r][5][i], raw[S][r][6][i], raw[S][r][7][i]}, e, h, l);
          const bool wok = i == 3 ? (pq[r] >= 0 && pq[r] <= 8) : (i == 0 ? pq[r] >= 1 : (pq[r] >= 1 && pq[r] <= 8));   // window column 4 q - 3 + i in 0..33
          if (wok) {
            *reinterpret_cast<u32x4 *>(dst + loff[r] + i * CFG::PSTR) = h;
            *reinterpret_cast<u32x4 *>(dst + CFG::PART + loff[r] + i * CFG::PSTR) = l;
          }
        }
      }
    };

    // step n (the consumers work on tile n): tile n + 3 is FETCHED first (its loads start the step: the memory pipe is the launch's
    // bound, 10 B/clk/CU, and must not idle while the split runs), then tile n + 2's maximum is published (loads of the previous
    // step), then tile n + 1 is split into its x buffer (maximum of the previous step).  Steps are identical whatever tiles exist.
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
    tile_max(IntC<0>{}, 0);
    __syncthreads();                                   // (P0) weights in LDS, maxima of tile 0
    tile_max(IntC<1>{}, 1);
    split_store(IntC<0>{}, 0);
    __syncthreads();                                   // (P1) x buffer 0, maxima of tile 1
    auto step = [&](auto r3, int n) {                  // R = n % 3
      constexpr int R = decltype(r3)::value;
      stamp(0);
      if constexpr ((ABL & 32) != 0) {
        if ((wg == 0 || wg == 9 || wg == 100 || wg == 255) && pw == 0 && stepno < 2048) {
          const long long t = __builtin_amdgcn_s_memtime();
          if (lane == 0) a.dbg[512 + (wg == 0 ? 0 : (wg == 9 ? 1 : (wg == 100 ? 2 : 3))) * 2048 + stepno] = t;
        }
      }
      fetch(IntC<R>{});                                // tile n + 3 (set R: tile n was split in step n - 1)
      stamp(1);
      tile_max(IntC<(R + 2) % 3>{}, n & 1);            // tile n + 2
      stamp(2);
      split_store(IntC<(R + 1) % 3>{}, (n + 1) & 1);   // tile n + 1
      stamp(3);
      __syncthreads();
      ++stepno;
    };
    for (int n = 0; n < N; n += 3) {
      step(IntC<0>{}, n);
      if (n + 1 < N) step(IntC<1>{}, n + 1);
      if (n + 2 < N) step(IntC<2>{}, n + 2);
    }
  } else {
    // =================================================================== consumer ===================================================
    if constexpr ((ABL & 64) != 0) __builtin_amdgcn_s_setprio(3);
    const int li = lane & 15, g = lane >> 4;
    int xo[CFG::NST];
#pragma unroll
    for (int s = 0; s < CFG::NST; ++s) {
      int kg = 4 * s + g;
      kg = kg < CFG::KG ? kg : CFG::KG - 1;
      const int tap = kg / CFG::CG8, cig = kg % CFG::CG8;
      const int ky = tap / 3, kx = tap % 3;
      xo[s] = (ky * CFG::IW + kx) * CFG::PSTR + cig * 16;
    }
    const int xrow = (wave * CFG::IW + li) * CFG::PSTR;                 // this lane's pixel of the wave's output row, k group 0
    const char *wl = lds + CFG::W_OFF + lane * 16;
    const char *wl_half = lds + CFG::W_OFF + CFG::WFULL + (lane & 31) * 16;
    const bool pad_lane = (CFG::KG % 4 != 0) && g >= (CFG::KG % 4);     // no k group behind this lane in the last stage
    const int *wexp = reinterpret_cast<const int *>(a.wimg) + CFG::WBYTES / 4;
    int ew[3];
    float bz[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      ew[c] = wexp[c * 16 + li];
      bz[c] = a.bias ? a.bias[c * 16 + li] : 0.f;
    }
    const unsigned plane_bytes = (unsigned)(CFG::C * TF * 4);

    __syncthreads();                                   // (P0)
    __syncthreads();                                   // (P1)
    Conv3hWalk wk;
    conv3h_walk_init(a, wg, wk);
    for (int n = 0; n < N; ++n) {
      const int tb = wk.b, t0 = wk.trow * 4, f0 = wk.strip * 32;
      conv3h_walk_next(a, wg, wk);
      if constexpr ((ABL & 32) != 0) {
        if (wg == 0 && wave == 0 && n < 64) {
          const long long t = __builtin_amdgcn_s_memtime();
          if (lane == 0) a.dbg[n * 8 + 4] = t;
        }
      }
      const int xb = CFG::X_OFF + (n & 1) * CFG::XBUF + xrow;
      const char *xs_[CFG::NST];                        // fragment addresses of the stages (k group of this lane), this tile's buffer
#pragma unroll
      for (int s = 0; s < CFG::NST; ++s) xs_[s] = lds + xb + xo[s];
      __builtin_amdgcn_sched_barrier(0);
      f32x4 acc[2][3];
#pragma unroll
      for (int p = 0; p < 2; ++p)
#pragma unroll
        for (int c = 0; c < 3; ++c) acc[p][c] = (f32x4){0.f, 0.f, 0.f, 0.f};
      f16x8 wf[2][3][2], xf[2][2][2];
      auto load_stage = [&](auto bufc, auto sc) {
        constexpr int BUF = decltype(bufc)::value, S = decltype(sc)::value;
        constexpr bool half = (S == CFG::NST - 1) && (CFG::KG % 4 != 0);
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
          for (int p = 0; p < 2; ++p) {
            if constexpr (half) wf[BUF][c][p] = *reinterpret_cast<const f16x8 *>(wl_half + (c * 2 + p) * 512);
            else wf[BUF][c][p] = *reinterpret_cast<const f16x8 *>(wl + ((S * 3 + c) * 2 + p) * 1024);
          }
        const char *xs = xs_[S];
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
          for (int p = 0; p < 2; ++p) {
            f16x8 v = *reinterpret_cast<const f16x8 *>(xs + q * 16 * CFG::PSTR + p * CFG::PART);
            if constexpr (half) {
              if (pad_lane) v = (f16x8){0, 0, 0, 0, 0, 0, 0, 0};
            }
            xf[BUF][q][p] = v;
          }
      };
      auto mfma_stage = [&](auto bufc) {
        constexpr int BUF = decltype(bufc)::value;
        if constexpr ((ABL & 1) != 0) {
#pragma unroll
          for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int c = 0; c < 3; ++c) acc[q][c].x += (float)xf[BUF][q][0][0] + (float)xf[BUF][q][1][1] + (float)wf[BUF][c][0][2] + (float)wf[BUF][c][1][3];
        } else {
          // smallest terms first; the six accumulators of a product are independent
#pragma unroll
          for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int c = 0; c < 3; ++c) acc[q][c] = ASX_MFMA_F16(xf[BUF][q][0], wf[BUF][c][1], acc[q][c]);
#pragma unroll
          for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int c = 0; c < 3; ++c) acc[q][c] = ASX_MFMA_F16(xf[BUF][q][1], wf[BUF][c][0], acc[q][c]);
#pragma unroll
          for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int c = 0; c < 3; ++c) acc[q][c] = ASX_MFMA_F16(xf[BUF][q][0], wf[BUF][c][0], acc[q][c]);
        }
      };
      // The fragments of stage s + 1 are read while the MFMAs of stage s issue: one `ds_read_b128` in front of every second MFMA
      // (scheduling groups; left alone, hipcc reads each fragment right before its first use and waits for it)
      load_stage(IntC<0>{}, IntC<0>{});
      __builtin_amdgcn_sched_barrier(0);
      auto run = [&](auto sc, auto &&self) {
        constexpr int S = decltype(sc)::value;
        if constexpr (S + 1 < CFG::NST) load_stage(IntC<(S + 1) & 1>{}, IntC<S + 1>{});
        mfma_stage(IntC<S & 1>{});
        if constexpr ((ABL & 1) == 0) {
          if constexpr (S + 1 < CFG::NST) {
#pragma unroll
            for (int i = 0; i < 9; ++i) {
              __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // one DS read
              __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);   // two MFMA
            }
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (S + 1 < CFG::NST) self(IntC<S + 1>{}, self);
      };
      run(IntC<0>{}, run);

      if constexpr ((ABL & 32) != 0) {
        if (wg == 0 && wave == 0 && n < 64) {
          const long long t = __builtin_amdgcn_s_memtime();
          if (lane == 0) a.dbg[n * 8 + 5] = t;
        }
      }
      // ---- epilogue: back to the operands' scale, bias, activation, float4 stores (lane: channel li of a tile, four pixels)
      const int ex = exps[n & 1];
      const int tt = t0 + wave;
      if constexpr ((ABL & 4) == 0) {
        if (tt < a.T) {
          __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(a.y + (int64_t)tb * a.y_bstride, 0, plane_bytes, 0x00020000);
#pragma unroll
          for (int c = 0; c < 3; ++c) {
#pragma unroll
            for (int q = 0; q < 2; ++q) {
              f32x4 v = acc[q][c];
              ldexp4_inplace(v, -(ex + ew[c]));
              v += bz[c];
              if (a.act == ACT_RELU) {
                v.x = fmaxf(v.x, 0.f);
                v.y = fmaxf(v.y, 0.f);
                v.z = fmaxf(v.z, 0.f);
                v.w = fmaxf(v.w, 0.f);
              }
              const int f = f0 + q * 16 + g * 4;
              const unsigned vo = (f < a.F) ? (unsigned)((((int64_t)(c * 16 + li) * a.T + tt) * a.F + f) * 4) : 0xfffffff0u;   // F % 4 == 0: in or out as a whole
              __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rs, (int)vo, 0, 0);
            }
          }
        }
      } else {
        float chk = 0.f;
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
          for (int c = 0; c < 3; ++c) chk += acc[q][c].x + acc[q][c].y + acc[q][c].z + acc[q][c].w;
        if (chk == 1.2345e-30f) a.y[0] = chk + (float)ex;
      }
      if constexpr ((ABL & 32) != 0) {
        if (wg == 0 && wave == 0 && n < 64) {
          const long long t = __builtin_amdgcn_s_memtime();
          if (lane == 0) a.dbg[n * 8 + 6] = t;
        }
      }
      __syncthreads();
    }
  }
}

}  // namespace asx
