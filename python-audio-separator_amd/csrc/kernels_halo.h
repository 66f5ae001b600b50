// Halo-tile 3x3 / k3 convolution for CHANNELS-LAST activations (Demucs decoder rewrite convs, hdemucs.py:252-330; the VR
// Conv2DBNActiv layers, vr_network/layers.py:8-40).
//
// gg_kernel (kernels_ht.h) treats a k x k conv as a row GEMM whose A rows are gathered tap by tap: every 32-float K stage
// DMAs BM x 32 activations AND the BN x 32 weight slab, i.e. each activation enters LDS nine times per N tile and the weight
// slab is re-fetched for every 128 rows -- 9.4 B/clk/CU of LDS-DMA at full MFMA rate against a ~11 B/clk path, 46 % MFMA busy.
// Here a workgroup owns a TO x TI block of 256 output pixels and 16*NREP output channels.  Per 8-channel chunk the HALOED
// input block ((TO + 2 DO) x (TI + 2 DI) pixels x 8 channels) enters LDS ONCE and serves all nine taps; only the weights
// stream, one tap row (3 taps x 8 channels x NT columns, 9 KB at NT = 96) per sub-stage, double buffered.  DMA bytes per flop
// drop 3.3x (2.9 B/clk/CU at full rate), LDS is 44 KB, so three workgroups share a CU.
//
// LDS images (both filled by global_load_lds_dwordx4, lane-linear):
//   input : 16-byte slots, slot(q, h) = 2 q + (h ^ ((q >> 3) & 1)) for pixel q of the haloed block (raster order) and channel
//           half h (channels 4h .. 4h+3 of the chunk).  Lanes 2 q', 2 q' + 1 of a DMA fetch the two halves of ONE pixel (32
//           contiguous bytes of global memory); the swap of the halves on every second group of 8 pixels makes the fragment
//           read -- lanes (li, lk) read 8 bytes of pixel q0 + li, half lk >> 1 -- cover 64 distinct banks per 32-lane group
//           for ANY q0 (ds_read_b64: bank = (a / 4) mod 64).
//   weight: [tap][lk (4 channel pairs)][NT columns][2 channels] with a plane stride of 2 NT + 32 floats, packed on the host in
//           exactly that order (hg_pack_weights), so a sub-stage is one contiguous copy.  The 16 lanes of a fragment row read
//           128 contiguous bytes and the planes of lk and lk + 1 start 32 banks apart: conflict-free both as ds_read_b64
//           (2 x 32 lanes, 64 banks) and as the ds_read2_b64 the compiler merges neighbouring fragments into (4 x 16 lanes,
//           32 banks) -- the first layout ([tap][half][NT][4]) was 2-way conflicting under ds_read2_b64:
//           SQ_LDS_BANK_CONFLICT 45 % of the LDS cycles (profiles/r03_pmc_halo.txt).
// One MFMA k-step = 4 channels: k index lk carries channel 4 (lk >> 1) + 2 (lk & 1) + kk for the kk-th of the two MFMAs fed
// by one 8-byte read -- the same permutation on both operands.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <vector>

namespace asx {

enum { HG_KC = 8 };

struct HgArgs {
  const float *x, *wp, *bias, *zeros;
  float *y;
  int O, I, ldc, NCH;           // input [B, O, I, ldc], NCH = Cin / 8 channel chunks
  int N, nbn;                   // output columns, N tiles
  int64_t x_bs, y_bs, ldy;
  int DO, DI, PO, PI;
  int til2, tilesO, tilesI;     // TI = 1 << til2, TO = 256 >> til2
  int IWt, NPIX, inb;           // haloed block width, pixels, floats per input buffer (multiple of 256)
  int mode, act;                // GG_DENSE / GG_GLU, tdf_act() enum
  int Cout;                     // GG_GLU: output channels (the GEMM's rows are value / gate fragment pairs, engine_ht.h: ht_glu_perm)
};

template <int NREP, int KO>
__global__ __launch_bounds__(256, (NREP <= 6 ? 3 : 2)) void hg_kernel(HgArgs a) {
  constexpr int KI = 3, MREP = 4, NT = 16 * NREP;
  constexpr int PS = 2 * NT + 32;            // floats per (tap, lk) weight plane
  constexpr int WSUB = (KI * 4 * PS + 255) / 256 * 256;   // floats per weight sub-stage (one tap row), whole 1-KiB DMA pieces
  constexpr int NWI = WSUB / 256;            // wave-issues per sub-stage
  extern __shared__ float lds_f[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, lk = lane >> 4;

  int lid = xcd_remap(blockIdx.x, gridDim.x);
  const int bn = lid % a.nbn;
  lid /= a.nbn;
  const int tI = lid % a.tilesI;
  lid /= a.tilesI;
  const int tO = lid % a.tilesO;
  const int b = lid / a.tilesO;
  const int TIm = (1 << a.til2) - 1;
  const int o0 = tO * (256 >> a.til2), i0 = tI << a.til2;

  const float *xb = a.x + (int64_t)b * a.x_bs;
  const float *wt = a.wp + (int64_t)bn * a.NCH * KO * WSUB;

  // per-lane source of this lane's input slots (identical for every chunk up to the channel offset)
  int sp_off[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int s = (wave + 4 * i) * 64 + lane;
    const int q = s >> 1;
    const int h = (s & 1) ^ ((q >> 3) & 1);
    const int r = q / a.IWt, c = q - r * a.IWt;
    const int o = o0 - a.PO + r, ii = i0 - a.PI + c;
    const bool ok = q < a.NPIX && (unsigned)o < (unsigned)a.O && (unsigned)ii < (unsigned)a.I;
    sp_off[i] = ok ? (o * a.I + ii) * a.ldc + h * 4 : -1;
  }
  const int nblk_in = (2 * a.NPIX + 63) >> 6;

  auto issue_in = [&](int c, int buf) {
    float *in_s = lds_f + buf * a.inb;
    const float *xc = xb + c * HG_KC;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int j = wave + 4 * i;
      if (j < nblk_in) {
        const float *src = sp_off[i] >= 0 ? xc + sp_off[i] : a.zeros;
        ASX_GLDS16(src, in_s + j * 256);
      }
    }
  };
  auto issue_w = [&](int s, int buf) {
    float *w_s = lds_f + 2 * a.inb + buf * WSUB;
    const float *ws = wt + (int64_t)s * WSUB + lane * 4;
#pragma unroll
    for (int i = 0; i < (NWI + 3) / 4; ++i) {
      const int q = wave + 4 * i;
      if (q < NWI) ASX_GLDS16(ws + q * 256, w_s + q * 256);
    }
  };

  // haloed-block pixel of (this lane's row of M tile m, tap (0, 0))
  int aq[MREP];
#pragma unroll
  for (int m = 0; m < MREP; ++m) {
    const int r = wave * 64 + m * 16 + li;
    aq[m] = (r >> a.til2) * a.IWt + (r & TIm);
  }

  f32x4 acc[NREP][MREP];
#pragma unroll
  for (int n = 0; n < NREP; ++n)
#pragma unroll
    for (int m = 0; m < MREP; ++m) acc[n][m] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int NS = a.NCH * KO;
  const int wlane = lk * PS + li * 2;   // this lane's float offset inside a tap's weight image
  const int hsel = lk >> 1, lo2 = (lk & 1) * 2;
  issue_in(0, 0);
  issue_w(0, 0);
  int c = 0, ky = 0;
  for (int s = 0; s < NS; ++s) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (s + 1 < NS) issue_w(s + 1, (s + 1) & 1);
    if (ky == 0 && c + 1 < a.NCH) issue_in(c + 1, (c + 1) & 1);
    const float *in_s = lds_f + (c & 1) * a.inb;
    const float *w_s = lds_f + 2 * a.inb + (s & 1) * WSUB;
#pragma unroll
    for (int kx = 0; kx < KI; ++kx) {
      const int toff = ky * a.DO * a.IWt + kx * a.DI;
      f32x2 wb[NREP];
#pragma unroll
      for (int n = 0; n < NREP; ++n) wb[n] = *reinterpret_cast<const f32x2 *>(&w_s[kx * (4 * PS) + n * 32 + wlane]);
      f32x2 xa[MREP];
#pragma unroll
      for (int m = 0; m < MREP; ++m) {
        const int q = aq[m] + toff;
        const int slot = q * 2 + (hsel ^ ((q >> 3) & 1));
        xa[m] = *reinterpret_cast<const f32x2 *>(&in_s[slot * 4 + lo2]);
      }
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int n = 0; n < NREP; ++n)
#pragma unroll
          for (int m = 0; m < MREP; ++m) acc[n][m] = ASX_MFMA(wb[n][kk], xa[m][kk], acc[n][m]);
    }
    if (++ky == KO) {
      ky = 0;
      ++c;
    }
  }

  // epilogue: lane (li, lk) holds columns n0 + n*16 + lk*4 .. +3 of pixel row wave*64 + m*16 + li
  const int n0 = bn * NT;
  float *yb = a.y + (int64_t)b * a.y_bs;
#pragma unroll
  for (int m = 0; m < MREP; ++m) {
    const int r = wave * 64 + m * 16 + li;
    const int o = o0 + (r >> a.til2), ii = i0 + (r & TIm);
    if (o >= a.O || ii >= a.I) continue;
    float *yr = yb + ((int64_t)o * a.I + ii) * a.ldy;
#pragma unroll
    for (int n = 0; n < NREP; ++n) {
      const int col = n0 + n * 16 + lk * 4;
      if (col >= a.N) continue;
      f32x4 v = acc[n][m];
      if (a.bias != nullptr) v += *reinterpret_cast<const f32x4 *>(a.bias + col);
      if (a.mode == GG_GLU) {
        if (n & 1) continue;                       // gate fragment: consumed with its value fragment
        const int oc = ((n0 + n * 16) >> 1) + lk * 4;
        if (oc >= a.Cout) continue;
        f32x4 g4 = acc[n + 1 < NREP ? n + 1 : n][m];
        if (a.bias != nullptr) g4 += *reinterpret_cast<const f32x4 *>(a.bias + col + 16);
        f32x4 o4;
        o4.x = v.x * fast_sigmoid(g4.x);
        o4.y = v.y * fast_sigmoid(g4.y);
        o4.z = v.z * fast_sigmoid(g4.z);
        o4.w = v.w * fast_sigmoid(g4.w);
        *reinterpret_cast<f32x4 *>(yr + oc) = o4;
      } else {
        f32x4 o4;
        o4.x = tdf_act(v.x, a.act);
        o4.y = tdf_act(v.y, a.act);
        o4.z = tdf_act(v.z, a.act);
        o4.w = tdf_act(v.w, a.act);
        *reinterpret_cast<f32x4 *>(yr + col) = o4;
      }
    }
  }
}

// N tile the halo kernel uses for n output columns (fewest padded columns; wider on ties)
static inline int hg_tile_n(int n) {
  static const int force = getenv("ASX_HALO_NT") ? atoi(getenv("ASX_HALO_NT")) : 0;
  if (force == 32 || force == 64 || force == 96 || force == 128) return force;
  if (n <= 32) return 32;
  if (n <= 64) return 64;
  static const bool split128 = getenv("ASX_HALO_SPLIT128") && atoi(getenv("ASX_HALO_SPLIT128")) != 0;   // A/B: 64-column tiles (3 workgroups per CU) where 128 would fit
  if (split128 && n % 128 == 0) return 64;
  const int p96 = (n + 95) / 96 * 96, p128 = (n + 127) / 128 * 128;
  return p96 <= p128 ? 96 : 128;
}

// floats of one weight sub-stage (a tap row of 3 taps) in the packed image
static inline int hg_wsub(int NT) { return (3 * 4 * (2 * NT + 32) + 255) / 256 * 256; }

// [N, K = tap*Cin + ci] row-major (the gg_kernel layout) -> [N tile][chunk][tap row][tap][lk][NT (+16 pad)][2], every tap row
// padded to whole 1-KiB pieces; lk carries channels 2 lk, 2 lk + 1 of the chunk
static inline void hg_pack_weights(const std::vector<float> &w, int N, int taps, int cin, int NT, std::vector<float> &out) {
  const int nbn = (N + NT - 1) / NT, nch = cin / HG_KC, K = taps * cin, KO = taps / 3, PS = 2 * NT + 32, WSUB = hg_wsub(NT);
  out.assign((size_t)nbn * nch * KO * WSUB, 0.f);
  for (int t = 0; t < nbn; ++t)
    for (int c = 0; c < nch; ++c)
      for (int ky = 0; ky < KO; ++ky) {
        float *sub = out.data() + ((size_t)(t * nch + c) * KO + ky) * WSUB;
        for (int kx = 0; kx < 3; ++kx)
          for (int lk = 0; lk < 4; ++lk)
            for (int n = 0; n < NT; ++n)
              for (int j = 0; j < 2; ++j) {
                const int row = t * NT + n;
                if (row < N) sub[(kx * 4 + lk) * PS + n * 2 + j] = w[(size_t)row * K + (size_t)(ky * 3 + kx) * cin + c * HG_KC + lk * 2 + j];
              }
      }
}

struct HgGeom {
  int til2, tilesO, tilesI, IWt, NPIX, inb;
};

// block shape for an O x I image: 1-D rows are 256 long, 2-D blocks as wide as the image allows (16-row M tiles stay inside
// one block row from TI = 16 up)
static inline bool hg_geometry(int O, int I, int KO, int DO, int DI, HgGeom *g) {
  int til2;
  if (KO == 1) til2 = 8;
  else if (I > 48) til2 = 6;
  else if (I > 16) til2 = 5;
  else if (I > 8) til2 = 4;
  else til2 = 3;
  const int TI = 1 << til2, TO = 256 >> til2;
  g->til2 = til2;
  g->tilesO = (O + TO - 1) / TO;
  g->tilesI = (I + TI - 1) / TI;
  g->IWt = TI + 2 * DI;
  g->NPIX = (TO + (KO - 1) * DO) * g->IWt;
  g->inb = ((2 * g->NPIX + 63) / 64) * 256;
  return g->NPIX <= 512;
}

template <int NREP, int KO>
static void hg_launch(const HgArgs &a, int64_t nblocks, hipStream_t s) {
  const int WSUB = hg_wsub(16 * NREP);
  const int lds = (2 * a.inb + 2 * WSUB) * 4;
  static int granted = 0;
  if (lds > granted) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&hg_kernel<NREP, KO>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    granted = lds;
  }
  hipLaunchKernelGGL((hg_kernel<NREP, KO>), dim3((unsigned)nblocks), dim3(256), lds, s, a);
}

static inline void hg_dispatch(const HgArgs &a, int NT, int KO, int B, hipStream_t s) {
  const int64_t nblocks = (int64_t)B * a.tilesO * a.tilesI * a.nbn;
  if (KO == 3) {
    switch (NT) {
      case 32: hg_launch<2, 3>(a, nblocks, s); break;
      case 64: hg_launch<4, 3>(a, nblocks, s); break;
      case 96: hg_launch<6, 3>(a, nblocks, s); break;
      default: hg_launch<8, 3>(a, nblocks, s); break;
    }
  } else {
    switch (NT) {
      case 32: hg_launch<2, 1>(a, nblocks, s); break;
      case 64: hg_launch<4, 1>(a, nblocks, s); break;
      case 96: hg_launch<6, 1>(a, nblocks, s); break;
      default: hg_launch<8, 1>(a, nblocks, s); break;
    }
  }
}

}  // namespace asx
