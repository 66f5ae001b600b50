// Row GEMM  y[M,N] = act(scale * (x[M,K] . W[N,K]^T + bias) + shift) (+ res), third generation: fp32 results from the bf16
// matrix pipe.  Replaces tdf2_kernel (kernels_gemm2.h) on the TDF blocks of ConvTDFNet (uvr_lib_v5/modules.py:57-74) and on every
// nn.Linear of the sibling nets whenever the launcher's preconditions hold (asx.hip: tdf3_ok).
//
// Arithmetic.  gfx950 has no TF32 and its fp32 MFMA runs at 1/16 of the bf16 rate.  Every fp32 operand is therefore written as an
// EXACT sum of three bf16 numbers, v = h + m + l (h = RNE_bf16(v), m = RNE_bf16(v - h), l = v - h - m: 8 + 8 + 8 significand bits,
// the two subtractions are exact in fp32 and l is representable, so nothing is lost), and a product of two operands as
//     w x  =  wh xh + (wh xm + wm xh) + (wm xm + wh xl + wl xh)   +   [wm xl + wl xm + wl xl  <= 2^-24 |w x|, dropped]
// -- six bf16 MFMAs (`v_mfma_f32_16x16x32_bf16`, products exact, fp32 accumulation) instead of eight fp32 MFMAs per 32-deep k
// step, at 1/16 of the cycles each: 2.67x the fp32 matrix peak for a per-product error of one fp32 rounding.  The dropped terms
// are of the size of the rounding an fp32 FMA chain commits on every product anyway; measured against a float64 GEMM the kernel
// is as close as the fp32-MFMA kernel it replaces (tests/test_gpu_parity.py::test_rowgemm_bf16x6_*).
//
// Dataflow per workgroup (4 waves, tile BM = 16 MREP rows x BN = 64 NREP columns, k step 32):
//   * W is split ONCE per weight tensor (w3_split_kernel) into an image in MFMA-fragment order, [n / 16][k / 32][part][lane][8 bf16]:
//     a wave reads the fragments of ITS 16 NREP columns straight from L2 into registers with one fully coalesced 1-KiB
//     `global_load_dwordx4` per fragment, a stage ahead -- no LDS space, no LDS reads and no barrier for the weight operand;
//   * x arrives as fp32 rows (one 128-byte line per row and stage), a stage ahead in registers; each element is split once
//     per workgroup (5.5 VALU instructions) and written to LDS as three bf16 images [part][row][4 chunks of 8 k], the chunk index
//     XOR-ed with h((row >> 2) & 3), h = {0, 2, 3, 1}, so that both the 16-byte writes and the `ds_read_b128` fragment reads (four
//     non-contiguous 16-lane groups) are conflict-free;
//   * LDS holds only the x parts: 2 x 3 x BM x 64 bytes = 48 KB at BM = 128 -- two workgroups (8 waves) per CU, bounded by the
//     register file (96 accumulators + 72 for two stages of weight fragments).
// Accumulator layout, tile -> workgroup map and epilogues are those of tdf2_kernel (the epilogue code is the same arithmetic,
// expression for expression, so activation / residual / rotary rounding is identical between the two kernels).
#pragma once
#include "kernels_net.h"

namespace asx {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

#define ASX_MFMA_BF16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16((a), (b), (c), 0, 0, 0)

__device__ __forceinline__ unsigned cvt_pk_bf16(float lo, float hi) {
  unsigned r;
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
  return r;
}

// exact three-way split of two floats into packed bf16 pairs (low half = first float)
__device__ __forceinline__ void split3_pair(float x0, float x1, unsigned &h, unsigned &m, unsigned &l) {
  h = cvt_pk_bf16(x0, x1);
  float r0 = __fsub_rn(x0, __uint_as_float(h << 16));
  float r1 = __fsub_rn(x1, __uint_as_float(h & 0xffff0000u));
  m = cvt_pk_bf16(r0, r1);
  r0 = __fsub_rn(r0, __uint_as_float(m << 16));
  r1 = __fsub_rn(r1, __uint_as_float(m & 0xffff0000u));
  l = cvt_pk_bf16(r0, r1);
}

__device__ __forceinline__ void split3_oct(const f32x4 &a, const f32x4 &b, u32x4 &h, u32x4 &m, u32x4 &l) {
  unsigned hh[4], mm[4], ll[4];
  split3_pair(a.x, a.y, hh[0], mm[0], ll[0]);
  split3_pair(a.z, a.w, hh[1], mm[1], ll[1]);
  split3_pair(b.x, b.y, hh[2], mm[2], ll[2]);
  split3_pair(b.z, b.w, hh[3], mm[3], ll[3]);
  h = (u32x4){hh[0], hh[1], hh[2], hh[3]};
  m = (u32x4){mm[0], mm[1], mm[2], mm[3]};
  l = (u32x4){ll[0], ll[1], ll[2], ll[3]};
}

// ---- fp16 x 3 variant (template parameter H of tdf3_kernel; asx_set_option "gemm_f16x3") ----------------------------------------
// The same idea on the fp16 matrix pipe (same rate as bf16 on gfx950), with HALF the MFMAs: an operand scaled by a power of two into
// fp16's range is written as v 2^e = h + l, h = RNE_f16(v 2^e), l = RNE_f16(v 2^e - h) (the subtraction is exact): 11 + 11 significand
// bits, |v 2^e - h - l| <= 2^-23 |v 2^e| while l stays above fp16's subnormal spacing 2^-24 -- i.e. for every element within 2^-14 of
// its block's largest (the block maximum sits in [2^12, 2^15)); smaller elements keep an ABSOLUTE error of 2^-25 in scaled units, below
// 2^-37 of the block maximum.  A product is
//     w x  =  wh xh + (wh xl + wl xh)   +   [wl xl <= 2^-22 |w x|, dropped]
// -- three MFMAs (`v_mfma_f32_16x16x32_f16`) per 32-deep k step instead of six.  Operand error 2^-23 and a dropped term of 2^-22 against
// the bf16 x 6 form's 2^-24, but three accumulator roundings per k step instead of six: against a float64 GEMM the results measured
// CLOSER than the bf16 x 6 kernel's on every shape of tools/proto_gemm3.hip (6.3e-7 against 8.5e-7 at K = 3072), both closer than the
// fp32-MFMA kernel (tests/test_gpu_parity.py::test_rowgemm_bf16x6_vs_float64[f16x3], ::test_rowgemm_f16x3_block_exponent).
// Scaling.  W: one exponent per group of FOUR output columns (the four a lane of the accumulator layout owns: free to apply), chosen by
// the image builder (largest |w| of the group in [2^14, 2^15)), stored behind the fragments.  x: one RUNNING exponent PER ROW, found on line: the four lanes that fetch a row's 32 floats of a stage reduce
// their largest |x| (two DPP steps) right before the split; when it would pass 2^15 under the row's exponent, the exponent drops (two
// bits of headroom) and the drop goes through a small LDS table to the MFMA side, which multiplies that row's accumulators by the exact
// power of two before the next stage's products arrive -- rare after a row's first stages, free when nothing changed (x 1.0).  The
// epilogue multiplies every accumulator by 2^-(e_x[row] + e_w[column group]) (exact) and continues as the bf16 x 6 kernel does.  A row never
// sees another row's magnitude: an Inf / NaN row poisons itself only, rows 2^100 apart in one tile keep their own precision.
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

#define ASX_MFMA_F16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_f16((a), (b), (c), 0, 0, 0)

__device__ __forceinline__ unsigned cvt_pk_f16(float lo, float hi) {
  unsigned r;
  asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
  return r;
}

// two floats, scaled by 2^e, as packed fp16 pairs h + l (low half = first float)
__device__ __forceinline__ void split2h_pair(float x0, float x1, int e, unsigned &h, unsigned &l) {
  x0 = __builtin_ldexpf(x0, e);
  x1 = __builtin_ldexpf(x1, e);
  h = cvt_pk_f16(x0, x1);
  const f16x2 hv = __builtin_bit_cast(f16x2, h);
  l = cvt_pk_f16(__fsub_rn(x0, (float)hv[0]), __fsub_rn(x1, (float)hv[1]));
}

__device__ __forceinline__ void split2h_oct(const f32x4 &a, const f32x4 &b, int e, u32x4 &h, u32x4 &l) {
  unsigned hh[4], ll[4];
  split2h_pair(a.x, a.y, e, hh[0], ll[0]);
  split2h_pair(a.z, a.w, e, hh[1], ll[1]);
  split2h_pair(b.x, b.y, e, hh[2], ll[2]);
  split2h_pair(b.z, b.w, e, hh[3], ll[3]);
  h = (u32x4){hh[0], hh[1], hh[2], hh[3]};
  l = (u32x4){ll[0], ll[1], ll[2], ll[3]};
}

// largest finite |.| of eight floats (NaN never wins a v_max; Inf is masked by the caller's slow path)
__device__ __forceinline__ float absmax_oct(const f32x4 &a, const f32x4 &b) {
  return fmaxf(fmaxf(fmaxf(fabsf(a.x), fabsf(a.y)), fmaxf(fabsf(a.z), fabsf(a.w))),
               fmaxf(fmaxf(fabsf(b.x), fabsf(b.y)), fmaxf(fabsf(b.z), fabsf(b.w))));
}
__device__ __forceinline__ float finite_or_zero(float v) {
  v = fabsf(v);
  return v <= 3.4028234663852886e38f ? v : 0.f;
}
__device__ __forceinline__ float absmax_oct_finite(const f32x4 &a, const f32x4 &b) {
  return fmaxf(fmaxf(fmaxf(finite_or_zero(a.x), finite_or_zero(a.y)), fmaxf(finite_or_zero(a.z), finite_or_zero(a.w))),
               fmaxf(fmaxf(finite_or_zero(b.x), finite_or_zero(b.y)), fmaxf(finite_or_zero(b.z), finite_or_zero(b.w))));
}

// maximum over the 64 lanes of a wave, returned in every lane (DPP: four row shifts, two row broadcasts, one readlane)
__device__ __forceinline__ float wave_max64(float v) {
  int b = __float_as_int(v);
#define ASX_WMAX_STEP(ctrl, rm)                                                          \
  {                                                                                      \
    const int t = __builtin_amdgcn_update_dpp(b, b, (ctrl), (rm), 0xf, false);           \
    b = __float_as_int(fmaxf(__int_as_float(b), __int_as_float(t)));                     \
  }
  ASX_WMAX_STEP(0x111, 0xf)   // row_shr:1
  ASX_WMAX_STEP(0x112, 0xf)   // row_shr:2
  ASX_WMAX_STEP(0x114, 0xf)   // row_shr:4
  ASX_WMAX_STEP(0x118, 0xf)   // row_shr:8   -> lane 15 of every row holds the row's maximum
  ASX_WMAX_STEP(0x142, 0xa)   // row_bcast:15 into rows 1 and 3
  ASX_WMAX_STEP(0x143, 0xc)   // row_bcast:31 into rows 2 and 3 -> lane 63 holds the wave's maximum
#undef ASX_WMAX_STEP
  return __int_as_float(__builtin_amdgcn_readlane(b, 63));
}

// v *= 2^e in place (tied operands; the epilogue's way back to the operands' own scale)
__device__ __forceinline__ void ldexp4_inplace(f32x4 &v, int e) {
#if defined(ASX_LDEXP_ASM)          // the form of round 5 (A/B builds of tools/proto_gemm3.hip only): hipcc pads no MFMA -> VALU wait states in front of an asm statement
  asm volatile("v_ldexp_f32 %0, %0, %4\n\tv_ldexp_f32 %1, %1, %4\n\tv_ldexp_f32 %2, %2, %4\n\tv_ldexp_f32 %3, %3, %4"
               : "+v"(v.x), "+v"(v.y), "+v"(v.z), "+v"(v.w)
               : "v"(e));
#else
  v.x = __builtin_ldexpf(v.x, e);
  v.y = __builtin_ldexpf(v.y, e);
  v.z = __builtin_ldexpf(v.z, e);
  v.w = __builtin_ldexpf(v.w, e);
#endif
}

// exponent e with m 2^e in [2^14, 2^15) (m > 0 finite); m == 0 gives 15
__device__ __forceinline__ int f16_scale_exp(float m) { return 15 - __builtin_amdgcn_frexp_expf(m); }
// running block exponents (tdf3_kernel rows, conv_wino6_kernel tile rows): rise when the block's largest element is this many bits below the
// range; never more than RISE_CAP bits above the lowest exponent the accumulators have carried
constexpr int F16X3_RISE = 10, F16X3_RISE_CAP = 40;

// W[N, K] fp32 -> fragment-ordered bf16 x 3 image: img[((nt * nk + ks) * 3 + part) * 64 + lane] = 8 bf16 of row nt * 16 + (lane & 15),
// k = ks * 32 + (lane >> 4) * 8 .. + 7; rows >= N and stages past K (the image holds an even number of stages) are zero.  One
// thread per (nt, ks, lane).
// cin > 0 (GATHER mode with a channel count that is not a multiple of 32): K = taps * cin and a stage is (tap, 32-channel chunk) --
// the last chunk of every tap is zero padded past cin (cin % 8 == 0), `nst` = taps * ceil(cin / 32) stages exist.
// the eight fp32 weights of fragment slot (nt, ks, lane) of either image (zeros past N / K / cin)
__device__ __forceinline__ void w3_fetch(const float *__restrict__ w, int N, int K, int cin, int nst, int nt, int ks, int lane, f32x4 &a,
                                         f32x4 &b) {
  const int n = nt * 16 + (lane & 15);
  a = (f32x4){0.f, 0.f, 0.f, 0.f};
  b = a;
  if (cin > 0) {
    const int nch = (cin + 31) >> 5;
    const int tap = ks / nch, ch = ks - tap * nch;
    const int kk = ch * 32 + (lane >> 4) * 8;
    if (n < N && ks < nst && kk < cin) {
      const float *p = w + (int64_t)n * K + (int64_t)tap * cin + kk;
      a = *reinterpret_cast<const f32x4 *>(p);
      b = *reinterpret_cast<const f32x4 *>(p + 4);
    }
  } else if (n < N && ks * 32 < K) {                   // K % 32 == 0: a stage is inside K or past it as a whole
    const float *p = w + (int64_t)n * K + ks * 32 + (lane >> 4) * 8;
    a = *reinterpret_cast<const f32x4 *>(p);
    b = *reinterpret_cast<const f32x4 *>(p + 4);
  }
}

__global__ __launch_bounds__(256) void w3_split_kernel(const float *__restrict__ w, u32x4 *__restrict__ img, int N, int K, int64_t total,
                                                       int cin = 0, int nst = 0) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int nk = cin > 0 ? ((nst + 1) & ~1) : ((K + 63) >> 6) * 2;   // stages of the image: an even number (the kernel runs stage pairs), zeros past the data
  const int lane = (int)(idx & 63);
  const int64_t f = idx >> 6;
  const int ks = (int)(f % nk);
  const int nt = (int)(f / nk);
  f32x4 a, b;
  w3_fetch(w, N, K, cin, nst, nt, ks, lane, a, b);
  u32x4 h, m, l;
  split3_oct(a, b, h, m, l);
  u32x4 *o = img + (f * 3) * 64 + lane;
  o[0] = h;
  o[64] = m;
  o[128] = l;
}

// fp16 x 3 image: img[((nt * nk + ks) * 2 + part) * 64 + lane], same fragment order.  Every group of FOUR rows of a 16-row tile (the
// four output columns one lane of the GEMM owns: accumulator layout) is scaled by its own 2^e, the group's largest |w| in
// [2^14, 2^15): an output channel never shares an exponent with more than three neighbours (weights with a folded normalisation can
// differ by orders of magnitude from channel to channel).  The int32 exponents e[ntiles][4] follow the fragments (at img + ntiles *
// nk * 128).  One workgroup per tile; a thread's items all belong to ONE row (lane & 15 is fixed by tid).
__global__ __launch_bounds__(256) void w3h_split_kernel(const float *__restrict__ w, u32x4 *__restrict__ img, int N, int K, int ntiles,
                                                        int cin = 0, int nst = 0) {
  const int nk = cin > 0 ? ((nst + 1) & ~1) : ((K + 63) >> 6) * 2;
  const int nt = blockIdx.x, tid = threadIdx.x;
  __shared__ float rowmax[256];
  __shared__ int gexp[4];
  float m = 0.f;
  for (int it = tid; it < nk * 64; it += 256) {
    f32x4 a, b;
    w3_fetch(w, N, K, cin, nst, nt, it >> 6, it & 63, a, b);
    m = fmaxf(m, absmax_oct_finite(a, b));
  }
  rowmax[tid] = m;
  __syncthreads();
  if (tid < 4) {
    float g = 0.f;
    for (int t = 0; t < 256; ++t)
      if (((t & 15) >> 2) == tid) g = fmaxf(g, rowmax[t]);
    gexp[tid] = g > 0.f ? f16_scale_exp(g) : 0;
  }
  __syncthreads();
  const int e = gexp[(tid & 15) >> 2];
  for (int it = tid; it < nk * 64; it += 256) {
    f32x4 a, b;
    w3_fetch(w, N, K, cin, nst, nt, it >> 6, it & 63, a, b);
    u32x4 h, l;
    split2h_oct(a, b, e, h, l);
    u32x4 *o = img + ((int64_t)(nt * nk + (it >> 6)) * 2) * 64 + (it & 63);
    o[0] = h;
    o[64] = l;
  }
  if (tid < 4) reinterpret_cast<int *>(img + (int64_t)ntiles * nk * 128)[nt * 4 + tid] = gexp[tid];
}

template <int V>
struct IntC {
  static constexpr int value = V;
};

// ABL (ASX_TDF3_ABL, measurement-only instantiations, results are garbage): bit 0 = no x loads / split after the prologue,
// bit 1 = no MFMA, bit 2 = no epilogue traffic, bit 3 = no weight loads after the prologue
// GATHER mode: the same GEMM with its A rows GATHERED from a channels-last image -- a stride-1 KO x KI convolution of the halo /
// gather kernels (kernels_halo.h, kernels_ht.h) as an implicit GEMM: row r = OUTPUT pixel (b, oo, j) in raster order, stage = (tap,
// 32-channel chunk), the x fetch of a stage reads 128 contiguous bytes of input pixel (oo SO + ky DO - PO, j SI + kx DI - PI) or the
// zero page (strided convolutions included).
// Per stage the work is exactly that of the plain GEMM (split cost per MFMA unchanged); the taps' re-reads hit L2.
struct RowGather {
  int O, I, KI, DO, DI, PO, PI, nch;                  // input image, taps per row, dilations, paddings, Cin / 32
  int SO, SI, OR, IR;                                 // strides and OUTPUT image: row r = (b, oo, j), input origin (oo SO - PO, j SI - PI)
  int cin;                                            // channels per tap (nch = ceil(cin / 32); the last chunk of a tap may be partial: cin % 8 == 0)
  int ldc;                                            // floats per pixel of x
  int64_t x_bs;                                       // floats per image of x
};

// H: the fp16 x 3 arithmetic (above) instead of bf16 x 6 -- two parts per operand, three MFMAs per product, x registers double
// buffered (a stage's rows are fetched a whole stage before their maximum is needed), 12 * BM + 16 more bytes of LDS: the exponent-drop table of either
// stage parity, the rows' current exponents and, per stage parity, the last stage some row's exponent dropped at (de_tab[2][BM], ex_tab[BM], fl_tab[2]).
// rmax[BM] (producer side of a pair image: the rows' largest |y| over the tile's columns) follows them.
constexpr int tdf3h_lds_bytes(int BM) { return 2 * 2 * BM * 64 + 16 * BM + 16; }
// PS (H only; instantiated by experimental builds and tools/proto_gemm3.hip -- measured round 6: the reader alone gains 3-17 %, the nets nothing,
// profiles/NOTES.md): x is a PAIR IMAGE (kernels_net.h, TdfDmaArgs::xexp) -- split once by the kernel that produced it instead of by every column
// tile of this GEMM: a chunk's two 16-byte loads are regrouped into its h and l parts (no arithmetic), brought from their span's exponent
// to the row's (the smallest of the row's spans: eight v_pk_mul_f16 by a power of two) and stored; the row's exponent never changes, so
// the accumulators are never rescaled.  Any H build WRITES a pair image when TdfDmaArgs::yexp is set (epilogue).
// NW: waves per workgroup, 4 or 8.  Eight waves share ONE x tile between two 4-wave column halves (tile 128 x 32 NREP NW columns: 128 x 384 for
// NREP = 3) -- half the x loads, splits and LDS stores per MFMA of the 4-wave form, one workgroup per CU instead of two.
template <int NREP, int MREP, int ABL = 0, bool GATHER = false, bool H = false, bool PS = false, int NW = 4>
__global__ __launch_bounds__(64 * NW, NW == 4 ? 2 : 1) void tdf3_kernel(TdfDmaArgs a, const u32x4 *__restrict__ w3, RowGather gq) {
  static_assert(!PS || (H && !GATHER), "pair-image operands: fp16 x 3 arithmetic, plain row GEMM");
  static_assert(NW == 4 || (NW == 8 && MREP == 8 && !GATHER), "eight waves: 128-row tiles of the plain row GEMM");
  constexpr int NT = 64 * NW;                         // threads
  constexpr int BM = 16 * MREP, BN = 16 * NREP * NW;
  constexpr int XC = BM * 4 / NT;                     // 8-float chunks per thread and stage (BM * 4 chunks / NT threads)
  constexpr int NP = H ? 2 : 3;                       // parts per operand
  constexpr int PART = BM * 64;                       // bytes of one part image
  constexpr int BUFB = NP * PART;                     // bytes of one stage buffer
  static_assert(MREP % 4 == 0 && XC >= 1, "tile shape");
  // H builds: ABL bits 0 (no x loads / split after the prologue), 2 (no epilogue traffic), 3 (no weight loads after the prologue) and
  // bit 4 = 16 (no per-row accumulator rescale) -- tools/proto_gemm3.hip
  extern __shared__ float lds_f[];
  char *lds = reinterpret_cast<char *>(lds_f);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, lk = lane >> 4;

  // ---- tile -> workgroup map (kernels_gemm2.h: column tiles partitioned over the XCDs so that an XCD keeps its slice of W hot)
  const int nbn = (a.N + BN - 1) / BN;
  int bg;
  int64_t bm_i;
  {
    const int nbm_i = (int)((a.M + BM - 1) / BM);
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    if (a.tile_map == 2 && (nbn & 7) == 0 && (nbm_i & 7) == 0 && (((nbm_i >> 3) * (nbn >> 3)) & 7) == 0) {
      // 8 x 8 super-tiles, dealt round-robin over the XCDs: the 64 workgroup slots of an XCD hold one super-tile, whose 8 row blocks
      // of x and 8 column tiles of W are each fetched once per XCD
      const int sid = (slot >> 6) * 8 + xcd, r = slot & 63, sc = nbn >> 3;
      bg = (sid % sc) * 8 + (r & 7);
      bm_i = (int64_t)(sid / sc) * 8 + (r >> 3);
    } else if (a.tile_map == 1) {
      const int lid = xcd_remap(blockIdx.x, gridDim.x);
      bg = lid % nbn;
      bm_i = lid / nbn;
    } else if ((gridDim.x & 7) == 0 && nbn >= 8 && (nbn & 7) == 0) {
      const int cx = nbn >> 3;
      bg = xcd * cx + slot % cx;
      bm_i = slot / cx;
    } else if ((gridDim.x & 7) == 0 && nbn < 8 && (8 % nbn) == 0 && nbm_i % (8 / nbn) == 0) {
      const int r = 8 / nbn;
      bg = xcd / r;
      bm_i = (int64_t)slot * r + (xcd % r);
    } else {
      const int lid = xcd_remap(blockIdx.x, gridDim.x);
      bg = lid % nbn;
      bm_i = lid / nbn;
    }
  }
  const int64_t m0 = bm_i * BM;
  const int n0 = bg * BN;
  const int64_t lda = a.lda ? a.lda : a.K, ldy = a.ldy ? a.ldy : a.N, ldr = a.ldr ? a.ldr : a.N;
  // K % 32 == 0 (launcher).  The stage loop runs PAIRS of stages: for an odd stage count the image carries one more stage of zero
  // weights, and the x loads of that stage re-read the last real one (finite data times zero).
  int nkx = a.K >> 5;                                  // stages that exist in x
  if constexpr (GATHER) nkx = (a.K / gq.cin) * gq.nch; // taps x chunks (a partial last chunk per tap when cin % 32 != 0)
  const int nk = (nkx + 1) & ~1;

  // ---- weight fragments: wave-uniform base per 16-column tile + lane * 16 bytes; tiles past N are clamped (masked at the store)
  const int ntiles = (a.N + 15) >> 4;
  const u32x4 *wb[NREP];
#pragma unroll
  for (int n = 0; n < NREP; ++n) {
    int nt = (n0 >> 4) + wave * NREP + n;
    nt = nt < ntiles ? nt : ntiles - 1;
    wb[n] = w3 + (int64_t)nt * nk * (64 * NP) + lane;
  }
  u32x4 wr[2][NREP][NP];
  auto load_w = [&](auto par, int ks) {
    constexpr int P = decltype(par)::value;
#pragma unroll
    for (int n = 0; n < NREP; ++n)
#pragma unroll
      for (int p = 0; p < NP; ++p) wr[P][n][p] = wb[n][(int64_t)ks * (64 * NP) + p * 64];
  };

  // chunk swizzle of the x part images: chunk ^ h((row >> 2) & 3), h = {0, 2, 3, 1}.  `ds_read_b128` is serviced in four
  // NON-contiguous 16-lane groups ({0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, ...: MI355X_MICROARCH.md, LDS): with the plain
  // XOR (h = identity) lanes (li 0-3, lk 0) and (li 4-7, lk 1) of a group met on the same banks -- 40 % of the LDS cycles were
  // conflict cycles in the first PMC pass; this permutation puts the sixteen lanes of every group on sixteen distinct slots.
  auto hsw = [](int g) { return g == 0 ? 0 : (g == 1 ? 2 : (g == 2 ? 3 : 1)); };
  // ---- x rows: chunk q = tid + NT i -> row q >> 2 of the tile, floats (q & 3) * 8 .. + 7 of the stage
  const float *xp[XC];
  int xw[XC];                                          // LDS byte offset of the chunk inside a part image
  int po[XC], pi[XC];                                  // GATHER: the row's pixel (o, i); o = -2^20 for rows past M (never valid)
#pragma unroll
  for (int i = 0; i < XC; ++i) {
    const int q = tid + NT * i;
    const int row = q >> 2, c = q & 3;
    int64_t r = m0 + row;
    if constexpr (GATHER) {
      const bool rok = r < a.M;
      r = rok ? r : a.M - 1;
      const int64_t img = (int64_t)gq.OR * gq.IR;
      const int64_t b = r / img;
      const int rem = (int)(r - b * img);
      const int oo = rem / gq.IR, jj = rem - oo * gq.IR;
      po[i] = oo * gq.SO;
      pi[i] = jj * gq.SI;
      xp[i] = a.x + b * gq.x_bs + ((int64_t)po[i] * gq.I + pi[i]) * gq.ldc + c * 8;
      if (!rok) po[i] = -(1 << 20);
    } else {
      r = r < a.M ? r : a.M - 1;
      xp[i] = a.x + r * lda + c * 8;
      po[i] = pi[i] = 0;
    }
    xw[i] = row * 64 + ((c ^ hsw((row >> 2) & 3)) << 4);
  }
  f32x4 xr[XC][2];
  const int *xe[XC];                                   // PS: the exponent spans of this thread's rows
  int ej[XC];                                          // PS: span exponent of the chunk held in xr
  auto load_x_sel = [&](int ks, int only) {            // only < 0: every chunk of the stage; else chunk `only` (H mode: a chunk's registers are refilled right after its split)
    const int kcs = ks < nkx ? ks : nkx - 1;           // wave-uniform clamp (see nk above)
    if constexpr (GATHER) {
      const int tap = kcs / gq.nch, ch = kcs - tap * gq.nch;
      const int ky = tap / gq.KI, kx = tap - ky * gq.KI;
      const int dy = ky * gq.DO - gq.PO, dx = kx * gq.DI - gq.PI;
      const int off = (dy * gq.I + dx) * gq.ldc + ch * 32;   // launcher: |off| < 2^31
#pragma unroll
      for (int i = 0; i < XC; ++i) {
        if (only >= 0 && i != only) continue;
        const bool ok = (unsigned)(po[i] + dy) < (unsigned)gq.O && (unsigned)(pi[i] + dx) < (unsigned)gq.I &&
                        ch * 32 + (tid & 3) * 8 < gq.cin;   // this thread's 8-channel group exists (partial last chunk)
        const float *src = ok ? xp[i] + off : a.zeros;
        xr[i][0] = *reinterpret_cast<const f32x4 *>(src);
        xr[i][1] = *reinterpret_cast<const f32x4 *>(src + 4);
      }
    } else {
      const int kc = kcs * 32;
#pragma unroll
      for (int i = 0; i < XC; ++i) {
        if (only >= 0 && i != only) continue;
        xr[i][0] = *reinterpret_cast<const f32x4 *>(xp[i] + kc);
        xr[i][1] = *reinterpret_cast<const f32x4 *>(xp[i] + kc + 4);
        if constexpr (PS) ej[i] = xe[i][(kcs * a.xexp_inv) >> 16];
      }
    }
  };
  auto load_x = [&](int ks) { load_x_sel(ks, -1); };
  auto split_chunk = [&](int buf, int i) {             // chunk i of the stage held in xr -> the three part images of `buf`
    char *dst = lds + buf * BUFB;
    u32x4 h, m, l;
    split3_oct(xr[i][0], xr[i][1], h, m, l);
    *reinterpret_cast<u32x4 *>(dst + xw[i]) = h;
    *reinterpret_cast<u32x4 *>(dst + PART + xw[i]) = m;
    *reinterpret_cast<u32x4 *>(dst + (NP - 1) * PART + xw[i]) = l;
  };
  // ---- H: scaled two-way split with ONE RUNNING EXPONENT PER ROW ------------------------------------------------------------------
  // The four lanes that fetch a row's 32 floats of a stage find their largest |x| (two quad-permute DPP steps); when it would pass
  // 2^15 under the row's exponent, the exponent drops (two bits of headroom) and the drop `de` goes into a table the MFMA side reads
  // after the next barrier -- the accumulators of that row are rescaled by 2^de (exact) before the stage's products are added.  A
  // second table keeps every row's current exponent for the epilogue.  Non-finite elements poison their own row only.
  int *de_tab = reinterpret_cast<int *>(lds + 2 * BUFB);              // [stage parity][BM]: row r at (r & 15) * MREP + (r >> 4)
  int *ex_tab = de_tab + 2 * BM;                                      // [BM]: the exponent each row's parts in LDS / accumulators carry
  int *fl_tab = ex_tab + BM;                                          // [stage parity]: the stage whose split last dropped some row's exponent (H)
  int *rmax = fl_tab + 4;                                             // [BM]: pair-image epilogue, bits of the rows' largest finite |y|
#ifdef ASX_EXPERIMENTAL_KERNELS
  if constexpr (H) {
    if (a.yexp != nullptr && tid < BM) rmax[tid] = 0;                 // (the stage loop's barriers order this before the epilogue's atomics)
  }
#endif
  int e_row[XC];
  int e_lo[XC];                                        // the lowest exponent the row has carried (its loudest stage so far)
  int xslot[XC];
#pragma unroll
  for (int i = 0; i < XC; ++i) {
    const int row = (tid + NT * i) >> 2;
    e_row[i] = 200;                                    // any first stage lowers it (accumulators are zero then)
    e_lo[i] = 200;
    xslot[i] = (row & 15) * MREP + (row >> 4);
  }
  auto split_chunk_h = [&](int buf, int i, int ks_of) {   // ks_of: the stage these rows are the x of
    float m = absmax_oct(xr[i][0], xr[i][1]);
    {
      int b = __float_as_int(m);
      int t = __builtin_amdgcn_update_dpp(b, b, 0xB1, 0xf, 0xf, false);   // quad_perm [1, 0, 3, 2]
      b = __float_as_int(fmaxf(__int_as_float(b), __int_as_float(t)));
      t = __builtin_amdgcn_update_dpp(b, b, 0x4E, 0xf, 0xf, false);       // quad_perm [2, 3, 0, 1]
      m = fmaxf(__int_as_float(b), __int_as_float(t));
    }
    // The exponent follows the row's magnitude along k BOTH ways (round 6): down at once when a stage would pass 2^15 (two bits of headroom), up
    // again when a stage's largest element sits more than F16X3_RISE bits below the range -- so that a row whose magnitude decays along k (a
    // spectrum's high bins, channels behind a folded normalisation) keeps 22-bit products instead of carrying its loudest stage's exponent to
    // the end.  Either way the accumulators follow by an exact power of two.  The rise is capped F16X3_RISE_CAP bits above the row's lowest
    // exponent so far: what such a stage adds is below 2^-40 of what the accumulators hold, and they can never be scaled out of fp32's range.
    const int need = f16_scale_exp(m);
    const int e_old = e_row[i];
    int e_new = e_old;
    if (need < e_old) e_new = need - 2;
    else if (need > e_old + F16X3_RISE && m > 0.f) e_new = min(need - 2, e_lo[i] + F16X3_RISE_CAP);
    e_lo[i] = min(e_lo[i], e_new);
    e_row[i] = e_new;
    if ((tid & 3) == 0) {
      de_tab[buf * BM + xslot[i]] = e_new - e_old;
      ex_tab[xslot[i]] = e_new;
      if (e_new != e_old) fl_tab[buf] = ks_of;         // every writer of a stage writes the same value
    }
    char *dst = lds + buf * BUFB;
    u32x4 h, l;
    split2h_oct(xr[i][0], xr[i][1], e_new, h, l);
    *reinterpret_cast<u32x4 *>(dst + xw[i]) = h;
    *reinterpret_cast<u32x4 *>(dst + PART + xw[i]) = l;
  };
  // PS: regroup, rescale to the row's exponent, store
  auto split_chunk_ps = [&](int buf, int i) {
    const int d = e_row[i] - ej[i];                      // <= 0: the row's exponent is its smallest span exponent
    const _Float16 f1 = (_Float16)__builtin_ldexpf(1.0f, d < -25 ? -25 : d);   // exact down to 2^-24 (fp16's smallest subnormal), 0 below
    const f16x2 ff = {f1, f1};
    auto sc = [&](float v) { return __builtin_bit_cast(unsigned, __builtin_bit_cast(f16x2, __float_as_uint(v)) * ff); };
    const u32x4 h = {sc(xr[i][0].x), sc(xr[i][0].y), sc(xr[i][1].x), sc(xr[i][1].y)};
    const u32x4 l = {sc(xr[i][0].z), sc(xr[i][0].w), sc(xr[i][1].z), sc(xr[i][1].w)};
    char *dst = lds + buf * BUFB;
    *reinterpret_cast<u32x4 *>(dst + xw[i]) = h;
    *reinterpret_cast<u32x4 *>(dst + PART + xw[i]) = l;
  };
  if constexpr (PS) {
#pragma unroll
    for (int i = 0; i < XC; ++i) {
      int64_t r = m0 + ((tid + NT * i) >> 2);
      r = r < a.M ? r : a.M - 1;
      xe[i] = a.xexp + r * a.xexp_n;
      int em = xe[i][0];
      for (int j = 1; j < a.xexp_n; ++j) em = min(em, xe[i][j]);
      e_row[i] = em;
      if ((tid & 3) == 0) ex_tab[xslot[i]] = em;         // read by the epilogue, behind the stage loop's barriers
    }
  }
  // MFMA side: this lane's rows are 16 m + li
  auto read_tab = [&](const int *tab, int (&out)[MREP]) {
    const u32x4 *q = reinterpret_cast<const u32x4 *>(tab + li * MREP);
#pragma unroll
    for (int j = 0; j < MREP / 4; ++j) {
      const u32x4 v = q[j];
      out[4 * j + 0] = (int)v.x;
      out[4 * j + 1] = (int)v.y;
      out[4 * j + 2] = (int)v.z;
      out[4 * j + 3] = (int)v.w;
    }
  };
  const int xf_off = li * 64 + ((lk ^ hsw((li >> 2) & 3)) << 4);   // fragment read: row 16 t + li, chunk lk

  f32x4 acc[NREP][MREP];
#pragma unroll
  for (int n = 0; n < NREP; ++n)
#pragma unroll
    for (int m = 0; m < MREP; ++m) acc[n][m] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // ---- prologue: stage 0 in LDS / registers, x of stage 1 in flight (same issue order as a steady stage: W, split, x)
  if constexpr (H) {
    load_x(0);
    load_w(IntC<0>{}, 0);
#pragma unroll
    for (int i = 0; i < XC; ++i) {
      if constexpr (PS) split_chunk_ps(0, i);
      else split_chunk_h(0, i, 0);
    }
    load_x(1);                                         // nk >= 2
  } else {
    load_x(0);
    load_w(IntC<0>{}, 0);
#pragma unroll
    for (int i = 0; i < XC; ++i) split_chunk(0, i);
    if (nk > 1) load_x(1);
  }

  // One stage.  MODE 0 (steady): fetch W of stage ks + 1, split x of stage ks + 1 (in registers since the previous stage) into the
  // other buffer, fetch x of stage ks + 2.  MODE 1 (second last): no x fetch.  MODE 2 (last): nothing but the MFMAs.  The modes
  // are separate code paths, not run-time conditions: a merged path made hipcc's counter pass wait for the just-issued prefetch
  // (`s_waitcnt vmcnt(0)` in front of the first MFMA of every stage).
  // Within a stage: fragment reads run one 16-row group ahead; the split of chunk i (44 VALU + 3 ds_write) is spread among the
  // MFMAs of row group 1 + 3 i by a scheduling-group pattern (one MFMA, three VALU).
  auto stage = [&](auto par, auto mode, int ks) {
    constexpr int P = decltype(par)::value;
    constexpr int MODE = decltype(mode)::value;
    __syncthreads();                                   // x parts of stage ks visible; the other buffer is free
    if constexpr (MODE < 2 && !(ABL & 8)) load_w(IntC<P ^ 1>{}, ks + 1);
    const char *xs = lds + P * BUFB + xf_off;
    bf16x8 xf[2][3];
#pragma unroll
    for (int p = 0; p < 3; ++p) xf[0][p] = *reinterpret_cast<const bf16x8 *>(xs + p * PART);
#pragma unroll
    for (int m = 0; m < MREP; ++m) {
      if (m + 1 < MREP) {
#pragma unroll
        for (int p = 0; p < 3; ++p) xf[(m + 1) & 1][p] = *reinterpret_cast<const bf16x8 *>(xs + p * PART + (m + 1) * 1024);
      }
#ifndef ASX_G3_NOSB
      __builtin_amdgcn_sched_barrier(0);
#endif
      const bool has_split = MODE < 2 && !(ABL & 1) && (m >= 1) && ((m - 1) % 3 == 0) && ((m - 1) / 3 < XC);
      if (has_split) split_chunk(P ^ 1, (m - 1) / 3);
      const bf16x8 xh = xf[m & 1][0], xm = xf[m & 1][1], xl = xf[m & 1][2];
      if constexpr ((ABL & 2) != 0) {
#pragma unroll
        for (int n = 0; n < NREP; ++n) {
          acc[n][m].x += (float)xh[0] + (float)xm[1] + (float)xl[2] + __uint_as_float(wr[P][n][0].x) + __uint_as_float(wr[P][n][1].y) +
                         __uint_as_float(wr[P][n][2].z);
        }
      } else {
        // smallest terms first; for a given product the NREP column tiles are independent accumulators (no back-to-back dependency)
#pragma unroll
        for (int n = 0; n < NREP; ++n) acc[n][m] = ASX_MFMA_BF16(__builtin_bit_cast(bf16x8, wr[P][n][2]), xh, acc[n][m]);
#pragma unroll
        for (int n = 0; n < NREP; ++n) acc[n][m] = ASX_MFMA_BF16(__builtin_bit_cast(bf16x8, wr[P][n][0]), xl, acc[n][m]);
#pragma unroll
        for (int n = 0; n < NREP; ++n) acc[n][m] = ASX_MFMA_BF16(__builtin_bit_cast(bf16x8, wr[P][n][1]), xm, acc[n][m]);
#pragma unroll
        for (int n = 0; n < NREP; ++n) acc[n][m] = ASX_MFMA_BF16(__builtin_bit_cast(bf16x8, wr[P][n][1]), xh, acc[n][m]);
#pragma unroll
        for (int n = 0; n < NREP; ++n) acc[n][m] = ASX_MFMA_BF16(__builtin_bit_cast(bf16x8, wr[P][n][0]), xm, acc[n][m]);
#pragma unroll
        for (int n = 0; n < NREP; ++n) acc[n][m] = ASX_MFMA_BF16(__builtin_bit_cast(bf16x8, wr[P][n][0]), xh, acc[n][m]);
      }
#ifndef ASX_G3_NOSGB
      if (has_split) {
#pragma unroll
        for (int g = 0; g < 6 * NREP; ++g) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // one MFMA
          __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);   // three VALU
        }
        __builtin_amdgcn_sched_group_barrier(0x200, 3, 0);     // the chunk's three ds_write
      }
#endif
#ifndef ASX_G3_NOSB
      __builtin_amdgcn_sched_barrier(0);
#endif
      if constexpr (MODE == 0 && !(ABL & 1)) {
        if (m == 1 + 3 * (XC - 1)) load_x(ks + 2);            // the registers of the chunk just split are free
      }
    }
  };
  // The same stage on the fp16 x 3 arithmetic.  After the barrier the exponent drops of the rows split during the previous stage
  // are applied to the accumulators (rare after the first stages); the split of stage ks + 1 is spread among the MFMAs as above (one
  // MFMA, four VALU), and every chunk's registers are refilled with stage ks + 2 right after its split -- a whole stage ahead.
  auto stage_h = [&](auto par, auto mode, int ks) {
    constexpr int P = decltype(par)::value;
    constexpr int MODE = decltype(mode)::value;
    __syncthreads();
    if constexpr (MODE < 2 && !(ABL & 8)) load_w(IntC<P ^ 1>{}, ks + 1);
    // The exponent drops of the rows split during the previous stage: the accumulators of such a row are multiplied by 2^de (exact; 0 for
    // de < -149: what they held is then below 2^-100 of what follows).  Rare after a tile's first stages, so the 4 MREP NREP multiplies
    // sit behind a workgroup-uniform test of the stage stamp the splitting threads leave in fl_tab (round 6: as an unconditional
    // multiply by 1.0 they were 4-10 % of the launch, profiles/r06_tdf3h_abl.txt).  In-place inline assembly: as a C++ multiply inside a
    // branch the compiler kept a second register set for the rescaled accumulators (256 VGPRs and 27 spilled against 220 and none:
    // tools/runs/r5_run15.sh).  ABL & 32: the unconditional form, for A/B.
    float fr[MREP];
    if constexpr ((ABL & 32) != 0) {
      int de[MREP];
      read_tab(de_tab + P * BM, de);
#pragma unroll
      for (int m = 0; m < MREP; ++m) fr[m] = __builtin_ldexpf(1.0f, de[m]);
#pragma unroll
      for (int n = 0; n < NREP; ++n) acc[n][0] *= fr[0];
    } else if constexpr (!(ABL & 16) && !PS) {
      if (__builtin_amdgcn_readfirstlane(fl_tab[P]) == ks) {
        int de[MREP];
        read_tab(de_tab + P * BM, de);
#pragma unroll
        for (int m = 0; m < MREP; ++m) {
          const float f = __builtin_ldexpf(1.0f, de[m]);
#pragma unroll
          for (int n = 0; n < NREP; ++n) {
            asm volatile("v_mul_f32 %0, %4, %0\n\tv_mul_f32 %1, %4, %1\n\tv_mul_f32 %2, %4, %2\n\tv_mul_f32 %3, %4, %3"
                         : "+v"(acc[n][m].x), "+v"(acc[n][m].y), "+v"(acc[n][m].z), "+v"(acc[n][m].w)
                         : "v"(f));
          }
        }
      }
    }
    const char *xs = lds + P * BUFB + xf_off;
    f16x8 xf[2][2];
#pragma unroll
    for (int p = 0; p < 2; ++p) xf[0][p] = *reinterpret_cast<const f16x8 *>(xs + p * PART);
#pragma unroll
    for (int m = 0; m < MREP; ++m) {
      if (m + 1 < MREP) {
#pragma unroll
        for (int p = 0; p < 2; ++p) xf[(m + 1) & 1][p] = *reinterpret_cast<const f16x8 *>(xs + p * PART + (m + 1) * 1024);
      }
      __builtin_amdgcn_sched_barrier(0);
      const bool has_split = MODE < 2 && !(ABL & 1) && (m >= 1) && ((m - 1) % 3 == 0) && ((m - 1) / 3 < XC);
      if (has_split) {
        if constexpr (PS) split_chunk_ps(P ^ 1, (m - 1) / 3);
        else split_chunk_h(P ^ 1, (m - 1) / 3, ks + 1);
      }
      const f16x8 xh = xf[m & 1][0], xl = xf[m & 1][1];
      // smallest terms first; the NREP column tiles of one product are independent accumulators
#pragma unroll
      for (int n = 0; n < NREP; ++n) acc[n][m] = ASX_MFMA_F16(__builtin_bit_cast(f16x8, wr[P][n][1]), xh, acc[n][m]);
#pragma unroll
      for (int n = 0; n < NREP; ++n) acc[n][m] = ASX_MFMA_F16(__builtin_bit_cast(f16x8, wr[P][n][0]), xl, acc[n][m]);
#pragma unroll
      for (int n = 0; n < NREP; ++n) acc[n][m] = ASX_MFMA_F16(__builtin_bit_cast(f16x8, wr[P][n][0]), xh, acc[n][m]);
      if (m + 1 < MREP && (ABL & 32)) {                // the next row group's accumulators, behind this group's MFMAs
#pragma unroll
        for (int n = 0; n < NREP; ++n) acc[n][m + 1] *= fr[m + 1];
      }
#ifndef ASX_G3_NOSGB
      if (has_split) {
#pragma unroll
        for (int g = 0; g < 3 * NREP; ++g) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // one MFMA
          __builtin_amdgcn_sched_group_barrier(0x002, PS ? 2 : 5, 0);   // five VALU (two with pair-image operands)
        }
        __builtin_amdgcn_sched_group_barrier(0x200, PS ? 2 : 4, 0);     // the chunk's ds_writes (two parts, two table entries)
      }
#endif
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (MODE == 0) {
        if (has_split) load_x_sel(ks + 2, (m - 1) / 3);
      }
    }
  };
  {
    // nk is even and >= 2 (see above) -- one tail sequence instead of three
    int ks = 0;
    if constexpr (H) {
      for (; ks + 2 < nk; ks += 2) {
        stage_h(IntC<0>{}, IntC<0>{}, ks);
        stage_h(IntC<1>{}, IntC<0>{}, ks + 1);
      }
      stage_h(IntC<0>{}, IntC<1>{}, ks);
      stage_h(IntC<1>{}, IntC<2>{}, ks + 1);
      // back to the operands' own scale: 2^-(e_x[row] + e_w[this lane's four columns]), exact
      const int *wexp = reinterpret_cast<const int *>(w3 + (int64_t)ntiles * nk * 128);
      int ex[MREP];
      read_tab(ex_tab, ex);                            // written before the last barrier (the last split is in stage nk - 2)
#pragma unroll
      for (int n = 0; n < NREP; ++n) {
        int nt = (n0 >> 4) + wave * NREP + n;
        nt = nt < ntiles ? nt : ntiles - 1;
        const int ew = wexp[nt * 4 + lk];
#pragma unroll
        for (int m = 0; m < MREP; ++m) {
          ldexp4_inplace(acc[n][m], -(ex[m] + ew));
        }
      }
    } else {
      for (; ks + 2 < nk; ks += 2) {
        stage(IntC<0>{}, IntC<0>{}, ks);
        stage(IntC<1>{}, IntC<0>{}, ks + 1);
      }
      stage(IntC<0>{}, IntC<1>{}, ks);
      stage(IntC<1>{}, IntC<2>{}, ks + 1);
    }
  }

  // ---- epilogue (the arithmetic of tdf2_kernel's three paths) ----------------------------------------------------------------
  const bool full = (m0 + BM <= a.M) && (n0 + BN <= a.N);
  if constexpr (GATHER && NREP == 2) {
    if (a.glu_cout > 0) {
      // GLU epilogue of the Demucs rewrite convs (kernels_halo.h: hg_kernel, GG_GLU): a wave's two column fragments are a value /
      // gate pair; output channel of value column c: c / 2 (fragment granularity), same arithmetic as hg_kernel
      const int c0 = n0 + wave * 32;                   // first column of the value fragment
      const int oc = (c0 >> 1) + lk * 4;
      if (c0 < a.N && oc < a.glu_cout) {
        f32x4 bv = {0.f, 0.f, 0.f, 0.f}, bg = bv;
        if (a.bias != nullptr) {
          bv = *reinterpret_cast<const f32x4 *>(a.bias + c0 + lk * 4);
          bg = *reinterpret_cast<const f32x4 *>(a.bias + c0 + 16 + lk * 4);
        }
#pragma unroll
        for (int m = 0; m < MREP; ++m) {
          const int64_t row = m0 + m * 16 + li;
          if (row >= a.M) continue;
          const f32x4 v = acc[0][m] + bv, g4 = acc[1][m] + bg;
          f32x4 o4;
          o4.x = v.x * __builtin_amdgcn_rcpf(1.0f + __expf(-g4.x));
          o4.y = v.y * __builtin_amdgcn_rcpf(1.0f + __expf(-g4.y));
          o4.z = v.z * __builtin_amdgcn_rcpf(1.0f + __expf(-g4.z));
          o4.w = v.w * __builtin_amdgcn_rcpf(1.0f + __expf(-g4.w));
          *reinterpret_cast<f32x4 *>(a.y + row * ldy + oc) = o4;
        }
      }
      return;
    }
  }
#ifdef ASX_EXPERIMENTAL_KERNELS
  if constexpr (H && !GATHER) {
    if (a.yexp != nullptr) {
      // ---- y as a PAIR IMAGE for the row GEMM that is its only reader (launcher: no residual, no rotary step, N % 32 == 0).  Pass 1: the
      // epilogue arithmetic of the generic path in place, the rows' largest finite |y| over this tile's columns into rmax (LDS atomics:
      // four lanes of four waves per row).  Pass 2: one exponent per (row, tile) -- the maximum lands in [2^14, 2^15) -- and each lane's
      // four columns as h0..h3 l0..l3 in the 16 bytes the fp32 values would have taken.
#pragma unroll
      for (int m = 0; m < MREP; ++m) {
        const int64_t row = m0 + m * 16 + li;
        const bool rok = row < a.M;
        const int c = rok ? (int)(((uint32_t)row / (uint32_t)a.T) % (uint32_t)a.C) : 0;
        const float sc = a.scale ? a.scale[c] : 1.f, sh = a.shift ? a.shift[c] : 0.f;
        const float rw = (rok && a.rscale) ? a.rscale[row] : 1.f;
        float mx = 0.f;
#pragma unroll
        for (int n = 0; n < NREP; ++n) {
          const int col = n0 + wave * 16 * NREP + n * 16 + lk * 4;
          const bool ok = rok && col < a.N;
          const f32x4 b4 = (ok && a.bias != nullptr) ? *reinterpret_cast<const f32x4 *>(a.bias + col) : (f32x4){0.f, 0.f, 0.f, 0.f};
          const f32x4 v = acc[n][m];
          f32x4 o;
          o.x = tdf_act(sc * __fmaf_rn(v.x, rw, b4.x) + sh, a.relu);
          o.y = tdf_act(sc * __fmaf_rn(v.y, rw, b4.y) + sh, a.relu);
          o.z = tdf_act(sc * __fmaf_rn(v.z, rw, b4.z) + sh, a.relu);
          o.w = tdf_act(sc * __fmaf_rn(v.w, rw, b4.w) + sh, a.relu);
          if (!ok) o = (f32x4){0.f, 0.f, 0.f, 0.f};
          acc[n][m] = o;
          mx = fmaxf(mx, fmaxf(fmaxf(finite_or_zero(o.x), finite_or_zero(o.y)), fmaxf(finite_or_zero(o.z), finite_or_zero(o.w))));
        }
        atomicMax(&rmax[m * 16 + li], __float_as_int(mx));   // non-negative floats order as their bit patterns
      }
      __syncthreads();
#pragma unroll
      for (int m = 0; m < MREP; ++m) {
        const int64_t row = m0 + m * 16 + li;
        const int e = f16_scale_exp(__int_as_float(rmax[m * 16 + li]));
#pragma unroll
        for (int n = 0; n < NREP; ++n) {
          const int col = n0 + wave * 16 * NREP + n * 16 + lk * 4;
          if (row >= a.M || col >= a.N) continue;
          unsigned h0, h1, l0, l1;
          split2h_pair(acc[n][m].x, acc[n][m].y, e, h0, l0);
          split2h_pair(acc[n][m].z, acc[n][m].w, e, h1, l1);
          *reinterpret_cast<u32x4 *>(a.y + row * ldy + col) = (u32x4){h0, h1, l0, l1};
        }
      }
      if (tid < BM && m0 + tid < a.M) a.yexp[(m0 + tid) * a.yexp_n + bg] = f16_scale_exp(__int_as_float(rmax[tid]));
      return;
    }
  }
#endif
  if constexpr ((ABL & 4) != 0) {
    float chk = 0.f;
#pragma unroll
    for (int n = 0; n < NREP; ++n)
#pragma unroll
      for (int m = 0; m < MREP; ++m) chk += acc[n][m].x + acc[n][m].y + acc[n][m].z + acc[n][m].w;
    if (chk == 1.2345e-30f) a.y[0] = chk;
    return;
  }
  if (full) {
    f32x4 bz[NREP];
#pragma unroll
    for (int n = 0; n < NREP; ++n) {
      const int col = n0 + wave * 16 * NREP + n * 16 + lk * 4;
      bz[n] = (a.bias != nullptr) ? *reinterpret_cast<const f32x4 *>(a.bias + col) : (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    const uint32_t voff_r = (uint32_t)((li * ldr + wave * 16 * NREP + lk * 4) * 4);
    const uint32_t voff_y = (uint32_t)((li * ldy + wave * 16 * NREP + lk * 4) * 4);
    if (a.relu == 1) {
      // ReLU + residual (every TDF layer of ConvTDFNet): the residual of RING 16-row groups stays in flight
      constexpr int RING = MREP < 4 ? MREP : 4;
      f32x4 rs[RING][NREP];
      auto fetch = [&](int m) {
        const char *rb = reinterpret_cast<const char *>(a.res) + ((m0 + m * 16) * ldr + n0) * 4;
#pragma unroll
        for (int n = 0; n < NREP; ++n)
          rs[m % RING][n] = (a.res == nullptr) ? (f32x4){0.f, 0.f, 0.f, 0.f}
                            : ((a.nt & 2) ? __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(rb + voff_r + n * 64))
                                          : *reinterpret_cast<const f32x4 *>(rb + voff_r + n * 64));
      };
#pragma unroll
      for (int m = 0; m < RING; ++m) fetch(m);
#pragma unroll
      for (int m = 0; m < MREP; ++m) {
        const uint32_t row = (uint32_t)(m0 + m * 16 + li);
        const int c = (int)((row / (uint32_t)a.T) % (uint32_t)a.C);
        const float sc = a.scale ? a.scale[c] : 1.f, sh = a.shift ? a.shift[c] : 0.f;
        char *yb = reinterpret_cast<char *>(a.y) + ((m0 + m * 16) * ldy + n0) * 4;
#pragma unroll
        for (int n = 0; n < NREP; ++n) {
          const f32x4 v = acc[n][m];
          const f32x4 r = rs[m % RING][n];
          f32x4 o;
          o.x = fmaxf(sc * (v.x + bz[n].x) + sh, 0.f) + r.x;
          o.y = fmaxf(sc * (v.y + bz[n].y) + sh, 0.f) + r.y;
          o.z = fmaxf(sc * (v.z + bz[n].z) + sh, 0.f) + r.z;
          o.w = fmaxf(sc * (v.w + bz[n].w) + sh, 0.f) + r.w;
          if (a.nt & 1) __builtin_nontemporal_store(o, reinterpret_cast<f32x4 *>(yb + voff_y + n * 64));
          else *reinterpret_cast<f32x4 *>(yb + voff_y + n * 64) = o;
        }
        if (m + RING < MREP) fetch(m + RING);
      }
    } else {
#pragma unroll
      for (int mg = 0; mg < MREP; mg += 2) {
        float sc[2], sh[2], rw[2];
        f32x4 rs[2][NREP];
#pragma unroll
        for (int m = 0; m < 2; ++m) {
          const uint32_t row = (uint32_t)(m0 + (mg + m) * 16 + li);
          const int c = (int)((row / (uint32_t)a.T) % (uint32_t)a.C);
          sc[m] = a.scale ? a.scale[c] : 1.f;
          sh[m] = a.shift ? a.shift[c] : 0.f;
          rw[m] = a.rscale ? a.rscale[row] : 1.f;
          const char *rb = reinterpret_cast<const char *>(a.res) + ((m0 + (mg + m) * 16) * ldr + n0) * 4;
#pragma unroll
          for (int n = 0; n < NREP; ++n)
            rs[m][n] = (a.res != nullptr) ? *reinterpret_cast<const f32x4 *>(rb + voff_r + n * 64) : (f32x4){0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int m = 0; m < 2; ++m) {
          char *yb = reinterpret_cast<char *>(a.y) + ((m0 + (mg + m) * 16) * ldy + n0) * 4;
#pragma unroll
          for (int n = 0; n < NREP; ++n) {
            const f32x4 v = acc[n][mg + m];
            f32x4 o;
            o.x = tdf_act(sc[m] * __fmaf_rn(v.x, rw[m], bz[n].x) + sh[m], a.relu) + rs[m][n].x;
            o.y = tdf_act(sc[m] * __fmaf_rn(v.y, rw[m], bz[n].y) + sh[m], a.relu) + rs[m][n].y;
            o.z = tdf_act(sc[m] * __fmaf_rn(v.z, rw[m], bz[n].z) + sh[m], a.relu) + rs[m][n].z;
            o.w = tdf_act(sc[m] * __fmaf_rn(v.w, rw[m], bz[n].w) + sh[m], a.relu) + rs[m][n].w;
            *reinterpret_cast<f32x4 *>(yb + voff_y + n * 64) =
                tdf_rot4(a, o, m0 + (mg + m) * 16 + li, n0 + wave * 16 * NREP + n * 16 + lk * 4);
          }
        }
      }
    }
  } else {
#pragma unroll
    for (int m = 0; m < MREP; ++m) {
      const int64_t row = m0 + m * 16 + li;
      const bool rok = row < a.M;
      const int c = rok ? (int)(((uint32_t)row / (uint32_t)a.T) % (uint32_t)a.C) : 0;
      const float sc = a.scale ? a.scale[c] : 1.f, sh = a.shift ? a.shift[c] : 0.f;
      const float rw = (rok && a.rscale) ? a.rscale[row] : 1.f;
#pragma unroll
      for (int n = 0; n < NREP; ++n) {
        const int col = n0 + wave * 16 * NREP + n * 16 + lk * 4;
        const f32x4 v = acc[n][m];
        if (!rok || col >= a.N) continue;            // N % 8 == 0 and col % 4 == 0: a float4 is inside or outside as a whole
        const f32x4 b4 = (a.bias != nullptr) ? *reinterpret_cast<const f32x4 *>(a.bias + col) : (f32x4){0.f, 0.f, 0.f, 0.f};
        const f32x4 r4 = (a.res != nullptr) ? *reinterpret_cast<const f32x4 *>(a.res + row * ldr + col) : (f32x4){0.f, 0.f, 0.f, 0.f};
        f32x4 o;
        o.x = tdf_act(sc * __fmaf_rn(v.x, rw, b4.x) + sh, a.relu) + r4.x;
        o.y = tdf_act(sc * __fmaf_rn(v.y, rw, b4.y) + sh, a.relu) + r4.y;
        o.z = tdf_act(sc * __fmaf_rn(v.z, rw, b4.z) + sh, a.relu) + r4.z;
        o.w = tdf_act(sc * __fmaf_rn(v.w, rw, b4.w) + sh, a.relu) + r4.w;
        *reinterpret_cast<f32x4 *>(a.y + row * ldy + col) = tdf_rot4(a, o, row, col);
      }
    }
  }
}

}  // namespace asx
