// Fast STFT / iSTFT path for n_fft = 6144, hop = 1024 (the UVR-MDX-NET HQ geometry): a real FFT of 6144 points as one
// complex FFT of 3072 = 12 x 16 x 16 points in THREE register-resident Stockham passes (radix 12, 16, 16) with LDS
// exchanges, instead of the generic six-pass radix-{4,3} loop of kernels_fft.h.  Replaces, on that geometry,
// torch.stft / torch.istft as used by uvr_lib_v5/stft.py:41,117 and the frame overlap-add inside torch.istft.
//
// The transform, per 256-thread workgroup and frame:
//   pass A: thread j holds z[j + 256 r], r < 12 (z[m] = x[2m] + i x[2m+1]); 12-point DFT in registers (3 x radix-4, W12
//           twiddles, 4 x radix-3), written as six 16-byte stores per thread (row stride 96 B: conflict-free for
//           ds_write_b128's 8-lane groups);
//   pass B: threads j < 192 read in[j + 192 r] (consecutive lanes, conflict-free ds_read_b64), twiddle by W192^(k r),
//           16-point DFT (radix-4 x radix-4), write out[q*192 + k + 12 r] into blocks padded to 204 so that a 16-lane
//           ds_write_b64 group never wraps onto its own banks;
//   pass C: same read pattern, twiddle W3072^(j r), 16-point DFT -> Z[j + 192 r];
//   split (forward) / merge (inverse): X[k] = E + W6144^k O pairs Z[k] with conj Z[3072 - k].
// Complex arithmetic is packed fp32 (one or two v_pk_* instructions per operation, see below).
//
// Kernels:
//   stft3p_kernel   forward, ~16 consecutive frames of one (chunk, channel) per workgroup, samples in an LDS ring
//   istft3p_kernel  inverse, ~16 frames per workgroup: spectrum rows prefetched global -> LDS by DMA, frames ACCUMULATED INTO AN
//                   LDS RING of n_fft floats (a frame covers six hops; after frame t hop t is complete, is divided by the
//                   window envelope, multiplied by the chunk's Hann window and written).  The [B, 2, T, n_fft] frame buffer of
//                   the generic path never exists; only the five partial hops at either end of a workgroup's frame range
//                   travel through a small seam buffer,
//   seam3_kernel    which folds them (tail of group g + head of group g + 1) in a fixed order -- deterministic, no atomics;
//   stft3_kernel / istft3_kernel   the first generation (one frame per workgroup forward; inverse without prefetch, also the
//                   denoise-combine path), kept behind ASX_FFT3P=0.
//
// The per-thread stage bodies are plain inline functions of (thread id, "LDS" pointers) so that tests/host/fft3_host.cpp
// can run them on the CPU, thread by thread, against a reference DFT (ASX_HOST_TEST).
#pragma once
#ifdef ASX_HOST_TEST
#include <cmath>
#include <cstdint>
struct float2 { float x, y; };
struct float4 { float x, y, z, w; };
static inline float2 make_float2(float x, float y) { return float2{x, y}; }
static inline float4 make_float4(float x, float y, float z, float w) { return float4{x, y, z, w}; }
#define ASX_HD inline
#else
#include <hip/hip_runtime.h>
#include <stdint.h>
#define ASX_HD __host__ __device__ __forceinline__
#endif

namespace asx {
namespace f3 {

constexpr int NFFT = 6144, NH = 3072, HOP = 1024, HPF = NFFT / HOP;   // hops per frame
constexpr int NB = 192;          // butterflies (= active threads) of passes B and C
constexpr int BSTRIDE = 204;     // padded block stride (complex) of the pass-B output
constexpr int LDS_X = 16 * BSTRIDE;                                   // float2 elements of the one exchange buffer (>= NH)

// ---- complex arithmetic -------------------------------------------------------------------------------------------------
// A complex number is one 64-bit register pair.  On the device every complex add, "add a value rotated by +-i", conjugate
// add and complex multiply is ONE or TWO packed-fp32 instructions (v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32) whose
// op_sel / neg modifiers do the swaps, conjugations and sign flips -- written as inline asm because the SLP vectoriser,
// left to itself, pairs unrelated scalars and spends a third of the kernel on v_mov shuffles.  The host build
// (ASX_HOST_TEST) states the same operations in scalar C++.
#ifdef ASX_HOST_TEST
typedef float2 cplx;
typedef float4 cplx2;
ASX_HD cplx mk(float x, float y) { return make_float2(x, y); }
ASX_HD cplx2 pair(cplx a, cplx b) { return make_float4(a.x, a.y, b.x, b.y); }
ASX_HD cplx ca(cplx a, cplx b) { return mk(a.x + b.x, a.y + b.y); }
ASX_HD cplx cs(cplx a, cplx b) { return mk(a.x - b.x, a.y - b.y); }
ASX_HD cplx cneg(cplx a) { return mk(-a.x, -a.y); }
ASX_HD cplx cscale(cplx a, float k) { return mk(a.x * k, a.y * k); }
ASX_HD cplx cmad(cplx a, cplx b, float k) { return mk(a.x + k * b.x, a.y + k * b.y); }     // a + k b
ASX_HD cplx emul(cplx a, cplx b) { return mk(a.x * b.x, a.y * b.y); }                        // element-wise
ASX_HD cplx cmul(cplx a, cplx w) { return mk(a.x * w.x - a.y * w.y, a.y * w.x + a.x * w.y); }
ASX_HD cplx cmulc(cplx a, cplx w) { return mk(a.x * w.x + a.y * w.y, a.y * w.x - a.x * w.y); }   // a conj(w)
ASX_HD cplx cmul_k(cplx a, cplx w) { return cmul(a, w); }
ASX_HD cplx cmulc_k(cplx a, cplx w) { return cmulc(a, w); }
ASX_HD cplx addc(cplx a, cplx b) { return mk(a.x + b.x, a.y - b.y); }                        // a + conj(b)
ASX_HD cplx subc(cplx a, cplx b) { return mk(a.x - b.x, a.y + b.y); }                        // a - conj(b)
ASX_HD cplx add_pi(cplx a, cplx b) { return mk(a.x - b.y, a.y + b.x); }                      // a + i b
ASX_HD cplx add_mi(cplx a, cplx b) { return mk(a.x + b.y, a.y - b.x); }                      // a - i b
#else
typedef float cplx __attribute__((ext_vector_type(2)));
typedef float cplx2 __attribute__((ext_vector_type(4)));
#define ASX_PK __device__ __forceinline__
ASX_PK cplx mk(float x, float y) { return (cplx){x, y}; }
ASX_PK cplx2 pair(cplx a, cplx b) { return __builtin_shufflevector(a, b, 0, 1, 2, 3); }
ASX_PK cplx ca(cplx a, cplx b) { return a + b; }
ASX_PK cplx cs(cplx a, cplx b) { return a - b; }
ASX_PK cplx cneg(cplx a) { return -a; }
ASX_PK cplx cscale(cplx a, float k) { return a * k; }
ASX_PK cplx cmad(cplx a, cplx b, float k) { return a + b * k; }
ASX_PK cplx emul(cplx a, cplx b) { return a * b; }
ASX_PK cplx cmul(cplx a, cplx w) {
  cplx t, r;
  asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(t) : "v"(a), "v"(w));
  asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[0,1,0]" : "=v"(r) : "v"(a), "v"(w), "v"(t));
  return r;
}
ASX_PK cplx cmulc(cplx a, cplx w) {
  cplx t, r;
  asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(t) : "v"(a), "v"(w));
  asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_hi:[0,1,0]" : "=v"(r) : "v"(a), "v"(w), "v"(t));
  return r;
}
// the same with a wave-uniform (compile-time) factor kept in a scalar register pair
ASX_PK cplx cmul_k(cplx a, cplx w) {
  cplx t, r;
  asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(t) : "v"(a), "s"(w));
  asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[0,1,0]" : "=v"(r) : "v"(a), "s"(w), "v"(t));
  return r;
}
ASX_PK cplx cmulc_k(cplx a, cplx w) {
  cplx t, r;
  asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(t) : "v"(a), "s"(w));
  asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_hi:[0,1,0]" : "=v"(r) : "v"(a), "s"(w), "v"(t));
  return r;
}
ASX_PK cplx addc(cplx a, cplx b) {
  cplx r;
  asm("v_pk_add_f32 %0, %1, %2 neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
ASX_PK cplx subc(cplx a, cplx b) {
  cplx r;
  asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1]" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
ASX_PK cplx add_pi(cplx a, cplx b) {       // (a.x - b.y, a.y + b.x)
  cplx r;
  asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
ASX_PK cplx add_mi(cplx a, cplx b) {       // (a.x + b.y, a.y - b.x)
  cplx r;
  asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
#undef ASX_HD
#define ASX_HD __device__ __forceinline__
#endif
// a + (SIGN i) b and a - (SIGN i) b: SIGN < 0 is the forward transform
template <int SIGN>
ASX_HD cplx addrot(cplx a, cplx b) { return SIGN < 0 ? add_mi(a, b) : add_pi(a, b); }
template <int SIGN>
ASX_HD cplx subrot(cplx a, cplx b) { return SIGN < 0 ? add_pi(a, b) : add_mi(a, b); }
// multiply by exp(SIGN i theta), K = (cos theta, sin theta) a compile-time constant
template <int SIGN>
ASX_HD cplx twk(cplx a, float c, float s) { return SIGN < 0 ? cmulc_k(a, mk(c, s)) : cmul_k(a, mk(c, s)); }
// multiply by a table twiddle (tables hold the forward sign; the inverse conjugates)
template <int SIGN>
ASX_HD cplx twt(cplx a, cplx w) { return SIGN < 0 ? cmul(a, w) : cmulc(a, w); }

template <int SIGN>
ASX_HD void dft4(cplx &a, cplx &b, cplx &c, cplx &d) {
  const cplx t0 = ca(a, c), t1 = cs(a, c), t2 = ca(b, d), t3 = cs(b, d);
  a = ca(t0, t2);
  c = cs(t0, t2);
  b = addrot<SIGN>(t1, t3);
  d = subrot<SIGN>(t1, t3);
}
template <int SIGN>
ASX_HD void dft3(cplx &a, cplx &b, cplx &c) {
  const float s3 = 0.86602540378443864676f;
  const cplx s = ca(b, c), d = cscale(cs(b, c), s3);
  const cplx m = cmad(a, s, -0.5f);
  a = ca(a, s);
  b = addrot<SIGN>(m, d);
  c = subrot<SIGN>(m, d);
}

// 12-point DFT, natural order in and out: n = 3 n1 + n2, k = k1 + 4 k2
template <int SIGN>
ASX_HD void dft12(cplx *v) {
  const float C1 = 0.86602540378443864676f, S1 = 0.5f;               // W12^1 = cos 30, sin 30
  const float C2 = 0.5f, S2 = 0.86602540378443864676f;               // W12^2
#pragma unroll
  for (int n2 = 0; n2 < 3; ++n2) dft4<SIGN>(v[n2], v[n2 + 3], v[n2 + 6], v[n2 + 9]);   // k1 = 0..3 at v[n2 + 3 k1]
  // W12^(n2 k1): n2 = 1: k1 = 1, 2, 3 -> W^1, W^2, W^3 (= -+i); n2 = 2: k1 = 1, 2, 3 -> W^2, W^4, W^6 (= -1)
  v[1 + 3] = twk<SIGN>(v[1 + 3], C1, S1);
  v[1 + 6] = twk<SIGN>(v[1 + 6], C2, S2);
  v[2 + 3] = twk<SIGN>(v[2 + 3], C2, S2);
  v[2 + 6] = twk<SIGN>(v[2 + 6], -C2, S2);
  // the W^3 = -+i of v[10] and the W^6 = -1 of v[11] are folded into the radix-3 butterfly of k1 = 3 below
  cplx o[12];
#pragma unroll
  for (int k1 = 0; k1 < 3; ++k1) {
    cplx x = v[3 * k1], y = v[3 * k1 + 1], z = v[3 * k1 + 2];
    dft3<SIGN>(x, y, z);
    o[k1] = x;
    o[k1 + 4] = y;
    o[k1 + 8] = z;
  }
  {
    // inputs a = v[9], b = (SIGN i) v[10], c = -v[11]
    const float s3 = 0.86602540378443864676f;
    const cplx a = v[9], p = v[10], q = v[11];
    const cplx s = SIGN < 0 ? add_mi(cneg(q), p) : add_pi(cneg(q), p);     // b + c = (SIGN i) p - q
    const cplx dd = SIGN < 0 ? add_mi(q, p) : add_pi(q, p);                // b - c = (SIGN i) p + q
    const cplx d = cscale(dd, s3);
    const cplx m = cmad(a, s, -0.5f);
    o[3] = ca(a, s);
    o[7] = addrot<SIGN>(m, d);
    o[11] = subrot<SIGN>(m, d);
  }
#pragma unroll
  for (int i = 0; i < 12; ++i) v[i] = o[i];
}

// 16-point DFT, natural order in and out: n = 4 n1 + n2, k = k1 + 4 k2
template <int SIGN>
ASX_HD void dft16(cplx *v) {
  const float C1 = 0.92387953251128675613f, S1 = 0.38268343236508977173f;   // W16^1
  const float C2 = 0.70710678118654752440f;                                  // W16^2 (cos = sin)
#pragma unroll
  for (int n2 = 0; n2 < 4; ++n2) dft4<SIGN>(v[n2], v[n2 + 4], v[n2 + 8], v[n2 + 12]);  // k1 at v[n2 + 4 k1]
  // W16^(n2 k1); the W^4 = -+i of v[10] is folded into the second-stage butterfly of k1 = 2
  v[1 + 4] = twk<SIGN>(v[1 + 4], C1, S1);       // 1
  v[1 + 8] = twk<SIGN>(v[1 + 8], C2, C2);       // 2
  v[1 + 12] = twk<SIGN>(v[1 + 12], S1, C1);     // 3
  v[2 + 4] = twk<SIGN>(v[2 + 4], C2, C2);       // 2
  v[2 + 12] = twk<SIGN>(v[2 + 12], -C2, C2);    // 6
  v[3 + 4] = twk<SIGN>(v[3 + 4], S1, C1);       // 3
  v[3 + 8] = twk<SIGN>(v[3 + 8], -C2, C2);      // 6
  v[3 + 12] = twk<SIGN>(v[3 + 12], -C1, -S1);   // 9
  cplx o[16];
#pragma unroll
  for (int k1 = 0; k1 < 4; ++k1) {
    cplx a = v[4 * k1], b = v[4 * k1 + 1], c = v[4 * k1 + 2], d = v[4 * k1 + 3];
    if (k1 == 2) {
      // c stands for (SIGN i) c:  t0 = a + (SIGN i) c, t1 = a - (SIGN i) c
      const cplx t0 = addrot<SIGN>(a, c), t1 = subrot<SIGN>(a, c), t2 = ca(b, d), t3 = cs(b, d);
      a = ca(t0, t2);
      c = cs(t0, t2);
      b = addrot<SIGN>(t1, t3);
      d = subrot<SIGN>(t1, t3);
    } else {
      dft4<SIGN>(a, b, c, d);
    }
    o[k1] = a;                                  // X[k1 + 4 k2] = k2-th output
    o[k1 + 4] = b;
    o[k1 + 8] = c;
    o[k1 + 12] = d;
  }
#pragma unroll
  for (int i = 0; i < 16; ++i) v[i] = o[i];
}

// ---- the three passes, per thread --------------------------------------------------------------------------------------
// pass A: thread j < 256 holds in[j + 256 r] (r < 12); Stockham radix 12, ns = 1: out[12 j + r]
template <int SIGN>
ASX_HD void pass_a(int j, cplx *a, cplx *bufA) {
  dft12<SIGN>(a);
  cplx2 *p = reinterpret_cast<cplx2 *>(bufA + 12 * j);
#pragma unroll
  for (int s = 0; s < 6; ++s) p[s] = pair(a[2 * s], a[2 * s + 1]);
}
// pass B: thread j < 192; radix 16, ns = 12: k = j % 12, q = j / 12; twiddle exp(SIGN 2 pi i k r / 192) = twB[r * 12 + k]
// (table holds the forward sign; the inverse conjugates); out[q * 192 + k + 12 r] in blocks of BSTRIDE.  Load and store are
// separate calls: with a barrier between them one LDS buffer of LDS_X elements serves every exchange.
ASX_HD void pass_b_load(int j, const cplx *buf, cplx *c) {
#pragma unroll
  for (int r = 0; r < 16; ++r) c[r] = buf[j + NB * r];
}
template <int SIGN>
ASX_HD void pass_b_store(int j, cplx *c, cplx *buf, const cplx *twB) {
  const int k = j % 12, q = j / 12;
#pragma unroll
  for (int r = 1; r < 16; ++r) c[r] = twt<SIGN>(c[r], twB[r * 12 + k]);
  dft16<SIGN>(c);
#pragma unroll
  for (int r = 0; r < 16; ++r) buf[q * BSTRIDE + k + 12 * r] = c[r];
}
// pass C: thread j < 192; radix 16, ns = 192 (q = 0, k = j); twiddle exp(SIGN 2 pi i j r / 3072) = twC[r * 192 + j]; result
// c[r] = Z[j + 192 r] stays in registers
ASX_HD void pass_c_load(int j, const cplx *buf, cplx *c) {
#pragma unroll
  for (int r = 0; r < 16; ++r) c[r] = buf[r * BSTRIDE + j];
}
template <int SIGN>
ASX_HD void pass_c_compute(int j, cplx *c, const cplx *twC) {
#pragma unroll
  for (int r = 1; r < 16; ++r) c[r] = twt<SIGN>(c[r], twC[r * NB + j]);
  dft16<SIGN>(c);
}

// forward split: X[k] = E + W6144^k O, E = (Z[k] + conj Z[NH - k]) / 2, O = -i (Z[k] - conj Z[NH - k]) / 2, k < NH
ASX_HD cplx split_bin(int k, const cplx *Z, cplx wk) {
  const cplx zk = Z[k], zc = Z[k == 0 ? 0 : NH - k];
  const cplx E = cscale(addc(zk, zc), 0.5f), D = cscale(subc(zk, zc), 0.5f);
  return add_mi(E, cmul(D, wk));                 // E + W (-i D) = E - i (W D)
}
// inverse merge: Z[k] = E + i O, E = (X[k] + conj X[NH - k]) / 2, O = W6144^{-k} (X[k] - conj X[NH - k]) / 2;  xc = X[NH - k]
ASX_HD cplx merge_bin(cplx xk, cplx xc, cplx wk) {
  const cplx E = cscale(addc(xk, xc), 0.5f), D = cscale(subc(xk, xc), 0.5f);
  return add_pi(E, cmulc(D, wk));
}

#ifndef ASX_HOST_TEST
// frames [group_start(g), group_start(g + 1)) belong to workgroup g of a (chunk, channel): T frames in n_groups even shares
__host__ __device__ __forceinline__ int group_start(int g, int T, int n_groups) { return (int)((int64_t)g * T / n_groups); }

// =======================================================================================================================
struct Stft3Args {
  const float *wave;           // song mix [2, N] (n_song >= 0) or [B, 2, C]
  const int64_t *chunk_start;  // [B] padded-domain start of each chunk (song mode)
  int64_t n_song;
  int trim;
  int64_t C;
  int T, dim_f, zero_low;
  float *spec;                 // [B, 4, T, dim_f]
  const float *window;         // [6144] periodic Hann
  const cplx *tw;            // [6144] exp(-2 pi i j / 6144)
  const cplx *twB, *twC;     // [16][12], [16][192]
  float sign;
  int64_t out_bstride;         // floats between batch items (0 = dense)
  int n_groups;                // stft3p_kernel: workgroups per (chunk, channel), each takes an even share of the T frames
};

constexpr int STFT3_LDS_BYTES = LDS_X * 8;
constexpr int ISTFT3_LDS_BYTES = LDS_X * 8 + NFFT * 4;

__global__ __launch_bounds__(256) void stft3_kernel(Stft3Args a) {
  extern __shared__ cplx lds3[];
  cplx *buf = lds3;
  const int t = blockIdx.x, ch = blockIdx.y, b = blockIdx.z;
  const int j = threadIdx.x;
  const int64_t C = a.C;
  const float *src;
  int64_t cstart = 0;
  if (a.n_song >= 0) {
    src = a.wave + (int64_t)ch * a.n_song;
    cstart = a.chunk_start[b];
  } else {
    src = a.wave + ((int64_t)b * 2 + ch) * C;
  }
  // frame element e sits at chunk position q = t * hop + e - n_fft / 2 (torch.stft centre padding, reflected at the chunk
  // ends, stft.py:41); song mode maps chunk position q to mix[cstart + q - trim] or 0 (mdx_separator.py:329-366)
  const int64_t q0 = (int64_t)t * HOP - NH;
  const bool inside_chunk = q0 >= 0 && q0 + NFFT <= C;
  const int64_t s0 = a.n_song >= 0 ? cstart + q0 - a.trim : q0;
  const bool fast = inside_chunk && (a.n_song < 0 || (s0 >= 0 && s0 + NFFT <= a.n_song)) &&
                    ((reinterpret_cast<uintptr_t>(src + s0) & 7) == 0);
  cplx v[12];
  const cplx *w2 = reinterpret_cast<const cplx *>(a.window);
  if (fast) {
    const cplx *s2 = reinterpret_cast<const cplx *>(src + s0);
#pragma unroll
    for (int r = 0; r < 12; ++r) {
      v[r] = emul(s2[j + 256 * r], w2[j + 256 * r]);
    }
  } else {
#pragma unroll
    for (int r = 0; r < 12; ++r) {
      float xe[2];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        int64_t q = q0 + 2 * (j + 256 * r) + h;
        if (q < 0) q = -q;
        if (q >= C) q = 2 * (C - 1) - q;
        if (a.n_song >= 0) {
          const int64_t i = cstart + q - a.trim;
          xe[h] = (i >= 0 && i < a.n_song) ? src[i] : 0.0f;
        } else {
          xe[h] = src[q];
        }
      }
      const cplx w = w2[j + 256 * r];
      v[r] = mk(xe[0] * w.x, xe[1] * w.y);
    }
  }
  cplx c[16];
  pass_a<-1>(j, v, buf);
  __syncthreads();
  if (j < NB) pass_b_load(j, buf, c);
  __syncthreads();
  if (j < NB) pass_b_store<-1>(j, c, buf, a.twB);
  __syncthreads();
  if (j < NB) pass_c_load(j, buf, c);
  __syncthreads();
  if (j < NB) {
    pass_c_compute<-1>(j, c, a.twC);
#pragma unroll
    for (int r = 0; r < 16; ++r) buf[j + NB * r] = c[r];
  }
  __syncthreads();
  const cplx *bufA = buf;
  const int64_t bst = a.out_bstride ? a.out_bstride : (int64_t)4 * a.T * a.dim_f;
  float *re = a.spec + (int64_t)b * bst + ((int64_t)(ch * 2) * a.T + t) * a.dim_f;
  float *im = re + (int64_t)a.T * a.dim_f;
#pragma unroll
  for (int r = 0; r < 12; ++r) {
    const int k = j + 256 * r;
    if (k >= a.dim_f) continue;
    cplx X = mk(0.f, 0.f);
    if (k >= a.zero_low) {
      X = cscale(split_bin(k, bufA, a.tw[k]), a.sign);
    }
    re[k] = X.x;
    im[k] = X.y;
  }
}

// -----------------------------------------------------------------------------------------------------------------------
// stft3p_kernel: a workgroup transforms G consecutive frames of one (chunk, channel).  Consecutive frames share five of
// their six hops, so the samples live in an LDS ring of n_fft floats: per frame only the ONE new hop (4 floats per thread,
// through the reflect / song-position mapping of stft3_kernel) is read from memory, requested one frame ahead.  The
// frame-invariant table values stay on chip as in istft3p_kernel (window pairs and pass-C twiddles in registers, pass-B
// table in LDS, the split twiddle W6144^(j + 256 r) as W6144^j times the constant W24^r).  52 KB of LDS and <= 168 registers:
// three workgroups share a CU.
// Per frame the kernel reads 4 KB and writes the two [dim_f] rows; stft3_kernel re-read the whole 24 KB frame and 75 KB of
// tables through L2.
constexpr int STFT3P_LDS_BYTES = LDS_X * 8 + NFFT * 4 + 16 * 12 * 8;    // 52 KB: three workgroups per CU

__global__ __launch_bounds__(256, 3) void stft3p_kernel(Stft3Args a) {
  extern __shared__ cplx lds3[];
  cplx *buf = lds3;
  float *ring = reinterpret_cast<float *>(lds3 + LDS_X);
  cplx *twBs = reinterpret_cast<cplx *>(ring + NFFT);
  const int g = blockIdx.x, ch = blockIdx.y, b = blockIdx.z;
  const int j = threadIdx.x;
  const int t0 = group_start(g, a.T, a.n_groups), t1 = group_start(g + 1, a.T, a.n_groups) - 1;   // inclusive
  const int64_t C = a.C;
  const float *src;
  int64_t cstart = 0;
  if (a.n_song >= 0) {
    src = a.wave + (int64_t)ch * a.n_song;
    cstart = a.chunk_start[b];
  } else {
    src = a.wave + ((int64_t)b * 2 + ch) * C;
  }
  // hop h of the padded chunk = chunk positions [h * hop - n_fft / 2, + hop); this thread's four samples of it
  auto hop4 = [&](int64_t h) -> float4 {
    const int64_t q0 = h * HOP - NH + 4 * j;
    const int64_t s0 = a.n_song >= 0 ? cstart + q0 - a.trim : q0;
    const bool inside = q0 >= 0 && q0 + 4 <= C && (a.n_song < 0 || (s0 >= 0 && s0 + 4 <= a.n_song));
    if (inside && (reinterpret_cast<uintptr_t>(src + s0) & 15) == 0) return *reinterpret_cast<const float4 *>(src + s0);
    float x[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      int64_t q = q0 + i;
      if (q < 0) q = -q;
      if (q >= C) q = 2 * (C - 1) - q;
      if (a.n_song >= 0) {
        const int64_t p = cstart + q - a.trim;
        x[i] = (p >= 0 && p < a.n_song) ? src[p] : 0.0f;
      } else {
        x[i] = src[q];
      }
    }
    return make_float4(x[0], x[1], x[2], x[3]);
  };
  if (j < 16 * 12) twBs[j] = a.twB[j];
#pragma unroll 1
  for (int d = 0; d < HPF; ++d) reinterpret_cast<float4 *>(ring + ((t0 + d) % HPF) * HOP)[j] = hop4(t0 + d);
  const int jb = j < NB ? j : 0;
  cplx win[12], wC[16];
#pragma unroll
  for (int r = 0; r < 12; ++r) win[r] = reinterpret_cast<const cplx *>(a.window)[j + 256 * r];
#pragma unroll
  for (int r = 1; r < 16; ++r) wC[r] = a.twC[r * NB + jb];
  const cplx twj = a.tw[j];
  const int64_t bst = a.out_bstride ? a.out_bstride : (int64_t)4 * a.T * a.dim_f;
  float *re0 = a.spec + (int64_t)b * bst + (int64_t)(ch * 2) * a.T * a.dim_f;
  const int64_t plane = (int64_t)a.T * a.dim_f;
  __syncthreads();
  for (int t = t0; t <= t1; ++t) {
    float4 nx = make_float4(0.f, 0.f, 0.f, 0.f);
    if (t < t1) nx = hop4((int64_t)t + HPF);                    // the hop frame t + 1 adds
    cplx v[12];
    {
      const cplx *ring2 = reinterpret_cast<const cplx *>(ring);
      const int base = (t % HPF) * (HOP / 2) + j;
#pragma unroll
      for (int r = 0; r < 12; ++r) {
        int pos = base + 256 * r;
        pos = pos >= NH ? pos - NH : pos;
        v[r] = emul(ring2[pos], win[r]);
      }
    }
    pass_a<-1>(j, v, buf);
    __syncthreads();
    cplx c[16];
    if (j < NB) pass_b_load(j, buf, c);
    __syncthreads();
    if (j < NB) pass_b_store<-1>(j, c, buf, twBs);
    __syncthreads();
    if (j < NB) pass_c_load(j, buf, c);
    __syncthreads();
    if (j < NB) {
#pragma unroll
      for (int r = 1; r < 16; ++r) c[r] = cmul(c[r], wC[r]);
      dft16<-1>(c);
#pragma unroll
      for (int r = 0; r < 16; ++r) buf[j + NB * r] = c[r];
    }
    // hop t (the oldest of frame t) is dead since the barrier after pass A: its slot takes the new hop
    if (t < t1) reinterpret_cast<float4 *>(ring + (t % HPF) * HOP)[j] = nx;
    __syncthreads();
    float *re = re0 + (int64_t)t * a.dim_f;
    float *im = re + plane;
    cplx twl = twj;
    asm volatile("" : "+v"(twl));                           // the eleven products below are recomputed per frame, not kept in 22 registers
#pragma unroll
    for (int r = 0; r < 12; ++r) {
      constexpr float C24[12] = {1.f, 0.96592582628906828675f, 0.86602540378443864676f, 0.70710678118654752440f, 0.5f,
                                 0.25881904510252076235f, 0.f, -0.25881904510252076235f, -0.5f, -0.70710678118654752440f,
                                 -0.86602540378443864676f, -0.96592582628906828675f};
      constexpr float S24[12] = {0.f, 0.25881904510252076235f, 0.5f, 0.70710678118654752440f, 0.86602540378443864676f,
                                 0.96592582628906828675f, 1.f, 0.96592582628906828675f, 0.86602540378443864676f,
                                 0.70710678118654752440f, 0.5f, 0.25881904510252076235f};
      const int k = j + 256 * r;
      if (k >= a.dim_f) continue;
      cplx X = mk(0.f, 0.f);
      if (k >= a.zero_low) {
        const cplx wk = r == 0 ? twl : cmulc_k(twl, mk(C24[r], S24[r]));   // W6144^(j + 256 r) = W6144^j W24^r
        X = cscale(split_bin(k, buf, wk), a.sign);
      }
      re[k] = X.x;
      im[k] = X.y;
    }
    __syncthreads();                                    // frame t + 1's pass A rewrites the exchange buffer
  }
}

// -----------------------------------------------------------------------------------------------------------------------
struct Istft3Args {
  const float *spec;       // [B (x2), 4, T, dim_f]
  int T, dim_f;
  int combine;             // 0, or the batch offset of the negated-input pass (denoise: 0.5 * spec[b] - 0.5 * spec[b + combine])
  int64_t in_bstride;      // floats between batch items (0 = dense)
  const float *window;     // [6144]
  const cplx *tw, *twB, *twC;
  const float *env;        // [n_fft + hop * (T - 1)] sum of squared windows
  const int64_t *n_act;    // [B] active length of each chunk (chunk Hann window), < 0: none; nullptr: none
  int64_t C;               // samples per chunk = hop * (T - 1)
  float *out;              // [B, 2, C]
  float *seam;             // [B, 2, n_groups, 2, 5 * hop] partial hops (head, tail) of every frame group
  int n_groups;            // workgroups per (chunk, channel); each takes an even share (>= 5) of the T frames
  const double *hann;      // np.hanning(C) in float64 (the chunk window of a full-length chunk), or nullptr
};

__device__ __forceinline__ double hanning3_f64(int64_t j, int64_t M) {
  if (M == 1) return 1.0;
  return 0.5 + 0.5 * cos(3.14159265358979323846 * (double)(2 * j + 1 - M) / (double)(M - 1));
}

// np.hanning(M) in float64, element by element the same expression emit_hop evaluates for a chunk of any other length
__global__ void hann3_table_kernel(int64_t M, double *out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < M) out[i] = hanning3_f64(i, M);
}

// finish one complete hop: acc / env, chunk window, write.  With C = hop * (T - 1) (the launcher's precondition) a hop lies
// either wholly inside the chunk (3 <= h <= T + 1) or wholly in the stripped centre padding, so the four samples of a lane
// move as one 16-byte access.
__device__ __forceinline__ void emit_hop(const Istft3Args &a, int b, int ch, int64_t h, const float4 v, int lane4) {
  if (h < HPF / 2 || h > (int64_t)a.T + 1) return;
  const int64_t m0 = h * HOP + 4 * lane4;      // padded-domain position of v.x
  const int64_t jo = m0 - NH;
  const float4 e = *reinterpret_cast<const float4 *>(a.env + m0);
  float y[4] = {v.x / e.x, v.y / e.y, v.z / e.z, v.w / e.w};
  const int64_t na = a.n_act ? a.n_act[b] : -1;
  if (na >= 0) {
    if (na == a.C && a.hann) {
      const double2 h0 = *reinterpret_cast<const double2 *>(a.hann + jo), h1 = *reinterpret_cast<const double2 *>(a.hann + jo + 2);
      y[0] = (float)((double)y[0] * h0.x);
      y[1] = (float)((double)y[1] * h0.y);
      y[2] = (float)((double)y[2] * h1.x);
      y[3] = (float)((double)y[3] * h1.y);
    } else {
#pragma unroll 1
      for (int i = 0; i < 4; ++i) y[i] = jo + i < na ? (float)((double)y[i] * hanning3_f64(jo + i, na)) : 0.f;
    }
  }
  *reinterpret_cast<float4 *>(a.out + ((int64_t)b * 2 + ch) * a.C + jo) = make_float4(y[0], y[1], y[2], y[3]);
}

// the same in two halves for istft3p_kernel: the table values a hop needs (window envelope, float64 chunk window) are
// requested when the frame starts and used when it ends, so their L2 latency is off the frame's critical path
struct EmitPre {
  float4 env;
  double2 h0, h1;
  int mode;              // 0: hop outside the chunk, 1: envelope only, 2: envelope + Hann table, 3: envelope + computed Hann
};
__device__ __forceinline__ EmitPre emit_prefetch(const Istft3Args &a, int64_t na, int64_t h, int lane4) {
  EmitPre p;
  p.mode = 0;
  p.env = make_float4(1.f, 1.f, 1.f, 1.f);
  p.h0 = p.h1 = make_double2(0.0, 0.0);
  if (h < HPF / 2 || h > (int64_t)a.T + 1) return p;
  const int64_t m0 = h * HOP + 4 * lane4;
  p.env = *reinterpret_cast<const float4 *>(a.env + m0);
  p.mode = na < 0 ? 1 : ((na == a.C && a.hann) ? 2 : 3);
  if (p.mode == 2) {
    p.h0 = *reinterpret_cast<const double2 *>(a.hann + (m0 - NH));
    p.h1 = *reinterpret_cast<const double2 *>(a.hann + (m0 - NH) + 2);
  }
  return p;
}
__device__ __forceinline__ void emit_finish(const Istft3Args &a, const EmitPre &p, int64_t na, int b, int ch, int64_t h, const float4 v,
                                            int lane4) {
  if (p.mode == 0) return;
  const int64_t jo = h * HOP + 4 * lane4 - NH;
  float y[4] = {v.x / p.env.x, v.y / p.env.y, v.z / p.env.z, v.w / p.env.w};
  if (p.mode == 2) {
    y[0] = (float)((double)y[0] * p.h0.x);
    y[1] = (float)((double)y[1] * p.h0.y);
    y[2] = (float)((double)y[2] * p.h1.x);
    y[3] = (float)((double)y[3] * p.h1.y);
  } else if (p.mode == 3) {
#pragma unroll 1
    for (int i = 0; i < 4; ++i) y[i] = jo + i < na ? (float)((double)y[i] * hanning3_f64(jo + i, na)) : 0.f;
  }
  *reinterpret_cast<float4 *>(a.out + ((int64_t)b * 2 + ch) * a.C + jo) = make_float4(y[0], y[1], y[2], y[3]);
}

__global__ __launch_bounds__(256, 2) void istft3_kernel(Istft3Args a) {
  extern __shared__ cplx lds3[];
  cplx *buf = lds3;
  float *ring = reinterpret_cast<float *>(lds3 + LDS_X);
  const int g = blockIdx.x, ch = blockIdx.y, b = blockIdx.z;
  const int j = threadIdx.x;
  const int t0 = group_start(g, a.T, a.n_groups), t1 = group_start(g + 1, a.T, a.n_groups) - 1;   // inclusive
  for (int i = j; i < NFFT; i += 256) ring[i] = 0.f;
  const int64_t bst = a.in_bstride ? a.in_bstride : (int64_t)4 * a.T * a.dim_f;
  const cplx *w2 = reinterpret_cast<const cplx *>(a.window);
  float *seam_head = a.seam + ((((int64_t)b * 2 + ch) * a.n_groups + g) * 2) * (5 * HOP);
  float *seam_tail = seam_head + 5 * HOP;
  for (int t = t0; t <= t1; ++t) {
    const float *re = a.spec + (int64_t)b * bst + ((int64_t)(ch * 2) * a.T + t) * a.dim_f;
    const float *im = re + (int64_t)a.T * a.dim_f;
    const float *re2 = re + (int64_t)a.combine * bst, *im2 = im + (int64_t)a.combine * bst;
    auto bin = [&](int k) -> cplx {
      if (k >= a.dim_f) return mk(0.f, 0.f);         // bins >= dim_f (incl. Nyquist) are zero (stft.py:58-68)
      cplx x = mk(re[k], im[k]);
      if (a.combine) {
        x.x = re2[k] * -0.5f + x.x * 0.5f;
        x.y = im2[k] * -0.5f + x.y * 0.5f;
      }
      if (k == 0) x.y = 0.f;                                    // c2r: the imaginary part of DC is ignored
      return x;
    };
    cplx v[12];
#pragma unroll
    for (int r = 0; r < 12; ++r) {
      const int k = j + 256 * r;
      v[r] = merge_bin(bin(k), bin(NH - k), a.tw[k]);           // k = 0 pairs with the (zero) Nyquist bin NH
    }
    pass_a<+1>(j, v, buf);
    __syncthreads();
    cplx c[16];
    if (j < NB) pass_b_load(j, buf, c);
    __syncthreads();
    if (j < NB) pass_b_store<+1>(j, c, buf, a.twB);
    __syncthreads();
    if (j < NB) {
      pass_c_load(j, buf, c);
      pass_c_compute<+1>(j, c, a.twC);
      const float scale = 1.0f / (float)NH;
      cplx *ring2 = reinterpret_cast<cplx *>(ring);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = j + NB * r;                                // samples 2m, 2m + 1 of the frame
        const cplx w = w2[m];
        const int pos = (t * (HOP / 2) + m) % NH;                // ring slot (cplx units): frame t starts at hop t
        cplx acc = ring2[pos];
        acc = ca(acc, emul(cscale(c[r], scale), w));
        ring2[pos] = acc;
      }
    }
    __syncthreads();
    // hop t is complete when every frame t-5 .. t has been added: frames before t0 are missing for the first five hops of the
    // group (unless the group starts the chunk), they travel as partial sums
    {
      float4 *slot = reinterpret_cast<float4 *>(ring + (t % HPF) * HOP) + j;
      const float4 v4 = *slot;
      *slot = make_float4(0.f, 0.f, 0.f, 0.f);
      if (t - t0 >= HPF - 1 || t0 == 0) emit_hop(a, b, ch, t, v4, j);
      else reinterpret_cast<float4 *>(seam_head + (t - t0) * HOP)[j] = v4;
    }
    // the ring slot is re-used by frame t + 1 only after three more barriers; the exchange buffer is rewritten by frame
    // t + 1's pass A only after the barrier above, when every pass-C read of frame t has completed
  }
  __syncthreads();
  // hops t1 + 1 .. t1 + 5 still miss the frames of the next group (or are final when this group ends the chunk)
  for (int d = 1; d < HPF; ++d) {
    const int64_t h = (int64_t)t1 + d;
    const float4 v4 = reinterpret_cast<const float4 *>(ring + (h % HPF) * HOP)[j];
    if (t1 == a.T - 1) emit_hop(a, b, ch, h, v4, j);
    else reinterpret_cast<float4 *>(seam_tail + (d - 1) * HOP)[j] = v4;
  }
}

// -----------------------------------------------------------------------------------------------------------------------
// istft3p_kernel: the frame loop of istft3_kernel with nothing but LDS, registers and barriers on its critical path.
//   * the spectrum rows of frame t + 1 (re, im: 2 x dim_f floats) are copied global -> LDS by `global_load_lds_dwordx4`
//     while frame t is being transformed (issued after the barrier that ends frame t's merge reads, awaited before the
//     barrier that ends frame t) -- no staging registers, and the merge reads X[k] / X[3072 - k] from LDS;
//   * the frame-invariant table values stay on chip: the 15 pass-C twiddles and the 16 window pairs (pre-scaled by
//     1 / 3072) of a thread in registers, the pass-B table (1.5 KB) in LDS, the merge twiddle W6144^(j + 256 r) as W6144^j
//     (one register pair) times the constant W24^r; the envelope / chunk-window values of the hop a frame completes are
//     requested when the frame starts.
// Same arithmetic as istft3_kernel except for the factored merge twiddle and the pre-scaled window (one fp32 rounding each).  Preconditions (checked
// by the launcher, which otherwise runs istft3_kernel): combine == 0, dim_f % 4 == 0, 16-byte aligned rows.
#define F3_GLDS16(gptr, lptr)                                                               \
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(gptr), \
                                   (__attribute__((address_space(3))) void *)(lptr), 16, 0, 0)

constexpr int ISTFT3P_LDS_BYTES = LDS_X * 8 + NFFT * 4 + 2 * NH * 4 + 16 * 12 * 8;

template <int ABL>   // 0 = product; 1: no spectrum DMA after the first frame, 2: no hop emit, 3: both (timing probes only)
__global__ __launch_bounds__(256, 2) void istft3p_kernel(Istft3Args a) {
  extern __shared__ cplx lds3[];
  cplx *buf = lds3;
  float *ring = reinterpret_cast<float *>(lds3 + LDS_X);
  float *st_re = ring + NFFT, *st_im = st_re + NH;
  cplx *twBs = reinterpret_cast<cplx *>(st_im + NH);              // [16][12] pass-B twiddles
  const int g = blockIdx.x, ch = blockIdx.y, b = blockIdx.z;
  const int j = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(j >> 6), lane = j & 63;
  const int t0 = group_start(g, a.T, a.n_groups), t1 = group_start(g + 1, a.T, a.n_groups) - 1;   // inclusive
  for (int i = j; i < NFFT + 2 * NH; i += 256) ring[i] = 0.f;          // ring + both stage rows (bins >= dim_f stay zero)
  if (j < 16 * 12) {
    twBs[j] = a.twB[j];
  }
  const int64_t bst = a.in_bstride ? a.in_bstride : (int64_t)4 * a.T * a.dim_f;
  const float *re0 = a.spec + (int64_t)b * bst + (int64_t)(ch * 2) * a.T * a.dim_f;
  const int64_t plane = (int64_t)a.T * a.dim_f;
  float *seam_head = a.seam + ((((int64_t)b * 2 + ch) * a.n_groups + g) * 2) * (5 * HOP);
  float *seam_tail = seam_head + 5 * HOP;
  const int jb = j < NB ? j : 0;
  // frame-invariant registers: W6144^j (the merge twiddle of bin j + 256 r is W6144^j * W24^r) and the pass-C twiddles
  const cplx twj = a.tw[j];
  cplx wC[16];
#pragma unroll
  for (int r = 1; r < 16; ++r) {
    wC[r] = a.twC[r * NB + jb];
  }
  cplx win[16];                                                       // this thread's window pairs, pre-scaled by 1 / 3072
#pragma unroll
  for (int r = 0; r < 16; ++r) win[r] = cscale(reinterpret_cast<const cplx *>(a.window)[jb + NB * r], 1.0f / (float)NH);
  const int64_t na = a.n_act ? a.n_act[b] : -1;
  auto issue = [&](int t) {
    const float *re = re0 + (int64_t)t * a.dim_f;
    const float *im = re + plane;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int base = (wave * 3 + i) * 256;           // first float of this wave-issue
      const int f = base + lane * 4;
      if (f < a.dim_f) {
        F3_GLDS16(re + f, st_re + base);
        F3_GLDS16(im + f, st_im + base);
      }
    }
  };
  __syncthreads();                                      // the zero fill precedes the first DMA
  issue(t0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int t = t0; t <= t1; ++t) {
    const bool emits = (t - t0 >= HPF - 1 || t0 == 0) && !(ABL & 2);              // else hop t still misses frames of the previous group
    const EmitPre pre = emits ? emit_prefetch(a, na, t, j) : EmitPre{};
    cplx v[12];
    cplx twl = twj;
    asm volatile("" : "+v"(twl.x), "+v"(twl.y));           // the eleven products below are recomputed per frame, not kept in 22 registers
#pragma unroll
    for (int r = 0; r < 12; ++r) {
      // W24^r = exp(-2 pi i r / 24)
      constexpr float C24[12] = {1.f, 0.96592582628906828675f, 0.86602540378443864676f, 0.70710678118654752440f, 0.5f,
                                 0.25881904510252076235f, 0.f, -0.25881904510252076235f, -0.5f, -0.70710678118654752440f,
                                 -0.86602540378443864676f, -0.96592582628906828675f};
      constexpr float S24[12] = {0.f, 0.25881904510252076235f, 0.5f, 0.70710678118654752440f, 0.86602540378443864676f,
                                 0.96592582628906828675f, 1.f, 0.96592582628906828675f, 0.86602540378443864676f,
                                 0.70710678118654752440f, 0.5f, 0.25881904510252076235f};
      const int k = j + 256 * r;
      cplx xk = mk(st_re[k], st_im[k]);
      cplx xc = mk(0.f, 0.f);                        // k = 0 pairs with the (zero) Nyquist bin
      if (k != 0) xc = mk(st_re[NH - k], st_im[NH - k]);
      else xk.y = 0.f;                                          // c2r: the imaginary part of DC is ignored
      const cplx wk = r == 0 ? twl : cmulc_k(twl, mk(C24[r], S24[r]));
      v[r] = merge_bin(xk, xc, wk);
    }
    pass_a<+1>(j, v, buf);
    __syncthreads();                                    // pass A visible; every merge read of the stage rows is done
    if (t < t1 && !(ABL & 1)) issue(t + 1);
    cplx c[16];
    if (j < NB) pass_b_load(j, buf, c);
    __syncthreads();
    if (j < NB) {
      const int k = j % 12, q = j / 12;
#pragma unroll
      for (int r = 1; r < 16; ++r) c[r] = cmulc(c[r], twBs[r * 12 + k]);
      dft16<+1>(c);
#pragma unroll
      for (int r = 0; r < 16; ++r) buf[q * BSTRIDE + k + 12 * r] = c[r];
    }
    __syncthreads();
    if (j < NB) {
      pass_c_load(j, buf, c);
#pragma unroll
      for (int r = 1; r < 16; ++r) c[r] = cmulc(c[r], wC[r]);
      dft16<+1>(c);
      cplx *ring2 = reinterpret_cast<cplx *>(ring);
      const int base = (t % HPF) * (HOP / 2) + j;               // ring slot (cplx units) of sample pair m = j
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        int pos = base + NB * r;
        pos = pos >= NH ? pos - NH : pos;
        cplx acc = ring2[pos];
        acc = ca(acc, emul(c[r], win[r]));
        ring2[pos] = acc;
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // this thread's share of frame t + 1 has landed
    __syncthreads();
    {
      float4 *slot = reinterpret_cast<float4 *>(ring + (t % HPF) * HOP) + j;
      const float4 v4 = *slot;
      *slot = make_float4(0.f, 0.f, 0.f, 0.f);
      if (emits) emit_finish(a, pre, na, b, ch, t, v4, j);
      else if (!(ABL & 2) || t - t0 < HPF - 1) reinterpret_cast<float4 *>(seam_head + (t - t0) * HOP)[j] = v4;
    }
  }
  __syncthreads();
  for (int d = 1; d < HPF; ++d) {
    const int64_t h = (int64_t)t1 + d;
    const float4 v4 = reinterpret_cast<const float4 *>(ring + (h % HPF) * HOP)[j];
    if (t1 == a.T - 1) emit_hop(a, b, ch, h, v4, j);
    else reinterpret_cast<float4 *>(seam_tail + (d - 1) * HOP)[j] = v4;
  }
}

// fold the seams: hop t0(g + 1) + d (d < 5) = tail_g[d] + head_{g+1}[d].  grid (5 * HOP / 1024, n_groups - 1, B * 2)
__global__ __launch_bounds__(256) void seam3_kernel(Istft3Args a) {
  const int g = blockIdx.y;                       // seam between group g and g + 1
  const int bc = blockIdx.z, b = bc >> 1, ch = bc & 1;
  const int d = blockIdx.x;
  const int lane4 = threadIdx.x;
  const float *tail = a.seam + ((((int64_t)b * 2 + ch) * a.n_groups + g) * 2 + 1) * (5 * HOP);
  const float *head = a.seam + ((((int64_t)b * 2 + ch) * a.n_groups + g + 1) * 2) * (5 * HOP);
  const int t0n = group_start(g + 1, a.T, a.n_groups);   // every group holds >= 5 frames, so all five head hops exist
  const float4 x = reinterpret_cast<const float4 *>(tail + d * HOP)[lane4];
  const float4 y = reinterpret_cast<const float4 *>(head + d * HOP)[lane4];
  emit_hop(a, b, ch, (int64_t)t0n + d, make_float4(x.x + y.x, x.y + y.y, x.z + y.z, x.w + y.w), lane4);
}
#endif  // ASX_HOST_TEST

}  // namespace f3
}  // namespace asx
