// Fast STFT / iSTFT path for n_fft = 6144, hop = 1024 (the UVR-MDX-NET HQ geometry): a real FFT of 6144 points as one
// complex FFT of 3072 = 12 x 16 x 16 points in THREE register-resident Stockham passes (radix 12, 16, 16) with two LDS
// exchanges, instead of the generic six-pass radix-{4,3} loop of kernels_fft.h.  Replaces, on that geometry,
// torch.stft / torch.istft as used by uvr_lib_v5/stft.py:41,117 and the frame overlap-add inside torch.istft.
//
// Forward (stft3_kernel), one workgroup of 256 threads per (frame, channel, chunk):
//   load + periodic-Hann window straight into registers: thread j holds z[j + 256 r], r < 12 (z[m] = x[2m] + i x[2m+1]);
//   pass A: 12-point DFT in registers (3 x radix-4, W12 twiddles, 4 x radix-3), written as six 16-byte stores per thread
//           (row stride 96 B: conflict-free for ds_write_b128's 8-lane groups);
//   pass B: threads j < 192 read in[j + 192 r] (consecutive lanes, conflict-free ds_read_b64), twiddle by W192^(k r),
//           16-point DFT (radix-4 x radix-4), write out[q*192 + k + 12 r] into blocks padded to 204 so that a 16-lane
//           ds_write_b64 group never wraps onto its own banks;
//   pass C: same read pattern, twiddle W3072^(j r) (coalesced table [r][j]), 16-point DFT -> Z[j + 192 r];
//   split : Z goes to LDS once more so that X[k] = E + W6144^k O can pair Z[k] with conj Z[3072 - k]; the two planes of a
//           [T, F] row are written coalesced.
// Inverse (istft3_kernel): a workgroup owns G consecutive frames of one (chunk, channel).  Per frame: X[k] and X[3072-k]
//   are read straight from the spectrogram (coalesced, ascending / descending), merged to Z, the same three passes run
//   with conjugate twiddles, and the windowed frame is ACCUMULATED INTO AN LDS RING of n_fft floats (a frame covers six
//   hops; after frame t hop t is complete, is divided by the window envelope, multiplied by the chunk's Hann window and
//   written out).  The [B, 2, T, n_fft] frame buffer of the generic path (12.6 MB per chunk written and re-read) never
//   exists; only the five partial hops at either end of a workgroup's frame range travel through a small seam buffer,
//   which seam3_kernel folds (tail of group g + head of group g + 1) in a fixed order -- deterministic, no atomics.
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

ASX_HD float2 cm(float2 a, float2 b) { return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
ASX_HD float2 ca(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
ASX_HD float2 cs(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
// multiply by -i (SIGN < 0, forward) or +i (SIGN > 0, inverse)
template <int SIGN>
ASX_HD float2 rot(float2 a) { return SIGN < 0 ? make_float2(a.y, -a.x) : make_float2(-a.y, a.x); }
// multiply by exp(SIGN * i * theta) given (cos theta, sin theta)
template <int SIGN>
ASX_HD float2 tw(float2 a, float c, float s) {
  return SIGN < 0 ? make_float2(a.x * c + a.y * s, a.y * c - a.x * s) : make_float2(a.x * c - a.y * s, a.y * c + a.x * s);
}

template <int SIGN>
ASX_HD void dft4(float2 &a, float2 &b, float2 &c, float2 &d) {
  const float2 t0 = ca(a, c), t1 = cs(a, c), t2 = ca(b, d), t3 = rot<SIGN>(cs(b, d));
  a = ca(t0, t2);
  b = ca(t1, t3);
  c = cs(t0, t2);
  d = cs(t1, t3);
}
template <int SIGN>
ASX_HD void dft3(float2 &a, float2 &b, float2 &c) {
  const float s3 = 0.86602540378443864676f;
  const float2 s = ca(b, c), d = cs(b, c);
  const float2 m = make_float2(a.x - 0.5f * s.x, a.y - 0.5f * s.y);
  const float2 r = rot<SIGN>(make_float2(s3 * d.x, s3 * d.y));
  a = ca(a, s);
  b = ca(m, r);
  c = cs(m, r);
}

// 12-point DFT, natural order in and out: n = 3 n1 + n2, k = k1 + 4 k2
template <int SIGN>
ASX_HD void dft12(float2 *v) {
  const float C1 = 0.86602540378443864676f, S1 = 0.5f;               // W12^1 = cos 30, sin 30
  const float C2 = 0.5f, S2 = 0.86602540378443864676f;               // W12^2
#pragma unroll
  for (int n2 = 0; n2 < 3; ++n2) dft4<SIGN>(v[n2], v[n2 + 3], v[n2 + 6], v[n2 + 9]);   // k1 = 0..3 at v[n2 + 3 k1]
  // W12^(n2 k1): n2 = 1: k1 = 1, 2, 3 -> W^1, W^2, W^3 (= -+i); n2 = 2: k1 = 1, 2, 3 -> W^2, W^4, W^6 (= -1)
  v[1 + 3] = tw<SIGN>(v[1 + 3], C1, S1);
  v[1 + 6] = tw<SIGN>(v[1 + 6], C2, S2);
  v[1 + 9] = rot<SIGN>(v[1 + 9]);
  v[2 + 3] = tw<SIGN>(v[2 + 3], C2, S2);
  v[2 + 6] = tw<SIGN>(v[2 + 6], -C2, S2);
  v[2 + 9] = make_float2(-v[2 + 9].x, -v[2 + 9].y);
#pragma unroll
  for (int k1 = 0; k1 < 4; ++k1) dft3<SIGN>(v[3 * k1], v[3 * k1 + 1], v[3 * k1 + 2]);  // X[k1 + 4 k2] at v[3 k1 + k2]
  // un-permute: slot 3 k1 + k2 holds X[k1 + 4 k2]
  float2 o[12];
#pragma unroll
  for (int k1 = 0; k1 < 4; ++k1)
#pragma unroll
    for (int k2 = 0; k2 < 3; ++k2) o[k1 + 4 * k2] = v[3 * k1 + k2];
#pragma unroll
  for (int i = 0; i < 12; ++i) v[i] = o[i];
}

// 16-point DFT, natural order in and out: n = 4 n1 + n2, k = k1 + 4 k2
template <int SIGN>
ASX_HD void dft16(float2 *v) {
  const float C1 = 0.92387953251128675613f, S1 = 0.38268343236508977173f;   // W16^1
  const float C2 = 0.70710678118654752440f;                                  // W16^2 (cos = sin)
#pragma unroll
  for (int n2 = 0; n2 < 4; ++n2) dft4<SIGN>(v[n2], v[n2 + 4], v[n2 + 8], v[n2 + 12]);  // k1 at v[n2 + 4 k1]
  // W16^(n2 k1)
  v[1 + 4] = tw<SIGN>(v[1 + 4], C1, S1);       // 1
  v[1 + 8] = tw<SIGN>(v[1 + 8], C2, C2);       // 2
  v[1 + 12] = tw<SIGN>(v[1 + 12], S1, C1);     // 3
  v[2 + 4] = tw<SIGN>(v[2 + 4], C2, C2);       // 2
  v[2 + 8] = rot<SIGN>(v[2 + 8]);              // 4
  v[2 + 12] = tw<SIGN>(v[2 + 12], -C2, C2);    // 6
  v[3 + 4] = tw<SIGN>(v[3 + 4], S1, C1);       // 3
  v[3 + 8] = tw<SIGN>(v[3 + 8], -C2, C2);      // 6
  v[3 + 12] = tw<SIGN>(v[3 + 12], -C1, -S1);   // 9
#pragma unroll
  for (int k1 = 0; k1 < 4; ++k1) dft4<SIGN>(v[4 * k1], v[4 * k1 + 1], v[4 * k1 + 2], v[4 * k1 + 3]);  // X[k1 + 4 k2] at v[4 k1 + k2]
  float2 o[16];
#pragma unroll
  for (int k1 = 0; k1 < 4; ++k1)
#pragma unroll
    for (int k2 = 0; k2 < 4; ++k2) o[k1 + 4 * k2] = v[4 * k1 + k2];
#pragma unroll
  for (int i = 0; i < 16; ++i) v[i] = o[i];
}

// ---- the three passes, per thread --------------------------------------------------------------------------------------
// pass A: thread j < 256 holds in[j + 256 r] (r < 12); Stockham radix 12, ns = 1: out[12 j + r]
template <int SIGN>
ASX_HD void pass_a(int j, float2 *a, float2 *bufA) {
  dft12<SIGN>(a);
  float4 *p = reinterpret_cast<float4 *>(bufA + 12 * j);
#pragma unroll
  for (int s = 0; s < 6; ++s) p[s] = make_float4(a[2 * s].x, a[2 * s].y, a[2 * s + 1].x, a[2 * s + 1].y);
}
// pass B: thread j < 192; radix 16, ns = 12: k = j % 12, q = j / 12; twiddle exp(SIGN 2 pi i k r / 192) = twB[r * 12 + k]
// (table holds the forward sign; the inverse conjugates); out[q * 192 + k + 12 r] in blocks of BSTRIDE.  Load and store are
// separate calls: with a barrier between them one LDS buffer of LDS_X elements serves every exchange.
ASX_HD void pass_b_load(int j, const float2 *buf, float2 *c) {
#pragma unroll
  for (int r = 0; r < 16; ++r) c[r] = buf[j + NB * r];
}
template <int SIGN>
ASX_HD void pass_b_store(int j, float2 *c, float2 *buf, const float2 *twB) {
  const int k = j % 12, q = j / 12;
#pragma unroll
  for (int r = 1; r < 16; ++r) {
    float2 w = twB[r * 12 + k];
    if (SIGN > 0) w.y = -w.y;
    c[r] = cm(c[r], w);
  }
  dft16<SIGN>(c);
#pragma unroll
  for (int r = 0; r < 16; ++r) buf[q * BSTRIDE + k + 12 * r] = c[r];
}
// pass C: thread j < 192; radix 16, ns = 192 (q = 0, k = j); twiddle exp(SIGN 2 pi i j r / 3072) = twC[r * 192 + j]; result
// c[r] = Z[j + 192 r] stays in registers
ASX_HD void pass_c_load(int j, const float2 *buf, float2 *c) {
#pragma unroll
  for (int r = 0; r < 16; ++r) c[r] = buf[r * BSTRIDE + j];
}
template <int SIGN>
ASX_HD void pass_c_compute(int j, float2 *c, const float2 *twC) {
#pragma unroll
  for (int r = 1; r < 16; ++r) {
    float2 w = twC[r * NB + j];
    if (SIGN > 0) w.y = -w.y;
    c[r] = cm(c[r], w);
  }
  dft16<SIGN>(c);
}

// forward split: X[k] = E + W6144^k O, E = (Z[k] + conj Z[NH - k]) / 2, O = -i (Z[k] - conj Z[NH - k]) / 2, k < NH
ASX_HD float2 split_bin(int k, const float2 *Z, float2 wk) {
  const float2 zk = Z[k];
  float2 zc = Z[k == 0 ? 0 : NH - k];
  zc.y = -zc.y;
  const float2 E = make_float2(0.5f * (zk.x + zc.x), 0.5f * (zk.y + zc.y));
  const float2 D = make_float2(0.5f * (zk.x - zc.x), 0.5f * (zk.y - zc.y));
  const float2 O = make_float2(D.y, -D.x);
  return ca(E, cm(wk, O));
}
// inverse merge: Z[k] = E + i O, E = (X[k] + conj X[NH - k]) / 2, O = W6144^{-k} (X[k] - conj X[NH - k]) / 2;  xc = X[NH - k]
ASX_HD float2 merge_bin(float2 xk, float2 xc, float2 wk) {
  xc.y = -xc.y;
  const float2 E = make_float2(0.5f * (xk.x + xc.x), 0.5f * (xk.y + xc.y));
  const float2 D = make_float2(0.5f * (xk.x - xc.x), 0.5f * (xk.y - xc.y));
  wk.y = -wk.y;
  const float2 O = cm(wk, D);
  return make_float2(E.x - O.y, E.y + O.x);
}

#ifndef ASX_HOST_TEST
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
  const float2 *tw;            // [6144] exp(-2 pi i j / 6144)
  const float2 *twB, *twC;     // [16][12], [16][192]
  float sign;
  int64_t out_bstride;         // floats between batch items (0 = dense)
};

constexpr int STFT3_LDS_BYTES = LDS_X * 8;
constexpr int ISTFT3_LDS_BYTES = LDS_X * 8 + NFFT * 4;

__global__ __launch_bounds__(256) void stft3_kernel(Stft3Args a) {
  extern __shared__ float2 lds3[];
  float2 *buf = lds3;
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
  const bool fast = inside_chunk && (a.n_song < 0 || (s0 >= 0 && s0 + NFFT <= a.n_song)) && ((s0 & 1) == 0);
  float2 v[12];
  const float2 *w2 = reinterpret_cast<const float2 *>(a.window);
  if (fast) {
    const float2 *s2 = reinterpret_cast<const float2 *>(src + s0);
#pragma unroll
    for (int r = 0; r < 12; ++r) {
      const float2 x = s2[j + 256 * r], w = w2[j + 256 * r];
      v[r] = make_float2(x.x * w.x, x.y * w.y);
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
      const float2 w = w2[j + 256 * r];
      v[r] = make_float2(xe[0] * w.x, xe[1] * w.y);
    }
  }
  float2 c[16];
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
  const float2 *bufA = buf;
  const int64_t bst = a.out_bstride ? a.out_bstride : (int64_t)4 * a.T * a.dim_f;
  float *re = a.spec + (int64_t)b * bst + ((int64_t)(ch * 2) * a.T + t) * a.dim_f;
  float *im = re + (int64_t)a.T * a.dim_f;
#pragma unroll
  for (int r = 0; r < 12; ++r) {
    const int k = j + 256 * r;
    if (k >= a.dim_f) continue;
    float2 X = make_float2(0.f, 0.f);
    if (k >= a.zero_low) {
      X = split_bin(k, bufA, a.tw[k]);
      X.x *= a.sign;
      X.y *= a.sign;
    }
    re[k] = X.x;
    im[k] = X.y;
  }
}

// -----------------------------------------------------------------------------------------------------------------------
struct Istft3Args {
  const float *spec;       // [B (x2), 4, T, dim_f]
  int T, dim_f;
  int combine;             // 0, or the batch offset of the negated-input pass (denoise: 0.5 * spec[b] - 0.5 * spec[b + combine])
  int64_t in_bstride;      // floats between batch items (0 = dense)
  const float *window;     // [6144]
  const float2 *tw, *twB, *twC;
  const float *env;        // [n_fft + hop * (T - 1)] sum of squared windows
  const int64_t *n_act;    // [B] active length of each chunk (chunk Hann window), < 0: none; nullptr: none
  int64_t C;               // samples per chunk = hop * (T - 1)
  float *out;              // [B, 2, C]
  float *seam;             // [B, 2, n_groups, 2, 5 * hop] partial hops (head, tail) of every frame group
  int G, n_groups;         // frames per workgroup (>= 5), groups per (chunk, channel) = max(1, T / G)
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

// finish one complete hop: acc / env, chunk window, write (positions outside [0, C) belong to the stripped centre padding)
__device__ __noinline__ void emit_hop(const Istft3Args &a, int b, int ch, int64_t h, const float4 v, int lane4) {
  const int64_t m0 = h * HOP + 4 * lane4;      // padded-domain position of v.x
  const float vv[4] = {v.x, v.y, v.z, v.w};
  const int64_t na = a.n_act ? a.n_act[b] : -1;
#pragma unroll 1
  for (int i = 0; i < 4; ++i) {
    const int64_t m = m0 + i, jo = m - NH;
    if (jo < 0 || jo >= a.C) continue;
    float y = vv[i] / a.env[m];
    if (na >= 0) y = jo < na ? (float)((double)y * ((na == a.C && a.hann) ? a.hann[jo] : hanning3_f64(jo, na))) : 0.f;
    a.out[((int64_t)b * 2 + ch) * a.C + jo] = y;
  }
}

__global__ __launch_bounds__(256, 2) void istft3_kernel(Istft3Args a) {
  extern __shared__ float2 lds3[];
  float2 *buf = lds3;
  float *ring = reinterpret_cast<float *>(lds3 + LDS_X);
  const int g = blockIdx.x, ch = blockIdx.y, b = blockIdx.z;
  const int j = threadIdx.x;
  const int t0 = g * a.G;
  const int t1 = (g == a.n_groups - 1) ? a.T - 1 : t0 + a.G - 1;   // inclusive; the last group takes the remainder (n_groups = max(1, T / G))
  for (int i = j; i < NFFT; i += 256) ring[i] = 0.f;
  const int64_t bst = a.in_bstride ? a.in_bstride : (int64_t)4 * a.T * a.dim_f;
  const float2 *w2 = reinterpret_cast<const float2 *>(a.window);
  float *seam_head = a.seam + ((((int64_t)b * 2 + ch) * a.n_groups + g) * 2) * (5 * HOP);
  float *seam_tail = seam_head + 5 * HOP;
  for (int t = t0; t <= t1; ++t) {
    const float *re = a.spec + (int64_t)b * bst + ((int64_t)(ch * 2) * a.T + t) * a.dim_f;
    const float *im = re + (int64_t)a.T * a.dim_f;
    const float *re2 = re + (int64_t)a.combine * bst, *im2 = im + (int64_t)a.combine * bst;
    auto bin = [&](int k) -> float2 {
      if (k >= a.dim_f) return make_float2(0.f, 0.f);         // bins >= dim_f (incl. Nyquist) are zero (stft.py:58-68)
      float2 x = make_float2(re[k], im[k]);
      if (a.combine) {
        x.x = re2[k] * -0.5f + x.x * 0.5f;
        x.y = im2[k] * -0.5f + x.y * 0.5f;
      }
      if (k == 0) x.y = 0.f;                                    // c2r: the imaginary part of DC is ignored
      return x;
    };
    float2 v[12];
#pragma unroll
    for (int r = 0; r < 12; ++r) {
      const int k = j + 256 * r;
      v[r] = merge_bin(bin(k), bin(NH - k), a.tw[k]);           // k = 0 pairs with the (zero) Nyquist bin NH
    }
    pass_a<+1>(j, v, buf);
    __syncthreads();
    float2 c[16];
    if (j < NB) pass_b_load(j, buf, c);
    __syncthreads();
    if (j < NB) pass_b_store<+1>(j, c, buf, a.twB);
    __syncthreads();
    if (j < NB) {
      pass_c_load(j, buf, c);
      pass_c_compute<+1>(j, c, a.twC);
      const float scale = 1.0f / (float)NH;
      float2 *ring2 = reinterpret_cast<float2 *>(ring);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = j + NB * r;                                // samples 2m, 2m + 1 of the frame
        const float2 w = w2[m];
        const int pos = (t * (HOP / 2) + m) % NH;                // ring slot (float2 units): frame t starts at hop t
        float2 acc = ring2[pos];
        acc.x += (c[r].x * scale) * w.x;
        acc.y += (c[r].y * scale) * w.y;
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

// fold the seams: hop t0(g + 1) + d (d < 5) = tail_g[d] + head_{g+1}[d].  grid (5 * HOP / 1024, n_groups - 1, B * 2)
__global__ __launch_bounds__(256) void seam3_kernel(Istft3Args a) {
  const int g = blockIdx.y;                       // seam between group g and g + 1
  const int bc = blockIdx.z, b = bc >> 1, ch = bc & 1;
  const int d = blockIdx.x;
  const int lane4 = threadIdx.x;
  const float *tail = a.seam + ((((int64_t)b * 2 + ch) * a.n_groups + g) * 2 + 1) * (5 * HOP);
  const float *head = a.seam + ((((int64_t)b * 2 + ch) * a.n_groups + g + 1) * 2) * (5 * HOP);
  const int t0n = (g + 1) * a.G;                  // every group holds >= G >= 5 frames, so all five head hops exist
  const float4 x = reinterpret_cast<const float4 *>(tail + d * HOP)[lane4];
  const float4 y = reinterpret_cast<const float4 *>(head + d * HOP)[lane4];
  emit_hop(a, b, ch, (int64_t)t0n + d, make_float4(x.x + y.x, x.y + y.y, x.z + y.z, x.w + y.w), lane4);
}
#endif  // ASX_HOST_TEST

}  // namespace f3
}  // namespace asx
