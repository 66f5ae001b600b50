// STFT / iSTFT / overlap-add kernels for gfx950 (wave64, LDS-staged Stockham FFT).
//
// Replaces torch.stft / torch.istft as used by uvr_lib_v5/stft.py:41,117 and the
// host-side windowed overlap-add of mdx_separator.py:358-401.
//
// A real FFT of n_fft points is computed as one complex FFT of Nh = n_fft/2
// points (packed even/odd samples) plus a split/merge pass, all inside LDS:
// two float2[Nh] ping-pong buffers, mixed radix {4,2,3,5} autosort (Stockham)
// stages so no bit-reversal pass is needed.  n_fft = 6144 -> Nh = 3072 =
// 4^5 * 3, 48 KiB of LDS per workgroup, 3 workgroups per CU.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace asx {

struct FftPlan {
  int n_fft;
  int nh;       // n_fft / 2
  int n_stage;
  int radix[16];
};

__device__ __forceinline__ float2 cmul(float2 a, float2 b) {
  return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}
__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
// multiply by -i (SIGN<0, forward) or +i (SIGN>0, inverse)
template <int SIGN>
__device__ __forceinline__ float2 crot(float2 a) {
  return SIGN < 0 ? make_float2(a.y, -a.x) : make_float2(-a.y, a.x);
}

template <int R, int SIGN>
__device__ __forceinline__ void butterfly(float2 *v) {
  if constexpr (R == 2) {
    float2 a = v[0], b = v[1];
    v[0] = cadd(a, b);
    v[1] = csub(a, b);
  } else if constexpr (R == 4) {
    float2 t0 = cadd(v[0], v[2]), t1 = csub(v[0], v[2]);
    float2 t2 = cadd(v[1], v[3]), t3 = crot<SIGN>(csub(v[1], v[3]));
    v[0] = cadd(t0, t2);
    v[1] = cadd(t1, t3);
    v[2] = csub(t0, t2);
    v[3] = csub(t1, t3);
  } else if constexpr (R == 3) {
    const float s3 = 0.86602540378443864676f;
    float2 s = cadd(v[1], v[2]), d = csub(v[1], v[2]);
    float2 m = make_float2(v[0].x - 0.5f * s.x, v[0].y - 0.5f * s.y);
    float2 r = crot<SIGN>(make_float2(s3 * d.x, s3 * d.y));
    v[0] = cadd(v[0], s);
    v[1] = cadd(m, r);
    v[2] = csub(m, r);
  } else if constexpr (R == 5) {
    const float c1 = 0.30901699437494742410f, c2 = -0.80901699437494742410f;
    const float s1 = 0.95105651629515357212f, s2 = 0.58778525229247312917f;
    float2 a = v[0];
    float2 S1 = cadd(v[1], v[4]), S2 = cadd(v[2], v[3]);
    float2 D1 = csub(v[1], v[4]), D2 = csub(v[2], v[3]);
    float2 p1 = make_float2(a.x + c1 * S1.x + c2 * S2.x, a.y + c1 * S1.y + c2 * S2.y);
    float2 p2 = make_float2(a.x + c2 * S1.x + c1 * S2.x, a.y + c2 * S1.y + c1 * S2.y);
    float2 q1 = crot<SIGN>(make_float2(s1 * D1.x + s2 * D2.x, s1 * D1.y + s2 * D2.y));
    float2 q2 = crot<SIGN>(make_float2(s2 * D1.x - s1 * D2.x, s2 * D1.y - s1 * D2.y));
    v[0] = cadd(a, cadd(S1, S2));
    v[1] = cadd(p1, q1);
    v[4] = csub(p1, q1);
    v[2] = cadd(p2, q2);
    v[3] = csub(p2, q2);
  } else if constexpr (R == 8) {
    // 8 = 4 x 2: n = 2 n1 + n2, k = k1 + 4 k2:  X[k1 + 4 k2] = sum_n2 (-1)^(n2 k2) W8^(n2 k1) DFT4_n1(v[2 n1 + n2])[k1]
    const float h = 0.70710678118654752440f;
    float2 e[4] = {v[0], v[2], v[4], v[6]}, o[4] = {v[1], v[3], v[5], v[7]};
    butterfly<4, SIGN>(e);
    butterfly<4, SIGN>(o);
    // W8^k1 (forward: exp(-i pi k1 / 4); inverse: conjugate)
    o[1] = SIGN < 0 ? make_float2(h * (o[1].x + o[1].y), h * (o[1].y - o[1].x)) : make_float2(h * (o[1].x - o[1].y), h * (o[1].y + o[1].x));
    o[2] = crot<SIGN>(o[2]);
    o[3] = SIGN < 0 ? make_float2(h * (o[3].y - o[3].x), -h * (o[3].x + o[3].y)) : make_float2(-h * (o[3].x + o[3].y), h * (o[3].x - o[3].y));
#pragma unroll
    for (int k1 = 0; k1 < 4; ++k1) {
      v[k1] = cadd(e[k1], o[k1]);
      v[k1 + 4] = csub(e[k1], o[k1]);
    }
  } else if constexpr (R == 16) {
    // 16 = 4 x 4: n = 4 n1 + n2, k = k1 + 4 k2:  X[k1 + 4 k2] = DFT4_n2( W16^(n2 k1) DFT4_n1(v[4 n1 + n2])[k1] )[k2]
    const float c1 = 0.92387953251128675613f, s1 = 0.38268343236508977173f, h = 0.70710678118654752440f;
    float2 t[4][4];
#pragma unroll
    for (int n2 = 0; n2 < 4; ++n2) {
      float2 c[4] = {v[n2], v[4 + n2], v[8 + n2], v[12 + n2]};
      butterfly<4, SIGN>(c);
#pragma unroll
      for (int k1 = 0; k1 < 4; ++k1) t[n2][k1] = c[k1];
    }
    // twiddles W16^(n2 k1), exponent m = n2 k1 in {1, 2, 3, 2, 4, 6, 3, 6, 9}; forward W = (cos, -sin)
    auto twm = [&](float2 a, float wr, float wi) {      // a * (wr + i wi), wi given for the FORWARD transform
      const float im = SIGN < 0 ? wi : -wi;
      return make_float2(a.x * wr - a.y * im, a.x * im + a.y * wr);
    };
    t[1][1] = twm(t[1][1], c1, -s1);
    t[1][2] = twm(t[1][2], h, -h);
    t[1][3] = twm(t[1][3], s1, -c1);
    t[2][1] = twm(t[2][1], h, -h);
    t[2][2] = crot<SIGN>(t[2][2]);                       // W16^4 = -i (forward)
    t[2][3] = twm(t[2][3], -h, -h);
    t[3][1] = twm(t[3][1], s1, -c1);
    t[3][2] = twm(t[3][2], -h, -h);
    t[3][3] = twm(t[3][3], -c1, s1);                     // W16^9 = -W16^1 = (-cos(pi/8), +sin(pi/8))
#pragma unroll
    for (int k1 = 0; k1 < 4; ++k1) {
      float2 c[4] = {t[0][k1], t[1][k1], t[2][k1], t[3][k1]};
      butterfly<4, SIGN>(c);
#pragma unroll
      for (int k2 = 0; k2 < 4; ++k2) v[k1 + 4 * k2] = c[k2];
    }
  }
}

// One Stockham stage.  tw[j] = exp(-2*pi*i*j / n_fft), j in [0, n_fft).
template <int R, int SIGN>
__device__ __forceinline__ void stockham_stage(const float2 *__restrict__ in, float2 *__restrict__ out, int nh,
                                               int ns, const float2 *__restrict__ tw) {
  const int nb = nh / R;
  const int tstep = (nh / (ns * R)) * 2;
  for (int j = threadIdx.x; j < nb; j += blockDim.x) {
    const int q = j / ns;
    const int k = j - q * ns;
    float2 v[R];
#pragma unroll
    for (int r = 0; r < R; ++r) v[r] = in[j + r * nb];
    const int base = k * tstep;
#pragma unroll
    for (int r = 1; r < R; ++r) {
      float2 w = tw[base * r];
      if (SIGN > 0) w.y = -w.y;
      v[r] = cmul(v[r], w);
    }
    butterfly<R, SIGN>(v);
    const int j0 = q * ns * R + k;
#pragma unroll
    for (int r = 0; r < R; ++r) out[j0 + r * ns] = v[r];
  }
}

// Full complex FFT of nh points held in LDS buffer `a` (scratch `b`).
// Returns the buffer that holds the result.  All threads of the block call it.
template <int SIGN>
__device__ float2 *fft_lds(float2 *a, float2 *b, const FftPlan &p, const float2 *__restrict__ tw) {
  int ns = 1;
  for (int s = 0; s < p.n_stage; ++s) {
    const int R = p.radix[s];
    __syncthreads();
    switch (R) {
      case 16: stockham_stage<16, SIGN>(a, b, p.nh, ns, tw); break;
      case 8: stockham_stage<8, SIGN>(a, b, p.nh, ns, tw); break;
      case 4: stockham_stage<4, SIGN>(a, b, p.nh, ns, tw); break;
      case 2: stockham_stage<2, SIGN>(a, b, p.nh, ns, tw); break;
      case 3: stockham_stage<3, SIGN>(a, b, p.nh, ns, tw); break;
      default: stockham_stage<5, SIGN>(a, b, p.nh, ns, tw); break;
    }
    ns *= R;
    float2 *t = a;
    a = b;
    b = t;
  }
  __syncthreads();
  return a;
}

// ---------------------------------------------------------------------------
// K1: STFT.  grid = (T, 2, B).  One workgroup per (frame, channel, chunk).
//
// The chunk waveform is either an explicit [B,2,C] buffer (song_len < 0) or a
// window into the resident song mix [2,N] (mdx_separator.py:329-366): chunk b
// starts at padded position chunk_start[b]; padded position j maps to
// mix[j - trim] for trim <= j < trim + N and to 0 elsewhere (left `trim`
// zeros, right padding, tail zero-fill of the last chunk).
// torch.stft(center=True) reflect-pads each chunk by n_fft/2 (stft.py:41).
//
// Output layout: tf_layout=1 -> [B,4,T,F] (engine-internal, coalesced);
//                tf_layout=0 -> [B,4,F,T] (reference layout, stft.py:44-56).
// zero_low: bins [0, zero_low) are written as 0 (mdx_separator.py:425).
// ---------------------------------------------------------------------------
struct StftArgs {
  const float *wave;          // [B,2,C] or song mix [2,N]
  const int64_t *chunk_start; // [B] padded-domain start of each chunk (song mode) or nullptr
  int64_t n_song;             // N (song mode) or -1
  int trim;                   // n_fft/2 (song mode)
  int64_t C;                  // samples per chunk
  int hop;
  int T;
  int dim_f;
  int zero_low;
  int tf_layout;
  float *spec;
  const float *window;        // [n_fft] periodic Hann
  const float2 *tw;           // [n_fft]
  float sign;                 // +1, or -1 to emit the negated spectrum (denoise pass)
  int subbands;               // k >= 1: bin j*F/k + f' of plane p goes to plane p*k + j (cac2cws, tfc_tdf_v3.py:216)
  int64_t out_bstride;        // floats between batch items of spec (tf_layout only); 0 = dense
};

__global__ __launch_bounds__(256) void stft_kernel(StftArgs a, FftPlan p) {
  extern __shared__ float2 lds[];
  float2 *bufA = lds;
  float2 *bufB = lds + p.nh;
  const int t = blockIdx.x, ch = blockIdx.y, b = blockIdx.z;
  const int half = p.nh;  // n_fft / 2
  const int64_t C = a.C;
  const float *src;
  int64_t cstart = 0;
  if (a.n_song >= 0) {
    src = a.wave + (int64_t)ch * a.n_song;
    cstart = a.chunk_start[b];
  } else {
    src = a.wave + ((int64_t)b * 2 + ch) * C;
  }
  // load + window; LDS float pairs (x[2m], x[2m+1]) are the packed complex input
  float *fa = reinterpret_cast<float *>(bufA);
  for (int e = threadIdx.x; e < p.n_fft; e += blockDim.x) {
    int64_t q = (int64_t)t * a.hop + e - half;
    if (q < 0) q = -q;
    if (q >= C) q = 2 * (C - 1) - q;
    float v;
    if (a.n_song >= 0) {
      const int64_t j = cstart + q - a.trim;  // index into the un-padded mix
      v = (j >= 0 && j < a.n_song) ? src[j] : 0.0f;
    } else {
      v = src[q];
    }
    fa[e] = v * a.window[e];
  }
  float2 *Z = fft_lds<-1>(bufA, bufB, p, a.tw);
  // split: X[k] = E + e^{-2 pi i k / n} * O,  E = (Z[k] + conj Z[Nh-k]) / 2,  O = -i (Z[k] - conj Z[Nh-k]) / 2
  const int nh = p.nh;
  for (int k = threadIdx.x; k < a.dim_f; k += blockDim.x) {
    float re = 0.f, im = 0.f;
    if (k >= a.zero_low) {
      const float2 zk = Z[k == nh ? 0 : k];
      float2 zc = Z[(k == 0 || k == nh) ? 0 : nh - k];
      zc.y = -zc.y;
      const float2 E = make_float2(0.5f * (zk.x + zc.x), 0.5f * (zk.y + zc.y));
      const float2 D = make_float2(0.5f * (zk.x - zc.x), 0.5f * (zk.y - zc.y));
      const float2 O = make_float2(D.y, -D.x);
      float2 w = (k == nh) ? make_float2(-1.f, 0.f) : a.tw[k];
      const float2 X = cadd(E, cmul(w, O));
      re = X.x * a.sign;
      im = X.y * a.sign;
    }
    if (a.tf_layout == 2) {
      // BS-Roformer: b t (f s c) -- frequency-major with the stereo channel interleaved (bs_roformer.py:455-459)
      reinterpret_cast<float2 *>(a.spec)[(((int64_t)b * a.T + t) * a.dim_f + k) * 2 + ch] = make_float2(re, im);
    } else if (a.tf_layout) {
      const int kb = a.subbands > 1 ? a.subbands : 1;
      const int fs = a.dim_f / kb;
      const int j = k / fs, fp = k - j * fs;
      const int64_t bst = a.out_bstride ? a.out_bstride : (int64_t)4 * a.T * a.dim_f;
      const int64_t base = (int64_t)b * bst + (((int64_t)(ch * 2) * kb + j) * a.T + t) * fs + fp;
      a.spec[base] = re;
      a.spec[base + (int64_t)kb * a.T * fs] = im;
    } else {
      const int64_t base = (((int64_t)b * 4 + ch * 2) * a.dim_f + k) * a.T + t;
      a.spec[base] = re;
      a.spec[base + (int64_t)a.dim_f * a.T] = im;
    }
  }
}

// ---------------------------------------------------------------------------
// K2: inverse FFT of every frame -> windowed frames [B,2,T,n_fft].
// grid = (T, 2, B).  Bins >= dim_f are zero (stft.py:58-68); the imaginary
// parts of DC and Nyquist are ignored (c2r semantics of torch.istft).
// combine != 0: spectrum = 0.5 * spec[b] - 0.5 * spec[b + B]   (denoise,
// mdx_separator.py:435-440; pass the positive batch first).
// ---------------------------------------------------------------------------
struct IstftArgs {
  const float *spec;  // [B(,x2),4,T,F] (tf_layout=1) or [B,4,F,T]
  int T;
  int dim_f;
  int tf_layout;
  int combine;        // 0 or B (offset, in chunks, of the negated-input batch)
  float *frames;      // [B,2,T,n_fft]
  const float *window;
  const float2 *tw;
  int subbands;        // cws2cac (tfc_tdf_v3.py:223): plane p*k + j holds bins j*F/k ...
  int n_inst;          // S >= 1 stems per batch item: blockIdx.z = b*S + s, planes [s*4k, (s+1)*4k)
  int64_t in_bstride;  // floats between batch items of spec (tf_layout only); 0 = dense
  const float *mask;   // tf_layout == 2 (BS-Roformer): complex mask [B, S, T, (f s c)] multiplied in (bs_roformer.py:503-506)
};

__device__ __forceinline__ float2 load_bin(const IstftArgs &a, int b, int ch, int t, int k) {
  if (k >= a.dim_f) return make_float2(0.f, 0.f);
  float re, im;
  if (a.tf_layout == 2) {
    const int S = a.n_inst > 1 ? a.n_inst : 1;
    const int bb = b / S, si = b - bb * S;
    const float2 x = reinterpret_cast<const float2 *>(a.spec)[(((int64_t)bb * a.T + t) * a.dim_f + k) * 2 + ch];
    const float2 m = reinterpret_cast<const float2 *>(a.mask)[((((int64_t)bb * S + si) * a.T + t) * a.dim_f + k) * 2 + ch];
    return make_float2(x.x * m.x - x.y * m.y, x.x * m.y + x.y * m.x);
  } else if (a.tf_layout) {
    const int kb = a.subbands > 1 ? a.subbands : 1;
    const int S = a.n_inst > 1 ? a.n_inst : 1;
    const int fs = a.dim_f / kb;
    const int j = k / fs, fp = k - j * fs;
    const int bb = b / S, si = b - bb * S;
    const int64_t bst = a.in_bstride ? a.in_bstride : (int64_t)S * 4 * a.T * a.dim_f;
    const int64_t base = (int64_t)bb * bst + ((((int64_t)si * 4 + ch * 2) * kb + j) * a.T + t) * fs + fp;
    re = a.spec[base];
    im = a.spec[base + (int64_t)kb * a.T * fs];
  } else {
    const int64_t base = (((int64_t)b * 4 + ch * 2) * a.dim_f + k) * a.T + t;
    re = a.spec[base];
    im = a.spec[base + (int64_t)a.dim_f * a.T];
  }
  return make_float2(re, im);
}

__global__ __launch_bounds__(256) void istft_kernel(IstftArgs a, FftPlan p) {
  extern __shared__ float2 lds[];
  float2 *bufA = lds;
  float2 *bufB = lds + p.nh;
  float2 *bufX = lds + 2 * p.nh;  // staged spectrum X[0..nh]
  const int t = blockIdx.x, ch = blockIdx.y, b = blockIdx.z;
  const int nh = p.nh;
  for (int k = threadIdx.x; k <= nh; k += blockDim.x) {
    float2 x = load_bin(a, b, ch, t, k);
    if (a.combine) {
      const float2 xn = load_bin(a, b + a.combine, ch, t, k);
      x.x = xn.x * -0.5f + x.x * 0.5f;
      x.y = xn.y * -0.5f + x.y * 0.5f;
    }
    if (k == 0 || k == nh) x.y = 0.f;
    bufX[k] = x;
  }
  __syncthreads();
  // merge: Z[k] = E + i*O, E = (X[k] + conj X[Nh-k]) / 2, O = e^{+2 pi i k / n} (X[k] - conj X[Nh-k]) / 2
  for (int k = threadIdx.x; k < nh; k += blockDim.x) {
    const float2 xk = bufX[k];
    float2 xc = bufX[nh - k];
    xc.y = -xc.y;
    const float2 E = make_float2(0.5f * (xk.x + xc.x), 0.5f * (xk.y + xc.y));
    const float2 D = make_float2(0.5f * (xk.x - xc.x), 0.5f * (xk.y - xc.y));
    float2 w = a.tw[k];
    w.y = -w.y;
    const float2 O = cmul(w, D);
    bufA[k] = make_float2(E.x - O.y, E.y + O.x);
  }
  float2 *z = fft_lds<+1>(bufA, bufB, p, a.tw);
  const float scale = 1.0f / (float)nh;
  float2 *dst = reinterpret_cast<float2 *>(a.frames + (((int64_t)b * 2 + ch) * a.T + t) * p.n_fft);
  const float2 *w2 = reinterpret_cast<const float2 *>(a.window);
  for (int m = threadIdx.x; m < nh; m += blockDim.x) {
    const float2 v = z[m];
    const float2 w = w2[m];
    dst[m] = make_float2((v.x * scale) * w.x, (v.y * scale) * w.y);
  }
}

// ---------------------------------------------------------------------------
// K2b: fold the windowed frames (torch.istft overlap-add, / sum w^2, strip
// n_fft/2) and, for the chunk loop, apply the symmetric Hann chunk window of
// mdx_separator.py:358,387 (np.hanning in float64, product rounded to f32).
//   out[b,ch,j] = (sum_t frames[b,ch,t,j + n/2 - t*hop]) / env[j + n/2]   (* v_b[j])
// n_act[b] < 0: no chunk window (plain STFT.inverse / overlap == 0).
// Samples j >= |n_act[b]| of a windowed chunk are written as 0.
// ---------------------------------------------------------------------------
__device__ __forceinline__ double hanning_f64(int64_t j, int64_t M) {
  // numpy.hanning: 0.5 + 0.5*cos(pi*n/(M-1)), n = 2j + 1 - M
  if (M == 1) return 1.0;
  return 0.5 + 0.5 * cos(3.14159265358979323846 * (double)(2 * j + 1 - M) / (double)(M - 1));
}

__global__ __launch_bounds__(256) void ola_kernel(const float *__restrict__ frames, const float *__restrict__ env,
                                                  const int64_t *__restrict__ n_act, int n_fft, int hop, int T,
                                                  int64_t C, float *__restrict__ out) {
  const int b = blockIdx.z, ch = blockIdx.y;
  const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= C) return;
  const int64_t m = j + n_fft / 2;
  int64_t t_lo = (m - n_fft + hop) / hop;  // ceil((m - n + 1) / hop) for m - n + 1 > 0
  if (m - n_fft + 1 <= 0) t_lo = 0;
  int64_t t_hi = m / hop;
  if (t_hi > T - 1) t_hi = T - 1;
  const float *fr = frames + ((int64_t)b * 2 + ch) * T * n_fft;
  float acc = 0.f;
  for (int64_t t = t_lo; t <= t_hi; ++t) acc += fr[t * n_fft + (m - t * hop)];
  float y = acc / env[m];
  if (n_act != nullptr) {
    const int64_t na = n_act[b];
    if (na >= 0) {
      if (j < na)
        y = (float)((double)y * hanning_f64(j, na));
      else
        y = 0.f;
    }
  }
  out[((int64_t)b * 2 + ch) * C + j] = y;
}

// ---------------------------------------------------------------------------
// K4/K5: fold every windowed chunk into the song (mdx_separator.py:386-401):
//   result[m] = sum_k yw_k[m - k*step]   divider[m] = sum_k v_k[m - k*step]
//   out[ch,i] = result[i + trim] / divider[i + trim]
// Gather form: each output sample visits its covering chunks in increasing k,
// the reference's accumulation order, so the sum is reproducible.
// windowed == 0 (overlap == 0): divider counts chunks.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void finalize_kernel(const float *__restrict__ chunk_out, int n_chunks, int64_t C,
                                                       int64_t step, int64_t L, int trim, int64_t N, int windowed,
                                                       float *__restrict__ out, const double *__restrict__ hann) {
  const int ch = blockIdx.y;
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  const int64_t m = i + trim;
  int64_t k_hi = m / step;
  if (k_hi > n_chunks - 1) k_hi = n_chunks - 1;
  int64_t k_lo = 0;
  if (m - C >= 0) k_lo = (m - C) / step + 1;
  float acc = 0.f, div = 0.f;
  for (int64_t k = k_lo; k <= k_hi; ++k) {
    const int64_t s = k * step;
    const int64_t j = m - s;
    int64_t na = L - s;
    if (na > C) na = C;
    if (j >= na) continue;
    acc += chunk_out[(k * 2 + ch) * C + j];
    if (windowed)
      div = (float)((double)div + ((hann && na == C) ? hann[j] : hanning_f64(j, na)));   // hann = np.hanning(C) in float64, the same expression
    else
      div += 1.0f;
  }
  out[(int64_t)ch * N + i] = acc / div;
}

// The divider of the fold is input-independent ("precompute"): divider[i + trim] for every output sample, built
// ONCE per plan with exactly the accumulation of finalize_kernel (float32 += float64 window value, covering chunks in
// increasing k), so that finalize4_kernel divides by bit-identical values.
__global__ __launch_bounds__(256) void finalize_div_kernel(int n_chunks, int64_t C, int64_t step, int64_t L, int trim, int64_t N,
                                                           int windowed, float *__restrict__ divider, const double *__restrict__ hann) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  const int64_t m = i + trim;
  int64_t k_hi = m / step;
  if (k_hi > n_chunks - 1) k_hi = n_chunks - 1;
  int64_t k_lo = 0;
  if (m - C >= 0) k_lo = (m - C) / step + 1;
  float div = 0.f;
  for (int64_t k = k_lo; k <= k_hi; ++k) {
    const int64_t s = k * step;
    const int64_t j = m - s;
    int64_t na = L - s;
    if (na > C) na = C;
    if (j >= na) continue;
    if (windowed)
      div = (float)((double)div + ((hann && na == C) ? hann[j] : hanning_f64(j, na)));
    else
      div += 1.0f;
  }
  divider[i] = div;
}

// K4/K5 with the divider table: four consecutive samples per thread (C, step, trim multiples of 4: a group never straddles
// a chunk boundary, and (k*2 + ch)*C + j stays 16-byte aligned), the covering chunks read as float4 in increasing k.
// Same sums, same division as finalize_kernel -> bit-identical output.
__global__ __launch_bounds__(256) void finalize4_kernel(const float *__restrict__ chunk_out, int n_chunks, int64_t C, int64_t step,
                                                        int64_t L, int trim, int64_t N, const float *__restrict__ divider,
                                                        float *__restrict__ out) {
  const int ch = blockIdx.y;
  const int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (i >= N) return;
  const int64_t m = i + trim;
  int64_t k_hi = m / step;
  if (k_hi > n_chunks - 1) k_hi = n_chunks - 1;
  int64_t k_lo = 0;
  if (m - C >= 0) k_lo = (m - C) / step + 1;
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  for (int64_t k = k_lo; k <= k_hi; ++k) {
    const int64_t s = k * step;
    const int64_t j = m - s;
    int64_t na = L - s;
    if (na > C) na = C;
    if (j >= na) continue;
    const float4 v = *reinterpret_cast<const float4 *>(chunk_out + (k * 2 + ch) * C + j);   // j + 3 < C: inside the chunk row
    acc[0] += v.x;
    if (j + 1 < na) acc[1] += v.y;
    if (j + 2 < na) acc[2] += v.z;
    if (j + 3 < na) acc[3] += v.w;
  }
  float *o = out + (int64_t)ch * N + i;
  if (i + 4 <= N) {
    const float4 d = *reinterpret_cast<const float4 *>(divider + i);
    float4 r;
    r.x = acc[0] / d.x;
    r.y = acc[1] / d.y;
    r.z = acc[2] / d.z;
    r.w = acc[3] / d.w;
    *reinterpret_cast<float4 *>(o) = r;
  } else {
    for (int e = 0; e < 4 && i + e < N; ++e) o[e] = acc[e] / divider[i + e];
  }
}

// ---------------------------------------------------------------------------
// Stem algebra of MDXSeparator.separate (mdx_separator.py:155-182) on the device:
//   peak = max|mix|                               (absmax_kernel -> *peak_bits, float bits of a value >= 0)
//   mix *= thr / peak  if peak > thr  (elif peak < amp: mix *= amp / peak)     spec_utils.py:99-115, in place
//   primary[i, ch]   = demixed[ch, i] * peak                                   mdx_separator.py:159
//   secondary[i, ch] = (-primary[i, ch] * compensate) + mix[ch, i]             mdx_separator.py:182
// All arithmetic is float32 with one rounding per operation (numpy semantics; no fma contraction).
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void absmax_kernel(const float *__restrict__ x, int64_t n, unsigned int *peak_bits) {
  float m = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    m = fmaxf(m, fabsf(x[i]));
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
  __shared__ float wm[4];
  if ((threadIdx.x & 63) == 0) wm[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    m = fmaxf(fmaxf(wm[0], wm[1]), fmaxf(wm[2], wm[3]));
    atomicMax(peak_bits, __float_as_uint(m));   // non-negative floats order like their bit patterns
  }
}

__global__ __launch_bounds__(256) void normalize_kernel(float *__restrict__ x, int64_t n, const unsigned int *peak_bits,
                                                        float max_peak, float min_peak, int has_min) {
  const float maxv = __uint_as_float(*peak_bits);
  float scale;
  if (maxv > max_peak) scale = __fdiv_rn(max_peak, maxv);
  else if (has_min && maxv < min_peak) scale = __fdiv_rn(min_peak, maxv);
  else return;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    x[i] = x[i] * scale;
}

// Writer edge (common_separator.py:309-337): spec_utils.normalize, then (stem * 32767).astype(np.int16) and channel
// interleave.  stem [2, N] planar -> pcm [N, 2] int16.  float32 with one rounding per operation, C truncation.
__global__ __launch_bounds__(256) void pcm16_kernel(const float *__restrict__ stem, int64_t N, const unsigned int *peak_bits,
                                                    float max_peak, float min_peak, int has_min, short *__restrict__ pcm) {
#pragma clang fp contract(off)
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  const float maxv = __uint_as_float(*peak_bits);
  float scale = 1.0f;
  bool scaled = false;
  if (maxv > max_peak) {
    scale = __fdiv_rn(max_peak, maxv);
    scaled = true;
  } else if (has_min && maxv < min_peak) {
    scale = __fdiv_rn(min_peak, maxv);
    scaled = true;
  }
  float l = stem[i], r = stem[N + i];
  if (scaled) {
    l = l * scale;
    r = r * scale;
  }
  const float lq = l * 32767.0f, rq = r * 32767.0f;
  short2 o;
  o.x = (short)(int)lq;
  o.y = (short)(int)rq;
  reinterpret_cast<short2 *>(pcm)[i] = o;
}

// secondary = mix - primary (mdxc_separator.py:406-468: the residual stem of a single-target model), float32
__global__ __launch_bounds__(256) void residual_kernel(const float *__restrict__ mix, const float *__restrict__ stem, int64_t n,
                                                       float *__restrict__ out) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) out[i] = mix[i] - stem[i];
}

// The same writer edge for a stem that is already [N, 2] interleaved (what asx_separate_dev produces and write_audio consumes):
// purely elementwise over the 2 N values, four per thread.  Bit-identical to pcm16_kernel on the transposed input.
__global__ __launch_bounds__(256) void pcm16_rows_kernel(const float *__restrict__ stem, int64_t n2, const unsigned int *peak_bits,
                                                         float max_peak, float min_peak, int has_min, short *__restrict__ pcm) {
#pragma clang fp contract(off)
  const int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (i >= n2) return;
  const float maxv = __uint_as_float(*peak_bits);
  float scale = 1.0f;
  bool scaled = false;
  if (maxv > max_peak) {
    scale = __fdiv_rn(max_peak, maxv);
    scaled = true;
  } else if (has_min && maxv < min_peak) {
    scale = __fdiv_rn(min_peak, maxv);
    scaled = true;
  }
  float v[4];
  const bool full = i + 4 <= n2;
  if (full) {
    const float4 q = *reinterpret_cast<const float4 *>(stem + i);
    v[0] = q.x;
    v[1] = q.y;
    v[2] = q.z;
    v[3] = q.w;
  } else {
    for (int j = 0; j < 4; ++j) v[j] = i + j < n2 ? stem[i + j] : 0.f;
  }
  short o[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    float x = v[j];
    if (scaled) x = x * scale;
    const float xq = x * 32767.0f;
    o[j] = (short)(int)xq;
  }
  if (full) {
    short4 w;
    w.x = o[0];
    w.y = o[1];
    w.z = o[2];
    w.w = o[3];
    *reinterpret_cast<short4 *>(pcm + i) = w;
  } else {
    for (int j = 0; j < 4 && i + j < n2; ++j) pcm[i + j] = o[j];
  }
}

// Decode edge: the data chunk of a RIFF/WAVE file, [frames, channels] interleaved little-endian samples, -> float32 planar
// [2, frames] (a mono file feeds both rows: common_separator.py:278-280), with the conversions libsndfile / audio_io.read_wav
// apply: PCM_16 x / 2^15, PCM_24 x / 2^23, PCM_32 (double)x / 2^31 rounded to float, IEEE float copied.  fmt = bits per sample
// (16, 24, 32) or 0x20 | 0x100 for float32.  Also folds max |x| into peak_bits (the "file is silent" check of prepare_mix).
__global__ __launch_bounds__(256) void pcm_decode_kernel(const unsigned char *__restrict__ raw, int64_t frames, int channels, int fmt,
                                                         float *__restrict__ out, unsigned int *peak_bits) {
  float m = 0.f;
  const int bps = (fmt & 0xff) / 8;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < frames; i += (int64_t)gridDim.x * blockDim.x) {
    float v[2];
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const int cc = c < channels ? c : channels - 1;
      const unsigned char *p = raw + (i * channels + cc) * bps;
      float x;
      if (fmt == 16) {
        const short q = (short)((unsigned)p[0] | ((unsigned)p[1] << 8));
        x = (float)q * (1.0f / 32768.0f);
      } else if (fmt == 24) {
        int q = (int)((unsigned)p[0] | ((unsigned)p[1] << 8) | ((unsigned)p[2] << 16));
        if (q >= (1 << 23)) q -= (1 << 24);
        x = (float)q * (1.0f / 8388608.0f);
      } else if (fmt == 32) {
        const int q = (int)((unsigned)p[0] | ((unsigned)p[1] << 8) | ((unsigned)p[2] << 16) | ((unsigned)p[3] << 24));
        x = (float)((double)q * (1.0 / 2147483648.0));
      } else {
        x = __uint_as_float((unsigned)p[0] | ((unsigned)p[1] << 8) | ((unsigned)p[2] << 16) | ((unsigned)p[3] << 24));
      }
      v[c] = x;
    }
    out[i] = v[0];
    out[frames + i] = v[1];
    m = fmaxf(m, fmaxf(fabsf(v[0]), fabsf(v[1])));
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
  __shared__ float wm[4];
  if ((threadIdx.x & 63) == 0) wm[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    m = fmaxf(fmaxf(wm[0], wm[1]), fmaxf(wm[2], wm[3]));
    atomicMax(peak_bits, __float_as_uint(m));
  }
}

__global__ __launch_bounds__(256) void stems_kernel(const float *__restrict__ demixed, const float *__restrict__ mix,
                                                    int64_t N, const unsigned int *peak_bits, float compensate,
                                                    float *__restrict__ primary, float *__restrict__ secondary) {
#pragma clang fp contract(off)  // numpy rounds the product before the add; an fma would differ by 1 ulp
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  const float peak = __uint_as_float(*peak_bits);
  // plain operators: the pragma above only governs expressions written in this function
  const float p0 = demixed[i] * peak, p1 = demixed[N + i] * peak;
  const float t0 = -p0 * compensate, t1 = -p1 * compensate;
  const float s0 = t0 + mix[i], s1 = t1 + mix[N + i];
  reinterpret_cast<float2 *>(primary)[i] = make_float2(p0, p1);
  reinterpret_cast<float2 *>(secondary)[i] = make_float2(s0, s1);
}

// ---------------------------------------------------------------------------
// MDXC (TFC branch) fold: accumulated[..., k*hop : k*hop+chunk] += out_k ; result = accumulated / overlap
// (mdxc_separator.py:398-402).  Gather form over the covering chunks in increasing k.
// chunk_out [n_chunks, S, 2, C];  out [S, 2, N];  sample i sits at padded position i + front.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void mdxc_finalize_kernel(const float *__restrict__ chunk_out, int n_chunks, int S,
                                                            int64_t C, int64_t hop, int64_t front, int64_t N,
                                                            float overlap, float *__restrict__ out) {
  const int sc = blockIdx.y;  // s*2 + ch
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  const int64_t m = i + front;
  int64_t k_hi = m / hop;
  if (k_hi > n_chunks - 1) k_hi = n_chunks - 1;
  int64_t k_lo = 0;
  if (m - C >= 0) k_lo = (m - C) / hop + 1;
  float acc = 0.f;
  for (int64_t k = k_lo; k <= k_hi; ++k) acc += chunk_out[((k * S * 2) + sc) * C + (m - k * hop)];
  out[(int64_t)sc * N + i] = acc / overlap;
}

// ---------------------------------------------------------------------------
// TFC-TDF v3 pre-activation blocks: InstanceNorm2d(affine) statistics and norm -> act.
// x is a channel-slice view [B, C, P] (P = T*F) with batch stride x_bstride.
// stats[b*C + c] = (mean, 1/sqrt(var + eps)), biased variance, accumulated in float64.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void instnorm_stats_kernel(const float *__restrict__ x, int64_t x_bstride, int C,
                                                             int64_t P, float eps, float2 *__restrict__ stats) {
  const int c = blockIdx.x, b = blockIdx.y;
  const float *xp = x + (int64_t)b * x_bstride + (int64_t)c * P;
  double s = 0.0, q = 0.0;
  if ((P & 3) == 0) {
    const float4 *x4 = reinterpret_cast<const float4 *>(xp);
    for (int64_t i = threadIdx.x; i < P / 4; i += blockDim.x) {
      const float4 v = x4[i];
      s += (double)v.x + (double)v.y + (double)v.z + (double)v.w;
      q += (double)v.x * v.x + (double)v.y * v.y + (double)v.z * v.z + (double)v.w * v.w;
    }
  } else {
    for (int64_t i = threadIdx.x; i < P; i += blockDim.x) {
      const double v = xp[i];
      s += v;
      q += v * v;
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    s += __shfl_xor(s, off);
    q += __shfl_xor(q, off);
  }
  __shared__ double ws[4], wq[4];
  if ((threadIdx.x & 63) == 0) {
    ws[threadIdx.x >> 6] = s;
    wq[threadIdx.x >> 6] = q;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    s = ws[0] + ws[1] + ws[2] + ws[3];
    q = wq[0] + wq[1] + wq[2] + wq[3];
    const double mean = s / (double)P;
    double var = q / (double)P - mean * mean;
    if (var < 0.0) var = 0.0;
    stats[(int64_t)b * C + c] = make_float2((float)mean, (float)(1.0 / sqrt(var + (double)eps)));
  }
}

__device__ __forceinline__ float v3_act(float v, int act) {
  if (act == 1) return fmaxf(v, 0.f);
  if (act == 2) return 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f));
  return v;
}

// y[b,c,:] = act((x - mean) * rstd * gamma[c] + beta[c]); stats == nullptr -> act only (norm = None).
__global__ __launch_bounds__(256) void norm_act_kernel(const float *__restrict__ x, int64_t x_bstride, int C, int64_t P,
                                                       const float2 *__restrict__ stats, const float *__restrict__ gamma,
                                                       const float *__restrict__ beta, int act, float *__restrict__ y) {
  const int c = blockIdx.y, b = blockIdx.z;
  const float *xp = x + (int64_t)b * x_bstride + (int64_t)c * P;
  float *yp = y + ((int64_t)b * C + c) * P;
  float mean = 0.f, sc = 1.f, sh = 0.f;
  if (stats != nullptr) {
    const float2 st = stats[(int64_t)b * C + c];
    mean = st.x;
    sc = st.y * gamma[c];
    sh = beta[c];
  }
  if ((P & 3) == 0) {
    const float4 *x4 = reinterpret_cast<const float4 *>(xp);
    float4 *y4 = reinterpret_cast<float4 *>(yp);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < P / 4; i += (int64_t)gridDim.x * blockDim.x) {
      float4 v = x4[i];
      v.x = v3_act((v.x - mean) * sc + sh, act);
      v.y = v3_act((v.y - mean) * sc + sh, act);
      v.z = v3_act((v.z - mean) * sc + sh, act);
      v.w = v3_act((v.w - mean) * sc + sh, act);
      y4[i] = v;
    }
  } else {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < P; i += (int64_t)gridDim.x * blockDim.x)
      yp[i] = v3_act((xp[i] - mean) * sc + sh, act);
  }
}

// ---------------------------------------------------------------------------
// GroupNorm(G, C) of the ConvTDFNet variant built with optimizer == 'adamw' (uvr_lib_v5/mdxnet.py:48-49: norm = GroupNorm(2, c)
// after every conv / linear): x [B, C, P] dense.  Two kernels: per-plane float64 sums, then normalise + affine + ReLU with the
// optional "+ res" (x + tdf(x), modules.py:74) or "* mul" (x *= ds_outputs[-i-1], mdxnet.py:113) of the consuming statement fused.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gn_partial_kernel(const float *__restrict__ x, int C, int64_t P, double2 *__restrict__ part) {
  const int c = blockIdx.x, b = blockIdx.y;
  const float *xp = x + ((int64_t)b * C + c) * P;
  double s = 0.0, q = 0.0;
  if ((P & 3) == 0) {
    const float4 *x4 = reinterpret_cast<const float4 *>(xp);
    for (int64_t i = threadIdx.x; i < P / 4; i += blockDim.x) {
      const float4 v = x4[i];
      s += (double)v.x + (double)v.y + (double)v.z + (double)v.w;
      q += (double)v.x * v.x + (double)v.y * v.y + (double)v.z * v.z + (double)v.w * v.w;
    }
  } else {
    for (int64_t i = threadIdx.x; i < P; i += blockDim.x) {
      const double v = xp[i];
      s += v;
      q += v * v;
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    s += __shfl_xor(s, off);
    q += __shfl_xor(q, off);
  }
  __shared__ double ws[4], wq[4];
  if ((threadIdx.x & 63) == 0) {
    ws[threadIdx.x >> 6] = s;
    wq[threadIdx.x >> 6] = q;
  }
  __syncthreads();
  if (threadIdx.x == 0) part[(int64_t)b * C + c] = make_double2(ws[0] + ws[1] + ws[2] + ws[3], wq[0] + wq[1] + wq[2] + wq[3]);
}

// y[b,c,:] = relu((x - mean_g) * rstd_g * gamma[c] + beta[c]) (+ res[b,c,:]) (* mul[b,c,:]); g = c / (C / G); y may alias x.
// The group's statistics are folded from the per-plane sums in plane order by every block (C / G <= a few hundred doubles).
__global__ __launch_bounds__(256) void gn_apply_kernel(const float *__restrict__ x, int C, int G, int64_t P, const double2 *__restrict__ part,
                                                       const float *__restrict__ gamma, const float *__restrict__ beta, float eps, int relu,
                                                       const float *__restrict__ res, const float *__restrict__ mul, float *__restrict__ y) {
  const int c = blockIdx.y, b = blockIdx.z;
  const int cpg = C / G, g = c / cpg;
  __shared__ float sm[2];
  if (threadIdx.x == 0) {
    double s = 0.0, q = 0.0;
    for (int j = 0; j < cpg; ++j) {
      const double2 v = part[(int64_t)b * C + g * cpg + j];
      s += v.x;
      q += v.y;
    }
    const double n = (double)cpg * (double)P;
    const double mean = s / n;
    double var = q / n - mean * mean;
    if (var < 0.0) var = 0.0;
    sm[0] = (float)mean;
    sm[1] = (float)(1.0 / sqrt(var + (double)eps));
  }
  __syncthreads();
  const float mean = sm[0], sc = sm[1] * gamma[c], sh = beta[c];
  const int64_t base = ((int64_t)b * C + c) * P;
  auto f = [&](float v) {
    v = (v - mean) * sc + sh;
    return relu ? fmaxf(v, 0.f) : v;
  };
  if ((P & 3) == 0) {
    const float4 *x4 = reinterpret_cast<const float4 *>(x + base);
    const float4 *r4 = res ? reinterpret_cast<const float4 *>(res + base) : nullptr;
    const float4 *m4 = mul ? reinterpret_cast<const float4 *>(mul + base) : nullptr;
    float4 *y4 = reinterpret_cast<float4 *>(y + base);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < P / 4; i += (int64_t)gridDim.x * blockDim.x) {
      float4 v = x4[i];
      v.x = f(v.x);
      v.y = f(v.y);
      v.z = f(v.z);
      v.w = f(v.w);
      if (r4) {
        const float4 r = r4[i];
        v.x += r.x;
        v.y += r.y;
        v.z += r.z;
        v.w += r.w;
      }
      if (m4) {
        const float4 m = m4[i];
        v.x *= m.x;
        v.y *= m.y;
        v.z *= m.z;
        v.w *= m.w;
      }
      y4[i] = v;
    }
  } else {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < P; i += (int64_t)gridDim.x * blockDim.x) {
      float v = f(x[base + i]);
      if (res) v += res[base + i];
      if (mul) v *= mul[base + i];
      y[base + i] = v;
    }
  }
}

// y_view[b, c, :] = a[b, c, :] * m[b, c, :]   ("x * first_conv_out", tfc_tdf_v3.py:257), y is a channel-slice view
__global__ __launch_bounds__(256) void mul_into_view_kernel(const float *__restrict__ a, const float *__restrict__ m,
                                                            int64_t CP, float *__restrict__ y, int64_t y_bstride) {
  const int b = blockIdx.y;
  const float *ap = a + (int64_t)b * CP, *mp = m + (int64_t)b * CP;
  float *yp = y + (int64_t)b * y_bstride;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < CP; i += (int64_t)gridDim.x * blockDim.x)
    yp[i] = ap[i] * mp[i];
}

// [B,4,F,T] <-> [B,4,T,F] (test hooks only; the path itself stays in [.,T,F]).
__global__ void transpose_last2_kernel(const float *__restrict__ in, float *__restrict__ out, int rows, int cols) {
  __shared__ float tile[32][33];
  const int64_t plane = (int64_t)blockIdx.z * rows * cols;
  int c = blockIdx.x * 32 + threadIdx.x;
  for (int rr = threadIdx.y; rr < 32; rr += blockDim.y) {
    int r = blockIdx.y * 32 + rr;
    if (r < rows && c < cols) tile[rr][threadIdx.x] = in[plane + (int64_t)r * cols + c];
  }
  __syncthreads();
  int r2 = blockIdx.y * 32 + threadIdx.x;
  for (int cc = threadIdx.y; cc < 32; cc += blockDim.y) {
    int c2 = blockIdx.x * 32 + cc;
    if (r2 < rows && c2 < cols) out[plane + (int64_t)c2 * rows + r2] = tile[threadIdx.x][cc];
  }
}

}  // namespace asx
