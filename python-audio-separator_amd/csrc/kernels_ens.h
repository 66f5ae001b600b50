// Spectral edges on the device: Ensembler.ensemble (audio_separator/separator/ensembler.py:12-160) and
// spec_utils.invert_stem (uvr_lib_v5/spec_utils.py:557-580).  Both are librosa STFT(2048, 1024) -> per-bin
// selection / averaging -> iSTFT; one workgroup per (frame, channel) transforms every input frame, combines the
// spectra in LDS and inverse-transforms the result, so no spectrogram is ever written to HBM.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace asx {

enum { ENS_AVG_WAVE = 0, ENS_MEDIAN_WAVE = 1, ENS_MIN_WAVE = 2, ENS_MAX_WAVE = 3, ENS_AVG_FFT = 4, ENS_MEDIAN_FFT = 5,
       ENS_MIN_FFT = 6, ENS_MAX_FFT = 7, ENS_UVR_MAX_SPEC = 8, ENS_UVR_MIN_SPEC = 9, ENS_ENSEMBLE_WAV = 10 };
constexpr int ENS_ABS_BLOCKS = 256;   // partial sums per (input, channel) of ensemble_wav
constexpr int ENS_MAX_K = 8;

__device__ __forceinline__ float median_small(float *v, int K) {
  for (int i = 1; i < K; ++i) {   // insertion sort, K <= 8
    const float x = v[i];
    int j = i - 1;
    while (j >= 0 && v[j] > x) {
      v[j + 1] = v[j];
      --j;
    }
    v[j + 1] = x;
  }
  return (K & 1) ? v[K / 2] : (v[K / 2 - 1] + v[K / 2]) / 2.0f;   // np.median: mean of the two middle values
}

// avg / median / min / max over K waveforms [K, 2, N] (ensembler.py:48-64); weights only for avg
__global__ __launch_bounds__(256) void ens_wave_kernel(const float *__restrict__ waves, int K, int64_t n2, int alg,
                                                       const double *__restrict__ weights, double wsum, float *__restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n2) return;
  if (alg == ENS_AVG_WAVE) {
    float acc = 0.f;   // `ensembled += w * weight` rounds to float32 after every term (float64 product)
    for (int k = 0; k < K; ++k) acc = (float)((double)acc + (double)waves[(int64_t)k * n2 + i] * weights[k]);
    out[i] = (float)((double)acc / wsum);
  } else if (alg == ENS_MEDIAN_WAVE) {
    float v[ENS_MAX_K];
    for (int k = 0; k < K; ++k) v[k] = waves[(int64_t)k * n2 + i];
    out[i] = median_small(v, K);
  } else {
    float best = waves[i], bm = fabsf(best);
    for (int k = 1; k < K; ++k) {   // np.argmin / np.argmax: the first extremum wins
      const float x = waves[(int64_t)k * n2 + i], m = fabsf(x);
      if (alg == ENS_MIN_WAVE ? m < bm : m > bm) {
        best = x;
        bm = m;
      }
    }
    out[i] = best;
  }
}

// spec_utils.ensemble_wav as Ensembler.ensemble calls it (uvr_lib_v5/spec_utils.py:1245-1266, ensembler.py:71-72): every channel
// is taken whole from the input whose mean |x| over that channel is smallest (np.array_split along axis 0 of a [2, N] array:
// section c < 2 is channel c, the other 238 sections are empty).  Three deterministic steps: partial sums of |x| in float64
// (fixed strided order per block), the argmin per channel (first minimum; a NaN sum wins like in np.argmin), the row copy.
__global__ __launch_bounds__(256) void ens_abssum_kernel(const float *__restrict__ waves, int64_t N, double *__restrict__ partial) {
  const int kc = blockIdx.y;                        // input * 2 + channel
  const float *x = waves + (int64_t)kc * N;
  double acc = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < N; i += (int64_t)gridDim.x * 256) acc += (double)fabsf(x[i]);
  __shared__ double sh[256];
  sh[threadIdx.x] = acc;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if ((int)threadIdx.x < off) sh[threadIdx.x] += sh[threadIdx.x + off];
    __syncthreads();
  }
  if (threadIdx.x == 0) partial[(int64_t)kc * gridDim.x + blockIdx.x] = sh[0];
}
__global__ void ens_pick_kernel(const double *__restrict__ partial, int K, int P, int *__restrict__ sel) {
  const int ch = threadIdx.x;
  if (ch >= 2) return;
  int best = 0;
  double bt = 0.0;
  for (int k = 0; k < K; ++k) {
    double t = 0.0;
    for (int b = 0; b < P; ++b) t += partial[(int64_t)(k * 2 + ch) * P + b];
    if (k == 0) {
      bt = t;
    } else if (!(bt != bt) && ((t != t) || t < bt)) {   // np.argmin: the first NaN, else the first minimum
      best = k;
      bt = t;
    }
  }
  sel[ch] = best;
}
__global__ __launch_bounds__(256) void ens_take_kernel(const float *__restrict__ waves, int64_t N, const int *__restrict__ sel,
                                                       float *__restrict__ out) {
  const int ch = blockIdx.y;
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < N) out[(int64_t)ch * N + i] = waves[((int64_t)sel[ch] * 2 + ch) * N + i];
}

// librosa.stft frame t of channel ch of wave [2, n] (centre, zero padding) -> X[0 .. nh] in LDS
__device__ __forceinline__ void ens_frame_spectrum(const float *__restrict__ wave, int64_t n, int ch, int t, int hop,
                                                   const float *__restrict__ window, const float2 *__restrict__ tw,
                                                   const FftPlan &p, float2 *bufA, float2 *bufB, float2 *X) {
  float *fa = reinterpret_cast<float *>(bufA);
  for (int e = threadIdx.x; e < p.n_fft; e += blockDim.x) {
    const int64_t q = (int64_t)t * hop + e - p.nh;
    fa[e] = (q >= 0 && q < n) ? wave[(int64_t)ch * n + q] * window[e] : 0.f;
  }
  float2 *Z = fft_lds<-1>(bufA, bufB, p, tw);
  const int nh = p.nh;
  for (int k = threadIdx.x; k <= nh; k += blockDim.x) {
    const float2 zk = Z[k == nh ? 0 : k];
    float2 zc = Z[(k == 0 || k == nh) ? 0 : nh - k];
    zc.y = -zc.y;
    const float2 E = make_float2(0.5f * (zk.x + zc.x), 0.5f * (zk.y + zc.y));
    const float2 D = make_float2(0.5f * (zk.x - zc.x), 0.5f * (zk.y - zc.y));
    const float2 O = make_float2(D.y, -D.x);
    const float2 w = (k == nh) ? make_float2(-1.f, 0.f) : tw[k];
    X[k] = cadd(E, cmul(w, O));
  }
  __syncthreads();
}

// X[0 .. nh] in LDS -> windowed inverse frame (librosa.istft's ytmp), scaled by `sign`
__device__ __forceinline__ void ens_inverse_frame(float2 *X, float *__restrict__ dstf, const float *__restrict__ window,
                                                  const float2 *__restrict__ tw, const FftPlan &p, float2 *bufA, float2 *bufB,
                                                  float sign) {
  const int nh = p.nh;
  if (threadIdx.x == 0) {
    X[0].y = 0.f;
    X[nh].y = 0.f;
  }
  __syncthreads();
  for (int k = threadIdx.x; k < nh; k += blockDim.x) {
    const float2 xk = X[k];
    float2 xc = X[nh - k];
    xc.y = -xc.y;
    const float2 E = make_float2(0.5f * (xk.x + xc.x), 0.5f * (xk.y + xc.y));
    const float2 D = make_float2(0.5f * (xk.x - xc.x), 0.5f * (xk.y - xc.y));
    float2 w = tw[k];
    w.y = -w.y;
    const float2 O = cmul(w, D);
    bufA[k] = make_float2(E.x - O.y, E.y + O.x);
  }
  float2 *z = fft_lds<+1>(bufA, bufB, p, tw);
  const float scale = sign / (float)nh;
  float2 *dst = reinterpret_cast<float2 *>(dstf);
  const float2 *w2 = reinterpret_cast<const float2 *>(window);
  for (int m = threadIdx.x; m < nh; m += blockDim.x) {
    const float2 v = z[m];
    const float2 w = w2[m];
    dst[m] = make_float2((v.x * scale) * w.x, (v.y * scale) * w.y);
  }
}

// *_fft and uvr_*_spec ensembles (ensembler.py:120-156, spec_utils.ensembling :583-607): grid = (T, 2).
// LDS: bufA, bufB (nh each), cur (nh+1), sel (nh+1), and for the median K * (nh+1) spectra.
__global__ __launch_bounds__(256) void ens_fft_kernel(const float *__restrict__ waves, int K, int64_t n, int alg,
                                                      const double *__restrict__ weights, double wsum, int hop,
                                                      float *__restrict__ frames, const float *__restrict__ window,
                                                      const float2 *__restrict__ tw, FftPlan p) {
  extern __shared__ float2 lds[];
  const int nh = p.nh;
  float2 *bufA = lds, *bufB = lds + nh, *cur = lds + 2 * nh, *sel = cur + (nh + 1), *all = sel + (nh + 1);
  const int t = blockIdx.x, ch = blockIdx.y, T = gridDim.x;
  for (int k = 0; k < K; ++k) {
    float2 *dst = (alg == ENS_MEDIAN_FFT) ? all + (size_t)k * (nh + 1) : cur;
    ens_frame_spectrum(waves + (int64_t)k * 2 * n, n, ch, t, hop, window, tw, p, bufA, bufB, dst);
    if (alg == ENS_MEDIAN_FFT) continue;
    for (int b = threadIdx.x; b <= nh; b += blockDim.x) {
      const float2 x = cur[b];
      if (alg == ENS_AVG_FFT) {
        // ense_spec (complex64) += s * weight (float64 product)
        float2 a = k == 0 ? make_float2(0.f, 0.f) : sel[b];
        a.x = (float)((double)a.x + (double)x.x * weights[k]);
        a.y = (float)((double)a.y + (double)x.y * weights[k]);
        sel[b] = a;
      } else if (k == 0) {
        sel[b] = x;
      } else {
        const float2 s = sel[b];
        const float mx = hypotf(x.x, x.y), ms = hypotf(s.x, s.y);
        bool take;
        if (alg == ENS_MIN_FFT) take = mx < ms;            // np.argmin: first minimum
        else if (alg == ENS_MAX_FFT) take = mx > ms;       // np.argmax: first maximum
        else if (alg == ENS_UVR_MIN_SPEC) take = mx <= ms; // np.where(|new| <= |cur|, new, cur)
        else take = mx >= ms;
        if (take) sel[b] = x;
      }
    }
    __syncthreads();
  }
  if (alg == ENS_MEDIAN_FFT) {
    for (int b = threadIdx.x; b <= nh; b += blockDim.x) {
      float re[ENS_MAX_K], im[ENS_MAX_K];
      for (int k = 0; k < K; ++k) {
        const float2 x = all[(size_t)k * (nh + 1) + b];
        re[k] = x.x;
        im[k] = x.y;
      }
      sel[b] = make_float2(median_small(re, K), median_small(im, K));
    }
  } else if (alg == ENS_AVG_FFT) {
    for (int b = threadIdx.x; b <= nh; b += blockDim.x) {
      const float2 a = sel[b];
      sel[b] = make_float2((float)((double)a.x / wsum), (float)((double)a.y / wsum));
    }
  }
  __syncthreads();
  ens_inverse_frame(sel, frames + ((int64_t)ch * T + t) * p.n_fft, window, tw, p, bufA, bufB, 1.0f);
}

// invert_audio (spec_utils.py:557-571): v = Y - max(|X|, |Y|) * exp(j angle(X)); invert_stem returns -istft(v)
__global__ __launch_bounds__(256) void ens_invert_kernel(const float *__restrict__ mix, const float *__restrict__ stem, int64_t n,
                                                         int hop, float *__restrict__ frames, const float *__restrict__ window,
                                                         const float2 *__restrict__ tw, FftPlan p) {
  extern __shared__ float2 lds[];
  const int nh = p.nh;
  float2 *bufA = lds, *bufB = lds + nh, *X = lds + 2 * nh, *Y = X + (nh + 1);
  const int t = blockIdx.x, ch = blockIdx.y, T = gridDim.x;
  ens_frame_spectrum(mix, n, ch, t, hop, window, tw, p, bufA, bufB, X);
  ens_frame_spectrum(stem, n, ch, t, hop, window, tw, p, bufA, bufB, Y);
  for (int b = threadIdx.x; b <= nh; b += blockDim.x) {
    const float2 x = X[b], y = Y[b];
    const float xm = hypotf(x.x, x.y), ym = hypotf(y.x, y.y);
    const float mm = xm >= ym ? xm : ym;
    const float ang = atan2f(x.y, x.x);
    Y[b] = make_float2(y.x - mm * cosf(ang), y.y - mm * sinf(ang));
  }
  __syncthreads();
  ens_inverse_frame(Y, frames + ((int64_t)ch * T + t) * p.n_fft, window, tw, p, bufA, bufB, -1.0f);
}

}  // namespace asx
