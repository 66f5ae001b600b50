// VR (vocal-remover) kernels for gfx950: architectures/vr_separator.py:255-375, uvr_lib_v5/spec_utils.py:74-96,
// 180-222, 250-429, 472-492, uvr_lib_v5/vr_network/{nets,layers}.py.
//
// Net activations are channels-last [B, F, W, C] (frequency rows outer, frames inner); every conv of
// CascadedASPPNet is a gg_kernel launch (kernels_ht.h) with BatchNorm folded into the weights; concatenations are
// channel slices of one buffer that producers write in place.  The kernels here are the non-GEMM rest: bilinear
// x2 upsampling, the depthwise dilated 3x3, the ASPP pooling branch, the multiband STFT / iSTFT with librosa
// framing, the polyphase resampler, magnitude patches and mask algebra.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace asx {

// librosa.filters.window_sumsquare over T frames: out[i] = sum_t w[i - t * hop]^2, accumulated in float64 in increasing t like
// the host loop it replaces (bit-identical), rounded to float32.  One thread per output sample; at most n_fft / hop terms.
__global__ __launch_bounds__(256) void vr_wss_kernel(const float *__restrict__ window, int n_fft, int hop, int T, int64_t n,
                                                     float *__restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  int64_t t_lo = (i - n_fft + hop) / hop;          // ceil((i - n_fft + 1) / hop) for i - n_fft + 1 > 0
  if (i - n_fft + 1 <= 0) t_lo = 0;
  int64_t t_hi = i / hop;
  if (t_hi > T - 1) t_hi = T - 1;
  double acc = 0.0;
  for (int64_t t = t_lo; t <= t_hi; ++t) {
    const double w = (double)window[i - t * hop];
    acc += w * w;
  }
  out[i] = (float)acc;
}

// F.interpolate(scale_factor=2, mode="bilinear", align_corners=True) (layers.py:154): x [B, h, w, C] (row stride ldx)
// -> y [B, 2h, 2w, (ldy)] channel slice [0, C).  src = dst * (in - 1) / (out - 1).
__global__ __launch_bounds__(256) void vr_upsample2x_kernel(const float *__restrict__ x, int h, int w, int C, int ldx,
                                                            float *__restrict__ y, int ldy, int64_t total) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;   // over B * 2h * 2w * C/4
  if (idx >= total) return;
  const int c4 = C / 4;
  const int c = (int)(idx % c4) * 4;
  int64_t p = idx / c4;
  const int ow = (int)(p % (2 * w));
  p /= 2 * w;
  const int oh = (int)(p % (2 * h));
  const int64_t b = p / (2 * h);
  const float sh = (h > 1) ? (float)(h - 1) / (float)(2 * h - 1) : 0.f;
  const float sw = (w > 1) ? (float)(w - 1) / (float)(2 * w - 1) : 0.f;
  const float fh = sh * (float)oh, fw = sw * (float)ow;
  const int h0 = (int)fh, w0 = (int)fw;
  const int h1 = h0 + (h0 < h - 1 ? 1 : 0), w1 = w0 + (w0 < w - 1 ? 1 : 0);
  const float ah = fh - (float)h0, aw = fw - (float)w0;
  const float *xb = x + b * (int64_t)h * w * ldx;
  const float4 v00 = *reinterpret_cast<const float4 *>(xb + ((int64_t)h0 * w + w0) * ldx + c);
  const float4 v01 = *reinterpret_cast<const float4 *>(xb + ((int64_t)h0 * w + w1) * ldx + c);
  const float4 v10 = *reinterpret_cast<const float4 *>(xb + ((int64_t)h1 * w + w0) * ldx + c);
  const float4 v11 = *reinterpret_cast<const float4 *>(xb + ((int64_t)h1 * w + w1) * ldx + c);
  const float w00 = (1.f - ah) * (1.f - aw), w01 = (1.f - ah) * aw, w10 = ah * (1.f - aw), w11 = ah * aw;
  float4 o;
  o.x = w00 * v00.x + w01 * v01.x + w10 * v10.x + w11 * v11.x;
  o.y = w00 * v00.y + w01 * v01.y + w10 * v10.y + w11 * v11.y;
  o.z = w00 * v00.z + w01 * v01.z + w10 * v10.z + w11 * v11.z;
  o.w = w00 * v00.w + w01 * v01.w + w10 * v10.w + w11 * v11.w;
  *reinterpret_cast<float4 *>(y + ((b * 2 * h + oh) * (int64_t)(2 * w) + ow) * ldy + c) = o;
}

// depthwise 3x3, dilation d, padding d (layers.py:75-86): x [B, h, w, C] -> y [B, h, w, C]; wt [9][C] (tap-major)
__global__ __launch_bounds__(256) void vr_dwconv_kernel(const float *__restrict__ x, int h, int w, int C, int d,
                                                        const float *__restrict__ wt, float *__restrict__ y, int64_t total) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int c = (int)(idx % C);
  int64_t p = idx / C;
  const int iw = (int)(p % w);
  p /= w;
  const int ih = (int)(p % h);
  const int64_t b = p / h;
  const float *xb = x + b * (int64_t)h * w * C;
  float acc = 0.f;
#pragma unroll
  for (int kh = 0; kh < 3; ++kh) {
    const int hh = ih + (kh - 1) * d;
    if (hh < 0 || hh >= h) continue;
#pragma unroll
    for (int kw = 0; kw < 3; ++kw) {
      const int ww = iw + (kw - 1) * d;
      if (ww < 0 || ww >= w) continue;
      acc += wt[(kh * 3 + kw) * C + c] * xb[((int64_t)hh * w + ww) * C + c];
    }
  }
  y[idx] = acc;
}

// nn.AdaptiveAvgPool2d((1, None)) (layers.py:226): mean over the frequency rows, x [B, h, w, C] -> y [B, w, C]
__global__ __launch_bounds__(256) void vr_rowmean_kernel(const float *__restrict__ x, int h, int w, int C,
                                                         float *__restrict__ y, int64_t total) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;   // over B * w * C
  if (idx >= total) return;
  const int64_t wc = (int64_t)w * C;
  const int64_t b = idx / wc, r = idx - b * wc;
  const float *xp = x + b * h * wc + r;
  float acc = 0.f;
  for (int i = 0; i < h; ++i) acc += xp[(int64_t)i * wc];
  y[idx] = acc / (float)h;
}

// bilinear resize of a 1-row map back to (h, w) with align_corners=True is a broadcast over the rows (layers.py:265):
// y[b, i, j, 0:C] (row stride ldy) = x[b, j, :]
__global__ __launch_bounds__(256) void vr_bcast_rows_kernel(const float *__restrict__ x, int h, int w, int C,
                                                            float *__restrict__ y, int ldy, int64_t total) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;   // over B * h * w * C
  if (idx >= total) return;
  const int c = (int)(idx % C);
  int64_t p = idx / C;
  const int j = (int)(p % w);
  p /= w;
  const int64_t b = p / h;
  y[(idx / C) * ldy + c] = x[(b * w + j) * C + c];
}

// ---------------------------------------------------------------------------
// scipy.signal.resample_poly as librosa.resample(res_type="polyphase") calls it (upfirdn with the zero-padded
// Kaiser(5.0) low-pass):  y[m] = sum_i x[i] * hp[(m + n_pre_remove) * down - i * up],  m < n_out.
// acc64 = 1 accumulates in float64 (synthesis chain, float64 in the reference), else float32 (analysis, float32 input).
// grid.y = channel.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void vr_resample_kernel(const float *__restrict__ x, int64_t n_in, const float *__restrict__ hp,
                                                          const double *__restrict__ hp64, int hlen, int up, int down,
                                                          int n_pre_remove, float *__restrict__ y, int64_t n_out, int acc64) {
  const int ch = blockIdx.y;
  const int64_t m = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (m >= n_out) return;
  const int64_t q = (m + n_pre_remove) * down;
  int64_t i_hi = q / up;
  if (i_hi > n_in - 1) i_hi = n_in - 1;
  int64_t i_lo = (q - hlen + up) / up;   // ceil((q - hlen + 1) / up)
  if (q - hlen + 1 <= 0) i_lo = 0;
  const float *xp = x + (int64_t)ch * n_in;
  if (acc64) {
    double a = 0.0;
    for (int64_t i = i_hi; i >= i_lo; --i) a += (double)xp[i] * hp64[q - i * up];
    y[(int64_t)ch * n_out + m] = (float)a;
  } else {
    float a = 0.f;
    for (int64_t i = i_hi; i >= i_lo; --i) a += xp[i] * hp[q - i * up];
    y[(int64_t)ch * n_out + m] = a;
  }
}

// ---------------------------------------------------------------------------
// libsamplerate SRC_SINC_FASTEST as librosa.resample(res_type="sinc_fastest") reaches it (python-samplerate: float32 planar
// data interleaved into frames, src_simple, float32 out), restating src_sinc.c's sinc_stereo_vari_process / calc_output_stereo:
// output frame m sits at input position m / ratio = b + frac; the coefficient table (`tab`, TL floats at INC entries per input
// sample, `dtab` = float differences of neighbours) is walked in 20.12 fixed point from start_filter_index =
// lrint(frac * float_increment * 4096) in steps of `inc_fp` = lrint(float_increment * 4096), float_increment = INC * min(ratio, 1),
// entries linearly interpolated in double; the LEFT half (frames b - k, farthest tap first, down to filter index >= 0) and the
// RIGHT half (frames b + 1 + k, farthest first, while filter index > 0) are accumulated in double in the library's order and
// summed, times `scale` = min(ratio, 1).  History before frame 0 and after the last frame is zero (prepare_data).  Frames
// m >= n_gen (the frames src_simple really generates, computed by the host) are librosa's fix_length zero padding.  The
// position is formed from the exact rational m * down / up -- or, for an irrational ratio (up == 0: the pitch-shift round
// trip of the MDXC path), from m * pos_step in double -- rather than by the library's running double sum (equal up to the
// sum's rounding, ~1e-13 frames for rational ratios, ~1e-9 at ten million frames otherwise).
// Table in LDS: the threads of a block walk it at a common stride from a handful of distinct start indices.  grid.y = channel.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void vr_sinc_kernel(const float *__restrict__ x, int64_t n_in, const float *__restrict__ tab,
                                                      const float *__restrict__ dtab, int TL, int half_len, int up, int down,
                                                      double pos_step, double float_inc, long long inc_fp, double scale, int64_t n_gen,
                                                      float *__restrict__ y, int64_t n_out) {
  extern __shared__ float sinc_lds[];
  float *c = sinc_lds, *dc = sinc_lds + TL;
  for (int i = threadIdx.x; i < TL; i += blockDim.x) {
    c[i] = tab[i];
    dc[i] = dtab[i];
  }
  __syncthreads();
  const int ch = blockIdx.y;
  const int64_t m = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (m >= n_out) return;
  float *yp = y + (int64_t)ch * n_out;
  if (m >= n_gen) {
    yp[m] = 0.f;
    return;
  }
  const float *xp = x + (int64_t)ch * n_in;
  int64_t b;
  double frac;
  if (up > 0) {
    const int64_t q = m * down;
    b = q / up;
    frac = (double)(q - b * up) / (double)up;
  } else {
    const double p = (double)m * pos_step;
    b = (int64_t)floor(p);
    frac = p - (double)b;
  }
  const long long max_fi = (long long)half_len << 12;
  const long long sfi = llrint(frac * float_inc * 4096.0);
  auto icoeff = [&](long long fi) -> double {
    const int idx = (int)(fi >> 12);
    return (double)c[idx] + (double)(fi & 4095) * (1.0 / 4096.0) * (double)dc[idx];
  };
  double left = 0.0, right = 0.0;
  {
    const long long cc = (max_fi - sfi) / inc_fp;
    long long fi = sfi + cc * inc_fp;
    int64_t i = b - cc;
    do {
      if (i >= 0 && i < n_in) left += icoeff(fi) * (double)xp[i];
      fi -= inc_fp;
      ++i;
    } while (fi >= 0);
  }
  {
    long long fi = inc_fp - sfi;
    const long long cr = (max_fi - fi) / inc_fp;
    fi += cr * inc_fp;
    int64_t i = b + 1 + cr;
    do {
      if (i >= 0 && i < n_in) right += icoeff(fi) * (double)xp[i];
      fi -= inc_fp;
      --i;
    } while (fi > 0);
  }
  yp[m] = (float)(scale * (left + right));
}

// ---------------------------------------------------------------------------
// librosa.stft (centre, zero padding, periodic Hann) of one band, written straight into the combined spectrogram
// (combine_spectrograms, spec_utils.py:250-281): bins [crop_start, crop_stop) * gain[bin] -> rows row_off + ...
// of X [2, Tmin, bins+1] (bin fastest).  Channel conversion of wave_to_spectrogram (spec_utils.py:289-300) on load:
// mode 0 L/R, 1 mid_side ((L+R)/2, L-R), 2 mid_side_b2 (R + L/2, L - R/2), 3 reverse; VR 5.1 convert_channels
// (spec_utils.py:232-247, applied to the waveform -- the STFT is linear): 4 mid_side_c, 5 stereo_n.  grid = (Tmin, 2).
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void vr_stft_kernel(const float *__restrict__ wave, int64_t n, int hop, int mode,
                                                      int crop_start, int crop_stop, int row_off, int nbins1,
                                                      const float *__restrict__ gain, float2 *__restrict__ X,
                                                      const float *__restrict__ window, const float2 *__restrict__ tw,
                                                      FftPlan p, float2 *__restrict__ he, int he_n) {
  extern __shared__ float2 lds[];
  float2 *bufA = lds;
  float2 *bufB = lds + p.nh;
  const int t = blockIdx.x, ch = blockIdx.y, T = gridDim.x;
  float *fa = reinterpret_cast<float *>(bufA);
  for (int e = threadIdx.x; e < p.n_fft; e += blockDim.x) {
    int64_t q = (int64_t)t * hop + e - p.nh;
    float v = 0.f;
    if (q >= 0 && q < n) {
      if (mode == 3) q = n - 1 - q;
      const float l = wave[q], r = wave[n + q];
      if (mode == 1) v = ch == 0 ? (l + r) / 2.0f : l - r;
      else if (mode == 2) v = ch == 0 ? r + l * 0.5f : l - r * 0.5f;
      else if (mode == 4) v = ch == 0 ? l + r * 0.25f : r - l * 0.25f;                       // mid_side_c (spec_utils.py:235)
      else if (mode == 5) v = ch == 0 ? (l + r * 0.25f) / 0.9375f : (r + l * 0.25f) / 0.9375f;   // stereo_n (:241)
      else v = ch == 0 ? l : r;
    }
    fa[e] = v * window[e];
  }
  float2 *Z = fft_lds<-1>(bufA, bufB, p, tw);
  const int nh = p.nh;
  for (int k = crop_start + threadIdx.x; k < crop_stop; k += blockDim.x) {
    const float2 zk = Z[k == nh ? 0 : k];
    float2 zc = Z[(k == 0 || k == nh) ? 0 : nh - k];
    zc.y = -zc.y;
    const float2 E = make_float2(0.5f * (zk.x + zc.x), 0.5f * (zk.y + zc.y));
    const float2 D = make_float2(0.5f * (zk.x - zc.x), 0.5f * (zk.y - zc.y));
    const float2 O = make_float2(D.y, -D.x);
    const float2 w = (k == nh) ? make_float2(-1.f, 0.f) : tw[k];
    float2 v = cadd(E, cmul(w, O));
    const int row = row_off + k - crop_start;
    const float g = gain[row];
    X[((int64_t)ch * T + t) * nbins1 + row] = make_float2(v.x * g, v.y * g);
  }
  // high_end_process (vr_separator.py:286-288): keep bins [n_fft/2 - he_n, n_fft/2) of the top band, he [2, T, he_n]
  if (he != nullptr)
    for (int i = threadIdx.x; i < he_n; i += blockDim.x) {
      const int k = nh - he_n + i;
      const float2 zk = Z[k];
      float2 zc = Z[k == 0 ? 0 : nh - k];
      zc.y = -zc.y;
      const float2 E = make_float2(0.5f * (zk.x + zc.x), 0.5f * (zk.y + zc.y));
      const float2 D = make_float2(0.5f * (zk.x - zc.x), 0.5f * (zk.y - zc.y));
      const float2 O = make_float2(D.y, -D.x);
      he[((int64_t)ch * T + t) * he_n + i] = cadd(E, cmul(tw[k], O));
    }
}

// max |X| (np.abs of complex64 = hypotf) -> float bits via atomicMax; X has n complex elements
__global__ __launch_bounds__(256) void vr_absmax_kernel(const float2 *__restrict__ X, int64_t n, unsigned int *peak_bits) {
  float m = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const float2 v = X[i];
    m = fmaxf(m, hypotf(v.x, v.y));
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
  __shared__ float wm[4];
  if ((threadIdx.x & 63) == 0) wm[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) atomicMax(peak_bits, __float_as_uint(fmaxf(fmaxf(wm[0], wm[1]), fmaxf(wm[2], wm[3]))));
}

// magnitude patches (vr_separator.py:294-303,344-347): patch k covers padded frames [k*roi, k*roi + W); padded frame p is
// frame p - pad_l (0 outside);  out [B, F=max_bin, W, 4] = (|X_L| / peak, |X_R| / peak, 0, 0).
__global__ __launch_bounds__(256) void vr_patch_kernel(const float2 *__restrict__ X, int T, int nbins1, int max_bin, int W,
                                                       int k0, int roi, int pad_l, const unsigned int *__restrict__ peak_bits,
                                                       float *__restrict__ out, int ld, int64_t total) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;   // over B * max_bin * W
  if (idx >= total) return;
  const int tl = (int)(idx % W);
  int64_t p = idx / W;
  const int f = (int)(p % max_bin);
  const int b = (int)(p / max_bin);
  const int t = (k0 + b) * roi + tl - pad_l;
  float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
  if (t >= 0 && t < T) {
    const float peak = __uint_as_float(*peak_bits);
    const float2 l = X[(int64_t)t * nbins1 + f], r = X[((int64_t)T + t) * nbins1 + f];
    o.x = hypotf(l.x, l.y) / peak;
    o.y = hypotf(r.x, r.y) / peak;
  }
  *reinterpret_cast<float4 *>(out + idx * ld) = o;
}

// mask assembly (nets.py:152-153 replicate pad, predict_mask :168-173, vr_separator.py:318-320,352-359):
// net output m [B, max_bin, W, 4] (sigmoid applied) -> M [2, T, bins+1]; local frame tl in [offset, W - offset) of patch
// k0 + b is frame (k0 + b) * roi + tl - offset - shift.  tta = 1: M = (M + m) * 0.5.
__global__ __launch_bounds__(256) void vr_mask_kernel(const float *__restrict__ m, int B, int max_bin, int W, int offset,
                                                      int k0, int roi, int shift, int T, int nbins1, int tta,
                                                      float *__restrict__ M, int64_t total) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;   // over B * roi_w * nbins1 ; f fastest
  if (idx >= total) return;
  const int roi_w = W - 2 * offset;
  const int f = (int)(idx % nbins1);
  int64_t p = idx / nbins1;
  const int tl = (int)(p % roi_w);
  const int b = (int)(p / roi_w);
  const int t = (k0 + b) * roi + tl - shift;
  if (t < 0 || t >= T) return;
  const int fs = f < max_bin ? f : max_bin - 1;
  const float2 v = *reinterpret_cast<const float2 *>(m + (((int64_t)b * max_bin + fs) * W + tl + offset) * 4);
  float *d0 = M + (int64_t)t * nbins1 + f, *d1 = M + ((int64_t)T + t) * nbins1 + f;
  if (tta) {
    *d0 = (*d0 + v.x) * 0.5f;
    *d1 = (*d1 + v.y) * 0.5f;
  } else {
    *d0 = v.x;
    *d1 = v.y;
  }
}

// adjust_aggr (spec_utils.py:472-492): mask[ch, f < split] **= e_lo[ch], mask[ch, f >= split] **= e_hi[ch]
__global__ __launch_bounds__(256) void vr_aggr_kernel(float *__restrict__ M, int T, int nbins1, int split, float lo0, float hi0,
                                                      float lo1, float hi1, int64_t total) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int f = (int)(idx % nbins1);
  const int ch = (int)(idx / ((int64_t)T * nbins1));
  const float e = f < split ? (ch ? lo1 : lo0) : (ch ? hi1 : hi0);
  M[idx] = powf(M[idx], e);
}

// per-frame min over (channel, bin) of the mask (merge_artifacts, spec_utils.py:187); one wave per frame
__global__ __launch_bounds__(256) void vr_frame_min_kernel(const float *__restrict__ M, int T, int nbins1, float *__restrict__ fmin) {
  const int lane = threadIdx.x & 63;
  const int t = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (t >= T) return;
  float m = INFINITY;
  for (int ch = 0; ch < 2; ++ch)
    for (int f = lane; f < nbins1; f += 64) m = fminf(m, M[((int64_t)ch * T + t) * nbins1 + f]);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) m = fminf(m, __shfl_xor(m, off));
  if (lane == 0) fmin[t] = m;
}

// y_mask += weight[t] * (1 - y_mask) (spec_utils.py:213-214)
__global__ __launch_bounds__(256) void vr_merge_kernel(float *__restrict__ M, int T, int nbins1, const float *__restrict__ weight,
                                                       int64_t total) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int t = (int)((idx / nbins1) % T);
  const float y = M[idx];
  M[idx] = y + weight[t] * (1.0f - y);
}

// ---------------------------------------------------------------------------
// Synthesis of one band (cmb_spectrogram_to_wave, spec_utils.py:341-396): spectrum rows [crop_start, crop_stop) =
// (which ? 1 - mask : mask) * X rows [row_off, ...) * gain[bin] (y_spec / v_spec of vr_separator.py:335-336 and the
// fft_hp / fft_lp filters folded into `gain`), non-finite values -> 0 (np.nan_to_num, :183-184); inverse FFT * window ->
// frames [2, T, n_fft].  grid = (T, 2).
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void vr_istft_kernel(const float2 *__restrict__ X, const float *__restrict__ M, int which,
                                                       int nbins1, int crop_start, int crop_stop, int row_off,
                                                       const float *__restrict__ gain, float *__restrict__ frames,
                                                       const float *__restrict__ window, const float2 *__restrict__ tw,
                                                       FftPlan p, const float2 *__restrict__ he, int he_n, int mirror_row) {
  extern __shared__ float2 lds[];
  float2 *bufA = lds;
  float2 *bufB = lds + p.nh;
  float2 *bufX = lds + 2 * p.nh;
  const int t = blockIdx.x, ch = blockIdx.y, T = gridDim.x;
  const int nh = p.nh;
  for (int k = threadIdx.x; k <= nh; k += blockDim.x) {
    float2 v = make_float2(0.f, 0.f);
    if (k >= crop_start && k < crop_stop) {
      const int64_t src = ((int64_t)ch * T + t) * nbins1 + row_off + k - crop_start;
      float mk = 1.0f;
      if (M != nullptr) {
        mk = M[src];
        if (which) mk = 1.0f - mk;
      }
      const float2 x = X[src];
      const float g = gain[k];
      v = make_float2(mk * x.x * g, mk * x.y * g);
      if (!isfinite(v.x)) v.x = 0.f;
      if (!isfinite(v.y)) v.y = 0.f;
    }
    if (he != nullptr && k >= nh - he_n && k < nh) {
      // spec_utils.mirroring("mirroring") + the extra_bins splice of cmb_spectrogram_to_wave (:351-354): bin i of the
      // kept high end competes with the flipped magnitude of the separated spectrogram below pre_filter_start - 10,
      // carried on the input's phase; the smaller magnitude wins
      const int i = k - (nh - he_n);
      const int64_t src = ((int64_t)ch * T + t) * nbins1 + mirror_row + (he_n - 1 - i);
      float mk = 1.0f;
      if (M != nullptr) {
        mk = M[src];
        if (which) mk = 1.0f - mk;
      }
      const float2 x = X[src];
      float2 sm = make_float2(mk * x.x, mk * x.y);
      if (!isfinite(sm.x)) sm.x = 0.f;
      if (!isfinite(sm.y)) sm.y = 0.f;
      const float mag = hypotf(sm.x, sm.y);
      const float2 ie = he[((int64_t)ch * T + t) * he_n + i];
      const float ia = hypotf(ie.x, ie.y);
      const float ang = atan2f(ie.y, ie.x);
      const float2 mir = make_float2(mag * cosf(ang), mag * sinf(ang));
      const float2 o = (ia <= hypotf(mir.x, mir.y)) ? ie : mir;
      const float g = gain[k];
      v = make_float2(o.x * g, o.y * g);
    }
    if (k == 0 || k == nh) v.y = 0.f;
    bufX[k] = v;
  }
  __syncthreads();
  for (int k = threadIdx.x; k < nh; k += blockDim.x) {
    const float2 xk = bufX[k];
    float2 xc = bufX[nh - k];
    xc.y = -xc.y;
    const float2 E = make_float2(0.5f * (xk.x + xc.x), 0.5f * (xk.y + xc.y));
    const float2 D = make_float2(0.5f * (xk.x - xc.x), 0.5f * (xk.y - xc.y));
    float2 w = tw[k];
    w.y = -w.y;
    const float2 O = cmul(w, D);
    bufA[k] = make_float2(E.x - O.y, E.y + O.x);
  }
  float2 *z = fft_lds<+1>(bufA, bufB, p, tw);
  const float scale = 1.0f / (float)nh;
  float2 *dst = reinterpret_cast<float2 *>(frames + ((int64_t)ch * T + t) * p.n_fft);
  const float2 *w2 = reinterpret_cast<const float2 *>(window);
  for (int m = threadIdx.x; m < nh; m += blockDim.x) {
    const float2 v = z[m];
    const float2 w = w2[m];
    dst[m] = make_float2((v.x * scale) * w.x, (v.y * scale) * w.y);
  }
}

// librosa.istft fold: y[j] = sum_t frames[t][j + n/2 - t*hop] / wss[j + n/2], length hop*(T-1); wss = squared-window sum
// over the T frames (computed on the host).  Then the channel conversion of spectrogram_to_wave (spec_utils.py:331-336)
// and `wave = np.add(wave, ...)` with the lower bands' resampled sum (`lower`, may be null).  grid.x over samples.
__global__ __launch_bounds__(256) void vr_ola_kernel(const float *__restrict__ frames, const float *__restrict__ wss, int n_fft,
                                                     int hop, int T, int64_t len, int mode, const float *__restrict__ lower,
                                                     float *__restrict__ out) {
  const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= len) return;
  const int64_t m = j + n_fft / 2;
  int64_t t_lo = (m - n_fft + hop) / hop;
  if (m - n_fft + 1 <= 0) t_lo = 0;
  int64_t t_hi = m / hop;
  if (t_hi > T - 1) t_hi = T - 1;
  float a[2];
  for (int ch = 0; ch < 2; ++ch) {
    const float *fr = frames + (int64_t)ch * T * n_fft;
    float acc = 0.f;
    for (int64_t t = t_lo; t <= t_hi; ++t) acc += fr[t * n_fft + (m - t * hop)];
    const float s = wss[m];
    a[ch] = s > 1.17549435e-38f ? acc / s : acc;
  }
  float l = a[0], r = a[1];
  int64_t jo = j;
  if (mode == 1) {
    l = a[0] + a[1] / 2.0f;
    r = a[0] - a[1] / 2.0f;
  } else if (mode == 2) {
    l = a[1] / 1.25f + 0.4f * a[0];
    r = a[0] / 1.25f - 0.4f * a[1];
  } else if (mode == 3) {
    jo = len - 1 - j;
  } else if (mode == 4) {   // mid_side_c (spec_utils.py:325)
    l = a[0] / 1.0625f - a[1] / 4.25f;
    r = a[1] / 1.0625f + a[0] / 4.25f;
  } else if (mode == 5) {   // stereo_n (:329)
    l = a[0] - a[1] * 0.25f;
    r = a[1] - a[0] * 0.25f;
  }
  if (lower != nullptr) {
    l += lower[jo];
    r += lower[len + jo];
  }
  out[jo] = l;
  out[len + jo] = r;
}

// test hook layouts: x [B, 2, nb1, W] (reference NCHW, rows >= max_bin ignored) -> hc [B, max_bin, W, ld] channels 0..3
__global__ __launch_bounds__(256) void vr_from_nchw_kernel(const float *__restrict__ x, int nb1, int max_bin, int W, int ld,
                                                           float *__restrict__ hc, int64_t total) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;   // over B * max_bin * W
  if (idx >= total) return;
  const int t = (int)(idx % W);
  int64_t p = idx / W;
  const int f = (int)(p % max_bin);
  const int64_t b = p / max_bin;
  float4 o = make_float4(x[((b * 2) * nb1 + f) * W + t], x[((b * 2 + 1) * nb1 + f) * W + t], 0.f, 0.f);
  *reinterpret_cast<float4 *>(hc + idx * ld) = o;
}

// m [B, max_bin, W, 4] -> y [B, 2, nb1, W] with the last row replicated (nets.py:153)
__global__ __launch_bounds__(256) void vr_to_nchw_kernel(const float *__restrict__ m, int nb1, int max_bin, int W,
                                                         float *__restrict__ y, int64_t total) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;   // over B * 2 * nb1 * W
  if (idx >= total) return;
  const int t = (int)(idx % W);
  int64_t p = idx / W;
  const int f = (int)(p % nb1);
  p /= nb1;
  const int ch = (int)(p % 2);
  const int64_t b = p / 2;
  const int fs = f < max_bin ? f : max_bin - 1;
  y[idx] = m[((b * max_bin + fs) * W + t) * 4 + ch];
}

// ---------------------------------------------------------------------------
// VR 5.1 LSTMModule (layers_new.py:129-149).
// ---------------------------------------------------------------------------
// hc [B, H, W, ld] channel 0 -> xs [W, B, H]   (hidden.permute(2, 0, 1): nframes, N, nbins)
__global__ __launch_bounds__(256) void vr_lstm_in_kernel(const float *__restrict__ hc, int B, int H, int W, int ld,
                                                         float *__restrict__ xs, int64_t total) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;   // over W * B * H, h fastest
  if (idx >= total) return;
  const int h = (int)(idx % H);
  int64_t p = idx / H;
  const int b = (int)(p % B);
  const int t = (int)(p / B);
  xs[idx] = hc[(((int64_t)b * H + h) * W + t) * ld];
}

// ys [W, B, H] -> dst[b, h, t, ch0] (row stride ld), channels ch0+1 .. ch0+3 zeroed (padding of the concat slot)
__global__ __launch_bounds__(256) void vr_lstm_out_kernel(const float *__restrict__ ys, int B, int H, int W, float *__restrict__ dst,
                                                          int ld, int ch0, int64_t total) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;   // over B * H * W, t fastest
  if (idx >= total) return;
  const int t = (int)(idx % W);
  int64_t p = idx / W;
  const int h = (int)(p % H);
  const int b = (int)(p / H);
  *reinterpret_cast<float4 *>(dst + idx * ld + ch0) = make_float4(ys[((int64_t)t * B + b) * H + h], 0.f, 0.f, 0.f);
}

// one direction of nn.LSTM over the whole sequence: xp [W, B, 2, 4*hs] = x W_ih^T + b_ih + b_hh (gate order i, f, g, o),
// whh [2, 4*hs, hs]; out [W, B, 2*hs] (forward half | reverse half).  grid = (B, 2), block = 4*hs threads (hs <= 128):
// thread g keeps row g of W_hh in registers, h and the gate pre-activations live in LDS.
template <int HS>
__global__ __launch_bounds__(4 * HS) void vr_lstm_seq_kernel(const float *__restrict__ xp, const float *__restrict__ whh, int W, int B,
                                                             float *__restrict__ out) {
  __shared__ float hsh[HS];
  __shared__ float gsh[4 * HS];
  const int b = blockIdx.x, dir = blockIdx.y, g = threadIdx.x;
  float wr[HS];
#pragma unroll
  for (int k = 0; k < HS; ++k) wr[k] = whh[((int64_t)dir * 4 * HS + g) * HS + k];
  if (g < HS) hsh[g] = 0.f;
  float c = 0.f;
  __syncthreads();
  for (int s = 0; s < W; ++s) {
    const int t = dir ? W - 1 - s : s;
    float a = xp[(((int64_t)t * B + b) * 2 + dir) * 4 * HS + g];
#pragma unroll
    for (int k = 0; k < HS; ++k) a += wr[k] * hsh[k];
    gsh[g] = a;
    __syncthreads();
    if (g < HS) {
      const float ig = 1.0f / (1.0f + expf(-gsh[g])), fg = 1.0f / (1.0f + expf(-gsh[HS + g]));
      const float gg = tanhf(gsh[2 * HS + g]), og = 1.0f / (1.0f + expf(-gsh[3 * HS + g]));
      c = fg * c + ig * gg;
      const float h = og * tanhf(c);
      hsh[g] = h;
      out[((int64_t)t * B + b) * 2 * HS + dir * HS + g] = h;
    }
    __syncthreads();
  }
}

}  // namespace asx
