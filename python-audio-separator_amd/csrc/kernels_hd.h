// Demucs v3 (HDemucs) kernels that the v4 set (kernels_ht.h) does not have: channel-group GroupNorm of the two
// innermost layers, the BLSTM and LocalState inserts of their DConv branches (uvr_lib_v5/demucs/demucs.py:19-66,
// 152-221) and the odd-length waveform input.  All fp32, channels-last.
#pragma once

namespace asx {

// waveform branch input (hdemucs.py:700-704): seg [B, 2, L] -> xt [B, Lp, 2] = (seg - mean) / (1e-5 + std), Lp = L
// rounded up to even; the pad sample is zero (the first encoder zero-pads to a multiple of its stride, hdemucs.py:147-149)
__global__ __launch_bounds__(256) void hd_time_norm_kernel(const float *__restrict__ seg, int64_t L, int64_t Lp,
                                                           const double *__restrict__ acc, float *__restrict__ xt) {
  const int64_t b = blockIdx.y;
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= Lp) return;
  float2 o = make_float2(0.f, 0.f);
  if (i < L) {
    float mean, stdv;
    sample_mean_std(acc, b, (double)(2 * L), mean, stdv);
    const float d = 1e-5f + stdv;
    o = make_float2((seg[(b * 2) * L + i] - mean) / d, (seg[(b * 2 + 1) * L + i] - mean) / d);
  }
  reinterpret_cast<float2 *>(xt)[b * Lp + i] = o;
}

// GroupNorm(G, C) over x [B, Rin, C] (Rin = every non-channel position), statistics from gstats_kernel
// (acc[(b*G + g)*2 + {sum, sum of squares}]); dst [B, R, C] keeps rows [r0, r0 + R) -- HDecLayer normalises the whole
// transposed-conv output and crops afterwards (hdemucs.py:321-327), so the cropped rows still count in the statistics.
//   mode 0: dst = gelu(norm(x)) (+ skip)        HEncLayer.norm1 / HDecLayer.norm2 (hdemucs.py:161, 322-329)
//   mode 1: dst[.., c] = norm(x)[c] * sigmoid(norm(x)[c + C/2])   GLU after norm2 / norm1 (hdemucs.py:169, 315)
//   mode 2: dst = norm(x) (+ skip)              the last decoder has no GELU
__global__ __launch_bounds__(256) void hd_gn_kernel(const float *__restrict__ x, int64_t Rin, int64_t r0, int64_t R, int C, int G,
                                                    const double *__restrict__ acc, const float *__restrict__ gam,
                                                    const float *__restrict__ bet, int mode, float *__restrict__ dst,
                                                    const float *__restrict__ skip, int64_t total) {
  const int Ce = mode == 1 ? C / 2 : C;
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;            // total = B*R*Ce
  const int c = (int)(idx % Ce);
  const int64_t row = idx / Ce;        // (b, r)
  const int64_t b = row / R;
  const int cg = C / G;
  const double cnt = (double)Rin * cg;
  float mean, rstd;
  group_mean_rstd(acc, b * G + c / cg, cnt, 1e-5f, mean, rstd);
  const float *xp = x + (b * Rin + (row - b * R) + r0) * C;
  float v = (xp[c] - mean) * rstd * gam[c] + bet[c];
  if (mode == 1) {
    float m2, r2;
    group_mean_rstd(acc, b * G + (c + Ce) / cg, cnt, 1e-5f, m2, r2);
    const float g = (xp[c + Ce] - m2) * r2 * gam[c + Ce] + bet[c + Ce];
    v = v * (1.0f / (1.0f + expf(-g)));
  } else if (mode == 0) {
    v = 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f));
  }
  if (skip != nullptr) v += skip[idx];
  dst[idx] = v;
}

// ---------------------------------------------------------------------------
// BLSTM (demucs.py:19-66): sequences longer than max_steps = 200 are cut into frames of 200 with stride 100
// (utils.unfold, zero padded), every frame is an independent sequence; the output keeps the middle of each frame.
// Sequence-major rows: row = step * Ntot + n_off + n with n = b * nfr + k; Ntot counts the sequences of every chunk group
// that shares the recurrence launches (engine_hd.h), n_off is this group's first sequence.
// ---------------------------------------------------------------------------
// h [B, T, H] -> xs [steps * Ntot, H]
__global__ __launch_bounds__(256) void hd_lstm_frame_kernel(const float *__restrict__ h, int B, int T, int H, int nfr,
                                                            int steps, int fstride, int Ntot, int n_off,
                                                            float *__restrict__ xs, int64_t total) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;   // over steps * N * H
  if (idx >= total) return;
  const int c = (int)(idx % H);
  int64_t p = idx / H;
  const int N = B * nfr;
  const int n = (int)(p % N);
  const int s = (int)(p / N);
  const int b = n / nfr, k = n - b * nfr;
  const int t = k * fstride + s;
  xs[((int64_t)s * Ntot + n_off + n) * H + c] = t < T ? h[((int64_t)b * T + t) * H + c] : 0.f;
}

// h [B, T, H] += lin rows picked as BLSTM.forward stitches them (demucs.py:50-64): frame 0 gives steps [0, 150),
// the last frame [50, 200), the others [50, 150); unframed (nfr == 1): row t.
__global__ __launch_bounds__(256) void hd_lstm_unframe_kernel(const float *__restrict__ lin, int B, int T, int H, int nfr,
                                                              int fstride, int Ntot, int n_off, float *__restrict__ h,
                                                              int64_t total) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;   // over B * T * H
  if (idx >= total) return;
  const int c = (int)(idx % H);
  int64_t p = idx / H;
  const int t = (int)(p % T);
  const int b = (int)(p / T);
  int k = 0;
  if (nfr > 1 && t >= fstride + fstride / 2) {
    k = (t - fstride / 2) / fstride;
    if (k > nfr - 1) k = nfr - 1;
  }
  const int s = t - k * fstride;
  h[idx] += lin[((int64_t)s * Ntot + n_off + b * nfr + k) * H + c];
}

// One time step of a bidirectional nn.LSTM layer for N sequences (gate order i, f, g, o):
//   gates = xp[t_dir][n][dir] + W_hh[dir] h_prev[dir][n];  c = f*c + i*g;  h = o * tanh(c)
// xp [steps*N, 2, 4H] (input projection + b_ih + b_hh, one GEMM for all steps), whh [2, 4H, H], hprev / hnext / cst
// [2, N, H], out [steps*N, 2H].  Direction 0 handles step s, direction 1 step steps-1-s.
// grid = (H / 4, 2, nz), tpz <= 8 sequence tiles per z block: a workgroup owns 4 hidden units -- 16 gate rows, the M side of v_mfma_f32_16x16x4_f32
// with row = unit*4 + gate, so that a lane's four accumulators are the four gates of one (unit, sequence) -- for up to
// 8 tiles of 16 sequences; its 4 waves split K, fragments come straight from L2 as 16-byte rows (the k order inside a
// 16-wide chunk is permuted identically for both operands), partial sums meet in LDS.  The recurrence is a chain of
// tiny GEMMs (N x 4H x H per direction); the kernel is latency-bound, so it keeps every load of a wave independent.
__global__ __launch_bounds__(256) void hd_lstm_step_kernel(const float *__restrict__ xp, const float *__restrict__ whh,
                                                           const float *__restrict__ hprev, float *__restrict__ hnext,
                                                           float *__restrict__ cst, float *__restrict__ out, int N, int H,
                                                           int s, int steps, int tpz) {
  __shared__ f32x4 red[4][8][64];
  const int u0 = blockIdx.x * 4, dir = blockIdx.y, nb = blockIdx.z * tpz * 16;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int m = lane & 15, kq = lane >> 4;
  const int t = dir ? steps - 1 - s : s;
  const int ntile = (N - nb + 15) / 16 < tpz ? (N - nb + 15) / 16 : tpz;
  const int kc = ((H / 16 + 3) / 4) * 16;
  const int k_lo = wave * kc, k_hi = (k_lo + kc < H) ? k_lo + kc : H;
  const float *wrow = whh + ((int64_t)dir * 4 * H + (int64_t)(m & 3) * H + u0 + (m >> 2)) * H + kq * 4;
  const float *hbase = hprev + (int64_t)dir * N * H + kq * 4;
  // epilogue operands of this thread's (sequence, unit) items: issued before the GEMM so that their HBM latency (xp is
  // streamed once) hides behind it
  float xin[2][4], cprev[2];
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int i = threadIdx.x + q * 256;
    const int j = i >> 6, l = i & 63;
    const int n = nb + j * 16 + (l & 15), u = u0 + (l >> 4);
    const bool ok = j < ntile && n < N;
    const float *xr = xp + (((int64_t)t * N + (ok ? n : 0)) * 2 + dir) * 4 * H + u;
#pragma unroll
    for (int g = 0; g < 4; ++g) xin[q][g] = ok ? xr[g * H] : 0.f;
    cprev[q] = ok ? cst[((int64_t)dir * N + n) * H + u] : 0.f;
  }
  f32x4 acc[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int k = k_lo; k < k_hi; k += 16) {
    const f32x4 a = *reinterpret_cast<const f32x4 *>(wrow + k);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      if (j < ntile) {
        int n = nb + j * 16 + m;
        if (n >= N) n = N - 1;   // padding columns: computed, never stored
        const f32x4 b = *reinterpret_cast<const f32x4 *>(hbase + (int64_t)n * H + k);
        acc[j] = ASX_MFMA(a.x, b.x, acc[j]);
        acc[j] = ASX_MFMA(a.y, b.y, acc[j]);
        acc[j] = ASX_MFMA(a.z, b.z, acc[j]);
        acc[j] = ASX_MFMA(a.w, b.w, acc[j]);
      }
    }
  }
#pragma unroll
  for (int j = 0; j < 8; ++j)
    if (j < ntile) red[wave][j][lane] = acc[j];
  __syncthreads();
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int i = threadIdx.x + q * 256;
    const int j = i >> 6, l = i & 63;
    const int n = nb + j * 16 + (l & 15), u = u0 + (l >> 4);
    if (j >= ntile || n >= N) continue;
    const f32x4 g = (red[0][j][l] + red[1][j][l]) + (red[2][j][l] + red[3][j][l]);
    const float gi = g.x + xin[q][0], gf = g.y + xin[q][1], gg = g.z + xin[q][2], go = g.w + xin[q][3];
    const float ig = 1.0f / (1.0f + expf(-gi)), fg = 1.0f / (1.0f + expf(-gf)), og = 1.0f / (1.0f + expf(-go));
    const int64_t si = ((int64_t)dir * N + n) * H + u;
    const float c = fg * cprev[q] + ig * tanhf(gg);
    cst[si] = c;
    const float hv = og * tanhf(c);
    hnext[si] = hv;
    out[((int64_t)t * N + n) * 2 * H + dir * H + u] = hv;
  }
}

// ---------------------------------------------------------------------------
// LocalState (demucs.py:152-221, heads = 4, ndecay = 4, nfreqs = 0): for every query s
//   score[t] = k_t . q_s / sqrt(dh) - |t - s| * D_s,   D_s = 1/4 * sum_f (f + 1) * sigmoid(decay_f[s]),   score[s] = -100
//   out[s] = sum_t softmax_t(score)[t] * content_t
// qkvd [B*T, ld]: query | key | content (H channels each, head h owns [h*DH, (h+1)*DH)) | decay logits [heads*4].
// One thread per query, keys staged through LDS in tiles of 32; grid = (ceil(T / 64), heads, B), block 64: one wave per
// workgroup puts two or more workgroups on every CU at the chunk sizes of apply_model (T = 1895: 480 workgroups).
// ---------------------------------------------------------------------------
template <int DH>
__global__ __launch_bounds__(64) void hd_local_attn_kernel(const float *__restrict__ qkvd, int ld, int T, int H,
                                                            float *__restrict__ out) {
  __shared__ __attribute__((aligned(16))) float Ks[32][DH];
  __shared__ __attribute__((aligned(16))) float Cs[32][DH];
  const int head = blockIdx.y, b = blockIdx.z;
  const int s = blockIdx.x * 64 + threadIdx.x;
  const bool live = s < T;
  const float *base = qkvd + (int64_t)b * T * ld;
  float q[DH], acc[DH];
  float D = 0.f;
  const float scale = 1.0f / sqrtf((float)DH);
  if (live) {
    const float *qr = base + (int64_t)s * ld + head * DH;
#pragma unroll
    for (int c = 0; c < DH; ++c) q[c] = qr[c] * scale;
    const float *dr = base + (int64_t)s * ld + 3 * H + head * 4;
#pragma unroll
    for (int f = 0; f < 4; ++f) D += (float)(f + 1) * (1.0f / (1.0f + expf(-dr[f])));
    D *= 0.25f;
  } else {
#pragma unroll
    for (int c = 0; c < DH; ++c) q[c] = 0.f;
  }
#pragma unroll
  for (int c = 0; c < DH; ++c) acc[c] = 0.f;
  float m = -3.0e38f, l = 0.f;
  for (int t0 = 0; t0 < T; t0 += 32) {
    __syncthreads();
    for (int i = threadIdx.x; i < 32 * DH; i += 64) {
      const int tt = i / DH, c = i - tt * DH;
      const int t = t0 + tt;
      float kv = 0.f, cv = 0.f;
      if (t < T) {
        const float *row = base + (int64_t)t * ld + head * DH + c;
        kv = row[H];
        cv = row[2 * H];
      }
      Ks[tt][c] = kv;
      Cs[tt][c] = cv;
    }
    __syncthreads();
    const int tn = T - t0 < 32 ? T - t0 : 32;
    for (int tt = 0; tt < tn; ++tt) {
      const int t = t0 + tt;
      float sc = 0.f;
#pragma unroll
      for (int c = 0; c < DH; c += 4) {
        const float4 kv = *reinterpret_cast<const float4 *>(&Ks[tt][c]);
        sc += (q[c] * kv.x + q[c + 1] * kv.y) + (q[c + 2] * kv.z + q[c + 3] * kv.w);
      }
      const int d = t > s ? t - s : s - t;
      sc -= (float)d * D;
      if (t == s) sc = -100.0f;
      if (sc > m) {
        const float r = expf(m - sc);
        l *= r;
#pragma unroll
        for (int c = 0; c < DH; ++c) acc[c] *= r;
        m = sc;
      }
      const float p = expf(sc - m);
      l += p;
#pragma unroll
      for (int c = 0; c < DH; c += 4) {
        const float4 cv = *reinterpret_cast<const float4 *>(&Cs[tt][c]);
        acc[c] += p * cv.x;
        acc[c + 1] += p * cv.y;
        acc[c + 2] += p * cv.z;
        acc[c + 3] += p * cv.w;
      }
    }
  }
  if (live) {
    const float inv = 1.0f / l;
    float *o = out + ((int64_t)b * T + s) * H + head * DH;
#pragma unroll
    for (int c = 0; c < DH; ++c) o[c] = acc[c] * inv;
  }
}

}  // namespace asx
