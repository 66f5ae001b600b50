// Spectral edges: Ensembler.ensemble and spec_utils.invert_stem on the engine (kernels_ens.h).  Included by asx.hip.
#pragma once

struct EnsCtx {
  FftPlan plan{};
  DevBuf window, tw, frames, wss, dweights, din, din2, dout, partial, sel;
  bool ready = false;
};

static void ens_destroy(EnsCtx *c) {
  for (DevBuf *b : {&c->window, &c->tw, &c->frames, &c->wss, &c->dweights, &c->din, &c->din2, &c->dout, &c->partial, &c->sel}) b->release();
  delete c;
}

static int ens_ctx(asx_engine *e) {
  if (!e->ens) e->ens = new EnsCtx();
  EnsCtx &c = *e->ens;
  if (c.ready) return ASX_OK;
  const int n_fft = 2048;
  REQUIRE(make_plan(n_fft, &c.plan), "n_fft 2048 plan");
  std::vector<float> w;
  host_window(n_fft, w);
  CHK(ht_up(c.window, w));
  std::vector<float> tw((size_t)n_fft * 2);
  for (int j = 0; j < n_fft; ++j) {
    const double ang = -2.0 * M_PI * (double)j / (double)n_fft;
    tw[2 * j] = (float)cos(ang);
    tw[2 * j + 1] = (float)sin(ang);
  }
  CHK(ht_up(c.tw, tw));
  const int lds = (int)(((size_t)c.plan.nh * 2 + (size_t)(c.plan.nh + 1) * (2 + ENS_MAX_K)) * sizeof(float2));
  (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&ens_fft_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&ens_invert_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  c.ready = true;
  return ASX_OK;
}

// librosa.istft fold of frames [2, T, 2048] -> out [2, len] (len = N for length=N, hop*(T-1) otherwise)
static int ens_fold(asx_engine *e, int T, int64_t len, float *out, hipStream_t s) {
  EnsCtx &c = *e->ens;
  const int nf = c.plan.n_fft, hop = 1024;
  std::vector<float> w;
  host_window(nf, w);
  const size_t cover = (size_t)nf + (size_t)hop * (T - 1);
  std::vector<double> ss(std::max(cover, (size_t)len + nf), 0.0);   // zero beyond the frames: those samples stay 0
  for (int t = 0; t < T; ++t)
    for (int k = 0; k < nf; ++k) ss[(size_t)t * hop + k] += (double)w[k] * (double)w[k];
  std::vector<float> ssf(ss.begin(), ss.end());
  CHK(c.wss.ensure(ssf.size() * 4));
  HIPCHK(hipMemcpyAsync(c.wss.p, ssf.data(), ssf.size() * 4, hipMemcpyHostToDevice, s));
  HIPCHK(hipStreamSynchronize(s));
  return timed(e, ASX_PROF_OLA, 0.0, 4.0 * 2 * (T * (double)nf + 2.0 * len), s, [&]() {
    hipLaunchKernelGGL(vr_ola_kernel, dim3((unsigned)((len + 255) / 256)), dim3(256), 0, s, c.frames.f(), c.wss.f(), nf, hop, T, len, 0,
                       (const float *)nullptr, out);
  });
}

static int ens_ensemble_dev(asx_engine *e, const float *waves, int K, int64_t N, int alg, const double *weights_host, float *out,
                            int64_t *n_out, hipStream_t s) {
  CHK(ens_ctx(e));
  EnsCtx &c = *e->ens;
  REQUIRE(K >= 2 && K <= ENS_MAX_K, "ensemble of %d inputs (2 .. %d are built)", K, ENS_MAX_K);
  REQUIRE(alg >= ENS_AVG_WAVE && alg <= ENS_ENSEMBLE_WAV, "unknown ensemble algorithm %d", alg);
  if (alg == ENS_ENSEMBLE_WAV) {
    CHK(c.partial.ensure((size_t)K * 2 * ENS_ABS_BLOCKS * 8));
    CHK(c.sel.ensure(16));
    *n_out = N;
    return timed(e, ASX_PROF_MISC, 0.0, 4.0 * (K + 2) * 2 * N, s, [&]() {
      hipLaunchKernelGGL(ens_abssum_kernel, dim3(ENS_ABS_BLOCKS, K * 2), dim3(256), 0, s, waves, N, reinterpret_cast<double *>(c.partial.p));
      hipLaunchKernelGGL(ens_pick_kernel, dim3(1), dim3(64), 0, s, reinterpret_cast<const double *>(c.partial.p), K, ENS_ABS_BLOCKS,
                         reinterpret_cast<int *>(c.sel.p));
      hipLaunchKernelGGL(ens_take_kernel, dim3((unsigned)((N + 255) / 256), 2), dim3(256), 0, s, waves, N,
                         reinterpret_cast<const int *>(c.sel.p), out);
    });
  }
  std::vector<double> w(K, 1.0);
  if (weights_host) w.assign(weights_host, weights_host + K);
  double wsum = 0.0;
  for (double v : w) wsum += v;
  CHK(c.dweights.ensure((size_t)K * 8));
  HIPCHK(hipMemcpyAsync(c.dweights.p, w.data(), (size_t)K * 8, hipMemcpyHostToDevice, s));
  HIPCHK(hipStreamSynchronize(s));
  const double *dw = reinterpret_cast<const double *>(c.dweights.p);
  if (alg <= ENS_MAX_WAVE) {
    const int64_t n2 = 2 * N;
    *n_out = N;
    return timed(e, ASX_PROF_MISC, 0.0, 4.0 * (K + 1) * n2, s, [&]() {
      hipLaunchKernelGGL(ens_wave_kernel, dim3((unsigned)((n2 + 255) / 256)), dim3(256), 0, s, waves, K, n2, alg, dw, wsum, out);
    });
  }
  const int hop = 1024, nh = c.plan.nh;
  const int T = (int)(1 + N / hop);
  CHK(c.frames.ensure((size_t)2 * T * c.plan.n_fft * 4));
  const size_t lds = ((size_t)nh * 2 + (size_t)(nh + 1) * (2 + (alg == ENS_MEDIAN_FFT ? K : 0))) * sizeof(float2);
  CHK(timed(e, ASX_PROF_STFT, 0.0, 4.0 * ((double)K * 2 * N + 2.0 * T * c.plan.n_fft), s, [&]() {
    hipLaunchKernelGGL(ens_fft_kernel, dim3(T, 2), dim3(256), lds, s, waves, K, N, alg, dw, wsum, hop, c.frames.f(), c.window.f(),
                       reinterpret_cast<const float2 *>(c.tw.p), c.plan);
  }));
  *n_out = alg >= ENS_UVR_MAX_SPEC ? (int64_t)hop * (T - 1) : N;   // spectrogram_to_wave_no_mp has no length argument
  return ens_fold(e, T, *n_out, out, s);
}

static int ens_invert_dev(asx_engine *e, const float *mix, const float *stem, int64_t N, float *out, int64_t *n_out, hipStream_t s) {
  CHK(ens_ctx(e));
  EnsCtx &c = *e->ens;
  const int hop = 1024, nh = c.plan.nh;
  const int T = (int)(1 + N / hop);
  REQUIRE(T >= 2, "input too short");
  CHK(c.frames.ensure((size_t)2 * T * c.plan.n_fft * 4));
  const size_t lds = ((size_t)nh * 2 + (size_t)(nh + 1) * 2) * sizeof(float2);
  CHK(timed(e, ASX_PROF_STFT, 0.0, 4.0 * (4.0 * N + 2.0 * T * c.plan.n_fft), s, [&]() {
    hipLaunchKernelGGL(ens_invert_kernel, dim3(T, 2), dim3(256), lds, s, mix, stem, N, hop, c.frames.f(), c.window.f(),
                       reinterpret_cast<const float2 *>(c.tw.p), c.plan);
  }));
  *n_out = (int64_t)hop * (T - 1);
  return ens_fold(e, T, *n_out, out, s);
}
