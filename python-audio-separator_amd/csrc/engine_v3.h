// TFC-TDF v3 (MDX23C) on the engine: uvr_lib_v5/tfc_tdf_v3.py:110-267 and the TFC branch of
// MDXCSeparator.demix (architectures/mdxc_separator.py:345-404).  Included by asx.hip (one TU).
//
// Pre-activation blocks: every conv / linear is preceded by norm -> act.  InstanceNorm needs
// the statistics of the whole (T, F) plane, so norm -> act is its own memory-bound pass
// (instnorm_stats_kernel + norm_act_kernel) that writes the activated tensor the MFMA kernels
// then DMA; shortcut adds, the TDF residual, the decoder concat (written in place through
// channel-slice views) and the final `x * first_conv_out` are fused or zero-copy.
#pragma once

struct V3Norm {
  DevBuf g, b;
  int c = 0;
};

struct V3Block {
  int in_c = 0, c = 0, f = 0;
  V3Norm n_tfc1, n_tdf0, n_tdf1, n_tfc2;
  ConvLayer tfc1, tfc2, shortcut;
  TdfLayer tdf0, tdf1;
};

struct V3Net {
  asx_v3_config cfg{};
  bool begun = false, ready = false;
  ConvLayer first, final0, final1;
  std::vector<std::vector<V3Block>> enc, dec;
  std::vector<V3Block> mid;
  std::vector<ConvLayer> ds, us;
  std::vector<V3Norm> ds_n, us_n;
  // workspace (sized for ws_batch chunks)
  int ws_batch = 0;
  DevBuf cat0, firstout, S, A, X1, X2, XB, CUR, H, HA, stats, out_spec, frames, chunk_out, d_starts;
  std::vector<DevBuf> lvl;
};

static void v3_free_norm(V3Norm &n) {
  n.g.release();
  n.b.release();
}
static void v3_free_block(V3Block &b) {
  v3_free_norm(b.n_tfc1);
  v3_free_norm(b.n_tdf0);
  v3_free_norm(b.n_tdf1);
  v3_free_norm(b.n_tfc2);
  free_conv(b.tfc1);
  free_conv(b.tfc2);
  free_conv(b.shortcut);
  free_tdf(b.tdf0);
  free_tdf(b.tdf1);
}
static void v3_free(V3Net &n) {
  free_conv(n.first);
  free_conv(n.final0);
  free_conv(n.final1);
  for (auto &sc : n.enc)
    for (auto &b : sc) v3_free_block(b);
  for (auto &sc : n.dec)
    for (auto &b : sc) v3_free_block(b);
  for (auto &b : n.mid) v3_free_block(b);
  for (auto &c : n.ds) free_conv(c);
  for (auto &c : n.us) free_conv(c);
  for (auto &x : n.ds_n) v3_free_norm(x);
  for (auto &x : n.us_n) v3_free_norm(x);
  DevBuf *bufs[] = {&n.cat0, &n.firstout, &n.S, &n.A, &n.X1, &n.X2, &n.XB, &n.CUR, &n.H, &n.HA, &n.stats,
                    &n.out_spec, &n.frames, &n.chunk_out, &n.d_starts};
  for (auto *b : bufs) b->release();
  for (auto &l : n.lvl) l.release();
  n.ready = false;
}

static void v3_destroy(V3Net *n) {
  v3_free(*n);
  delete n;
}

// ---------------------------------------------------------------------------
// weights (reference state_dict names, tfc_tdf_v3.py)
// ---------------------------------------------------------------------------
static int v3_load_norm(asx_engine *e, V3Norm &n, const std::string &prefix, int c) {
  n.c = c;
  if (e->v3->cfg.norm == 0) return ASX_OK;
  const float *g, *b;
  CHK(get_tensor(e, prefix + ".weight", c, &g));
  CHK(get_tensor(e, prefix + ".bias", c, &b));
  CHK(n.g.ensure((size_t)c * 4));
  CHK(n.b.ensure((size_t)c * 4));
  HIPCHK(hipMemcpy(n.g.p, g, (size_t)c * 4, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(n.b.p, b, (size_t)c * 4, hipMemcpyHostToDevice));
  return ASX_OK;
}

static int v3_load_conv(asx_engine *e, ConvLayer &L, int kind, const std::string &name, int cin, int cout, int ntap) {
  const float *w;
  CHK(get_tensor(e, name, (int64_t)cin * cout * ntap, &w));
  CHK(conv_setup(L, kind, cin, cout, 0));
  CHK(conv_pack(L, w, nullptr, e->winograd));
  return ASX_OK;
}

static int v3_load_tfc_tdf(asx_engine *e, std::vector<V3Block> &blocks, const std::string &prefix, int in_c, int c,
                           int f) {
  const asx_v3_config &cf = e->v3->cfg;
  blocks.assign(cf.num_blocks_per_scale, V3Block());
  const int fb = f / cf.bottleneck_factor;
  for (int j = 0; j < cf.num_blocks_per_scale; ++j) {
    V3Block &b = blocks[j];
    const std::string p = prefix + ".blocks." + std::to_string(j);
    b.in_c = in_c;
    b.c = c;
    b.f = f;
    CHK(v3_load_norm(e, b.n_tfc1, p + ".tfc1.0", in_c));
    CHK(v3_load_conv(e, b.tfc1, CK_3X3, p + ".tfc1.2.weight", in_c, c, 9));
    CHK(v3_load_norm(e, b.n_tdf0, p + ".tdf.0", c));
    const float *w;
    CHK(get_tensor(e, p + ".tdf.2.weight", (int64_t)fb * f, &w));
    CHK(tdf_pack(b.tdf0, fb, f, c, w, nullptr, nullptr, nullptr));
    CHK(v3_load_norm(e, b.n_tdf1, p + ".tdf.3", c));
    CHK(get_tensor(e, p + ".tdf.5.weight", (int64_t)f * fb, &w));
    CHK(tdf_pack(b.tdf1, f, fb, c, w, nullptr, nullptr, nullptr));
    CHK(v3_load_norm(e, b.n_tfc2, p + ".tfc2.0", c));
    CHK(v3_load_conv(e, b.tfc2, CK_3X3, p + ".tfc2.2.weight", c, c, 9));
    CHK(v3_load_conv(e, b.shortcut, CK_1X1, p + ".shortcut.weight", in_c, c, 1));
    in_c = c;
  }
  return ASX_OK;
}

// ---------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------
struct V3View {
  float *p;
  int64_t bstride;  // floats between batch items
};

static int v3_norm_act(asx_engine *e, const V3Norm &n, V3View x, int C, int64_t P, int B, float *y, hipStream_t s) {
  V3Net &net = *e->v3;
  const bool has_norm = net.cfg.norm != 0;
  float2 *st = reinterpret_cast<float2 *>(net.stats.p);
  const int act = net.cfg.act == 1 ? 2 : 1;  // cfg.act: 0 relu, 1 gelu -> kernel enum 1 relu, 2 gelu
  const double bytes_r = 4.0 * B * C * (double)P;
  if (has_norm) {
    CHK(timed(e, ASX_PROF_MISC, 0.0, bytes_r, s, [&]() {
      hipLaunchKernelGGL(instnorm_stats_kernel, dim3(C, B), dim3(256), 0, s, x.p, x.bstride, C, P, 1e-5f, st);
    }));
  }
  const unsigned gx = (unsigned)std::min<int64_t>((P / 4 + 255) / 256 + 1, 64);
  const float *gp = n.g.f(), *bp = n.b.f();
  CHK(timed(e, ASX_PROF_MISC, 0.0, 2.0 * bytes_r, s, [&]() {
    hipLaunchKernelGGL(norm_act_kernel, dim3(gx, C, B), dim3(256), 0, s, x.p, x.bstride, C, P,
                       has_norm ? st : nullptr, gp, bp, act, y);
  }));
  return ASX_OK;
}

// One TFC_TDF (tfc_tdf_v3.py:110-148).  x_in: view with blocks[0].in_c channels; dest: view with c channels.
static int v3_tfc_tdf(asx_engine *e, const std::vector<V3Block> &blocks, V3View x_in, V3View dest, int B, int T, int F,
                      hipStream_t s) {
  V3Net &n = *e->v3;
  const int64_t P = (int64_t)T * F;
  V3View x = x_in;
  for (size_t j = 0; j < blocks.size(); ++j) {
    const V3Block &b = blocks[j];
    const bool last = j + 1 == blocks.size();
    ConvView vs;
    vs.x_bstride = x.bstride;
    CHK(conv_launch(e, b.shortcut, x.p, nullptr, n.S.f(), B, T, F, s, vs));                    // s = shortcut(x)
    CHK(v3_norm_act(e, b.n_tfc1, x, b.in_c, P, B, n.A.f(), s));
    CHK(conv_launch(e, b.tfc1, n.A.f(), nullptr, n.X1.f(), B, T, F, s));                        // x = tfc1(x)
    CHK(v3_norm_act(e, b.n_tdf0, V3View{n.X1.f(), (int64_t)b.c * P}, b.c, P, B, n.A.f(), s));
    const int64_t M = (int64_t)B * b.c * T;
    CHK(tdf_launch(e, b.tdf0, n.A.f(), nullptr, n.H.f(), M, T, s, 0));
    const int fb = b.tdf0.n;
    CHK(v3_norm_act(e, b.n_tdf1, V3View{n.H.f(), (int64_t)b.c * T * fb}, b.c, (int64_t)T * fb, B, n.HA.f(), s));
    CHK(tdf_launch(e, b.tdf1, n.HA.f(), n.X1.f(), n.X2.f(), M, T, s, 0));                       // x = x + tdf(x)
    CHK(v3_norm_act(e, b.n_tfc2, V3View{n.X2.f(), (int64_t)b.c * P}, b.c, P, B, n.A.f(), s));
    ConvView vo;
    vo.res = n.S.f();                                                                          // x = tfc2(x) + s
    V3View out = last ? dest : V3View{n.XB.f(), (int64_t)b.c * P};
    vo.y_bstride = out.bstride;
    CHK(conv_launch(e, b.tfc2, n.A.f(), nullptr, out.p, B, T, F, s, vo));
    x = out;
  }
  return ASX_OK;
}

static int v3_ensure_workspace(asx_engine *e, int B) {
  V3Net &n = *e->v3;
  if (B <= n.ws_batch) return ASX_OK;
  const asx_v3_config &cf = n.cfg;
  const int T = e->cfg.segment_size, k = cf.num_subbands, Fs = e->cfg.dim_f / k;
  const int dim_c = k * cf.num_channels * 2;
  const int64_t P0 = (int64_t)T * Fs;
  const int c0 = cf.num_channels_model;
  const size_t big = (size_t)B * 2 * c0 * P0 * 4;  // the level-0 decoder sees 2*c0 channels
  CHK(n.cat0.ensure((size_t)B * (dim_c + c0) * P0 * 4));
  CHK(n.firstout.ensure((size_t)B * c0 * P0 * 4));
  CHK(n.A.ensure(big));
  CHK(n.S.ensure(big / 2));
  CHK(n.X1.ensure(big / 2));
  CHK(n.X2.ensure(big / 2));
  CHK(n.XB.ensure(big / 2));
  CHK(n.CUR.ensure(big / 2));
  CHK(n.H.ensure(big / 2 / cf.bottleneck_factor + 256));
  CHK(n.HA.ensure(big / 2 / cf.bottleneck_factor + 256));
  int cmax = c0 + cf.growth * cf.num_scales;
  CHK(n.stats.ensure((size_t)B * 2 * cmax * sizeof(float2)));
  n.lvl.resize(cf.num_scales);
  int c = c0;
  int64_t P = P0;
  for (int i = 0; i < cf.num_scales; ++i) {
    CHK(n.lvl[i].ensure((size_t)B * 2 * c * P * 4));
    c += cf.growth;
    P /= 4;
  }
  CHK(n.out_spec.ensure((size_t)B * cf.num_targets * dim_c * P0 * 4));
  CHK(n.frames.ensure((size_t)B * cf.num_targets * 2 * T * e->cfg.n_fft * 4));
  n.ws_batch = B;
  return ASX_OK;
}

// spec (cws layout) is already in cat0[:, 0:dim_c]; result spec goes to out_spec [B, S*dim_c, T, Fs]
static int v3_core_dev(asx_engine *e, int B, hipStream_t s) {
  V3Net &n = *e->v3;
  const asx_v3_config &cf = n.cfg;
  const int T = e->cfg.segment_size, k = cf.num_subbands, Fs = e->cfg.dim_f / k;
  const int dim_c = k * cf.num_channels * 2;
  const int c0 = cf.num_channels_model;
  const int64_t P0 = (int64_t)T * Fs;
  const int64_t cat_bs = (int64_t)(dim_c + c0) * P0;
  {
    ConvView v;
    v.x_bstride = cat_bs;
    CHK(conv_launch(e, n.first, n.cat0.f(), nullptr, n.firstout.f(), B, T, Fs, s, v));
  }
  V3View x{n.firstout.f(), (int64_t)c0 * P0};
  int c = c0, t = T, f = Fs;
  for (int i = 0; i < cf.num_scales; ++i) {
    const int64_t P = (int64_t)t * f;
    V3View skip{n.lvl[i].f() + (int64_t)c * P, (int64_t)2 * c * P};   // second half of the decoder concat buffer
    CHK(v3_tfc_tdf(e, n.enc[i], x, skip, B, t, f, s));
    CHK(v3_norm_act(e, n.ds_n[i], skip, c, P, B, n.A.f(), s));
    CHK(conv_launch(e, n.ds[i], n.A.f(), nullptr, n.CUR.f(), B, t, f, s));
    c += cf.growth;
    t /= 2;
    f /= 2;
    x = V3View{n.CUR.f(), (int64_t)c * t * f};
  }
  CHK(v3_tfc_tdf(e, n.mid, x, V3View{n.CUR.f(), (int64_t)c * t * f}, B, t, f, s));
  for (int i = 0; i < cf.num_scales; ++i) {
    const int lv = cf.num_scales - 1 - i;
    const int64_t Pin = (int64_t)t * f;
    CHK(v3_norm_act(e, n.us_n[i], V3View{n.CUR.f(), (int64_t)c * Pin}, c, Pin, B, n.A.f(), s));
    const int co = c - cf.growth;
    const int64_t Pout = Pin * 4;
    ConvView v;
    v.y_bstride = (int64_t)2 * co * Pout;            // first half of the concat buffer (torch.cat([x, skip], 1))
    v.act = ACT_NONE;
    CHK(conv_launch(e, n.us[i], n.A.f(), nullptr, n.lvl[lv].f(), B, t, f, s, v));
    c = co;
    t *= 2;
    f *= 2;
    CHK(v3_tfc_tdf(e, n.dec[i], V3View{n.lvl[lv].f(), (int64_t)2 * c * Pout}, V3View{n.CUR.f(), (int64_t)c * Pout}, B, t,
                   f, s));
  }
  // x * first_conv_out -> cat0[:, dim_c:]   (tfc_tdf_v3.py:257-259)
  {
    const int64_t CP = (int64_t)c0 * P0;
    const unsigned gx = (unsigned)std::min<int64_t>((CP + 255) / 256, 4096);
    float *ydst = n.cat0.f() + (int64_t)dim_c * P0;
    CHK(timed(e, ASX_PROF_MISC, 0.0, 12.0 * B * CP, s, [&]() {
      hipLaunchKernelGGL(mul_into_view_kernel, dim3(gx, B), dim3(256), 0, s, n.CUR.f(), n.firstout.f(), CP, ydst,
                         cat_bs);
    }));
  }
  ConvView vf;
  vf.act = cf.act == 1 ? ACT_GELU : ACT_RELU;
  CHK(conv_launch(e, n.final0, n.cat0.f(), nullptr, n.X1.f(), B, T, Fs, s, vf));
  CHK(conv_launch(e, n.final1, n.X1.f(), nullptr, n.out_spec.f(), B, T, Fs, s));
  return ASX_OK;
}

// chunk waves -> separated chunk waves [B, S, 2, C]; input either explicit chunks [B,2,C] (n_song < 0)
// or windows of the resident mix (song mode, front zeros = `front`)
static int v3_chunks_dev(asx_engine *e, const float *wave, const int64_t *d_starts, int64_t n_song, int front, int B,
                         float *out, hipStream_t s) {
  V3Net &n = *e->v3;
  const asx_v3_config &cf = n.cfg;
  const int T = e->cfg.segment_size, k = cf.num_subbands, Fs = e->cfg.dim_f / k;
  const int dim_c = k * cf.num_channels * 2;
  const int64_t C = (int64_t)e->cfg.hop_length * (T - 1);
  const int64_t P0 = (int64_t)T * Fs;
  CHK(v3_ensure_workspace(e, B));
  {
    StftArgs a{};
    a.wave = wave;
    a.chunk_start = d_starts;
    a.n_song = n_song;
    a.trim = front;
    a.C = C;
    a.hop = e->cfg.hop_length;
    a.T = T;
    a.dim_f = e->cfg.dim_f;
    a.zero_low = 0;
    a.tf_layout = 1;
    a.spec = n.cat0.f();
    a.window = e->d_window.f();
    a.tw = reinterpret_cast<const float2 *>(e->d_tw.p);
    a.sign = 1.0f;
    a.subbands = k;
    a.out_bstride = (int64_t)(dim_c + cf.num_channels_model) * P0;
    FftPlan p = e->plan;
    CHK(timed(e, ASX_PROF_STFT, 0.0, 4.0 * ((double)B * 2 * C + (double)B * 4 * T * e->cfg.dim_f), s, [&]() {
      hipLaunchKernelGGL(stft_kernel, dim3(T, 2, B), dim3(256), stft_lds(p), s, a, p);
    }));
  }
  CHK(v3_core_dev(e, B, s));
  const int S = cf.num_targets;
  {
    IstftArgs a{};
    a.spec = n.out_spec.f();
    a.T = T;
    a.dim_f = e->cfg.dim_f;
    a.tf_layout = 1;
    a.combine = 0;
    a.frames = n.frames.f();
    a.window = e->d_window.f();
    a.tw = reinterpret_cast<const float2 *>(e->d_tw.p);
    a.subbands = k;
    a.n_inst = S;
    a.in_bstride = (int64_t)S * dim_c * P0;
    FftPlan p = e->plan;
    CHK(timed(e, ASX_PROF_ISTFT, 0.0, 4.0 * ((double)B * S * 4 * T * e->cfg.dim_f + (double)B * S * 2 * T * e->cfg.n_fft),
              s, [&]() { hipLaunchKernelGGL(istft_kernel, dim3(T, 2, B * S), dim3(256), istft_lds(p), s, a, p); }));
  }
  CHK(ola_launch(e, n.frames.f(), e->d_env.f(), nullptr, B * S, T, C, out, s));
  return ASX_OK;
}

static double v3_flops(const asx_engine *e, int batch) {
  if (!e->v3 || !e->v3->begun) return 0.0;
  const asx_v3_config &cf = e->v3->cfg;
  const double T = e->cfg.segment_size, k = cf.num_subbands, Fs = e->cfg.dim_f / k;
  const double dim_c = k * cf.num_channels * 2;
  double c = cf.num_channels_model, t = T, f = Fs, fl = 2.0 * dim_c * c * T * Fs;
  auto tfc_tdf = [&](double in_c, double cc, double tt, double ff) {
    double r = 0;
    for (int j = 0; j < cf.num_blocks_per_scale; ++j) {
      r += 2.0 * in_c * cc * tt * ff;                                  // shortcut
      r += 2.0 * 9.0 * in_c * cc * tt * ff + 2.0 * 9.0 * cc * cc * tt * ff;  // tfc1, tfc2
      r += 2.0 * 2.0 * cc * tt * ff * (ff / cf.bottleneck_factor);     // two linears
      in_c = cc;
    }
    return r;
  };
  for (int i = 0; i < cf.num_scales; ++i) {
    fl += tfc_tdf(c, c, t, f);
    fl += 2.0 * 4.0 * c * (c + cf.growth) * (t / 2) * (f / 2);
    c += cf.growth;
    t /= 2;
    f /= 2;
  }
  fl += tfc_tdf(c, c, t, f);
  for (int i = 0; i < cf.num_scales; ++i) {
    fl += 2.0 * 4.0 * c * (c - cf.growth) * t * f;
    c -= cf.growth;
    t *= 2;
    f *= 2;
    fl += tfc_tdf(2 * c, c, t, f);
  }
  fl += 2.0 * (c + dim_c) * c * T * Fs + 2.0 * c * cf.num_targets * dim_c * T * Fs;
  return fl * batch;
}
