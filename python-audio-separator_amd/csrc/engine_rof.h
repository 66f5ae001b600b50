// BS-Roformer on the engine: uvr_lib_v5/roformer/bs_roformer.py:300-522 and the Roformer branch of
// MDXCSeparator.demix (architectures/mdxc_separator.py:272-343).  Included by asx.hip (one TU).
#pragma once

struct RofLin {
  int n = 0, k = 0;
  DevBuf w, b;   // w [n, k] (nn.Linear layout), optional bias [n]
  bool has_bias = false;
};

struct RofAttn {
  bool norm_folded = false;   // gamma of `norm` is folded into qkv / gates (rows are scaled in their epilogues)
  DevBuf norm_g;
  RofLin qkv, gates, out;
  DevBuf rot_tab;   // [n_pos, dh/2] (cos, sin)
};

struct RofFF {
  bool norm_folded = false;   // gamma of the feed-forward RMSNorm is folded into l1
  DevBuf norm_g;
  RofLin l1, l2;
};

struct RofLayer {
  RofAttn attn;
  RofFF ff;
};

struct RofNet {
  asx_rof_config cfg{};
  bool begun = false, ready = false;
  std::vector<int> band_dim, band_off;   // per band: 2 * f * channels, offset inside the (f s c) spectrum vector
  std::vector<int> mask_off;             // per band: offset inside the concatenated band masks
  int W = 0;                              // 2 * 2 * n_bins: width of the spectrum vector
  int MW = 0;                             // sum of band dims (> W for overlapping mel bands)
  std::vector<DevBuf> tnorm_t, tnorm_f;   // mel: output RMSNorm of every time / frequency Transformer
  DevBuf d_bstart, d_moff, d_jlo, d_jhi, MASKB;
  std::vector<DevBuf> bs_gamma;
  std::vector<RofLin> bs_lin;
  std::vector<std::vector<RofLayer>> time_l, freq_l;   // [depth][transformer depth]
  DevBuf final_g;
  std::vector<std::vector<std::vector<RofLin>>> mask;  // [stem][band][mlp layer]
  // workspace
  int ws_batch = 0;
  DevBuf X0, XB, TOK, XN, RS, QKV, ATT, GATE, FFH, FFE, HID, GLU, MASK, frames, chunk_out, d_starts, d_window;
  DevBuf win_synth;   // stft_normalized: the synthesis window x sqrt(n_fft) (torch.istft multiplies its input by sqrt(n_fft) first)
};

static void rof_free_lin(RofLin &l) {
  l.w.release();
  l.b.release();
}
static void rof_free_layer(RofLayer &l) {
  l.attn.norm_g.release();
  rof_free_lin(l.attn.qkv);
  rof_free_lin(l.attn.gates);
  rof_free_lin(l.attn.out);
  l.attn.rot_tab.release();
  l.ff.norm_g.release();
  rof_free_lin(l.ff.l1);
  rof_free_lin(l.ff.l2);
}
static void rof_free(RofNet &n) {
  for (auto &g : n.bs_gamma) g.release();
  for (auto &l : n.bs_lin) rof_free_lin(l);
  for (auto &d : n.time_l)
    for (auto &l : d) rof_free_layer(l);
  for (auto &d : n.freq_l)
    for (auto &l : d) rof_free_layer(l);
  n.final_g.release();
  for (auto &g : n.tnorm_t) g.release();
  for (auto &g : n.tnorm_f) g.release();
  for (DevBuf *b : {&n.d_bstart, &n.d_moff, &n.d_jlo, &n.d_jhi, &n.MASKB}) b->release();
  for (auto &s : n.mask)
    for (auto &b : s)
      for (auto &l : b) rof_free_lin(l);
  DevBuf *bufs[] = {&n.X0, &n.XB, &n.TOK, &n.XN, &n.RS, &n.QKV, &n.ATT, &n.GATE, &n.FFH, &n.FFE, &n.HID, &n.GLU,
                    &n.MASK, &n.frames, &n.chunk_out, &n.d_starts, &n.d_window, &n.win_synth};
  for (auto *b : bufs) b->release();
  n.ready = false;
  n.ws_batch = 0;
}
static void rof_destroy(RofNet *n) {
  rof_free(*n);
  delete n;
}

static int rof_upload(asx_engine *e, DevBuf &d, const std::string &name, int64_t numel) {
  const float *p;
  CHK(get_tensor(e, name, numel, &p));
  CHK(d.ensure((size_t)numel * 4));
  HIPCHK(hipMemcpy(d.p, p, (size_t)numel * 4, hipMemcpyHostToDevice));
  return ASX_OK;
}

static int rof_load_lin(asx_engine *e, RofLin &l, const std::string &name, int n, int k, bool bias) {
  l.n = n;
  l.k = k;
  l.has_bias = bias;
  CHK(rof_upload(e, l.w, name + ".weight", (int64_t)n * k));
  if (bias) CHK(rof_upload(e, l.b, name + ".bias", n));
  return ASX_OK;
}

// nn.Linear weight [n, k] with the gamma [k] of the RMSNorm in front of it folded in: w'[i][j] = w[i][j] * gamma[j]
static int rof_load_lin_folded(asx_engine *e, RofLin &l, const std::string &name, int n, int k, bool bias,
                               const std::string &gamma_name) {
  l.n = n;
  l.k = k;
  l.has_bias = bias;
  const float *w, *g;
  CHK(get_tensor(e, name + ".weight", (int64_t)n * k, &w));
  CHK(get_tensor(e, gamma_name, k, &g));
  std::vector<float> f((size_t)n * k);
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < k; ++j) f[(size_t)i * k + j] = w[(size_t)i * k + j] * g[j];
  CHK(l.w.ensure(f.size() * 4));
  HIPCHK(hipMemcpy(l.w.p, f.data(), f.size() * 4, hipMemcpyHostToDevice));
  if (bias) CHK(rof_upload(e, l.b, name + ".bias", n));
  return ASX_OK;
}

// RMSNorms folded into the projections behind them (default; ASX_ROF_NORMFUSE=0 keeps the separate normalisation pass)
static bool rof_norm_fuse() {
  static const bool on = !(getenv("ASX_ROF_NORMFUSE") && atoi(getenv("ASX_ROF_NORMFUSE")) == 0);
  return on;
}

// cos/sin table exactly as torch builds it: angle = float32(pos) * float32(freq) (one float32
// rounding), then cos/sin of that float32 angle.
static int rof_rot_table(asx_engine *e, DevBuf &tab, const std::string &name, int n_pos, int half) {
  const float *fr;
  CHK(get_tensor(e, name, half, &fr));
  std::vector<float> t((size_t)n_pos * half * 2);
  for (int p = 0; p < n_pos; ++p)
    for (int i = 0; i < half; ++i) {
      const float ang = (float)p * fr[i];
      t[((size_t)p * half + i) * 2] = (float)cos((double)ang);
      t[((size_t)p * half + i) * 2 + 1] = (float)sin((double)ang);
    }
  CHK(tab.ensure(t.size() * 4));
  HIPCHK(hipMemcpy(tab.p, t.data(), t.size() * 4, hipMemcpyHostToDevice));
  return ASX_OK;
}

static int rof_load_layer(asx_engine *e, RofLayer &L, const std::string &p, int n_pos) {
  const asx_rof_config &c = e->rof->cfg;
  const int d = c.dim, inner = c.heads * c.dim_head;
  CHK(rof_upload(e, L.attn.norm_g, p + ".0.norm.gamma", d));
  // (the gates projection has a scalar-epilogue fallback for head counts that are not a multiple of 4: no folding there)
  L.attn.norm_folded = rof_norm_fuse() && c.heads % 4 == 0;
  if (L.attn.norm_folded) {
    CHK(rof_load_lin_folded(e, L.attn.qkv, p + ".0.to_qkv", 3 * inner, d, false, p + ".0.norm.gamma"));
    CHK(rof_load_lin_folded(e, L.attn.gates, p + ".0.to_gates", c.heads, d, true, p + ".0.norm.gamma"));
  } else {
    CHK(rof_load_lin(e, L.attn.qkv, p + ".0.to_qkv", 3 * inner, d, false));
    CHK(rof_load_lin(e, L.attn.gates, p + ".0.to_gates", c.heads, d, true));
  }
  CHK(rof_load_lin(e, L.attn.out, p + ".0.to_out.0", d, inner, false));
  CHK(rof_rot_table(e, L.attn.rot_tab, p + ".0.rotary_embed.freqs", n_pos, c.dim_head / 2));
  CHK(rof_upload(e, L.ff.norm_g, p + ".1.net.0.gamma", d));
  L.ff.norm_folded = rof_norm_fuse();
  if (L.ff.norm_folded) CHK(rof_load_lin_folded(e, L.ff.l1, p + ".1.net.1", 4 * d, d, true, p + ".1.net.0.gamma"));
  else CHK(rof_load_lin(e, L.ff.l1, p + ".1.net.1", 4 * d, d, true));
  CHK(rof_load_lin(e, L.ff.l2, p + ".1.net.4", d, 4 * d, true));
  return ASX_OK;
}

// y[M, n] (row stride ldy) = act(x[M, k] (row stride lda) @ w^T + b) (+ res)
// rotary epilogue of a qkv projection (kernels_net.h: TdfDmaArgs::rot_*); tab == nullptr: none
struct RofRot {
  const float2 *tab = nullptr;
  int cols = 0, half = 0, pos_mod = 1;
  int64_t pos_div = 1;
};

// activation code of the feed-forward GELU: 2 = libm erff, 6 = fast_erf (kernels_net.h); ASX_ROF_GELU overrides (1 = ReLU: a timing
// probe with wrong results)
static int rof_gelu_act() {
  static const int v = getenv("ASX_ROF_GELU") ? atoi(getenv("ASX_ROF_GELU")) : 2;
  return v;
}

static int rof_gemm_args(asx_engine *e, const RofLin &L, const float *x, int64_t lda, int64_t M, float *y, int64_t ldy,
                         int act, const float *res, int64_t ldr, const RofRot *rot, const float *rscale, TdfDmaArgs &d) {
  if ((L.k & 3) || (lda & 3) || (reinterpret_cast<uintptr_t>(x) & 15)) {
    set_err("rof_gemm: K and the row stride must be multiples of 4 floats (K=%d lda=%lld)", L.k, (long long)lda);
    return ASX_ERR_INVALID;
  }
  d = TdfDmaArgs{};
  d.x = x;
  d.w = L.w.f();
  d.bias = L.has_bias ? L.b.f() : nullptr;
  d.scale = nullptr;
  d.shift = nullptr;
  d.res = res;
  d.zeros = e->d_zeros.f();
  d.y = y;
  d.M = M;
  d.N = L.n;
  d.K = L.k;
  d.C = 1;
  d.T = 1;
  d.relu = act;
  d.lda = lda;
  d.rscale = rscale;
  if (rot && rot->tab) {
    d.rot_tab = rot->tab;
    d.rot_cols = rot->cols;
    d.rot_half = rot->half;
    d.rot_pos_mod = rot->pos_mod;
    d.rot_pos_div = rot->pos_div;
  }
  // the float4 epilogue needs 16-byte aligned rows; otherwise force the scalar epilogue by an odd N check inside
  d.ldy = ldy;
  d.ldr = ldr;
  const bool vec_ok = (ldy % 4 == 0) && (res == nullptr || ldr % 4 == 0) && (L.n % 4 == 0) &&
                      ((reinterpret_cast<uintptr_t>(y) & 15) == 0) &&
                      (res == nullptr || (reinterpret_cast<uintptr_t>(res) & 15) == 0);
  if (!vec_ok) {
    set_err("rof_gemm: output rows must be 16-byte aligned (N=%d ldy=%lld)", L.n, (long long)ldy);
    return ASX_ERR_INVALID;
  }
  return ASX_OK;
}

// pair_out / pair_in: y written / x read as a pair image (kernels_net.h TdfDmaArgs::yexp / xexp) with `pair_cols` columns per exponent span --
// only behind tdf3h_will_run() for the layer
static int rof_gemm(asx_engine *e, const RofLin &L, const float *x, int64_t lda, int64_t M, float *y, int64_t ldy,
                    int act, const float *res, int64_t ldr, hipStream_t s, const RofRot *rot = nullptr,
                    const float *rscale = nullptr, int *pair_out = nullptr, const int *pair_in = nullptr, int pair_cols = 0) {
  if (M <= 0) return ASX_OK;
  TdfDmaArgs d;
  CHK(rof_gemm_args(e, L, x, lda, M, y, ldy, act, res, ldr, rot, rscale, d));
  if (pair_out) {
    d.yexp = pair_out;
    d.yexp_n = (L.n + pair_cols - 1) / pair_cols;
  }
  if (pair_in) tdf3_set_xexp(d, pair_in, pair_cols);
  const double flops = 2.0 * (double)M * L.n * L.k;
  const double bytes = 4.0 * ((double)M * L.k + (double)M * L.n * (res ? 2 : 1) + (double)L.n * L.k);
  return timed(e, ASX_PROF_TDF, flops, bytes, s, [&]() {
    launch_tdf_dma_auto(e, d, s);
  });
}

static int rof_rmsnorm(asx_engine *e, const float *x, int64_t lda, int d, const float *g, float *y, int64_t ldy,
                       int64_t M, hipStream_t s) {
  return timed(e, ASX_PROF_MISC, 0.0, 8.0 * M * d, s, [&]() {
    hipLaunchKernelGGL(rmsnorm_kernel, dim3((unsigned)((M + 3) / 4)), dim3(256), 0, s, x, lda, d, g, y, ldy, M);
  });
}

static int rof_rownorm(asx_engine *e, const float *x, int64_t lda, int d, float *r, int64_t M, hipStream_t s) {
  return timed(e, ASX_PROF_MISC, 0.0, 4.0 * M * d, s, [&]() {
    hipLaunchKernelGGL(rownorm_kernel, dim3((unsigned)((M + 3) / 4)), dim3(256), 0, s, x, lda, d, r, M);
  });
}

// one Transformer (bs_roformer.py:136-160, norm_output = False) over the token matrix TOK [M, D]
static int rof_transformer(asx_engine *e, std::vector<RofLayer> &layers, bool time_axis, int B, hipStream_t s,
                           const DevBuf *out_norm = nullptr) {
  RofNet &n = *e->rof;
  const asx_rof_config &c = n.cfg;
  const int T = e->cfg.segment_size, Fb = c.n_bands, D = c.dim, H = c.heads, inner = H * c.dim_head;
  const int64_t M = (int64_t)B * T * Fb;
  // The row-wise launches run over M rounded up to a multiple of 8 (the workspace holds the pad rows): the second-generation
  // row GEMM needs M % 8 == 0, and M = B * T * bands is odd * 2 for the public layouts unless B is a multiple of 4.  Pad rows
  // hold garbage, feed only themselves (every one of these kernels is row-independent) and are never read by the attention,
  // the band split or the mask estimator.
  const int64_t Mg = (M + 7) & ~(int64_t)7;
  for (auto &L : layers) {
    // attention: x = attn(x) + x
    // RMSNorm in front of qkv / gates: folded (TOK is the operand, 1 / |x| scales the rows in the epilogue) or explicit
    const float *ain = n.TOK.f(), *ars = nullptr;
    if (L.attn.norm_folded) {
      CHK(rof_rownorm(e, n.TOK.f(), D, D, n.RS.f(), Mg, s));
      ars = n.RS.f();
    } else {
      CHK(rof_rmsnorm(e, n.TOK.f(), D, D, L.attn.norm_g.f(), n.XN.f(), D, Mg, s));
      ain = n.XN.f();
    }
    // rotary on q and k: in the projection's epilogue (default), or as a separate in-place pass (ASX_ROF_FUSE=0)
    static const bool fuse_rot = !(getenv("ASX_ROF_FUSE") && atoi(getenv("ASX_ROF_FUSE")) == 0);
    RofRot rr;
    if (fuse_rot && c.dim_head % 4 == 0 && M < (1ll << 31)) {
      rr.tab = reinterpret_cast<const float2 *>(L.attn.rot_tab.p);
      rr.cols = 2 * inner;
      rr.half = c.dim_head / 2;
      rr.pos_div = time_axis ? Fb : 1;
      rr.pos_mod = time_axis ? T : Fb;
    }
    CHK(rof_gemm(e, L.attn.qkv, ain, D, Mg, n.QKV.f(), 3 * inner, 0, nullptr, 0, s, &rr, ars));
    if (!rr.tab) {
      const int64_t tot = M * 2 * H * (c.dim_head / 2);
      const int64_t pos_div = time_axis ? Fb : 1;
      const int pos_mod = time_axis ? T : Fb;
      float2 *tab = reinterpret_cast<float2 *>(L.attn.rot_tab.p);
      CHK(timed(e, ASX_PROF_MISC, 0.0, 16.0 * tot, s, [&]() {
        hipLaunchKernelGGL(rotary_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s, n.QKV.f(),
                           (int64_t)3 * inner, M, H, c.dim_head, tab, pos_div, pos_mod);
      }));
    }
    {
      // gates: [M, heads]; rows are padded to a multiple of 4 floats
      const int gl = (H + 3) / 4 * 4;
      RofLin &G = L.attn.gates;
      if (H % 4 == 0) {
        CHK(rof_gemm(e, G, ain, D, Mg, n.GATE.f(), gl, 0, nullptr, 0, s, nullptr, ars));
      } else {
        // tiny head counts (tests): scalar epilogue through the generic register-staged kernel
        TdfArgs a{};
        a.x = n.XN.f();
        a.w = G.w.f();
        a.bias = G.b.f();
        a.scale = nullptr;
        a.shift = nullptr;
        a.res = nullptr;
        a.y = n.GATE.f();
        a.M = M;
        a.N = H;
        a.K = D;
        a.C = 1;
        a.T = 1;
        a.relu = 0;
        CHK(timed(e, ASX_PROF_TDF, 2.0 * M * H * D, 4.0 * M * (D + H), s, [&]() { launch_tdf_t<1, 4>(a, s); }));
      }
      AttnArgs aa{};
      aa.qkv = n.QKV.f();
      aa.gate = n.GATE.f();
      aa.out = n.ATT.f();
      aa.heads = H;
      aa.gate_ld = (H % 4 == 0) ? gl : H;
      aa.scale = 1.0f / sqrtf((float)c.dim_head);
      static const int attn_exact = getenv("ASX_ATTN_EXACT") != nullptr;
      aa.exact = attn_exact;
      int64_t nseq;
      if (time_axis) {
        aa.len = T;
        aa.row_stride = Fb;
        aa.inner_cnt = Fb;
        aa.outer_stride = (int64_t)T * Fb;
        aa.inner_stride = 1;
        nseq = (int64_t)B * Fb;
      } else {
        aa.len = Fb;
        aa.row_stride = 1;
        aa.inner_cnt = 1;
        aa.outer_stride = Fb;
        aa.inner_stride = 0;
        nseq = (int64_t)B * T;
      }
      const int qtiles = (aa.len + 63) / 64;
      const double fl = 4.0 * (double)nseq * H * (double)aa.len * aa.len * c.dim_head;
      CHK(timed(e, ASX_PROF_CONV1X1, fl, 4.0 * M * 4 * inner, s, [&]() {
#ifdef ASX_EXPERIMENTAL_KERNELS
        static const bool v1 = getenv("ASX_ATTN_V1") && atoi(getenv("ASX_ATTN_V1")) != 0;   // the 4-byte-fragment kernel (A/B)
#else
        constexpr bool v1 = false;
#endif
        static const bool attn_db = getenv("ASX_ATTN_DB") && atoi(getenv("ASX_ATTN_DB")) != 0;   // A/B (default off until measured)
        static const int qw = getenv("ASX_ATTN_QW") ? atoi(getenv("ASX_ATTN_QW")) : 1;   // 2: 128 queries per workgroup (measured slower: 357 vs 328 ms)
        // bf16 x 6 form (kernels_rof.h: attention6_kernel) under the process-wide switch of the row GEMM; ASX_ATTN6=0: A/B
        static const bool attn6 = !(getenv("ASX_ATTN6") && atoi(getenv("ASX_ATTN6")) == 0);
        if (attn6 && e->gemm_bf16x6 > 0 && !v1) {
          AttnArgs a2 = aa;
          static const int qw6 = getenv("ASX_ATTN6_QW") ? atoi(getenv("ASX_ATTN6_QW")) : 2;   // 128 queries per workgroup on long sequences
          const bool h3 = e->gemm_f16x3 > 0;           // fp16 x 3 arithmetic (kernels_rof.h: template parameter H)
          if (qw6 >= 2 && aa.len > 128) {
            a2.nqt = (aa.len + 127) / 128;
            const dim3 g2((unsigned)((int64_t)a2.nqt * H * nseq));
            if (h3) hipLaunchKernelGGL((attention6_kernel<2, true>), g2, dim3(256), 0, s, a2);
            else hipLaunchKernelGGL(attention6_kernel<2>, g2, dim3(256), 0, s, a2);
          } else {
            a2.nqt = qtiles;
            const dim3 g1((unsigned)((int64_t)qtiles * H * nseq));
            if (h3) hipLaunchKernelGGL((attention6_kernel<1, true>), g1, dim3(256), 0, s, a2);
            else hipLaunchKernelGGL(attention6_kernel<1>, g1, dim3(256), 0, s, a2);
          }
          g_attn6_launches.fetch_add(1);
          if (h3) g_attn6h_launches.fetch_add(1);
          e->prof_nprod = h3 ? 3 : 6;
        } else
#ifdef ASX_EXPERIMENTAL_KERNELS
        if (v1) hipLaunchKernelGGL(attention_kernel, dim3(qtiles, H, (unsigned)nseq), dim3(256), 0, s, aa);
        else
#endif
        if (qw >= 2 && aa.len > 64) {
          AttnArgs a2 = aa;
          a2.nqt = (aa.len + 127) / 128;
          hipLaunchKernelGGL(attention2_kernel<2>, dim3((unsigned)(a2.nqt * H * nseq)), dim3(256), 0, s, a2);
        } else {
          AttnArgs a2 = aa;
          a2.nqt = qtiles;
          const dim3 grid((unsigned)((int64_t)qtiles * H * nseq));   // 1-D, XCD-aware (kernels_rof.h)
          if (attn_db && aa.len > 128)   // several key tiles: one barrier per tile (double-buffered K / V)
            hipLaunchKernelGGL((attention2_kernel<1, true>), grid, dim3(256), 0, s, a2);
          else hipLaunchKernelGGL(attention2_kernel<1>, grid, dim3(256), 0, s, a2);
        }
      }));
    }
    CHK(rof_gemm(e, L.attn.out, n.ATT.f(), inner, Mg, n.TOK.f(), D, 0, n.TOK.f(), D, s));   // + x (in place: each
    // output element depends only on ATT and on the same TOK element it overwrites)
    // feed-forward: x = ff(x) + x.  The hidden activations FFH have one reader, the second linear: when both linears run on the fp16 x 3 row
    // GEMM the first writes them as a pair image (split once, in its GELU epilogue) and the second multiplies the parts as they are
    // (kernels_gemm3.h, "operands split by their producer")
    int *pair_tab = nullptr;
    int pair_cols = 0;
    {
      TdfDmaArgs d1, d2;
      if (rof_gemm_args(e, L.ff.l1, n.TOK.f(), D, Mg, n.FFH.f(), 4 * D, rof_gelu_act(), nullptr, 0, nullptr, nullptr, d1) == ASX_OK &&
          rof_gemm_args(e, L.ff.l2, n.FFH.f(), 4 * D, Mg, n.TOK.f(), D, 0, n.TOK.f(), D, nullptr, nullptr, d2) == ASX_OK && (4 * D) % 32 == 0 &&
          tdf3h_will_run(e, d1, s) && tdf3h_will_run(e, d2, s)) {
        pair_cols = tdf3_tile_cols(e, d1);
        pair_tab = reinterpret_cast<int *>(n.FFE.p);
      }
    }
    if (L.ff.norm_folded) {
      CHK(rof_rownorm(e, n.TOK.f(), D, D, n.RS.f(), Mg, s));
      CHK(rof_gemm(e, L.ff.l1, n.TOK.f(), D, Mg, n.FFH.f(), 4 * D, rof_gelu_act(), nullptr, 0, s, nullptr, n.RS.f(), pair_tab, nullptr, pair_cols));   // GELU
    } else {
      CHK(rof_rmsnorm(e, n.TOK.f(), D, D, L.ff.norm_g.f(), n.XN.f(), D, Mg, s));
      CHK(rof_gemm(e, L.ff.l1, n.XN.f(), D, Mg, n.FFH.f(), 4 * D, rof_gelu_act(), nullptr, 0, s, nullptr, nullptr, pair_tab, nullptr, pair_cols));   // GELU
    }
    CHK(rof_gemm(e, L.ff.l2, n.FFH.f(), 4 * D, Mg, n.TOK.f(), D, 0, n.TOK.f(), D, s, nullptr, nullptr, nullptr, pair_tab, pair_cols));
  }
  // MelBandRoformer: Transformer(norm_output=True) (mel_band_roformer.py:111,120); in place (a row is read, then written)
  if (out_norm) CHK(rof_rmsnorm(e, n.TOK.f(), D, D, out_norm->f(), n.TOK.f(), D, Mg, s));
  return ASX_OK;
}

// torch.istft(normalized=True) multiplies the spectrum by sqrt(n_fft) before the inverse transform: folded into the synthesis window
// (the fold's envelope keeps the plain window squared)
static int rof_build_win_synth(asx_engine *e) {
  RofNet &n = *e->rof;
  if (!n.cfg.stft_normalized || n.win_synth.p) return ASX_OK;
  std::vector<float> w;
  if ((int)e->custom_window.size() == e->cfg.n_fft) w = e->custom_window;
  else host_window(e->cfg.n_fft, w, e->cfg.win_length);
  const float sc = (float)sqrt((double)e->cfg.n_fft);
  for (auto &v : w) v *= sc;
  CHK(n.win_synth.ensure(w.size() * 4));
  HIPCHK(hipMemcpy(n.win_synth.p, w.data(), w.size() * 4, hipMemcpyHostToDevice));
  return ASX_OK;
}

static int rof_ensure_workspace(asx_engine *e, int B) {
  CHK(rof_build_win_synth(e));
  RofNet &n = *e->rof;
  if (B <= n.ws_batch) return ASX_OK;
  const asx_rof_config &c = n.cfg;
  const int T = e->cfg.segment_size, Fb = c.n_bands, D = c.dim, inner = c.heads * c.dim_head;
  const size_t M = (((size_t)B * T * Fb) + 7) & ~(size_t)7, BT = (size_t)B * T;   // token rows, padded to a multiple of 8 (rof_transformer)
  const int hid = D * (c.mel ? 4 : c.mlp_expansion_factor);
  int maxd = 0;
  for (int d : n.band_dim) maxd = std::max(maxd, d);
  CHK(n.X0.ensure(BT * n.W * 4));
  CHK(n.XB.ensure(BT * maxd * 4 + 256));
  CHK(n.TOK.ensure(M * D * 4));
  CHK(n.XN.ensure(M * D * 4));
  CHK(n.RS.ensure(M * 4));
  CHK(n.QKV.ensure(M * 3 * inner * 4));
  CHK(n.ATT.ensure(M * inner * 4));
  CHK(n.GATE.ensure(M * ((c.heads + 3) / 4 * 4) * 4));
  CHK(n.FFH.ensure(M * 4 * D * 4));
  CHK(n.FFE.ensure(M * (size_t)((4 * D + 127) / 128) * 4 + 256));   // exponent spans of FFH as a pair image
  CHK(n.HID.ensure(BT * hid * 4));
  CHK(n.GLU.ensure(BT * 2 * maxd * 4));
  CHK(n.MASK.ensure(BT * c.num_stems * n.W * 4));
  if (c.mel) CHK(n.MASKB.ensure(BT * c.num_stems * n.MW * 4));
  CHK(n.frames.ensure((size_t)B * c.num_stems * 2 * T * e->cfg.n_fft * 4));
  n.ws_batch = B;
  return ASX_OK;
}

// wave chunks -> separated chunks [B, S, 2, C]
static int rof_chunks_dev(asx_engine *e, const float *wave, const int64_t *d_starts, int64_t n_song, int B, float *out,
                          hipStream_t s) {
  RofNet &n = *e->rof;
  const asx_rof_config &c = n.cfg;
  const int T = e->cfg.segment_size, Fb = c.n_bands, D = c.dim, S = c.num_stems;
  const int64_t C = (int64_t)e->cfg.hop_length * (T - 1);
  const int64_t BT = (int64_t)B * T, M = BT * Fb;
  CHK(rof_ensure_workspace(e, B));
  {
    StftArgs a{};
    a.wave = wave;
    a.chunk_start = d_starts;
    a.n_song = n_song;
    a.trim = 0;
    a.C = C;
    a.hop = e->cfg.hop_length;
    a.T = T;
    a.dim_f = e->cfg.dim_f;
    a.zero_low = 0;
    a.tf_layout = 2;
    a.spec = n.X0.f();
    a.window = e->d_window.f();
    a.tw = reinterpret_cast<const float2 *>(e->d_tw.p);
    // torch.stft(normalized=True) (bs_roformer.py:332, 384): the spectrum times n_fft^-1/2
    a.sign = c.stft_normalized ? (float)(1.0 / sqrt((double)e->cfg.n_fft)) : 1.0f;
    FftPlan p = e->plan;
    CHK(timed(e, ASX_PROF_STFT, 0.0, 4.0 * ((double)B * 2 * C + (double)BT * n.W), s, [&]() {
      hipLaunchKernelGGL(stft_kernel, dim3(T, 2, B), dim3(256), stft_lds(p), s, a, p);
    }));
  }
  // band split (bs_roformer.py:163-185): per band RMSNorm + Linear -> TOK[(b t), band, :]
  for (int j = 0; j < Fb; ++j) {
    const int din = n.band_dim[j];
    CHK(rof_rmsnorm(e, n.X0.f() + n.band_off[j], n.W, din, n.bs_gamma[j].f(), n.XB.f(), din, BT, s));
    CHK(rof_gemm(e, n.bs_lin[j], n.XB.f(), din, BT, n.TOK.f() + (int64_t)j * D, (int64_t)Fb * D, 0, nullptr, 0, s));
  }
  for (int i = 0; i < c.depth; ++i) {
    CHK(rof_transformer(e, n.time_l[i], true, B, s, c.mel ? &n.tnorm_t[i] : nullptr));
    CHK(rof_transformer(e, n.freq_l[i], false, B, s, c.mel ? &n.tnorm_f[i] : nullptr));
  }
  const float *feat = n.TOK.f();
  if (!c.mel) {
    CHK(rof_rmsnorm(e, n.TOK.f(), D, D, n.final_g.f(), n.XN.f(), D, M, s));
    feat = n.XN.f();
  }
  // mask estimators (bs_roformer.py:205-229): per stem, per band MLP (tanh) + GLU -> MASK[b, stem, t, band slice]
  for (int st = 0; st < S; ++st)
    for (int j = 0; j < Fb; ++j) {
      const int din = n.band_dim[j];
      auto &mlp = n.mask[st][j];
      const float *cur = feat + (int64_t)j * D;
      int64_t ld = (int64_t)Fb * D;
      for (size_t li = 0; li + 1 < mlp.size(); ++li) {
        float *dst = (li & 1) ? n.FFH.f() : n.HID.f();
        CHK(rof_gemm(e, mlp[li], cur, ld, BT, dst, mlp[li].n, 3, nullptr, 0, s));   // tanh
        cur = dst;
        ld = mlp[li].n;
      }
      CHK(rof_gemm(e, mlp.back(), cur, ld, BT, n.GLU.f(), 2 * din, 0, nullptr, 0, s));
      {
        const int64_t tot = BT * din;
        float *dst = c.mel ? n.MASKB.f() + n.mask_off[j] : n.MASK.f() + n.band_off[j];
        const float *src = n.GLU.f();
        const int64_t Wl = c.mel ? n.MW : n.W;
        CHK(timed(e, ASX_PROF_MISC, 0.0, 12.0 * tot, s, [&]() {
          hipLaunchKernelGGL(glu_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s, src, din, BT, T, S, st, dst,
                             Wl);
        }));
      }
    }
  if (c.mel) {   // band masks -> per-bin masks: sum over the covering bands / their number (mel_band_roformer.py:404-416)
    const int64_t tot = BT * S * n.W;
    CHK(timed(e, ASX_PROF_MISC, 0.0, 4.0 * (double)BT * S * (n.W + n.MW), s, [&]() {
      hipLaunchKernelGGL(mel_mask_merge_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s, n.MASKB.f(), n.MW, n.W,
                         reinterpret_cast<const int *>(n.d_bstart.p), reinterpret_cast<const int *>(n.d_moff.p),
                         reinterpret_cast<const int *>(n.d_jlo.p), reinterpret_cast<const int *>(n.d_jhi.p), n.MASK.f(), tot);
    }));
  }
  {
    IstftArgs a{};
    a.spec = n.X0.f();
    a.mask = n.MASK.f();
    a.T = T;
    a.dim_f = e->cfg.dim_f;
    a.tf_layout = 2;
    a.combine = 0;
    a.frames = n.frames.f();
    a.window = c.stft_normalized ? n.win_synth.f() : e->d_window.f();
    a.tw = reinterpret_cast<const float2 *>(e->d_tw.p);
    a.n_inst = S;
    FftPlan p = e->plan;
    CHK(timed(e, ASX_PROF_ISTFT, 0.0, 4.0 * ((double)BT * n.W * (1 + S) + (double)B * S * 2 * T * e->cfg.n_fft), s,
              [&]() { hipLaunchKernelGGL(istft_kernel, dim3(T, 2, B * S), dim3(256), istft_lds(p), s, a, p); }));
  }
  CHK(ola_launch(e, n.frames.f(), e->d_env.f(), nullptr, B * S, T, C, out, s));
  return ASX_OK;
}

static double rof_flops(const asx_engine *e, int batch) {
  if (!e->rof || !e->rof->begun) return 0.0;
  const RofNet &n = *e->rof;
  const asx_rof_config &c = n.cfg;
  const double T = e->cfg.segment_size, Fb = c.n_bands, D = c.dim, inner = c.heads * c.dim_head;
  const double M = T * Fb;
  double fl = 0;
  for (int d : n.band_dim) fl += 2.0 * T * d * D;
  const double per_layer = 2.0 * M * (D * 3 * inner + D * c.heads + inner * D + 2 * D * 4 * D);
  const double att_t = 4.0 * Fb * c.heads * T * T * c.dim_head, att_f = 4.0 * T * c.heads * Fb * Fb * c.dim_head;
  fl += c.depth * (c.time_depth * (per_layer + att_t) + c.freq_depth * (per_layer + att_f));
  const double hid = D * (c.mel ? 4 : c.mlp_expansion_factor);
  for (int d : n.band_dim) {
    double m = 0, in = D;
    for (int li = 0; li + 1 < c.mask_estimator_depth + (c.mel ? 1 : 0); ++li) {
      m += 2.0 * T * in * hid;
      in = hid;
    }
    m += 2.0 * T * in * 2 * d;
    fl += c.num_stems * m;
  }
  return fl * batch;
}
