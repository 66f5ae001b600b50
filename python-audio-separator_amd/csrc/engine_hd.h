// Demucs v3 (HDemucs, the reference's `hdemucs_mmi`) on the engine: uvr_lib_v5/demucs/hdemucs.py:362-782 with
// demucs.py's BLSTM / LocalState DConv inserts, under apply_model (apply.py:124-260) whose leaf call here runs every
// chunk at its own length (HDemucs has no valid_length, apply.py:251-256).  Included by asx.hip after engine_ht.h.
//
// Structure built (HDemucs' defaults): D = depth - 2 strided levels on both branches -- the same HEncLayer / HDecLayer
// as Demucs v4, so they are e->ht's levels (engine_ht.h) -- then
//   level A (last_freq): the remaining kernel_size frequency rows collapse into one (conv kernel_size x 1, no pad); the
//            waveform branch contributes a bare strided conv that is added before the norm (`inject`), and ends here;
//   level Z: time-only conv (kernel 2*time_stride) on the merged branch.
// Both have GroupNorm(norm_groups) around their GELU / GLU and a BLSTM + LocalState in each DConv layer.
// Layout: [B, T, C] channels-last (T = spectrogram frames).
#pragma once

struct HdLstm {
  HtGemm ih0, ih1, lin;
  DevBuf whh0, whh1;
};
struct HdAttn {
  HtGemm qkvd, proj;
};
struct HdIns {
  HdLstm lstm;
  HdAttn attn;
};
struct HdNorm {
  DevBuf w, b;
};

struct HdNet {
  asx_hd_config cfg{};
  bool begun = false, ready = false;
  int D = 0, CA = 0, CZ = 0, CD = 0;   // CD = channels of level D-1
  HtEnc encA, encZ;                    // conv + DConv + rewrite (dense: the norm sits between it and the GLU)
  HtGemm tencA;
  std::vector<HdIns> insA, insZ;
  HdNorm n1A, n2A, n1Z, n2Z;           // encoder norms
  HtGemm decZ_rw, decZ_tr, decA_rw, decA_tr, tdecA_tr;
  HdNorm dZn1, dZn2, dAn1, dAn2, tdAn2;
  DevBuf ws, tmp_out, starts;
  int64_t ws_len = 0;
  int ws_batch = 0;
  struct {
    float *inj, *ya, *rwA, *skA, *yz, *rwZ, *skZ, *g, *tr, *dAin, *pre, *qkvd, *att;
  } b;
};

static void hd_free(HdNet &n) {
  auto fg = [](HtGemm &g) {
    g.w.release();
    g.b.release();
    g.wh.release();
  };
  for (HtEnc *E : {&n.encA, &n.encZ}) {
    fg(E->conv);
    fg(E->rewrite);
    for (auto &d : E->dc) {
      fg(d.c1);
      fg(d.c2);
      for (DevBuf *p : {&d.g1w, &d.g1b, &d.g2w, &d.g2b, &d.ls}) p->release();
    }
    E->dc.clear();
  }
  for (auto *v : {&n.insA, &n.insZ}) {
    for (auto &x : *v) {
      for (HtGemm *g : {&x.lstm.ih0, &x.lstm.ih1, &x.lstm.lin, &x.attn.qkvd, &x.attn.proj}) fg(*g);
      x.lstm.whh0.release();
      x.lstm.whh1.release();
    }
    v->clear();
  }
  for (HtGemm *g : {&n.tencA, &n.decZ_rw, &n.decZ_tr, &n.decA_rw, &n.decA_tr, &n.tdecA_tr}) fg(*g);
  for (HdNorm *g : {&n.n1A, &n.n2A, &n.n1Z, &n.n2Z, &n.dZn1, &n.dZn2, &n.dAn1, &n.dAn2, &n.tdAn2}) {
    g->w.release();
    g->b.release();
  }
  for (DevBuf *p : {&n.ws, &n.tmp_out, &n.starts}) p->release();
  n.ready = false;
  n.begun = false;
  n.ws_batch = 0;
  n.ws_len = 0;
}

static void hd_destroy(HdNet *n) {
  hd_free(*n);
  delete n;
}

static int hd_norm_load(asx_engine *e, HdNorm &g, const std::string &name, int c) {
  CHK(ht_up_named(e, g.w, name + ".weight", c));
  return ht_up_named(e, g.b, name + ".bias", c);
}

// nn.LSTM layer `l` of both directions as one input GEMM: rows [fwd i f g o | rev i f g o], bias = b_ih + b_hh
static int hd_pack_lstm_layer(asx_engine *e, HtGemm &g, DevBuf &whh, const std::string &p, int l, int H, int din) {
  std::vector<float> pw((size_t)8 * H * din), pb((size_t)8 * H), ph((size_t)8 * H * H);
  for (int dir = 0; dir < 2; ++dir) {
    const std::string sfx = "_l" + std::to_string(l) + (dir ? "_reverse" : "");
    const float *wi, *wh, *bi, *bh;
    CHK(get_tensor(e, p + ".lstm.weight_ih" + sfx, (int64_t)4 * H * din, &wi));
    CHK(get_tensor(e, p + ".lstm.weight_hh" + sfx, (int64_t)4 * H * H, &wh));
    CHK(get_tensor(e, p + ".lstm.bias_ih" + sfx, 4 * H, &bi));
    CHK(get_tensor(e, p + ".lstm.bias_hh" + sfx, 4 * H, &bh));
    std::copy(wi, wi + (size_t)4 * H * din, pw.begin() + (size_t)dir * 4 * H * din);
    std::copy(wh, wh + (size_t)4 * H * H, ph.begin() + (size_t)dir * 4 * H * H);
    for (int i = 0; i < 4 * H; ++i) pb[(size_t)dir * 4 * H + i] = bi[i] + bh[i];
  }
  g.n = 8 * H;
  g.k = din;
  CHK(ht_up(g.w, pw));
  CHK(ht_up(g.b, pb));
  return ht_up(whh, ph);
}

static int hd_load_ins(asx_engine *e, HdIns &x, const std::string &q, int H) {
  REQUIRE(H % 16 == 0, "DConv hidden size %d of a BLSTM / LocalState level must be a multiple of 16", H);
  const std::string pl = q + ".3", pa = q + ".4";
  CHK(hd_pack_lstm_layer(e, x.lstm.ih0, x.lstm.whh0, pl, 0, H, H));
  CHK(hd_pack_lstm_layer(e, x.lstm.ih1, x.lstm.whh1, pl, 1, H, 2 * H));
  CHK(ht_pack_linear(e, x.lstm.lin, pl + ".linear.weight", pl + ".linear.bias", H, 0, H, 2 * H, nullptr));
  // LocalState: query | key | content | query_decay as one GEMM (1x1 convs, demucs.py:169-181)
  const int N = 3 * H + 16;
  std::vector<float> pw((size_t)N * H), pb((size_t)N);
  const char *names[4] = {"query", "key", "content", "query_decay"};
  int row = 0;
  for (int i = 0; i < 4; ++i) {
    const int rows = i < 3 ? H : 16;
    const float *w, *b;
    CHK(get_tensor(e, pa + "." + names[i] + ".weight", (int64_t)rows * H, &w));
    CHK(get_tensor(e, pa + "." + names[i] + ".bias", rows, &b));
    std::copy(w, w + (size_t)rows * H, pw.begin() + (size_t)row * H);
    std::copy(b, b + rows, pb.begin() + row);
    row += rows;
  }
  x.attn.qkvd.n = N;
  x.attn.qkvd.k = H;
  CHK(ht_up(x.attn.qkvd.w, pw));
  CHK(ht_up(x.attn.qkvd.b, pb));
  return ht_pack_linear(e, x.attn.proj, pa + ".proj.weight", pa + ".proj.bias", H, 0, H, H, nullptr);
}

// Conv2d [cout, cin, 3, 3] applied to a single frequency row with padding 1: only the middle frequency tap sees data
static int hd_pack_conv_mid(asx_engine *e, HtGemm &g, const std::string &name, int cout, int cin) {
  const float *w, *b;
  CHK(get_tensor(e, name + ".weight", (int64_t)cout * cin * 9, &w));
  CHK(get_tensor(e, name + ".bias", cout, &b));
  const int K = 3 * cin;
  std::vector<float> pw((size_t)cout * K), pb(b, b + cout);
  for (int n = 0; n < cout; ++n)
    for (int ci = 0; ci < cin; ++ci)
      for (int c = 0; c < 3; ++c) pw[(size_t)n * K + (size_t)c * cin + ci] = w[(((size_t)n * cin + ci) * 3 + 1) * 3 + c];
  g.n = cout;
  g.k = K;
  CHK(ht_up(g.w, pw));
  return ht_up(g.b, pb);
}

static int hd_commit(asx_engine *e) {
  HdNet &h = *e->hd;
  HtNet &n = *e->ht;
  const asx_hd_config &c = h.cfg;
  const int D = c.depth - 2;
  h.D = D;
  REQUIRE(make_plan(c.nfft, &n.plan), "nfft/2 = %d must factor into {2,3,5}", c.nfft / 2);
  n.F.assign(D + 1, 0);
  n.L.assign(D + 1, 0);
  n.C.assign(D, 0);
  n.F[0] = c.nfft / 2;
  int ch = c.channels;
  for (int i = 0; i < D; ++i) {
    REQUIRE(n.F[i] > c.kernel_size && n.F[i] % c.stride == 0, "level %d has %d frequency rows", i, n.F[i]);
    n.F[i + 1] = n.F[i] / c.stride;
    n.C[i] = ch;
    REQUIRE(ch % 4 == 0, "channel counts must be multiples of 4 (level %d has %d)", i, ch);
    ch *= c.growth;
  }
  REQUIRE(n.F[D] == c.kernel_size, "level %d must see kernel_size = %d frequency rows (nfft / 2 / stride^(depth-2)), got %d", D,
          c.kernel_size, n.F[D]);
  int hop = c.nfft / 4, q = 1;
  for (int i = 0; i <= D; ++i) q *= c.stride;
  REQUIRE(q == hop, "stride^(depth-1) = %d must equal the hop %d so that both branches meet on the frame grid", q, hop);
  h.CD = n.C[D - 1];
  h.CA = h.CD * c.growth;
  h.CZ = h.CA * c.growth;
  CHK(ht_commit_tables(e, c.segment_samples));
  n.enc.assign(D, HtEnc());
  n.tenc.assign(D, HtEnc());
  n.dec.assign(D, HtDec());
  n.tdec.assign(D, HtDec());
  for (int i = 0; i < D; ++i) CHK(ht_commit_level(e, i, c.depth - 1 - i, c.depth - 2 - i));
  if (c.freq_emb_scale != 0.f) {
    const float *w;
    CHK(get_tensor(e, "freq_emb.embedding.weight", (int64_t)n.F[1] * n.C[0], &w));
    std::vector<float> fe((size_t)n.F[1] * n.C[0]);
    for (size_t i = 0; i < fe.size(); ++i) fe[i] = c.freq_emb_scale * (w[i] * 10.0f);
    CHK(ht_up(n.femb, fe));
  }
  n.Ct = 0;
  n.hidden = 0;
  // level A
  const std::string sA = std::to_string(D), sZ = std::to_string(D + 1);
  const int CA = h.CA, CZ = h.CZ, CD = h.CD, G = c.norm_groups;
  REQUIRE(CA % (4 * G) == 0 && CD % G == 0, "norm_groups %d does not divide the channel counts", G);
  h.encA.cin = CD;
  h.encA.cout = CA;
  CHK(ht_pack_conv(e, h.encA.conv, "encoder." + sA + ".conv", CA, CD, c.kernel_size, 1, false));
  CHK(ht_pack_conv(e, h.tencA, "tencoder." + sA + ".conv", CA, CD, c.kernel_size, 1, false));
  CHK(ht_pack_conv(e, h.encA.rewrite, "encoder." + sA + ".rewrite", 2 * CA, CA, 1, 1, false));
  CHK(hd_norm_load(e, h.n1A, "encoder." + sA + ".norm1", CA));
  CHK(hd_norm_load(e, h.n2A, "encoder." + sA + ".norm2", 2 * CA));
  h.encZ.cin = CA;
  h.encZ.cout = CZ;
  CHK(ht_pack_conv(e, h.encZ.conv, "encoder." + sZ + ".conv", CZ, CA, 2 * c.time_stride, 1, false));
  CHK(ht_pack_conv(e, h.encZ.rewrite, "encoder." + sZ + ".rewrite", 2 * CZ, CZ, 1, 1, false));
  CHK(hd_norm_load(e, h.n1Z, "encoder." + sZ + ".norm1", CZ));
  CHK(hd_norm_load(e, h.n2Z, "encoder." + sZ + ".norm2", 2 * CZ));
  h.encA.dc.assign(c.dconv_depth, HtDconv());
  h.encZ.dc.assign(c.dconv_depth, HtDconv());
  h.insA.assign(c.dconv_depth, HdIns());
  h.insZ.assign(c.dconv_depth, HdIns());
  for (int d = 0; d < c.dconv_depth; ++d) {
    CHK(ht_load_dconv(e, h.encA.dc[d], "encoder." + sA + ".dconv", CA, c.dconv_comp, d, 2));
    CHK(ht_load_dconv(e, h.encZ.dc[d], "encoder." + sZ + ".dconv", CZ, c.dconv_comp, d, 2));
    CHK(hd_load_ins(e, h.insA[d], "encoder." + sA + ".dconv.layers." + std::to_string(d), CA / c.dconv_comp));
    CHK(hd_load_ins(e, h.insZ[d], "encoder." + sZ + ".dconv.layers." + std::to_string(d), CZ / c.dconv_comp));
  }
  // decoders: decoder.0 undoes level Z, decoder.1 level A, tdecoder.0 the waveform half of level A
  CHK(ht_pack_conv(e, h.decZ_rw, "decoder.0.rewrite", 2 * CZ, CZ, 3, 1, false, 0, 0, true));
  CHK(hd_norm_load(e, h.dZn1, "decoder.0.norm1", 2 * CZ));
  CHK(ht_pack_convtr(e, h.decZ_tr, "decoder.0.conv_tr", CZ, CA, 2 * c.time_stride, c.time_stride));
  CHK(hd_norm_load(e, h.dZn2, "decoder.0.norm2", CA));
  CHK(hd_pack_conv_mid(e, h.decA_rw, "decoder.1.rewrite", 2 * CA, CA));
  CHK(hd_norm_load(e, h.dAn1, "decoder.1.norm1", 2 * CA));
  CHK(ht_pack_convtr(e, h.decA_tr, "decoder.1.conv_tr", CA, CD, c.kernel_size, c.stride));
  CHK(hd_norm_load(e, h.dAn2, "decoder.1.norm2", CD));
  CHK(ht_pack_convtr(e, h.tdecA_tr, "tdecoder.0.conv_tr", CA, CD, c.kernel_size, c.stride));
  CHK(hd_norm_load(e, h.tdAn2, "tdecoder.0.norm2", CD));
  (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&ht_stft_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)stft_lds(n.plan));
  (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&ht_istft_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)ht_istft_lds(n.plan));
  h.ready = true;
  return ASX_OK;
}

// ---- workspace --------------------------------------------------------------------------------------------------
struct HdDims {
  int T, T2, nfrA, nfrZ, stepsA, stepsZ;
};
static HdDims hd_dims(const HdNet &h, int64_t L) {
  HdDims d;
  const int hop = h.cfg.nfft / 4, ts = h.cfg.time_stride;
  d.T = (int)((L + hop - 1) / hop);
  d.T2 = (d.T + ts - 1) / ts;
  d.stepsA = d.T > 200 ? 200 : d.T;
  d.nfrA = d.T > 200 ? (d.T + 99) / 100 : 1;
  d.stepsZ = d.T2 > 200 ? 200 : d.T2;
  d.nfrZ = d.T2 > 200 ? (d.T2 + 99) / 100 : 1;
  return d;
}

static int hd_ensure_workspace(asx_engine *e, int B, int64_t L) {
  HdNet &h = *e->hd;
  HtNet &n = *e->ht;
  const asx_hd_config &c = h.cfg;
  const int D = h.D;
  const HdDims d = hd_dims(h, L);
  if (L != h.ws_len) {
    n.T = d.T;
    n.L[0] = L;
    for (int i = 0; i < D; ++i) n.L[i + 1] = (n.L[i] + c.stride - 1) / c.stride;
    n.ws_batch = 0;
    h.ws_batch = 0;
    h.ws_len = L;
  }
  CHK(ht_ensure_workspace(e, B));
  if (B <= h.ws_batch) return ASX_OK;
  const int CA = h.CA, CZ = h.CZ, HA = CA / c.dconv_comp, HZ = CZ / c.dconv_comp;
  // the DConv scratch of e->ht (h, rowstat) is sized by the strided levels; levels A / Z must fit in it
  {
    size_t hcap = 0, rcap = 0;
    for (int i = 0; i < D; ++i) {
      const int hp = (n.C[i] / c.dconv_comp + 3) & ~3;
      const size_t rows = std::max((size_t)B * d.T * n.F[i + 1], (size_t)B * n.L[i + 1]);
      hcap = std::max(hcap, rows * hp);
      const int np = ht_glu_rows(n.C[i]);
      rcap = std::max(rcap, rows * std::max<size_t>((np + gg_tile_n(np, true) - 1) / gg_tile_n(np, true), 1) * 2);
    }
    const size_t hneed = std::max((size_t)B * d.T * HA, (size_t)B * d.T2 * HZ);
    auto gtiles = [](int ch) {
      const int np = ht_glu_rows(ch);
      return (size_t)std::max((np + gg_tile_n(np, true) - 1) / gg_tile_n(np, true), (ch / 2 + 15) / 16);   // c2 (GLU rows) and c1 (hid columns)
    };
    const size_t rneed = std::max((size_t)B * d.T * gtiles(CA), (size_t)B * d.T2 * gtiles(CZ)) * 2;
    REQUIRE(hneed <= hcap && rneed <= rcap, "DConv scratch of the strided levels is too small for the inner levels (%zu/%zu, %zu/%zu)",
            hneed, hcap, rneed, rcap);
  }
  size_t off = 0;
  std::vector<std::pair<float **, size_t>> plan;
  auto want = [&](float *&p, size_t floats) {
    plan.push_back({&p, off});
    off += (floats * 4 + 255) & ~(size_t)255;
  };
  auto &b = h.b;
  const size_t BT = (size_t)B * d.T, BT2 = (size_t)B * d.T2;
  want(b.inj, BT * CA);
  want(b.ya, BT * CA);
  want(b.rwA, BT * 2 * CA);
  want(b.skA, BT * CA);
  want(b.yz, BT2 * CZ);
  want(b.rwZ, BT2 * 2 * CZ);
  want(b.skZ, BT2 * CZ);
  want(b.g, std::max(BT2 * CZ, BT * CA));
  want(b.tr, std::max(std::max((size_t)B * (d.T2 + 1) * c.time_stride * CA, BT * n.F[D] * h.CD), (size_t)B * (d.T + 1) * c.stride * h.CD));
  want(b.dAin, BT * CA);
  want(b.pre, BT * CA);
  want(b.qkvd, std::max(BT * (3 * HA + 16), BT2 * (3 * HZ + 16)));
  want(b.att, std::max(BT * HA, BT2 * HZ));
  CHK(h.ws.ensure(off));
  for (auto &pr : plan) *pr.first = reinterpret_cast<float *>(reinterpret_cast<char *>(h.ws.p) + pr.second);
  h.ws_batch = B;
  return ASX_OK;
}

// ---- stages -----------------------------------------------------------------------------------------------------
// GroupNorm(G, C) over x [B, Rin, C] -> dst [B, R, C] = rows [r0, r0 + R) (hd_gn_kernel modes); Rin = 0: Rin = R
static int hd_group_norm(asx_engine *e, const float *x, int B, int64_t R, int C, const HdNorm &g, int mode, float *dst, const float *skip,
                         hipStream_t s, int64_t Rin = 0, int64_t r0 = 0) {
  HtNet &n = *e->ht;
  const int G = e->hd->cfg.norm_groups;
  if (!Rin) Rin = R;
  CHK(ht_stats(e, x, B, Rin, C, C / G, C, C, G, n.b.acc_g, s));
  const int64_t total = (int64_t)B * R * (mode == 1 ? C / 2 : C);
  return timed(e, ASX_PROF_MISC, 0.0, 4.0 * (double)B * R * C * 2, s, [&]() {
    hipLaunchKernelGGL(hd_gn_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, x, Rin, r0, R, C, G, n.b.acc_g, g.w.f(), g.b.f(),
                       mode, dst, skip, total);
  });
}

// ---- chunk groups ---------------------------------------------------------------------------------------------------
// A forward call works on one or more groups of chunks, one length each.  Every group has its own activations: the
// first lives in e->ht / e->hd, the others in clones of the two nets that alias the weights and own a workspace
// (e->ht_cl / e->hd_cl).  The groups advance phase by phase and meet at every BLSTM, where the sequences of all groups
// with the same step count (200 for anything longer than 200 frames, demucs.py:41-48) share the 2 x 200 recurrence
// launches: the step chain is latency-bound and costs the same for 4 sequences as for 400, and a song's chunks come in
// only a few lengths (the split length plus a tail or two per shift).
struct HdGroup {
  HtNet *ht;
  HdNet *hd;
  int B;
  int64_t L;
  const float *seg;   // [B, 2, L]
  float *out;         // [B, S, 2, L]
  HdDims dm;
};

static void hd_drop_clones(asx_engine *e) {
  for (HtNet *n : e->ht_cl) {
    for (DevBuf *p : {&n->ws, &n->acc, &n->chunk_out, &n->d_starts, &n->seg, &n->ref}) p->release();   // ref_acc is never the clone's own
    delete n;
  }
  for (HdNet *h : e->hd_cl) {
    for (DevBuf *p : {&h->ws, &h->tmp_out, &h->starts}) p->release();
    delete h;
  }
  e->ht_cl.clear();
  e->hd_cl.clear();
  e->hd_lstm_ws.release();
}

static int hd_clone(asx_engine *e, size_t idx) {
  while (e->hd_cl.size() <= idx) {
    HtNet *n = new HtNet(*e->ht);
    for (DevBuf *p : {&n->ws, &n->acc, &n->chunk_out, &n->d_starts, &n->seg, &n->ref, &n->ref_acc}) *p = DevBuf();
    n->ws_batch = 0;
    HdNet *h = new HdNet(*e->hd);
    for (DevBuf *p : {&h->ws, &h->tmp_out, &h->starts}) *p = DevBuf();
    h->ws_batch = 0;
    h->ws_len = 0;
    e->ht_cl.push_back(n);
    e->hd_cl.push_back(h);
  }
  return ASX_OK;
}

// BLSTM(dim, layers=2, max_steps=200, skip=True) (demucs.py:33-66) in place on the DConv hidden buffers b.h [B, T, H] of
// every group, level A (T frames) or Z (T2)
static int hd_blstm_joint(asx_engine *e, std::vector<HdGroup> &G, bool levelZ, size_t d, hipStream_t s) {
  HdNet &h0 = *G[0].hd;
  const HdLstm &L = (levelZ ? h0.insZ : h0.insA)[d].lstm;
  const int H = (levelZ ? h0.CZ : h0.CA) / h0.cfg.dconv_comp;
  std::map<int, std::vector<size_t>> by_steps;
  for (size_t gi = 0; gi < G.size(); ++gi) by_steps[levelZ ? G[gi].dm.stepsZ : G[gi].dm.stepsA].push_back(gi);
  auto pad = [](size_t f) { return (f + 63) & ~(size_t)63; };
  auto count = [&](const std::vector<size_t> &members) {
    int N = 0;
    for (size_t gi : members) N += G[gi].B * (levelZ ? G[gi].dm.nfrZ : G[gi].dm.nfrA);
    return N;
  };
  {   // one allocation for the largest set, before anything of this call is in flight
    size_t need = 0;
    for (auto &kv : by_steps) {
      const size_t N = (size_t)count(kv.second), rows = (size_t)kv.first * N;
      need = std::max(need, pad(rows * H) + pad(rows * 8 * H) + pad(rows * 2 * H) + pad(4 * N * H) + pad(2 * N * H));
    }
    CHK(e->hd_lstm_ws.ensure(need * 4));
  }
  for (auto &kv : by_steps) {
    const int steps = kv.first;
    const int N = count(kv.second);
    const int64_t rows = (int64_t)steps * N;
    const size_t o_xs = 0, o_xp = o_xs + pad((size_t)rows * H), o_out = o_xp + pad((size_t)rows * 8 * H), o_hb = o_out + pad((size_t)rows * 2 * H),
                 o_cst = o_hb + pad((size_t)4 * N * H);
    float *xs = e->hd_lstm_ws.f() + o_xs, *xp = e->hd_lstm_ws.f() + o_xp, *out = e->hd_lstm_ws.f() + o_out, *hb = e->hd_lstm_ws.f() + o_hb,
          *cst = e->hd_lstm_ws.f() + o_cst;
    int n_off = 0;
    for (size_t gi : kv.second) {
      const HdGroup &g = G[gi];
      const int T = levelZ ? g.dm.T2 : g.dm.T, nfr = levelZ ? g.dm.nfrZ : g.dm.nfrA;
      const int64_t tot = (int64_t)steps * g.B * nfr * H;
      CHK(timed(e, ASX_PROF_MISC, 0.0, 8.0 * (double)tot, s, [&]() {
        hipLaunchKernelGGL(hd_lstm_frame_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s, g.ht->b.h, g.B, T, H, nfr, steps, 100, N,
                           n_off, xs, tot);
      }));
      n_off += g.B * nfr;
    }
    const float *in = xs;
    int din = H;
    for (int layer = 0; layer < 2; ++layer) {
      const HtGemm &ih = layer ? L.ih1 : L.ih0;
      const float *whh = layer ? L.whh1.f() : L.whh0.f();
      CHK(ht_linear(e, ih, in, din, rows, xp, 8 * H, 0, nullptr, 0, s));
      HIPCHK(hipMemsetAsync(hb, 0, (size_t)4 * N * H * 4, s));
      HIPCHK(hipMemsetAsync(cst, 0, (size_t)2 * N * H * 4, s));
      // sequence tiles of 16 are spread evenly over the z blocks (<= 8 tiles each): 18 tiles run as 6 + 6 + 6, not 8 + 8 + 2
      const int tiles = (N + 15) / 16, nz = (tiles + 7) / 8, tpz = (tiles + nz - 1) / nz;
      const dim3 grid((unsigned)(H / 4), 2, (unsigned)nz);
      CHK(timed(e, ASX_PROF_CONV1X1, 2.0 * steps * 2.0 * N * 4.0 * H * H, 4.0 * steps * 2.0 * (4.0 * H * H + 10.0 * N * H), s, [&]() {
        for (int st = 0; st < steps; ++st) {
          float *hp = hb + (size_t)(st & 1) * 2 * N * H, *hn = hb + (size_t)((st + 1) & 1) * 2 * N * H;
          hipLaunchKernelGGL(hd_lstm_step_kernel, grid, dim3(256), 0, s, xp, whh, hp, hn, cst, out, N, H, st, steps, tpz);
        }
      }));
      in = out;
      din = 2 * H;
    }
    CHK(ht_linear(e, L.lin, out, 2 * H, rows, xs, H, 0, nullptr, 0, s));
    n_off = 0;
    for (size_t gi : kv.second) {
      const HdGroup &g = G[gi];
      const int T = levelZ ? g.dm.T2 : g.dm.T, nfr = levelZ ? g.dm.nfrZ : g.dm.nfrA;
      const int64_t tot = (int64_t)g.B * T * H;
      CHK(timed(e, ASX_PROF_MISC, 0.0, 12.0 * (double)tot, s, [&]() {
        hipLaunchKernelGGL(hd_lstm_unframe_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s, xs, g.B, T, H, nfr, 100, N, n_off,
                           g.ht->b.h, tot);
      }));
      n_off += g.B * nfr;
    }
  }
  return ASX_OK;
}

template <int DH>
static void hd_launch_attn(const float *qkvd, int ld, int T, int H, float *out, int B, hipStream_t s) {
  hipLaunchKernelGGL((hd_local_attn_kernel<DH>), dim3((unsigned)((T + 63) / 64), 4, (unsigned)B), dim3(64), 0, s, qkvd, ld, T, H, out);
}

// LocalState in place on hbuf [B, T, H] (demucs.py:197-221)
static int hd_local_state(asx_engine *e, const HdAttn &A, float *hbuf, int B, int T, int H, hipStream_t s) {
  auto &b = e->hd->b;
  const int ld = 3 * H + 16, dh = H / 4;
  const int64_t M = (int64_t)B * T;
  CHK(ht_linear(e, A.qkvd, hbuf, H, M, b.qkvd, ld, 0, nullptr, 0, s));
  int bad = 0;
  CHK(timed(e, ASX_PROF_CONV1X1, 4.0 * B * 4.0 * (double)T * T * dh, 4.0 * (double)M * (ld + H), s, [&]() {
    if (dh % 16 == 0 && (dh <= 64 || dh == 96)) {
      // the flash attention of the v4 transformer (kernels_ht.h) with the per-query decay slope and the -100 diagonal
      MhaArgs a{};
      a.q = b.qkvd;
      a.k = b.qkvd + H;
      a.v = b.qkvd + 2 * H;
      a.out = b.att;
      a.ldq = a.ldk = a.ldv = ld;
      a.ldo = H;
      a.nq = a.nk = T;
      a.scale = 1.0f / sqrtf((float)dh);
      static const int attn_exact = getenv("ASX_ATTN_EXACT") != nullptr;
      a.exact = attn_exact;
      a.decay = b.qkvd + 3 * H;
      a.ldd = ld;
      a.nqt = (T + 63) / 64;
      a.heads = 4;
      const dim3 grid((unsigned)(a.nqt * 4 * B));   // 1-D, XCD-aware (kernels_ht.h)
      static const bool hd_mha_db = !(getenv("ASX_MHA_DB") && atoi(getenv("ASX_MHA_DB")) == 0);
      switch (dh / 16) {
        case 1: hipLaunchKernelGGL((mha_kernel<1, true>), grid, dim3(256), 0, s, a); break;
        case 2: hipLaunchKernelGGL((mha_kernel<2, true>), grid, dim3(256), 0, s, a); break;
        case 3:
          if (hd_mha_db) hipLaunchKernelGGL((mha_kernel<3, true, true>), grid, dim3(256), 0, s, a);
          else hipLaunchKernelGGL((mha_kernel<3, true>), grid, dim3(256), 0, s, a);
          break;
        case 4: hipLaunchKernelGGL((mha_kernel<4, true>), grid, dim3(256), 0, s, a); break;
        default: hipLaunchKernelGGL((mha_kernel<6, true>), grid, dim3(256), 0, s, a); break;
      }
      return;
    }
    switch (dh) {   // narrow heads: one thread per query
      case 4: hd_launch_attn<4>(b.qkvd, ld, T, H, b.att, B, s); break;
      case 8: hd_launch_attn<8>(b.qkvd, ld, T, H, b.att, B, s); break;
      case 12: hd_launch_attn<12>(b.qkvd, ld, T, H, b.att, B, s); break;
      case 24: hd_launch_attn<24>(b.qkvd, ld, T, H, b.att, B, s); break;
      default: bad = 1;
    }
  }));
  REQUIRE(!bad, "LocalState head dim %d is not built (4, 8, 12, 24 and multiples of 16 up to 64, 96)", dh);
  return ht_linear(e, A.proj, b.att, H, M, hbuf, H, 0, hbuf, H, s);
}

// HDemucs.forward (hdemucs.py:670-782) of the group in e->ht / e->hd, first phase: spectrogram, standardisation, the
// strided encoder levels, level A up to the GELU of its first DConv layer
static int hd_phase_front(asx_engine *e, const HdGroup &gr, hipStream_t s) {
  HdNet &h = *e->hd;
  HtNet &n = *e->ht;
  const asx_hd_config &c = h.cfg;
  const int B = gr.B;
  const int64_t L = gr.L;
  const float *seg = gr.seg;
  CHK(hd_ensure_workspace(e, B, L));
  auto &b = n.b;
  auto &w = h.b;
  const HdDims &dm = gr.dm;
  const int D = h.D, T = dm.T, hop = c.nfft / 4;
  const int F0 = n.F[0], CA = h.CA, CD = h.CD;
  const int64_t Lp = (L + 1) & ~(int64_t)1;
  // spectrogram + standardisation of both branches (hdemucs.py:680-704); pad1d's zero extension of short inputs (:21-34)
  int64_t el = 0, Lv = L;
  {
    const int64_t left = hop / 2 * 3, right = left + (int64_t)T * hop - L, mx = std::max(left, right);
    if (L <= mx) {
      const int64_t extra = mx - L + 1, er = std::min(right, extra);
      el = extra - er;
      Lv = L + extra;
    }
  }
  CHK(timed(e, ASX_PROF_STFT, 0.0, 4.0 * (double)B * (2.0 * L + 4.0 * T * F0), s, [&]() {
    hipLaunchKernelGGL(ht_stft_kernel, dim3(T, 2, B), dim3(256), stft_lds(n.plan), s, seg, L, el, Lv, hop, T, b.xf0, n.window.f(),
                       reinterpret_cast<const float2 *>(n.tw.p), n.plan);
  }));
  const int64_t nf = (int64_t)T * F0 * 4;
  CHK(ht_stats(e, b.xf0, B, T, (int64_t)F0 * 4, (int64_t)F0 * 4, 4, 4, 1, b.acc_f, s));
  CHK(timed(e, ASX_PROF_MISC, 0.0, 8.0 * (double)B * nf, s, [&]() {
    hipLaunchKernelGGL(std_norm_kernel, dim3((unsigned)((nf + 255) / 256), B), dim3(256), 0, s, b.xf0, nf, b.acc_f);
  }));
  CHK(ht_stats(e, seg, B, 1, 2 * L, 2 * L, 1, 1, 1, b.acc_t, s));
  CHK(timed(e, ASX_PROF_MISC, 0.0, 16.0 * (double)B * L, s, [&]() {
    hipLaunchKernelGGL(hd_time_norm_kernel, dim3((unsigned)((Lp + 255) / 256), B), dim3(256), 0, s, seg, L, Lp, b.acc_t, b.xt0);
  }));
  for (int i = 0; i < D; ++i) CHK(ht_enc_level(e, i, B, s));
  // ---- level A (hdemucs.py:715-733 with HEncLayer.forward :139-170) ----
  HtGeom g;   // tencoder: bare conv, its output is injected into the spectrogram branch
  g.I = (int)n.L[D];
  g.Cin = CD;
  g.ldc = CD;
  g.KI = c.kernel_size;
  g.PI = c.kernel_size / 4;
  g.SI = c.stride;
  g.IR = T;
  CHK(ht_gg(e, h.tencA, b.skt[D - 1], g, B, w.inj, CA, GG_DENSE, 0, nullptr, 0, 0, 0, 0, s));
  HtGeom f;   // all remaining frequency rows -> one
  f.O = T;
  f.I = n.F[D];
  f.Cin = CD;
  f.ldc = CD;
  f.KI = n.F[D];
  f.SI = c.stride;
  f.IR = 1;
  CHK(ht_gg(e, h.encA.conv, b.skf[D - 1], f, (int64_t)B * T, w.ya, CA, GG_DENSE, 0, w.inj, CA, 0, 0, 0, s));
  CHK(hd_group_norm(e, w.ya, B, T, CA, h.n1A, 0, w.ya, nullptr, s));
  return ht_dconv_layer(e, h.encA, 0, 1, w.ya, B, T, 1, true, s);
}

// the phase after the BLSTM of DConv layer d of level A / Z: LocalState, the rest of the layer, then everything up to the
// next BLSTM (or the end of the network)
static int hd_phase_after(asx_engine *e, const HdGroup &gr, bool levelZ, size_t d, hipStream_t s) {
  HdNet &h = *e->hd;
  HtNet &n = *e->ht;
  const asx_hd_config &c = h.cfg;
  auto &b = n.b;
  auto &w = h.b;
  const HdDims &dm = gr.dm;
  const int B = gr.B, D = h.D, S = c.n_sources, T = dm.T, T2 = dm.T2, hop = c.nfft / 4;
  const int64_t L = gr.L;
  const int F0 = n.F[0], CA = h.CA, CZ = h.CZ, CD = h.CD;
  const size_t K = h.encA.dc.size();
  if (!levelZ) {
    const int HA = CA / c.dconv_comp;
    CHK(hd_local_state(e, h.insA[d].attn, b.h, B, T, HA, s));
    CHK(ht_dconv_layer(e, h.encA, d, 2, w.ya, B, T, 1, true, s));
    if (d + 1 < K) return ht_dconv_layer(e, h.encA, d + 1, 1, w.ya, B, T, 1, true, s);
    CHK(ht_linear(e, h.encA.rewrite, w.ya, CA, (int64_t)B * T, w.rwA, 2 * CA, 0, nullptr, 0, s));
    CHK(hd_group_norm(e, w.rwA, B, T, 2 * CA, h.n2A, 1, w.skA, nullptr, s));
    // ---- level Z: time-only layer on the merged branch ----
    HtGeom g;
    g.I = T;
    g.Cin = CA;
    g.ldc = CA;
    g.KI = 2 * c.time_stride;
    g.PI = c.time_stride / 2;
    g.SI = c.time_stride;
    g.IR = T2;
    CHK(ht_gg(e, h.encZ.conv, w.skA, g, B, w.yz, CZ, GG_DENSE, 0, nullptr, 0, 0, 0, 0, s));
    CHK(hd_group_norm(e, w.yz, B, T2, CZ, h.n1Z, 0, w.yz, nullptr, s));
    return ht_dconv_layer(e, h.encZ, 0, 1, w.yz, B, 1, T2, false, s);
  }
  const int HZ = CZ / c.dconv_comp;
  CHK(hd_local_state(e, h.insZ[d].attn, b.h, B, T2, HZ, s));
  CHK(ht_dconv_layer(e, h.encZ, d, 2, w.yz, B, 1, T2, false, s));
  if (d + 1 < K) return ht_dconv_layer(e, h.encZ, d + 1, 1, w.yz, B, 1, T2, false, s);
  CHK(ht_linear(e, h.encZ.rewrite, w.yz, CZ, (int64_t)B * T2, w.rwZ, 2 * CZ, 0, nullptr, 0, s));
  CHK(hd_group_norm(e, w.rwZ, B, T2, 2 * CZ, h.n2Z, 1, w.skZ, nullptr, s));
  // ---- decoder of level Z (x = 0 + skip; hdemucs.py:303-330) ----
  {
    HtGeom g;
    g.I = T2;
    g.Cin = CZ;
    g.ldc = CZ;
    g.KI = 3;
    g.PI = 1;
    g.IR = T2;
    CHK(ht_gg(e, h.decZ_rw, w.skZ, g, B, w.rwZ, 2 * CZ, GG_DENSE, 0, nullptr, 0, 0, 0, 0, s));
    CHK(hd_group_norm(e, w.rwZ, B, T2, 2 * CZ, h.dZn1, 1, w.g, nullptr, s));
    HtGeom t;
    t.I = T2;
    t.Cin = CZ;
    t.ldc = CZ;
    t.KI = 2;
    t.PI = 1;
    t.IR = T2 + 1;
    t.So = c.time_stride;
    t.crop = 0;   // norm2 sees the uncropped output (hdemucs.py:321-327)
    const int full = (T2 + 1) * c.time_stride;
    CHK(ht_gg(e, h.decZ_tr, w.g, t, B, w.tr, CA, GG_CONVT, 0, nullptr, CA, 0, full, CA, s));
    CHK(hd_group_norm(e, w.tr, B, T, CA, h.dZn2, 0, w.dAin, w.skA, s, full, c.time_stride / 2));   // gelu(norm2)[crop] + the skip of level A
  }
  // ---- decoder of level A: spectrogram half and the waveform half fed by `pre` ----
  {
    HtGeom g;
    g.O = T;
    g.I = 1;
    g.Cin = CA;
    g.ldc = CA;
    g.KO = 3;
    g.PO = 1;
    g.IR = 1;
    CHK(ht_gg(e, h.decA_rw, w.dAin, g, (int64_t)B * T, w.rwA, 2 * CA, GG_DENSE, 0, nullptr, 0, 0, 0, 0, s));
    CHK(hd_group_norm(e, w.rwA, B, T, 2 * CA, h.dAn1, 1, w.pre, nullptr, s));
    HtGeom t;   // ConvTranspose (kernel_size x 1) from one row to kernel_size rows, no crop
    t.O = T;
    t.I = 1;
    t.Cin = CA;
    t.ldc = CA;
    t.KI = 2;
    t.PI = 1;
    t.IR = 2;
    t.crop = 0;
    CHK(ht_gg(e, h.decA_tr, w.pre, t, (int64_t)B * T, w.tr, CD, GG_CONVT, 0, nullptr, CD, 0, n.F[D], CD, s));
    CHK(hd_group_norm(e, w.tr, B, (int64_t)T * n.F[D], CD, h.dAn2, 0, b.df[D], b.skf[D - 1], s));
    HtGeom u;   // tdecoder.0 (empty): ConvTranspose1d of pre[:, :, 0]
    u.I = T;
    u.Cin = CA;
    u.ldc = CA;
    u.KI = 2;
    u.PI = 1;
    u.IR = T + 1;
    u.crop = 0;
    const int full = (T + 1) * c.stride;
    CHK(ht_gg(e, h.tdecA_tr, w.pre, u, B, w.tr, CD, GG_CONVT, 0, nullptr, CD, 0, full, CD, s));
    CHK(hd_group_norm(e, w.tr, B, n.L[D], CD, h.tdAn2, 0, b.dt[D], b.skt[D - 1], s, full, c.kernel_size / 4));
  }
  for (int i = D - 1; i >= 0; --i) CHK(ht_dec_level(e, i, B, s));
  // CaC -> iSTFT, + waveform branch (hdemucs.py:760-781)
  const int64_t nf = (int64_t)T * F0 * 4;
  CHK(timed(e, ASX_PROF_ISTFT, 0.0, 4.0 * (double)B * S * 2 * T * (2.0 * F0 + c.nfft), s, [&]() {
    hipLaunchKernelGGL(ht_istft_kernel, dim3(T, S * 2, B), dim3(256), ht_istft_lds(n.plan), s, b.df[0], T, 4 * S, b.acc_f, (double)nf, b.frames,
                       n.window.f(), reinterpret_cast<const float2 *>(n.tw.p), n.plan);
  }));
  return timed(e, ASX_PROF_OLA, 0.0, 4.0 * (double)B * S * 2 * (T * (double)c.nfft + 2.0 * L), s, [&]() {
    hipLaunchKernelGGL(ht_ola_kernel, dim3((unsigned)((L + 255) / 256), S * 2, B), dim3(256), 0, s, b.frames, n.env_hop.f(), c.nfft, hop, T, L,
                       b.dt[0], 2 * S, b.acc_t, gr.out);
  });
}

// all groups, phase by phase (G[0] must be the engine's own nets)
static int hd_forward_groups(asx_engine *e, std::vector<HdGroup> &G, hipStream_t s) {
  HtNet *const ht0 = e->ht;
  HdNet *const hd0 = e->hd;
  int rc = ASX_OK;
  auto each = [&](const std::function<int(const HdGroup &)> &fn) {
    for (HdGroup &g : G) {
      e->ht = g.ht;
      e->hd = g.hd;
      rc = fn(g);
      e->ht = ht0;
      e->hd = hd0;
      if (rc != ASX_OK) return rc;
    }
    return rc;
  };
  for (HdGroup &g : G) {
    REQUIRE(g.L >= 1 && g.B >= 1, "empty segment");
    g.dm = hd_dims(*hd0, g.L);
  }
  CHK(each([&](const HdGroup &g) { return hd_phase_front(e, g, s); }));
  const size_t K = hd0->encA.dc.size();
  for (int lvl = 0; lvl < 2; ++lvl)
    for (size_t d = 0; d < K; ++d) {
      CHK(hd_blstm_joint(e, G, lvl == 1, d, s));
      CHK(each([&](const HdGroup &g) { return hd_phase_after(e, g, lvl == 1, d, s); }));
    }
  return ASX_OK;
}

// HDemucs.forward for B segments of L samples: seg [B, 2, L] -> out [B, S, 2, L]
static int hd_forward_dev(asx_engine *e, const float *seg, int B, int64_t L, float *out, hipStream_t s) {
  std::vector<HdGroup> G{HdGroup{e->ht, e->hd, B, L, seg, out, HdDims{}}};
  return hd_forward_groups(e, G, s);
}

// 2*MAC of the GEMM-shaped work of one segment of L samples
static double hd_flops(const asx_engine *e, int64_t L) {
  const HdNet &h = *e->hd;
  const HtNet &n = *e->ht;
  const asx_hd_config &c = h.cfg;
  const HdDims d = hd_dims(h, L);
  const int D = h.D, S = c.n_sources;
  double mac = 0.0;
  auto dconv = [&](double rows, int C, bool ins, double T) {
    const double hid = (double)C / c.dconv_comp;
    double m = 0.0;
    for (int k = 0; k < c.dconv_depth; ++k) {
      m += rows * (3.0 * C * hid + hid * 2.0 * C);
      if (ins) {
        const double steps = T > 200 ? 200 : T, nfr = T > 200 ? ceil(T / 100.0) : 1;
        m += steps * nfr * (8.0 * hid * hid + 16.0 * hid * hid + 2.0 * hid * hid + 16.0 * hid * hid);   // ih0, ih1, linear, recurrences
        m += rows * (3.0 * hid + 16) * hid + rows * hid * hid + 2.0 * T * T * hid;
      }
    }
    return m;
  };
  int64_t Lw = L;
  for (int i = 0; i < D; ++i) {
    const int C = n.C[i], cz = i == 0 ? 4 : n.C[i - 1], ct = i == 0 ? 2 : n.C[i - 1];
    const int64_t L1 = (Lw + c.stride - 1) / c.stride;
    const double rf = (double)d.T * n.F[i + 1], rt = (double)L1;
    const int oz = i == 0 ? 4 * S : n.C[i - 1], ot = i == 0 ? 2 * S : n.C[i - 1];
    mac += rf * (8.0 * cz * C + 2.0 * C * C + 9.0 * C * 2 * C + 8.0 * C * oz) + dconv(rf, C, false, 0);
    mac += rt * (8.0 * ct * C + 2.0 * C * C + 3.0 * C * 2 * C + 8.0 * C * ot) + dconv(rt, C, false, 0);
    Lw = L1;
  }
  const double T = d.T, T2 = d.T2, CA = h.CA, CZ = h.CZ, CD = h.CD;
  mac += T * (2.0 * c.kernel_size * CD * CA + 2.0 * CA * CA) + dconv(T, h.CA, true, T);
  mac += T2 * (2.0 * c.time_stride * CA * CZ + 2.0 * CZ * CZ) + dconv(T2, h.CZ, true, T2);
  mac += T2 * (3.0 * CZ * 2 * CZ + 2.0 * c.time_stride * CZ * CA);
  mac += T * (3.0 * CA * 2 * CA + 2.0 * c.kernel_size * CA * CD);
  return 2.0 * mac;
}

// ---- apply_model around it (apply.py:195-260) ---------------------------------------------------------------------
struct HdPlan {
  int64_t stride, segment, max_shift;
  std::vector<HtShift> shifts;
  std::vector<int64_t> starts, clen;   // per chunk: song index of its first sample, its length
};

static int hd_plan(const asx_engine *e, int64_t N, int32_t shifts, const int64_t *offsets, double overlap, HdPlan &p) {
  const asx_hd_config &c = e->hd->cfg;
  p.segment = c.segment_samples;
  p.stride = (int64_t)((1.0 - overlap) * (double)p.segment);
  REQUIRE(p.stride >= 1 && p.stride <= p.segment, "overlap %g gives a bad stride", overlap);
  p.max_shift = shifts > 0 ? c.samplerate / 2 : 0;
  p.shifts.clear();
  p.starts.clear();
  p.clen.clear();
  const int nsh = shifts > 0 ? shifts : 1;
  for (int si = 0; si < nsh; ++si) {
    HtShift sh;
    sh.offset = shifts > 0 ? offsets[si] : 0;
    REQUIRE(sh.offset >= 0 && sh.offset <= p.max_shift, "shift offset %lld outside [0, %lld]", (long long)sh.offset, (long long)p.max_shift);
    sh.VL = N + p.max_shift - sh.offset;
    sh.first = (int)p.starts.size();
    for (int64_t off = 0; off < sh.VL; off += p.stride) {
      p.starts.push_back(sh.offset + off - p.max_shift);   // no padding: the chunk itself is the model input
      p.clen.push_back(std::min(sh.VL - off, p.segment));
    }
    sh.nk = (int)p.starts.size() - sh.first;
    p.shifts.push_back(sh);
  }
  return ASX_OK;
}

// chunk forwards [k0, k1) -> chunk_out [k1-k0, S, 2, segment] (each row holds clen valid samples).  Chunks of equal
// length form one group (up to max_batch of them); up to HD_MAX_GROUPS groups advance together and share their BLSTM
// launches (hd_forward_groups).  (Running the tail groups on a second stream instead was measured and gave nothing: the
// two HSA queues never had kernels in flight together)
constexpr int HD_MAX_GROUPS = 6;

static int hd_segments_dev(asx_engine *e, const float *mix_dev, int64_t N, const HdPlan &p, uint32_t flags, int k0, int k1, float *chunk_out,
                           hipStream_t s) {
  HdNet &h = *e->hd;
  HtNet &n = *e->ht;
  const int S = h.cfg.n_sources;
  const int standardize = (flags & ASX_HT_STANDARDIZE) ? 1 : 0;
  if (k1 <= k0) return ASX_OK;
  if (standardize) CHK(ht_ref_stats(e, mix_dev, N, s));
  std::vector<int> order;
  for (int k = k0; k < k1; ++k) order.push_back(k);
  std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return p.clen[a] > p.clen[b]; });
  // 16: the BLSTM recurrences cost the same 200 steps whatever the batch, so the equal-length chunks of a song go together
  // (4-min song, 44-s chunks: 730 -> 789x real time against batches of 4; ~2.2 GB of workspace per 44-s chunk)
  const int maxB = h.cfg.max_batch > 0 ? h.cfg.max_batch : 16;
  const int nk = (int)order.size();
  std::vector<int64_t> st(nk);
  for (int i = 0; i < nk; ++i) st[i] = p.starts[order[i]];
  static const int max_groups = getenv("ASX_HD_GROUPS") ? std::max(1, std::min(HD_MAX_GROUPS, atoi(getenv("ASX_HD_GROUPS")))) : HD_MAX_GROUPS;
  int i = 0;
  while (i < nk) {
    std::vector<HdGroup> G;
    std::vector<int> first;
    while (i < nk && (int)G.size() < max_groups) {
      const int64_t L = p.clen[order[i]];
      int j = i;
      while (j < nk && p.clen[order[j]] == L && j - i < maxB) ++j;
      HtNet *gn = &n;
      HdNet *gh = &h;
      if (!G.empty()) {
        CHK(hd_clone(e, G.size() - 1));
        gn = e->ht_cl[G.size() - 1];
        gh = e->hd_cl[G.size() - 1];
      }
      const int B = j - i;
      CHK(gn->seg.ensure((size_t)B * 2 * L * 4));
      CHK(gh->tmp_out.ensure((size_t)B * S * 2 * L * 4));
      ht_gather_launch(mix_dev, N, st.data() + i, B, L, reinterpret_cast<const double *>(n.ref_acc.p), standardize, gn->seg.f(), s);
      HIPCHK(hipGetLastError());
      G.push_back(HdGroup{gn, gh, B, L, gn->seg.f(), gh->tmp_out.f(), HdDims{}});
      first.push_back(i);
      i = j;
    }
    CHK(hd_forward_groups(e, G, s));
    for (size_t gi = 0; gi < G.size(); ++gi)
      for (int bi = 0; bi < G[gi].B; ++bi)
        HIPCHK(hipMemcpy2DAsync(chunk_out + (size_t)(order[first[gi] + bi] - k0) * S * 2 * p.segment, (size_t)p.segment * 4,
                                G[gi].out + (size_t)bi * S * 2 * G[gi].L, (size_t)G[gi].L * 4, (size_t)G[gi].L * 4, (size_t)S * 2,
                                hipMemcpyDeviceToDevice, s));
  }
  return ASX_OK;
}

static int hd_fold_dev(asx_engine *e, const float *mix_dev, int64_t N, const HdPlan &p, uint32_t flags, const float *chunk_out, float *out_dev,
                       hipStream_t s) {
  HtNet &n = *e->ht;
  const int S = e->hd->cfg.n_sources;
  const int standardize = (flags & ASX_HT_STANDARDIZE) ? 1 : 0;
  const int swap01 = (flags & ASX_HT_SWAP01) ? 1 : 0;
  if (standardize) CHK(ht_ref_stats(e, mix_dev, N, s));
  const int nsh = (int)p.shifts.size();
  for (int si = 0; si < nsh; ++si) {
    const HtShift &sh = p.shifts[si];
    CHK(timed(e, ASX_PROF_FINALIZE, 0.0, 4.0 * ((double)sh.nk * S * 2 * p.segment + 2.0 * S * 2 * N), s, [&]() {
      hipLaunchKernelGGL(ht_fold_kernel, dim3((unsigned)((N + 255) / 256), S * 2), dim3(256), 0, s, chunk_out + (size_t)sh.first * S * 2 * p.segment,
                         sh.nk, S * 2, p.segment, p.stride, p.segment, sh.VL, p.max_shift - sh.offset, n.fold_w.f(), si == 0 ? 1 : 0,
                         si == nsh - 1 ? 1 : 0, nsh, reinterpret_cast<const double *>(n.ref_acc.p), standardize, swap01, 0, N, out_dev);
    }));
  }
  return ASX_OK;
}

static int hd_demix_dev(asx_engine *e, const float *mix_dev, int64_t N, int32_t shifts, const int64_t *offsets, double overlap, uint32_t flags,
                        float *out_dev, hipStream_t s) {
  HtNet &n = *e->ht;
  HdPlan p;
  CHK(hd_plan(e, N, shifts, offsets, overlap, p));
  const int nseg = (int)p.starts.size();
  CHK(n.chunk_out.ensure((size_t)nseg * e->hd->cfg.n_sources * 2 * p.segment * 4));
  CHK(hd_segments_dev(e, mix_dev, N, p, flags, 0, nseg, n.chunk_out.f(), s));
  return hd_fold_dev(e, mix_dev, N, p, flags, n.chunk_out.f(), out_dev, s);
}
