// Demucs v4 (HTDemucs) on the engine: uvr_lib_v5/demucs/htdemucs.py:483-620 (forward), apply.py:124-260
// (apply_model: shifts + split with triangular fold) and DemucsSeparator.demix_demucs
// (architectures/demucs_separator.py:162-194).  Included by asx.hip (one TU).
//
// Layout: spectrogram branch [B, T, F_i, C_i], waveform branch [B, 1, L_i, C_i], both channels-last; see
// kernels_ht.h.  Every conv / linear is one gg_kernel or tdf_dma_kernel launch:
//   encoder i : conv k8/s4 + GELU | DConv x depth { k3 dilated -> GroupNorm+GELU -> 1x1 -> GroupNorm+GLU+LayerScale+res }
//               | rewrite 1x1 + GLU (+ frequency embedding on layer 0)
//   transformer: norm_in + sinusoidal table | {self | cross} layers with LayerScale folded into out_proj / linear2
//   decoder i : (x + skip fused into the producer) rewrite 3x3 / k3 + GLU | ConvTranspose k8/s4 as a 2-tap GEMM with
//               a 4-position scatter epilogue (+ GELU + next skip)
#pragma once

struct HtGemm {
  DevBuf w, b;
  int n = 0, k = 0;
  DevBuf wh;                    // halo-tile packing of a k3 / 3x3 conv (kernels_halo.h); empty: not eligible
  int wh_nt = 0, wh_taps = 0;   // its N tile and tap count
  int glu_c = 0;                // != 0: rows are in GLU order (ht_glu_perm) for glu_c output channels
};
struct HtDconv {
  HtGemm c1, c2;
  DevBuf g1w, g1b, g2w, g2b, ls;
  int hid = 0, hp = 0;
};
struct HtEnc {
  HtGemm conv, rewrite;
  std::vector<HtDconv> dc;
  int cin = 0, cout = 0;
};
struct HtDec {
  HtGemm rewrite, convtr;
  int cin = 0, cout = 0;   // cin = channels entering the layer (= encoder cout), cout = channels leaving it
};
struct HtTLayer {
  bool cross = false;
  DevBuf n1w, n1b, n2w, n2b, n3w, n3b, now, nob;
  HtGemm q, kv, out, l1, l2;   // self: q holds the packed in_proj (N = 3C)
};

struct HtNet {
  asx_ht_config cfg{};
  bool begun = false, ready = false;
  FftPlan plan{};
  DevBuf window, tw, env_hop, fold_w;
  int T = 0, Ct = 0, hidden = 0;
  std::vector<int> F, C;        // F[0..D], C[0..D-1] (C[i] = channels of encoder i's output)
  std::vector<int64_t> L;       // L[0..D]
  std::vector<HtEnc> enc, tenc;
  std::vector<HtDec> dec, tdec; // indexed by encoder level i (decoder of level i undoes encoder i)
  DevBuf femb;
  HtGemm up, up_t, down, down_t;
  DevBuf nin_w, nin_b, nint_w, nint_b, pos_f, pos_t;
  std::vector<HtTLayer> lay, lay_t;
  // workspace
  int ws_batch = 0;
  DevBuf ws, acc, chunk_out, d_starts, seg, ref, ref_acc;
  struct {
    float *xf0, *xt0, *yf, *yt, *h, *z, *rw, *tokf, *tokt, *xn, *xn2, *qkv, *kvb, *attf, *attt, *ffh, *frames;
    std::vector<float *> skf, skt, df, dt;
    double *acc_f, *acc_t, *acc_g, *acc_g2;
    float *rowstat, *mr_g;
    unsigned *ticket;   // rowstat_reduce_kernel's arrival counters (zero between launches)
  } b;
};

static void ht_free(HtNet &n) {
  auto fg = [](HtGemm &g) {
    g.w.release();
    g.b.release();
    g.wh.release();
  };
  for (auto *v : {&n.enc, &n.tenc})
    for (auto &e : *v) {
      fg(e.conv);
      fg(e.rewrite);
      for (auto &d : e.dc) {
        fg(d.c1);
        fg(d.c2);
        for (DevBuf *p : {&d.g1w, &d.g1b, &d.g2w, &d.g2b, &d.ls}) p->release();
      }
    }
  for (auto *v : {&n.dec, &n.tdec})
    for (auto &d : *v) {
      fg(d.rewrite);
      fg(d.convtr);
    }
  for (auto *v : {&n.lay, &n.lay_t})
    for (auto &l : *v) {
      for (DevBuf *p : {&l.n1w, &l.n1b, &l.n2w, &l.n2b, &l.n3w, &l.n3b, &l.now, &l.nob}) p->release();
      for (HtGemm *g : {&l.q, &l.kv, &l.out, &l.l1, &l.l2}) fg(*g);
    }
  for (HtGemm *g : {&n.up, &n.up_t, &n.down, &n.down_t}) fg(*g);
  for (DevBuf *p : {&n.window, &n.tw, &n.env_hop, &n.fold_w, &n.femb, &n.nin_w, &n.nin_b, &n.nint_w, &n.nint_b, &n.pos_f,
                    &n.pos_t, &n.ws, &n.acc, &n.chunk_out, &n.d_starts, &n.seg, &n.ref, &n.ref_acc})
    p->release();
  n.enc.clear();
  n.tenc.clear();
  n.dec.clear();
  n.tdec.clear();
  n.lay.clear();
  n.lay_t.clear();
  n.ready = false;
  n.ws_batch = 0;
}
static void ht_destroy(HtNet *n) {
  ht_free(*n);
  delete n;
}

static int ht_up(DevBuf &d, const std::vector<float> &h) {
  CHK(d.ensure(h.size() * 4));
  HIPCHK(hipMemcpy(d.p, h.data(), h.size() * 4, hipMemcpyHostToDevice));
  return ASX_OK;
}
static int ht_up_named(asx_engine *e, DevBuf &d, const std::string &name, int64_t numel) {
  const float *p;
  CHK(get_tensor(e, name, numel, &p));
  CHK(d.ensure((size_t)numel * 4));
  HIPCHK(hipMemcpy(d.p, p, (size_t)numel * 4, hipMemcpyHostToDevice));
  return ASX_OK;
}

// second packing of a stride-1 k3 / 3x3 conv for the halo-tile kernel (pw = [N, taps*cinp] as uploaded to g.w)
static int ht_halo_pack(HtGemm &g, const std::vector<float> &pw, int taps, int cinp) {
  static const bool off = getenv("ASX_HALO") != nullptr && atoi(getenv("ASX_HALO")) == 0;
  if (off || (taps != 3 && taps != 9) || (cinp % HG_KC) != 0 || (g.n & 3)) return ASX_OK;
  std::vector<float> ph;
  g.wh_nt = hg_tile_n(g.n);
  g.wh_taps = taps;
  hg_pack_weights(pw, g.n, taps, cinp, g.wh_nt, ph);
  return ht_up(g.wh, ph);
}

// GLU row order: blocks of 32 rows = [a_c .. a_c+15 | g_c .. g_c+15] (c a multiple of 16), i.e. one MFMA fragment of 16 value
// rows followed by the fragment of their 16 gates.  A lane of the GEMM epilogue then holds a[c + 4 lk .. + 3] in fragment 2 j and
// the matching gates in fragment 2 j + 1: FOUR consecutive GLU outputs, written (and, in the DConv epilogue, read-modified) as one
// 16-byte access.  (Rounds 1-2 interleaved (a, a, g, g) per four rows: two outputs per lane, 8-byte accesses -- the memory-bound
// 1x1 / DConv launches ran at 1.4-2.6 TB/s.)  The channel count is padded to a multiple of 16 (zero rows): N = 32 ceil(c / 16).
static inline int ht_glu_rows(int c) { return 32 * ((c + 15) / 16); }
static void ht_glu_perm(std::vector<float> &w, std::vector<float> &b, int c, int k) {
  const int np = ht_glu_rows(c);
  std::vector<float> w2((size_t)np * k, 0.f), b2((size_t)np, 0.f);
  for (int ch = 0; ch < c; ++ch)
    for (int half = 0; half < 2; ++half) {
      const int src = half * c + ch;
      const int dst = 32 * (ch >> 4) + 16 * half + (ch & 15);
      std::copy(w.begin() + (size_t)src * k, w.begin() + (size_t)(src + 1) * k, w2.begin() + (size_t)dst * k);
      b2[dst] = b[src];
    }
  w.swap(w2);
  b.swap(b2);
}

// torch conv weight [Cout, Cin, KA, KB] (KB = 1 for 1-D) -> [Npad, (to*KI + ti)*CinP + ci];
// inner_first: the first kernel axis (KA) is the inner (frequency) tap, the second the outer (time) tap
static int ht_pack_conv(asx_engine *e, HtGemm &g, const std::string &name, int cout, int cin, int ka, int kb,
                        bool glu, int npad = 0, int cinp = 0, bool halo = false) {
  const float *w, *b;
  CHK(get_tensor(e, name + ".weight", (int64_t)cout * cin * ka * kb, &w));
  CHK(get_tensor(e, name + ".bias", cout, &b));
  if (!npad) npad = cout;
  if (!cinp) cinp = cin;
  const int K = ka * kb * cinp;
  std::vector<float> pw((size_t)npad * K, 0.f), pb((size_t)npad, 0.f);
  for (int n = 0; n < cout; ++n) {
    pb[n] = b[n];
    for (int ci = 0; ci < cin; ++ci)
      for (int a = 0; a < ka; ++a)
        for (int c = 0; c < kb; ++c)
          pw[(size_t)n * K + ((size_t)c * ka + a) * cinp + ci] = w[(((size_t)n * cin + ci) * ka + a) * kb + c];
  }
  if (glu) {
    pw.resize((size_t)cout * K);      // (npad == cout for every GLU layer)
    pb.resize((size_t)cout);
    ht_glu_perm(pw, pb, cout / 2, K);
    npad = ht_glu_rows(cout / 2);
    g.glu_c = cout / 2;
  }
  g.n = npad;
  g.k = K;
  CHK(ht_up(g.w, pw));
  CHK(ht_up(g.b, pb));
  if (halo) CHK(ht_halo_pack(g, pw, ka * kb, cinp));
  return ASX_OK;
}

// ConvTranspose weight [Cin, Cout, 8(,1)], stride 4: W[(r, co)][tap*Cin + ci] = w[ci][co][r + 4*(1 - tap)]
static int ht_pack_convtr(asx_engine *e, HtGemm &g, const std::string &name, int cin, int cout, int ksz, int stride) {
  const float *w, *b;
  CHK(get_tensor(e, name + ".weight", (int64_t)cin * cout * ksz, &w));
  CHK(get_tensor(e, name + ".bias", cout, &b));
  const int N = stride * cout, K = 2 * cin;
  std::vector<float> pw((size_t)N * K), pb((size_t)N);
  for (int r = 0; r < stride; ++r)
    for (int co = 0; co < cout; ++co) {
      pb[(size_t)r * cout + co] = b[co];
      for (int tap = 0; tap < 2; ++tap)
        for (int ci = 0; ci < cin; ++ci)
          pw[((size_t)r * cout + co) * K + (size_t)tap * cin + ci] = w[((size_t)ci * cout + co) * ksz + r + stride * (1 - tap)];
    }
  g.n = N;
  g.k = K;
  CHK(ht_up(g.w, pw));
  CHK(ht_up(g.b, pb));
  return ASX_OK;
}

// nn.Linear rows [r0, r1) of `name` (+ per-output-row scale: LayerScale folded in, transformer.py:258,263)
static int ht_pack_linear(asx_engine *e, HtGemm &g, const std::string &wname, const std::string &bname, int rows_total,
                          int r0, int r1, int k, const float *scale) {
  const float *w, *b;
  CHK(get_tensor(e, wname, (int64_t)rows_total * k, &w));
  CHK(get_tensor(e, bname, rows_total, &b));
  const int n = r1 - r0;
  std::vector<float> pw((size_t)n * k), pb((size_t)n);
  for (int i = 0; i < n; ++i) {
    const float sc = scale ? scale[i] : 1.0f;
    pb[i] = b[r0 + i] * sc;
    for (int j = 0; j < k; ++j) pw[(size_t)i * k + j] = w[(size_t)(r0 + i) * k + j] * sc;
  }
  g.n = n;
  g.k = k;
  CHK(ht_up(g.w, pw));
  CHK(ht_up(g.b, pb));
  return ASX_OK;
}

// ins: modules inserted between the GELU and the 1x1 conv (Demucs v3's BLSTM / LocalState shift the Sequential's indices)
static int ht_load_dconv(asx_engine *e, HtDconv &d, const std::string &p, int ch, int comp, int idx, int ins = 0) {
  d.hid = ch / comp;
  d.hp = (d.hid + 3) & ~3;
  const std::string q = p + ".layers." + std::to_string(idx);
  CHK(ht_pack_conv(e, d.c1, q + ".0", d.hid, ch, 3, 1, false, d.hp, 0));
  CHK(ht_pack_conv(e, d.c2, q + "." + std::to_string(3 + ins), 2 * ch, d.hid, 1, 1, true, 0, d.hp));   // GLU row order for the fused epilogue
  const float *g, *b;
  CHK(get_tensor(e, q + ".1.weight", d.hid, &g));
  CHK(get_tensor(e, q + ".1.bias", d.hid, &b));
  std::vector<float> gw(d.hp, 0.f), gb(d.hp, 0.f);
  std::copy(g, g + d.hid, gw.begin());
  std::copy(b, b + d.hid, gb.begin());
  CHK(ht_up(d.g1w, gw));
  CHK(ht_up(d.g1b, gb));
  {
    const float *g2, *b2;
    CHK(get_tensor(e, q + "." + std::to_string(4 + ins) + ".weight", 2 * ch, &g2));
    CHK(get_tensor(e, q + "." + std::to_string(4 + ins) + ".bias", 2 * ch, &b2));
    std::vector<float> pg(g2, g2 + 2 * ch), pb2(b2, b2 + 2 * ch);
    ht_glu_perm(pg, pb2, ch, 1);     // (weights, "bias") = (gamma, beta), one float per row
    CHK(ht_up(d.g2w, pg));
    CHK(ht_up(d.g2b, pb2));
  }
  CHK(ht_up_named(e, d.ls, q + "." + std::to_string(6 + ins) + ".scale", ch));
  return ASX_OK;
}

static int ht_load_tlayer(asx_engine *e, HtTLayer &L, const std::string &p, bool cross, int C, int hidden) {
  L.cross = cross;
  const std::string at = p + (cross ? ".cross_attn" : ".self_attn");
  const float *g1, *g2;
  CHK(get_tensor(e, p + ".gamma_1.scale", C, &g1));
  CHK(get_tensor(e, p + ".gamma_2.scale", C, &g2));
  if (cross) {
    CHK(ht_pack_linear(e, L.q, at + ".in_proj_weight", at + ".in_proj_bias", 3 * C, 0, C, C, nullptr));
    CHK(ht_pack_linear(e, L.kv, at + ".in_proj_weight", at + ".in_proj_bias", 3 * C, C, 3 * C, C, nullptr));
  } else {
    CHK(ht_pack_linear(e, L.q, at + ".in_proj_weight", at + ".in_proj_bias", 3 * C, 0, 3 * C, C, nullptr));
  }
  CHK(ht_pack_linear(e, L.out, at + ".out_proj.weight", at + ".out_proj.bias", C, 0, C, C, g1));
  CHK(ht_pack_linear(e, L.l1, p + ".linear1.weight", p + ".linear1.bias", hidden, 0, hidden, C, nullptr));
  CHK(ht_pack_linear(e, L.l2, p + ".linear2.weight", p + ".linear2.bias", C, 0, C, hidden, g2));
  CHK(ht_up_named(e, L.n1w, p + ".norm1.weight", C));
  CHK(ht_up_named(e, L.n1b, p + ".norm1.bias", C));
  CHK(ht_up_named(e, L.n2w, p + ".norm2.weight", C));
  CHK(ht_up_named(e, L.n2b, p + ".norm2.bias", C));
  if (cross) {
    CHK(ht_up_named(e, L.n3w, p + ".norm3.weight", C));
    CHK(ht_up_named(e, L.n3b, p + ".norm3.bias", C));
  }
  CHK(ht_up_named(e, L.now, p + ".norm_out.weight", C));
  CHK(ht_up_named(e, L.nob, p + ".norm_out.bias", C));
  return ASX_OK;
}

// sinusoidal tables in float32 with torch's operation order (transformer.py:18-46); the host may override them
// with tables it computed itself (tensors "pos_emb_freq" [T*Fr, C] and "pos_emb_time" [L, C])
static void ht_sin_1d(int length, int dim, std::vector<float> &t) {
  const int half = dim / 2;
  t.assign((size_t)length * dim, 0.f);
  for (int p = 0; p < length; ++p)
    for (int i = 0; i < half; ++i) {
      const float ex = (float)i / (float)(half - 1);
      const float den = powf(10000.0f, ex);
      const float ph = (float)p / den;
      t[(size_t)p * dim + i] = cosf(ph);
      t[(size_t)p * dim + half + i] = sinf(ph);
    }
}
static void ht_sin_2d(int C, int Fr, int T1, std::vector<float> &t) {   // rows (t1, fr)
  const int dm = C / 2;
  t.assign((size_t)T1 * Fr * C, 0.f);
  const float k = -(logf(10000.0f) / (float)dm);
  for (int i = 0; i < dm; i += 2) {
    const float div = expf((float)i * k);
    for (int t1 = 0; t1 < T1; ++t1)
      for (int fr = 0; fr < Fr; ++fr) {
        float *row = &t[((size_t)t1 * Fr + fr) * C];
        row[i] = sinf((float)t1 * div);
        row[i + 1] = cosf((float)t1 * div);
        row[dm + i] = sinf((float)fr * div);
        row[dm + i + 1] = cosf((float)fr * div);
      }
  }
}

// STFT tables and the triangular transition window of apply_model (apply.py:226-231) for segments of TL samples
static int ht_commit_tables(asx_engine *e, int64_t TL) {
  HtNet &n = *e->ht;
  const asx_ht_config &c = n.cfg;
  const int hop = c.nfft / 4;
  // tables
  {
    std::vector<float> w;
    host_window(c.nfft, w);
    CHK(ht_up(n.window, w));
    std::vector<float> tw((size_t)c.nfft * 2);
    for (int j = 0; j < c.nfft; ++j) {
      const double ang = -2.0 * M_PI * (double)j / (double)c.nfft;
      tw[2 * j] = (float)cos(ang);
      tw[2 * j + 1] = (float)sin(ang);
    }
    CHK(ht_up(n.tw, tw));
    std::vector<float> env((size_t)hop, 0.f);
    for (int r = 0; r < hop; ++r)
      for (int i = c.nfft / hop - 1; i >= 0; --i) env[r] += w[r + i * hop] * w[r + i * hop];   // frame order of torch's fold
    CHK(ht_up(n.env_hop, env));
    // triangular transition window (apply.py:226-231), float32 like torch
    std::vector<float> fw((size_t)TL);
    const int64_t h1 = TL / 2, h2 = TL - TL / 2;
    const float mx = (float)std::max(h1, h2);
    for (int64_t i = 0; i < h1; ++i) fw[i] = (float)(i + 1) / mx;
    for (int64_t i = 0; i < h2; ++i) fw[h1 + i] = (float)(h2 - i) / mx;
    CHK(ht_up(n.fold_w, fw));
  }
  return ASX_OK;
}

// weights of level i; the decoders are stored under `decoder.<dec_idx>` / `tdecoder.<tdec_idx>`
static int ht_commit_level(asx_engine *e, int i, int dec_idx, int tdec_idx) {
  HtNet &n = *e->ht;
  const asx_ht_config &c = n.cfg;
  const int S = c.n_sources, AC = 2;
  {
    const int cin_z = i == 0 ? 2 * AC : n.C[i - 1], cin_t = i == 0 ? AC : n.C[i - 1], co = n.C[i];
    const std::string si = std::to_string(i), sj = std::to_string(dec_idx), sjt = std::to_string(tdec_idx);
    HtEnc &E = n.enc[i], &Et = n.tenc[i];
    E.cin = cin_z;
    E.cout = co;
    Et.cin = cin_t;
    Et.cout = co;
    CHK(ht_pack_conv(e, E.conv, "encoder." + si + ".conv", co, cin_z, c.kernel_size, 1, false));
    CHK(ht_pack_conv(e, Et.conv, "tencoder." + si + ".conv", co, cin_t, c.kernel_size, 1, false));
    CHK(ht_pack_conv(e, E.rewrite, "encoder." + si + ".rewrite", 2 * co, co, 1, 1, true));
    CHK(ht_pack_conv(e, Et.rewrite, "tencoder." + si + ".rewrite", 2 * co, co, 1, 1, true));
    E.dc.assign(c.dconv_depth, HtDconv());
    Et.dc.assign(c.dconv_depth, HtDconv());
    for (int d = 0; d < c.dconv_depth; ++d) {
      CHK(ht_load_dconv(e, E.dc[d], "encoder." + si + ".dconv", co, c.dconv_comp, d));
      CHK(ht_load_dconv(e, Et.dc[d], "tencoder." + si + ".dconv", co, c.dconv_comp, d));
    }
    const int out_z = i == 0 ? 2 * AC * S : n.C[i - 1], out_t = i == 0 ? AC * S : n.C[i - 1];
    HtDec &Dz = n.dec[i], &Dt = n.tdec[i];
    Dz.cin = co;
    Dz.cout = out_z;
    Dt.cin = co;
    Dt.cout = out_t;
    REQUIRE(out_z % 4 == 0 && out_t % 4 == 0, "decoder output channels must be multiples of 4");
    CHK(ht_pack_conv(e, Dz.rewrite, "decoder." + sj + ".rewrite", 2 * co, co, 3, 3, true, 0, 0, true));
    CHK(ht_pack_conv(e, Dt.rewrite, "tdecoder." + sjt + ".rewrite", 2 * co, co, 3, 1, true, 0, 0, true));
    CHK(ht_pack_convtr(e, Dz.convtr, "decoder." + sj + ".conv_tr", co, out_z, c.kernel_size, c.stride));
    CHK(ht_pack_convtr(e, Dt.convtr, "tdecoder." + sjt + ".conv_tr", co, out_t, c.kernel_size, c.stride));
  }
  return ASX_OK;
}

static int ht_commit(asx_engine *e) {
  HtNet &n = *e->ht;
  const asx_ht_config &c = n.cfg;
  const int D = c.depth;
  REQUIRE(make_plan(c.nfft, &n.plan), "nfft/2 = %d must factor into {2,3,5}", c.nfft / 2);
  const int hop = c.nfft / 4;
  const int64_t TL = c.segment_samples;
  REQUIRE(TL % 2 == 0 && TL > c.nfft, "segment length %lld must be even and exceed nfft", (long long)TL);
  n.T = (int)((TL + hop - 1) / hop);
  n.F.assign(D + 1, 0);
  n.L.assign(D + 1, 0);
  n.C.assign(D, 0);
  n.F[0] = c.nfft / 2;
  n.L[0] = TL;
  int ch = c.channels;
  for (int i = 0; i < D; ++i) {
    REQUIRE(n.F[i] > c.kernel_size && n.F[i] % c.stride == 0,
            "layer %d: %d frequency rows -- only the all-frequency-layer structure (nfft/2 / stride^depth > 1) is built", i,
            n.F[i]);
    n.F[i + 1] = n.F[i] / c.stride;
    n.L[i + 1] = (n.L[i] + c.stride - 1) / c.stride;
    n.C[i] = ch;
    REQUIRE(ch % 4 == 0, "channel counts must be multiples of 4 (layer %d has %d)", i, ch);
    ch *= c.growth;
  }
  CHK(ht_commit_tables(e, TL));
  n.enc.assign(D, HtEnc());
  n.tenc.assign(D, HtEnc());
  n.dec.assign(D, HtDec());
  n.tdec.assign(D, HtDec());
  for (int i = 0; i < D; ++i) CHK(ht_commit_level(e, i, D - 1 - i, D - 1 - i));
  if (c.freq_emb_scale != 0.f) {
    const float *w;
    CHK(get_tensor(e, "freq_emb.embedding.weight", (int64_t)n.F[1] * n.C[0], &w));
    std::vector<float> fe((size_t)n.F[1] * n.C[0]);
    for (size_t i = 0; i < fe.size(); ++i) fe[i] = c.freq_emb_scale * (w[i] * 10.0f);   // ScaledEmbedding.scale = 10
    CHK(ht_up(n.femb, fe));
  }
  const int Cb = n.C[D - 1];
  n.Ct = c.bottom_channels > 0 ? c.bottom_channels : Cb;
  n.hidden = c.t_hidden;
  if (c.t_layers > 0) {
    const int Ct = n.Ct;
    REQUIRE(Ct % c.t_heads == 0 && (Ct / c.t_heads == 48 || Ct / c.t_heads == 64),
            "transformer head dim %d: only 48 and 64 are built", Ct / c.t_heads);
    if (c.bottom_channels > 0) {
      CHK(ht_pack_conv(e, n.up, "channel_upsampler", Ct, Cb, 1, 1, false));
      CHK(ht_pack_conv(e, n.up_t, "channel_upsampler_t", Ct, Cb, 1, 1, false));
      CHK(ht_pack_conv(e, n.down, "channel_downsampler", Cb, Ct, 1, 1, false));
      CHK(ht_pack_conv(e, n.down_t, "channel_downsampler_t", Cb, Ct, 1, 1, false));
    }
    CHK(ht_up_named(e, n.nin_w, "crosstransformer.norm_in.weight", Ct));
    CHK(ht_up_named(e, n.nin_b, "crosstransformer.norm_in.bias", Ct));
    CHK(ht_up_named(e, n.nint_w, "crosstransformer.norm_in_t.weight", Ct));
    CHK(ht_up_named(e, n.nint_b, "crosstransformer.norm_in_t.bias", Ct));
    const int Fr = n.F[D], T1 = n.T;
    const int64_t T2 = n.L[D];
    const float *pf = nullptr, *pt = nullptr;
    CHK(get_tensor(e, "pos_emb_freq", (int64_t)T1 * Fr * Ct, &pf, true));
    CHK(get_tensor(e, "pos_emb_time", T2 * Ct, &pt, true));
    std::vector<float> tf, tt;
    if (pf) tf.assign(pf, pf + (size_t)T1 * Fr * Ct);
    else ht_sin_2d(Ct, Fr, T1, tf);
    if (pt) tt.assign(pt, pt + (size_t)T2 * Ct);
    else ht_sin_1d((int)T2, Ct, tt);
    CHK(ht_up(n.pos_f, tf));
    CHK(ht_up(n.pos_t, tt));
    n.lay.assign(c.t_layers, HtTLayer());
    n.lay_t.assign(c.t_layers, HtTLayer());
    for (int i = 0; i < c.t_layers; ++i) {
      CHK(ht_load_tlayer(e, n.lay[i], "crosstransformer.layers." + std::to_string(i), i % 2 == 1, Ct, n.hidden));
      CHK(ht_load_tlayer(e, n.lay_t[i], "crosstransformer.layers_t." + std::to_string(i), i % 2 == 1, Ct, n.hidden));
    }
  }
  (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&ht_stft_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)stft_lds(n.plan));
  (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&ht_istft_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)ht_istft_lds(n.plan));
  n.ready = true;
  return ASX_OK;
}

// ---- launch helpers ---------------------------------------------------------------------------------------------
struct HtGeom {
  int O = 1, I = 1, Cin = 0, ldc = 0;
  int KO = 1, KI = 1, DO = 1, DI = 1, PO = 0, PI = 0, SI = 1;
  int IR = 1;
  int SO = 1, OR = 0;            // OR = 0: same as O
  int64_t x_bs = 0, y_bs = 0;    // 0: dense
  int crop = 2, So = 4;          // GG_CONVT scatter: position So*j - crop + r
};

// GroupNorm fusion of the DConv GEMMs (kernels_ht.h: GgArgs::stat_acc ...)
struct HtFuse {
  float2 *row_stat = nullptr;
  const float2 *stat_in = nullptr;
  int64_t g_outer = 1;
  int g_mod = 1;
  const float *gamma = nullptr, *beta = nullptr, *ls = nullptr;
};

// y = epilogue(gather(x) @ W^T + b)
static int ht_gg(asx_engine *e, const HtGemm &g, const float *x, const HtGeom &q, int64_t rows_outer, float *y, int64_t ldy,
                 int mode, int act, const float *res, int64_t ldr, int res_mod, int Iout, int Cout, hipStream_t s,
                 const HtFuse *fz = nullptr) {
  GgArgs a{};
  a.x = x;
  a.w = g.w.f();
  a.bias = g.b.f();
  a.res = res;
  a.zeros = e->d_zeros.f();
  a.y = y;
  a.M = rows_outer * q.IR;   // rows_outer = B * (output rows on the outer axis)
  a.N = g.n;
  a.K = g.k;
  a.O = q.O;
  a.I = q.I;
  a.Cin = q.Cin;
  a.ldc = q.ldc;
  a.KO = q.KO;
  a.KI = q.KI;
  a.DO = q.DO;
  a.DI = q.DI;
  a.PO = q.PO;
  a.PI = q.PI;
  a.SI = q.SI;
  a.IR = q.IR;
  a.SO = q.SO;
  a.OR = q.OR ? q.OR : q.O;
  a.x_bs = q.x_bs ? q.x_bs : (int64_t)q.O * q.I * q.ldc;
  a.y_bs = q.y_bs ? q.y_bs : (int64_t)a.OR * q.IR * ldy;
  a.inv_cin = 1.0f / (float)q.Cin;
  a.mode = mode;
  a.act = act;
  a.Iout = Iout;
  a.Cout = (mode == GG_GLU || mode == GG_GNGLU) ? g.glu_c : Cout;
  a.glu_rows = g.glu_c;
  if ((mode == GG_GLU || mode == GG_GNGLU) && (g.glu_c == 0 || (g.glu_c & 3) || (ldy & 3) || (res && (ldr & 3)))) {
    set_err("ht_gg: GLU epilogue on a GEMM without GLU row order, or channels / row strides not multiples of 4");
    return ASX_ERR_INVALID;
  }
  a.crop = q.crop;
  a.So = q.So;
  a.ldy = ldy;
  a.ldr = ldr;
  a.res_mod = res_mod;
  if (fz) {
    a.row_stat = fz->row_stat;
    a.stat_in = fz->stat_in;
    a.g_outer = fz->g_outer;
    a.g_mod = fz->g_mod;
    a.gamma = fz->gamma;
    a.beta = fz->beta;
    a.ls = fz->ls;
  } else {
    a.g_outer = 1;
    a.g_mod = 1;
  }
  if (g.k != q.KO * q.KI * q.Cin || (q.Cin & 3) || (q.ldc & 3) || (g.n & 3) || (ldy & 1) || a.M <= 0) {
    set_err("ht_gg: bad geometry (K=%d taps=%dx%d Cin=%d ldc=%d N=%d)", g.k, q.KO, q.KI, q.Cin, q.ldc, g.n);
    return ASX_ERR_INVALID;
  }
  const double flops = 2.0 * (double)a.M * g.n * g.k;
  const double out_elems = mode == GG_STATS ? 0.0 : (mode == GG_GNGLU ? (double)a.M * g.n : (double)a.M * g.n / (mode == GG_GLU ? 2 : 1));
  const double bytes = 4.0 * ((double)rows_outer * q.I * q.Cin + out_elems + (double)g.n * g.k + (res ? out_elems : 0.0));
  const int cls = mode == GG_CONVT ? ASX_PROF_UP : (q.SI > 1 ? ASX_PROF_DOWN : ASX_PROF_CONV3X3);
  // 64-row instead of 128-row tiles (three workgroups per CU instead of two, twice the weight traffic per flop) for launches that
  // are HBM-bound (algorithmic flop / byte under ASX_GG_LOWAI) or whose 128-row grid would not fill the chip twice
  // (ASX_GG_SMALLGRID workgroups: the inner Demucs levels).  Measured (profiles/NOTES.md): threshold 0 / 40 / 90 / 200 / inf ->
  // htdemucs 956 / 965 / 975 / 974 / 976x, hdemucs_mmi 1120 / 1130 / 1148 / 1163 / 1181x, VR 522 / 520 / 521 / 517 / 516x.
  static const double lowai_thr = getenv("ASX_GG_LOWAI") ? atof(getenv("ASX_GG_LOWAI")) : 90.0;
  static const int64_t small_grid = getenv("ASX_GG_SMALLGRID") ? atoll(getenv("ASX_GG_SMALLGRID")) : 1024;
  {
    const int tn = gg_tile_n(a.N, g.glu_c != 0);
    const int64_t blocks128 = ((a.M + 127) / 128) * ((a.N + tn - 1) / tn);
    a.lowai = ((bytes > 0.0 && flops / bytes < lowai_thr) || blocks128 < small_grid) ? 1 : 0;
  }
  // stride-1 dense convs with 32-aligned channel counts: the bf16 x 6 row GEMM in GATHER mode (kernels_gemm3.h) -- an implicit GEMM
  // whose A rows are the pixels of the channels-last image, one (tap, 32-channel chunk) per stage.  ASX_GATHER6=0: A/B.
  {
    static const bool gather6 = !(getenv("ASX_GATHER6") && atoi(getenv("ASX_GATHER6")) == 0);
    auto a16 = [](const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
    static const int gather_minn = getenv("ASX_GATHER6_MINN") ? atoi(getenv("ASX_GATHER6_MINN")) : 48;   // narrowest dense layer routed here
    static const bool gather_glu = !(getenv("ASX_GATHER6_GLU") && atoi(getenv("ASX_GATHER6_GLU")) == 0);
    const bool glu = mode == GG_GLU && gather_glu && g.glu_c > 0 && g.n % 32 == 0;
    static const bool gather_strided = !(getenv("ASX_GATHER6_STRIDED") && atoi(getenv("ASX_GATHER6_STRIDED")) == 0);
    const bool unit = q.SI == 1 && q.SO == 1 && q.IR == q.I && a.OR == q.O;
    // channel counts off the 32-grid (48, 80, ...): the last chunk of every tap is zero padded -- worth it from 70 % chunk use
    // (ASX_GATHER6_PARTIAL=0: only multiples of 32)
    static const bool gather_partial = !(getenv("ASX_GATHER6_PARTIAL") && atoi(getenv("ASX_GATHER6_PARTIAL")) == 0);
    const bool cin_ok = q.Cin % 32 == 0 || (gather_partial && q.Cin % 8 == 0 && q.Cin > 32 && 10 * q.Cin >= 7 * 32 * ((q.Cin + 31) / 32));
    if (gather6 && e->gemm_bf16x6 > 0 && (mode == GG_DENSE || glu) && res == nullptr && fz == nullptr && (unit || gather_strided) &&
        q.SI >= 1 && q.SO >= 1 && q.KO <= 3 && q.KO * q.KI > 1 && cin_ok && g.n % 8 == 0 && g.n >= (glu ? 65 : gather_minn) && (q.ldc & 3) == 0 && (ldy & 3) == 0 &&
        a.y_bs == (int64_t)a.OR * q.IR * ldy && a16(x) && a16(y) && a16(g.w.p) && a16(g.b.p) && a.M < (1ll << 31) &&
        (int64_t)(q.KO * q.DO + q.PO + 1) * q.I * q.ldc < (1ll << 31) && (uint64_t)16 * (uint64_t)ldy * 4 < (1ull << 31)) {
      TdfDmaArgs d{};
      d.x = x;
      d.w = g.w.f();
      d.bias = g.b.f();
      d.zeros = e->d_zeros.f();
      d.y = y;
      d.M = a.M;
      d.N = g.n;
      d.K = g.k;
      d.C = 1;
      d.T = 1;
      d.relu = act;
      d.ldy = ldy;
      d.glu_cout = glu ? g.glu_c : 0;
      RowGather gq{};
      gq.O = q.O;
      gq.I = q.I;
      gq.KI = q.KI;
      gq.DO = q.DO;
      gq.DI = q.DI;
      gq.PO = q.PO;
      gq.PI = q.PI;
      gq.nch = (q.Cin + 31) / 32;
      gq.cin = q.Cin;
      gq.SO = q.SO;
      gq.SI = q.SI;
      gq.OR = a.OR;
      gq.IR = q.IR;
      gq.ldc = q.ldc;
      gq.x_bs = a.x_bs;
      bool done = false;
      CHK(timed(e, cls, flops, bytes, s, [&]() { done = launch_tdf3_gather_auto(e, d, gq, s); }));
      if (done) return ASX_OK;
    }
  }
  // stride-1 k3 / 3x3 convs with a halo packing run on the halo-tile kernel (the input block enters LDS once for all taps)
  HgGeom hgm;
  // A/B knob: launches whose halo grid would be smaller than ASX_HALO_MINBLK workgroups stay on gg_kernel (128-row tiles)
  static const int64_t halo_minblk = getenv("ASX_HALO_MINBLK") ? atoll(getenv("ASX_HALO_MINBLK")) : 0;
  if (g.wh.p != nullptr && q.SI == 1 && q.SO == 1 && q.KI == 3 && (q.KO == 1 || q.KO == 3) && q.KO * q.KI == g.wh_taps &&
      (mode == GG_DENSE || mode == GG_GLU) && res == nullptr && a.row_stat == nullptr && q.IR == q.I && a.OR == q.O &&
      (q.KO == 3 || q.O == 1) && (((uintptr_t)x | (uintptr_t)y) & 15) == 0 && hg_geometry(q.O, q.I, q.KO, q.DO, q.DI, &hgm) && (int64_t)q.O * q.I * q.ldc < (1ll << 31) &&   // 32-bit lane offsets inside an image
     
      (rows_outer / q.O) * hgm.tilesO * hgm.tilesI * ((g.n + g.wh_nt - 1) / g.wh_nt) >= halo_minblk) {
    HgArgs h{};
    h.x = x;
    h.wp = g.wh.f();
    h.bias = g.b.f();
    h.zeros = e->d_zeros.f();
    h.y = y;
    h.O = q.O;
    h.I = q.I;
    h.ldc = q.ldc;
    h.NCH = q.Cin / HG_KC;
    h.N = g.n;
    h.nbn = (g.n + g.wh_nt - 1) / g.wh_nt;
    h.x_bs = a.x_bs;
    h.y_bs = a.y_bs;
    h.ldy = ldy;
    h.DO = q.KO == 3 ? q.DO : 0;
    h.DI = q.DI;
    h.PO = q.KO == 3 ? q.PO : 0;
    h.PI = q.PI;
    h.til2 = hgm.til2;
    h.tilesO = hgm.tilesO;
    h.tilesI = hgm.tilesI;
    h.IWt = hgm.IWt;
    h.NPIX = hgm.NPIX;
    h.inb = hgm.inb;
    h.mode = mode;
    h.act = act;
    h.Cout = g.glu_c;
    const int Bn = (int)(rows_outer / q.O);
    return timed(e, cls, flops, bytes, s, [&]() {
      hg_dispatch(h, g.wh_nt, q.KO, Bn, s);
    });
  }
  return timed(e, cls, flops, bytes, s, [&]() {
    ht_gg_dispatch(a, s);
  });
}

// plain linear through the tuned row GEMM of kernels_net.h
static int ht_linear(asx_engine *e, const HtGemm &g, const float *x, int64_t lda, int64_t M, float *y, int64_t ldy, int act,
                     const float *res, int64_t ldr, hipStream_t s) {
  TdfDmaArgs d{};
  d.x = x;
  d.w = g.w.f();
  d.bias = g.b.f();
  d.res = res;
  d.zeros = e->d_zeros.f();
  d.y = y;
  d.M = M;
  d.N = g.n;
  d.K = g.k;
  d.C = 1;
  d.T = 1;
  d.relu = act;
  d.lda = lda;
  d.ldy = ldy;
  d.ldr = ldr;
  // 64 x 128 tiles, three workgroups per CU: 56.2 -> 53.7 ms per song on the transformer linears (ASX_HT_LINEAR_SMALL=0: A/B)
  static const bool lin_small = !(getenv("ASX_HT_LINEAR_SMALL") && atoi(getenv("ASX_HT_LINEAR_SMALL")) == 0);
  d.prefer_small = lin_small ? 1 : 0;
  if ((g.k & 3) || (lda & 3) || (ldy & 3) || (g.n & 3) || (res && (ldr & 3))) {
    set_err("ht_linear: K, N and row strides must be multiples of 4 floats");
    return ASX_ERR_INVALID;
  }
  const double flops = 2.0 * (double)M * g.n * g.k;
  const double bytes = 4.0 * ((double)M * g.k + (double)M * g.n * (res ? 2 : 1) + (double)g.n * g.k);
  return timed(e, ASX_PROF_TDF, flops, bytes, s, [&]() {
    launch_tdf_dma_auto(e, d, s);
  });
}

static int ht_ln(asx_engine *e, const float *x, int C, const DevBuf &g, const DevBuf &b, const float *pos, int64_t pos_mod,
                 float *y, int64_t M, hipStream_t s) {
  return timed(e, ASX_PROF_MISC, 0.0, 8.0 * (double)M * C, s, [&]() {
    hipLaunchKernelGGL(layernorm_kernel, dim3((unsigned)((M + 3) / 4)), dim3(256), 0, s, x, (int64_t)C, C, g.f(), b.f(),
                       pos, pos_mod, y, (int64_t)C, M, 1e-5f);
  });
}

// statistics of x viewed as [G1, R, P]; groups of gdiv consecutive plane elements; acc must hold G1*G2*2 doubles
static int ht_stats(asx_engine *e, const float *x, int G1, int64_t R, int64_t P, int64_t gdiv, int ld, int Cn, int G2,
                    double *acc, hipStream_t s) {
  HIPCHK(hipMemsetAsync(acc, 0, (size_t)G1 * G2 * 16, s));
  if (R == 1 && G2 == 1 && gdiv >= P && ld == 1 && P >= 65536) {
    // one flat group per item: fold the plane into rows so that a thread sums a column instead of 256 threads
    // hammering one LDS / global atomic each (0.5 ms -> tens of microseconds for a 688 k-sample segment)
    for (int64_t d = 1024; d >= 2; --d)
      if (P % d == 0 && P / d >= 256) {
        R = d;
        P /= d;
        gdiv = P;
        break;
      }
  }
  int64_t rsplit = R / 96;
  if (rsplit < 1) rsplit = 1;
  if (rsplit > 1024) rsplit = 1024;
  const unsigned gx = P >= 256 ? (unsigned)((P + 255) / 256) : 1u;
  return timed(e, ASX_PROF_MISC, 0.0, 4.0 * (double)G1 * R * P, s, [&]() {
    hipLaunchKernelGGL(gstats_kernel, dim3(gx, (unsigned)rsplit, (unsigned)G1), dim3(256), 0, s, x, R, P, gdiv, ld, Cn, G2,
                       acc);
  });
}

static int ht_gn(asx_engine *e, float *x, int G1, int64_t R, int G2, int ld, int Cn, const double *acc, const float *gam,
                 const float *bet, int mode, float *dst, int dst_ld, const float *ls, hipStream_t s) {
  const int Ce = mode == 1 ? Cn / 2 : Cn;
  const int64_t total = (int64_t)G1 * R * G2 * Ce;
  return timed(e, ASX_PROF_MISC, 0.0, 8.0 * (double)total, s, [&]() {
    hipLaunchKernelGGL(gn_apply_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, x, R, G2, ld, Cn, acc, gam,
                       bet, 1e-5f, mode, dst, dst_ld, ls, total);
  });
}

static int ht_mha(asx_engine *e, const float *q, int64_t ldq, const float *k, const float *v, int64_t ldkv, float *out,
                  int64_t ldo, int B, int nq, int nk, int heads, int dh, hipStream_t s) {
  MhaArgs a{};
  a.q = q;
  a.k = k;
  a.v = v;
  a.out = out;
  a.ldq = ldq;
  a.ldk = ldkv;
  a.ldv = ldkv;
  a.ldo = ldo;
  a.nq = nq;
  a.nk = nk;
  a.scale = 1.0f / sqrtf((float)dh);
  static const int attn_exact = getenv("ASX_ATTN_EXACT") != nullptr;
  a.exact = attn_exact;
  const double flops = 4.0 * (double)B * heads * (double)nq * nk * dh;
  const double bytes = 4.0 * (double)B * heads * dh * (2.0 * nq + 2.0 * nk);
  a.nqt = (nq + 63) / 64;
  a.heads = heads;
  const dim3 grid((unsigned)(a.nqt * heads * B));   // 1-D, XCD-aware (kernels_ht.h)
  return timed(e, ASX_PROF_CONV1X1, flops, bytes, s, [&]() {   // profile class shared with the Roformer attention
    // dh = 48: the single-buffered build runs FOUR workgroups per CU (118 registers, 26 KB of LDS): 55.0 -> 51.5 ms per song
    // against the double-buffered, one-barrier build with three (ASX_MHA_DB=1: A/B) -- occupancy, not the barrier count
    static const bool mha_db = getenv("ASX_MHA_DB") && atoi(getenv("ASX_MHA_DB")) != 0;
    // bf16 x 6 form (kernels_ht.h: mha6_kernel) under the process-wide switch of the row GEMM; ASX_MHA6=0: A/B
    static const bool mha6 = !(getenv("ASX_MHA6") && atoi(getenv("ASX_MHA6")) == 0);
    if (mha6 && e->gemm_bf16x6 > 0 && a.decay == nullptr && (dh == 48 || dh == 64) && (ldq & 3) == 0 && (ldkv & 3) == 0 &&
        (ldo & 3) == 0) {
      MhaArgs a6 = a;
      const bool wide = nq > 128;                      // 128 queries per workgroup on long sequences
      a6.nqt = wide ? (nq + 127) / 128 : (nq + 63) / 64;
      const dim3 grid6((unsigned)(a6.nqt * heads * B));
      if (e->gemm_f16x3 > 0) {                         // fp16 x 3 arithmetic (kernels_ht.h: template parameter H)
        if (dh == 48) {
          if (wide) hipLaunchKernelGGL((mha6_kernel<3, 2, true>), grid6, dim3(256), 0, s, a6);
          else hipLaunchKernelGGL((mha6_kernel<3, 1, true>), grid6, dim3(256), 0, s, a6);
        } else {
          if (wide) hipLaunchKernelGGL((mha6_kernel<4, 2, true>), grid6, dim3(256), 0, s, a6);
          else hipLaunchKernelGGL((mha6_kernel<4, 1, true>), grid6, dim3(256), 0, s, a6);
        }
        g_attn6h_launches.fetch_add(1);
      } else if (dh == 48) {
        if (wide) hipLaunchKernelGGL((mha6_kernel<3, 2>), grid6, dim3(256), 0, s, a6);
        else hipLaunchKernelGGL((mha6_kernel<3, 1>), grid6, dim3(256), 0, s, a6);
      } else {
        if (wide) hipLaunchKernelGGL((mha6_kernel<4, 2>), grid6, dim3(256), 0, s, a6);
        else hipLaunchKernelGGL((mha6_kernel<4, 1>), grid6, dim3(256), 0, s, a6);
      }
      g_attn6_launches.fetch_add(1);
      e->prof_nprod = e->gemm_f16x3 > 0 ? 3 : 6;
      return;
    }
    if (dh == 48 && mha_db) hipLaunchKernelGGL((mha_kernel<3, false, true>), grid, dim3(256), 0, s, a);
    else if (dh == 48) hipLaunchKernelGGL((mha_kernel<3>), grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL((mha_kernel<4>), grid, dim3(256), 0, s, a);
  });
}

// ---- workspace --------------------------------------------------------------------------------------------------
static int ht_ensure_workspace(asx_engine *e, int B) {
  HtNet &n = *e->ht;
  if (B <= n.ws_batch) return ASX_OK;
  const asx_ht_config &c = n.cfg;
  const int D = c.depth, S = c.n_sources, T = n.T;
  size_t off = 0;
  auto take = [&](size_t floats) {
    const size_t o = off;
    off += (floats * 4 + 255) & ~(size_t)255;
    return o;
  };
  std::vector<std::pair<float **, size_t>> plan;
  auto want = [&](float *&p, size_t floats) { plan.push_back({&p, take(floats)}); };
  auto &b = n.b;
  const size_t BT = (size_t)B * T;
  want(b.xf0, BT * n.F[0] * 4);
  want(b.xt0, (size_t)B * ((n.L[0] + 1) & ~(int64_t)1) * 2);
  size_t yf = 0, yt = 0, h = 0, rw = 0;
  for (int i = 0; i < D; ++i) {
    const int hp = (n.C[i] / c.dconv_comp + 3) & ~3;
    yf = std::max(yf, BT * n.F[i + 1] * n.C[i]);
    yt = std::max(yt, (size_t)B * n.L[i + 1] * n.C[i]);
    h = std::max(h, std::max(BT * n.F[i + 1], (size_t)B * n.L[i + 1]) * hp);
    rw = std::max(rw, std::max(BT * n.F[i + 1], (size_t)B * n.L[i + 1]) * n.C[i]);
  }
  want(b.yf, yf);
  want(b.yt, yt);
  want(b.h, h);
  want(b.rw, rw);
  b.skf.assign(D, nullptr);
  b.skt.assign(D, nullptr);
  b.df.assign(D + 1, nullptr);
  b.dt.assign(D + 1, nullptr);
  for (int i = 0; i < D; ++i) {
    want(b.skf[i], BT * n.F[i + 1] * n.C[i]);
    want(b.skt[i], (size_t)B * n.L[i + 1] * n.C[i]);
  }
  // decoder inputs: df[i] feeds decoder i (shape of skf[i]); df[0]' = final output, kept in df[D] slot as "out"
  for (int i = 0; i < D; ++i) {
    want(b.df[i + 1], BT * n.F[i + 1] * n.C[i]);
    want(b.dt[i + 1], (size_t)B * n.L[i + 1] * n.C[i]);
  }
  want(b.df[0], BT * n.F[0] * 4 * S);
  want(b.dt[0], (size_t)B * n.L[0] * 2 * S);
  if (c.t_layers > 0) {
    const size_t NF = BT * n.F[D], NT = (size_t)B * n.L[D], NM = std::max(NF, NT);
    want(b.tokf, NF * n.Ct);
    want(b.tokt, NT * n.Ct);
    want(b.xn, NM * n.Ct);
    want(b.xn2, NM * n.Ct);
    want(b.qkv, NM * 3 * n.Ct);
    want(b.kvb, NM * 2 * n.Ct);
    want(b.attf, NM * n.Ct);
    want(b.attt, NM * n.Ct);
    want(b.ffh, NM * n.hidden);
  }
  want(b.frames, (size_t)B * S * 2 * T * c.nfft);
  {   // per-row GroupNorm partials of the DConv GEMMs: [N tiles][rows][2]
    size_t rsz = 0;
    for (int i = 0; i < D; ++i) {
      const int np = ht_glu_rows(n.C[i]);
      const size_t tiles = std::max<size_t>((np + gg_tile_n(np, true) - 1) / gg_tile_n(np, true), 1);
      rsz = std::max(rsz, std::max(BT * n.F[i + 1], (size_t)B * n.L[i + 1]) * tiles * 2);
    }
    want(b.rowstat, rsz);
    want(b.mr_g, (size_t)B * std::max(n.F[1], 1) * 2);
  }
  CHK(n.ws.ensure(off));
  for (auto &pr : plan) *pr.first = reinterpret_cast<float *>(reinterpret_cast<char *>(n.ws.p) + pr.second);
  // float64 accumulators: per-sample (freq, time) + group-norm groups (max B * F[1])
  const size_t ng = (size_t)B * std::max(n.F[1], 1);
  CHK(n.acc.ensure((2 * (size_t)B + 2 * ng) * 16 + (size_t)B * 4));
  b.acc_f = reinterpret_cast<double *>(n.acc.p);
  b.acc_t = b.acc_f + 2 * (size_t)B;
  b.acc_g = b.acc_t + 2 * (size_t)B;
  b.acc_g2 = b.acc_g + 2 * ng;
  b.ticket = reinterpret_cast<unsigned *>(b.acc_g2 + 2 * ng);
  HIPCHK(hipMemset(b.ticket, 0, (size_t)B * 4));
  n.ws_batch = B;
  return ASX_OK;
}

// One layer d of the DConv residual branch (demucs.py:99-179) on y [B, O, I, C] in place; along_outer: the conv runs over
// the outer axis.  part 1 (head): dilated conv + GroupNorm + GELU -> h [B*O*I, hp] (the workspace's b.h); part 2 (tail):
// 1x1 conv + GroupNorm + GLU + LayerScale + residual add into y; 3: both.  Demucs v3 runs its BLSTM / LocalState inserts on
// h between the two parts (engine_hd.h).
static int ht_dconv_layer(asx_engine *e, const HtEnc &E, size_t d, int part, float *y, int B, int O, int I, bool along_outer,
                          hipStream_t s) {
  HtNet &n = *e->ht;
  const int C = E.cout;
  const HtDconv &dc = E.dc[d];
  const int dil = 1 << d;
  // GroupNorm(1, .) per conv batch item: (b, f) over (t, c) on the spectrogram branch, b over (l, c) on the waveform.
  // Statistics are accumulated by the producing GEMM's epilogue; the 1x1 conv runs twice (K = C/8 is tiny) --
  // once for the statistics, once to normalise + GLU + LayerScale + add into y -- so its 2C-wide output never
  // reaches HBM.
  const int G2 = along_outer ? I : 1;
  const int64_t R = along_outer ? O : I;
  const int64_t M = (int64_t)B * O * I;
  auto fold = [&](int ncols, double count, double *acc, float2 *mr, bool glu = false) {
    const int ntile = (ncols + gg_tile_n(ncols, glu) - 1) / gg_tile_n(ncols, glu);   // N tiles of ht_gg's launch choice
    unsigned gx = (unsigned)((G2 + 15) / 16);
    if (G2 == 1) {
      gx = (unsigned)std::min<int64_t>(64, std::max<int64_t>(1, R / 4096));
      if (gx > 1) HIPCHK(hipMemsetAsync(acc, 0, (size_t)B * 16, s));
    }
    return timed(e, ASX_PROF_MISC, 0.0, 8.0 * (double)M * ntile, s, [&]() {
      hipLaunchKernelGGL(rowstat_reduce_kernel, dim3(gx, (unsigned)B), dim3(256), 0, s, reinterpret_cast<const float2 *>(n.b.rowstat), M,
                         ntile, (int64_t)O * I, G2, R, count, 1e-5f, acc, mr, n.b.ticket);
    });
  };
  HtFuse f1;
  f1.row_stat = reinterpret_cast<float2 *>(n.b.rowstat);
  f1.g_outer = (int64_t)O * I;
  f1.g_mod = G2;
  if (part & 1) {
    HtGeom g;
    g.O = O;
    g.I = I;
    g.Cin = C;
    g.ldc = C;
    g.IR = I;
    if (along_outer) {
      g.KO = 3;
      g.DO = dil;
      g.PO = dil;
    } else {
      g.KI = 3;
      g.DI = dil;
      g.PI = dil;
    }
    CHK(ht_gg(e, dc.c1, y, g, (int64_t)B * O, n.b.h, dc.hp, GG_DENSE, 0, nullptr, 0, 0, 0, 0, s, &f1));
    CHK(fold(dc.hp, (double)R * dc.hid, n.b.acc_g, reinterpret_cast<float2 *>(n.b.mr_g)));
    CHK(ht_gn(e, n.b.h, B, R, G2, dc.hp, dc.hid, n.b.acc_g, dc.g1w.f(), dc.g1b.f(), 0, nullptr, 0, nullptr, s));
  }
  if (part & 2) {
    HtGeom g1;
    g1.O = O;
    g1.I = I;
    g1.Cin = dc.hp;
    g1.ldc = dc.hp;
    g1.IR = I;
    CHK(ht_gg(e, dc.c2, n.b.h, g1, (int64_t)B * O, nullptr, 2 * C, GG_STATS, 0, nullptr, 0, 0, 0, 0, s, &f1));
    CHK(fold(dc.c2.n, (double)R * 2 * C, n.b.acc_g2, reinterpret_cast<float2 *>(n.b.mr_g), true));
    HtFuse f3 = f1;
    f3.row_stat = nullptr;
    f3.stat_in = reinterpret_cast<const float2 *>(n.b.mr_g);
    f3.gamma = dc.g2w.f();
    f3.beta = dc.g2b.f();
    f3.ls = dc.ls.f();
    CHK(ht_gg(e, dc.c2, n.b.h, g1, (int64_t)B * O, y, C, GG_GNGLU, 0, nullptr, 0, 0, 0, 0, s, &f3));
  }
  return ASX_OK;
}

static int ht_dconv(asx_engine *e, const HtEnc &E, float *y, int B, int O, int I, bool along_outer, hipStream_t s) {
  for (size_t d = 0; d < E.dc.size(); ++d) CHK(ht_dconv_layer(e, E, d, 3, y, B, O, I, along_outer, s));
  return ASX_OK;
}

static int ht_tlayer_ff(asx_engine *e, const HtTLayer &L, float *x, int B, int64_t N, const DevBuf &nw, const DevBuf &nb,
                        hipStream_t s) {
  HtNet &n = *e->ht;
  const int C = n.Ct;
  const int64_t M = (int64_t)B * N;
  CHK(ht_ln(e, x, C, nw, nb, nullptr, 1, n.b.xn, M, s));
  CHK(ht_linear(e, L.l1, n.b.xn, C, M, n.b.ffh, n.hidden, 2, nullptr, 0, s));
  CHK(ht_linear(e, L.l2, n.b.ffh, n.hidden, M, x, C, 0, x, C, s));
  // norm_out = MyGroupNorm(1, C) over the whole (tokens, channels) sample (transformer.py:181-187, 265)
  CHK(ht_stats(e, x, B, N, C, C, C, C, 1, n.b.acc_g, s));
  CHK(ht_gn(e, x, B, N, 1, C, C, n.b.acc_g, L.now.f(), L.nob.f(), 2, nullptr, 0, nullptr, s));
  return ASX_OK;
}

static int ht_self_layer(asx_engine *e, const HtTLayer &L, float *x, int B, int64_t N, hipStream_t s) {
  HtNet &n = *e->ht;
  const int C = n.Ct, H = n.cfg.t_heads;
  const int64_t M = (int64_t)B * N;
  CHK(ht_ln(e, x, C, L.n1w, L.n1b, nullptr, 1, n.b.xn, M, s));
  CHK(ht_linear(e, L.q, n.b.xn, C, M, n.b.qkv, 3 * C, 0, nullptr, 0, s));
  CHK(ht_mha(e, n.b.qkv, 3 * C, n.b.qkv + C, n.b.qkv + 2 * C, 3 * C, n.b.attf, C, B, (int)N, (int)N, H, C / H, s));
  CHK(ht_linear(e, L.out, n.b.attf, C, M, x, C, 0, x, C, s));
  return ht_tlayer_ff(e, L, x, B, N, L.n2w, L.n2b, s);
}

// attention half of a cross layer: att = MHA(norm1(q), norm2(k), norm2(k)) (transformer.py:379-383)
static int ht_cross_attn(asx_engine *e, const HtTLayer &L, const float *q, int64_t Nq, const float *k, int64_t Nk, float *att,
                         int B, hipStream_t s) {
  HtNet &n = *e->ht;
  const int C = n.Ct, H = n.cfg.t_heads;
  CHK(ht_ln(e, q, C, L.n1w, L.n1b, nullptr, 1, n.b.xn, (int64_t)B * Nq, s));
  CHK(ht_ln(e, k, C, L.n2w, L.n2b, nullptr, 1, n.b.xn2, (int64_t)B * Nk, s));
  CHK(ht_linear(e, L.q, n.b.xn, C, (int64_t)B * Nq, n.b.qkv, C, 0, nullptr, 0, s));
  CHK(ht_linear(e, L.kv, n.b.xn2, C, (int64_t)B * Nk, n.b.kvb, 2 * C, 0, nullptr, 0, s));
  return ht_mha(e, n.b.qkv, C, n.b.kvb, n.b.kvb + C, 2 * C, att, C, B, (int)Nq, (int)Nk, H, C / H, s);
}

// ---- HTDemucs.forward on B full-length segments: seg [B, 2, TL] -> out [B, S, 2, TL] ----------------------------------
// encoder level i of both branches (hdemucs.py:139-170): skf[i-1] / skt[i-1] (or the network inputs) -> skf[i] / skt[i]
static int ht_enc_level(asx_engine *e, int i, int B, hipStream_t s) {
  HtNet &n = *e->ht;
  const asx_ht_config &c = n.cfg;
  auto &b = n.b;
  const int T = n.T;
  const int64_t TL = n.L[0];
  {
    const HtEnc &E = n.enc[i], &Et = n.tenc[i];
    const int C = n.C[i];
    {   // spectrogram branch (hdemucs.py:139-170, freq=True)
      HtGeom g;
      g.O = T;
      g.I = n.F[i];
      g.Cin = E.cin;
      g.ldc = E.cin;
      g.KI = c.kernel_size;
      g.PI = c.kernel_size / 4;
      g.SI = c.stride;
      g.IR = n.F[i + 1];
      CHK(ht_gg(e, E.conv, i == 0 ? b.xf0 : b.skf[i - 1], g, (int64_t)B * T, b.yf, C, GG_DENSE, 2, nullptr, 0, 0, 0, 0, s));
      CHK(ht_dconv(e, E, b.yf, B, T, n.F[i + 1], true, s));
      HtGeom r;
      r.O = T;
      r.I = n.F[i + 1];
      r.Cin = C;
      r.ldc = C;
      r.IR = n.F[i + 1];
      const bool emb = i == 0 && c.freq_emb_scale != 0.f;
      CHK(ht_gg(e, E.rewrite, b.yf, r, (int64_t)B * T, b.skf[i], C, GG_GLU, 0, emb ? n.femb.f() : nullptr, C,
                emb ? n.F[1] : 0, 0, 0, s));
    }
    {   // waveform branch (freq=False); layer 0 reads the stereo signal as pairs of samples x 2 channels
      HtGeom g;
      g.O = 1;
      if (i == 0) {
        g.I = (int)((TL + 1) / 2);              // an odd length carries one zero sample (Demucs v3 chunks, engine_hd.h)
        g.x_bs = (int64_t)g.I * 4;
        g.Cin = 4;
        g.ldc = 4;
        g.KI = c.kernel_size / 2;
        g.PI = c.kernel_size / 8;
        g.SI = c.stride / 2;
      } else {
        g.I = (int)n.L[i];
        g.Cin = Et.cin;
        g.ldc = Et.cin;
        g.KI = c.kernel_size;
        g.PI = c.kernel_size / 4;
        g.SI = c.stride;
      }
      g.IR = (int)n.L[i + 1];
      CHK(ht_gg(e, Et.conv, i == 0 ? b.xt0 : b.skt[i - 1], g, B, b.yt, C, GG_DENSE, 2, nullptr, 0, 0, 0, 0, s));
      CHK(ht_dconv(e, Et, b.yt, B, 1, (int)n.L[i + 1], false, s));
      HtGeom r;
      r.I = (int)n.L[i + 1];
      r.Cin = C;
      r.ldc = C;
      r.IR = (int)n.L[i + 1];
      CHK(ht_gg(e, Et.rewrite, b.yt, r, B, b.skt[i], C, GG_GLU, 0, nullptr, 0, 0, 0, 0, s));
    }
  }
  return ASX_OK;
}

// decoder level i of both branches (hdemucs.py:303-330): df[i+1] / dt[i+1] (already x + skip) -> df[i] / dt[i]
// (+ GELU + the next level's skip unless i == 0)
static int ht_dec_level(asx_engine *e, int i, int B, hipStream_t s) {
  HtNet &n = *e->ht;
  auto &b = n.b;
  const int T = n.T;
  {
    const HtDec &Dz = n.dec[i], &Dt = n.tdec[i];
    const int C = n.C[i];
    {
      HtGeom g;
      g.O = T;
      g.I = n.F[i + 1];
      g.Cin = C;
      g.ldc = C;
      g.KO = 3;
      g.KI = 3;
      g.PO = 1;
      g.PI = 1;
      g.IR = n.F[i + 1];
      CHK(ht_gg(e, Dz.rewrite, b.df[i + 1], g, (int64_t)B * T, b.rw, C, GG_GLU, 0, nullptr, 0, 0, 0, 0, s));
      HtGeom t;
      t.O = T;
      t.I = n.F[i + 1];
      t.Cin = C;
      t.ldc = C;
      t.KI = 2;
      t.PI = 1;
      t.IR = n.F[i + 1] + 1;
      CHK(ht_gg(e, Dz.convtr, b.rw, t, (int64_t)B * T, b.df[i], Dz.cout, GG_CONVT, i == 0 ? 0 : 2,
                i == 0 ? nullptr : b.skf[i - 1], Dz.cout, 0, n.F[i], Dz.cout, s));
    }
    {
      HtGeom g;
      g.I = (int)n.L[i + 1];
      g.Cin = C;
      g.ldc = C;
      g.KI = 3;
      g.PI = 1;
      g.IR = (int)n.L[i + 1];
      CHK(ht_gg(e, Dt.rewrite, b.dt[i + 1], g, B, b.rw, C, GG_GLU, 0, nullptr, 0, 0, 0, 0, s));
      HtGeom t;
      t.I = (int)n.L[i + 1];
      t.Cin = C;
      t.ldc = C;
      t.KI = 2;
      t.PI = 1;
      t.IR = (int)n.L[i + 1] + 1;
      CHK(ht_gg(e, Dt.convtr, b.rw, t, B, b.dt[i], Dt.cout, GG_CONVT, i == 0 ? 0 : 2, i == 0 ? nullptr : b.skt[i - 1],
                Dt.cout, 0, (int)n.L[i], Dt.cout, s));
    }
  }
  return ASX_OK;
}

static int ht_forward_dev(asx_engine *e, const float *seg, int B, float *out, hipStream_t s) {
  HtNet &n = *e->ht;
  const asx_ht_config &c = n.cfg;
  CHK(ht_ensure_workspace(e, B));
  auto &b = n.b;
  const int D = c.depth, S = c.n_sources, T = n.T, hop = c.nfft / 4;
  const int64_t TL = n.L[0];
  const int F0 = n.F[0];
  // spectrogram + per-sample standardisation of both branches (htdemucs.py:505-519)
  CHK(timed(e, ASX_PROF_STFT, 0.0, 4.0 * (double)B * (2.0 * TL + 4.0 * T * F0), s, [&]() {
    hipLaunchKernelGGL(ht_stft_kernel, dim3(T, 2, B), dim3(256), stft_lds(n.plan), s, seg, TL, (int64_t)0, TL, hop, T, b.xf0, n.window.f(),
                       reinterpret_cast<const float2 *>(n.tw.p), n.plan);
  }));
  const int64_t nf = (int64_t)T * F0 * 4;
  CHK(ht_stats(e, b.xf0, B, T, (int64_t)F0 * 4, (int64_t)F0 * 4, 4, 4, 1, b.acc_f, s));
  CHK(timed(e, ASX_PROF_MISC, 0.0, 8.0 * (double)B * nf, s, [&]() {
    hipLaunchKernelGGL(std_norm_kernel, dim3((unsigned)((nf + 255) / 256), B), dim3(256), 0, s, b.xf0, nf, b.acc_f);
  }));
  CHK(ht_stats(e, seg, B, 1, 2 * TL, 2 * TL, 1, 1, 1, b.acc_t, s));
  CHK(timed(e, ASX_PROF_MISC, 0.0, 16.0 * (double)B * TL, s, [&]() {
    hipLaunchKernelGGL(time_norm_kernel, dim3((unsigned)((TL + 255) / 256), B), dim3(256), 0, s, seg, TL, b.acc_t, b.xt0);
  }));
  // encoders
  for (int i = 0; i < D; ++i) CHK(ht_enc_level(e, i, B, s));
  // cross transformer (transformer.py:520-548)
  const int Cb = n.C[D - 1];
  const int64_t NF = (int64_t)T * n.F[D], NT = n.L[D];
  float *xin_f = b.df[D], *xin_t = b.dt[D];   // decoder inputs of the deepest level
  if (c.t_layers > 0) {
    const int Ct = n.Ct;
    const float *xf = b.skf[D - 1], *xt = b.skt[D - 1];
    if (c.bottom_channels > 0) {
      CHK(ht_linear(e, n.up, xf, Cb, B * NF, b.xn, Ct, 0, nullptr, 0, s));
      CHK(ht_ln(e, b.xn, Ct, n.nin_w, n.nin_b, n.pos_f.f(), NF, b.tokf, B * NF, s));
      CHK(ht_linear(e, n.up_t, xt, Cb, B * NT, b.xn, Ct, 0, nullptr, 0, s));
      CHK(ht_ln(e, b.xn, Ct, n.nint_w, n.nint_b, n.pos_t.f(), NT, b.tokt, B * NT, s));
    } else {
      CHK(ht_ln(e, xf, Ct, n.nin_w, n.nin_b, n.pos_f.f(), NF, b.tokf, B * NF, s));
      CHK(ht_ln(e, xt, Ct, n.nint_w, n.nint_b, n.pos_t.f(), NT, b.tokt, B * NT, s));
    }
    for (int i = 0; i < c.t_layers; ++i) {
      const HtTLayer &Lf = n.lay[i], &Lt = n.lay_t[i];
      if (!Lf.cross) {
        CHK(ht_self_layer(e, Lf, b.tokf, B, NF, s));
        CHK(ht_self_layer(e, Lt, b.tokt, B, NT, s));
      } else {
        // both attentions read the pre-layer tokens (transformer.py:543-545), then each branch updates in place
        CHK(ht_cross_attn(e, Lf, b.tokf, NF, b.tokt, NT, b.attf, B, s));
        CHK(ht_cross_attn(e, Lt, b.tokt, NT, b.tokf, NF, b.attt, B, s));
        CHK(ht_linear(e, Lf.out, b.attf, Ct, B * NF, b.tokf, Ct, 0, b.tokf, Ct, s));
        CHK(ht_tlayer_ff(e, Lf, b.tokf, B, NF, Lf.n3w, Lf.n3b, s));
        CHK(ht_linear(e, Lt.out, b.attt, Ct, B * NT, b.tokt, Ct, 0, b.tokt, Ct, s));
        CHK(ht_tlayer_ff(e, Lt, b.tokt, B, NT, Lt.n3w, Lt.n3b, s));
      }
    }
    if (c.bottom_channels > 0) {   // + skip of the deepest level fused as the residual (hdemucs.py:305)
      CHK(ht_linear(e, n.down, b.tokf, Ct, B * NF, xin_f, Cb, 0, b.skf[D - 1], Cb, s));
      CHK(ht_linear(e, n.down_t, b.tokt, Ct, B * NT, xin_t, Cb, 0, b.skt[D - 1], Cb, s));
    } else {
      HIPCHK(hipMemcpyAsync(xin_f, b.tokf, (size_t)B * NF * Cb * 4, hipMemcpyDeviceToDevice, s));
      HIPCHK(hipMemcpyAsync(xin_t, b.tokt, (size_t)B * NT * Cb * 4, hipMemcpyDeviceToDevice, s));
    }
  } else {
    HIPCHK(hipMemcpyAsync(xin_f, b.skf[D - 1], (size_t)B * NF * Cb * 4, hipMemcpyDeviceToDevice, s));
    HIPCHK(hipMemcpyAsync(xin_t, b.skt[D - 1], (size_t)B * NT * Cb * 4, hipMemcpyDeviceToDevice, s));
  }
  if (c.t_layers == 0 || c.bottom_channels == 0) {
    const int64_t n1 = (int64_t)B * NF * Cb, n2 = (int64_t)B * NT * Cb;
    hipLaunchKernelGGL(add_kernel, dim3((unsigned)((n1 + 255) / 256)), dim3(256), 0, s, xin_f, b.skf[D - 1], n1);
    hipLaunchKernelGGL(add_kernel, dim3((unsigned)((n2 + 255) / 256)), dim3(256), 0, s, xin_t, b.skt[D - 1], n2);
    HIPCHK(hipGetLastError());
  }
  // decoders, deepest first (hdemucs.py:303-330); the input already holds x + skip
  for (int i = D - 1; i >= 0; --i) CHK(ht_dec_level(e, i, B, s));
  // CaC -> iSTFT, + waveform branch (htdemucs.py:589-612)
  CHK(timed(e, ASX_PROF_ISTFT, 0.0, 4.0 * (double)B * S * 2 * T * (2.0 * F0 + c.nfft), s, [&]() {
    hipLaunchKernelGGL(ht_istft_kernel, dim3(T, S * 2, B), dim3(256), ht_istft_lds(n.plan), s, b.df[0], T, 4 * S, b.acc_f,
                       (double)nf, b.frames, n.window.f(), reinterpret_cast<const float2 *>(n.tw.p), n.plan);
  }));
  return timed(e, ASX_PROF_OLA, 0.0, 4.0 * (double)B * S * 2 * (T * (double)c.nfft + 2.0 * TL), s, [&]() {
    hipLaunchKernelGGL(ht_ola_kernel, dim3((unsigned)((TL + 255) / 256), S * 2, B), dim3(256), 0, s, b.frames,
                       n.env_hop.f(), c.nfft, hop, T, TL, b.dt[0], 2 * S, b.acc_t, out);
  });
}

// 2*MAC per segment of the GEMM-shaped work
static double ht_flops(const asx_engine *e) {
  const HtNet &n = *e->ht;
  const asx_ht_config &c = n.cfg;
  const int D = c.depth;
  double f = 0.0;
  for (int i = 0; i < D; ++i) {
    const double pf = (double)n.T * n.F[i + 1], pt = (double)n.L[i + 1];
    const double C = n.C[i], hid = n.C[i] / c.dconv_comp;
    const double cin_z = i == 0 ? 4 : n.C[i - 1], cin_t = i == 0 ? 2 : n.C[i - 1];
    const double out_z = i == 0 ? 4.0 * c.n_sources : n.C[i - 1], out_t = i == 0 ? 2.0 * c.n_sources : n.C[i - 1];
    f += 2.0 * pf * C * cin_z * c.kernel_size + 2.0 * pt * C * cin_t * c.kernel_size;        // encoder conv
    f += (pf + pt) * c.dconv_depth * 2.0 * (3.0 * C * hid + hid * 2.0 * C);                 // DConv
    f += (pf + pt) * 2.0 * C * 2.0 * C;                                                       // rewrite
    f += pf * 2.0 * 9.0 * C * 2.0 * C + pt * 2.0 * 3.0 * C * 2.0 * C;                         // decoder rewrite
    f += (pf * out_z + pt * out_t) * 2.0 * C * c.kernel_size;                                 // transposed conv
  }
  if (c.t_layers > 0) {
    const double NF = (double)n.T * n.F[D], NT = (double)n.L[D], Ct = n.Ct, H = n.hidden, Cb = n.C[D - 1];
    if (c.bottom_channels > 0) f += (NF + NT) * 4.0 * Cb * Ct;
    for (int i = 0; i < c.t_layers; ++i) {
      f += (NF + NT) * (2.0 * 4.0 * Ct * Ct + 4.0 * Ct * H);
      if (i % 2 == 0) f += 4.0 * Ct * (NF * NF + NT * NT);
      else f += 8.0 * Ct * NF * NT;
    }
  }
  return f;
}

// ---- apply_model + demix_demucs ---------------------------------------------------------------------------------------
// The segment-forwards of one call, in the reference's order: shift 0's chunks, shift 1's chunks, ...
struct HtShift {
  int64_t offset, VL;
  int first, nk;          // range inside the global segment list
};
struct HtPlan {
  int64_t stride, segment, max_shift;
  std::vector<HtShift> shifts;
  std::vector<int64_t> starts;   // song index of model-input sample 0 of every segment
};

static int ht_plan(const asx_engine *e, int64_t N, int32_t shifts, const int64_t *offsets, double overlap, HtPlan &p) {
  const HtNet &n = *e->ht;
  const int64_t TL = n.L[0];
  p.segment = TL;
  p.stride = (int64_t)((1.0 - overlap) * (double)TL);   // int((1 - overlap) * segment), apply.py:220
  REQUIRE(p.stride >= 1 && p.stride <= TL, "overlap %g gives a bad stride", overlap);
  p.max_shift = shifts > 0 ? n.cfg.samplerate / 2 : 0;
  p.shifts.clear();
  p.starts.clear();
  const int nsh = shifts > 0 ? shifts : 1;
  for (int si = 0; si < nsh; ++si) {
    HtShift sh;
    sh.offset = shifts > 0 ? offsets[si] : 0;
    REQUIRE(sh.offset >= 0 && sh.offset <= p.max_shift, "shift offset %lld outside [0, %lld]", (long long)sh.offset, (long long)p.max_shift);
    // view = padded_mix[offset : offset + N + max_shift - offset]; padded index q <-> song index q - max_shift
    sh.VL = N + p.max_shift - sh.offset;
    sh.first = (int)p.starts.size();
    for (int64_t off = 0; off < sh.VL; off += p.stride) {
      const int64_t clen = std::min(sh.VL - off, TL);
      p.starts.push_back(sh.offset + off - (TL - clen) / 2 - p.max_shift);
    }
    sh.nk = (int)p.starts.size() - sh.first;
    p.shifts.push_back(sh);
  }
  return ASX_OK;
}

static int ht_ref_stats(asx_engine *e, const float *mix_dev, int64_t N, hipStream_t s) {
  // ref = mix.mean(0); mean / unbiased std of ref (demucs_separator.py:171-173)
  HtNet &n = *e->ht;
  CHK(n.ref.ensure((size_t)N * 4));
  CHK(n.ref_acc.ensure(16));
  hipLaunchKernelGGL(ht_mono_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, s, mix_dev, N, n.ref.f());
  HIPCHK(hipGetLastError());
  return ht_stats(e, n.ref.f(), 1, 1, N, N, 1, 1, 1, reinterpret_cast<double *>(n.ref_acc.p), s);
}

// segment-forwards [k0, k1) of the global list -> chunk_out [k1-k0, S, 2, TL]
static int ht_segments_dev(asx_engine *e, const float *mix_dev, int64_t N, const HtPlan &p, uint32_t flags, int k0, int k1, float *chunk_out,
                           hipStream_t s) {
  HtNet &n = *e->ht;
  const asx_ht_config &c = n.cfg;
  const int S = c.n_sources;
  const int64_t TL = n.L[0];
  const int standardize = (flags & ASX_HT_STANDARDIZE) ? 1 : 0;
  if (k1 <= k0) return ASX_OK;
  if (standardize) CHK(ht_ref_stats(e, mix_dev, N, s));
  const int nk = k1 - k0;
  // segments per forward: an engine knob (results do not depend on it).  4-minute song, 84 segments: 14 / 21 / 28 / 42 per batch
  // -> 871 / 899 / 926 / 928x real time (round 3; round 1: 611 vs 536x for 16 vs 8) -- the inner levels and the transformer fill
  // the chip only with many segments.
  const int maxB = c.max_batch > 0 ? c.max_batch : 32;
  const int nbatch = (nk + maxB - 1) / maxB;
  const int per = (nk + nbatch - 1) / nbatch;
  CHK(n.seg.ensure((size_t)per * 2 * TL * 4));
  for (int j = 0; j < nk; j += per) {
    const int B = std::min(per, nk - j);
    ht_gather_launch(mix_dev, N, p.starts.data() + k0 + j, B, TL, reinterpret_cast<const double *>(n.ref_acc.p), standardize, n.seg.f(), s);
    HIPCHK(hipGetLastError());
    CHK(ht_forward_dev(e, n.seg.f(), B, chunk_out + (size_t)j * S * 2 * TL, s));
  }
  return ASX_OK;
}

// triangular fold per shift, mean over shifts, de-standardise, stem swap: chunk_out [all segments, S, 2, TL] -> out [S, 2, N]
static int ht_fold_dev(asx_engine *e, const float *mix_dev, int64_t N, const HtPlan &p, uint32_t flags, const float *chunk_out, float *out_dev,
                       hipStream_t s) {
  HtNet &n = *e->ht;
  const int S = n.cfg.n_sources;
  const int64_t TL = n.L[0];
  const int standardize = (flags & ASX_HT_STANDARDIZE) ? 1 : 0;
  const int swap01 = (flags & ASX_HT_SWAP01) ? 1 : 0;
  if (standardize) CHK(ht_ref_stats(e, mix_dev, N, s));
  const int nsh = (int)p.shifts.size();
  for (int si = 0; si < nsh; ++si) {
    const HtShift &sh = p.shifts[si];
    CHK(timed(e, ASX_PROF_FINALIZE, 0.0, 4.0 * ((double)sh.nk * S * 2 * TL + 2.0 * S * 2 * N), s, [&]() {
      hipLaunchKernelGGL(ht_fold_kernel, dim3((unsigned)((N + 255) / 256), S * 2), dim3(256), 0, s, chunk_out + (size_t)sh.first * S * 2 * TL,
                         sh.nk, S * 2, TL, p.stride, p.segment, sh.VL, p.max_shift - sh.offset, n.fold_w.f(), si == 0 ? 1 : 0,
                         si == nsh - 1 ? 1 : 0, nsh, reinterpret_cast<const double *>(n.ref_acc.p), standardize, swap01, 1, N, out_dev);
    }));
  }
  return ASX_OK;
}

// mix_dev [2, N] -> out_dev [S, 2, N].  offsets: `shifts` draws of random.randint(0, samplerate/2) made by the host
// (apply.py:209); shifts == 0 runs the plain split path.
static int ht_demix_dev(asx_engine *e, const float *mix_dev, int64_t N, int32_t shifts, const int64_t *offsets, double overlap,
                        uint32_t flags, float *out_dev, hipStream_t s) {
  HtNet &n = *e->ht;
  HtPlan p;
  CHK(ht_plan(e, N, shifts, offsets, overlap, p));
  const int nseg = (int)p.starts.size();
  CHK(n.chunk_out.ensure((size_t)nseg * n.cfg.n_sources * 2 * n.L[0] * 4));
  CHK(ht_segments_dev(e, mix_dev, N, p, flags, 0, nseg, n.chunk_out.f(), s));
  return ht_fold_dev(e, mix_dev, N, p, flags, n.chunk_out.f(), out_dev, s);
}
