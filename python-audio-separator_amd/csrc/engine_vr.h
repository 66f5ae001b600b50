// VR architecture on the engine: VRSeparator.loading_mix / inference_vr / spec_to_wav
// (architectures/vr_separator.py:255-375), the spec_utils band logic they call and CascadedASPPNet
// (uvr_lib_v5/vr_network/nets.py:95-175, layers.py).  Included by asx.hip (one TU).
#pragma once

struct VrConv {
  HtGemm g;
  int cin_p = 0, cout = 0, k = 1, act = 1;
};
struct VrSep {
  DevBuf dw;     // [9][C]
  VrConv pw;
  int dil = 1;
};
struct VrBase {
  int nin_p = 0, ch = 0, n_enc = 4, nb = 5;
  std::vector<VrConv> enc1, enc2, dec;   // index 0 = level 1
  VrConv aspp1, aspp2, bott;
  std::vector<VrSep> sep;
  std::vector<int> C;                    // channels per level
};
struct VrFilt {
  int up = 1, down = 1, hlen = 0, n_pre_remove = 0;
  DevBuf h32, h64;
};
struct VrBand {
  asx_vr_band b{};
  FftPlan plan{};
  DevBuf window, tw, gain_syn;
  VrFilt ana, syn;    // ana: band d+1 -> d (float32); syn: band d -> d+1 (float64 accumulate)
};

struct VrNet {
  asx_vr_config cfg{};
  bool begun = false, ready = false;
  VrBase s1l, s1h, s2, s3;
  VrConv br2, br3, outc;
  std::vector<VrBand> band;   // index 0 = band 1
  DevBuf gain_ana;            // [bins + 1]
  int max_bin = 0, nb1 = 0, ctot = 0;
  // workspace
  int ws_batch = 0;
  DevBuf ws;
  struct {
    float *hc, *y2, *y3, *h3, *mk, *pool, *pool2, *tmp, *cat, *bn;
    std::vector<float *> D, E, O;
  } b;
  DevBuf X, M, M2, peak, fmin, wgt, frames, wss;
  std::vector<DevBuf> wav_ana, wav_syn, wav_up;
};

static void vr_free(VrNet &n) {
  auto fc = [](VrConv &c) {
    c.g.w.release();
    c.g.b.release();
  };
  for (VrBase *bs : {&n.s1l, &n.s1h, &n.s2, &n.s3}) {
    for (auto *v : {&bs->enc1, &bs->enc2, &bs->dec})
      for (auto &c : *v) fc(c);
    fc(bs->aspp1);
    fc(bs->aspp2);
    fc(bs->bott);
    for (auto &s : bs->sep) {
      s.dw.release();
      fc(s.pw);
    }
    bs->enc1.clear();
    bs->enc2.clear();
    bs->dec.clear();
    bs->sep.clear();
  }
  fc(n.br2);
  fc(n.br3);
  fc(n.outc);
  for (auto &b : n.band) {
    for (DevBuf *p : {&b.window, &b.tw, &b.gain_syn, &b.ana.h32, &b.ana.h64, &b.syn.h32, &b.syn.h64}) p->release();
  }
  n.band.clear();
  for (DevBuf *p : {&n.gain_ana, &n.ws, &n.X, &n.M, &n.M2, &n.peak, &n.fmin, &n.wgt, &n.frames, &n.wss}) p->release();
  for (auto *v : {&n.wav_ana, &n.wav_syn, &n.wav_up})
    for (auto &d : *v) d.release();
  n.ready = false;
  n.ws_batch = 0;
}
static void vr_destroy(VrNet *n) {
  vr_free(*n);
  delete n;
}

// Conv2d (no bias) + BatchNorm2d (eval) folded; weight [cout, cin, k, k] with kh on the frequency (outer) axis ->
// W[n][(kh*k + kw) * cin_p + map[ci]];  map: reference input channel -> padded engine channel
static int vr_pack_conv(asx_engine *e, VrConv &c, const std::string &wname, const std::string &bn, int cout, int cin, int k,
                        const std::vector<int> &map, int cin_p, int act, int cout_p = 0) {
  const float *w;
  CHK(get_tensor(e, wname, (int64_t)cout * cin * k * k, &w));
  const float *gam = nullptr, *bet = nullptr, *mu = nullptr, *var = nullptr;
  if (!bn.empty()) {
    CHK(get_tensor(e, bn + ".weight", cout, &gam));
    CHK(get_tensor(e, bn + ".bias", cout, &bet));
    CHK(get_tensor(e, bn + ".running_mean", cout, &mu));
    CHK(get_tensor(e, bn + ".running_var", cout, &var));
  }
  if (!cout_p) cout_p = cout;
  const int K = k * k * cin_p;
  std::vector<float> pw((size_t)cout_p * K, 0.f), pb((size_t)cout_p, 0.f);
  for (int n = 0; n < cout; ++n) {
    double sc = 1.0, sh = 0.0;
    if (gam) {
      sc = (double)gam[n] / sqrt((double)var[n] + 1e-5);
      sh = (double)bet[n] - (double)mu[n] * sc;
    }
    pb[n] = (float)sh;
    for (int ci = 0; ci < cin; ++ci)
      for (int kh = 0; kh < k; ++kh)
        for (int kw = 0; kw < k; ++kw)
          pw[(size_t)n * K + ((size_t)kh * k + kw) * cin_p + map[ci]] = (float)((double)w[(((size_t)n * cin + ci) * k + kh) * k + kw] * sc);
  }
  c.cin_p = cin_p;
  c.cout = cout_p;
  c.k = k;
  c.act = act;
  c.g.n = cout_p;
  c.g.k = K;
  CHK(ht_up(c.g.w, pw));
  CHK(ht_up(c.g.b, pb));
  return ASX_OK;
}

static std::vector<int> vr_ident(int n) {
  std::vector<int> m(n);
  for (int i = 0; i < n; ++i) m[i] = i;
  return m;
}
// [x(2) | rest] -> [x(2) pad(2) | rest]
static std::vector<int> vr_xmap(int n) {
  std::vector<int> m(n);
  for (int i = 0; i < n; ++i) m[i] = i < 2 ? i : i + 2;
  return m;
}

static int vr_load_base(asx_engine *e, VrBase &bs, const std::string &p, int nin, int ch, int arch) {
  bs.n_enc = arch == 129605 ? 5 : 4;
  bs.nb = arch == 129605 ? 6 : ((arch == 537238 || arch == 537227 || arch == 33966) ? 7 : 5);
  bs.ch = ch;
  bs.nin_p = (nin + 3) & ~3;
  REQUIRE(ch % 4 == 0, "VR net width %d must be a multiple of 4", ch);
  bs.C.assign(bs.n_enc, 0);
  bs.enc1.assign(bs.n_enc, VrConv());
  bs.enc2.assign(bs.n_enc, VrConv());
  bs.dec.assign(bs.n_enc, VrConv());
  int cin = nin;
  for (int i = 0; i < bs.n_enc; ++i) {
    const int c = ch << i;
    bs.C[i] = c;
    const std::string en = p + ".enc" + std::to_string(i + 1);
    const int cin_p = i == 0 ? bs.nin_p : cin;
    CHK(vr_pack_conv(e, bs.enc1[i], en + ".conv1.conv.0.weight", en + ".conv1.conv.1", c, cin, 3, (i == 0 && nin == 2) ? vr_xmap(cin) : vr_ident(cin),
                     cin_p, 4));
    CHK(vr_pack_conv(e, bs.enc2[i], en + ".conv2.conv.0.weight", en + ".conv2.conv.1", c, c, 3, vr_ident(c), c, 4));
    const std::string dn = p + ".dec" + std::to_string(i + 1) + ".conv";
    CHK(vr_pack_conv(e, bs.dec[i], dn + ".conv.0.weight", dn + ".conv.1", c, 3 * c, 3, vr_ident(3 * c), 3 * c, 1));
    cin = c;
  }
  const int ca = bs.C.back();
  const std::string a = p + ".aspp";
  CHK(vr_pack_conv(e, bs.aspp1, a + ".conv1.1.conv.0.weight", a + ".conv1.1.conv.1", ca, ca, 1, vr_ident(ca), ca, 1));
  CHK(vr_pack_conv(e, bs.aspp2, a + ".conv2.conv.0.weight", a + ".conv2.conv.1", ca, ca, 1, vr_ident(ca), ca, 1));
  const int dil[5] = {4, 8, 16, 16, 16};
  bs.sep.assign(bs.nb - 2, VrSep());
  for (int j = 0; j < bs.nb - 2; ++j) {
    const std::string s = a + ".conv" + std::to_string(j + 3);
    const float *dw;
    CHK(get_tensor(e, s + ".conv.0.weight", (int64_t)ca * 9, &dw));
    std::vector<float> t((size_t)9 * ca);
    for (int c = 0; c < ca; ++c)
      for (int q = 0; q < 9; ++q) t[(size_t)q * ca + c] = dw[(size_t)c * 9 + q];
    CHK(ht_up(bs.sep[j].dw, t));
    bs.sep[j].dil = dil[j];
    CHK(vr_pack_conv(e, bs.sep[j].pw, s + ".conv.1.weight", s + ".conv.2", ca, ca, 1, vr_ident(ca), ca, 1));
  }
  CHK(vr_pack_conv(e, bs.bott, a + ".bottleneck.0.conv.0.weight", a + ".bottleneck.0.conv.1", 2 * ca, bs.nb * ca, 1,
                   vr_ident(bs.nb * ca), bs.nb * ca, 1));
  return ASX_OK;
}

static double vr_i0(double x) {
  double s = 1.0, t = 1.0;
  const double q = x * x / 4.0;
  for (int k = 1; k < 200; ++k) {
    t *= q / ((double)k * k);
    s += t;
    if (t < 1e-18 * s) break;
  }
  return s;
}

// scipy.signal.resample_poly's filter: firwin(2*half_len + 1, 1/max(up, down), window=("kaiser", 5.0)) * up, left-padded
// with n_pre_pad zeros.  f32: the float32 variant (`.astype(x.dtype)` then `h *= up` in float32) used for float32 input.
static int vr_design_filter(VrFilt &f, int orig_sr, int target_sr) {
  int a = target_sr, b = orig_sr;
  while (b) {
    const int t = a % b;
    a = b;
    b = t;
  }
  f.up = target_sr / a;
  f.down = orig_sr / a;
  if (f.up == 1 && f.down == 1) return ASX_OK;
  const int max_rate = std::max(f.up, f.down);
  const double fc = 1.0 / max_rate;
  const int half_len = 10 * max_rate;
  const int nt = 2 * half_len + 1;
  std::vector<double> h(nt);
  const double alpha = 0.5 * (nt - 1);
  double sum = 0.0;
  for (int i = 0; i < nt; ++i) {
    const double m = i - alpha;
    const double x = fc * m;
    const double sinc = x == 0.0 ? 1.0 : sin(M_PI * x) / (M_PI * x);
    const double r = (i - alpha) / alpha;
    const double win = vr_i0(5.0 * sqrt(std::max(0.0, 1.0 - r * r))) / vr_i0(5.0);
    h[i] = fc * sinc * win;
    sum += h[i];
  }
  for (auto &v : h) v /= sum;
  const int n_pre_pad = f.down - half_len % f.down;
  f.n_pre_remove = (half_len + n_pre_pad) / f.down;
  f.hlen = n_pre_pad + nt;
  std::vector<float> h32((size_t)f.hlen, 0.f);
  std::vector<double> h64((size_t)f.hlen, 0.0);
  for (int i = 0; i < nt; ++i) {
    h32[n_pre_pad + i] = (float)h[i] * (float)f.up;
    h64[n_pre_pad + i] = h[i] * (double)f.up;
  }
  CHK(ht_up(f.h32, h32));
  CHK(f.h64.ensure(h64.size() * 8));
  HIPCHK(hipMemcpy(f.h64.p, h64.data(), h64.size() * 8, hipMemcpyHostToDevice));
  return ASX_OK;
}

static int vr_commit(asx_engine *e) {
  VrNet &n = *e->vr;
  const asx_vr_config &c = n.cfg;
  n.nb1 = c.bins + 1;
  n.max_bin = c.bins;          // CascadedASPPNet(n_fft = bins * 2): max_bin = n_fft // 2
  const int ch1 = c.cap[0], c2b = c.cap[1], ch2 = c.cap[2], c3b = c.cap[3], ch3 = c.cap[4];
  const int lev = c.arch == 129605 ? 5 : 4;
  REQUIRE(c.bins % (2 << lev) == 0, "bins %d: each half band must halve %d times", c.bins, lev);
  REQUIRE(c.window_size % (1 << lev) == 0 && c.window_size > 2 * c.offset, "window_size %d must be a multiple of %d and exceed 2*offset",
          c.window_size, 1 << lev);
  n.ctot = 4 + ch1 + ch2;
  CHK(vr_load_base(e, n.s1l, "stg1_low_band_net", 2, ch1, c.arch));
  CHK(vr_load_base(e, n.s1h, "stg1_high_band_net", 2, ch1, c.arch));
  CHK(vr_pack_conv(e, n.br2, "stg2_bridge.conv.0.weight", "stg2_bridge.conv.1", c2b, 2 + ch1, 1, vr_xmap(2 + ch1), 4 + ch1, 1));
  CHK(vr_load_base(e, n.s2, "stg2_full_band_net", c2b, ch2, c.arch));
  CHK(vr_pack_conv(e, n.br3, "stg3_bridge.conv.0.weight", "stg3_bridge.conv.1", c3b, 2 + ch1 + ch2, 1, vr_xmap(2 + ch1 + ch2),
                   4 + ch1 + ch2, 1));
  CHK(vr_load_base(e, n.s3, "stg3_full_band_net", c3b, ch3, c.arch));
  CHK(vr_pack_conv(e, n.outc, "out.weight", "", 2, ch3, 1, vr_ident(ch3), ch3, 5, 4));
  REQUIRE(c2b % 4 == 0 && c3b % 4 == 0, "bridge widths must be multiples of 4");
  // bands
  const int NB = c.n_bands;
  n.band.assign(NB, VrBand());
  int off = 0;
  for (int d = 0; d < NB; ++d) {
    VrBand &B = n.band[d];
    B.b = c.band[d];
    REQUIRE(make_plan(B.b.n_fft, &B.plan), "band %d: n_fft/2 = %d must factor into {2,3,5}", d + 1, B.b.n_fft / 2);
    REQUIRE(B.b.crop_start >= 0 && B.b.crop_stop <= B.b.n_fft / 2 + 1 && B.b.crop_start < B.b.crop_stop, "band %d: bad crop", d + 1);
    std::vector<float> w;
    host_window(B.b.n_fft, w);
    CHK(ht_up(B.window, w));
    std::vector<float> tw((size_t)B.b.n_fft * 2);
    for (int j = 0; j < B.b.n_fft; ++j) {
      const double ang = -2.0 * M_PI * (double)j / (double)B.b.n_fft;
      tw[2 * j] = (float)cos(ang);
      tw[2 * j + 1] = (float)sin(ang);
    }
    CHK(ht_up(B.tw, tw));
    // synthesis gains: fft_hp_filter / fft_lp_filter of cmb_spectrogram_to_wave (spec_utils.py:355-388, 411-429)
    const int nbin = B.b.n_fft / 2 + 1;
    std::vector<double> g(nbin, 1.0);
    auto hp = [&](int bs, int be) {   // fft_hp_filter(spec, bs, be)
      double gg = 1.0;
      for (int b = bs; b > be; --b) {
        gg -= 1.0 / (bs - be);
        if (b >= 0 && b < nbin) g[b] *= gg;
      }
      for (int b = 0; b <= be && b < nbin; ++b) g[b] = 0.0;
    };
    auto lp = [&](int bs, int be) {   // fft_lp_filter(spec, bs, be)
      double gg = 1.0;
      for (int b = bs; b < be; ++b) {
        gg -= 1.0 / (be - bs);
        if (b >= 0 && b < nbin) g[b] *= gg;
      }
      for (int b = std::max(be, 0); b < nbin; ++b) g[b] = 0.0;
    };
    if (d == NB - 1) {
      if (B.b.hpf_start > 0) hp(B.b.hpf_start, B.b.hpf_stop - 1);
    } else if (d == 0) {
      lp(B.b.lpf_start, B.b.lpf_stop);
    } else {
      hp(B.b.hpf_start, B.b.hpf_stop - 1);
      lp(B.b.lpf_start, B.b.lpf_stop);
    }
    std::vector<float> gf(g.begin(), g.end());
    CHK(ht_up(B.gain_syn, gf));
    if (d + 1 < NB) {
      CHK(vr_design_filter(B.ana, c.band[d + 1].sr, B.b.sr));
      CHK(vr_design_filter(B.syn, B.b.sr, c.band[d + 1].sr));
    }
    off += B.b.crop_stop - B.b.crop_start;
  }
  REQUIRE(off <= c.bins, "Too much bins");
  // analysis gains over the combined rows (combine_spectrograms, spec_utils.py:266-279), float32 like complex64 *= float
  {
    std::vector<float> g((size_t)n.nb1, 1.0f);
    if (c.pre_filter_start > 0) {
      if (NB == 1) {
        double gg = 1.0;
        for (int b = c.pre_filter_start; b < c.pre_filter_stop; ++b) {
          gg -= 1.0 / (c.pre_filter_stop - c.pre_filter_start);
          if (b < n.nb1) g[b] = (float)gg;
        }
        for (int b = c.pre_filter_stop; b < n.nb1; ++b) g[b] = 0.f;
      } else {
        double gp = 1.0;
        for (int b = c.pre_filter_start + 1; b < c.pre_filter_stop; ++b) {
          const double gg = pow(10.0, -(b - c.pre_filter_start) * (3.5 - gp) / 20.0);
          gp = gg;
          if (b < n.nb1) g[b] = (float)gg;
        }
      }
    }
    CHK(ht_up(n.gain_ana, g));
  }
  int max_lds_f = 0, max_lds_i = 0;
  for (auto &B : n.band) {
    max_lds_f = std::max(max_lds_f, (int)stft_lds(B.plan));
    max_lds_i = std::max(max_lds_i, (int)istft_lds(B.plan));
  }
  (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&vr_stft_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, max_lds_f);
  (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&vr_istft_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, max_lds_i);
  n.ready = true;
  return ASX_OK;
}

// ---- net ----------------------------------------------------------------------------------------------------------
static int vr_conv(asx_engine *e, const VrConv &c, const float *x, int ldc, int64_t x_bs, int B, int H, int W, int stride, float *y,
                   int ldy, int64_t y_bs, hipStream_t s) {
  HtGeom g;
  g.O = H;
  g.I = W;
  g.Cin = c.cin_p;
  g.ldc = ldc;
  g.KO = c.k;
  g.KI = c.k;
  g.PO = c.k / 2;
  g.PI = c.k / 2;
  g.SO = stride;
  g.SI = stride;
  g.OR = H / stride;
  g.IR = W / stride;
  g.x_bs = x_bs;
  g.y_bs = y_bs;
  return ht_gg(e, c.g, x, g, (int64_t)B * g.OR, y, ldy, GG_DENSE, c.act, nullptr, 0, 0, 0, 0, s);
}

template <class K, class... A>
static int vr_ew(asx_engine *e, hipStream_t s, int64_t total, double bytes, K kern, A... args) {
  return timed(e, ASX_PROF_MISC, 0.0, bytes, s, [&]() {
    hipLaunchKernelGGL(kern, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, args..., total);
  });
}

// BaseASPPNet.__call__ (nets.py:46-62): x view [B, H, W, nin_p] -> y view [B, H, W, ch]
static int vr_base(asx_engine *e, const VrBase &bs, const float *x, int x_ld, int64_t x_bs, int B, int H, int W, float *y, int y_ld,
                   int64_t y_bs, hipStream_t s) {
  VrNet &n = *e->vr;
  auto &b = n.b;
  const int L = bs.n_enc;
  const float *in = x;
  int in_ld = x_ld;
  int64_t in_bs = x_bs;
  int h = H, w = W;
  for (int i = 0; i < L; ++i) {
    const int c = bs.C[i];
    float *D = b.D[i];
    CHK(vr_conv(e, bs.enc1[i], in, in_ld, in_bs, B, h, w, 1, D + 2 * c, 3 * c, 0, s));       // skip -> its slot of the decoder concat
    CHK(vr_conv(e, bs.enc2[i], D + 2 * c, 3 * c, (int64_t)h * w * 3 * c, B, h, w, 2, b.E[i], c, 0, s));
    in = b.E[i];
    in_ld = c;
    in_bs = 0;
    h /= 2;
    w /= 2;
  }
  const int ca = bs.C[L - 1];
  const float *E = b.E[L - 1];
  const int catc = bs.nb * ca;
  // ASPP (layers.py:254-288)
  CHK(vr_ew(e, s, (int64_t)B * w * ca, 4.0 * B * h * w * ca, vr_rowmean_kernel, E, h, w, ca, b.pool));
  CHK(vr_conv(e, bs.aspp1, b.pool, ca, 0, B, 1, w, 1, b.pool2, ca, 0, s));
  CHK(vr_ew(e, s, (int64_t)B * h * w * ca, 4.0 * B * h * w * ca, vr_bcast_rows_kernel, (const float *)b.pool2, h, w, ca, b.cat, catc));
  CHK(vr_conv(e, bs.aspp2, E, ca, 0, B, h, w, 1, b.cat + ca, catc, 0, s));
  for (size_t j = 0; j < bs.sep.size(); ++j) {
    CHK(vr_ew(e, s, (int64_t)B * h * w * ca, 8.0 * B * h * w * ca, vr_dwconv_kernel, E, h, w, ca, bs.sep[j].dil, (const float *)bs.sep[j].dw.f(),
              b.tmp));
    CHK(vr_conv(e, bs.sep[j].pw, b.tmp, ca, 0, B, h, w, 1, b.cat + (2 + j) * ca, catc, 0, s));
  }
  CHK(vr_conv(e, bs.bott, b.cat, catc, 0, B, h, w, 1, b.bn, 2 * ca, 0, s));
  const float *prev = b.bn;
  for (int i = L - 1; i >= 0; --i) {
    const int c = bs.C[i];
    float *D = b.D[i];
    CHK(vr_ew(e, s, (int64_t)B * 2 * h * 2 * w * (2 * c / 4), 4.0 * B * 5.0 * h * w * 2 * c, vr_upsample2x_kernel, prev, h, w, 2 * c, 2 * c, D, 3 * c));
    h *= 2;
    w *= 2;
    if (i > 0) {
      CHK(vr_conv(e, bs.dec[i], D, 3 * c, 0, B, h, w, 1, b.O[i], c, 0, s));
      prev = b.O[i];
    } else {
      CHK(vr_conv(e, bs.dec[i], D, 3 * c, 0, B, h, w, 1, y, y_ld, y_bs, s));
    }
  }
  return ASX_OK;
}

static int vr_ensure_workspace(asx_engine *e, int B) {
  VrNet &n = *e->vr;
  if (B <= n.ws_batch) return ASX_OK;
  const asx_vr_config &c = n.cfg;
  const size_t P = (size_t)B * n.max_bin * c.window_size;   // positions at full resolution
  size_t off = 0;
  std::vector<std::pair<float **, size_t>> plan;
  auto want = [&](float *&p, size_t floats) {
    plan.push_back({&p, off});
    off += (floats * 4 + 255) & ~(size_t)255;
  };
  auto &b = n.b;
  const int chmax = std::max(std::max(c.cap[0], c.cap[2]), c.cap[4]);
  const int L = c.arch == 129605 ? 5 : 4;
  want(b.hc, P * n.ctot);
  want(b.y2, P * c.cap[1]);
  want(b.y3, P * c.cap[3]);
  want(b.h3, P * c.cap[4]);
  want(b.mk, P * 4);
  b.D.assign(L, nullptr);
  b.E.assign(L, nullptr);
  b.O.assign(L, nullptr);
  for (int i = 0; i < L; ++i) {
    const size_t pi = P >> (2 * i);
    want(b.D[i], pi * 3 * ((size_t)chmax << i));
    want(b.E[i], (pi / 4) * ((size_t)chmax << i));
    want(b.O[i], pi * ((size_t)chmax << i));
  }
  const size_t pa = P >> (2 * L);
  const size_t ca = (size_t)chmax << (L - 1);
  want(b.pool, (size_t)B * c.window_size * ca);
  want(b.pool2, (size_t)B * c.window_size * ca);
  want(b.tmp, pa * ca);
  want(b.cat, pa * ca * 7);
  want(b.bn, pa * 2 * ca);
  CHK(n.ws.ensure(off));
  for (auto &pr : plan) *pr.first = reinterpret_cast<float *>(reinterpret_cast<char *>(n.ws.p) + pr.second);
  n.ws_batch = B;
  return ASX_OK;
}

// CascadedASPPNet.forward (nets.py:132-161) on B patches already in hc[..., 0:4]; result (sigmoid mask) in b.mk [B, max_bin, W, 4]
static int vr_net_dev(asx_engine *e, int B, hipStream_t s) {
  VrNet &n = *e->vr;
  const asx_vr_config &c = n.cfg;
  auto &b = n.b;
  const int F = n.max_bin, W = c.window_size, ct = n.ctot, ch1 = c.cap[0];
  const int bw = F / 2;
  const int64_t hc_bs = (int64_t)F * W * ct;
  CHK(vr_base(e, n.s1l, b.hc, ct, hc_bs, B, bw, W, b.hc + 4, ct, hc_bs, s));
  CHK(vr_base(e, n.s1h, b.hc + (int64_t)bw * W * ct, ct, hc_bs, B, bw, W, b.hc + (int64_t)bw * W * ct + 4, ct, hc_bs, s));
  CHK(vr_conv(e, n.br2, b.hc, ct, 0, B, F, W, 1, b.y2, c.cap[1], 0, s));
  CHK(vr_base(e, n.s2, b.y2, c.cap[1], 0, B, F, W, b.hc + 4 + ch1, ct, hc_bs, s));
  CHK(vr_conv(e, n.br3, b.hc, ct, 0, B, F, W, 1, b.y3, c.cap[3], 0, s));
  CHK(vr_base(e, n.s3, b.y3, c.cap[3], 0, B, F, W, b.h3, c.cap[4], 0, s));
  return vr_conv(e, n.outc, b.h3, c.cap[4], 0, B, F, W, 1, b.mk, 4, 0, s);
}

static double vr_flops_patch(const asx_engine *e) {
  const VrNet &n = *e->vr;
  const asx_vr_config &c = n.cfg;
  auto base = [&](const VrBase &bs, double P) {
    double f = 0.0;
    double p = P;
    for (int i = 0; i < bs.n_enc; ++i) {
      f += 2.0 * p * bs.enc1[i].g.n * bs.enc1[i].g.k + 2.0 * (p / 4) * bs.enc2[i].g.n * bs.enc2[i].g.k + 2.0 * p * bs.dec[i].g.n * bs.dec[i].g.k;
      p /= 4;
    }
    f += 2.0 * p * (bs.aspp2.g.n * (double)bs.aspp2.g.k * (1 + bs.sep.size()) + bs.bott.g.n * (double)bs.bott.g.k);
    return f;
  };
  const double P = (double)n.max_bin * c.window_size;
  return base(n.s1l, P / 2) + base(n.s1h, P / 2) + base(n.s2, P) + base(n.s3, P) +
         2.0 * P * (n.br2.g.n * (double)n.br2.g.k + n.br3.g.n * (double)n.br3.g.k + 2.0 * n.outc.g.k);
}

// ---- signal chain ---------------------------------------------------------------------------------------------------
static int vr_resample(asx_engine *e, const VrFilt &f, const float *x, int64_t n_in, float *y, int64_t n_out, int acc64, hipStream_t s) {
  return timed(e, ASX_PROF_MISC, 0.0, 8.0 * (n_in + n_out), s, [&]() {
    hipLaunchKernelGGL(vr_resample_kernel, dim3((unsigned)((n_out + 255) / 256), 2), dim3(256), 0, s, x, n_in, f.h32.f(),
                       reinterpret_cast<const double *>(f.h64.p), f.hlen, f.up, f.down, f.n_pre_remove, y, n_out, acc64);
  });
}

static int64_t vr_resampled_len(const VrFilt &f, int64_t n_in) {   // ceil(n * ratio) == resample_poly's n_out
  const int64_t t = n_in * f.up;
  return t / f.down + (t % f.down ? 1 : 0);
}

// frames of the combined spectrogram and output length for an input of n samples at the top band's rate
static int vr_plan(const VrNet &n, int64_t n_samples, int *T, int64_t *n_out) {
  const int NB = n.cfg.n_bands;
  int64_t len = n_samples;
  int Tmin = 0;
  for (int d = NB - 1; d >= 0; --d) {
    if (d < NB - 1) len = vr_resampled_len(n.band[d].ana, len);
    const int t = (int)(1 + len / n.band[d].b.hl);
    Tmin = d == NB - 1 ? t : std::min(Tmin, t);
  }
  *T = Tmin;
  *n_out = (int64_t)n.band[NB - 1].b.hl * (Tmin - 1);
  return ASX_OK;
}

// loading_mix (vr_separator.py:255-291): wave [2, n] float32 at band[N].sr -> X [2, T, bins+1] complex64
static int vr_analysis_dev(asx_engine *e, const float *wave, int64_t n_samples, int T, hipStream_t s) {
  VrNet &n = *e->vr;
  const int NB = n.cfg.n_bands;
  n.wav_ana.resize(NB);
  CHK(n.X.ensure((size_t)2 * T * n.nb1 * 8));
  HIPCHK(hipMemsetAsync(n.X.p, 0, (size_t)2 * T * n.nb1 * 8, s));
  const float *cur = wave;
  int64_t len = n_samples;
  std::vector<int> row_off(NB, 0);
  for (int d = 1; d < NB; ++d) row_off[d] = row_off[d - 1] + n.band[d - 1].b.crop_stop - n.band[d - 1].b.crop_start;
  for (int d = NB - 1; d >= 0; --d) {
    VrBand &B = n.band[d];
    if (d < NB - 1 && !(B.ana.up == 1 && B.ana.down == 1)) {
      const int64_t lo = vr_resampled_len(B.ana, len);
      CHK(n.wav_ana[d].ensure((size_t)2 * lo * 4));
      CHK(vr_resample(e, B.ana, cur, len, n.wav_ana[d].f(), lo, 0, s));
      cur = n.wav_ana[d].f();
      len = lo;
    }
    CHK(timed(e, ASX_PROF_STFT, 0.0, 8.0 * len + 16.0 * T * (B.b.crop_stop - B.b.crop_start), s, [&]() {
      hipLaunchKernelGGL(vr_stft_kernel, dim3(T, 2), dim3(256), stft_lds(B.plan), s, cur, len, B.b.hl, n.cfg.channel_mode,
                         B.b.crop_start, B.b.crop_stop, row_off[d], n.nb1, n.gain_ana.f(), reinterpret_cast<float2 *>(n.X.p),
                         B.window.f(), reinterpret_cast<const float2 *>(B.tw.p), B.plan);
    }));
  }
  return ASX_OK;
}

// one pass of inference_vr._execute over all patches (vr_separator.py:294-327)
static int vr_mask_pass(asx_engine *e, int T, int pad_l, int shift, int patches, int tta, float *M, hipStream_t s) {
  VrNet &n = *e->vr;
  const asx_vr_config &c = n.cfg;
  const int W = c.window_size, roi = W - 2 * c.offset;
  const int maxB = c.max_batch > 0 ? c.max_batch : 4;
  CHK(vr_ensure_workspace(e, std::min(maxB, patches)));
  for (int k0 = 0; k0 < patches; k0 += maxB) {
    const int B = std::min(maxB, patches - k0);
    CHK(vr_ew(e, s, (int64_t)B * n.max_bin * W, 16.0 * B * n.max_bin * W, vr_patch_kernel, reinterpret_cast<const float2 *>(n.X.p), T, n.nb1,
              n.max_bin, W, k0, roi, pad_l, reinterpret_cast<const unsigned int *>(n.peak.p), n.b.hc, n.ctot));
    CHK(vr_net_dev(e, B, s));
    CHK(vr_ew(e, s, (int64_t)B * roi * n.nb1, 16.0 * B * roi * n.nb1, vr_mask_kernel, (const float *)n.b.mk, B, n.max_bin, W, c.offset, k0, roi,
              shift, T, n.nb1, tta, M));
  }
  return ASX_OK;
}

// weight of merge_artifacts (spec_utils.py:187-211) from the per-frame minimum of the mask; false = the reference's
// try block would have raised (mask left unchanged)
static bool vr_artifact_weight(const std::vector<float> &fmin, float thres, std::vector<float> &weight) {
  const int T = (int)fmin.size(), min_range = 64, fade = 32;
  std::vector<int> idx;
  for (int t = 0; t < T; ++t)
    if (fmin[t] > thres) idx.push_back(t);
  if (idx.empty()) return false;   // idx[0] raises IndexError
  std::vector<int> st{idx[0]}, en;
  for (size_t i = 1; i < idx.size(); ++i)
    if (idx[i] - idx[i - 1] != 1) {
      en.push_back(idx[i - 1]);
      st.push_back(idx[i]);
    }
  en.push_back(idx.back());
  weight.assign(T, 0.f);
  bool have_old = false;
  int old_e = 0;
  auto lin = [&](int i, bool up) { return (float)(up ? (double)i / (fade - 1) : 1.0 - (double)i / (fade - 1)); };
  auto put = [&](int a, int b, auto f) -> bool {   // numpy slice assignment weight[a:b] = vec(len fade or scalar)
    // negative indices wrap like numpy; a broadcast mismatch raises in the reference
    int aa = a < 0 ? a + T : a, bb = b < 0 ? b + T : b;
    aa = std::min(std::max(aa, 0), T);
    bb = std::min(std::max(bb, 0), T);
    return f(aa, bb);
  };
  for (size_t r = 0; r < st.size(); ++r) {
    if (!(en[r] - st[r] > min_range)) continue;
    int s0 = st[r], e0 = en[r];
    if (have_old && s0 - old_e < fade) s0 = old_e - fade * 2;
    if (s0 != 0) {
      if (!put(s0, s0 + fade, [&](int a, int b) {
            if (b - a != fade) return b - a <= 0 ? true : false;
            for (int i = 0; i < fade; ++i) weight[a + i] = lin(i, true);
            return true;
          }))
        return false;
    } else {
      s0 -= fade;
    }
    if (e0 != T) {
      if (!put(e0 - fade, e0, [&](int a, int b) {
            if (b - a != fade) return b - a <= 0 ? true : false;
            for (int i = 0; i < fade; ++i) weight[a + i] = lin(i, false);
            return true;
          }))
        return false;
    } else {
      e0 += fade;
    }
    put(s0 + fade, e0 - fade, [&](int a, int b) {
      for (int i = a; i < b; ++i) weight[i] = 1.f;
      return true;
    });
    old_e = e0;
    have_old = true;
  }
  return true;
}

// cmb_spectrogram_to_wave (spec_utils.py:341-396) of y_spec (which = 0) or v_spec (which = 1) -> out [2, n_out]
static int vr_synthesis_dev(asx_engine *e, int which, const float *M, int T, float *out, int64_t n_out, hipStream_t s) {
  VrNet &n = *e->vr;
  const int NB = n.cfg.n_bands;
  n.wav_syn.resize(NB);
  n.wav_up.resize(NB);
  int row_off = 0;
  const float *lower = nullptr;
  int64_t lower_len = 0;
  for (int d = 0; d < NB; ++d) {
    VrBand &B = n.band[d];
    const int nf = B.b.n_fft, hop = B.b.hl;
    const int64_t len = (int64_t)hop * (T - 1);
    if (lower != nullptr && lower_len != len) {
      set_err("band %d: the resampled lower bands have %lld samples, this band %lld (hl / sr ratios must agree)", d + 1,
              (long long)lower_len, (long long)len);
      return ASX_ERR_INVALID;
    }
    CHK(n.frames.ensure((size_t)2 * T * nf * 4));
    // squared-window sum over the T frames (librosa.filters.window_sumsquare)
    std::vector<float> w;
    host_window(nf, w);
    std::vector<double> ss((size_t)nf + (size_t)hop * (T - 1), 0.0);
    for (int t = 0; t < T; ++t)
      for (int k = 0; k < nf; ++k) ss[(size_t)t * hop + k] += (double)w[k] * (double)w[k];
    std::vector<float> ssf(ss.begin(), ss.end());
    CHK(n.wss.ensure(ssf.size() * 4));
    HIPCHK(hipMemcpyAsync(n.wss.p, ssf.data(), ssf.size() * 4, hipMemcpyHostToDevice, s));
    HIPCHK(hipStreamSynchronize(s));
    CHK(timed(e, ASX_PROF_ISTFT, 0.0, 4.0 * 2 * T * (3.0 * (B.b.crop_stop - B.b.crop_start) + nf), s, [&]() {
      hipLaunchKernelGGL(vr_istft_kernel, dim3(T, 2), dim3(256), istft_lds(B.plan), s, reinterpret_cast<const float2 *>(n.X.p), M, which,
                         n.nb1, B.b.crop_start, B.b.crop_stop, row_off, B.gain_syn.f(), n.frames.f(), B.window.f(),
                         reinterpret_cast<const float2 *>(B.tw.p), B.plan);
    }));
    row_off += B.b.crop_stop - B.b.crop_start;
    float *dst = d == NB - 1 ? out : nullptr;
    if (!dst) {
      CHK(n.wav_syn[d].ensure((size_t)2 * len * 4));
      dst = n.wav_syn[d].f();
    }
    CHK(timed(e, ASX_PROF_OLA, 0.0, 4.0 * 2 * (T * (double)nf + 2.0 * len), s, [&]() {
      hipLaunchKernelGGL(vr_ola_kernel, dim3((unsigned)((len + 255) / 256)), dim3(256), 0, s, n.frames.f(), n.wss.f(), nf, hop, T, len,
                         n.cfg.channel_mode, lower, dst);
    }));
    if (d < NB - 1) {
      if (B.syn.up == 1 && B.syn.down == 1) {
        lower = dst;
        lower_len = len;
      } else {
        lower_len = vr_resampled_len(B.syn, len);
        CHK(n.wav_up[d].ensure((size_t)2 * lower_len * 4));
        CHK(vr_resample(e, B.syn, dst, len, n.wav_up[d].f(), lower_len, 1, s));
        lower = n.wav_up[d].f();
      }
    }
  }
  (void)n_out;
  return ASX_OK;
}

// VRSeparator.separate on arrays (vr_separator.py:168-236): wave [2, n] -> primary [2, n_out], secondary [2, n_out]
static int vr_separate_dev(asx_engine *e, const float *wave, int64_t n_samples, const asx_vr_params *pr, float *primary, float *secondary,
                           hipStream_t s) {
  VrNet &n = *e->vr;
  const asx_vr_config &c = n.cfg;
  int T;
  int64_t n_out;
  CHK(vr_plan(n, n_samples, &T, &n_out));
  REQUIRE(T >= 2, "input too short: %d frames", T);
  CHK(vr_analysis_dev(e, wave, n_samples, T, s));
  CHK(n.peak.ensure(4));
  HIPCHK(hipMemsetAsync(n.peak.p, 0, 4, s));
  const int64_t nx = (int64_t)2 * T * n.nb1;
  hipLaunchKernelGGL(vr_absmax_kernel, dim3(512), dim3(256), 0, s, reinterpret_cast<const float2 *>(n.X.p), nx,
                     reinterpret_cast<unsigned int *>(n.peak.p));
  HIPCHK(hipGetLastError());
  // make_padding (spec_utils.py:86-96)
  const int W = c.window_size;
  int roi = W - 2 * c.offset;
  if (roi == 0) roi = W;
  const int patches = T / roi + 1;
  CHK(n.M.ensure((size_t)nx * 4));
  CHK(vr_mask_pass(e, T, c.offset, 0, patches, 0, n.M.f(), s));
  if (pr->enable_tta) CHK(vr_mask_pass(e, T, c.offset + roi / 2, roi / 2, patches + 1, 1, n.M.f(), s));
  // adjust_aggr (spec_utils.py:472-492)
  double aggr = (double)pr->aggr_value * 2.0;
  if (aggr != 0.0) {
    if (pr->is_non_accom) aggr = 1.0 - aggr;
    double a0 = aggr, a1 = aggr;
    if (pr->has_corr) {
      a0 += pr->corr_left;
      a1 += pr->corr_right;
    }
    CHK(vr_ew(e, s, nx, 8.0 * nx, vr_aggr_kernel, n.M.f(), T, n.nb1, (int)pr->split_bin, (float)(1.0 + a0 / 3.0), (float)(1.0 + a0),
              (float)(1.0 + a1 / 3.0), (float)(1.0 + a1)));
  }
  if (pr->enable_post_process) {
    CHK(n.fmin.ensure((size_t)T * 4));
    CHK(n.wgt.ensure((size_t)T * 4));
    hipLaunchKernelGGL(vr_frame_min_kernel, dim3((T + 3) / 4), dim3(256), 0, s, n.M.f(), T, n.nb1, n.fmin.f());
    HIPCHK(hipGetLastError());
    std::vector<float> fmin(T), weight;
    HIPCHK(hipMemcpyAsync(fmin.data(), n.fmin.p, (size_t)T * 4, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    if (vr_artifact_weight(fmin, pr->post_thres, weight)) {
      HIPCHK(hipMemcpyAsync(n.wgt.p, weight.data(), (size_t)T * 4, hipMemcpyHostToDevice, s));
      HIPCHK(hipStreamSynchronize(s));
      CHK(vr_ew(e, s, nx, 8.0 * nx, vr_merge_kernel, n.M.f(), T, n.nb1, (const float *)n.wgt.f()));
    }
  }
  if (primary) CHK(vr_synthesis_dev(e, 0, n.M.f(), T, primary, n_out, s));
  if (secondary) CHK(vr_synthesis_dev(e, 1, n.M.f(), T, secondary, n_out, s));
  return ASX_OK;
}
