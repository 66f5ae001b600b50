// The MDX / ConvTDFNet path of libasx.so (included by asx.hip only, after engine_core.h): kernel launchers and weight packing of the
// convolution families (direct, three Winograd generations on the fp32 pipe, the bf16 x 6 Winograd), the row-GEMM launchers (tdf2 /
// tdf3 with the per-engine split-image cache) that the sibling engines share, the STFT / iSTFT / fold launches, the GroupNorm pass
// of the 'adamw' net variant, the net forward and the workspace.  Split out of asx.hip in round 5.
#pragma once

// ----------------------------------------------------------------------------
// kernel launchers
// ----------------------------------------------------------------------------
template <class CFG>
static void launch_conv_t(const ConvArgs &a, int nblk, hipStream_t s) {
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&conv_mfma_kernel<CFG>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, CFG::LDS_BYTES);
    attr_done = true;
  }
  hipLaunchKernelGGL(conv_mfma_kernel<CFG>, dim3(nblk), dim3(256), CFG::LDS_BYTES, s, a);
}

static int conv_tile_h(int kind) { return kind == CK_UP ? 4 : 8; }

static int pick_nrep_conv(int cout) {
  const int ct = (cout + 15) / 16;
  int best = 1, best_pad = 1 << 30;
  for (int n = 3; n >= 1; --n) {
    const int pad = ((ct + n - 1) / n) * n - ct;
    if (pad < best_pad) {
      best_pad = pad;
      best = n;
    }
  }
  return best;
}
static int pick_nrep_up(int cout) {
  const int ct = (cout + 15) / 16;  // virtual tiles = 4*ct, pairs must stay together
  static const int force = getenv("ASX_UP_NREP") ? atoi(getenv("ASX_UP_NREP")) : 0;   // A/B: 4 = four virtual tiles per workgroup wherever they divide
  if (force == 4 && (4 * ct) % 4 == 0) return 4;
  if ((4 * ct) % 6 == 0) return 6;
  if ((4 * ct) % 4 == 0) return 4;
  return 2;
}

static int conv_setup(ConvLayer &L, int kind, int cin, int cout, int relu) {
  L.kind = kind;
  L.cin = cin;
  L.cout = cout;
  L.relu = relu;
  if (kind == CK_UP) {
    L.nrep = pick_nrep_up(cout);
    L.kc = 8;
    const int vt = 4 * ((cout + 15) / 16);
    L.cg = (vt + L.nrep - 1) / L.nrep;
  } else {
    L.nrep = pick_nrep_conv(cout);
    L.kc = (kind == CK_3X3) ? 8 : (kind == CK_DOWN ? 4 : (cin >= 16 ? 16 : 4));
    // 3x3 convs stage FOUR channels at a time (37 KB of LDS instead of 75 KB at eight): a third co-resident workgroup
    // per CU covers the prologue / epilogue / barrier gaps of the other two.  Measured on the bench configuration: 3x3 class
    // 234.9 -> 224.0 ms (85.1 -> 89.4 % of the fp32-MFMA peak), every level gains.  ASX_CONV_KC4=<cin threshold> for the
    // A/B (0 = eight channels everywhere).
    static const int kc4 = getenv("ASX_CONV_KC4") ? atoi(getenv("ASX_CONV_KC4")) : (1 << 30);
    // (also for the two-tile channel groups of the MDX23C widths: ASX_CONV_KC4_N2=0 keeps their eight-channel stages)
    static const int kc4_n2 = getenv("ASX_CONV_KC4_N2") ? atoi(getenv("ASX_CONV_KC4_N2")) : 1;
    if (kind == CK_3X3 && cin <= kc4 && cin % 4 == 0 && (pick_nrep_conv(cout) == 3 || (kc4_n2 && pick_nrep_conv(cout) == 2))) L.kc = 4;
    // 2x2 / stride-2 conv: two-channel stages (four workgroups per CU); ASX_DOWN_KC2=0 keeps the four-channel stages
    static const int down_kc2 = getenv("ASX_DOWN_KC2") ? atoi(getenv("ASX_DOWN_KC2")) : 1;
    if (kind == CK_DOWN && down_kc2 && cin % 2 == 0 && pick_nrep_conv(cout) == 3) L.kc = 2;
    L.cg = ((cout + 15) / 16 + L.nrep - 1) / L.nrep;
  }
  L.nci = (cin + L.kc - 1) / L.kc;
  return ASX_OK;
}

// w layouts: 3x3 [cout,cin,3,3]; down [cout,cin,2,2]; 1x1 [cout,cin]; up [cin,cout,2,2]
// packed: [CG][NCI][WSTAGE] with WSTAGE = roundup(ntap*kc*NWP, 256) floats; row (tap, kc) holds NWP floats
// wino_mode: the engine's "winograd" option at load time.  The image of the default kernel (3) is always built -- the option may be
// switched between 0 and 3 on a loaded net (bench.py's direct_kernel leg does) -- the images of the earlier generations (1, 2) only
// when the option selects them before the weights arrive.
static int conv_pack(ConvLayer &L, const float *w, const float *b, int wino_mode = 3) {
  const int ntap = L.kind == CK_3X3 ? 9 : (L.kind == CK_DOWN ? 4 : 1);
  const int NW = 16 * L.nrep;
  const int NWP = (L.nrep % 2 == 0) ? NW + 16 : NW;
  const size_t wstage = (((size_t)ntap * L.kc * NWP + 255) / 256) * 256;
  const size_t per_cg = (size_t)L.nci * wstage;
  std::vector<float> wp(per_cg * L.cg, 0.f);
  const int CT = (L.cout + 15) / 16;
  for (int cg = 0; cg < L.cg; ++cg)
    for (int ci = 0; ci < L.nci; ++ci)
      for (int tap = 0; tap < ntap; ++tap)
        for (int kc = 0; kc < L.kc; ++kc) {
          const int c = ci * L.kc + kc;
          if (c >= L.cin) continue;
          float *dst = &wp[cg * per_cg + ci * wstage + ((size_t)tap * L.kc + kc) * NWP];
          for (int n = 0; n < NW; ++n) {
            float v = 0.f;
            if (L.kind == CK_UP) {
              const int nt = cg * L.nrep + n / 16;
              const int pair = nt / 2, dx = nt & 1;
              const int dy = pair / CT, ct = pair % CT;
              const int co = ct * 16 + (n & 15);
              if (dy < 2 && co < L.cout) v = w[(((size_t)c * L.cout + co) * 2 + dy) * 2 + dx];
            } else {
              const int co = cg * NW + n;
              if (co < L.cout) v = w[((size_t)co * L.cin + c) * ntap + tap];
            }
            dst[n] = v;
          }
        }
  if (L.kind == CK_3X3) {
    // U = G g G^T in float64, laid out [cg48][ci8][xi][pair][cout 48][2]
    static const double G[4][3] = {{1, 0, 0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0, 0, 1}};
    L.wu_cg = (L.cout + 47) / 48;
    L.wu_nci = (L.cin + 7) / 8;
    L.wu3_nci = ((L.cin + 7) / 8) * 2;     // four-channel stage images, an even number of them (zero padded): the kernel's 8-channel build reads pairs
    std::vector<float> wu((size_t)L.wu_cg * L.wu_nci * WinoCfg::USTAGE, 0.f), wu2(wu.size(), 0.f);
    std::vector<float> wu3((size_t)L.wu_cg * L.wu3_nci * Wino3Cfg::USTAGE, 0.f);
    for (int co = 0; co < L.cout; ++co)
      for (int c = 0; c < L.cin; ++c) {
        const float *g = &w[((size_t)co * L.cin + c) * 9];
        double t[4][3], U[4][4];
        for (int a = 0; a < 4; ++a)
          for (int j = 0; j < 3; ++j) t[a][j] = G[a][0] * g[0 * 3 + j] + G[a][1] * g[1 * 3 + j] + G[a][2] * g[2 * 3 + j];
        for (int a = 0; a < 4; ++a)
          for (int bb = 0; bb < 4; ++bb) U[a][bb] = t[a][0] * G[bb][0] + t[a][1] * G[bb][1] + t[a][2] * G[bb][2];
        const int cgi = co / 48, col = co % 48, ci = c / 8, pair = (c % 8) / 2, e = c & 1;
        float *dst = &wu[((size_t)cgi * L.wu_nci + ci) * WinoCfg::USTAGE];
        float *dst2 = &wu2[((size_t)cgi * L.wu_nci + ci) * WinoCfg::USTAGE];
        float *dst3 = &wu3[((size_t)cgi * L.wu3_nci + c / 4) * Wino3Cfg::USTAGE + ((size_t)(c % 4) * 16 + col % 16) * Wino3Cfg::ULS];
        for (int a = 0; a < 4; ++a)
          for (int bb = 0; bb < 4; ++bb) {
            dst[(((a * 4 + bb) * 4 + pair) * 48 + col) * 2 + e] = (float)U[a][bb];
            dst2[((a * 4 + bb) * 8 + (c % 8)) * 48 + col] = (float)U[a][bb];
            dst3[(a * 4 + bb) * 3 + col / 16] = (float)U[a][bb];
          }
      }
#ifdef ASX_EXPERIMENTAL_KERNELS
    if (wino_mode == 1) {
      CHK(L.wu.ensure(wu.size() * 4));
      HIPCHK(hipMemcpy(L.wu.p, wu.data(), wu.size() * 4, hipMemcpyHostToDevice));
    }
    if (wino_mode == 2) {
      CHK(L.wu2.ensure(wu2.size() * 4));
      HIPCHK(hipMemcpy(L.wu2.p, wu2.data(), wu2.size() * 4, hipMemcpyHostToDevice));
    }
#else
    (void)wino_mode;
#endif
    CHK(L.wu3.ensure(wu3.size() * 4));
    HIPCHK(hipMemcpy(L.wu3.p, wu3.data(), wu3.size() * 4, hipMemcpyHostToDevice));
    // bf16 x 6 image (kernels_wino6.h) for every layer wide enough to be worth a 32-channel stage; which layers RUN on it is the
    // engine's "winograd_bf16x6" option at launch time
    L.wu6_nci = 0;
    if (L.cin >= 64) {
      std::vector<uint32_t> w6;
      int cg6 = 0, nci6 = 0;
      wino6_pack(w, L.cout, L.cin, w6, &cg6, &nci6);
      if (cg6 == L.wu_cg) {
        CHK(L.wu6.ensure(w6.size() * 4));
        HIPCHK(hipMemcpy(L.wu6.p, w6.data(), w6.size() * 4, hipMemcpyHostToDevice));
        L.wu6_nci = nci6;
        wino6_pack_h(w, L.cout, L.cin, w6, &cg6, &nci6);   // which of the two images a launch reads is the engine's "gemm_f16x3" option then
        CHK(L.wu6h.ensure(w6.size() * 4));
        HIPCHK(hipMemcpy(L.wu6h.p, w6.data(), w6.size() * 4, hipMemcpyHostToDevice));
      }
    }
    // direct fp16 x 3 kernel (kernels_conv3h.h): layers of 48 n -> 48 n channels as n x n images of 48 x 48 (output slice major)
    if (L.cin == L.cout && L.cin % Conv3hCfg::C == 0 && L.cin <= 3 * Conv3hCfg::C) {   // (up to 144 channels: on the deeper levels the n x n slice launches lose to conv_wino6_kernel -- 1.95 / 1.13 / 1.45 against 1.59 / 0.61 / 0.21 ms at 192 / 240 / 288 channels, profiles/r06_conv3h_deeper_levels.txt)
      const int nb = L.cin / Conv3hCfg::C;
      std::vector<uint32_t> all, one;
      std::vector<float> ws((size_t)Conv3hCfg::C * Conv3hCfg::C * 9);
      for (int ob = 0; ob < nb; ++ob)
        for (int ib = 0; ib < nb; ++ib) {
          for (int co = 0; co < Conv3hCfg::C; ++co)
            memcpy(&ws[(size_t)co * Conv3hCfg::C * 9], &w[(((size_t)(ob * Conv3hCfg::C + co)) * L.cin + (size_t)ib * Conv3hCfg::C) * 9], (size_t)Conv3hCfg::C * 9 * 4);
          conv3h_pack(ws.data(), one);
          all.insert(all.end(), one.begin(), one.end());
        }
      CHK(L.w3h.ensure(all.size() * 4));
      HIPCHK(hipMemcpy(L.w3h.p, all.data(), all.size() * 4, hipMemcpyHostToDevice));
    }
    // weight-stationary image: only where all of U fits the registers of eight waves (Cin <= 96) without much zero padding
    L.wus_ks = 0;
#ifdef ASX_EXPERIMENTAL_KERNELS
    L.wus_ks = (L.cin > 40 && L.cin <= 48) ? 12 : ((L.cin > 88 && L.cin <= 96) ? 24 : 0);
#endif
#ifdef ASX_EXPERIMENTAL_KERNELS
    if (L.wus_ks) {
      const size_t per = (size_t)8 * (6 * L.wus_ks / 4) * 64 * 4;
      std::vector<float> wus((size_t)L.wu_cg * per, 0.f);
      for (int co = 0; co < L.cout; ++co)
        for (int c = 0; c < L.cin; ++c) {
          const float *g = &w[((size_t)co * L.cin + c) * 9];
          double t[4][3];
          for (int a = 0; a < 4; ++a)
            for (int j = 0; j < 3; ++j) t[a][j] = G[a][0] * g[0 * 3 + j] + G[a][1] * g[1 * 3 + j] + G[a][2] * g[2 * 3 + j];
          float *dst = &wus[(size_t)(co / 48) * per];
          for (int a = 0; a < 4; ++a)
            for (int bb = 0; bb < 4; ++bb) {
              const float u = (float)(t[a][0] * G[bb][0] + t[a][1] * G[bb][1] + t[a][2] * G[bb][2]);
              dst[L.wus_ks == 12 ? winos_u_index<12>(a, bb, c, co % 48) : winos_u_index<24>(a, bb, c, co % 48)] = u;
            }
        }
      CHK(L.wus.ensure(wus.size() * 4));
      HIPCHK(hipMemcpy(L.wus.p, wus.data(), wus.size() * 4, hipMemcpyHostToDevice));
    }
#endif
  }
  if (L.kind == CK_DOWN) {
    // bf16 x 6 image of the 2 x 2 / stride-2 conv (kernels_updown6.h); which kernel RUNS is the engine's "gemm_bf16x6" option at launch time
    std::vector<uint32_t> w6;
    static const int wide = getenv("ASX_DOWN6_WIDE") ? atoi(getenv("ASX_DOWN6_WIDE")) : 1;   // A/B: 96-channel workgroups where Cout % 96 == 0
    L.wd6_nrep = (wide && L.cout % 96 == 0) ? 6 : 3;
    if (L.wd6_nrep == 6) down6_pack<6>(w, L.cout, L.cin, w6, &L.wd6_cg, &L.wd6_nst);
    else down6_pack<3>(w, L.cout, L.cin, w6, &L.wd6_cg, &L.wd6_nst);
    CHK(L.wd6.ensure(w6.size() * 4));
    HIPCHK(hipMemcpy(L.wd6.p, w6.data(), w6.size() * 4, hipMemcpyHostToDevice));
  }
  if (L.kind == CK_UP) {
    // bf16 x 6 image of the transposed conv (kernels_updown6.h), same virtual-tile grouping as the fp32 image (L.nrep tiles per workgroup)
    std::vector<uint32_t> w6;
    if (L.nrep == 6) up6_pack<6>(w, L.cout, L.cin, w6, &L.wup6_cg, &L.wup6_nst);
    else if (L.nrep == 4) up6_pack<4>(w, L.cout, L.cin, w6, &L.wup6_cg, &L.wup6_nst);
    else up6_pack<2>(w, L.cout, L.cin, w6, &L.wup6_cg, &L.wup6_nst);
    CHK(L.wup6.ensure(w6.size() * 4));
    HIPCHK(hipMemcpy(L.wup6.p, w6.data(), w6.size() * 4, hipMemcpyHostToDevice));
  }
  const int nb = (L.kind == CK_UP) ? CT * 16 : std::max(L.cg * NW, ((L.cout + 47) / 48) * 48);
  std::vector<float> bp(nb, 0.f);
  for (int i = 0; i < L.cout; ++i) bp[i] = b ? b[i] : 0.f;
  CHK(L.w.ensure(wp.size() * 4));
  CHK(L.b.ensure(bp.size() * 4));
  HIPCHK(hipMemcpy(L.w.p, wp.data(), wp.size() * 4, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(L.b.p, bp.data(), bp.size() * 4, hipMemcpyHostToDevice));
  return ASX_OK;
}

template <class CFG>
static void launch_conv_dma_t(const ConvArgs &a, int nblk, hipStream_t s) {
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&conv_dma_kernel<CFG>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, CFG::LDS_BYTES);
    attr_done = true;
  }
  hipLaunchKernelGGL(conv_dma_kernel<CFG>, dim3(nblk), dim3(256), CFG::LDS_BYTES, s, a);
}

static std::atomic<long long> g_wino6_launches{0};   // launches of conv_wino6_kernel (kernels_wino6.h) since the process started
static std::atomic<long long> g_wino6h_launches{0};  // ... of which on the fp16 x 3 arithmetic
static std::atomic<long long> g_conv3h_launches{0};  // launches of conv3h_kernel (kernels_conv3h.h)
static std::atomic<long long> g_down6_launches{0};   // launches of conv_down6_kernel (kernels_updown6.h)
static std::atomic<long long> g_up6_launches{0};     // launches of conv_up6_kernel

// Optional view description of a conv's operands (channel slices of larger buffers).
struct ConvView {
  int64_t x_bstride = 0;    // 0 = dense [B, cin, T, F]
  int64_t y_bstride = 0;    // 0 = dense [B, cout, To, Fo]
  const float *res = nullptr;
  int64_t aux_bstride = 0;  // of skip / res; 0 = dense
  int act = -1;             // -1 = the layer's own activation
};

// x [B,cin,T,F] -> y
static int conv_launch(asx_engine *e, const ConvLayer &L, const float *x, const float *skip, float *y, int B, int T,
                       int F, hipStream_t s, const ConvView &v = ConvView()) {
  ConvArgs a{};
  a.x = x;
  a.wp = L.w.f();
  a.bias = L.b.f();
  a.skip = skip;
  a.res = v.res;
  a.zeros = e->d_zeros.f();
  a.y = y;
  a.B = B;
  a.Cin = L.cin;
  a.Cout = L.cout;
  a.T = T;
  a.F = F;
  a.act = v.act >= 0 ? v.act : (L.relu ? ACT_RELU : ACT_NONE);
  a.CG = L.cg;
  a.NCI = L.nci;
  static const int nt_mode = getenv("ASX_NT") ? atoi(getenv("ASX_NT")) : 0;   // bit 0: conv stores, bit 1: TDF stores, bit 2: TDF residual loads
  a.nt = nt_mode & 1;
  int cls = ASX_PROF_CONV3X3;
  double taps = 9;
  int64_t out_plane;
  if (L.kind == CK_DOWN) {
    a.To = T / 2;
    a.Fo = F / 2;
    cls = ASX_PROF_DOWN;
    taps = 4;
    out_plane = (int64_t)a.To * a.Fo;
  } else {
    a.To = T;
    a.Fo = F;
    out_plane = (int64_t)T * F;
    if (L.kind == CK_1X1) {
      cls = ASX_PROF_CONV1X1;
      taps = 1;
    } else if (L.kind == CK_UP) {
      cls = ASX_PROF_UP;
      taps = 4;
      out_plane = (int64_t)4 * T * F;
    }
  }
  a.x_bstride = v.x_bstride ? v.x_bstride : (int64_t)L.cin * T * F;
  a.y_bstride = v.y_bstride ? v.y_bstride : (int64_t)L.cout * out_plane;
  a.aux_bstride = v.aux_bstride ? v.aux_bstride : (int64_t)L.cout * out_plane;
  const int th = conv_tile_h(L.kind);
  a.tilesT = (a.To + th - 1) / th;
  a.tilesF = (a.Fo + 63) / 64;
  const int nblk = a.CG * a.tilesT * a.tilesF * B;
  if (nblk <= 0) return ASX_OK;
  const double outpix = (double)B * a.To * a.Fo;
  const double flops = 2.0 * taps * L.cin * L.cout * outpix;
  double bytes = 4.0 * ((double)B * L.cin * T * F + (double)L.cout * outpix * (L.kind == CK_UP ? 4 : 1));
  if (L.kind == CK_UP && skip) bytes += 4.0 * (double)L.cout * outpix * 4;  // skip read
  if (v.res) bytes += 4.0 * (double)L.cout * outpix;
  int bad = 0;
  const bool dma = (F % 4 == 0) && ((reinterpret_cast<uintptr_t>(x) & 15) == 0) && (a.x_bstride % 4 == 0) &&
                   getenv("ASX_NO_DMA") == nullptr;
  const ConvArgs &d = a;
#ifdef ASX_EXPERIMENTAL_KERNELS   // measured-and-rejected generations (profiles/NOTES.md): python build.py --experimental, or ASX_EXPERIMENTAL=1 in the environment of the build
  if (L.kind == CK_3X3 && e->winograd == 3 && e->winos == 1 && dma && L.wus.p != nullptr && a.Fo % 32 == 0 && v.res == nullptr &&
      (a.act == ACT_RELU || a.act == ACT_NONE) && (int64_t)L.cin * T * F < ((int64_t)1 << 30)) {
    // weight-stationary Winograd (kernels_winos.h): a workgroup walks a 32-pixel-wide column strip, one tile row per step
    ConvArgs wa = a;
    wa.wp = L.wus.f();
    wa.CG = L.wu_cg;
    wa.tilesF = a.Fo / 32;
    const int NS = (a.To + 1) / 2;
    const int base = wa.CG * wa.tilesF * B;
    // row blocks: enough workgroups to fill the chip a few times over, but at least 8 tile rows each (prologue: weights + 4 rows)
    int RB = std::max(1, std::min((NS + 7) / 8, (2048 + base - 1) / base));
    const int SPB = (NS + RB - 1) / RB;
    RB = (NS + SPB - 1) / SPB;
    wa.tilesT = RB;
    wa.NCI = SPB;
    const int nb = base * RB;
    auto gos = [&](auto kern, int lds) {
      {
        static std::mutex attr_mutex;
        static std::set<const void *> attr_done;
        std::lock_guard<std::mutex> lock(attr_mutex);
        if (attr_done.insert(reinterpret_cast<const void *>(kern)).second)
          (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
      }
      return timed(e, cls, flops, bytes, s, [&]() { hipLaunchKernelGGL(kern, dim3(nb), dim3(512), lds, s, wa); });
    };
    static const int abls = getenv("ASX_WINOS_ABL") ? atoi(getenv("ASX_WINOS_ABL")) : 0;   // timing probes (results invalid)
    const bool k12 = L.wus_ks == 12;
    const bool ragged = (a.To & 1) || (L.cout % 48) != 0;
    if (abls && !ragged) {
      switch (abls) {
        case 1: return k12 ? gos(&conv_winos_kernel<12, 1>, WinoSCfg<12>::LDS_BYTES) : gos(&conv_winos_kernel<24, 1>, WinoSCfg<24>::LDS_BYTES);
        case 2: return k12 ? gos(&conv_winos_kernel<12, 2>, WinoSCfg<12>::LDS_BYTES) : gos(&conv_winos_kernel<24, 2>, WinoSCfg<24>::LDS_BYTES);
        case 4: return k12 ? gos(&conv_winos_kernel<12, 4>, WinoSCfg<12>::LDS_BYTES) : gos(&conv_winos_kernel<24, 4>, WinoSCfg<24>::LDS_BYTES);
        case 5: return k12 ? gos(&conv_winos_kernel<12, 5>, WinoSCfg<12>::LDS_BYTES) : gos(&conv_winos_kernel<24, 5>, WinoSCfg<24>::LDS_BYTES);
        case 13: return k12 ? gos(&conv_winos_kernel<12, 13>, WinoSCfg<12>::LDS_BYTES) : gos(&conv_winos_kernel<24, 13>, WinoSCfg<24>::LDS_BYTES);
        default: break;
      }
    }
    if (ragged) return k12 ? gos(&conv_winos_kernel<12, 0, true>, WinoSCfg<12>::LDS_BYTES) : gos(&conv_winos_kernel<24, 0, true>, WinoSCfg<24>::LDS_BYTES);
    return k12 ? gos(&conv_winos_kernel<12, 0, false>, WinoSCfg<12>::LDS_BYTES) : gos(&conv_winos_kernel<24, 0, false>, WinoSCfg<24>::LDS_BYTES);
  }
#endif
  if (L.kind == CK_3X3 && e->winograd == 3 && e->gemm_bf16x6 > 0 && e->gemm_f16x3 > 0 && e->conv3h > 0 && L.cin <= e->conv3h && L.w3h.p != nullptr && dma &&
      v.res == nullptr && (a.act == ACT_RELU || a.act == ACT_NONE) && F % 32 == 0 && (int64_t)Conv3hCfg::C * T * F < ((int64_t)1 << 29) &&
      a.y_bstride % 4 == 0 && (reinterpret_cast<uintptr_t>(y) & 15) == 0) {
    // direct implicit GEMM on the fp16 pipe, weights resident in LDS: one persistent 512-thread workgroup per CU (the walk assumes 8 XCDs x 32).
    // A layer of 48 n channels runs as n x n launches: for every 48-channel slice of the OUTPUT, the input slices one after the other, each
    // added to the sum of those before it (a.prev = the output itself), bias with the first, activation with the last.
    const int nb = L.cin / Conv3hCfg::C;
    const int tilesT = (T + 3) / 4, tilesF = F / 32;
    // walk geometry: bands of 32 / 16 / 8 strips -- whole bands on the plane's width, charged for the extra block per segment of T and for how evenly
    // the (b, band) items deal out over the 8 XCDs (a 7-chunk strong-scaling pass has 21 items of 32 strips at level 0: 3 / 2 per XCD; 84 of 8 strips: 11 / 10)
    int bw = 32;
    double best = 1e30;
    for (int cand : {32, 16, 8}) {
      if ((tilesT * cand) % 32 != 0) continue;
      const int tps = tilesT * cand / 32;
      const int bands = (tilesF + cand - 1) / cand, items = B * bands;
      const double cost = (double)(bands * cand) / tilesF * (tps + 1.0) / tps * (double)((items + 7) / 8) / (items / 8.0);
      if (cost < best - 1e-9) {
        best = cost;
        bw = cand;
      }
    }
    if (best > 1e29) bw = 32;                          // (tilesT not divisible: one segment)
    {
      static std::mutex attr_mutex;
      static bool attr_done = false;
      std::lock_guard<std::mutex> lock(attr_mutex);
      if (!attr_done) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&conv3h_kernel<0, false>), hipFuncAttributeMaxDynamicSharedMemorySize, Conv3hCfg::LDS_BYTES);
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&conv3h_kernel<0, true>), hipFuncAttributeMaxDynamicSharedMemorySize, Conv3hCfg::LDS_BYTES);
        attr_done = true;
      }
    }
    const int64_t plane = (int64_t)T * F;
    return timed(e, cls, flops, bytes, s, [&]() {
      e->prof_nprod = 3;
      for (int ob = 0; ob < nb; ++ob)
        for (int ib = 0; ib < nb; ++ib) {
          Conv3hArgs ca{};
          ca.x = x + (int64_t)ib * Conv3hCfg::C * plane;
          ca.y = y + (int64_t)ob * Conv3hCfg::C * plane;
          ca.prev = ib > 0 ? ca.y : nullptr;
          ca.wimg = reinterpret_cast<const u32x4 *>(reinterpret_cast<const uint32_t *>(L.w3h.p) + (size_t)(ob * nb + ib) * Conv3hCfg::IMG_U32);
          ca.bias = ib == 0 ? L.b.f() + ob * Conv3hCfg::C : nullptr;
          ca.B = B;
          ca.T = T;
          ca.F = F;
          ca.x_bstride = a.x_bstride;
          ca.y_bstride = a.y_bstride;
          ca.act = ib == nb - 1 ? a.act : ACT_NONE;
          ca.tilesT = tilesT;
          ca.tilesF = tilesF;
          ca.bw = bw;
          ca.tps = (tilesT * bw) % 32 == 0 ? tilesT * bw / 32 : tilesT;
          if (ib > 0) hipLaunchKernelGGL((conv3h_kernel<0, true>), dim3(256), dim3(512), Conv3hCfg::LDS_BYTES, s, ca);
          else hipLaunchKernelGGL((conv3h_kernel<0, false>), dim3(256), dim3(512), Conv3hCfg::LDS_BYTES, s, ca);
          g_conv3h_launches.fetch_add(1);
        }
    });
  }
  if (L.kind == CK_3X3 && e->winograd == 3 && e->gemm_bf16x6 > 0 && e->wino6 > 0 && L.cin >= e->wino6 && dma && L.wu6_nci > 0 &&
      (int64_t)T * F < ((int64_t)1 << 24)) {
    // Winograd on the bf16 pipe, one workgroup per (8 x 32 tile, 48-channel group); the groups of a tile are consecutive on one XCD
    ConvArgs wa = a;
    const bool h3 = e->gemm_f16x3 > 0 && L.wu6h.p != nullptr;   // fp16 x 3 arithmetic (kernels_wino6.h: template parameter H)
    wa.wp = reinterpret_cast<const float *>(h3 ? L.wu6h.p : L.wu6.p);
    wa.CG = L.wu_cg;
    wa.NCI = L.wu6_nci;
    wa.tilesT = (a.To + Wino6Cfg::TH - 1) / Wino6Cfg::TH;
    wa.tilesF = (a.Fo + Wino6Cfg::TW - 1) / Wino6Cfg::TW;
    const int64_t S = (int64_t)wa.tilesT * wa.tilesF * B;
    const int64_t nb = ((S + 7) / 8) * 8 * wa.CG;
    if (nb < ((int64_t)1 << 31)) {
      {
        static std::mutex attr_mutex;
        static bool attr_done = false;
        std::lock_guard<std::mutex> lock(attr_mutex);
        if (!attr_done) {
          (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&conv_wino6_kernel<0, 1>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    Wino6Cfg::LDS_BYTES);
          (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&conv_wino6_kernel<0, 1, true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    Wino6Cfg::LDS_BYTES);
          attr_done = true;
        }
      }
      g_wino6_launches.fetch_add(1);
      if (h3) g_wino6h_launches.fetch_add(1);
      return timed(e, cls, flops, bytes, s, [&]() {
        e->prof_nprod = h3 ? 3 : 6;
        if (h3) hipLaunchKernelGGL((conv_wino6_kernel<0, 1, true>), dim3((unsigned)nb), dim3(512), Wino6Cfg::LDS_BYTES, s, wa);
        else hipLaunchKernelGGL((conv_wino6_kernel<0, 1>), dim3((unsigned)nb), dim3(512), Wino6Cfg::LDS_BYTES, s, wa);
      });
    }
  }
  if (L.kind == CK_3X3 && e->winograd == 3 && dma && L.wu3.p != nullptr) {
    ConvArgs wa = a;
    wa.wp = L.wu3.f();
    wa.CG = L.wu_cg;
    wa.NCI = L.wu3_nci;
    wa.tilesT = (a.To + Wino3Cfg::TH - 1) / Wino3Cfg::TH;
    wa.tilesF = (a.Fo + Wino3Cfg::TW - 1) / Wino3Cfg::TW;
    const int nb = wa.CG * wa.tilesT * wa.tilesF * B;
    auto go = [&](auto kern, int lds, int stages) {
      {
        static std::mutex attr_mutex;            // engines may be driven from several host threads (one per bag member / rank)
        static std::set<const void *> attr_done;
        std::lock_guard<std::mutex> lock(attr_mutex);
        if (attr_done.insert(reinterpret_cast<const void *>(kern)).second)
          (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
      }
      wa.NCI = stages;
      return timed(e, cls, flops, bytes, s, [&]() { hipLaunchKernelGGL(kern, dim3(nb), dim3(256), lds, s, wa); });
    };
#ifdef ASX_EXPERIMENTAL_KERNELS
    static const int abl3 = getenv("ASX_WINO_ABL") ? atoi(getenv("ASX_WINO_ABL")) : 0;   // timing probes (results invalid)
    // A/B builds: 6 (default): 4-channel stages x 2 LDS buffers, raw planes at an odd float stride; 5 / 4: rings of 3 / 4 buffers
    // (two / three stages of DMA in flight, counted vmcnt); 0 / 2: 4 / 3 buffers at the even stride; 1: 8-channel stages x 2 buffers
    static const int wcfg = getenv("ASX_WINO_CFG") ? atoi(getenv("ASX_WINO_CFG")) : 6;
    if (abl3) {
      switch (abl3) {
        case 1: return go(&conv_wino3_kernel<1>, Wino3Cfg::LDS_BYTES, L.wu3_nci);
        case 2: return go(&conv_wino3_kernel<2>, Wino3Cfg::LDS_BYTES, L.wu3_nci);
        case 4: return go(&conv_wino3_kernel<4>, Wino3Cfg::LDS_BYTES, L.wu3_nci);
        case 16: return go(&conv_wino3_kernel<16>, Wino3Cfg::LDS_BYTES, L.wu3_nci);
        default: break;
      }
    }
    if (wcfg == 1) return go(&conv_wino3_kernel<0, 8, 2>, Wino3CfgT<8, 2>::LDS_BYTES, L.wu3_nci / 2);
    if (wcfg == 2) return go(&conv_wino3_kernel<0, 4, 3>, Wino3CfgT<4, 3>::LDS_BYTES, L.wu3_nci);
    if (wcfg == 0) return go(&conv_wino3_kernel<0>, Wino3Cfg::LDS_BYTES, L.wu3_nci);
    if (wcfg == 4) return go(&conv_wino3_kernel<0, 4, 4, 1>, Wino3CfgT<4, 4, 1>::LDS_BYTES, L.wu3_nci);
    if (wcfg == 5) return go(&conv_wino3_kernel<0, 4, 3, 1>, Wino3CfgT<4, 3, 1>::LDS_BYTES, L.wu3_nci);
#endif
    return go(&conv_wino3_kernel<0, 4, 2, 1>, Wino3CfgT<4, 2, 1>::LDS_BYTES, L.wu3_nci);
  }
#ifdef ASX_EXPERIMENTAL_KERNELS
  if (L.kind == CK_3X3 && e->winograd >= 2 && dma && L.wu2.p != nullptr) {
    ConvArgs wa = a;
    wa.wp = L.wu2.f();
    wa.CG = L.wu_cg;
    wa.NCI = L.wu_nci;
    wa.tilesT = (a.To + 7) / 8;
    wa.tilesF = (a.Fo + 31) / 32;
    const int nb = wa.CG * wa.tilesT * wa.tilesF * B;
    static bool attr_done = false;
    if (!attr_done) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&conv_wino2_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                Wino2Cfg<4>::LDS_BYTES);
      attr_done = true;
    }
    return timed(e, cls, flops, bytes, s, [&]() {
      hipLaunchKernelGGL(conv_wino2_kernel<4>, dim3(nb), dim3(256), Wino2Cfg<4>::LDS_BYTES, s, wa);
    });
  }
  if (L.kind == CK_3X3 && e->winograd == 1 && dma && L.wu.p != nullptr) {
    ConvArgs wa = a;
    wa.wp = L.wu.f();
    wa.CG = L.wu_cg;
    wa.NCI = L.wu_nci;
    wa.tilesT = (a.To + WinoCfg::TH - 1) / WinoCfg::TH;
    wa.tilesF = (a.Fo + WinoCfg::TW - 1) / WinoCfg::TW;
    const int nb = wa.CG * wa.tilesT * wa.tilesF * B;
    static bool attr_done = false;
    if (!attr_done) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&conv_wino_kernel),
                                hipFuncAttributeMaxDynamicSharedMemorySize, WinoCfg::LDS_BYTES);
      attr_done = true;
    }
    return timed(e, cls, flops, bytes, s, [&]() {
      hipLaunchKernelGGL(conv_wino_kernel, dim3(nb), dim3(256), WinoCfg::LDS_BYTES, s, wa);
    });
  }
#endif
  if (L.kind == CK_DOWN && e->gemm_bf16x6 > 0 && e->down6 > 0 && L.wd6.p != nullptr && dma && (a.Fo & 3) == 0 && (reinterpret_cast<uintptr_t>(y) & 15) == 0 &&
      a.y_bstride % 4 == 0 && (v.res == nullptr || ((reinterpret_cast<uintptr_t>(v.res) & 15) == 0 && a.aux_bstride % 4 == 0))) {
    // the stride-2 conv on the 16-bit matrix pipe (bf16 x 6: exact three-way split operands, fp32 accumulation; kernels_updown6.h)
    ConvArgs da = a;
    da.wp = reinterpret_cast<const float *>(L.wd6.p);
    da.CG = L.wd6_cg;
    da.NCI = L.wd6_nst;
    // (tiles of 4 output rows -- template parameter TH of the kernel, half the weight traffic per pixel -- measured 35 % / 100 % slower on the deeper levels /
    // everywhere: three workgroups per CU instead of four, profiles/r06_down6_ab.txt; not instantiated)
    da.tilesT = (a.To + Down6Cfg::TH - 1) / Down6Cfg::TH;
    da.tilesF = (a.Fo + Down6Cfg::TW - 1) / Down6Cfg::TW;
    const int nb6 = da.CG * da.tilesT * da.tilesF * B;
    static bool attr6 = false;
    if (!attr6) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&conv_down6_kernel<3>), hipFuncAttributeMaxDynamicSharedMemorySize, Down6CfgT<3>::LDS_BYTES);
      (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&conv_down6_kernel<6>), hipFuncAttributeMaxDynamicSharedMemorySize, Down6CfgT<6>::LDS_BYTES);
      attr6 = true;
    }
    g_down6_launches.fetch_add(1);
    return timed(e, cls, flops, bytes, s, [&]() {
      e->prof_nprod = 6;
      if (L.wd6_nrep == 6) hipLaunchKernelGGL(conv_down6_kernel<6>, dim3(nb6), dim3(256), Down6CfgT<6>::LDS_BYTES, s, da);
      else hipLaunchKernelGGL(conv_down6_kernel<3>, dim3(nb6), dim3(256), Down6CfgT<3>::LDS_BYTES, s, da);
    });
  }
  if (L.kind == CK_UP && e->gemm_bf16x6 > 0 && e->up6 > 0 && L.wup6.p != nullptr && dma && (reinterpret_cast<uintptr_t>(y) & 15) == 0 && a.y_bstride % 4 == 0 &&
      (skip == nullptr || ((reinterpret_cast<uintptr_t>(skip) & 15) == 0 && a.aux_bstride % 4 == 0))) {
    // the transposed conv on the 16-bit matrix pipe (bf16 x 6; kernels_updown6.h)
    ConvArgs ua = a;
    ua.wp = reinterpret_cast<const float *>(L.wup6.p);
    ua.CG = L.wup6_cg;
    ua.NCI = L.wup6_nst;
    ua.tilesT = (T + 1) / 2;
    ua.tilesF = (F + 63) / 64;
    const int nbu = ua.CG * ua.tilesT * ua.tilesF * B;
    static bool attru = false;
    if (!attru) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&conv_up6_kernel<6>), hipFuncAttributeMaxDynamicSharedMemorySize, Up6CfgT<6>::LDS_BYTES);
      (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&conv_up6_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize, Up6CfgT<4>::LDS_BYTES);
      (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&conv_up6_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, Up6CfgT<2>::LDS_BYTES);
      attru = true;
    }
    g_up6_launches.fetch_add(1);
    return timed(e, cls, flops, bytes, s, [&]() {
      e->prof_nprod = 6;
      if (L.nrep == 6) hipLaunchKernelGGL(conv_up6_kernel<6>, dim3(nbu), dim3(256), Up6CfgT<6>::LDS_BYTES, s, ua);
      else if (L.nrep == 4) hipLaunchKernelGGL(conv_up6_kernel<4>, dim3(nbu), dim3(256), Up6CfgT<4>::LDS_BYTES, s, ua);
      else hipLaunchKernelGGL(conv_up6_kernel<2>, dim3(nbu), dim3(256), Up6CfgT<2>::LDS_BYTES, s, ua);
    });
  }
#define ASX_CONV_CASE(KH, KW, S, PAD, NR, KC, RPW, EPI)                                      \
  do {                                                                                       \
    if (dma) launch_conv_dma_t<ConvDmaCfg<KH, KW, S, PAD, NR, KC, RPW, EPI>>(d, nblk, s);    \
    else launch_conv_t<ConvCfg<KH, KW, S, PAD, NR, KC, RPW, EPI>>(a, nblk, s);               \
  } while (0)
  CHK(timed(e, cls, flops, bytes, s, [&]() {
    switch (L.kind) {
      case CK_3X3:
        if (L.nrep == 3 && L.kc == 4) ASX_CONV_CASE(3, 3, 1, 1, 3, 4, 2, EPI_BIAS_ACT);
        else if (L.nrep == 3) ASX_CONV_CASE(3, 3, 1, 1, 3, 8, 2, EPI_BIAS_ACT);
        else if (L.nrep == 2 && L.kc == 4) ASX_CONV_CASE(3, 3, 1, 1, 2, 4, 2, EPI_BIAS_ACT);
        else if (L.nrep == 2) ASX_CONV_CASE(3, 3, 1, 1, 2, 8, 2, EPI_BIAS_ACT);
        else ASX_CONV_CASE(3, 3, 1, 1, 1, 8, 2, EPI_BIAS_ACT);
        break;
      case CK_DOWN:
        if (L.nrep == 3 && L.kc == 2) ASX_CONV_CASE(2, 2, 2, 0, 3, 2, 2, EPI_BIAS_ACT);
        else if (L.nrep == 3) ASX_CONV_CASE(2, 2, 2, 0, 3, 4, 2, EPI_BIAS_ACT);
        else if (L.nrep == 2) ASX_CONV_CASE(2, 2, 2, 0, 2, 4, 2, EPI_BIAS_ACT);
        else ASX_CONV_CASE(2, 2, 2, 0, 1, 4, 2, EPI_BIAS_ACT);
        break;
      case CK_1X1:
        if (L.kc == 16) {
          if (L.nrep == 3) ASX_CONV_CASE(1, 1, 1, 0, 3, 16, 2, EPI_BIAS_ACT);
          else if (L.nrep == 2) ASX_CONV_CASE(1, 1, 1, 0, 2, 16, 2, EPI_BIAS_ACT);
          else ASX_CONV_CASE(1, 1, 1, 0, 1, 16, 2, EPI_BIAS_ACT);
        } else {
          if (L.nrep == 3) ASX_CONV_CASE(1, 1, 1, 0, 3, 4, 2, EPI_BIAS_ACT);
          else if (L.nrep == 2) ASX_CONV_CASE(1, 1, 1, 0, 2, 4, 2, EPI_BIAS_ACT);
          else ASX_CONV_CASE(1, 1, 1, 0, 1, 4, 2, EPI_BIAS_ACT);
        }
        break;
      case CK_UP:
        if (L.nrep == 6) ASX_CONV_CASE(1, 1, 1, 0, 6, 8, 1, EPI_UP_MULSKIP);
        else if (L.nrep == 4) ASX_CONV_CASE(1, 1, 1, 0, 4, 8, 1, EPI_UP_MULSKIP);
        else ASX_CONV_CASE(1, 1, 1, 0, 2, 8, 1, EPI_UP_MULSKIP);
        break;
      default: bad = 1;
    }
  }));
#undef ASX_CONV_CASE
  if (bad) {
    set_err("conv_launch: bad kind");
    return ASX_ERR_INVALID;
  }
  return ASX_OK;
}

template <int NREP, int MREP, bool KVEC>
static void launch_tdf_tt(const TdfArgs &a, hipStream_t s) {
  using CFG = TdfCfg<NREP, MREP>;
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&tdf_mfma_kernel<NREP, MREP, KVEC>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, CFG::LDS_BYTES);
    attr_done = true;
  }
  const int64_t nbm = (a.M + CFG::BM - 1) / CFG::BM;
  const int nbn = (a.N + CFG::BN - 1) / CFG::BN;
  hipLaunchKernelGGL((tdf_mfma_kernel<NREP, MREP, KVEC>), dim3((unsigned)(nbm * nbn)), dim3(256), CFG::LDS_BYTES, s,
                     a);
}
template <int NREP, int MREP>
static void launch_tdf_t(const TdfArgs &a, hipStream_t s) {
  if ((a.K & 3) == 0 && a.K >= 4) launch_tdf_tt<NREP, MREP, true>(a, s);
  else launch_tdf_tt<NREP, MREP, false>(a, s);
}

template <int NREP, int MREP, int BK>
static void launch_tdf_dma_tt(const TdfDmaArgs &a, hipStream_t s) {
  using CFG = TdfDmaCfg<NREP, MREP, BK>;
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&tdf_dma_kernel<NREP, MREP, BK>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, CFG::LDS_BYTES);
    attr_done = true;
  }
  const int64_t nbm = (a.M + CFG::BM - 1) / CFG::BM;
  const int nbn = (a.N + CFG::BN - 1) / CFG::BN;
  hipLaunchKernelGGL((tdf_dma_kernel<NREP, MREP, BK>), dim3((unsigned)(nbm * nbn)), dim3(256), CFG::LDS_BYTES, s, a);
}
template <int NREP, int MREP>
static void launch_tdf_dma_t(const TdfDmaArgs &a, hipStream_t s) {
  static const bool bk64 = getenv("ASX_GEMM_BK64") != nullptr;
  if (bk64 && NREP == 3 && MREP == 8 && a.K >= 128) launch_tdf_dma_tt<3, 8, 64>(a, s);
  else launch_tdf_dma_tt<NREP, MREP, 32>(a, s);
}

// tile choice for the LDS-DMA row GEMM: 128 x 192 or 128 x 128 for wide outputs, by a two-term cost model -- padding of
// the last column tile and wave quantisation over the 512 workgroup slots (2 per CU); the narrower tile moves 20 % more
// bytes per flop, charged as 4 %.  Measured: N = 512 out-proj / FF2 of BS-Roformer 103 -> 111 TFLOP/s (a third 192-wide tile
// would be 1/3 empty); HTDemucs transformer linears (43 k rows, N = 384 .. 1536) 73 -> 92 TFLOP/s.
// ASX_GEMM_T128=0 forces 128 x 192, =2 forces 128 x 128 (tuning aid).
// second-generation row GEMM (kernels_gemm2.h).  ASX_TDF2: 0 = tdf_dma_kernel only, 1 = tdf2 one tile per workgroup,
// 2 = + persistent over the column tiles of a row tile on short-K layers, 3 = + start stagger (ASX_TDF2_SBIT: block-id bit).
static int tdf2_mode() {
  static const int m = getenv("ASX_TDF2") ? atoi(getenv("ASX_TDF2")) : ASX_TDF2_DEFAULT;
  return m;
}
static bool tdf2_ok(const TdfDmaArgs &d) {
  auto a16 = [](const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
  const int64_t lda = d.lda ? d.lda : d.K, ldy = d.ldy ? d.ldy : d.N, ldr = d.ldr ? d.ldr : d.N;
  if (d.relu == 1 && (d.rscale != nullptr || d.rot_tab != nullptr)) return false;   // its ReLU ring epilogue applies neither
  return tdf2_mode() > 0 && d.K % 32 == 0 && d.K >= 32 && d.M % 8 == 0 && d.M >= 8 && d.M < (1ll << 31) && d.T > 0 && d.C > 0 && d.N % 8 == 0 && d.N >= 8 && lda % 4 == 0 &&
         ldy % 4 == 0 && ldr % 4 == 0 && a16(d.x) && a16(d.w) && a16(d.y) && (!d.res || a16(d.res)) && (!d.bias || a16(d.bias)) &&
         (uint64_t)8 * (uint64_t)lda * 4 < (1ull << 31) && (uint64_t)8 * (uint64_t)d.K * 4 < (1ull << 31) &&
         (uint64_t)16 * (uint64_t)ldy * 4 < (1ull << 31) && (uint64_t)16 * (uint64_t)ldr * 4 < (1ull << 31);
}
template <int NREP, int MREP, int ABL, int BK = 32>
static void launch_tdf2_abl(const TdfDmaArgs &a, hipStream_t s) {
  constexpr int BM = 16 * MREP, BN = 64 * NREP, LDS_BYTES = 2 * (BM + BN) * BK * 4;
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&tdf2_kernel<NREP, MREP, ABL, BK>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    attr_done = true;
  }
  static const int sbit = getenv("ASX_TDF2_SBIT") ? atoi(getenv("ASX_TDF2_SBIT")) : 8;
  const int64_t nbm = (a.M + BM - 1) / BM;
  const int nbn = (a.N + BN - 1) / BN;
  const int mode = tdf2_mode();
  // persistent over the column tiles when the K loop is short (prologue / epilogue are a visible share of a tile) and the
  // row tiles alone fill the 512 workgroup slots several times over
  const bool persist = mode >= 2 && nbn >= 2 && a.K <= 768 && nbm >= 2048;
  const int tiles = persist ? nbn : 1;
  hipLaunchKernelGGL((tdf2_kernel<NREP, MREP, ABL, BK>), dim3((unsigned)(nbm * (nbn / tiles))), dim3(256), LDS_BYTES, s, a, tiles,
                     (mode >= 3 && persist) ? sbit : -1);
}
template <int NREP, int MREP>
static void launch_tdf2(const TdfDmaArgs &a, hipStream_t s) {
#ifdef ASX_EXPERIMENTAL_KERNELS
  static const int abl = getenv("ASX_TDF2_ABL") ? atoi(getenv("ASX_TDF2_ABL")) : 0;   // ablation builds exist for the 128 x 192 tile only
  if constexpr (NREP == 3 && MREP == 8) {
    switch (abl) {
      case 1: return launch_tdf2_abl<3, 8, 1>(a, s);
      case 2: return launch_tdf2_abl<3, 8, 2>(a, s);
      case 3: return launch_tdf2_abl<3, 8, 3>(a, s);
      case 4: return launch_tdf2_abl<3, 8, 4>(a, s);
      case 5: return launch_tdf2_abl<3, 8, 5>(a, s);
      case 7: return launch_tdf2_abl<3, 8, 7>(a, s);
      case 8: return launch_tdf2_abl<3, 8, 8>(a, s);
      case 16: return launch_tdf2_abl<3, 8, 16>(a, s);
      default: break;
    }
  }
  // ASX_TDF2_BK16: 16-float stages (40 KB of LDS, launch bound 3) -- 1: on the 128 x 192 tile, 2: on a 64 x 192 tile
  static const int bk16 = getenv("ASX_TDF2_BK16") ? atoi(getenv("ASX_TDF2_BK16")) : 0;
  if constexpr (NREP == 3 && MREP == 8) {
    if (a.M % 16 == 0 && a.N % 16 == 0 && a.relu == 1 && a.K % 16 == 0) {
      if (bk16 == 1) return launch_tdf2_abl<3, 8, 0, 16>(a, s);
      if (bk16 == 2) return launch_tdf2_abl<3, 4, 0, 16>(a, s);
    }
  }
#endif
  launch_tdf2_abl<NREP, MREP, 0>(a, s);
}

// ---- third-generation row GEMM (kernels_gemm3.h): fp32 results from six bf16 MFMA products on exactly split operands ----------
// Per-engine switch (asx_engine::gemm_bf16x6): ASX_GEMM_BF16X6 (default 1) or asx_set_option(e, "gemm_bf16x6", n); 0 = the fp32-MFMA kernels only.
// launches of tdf3_kernel since the process started (tests assert that the path under test is the one that ran)
static std::atomic<long long> g_tdf3_launches{0};
static std::atomic<long long> g_tdf3h_launches{0};   // ... of which on the fp16 x 3 arithmetic (plain and GATHER mode)
static std::atomic<long long> g_tdf3ps_launches{0};  // ... of which read x as a pair image (operands split by their producer)
static std::atomic<long long> g_attn6_launches{0};   // launches of attention6_kernel (kernels_rof.h) / mha6_kernel (kernels_ht.h)
static std::atomic<long long> g_attn6h_launches{0};  // ... of which on the fp16 x 3 arithmetic

// The split image of a weight matrix is built on first use and cached PER ENGINE by (pointer, N, K, cin) (asx_engine::w3).  Every
// entry point that uploads or frees weights of an engine flushes that engine's images (w3_flush) -- an address reused by another tensor
// of the same shape can therefore never meet a stale image, and no other engine's load / destroy touches them (round 5: the cache
// was process-wide, so a second engine's commit freed images a captured hipGraph of the first still pointed at).
static void w3_flush(asx_engine *e) {
  if (!e) return;                                      // the entry points call this before they validate their arguments
  std::lock_guard<std::mutex> lk(e->w3_mu);
  for (auto &en : e->w3) (void)hipFree(en.img);
  e->w3.clear();
}

// drop the images of ONE weight buffer (a temporary layer of the single-op test hooks, about to be freed)
static void w3_drop(asx_engine *e, const void *w) {
  std::lock_guard<std::mutex> lk(e->w3_mu);
  for (size_t i = 0; i < e->w3.size();)
    if (e->w3[i].w == w) {
      (void)hipFree(e->w3[i].img);
      e->w3.erase(e->w3.begin() + (long)i);
    } else {
      ++i;
    }
}

static const u32x4 *w3_image(asx_engine *e, const float *w, int N, int K, hipStream_t s, int cin = 0, int kind = 0) {
  std::lock_guard<std::mutex> lk(e->w3_mu);
  for (auto &en : e->w3)
    if (en.w == w && en.N == N && en.K == K && en.cin == cin && en.kind == kind) return reinterpret_cast<const u32x4 *>(en.img);
  {
    // the build below synchronises the stream: inside a hipGraph capture that would invalidate the capture.  Say so and let the
    // caller run the fp32 kernel of this launch -- a capture belongs behind one warm-up forward (and behind any asx_set_option that changes the arithmetic)
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone) {
      set_err("split weight image of a %d x %d matrix requested while the stream is capturing: run one forward (after the last asx_set_option) before the capture", N, K);
      return nullptr;
    }
  }
  const int nst = cin > 0 ? (K / cin) * ((cin + 31) / 32) : 0;
  const int ntiles = (N + 15) / 16, nk = cin > 0 ? ((nst + 1) & ~1) : ((K + 63) / 64) * 2;   // an even number of 32-wide stages (zero padded)
  W3Entry en{w, N, K, cin, kind, nullptr};
  const size_t bytes = kind == 1 ? (size_t)ntiles * nk * 2 * 1024 + (size_t)ntiles * 16 : (size_t)ntiles * nk * 3 * 1024;   // kind 1: + four exponents per tile
  if (hipMalloc(&en.img, bytes) != hipSuccess) return nullptr;
  const int64_t total = (int64_t)ntiles * nk * 64;
  if (kind == 1)
    hipLaunchKernelGGL(w3h_split_kernel, dim3((unsigned)ntiles), dim3(256), 0, s, w, reinterpret_cast<u32x4 *>(en.img), N, K, ntiles, cin, nst);
  else
    hipLaunchKernelGGL(w3_split_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, w, reinterpret_cast<u32x4 *>(en.img), N, K,
                     total, cin, nst);
  // built once per weight tensor (the first forward after a load): make the image visible to every stream before it is published
  if (hipStreamSynchronize(s) != hipSuccess) {
    (void)hipFree(en.img);
    return nullptr;
  }
  e->w3.push_back(en);
  return reinterpret_cast<const u32x4 *>(en.img);
}

static bool tdf3_ok(const asx_engine *e, const TdfDmaArgs &d) {
  auto a16 = [](const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
  const int64_t lda = d.lda ? d.lda : d.K, ldy = d.ldy ? d.ldy : d.N, ldr = d.ldr ? d.ldr : d.N;
  if (d.relu == 1 && (d.rscale != nullptr || d.rot_tab != nullptr)) return false;   // as tdf2_ok: the ReLU ring epilogue applies neither
  return e->gemm_bf16x6 > 0 && d.K % 32 == 0 && d.K >= 64 && d.M >= 1 && d.M < (1ll << 31) && d.T > 0 && d.C > 0 && d.N % 8 == 0 &&
         d.N > 64 && lda % 4 == 0 && ldy % 4 == 0 && ldr % 4 == 0 && a16(d.x) && a16(d.w) && a16(d.y) && (!d.res || a16(d.res)) &&
         (!d.bias || a16(d.bias)) && (uint64_t)16 * (uint64_t)ldy * 4 < (1ull << 31) && (uint64_t)16 * (uint64_t)ldr * 4 < (1ull << 31);
}
template <int NREP, int MREP, int ABL, bool H = false, bool PS = false>
static void launch_tdf3_abl(const TdfDmaArgs &a0, const u32x4 *w3, hipStream_t s) {
  constexpr int BM = 16 * MREP, BN = 64 * NREP, LDS_BYTES = H ? tdf3h_lds_bytes(BM) : 2 * 3 * BM * 64;
  TdfDmaArgs a = a0;
  const int64_t nbm = (a.M + BM - 1) / BM;
  const int nbn = (a.N + BN - 1) / BN;
  // tile -> XCD map (kernels_gemm3.h); ASX_TDF3_MAP = "<narrow><wide>" digits for N < 8 tiles / N >= 8 tiles (A/B switch)
  // Default (round 5, profiles/r05_tdf3_tile_map_ab.txt + r05_pmc_tdf3.json): fewer than 8 column tiles -> map 1 (the column tiles of a
  // row block are neighbours on one XCD, so x crosses the fabric once instead of twice: level-0 first TDF linear 6.86 -> 6.69 ms,
  // fabric traffic 2.0x -> ~1.0x algorithmic); 8 or more -> map 0 (column tiles partitioned over the XCDs; maps 1 / 2 measure the same).
  static const int map_env = getenv("ASX_TDF3_MAP") ? atoi(getenv("ASX_TDF3_MAP")) : -1;
  a.tile_map = map_env >= 0 ? (nbn >= 8 ? map_env % 10 : map_env / 10) : (nbn < 8 ? 1 : 0);
  // N = 384 exactly (the first TDF linear of level 0): ONE 8-wave workgroup per row block computes both 192-column halves from one x tile -- half the
  // x loads, splits and LDS stores per MFMA (kernels_gemm3.h, template parameter NW).  ASX_TDF3_NW8=0: A/B
  if constexpr (H && !PS && ABL == 0 && NREP == 3 && MREP == 8) {
    static const bool nw8 = !(getenv("ASX_TDF3_NW8") && atoi(getenv("ASX_TDF3_NW8")) == 0);
    if (nw8 && a.N == 384 && a.yexp == nullptr) {        // (a pair-image producer writes one exponent span per 192-column tile: the 4-wave form)
      hipLaunchKernelGGL((tdf3_kernel<3, 8, 0, false, true, false, 8>), dim3((unsigned)nbm), dim3(512), LDS_BYTES, s, a, w3, RowGather{});
      g_tdf3_launches.fetch_add(1);
      g_tdf3h_launches.fetch_add(1);
      return;
    }
  }
  hipLaunchKernelGGL((tdf3_kernel<NREP, MREP, ABL, false, H, PS>), dim3((unsigned)(nbm * nbn)), dim3(256), LDS_BYTES, s, a, w3, RowGather{});
  g_tdf3_launches.fetch_add(1);
  if (H) g_tdf3h_launches.fetch_add(1);
  if (PS) g_tdf3ps_launches.fetch_add(1);
}
// GATHER mode (kernels_gemm3.h): stride-1 convolutions of the channels-last nets as implicit GEMMs on the same kernel
static std::atomic<long long> g_tdf3_gather_launches{0};
template <int NREP, int MREP>
static bool launch_tdf3_gather(asx_engine *e, const TdfDmaArgs &a, const RowGather &gq, hipStream_t s) {
  const bool h = e->gemm_f16x3 > 0;
  const u32x4 *w3 = w3_image(e, a.w, a.N, a.K, s, (gq.cin & 31) ? gq.cin : 0, h ? 1 : 0);   // channel counts off the 32-grid get their own padded image
  if (!w3) return false;
  e->prof_nprod = h ? 3 : 6;
  constexpr int BM = 16 * MREP, BN = 64 * NREP;
  const int64_t nbm = (a.M + BM - 1) / BM;
  const int nbn = (a.N + BN - 1) / BN;
  if (h) {
    hipLaunchKernelGGL((tdf3_kernel<NREP, MREP, 0, true, true>), dim3((unsigned)(nbm * nbn)), dim3(256), tdf3h_lds_bytes(BM), s, a, w3, gq);
    g_tdf3h_launches.fetch_add(1);
  } else {
    hipLaunchKernelGGL((tdf3_kernel<NREP, MREP, 0, true>), dim3((unsigned)(nbm * nbn)), dim3(256), 2 * 3 * BM * 64, s, a, w3, gq);
  }
  g_tdf3_gather_launches.fetch_add(1);
  return true;
}
static bool launch_tdf3_gather_auto(asx_engine *e, const TdfDmaArgs &d, const RowGather &gq, hipStream_t s) {
  if (d.glu_cout > 0) {                                // value / gate fragment pairs: 128-column tiles (two fragments per wave)
    const bool small = (d.M + 127) / 128 * ((d.N + 127) / 128) < 1024;   // grid too small for 128-row tiles to fill the chip twice
    return small ? launch_tdf3_gather<2, 4>(e, d, gq, s) : launch_tdf3_gather<2, 8>(e, d, gq, s);
  }
  if (d.N <= 64) return launch_tdf3_gather<1, 8>(e, d, gq, s);   // narrow layers (48 .. 64 columns): one fragment per wave
  if (d.N > 128) {
    const double rows = (double)((d.M + 127) / 128);
    auto cost = [&](int bn, double eff) { return ceil(rows * (double)((d.N + bn - 1) / bn) / 512.0) * bn / eff; };
    return cost(128, 0.96) < cost(192, 1.0) ? launch_tdf3_gather<2, 8>(e, d, gq, s) : launch_tdf3_gather<3, 8>(e, d, gq, s);
  }
  return launch_tdf3_gather<2, 4>(e, d, gq, s);
}
template <int NREP, int MREP>
static bool launch_tdf3(asx_engine *e, const TdfDmaArgs &a, hipStream_t s) {
  static const int abl0 = getenv("ASX_TDF3_ABL") ? atoi(getenv("ASX_TDF3_ABL")) : 0;
  // ASX_F16X3_N (bisection aid, kept: tools/debug_rof_race.py found the rotary-epilogue anomaly with it): fp16 x 3 on the launches with this N only (< 0: all but)
  static const int only_n = getenv("ASX_F16X3_N") ? atoi(getenv("ASX_F16X3_N")) : 0;
  const bool h = e->gemm_f16x3 > 0 && abl0 == 0 && (only_n == 0 || (only_n > 0 ? a.N == only_n : a.N != -only_n));
  const u32x4 *w3 = w3_image(e, a.w, a.N, a.K, s, 0, h ? 1 : 0);
  if (!w3) return false;                               // out of memory for the image: the caller falls back to the fp32 kernels
  e->prof_nprod = h ? 3 : 6;
  if ((a.xexp != nullptr || a.yexp != nullptr) && !h) {
    // a pair image is only ever set up behind tdf3h_will_run(): reaching this line means the two disagree -- stop rather than multiply garbage
    fprintf(stderr, "asx: pair-image operands on a launch that is not on the fp16 x 3 row GEMM (N=%d K=%d)\n", a.N, a.K);
    abort();
  }
  if (h) {
#ifdef ASX_EXPERIMENTAL_KERNELS
    if (a.xexp != nullptr) launch_tdf3_abl<NREP, MREP, 0, true, true>(a, w3, s);
    else
#else
    if (a.xexp != nullptr || a.yexp != nullptr) {
      fprintf(stderr, "asx: pair-image operands in a library built without them\n");
      abort();
    }
#endif
    launch_tdf3_abl<NREP, MREP, 0, true>(a, w3, s);
    return true;
  }
#ifdef ASX_EXPERIMENTAL_KERNELS
  static const int abl = getenv("ASX_TDF3_ABL") ? atoi(getenv("ASX_TDF3_ABL")) : 0;   // ablation builds exist for the 128 x 192 tile only
  if constexpr (NREP == 3 && MREP == 8) {
    switch (abl) {
      case 1: launch_tdf3_abl<3, 8, 1>(a, w3, s); return true;
      case 2: launch_tdf3_abl<3, 8, 2>(a, w3, s); return true;
      case 4: launch_tdf3_abl<3, 8, 4>(a, w3, s); return true;
      case 8: launch_tdf3_abl<3, 8, 8>(a, w3, s); return true;
      case 13: launch_tdf3_abl<3, 8, 13>(a, w3, s); return true;
      default: break;
    }
  }
#endif
  launch_tdf3_abl<NREP, MREP, 0>(a, w3, s);
  return true;
}

// which tile form launch_tdf_dma_auto gives a layer on tdf3_kernel: 0 = not tdf3's (N <= 64 or tdf3_ok says no), 1 = 64 x 128, 2 = 128 x 128, 3 = 128 x 192
static int tdf3_tile_form(const asx_engine *e, const TdfDmaArgs &d) {
  static const int t128 = getenv("ASX_GEMM_T128") ? atoi(getenv("ASX_GEMM_T128")) : 1;
  static const int small = getenv("ASX_TDF2_SMALL") ? atoi(getenv("ASX_TDF2_SMALL")) : 0;   // A/B: 64 x 128 tiles (3+ workgroups per CU) on short-K layers
  if (!tdf3_ok(e, d) || d.N <= 64) return 0;
  if (d.N <= 128) return 1;
  if ((small && d.K <= small) || d.prefer_small) return 1;
  const double rows = (double)((d.M + 127) / 128);
  auto cost = [&](int bn, double eff) {
    const double blocks = rows * (double)((d.N + bn - 1) / bn);
    return ceil(blocks / 512.0) * bn / eff;
  };
  // ASX_TDF3_EFF128: relative efficiency charged to the 128-column tile of the bf16 x 6 kernel.  Measured on the BS-Roformer and
  // HTDemucs linears: 0.96 / 0.85 / 0.75 -> 1251 / 1262 / 1262 ms and 30.5 / 30.4 / 30.5 ms per song -- no reason to move off
  // the fp32 kernel's figure (N = 512 stays on four 128-column tiles).
  static const double eff128 = getenv("ASX_TDF3_EFF128") ? atof(getenv("ASX_TDF3_EFF128")) : 0.96;
  // fp16 x 3: the 128-column tile needs 166 registers -- three workgroups per CU -- and measures as fast per MAC as the 192-column
  // one or faster on the SHORT-K shapes (BS-Roformer FF1 4.45 vs 4.74 ms, qkv 2.85 vs 3.28, HTDemucs linear 0.266 vs 0.328:
  // profiles/r05_gemm_f16x3.txt, "tile forms"): no handicap there.  Long K keeps it: the level-0 / level-1 first TDF linears
  // (K = 3072 / 1536) are 3 % / 35 % slower on the narrow tile.
  const double e128 = (e->gemm_f16x3 > 0 && d.K <= 512 && !getenv("ASX_TDF3_EFF128")) ? 1.0 : eff128;
  const bool narrow3 = t128 == 2 || (t128 == 1 && cost(128, e128) < cost(192, 1.0));
  return narrow3 ? 2 : 3;
}

// Will launch_tdf_dma_auto run this layer on the fp16 x 3 form of tdf3_kernel?  Only then may its x be a pair image, or its y be written as
// one (TdfDmaArgs::xexp / yexp).  Builds (and caches) the layer's split weight image, so that the launch itself cannot fall back.
static bool tdf3h_will_run(asx_engine *e, const TdfDmaArgs &d, hipStream_t s) {
  static const bool dbg = getenv("ASX_TDF3_ABL") != nullptr || getenv("ASX_F16X3_N") != nullptr;   // bisection aids of launch_tdf3: no pair images with them
  if (dbg || e->gemm_f16x3 <= 0 || e->pair_images <= 0 || tdf3_tile_form(e, d) == 0) return false;
  return w3_image(e, d.w, d.N, d.K, s, 0, 1) != nullptr;
}
// columns per exponent span of the pair image this layer writes (its column tile)
static int tdf3_tile_cols(const asx_engine *e, const TdfDmaArgs &d) { return tdf3_tile_form(e, d) == 3 ? 192 : 128; }
// consumer side of a pair image written in `cols`-column spans: K of the consumer = N of the producer
static void tdf3_set_xexp(TdfDmaArgs &d, const int *tab, int cols) {
  d.xexp = tab;
  d.xexp_gs = cols / 32;
  d.xexp_n = (d.K + cols - 1) / cols;
  d.xexp_inv = (65536 + d.xexp_gs - 1) / d.xexp_gs;   // (stage * inv) >> 16 == stage / gs for every stage < 65536 / gs (gs <= 64: K < 2^21 / ... checked by the caller's K)
}

static void launch_tdf_dma_auto(asx_engine *e, const TdfDmaArgs &d, hipStream_t s) {
  static const int t128 = getenv("ASX_GEMM_T128") ? atoi(getenv("ASX_GEMM_T128")) : 1;
  const bool v2 = tdf2_ok(d);
  static const int small = getenv("ASX_TDF2_SMALL") ? atoi(getenv("ASX_TDF2_SMALL")) : 0;
  const int form = tdf3_tile_form(e, d);
  if (form == 1 && d.N > 128 && launch_tdf3<2, 4>(e, d, s)) return;
  if (v2 && ((small && d.K <= small) || d.prefer_small) && d.N > 128) return launch_tdf2<2, 4>(d, s);
  if (d.N > 128) {
    const double rows = (double)((d.M + 127) / 128);
    auto cost = [&](int bn, double eff) {
      const double blocks = rows * (double)((d.N + bn - 1) / bn);
      return ceil(blocks / 512.0) * bn / eff;
    };
    const bool narrow = t128 == 2 || (t128 == 1 && cost(128, 0.96) < cost(192, 1.0));
    if (form == 2 && launch_tdf3<2, 8>(e, d, s)) return;
    if (form == 3 && launch_tdf3<3, 8>(e, d, s)) return;
    if (narrow) v2 ? launch_tdf2<2, 8>(d, s) : launch_tdf_dma_t<2, 8>(d, s);
    else v2 ? launch_tdf2<3, 8>(d, s) : launch_tdf_dma_t<3, 8>(d, s);
  } else if (d.N > 64) {
    if (form == 1 && launch_tdf3<2, 4>(e, d, s)) return;
    v2 ? launch_tdf2<2, 4>(d, s) : launch_tdf_dma_t<2, 4>(d, s);
  } else {
    launch_tdf_dma_t<1, 4>(d, s);
  }
}

// pair_out / pair_in (tdf_block): y written / x read as a pair image with its exponent spans in `pair_tab` (`pair_cols` columns per span)
static void tdf_fill_args(asx_engine *e, const TdfLayer &L, const float *x, const float *res, float *y, int64_t M, int T, int relu, TdfDmaArgs &d) {
  d = TdfDmaArgs{};
  d.x = x;
  d.w = L.w.f();
  d.bias = L.has_bias ? L.bias.f() : nullptr;
  d.scale = L.scale.p ? L.scale.f() : nullptr;
  d.shift = L.shift.p ? L.shift.f() : nullptr;
  d.res = res;
  d.zeros = e->d_zeros.f();
  d.y = y;
  d.M = M;
  d.N = L.n;
  d.K = L.k;
  d.C = L.c;
  d.T = T;
  d.relu = relu;
  static const int nt_mode = getenv("ASX_NT") ? atoi(getenv("ASX_NT")) : 0;
  d.nt = ((nt_mode >> 1) & 1) | ((nt_mode >> 2) & 1) << 1;
}

static int tdf_launch(asx_engine *e, const TdfLayer &L, const float *x, const float *res, float *y, int64_t M,
                      int T, hipStream_t s, int relu = 1, int *pair_out = nullptr, const int *pair_in = nullptr, int pair_cols = 0) {
  TdfArgs a{};
  a.x = x;
  a.w = L.w.f();
  a.bias = L.has_bias ? L.bias.f() : nullptr;
  a.scale = L.scale.p ? L.scale.f() : nullptr;
  a.shift = L.shift.p ? L.shift.f() : nullptr;
  a.relu = relu;
  a.res = res;
  a.y = y;
  a.M = M;
  a.N = L.n;
  a.K = L.k;
  a.C = L.c;
  a.T = T;
  if (M <= 0) return ASX_OK;
  const double flops = 2.0 * (double)M * L.n * L.k;
  const double bytes = 4.0 * ((double)M * L.k + (double)M * L.n * (res ? 2 : 1) + (double)L.n * L.k);
  const bool dma = (L.k % 4 == 0) && ((reinterpret_cast<uintptr_t>(x) & 15) == 0) && getenv("ASX_NO_DMA") == nullptr;
  TdfDmaArgs d;
  tdf_fill_args(e, L, x, res, y, M, T, relu, d);
  if (pair_out) {
    d.yexp = pair_out;
    d.yexp_n = (L.n + pair_cols - 1) / pair_cols;
  }
  if (pair_in) tdf3_set_xexp(d, pair_in, pair_cols);
  if ((pair_out || pair_in) && !dma) {
    set_err("tdf_launch: pair image on a layer the row-GEMM kernels cannot take");
    return ASX_ERR_STATE;
  }
  return timed(e, ASX_PROF_TDF, flops, bytes, s, [&]() {
    if (dma) {
      launch_tdf_dma_auto(e, d, s);
    } else {
      if (L.n > 128) launch_tdf_t<3, 8>(a, s);
      else if (L.n > 64) launch_tdf_t<2, 4>(a, s);
      else launch_tdf_t<1, 4>(a, s);
    }
  });
}

// out = x + tdf1(tdf0(x)): the two linears of a TDF block with a bottleneck (modules.py:61-70).  The bottleneck activations H have one reader,
// the second linear: when both linears run on the fp16 x 3 row GEMM the first writes H as a pair image (split once, in its epilogue; exponent
// spans in HE) and the second multiplies the parts as they are (kernels_gemm3.h, "operands split by their producer").
static int tdf_pair_launch(asx_engine *e, const TdfLayer &L0, const TdfLayer &L1, const float *x, float *H, int *HE, float *out, int64_t M, int T,
                           hipStream_t s) {
  int *pair_tab = nullptr;
  int pair_cols = 0;
  if (M > 0 && HE != nullptr) {
    TdfDmaArgs d0, d1;
    tdf_fill_args(e, L0, x, nullptr, H, M, T, 1, d0);
    tdf_fill_args(e, L1, H, x, out, M, T, 1, d1);
    const bool al = ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(H)) & 15) == 0 && getenv("ASX_NO_DMA") == nullptr;
    if (al && L0.n % 32 == 0 && L0.k % 4 == 0 && L1.k % 4 == 0 && tdf3h_will_run(e, d0, s) && tdf3h_will_run(e, d1, s)) {
      pair_cols = tdf3_tile_cols(e, d0);
      pair_tab = HE;
    }
  }
  CHK(tdf_launch(e, L0, x, nullptr, H, M, T, s, 1, pair_tab, nullptr, pair_cols));
  return tdf_launch(e, L1, H, x, out, M, T, s, 1, nullptr, pair_tab, pair_cols);
}

static int tdf_pack(TdfLayer &L, int n, int k, int c, const float *w, const float *bias, const float *scale,
                    const float *shift) {
  L.n = n;
  L.k = k;
  L.c = c;
  L.has_bias = bias != nullptr;
  CHK(L.w.ensure((size_t)n * k * 4));
  HIPCHK(hipMemcpy(L.w.p, w, (size_t)n * k * 4, hipMemcpyHostToDevice));
  if (bias) {
    CHK(L.bias.ensure((size_t)n * 4));
    HIPCHK(hipMemcpy(L.bias.p, bias, (size_t)n * 4, hipMemcpyHostToDevice));
  }
  if (scale && shift) {
    CHK(L.scale.ensure((size_t)c * 4));
    CHK(L.shift.ensure((size_t)c * 4));
    HIPCHK(hipMemcpy(L.scale.p, scale, (size_t)c * 4, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(L.shift.p, shift, (size_t)c * 4, hipMemcpyHostToDevice));
  }
  return ASX_OK;
}

// ----------------------------------------------------------------------------
// STFT / iSTFT launch helpers (device buffers)
// ----------------------------------------------------------------------------
static int stft_launch(asx_engine *e, const float *wave, const int64_t *d_starts, int64_t n_song, int B, int64_t C,
                       int T, float *spec, int tf_layout, int zero_low, float sign, hipStream_t s) {
  StftArgs a{};
  a.wave = wave;
  a.chunk_start = d_starts;
  a.n_song = n_song;
  a.trim = e->cfg.n_fft / 2;
  a.C = C;
  a.hop = e->cfg.hop_length;
  a.T = T;
  a.dim_f = e->cfg.dim_f;
  a.zero_low = zero_low;
  a.tf_layout = tf_layout;
  a.spec = spec;
  a.window = e->d_window.f();
  a.tw = reinterpret_cast<const float2 *>(e->d_tw.p);
  a.sign = sign;
  const double bytes = 4.0 * ((double)B * 2 * C + (double)B * 4 * T * e->cfg.dim_f);
  FftPlan p = e->plan;
  if (e->fft3 && tf_layout == 1 && e->cfg.dim_f <= f3::NH) {
    // n_fft 6144 / hop 1024, engine-internal [B, 4, T, F] layout: three-pass register FFT (kernels_fft3.h)
    f3::Stft3Args f{};
    f.wave = wave;
    f.chunk_start = d_starts;
    f.n_song = n_song;
    f.trim = a.trim;
    f.C = C;
    f.T = T;
    f.dim_f = a.dim_f;
    f.zero_low = zero_low;
    f.spec = spec;
    f.window = a.window;
    f.tw = reinterpret_cast<const f3::cplx *>(a.tw);
    f.twB = reinterpret_cast<const f3::cplx *>(e->d_tw3.p);
    f.twC = f.twB + 16 * 12;
    f.sign = sign;
    // ~16 frames per workgroup, in even shares: many short workgroups balance better over the CUs than two or three rounds
    // of long ones (measured: 8 / 16 frames 0.257 / 0.253 ms, 20 frames in exactly two rounds 0.368 ms)
    static const int GS = getenv("ASX_FFT3_GS") ? std::max(1, atoi(getenv("ASX_FFT3_GS"))) : 16;
    f.n_groups = std::max(1, T / GS);
    return timed(e, ASX_PROF_STFT, 0.0, bytes, s, [&]() {
      if (e->fft3p)
        hipLaunchKernelGGL(f3::stft3p_kernel, dim3(f.n_groups, 2, B), dim3(256), f3::STFT3P_LDS_BYTES, s, f);
      else
        hipLaunchKernelGGL(f3::stft3_kernel, dim3(T, 2, B), dim3(256), f3::STFT3_LDS_BYTES, s, f);
    });
  }
  return timed(e, ASX_PROF_STFT, 0.0, bytes, s, [&]() {
    hipLaunchKernelGGL(stft_kernel, dim3(T, 2, B), dim3(256), stft_lds(p), s, a, p);
  });
}

static int istft_launch(asx_engine *e, const float *spec, int B, int T, int tf_layout, int combine, float *frames,
                        hipStream_t s) {
  IstftArgs a{};
  a.spec = spec;
  a.T = T;
  a.dim_f = e->cfg.dim_f;
  a.tf_layout = tf_layout;
  a.combine = combine;
  a.frames = frames;
  a.window = e->d_window.f();
  a.tw = reinterpret_cast<const float2 *>(e->d_tw.p);
  const double bytes = 4.0 * ((double)B * 4 * T * e->cfg.dim_f * (combine ? 2 : 1) + (double)B * 2 * T * e->cfg.n_fft);
  FftPlan p = e->plan;
  return timed(e, ASX_PROF_ISTFT, 0.0, bytes, s, [&]() {
    hipLaunchKernelGGL(istft_kernel, dim3(T, 2, B), dim3(256), istft_lds(p), s, a, p);
  });
}

static int ola_launch(asx_engine *e, const float *frames, const float *env, const int64_t *d_nact, int B, int T,
                      int64_t C, float *out, hipStream_t s) {
  const double bytes = 4.0 * ((double)B * 2 * T * e->cfg.n_fft + (double)B * 2 * C);
  const int n_fft = e->cfg.n_fft, hop = e->cfg.hop_length;
  return timed(e, ASX_PROF_OLA, 0.0, bytes, s, [&]() {
    hipLaunchKernelGGL(ola_kernel, dim3((unsigned)((C + 255) / 256), 2, B), dim3(256), 0, s, frames, env, d_nact,
                       n_fft, hop, T, C, out);
  });
}

// STFT.inverse + the chunk's Hann window: the fast path (n_fft 6144 / hop 1024) inverse-transforms, overlap-adds in an LDS
// ring and writes the chunk directly (kernels_fft3.h); every other geometry runs istft_kernel -> frames -> ola_kernel.
static int istft_ola_launch(asx_engine *e, const float *spec, int B, int T, int combine, const int64_t *d_nact, int64_t C,
                            float *out, hipStream_t s) {
  // emit_hop / emit_finish / seam3_kernel store float4: `out` (the caller's chunk buffer + a chunk offset) must be 16-byte aligned
  if (!(e->fft3 && e->cfg.dim_f <= f3::NH && C == (int64_t)f3::HOP * (T - 1) && (reinterpret_cast<uintptr_t>(out) & 15) == 0)) {
    CHK(istft_launch(e, spec, B, T, 1, combine, e->frames.f(), s));
    return ola_launch(e, e->frames.f(), e->d_env.f(), d_nact, B, T, C, out, s);
  }
  // >= 16 frames per workgroup, in even shares.  The grouping depends on T only: a hop on a seam is summed as
  // (tail partial) + (head partial), so results stay bit-identical whatever the batch size.
  static const int G = getenv("ASX_FFT3_G") ? std::max(5, atoi(getenv("ASX_FFT3_G"))) : 16;
  const int ng = std::max(1, T / G);
  CHK(e->seam3.ensure((size_t)B * 2 * ng * 2 * 5 * f3::HOP * 4));
  f3::Istft3Args f{};
  f.spec = spec;
  f.T = T;
  f.dim_f = e->cfg.dim_f;
  f.combine = combine;
  f.window = e->d_window.f();
  f.tw = reinterpret_cast<const f3::cplx *>(e->d_tw.p);
  f.twB = reinterpret_cast<const f3::cplx *>(e->d_tw3.p);
  f.twC = f.twB + 16 * 12;
  f.env = e->d_env.f();
  f.n_act = d_nact;
  f.C = C;
  f.out = out;
  f.seam = e->seam3.f();
  f.n_groups = ng;
  f.hann = (C == (int64_t)e->cfg.hop_length * (e->cfg.segment_size - 1) && e->d_hann3.p) ? reinterpret_cast<const double *>(e->d_hann3.p) : nullptr;
  // algorithmic bytes: the spectrogram read once, the chunk written once (the seam buffer's round trip is overhead, not counted)
  const double bytes = 4.0 * ((double)B * 4 * T * e->cfg.dim_f * (combine ? 2 : 1) + (double)B * 2 * C);
  return timed(e, ASX_PROF_ISTFT, 0.0, bytes, s, [&]() {
    const bool aligned = e->cfg.dim_f % 4 == 0 && (reinterpret_cast<uintptr_t>(spec) & 15) == 0;
    if (e->fft3p && combine == 0 && aligned)
    {
      static const int abl = getenv("ASX_ISTFT_ABL") ? atoi(getenv("ASX_ISTFT_ABL")) : 0;   // timing probes (results invalid)
      if (abl == 1) hipLaunchKernelGGL(f3::istft3p_kernel<1>, dim3(ng, 2, B), dim3(256), f3::ISTFT3P_LDS_BYTES, s, f);
      else if (abl == 2) hipLaunchKernelGGL(f3::istft3p_kernel<2>, dim3(ng, 2, B), dim3(256), f3::ISTFT3P_LDS_BYTES, s, f);
      else if (abl == 3) hipLaunchKernelGGL(f3::istft3p_kernel<3>, dim3(ng, 2, B), dim3(256), f3::ISTFT3P_LDS_BYTES, s, f);
      else hipLaunchKernelGGL(f3::istft3p_kernel<0>, dim3(ng, 2, B), dim3(256), f3::ISTFT3P_LDS_BYTES, s, f);
    }
    else
      hipLaunchKernelGGL(f3::istft3_kernel, dim3(ng, 2, B), dim3(256), f3::ISTFT3_LDS_BYTES, s, f);
    if (ng > 1) hipLaunchKernelGGL(f3::seam3_kernel, dim3(5, ng - 1, B * 2), dim3(256), 0, s, f);
  });
}

// ----------------------------------------------------------------------------
// net forward on device buffers: spec_in [B,4,T,F] (TF layout) -> spec_out
// ----------------------------------------------------------------------------
// GroupNorm(2, C) + ReLU (+ res) (* mul) over a dense [B, C, P] tensor, in place or into y (kernels_fft.h: gn_*_kernel); the ConvTDFNet
// variant built with optimizer == 'adamw' (uvr_lib_v5/mdxnet.py:48-49) normalises with batch-dependent statistics, so the norm cannot
// be folded into the weights: every conv / linear runs bare (bias only) and this pass follows it.
static int gn_launch(asx_engine *e, const float *x, int B, int C, int64_t P, const DevBuf &gw, const DevBuf &gb, const float *res,
                     const float *mul, float *y, hipStream_t s) {
  if (B <= 0) return ASX_OK;
  REQUIRE(C % 2 == 0 && gw.p && gb.p && e->gn_part.bytes >= (size_t)B * C * 16, "GroupNorm(2, %d): channels / workspace", C);
  double2 *part = reinterpret_cast<double2 *>(e->gn_part.p);
  const double bytes = 4.0 * (double)B * C * (double)P;
  CHK(timed(e, ASX_PROF_MISC, 0.0, bytes, s,
            [&]() { hipLaunchKernelGGL(gn_partial_kernel, dim3((unsigned)C, (unsigned)B), dim3(256), 0, s, x, C, P, part); }));
  const int nx = (int)std::max<int64_t>(1, std::min<int64_t>(32, (P / 4 + 1023) / 1024));
  return timed(e, ASX_PROF_MISC, 0.0, bytes * (2 + (res ? 1 : 0) + (mul ? 1 : 0)), s, [&]() {
    hipLaunchKernelGGL(gn_apply_kernel, dim3((unsigned)nx, (unsigned)C, (unsigned)B), dim3(256), 0, s, x, C, 2, P, part, gw.f(), gb.f(), 1e-5f, 1, res,
                       mul, y);
  });
}

static int block_forward(asx_engine *e, const Block &blk, float *&cur, float *dest, int B, hipStream_t s) {
  // TFC convs rotate through e->R; the TDF output goes to `dest` (or a free R buffer when dest == nullptr)
  auto next_free = [&](const float *a, const float *b) -> float * {
    for (int i = 0; i < 3; ++i)
      if (e->R[i].f() != a && e->R[i].f() != b) return e->R[i].f();
    return nullptr;
  };
  const bool gn = e->net.norm == 1;
  const int64_t plane = (int64_t)blk.t * blk.f;
  for (size_t j = 0; j < blk.tfc.size(); ++j) {
    float *out = next_free(cur, nullptr);
    CHK(conv_launch(e, blk.tfc[j], cur, nullptr, out, B, blk.t, blk.f, s));
    if (gn) CHK(gn_launch(e, out, B, blk.c, plane, blk.tfc[j].gn_w, blk.tfc[j].gn_b, nullptr, nullptr, out, s));
    cur = out;
  }
  const int64_t M = (int64_t)B * blk.c * blk.t;
  const int bnk = e->net.bn;
  if (bnk < 0) {                                       // bn is None: TFC only (modules.py:52, 74)
    if (dest) {
      HIPCHK(hipMemcpyAsync(dest, cur, (size_t)M * blk.f * 4, hipMemcpyDeviceToDevice, s));
      cur = dest;
    }
    return ASX_OK;
  }
  if (bnk == 0) {                                      // one Linear(f, f) + norm + ReLU (modules.py:55-60), x + tdf(x)
    float *out = dest ? dest : next_free(cur, nullptr);
    if (gn) {
      float *tmp = next_free(cur, out);
      CHK(tdf_launch(e, blk.tdf0, cur, nullptr, tmp, M, blk.t, s, 0));
      CHK(gn_launch(e, tmp, B, blk.c, plane, blk.tdf0.gn_w, blk.tdf0.gn_b, cur, nullptr, out, s));
    } else {
      CHK(tdf_launch(e, blk.tdf0, cur, cur, out, M, blk.t, s));
    }
    cur = out;
    return ASX_OK;
  }
  if (gn) {
    CHK(tdf_launch(e, blk.tdf0, cur, nullptr, e->H.f(), M, blk.t, s, 0));
    CHK(gn_launch(e, e->H.f(), B, blk.c, (int64_t)blk.t * (blk.f / bnk), blk.tdf0.gn_w, blk.tdf0.gn_b, nullptr, nullptr, e->H.f(), s));
    float *out = dest ? dest : next_free(cur, nullptr);
    float *tmp = dest ? next_free(cur, nullptr) : out;
    CHK(tdf_launch(e, blk.tdf1, e->H.f(), nullptr, tmp, M, blk.t, s, 0));
    CHK(gn_launch(e, tmp, B, blk.c, plane, blk.tdf1.gn_w, blk.tdf1.gn_b, cur, nullptr, out, s));
    cur = out;
    return ASX_OK;
  }
  // (A/B ASX_TDF_INPLACE=1: x + tdf(x) written over x where no skip copy is needed -- every output element depends on the
  // same element of x only)
  static const bool inplace = getenv("ASX_TDF_INPLACE") && atoi(getenv("ASX_TDF_INPLACE")) != 0;
  float *out = dest ? dest : (inplace ? cur : next_free(cur, nullptr));
  CHK(tdf_pair_launch(e, blk.tdf0, blk.tdf1, cur, e->H.f(), reinterpret_cast<int *>(e->HE.p), out, M, blk.t, s));
  cur = out;
  return ASX_OK;
}

static int net_forward_dev(asx_engine *e, const float *spec_in, float *spec_out, int B, hipStream_t s) {
  if (!e->net_ready) {
    set_err("net weights not committed");
    return ASX_ERR_STATE;
  }
  const int T = e->net.dim_t, F = e->net.dim_f;
  const int n = e->net.num_blocks / 2;
  const bool gn = e->net.norm == 1;
  float *cur = e->R[0].f();
  CHK(conv_launch(e, e->first, spec_in, nullptr, cur, B, T, F, s));
  if (gn) CHK(gn_launch(e, cur, B, e->net.g, (int64_t)T * F, e->first.gn_w, e->first.gn_b, nullptr, nullptr, cur, s));
  for (int i = 0; i < n; ++i) {
    CHK(block_forward(e, e->enc[i], cur, e->skip[i].f(), B, s));
    float *out = e->R[0].f();
    CHK(conv_launch(e, e->ds[i], cur, nullptr, out, B, e->enc[i].t, e->enc[i].f, s));
    if (gn)
      CHK(gn_launch(e, out, B, e->ds[i].cout, (int64_t)(e->enc[i].t / 2) * (e->enc[i].f / 2), e->ds[i].gn_w, e->ds[i].gn_b, nullptr, nullptr, out, s));
    cur = out;
  }
  CHK(block_forward(e, e->mid, cur, nullptr, B, s));
  for (int i = 0; i < n; ++i) {
    const Block &blk = e->dec[i];
    float *out = nullptr;
    for (int r = 0; r < 3; ++r)
      if (e->R[r].f() != cur) {
        out = e->R[r].f();
        break;
      }
    // input at (t/2, f/2) -> (t, f), multiplied by the matching encoder output (mdxnet.py:113)
    CHK(conv_launch(e, e->us[i], cur, gn ? nullptr : e->skip[n - 1 - i].f(), out, B, blk.t / 2, blk.f / 2, s));
    if (gn)   // norm + ReLU first, then `x *= ds_outputs[-i - 1]` (mdxnet.py:111-113)
      CHK(gn_launch(e, out, B, e->us[i].cout, (int64_t)blk.t * blk.f, e->us[i].gn_w, e->us[i].gn_b, nullptr, e->skip[n - 1 - i].f(), out, s));
    cur = out;
    CHK(block_forward(e, blk, cur, nullptr, B, s));
  }
  CHK(conv_launch(e, e->final_, cur, nullptr, spec_out, B, T, F, s));
  return ASX_OK;
}

// ----------------------------------------------------------------------------
// workspace
// ----------------------------------------------------------------------------
static int ensure_workspace(asx_engine *e, int Bchunks, bool need_net) {
  const int T = e->cfg.segment_size, Fq = e->cfg.dim_f;
  const int mult = (need_net && e->cfg.enable_denoise) ? 2 : 1;
  const size_t Bn = (size_t)Bchunks * mult;
  CHK(e->spec_in.ensure(Bn * 4 * T * Fq * 4));
  CHK(e->frames.ensure((size_t)Bchunks * 2 * T * e->cfg.n_fft * 4));
  if (need_net) {
    CHK(e->spec_out.ensure(Bn * 4 * T * Fq * 4));
    const size_t lvl0 = Bn * e->net.g * T * Fq * 4;
    for (int i = 0; i < 3; ++i) CHK(e->R[i].ensure(lvl0));
    CHK(e->H.ensure(Bn * e->net.g * T * (Fq / std::max(1, e->net.bn)) * 4 + 256));
    // exponent spans of H as a pair image (tdf_block): one int per (row, 128-column tile) at most
    CHK(e->HE.ensure(Bn * e->net.g * T * (size_t)((Fq / std::max(1, e->net.bn) + 127) / 128) * 4 + 256));
    if (e->net.norm == 1) CHK(e->gn_part.ensure(Bn * (size_t)e->net.g * (e->net.num_blocks / 2 + 1) * 16));
    const int n = e->net.num_blocks / 2;
    e->skip.resize(n);
    for (int i = 0; i < n; ++i) {
      const size_t c = (size_t)e->net.g * (i + 1), t = T >> i, f = Fq >> i;
      CHK(e->skip[i].ensure(Bn * c * t * f * 4));
    }
  }
  e->ws_batch = std::max(e->ws_batch, Bchunks);
  return ASX_OK;
}

