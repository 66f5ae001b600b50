// libasx.so -- engine + C ABI (include/asx.h) of the MI355X demix path.
//
// One engine = one GPU.  The engine owns: FFT tables, the packed ConvTDFNet
// weights, and a device workspace sized for `max_batch` chunks.  The chunk loop
// of MDXSeparator.demix (mdx_separator.py:348-392) is executed as batches of
// independent chunks: STFT -> net -> iSTFT -> fold+window per batch, then one
// gather pass folds all windowed chunks into the song (result / divider).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <map>
#include <string>
#include <mutex>
#include <set>
#include <vector>

#include "../../include/asx.h"
#include "kernels_fft.h"
#include "kernels_fft3.h"
#include "kernels_net.h"
#include "kernels_gemm2.h"
#include "kernels_gemm3.h"
#include "kernels_wino.h"
#ifdef ASX_EXPERIMENTAL_KERNELS
#include "kernels_winos.h"   // weight-stationary Winograd: measured slower (profiles/NOTES.md round 4)
#endif
#include "kernels_wino6.h"
#include "kernels_conv3h.h"
#include "kernels_updown6.h"
#ifndef ASX_TDF2_DEFAULT
#define ASX_TDF2_DEFAULT 1
#endif
#include "kernels_rof.h"
#include "kernels_ht.h"
#include "kernels_halo.h"
#include "kernels_hd.h"
#include "kernels_vr.h"
#include "kernels_ens.h"

using namespace asx;

#include "engine_core.h"
#include "engine_mdx.h"

// ----------------------------------------------------------------------------
// C ABI
// ----------------------------------------------------------------------------
extern "C" {

int asx_abi_version(void) { return ASX_ABI_VERSION; }
const char *asx_last_error(void) { return g_err; }

int asx_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

int asx_engine_create(int device, const asx_mdx_config *cfg, asx_engine **out) {
  REQUIRE(cfg && out, "asx_engine_create: null argument");
  REQUIRE(cfg->n_fft >= 8 && cfg->n_fft % 2 == 0, "n_fft must be even and >= 8 (got %d)", cfg->n_fft);
  REQUIRE(cfg->hop_length > 0 && cfg->segment_size >= 2, "bad hop_length/segment_size");
  REQUIRE(cfg->dim_f > 0 && cfg->dim_f <= cfg->n_fft / 2 + 1, "dim_f %d out of range for n_fft %d", cfg->dim_f,
          cfg->n_fft);
  REQUIRE(cfg->overlap >= 0.0 && cfg->overlap < 1.0, "overlap must be in [0,1)");
  REQUIRE(cfg->win_length >= 0 && cfg->win_length <= cfg->n_fft, "win_length %d out of range for n_fft %d", cfg->win_length, cfg->n_fft);
  const int64_t C = (int64_t)cfg->hop_length * (cfg->segment_size - 1);
  REQUIRE(C > cfg->n_fft / 2, "chunk_size %lld must exceed n_fft/2 (reflect padding)", (long long)C);
  REQUIRE(C - cfg->n_fft > 0, "chunk_size %lld must exceed n_fft (gen_size > 0)", (long long)C);
  FftPlan plan{};
  REQUIRE(make_plan(cfg->n_fft, &plan), "n_fft/2 = %d must factor into {2,3,5}", cfg->n_fft / 2);
  int ndev = 0;
  HIPCHK(hipGetDeviceCount(&ndev));
  REQUIRE(device >= 0 && device < ndev, "device %d not available (%d visible)", device, ndev);
  HIPCHK(hipSetDevice(device));
  asx_engine *e = new asx_engine();
  e->device = device;
  e->cfg = *cfg;
  e->plan = plan;
  std::vector<float> w;
  host_window(cfg->n_fft, w, cfg->win_length);
  std::vector<float> tw((size_t)cfg->n_fft * 2);
  for (int j = 0; j < cfg->n_fft; ++j) {
    const double ang = -2.0 * M_PI * (double)j / (double)cfg->n_fft;
    tw[2 * j] = (float)cos(ang);
    tw[2 * j + 1] = (float)sin(ang);
  }
  std::vector<float> env;
  host_env(cfg->n_fft, cfg->hop_length, cfg->segment_size, env, cfg->win_length);
  int rc = ASX_OK;
  if ((rc = e->d_window.ensure(w.size() * 4)) == ASX_OK && (rc = e->d_tw.ensure(tw.size() * 4)) == ASX_OK &&
      (rc = e->d_env.ensure(env.size() * 4)) == ASX_OK) {
    if (hipMemcpy(e->d_window.p, w.data(), w.size() * 4, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(e->d_tw.p, tw.data(), tw.size() * 4, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(e->d_env.p, env.data(), env.size() * 4, hipMemcpyHostToDevice) != hipSuccess) {
      set_err("table upload failed");
      rc = ASX_ERR_HIP;
    }
  }
  if (rc == ASX_OK && cfg->n_fft == f3::NFFT && cfg->hop_length == f3::HOP && !(getenv("ASX_FFT3") && atoi(getenv("ASX_FFT3")) == 0)) {
    std::vector<float> t3((size_t)(16 * 12 + 16 * f3::NB) * 2);
    for (int r = 0; r < 16; ++r)
      for (int k = 0; k < 12; ++k) {
        const double ang = -2.0 * M_PI * (double)(k * r) / 192.0;
        t3[(size_t)(r * 12 + k) * 2] = (float)cos(ang);
        t3[(size_t)(r * 12 + k) * 2 + 1] = (float)sin(ang);
      }
    for (int r = 0; r < 16; ++r)
      for (int j = 0; j < f3::NB; ++j) {
        const double ang = -2.0 * M_PI * (double)(j * r) / (double)f3::NH;
        t3[(size_t)(16 * 12 + r * f3::NB + j) * 2] = (float)cos(ang);
        t3[(size_t)(16 * 12 + r * f3::NB + j) * 2 + 1] = (float)sin(ang);
      }
    if ((rc = e->d_tw3.ensure(t3.size() * 4)) == ASX_OK) {
      if (hipMemcpy(e->d_tw3.p, t3.data(), t3.size() * 4, hipMemcpyHostToDevice) != hipSuccess) {
        set_err("table upload failed");
        rc = ASX_ERR_HIP;
      } else {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&f3::stft3_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  f3::STFT3_LDS_BYTES);
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&f3::istft3_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  f3::ISTFT3_LDS_BYTES);
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&f3::istft3p_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  f3::ISTFT3P_LDS_BYTES);
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&f3::istft3p_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  f3::ISTFT3P_LDS_BYTES);
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&f3::istft3p_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  f3::ISTFT3P_LDS_BYTES);
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&f3::istft3p_kernel<3>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  f3::ISTFT3P_LDS_BYTES);
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&f3::stft3p_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  f3::STFT3P_LDS_BYTES);
        e->fft3 = true;
        e->fft3p = !(getenv("ASX_FFT3P") && atoi(getenv("ASX_FFT3P")) == 0);
        if ((rc = e->d_hann3.ensure((size_t)C * 8)) == ASX_OK) {
          hipLaunchKernelGGL(f3::hann3_table_kernel, dim3((unsigned)((C + 255) / 256)), dim3(256), 0, nullptr, C,
                             reinterpret_cast<double *>(e->d_hann3.p));
          if (hipDeviceSynchronize() != hipSuccess) {
            set_err("hann table kernel failed");
            rc = ASX_ERR_HIP;
          }
        }
      }
    }
  }
  if (rc == ASX_OK) rc = e->d_zeros.ensure(256);
  if (rc == ASX_OK && hipMemset(e->d_zeros.p, 0, 256) != hipSuccess) {
    set_err("zero page init failed");
    rc = ASX_ERR_HIP;
  }
  if (rc == ASX_OK) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&stft_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)stft_lds(plan));
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&istft_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)istft_lds(plan));
  }
  if (rc != ASX_OK) {
    asx_engine_destroy(e);
    return rc;
  }
  *out = e;
  return ASX_OK;
}

static void v3_destroy(V3Net *n);
static void rof_destroy(RofNet *n);
static void ht_destroy(HtNet *n);
static void hd_destroy(HdNet *n);
static void hd_drop_clones(asx_engine *e);
static void vr_destroy(VrNet *n);
static void ens_destroy(EnsCtx *c);
static void free_conv(ConvLayer &L) {
  L.w.release();
  L.b.release();
  L.wu.release();
  L.wu2.release();
  L.wu3.release();
  L.wus.release();
  L.wu6.release();
  L.wu6h.release();
  L.w3h.release();
  L.wd6.release();
  L.wup6.release();
  L.gn_w.release();
  L.gn_b.release();
}
static void free_tdf(TdfLayer &L) {
  L.w.release();
  L.bias.release();
  L.scale.release();
  L.shift.release();
  L.gn_w.release();
  L.gn_b.release();
}
static void free_block(Block &b) {
  for (auto &c : b.tfc) free_conv(c);
  free_tdf(b.tdf0);
  free_tdf(b.tdf1);
}

void asx_engine_destroy(asx_engine *e) {
  w3_flush(e);
  if (!e) return;
  (void)hipSetDevice(e->device);
  for (auto &r : e->recs) {
    (void)hipEventDestroy(r.a);
    (void)hipEventDestroy(r.b);
  }
  if (e->div_ev) (void)hipEventDestroy(e->div_ev);
  e->sinc_tab.release();
  e->d_window.release();
  e->d_tw.release();
  e->d_env.release();
  e->d_tw3.release();
  e->d_hann3.release();
  e->seam3.release();
  e->d_zeros.release();
  free_conv(e->first);
  free_conv(e->final_);
  for (auto &b : e->enc) free_block(b);
  for (auto &b : e->dec) free_block(b);
  free_block(e->mid);
  for (auto &c : e->ds) free_conv(c);
  for (auto &c : e->us) free_conv(c);
  e->spec_in.release();
  e->spec_out.release();
  for (auto &r : e->R) r.release();
  e->H.release();
  e->frames.release();
  e->chunk_out.release();
  e->d_starts.release();
  e->d_nact.release();
  e->d_peak.release();
  e->d_demixed.release();
  e->d_div.release();
  for (auto &sk : e->skip) sk.release();
  if (e->v3) v3_destroy(e->v3);
  if (e->rof) rof_destroy(e->rof);
  hd_drop_clones(e);
  if (e->ht) ht_destroy(e->ht);
  if (e->hd) hd_destroy(e->hd);
  if (e->vr) vr_destroy(e->vr);
  if (e->ens) ens_destroy(e->ens);
  delete e;
}

// ---- weights ---------------------------------------------------------------
int asx_net_begin(asx_engine *e, const asx_net_config *cfg) {
  w3_flush(e);
  REQUIRE(e && cfg, "asx_net_begin: null argument");
  REQUIRE(cfg->dim_f == e->cfg.dim_f, "net dim_f %d != engine dim_f %d", cfg->dim_f, e->cfg.dim_f);
  REQUIRE(cfg->dim_t == e->cfg.segment_size, "net dim_t %d != segment_size %d", cfg->dim_t, e->cfg.segment_size);
  REQUIRE(cfg->k == 3, "only k=3 TFC kernels are supported (got %d)", cfg->k);
  REQUIRE(cfg->dim_c > 0 && cfg->g > 0 && cfg->l > 0 && cfg->bn >= -1, "bad net hyper-parameters");
  REQUIRE(cfg->norm == 0 || (cfg->norm == 1 && cfg->g % 2 == 0), "norm must be 0 (BatchNorm, folded by the host) or 1 (GroupNorm(2, c): even g)");
  REQUIRE(cfg->num_blocks >= 1 && cfg->num_blocks % 2 == 1, "num_blocks must be odd");
  const int n = cfg->num_blocks / 2;
  REQUIRE((cfg->dim_f % (1 << n)) == 0 && (cfg->dim_t % (1 << n)) == 0,
          "dim_f and dim_t must be divisible by 2^%d", n);
  REQUIRE(cfg->bn <= 0 || ((cfg->dim_f >> n) % cfg->bn) == 0, "dim_f / 2^n must be divisible by bn");
  e->net = *cfg;
  e->host_tensors.clear();
  e->net_begun = true;
  e->net_ready = false;
  return ASX_OK;
}

int asx_net_set_tensor(asx_engine *e, const char *name, const float *host, int64_t numel) {
  REQUIRE(e && name && host && numel > 0, "asx_net_set_tensor: bad argument");
  if (!e->net_begun) {
    set_err("asx_net_set_tensor before asx_net_begin");
    return ASX_ERR_STATE;
  }
  e->host_tensors[name].assign(host, host + numel);
  return ASX_OK;
}

static int get_tensor(asx_engine *e, const std::string &name, int64_t numel, const float **out, bool optional = false) {
  auto it = e->host_tensors.find(name);
  if (it == e->host_tensors.end()) {
    if (optional) {
      *out = nullptr;
      return ASX_OK;
    }
    set_err("missing tensor '%s'", name.c_str());
    return ASX_ERR_INVALID;
  }
  if ((int64_t)it->second.size() != numel) {
    set_err("tensor '%s': expected %lld elements, got %zu", name.c_str(), (long long)numel, it->second.size());
    return ASX_ERR_INVALID;
  }
  *out = it->second.data();
  return ASX_OK;
}

// GroupNorm affine of a layer: "<name>.gn_w" / "<name>.gn_b" [c] (asx_net_config.norm == 1)
static int load_gn(asx_engine *e, const std::string &name, int c, DevBuf &gw, DevBuf &gb) {
  const float *w, *b;
  CHK(get_tensor(e, name + ".gn_w", c, &w));
  CHK(get_tensor(e, name + ".gn_b", c, &b));
  CHK(gw.ensure((size_t)c * 4));
  CHK(gb.ensure((size_t)c * 4));
  HIPCHK(hipMemcpy(gw.p, w, (size_t)c * 4, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(gb.p, b, (size_t)c * 4, hipMemcpyHostToDevice));
  return ASX_OK;
}

static int build_block(asx_engine *e, Block &blk, const std::string &pre, int c, int t, int f) {
  const asx_net_config &n = e->net;
  const bool gn = n.norm == 1;
  blk.c = c;
  blk.t = t;
  blk.f = f;
  blk.tfc.resize(n.l);
  for (int j = 0; j < n.l; ++j) {
    const float *w, *b;
    const std::string nm = pre + ".tfc" + std::to_string(j);
    CHK(get_tensor(e, nm + ".w", (int64_t)c * c * 9, &w));
    CHK(get_tensor(e, nm + ".b", c, &b));
    CHK(conv_setup(blk.tfc[j], CK_3X3, c, c, gn ? 0 : 1));
    CHK(conv_pack(blk.tfc[j], w, b, e->winograd));
    if (gn) CHK(load_gn(e, nm, c, blk.tfc[j].gn_w, blk.tfc[j].gn_b));
  }
  if (n.bn < 0) return ASX_OK;                         // bn is None: no TDF branch (modules.py:52)
  const int fb = n.bn == 0 ? f : f / n.bn;             // bn == 0: ONE Linear(f, f) (modules.py:55-60)
  const float *w, *bias, *sc = nullptr, *sh = nullptr;
  CHK(get_tensor(e, pre + ".tdf0.w", (int64_t)fb * f, &w));
  CHK(get_tensor(e, pre + ".tdf0.bias", fb, &bias, !n.tdf_bias));
  if (!gn) {
    CHK(get_tensor(e, pre + ".tdf0.scale", c, &sc));
    CHK(get_tensor(e, pre + ".tdf0.shift", c, &sh));
  }
  CHK(tdf_pack(blk.tdf0, fb, f, c, w, bias, sc, sh));
  if (gn) CHK(load_gn(e, pre + ".tdf0", c, blk.tdf0.gn_w, blk.tdf0.gn_b));
  if (n.bn == 0) return ASX_OK;
  CHK(get_tensor(e, pre + ".tdf1.w", (int64_t)f * fb, &w));
  CHK(get_tensor(e, pre + ".tdf1.bias", f, &bias, !n.tdf_bias));
  if (!gn) {
    CHK(get_tensor(e, pre + ".tdf1.scale", c, &sc));
    CHK(get_tensor(e, pre + ".tdf1.shift", c, &sh));
  }
  CHK(tdf_pack(blk.tdf1, f, fb, c, w, bias, sc, sh));
  if (gn) CHK(load_gn(e, pre + ".tdf1", c, blk.tdf1.gn_w, blk.tdf1.gn_b));
  return ASX_OK;
}

int asx_net_commit(asx_engine *e) {
  w3_flush(e);
  REQUIRE(e, "asx_net_commit: null engine");
  if (!e->net_begun) {
    set_err("asx_net_commit before asx_net_begin");
    return ASX_ERR_STATE;
  }
  HIPCHK(hipSetDevice(e->device));
  const asx_net_config &n = e->net;
  const int nn = n.num_blocks / 2;
  const float *w, *b;
  CHK(get_tensor(e, "first.w", (int64_t)n.g * n.dim_c, &w));
  CHK(get_tensor(e, "first.b", n.g, &b));
  const bool gn = n.norm == 1;
  CHK(conv_setup(e->first, CK_1X1, n.dim_c, n.g, gn ? 0 : 1));
  CHK(conv_pack(e->first, w, b, e->winograd));
  if (gn) CHK(load_gn(e, "first", n.g, e->first.gn_w, e->first.gn_b));
  e->enc.assign(nn, Block());
  e->dec.assign(nn, Block());
  e->ds.assign(nn, ConvLayer());
  e->us.assign(nn, ConvLayer());
  int c = n.g, t = n.dim_t, f = n.dim_f;
  for (int i = 0; i < nn; ++i) {
    CHK(build_block(e, e->enc[i], "enc" + std::to_string(i), c, t, f));
    CHK(get_tensor(e, "ds" + std::to_string(i) + ".w", (int64_t)(c + n.g) * c * 4, &w));
    CHK(get_tensor(e, "ds" + std::to_string(i) + ".b", c + n.g, &b));
    CHK(conv_setup(e->ds[i], CK_DOWN, c, c + n.g, gn ? 0 : 1));
    CHK(conv_pack(e->ds[i], w, b, e->winograd));
    if (gn) CHK(load_gn(e, "ds" + std::to_string(i), c + n.g, e->ds[i].gn_w, e->ds[i].gn_b));
    c += n.g;
    t /= 2;
    f /= 2;
  }
  CHK(build_block(e, e->mid, "mid", c, t, f));
  for (int i = 0; i < nn; ++i) {
    CHK(get_tensor(e, "us" + std::to_string(i) + ".w", (int64_t)c * (c - n.g) * 4, &w));
    CHK(get_tensor(e, "us" + std::to_string(i) + ".b", c - n.g, &b));
    CHK(conv_setup(e->us[i], CK_UP, c, c - n.g, gn ? 0 : 1));
    CHK(conv_pack(e->us[i], w, b, e->winograd));
    if (gn) CHK(load_gn(e, "us" + std::to_string(i), c - n.g, e->us[i].gn_w, e->us[i].gn_b));
    c -= n.g;
    t *= 2;
    f *= 2;
    CHK(build_block(e, e->dec[i], "dec" + std::to_string(i), c, t, f));
  }
  CHK(get_tensor(e, "final.w", (int64_t)n.dim_c * n.g, &w));
  CHK(get_tensor(e, "final.b", n.dim_c, &b));
  CHK(conv_setup(e->final_, CK_1X1, n.g, n.dim_c, 0));
  CHK(conv_pack(e->final_, w, b, e->winograd));
  e->host_tensors.clear();
  e->net_ready = true;
  return ASX_OK;
}

double asx_net_flops(const asx_engine *e, int32_t batch) {
  if (!e || !e->net_begun) return 0.0;
  const asx_net_config &d = e->net;
  const int n = d.num_blocks / 2;
  double fl = 2.0 * d.dim_c * d.g * (double)d.dim_t * d.dim_f;
  auto block = [&](double c, double t, double f) {
    const double tdf = d.bn > 0 ? 2.0 * 2.0 * c * t * f * (f / d.bn) : (d.bn == 0 ? 2.0 * c * t * f * f : 0.0);
    return d.l * 2.0 * 9.0 * c * c * t * f + tdf;
  };
  double c = d.g, t = d.dim_t, f = d.dim_f;
  for (int i = 0; i < n; ++i) {
    fl += block(c, t, f);
    fl += 2.0 * 4.0 * c * (c + d.g) * (t / 2) * (f / 2);
    c += d.g;
    t /= 2;
    f /= 2;
  }
  fl += block(c, t, f);
  for (int i = 0; i < n; ++i) {
    fl += 2.0 * 4.0 * c * (c - d.g) * t * f;
    c -= d.g;
    t *= 2;
    f *= 2;
    fl += block(c, t, f);
  }
  fl += 2.0 * d.g * d.dim_c * (double)d.dim_t * d.dim_f;
  return fl * batch;
}

}  // extern "C" (the engine headers below define templates)
#include "engine_v3.h"
#include "engine_rof.h"
#include "engine_ht.h"
#include "engine_hd.h"
#include "engine_vr.h"
#include "engine_ens.h"
extern "C" {

// ---- plan ------------------------------------------------------------------
// torch.stft / istft `window=`: a table of n_fft floats (a window function evaluated over win_length, zero padded to n_fft at both
// ends like torch does) replacing the periodic Hann built at engine creation -- BSRoformer's `stft_window_fn` (bs_roformer.py:333, 386).
int asx_set_stft_window(asx_engine *e, const float *window_host, int32_t n) {
  REQUIRE(e && window_host, "asx_set_stft_window: null argument");
  REQUIRE(n == e->cfg.n_fft, "asx_set_stft_window: %d values for n_fft %d", n, e->cfg.n_fft);
  HIPCHK(hipSetDevice(e->device));
  HIPCHK(hipDeviceSynchronize());
  e->custom_window.assign(window_host, window_host + n);
  std::vector<float> env;
  host_env(n, e->cfg.hop_length, e->cfg.segment_size, env, e->cfg.win_length, &e->custom_window);
  HIPCHK(hipMemcpy(e->d_window.p, window_host, (size_t)n * 4, hipMemcpyHostToDevice));
  CHK(e->d_env.ensure(env.size() * 4));
  HIPCHK(hipMemcpy(e->d_env.p, env.data(), env.size() * 4, hipMemcpyHostToDevice));
  if (e->rof) e->rof->win_synth.release();             // rebuilt from the new table at the next forward
  return ASX_OK;
}

int asx_plan_query(const asx_engine *e, int64_t N, uint32_t flags, asx_plan *out) {
  REQUIRE(e && out, "asx_plan_query: null argument");
  REQUIRE(N >= 1, "n_samples must be >= 1 (an empty mix raises in the reference, common_separator.py:267)");
  const bool match = (flags & ASX_FLAG_MATCH_MIX) != 0;
  // python: overlap is a float (double); 0.02 for the match-mix pass (mdx_separator.py:311)
  const double overlap = match ? 0.02 : e->cfg.overlap;
  asx_plan p{};
  p.n_samples = N;
  p.trim = e->cfg.n_fft / 2;
  p.chunk_size = (int64_t)e->cfg.hop_length * (e->cfg.segment_size - 1);
  p.gen_size = p.chunk_size - 2 * p.trim;
  p.pad = p.gen_size + p.trim - (N % p.gen_size);
  p.padded_len = p.trim + N + p.pad;
  p.step = (int64_t)((1.0 - overlap) * (double)p.chunk_size);  // int() truncation (mdx_separator.py:335)
  REQUIRE(p.step >= 1, "overlap too close to 1: step == 0");
  p.n_chunks = (int32_t)((p.padded_len + p.step - 1) / p.step);
  p.n_frames = e->cfg.segment_size;
  *out = p;
  return ASX_OK;
}

// ---- chunk batches -----------------------------------------------------------
__global__ void chunk_table_kernel(int k0, int nk, int64_t step, int64_t chunk, int64_t padded_len, int win,
                                   int64_t *__restrict__ starts, int64_t *__restrict__ nact) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nk) return;
  const int64_t st = (int64_t)(k0 + i) * step;
  const int64_t na = chunk < padded_len - st ? chunk : padded_len - st;
  starts[i] = st;
  nact[i] = win ? na : -1;
}

static bool windowed_mode(const asx_engine *e, uint32_t flags) {
  return (flags & ASX_FLAG_MATCH_MIX) ? true : (e->cfg.overlap != 0.0);
}

int asx_demix_chunks_dev(asx_engine *e, const float *mix_dev, int64_t N, int32_t k0, int32_t k1,
                         float *chunk_out_dev, uint32_t flags, void *stream) {
  REQUIRE(e && mix_dev && chunk_out_dev, "asx_demix_chunks_dev: null argument");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  HIPCHK(hipSetDevice(e->device));
  asx_plan p;
  CHK(asx_plan_query(e, N, flags, &p));
  REQUIRE(k0 >= 0 && k1 <= p.n_chunks && k0 <= k1, "chunk range [%d,%d) outside [0,%d)", k0, k1, p.n_chunks);
  const bool match = (flags & ASX_FLAG_MATCH_MIX) != 0;
  const bool need_net = !match;
  if (need_net && !e->net_ready) {
    set_err("asx_demix: net weights not committed");
    return ASX_ERR_STATE;
  }
  const int nk = k1 - k0;
  if (nk == 0) return ASX_OK;
  const int maxB = pick_batch(e);
  const int nbatch = (nk + maxB - 1) / maxB;
  const int per = (nk + nbatch - 1) / nbatch;
  CHK(ensure_workspace(e, per, need_net));
  // chunk tables, built on the device: start of chunk k = k * step, active length = min(chunk, L - start), or negative
  // for "no chunk window" (overlap == 0, mdx_separator.py:389-390).  No host copy, no host synchronisation: the call
  // only enqueues work on `stream` (hipGraph-capturable, and a gather of step k can overlap the compute of step k + 1).
  const bool win = windowed_mode(e, flags);
  CHK(e->d_starts.ensure((size_t)nk * 8));
  CHK(e->d_nact.ensure((size_t)nk * 8));
  hipLaunchKernelGGL(chunk_table_kernel, dim3((unsigned)((nk + 255) / 256)), dim3(256), 0, s, k0, nk, p.step, p.chunk_size,
                     p.padded_len, win ? 1 : 0, reinterpret_cast<int64_t *>(e->d_starts.p), reinterpret_cast<int64_t *>(e->d_nact.p));
  HIPCHK(hipGetLastError());
  const int T = e->cfg.segment_size;
  const int64_t C = p.chunk_size;
  const size_t spec_elems = (size_t)4 * T * e->cfg.dim_f;
  for (int b0 = 0; b0 < nk; b0 += per) {
    const int B = std::min(per, nk - b0);
    const int64_t *ds = reinterpret_cast<const int64_t *>(e->d_starts.p) + b0;
    const int64_t *dn = reinterpret_cast<const int64_t *>(e->d_nact.p) + b0;
    CHK(stft_launch(e, mix_dev, ds, N, B, C, T, e->spec_in.f(), 1, 3, 1.0f, s));
    const float *spec_final = e->spec_in.f();
    int combine = 0;
    if (need_net) {
      int Bn = B;
      if (e->cfg.enable_denoise) {
        // second half of the batch: the negated spectrum (mdx_separator.py:437)
        CHK(stft_launch(e, mix_dev, ds, N, B, C, T, e->spec_in.f() + (size_t)B * spec_elems, 1, 3, -1.0f, s));
        Bn = 2 * B;
        combine = B;
      }
      CHK(net_forward_dev(e, e->spec_in.f(), e->spec_out.f(), Bn, s));
      spec_final = e->spec_out.f();
    }
    CHK(istft_ola_launch(e, spec_final, B, T, combine, dn, C, chunk_out_dev + (size_t)b0 * 2 * C, s));
  }
  return ASX_OK;
}

int asx_finalize_dev(asx_engine *e, const float *chunk_out_dev, int64_t N, float *out_dev, uint32_t flags,
                     void *stream) {
  REQUIRE(e && chunk_out_dev && out_dev, "asx_finalize_dev: null argument");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  HIPCHK(hipSetDevice(e->device));
  asx_plan p;
  CHK(asx_plan_query(e, N, flags, &p));
  const int win = windowed_mode(e, flags) ? 1 : 0;
  const double bytes = 4.0 * ((double)p.n_chunks * 2 * p.chunk_size + 2.0 * N);
  const double *hann = (e->fft3 && e->d_hann3.p) ? reinterpret_cast<const double *>(e->d_hann3.p) : nullptr;
  static const bool fin4 = !(getenv("ASX_FINALIZE4") && atoi(getenv("ASX_FINALIZE4")) == 0);
  // vector path: the divider (input-independent) comes from a table built once per plan; needs 4-sample alignment of the
  // chunk geometry and 16-byte aligned buffers
  if (fin4 && p.chunk_size % 4 == 0 && p.step % 4 == 0 && p.trim % 4 == 0 && N % 4 == 0 && p.chunk_size + (int64_t)p.trim >= 0 &&
      (((uintptr_t)chunk_out_dev | (uintptr_t)out_dev) & 15) == 0) {
    const DivKey key{N, p.n_chunks, p.chunk_size, p.step, p.padded_len, p.trim, win, hann != nullptr ? 1 : 0};
    if (!(e->div_key == key) || !e->d_div.p) {
      CHK(e->d_div.ensure((size_t)N * 4));
      CHK(timed(e, ASX_PROF_MISC, 0.0, 4.0 * N, s, [&]() {
        hipLaunchKernelGGL(finalize_div_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, s, p.n_chunks, p.chunk_size, p.step,
                           p.padded_len, p.trim, N, win, e->d_div.f(), hann);
      }));
      e->div_key = key;
      // the table is shared by later calls: a call on a different stream must be ordered behind this build
      if (e->div_ev == nullptr) HIPCHK(hipEventCreateWithFlags(&e->div_ev, hipEventDisableTiming));
      hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
      (void)hipStreamIsCapturing(s, &cap);
      if (cap == hipStreamCaptureStatusNone) {
        HIPCHK(hipEventRecord(e->div_ev, s));
        e->div_stream = s;
      } else {
        e->div_stream = s;          // built inside a capture: the graph owns the ordering, nothing to wait for outside it
      }
    } else if (s != e->div_stream && e->div_ev != nullptr && hipEventQuery(e->div_ev) == hipErrorNotReady) {
      HIPCHK(hipStreamWaitEvent(s, e->div_ev, 0));
    }
    return timed(e, ASX_PROF_FINALIZE, 0.0, bytes, s, [&]() {
      hipLaunchKernelGGL(finalize4_kernel, dim3((unsigned)((N / 4 + 255) / 256), 2), dim3(256), 0, s, chunk_out_dev, p.n_chunks,
                         p.chunk_size, p.step, p.padded_len, p.trim, N, e->d_div.f(), out_dev);
    });
  }
  return timed(e, ASX_PROF_FINALIZE, 0.0, bytes, s, [&]() {
    hipLaunchKernelGGL(finalize_kernel, dim3((unsigned)((N + 255) / 256), 2), dim3(256), 0, s, chunk_out_dev,
                       p.n_chunks, p.chunk_size, p.step, p.padded_len, p.trim, N, win, out_dev, hann);
  });
}

int asx_demix_dev(asx_engine *e, const float *mix_dev, int64_t N, float *out_dev, uint32_t flags, void *stream) {
  REQUIRE(e && mix_dev && out_dev, "asx_demix_dev: null argument");
  HIPCHK(hipSetDevice(e->device));
  asx_plan p;
  CHK(asx_plan_query(e, N, flags, &p));
  CHK(e->chunk_out.ensure((size_t)p.n_chunks * 2 * p.chunk_size * 4));
  CHK(asx_demix_chunks_dev(e, mix_dev, N, 0, p.n_chunks, e->chunk_out.f(), flags, stream));
  CHK(asx_finalize_dev(e, e->chunk_out.f(), N, out_dev, flags, stream));
  return ASX_OK;
}

int asx_demix(asx_engine *e, const float *mix_host, int64_t N, float *out_host, uint32_t flags) {
  REQUIRE(e && mix_host && out_host, "asx_demix: null argument");
  REQUIRE(N >= 1, "n_samples must be >= 1");
  HIPCHK(hipSetDevice(e->device));
  DevBuf dmix, dout;
  int rc = dmix.ensure((size_t)2 * N * 4);
  if (rc == ASX_OK) rc = dout.ensure((size_t)2 * N * 4);
  if (rc == ASX_OK && hipMemcpy(dmix.p, mix_host, (size_t)2 * N * 4, hipMemcpyHostToDevice) != hipSuccess) {
    set_err("asx_demix: H2D copy failed");
    rc = ASX_ERR_HIP;
  }
  if (rc == ASX_OK) rc = asx_demix_dev(e, dmix.f(), N, dout.f(), flags, nullptr);
  if (rc == ASX_OK && hipStreamSynchronize(nullptr) != hipSuccess) {
    set_err("asx_demix: device execution failed: %s", hipGetErrorString(hipGetLastError()));
    rc = ASX_ERR_HIP;
  }
  if (rc == ASX_OK && hipMemcpy(out_host, dout.p, (size_t)2 * N * 4, hipMemcpyDeviceToHost) != hipSuccess) {
    set_err("asx_demix: D2H copy failed");
    rc = ASX_ERR_HIP;
  }
  dmix.release();
  dout.release();
  return rc;
}

// ---- stem algebra ------------------------------------------------------------
int asx_separate_dev(asx_engine *e, float *mix_dev, int64_t N, float max_peak, float min_peak, int32_t has_min,
                     float compensate, float *primary_dev, float *secondary_dev, void *stream) {
  REQUIRE(e && mix_dev && primary_dev && secondary_dev, "asx_separate_dev: null argument");
  REQUIRE(N >= 1, "n_samples must be >= 1");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  HIPCHK(hipSetDevice(e->device));
  CHK(e->d_peak.ensure(256));
  CHK(e->d_demixed.ensure((size_t)2 * N * 4));
  unsigned int *pk = reinterpret_cast<unsigned int *>(e->d_peak.p);
  HIPCHK(hipMemsetAsync(pk, 0, 4, s));
  const int64_t n2 = 2 * N;
  const unsigned nb = (unsigned)std::min<int64_t>((n2 + 255) / 256, 2048);
  CHK(timed(e, ASX_PROF_MISC, 0.0, 4.0 * n2, s, [&]() {
    hipLaunchKernelGGL(absmax_kernel, dim3(nb), dim3(256), 0, s, mix_dev, n2, pk);
  }));
  CHK(timed(e, ASX_PROF_MISC, 0.0, 8.0 * n2, s, [&]() {
    hipLaunchKernelGGL(normalize_kernel, dim3(nb), dim3(256), 0, s, mix_dev, n2, pk, max_peak, min_peak, has_min);
  }));
  CHK(asx_demix_dev(e, mix_dev, N, e->d_demixed.f(), 0, stream));
  CHK(timed(e, ASX_PROF_MISC, 0.0, 4.0 * 4 * n2, s, [&]() {
    hipLaunchKernelGGL(stems_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, s, e->d_demixed.f(), mix_dev, N,
                       pk, compensate, primary_dev, secondary_dev);
  }));
  return ASX_OK;
}

int asx_separate(asx_engine *e, float *mix_host, int64_t N, float max_peak, float min_peak, int32_t has_min,
                 float compensate, float *primary_host, float *secondary_host) {
  REQUIRE(e && mix_host && primary_host && secondary_host, "asx_separate: null argument");
  REQUIRE(N >= 1, "n_samples must be >= 1");
  HIPCHK(hipSetDevice(e->device));
  DevBuf dmix, dp, ds;
  const size_t bytes = (size_t)2 * N * 4;
  int rc = dmix.ensure(bytes);
  if (rc == ASX_OK) rc = dp.ensure(bytes);
  if (rc == ASX_OK) rc = ds.ensure(bytes);
  if (rc == ASX_OK && hipMemcpy(dmix.p, mix_host, bytes, hipMemcpyHostToDevice) != hipSuccess) {
    set_err("asx_separate: H2D copy failed");
    rc = ASX_ERR_HIP;
  }
  if (rc == ASX_OK)
    rc = asx_separate_dev(e, dmix.f(), N, max_peak, min_peak, has_min, compensate, dp.f(), ds.f(), nullptr);
  if (rc == ASX_OK && hipStreamSynchronize(nullptr) != hipSuccess) {
    set_err("asx_separate: device execution failed: %s", hipGetErrorString(hipGetLastError()));
    rc = ASX_ERR_HIP;
  }
  if (rc == ASX_OK && (hipMemcpy(mix_host, dmix.p, bytes, hipMemcpyDeviceToHost) != hipSuccess ||
                       hipMemcpy(primary_host, dp.p, bytes, hipMemcpyDeviceToHost) != hipSuccess ||
                       hipMemcpy(secondary_host, ds.p, bytes, hipMemcpyDeviceToHost) != hipSuccess)) {
    set_err("asx_separate: D2H copy failed");
    rc = ASX_ERR_HIP;
  }
  dmix.release();
  dp.release();
  ds.release();
  return rc;
}

// spec_utils.normalize(wave, max_peak, min_peak) (uvr_lib_v5/spec_utils.py:99-115) on any float32 array, in place:
// maxv = |wave|.max(); > max_peak: *= max_peak / maxv; else (min_peak given and maxv < min_peak): *= min_peak / maxv.
int asx_normalize(asx_engine *e, float *wave_host, int64_t numel, float max_peak, float min_peak, int32_t has_min, float *peak_before) {
  REQUIRE(e && wave_host && numel >= 1, "asx_normalize: bad argument");
  HIPCHK(hipSetDevice(e->device));
  CHK(e->d_peak.ensure(256));
  DevBuf d;
  int rc = d.ensure((size_t)numel * 4);
  unsigned int *pk = reinterpret_cast<unsigned int *>(e->d_peak.p);
  float maxv = 0.f;
  const unsigned nb = (unsigned)std::min<int64_t>((numel + 255) / 256, 2048);
  if (rc == ASX_OK && (hipMemcpy(d.p, wave_host, (size_t)numel * 4, hipMemcpyHostToDevice) != hipSuccess ||
                       hipMemsetAsync(pk, 0, 4, nullptr) != hipSuccess)) {
    set_err("asx_normalize: H2D copy failed");
    rc = ASX_ERR_HIP;
  }
  if (rc == ASX_OK) {
    hipLaunchKernelGGL(absmax_kernel, dim3(nb), dim3(256), 0, nullptr, d.f(), numel, pk);
    hipLaunchKernelGGL(normalize_kernel, dim3(nb), dim3(256), 0, nullptr, d.f(), numel, pk, max_peak, min_peak, has_min);
    if (hipGetLastError() != hipSuccess || hipMemcpy(&maxv, pk, 4, hipMemcpyDeviceToHost) != hipSuccess ||
        hipMemcpy(wave_host, d.p, (size_t)numel * 4, hipMemcpyDeviceToHost) != hipSuccess) {
      set_err("asx_normalize: device execution failed");
      rc = ASX_ERR_HIP;
    }
  }
  if (peak_before) *peak_before = maxv;
  d.release();
  return rc;
}

// the same on an array that is already in HBM (MDXCSeparator.separate normalises the mix and every stem, mdxc_separator.py:147,
// 170-190): in place, only enqueues work
int asx_normalize_dev(asx_engine *e, float *wave_dev, int64_t numel, float max_peak, float min_peak, int32_t has_min, void *stream) {
  REQUIRE(e && wave_dev && numel >= 1, "asx_normalize_dev: bad argument");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  HIPCHK(hipSetDevice(e->device));
  CHK(e->d_peak.ensure(256));
  unsigned int *pk = reinterpret_cast<unsigned int *>(e->d_peak.p) + 2;   // its own word (0: separate / pcm16, 1: decode)
  HIPCHK(hipMemsetAsync(pk, 0, 4, s));
  const unsigned nb = (unsigned)std::min<int64_t>((numel + 255) / 256, 2048);
  CHK(timed(e, ASX_PROF_MISC, 0.0, 4.0 * numel, s, [&]() { hipLaunchKernelGGL(absmax_kernel, dim3(nb), dim3(256), 0, s, wave_dev, numel, pk); }));
  return timed(e, ASX_PROF_MISC, 0.0, 8.0 * numel, s, [&]() {
    hipLaunchKernelGGL(normalize_kernel, dim3(nb), dim3(256), 0, s, wave_dev, numel, pk, max_peak, min_peak, has_min);
  });
}

// out = mix - stem (the residual stem of a single-target MDXC / Roformer model, mdxc_separator.py:406-468), float32
int asx_residual_dev(asx_engine *e, const float *mix_dev, const float *stem_dev, float *out_dev, int64_t numel, void *stream) {
  REQUIRE(e && mix_dev && stem_dev && out_dev && numel >= 1, "asx_residual_dev: bad argument");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  HIPCHK(hipSetDevice(e->device));
  const unsigned nb = (unsigned)std::min<int64_t>((numel + 255) / 256, 4096);
  return timed(e, ASX_PROF_MISC, 0.0, 12.0 * numel, s, [&]() {
    hipLaunchKernelGGL(residual_kernel, dim3(nb), dim3(256), 0, s, mix_dev, stem_dev, numel, out_dev);
  });
}

// librosa.resample(res_type="sinc_fastest") for an arbitrary ratio (engine_vr.h resample_sinc_dev)
int asx_resample_sinc_dev(asx_engine *e, const float *x_dev, int32_t channels, int64_t n_in, double ratio, int32_t mono_calls,
                          float *y_dev, int64_t n_out, void *stream) {
  REQUIRE(e && x_dev && y_dev && channels >= 1 && channels <= 65535 && n_in >= 1 && n_out >= 1, "asx_resample_sinc_dev: bad argument");
  REQUIRE(ratio > 1.0 / 256 && ratio < 256.0, "asx_resample_sinc_dev: ratio %g outside libsamplerate's (1/256, 256)", ratio);
  HIPCHK(hipSetDevice(e->device));
  return resample_sinc_dev(e, e->sinc_tab, x_dev, channels, n_in, ratio, mono_calls, y_dev, n_out, reinterpret_cast<hipStream_t>(stream));
}

// ---- stage hooks -------------------------------------------------------------
static int to_dev(DevBuf &d, const float *h, size_t n) {
  CHK(d.ensure(n * 4));
  HIPCHK(hipMemcpy(d.p, h, n * 4, hipMemcpyHostToDevice));
  return ASX_OK;
}
static int to_host(float *h, const DevBuf &d, size_t n) {
  HIPCHK(hipDeviceSynchronize());
  HIPCHK(hipMemcpy(h, d.p, n * 4, hipMemcpyDeviceToHost));
  return ASX_OK;
}
struct BufGuard {
  std::vector<DevBuf *> v;
  ~BufGuard() {
    for (auto *b : v) b->release();
  }
};

static int transpose_launch(const float *in, float *out, int planes, int rows, int cols, hipStream_t s) {
  hipLaunchKernelGGL(transpose_last2_kernel, dim3((cols + 31) / 32, (rows + 31) / 32, planes), dim3(32, 8), 0, s, in,
                     out, rows, cols);
  HIPCHK(hipGetLastError());
  return ASX_OK;
}

// write_audio_pydub's array work (common_separator.py:309-337): stem [2, N] planar float32 -> normalised int16 [N, 2];
// *peak_after = max |stem| after normalisation (the caller skips near-silent stems: < 1e-6, :312-315)
int asx_pcm16_dev(asx_engine *e, const float *stem_dev, int64_t N, float max_peak, float min_peak, int32_t has_min, int16_t *pcm_dev,
                  float *peak_after, void *stream) {
  REQUIRE(e && stem_dev && pcm_dev && N >= 1, "asx_pcm16_dev: bad argument");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  HIPCHK(hipSetDevice(e->device));
  CHK(e->d_peak.ensure(256));
  unsigned int *pk = reinterpret_cast<unsigned int *>(e->d_peak.p);
  HIPCHK(hipMemsetAsync(pk, 0, 4, s));
  const int64_t n2 = 2 * N;
  const unsigned nb = (unsigned)std::min<int64_t>((n2 + 255) / 256, 2048);
  CHK(timed(e, ASX_PROF_MISC, 0.0, 4.0 * n2, s, [&]() { hipLaunchKernelGGL(absmax_kernel, dim3(nb), dim3(256), 0, s, stem_dev, n2, pk); }));
  CHK(timed(e, ASX_PROF_MISC, 0.0, 6.0 * n2, s, [&]() {
    hipLaunchKernelGGL(pcm16_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, s, stem_dev, N, pk, max_peak, min_peak, has_min,
                       reinterpret_cast<short *>(pcm_dev));
  }));
  if (peak_after) {
    float maxv = 0.f;
    HIPCHK(hipMemcpyAsync(&maxv, pk, 4, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    float scale = 1.0f;
    if (maxv > max_peak) scale = max_peak / maxv;
    else if (has_min && maxv < min_peak) scale = min_peak / maxv;
    *peak_after = maxv * scale;
  }
  return ASX_OK;
}

int asx_pcm16_rows_dev(asx_engine *e, const float *stem_rows_dev, int64_t N, float max_peak, float min_peak, int32_t has_min,
                       int16_t *pcm_dev, float *peak_after, void *stream) {
  REQUIRE(e && stem_rows_dev && pcm_dev && N >= 1, "asx_pcm16_rows_dev: bad argument");
  REQUIRE((((uintptr_t)stem_rows_dev) & 15) == 0 && (((uintptr_t)pcm_dev) & 7) == 0, "asx_pcm16_rows_dev: stem must be 16-byte and pcm 8-byte aligned");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  HIPCHK(hipSetDevice(e->device));
  CHK(e->d_peak.ensure(256));
  unsigned int *pk = reinterpret_cast<unsigned int *>(e->d_peak.p);
  HIPCHK(hipMemsetAsync(pk, 0, 4, s));
  const int64_t n2 = 2 * N;
  const unsigned nb = (unsigned)std::min<int64_t>((n2 + 255) / 256, 2048);
  CHK(timed(e, ASX_PROF_MISC, 0.0, 4.0 * n2, s, [&]() { hipLaunchKernelGGL(absmax_kernel, dim3(nb), dim3(256), 0, s, stem_rows_dev, n2, pk); }));
  CHK(timed(e, ASX_PROF_MISC, 0.0, 6.0 * n2, s, [&]() {
    hipLaunchKernelGGL(pcm16_rows_kernel, dim3((unsigned)((n2 / 4 + 256) / 256)), dim3(256), 0, s, stem_rows_dev, n2, pk, max_peak, min_peak,
                       has_min, reinterpret_cast<short *>(pcm_dev));
  }));
  if (peak_after) {
    float maxv = 0.f;
    HIPCHK(hipMemcpyAsync(&maxv, pk, 4, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    float scale = 1.0f;
    if (maxv > max_peak) scale = max_peak / maxv;
    else if (has_min && maxv < min_peak) scale = min_peak / maxv;
    *peak_after = maxv * scale;
  }
  return ASX_OK;
}

int asx_pcm_decode_dev(asx_engine *e, const void *raw_dev, int64_t frames, int32_t channels, int32_t sample_format, float *mix_dev,
                       float *peak, void *stream) {
  REQUIRE(e && raw_dev && mix_dev && frames >= 1, "asx_pcm_decode_dev: bad argument");
  REQUIRE(channels >= 1 && channels <= 64, "asx_pcm_decode_dev: %d channels", channels);
  REQUIRE(sample_format == 16 || sample_format == 24 || sample_format == 32 || sample_format == 0x120,
          "asx_pcm_decode_dev: sample format %d (16, 24, 32 = integer PCM bits, 0x120 = IEEE float32)", sample_format);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  HIPCHK(hipSetDevice(e->device));
  CHK(e->d_peak.ensure(256));
  unsigned int *pk = reinterpret_cast<unsigned int *>(e->d_peak.p) + 1;   // its own word: asx_separate_dev uses word 0
  HIPCHK(hipMemsetAsync(pk, 0, 4, s));
  const unsigned nb = (unsigned)std::min<int64_t>((frames + 255) / 256, 4096);
  CHK(timed(e, ASX_PROF_MISC, 0.0, (double)frames * (channels * (sample_format & 0xff) / 8 + 8.0), s, [&]() {
    hipLaunchKernelGGL(pcm_decode_kernel, dim3(nb), dim3(256), 0, s, reinterpret_cast<const unsigned char *>(raw_dev), frames, (int)channels,
                       (int)sample_format, mix_dev, pk);
  }));
  if (peak) {
    HIPCHK(hipMemcpyAsync(peak, pk, 4, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
  }
  return ASX_OK;
}

int asx_pcm16(asx_engine *e, const float *stem_host, int64_t N, float max_peak, float min_peak, int32_t has_min, int16_t *pcm_host,
              float *peak_after) {
  REQUIRE(e && stem_host && pcm_host && N >= 1, "asx_pcm16: bad argument");
  HIPCHK(hipSetDevice(e->device));
  DevBuf ds, dp;
  BufGuard g{{&ds, &dp}};
  CHK(to_dev(ds, stem_host, (size_t)2 * N));
  CHK(dp.ensure((size_t)2 * N * 2));
  CHK(asx_pcm16_dev(e, ds.f(), N, max_peak, min_peak, has_min, reinterpret_cast<int16_t *>(dp.p), peak_after, nullptr));
  HIPCHK(hipDeviceSynchronize());
  HIPCHK(hipMemcpy(pcm_host, dp.p, (size_t)2 * N * 2, hipMemcpyDeviceToHost));
  return ASX_OK;
}

int asx_resample_sinc(asx_engine *e, const float *x_host, int32_t channels, int64_t n_in, double ratio, int32_t mono_calls,
                      float *y_host, int64_t n_out) {
  REQUIRE(e && x_host && y_host && channels >= 1 && n_in >= 1 && n_out >= 1, "asx_resample_sinc: bad argument");
  HIPCHK(hipSetDevice(e->device));
  DevBuf dx, dy;
  BufGuard g{{&dx, &dy}};
  CHK(to_dev(dx, x_host, (size_t)channels * n_in));
  CHK(dy.ensure((size_t)channels * n_out * 4));
  CHK(asx_resample_sinc_dev(e, dx.f(), channels, n_in, ratio, mono_calls, dy.f(), n_out, nullptr));
  return to_host(y_host, dy, (size_t)channels * n_out);
}

int asx_stft(asx_engine *e, const float *wave_host, int32_t B, int64_t C, float *spec_host) {
  REQUIRE(e && wave_host && spec_host && B > 0, "asx_stft: bad argument");
  REQUIRE(C > e->cfg.n_fft / 2, "asx_stft: n_time %lld must exceed n_fft/2 (reflect padding)", (long long)C);
  HIPCHK(hipSetDevice(e->device));
  const int T = (int)(C / e->cfg.hop_length) + 1;
  const size_t ns = (size_t)B * 4 * e->cfg.dim_f * T;
  DevBuf dw, ds;
  BufGuard g{{&dw, &ds}};
  CHK(to_dev(dw, wave_host, (size_t)B * 2 * C));
  CHK(ds.ensure(ns * 4));
  CHK(stft_launch(e, dw.f(), nullptr, -1, B, C, T, ds.f(), 0, 0, 1.0f, nullptr));
  CHK(to_host(spec_host, ds, ns));
  return ASX_OK;
}

int asx_istft(asx_engine *e, const float *spec_host, int32_t B, int32_t T, float *wave_host) {
  REQUIRE(e && spec_host && wave_host && B > 0 && T >= 1, "asx_istft: bad argument");
  HIPCHK(hipSetDevice(e->device));
  const int n = e->cfg.n_fft, hop = e->cfg.hop_length;
  const int64_t C = (int64_t)hop * (T - 1);
  REQUIRE(C > 0, "asx_istft: need at least 2 frames");
  const size_t ns = (size_t)B * 4 * e->cfg.dim_f * T;
  DevBuf dsp, dfr, denv, dout;
  BufGuard g{{&dsp, &dfr, &denv, &dout}};
  CHK(to_dev(dsp, spec_host, ns));
  CHK(dfr.ensure((size_t)B * 2 * T * n * 4));
  CHK(dout.ensure((size_t)B * 2 * C * 4));
  std::vector<float> env;
  host_env(n, hop, T, env, e->cfg.win_length, &e->custom_window);
  CHK(to_dev(denv, env.data(), env.size()));
  CHK(istft_launch(e, dsp.f(), B, T, 0, 0, dfr.f(), nullptr));
  CHK(ola_launch(e, dfr.f(), denv.f(), nullptr, B, T, C, dout.f(), nullptr));
  CHK(to_host(wave_host, dout, (size_t)B * 2 * C));
  return ASX_OK;
}

int asx_net_forward(asx_engine *e, const float *spec_host, int32_t B, float *out_host) {
  REQUIRE(e && spec_host && out_host && B > 0, "asx_net_forward: bad argument");
  if (!e->net_ready) {
    set_err("asx_net_forward: net weights not committed");
    return ASX_ERR_STATE;
  }
  HIPCHK(hipSetDevice(e->device));
  const int T = e->net.dim_t, Fq = e->net.dim_f, dc = e->net.dim_c;
  REQUIRE(dc == 4, "asx_net_forward: dim_c must be 4");
  const bool den = e->cfg.enable_denoise;
  e->cfg.enable_denoise = 0;  // size the workspace for exactly B
  int rc = ensure_workspace(e, B, true);
  e->cfg.enable_denoise = den;
  CHK(rc);
  const size_t ns = (size_t)B * dc * Fq * T;
  DevBuf din, dout;
  BufGuard g{{&din, &dout}};
  CHK(to_dev(din, spec_host, ns));
  CHK(dout.ensure(ns * 4));
  // reference layout [B,4,F,T] -> engine layout [B,4,T,F]
  CHK(transpose_launch(din.f(), e->spec_in.f(), B * dc, Fq, T, nullptr));
  CHK(net_forward_dev(e, e->spec_in.f(), e->spec_out.f(), B, nullptr));
  CHK(transpose_launch(e->spec_out.f(), dout.f(), B * dc, T, Fq, nullptr));
  CHK(to_host(out_host, dout, ns));
  return ASX_OK;
}

int asx_run_model(asx_engine *e, const float *wave_host, int32_t B, float *out_host, uint32_t flags) {
  REQUIRE(e && wave_host && out_host && B > 0, "asx_run_model: bad argument");
  HIPCHK(hipSetDevice(e->device));
  const bool match = (flags & ASX_FLAG_MATCH_MIX) != 0;
  if (!match && !e->net_ready) {
    set_err("asx_run_model: net weights not committed");
    return ASX_ERR_STATE;
  }
  const int T = e->cfg.segment_size;
  const int64_t C = (int64_t)e->cfg.hop_length * (T - 1);
  CHK(ensure_workspace(e, B, !match));
  DevBuf dw, dout;
  BufGuard g{{&dw, &dout}};
  CHK(to_dev(dw, wave_host, (size_t)B * 2 * C));
  CHK(dout.ensure((size_t)B * 2 * C * 4));
  const size_t spec_elems = (size_t)4 * T * e->cfg.dim_f;
  CHK(stft_launch(e, dw.f(), nullptr, -1, B, C, T, e->spec_in.f(), 1, 3, 1.0f, nullptr));
  const float *spec_final = e->spec_in.f();
  int combine = 0;
  if (!match) {
    int Bn = B;
    if (e->cfg.enable_denoise) {
      CHK(stft_launch(e, dw.f(), nullptr, -1, B, C, T, e->spec_in.f() + (size_t)B * spec_elems, 1, 3, -1.0f, nullptr));
      Bn = 2 * B;
      combine = B;
    }
    CHK(net_forward_dev(e, e->spec_in.f(), e->spec_out.f(), Bn, nullptr));
    spec_final = e->spec_out.f();
  }
  CHK(istft_ola_launch(e, spec_final, B, T, combine, nullptr, C, dout.f(), nullptr));
  CHK(to_host(out_host, dout, (size_t)B * 2 * C));
  return ASX_OK;
}

// ---- single-layer hooks --------------------------------------------------------
int asx_op_conv(asx_engine *e, const char *op, const float *x_host, int32_t B, int32_t cin, int32_t t, int32_t f,
                const float *w_host, const float *b_host, int32_t cout, const float *aux_host, int32_t relu,
                float *y_host) {
  REQUIRE(e && op && x_host && w_host && y_host && B > 0 && cin > 0 && cout > 0 && t > 0 && f > 0,
          "asx_op_conv: bad argument");
  HIPCHK(hipSetDevice(e->device));
  int kind;
  int to = t, fo = f;
  if (!strcmp(op, "conv3x3")) kind = CK_3X3;
  else if (!strcmp(op, "down")) {
    kind = CK_DOWN;
    to = t / 2;
    fo = f / 2;
  } else if (!strcmp(op, "conv1x1")) kind = CK_1X1;
  else if (!strcmp(op, "up")) {
    kind = CK_UP;
    to = 2 * t;
    fo = 2 * f;
    REQUIRE(aux_host, "asx_op_conv(up): skip tensor required");
  } else {
    set_err("asx_op_conv: unknown op '%s'", op);
    return ASX_ERR_INVALID;
  }
  ConvLayer L;
  DevBuf dx, dy, dskip;
  BufGuard g{{&dx, &dy, &dskip, &L.w, &L.b, &L.wu, &L.wu2, &L.wu3, &L.wus, &L.wu6, &L.wu6h, &L.w3h, &L.wd6, &L.wup6}};
  CHK(conv_setup(L, kind, cin, cout, relu ? 1 : 0));
  CHK(conv_pack(L, w_host, b_host, e->winograd));
  CHK(to_dev(dx, x_host, (size_t)B * cin * t * f));
  const size_t ny = (size_t)B * cout * to * fo;
  CHK(dy.ensure(ny * 4));
  HIPCHK(hipMemset(dy.p, 0xff, ny * 4));  // NaN canary: every output element must be written
  if (kind == CK_UP) CHK(to_dev(dskip, aux_host, ny));
  CHK(conv_launch(e, L, dx.f(), dskip.f(), dy.f(), B, t, f, nullptr));
  CHK(to_host(y_host, dy, ny));
  return ASX_OK;
}

int asx_op_tdf(asx_engine *e, const float *x_host, int32_t B, int32_t c, int32_t t, int32_t k, const float *w_host,
               const float *bias_host, int32_t n, const float *scale_host, const float *shift_host,
               const float *res_host, float *y_host) {
  REQUIRE(e && x_host && w_host && scale_host && shift_host && y_host && B > 0 && c > 0 && t > 0 && k > 0 && n > 0,
          "asx_op_tdf: bad argument");
  HIPCHK(hipSetDevice(e->device));
  TdfLayer L;
  DevBuf dx, dy, dres;
  BufGuard g{{&dx, &dy, &dres, &L.w, &L.bias, &L.scale, &L.shift}};
  CHK(tdf_pack(L, n, k, c, w_host, bias_host, scale_host, shift_host));
  const int64_t M = (int64_t)B * c * t;
  CHK(to_dev(dx, x_host, (size_t)M * k));
  CHK(dy.ensure((size_t)M * n * 4));
  HIPCHK(hipMemset(dy.p, 0xff, (size_t)M * n * 4));
  if (res_host) CHK(to_dev(dres, res_host, (size_t)M * n));
  const int rc = tdf_launch(e, L, dx.f(), res_host ? dres.f() : nullptr, dy.f(), M, t, nullptr);
  const int rc2 = rc == ASX_OK ? to_host(y_host, dy, (size_t)M * n) : rc;   // synchronises: the launch is done with the image
  w3_drop(e, L.w.p);                                   // the temporary layer's split image goes with its weight buffer
  return rc2;
}

// x + tdf1(tdf0(x)): the two linears of a TDF block with a bottleneck, launched the way the net launches them (tdf_pair_launch: the bottleneck
// activations as a pair image when option "gemm_pair_images" and the shapes allow).  h_host (optional): the bottleneck activations as fp32 --
// decoded from the pair image when one was written
int asx_op_tdf_block(asx_engine *e, const float *x_host, int32_t B, int32_t c, int32_t t, int32_t f, const float *w0_host, const float *scale0_host,
                     const float *shift0_host, int32_t n8, const float *w1_host, const float *scale1_host, const float *shift1_host, float *y_host,
                     float *h_host) {
  REQUIRE(e && x_host && w0_host && w1_host && scale0_host && shift0_host && scale1_host && shift1_host && y_host && B > 0 && c > 0 && t > 0 &&
              f > 0 && n8 > 0,
          "asx_op_tdf_block: bad argument");
  HIPCHK(hipSetDevice(e->device));
  TdfLayer L0, L1;
  DevBuf dx, dy, dh, dhe;
  BufGuard g{{&dx, &dy, &dh, &dhe, &L0.w, &L0.bias, &L0.scale, &L0.shift, &L1.w, &L1.bias, &L1.scale, &L1.shift}};
  CHK(tdf_pack(L0, n8, f, c, w0_host, nullptr, scale0_host, shift0_host));
  CHK(tdf_pack(L1, f, n8, c, w1_host, nullptr, scale1_host, shift1_host));
  const int64_t M = (int64_t)B * c * t;
  CHK(to_dev(dx, x_host, (size_t)M * f));
  CHK(dy.ensure((size_t)M * f * 4));
  CHK(dh.ensure((size_t)M * n8 * 4 + 256));
  CHK(dhe.ensure((size_t)M * ((n8 + 127) / 128) * 4 + 256));
  HIPCHK(hipMemset(dy.p, 0xff, (size_t)M * f * 4));
  HIPCHK(hipMemset(dh.p, 0xff, (size_t)M * n8 * 4));
  HIPCHK(hipMemset(dhe.p, 0xff, (size_t)M * ((n8 + 127) / 128) * 4));
  const long long ps0 = g_tdf3ps_launches.load();
  int rc = tdf_pair_launch(e, L0, L1, dx.f(), dh.f(), reinterpret_cast<int *>(dhe.p), dy.f(), M, t, nullptr);
  if (rc == ASX_OK) rc = to_host(y_host, dy, (size_t)M * f);   // synchronises: the launches are done with the images
  if (rc == ASX_OK && h_host) {
    rc = to_host(h_host, dh, (size_t)M * n8);
    if (rc == ASX_OK && g_tdf3ps_launches.load() != ps0) {
      // pair image -> fp32 on the host: groups of four columns as h0..h3 l0..l3, exponent per (row, column tile of the first linear)
      TdfDmaArgs d0;
      tdf_fill_args(e, L0, dx.f(), nullptr, dh.f(), M, t, 1, d0);
      const int cols = tdf3_tile_cols(e, d0), nsp = (n8 + cols - 1) / cols;
      std::vector<int> ex((size_t)M * nsp);
      HIPCHK(hipMemcpy(ex.data(), dhe.p, ex.size() * 4, hipMemcpyDeviceToHost));
      std::vector<uint16_t> raw(8);
      for (int64_t r = 0; r < M; ++r)
        for (int g4 = 0; g4 < n8 / 4; ++g4) {
          float *grp = h_host + r * n8 + g4 * 4;
          memcpy(raw.data(), grp, 16);
          for (int i = 0; i < 4; ++i) grp[i] = ldexpf(conv3h_f16_f(raw[i]) + conv3h_f16_f(raw[4 + i]), -ex[(size_t)r * nsp + (g4 * 4) / cols]);
        }
    }
  }
  w3_drop(e, L0.w.p);
  w3_drop(e, L1.w.p);
  return rc;
}

// ---- MDXC / TFC-TDF v3 --------------------------------------------------------------
int asx_v3_begin(asx_engine *e, const asx_v3_config *cfg) {
  w3_flush(e);
  REQUIRE(e && cfg, "asx_v3_begin: null argument");
  REQUIRE(cfg->num_channels == 2, "only stereo (num_channels = 2) is supported");
  REQUIRE(cfg->num_subbands >= 1 && e->cfg.dim_f % cfg->num_subbands == 0, "dim_f must be divisible by num_subbands");
  REQUIRE(cfg->num_scales >= 1 && cfg->num_blocks_per_scale >= 1 && cfg->num_channels_model > 0 && cfg->growth >= 0 &&
              cfg->bottleneck_factor >= 1 && cfg->num_targets >= 1,
          "bad TFC-TDF v3 hyper-parameters");
  REQUIRE(cfg->norm == 0 || cfg->norm == 1, "norm must be None (0) or InstanceNorm (1)");
  REQUIRE(cfg->act == 0 || cfg->act == 1, "act must be relu (0) or gelu (1)");
  const int fs = e->cfg.dim_f / cfg->num_subbands;
  REQUIRE(fs % (1 << cfg->num_scales) == 0 && e->cfg.segment_size % (1 << cfg->num_scales) == 0,
          "dim_f / num_subbands and segment_size must be divisible by 2^num_scales");
  REQUIRE((fs >> cfg->num_scales) % cfg->bottleneck_factor == 0, "bottleneck_factor must divide the deepest dim_f");
  if (!e->v3) e->v3 = new V3Net();
  v3_free(*e->v3);
  e->v3->cfg = *cfg;
  e->v3->begun = true;
  e->v3->ws_batch = 0;
  e->host_tensors.clear();
  e->net_begun = true;   // asx_net_set_tensor() is shared
  return ASX_OK;
}

int asx_v3_commit(asx_engine *e) {
  w3_flush(e);
  REQUIRE(e, "asx_v3_commit: null engine");
  if (!e->v3 || !e->v3->begun) {
    set_err("asx_v3_commit before asx_v3_begin");
    return ASX_ERR_STATE;
  }
  HIPCHK(hipSetDevice(e->device));
  V3Net &n = *e->v3;
  const asx_v3_config &cf = n.cfg;
  const int k = cf.num_subbands, dim_c = k * cf.num_channels * 2;
  int c = cf.num_channels_model, f = e->cfg.dim_f / k;
  CHK(v3_load_conv(e, n.first, CK_1X1, "first_conv.weight", dim_c, c, 1));
  n.enc.assign(cf.num_scales, {});
  n.dec.assign(cf.num_scales, {});
  n.ds.assign(cf.num_scales, ConvLayer());
  n.us.assign(cf.num_scales, ConvLayer());
  n.ds_n.assign(cf.num_scales, V3Norm());
  n.us_n.assign(cf.num_scales, V3Norm());
  for (int i = 0; i < cf.num_scales; ++i) {
    const std::string p = "encoder_blocks." + std::to_string(i);
    CHK(v3_load_tfc_tdf(e, n.enc[i], p + ".tfc_tdf", c, c, f));
    CHK(v3_load_norm(e, n.ds_n[i], p + ".downscale.conv.0", c));
    CHK(v3_load_conv(e, n.ds[i], CK_DOWN, p + ".downscale.conv.2.weight", c, c + cf.growth, 4));
    c += cf.growth;
    f /= 2;
  }
  CHK(v3_load_tfc_tdf(e, n.mid, "bottleneck_block", c, c, f));
  for (int i = 0; i < cf.num_scales; ++i) {
    const std::string p = "decoder_blocks." + std::to_string(i);
    CHK(v3_load_norm(e, n.us_n[i], p + ".upscale.conv.0", c));
    CHK(v3_load_conv(e, n.us[i], CK_UP, p + ".upscale.conv.2.weight", c, c - cf.growth, 4));
    c -= cf.growth;
    f *= 2;
    CHK(v3_load_tfc_tdf(e, n.dec[i], p + ".tfc_tdf", 2 * c, c, f));
  }
  CHK(v3_load_conv(e, n.final0, CK_1X1, "final_conv.0.weight", c + dim_c, c, 1));
  CHK(v3_load_conv(e, n.final1, CK_1X1, "final_conv.2.weight", c, cf.num_targets * dim_c, 1));
  e->host_tensors.clear();
  n.ready = true;
  return ASX_OK;
}

double asx_v3_flops(const asx_engine *e, int32_t batch) { return e ? v3_flops(e, batch) : 0.0; }

int asx_v3_forward(asx_engine *e, const float *wave_host, int32_t B, float *out_host) {
  REQUIRE(e && wave_host && out_host && B > 0, "asx_v3_forward: bad argument");
  if (!e->v3 || !e->v3->ready) {
    set_err("asx_v3_forward: weights not committed");
    return ASX_ERR_STATE;
  }
  HIPCHK(hipSetDevice(e->device));
  const int64_t C = (int64_t)e->cfg.hop_length * (e->cfg.segment_size - 1);
  const int S = e->v3->cfg.num_targets;
  DevBuf dw, dout;
  BufGuard g{{&dw, &dout}};
  CHK(to_dev(dw, wave_host, (size_t)B * 2 * C));
  CHK(dout.ensure((size_t)B * S * 2 * C * 4));
  CHK(v3_chunks_dev(e, dw.f(), nullptr, -1, 0, B, dout.f(), nullptr));
  CHK(to_host(out_host, dout, (size_t)B * S * 2 * C));
  return ASX_OK;
}

int asx_mdxc_plan(const asx_engine *e, int64_t N, int32_t overlap, asx_plan *out) {
  REQUIRE(e && out, "asx_mdxc_plan: null argument");
  REQUIRE(N >= 1 && overlap >= 1, "n_samples and overlap must be >= 1");
  asx_plan p{};
  p.n_samples = N;
  p.chunk_size = (int64_t)e->cfg.hop_length * (e->cfg.segment_size - 1);
  p.step = p.chunk_size / overlap;                       // hop_size (mdxc_separator.py:364)
  REQUIRE(p.step >= 1, "overlap larger than chunk_size");
  int64_t r = (N - p.chunk_size) % p.step;               // Python floor-mod (:368)
  if (r < 0) r += p.step;
  p.pad = p.step - r;
  p.trim = (int32_t)(p.chunk_size - p.step);             // zeros in front (:371)
  p.gen_size = p.step;
  p.padded_len = p.trim + N + p.pad + p.chunk_size - p.step;
  p.n_chunks = (int32_t)((p.padded_len - p.chunk_size) / p.step + 1);   // Tensor.unfold (:374)
  p.n_frames = e->cfg.segment_size;
  *out = p;
  return ASX_OK;
}

// chunks [k0, k1) of the unfold loop (mdxc_separator.py:374-392) -> chunk_out [k1-k0, S, 2, chunk]
int asx_mdxc_chunks_dev(asx_engine *e, const float *mix_dev, int64_t N, int32_t overlap, int32_t k0, int32_t k1, float *chunk_out_dev,
                        void *stream) {
  REQUIRE(e && mix_dev && chunk_out_dev, "asx_mdxc_chunks_dev: null argument");
  if (!e->v3 || !e->v3->ready) {
    set_err("asx_mdxc_demix: weights not committed");
    return ASX_ERR_STATE;
  }
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  HIPCHK(hipSetDevice(e->device));
  asx_plan p;
  CHK(asx_mdxc_plan(e, N, overlap, &p));
  REQUIRE(k0 >= 0 && k0 <= k1 && k1 <= p.n_chunks, "chunk range [%d, %d) outside [0, %d)", k0, k1, p.n_chunks);
  if (k1 == k0) return ASX_OK;
  V3Net &n = *e->v3;
  const int S = n.cfg.num_targets;
  const int64_t C = p.chunk_size;
  const int nk = k1 - k0;
  CHK(n.d_starts.ensure((size_t)nk * 16));   // starts | (unused) active lengths, built on the device: no host sync
  hipLaunchKernelGGL(chunk_table_kernel, dim3((unsigned)((nk + 255) / 256)), dim3(256), 0, s, k0, nk, p.step, p.chunk_size,
                     p.padded_len, 0, reinterpret_cast<int64_t *>(n.d_starts.p), reinterpret_cast<int64_t *>(n.d_starts.p) + nk);
  HIPCHK(hipGetLastError());
  const int maxB = e->cfg.max_batch > 0 ? e->cfg.max_batch : 8;
  const int nbatch = (nk + maxB - 1) / maxB;
  const int per = (nk + nbatch - 1) / nbatch;
  for (int j = 0; j < nk; j += per) {
    const int B = std::min(per, nk - j);
    CHK(v3_chunks_dev(e, mix_dev, reinterpret_cast<const int64_t *>(n.d_starts.p) + j, N, p.trim, B,
                      chunk_out_dev + (size_t)j * S * 2 * C, s));
  }
  return ASX_OK;
}

// uniform fold of ALL chunks, / overlap (mdxc_separator.py:246-255, 394-404): chunk_out [n_chunks, S, 2, chunk] -> out [S, 2, N]
int asx_mdxc_finalize_dev(asx_engine *e, const float *chunk_out_dev, int64_t N, int32_t overlap, float *out_dev, void *stream) {
  REQUIRE(e && chunk_out_dev && out_dev, "asx_mdxc_finalize_dev: null argument");
  if (!e->v3 || !e->v3->ready) {
    set_err("asx_mdxc_demix: weights not committed");
    return ASX_ERR_STATE;
  }
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  HIPCHK(hipSetDevice(e->device));
  asx_plan p;
  CHK(asx_mdxc_plan(e, N, overlap, &p));
  const int S = e->v3->cfg.num_targets;
  const int64_t C = p.chunk_size;
  const double bytes = 4.0 * ((double)p.n_chunks * S * 2 * C + (double)S * 2 * N);
  return timed(e, ASX_PROF_FINALIZE, 0.0, bytes, s, [&]() {
    hipLaunchKernelGGL(mdxc_finalize_kernel, dim3((unsigned)((N + 255) / 256), S * 2), dim3(256), 0, s, chunk_out_dev, p.n_chunks, S, C,
                       p.step, (int64_t)p.trim, N, (float)overlap, out_dev);
  });
}

int asx_mdxc_demix_dev(asx_engine *e, const float *mix_dev, int64_t N, int32_t overlap, float *out_dev, void *stream) {
  REQUIRE(e && mix_dev && out_dev, "asx_mdxc_demix_dev: null argument");
  if (!e->v3 || !e->v3->ready) {
    set_err("asx_mdxc_demix: weights not committed");
    return ASX_ERR_STATE;
  }
  asx_plan p;
  CHK(asx_mdxc_plan(e, N, overlap, &p));
  V3Net &n = *e->v3;
  CHK(n.chunk_out.ensure((size_t)p.n_chunks * n.cfg.num_targets * 2 * p.chunk_size * 4));
  CHK(asx_mdxc_chunks_dev(e, mix_dev, N, overlap, 0, p.n_chunks, n.chunk_out.f(), stream));
  return asx_mdxc_finalize_dev(e, n.chunk_out.f(), N, overlap, out_dev, stream);
}

int asx_mdxc_demix(asx_engine *e, const float *mix_host, int64_t N, int32_t overlap, float *out_host) {
  REQUIRE(e && mix_host && out_host, "asx_mdxc_demix: null argument");
  REQUIRE(N >= 1, "n_samples must be >= 1");
  if (!e->v3 || !e->v3->ready) {
    set_err("asx_mdxc_demix: weights not committed");
    return ASX_ERR_STATE;
  }
  HIPCHK(hipSetDevice(e->device));
  const int S = e->v3->cfg.num_targets;
  DevBuf dmix, dout;
  BufGuard g{{&dmix, &dout}};
  CHK(to_dev(dmix, mix_host, (size_t)2 * N));
  CHK(dout.ensure((size_t)S * 2 * N * 4));
  CHK(asx_mdxc_demix_dev(e, dmix.f(), N, overlap, dout.f(), nullptr));
  CHK(to_host(out_host, dout, (size_t)S * 2 * N));
  return ASX_OK;
}

// ---- BS-Roformer ------------------------------------------------------------------
int asx_rof_begin(asx_engine *e, const asx_rof_config *cfg) {
  w3_flush(e);
  REQUIRE(e && cfg, "asx_rof_begin: null argument");
  REQUIRE(cfg->dim_head == 64, "dim_head must be 64 (got %d)", cfg->dim_head);
  REQUIRE(cfg->dim > 0 && cfg->dim % 4 == 0 && cfg->depth >= 1 && cfg->heads >= 1 && cfg->num_stems >= 1 &&
              cfg->time_depth >= 1 && cfg->freq_depth >= 1 && cfg->mlp_expansion_factor >= 1 &&
              cfg->mask_estimator_depth >= 1 && cfg->n_out >= 1,
          "bad BS-Roformer hyper-parameters");
  REQUIRE(cfg->n_bands >= 2 && cfg->n_bands <= 128, "n_bands must be in [2, 128]");
  int sum = 0;
  for (int j = 0; j < cfg->n_bands; ++j) {
    REQUIRE(cfg->freqs_per_bands[j] >= 1, "freqs_per_bands must be positive");
    sum += cfg->freqs_per_bands[j];
  }
  REQUIRE(e->cfg.dim_f == e->cfg.n_fft / 2 + 1, "dim_f must be n_fft/2 + 1 = %d", e->cfg.n_fft / 2 + 1);
  if (cfg->mel) {
    for (int j = 0; j < cfg->n_bands; ++j) {
      REQUIRE(cfg->band_start[j] >= 0 && cfg->band_start[j] + cfg->freqs_per_bands[j] <= e->cfg.dim_f, "mel band %d out of range", j);
      REQUIRE(j == 0 || (cfg->band_start[j] >= cfg->band_start[j - 1] &&
                         cfg->band_start[j] + cfg->freqs_per_bands[j] >= cfg->band_start[j - 1] + cfg->freqs_per_bands[j - 1]),
              "mel bands must be ordered");
    }
  } else {
    REQUIRE(sum == e->cfg.dim_f, "sum(freqs_per_bands) = %d must equal dim_f = n_fft/2 + 1 = %d", sum, e->cfg.dim_f);
  }
  if (!e->rof) e->rof = new RofNet();
  rof_free(*e->rof);
  RofNet &n = *e->rof;
  n.cfg = *cfg;
  n.band_dim.clear();
  n.band_off.clear();
  n.mask_off.clear();
  int off = 0;
  for (int j = 0; j < cfg->n_bands; ++j) {
    n.band_dim.push_back(4 * cfg->freqs_per_bands[j]);   // 2 (complex) * 2 (stereo) * freqs
    n.band_off.push_back(cfg->mel ? 4 * cfg->band_start[j] : off);
    n.mask_off.push_back(off);
    off += 4 * cfg->freqs_per_bands[j];
  }
  n.W = 4 * e->cfg.dim_f;   // width of the per-frame spectrum vector (f s c)
  n.MW = off;               // width of the concatenated band masks (= W unless the bands overlap)
  n.begun = true;
  e->host_tensors.clear();
  e->net_begun = true;
  return ASX_OK;
}

int asx_rof_commit(asx_engine *e) {
  w3_flush(e);
  REQUIRE(e, "asx_rof_commit: null engine");
  if (!e->rof || !e->rof->begun) {
    set_err("asx_rof_commit before asx_rof_begin");
    return ASX_ERR_STATE;
  }
  HIPCHK(hipSetDevice(e->device));
  RofNet &n = *e->rof;
  const asx_rof_config &c = n.cfg;
  const int T = e->cfg.segment_size, Fb = c.n_bands, D = c.dim;
  n.bs_gamma.assign(Fb, DevBuf());
  n.bs_lin.assign(Fb, RofLin());
  for (int j = 0; j < Fb; ++j) {
    const std::string p = "band_split.to_features." + std::to_string(j);
    CHK(rof_upload(e, n.bs_gamma[j], p + ".0.gamma", n.band_dim[j]));
    CHK(rof_load_lin(e, n.bs_lin[j], p + ".1", D, n.band_dim[j], true));
  }
  n.time_l.assign(c.depth, {});
  n.freq_l.assign(c.depth, {});
  for (int i = 0; i < c.depth; ++i) {
    n.time_l[i].assign(c.time_depth, RofLayer());
    n.freq_l[i].assign(c.freq_depth, RofLayer());
    for (int j = 0; j < c.time_depth; ++j)
      CHK(rof_load_layer(e, n.time_l[i][j], "layers." + std::to_string(i) + ".0.layers." + std::to_string(j), T));
    for (int j = 0; j < c.freq_depth; ++j)
      CHK(rof_load_layer(e, n.freq_l[i][j], "layers." + std::to_string(i) + ".1.layers." + std::to_string(j), Fb));
  }
  if (c.mel) {
    n.tnorm_t.assign(c.depth, DevBuf());
    n.tnorm_f.assign(c.depth, DevBuf());
    for (int i = 0; i < c.depth; ++i) {
      CHK(rof_upload(e, n.tnorm_t[i], "layers." + std::to_string(i) + ".0.norm.gamma", D));
      CHK(rof_upload(e, n.tnorm_f[i], "layers." + std::to_string(i) + ".1.norm.gamma", D));
    }
    // bins -> covering band range (bands are ordered, so the covering set is one run of bands)
    const int F = e->cfg.dim_f;
    std::vector<int> jlo(F, Fb), jhi(F, -1), bst(Fb), mo(Fb);
    for (int j = 0; j < Fb; ++j) {
      bst[j] = c.band_start[j];
      mo[j] = n.mask_off[j];
      for (int f = c.band_start[j]; f < c.band_start[j] + c.freqs_per_bands[j]; ++f) {
        jlo[f] = std::min(jlo[f], j);
        jhi[f] = std::max(jhi[f], j);
      }
    }
    for (int f = 0; f < F; ++f) {
      REQUIRE(jhi[f] >= jlo[f], "all frequencies need to be covered by all bands for now (bin %d is not)", f);
      for (int j = jlo[f]; j <= jhi[f]; ++j)
        REQUIRE(f >= c.band_start[j] && f < c.band_start[j] + c.freqs_per_bands[j], "mel bands covering bin %d are not one run", f);
    }
    auto upi = [&](DevBuf &d, const std::vector<int> &v) -> int {
      CHK(d.ensure(v.size() * 4));
      HIPCHK(hipMemcpy(d.p, v.data(), v.size() * 4, hipMemcpyHostToDevice));
      return ASX_OK;
    };
    CHK(upi(n.d_bstart, bst));
    CHK(upi(n.d_moff, mo));
    CHK(upi(n.d_jlo, jlo));
    CHK(upi(n.d_jhi, jhi));
  } else {
    CHK(rof_upload(e, n.final_g, "final_norm.gamma", D));
  }
  const int hid = D * (c.mel ? 4 : c.mlp_expansion_factor);   // mel: MaskEstimator is built with the default factor 4
  const int n_lin = c.mask_estimator_depth + (c.mel ? 1 : 0);  // mel MLP: dims = (in, hidden * depth, out)
  n.mask.assign(c.num_stems, {});
  for (int st = 0; st < c.num_stems; ++st) {
    n.mask[st].assign(Fb, {});
    for (int j = 0; j < Fb; ++j) {
      auto &mlp = n.mask[st][j];
      mlp.assign(n_lin, RofLin());
      int in = D;
      for (int li = 0; li < n_lin; ++li) {
        const int out = (li + 1 == n_lin) ? 2 * n.band_dim[j] : hid;
        CHK(rof_load_lin(e, mlp[li],
                         "mask_estimators." + std::to_string(st) + ".to_freqs." + std::to_string(j) + ".0." +
                             std::to_string(2 * li),
                         out, in, true));
        in = out;
      }
    }
  }
  // Hamming fold window (scipy.signal.windows.hamming(chunk), float64 -> float32, mdxc_separator.py:310)
  const int64_t C = (int64_t)e->cfg.hop_length * (T - 1);
  std::vector<float> w((size_t)C);
  for (int64_t i = 0; i < C; ++i)
    w[i] = (float)(C == 1 ? 1.0 : 0.54 - 0.46 * cos(2.0 * M_PI * (double)i / (double)(C - 1)));
  CHK(n.d_window.ensure((size_t)C * 4));
  HIPCHK(hipMemcpy(n.d_window.p, w.data(), (size_t)C * 4, hipMemcpyHostToDevice));
  e->host_tensors.clear();
  n.ready = true;
  return ASX_OK;
}

double asx_rof_flops(const asx_engine *e, int32_t batch) { return e ? rof_flops(e, batch) : 0.0; }

int asx_rof_forward(asx_engine *e, const float *wave_host, int32_t B, float *out_host) {
  REQUIRE(e && wave_host && out_host && B > 0, "asx_rof_forward: bad argument");
  if (!e->rof || !e->rof->ready) {
    set_err("asx_rof_forward: weights not committed");
    return ASX_ERR_STATE;
  }
  HIPCHK(hipSetDevice(e->device));
  const int64_t C = (int64_t)e->cfg.hop_length * (e->cfg.segment_size - 1);
  const int S = e->rof->cfg.num_stems;
  DevBuf dw, dout;
  BufGuard g{{&dw, &dout}};
  CHK(to_dev(dw, wave_host, (size_t)B * 2 * C));
  CHK(dout.ensure((size_t)B * S * 2 * C * 4));
  CHK(rof_chunks_dev(e, dw.f(), nullptr, -1, B, dout.f(), nullptr));
  CHK(to_host(out_host, dout, (size_t)B * S * 2 * C));
  return ASX_OK;
}

// start of Roformer chunk k0 + t: t * step, the last one re-anchored to N - C (mdxc_separator.py:323-336)
__global__ void rof_starts_kernel(int k0, int nk, int64_t step, int64_t N, int64_t C, int64_t *__restrict__ starts) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= nk) return;
  const int64_t i = (int64_t)(k0 + t) * step;
  starts[t] = i + C > N ? N - C : i;
}

static int rof_starts(asx_engine *e, int64_t N, int64_t step, std::vector<int64_t> &starts) {
  const int64_t C = (int64_t)e->cfg.hop_length * (e->cfg.segment_size - 1);
  REQUIRE(N >= C, "mix (%lld samples) shorter than one chunk (%lld): not supported on the Roformer path", (long long)N, (long long)C);
  REQUIRE(step >= 1 && step <= C, "step must be in [1, chunk_size]");
  starts.clear();
  for (int64_t i = 0; i < N; i += step) starts.push_back(i + C > N ? N - C : i);   // tail re-anchored (:323-336)
  return ASX_OK;
}

int asx_rof_plan(const asx_engine *e, int64_t N, int64_t step, int32_t *n_chunks, int64_t *chunk_size) {
  REQUIRE(e && n_chunks && chunk_size, "asx_rof_plan: null argument");
  std::vector<int64_t> starts;
  CHK(rof_starts(const_cast<asx_engine *>(e), N, step, starts));
  *n_chunks = (int32_t)starts.size();
  *chunk_size = (int64_t)e->cfg.hop_length * (e->cfg.segment_size - 1);
  return ASX_OK;
}

// chunks [k0, k1) of the Roformer loop (mdxc_separator.py:318-336) -> chunk_out [k1-k0, S, 2, chunk]
int asx_rof_chunks_dev(asx_engine *e, const float *mix_dev, int64_t N, int64_t step, int32_t k0, int32_t k1, float *chunk_out_dev,
                       void *stream) {
  REQUIRE(e && mix_dev && chunk_out_dev, "asx_rof_chunks_dev: null argument");
  if (!e->rof || !e->rof->ready) {
    set_err("asx_rof_demix: weights not committed");
    return ASX_ERR_STATE;
  }
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  HIPCHK(hipSetDevice(e->device));
  RofNet &n = *e->rof;
  const int S = n.cfg.num_stems;
  const int64_t C = (int64_t)e->cfg.hop_length * (e->cfg.segment_size - 1);
  std::vector<int64_t> starts;
  CHK(rof_starts(e, N, step, starts));
  REQUIRE(k0 >= 0 && k0 <= k1 && k1 <= (int)starts.size(), "chunk range [%d, %d) outside [0, %d)", k0, k1, (int)starts.size());
  if (k1 == k0) return ASX_OK;
  const int nk = k1 - k0;
  CHK(n.d_starts.ensure((size_t)starts.size() * 8));
  // chunk starts built on the device (like chunk_table_kernel for MDX): the call only enqueues work
  hipLaunchKernelGGL(rof_starts_kernel, dim3((unsigned)((nk + 255) / 256)), dim3(256), 0, s, k0, nk, step, N, C,
                     reinterpret_cast<int64_t *>(n.d_starts.p));
  HIPCHK(hipGetLastError());
  const int maxB = e->cfg.max_batch > 0 ? e->cfg.max_batch : 8;
  const int nbatch = (nk + maxB - 1) / maxB;
  const int per = (nk + nbatch - 1) / nbatch;
  for (int j = 0; j < nk; j += per) {
    const int B = std::min(per, nk - j);
    CHK(rof_chunks_dev(e, mix_dev, reinterpret_cast<const int64_t *>(n.d_starts.p) + j, N, B, chunk_out_dev + (size_t)j * S * 2 * C, s));
  }
  return ASX_OK;
}

// Hamming-weighted fold of ALL chunks / counter.clamp(1e-10) (mdxc_separator.py:310-343)
int asx_rof_finalize_dev(asx_engine *e, const float *chunk_out_dev, int64_t N, int64_t step, float *out_dev, void *stream) {
  REQUIRE(e && chunk_out_dev && out_dev, "asx_rof_finalize_dev: null argument");
  if (!e->rof || !e->rof->ready) {
    set_err("asx_rof_demix: weights not committed");
    return ASX_ERR_STATE;
  }
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  HIPCHK(hipSetDevice(e->device));
  RofNet &n = *e->rof;
  const int S = n.cfg.num_stems;
  const int64_t C = (int64_t)e->cfg.hop_length * (e->cfg.segment_size - 1);
  std::vector<int64_t> starts;
  CHK(rof_starts(e, N, step, starts));
  const int nk = (int)starts.size();
  CHK(n.d_starts.ensure((size_t)nk * 8));
  hipLaunchKernelGGL(rof_starts_kernel, dim3((unsigned)((nk + 255) / 256)), dim3(256), 0, s, 0, nk, step, N, C,
                     reinterpret_cast<int64_t *>(n.d_starts.p));
  HIPCHK(hipGetLastError());
  const int n_out = n.cfg.n_out;
  return timed(e, ASX_PROF_FINALIZE, 0.0, 4.0 * ((double)nk * S * 2 * C + 2.0 * n_out * N), s, [&]() {
    hipLaunchKernelGGL(roformer_finalize_kernel, dim3((unsigned)((N + 255) / 256), n_out * 2), dim3(256), 0, s, chunk_out_dev,
                       reinterpret_cast<const int64_t *>(n.d_starts.p), nk, S, C, n.d_window.f(), N, out_dev);
  });
}

int asx_rof_demix_dev(asx_engine *e, const float *mix_dev, int64_t N, int64_t step, float *out_dev, void *stream) {
  REQUIRE(e && mix_dev && out_dev, "asx_rof_demix_dev: null argument");
  if (!e->rof || !e->rof->ready) {
    set_err("asx_rof_demix: weights not committed");
    return ASX_ERR_STATE;
  }
  RofNet &n = *e->rof;
  std::vector<int64_t> starts;
  CHK(rof_starts(e, N, step, starts));
  const int64_t C = (int64_t)e->cfg.hop_length * (e->cfg.segment_size - 1);
  CHK(n.chunk_out.ensure(starts.size() * n.cfg.num_stems * 2 * C * 4));
  CHK(asx_rof_chunks_dev(e, mix_dev, N, step, 0, (int32_t)starts.size(), n.chunk_out.f(), stream));
  return asx_rof_finalize_dev(e, n.chunk_out.f(), N, step, out_dev, stream);
}

int asx_rof_demix(asx_engine *e, const float *mix_host, int64_t N, int64_t step, float *out_host) {
  REQUIRE(e && mix_host && out_host, "asx_rof_demix: null argument");
  if (!e->rof || !e->rof->ready) {
    set_err("asx_rof_demix: weights not committed");
    return ASX_ERR_STATE;
  }
  HIPCHK(hipSetDevice(e->device));
  const int n_out = e->rof->cfg.n_out;
  DevBuf dmix, dout;
  BufGuard g{{&dmix, &dout}};
  CHK(to_dev(dmix, mix_host, (size_t)2 * N));
  CHK(dout.ensure((size_t)n_out * 2 * N * 4));
  CHK(asx_rof_demix_dev(e, dmix.f(), N, step, dout.f(), nullptr));
  CHK(to_host(out_host, dout, (size_t)n_out * 2 * N));
  return ASX_OK;
}

// ---- options -----------------------------------------------------------------------
// ---- Demucs v4 ---------------------------------------------------------------------
int asx_ht_begin(asx_engine *e, const asx_ht_config *cfg) {
  w3_flush(e);
  REQUIRE(e && cfg, "asx_ht_begin: null argument");
  REQUIRE(cfg->n_sources >= 1 && cfg->channels >= 4 && cfg->growth >= 1 && cfg->depth >= 1 && cfg->depth <= 8,
          "bad HTDemucs hyper-parameters");
  REQUIRE(cfg->kernel_size == 8 && cfg->stride == 4, "only kernel_size 8 / stride 4 is built (got %d / %d)",
          cfg->kernel_size, cfg->stride);
  REQUIRE(cfg->dconv_depth >= 0 && cfg->dconv_depth <= 4 && cfg->dconv_comp >= 1 && cfg->channels % cfg->dconv_comp == 0,
          "bad DConv hyper-parameters");
  REQUIRE(cfg->nfft >= 64 && cfg->nfft % 8 == 0, "bad nfft %d", cfg->nfft);
  REQUIRE(cfg->t_layers >= 0 && (cfg->t_layers == 0 || (cfg->t_heads >= 1 && cfg->t_hidden >= 4 && cfg->t_hidden % 4 == 0)),
          "bad transformer hyper-parameters");
  REQUIRE(cfg->samplerate > 0 && cfg->segment_samples > 0, "bad samplerate / segment");
  hd_drop_clones(e);
  if (e->hd) hd_free(*e->hd);   // a Demucs v3 net shares e->ht: it is gone with these weights
  if (!e->ht) e->ht = new HtNet();
  ht_free(*e->ht);
  e->ht->cfg = *cfg;
  e->ht->begun = true;
  e->host_tensors.clear();
  e->net_begun = true;
  return ASX_OK;
}

int asx_ht_commit(asx_engine *e) {
  w3_flush(e);
  REQUIRE(e, "asx_ht_commit: null engine");
  if (!e->ht || !e->ht->begun) {
    set_err("asx_ht_commit before asx_ht_begin");
    return ASX_ERR_STATE;
  }
  HIPCHK(hipSetDevice(e->device));
  const int rc = ht_commit(e);
  e->host_tensors.clear();
  return rc;
}

double asx_ht_flops(const asx_engine *e) { return (e && e->ht && e->ht->ready) ? ht_flops(e) : 0.0; }

int asx_ht_forward(asx_engine *e, const float *mix_host, int32_t B, int64_t length, float *out_host) {
  REQUIRE(e && mix_host && out_host && B > 0, "asx_ht_forward: bad argument");
  if (!e->ht || !e->ht->ready) {
    set_err("asx_ht_forward: weights not committed");
    return ASX_ERR_STATE;
  }
  HIPCHK(hipSetDevice(e->device));
  const int64_t TL = e->ht->L[0];
  const int S = e->ht->cfg.n_sources;
  REQUIRE(length >= 1 && length <= TL, "length %lld outside [1, %lld]", (long long)length, (long long)TL);
  DevBuf din, dout;
  BufGuard g{{&din, &dout}};
  CHK(din.ensure((size_t)B * 2 * TL * 4));
  CHK(dout.ensure((size_t)B * S * 2 * TL * 4));
  HIPCHK(hipMemset(din.p, 0, (size_t)B * 2 * TL * 4));
  HIPCHK(hipMemcpy2D(din.p, (size_t)TL * 4, mix_host, (size_t)length * 4, (size_t)length * 4, (size_t)B * 2,
                     hipMemcpyHostToDevice));
  CHK(ht_forward_dev(e, din.f(), B, dout.f(), nullptr));
  HIPCHK(hipDeviceSynchronize());
  HIPCHK(hipMemcpy2D(out_host, (size_t)length * 4, dout.p, (size_t)TL * 4, (size_t)length * 4, (size_t)B * S * 2,
                     hipMemcpyDeviceToHost));
  return ASX_OK;
}

int asx_ht_demix_dev(asx_engine *e, const float *mix_dev, int64_t N, int32_t shifts, const int64_t *offsets, double overlap,
                     uint32_t flags, float *out_dev, void *stream) {
  REQUIRE(e && mix_dev && out_dev && N >= 2, "asx_ht_demix_dev: bad argument");
  REQUIRE(shifts >= 0 && (shifts == 0 || offsets), "shifts > 0 needs the offsets array");
  REQUIRE(overlap >= 0.0 && overlap < 1.0, "overlap must be in [0, 1)");
  if (!e->ht || !e->ht->ready) {
    set_err("asx_ht_demix: weights not committed");
    return ASX_ERR_STATE;
  }
  HIPCHK(hipSetDevice(e->device));
  return ht_demix_dev(e, mix_dev, N, shifts, offsets, overlap, flags, out_dev, reinterpret_cast<hipStream_t>(stream));
}

#define HT_READY(fn)                                   \
  do {                                                 \
    if (!e->ht || !e->ht->ready) {                     \
      set_err(fn ": weights not committed");           \
      return ASX_ERR_STATE;                            \
    }                                                  \
  } while (0)

int asx_ht_plan(const asx_engine *e, int64_t N, int32_t shifts, const int64_t *offsets, double overlap, int32_t *n_segments,
                int64_t *segment_samples) {
  REQUIRE(e && n_segments && segment_samples && N >= 2 && shifts >= 0 && (shifts == 0 || offsets), "asx_ht_plan: bad argument");
  HT_READY("asx_ht_plan");
  HtPlan p;
  CHK(ht_plan(e, N, shifts, offsets, overlap, p));
  *n_segments = (int32_t)p.starts.size();
  *segment_samples = p.segment;
  return ASX_OK;
}

int asx_ht_segments_dev(asx_engine *e, const float *mix_dev, int64_t N, int32_t shifts, const int64_t *offsets, double overlap,
                        uint32_t flags, int32_t k0, int32_t k1, float *chunk_out_dev, void *stream) {
  REQUIRE(e && mix_dev && chunk_out_dev && N >= 2 && shifts >= 0 && (shifts == 0 || offsets), "asx_ht_segments_dev: bad argument");
  HT_READY("asx_ht_segments_dev");
  HIPCHK(hipSetDevice(e->device));
  HtPlan p;
  CHK(ht_plan(e, N, shifts, offsets, overlap, p));
  REQUIRE(k0 >= 0 && k0 <= k1 && k1 <= (int)p.starts.size(), "segment range [%d, %d) outside [0, %d)", k0, k1, (int)p.starts.size());
  return ht_segments_dev(e, mix_dev, N, p, flags, k0, k1, chunk_out_dev, reinterpret_cast<hipStream_t>(stream));
}

int asx_ht_fold_dev(asx_engine *e, const float *mix_dev, int64_t N, int32_t shifts, const int64_t *offsets, double overlap, uint32_t flags,
                    const float *chunk_out_dev, float *out_dev, void *stream) {
  REQUIRE(e && mix_dev && chunk_out_dev && out_dev && N >= 2 && shifts >= 0 && (shifts == 0 || offsets), "asx_ht_fold_dev: bad argument");
  HT_READY("asx_ht_fold_dev");
  HIPCHK(hipSetDevice(e->device));
  HtPlan p;
  CHK(ht_plan(e, N, shifts, offsets, overlap, p));
  return ht_fold_dev(e, mix_dev, N, p, flags, chunk_out_dev, out_dev, reinterpret_cast<hipStream_t>(stream));
}

int asx_ht_demix(asx_engine *e, const float *mix_host, int64_t N, int32_t shifts, const int64_t *offsets, double overlap,
                 uint32_t flags, float *out_host) {
  REQUIRE(e && mix_host && out_host && N >= 2, "asx_ht_demix: bad argument");
  if (!e->ht || !e->ht->ready) {
    set_err("asx_ht_demix: weights not committed");
    return ASX_ERR_STATE;
  }
  HIPCHK(hipSetDevice(e->device));
  const int S = e->ht->cfg.n_sources;
  DevBuf dm, dout;
  BufGuard g{{&dm, &dout}};
  CHK(to_dev(dm, mix_host, (size_t)2 * N));
  CHK(dout.ensure((size_t)S * 2 * N * 4));
  CHK(asx_ht_demix_dev(e, dm.f(), N, shifts, offsets, overlap, flags, dout.f(), nullptr));
  CHK(to_host(out_host, dout, (size_t)S * 2 * N));
  return ASX_OK;
}

// ---- Demucs v3 (HDemucs) -------------------------------------------------------------
int asx_hd_begin(asx_engine *e, const asx_hd_config *cfg) {
  w3_flush(e);
  REQUIRE(e && cfg, "asx_hd_begin: null argument");
  REQUIRE(cfg->n_sources >= 1 && cfg->channels >= 4 && cfg->growth >= 1 && cfg->depth >= 3 && cfg->depth <= 8, "bad HDemucs hyper-parameters");
  REQUIRE(cfg->kernel_size == 8 && cfg->stride == 4 && cfg->time_stride == 2,
          "only kernel_size 8 / stride 4 / time_stride 2 is built (got %d / %d / %d)", cfg->kernel_size, cfg->stride, cfg->time_stride);
  REQUIRE(cfg->dconv_depth >= 1 && cfg->dconv_depth <= 4 && cfg->dconv_comp >= 1 && cfg->channels % cfg->dconv_comp == 0, "bad DConv hyper-parameters");
  REQUIRE(cfg->norm_starts == cfg->depth - 2 && cfg->dconv_lstm == cfg->depth - 2 && cfg->dconv_attn == cfg->depth - 2,
          "only the default structure (GroupNorm, BLSTM and LocalState on the two innermost levels: norm_starts = dconv_lstm = dconv_attn = depth - 2) "
          "is built");
  REQUIRE(cfg->norm_groups >= 1, "bad norm_groups");
  REQUIRE(cfg->nfft >= 64 && cfg->nfft % 8 == 0, "bad nfft %d", cfg->nfft);
  REQUIRE(cfg->samplerate > 0 && cfg->segment_samples >= 1, "bad samplerate / segment");
  hd_drop_clones(e);
  if (!e->ht) e->ht = new HtNet();
  ht_free(*e->ht);
  if (!e->hd) e->hd = new HdNet();
  hd_free(*e->hd);
  e->hd->cfg = *cfg;
  e->hd->begun = true;
  asx_ht_config &h = e->ht->cfg;
  h = asx_ht_config{};
  h.n_sources = cfg->n_sources;
  h.channels = cfg->channels;
  h.growth = cfg->growth;
  h.nfft = cfg->nfft;
  h.depth = cfg->depth - 2;
  h.kernel_size = cfg->kernel_size;
  h.stride = cfg->stride;
  h.dconv_depth = cfg->dconv_depth;
  h.dconv_comp = cfg->dconv_comp;
  h.samplerate = cfg->samplerate;
  h.segment_samples = cfg->segment_samples;
  h.freq_emb_scale = cfg->freq_emb_scale;
  h.max_batch = cfg->max_batch;
  e->host_tensors.clear();
  e->net_begun = true;
  return ASX_OK;
}

int asx_hd_commit(asx_engine *e) {
  w3_flush(e);
  REQUIRE(e, "asx_hd_commit: null engine");
  if (!e->hd || !e->hd->begun) {
    set_err("asx_hd_commit before asx_hd_begin");
    return ASX_ERR_STATE;
  }
  HIPCHK(hipSetDevice(e->device));
  const int rc = hd_commit(e);
  e->host_tensors.clear();
  return rc;
}

#define HD_READY(fn)                                   \
  do {                                                 \
    if (!e->hd || !e->hd->ready) {                     \
      set_err(fn ": weights not committed");           \
      return ASX_ERR_STATE;                            \
    }                                                  \
  } while (0)

double asx_hd_flops(const asx_engine *e, int64_t length) { return (e && e->hd && e->hd->ready && length > 0) ? hd_flops(e, length) : 0.0; }

int asx_hd_forward(asx_engine *e, const float *mix_host, int32_t B, int64_t length, float *out_host) {
  REQUIRE(e && mix_host && out_host && B > 0 && length > 0, "asx_hd_forward: bad argument");
  HD_READY("asx_hd_forward");
  HIPCHK(hipSetDevice(e->device));
  const int S = e->hd->cfg.n_sources;
  DevBuf din, dout;
  BufGuard g{{&din, &dout}};
  CHK(to_dev(din, mix_host, (size_t)B * 2 * length));
  CHK(dout.ensure((size_t)B * S * 2 * length * 4));
  CHK(hd_forward_dev(e, din.f(), B, length, dout.f(), nullptr));
  HIPCHK(hipDeviceSynchronize());
  return to_host(out_host, dout, (size_t)B * S * 2 * length);
}

int asx_hd_demix_dev(asx_engine *e, const float *mix_dev, int64_t N, int32_t shifts, const int64_t *offsets, double overlap, uint32_t flags,
                     float *out_dev, void *stream) {
  REQUIRE(e && mix_dev && out_dev && N >= 2, "asx_hd_demix_dev: bad argument");
  REQUIRE(shifts >= 0 && (shifts == 0 || offsets), "shifts > 0 needs the offsets array");
  REQUIRE(overlap >= 0.0 && overlap < 1.0, "overlap must be in [0, 1)");
  HD_READY("asx_hd_demix_dev");
  HIPCHK(hipSetDevice(e->device));
  return hd_demix_dev(e, mix_dev, N, shifts, offsets, overlap, flags, out_dev, reinterpret_cast<hipStream_t>(stream));
}

int asx_hd_demix(asx_engine *e, const float *mix_host, int64_t N, int32_t shifts, const int64_t *offsets, double overlap, uint32_t flags,
                 float *out_host) {
  REQUIRE(e && mix_host && out_host && N >= 2, "asx_hd_demix: bad argument");
  HD_READY("asx_hd_demix");
  HIPCHK(hipSetDevice(e->device));
  const int S = e->hd->cfg.n_sources;
  DevBuf dm, dout;
  BufGuard g{{&dm, &dout}};
  CHK(to_dev(dm, mix_host, (size_t)2 * N));
  CHK(dout.ensure((size_t)S * 2 * N * 4));
  CHK(asx_hd_demix_dev(e, dm.f(), N, shifts, offsets, overlap, flags, dout.f(), nullptr));
  return to_host(out_host, dout, (size_t)S * 2 * N);
}

int asx_hd_plan(const asx_engine *e, int64_t N, int32_t shifts, const int64_t *offsets, double overlap, int32_t *n_segments,
                int64_t *segment_samples) {
  REQUIRE(e && n_segments && segment_samples && N >= 2 && shifts >= 0 && (shifts == 0 || offsets), "asx_hd_plan: bad argument");
  HD_READY("asx_hd_plan");
  HdPlan p;
  CHK(hd_plan(e, N, shifts, offsets, overlap, p));
  *n_segments = (int32_t)p.starts.size();
  *segment_samples = p.segment;
  return ASX_OK;
}

int asx_hd_segments_dev(asx_engine *e, const float *mix_dev, int64_t N, int32_t shifts, const int64_t *offsets, double overlap, uint32_t flags,
                        int32_t k0, int32_t k1, float *chunk_out_dev, void *stream) {
  REQUIRE(e && mix_dev && chunk_out_dev && N >= 2 && shifts >= 0 && (shifts == 0 || offsets), "asx_hd_segments_dev: bad argument");
  HD_READY("asx_hd_segments_dev");
  HIPCHK(hipSetDevice(e->device));
  HdPlan p;
  CHK(hd_plan(e, N, shifts, offsets, overlap, p));
  REQUIRE(k0 >= 0 && k0 <= k1 && k1 <= (int)p.starts.size(), "segment range [%d, %d) outside [0, %d)", k0, k1, (int)p.starts.size());
  return hd_segments_dev(e, mix_dev, N, p, flags, k0, k1, chunk_out_dev, reinterpret_cast<hipStream_t>(stream));
}

int asx_hd_fold_dev(asx_engine *e, const float *mix_dev, int64_t N, int32_t shifts, const int64_t *offsets, double overlap, uint32_t flags,
                    const float *chunk_out_dev, float *out_dev, void *stream) {
  REQUIRE(e && mix_dev && chunk_out_dev && out_dev && N >= 2 && shifts >= 0 && (shifts == 0 || offsets), "asx_hd_fold_dev: bad argument");
  HD_READY("asx_hd_fold_dev");
  HIPCHK(hipSetDevice(e->device));
  HdPlan p;
  CHK(hd_plan(e, N, shifts, offsets, overlap, p));
  return hd_fold_dev(e, mix_dev, N, p, flags, chunk_out_dev, out_dev, reinterpret_cast<hipStream_t>(stream));
}

// ---- VR ------------------------------------------------------------------------------
int asx_vr_begin(asx_engine *e, const asx_vr_config *cfg) {
  w3_flush(e);
  REQUIRE(e && cfg, "asx_vr_begin: null argument");
  REQUIRE(cfg->n_bands >= 1 && cfg->n_bands <= 8 && cfg->bins >= 32, "bad band layout");
  REQUIRE(cfg->channel_mode >= 0 && cfg->channel_mode <= 3, "bad channel_mode");
  for (int i = 0; i < (cfg->v51 ? 2 : 5); ++i) REQUIRE(cfg->cap[i] >= 4 && cfg->cap[i] % 4 == 0, "net widths must be multiples of 4");
  REQUIRE(cfg->window_size >= 16 && cfg->offset >= 0, "bad window_size / offset");
  for (int d = 0; d < cfg->n_bands; ++d)
    REQUIRE(cfg->band[d].sr > 0 && cfg->band[d].hl > 0 && cfg->band[d].n_fft >= 8 && cfg->band[d].n_fft % 2 == 0, "band %d: bad sr / hl / n_fft",
            d + 1);
  if (!e->vr) e->vr = new VrNet();
  vr_free(*e->vr);
  e->vr->cfg = *cfg;
  e->vr->begun = true;
  e->host_tensors.clear();
  e->net_begun = true;
  return ASX_OK;
}

int asx_vr_commit(asx_engine *e) {
  w3_flush(e);
  REQUIRE(e, "asx_vr_commit: null engine");
  if (!e->vr || !e->vr->begun) {
    set_err("asx_vr_commit before asx_vr_begin");
    return ASX_ERR_STATE;
  }
  HIPCHK(hipSetDevice(e->device));
  const int rc = vr_commit(e);
  e->host_tensors.clear();
  return rc;
}

double asx_vr_flops(const asx_engine *e) { return (e && e->vr && e->vr->ready) ? vr_flops_patch(e) : 0.0; }

#define VR_READY(fn)                                   \
  do {                                                 \
    if (!e->vr || !e->vr->ready) {                     \
      set_err(fn ": weights not committed");           \
      return ASX_ERR_STATE;                            \
    }                                                  \
  } while (0)

int asx_vr_plan(const asx_engine *e, int64_t n_samples, int32_t *n_frames, int64_t *n_out) {
  REQUIRE(e && n_frames && n_out && n_samples > 0, "asx_vr_plan: bad argument");
  VR_READY("asx_vr_plan");
  int T;
  CHK(vr_plan(*e->vr, n_samples, &T, n_out));
  *n_frames = T;
  return ASX_OK;
}

int asx_vr_forward(asx_engine *e, const float *x_host, int32_t B, float *out_host) {
  REQUIRE(e && x_host && out_host && B > 0, "asx_vr_forward: bad argument");
  VR_READY("asx_vr_forward");
  HIPCHK(hipSetDevice(e->device));
  VrNet &n = *e->vr;
  const int W = n.cfg.window_size;
  const size_t numel = (size_t)B * 2 * n.nb1 * W;
  DevBuf dx, dy;
  BufGuard g{{&dx, &dy}};
  CHK(to_dev(dx, x_host, numel));
  CHK(dy.ensure(numel * 4));
  CHK(n.cfg.v51 ? vr51_ensure_workspace(e, B) : vr_ensure_workspace(e, B));
  const int64_t P = (int64_t)B * n.max_bin * W;
  hipLaunchKernelGGL(vr_from_nchw_kernel, dim3((unsigned)((P + 255) / 256)), dim3(256), 0, nullptr, dx.f(), n.nb1, n.max_bin, W, n.ctot,
                     n.b.hc, P);
  HIPCHK(hipGetLastError());
  CHK(n.cfg.v51 ? vr51_net_dev(e, B, nullptr) : vr_net_dev(e, B, nullptr));
  hipLaunchKernelGGL(vr_to_nchw_kernel, dim3((unsigned)((numel + 255) / 256)), dim3(256), 0, nullptr, n.b.mk, n.nb1, n.max_bin, W, dy.f(),
                     (int64_t)numel);
  HIPCHK(hipGetLastError());
  CHK(to_host(out_host, dy, numel));
  return ASX_OK;
}

int asx_vr_analysis(asx_engine *e, const float *wave_host, int64_t n_samples, float *spec_host) {
  REQUIRE(e && wave_host && spec_host && n_samples > 0, "asx_vr_analysis: bad argument");
  VR_READY("asx_vr_analysis");
  HIPCHK(hipSetDevice(e->device));
  VrNet &n = *e->vr;
  int T;
  int64_t n_out;
  CHK(vr_plan(n, n_samples, &T, &n_out));
  DevBuf dw;
  BufGuard g{{&dw}};
  CHK(to_dev(dw, wave_host, (size_t)2 * n_samples));
  n.he_n = 0;
  CHK(vr_analysis_dev(e, dw.f(), n_samples, T, nullptr));
  CHK(to_host(spec_host, n.X, (size_t)2 * T * n.nb1 * 2));
  return ASX_OK;
}

int asx_vr_separate_dev(asx_engine *e, const float *wave_dev, int64_t n_samples, const asx_vr_params *params, float *primary_dev,
                        float *secondary_dev, void *stream) {
  REQUIRE(e && wave_dev && params && n_samples > 0, "asx_vr_separate_dev: bad argument");
  VR_READY("asx_vr_separate");
  HIPCHK(hipSetDevice(e->device));
  return vr_separate_dev(e, wave_dev, n_samples, params, primary_dev, secondary_dev, reinterpret_cast<hipStream_t>(stream));
}

int asx_vr_separate(asx_engine *e, const float *wave_host, int64_t n_samples, const asx_vr_params *params, float *primary_host,
                    float *secondary_host) {
  REQUIRE(e && wave_host && params && n_samples > 0, "asx_vr_separate: bad argument");
  VR_READY("asx_vr_separate");
  HIPCHK(hipSetDevice(e->device));
  int T;
  int64_t n_out;
  CHK(vr_plan(*e->vr, n_samples, &T, &n_out));
  DevBuf dw, dp, ds;
  BufGuard g{{&dw, &dp, &ds}};
  CHK(to_dev(dw, wave_host, (size_t)2 * n_samples));
  if (primary_host) CHK(dp.ensure((size_t)2 * n_out * 4));
  if (secondary_host) CHK(ds.ensure((size_t)2 * n_out * 4));
  CHK(vr_separate_dev(e, dw.f(), n_samples, params, primary_host ? dp.f() : nullptr, secondary_host ? ds.f() : nullptr, nullptr));
  if (primary_host) CHK(to_host(primary_host, dp, (size_t)2 * n_out));
  if (secondary_host) CHK(to_host(secondary_host, ds, (size_t)2 * n_out));
  return ASX_OK;
}

// launch counters of the bf16 x 6 kernels since the process started (tests assert that the path under test is the one that ran)
int asx_counter(const asx_engine *e, const char *name, int64_t *out) {
  REQUIRE(e && name && out, "asx_counter: null argument");
  const std::string nm(name);
  if (nm == "tdf3_launches") *out = (int64_t)g_tdf3_launches.load();
  else if (nm == "tdf3h_launches") *out = (int64_t)g_tdf3h_launches.load();
  else if (nm == "attn6_launches") *out = (int64_t)g_attn6_launches.load();
  else if (nm == "attn6h_launches") *out = (int64_t)g_attn6h_launches.load();
  else if (nm == "tdf3_gather_launches") *out = (int64_t)g_tdf3_gather_launches.load();
  else if (nm == "wino6_launches") *out = (int64_t)g_wino6_launches.load();
  else if (nm == "wino6h_launches") *out = (int64_t)g_wino6h_launches.load();
  else if (nm == "conv3h_launches") *out = (int64_t)g_conv3h_launches.load();
  else if (nm == "tdf3_pair_image_launches") *out = (int64_t)g_tdf3ps_launches.load();
  else if (nm == "down6_launches") *out = (int64_t)g_down6_launches.load();
  else if (nm == "up6_launches") *out = (int64_t)g_up6_launches.load();
  else {
    set_err("asx_counter: unknown counter '%s'", name);
    return ASX_ERR_INVALID;
  }
  return ASX_OK;
}

// debug hook: copy `numel` floats of a named engine workspace buffer to the host (tests / bring-up only)
int asx_debug_trace(uint64_t *host, int64_t n_u64) {
  REQUIRE(host && n_u64 > 0 && n_u64 <= 4096 * 8, "asx_debug_trace: bad argument");
  HIPCHK(hipDeviceSynchronize());
  HIPCHK(hipMemcpyFromSymbol(host, HIP_SYMBOL(asx_dbg_trace), (size_t)n_u64 * 8));
  return ASX_OK;
}

int asx_debug_fetch(asx_engine *e, const char *name, float *host, int64_t numel) {
  REQUIRE(e && name && host && numel > 0, "asx_debug_fetch: bad argument");
  HIPCHK(hipSetDevice(e->device));
  const float *src = nullptr;
  const std::string nm(name);
  if (e->vr && e->vr->ws_batch > 0) {
    auto &b = e->vr->b;
    if (nm == "vr.hc") src = b.hc;
    else if (nm == "vr.y2") src = b.y2;
    else if (nm == "vr.y3") src = b.y3;
    else if (nm == "vr.h3") src = b.h3;
    else if (nm == "vr.mk") src = b.mk;
    else if (nm == "vr.cat") src = b.cat;
    else if (nm == "vr.bn") src = b.bn;
    else if (nm == "vr.pool") src = b.pool;
    else if (nm == "vr.pool2") src = b.pool2;
    else if (nm == "vr.tmp") src = b.tmp;
    else if (nm.size() == 5 && nm.compare(0, 3, "vr.") == 0 && nm[4] >= '0' && nm[4] < '0' + (int)b.D.size()) {
      const int i = nm[4] - '0';
      src = nm[3] == 'D' ? b.D[i] : (nm[3] == 'E' ? b.E[i] : (nm[3] == 'O' ? b.O[i] : nullptr));
    }
  }
  if (e->hd && e->hd->ws_batch > 0 && e->ht && nm.compare(0, 3, "hd.") == 0) {
    auto &w = e->hd->b;
    auto &b = e->ht->b;
    const int D = e->hd->D;
    const int i = nm.size() == 7 ? nm[6] - '0' : -1;
    if (nm == "hd.inj") src = w.inj;
    else if (nm == "hd.ya") src = w.ya;
    else if (nm == "hd.yz") src = w.yz;
    else if (nm == "hd.skA") src = w.skA;
    else if (nm == "hd.skZ") src = w.skZ;
    else if (nm == "hd.dAin") src = w.dAin;
    else if (nm == "hd.pre") src = w.pre;
    else if (nm.compare(0, 6, "hd.skf") == 0 && i >= 0 && i < D) src = b.skf[i];
    else if (nm.compare(0, 6, "hd.skt") == 0 && i >= 0 && i < D) src = b.skt[i];
    else if (nm.compare(0, 6, "hd.df_") == 0 && i >= 0 && i <= D) src = b.df[i];
    else if (nm.compare(0, 6, "hd.dt_") == 0 && i >= 0 && i <= D) src = b.dt[i];
  }
  if (!src) {
    set_err("asx_debug_fetch: unknown buffer '%s'", name);
    return ASX_ERR_INVALID;
  }
  HIPCHK(hipDeviceSynchronize());
  HIPCHK(hipMemcpy(host, src, (size_t)numel * 4, hipMemcpyDeviceToHost));
  return ASX_OK;
}

// ---- spectral edges ------------------------------------------------------------------
int asx_ensemble(asx_engine *e, const float *waves_host, int32_t K, int64_t N, int32_t algorithm, const double *weights,
                 float *out_host, int64_t *n_out) {
  REQUIRE(e && waves_host && out_host && n_out && N >= 1, "asx_ensemble: bad argument");
  HIPCHK(hipSetDevice(e->device));
  CHK(ens_ctx(e));
  EnsCtx &c = *e->ens;
  CHK(c.din.ensure((size_t)K * 2 * N * 4));
  CHK(c.dout.ensure((size_t)2 * N * 4));
  HIPCHK(hipMemcpy(c.din.p, waves_host, (size_t)K * 2 * N * 4, hipMemcpyHostToDevice));
  CHK(ens_ensemble_dev(e, c.din.f(), K, N, algorithm, weights, c.dout.f(), n_out, nullptr));
  HIPCHK(hipDeviceSynchronize());
  HIPCHK(hipMemcpy(out_host, c.dout.p, (size_t)2 * (*n_out) * 4, hipMemcpyDeviceToHost));
  return ASX_OK;
}

int asx_ensemble_dev(asx_engine *e, const float *waves_dev, int32_t K, int64_t N, int32_t algorithm, const double *weights,
                     float *out_dev, int64_t *n_out, void *stream) {
  REQUIRE(e && waves_dev && out_dev && n_out && N >= 1, "asx_ensemble_dev: bad argument");
  HIPCHK(hipSetDevice(e->device));
  return ens_ensemble_dev(e, waves_dev, K, N, algorithm, weights, out_dev, n_out, reinterpret_cast<hipStream_t>(stream));
}

int asx_invert_stem(asx_engine *e, const float *mix_host, const float *stem_host, int64_t N, float *out_host, int64_t *n_out) {
  REQUIRE(e && mix_host && stem_host && out_host && n_out && N >= 1, "asx_invert_stem: bad argument");
  HIPCHK(hipSetDevice(e->device));
  CHK(ens_ctx(e));
  EnsCtx &c = *e->ens;
  CHK(c.din.ensure((size_t)2 * N * 4));
  CHK(c.din2.ensure((size_t)2 * N * 4));
  CHK(c.dout.ensure((size_t)2 * N * 4));
  HIPCHK(hipMemcpy(c.din.p, mix_host, (size_t)2 * N * 4, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(c.din2.p, stem_host, (size_t)2 * N * 4, hipMemcpyHostToDevice));
  CHK(ens_invert_dev(e, c.din.f(), c.din2.f(), N, c.dout.f(), n_out, nullptr));
  HIPCHK(hipDeviceSynchronize());
  HIPCHK(hipMemcpy(out_host, c.dout.p, (size_t)2 * (*n_out) * 4, hipMemcpyDeviceToHost));
  return ASX_OK;
}

// ---- BagOfModels combine on the device (every member keeps its own engine / weights; no float stem visits the host) -------
int asx_ht_standardize_dev(asx_engine *e, const float *mix_dev, int64_t N, float *out_dev, void *stream) {
  REQUIRE(e && mix_dev && out_dev && N >= 2, "asx_ht_standardize_dev: bad argument");
  REQUIRE(e->ht != nullptr, "asx_ht_standardize_dev: no Demucs model committed on this engine");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  HIPCHK(hipSetDevice(e->device));
  CHK(ht_ref_stats(e, mix_dev, N, s));
  return timed(e, ASX_PROF_MISC, 0.0, 16.0 * N, s, [&]() {
    hipLaunchKernelGGL(ht_standardize_kernel, dim3((unsigned)((2 * N + 255) / 256)), dim3(256), 0, s, mix_dev, 2 * N, N,
                       reinterpret_cast<const double *>(e->ht->ref_acc.p), out_dev);
  });
}

int asx_ht_bag_accumulate_dev(asx_engine *e, float *est_dev, const float *member_dev, const float *weights, int32_t S, int64_t N,
                              int32_t first, void *stream) {
  REQUIRE(e && est_dev && member_dev && weights && S >= 1 && S <= 16 && N >= 1, "asx_ht_bag_accumulate_dev: bad argument");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  HIPCHK(hipSetDevice(e->device));
  BagWeights w{};
  for (int i = 0; i < S; ++i) w.v[i] = weights[i];
  const int64_t per = 2 * N;
  return timed(e, ASX_PROF_MISC, 0.0, 12.0 * S * per, s, [&]() {
    hipLaunchKernelGGL(ht_bag_accumulate_kernel, dim3((unsigned)((per + 255) / 256), (unsigned)S), dim3(256), 0, s, est_dev, member_dev, per, w,
                       (int)first);
  });
}

int asx_ht_bag_finish_dev(asx_engine *e, const float *est_dev, const float *totals, int32_t S, const float *mix_dev, int64_t N,
                          uint32_t flags, float *out_dev, void *stream) {
  REQUIRE(e && est_dev && totals && out_dev && S >= 1 && S <= 16 && N >= 2, "asx_ht_bag_finish_dev: bad argument");
  REQUIRE(est_dev != out_dev, "asx_ht_bag_finish_dev: est and out must be different buffers (the stem swap is not in place)");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  HIPCHK(hipSetDevice(e->device));
  const int standardize = (flags & ASX_HT_STANDARDIZE) ? 1 : 0;
  if (standardize) {
    REQUIRE(e->ht != nullptr && mix_dev != nullptr, "asx_ht_bag_finish_dev: de-standardising needs the mix and a committed Demucs model");
    CHK(ht_ref_stats(e, mix_dev, N, s));
  }
  BagWeights t{};
  for (int i = 0; i < S; ++i) t.v[i] = totals[i];
  const int64_t per = 2 * N;
  return timed(e, ASX_PROF_MISC, 0.0, 8.0 * S * per, s, [&]() {
    hipLaunchKernelGGL(ht_bag_finish_kernel, dim3((unsigned)((per + 255) / 256), (unsigned)S), dim3(256), 0, s, est_dev, per, N, t,
                       standardize ? reinterpret_cast<const double *>(e->ht->ref_acc.p) : nullptr, standardize,
                       (flags & ASX_HT_SWAP01) ? 1 : 0, out_dev);
  });
}

int asx_set_option(asx_engine *e, const char *key, int32_t value) {
  REQUIRE(e && key, "asx_set_option: null argument");
  if (!strcmp(key, "winograd")) {
#ifndef ASX_EXPERIMENTAL_KERNELS
    REQUIRE(value <= 0 || value == 3, "asx_set_option: winograd = %d names a superseded kernel generation that only an experimental build carries "
            "(python build.py --experimental); this library has 0 (direct kernel) and 3 (the default)", (int)value);
#endif
    e->winograd = value < 0 ? 0 : (int)value;
    return ASX_OK;
  }
  if (!strcmp(key, "winograd_stationary")) {
#ifndef ASX_EXPERIMENTAL_KERNELS
    REQUIRE(value <= 0, "asx_set_option: the weight-stationary Winograd kernel (measured slower) is in experimental builds only (python build.py --experimental)");
#endif
    e->winos = value < 0 ? 0 : (int)value;
    return ASX_OK;
  }
  if (!strcmp(key, "winograd_bf16x6")) {
    e->wino6 = value < 0 ? 0 : (int)value;
    return ASX_OK;
  }
  if (!strcmp(key, "gemm_bf16x6")) {                 // this engine only (round 5; it was process-wide before)
    e->gemm_bf16x6 = value > 0 ? 1 : 0;
    return ASX_OK;
  }
  if (!strcmp(key, "gemm_f16x3")) {
    e->gemm_f16x3 = value > 0 ? 1 : 0;
    return ASX_OK;
  }
  if (!strcmp(key, "conv_direct_f16x3")) {
    e->conv3h = value < 0 ? 0 : (int)value;
    return ASX_OK;
  }
  if (!strcmp(key, "conv_down_bf16x6")) {
    e->down6 = value > 0 ? 1 : 0;
    return ASX_OK;
  }
  if (!strcmp(key, "conv_up_bf16x6")) {
    e->up6 = value > 0 ? 1 : 0;
    return ASX_OK;
  }
  if (!strcmp(key, "gemm_pair_images")) {
#ifndef ASX_EXPERIMENTAL_KERNELS
    if (value > 0) {
      set_err("asx_set_option: gemm_pair_images = 1 needs an experimental build (python build.py --experimental): the pair-image reader measured no faster in the nets and is not in the default library");
      return ASX_ERR_INVALID;
    }
#endif
    e->pair_images = value > 0 ? 1 : 0;
    return ASX_OK;
  }
  set_err("asx_set_option: unknown option '%s'", key);
  return ASX_ERR_INVALID;
}

// ---- profiling -------------------------------------------------------------------
int asx_profile_enable(asx_engine *e, int32_t on) {
  REQUIRE(e, "asx_profile_enable: null engine");
  for (auto &r : e->recs) {
    (void)hipEventDestroy(r.a);
    (void)hipEventDestroy(r.b);
  }
  e->recs.clear();
  e->prof = on != 0;
  return ASX_OK;
}

int asx_profile_read(asx_engine *e, asx_profile *out) {
  REQUIRE(e && out, "asx_profile_read: null argument");
  HIPCHK(hipSetDevice(e->device));
  HIPCHK(hipDeviceSynchronize());
  memset(out, 0, sizeof(*out));
  static const bool dump = getenv("ASX_PROF_DUMP") != nullptr;   // one line per launch on stderr (tuning aid)
  for (auto &r : e->recs) {
    float ms = 0.f;
    HIPCHK(hipEventElapsedTime(&ms, r.a, r.b));
    if (dump) fprintf(stderr, "asx_prof cls=%d ms=%.4f gflop=%.3f mbytes=%.2f\n", r.cls, ms, r.flops * 1e-9, r.bytes * 1e-6);
    out->launches[r.cls] += 1;
    out->ms[r.cls] += ms;
    out->flops[r.cls] += r.flops;
    out->bytes[r.cls] += r.bytes;
  }
  return ASX_OK;
}

int asx_profile_launches(asx_engine *e, asx_launch_rec *out, int32_t cap, int32_t *n) {
  REQUIRE(e && n && (out || cap == 0), "asx_profile_launches: null argument");
  HIPCHK(hipSetDevice(e->device));
  HIPCHK(hipDeviceSynchronize());
  *n = (int32_t)e->recs.size();
  for (int32_t i = 0; i < *n && i < cap; ++i) {
    float ms = 0.f;
    HIPCHK(hipEventElapsedTime(&ms, e->recs[i].a, e->recs[i].b));
    out[i].cls = e->recs[i].cls | (e->recs[i].nprod << 8);
    out[i].ms = ms;
    out[i].flops = e->recs[i].flops;
    out[i].bytes = e->recs[i].bytes;
  }
  return ASX_OK;
}

}  // extern "C"
