/*
 * asx.h -- C ABI of the MI355X-native demix engine (libasx.so).
 *
 * Drop-in boundary for ONE hot path of nomadkaraoke/python-audio-separator: the
 * chunked-spectrogram demix loop behind its architecture plugins.  Primary: the MDX
 * plugin (audio_separator/separator/architectures/mdx_separator.py, MDXSeparator.demix
 * :293-412 and run_model :414-450, with uvr_lib_v5/stft.py:20-126 and the ConvTDFNet
 * graph of uvr_lib_v5/mdxnet.py:30-120 that the reference executes through onnxruntime
 * at mdx_separator.py:122-123).  Sibling loops behind the same handle: MDXC (TFC-TDF v3,
 * BS / Mel-Band Roformer: mdxc_separator.py:257-468), Demucs v4 / v3
 * (demucs_separator.py:162-194, uvr_lib_v5/demucs/apply.py:124-260), VR
 * (vr_separator.py:255-375), and the normalise / quantise / ensemble / invert edges
 * (common_separator.py:309-337, ensembler.py, uvr_lib_v5/spec_utils.py:99,573).
 *
 * Plain C: pointers and sizes only, no torch/STL types.  Every function returns
 * ASX_OK (0) or a positive error code and never throws; the message of the last
 * failing call on the calling thread is available from asx_last_error().
 *
 * Ownership: the caller owns every buffer it passes in; the engine owns its
 * device workspace and copies weights at asx_net_commit().  One engine is bound
 * to one GPU; calls on one engine must be serialised by the caller (the
 * reference object is not re-entrant either).  "host" pointers are
 * pageable or pinned host memory, "dev" pointers are HIP device pointers on the
 * engine's GPU.  `stream` is a hipStream_t passed as void* (NULL = the null
 * stream).  asx_demix_dev / asx_demix_chunks_dev / asx_finalize_dev / asx_separate_dev, the
 * asx_mdxc_*_dev, asx_rof_*_dev, asx_ht_*_dev and asx_hd_*_dev calls only enqueue work on `stream` once the
 * engine's workspace has reached its size and the split weight images of the bf16 x 6 kernels exist (both happen in the first
 * call after weights were loaded: the images are built lazily, with one stream synchronisation each): no host copy, no host
 * synchronisation afterwards (chunk / segment
 * start tables are built by a kernel or travel in the kernel arguments) -- they can be captured into a hipGraph
 * and a collective on step k can overlap the compute of step k + 1.  asx_vr_separate_dev synchronises `stream`
 * only when enable_post_process asks for merge_artifacts (its run-length pass is host code).
 *
 * All audio is float32.  Layouts are C-order.
 */
#ifndef ASX_H
#define ASX_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ASX_OK 0
#define ASX_ERR_INVALID 1 /* bad argument / inconsistent configuration       */
#define ASX_ERR_HIP 2     /* a HIP runtime call failed (no GPU, OOM, launch) */
#define ASX_ERR_STATE 3   /* call order violated (e.g. demix before commit)  */

#define ASX_ABI_VERSION 7

/* flags of asx_demix*(): */
#define ASX_FLAG_MATCH_MIX 1u /* demix(mix, is_match_mix=True): overlap 0.02, no net (mdx_separator.py:308-313, :429-432) */

typedef struct asx_engine asx_engine;

/* Scalars of MDXSeparator that shape the path.
 *   n_fft        model_data["mdx_n_fft_scale_set"]   mdx_separator.py:72
 *   hop_length   arch_config["hop_length"]           mdx_separator.py:59
 *   dim_f        model_data["mdx_dim_f_set"]         mdx_separator.py:70
 *   segment_size arch_config["segment_size"]         mdx_separator.py:31   (frames per chunk)
 *   overlap      arch_config["overlap"]              mdx_separator.py:36
 *   enable_denoise arch_config["enable_denoise"]     mdx_separator.py:62
 *   max_batch    chunks processed per device batch.  Engine knob: the reference's
 *                batch_size never batches chunks and results do
 *                not depend on it.  0 = engine default.
 */
typedef struct asx_mdx_config {
  int32_t n_fft;
  int32_t hop_length;
  int32_t dim_f;
  int32_t segment_size;
  double overlap;       /* a Python float: step = int((1 - overlap) * chunk_size) is evaluated in double, mdx_separator.py:335 */
  int32_t enable_denoise;
  int32_t max_batch;
  int32_t win_length;   /* torch.stft / istft win_length: a periodic Hann window of this length, zero padded to n_fft at both
                         * ends (BSRoformer stft_win_length, bs_roformer.py:376-377); 0 = n_fft */
  int32_t reserved;     /* 0 */
} asx_mdx_config;

/* ConvTDFNet hyper-parameters (uvr_lib_v5/mdxnet.py:31-52); dim_t must equal
 * segment_size, dim_f must equal asx_mdx_config.dim_f.  tdf_bias != 0 when the
 * TDF Linear layers carry a bias (modules.py:57-66).
 * bn (modules.py:52-70): > 0 = Linear(f, f / bn) + norm + ReLU + Linear(f / bn, f) + norm + ReLU; 0 = ONE Linear(f, f) + norm + ReLU
 * (tensors <blk>.tdf0.* only); -1 = `bn is None`, no TDF branch (ABI 6).
 * norm (ABI 6; mdxnet.py:45-49): 0 = BatchNorm2d (optimizer 'rmsprop'), folded into the weights by the host; 1 = GroupNorm(2, c)
 * (optimizer 'adamw'): statistics depend on the input, so every conv / linear tensor arrives UNFOLDED and each normalised layer
 * carries its affine as "<layer>.gn_w" / "<layer>.gn_b" [channels] (layers: first, <blk>.tfc<j>, <blk>.tdf0, <blk>.tdf1, ds<i>, us<i>);
 * the TDF tensors then have no .scale / .shift. */
typedef struct asx_net_config {
  int32_t dim_c;
  int32_t dim_f;
  int32_t dim_t;
  int32_t g;
  int32_t l;
  int32_t num_blocks;
  int32_t k;
  int32_t bn;
  int32_t tdf_bias;
  int32_t norm;
} asx_net_config;

/* Index arithmetic of one demix() call (mdx_separator.py:308-348). */
typedef struct asx_plan {
  int64_t n_samples;  /* N                                    */
  int64_t padded_len; /* L = trim + N + pad                   */
  int64_t chunk_size; /* hop * (segment_size - 1)             */
  int64_t gen_size;   /* chunk_size - 2 * trim                */
  int64_t pad;
  int64_t step;       /* int((1 - overlap) * chunk_size)      */
  int32_t trim;       /* n_fft / 2                            */
  int32_t n_chunks;   /* len(range(0, L, step))               */
  int32_t n_frames;   /* frames per chunk = segment_size      */
  int32_t reserved;
} asx_plan;

/* Per-kernel-class device time, collected with hipEvents on the launch stream
 * while profiling is enabled. */
#define ASX_PROF_STFT 0
#define ASX_PROF_CONV3X3 1
#define ASX_PROF_TDF 2
#define ASX_PROF_DOWN 3
#define ASX_PROF_UP 4
#define ASX_PROF_CONV1X1 5
#define ASX_PROF_ISTFT 6
#define ASX_PROF_OLA 7
#define ASX_PROF_FINALIZE 8
#define ASX_PROF_MISC 9
#define ASX_PROF_NCLASS 10
typedef struct asx_profile {
  int64_t launches[ASX_PROF_NCLASS];
  double ms[ASX_PROF_NCLASS];    /* summed kernel time                       */
  double flops[ASX_PROF_NCLASS]; /* algorithmic 2*MAC of those launches      */
  double bytes[ASX_PROF_NCLASS]; /* algorithmic HBM bytes of those launches  */
} asx_profile;

/* ---- lifetime ---------------------------------------------------------- */
int asx_abi_version(void);
const char *asx_last_error(void);
/* Number of visible HIP devices (0 when there is none; never fails). */
int asx_device_count(void);
int asx_engine_create(int device, const asx_mdx_config *cfg, asx_engine **out);
void asx_engine_destroy(asx_engine *e);
/* The `window=` of torch.stft / istft (ABI 6): `n` = n_fft floats -- a window function evaluated over win_length and zero padded to
 * n_fft at both ends, as torch does -- replacing the periodic Hann the engine builds.  BSRoformer's `stft_window_fn`
 * (uvr_lib_v5/roformer/bs_roformer.py:333, 386).  Call it before the first forward (it synchronises the device). */
int asx_set_stft_window(asx_engine *e, const float *window_host, int32_t n);

/* ---- model weights: replaces ort.InferenceSession(model_path)  mdx_separator.py:122 ----
 * The host hands over BatchNorm-folded fp32 tensors by canonical name, the
 * engine packs them into its kernel layouts and uploads them at commit.
 * Names (blk = enc<i> | mid | dec<i>, i counted like mdxnet.py:63-93):
 *   first.w [g,dim_c]  first.b [g]                       first_conv   mdxnet.py:54-58
 *   <blk>.tfc<j>.w [c,c,k,k]  <blk>.tfc<j>.b [c]         TFC convs    modules.py:11-17
 *   <blk>.tdf0.w [f/bn,f]  .bias [f/bn]  .scale [c]  .shift [c]   TDF  modules.py:62-70
 *   <blk>.tdf1.w [f,f/bn]  .bias [f]     .scale [c]  .shift [c]
 *   ds<i>.w [c+g,c,2,2]  ds<i>.b [c+g]                   mdxnet.py:66-72
 *   us<i>.w [c,c-g,2,2]  us<i>.b [c-g]                   mdxnet.py:80-86 (ConvTranspose2d layout)
 *   final.w [dim_c,g]  final.b [dim_c]                   mdxnet.py:93-95
 * TDF epilogue: relu(scale[c] * (x @ w.T + bias) + shift[c]).
 */
int asx_net_begin(asx_engine *e, const asx_net_config *cfg);
int asx_net_set_tensor(asx_engine *e, const char *name, const float *host, int64_t numel);
int asx_net_commit(asx_engine *e);
/* Algorithmic FLOPs (2*MAC, conv + linear) of one forward over `batch` chunks. */
double asx_net_flops(const asx_engine *e, int32_t batch);

/* ---- the path ---------------------------------------------------------- */
/* Index plan of demix() for a mix of n_samples per channel. */
int asx_plan_query(const asx_engine *e, int64_t n_samples, uint32_t flags, asx_plan *out);

/* MDXSeparator.demix (mdx_separator.py:293): mix [2,N] -> out [2,N]. */
int asx_demix(asx_engine *e, const float *mix_host, int64_t n_samples, float *out_host, uint32_t flags);
int asx_demix_dev(asx_engine *e, const float *mix_dev, int64_t n_samples, float *out_dev, uint32_t flags,
                  void *stream);

/* Sharded form: chunks [chunk_begin, chunk_end) of the plan are
 * run and their windowed outputs written to chunk_out_dev[(k - chunk_begin), 2, chunk_size];
 * asx_finalize_dev() then folds ALL n_chunks windowed chunks (gathered by the
 * caller) into out [2,N] = (result / divider)[trim:-trim][:N]  (mdx_separator.py:386-401). */
int asx_demix_chunks_dev(asx_engine *e, const float *mix_dev, int64_t n_samples, int32_t chunk_begin,
                         int32_t chunk_end, float *chunk_out_dev, uint32_t flags, void *stream);
int asx_finalize_dev(asx_engine *e, const float *chunk_out_dev, int64_t n_samples, float *out_dev,
                     uint32_t flags, void *stream);

/* The array part of MDXSeparator.separate (mdx_separator.py:155-182) without leaving the device:
 *   peak = max|mix|; normalize(mix, max_peak, min_peak) IN PLACE (uvr_lib_v5/spec_utils.py:99-115);
 *   primary = demix(mix).T * peak; secondary = (-primary * compensate) + mix.T.
 * mix [2,N] is overwritten with the normalised mix (like the reference's in-place normalize);
 * primary / secondary are [N,2] (interleaved stereo, the layout write_audio consumes,
 * common_separator.py:330-337).  has_min_peak = 0 mirrors min_peak=None. */
int asx_separate(asx_engine *e, float *mix_host, int64_t n_samples, float max_peak, float min_peak,
                 int32_t has_min_peak, float compensate, float *primary_host, float *secondary_host);
int asx_separate_dev(asx_engine *e, float *mix_dev, int64_t n_samples, float max_peak, float min_peak,
                     int32_t has_min_peak, float compensate, float *primary_dev, float *secondary_dev, void *stream);

/* ---- MDXC / TFC-TDF v3 (MDX23C): uvr_lib_v5/tfc_tdf_v3.py + the TFC branch of MDXCSeparator.demix ----
 * The engine's n_fft / hop_length / dim_f / segment_size (asx_mdx_config) are config.audio.* and
 * inference.dim_t (or the overridden segment size, mdxc_separator.py:354-359).
 *   norm: 0 = None, 1 = InstanceNorm (affine)          tfc_tdf_v3.py:55-69
 *   act:  0 = relu, 1 = gelu                            tfc_tdf_v3.py:72-80
 * Weights are handed over with asx_net_set_tensor() under the reference's own state_dict keys
 * (e.g. "encoder_blocks.0.tfc_tdf.blocks.1.tfc1.2.weight"); scale is fixed to [2, 2]. */
typedef struct asx_v3_config {
  int32_t num_channels;         /* config.audio.num_channels (2)        */
  int32_t num_subbands;         /* config.model.num_subbands            */
  int32_t num_scales;           /* config.model.num_scales              */
  int32_t num_blocks_per_scale; /* config.model.num_blocks_per_scale    */
  int32_t num_channels_model;   /* config.model.num_channels            */
  int32_t growth;               /* config.model.growth                  */
  int32_t bottleneck_factor;    /* config.model.bottleneck_factor       */
  int32_t norm;
  int32_t act;
  int32_t num_targets;          /* 1 if training.target_instrument else len(training.instruments) */
} asx_v3_config;
int asx_v3_begin(asx_engine *e, const asx_v3_config *cfg);
int asx_v3_commit(asx_engine *e);
double asx_v3_flops(const asx_engine *e, int32_t batch);
/* TFC_TDF_net.forward (tfc_tdf_v3.py:230): wave [B,2,chunk] -> [B,S,2,chunk] (S = num_targets). */
int asx_v3_forward(asx_engine *e, const float *wave_host, int32_t batch, float *out_host);
/* Index arithmetic of the TFC branch (mdxc_separator.py:361-372): chunk_size, step = hop_size,
 * pad = pad_size, trim = chunk_size - hop_size (front zeros), padded_len, n_chunks. */
int asx_mdxc_plan(const asx_engine *e, int64_t n_samples, int32_t overlap, asx_plan *out);
/* MDXCSeparator.demix, TFC branch (mdxc_separator.py:345-404): mix [2,N] -> out [S,2,N]
 * (= accumulated[..., chunk-hop : -(pad+chunk-hop)] / overlap). */
int asx_mdxc_demix(asx_engine *e, const float *mix_host, int64_t n_samples, int32_t overlap, float *out_host);
/* halves of asx_mdxc_demix_dev for multi-GPU sharding: chunk_out [n_chunks (asx_mdxc_plan), S, 2, chunk_size] */
int asx_mdxc_chunks_dev(asx_engine *e, const float *mix_dev, int64_t n_samples, int32_t overlap, int32_t k0, int32_t k1,
                        float *chunk_out_dev, void *stream);
int asx_mdxc_finalize_dev(asx_engine *e, const float *chunk_out_dev, int64_t n_samples, int32_t overlap, float *out_dev, void *stream);
int asx_mdxc_demix_dev(asx_engine *e, const float *mix_dev, int64_t n_samples, int32_t overlap, float *out_dev,
                       void *stream);

/* ---- BS-Roformer: uvr_lib_v5/roformer/bs_roformer.py + the Roformer branch of MDXCSeparator.demix ----
 * Engine geometry (asx_mdx_config): n_fft = stft_n_fft (= stft_win_length), hop_length =
 * stft_hop_length, dim_f = n_fft/2 + 1, segment_size = inference.dim_t (mdxc_separator.py:276-300).
 * Weights come in under the reference's own state_dict keys.  dim_head must be 64.  Covers BSRoformer and
 * MelBandRoformer (cfg.mel). */
typedef struct asx_rof_config {
  int32_t dim, depth, heads, dim_head;        /* roformer_loader.py:123-135 */
  int32_t num_stems;
  int32_t time_depth, freq_depth;             /* time_/freq_transformer_depth */
  int32_t mlp_expansion_factor, mask_estimator_depth;
  int32_t n_bands;
  int32_t n_out;                              /* len(training.instruments): rows of the result (mdxc_separator.py:316) */
  int32_t freqs_per_bands[128];
  /* Mel-Band Roformer (uvr_lib_v5/roformer/mel_band_roformer.py): overlapping mel bands -- band j covers frequency bins
   * [band_start[j], band_start[j] + freqs_per_bands[j]) (the support of librosa.filters.mel, computed by the host); band
   * masks are summed onto their bins and divided by the number of covering bands (:404-416); every Transformer ends
   * with its own RMSNorm ("layers.i.k.norm.gamma") instead of one final_norm; the mask MLP has depth+1 linears of
   * hidden width 4*dim. */
  int32_t mel;
  int32_t band_start[128];
  int32_t stft_normalized;                    /* ABI 6: torch.stft / istft normalized=True (bs_roformer.py:332, 384) */
} asx_rof_config;
int asx_rof_begin(asx_engine *e, const asx_rof_config *cfg);
int asx_rof_commit(asx_engine *e);
double asx_rof_flops(const asx_engine *e, int32_t batch);
/* BSRoformer.forward (bs_roformer.py:418): wave [B,2,chunk] -> [B,S,2,chunk]. */
int asx_rof_forward(asx_engine *e, const float *wave_host, int32_t batch, float *out_host);
/* Roformer branch of MDXCSeparator.demix (mdxc_separator.py:272-343): Hamming-weighted fold with the last
 * chunk re-anchored at the tail; `step` = min(int(overlap * sample_rate), chunk_size) in samples.
 * mix [2,N] (N >= chunk_size) -> out [n_out,2,N]. */
int asx_rof_demix(asx_engine *e, const float *mix_host, int64_t n_samples, int64_t step, float *out_host);
int asx_rof_demix_dev(asx_engine *e, const float *mix_dev, int64_t n_samples, int64_t step, float *out_dev,
                      void *stream);
/* halves of asx_rof_demix_dev for multi-GPU sharding: chunk_out [n_chunks, S, 2, chunk_size] */
int asx_rof_plan(const asx_engine *e, int64_t n_samples, int64_t step, int32_t *n_chunks, int64_t *chunk_size);
int asx_rof_chunks_dev(asx_engine *e, const float *mix_dev, int64_t n_samples, int64_t step, int32_t k0, int32_t k1,
                       float *chunk_out_dev, void *stream);
int asx_rof_finalize_dev(asx_engine *e, const float *chunk_out_dev, int64_t n_samples, int64_t step, float *out_dev, void *stream);

/* ---- Demucs v4 / HTDemucs -----------------------------------------------------
 * Replaces HTDemucs(**kwargs) + load_state_dict (uvr_lib_v5/demucs/htdemucs.py:32-382, demucs_separator.py:121-134),
 * HTDemucs.forward (htdemucs.py:483-620), apply_model (uvr_lib_v5/demucs/apply.py:124-260) and
 * DemucsSeparator.demix_demucs (architectures/demucs_separator.py:162-194).
 * The engine's asx_mdx_config is not used by this path.  Built structure: every layer a frequency layer
 * (nfft/2 / stride^depth > 1), kernel 8 / stride 4, DConv in the encoders, CaC, sinusoidal embeddings, norm_first
 * transformer with LayerScale and norm_out, head dim 48 or 64.  Weights come in under the checkpoint's own
 * state_dict keys; optional "pos_emb_freq" [T*Fr, C] / "pos_emb_time" [L, C] override the sinusoidal tables. */
typedef struct asx_ht_config {
  int32_t n_sources;                     /* len(sources) */
  int32_t channels, growth, nfft, depth; /* htdemucs.py:40-47 */
  int32_t kernel_size, stride;           /* 8, 4 */
  int32_t dconv_depth, dconv_comp;       /* 2, 8 */
  int32_t bottom_channels;               /* 0 = none */
  int32_t t_layers, t_heads, t_hidden;   /* t_hidden = int(C * t_hidden_scale) */
  int32_t samplerate;
  int32_t segment_samples;               /* int(segment * samplerate): training length = split segment */
  float freq_emb_scale;                  /* freq_emb (0.2); 0 = no embedding */
  int32_t max_batch;                     /* segments per forward batch (0 = 16) */
} asx_ht_config;
#define ASX_HT_STANDARDIZE 1u /* (mix - ref.mean()) / ref.std() before, * std + mean after (demucs_separator.py:171-185) */
#define ASX_HT_SWAP01 2u      /* sources[[0, 1]] = sources[[1, 0]] (demucs_separator.py:187) */
int asx_ht_begin(asx_engine *e, const asx_ht_config *cfg);
int asx_ht_commit(asx_engine *e);
double asx_ht_flops(const asx_engine *e);   /* 2*MAC of one segment */
/* HTDemucs.forward: mix [B, 2, length] (length <= segment_samples, zero padded like htdemucs.py:497-502)
 * -> out [B, S, 2, length]. */
int asx_ht_forward(asx_engine *e, const float *mix_host, int32_t batch, int64_t length, float *out_host);
/* apply_model(model, mix[None], shifts, split=True, overlap) (+ the demix_demucs framing selected by `flags`):
 * mix [2, N] -> out [S, 2, N].  offsets[shifts]: the random.randint(0, samplerate/2) draws (apply.py:209) made by
 * the host; shifts = 0 -> no shift trick. */
int asx_ht_demix(asx_engine *e, const float *mix_host, int64_t n_samples, int32_t shifts, const int64_t *offsets,
                 double overlap, uint32_t flags, float *out_host);
int asx_ht_demix_dev(asx_engine *e, const float *mix_dev, int64_t n_samples, int32_t shifts, const int64_t *offsets,
                     double overlap, uint32_t flags, float *out_dev, void *stream);
/* The two halves of asx_ht_demix_dev for multi-GPU sharding: the segment-forwards of a call form one list
 * (shift 0's chunks, shift 1's, ...; asx_ht_plan gives its length); a rank runs any sub-range, the folding rank needs all
 * of them: chunk_out [n_segments, S, 2, segment_samples]. */
int asx_ht_plan(const asx_engine *e, int64_t n_samples, int32_t shifts, const int64_t *offsets, double overlap, int32_t *n_segments,
                int64_t *segment_samples);
int asx_ht_segments_dev(asx_engine *e, const float *mix_dev, int64_t n_samples, int32_t shifts, const int64_t *offsets, double overlap,
                        uint32_t flags, int32_t k0, int32_t k1, float *chunk_out_dev, void *stream);
int asx_ht_fold_dev(asx_engine *e, const float *mix_dev, int64_t n_samples, int32_t shifts, const int64_t *offsets, double overlap,
                    uint32_t flags, const float *chunk_out_dev, float *out_dev, void *stream);

/* ---- Demucs v3 (HDemucs: the `hdemucs_mmi` entry of the reference's Demucs list) --------------------------
 * Replaces HDemucs(**kwargs) + load_state_dict (uvr_lib_v5/demucs/hdemucs.py:362-571) and HDemucs.forward
 * (:670-782) under the same apply_model / demix_demucs framing as Demucs v4 (apply.py:124-260,
 * architectures/demucs_separator.py:162-194).  Structure built: the class defaults -- CaC, depth - 2 strided levels
 * on both branches, the last-frequency level with the waveform branch injected, one time-only level; GroupNorm,
 * BLSTM(max_steps 200) and LocalState on those two innermost levels.  HDemucs has no valid_length: every chunk
 * of apply_model runs at its own length (apply.py:251-256), so asx_hd_forward takes any length >= 1 (short inputs get
 * pad1d's zero extension before the reflection, hdemucs.py:21-34).
 * Weights come in under the checkpoint's own state_dict keys (asx_net_set_tensor). */
typedef struct asx_hd_config {
  int32_t n_sources;
  int32_t channels, growth, nfft, depth;     /* 48, 2, 4096, 6 */
  int32_t kernel_size, stride, time_stride;  /* 8, 4, 2 */
  int32_t norm_starts, norm_groups;          /* depth - 2, 4 */
  int32_t dconv_depth, dconv_comp;           /* 2, 4 */
  int32_t dconv_attn, dconv_lstm;            /* depth - 2 */
  int32_t samplerate;
  int32_t segment_samples;                   /* int(samplerate * segment): the split length of apply_model */
  float freq_emb_scale;                      /* freq_emb (0.2); 0 = no embedding */
  int32_t max_batch;                         /* equal-length chunks per forward batch (0 = 16) */
} asx_hd_config;
int asx_hd_begin(asx_engine *e, const asx_hd_config *cfg);
int asx_hd_commit(asx_engine *e);
double asx_hd_flops(const asx_engine *e, int64_t length);   /* 2*MAC of one chunk of `length` samples */
/* HDemucs.forward: mix [B, 2, length] -> out [B, S, 2, length] */
int asx_hd_forward(asx_engine *e, const float *mix_host, int32_t batch, int64_t length, float *out_host);
/* apply_model(model, mix[None], shifts, split=True, overlap) + the demix_demucs framing in `flags` (ASX_HT_*):
 * mix [2, N] -> out [S, 2, N]; offsets as for asx_ht_demix. */
int asx_hd_demix(asx_engine *e, const float *mix_host, int64_t n_samples, int32_t shifts, const int64_t *offsets, double overlap,
                 uint32_t flags, float *out_host);
int asx_hd_demix_dev(asx_engine *e, const float *mix_dev, int64_t n_samples, int32_t shifts, const int64_t *offsets, double overlap,
                     uint32_t flags, float *out_dev, void *stream);
/* sharded halves, as asx_ht_plan / asx_ht_segments_dev / asx_ht_fold_dev: chunk_out [n_segments, S, 2, segment_samples],
 * row k holds the chunk's own length of valid samples from column 0 */
int asx_hd_plan(const asx_engine *e, int64_t n_samples, int32_t shifts, const int64_t *offsets, double overlap, int32_t *n_segments,
                int64_t *segment_samples);
int asx_hd_segments_dev(asx_engine *e, const float *mix_dev, int64_t n_samples, int32_t shifts, const int64_t *offsets, double overlap,
                        uint32_t flags, int32_t k0, int32_t k1, float *chunk_out_dev, void *stream);
int asx_hd_fold_dev(asx_engine *e, const float *mix_dev, int64_t n_samples, int32_t shifts, const int64_t *offsets, double overlap,
                    uint32_t flags, const float *chunk_out_dev, float *out_dev, void *stream);

/* ---- VR architecture -----------------------------------------------------------------
 * Replaces nets.determine_model_capacity(...) + load_state_dict (architectures/vr_separator.py:168-176,
 * uvr_lib_v5/vr_network/nets.py:65-93), VRSeparator.loading_mix (:255-291), inference_vr (:293-366) and spec_to_wav
 * (:368-375) with the spec_utils functions they call (wave_to_spectrogram, combine_spectrograms, preprocess,
 * make_padding, adjust_aggr, merge_artifacts, cmb_spectrogram_to_wave, spectrogram_to_wave, fft_lp/hp_filter).
 * band[d-1] = modelparams JSON "band"[d] (uvr_lib_v5/vr_network/modelparams/ *.json).  Resampling between bands (ABI 5):
 *   ASX_VR_RES_POLYPHASE     scipy.signal.resample_poly as librosa.resample(res_type="polyphase") calls it;
 *   ASX_VR_RES_SINC_FASTEST  libsamplerate's SRC_SINC_FASTEST as librosa.resample(res_type="sinc_fastest") reaches it through
 *                            python-samplerate (float32 in / out, src_simple): the library's published algorithm (src_sinc.c) on a
 *                            regenerated Kaiser-sinc coefficient table -- the library and fastest_coeffs.h are not available, so this
 *                            converter is "parity unpinned" (INTEGRATION.md "VR resampler").
 * band[d-1].res_type converts band d+1 -> d in loading_mix (vr_separator.py:267,282: the band's own "res_type"; every type other
 * than "sinc_fastest" is served by the polyphase filter); synth_res_type converts band d -> d+1 in cmb_spectrogram_to_wave
 * (spec_utils.py:374,390: the module-level `wav_resolution`, "sinc_fastest" everywhere except macOS on ARM, :33-38).
 */
#define ASX_VR_RES_POLYPHASE 0
#define ASX_VR_RES_SINC_FASTEST 1
typedef struct asx_vr_band {
  int32_t sr, hl, n_fft, crop_start, crop_stop, hpf_start, hpf_stop, lpf_start, lpf_stop;
  int32_t convert;        /* VR 5.1 "convert_channels" (spec_utils.py:232-247): 0 none, 1 mid_side, 4 mid_side_c, 5 stereo_n */
  int32_t res_type;       /* ASX_VR_RES_*: analysis converter of this band (band d+1 -> d); ABI 5 */
} asx_vr_band;
typedef struct asx_vr_config {
  int32_t bins, n_bands, pre_filter_start, pre_filter_stop;
  int32_t channel_mode;   /* 0 stereo, 1 mid_side, 2 mid_side_b2, 3 reverse (model_param_init.py:62-65) */
  int32_t arch;           /* nn_arch_size (vr_separator.py:161-164) */
  int32_t cap[6];         /* stage-1 width, stage-2 bridge out, stage-2 width, stage-3 bridge out, stage-3 width, 0 (nets.py:74-86) */
  int32_t window_size;    /* arch_config["window_size"] */
  int32_t offset;         /* CascadedASPPNet.offset = 128 */
  int32_t max_batch;      /* patches per forward batch (0 = 4); no effect on the result */
  int32_t v51;            /* 1: VR 5.1 -- nets_new.CascadedNet(n_fft, nn_arch_size, nout = cap[0], nout_lstm = cap[1]) and the
                             is_v51_model branches of spec_utils (per-band convert_channels, get_lp/hp_filter_mask); offset 64 */
  int32_t synth_res_type; /* ASX_VR_RES_*: converter of the synthesis chain (`wav_resolution`, spec_utils.py:33-38); ABI 5 */
  asx_vr_band band[8];
} asx_vr_config;
typedef struct asx_vr_params {
  float aggr_value;        /* aggressiveness["value"] = int(aggression) / 100 */
  int32_t split_bin;       /* band[1].crop_stop */
  int32_t is_non_accom;    /* primary stem in CommonSeparator.NON_ACCOM_STEMS */
  int32_t has_corr;
  float corr_left, corr_right;   /* aggr_correction */
  int32_t enable_tta, enable_post_process;
  float post_thres;
  int32_t high_end_process;   /* arch_config["high_end_process"]: spec_utils.mirroring("mirroring") of the input's high end */
} asx_vr_params;
int asx_vr_begin(asx_engine *e, const asx_vr_config *cfg);
int asx_vr_commit(asx_engine *e);
double asx_vr_flops(const asx_engine *e);   /* 2*MAC of the net on one window */
/* frames of the combined spectrogram and length of the separated waves for n_samples input samples */
int asx_vr_plan(const asx_engine *e, int64_t n_samples, int32_t *n_frames, int64_t *n_out);
/* CascadedASPPNet.forward, eval (nets.py:132-161): x [B, 2, bins+1, window_size] -> mask, same shape. */
int asx_vr_forward(asx_engine *e, const float *x_host, int32_t batch, float *out_host);
/* loading_mix: wave [2, n] (float32 at the top band's rate) -> X_spec [2, n_frames, bins+1] complex64 (re, im pairs;
 * note: frames outer, bins inner -- the transpose of the reference's [2, bins+1, n_frames]). */
int asx_vr_analysis(asx_engine *e, const float *wave_host, int64_t n_samples, float *spec_host);
/* the array part of VRSeparator.separate: wave [2, n] -> spec_to_wav(y_spec) [2, n_out], spec_to_wav(v_spec) [2, n_out]
 * (either output may be NULL). */
int asx_vr_separate(asx_engine *e, const float *wave_host, int64_t n_samples, const asx_vr_params *params, float *primary_host,
                    float *secondary_host);
int asx_vr_separate_dev(asx_engine *e, const float *wave_dev, int64_t n_samples, const asx_vr_params *params, float *primary_dev,
                        float *secondary_dev, void *stream);

/* librosa.resample(y, orig_sr, target_sr, res_type="sinc_fastest") for ANY ratio = float(target_sr) / orig_sr, i.e. libsamplerate's
 * SRC_SINC_FASTEST through python-samplerate as above (restated algorithm, regenerated table: parity unpinned).  Replaces
 * spec_utils.change_pitch_semitones (uvr_lib_v5/spec_utils.py:783-790; MDXC pitch_shift, mdxc_separator.py:230-243,268-270) and
 * is the test hook of the converter.  x [channels, n_in] planar -> y [channels, n_out], n_out = ceil(n_in * ratio) (librosa's
 * fix_length; frames the library does not generate are zero).  mono_calls != 0: every channel is its own one-channel call -- the
 * library's end-of-input test then drops the last frame when n_in * ratio is an integer (change_pitch_semitones resamples channel
 * by channel); 0: the channels are the interleaved channels of one call (the VR chain).  ABI 5. */
int asx_resample_sinc(asx_engine *e, const float *x_host, int32_t channels, int64_t n_in, double ratio, int32_t mono_calls,
                      float *y_host, int64_t n_out);
int asx_resample_sinc_dev(asx_engine *e, const float *x_dev, int32_t channels, int64_t n_in, double ratio, int32_t mono_calls,
                          float *y_dev, int64_t n_out, void *stream);

/* BagOfModels on the device (uvr_lib_v5/demucs/apply.py:169-196 + demucs_separator.py:171-189; ABI 4).  Every member of a
 * bag keeps its own engine (weights resident across files); the caller demixes the STANDARDISED mix with each member
 * (flags 0) and combines without a host round trip:
 *   asx_ht_standardize_dev     out = (mix - ref.mean()) / ref.std(), ref = mix.mean(0)          [2, N] -> [2, N]
 *   asx_ht_bag_accumulate_dev  est = first ? member * w[k] : est + member * w[k]                 [S, 2, N]; w: S host floats
 *   asx_ht_bag_finish_dev      out = est / totals[k] (* ref.std() + ref.mean() with ASX_HT_STANDARDIZE; sources 0 / 1 swapped
 *                              with ASX_HT_SWAP01); est and out are different buffers.
 * The engine passed in only provides the device, the stream plumbing and the mix statistics (any member's engine). */
int asx_ht_standardize_dev(asx_engine *e, const float *mix_dev, int64_t n_samples, float *out_dev, void *stream);
int asx_ht_bag_accumulate_dev(asx_engine *e, float *est_dev, const float *member_dev, const float *weights, int32_t n_sources,
                              int64_t n_samples, int32_t first, void *stream);
int asx_ht_bag_finish_dev(asx_engine *e, const float *est_dev, const float *totals, int32_t n_sources, const float *mix_dev,
                          int64_t n_samples, uint32_t flags, float *out_dev, void *stream);

/* Writer edge: spec_utils.normalize + (stem * 32767).astype(np.int16) + channel interleave of
 * CommonSeparator.write_audio_pydub (common_separator.py:309-337).  stem [2, N] planar -> pcm [N, 2]; bit-exact.
 * *peak_after (optional) = max |stem| after normalisation: the reference skips the file when it is < 1e-6. */
int asx_pcm16(asx_engine *e, const float *stem_host, int64_t n_samples, float max_peak, float min_peak, int32_t has_min,
              int16_t *pcm_host, float *peak_after);
int asx_pcm16_dev(asx_engine *e, const float *stem_dev, int64_t n_samples, float max_peak, float min_peak, int32_t has_min,
                  int16_t *pcm_dev, float *peak_after, void *stream);

/* The same edge for a stem that is ALREADY [N, 2] interleaved on the device -- the layout asx_separate_dev writes and
 * write_audio receives (common_separator.py:330-337) -- so that a file-level caller never brings the float stem to the host:
 * stem_rows [N, 2] -> pcm [N, 2]; bit-identical to asx_pcm16 on the transposed array.  (ABI 4) */
int asx_pcm16_rows_dev(asx_engine *e, const float *stem_rows_dev, int64_t n_samples, float max_peak, float min_peak, int32_t has_min,
                       int16_t *pcm_dev, float *peak_after, void *stream);

/* Decode edge (common_separator.py:217-282 prepare_mix -> librosa.load -> libsndfile): the `data` chunk of a RIFF/WAVE file,
 * [frames, channels] interleaved little-endian samples already copied to the device, -> float32 planar mix [2, frames]
 * (a mono file feeds both rows, :278-280; channels beyond the second are ignored).  sample_format: 16 / 24 / 32 = integer
 * PCM bits (x / 2^(bits-1), PCM_32 through double like libsndfile), 0x120 = IEEE float32.  *peak (optional, synchronises)
 * = max |mix|: prepare_mix raises on a silent file (:268-271).  (ABI 4) */
int asx_pcm_decode_dev(asx_engine *e, const void *raw_dev, int64_t frames, int32_t channels, int32_t sample_format, float *mix_dev,
                       float *peak, void *stream);

/* spec_utils.normalize(wave, max_peak, min_peak) (uvr_lib_v5/spec_utils.py:99-115) in place on any float32 array of
 * `numel` values (MDXCSeparator.separate applies it to the mix and to every stem, mdxc_separator.py:147,170-190);
 * *peak_before (optional) = max |wave| before scaling.  has_min_peak = 0 mirrors min_peak=None. */
int asx_normalize(asx_engine *e, float *wave_host, int64_t numel, float max_peak, float min_peak, int32_t has_min, float *peak_before);

/* The same on an array already in HBM, in place, no host synchronisation (ABI 4); and the residual stem of a single-target
 * MDXC / Roformer model, out = mix - stem (mdxc_separator.py:406-468). */
int asx_normalize_dev(asx_engine *e, float *wave_dev, int64_t numel, float max_peak, float min_peak, int32_t has_min, void *stream);
int asx_residual_dev(asx_engine *e, const float *mix_dev, const float *stem_dev, float *out_dev, int64_t numel, void *stream);

/* Spectral edges, librosa STFT(2048, 1024) semantics:
 * asx_ensemble   = Ensembler.ensemble (audio_separator/separator/ensembler.py:12-160) over K equal-length stereo waves
 *                  [K, 2, N]; algorithm: 0 avg_wave, 1 median_wave, 2 min_wave, 3 max_wave, 4 avg_fft, 5 median_fft,
 *                  6 min_fft, 7 max_fft, 8 uvr_max_spec, 9 uvr_min_spec, 10 ensemble_wav (spec_utils.py:1245-1266: each
 *                  channel from the input with the smallest mean |x|); weights [K] (avg_* only) or NULL; out [2, *n_out]
 *                  (*n_out = N, or 1024 * (N / 1024) for the uvr_* algorithms, which do not pass a length to istft).
 * asx_invert_stem = spec_utils.invert_stem(mixture, stem) (uvr_lib_v5/spec_utils.py:573-580), planar [2, *n_out]. */
int asx_ensemble(asx_engine *e, const float *waves_host, int32_t k, int64_t n_samples, int32_t algorithm, const double *weights,
                 float *out_host, int64_t *n_out);
int asx_ensemble_dev(asx_engine *e, const float *waves_dev, int32_t k, int64_t n_samples, int32_t algorithm, const double *weights,
                     float *out_dev, int64_t *n_out, void *stream);
int asx_invert_stem(asx_engine *e, const float *mix_host, const float *stem_host, int64_t n_samples, float *out_host, int64_t *n_out);

/* Launch counters of this process (tests use them to prove which kernel family ran; ABI 6 -- until ABI 5 they travelled through a
 * float of asx_debug_fetch, which stops resolving single launches past 2^24): "tdf3_launches" (split-operand row GEMM, either arithmetic),
 * "tdf3h_launches" (those of them, plain or GATHER mode, that ran the fp16 x 3 arithmetic),
 * "tdf3_gather_launches" (its GATHER mode: channels-last convolutions), "attn6_launches" (attention6_kernel / mha6_kernel),
 * "attn6h_launches" (those of them on the fp16 x 3 arithmetic),
 * "wino6_launches" (conv_wino6_kernel: Winograd F(2x2,3x3) on the 16-bit pipe), "wino6h_launches" (those on the fp16 x 3 arithmetic),
 * "conv3h_launches" (ABI 7: conv3h_kernel, the direct fp16 x 3 convolution of the 48-channel level),
 * "down6_launches" / "up6_launches" (ABI 7: conv_down6_kernel / conv_up6_kernel, the level-change convs on the 16-bit matrix pipe),
 * "tdf3_pair_image_launches" (ABI 7: the tdf3_kernel launches that read their x operand as a pair image -- option "gemm_pair_images", experimental builds).
 * ASX_ERR_INVALID for an unknown name. */
int asx_counter(const asx_engine *e, const char *name, int64_t *out);

/* bring-up hook: copy a named engine workspace buffer ("vr.hc", "vr.D0", ...) to the host. */
int asx_debug_fetch(asx_engine *e, const char *name, float *host, int64_t numel);
/* measurement hook: the s_memtime timeline the ASX_TDF2_ABL=16 build of the row GEMM records (8 x uint64 per workgroup). */
int asx_debug_trace(uint64_t *host, int64_t n_u64);

/* ---- stage hooks (host buffers; mirror the reference's own test surface) ---- */
/* STFT.__call__ (stft.py:20): wave [B,2,C] -> spec [B,4,dim_f,C/hop+1]. */
int asx_stft(asx_engine *e, const float *wave_host, int32_t batch, int64_t n_time, float *spec_host);
/* STFT.inverse (stft.py:99): spec [B,4,dim_f,T] -> wave [B,2,hop*(T-1)]. */
int asx_istft(asx_engine *e, const float *spec_host, int32_t batch, int32_t n_frames, float *wave_host);
/* model_run(spek) (mdx_separator.py:123): spec [B,dim_c,dim_f,dim_t] -> same shape. */
int asx_net_forward(asx_engine *e, const float *spec_host, int32_t batch, float *out_host);
/* MDXSeparator.run_model (mdx_separator.py:414): wave [B,2,chunk_size] -> [B,2,chunk_size]. */
int asx_run_model(asx_engine *e, const float *wave_host, int32_t batch, float *out_host, uint32_t flags);

/* Single-layer hooks used by the parity tests (activations [B,C,T,F], F fastest):
 *   op = "conv3x3"  w [cout,cin,3,3]  b [cout]                 -> relu(conv(x)+b)
 *   op = "down"     w [cout,cin,2,2]  b [cout]   stride 2      -> relu(conv(x)+b)         [B,cout,T/2,F/2]
 *   op = "up"       w [cin,cout,2,2]  b [cout]   stride 2, aux = skip [B,cout,2T,2F]  -> relu(convT(x)+b)*skip
 *   op = "conv1x1"  w [cout,cin]      b [cout]
 * `relu` = 0 drops the ReLU of any of them (ABI 6; until ABI 5 only conv1x1 honoured it): the GroupNorm variant of the net runs its
 * convolutions bare.
 *   op = "tdf"      w [n,k] bias [n] with x viewed as rows of length k=F;
 *                   aux0 = scale [C], aux1 = shift [C], aux2 = residual or NULL -> relu(scale*(xW^T+bias)+shift) (+res)
 */
int asx_op_conv(asx_engine *e, const char *op, const float *x_host, int32_t batch, int32_t cin, int32_t t,
                int32_t f, const float *w_host, const float *b_host, int32_t cout, const float *aux_host,
                int32_t relu, float *y_host);
int asx_op_tdf(asx_engine *e, const float *x_host, int32_t batch, int32_t c, int32_t t, int32_t k,
               const float *w_host, const float *bias_host, int32_t n, const float *scale_host,
               const float *shift_host, const float *res_host, float *y_host);
/* (ABI 7) The two linears of a TDF block with a bottleneck, launched the way the net launches them (reference: TFC_TDF.tdf, modules.py:61-70 --
 * Linear(f, f / bn) + norm + ReLU + Linear(f / bn, f) + norm + ReLU, then x + tdf(x)): y = x + relu(scale1 (relu(scale0 (x W0^T) + shift0) W1^T) + shift1)
 * on x [B, c, t, f] viewed as rows of length f, W0 [n8, f], W1 [f, n8], scales / shifts per channel [c].  With option "gemm_pair_images" and shapes
 * the fp16 x 3 row GEMM takes, the bottleneck activations travel between the two launches as a pair image.  h_host (optional, [B, c, t, n8]):
 * the bottleneck activations as fp32 (decoded from the pair image when one was written) -- single-op test hook like asx_op_tdf. */
int asx_op_tdf_block(asx_engine *e, const float *x_host, int32_t batch, int32_t c, int32_t t, int32_t f, const float *w0_host,
                     const float *scale0_host, const float *shift0_host, int32_t n8, const float *w1_host, const float *scale1_host,
                     const float *shift1_host, float *y_host, float *h_host);

/* ---- options ------------------------------------------------------------ */
/* "winograd": kernel of the 3x3 / pad-1 TFC convolutions.  3 (default; also ASX_WINOGRAD in the environment) = Winograd
 * F(2x2,3x3) in fp32 (2.25x fewer multiply-accumulates; results differ from the direct kernel by a few float32 ulps per
 * layer, whole-song deviation from the CPU oracle stays below 1e-5 relative RMS, profiles/r03_fullsong_parity.json);
 * 0 = the direct MFMA kernel; 1 / 2 = earlier Winograd generations kept for A/B measurements.
 * "winograd_stationary": 1 (also ASX_WINOS) = layers with at most 96 input channels run the weight-stationary form of the same
 * transform (csrc/kernels_winos.h: the transformed weights stay in registers, positions split over eight waves) when "winograd"
 * is 3; 0 (default: the stationary form measured slower, profiles/NOTES.md) = conv_wino3_kernel for every layer.
 * "winograd_bf16x6" (ABI 6; also ASX_WINO6): minimum input-channel count from which a 3x3 TFC convolution runs Winograd F(2x2,3x3) on the
 * bf16 matrix pipe (conv_wino6_kernel, csrc/kernels_wino6.h: the sixteen transform-domain GEMMs as six bf16 MFMA products on exactly
 * split operands -- the arithmetic of "gemm_bf16x6", which must be on) instead of conv_wino3_kernel; needs "winograd" = 3.  Default 144
 * (levels 2 .. 5 of the HQ_3 net: measured 1.08-1.24x faster there, equal at 96 channels, slower at 48); 0 = never.
 * "conv_direct_f16x3" (ABI 7; also ASX_CONV3H): 1 (default) = a 3x3 TFC convolution of 48 -> 48 channels on planes whose width is a multiple of
 * 32 (level 0 of the HQ_3 geometry, the widest planes of the net) runs conv3h_kernel (csrc/kernels_conv3h.h): a DIRECT implicit GEMM on the fp16
 * matrix pipe with the arithmetic of "gemm_f16x3" (which must be on, as "gemm_bf16x6" and "winograd" = 3) -- the two-part weight image stays in
 * LDS for the whole launch, producer waves fetch four input rows per step into a ring walked down T and split them under one running
 * power-of-two exponent per walk, consumer waves multiply; 5.3-5.8 ms per launch of 55 chunks against 8.5-8.9 on conv_wino3_kernel.  0 = conv_wino3_kernel.
 * "conv_down_bf16x6" / "conv_up_bf16x6" (ABI 7; also ASX_DOWN6 / ASX_UP6): 1 (default) = the 2 x 2 / stride-2 convolutions between the levels
 * (mdxnet.py:66-72) and the transposed ones of the decoder with their `x *= skip` (mdxnet.py:80-86, 113) run conv_down6_kernel / conv_up6_kernel
 * (csrc/kernels_updown6.h) while "gemm_bf16x6" is on: the arithmetic of that option -- both fp32 operands split EXACTLY into three bf16 parts, six
 * bf16 MFMA products per multiply-add, fp32 accumulation; the weights pre-split into a fragment-ordered image when they are loaded -- on
 * input planes the LDS-DMA brings in as fp32; 5.5 against 8.9 ms and 8.4 against 10.2 ms per 4-minute song, closer to float64 than the
 * fp32-MFMA kernels they replace.  0 = conv_dma_kernel<2, 2, 2, 0, ...> / <1, 1, 1, 0, ..., EPI_UP> (fp32 MFMA).
 * "gemm_pair_images" (ABI 7; experimental builds only -- the default library accepts 0 and refuses 1): a matrix whose only reader is a row GEMM
 * on the "gemm_f16x3" arithmetic written by its producer (the epilogue of the row GEMM in front of it) as the two fp16 parts the reader
 * multiplies -- four consecutive elements as h0 h1 h2 h3 l0 l1 l2 l3 in the 16 bytes of their fp32 values, one power-of-two exponent per
 * (row, column tile of the producer) in a table beside it -- so that no column tile of the reader splits its rows again (the bottleneck
 * activations of every TDF block, the hidden activations of the Roformer feed-forward).  Measured round 6: the reading launch alone 3-17 %
 * faster, the nets unchanged (profiles/NOTES.md); default 0.
 * "gemm_bf16x6" (per engine since ABI 6 -- it was process-wide; also ASX_GEMM_BF16X6 in the environment): 1 (default) = every row GEMM whose shape allows it
 * (K % 32 == 0, K >= 64, N > 64, N % 8 == 0, 16-byte aligned rows) runs csrc/kernels_gemm3.h -- both fp32 operands split EXACTLY into three
 * bf16 parts, six bf16 MFMA products with fp32 accumulation, the dropped cross terms below 2^-24 of a product: fp32-grade results
 * (closer to a float64 GEMM than the fp32-MFMA kernel on every measured shape) at 1.7-1.9x its speed; the same switch covers the
 * GATHER mode of that kernel (stride-1 / strided convolutions of the channels-last VR and Demucs nets with Cin % 32 == 0 and at least
 * 48 output channels) and the attention kernels of the Roformer / HTDemucs transformers (attention6_kernel, mha6_kernel);
 * 0 = the fp32-MFMA kernels (csrc/kernels_gemm2.h, kernels_halo.h, kernels_ht.h, kernels_rof.h) everywhere.
 * "gemm_f16x3" (also ASX_GEMM_F16X3): which split the row GEMMs and GATHER-mode convolutions above use while "gemm_bf16x6" is on.
 * 1 (default) = fp16 x 3: every row of x scaled by its own running power of two, every group of four output columns of W by one chosen at load, both
 * split into TWO fp16 parts (11 + 11 significand bits), three fp16 MFMA products per multiply-add instead of six, the dropped term
 * below 2^-22 of a product; elements more than 2^14 below their row's (tile's) largest keep an absolute error of 2^-37 of that
 * largest instead of a relative one.  Measured against a float64 GEMM: closer than the bf16 x 6 form on every shape (fewer accumulator
 * roundings), 1.1-1.5x its speed (profiles/r05_gemm_f16x3.txt).  The attention kernels follow the same switch (one exponent per query,
 * per 64-key tile of K, a running one per tile of V, none for the probabilities; 1.16-1.31x, profiles/r05_attention_f16x3.txt).
 * conv_wino6_kernel too (U scaled per (position, output channel) at load, V by one running exponent per (wave, tile row);
 * 1.04-1.11x, profiles/r05_wino6_f16x3.txt).  0 = bf16 x 6 (exact three-way split) in all of them.
 * The split weight images these kernels read are built on the FIRST forward after a load (one small kernel + one stream
 * synchronise per weight tensor) and belong to the engine: they are freed only when this engine's weights are re-loaded or the engine
 * is destroyed, never by another engine of the process -- so a hipGraph captured after one warm-up call stays valid while other
 * engines load and unload.  Non-finite inputs are outside the parity domain of either setting and behave differently: an Inf / NaN
 * in a GEMM row makes that row's accumulators NaN under 1 (h = Inf, v - h = NaN) and +-Inf / NaN under 0; the ReLU epilogues then
 * return 0 for NaN (maxNum).  No other row is affected (tests/test_gpu_parity.py::test_rowgemm_bf16x6_nonfinite_rows). */
int asx_set_option(asx_engine *e, const char *key, int32_t value);

/* ---- profiling ---------------------------------------------------------- */
int asx_profile_enable(asx_engine *e, int32_t on); /* clears the counters */
int asx_profile_read(asx_engine *e, asx_profile *out); /* synchronises the device */
/* The launches behind those sums, one record each (class, hipEvent milliseconds, algorithmic flops and bytes), in launch order:
 * a profile class can mix MFMA-bound and HBM-bound launches (the Demucs "conv" class does), and a roofline is a per-launch
 * statement.  Writes min(*n, cap) records, *n = launches recorded.  Synchronises the device.  (ABI 4) */
/* ABI 7: bits 8..15 of `cls` carry the number of 16-bit MFMA products per multiply-add the launch's kernel executed -- 6 (bf16 x 6: exactly
 * split operands), 3 (fp16 x 3) or 0 (an fp32-MFMA / VALU kernel); the class is `cls & 0xff`.  A roofline fraction of a class that mixes
 * pipes is executed work over the peak of the pipe that executed it, launch by launch. */
typedef struct asx_launch_rec {
  int32_t cls;
  float ms;
  double flops, bytes;
} asx_launch_rec;
int asx_profile_launches(asx_engine *e, asx_launch_rec *out, int32_t cap, int32_t *n);

#ifdef __cplusplus
}
#endif
#endif /* ASX_H */
