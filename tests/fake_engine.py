"""TEST DOUBLE of audio_separator_amd.engine.Engine backed by the CPU oracles (oracle/*.py).

Test infrastructure only: it lets the `-m "not gpu"` suite drive the plugin-surface classes (architectures/*.py,
common_separator.py, the model-file readers, the configuration mirrors) end to end in this GPU-less container and
compare what they hand to the writer with the reference's own ``separate()`` goldens.  The product never imports this
module; on a GPU box the same tests run against libasx.so (tests/test_gpu_separate.py).
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F

from oracle import demucs_oracle as D
from oracle import ensemble_oracle as EO
from oracle import hdemucs_oracle as H
from oracle import mdx_oracle as O
from oracle import mdxc_oracle as M
from oracle import roformer_oracle as R
from oracle import vr_oracle as V


def _t(x):
    return torch.as_tensor(np.asarray(x, np.float32))


class OracleEngine:
    created = 0

    def __init__(self, cfg, device: int = 0):
        OracleEngine.created += 1
        self.cfg = cfg
        self.device = device
        self._net = None

    def close(self):
        pass

    # ---- MDX ------------------------------------------------------------------------------------------
    def _params(self, compensate=1.0):
        c = self.cfg
        return O.MDXParams(n_fft=c.n_fft, hop_length=c.hop_length, dim_f=c.dim_f, segment_size=c.segment_size,
                           overlap=c.overlap, enable_denoise=bool(c.enable_denoise), compensate=compensate)

    def load_net(self, net_cfg, tensors):
        """Folded ConvTDFNet tensors (weights.fold_convtdf_state / onnx_reader) -> a torch forward (mdxnet.py:97-120)."""
        t = {k: _t(v) for k, v in tensors.items()}
        n = net_cfg.num_blocks // 2
        nc = net_cfg

        def block(x, name):
            for j in range(nc.l):
                x = F.relu(F.conv2d(x, t[f"{name}.tfc{j}.w"], t[f"{name}.tfc{j}.b"], padding=nc.k // 2))
            h = x
            for i in (0, 1):
                h = F.linear(h, t[f"{name}.tdf{i}.w"], t.get(f"{name}.tdf{i}.bias"))
                h = F.relu(h * t[f"{name}.tdf{i}.scale"][None, :, None, None] + t[f"{name}.tdf{i}.shift"][None, :, None, None])
            return x + h

        def forward(spec):
            x = _t(spec)
            x = F.relu(F.conv2d(x, t["first.w"][:, :, None, None], t["first.b"])).transpose(-1, -2)
            skips = []
            for i in range(n):
                x = block(x, f"enc{i}")
                skips.append(x)
                x = F.relu(F.conv2d(x, t[f"ds{i}.w"], t[f"ds{i}.b"], stride=2))
            x = block(x, "mid")
            for i in range(n):
                x = F.relu(F.conv_transpose2d(x, t[f"us{i}.w"], t[f"us{i}.b"], stride=2)) * skips[-i - 1]
                x = block(x, f"dec{i}")
            x = x.transpose(-1, -2)
            return F.conv2d(x, t["final.w"][:, :, None, None], t["final.b"]).numpy()
        self._net = forward

    def net_forward(self, spec):
        return self._net(spec)

    def stft(self, wave):
        return O.stft_forward(np.asarray(wave, np.float32), self.cfg.n_fft, self.cfg.hop_length, self.cfg.dim_f)

    def istft(self, spec):
        return O.stft_inverse(np.asarray(spec, np.float32), self.cfg.n_fft, self.cfg.hop_length)

    def run_model(self, wave, is_match_mix=False):
        return O.run_model(np.asarray(wave, np.float32), self._params(), self._net, is_match_mix)

    def demix(self, mix, is_match_mix=False):
        return O.demix(mix, self._params(), self._net, is_match_mix)

    def separate(self, mix, max_peak, min_peak, compensate):
        return O.separate_stems(mix, self._params(compensate), self._net, max_peak, min_peak)

    # ---- edges ------------------------------------------------------------------------------------------
    def normalize(self, wave, max_peak=1.0, min_peak=None):
        return O.normalize(wave, max_peak, min_peak)

    def pcm16(self, stem, max_peak=1.0, min_peak=None):
        a = O.normalize(np.array(stem, np.float32, copy=True), max_peak, min_peak)
        return (a * 32767).astype(np.int16), float(np.abs(a).max())

    def invert_stem(self, mixture, stem):
        return np.asarray(EO.invert_stem(mixture, stem), np.float32)

    # ---- MDXC -------------------------------------------------------------------------------------------
    def load_v3(self, v3, state_dict):
        c = self.cfg
        self._v3 = M.V3Config(n_fft=c.n_fft, hop_length=c.hop_length, dim_f=c.dim_f, dim_t=c.segment_size,
                              num_channels=v3.num_channels, num_subbands=v3.num_subbands, num_scales=v3.num_scales,
                              num_blocks_per_scale=v3.num_blocks_per_scale, num_channels_model=v3.num_channels_model,
                              growth=v3.growth, bottleneck_factor=v3.bottleneck_factor, norm=v3.norm, act=v3.act,
                              instruments=tuple(f"s{i}" for i in range(max(v3.num_targets, 2))),
                              target_instrument="s0" if v3.num_targets == 1 else None)
        self._v3_sd = {k: _t(v) for k, v in state_dict.items()}

    def mdxc_demix(self, mix, overlap):
        out = M.mdxc_demix(mix, self._v3_sd, self._v3, int(overlap))
        return out if out.ndim == 3 else out[None]

    def load_rof(self, rc, state_dict):
        c = self.cfg
        self._rof = R.RoformerConfig(dim=rc.dim, depth=rc.depth, num_stems=rc.num_stems, time_transformer_depth=rc.time_transformer_depth,
                                     freq_transformer_depth=rc.freq_transformer_depth, freqs_per_bands=tuple(rc.freqs_per_bands),
                                     dim_head=rc.dim_head, heads=rc.heads, mlp_expansion_factor=rc.mlp_expansion_factor,
                                     mask_estimator_depth=rc.mask_estimator_depth, stft_n_fft=c.n_fft, stft_hop_length=c.hop_length,
                                     stft_win_length=c.n_fft, dim_t=c.segment_size, sample_rate=1,
                                     instruments=tuple(f"s{i}" for i in range(rc.n_out)), mel=rc.mel, band_starts=tuple(rc.band_starts))
        self._rof_sd = {k: _t(v) for k, v in state_dict.items()}

    def rof_demix(self, mix, step):
        return R.roformer_demix(mix, self._rof_sd, self._rof, overlap=int(step))   # sample_rate 1: overlap counts samples

    # ---- Demucs -----------------------------------------------------------------------------------------
    def load_ht(self, hc, state_dict, pos_tables=True):
        self._ht = D.HTConfig(sources=tuple(hc.sources), channels=hc.channels, growth=hc.growth, nfft=hc.nfft, depth=hc.depth,
                              kernel_size=hc.kernel_size, stride=hc.stride, dconv_depth=hc.dconv_depth, dconv_comp=hc.dconv_comp,
                              freq_emb=hc.freq_emb, bottom_channels=hc.bottom_channels, t_layers=hc.t_layers, t_heads=hc.t_heads,
                              t_hidden_scale=hc.t_hidden_scale, samplerate=hc.samplerate, segment=hc.segment)
        self._ht_sd = {k: torch.as_tensor(np.asarray(v.detach().cpu() if hasattr(v, "detach") else v)).float() for k, v in state_dict.items()}

    def ht_demix(self, mix, shifts=0, offsets=None, overlap=0.25, standardize=False, swap01=False):
        if standardize:
            assert swap01
            return D.demix_demucs(mix, self._ht_sd, self._ht, shifts=shifts, overlap=overlap, split=True, offsets=offsets)
        fn = lambda x: D.ht_forward(x.numpy() if hasattr(x, "numpy") else x, self._ht_sd, self._ht)  # noqa: E731
        return np.asarray(D.apply_model(fn, np.asarray(mix, np.float32)[None], self._ht, shifts=shifts, split=True, overlap=overlap,
                                        offsets=offsets)[0])

    def ht_forward(self, mix):
        return np.asarray(D.ht_forward(np.asarray(mix, np.float32), self._ht_sd, self._ht))

    def load_hd(self, hc, state_dict):
        raise NotImplementedError("the CPU double covers HTDemucs packages only")

    # ---- VR ---------------------------------------------------------------------------------------------
    def load_vr(self, model_params, arch, capacity, state_dict, window_size=512, offset=128, max_batch=0, v51=None,
                wav_resolution="polyphase"):
        assert v51 is None
        self._vr_res = wav_resolution
        self._vr = (V.ModelParams(model_params), int(arch), {k: _t(v) for k, v in state_dict.items()}, int(window_size), int(offset))
        self.vr_bins = model_params["bins"]
        self.vr_window = int(window_size)

    def vr_separate(self, wave, aggr_value, split_bin, is_non_accom=False, aggr_correction=None, enable_tta=False,
                    enable_post_process=False, post_thres=0.2, high_end_process=False):
        mp, arch, sd, win, off = self._vr
        p, s = V.vr_separate(np.asarray(wave, np.float32), sd, arch, mp, window_size=win, batch_size=2,
                             aggression=round(aggr_value * 100), is_non_accom_stem=is_non_accom, enable_tta=enable_tta,
                             enable_post_process=enable_post_process, post_process_threshold=post_thres, offset=off,
                             high_end_process=high_end_process, wav_resolution=self._vr_res)
        return np.asarray(p, np.float32).T, np.asarray(s, np.float32).T

    def resample_sinc(self, x, ratio, mono_calls=False):
        x = np.asarray(x, np.float32)
        one = x.ndim == 1
        x2 = x[None] if one else x
        y = V.src_simple_sinc_fastest(x2, float(ratio), mono=bool(mono_calls) or one)
        n_out = int(np.ceil(x2.shape[1] * float(ratio)))
        y = np.pad(y, ((0, 0), (0, n_out - y.shape[1])))
        return y[0] if one else y

    def vr_forward(self, x):
        mp, arch, sd, win, off = self._vr
        return np.asarray(V.cascaded_forward(_t(x), sd, arch, mp.param["bins"] * 2))


def install(monkeypatch):
    """Point every host module's ``Engine`` at the double."""
    import audio_separator_amd  # noqa: F401
    from audio_separator_amd import demucs, mdx, mdxc, vr
    for mod in (mdx, mdxc, demucs, vr):
        monkeypatch.setattr(mod, "Engine", OracleEngine)
    OracleEngine.created = 0
