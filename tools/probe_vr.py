"""On-GPU perf probe of the VR path on the 4band_44100 layout with an HP-size CascadedASPPNet (synthetic weights)."""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from oracle import vr_oracle as V
import audio_separator_amd as A

secs = float(sys.argv[1]) if len(sys.argv) > 1 else 10.0
mb = int(sys.argv[2]) if len(sys.argv) > 2 else 4
arch = int(sys.argv[3]) if len(sys.argv) > 3 else 123821
# uvr_lib_v5/vr_network/modelparams/4band_44100.json
MP = {"bins": 768, "unstable_bins": 7, "reduction_bins": 668, "sr": 44100, "pre_filter_start": 740, "pre_filter_stop": 768,
      "band": {1: {"sr": 11025, "hl": 128, "n_fft": 1024, "crop_start": 0, "crop_stop": 186, "lpf_start": 37, "lpf_stop": 73, "res_type": "polyphase"},
               2: {"sr": 11025, "hl": 128, "n_fft": 512, "crop_start": 4, "crop_stop": 185, "hpf_start": 36, "hpf_stop": 18, "lpf_start": 93, "lpf_stop": 185, "res_type": "polyphase"},
               3: {"sr": 22050, "hl": 256, "n_fft": 512, "crop_start": 46, "crop_stop": 186, "hpf_start": 93, "hpf_stop": 46, "lpf_start": 164, "lpf_stop": 186, "res_type": "polyphase"},
               4: {"sr": 44100, "hl": 512, "n_fft": 768, "crop_start": 121, "crop_stop": 382, "hpf_start": 138, "hpf_stop": 123, "res_type": "sinc_medium"}}}
t0 = time.time()
sd = V.make_vr_state(arch, 0)
print("weights", sum(v.numel() for v in sd.values()) / 1e6, "M params", round(time.time() - t0, 1), "s")
dm = A.VRDemixer({"model_params": MP, "primary_stem_name": "Instrumental", "torch_device": 0},
                 {"window_size": 512, "batch_size": mb, "aggression": 5}, state_dict=sd, nn_arch_size=arch, max_batch=mb)
eng = dm.engine
n = int(44100 * secs)
rng = np.random.default_rng(0)
wave = (0.3 * rng.standard_normal((2, n))).astype(np.float32)
T, n_out = eng.vr_plan(n)
patches = T // 256 + 1
print("frames", T, "patches", patches, "GFLOP/patch", eng.vr_flops() / 1e9)
p, s = dm.separate_stems(wave)
t0 = time.time()
p, s = dm.separate_stems(wave)
dt = time.time() - t0
print(f"audio {secs}s wall {dt*1e3:.1f} ms (host buffers) RTF {secs/dt:.1f} net TF/s {eng.vr_flops()*patches/dt/1e12:.1f} finite {bool(np.isfinite(p).all() and np.isfinite(s).all())}")
eng.profile_enable(True)
p, s = dm.separate_stems(wave)
prof = eng.profile_read()
tot = sum(v["ms"] for v in prof.values())
names = {"conv3x3": "gg-conv", "down": "gg-strided"}
for k, v in prof.items():
    if v["launches"]:
        tf = v["flops"] / (v["ms"] * 1e-3) / 1e12 if v["ms"] > 0 else 0
        gb = v["bytes"] / (v["ms"] * 1e-3) / 1e9 if v["ms"] > 0 else 0
        print(f"{names.get(k, k):11s} launches {v['launches']:5d}  ms {v['ms']:9.2f}  TF/s {tf:7.1f}  GB/s {gb:8.1f}")
print("kernel ms total", round(tot, 2), "-> RTF on kernels", round(secs / (tot * 1e-3), 1))
