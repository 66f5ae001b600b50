"""On-GPU perf probe of the BS-Roformer path on the public ep_317 layout (synthetic weights)."""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from oracle import roformer_oracle as R
import audio_separator_amd as A

secs = float(sys.argv[1]) if len(sys.argv) > 1 else 64.0
mb = int(sys.argv[2]) if len(sys.argv) > 2 else 8
cfg = R.RoformerConfig(freqs_per_bands=R.DEFAULT_FREQS_PER_BANDS)   # dim 512, depth 12, 8 heads, T = 801, hop 441
t0 = time.time()
sd = R.make_roformer_state(cfg, 0)
print("weights", sum(v.numel() for v in sd.values()) / 1e6, "M params", round(time.time() - t0, 1), "s")
dm = A.MDXCDemixer({"model_data": cfg.as_model_data(), "torch_device": 0, "secondary_stem_name": "other"},
                   {"overlap": 8}, state_dict=sd, max_batch=mb)
eng = dm.engine
N = int(44100 * secs)
C = 441 * 800
mix = torch.tensor((0.3 * np.random.default_rng(0).standard_normal((2, N))).astype(np.float32)).cuda()
out = torch.empty((2, 2, N), dtype=torch.float32, device="cuda")
s = torch.cuda.current_stream().cuda_stream
nchunks = len(R.roformer_plan(N, cfg, 8)[2])
print("chunks", nchunks, "GFLOP/chunk", eng.rof_flops(1) / 1e9)
eng.rof_demix_dev(mix.data_ptr(), N, C, out.data_ptr(), stream=s)
torch.cuda.synchronize()
t0 = time.time()
eng.rof_demix_dev(mix.data_ptr(), N, C, out.data_ptr(), stream=s)
torch.cuda.synchronize()
dt = time.time() - t0
print(f"audio {secs}s chunks {nchunks} wall {dt*1e3:.1f} ms RTF {secs/dt:.1f} net TF/s {eng.rof_flops(nchunks)/dt/1e12:.1f} finite {bool(torch.isfinite(out).all())}")
eng.profile_enable(True)
eng.rof_demix_dev(mix.data_ptr(), N, C, out.data_ptr(), stream=s)
prof = eng.profile_read()
names = {"tdf": "gemm", "conv1x1": "attention"}
for k, v in prof.items():
    if v["launches"]:
        tf = v["flops"] / (v["ms"] * 1e-3) / 1e12 if v["ms"] > 0 else 0
        gb = v["bytes"] / (v["ms"] * 1e-3) / 1e9 if v["ms"] > 0 else 0
        print(f"{names.get(k, k):9s} launches {v['launches']:5d}  ms {v['ms']:9.2f}  TF/s {tf:7.1f}  GB/s {gb:8.1f}")
