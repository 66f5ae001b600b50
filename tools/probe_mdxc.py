"""On-GPU perf probe of the MDXC path on the public MDX23C layout (synthetic weights)."""
import json
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from oracle import mdxc_oracle as M
import audio_separator_amd as A

secs = float(sys.argv[1]) if len(sys.argv) > 1 else 30.0
overlap = int(sys.argv[2]) if len(sys.argv) > 2 else 8
mb = int(sys.argv[3]) if len(sys.argv) > 3 else 8
cfg = M.V3Config()          # n_fft 8192, hop 1024, dim_f 4096, dim_t 256, 4 subbands, 5 scales, c=g=128, InstanceNorm, gelu
t0 = time.time()
sd = M.make_v3_state(cfg, 0)
print("weights", sum(v.numel() for v in sd.values()) / 1e6, "M params", time.time() - t0, "s")
dm = A.MDXCDemixer({"model_data": cfg.as_model_data(), "torch_device": 0}, {"overlap": overlap}, state_dict=sd, max_batch=mb)
eng = dm.engine
N = int(44100 * secs)
mix = torch.tensor((0.3 * np.random.default_rng(0).standard_normal((2, N))).astype(np.float32)).cuda()
out = torch.empty((2, 2, N), dtype=torch.float32, device="cuda")
s = torch.cuda.current_stream().cuda_stream
plan = eng.mdxc_plan(N, overlap)
print("plan", plan, "GFLOP/chunk", eng.v3_flops(1) / 1e9)
eng.mdxc_demix_dev(mix.data_ptr(), N, overlap, out.data_ptr(), stream=s)
torch.cuda.synchronize()
t0 = time.time()
eng.mdxc_demix_dev(mix.data_ptr(), N, overlap, out.data_ptr(), stream=s)
torch.cuda.synchronize()
dt = time.time() - t0
print(f"audio {secs}s chunks {plan['n_chunks']} wall {dt*1e3:.1f} ms RTF {secs/dt:.1f} net TF/s {eng.v3_flops(plan['n_chunks'])/dt/1e12:.1f} finite {bool(torch.isfinite(out).all())}")
eng.profile_enable(True)
eng.mdxc_demix_dev(mix.data_ptr(), N, overlap, out.data_ptr(), stream=s)
prof = eng.profile_read()
for k, v in prof.items():
    if v["launches"]:
        tf = v["flops"] / (v["ms"] * 1e-3) / 1e12 if v["ms"] > 0 else 0
        gb = v["bytes"] / (v["ms"] * 1e-3) / 1e9 if v["ms"] > 0 else 0
        print(f"{k:9s} launches {v['launches']:5d}  ms {v['ms']:9.2f}  TF/s {tf:7.1f}  GB/s {gb:8.1f}")
