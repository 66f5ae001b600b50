"""Quick on-GPU perf probe: HQ_3-shaped net, synthetic weights, per-kernel-class time."""
import json
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from oracle import mdx_oracle as O
import audio_separator_amd as A

secs = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
mb = int(sys.argv[2]) if len(sys.argv) > 2 else 16
d = O.NetDims()
sd = O.make_convtdf_state(d, seed=0)
N = int(44100 * secs)
mix = torch.tensor(O.synth_mix(N, seed=0)).cuda()
out = torch.empty_like(mix)
eng = A.Engine(A.MDXConfig(max_batch=mb))
import os
if os.environ.get('WINO'):
    eng.set_option('winograd', int(os.environ['WINO']))
eng.load_net(A.NetConfig(), A.fold_convtdf_state(sd, d.num_blocks, d.l))
s = torch.cuda.current_stream().cuda_stream
eng.demix_dev(mix.data_ptr(), N, out.data_ptr(), stream=s)
torch.cuda.synchronize()
t0 = time.time()
eng.demix_dev(mix.data_ptr(), N, out.data_ptr(), stream=s)
torch.cuda.synchronize()
dt = time.time() - t0
nch = eng.plan(N)["n_chunks"]
print(f"audio {secs}s chunks {nch} wall {dt*1e3:.1f} ms  RTF {secs/dt:.1f}  net TF/s {eng.net_flops(nch)/dt/1e12:.1f}")
eng.profile_enable(True)
eng.demix_dev(mix.data_ptr(), N, out.data_ptr(), stream=s)
prof = eng.profile_read()
eng.profile_enable(False)
for k, v in prof.items():
    if v["launches"]:
        tf = v["flops"] / (v["ms"] * 1e-3) / 1e12 if v["ms"] > 0 else 0
        gb = v["bytes"] / (v["ms"] * 1e-3) / 1e9 if v["ms"] > 0 else 0
        print(f"{k:9s} launches {v['launches']:5d}  ms {v['ms']:9.2f}  TF/s {tf:7.1f}  GB/s {gb:8.1f}")
print(json.dumps(prof))
