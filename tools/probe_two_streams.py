"""Do two independent demix passes on two HIP streams overlap usefully?  The row-GEMM (TDF) launches wait on memory at
71 % of the MFMA peak while the 3x3 convs run at 90 %: if workgroups of both kinds share the CUs, the convs of one pass can
fill the stalls of the other pass's row GEMMs.  Two engines (own weights + workspace), one song each; sequential on one
stream vs concurrent on two, with an optional start offset of the second stream."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from oracle import mdx_oracle as O
import audio_separator_amd as A

SR = 44100
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
N = int(SR * secs)
d = O.NetDims()
sd = O.make_convtdf_state(d, seed=0)
folded = A.fold_convtdf_state(sd, d.num_blocks, d.l)
engs = []
for i in range(2):
    e = A.Engine(A.MDXConfig(), device=0)
    e.load_net(A.NetConfig(), folded)
    engs.append(e)
dev = torch.device("cuda", 0)
mixes = [torch.from_numpy(O.synth_mix(N, seed=s)).to(dev) for s in range(2)]
outs = [torch.empty((2, N), dtype=torch.float32, device=dev) for _ in range(2)]
streams = [torch.cuda.Stream(device=dev) for _ in range(2)]


def run(concurrent, offset_ms=0.0, reps=4):
    torch.cuda.synchronize()
    ts = []
    for r in range(reps + 1):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(2):
            st = streams[i] if concurrent else streams[0]
            if concurrent and i == 1 and offset_ms > 0:
                with torch.cuda.stream(st):
                    torch.cuda._sleep(int(offset_ms * 1e-3 * 2.4e9))
            engs[i].demix_dev(mixes[i].data_ptr(), N, outs[i].data_ptr(), stream=st.cuda_stream)
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    return min(ts[1:]) * 1e3


res = {"seconds_per_song": secs, "sequential_ms": run(False)}
ref = [o.clone() for o in outs]
for off in (0.0, 5.0, 10.0, 20.0, 40.0):
    res[f"concurrent_offset_{off:g}ms"] = run(True, off)
res["same_result"] = bool(all(torch.equal(a, b) for a, b in zip(ref, outs)))
print(json.dumps(res))
