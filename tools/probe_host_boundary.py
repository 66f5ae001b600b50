"""PCIe-inclusive rate of the host-buffer entry point: asx_demix(mix_host [2, N]) -> out_host, 4-minute song, weights resident.
The bench's `value` starts with the song already in HBM; this is the same job through the boundary that hands over host memory."""
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

N = 44100 * 240
d = O.NetDims()
sd = O.make_convtdf_state(d, seed=0)
eng = A.Engine(A.MDXConfig(), device=0)
eng.load_net(A.NetConfig(), A.fold_convtdf_state(sd, d.num_blocks, d.l))
mix = O.synth_mix(N, seed=0)
eng.demix(mix)
ts = []
for _ in range(5):
    t0 = time.perf_counter()
    out = eng.demix(mix)
    ts.append(time.perf_counter() - t0)
dev = torch.from_numpy(mix).cuda()
o = torch.empty_like(dev)
s = torch.cuda.current_stream().cuda_stream
eng.demix_dev(dev.data_ptr(), N, o.data_ptr(), stream=s)
torch.cuda.synchronize()
td = []
for _ in range(5):
    t0 = time.perf_counter()
    eng.demix_dev(dev.data_ptr(), N, o.data_ptr(), stream=s)
    torch.cuda.synchronize()
    td.append(time.perf_counter() - t0)
h, g = float(np.median(ts)), float(np.median(td))
print(json.dumps({"host_buffers_ms": round(h * 1e3, 2), "host_buffers_rtf": round(240 / h, 1), "device_buffers_ms": round(g * 1e3, 2),
                  "device_buffers_rtf": round(240 / g, 1), "pcie_and_staging_ms": round((h - g) * 1e3, 2),
                  "note": "pageable numpy arrays in and out (2 x 84.7 MB each way)"}))
