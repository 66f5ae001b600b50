"""On-GPU perf probe of the Demucs v3 path on the public hdemucs_mmi layout (synthetic weights, segment 44 s)."""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from oracle import hdemucs_oracle as H
import audio_separator_amd as A

secs = float(sys.argv[1]) if len(sys.argv) > 1 else 240.0
mb = int(sys.argv[2]) if len(sys.argv) > 2 else 4
shifts = int(sys.argv[3]) if len(sys.argv) > 3 else 2
oc = H.HDConfig(segment=44)   # channels 48, depth 6, nfft 4096
t0 = time.time()
sd = H.make_hd_state(oc, 0)
print("weights", sum(v.numel() for v in sd.values()) / 1e6, "M params", round(time.time() - t0, 1), "s")
hc = A.HDConfig(segment=44, max_batch=mb)
eng = A.Engine(A.MDXConfig(n_fft=4096, hop_length=1024, dim_f=2048, segment_size=8))
eng.load_hd(hc, sd)
N = int(44100 * secs)
mix = torch.tensor((0.3 * np.random.default_rng(0).standard_normal((2, N))).astype(np.float32)).cuda()
out = torch.empty((4, 2, N), dtype=torch.float32, device="cuda")
s = torch.cuda.current_stream().cuda_stream
offs = [11025, 3000, 17000, 9000][:shifts]
TL = hc.segment_samples
stride = int(0.75 * TL)
lens = []
for o in (offs if shifts else [None]):
    vl = N + 22050 - o if shifts else N
    lens += [min(vl - k, TL) for k in range(0, vl, stride)]
flops = sum(eng.hd_flops(l) for l in lens)
print("chunks", len(lens), "lengths (s)", [round(l / 44100, 1) for l in lens], "GFLOP", flops / 1e9)
eng.hd_demix_dev(mix.data_ptr(), N, out.data_ptr(), shifts=shifts, offsets=offs, flags=3, stream=s)
torch.cuda.synchronize()
t0 = time.time()
eng.hd_demix_dev(mix.data_ptr(), N, out.data_ptr(), shifts=shifts, offsets=offs, flags=3, stream=s)
torch.cuda.synchronize()
dt = time.time() - t0
print(f"audio {secs}s chunks {len(lens)} wall {dt*1e3:.1f} ms RTF {secs/dt:.1f} net TF/s {flops/dt/1e12:.1f} finite {bool(torch.isfinite(out).all())} "
      f"mem {torch.cuda.mem_get_info()[0] / 2**30:.1f} GiB free")
eng.profile_enable(True)
eng.hd_demix_dev(mix.data_ptr(), N, out.data_ptr(), shifts=shifts, offsets=offs, flags=3, stream=s)
prof = eng.profile_read()
names = {"tdf": "linear", "conv1x1": "lstm+attn", "conv3x3": "gg-conv", "down": "gg-strided", "up": "gg-convT"}
for k, v in prof.items():
    if v["launches"]:
        tf = v["flops"] / (v["ms"] * 1e-3) / 1e12 if v["ms"] > 0 else 0
        gb = v["bytes"] / (v["ms"] * 1e-3) / 1e9 if v["ms"] > 0 else 0
        print(f"{names.get(k, k):11s} launches {v['launches']:5d}  ms {v['ms']:9.2f}  TF/s {tf:7.1f}  GB/s {gb:8.1f}")
