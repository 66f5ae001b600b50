#!/usr/bin/env python3
"""Is the bf16 x 6 leg of the BS-Roformer forward deterministic?  Repeats rof_forward on one chunk and compares runs bit for bit and
against the fp32-MFMA leg.  Environment switches (read once per process by the library) select which bf16 x 6 kernels take part:
ASX_ATTN6=0 keeps attention on the fp32 pipe.  usage: debug_rof_race.py [depth] [repeats] [6first]"""
import sys

import numpy as np

sys.path.insert(0, ".")
from oracle import roformer_oracle as R
import audio_separator_amd as A

depth = int(sys.argv[1]) if len(sys.argv) > 1 else 12
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
cfg = R.RoformerConfig(depth=depth, freqs_per_bands=R.DEFAULT_FREQS_PER_BANDS)
sd = R.make_roformer_state(cfg, 0)
dm = A.MDXCDemixer({"model_data": cfg.as_model_data(), "torch_device": 0, "secondary_stem_name": "other"}, {"overlap": 8}, state_dict=sd, max_batch=1)
C = cfg.stft_hop_length * (cfg.dim_t - 1)
x = (0.3 * np.random.default_rng(3).standard_normal((1, 2, C))).astype(np.float32)
eng = dm.engine


def rel(a, b):
    a, b = a.astype(np.float64), b.astype(np.float64)
    return float(np.sqrt(np.mean((a - b) ** 2)) / np.sqrt(np.mean(b ** 2)))


first6 = len(sys.argv) > 3 and sys.argv[3] == "6first"     # the bf16 x 6 leg is the FIRST forward after the load (lazy split images, fresh workspace)
runs = []
if first6:
    eng.set_option("gemm_bf16x6", 1)
    runs = [eng.rof_forward(x) for _ in range(reps)]
eng.set_option("gemm_bf16x6", 0)
y32 = eng.rof_forward(x)
y32b = eng.rof_forward(x)
print("fp32 leg deterministic:", np.array_equal(y32, y32b, equal_nan=True), "finite", np.isfinite(y32).all())
eng.set_option("gemm_bf16x6", 1)
if not first6:
    runs = [eng.rof_forward(x) for _ in range(reps)]
for i, y in enumerate(runs):
    print(f"bf16x6 run {i}: vs fp32 {rel(y, y32):.3e}  equal to run 0: {np.array_equal(y, runs[0])}  finite {np.isfinite(y).all()}")
