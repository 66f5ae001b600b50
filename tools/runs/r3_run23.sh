#!/bin/bash
# round-3 GPU run 23: fused inverse STFT + overlap-add of the Demucs nets (bit-identical by construction): tests, A/B, group size
set -u
O=gpurun_out/r3w
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_demucs.py tests/test_gpu_hdemucs.py tests/test_gpu_fullsize.py tests/test_gpu_sharding.py -q -x -m gpu -k "not vr and not mdx23c and not rof" > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log
tail -3 $O/pytest.log
S="python tools/bench_siblings.py --cpu 0 --steps 2 --workloads htdemucs,hdemucs"
timeout 600 $S > $O/sib_fused.jsonl 2> $O/sib_fused.err
ASX_HT_FUSED_OLA=0 timeout 600 $S > $O/sib_two.jsonl 2> $O/sib_two.err
ASX_HT_OLA_G=32 timeout 600 $S > $O/sib_g32.jsonl 2> $O/sib_g32.err
ASX_HT_OLA_G=8 timeout 600 $S > $O/sib_g8.jsonl 2> $O/sib_g8.err
python - <<'PY'
import json,glob,os
for f in sorted(glob.glob('gpurun_out/r3w/sib_*.jsonl')):
    for l in open(f):
        r=json.loads(l); print(os.path.basename(f), r['config']['workload'][:14], r['value'], r['ms_per_step'], {k[:10]:v for k,v in r['kernel_ms'].items() if k in ('istft','ola','stft')})
PY
python - <<'PY'
# bit-identity of the fused path against the two-kernel path on one htdemucs segment batch
import os, sys, subprocess, numpy as np
code = """
import sys, numpy as np
sys.path.insert(0, '.')
from fractions import Fraction
from oracle import demucs_oracle as D
import audio_separator_amd as A
oc = D.HTConfig(); sd = D.make_ht_state(oc, 0)
eng = A.Engine(A.MDXConfig(n_fft=4096, hop_length=1024, dim_f=2048, segment_size=8)); eng.load_ht(A.HTConfig(segment=Fraction(39, 5)), sd)
x = (0.3 * np.random.default_rng(0).standard_normal((3, 2, oc.training_length))).astype(np.float32)
np.save(sys.argv[1], eng.ht_forward(x))
"""
outs = []
for flag in ("1", "0"):
    env = dict(os.environ, ASX_HT_FUSED_OLA=flag)
    path = f"/tmp/ht_fwd_{flag}.npy"
    subprocess.check_call([sys.executable, "-c", code, path], env=env)
    outs.append(np.load(path))
print("fused vs two-kernel: identical =", bool(np.array_equal(outs[0], outs[1])), "max abs diff", float(np.abs(outs[0] - outs[1]).max()))
PY
