#!/bin/bash
# round 6, run 26: shard-time model with this round's kernels (one rank's chunk ranges at G = 1 / 2 / 4 / 8 on one GPU) + conv3h op tests after the band-choice change
mkdir -p gpurun_out/r6h
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "conv3x3_direct or winograd_hq3_excerpt" 2>&1 | tail -2
timeout 900 python -m pytest tests/test_gpu_sharding.py tests/test_gpu_fullsong.py -x -q -k "graph or mdx_hq3" 2>&1 | tail -2
timeout 1200 python tools/probe_shard_model.py --reps 3 > gpurun_out/r6h/scale_model.json 2> gpurun_out/r6h/scale_model.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r6h/scale_model.json"))
print(d["one_gpu"])
for g in ("2", "4", "8"):
    s = d["strong"][g]
    print(g, s["compute_ms"], s.get("per_chunk_efficiency_vs_55"), s["predicted_ms"], s["predicted_rtf"], s.get("predicted_speedup"))
print({k: v for k, v in d.get("weak", {}).items()} if isinstance(d.get("weak"), dict) else None)
PY
