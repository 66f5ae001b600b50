#!/bin/bash
# Round 5, first GPU call: the GPU suite on the per-engine split-image tree (whole-workload digests excepted: their oracle records were
# still being computed), the full-depth BS-Roformer chunk, the bf16 x 6 edge-value tests; harness A/Bs (wino6 persistent / per tile /
# per item; tdf3 tile -> XCD maps); the shard-time model probe; a short bench line with the per-level table.
set -u
O=$GRAFT_REPO_ROOT/gpurun_out/r5a
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests -m gpu -x -q --deselect tests/test_gpu_fullsong.py -s > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log
grep -n "rel-RMS\|passed\|failed\|error" $O/pytest.log | tail -30
# wino6 forms: persistent (grid 256), one workgroup per spatial tile (grid >= tiles), one workgroup per item
for cfg in "256 0" "100000000 0" "256 1"; do
  set -- $cfg
  echo "== wino6 grid=$1 one=$2" >> $O/wino6.log
  timeout 120 tools/experimental/proto_wino6 0 0 99 0 $1 $2 >> $O/wino6.log 2>&1
done
cat $O/wino6.log
for map in 0 1 2; do
  echo "== tdf3 tile_map=$map" >> $O/gemm3.log
  timeout 120 tools/proto_gemm3 0 3 10 $map >> $O/gemm3.log 2>&1
done
cat $O/gemm3.log
timeout 300 python tools/probe_shard_model.py > $O/scale_model.json 2> $O/scale_model.err; echo "shard model rc=$?"; tail -3 $O/scale_model.err
timeout 300 python bench.py --steps 10 --warmup 3 --cpu-seconds 0 --siblings 0 --file-level 0 --traffic stored > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d = json.load(open("gpurun_out/r5a/bench.json"))
print(d["value"], d["ms_per_step"], d["kernel_ms"])
print(json.dumps(d["roofline"]["per_level"], indent=None))
PY
