#!/bin/bash
# after the rotary-epilogue fix: harness qkv shapes on the three tile forms (whole-output determinism), the BS-Roformer determinism probe,
# the whole GPU suite, the bench line
set -u
O=$GRAFT_REPO_ROOT/gpurun_out/r5t
mkdir -p $O
cd $GRAFT_REPO_ROOT
for t in 0 1 2; do echo "== f16x3 tile $t"; timeout 600 tools/proto_gemm3 0 12 15 0 1 1 $t 2>&1 | grep "full compare"; done | tee $O/qkv_rot.txt
echo "== bf16x6"; timeout 600 tools/proto_gemm3 0 12 13 0 0 1 0 2>&1 | grep "full compare" | tee -a $O/qkv_rot.txt
( echo "== f16x3"; timeout 300 python tools/debug_rof_race.py 12 4 6first; echo "== bf16x6"; ASX_GEMM_F16X3=0 timeout 300 python tools/debug_rof_race.py 12 3 6first ) 2>&1 | grep -v amdgpu.ids | tee $O/rof_race.txt
timeout 1500 python -m pytest tests -x -q -m gpu -W default 2>&1 | tail -8 | tee $O/pytest_tail.txt
timeout 400 python bench.py --steps 20 --warmup 5 > $O/bench_n1.json 2> $O/bench_n1.err; echo "bench rc=$?"
python -c "
import json; d=json.load(open('$O/bench_n1.json')); print(d['value'], d['ms_per_step'], d['kernel_ms'], {k:(v.get('value')) for k,v in d.get('siblings',{}).items()}, d['file_level']['rtf'])" | tee -a $O/pytest_tail.txt
