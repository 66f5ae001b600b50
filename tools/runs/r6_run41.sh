#!/bin/bash
# timing probe: the down / up convs with HALF of their fp32 MFMAs (library built with -DASX_ABL_UPDOWN, wrong results) against the shipped library:
# how much of their time is the matrix pipe?
set -u
O=$GRAFT_REPO_ROOT/gpurun_out/r6g
mkdir -p $O
cd $GRAFT_REPO_ROOT
run() {
  timeout 600 python bench.py --steps 3 --warmup 1 --no-arith-ab --siblings 0 --file-level 0 --cpu-seconds 0 --traffic stored 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', d['value'], d['ms_per_step'], d['kernel_ms'])"
}
run shipped
cp python-audio-separator_amd/libasx.so /tmp/libasx_keep.so
cp tools/experimental/libasx_abl.so python-audio-separator_amd/libasx.so
run half_mfma
cp /tmp/libasx_keep.so python-audio-separator_amd/libasx.so
run shipped
