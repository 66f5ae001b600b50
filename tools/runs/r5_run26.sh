#!/bin/bash
# conv_wino6_kernel on the fp16 x 3 arithmetic (template parameter H) against its bf16 x 6 form and conv_wino3_kernel: harness,
# one workgroup per item; small shapes against a float64 convolution, levels 0 .. 5 of the HQ_3 net timed
set -u
O=$GRAFT_REPO_ROOT/gpurun_out/r5A
mkdir -p $O
cd $GRAFT_REPO_ROOT
( echo "== bf16 x 6"; timeout 300 tools/experimental/proto_wino6 0 0 99 0 256 1 0
  echo "== fp16 x 3"; timeout 300 tools/experimental/proto_wino6 0 0 99 0 256 1 1 ) 2>&1 | grep -v amdgpu.ids | tee $O/wino6h.txt
