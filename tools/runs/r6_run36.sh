#!/bin/bash
# conv_wino6_kernel<H>: accumulator rescale behind the wave-uniform exponent test (new) against the unconditional multiply (old); levels 2-5, checks on the small shapes
set -u
O=$GRAFT_REPO_ROOT/gpurun_out/r6f
mkdir -p $O
cd $GRAFT_REPO_ROOT
for v in new old new old; do
  echo "== $v"
  timeout 300 tools/experimental/proto_wino6_$v 0 10 13 0 256 1 1 2>&1 | grep -v "amdgpu.ids"
done | tee $O/wino6_rescale_ab.txt
echo "== checks (new)" | tee -a $O/wino6_rescale_ab.txt
timeout 300 tools/experimental/proto_wino6_new 0 0 7 0 256 1 1 2>&1 | grep -v "amdgpu.ids" | tee -a $O/wino6_rescale_ab.txt
