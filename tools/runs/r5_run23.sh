#!/bin/bash
# attention6_kernel on the fp16 x 3 arithmetic (template parameter H) against its bf16 x 6 form and attention2_kernel: stand-alone harness
set -u
O=$GRAFT_REPO_ROOT/gpurun_out/r5w
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 600 tools/proto_attn6 2>&1 | grep -v "amdgpu.ids" | tee $O/attn6h.txt
