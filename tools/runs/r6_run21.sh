#!/bin/bash
# round 6, run 21: the rotary-epilogue anomaly of round 5 -- packed / scalar rotary x inline-asm / builtin ldexp of the accumulators, whole-output
# determinism of the BS-Roformer qkv projection on the harness (five runs each)
mkdir -p gpurun_out/r6c
cd $GRAFT_REPO_ROOT
for n in packed_asm packed_builtin scalar_asm scalar_builtin; do
  echo "== $n"
  for i in 1 2 3; do timeout 300 tools/experimental/proto_gemm3_$n 0 12 15 0 1 1 0 2>&1 | grep 'full compare' | cut -c1-170; done
done > gpurun_out/r6c/rotary_ab.txt 2>&1
cat gpurun_out/r6c/rotary_ab.txt
