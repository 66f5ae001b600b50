#!/bin/bash
# the distributed code path on one GPU: torchrun world 1 (RCCL init, gather, barrier, all_reduce), both modes
set -u
O=gpurun_out/r2y
mkdir -p $O
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 3 --warmup 1 --cpu-seconds 0 --siblings 0 > $O/tr_files.json 2> $O/tr_files.err
BENCH_FORCE_DIST=1 timeout 600 python bench.py --gpus 1 --steps 3 --warmup 1 --cpu-seconds 0 --siblings 0 --songs-per-rank 2 > $O/fd_files2.json 2> $O/fd_files2.err
BENCH_FORCE_DIST=1 timeout 600 python bench.py --gpus 1 --steps 3 --warmup 1 --cpu-seconds 0 --siblings 0 --mode chunks > $O/fd_chunks.json 2> $O/fd_chunks.err
for f in tr_files fd_files2 fd_chunks; do
python - <<PY
import json
t=open('$O/$f.json').read().strip().splitlines()
print('$f', len(t), 'line(s)')
r=json.loads(t[-1]); print({k:r[k] for k in ('value','n_gpus','ms_per_step','scaling','rccl') if k in r}, r['config'].get('mode'), r['config'].get('songs_per_step'))
PY
done
tail -2 $O/fd_chunks.err
