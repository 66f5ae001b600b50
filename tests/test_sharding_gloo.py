"""The N>1 path on CPU: world_size-2 gloo processes drive the SAME sharding code
that runs over RCCL on the GPUs, with a test-only engine adapter that computes
chunks with the CPU oracle.  The gathered + folded result must equal the
single-process demix."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import mdx_oracle as O
from audio_separator_amd.sharding import partition_chunks, sharded_demix

P = O.MDXParams(n_fft=96, hop_length=16, dim_f=32, segment_size=16, overlap=0.25)
DIMS = O.NetDims(dim_c=4, dim_f=32, dim_t=16, g=8, l=2, num_blocks=5, k=3, bn=4)


class OracleAdapter:
    """Test double for HipEngineAdapter (tests only; the product binds libasx.so)."""

    def __init__(self, match=False):
        self.run = O.make_model_run(O.make_convtdf_state(DIMS, seed=3), DIMS)
        self.match = match

    def plan(self, n):
        cs, gen, pad, L, step, starts, _ = O.chunk_plan(n, P, self.match)
        return {"chunk_size": cs, "n_chunks": len(starts), "step": step, "padded_len": L}

    def demix_chunks(self, mix, n, k0, k1, out):
        out.copy_(torch.from_numpy(O.demix_chunks(mix.numpy(), P, self.run, k0, k1, self.match)))

    def finalize(self, chunks, n, out):
        out.copy_(torch.from_numpy(np.ascontiguousarray(O.fold_chunks(chunks.numpy(), n, P, self.match))))


def test_partition():
    assert partition_chunks(55, 8) == [(0, 7), (7, 14), (14, 21), (21, 28), (28, 35), (35, 42), (42, 49), (49, 55)]
    assert partition_chunks(3, 4) == [(0, 1), (1, 2), (2, 3), (3, 3)]
    for n in (1, 7, 55, 64):
        for w in (1, 2, 3, 8):
            r = partition_chunks(n, w)
            assert r[0][0] == 0 and r[-1][1] == n and all(a[1] == b[0] for a, b in zip(r, r[1:]))
            assert max(b - a for a, b in r) - min(b - a for a, b in r) <= 1


def test_fold_of_chunks_equals_demix():
    mix = (0.4 * np.random.default_rng(5).standard_normal((2, 3100))).astype(np.float32)
    run = O.make_model_run(O.make_convtdf_state(DIMS, seed=3), DIMS)
    nk = len(O.chunk_plan(3100, P)[5])
    a = O.fold_chunks(O.demix_chunks(mix, P, run, 0, nk), 3100, P)
    b = O.demix(mix, P, run)
    assert np.array_equal(a, b)


def test_single_process_path():
    mix = torch.from_numpy((0.4 * np.random.default_rng(6).standard_normal((2, 2500))).astype(np.float32))
    out = sharded_demix(OracleAdapter(), mix)
    ref = O.demix(mix.numpy(), P, O.make_model_run(O.make_convtdf_state(DIMS, seed=3), DIMS))
    assert np.array_equal(out.numpy(), ref)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, n, match, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mix = torch.from_numpy((0.4 * np.random.default_rng(7).standard_normal((2, n))).astype(np.float32))
    out = sharded_demix(OracleAdapter(match), mix)
    if rank == 0:
        q.put(out.numpy())
    else:
        assert out is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,n,match", [(2, 3000, False), (2, 150, False), (3, 2000, True)])
def test_sharded_demix_gloo(world, n, match):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, match, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    mix = (0.4 * np.random.default_rng(7).standard_normal((2, n))).astype(np.float32)
    run = None if match else O.make_model_run(O.make_convtdf_state(DIMS, seed=3), DIMS)
    ref = O.demix(mix, P, run, is_match_mix=match)
    assert np.array_equal(got, ref)
