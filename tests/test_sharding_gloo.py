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
    local_fold = True

    def __init__(self, match=False):
        self.run = O.make_model_run(O.make_convtdf_state(DIMS, seed=3), DIMS)
        self.match = match

    def plan(self, n):
        cs, gen, pad, L, step, starts, _ = O.chunk_plan(n, P, self.match)
        return {"chunk_size": cs, "n_chunks": len(starts), "step": step, "padded_len": L, "trim": P.trim}

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


def _worker(rank, world, port, n, match, q, fold="auto"):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mix = torch.from_numpy((0.4 * np.random.default_rng(7).standard_normal((2, n))).astype(np.float32))
    out = sharded_demix(OracleAdapter(match), mix, fold=fold)
    if rank == 0:
        q.put(out.numpy())
    else:
        assert out is None
    dist.barrier()
    dist.destroy_process_group()


def test_local_fold_ownership():
    """fold="local": the ranks' sample ranges tile [0, N), and every chunk that covers a rank's range is its own or one of the
    `halo_chunks` before its first (HQ_3 geometry: 55 chunks over 8 ranks = 7 x 7 + 6, one halo chunk, 65,280 overlapping samples)"""
    from audio_separator_amd.sharding import halo_chunks, owned_samples
    p = O.MDXParams()
    n = 10_584_000
    cs, gen, pad, L, step, starts, _ = O.chunk_plan(n, p, False)
    plan = {"chunk_size": cs, "n_chunks": len(starts), "step": step, "padded_len": L, "trim": p.trim}
    assert halo_chunks(plan) == 1 and cs - step == 65280
    for world in (1, 2, 4, 8, 55, 64):
        ranges = partition_chunks(len(starts), world)
        own = owned_samples(plan, ranges, n)
        live = [(a, b) for a, b in own if b > a]
        assert live[0][0] == 0 and live[-1][1] == n and all(x[1] == y[0] for x, y in zip(live, live[1:]))
        for (k0, k1), (j0, j1) in zip(ranges, own):
            if j1 <= j0:
                continue
            for j in (j0, j1 - 1):
                pos = j + p.trim
                cover = [k for k, st in enumerate(starts) if st <= pos < st + cs]
                assert min(cover) >= k0 - 1 and max(cover) < k1


@pytest.mark.parametrize("world,n,match,fold", [(2, 3000, False, "auto"), (2, 150, False, "auto"), (3, 2000, True, "auto"),
                                                (2, 3000, False, "dst"), (3, 2000, True, "dst"),
                                                (4, 9700, False, "local"), (8, 9700, False, "local"), (8, 9700, False, "dst"),
                                                (8, 700, False, "local")])
def test_sharded_demix_gloo(world, n, match, fold):
    """world 2 / 3 / 4 / 8, both exchange schemes; n = 9700 is 55 chunks on this geometry (uneven ranges: 7 x 7 + 6 at world 8,
    14 + 14 + 14 + 13 at world 4), n = 700 fewer chunks than ranks (empty ranges at the end)"""
    if n == 9700:
        assert len(O.chunk_plan(n, P, match)[5]) == 55
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, match, q, fold)) for r in range(world)]
    for p in procs:
        p.start()
    got = q.get(timeout=300)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    mix = (0.4 * np.random.default_rng(7).standard_normal((2, n))).astype(np.float32)
    run = None if match else O.make_model_run(O.make_convtdf_state(DIMS, seed=3), DIMS)
    ref = O.demix(mix, P, run, is_match_mix=match)
    assert np.array_equal(got, ref)


def _worker_ws(rank, world, port, lengths, q, fold="dst", poison=False):
    """Three songs through ONE ShardWorkspace (bench.py --mode chunks): buffers are allocated once per shape and reused, equal
    chunk ranges take the no-compaction path (the gathered slab IS the chunk list), unequal ones the copy path."""
    from audio_separator_amd.sharding import ShardWorkspace
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ws = ShardWorkspace(poison=poison)
    ad = OracleAdapter()
    outs, ptrs = [], []
    for i, n in enumerate(lengths):
        mix = torch.from_numpy((0.4 * np.random.default_rng(20 + i).standard_normal((2, n))).astype(np.float32))
        out = sharded_demix(ad, mix, workspace=ws, fold=fold)
        if rank == 0:
            outs.append(out.numpy().copy())
            ptrs.append(ws.bufs["local" if fold == "dst" else "allc"].data_ptr())
        else:
            assert out is None
    if rank == 0:
        q.put((outs, ptrs))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("fold", ["dst", "local"])
def test_sharded_demix_with_workspace_gloo(fold):
    lengths = [3000, 3000, 2100]          # same shape twice (buffers reused), then another plan (re-allocated once)
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_ws, args=(r, world, port, lengths, q, fold)) for r in range(world)]
    for p in procs:
        p.start()
    outs, ptrs = q.get(timeout=180)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    run = O.make_model_run(O.make_convtdf_state(DIMS, seed=3), DIMS)
    even = []
    for i, n in enumerate(lengths):
        mix = (0.4 * np.random.default_rng(20 + i).standard_normal((2, n))).astype(np.float32)
        assert np.array_equal(outs[i], O.demix(mix, P, run))
        nk = len(O.chunk_plan(n, P)[5])
        even.append(nk % world == 0)
    assert ptrs[0] == ptrs[1]                     # the second song of the same length allocated nothing
    assert True in even and False in even, even   # both the slab-is-the-list path and the compaction path ran


def test_local_fold_ignores_foreign_chunks():
    """The local-fold scheme hands the fold a chunk list in which only the rank's own chunks and its seam halo are valid.  With every
    other slot poisoned with NaN (ShardWorkspace(poison=True)) and the workspace reused across two DIFFERENT songs of one length
    and a third of another, the gathered result must still be bit-identical to one process: the fold of an owned sample range
    reads nothing but its covering chunks (the invariant was implicit)."""
    lengths = [3000, 3000, 2100]
    world = 3
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_ws, args=(r, world, port, lengths, q, "local", True)) for r in range(world)]
    for p in procs:
        p.start()
    outs, _ = q.get(timeout=180)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    run = O.make_model_run(O.make_convtdf_state(DIMS, seed=3), DIMS)
    for i, n in enumerate(lengths):
        mix = (0.4 * np.random.default_rng(20 + i).standard_normal((2, n))).astype(np.float32)
        assert np.isfinite(outs[i]).all()
        assert np.array_equal(outs[i], O.demix(mix, P, run))


# ---- sibling loops: Roformer chunks and Demucs segment-forwards through the same driver ------------------------------
from fractions import Fraction  # noqa: E402

from oracle import demucs_oracle as D  # noqa: E402
from oracle import hdemucs_oracle as H  # noqa: E402
from oracle import roformer_oracle as R  # noqa: E402

RCFG = R.RoformerConfig(dim=32, depth=1, heads=2, dim_head=64, freqs_per_bands=(2, 2, 4, 8, 17), stft_n_fft=64, stft_hop_length=16,
                        stft_win_length=64, dim_t=21, sample_rate=100, mlp_expansion_factor=2)
HCFG = H.HDConfig(channels=8, nfft=1024, depth=5, norm_starts=3, dconv_attn=3, dconv_lstm=3, samplerate=8000, segment=2)
DCFG = D.HTConfig(channels=16, nfft=1024, depth=3, bottom_channels=128, t_layers=1, t_heads=2, samplerate=8000, segment=Fraction(1, 1))


class RoformerOracleAdapter:
    stems, out_stems = 1, 2

    def __init__(self, overlap=2):
        self.sd = R.make_roformer_state(RCFG, 7)
        self.ov = overlap

    def plan(self, n):
        cs, step, starts = R.roformer_plan(n, RCFG, self.ov)
        return {"chunk_size": cs, "n_chunks": len(starts)}

    def demix_chunks(self, mix, n, k0, k1, out):
        out.copy_(torch.from_numpy(R.roformer_chunks(mix.numpy(), self.sd, RCFG, self.ov, k0, k1)))

    def finalize(self, chunks, n, out):
        out.copy_(torch.from_numpy(np.ascontiguousarray(R.roformer_fold(chunks.numpy(), n, RCFG, self.ov))))


class DemucsOracleAdapter:
    stems = out_stems = 4

    def __init__(self, shifts=1, offsets=(1234,), overlap=0.25):
        self.sd = D.make_ht_state(DCFG, 11)
        self.kw = dict(shifts=shifts, offsets=list(offsets), overlap=overlap)

    def bind_mix(self, mix):
        self.mix = mix.numpy()

    def plan(self, n):
        plan, _, _ = D.segment_plan(n, DCFG, **self.kw)
        return {"chunk_size": DCFG.training_length, "n_chunks": len(plan)}

    def demix_chunks(self, mix, n, k0, k1, out):
        out.copy_(torch.from_numpy(D.demucs_segments(mix.numpy(), self.sd, DCFG, k0=k0, k1=k1, **self.kw)))

    def finalize(self, chunks, n, out):
        out.copy_(torch.from_numpy(np.ascontiguousarray(D.demucs_fold(self.mix, chunks.numpy(), DCFG, **self.kw))))


class HDemucsOracleAdapter(DemucsOracleAdapter):
    """Demucs v3: chunks at their own length (rows of the slab hold them from column 0)"""

    def __init__(self, shifts=1, offsets=(1234,), overlap=0.25):
        self.sd = H.make_hd_state(HCFG, 21)
        self.kw = dict(shifts=shifts, offsets=list(offsets), overlap=overlap)

    def plan(self, n):
        plan, _, _, seg = H.hd_segment_plan(n, HCFG, **self.kw)
        return {"chunk_size": seg, "n_chunks": len(plan)}

    def demix_chunks(self, mix, n, k0, k1, out):
        out.copy_(torch.from_numpy(H.hd_segments(mix.numpy(), self.sd, HCFG, k0=k0, k1=k1, **self.kw)))

    def finalize(self, chunks, n, out):
        out.copy_(torch.from_numpy(np.ascontiguousarray(H.hd_fold(self.mix, chunks.numpy(), HCFG, **self.kw))))


def test_segment_form_equals_demix():
    """the chunk-list restatements used by the adapters reproduce the oracle's monolithic loops"""
    mix = (0.4 * np.random.default_rng(15).standard_normal((2, 1000))).astype(np.float32)
    a = RoformerOracleAdapter()
    nk = a.plan(1000)["n_chunks"]
    got = R.roformer_fold(R.roformer_chunks(mix, a.sd, RCFG, 2, 0, nk), 1000, RCFG, 2)
    assert np.array_equal(got, R.roformer_demix(mix, a.sd, RCFG, overlap=2))
    mixd = (0.3 * np.random.default_rng(16).standard_normal((2, 13000)) + 0.02).astype(np.float32)
    d = DemucsOracleAdapter(shifts=2, offsets=(1234, 77))
    nseg = d.plan(13000)["n_chunks"]
    got = D.demucs_fold(mixd, D.demucs_segments(mixd, d.sd, DCFG, k0=0, k1=nseg, **d.kw), DCFG, **d.kw)
    want = D.demix_demucs(mixd, d.sd, DCFG, shifts=2, overlap=0.25, offsets=[1234, 77])
    assert np.abs(got - want).max() / np.abs(want).max() < 1e-6
    mixh = (0.3 * np.random.default_rng(19).standard_normal((2, 40011)) - 0.01).astype(np.float32)
    hd = HDemucsOracleAdapter(shifts=2, offsets=(1234, 77))
    nseg = hd.plan(40011)["n_chunks"]
    got = H.hd_fold(mixh, H.hd_segments(mixh, hd.sd, HCFG, k0=0, k1=nseg, **hd.kw), HCFG, **hd.kw)
    want = H.demix_hdemucs(mixh, hd.sd, HCFG, shifts=2, overlap=0.25, offsets=[1234, 77])
    assert np.abs(got - want).max() / np.abs(want).max() < 1e-6


def _sib_worker(rank, world, port, kind, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    if kind == "roformer":
        mix = torch.from_numpy((0.4 * np.random.default_rng(17).standard_normal((2, 1400))).astype(np.float32))
        out = sharded_demix(RoformerOracleAdapter(), mix)
    elif kind == "hdemucs":
        mix = torch.from_numpy((0.3 * np.random.default_rng(20).standard_normal((2, 50000))).astype(np.float32))
        out = sharded_demix(HDemucsOracleAdapter(), mix)
    else:
        mix = torch.from_numpy((0.3 * np.random.default_rng(18).standard_normal((2, 21000))).astype(np.float32))
        out = sharded_demix(DemucsOracleAdapter(), mix)
    if rank == 0:
        q.put(out.numpy())
    else:
        assert out is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("kind", ["roformer", "demucs", "hdemucs"])
def test_sharded_siblings_gloo(kind):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_sib_worker, args=(r, 2, port, kind, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get(timeout=300)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    if kind == "roformer":
        mix = (0.4 * np.random.default_rng(17).standard_normal((2, 1400))).astype(np.float32)
        assert np.array_equal(got, R.roformer_demix(mix, R.make_roformer_state(RCFG, 7), RCFG, overlap=2))
    elif kind == "hdemucs":
        mix = (0.3 * np.random.default_rng(20).standard_normal((2, 50000))).astype(np.float32)
        want = H.demix_hdemucs(mix, H.make_hd_state(HCFG, 21), HCFG, shifts=1, overlap=0.25, offsets=[1234])
        assert np.abs(got - want).max() / np.abs(want).max() < 1e-6
    else:
        mix = (0.3 * np.random.default_rng(18).standard_normal((2, 21000))).astype(np.float32)
        want = D.demix_demucs(mix, D.make_ht_state(DCFG, 11), DCFG, shifts=1, overlap=0.25, offsets=[1234])
        assert np.abs(got - want).max() / np.abs(want).max() < 1e-6
