"""Multi-GPU sharding of the chunk loop (one process per GPU, torch.distributed).

The chunks of MDXSeparator.demix are independent given the replicated mix
(mdx_separator.py:348-392), so rank r runs a contiguous chunk range
on its own engine.  Two exchange schemes, both bit-identical to a one-GPU run
(the fold is a gather over covering chunks in fixed chunk order):

  * fold="local" (default for uniform-stride loops; SURVEY.md 8e's sketch): every rank folds ITS OWN contiguous sample
    range.  A chunk reaches chunk_size - step samples into the next chunk's stride, so rank r needs, besides its own
    chunks, only the last ceil(C / step) - 1 chunks of the ranks before it (one 2-MB chunk on the HQ_3 geometry, of which
    the 65,280-sample tail is what overlaps): point-to-point sends to the right neighbour, no fold and no 115-MB chunk
    gather on the destination's critical path.  The folded slabs [2, ~N / G] are then gathered to rank ``dst``.
  * fold="dst": ONE gather of all windowed chunks to rank ``dst``, which folds them (loops without a uniform chunk stride:
    the Roformer tail chunk, the Demucs shifts).

The driver is backend agnostic: it talks to an *engine adapter* with three
methods operating on torch tensors that live on the adapter's device,

    plan(n_samples) -> dict            (chunk_size, n_chunks, ...)
    demix_chunks(mix, n, k0, k1, out)  windowed chunks [k1-k0, 2, C] -> out
    finalize(chunks, n, out)           all chunks [n_chunks, 2, C] -> out [2, n]

and optionally ``stems`` (rows S of a chunk [S, 2, C]; default none) and
``out_stems`` (rows of the result [S_out, 2, n]).

``HipEngineAdapter`` binds those to libasx.so for the MDX loop; ``MdxcAdapter``,
``RoformerAdapter`` and ``DemucsAdapter`` do the same for the sibling loops
(MDXC unfold / uniform fold, Roformer Hamming fold, Demucs apply_model whose "chunks"
are the segment-forwards of every shift) -- SURVEY.md 8e.  With backend "nccl" the
gather is RCCL over xGMI; the CPU tests drive the same code over "gloo".
"""
from __future__ import annotations

from typing import List, Tuple


def partition_chunks(n_chunks: int, world: int) -> List[Tuple[int, int]]:
    """Contiguous, balanced chunk ranges; earlier ranks take the remainder."""
    base, rem = divmod(n_chunks, world)
    out, k = [], 0
    for r in range(world):
        n = base + (1 if r < rem else 0)
        out.append((k, k + n))
        k += n
    return out


def owned_samples(plan: dict, ranges, n: int):
    """Output sample ranges [j0, j1) of the ranks for the local fold: rank r owns the padded positions its chunks START in,
    [k0 step, k1 step) -- the last non-empty rank up to the padded end -- shifted by the plan's trim and clipped to [0, n)."""
    step, trim, L = int(plan["step"]), int(plan.get("trim", 0)), int(plan["padded_len"])
    last = max((r for r, (a, b) in enumerate(ranges) if b > a), default=-1)
    out = []
    for r, (a, b) in enumerate(ranges):
        if b <= a:
            out.append((0, 0))
            continue
        p0, p1 = a * step, (L if r == last else b * step)
        out.append((min(n, max(0, p0 - trim)), min(n, max(0, p1 - trim))))
    return out


def halo_chunks(plan: dict) -> int:
    """How many chunks before a rank's first one reach into its sample range: ceil(chunk_size / step) - 1."""
    C, step = int(plan["chunk_size"]), int(plan["step"])
    return -(-C // step) - 1


class HipEngineAdapter:
    """Engine adapter over libasx.so for CUDA(HIP) torch tensors."""
    # chunk k starts at k * step and asx_finalize_dev is a per-sample gather over the covering chunks with an analytic divider
    # (finalize4_kernel / finalize_div_kernel): a rank can fold its own sample range from its own chunks + the seam halo
    # (sharded_demix fold="local"; contract spelled out in _sharded_demix_local_fold)
    local_fold = True

    def __init__(self, engine, is_match_mix: bool = False):
        self.engine = engine
        self.is_match_mix = is_match_mix

    def _stream(self):
        import torch
        return torch.cuda.current_stream().cuda_stream

    def plan(self, n_samples: int) -> dict:
        return self.engine.plan(n_samples, self.is_match_mix)

    def demix_chunks(self, mix, n_samples, k0, k1, out):
        self.engine.demix_chunks_dev(mix.data_ptr(), n_samples, k0, k1, out.data_ptr(), self.is_match_mix,
                                     self._stream())

    def finalize(self, chunks, n_samples, out):
        self.engine.finalize_dev(chunks.data_ptr(), n_samples, out.data_ptr(), self.is_match_mix, self._stream())


class _SiblingAdapter:
    def __init__(self, engine):
        self.engine = engine

    def _stream(self):
        import torch
        return torch.cuda.current_stream().cuda_stream


class MdxcAdapter(_SiblingAdapter):
    """MDXCSeparator.demix, TFC branch (mdxc_separator.py:345-404)."""

    def __init__(self, engine, overlap: int):
        super().__init__(engine)
        self.overlap = int(overlap)
        self.stems = self.out_stems = engine.v3_cfg.num_targets

    def plan(self, n):
        return self.engine.mdxc_plan(n, self.overlap)

    def demix_chunks(self, mix, n, k0, k1, out):
        self.engine.mdxc_chunks_dev(mix.data_ptr(), n, self.overlap, k0, k1, out.data_ptr(), self._stream())

    def finalize(self, chunks, n, out):
        self.engine.mdxc_finalize_dev(chunks.data_ptr(), n, self.overlap, out.data_ptr(), self._stream())


class RoformerAdapter(_SiblingAdapter):
    """MDXCSeparator.demix, Roformer branch (mdxc_separator.py:272-343); step in samples."""

    def __init__(self, engine, step: int):
        super().__init__(engine)
        self.step = int(step)
        self.stems = engine.rof_cfg.num_stems
        self.out_stems = engine.rof_cfg.n_out

    def plan(self, n):
        return self.engine.rof_plan(n, self.step)

    def demix_chunks(self, mix, n, k0, k1, out):
        self.engine.rof_chunks_dev(mix.data_ptr(), n, self.step, k0, k1, out.data_ptr(), self._stream())

    def finalize(self, chunks, n, out):
        self.engine.rof_finalize_dev(chunks.data_ptr(), n, self.step, out.data_ptr(), self._stream())


class DemucsAdapter(_SiblingAdapter):
    """apply_model + demix_demucs (apply.py:124-260, demucs_separator.py:162-194); offsets = the shift draws, identical
    on every rank."""

    def __init__(self, engine, shifts=0, offsets=None, overlap=0.25, flags=3, v3=False):
        """v3: the engine holds a Demucs v3 HDemucs (load_hd) instead of an HTDemucs; same list-of-chunk-forwards split,
        rows of the chunk slab hold each chunk's own length."""
        super().__init__(engine)
        self.kw = dict(shifts=shifts, offsets=offsets, overlap=overlap)
        self.flags = flags
        self.stems = self.out_stems = len((engine.hd_cfg if v3 else engine.ht_cfg).sources)
        self._plan, self._segments, self._fold = ((engine.hd_plan, engine.hd_segments_dev, engine.hd_fold_dev) if v3 else
                                                  (engine.ht_plan, engine.ht_segments_dev, engine.ht_fold_dev))

    def plan(self, n):
        return self._plan(n, **self.kw)

    def demix_chunks(self, mix, n, k0, k1, out):
        self._segments(mix.data_ptr(), n, k0, k1, out.data_ptr(), flags=self.flags, stream=self._stream(), **self.kw)

    def finalize(self, chunks, n, out):
        self._fold(mix_ptr=self._mix.data_ptr(), n=n, chunks_ptr=chunks.data_ptr(), out_ptr=out.data_ptr(), flags=self.flags,
                   stream=self._stream(), **self.kw)

    def bind_mix(self, mix):          # the fold re-derives the standardisation statistics from the mix
        self._mix = mix


class FilesPipeline:
    """BASELINE config 5 (files sharded over the ranks, stems gathered to rank ``dst``): every rank demixes its own songs
    into one of TWO stem buffers and hands the buffer to ONE asynchronous ``dist.gather``; the gather of step k drains while
    step k + 1 computes into the other buffer, and a buffer is reused only after its gather has completed.

    ``demix(mix, out)`` is the per-song hot path (libasx.so's ``demix_dev`` on the GPU; any callable in the CPU tests).
    ``on_gathered(step, slabs)`` (rank ``dst`` only) is called when the gather of ``step`` has completed, before its buffers
    are reused: ``slabs[r]`` is rank r's [S, 2, N] stems of that step.  With ``overlap=False`` the gather blocks."""

    def __init__(self, demix, mixes, world: int, rank: int, use_dist: bool, dst: int = 0, overlap: bool = True, on_gathered=None):
        import torch
        self.demix, self.mixes, self.world, self.rank, self.dst = demix, mixes, world, rank, dst
        self.use_dist, self.overlap, self.on_gathered = use_dist, overlap, on_gathered
        S, shape = len(mixes), tuple(mixes[0].shape)
        mk = lambda: torch.empty((S,) + shape, dtype=mixes[0].dtype, device=mixes[0].device)  # noqa: E731
        self.outs = [mk(), mk()]
        self.gathered = [[mk() for _ in range(world)] for _ in range(2)] if (use_dist and rank == dst) else [None, None]
        self.pending = [None, None]       # (work, step) per buffer
        self.gather_bytes_per_step = (self.outs[0].numel() * self.outs[0].element_size() * (world - 1)) if use_dist else 0

    def _complete(self, b):
        if self.pending[b] is None:
            return
        work, step = self.pending[b]
        if work is not None:
            work.wait()
        self.pending[b] = None
        if self.on_gathered is not None and self.gathered[b] is not None:
            self.on_gathered(step, self.gathered[b])

    def step(self, k: int):
        import torch.distributed as dist
        b = k & 1
        self._complete(b)                 # the stems buffer is free again once its gather has drained
        for s, mix in enumerate(self.mixes):
            self.demix(mix, self.outs[b][s])
        if self.use_dist:
            if self.overlap:
                self.pending[b] = (dist.gather(self.outs[b], self.gathered[b], dst=self.dst, async_op=True), k)
            else:
                dist.gather(self.outs[b], self.gathered[b], dst=self.dst)
                self.pending[b] = (None, k)
                self._complete(b)

    def drain(self):
        # oldest first, so on_gathered sees the steps in order
        order = sorted((b for b in (0, 1) if self.pending[b] is not None), key=lambda b: self.pending[b][1])
        for b in order:
            self._complete(b)


class ShardWorkspace:
    """Buffers of sharded_demix, allocated once per (shape, world) and reused: the strong-scaling loop then times compute +
    gather + fold, not the caching allocator.  ``timings`` (when ``timed``) holds event-measured milliseconds of the last
    call: local compute, gather, fold.  ``poison`` (tests): the local-fold scheme fills every chunk slot it does NOT hold with NaN
    before folding, which proves on each call that the fold of the owned sample range reads none of them (the contract of
    ``local_fold`` below)."""

    def __init__(self, timed: bool = False, poison: bool = False, graph: bool = False):
        self.key = None
        self.timed = timed
        self.poison = poison
        self.timings = {}
        # graph: the rank's chunk-range compute (adapter.demix_chunks: ~130 kernel launches for 7 chunks of the HQ_3 net, a 25-30 ms step at 8 GPUs --
        # where host launch jitter shows first) is captured into a hipGraph on its second call with the same arguments and replayed from then on;
        # the first call runs eagerly (it sizes the engine's workspace and builds its weight images).  The collectives stay outside.
        self.graph = graph
        self._graphs = {}
        self.graph_replays = 0
        self.graph_error = None

    def compute(self, adapter, mix, n, k0, k1, out):
        """adapter.demix_chunks(mix, n, k0, k1, out), through a captured graph when `graph` is set and the tensors are on the GPU"""
        if not self.graph or not mix.is_cuda or self.graph_error is not None:
            return adapter.demix_chunks(mix, n, k0, k1, out)
        import torch
        key = (type(adapter).__name__, id(getattr(adapter, "engine", adapter)), mix.data_ptr(), int(n), int(k0), int(k1), out.data_ptr())
        ent = self._graphs.get(key)
        if ent is None:                                 # first call: eager
            self._graphs[key] = "warm"
            return adapter.demix_chunks(mix, n, k0, k1, out)
        if ent == "warm":                               # second call: capture (nothing runs), then replay
            try:
                g = torch.cuda.CUDAGraph()
                torch.cuda.synchronize()
                # thread_local: another thread of the process (RCCL's watchdog polling its events) must not invalidate the capture
                with torch.cuda.graph(g, capture_error_mode="thread_local"):
                    adapter.demix_chunks(mix, n, k0, k1, out)
                self._graphs[key] = ent = g
            except Exception as e:                      # a capture the runtime refuses must never take the run down: stay eager
                self.graph_error = f"{type(e).__name__}: {e}"
                self._graphs.pop(key, None)
                return adapter.demix_chunks(mix, n, k0, k1, out)
        ent.replay()
        self.graph_replays += 1

    def get(self, key, make):
        if self.key != key:
            self.key = key
            self.bufs = make()
        return self.bufs


def _sharded_demix_local_fold(adapter, mix, plan, ranges, group, dst, workspace, world, rank, is_dst):
    """fold="local": own chunks + the halo chunks of the ranks before -> fold of the own sample range -> gather of the slabs."""
    import torch
    import torch.distributed as dist
    n = mix.shape[-1]
    nk, C = plan["n_chunks"], plan["chunk_size"]
    k0, k1 = ranges[rank]
    own = owned_samples(plan, ranges, n)
    W = max(1, max(j1 - j0 for j0, j1 in own))
    h = halo_chunks(plan)
    owner = [r for r, (a, b) in enumerate(ranges) for _ in range(a, b)]          # chunk -> rank

    def make():
        bufs = {"allc": torch.zeros((nk, 2, C), dtype=torch.float32, device=mix.device),
                "full": torch.empty(tuple(mix.shape), dtype=torch.float32, device=mix.device),
                "slab": torch.zeros((2, W), dtype=torch.float32, device=mix.device)}
        if is_dst:
            bufs["out"] = torch.empty(tuple(mix.shape), dtype=torch.float32, device=mix.device)
            bufs["slabs"] = [torch.empty((2, W), dtype=torch.float32, device=mix.device) for _ in range(world)]
        return bufs
    bufs = workspace.get(("local", nk, C, n, W, world, str(mix.device), is_dst), make) if workspace is not None else make()
    timed = workspace is not None and workspace.timed and mix.is_cuda
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)] if timed else None
    allc = bufs["allc"]
    if timed:
        ev[0].record()
    if k1 > k0:
        if workspace is not None:
            workspace.compute(adapter, mix, n, k0, k1, allc[k0:k1])
        else:
            adapter.demix_chunks(mix, n, k0, k1, allc[k0:k1])
    if timed:
        ev[1].record()
    # seam halos: chunk k goes to every LATER rank whose first chunk lies within h chunks of it (its right neighbour, unless
    # ranges are shorter than the halo)
    ops = []
    to_global = (lambda r: dist.get_global_rank(group, r)) if group is not None else (lambda r: r)
    for r, (a, b) in enumerate(ranges):
        if b <= a or r == 0:
            continue
        for k in range(max(0, a - h), a):
            q = owner[k]
            if q == rank and r != rank:
                ops.append(dist.P2POp(dist.isend, allc[k], to_global(r), group))
            elif r == rank and q != rank:
                ops.append(dist.P2POp(dist.irecv, allc[k], to_global(q), group))
    if ops:
        for w in dist.batch_isend_irecv(ops):
            w.wait()
    j0, j1 = own[rank]
    if workspace is not None and getattr(workspace, "poison", False):
        held = set(range(k0, k1)) | set(range(max(0, k0 - h), k0))
        for k in range(nk):
            if k not in held:
                allc[k].fill_(float("nan"))
    if j1 > j0:
        # CONTRACT of `local_fold` adapters: finalize is a pure per-sample GATHER -- out[j] depends only on the chunks that cover j
        # (visited in chunk order) and on an input-independent divider.  Every chunk covering [j0, j1) is here (own range + halo);
        # all other slots of `allc` are zeros, stale data of the previous song, or NaN under ShardWorkspace(poison=True), and only
        # [j0, j1) of the result is read.  A scatter / accumulate fold or a measured divider would break this silently: such an
        # adapter must not set local_fold (tests/test_sharding_gloo.py::test_local_fold_ignores_foreign_chunks).
        adapter.finalize(allc, n, bufs["full"])
        bufs["slab"][:, : j1 - j0].copy_(bufs["full"][:, j0:j1])
    if timed:
        ev[2].record()
    dist.gather(bufs["slab"], bufs["slabs"] if is_dst else None, dst=dst, group=group)
    out = None
    if is_dst:
        out = bufs["out"]
        for r, (a, b) in enumerate(own):
            if b > a:
                out[:, a:b].copy_(bufs["slabs"][r][:, : b - a])
    if timed:
        ev[3].record()
        torch.cuda.synchronize()
        workspace.timings = {"compute_ms": ev[0].elapsed_time(ev[1]), "fold_ms": ev[1].elapsed_time(ev[2]),
                             "gather_ms": ev[2].elapsed_time(ev[3]), "scheme": "local fold: halo send/recv + fold of the own sample "
                             "range (fold_ms), then one gather of [2, N / G] slabs + placement on dst (gather_ms)"}
    return out


def sharded_demix(adapter, mix, group=None, dst: int = 0, workspace: ShardWorkspace | None = None, fold: str = "auto"):
    """Demix one song across all ranks of ``group``.

    ``mix``: float32 tensor [2, N] on the adapter's device, identical on every
    rank.  Returns the separated [2, N] tensor on rank ``dst`` and None elsewhere.
    With a ``workspace`` no tensor is allocated after the first call and the returned tensor is the workspace's (it is
    overwritten by the next call).  ``fold``: "local" / "dst" (module docstring); "auto" = local when the adapter declares
    ``local_fold`` (uniform chunk stride, one [2, N] output) and its plan carries ``step`` / ``padded_len``.
    """
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0          # position inside the group: picks the chunk range
    is_dst = (dist.get_rank() == dst) if dist.is_initialized() else True  # `dst` is a GLOBAL rank, like dist.gather's
    n = mix.shape[-1]
    plan = adapter.plan(n)
    nk, C = plan["n_chunks"], plan["chunk_size"]
    ranges = partition_chunks(nk, world)
    can_local = getattr(adapter, "local_fold", False) and "step" in plan and "padded_len" in plan and \
        getattr(adapter, "stems", None) is None and getattr(adapter, "out_stems", None) is None
    if fold == "local" and not can_local:
        raise ValueError("fold='local' needs a uniform-stride adapter (local_fold) whose plan has step / padded_len")
    if world > 1 and (fold == "local" or (fold == "auto" and can_local)):
        return _sharded_demix_local_fold(adapter, mix, plan, ranges, group, dst, workspace, world, rank, is_dst)
    k0, k1 = ranges[rank]
    per = max(b - a for a, b in ranges)          # equal-size slabs keep it a single gather
    stems = getattr(adapter, "stems", None)
    out_stems = getattr(adapter, "out_stems", None)
    cshape = (per, 2, C) if stems is None else (per, stems, 2, C)
    oshape = tuple(mix.shape) if out_stems is None else (out_stems, 2, n)
    if hasattr(adapter, "bind_mix"):
        adapter.bind_mix(mix)
    even = all(b - a == per for a, b in ranges)     # equal ranges: the gathered slabs ARE the chunk list, no compaction copy

    def make():
        bufs = {"local": torch.zeros(cshape, dtype=torch.float32, device=mix.device)}
        if is_dst:
            bufs["out"] = torch.empty(oshape, dtype=torch.float32, device=mix.device)
            if world > 1:
                bufs["slab"] = torch.empty((world * per,) + cshape[1:], dtype=torch.float32, device=mix.device)
                if not even:
                    bufs["allc"] = torch.empty((nk,) + cshape[1:], dtype=torch.float32, device=mix.device)
        return bufs
    bufs = workspace.get((cshape, oshape, world, str(mix.device), is_dst), make) if workspace is not None else make()
    timed = workspace is not None and workspace.timed and mix.is_cuda
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)] if timed else None
    local = bufs["local"]
    if timed:
        ev[0].record()
    if k1 > k0:
        if workspace is not None:
            workspace.compute(adapter, mix, n, k0, k1, local[: k1 - k0])
        else:
            adapter.demix_chunks(mix, n, k0, k1, local[: k1 - k0])
    if timed:
        ev[1].record()
    if world == 1:
        out = bufs["out"]
        adapter.finalize(local[:nk], n, out)
        if timed:
            ev[2].record()
            ev[3].record()
            torch.cuda.synchronize()
            workspace.timings = {"compute_ms": ev[0].elapsed_time(ev[1]), "gather_ms": 0.0, "fold_ms": ev[1].elapsed_time(ev[2])}
        return out
    slabs = [bufs["slab"][r * per:(r + 1) * per] for r in range(world)] if is_dst else None
    dist.gather(local, slabs, dst=dst, group=group)
    if timed:
        ev[2].record()
    out = None
    if is_dst:
        if even:
            allc = bufs["slab"]
        else:
            allc = bufs["allc"]
            for r, (a, b) in enumerate(ranges):
                if b > a:
                    allc[a:b].copy_(slabs[r][: b - a])
        out = bufs["out"]
        adapter.finalize(allc, n, out)
    if timed:
        ev[3].record()
        torch.cuda.synchronize()
        workspace.timings = {"compute_ms": ev[0].elapsed_time(ev[1]), "gather_ms": ev[1].elapsed_time(ev[2]),
                             "fold_ms": ev[2].elapsed_time(ev[3])}
    return out
