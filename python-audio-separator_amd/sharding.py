"""Multi-GPU sharding of the chunk loop (one process per GPU, torch.distributed).

The chunks of MDXSeparator.demix are independent given the replicated mix
(mdx_separator.py:348-392), so rank r runs a contiguous chunk range
on its own engine and the only exchange is ONE gather of the windowed chunk
outputs to rank 0, which folds them (result / divider) exactly like the
single-GPU path -- the gathered result is bit-identical to a one-GPU run.

The driver is backend agnostic: it talks to an *engine adapter* with three
methods operating on torch tensors that live on the adapter's device,

    plan(n_samples) -> dict            (chunk_size, n_chunks, ...)
    demix_chunks(mix, n, k0, k1, out)  windowed chunks [k1-k0, 2, C] -> out
    finalize(chunks, n, out)           all chunks [n_chunks, 2, C] -> out [2, n]

``HipEngineAdapter`` binds those to libasx.so.  With backend "nccl" the gather
is RCCL over xGMI; the CPU tests drive the same code over "gloo".
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


class HipEngineAdapter:
    """Engine adapter over libasx.so for CUDA(HIP) torch tensors."""

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


def sharded_demix(adapter, mix, group=None, dst: int = 0):
    """Demix one song across all ranks of ``group``.

    ``mix``: float32 tensor [2, N] on the adapter's device, identical on every
    rank.  Returns the separated [2, N] tensor on rank ``dst`` and None elsewhere.
    """
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    n = mix.shape[-1]
    plan = adapter.plan(n)
    nk, C = plan["n_chunks"], plan["chunk_size"]
    ranges = partition_chunks(nk, world)
    k0, k1 = ranges[rank]
    per = max(b - a for a, b in ranges)          # equal-size slabs keep it a single gather
    local = torch.zeros((per, 2, C), dtype=torch.float32, device=mix.device)
    if k1 > k0:
        adapter.demix_chunks(mix, n, k0, k1, local[: k1 - k0])
    if world == 1:
        out = torch.empty_like(mix)
        adapter.finalize(local[:nk], n, out)
        return out
    slabs = [torch.empty_like(local) for _ in range(world)] if rank == dst else None
    dist.gather(local, slabs, dst=dst, group=group)
    if rank != dst:
        return None
    allc = torch.cat([slabs[r][: b - a] for r, (a, b) in enumerate(ranges)], dim=0)
    out = torch.empty_like(mix)
    adapter.finalize(allc, n, out)
    return out
