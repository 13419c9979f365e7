"""Sample-partitioned rendering across the GPUs of one node.

One process per GPU (torch.distributed; backend "nccl" is RCCL over xGMI).
The scene is replicated; rank g of G renders a contiguous share of every pass's
samples for all pixels into its own float radiance-sum buffer, and one
reduce(SUM) per pass brings the buffers to rank 0, which applies the mean,
gamma and quantisation of /root/reference/src/lib.rs:283-295.  This replaces
the reference's only parallelism, threads pulling pixel chunks off a shared
counter (/root/reference/src/lib.rs:213-235).

Every (sample, pixel) has its own generator sub-stream, so the set of samples
rendered does not depend on G; only the order in which per-rank partial sums
are added differs from the single-GPU left fold (last-bit float differences).
"""
from __future__ import annotations

from typing import Callable, Protocol


def sample_share(begin: int, count: int, world: int, rank: int) -> tuple[int, int]:
    """Contiguous share of samples [begin, begin+count) for `rank` of `world`:
    sizes differ by at most one, earlier ranks take the larger shares, shares
    tile the range exactly.  Returns (first sample, number of samples)."""
    if world < 1 or not 0 <= rank < world:
        raise ValueError("bad rank/world")
    base, extra = divmod(count, world)
    mine = base + (1 if rank < extra else 0)
    first = begin + rank * base + min(rank, extra)
    return first, mine


class SumsBackend(Protocol):
    """What a rank needs: radiance sums of a sample range as a torch tensor
    (3 floats per pixel, on the device the process group reduces on)."""

    def render_sums(self, sample_begin: int, sample_count: int): ...

    def zeros(self): ...


def render_pass(backend: SumsBackend, sample_begin: int, sample_count: int, world: int, rank: int,
                reduce_to_root: Callable | None):
    """One pass: this rank's share, then the reduce.  Returns the summed tensor
    (complete on rank 0 only).  `reduce_to_root(tensor)` is
    `dist.reduce(tensor, dst=0, op=SUM)`; None when world == 1."""
    first, mine = sample_share(sample_begin, sample_count, world, rank)
    sums = backend.render_sums(first, mine) if mine > 0 else backend.zeros()
    if world > 1:
        reduce_to_root(sums)
    return sums


class GpuSumsBackend:
    """SumsBackend over librt_mi355.so: the library writes straight into a
    torch CUDA tensor through its device pointer."""

    def __init__(self, renderer, params, device):
        import torch

        self._torch = torch
        self.renderer, self.params = renderer, params
        h, w = params.tile_shape()
        self.buffer = torch.empty(h * w * 3, dtype=torch.float32, device=device)

    def render_sums(self, sample_begin: int, sample_count: int):
        self.renderer.render_sums_device(self.params, sample_begin, sample_count, self.buffer.data_ptr())
        return self.buffer

    def zeros(self):
        return self.buffer.zero_()
