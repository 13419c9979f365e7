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

`render_pass_exact` removes that difference: the ranks trace contiguous pieces of
the pass at the same time, then the running sums go from rank to rank in sample
order and every rank folds its samples onto them, which is the single-GPU
sequence of additions.  The price is a chain of point-to-point transfers of the
sums (world - 1 per round of pieces) instead of one reduce.
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


def fold_in_sample_order(trace, fold, buffer, zeros, reserve, sample_begin: int, sample_count: int, world: int,
                         rank: int, send: Callable | None, recv: Callable | None, agree_min: Callable | None = None):
    """The running sums of samples [sample_begin, sample_begin + sample_count) by the single-GPU
    sequence of additions, over `world` ranks; complete on the last rank.

    The samples are taken in rounds of one contiguous piece per rank (rank k the k-th piece of
    the round).  In a round every rank traces its piece at once, then the sums go down the
    ranks in order, each folding its piece onto them; the last rank hands them to rank 0 for
    the next round.  A rank starts tracing its next piece as soon as it has passed the sums
    on, so the ranks work side by side however many rounds a pass takes.  A piece is no
    larger than what the fullest rank holds in flight (`reserve(count)` -> samples one
    `trace` call takes; `agree_min(n)` -> the minimum of n over the ranks, e.g. an all-reduce
    with MIN.  With more than one rank it is required: `reserve` answers with what fits the
    rank's free memory, and ranks that disagreed on the piece size would wait for transfers
    that never come).

    trace(first, n) -> samples traced; fold(carry_on) -> sums; buffer() -> the tensor to
    receive into; zeros() -> sums of nothing; send(sums, dst) / recv(sums, src)."""
    if world > 1 and agree_min is None:
        raise ValueError("agree_min is required with more than one rank (the ranks must use one piece size)")
    per_rank = -(-sample_count // world)
    fit = reserve(per_rank) if per_rank else 1
    if world > 1:
        fit = int(agree_min(fit))
    if fit < 1:
        raise RuntimeError("no room for a single sample in flight")
    sums, done = None, 0
    while True:
        left = sample_count - done
        piece = min(fit, -(-left // world))
        final = world * piece >= left
        at = rank * piece
        mine = max(0, min(piece, left - at))
        if mine and trace(sample_begin + done + at, mine) != mine:
            raise RuntimeError("traced fewer samples than reserved room for")
        carry_on = rank > 0 or done > 0  # someone holds the fold so far: the rank before, or the last rank of the round before
        if carry_on and world > 1:
            sums = buffer()
            recv(sums, rank - 1 if rank > 0 else world - 1)
        if mine:
            sums = fold(carry_on)
        elif not carry_on:
            sums = zeros()  # no samples at all
        if rank + 1 < world:
            send(sums, rank + 1)
        elif not final and world > 1:
            send(sums, 0)
        if final:
            return sums
        done += world * piece


def render_pass_exact(backend, sample_begin: int, sample_count: int, world: int, rank: int,
                      send: Callable | None, recv: Callable | None, to_rank0: bool = True,
                      agree_min: Callable | None = None):
    """One pass whose result does not depend on `world`, bit for bit.  Returns the tensor
    of sums, complete on the last rank (world - 1) and, with `to_rank0`, on rank 0.
    `send(tensor, dst)` / `recv(tensor, src)` are `dist.send` / `dist.recv`; None when
    world == 1.  `backend`: reserve(count) -> samples one trace call takes, trace(begin,
    count) -> samples traced, fold(carry_on) -> tensor, zeros(), .buffer (the tensor all of
    them work on).  `agree_min`: see fold_in_sample_order."""
    fold_in_sample_order(backend.trace, backend.fold, lambda: backend.buffer, backend.zeros, backend.reserve,
                         sample_begin, sample_count, world, rank, send, recv, agree_min)
    if to_rank0 and world > 1:
        if rank == world - 1:
            send(backend.buffer, 0)
        elif rank == 0:
            recv(backend.buffer, world - 1)
    return backend.buffer


class GpuSumsBackend:
    """SumsBackend over librt_mi355.so: the library writes straight into a
    torch CUDA tensor through its device pointer.

    Ordering contract.  The library works on a stream of its own and returns from
    every call only after that stream has drained, so torch (and RCCL) may use the
    buffer right after a method here returns.  The other direction is the caller's
    duty: torch.distributed's NCCL operations return once they are *enqueued*, so
    `reduce_to_root`, `send` and `recv` passed to render_pass / render_pass_exact
    must not return before the transfer has completed on the device (finish them
    with torch.cuda.synchronize(), or wait on the work handle and synchronise the
    stream), or the library's next call may read or overwrite data in flight.
    `settle()` does that synchronisation; fold() and render_sums() call it first, so
    a bare dist.recv / dist.reduce as callback is safe with this backend."""

    def __init__(self, renderer, params, device):
        import torch

        self._torch = torch
        self.renderer, self.params = renderer, params
        h, w = params.tile_shape()
        self.buffer = torch.empty(h * w * 3, dtype=torch.float32, device=device)

    def settle(self) -> None:
        """Wait for everything torch and RCCL have enqueued on this device."""
        self._torch.cuda.synchronize(self.buffer.device)

    def render_sums(self, sample_begin: int, sample_count: int):
        self.settle()  # an earlier collective may still be reading the buffer
        self.renderer.render_sums_device(self.params, sample_begin, sample_count, self.buffer.data_ptr())
        return self.buffer

    def zeros(self):
        self.buffer.zero_()
        self.settle()
        return self.buffer

    def reserve(self, sample_count: int) -> int:
        return self.renderer.reserve_samples(self.params, sample_count)

    def trace(self, sample_begin: int, sample_count: int) -> int:
        return self.renderer.trace_samples(self.params, sample_begin, sample_count)

    def fold(self, carry_on: bool):
        self.settle()  # a receive into the buffer must have landed before it is folded onto
        self.renderer.fold_samples(self.params, carry_on, self.buffer.data_ptr(), sums_on_device=True)
        return self.buffer
