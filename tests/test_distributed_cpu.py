"""The multi-GPU driver logic on CPU: two processes over gloo, each rendering
its share of the samples (with the CPU oracle standing in for the GPU library
behind the same SumsBackend interface), reduced onto rank 0."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from raytracer_amd.distributed import render_pass, render_pass_exact, sample_share


def test_sample_shares_tile_the_range():
    for begin, count, world in ((0, 256, 8), (10, 7, 3), (0, 2, 4), (5, 0, 2), (0, 1, 1)):
        shares = [sample_share(begin, count, world, r) for r in range(world)]
        pos = begin
        for first, n in shares:
            assert first == pos and n >= 0
            pos += n
        assert pos == begin + count
        sizes = [n for _, n in shares]
        assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        sample_share(0, 4, 2, 2)


class OracleBackend:
    """SumsBackend over the CPU oracle (test stand-in for GpuSumsBackend)."""

    def __init__(self, scene):
        self.scene = scene

    def render_sums(self, begin, count):
        from oracle import binding as O
        return torch.from_numpy(O.render_sums(self.scene, self.scene.params, begin, count).reshape(-1).copy())

    def zeros(self):
        h, w = self.scene.params.tile_shape()
        return torch.zeros(h * w * 3, dtype=torch.float32)


def _worker(rank, world, port, spp, out_path):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from tests import scenes
    scene = scenes.materials(w=24, h=16, spp=spp)
    sums = render_pass(OracleBackend(scene), 3, spp, world, rank, lambda t: dist.reduce(t, dst=0, op=dist.ReduceOp.SUM))
    if rank == 0:
        np.save(out_path, sums.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world, spp", [(2, 6), (2, 1), (3, 5)])
def test_sample_partitioned_render_matches_single_process(tmp_path, world, spp):
    from oracle import binding as O
    from tests import scenes

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = str(tmp_path / "sums.npy")
    mp.spawn(_worker, args=(world, port, spp, out), nprocs=world, join=True)
    got = np.load(out).reshape(16, 24, 3)
    scene = scenes.materials(w=24, h=16, spp=spp)
    want = O.render_sums(scene, scene.params, 3, spp)
    # identical samples; only the association of the per-rank partial sums differs
    np.testing.assert_allclose(got, want, rtol=2e-6, atol=1e-6)
    # and exactly equal to adding the per-rank oracle sums in rank order
    parts = [O.render_sums(scene, scene.params, *sample_share(3, spp, world, r)) for r in range(world)
             if sample_share(3, spp, world, r)[1] > 0]
    acc = parts[0]
    for p in parts[1:]:
        acc = acc + p
    if world == 2:
        assert np.array_equal(got, acc)


class OracleExactBackend:
    """trace / fold / zeros / buffer over the CPU oracle: the per-sample radiance is kept
    on the side (at most `capacity` samples at a time), the fold is the sequential
    binary32 addition of /root/reference/src/lib.rs:274-279."""

    def __init__(self, scene, capacity):
        self.scene, self.capacity, self.pending = scene, capacity, []
        h, w = scene.params.tile_shape()
        self.buffer = torch.full((h * w * 3,), float("nan"), dtype=torch.float32)

    def reserve(self, count):
        return min(count, self.capacity)

    def trace(self, begin, count):
        from oracle import binding as O
        n = min(count, self.capacity)
        self.pending = [torch.from_numpy(O.render_sums(self.scene, self.scene.params, begin + k, 1).reshape(-1).copy())
                        for k in range(n)]
        return n

    def fold(self, carry_on):
        samples = list(self.pending)
        acc = self.buffer.clone() if carry_on else samples.pop(0)
        for x in samples:
            acc = acc + x
        self.buffer.copy_(acc)
        self.pending = []
        return self.buffer

    def zeros(self):
        return self.buffer.zero_()


def _exact_worker(rank, world, port, spp, capacity, out_path):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from tests import scenes
    scene = scenes.materials(w=24, h=16, spp=spp)
    def agree_min(n):
        t = torch.tensor([n])
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        return int(t)

    # the ranks hold different numbers of samples in flight: the pieces follow the smallest
    sums = render_pass_exact(OracleExactBackend(scene, capacity + rank), 0, spp, world, rank, dist.send, dist.recv,
                             agree_min=agree_min if world > 1 else None)
    if rank == 0:
        np.save(out_path, sums.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world, spp, capacity", [(2, 7, 8), (3, 8, 2), (3, 2, 8), (1, 5, 2), (2, 9, 1)])
def test_exact_pass_is_the_single_process_fold_bit_for_bit(tmp_path, world, spp, capacity):
    from oracle import binding as O
    from tests import scenes

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = str(tmp_path / "exact.npy")
    mp.spawn(_exact_worker, args=(world, port, spp, capacity, out), nprocs=world, join=True)
    got = np.load(out).reshape(16, 24, 3)
    scene = scenes.materials(w=24, h=16, spp=spp)
    want = O.render_sums(scene, scene.params, 0, spp)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
