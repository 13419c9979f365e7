"""Trace-then-fold (rt_trace_samples / rt_fold_samples) and the multi-GPU pass built on
it (raytracer_amd.distributed.render_pass_exact): the radiance sums do not depend on how
many GPUs shared the samples — they are the oracle's single fold, bit for bit."""
import os
import socket

import numpy as np
import pytest

from oracle import binding as O
from raytracer_amd import _ffi as F
from raytracer_amd.distributed import sample_share

from . import scenes

pytestmark = pytest.mark.gpu


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


@pytest.mark.parametrize("in_flight", [0, 2])
def test_shares_traced_apart_and_folded_in_order(renderer, in_flight):
    from raytracer_amd.renderer import RtError

    s = scenes.materials(w=40, h=24, spp=9, bounces=12)
    renderer.upload(s)
    renderer.set_option(F.RT_OPT_SAMPLES_IN_FLIGHT, in_flight)
    try:
        whole = O.render_sums(s, s.params, 0, 9, threads=4)
        acc = np.full((24, 40, 3), np.nan, dtype=np.float32)
        with pytest.raises(RtError):
            renderer.fold_samples(s.params, False, acc)  # nothing traced yet
        for rank in range(3):  # the three shares of a world of three, one after the other on this GPU
            first, mine = sample_share(0, 9, 3, rank)
            done = 0
            while done < mine:
                n = renderer.trace_samples(s.params, first + done, mine - done)
                assert 1 <= n <= mine - done and (in_flight == 0 or n <= in_flight)
                renderer.fold_samples(s.params, rank > 0 or done > 0, acc)
                done += n
        assert np.array_equal(bits(acc), bits(whole))
        # samples traced for one region cannot be folded as another's
        tile = s.params.copy()
        tile.tile_x, tile.tile_y, tile.tile_w, tile.tile_h = 0, 0, 20, 24
        other = s.params.copy()
        other.tile_x, other.tile_y, other.tile_w, other.tile_h = 20, 0, 20, 24
        renderer.trace_samples(tile, 0, 1)
        with pytest.raises(RtError):
            renderer.fold_samples(other, False, np.zeros((24, 20, 3), dtype=np.float32))
        half = renderer.fold_samples(tile, False, np.zeros((24, 20, 3), dtype=np.float32))
        assert np.array_equal(bits(half), bits(O.render_sums(s, tile, 0, 1)))
        # a render in between discards what was traced
        renderer.trace_samples(s.params, 0, 1)
        renderer.render_sums(s.params, 0, 1)
        with pytest.raises(RtError):
            renderer.fold_samples(s.params, False, acc)
    finally:
        renderer.set_option(F.RT_OPT_SAMPLES_IN_FLIGHT, 0)


def _exact_worker(rank, world, port, asset_root, out_path):
    """One rank of the exact pass, all ranks on GPU 0, the sums travelling over gloo
    through host memory (a one-GPU box has no second GPU for RCCL to reach)."""
    import torch  # noqa: F401  (before the renderer: torch's HIP runtime must be the one loaded)
    import torch.distributed as dist

    from raytracer_amd.distributed import GpuSumsBackend, render_pass_exact
    from raytracer_amd.renderer import Renderer

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)

    def send(t, dst):
        dist.send(t.cpu(), dst)

    def recv(t, src):
        host = torch.empty(t.shape, dtype=t.dtype)
        dist.recv(host, src)
        t.copy_(host)

    scene = scenes.room(asset_root, w=72, h=48, spp=7, bounces=8)
    with Renderer(0) as r:
        r.upload(scene)
        backend = GpuSumsBackend(r, scene.params, torch.device("cuda", 0))
        # rank 1 holds a single sample in flight: the pass then takes rounds of one sample per rank
        r.set_option(F.RT_OPT_SAMPLES_IN_FLIGHT, 1 if rank == 1 else 0)

        def agree_min(n):
            t = torch.tensor([n])
            dist.all_reduce(t, op=dist.ReduceOp.MIN)
            return int(t)

        sums = render_pass_exact(backend, 0, 7, world, rank, send, recv, agree_min=agree_min)
        if rank == 0:
            np.save(out_path, sums.cpu().numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_exact_pass_over_several_ranks(asset_root, tmp_path, world):
    import torch.multiprocessing as mp

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = str(tmp_path / "exact.npy")
    mp.spawn(_exact_worker, args=(world, port, asset_root, out), nprocs=world, join=True)
    scene = scenes.room(asset_root, w=72, h=48, spp=7, bounces=8)
    want = O.render_sums(scene, scene.params, 0, 7, threads=4)
    assert np.array_equal(bits(np.load(out).reshape(48, 72, 3)), bits(want))
