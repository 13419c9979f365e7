"""Tile-by-tile rendering into the memory-mapped image on the GPU: the tiled
driver must store exactly the bytes of an untiled `rt_render` (and of the CPU
oracle), through incremental passes and a `continue` restart from the file."""
import os
import subprocess

import numpy as np
import pytest

from oracle import binding as O
from raytracer_amd import _ffi as F
from raytracer_amd import assets
from raytracer_amd.tiles import GpuTileBackend, Image, render_tiled

from . import scenes

pytestmark = pytest.mark.gpu


def _with(scene, spp, incremental=None, cont=None):
    p = scene.params.copy()
    p.samples_per_pixel = spp
    p.incremental = incremental or 0
    p.has_continue = 0 if cont is None else 1
    p.continue_sampling = cont or 0
    return p


def test_tiled_image_equals_untiled_and_oracle(renderer, asset_root, tmp_path):
    scene = scenes.room(asset_root, w=96, h=60, spp=4, bounces=10)
    renderer.upload(scene)
    p = _with(scene, 4, incremental=2)
    want = renderer.render(p)
    cpu, _ = O.render(scene, p, threads=4)
    assert np.array_equal(want, cpu)
    seen = []
    with Image(str(tmp_path / "out.ppm"), 96, 60) as img:
        render_tiled(GpuTileBackend(renderer), img, p, 40, 25, progress=seen.append)
        assert np.array_equal(img.pixels(), want)
    assert seen == [2, 4]


def test_tiled_continue_from_the_file(renderer, asset_root, tmp_path):
    scene = scenes.room(asset_root, w=64, h=48, spp=2, bounces=8)
    renderer.upload(scene)
    path = str(tmp_path / "c.ppm")
    first, second = _with(scene, 4, incremental=2), _with(scene, 2, incremental=2, cont=2)
    with Image(path, 64, 48) as img:
        render_tiled(GpuTileBackend(renderer), img, first, 64, 16)
    with Image(path, 64, 48) as img:  # a new mapping of the same file: its bytes are the checkpoint
        render_tiled(GpuTileBackend(renderer), img, second, 20, 48)
        got = img.pixels().copy()
    want, _ = O.render(scene, first, threads=4)
    want, _ = O.render(scene, second, want, threads=4)
    assert np.array_equal(got, want)
    # and the untiled GPU path agrees
    gpu = renderer.render(second, renderer.render(first))
    assert np.array_equal(got, gpu)


def test_group_of_one_rank_uses_the_sums_path_identically(renderer, asset_root, tmp_path):
    # the path several ranks of a group take (render_sums -> reduce -> resolve), with a group of one
    import torch

    scene = scenes.materials(w=48, h=32, spp=4)
    renderer.upload(scene)
    p = _with(scene, 4, incremental=2)
    be = GpuTileBackend(renderer, torch.device("cuda", 0))

    reduced = []
    with Image(str(tmp_path / "g.ppm"), 48, 32) as img:
        from raytracer_amd import tiles
        grid = tiles.tile_grid(48, 32, 16, 16)
        chunk, n_passes, running, first = tiles.passes(p)
        for k in range(n_passes):
            for rect in grid:
                tp = tiles.pass_params(p, rect, chunk, running, k)
                sums = be.render_sums(tp, k * chunk, chunk)
                reduced.append(float(sums.sum()))
                x, y, w, h = rect
                img.write_tile(x, y, be.resolve(tp, sums, chunk, running, k, img.read_tile(x, y, w, h)))
        got = img.pixels().copy()
    assert np.array_equal(got, renderer.render(p)) and all(np.isfinite(reduced))


def test_command_line_renderer_tiles_large_frames(asset_root, tmp_path):
    exe = os.path.join(F.LIB_DIR, "rt_render")
    scene_file = tmp_path / "scene"
    scene_file.write_text(assets.room_scene_text("room", 80, 50, 4, 8, incremental=2))
    env = dict(os.environ)
    out_a, out_b = str(tmp_path / "a.ppm"), str(tmp_path / "b.ppm")
    subprocess.run([exe, str(scene_file), out_a, "0"], cwd=asset_root, check=True, env=env, timeout=120)
    subprocess.run([exe, str(scene_file), out_b, "0", "32"], cwd=asset_root, check=True, env=env, timeout=120)
    a, b = open(out_a, "rb").read(), open(out_b, "rb").read()
    assert a.startswith(b"P6\n80 50 255\n") and a == b
    # a list of GPUs (here GPU 0 three times) shares every pass's samples: the same file
    out_c = str(tmp_path / "c.ppm")
    subprocess.run([exe, str(scene_file), out_c, "0,0,0", "32"], cwd=asset_root, check=True, env=env, timeout=120)
    assert open(out_c, "rb").read() == a
    scene = scenes.room(asset_root, w=80, h=50, spp=4, bounces=8, incremental=2)
    cpu, _ = O.render(scene, scene.params, threads=4)
    assert a[len(b"P6\n80 50 255\n"):] == cpu.tobytes()
    # the opt-in refined tree (RT_RENDER_MAX_LEAF): the oracle on the same refined scene stores the same bytes
    out_d = str(tmp_path / "d.ppm")
    subprocess.run([exe, str(scene_file), out_d, "0"], cwd=asset_root, check=True, env=dict(env, RT_RENDER_MAX_LEAF="4"), timeout=120)
    refined, _ = scene.refined(4)
    cpu_refined, _ = O.render(refined, refined.params, threads=4)
    assert open(out_d, "rb").read()[len(b"P6\n80 50 255\n"):] == cpu_refined.tobytes()


def _hybrid_worker(rank, world, port, group_size, path, asset_root, exact=False):
    """One rank of the tile x sample hybrid, all ranks on GPU 0, reducing over gloo
    through host memory (a one-GPU box has no second GPU for RCCL to reach)."""
    import torch  # noqa: F401  (before the renderer: torch's HIP runtime must be the one loaded)
    import torch.distributed as dist

    from raytracer_amd.renderer import Renderer

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n_groups = world // group_size
    groups = [dist.new_group(list(range(g * group_size, (g + 1) * group_size))) for g in range(n_groups)]
    mine, root = groups[rank // group_size], (rank // group_size) * group_size

    def reduce_in_group(t):
        host = t.cpu()
        dist.reduce(host, dst=root, op=dist.ReduceOp.SUM, group=mine)
        t.copy_(host)

    def send_to_member(t, member):
        dist.send(t.cpu(), root + member)

    def recv_from_member(t, member):
        host = torch.empty(t.shape, dtype=t.dtype)
        dist.recv(host, root + member)
        t.copy_(host)

    def agree_min(n):
        t = torch.tensor([n], dtype=torch.int64)
        dist.all_reduce(t, op=dist.ReduceOp.MIN, group=mine)
        return int(t.item())

    scene = scenes.room(asset_root, w=72, h=48, spp=4, bounces=8)
    p = _with(scene, 4, incremental=2)
    with Renderer(0) as r:
        r.upload(scene)
        with Image(path, 72, 48) as img:
            render_tiled(GpuTileBackend(r, torch.device("cuda", 0)), img, p, 24, 24, world=world, rank=rank,
                         group_size=group_size, reduce_in_group=reduce_in_group if group_size > 1 and not exact else None,
                         exact_in_group=(send_to_member, recv_from_member, agree_min) if exact else None, end_of_pass=dist.barrier)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world, group_size, exact", [(2, 1, False), (2, 2, False), (4, 2, False), (2, 2, True), (3, 3, True)])
def test_tiles_dealt_to_groups_of_ranks(asset_root, tmp_path, world, group_size, exact):
    import socket

    import torch.multiprocessing as mp

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    path = str(tmp_path / "hybrid.ppm")
    mp.spawn(_hybrid_worker, args=(world, port, group_size, path, asset_root, exact), nprocs=world, join=True)
    scene = scenes.room(asset_root, w=72, h=48, spp=4, bounces=8)
    want, _ = O.render(scene, _with(scene, 4, incremental=2), threads=4)
    with Image(path, 72, 48) as img:
        got = img.pixels().copy()
    if group_size == 1 or exact:
        assert np.array_equal(got, want)  # tiles only, or sums handed over in sample order: no arithmetic differs
    else:
        # the group's partial sums are added in another association: a truncated byte may move by one
        assert np.abs(got.astype(int) - want.astype(int)).max() <= 1
