"""The tile-by-tile image driver on CPU: tile geometry, the mmap'd image's tile
copies, and the driver itself with the CPU oracle standing in for the GPU
library behind the TileBackend interface — alone, and as two gloo processes
(tile partition without a collective; sample partition with a reduce)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from raytracer_amd.tiles import Image, passes, render_tiled, tile_grid, tiles_of_group


def test_tile_grid_covers_the_image_once():
    for w, h, tw, th in ((64, 36, 16, 16), (50, 30, 16, 7), (5, 3, 8, 8), (16384, 16384, 2048, 2048)):
        grid = tile_grid(w, h, tw, th)
        cover = np.zeros((h, w), dtype=np.int32) if w * h < 1 << 20 else None
        area = 0
        for x, y, a, b in grid:
            assert 0 < a <= tw and 0 < b <= th and x + a <= w and y + b <= h
            area += a * b
            if cover is not None:
                cover[y:y + b, x:x + a] += 1
        assert area == w * h
        if cover is not None:
            assert (cover == 1).all()
    assert len(tile_grid(16384, 16384, 2048, 2048)) == 64
    with pytest.raises(ValueError):
        tile_grid(4, 4, 0, 1)


def test_groups_partition_the_tiles():
    for n_tiles, n_groups in ((64, 8), (7, 3), (2, 4), (1, 1)):
        seen = sorted(t for g in range(n_groups) for t in tiles_of_group(n_tiles, n_groups, g))
        assert seen == list(range(n_tiles))
    with pytest.raises(ValueError):
        tiles_of_group(4, 2, 2)


def test_image_tiles_round_trip(tmp_path):
    path = str(tmp_path / "t" / "img.ppm")
    with Image(path, 7, 5) as img:
        px = img.pixels()
        px[...] = np.arange(7 * 5 * 3, dtype=np.uint8).reshape(5, 7, 3)
        t = img.read_tile(2, 1, 4, 3)
        assert np.array_equal(t, px[1:4, 2:6])
        img.write_tile(3, 2, np.full((2, 3, 3), 255, np.uint8))
        assert (px[2:4, 3:6] == 255).all() and px[1, 3, 0] == np.arange(105, dtype=np.uint8).reshape(5, 7, 3)[1, 3, 0]
        with pytest.raises(ValueError):
            img.read_tile(5, 0, 3, 1)
        with pytest.raises(ValueError):
            img.write_tile(0, 4, np.zeros((2, 1, 3), np.uint8))
        img.flush()
    raw = open(path, "rb").read()
    assert raw.startswith(b"P6\n7 5 255\n") and len(raw) == len(b"P6\n7 5 255\n") + 105


class OracleTileBackend:
    """TileBackend over the CPU oracle (test stand-in for GpuTileBackend)."""

    def __init__(self, scene, pooled=False):
        self.scene = scene
        self.made = []  # buffers handed to the tile loop's pool
        if pooled:
            self.pixel_buffer = self._pixel_buffer

    def _pixel_buffer(self, h, w):
        self.made.append(np.full((h, w, 3), 0xCD, dtype=np.uint8))  # stale bytes: every use must overwrite them
        return self.made[-1]

    def render_tile_pass(self, params, old, out=None):
        from oracle import binding as O
        rgb8, _ = O.render(self.scene, params, None if old is None else np.ascontiguousarray(old))
        if out is not None:
            out[...] = rgb8
            return out
        return rgb8

    def render_sums(self, params, begin, count):
        from oracle import binding as O
        return torch.from_numpy(O.render_sums(self.scene, params, begin, count).reshape(-1).copy())

    def zeros(self, params):
        h, w = params.tile_shape()
        return torch.zeros(h * w * 3, dtype=torch.float32)

    def resolve(self, params, sums, n_samples, running, k, old, out=None):
        from oracle import binding as O
        h, w = params.tile_shape()
        rgb8 = O.resolve(params, sums.numpy().reshape(h, w, 3), n_samples, running, k,
                         None if old is None else np.ascontiguousarray(old))
        if out is not None:
            out[...] = rgb8
            return out
        return rgb8

    # the exact hand-over inside a group: per-sample radiance kept until the fold
    def sums(self, params):
        h, w = params.tile_shape()
        if getattr(self, "_sums", None) is None or self._sums.numel() != h * w * 3:
            self._sums = torch.zeros(h * w * 3, dtype=torch.float32)
        return self._sums

    def reserve(self, params, count):
        return min(count, 1)

    def trace(self, params, begin, count):
        from oracle import binding as O
        n = min(count, 1)  # one sample in flight: a pass of several samples per member takes several rounds
        self._pending = [torch.from_numpy(O.render_sums(self.scene, params, begin + j, 1).reshape(-1).copy()) for j in range(n)]
        return n

    def fold(self, params, carry_on):
        t = self.sums(params)
        samples = list(self._pending)
        acc = t.clone() if carry_on else samples.pop(0)
        for x in samples:
            acc = acc + x
        t.copy_(acc)
        return t


def _scene(spp, incremental=None, cont=None):
    from tests import scenes
    s = scenes.materials(w=40, h=28, spp=spp)
    p = s.params
    p.incremental = incremental or 0
    p.has_continue = 0 if cont is None else 1
    p.continue_sampling = cont or 0
    return s


@pytest.mark.parametrize("incremental", [None, 2])
def test_tiled_render_equals_untiled(tmp_path, incremental):
    from oracle import binding as O
    scene = _scene(4, incremental)
    want, _ = O.render(scene, scene.params)
    with Image(str(tmp_path / "a.ppm"), 40, 28) as img:
        render_tiled(OracleTileBackend(scene), img, scene.params, 16, 12)
        assert np.array_equal(img.pixels(), want)


def test_tiled_continue_resumes_from_the_file(tmp_path):
    from oracle import binding as O
    path = str(tmp_path / "c.ppm")
    first = _scene(4, incremental=2)
    done = []
    with Image(path, 40, 28) as img:
        render_tiled(OracleTileBackend(first), img, first.params, 13, 28, progress=done.append)
    assert done == [2, 4]
    second = _scene(2, incremental=2, cont=2)  # one more pass, numbered 2
    with Image(path, 40, 28) as img:               # reopened: the bytes are the checkpoint
        render_tiled(OracleTileBackend(second), img, second.params, 40, 5, progress=done.append)
        got = img.pixels().copy()
    assert done == [2, 4, 4]  # lib.rs:238-246 prints continue + passes done * samples per pass
    want, _ = O.render(first, first.params)
    want, _ = O.render(second, second.params, want)
    assert np.array_equal(got, want)
    assert passes(second.params) == (2, 1, True, 2)


@pytest.mark.parametrize("incremental", [None, 2])
def test_tile_loop_with_a_pool_of_backend_buffers(tmp_path, incremental):
    """A backend that offers pixel buffers (page-locked ones on a GPU): the loop keeps at most three
    going round, tiles of different sizes share them, and the image is the same."""
    from oracle import binding as O
    scene = _scene(4, incremental=incremental)
    want, _ = O.render(scene, scene.params)
    backend = OracleTileBackend(scene, pooled=True)
    with Image(str(tmp_path / "p.ppm"), 40, 28) as img:
        render_tiled(backend, img, scene.params, 16, 12)  # 3 x 3 tiles, the last column and row smaller
        assert np.array_equal(img.pixels(), want)
    passes_run = 2 if incremental else 1
    assert 1 <= len(backend.made) <= 3 * passes_run and all(b.shape == (12, 16, 3) for b in backend.made)


def _worker(rank, world, port, group_size, path, exact=False):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    scene = _scene(4, incremental=2)
    n_groups = world // group_size
    groups = [dist.new_group(list(range(g * group_size, (g + 1) * group_size))) for g in range(n_groups)]
    mine = groups[rank // group_size]
    root = (rank // group_size) * group_size

    def agree_min(n):
        t = torch.tensor([n], dtype=torch.int64)
        dist.all_reduce(t, op=dist.ReduceOp.MIN, group=mine)
        return int(t.item())

    with Image(path, 40, 28) as img:
        render_tiled(OracleTileBackend(scene), img, scene.params, 16, 16, world=world, rank=rank, group_size=group_size,
                     reduce_in_group=(lambda t: dist.reduce(t, dst=root, op=dist.ReduceOp.SUM, group=mine))
                     if group_size > 1 and not exact else None,
                     exact_in_group=((lambda t, m: dist.send(t, root + m)), (lambda t, m: dist.recv(t, root + m)), agree_min) if exact else None,
                     end_of_pass=dist.barrier)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world, group_size, exact", [(2, 1, False), (2, 2, False), (3, 1, False), (2, 2, True), (3, 3, True)])
def test_tiles_across_processes(tmp_path, world, group_size, exact):
    from oracle import binding as O
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    path = str(tmp_path / "m.ppm")
    mp.spawn(_worker, args=(world, port, group_size, path, exact), nprocs=world, join=True)
    with Image(path, 40, 28) as img:
        got = img.pixels().copy()
    scene = _scene(4, incremental=2)
    want, _ = O.render(scene, scene.params)
    if group_size == 1 or exact:
        assert np.array_equal(got, want)  # tile partition, or sums handed over in sample order: no arithmetic differs
    else:
        # sample partition inside the group: partial sums are added in another
        # association, which can move a truncated byte by one
        assert np.abs(got.astype(int) - want.astype(int)).max() <= 1
