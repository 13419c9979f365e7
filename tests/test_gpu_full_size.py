"""The headline workload at its full size (the generated breakfast-room-class
scene at 1920x1080, 60 bounces) checked through properties that do not need
the CPU oracle to render the whole frame: the GPU's full-frame result must
contain, bit for bit, what the oracle computes for tiles of that frame; primary
hits of the whole frame must equal the oracle's; and the result must not depend
on how the frame is cut into tiles or sample batches."""
import os

import numpy as np
import pytest

from oracle import binding as O
from raytracer_amd import _ffi as F
from raytracer_amd import assets, parse

pytestmark = pytest.mark.gpu

W, H, BOUNCES, SPP = 1920, 1080, 60, 2


@pytest.fixture(scope="module")
def room(tmp_path_factory):
    root = str(tmp_path_factory.mktemp("room_full"))
    n = assets.write_room(root, "room", detail=4)
    assert n > 250000
    return parse(assets.room_scene_text("room", W, H, SPP, BOUNCES), root)


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def test_full_frame_contains_oracle_tiles(renderer, room):
    renderer.upload(room)
    renderer.set_option(F.RT_OPT_SAMPLES_IN_FLIGHT, 0)
    renderer.reset_stats()
    full = renderer.render_sums(room.params, 0, SPP)
    st = renderer.stats()
    assert st["paths"] == W * H * SPP and st["segments"] > 20 * st["paths"]
    assert np.isfinite(full).all() and (full >= 0).all()
    rng = np.random.default_rng(8)
    for _ in range(5):
        tw, th = 96, 54
        x0, y0 = int(rng.integers(0, W - tw)), int(rng.integers(0, H - th))
        p = room.params.copy()
        p.tile_x, p.tile_y, p.tile_w, p.tile_h = x0, y0, tw, th
        want = O.render_sums(room, p, 0, SPP, threads=8)
        assert np.array_equal(bits(full[y0:y0 + th, x0:x0 + tw]), bits(want))
        # the same tile rendered on its own by the GPU
        assert np.array_equal(bits(renderer.render_sums(p, 0, SPP)), bits(want))


def test_full_frame_primary_hits_equal_oracle(renderer, room):
    renderer.upload(room)
    gpu = renderer.trace_primary(room.params, 0, True)
    # the oracle traces the frame in horizontal bands (threads are per call)
    for y0 in range(0, H, 270):
        p = room.params.copy()
        p.tile_x, p.tile_y, p.tile_w, p.tile_h = 0, y0, W, 270
        cpu = O.trace_primary(room, p, 0, True)
        band = gpu[y0:y0 + 270]
        assert np.array_equal(band["kind"], cpu["kind"]) and np.array_equal(band["prim"], cpu["prim"])
        assert np.array_equal(bits(band["t"]), bits(cpu["t"]))
    assert (gpu["kind"] == F.RT_SHAPE_TRIANGLE).mean() > 0.8


def test_result_is_independent_of_batching(renderer, room):
    renderer.upload(room)
    p = room.params.copy()
    p.tile_x, p.tile_y, p.tile_w, p.tile_h = 600, 400, 512, 256
    ref = None
    for in_flight, slots in ((1, 8), (2, 4), (0, 16)):
        renderer.set_option(F.RT_OPT_SAMPLES_IN_FLIGHT, in_flight)
        renderer.set_option(F.RT_OPT_TRACE_LDS_SLOTS, slots)
        got = renderer.render_sums(p, 4, 3)
        ref = got if ref is None else ref
        assert np.array_equal(bits(got), bits(ref))
    renderer.set_option(F.RT_OPT_SAMPLES_IN_FLIGHT, 0)
    renderer.set_option(F.RT_OPT_TRACE_LDS_SLOTS, 8)


def test_result_is_independent_of_the_number_of_chains(renderer, room):
    # A batch this large is split into independent chains on separate streams whose kernels run
    # side by side; enough paths (33 M) that they really do overlap for most of the render, so
    # that anything the chains wrongly share (the stack overflow area once was) shows.
    renderer.upload(room)
    renderer.set_option(F.RT_OPT_SAMPLES_IN_FLIGHT, 0)
    ref = None
    for chains in (1, 2, 2, 3, 4):
        renderer.set_option(F.RT_OPT_CHAINS, chains)
        renderer.reset_stats()
        got = renderer.render_sums(room.params, 0, 16)
        st = renderer.stats()
        if ref is None:
            ref, ref_segments = got, st["segments"]
        assert np.array_equal(bits(got), bits(ref)), f"{chains} chains: {np.count_nonzero(bits(got) != bits(ref))} values differ"
        assert st["segments"] == ref_segments and st["paths"] == W * H * 16
    renderer.set_option(F.RT_OPT_CHAINS, 2)
    with pytest.raises(Exception):
        renderer.set_option(F.RT_OPT_CHAINS, 5)
    # and against the oracle: tiles of the 16-sample frame rendered with two chains
    got = renderer.render_sums(room.params, 0, 16)
    for x0, y0 in ((100, 900), (1500, 300)):
        p = room.params.copy()
        p.tile_x, p.tile_y, p.tile_w, p.tile_h = x0, y0, 64, 36
        assert np.array_equal(bits(got[y0:y0 + 36, x0:x0 + 64]), bits(O.render_sums(room, p, 0, 16, threads=8)))


def test_one_large_batch_equals_many_small_ones(renderer, room):
    # 128 samples as one batch of 2.7e8 paths in two chains (the default for a frame this size)
    # against sixteen batches of 8 samples in one chain: the same left fold over the same samples
    renderer.upload(room)
    renderer.set_option(F.RT_OPT_SAMPLES_IN_FLIGHT, 0)
    renderer.set_option(F.RT_OPT_CHAINS, 2)
    renderer.reset_stats()
    big = renderer.render_sums(room.params, 7, 128)
    big_stats = renderer.stats()
    renderer.set_option(F.RT_OPT_SAMPLES_IN_FLIGHT, 8)
    renderer.set_option(F.RT_OPT_CHAINS, 1)
    renderer.reset_stats()
    small = renderer.render_sums(room.params, 7, 128)
    small_stats = renderer.stats()
    renderer.set_option(F.RT_OPT_SAMPLES_IN_FLIGHT, 0)
    renderer.set_option(F.RT_OPT_CHAINS, 2)
    assert np.array_equal(bits(big), bits(small))
    assert big_stats["segments"] == small_stats["segments"] and big_stats["hits"] == small_stats["hits"]
    assert big_stats["paths"] == W * H * 128


# ---------------------------------------------------------------- island-tree class

@pytest.fixture(scope="module")
def tree(tmp_path_factory):
    root = str(tmp_path_factory.mktemp("tree_full"))
    n = assets.write_tree(root, "tree", n_leaves=250_000)
    assert n > 500_000
    return parse(assets.tree_scene_text("tree", 1000, 1000, 2, 50), root)


def test_deep_tree_full_frame_contains_oracle_tiles(renderer, tree):
    """Half a million thin overlapping quads: long leaves, a deep tree, 50 bounces."""
    renderer.upload(tree)
    renderer.set_option(F.RT_OPT_SAMPLES_IN_FLIGHT, 0)
    renderer.set_option(F.RT_OPT_TRACE_LDS_SLOTS, 6)
    gpu_hits = renderer.trace_primary(tree.params, 0, True)
    full = renderer.render_sums(tree.params, 0, 2)
    assert np.isfinite(full).all()
    for x0, y0 in ((450, 380), (300, 520), (640, 300)):
        p = tree.params.copy()
        p.tile_x, p.tile_y, p.tile_w, p.tile_h = x0, y0, 64, 48
        want = O.render_sums(tree, p, 0, 2, threads=8)
        assert np.array_equal(bits(full[y0:y0 + 48, x0:x0 + 64]), bits(want))
        cpu_hits = O.trace_primary(tree, p, 0, True)
        band = gpu_hits[y0:y0 + 48, x0:x0 + 64]
        assert np.array_equal(band["prim"], cpu_hits["prim"]) and np.array_equal(bits(band["t"]), bits(cpu_hits["t"]))
    # enough paths for two chains (deep stacks: each chain's overflow area is in heavy use); a batch
    # of this size would otherwise take the one-launch loop, which is compared with both
    fused = renderer.render_sums(tree.params, 2, 12)
    renderer.set_option(F.RT_OPT_FUSED_LOOP, 0)
    try:
        renderer.set_option(F.RT_OPT_CHAINS, 1)
        one = renderer.render_sums(tree.params, 2, 12)
        renderer.set_option(F.RT_OPT_CHAINS, 2)
        two = renderer.render_sums(tree.params, 2, 12)
    finally:
        renderer.set_option(F.RT_OPT_FUSED_LOOP, 2)
    assert np.array_equal(bits(one), bits(two)) and np.array_equal(bits(one), bits(fused))


# ---------------------------------------------------------------- 16K x 16K image

def test_16k_image_through_the_mapped_file_with_continue(renderer, room, tmp_path):
    """The 16384 x 16384 frame (805 MB of pixels) rendered tile by tile into the
    memory-mapped file, one pass, then one more pass resumed from the file with
    `continue`; spot tiles are compared with the CPU oracle run the same way."""
    from raytracer_amd.tiles import GpuTileBackend, Image, render_tiled

    W16 = 16384
    renderer.upload(room)
    renderer.set_option(F.RT_OPT_SAMPLES_IN_FLIGHT, 0)
    p1 = room.params.copy()
    p1.width = p1.height = W16
    p1.top_edge[:] = [1.92, 0.0, 0.0]
    p1.left_edge[:] = [0.0, -1.92, 0.0]  # square window
    p1.samples_per_pixel, p1.incremental, p1.max_bounces = 1, 1, 6
    p2 = p1.copy()
    p2.has_continue, p2.continue_sampling = 1, 1
    path = str(tmp_path / "big.ppm")
    renderer.reset_stats()
    with Image(path, W16, W16) as img:
        render_tiled(GpuTileBackend(renderer), img, p1, 4096, 4096)
    assert os.path.getsize(path) == len(b"P6\n16384 16384 255\n") + W16 * W16 * 3
    with Image(path, W16, W16) as img:
        render_tiled(GpuTileBackend(renderer), img, p2, 4096, 4096)
        px = img.pixels()
        assert renderer.stats()["paths"] == 2 * W16 * W16
        assert px[::997, ::991].max() > 0
        for x0, y0 in ((4090, 4090), (9000, 12000), (16300, 100)):  # the first straddles four tiles
            t1, t2 = p1.copy(), p2.copy()
            for t in (t1, t2):
                t.tile_x, t.tile_y, t.tile_w, t.tile_h = x0, y0, 48, 32
            want, _ = O.render(room, t1, threads=8)
            want, _ = O.render(room, t2, want, threads=8)
            assert np.array_equal(px[y0:y0 + 32, x0:x0 + 48], want)
    os.remove(path)
