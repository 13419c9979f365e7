"""The headline workload at its full size (the generated breakfast-room-class
scene at 1920x1080, 60 bounces) checked through properties that do not need
the CPU oracle to render the whole frame: the GPU's full-frame result must
contain, bit for bit, what the oracle computes for tiles of that frame; primary
hits of the whole frame must equal the oracle's; and the result must not depend
on how the frame is cut into tiles or sample batches."""
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
