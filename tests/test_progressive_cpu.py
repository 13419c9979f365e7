"""Host logic of raytracer_amd/progressive.py without a GPU: the side file and the
pass / tile / restart bookkeeping, against a stand-in renderer whose sample k of
pixel (x, y) is a known number."""
import numpy as np
import pytest

from raytracer_amd import _ffi as F
from raytracer_amd.progressive import Accumulator, render_progressive
from raytracer_amd.tiles import Image


class FakeRenderer:
    """Sample k of pixel (x, y) is x + 100 y + 10000 (k + 1), in all three channels."""

    def __init__(self):
        self.calls = []

    def render_sums_onto(self, p, begin, count, sums):
        h, w = p.tile_shape()
        assert sums.shape == (h, w, 3) and sums.dtype == np.float32 and sums.flags.c_contiguous
        self.calls.append((p.tile_x, p.tile_y, w, h, begin, count))
        ys, xs = np.mgrid[p.tile_y:p.tile_y + h, p.tile_x:p.tile_x + w]
        if begin == 0:
            sums[:] = 0.0
        for k in range(begin, begin + count):
            sums += (xs + 100 * ys + 10000 * (k + 1))[..., None].astype(np.float32)
        return sums

    def resolve(self, p, sums, n):
        return np.clip(sums / n / 40000.0 * 255.0, 0, 255).astype(np.uint8)


def params(w, h, seed=7):
    p = F.RtRenderParams()
    p.width, p.height, p.samples_per_pixel, p.max_bounces = w, h, 1, 4
    p.rng_mode, p.rng_seed = F.RT_RNG_SUBSTREAM, seed
    return p


def expected_sums(w, h, n):
    ys, xs = np.mgrid[0:h, 0:w]
    total = np.zeros((h, w), dtype=np.float32)
    for k in range(n):
        total += (xs + 100 * ys + 10000 * (k + 1)).astype(np.float32)
    return total


@pytest.mark.parametrize("tile", [None, 4])
def test_passes_tiles_and_restart(tmp_path, tile):
    w, h = 10, 7
    path, side = str(tmp_path / "p.ppm"), str(tmp_path / "p.acc")
    r = FakeRenderer()
    seen = []
    with Image(path, w, h) as img:
        assert render_progressive(r, img, params(w, h), 2, 3, side, tile=tile, progress=lambda k, n: seen.append((k, n))) == 6
    assert seen == [(1, 3), (2, 6)]
    n_tiles = 1 if tile is None else 3 * 2
    assert len(r.calls) == 2 * n_tiles and {c[4] for c in r.calls} == {0, 3} and {c[5] for c in r.calls} == {3}
    with Image(path, w, h) as img:  # a restart carries on at sample 6
        assert render_progressive(r, img, params(w, h), 1, 2, side, tile=tile) == 8
        acc = Accumulator(side, w, h, 7)
        assert acc.samples == 8 and not acc.in_progress
        assert np.array_equal(acc.sums[..., 0], expected_sums(w, h, 8))
        assert np.array_equal(img.pixels(), r.resolve(None, acc.sums, 8))
        acc.close()
        assert render_progressive(r, img, params(w, h), 0, 1, side) == 8  # no pass: only reports


def test_side_file_is_started_over_for_another_frame_and_refused_when_cut_off(tmp_path):
    side = str(tmp_path / "q.acc")
    a = Accumulator(side, 8, 4, 1)
    a.sums[:] = 5.0
    a.mark(9, False)
    a.close()
    b = Accumulator(side, 8, 4, 1)  # same frame: kept
    assert b.samples == 9 and float(b.sums[0, 0, 0]) == 5.0
    b.close()
    for shape_seed in ((8, 5, 1), (8, 4, 2)):  # other size or generator seed: fresh
        c = Accumulator(side, *shape_seed)
        assert c.samples == 0 and not c.sums.any()
        c.mark(3, True)
        c.close()
    r = FakeRenderer()
    with Image(str(tmp_path / "q.ppm"), 8, 4) as img:
        with pytest.raises(RuntimeError):
            render_progressive(r, img, params(8, 4, seed=2), 1, 1, side)
        with pytest.raises(ValueError):
            render_progressive(r, img, params(9, 4), 1, 1, side)
        with pytest.raises(ValueError):
            render_progressive(r, img, params(8, 4), 1, 0, side)
    assert r.calls == []
