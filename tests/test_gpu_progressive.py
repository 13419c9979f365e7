"""Progressive rendering on the f32 side file (raytracer_amd/progressive.py,
rt_render_sums_onto): whatever the split into passes, tiles and restarts, the
image equals the one-shot render of the same sample count — the oracle's."""
import os
import subprocess

import numpy as np
import pytest

from oracle import binding as O
from raytracer_amd import _ffi as F
from raytracer_amd import assets
from raytracer_amd.progressive import Accumulator, render_progressive
from raytracer_amd.tiles import Image

from . import scenes

pytestmark = pytest.mark.gpu


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def test_fold_carried_on_equals_one_fold(renderer):
    s = scenes.materials(w=40, h=24, spp=7, bounces=12)
    renderer.upload(s)
    whole = O.render_sums(s, s.params, 0, 7, threads=4)
    acc = np.zeros((24, 40, 3), dtype=np.float32)
    acc[:] = np.nan  # ignored when the fold starts at sample 0
    for begin, count in ((0, 3), (3, 1), (4, 3)):
        renderer.render_sums_onto(s.params, begin, count, acc)
    assert np.array_equal(bits(acc), bits(whole))
    assert np.array_equal(bits(renderer.render_sums(s.params, 0, 7)), bits(whole))
    with pytest.raises(ValueError):
        renderer.render_sums_onto(s.params, 0, 1, np.zeros((24, 40, 3), dtype=np.float64))


@pytest.mark.parametrize("tile", [None, 16])
def test_progressive_image_equals_the_one_shot_render(renderer, asset_root, tmp_path, tile):
    s = scenes.room(asset_root, w=48, h=30, spp=6, bounces=10)
    renderer.upload(s)
    path, side = str(tmp_path / "img.ppm"), str(tmp_path / "img.ppm.acc")
    stages = []
    with Image(path, 48, 30) as img:
        n = render_progressive(renderer, img, s.params, 2, 2, side, tile=tile,
                               progress=lambda k, total: stages.append((k, total, img.pixels().copy())))
        assert n == 4 and [(k, t) for k, t, _ in stages] == [(1, 2), (2, 4)]
    # a second run (a restart) carries on from the side file
    with Image(path, 48, 30) as img:
        assert render_progressive(renderer, img, s.params, 1, 2, side, tile=tile) == 6
        final = img.pixels().copy()
    for n_samples, got in ((2, stages[0][2]), (4, stages[1][2]), (6, final)):
        p = s.params.copy()
        p.samples_per_pixel = n_samples
        want, _ = O.render(s, p, threads=4)
        assert np.array_equal(got, want), n_samples
    # the reference's byte-averaging passes do not reach that image (they re-truncate every pass)
    p = s.params.copy()
    p.samples_per_pixel, p.incremental = 6, 2
    averaged, _ = O.render(s, p, threads=4)
    assert not np.array_equal(averaged, final)


def test_side_file_of_another_image_or_a_cut_off_pass(renderer, tmp_path):
    s = scenes.sphere_plane(w=16, h=8, spp=1)
    renderer.upload(s)
    path, side = str(tmp_path / "a.ppm"), str(tmp_path / "a.acc")
    with Image(path, 16, 8) as img:
        assert render_progressive(renderer, img, s.params, 1, 1, side) == 1
        # another seed: the file is started over rather than mixed
        p = s.params.copy()
        p.rng_seed = 12345
        assert render_progressive(renderer, img, p, 1, 1, side) == 1
        acc = Accumulator(side, 16, 8, 12345)
        acc.mark(acc.samples, True)  # as left behind by a run that died mid-pass
        acc.close()
        with pytest.raises(RuntimeError):
            render_progressive(renderer, img, p, 1, 1, side)
    assert os.path.getsize(side) == 32 + 16 * 8 * 12


def test_command_line_renderer_with_an_accumulator_file(asset_root, tmp_path):
    """rt_render <scene> <out> <gpu> <tile> <acc>: two runs of 4 samples in passes of 2, tiled,
    leave the one-shot image of 8 samples and a side file the Python reader agrees with."""
    exe = os.path.join(F.LIB_DIR, "rt_render")
    scene_file = tmp_path / "scene"
    scene_file.write_text(assets.room_scene_text("room", 80, 50, 4, 8, incremental=2))
    out, side = str(tmp_path / "a.ppm"), str(tmp_path / "a.acc")
    for _ in range(2):
        subprocess.run([exe, str(scene_file), out, "0", "32", side], cwd=asset_root, check=True, env=dict(os.environ), timeout=120)
    scene = scenes.room(asset_root, w=80, h=50, spp=8, bounces=8)
    want, _ = O.render(scene, scene.params, threads=4)
    got = open(out, "rb").read()
    assert got[len(b"P6\n80 50 255\n"):] == want.tobytes()
    acc = Accumulator(side, 80, 50, scene.params.rng_seed)
    assert acc.samples == 8 and not acc.in_progress
    whole = O.render_sums(scene, scene.params, 0, 8, threads=4)
    assert np.array_equal(bits(acc.sums), bits(whole))
    acc.mark(8, True)
    acc.close()
    cut = subprocess.run([exe, str(scene_file), out, "0", "32", side], cwd=asset_root, env=dict(os.environ), timeout=120,
                         capture_output=True)
    assert cut.returncode == 1 and b"cut off" in cut.stderr
