"""GPU parity on the inputs the reference itself holds: boombox_4k.obj (two
`usemtl` groups, `vt` + `vn`, two 4096 x 4096 `map_Kd` textures whose P6 files are
generated because the checkout lacks them), hollow knight.obj (Cyrillic material
names, solid `Kd`), smooth_uv_sphere.obj (no mtllib), with the screen and camera
of /root/reference/scenes/boombox:2-3.  Every value (primary hit distance bits,
family and primitive ids, per-pixel radiance bits, stored bytes, traversal
counters) must equal the CPU oracle's: tolerance 0."""
import numpy as np
import pytest

from oracle import binding as O
from raytracer_amd import _ffi as F
from raytracer_amd import assets, parse

from .test_gpu_parity import assert_hits_equal, bits, check_scene

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def staged(tmp_path_factory):
    root = str(tmp_path_factory.mktemp("reference_assets_gpu"))
    assets.stage_reference_assets(root, texture_size=4096)  # the size boombox_4k.mtl's file names promise
    return root


def test_boombox_with_the_camera_of_the_reference_scene(renderer, staged):
    s = parse(assets.reference_scene_text("boombox_4k", 160, 160, 3, 10), staged)
    assert len(s.triangles) == 10168 and s.texels.size == 2 * 4096 * 4096 * 3
    check_scene(renderer, s)
    hits = renderer.trace_primary(s.params, 0, True)
    tri = hits["kind"] == F.RT_SHAPE_TRIANGLE
    assert 0.02 < tri.mean() < 0.5
    # both `usemtl` groups are seen from this camera
    assert set(np.unique(s.triangles["material"][hits["prim"][tri]]).tolist()) == {0, 1}


def test_large_batches_of_a_frame_of_mostly_misses_through_whichever_loop_measures_faster(renderer, staged):
    """A batch of 2^26 paths of the reference's boombox frame: the first two large batches after an upload are each
    rendered as two halves, one through each bounce loop, and the faster loop is kept (here the one launch: 19 of 20 rays
    miss).  Whatever is chosen, the sums are those of the per-bounce loop and of the one-launch loop, bit for bit,
    on the first render (which measures) and on the second (which uses the choice), and the oracle's on a tile."""
    s = parse(assets.reference_scene_text("boombox_4k", 1024, 1024, 64, 10), staged)
    renderer.upload(s)
    first = renderer.render_sums(s.params, 0, 64)   # measures: halves of 2^25 paths through either loop
    second = renderer.render_sums(s.params, 0, 64)  # measures again, the loops in the other order
    third = renderer.render_sums(s.params, 0, 64)   # renders with the choice
    assert np.array_equal(bits(first), bits(second)) and np.array_equal(bits(first), bits(third))
    for forced in (0, 1):
        renderer.set_option(F.RT_OPT_FUSED_LOOP, forced)
        try:
            assert np.array_equal(bits(renderer.render_sums(s.params, 0, 64)), bits(first)), forced
        finally:
            renderer.set_option(F.RT_OPT_FUSED_LOOP, 2)
    t = s.params.copy()
    t.tile_x, t.tile_y, t.tile_w, t.tile_h = 448, 480, 128, 64  # across the mesh
    cpu = O.render_sums(s, t, 0, 64, threads=8)
    assert np.array_equal(bits(first[480:544, 448:576]), bits(cpu))


def test_boombox_close_up_textured_hits_dominate(renderer, staged):
    # the same camera line, the window narrowed to the mesh's bounding box (x +-0.36, y 0..0.47)
    text = assets.reference_scene_text("boombox_4k", 128, 80, 4, 10).replace(
        "screen(-1.6 1 -0.5, 3.2 0 0, 0 -2 0", "screen(-0.4 0.5 -0.5, 0.8 0 0, 0 -0.5 0")
    s = parse(text, staged)
    assert list(s.params.top_left) == [np.float32(-0.4), 0.5, -0.5]
    check_scene(renderer, s)
    hits = renderer.trace_primary(s.params, 0, False)
    assert (hits["kind"] == F.RT_SHAPE_TRIANGLE).mean() > 0.4


def test_hollow_knight(renderer, staged):
    s = parse(assets.reference_scene_text("hollow knight", 96, 96, 3, 10), staged)
    assert len(s.triangles) == 18352 and len(s.planes) == 1
    check_scene(renderer, s)


def test_smooth_uv_sphere(renderer, staged):
    s = parse(assets.reference_scene_text("smooth_uv_sphere", 96, 64, 4, 10), staged)
    assert len(s.triangles) == 960
    check_scene(renderer, s)


def test_random_rays_against_the_reference_meshes(renderer, staged):
    rng = np.random.default_rng(17)
    for name, lo, hi in (("boombox_4k", (-0.5, -0.1, -0.3), (0.5, 0.6, 0.3)),
                         ("hollow knight", (-2.5, -0.5, -2.0), (2.5, 5.0, 1.5))):
        s = parse(assets.reference_scene_text(name, 16, 16, 1, 4), staged)
        renderer.upload(s)
        n = 30000
        o = rng.uniform(lo, hi, size=(n, 3)).astype(np.float32)
        d = rng.normal(size=(n, 3)).astype(np.float32)
        d /= np.linalg.norm(d, axis=1, keepdims=True)
        d[::9, 2] = 0.0
        gpu, cpu = renderer.trace_rays(o, d), O.trace_rays(s, o, d)
        assert_hits_equal(gpu, cpu)
        assert (gpu["kind"] == F.RT_SHAPE_TRIANGLE).mean() > 0.05


def test_gpu_built_tree_of_the_reference_meshes_equals_the_host_builders(renderer, staged):
    for name in ("boombox_4k", "hollow knight", "smooth_uv_sphere"):
        text = assets.reference_scene_text(name, 16, 16, 1, 4)
        host, dev = parse(text, staged), parse(text, staged, renderer=renderer)
        assert np.array_equal(host.triangles, dev.triangles)
        assert np.array_equal(host.triangle_nodes["a"], dev.triangle_nodes["a"])
        assert np.array_equal(host.triangle_nodes["b"], dev.triangle_nodes["b"])
        assert np.array_equal(host.triangle_nodes["min"], dev.triangle_nodes["min"])  # as numbers: a zero may differ in sign
        assert np.array_equal(host.triangle_nodes["max"], dev.triangle_nodes["max"])
