"""GPU parity: every result of librt_mi355.so, reached through its C ABI, is
compared bit for bit with the CPU oracle on the same inputs.

The bar is exact equality — hit distances, primitive ids, per-pixel radiance
sums (float bit patterns) and stored bytes — because the kernels repeat the
oracle's IEEE binary32 operation order (no FMA, true division) and samples of a
pixel are folded in sample order.
"""
import numpy as np
import pytest

from oracle import binding as O
from raytracer_amd import _ffi as F
from raytracer_amd import parse
from raytracer_amd.renderer import RtError

from . import scenes

pytestmark = pytest.mark.gpu


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def assert_hits_equal(gpu, cpu):
    assert np.array_equal(gpu["kind"], cpu["kind"]), "shape family differs"
    assert np.array_equal(gpu["prim"], cpu["prim"]), "primitive id differs"
    assert np.array_equal(bits(gpu["t"]), bits(cpu["t"])), "hit distance differs"


def check_scene(renderer, scene, spp=None, traversal_counters=True):
    p = scene.params
    spp = spp or p.samples_per_pixel
    renderer.upload(scene)
    # primary hits, with and without jitter
    for jitter in (False, True):
        assert_hits_equal(renderer.trace_primary(p, 1, jitter), O.trace_primary(scene, p, 1, jitter))
    # radiance sums: exact, and the same work counted
    renderer.set_option(F.RT_OPT_COUNT_TRAVERSAL, 1 if traversal_counters else 0)
    renderer.reset_stats()
    c = O.OracleCounters()
    cpu = O.render_sums(scene, p, 0, spp, threads=4, counters=c)
    gpu = renderer.render_sums(p, 0, spp)
    renderer.set_option(F.RT_OPT_COUNT_TRAVERSAL, 0)
    assert np.array_equal(bits(gpu), bits(cpu)), f"radiance differs in {np.count_nonzero(bits(gpu) != bits(cpu))} values"
    s = renderer.stats()
    assert s["paths"] == c.paths and s["segments"] == c.segments and s["hits"] == c.hits
    if traversal_counters:
        assert s["branch_visits"] == c.branch_visits
        assert s["prim_tests"] == list(c.prim_tests)
    # stored image
    cpu_img, _ = O.render(scene, p, threads=4)
    gpu_img = renderer.render(p)
    assert np.array_equal(gpu_img, cpu_img)
    # a frame this small takes the one-launch bounce loop where the scene allows it; the
    # launch-per-bounce loop (what large batches use) must store the same bytes
    renderer.set_option(F.RT_OPT_FUSED_LOOP, 0)
    try:
        assert np.array_equal(renderer.render(p), cpu_img)
    finally:
        renderer.set_option(F.RT_OPT_FUSED_LOOP, 2)
    return gpu


def test_sphere_plane(renderer):
    check_scene(renderer, scenes.sphere_plane(w=64, h=36, spp=4))


def test_all_materials(renderer):
    check_scene(renderer, scenes.materials(w=80, h=50, spp=6, bounces=12))


def test_mesh_with_vertex_normals(renderer, asset_root):
    check_scene(renderer, scenes.blob(asset_root, w=48, h=48, spp=3))


def test_textured_room(renderer, asset_root):
    check_scene(renderer, scenes.room(asset_root, w=64, h=36, spp=3, bounces=20))


def test_textures_of_sizes_that_do_not_fill_the_device_tiles(renderer, tmp_path):
    # The device keeps texels in tiles of 8 x 4: widths and heights below, at and just past a
    # tile's, a single texel, and texture coordinates far outside [0, 1] (wrapped by
    # material.rs:197-206) so that every edge texel, first and last row included, is sampled.
    from raytracer_amd import assets

    rng = np.random.default_rng(12)
    sizes = [(1, 1), (2, 2), (7, 3), (8, 4), (9, 5), (13, 7), (16, 9), (31, 33)]
    d = tmp_path / "obj" / "quads"
    d.mkdir(parents=True)
    w = assets.ObjWriter()
    for k, (tw, th) in enumerate(sizes):
        assets.write_ppm(str(d / f"t{k}.ppm"), rng.integers(0, 256, size=(th, tw, 3), dtype=np.uint8))
        w.add_material(f"m{k}", map_kd=f"t{k}.ppm")
        x0, y0 = -2.0 + (k % 4), 0.1 + (k // 4) * 1.1
        verts = [(x0, y0, 0), (x0 + 0.95, y0, 0.2), (x0 + 0.95, y0 + 1.0, 0.2), (x0, y0 + 1.0, 0)]
        uvs = [(-1.3, -0.7), (2.6, -0.7), (2.6, 1.9), (-1.3, 1.9)]
        w.add(f"m{k}", verts, [(0, 1, 2), (0, 2, 3)], uvs=uvs)
    w.write(str(d / "quads.obj"), "quads.mtl")
    text = ("screen(-2.1 2.4 2, 4.2 0 0, 0 -2.4 0, 96, 56, 3, 5)\ncamera(0 1.2 6)\nspheres()\nplanes()\n"
            "obj((quads/quads))\ntriangles()\n")
    s = parse(text, str(tmp_path))
    assert len(s.triangles) == 16 and int((s.materials["color_kind"] == F.RT_COLOR_TEXTURE).sum()) == len(sizes)
    gpu = check_scene(renderer, s)
    assert len(np.unique(bits(gpu))) > 500  # the textures are seen, not one flat colour


def test_tree_of_overlapping_leaf_quads(renderer, asset_root):
    # island-tree class: fan-triangulated quads, long leaves, textured and solid materials
    s = scenes.tree(asset_root, w=56, h=56, spp=3, bounces=16)
    counts = s.triangle_nodes["b"][(s.triangle_nodes["b"] & F.RT_LEAF_BIT) != 0] & 0x7FFFFFFF
    assert counts.max() >= 15  # leaves too long for an inline count
    check_scene(renderer, s)


def test_large_sphere_and_plane_families_are_traversed(renderer):
    # more than 16 shapes per family: each goes through its own traversal launches
    s = scenes.many_spheres()
    assert len(s.sphere_nodes) > 20 and len(s.plane_nodes) == 1 and len(s.planes) == 20
    check_scene(renderer, s)
    hits = renderer.trace_primary(s.params, 0, True)
    assert {int(k) for k in np.unique(hits["kind"])} >= {F.RT_SHAPE_SPHERE, F.RT_SHAPE_PLANE}


def test_empty_scene_is_sky(renderer):
    s = scenes.empty()
    gpu = check_scene(renderer, s)
    assert np.all(gpu > 0)


def test_zero_bounces_is_black(renderer):
    s = scenes.sphere_plane(w=16, h=8, spp=2, bounces=0)
    gpu = check_scene(renderer, s, traversal_counters=False)
    assert np.all(gpu == 0)


def test_one_bounce_returns_throughput(renderer):
    # bounce-limit exhaustion keeps the accumulated colour (not black)
    gpu = check_scene(renderer, scenes.sphere_plane(w=32, h=18, spp=2, bounces=1))
    assert gpu.max() > 0


def test_deep_tree_spills_past_lds_stack(renderer):
    s = scenes.deep_chain(n=60, w=48, h=16)
    assert scenes.bvh_depth(s.triangle_nodes) > 20
    check_scene(renderer, s)
    # rays along the chain visit many levels
    o = np.tile(np.array([[-1.0, 0.0, -3.0]], np.float32), (64, 1))
    d = np.tile(np.array([[1.0, 0.0, 0.0]], np.float32), (64, 1))
    d[:, 1] = np.linspace(-0.1, 0.1, 64)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    assert_hits_equal(renderer.trace_rays(o, d), O.trace_rays(s, o, d))


def test_long_leaf_and_equal_distance_ties(renderer):
    s = scenes.stacked_duplicates(n=40, w=24, h=24)
    assert len(s.triangle_nodes) == 1
    check_scene(renderer, s)
    hits = renderer.trace_primary(s.params, 0, False)
    hit = hits["kind"] == F.RT_SHAPE_TRIANGLE
    assert hit.any() and np.all(hits["prim"][hit] == 0)  # lowest index wins an exact tie


def test_shared_edges_and_axis_aligned_rays(renderer):
    s = scenes.grid_wall(n=16, w=33, h=33)
    check_scene(renderer, s)
    # rays aimed exactly at grid vertices and edge midpoints, axis-aligned directions
    # (zero direction components give +-inf and NaN in the slab test)
    g = np.linspace(-4, 4, 33, dtype=np.float32)
    xx, yy = np.meshgrid(g, g)
    o = np.stack([xx.ravel(), yy.ravel(), np.full(xx.size, 4.0, np.float32)], -1)
    d = np.tile(np.array([[0.0, 0.0, -1.0]], np.float32), (o.shape[0], 1))
    assert_hits_equal(renderer.trace_rays(o, d), O.trace_rays(s, o, d))
    # rays lying in the wall's plane and rays starting on it
    o2 = np.stack([np.full(33, -6.0, np.float32), g, np.zeros(33, np.float32)], -1)
    d2 = np.tile(np.array([[1.0, 0.0, 0.0]], np.float32), (33, 1))
    assert_hits_equal(renderer.trace_rays(o2, d2), O.trace_rays(s, o2, d2))


def test_random_rays_against_room(renderer, asset_root):
    s = scenes.room(asset_root)
    renderer.upload(s)
    rng = np.random.default_rng(3)
    n = 20000
    o = rng.uniform([-2.9, 0.1, -3.9], [2.9, 3.1, 4.9], size=(n, 3)).astype(np.float32)
    d = rng.normal(size=(n, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    d[::7, 0] = 0.0  # some with exact zeros
    d[::11, 1] = 0.0
    assert_hits_equal(renderer.trace_rays(o, d), O.trace_rays(s, o, d))


def test_tile_equals_crop_of_full_frame(renderer):
    s = scenes.materials(w=40, h=30, spp=3)
    renderer.upload(s)
    full = renderer.render_sums(s.params, 0, 3)
    p = s.params.copy()
    p.tile_x, p.tile_y, p.tile_w, p.tile_h = 7, 5, 13, 11
    tile = renderer.render_sums(p, 0, 3)
    assert np.array_equal(bits(tile), bits(full[5:16, 7:20]))
    assert np.array_equal(bits(tile), bits(O.render_sums(s, p, 0, 3)))


def test_samples_in_flight_do_not_change_results(renderer, asset_root):
    s = scenes.room(asset_root, w=32, h=18, spp=5, bounces=10)
    renderer.upload(s)
    ref = None
    for in_flight in (1, 2, 5, 0):
        renderer.set_option(F.RT_OPT_SAMPLES_IN_FLIGHT, in_flight)
        got = renderer.render_sums(s.params, 2, 5)
        if ref is None:
            ref = got
        assert np.array_equal(bits(got), bits(ref))
    assert np.array_equal(bits(ref), bits(O.render_sums(s, s.params, 2, 5)))


def test_sample_ranges_are_independent_substreams(renderer):
    # sums over [0,6) == fold of [0,2) then [2,6) only up to rounding order, but each range
    # matches the oracle exactly and depends on nothing else
    s = scenes.sphere_plane(w=32, h=18, spp=6)
    renderer.upload(s)
    for begin, count in ((0, 2), (2, 4), (1000000, 3)):
        assert np.array_equal(bits(renderer.render_sums(s.params, begin, count)),
                              bits(O.render_sums(s, s.params, begin, count)))


def test_incremental_and_continue(renderer):
    # incremental(2): three passes of 2 samples, running average of stored bytes
    s = scenes.materials(w=40, h=24, spp=6, prefix="incremental(2)\n")
    renderer.upload(s)
    passes = []
    gpu = renderer.render(s.params, progress=passes.append)
    cpu, _ = O.render(s, s.params, threads=2)
    assert np.array_equal(gpu, cpu)
    assert passes == [2, 4, 6]
    # continue(3) resumes from the stored bytes with fresh sub-streams
    s2 = scenes.materials(w=40, h=24, spp=4, prefix="continue(3)\nincremental(2)\n")
    gpu2 = renderer.render(s2.params, rgb8=gpu.copy())
    cpu2, _ = O.render(s2, s2.params, rgb8=cpu.copy(), threads=2)
    assert np.array_equal(gpu2, cpu2)
    assert not np.array_equal(gpu2, gpu)


@pytest.mark.parametrize("together", [0, 1, 3, 64])
def test_short_passes_traced_together_store_the_same_bytes(renderer, asset_root, together):
    """rt_render traces the paths of several short passes in one batch and finishes the passes in order
    (RT_OPT_PASSES_IN_FLIGHT: automatic, one at a time, three at a time with a remainder, more than there
    are): the bytes are the oracle's pass-by-pass bytes, progress is reported per pass, and a tile and a
    resumed render behave the same."""
    s = scenes.room(asset_root, w=96, h=54, spp=7, bounces=10, incremental=1)
    renderer.upload(s)
    cpu, _ = O.render(s, s.params, threads=4)
    renderer.set_option(F.RT_OPT_PASSES_IN_FLIGHT, together)
    try:
        seen = []
        assert np.array_equal(renderer.render(s.params, progress=seen.append), cpu)
        assert seen == [1, 2, 3, 4, 5, 6, 7]
        # resumed: continue(7), four more passes of one sample, from the stored bytes
        more = scenes.room(asset_root, w=96, h=54, spp=4, bounces=10, incremental=1, cont=7)
        cpu2, _ = O.render(more, more.params, rgb8=cpu.copy(), threads=4)
        assert np.array_equal(renderer.render(more.params, rgb8=cpu.copy()), cpu2)
        # a tile of the frame
        t = s.params.copy()
        t.tile_x, t.tile_y, t.tile_w, t.tile_h = 16, 8, 40, 30
        cpu_t, _ = O.render(s, t, threads=4)
        assert np.array_equal(renderer.render(t), cpu_t)
    finally:
        renderer.set_option(F.RT_OPT_PASSES_IN_FLIGHT, 0)


def test_resolve_matches_oracle(renderer):
    s = scenes.sphere_plane(w=32, h=18, spp=4)
    renderer.upload(s)
    sums = renderer.render_sums(s.params, 0, 4)
    # include values outside [0,1], NaN and infinities: the cast saturates, NaN -> 0
    sums[0, 0] = [np.nan, np.inf, -1.0]
    sums[0, 1] = [1e30, 4.0, 3.9999]
    assert np.array_equal(renderer.resolve(s.params, sums, 4), O.resolve(s.params, sums, 4))
    old = np.random.default_rng(1).integers(0, 256, size=(18, 32, 3), dtype=np.uint8)
    assert np.array_equal(renderer.resolve(s.params, sums, 4, True, 7, old.copy()),
                          O.resolve(s.params, sums, 4, True, 7, old.copy()))


TRACE_DEFAULTS = {F.RT_OPT_TRACE_SHARE: 1, F.RT_OPT_TRACE_PAIR_MIN: 384, F.RT_OPT_TRACE_LDS_SLOTS: 6,
                  F.RT_OPT_TRACE_WAVES: 8, F.RT_OPT_TRACE_CACHE_NODES: 144, F.RT_OPT_TRACE_REFILL_MIN: 12}


def check_traversal_scenes(renderer, asset_root):
    check_scene(renderer, scenes.room(asset_root, w=64, h=36, spp=3, bounces=20))
    check_scene(renderer, scenes.blob(asset_root, w=48, h=48, spp=2))
    check_scene(renderer, scenes.stacked_duplicates(n=150, w=24, h=24))  # long leaf, several chunks, exact ties
    check_scene(renderer, scenes.deep_chain(n=60, w=48, h=16))
    check_scene(renderer, scenes.grid_wall(n=16, w=33, h=33))


@pytest.mark.parametrize("share,pair_min,slots", [(0, 384, 8), (1, 1, 8), (1, 48, 2), (1, 300, 16)])
def test_traversal_variants_give_identical_results(renderer, asset_root, share, pair_min, slots):
    # one shape per lane and step, or pending triangle tests dealt out to all lanes;
    # few LDS stack slots push most of the stack to the overflow area
    renderer.set_option(F.RT_OPT_TRACE_SHARE, share)
    renderer.set_option(F.RT_OPT_TRACE_PAIR_MIN, pair_min)
    renderer.set_option(F.RT_OPT_TRACE_LDS_SLOTS, slots)
    try:
        check_traversal_scenes(renderer, asset_root)
    finally:
        for k, v in TRACE_DEFAULTS.items():
            renderer.set_option(k, v)


@pytest.mark.parametrize("waves,cache_nodes,slots,refill", [(1, 0, 8, 1), (8, 0, 8, 64), (8, 3, 4, 12), (16, 1024, 8, 1),
                                                           (16, 2048, 2, 33)])
def test_workgroup_shapes_and_cached_tree_top(renderer, asset_root, waves, cache_nodes, slots, refill):
    # one or several waves per workgroup sharing an LDS copy of the first branch
    # records (more of them than the tree has, and very few, are both covered);
    # refilling idle lanes one by one or only once the whole wave is idle
    renderer.set_option(F.RT_OPT_TRACE_WAVES, waves)
    renderer.set_option(F.RT_OPT_TRACE_CACHE_NODES, cache_nodes)
    renderer.set_option(F.RT_OPT_TRACE_LDS_SLOTS, slots)
    renderer.set_option(F.RT_OPT_TRACE_REFILL_MIN, refill)
    try:
        check_traversal_scenes(renderer, asset_root)
    finally:
        for k, v in TRACE_DEFAULTS.items():
            renderer.set_option(k, v)


def test_workgroup_that_cannot_fit_is_refused(renderer):
    s = scenes.grid_wall(n=16, w=33, h=33)
    renderer.upload(s)
    renderer.set_option(F.RT_OPT_TRACE_WAVES, 16)
    renderer.set_option(F.RT_OPT_TRACE_LDS_SLOTS, 64)  # 16 waves x 33 KB of stack exceed a compute unit's LDS
    try:
        with pytest.raises(RtError):
            renderer.trace_primary(s.params, 0, False)
    finally:
        for k, v in TRACE_DEFAULTS.items():
            renderer.set_option(k, v)
    check_scene(renderer, s)  # the context stays usable


def test_shared_reciprocal_division_is_ieee_division(renderer):
    # 2^29 operand pairs over the whole admitted regime, edges included: zero differences
    assert renderer.self_test_division(1 << 29) == 0


def test_triangle_reciprocal_is_ieee_division(renderer):
    # exhaustive: all 2 x 80 x 2^23 values of the range
    assert renderer.self_test_reciprocal() == 0


def test_texel_conversion_is_the_true_quotient(renderer):
    # exhaustive: 256 bytes in each of the three positions of a texel word, against byte / 255.0f
    assert renderer.self_test_unit_byte() == 0


def test_extreme_coordinates_take_the_plain_division_path(renderer):
    # coordinates outside the regime of the shared-reciprocal division (tiny, huge,
    # and directions with tiny or zero components) must still match exactly
    rng = np.random.default_rng(9)
    # (3e10: determinants above 2^60, where the triangle test takes the general division as well)
    for scale in (1e-25, 0.05, 3e4, 3e10, 1e22):
        v, f, _ = __import__("raytracer_amd").assets.grid_patch((-4 * scale, -4 * scale, 0), (8 * scale, 0, 0),
                                                                   (0, 8 * scale, 0), 8, 8)
        s = scenes.Scene.from_shapes(scenes.camera_params(w=16, h=16), triangles=scenes._triangles(v[f]),
                                     materials=scenes._one_material())
        renderer.upload(s)
        o = (rng.uniform(-3, 3, (2000, 3)) * scale).astype(np.float32)
        o[:, 2] = 5 * scale
        d = rng.normal(size=(2000, 3)).astype(np.float32)
        d[:, 2] = -np.abs(d[:, 2]) - 0.5
        d /= np.linalg.norm(d, axis=1, keepdims=True)
        d[::5, 0] = 1e-35
        d[::7, 1] = 0.0
        gpu, cpu = renderer.trace_rays(o, d), O.trace_rays(s, o, d)
        assert_hits_equal(gpu, cpu)
        # at the extreme scales nothing can be hit (|det| < 2e-6, or overflow), identically on both sides
        assert (gpu["kind"] == F.RT_SHAPE_TRIANGLE).any() == (1e-3 < scale < 1e12)


def test_errors_are_reported_not_fatal(renderer):
    from raytracer_amd.renderer import Renderer, RtError

    s = scenes.sphere_plane(w=16, h=8, spp=1)
    with Renderer(0) as r:
        with pytest.raises(RtError) as e:
            r.render_sums(s.params, 0, 1)  # nothing uploaded yet
        assert e.value.code == -5
        bad = scenes.sphere_plane(w=16, h=8, spp=1)
        bad.spheres["material"] = 99
        with pytest.raises(RtError) as e:
            r.upload(bad)
        assert e.value.code == -1
        r.upload(s)
        p = s.params.copy()
        p.rng_mode = F.RT_RNG_SERIAL
        with pytest.raises(RtError):
            r.render_sums(p, 0, 1)
        p = s.params.copy()
        p.tile_x, p.tile_w, p.tile_h = 10, 10, 4
        with pytest.raises(RtError):
            r.render_sums(p, 0, 1)


@pytest.mark.parametrize("which", ["room", "tree", "materials"])
def test_random_views(renderer, asset_root, which):
    """Cameras at random places (inside furniture, grazing walls, looking away from the scene):
    radiance and counters still equal the oracle's bit for bit."""
    base = {"room": lambda: scenes.room(asset_root, w=40, h=24, spp=2, bounces=14),
            "tree": lambda: scenes.tree(asset_root, w=40, h=24, spp=2, bounces=14),
            "materials": lambda: scenes.materials(w=40, h=24, spp=2, bounces=14)}[which]()
    renderer.upload(base)
    rng = np.random.default_rng({"room": 1, "tree": 2, "materials": 3}[which])
    for _ in range(6):
        p = base.params.copy()
        cam = rng.uniform(-2.5, 2.5, 3) + np.array([0.0, 1.5, 0.0])
        look = rng.normal(size=3)
        look /= np.linalg.norm(look)
        right = np.cross(look, [0.0, 1.0, 0.0])
        right = right / np.linalg.norm(right) if np.linalg.norm(right) > 1e-3 else np.array([1.0, 0.0, 0.0])
        up = np.cross(right, look)
        centre = cam + look
        p.camera[:] = [float(v) for v in cam]
        p.top_edge[:] = [float(v) for v in right * 1.6]
        p.left_edge[:] = [float(v) for v in -up * 0.96]
        p.top_left[:] = [float(v) for v in centre - right * 0.8 + up * 0.48]
        renderer.reset_stats()
        c = O.OracleCounters()
        cpu = O.render_sums(base, p, 0, 2, threads=4, counters=c)
        gpu = renderer.render_sums(p, 0, 2)
        assert np.array_equal(bits(gpu), bits(cpu))
        s = renderer.stats()
        assert s["segments"] == c.segments and s["hits"] == c.hits
        assert_hits_equal(renderer.trace_primary(p, 0, True), O.trace_primary(base, p, 0, True))


def test_sphere_wins_an_exact_tie_with_a_plane_when_only_the_planes_are_scanned_inline(renderer):
    # 21 spheres (a real tree: traversed by the kernel) and one plane (small enough to be scanned
    # where rays are created).  The ray below meets the sphere resting on the plane and the plane
    # at exactly t = 5; lib.rs:333 keeps the first family, the sphere.
    from raytracer_amd.scene import PLANE_DT, SPHERE_DT

    sp = np.zeros(21, dtype=SPHERE_DT)
    sp["center"][0], sp["radius"][0] = (0, 1, 0), 1.0
    for i in range(1, 21):
        sp["center"][i], sp["radius"][i] = (10.0 + 3 * i, 1.0, -5.0 + (i % 5)), 0.5
    pl = np.zeros(1, dtype=PLANE_DT)
    pl["point"], pl["normal"] = (0, 0, 0), (0, 1, 0)
    s = scenes.Scene.from_shapes(scenes.camera_params(w=16, h=16), spheres=sp, planes=pl, materials=scenes._one_material())
    assert len(s.sphere_nodes) > 1
    renderer.upload(s)
    o = np.array([[0, -5, 0], [0.25, -5, 0], [30, -5, 0]], np.float32)
    d = np.array([[0, 1, 0]] * 3, np.float32)
    gpu, cpu = renderer.trace_rays(o, d), O.trace_rays(s, o, d)
    assert_hits_equal(gpu, cpu)
    assert gpu["t"][0] == 5.0 and gpu["kind"][0] == F.RT_SHAPE_SPHERE
    assert gpu["kind"][2] == F.RT_SHAPE_PLANE
    check_scene(renderer, s)


def test_malformed_descriptions_are_refused(renderer):
    from raytracer_amd.scene import MATERIAL_DT

    # a node with two parents (a graph, not a tree) would make the stack depth estimate wrong
    s = scenes.grid_wall(n=8, w=16, h=16)
    bad = scenes.grid_wall(n=8, w=16, h=16)
    branches = np.nonzero((bad.triangle_nodes["b"] & F.RT_LEAF_BIT) == 0)[0]
    last = int(branches[-1])
    bad.triangle_nodes["b"][0] = bad.triangle_nodes["b"][last]  # the root's second child is now shared with `last`
    with pytest.raises(RtError, match="two parents"):
        renderer.upload(bad)
    # a texture on a sphere
    t = scenes.sphere_plane(w=16, h=8, spp=1)
    m = np.zeros(1, dtype=MATERIAL_DT)
    m["color_kind"], m["tex_width"], m["tex_height"] = F.RT_COLOR_TEXTURE, 2, 2
    t.materials = np.concatenate([t.materials, m])
    t.texels = np.zeros(12, np.uint8)
    t.spheres["material"] = len(t.materials) - 1
    with pytest.raises(RtError, match="triangles only"):
        renderer.upload(t)
    # a texture whose size in bytes does not fit 64 bits must not pass the size check by wrapping round
    u = scenes.sphere_plane(w=16, h=8, spp=1)
    m = np.zeros(1, dtype=MATERIAL_DT)
    m["color_kind"], m["tex_width"], m["tex_height"] = F.RT_COLOR_TEXTURE, 0xFFFFFFFF, 0xFFFFFFFF
    u.materials = np.concatenate([u.materials, m])
    u.texels = np.zeros(12, np.uint8)
    with pytest.raises(RtError, match="outside the texel array"):
        renderer.upload(u)
    check_scene(renderer, s)  # the context stays usable


def test_fast_slab_option_is_off_by_default_and_close_to_exact_when_on(renderer, asset_root):
    # The reciprocal slab test is an explicit option outside the bit-exact contract: hit distances are
    # still computed by the exact triangle test, so wherever the same primitive is found the distance
    # bits are equal; primitives may differ only where two candidates are equally far to rounding.
    s = scenes.room(asset_root, w=96, h=54, spp=4, bounces=12)
    renderer.upload(s)
    exact = renderer.trace_primary(s.params, 0, True)
    exact_sums = renderer.render_sums(s.params, 0, 4)
    assert_hits_equal(exact, O.trace_primary(s, s.params, 0, True))  # default = exact
    renderer.set_option(F.RT_OPT_FAST_SLAB, 1)
    try:
        fast = renderer.trace_primary(s.params, 0, True)
        fast_sums = renderer.render_sums(s.params, 0, 4)
    finally:
        renderer.set_option(F.RT_OPT_FAST_SLAB, 0)
    same = (fast["kind"] == exact["kind"]) & (fast["prim"] == exact["prim"])
    assert same.mean() > 0.999
    assert np.array_equal(bits(fast["t"][same]), bits(exact["t"][same]))
    # tolerance of the fast mode (not of the default): RMSE 1e-3 of the mean radiance at 4 spp on this
    # small frame, where one diverging path moves a pixel visibly; the exact mode's tolerance stays 0
    err = (fast_sums - exact_sums).astype(np.float64) / 4
    assert np.sqrt(np.mean(err ** 2)) < 1e-2
    assert (bits(fast_sums) != bits(exact_sums)).mean() < 0.02
    # the option is really off again
    assert np.array_equal(bits(renderer.render_sums(s.params, 0, 4)), bits(exact_sums))
