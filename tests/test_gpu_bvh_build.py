"""The GPU BVH builder against the host builder (itself checked node for node
against an independent restatement in tests/test_host_ingest.py): same node
array in the reference's push order, same permutation of the triangles, and —
what the tree is for — renders that are bit-identical."""
import ctypes as C
import time

import numpy as np
import pytest

from oracle import binding as O
from raytracer_amd import _ffi as F
from raytracer_amd.renderer import RtError
from raytracer_amd.scene import NODE_DT, TRIANGLE_DT, Scene

from . import scenes

pytestmark = pytest.mark.gpu


def host_build(triangles):
    lib = F.host_lib()
    tris = np.ascontiguousarray(triangles, dtype=TRIANGLE_DT).copy()
    n = F.u32(0)
    raw = lib.rth_build_bvh_triangles(tris.ctypes.data if tris.size else None, tris.size, C.byref(n))
    buf = (C.c_char * (n.value * NODE_DT.itemsize)).from_address(raw)
    nodes = np.frombuffer(buf, dtype=NODE_DT, count=n.value).copy()
    lib.rth_free(raw)
    return nodes, tris


def assert_same_tree(gpu_nodes, gpu_tris, cpu_nodes, cpu_tris):
    assert len(gpu_nodes) == len(cpu_nodes)
    assert np.array_equal(gpu_nodes["a"], cpu_nodes["a"]) and np.array_equal(gpu_nodes["b"], cpu_nodes["b"])
    # boxes equal as numbers (a zero may carry either sign: min/max folds in another order)
    assert np.array_equal(gpu_nodes["min"], cpu_nodes["min"]) and np.array_equal(gpu_nodes["max"], cpu_nodes["max"])
    assert gpu_tris.tobytes() == cpu_tris.tobytes()  # the same in-place permutation


def shuffled(triangles, seed):
    t = triangles.copy()
    np.random.default_rng(seed).shuffle(t)
    return t


@pytest.mark.parametrize("which", ["room", "blob", "tree"])
def test_generated_meshes(renderer, asset_root, which):
    scene = getattr(scenes, which)(asset_root)
    tris = shuffled(scene.triangles, 5)  # any input order: the builder defines the output order
    assert_same_tree(*renderer.build_bvh_triangles(tris), *host_build(tris))


@pytest.mark.parametrize("n", [0, 1, 7, 8, 9, 64, 1000])
def test_small_and_random_soups(renderer, n):
    rng = np.random.default_rng(n)
    t = np.zeros(n, dtype=TRIANGLE_DT)
    t["a"] = rng.uniform(-5, 5, size=(n, 3))
    t["e1"] = rng.normal(size=(n, 3)) * 0.3
    t["e2"] = rng.normal(size=(n, 3)) * 0.3
    t["normals_idx"] = t["uv_idx"] = t["bary_idx"] = F.RT_NO_INDEX
    t["material"] = np.arange(n) % 3
    assert_same_tree(*renderer.build_bvh_triangles(t), *host_build(t))


def test_degenerate_inputs(renderer):
    # identical triangles cannot be separated: one long leaf; flat (axis-aligned) soups have a zero extent;
    # coordinates with both signs of zero
    t = np.zeros(300, dtype=TRIANGLE_DT)
    t["a"], t["e1"], t["e2"] = (1, 2, 3), (1, 0, 0), (0, 1, 0)
    t["normals_idx"] = t["uv_idx"] = t["bary_idx"] = F.RT_NO_INDEX
    nodes, _ = renderer.build_bvh_triangles(t)
    assert len(nodes) == 1 and nodes["b"][0] == (300 | F.RT_LEAF_BIT)
    assert_same_tree(nodes, t, *host_build(t))
    flat = scenes.grid_wall(24).triangles
    flat["a"][::3, 2] = -0.0
    assert_same_tree(*renderer.build_bvh_triangles(flat), *host_build(flat))
    deep = scenes.deep_chain(60).triangles
    assert_same_tree(*renderer.build_bvh_triangles(deep), *host_build(deep))


def test_non_finite_coordinates_are_left_to_the_host_builder(renderer):
    t = np.zeros(20, dtype=TRIANGLE_DT)
    t["e1"], t["e2"] = (1, 0, 0), (0, 1, 0)
    t["a"][7, 1] = np.inf
    with pytest.raises(RtError, match="non-finite"):
        renderer.build_bvh_triangles(t)


def test_renders_with_the_gpu_built_tree_are_bit_identical(renderer, asset_root):
    scene = scenes.room(asset_root, w=64, h=36, spp=3, bounces=12)
    tris = shuffled(scene.triangles, 11)
    nodes, ordered = renderer.build_bvh_triangles(tris)
    built = Scene(scene.params, scene.spheres, scene.planes, ordered, scene.vertex_normals, scene.tex_coords,
                  scene.bary_precomputed, scene.materials, scene.texels, scene.sphere_nodes, scene.plane_nodes, nodes)
    cpu_nodes, cpu_ordered = host_build(tris)
    ref = Scene(scene.params, scene.spheres, scene.planes, cpu_ordered, scene.vertex_normals, scene.tex_coords,
                scene.bary_precomputed, scene.materials, scene.texels, scene.sphere_nodes, scene.plane_nodes, cpu_nodes)
    renderer.upload(built)
    got = renderer.render_sums(built.params, 0, 3)
    hits = renderer.trace_primary(built.params, 0, True)
    want = O.render_sums(ref, ref.params, 0, 3, threads=4)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    cpu_hits = O.trace_primary(ref, ref.params, 0, True)
    assert np.array_equal(hits["prim"], cpu_hits["prim"]) and np.array_equal(hits["t"].view(np.uint32), cpu_hits["t"].view(np.uint32))


def test_full_size_room_and_build_time(renderer, tmp_path_factory):
    from raytracer_amd import assets, parse
    root = str(tmp_path_factory.mktemp("room_bvh"))
    assets.write_room(root, "room", detail=4)
    scene = parse(assets.room_scene_text("room", 64, 36, 1, 4), root)
    tris = shuffled(scene.triangles, 3)
    renderer.build_bvh_triangles(tris[:1000])  # first call pays for loading the kernels
    t0 = time.perf_counter()
    gpu = renderer.build_bvh_triangles(tris)
    t_gpu = time.perf_counter() - t0
    t0 = time.perf_counter()
    cpu = host_build(tris)
    t_cpu = time.perf_counter() - t0
    assert_same_tree(*gpu, *cpu)
    print(f"\\nBVH over {len(tris)} triangles: GPU {t_gpu * 1e3:.1f} ms, host {t_cpu * 1e3:.1f} ms ({t_cpu / t_gpu:.0f}x)")
    assert t_gpu < t_cpu


def test_scene_parsed_with_the_gpu_builder_equals_the_host_parse(renderer, asset_root):
    from raytracer_amd import assets, parse
    text = assets.room_scene_text("room", 32, 18, 1, 4)
    a, b = parse(text, asset_root), parse(text, asset_root, renderer=renderer)
    assert_same_tree(b.triangle_nodes, b.triangles, a.triangle_nodes, a.triangles)
    assert a.sphere_nodes.tobytes() == b.sphere_nodes.tobytes() and a.materials.tobytes() == b.materials.tobytes()


def test_sphere_family(renderer):
    from raytracer_amd.scene import SPHERE_DT
    lib = F.host_lib()
    for n in (0, 5, 8, 300, 20000):
        rng = np.random.default_rng(n + 1)
        sp = np.zeros(n, dtype=SPHERE_DT)
        sp["center"] = rng.uniform(-10, 10, size=(n, 3))
        sp["radius"] = rng.uniform(0.01, 0.5, size=n)
        sp["material"] = np.arange(n) % 5
        host = sp.copy()
        cnt = F.u32(0)
        raw = lib.rth_build_bvh_spheres(host.ctypes.data if n else None, n, C.byref(cnt))
        buf = (C.c_char * (cnt.value * NODE_DT.itemsize)).from_address(raw)
        cpu_nodes = np.frombuffer(buf, dtype=NODE_DT, count=cnt.value).copy()
        lib.rth_free(raw)
        assert_same_tree(*renderer.build_bvh_spheres(sp), cpu_nodes, host)
