"""Host-side ingest (librt_host.so): scene-file parser, OBJ/MTL/PPM loaders,
BVH builder and the memory-mapped image, against the behaviour of the
reference's src/config.rs, src/obj.rs, src/material.rs:150-189, src/bvh.rs:46-224
and src/mmap.rs.  No GPU needed."""
import ctypes as C
import os

import numpy as np
import pytest

from oracle import bvh_ref
from raytracer_amd import _ffi as F
from raytracer_amd import assets, make_params, parse
from raytracer_amd.scene import NODE_DT, SPHERE_DT, TRIANGLE_DT, Scene

from . import scenes

f32 = np.float32
REFERENCE_OBJ = "/root/reference/obj"


# ---------------------------------------------------------------- scene files


def test_parse_plumbing_scene():
    s = scenes.sphere_plane(w=640, h=360, spp=16, bounces=10)
    p = s.params
    assert (p.width, p.height, p.samples_per_pixel, p.max_bounces) == (640, 360, 16, 10)
    assert p.incremental == 0 and p.has_continue == 0
    assert list(p.top_left) == [f32(-1.7778), 2.0, 1.0] and list(p.camera) == [0.0, 1.0, 5.0]
    assert len(s.spheres) == 1 and len(s.planes) == 1 and len(s.triangles) == 0
    assert s.materials["kind"].tolist() == [F.RT_MAT_LAMBERTIAN] * 2
    assert s.materials["rgb"][1].tolist() == [f32(0.1), f32(0.1), f32(0.6)]
    # one root leaf per family; the plane's box is infinite, the empty family's inverted
    assert len(s.sphere_nodes) == len(s.plane_nodes) == len(s.triangle_nodes) == 1
    assert np.all(np.isinf(s.plane_nodes[0]["min"])) and s.triangle_nodes[0]["b"] == F.RT_LEAF_BIT
    assert s.triangle_nodes[0]["min"].tolist() == [np.inf] * 3


def test_parse_switches_and_materials():
    s = scenes.materials(prefix="continue(3)\nincremental(2)\n", spp=4)
    assert s.params.incremental == 2 and s.params.has_continue == 1 and s.params.continue_sampling == 3
    kinds = s.materials["kind"].tolist()
    assert kinds.count(F.RT_MAT_GLASS) == 2 and F.RT_MAT_LIGHT in kinds
    assert f32(0.4) in s.materials["param"]
    # the plane normal is normalised on load (config.rs:74); scene triangles carry no side tables
    assert np.allclose(np.linalg.norm(s.planes["normal"], axis=1), 1.0)
    assert np.all(s.triangles["bary_idx"] == F.RT_NO_INDEX)


def test_materials_are_interned_by_equality():
    text = ("screen(-1 1 1, 2 0 0, 0 -2 0, 4, 4, 1, 2)\ncamera(0 0 3)\n"
            "spheres((0 0 0, 1, lambertian, 1 0 0), (3 0 0, 1, lambertian, 1 0 0), (6 0 0, 1, metal 0.5, 1 0 0), "
            "(9 0 0, 1, metal 0.5, 1 0 0), (12 0 0, 1, metal 0.25, 1 0 0))\nplanes((0 0 0, 0 1 0, lambertian, 1 0 0))\ntriangles()\n")
    s = parse(text)
    assert len(s.materials) == 3  # config.rs:157-178
    by_x = {float(sp["center"][0]): int(sp["material"]) for sp in s.spheres}
    assert by_x[0.0] == by_x[3.0] == int(s.planes[0]["material"]) and by_x[6.0] == by_x[9.0] != by_x[12.0]


def test_parser_stops_once_everything_is_seen(asset_root):
    # config.rs:32-37,93: `obj(...)` counts as the triangles entry, later lines are never read
    text = assets.blob_scene_text("blob", 8, 8, 1, 2).replace("triangles()\n", "this line is never parsed\n")
    assert len(parse(text, asset_root).triangles) == 1280
    # scene triangles given before the obj line are kept and the obj's appended
    text = ("screen(-1 1 1, 2 0 0, 0 -2 0, 4, 4, 1, 2)\ncamera(0 0 3)\nspheres()\nplanes()\n"
            "triangles((0 0 0, 1 0 0, 0 1 0, light, 1 1 1))\n")
    assert len(parse(text).triangles) == 1


@pytest.mark.parametrize("text, message", [
    ("screen(-1 1 1, 2 0 0, 0 -2 0, 4, 4, 1, 2)\ncamera(0 0 3)\nspheres()\n", "ended"),
    ("sky(blue)\n", "Unknown entry"),
    ("incremental(3)\nscreen(-1 1 1, 2 0 0, 0 -2 0, 4, 4, 4, 2)\ncamera(0 0 3)\nspheres()\nplanes()\ntriangles()\n", "divide"),
    ("incremental(0)\n", "incremental"),
    ("screen(-1 1 1, 2 0 0, 0 -2 0, 4, 4, 4)\n", "7 fields"),
    ("spheres((0 0 0, 1, plastic, 1 0 0))\n", "Unknown material"),
    ("camera(0 0)\n", "three numbers"),
    ("camera(0 0  3)\n", "not a number"),
])
def test_parse_errors(text, message):
    # the reference panics on all of these; here they are reported
    with pytest.raises(ValueError, match=message):
        parse(text)


# ------------------------------------------------------------------------ OBJ


def write(path, text):
    os.makedirs(os.path.dirname(path), exist_ok=True)
    with open(path, "w") as f:
        f.write(text)


def obj_scene(root, name):
    return parse(f"screen(-1 1 1, 2 0 0, 0 -2 0, 4, 4, 1, 2)\ncamera(0 0 3)\nspheres()\nplanes()\nobj(({name}))\n", root)


def test_obj_fan_triangulation_indices_and_tables(tmp_path):
    root = str(tmp_path)
    write(f"{root}/obj/quad.obj",
          "# a quad and a triangle\nv 0 0 0\nv 1 0 0\nv 1 1 0\nv 0 1 0\nv 0 0 2 0.5 0.5 0.5\n"
          "vt 0 0\nvt 1 0\nvt 1 1\nvt 0 1 0\nvn 0 0 2\nvn 0 0 1\n"
          "f 1/1/1 2/2/1 3/3/2 4/4/2\n"      # quad -> (1,2,3), (1,3,4); full tables
          "f -1 -5 -4\n"                      # negative indices, positions only
          "f 1/1 2/2 3/3\n"                   # texture coordinates only
          "f 1//1 2//1 3//2\n")               # normals only
    s = obj_scene(root, "quad")
    assert len(s.triangles) == 5
    order = np.argsort(s.triangles["bary_idx"].astype(np.int64) + (s.triangles["bary_idx"] == F.RT_NO_INDEX) * 10**12)
    t = s.triangles[order]
    # fan: both triangles of the quad start at vertex 1
    assert t[0]["a"].tolist() == t[1]["a"].tolist() == [0, 0, 0]
    assert t[0]["e1"].tolist() == [1, 0, 0] and t[0]["e2"].tolist() == [1, 1, 0]
    assert t[1]["e1"].tolist() == [1, 1, 0] and t[1]["e2"].tolist() == [0, 1, 0]
    assert s.vertex_normals.shape == (3, 3, 3) and s.tex_coords.shape == (3, 3, 2) and s.bary_precomputed.shape == (4, 4)
    assert np.allclose(np.linalg.norm(s.vertex_normals, axis=2), 1.0)       # normalised on load, obj.rs:136
    # barycentric precomputation d00, d01, d11, d00*d11 - d01^2 (obj.rs:142-153)
    assert s.bary_precomputed[0].tolist() == [1, 1, 2, 1]
    none = s.triangles[s.triangles["bary_idx"] == F.RT_NO_INDEX][0]
    assert none["a"].tolist() == [0, 0, 2] and none["e1"].tolist() == [0, 0, -2]
    uv_only = s.triangles[(s.triangles["uv_idx"] != F.RT_NO_INDEX) & (s.triangles["normals_idx"] == F.RT_NO_INDEX)]
    n_only = s.triangles[(s.triangles["uv_idx"] == F.RT_NO_INDEX) & (s.triangles["normals_idx"] != F.RT_NO_INDEX)]
    assert len(uv_only) == 1 and len(n_only) == 1
    # no mtllib: one default grey Lambertian (obj.rs:73-81)
    assert len(s.materials) == 1 and s.materials[0]["rgb"].tolist() == [0.5, 0.5, 0.5]


def test_obj_materials_and_usemtl_sections(tmp_path):
    root = str(tmp_path)
    os.makedirs(f"{root}/obj")
    assets.write_ppm(f"{root}/obj/tex.ppm", np.arange(2 * 3 * 3, dtype=np.uint8).reshape(2, 3, 3))
    write(f"{root}/obj/m.mtl", "# header\nnewmtl red\nKd 1 0 0\nKa 0 0 0\n\nnewmtl tex\nKd 0 1 0\nmap_Kd tex.ppm\n\nnewmtl plain\nNs 10\n")
    write(f"{root}/obj/m.obj",
          "mtllib m.mtl\nv 0 0 0\nv 1 0 0\nv 0 1 0\nf 1 2 3\nusemtl red\nf 1 2 3\nusemtl tex\nf 1 2 3\nusemtl nosuch\nf 1 2 3\nusemtl plain\nf 1 2 3\n")
    s = obj_scene(root, "m")
    m = s.materials
    # interned in MTL order first: red, tex (map_Kd wins over Kd), plain (grey); the default grey
    # for the piece before the first usemtl and for the unknown name is the same grey
    assert len(m) == 3
    assert m[0]["rgb"].tolist() == [1, 0, 0] and m[1]["color_kind"] == F.RT_COLOR_TEXTURE
    assert (m[1]["tex_width"], m[1]["tex_height"]) == (3, 2) and s.texels.tolist() == list(range(18))
    assert m[2]["rgb"].tolist() == [0.5, 0.5, 0.5]
    assert np.all(m["kind"] == F.RT_MAT_LAMBERTIAN)  # obj.rs:219-226
    # the builder permutes triangles; compare as a multiset
    assert sorted(s.triangles["material"].tolist()) == [0, 1, 2, 2, 2]


def test_obj_errors(tmp_path):
    root = str(tmp_path)
    write(f"{root}/obj/bad.obj", "v 0 0 0\nv 1 0 0\nv 0 1 0\nf 1/1/1 2/1/1 3/1/1\n")  # vt/vn indices without vt/vn lines
    with pytest.raises(ValueError, match="out of range"):
        obj_scene(root, "bad")
    write(f"{root}/obj/nomtl.obj", "mtllib missing.mtl\nv 0 0 0\n")
    with pytest.raises(ValueError, match="cannot open"):
        obj_scene(root, "nomtl")
    with pytest.raises(ValueError, match="cannot open"):
        obj_scene(root, "does_not_exist")


def test_ppm_header_grammar(tmp_path):
    lib = F.host_lib()

    def load(data):
        path = str(tmp_path / "t.ppm")
        with open(path, "wb") as f:
            f.write(data)
        w, h, err = F.u32(0), F.u32(0), C.create_string_buffer(256)
        ptr = lib.rth_load_ppm_p6(path.encode(), C.byref(w), C.byref(h), err, len(err))
        if not ptr:
            return None, err.value.decode()
        out = bytes((C.c_uint8 * (w.value * h.value * 3)).from_address(ptr))
        lib.rth_free(ptr)
        return (w.value, h.value, out), ""

    ok, _ = load(b"P6\n2 1\n255\n" + bytes(range(6)))
    assert ok == (2, 1, bytes(range(6)))
    assert load(b"P5\n2 1\n255\n" + bytes(6))[0] is None                       # magic
    assert load(b"P6\n2 1\n65535\n" + bytes(12))[0] is None                    # maxval must be 255
    assert load(b"P6\n2 1\n255\n" + bytes(5))[0] is None                       # pixel count
    assert load(b"P6\n# comment\n2 1\n255\n" + bytes(6))[0] is None            # comments are not understood


@pytest.mark.skipif(not os.path.isdir(REFERENCE_OBJ), reason="reference checkout not present")
def test_reference_assets_load_like_the_reference_loader():
    # the two assets of the checkout that the reference's own loader can read as they are
    s = obj_scene("/root/reference", "smooth_uv_sphere")
    assert len(s.triangles) == 960 and len(s.materials) == 1 and s.vertex_normals.shape[0] == 960
    s = obj_scene("/root/reference", "hollow knight")
    assert len(s.triangles) == 18352 and np.all(s.triangles["normals_idx"] != F.RT_NO_INDEX)
    assert np.all(s.materials["color_kind"] == F.RT_COLOR_SOLID)
    # faces reference vn indices but the file has no vn lines: the reference's loader panics
    with pytest.raises(ValueError, match="out of range"):
        obj_scene("/root/reference", "uv_sphere")


# ------------------------------------------------------------------------ BVH


def check_against_python_builder(bounds, nodes, order_of_c):
    ref_nodes, ref_order = bvh_ref.build(bounds)
    assert len(ref_nodes) == len(nodes)
    for a, b in zip(ref_nodes, nodes):
        assert a["a"] == int(b["a"]) and a["b"] == int(b["b"])
        assert np.array_equal(a["min"].view(np.uint32), b["min"].view(np.uint32))
        assert np.array_equal(a["max"].view(np.uint32), b["max"].view(np.uint32))
    assert ref_order == order_of_c


def test_bvh_builder_matches_independent_restatement_triangles():
    rng = np.random.default_rng(11)
    n = 700
    a = rng.uniform(-5, 5, (n, 3)).astype(f32)
    tris = np.zeros(n, dtype=TRIANGLE_DT)
    tris["a"], tris["e1"], tris["e2"] = a, rng.normal(0, 0.4, (n, 3)), rng.normal(0, 0.4, (n, 3))
    tris["material"] = np.arange(n)  # remember the input position
    s = Scene.from_shapes(scenes.camera_params(), triangles=tris)
    bounds = [bvh_ref.triangle_bounds(t["a"], t["e1"], t["e2"]) for t in tris]
    check_against_python_builder(bounds, s.triangle_nodes, s.triangles["material"].tolist())
    assert len(s.triangle_nodes) > 50 and scenes.bvh_depth(s.triangle_nodes) > 5
    # a node holding 7 shapes or fewer is never split: 15 + cost >= 2*n (bvh.rs:169)
    leaves = s.triangle_nodes[(s.triangle_nodes["b"] & F.RT_LEAF_BIT) != 0]
    assert ((leaves["b"] & 0x7FFFFFFF) >= 1).all()
    # pre-order: the first child is always the next node
    branches = np.nonzero((s.triangle_nodes["b"] & F.RT_LEAF_BIT) == 0)[0]
    assert np.array_equal(s.triangle_nodes["a"][branches], branches + 1)


def test_bvh_builder_matches_independent_restatement_spheres_and_degenerates():
    rng = np.random.default_rng(4)
    sp = np.zeros(300, dtype=SPHERE_DT)
    sp["center"], sp["radius"], sp["material"] = rng.uniform(-9, 9, (300, 3)), rng.uniform(0.05, 0.5, 300), np.arange(300)
    s = Scene.from_shapes(scenes.camera_params(), spheres=sp)
    check_against_python_builder([bvh_ref.sphere_bounds(x["center"], x["radius"]) for x in sp], s.sphere_nodes,
                                 s.spheres["material"].tolist())
    # many shapes with the same centroid interleaved with others: exercises the swap sequence
    tris = np.zeros(60, dtype=TRIANGLE_DT)
    tris["a"] = [[(i % 3) * 4.0, 0, 0] for i in range(60)]
    tris["e1"], tris["e2"], tris["material"] = [1, 0, 0], [0, 1, 0], np.arange(60)
    s = Scene.from_shapes(scenes.camera_params(), triangles=tris)
    check_against_python_builder([bvh_ref.triangle_bounds(t["a"], t["e1"], t["e2"]) for t in tris], s.triangle_nodes,
                                 s.triangles["material"].tolist())


def test_bvh_planes_and_empty_families_are_single_leaves():
    s = scenes.materials()
    assert len(s.plane_nodes) == 1 and int(s.plane_nodes[0]["b"]) == (2 | F.RT_LEAF_BIT)
    e = scenes.empty()
    for nodes in (e.sphere_nodes, e.plane_nodes, e.triangle_nodes):
        assert len(nodes) == 1 and int(nodes[0]["a"]) == 0 and int(nodes[0]["b"]) == F.RT_LEAF_BIT


def test_partition_in_place_semantics():
    # Rust core's algorithm: first false from the front swapped with last true from the back
    items = [5, 1, 6, 2, 7, 3]
    n = bvh_ref.partition_in_place(items, lambda v: v < 4)
    assert n == 3 and items == [3, 1, 2, 6, 7, 5]
    items = [1, 2, 3]
    assert bvh_ref.partition_in_place(items, lambda v: v < 4) == 3 and items == [1, 2, 3]
    items = [9, 8]
    assert bvh_ref.partition_in_place(items, lambda v: v < 4) == 0 and items == [9, 8]


def test_every_shape_is_in_exactly_one_leaf(asset_root):
    s = scenes.room(asset_root)
    n = s.triangle_nodes
    leaves = n[(n["b"] & F.RT_LEAF_BIT) != 0]
    first, count = leaves["a"].astype(np.int64), (leaves["b"] & 0x7FFFFFFF).astype(np.int64)
    order = np.argsort(first)
    assert first[order][0] == 0 and np.array_equal(first[order][1:], (first + count)[order][:-1])
    assert (first + count).max() == len(s.triangles)
    # every leaf box contains its triangles
    for lf in leaves[:200]:
        t = s.triangles[int(lf["a"]):int(lf["a"]) + int(lf["b"] & 0x7FFFFFFF)]
        pts = np.concatenate([t["a"], t["a"] + t["e1"], t["a"] + t["e2"]])
        assert (pts >= lf["min"]).all() and (pts <= lf["max"]).all()


# ---------------------------------------------------------------------- image


def test_image_sink_keeps_old_pixels(tmp_path):
    """src/lib.rs:70-83 + src/mmap.rs:41-77: header rewritten, never truncated."""
    lib = F.host_lib()
    path = str(tmp_path / "out.ppm")
    err = C.create_string_buffer(256)
    img = lib.rth_image_open(path.encode(), 5, 3, err, len(err))
    assert img, err.value
    header = b"P6\n5 3 255\n"
    assert lib.rth_image_header_len(img) == len(header)
    px = (C.c_uint8 * 45).from_address(lib.rth_image_pixels(img))
    px[:] = list(range(45))
    lib.rth_image_flush(img)
    lib.rth_image_close(img)
    assert open(path, "rb").read() == header + bytes(range(45))
    img = lib.rth_image_open(path.encode(), 5, 3, err, len(err))   # reopening keeps the pixels (the checkpoint)
    assert bytes((C.c_uint8 * 45).from_address(lib.rth_image_pixels(img))) == bytes(range(45))
    lib.rth_image_close(img)
    assert not lib.rth_image_open(str(tmp_path / "no" / "dir.ppm").encode(), 2, 2, err, len(err)) and err.value


def test_parser_takes_a_triangle_builder_and_falls_back_when_it_declines(asset_root):
    """rth_parse_scene_with: the hook the GPU builder is plugged in through."""
    from raytracer_amd import assets
    from raytracer_amd.scene import _view, NODE_DT, TRIANGLE_DT
    lib = F.host_lib()
    text = assets.blob_scene_text("blob", 16, 16, 1, 2).encode()
    calls = []

    def parse_with(fn):
        err = C.create_string_buffer(256)
        hook = F.RthTriangleBvhFn(fn)
        h = lib.rth_parse_scene_with(text, asset_root.encode(), hook, None, err, len(err))
        assert h, err.value
        d = F.RtSceneDesc()
        lib.rth_scene_desc(h, C.byref(d))
        out = (_view(d.triangle_nodes, d.n_triangle_nodes, NODE_DT), _view(d.triangles, d.n_triangles, TRIANGLE_DT))
        lib.rth_free_scene(h)
        return out

    def decline(_user, shapes, n, n_nodes):
        calls.append(("decline", n))
        return None

    def through_host_builder(_user, shapes, n, n_nodes):
        calls.append(("host", n))
        return lib.rth_build_bvh_triangles(shapes, n, n_nodes)

    plain = obj_scene(asset_root, "blob")
    for fn in (decline, through_host_builder):
        nodes, tris = parse_with(fn)
        assert nodes.tobytes() == plain.triangle_nodes.tobytes() and tris.tobytes() == plain.triangles.tobytes()
    assert calls == [("decline", len(plain.triangles)), ("host", len(plain.triangles))]
