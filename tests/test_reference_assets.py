"""The meshes the reference's own checkout holds (tests/golden/assets, byte for
byte /root/reference/obj/{boombox_4k,hollow knight,smooth_uv_sphere}.obj and their
.mtl files) through the host loader, against a second restatement of
/root/reference/src/obj.rs written separately in Python (oracle/obj_ref.py).
No GPU needed; the GPU side is tests/test_gpu_reference_assets.py."""
import os

import numpy as np
import pytest

from oracle import obj_ref
from raytracer_amd import _ffi as F
from raytracer_amd import assets, parse

MESHES = ["boombox_4k", "hollow knight", "smooth_uv_sphere"]


@pytest.fixture(scope="module")
def staged(tmp_path_factory):
    root = str(tmp_path_factory.mktemp("reference_assets"))
    assets.stage_reference_assets(root, texture_size=64)
    return root


def load_both(root, name):
    text = f"screen(-1 1 1, 2 0 0, 0 -2 0, 4, 4, 1, 2)\ncamera(0 0 3)\nspheres()\nplanes()\nobj(({name}))\n"
    s = parse(text, root)
    mats = obj_ref.Materials()
    ref = obj_ref.parse(os.path.join(root, f"obj/{name}.obj"), mats,
                        lambda p: open(p, encoding="utf-8").read(), lambda p: os.path.normpath(p))
    return s, ref, mats


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


@pytest.mark.parametrize("name", MESHES)
def test_loader_equals_independent_restatement(staged, name):
    s, ref, mats = load_both(staged, name)
    n = len(ref["corners"])
    assert len(s.triangles) == n
    # every triangle of these meshes has vertex normals, hence a barycentric row, numbered in
    # creation order: that undoes the permutation of the BVH build
    assert all(b is not None for b in ref["bary"])
    order = np.argsort(s.triangles["bary_idx"])
    t = s.triangles[order]
    assert np.array_equal(t["bary_idx"], np.arange(n))
    c = np.array(ref["corners"], dtype=np.float32)  # [n][3][3]
    assert np.array_equal(bits(t["a"]), bits(c[:, 0]))
    assert np.array_equal(bits(t["e1"]), bits(c[:, 1] - c[:, 0]))
    assert np.array_equal(bits(t["e2"]), bits(c[:, 2] - c[:, 0]))
    assert np.array_equal(t["material"], np.array(ref["material"], dtype=np.uint32))
    assert np.array_equal(bits(s.bary_precomputed), bits(np.array(ref["bary"], dtype=np.float32)))
    assert np.array_equal(t["normals_idx"], np.arange(n))
    assert np.array_equal(bits(s.vertex_normals), bits(np.array(ref["normals"], dtype=np.float32)))
    has_uv = np.array([u is not None for u in ref["uv"]])
    assert np.array_equal(t["uv_idx"] != F.RT_NO_INDEX, has_uv)
    assert np.array_equal(bits(s.tex_coords), bits(np.array([u for u in ref["uv"] if u is not None], dtype=np.float32).reshape(-1, 3, 2)))
    # material table: same order, same kinds
    assert len(s.materials) == len(mats.items)
    for m, (kind, color_kind, value) in zip(s.materials, mats.items):
        assert m["kind"] == F.RT_MAT_LAMBERTIAN and kind == "lambertian"
        if color_kind == "solid":
            assert m["color_kind"] == F.RT_COLOR_SOLID and np.array_equal(bits(m["rgb"]), bits(np.array(value)))
        else:
            assert m["color_kind"] == F.RT_COLOR_TEXTURE and (m["tex_width"], m["tex_height"]) == (64, 64)


def test_known_facts_of_the_three_meshes(staged):
    s, ref, mats = load_both(staged, "boombox_4k")
    # two `usemtl` groups with `map_Kd` textures (boombox_4k.mtl:11,22: the lines `map_Ns`, `map_refl`,
    # `map_Bump`, `map_d` also start with "map_" and must not be taken for it), then the default
    # grey interned for the text before the first `usemtl`, which holds no face
    assert len(s.triangles) == 10168
    assert s.materials["color_kind"].tolist() == [F.RT_COLOR_TEXTURE, F.RT_COLOR_TEXTURE, F.RT_COLOR_SOLID]
    assert [os.path.basename(i[2]) for i in mats.items[:2]] == ["boombox_diff_4k.ppm", "boombox_speakers_diff_4k.ppm"]
    assert np.bincount(s.triangles["material"]).tolist() == [9896, 272]
    assert s.tex_coords.shape[0] == 10168 and s.vertex_normals.shape[0] == 10168
    s, ref, mats = load_both(staged, "hollow knight")
    # material names are Cyrillic; `Kd` colours 0 and 0.8; six `usemtl` lines over three objects
    assert len(s.triangles) == 18352 and np.all(s.materials["color_kind"] == F.RT_COLOR_SOLID)
    assert s.materials["rgb"][:, 0].tolist() == [0.0, np.float32(0.8), 0.5]
    assert np.bincount(s.triangles["material"]).tolist() == [15068, 3284]
    s, ref, mats = load_both(staged, "smooth_uv_sphere")
    assert len(s.triangles) == 960 and len(s.materials) == 1 and s.vertex_normals.shape[0] == 960


def test_scene_file_of_the_reference_is_read(staged):
    # tests/golden/assets/scenes/boombox is /root/reference/scenes/boombox: incremental(10), a
    # 2000 x 2000 window at 1000 samples; the texture size does not matter here
    with open(os.path.join(assets.FIXTURE_DIR, "scenes", "boombox")) as f:
        s = parse(f.read(), staged)
    p = s.params
    assert (p.width, p.height, p.samples_per_pixel, p.max_bounces, p.incremental) == (2000, 2000, 1000, 10, 10)
    assert list(p.camera) == [0.0, 0.0, -40.0] and list(p.top_left) == [-1.0, 1.0, -0.5]
    assert len(s.triangles) == 10168 and len(s.spheres) == 0 and len(s.planes) == 0


def test_parse_f32_rounds_once():
    # a decimal a hair above the midpoint of two binary32 values whose nearest double IS the midpoint
    lo = np.float32(1.0)
    hi = np.nextafter(lo, np.float32(2))
    mid = (float(lo) + float(hi)) / 2
    text = "1.000000059604644775390625" + "0000000000000000001"  # the midpoint's exact expansion, plus a hair
    assert float(text) == mid and obj_ref.parse_f32(text) == hi
    assert obj_ref.parse_f32("1.000000059604644775390625") == lo  # the tie itself goes to the even neighbour
    assert obj_ref.parse_f32("0.072258") == np.float32(0.072258)
