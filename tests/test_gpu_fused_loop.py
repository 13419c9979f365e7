"""The whole bounce loop as one launch (RT_OPT_FUSED_LOOP, kernel k_path): every wave takes
its own block of paths through all bounces.  Each path goes through the operations of the
launch-per-bounce loop, so radiance sums, stored bytes and the ray / hit counts must be the
oracle's, bit for bit — with one wave's worth of paths, with blocks that end in the middle of
a wave, with deep stacks that overflow to memory, with textures and every material."""
import numpy as np
import pytest

from oracle import binding as O
from raytracer_amd import _ffi as F
from raytracer_amd import assets, parse

from . import scenes
from .test_gpu_parity import bits

pytestmark = pytest.mark.gpu


def check_fused(renderer, scene, spp=None):
    p = scene.params
    spp = spp or p.samples_per_pixel
    renderer.upload(scene)
    c = O.OracleCounters()
    cpu = O.render_sums(scene, p, 0, spp, threads=4, counters=c)
    renderer.set_option(F.RT_OPT_FUSED_LOOP, 1)
    try:
        renderer.reset_stats()
        gpu = renderer.render_sums(p, 0, spp)
        s = renderer.stats()
        img = renderer.render(p)
    finally:
        renderer.set_option(F.RT_OPT_FUSED_LOOP, 2)
    assert np.array_equal(bits(gpu), bits(cpu)), f"radiance differs in {np.count_nonzero(bits(gpu) != bits(cpu))} values"
    assert s["paths"] == c.paths and s["segments"] == c.segments and s["hits"] == c.hits
    assert np.array_equal(img, O.render(scene, p, threads=4)[0])
    # and the launch-per-bounce loop gives the same
    renderer.set_option(F.RT_OPT_FUSED_LOOP, 0)
    try:
        assert np.array_equal(bits(renderer.render_sums(p, 0, spp)), bits(gpu))
    finally:
        renderer.set_option(F.RT_OPT_FUSED_LOOP, 2)


def test_textured_room_with_all_materials(renderer, asset_root):
    check_fused(renderer, scenes.room(asset_root, w=64, h=36, spp=5, bounces=20))


def test_odd_sizes_and_one_wave(renderer, asset_root):
    check_fused(renderer, scenes.room(asset_root, w=7, h=9, spp=1, bounces=12))    # 63 paths: one wave, one lane idle
    check_fused(renderer, scenes.room(asset_root, w=13, h=5, spp=3, bounces=12))   # 195 paths
    check_fused(renderer, scenes.blob(asset_root, w=61, h=47, spp=3, bounces=10))  # vertex normals; blocks end mid-wave


def test_tree_and_deep_stacks(renderer, asset_root):
    check_fused(renderer, scenes.tree(asset_root, w=48, h=48, spp=3, bounces=12))
    check_fused(renderer, scenes.deep_chain(n=60))        # stack deeper than its LDS part
    check_fused(renderer, scenes.stacked_duplicates(n=40))  # exact ties
    check_fused(renderer, scenes.grid_wall(n=16))


def test_one_bounce_and_bounce_limit(renderer, asset_root):
    check_fused(renderer, scenes.room(asset_root, w=32, h=20, spp=2, bounces=1))
    check_fused(renderer, scenes.room(asset_root, w=32, h=20, spp=2, bounces=3))


def test_scenes_it_does_not_apply_to_take_the_other_loop(renderer):
    # spheres and planes only (no triangles), and a sphere family with a tree of its own
    check_fused(renderer, scenes.materials(w=40, h=24, spp=3, bounces=12))
    check_fused(renderer, scenes.many_spheres(n=300, n_planes=20, w=48, h=32, spp=2, bounces=8))


def test_reference_mesh(renderer, tmp_path_factory):
    root = str(tmp_path_factory.mktemp("reference_assets_fused"))
    assets.stage_reference_assets(root, texture_size=256)
    check_fused(renderer, parse(assets.reference_scene_text("hollow knight", 96, 96, 3, 10), root))


def test_mid_size_batch_takes_the_fused_loop_by_default(renderer, asset_root):
    # batches below 3 * 2^22 paths go through k_path without being asked; compare with the launch-per-bounce loop
    s = scenes.room(asset_root, w=1024, h=512, spp=5, bounces=16)
    renderer.upload(s)
    renderer.reset_stats()
    fused = renderer.render_sums(s.params, 0, 5)
    st_fused = renderer.stats()
    renderer.set_option(F.RT_OPT_FUSED_LOOP, 0)
    try:
        renderer.reset_stats()
        plain = renderer.render_sums(s.params, 0, 5)
        st_plain = renderer.stats()
    finally:
        renderer.set_option(F.RT_OPT_FUSED_LOOP, 2)
    assert np.array_equal(bits(fused), bits(plain))
    assert (st_fused["segments"], st_fused["hits"]) == (st_plain["segments"], st_plain["hits"])
    # oracle on a tile of it
    tile = s.params.copy()
    tile.tile_x, tile.tile_y, tile.tile_w, tile.tile_h = 480, 240, 64, 32
    assert np.array_equal(bits(fused[240:272, 480:544]), bits(O.render_sums(s, tile, 0, 5, threads=8)))
