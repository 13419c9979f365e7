"""The refined tree (Scene.refined, rth_refine_bvh_triangles) on the GPU.  A refined scene is an
ordinary scene description, so the kernels must agree with the CPU oracle on it bit for bit
(distances, ids in the refined order, radiance, bytes, traversal counters); and against the
reference's tree the hit distances must be the same numbers, the ids equal through `order`
except where two triangles tie."""
import numpy as np
import pytest

from raytracer_amd import _ffi as F

from . import scenes
from .test_gpu_parity import bits, check_scene

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("max_leaf", [2, 8])
def test_refined_scenes_match_the_oracle(renderer, asset_root, max_leaf):
    for s in (scenes.tree(asset_root, w=48, h=48, spp=3), scenes.room(asset_root, w=64, h=36, spp=3, bounces=12)):
        r, _ = s.refined(max_leaf)
        assert r.triangle_nodes.size > s.triangle_nodes.size
        check_scene(renderer, r)


def test_refined_tree_against_the_reference_tree(renderer, asset_root):
    s = scenes.tree(asset_root, w=160, h=160, spp=2, bounces=12)
    r, order = s.refined(4)
    renderer.upload(s)
    a = renderer.trace_primary(s.params, 0, True)
    sums_a = renderer.render_sums(s.params, 0, 2)
    renderer.set_option(F.RT_OPT_COUNT_TRAVERSAL, 1)
    renderer.reset_stats()
    renderer.render_sums(s.params, 0, 1)
    tests_a = renderer.stats()["prim_tests"][2]
    renderer.upload(r)
    renderer.reset_stats()
    renderer.render_sums(s.params, 0, 1)
    tests_b = renderer.stats()["prim_tests"][2]
    renderer.set_option(F.RT_OPT_COUNT_TRAVERSAL, 0)
    b = renderer.trace_primary(s.params, 0, True)
    sums_b = renderer.render_sums(s.params, 0, 2)
    assert np.array_equal(bits(a["t"]), bits(b["t"])) and np.array_equal(a["kind"], b["kind"])
    tri = b["kind"] == F.RT_SHAPE_TRIANGLE
    mapped = b["prim"].copy()
    mapped[tri] = order[b["prim"][tri]]
    assert (mapped != a["prim"]).mean() < 0.005
    assert (bits(sums_a) != bits(sums_b)).mean() < 0.01
    assert tests_b * 3 < tests_a  # what the refinement is for: far fewer triangle tests per ray
