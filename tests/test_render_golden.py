"""Frozen results (tests/golden/render_golden.npz, written by
tests/golden/make_render_golden.py from the CPU restatement): the oracle must
still produce them, and so must the GPU — hit for hit, bit for bit."""
import os

import numpy as np
import pytest

from oracle import binding as O

from .golden.make_render_golden import golden_scenes

GOLDEN = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "render_golden.npz"), allow_pickle=False)
NAMES = ["sphere_plane", "materials", "blob"]


def check(name, hits, sums, rgb8):
    assert np.array_equal(hits["kind"].astype(np.uint8), GOLDEN[f"{name}.hit_kind"])
    assert np.array_equal(hits["prim"], GOLDEN[f"{name}.hit_prim"])
    assert np.array_equal(hits["t"].view(np.uint32), GOLDEN[f"{name}.hit_t_bits"])
    assert np.array_equal(np.ascontiguousarray(sums).view(np.uint32), GOLDEN[f"{name}.sums_bits"])
    assert np.array_equal(rgb8, GOLDEN[f"{name}.rgb8"])


@pytest.mark.parametrize("name", NAMES)
def test_oracle_reproduces_the_frozen_results(asset_root, name):
    scene = golden_scenes(asset_root)[name]
    p = scene.params
    check(name, O.trace_primary(scene, p, 1, True), O.render_sums(scene, p, 0, p.samples_per_pixel), O.render(scene, p)[0])
    if name == "blob":
        assert scene.triangle_nodes.view(np.uint8).tobytes() == GOLDEN["blob.nodes"].tobytes()


@pytest.mark.gpu
@pytest.mark.parametrize("name", NAMES)
def test_gpu_reproduces_the_frozen_results(renderer, asset_root, name):
    scene = golden_scenes(asset_root)[name]
    p = scene.params
    renderer.upload(scene)
    check(name, renderer.trace_primary(p, 1, True), renderer.render_sums(p, 0, p.samples_per_pixel), renderer.render(p))
