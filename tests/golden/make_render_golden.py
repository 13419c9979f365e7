#!/usr/bin/env python3
"""Writes tests/golden/render_golden.npz: results of the CPU restatement
(oracle/rt_oracle.c) on three small scenes, frozen so that neither the oracle
nor the GPU path can drift unnoticed.

The reference ships no recorded outputs and cannot be built here, so these are
not the reference's numbers: they are this repository's restatement of its
formulas at the commit that wrote them (what pins the restatement itself is in
tests/test_oracle_golden.py).  Regenerate only on purpose:

  python tests/golden/make_render_golden.py
"""
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from oracle import binding as O  # noqa: E402
from raytracer_amd import assets  # noqa: E402
from tests import scenes  # noqa: E402


def golden_scenes(asset_root):
    """name -> scene; the same constructors the tests call."""
    return {
        "sphere_plane": scenes.sphere_plane(w=64, h=36, spp=4, bounces=10),   # the plumbing configuration, reduced
        "materials": scenes.materials(w=32, h=20, spp=4, bounces=12),         # every material kind, all three families
        "blob": scenes.blob(asset_root, w=24, h=24, spp=2, bounces=6),        # mesh with vertex normals behind a real tree
    }


def main():
    root = tempfile.mkdtemp(prefix="golden_assets_")
    assets.write_blob(root, "blob", subdivisions=3)
    out = {}
    for name, scene in golden_scenes(root).items():
        p = scene.params
        hits = O.trace_primary(scene, p, 1, True)
        out[f"{name}.hit_kind"] = hits["kind"].astype(np.uint8)
        out[f"{name}.hit_prim"] = hits["prim"]
        out[f"{name}.hit_t_bits"] = hits["t"].view(np.uint32)
        out[f"{name}.sums_bits"] = O.render_sums(scene, p, 0, p.samples_per_pixel).view(np.uint32)
        out[f"{name}.rgb8"] = O.render(scene, p)[0]
        if name == "blob":
            out["blob.nodes"] = scene.triangle_nodes.view(np.uint8)  # the builder's array, node for node
    np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "render_golden.npz"), **out)
    print({k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
