#!/usr/bin/env python3
"""Renders 32 samples of the headline frame with 1, 2, 2 and 4 chains per batch and
counts radiance values that differ from the single-chain result (must be 0:
chains only change how launches are scheduled).  Found the stack overflow area
that concurrent traversal launches of two chains once shared.

  python tools/chain_check.py
"""
import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, argparse
import bench
from raytracer_amd import _ffi as F
from raytracer_amd.renderer import Renderer
args = argparse.Namespace(workload="room", width=1920, height=1080, spp=32, bounces=60, detail=4)
r = Renderer(0)
scene = bench.load_scene(args, r)
r.upload(scene)
out = {}
for chains in (1, 2, 2, 4):
    r.set_option(F.RT_OPT_CHAINS, chains)
    out.setdefault(chains, []).append(r.render_sums(scene.params, 0, 32).view(np.uint32))
ref = out[1][0]
for c, arrs in out.items():
    for a in arrs:
        print("chains", c, "differing values:", int(np.count_nonzero(a != ref)))
