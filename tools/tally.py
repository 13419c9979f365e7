#!/usr/bin/env python3
"""Diagnostic tallies of the triangle traversal (how full its steps run, and where in the
tree its branch visits land) on a bench workload, from the counting kernels:

  python tools/tally.py [--workload room] [--spp 2]
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["RT_MI355_DEBUG_TALLIES"] = "1"
import bench  # noqa: E402
from raytracer_amd import _ffi as F  # noqa: E402
from raytracer_amd.renderer import Renderer  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="room")
ap.add_argument("--width", type=int, default=1920)
ap.add_argument("--height", type=int, default=1080)
ap.add_argument("--spp", type=int, default=2)
ap.add_argument("--bounces", type=int, default=60)
ap.add_argument("--detail", type=int, default=4)
args = ap.parse_args()
scene = bench.load_scene(args)
r = Renderer(0)
r.upload(scene)
r.set_option(F.RT_OPT_COUNT_TRAVERSAL, 1)
r.reset_stats()
r.render_sums(scene.params, 0, args.spp)
print(r.stats())
r.reset_stats()  # prints the tallies to stderr
