#!/usr/bin/env python3
"""Rate of a render in passes of few samples (the reference's shipped scenes use incremental(1)):
rt_render tracing one pass at a time against several passes together (RT_OPT_PASSES_IN_FLIGHT).

  python tools/incremental_rate.py [--width 1920 --height 1080 --spp 64 --incremental 1]
"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from raytracer_amd import _ffi as F  # noqa: E402
from raytracer_amd.renderer import Renderer  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="room")
ap.add_argument("--width", type=int, default=1920)
ap.add_argument("--height", type=int, default=1080)
ap.add_argument("--spp", type=int, default=64)
ap.add_argument("--incremental", type=int, default=1)
a = ap.parse_args()
_, _, _, b = bench.WORKLOADS[a.workload]
args = argparse.Namespace(workload=a.workload, width=a.width, height=a.height, spp=a.spp, bounces=b, detail=4, fast=False)
r = Renderer(0)
scene = bench.load_scene(args, r)
r.upload(scene)
p = scene.params.copy()
p.incremental = a.incremental
images = {}
for label, in_flight in (("one pass at a time", 1), ("passes together (automatic)", 0), ("one pass at a time", 1), ("passes together (automatic)", 0)):
    r.set_option(F.RT_OPT_PASSES_IN_FLIGHT, in_flight)
    img = np.zeros((a.height, a.width, 3), np.uint8)
    r.reset_stats()
    t0 = time.perf_counter()
    r.render(p, img)
    dt = time.perf_counter() - t0
    rays = r.stats()["segments"]
    images.setdefault(in_flight, img)
    print(f"{label}: {a.spp // a.incremental} passes of {a.incremental} spp at {a.width}x{a.height}: {dt * 1e3:.1f} ms, {rays / dt / 1e6:.1f} Mrays/s")
print("same bytes:", bool(np.array_equal(images[0], images[1])))
