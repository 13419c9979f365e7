#!/usr/bin/env python3
"""Does a GPU whose rays all come from one band of the image render them faster than rays from
the whole frame (the tree then fits the L2 slices better)?  Renders a workload's frame whole and
as 8 horizontal bands, one after the other, and compares the summed times.

  python tools/band_locality.py [--workload tree] [--spp 32]
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from raytracer_amd.renderer import Renderer  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="tree")
ap.add_argument("--spp", type=int, default=None)
ap.add_argument("--bands", type=int, default=8)
a = ap.parse_args()
w, h, spp, b = bench.WORKLOADS[a.workload]
args = argparse.Namespace(workload=a.workload, width=w, height=h, spp=a.spp or spp, bounces=b, detail=4, fast=False)
r = Renderer(0)
scene = bench.load_scene(args, r)
r.upload(scene)
p = scene.params


def timed(params):
    r.render_sums(params, 0, args.spp)  # warm-up (workspace)
    r.reset_stats()
    t0 = time.perf_counter()
    r.render_sums(params, 0, args.spp)
    dt = time.perf_counter() - t0
    return dt, r.stats()["segments"]


whole_t, whole_rays = timed(p)
print(f"whole frame: {whole_t * 1e3:.1f} ms, {whole_rays / whole_t / 1e6:.1f} Mrays/s")
total_t, total_rays = 0.0, 0
for k in range(a.bands):
    q = p.copy()
    q.tile_x, q.tile_w = 0, w
    q.tile_y = h * k // a.bands
    q.tile_h = h * (k + 1) // a.bands - q.tile_y
    dt, rays = timed(q)
    total_t += dt
    total_rays += rays
    print(f"band {k}: {dt * 1e3:.1f} ms, {rays / dt / 1e6:.1f} Mrays/s ({rays} rays)")
print(f"bands together: {total_t * 1e3:.1f} ms, {total_rays / total_t / 1e6:.1f} Mrays/s; rays {total_rays} vs {whole_rays}")
