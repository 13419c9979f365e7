#!/usr/bin/env python3
"""Do two contexts on one GPU overlap the VALU-bound traversal of one with the
HBM-bound shading of the other?  Renders the bench scene's samples on one
context, then split over two contexts driven by two host threads, for several
shares of the compute units' wave slots given to each context's traversal.

  python tools/overlap_probe.py [--spp 32]
"""
import argparse
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from raytracer_amd import _ffi as F  # noqa: E402
from raytracer_amd.renderer import Renderer  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--spp", type=int, default=32)
ap.add_argument("--width", type=int, default=1920)
ap.add_argument("--height", type=int, default=1080)
ap.add_argument("--bounces", type=int, default=60)
ap.add_argument("--detail", type=int, default=4)
ap.add_argument("--workload", default="room")
args = ap.parse_args()
scene = bench.load_scene(args)
ctxs = [Renderer(0), Renderer(0)]
for r in ctxs:
    r.upload(scene)
    r.render_sums(scene.params, 0, 1)


def run(r, begin, count, in_flight):
    r.set_option(F.RT_OPT_SAMPLES_IN_FLIGHT, in_flight)
    r.render_sums(scene.params, begin, count)


def timed(n_ctx, waves_per_cu, in_flight):
    for r in ctxs:
        r.set_option(F.RT_OPT_TRACE_WAVES_PER_CU, waves_per_cu)
        r.reset_stats()
    share = args.spp // n_ctx
    threads = [threading.Thread(target=run, args=(ctxs[i], i * share, share, in_flight)) for i in range(n_ctx)]
    t0 = time.perf_counter()
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    dt = time.perf_counter() - t0
    rays = sum(ctxs[i].stats()["segments"] for i in range(n_ctx))
    return rays / dt / 1e6, dt


for label, n_ctx, wpc, inflight in (("one context, all waves", 1, 0, 0), ("one context, all waves", 1, 0, 0),
                                    ("one context, 16 in flight", 1, 0, 16),
                                    ("two contexts, all waves each", 2, 0, 16), ("two contexts, 16 waves/CU each", 2, 16, 16),
                                    ("two contexts, 24 waves/CU each", 2, 24, 16), ("two contexts, 8 waves/CU each", 2, 8, 16),
                                    ("one context, all waves", 1, 0, 0)):
    mrays, dt = timed(n_ctx, wpc, inflight)
    print(f"{label:34s} {mrays:8.1f} Mrays/s  ({dt * 1e3:.0f} ms)", flush=True)
