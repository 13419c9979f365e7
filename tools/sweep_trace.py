#!/usr/bin/env python3
"""Interleaved A/B sweep of the traversal kernel's tuning knobs on the bench scene.

  python tools/sweep_trace.py [--spp 4] [--rounds 3] slots=16,12 phase=24,16 chunk=256

Every combination is run `rounds` times, interleaved in one process; prints the
median traversal-kernel time and whole-render time per combination.
"""
import argparse
import itertools
import os
import statistics
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from raytracer_amd import _ffi as F  # noqa: E402
from raytracer_amd.renderer import Renderer  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--spp", type=int, default=4)
ap.add_argument("--rounds", type=int, default=3)
ap.add_argument("--width", type=int, default=1920)
ap.add_argument("--height", type=int, default=1080)
ap.add_argument("--bounces", type=int, default=60)
ap.add_argument("--detail", type=int, default=4)
ap.add_argument("--workload", default="room", choices=["room", "blob", "tree"])
ap.add_argument("--combo", action="append", default=[], help="waves,cache,slots,refill: run exactly these instead of the product")
ap.add_argument("knobs", nargs="*")
args = ap.parse_args()
knobs = {"slots": [6], "phase": [24], "chunk": [256], "inflight": [0], "share": [1], "pairs": [384], "waves": [8], "cache": [144], "refill": [12]}
for k in args.knobs:
    name, vals = k.split("=")
    knobs[name] = [int(v) for v in vals.split(",")]

scene = bench.load_scene(args)
r = Renderer(0)
r.upload(scene)
r.set_option(F.RT_OPT_TIME_KERNELS, 1)
combos = list(itertools.product(knobs["slots"], knobs["phase"], knobs["chunk"], knobs["inflight"], knobs["share"], knobs["pairs"], knobs["waves"], knobs["cache"], knobs["refill"]))
if args.combo:
    combos = []
    for c in args.combo:
        w, k, sl, rf = (int(v) for v in c.split(","))
        combos.append((sl, knobs["phase"][0], knobs["chunk"][0], knobs["inflight"][0], knobs["share"][0], knobs["pairs"][0], w, k, rf))
res = {c: [] for c in combos}
r.render_sums(scene.params, 0, 1)  # warm-up
for rnd in range(args.rounds):
    for c in combos:
        r.set_option(F.RT_OPT_TRACE_LDS_SLOTS, c[0])
        r.set_option(F.RT_OPT_TRACE_PHASE_MIN, c[1])
        r.set_option(F.RT_OPT_TRACE_CHUNK, c[2])
        r.set_option(F.RT_OPT_SAMPLES_IN_FLIGHT, c[3])
        r.set_option(F.RT_OPT_TRACE_SHARE, c[4])
        r.set_option(F.RT_OPT_TRACE_PAIR_MIN, c[5])
        r.set_option(F.RT_OPT_TRACE_WAVES, c[6])
        r.set_option(F.RT_OPT_TRACE_CACHE_NODES, c[7])
        r.set_option(F.RT_OPT_TRACE_REFILL_MIN, c[8])
        r.reset_stats()
        t0 = time.perf_counter()
        r.render_sums(scene.params, 100 * rnd, args.spp)
        dt = time.perf_counter() - t0
        s = r.stats()
        res[c].append((s["trace_ms"], dt * 1e3, s["segments"]))
print("slots phase chunk inflight share pairs waves cache refill | trace_ms  total_ms  Mrays/s")
for c in combos:
    tr = statistics.median(x[0] for x in res[c])
    tot = statistics.median(x[1] for x in res[c])
    seg = res[c][0][2]
    print(f"{c[0]:5d} {c[1]:5d} {c[2]:5d} {c[3]:8d} {c[4]:5d} {c[5]:5d} {c[6]:5d} {c[7]:5d} {c[8]:6d} | {tr:8.1f}  {tot:8.1f}  {seg / tot / 1e3:8.1f}", flush=True)
