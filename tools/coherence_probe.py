#!/usr/bin/env python3
"""How much faster would the traversal be if the rays of a launch were ordered?

Builds rays like a deep bounce's (origins on the scene's surfaces, directions
uniform over the hemisphere that faces back along the ray that found the
origin) by tracing three generations through rt_trace_rays, then times the
traversal of the same third-generation rays in several orders:

  as made     the order they were generated in (origins of neighbours unrelated)
  shuffled    a random permutation
  cell+oct    sorted by (Morton code of the origin's cell of a 2^k grid, direction octant)
  oct+cell    sorted by (direction octant, Morton code)

  python tools/coherence_probe.py [--workload room] [--rays 16000000] [--bits 5]
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from raytracer_amd import _ffi as F  # noqa: E402
from raytracer_amd.renderer import Renderer  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="room", choices=["room", "blob", "tree", "boombox", "knight"])
ap.add_argument("--rays", type=int, default=1 << 24)
ap.add_argument("--bits", type=int, nargs="*", default=[3, 5, 7])
ap.add_argument("--detail", type=int, default=4)
ap.add_argument("--rounds", type=int, default=3)
args = ap.parse_args()
args.width, args.height, args.spp, args.bounces = bench.WORKLOADS[args.workload][:4]

scene = bench.load_scene(args)
r = Renderer(0)
r.upload(scene)
r.set_option(F.RT_OPT_TIME_KERNELS, 1)
rng = np.random.default_rng(7)
n = args.rays


def unit(m):
    v = rng.standard_normal((m, 3)).astype(np.float32)
    return v / np.linalg.norm(v, axis=1, keepdims=True)


cam = np.array([scene.params.camera[i] for i in range(3)], dtype=np.float32)
o = np.broadcast_to(cam, (n, 3)).copy()
d = unit(n)
for gen in range(3):
    hits = r.trace_rays(o, d)
    ok = np.isfinite(hits["t"]) & (hits["kind"] != 0xFFFFFFFF) & (hits["t"] > 0)
    print(f"generation {gen}: {ok.mean():.3f} of the rays hit", flush=True)
    # those that left the scene start again from a hit point of another ray
    src = np.where(ok)[0]
    pick = np.where(ok, np.arange(n), src[rng.integers(0, len(src), n)])
    t = hits["t"][pick][:, None]
    p = o[pick] + d[pick] * (t * np.float32(0.999))
    nd = unit(n)
    facing = np.sum(nd * d[pick], axis=1) > 0  # pointing along the incoming ray = into the surface: turn round
    nd[facing] *= -1
    o, d = p.astype(np.float32), nd.astype(np.float32)

lo, hi = o.min(axis=0), o.max(axis=0)


def morton(cells, bits):
    code = np.zeros(len(cells), dtype=np.uint64)
    for b in range(bits):
        for a in range(3):
            code |= ((cells[:, a] >> b) & 1).astype(np.uint64) << np.uint64(3 * b + a)
    return code


def order_by(bits, octant_first):
    cells = np.minimum(((o - lo) / (hi - lo + 1e-20) * (1 << bits)).astype(np.int64), (1 << bits) - 1)
    m = morton(cells, bits)
    octant = ((d[:, 0] < 0).astype(np.uint64) | ((d[:, 1] < 0).astype(np.uint64) << np.uint64(1)) | ((d[:, 2] < 0).astype(np.uint64) << np.uint64(2)))
    key = (octant << np.uint64(3 * bits)) | m if octant_first else (m << np.uint64(3)) | octant
    return np.argsort(key, kind="stable")


orders = {"as made": np.arange(n), "shuffled": rng.permutation(n)}
for b in args.bits:
    orders[f"cell{b}+oct"] = order_by(b, False)
    orders[f"oct+cell{b}"] = order_by(b, True)
want = None
res = {k: [] for k in orders}
for rnd in range(args.rounds + 1):
    for name, perm in orders.items():
        oo, dd = np.ascontiguousarray(o[perm]), np.ascontiguousarray(d[perm])
        r.reset_stats()
        hits = r.trace_rays(oo, dd)
        s = r.stats()
        if rnd:
            res[name].append(s["trace_ms"])
        back = np.empty_like(hits)
        back[perm] = hits
        if want is None:
            want = back
        assert np.array_equal(back, want), name  # the same rays have the same answers in any order
for name in orders:
    ms = float(np.median(res[name]))
    print(f"{name:12s} {ms:8.2f} ms  {n / ms / 1e3:8.1f} Mrays/s", flush=True)
