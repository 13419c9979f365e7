#!/usr/bin/env python3
"""Construction time of the triangle BVH: rt_build_bvh_triangles (GPU) beside
rth_build_bvh_triangles (host), on the generated workloads, with the trees
compared node for node.  One JSON line per workload.

  python tools/bench_bvh_build.py [--repeat 5]
"""
import argparse
import ctypes as C
import json
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from raytracer_amd import _ffi as F  # noqa: E402
from raytracer_amd import assets, parse  # noqa: E402
from raytracer_amd.renderer import Renderer  # noqa: E402
from raytracer_amd.scene import NODE_DT  # noqa: E402


def host_build(tris):
    lib = F.host_lib()
    t = tris.copy()
    n = F.u32(0)
    t0 = time.perf_counter()
    raw = lib.rth_build_bvh_triangles(t.ctypes.data, t.size, C.byref(n))
    dt = time.perf_counter() - t0
    buf = (C.c_char * (n.value * NODE_DT.itemsize)).from_address(raw)
    nodes = np.frombuffer(buf, dtype=NODE_DT, count=n.value).copy()
    lib.rth_free(raw)
    return nodes, t, dt


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--repeat", type=int, default=5)
    args = ap.parse_args()
    root = tempfile.mkdtemp(prefix="bvh_bench_")
    assets.write_room(root, "room", detail=4)
    assets.write_blob(root, "blob", subdivisions=6)
    assets.write_tree(root, "tree", n_leaves=500_000)
    texts = {"room": assets.room_scene_text("room", 16, 16, 1, 2), "blob": assets.blob_scene_text("blob", 16, 16, 1, 2),
             "tree": assets.tree_scene_text("tree", 16, 16, 1, 2)}
    r = Renderer(0)
    for name, text in texts.items():
        tris = parse(text, root, renderer=r).triangles.copy()
        np.random.default_rng(1).shuffle(tris)  # file order is not special; any order gives a valid input
        r.build_bvh_triangles(tris[:4096])      # kernels loaded
        best = None
        for _ in range(args.repeat):
            t0 = time.perf_counter()
            g_nodes, g_tris = r.build_bvh_triangles(tris)
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        c_nodes, c_tris, host_s = host_build(tris)
        same = (np.array_equal(g_nodes["a"], c_nodes["a"]) and np.array_equal(g_nodes["b"], c_nodes["b"]) and
                np.array_equal(g_nodes["min"], c_nodes["min"]) and np.array_equal(g_nodes["max"], c_nodes["max"]) and
                g_tris.tobytes() == c_tris.tobytes())
        print(json.dumps({"workload": name, "triangles": int(tris.size), "nodes": int(len(g_nodes)),
                          "gpu_ms": round(best * 1e3, 2), "host_ms": round(host_s * 1e3, 1),
                          "speedup": round(host_s / best, 1), "mtris_per_s_gpu": round(tris.size / best / 1e6, 2),
                          "identical_tree_and_permutation": bool(same),
                          "note": "GPU time includes the copy of the triangles to the device, the node array and "
                                  "permutation back, and the host-side depth-first renumbering and permutation"}),
              flush=True)
    r.close()


if __name__ == "__main__":
    main()
