#!/usr/bin/env python3
"""Render time by batch size for the two bounce loops (one launch per batch, RT_OPT_FUSED_LOOP = 1,
against one launch per bounce, = 0) on the bench scene: a 256 x 256 window and the whole 1080p
frame at 1 .. 32 samples per pixel, best of three.  profiles/r02_fused_loop.txt is its output.

  python tools/batch_size_sweep.py
"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import argparse
import bench
from raytracer_amd import _ffi as F
from raytracer_amd.renderer import Renderer
args = argparse.Namespace(workload="room", width=1920, height=1080, spp=4, bounces=60, detail=4, tile=4096)
scene = bench.load_scene(args)
r = Renderer(0); r.upload(scene)
p = scene.params
for spp in (1, 2, 4, 8, 16, 32):
    for w, h in ((256, 256), (1920, 1080)):
        if w == 256 and spp > 8: continue
        q = p.copy()
        q.tile_x, q.tile_y, q.tile_w, q.tile_h = 800, 400, w, h
        if w == 1920: q.tile_x = q.tile_y = 0; q.tile_w = q.tile_h = 0
        res = {}
        for mode in (0, 1):
            r.set_option(F.RT_OPT_FUSED_LOOP, mode)
            r.render_sums(q, 0, spp)
            best = 1e9
            for _ in range(3):
                r.reset_stats(); t0 = time.perf_counter(); r.render_sums(q, 0, spp); dt = time.perf_counter() - t0
                best = min(best, dt)
            res[mode] = (best, r.stats()["segments"])
        n = w * h * spp
        print(f"{w}x{h} spp {spp:2d} paths {n:9d}: plain {res[0][0]*1e3:8.2f} ms  fused {res[1][0]*1e3:8.2f} ms   ({res[0][1]/res[0][0]/1e6:7.1f} vs {res[1][1]/res[1][0]/1e6:7.1f} Mrays/s)", flush=True)
