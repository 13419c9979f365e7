#!/usr/bin/env python3
"""Soak: renders the headline frame over and over through each bounce loop (two chains side by side, one chain,
the one-launch loop) and checks that every render gives the same bits — a data race between launches
that run side by side would show as a difference once in a while.

  python tools/soak.py [rounds]
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from raytracer_amd import _ffi as F  # noqa: E402
from raytracer_amd import assets, parse  # noqa: E402
from raytracer_amd.renderer import Renderer  # noqa: E402

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 12
root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "assets_cache", "soak")
assets.write_room(root, "room", detail=4)
scene = parse(assets.room_scene_text("room", 1920, 1080, 16, 60), root)
with Renderer(0) as r:
    r.upload(scene)
    modes = {"two chains": (F.RT_OPT_FUSED_LOOP, 0, 2, 16), "one chain": (F.RT_OPT_FUSED_LOOP, 0, 1, 16),
             "one launch": (F.RT_OPT_FUSED_LOOP, 1, 2, 4)}
    ref = {}
    t0 = time.time()
    for k in range(rounds):
        for name, (opt, val, chains, spp) in modes.items():
            r.set_option(opt, val)
            r.set_option(F.RT_OPT_CHAINS, chains)
            got = r.render_sums(scene.params, 3, spp).view(np.uint32)
            if name not in ref:
                ref[name] = got.copy()
            elif not np.array_equal(got, ref[name]):
                print(f"round {k}, {name}: {np.count_nonzero(got != ref[name])} values differ", flush=True)
                sys.exit(1)
        print(f"round {k} ok ({time.time() - t0:.0f} s)", flush=True)
    # the loops agree with each other on the samples they share
    r.set_option(F.RT_OPT_FUSED_LOOP, 0)
    assert np.array_equal(r.render_sums(scene.params, 3, 4).view(np.uint32), ref["one launch"])

    # short passes: rt_render traces the paths of several passes together and finishes them in order; the
    # loop is the library's own choice (measured on the first large batches after an upload)
    r.set_option(F.RT_OPT_FUSED_LOOP, 2)
    r.set_option(F.RT_OPT_CHAINS, 2)
    p = scene.params.copy()
    p.samples_per_pixel, p.incremental = 24, 1
    want = None
    for k in range(rounds):
        r.upload(scene)  # (forgets the measured loop: every round measures again)
        got = r.render(p)
        if want is None:
            r.set_option(F.RT_OPT_PASSES_IN_FLIGHT, 1)
            want = r.render(p)  # one pass at a time
            r.set_option(F.RT_OPT_PASSES_IN_FLIGHT, 0)
        if not np.array_equal(got, want):
            print(f"short passes, round {k}: {np.count_nonzero(got != want)} bytes differ", flush=True)
            sys.exit(1)
    print(f"short passes ok ({time.time() - t0:.0f} s)", flush=True)

# the same frame over a list of GPUs (here: this GPU twice): bands of rows for short passes, samples dealt out for long ones
from raytracer_amd.renderer import RendererGroup  # noqa: E402

with RendererGroup([0, 0]) as g:
    g.upload(scene)
    for k in range(max(2, rounds // 3)):
        if not np.array_equal(g.render(p), want):
            print(f"group, short passes, round {k}: bytes differ", flush=True)
            sys.exit(1)
    long_pass = scene.params.copy()
    long_pass.samples_per_pixel, long_pass.incremental = 32, 0
    first = g.render(long_pass)
    for k in range(max(2, rounds // 3)):
        if not np.array_equal(g.render(long_pass), first):
            print(f"group, one long pass, round {k}: bytes differ", flush=True)
            sys.exit(1)
    print(f"group ok ({time.time() - t0:.0f} s)", flush=True)
print("soak ok")
