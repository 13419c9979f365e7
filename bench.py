#!/usr/bin/env python3
"""Headline benchmark: Mrays/s (all bounces) on the breakfast-room-class scene
at 1920x1080, 256 samples per pixel, 60 bounces.

  python bench.py --gpus N --steps K --warmup W

A step is one complete render of the frame: every rank renders its share of the
256 samples of every pixel (rt_render_sums into a device buffer), the radiance
buffers are summed onto rank 0 (RCCL reduce; nothing at N=1) and rank 0 applies
mean/gamma/quantisation and reads the bytes back (rt_resolve).  Scene parsing
and BVH construction happen once before the timed region; the scene is
resident in HBM when timing starts.  A "ray" is one iteration of the bounce
loop (/root/reference/src/lib.rs:313).  The breakfast_room mesh itself is not
in the reference checkout, so the scene is the generated stand-in of
raytracer_amd.assets.write_room (same camera as scenes/breakfast-room).

Rank 0 prints one JSON line.

Several GPUs.  Under torch.distributed.run (RANK / WORLD_SIZE set) every process is
one rank on its own GPU.  Started plainly with --gpus N > 1, this program starts
its N ranks itself, as the reference starts its own workers inside the one call
(/root/reference/src/lib.rs:195-219): before anything has touched a GPU it runs
`python -m torch.distributed.run --nproc-per-node N bench.py ...` as a child
process, passes rank 0's JSON line through and exits with the child's status.
--backend group renders the same step inside ONE process instead, through the
library's own GPU list (rt_group_*: one context and one host thread per GPU, the
sums handed from GPU to GPU in sample order, no collective library involved).
"""
from __future__ import annotations

import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from raytracer_amd import _ffi as F  # noqa: E402
from raytracer_amd import assets, parse  # noqa: E402
from raytracer_amd.distributed import GpuSumsBackend, render_pass, render_pass_exact  # noqa: E402
from raytracer_amd.scene import Scene  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E, spec


VALU_PEAK_GINSTR = 1024 * 2.4 / 2.0  # 1024 SIMDs x 2.4 GHz, one wave64 vector instruction per 2 cycles: 1228.8 G/s
PROFILE = os.path.join(ROOT, "profiles", "r03_counters.json")


def _code_only(text: str) -> str:
    """C++ / make source without comments and without blank space (string literals kept)."""
    out, i, n = [], 0, len(text)
    while i < n:
        c = text[i]
        if c == '"' or c == "'":  # a literal: copy through its closing quote
            j = i + 1
            while j < n and text[j] != c:
                j += 2 if text[j] == "\\" else 1
            out.append(text[i:j + 1])
            i = j + 1
        elif text.startswith("//", i):
            i = text.find("\n", i) if text.find("\n", i) >= 0 else n
        elif text.startswith("/*", i):
            j = text.find("*/", i + 2)
            i = n if j < 0 else j + 2
        elif c.isspace():
            i += 1
        else:
            out.append(c)
            i += 1
    return "".join(out)


def kernel_source_hash() -> str:
    """Identifies the device code a counter profile belongs to: a hash over the
    sources of the render loop's kernels and of their launches, and the build flags
    (the Makefile) — the code only: comments and blank space do not count."""
    import hashlib

    h = hashlib.sha256()
    base = os.path.join(ROOT, "raytracer_amd", "csrc")
    for path in [os.path.join(base, "device", n) for n in ("kernels.h", "layout.h", "rt_context.hip")] + [os.path.join(base, "Makefile")]:
        with open(path, encoding="utf-8") as f:
            text = f.read()
        if path.endswith("Makefile"):
            text = "\n".join(line.split("#", 1)[0].rstrip() for line in text.splitlines())
        else:
            text = _code_only(text)
        h.update(os.path.basename(path).encode() + b"\0" + text.encode())
    return h.hexdigest()[:16]


def recorded_counters(args):
    """Hardware counters of the traversal kernel, per launch and per ray, from the
    counter passes recorded in profiles/ (tools/pmc_profile.sh + tools/pmc_record.py;
    counters cannot be collected inside a timed run).  Only a record made from these
    very kernel sources and for this workload is used: otherwise None."""
    try:
        with open(PROFILE) as f:
            rec = json.load(f)
    except (OSError, ValueError):
        return None
    if rec.get("source_hash") != kernel_source_hash():
        return None
    key = f"{args.workload}:{args.width}x{args.height}:b{args.bounces}:d{args.detail}:fast{int(args.fast)}"
    if getattr(args, "fast_build", 0):
        key += f":leaf{args.fast_build}"
    return rec.get("workloads", {}).get(key)


WORKLOADS = {
    # name: (width, height, spp, bounces) defaults and what the scene is
    "room": (1920, 1080, 256, 60),   # the metric's configuration (breakfast-room class)
    "blob": (1920, 1080, 64, 10),    # bunny class: smooth mesh with vertex normals, one grey material
    "tree": (2000, 2000, 32, 50),    # island-tree class: a million thin overlapping quads, deep tree
    # the room at 16384 x 16384 into the memory-mapped image file, tile by tile; a step is one
    # more one-sample pass over the whole image, resumed from the file (`continue`)
    "room16k": (16384, 16384, 1, 60),
    # meshes the reference's own checkout holds (tests/golden/assets): boombox_4k.obj with the camera of
    # scenes/boombox and generated 4096^2 textures in place of the two .ppm files the checkout lacks
    "boombox": (2000, 2000, 64, 10),
    "knight": (1920, 1080, 64, 10),  # "hollow knight.obj", framed from its bounding box (no scene file exists for it)
}


def scene_text(args) -> tuple[str, str]:
    """(scene-file text, one-line description) of the chosen workload."""
    if args.workload == "room16k":
        text = assets.room_scene_text("room", args.width, args.height, args.spp, args.bounces, incremental=args.spp)
        text = text.replace("1.92 0 0, 0 -1.08 0", "1.92 0 0, 0 -1.92 0")  # square window for the square image
        return (text, "breakfast-room-class generated room ({n} triangles) at {w}x{h} into the memory-mapped image, "
                      "{t}x{t} tiles, {spp} spp per pass, {b} bounces, every step resumes from the file")
    if args.workload == "room":
        return (assets.room_scene_text("room", args.width, args.height, args.spp, args.bounces),
                "breakfast-room-class generated room ({n} triangles, 5 ppm textures, metal/glass/light spheres), "
                "{w}x{h}, {spp} spp, {b} bounces, camera of scenes/breakfast-room")
    if args.workload == "blob":
        return (assets.blob_scene_text("blob", args.width, args.height, args.spp, args.bounces),
                "bunny-class generated smooth mesh ({n} triangles, vertex normals, default grey Lambertian), "
                "{w}x{h}, {spp} spp, {b} bounces, camera of scenes/bunny")
    if args.workload == "boombox":
        return (assets.reference_scene_text("boombox_4k", args.width, args.height, args.spp, args.bounces),
                "the reference's obj/boombox_4k.obj ({n} triangles, two usemtl groups, generated 4096x4096 P6 textures "
                "under the names its .mtl gives), {w}x{h}, {spp} spp, {b} bounces, screen and camera of scenes/boombox")
    if args.workload == "knight":
        return (assets.reference_scene_text("hollow knight", args.width, args.height, args.spp, args.bounces),
                "the reference's obj/hollow knight.obj ({n} triangles, vertex normals, solid Kd materials) on a ground plane, "
                "{w}x{h}, {spp} spp, {b} bounces, framed from its bounding box")
    return (assets.tree_scene_text("tree", args.width, args.height, args.spp, args.bounces),
            "island-tree-class generated tree ({n} triangles: overlapping leaf quads, textured island and bark), "
            "{w}x{h}, {spp} spp, {b} bounces, camera of scenes/island-tree")


def load_scene(args, renderer=None) -> Scene:
    """Generate (once) and parse the workload's scene.  The parsed arrays are cached
    next to the generated OBJ so that several ranks do not each rebuild the BVH."""
    root = os.path.join(ROOT, "assets_cache", f"{'room' if args.workload == 'room16k' else args.workload}_d{args.detail}")
    cache = os.path.join(root, "scene_v1.npz")
    text, _ = scene_text(args)
    fields = ("spheres", "planes", "triangles", "vertex_normals", "tex_coords", "bary_precomputed", "materials",
              "texels", "sphere_nodes", "plane_nodes", "triangle_nodes")
    if os.path.exists(cache):
        z = np.load(cache, allow_pickle=False)
        # the screen/camera block comes from the text; the obj(...) line is dropped so nothing is re-read
        params = parse("\n".join(l for l in text.splitlines() if not l.startswith("obj(")) + "\n", root).params
        return Scene(params, *[z[k] for k in fields])
    t0 = time.time()
    if args.workload in ("room", "room16k") and not os.path.exists(os.path.join(root, "obj", "room", "room.obj")):
        assets.write_room(root, "room", detail=args.detail)
    elif args.workload == "blob" and not os.path.exists(os.path.join(root, "obj", "blob.obj")):
        assets.write_blob(root, "blob", subdivisions=min(8, 2 + args.detail))   # detail 4: 81,920 triangles
    elif args.workload in ("boombox", "knight"):
        assets.stage_reference_assets(root, texture_size=4096)
    elif args.workload == "tree" and not os.path.exists(os.path.join(root, "obj", "tree.obj")):
        assets.write_tree(root, "tree", n_leaves=125_000 * args.detail)          # detail 4: 1.04e6 triangles
    scene = parse(text, root, renderer=renderer)  # triangle BVH built on the GPU (same tree as the host builder's)
    tmp = cache + f".{os.getpid()}.tmp.npz"
    np.savez(tmp, **{k: getattr(scene, k) for k in fields})
    os.replace(tmp, cache)
    print(f"[bench] scene generated, parsed and BVH built in {time.time() - t0:.1f}s "
          f"({len(scene.triangles)} triangles, {len(scene.triangle_nodes)} nodes)", file=sys.stderr)
    return scene


def cpu_baseline(scene: Scene, args) -> dict:
    """The CPU restatement of the reference's multithreaded renderer
    (oracle/rt_oracle.c: T x T pixel chunks off an atomic counter, worker i pinned
    to core i, /root/reference/src/lib.rs:195-235), compiled here for this machine
    (-O3 -march=native, no FMA contraction, no fast-math) and timed on all the
    cores this process may use, on a bounded sample of the same frame: every k-th
    row of the whole image, so that the mix of paths is the frame's."""
    import tempfile

    from oracle import binding as O

    allowed = len(os.sched_getaffinity(0))
    share = int(os.environ.get("OMP_NUM_THREADS", "0"))  # the box's CPU share for this job, when it states one
    threads = min(allowed, share) if share > 0 else allowed
    native = O.build_native(tempfile.mkdtemp(prefix="rt_oracle_native_"))
    library = O.lib(native) if native else O.lib()
    p = scene.params.copy()
    p.tile_x = p.tile_y = p.tile_w = p.tile_h = 0
    p.has_continue = p.continue_sampling = p.incremental = 0
    stride = max(1, args.height // 60)  # about 60 rows spread over the frame
    t0 = time.perf_counter()
    O.render_sums_rows(scene, p, 0, 1, stride, threads=threads, library=library)
    probe = time.perf_counter() - t0
    spp = int(max(1, min(256, round(15.0 / max(probe, 1e-3)))))  # aim at about 15 s of CPU work
    c = O.OracleCounters()
    t0 = time.perf_counter()
    O.render_sums_rows(scene, p, 1, spp, stride, threads=threads, counters=c, library=library)
    dt = time.perf_counter() - t0
    rows = (args.height + stride - 1) // stride
    return {"value": round(c.segments / dt / 1e6, 3), "unit": "Mrays/s", "cores": threads, "kind": "port",
            "build": "gcc -O3 -march=native -ffp-contract=off, compiled on this host" if native
                     else "prebuilt -O2 library (no compiler on this host)",
            "host_cpus_visible": os.cpu_count(), "cpus_allowed": allowed,
            "sample": f"every {stride}th row of the same frame ({rows} rows x {args.width} pixels), {spp} spp, "
                      f"{args.bounces} bounces: {c.segments} rays in {dt:.1f}s on {threads} pinned threads",
            "rays_per_sample": round(c.segments / max(c.paths, 1), 3)}


def launch_ranks(n: int, argv: list[str]) -> int:
    """Start this program's N ranks as a child process group (torch.distributed.run on
    127.0.0.1, a free port) and return its exit status.  Nothing in this process has
    touched a GPU or initialised HIP: the parent only waits.  Rank 0's JSON line reaches
    stdout through the inherited descriptor."""
    import socket
    import subprocess

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    print(f"[bench] starting {n} ranks: {' '.join(cmd)}", file=sys.stderr, flush=True)
    return subprocess.run(cmd, env=dict(os.environ)).returncode


def main_group(args) -> None:
    """--backend group: the step rendered by ONE process through the library's GPU list
    (RendererGroup = rt_group_*): every GPU traces its contiguous piece of the pass's samples
    on a host thread of its own, the running sums go from GPU to GPU in sample order, the
    last holder resolves.  The stored bytes are the single-GPU bytes."""
    import torch  # (before the library: one HIP runtime per process)

    from raytracer_amd.renderer import Renderer, RendererGroup

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the render loop has no CPU path")
    if args.workload == "room16k":
        raise SystemExit("--backend group renders whole frames; the tiled 16K workload runs with --backend ranks")
    n = args.gpus
    ordinals = [0] * n if args.rehearse_on_one_gpu else list(range(n))
    if not args.rehearse_on_one_gpu and torch.cuda.device_count() < n:
        raise SystemExit(f"--gpus {n} but only {torch.cuda.device_count()} visible")
    with Renderer(0) as builder:
        scene = load_scene(args, builder)
    g = RendererGroup(ordinals)
    g.upload(scene)
    chains = 2 if args.chains is None else args.chains
    g.set_option(F.RT_OPT_CHAINS, chains)
    g.set_option(F.RT_OPT_TRACE_WAVES_PER_CU, args.trace_waves_per_cu)
    if args.shade_groups is not None:
        g.set_option(F.RT_OPT_SHADE_GROUPS_PER_CU, args.shade_groups)
    for kv in args.opt:
        name, value = kv.split("=")
        g.set_option(getattr(F, "RT_OPT_" + name), int(value))
    if args.fast:
        g.set_option(F.RT_OPT_FAST_SLAB, 1)
    params = scene.params
    rgb8 = np.zeros((args.height, args.width, 3), dtype=np.uint8)

    def sync_all():
        for d in set(ordinals):
            torch.cuda.synchronize(d)

    for _ in range(args.warmup):
        g.render(params, rgb8)
    g.set_option(F.RT_OPT_TIME_KERNELS, 1)
    g.reset_stats()
    sync_all()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        g.render(params, rgb8)
    sync_all()
    dt = time.perf_counter() - t0
    st = g.stats()
    before = rgb8.copy()
    g.render(params, rgb8)
    rays_per_s = st["segments"] / dt
    out = {
        "metric": "Mrays/s (all bounces) breakfast-room 1080p" if args.workload == "room" else f"Mrays/s (all bounces) {args.workload}",
        "value": round(rays_per_s / 1e6, 3), "unit": "Mrays/s", "n_gpus": n, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {
            "workload": scene_text(args)[1].format(n=len(scene.triangles), w=args.width, h=args.height, spp=args.spp, b=args.bounces, t=0),
            "partition": "samples",
            "collective": "none: one process, one host thread per GPU, per-pixel sums copied GPU to GPU in sample order (rt_group_*)",
            "backend": "group",
            "arithmetic": "fast slab test (not bit-exact)" if args.fast else "exact (bit-identical to the CPU oracle and to one GPU)",
        },
        "bench_args": {"workload": args.workload, "width": args.width, "height": args.height, "spp": args.spp,
                       "bounces": args.bounces, "detail": args.detail, "chains": chains, "gpu_ordinals": ordinals},
        "samples_per_s": round(st["paths"] / dt, 1),
        "image_repeats": bool(np.array_equal(before, rgb8)),
        "rays_per_sample": round(st["segments"] / max(st["paths"], 1), 3),
        "roofline": {"bound": "valu_issue", "kernel": "k_trace<TRIANGLE>", "unit": "G wave-instr/s", "peak": VALU_PEAK_GINSTR,
                     "achieved": None, "frac": None, "traffic": None,
                     "avg_launch_ms": round(st["trace_ms"] / max(st["trace_launches"], 1), 4), "launches": int(st["trace_launches"]),
                     "note": "fractions are quoted by the one-GPU run; this line times the GPU list"},
    }
    print(json.dumps(out), flush=True)
    g.close()


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=2)  # (two: the library measures which bounce loop suits the scene over its first two large batches)
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="room",
                    help="room = the metric's configuration; blob / tree = the bunny-class and island-tree-class scenes")
    ap.add_argument("--width", type=int, default=None)
    ap.add_argument("--height", type=int, default=None)
    ap.add_argument("--spp", type=int, default=None)
    ap.add_argument("--bounces", type=int, default=None)
    ap.add_argument("--detail", type=int, default=4, help="size of the generated scene (4 = 2.9e5 / 8.2e4 / 1.0e6 triangles)")
    ap.add_argument("--tile", type=int, default=None,
                    help="tile edge of the room16k workload (default: 8192 on one or two GPUs, where the larger batch "
                         "renders 12 %% faster, 4096 beyond, so that every GPU has several tiles)")
    ap.add_argument("--trace-waves-per-cu", type=int, default=0, help="cap on resident traversal waves per CU (0 = automatic)")
    ap.add_argument("--shade-groups", type=int, default=None, help="shading workgroups per CU (library default: 8)")
    ap.add_argument("--chains", type=int, default=None, help="independent chains a batch is split into (library default: 2)")
    ap.add_argument("--opt", action="append", default=[], metavar="NAME=VALUE",
                    help="library tuning option by the name of its RT_OPT_ constant, e.g. TRACE_REFILL_MIN=8 (repeatable)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--fast", action="store_true",
                    help="RT_OPT_FAST_SLAB: reciprocal-direction slab test with fused multiply-add (not bit-exact; the line "
                         "then carries the mismatch figures against the exact mode)")
    ap.add_argument("--fast-build", type=int, default=0, metavar="MAX_LEAF",
                    help="opt-in, not the reference's tree: split every leaf of the triangle BVH above MAX_LEAF triangles on, past "
                         "the reference's stop rule (rth_refine_bvh_triangles); intersection arithmetic stays exact and the line "
                         "carries the mismatch figures against the reference's tree")
    ap.add_argument("--exact", action="store_true",
                    help="with several GPUs: hand the running sums from rank to rank in sample order instead of one "
                         "reduce, so that the image is the single-GPU image bit for bit")
    ap.add_argument("--backend", choices=("ranks", "group"), default="ranks",
                    help="with several GPUs: one process per GPU over torch.distributed/RCCL (ranks, the default), or one "
                         "process driving all GPUs through the library's rt_group_* entry points (group)")
    ap.add_argument("--dry-run-launch", action="store_true",
                    help="development aid for machines without a GPU: start the ranks, form the process group over gloo, "
                         "agree on a value and print a line without any rendering (no number is measured)")
    ap.add_argument("--rehearse-on-one-gpu", action="store_true",
                    help="development aid: run all ranks on GPU 0 and reduce over gloo through host memory, to exercise "
                         "the multi-rank control flow on a one-GPU machine (numbers are meaningless)")
    args = ap.parse_args()
    dw, dh, dspp, db = WORKLOADS[args.workload]
    args.width, args.height = args.width or dw, args.height or dh
    args.spp, args.bounces = args.spp or dspp, args.bounces if args.bounces is not None else db

    if args.backend == "group" and args.gpus > 1:
        if "WORLD_SIZE" in os.environ and int(os.environ["WORLD_SIZE"]) > 1:
            raise SystemExit("--backend group is one process for all GPUs: start it plainly, not under torch.distributed.run")
        return main_group(args)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(launch_ranks(args.gpus, sys.argv[1:]))

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.tile is None:
        args.tile = 8192 if world <= 2 else 4096
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: start bench.py plainly (it starts its own ranks) or "
                         f"under torch.distributed.run with --nproc-per-node {args.gpus}")
    if args.dry_run_launch:
        # the launch path alone: rendezvous, one collective, rank 0's line
        if world > 1:
            dist.init_process_group("gloo")
        t = torch.tensor([rank + 1], dtype=torch.int64)
        if world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
            dist.barrier()
        if rank == 0:
            print(json.dumps({"dry_run_launch": True, "n_gpus": world, "ranks_seen": int(t.item()) == world * (world + 1) // 2,
                              "value": None, "config": {"partition": "samples", "backend": "ranks",
                                                        "collective": "reduce(sum) of f32 radiance per step"}}), flush=True)
        if world > 1:
            dist.destroy_process_group()
        return
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the render loop has no CPU path")
    if args.rehearse_on_one_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        if args.rehearse_on_one_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=device)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    from raytracer_amd.renderer import Renderer

    r = Renderer(local_rank)
    # rank 0 builds the cache first so the others only load it
    if rank == 0:
        scene = load_scene(args, r)
    barrier()
    if rank != 0:
        scene = load_scene(args, r)
    r.upload(scene)
    # ---- the refined tree against the reference's tree on this very frame (untimed), then the refined one is the scene
    build_report = None
    reference_scene = scene  # the CPU baseline renders the reference's own tree
    if args.fast_build:
        t_refine = time.perf_counter()
        refined, order = scene.refined(args.fast_build)
        t_refine = time.perf_counter() - t_refine
        if args.workload != "room16k":
            check_spp = min(args.spp, 4)
            exact_hits = r.trace_primary(scene.params, 0, True)
            exact_sums = r.render_sums(scene.params, 0, check_spp)
            r.upload(refined)
            fast_hits = r.trace_primary(scene.params, 0, True)
            fast_sums = r.render_sums(scene.params, 0, check_spp)
            fast_prim = fast_hits["prim"].copy()
            tri = fast_hits["kind"] == F.RT_SHAPE_TRIANGLE
            fast_prim[tri] = order[fast_prim[tri]]  # positions in the refined order -> the reference's ids
            differ = (exact_hits["kind"] != fast_hits["kind"]) | (exact_hits["prim"] != fast_prim)
            err = ((fast_sums - exact_sums) / check_spp).astype(np.float64)
            img_e, img_f = r.resolve(scene.params, exact_sums, check_spp), r.resolve(scene.params, fast_sums, check_spp)
            leaf = (refined.triangle_nodes["b"] & np.uint32(F.RT_LEAF_BIT)) != 0
            build_report = {
                "max_leaf": args.fast_build, "refine_seconds_on_host": round(t_refine, 3),
                "nodes_reference_tree": int(scene.triangle_nodes.size), "nodes_refined_tree": int(refined.triangle_nodes.size),
                "largest_leaf_reference_tree": int((scene.triangle_nodes["b"][(scene.triangle_nodes["b"] & np.uint32(F.RT_LEAF_BIT)) != 0]
                                                    & np.uint32(0x7FFFFFFF)).max()),
                "largest_leaf_refined_tree": int((refined.triangle_nodes["b"][leaf] & np.uint32(0x7FFFFFFF)).max()),
                "pixels": int(differ.size), "spp_compared": check_spp,
                "primary_hit_id_mismatches": int(differ.sum()), "primary_hit_id_mismatch_rate": float(differ.mean()),
                "primary_hit_distance_bits_differing": int((exact_hits["t"].view(np.uint32) != fast_hits["t"].view(np.uint32)).sum()),
                "radiance_values_differing": int((exact_sums.view(np.uint32) != fast_sums.view(np.uint32)).sum()),
                "radiance_rmse": float(np.sqrt(np.mean(err ** 2))), "radiance_max_abs": float(np.abs(err).max()),
                "u8_values_differing": int((img_e != img_f).sum()),
                "u8_max_abs": int(np.abs(img_e.astype(np.int16) - img_f.astype(np.int16)).max()),
            }
        else:
            r.upload(refined)
        scene = refined
    chains = 2 if args.chains is None else args.chains  # 2 is the library's default
    r.set_option(F.RT_OPT_CHAINS, chains)
    r.set_option(F.RT_OPT_TRACE_WAVES_PER_CU, args.trace_waves_per_cu)
    if args.shade_groups is not None:
        r.set_option(F.RT_OPT_SHADE_GROUPS_PER_CU, args.shade_groups)
    for kv in args.opt:
        name, value = kv.split("=")
        r.set_option(getattr(F, "RT_OPT_" + name), int(value))
    params = scene.params
    tiled = args.workload == "room16k"
    count_params = params
    if tiled:  # traversal statistics from one tile-sized window in the middle of the frame
        count_params = params.copy()
        count_params.tile_x, count_params.tile_y = args.width // 2 - 960, args.height // 2 - 540
        count_params.tile_w, count_params.tile_h = 1920, 1080
    backend = GpuSumsBackend(r, count_params, device)
    rgb8 = None if tiled else np.zeros((args.height, args.width, 3), dtype=np.uint8)

    def reduce_to_root(t):
        if args.rehearse_on_one_gpu:
            host = t.cpu()
            dist.reduce(host, dst=0, op=dist.ReduceOp.SUM)
            t.copy_(host)
        else:
            dist.reduce(t, dst=0, op=dist.ReduceOp.SUM)
        torch.cuda.synchronize()

    def send_sums(t, dst):
        dist.send(t.cpu() if args.rehearse_on_one_gpu else t, dst)

    def recv_sums(t, src):
        if args.rehearse_on_one_gpu:
            host = torch.empty(t.shape, dtype=t.dtype)
            dist.recv(host, src)
            t.copy_(host)
        else:
            dist.recv(t, src)
        torch.cuda.synchronize()

    def agree_min(n):
        t = torch.tensor([n], dtype=torch.int64, device="cpu" if args.rehearse_on_one_gpu else device)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        return int(t.item())

    step_no = [0]
    if args.workload == "room16k":
        # tiles dealt to the ranks, every rank writing its tiles into the one mapped file:
        # no collective on the data path, one barrier per pass
        import tempfile

        from raytracer_amd.tiles import GpuTileBackend, Image, render_tiled

        path = os.path.join(os.environ.get("TMPDIR", tempfile.gettempdir()), f"rt_bench_{args.width}x{args.height}.ppm")
        image = Image(path, args.width, args.height)
        tile_backend = GpuTileBackend(r, device)

        def step():
            p = params.copy()
            p.has_continue, p.continue_sampling = 1, step_no[0]
            step_no[0] += 1
            render_tiled(tile_backend, image, p, args.tile, args.tile, world=world, rank=rank, group_size=1,
                         end_of_pass=(dist.barrier if world > 1 else None))
    else:
        def step():
            if args.exact and world > 1:
                sums = render_pass_exact(backend, 0, args.spp, world, rank, send_sums, recv_sums, agree_min=agree_min)
            else:
                sums = render_pass(backend, 0, args.spp, world, rank, reduce_to_root)
            if rank == 0:
                r.resolve(params, sums.data_ptr(), args.spp, rgb8=rgb8, sums_on_device=True)

    # ---- traversal statistics of this workload (untimed, counting kernels)
    count_spp = min(args.spp, 2)
    r.set_option(F.RT_OPT_COUNT_TRAVERSAL, 1)
    r.reset_stats()
    backend.render_sums(0, count_spp)
    cs = r.stats()
    r.set_option(F.RT_OPT_COUNT_TRAVERSAL, 0)
    seg = max(cs["segments"], 1)
    B, T = cs["branch_visits"] / seg, cs["prim_tests"][2] / seg
    S, P, H = cs["prim_tests"][0] / seg, cs["prim_tests"][1] / seg, cs["hits"] / seg
    rays_per_path = cs["segments"] / max(cs["paths"], 1)

    # ---- the fast slab test against the exact mode on this very frame (untimed)
    fast_report = None
    if args.fast and not tiled:
        check_spp = min(args.spp, 8)
        exact_hits = r.trace_primary(params, 0, True)
        exact_sums = r.render_sums(params, 0, check_spp)
        r.set_option(F.RT_OPT_FAST_SLAB, 1)
        fast_hits = r.trace_primary(params, 0, True)
        fast_sums = r.render_sums(params, 0, check_spp)
        differ = (exact_hits["kind"] != fast_hits["kind"]) | (exact_hits["prim"] != fast_hits["prim"])
        same_id = ~differ
        mean_e, mean_f = exact_sums / check_spp, fast_sums / check_spp
        err = (mean_f - mean_e).astype(np.float64)
        img_e, img_f = r.resolve(params, exact_sums, check_spp), r.resolve(params, fast_sums, check_spp)
        fast_report = {
            "primary_hit_id_mismatch_rate": float(differ.mean()), "primary_hit_id_mismatches": int(differ.sum()),
            "primary_hit_distance_bits_differ_where_id_equal": int((exact_hits["t"].view(np.uint32)[same_id]
                                                                    != fast_hits["t"].view(np.uint32)[same_id]).sum()),
            "pixels": int(differ.size), "spp_compared": check_spp,
            "radiance_values_differing": int((exact_sums.view(np.uint32) != fast_sums.view(np.uint32)).sum()),
            "radiance_rmse": float(np.sqrt(np.mean(err ** 2))), "radiance_max_abs": float(np.abs(err).max()),
            "u8_max_abs": int(np.abs(img_e.astype(np.int16) - img_f.astype(np.int16)).max()),
            "u8_values_differing": int((img_e != img_f).sum()),
        }
    elif args.fast:
        r.set_option(F.RT_OPT_FAST_SLAB, 1)

    for _ in range(args.warmup):
        step()
    r.set_option(F.RT_OPT_TIME_KERNELS, 1)
    r.reset_stats()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    st = r.stats()
    # Every step renders the same samples, so the stored bytes must come out the same whatever
    # was running beside what on the GPU: one more step after the timed region, compared byte for byte.
    repeatable = None
    if not tiled:
        before = rgb8.copy() if rank == 0 else None
        step()
        barrier()
        if rank == 0:
            repeatable = bool(np.array_equal(before, rgb8)) if world == 1 or args.exact else bool(np.abs(before.astype(np.int16) - rgb8).max() <= 1)

    agg = torch.tensor([dt, float(st["segments"]), float(st["paths"]), st["trace_ms"], float(st["trace_launches"])],
                       dtype=torch.float64, device="cpu" if args.rehearse_on_one_gpu else device)
    if world > 1:
        mx = agg.clone()
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        dist.all_reduce(agg, op=dist.ReduceOp.SUM)
        dt = float(mx[0])
    segments, paths = float(agg[1]), float(agg[2])

    if rank == 0:
        # Algorithmic bytes per ray on the reference's data sizes: each branch visit
        # reads two 36-byte child nodes, each triangle test a 36-byte a/e1/e2.
        # The dominant kernel is the triangle-BVH traversal; per ray it also reads
        # the 24-byte ray and reads+writes the 12-byte hit record.
        trace_bytes_per_ray = 72.0 * B + 36.0 * T + 24.0 + 2 * 12.0
        whole_bytes_per_ray = 72.0 * B + 36.0 * T + 24.0 * S + 32.0 * P + 120.0 + 140.0 * H + 24.0 / rays_per_path
        my_trace_ms, my_launches, my_segments = st["trace_ms"], max(st["trace_launches"], 1), st["segments"]
        avg_launch_s = my_trace_ms / 1e3 / my_launches
        rays_per_launch = my_segments / my_launches
        algorithmic_gbs = trace_bytes_per_ray * my_segments / (my_trace_ms / 1e3) / 1e9 if my_trace_ms > 0 else 0.0
        rays_per_s = segments / dt
        rec = recorded_counters(args)
        # batches below 3 * 2^22 paths run as one launch of k_path (the library's own rule); the counter
        # record describes k_trace, so no fraction is quoted for such a run
        pixels = (params.tile_w * params.tile_h) if params.tile_w else params.width * params.height
        one_launch = not tiled and pixels * min(args.spp, max(1, (1 << 29) // pixels)) < (3 << 22)
        # ... and larger ones too where the library measured that loop to be the faster one for this scene (a few timed
        # launches per step instead of two per bounce and chain)
        one_launch = one_launch or (not tiled and st["trace_launches"] <= 4 * args.steps)
        kernel_rec = None
        if rec is not None:
            kernel_rec = rec.get("path") if one_launch else rec.get("trace")
        rec = {"trace": kernel_rec} if kernel_rec else None  # (the fields below are read from the dominant kernel's record)
        # The traversal kernel's working set (branch records, triangles) lives in LDS, L2 and the
        # Infinity Cache, so HBM does not bound it; vector-instruction issue does.  Roof: every SIMD
        # issuing one wave64 vector instruction per 2 cycles at 2.4 GHz.  Instructions per ray come from
        # the hardware-counter passes recorded for these kernel sources (SQ_INSTS_VALU per launch over
        # the rays of the launch); without a matching record the figures are null.
        roofline = {
            "bound": "valu_issue",
            "kernel": "k_path (whole bounce loop, one launch per batch)" if one_launch else "k_trace<TRIANGLE>",
            "unit": "G wave-instr/s",
            "peak": VALU_PEAK_GINSTR,
            "achieved": None, "frac": None, "traffic": None,
            "avg_launch_ms": round(avg_launch_s * 1e3, 4),
            "launches": int(my_launches),
            "rays_per_launch": round(rays_per_launch),
            # a batch runs as `chains` independent chains whose traversal launches are resident side by side,
            # each on its share of the wave slots
            "launch_concurrency": chains,
            "kernel_time_share": round(my_trace_ms / 1e3 / dt, 4),
            "per_ray": {"branch_visits": round(B, 3), "triangle_tests": round(T, 3), "sphere_tests": round(S, 3),
                        "plane_tests": round(P, 3), "hit_fraction": round(H, 4)},
            # the byte roofline on the reference's data sizes, visit by visit: reported, not binding
            "algorithmic_hbm": {"binding": False, "bytes_per_ray": round(trace_bytes_per_ray, 1),
                                "bytes_per_launch": round(trace_bytes_per_ray * rays_per_launch),
                                "achieved_GBps_per_launch": round(algorithmic_gbs, 2), "peak_GBps": HBM_PEAK_GBS,
                                "whole_path_bytes_per_ray": round(whole_bytes_per_ray, 1),
                                "whole_path_GBps_over_wall": round(rays_per_s * whole_bytes_per_ray / 1e9, 2)},
        }
        if rec is not None and avg_launch_s > 0:
            ipr = rec["trace"]["valu_insts_per_ray"]
            per_launch = ipr * rays_per_launch / avg_launch_s / 1e9
            over_wall = ipr * my_segments / dt / 1e9  # this rank's traversal launches together, per second of the timed region
            roofline.update({
                # definition: one launch's instructions over its own duration (HIP events on the launch's stream)
                "achieved": round(per_launch, 2),
                "frac": round(per_launch / VALU_PEAK_GINSTR, 5),
                # with `launch_concurrency` launches resident together, the machine issues for all of them:
                "achieved_all_launches_over_wall": round(over_wall, 2),
                "frac_all_launches_over_wall": round(over_wall / VALU_PEAK_GINSTR, 5),
                "valu_insts_per_ray": round(ipr, 3),
                "valu_insts_per_launch": round(ipr * rays_per_launch),
                "salu_per_valu": rec["trace"].get("salu_per_valu"),
                "lanes_per_valu_inst": rec["trace"].get("lanes_per_valu_inst"),
                "valu_busy_alone": rec["trace"].get("valu_busy"),
                "lds_bank_conflict_share": rec["trace"].get("lds_bank_conflict_share"),
                # measured memory-side bytes per launch (FETCH_SIZE with its gfx950 correction + WRITE_SIZE), and
                # what they are of the HBM peak over the launch's duration
                "traffic": rec["trace"].get("hbm_bytes_per_launch"),
                "traffic_unit": "bytes per launch (memory-side counters of the recorded profile, scaled to this run's rays per launch)",
                "traffic_frac_of_hbm_peak": None,
                "profile": {"file": os.path.relpath(PROFILE, ROOT), "source_hash": kernel_source_hash()},
            })
            # The same against what this kernel's own instruction mix can issue: its packed, min/max, compare
            # and select instructions hold a SIMD's issue for about twice as long as a plain FP32 one
            # (profiles/r01_valu_issue_costs.txt); the counter record has the cycles the vector ALUs were busy
            # per instruction of the kernel (SQ_ACTIVE_INST_VALU x 4 / SQ_INSTS_VALU).
            cpi = rec["trace"].get("valu_cycles_per_inst")
            if cpi:
                mix_peak = 1024 * 2.4 / cpi
                roofline["issue_cycles_per_inst_of_this_mix"] = cpi
                roofline["frac_at_this_mix"] = round(per_launch / mix_peak, 5)
                roofline["frac_all_launches_over_wall_at_this_mix"] = round(over_wall / mix_peak, 5)
            if rec["trace"].get("hbm_bytes_per_ray") is not None:
                traffic = rec["trace"]["hbm_bytes_per_ray"] * rays_per_launch
                roofline["traffic"] = round(traffic)
                roofline["traffic_frac_of_hbm_peak"] = round(traffic / avg_launch_s / 1e9 / HBM_PEAK_GBS, 5)
        else:
            roofline["note"] = ("no counter record for these kernel sources and this workload in profiles/: "
                                "run tools/pmc_profile.sh and tools/pmc_record.py")
        out = {
            "metric": "Mrays/s (all bounces) breakfast-room 1080p" if args.workload == "room" else f"Mrays/s (all bounces) {args.workload}",
            "value": round(rays_per_s / 1e6, 3),
            "unit": "Mrays/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3),
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": scene_text(args)[1].format(n=len(scene.triangles), w=args.width, h=args.height, spp=args.spp,
                                                       b=args.bounces, t=args.tile),
                "partition": ("tiles" if tiled else "samples") if world > 1 else "none",
                "backend": "ranks" if world > 1 else "one process",
                "collective": ("none (each rank writes its tiles into the mapped file)" if tiled
                               else "chain of send/recv of f32 radiance per step (exact fold)" if args.exact
                               else "reduce(sum) of f32 radiance per step") if world > 1 else "none",
                "arithmetic": "fast slab test (reciprocal direction, fused multiply-add): not bit-exact" if args.fast
                              else "exact (bit-identical to the CPU oracle)",
                "tree": (f"refined: leaves of the reference's tree split on down to {args.fast_build} triangles (opt-in; not the "
                         "reference's tree: ids can differ where distances tie)") if args.fast_build
                        else "the reference's tree (bvh.rs:46-224), node for node",
            },
            "bench_args": {"workload": args.workload, "width": args.width, "height": args.height, "spp": args.spp,
                           "bounces": args.bounces, "detail": args.detail, "chains": chains, "fast_build": args.fast_build},
            "samples_per_s": round(paths / dt, 1),
            "image_repeats": repeatable,  # an extra, untimed step stored the same bytes as the last timed one
            "rays_per_sample": round(rays_per_path, 3),
            "roofline": roofline,
        }
        if fast_report is not None:
            out["fast_mode_vs_exact"] = fast_report
        if build_report is not None:
            out["refined_tree_vs_reference_tree"] = build_report
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(reference_scene, args)
        print(json.dumps(out), flush=True)

    if tiled:
        image.close()
        if rank == 0:
            os.remove(path)
    r.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
