"""Child process of tests/test_gpu_nccl.py: one rank of a world of one over the
"nccl" backend (RCCL), with the render library loaded BEFORE torch."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from raytracer_amd import _ffi as F  # noqa: E402

F.device_lib()  # the library (and the HIP runtime it pulls in) first, torch afterwards

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from raytracer_amd import assets, parse  # noqa: E402
from raytracer_amd.distributed import GpuSumsBackend, render_pass, render_pass_exact  # noqa: E402
from raytracer_amd.renderer import Renderer  # noqa: E402


def main():
    asset_root, spp = sys.argv[1], int(sys.argv[2])
    device = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", device_id=device)
    scene = parse(assets.room_scene_text("room", 240, 136, spp, 12), asset_root)
    out = {}
    with Renderer(0) as r:
        r.upload(scene)
        want = r.render(scene.params)  # the non-distributed render
        backend = GpuSumsBackend(r, scene.params, device)
        reduced = []

        def reduce_to_root(t):  # bare collective: returns once enqueued
            dist.reduce(t, dst=0, op=dist.ReduceOp.SUM)
            reduced.append(True)

        # world of one: render_pass does not reduce; RCCL is exercised on the same tensor explicitly
        sums = render_pass(backend, 0, spp, 1, 0, None)
        reduce_to_root(sums)
        dist.all_reduce(sums, op=dist.ReduceOp.SUM)
        got = r.resolve(scene.params, sums.data_ptr(), spp, sums_on_device=True)
        out["reduce_bytes_equal"] = bool(np.array_equal(got, want))
        # the exact hand-over variant (no transfer at world 1), then RCCL traffic, then the library again
        sums2 = render_pass_exact(backend, 0, spp, 1, 0, None, None)
        dist.broadcast(sums2, src=0)
        got2 = r.resolve(scene.params, sums2.data_ptr(), spp, sums_on_device=True)
        out["exact_bytes_equal"] = bool(np.array_equal(got2, want))
        # library writes, RCCL reads at once, library overwrites at once: the backend's ordering contract
        a = backend.render_sums(0, 2).clone()
        dist.all_reduce(backend.buffer, op=dist.ReduceOp.SUM)  # enqueued, not awaited
        b = backend.render_sums(2, 2).clone()
        torch.cuda.synchronize()
        out["sums_are_independent_of_pending_collectives"] = bool(
            torch.equal(a.cpu(), torch.from_numpy(r.render_sums(scene.params, 0, 2).reshape(-1))) and
            torch.equal(b.cpu(), torch.from_numpy(r.render_sums(scene.params, 2, 2).reshape(-1))))
    with open(f"/proc/{os.getpid()}/maps") as f:
        maps = f.read()
    out["hip_runtimes_mapped"] = sorted({l.split()[-1] for l in maps.splitlines() if "libamdhip64" in l})
    out["rccl_mapped"] = any("librccl" in l for l in maps.splitlines())
    dist.destroy_process_group()
    print("RESULT " + json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
