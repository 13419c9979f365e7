"""bench.py starts its own ranks: `python bench.py --gpus N` with no launcher around it
runs N processes (torch.distributed.run on 127.0.0.1) and passes rank 0's line through,
as the reference starts its own workers inside the one call
(/root/reference/src/lib.rs:195-219).  Without a GPU only the launch path can run
(--dry-run-launch: rendezvous and one collective over gloo, no rendering); on the GPU box
the real step runs with both ranks on GPU 0 (--rehearse-on-one-gpu), and through the
library's own GPU list (--backend group)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(*flags, timeout=600):
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):  # a plain start, whatever runs the tests
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *flags], capture_output=True, text=True, env=env,
                       timeout=timeout, cwd=ROOT)
    lines = [l for l in p.stdout.splitlines() if l.startswith("{") and l.endswith("}")]
    return p, [json.loads(l) for l in lines]


def test_plain_start_with_two_gpus_starts_two_ranks():
    p, lines = run_bench("--gpus", "2", "--dry-run-launch")
    assert p.returncode == 0, p.stderr[-2000:]
    assert len(lines) == 1  # rank 0 alone prints
    assert lines[0]["n_gpus"] == 2 and lines[0]["ranks_seen"] is True
    assert lines[0]["config"]["partition"] == "samples" and "collective" in lines[0]["config"]


def test_a_failing_rank_fails_the_plain_start():
    # no GPU here (or, on a GPU box, an impossible option): the children's failure is the parent's status
    p, lines = run_bench("--gpus", "2", "--steps", "1", "--warmup", "0", "--opt", "TRACE_WAVES=3")
    assert p.returncode != 0 and not lines


SMALL = ("--width", "320", "--height", "180", "--spp", "4", "--bounces", "8", "--detail", "1", "--steps", "1", "--warmup", "1",
         "--no-cpu-baseline")


@pytest.mark.gpu
def test_two_ranks_on_one_gpu_render_the_step():
    p, lines = run_bench("--gpus", "2", "--rehearse-on-one-gpu", *SMALL)
    assert p.returncode == 0, p.stderr[-3000:]
    line = lines[-1]
    assert line["n_gpus"] == 2 and line["config"]["partition"] == "samples" and line["config"]["backend"] == "ranks"
    assert "reduce" in line["config"]["collective"] and line["value"] > 0 and line["image_repeats"] is True
    # the two ranks together traced the frame's paths
    assert line["samples_per_s"] > 0 and abs(line["rays_per_sample"] - line["value"] * 1e6 / line["samples_per_s"]) < 1e-2


@pytest.mark.gpu
def test_group_backend_renders_the_same_step_in_one_process():
    p1, one = run_bench("--gpus", "1", *SMALL)
    p2, two = run_bench("--gpus", "2", "--backend", "group", "--rehearse-on-one-gpu", *SMALL)
    assert p1.returncode == 0 and p2.returncode == 0, (p1.stderr[-2000:], p2.stderr[-2000:])
    a, b = one[-1], two[-1]
    assert b["n_gpus"] == 2 and b["config"]["backend"] == "group" and b["config"]["partition"] == "samples"
    assert b["image_repeats"] is True and b["value"] > 0
    # the same frame: rays per sample agree (the one-GPU line counts them over 2 of the 4 samples, the group over all)
    assert abs(a["rays_per_sample"] - b["rays_per_sample"]) < 0.05
