"""The multi-GPU driver's real backend on the hardware at hand: a fresh process
(its environment set before anything touches the GPU) loads the render library
first and torch second, joins a process group of one rank over "nccl" (RCCL), and
runs render_pass + collectives + rt_resolve on torch tensors the library fills
through raw device pointers.  The stored bytes must equal the plain render's."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_world_of_one_over_rccl_equals_the_plain_render(asset_root):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE="1", RANK="0", LOCAL_RANK="0",
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "nccl_child.py"), asset_root, "6"], env=env,
                       capture_output=True, text=True, timeout=170)
    assert p.returncode == 0, p.stderr[-2000:]
    line = next(l for l in p.stdout.splitlines() if l.startswith("RESULT "))
    out = json.loads(line[len("RESULT "):])
    assert out["reduce_bytes_equal"] and out["exact_bytes_equal"]
    assert out["sums_are_independent_of_pending_collectives"]
    assert len(out["hip_runtimes_mapped"]) == 1, out["hip_runtimes_mapped"]  # one runtime although the library came first
    assert out["rccl_mapped"]
