"""Renders repeat bit for bit: the headline frame a few times through each bounce loop (two chains side by side,
one chain, the one launch), a render of many one-sample passes traced together, and the same over a list of
GPUs — a data race between launches that run side by side shows as a difference once in a while
(tools/soak.py runs the same check for as many rounds as asked)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_a_short_soak_finds_every_render_identical():
    done = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "soak.py"), "3"], cwd=ROOT, capture_output=True, text=True,
                          timeout=280)
    assert done.returncode == 0, done.stdout[-2000:] + done.stderr[-2000:]
    assert "soak ok" in done.stdout and "short passes ok" in done.stdout and "group ok" in done.stdout
