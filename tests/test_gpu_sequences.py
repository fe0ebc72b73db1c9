"""Sequences of calls on one handle: everything that lives across calls (pair-list reuse, anchored neighbour lists, side stream,
candidate ring, frozen problem, parameter changes) on, against a handle with all of it off — same bits (tools/sequence_fuzz.py runs
the same comparison over hundreds of random sequences; 1150 identical in round 3)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_random_call_sequences_do_not_depend_on_the_cross_call_state():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "sequence_fuzz.py"), "8", "9000"], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    assert "8/8 sequences identical" in out.stdout, out.stdout[-3000:]
