"""Randomised bit-exact parity sweep (tests/fuzz_parity.py): random scenes incl. grid-aligned boxes and cameras
on slab planes, frame sizes, spp, bounce limits, chunkings, compiled / generic kernels, 2-8 strip splits."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_random_cases_match_the_oracle():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "fuzz_parity.py"), "60", "20261003"],
                         capture_output=True, text=True, timeout=900)
    tail = "\n".join(out.stdout.splitlines()[-5:])
    assert out.returncode == 0, tail + out.stderr[-500:]
    assert "60 cases, 0 mismatching" in out.stdout, tail
