import json
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import rtlibs  # noqa: E402


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    _install_abort_trace()


def _install_abort_trace():
    """An abort() under a library call (RCCL, the HIP runtime, ...) leaves its C stack on stderr (tests/csrc/abort_trace.c).
    pytest's own faulthandler runs first and re-raises; ours is installed after it, so it is the one the signal finds.
    Best effort: no compiler, no trace.  RCCL is asked to say what it warns about, for the same reason."""
    import ctypes, subprocess, tempfile
    os.environ.setdefault("NCCL_DEBUG", "WARN")
    src = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc", "abort_trace.c")
    lib = os.path.join(tempfile.gettempdir(), f"rt_abort_trace_{os.getuid()}.so")
    try:
        if not os.path.exists(lib) or os.path.getmtime(lib) < os.path.getmtime(src):
            subprocess.check_call(["gcc", "-O1", "-shared", "-fPIC", "-o", lib + ".tmp", src], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            os.replace(lib + ".tmp", lib)
        ctypes.CDLL(lib)
    except Exception:
        pass


@pytest.fixture(scope="session")
def golden():
    z = np.load(os.path.join(rtlibs.GOLDEN_DIR, "reference_vectors.npz"))
    return {k: z[k] for k in z.files}


@pytest.fixture(scope="session")
def golden_meta():
    with open(os.path.join(rtlibs.GOLDEN_DIR, "reference_meta.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def oracle():
    return rtlibs.Oracle()


@pytest.fixture(scope="session")
def scene_paths():
    return [os.path.join(rtlibs.DATA_DIR, f"scene_{i}.txt") for i in range(3)]
