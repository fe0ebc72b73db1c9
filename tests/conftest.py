import importlib
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PKG = "spatial-temporal-lidar-camera-calibration_amd"

# torch first: its wheel bundles its own ROCm libraries (libamdhip64, librccl, librocm_smi64). Loaded first, they are the one HIP
# runtime of the process and the library's own dependency resolves to them (same SONAME); loaded second, behind /opt/rocm's copies,
# the process runs torch on a runtime it was not built for — a torch stream handed to the C-ABI then aborts inside libamdhip64
# (seen in round 3). bench.py imports torch first for the same reason; INTEGRATION.md says so for integrators.
try:
    import torch  # noqa: F401
except Exception:   # the CPU tier runs without it too
    torch = None


# the library reads its debug environment overrides (IBA_FACTOR_MFMA, IBA_NN_ROUNDS, IBA_FACTOR_V2, ...) only when IBA_DEBUG_ENV=1 is set as well
# (round 6): the tests are the users of those overrides
os.environ.setdefault("IBA_DEBUG_ENV", "1")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def pkg():
    return importlib.import_module(PKG)


@pytest.fixture(scope="session")
def synth():
    return importlib.import_module(PKG + ".synth")


@pytest.fixture(scope="session")
def abi():
    return importlib.import_module(PKG + ".abi")


@pytest.fixture(scope="session")
def ob():
    """oracle binding (test infrastructure)"""
    from oracle import binding
    binding.lib()
    return binding


_scene_cache = {}


@pytest.fixture(scope="session")
def scene_small(synth):
    """12 keyframes x 4000 points, 2000 keypoints each (seed 1)."""
    if "small" not in _scene_cache:
        _scene_cache["small"] = synth.make_scene(n_frames=12, pts_per_frame=4000, seed=1)
    return _scene_cache["small"]
