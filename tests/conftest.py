import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


_DIST_JOB = {}


def _gpu_visible():
    """Is there a GPU, decided WITHOUT the HIP runtime (torch.cuda.device_count() may fall through to
    hipGetDeviceCount, which initialises it): the kernel driver's device node and the visibility variables."""
    if not os.path.exists("/dev/kfd"):
        return False
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        if os.environ.get(var, None) in ("", "-1"):
            return False
    return True


def pytest_collection_finish(session):
    """The 2-rank sharding job of tests/test_gpu_dist.py runs as a child process tree that must be started BEFORE this
    process touches the GPU (a process that has initialised HIP must not fork+exec on the GPU boxes).  Collection
    imports the test modules but runs nothing, so this hook -- after deselection (-m, -k), before the first test --
    starts it, and only when the test that reads its result is among the selected items."""
    if "proc" in _DIST_JOB or not _gpu_visible():
        return
    if not any(it.nodeid.endswith("test_gpu_dist.py::test_two_ranks_hip_engine_equal_single_process") for it in session.items):
        return
    import subprocess
    import tempfile

    out = os.path.join(tempfile.mkdtemp(prefix="ph_dist_"), "result.json")
    proc = subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "dist_gpu_job.py"), "--launch", out])
    _DIST_JOB.update(proc=proc, out=out)


def pytest_sessionfinish(session, exitstatus):
    """The launcher has its own 600 s deadline and kills its ranks; a session that ends early (failure with -x,
    interrupt) must not leave them holding the GPU: terminate the launcher (it is our own child, exact PID) and,
    if it does not go, kill it."""
    proc = _DIST_JOB.get("proc")
    if proc is None or proc.poll() is not None:
        return
    import subprocess

    try:
        proc.wait(timeout=5 if exitstatus else 660)
    except subprocess.TimeoutExpired:
        proc.terminate()
        try:
            proc.wait(timeout=10)
        except subprocess.TimeoutExpired:
            proc.kill()
            proc.wait()


@pytest.fixture(scope="session")
def dist_gpu_job():
    if "proc" not in _DIST_JOB:
        pytest.skip("no GPU visible when the tests were collected")
    return _DIST_JOB["proc"], _DIST_JOB["out"]


@pytest.fixture(scope="session")
def golden():
    cache = {}

    def load(name):
        if name not in cache:
            cache[name] = dict(np.load(os.path.join(GOLDEN, name + ".npz")))
        return cache[name]

    return load


def rel_err(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    scale = max(np.max(np.abs(b)) if b.size else 0.0, 1e-300)
    return float(np.max(np.abs(a - b)) / scale) if b.size else 0.0


def elem_err(a, b, floor_frac=1e-5):
    """Element-wise relative error max_i |a_i - b_i| / max(|b_i|, floor) with the absolute floor
    floor_frac * max|b|: weak entries (small-p sweep values, weak periods' powers) are held to
    the same relative bar as the strongest one, down to entries 1e-5 of it; below the floor the
    bar degrades to the max-norm one (BLAS nrm2 / the float32 reference carry absolute noise)."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    if not b.size:
        return 0.0
    floor = max(floor_frac * float(np.max(np.abs(b))), 1e-300)
    return float(np.max(np.abs(a - b) / np.maximum(np.abs(b), floor)))
