import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


# The two ranks of tests/test_xgmi_all_reduce.py: fresh processes started here, before this process has touched the
# GPU (a process that has initialised the GPU must not start programs that use it by fork + exec).
XGMI_JOB = None


def pytest_sessionstart(session):
    global XGMI_JOB
    expr = session.config.getoption("markexpr", "") or ""
    if "gpu" not in expr or "not gpu" in expr:
        return
    import socket
    import subprocess
    import tempfile
    try:
        import torch
        if torch.cuda.device_count() < 1:        # counting devices does not initialise the GPU
            return
    except Exception:
        return
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = tempfile.mkdtemp(prefix="pq_xgmi_")
    world = 2
    # the workers' stdout and stderr go to files the test prints when a rank leaves no verdict (a crash)
    procs = []
    for r in range(world):
        log = open(os.path.join(out, f"rank{r}.log"), "wb")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "xgmi_worker.py"), str(r), str(world),
                                       str(port), out], stdout=log, stderr=subprocess.STDOUT))
        log.close()
    XGMI_JOB = {"procs": procs, "dir": out, "world": world}


def xgmi_worker_log(rank, tail=4000):
    """The end of what worker `rank` wrote to stdout / stderr."""
    try:
        with open(os.path.join(XGMI_JOB["dir"], f"rank{rank}.log"), "rb") as f:
            return f.read()[-tail:].decode("utf-8", "replace")
    except (OSError, TypeError):
        return "(no log)"


def pytest_sessionfinish(session, exitstatus):
    if XGMI_JOB is not None:
        for p in XGMI_JOB["procs"]:
            if p.poll() is None:
                p.kill()


class Golden:
    """Golden vectors produced by the reference's own Python (tests/golden/make_golden.py)."""

    def __init__(self):
        d = os.path.join(ROOT, "tests", "golden")
        with open(os.path.join(d, "pq_golden.json")) as f:
            self.manifest = json.load(f)
        self.arrays = np.load(os.path.join(d, "pq_golden.npz"), allow_pickle=False)

    def cases(self, kind=None):
        return [c for c in self.manifest["cases"] if kind is None or c["kind"] == kind]

    def get(self, case, name):
        key = f"{case['id']}_{name}"
        return self.arrays[key] if key in self.arrays.files else None


_GOLDEN = None


def golden():
    global _GOLDEN
    if _GOLDEN is None:
        _GOLDEN = Golden()
    return _GOLDEN


@pytest.fixture(scope="session")
def gold():
    return golden()


@pytest.fixture(autouse=True)
def _dispatch_knobs_restored():
    """The library reads its PQ_AMD_* diagnostic knobs once; a test that sets them (monkeypatch.setenv followed by
    _lib.reload_knobs()) leaves the library on the restored environment again."""
    yield
    try:
        from vllm_pq_amd import _lib
        if _lib._lib is not None:
            _lib.reload_knobs()
    except Exception:
        pass
