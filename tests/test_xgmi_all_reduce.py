"""The library's peer-buffer all-reduce between two processes that share cuda:0 (one GPU per lease): IPC handle
exchange, flag protocol, one- and two-shot sums, in place and out of place, HIP-graph replay; known answer as in the
reference's comm-op test (tests/distributed/test_comm_ops.py:30-38).  The two ranks are fresh child processes which
conftest.py starts at session start, before this process touches the GPU (tests/xgmi_worker.py); this test collects
their verdicts.  Bandwidth over xGMI is not measured here: both ranks sit on one device."""
import json
import os
import time

import pytest

import conftest

pytestmark = pytest.mark.gpu


@pytest.mark.timeout(400)
def test_two_ranks_on_one_gpu():
    job = conftest.XGMI_JOB
    if job is None:
        pytest.skip("the worker processes are started only for -m gpu sessions on a machine with a GPU")
    deadline = time.time() + 360
    for p in job["procs"]:
        p.wait(timeout=max(1.0, deadline - time.time()))
    for r in range(job["world"]):
        path = os.path.join(job["dir"], f"rank{r}.json")
        assert os.path.exists(path), (f"rank {r} left no result (exit code {job['procs'][r].returncode}); its output:\n"
                                      + conftest.xgmi_worker_log(r))
        with open(path) as f:
            res = json.load(f)
        assert res["ok"], res.get("error")
        assert res["checks"] == ["known answer", "sizes and dtypes", "graph replay",
                                 "row-parallel layer with fused sum + add + norm", "fused add + norm",
                                 "timeout poisons the output"]
