"""The sharded runners with the HIP engine on every rank (VERDICT r2 #5b): two gloo ranks on the one GPU of the
test box, started by tests/conftest.py at session start (tests/dist_gpu_job.py has the job)."""
import json
import os

import pytest

pytestmark = pytest.mark.gpu


def test_two_ranks_hip_engine_equal_single_process(dist_gpu_job):
    proc, out_path = dist_gpu_job
    rc = proc.wait(timeout=900)
    assert os.path.exists(out_path), f"the 2-rank job left no result (exit code {rc})"
    with open(out_path) as fh:
        res = json.load(fh)
    assert res.get("ok"), res
    assert rc == 0
    assert res["world"] == 2 and res["backend"] == "gloo"
    assert set(res["checks"]) == {f"{t}_{r}" for t in ("small_to_large", "m_best", "qo_find_periods") for r in ("sharded", "pipelined")}
