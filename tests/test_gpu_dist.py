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


def test_two_contexts_from_two_threads():
    """A caller without torch.distributed shards a host batch itself: one ph_ctx per device, each driven from its own
    thread.  Here two contexts on the one GPU of the test box work on the two halves of a
    batch concurrently (ctypes releases the GIL during the calls); the halves must equal the one-context results."""
    import threading

    import numpy as np

    import __graft_entry__ as ge

    ge.build()
    from pyperiod_amd import PeriodEngine
    from pyperiod_amd.dist import shard_bounds
    from pyperiod_amd.synth import multi_sinusoid_batch

    x = multi_sinusoid_batch(4000, 9, 2048)
    ref_eng = PeriodEngine(0)
    want_m = ref_eng.m_best(x, 5)
    want_s = ref_eng.small_to_large(x, 0.05, cap=48)
    engines = [PeriodEngine(0), PeriodEngine(0)]
    got = [None, None]
    errs = []

    def work(r):
        try:
            lo, hi = shard_bounds(x.shape[0], 2, r)
            for _ in range(3):  # several calls per thread: the two streams really interleave
                got[r] = (engines[r].m_best(x[lo:hi], 5), engines[r].small_to_large(x[lo:hi], 0.05, cap=48))
        except Exception as e:  # noqa: BLE001
            errs.append(e)

    threads = [threading.Thread(target=work, args=(r,)) for r in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(300)
    assert not errs, errs
    lo1 = shard_bounds(x.shape[0], 2, 1)[0]
    for k in range(4):  # periods, powers, bases, status of m_best
        assert np.array_equal(np.concatenate([got[0][0][k], got[1][0][k]]), want_m[k])
    for k in (0, 1, 2, 4):  # counts, periods, powers, status of small_to_large (bases rows beyond count are unspecified)
        assert np.array_equal(np.concatenate([got[0][1][k], got[1][1][k]]), want_s[k])
    assert lo1 == 5
    for e in engines + [ref_eng]:
        e.close()
