"""bench.py's command line: the N > 1 line must never describe a one-GPU job."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_gpus_mismatch_with_world_size_exits_nonzero():
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode != 0
    assert "WORLD_SIZE=1" in (r.stderr + r.stdout)
    assert '"n_gpus"' not in r.stdout  # no bench line was printed


def test_launcher_relays_failure_of_its_ranks():
    """Without WORLD_SIZE, --gpus 2 starts two ranks itself; here they cannot run (no GPU in the CPU container, or
    an unknown backend), and the parent must come back non-zero instead of printing a 1-GPU line."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                        "--backend", "no-such-backend", "--no-cpu-baseline", "--no-c4"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode != 0
    assert '"n_gpus": 1' not in r.stdout


def test_csrc_hash_guards_recorded_traffic():
    sys.path.insert(0, ROOT)
    import bench

    h = bench.csrc_hash()
    assert len(h) == 16 and h == bench.csrc_hash()
    val, why, ms = bench.load_recorded_traffic("no_such_kernel")
    assert val is None and ms is None
