"""The float screen of the window-pair kernels against its rigorous radius, value by value.

The parity tests only see the screen through the decisions it feeds; this test looks at the numbers: the pass test bed
(tools/micro/pair_pass_bench.hip, the passes of pyperiod_amd/csrc/ph_pair.h compiled as they are) folds one window
pair for every base period >= 64 with the single-, two- and four-class passes and compares each value with the fp64
fold of the same float samples.  |screen - exact| must stay below pair_radius(rows, q) x sum of squares -- the bound
k_mbest_step1_pair / k_small_to_large_pair / k_best_correlation_pair prune with (Periods.py:501-515, :246-287)."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_screen_values_stay_inside_the_rigorous_radius(tmp_path):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc on this box")
    exe = str(tmp_path / "pair_pass_bench")
    subprocess.run(
        [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-Wno-unused-value", "-I", os.path.join(ROOT, "pyperiod_amd", "csrc"),
         os.path.join(ROOT, "tools", "micro", "pair_pass_bench.hip"), "-o", exe],
        check=True, timeout=600, cwd=str(tmp_path))
    # classes, first base period, end of the base periods (classes x (end - 1) <= N = 4096)
    for classes, lo, hi in ((1, 64, 2048), (2, 64, 2048), (4, 64, 1024)):
        out = subprocess.run([exe, str(classes), str(lo), str(hi)], check=True, timeout=300, capture_output=True, text=True).stdout
        m = re.search(r"(\d+) screen values against the fp64 fold: largest \|error\| / \(pair_radius x sum of squares\) = ([0-9.eE+-]+)", out)
        assert m, out
        assert int(m.group(1)) == 2 * (hi - lo) * {1: 1, 2: 2, 4: 3}[classes]
        assert float(m.group(2)) < 1.0, out
