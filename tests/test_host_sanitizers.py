"""AddressSanitizer + UBSan over the HOST half of libperiod_hip.so: period_hip.hip is compiled
host-only with -fsanitize=address,undefined, linked against tests/hipstub/hip_stub.cpp (host memory in place of the
HIP runtime; kernels do not run) and driven through every entry point of the C ABI by tests/hipstub/host_driver.cpp.
Round 1's advisor found an out-of-bounds read in a table check this way by hand; this keeps looking."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


@pytest.mark.skipif(not os.path.exists(HIPCC) or not os.path.exists("/opt/rocm/lib/llvm/bin/clang++"), reason="ROCm clang not available")
def test_host_half_under_asan_ubsan(tmp_path):
    src = os.path.join(ROOT, "pyperiod_amd", "csrc", "period_hip.hip")
    stub = os.path.join(ROOT, "tests", "hipstub")
    san = ["-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer", "-g", "-O1", "-std=c++17"]
    obj = str(tmp_path / "period_host.o")
    subprocess.run([HIPCC, "--cuda-host-only", *san, "-Wno-unused-function", "-c", src, "-o", obj], check=True, cwd=ROOT,
                   timeout=900)
    # the fat binary symbol the host object refers to (device code is not built here)
    nm = subprocess.run(["nm", "-u", obj], check=True, capture_output=True, text=True).stdout
    fat = [ln.split()[-1] for ln in nm.splitlines() if "__hip_fatbin" in ln]
    exe = str(tmp_path / "host_driver")
    # plain clang++ link: no HIP runtime library on the line, the stub provides the symbols
    clang = "/opt/rocm/lib/llvm/bin/clang++"
    cmd = [clang, *san, "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", os.path.join(stub, "hip_stub.cpp"),
           os.path.join(stub, "host_driver.cpp"), obj, "-o", exe] + [f"-Wl,--defsym,{s}=0" for s in fat]
    subprocess.run(cmd, check=True, cwd=ROOT, timeout=900)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    r = subprocess.run([exe], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-6000:]
    assert "host sanitizer driver ok" in r.stdout
    assert "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr, r.stderr[-6000:]
