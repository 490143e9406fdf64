"""SURVEY section 5 'sanitizer build': the host-side layout logic of the library (csrc/mlp_plan.h - slab stream,
activation / gradient tile-row layouts, transposed stream, warp plan) compiled for the CPU with ASan + UBSan and swept over
the descriptor space (tests/native/plan_sanitize.cpp).  ~40 s including the compile."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("g++") is None, reason="g++ not available")
def test_plan_logic_under_asan_ubsan(tmp_path):
    exe = str(tmp_path / "plan_sanitize")
    src = os.path.join(ROOT, "tests", "native", "plan_sanitize.cpp")
    r = subprocess.run(["g++", "-std=c++17", "-O2", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-o", exe, src],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600,
                       env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1", UBSAN_OPTIONS="print_stacktrace=1"))
    assert r.returncode == 0, (r.stdout[-1000:], r.stderr[-3000:])
    assert "plans checked:" in r.stdout and "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr
