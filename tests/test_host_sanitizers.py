"""SURVEY.md §5's "-fsanitize=address host build", committed (VERDICT r04 next #7a): the host C++ of the assignment solver and of
the NumPy-exact RNG (pm_lsap_core.cpp, pm_lsap.cpp, pm_host_rng.cpp: ~1 500 lines of pointer arithmetic, heaps and AVX-512
intrinsics that no GPU tool looks at) is rebuilt with -fsanitize=address,undefined (platymatch_amd.build.build_sanitized) and its
own CPU tests are re-run against that library, the sanitizer runtimes preloaded into the interpreter.  A heap overflow, a
use-after-free or undefined behaviour in those units ends the child non-zero with the sanitizer's report.  (GPU AddressSanitizer is
not available on this pool: host code only.)"""
import os
import subprocess
import sys

import pytest

from conftest import ROOT


def test_host_solver_and_rng_tests_are_clean_under_address_and_ub_sanitizers():
    from platymatch_amd import build as B
    runtimes = B.sanitizer_runtimes()
    if len(runtimes) < 2:
        pytest.skip("the host compiler ships no shared libasan / libubsan")
    lib = B.build_sanitized()
    env = dict(os.environ, PM_LIB_PATH=lib, LD_PRELOAD=":".join(runtimes),
               # CPython and NumPy do not free everything at exit: leak reports are not what this test is about
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=0:exitcode=66", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1:exitcode=67",
               OMP_NUM_THREADS="2")
    modules = ["tests/test_lsap_core.py", "tests/test_lsap.py", "tests/test_host_logic.py"]     # (the RNG replica: test_host_logic's draws test)
    p = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-s", "-m", "not gpu", "-p", "no:cacheprovider"] + modules,
                       env=env, cwd=ROOT, capture_output=True, text=True, timeout=1500)
    text = p.stdout + p.stderr
    assert "AddressSanitizer" not in text and "runtime error:" not in text, text[-4000:]
    assert p.returncode == 0, text[-4000:]
    assert " passed" in p.stdout and "failed" not in p.stdout, p.stdout[-1500:]
    # the child really ran against the instrumented library
    probe = subprocess.run([sys.executable, "-c", "from platymatch_amd import _native as n; n.load(); print(n.LIB_PATH)"],
                           env=env, cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert probe.returncode == 0 and probe.stdout.strip().endswith(os.path.join("_sanitized", "libplatymatch_hip.so")), probe.stderr[-1500:]
