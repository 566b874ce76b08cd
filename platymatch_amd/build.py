"""Build libplatymatch_hip.so (hand-written HIP for gfx950) in-tree with hipcc.

Usage: python -m platymatch_amd.build [--force]
The library lands next to this file so that it travels with the source tree.
-ffp-contract=off: the kernels' float64 arithmetic is specified one rounding at a time
(fused operations are written explicitly), which is what makes results reproducible
against the CPU oracle bit for bit.
"""
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "_build")
LIB = os.path.join(HERE, "libplatymatch_hip.so")
SOURCES = ["pm_api.hip", "pm_stats.hip", "pm_shape_context.hip", "pm_chi2.hip", "pm_transform.hip",
           "pm_icp.hip", "pm_icp_grid.hip", "pm_similar.hip", "pm_ransac.hip", "pm_eval.hip", "pm_lsap_dev.hip", "pm_lsap_resident.hip", "pm_lsap.cpp", "pm_lsap_core.cpp",
           "pm_host_rng.cpp"]
ARCH = "gfx950"
FLAGS = ["-O3", "-std=c++17", "-fPIC", "--offload-arch=" + ARCH, "-ffp-contract=off", "-fno-fast-math",
         "-fgpu-rdc" if False else "-fno-gpu-rdc", "-Wall", "-Wno-unused-function"]


def _hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: the HIP toolchain is required to build platymatch_amd")
    return exe


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


SANITIZE_FLAGS = ["-O1", "-g", "-fno-omit-frame-pointer", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined"]
ASAN_LIB = os.path.join(HERE, "_sanitized", "libplatymatch_hip.so")


def sanitizer_runtimes():
    """The shared sanitizer runtimes of the host compiler (to be LD_PRELOADed into an interpreter that loads ASAN_LIB)."""
    gxx = shutil.which("g++") or "g++"
    libs = [subprocess.run([gxx, "-print-file-name=" + n], capture_output=True, text=True).stdout.strip() for n in ("libasan.so", "libubsan.so")]
    return [x for x in libs if os.path.isabs(x) and os.path.exists(x)]


def build_sanitized(verbose=False):
    """SURVEY.md §5's sanitizer build of the HOST C++ (VERDICT r04 next #7a): the solver / RNG units (pm_lsap.cpp, pm_lsap_core.cpp,
    pm_host_rng.cpp — pointer-heavy code no GPU tool looks at) compiled with -fsanitize=address,undefined and linked, with the
    product's own device objects, into a SECOND library (platymatch_amd/_sanitized/; the product library is untouched).  Use:
    PM_LIB_PATH=<that file> LD_PRELOAD=<sanitizer_runtimes()> python -m pytest tests/test_lsap_core.py ...
    (tests/test_host_sanitizers.py does exactly that).  GPU AddressSanitizer is not available on this pool: host code only.
    Also reached by  PM_HOST_SANITIZE=1 python -m platymatch_amd.build."""
    build_native()
    out_dir = os.path.dirname(ASAN_LIB)
    os.makedirs(out_dir, exist_ok=True)
    gxx = shutil.which("g++") or "g++"
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")] + [os.path.join(HERE, "..", "include", "platymatch_hip.h")]
    objs = []
    for src in SOURCES:
        plain = os.path.join(OBJ, src.replace(".hip", ".o").replace(".cpp", ".o"))
        if not src.endswith(".cpp"):
            objs.append(plain)
            continue
        s, o = os.path.join(CSRC, src), os.path.join(out_dir, src.replace(".cpp", ".o"))
        objs.append(o)
        if _stale(o, [s] + headers):
            cmd = [gxx] + SANITIZE_FLAGS + ["-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-Wall", "-c", s, "-o", o]
            if verbose:
                print(" ".join(cmd), flush=True)
            r = subprocess.run(cmd, capture_output=True, text=True)
            if r.returncode != 0:
                raise RuntimeError("g++ (sanitized) failed:\n%s\n%s" % (" ".join(cmd), r.stderr))
    if _stale(ASAN_LIB, objs):
        r = subprocess.run([_hipcc(), "-shared", "-fPIC", "--offload-arch=" + ARCH, "-o", ASAN_LIB] + objs, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link (sanitized) failed:\n%s" % r.stderr)
    return ASAN_LIB


def build_native(force=False, verbose=False):
    os.makedirs(OBJ, exist_ok=True)
    hipcc = _hipcc()
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    headers.append(os.path.join(HERE, "..", "include", "platymatch_hip.h"))
    jobs = []
    objs = []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(OBJ, src.replace(".hip", ".o").replace(".cpp", ".o"))
        objs.append(o)
        if force or _stale(o, [s] + headers):
            if src.endswith(".cpp"):   # host-only translation units: plain C++ compiler, same floating-point discipline
                jobs.append([shutil.which("g++") or "g++", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math",
                             "-Wall", "-c", s, "-o", o])
            else:
                jobs.append([hipcc] + FLAGS + ["-c", s, "-o", o])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed:\n%s\n%s" % (" ".join(cmd), r.stderr))
        return r.stderr

    with ThreadPoolExecutor(max_workers=min(4, max(1, len(jobs)))) as ex:
        for warn in ex.map(run, jobs):
            if verbose and warn:
                print(warn)
    if force or jobs or _stale(LIB, objs):
        run([hipcc, "-shared", "-fPIC", "--offload-arch=" + ARCH, "-o", LIB] + objs)
    return LIB


if __name__ == "__main__":
    print(build_native(force="--force" in sys.argv, verbose=True))
    if os.environ.get("PM_HOST_SANITIZE") == "1":
        print(build_sanitized(verbose=True))
        print("LD_PRELOAD=" + ":".join(sanitizer_runtimes()))
