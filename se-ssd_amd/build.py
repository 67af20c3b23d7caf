"""Build the gfx950 C-ABI library (se-ssd_amd/lib/libsessd_hip.so) with hipcc.

Cross-compiles without a GPU. One object per csrc/*.hip (rebuilt only when the
source or a header is newer), then one shared link. No cmake, no torch headers:
the library's boundary is plain C (include/sessd_hip.h).
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
OBJDIR = os.path.join(HERE, "build")
LIB = os.path.join(LIBDIR, "libsessd_hip.so")
ARCH = "gfx950"
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function",
         "-I", os.path.join(os.path.dirname(HERE), "include"), "-I", CSRC]
# Geometry / index kernels must round like the reference's scalar CPU code: no FMA contraction.
# The MFMA / FMA-chain kernels (sparse and dense convolutions) contract freely.
CONTRACT_FAST = {"dense_conv.hip", "dense_wino_sk.hip", "sparse_conv.hip"}


def _newer(src_list, target):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in src_list)


def build(verbose=True, force=False):
    os.makedirs(LIBDIR, exist_ok=True)
    os.makedirs(OBJDIR, exist_ok=True)
    srcs = sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hpp")]
    hdrs += [os.path.join(os.path.dirname(HERE), "include", h) for h in ("sessd_hip.h", "sessd_hip_types.h")]
    hdrs = [h for h in hdrs if os.path.exists(h)]
    jobs = []
    objs = []
    for s in srcs:
        src = os.path.join(CSRC, s)
        obj = os.path.join(OBJDIR, s[:-4] + ".o")
        objs.append(obj)
        if force or _newer([src] + hdrs, obj):
            extra = [] if s in CONTRACT_FAST else ["-ffp-contract=off"]
            jobs.append([HIPCC] + FLAGS + extra + ["-c", src, "-o", obj])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed:\n" + r.stdout + r.stderr)
        if verbose and r.stderr.strip():
            print(r.stderr, file=sys.stderr)

    with ThreadPoolExecutor(max_workers=min(8, max(1, len(jobs)))) as ex:
        list(ex.map(run, jobs))
    if jobs or force or _newer(objs, LIB):
        run([HIPCC, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", LIB] + objs)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
