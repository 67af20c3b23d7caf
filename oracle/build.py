"""TEST INFRASTRUCTURE ONLY. Builds the CPU oracle (plain C, gcc) and, when
/root/reference is present, the compiled REFERENCE iou3d CPU path (oracle/_ref).

  oracle/liboracle.so          <- oracle/*.c           (restatements; checker only)
  oracle/_ref/libref_iou3d.so  <- /root/reference/det3d/core/iou3d/src/iou3d_cpu.cpp
                                   + oracle/ref_iou3d_stub.cpp  (reference sources are
                                   compiled where they lie; nothing is copied)
  oracle/_ref/ref_nms_cpu*.so  <- /root/reference/det3d/ops/nms/nms_cpu.h (greedy rotated NMS, DI-NMS), through
                                   oracle/ref_nms_stub.cpp with <boost/geometry.hpp> resolved to oracle/boost_shim
                                   (boost is not installed: the shim supplies the two polygon-area calls, see its header)
All are git-ignored and travel to the GPU box with the gpurun snapshot.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "liboracle.so")
REF_DIR = os.path.join(HERE, "_ref")
REF_LIB = os.path.join(REF_DIR, "libref_iou3d.so")
REF_SRC = "/root/reference/det3d/core/iou3d/src/iou3d_cpu.cpp"
C_SRCS = ["voxelize.c", "iou3d.c", "rotate_nms.c", "rotate_iou_eval.c", "di_nms.c"]


def _newer(srcs, target):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in srcs)


def build_oracle(verbose=False):
    srcs = [os.path.join(HERE, s) for s in C_SRCS if os.path.exists(os.path.join(HERE, s))]
    if _newer(srcs, LIB):
        # -ffp-contract=off: no FMA fusion, float32 steps stay float32 steps
        cmd = ["gcc", "-O2", "-std=c11", "-ffp-contract=off", "-fno-fast-math", "-shared", "-fPIC", "-o", LIB] + srcs + ["-lm"]
        if verbose:
            print(" ".join(cmd))
        subprocess.run(cmd, check=True)
    return LIB


def build_ref(verbose=False):
    """Compile the reference's own iou3d_cpu.cpp. Only possible where /root/reference exists."""
    if not os.path.exists(REF_SRC):
        return REF_LIB if os.path.exists(REF_LIB) else None
    stub = os.path.join(HERE, "ref_iou3d_stub.cpp")
    if not _newer([stub, REF_SRC], REF_LIB):
        return REF_LIB
    os.makedirs(REF_DIR, exist_ok=True)
    import sysconfig
    import torch
    from torch.utils import cpp_extension as ce
    inc = []
    for p in ce.include_paths():
        inc += ["-I", p]
    inc += ["-I", sysconfig.get_paths()["include"]]
    tlib = os.path.join(os.path.dirname(torch.__file__), "lib")
    cmd = ["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-w", "-D_GLIBCXX_USE_CXX11_ABI=%d" % int(torch._C._GLIBCXX_USE_CXX11_ABI)]
    cmd += inc + [stub, REF_SRC, "-o", REF_LIB, "-L", tlib, "-Wl,-rpath," + tlib, "-ltorch", "-ltorch_cpu", "-lc10"]
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True)
    return REF_LIB


REF_NMS_H = "/root/reference/det3d/ops/nms/nms_cpu.h"


def ref_nms_path():
    import sysconfig
    return os.path.join(REF_DIR, "ref_nms_cpu" + sysconfig.get_config_var("EXT_SUFFIX"))


def build_ref_nms(verbose=False):
    """Compile the reference's own nms_cpu.h (where it lies) into a Python module; boost::geometry comes from oracle/boost_shim.
    Only possible where /root/reference exists; returns the module path or None."""
    out = ref_nms_path()
    if not os.path.exists(REF_NMS_H):
        return out if os.path.exists(out) else None
    stub, shim = os.path.join(HERE, "ref_nms_stub.cpp"), os.path.join(HERE, "boost_shim", "boost", "geometry.hpp")
    if not _newer([stub, shim, REF_NMS_H], out):
        return out
    os.makedirs(REF_DIR, exist_ok=True)
    import sysconfig
    import pybind11
    cmd = ["g++", "-O2", "-std=c++14", "-shared", "-fPIC", "-w", "-ffp-contract=off", "-I", os.path.join(HERE, "boost_shim"), "-I",
           pybind11.get_include(), "-I", sysconfig.get_paths()["include"], '-DREF_NMS_CPU_H="%s"' % REF_NMS_H, stub, "-o", out]
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True)
    return out


if __name__ == "__main__":
    print(build_oracle(True))
    print(build_ref(True))
    print(build_ref_nms(True))
