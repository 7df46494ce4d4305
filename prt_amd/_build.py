"""In-tree build of libprt_hip.so for gfx950 (hipcc cross-compiles without a GPU)."""
import os
import shutil
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB_DIR = os.path.join(_HERE, "lib")
LIB = os.path.join(LIB_DIR, "libprt_hip.so")

SOURCES = [
    "prt_kernels.hip",
    "host/prt_host.cpp",
    "host/prt_bvh.cpp",
    "host/prt_models.cpp",
    "host/prt_host_capi.cpp",
]
HEADERS = ["prt_device.h", "prt_devmath.h", "host/prt.h", "host/cornell_data.inc", "../../include/prt_hip.h",
           "../../include/prt_host.h"]

# -ffp-contract=off: the reference's object code has no FMA, and results must match it bit for bit.
# No fast-math; HIP's default correctly-rounded f32 divide/sqrt is kept.
# -fno-slp-vectorize: the trace kernels are VALU-issue-bound (PMC: 75 % busy at 24 of 64 lanes per instruction); the SLP
# vectoriser's v_pk_* pairs cost more v_mov shuffling than they save (C3 frame 591 -> 551 ms).  The node step keeps its
# hand-packed (lo, hi) slab pairs, whose operands come out of the loads already paired.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-slp-vectorize", "-fPIC", "-shared", "-pthread",
         "-Wall", "-Wno-unused-function", "-Wno-unused-variable"]


def hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: the HIP library cannot be built")
    return exe


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


def build_library(force=False, verbose=False):
    if not force and not needs_build():
        return LIB
    os.makedirs(LIB_DIR, exist_ok=True)
    tmp = f"{LIB}.tmp.{os.getpid()}"  # concurrent builders (several ranks) never share a partial file
    cmd = [hipcc()] + FLAGS + [os.path.join(CSRC, s) for s in SOURCES] + ["-o", tmp]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    os.replace(tmp, LIB)
    return LIB
