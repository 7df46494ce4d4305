"""In-tree build of libprt_hip.so for gfx950 (hipcc cross-compiles without a GPU)."""
import os
import shutil
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB_DIR = os.path.join(_HERE, "lib")
LIB = os.path.join(LIB_DIR, "libprt_hip.so")
TEST_LIB = os.path.join(LIB_DIR, "libprt_hip_test.so")  # same sources + -DPRT_TEST_ENTRY_POINTS (include/prt_hip_test.h)

SOURCES = [
    "prt_kernels.hip",
    "prt_gather.hip",
    "prt_bvh_build.hip",
    "host/prt_host.cpp",
    "host/prt_bvh.cpp",
    "host/prt_models.cpp",
    "host/prt_host_capi.cpp",
]
# every header under csrc/ (globbed: a new header can never be forgotten here) plus the interface headers: needs_build() and
# source_sha16() both derive from these lists, so a stale library can neither be loaded silently nor stamp a measurement
KERNEL_HEADERS = sorted(f for f in os.listdir(CSRC) if f.endswith(".h"))
HEADERS = KERNEL_HEADERS + ["host/prt.h", "host/cornell_data.inc", "../../include/prt_hip.h", "../../include/prt_host.h",
                            "../../include/prt_hip_test.h"]

# -ffp-contract=off: the reference's object code has no FMA, and results must match it bit for bit.
# No fast-math; HIP's default correctly-rounded f32 divide/sqrt is kept.
# -fno-slp-vectorize: the SLP vectoriser's v_pk_* pairs cost more v_mov shuffling than they save (C3 frame 591 -> 551 ms,
# measured); the node step keeps its hand-packed (lo, hi) slab pairs, whose operands come out of the loads already paired.
# (What bounds the trace kernels is in DESIGN.md section 4, from the counter files under profiles/.)
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-slp-vectorize", "-fPIC", "-shared", "-pthread", "-ldl",
         "-Wall", "-Wno-unused-function", "-Wno-unused-variable"]


def hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: the HIP library cannot be built")
    return exe


def needs_build(lib=LIB):
    if not os.path.exists(lib):
        return True
    t = os.path.getmtime(lib)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


def build_library(force=False, verbose=False, test_entry_points=False):
    lib = TEST_LIB if test_entry_points else LIB
    if not force and not needs_build(lib):
        return lib
    os.makedirs(LIB_DIR, exist_ok=True)
    tmp = f"{lib}.tmp.{os.getpid()}"  # concurrent builders (several ranks) never share a partial file
    cmd = [hipcc()] + FLAGS + [f'-DPRT_SOURCE_SHA16="{source_sha16()}"'] + (["-DPRT_TEST_ENTRY_POINTS"] if test_entry_points else []) + \
        [os.path.join(CSRC, s) for s in SOURCES] + ["-o", tmp]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    os.replace(tmp, lib)
    return lib


def source_sha16():
    """First 16 hex digits of the SHA-256 over the kernel sources: profiles/*.json carry it, so that bench.py can tell whether
    a committed counter summary was taken with the code it is running."""
    import hashlib
    h = hashlib.sha256()
    for name in sorted([s for s in SOURCES if s.endswith(".hip")] + KERNEL_HEADERS):
        with open(os.path.join(CSRC, name), "rb") as f:
            h.update(name.encode() + b"\0" + f.read())
    return h.hexdigest()[:16]
