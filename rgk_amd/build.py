"""Compile the HIP product library in-tree: rgk_amd/csrc/librgk_hip.so (gfx950)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(CSRC, "librgk_hip.so")
SOURCES = ["rgk_kernels.hip", "rgk_build.hip", "rgk_host.cpp", "rgk_output.cpp", "rgk_accum.cpp", "rgk_comm.cpp"]
HEADERS = ["rgk_kernels.h", "rgk_build.h", "rgk_device.h", "rgk_trace.h", "rgk_bdpt.h", "device_types.h", os.path.join("..", "..", "include", "rgk.h"), os.path.join("..", "..", "include", "rgk_libm.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
         # CPU/GPU agreement: no FMA contraction on either side (DESIGN.md "Numerics")
         "-ffp-contract=off", "-fno-fast-math", "-Wall", "-Wno-unused-function", "-x", "hip"]
LIBS = ["-ldl"]


def up_to_date():
    if not os.path.exists(LIB):
        return False
    t = os.path.getmtime(LIB)
    return all(os.path.getmtime(os.path.join(CSRC, f)) <= t for f in SOURCES + HEADERS)


def build(force=False, verbose=True, extra=(), out=None):
    """`extra` / `out`: tuning variants (-DRGK_TOP_NODES=...), built beside the product library."""
    if out is None and up_to_date() and not force:
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc] + FLAGS + list(extra) + [os.path.join(CSRC, s) for s in SOURCES] + ["-o", out or LIB] + LIBS
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    return out or LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
