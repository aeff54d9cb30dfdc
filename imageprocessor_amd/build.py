"""Builds libipx.so (HIP kernels + runtime + C ABI) for gfx950 with hipcc, in-tree."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libipx.so")
SOURCES = ["ipx_kernels.hip", "ipx_band.hip", "ipx_band_ycc.hip", "ipx_band_nrgba.hip", "ipx_jpeg.hip", "ipx_jpeg_entropy.hip", "ipx_jpeg_dec.hip", "ipx_jpeg_dec_par.hip", "ipx_runtime.hip", "ipx_jpeg_runtime.hip", "ipx_host.cpp", "ipx_ops.cpp", "ipx_font.cpp", "ipx_jpeg_host.cpp", "ipx_jpeg_dec_host.cpp"]
HEADERS = ["ipx_internal.h", "ipx_runtime_internal.h", "ipx_device.h", "ipx_band_common.h", os.path.join("..", "..", "include", "ipx.h")]
# -ffp-contract=off: the bilinear taps must round every float64 product before the add, as the
# reference's GOAMD64=v1 build does (no FMA); the kernels also carry `#pragma clang fp contract(off)`.
FLAGS = ["-O3", "-std=c++17", "-fPIC", "-shared", "--offload-arch=gfx950", "-ffp-contract=off",
         "-fno-fast-math", "-Wall", "-Wno-unused-function"]


def hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return "hipcc"


def stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(os.path.join(CSRC, f)) > t for f in SOURCES + HEADERS)


def build(force=False, verbose=False):
    if not force and not stale():
        return LIB
    cmd = [hipcc()] + FLAGS + ["-o", LIB] + [os.path.join(CSRC, f) for f in SOURCES]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv, verbose=True)
    print(LIB)
