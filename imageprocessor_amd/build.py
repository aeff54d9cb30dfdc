"""Builds libipx.so (HIP kernels + runtime + C ABI) for gfx950 with hipcc, in-tree.

Every source becomes an object under imageprocessor_amd/build/ (rebuilt when it or a header is newer), objects are compiled
in parallel and linked into imageprocessor_amd/libipx.so."""
import concurrent.futures as cf
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "build")
LIB = os.path.join(HERE, "libipx.so")
SOURCES = ["ipx_kernels.hip", "ipx_ks_generic.hip", "ipx_ks_fused.hip", "ipx_jpeg.hip", "ipx_jpeg_entropy.hip", "ipx_jpeg_dec.hip", "ipx_jpeg_dec_par.hip", "ipx_runtime.hip", "ipx_jpeg_runtime.hip", "ipx_pool.hip", "ipx_host.cpp", "ipx_ks_host.cpp", "ipx_batcher.cpp", "ipx_ops.cpp", "ipx_font.cpp", "ipx_jpeg_host.cpp", "ipx_jpeg_dec_host.cpp", "ipx_jpeg_dec_prog.cpp"]
HEADERS = ["ipx_internal.h", "ipx_runtime_internal.h", "ipx_device.h", "ipx_ks.h", "ipx_threads.h", "ipx_batcher.h", "ipx_pool_core.h", os.path.join("..", "..", "include", "ipx.h")]
# -ffp-contract=off: the kernel scaler must round every float64 product before the add, as the
# reference's GOAMD64=v1 build does (no FMA); the kernels also carry `#pragma clang fp contract(off)`.
FLAGS = ["-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=off",
         "-fno-fast-math", "-Wall", "-Wno-unused-function"] + os.environ.get("IPX_CXXFLAGS", "").split()   # e.g. -DIPX_DIAG=1: phase stamps


def hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return "hipcc"


def sources():
    return [f for f in SOURCES if os.path.exists(os.path.join(CSRC, f))]


def _obj(src):
    return os.path.join(OBJ, src + ".o")


def _newest_header():
    return max(os.path.getmtime(os.path.join(CSRC, h)) for h in HEADERS if os.path.exists(os.path.join(CSRC, h)))


def _stale_objects(force):
    hdr = _newest_header()
    out = []
    for f in sources():
        o = _obj(f)
        if force or not os.path.exists(o) or os.path.getmtime(o) < max(hdr, os.path.getmtime(os.path.join(CSRC, f))):
            out.append(f)
    return out


def stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(os.path.join(CSRC, f)) > t for f in sources() + HEADERS if os.path.exists(os.path.join(CSRC, f)))


def build(force=False, verbose=False, jobs=None):
    if not force and not stale():
        return LIB
    os.makedirs(OBJ, exist_ok=True)
    todo = _stale_objects(force)

    def compile_one(f):
        cmd = [hipcc()] + FLAGS + ["-c", "-o", _obj(f), os.path.join(CSRC, f)]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)

    jobs = jobs or max(1, min(len(todo), (os.cpu_count() or 2)))
    if todo:
        with cf.ThreadPoolExecutor(jobs) as ex:
            list(ex.map(compile_one, todo))
    cmd = [hipcc(), "-shared", "-fPIC", "--offload-arch=gfx950", "-o", LIB] + [_obj(f) for f in sources()]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv, verbose=True)
    print(LIB)
