#!/bin/bash
# Diagnostic build of libipx (phase stamps via IPX_STAMPS=1, ablations via IPX_DBG=1|2) -> tools/bin/libipx_diag.so.
# Use with IPX_LIB=$PWD/tools/bin/libipx_diag.so; never quote its run time.
set -e
cd "$(dirname "$0")/../imageprocessor_amd"
mkdir -p ../tools/bin
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -shared --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -DIPX_DIAG=1 \
    -o ../tools/bin/libipx_diag.so csrc/ipx_kernels.hip csrc/ipx_band.hip csrc/ipx_runtime.hip csrc/ipx_host.cpp csrc/ipx_ops.cpp
echo ../tools/bin/libipx_diag.so
