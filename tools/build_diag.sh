#!/bin/bash
# Diagnostic build of libipx (phase stamps via IPX_STAMPS=1, ablations via IPX_DBG=1|2) -> tools/bin/libipx_diag.so.
# Use with IPX_LIB=$PWD/tools/bin/libipx_diag.so; never quote its run time.
set -e
cd "$(dirname "$0")/.."
mkdir -p tools/bin
srcs=$(python3 -c "import sys; sys.path.insert(0, 'imageprocessor_amd'); import build; print(' '.join('imageprocessor_amd/csrc/' + f for f in build.sources()))")
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -shared --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -DIPX_DIAG=1 \
    -o tools/bin/libipx_diag.so $srcs
echo tools/bin/libipx_diag.so
