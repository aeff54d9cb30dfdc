#!/bin/bash
# A variant build of libipx for A/B runs: tools/build_variant.sh <name> <extra hipcc flags...> -> tools/bin/libipx_<name>.so (use with IPX_LIB=...)
set -e
cd "$(dirname "$0")/.."
name=$1; shift
mkdir -p tools/bin
srcs=$(python3 -c "import sys; sys.path.insert(0, 'imageprocessor_amd'); import build; print(' '.join('imageprocessor_amd/csrc/' + f for f in build.sources()))")
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -shared --offload-arch=gfx950 -ffp-contract=off -fno-fast-math "$@" -o tools/bin/libipx_$name.so $srcs
echo tools/bin/libipx_$name.so
