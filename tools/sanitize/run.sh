#!/bin/bash
# Builds the host-side parsers with AddressSanitizer + UBSan (CPU only; GPU sanitizers are not available on the pool) and feeds them
# mutated JPEG headers and TrueType files.  shift-base is off: the 26.6 fixed-point code shifts negative values left on purpose (as the Go code
# it restates does; defined in C++20, and neither gcc nor clang treats it otherwise).  usage: tools/sanitize/run.sh [cases per seed file]
set -e
cd "$(dirname "$0")/../.."
out=/tmp/ipx_sanitize
mkdir -p $out
python3 - $out <<'PY'
import io, sys
import numpy as np
from PIL import Image
out = sys.argv[1]
yy, xx = np.mgrid[0:120, 0:160]
img = np.stack([np.sin(xx / 9.0) * 100 + 128, np.cos(yy / 7.0) * 100 + 128, (xx * 2 + yy) % 256], -1).clip(0, 255).astype(np.uint8)
for name, kw in (("a", {}), ("b", {"restart_marker_rows": 1}), ("c", {"optimize": True, "subsampling": 0}), ("d", {"progressive": True}), ("e", {"progressive": True, "subsampling": 0, "optimize": True})):
    Image.fromarray(img).save("%s/%s.jpg" % (out, name), "JPEG", quality=85, **kw)
Image.fromarray(img[..., 0]).save(out + "/g.jpg", "JPEG", quality=80)
PY
CXX="g++ -std=c++17 -O1 -g -fsanitize=address,undefined -fno-sanitize-recover=undefined -fno-sanitize=shift-base -fno-omit-frame-pointer -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include"
$CXX -o $out/fuzz_parsers tools/sanitize/fuzz_parsers.cpp imageprocessor_amd/csrc/ipx_font.cpp imageprocessor_amd/csrc/ipx_jpeg_dec_host.cpp imageprocessor_amd/csrc/ipx_jpeg_dec_prog.cpp
fonts=$(ls /usr/share/fonts/truetype/dejavu/DejaVuSans.ttf /usr/share/fonts/truetype/dejavu/DejaVuSerif-Bold.ttf 2>/dev/null || true)
ASAN_OPTIONS=detect_leaks=1:abort_on_error=1 UBSAN_OPTIONS=print_stacktrace=1 $out/fuzz_parsers ${1:-3000} $out/*.jpg $fonts
