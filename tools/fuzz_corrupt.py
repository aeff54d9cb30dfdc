#!/usr/bin/env python3
"""Corrupted JPEGs through the GPU decoder: it must never crash or hang, must agree with the oracle on the verdict, and -- where the
damaged file still decodes -- on every byte.  usage: tools/fuzz_corrupt.py [cases] [seed]"""
import io
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
from PIL import Image, ImageFile  # noqa: E402

import imageprocessor_amd as ipx  # noqa: E402
import oracle  # noqa: E402

ImageFile.MAXBLOCK = 1 << 26
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 300
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 3)
ctx = ipx.Context()
yy, xx = np.mgrid[0:333, 0:500]
base = np.stack([np.sin(xx / 9.0) * 100 + 128, np.cos(yy / 7.0) * 100 + 128, (xx * 2 + yy) % 256], -1)
img = (base + rng.normal(0, 10, base.shape)).clip(0, 255).astype(np.uint8)
clean = []
for kw in ({}, {"restart_marker_rows": 1}, {"optimize": True, "subsampling": 0}, {"subsampling": 1, "quality": 95}, {"progressive": True},
           {"progressive": True, "subsampling": 0, "optimize": True}):
    buf = io.BytesIO()
    Image.fromarray(img).save(buf, "JPEG", **{"quality": 85, **kw})
    clean.append(buf.getvalue())
buf = io.BytesIO()
Image.fromarray(img[..., 0]).save(buf, "JPEG", quality=80)
clean.append(buf.getvalue())
verdicts = {"ok": 0, "malformed": 0, "unsupported": 0}
bad = 0
for t in range(cases):
    f = bytearray(clean[t % len(clean)])
    sos = f.index(b"\xff\xda")
    kind = rng.integers(0, 5)
    if kind == 0:      # flips inside the scan
        for _ in range(int(rng.integers(1, 6))):
            f[int(rng.integers(sos + 14, len(f) - 2))] ^= 1 << int(rng.integers(0, 8))
    elif kind == 1:    # random bytes inside the scan (may create markers)
        for _ in range(int(rng.integers(1, 4))):
            f[int(rng.integers(sos + 14, len(f) - 2))] = int(rng.integers(0, 256))
    elif kind == 2:    # flips in the headers
        for _ in range(int(rng.integers(1, 4))):
            f[int(rng.integers(2, sos + 14))] ^= 1 << int(rng.integers(0, 8))
    elif kind == 3:    # truncation
        f = f[:int(rng.integers(sos, len(f)))]
    else:              # a chunk of the scan removed
        a = int(rng.integers(sos + 14, len(f) - 10))
        del f[a:a + int(rng.integers(1, 2000))]
    f = bytes(f)
    try:
        want = oracle.jpeg_decode(f)
        verdict = "ok"
    except ValueError as e:
        want, verdict = None, str(e)
    verdicts[verdict] += 1
    info, st = ctx.jpeg_decode_batch([f, clean[t % len(clean)]])     # the damaged file next to a good one of the same kind
    exp = {"ok": 0, "malformed": -1, "unsupported": -4}[verdict]
    if want is not None and want["dc_wide"]:
        exp, verdict = -4, "unsupported"    # a DC value beyond int16: Go's int32 arithmetic decodes it, the GPU pipeline hands it back
    same_shape = want is None or info is None or (want["w"], want["h"], want["ratio"]) == (info["w"], info["h"], info["ratio"])
    if verdict == "ok" and not same_shape:
        continue                                                       # the damage changed the size: the batch rule refuses one of the two, fine
    if st[0] != exp and not (exp != 0 and st[0] in (-1, -4)):   # a broken file is refused either way; which of Go's two error kinds it earns is not part of the contract
        bad += 1
        print("VERDICT MISMATCH case", t, "kind", int(kind), "oracle", verdict, "gpu", st[0])
        open("gpurun_out/corrupt_fail_%d.jpg" % t, "wb").write(f)
    elif verdict == "ok":
        for k in ("y", "cb", "cr") if want["ratio"] != 4 else ("y",):
            if not np.array_equal(info[k][0], want[k]):
                bad += 1
                print("PLANE MISMATCH case", t, "kind", int(kind), k)
                open("gpurun_out/corrupt_fail_%d.jpg" % t, "wb").write(f)
                break
    if t % 50 == 49:
        print("...", t + 1, "cases", verdicts, bad, "mismatches", flush=True)
print("done:", cases, "cases", verdicts, bad, "mismatches")
