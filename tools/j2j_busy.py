#!/usr/bin/env python3
"""How busy the GPU is during compressed-in / compressed-out batches: the union of the kernel intervals of a rocprofv3 kernel trace
(tools/prof_j2j.sh writes one) against the span they cover, per repetition of the bench.  usage: tools/j2j_busy.py gpurun_out/prof_j2j"""
import csv
import glob
import sys

root = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/prof_j2j"
for f in glob.glob(root + "/*/*_kernel_trace.csv"):
    iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f)))
    # repetitions are separated by idle gaps of more than 3 ms (the bench's host work between calls)
    groups, cur = [], [iv[0]]
    end = iv[0][1]
    for s, e, n in iv[1:]:
        if s - end > 3_000_000:
            groups.append(cur)
            cur = []
        cur.append((s, e, n))
        end = max(end, e)
    groups.append(cur)
    for g in groups:
        span = max(e for _, e, _ in g) - g[0][0]
        busy, last = 0, g[0][0]
        for s, e, _ in g:
            if e > last:
                busy += e - max(s, last)
                last = e
        total = sum(e - s for s, e, _ in g)
        print("%5d kernels over %7.2f ms: GPU busy %7.2f ms = %5.1f %%, kernel time summed %7.2f ms (overlap factor %.2f)"
              % (len(g), span / 1e6, busy / 1e6, 100.0 * busy / span, total / 1e6, total / max(busy, 1)))
