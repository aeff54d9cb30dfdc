#!/usr/bin/env python3
"""Timeline of the LAST compressed-in / compressed-out call in a rocprofv3 kernel trace (tools/prof_j2j.sh writes one): every kernel with
its start relative to the call's first kernel, its duration and the idle gap before it.  Says where a small call's milliseconds go
(dependent kernels, host round trips between them).  usage: tools/j2j_timeline.py gpurun_out/prof_j2j"""
import csv
import glob
import sys

root = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/prof_j2j"
for f in glob.glob(root + "/*/*_kernel_trace.csv"):
    iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f)))
    groups, cur, end = [], [iv[0]], iv[0][1]
    for s, e, n in iv[1:]:
        if s - end > 1_500_000:
            groups.append(cur)
            cur = []
        cur.append((s, e, n))
        end = max(end, e)
    groups.append(cur)
    g = groups[-1]
    t0, last = g[0][0], g[0][0]
    busy = 0
    for s, e, n in g:
        print("%8.1f us  +%6.1f us gap  %7.1f us  %s" % ((s - t0) / 1e3, max(0, s - last) / 1e3, (e - s) / 1e3, n[:90]))
        if e > last:
            busy += e - max(s, last)
            last = e
    print("%d kernels, span %.1f us, busy %.1f us" % (len(g), (last - t0) / 1e3, busy / 1e3))
