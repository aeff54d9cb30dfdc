#!/usr/bin/env python3
"""Condenses a tools/prof.sh output directory (rocprofv3 CSVs) into one text summary + JSON:
per-kernel duration statistics from --kernel-trace --stats, mean PMC values per kernel from the
separate --pmc passes, and HBM bytes per launch with the gfx950 corrections of
MI355X_MICROARCH.md (FETCH_SIZE counts half of a 16-B/lane stream; WRITE_SIZE is exact; unit KiB)."""
import collections
import csv
import glob
import json
import os
import sys

src = sys.argv[1]
out = {"kernel_stats": [], "pmc": {}, "hbm": {}}
for f in glob.glob(os.path.join(src, "trace", "*", "*_kernel_stats.csv")):
    for r in csv.DictReader(open(f)):
        out["kernel_stats"].append({k: r[k] for k in ("Name", "Calls", "AverageNs", "MinNs", "MaxNs", "Percentage")})
for d in sorted(glob.glob(os.path.join(src, "pmc_*"))):
    if not os.path.isdir(d):
        continue
    for f in glob.glob(os.path.join(d, "*", "*_counter_collection.csv")):
        acc = collections.defaultdict(list)
        meta = {}
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"]
            if "ipx" not in name:
                continue
            acc[(name, r["Counter_Name"])].append(float(r["Counter_Value"]))
            meta[name] = {"VGPR_Count": r.get("VGPR_Count"), "SGPR_Count": r.get("SGPR_Count"),
                          "LDS_Block_Size": r.get("LDS_Block_Size"), "Workgroup_Size": r.get("Workgroup_Size"),
                          "Grid_Size": r.get("Grid_Size")}
        for (name, ctr), v in acc.items():
            k = out["pmc"].setdefault(name, {"launches_sampled": len(v)})
            k[ctr] = sum(v) / len(v)
            k.update(meta[name])
for name, c in out["pmc"].items():
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        rd = c["FETCH_SIZE"] * 1024 * 2      # gfx950: FETCH_SIZE reports half of a 16-B/lane stream
        wr = c["WRITE_SIZE"] * 1024
        out["hbm"][name] = {"read_bytes_per_launch": rd, "write_bytes_per_launch": wr, "total": rd + wr}
print("== kernel-trace --stats")
for k in out["kernel_stats"]:
    print("  %-70s calls %4s  avg %10.1f us  min %10.1f  max %10.1f  (%s%%)" % (
        k["Name"][:70], k["Calls"], float(k["AverageNs"]) / 1e3, float(k["MinNs"]) / 1e3, float(k["MaxNs"]) / 1e3, k["Percentage"]))
# the same from the trace itself, without the first launches: bench.py's warm-up steps are each buffer set's first launch (first touch of
# 20 GB, up to 1.7x slower) and rocprofv3's statistics average them in; bench.py's own clock starts after them
trace = collections.defaultdict(list)
for f in glob.glob(os.path.join(src, "trace", "*", "*_kernel_trace.csv")):
    for r in csv.DictReader(open(f)):
        trace[r["Kernel_Name"]].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
print("== kernel-trace, launches after the first three of each kernel (bench.py's warm-up steps)")
for name, v in trace.items():
    if "ipx" not in name or len(v) <= 3:
        continue
    d = [x[1] for x in sorted(v)[3:]]
    out.setdefault("timed", {})[name] = {"launches": len(d), "avg_us": sum(d) / len(d) / 1e3, "min_us": min(d) / 1e3, "max_us": max(d) / 1e3}
    print("  %-70s launches %4d  avg %10.1f us  min %10.1f  max %10.1f" % (name[:70], len(d), sum(d) / len(d) / 1e3, min(d) / 1e3, max(d) / 1e3))
print("== PMC means per launch (separate --pmc passes)")
for name, c in out["pmc"].items():
    print("  " + name[:100])
    for ctr, v in sorted(c.items()):
        print("    %-26s %s" % (ctr, ("%.6g" % v) if isinstance(v, float) else v))
print("== HBM bytes per launch (FETCH_SIZE x 1024 x 2, WRITE_SIZE x 1024)")
for name, h in out["hbm"].items():
    print("  %-60s read %.4g  write %.4g  total %.4g" % (name[:60], h["read_bytes_per_launch"], h["write_bytes_per_launch"], h["total"]))
json.dump(out, open(os.path.join(src, "summary.json"), "w"), indent=1)
