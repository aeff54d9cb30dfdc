# rocprofv3 kernel statistics of the compressed-in / compressed-out bench (tools/bench_j2j.py N reps), summary on stdout
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_j2j
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_j2j -- python3 tools/bench_j2j.py ${1:-1024} ${2:-4} > gpurun_out/prof_j2j.log 2>&1
tail -2 gpurun_out/prof_j2j.log
python3 - <<PY
import csv, glob
rows = []
for f in glob.glob("gpurun_out/prof_j2j/*/*_kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        rows.append((float(r["TotalDurationNs"]) if "TotalDurationNs" in r else float(r["AverageNs"]) * int(r["Calls"]), r))
tot = sum(t for t, _ in rows)
for t, r in sorted(rows, key=lambda x: -x[0])[:22]:
    print("  %-60s calls %5s avg %9.1f us total %8.2f ms %5.1f%%" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3, t / 1e6, 100 * t / tot))
print("  total kernel time %.1f ms" % (tot / 1e6))
PY
