# rocprofv3 kernel statistics of the decode bench alone (tools/bench_jpeg_dec.py 1024), summary on stdout
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_dec -- python3 tools/bench_jpeg_dec.py ${1:-1024} > gpurun_out/prof_dec.log 2>&1
python3 - <<PY
import csv, glob
for f in glob.glob("gpurun_out/prof_dec/*/*_kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        print("  %-64s calls %4s avg %9.1f us min %9.1f max %9.1f" % (r["Name"][:64], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
PY
