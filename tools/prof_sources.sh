# rocprofv3 kernel statistics and HBM counters of the source-type benches (tools/bench_sources.py), summary on stdout
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/prof_sources
mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out/src -- python3 tools/bench_sources.py 512 > $out/src.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/src_fetch -- python3 tools/bench_sources.py 512 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/src_write -- python3 tools/bench_sources.py 512 > /dev/null 2>&1
python3 - $out <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
tag = "src"
def short(n):      # the template arguments tell the instantiations of band_conv_kernel apart
    for a in ("void ", "ipx::(anonymous namespace)::", "ipx::", "(anonymous namespace)::"):
        n = n.replace(a, "")
    return n.split("(")[0][:96]
print("==", tag, "\n   " + "\n   ".join(l for l in open(out + "/" + tag + ".log").read().strip().split("\n") if "amdgpu.ids" not in l))
for f in glob.glob(out + "/" + tag + "/*/*_kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        print("  %-96s calls %4s avg %9.1f us min %9.1f max %9.1f" % (short(r["Name"]), r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
for d, ctr, mul in ((tag + "_fetch", "FETCH_SIZE", 2048), (tag + "_write", "WRITE_SIZE", 1024)):
    for f in glob.glob(out + "/" + d + "/*/*_counter_collection.csv"):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "ipx" in r["Kernel_Name"]:
                acc[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
        for k, v in acc.items():
            print("  %-96s %s bytes per launch (x%d): %.4g  (n=%d)" % (k, ctr, mul, max(v) * mul, len(v)))
PY
