# usage (GPU box): tools/prof_bimodal.sh -> per-dispatch duration and memory-side stall counters of band_pipe_kernel for several buffer sets
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/prof_bimodal
rm -rf $out; mkdir -p $out
for pass in "TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_STALL_sum" "TCC_TOO_MANY_EA_WRREQS_STALL_sum TCC_TAG_STALL_sum GRBM_UTCL2_BUSY"; do
rm -rf $out; mkdir -p $out
timeout -k 10 150 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $out -- python3 tools/bimodal.py 5 1024 > $out/log.txt 2>&1 || { echo "pass failed"; tail -3 $out/log.txt; exit 1; }
python3 - $out <<'PY'
import csv, glob, sys, collections
d = sys.argv[1]
dur = {}
for f in glob.glob(d + "/*/*_kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        if "band_pipe" in r["Kernel_Name"]:
            dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
cnt = collections.defaultdict(dict)
for f in glob.glob(d + "/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "band_pipe" in r["Kernel_Name"]:
            cnt[r["Dispatch_Id"]][r["Counter_Name"]] = float(r["Counter_Value"])
names = sorted({k for v in cnt.values() for k in v})
print("dispatch ms " + " ".join(names))
for k in sorted(cnt, key=int):
    print(k, "%.3f" % dur.get(k, -1), " ".join("%.4g" % cnt[k].get(n, -1) for n in names))
PY
done
