# usage (on the GPU box): tools/pmc_insts.sh <tag> [bench args]  -> instruction counts per launch of the band kernels (one --pmc pass)
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/pmci_$tag
rm -rf $out; mkdir -p $out
# IPX_PMC_CMD="python3 tools/bench_ycbcr.py 1024": profile that instead of bench.py
if [ -n "$IPX_PMC_CMD" ]; then cmd="$IPX_PMC_CMD"; else cmd="python3 bench.py --steps 2 --warmup 1 --cpu-seconds 0 --e2e-frames 0 $*"; fi
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_WAIT_ANY --output-format csv -d $out -- $cmd > $out/log.txt 2>&1
python3 - "$out" "$tag" <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "band" in r["Kernel_Name"]:
            acc[(r["Kernel_Name"].split("(")[0][-60:], r["Counter_Name"])].append(float(r["Counter_Value"]))
print("==", sys.argv[2])
for (k, c), v in sorted(acc.items()):
    print("  %-62s %-18s %.4g" % (k, c, sum(v) / len(v)))
PY
