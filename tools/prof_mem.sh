# usage: tools/prof_mem.sh <tag> [bench args...]  -- HBM traffic counters only (FETCH_SIZE / WRITE_SIZE in separate passes)
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/prof_$tag
mkdir -p $out
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -- python3 bench.py --steps 3 --warmup 1 --cpu-seconds 0 --e2e-frames 0 "$@" > $out/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -- python3 bench.py --steps 3 --warmup 1 --cpu-seconds 0 --e2e-frames 0 "$@" > $out/pmc_write.log 2>&1
python3 - $out <<'PY'
import csv,glob,sys,collections
out=sys.argv[1]
for d in ["pmc_fetch","pmc_write"]:
    for f in glob.glob(out+"/"+d+"/*/*_counter_collection.csv"):
        acc=collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "band" in r["Kernel_Name"]:
                acc[(r["Kernel_Name"][:60],r["Counter_Name"])].append(float(r["Counter_Value"]))
        for k,v in acc.items():
            print(out,k,"n=%d"%len(v),"mean_KB=%.6g"%(sum(v)/len(v)))
PY
