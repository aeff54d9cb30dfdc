# usage (GPU box): bash tools/fix_grid.sh  -- the exact pass's duration for a few block sizes / block budgets.  The knobs it sets
# (IPX_FIX_THREADS, IPX_FIX_BLOCKS) existed only in the build this was run with; launch_ks_fix carries the result as constants.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for cfg in "256 8192" "256 2048" "256 1024" "512 4096" "1024 4096" "1024 2048" "1024 1024"; do set -- $cfg
rm -rf gpurun_out/prof_fixg
IPX_FIX_THREADS=$1 IPX_FIX_BLOCKS=$2 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_fixg -- python3 bench.py --steps 5 --warmup 2 --cpu-seconds 0 --e2e-frames 0 --copy-gib 0 > /dev/null 2>&1
python3 - "$1" "$2" <<'PY'
import csv,glob,sys
f=glob.glob('gpurun_out/prof_fixg/*/*kernel_stats.csv')[0]
r={x['Name'][:50]:float(x['AverageNs'])/1e3 for x in csv.DictReader(open(f)) if 'fix' in x['Name']}
print("threads %s blocks %s:" % (sys.argv[1], sys.argv[2]), {k[-14:]:round(v,1) for k,v in r.items()}, flush=True)
PY
done
