# A/B of two builds of libipx inside ONE gpurun call, repetitions interleaved.
# usage: WL="--workload full" REPS=3 tools/ab_lib.sh tools/bin/libipx_prev.so imageprocessor_amd/libipx.so
run() { IPX_LIB=$PWD/$1 timeout -k 10 200 python bench.py --steps 20 --warmup 3 --cpu-seconds 0 --e2e-frames 0 $WL 2>gpurun_out/err.tmp | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-50s' % sys.argv[1], d['value'], d['roofline']['avg_launch_ms'], d['roofline']['frac'])" "$1"; }
for rep in $(seq ${REPS:-3}); do for lib in "$@"; do run $lib || exit 1; done; done
