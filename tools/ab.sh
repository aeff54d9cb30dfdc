# A/B runs of bench.py inside ONE gpurun call (device-to-device spread is +-6 %, and run-to-run spread of one
# config is several % too: every config is run REPS times, interleaved).
# usage: WL="--workload full" REPS=3 tools/ab.sh "A=1" "IPX_PIPE_NT=512" ...
run() { env "$@" IPX_DEBUG=1 timeout -k 10 200 python bench.py --steps 20 --warmup 3 --cpu-seconds 0 --e2e-frames 0 $WL 2>gpurun_out/err.tmp | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-50s' % sys.argv[1], d['value'], d['roofline']['avg_launch_ms'], d['roofline']['frac'])" "$*"; }
for rep in $(seq ${REPS:-3}); do for cfg in "$@"; do run $cfg || exit 1; done; done
