# usage: tools/bench_workloads.sh [workloads...]   -- ms per 1024-frame launch of operator subsets (environment knobs pass through)
for w in ${@:-wm resize thumb resize-wm full}; do
python bench.py --workload $w --steps 6 --warmup 2 --cpu-seconds 0 --e2e-frames 0 --copy-gib 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$w', d['ms_per_step'], d['roofline']['avg_launch_ms_by_set'])"
done
