python bench.py --workload resize --steps 12 --warmup 3 --e2e-frames 0 --cpu-seconds 4 > gpurun_out/r03_bench_resize.json 2>/dev/null
python bench.py --workload full-keepaspect --steps 12 --warmup 3 --e2e-frames 0 --cpu-seconds 0 --copy-gib 0 > gpurun_out/r03_bench_keepaspect.json 2>/dev/null
python bench.py --mixed 600 --steps 6 --warmup 2 > gpurun_out/r03_bench_mixed.json 2>/dev/null
python bench.py --width 3840 --height 2160 --frames 256 --steps 8 --warmup 2 --e2e-frames 0 --cpu-seconds 0 --copy-gib 0 > gpurun_out/r03_bench_4k.json 2>/dev/null
python tools/bench_sources.py 512 > gpurun_out/r03_sources.txt 2>/dev/null
python tools/bench_ycbcr.py 1024 full 3 >> gpurun_out/r03_sources.txt 2>/dev/null
IPX_PROF_CMD="python3 tools/bench_ycbcr.py 1024 full 1" bash tools/prof.sh r03_ycc > /dev/null 2>&1
bash tools/prof.sh r03_resize --workload resize --steps 10 --warmup 2 > /dev/null 2>&1
for f in resize keepaspect mixed 4k; do python - <<PY
import json
d=json.loads(open("gpurun_out/r03_bench_$f.json").read().strip().splitlines()[-1])
print("$f", d["value"], d["ms_per_step"], d.get("checked"), d["roofline"]["frac"], d["roofline"].get("avg_launch_ms_by_set"))
PY
done
cat gpurun_out/r03_sources.txt | grep -v amdgpu
head -12 gpurun_out/prof_r03_ycc/summary.txt
head -8 gpurun_out/prof_r03_resize/summary.txt
