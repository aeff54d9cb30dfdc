# usage (on the GPU box): bash tools/collect_r03.sh 1|2   -- everything round 3's profiles/ and DESIGN.md quote, in two calls of about ten minutes
part=${1:-1}
if [ "$part" = 1 ]; then
python bench.py --steps 12 --warmup 3 > gpurun_out/r03_bench_full_1080p.json 2>/dev/null; echo "full done" >> gpurun_out/collect_progress.txt
bash tools/prof.sh r03_full > /dev/null 2>&1; echo "prof full done" >> gpurun_out/collect_progress.txt
python bench.py --workload resize --steps 12 --warmup 3 --e2e-frames 0 --cpu-seconds 4 > gpurun_out/r03_bench_resize.json 2>/dev/null
python bench.py --workload full-keepaspect --steps 12 --warmup 3 --e2e-frames 0 --cpu-seconds 0 --copy-gib 0 > gpurun_out/r03_bench_keepaspect.json 2>/dev/null
python bench.py --mixed 600 --steps 6 --warmup 2 > gpurun_out/r03_bench_mixed.json 2>/dev/null
python bench.py --width 3840 --height 2160 --frames 256 --steps 8 --warmup 2 --e2e-frames 0 --cpu-seconds 0 --copy-gib 0 > gpurun_out/r03_bench_4k.json 2>/dev/null
IPX_KS_FAST=0 python bench.py --steps 8 --warmup 3 --e2e-frames 0 --cpu-seconds 0 --copy-gib 0 > gpurun_out/r03_bench_full_float64.json 2>/dev/null
for f in full_1080p resize keepaspect mixed 4k full_float64; do python - <<PY
import json
d=json.loads(open("gpurun_out/r03_bench_$f.json").read().strip().splitlines()[-1])
print("$f", d["value"], d["ms_per_step"], d.get("checked"), d["roofline"]["frac"], d["roofline"].get("avg_launch_ms_by_set"), d["roofline"].get("frac_of_copy_ceiling"))
PY
done
head -14 gpurun_out/prof_r03_full/summary.txt
else
python tools/bench_sources.py 512 > gpurun_out/r03_sources.txt 2>/dev/null
python tools/bench_ycbcr.py 1024 full 3 >> gpurun_out/r03_sources.txt 2>/dev/null
IPX_KS_FAST=0 python tools/bench_ycbcr.py 1024 full 3 2>/dev/null | sed 's/^/float64 throughout: /' >> gpurun_out/r03_sources.txt
IPX_PROF_CMD="python3 tools/bench_ycbcr.py 1024 full 1" bash tools/prof.sh r03_ycc > /dev/null 2>&1; echo "prof ycc done" >> gpurun_out/collect_progress.txt
bash tools/prof.sh r03_resize --workload resize --steps 10 --warmup 2 > /dev/null 2>&1; echo "prof resize done" >> gpurun_out/collect_progress.txt
for n in 8 64 256 1024; do python tools/bench_j2j.py $n 4 2>/dev/null | tail -1; done > gpurun_out/r03_j2j.txt
python tools/bench_batcher.py > gpurun_out/r03_batcher.txt 2>/dev/null
cat gpurun_out/r03_sources.txt | grep -v amdgpu
cat gpurun_out/r03_j2j.txt; tail -8 gpurun_out/r03_batcher.txt
head -12 gpurun_out/prof_r03_ycc/summary.txt
head -8 gpurun_out/prof_r03_resize/summary.txt
fi
