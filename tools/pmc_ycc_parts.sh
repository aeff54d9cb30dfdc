# usage (on the GPU box): tools/pmc_ycc_parts.sh  -> VALU / SALU / LDS instruction counts and time of band_conv_kernel per operator set
for ops in wm resize thumb full; do
  IPX_PMC_CMD="python3 tools/bench_ycbcr.py 1024 $ops" bash tools/pmc_insts.sh ycc_$ops 2>&1 | grep -E "==|VALU|SALU|INSTS_LDS|WAIT_ANY|WAVE_CYCLES"
  python3 tools/bench_ycbcr.py 1024 $ops 2>&1 | grep frames
done
