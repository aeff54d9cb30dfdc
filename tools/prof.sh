# usage: tools/prof.sh <tag> [bench args...]   (run on the GPU box; writes under gpurun_out/prof_<tag>/)
# IPX_PROF_CMD="python3 tools/bench_ycbcr.py 1024": profile that command instead of bench.py
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/prof_$tag
rm -rf $out; mkdir -p $out
if [ -n "$IPX_PROF_CMD" ]; then long="$IPX_PROF_CMD"; short="$IPX_PROF_CMD"
else long="python3 bench.py --cpu-seconds 0 --e2e-frames 0 --copy-gib 0 $*"; short="python3 bench.py --steps 3 --warmup 1 --cpu-seconds 0 --e2e-frames 0 --copy-gib 0 $*"; fi
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- $long > $out/trace.log 2>&1; echo "trace done" >> $out/progress.txt
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --output-format csv -d $out/pmc_sq -- $short > $out/pmc_sq.log 2>&1; echo "sq done" >> $out/progress.txt
rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM --output-format csv -d $out/pmc_sq2 -- $short > $out/pmc_sq2.log 2>&1; echo "sq2 done" >> $out/progress.txt
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -- $short > $out/pmc_fetch.log 2>&1; echo "fetch done" >> $out/progress.txt
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -- $short > $out/pmc_write.log 2>&1; echo "write done" >> $out/progress.txt
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE --output-format csv -d $out/pmc_tcc -- $short > $out/pmc_tcc.log 2>&1; echo "tcc done" >> $out/progress.txt
python3 tools/summarize_prof.py $out > $out/summary.txt 2>&1
