# usage: tools/prof.sh <tag> [bench args...]   (run on the GPU box; writes under gpurun_out/prof_<tag>/)
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/prof_$tag
mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 bench.py --cpu-seconds 0 --e2e-frames 0 "$@" > $out/trace.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --output-format csv -d $out/pmc_sq -- python3 bench.py --steps 3 --warmup 1 --cpu-seconds 0 --e2e-frames 0 "$@" > $out/pmc_sq.log 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM --output-format csv -d $out/pmc_sq2 -- python3 bench.py --steps 3 --warmup 1 --cpu-seconds 0 --e2e-frames 0 "$@" > $out/pmc_sq2.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -- python3 bench.py --steps 3 --warmup 1 --cpu-seconds 0 --e2e-frames 0 "$@" > $out/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -- python3 bench.py --steps 3 --warmup 1 --cpu-seconds 0 --e2e-frames 0 "$@" > $out/pmc_write.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE --output-format csv -d $out/pmc_tcc -- python3 bench.py --steps 3 --warmup 1 --cpu-seconds 0 --e2e-frames 0 "$@" > $out/pmc_tcc.log 2>&1
find $out -name "*.csv" | head -40
