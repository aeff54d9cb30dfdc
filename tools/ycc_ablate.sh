# usage (GPU box): tools/ycc_ablate.sh -> run time of band_conv_kernel with parts switched off (diagnostic build; never quoted as performance)
export IPX_LIB=$PWD/tools/bin/libipx_diag.so
for ops in full wm; do
  for dbg in 0 1 4 5 16 17 21 8 9; do
    echo -n "ops=$ops dbg=$dbg  "; IPX_DBG=$dbg python3 tools/bench_ycbcr.py 1024 $ops 2>&1 | grep frames | sed 's/.*frames in//; s/=.*//'
  done
done
