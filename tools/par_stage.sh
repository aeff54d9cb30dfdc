# Compressed-in / compressed-out calls of 1..1024 files with the Huffman passes reading the scan through L1 / L2 (IPX_JPEG_PAR_STAGE=0) and
# from rows staged in LDS (=1): where the staged form stops paying.  usage: bash tools/par_stage.sh [sizes...]
for n in ${@:-1 8 64 256}; do
  for st in 0 1; do
    echo -n "files $n stage $st: "
    IPX_JPEG_PAR_STAGE=$st python3 tools/bench_j2j.py $n 5 2>&1 | tail -1
  done
done
