# A/B of two builds of the library on one box: tools/ab_lib.sh <other.so> -- alternates IPX_LIB between the tree's libipx.so and the other one
other=$1
for r in 1 2 3; do
  for lib in "" "$other"; do
    j=$(IPX_LIB=$lib timeout -k 10 100 python tools/bench_j2j.py 1024 5 2>&1 | grep 'images/s' | sed 's/.*files in//')
    d=$(IPX_LIB=$lib timeout -k 10 100 python tools/bench_jpeg_dec.py 1024 2>&1 | grep 'GPU decode' | sed 's/.*batch of  1024: //; s/ frames.*//' | tr '\n' ' ')
    echo "${lib:-tree}: j2j $j | decode $d"
  done
done
