# A compressed-in / compressed-out call of n files cut into parts of at least p files that run side by side on lanes of their own
# (IPX_JPEG_JPEG_PART; at most four parts): where cutting a medium-sized call pays.  usage: bash tools/j2j_small_parts.sh
for n in 64 128 256 512; do
  for p in 256 128 64 32; do
    [ $p -gt $n ] && continue
    echo -n "files $n, parts of >= $p: "
    IPX_JPEG_JPEG_PART=$p python3 tools/bench_j2j.py $n 5 2>&1 | tail -1
  done
done
