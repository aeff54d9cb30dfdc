# tools/j2j_parts.sh -- how a compressed-in / compressed-out batch of 1024 files responds to the number of parts it is cut into
# (IPX_JPEG_JPEG_MAXPARTS / _PART), the lanes of the context and the cap on concurrently running parts (IPX_JPEG_JPEG_PARALLEL)
for round in 1 2 3; do
for cfg in "4 3 384 3" "4 4 256 4" "6 4 256 4" "6 5 200 5" "4 3 256 3"; do set -- $cfg; echo "lanes $1 maxparts $2 part $3 parallel $4: $(IPX_BENCH_LANES=$1 IPX_JPEG_JPEG_MAXPARTS=$2 IPX_JPEG_JPEG_PART=$3 IPX_JPEG_JPEG_PARALLEL=$4 timeout -k 10 120 python tools/bench_j2j.py ${N:-1024} 5 2>&1 | grep "images/s" | head -1 | sed 's/.*files in//')"; done
done
