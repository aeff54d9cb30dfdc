#!/bin/bash
# Which copy engine the runtime picked for every copy of one run (AMD_LOG_LEVEL=4 lines of rocblit.cpp), as a run-length summary.
# usage: tools/engine_log.sh OUT.txt -- command ...
out=$1; shift; shift
AMD_LOG_LEVEL=4 "$@" 2> "$out.raw" || exit 1
grep -E "HSA Copy|Query copy engine" "$out.raw" | sed -E 's/.*(Query copy engine status [0-9a-fx]+).*(free_engine_mask [0-9a-fx]+), (rec_engine_mask [0-9a-fx]+)/Q \2/; s/.*HSA Copy (copy_engine=[0-9a-fx]+).*size=([0-9]+).*engineType=([0-9]).*/C \1 type=\3/; s/.*HSA Copy dst.*engineType=([0-9]).*/C regular-api type=\1/' | cut -c1-60 | uniq -c > "$out"
rm -f "$out.raw"
