#!/bin/bash
# The host-side runtime logic that has no GPU in it -- the micro-batcher (csrc/ipx_batcher.cpp) and the process-wide thread pool
# (csrc/ipx_threads.h) -- under ThreadSanitizer on the CPU, against a fake job backend (tools/sanitize/batcher_host_test.cpp).
set -e
cd "$(dirname "$0")/../.."
out=/tmp/ipx_sanitize
mkdir -p $out
g++ -std=c++17 -O1 -g -fsanitize=thread -fno-omit-frame-pointer -pthread -o $out/batcher_tsan tools/sanitize/batcher_host_test.cpp
TSAN_OPTIONS=halt_on_error=1:second_deadlock_stack=1 $out/batcher_tsan
echo "no sanitizer report"
