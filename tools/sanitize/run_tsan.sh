#!/bin/bash
# The host-side runtime logic that has no GPU in it -- the micro-batcher (csrc/ipx_batcher.cpp), the process-wide thread pool
# (csrc/ipx_threads.h) and the pool's queue / tickets / feeders (csrc/ipx_pool_core.h) -- under ThreadSanitizer on the CPU, against fake
# device work (tools/sanitize/batcher_host_test.cpp, tools/sanitize/pool_host_test.cpp).
set -e
cd "$(dirname "$0")/../.."
out=/tmp/ipx_sanitize
mkdir -p $out
g++ -std=c++17 -O1 -g -fsanitize=thread -fno-omit-frame-pointer -pthread -o $out/batcher_tsan tools/sanitize/batcher_host_test.cpp
TSAN_OPTIONS=halt_on_error=1:second_deadlock_stack=1 $out/batcher_tsan
g++ -std=c++17 -O1 -g -fsanitize=thread -fno-omit-frame-pointer -pthread -o $out/pool_tsan tools/sanitize/pool_host_test.cpp
TSAN_OPTIONS=halt_on_error=1:second_deadlock_stack=1 $out/pool_tsan
echo "no sanitizer report"
