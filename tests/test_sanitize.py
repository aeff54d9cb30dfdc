"""The two host-side parsers that read bytes from outside (JPEG markers of every upload, TrueType tables) under AddressSanitizer + UBSan
on the CPU (GPU sanitizers are not available): tools/sanitize/run.sh builds them with gcc and feeds them mutated headers and fonts."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("g++") is None or not os.path.isdir("/opt/rocm/include"), reason="needs g++ and the HIP headers")
def test_parsers_under_asan_ubsan():
    r = subprocess.run([os.path.join(ROOT, "tools", "sanitize", "run.sh"), "400"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "no sanitizer report" in r.stdout


@pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++")
def test_batcher_thread_pool_and_pool_core_under_tsan():
    """The host runtime logic that has no GPU in it, under ThreadSanitizer.  The micro-batcher's queue / timer / tickets
    (csrc/ipx_batcher.cpp) and the process-wide thread pool (csrc/ipx_threads.h) against a fake job backend: 8 submitters, 3200 files of
    three keys, scrambled waits and releases, releases without a wait, refused jobs, tickets nobody collects, destruction with work
    pending.  The pool's queue / tickets / feeders (csrc/ipx_pool_core.h, what ipx_pool.hip runs its GPU chunks on) with a memcpy as the
    device work: 8 submitters x 40 jobs over 3 slots x 2 feeders, waits out of order, polls, releases of running jobs, failing jobs,
    largest-first order, stop() with 20 jobs queued, submit after stop."""
    r = subprocess.run([os.path.join(ROOT, "tools", "sanitize", "run_tsan.sh")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "no sanitizer report" in r.stdout and "batcher ok" in r.stdout and "thread pool ok" in r.stdout
    assert r.stdout.count("pool core ok") == 2
