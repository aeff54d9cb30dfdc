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
