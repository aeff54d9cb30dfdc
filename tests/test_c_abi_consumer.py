"""tests/c_abi_consumer.c: the cgo call sequence of go/ipx replayed from plain C99 (gcc -std=c99 -pedantic -Werror, no C++),
compared with the oracle byte for byte.  Compiling and linking it needs no GPU; running it does."""
import os
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
EXE = os.path.join(HERE, "c_abi_consumer.bin")


def _build():
    from imageprocessor_amd import build
    import oracle
    build.build()
    oracle.build()
    cmd = ["gcc", "-std=c99", "-pedantic", "-Wall", "-Wextra", "-Werror", "-O1", "-I" + os.path.join(ROOT, "include"),
           "-I" + os.path.join(ROOT, "oracle"), "-o", EXE, os.path.join(HERE, "c_abi_consumer.c"),
           "-L" + os.path.join(ROOT, "imageprocessor_amd"), "-lipx", "-L" + os.path.join(ROOT, "oracle"), "-lipx_oracle",
           "-Wl,-rpath," + os.path.join(ROOT, "imageprocessor_amd"), "-Wl,-rpath," + os.path.join(ROOT, "oracle"),
           "-Wl,-rpath,/opt/rocm/lib", "-L/opt/rocm/lib", "-lamdhip64"]
    subprocess.check_call(cmd)


def test_c99_consumer_compiles_and_links():
    _build()
    assert os.path.exists(EXE)


@pytest.mark.gpu
def test_c99_consumer_runs_on_the_gpu():
    _build()
    r = subprocess.run([EXE], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "c_abi_consumer ok" in r.stdout
