"""The C++ host layer (include/GpuLinearSystem.h: the reference's LinearSystem /
SolverAlgorithm / EquationSystem surface over the C ABI) is tested by a C++ program written
like the reference's own unit tests (tests/cpp/test_host.cpp, cf. unit_tests/UnitTestTpetra.C);
it is built by __graft_entry__.build() (`make -C tests/cpp`) and run from here."""
import os
import shutil
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
BIN = os.path.join(HERE, "cpp", "test_host.bin")


def _run(args, timeout):
    assert os.path.exists(BIN), "tests/cpp/test_host.bin is missing: run __graft_entry__.build()"
    r = subprocess.run([BIN] + args, capture_output=True, text=True, timeout=timeout)
    print(r.stdout)
    print(r.stderr)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    return r.stdout


def test_header_is_plain_cxx11():
    """the host layer must compile inside the reference (C++11, no HIP, no torch)"""
    gxx = shutil.which("g++")
    if gxx is None:
        pytest.skip("no g++")
    subprocess.check_call([gxx, "-std=c++11", "-Wall", "-Wextra", "-Werror", "-fsyntax-only",
                           "-I" + os.path.join(ROOT, "include"), "-x", "c++",
                           os.path.join(ROOT, "include", "GpuLinearSystem.h")])


def test_cpp_host_layer_host_only():
    out = _run(["--host-only"], 60)
    assert "0 failures" in out


@pytest.mark.gpu
def test_cpp_host_layer_against_oracle(tmp_path):
    out = _run(["--tmp", str(tmp_path)], 300)
    assert "0 failures" in out and "10 tests" in out
