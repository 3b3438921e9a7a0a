"""The C++ host side (include/g4s/*.hpp) over the C-ABI: compiles everywhere; on the GPU box the reference-shaped driver
(examples/spgemm_driver.cpp: mkl(A,B,C,timing) ×11, Timings table, SpMV, GraphProcess) runs and checks its own results."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build(out):
    lib = os.path.join(ROOT, "g4s_amd", "lib")
    cmd = ["g++", "-std=c++17", "-O2", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", "spgemm_driver.cpp"),
           "-L" + lib, "-lg4s_hip", "-Wl,-rpath," + lib, "-L/opt/rocm/lib", "-Wl,-rpath,/opt/rocm/lib", "-o", out]
    subprocess.check_call(cmd)


def test_cpp_host_headers_compile(tmp_path):
    _build(str(tmp_path / "spgemm_driver"))


@pytest.mark.gpu
def test_cpp_driver_runs_on_gpu(tmp_path):
    exe = str(tmp_path / "spgemm_driver")
    _build(exe)
    out = subprocess.run([exe, "20000"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "CHECK OK" in out.stdout and "spmm" in out.stdout and "GFLOPS" in out.stdout
