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
    assert "CHECK OK" in out.stdout and "spmm" in out.stdout and "perf(Gflops):" in out.stdout


def _build_example(src, out):
    lib = os.path.join(ROOT, "g4s_amd", "lib")
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", src),
                           "-L" + lib, "-lg4s_hip", "-Wl,-rpath," + lib, "-L/opt/rocm/lib", "-Wl,-rpath,/opt/rocm/lib", "-o", out])


MTX_FILES = {
    "general.mtx": "%%MatrixMarket matrix coordinate real general\n% comment\n3 4 5\n3 1 1.5\n1 2 -2\n1 1 4\n2 4 7e-1\n1 1 0.25\n",
    "pattern.mtx": "%%MatrixMarket matrix coordinate pattern general\n2 2 2\n2 2\n1 1\n",
    "symmetric.mtx": "%%MatrixMarket matrix coordinate real symmetric\n4 4 7\n1 1 2\n2 1 -1\n2 2 2\n3 2 -1\n3 3 2\n4 3 -1\n4 4 2\n",
    "skew.mtx": "%%MatrixMarket matrix coordinate real skew-symmetric\n3 3 2\n2 1 5\n3 2 -3\n",
    "complex.mtx": "%%MatrixMarket matrix coordinate complex general\n2 2 2\n1 1 3.0 9.0\n2 1 -1 0\n",
    "integer.mtx": "%%MatrixMarket matrix coordinate integer general\n2 3 3\n1 3 7\n2 2 -4\n1 1 2\n",
}


def test_cpp_mtx_reader_matches_oracle(tmp_path, oracle):
    """g4s::read_matrix_market (include/g4s/mtx.hpp) against the oracle's restatement of CSR::construct (CSR.h:485-669)."""
    import numpy as np
    exe = str(tmp_path / "mtx_dump")
    _build_example("mtx_dump.cpp", exe)
    for name, text in MTX_FILES.items():
        f = tmp_path / name
        f.write_text(text)
        out = subprocess.run([exe, str(f)], capture_output=True, text=True, check=True).stdout.split("\n")
        rows, cols, nnz = map(int, out[0].split())
        rp = np.array([int(v) for v in out[1:rows + 2]])
        ent = [l.split() for l in out[rows + 2:rows + 2 + nnz]]
        ci, va = np.array([int(c) for c, _ in ent]), np.array([float(v) for _, v in ent])
        r, c, orp, oci, ova = oracle.mtx_read(str(f))
        assert (rows, cols) == (r, c) and np.array_equal(rp, orp) and np.array_equal(ci, oci) and np.array_equal(va, ova), name
    for bad in ["%%MatrixMarket matrix array real general\n1 1\n1\n", "%%MatrixMarket matrix coordinate real hermitian\n1 1 1\n1 1 1\n", "garbage\n",
                "%%MatrixMarket matrix coordinate real general\n2 2 3\n1 1 1\n"]:
        f = tmp_path / "bad.mtx"
        f.write_text(bad)
        assert subprocess.run([exe, str(f)], capture_output=True).returncode == 1


def _reference_timings_layout(ms, total_flop):
    """The bytes Timings::print and reg_print put on stdout (mm/src/Timings.cpp:36-65) for stage times in seconds t = ms/1000."""
    c, s, v, o, e, d, t = (x / 1000 for x in ms)
    G = total_flop / 1000000000
    sum_total = c + s + v + o + e + d
    lines = ["total flop %f" % total_flop, "time(ms):"]
    for name, x in (("create", c), ("spmm", s), ("convert", v), ("order", o), ("export_csr", e), ("destroy", d), ("sum_total", sum_total)):
        lines.append("    %-18s %8.3fms %6.2f%%" % (name, 1000 * x, x / t * 100))
    lines.append("perf(Gflops):")
    for name, x in (("create", c), ("spmm", s), ("convert", v), ("order", o), ("export_csr", e), ("destroy", d), ("total", t)):
        lines.append("    %-18s %6.2f" % (name, G / x))
    lines.append("%e" % (G / t))
    return "\n".join(lines) + "\n"


def test_timings_print_has_the_reference_layout(tmp_path):
    """a12 / f3: g4s::Timings::print writes what mm/src/Timings.cpp:36-60 writes — `total flop`, the `time(ms):` block with percentages of
    `total` and a `sum_total` line, the `perf(Gflops):` block — and reg_print (:62-65) one %le line."""
    exe = str(tmp_path / "timings_print")
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", "timings_print.cpp"), "-o", exe])
    for ms, flop in (((1.25, 80.5, 0.0625, 3.5, 0.75, 2.0, 90.0), 2 * 2.49e9), ((0.001, 12345.678, 5.0, 0.5, 100.0, 0.25, 13000.0), 123456789012.0)):
        out = subprocess.run([exe] + [repr(x) for x in ms] + [repr(flop)], capture_output=True, text=True, check=True).stdout
        assert out == _reference_timings_layout(ms, flop)
    # the fixed-width fields of the reference, literally
    assert "    create                1.250ms   1.39%\n" in _reference_timings_layout((1.25, 80.5, 0.0625, 3.5, 0.75, 2.0, 90.0), 1e9)


@pytest.mark.gpu
def test_cpp_mkl_spgemm_cli(tmp_path, oracle):
    """The reference benchmark's command line (mm/src/mkl_spgemm.cpp) on the device library: C = A·B from .mtx files, stage table."""
    import numpy as np
    import scipy.sparse as sp
    exe = str(tmp_path / "mkl_spgemm_g4s")
    _build_example("mkl_spgemm_g4s.cpp", exe)
    rng = np.random.default_rng(3)
    A = sp.random(60, 50, density=0.1, random_state=rng, format="coo")
    B = sp.random(45, 70, density=0.1, random_state=rng, format="coo")      # inner dimensions differ: the driver cuts both to 45
    for name, M in (("a.mtx", A), ("b.mtx", B)):
        with open(tmp_path / name, "w") as f:
            f.write(f"%%MatrixMarket matrix coordinate real general\n{M.shape[0]} {M.shape[1]} {M.nnz}\n")
            for i, j, v in zip(M.row, M.col, M.data):
                f.write(f"{int(i) + 1} {int(j) + 1} {float(v)!r}\n")
    # argv as the reference's: mat1 mat2 threads (ignored, mm/src/mkl_spgemm.cpp:61), then this driver's --dump extension
    out = subprocess.run([exe, str(tmp_path / "a.mtx"), str(tmp_path / "b.mtx"), "80", "--dump"], capture_output=True, text=True, timeout=300,
                         env=dict(os.environ, G4S_BENCH_ITERS="2"))
    assert out.returncode == 0, out.stderr
    head = out.stdout.split("\n")
    assert head[0] == f"从文件 {tmp_path / 'a.mtx'} 读取矩阵A:"                      # mkl_spgemm.cpp:38
    flop = oracle.flop(*[np.asarray(a) for a in (A.tocsr()[:, :45].indptr.astype(np.int32), A.tocsr()[:, :45].indices.astype(np.int32),
                                                   B.tocsr()[:45, :].indptr.astype(np.int32))])
    assert head[1] == "total flop %f" % (2 * flop) and head[2] == "time(ms):" and head[10] == "perf(Gflops):"
    assert [l.split()[0] for l in head[3:10]] == ["create", "spmm", "convert", "order", "export_csr", "destroy", "sum_total"]
    assert [l.split()[0] for l in head[11:18]] == ["create", "spmm", "convert", "order", "export_csr", "destroy", "total"]
    # name routing of :18-37: a bare name goes to <dir>/suite_sparse/<name>/<name>.mtx, *G500* to <dir>/G500/<name>.mtx
    os.makedirs(tmp_path / "matrix" / "suite_sparse" / "tiny", exist_ok=True)
    os.makedirs(tmp_path / "matrix" / "G500", exist_ok=True)
    os.replace(tmp_path / "a.mtx", tmp_path / "matrix" / "suite_sparse" / "tiny" / "tiny.mtx")
    os.replace(tmp_path / "b.mtx", tmp_path / "matrix" / "G500" / "G500_x.mtx")
    out2 = subprocess.run([exe, "tiny", "G500_x", "--dump"], capture_output=True, text=True, timeout=300, cwd=tmp_path,
                          env=dict(os.environ, G4S_BENCH_ITERS="1", G4S_MATRIX_DIR=str(tmp_path / "matrix")))
    assert out2.returncode == 0, out2.stderr
    assert [l for l in out2.stdout.splitlines() if l.startswith("C ")] == [l for l in out.stdout.splitlines() if l.startswith("C ")]
    assert subprocess.run([exe, "no_such_matrix"], capture_output=True, cwd=tmp_path).returncode == 1
    want = (A.tocsr()[:, :45] @ B.tocsr()[:45, :]).toarray()
    got = np.zeros_like(want)
    for line in out.stdout.splitlines():
        if line.startswith("C "):
            _, r, c, v = line.split()
            got[int(r), int(c)] = float(v)
    assert np.allclose(got, want, rtol=1e-12, atol=1e-14)


def test_dense_compare_driver_compiles(tmp_path):
    _build_example("dense_compare_g4s.cpp", str(tmp_path / "dense_compare_g4s"))


@pytest.mark.gpu
def test_dense_compare_driver_runs_on_gpu(tmp_path):
    """a14: the reference's dense comparison programs (mm/src/cblas_dxxmm.c, mv/mv.c) against the device library, one line per routine."""
    exe = str(tmp_path / "dense_compare_g4s")
    _build_example("dense_compare_g4s.cpp", exe)
    out = subprocess.run([exe, "700"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "CHECK OK" in out.stdout, out.stdout + out.stderr
    for name in ("cblas_dsymm", "cblas_dtrmm", "cblas_dgemm", "matrix_multiply_dsymv", "matrix_multiply_dtrmv", "matrix_multiply_sspmv", "matrix_multiply_dgemv"):
        assert name + " 运行时间：" in out.stdout


def _build_c_example(out):
    lib = os.path.join(ROOT, "g4s_amd", "lib")
    subprocess.check_call(["gcc", "-std=c99", "-O2", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", "citcoms_like.c"),
                           "-L" + lib, "-lg4s_hip", "-Wl,-rpath," + lib, "-L/opt/rocm/lib", "-Wl,-rpath,/opt/rocm/lib", "-lm", "-o", out])


def test_c_linkage_of_spmm_dense(tmp_path):
    """A plain C99 translation unit declares the reference's `extern void spmm_dense(...)` prototype and links (citcoms/bin/Citcom.c:45-48)."""
    _build_c_example(str(tmp_path / "citcoms_like"))


@pytest.mark.gpu
def test_citcoms_shaped_c_caller_runs_on_gpu(tmp_path):
    exe = str(tmp_path / "citcoms_like")
    _build_c_example(exe)
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "CHECK OK" in out.stdout, out.stdout + out.stderr
