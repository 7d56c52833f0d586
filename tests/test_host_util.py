"""Host utilities of the drivers (rows f1/f2): CSV ingest with label folding,
command-line parsing, output formats.  CPU only."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

from mcmc_amd import samplers as S  # noqa: E402
from mcmc_amd.likelihood import synthetic_host  # noqa: E402


def _write_csv(tmp_path, n, d, seed):
    out = os.path.join(tmp_path, "synthetic.csv")
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "synthetic_csv.py"), "-sz", str(n), "-dim",
                           str(d), "-seed", str(seed), "-o", out])
    return out


def test_read_data_folds_label_and_matches_generator(tmp_path):
    n, d = 777, 5
    path = _write_csv(str(tmp_path), n, d, 99)
    lib = S.load_samplers()
    lib.read_data.argtypes = [C.c_char_p, C.c_int, S.data_str, S.mcmc_str]
    lib.read_data_sp.argtypes = [C.c_char_p, C.c_int, S.data_str, S.mcmc_str]
    want = synthetic_host(99, n, d)
    col = np.zeros((n, d), order="F")
    mcin = S.mcmc_str(ddata=d, Nd=n, impl=S.IMPL_GPU)
    lib.read_data(path.encode(), 1, S.data_str(data=col.ctypes.data), mcin)  # ColMajor
    assert np.array_equal(col, want)
    row = np.zeros((n, d), order="C")
    lib.read_data(path.encode(), 0, S.data_str(data=row.ctypes.data), mcin)  # RowMajor
    assert np.array_equal(row, want)
    f32 = np.zeros((n, d), order="F", dtype=np.float32)
    lib.read_data_sp(path.encode(), 1, S.data_str(dataf=f32.ctypes.data), mcin)
    assert np.array_equal(f32, want.astype(np.float32))
    # impl == MP fills both copies; fewer rows than the file holds is fine (reads the first Nd)
    both_d = np.zeros((100, d), order="F")
    both_f = np.zeros((100, d), order="F", dtype=np.float32)
    lib.read_data(path.encode(), 1, S.data_str(data=both_d.ctypes.data, dataf=both_f.ctypes.data),
                  S.mcmc_str(ddata=d, Nd=100, impl=S.IMPL_MP))
    assert np.array_equal(both_d, want[:100]) and np.array_equal(both_f, want[:100].astype(np.float32))


def test_cli_rejects_unknown_flag_and_missing_argument():
    exe = os.path.join(ROOT, "mcmc_amd", "bin", "mcmc_gpu")
    r = subprocess.run([exe, "-bogus", "1"], capture_output=True, text=True)
    assert r.returncode == 1 and "Unknown option" in r.stderr
    r = subprocess.run([exe, "-sz"], capture_output=True, text=True)
    assert r.returncode == 1 and "Missing argument to -sz." in r.stderr
    r = subprocess.run([exe, "-tune", "5"], capture_output=True, text=True)
    assert r.returncode == 1 and "valid value" in r.stderr


def test_write_data_format(tmp_path):
    lib = S.load_samplers()
    lib.write_data.argtypes = [C.c_char_p, C.c_void_p, C.c_int, C.c_int]
    a = np.array([[1.5, -2.25], [0.1, 3.0]])
    stem = os.path.join(str(tmp_path), "raw_out")
    lib.write_data(stem.encode(), a.ctypes.data, 2, 2)
    assert open(stem + ".csv").read() == "1.500000, -2.250000\n0.100000, 3.000000\n"
