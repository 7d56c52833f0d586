"""bin/mcmc_gpu and bin/mcmc_cpu (the CPU path's driver, run on the GPU) end to end: CSV in, chain CSV out, vs the oracle chain
(src/mcmc_gpu/gpu_host.c flow; the RNG seed comes from GSL_RNG_SEED like the
reference's scr/sh_scr/multiple_runs_gpu.sh:5)."""
import os
import subprocess
import sys

import numpy as np
import pytest

import oracle_py as O

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("binary,marker", [("mcmc_gpu", "Reduction Kernel"), ("mcmc_cpu", "Acceptance Ratio")])
def test_mcmc_gpu_binary_reproduces_oracle_chain(tmp_path, binary, marker):
    from mcmc_amd.likelihood import synthetic_host

    n, d, Ns, burn = 5000, 3, 200, 50
    csv = os.path.join(str(tmp_path), "synthetic.csv")
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "synthetic_csv.py"), "-sz", str(n), "-dim",
                           str(d), "-seed", "12345", "-o", csv])
    out = os.path.join(str(tmp_path), "gpu_")
    env = dict(os.environ, GSL_RNG_SEED="7")
    r = subprocess.run([os.path.join(ROOT, "mcmc_amd", "bin", binary), "-d", "1", "-sz", str(n), "-dim", str(d),
                        "-samp", str(Ns), "-burn", str(burn), "-lag", "20", "-tune", "0", "-in", csv, "-out", out],
                       capture_output=True, text=True, env=env, timeout=120)
    assert r.returncode == 0, r.stderr
    assert "Acceptance Ratio" in r.stdout and marker in r.stdout
    got = np.loadtxt(out + "raw_out.csv", delimiter=",")
    X = synthetic_host(12345, n, d)
    ref = O.cpu_sampler(X, 7, Ns, burn, 2.38 / np.sqrt(d))
    want = np.array([[float("%f" % v) for v in row] for row in ref["samples"]])
    assert got.shape == want.shape and np.array_equal(got, want)
    for f in ("burn_out", "norm_raw_out", "norm_burn_out", "mean_out", "circular_autocorrelation"):
        assert os.path.exists(out + f + ".csv")
