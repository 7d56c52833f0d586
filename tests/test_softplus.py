"""mhlik_softplus.h: -log(1 + e^x) for the K-chain double kernel's epilogue.
Host build (gcc) against an 80-bit long double evaluation; the gfx950 build
is checked by tools/check_softplus.hip (tests/test_gpu_batched.py)."""
import os
import subprocess
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SRC = r"""
#include "mhlik_softplus.h"
#include <stdio.h>
int main(void){
  for (long i = 0; i <= 400000; i++) {
    double x = -750.0 + 1500.0 * (double)i / 400000.0;
    if (i % 7 == 0) x = -40.0 + 80.0 * (double)i / 400000.0;   /* denser where both terms matter */
    double f = mhlik_neg_softplus(x);
    fwrite(&x, 8, 1, stdout); fwrite(&f, 8, 1, stdout);
  }
  return 0; }
"""


def test_neg_softplus_host_accuracy():
    with tempfile.TemporaryDirectory() as td:
        c = os.path.join(td, "t.c")
        open(c, "w").write(SRC)
        exe = os.path.join(td, "t")
        subprocess.check_call(["gcc", "-O2", "-ffp-contract=off", "-I" + os.path.join(ROOT, "mcmc_amd", "csrc"), c,
                               "-lm", "-o", exe])
        raw = subprocess.run([exe], capture_output=True).stdout
    a = np.frombuffer(raw, dtype=np.float64).reshape(-1, 2)
    x, f = a[:, 0], a[:, 1]
    over = x > 709.782712893384
    assert np.all(np.isneginf(f[over])), "the reference formula overflows to -inf above 709.78"
    xl = x[~over].astype(np.longdouble)
    want = -(np.maximum(xl, 0) + np.log1p(np.exp(-np.abs(xl))))
    got = f[~over].astype(np.longdouble)
    normal = xl > -708  # below that the result is a denormal: absolute, not relative, accuracy
    err = np.abs(got[normal] - want[normal]) / np.abs(want[normal])
    assert float(err.max()) < 6e-16, float(err.max())
    assert float(np.abs(got[~normal] - want[~normal]).max()) < 1e-307
    # against libm's own exp/log the way the reference writes it
    ref = -np.log(1.0 + np.exp(x[~over]))
    ok = np.abs(ref) > 1e-3   # below that 1 + e^x has lost the bits, the reference's formula is the inexact one
    assert np.max(np.abs(f[~over][ok] - ref[ok]) / np.abs(ref[ok])) < 5e-13
    assert np.all(f[x < -746] == 0.0)
