#!/usr/bin/env python3
"""Seeded stand-in for the reference's scr/data_scr/synthetic_generation.py:
writes data/<dir>/synthetic.csv-style rows  `label, x_1, ..., x_d`  (x_1 = 1 is
the bias) and the betas, from the same counter-based generator the device
fill uses (mhlik_fill_synthetic), so a CSV run and a device-generated run see
the same data.  The reference's script draws from numpy's unseeded global
generator; the distribution is the same (synthetic_generation.py:33-65).

    python tools/synthetic_csv.py -sz 10000 -dim 3 -seed 12345 -o data/host/synthetic.csv
"""
import argparse
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mcmc_amd import _lib  # noqa: E402
from mcmc_amd.likelihood import synthetic_host  # noqa: E402


def generate(n, d, seed):
    Xt = synthetic_host(seed, n, d)          # -y * x, column 0 is -y
    y = -Xt[:, 0]
    x = -Xt * y[:, None]
    return y, x


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("-sz", type=int, required=True)
    ap.add_argument("-dim", type=int, required=True)
    ap.add_argument("-seed", type=int, default=12345)
    ap.add_argument("-o", required=True)
    a = ap.parse_args()
    y, x = generate(a.sz, a.dim, a.seed)
    os.makedirs(os.path.dirname(os.path.abspath(a.o)), exist_ok=True)
    with open(a.o, "w") as f:
        for i in range(a.sz):
            f.write("%d," % int(y[i]) + ",".join(repr(float(v)) for v in x[i]) + "\n")
    beta = (C.c_int32 * a.dim)()
    _lib.load_mhlik().mhlik_synthetic_beta(a.seed, a.dim, beta)
    with open(os.path.join(os.path.dirname(os.path.abspath(a.o)), "synthetic_betas.csv"), "w") as f:
        f.write("\n".join(str(b) for b in beta) + "\n")


if __name__ == "__main__":
    main()
