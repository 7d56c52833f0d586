"""The split-bf16 K-chain float pass exists twice in source: the production shape alone in
mcmc_amd/csrc/mhlik_batched_x3.hpp (what libmhlik.so is built from) and, with the experiment switches of the
ablation tools, in tools/x3_lab_variants.hpp.  So that an edit of one cannot drift away from the other unnoticed --
and so that the "production" row of profiles/r0*_x3_ablate.txt is the product -- both are compiled for gfx950 here
and the lab kernel with every switch off must be, instruction for instruction, the product kernel.  CPU test: hipcc
cross-compiles without a GPU (about three seconds)."""
import difflib
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = "/opt/rocm/bin/hipcc"

SRC = """
#include "x3_lab_variants.hpp"
using namespace mhlik;
#define ARGS const float *, TileGeom, const int32_t *, const u32x4 *, int, double *, int64_t, int, int64_t
template __global__ void mhlik::lik_batched_x3_kernel<4, 6, 2>(ARGS);
template __global__ void mhlik::lab_lik_batched_x3_kernel<4, 6, 0, 2>(ARGS);
template __global__ void mhlik::lik_batched_x3_kernel<1, 6, 2>(ARGS);
template __global__ void mhlik::lab_lik_batched_x3_kernel<1, 6, 0, 2>(ARGS);
"""


def _bodies(asm):
    out = {}
    for m in re.finditer(r"^(_ZN5mhlik\d+(?:lab_)?lik_batched_x3_kernel\w+):[^\n]*\n(.*?)\n\s*s_endpgm", asm, re.S | re.M):
        lines = [re.sub(r"\.LBB\d+_", ".LBB_", ln.split(";")[0].rstrip()) for ln in m.group(2).splitlines()]
        out[m.group(1)] = [ln for ln in lines if ln.strip() and not ln.strip().startswith(".")]
    return out


def test_lab_kernel_with_all_switches_off_is_the_product_kernel(tmp_path):
    src = tmp_path / "both.hip"
    src.write_text(SRC)
    asm = tmp_path / "both.s"
    subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-ffp-contract=off", "-O3", "-std=c++17",
                           "-I" + os.path.join(ROOT, "mcmc_amd", "csrc"), "-I" + os.path.join(ROOT, "include"),
                           "-I" + os.path.join(ROOT, "tools"), "-S", "--cuda-device-only", str(src), "-o", str(asm)],
                          stderr=subprocess.DEVNULL)
    bodies = _bodies(asm.read_text())
    assert len(bodies) == 4, list(bodies)
    for nb in ("Li4ELi6E", "Li1ELi6E"):
        prod = [k for k in bodies if "lab_" not in k and nb in k]
        lab = [k for k in bodies if "lab_" in k and nb in k]
        assert len(prod) == 1 and len(lab) == 1
        a, b = bodies[prod[0]], bodies[lab[0]]
        assert len(a) > 1000
        diff = list(difflib.unified_diff(a, b, lineterm="", n=0))
        assert not diff, "\n".join(diff[:40])


def test_product_header_has_no_lab_switches():
    for name in ("mhlik_batched_x3.hpp", "mhlik_batched.hpp", "mhlik_mp_batched.hpp", "mhlik_kernels.hpp", "mhlik_chain.hpp"):
        text = open(os.path.join(ROOT, "mcmc_amd", "csrc", name)).read()
        assert "EXP" not in text and "lab:" not in text and "Lab variant" not in text and "dbg" not in text, name
