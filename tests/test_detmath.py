"""mhlik_detmath.h: the deterministic float exp used on both sides of the
mixed-precision check stays within 1 ulp of libm expf (compiled here with
gcc; bit equality host vs gfx950 is checked by tools/check_detmath.hip)."""
import os
import subprocess
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SRC = r"""
#include "mhlik_detmath.h"
#include <stdio.h>
#include <stdlib.h>
int main(void){ int worst=0; long n=0;
  for (float x=-87.0f; x<88.7f; x+=0.00037f){ float a=mhlik_expf_det(x), b=expf(x); int32_t ia,ib;
    memcpy(&ia,&a,4); memcpy(&ib,&b,4); int u=abs(ia-ib); if(u>worst) worst=u; n++; }
  if (mhlik_expf_det(-200.0f)!=0.0f || !(mhlik_expf_det(100.0f)>3e38f) || mhlik_expf_det(0.0f)!=1.0f) return 2;
  if (mhlik_sigmoidf_det(0.0f)!=0.5f) return 3;
  printf("%d %ld\n", worst, n); return worst<=1 ? 0 : 1; }
"""


def test_expf_det_within_one_ulp_of_libm():
    with tempfile.TemporaryDirectory() as td:
        c = os.path.join(td, "t.c")
        open(c, "w").write(SRC)
        exe = os.path.join(td, "t")
        subprocess.check_call(["gcc", "-O2", "-ffp-contract=off", "-I" + os.path.join(ROOT, "mcmc_amd", "csrc"), c,
                               "-lm", "-o", exe])
        out = subprocess.run([exe], capture_output=True, text=True)
        assert out.returncode == 0, out.stdout + out.stderr
