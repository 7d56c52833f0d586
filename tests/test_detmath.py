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


BOTH = r"""
#include "mhlik_detmath.h"   /* the product's statement (host + device) */
#include "mhlik_synth.h"
#include "mp_math.h"         /* the oracle's own statement (oracle/mp_math.h) */
#include <stdio.h>
#include <stdlib.h>
int main(void){
  /* every 61st float bit pattern over the whole range (70M arguments, both signs, denormals, inf, NaN) */
  long n=0; int worst=0;
  for (uint64_t b=0; b<=0xffffffffull; b+=61){ uint32_t u=(uint32_t)b; float x; memcpy(&x,&u,4);
    float p=mhlik_expf_det(x), o=oracle_expf_det(x); uint32_t ip,io; memcpy(&ip,&p,4); memcpy(&io,&o,4);
    if (ip!=io && !(p!=p && o!=o)) { printf("expf differs at %a: %a vs %a\n", x, p, o); return 1; }
    float sp=mhlik_sigmoidf_det(x), so=oracle_sigmoidf_det(x); memcpy(&ip,&sp,4); memcpy(&io,&so,4);
    if (ip!=io && !(sp!=sp && so!=so)) { printf("sigmoid differs at %a\n", x); return 1; }
    if (x>-87.0f && x<88.7f){ float l=expf(x); int32_t a,c; memcpy(&a,&o,4); memcpy(&c,&l,4); int d=abs(a-c); if(d>worst) worst=d; }
    n++; }
  if (worst>1) { printf("oracle expf is %d ulp from libm\n", worst); return 2; }
  for (uint64_t seed=0; seed<40; seed++) for (uint64_t call=0; call<50; call++) for (uint64_t blk=0; blk<500; blk+=7){
    uint64_t s2=seed*0x9E3779B97F4A7C15ull+seed, c2=call*call*1000003ull, b2=blk*blk*blk;
    if (mhlik_mp_uniform(s2,c2,b2)!=oracle_mp_uniform(s2,c2,b2)) { printf("uniform differs\n"); return 3; }
    double u=oracle_mp_uniform(s2,c2,b2); if(!(u>=0.0 && u<1.0)) return 4; }
  printf("%ld arguments, oracle vs libm worst %d ulp\n", n, worst); return 0; }
"""


def test_oracle_and_product_statements_agree_bit_for_bit():
    """oracle/mp_math.h (the oracle's own float sigmoid and per-block uniform) and the product's mhlik_detmath.h /
    mhlik_synth.h are two separately written statements of one definition: same bits for every argument tried, and
    the oracle's expf within 1 ulp of libm."""
    with tempfile.TemporaryDirectory() as td:
        c = os.path.join(td, "both.c")
        open(c, "w").write(BOTH)
        exe = os.path.join(td, "both")
        subprocess.check_call(["gcc", "-O2", "-ffp-contract=off", "-I" + os.path.join(ROOT, "mcmc_amd", "csrc"),
                               "-I" + os.path.join(ROOT, "oracle"), c, "-lm", "-o", exe])
        out = subprocess.run([exe], capture_output=True, text=True)
        assert out.returncode == 0, out.stdout + out.stderr


def test_oracle_mp_path_uses_no_product_header():
    src = open(os.path.join(ROOT, "oracle", "mp_path.c")).read() + open(os.path.join(ROOT, "oracle", "mp_math.h")).read()
    includes = [ln for ln in src.splitlines() if ln.lstrip().startswith("#include")]
    assert not [ln for ln in includes if "mhlik" in ln], includes


def test_expf_det_within_one_ulp_of_libm():
    with tempfile.TemporaryDirectory() as td:
        c = os.path.join(td, "t.c")
        open(c, "w").write(SRC)
        exe = os.path.join(td, "t")
        subprocess.check_call(["gcc", "-O2", "-ffp-contract=off", "-I" + os.path.join(ROOT, "mcmc_amd", "csrc"), c,
                               "-lm", "-o", exe])
        out = subprocess.run([exe], capture_output=True, text=True)
        assert out.returncode == 0, out.stdout + out.stderr
