"""The C-ABI libraries load and export every symbol include/*.h declares; the
by-value structs have the reference's layout; the product fails loudly
without a GPU.  No compute calls here (no GPU in this container)."""
import ctypes as C
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared(header, pattern):
    txt = open(os.path.join(ROOT, "include", header)).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(pattern, txt)))


def test_mhlik_exports_every_declared_symbol():
    from mcmc_amd import _lib

    lib = _lib.load_mhlik()
    names = _declared("mhlik.h", r"\b(mhlik_[a-z0-9_]+)\s*\(")
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), "libmhlik.so lacks %s" % n
    assert set(names) == set(_lib.EXPORTED_SYMBOLS), set(names) ^ set(_lib.EXPORTED_SYMBOLS)
    assert b"gfx950" in lib.mhlik_version()


def test_sampler_library_exports_every_declared_symbol():
    from mcmc_amd import samplers

    S = samplers.load_samplers()
    names = _declared("mcmc_samplers.h", r"\b([A-Za-z_][A-Za-z0-9_]*)\s*\(")
    names = [n for n in names if n not in ("defined", "sizeof")]
    for n in names:
        assert hasattr(S, n), "libmcmc_amd.so lacks %s" % n
    assert set(samplers.EXPORTED_SYMBOLS) <= set(names)


def test_no_cuda_or_torch_types_in_the_abi():
    for h in ("mhlik.h", "mcmc_samplers.h", "mcmc_types.h"):
        txt = open(os.path.join(ROOT, "include", h)).read()
        code = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
        for bad in ("cublas", "curand", "cudaStream", "hipStream", "torch", "at::", "dev_v_str"):
            assert bad not in code, (h, bad)


def test_struct_layouts_match_reference_header():
    """Compile a probe against include/mcmc_types.h and compare sizes/offsets
    with the ctypes mirrors (which follow src/util/resources.h field by field)."""
    from mcmc_amd import samplers as S

    probe = r"""
    #include "mcmc_types.h"
    #include <stdio.h>
    int main(void){
      printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu\n", sizeof(data_str), sizeof(mcmc_v_str), sizeof(mcmc_str),
             sizeof(mp_str), sizeof(mcmc_tune_str), sizeof(mcmc_int_v), sizeof(mcmc_int), sizeof(gpu_v_str), sizeof(out_str));
      printf("%zu %zu %zu %zu\n", offsetof(out_str, acceptance), offsetof(mcmc_tune_str, rwsdf),
             offsetof(mcmc_int, cposteriorf), offsetof(mp_str, ratio));
      return 0; }
    """
    import tempfile

    with tempfile.TemporaryDirectory() as td:
        c = os.path.join(td, "p.c")
        open(c, "w").write(probe)
        exe = os.path.join(td, "p")
        subprocess.check_call(["gcc", "-I" + os.path.join(ROOT, "include"),
                               "-I" + os.path.join(ROOT, "mcmc_amd", "csrc", "gslcompat"), c, "-o", exe])
        out = subprocess.check_output([exe], text=True).split()
    sizes = [int(x) for x in out[:9]]
    mirrors = [S.data_str, S.mcmc_v_str, S.mcmc_str, S.mp_str, S.mcmc_tune_str, S.mcmc_int_v, S.mcmc_int,
               S.gpu_v_str, S.out_str]
    assert sizes == [C.sizeof(m) for m in mirrors]
    offs = [int(x) for x in out[9:]]
    assert offs == [S.out_str.acceptance.offset, S.mcmc_tune_str.rwsdf.offset, S.mcmc_int.cposteriorf.offset,
                    S.mp_str.ratio.offset]


@pytest.mark.skipif(os.path.exists("/dev/kfd"), reason="only meaningful without a GPU")
def test_fails_loudly_without_gpu():
    from mcmc_amd import Likelihood, MhlikError

    with pytest.raises(MhlikError) as e:
        Likelihood(100, 3)
    assert "no CPU fallback" in str(e.value)


def test_product_never_references_the_oracle():
    """Nothing under mcmc_amd/ or include/ may import, link or name oracle/."""
    bad = []
    for base in ("mcmc_amd", "include"):
        for dp, _, fs in os.walk(os.path.join(ROOT, base)):
            for f in fs:
                if f.endswith((".so", ".pyc")):
                    continue
                txt = open(os.path.join(dp, f), errors="ignore").read()
                if re.search(r"oracle_py|liboracle|oracle/|import oracle|#include \"oracle", txt):
                    # comments that explain the separation are fine; code references are not
                    for line in txt.splitlines():
                        if re.search(r"oracle_py|liboracle|import oracle|#include \"oracle", line):
                            bad.append((f, line.strip()))
    assert not bad, bad


def test_config_validation_without_gpu():
    from mcmc_amd import _lib

    lib = _lib.load_mhlik()
    ctx = C.c_void_p()
    cfg = _lib.Config()
    cfg.n_rows, cfg.n_cols, cfg.precision, cfg.world, cfg.segments = 10, 0, 1, 1, 0
    assert lib.mhlik_create(C.byref(cfg), C.byref(ctx)) == 1  # MHLIK_EINVAL before any device call
    cfg.n_cols, cfg.world, cfg.segments = 3, 3, 8
    assert lib.mhlik_create(C.byref(cfg), C.byref(ctx)) == 1  # segments not a multiple of world
    assert b"multiple of world" in lib.mhlik_last_error(None)
    assert lib.mhlik_synthetic_beta(1, 0, None) == 1
