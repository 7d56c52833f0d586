"""The host C code under the thread and address sanitizers, on the CPU.

samplers/multichain.c runs a helper thread (it draws chunk i+1's random numbers and files chunk i-1's rows while the
device runs chunk i) and keeps its page-locked chunk buffers in a process-wide cache behind a mutex.  Here that pipeline
is built with -fsanitize=thread and with -fsanitize=address,undefined against a TEST DOUBLE of the mhlik_* entry points
(tests/doubles/mhlik_double.c: it copies the buffers through and accepts by a fixed rule; it is never linked into the
product) and driven from two threads at once; the driver also replays every chain from its generator and compares every
row.  The whole CPU suite under ASan + UBSan (libmcmc_amd.so and liboracle.so rebuilt with the sanitizers, Python
started with libasan preloaded) is recorded in DESIGN.md section 5; it takes a minute and is not repeated here."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "mcmc_amd", "csrc")
SOURCES = [os.path.join(ROOT, "tests", "doubles", "kc_pipeline_main.c"), os.path.join(ROOT, "tests", "doubles", "mhlik_double.c"),
           os.path.join(CSRC, "samplers", "multichain.c"), os.path.join(CSRC, "samplers", "mh_core.c"),
           os.path.join(CSRC, "gslcompat", "gslcompat.c")]


@pytest.mark.parametrize("sanitizer", ["thread", "address,undefined"])
def test_kchain_chunk_pipeline_under_sanitizer(tmp_path, sanitizer):
    exe = str(tmp_path / "kc_pipeline")
    subprocess.check_call(["gcc", "-fsanitize=" + sanitizer, "-fno-sanitize-recover=all", "-fno-omit-frame-pointer", "-O1", "-g",
                           "-std=gnu11", "-ffp-contract=off", "-pthread", "-I" + os.path.join(ROOT, "include"),
                           "-I" + os.path.join(CSRC, "gslcompat"), "-I" + os.path.join(CSRC, "samplers")] + SOURCES +
                          ["-lm", "-o", exe])
    env = dict(os.environ, TSAN_OPTIONS="halt_on_error=1 exitcode=66", ASAN_OPTIONS="detect_leaks=1 exitcode=66",
               UBSAN_OPTIONS="halt_on_error=1")
    env.pop("MCMC_AMD_HOST_LOOP", None)
    out = subprocess.run([exe], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    assert "kc pipeline ok" in out.stdout
    assert "WARNING: ThreadSanitizer" not in out.stderr and "ERROR: AddressSanitizer" not in out.stderr
