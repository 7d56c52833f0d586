"""ctypes access to oracle/liboracle.so -- the CPU restatement of the
reference (TEST INFRASTRUCTURE: only tests/, smoke() and bench.py's
cpu_baseline leg import this)."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB = os.path.join(ORACLE_DIR, "liboracle.so")


class data_str(C.Structure):
    _fields_ = [("data", C.c_void_p), ("dataf", C.c_void_p), ("labels", C.c_void_p), ("mvout", C.c_void_p),
                ("mvoutf", C.c_void_p), ("zlabels", C.c_void_p), ("zidx", C.c_void_p)]


class mcmc_v_str(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("samples", "nsamples", "burn", "nburn", "sample_means", "samplesf",
                                          "nsamplesf", "burnf", "nburnf", "sample_meansf")]


class mcmc_str(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("ddata", "Nd", "Ndmap", "Ns", "burnin", "tune", "impl")]


class mp_str(C.Structure):
    _fields_ = [("dataBlocks", C.c_int), ("dataBlockSz", C.c_int), ("brightBlocks", C.c_int),
                ("darkBlocks", C.c_int), ("ResampFactor", C.c_int), ("ratio", C.c_float)]


class mcmc_tune_str(C.Structure):
    _fields_ = [("rwsd", C.c_double), ("rwsdf", C.c_float)]


class gpu_v_str(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("size", "threads", "blocks", "maxThreads", "maxBlocks", "kernel", "cpuThresh")]


class out_str(C.Structure):
    _fields_ = [(n, C.c_float) for n in ("cuTime", "cuBandwidth", "kernelTime", "kernelBandwidth", "gpuTime",
                                         "gpuBandwidth", "cpuTime", "tuneTime", "burnTime", "mcmcTime", "samplerTime")] + \
               [(n, C.c_int) for n in ("Nd", "dim", "kernel", "blocksz")] + \
               [(n, C.c_float) for n in ("sd", "acceptance", "ess")] + \
               [(n, C.c_int) for n in ("samples", "burnSamples", "device")]


class trace_rec(C.Structure):
    _fields_ = [("lik", C.c_double), ("ratio", C.c_double), ("logu", C.c_double), ("accept", C.c_int)]


TRACE_DTYPE = np.dtype([("lik", "f8"), ("ratio", "f8"), ("logu", "f8"), ("accept", "i4"), ("_pad", "i4")])


def build():
    subprocess.check_call(["make", "-s", "-C", ORACLE_DIR])


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB):
        build()
    L = C.CDLL(LIB, mode=C.RTLD_LOCAL)
    P = C.c_void_p
    L.gsl_rng_alloc.restype = P
    L.gsl_rng_alloc.argtypes = [P]
    L.gsl_rng_free.argtypes = [P]
    L.gsl_rng_set.argtypes = [P, C.c_ulong]
    L.gsl_rng_get.restype = C.c_ulong
    L.gsl_rng_get.argtypes = [P]
    L.gsl_rng_uniform.restype = C.c_double
    L.gsl_rng_uniform.argtypes = [P]
    L.gsl_rng_uniform_int.restype = C.c_ulong
    L.gsl_rng_uniform_int.argtypes = [P, C.c_ulong]
    L.gsl_ran_gaussian_ziggurat.restype = C.c_double
    L.gsl_ran_gaussian_ziggurat.argtypes = [P, C.c_double]
    L.gsl_ran_gaussian_pdf.restype = C.c_double
    L.gsl_ran_gaussian_pdf.argtypes = [C.c_double, C.c_double]
    L.log_likelihood.restype = C.c_double
    L.log_likelihood.argtypes = [P, data_str, mcmc_str, C.POINTER(out_str)]
    L.log_prior.restype = C.c_double
    L.log_prior.argtypes = [P, mcmc_str]
    L.cpu_sampler.argtypes = [data_str, P, mcmc_str, mcmc_tune_str, mcmc_v_str, C.POINTER(out_str)]
    L.tune_target_a_cpu_v2.argtypes = [data_str, P, mcmc_str, C.POINTER(mcmc_tune_str), P, C.c_double, C.c_int]
    L.tune_target_a_cpu.argtypes = [data_str, P, mcmc_str, C.POINTER(mcmc_tune_str), P]
    L.tune_ess_cpu.argtypes = [data_str, P, mcmc_str, C.POINTER(mcmc_tune_str), P, C.c_int]
    L.oracle_dgemv_colmajor_n.argtypes = [C.c_long, C.c_long, C.c_double, P, C.c_long, P, C.c_double, P]
    L.oracle_sgemv_colmajor_n.argtypes = [C.c_long, C.c_long, C.c_float, P, C.c_long, P, C.c_float, P]
    L.circular_autocorrelation.restype = C.c_double
    L.circular_autocorrelation.argtypes = [P, P, C.c_int, C.c_int, C.c_int]
    L.get_ess.restype = C.c_double
    L.get_ess.argtypes = [P, C.c_int, C.c_int, C.c_int, P]
    L.oracle_tree_reduce_d.restype = C.c_double
    L.oracle_tree_reduce_d.argtypes = [P, C.c_long, C.c_int, C.c_int, C.c_int]
    L.oracle_tree_reduce_f.restype = C.c_float
    L.oracle_tree_reduce_f.argtypes = [P, C.c_long, C.c_int, C.c_int, C.c_int]
    L.oracle_gpu_likelihood_d.restype = C.c_double
    L.oracle_gpu_likelihood_d.argtypes = [P, C.c_long, C.c_int, C.c_long, P, P, C.c_int, C.c_int]
    L.oracle_gpu_likelihood_f.restype = C.c_float
    L.oracle_gpu_likelihood_f.argtypes = [P, C.c_long, C.c_int, C.c_long, P, P, C.c_int, C.c_int]
    L.oracle_trace_begin.argtypes = [P, C.c_long]
    L.oracle_trace_end.restype = C.c_long
    L.oracle_set_verbose.argtypes = [C.c_int]
    L.oracle_log_likelihood_mt.restype = C.c_double
    L.oracle_log_likelihood_mt.argtypes = [P, P, C.c_long, C.c_int, C.c_long, C.c_int]
    L.oracle_log_likelihood_batched.argtypes = [P, C.c_int, P, C.c_long, C.c_int, C.c_long, P, P]
    _lib = L
    return L


def ptr(a):
    return a.ctypes.data_as(C.c_void_p)


class Rng:
    """gsl_rng (MT19937) from the gslcompat shim, seeded like init_rng_seed."""

    def __init__(self, seed=0, L=None):
        self.L = L or lib()
        mt = C.c_void_p.in_dll(self.L, "gsl_rng_mt19937")
        self.r = self.L.gsl_rng_alloc(mt)
        self.L.gsl_rng_set(self.r, seed)

    def set(self, seed):
        self.L.gsl_rng_set(self.r, seed)

    def __del__(self):
        try:
            self.L.gsl_rng_free(self.r)
        except Exception:
            pass


def log_likelihood(X, theta):
    """oracle log_likelihood (mcmc_cpu.c:419) on a Fortran-ordered X [N, d]."""
    L = lib()
    X = np.asfortranarray(X, dtype=np.float64)
    N, d = X.shape
    mvout = np.empty(max(N, 1), dtype=np.float64)
    th = np.ascontiguousarray(theta, dtype=np.float64)
    data = data_str(data=ptr(X).value, mvout=ptr(mvout).value)
    mcin = mcmc_str(ddata=d, Nd=N, Ns=1)
    res = out_str()
    val = L.log_likelihood(ptr(th), data, mcin, C.byref(res))
    return val, mvout[:N]


def cpu_sampler(X, seed, Ns, burnin, rwsd, theta0=None, trace=True):
    """Run the oracle cpu_sampler; returns dict(samples, burn, acceptance, trace)."""
    L = lib()
    X = np.asfortranarray(X, dtype=np.float64)
    N, d = X.shape
    mvout = np.empty(max(N, 1), dtype=np.float64)
    samples = np.zeros((Ns, d))
    burn = np.zeros((max(burnin, 1), d))
    if theta0 is not None:
        burn[0, :] = theta0
    data = data_str(data=ptr(X).value, mvout=ptr(mvout).value)
    mcin = mcmc_str(ddata=d, Nd=N, Ns=Ns, burnin=burnin, impl=0)
    mct = mcmc_tune_str(rwsd=rwsd, rwsdf=rwsd)
    mcdata = mcmc_v_str(samples=ptr(samples).value, burn=ptr(burn).value)
    res = out_str()
    rng = Rng(seed, L)
    cap = Ns + burnin + 8
    tr = np.zeros(cap, dtype=TRACE_DTYPE)
    if trace:
        L.oracle_trace_begin(ptr(tr), cap)
    L.cpu_sampler(data, rng.r, mcin, mct, mcdata, C.byref(res))
    n = L.oracle_trace_end() if trace else 0
    return dict(samples=samples, burn=burn, acceptance=res.acceptance, trace=tr[:n], res=res)


def sp_sampler_forced(seed, d, rwsdf, burnin, Ns, lik, theta0=None):
    """oracle_sp_sampler (oracle/sp_path.c): the single-precision sampler's host arithmetic with the float
    likelihoods fed by the caller (lik[0] = chain start, then one per iteration).  Returns dict(samples, burn,
    proposals, accepted, cposterior, trace)."""
    L = lib()
    L.oracle_sp_sampler.restype = C.c_long
    L.oracle_sp_sampler.argtypes = [C.c_void_p, C.c_int, C.c_float, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                    C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_float), C.c_void_p, C.c_long]
    lik = np.ascontiguousarray(lik, dtype=np.float32)
    assert len(lik) == burnin + Ns and burnin >= 1
    samples = np.zeros((Ns, d), dtype=np.float32)
    burn = np.zeros((burnin, d), dtype=np.float32)
    if theta0 is not None:
        burn[0, :] = theta0
    n = burnin - 1 + Ns
    props = np.zeros((n, d), dtype=np.float32)
    tr = np.zeros(n, dtype=TRACE_DTYPE)
    acc, cp = C.c_int(0), C.c_float(0)
    rng = Rng(seed, L)
    got = L.oracle_sp_sampler(rng.r, d, float(rwsdf), burnin, Ns, ptr(lik), ptr(burn), ptr(samples), ptr(props),
                              C.byref(acc), C.byref(cp), ptr(tr), n)
    assert got == n
    return dict(samples=samples, burn=burn, proposals=props, accepted=acc.value, cposterior=cp.value, trace=tr)
