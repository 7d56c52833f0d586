"""Python mirror of the C sampler / tuner entry points (include/mcmc_samplers.h).

The MH loops themselves are C (libmcmc_amd.so, reference signatures:
src/mcmc_gpu/mcmc_gpu.h:16-49, src/mcmc_gpu_sp/mcmc_gpu_sp.h:16-51,
src/mcmc_mp/mcmc_mp.h:18-52); this module only builds the by-value structs
(src/util/resources.h) and calls them, so tests and the bench read like the
reference's own drivers (src/mcmc_gpu/gpu_host.c:44-70).
"""
import ctypes as C
import os

import numpy as np

from . import _lib as L


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


class mcmc_int_v(C.Structure):
    _fields_ = [("proposed", C.c_void_p), ("current", C.c_void_p), ("proposedf", C.c_void_p), ("currentf", C.c_void_p)]


class mcmc_int(C.Structure):
    _fields_ = [("cposterior", C.c_double), ("pposterior", C.c_double), ("acceptance", C.c_double), ("u", C.c_double),
                ("cposteriorf", C.c_float), ("pposteriorf", C.c_float), ("acceptancef", C.c_float), ("uf", C.c_float)]


class gpu_v_str(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("size", "threads", "blocks", "maxThreads", "maxBlocks", "kernel", "cpuThresh")]


class out_str(C.Structure):
    _fields_ = [(n, C.c_float) for n in ("cuTime", "cuBandwidth", "kernelTime", "kernelBandwidth", "gpuTime",
                                         "gpuBandwidth", "cpuTime", "tuneTime", "burnTime", "mcmcTime", "samplerTime")] + \
               [(n, C.c_int) for n in ("Nd", "dim", "kernel", "blocksz")] + \
               [(n, C.c_float) for n in ("sd", "acceptance", "ess")] + \
               [(n, C.c_int) for n in ("samples", "burnSamples", "device")]


TRACE_DTYPE = np.dtype([("lik", "f8"), ("ratio", "f8"), ("logu", "f8"), ("accept", "i4"), ("_pad", "i4")])

IMPL_CPU, IMPL_GPU, IMPL_SP, IMPL_MP = 0, 1, 2, 3

_slib = None

EXPORTED_SYMBOLS = [
    "gpu_sampler", "tune_ess_gpu", "tune_target_a_gpu_v2", "tune_target_a_gpu", "acceptance_ratio_gpu",
    "log_prior_gpu", "print_gpu_info", "sp_sampler", "tune_ess_sp", "tune_target_a_sp_v2", "acceptance_ratio_sp",
    "log_prior_sp", "mp_sampler", "acceptance_ratio_mp", "log_prior_mp", "get_lowerbound", "get_blockSz",
    "circular_autocorrelation", "circular_autocorrelation_sp", "get_ess", "get_ess_sp",
    "calculate_normalised_sample_means", "calculate_normalised_sample_means_sp", "init_rng", "init_rng_seed",
    "free_rng", "metropolis_gpu", "burn_in_metropolis_gpu", "gpu_likelihood_d", "gpu_likelihood_f",
    "mcmc_amd_attach_context", "mcmc_amd_attached_context", "mcmc_amd_trace_begin", "mcmc_amd_trace_end",
    "mcmc_amd_set_verbose", "mcmc_amd_multichain_sampler", "mcmc_amd_mp_multichain_sampler",
    "cpu_sampler", "tune_ess_cpu", "tune_target_a_cpu_v2", "tune_target_a_cpu", "acceptance_ratio", "log_prior",
    "log_likelihood",
]


def load_samplers():
    """Load libmcmc_amd.so (which pulls in libmhlik.so).  No fallback."""
    global _slib
    if _slib is not None:
        return _slib
    L.load_mhlik()
    path = L.lib_path("libmcmc_amd.so")
    if not os.path.exists(path):
        raise ImportError("mcmc_amd: %s not found; build it with `make -C mcmc_amd/csrc`" % path)
    S = C.CDLL(path, mode=C.RTLD_LOCAL)
    P = C.c_void_p
    S.gpu_sampler.argtypes = [data_str, P, mcmc_str, mcmc_tune_str, mcmc_v_str, gpu_v_str, C.POINTER(out_str)]
    S.sp_sampler.argtypes = [data_str, P, mcmc_str, mcmc_tune_str, mcmc_v_str, gpu_v_str, C.POINTER(out_str)]
    S.mp_sampler.argtypes = [data_str, P, C.POINTER(mp_str), mcmc_str, mcmc_tune_str, mcmc_v_str, C.c_double,
                             C.POINTER(out_str)]
    S.tune_target_a_gpu_v2.argtypes = [data_str, P, mcmc_str, C.POINTER(mcmc_tune_str), gpu_v_str, P, C.c_double, C.c_int]
    S.tune_target_a_gpu.argtypes = [data_str, P, mcmc_str, C.POINTER(mcmc_tune_str), gpu_v_str, P, C.c_int]
    S.tune_ess_gpu.argtypes = [data_str, P, mcmc_str, C.POINTER(mcmc_tune_str), gpu_v_str, P, C.c_int]
    S.tune_target_a_sp_v2.argtypes = [data_str, P, mcmc_str, C.POINTER(mcmc_tune_str), gpu_v_str, P, C.c_float, C.c_int]
    S.tune_ess_sp.argtypes = [data_str, P, mcmc_str, C.POINTER(mcmc_tune_str), gpu_v_str, P, C.c_int]
    S.get_lowerbound.restype = C.c_double
    S.get_lowerbound.argtypes = [data_str, P, P, mcmc_str, mcmc_tune_str]
    S.get_blockSz.argtypes = [data_str, P, C.POINTER(mp_str), mcmc_str, mcmc_tune_str, P, C.c_double]
    S.get_ess.restype = C.c_double
    S.get_ess.argtypes = [P, C.c_int, C.c_int, C.c_int, P]
    S.get_ess_sp.restype = C.c_float
    S.get_ess_sp.argtypes = [P, C.c_int, C.c_int, C.c_int, P]
    S.circular_autocorrelation.restype = C.c_double
    S.circular_autocorrelation.argtypes = [P, P, C.c_int, C.c_int, C.c_int]
    S.log_prior_gpu.restype = C.c_double
    S.log_prior_gpu.argtypes = [P, mcmc_str]
    S.log_prior_sp.restype = C.c_float
    S.log_prior_sp.argtypes = [P, mcmc_str]
    S.log_prior_mp.restype = C.c_double
    S.log_prior_mp.argtypes = [P, mcmc_str]
    S.init_rng_seed.argtypes = [C.POINTER(P), C.c_ulong]
    S.init_rng.argtypes = [C.POINTER(P)]
    S.free_rng.argtypes = [P]
    S.gsl_rng_set.argtypes = [P, C.c_ulong]
    S.mcmc_amd_attach_context.argtypes = [P]
    S.mcmc_amd_attached_context.restype = P
    S.mcmc_amd_trace_begin.argtypes = [P, C.c_long]
    S.mcmc_amd_trace_end.restype = C.c_long
    S.mcmc_amd_set_verbose.argtypes = [C.c_int]
    S.mcmc_amd_set_verbose(0)
    _slib = S
    return S


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


class Rng:
    """gsl_rng handle seeded like the reference's init_rng_seed (alloc_util.c:261)."""

    def __init__(self, seed=0):
        self.S = load_samplers()
        self.r = C.c_void_p()
        self.S.init_rng_seed(C.byref(self.r), int(seed))

    def set(self, seed):
        self.S.gsl_rng_set(self.r, int(seed))

    def __del__(self):
        try:
            if self.r:
                self.S.free_rng(self.r)
        except Exception:
            pass


class attached:
    """Context manager: make the samplers use an existing Likelihood context."""

    def __init__(self, lik):
        self.lik = lik

    def __enter__(self):
        load_samplers().mcmc_amd_attach_context(self.lik.handle)
        return self.lik

    def __exit__(self, *exc):
        load_samplers().mcmc_amd_attach_context(None)


def _run_traced(S, call, cap):
    tr = np.zeros(cap, dtype=TRACE_DTYPE)
    S.mcmc_amd_trace_begin(_ptr(tr), cap)
    try:
        call()
    finally:
        n = S.mcmc_amd_trace_end()
    return tr[:min(n, cap)]


def gpu_sampler(X, seed, Ns, burnin, rwsd, theta0=None, trace=True, rng=None, n_rows=None, n_cols=None):
    """gpu_sampler (mcmc_gpu.h:16).  X: Fortran-ordered float64 [N, d], or None
    when a context is attached (then pass n_rows, n_cols)."""
    S = load_samplers()
    if X is not None:
        X = np.asfortranarray(X, dtype=np.float64)
        N, d = X.shape
    else:
        N, d = int(n_rows), int(n_cols)
    samples = np.zeros((Ns, d))
    burn = np.zeros((max(burnin, 1), d))
    if theta0 is not None:
        burn[0, :] = theta0
    data = data_str(data=_ptr(X).value if X is not None else None)
    mcin = mcmc_str(ddata=d, Nd=N, Ns=Ns, burnin=burnin, impl=IMPL_GPU)
    mct = mcmc_tune_str(rwsd=rwsd, rwsdf=rwsd)
    mcdata = mcmc_v_str(samples=_ptr(samples).value, burn=_ptr(burn).value)
    res = out_str()
    rng = rng or Rng(seed)
    call = lambda: S.gpu_sampler(data, rng.r, mcin, mct, mcdata, gpu_v_str(), C.byref(res))
    tr = _run_traced(S, call, Ns + burnin + 8) if trace else (call(), None)[1]
    return dict(samples=samples, burn=burn, acceptance=res.acceptance, trace=tr, res=res)


def sp_sampler(Xf, seed, Ns, burnin, rwsdf, theta0=None, trace=True, rng=None, n_rows=None, n_cols=None):
    """sp_sampler (mcmc_gpu_sp.h:16).  Xf: Fortran-ordered float32 [N, d]."""
    S = load_samplers()
    if Xf is not None:
        Xf = np.asfortranarray(Xf, dtype=np.float32)
        N, d = Xf.shape
    else:
        N, d = int(n_rows), int(n_cols)
    samples = np.zeros((Ns, d), dtype=np.float32)
    burn = np.zeros((max(burnin, 1), d), dtype=np.float32)
    if theta0 is not None:
        burn[0, :] = theta0
    data = data_str(dataf=_ptr(Xf).value if Xf is not None else None)
    mcin = mcmc_str(ddata=d, Nd=N, Ns=Ns, burnin=burnin, impl=IMPL_SP)
    mct = mcmc_tune_str(rwsd=rwsdf, rwsdf=rwsdf)
    mcdata = mcmc_v_str(samplesf=_ptr(samples).value, burnf=_ptr(burn).value)
    res = out_str()
    rng = rng or Rng(seed)
    call = lambda: S.sp_sampler(data, rng.r, mcin, mct, mcdata, gpu_v_str(), C.byref(res))
    tr = _run_traced(S, call, Ns + burnin + 8) if trace else (call(), None)[1]
    return dict(samples=samples, burn=burn, acceptance=res.acceptance, trace=tr, res=res)


def mp_sampler(X, seed, Ns, burnin, rwsd, block_rows, blocks, resamp_factor, bound, trace=True, rng=None):
    """mp_sampler (mcmc_mp.h:18).  X float64 [N, d]; the float copy is its rounding."""
    S = load_samplers()
    X = np.asfortranarray(X, dtype=np.float64)
    Xf = np.asfortranarray(X.astype(np.float32))
    N, d = X.shape
    samples = np.zeros((Ns, d))
    burn = np.zeros((max(burnin, 1), d))
    data = data_str(data=_ptr(X).value, dataf=_ptr(Xf).value)
    mcin = mcmc_str(ddata=d, Nd=N, Ns=Ns, burnin=burnin, impl=IMPL_MP)
    mct = mcmc_tune_str(rwsd=rwsd, rwsdf=rwsd)
    mcdata = mcmc_v_str(samples=_ptr(samples).value, burn=_ptr(burn).value)
    mp = mp_str(dataBlocks=blocks, dataBlockSz=block_rows, ResampFactor=resamp_factor)
    res = out_str()
    rng = rng or Rng(seed)
    call = lambda: S.mp_sampler(data, rng.r, C.byref(mp), mcin, mct, mcdata, float(bound), C.byref(res))
    tr = _run_traced(S, call, Ns + burnin + 8) if trace else (call(), None)[1]
    return dict(samples=samples, burn=burn, acceptance=res.acceptance, trace=tr, res=res, mp=mp)


def walk_to_contested(N, d, seed, theta, rwsd, reps=14, run=300, med_lo=0.5, med_hi=10.0):
    """Position a chain where accept/reject decisions are contested, on the
    attached context.  Short gpu_sampler runs continue from where the last one
    stopped, and after each the random-walk sd is rescaled until the run's
    median |log ratio| lies in [med_lo, med_hi] with 15-60 % of the proposals
    accepted: log u and the ratio are then of the same size, so every decision
    hinges on a log-likelihood difference of a few nats.  (Acceptance alone is
    not the criterion: while a chain is still climbing, ~40 % of its proposals
    go uphill and are accepted by hundreds of nats, the rest rejected by as
    much.)  Returns (theta, rwsd, acceptance, median |ratio|) of the last run.
    Parity-harness helper; the update is tune_target_a_gpu_v2's multiplicative
    rule (mcmc_gpu.cu:434-462) driven by the ratio instead of the acceptance."""
    theta = np.array(theta, dtype=np.float64)
    acc = med = 0.0
    for rep in range(reps):
        r = gpu_sampler(None, seed + 1000 + rep, run, 1, rwsd, theta0=theta, n_rows=N, n_cols=d)
        theta = r["samples"][-1].copy()
        acc = float(r["acceptance"])
        med = float(np.median(np.abs(r["trace"]["ratio"])))
        if med_lo <= med <= med_hi and 0.15 <= acc <= 0.6:
            break
        if med > med_hi or med < med_lo:
            rwsd *= min(max((3.0 / max(med, 1e-6)) ** 0.7, 0.02), 20.0)  # |ratio| ~ sd while climbing, ~ sd^2 at the mode
        else:
            rwsd *= 0.5 if acc < 0.15 else 1.5
    return theta, rwsd, acc, med


def tune_target_a_gpu_v2(X, seed, burnin, rwsd, ratio=0.25, max_reps=40, rng=None, n_rows=None, n_cols=None):
    S = load_samplers()
    if X is not None:
        X = np.asfortranarray(X, dtype=np.float64)
        N, d = X.shape
    else:
        N, d = int(n_rows), int(n_cols)
    init = np.zeros(d)
    data = data_str(data=_ptr(X).value if X is not None else None)
    mcin = mcmc_str(ddata=d, Nd=N, Ns=1, burnin=burnin, impl=IMPL_GPU)
    mct = mcmc_tune_str(rwsd=rwsd, rwsdf=rwsd)
    rng = rng or Rng(seed)
    S.tune_target_a_gpu_v2(data, rng.r, mcin, C.byref(mct), gpu_v_str(), _ptr(init), ratio, max_reps)
    return mct.rwsd


def cpu_sampler(X, seed, Ns, burnin, rwsd, theta0=None, trace=True):
    """cpu_sampler (mcmc_cpu.h:12) of the product library: the CPU path's entry
    point, evaluated on the GPU."""
    S = load_samplers()
    X = np.asfortranarray(X, dtype=np.float64)
    N, d = X.shape
    samples = np.zeros((Ns, d))
    burn = np.zeros((max(burnin, 1), d))
    if theta0 is not None:
        burn[0, :] = theta0
    mcin = mcmc_str(ddata=d, Nd=N, Ns=Ns, burnin=burnin, impl=IMPL_CPU)
    mct = mcmc_tune_str(rwsd=rwsd, rwsdf=rwsd)
    mcdata = mcmc_v_str(samples=_ptr(samples).value, burn=_ptr(burn).value)
    res = out_str()
    rng = Rng(seed)
    S.cpu_sampler.argtypes = [data_str, C.c_void_p, mcmc_str, mcmc_tune_str, mcmc_v_str, C.POINTER(out_str)]
    call = lambda: S.cpu_sampler(data_str(data=_ptr(X).value), rng.r, mcin, mct, mcdata, C.byref(res))
    tr = _run_traced(S, call, Ns + burnin + 8) if trace else (call(), None)[1]
    return dict(samples=samples, burn=burn, acceptance=res.acceptance, trace=tr, res=res)


def log_likelihood(X, theta):
    """log_likelihood (mcmc_cpu.h:34) of the product library."""
    S = load_samplers()
    X = np.asfortranarray(X, dtype=np.float64)
    N, d = X.shape
    th = np.ascontiguousarray(theta, dtype=np.float64)
    S.log_likelihood.restype = C.c_double
    S.log_likelihood.argtypes = [C.c_void_p, data_str, mcmc_str, C.POINTER(out_str)]
    res = out_str()
    return S.log_likelihood(_ptr(th), data_str(data=_ptr(X).value), mcmc_str(ddata=d, Nd=N, Ns=1, burnin=0, impl=IMPL_CPU),
                            C.byref(res))


def tune_ess_gpu(X, seed, burnin, rwsd, length, rng=None):
    """tune_ess_gpu (mcmc_gpu.h:32): returns the selected random-walk sd."""
    S = load_samplers()
    X = np.asfortranarray(X, dtype=np.float64)
    N, d = X.shape
    init = np.zeros(d)
    mcin = mcmc_str(ddata=d, Nd=N, Ns=1, burnin=burnin, impl=IMPL_GPU)
    mct = mcmc_tune_str(rwsd=rwsd, rwsdf=rwsd)
    rng = rng or Rng(seed)
    S.tune_ess_gpu(data_str(data=_ptr(X).value), rng.r, mcin, C.byref(mct), gpu_v_str(), _ptr(init), int(length))
    return mct.rwsd


def multichain_sampler(lik, seeds, Ns, burnin, rwsd, use_f32=False):
    """K lock-step chains on an existing Likelihood context (batched MFMA pass
    per iteration); chain k uses its own generator seeded seeds[k]."""
    S = load_samplers()
    K, d = len(seeds), lik.n_cols
    rngs = [Rng(s) for s in seeds]
    arr = (C.c_void_p * K)(*[r.r for r in rngs])
    samples = np.zeros((K, Ns, d))
    burn = np.zeros((K, max(burnin, 1), d))
    acc = np.zeros(K, dtype=np.int32)
    mcin = mcmc_str(ddata=d, Nd=lik.n_rows, Ns=Ns, burnin=burnin, impl=IMPL_GPU)
    S.mcmc_amd_multichain_sampler.argtypes = [C.c_void_p, C.c_int, C.c_void_p, mcmc_str, mcmc_tune_str, C.c_void_p,
                                              C.c_void_p, C.c_void_p, C.c_int]
    S.mcmc_amd_multichain_sampler(lik.handle, K, arr, mcin, mcmc_tune_str(rwsd=rwsd, rwsdf=rwsd), _ptr(samples),
                                  _ptr(burn), _ptr(acc), int(use_f32))
    return dict(samples=samples, burn=burn, accepted=acc)


def mp_multichain_sampler(lik, seeds, Ns, burnin, rwsd, block_rows, blocks, resamp_factor, bound, shared_phase=False,
                          theta0=None):
    """K lock-step mixed-precision chains on an existing Likelihood context
    (both matrices; segments=1 or segment_align=1024); chain k is mp_sampler with generator seeds[k]."""
    S = load_samplers()
    K, d = len(seeds), lik.n_cols
    rngs = [Rng(s) for s in seeds]
    arr = (C.c_void_p * K)(*[r.r for r in rngs])
    samples = np.zeros((K, Ns, d))
    burn = np.zeros((K, max(burnin, 1), d))
    acc = np.zeros(K, dtype=np.int32)
    bright = np.zeros(K, dtype=np.int32)
    mcin = mcmc_str(ddata=d, Nd=lik.n_rows, Ns=Ns, burnin=burnin, impl=IMPL_MP)
    mp = mp_str(dataBlocks=blocks, dataBlockSz=block_rows, ResampFactor=resamp_factor)
    S.mcmc_amd_mp_multichain_sampler.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.POINTER(mp_str), mcmc_str,
                                                 mcmc_tune_str, C.c_void_p, C.c_void_p, C.c_void_p, C.c_double,
                                                 C.c_int, C.c_void_p, C.c_void_p]
    th0 = None if theta0 is None else np.ascontiguousarray(theta0, dtype=np.float64)
    S.mcmc_amd_mp_multichain_sampler(lik.handle, K, arr, C.byref(mp), mcin, mcmc_tune_str(rwsd=rwsd, rwsdf=rwsd),
                                     _ptr(samples), _ptr(burn), _ptr(acc), float(bound), int(shared_phase),
                                     _ptr(bright), None if th0 is None else _ptr(th0))
    return dict(samples=samples, burn=burn, accepted=acc, bright=bright, mp=mp)


def get_ess(samples, max_lag):
    S = load_samplers()
    s = np.ascontiguousarray(samples, dtype=np.float64)
    ac = np.zeros(max_lag)
    return S.get_ess(_ptr(s), s.shape[0], s.shape[1], max_lag, _ptr(ac))
