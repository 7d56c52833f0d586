"""Host-side mirror of the likelihood context (include/mhlik.h).

`Likelihood` owns one `mhlik_ctx`: the design matrix resident in HBM and the
kernels that evaluate  L(theta) = -sum_i log(1 + exp(x_i . theta))  on it,
replacing the reference's gpu_likelihood_d / gpu_likelihood_f / mp_likelihood
(src/mcmc_gpu/mcmc_gpu.cu:214, src/mcmc_gpu_sp/mcmc_gpu_sp.cu:216,
src/mcmc_mp/mcmc_mp.cu:475).
"""
import ctypes as C

import numpy as np

from . import _lib as L

F64, F32 = 1, 2
LAYOUT_DEFAULT, LAYOUT_TILED, LAYOUT_COLMAJOR = 0, 1, 2


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


class Likelihood:
    def __init__(self, n_rows, n_cols, precision=F64, device=-1, rank=0, world=1, segments=0,
                 layout=LAYOUT_DEFAULT, variant=0, timing=False, segment_align=0):
        self._lib = L.load_mhlik()
        self._ctx = C.c_void_p()
        cfg = L.Config()
        cfg.n_rows, cfg.n_cols, cfg.precision, cfg.device = int(n_rows), int(n_cols), int(precision), int(device)
        cfg.rank, cfg.world, cfg.segments = int(rank), int(world), int(segments)
        cfg.layout, cfg.variant, cfg.timing = int(layout), int(variant), int(bool(timing))
        cfg.segment_align = int(segment_align)
        rc = self._lib.mhlik_create(C.byref(cfg), C.byref(self._ctx))
        if rc != 0:
            raise L.MhlikError(rc, (self._lib.mhlik_last_error(None) or b"").decode())
        self.n_rows, self.n_cols, self.precision = int(n_rows), int(n_cols), int(precision)
        self.rank, self.world = int(rank), int(world)
        self.segments = int(segments) if segments else 8
        b, n = C.c_int64(), C.c_int64()
        self._chk(self._lib.mhlik_local_rows(self._ctx, C.byref(b), C.byref(n)))
        self.local_begin, self.local_rows = b.value, n.value

    # -- plumbing ---------------------------------------------------------
    def _chk(self, rc):
        if rc != 0:
            raise L.MhlikError(rc, (self._lib.mhlik_last_error(self._ctx) or b"").decode())

    @property
    def handle(self):
        """The raw mhlik_ctx* (for the C samplers)."""
        return self._ctx

    def close(self):
        if self._ctx:
            self._lib.mhlik_destroy(self._ctx)
            self._ctx = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- data ---------------------------------------------------------------
    def upload(self, X, row0=0):
        """X: Fortran-ordered (column-major) array [rows, d]; row 0 of X is
        global row `row0` (data_util.c:33 layout: X[i, j] = -y_i * x_ij)."""
        X = np.asarray(X)
        if X.ndim != 2 or X.shape[1] != self.n_cols:
            raise ValueError("X must be [rows, %d]" % self.n_cols)
        if not X.flags.f_contiguous:
            X = np.asfortranarray(X)
        ld = X.shape[0]
        if X.dtype == np.float64:
            self._chk(self._lib.mhlik_upload_f64(self._ctx, _ptr(X), ld, row0))
        elif X.dtype == np.float32:
            self._chk(self._lib.mhlik_upload_f32(self._ctx, _ptr(X), ld, row0))
        else:
            raise TypeError("X must be float64 or float32")

    def fill_synthetic(self, seed):
        self._chk(self._lib.mhlik_fill_synthetic(self._ctx, int(seed)))

    def download(self, dtype=np.float64):
        X = np.empty((self.local_rows, self.n_cols), dtype=dtype, order="F")
        ld = max(self.local_rows, 1)
        if dtype == np.float64:
            self._chk(self._lib.mhlik_download_f64(self._ctx, _ptr(X), ld))
        else:
            self._chk(self._lib.mhlik_download_f32(self._ctx, _ptr(X), ld))
        return X

    # -- evaluation ---------------------------------------------------------
    def loglik(self, theta):
        th = np.ascontiguousarray(theta, dtype=np.float64)
        if th.shape != (self.n_cols,):
            raise ValueError("theta must have %d entries" % self.n_cols)
        out = C.c_double()
        self._chk(self._lib.mhlik_loglik_f64(self._ctx, _ptr(th), C.byref(out)))
        return out.value

    def loglik_noalloc(self, theta):
        """mhlik_loglik_f64_noalloc: tile partials in the library's static device array, no allocation in the call."""
        th = np.ascontiguousarray(theta, dtype=np.float64)
        if th.shape != (self.n_cols,):
            raise ValueError("theta must have %d entries" % self.n_cols)
        out = C.c_double()
        self._chk(self._lib.mhlik_loglik_f64_noalloc(self._ctx, _ptr(th), C.byref(out)))
        return out.value

    def loglik_f32(self, theta):
        th = np.ascontiguousarray(theta, dtype=np.float32)
        if th.shape != (self.n_cols,):
            raise ValueError("theta must have %d entries" % self.n_cols)
        out = C.c_float()
        self._chk(self._lib.mhlik_loglik_f32(self._ctx, _ptr(th), C.byref(out)))
        return out.value

    def loglik_batched(self, Theta, dtype=np.float64):
        Th = np.ascontiguousarray(Theta, dtype=dtype)
        if Th.ndim != 2 or Th.shape[1] != self.n_cols:
            raise ValueError("Theta must be [K, %d]" % self.n_cols)
        out = np.empty(Th.shape[0], dtype=np.float64)
        fn = self._lib.mhlik_loglik_batched_f64 if dtype == np.float64 else self._lib.mhlik_loglik_batched_f32
        self._chk(fn(self._ctx, _ptr(Th), Th.shape[0], _ptr(out)))
        return out

    def loglik_batched_begin(self, Theta, dtype=np.float64):
        """First half of loglik_batched: queue the evaluation and return; loglik_batched_end() collects it."""
        Th = np.ascontiguousarray(Theta, dtype=dtype)
        if Th.ndim != 2 or Th.shape[1] != self.n_cols:
            raise ValueError("Theta must be [K, %d]" % self.n_cols)
        fn = self._lib.mhlik_loglik_batched_begin_f64 if dtype == np.float64 else self._lib.mhlik_loglik_batched_begin_f32
        self._chk(fn(self._ctx, _ptr(Th), Th.shape[0]))
        self._begun = Th.shape[0]

    def loglik_batched_end(self):
        out = np.empty(getattr(self, "_begun", 0) or 1, dtype=np.float64)
        self._chk(self._lib.mhlik_loglik_batched_end(self._ctx, _ptr(out)))
        return out

    def set_batched_f32_mode(self, mode):
        """'exact' (f32 MFMA, the single-theta kernel's fmaf chain) or 'split' (three bfloat16 pieces, default)."""
        self.set_batched_f32_mode_raw({"exact": 0, "split": 1}[mode])

    def set_batched_f64_mode(self, mode):
        """'fast' (table-driven softplus, default) or 'exact' (full-length series) epilogue of the double K-chain pass."""
        self._chk(self._lib.mhlik_set_batched_f64_mode(self._ctx, {"exact": 0, "fast": 1}[mode]))

    def set_batched_f32_mode_raw(self, mode):
        self._chk(self._lib.mhlik_set_batched_f32_mode(self._ctx, int(mode)))

    def rowdots(self, theta):
        th = np.ascontiguousarray(theta, dtype=np.float64)
        out = np.empty(self.local_rows, dtype=np.float64)
        self._chk(self._lib.mhlik_rowdots_f64(self._ctx, _ptr(th), _ptr(out)))
        return out

    def segment_sums(self):
        out = np.zeros(self.segments, dtype=np.float64)
        self._chk(self._lib.mhlik_last_segment_sums(self._ctx, _ptr(out)))
        return out

    # -- mixed precision ----------------------------------------------------
    def mp_reset(self, block_rows, blocks):
        self._chk(self._lib.mhlik_mp_reset(self._ctx, int(block_rows), int(blocks)))

    def mp_loglik(self, theta, block_rows, blocks, resamp_factor, resamp_choose, u_seed, u_call, bound_eps):
        th = np.ascontiguousarray(theta, dtype=np.float64)
        st = L.MpState(int(block_rows), int(blocks), int(resamp_factor), int(resamp_choose), int(u_seed), int(u_call),
                       float(bound_eps))
        out, nb, nd = C.c_double(), C.c_int32(), C.c_int32()
        self._chk(self._lib.mhlik_mp_loglik(self._ctx, C.byref(st), _ptr(th), C.byref(out), C.byref(nb), C.byref(nd)))
        return out.value, nb.value, nd.value

    def mp_last_slots(self):
        """[double sums | float sums | bright counts] per segment of the last mp_loglik (mp_bound: [sums | minima])."""
        out = np.zeros(3 * self.segments, dtype=np.float64)
        self._chk(self._lib.mhlik_mp_last_slots(self._ctx, _ptr(out)))
        return out

    def mp_bound(self, theta):
        th = np.ascontiguousarray(theta, dtype=np.float64)
        s, e = C.c_double(), C.c_double()
        self._chk(self._lib.mhlik_mp_bound(self._ctx, _ptr(th), C.byref(s), C.byref(e)))
        return s.value, e.value

    # -- mixed precision, K lock-step chains ----------------------------------
    def mp_reset_batched(self, block_rows, blocks, K):
        self._chk(self._lib.mhlik_mp_reset_batched(self._ctx, int(block_rows), int(blocks), int(K)))
        self._mpb = (int(block_rows), int(blocks), int(K))

    def mp_loglik_batched(self, Theta, resamp_factor, resamp_choose, u_seed, u_call, bound_eps):
        """Theta [K, d]; resamp_choose, u_seed: one per chain.  Returns (values[K], bright[K], dark[K])."""
        block_rows, blocks, K = self._mpb
        Th = np.ascontiguousarray(Theta, dtype=np.float64)
        if Th.shape != (K, self.n_cols):
            raise ValueError("Theta must be [%d, %d]" % (K, self.n_cols))
        ch = np.ascontiguousarray(resamp_choose, dtype=np.int32)
        sd = np.ascontiguousarray(u_seed, dtype=np.uint64)
        if ch.shape != (K,) or sd.shape != (K,):
            raise ValueError("resamp_choose and u_seed must have one entry per chain")
        st = L.MpBatch(block_rows, blocks, int(resamp_factor), K, _ptr(ch).value, _ptr(sd).value, int(u_call),
                       float(bound_eps))
        out = np.empty(K)
        nb, nd = np.empty(K, dtype=np.int32), np.empty(K, dtype=np.int32)
        self._chk(self._lib.mhlik_mp_loglik_batched(self._ctx, C.byref(st), _ptr(Th), _ptr(out), _ptr(nb), _ptr(nd)))
        return out, nb, nd

    def mp_last_slots_batched(self):
        """[K, 4, segments] per-segment values of the last K-chain mixed-precision evaluation (sharded contexts)."""
        _, _, K = self._mpb
        out = np.zeros((K, 4, self.segments))
        self._chk(self._lib.mhlik_mp_last_slots_batched(self._ctx, K, _ptr(out)))
        return out

    def mp_indicators_batched(self):
        block_rows, blocks, K = self._mpb
        z = np.empty((K, blocks), dtype=np.int8)
        self._chk(self._lib.mhlik_mp_indicators_batched(self._ctx, _ptr(z)))
        return z

    # -- multi GPU ------------------------------------------------------------
    # ---- MH chain kept on the device (mhlik_chain_*) ----
    def chain_begin(self, theta0, prior_sd, dtype=np.float64):
        """Start a chain at theta0; returns cposterior = prior + L(theta0)."""
        th = np.ascontiguousarray(theta0, dtype=dtype)
        if dtype == np.float64:
            out = C.c_double()
            self._chk(self._lib.mhlik_chain_begin_f64(self._ctx, _ptr(th), float(prior_sd), C.byref(out)))
        else:
            out = C.c_float()
            self._chk(self._lib.mhlik_chain_begin_f32(self._ctx, _ptr(th), float(prior_sd), C.byref(out)))
        return out.value

    def chain_resume(self, current, cposterior, prior_sd, dtype=np.float64):
        cur = np.ascontiguousarray(current, dtype=dtype)
        fn = self._lib.mhlik_chain_resume_f64 if dtype == np.float64 else self._lib.mhlik_chain_resume_f32
        self._chk(fn(self._ctx, _ptr(cur), float(cposterior), float(prior_sd)))

    def chain_run(self, increments, log_u, dtype=np.float64):
        """Run len(log_u) iterations; returns dict(samples, lik, ratio, accepted, n_accepted, kernel_ms)."""
        inc = np.ascontiguousarray(increments, dtype=dtype)
        lu = np.ascontiguousarray(log_u, dtype=dtype)
        n, d = inc.shape
        samples = np.empty((n, d), dtype=dtype)
        lik, ratio, acc = np.empty(n), np.empty(n), np.empty(n, dtype=np.uint8)
        B = L.ChainBatchF64 if dtype == np.float64 else L.ChainBatchF32
        b = B(n=n, increments=_ptr(inc).value, log_u=_ptr(lu).value, samples=_ptr(samples).value,
              lik=_ptr(lik).value, ratio=_ptr(ratio).value, accepted=_ptr(acc).value)
        fn = self._lib.mhlik_chain_run_f64 if dtype == np.float64 else self._lib.mhlik_chain_run_f32
        self._chk(fn(self._ctx, C.byref(b)))
        return dict(samples=samples, lik=lik, ratio=ratio, accepted=acc, n_accepted=b.n_accepted,
                    kernel_ms=b.kernel_ms, kernel_samples=b.kernel_samples)

    def chain_submit(self, increments, log_u, dtype=np.float64, want_samples=True):
        """Queue len(log_u) <= 4096 iterations and return at once (at most two batches in flight)."""
        inc = np.ascontiguousarray(increments, dtype=dtype)
        lu = np.ascontiguousarray(log_u, dtype=dtype)
        B = L.ChainBatchF64 if dtype == np.float64 else L.ChainBatchF32
        b = B(n=inc.shape[0], increments=_ptr(inc).value, log_u=_ptr(lu).value,
              samples=_ptr(inc).value if want_samples else None)  # any non-NULL pointer: "keep the rows"
        fn = self._lib.mhlik_chain_submit_f64 if dtype == np.float64 else self._lib.mhlik_chain_submit_f32
        self._chk(fn(self._ctx, C.byref(b)))

    def chain_collect(self, n, dtype=np.float64, want_samples=True):
        """Wait for the oldest batch in flight (n iterations) and return its outputs like chain_run."""
        samples = np.empty((n, self.n_cols), dtype=dtype) if want_samples else None
        lik, ratio, acc = np.empty(n), np.empty(n), np.empty(n, dtype=np.uint8)
        B = L.ChainBatchF64 if dtype == np.float64 else L.ChainBatchF32
        b = B(n=n, samples=_ptr(samples).value if want_samples else None, lik=_ptr(lik).value,
              ratio=_ptr(ratio).value, accepted=_ptr(acc).value)
        fn = self._lib.mhlik_chain_collect_f64 if dtype == np.float64 else self._lib.mhlik_chain_collect_f32
        self._chk(fn(self._ctx, C.byref(b)))
        return dict(samples=samples, lik=lik, ratio=ratio, accepted=acc, n_accepted=b.n_accepted,
                    kernel_ms=b.kernel_ms, kernel_samples=b.kernel_samples)

    def chain_state(self, dtype=np.float64):
        cur = np.empty(self.n_cols, dtype=dtype)
        if dtype == np.float64:
            cp = C.c_double()
            self._chk(self._lib.mhlik_chain_state_f64(self._ctx, _ptr(cur), C.byref(cp)))
        else:
            cp = C.c_float()
            self._chk(self._lib.mhlik_chain_state_f32(self._ctx, _ptr(cur), C.byref(cp)))
        return cur, cp.value

    # ---- K lock-step chains kept on the device (mhlik_kchain_*) ----
    def kchain_begin(self, theta0, prior_sd, mode=1):
        """Start K chains at theta0 [K, d] (mode 1 = f64, 2 = f32 pass); returns cposterior[K]."""
        th = np.ascontiguousarray(theta0, dtype=np.float64)
        K = th.shape[0]
        cp = np.empty(K)
        self._chk(self._lib.mhlik_kchain_begin(self._ctx, int(mode), K, _ptr(th), float(prior_sd), None, _ptr(cp)))
        return cp

    def kchain_resume(self, current, cposterior, prior_sd, mode=1):
        cur = np.ascontiguousarray(current, dtype=np.float64)
        cp = np.ascontiguousarray(cposterior, dtype=np.float64)
        self._chk(self._lib.mhlik_kchain_resume(self._ctx, int(mode), cur.shape[0], _ptr(cur), _ptr(cp), float(prior_sd)))

    def kchain_run(self, increments, log_u):
        """increments [n, K, d], log_u [n, K]; returns dict(samples, lik, ratio, accepted, kernel_ms)."""
        inc = np.ascontiguousarray(increments, dtype=np.float64)
        lu = np.ascontiguousarray(log_u, dtype=np.float64)
        n, K, d = inc.shape
        samples = np.empty((n, K, d))
        lik, ratio, acc = np.empty((n, K)), np.empty((n, K)), np.empty((n, K), dtype=np.uint8)
        b = L.KChainBatch(n=n, increments=_ptr(inc).value, log_u=_ptr(lu).value, samples=_ptr(samples).value,
                          lik=_ptr(lik).value, ratio=_ptr(ratio).value, accepted=_ptr(acc).value)
        self._chk(self._lib.mhlik_kchain_run(self._ctx, C.byref(b)))
        return dict(samples=samples, lik=lik, ratio=ratio, accepted=acc, kernel_ms=b.kernel_ms,
                    kernel_samples=b.kernel_samples)

    def kchain_state(self, K):
        cur, cp = np.empty((K, self.n_cols)), np.empty(K)
        self._chk(self._lib.mhlik_kchain_state(self._ctx, _ptr(cur), _ptr(cp)))
        return cur, cp

    # -- chain statistics ---------------------------------------------------------
    def autocorrelation(self, samples, lags, dtype=np.float64):
        """circular_autocorrelation on the device: returns (out_v[lags], sum of out_v[1:])."""
        m = np.ascontiguousarray(samples, dtype=dtype)
        out = np.empty(lags, dtype=dtype)
        if dtype == np.float64:
            s = C.c_double()
            self._chk(self._lib.mhlik_autocorrelation_f64(self._ctx, _ptr(m), m.shape[0], m.shape[1], lags, _ptr(out), C.byref(s)))
        else:
            s = C.c_float()
            self._chk(self._lib.mhlik_autocorrelation_f32(self._ctx, _ptr(m), m.shape[0], m.shape[1], lags, _ptr(out), C.byref(s)))
        return out, s.value

    @staticmethod
    def make_comm_id():
        lib = L.load_mhlik()
        buf = (C.c_uint8 * L.COMM_ID_BYTES)()
        rc = lib.mhlik_comm_make_id(buf)
        if rc != 0:
            raise L.MhlikError(rc, (lib.mhlik_last_error(None) or b"").decode())
        return bytes(buf)

    def comm_init(self, comm_id):
        buf = (C.c_uint8 * L.COMM_ID_BYTES).from_buffer_copy(comm_id)
        self._chk(self._lib.mhlik_comm_init(self._ctx, buf))

    def comm_destroy(self):
        self._chk(self._lib.mhlik_comm_destroy(self._ctx))

    def comm_ranks(self):
        n = C.c_int32()
        self._chk(self._lib.mhlik_comm_ranks(self._ctx, C.byref(n)))
        return n.value

    # -- instrumentation ----------------------------------------------------------
    def last_kernel_name(self):
        return (self._lib.mhlik_last_kernel_name(self._ctx) or b"").decode()

    def last_kernel(self):
        ms, nbytes = C.c_float(), C.c_int64()
        self._chk(self._lib.mhlik_last_kernel_ms(self._ctx, C.byref(ms), C.byref(nbytes)))
        return ms.value, nbytes.value

    def info(self):
        inf = L.Info()
        self._chk(self._lib.mhlik_get_info(self._ctx, C.byref(inf)))
        return {k: getattr(inf, k) for k, _ in L.Info._fields_ if k != "reserved"}


def synthetic_host(seed, n_rows, n_cols, row0=0, rows=None):
    """Host copy of the seeded synthetic data set (same bits as the device fill)."""
    lib = L.load_mhlik()
    rows = n_rows - row0 if rows is None else rows
    X = np.empty((rows, n_cols), dtype=np.float64, order="F")
    rc = lib.mhlik_synthetic_host_f64(int(seed), int(n_rows), int(n_cols), int(row0), int(rows), _ptr(X), max(rows, 1))
    if rc != 0:
        raise L.MhlikError(rc, "mhlik_synthetic_host_f64: bad arguments")
    return X
