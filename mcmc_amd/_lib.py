"""ctypes binding of libmhlik.so (include/mhlik.h).  Fails loudly when the
library is missing: there is no Python or CPU substitute for the HIP path."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))

MAX_SEGMENTS = 64
MAX_BATCH = 64
COMM_ID_BYTES = 128


class MhlikError(RuntimeError):
    def __init__(self, code, text):
        super().__init__("mhlik error %d: %s" % (code, text))
        self.code = code


class Config(C.Structure):
    _fields_ = [
        ("n_rows", C.c_int64),
        ("n_cols", C.c_int32),
        ("precision", C.c_int32),
        ("device", C.c_int32),
        ("rank", C.c_int32),
        ("world", C.c_int32),
        ("segments", C.c_int32),
        ("layout", C.c_int32),
        ("variant", C.c_int32),
        ("timing", C.c_int32),
        ("segment_align", C.c_int32),
        ("reserved", C.c_int32 * 5),
    ]


class Info(C.Structure):
    _fields_ = [
        ("n_rows", C.c_int64),
        ("n_cols", C.c_int32),
        ("precision", C.c_int32),
        ("rank", C.c_int32),
        ("world", C.c_int32),
        ("local_rows", C.c_int64),
        ("tiles_f64", C.c_int64),
        ("tiles_f32", C.c_int64),
        ("tile_rows_f64", C.c_int32),
        ("tile_rows_f32", C.c_int32),
        ("layout", C.c_int32),
        ("variant", C.c_int32),
        ("compute_units", C.c_int32),
        ("reserved", C.c_int32),
        ("hbm_bytes", C.c_int64),
    ]


class MpState(C.Structure):
    _fields_ = [
        ("block_rows", C.c_int32),
        ("blocks", C.c_int32),
        ("resamp_factor", C.c_int32),
        ("resamp_choose", C.c_int32),
        ("u_seed", C.c_uint64),
        ("u_call", C.c_uint64),
        ("bound_eps", C.c_double),
    ]


class MpBatch(C.Structure):
    _fields_ = [
        ("block_rows", C.c_int32),
        ("blocks", C.c_int32),
        ("resamp_factor", C.c_int32),
        ("K", C.c_int32),
        ("resamp_choose", C.c_void_p),
        ("u_seed", C.c_void_p),
        ("u_call", C.c_uint64),
        ("bound_eps", C.c_double),
    ]


class ChainBatchF64(C.Structure):
    _fields_ = [("n", C.c_int64), ("increments", C.c_void_p), ("log_u", C.c_void_p), ("samples", C.c_void_p),
                ("lik", C.c_void_p), ("ratio", C.c_void_p), ("accepted", C.c_void_p), ("n_accepted", C.c_int64),
                ("kernel_ms", C.c_float), ("kernel_samples", C.c_int32)]


class ChainBatchF32(ChainBatchF64):
    pass


class KChainMp(C.Structure):
    _fields_ = [("resamp_factor", C.c_int32), ("reserved", C.c_int32), ("resamp_choose", C.c_void_p),
                ("u_seed", C.c_void_p), ("u_call", C.c_uint64), ("bound_eps", C.c_double)]


class KChainBatch(C.Structure):
    _fields_ = [("n", C.c_int64), ("increments", C.c_void_p), ("log_u", C.c_void_p), ("mp", C.POINTER(KChainMp)),
                ("samples", C.c_void_p), ("lik", C.c_void_p), ("ratio", C.c_void_p), ("accepted", C.c_void_p),
                ("bright_blocks", C.c_void_p), ("dark_blocks", C.c_void_p), ("kernel_ms", C.c_float),
                ("kernel_samples", C.c_int32)]


KCHAIN_F64, KCHAIN_F32, KCHAIN_MP = 1, 2, 3


def lib_path(name="libmhlik.so"):
    return os.path.join(_HERE, "lib", name)


_lib = None

# every symbol include/mhlik.h declares: (name, restype, argtypes)
_P = C.c_void_p
_SIGS = [
    ("mhlik_create", C.c_int, [C.POINTER(Config), C.POINTER(_P)]),
    ("mhlik_destroy", None, [_P]),
    ("mhlik_last_error", C.c_char_p, [_P]),
    ("mhlik_local_rows", C.c_int, [_P, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    ("mhlik_upload_f64", C.c_int, [_P, _P, C.c_int64, C.c_int64]),
    ("mhlik_upload_f32", C.c_int, [_P, _P, C.c_int64, C.c_int64]),
    ("mhlik_fill_synthetic", C.c_int, [_P, C.c_uint64]),
    ("mhlik_synthetic_host_f64", C.c_int, [C.c_uint64, C.c_int64, C.c_int32, C.c_int64, C.c_int64, _P, C.c_int64]),
    ("mhlik_synthetic_beta", C.c_int, [C.c_uint64, C.c_int32, _P]),
    ("mhlik_download_f64", C.c_int, [_P, _P, C.c_int64]),
    ("mhlik_download_f32", C.c_int, [_P, _P, C.c_int64]),
    ("mhlik_loglik_f64", C.c_int, [_P, _P, C.POINTER(C.c_double)]),
    ("mhlik_loglik_f32", C.c_int, [_P, _P, C.POINTER(C.c_float)]),
    ("mhlik_loglik_f64_noalloc", C.c_int, [_P, _P, C.POINTER(C.c_double)]),
    ("mhlik_loglik_batched_f64", C.c_int, [_P, _P, C.c_int32, _P]),
    ("mhlik_loglik_batched_f32", C.c_int, [_P, _P, C.c_int32, _P]),
    ("mhlik_loglik_batched_begin_f64", C.c_int, [_P, _P, C.c_int32]),
    ("mhlik_loglik_batched_begin_f32", C.c_int, [_P, _P, C.c_int32]),
    ("mhlik_loglik_batched_end", C.c_int, [_P, _P]),
    ("mhlik_set_batched_f32_mode", C.c_int, [_P, C.c_int32]),
    ("mhlik_set_batched_f64_mode", C.c_int, [_P, C.c_int32]),
    ("mhlik_rowdots_f64", C.c_int, [_P, _P, _P]),
    ("mhlik_mp_reset", C.c_int, [_P, C.c_int32, C.c_int32]),
    ("mhlik_mp_loglik", C.c_int, [_P, C.POINTER(MpState), _P, C.POINTER(C.c_double), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    ("mhlik_mp_bound", C.c_int, [_P, _P, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    ("mhlik_mp_last_slots", C.c_int, [_P, _P]),
    ("mhlik_mp_last_slots_batched", C.c_int, [_P, C.c_int32, _P]),
    ("mhlik_mp_reset_batched", C.c_int, [_P, C.c_int32, C.c_int32, C.c_int32]),
    ("mhlik_mp_loglik_batched", C.c_int, [_P, C.POINTER(MpBatch), _P, _P, _P, _P]),
    ("mhlik_mp_loglik_batched_begin", C.c_int, [_P, C.POINTER(MpBatch), _P]),
    ("mhlik_mp_loglik_batched_end", C.c_int, [_P, _P, _P, _P]),
    ("mhlik_mp_indicators_batched", C.c_int, [_P, _P]),
    ("mhlik_chain_begin_f64", C.c_int, [_P, _P, C.c_double, C.POINTER(C.c_double)]),
    ("mhlik_chain_begin_f32", C.c_int, [_P, _P, C.c_double, C.POINTER(C.c_float)]),
    ("mhlik_chain_resume_f64", C.c_int, [_P, _P, C.c_double, C.c_double]),
    ("mhlik_chain_resume_f32", C.c_int, [_P, _P, C.c_float, C.c_double]),
    ("mhlik_chain_run_f64", C.c_int, [_P, C.POINTER(ChainBatchF64)]),
    ("mhlik_chain_run_f32", C.c_int, [_P, C.POINTER(ChainBatchF32)]),
    ("mhlik_chain_submit_f64", C.c_int, [_P, C.POINTER(ChainBatchF64)]),
    ("mhlik_chain_submit_f32", C.c_int, [_P, C.POINTER(ChainBatchF32)]),
    ("mhlik_chain_collect_f64", C.c_int, [_P, C.POINTER(ChainBatchF64)]),
    ("mhlik_chain_collect_f32", C.c_int, [_P, C.POINTER(ChainBatchF32)]),
    ("mhlik_chain_state_f64", C.c_int, [_P, _P, C.POINTER(C.c_double)]),
    ("mhlik_chain_state_f32", C.c_int, [_P, _P, C.POINTER(C.c_float)]),
    ("mhlik_host_alloc", C.c_int, [C.c_size_t, C.POINTER(_P)]),
    ("mhlik_host_free", None, [_P]),
    ("mhlik_kchain_begin", C.c_int, [_P, C.c_int32, C.c_int32, _P, C.c_double, C.POINTER(KChainMp), _P]),
    ("mhlik_kchain_resume", C.c_int, [_P, C.c_int32, C.c_int32, _P, _P, C.c_double]),
    ("mhlik_kchain_run", C.c_int, [_P, C.POINTER(KChainBatch)]),
    ("mhlik_kchain_state", C.c_int, [_P, _P, _P]),
    ("mhlik_autocorrelation_f64", C.c_int, [_P, _P, C.c_int64, C.c_int32, C.c_int32, _P, C.POINTER(C.c_double)]),
    ("mhlik_autocorrelation_f32", C.c_int, [_P, _P, C.c_int64, C.c_int32, C.c_int32, _P, C.POINTER(C.c_float)]),
    ("mhlik_comm_make_id", C.c_int, [_P]),
    ("mhlik_comm_init", C.c_int, [_P, _P]),
    ("mhlik_comm_destroy", C.c_int, [_P]),
    ("mhlik_comm_ranks", C.c_int, [_P, C.POINTER(C.c_int32)]),
    ("mhlik_last_kernel_name", C.c_char_p, [_P]),
    ("mhlik_last_segment_sums", C.c_int, [_P, _P]),
    ("mhlik_last_kernel_ms", C.c_int, [_P, C.POINTER(C.c_float), C.POINTER(C.c_int64)]),
    ("mhlik_get_info", C.c_int, [_P, C.POINTER(Info)]),
    ("mhlik_version", C.c_char_p, []),
]
EXPORTED_SYMBOLS = [s[0] for s in _SIGS]


def load_mhlik():
    """Load libmhlik.so and declare its prototypes.  Raises if it is missing:
    build it with `python -c "import __graft_entry__ as g; g.build()"` or
    `make -C mcmc_amd/csrc`."""
    global _lib
    if _lib is not None:
        return _lib
    path = lib_path()
    if not os.path.exists(path):
        raise ImportError(
            "mcmc_amd: %s not found -- the MI355X likelihood path has no CPU fallback; "
            "build it with `make -C mcmc_amd/csrc`" % path
        )
    # PyTorch wheels bundle their own HIP runtime and RCCL (torch 2.10+rocm7.0 here, beside ROCm 7.2 in /opt/rocm).
    # Whichever copy a process loads FIRST serves every later DT_NEEDED / dlopen of the same soname, so a process that
    # loads this library, creates a context and only then imports torch ends up with two runtimes: RCCL then fails
    # ("ncclCommInitRank: unhandled cuda error") or the process aborts at exit ("double free or corruption").  The
    # Python layer uses torch.distributed as its control plane anyway, so it settles the order here: torch first.
    # MCMC_AMD_NO_TORCH=1 skips this (a process that never imports torch; the C drivers never do).
    if os.environ.get("MCMC_AMD_NO_TORCH") != "1":
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
    lib = C.CDLL(path, mode=C.RTLD_LOCAL)
    for name, res, args in _SIGS:
        fn = getattr(lib, name)  # AttributeError if the library lacks a declared symbol
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib
