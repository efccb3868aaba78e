"""ctypes binding of include/fmcmc_amd.h. Fails loudly when the HIP library is missing:
the engine has no CPU fallback."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("FMCMC_AMD_LIB") or os.path.join(_HERE, "lib", "libfmcmc_amd.so")  # override: A/B builds

_dp = C.POINTER(C.c_double)

FAM_GAUSSIAN_LINREG, FAM_LOGISTIC, FAM_IID_NORMAL = 1, 2, 3
KERNEL_NORMAL, KERNEL_NORMAL_REFLECTIVE, KERNEL_ADAPT, KERNEL_RAM, KERNEL_UNIF, KERNEL_UNIF_REFLECTIVE = 1, 2, 3, 4, 5, 6
RAM_QFUN_T_K, RAM_QFUN_NORMAL, RAM_QFUN_T_DF = 0, 1, 2   # fmcmc_kernel.ram_qfun
KERNEL_NMIRROR, KERNEL_UMIRROR = 7, 8
MIRROR_KERNELS = (KERNEL_NMIRROR, KERNEL_UMIRROR)
SIMPLE_KERNELS = (KERNEL_NORMAL, KERNEL_NORMAL_REFLECTIVE, KERNEL_UNIF, KERNEL_UNIF_REFLECTIVE) + MIRROR_KERNELS
SCHEME_JOINT, SCHEME_ORDERED, SCHEME_RANDOM, SCHEME_EXPLICIT = 0, 1, 2, 3
ABI_VERSION = 6
RNG_PHILOX, RNG_FED = 0, 1
OK, ERR_ARG, ERR_DEVICE, ERR_CHAIN, ERR_UNSUPPORTED = 0, 1, 2, 3, 4
CHAIN_OK, CHAIN_NAN_LOGPOST, CHAIN_NAN_RATIO, CHAIN_NOT_PD, CHAIN_BAD_WINDOW, CHAIN_SYNC_TIMEOUT = 0, 1, 2, 3, 4, 5
MAX_K = 128        # (FMCMC_MAX_K; every kernel / option up to MAX_K_WAVE, the joint simple kernels, kernel_adapt and kernel_ram beyond)
MAX_K_WAVE = 64

EXPORTS = ["fmcmc_abi_version", "fmcmc_last_error", "fmcmc_last_kernel", "fmcmc_device_count", "fmcmc_kept_rows",
           "fmcmc_validate", "fmcmc_mcmc_run_dev", "fmcmc_mcmc_run_host", "fmcmc_gelman_partial_len",
           "fmcmc_gelman_work_len",
           "fmcmc_gelman_partial_dev", "fmcmc_gelman_finish", "fmcmc_detmath_dev", "fmcmc_rng_stream_dev"]


class Model(C.Structure):
    _fields_ = [("family", C.c_int32), ("p", C.c_int32), ("n", C.c_int64), ("X", C.c_void_p),
                ("y", C.c_void_p), ("intercept", C.c_int32), ("guard", C.c_int32),
                ("prior_div", C.c_double)]


class Kernel(C.Structure):
    _fields_ = [("kind", C.c_int32), ("k", C.c_int32), ("mu", C.c_void_p), ("scale", C.c_void_p),
                ("lb", C.c_void_p), ("ub", C.c_void_p), ("fixed", C.c_void_p), ("scheme", C.c_int32),
                ("freq", C.c_int32), ("warmup", C.c_int32), ("bw", C.c_int32), ("until", C.c_double),
                ("eps", C.c_double), ("arate", C.c_double), ("Sd", C.c_double),
                ("scheme_seq", C.c_void_p), ("scheme_len", C.c_int32), ("nadapt", C.c_int32),
                ("constr", C.c_void_p), ("h_fixed", C.c_void_p), ("h_lb", C.c_void_p), ("h_ub", C.c_void_p),
                ("h_scale", C.c_void_p), ("h_scheme_seq", C.c_void_p),
                ("ram_qfun", C.c_int32), ("reserved", C.c_int32), ("ram_df", C.c_double), ("ram_eta_exp", C.c_double)]


class Run(C.Structure):
    _fields_ = [("nchains", C.c_int64), ("nsteps", C.c_int64), ("burnin", C.c_int64),
                ("thin", C.c_int64), ("seed", C.c_uint64), ("chain_base", C.c_int64),
                ("step_base", C.c_int64), ("rng_mode", C.c_int32), ("reserved", C.c_int32),
                ("fed_logu", C.c_void_p), ("fed_z", C.c_void_p)]


class State(C.Structure):
    _fields_ = [("theta0", C.c_void_p), ("f0", C.c_void_p), ("abs_iter", C.c_void_p),
                ("Sigma", C.c_void_p), ("mean_prev", C.c_void_p), ("have_mean", C.c_void_p),
                ("nerrors", C.c_void_p), ("fresh", C.c_int32), ("reserved", C.c_int32),
                ("scheme_cols", C.c_void_p), ("mirror_mu", C.c_void_p), ("mirror_scale", C.c_void_p),
                ("obs_arate", C.c_void_p)]


class Out(C.Structure):
    _fields_ = [("samples", C.c_void_p), ("logpost", C.c_void_p), ("draws", C.c_void_p),
                ("accept_count", C.c_void_p), ("accept_bits", C.c_void_p), ("status", C.c_void_p),
                ("status_step", C.c_void_p), ("status_theta", C.c_void_p), ("ld_rows", C.c_int64)]


_lib = None


def lib():
    """Loads fmcmc_amd/lib/libfmcmc_amd.so (built by fmcmc_amd.build / __graft_entry__.build)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "HIP engine library not found at %s. Build it with `python -m fmcmc_amd.build` "
                "(hipcc --offload-arch=gfx950). There is no CPU fallback." % LIB_PATH)
        L = C.CDLL(LIB_PATH)
        L.fmcmc_abi_version.restype = C.c_int
        if L.fmcmc_abi_version() != ABI_VERSION:
            raise RuntimeError("%s has ABI version %d, this package binds version %d: rebuild with "
                               "`python -m fmcmc_amd.build`." % (LIB_PATH, L.fmcmc_abi_version(), ABI_VERSION))
        L.fmcmc_last_error.restype = C.c_char_p
        L.fmcmc_last_kernel.restype = C.c_char_p
        L.fmcmc_device_count.restype = C.c_int
        L.fmcmc_kept_rows.restype = C.c_int64
        L.fmcmc_kept_rows.argtypes = [C.c_int64, C.c_int64, C.c_int64]
        L.fmcmc_validate.restype = C.c_int
        L.fmcmc_validate.argtypes = [C.POINTER(Model), C.POINTER(Kernel), C.POINTER(Run)]
        for nm in ("fmcmc_mcmc_run_dev", "fmcmc_mcmc_run_host"):
            f = getattr(L, nm)
            f.restype = C.c_int
        L.fmcmc_mcmc_run_dev.argtypes = [C.POINTER(Model), C.POINTER(Kernel), C.POINTER(Run),
                                         C.POINTER(State), C.POINTER(Out), C.c_void_p]
        L.fmcmc_mcmc_run_host.argtypes = [C.POINTER(Model), C.POINTER(Kernel), C.POINTER(Run),
                                          C.POINTER(State), C.POINTER(Out), C.c_int]
        L.fmcmc_gelman_partial_len.restype = C.c_int64
        L.fmcmc_gelman_partial_len.argtypes = [C.c_int32]
        L.fmcmc_gelman_partial_dev.restype = C.c_int
        L.fmcmc_gelman_work_len.restype = C.c_int64
        L.fmcmc_gelman_work_len.argtypes = [C.c_int64, C.c_int32]
        L.fmcmc_gelman_partial_dev.argtypes = [C.c_void_p, C.c_int64, C.c_int32, C.c_int64, C.c_int64,
                                               C.c_int64, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p,
                                               C.c_void_p, C.c_void_p]
        L.fmcmc_gelman_finish.restype = C.c_int
        L.fmcmc_gelman_finish.argtypes = [_dp, C.c_int32, C.c_int64, _dp, _dp]
        L.fmcmc_rng_stream_dev.restype = C.c_int
        L.fmcmc_rng_stream_dev.argtypes = [C.c_uint64, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_int32,
                                           C.c_double, C.c_void_p, C.c_void_p, C.c_void_p]
        L.fmcmc_detmath_dev.restype = C.c_int
        L.fmcmc_detmath_dev.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_uint64, C.c_void_p]
        _lib = L
    return _lib


def last_error():
    return lib().fmcmc_last_error().decode("utf-8", "replace")


def last_kernel():
    """Kernel variant chosen by this thread's last sweep (diagnostic; results never depend on it)."""
    return lib().fmcmc_last_kernel().decode("utf-8", "replace")
