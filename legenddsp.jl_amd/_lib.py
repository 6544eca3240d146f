"""Loader of csrc/libldsp_hip.so (the C ABI of include/ldsp.h) and the Context
object every compute entry point goes through.

There is NO fallback path: if the shared library is missing or a call fails,
a LdspError is raised.  PyTorch is used only as the owner of device memory and
of the HIP stream (tensor.data_ptr() / torch.cuda.current_stream()).
"""
import ctypes as C
import os
import subprocess
import threading

from . import _abi

_HERE = os.path.dirname(os.path.abspath(__file__))
_CSRC = os.path.join(_HERE, "csrc")
_SO = os.environ.get("LDSP_HIP_LIB") or os.path.join(_CSRC, "libldsp_hip.so")   # LDSP_HIP_LIB: another build of the same ABI (tuning experiments)


class LdspError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"ldsp error {code}: {msg}")
        self.code = code


def build(force=False, quiet=True):
    """Compile every HIP source for gfx950 into csrc/libldsp_hip.so (in-tree)."""
    cmd = ["make", "-j4", "-C", _CSRC] + (["-B"] if force else []) + (["-s"] if quiet else [])   # four translation units
    subprocess.check_call(cmd)
    return _SO


_lib = None
_lock = threading.Lock()

_VOIDP = C.c_void_p
_I64 = C.c_int64
_I32 = C.c_int32
_DBL = C.c_double


def _declare(lib):
    def sig(name, argtypes, restype=C.c_int):
        fn = getattr(lib, name)
        fn.argtypes = argtypes
        fn.restype = restype

    sig("ldsp_abi_version", [])
    sig("ldsp_abi_sizeof", [C.c_int], C.c_int64)
    sig("ldsp_last_error_string", [], C.c_char_p)
    sig("ldsp_ctx_create", [C.c_int, C.POINTER(_VOIDP)])
    sig("ldsp_ctx_destroy", [_VOIDP])
    sig("ldsp_ctx_set_stream", [_VOIDP, _VOIDP])
    sig("ldsp_ctx_use_own_stream", [_VOIDP])
    sig("ldsp_ctx_synchronize", [_VOIDP])
    sig("ldsp_ctx_set_option", [_VOIDP, C.c_char_p, _I64])
    sig("ldsp_ctx_enable_timing", [_VOIDP, C.c_int])
    sig("ldsp_ctx_last_kernel_ms", [_VOIDP, C.POINTER(C.c_float)])
    sig("ldsp_ctx_last_stage_ms", [_VOIDP, C.c_int, C.POINTER(C.c_float)])
    sig("ldsp_ctx_last_kernel_name", [_VOIDP], C.c_char_p)
    sig("ldsp_icpc_check_params", [_VOIDP])
    sig("ldsp_icpc_run", [_VOIDP, _VOIDP, _I64, C.POINTER(_abi.IcpcParams), C.POINTER(_abi.IcpcOut)])
    sig("ldsp_icpc_pz_trap_run", [_VOIDP, _VOIDP, _I64, C.POINTER(_abi.IcpcParams), _VOIDP, _VOIDP])
    sig("ldsp_icpc_pz_trap_run_u16", [_VOIDP, _VOIDP, _I64, C.POINTER(_abi.IcpcParams), _VOIDP, _VOIDP])
    sig("ldsp_trap_grid_run", [_VOIDP, _VOIDP, _I64, C.POINTER(_abi.TrapGridParams), _I32, _VOIDP, _VOIDP, _VOIDP])
    sig("ldsp_sg_grid_run", [_VOIDP, _VOIDP, _I64, C.POINTER(_abi.TrapGridParams), C.POINTER(_abi.Trap), _DBL, _DBL, _I32, _VOIDP, _I32,
                              _VOIDP, _VOIDP, _VOIDP, _VOIDP, _VOIDP, _VOIDP, _VOIDP])
    sig("ldsp_fir_grid_run", [_VOIDP, _VOIDP, _I64, C.POINTER(_abi.TrapGridParams), _I32, _I32, _VOIDP, _VOIDP, _VOIDP])
    sig("ldsp_cusp_coeffs", [C.POINTER(_abi.CuspZac), _VOIDP])
    sig("ldsp_zac_coeffs", [C.POINTER(_abi.CuspZac), _VOIDP])
    sig("ldsp_sg_coeffs", [_I32, _I32, _I32, _VOIDP])
    # optional symbols (declared in include/ldsp.h; bound when present)
    opt = {
        "ldsp_sipm_run": [_VOIDP, _VOIDP, _I64, C.POINTER(_abi.SipmParams), C.POINTER(_abi.SipmOut)],
        "ldsp_sipm_run_u16": [_VOIDP, _VOIDP, _I64, C.POINTER(_abi.SipmParams), C.POINTER(_abi.SipmOut)],
        "ldsp_rdfilt_invcr": [_VOIDP, _VOIDP, _I64, _I32, _DBL, _VOIDP],
        "ldsp_rdfilt_integrator": [_VOIDP, _VOIDP, _I64, _I32, _DBL, _VOIDP],
        "ldsp_rdfilt_trap": [_VOIDP, _VOIDP, _I64, _I32, _abi.Trap, _VOIDP],
        "ldsp_rdfilt_fir": [_VOIDP, _VOIDP, _I64, _I32, _VOIDP, _I32, _VOIDP],
        "ldsp_rdfilt_derivative": [_VOIDP, _VOIDP, _I64, _I32, _DBL, _VOIDP],
        "ldsp_rdfilt_haar": [_VOIDP, _VOIDP, _I64, _I32, _I32, _VOIDP],
        "ldsp_rdfilt_moving_window": [_VOIDP, _VOIDP, _I64, _I32, _I32, _VOIDP],
        "ldsp_rdfilt_moving_window_multi": [_VOIDP, _VOIDP, _I64, _I32, _I32, _VOIDP],
        "ldsp_rdfilt_affine": [_VOIDP, _VOIDP, _I64, _I32, _I32, _I32, _DBL, _DBL, _VOIDP, _I32, _VOIDP],
        "ldsp_signalstats": [_VOIDP, _VOIDP, _I64, _I32, _I32, _I32, _DBL, _DBL] + [_VOIDP] * 4,
        "ldsp_tailstats": [_VOIDP, _VOIDP, _I64, _I32, _I32, _I32, _DBL, _DBL] + [_VOIDP] * 3,
        "ldsp_extremestats": [_VOIDP, _VOIDP, _I64, _I32, _I32, _I32, _DBL, _DBL] + [_VOIDP] * 4,
        "ldsp_thresholdstats": [_VOIDP, _VOIDP, _I64, _I32, _DBL, _DBL, _VOIDP],
        "ldsp_thresholdstats_mad": [_VOIDP, _VOIDP, _I64, _I32, _DBL, _DBL, _VOIDP],
        "ldsp_saturation": [_VOIDP, _VOIDP, _I64, _I32, _I32, _I32, _DBL, _DBL] + [_VOIDP] * 4,
        "ldsp_get_wvf_maximum": [_VOIDP, _VOIDP, _I64, _I32, _I32, _I32, _VOIDP],
        "ldsp_intersect": [_VOIDP, _VOIDP, _I64, _I32, _DBL, _DBL, _VOIDP, _I32, _VOIDP, _VOIDP],
        "ldsp_intersect_maximum": [_VOIDP, _VOIDP, _I64, _I32, _DBL, _DBL, _VOIDP, _I32, _I32, C.POINTER(_abi.TrigOut)],
        "ldsp_multi_intersect": [_VOIDP, _VOIDP, _I64, _I32, _DBL, _DBL, _VOIDP, _I32, _I32, _I32, _I32, _I32, _VOIDP, _VOIDP],
        "ldsp_signal_estimator": [_VOIDP, _VOIDP, _I64, _I32, _DBL, _DBL, _VOIDP, _abi.Dni, _VOIDP],
        "ldsp_icpc_run_opts": [_VOIDP, _VOIDP, _I64, C.POINTER(_abi.IcpcParams), C.POINTER(_abi.IcpcOpts), C.POINTER(_abi.IcpcOut)],
        "ldsp_qc_features": [_VOIDP, _VOIDP, _I64, _I32, _I32, _I32, _I32, _VOIDP, _VOIDP],
        "ldsp_qc_features_len": [_I32, _I32],
    }
    for name, argtypes in opt.items():
        if hasattr(lib, name):
            sig(name, argtypes)


# every symbol include/ldsp.h declares (checked by tests/test_abi.py)
DECLARED_SYMBOLS = [
    "ldsp_abi_version", "ldsp_abi_sizeof", "ldsp_ctx_create", "ldsp_ctx_destroy", "ldsp_ctx_set_stream",
    "ldsp_ctx_use_own_stream",
    "ldsp_ctx_synchronize", "ldsp_last_error_string", "ldsp_ctx_set_option", "ldsp_ctx_enable_timing",
    "ldsp_ctx_last_kernel_ms", "ldsp_ctx_last_stage_ms", "ldsp_ctx_last_kernel_name", "ldsp_icpc_check_params", "ldsp_icpc_run", "ldsp_icpc_pz_trap_run", "ldsp_icpc_pz_trap_run_u16", "ldsp_trap_grid_run", "ldsp_fir_grid_run", "ldsp_sg_grid_run", "ldsp_sipm_run", "ldsp_sipm_run_u16",
    "ldsp_rdfilt_invcr", "ldsp_rdfilt_integrator", "ldsp_rdfilt_trap", "ldsp_rdfilt_fir",
    "ldsp_rdfilt_derivative", "ldsp_rdfilt_haar", "ldsp_rdfilt_moving_window",
    "ldsp_rdfilt_moving_window_multi", "ldsp_rdfilt_affine", "ldsp_cusp_coeffs", "ldsp_zac_coeffs",
    "ldsp_sg_coeffs", "ldsp_signalstats", "ldsp_tailstats", "ldsp_extremestats", "ldsp_thresholdstats",
    "ldsp_thresholdstats_mad", "ldsp_saturation", "ldsp_get_wvf_maximum", "ldsp_intersect",
    "ldsp_intersect_maximum", "ldsp_multi_intersect", "ldsp_signal_estimator", "ldsp_qc_features", "ldsp_qc_features_len", "ldsp_icpc_run_opts",
]


def lib():
    """The loaded shared library; raises if it has not been built."""
    global _lib
    with _lock:
        if _lib is None:
            if not os.path.exists(_SO):
                raise LdspError(-100, f"{_SO} not found: run __graft_entry__.build() "
                                      "(legenddsp_jl_amd has no CPU fallback)")
            if os.environ.get("LDSP_HIP_LIB") and not os.environ.get("LDSP_ALLOW_STALE"):
                # a development build selected by hand: refuse one that is older than the sources it was built from (a stale
                # diagnostic library once ended a profiling run with an undefined symbol after the minutes had been spent)
                newest = max((os.path.getmtime(os.path.join(_CSRC, f)) for f in os.listdir(_CSRC) if f.endswith((".hip", ".hpp", ".inc"))), default=0.0)
                if os.path.getmtime(_SO) < newest:
                    raise LdspError(-104, f"{_SO} is older than the sources in {_CSRC}: rebuild it (tools/dev_build*.sh) or set LDSP_ALLOW_STALE=1")
            l = C.CDLL(_SO)
            _declare(l)
            if l.ldsp_abi_version() != _abi.LDSP_ABI_VERSION:
                raise LdspError(-101, "ABI version mismatch between _abi.py and libldsp_hip.so")
            for which, st in enumerate((_abi.IcpcParams, _abi.IcpcOut, _abi.SipmParams, _abi.SipmOut, _abi.TrigOut, _abi.IcpcOpts)):
                if l.ldsp_abi_sizeof(which) != C.sizeof(st):
                    raise LdspError(-102, f"struct size mismatch for {st.__name__}: "
                                          f"{l.ldsp_abi_sizeof(which)} (C) vs {C.sizeof(st)} (ctypes)")
            _lib = l
    return _lib


def check(rc):
    if rc != 0:
        raise LdspError(rc, lib().ldsp_last_error_string().decode(errors="replace"))


class Context:
    """One ldsp_ctx per GPU (single-owner, like the reference's stateless functors
    it is re-entrant only across distinct contexts)."""

    def __init__(self, device=None, use_torch_stream=True):
        import torch
        if not torch.cuda.is_available():
            raise LdspError(-103, "no HIP device visible: legenddsp_jl_amd has no CPU fallback")
        if device is None:
            device = torch.cuda.current_device()
        self.device = int(torch.device("cuda", device).index) if not isinstance(device, int) else device
        self._h = _VOIDP()
        check(lib().ldsp_ctx_create(self.device, C.byref(self._h)))
        self.use_torch_stream = use_torch_stream

    def bind_stream(self):
        if self.use_torch_stream:
            import torch
            s = torch.cuda.current_stream(self.device).cuda_stream
            check(lib().ldsp_ctx_set_stream(self._h, _VOIDP(s)))

    def set_stream(self, hip_stream):
        """Launch on the given hipStream_t (an integer handle, e.g. torch.cuda.Stream().cuda_stream; 0 / None = the null stream).  The
        stream stays the caller's: see the lifetime rule in include/ldsp.h.  Use with use_torch_stream=False."""
        check(lib().ldsp_ctx_set_stream(self._h, _VOIDP(hip_stream or None)))

    def use_own_stream(self):
        check(lib().ldsp_ctx_use_own_stream(self._h))

    @property
    def handle(self):
        return self._h

    def set_option(self, key, value):
        check(lib().ldsp_ctx_set_option(self._h, key.encode(), int(value)))

    def enable_timing(self, on=True):
        check(lib().ldsp_ctx_enable_timing(self._h, int(on)))

    def last_stage_ms(self, stage):
        ms = C.c_float()
        check(lib().ldsp_ctx_last_stage_ms(self._h, int(stage), C.byref(ms)))
        return ms.value

    def last_kernel_name(self):
        """rocprofv3's name (inside ldsp::, no template arguments) of the dominant kernel of the last run call."""
        return lib().ldsp_ctx_last_kernel_name(self._h).decode()

    def last_kernel_ms(self):
        ms = C.c_float()
        check(lib().ldsp_ctx_last_kernel_ms(self._h, C.byref(ms)))
        return ms.value

    def synchronize(self):
        check(lib().ldsp_ctx_synchronize(self._h))

    def close(self):
        if self._h:
            lib().ldsp_ctx_destroy(self._h)
            self._h = _VOIDP()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_default_ctx = {}


def default_context(device=None):
    import torch
    dev = torch.cuda.current_device() if device is None else int(torch.device("cuda", device).index if not isinstance(device, int) else device)
    if dev not in _default_ctx:
        _default_ctx[dev] = Context(dev)
    return _default_ctx[dev]
