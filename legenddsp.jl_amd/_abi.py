"""ctypes mirror of include/ldsp.h (struct layouts, limits, status codes).

Pure declarations: importing this module loads no native code.
"""
import ctypes as C

LDSP_ABI_VERSION = 4
LDSP_OK = 0
LDSP_ERR_INVALID_ARG = -1
LDSP_ERR_WINDOW = -2
LDSP_ERR_HIP = -3
LDSP_ERR_UNSUPPORTED = -4
LDSP_ERR_NOMEM = -5

LDSP_MAX_L = 32768
LDSP_MAX_EST_PTS = 64
LDSP_MAX_EST_DEG = 5
LDSP_MAX_SG_PTS = 65
LDSP_MAX_FIR_TAPS = 8192
LDSP_MAX_TRIG = 64
LDSP_MAX_MULTI = 128

f32p = C.POINTER(C.c_float)
i32p = C.POINTER(C.c_int32)
f64p = C.POINTER(C.c_double)


class Trap(C.Structure):
    _fields_ = [("navg", C.c_int32), ("ngap", C.c_int32), ("navg2", C.c_int32)]

    def __repr__(self):
        return f"Trap({self.navg},{self.ngap},{self.navg2})"

    @property
    def flen(self):
        return self.navg + self.ngap + self.navg2


class CuspZac(C.Structure):
    _fields_ = [("sigma", C.c_double), ("flat", C.c_int32), ("length", C.c_int32),
                ("tau", C.c_double), ("beta", C.c_double)]


class Dni(C.Structure):
    _fields_ = [("npts", C.c_int32), ("degree", C.c_int32)]


class IcpcParams(C.Structure):
    _fields_ = [
        ("L", C.c_int32), ("_pad0", C.c_int32),
        ("t_first", C.c_double), ("dt", C.c_double), ("unit_per_us", C.c_double),
        ("sat_low", C.c_double), ("sat_high", C.c_double),
        ("bl_from", C.c_int32), ("bl_until", C.c_int32),
        ("tail_from", C.c_int32), ("tail_until", C.c_int32),
        ("pz_c", C.c_double),
        ("t0_trap", Trap), ("t0_mintot", C.c_int32), ("t0_threshold", C.c_double),
        ("t0inv_trap", Trap), ("tx_mintot", C.c_int32),
        ("int_est", Dni),
        ("qdrift_d1", C.c_double), ("qdrift_d2", C.c_double),
        ("lq_d1", C.c_double), ("lq_d2", C.c_double),
        ("trap_fixed", Trap * 3), ("trap_opt", Trap),
        ("trap_pickoff", C.c_double), ("sig_est", Dni),
        ("cusp", CuspZac), ("zac", CuspZac),
        ("cusp_pickoff", C.c_double), ("zac_pickoff", C.c_double),
        ("sg_npts", C.c_int32 * 3), ("sg_degree", C.c_int32),
        ("cur_left", C.c_double), ("cur_right", C.c_double),
        ("intrace_nsigma", C.c_double), ("intrace_mintot", C.c_int32), ("_pad1", C.c_int32),
        ("bl_left", C.c_double), ("bl_right", C.c_double),
    ]


ICPC_F32_COLS = [
    "blmean", "blsigma", "blslope", "bloffset",
    "tailmean", "tailsigma", "tailslope", "tailoffset",
    "t0", "t10", "t50", "t80", "t90", "t99", "t50_current", "drift_time",
    "tail_tau", "tail_mean", "tail_sigma",
    "e_max", "e_min",
    "e_10410", "e_535", "e_313", "e_10410_inv", "e_313_inv", "t0_inv",
    "e_trap", "e_cusp", "e_zac",
    "e_trap_max", "e_cusp_max", "e_zac_max",
    "t_trap_max", "t_cusp_max", "t_zac_max",
    "qdrift", "lq",
    "a_sg", "a_60", "a_100", "a_raw",
    "inTrace_intersect",
]
ICPC_I32_COLS = ["inTrace_n", "n_sat_low", "n_sat_high", "n_sat_low_cons", "n_sat_high_cons"]
ICPC_COLS = ICPC_F32_COLS + ICPC_I32_COLS  # order of ldsp_icpc_out and of the oracle's columns
assert len(ICPC_COLS) == 48


class IcpcOut(C.Structure):
    _fields_ = [(c, C.c_void_p) for c in ICPC_COLS] + [("stride", C.c_int64)]


class SipmParams(C.Structure):
    _fields_ = [
        ("L", C.c_int32), ("_pad0", C.c_int32),
        ("t_first", C.c_double), ("dt", C.c_double), ("unit_per_us", C.c_double),
        ("trunc_from", C.c_int32), ("trunc_until", C.c_int32),
        ("sg_npts", C.c_int32), ("sg_degree", C.c_int32),
        ("sg_mintot", C.c_int32), ("sg_maxtot", C.c_int32),
        ("sg_min_thr", C.c_double), ("sg_max_thr", C.c_double), ("sg_nsigma", C.c_double),
        ("sg_min_dc_thr", C.c_double), ("sg_max_dc_thr", C.c_double), ("sg_nsigma_dc", C.c_double),
        ("pz_c", C.c_double), ("trap", Trap),
        ("trap_mintot", C.c_int32), ("trap_maxtot", C.c_int32), ("_pad1", C.c_int32),
        ("trap_min_thr", C.c_double), ("trap_max_thr", C.c_double), ("trap_nsigma", C.c_double),
        ("trap_min_dc_thr", C.c_double), ("trap_max_dc_thr", C.c_double), ("trap_nsigma_dc", C.c_double),
    ]


class IcpcOpts(C.Structure):
    """ldsp_icpc_opts: per-call variations of ldsp_icpc_run_opts (explicit arguments, no context state)."""
    _fields_ = [("ext_baseline", C.c_void_p), ("ext_baseline_scale", C.c_double), ("main_only", C.c_int32), ("in_u16", C.c_int32)]


# dtype of the slabs of one trigger group (ldsp_trig_out): positions are Float64 like the reference's time axis
# (src/dsp_sipm.jl:87-88, ragged columns :149-156), `max` is a value of the float32 signal
TRIG_FIELDS = ("x", "x_high", "x_tot", "max")
TRIG_DTYPES = {"x": "float64", "x_high": "float64", "x_tot": "float64", "max": "float32"}


class TrigOut(C.Structure):
    _fields_ = [("count", C.c_void_p), ("x", C.c_void_p), ("x_high", C.c_void_p),
                ("x_tot", C.c_void_p), ("max", C.c_void_p), ("cap", C.c_int32), ("_pad", C.c_int32)]


SIPM_SCALAR_COLS = [
    "t_max", "t_min", "t_max_lar", "t_min_lar", "e_max", "e_min", "e_max_lar", "e_min_lar",
    "blmean", "blsigma", "blslope", "bloffset", "wfmean", "wfsigma", "wfslope", "wfoffset",
    "threshold", "threshold_DC", "threshold_trap", "threshold_DC_trap",
]
SIPM_TRIG_GROUPS = ["trig", "trig_DC", "trig_trap", "trig_DC_trap"]


class SipmOut(C.Structure):
    _fields_ = [(c, C.c_void_p) for c in SIPM_SCALAR_COLS] + [(g, TrigOut) for g in SIPM_TRIG_GROUPS]


LDSP_MAX_GRID = 64


class TrapGridParams(C.Structure):
    """ldsp_trapgrid_params (include/ldsp.h)."""
    _fields_ = [("L", C.c_int32), ("_pad0", C.c_int32), ("t_first", C.c_double), ("dt", C.c_double),
                ("bl_from", C.c_int32), ("bl_until", C.c_int32), ("pz_c", C.c_double), ("sig_est", Dni),
                ("pick_mode", C.c_int32), ("tx_mintot", C.c_int32), ("pick_time", C.c_double)]
