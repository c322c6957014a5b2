"""ctypes binding of libsbr_amd.so (include/sbr_amd.h).  Plumbing only: no numerics live here.

The library is HIP-only.  Loading works on a machine without a GPU (so the symbol table can be
checked there), but sbr_create fails loudly - there is no CPU fallback anywhere in the product.
"""
import ctypes as C
import os

from . import build as _build

NX, NOBS, NSTATE, NACT, NCTRL, KLA_HIST = 14, 18, 15, 2, 25, 10
NSCEN, NSERIES, NSAMP = 8, 14, 48
NCYC_ACT, NCYC_OBS, NCYC_DIAG = 3, 3, 12
ABI_VERSION = 6      # SBR_ABI_VERSION of include/sbr_amd.h; load() refuses a library that reports another
NTRACE = 36          # per traced env and call (enum SBR_TR_* in sbr_amd.h)
(TR_T, TR_X0, TR_KLA, TR_EC, TR_REWARD, TR_DONE, TR_U_DO, TR_U_EC, TR_E_EC, TR_IE_EC, TR_DCV_EC, TR_R_EQI, TR_R_OCI, TR_R_AE,
 TR_R_EC, TR_N_IV, TR_KLA_FIRST, TR_EC_FIRST, TR_E_EC_FIRST, TR_IE_EC_FIRST, TR_DCV_EC_FIRST, TR_PLAN, TR_PLAN_FIRST) = (
     0, 1, 15, 16, 17, 18, 19, 20, 21, 22, 23, 24, 25, 26, 27, 28, 29, 30, 31, 32, 33, 34, 35)
# rows of the ctrl block (enum in sbr_amd.h)
C_T, C_SO_M1, C_SO_M2, C_SNO_M1, C_SNO_M2, C_IE_DO, C_IE_EC, C_EC_LAST = range(8)
C_KLA_HIST0 = 8
C_KLA_LAST = C_KLA_HIST0 + KLA_HIST - 1
C_QW, C_RETURN, C_STEPS, C_DONE, C_STATUS, C_KLA_SUM, C_PLAN = (C_KLA_LAST + 1, C_KLA_LAST + 2, C_KLA_LAST + 3, C_KLA_LAST + 4,
                                                                 C_KLA_LAST + 5, C_KLA_LAST + 6, C_KLA_LAST + 7)
PLAN_SLAVED = 128    # SBR_PLAN_SLAVED: C_PLAN / TR_PLAN = Butcher-5 step count of the env's last interval + this bit if So was held
# sbr_query (enum SBR_Q_* in sbr_amd.h)
(Q_ONE_WAVE_ENVS, Q_STEP_SMALL_BATCH_ENVS, Q_STEP_BLOCK, Q_STEP_WAVES, Q_STEP_TWO_WAVES_ABOVE_ENVS, Q_FUSED_ONE_WAVE_MAX_ENVS,
 Q_ROLLOUT_WAVES, Q_RESET_BLOCK, Q_SCHEME) = range(9)
ST_NEGATIVE, ST_NEAR_POLE, ST_NONFINITE = 1, 2, 4     # SBR_ST_* bits of the status row
# cfg.reward_kind: module_reward_EQIOCI.py (SBROS-v1) / module_reward_continuous_G2ANET.py / module_reward_continuous.py
REWARD_KINDS = {"eqi_oci": 0, "g2anet": 1, "oci": 2}

_DBL = ("Ya Yh fp ixb ixp muH Ks Koh Kno bH eta_g eta_h kh Kx muA Knh bA Koa ka "
        "WV IV dt t_delta t_cycle T_fill T3_0 T3_end T4_end T5_end t_settle t_draw "
        "So_sat Kla_min Kla_max Kc_DO tauI_DO tauD_DO EC_min EC_max Kc_EC tauI_EC tauD_EC EC_conc "
        "act_DO_max act_EC_max biomass_setpoint Qeff settler_area settler_vmax").split()


class SbrConfig(C.Structure):
    _fields_ = [(n, C.c_double) for n in _DBL] + [
        ("t_ratio", C.c_double * 8), ("cyc_Kc", C.c_double), ("cyc_tauI", C.c_double), ("cyc_tauD", C.c_double),
        ("cyc_dt", C.c_double),
        ("x0", C.c_double * NX), ("substeps", C.c_int32), ("out_f64", C.c_int32),
        ("terminal", C.c_int32), ("reward_kind", C.c_int32), ("act_f64", C.c_int32), ("random_scenario", C.c_int32),
        ("scheme", C.c_int32), ("reserved_", C.c_int32)]


class SbrError(RuntimeError):
    pass


# every symbol include/sbr_amd.h declares: name -> (restype, argtypes)
_VP, _I64, _U64, _I32 = C.c_void_p, C.c_int64, C.c_uint64, C.c_int32
SYMBOLS = {
    "sbr_version": (C.c_char_p, []),
    "sbr_abi_version": (C.c_int, []),
    "sbr_default_config": (C.c_int, [C.POINTER(SbrConfig)]),
    "sbr_device_count": (C.c_int, []),
    "sbr_rows_thresholds": (C.c_int, [C.POINTER(SbrConfig), C.POINTER(C.c_double)]),
    "sbr_create": (C.c_int, [_I64, C.c_int, _I64, C.POINTER(SbrConfig), C.POINTER(_VP)]),
    "sbr_destroy": (C.c_int, [_VP]),
    "sbr_last_error": (C.c_char_p, [_VP]),
    "sbr_num_envs": (_I64, [_VP]),
    "sbr_query": (C.c_int, [_VP, _I32, C.POINTER(C.c_int64)]),
    "sbr_set_influent_tables": (C.c_int, [_VP, _VP, _VP]),
    "sbr_reset": (C.c_int, [_VP, _U64, _VP, _VP, _VP, _VP, _VP, _VP]),
    "sbr_reset_carry": (C.c_int, [_VP, _U64, _VP, _VP, _VP, _VP, _VP, _VP]),
    "sbr_set_trace": (C.c_int, [_VP, _VP, _I64, _I64, _I32]),
    "sbr_step": (C.c_int, [_VP, _VP, _VP, _VP, _VP, _VP, _VP]),
    "sbr_cycle_reset": (C.c_int, [_VP, _U64, _VP, _VP, _VP, _VP, _I32, _VP, _VP]),
    "sbr_cycle_step": (C.c_int, [_VP, _VP, _VP, _VP, _VP, _VP]),
    "sbr_rollout": (C.c_int, [_VP, _I32, _U64, _VP, _VP, _VP]),
    "sbr_reduce_stats": (C.c_int, [_VP, _VP, _I64, _VP, _VP]),
    "sbr_get_state": (C.c_int, [_VP, _VP, _VP, _VP]),
    "sbr_set_state": (C.c_int, [_VP, _VP, _VP, _VP]),
    "sbr_get_ctrl_row": (C.c_int, [_VP, _I32, _VP, _VP]),
    "sbr_get_influent": (C.c_int, [_VP, _VP, _VP]),
    "sbr_eval_rhs": (C.c_int, [_VP, _I32, _I64, _VP, _VP, _VP, _VP, _VP, _VP]),
    "sbr_eval_substeps": (C.c_int, [_VP, _I32, _I64, _I32, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP]),
    "sbr_draw_normals": (C.c_int, [_VP, _U64, _VP, _VP]),
    "sbr_draw_scenarios": (C.c_int, [_VP, _U64, _VP, _VP]),
    "sbr_synchronize": (C.c_int, [_VP, _VP]),
    "sbr_timer_start": (C.c_int, [_VP, _VP]),
    "sbr_timer_stop": (C.c_int, [_VP, _VP, C.POINTER(C.c_float)]),
}

_lib = None


def library_path():
    """In-tree libsbr_amd.so; SBR_AMD_LIB overrides it (kernel A/B experiments only, scripts/gpu_ab.py)."""
    return os.environ.get("SBR_AMD_LIB") or _build.LIB


def load(build_if_missing=True):
    """dlopen libsbr_amd.so.  Raises if it is missing and cannot be built: the product has no fallback."""
    global _lib
    if _lib is not None:
        return _lib
    path = library_path()
    if path != _build.LIB:
        build_if_missing = False
    if not os.path.exists(path) or (build_if_missing and _build.is_stale()):
        if not build_if_missing:
            raise SbrError("libsbr_amd.so is missing (%s); run python -c 'import __graft_entry__ as g; g.build()'" % path)
        _build.build_library()
    # PyTorch-ROCm ships its own libamdhip64 / libhsa-runtime64.  Whichever HIP runtime is mapped first serves the whole
    # process, and with this library first (system ROCm) and torch second the device enumeration of the later one fails
    # ("no HIP device visible", seen when build() and smoke() ran in one process).  torch owns the device memory this package
    # works on, so its runtime goes first - always the same order, the one every test exercises.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(path)
    for name, (res, args) in SYMBOLS.items():
        try:
            fn = getattr(lib, name)    # AttributeError if the .so does not export a declared symbol
        except AttributeError:
            if path == _build.LIB:
                raise
            continue                   # an A/B variant built from an older source tree may lack newer entry points
        fn.restype, fn.argtypes = res, args
    # every library is checked, also one selected through SBR_AMD_LIB (the A/B builds of scripts/): a variant built from older
    # sources would be handed structs and records of another layout.  A library without the symbol counts as a mismatch;
    # SBR_AMD_ALLOW_ABI_MISMATCH=1 is the explicit way to load one anyway (deliberate old-variant A/Bs).
    try:
        reported = int(lib.sbr_abi_version())
    except AttributeError:
        reported = None
    if reported != ABI_VERSION and os.environ.get("SBR_AMD_ALLOW_ABI_MISMATCH") != "1":
        raise SbrError("%s reports ABI version %s, this binding was written for %d: rebuild the library (python -c 'import "
                       "__graft_entry__ as g; g.build()'), or set SBR_AMD_ALLOW_ABI_MISMATCH=1 for a deliberate old-variant A/B"
                       % (os.path.basename(path), reported, ABI_VERSION))
    _lib = lib
    return lib


def default_config():
    cfg = SbrConfig()
    rc = load().sbr_default_config(C.byref(cfg))
    if rc != 0:
        raise SbrError("sbr_default_config failed: %d" % rc)
    return cfg


def check(rc, handle=None):
    if rc != 0:
        msg = load().sbr_last_error(handle)
        raise SbrError("libsbr_amd error %d: %s" % (rc, msg.decode() if msg else "?"))
