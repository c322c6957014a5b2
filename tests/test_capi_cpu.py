"""CPU-side checks of the C ABI: libsbr_amd.so loads without a GPU, exports every function include/sbr_amd.h
declares, its defaults are the reference's constants, and - without a device - it refuses loudly instead of
falling back to anything.  No compute call is made here."""
import ctypes as C
import os
import re

import numpy as np
import pytest
from conftest import ROOT, golden

from gym_sbr2_amd import _capi


def _declared_functions():
    text = open(os.path.join(ROOT, "include", "sbr_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(sbr_[a-z_0-9]+)\s*\(", text)))


def test_every_declared_symbol_is_exported_and_bound():
    lib = _capi.load()
    declared = _declared_functions()
    assert len(declared) >= 18
    assert declared == sorted(_capi.SYMBOLS), set(declared) ^ set(_capi.SYMBOLS)
    raw = C.CDLL(_capi.library_path())
    for name in declared:
        assert getattr(raw, name) is not None
    assert b"gfx950" in lib.sbr_version()


def test_header_constants_match_the_binding():
    text = open(os.path.join(ROOT, "include", "sbr_amd.h")).read()
    defs = {k: int(v) for k, v in re.findall(r"#define\s+(SBR_[A-Z_]+)\s+(\d+)\s", text)}
    assert (defs["SBR_NX"], defs["SBR_NOBS"], defs["SBR_NSTATE"], defs["SBR_NCTRL"], defs["SBR_KLA_HIST"]) == (
        _capi.NX, _capi.NOBS, _capi.NSTATE, _capi.NCTRL, _capi.KLA_HIST)
    assert (defs["SBR_ST_NEGATIVE"], defs["SBR_ST_NEAR_POLE"], defs["SBR_ST_NONFINITE"]) == (1, 2, 4)
    assert _capi.C_STATUS == 22 and _capi.C_KLA_SUM == 23 and _capi.C_PLAN == _capi.NCTRL - 1 == 24 and _capi.C_KLA_LAST == _capi.C_KLA_HIST0 + 9 == 17
    assert defs["SBR_PLAN_SLAVED"] == _capi.PLAN_SLAVED == 128 and defs["SBR_NTRACE"] == _capi.NTRACE and defs["SBR_ABI_VERSION"] == _capi.ABI_VERSION == 6
    tr = re.findall(r"SBR_TR_[A-Z_0-9]+", text.split("enum { SBR_TR_T")[1].split("};")[0])
    assert tr[-2:] == ["SBR_TR_PLAN", "SBR_TR_PLAN_FIRST"] and (_capi.TR_PLAN, _capi.TR_PLAN_FIRST) == (34, 35) == (_capi.NTRACE - 2, _capi.NTRACE - 1)
    q = re.findall(r"^\s+(SBR_Q_[A-Z_]+)", text.split("int sbr_query")[0].split("enum {")[-1], re.M)
    assert q == ["SBR_Q_" + n[2:] for n in ("Q_ONE_WAVE_ENVS", "Q_STEP_SMALL_BATCH_ENVS", "Q_STEP_BLOCK", "Q_STEP_WAVES", "Q_STEP_TWO_WAVES_ABOVE_ENVS",
                                           "Q_FUSED_ONE_WAVE_MAX_ENVS", "Q_ROLLOUT_WAVES", "Q_RESET_BLOCK", "Q_SCHEME")]
    assert [getattr(_capi, n[4:]) for n in q] == list(range(9))
    enum_names = re.findall(r"SBR_C_[A-Z_0-9]+", text.split("enum {")[1].split("};")[0])
    assert enum_names[:8] == ["SBR_C_T", "SBR_C_SO_M1", "SBR_C_SO_M2", "SBR_C_SNO_M1", "SBR_C_SNO_M2", "SBR_C_IE_DO", "SBR_C_IE_EC",
                              "SBR_C_EC_LAST"]


def test_default_config_is_the_reference():
    c = golden("constants")
    cfg = _capi.default_config()
    for key, val in [("T1_end", cfg.T_fill), ("T3_0", cfg.T3_0), ("T3_end", cfg.T3_end), ("T4_end", cfg.T4_end),
                     ("T5_end", cfg.T5_end), ("So_sat", cfg.So_sat), ("dt", cfg.dt), ("t_delta", cfg.t_delta),
                     ("t_cycle", cfg.t_cycle), ("EC_conc", cfg.EC_conc)]:
        assert float(c[key]) == val, key
    assert (cfg.Kla_min, cfg.Kla_max) == tuple(c["DO_control_par"][4:6])
    assert (cfg.EC_min, cfg.EC_max) == tuple(c["EC_control_par"][4:6])
    assert (cfg.t_settle, cfg.t_draw) == (float(c["t_ratio"][5]), float(c["t_ratio"][6]))
    k = golden("rhs_kat")
    assert [cfg.Ya, cfg.Yh, cfg.fp, cfg.ixb, cfg.ixp] == k["Spar"].tolist()
    assert [cfg.muH, cfg.Ks, cfg.Koh, cfg.Kno, cfg.bH, cfg.eta_g, cfg.eta_h, cfg.kh, cfg.Kx, cfg.muA, cfg.Knh, cfg.bA,
            cfg.Koa, cfg.ka] == k["Kpar"].tolist()
    assert np.array_equal(np.array(cfg.x0[:]), golden("sbros_const_2_5")["x0_init"])
    assert (cfg.substeps, cfg.terminal, cfg.out_f64, cfg.act_f64, cfg.reward_kind) == (10, 1, 0, 0, 0)
    # the oracle's parameter block has the same layout and the same defaults
    from oracle import sbr_oracle as O
    p = O.default_params()
    assert C.sizeof(p) == C.sizeof(cfg)
    for name, ctype in _capi.SbrConfig._fields_:
        a, b = getattr(p, name), getattr(cfg, name)
        if hasattr(a, "__len__"):
            assert list(a) == list(b), name
        elif name != "out_f64":          # the oracle always returns float64
            assert a == b, name
    assert list(cfg.t_ratio) == c["t_ratio"].tolist() and (cfg.cyc_Kc, cfg.cyc_tauI, cfg.cyc_tauD, cfg.cyc_dt) == (5.0, 0.00035, 0.005, 0.02 / 24)


def test_packaged_influent_tables_are_the_captured_ones():
    from gym_sbr2_amd.vec_env import load_influent_tables
    means, stds = load_influent_tables()
    t = golden("influent_tables")
    assert np.array_equal(means, t["means"]) and np.array_equal(stds, t["stds"]) and means.shape == (8, 14, 48)


def test_no_gpu_means_a_loud_error_not_a_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    lib = _capi.load()
    assert lib.sbr_device_count() == 0
    h = C.c_void_p()
    rc = lib.sbr_create(8, 0, 0, None, C.byref(h))
    assert rc == -2 and not h.value and b"no CPU path" in lib.sbr_last_error(None)
    import gym_sbr2_amd
    with pytest.raises(_capi.SbrError):
        gym_sbr2_amd.SbrOSVec(8)
    with pytest.raises(_capi.SbrError):
        gym_sbr2_amd.make("SBROS-v1")
    assert gym_sbr2_amd.registered_ids() == ["SBR-v2", "SBROS-v1"]


def test_product_never_imports_the_oracle():
    """The oracle is test infrastructure: nothing under gym_sbr2_amd/ may import, load or call it."""
    pkg = os.path.join(ROOT, "gym_sbr2_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert "sbr_oracle" not in src and "import oracle" not in src and "from oracle" not in src, f


def test_interval_row_count_thresholds_reproduce_the_ieee_quotient():
    """Host logic behind len(t_range) = int(((t + t_delta) - t)/dt) (gym_SBR_oneshot.py:1339, :1384): the kernels replace the
    division by two comparisons against thresholds found on the host.  For every span within +-200 ulp of 9 dt and 10 dt,
    and for every span the reference's own time recurrence produces over an episode, the classification equals the IEEE
    quotient truncated, and both row counts occur."""
    lib = _capi.load()
    cfg = _capi.default_config()
    thr = (C.c_double * 2)()
    assert lib.sbr_rows_thresholds(C.byref(cfg), thr) == 0 and lib.sbr_rows_thresholds(None, thr) == 0
    t9, t10 = thr[0], thr[1]
    dt, t_delta = cfg.dt, cfg.t_delta

    def rows_fast(span):
        return 10 if span >= t10 else (9 if span >= t9 else None)
    for centre in (9 * dt, 10 * dt):
        s = centre
        for _ in range(200):
            s = np.nextafter(s, 0.0)
        for _ in range(400):
            want = int(s / dt)
            got = rows_fast(s)
            assert got == want or (got is None and want < 9), (s, got, want)
            s = np.nextafter(s, 1.0)
    assert int(np.nextafter(t10, 0.0) / dt) == 9 and int(t10 / dt) == 10 and int(np.nextafter(t9, 0.0) / dt) == 8 and int(t9 / dt) == 9
    seen, t = set(), cfg.T_fill
    for _ in range(466):                     # the time recurrence of an episode: t += t_delta in float64
        t1 = t + t_delta
        span = t1 - t
        assert rows_fast(span) == int(span / dt)
        seen.add(rows_fast(span)); t = t1
    assert seen == {9, 10}
    e = golden("sbros_const_2_5")
    spans = e["iv_t_end"] - e["iv_t_start"]
    assert [rows_fast(s) for s in spans] == e["iv_n_rows"].tolist()      # the reference's own 466 intervals


def test_header_is_plain_c_and_the_c_demo_compiles_against_it(tmp_path):
    """The boundary is a C ABI: include/sbr_amd.h must be valid C99 on its own (no C++, no HIP types), and the plain-C caller
    examples/c_abi_demo.c must compile and link against the library with gcc.  (It runs on the GPU box:
    tests/test_gpu_parity.py::test_c_abi_from_plain_c_without_python_or_torch; here it must fail loudly for want of a device.)"""
    import shutil
    import subprocess
    if shutil.which("gcc") is None:
        pytest.skip("gcc not available")
    hdr = os.path.join(ROOT, "include", "sbr_amd.h")
    subprocess.check_call(["gcc", "-std=c99", "-pedantic", "-Wall", "-Werror", "-fsyntax-only", "-x", "c", hdr])
    if not os.path.exists("/opt/rocm/include/hip/hip_runtime_api.h"):
        pytest.skip("HIP runtime headers not available")
    from gym_sbr2_amd import _capi, build as B
    _capi.load()
    exe, libdir = str(tmp_path / "c_abi_demo"), os.path.dirname(B.LIB)
    subprocess.check_call(["gcc", "-std=c99", "-O1", "-Wall", "-D__HIP_PLATFORM_AMD__", "-I", os.path.join(ROOT, "include"), "-I", "/opt/rocm/include",
                           os.path.join(ROOT, "examples", "c_abi_demo.c"), "-L", libdir, "-lsbr_amd", "-L", "/opt/rocm/lib", "-lamdhip64",
                           "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib", "-o", exe])
    import torch
    if not torch.cuda.is_available():
        p = subprocess.run([exe], capture_output=True, text=True, timeout=60)
        assert p.returncode != 0 and "no CPU path" in p.stderr
