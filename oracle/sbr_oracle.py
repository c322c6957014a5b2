"""ctypes front-end of oracle/sbr_oracle.c (TEST INFRASTRUCTURE - CPU oracle, layer 2).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "libsbr_oracle.so")
NX, NOBS, NSTATE, KLA_HIST = 14, 18, 15, 10


class Params(C.Structure):
    _fields_ = [(n, C.c_double) for n in (
        "Ya Yh fp ixb ixp muH Ks Koh Kno bH eta_g eta_h kh Kx muA Knh bA Koa ka "
        "WV IV dt t_delta t_cycle T_fill T3_0 T3_end T4_end T5_end t_settle t_draw "
        "So_sat Kla_min Kla_max Kc_DO tauI_DO tauD_DO EC_min EC_max Kc_EC tauI_EC tauD_EC EC_conc "
        "act_DO_max act_EC_max biomass_setpoint Qeff settler_area settler_vmax").split()] + [
        ("t_ratio", C.c_double * 8), ("cyc_Kc", C.c_double), ("cyc_tauI", C.c_double), ("cyc_tauD", C.c_double),
        ("cyc_dt", C.c_double),
        ("x0", C.c_double * NX), ("substeps", C.c_int32), ("out_f64", C.c_int32),
        ("terminal", C.c_int32), ("reward_kind", C.c_int32), ("act_f64", C.c_int32), ("random_scenario", C.c_int32),
        ("scheme", C.c_int32), ("reserved_", C.c_int32)]


class Env(C.Structure):
    _fields_ = [("x", C.c_double * NX), ("t", C.c_double),
                ("so_m1", C.c_double), ("so_m2", C.c_double), ("sno_m1", C.c_double), ("sno_m2", C.c_double),
                ("ie_do", C.c_double), ("ie_ec", C.c_double),
                ("kla_last", C.c_double), ("ec_last", C.c_double), ("ec_prev", C.c_double),
                ("u_do", C.c_double), ("u_ec", C.c_double),
                ("kla_hist", C.c_double * KLA_HIST),
                ("qw", C.c_double), ("ret", C.c_double), ("steps", C.c_double), ("done", C.c_double),
                ("status", C.c_double), ("kla_sum", C.c_double), ("influent", C.c_double * NX), ("x_start", C.c_double * NX), ("span", C.c_double),
                ("n_rows", C.c_int32), ("n_intervals", C.c_int32), ("scheme_steps", C.c_int32), ("scheme_plan", C.c_int32)]


ENV_DTYPE = np.dtype([("x", "f8", NX), ("t", "f8"), ("so_m1", "f8"), ("so_m2", "f8"), ("sno_m1", "f8"),
                      ("sno_m2", "f8"), ("ie_do", "f8"), ("ie_ec", "f8"), ("kla_last", "f8"), ("ec_last", "f8"),
                      ("ec_prev", "f8"), ("u_do", "f8"), ("u_ec", "f8"), ("kla_hist", "f8", KLA_HIST),
                      ("qw", "f8"), ("ret", "f8"), ("steps", "f8"), ("done", "f8"), ("status", "f8"), ("kla_sum", "f8"),
                      ("influent", "f8", NX),
                      ("x_start", "f8", NX), ("span", "f8"), ("n_rows", "i4"), ("n_intervals", "i4"), ("scheme_steps", "i4"),
                      ("scheme_plan", "i4")], align=True)


def _src_hash():
    import hashlib
    h = hashlib.sha256()
    for name in ("sbr_oracle.c", "Makefile"):
        with open(os.path.join(_HERE, name), "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def build(force=False):
    """Stale = the library was built from other sources (content hash next to it; mtimes do not survive a copied tree).
    Serialised by a file lock so that several processes can call this at once."""
    import fcntl
    tag = _LIB + ".srchash"

    def stale():
        return not (os.path.exists(_LIB) and os.path.exists(tag) and open(tag).read().strip() == _src_hash())
    if force or stale():
        with open(_LIB + ".lock", "w") as lock:
            fcntl.flock(lock, fcntl.LOCK_EX)
            if force or stale():
                subprocess.check_call(["make", "-C", _HERE, "-s", "-B", "libsbr_oracle.so"])
                with open(tag, "w") as f:
                    f.write(_src_hash() + "\n")
    return _LIB


_lib = None
_variant = ""          # "" = strict build (no FMA contraction); "_fma" = same source with -mfma -ffp-contract=fast


def use_variant(suffix):
    """Switch to another build of the SAME source (tests/test_oracle_golden.py uses "_fma" as a control for how
    far pure rounding differences get amplified by the closed loop).  Returns the previous variant."""
    global _lib, _variant
    prev = _variant
    if suffix != _variant:
        if suffix:
            subprocess.check_call(["make", "-C", _HERE, "-s", "libsbr_oracle%s.so" % suffix])
        _lib, _variant = None, suffix
    return prev


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build() if not _variant else os.path.join(_HERE, "libsbr_oracle%s.so" % _variant))
        assert _lib.sbro_sizeof_env() == C.sizeof(Env) == ENV_DTYPE.itemsize, "oracle env layout drifted"
        assert _lib.sbro_sizeof_params() == C.sizeof(Params), "oracle params layout drifted"
    return _lib


def default_params(scheme=None):
    """The reference's constants; scheme as the product's sbr_default_config() (1 = adaptive Butcher-5) unless given
    (0 = RK4 x substeps)."""
    p = Params()
    lib().sbro_default_params(C.byref(p))
    if scheme is not None:
        p.scheme = int(scheme)
    return p


def _p(a, t=C.c_double):
    return None if a is None else a.ctypes.data_as(C.POINTER(t))


class OracleBatch:
    """N environments stepped by the C oracle (RK4, fp64, OpenMP over envs)."""

    def __init__(self, n, params=None, nthreads=1, first_env_id=0):
        self.n, self.nthreads, self.first_env_id = int(n), int(nthreads), int(first_env_id)
        self.p = params if params is not None else default_params()
        self.envs = np.zeros(self.n, dtype=ENV_DTYPE)

    def _envp(self):
        return self.envs.ctypes.data_as(C.POINTER(Env))

    def mix(self, means, stds, scenario, rnd):
        """influent_mixed [n][14] from tables[8,14,48], scenario [n], rnd [n][48]."""
        out = np.empty((self.n, NX))
        means = np.ascontiguousarray(means, dtype=np.float64)
        stds = np.ascontiguousarray(stds, dtype=np.float64)
        rnd = np.ascontiguousarray(rnd, dtype=np.float64)
        for i in range(self.n):
            s = int(scenario[i])
            lib().sbro_influent_mix(_p(means[s]), _p(stds[s]), _p(rnd[i]), _p(out[i]))
        return out

    def normals(self, seed):
        out = np.empty((self.n, 48))
        for i in range(self.n):
            lib().sbro_draw_normals(C.c_uint64(seed), C.c_uint64(self.first_env_id + i), _p(out[i]))
        return out

    def scenarios(self, seed):
        """The scenario each env draws at a reset with cfg.random_scenario = 1 (np.random.choice(8, 1), gym_SBR_env4.py:107)."""
        f = lib().sbro_scenario_draw
        f.restype = C.c_int32
        return np.array([f(C.c_uint64(seed), C.c_uint64(self.first_env_id + i)) for i in range(self.n)], dtype=np.int32)

    def reward_parts(self):
        """[n][4]: EQI2, OCI2, AE_OCI2, EC_OCI2 of the call just made (module_reward_EQIOCI.py:109-112)."""
        out = np.empty((self.n, 4))
        envs = self._envp()
        for i in range(self.n):
            lib().sbro_reward_parts(C.byref(self.p), C.byref(envs[i]), _p(out[i]))
        return out

    def load_state(self, x, ctrl):
        """Overwrite the plant/controller state from the product's PUBLIC layout: x [14][n], ctrl [24][n]
        (rows as in include/sbr_amd.h).  Used to re-synchronise the oracle to the device before a call.  The set-points
        in force and the EC before EC[-1] are temporaries of one call (every interval overwrites them first)."""
        x, ctrl = np.asarray(x, dtype=np.float64), np.asarray(ctrl, dtype=np.float64)
        e = self.envs
        e["x"] = x.T
        for row, name in enumerate(["t", "so_m1", "so_m2", "sno_m1", "sno_m2", "ie_do", "ie_ec", "ec_last"]):
            e[name] = ctrl[row]
        e["ec_prev"] = ctrl[7]
        e["kla_hist"] = ctrl[8:18].T
        e["kla_last"] = ctrl[17]
        e["qw"], e["ret"], e["steps"], e["done"], e["status"] = ctrl[18], ctrl[19], ctrl[20], ctrl[21], ctrl[22]
        e["kla_sum"] = ctrl[23]

    def reset(self, influent):
        influent = np.ascontiguousarray(np.broadcast_to(influent, (self.n, NX)), dtype=np.float64)
        obs = np.empty((self.n, NOBS))
        lib().sbro_batch_reset(C.byref(self.p), C.c_int64(self.n), self._envp(), _p(influent), _p(obs),
                               C.c_int(self.nthreads))
        return obs

    def reset_carry(self, influent):
        """New cycle from each env's own current state (x0 := x, IV := x[0])."""
        influent = np.ascontiguousarray(np.broadcast_to(influent, (self.n, NX)), dtype=np.float64)
        obs = np.empty((self.n, NOBS))
        lib().sbro_batch_reset_carry(C.byref(self.p), C.c_int64(self.n), self._envp(), _p(influent), _p(obs),
                                     C.c_int(self.nthreads))
        return obs

    def step(self, action, want_obs=True):
        # float64 actions, like the reference; to mirror the product (float32 action tensors) pass
        # actions.astype(np.float32) - the values are then used exactly
        action = np.ascontiguousarray(np.broadcast_to(action, (self.n, 2)), dtype=np.float64)
        obs = np.empty((self.n, NOBS)) if want_obs else None
        state = np.empty((self.n, NSTATE)) if want_obs else None
        reward = np.empty(self.n)
        done = np.empty(self.n, dtype=np.uint8)
        lib().sbro_batch_step(C.byref(self.p), C.c_int64(self.n), self._envp(), _p(action), _p(obs),
                              _p(state), _p(reward), _p(done, C.c_uint8), C.c_int(self.nthreads))
        return obs, state, reward, done

    def rollout(self, n_steps, policy_seed):
        ret = np.empty(self.n)
        lib().sbro_batch_rollout(C.byref(self.p), C.c_int64(self.n), self._envp(), C.c_int64(self.first_env_id),
                                 C.c_int32(n_steps), C.c_uint64(policy_seed), _p(ret), C.c_int(self.nthreads))
        return ret

    def policy_actions(self, n_steps, policy_seed):
        """actions [n_steps][n][2] of the on-device random policy, for calls 0..n_steps-1."""
        out = np.empty((n_steps, self.n, 2), dtype=np.float32)
        a = (C.c_float * 2)()
        for s in range(n_steps):
            for i in range(self.n):
                lib().sbro_policy_action(C.byref(self.p), C.c_uint64(policy_seed),
                                         C.c_uint64(self.first_env_id + i), C.c_uint32(s), a)
                out[s, i] = a[0], a[1]
        return out


def eval_rhs(kind, x, kla, ec, loading=None, params=None):
    p = params if params is not None else default_params()
    x = np.ascontiguousarray(x, dtype=np.float64)
    n = len(x)
    dx = np.empty_like(x)
    kla = np.ascontiguousarray(kla, dtype=np.float64)
    ec = np.ascontiguousarray(ec, dtype=np.float64)
    ld = None if loading is None else np.ascontiguousarray(loading, dtype=np.float64)
    lib().sbro_eval_rhs(C.byref(p), C.c_int(kind), C.c_int64(n), _p(x), _p(kla), _p(ec), _p(ld), _p(dx))
    return dx


def rk4(kind, x, span, n, kla, ec=0.0, loading=None, params=None):
    p = params if params is not None else default_params(scheme=0)
    x = np.array(x, dtype=np.float64)
    ld = None if loading is None else np.ascontiguousarray(loading, dtype=np.float64)
    lib().sbro_rk4(C.byref(p), C.c_int(kind), _p(x), C.c_double(span), C.c_int(n), C.c_double(kla), C.c_double(ec),
                   _p(ld))
    return x


def reaction_interval(x, span, kla, ec=0.0, params=None, scheme=1):
    """One reaction interval by the scheme-aware integrator of the C oracle (scheme 1: the adaptive Butcher-5 of round 5).
    Returns (x_end, step count; 0 = fell back to RK4, -1 = scheme 0)."""
    p = params if params is not None else default_params()
    p.scheme = scheme
    x = np.array(x, dtype=np.float64)
    fn = lib().sbro_reaction_interval
    fn.restype = C.c_int
    n = fn(C.byref(p), _p(x), C.c_double(span), C.c_double(kla), C.c_double(ec))
    return x, int(n)


NCYC_DIAG = 12   # qw, EQI, OCI, Ntot, COD, Snh, BOD5, Sno (effluent), mean Kla of phases 3, 5, 8, Xf


class OracleCycleBatch:
    """N per-cycle (`SBR-v2`) environments stepped by the C oracle: one step() = one whole 12 h cycle."""

    def __init__(self, n, params=None, nthreads=1):
        self.n, self.nthreads = int(n), int(nthreads)
        self.p = params if params is not None else default_params()
        self.x = np.tile(np.array(self.p.x0[:], dtype=np.float64), (self.n, 1))
        self.influent = np.zeros((self.n, NX))

    def reset(self, influent, carry_over=False):
        self.influent = np.ascontiguousarray(np.broadcast_to(influent, (self.n, NX)), dtype=np.float64).copy()
        if not carry_over:
            self.x = np.tile(np.array(self.p.x0[:], dtype=np.float64), (self.n, 1))
        st = np.empty((self.n, 3))
        for i in range(self.n):
            lib().sbro_cycle_reset_state(_p(self.x[i]), _p(self.influent[i]), _p(st[i]))
        return st

    def step(self, action):
        action = np.ascontiguousarray(np.broadcast_to(action, (self.n, 3)), dtype=np.float64)
        st, rew, diag = np.empty((self.n, 3)), np.empty(self.n), np.empty((self.n, NCYC_DIAG))
        self.x = np.ascontiguousarray(self.x)
        lib().sbro_batch_cycle_step(C.byref(self.p), C.c_int64(self.n), _p(self.x), _p(self.influent), _p(action), _p(st),
                                    _p(rew), _p(diag), C.c_int(self.nthreads))
        return st, rew, diag

    def step_logged(self, i, action):
        """One env, with the per-interval Kla of the six PID phases (6 x 256, NaN-padded)."""
        log = np.full((6, 256), np.nan)
        st, rew, diag = np.empty(3), np.empty(1), np.empty(NCYC_DIAG)
        a = np.ascontiguousarray(action, dtype=np.float64)
        x = np.ascontiguousarray(self.x[i]).copy()
        lib().sbro_cycle_step(C.byref(self.p), _p(x), _p(np.ascontiguousarray(self.influent[i])), _p(a), _p(st), _p(rew),
                              _p(diag), _p(log))
        self.x[i] = x
        return st, float(rew[0]), diag, log
