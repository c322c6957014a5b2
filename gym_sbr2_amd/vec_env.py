"""Batched SBROS-v1 environment on one MI355X: the host-side mirror of the reference's SbrOS
(gym_SBR/envs/gym_SBR_oneshot.py:99) for N independent reactors.

PyTorch is used for device memory and streams only; every number comes out of libsbr_amd.so
(HIP kernels) through the C ABI of include/sbr_amd.h.  There is no CPU path.
"""
import ctypes as C
import os

import numpy as np
import torch

from . import _capi

_DATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "influent_tables.npz")


def load_influent_tables():
    """The eight influent scenarios of buffer_tank3.py:18-1197 as data: means, stds [8][14][48]
    (13 concentrations + flow q; captured from the reference by oracle/gen_golden.py)."""
    t = np.load(_DATA)
    return np.ascontiguousarray(t["means"], dtype=np.float64), np.ascontiguousarray(t["stds"], dtype=np.float64)


def _ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


# the handle of torch's current stream on a device: torch's own fast accessor (an int, no Stream object) where it exists
_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None) or (lambda idx: torch.cuda.current_stream(idx).cuda_stream)


class SbrOSVec:
    """N SBROS-v1 environments on one GPU.

    reset()  -> obs [N,18]                         (SbrOS.reset, :168-438; obs_DO[9] ++ obs_EC[9])
    step(a)  -> obs [N,18], state [N,15], reward [N], done [N] uint8   (SbrOS.step, :843-1273)
    action a: [N,2] float32 (or float64 with action_dtype=torch.float64, what the reference's step() receives)
    = (DO set-point, NO3 set-point), clipped to [0,8] x [0,15] as in the reference.
    """

    def __init__(self, num_envs, device=0, first_env_id=0, out_dtype=torch.float32, config=None, tables=None,
                 action_dtype=torch.float32, reward=None, random_scenario=None):
        if not torch.cuda.is_available():
            raise _capi.SbrError("SbrOSVec needs a HIP device (torch.cuda.is_available() is False); "
                                 "this package has no CPU fallback")
        self.lib = _capi.load()
        self.num_envs = int(num_envs)
        self.device = torch.device("cuda", device if isinstance(device, int) else torch.device(device).index or 0)
        self.first_env_id = int(first_env_id)
        self.cfg = config if config is not None else _capi.default_config()
        if reward is not None:                # "eqi_oci" (the reference's SBROS-v1 reward) | "g2anet" | "oci"
            if reward not in _capi.REWARD_KINDS:
                raise ValueError("reward must be one of %s" % sorted(_capi.REWARD_KINDS))
            self.cfg.reward_kind = _capi.REWARD_KINDS[reward]
        if random_scenario is not None:       # reset(scenario=None) draws one of the 8 scenarios per env (SbrEnv4, gym_SBR_env4.py:107)
            self.cfg.random_scenario = 1 if random_scenario else 0
        if out_dtype not in (torch.float32, torch.float64):
            raise ValueError("out_dtype must be torch.float32 or torch.float64")
        self.out_dtype = out_dtype
        self.cfg.out_f64 = 1 if out_dtype == torch.float64 else 0
        if action_dtype not in (torch.float32, torch.float64):
            raise ValueError("action_dtype must be torch.float32 or torch.float64")
        self.action_dtype = action_dtype
        self.cfg.act_f64 = 1 if action_dtype == torch.float64 else 0
        self._h = C.c_void_p()
        _capi.check(self.lib.sbr_create(self.num_envs, self.device.index, self.first_env_id, C.byref(self.cfg),
                                        C.byref(self._h)))
        means, stds = tables if tables is not None else load_influent_tables()
        means = np.ascontiguousarray(means, dtype=np.float64)
        stds = np.ascontiguousarray(stds, dtype=np.float64)
        assert means.shape == stds.shape == (_capi.NSCEN, _capi.NSERIES, _capi.NSAMP)
        _capi.check(self.lib.sbr_set_influent_tables(self._h, means.ctypes.data_as(C.c_void_p),
                                                     stds.ctypes.data_as(C.c_void_p)), self._h)
        n, dev = self.num_envs, self.device
        self._ashape = (n, 2)
        self._sbr_step = self.lib.sbr_step
        self._outs = [None] * 4
        self._step_out = None
        self.obs = torch.empty((n, _capi.NOBS), dtype=out_dtype, device=dev)
        self.state = torch.empty((n, _capi.NSTATE), dtype=out_dtype, device=dev)
        self.reward = torch.empty((n,), dtype=out_dtype, device=dev)
        self.done = torch.empty((n,), dtype=torch.uint8, device=dev)

    # The output buffers step() and reset() write.  They may be replaced by the caller (same shape, dtype and device, e.g. a
    # slice of a rollout buffer); their addresses are converted once here, so that the host side of a step is ~5 us of Python
    # (scripts/gpu_host_overhead.py).
    def _set_out(self, k, t):
        shape, dtype = (((self.num_envs, _capi.NOBS), self.out_dtype), ((self.num_envs, _capi.NSTATE), self.out_dtype),
                        ((self.num_envs,), self.out_dtype), ((self.num_envs,), torch.uint8))[k]
        if not (isinstance(t, torch.Tensor) and t.shape == shape and t.dtype == dtype and t.device == self.device
                and t.is_contiguous()):
            raise ValueError("output buffer %d must be a contiguous %s tensor of shape %s on %s" % (k, dtype, shape, self.device))
        self._outs[k] = t
        if all(o is not None for o in self._outs):
            self._step_out = tuple(C.c_void_p(o.data_ptr()) for o in self._outs)

    obs = property(lambda self: self._outs[0], lambda self, t: self._set_out(0, t))
    state = property(lambda self: self._outs[1], lambda self, t: self._set_out(1, t))
    reward = property(lambda self: self._outs[2], lambda self, t: self._set_out(2, t))
    done = property(lambda self: self._outs[3], lambda self, t: self._set_out(3, t))

    # ------------------------------------------------------------------ plumbing
    def _stream(self):
        return C.c_void_p(_raw_stream(self.device.index))

    def _dev(self, a, dtype, shape):
        if a is None:
            return None
        t = torch.as_tensor(a, dtype=dtype, device=self.device).contiguous()
        if tuple(t.shape) != tuple(shape):
            raise ValueError("expected shape %s, got %s" % (tuple(shape), tuple(t.shape)))
        return t

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            torch.cuda.synchronize(self.device)
            self.lib.sbr_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------------ the gym-like surface
    def reset(self, seed=0, scenario=None, rnd=None, influent=None, mask=None, carry_over=False):
        """carry_over=True starts the new cycle from each env's own current state (multi-cycle operation: x0 := x,
        IV := x[0]; disabled in the reference, gym_SBR_oneshot.py:260-268) instead of the configured start state."""
        n = self.num_envs
        sc = self._dev(scenario, torch.int32, (n,))
        rn = self._dev(rnd, torch.float64, (n, _capi.NSAMP))
        inf = self._dev(influent, torch.float64, (n, _capi.NX))
        mk = self._dev(mask, torch.uint8, (n,))
        with torch.cuda.device(self.device):
            fn = self.lib.sbr_reset_carry if carry_over else self.lib.sbr_reset
            _capi.check(fn(self._h, C.c_uint64(int(seed)), _ptr(sc), _ptr(rn), _ptr(inf), _ptr(mk), _ptr(self.obs),
                           self._stream()), self._h)
        self._keep = (sc, rn, inf, mk)      # keep inputs alive until the stream has consumed them
        return self.obs

    def step(self, action):
        a = action if (isinstance(action, torch.Tensor) and action.dtype == self.action_dtype and action.is_contiguous()
                       and action.device == self.device) else self._dev(action, self.action_dtype, self._ashape)
        if a.shape != self._ashape:
            raise ValueError("action must have shape [N,2]")
        o, s, r, d = self._step_out
        rc = self._sbr_step(self._h, a.data_ptr(), o, s, r, d, _raw_stream(self.device.index))
        if rc:
            _capi.check(rc, self._h)
        self._keep_a = a
        return tuple(self._outs)

    def enable_host_io(self):
        """Small batches driven from the host (the reference-shaped single env): allocate PINNED host buffers for the action
        and for obs/state/reward/done and let the kernel read and write them directly over PCIe (pinned host memory is
        device-accessible), so that a step costs one launch and one stream synchronisation instead of one host-to-device
        and four device-to-host copies.  Returns the numpy views that step_host() fills."""
        np_out = {torch.float32: np.float32, torch.float64: np.float64}
        n = self.num_envs
        self._h_act = torch.empty((n, 2), dtype=self.action_dtype).pin_memory()
        self._h_obs = torch.empty((n, _capi.NOBS), dtype=self.out_dtype).pin_memory()
        self._h_state = torch.empty((n, _capi.NSTATE), dtype=self.out_dtype).pin_memory()
        self._h_reward = torch.empty((n,), dtype=self.out_dtype).pin_memory()
        self._h_done = torch.empty((n,), dtype=torch.uint8).pin_memory()
        self._h_views = (self._h_act.numpy(), self._h_obs.numpy(), self._h_state.numpy(), self._h_reward.numpy(),
                         self._h_done.numpy())
        self._h_ptrs = tuple(C.c_void_p(t.data_ptr()) for t in (self._h_act, self._h_obs, self._h_state, self._h_reward, self._h_done))
        self._sbr_sync = self.lib.sbr_synchronize
        return self._h_views

    def step_host(self, action):
        """step() through the pinned host buffers of enable_host_io(): action is array-like [N,2]; returns numpy views of
        obs, state, reward, done, valid until the next call (the stream has been synchronised).  Two C calls - sbr_step and
        sbr_synchronize - with every pointer converted once in enable_host_io()."""
        self._h_views[0][...] = action
        st = _raw_stream(self.device.index)
        rc = self._sbr_step(self._h, *self._h_ptrs, st) or self._sbr_sync(self._h, st)
        if rc:
            _capi.check(rc, self._h)
        return self._h_views[1:]

    def capture_steps(self, actions):
        """Capture one step() per action tensor of `actions` (a sequence of [N,2] device tensors at fixed addresses) into a
        HIP graph and return it; graph.replay() then issues all of them with one host call (0.4 us per step instead of
        ~8 us through Python).  obs/state/reward/done hold the outputs of the LAST captured step after a replay.  Nothing in
        sbr_step allocates or synchronises, which is what makes it capturable."""
        g = torch.cuda.CUDAGraph()
        side = torch.cuda.Stream(device=self.device)
        side.wait_stream(torch.cuda.current_stream(self.device))
        # capture_error_mode "thread_local": only THIS thread's calls are checked against the capture.  With a process group up,
        # RCCL's watchdog thread polls its events from another thread, which the default ("global") mode may take for an illegal call
        # during capture and invalidate the graph - in a multi-rank run, of all places.  sbr_step itself makes no such call.
        with torch.cuda.stream(side):
            with torch.cuda.graph(g, stream=side, capture_error_mode="thread_local"):
                for a in actions:
                    self.step(a)
        torch.cuda.current_stream(self.device).wait_stream(side)
        return g

    def rollout(self, n_steps, policy_seed=0, return_actions=False):
        """n_steps fused step() calls per env with the on-device uniform random policy; returns the
        per-env sum of rewards [N] float64 (and the sampled actions [n_steps,N,2] float32)."""
        ret = torch.empty((self.num_envs,), dtype=torch.float64, device=self.device)
        acts = (torch.empty((int(n_steps), self.num_envs, 2), dtype=torch.float32, device=self.device)
                if return_actions else None)
        _capi.check(self.lib.sbr_rollout(self._h, int(n_steps), C.c_uint64(int(policy_seed)), _ptr(ret), _ptr(acts),
                                         self._stream()), self._h)
        return (ret, acts) if return_actions else ret

    def enable_trace(self, n_envs=1, capacity=463):
        """Trajectory export: every step() appends one record (_capi.TR_*: t, x(14), Kla, EC, reward, done, the set-points in
        force, the NO3-PID's e/ie/dcv and the four reward diagnostics) for the first n_envs envs at index = calls since
        reset.  Returns the buffer [capacity, NTRACE, n_envs] float64 (NaN where nothing was written)."""
        self._trace = torch.full((int(capacity), _capi.NTRACE, int(n_envs)), float("nan"), dtype=torch.float64,
                                 device=self.device)
        _capi.check(self.lib.sbr_set_trace(self._h, _ptr(self._trace), int(n_envs), int(capacity), _capi.NTRACE), self._h)
        return self._trace

    def disable_trace(self):
        _capi.check(self.lib.sbr_set_trace(self._h, None, 0, 0, _capi.NTRACE), self._h)
        self._trace = None

    # ------------------------------------------------------------------ inspection / parity injection
    def get_state(self):
        x = torch.empty((_capi.NX, self.num_envs), dtype=torch.float64, device=self.device)
        ctrl = torch.empty((_capi.NCTRL, self.num_envs), dtype=torch.float64, device=self.device)
        _capi.check(self.lib.sbr_get_state(self._h, _ptr(x), _ptr(ctrl), self._stream()), self._h)
        return x, ctrl

    def set_state(self, x=None, ctrl=None):
        x = self._dev(x, torch.float64, (_capi.NX, self.num_envs))
        ctrl = self._dev(ctrl, torch.float64, (_capi.NCTRL, self.num_envs))
        _capi.check(self.lib.sbr_set_state(self._h, _ptr(x), _ptr(ctrl), self._stream()), self._h)
        torch.cuda.current_stream(self.device).synchronize()

    def influent(self):
        out = torch.empty((_capi.NX, self.num_envs), dtype=torch.float64, device=self.device)
        _capi.check(self.lib.sbr_get_influent(self._h, _ptr(out), self._stream()), self._h)
        return out

    def query(self, what):
        """One of the library's own decisions for this handle (sbr_query, _capi.Q_*): launch shapes and their thresholds, which
        depend on the device's CU count."""
        out = C.c_int64()
        _capi.check(self.lib.sbr_query(self._h, int(what), C.byref(out)), self._h)
        return int(out.value)

    def plan(self, out=None):
        """What cfg.scheme = 1 did in each env's last control interval ([N] int64): Butcher-5 step count (& 127) and
        _capi.PLAN_SLAVED if dissolved oxygen was held.  0 = nothing to report (cfg.scheme 0, or no step() since the reset)."""
        return self.ctrl_row(_capi.C_PLAN, out).to(torch.int64)

    def status(self):
        """Sticky domain-of-validity bits per env (int64; _capi.ST_NEGATIVE | ST_NEAR_POLE | ST_NONFINITE): the
        reference model has no guards and can be driven to negative ammonia / a Monod pole by aggressive policies."""
        return self.ctrl_row(_capi.C_STATUS).to(torch.int64)

    def ctrl_row(self, row, out=None):
        """One row of the controller/bookkeeping block ([N] float64), copied on the current stream without a host sync."""
        if out is None:
            out = torch.empty((self.num_envs,), dtype=torch.float64, device=self.device)
        _capi.check(self.lib.sbr_get_ctrl_row(self._h, int(row), _ptr(out), self._stream()), self._h)
        return out

    def episode_returns(self, out=None):
        """Sum of rewards since reset, per env ([N] float64)."""
        return self.ctrl_row(_capi.C_RETURN, out)

    def stats(self, values):
        v = self._dev(values, torch.float64, (values.numel(),))
        out = torch.empty((4,), dtype=torch.float64, device=self.device)
        _capi.check(self.lib.sbr_reduce_stats(self._h, _ptr(v), v.numel(), _ptr(out), self._stream()), self._h)
        s, mn, mx, cnt = out.tolist()
        return {"sum": s, "min": mn, "max": mx, "count": cnt, "mean": s / cnt if cnt else float("nan")}

    def eval_rhs(self, kind, x, kla, ec, loading=None):
        x = torch.as_tensor(x, dtype=torch.float64, device=self.device).contiguous()
        n = x.shape[0]
        kla = self._dev(kla, torch.float64, (n,))
        ec = self._dev(ec, torch.float64, (n,))
        ld = self._dev(loading, torch.float64, (n, _capi.NX))
        dx = torch.empty_like(x)
        _capi.check(self.lib.sbr_eval_rhs(self._h, int(kind), n, _ptr(x), _ptr(kla), _ptr(ec), _ptr(ld), _ptr(dx),
                                          self._stream()), self._h)
        return dx

    def eval_substeps(self, x0, kla, h, ec=None, loading=None, kind=0, n_sub=None):
        """RK4 nodes and node slopes of n independent integration spans (sbr_eval_substeps): x0 [n,14], kla / h [n] (h = substep
        length) -> xs, dxs [n, n_sub + 1, 14] float64.  kind 0 control interval (ec [n]; n_sub defaults to cfg.substeps), 1 fill
        (loading [n,14]), 2 idle, 3 settle + draw, then idle.  For trajectory export (SbrOS.trajectory(dense=True))."""
        x0 = torch.as_tensor(x0, dtype=torch.float64, device=self.device).contiguous()
        n = x0.shape[0]
        n_sub = int(self.cfg.substeps if n_sub is None else n_sub)
        kla, h = self._dev(kla, torch.float64, (n,)), self._dev(h, torch.float64, (n,))
        ec = self._dev(ec, torch.float64, (n,))
        ld = self._dev(loading, torch.float64, (n, _capi.NX))
        xs = torch.empty((n, n_sub + 1, _capi.NX), dtype=torch.float64, device=self.device)
        dxs = torch.empty_like(xs)
        _capi.check(self.lib.sbr_eval_substeps(self._h, int(kind), n, n_sub, _ptr(x0), _ptr(kla), _ptr(ec), _ptr(ld), _ptr(h),
                                               _ptr(xs), _ptr(dxs), self._stream()), self._h)
        return xs, dxs

    def draw_scenarios(self, seed):
        """The scenario every env draws at reset(seed, scenario=None) when the config has random_scenario = 1 ([N] int32)."""
        out = torch.empty((self.num_envs,), dtype=torch.int32, device=self.device)
        _capi.check(self.lib.sbr_draw_scenarios(self._h, C.c_uint64(int(seed)), _ptr(out), self._stream()), self._h)
        return out

    def draw_normals(self, seed):
        out = torch.empty((self.num_envs, _capi.NSAMP), dtype=torch.float64, device=self.device)
        _capi.check(self.lib.sbr_draw_normals(self._h, C.c_uint64(int(seed)), _ptr(out), self._stream()), self._h)
        return out

    # ------------------------------------------------------------------ device timing (bench.py)
    def timer_start(self):
        _capi.check(self.lib.sbr_timer_start(self._h, self._stream()), self._h)

    def timer_stop(self):
        ms = C.c_float()
        _capi.check(self.lib.sbr_timer_stop(self._h, self._stream(), C.byref(ms)), self._h)
        return float(ms.value)
