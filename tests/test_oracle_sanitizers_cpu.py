"""The C oracle under UndefinedBehaviorSanitizer: every GPU parity bound rests on oracle/sbr_oracle.c, so
the checker itself is run once with `-fsanitize=undefined,float-cast-overflow,bounds -fno-sanitize-recover=all` over a whole
golden SBROS-v1 episode (reset, 463 calls incl. the three double steps, the dosing intervals in scaled-mass variables, settle /
draw / idle), a fused rollout, the carry-over reset and two SBR-v2 cycles.  Any undefined behaviour aborts the child process;
the results must equal the regular build's bit for bit (same source, no FMA contraction in either)."""
import os
import shutil
import subprocess
import sys

import pytest
from conftest import ROOT

CHILD = r'''
import sys
sys.path.insert(0, %r)
import numpy as np
from oracle import sbr_oracle as O
from gym_sbr2_amd.vec_env import load_influent_tables
variant = sys.argv[1]
O.use_variant(variant)
means, stds = load_influent_tables()
e = np.load(%r)
n = 3
b = O.OracleBatch(n, nthreads=1, first_env_id=5)
rnd = np.stack([e["rnd"], np.zeros(48), b.normals(7)[2]])
infl = b.mix(means, stds, np.array([6, 0, 3], dtype=np.int32), rnd)
out = [b.reset(infl)]
acts = e["actions"]
for c in range(463):
    a = np.stack([acts[c], [0.0, 0.0], [8.0, 15.0]])
    o, s, r, d = b.step(a)
    out += [o, s, r, d.astype(np.float64)]
out.append(np.array(b.envs["x"])); out.append(np.array(b.envs["qw"])); out.append(b.reward_parts())
out.append(b.reset_carry(infl)); out.append(b.rollout(100, 3)); out.append(b.policy_actions(2, 3).astype(np.float64))
out.append(b.scenarios(4).astype(np.float64))
c = O.OracleCycleBatch(2, nthreads=1)
g = np.load(%r)
for k in range(2):
    mix2 = O.OracleBatch(2).mix(means, stds, np.array([0, 5], dtype=np.int32), np.stack([g["rnd"][k], g["rnd"][k + 1]]))
    out.append(c.reset(mix2, carry_over=(k == 1)))
    st, rew, diag = c.step(np.stack([g["actions"][k], [0.0, 1.0, 0.3]]))
    out += [st, rew, diag]
np.save(sys.argv[2], np.concatenate([np.asarray(v, dtype=np.float64).ravel() for v in out]))
print("ok", variant or "regular")
'''


@pytest.mark.skipif(shutil.which("gcc") is None, reason="gcc not available")
def test_c_oracle_runs_clean_under_ubsan_and_gives_the_same_numbers(tmp_path):
    import numpy as np
    code = CHILD % (ROOT, os.path.join(ROOT, "tests", "golden", "sbros_random_a.npz"), os.path.join(ROOT, "tests", "golden", "sbrv2_cycles.npz"))
    res = {}
    for variant in ("", "_ubsan"):
        out = str(tmp_path / ("res%s.npy" % variant))
        env = dict(os.environ, UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1", OMP_NUM_THREADS="1")
        p = subprocess.run([sys.executable, "-c", code, variant, out], capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
        assert p.returncode == 0 and "runtime error" not in p.stderr, (variant, p.stderr[-3000:])
        res[variant] = np.load(out)
    assert res[""].shape == res["_ubsan"].shape and res[""].size > 20000
    same = (res[""] == res["_ubsan"]) | (np.isnan(res[""]) & np.isnan(res["_ubsan"]))
    assert same.all(), int((~same).sum())


HOST_CHILD = r'''
import ctypes as C, sys
sys.path.insert(0, %r)
from gym_sbr2_amd import _capi
lib = C.CDLL(sys.argv[1])
lib.sbr_last_error.restype = C.c_char_p; lib.sbr_version.restype = C.c_char_p; lib.sbr_num_envs.restype = C.c_int64
cfg = _capi.SbrConfig()
assert lib.sbr_default_config(C.byref(cfg)) == 0 and lib.sbr_default_config(None) == -1
thr = (C.c_double * 2)()
assert lib.sbr_rows_thresholds(C.byref(cfg), thr) == 0 and lib.sbr_rows_thresholds(None, thr) == 0 and lib.sbr_rows_thresholds(None, None) == -1
cfg.dt = 0.0
assert lib.sbr_rows_thresholds(C.byref(cfg), thr) == -1
assert lib.sbr_abi_version() == _capi.ABI_VERSION and b"gfx950" in lib.sbr_version() and lib.sbr_device_count() == 0
h = C.c_void_p()
assert lib.sbr_create(C.c_int64(8), 0, C.c_int64(0), None, C.byref(h)) == -2 and b"no CPU path" in lib.sbr_last_error(None)
assert lib.sbr_create(C.c_int64(0), 0, C.c_int64(0), None, C.byref(h)) == -1 and lib.sbr_create(C.c_int64(8), 0, C.c_int64(0), None, None) == -1
assert lib.sbr_destroy(None) == 0 and lib.sbr_num_envs(None) == 0
for fn, args in (("sbr_step", (None,) * 7), ("sbr_reset", (None, C.c_uint64(0)) + (None,) * 6), ("sbr_set_trace", (None, None, C.c_int64(0), C.c_int64(0), 34)),
                 ("sbr_rollout", (None, 1, C.c_uint64(0), None, None, None)), ("sbr_get_state", (None,) * 4), ("sbr_synchronize", (None, None))):
    assert getattr(lib, fn)(*args) != 0, fn
print("ok")
'''


@pytest.mark.skipif(shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"), reason="hipcc not available")
def test_host_half_of_the_library_under_ubsan_without_a_device(tmp_path):
    """The no-device paths of libsbr_amd.so (what a machine without a GPU can reach: configuration defaults, the row-count
    thresholds, argument checks, the loud NO_DEVICE failure) built with the host compiler's UndefinedBehaviorSanitizer in trap
    mode (`-fsanitize=undefined -fsanitize-trap=undefined`: no runtime library needed inside a dlopen'ed .so; undefined
    behaviour kills the child with SIGILL).  The device code is compiled as always - the sanitizer does not exist for amdgcn."""
    from gym_sbr2_amd import build as B
    lib = str(tmp_path / "libsbr_amd_ubsan.so")
    flags = [f for f in B.FLAGS if f != "-O3"] + ["-O1", "-fsanitize=undefined", "-fsanitize-trap=undefined"]
    subprocess.check_call([B.hipcc()] + flags + ["-o", lib, B.SRC], stderr=subprocess.DEVNULL)
    env = dict(os.environ, HIP_VISIBLE_DEVICES="", ROCR_VISIBLE_DEVICES="")
    p = subprocess.run([sys.executable, "-c", HOST_CHILD % ROOT, lib], capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert p.returncode == 0 and p.stdout.strip().endswith("ok"), (p.returncode, p.stderr[-2000:])
