"""First contact with the hardware: measure GPU-vs-oracle differences (to set test tolerances from data)
and a first timing.  Writes gpurun_out/explore.txt."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from gym_sbr2_amd import SbrOSVec, _capi
from gym_sbr2_amd.vec_env import load_influent_tables
from oracle import sbr_oracle as O, sbr_params as P

os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
out = open(os.path.join(ROOT, "gpurun_out", "explore.txt"), "w")
def say(*a):
    s = " ".join(str(x) for x in a); print(s, flush=True); out.write(s + "\n"); out.flush()
g = lambda n: np.load(os.path.join(ROOT, "tests", "golden", n + ".npz"))
def gate(x, ref): return np.abs(x - ref) / (P.RTOL_GATE * np.abs(ref) + P.RTOL_GATE * P.STATE_SCALE)
means, stds = load_influent_tables()
say("device", torch.cuda.get_device_name(0))

# 1. RHS known answers
k = g("rhs_kat"); env1 = SbrOSVec(len(k["X"]), out_dtype=torch.float64)
for kind, key in [(0, "d_reaction"), (1, "d_filling"), (2, "d_idle")]:
    ec = k["ec"] if kind == 0 else np.zeros_like(k["ec"])
    d = env1.eval_rhs(kind, k["X"], k["kla"], ec, k["loading"] if kind == 1 else None).cpu().numpy()
    rel = np.abs(d - k[key]) / np.abs(k[key]).max(axis=1, keepdims=True)
    say("rhs kind", kind, "max err relative to row max: %.3e" % rel.max())
env1.close()

# 2. the six golden episodes as a batch of 6 envs
names = ["const_2_5", "random_a", "random_b", "zeros", "max", "det_influent"]
E = [g("sbros_" + n) for n in names]
n = len(E); ncall = 463
rnd = np.stack([e["rnd"] for e in E]); acts = np.stack([e["actions"][:ncall] for e in E], axis=1).astype(np.float32)
env = SbrOSVec(n, out_dtype=torch.float64); ora = O.OracleBatch(n)
obs = env.reset(rnd=rnd).cpu().numpy()
infl = ora.mix(means, stds, [6] * n, rnd); oobs = ora.reset(infl)
say("influent max abs diff vs oracle %.3e ; vs golden %.3e" % (np.abs(env.influent().cpu().numpy().T - ora.envs["influent"]).max(),
    np.abs(env.influent().cpu().numpy().T - np.stack([e["influent_mixed"] for e in E])).max()))
x, ctrl = env.get_state()
say("post-fill: max gate vs oracle %.3e ; vs golden %.3e ; reset obs diff %.3e" % (gate(x.cpu().numpy().T, ora.envs["x"]).max(),
    gate(x.cpu().numpy().T, np.stack([e["x_postfill"] for e in E])).max(), np.abs(obs - oobs).max()))
w = dict(x=np.zeros(n), obs=np.zeros(n), state=np.zeros(n), rew=np.zeros(n), gold=np.zeros(n), kla=np.zeros(n), ec=np.zeros(n), ie=np.zeros(n))
for c in range(ncall):
    o, s, r, d = env.step(torch.from_numpy(acts[c]).cuda())
    oo, os_, orr, od = ora.step(acts[c].astype(np.float64))
    x, ctrl = env.get_state(); x = x.cpu().numpy().T; ctrl = ctrl.cpu().numpy()
    assert np.array_equal(d.cpu().numpy(), od), ("done mismatch at call", c)
    w["x"] = np.maximum(w["x"], gate(x, ora.envs["x"]).max(1))
    w["obs"] = np.maximum(w["obs"], np.abs(o.cpu().numpy() - oo).max(1)); w["state"] = np.maximum(w["state"], np.abs(s.cpu().numpy() - os_).max(1))
    w["rew"] = np.maximum(w["rew"], np.abs(r.cpu().numpy() - orr))
    w["kla"] = np.maximum(w["kla"], np.abs(ctrl[_capi.C_KLA_LAST] - ora.envs["kla_last"])); w["ec"] = np.maximum(w["ec"], np.abs(ctrl[_capi.C_EC_LAST] - ora.envs["ec_last"]))
    w["ie"] = np.maximum(w["ie"], np.abs(ctrl[_capi.C_IE_DO] - ora.envs["ie_do"]))
    if c < ncall - 1:
        w["gold"] = np.maximum(w["gold"], np.array([gate(x[i], E[i]["step_x_end"][c]).max() for i in range(n)]))
for i, nm in enumerate(names):
    say("%-13s GPU-vs-oracle: gate(x) %.3e obs %.3e state %.3e reward %.3e Kla %.3e EC %.3e ie_DO %.3e | GPU-vs-golden gate %.3f | after idle vs golden gate %.3f | Qw gpu %.15g oracle %.15g golden %.15g | return gpu %.15g golden %.15g" % (
        nm, w["x"][i], w["obs"][i], w["state"][i], w["rew"][i], w["kla"][i], w["ec"][i], w["ie"][i], w["gold"][i],
        gate(x[i], E[i]["term_x_after_idle"]).max(), ctrl[_capi.C_QW][i], ora.envs["qw"][i], float(E[i]["term_Qw"]),
        ctrl[_capi.C_RETURN][i], float(E[i]["episode_return"])))
env.close()

# 3. normals
envn = SbrOSVec(512); z = envn.draw_normals(7).cpu().numpy(); zo = O.OracleBatch(512).normals(7)
say("philox normals: max abs diff vs oracle %.3e  mean %.4f std %.4f" % (np.abs(z - zo).max(), z.mean(), z.std())); envn.close()

# 4. first timing
for N in (4096, 65536, 262144):
    env = SbrOSVec(N); env.reset(scenario=(np.arange(N) % 8).astype(np.int32), rnd=np.zeros((N, 48)))
    a = torch.rand(N, 2, device="cuda") * torch.tensor([8.0, 15.0], device="cuda")
    for _ in range(20): env.step(a)
    torch.cuda.synchronize(); env.timer_start(); t0 = time.time(); K = 200
    for _ in range(K): env.step(a)
    ms = env.timer_stop(); torch.cuda.synchronize(); wall = time.time() - t0
    say("N=%d: %.2f us/launch (events), wall %.2f us/step -> %.3e env-steps/s ; 513 B/env-step -> %.1f GB/s" % (N, ms * 1e3 / K, wall * 1e6 / K, N * K / (ms * 1e-3), N * 513 / (ms * 1e-3 / K) / 1e9))
    t0 = time.time(); env.reset(scenario=(np.arange(N) % 8).astype(np.int32), rnd=np.zeros((N, 48))); torch.cuda.synchronize(); say("   reset wall %.2f ms" % ((time.time() - t0) * 1e3))
    t0 = time.time(); ret = env.rollout(462, 1); torch.cuda.synchronize(); dt = time.time() - t0
    say("   fused rollout 462 calls: %.2f ms -> %.3e env-steps/s ; mean return %.6f" % (dt * 1e3, N * 462 / dt, env.stats(ret)["mean"]))
    env.close()
say("done")
