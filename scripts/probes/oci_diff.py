"""Where does the done-call OCI reward of a free-running device episode differ from the oracle's?  (diagnostic)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import gym_sbr2_amd as G
from gym_sbr2_amd import _capi
from oracle import sbr_oracle as O
t = np.load(os.path.join(os.path.dirname(G.__file__), "data", "influent_tables.npz"))
means, stds = t["means"], t["stds"]
cfg = _capi.default_config(); cfg.reward_kind = 2; cfg.act_f64 = 1
n = 192
env = G.SbrOSVec(n, out_dtype=torch.float64, config=cfg)
p = O.default_params(); p.reward_kind = 2
ora = O.OracleBatch(n, params=p)
z = np.random.RandomState(2).randn(n, 48); scen = (np.arange(n) % 8).astype(np.int32)
env.reset(scenario=scen, rnd=z); ora.reset(ora.mix(means, stds, scen, z))
a = np.column_stack([np.linspace(0.5, 4.0, n), np.linspace(0.0, 12.0, n)])
at = torch.from_numpy(a).cuda()
for c in range(463):
    _, _, r, d = env.step(at); _, _, orr, od = ora.step(a)
    if c in (100, 300, 461):
        ks = env.ctrl_row(_capi.C_KLA_SUM).cpu().numpy()
        print(c, "max |dksum|", np.abs(ks - ora.envs["kla_sum"]).max(), "rel", np.abs(ks / ora.envs["kla_sum"] - 1).max(),
              "max |dKla_last|", np.abs(env.ctrl_row(_capi.C_KLA_LAST).cpu().numpy() - ora.envs["kla_last"]).max())
r = r.cpu().numpy()
ks = env.ctrl_row(_capi.C_KLA_SUM).cpu().numpy(); dks = ks - ora.envs["kla_sum"]
dqw = env.ctrl_row(_capi.C_QW).cpu().numpy() - ora.envs["qw"]
coef = 8.000000000006622 / 1800 * 1.32 * (0.002 / 24)
print("done: max|dr|", np.abs(r - orr).max(), "max|dks|", np.abs(dks).max(), "rel", np.abs(dks / ks).max(), "max|dqw|", np.abs(dqw).max())
print("residual after removing both terms:", np.abs((r - orr) + 0.05 * dqw + coef * dks).max())
x, ctrl = env.get_state()
st = ctrl[_capi.C_STATUS].cpu().numpy().astype(int)
print("status bits set:", np.bincount(st, minlength=8).tolist(), " worst env", int(np.abs(dks).argmax()), "status there", st[np.abs(dks).argmax()])
clean = (st & _capi.ST_NEAR_POLE) == 0
print("clean envs:", int(clean.sum()), "max|dks| rel", np.abs(dks / ks)[clean].max(), "max|dqw|", np.abs(dqw)[clean].max(),
      "max|dr|", np.abs(r - orr)[clean].max(), " penalised:", int((r < -200).sum()), "of which clean", int(((r < -200) & clean).sum()))
