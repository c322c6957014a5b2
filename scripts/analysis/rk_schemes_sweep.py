"""CPU experiment behind DESIGN.md 4.3: accuracy of explicit Runge-Kutta schemes per RHS evaluation on the golden intervals (test infrastructure: uses oracle/)."""
import sys, ctypes as C, numpy as np
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__)))))
from oracle import sbr_oracle as O
from tests.conftest import EPISODES, golden
lib = O.lib(); p = O.default_params()
scale = np.array([1.32,30,30,1500,150,3000,2000,600,8,20,20,10,10,10.])
def gate(x, ref): return (np.abs(x - ref) / (1e-5*np.abs(ref) + 1e-5*scale)).max()
lib.sbro_rhs_reaction.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.c_double, C.c_double, C.POINTER(C.c_double)]
def f(x, kla, ec):
    d = np.empty(14); lib.sbro_rhs_reaction(C.byref(p), O._p(np.ascontiguousarray(x)), kla, ec, O._p(d)); return d

def erk(A, b, c=None):
    A = [np.array(r, dtype=float) for r in A]; b = np.array(b, dtype=float)
    def step(x, h, kla, ec):
        ks = []
        for i in range(len(b)):
            y = x.copy()
            for j in range(i):
                if A[i][j] != 0: y = y + h * A[i][j] * ks[j]
            ks.append(f(y, kla, ec))
        out = x.copy()
        for i in range(len(b)):
            if b[i] != 0: out = out + h * b[i] * ks[i]
        return out
    return step, len(b)

RK4 = erk([[],[.5],[0,.5],[0,0,1]], [1/6,1/3,1/3,1/6])
# Kutta 3/8 rule
RK38 = erk([[],[1/3],[-1/3,1],[1,-1,1]], [1/8,3/8,3/8,1/8])
# Butcher RK5 (6 stages)
B5 = erk([[],[1/4],[1/8,1/8],[0,-1/2,1],[3/16,0,0,9/16],[-3/7,2/7,12/7,-12/7,8/7]], [7/90,0,32/90,12/90,32/90,7/90])
# Dormand-Prince 5 (use 5th-order weights; 6 effective stages, the 7th is FSAL and has b=0)
DP5 = erk([[],[1/5],[3/40,9/40],[44/45,-56/15,32/9],[19372/6561,-25360/2187,64448/6561,-212/729],
           [9017/3168,-355/33,46732/5247,49/176,-5103/18656]], [35/384,0,500/1113,125/192,-2187/6784,11/84])
# Cash-Karp 5
CK5 = erk([[],[1/5],[3/40,9/40],[3/10,-9/10,6/5],[-11/54,5/2,-70/27,35/27],[1631/55296,175/512,575/13824,44275/110592,253/4096]],
          [37/378,0,250/621,125/594,0,512/1771])
# Butcher 7-stage RK6
s21=np.sqrt(21)
RK6 = erk([[],[1],[3/8,1/8],[8/27,2/27,8/27],[3*(3*s21-7)/392,-8*(7-s21)/392,48*(7-s21)/392,-3*(21-s21)/392],
           [-5*(231+51*s21)/1960,-40*(7+s21)/1960,-320*s21/1960,3*(21+121*s21)/1960,392*(6+s21)/1960],
           [15*(22+7*s21)/180,120/180,40*(7*s21-5)/180,-63*(3*s21-2)/180,-14*(49+9*s21)/180,70*(7-s21)/180]],
          [9/180,0,64/180,0,49/180,49/180,9/180])
methods = {"RK4": RK4, "RK4-3/8": RK38, "Butcher5": B5, "DP5": DP5, "CashKarp5": CK5, "Butcher6": RK6}
# sanity: order check on y' = -y... skipped; consistency: sum b = 1
# collect intervals (subsample for speed: every 3rd interval of each episode + all with EC>0 switching)
ivs = []
for name in EPISODES:
    e = golden("sbros_" + name)
    for i in range(0, len(e["iv_kind"])):
        ivs.append((e["iv_x_start"][i], float(e["iv_t_end"][i]) - float(e["iv_t_start"][i]), float(e["iv_Kla"][i]), float(e["iv_EC"][i])))
print(len(ivs), "intervals")
# exact: RK4 with 80 substeps through the C routine
lib.sbro_rk4.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_double), C.c_double, C.c_int, C.c_double, C.c_double, C.POINTER(C.c_double)]
exact = []
for x0, span, kla, ec in ivs:
    x = x0.copy(); lib.sbro_rk4(C.byref(p), 0, O._p(x), span, 160, kla, ec, None); exact.append(x)
# pre-select the 150 hardest intervals by the RK4-5 error to keep the python loops short
hard = []
for k, (x0, span, kla, ec) in enumerate(ivs):
    x = x0.copy(); lib.sbro_rk4(C.byref(p), 0, O._p(x), span, 5, kla, ec, None); hard.append(gate(x, exact[k]))
order = np.argsort(hard)[::-1][:150]
print("RK4-5 worst (C):", max(hard))
for name, (step, stages) in methods.items():
    for n in (2, 3, 4, 5, 6, 8, 10):
        if stages * n > 44: continue
        worst = 0
        for k in order:
            x0, span, kla, ec = ivs[k]
            x = x0.copy(); h = span / n
            for _ in range(n): x = step(x, h, kla, ec)
            worst = max(worst, gate(x, exact[k]))
        print("%-10s n=%2d  RHS/interval=%3d  worst gate %.4f" % (name, n, stages * n, worst), flush=True)
