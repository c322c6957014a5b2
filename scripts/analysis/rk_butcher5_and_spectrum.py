import sys, ctypes as C, numpy as np
src = open(__import__('os').path.join(__import__('os').path.dirname(__import__('os').path.abspath(__file__)), 'rk_schemes_sweep.py')).read()
exec(src.split("# collect intervals")[0])
ivs = []
for name in EPISODES:
    e = golden("sbros_" + name)
    for i in range(len(e["iv_kind"])):
        ivs.append((e["iv_x_start"][i], float(e["iv_t_end"][i]) - float(e["iv_t_start"][i]), float(e["iv_Kla"][i]), float(e["iv_EC"][i]), e["iv_x_end"][i]))
lib.sbro_rk4.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_double), C.c_double, C.c_int, C.c_double, C.c_double, C.POINTER(C.c_double)]
step, stages = B5
for n in (4, 5):
    w_exact = w_ref = 0; wk = None
    for k, (x0, span, kla, ec, xref) in enumerate(ivs):
        xe = x0.copy(); lib.sbro_rk4(C.byref(p), 0, O._p(xe), span, 160, kla, ec, None)
        x = x0.copy(); h = span / n
        for _ in range(n): x = step(x, h, kla, ec)
        g = gate(x, xe)
        if g > w_exact: w_exact, wk = g, k
        w_ref = max(w_ref, gate(x, xref))
    print("Butcher5 n=%d (%d RHS): worst gate vs RK4-160 %.4f (interval %d, kla %.1f ec %.2e), vs reference LSODA %.4f" % (n, stages*n, w_exact, wk, ivs[wk][2], ivs[wk][3], w_ref), flush=True)
# stiffness: largest |eigenvalue| of the Jacobian (central differences) over golden interval start AND end states
def jac(x, kla, ec):
    J = np.empty((14, 14))
    for j in range(14):
        d = 1e-6 * max(1.0, abs(x[j])); xp = x.copy(); xm = x.copy(); xp[j] += d; xm[j] -= d
        J[:, j] = (f(xp, kla, ec) - f(xm, kla, ec)) / (2 * d)
    return J
lam = []
for k in range(0, len(ivs), 2):
    x0, span, kla, ec, xref = ivs[k]
    for x in (x0, xref):
        ev = np.linalg.eigvals(jac(x, kla, ec))
        lam.append((np.abs(ev.real).max(), np.abs(ev.imag).max(), kla, x[8]))
lam = np.array(lam)
i = lam[:, 0].argmax()
print("golden states: max |Re lambda| = %.0f /d (Kla %.0f, So %.3g), max |Im| = %.0f;  dt*lambda = %.3f" % (lam[i, 0], lam[i, 2], lam[i, 3], lam[:, 1].max(), lam[i, 0] * 0.002 / 24))
print("percentiles of |Re lambda|: 50%% %.0f  90%% %.0f  99%% %.0f" % tuple(np.percentile(lam[:, 0], [50, 90, 99])))
