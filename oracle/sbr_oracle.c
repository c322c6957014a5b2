/*
 * sbr_oracle.c - CPU oracle, layer 2 (TEST INFRASTRUCTURE, not product code).
 *
 * Plain-C, fp64 restatement of the SBROS-v1 path of SungKu/gym-SBR2
 * (/root/reference/gym_SBR/envs/gym_SBR_oneshot.py::SbrOS) with the fixed-step RK4 integrator
 * that the HIP kernels use (10 substeps per control interval) in place of SciPy's LSODA.
 * Each function cites the reference lines it follows.  It is pinned by tests/test_oracle_golden.py
 * against the tests/golden fixtures (captured from the running reference by oracle/gen_golden.py) and
 * against oracle/sbr_ref.py (same algorithm with the reference's own LSODA).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 * The product (libsbr_amd.so) never links, loads or calls it.
 *
 * Build: make -C oracle   (gcc -O2 -ffp-contract=off -fopenmp -shared -fPIC)
 */
#include <math.h>
#include <stdint.h>
#include <string.h>

#define NX 14
#define KLA_HIST 10

typedef struct {
    double Ya, Yh, fp, ixb, ixp;
    double muH, Ks, Koh, Kno, bH, eta_g, eta_h, kh, Kx, muA, Knh, bA, Koa, ka;
    double WV, IV, dt, t_delta, t_cycle;
    double T_fill, T3_0, T3_end, T4_end, T5_end;
    double t_settle, t_draw;
    double So_sat, Kla_min, Kla_max, Kc_DO, tauI_DO, tauD_DO;
    double EC_min, EC_max, Kc_EC, tauI_EC, tauD_EC, EC_conc;
    double act_DO_max, act_EC_max;
    double biomass_setpoint, Qeff, settler_area, settler_vmax;
    double t_ratio[8];
    double cyc_Kc, cyc_tauI, cyc_tauD, cyc_dt;
    double x0[NX];
    int32_t substeps, out_f64, terminal, reward_kind, act_f64, random_scenario;
    int32_t scheme;                 /* 0: RK4 x substeps per control interval; 1: adaptive Butcher-5 (b5a_interval) */
    int32_t reserved_;
} sbro_params;

/* one environment; field order is part of the ctypes contract in oracle/sbr_oracle.py */
typedef struct {
    double x[NX];
    double t;
    double so_m1, so_m2, sno_m1, sno_m2;
    double ie_do, ie_ec;
    double kla_last, ec_last, ec_prev;
    double u_do, u_ec;
    double kla_hist[KLA_HIST];      /* oldest first; [KLA_HIST-1] is the current interval's Kla */
    double qw, ret, steps, done, status;
    double kla_sum;                 /* sum(Kla) over the episode's whole list, in append order (reward_kind 2) */
    double influent[NX];            /* loading vector, [0] = Qin/T_fill */
    double x_start[NX];             /* start state of the last interval (for xdot) */
    double span;                    /* t_range[-1]-t_range[0] of the last interval */
    int32_t n_rows;                 /* 9 or 10: len(t_range) of the last interval */
    int32_t n_intervals;            /* intervals run by the last step() call */
    int32_t scheme_steps;           /* scheme 1: step count of the last interval (-1: scheme 0) */
    int32_t scheme_plan;            /* the product's plan codes (SBR_C_PLAN / SBR_TR_PLAN, include/sbr_amd.h): bits 0-7 = step count of
                                       the LAST interval of the last step() call + 128 if dissolved oxygen was held, bits 8-15 = the
                                       same for the call's FIRST interval (equal when it ran one); 0 under scheme 0 */
} sbro_env;

static double status_bits(const sbro_params* p, const double* x, double status);

static const double X1_STATE[15] = {0.5, 1.32, 30, 30, 1500, 150, 3000, 2000, 600, 8, 20, 20, 10, 10, 10};

/* ---------------------------------------------------------------------------------- defaults */
void sbro_default_params(sbro_params* p) {
    /* gym_SBR_oneshot.py:116-119 */
    p->Ya = 0.24; p->Yh = 0.67; p->fp = 0.08; p->ixb = 0.08; p->ixp = 0.06;
    p->muH = 4.0; p->Ks = 10.0; p->Koh = 0.2; p->Kno = 0.5; p->bH = 0.3; p->eta_g = 0.8; p->eta_h = 0.8;
    p->kh = 3.0; p->Kx = 0.1; p->muA = 0.5; p->Knh = 1.0; p->bA = 0.05; p->Koa = 0.4; p->ka = 0.05;
    /* :25-37, :197 */
    p->WV = 1.32; p->IV = 0.6161484733495801; p->dt = 0.002 / 24; p->t_delta = p->dt * 10; p->t_cycle = 12.0 / 24;
    /* module_batch_time.py:3-116, values asserted against tests/golden/constants.npz */
    p->T_fill = 0.021; p->T3_0 = 0.06416666666666668; p->T3_end = 0.2516666666666667;
    p->T4_end = 0.4085000000000001; p->T5_end = 0.40933333333333344;
    p->t_settle = 8.3 / 100; p->t_draw = 2.1 / 100;
    /* :80-96 ; So_sat = DO_set(15), module_temperature.py:3-20 */
    p->So_sat = 8.000000000006622; p->Kla_min = 0; p->Kla_max = 240; p->Kc_DO = 100; p->tauI_DO = 20; p->tauD_DO = 0;
    p->EC_min = 0; p->EC_max = 0.0005; p->Kc_EC = 100; p->tauI_EC = 20; p->tauD_EC = 0; p->EC_conc = 1200000 * 4.0;
    p->act_DO_max = 8; p->act_EC_max = 15;
    /* :123-124, :2189, :2211 */
    p->biomass_setpoint = 2700; p->Qeff = 0.66; p->settler_area = (1.25 / 2) * (1.25 / 2); p->settler_vmax = 474;
    static const double tr[8] = {4.2 / 100, 8.3 / 100, 37.5 / 100, 31.2 / 100, 2.1 / 100, 8.3 / 100, 2.1 / 100, 6.3 / 100};
    memcpy(p->t_ratio, tr, sizeof tr);
    p->cyc_Kc = 5.0; p->cyc_tauI = 0.00035; p->cyc_tauD = 0.005; p->cyc_dt = 0.02 / 24;    /* gym_SBR_env2.py:48 */
    static const double x0[NX] = {0.6161484733495801, 30, 0.571098000538576, 1440.01157895393, 31.254221999137,
                                  2599.2714348941, 168.915006750837, 551.901552960823, 2.16607843793004,
                                  13.3791460027604, 0.00562880208518134, 0.35996687629947, 1.86916737961228,
                                  3.790463057094611};
    memcpy(p->x0, x0, sizeof x0);
    p->substeps = 10; p->out_f64 = 1; p->terminal = 1; p->reward_kind = 0; p->act_f64 = 0; p->random_scenario = 0;
    p->scheme = 1; p->reserved_ = 0;          /* the product's default (sbr_default_config) */
}

int sbro_sizeof_env(void) { return (int)sizeof(sbro_env); }
int sbro_sizeof_params(void) { return (int)sizeof(sbro_params); }

/* ---------------------------------------------------------------------------------- RHS */
/* conversion rates r[1..13] incl. aeration: process rates :1660-1685, stoichiometry :1689-1725,
 * combination :1731-1755 */
static void conversion(const sbro_params* p, const double* x, double kla, double* r) {
    const double ss = x[2], xs = x[4], xbh = x[5], xba = x[6], so = x[8], sno = x[9], snh = x[10], snd = x[11],
                 xnd = x[12];
    const double rho1 = p->muH * (ss / (p->Ks + ss)) * (so / (p->Koh + so)) * xbh;
    const double rho2 = p->muH * (ss / (p->Ks + ss)) * (p->Koh / (so + p->Koh)) * (sno / (p->Kno + sno)) * p->eta_g * xbh;
    const double rho3 = p->muA * (snh / (p->Knh + snh)) * (so / (p->Koa + so)) * xba;
    const double rho4 = p->bH * xbh;
    const double rho5 = p->bA * xba;
    const double rho6 = p->ka * snd * xbh;
    const double rho7 = p->kh * ((xs / xbh) / (p->Kx + (xs / xbh))) *
                        ((so / (p->Koh + so)) + p->eta_h * (p->Koh / (so + p->Koh)) * (sno / (p->Kno + sno))) * xbh;
    const double rho8 = (xnd / xs) * rho7;
    const double Yh = p->Yh, Ya = p->Ya, ixb = p->ixb, ixp = p->ixp, fp = p->fp;
    r[0] = 0; r[1] = 0; r[3] = 0;
    r[2] = (-1 / Yh) * rho1 + (-1 / Yh) * rho2 + rho7;
    r[4] = (1 - ixp) * rho4 + (1 - ixp) * rho5 + (-1.0) * rho7;
    r[5] = rho1 + rho2 + (-1.0) * rho4;
    r[6] = rho3 + (-1.0) * rho5;
    r[7] = ixp * rho4 + ixp * rho5;
    r[8] = (-(1 - Yh) / Yh) * rho1 + (-(4.57 - Ya) / Ya) * rho3 + kla * (p->So_sat - so);
    r[9] = (-((1 - Yh) / (2.86 * Yh))) * rho2 + (1 / Ya) * rho3;
    r[10] = (-ixb) * rho1 + (-ixb) * rho2 + (-ixb - 1 / Ya) * rho3 + rho6;
    r[11] = (-1.0) * rho6 + rho8;
    r[12] = (ixb - fp * ixp) * rho4 + (ixb - fp * ixp) * rho5 + (-1.0) * rho8;
    r[13] = (-ixb / 14) * rho1 + ((1 - Yh) / (14 * 2.86 * Yh) - ixb / 14) * rho2 + (-ixb / 14 - 1 / (7 * Ya)) * rho3 +
            (1.0 / 14) * rho6;
}

/* reaction_dxdt :1658-1787 */
void sbro_rhs_reaction(const sbro_params* p, const double* x, double kla, double ec, double* d) {
    double r[NX];
    conversion(p, x, kla, r);
    const double q = ec / x[0];
    d[0] = 0 + ec;
    for (int i = 1; i < NX; ++i) d[i] = r[i] + q * (i == 2 ? (p->EC_conc - x[i]) : (-x[i]));
}

/* filling_dxdt :1424-1583, EC = 0 (the only way the path calls it).  The reference's in-place
 * dilution block :1523-1551 is then x[i] = (x[i]*V)/V - the identity up to 1 ulp - and is not restated. */
void sbro_rhs_fill(const sbro_params* p, const double* x, double kla, const double* loading, double* d) {
    double r[NX];
    conversion(p, x, kla, r);
    const double q = loading[0] / x[0];
    d[0] = loading[0];
    for (int i = 1; i < NX; ++i) d[i] = r[i] + q * (loading[i] - x[i]);
}

/* idle_dxdt :2424-2552 */
void sbro_rhs_idle(const sbro_params* p, const double* x, double kla, double* d) {
    double r[NX];
    conversion(p, x, kla, r);
    d[0] = 0;
    for (int i = 1; i < NX; ++i) d[i] = r[i];
}

void sbro_eval_rhs(const sbro_params* p, int kind, int64_t n, const double* x, const double* kla, const double* ec,
                   const double* loading, double* dx) {
    for (int64_t i = 0; i < n; ++i) {
        if (kind == 0) sbro_rhs_reaction(p, x + i * NX, kla[i], ec[i], dx + i * NX);
        else if (kind == 1) sbro_rhs_fill(p, x + i * NX, kla[i], loading + i * NX, dx + i * NX);
        else sbro_rhs_idle(p, x + i * NX, kla[i], dx + i * NX);
    }
}

static void rk4_reaction_w(const sbro_params* p, double* x, double span, int n, double kla, double ec);

/* classical RK4, n equal substeps over [0, span]; the systems are autonomous inside a span.
 * kind 0: reaction(kla, ec)  1: fill(kla, loading)  2: idle(kla) */
static void rk4_span(const sbro_params* p, int kind, double* x, double span, int n, double kla, double ec,
                     const double* loading) {
    if (kind == 0 && ec != 0.0) { rk4_reaction_w(p, x, span, n, kla, ec); return; }
    const double h = span / n;
    double k1[NX], k2[NX], k3[NX], k4[NX], y[NX];
    for (int s = 0; s < n; ++s) {
#define F(in, out)                                                \
    do {                                                          \
        if (kind == 0) sbro_rhs_reaction(p, in, kla, ec, out);    \
        else if (kind == 1) sbro_rhs_fill(p, in, kla, loading, out); \
        else sbro_rhs_idle(p, in, kla, out);                      \
    } while (0)
        F(x, k1);
        for (int i = 0; i < NX; ++i) y[i] = x[i] + (0.5 * h) * k1[i];
        F(y, k2);
        for (int i = 0; i < NX; ++i) y[i] = x[i] + (0.5 * h) * k2[i];
        F(y, k3);
        for (int i = 0; i < NX; ++i) y[i] = x[i] + h * k3[i];
        F(y, k4);
        for (int i = 0; i < NX; ++i) x[i] = x[i] + (h / 6.0) * (k1[i] + 2.0 * k2[i] + 2.0 * k3[i] + k4[i]);
#undef F
    }
}

/* A reaction interval that doses carbon (ec != 0), integrated in SCALED-MASS variables (round 4; the HIP kernels' dosing
 * loop sbr_rk4_dose integrates the same system, and so does oracle/sbr_ref.py rk4_reaction_w, bit for bit like this):
 *     w_i = c_i V/V0 = c_i s,   V0 = the volume at the start of the interval,   s = V/V0.
 * reaction_dxdt (:1658-1787) is  V' = ec,  c_i' = r_i(c) + (ec/V)(c_in,i - c_i)  with c_in = EC_conc for Ss and 0 otherwise;
 * substituting c = w/s gives the SAME differential equations without the dilution terms,
 *     V' = ec,   w_i' = s r_i(w/s) + (ec/V0) c_in,i ,
 * an autonomous system in (V, w) on which classical RK4 is run as before; at the end of the interval c = w/s.  It is a
 * restatement of the reference's right-hand side in other variables, not another model: r is the reference's conversion()
 * evaluated on c = w/s.  RK4 on w and RK4 on c are both fourth-order discretisations of that system and differ by
 * ~(ec h/V) x the local truncation error (1e-14 relative per substep); both stay inside the same gates of the reference's
 * LSODA trajectories (tests/test_oracle_golden.py).  With ec == 0 the two coincide operation for operation (s = 1), so
 * rk4_span keeps the plain form there. */
static void rhs_reaction_w(const sbro_params* p, const double* y, double v0, double kla, double ec, double* d) {
    double c[NX], r[NX];
    const double s = y[0] / v0;
    c[0] = y[0];
    for (int i = 1; i < NX; ++i) c[i] = y[i] / s;
    conversion(p, c, kla, r);
    const double q0 = ec / v0;
    d[0] = 0 + ec;
    for (int i = 1; i < NX; ++i) d[i] = s * r[i] + q0 * (i == 2 ? p->EC_conc : 0.0);
}

static void rk4_reaction_w(const sbro_params* p, double* x, double span, int n, double kla, double ec) {
    const double h = span / n;
    const double v0 = x[0];
    double k1[NX], k2[NX], k3[NX], k4[NX], y[NX];
    for (int s = 0; s < n; ++s) {            /* x = (V, w); w = c at the start of the interval (s = 1) */
        rhs_reaction_w(p, x, v0, kla, ec, k1);
        for (int i = 0; i < NX; ++i) y[i] = x[i] + (0.5 * h) * k1[i];
        rhs_reaction_w(p, y, v0, kla, ec, k2);
        for (int i = 0; i < NX; ++i) y[i] = x[i] + (0.5 * h) * k2[i];
        rhs_reaction_w(p, y, v0, kla, ec, k3);
        for (int i = 0; i < NX; ++i) y[i] = x[i] + h * k3[i];
        rhs_reaction_w(p, y, v0, kla, ec, k4);
        for (int i = 0; i < NX; ++i) x[i] = x[i] + (h / 6.0) * (k1[i] + 2.0 * k2[i] + 2.0 * k3[i] + k4[i]);
    }
    const double s_end = x[0] / v0;
    for (int i = 1; i < NX; ++i) x[i] = x[i] / s_end;
}

/* ---------------------------------------------------------------------------------- scheme 1 ("B5A", round 5)
 * A span with Kla and the flows held, by Butcher's six-stage fifth-order scheme with a step count chosen from the plant's own
 * state at the start of every macro interval (a control interval; the idle phase is cut into ceil(rows/10) of them).  Same
 * operations in the same order as oracle/sbr_ref.py b5a_plan / b5_step / b5a_span (bit-identical); the reasoning is written
 * there and in DESIGN.md 3.0.   kind 0: reaction (ec != 0: scaled-mass form)   2: idle / closed reactor.  The fill phase is never
 * integrated this way (not plannable from its start state: the inflow changes Ss and Snh severalfold within an interval). */
#define B5A_SO_SLAVED 1e-9
#define B5A_Z1 0.3
#define B5A_Z2 1.0
#define B5A_Z_STAB 2.5
#define B5A_ZS1 0.15
#define B5A_ZS2 0.5
#define B5A_N_MAX 64
/* round 6, STUDY ONLY - off (0) by default and absent from the product.  ZR_STAB: the stability bound of the oxygen mode applied to the
 * three OTHER fast modes as well (uptake of Ss, Snh, Sno: their own decay rates at the interval's start), n >= floor(j span / zr) + 1.
 * Without effect with the reference's kinetic constants (rate x t_delta <= 1.0 on every captured interval); with 4 x faster kinetics it
 * halves scheme 1's blow-ups (tests/test_oracle_golden.py::test_scheme1_under_perturbed_kinetic_constants switches it on to keep that on
 * record).  Not adopted: three more wave-uniform constants in the plan cost the step kernel 0.2 us of 11.7 per call (profiles/r06_notes.md). */
static double g_zr_stab = 0.0;
void sbro_set_plan_knobs(double zr_stab) { g_zr_stab = zr_stab; }

static void b5a_rhs(const sbro_params* p, int kind, const double* y, double v0, double kla, double ec, int hold_so, double* k) {
    if (kind == 2) sbro_rhs_idle(p, y, kla, k);
    else if (ec != 0.0) rhs_reaction_w(p, y, v0, kla, ec, k);
    else sbro_rhs_reaction(p, y, kla, ec, k);
    if (hold_so) k[8] = 0.0;
}

/* one macro interval; returns the plan code: step count + 128 if So was held (slaved) */
static int b5a_macro(const sbro_params* p, int kind, double* x, double span, double kla, double ec) {
    static const double A21 = 0.25, A31 = 0.125, A32 = 0.125, A42 = -0.5, A43 = 1.0, A51 = 3.0 / 16.0, A54 = 9.0 / 16.0,
                        A61 = -3.0 / 7.0, A62 = 2.0 / 7.0, A63 = 12.0 / 7.0, A64 = -12.0 / 7.0, A65 = 8.0 / 7.0,
                        B1 = 7.0 / 90.0, B3 = 32.0 / 90.0, B4 = 12.0 / 90.0, B5 = 32.0 / 90.0, B6 = 7.0 / 90.0;
    const double v0 = x[0];
    const int dose = (kind == 0 && ec != 0.0);
    double k1[NX], k2[NX], k3[NX], k4[NX], k5[NX], k6[NX], y[NX];
    b5a_rhs(p, kind, x, v0, kla, ec, 0, k1);
    /* the plan (python: b5a_plan) */
    const double ss = x[2], xbh = x[5], xba = x[6], so = x[8], sno = x[9], snh = x[10];
    const double p2 = ss + k1[2] * span;
    const double ss_hi = p2 > ss ? p2 : ss;
    const double p10 = snh + k1[10] * span;
    const double snh_hi = p10 > snh ? p10 : snh;
    const double c1 = ((1 - p->Yh) / p->Yh) * p->muH * xbh;
    const double c3 = ((4.57 - p->Ya) / p->Ya) * p->muA * xba;
    const double m1s = ss / (p->Ks + ss), m3s = snh / (p->Knh + snh);
    const double m1 = ss_hi / (p->Ks + ss_hi), m3 = snh_hi / (p->Knh + snh_hi);
    const double a1 = c1 * m1, a3 = c3 * m3;
#define LAM(s_) (a1 * p->Koh / ((p->Koh + (s_)) * (p->Koh + (s_))) + a3 * p->Koa / ((p->Koa + (s_)) * (p->Koa + (s_))) + kla)
    const int slaved = (fabs(so) < B5A_SO_SLAVED) && (kla * p->So_sat * span < B5A_SO_SLAVED);
    const double slope_hi = k1[8] - c1 * (m1 - m1s) * (so / (p->Koh + so)) - c3 * (m3 - m3s) * (so / (p->Koa + so));
    const double slope = slope_hi < k1[8] ? slope_hi : k1[8];
    const double proj = so + slope * span;
    const double lo1 = proj < so ? proj : so;
    const double so_lo = lo1 > 0.0 ? lo1 : 0.0;
    const double z_ub = LAM(so_lo) * span;
    const double lam0 = LAM(0.0);
#undef LAM
    int n;
    if (slaved) n = 2;
    else if (z_ub < B5A_Z1) n = 1;
    else if (z_ub < B5A_Z2) n = 2;
    else {                                  /* the knee: at least four steps, and lam(0) h <= Z_STAB (Butcher-5: stable to 3.39) */
        const double q = lam0 * span / B5A_Z_STAB;
        n = q < 4.0 ? 4 : (!(q < (double)B5A_N_MAX) ? B5A_N_MAX : (int)q + 1);
        /* ... for a state inside the model's domain: there a1 and a3 are bounded (the Monod factors of Ss and Snh lie in [0, 1]).
         * Outside - Ss or Snh negative towards or beyond its pole, or NaN - the premise is gone and the state is garbage (the reference
         * has no guards either): the count stays at the knee's four, so that one such env cannot make a whole batch wait for its 64
         * steps (round 6).  No in-domain state is affected. */
        if (!(fabs(m1 - 0.5) <= 0.5 && fabs(m3 - 0.5) <= 0.5)) n = 4;          /* (also true for NaN) */
    }
    {   /* how far the arguments of the other Monod terms move within the interval */
        double zs = fabs(k1[2]) * span / (p->Ks + fabs(ss));
        const double z10 = fabs(k1[10]) * span / (p->Knh + fabs(snh));
        const double z9 = fabs(k1[9]) * span / (p->Kno + fabs(sno));
        zs = z10 > zs ? z10 : zs;
        zs = z9 > zs ? z9 : zs;
        const int n_s = zs < B5A_ZS1 ? 1 : (zs < B5A_ZS2 ? 2 : 4);
        n = n_s > n ? n_s : n;
        /* ... and never fewer than the stability of those three modes asks for (round 6): their own decay rates */
        const double i2 = 1.0 / (p->Ks + fabs(ss)), i10 = 1.0 / (p->Knh + fabs(snh)), i9 = 1.0 / (p->Kno + fabs(sno));
        const double j2 = ((p->muH / p->Yh) * p->Ks) * xbh * (i2 * i2);
        const double j10 = ((p->ixb + 1 / p->Ya) * p->muA * p->Knh) * xba * (i10 * i10);
        const double j9 = (((1 - p->Yh) / (2.86 * p->Yh)) * p->muH * p->eta_g * p->Kno) * xbh * m1 * (i9 * i9);
        double jr = j2 > j10 ? j2 : j10;
        jr = j9 > jr ? j9 : jr;
        if (g_zr_stab > 0) {
            const double qr = jr * span / g_zr_stab;
            const int n_r = qr < 1.0 ? 1 : (!(qr < (double)B5A_N_MAX) ? B5A_N_MAX : (int)qr + 1);
            n = n_r > n ? n_r : n;
        }
    }
    const double h = span / n;
    if (slaved) k1[8] = 0.0;
    for (int s = 0; s < n; ++s) {
        if (s > 0) b5a_rhs(p, kind, x, v0, kla, ec, slaved, k1);
        for (int i = 0; i < NX; ++i) y[i] = x[i] + (h * A21) * k1[i];
        b5a_rhs(p, kind, y, v0, kla, ec, slaved, k2);
        for (int i = 0; i < NX; ++i) y[i] = x[i] + (h * A31) * k1[i] + (h * A32) * k2[i];
        b5a_rhs(p, kind, y, v0, kla, ec, slaved, k3);
        for (int i = 0; i < NX; ++i) y[i] = x[i] + (h * A42) * k2[i] + (h * A43) * k3[i];
        b5a_rhs(p, kind, y, v0, kla, ec, slaved, k4);
        for (int i = 0; i < NX; ++i) y[i] = x[i] + (h * A51) * k1[i] + (h * A54) * k4[i];
        b5a_rhs(p, kind, y, v0, kla, ec, slaved, k5);
        for (int i = 0; i < NX; ++i)
            y[i] = x[i] + (h * A61) * k1[i] + (h * A62) * k2[i] + (h * A63) * k3[i] + (h * A64) * k4[i] + (h * A65) * k5[i];
        b5a_rhs(p, kind, y, v0, kla, ec, slaved, k6);
        for (int i = 0; i < NX; ++i)
            x[i] = x[i] + (h * B1) * k1[i] + (h * B3) * k3[i] + (h * B4) * k4[i] + (h * B5) * k5[i] + (h * B6) * k6[i];
    }
    if (dose) {
        const double s_end = x[0] / v0;
        for (int i = 1; i < NX; ++i) x[i] = x[i] / s_end;
    }
    if (slaved) x[8] = x[8] / (1.0 + lam0 * span);
    return n + (slaved ? 128 : 0);
}

/* m macro intervals of span/m each; returns the plan code of the last one */
static int b5a_span(const sbro_params* p, int kind, double* x, double span, int m, double kla, double ec) {
    const double hm = span / m;
    int n = 0;
    for (int j = 0; j < m; ++j) n = b5a_macro(p, kind, x, hm, kla, ec);
    return n;
}

/* scheme-aware integration of one reaction interval (python: SbrOsRef._integrate); returns the step count (-1: scheme 0) */
static int reaction_interval_plan(const sbro_params* p, double* x, double span, double kla, double ec) {
    if (p->scheme == 1) return b5a_span(p, 0, x, span, 1, kla, ec);
    rk4_span(p, 0, x, span, p->substeps, kla, ec, 0);
    return -1;
}
int sbro_reaction_interval(const sbro_params* p, double* x, double span, double kla, double ec) {
    const int plan = reaction_interval_plan(p, x, span, kla, ec);
    return plan < 0 ? plan : (plan & 127);
}
/* the same, returning the plan code (step count + 128 if slaved; -1: scheme 0) */
int sbro_reaction_interval_plan(const sbro_params* p, double* x, double span, double kla, double ec) {
    return reaction_interval_plan(p, x, span, kla, ec);
}

/* the idle phase (kind 2) under the handle's scheme: scheme 1 cuts its `rows` RK4-substep-long span into ceil(rows/10) macro
 * intervals of the adaptive scheme */
static void idle_span(const sbro_params* p, double* x, double span, int rows, double kla) {
    if (p->scheme == 1) b5a_span(p, 2, x, span, (rows + 9) / 10, kla, 0);
    else rk4_span(p, 2, x, span, rows, kla, 0, 0);
}

void sbro_rk4(const sbro_params* p, int kind, double* x, double span, int n, double kla, double ec,
              const double* loading) {
    rk4_span(p, kind, x, span, n, kla, ec, loading);
}

/* ---------------------------------------------------------------------------------- influent */
/* buffer_tank3.py:68-107 for one scenario.  means/stds: [14][48] (row 13 = flow q). */
void sbro_influent_mix(const double* means, const double* stds, const double* rnd, double* out) {
    double q[48], sq = 0;
    for (int k = 0; k < 48; ++k) q[k] = means[13 * 48 + k] + stds[13 * 48 + k] * rnd[k];
    for (int k = 0; k < 48; ++k) sq = sq + q[k];
    out[0] = 0.66;
    for (int j = 0; j < 13; ++j) {
        double s = 0;
        for (int k = 0; k < 48; ++k) s = s + (means[j * 48 + k] + stds[j * 48 + k] * rnd[k]) * q[k];
        out[1 + j] = s / sq;
    }
}

/* Philox4x32-10 (Salmon et al., SC'11), the generator sbr_reset uses on the device when no rnd
 * vector is passed.  counter = (i, 0, env_lo, env_hi), key = (seed_lo, seed_hi). */
static void philox4x32_10(uint32_t c[4], uint32_t k0, uint32_t k1) {
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c[0];
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * c[2];
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0;
        const uint32_t n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
        const uint32_t n3 = (uint32_t)p0;
        c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
}

static inline double u53(uint32_t hi, uint32_t lo) { /* (0,1] */
    const uint64_t v = (((uint64_t)hi << 32) | lo) >> 11;
    return ((double)v + 1.0) * (1.0 / 9007199254740992.0);
}

/* 48 standard normals for one env: Box-Muller on 24 Philox blocks (stream 0 = influent noise) */
void sbro_draw_normals(uint64_t seed, uint64_t env_id, double* out) {
    for (uint32_t i = 0; i < 24; ++i) {
        uint32_t c[4] = {i, 0u, (uint32_t)env_id, (uint32_t)(env_id >> 32)};
        philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
        const double u1 = u53(c[0], c[1]), u2 = u53(c[2], c[3]);
        const double rad = sqrt(-2.0 * log(u1)), ang = 6.283185307179586476925286766559 * u2;
        out[2 * i] = rad * cos(ang);
        out[2 * i + 1] = rad * sin(ang);
    }
}

/* uniform random action of call `step` (stream 1 = policy) */
void sbro_policy_action(const sbro_params* p, uint64_t seed, uint64_t env_id, uint32_t step, float* a) {
    uint32_t c[4] = {step, 1u, (uint32_t)env_id, (uint32_t)(env_id >> 32)};
    philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
    a[0] = (float)(u53(c[0], c[1]) * p->act_DO_max);
    a[1] = (float)(u53(c[2], c[3]) * p->act_EC_max);
}

/* scenario of one env for a reset with cfg.random_scenario = 1 (stream 2): uniform over the 8 influent scenarios, the
 * restatement of np.random.choice(8, 1) at gym_SBR_env4.py:107 with the product's counter-based generator */
int32_t sbro_scenario_draw(uint64_t seed, uint64_t env_id) {
    uint32_t c[4] = {0u, 2u, (uint32_t)env_id, (uint32_t)(env_id >> 32)};
    philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
    return (int32_t)(c[0] & 7u);
}

/* ---------------------------------------------------------------------------------- observations */
static double clip1(double v) { return v > 1 ? 1 : (v < -1 ? -1 : v); }

/* obs_DO ++ obs_EC (:1027-1114); t_obs and x are what the observation reports, xa..xb the xdot span */
static void build_obs(double t_obs, const double* xrep, const double* xa, const double* xb, double* obs) {
    obs[0] = t_obs / 0.5; obs[1] = xrep[5] / 2000; obs[2] = xrep[6] / 500; obs[3] = xrep[8] / 8.; obs[4] = xrep[10] / 10;
    obs[5] = clip1((xb[5] - xa[5]) / 4000); obs[6] = clip1((xb[6] - xa[6]) / 500);
    obs[7] = clip1((xb[8] - xa[8]) / 8); obs[8] = clip1((xb[10] - xa[10]) / 50);
    obs[9] = t_obs / 0.5; obs[10] = xrep[2] / 30; obs[11] = xrep[5] / 2000; obs[12] = xrep[9] / 10; obs[13] = xrep[10] / 10;
    obs[14] = clip1((xb[2] - xa[2]) / 50); obs[15] = clip1((xb[5] - xa[5]) / 4000);
    obs[16] = clip1((xb[9] - xa[9]) / 50); obs[17] = clip1((xb[10] - xa[10]) / 50);
}

static void build_state(double t_obs, const double* x, double* state) {
    state[0] = t_obs / X1_STATE[0];
    for (int i = 0; i < NX; ++i) state[1 + i] = x[i] / X1_STATE[1 + i];
}

/* ---------------------------------------------------------------------------------- reset */
/* SbrOS.reset :168-438 with Sim_filling :1585-1654.  influent_in[14]: flow-weighted influent
 * ([0] is overwritten by Qin/T_fill as at :287).  obs may be NULL. */
static void reset_from(const sbro_params* p, sbro_env* e, const double* influent_in, const double* x0, double iv, double* obs);

void sbro_reset(const sbro_params* p, sbro_env* e, const double* influent_in, double* obs) {
    reset_from(p, e, influent_in, p->x0, p->IV, obs);
}

/* multi-cycle operation: the new cycle starts from the env's own current state, x0 := x, IV := x[0]
 * (x0_new / IV_new, gym_SBR_env2.py:152-153; disabled upstream at gym_SBR_oneshot.py:260-268) */
void sbro_reset_carry(const sbro_params* p, sbro_env* e, const double* influent_in, double* obs) {
    double x0[NX];
    memcpy(x0, e->x, sizeof x0);
    reset_from(p, e, influent_in, x0, x0[0], obs);
}

static void reset_from(const sbro_params* p, sbro_env* e, const double* influent_in, const double* x0v, double iv, double* obs) {
    const double qin = p->WV - iv;
    memcpy(e->influent, influent_in, sizeof e->influent);
    e->influent[0] = qin / p->T_fill;
    memcpy(e->x, x0v, sizeof e->x);
    e->u_do = 0; e->u_ec = 15;
    /* DO-PID at t_start == 0: ie = 0, dcv = 0, set-point 0 (:1593-1617) */
    const double err = 0 - x0v[8];
    double ie = 0;
    double kla = p->Kc_DO * err + p->Kc_DO / p->tauI_DO * ie + p->Kc_DO * p->tauD_DO * 0 + 0;
    if (kla > p->Kla_max) { kla = p->Kla_max; ie = ie - err * p->dt; }
    if (kla < p->Kla_min) { kla = p->Kla_min; ie = ie - err * p->dt; }
    e->ie_do = ie; e->ie_ec = 0;
    const double t_end = 0 + p->T_fill;                 /* t_ratio[0]*0.5 == T_fill */
    const int n_rows = (int)((t_end - 0) / p->dt);      /* 252 */
    double x0c[NX];
    memcpy(x0c, e->x, sizeof x0c);
    rk4_span(p, 1, e->x, t_end, n_rows, kla, 0, e->influent);      /* the fill phase: RK4 under either scheme (not plannable) */
    e->so_m2 = x0v[8]; e->so_m1 = e->x[8];
    e->sno_m2 = x0v[9]; e->sno_m1 = e->x[2];          /* :1652 stores Ss in the Sno memory */
    e->t = t_end;
    /* Kla list = [0, kla] replicated (:323): the tail alternates, newest = kla */
    for (int j = 0; j < KLA_HIST; ++j) e->kla_hist[j] = ((KLA_HIST - 1 - j) % 2 == 0) ? kla : 0.0;
    e->kla_last = kla; e->ec_last = 0; e->ec_prev = 0;
    e->kla_sum = 0;                                      /* python sum() over [0, kla]*126 (:323), left to right */
    for (int j = 0; j < n_rows / 2; ++j) { e->kla_sum = e->kla_sum + 0.0; e->kla_sum = e->kla_sum + kla; }
    e->qw = 0; e->ret = 0; e->steps = 0; e->done = 0;
    e->status = status_bits(p, e->x, 0);
    memcpy(e->x_start, x0c, sizeof x0c);
    e->span = t_end; e->n_rows = n_rows; e->n_intervals = 0;
    if (obs) {
        /* volume blend of influent and post-fill state (:346-361) */
        double xr[NX];
        for (int i = 0; i < NX; ++i) xr[i] = (qin * e->influent[i] + e->x[i] * iv) / (qin + iv);
        build_obs(e->t, xr, x0c, e->x, obs);
    }
}

/* sticky domain-of-validity bits, same definition as SBR_ST_* in include/sbr_amd.h (not in the reference, which
 * has no guards: this only REPORTS that a concentration went negative / approached a pole of x/(K+x)) */
static double status_bits(const sbro_params* p, const double* x, double status) {
    int st = (int)status;
    const double lo = -1e-6;
    if (x[2] < lo || x[4] < lo || x[5] < lo || x[8] < lo || x[9] < lo || x[10] < lo) st |= 1;
    const double ko = p->Koh < p->Koa ? p->Koh : p->Koa;
    if (x[2] < -0.5 * p->Ks || x[8] < -0.5 * ko || x[9] < -0.5 * p->Kno || x[10] < -0.5 * p->Knh) st |= 2;
    double sum = 0;
    for (int i = 0; i < NX; ++i) sum += x[i];
    if (!(fabs(sum) < 1.7e308)) st |= 4;
    return (double)st;
}

/* ---------------------------------------------------------------------------------- interval */
/* Sim_aero_rxn :1877-1963 / Sim_anaero_rxn :1965-2051 + run_*_step :1331-1419 */
static void interval(const sbro_params* p, sbro_env* e, int aerobic) {
    const double t0 = e->t, t1 = t0 + p->t_delta;
    const int n_rows = (int)((t1 - t0) / p->dt);        /* 9 or 10, fp-dependent (:1339,:1384) */
    /* DO PID */
    const double sp = aerobic ? e->u_do : 0;
    const double err = sp - e->so_m1;
    const double dcv = (e->so_m1 - e->so_m2) / p->dt;
    e->ie_do = e->ie_do + err * p->dt;
    double kla = aerobic ? (p->Kc_DO * err + p->Kc_DO / p->tauI_DO * e->ie_do + p->Kc_DO * p->tauD_DO * dcv + e->kla_last) : 0;
    if (kla > p->Kla_max) { kla = p->Kla_max; e->ie_do = e->ie_do - err * p->dt; }
    if (kla < p->Kla_min) { kla = p->Kla_min; e->ie_do = e->ie_do - err * p->dt; }
    /* NO3 PID (error sign reversed) */
    const double err2 = e->sno_m1 - e->u_ec;
    const double dcv2 = (e->sno_m1 - e->sno_m2) / p->dt;
    e->ie_ec = e->ie_ec + err2 * p->dt;
    double ec = aerobic ? 0 : (p->Kc_EC * err2 + p->Kc_EC / p->tauI_EC * e->ie_ec + p->Kc_EC * p->tauD_EC * dcv2 + e->ec_last);
    if (ec < p->EC_min) { ec = p->EC_min; e->ie_ec = e->ie_ec - err2 * p->dt; }
    else if (ec > p->EC_max) { ec = p->EC_max; e->ie_ec = e->ie_ec - err2 * p->dt; }
    memcpy(e->x_start, e->x, sizeof e->x_start);
    {
        const int plan = reaction_interval_plan(p, e->x, t1 - t0, kla, ec);
        const int code = plan < 0 ? 0 : plan;
        e->scheme_steps = plan < 0 ? plan : (plan & 127);
        e->scheme_plan = e->n_intervals == 0 ? (code | (code << 8)) : ((e->scheme_plan & 0xff00) | code);
    }
    for (int j = 0; j < KLA_HIST - 1; ++j) e->kla_hist[j] = e->kla_hist[j + 1];
    e->kla_hist[KLA_HIST - 1] = kla;
    e->kla_sum = e->kla_sum + kla;
    e->ec_prev = e->ec_last; e->ec_last = ec; e->kla_last = kla;
    e->so_m2 = e->so_m1; e->so_m1 = e->x[8];
    e->sno_m2 = e->sno_m1; e->sno_m1 = e->x[9];
    e->t = t1; e->span = t1 - t0; e->n_rows = n_rows; e->n_intervals += 1;
    e->status = status_bits(p, e->x, e->status);
}

/* module_reward_continuous_G2ANET.py:4-45: piecewise-linear in Ss, So, Sno, Snh of the end state */
double sbro_reward_g2anet(const double* x) {
    const double ss = x[2], so = x[8], sno = x[9], snh = x[10];
    const double r_ec = ss < 0 ? 1 : -(ss - 0) / (10 - 0) + 1;
    const double r_e = so < 1.5 ? 0 : -(1 / (8 - 1.5)) * (so - 8) + 0;
    const double r_sno = sno < 4 ? 1 : -(sno - 4) / (20 - 4) + 1;
    const double r_snh = snh < 4 ? 1 : -(snh - 4) / (20 - 4) + 1;
    return (1 * r_ec + 1.5 * r_e + 2 * r_sno + 2 * r_snh) / 10;
}

/* module_reward_continuous.py:4-65 (the reward of SbrEnv3/SbrEnv4): operating cost only.  batch_type 0 = fill interval,
 * 1 = reaction interval (Kla[-1] of the list), 2 = end of cycle (sum(Kla) of the whole list, pumping, ammonia penalty).
 * The caller passes Kla[-1] and sum(Kla) instead of the list. */
double sbro_reward_oci(double so_sat, double kla_last, double kla_sum, int32_t batch_type, double qin, double qw,
                       double q_eff, double snh_eff) {
    const double t_delta = 0.002 / 24;
    double pe = 0, ae_dt = 0, r_snh = 0;
    if (batch_type == 0) { pe = 0.004 * qin; ae_dt = 1.32 * kla_last * t_delta; }
    if (batch_type == 1) { ae_dt = 1.32 * kla_last * t_delta; pe = 0; }
    if (batch_type == 2) {
        pe = (0.05 * qw + 0.004 * q_eff);
        ae_dt = 1.32 * kla_sum * t_delta;
        r_snh = snh_eff < 4 ? 0 : -246;
    }
    const double ae = so_sat / (1.8 * 1000) * (ae_dt);
    const double oci = ae + pe;
    const double r_oci = 0.5 - oci;
    return r_oci + r_snh;
}

/* module_reward_EQIOCI.py:4-115 */
static double reward_of(const sbro_params* p, const sbro_env* e) {
    const double* x = e->x;
    if (p->reward_kind == 1) return sbro_reward_g2anet(x);
    if (p->reward_kind == 2) return sbro_reward_oci(p->So_sat, e->kla_last, 0, 1, 0, 0, 0, 0);      /* reaction interval */
    const double xi = x[3], xs = x[4], xbh = x[5], xba = x[6], xp = x[7];
    const double snkj = x[10] + x[11] + x[12] + 0.08 * (xbh + xba) + 0.06 * (xp + xi);
    const double ss_ = 0.75 * (xs + xi + xbh + xba + xp);
    const double bod5 = 0.25 * (x[2] + xs + (1 - 0.08) * (xbh + xba));
    const double cod = x[2] + x[1] + xs + xi + xbh + xba + xp;
    const double eqi = (2 * ss_ + 1 * cod + 30 * snkj + 10 * x[9] + 2 * bod5) * (1.0 / 1000) * 0.66;
    const double eqi2 = eqi / 10;
    const double td = 0.002 / 24;
    const int n = e->n_rows;
    double ksum = 0;                                     /* Kla[-n:-1]: the n-1 values before the current */
    for (int j = KLA_HIST - n; j < KLA_HIST - 1; ++j) ksum = ksum + e->kla_hist[j];
    const double ae = 8 / (e->span * 1.8 * 1000) * (1.32 * ksum * td);
    double esum = 0 + e->ec_prev;                        /* EC[-n:-1]: previous interval's last + (n-2) current */
    for (int j = 0; j < n - 2; ++j) esum = esum + e->ec_last;
    const double ec_oci = p->EC_conc * esum * td / (e->span * 1000);
    const double oci = ae + ec_oci;
    return (1 - (eqi2 * eqi2 + oci * oci)) / 473;
}

/* the four diagnostics module_reward_EQIOCI.py:109-112 appends per call: EQI2 (:60), OCI2 = AE_OCI2 + EC_OCI2 (:101),
 * AE_OCI2 = AE_OCI/AE_OCI_max (:72-73), EC_OCI2 = EC_OCI/EC_OCI_max (:80-81); evaluated on the env as sbro_step left it */
void sbro_reward_parts(const sbro_params* p, const sbro_env* e, double* out4) {
    const double* x = e->x;
    const double xi = x[3], xs = x[4], xbh = x[5], xba = x[6], xp = x[7];
    const double snkj = x[10] + x[11] + x[12] + 0.08 * (xbh + xba) + 0.06 * (xp + xi);
    const double ss_ = 0.75 * (xs + xi + xbh + xba + xp);
    const double bod5 = 0.25 * (x[2] + xs + (1 - 0.08) * (xbh + xba));
    const double cod = x[2] + x[1] + xs + xi + xbh + xba + xp;
    const double eqi = (2 * ss_ + 1 * cod + 30 * snkj + 10 * x[9] + 2 * bod5) * (1.0 / 1000) * 0.66;
    const double td = 0.002 / 24, so_sat = 8;
    const int n = e->n_rows;
    double ksum = 0;
    for (int j = KLA_HIST - n; j < KLA_HIST - 1; ++j) ksum = ksum + e->kla_hist[j];
    const double ae = so_sat / (e->span * 1.8 * 1000) * (1.32 * ksum * td);
    const double ae_max = 1.32 * (240 * 11) * td * (so_sat / ((td * 11) * 1.8 * 1000));
    double esum = 0 + e->ec_prev;
    for (int j = 0; j < n - 2; ++j) esum = esum + e->ec_last;
    const double ec_oci = p->EC_conc * esum * td / (e->span * 1000);
    const double ec_max = p->EC_conc * (0.0005 * 11) * td / ((td * 11) * 1000);
    out4[0] = eqi / 10; out4[2] = ae / ae_max; out4[3] = ec_oci / ec_max; out4[1] = out4[2] + out4[3];
}

/* ---------------------------------------------------------------------------------- terminal */
/* settle (:2171-2262, closed form of the linear layer system, see oracle/sbr_ref.py
 * settle_closed_form), draw/waste (:2327-2393), idle (:2554-2597) */
static void terminal(const sbro_params* p, sbro_env* e) {
    double* x = e->x;
    const double xf = 0.75 * (x[3] + x[4] + x[5] + x[6] + x[7]);
    const double vs = x[0], z = vs / p->settler_area;
    const double t_set = p->t_settle * p->t_cycle;
    const double a = p->settler_vmax / z * t_set, ea = exp(-a);
    double sx[10], term = 1.0, partial = 0.0, others = 0.0;
    for (int j = 0; j < 9; ++j) { partial += term; sx[9 - j] = xf * ea * partial; term *= a / (j + 1); }
    for (int j = 1; j < 10; ++j) others += sx[j];
    sx[0] = 10.0 * xf - others;
    const double t_after_draw = (e->t + t_set) + p->t_draw * p->t_cycle;
    const double layer_v = vs / 10;
    double resid_v = vs - p->Qeff;
    int m = (int)ceil(nearbyint(p->Qeff / layer_v));     /* python round() = half-to-even */
    if (m < 1) m = 1;
    if (m > 9) m = 9;
    double w[10], rs[10], wsum = 0;
    for (int i = 0; i < 10 - m; ++i) { w[i] = layer_v * sx[i]; rs[i] = sx[i]; }
    for (int i = 0; i < 10 - m; ++i) wsum = wsum + w[i];
    double waste = wsum - p->biomass_setpoint * resid_v;
    double qw = NAN;
    for (int i = 0; i < 10 - m; ++i) {
        const double rest = waste - w[i];
        if (rest > 0) { waste = rest; rs[i] = 0; w[i] = 0; resid_v -= layer_v; }
        else {
            qw = waste / (rs[i] - p->biomass_setpoint);
            w[i] = w[i] - qw * rs[i];
            resid_v -= qw;
            rs[i] = w[i] / (layer_v - qw);
            break;
        }
    }
    wsum = 0;
    for (int i = 0; i < 10 - m; ++i) wsum = wsum + w[i];
    const double sx2 = wsum / resid_v;
    x[0] = resid_v;
    for (int i = 3; i <= 7; ++i) x[i] = x[i] * (1 / 0.75) * sx2 / xf;
    e->qw = qw;
    /* idle: one DO-PID update (So[-1] == So[-2] == x[8] after settle/draw), then conversion only */
    const double err = e->u_do - x[8];
    e->ie_do = e->ie_do + err * p->dt;
    double kla = p->Kc_DO * err + p->Kc_DO / p->tauI_DO * e->ie_do + p->Kc_DO * p->tauD_DO * 0.0 + e->kla_last;
    if (kla > p->Kla_max) { kla = p->Kla_max; e->ie_do = e->ie_do - err * p->dt; }
    if (kla < p->Kla_min) { kla = p->Kla_min; e->ie_do = e->ie_do - err * p->dt; }
    const int n_rows = (int)((p->t_cycle - t_after_draw) / p->dt);
    idle_span(p, x, p->t_cycle - t_after_draw, n_rows, kla);
    for (int j = 0; j < KLA_HIST - 1; ++j) e->kla_hist[j] = e->kla_hist[j + 1];   /* Kla.append, :2578 */
    e->kla_hist[KLA_HIST - 1] = kla;
    e->kla_sum = e->kla_sum + kla;
    e->kla_last = kla;
}

/* ---------------------------------------------------------------------------------- step */
/* SbrOS.step :843-1273.  obs[18], state[15] may be NULL. */
void sbro_step(const sbro_params* p, sbro_env* e, const double* action, double* obs, double* state, double* reward,
               uint8_t* done) {
    if (e->done != 0) {                                   /* finished env: wait for reset (reference: caller resets) */
        if (reward) *reward = 0;
        if (done) *done = 1;
        if (obs) build_obs(p->t_cycle, e->x, e->x, e->x, obs);
        if (state) build_state(p->t_cycle, e->x, state);
        return;
    }
    double a0 = action[0], a1 = action[1];   /* float64, like the reference; the product takes float32 */
    a0 = a0 < 0 ? 0 : (a0 > p->act_DO_max ? p->act_DO_max : a0);
    a1 = a1 < 0 ? 0 : (a1 > p->act_EC_max ? p->act_EC_max : a1);
    e->n_intervals = 0; e->scheme_plan = 0;
    /* four sequential tests on the running time (:860, :896, :931, :963) */
    if (e->t < p->T3_0) { e->u_ec = a1; e->u_do = 0; interval(p, e, 0); }
    if (e->t >= p->T3_0 && e->t <= p->T3_end) { e->u_do = a0; e->u_ec = 0; interval(p, e, 1); }
    if (e->t > p->T3_end && e->t <= p->T4_end) { e->u_ec = a1; e->u_do = 0; interval(p, e, 0); }
    if (e->t > p->T4_end) { e->u_do = a0; e->u_ec = 0; interval(p, e, 1); }
    double r = reward_of(p, e);
    double t_obs = e->t;
    double xa[NX];
    memcpy(xa, e->x_start, sizeof xa);
    uint8_t dn = 0;
    if (e->t >= p->T5_end) {                              /* :1122 */
        dn = 1; e->done = 1;
        if (p->terminal) {
            memcpy(xa, e->x, sizeof xa); terminal(p, e); t_obs = p->t_cycle;
            /* reward_kind 2: the cycle's last reward is the end-of-cycle branch, which needs Qw, the effluent
             * (eff_component[0] = Qeff, [3] = Snh of the drawn water = the pre-settle Snh) and sum(Kla) incl. idle */
            if (p->reward_kind == 2) r = sbro_reward_oci(p->So_sat, e->kla_last, e->kla_sum, 2, 0, e->qw, p->Qeff, xa[10]);
        }
    }
    e->ret += r; e->steps += 1;
    if (obs) build_obs(t_obs, e->x, xa, e->x, obs);
    if (state) build_state(t_obs, e->x, state);
    if (reward) *reward = r;
    if (done) *done = dn;
}

/* ---------------------------------------------------------------------------------- batched */
void sbro_batch_reset(const sbro_params* p, int64_t n, sbro_env* envs, const double* influent /* [n][14] */,
                      double* obs /* [n][18] or NULL */, int nthreads) {
#pragma omp parallel for num_threads(nthreads) schedule(static)
    for (int64_t i = 0; i < n; ++i) sbro_reset(p, envs + i, influent + i * NX, obs ? obs + i * 18 : 0);
}

void sbro_batch_reset_carry(const sbro_params* p, int64_t n, sbro_env* envs, const double* influent, double* obs, int nthreads) {
#pragma omp parallel for num_threads(nthreads) schedule(static)
    for (int64_t i = 0; i < n; ++i) sbro_reset_carry(p, envs + i, influent + i * NX, obs ? obs + i * 18 : 0);
}

void sbro_batch_step(const sbro_params* p, int64_t n, sbro_env* envs, const double* action, double* obs, double* state,
                     double* reward, uint8_t* done, int nthreads) {
#pragma omp parallel for num_threads(nthreads) schedule(static)
    for (int64_t i = 0; i < n; ++i)
        sbro_step(p, envs + i, action + 2 * i, obs ? obs + 18 * i : 0, state ? state + 15 * i : 0,
                  reward ? reward + i : 0, done ? done + i : 0);
}

/* n_steps calls with the on-device random policy's actions (same Philox stream as sbr_rollout) */
void sbro_batch_rollout(const sbro_params* p, int64_t n, sbro_env* envs, int64_t first_env_id, int32_t n_steps,
                        uint64_t policy_seed, double* returns, int nthreads) {
#pragma omp parallel for num_threads(nthreads) schedule(static)
    for (int64_t i = 0; i < n; ++i) {
        double acc = 0;
        for (int32_t s = 0; s < n_steps; ++s) {
            float a[2];
            double r, ad[2];
            const uint32_t call = (uint32_t)envs[i].steps;
            sbro_policy_action(p, policy_seed, (uint64_t)(first_env_id + i), call, a);
            ad[0] = a[0]; ad[1] = a[1];                 /* the device policy samples float32 actions */
            sbro_step(p, envs + i, ad, 0, 0, &r, 0);
            acc += r;
        }
        if (returns) returns[i] = acc;
    }
}


/* =================================================================================== per-cycle env SBR-v2
 * SbrEnv2.step (gym_SBR_env2.py:131-171) = SBR_model_FB.run (SBR_model_FB.py:8-295): five PID-controlled phases
 * (sub_phases_FB.py filling.sim_rxn :178-271, rxn.sim_rxn :406-500), settle (:716-775, closed form), draw + effluent
 * quality (:780-915), aerated idle, reward module_reward.py:4-51.  RK4 with p->substeps substeps per control interval. */
#define NCYC_DIAG 12     /* qw, EQI, OCI, eff[1..5] = Ntot COD Snh BOD5 Sno, mean Kla of phases 3, 5, 8, Xf */

/* numpy.linspace(a, b, n)[i] */
static double lin(double a, double b, int n, int i) {
    if (i == n - 1) return b;
    const double step = (b - a) / (double)(n - 1);
    return (double)i * step + a;
}

/* one phase: positional PID with bias Kla[0]; interval 0 overwrites Kla[0], so later intervals use the controlled
 * value of interval 0 as bias (sub_phases_FB.py:219,243).  Returns the last Kla; *ksum gets sum(Kla), *n_iv the count. */
static double cycle_phase(const sbro_params* p, double* x, double t_start, double t_end, double t_delta, double sp,
                          double kla_in, const double* loading, double* ksum, int* n_iv_out, double* kla_log) {
    const int n2 = (int)((t_end - t_start) / (t_delta * 10));
    const int n_iv = n2 - 1;
    double so = x[8], so_prev = x[8], ie = 0, bias = kla_in, k = kla_in, sum = 0;
    for (int i = 0; i < n_iv; ++i) {
        const double g0 = lin(t_start, t_end, n2, i), g1 = lin(t_start, t_end, n2, i + 1);
        const double e = sp - so;
        double dcv = 0;
        if (i >= 1) { dcv = (so - so_prev) / p->cyc_dt; ie = ie + e * p->cyc_dt; }
        k = p->cyc_Kc * e + p->cyc_Kc / p->cyc_tauI * ie + p->cyc_Kc * p->cyc_tauD * dcv + bias;
        if (k > p->Kla_max) { k = p->Kla_max; ie = ie - e * p->cyc_dt; }
        if (k < p->Kla_min) { k = p->Kla_min; ie = ie - e * p->cyc_dt; }
        if (i == 0) bias = k;
        if (p->scheme == 1 && !loading) b5a_span(p, 2, x, g1 - g0, 1, k, 0);     /* scheme 1: every interval but the fill phase's */
        else rk4_span(p, loading ? 1 : 2, x, g1 - g0, p->substeps, k, 0, loading);
        sum = sum + k;
        if (kla_log) kla_log[i] = k;
        so_prev = so; so = x[8];
    }
    *ksum = sum; *n_iv_out = n_iv;
    return k;
}

/* x: start state in, end-of-cycle state out.  influent[14] ([0] is replaced by Qin/t_phs1).  action[3] in [0,1].
 * state3 = [Qeff, COD_eff, Snh_eff/30]; diag[NCYC_DIAG]; kla_log (6 x 256 doubles, phases 1..5 and 8) may be NULL. */
void sbro_cycle_step(const sbro_params* p, double* x, const double* influent, const double* action, double* state3,
                     double* reward, double* diag, double* kla_log) {
    const double t_delta = 0.002 / 24;
    double a[3], sp[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tph[8], loading[NX];
    for (int j = 0; j < 3; ++j) a[j] = action[j] < 0 ? 0 : (action[j] > 1 ? 1 : action[j]);
    sp[2] = a[0] * 8; sp[4] = a[1] * 8; sp[7] = a[2] * 8;
    for (int j = 0; j < 8; ++j) tph[j] = p->t_cycle * p->t_ratio[j];
    const double iv = x[0], qin = p->WV - iv;
    memcpy(loading, influent, sizeof loading);
    loading[0] = qin / (p->t_cycle * p->t_ratio[0]);
    double ksum[6], klast = 0;
    int niv[6];
    double t_start = 0, t_end = 0 + tph[0];
    klast = cycle_phase(p, x, t_start, t_end, t_delta, sp[0], 0.0, loading, &ksum[0], &niv[0], kla_log ? kla_log : 0);
    for (int ph = 1; ph <= 4; ++ph) {
        t_start = t_end + t_delta; t_end = t_start + tph[ph];
        klast = cycle_phase(p, x, t_start, t_end, t_delta, sp[ph], klast, 0, &ksum[ph], &niv[ph], kla_log ? kla_log + 256 * ph : 0);
    }
    const double kla5_last = klast;
    /* settle */
    t_start = t_end + t_delta; t_end = t_start + tph[5];
    const double xf = 0.75 * (x[3] + x[4] + x[5] + x[6] + x[7]);
    const double vs = x[0], z = vs / p->settler_area;
    const double aa = p->settler_vmax / z * (t_end - t_start), ea = exp(-aa);
    double sx[10], term = 1.0, partial = 0.0, others = 0.0;
    for (int j = 0; j < 9; ++j) { partial += term; sx[9 - j] = xf * ea * partial; term *= aa / (j + 1); }
    for (int j = 1; j < 10; ++j) others += sx[j];
    sx[0] = 10.0 * xf - others;
    /* draw */
    t_start = t_end + t_delta; t_end = t_start + tph[6];
    const double layer_v = vs / 10;
    double resid_v = vs - p->Qeff;
    int m = (int)ceil(nearbyint(p->Qeff / layer_v));
    if (m < 1) m = 1;
    if (m > 9) m = 9;
    double sx_eff = 0;
    for (int i = 10 - m; i < 9; ++i) sx_eff = sx_eff + sx[i] * layer_v;          /* sum(sX[-m:-1]*layer_volume) */
    double xe[NX];
    memcpy(xe, x, sizeof xe);
    xe[0] = p->Qeff;
    for (int i = 3; i <= 7; ++i) xe[i] = xe[i] * (1 / 0.75) * sx_eff / xf;
    double w[10], rs[10], wsum = 0;
    for (int i = 0; i < 10 - m; ++i) { w[i] = layer_v * sx[i]; rs[i] = sx[i]; }
    for (int i = 0; i < 10 - m; ++i) wsum = wsum + w[i];
    double waste = wsum - p->biomass_setpoint * resid_v, qw = NAN;
    for (int i = 0; i < 10 - m; ++i) {
        const double rest = waste - w[i];
        if (rest > 0) { waste = rest; rs[i] = 0; w[i] = 0; resid_v -= layer_v; }
        else { qw = waste / (rs[i] - p->biomass_setpoint); w[i] = w[i] - qw * rs[i]; resid_v -= qw; rs[i] = w[i] / (layer_v - qw); break; }
    }
    wsum = 0;
    for (int i = 0; i < 10 - m; ++i) wsum = wsum + w[i];
    const double sx2 = wsum / resid_v;
    /* effluent quality on xe (cal_eq :860-915) */
    const double snkj = xe[10] + xe[11] + xe[12] + 0.08 * (xe[5] + xe[6]) + 0.06 * (xe[7] + xe[3]);
    const double ntot = xe[9] + snkj;
    const double ss_ = 0.75 * (xe[4] + xe[3] + xe[5] + xe[6] + xe[7]);
    const double bod5 = 0.25 * (xe[2] + xe[4] + (1 - 0.08) * (xe[5] + xe[6]));
    const double cod = xe[2] + xe[1] + xe[4] + xe[3] + xe[5] + xe[6] + xe[7];
    const double eqi = (2 * ss_ + 1 * cod + 30 * snkj + 10 * xe[9] + 2 * bod5) * (1.0 / 1000) * 0.66;
    const double snh_eff = xe[10], sno_eff = xe[9];
    x[0] = resid_v;
    for (int i = 3; i <= 7; ++i) x[i] = x[i] * (1 / 0.75) * sx2 / xf;
    /* aerated idle from the drawn reactor, bias = last Kla of phase 5 */
    t_start = t_end + t_delta; t_end = t_start + tph[7];
    cycle_phase(p, x, t_start, t_end, t_delta, sp[7], kla5_last, 0, &ksum[5], &niv[5], kla_log ? kla_log + 256 * 5 : 0);
    /* reward */
    const double td = 0.002 / 24;
    const double ae3 = 1.32 * ksum[2] * td / (niv[2] * td);
    const double ae5 = 1.32 * ksum[4] * td / (niv[4] * td);
    const double ae8 = (1.32 - qw) * ksum[5] * td / (niv[5] * td);
    const double ae = p->So_sat / (1.8 * 1000) * (ae3 + ae5 + ae8);
    const double pe = (0.004 * qin + 0.05 * qw + 0.004 * p->Qeff);
    const double me = 0.005 * 1.32 * 24 + 0.005 * 1.32 * 24;
    const double oci = ae + pe + me;
    *reward = (5 - oci) + (snh_eff < 4 ? 0 : -20);
    state3[0] = p->Qeff; state3[1] = cod; state3[2] = snh_eff / 30;
    if (diag) {
        diag[0] = qw; diag[1] = eqi; diag[2] = oci; diag[3] = ntot; diag[4] = cod; diag[5] = snh_eff; diag[6] = bod5;
        diag[7] = sno_eff; diag[8] = ksum[2] / niv[2]; diag[9] = ksum[4] / niv[4]; diag[10] = ksum[5] / niv[5]; diag[11] = xf;
    }
}

/* reset observation of SbrEnv2 (gym_SBR_env2.py:108-119): sums of start state and influent */
void sbro_cycle_reset_state(const double* x0, const double* influent, double* state3) {
    double tot[NX];
    for (int i = 0; i < NX; ++i) tot[i] = x0[i] + influent[i];
    const double cod = tot[1] + tot[2] + tot[3] + tot[4] + tot[5] + tot[6] + tot[7];
    state3[0] = tot[0]; state3[1] = (cod - 5145) / 10; state3[2] = tot[10] / 30;
}

void sbro_batch_cycle_step(const sbro_params* p, int64_t n, double* x /* [n][14] */, const double* influent /* [n][14] */,
                           const double* action /* [n][3] */, double* state3, double* reward, double* diag, int nthreads) {
#pragma omp parallel for num_threads(nthreads) schedule(static)
    for (int64_t i = 0; i < n; ++i)
        sbro_cycle_step(p, x + i * NX, influent + i * NX, action + 3 * i, state3 + 3 * i, reward + i,
                        diag ? diag + NCYC_DIAG * i : 0, 0);
}
