// sbr_device.h - device functions of the batched SBR environment (gfx950 / CDNA4, fp64 VALU).
//
// Mapping: ONE LANE PER ENVIRONMENT.  The plant (14 states) and both controllers of an env live in
// that lane's VGPRs; every constant of the model is wave-uniform and arrives through the kernel
// argument segment, i.e. in SGPRs via scalar loads (no VGPR and no LDS traffic for constants).
// Plant/controller state is struct-of-arrays [field][N] float64 in HBM, so a wave's load of one field
// is one contiguous 512-byte segment.  There is nothing GEMM-shaped here: no MFMA.
//
// Every function cites the reference lines it replaces
// (/root/reference/gym_SBR/envs/gym_SBR_oneshot.py unless another file is named).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/sbr_amd.h"

#define SBR_DEV __device__ __forceinline__

// ---------------------------------------------------------------------------------------------------
// Wave-uniform model constants: sbr_config + everything that can be folded on the host (in fp64,
// with the same expressions the reference evaluates per RHS call, :1689-1725).
struct SbrPar {
    // kinetics
    double muH, Ks, Koh, Kno, bH, eta_g, eta_h, kh, Kx, muA, Knh, bA, Koa, ka;
    // stoichiometry, folded
    double n2_12;    // -1/Yh                       (nu2_1 = nu2_2)
    double n4_45;    // 1 - ixp                     (nu4_4 = nu4_5)
    double n7_45;    // ixp
    double n8_1;     // -(1-Yh)/Yh
    double n8_3;     // -(4.57-Ya)/Ya
    double n9_2;     // -((1-Yh)/(2.86*Yh))
    double n9_3;     // 1/Ya
    double n10_12;   // -ixb
    double n10_3;    // -ixb - 1/Ya
    double n12_45;   // ixb - fp*ixp
    double n13_1;    // -ixb/14
    double n13_2;    // (1-Yh)/(14*2.86*Yh) - ixb/14
    double n13_3;    // -ixb/14 - 1/(7*Ya)
    double n13_6;    // 1/14
    // plant, time grid
    double WV, IV, dt, t_delta, t_cycle, T_fill, T3_0, T3_end, T4_end, T5_end, t_settle, t_draw;
    double qin;          // WV - IV
    double load0;        // qin / T_fill                                   (:287)
    // controllers
    double So_sat, Kla_min, Kla_max, Kc_DO, KcI_DO, KcD_DO;   // KcI = Kc/tauI, KcD = Kc*tauD  (:1898-1900)
    double EC_min, EC_max, Kc_EC, KcI_EC, KcD_EC, EC_conc;
    double act_DO_max, act_EC_max;
    // terminal
    double biomass_setpoint, Qeff, settler_area, settler_vmax;
    double x0[SBR_NX];
    double muH_etag;     // muH * eta_g
    double t_ph[8];      // phase lengths t_cycle * t_ratio[k]                 (SBR_model_FB.py:18-27)
    double cyc_Kc, cyc_KcI, cyc_KcD, cyc_dt;   // positional PID of the per-cycle env (sub_phases_FB.py:205-243)
    int32_t substeps, terminal, fill_rows, reward_kind;
};

// ---------------------------------------------------------------------------------------------------
// 1/d for the Monod denominators: v_rcp_f64 seed (measured on MI355X: relative error <= 2^-24.4) and ONE Newton step
// carried to second order, r0*(1 + e + e^2) with e = 1 - d*r0: the truncation error e^3 ~ 1e-22 is far below half an
// ulp, and the result equalled IEEE 1/d on all 4.2 M denominators probed (scripts/probes/rcp_accuracy.hip; two plain
// Newton steps give the same, one plain step leaves 2.2e-15).  4 instructions and a 4-deep dependent chain, against
// ~12 for LLVM's full f64 division (v_div_scale x2, v_rcp, ~6 fma, v_div_fmas, v_div_fixup), whose range handling
// the denominators here (K + x, O(0.1 .. 3000)) do not need.
SBR_DEV double sbr_rcp(double d) {
    const double r = __builtin_amdgcn_rcp(d);
    const double e = __builtin_fma(-d, r, 1.0);
    return __builtin_fma(r, __builtin_fma(e, e, e), r);
}

// Conversion rates r[i] of the 11 reacting components (Si, Xi and V do not react), incl. aeration.
// Process rates :1660-1685, combination :1731-1755.  The reference's ten quotients are evaluated with SEVEN shared
// reciprocals: So/(Koh+So) and Koh/(Koh+So) share one, and (Xs/Xbh)/(Kx + Xs/Xbh) is written Xs/(Kx*Xbh + Xs).
// Equal coefficients are factored (nu2_1 = nu2_2, nu4_4 = nu4_5, ...).  Parity is to tolerance, not bitwise.
template <bool WITH_V>
SBR_DEV double sbr_conversion(const SbrPar& p, const double (&x)[SBR_NX], double kla, double (&r)[SBR_NX]) {
    const double ss = x[2], xs = x[4], xbh = x[5], xba = x[6], so = x[8], sno = x[9], snh = x[10], snd = x[11],
                 xnd = x[12];
#ifndef SBR_BATCH_RCP
#define SBR_BATCH_RCP 1
#endif
#ifndef SBR_RHO8_DIRECT
#define SBR_RHO8_DIRECT 1
#endif
#if SBR_BATCH_RCP
    // v_rcp_f64 issues in ~32 cycles on gfx950 (8 FMA slots; measured: 392 + 28 x 32 cycles = the 1.02 us of a substep), so the
    // reciprocals are taken from ONE: R = 1/(d1 d2 ... d6), then 1/dk = R x (product of the others), peeled off with two
    // multiplications each (Montgomery's trick): 15 multiplications + 1 reciprocal instead of 6 reciprocals.  Rounding
    // grows to ~7 ulp (parity bounds are >= 1e-11 relative); the product (~1e9) cannot over- or underflow for finite states.
    // The reference's seventh quotient, Xnd/Xs in rho8 = (Xnd/Xs) rho7 (:1685), needs no reciprocal: rho7 carries the factor
    // Xs, so rho8 = Xnd x (rho7 / Xs) is formed from the common factor c7 (SBR_RHO8_DIRECT; same value to rounding, and the
    // continuous extension at Xs -> 0 where the reference evaluates 0/0).  Measured: -1.6 % per k_step launch, -4 % fused.
    const double d1 = p.Ks + ss, d2 = p.Koh + so, d3 = p.Kno + sno, d4 = p.Knh + snh, d5 = p.Koa + so;
    const double d6 = __builtin_fma(p.Kx, xbh, xs), d7 = xs;
#if SBR_BATCH_RCP == 2
    // two independent chains (4 + 3 denominators): one more reciprocal, but half the serial dependency depth
    const double q2 = d1 * d2, q3 = q2 * d3;
    double Ra = sbr_rcp(q3 * d4);
    const double rd = Ra * q3; Ra = Ra * d4;
    const double rc = Ra * q2; Ra = Ra * d3;
    const double rb = Ra * d1, ra = Ra * d2;
    const double rv = WITH_V ? sbr_rcp(x[0]) : 0.0;
    const double s2 = d5 * d6;
    double Rb = sbr_rcp(s2 * d7);
    const double rg = Rb * s2; Rb = Rb * d7;
    const double re = Rb * d6, rf = Rb * d5;
#else
    const double p2 = d1 * d2, p3 = p2 * d3, p4 = p3 * d4, p5 = p4 * d5, p6 = p5 * d6;
    double R, rv = 0.0;
#if SBR_RHO8_DIRECT
    // rho8 = (Xnd/Xs) rho7 and rho7 = kh Xs/(Kx Xbh + Xs) [..] Xbh: Xs cancels, so 1/Xs is not needed at all
    if (WITH_V) {
        R = sbr_rcp(p6 * x[0]);
        rv = R * p6; R = R * x[0];
    } else {
        R = sbr_rcp(p6);
    }
    (void)d7;
#else
    if (WITH_V) {                        // dosing / fill: 1/V for the dilution terms rides in the same batch
        const double p7 = p6 * d7;
        R = sbr_rcp(p7 * x[0]);
        rv = R * p7; R = R * x[0];
    } else {
        R = sbr_rcp(p6 * d7);
    }
    const double rg = R * p6; R = R * d7;
#endif
    const double rf = R * p5; R = R * d6;
    const double re = R * p4; R = R * d5;
    const double rd = R * p3; R = R * d4;
    const double rc = R * p2; R = R * d3;
    const double rb = R * d1, ra = R * d2;
#endif
#else
    const double ra = sbr_rcp(p.Ks + ss);
    const double rb = sbr_rcp(p.Koh + so);
    const double rc = sbr_rcp(p.Kno + sno);
    const double rd = sbr_rcp(p.Knh + snh);
    const double re = sbr_rcp(p.Koa + so);
    const double rf = sbr_rcp(__builtin_fma(p.Kx, xbh, xs));
    const double rg = sbr_rcp(xs);
    const double rv = WITH_V ? sbr_rcp(x[0]) : 0.0;
#endif
    const double m_so = so * rb;                     // So/(Koh+So)
    const double inox = (p.Koh * rb) * (sno * rc);   // Koh/(Koh+So) * Sno/(Kno+Sno)
    const double g = (ss * ra) * xbh;                // Ss/(Ks+Ss) * Xbh
    const double rho1 = p.muH * g * m_so;
    const double rho2 = p.muH_etag * g * inox;
    const double rho3 = p.muA * (snh * rd) * (so * re) * xba;
    const double rho4 = p.bH * xbh;
    const double rho5 = p.bA * xba;
    const double rho6 = p.ka * snd * xbh;
#if SBR_RHO8_DIRECT && SBR_BATCH_RCP == 1
    const double c7 = p.kh * rf * __builtin_fma(p.eta_h, inox, m_so) * xbh;
    const double rho7 = xs * c7, rho8 = xnd * c7;
#else
    const double rho7 = p.kh * (xs * rf) * __builtin_fma(p.eta_h, inox, m_so) * xbh;
    const double rho8 = (xnd * rg) * rho7;
#endif
    const double s12 = rho1 + rho2, s45 = rho4 + rho5;
    r[0] = 0.0; r[1] = 0.0; r[3] = 0.0;
    r[2] = __builtin_fma(p.n2_12, s12, rho7);
    r[4] = __builtin_fma(p.n4_45, s45, -rho7);
    r[5] = s12 - rho4;
    r[6] = rho3 - rho5;
    r[7] = p.n7_45 * s45;
    r[8] = p.n8_1 * rho1 + p.n8_3 * rho3 + kla * (p.So_sat - so);
    r[9] = p.n9_2 * rho2 + p.n9_3 * rho3;
    r[10] = p.n10_12 * s12 + p.n10_3 * rho3 + rho6;
    r[11] = rho8 - rho6;
    r[12] = __builtin_fma(p.n12_45, s45, -rho8);
    r[13] = p.n13_1 * rho1 + p.n13_2 * rho2 + p.n13_3 * rho3 + p.n13_6 * rho6;
    return rv;                           // 1/V if WITH_V
}

// Right-hand sides.  KIND 0: reaction_dxdt :1658-1787 (dosing ec, dilution ec/V)
//                    KIND 1: filling_dxdt :1424-1583 at EC = 0 (loading vector ld, ld[0] = inflow)
//                    KIND 2: idle_dxdt :2424-2552 (conversion only)
//                    KIND 3: reaction with ec == 0 for every lane of the wave: V, Si, Xi are constant
template <int KIND>
SBR_DEV void sbr_rhs(const SbrPar& p, const double (&x)[SBR_NX], double kla, double ec, const double (&ld)[SBR_NX],
                     double (&d)[SBR_NX]) {
    double r[SBR_NX];
    const double rv = sbr_conversion<(KIND == 0 || KIND == 1)>(p, x, kla, r);
    if (KIND == 0) {
        const double q = ec * rv;
        d[0] = ec;
#pragma unroll
        for (int i = 1; i < SBR_NX; ++i) d[i] = r[i] + q * (i == 2 ? (p.EC_conc - x[i]) : (-x[i]));
    } else if (KIND == 1) {
        const double q = ld[0] * rv;
        d[0] = ld[0];
#pragma unroll
        for (int i = 1; i < SBR_NX; ++i) d[i] = r[i] + q * (ld[i] - x[i]);
    } else {
#pragma unroll
        for (int i = 0; i < SBR_NX; ++i) d[i] = r[i];
    }
}

// Classical RK4, n equal substeps over `span`; autonomous inside a span (Kla, EC held).  Low-storage
// form: x, the running combination and one stage vector are live (3 x 14 doubles).
template <int KIND>
SBR_DEV void sbr_rk4(const SbrPar& p, double (&x)[SBR_NX], double span, int n, double kla, double ec,
                     const double (&ld)[SBR_NX]) {
    const double h = span / (double)n;
    const double h2 = 0.5 * h, h6 = h / 6.0, h3 = h / 3.0;
    for (int s = 0; s < n; ++s) {
        double k[SBR_NX], y[SBR_NX], acc[SBR_NX];
        sbr_rhs<KIND>(p, x, kla, ec, ld, k);
#pragma unroll
        for (int i = 0; i < SBR_NX; ++i) { acc[i] = x[i] + h6 * k[i]; y[i] = x[i] + h2 * k[i]; }
        sbr_rhs<KIND>(p, y, kla, ec, ld, k);
#pragma unroll
        for (int i = 0; i < SBR_NX; ++i) { acc[i] += h3 * k[i]; y[i] = x[i] + h2 * k[i]; }
        sbr_rhs<KIND>(p, y, kla, ec, ld, k);
#pragma unroll
        for (int i = 0; i < SBR_NX; ++i) { acc[i] += h3 * k[i]; y[i] = x[i] + h * k[i]; }
        sbr_rhs<KIND>(p, y, kla, ec, ld, k);
#pragma unroll
        for (int i = 0; i < SBR_NX; ++i) x[i] = acc[i] + h6 * k[i];
    }
}

// ---------------------------------------------------------------------------------------------------
// Per-env controller registers that are live ACROSS the RK4 loop - kept small on purpose.  The Kla history and
// the bookkeeping rows (return, steps, status) are not needed until after the integration: the step kernel loads them
// up front with everything else (one exposed round trip) and parks them in LDS, so the hot loop keeps its VGPRs.
struct SbrCtl {
    double t, so_m1, so_m2, sno_m1, sno_m2, ie_do, ie_ec, ec_last, ec_prev, u_do, u_ec;
    double kla_last;          // Kla[-1]: bias of the velocity-form DO-PID
    double knew[2];           // Kla values appended by this call: 1, or 2 on a phase-boundary call (sbr_create checks
                              // that every phase is longer than t_delta, so a third interval cannot fire)
    int n_new;
    int st_new;               // SBR_ST_* bits raised by this call
    double span;              // t_range[-1] - t_range[0] of the last interval
    int rows;                 // len(t_range) of the last interval: 9 or 10
};

// the six components whose change over the (last) interval the observation reports (:1069-1076)
#define SBR_NXD 6
SBR_DEV void sbr_take6(const double (&x)[SBR_NX], double (&x6)[SBR_NXD]) {
    x6[0] = x[2]; x6[1] = x[5]; x6[2] = x[6]; x6[3] = x[8]; x6[4] = x[9]; x6[5] = x[10];
}
// where the start values of an interval are parked while the RK4 loop runs: registers, or this lane's LDS slots
struct SbrX6Reg {
    double v[SBR_NXD];
    SBR_DEV void put(const double (&x)[SBR_NX]) { sbr_take6(x, v); }
    SBR_DEV void get(double (&o)[SBR_NXD]) const {
#pragma unroll
        for (int j = 0; j < SBR_NXD; ++j) o[j] = v[j];
    }
};
#ifndef SBR_BLOCK
#define SBR_BLOCK 256
#endif
struct SbrX6Lds {          // slot j of lane l lives at base[j * SBR_BLOCK + l]: conflict-free, 8-byte accesses
    double* base;
    SBR_DEV void put(const double (&x)[SBR_NX]) {
        base[0 * SBR_BLOCK] = x[2]; base[1 * SBR_BLOCK] = x[5]; base[2 * SBR_BLOCK] = x[6]; base[3 * SBR_BLOCK] = x[8]; base[4 * SBR_BLOCK] = x[9];
        base[5 * SBR_BLOCK] = x[10];
    }
    SBR_DEV void get(double (&o)[SBR_NXD]) const {
#pragma unroll
        for (int j = 0; j < SBR_NXD; ++j) o[j] = base[j * SBR_BLOCK];
    }
};

// Sticky domain-of-validity bits (SBR_ST_* in sbr_amd.h), evaluated on the end state of an interval.  x < -K/2 is
// "within 50 % of the pole of x/(K+x)".  Pure bookkeeping: nothing in the dynamics reads it.
SBR_DEV int sbr_status_bits(const SbrPar& p, const double (&x)[SBR_NX]) {
    int st = 0;
    const double lo = -1e-6;
    if (x[2] < lo || x[4] < lo || x[5] < lo || x[8] < lo || x[9] < lo || x[10] < lo) st |= SBR_ST_NEGATIVE;
    const double ko = p.Koh < p.Koa ? p.Koh : p.Koa;
    if (x[2] < -0.5 * p.Ks || x[8] < -0.5 * ko || x[9] < -0.5 * p.Kno || x[10] < -0.5 * p.Knh) st |= SBR_ST_NEAR_POLE;
    double sum = 0.0;
#pragma unroll
    for (int i = 0; i < SBR_NX; ++i) sum += x[i];
    if (!(fabs(sum) < 1.7e308)) st |= SBR_ST_NONFINITE;          // NaN or inf anywhere
    return st;
}

// One control interval: Sim_aero_rxn :1877-1963 / Sim_anaero_rxn :1965-2051, run_*_step :1331-1419.
// xs6 receives the interval's start values of the xdot components.
template <typename X6>
SBR_DEV void sbr_interval(const SbrPar& p, SbrCtl& c, double (&x)[SBR_NX], X6& xs6, bool aerobic) {
    const double t0 = c.t, t1 = t0 + p.t_delta;
    const double span = t1 - t0;
    c.rows = (int)(span / p.dt);              // 9 or 10: fp rounding of (t+t_delta)-t   (:1339, :1384)
    c.rows = c.rows < 2 ? 2 : (c.rows > SBR_KLA_HIST ? SBR_KLA_HIST : c.rows);   // bounded even if t was injected as garbage
    // DO-PID -> Kla (velocity form: bias is the previous Kla).  In anoxic intervals the output is forced to 0
    // but the integral keeps winding with set-point 0 (:1974-1997).
    const double e = (aerobic ? c.u_do : 0.0) - c.so_m1;
    const double dcv = (c.so_m1 - c.so_m2) / p.dt;
    c.ie_do = c.ie_do + e * p.dt;
    double kla = aerobic ? (p.Kc_DO * e + p.KcI_DO * c.ie_do + p.KcD_DO * dcv + c.kla_last) : 0.0;
    if (kla > p.Kla_max) { kla = p.Kla_max; c.ie_do = c.ie_do - e * p.dt; }
    if (kla < p.Kla_min) { kla = p.Kla_min; c.ie_do = c.ie_do - e * p.dt; }
    // NO3-PID -> EC (error sign reversed, :2006); forced to 0 in aerobic intervals while its integral winds (:1918-1937)
    const double e2 = c.sno_m1 - c.u_ec;
    const double dcv2 = (c.sno_m1 - c.sno_m2) / p.dt;
    c.ie_ec = c.ie_ec + e2 * p.dt;
    double ec = aerobic ? 0.0 : (p.Kc_EC * e2 + p.KcI_EC * c.ie_ec + p.KcD_EC * dcv2 + c.ec_last);
    if (ec < p.EC_min) { ec = p.EC_min; c.ie_ec = c.ie_ec - e2 * p.dt; }
    else if (ec > p.EC_max) { ec = p.EC_max; c.ie_ec = c.ie_ec - e2 * p.dt; }
    xs6.put(x);
    // wave-uniform choice: if no lane doses, V/Si/Xi are constants of the interval
    double nold[SBR_NX];
#pragma unroll
    for (int i = 0; i < SBR_NX; ++i) nold[i] = 0.0;
    if (__builtin_amdgcn_ballot_w64(ec != 0.0) == 0ull) sbr_rk4<3>(p, x, span, p.substeps, kla, 0.0, nold);
    else sbr_rk4<0>(p, x, span, p.substeps, kla, ec, nold);
    if (c.n_new == 0) c.knew[0] = kla; else c.knew[1] = kla;      // n_new <= 2, see SbrCtl
    c.n_new += 1;
    c.kla_last = kla;
    c.ec_prev = c.ec_last; c.ec_last = ec;
    c.so_m2 = c.so_m1; c.so_m1 = x[8];
    c.sno_m2 = c.sno_m1; c.sno_m1 = x[9];
    c.t = t1; c.span = span;
    c.st_new |= sbr_status_bits(p, x);
}

// The phase logic of SbrOS.step (:860-1010): four sequential tests on the running time (:860, :896, :931, :963); a call
// that crosses a phase boundary runs a second interval.  Written as a loop over the four tests so that the interval
// code exists ONCE in the kernel.  Envs reset together are in lockstep, so the branch is wave-uniform in practice;
// divergent waves are still correct.
template <typename X6>
SBR_DEV void sbr_run_intervals(const SbrPar& p, SbrCtl& c, double (&x)[SBR_NX], double a0, double a1, X6& xs6) {
    a0 = a0 < 0.0 ? 0.0 : (a0 > p.act_DO_max ? p.act_DO_max : a0);       // :901-906
    a1 = a1 < 0.0 ? 0.0 : (a1 > p.act_EC_max ? p.act_EC_max : a1);       // :865-870
    c.n_new = 0; c.st_new = 0;
#pragma unroll 1
    for (int pass = 0; pass < 4; ++pass) {
        const double t = c.t;
        const bool aerobic = (pass & 1) != 0;
        const bool fire = pass == 0 ? (t < p.T3_0)
                        : pass == 1 ? (t >= p.T3_0 && t <= p.T3_end)
                        : pass == 2 ? (t > p.T3_end && t <= p.T4_end)
                                    : (t > p.T4_end);
        if (fire && c.n_new < 2) {
            if (aerobic) { c.u_do = a0; c.u_ec = 0.0; } else { c.u_ec = a1; c.u_do = 0.0; }
            sbr_interval(p, c, x, xs6, aerobic);
        }
    }
}

// Kla list bookkeeping: hist is oldest-first, hist[9] = Kla[-1].
SBR_DEV void sbr_hist_push(double (&hist)[SBR_KLA_HIST], double k) {
#pragma unroll
    for (int j = 0; j < SBR_KLA_HIST - 1; ++j) hist[j] = hist[j + 1];
    hist[SBR_KLA_HIST - 1] = k;
}
SBR_DEV void sbr_hist_apply(const SbrCtl& c, double (&hist)[SBR_KLA_HIST]) {
    if (c.n_new > 0) sbr_hist_push(hist, c.knew[0]);
    if (c.n_new > 1) sbr_hist_push(hist, c.knew[1]);
}

// module_reward_EQIOCI.py:4-115.  Kla got one append per interval, EC got rows-1:  Kla[-rows:-1] is
// the rows-1 values BEFORE the current one, EC[-rows:-1] = last value of the previous interval +
// (rows-2) x current.
SBR_DEV double sbr_reward(const SbrPar& p, const SbrCtl& c, const double (&hist)[SBR_KLA_HIST], const double (&x)[SBR_NX]) {
    const double xi = x[3], xs = x[4], xbh = x[5], xba = x[6], xp = x[7];
    const double snkj = x[10] + x[11] + x[12] + 0.08 * (xbh + xba) + 0.06 * (xp + xi);
    const double ss_ = 0.75 * (xs + xi + xbh + xba + xp);
    const double bod5 = 0.25 * (x[2] + xs + (1 - 0.08) * (xbh + xba));
    const double cod = x[2] + x[1] + xs + xi + xbh + xba + xp;
    const double eqi = (2 * ss_ + cod + 30 * snkj + 10 * x[9] + 2 * bod5) * (1.0 / 1000) * 0.66;
    const double eqi2 = eqi / 10;
    const double td = 0.002 / 24;
    double ksum = (c.rows >= 10) ? hist[0] : 0.0;          // 9 previous values if rows == 10, else 8
#pragma unroll
    for (int j = 1; j < SBR_KLA_HIST - 1; ++j) ksum = ksum + hist[j];
    const double ae = 8 / (c.span * 1.8 * 1000) * (1.32 * ksum * td);
    double esum = c.ec_prev;
    for (int j = 0; j < c.rows - 2; ++j) esum = esum + c.ec_last;
    const double ec_oci = p.EC_conc * esum * td / (c.span * 1000);
    const double oci = ae + ec_oci;
    return (1 - (eqi2 * eqi2 + oci * oci)) / 473;
}

// module_reward_continuous_G2ANET.py:4-45 (cfg.reward_kind = 1): piecewise-linear in Ss, So, Sno, Snh of the end state
SBR_DEV double sbr_reward_g2anet(const double (&x)[SBR_NX]) {
    const double ss = x[2], so = x[8], sno = x[9], snh = x[10];
    const double r_ec = ss < 0 ? 1.0 : -(ss - 0) / (10 - 0) + 1;
    const double r_e = so < 1.5 ? 0.0 : -(1 / (8 - 1.5)) * (so - 8) + 0;
    const double r_sno = sno < 4 ? 1.0 : -(sno - 4) / (20 - 4) + 1;
    const double r_snh = snh < 4 ? 1.0 : -(snh - 4) / (20 - 4) + 1;
    return (1 * r_ec + 1.5 * r_e + 2 * r_sno + 2 * r_snh) / 10;
}

// module_reward_continuous.py:4-65 (cfg.reward_kind = 2, the reward of SbrEnv3/SbrEnv4): operating cost only.
// batch_type 1 = reaction interval: aeration energy of Kla[-1]; 2 = end of cycle: sum(Kla) of the whole list, pumping of
// the wastage and the effluent, -246 when the effluent ammonia is not below 4 g/m3.  (Branch 0, a fill interval, adds
// 0.004*Qin of pumping; the SBROS-v1 plant fills inside reset(), so no call pays it.)
SBR_DEV double sbr_reward_oci(const SbrPar& p, int batch_type, double kla_last, double kla_sum, double qw, double snh_eff) {
    const double td = 0.002 / 24;
    double pe = 0.0, ae_dt = 1.32 * kla_last * td, r_snh = 0.0;
    if (batch_type == 2) {
        pe = (0.05 * qw + 0.004 * p.Qeff);
        ae_dt = 1.32 * kla_sum * td;
        r_snh = snh_eff < 4 ? 0.0 : -246.0;
    }
    const double ae = p.So_sat / (1.8 * 1000) * ae_dt;
    const double oci = ae + pe;
    return (0.5 - oci) + r_snh;
}

SBR_DEV double sbr_clip1(double v) { return v > 1.0 ? 1.0 : (v < -1.0 ? -1.0 : v); }

// obs_DO ++ obs_EC :1027-1114 (normalisers :150-156, xdot scales :1069-1076) and state = [t, x] / x_1_state
// (:153, :1020-1025), 33 values, written to `o` with stride `st` (the step kernel passes its LDS staging tile).
// xr: what is reported; xa6 -> xr: span of the clipped state change.  Divisions by the constant normalisers are
// multiplications by their reciprocals (<= 1 ulp in fp64, invisible after the float32 cast).
template <typename OutT>
SBR_DEV void sbr_write_obs(OutT* o, int st, double t_obs, const double (&xr)[SBR_NX], const double (&xa6)[SBR_NXD],
                           const double (&xb)[SBR_NX]) {
    const double tt = t_obs * 2.0;
    const double dss = sbr_clip1((xb[2] - xa6[0]) * (1.0 / 50)), dxh = sbr_clip1((xb[5] - xa6[1]) * (1.0 / 4000));
    const double dxa = sbr_clip1((xb[6] - xa6[2]) * (1.0 / 500)), dso = sbr_clip1((xb[8] - xa6[3]) * (1.0 / 8));
    const double dno = sbr_clip1((xb[9] - xa6[4]) * (1.0 / 50)), dnh = sbr_clip1((xb[10] - xa6[5]) * (1.0 / 50));
    o[0 * st] = (OutT)tt; o[1 * st] = (OutT)(xr[5] * (1.0 / 2000)); o[2 * st] = (OutT)(xr[6] * (1.0 / 500));
    o[3 * st] = (OutT)(xr[8] * (1.0 / 8)); o[4 * st] = (OutT)(xr[10] * (1.0 / 10));
    o[5 * st] = (OutT)dxh; o[6 * st] = (OutT)dxa; o[7 * st] = (OutT)dso; o[8 * st] = (OutT)dnh;
    o[9 * st] = (OutT)tt; o[10 * st] = (OutT)(xr[2] * (1.0 / 30)); o[11 * st] = (OutT)(xr[5] * (1.0 / 2000));
    o[12 * st] = (OutT)(xr[9] * (1.0 / 10)); o[13 * st] = (OutT)(xr[10] * (1.0 / 10));
    o[14 * st] = (OutT)dss; o[15 * st] = (OutT)dxh; o[16 * st] = (OutT)dno; o[17 * st] = (OutT)dnh;
}

template <typename OutT>
SBR_DEV void sbr_write_state(OutT* s, int st, double t_obs, const double (&x)[SBR_NX]) {
    s[0 * st] = (OutT)(t_obs * 2.0); s[1 * st] = (OutT)(x[0] * (1.0 / 1.32)); s[2 * st] = (OutT)(x[1] * (1.0 / 30));
    s[3 * st] = (OutT)(x[2] * (1.0 / 30)); s[4 * st] = (OutT)(x[3] * (1.0 / 1500)); s[5 * st] = (OutT)(x[4] * (1.0 / 150));
    s[6 * st] = (OutT)(x[5] * (1.0 / 3000)); s[7 * st] = (OutT)(x[6] * (1.0 / 2000)); s[8 * st] = (OutT)(x[7] * (1.0 / 600));
    s[9 * st] = (OutT)(x[8] * (1.0 / 8)); s[10 * st] = (OutT)(x[9] * (1.0 / 20)); s[11 * st] = (OutT)(x[10] * (1.0 / 20));
    s[12 * st] = (OutT)(x[11] * (1.0 / 10)); s[13 * st] = (OutT)(x[12] * (1.0 / 10)); s[14 * st] = (OutT)(x[13] * (1.0 / 10));
}

// ---------------------------------------------------------------------------------------------------
// Terminal phases of the last call of an episode: settle (:2171-2262; v == vmax always, so the layer
// system is linear and has the closed form below), draw + wastage (:2327-2393), idle (:2554-2597).
// settle: layer concentrations after t_set (closed form, see above); returns Xf
SBR_DEV double sbr_settle(const SbrPar& p, const double (&x)[SBR_NX], double t_set, double (&sx)[10]) {
    const double xf = 0.75 * (x[3] + x[4] + x[5] + x[6] + x[7]);
    const double z = x[0] / p.settler_area;
    const double a = p.settler_vmax / z * t_set, ea = exp(-a);
    // sX[9-j] = Xf e^-a sum_{m<=j} a^m/m!, sX[0] = 10 Xf - sum(others)
    double term = 1.0, partial = 0.0, others = 0.0;
#pragma unroll
    for (int j = 0; j < 9; ++j) { partial += term; sx[9 - j] = xf * ea * partial; term *= a / (double)(j + 1); }
#pragma unroll
    for (int j = 1; j < 10; ++j) others += sx[j];
    sx[0] = 10.0 * xf - others;
    return xf;
}

// draw + wastage (:2327-2393 / sub_phases_FB.py:780-855): x becomes the reactor after the draw; returns Qw and, in
// sx_eff, the sludge that left with the effluent (sum(sX[-m:-1]*layer_volume), which drops the last layer)
SBR_DEV double sbr_draw(const SbrPar& p, double (&x)[SBR_NX], const double (&sx)[10], double xf, double& sx_eff) {
    const double vs = x[0];
    const double layer_v = vs / 10;
    double resid_v = vs - p.Qeff;
    int m = (int)ceil(rint(p.Qeff / layer_v));        // python round() is half-to-even = rint
    m = m < 1 ? 1 : (m > 9 ? 9 : m);
    const int nl = 10 - m;                            // layers that stay
    double wsum = 0.0, se = 0.0;
#pragma unroll
    for (int i = 0; i < 9; ++i) { if (i < nl) wsum = wsum + layer_v * sx[i]; else se = se + sx[i] * layer_v; }
    sx_eff = se;
    double waste = wsum - p.biomass_setpoint * resid_v;
    double qw = __builtin_nan("");
    bool open = true;
    double wkeep = 0.0;                               // sum of the weights that remain, in layer order
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        if (i < nl) {
            double w = layer_v * sx[i];
            if (open) {
                const double rest = waste - w;
                if (rest > 0) { waste = rest; w = 0.0; resid_v -= layer_v; }
                else { qw = waste / (sx[i] - p.biomass_setpoint); w = w - qw * sx[i]; resid_v -= qw; open = false; }
            }
            wkeep = wkeep + w;
        }
    }
    const double sx2 = wkeep / resid_v;
    x[0] = resid_v;
#pragma unroll
    for (int i = 3; i <= 7; ++i) x[i] = x[i] * (1 / 0.75) * sx2 / xf;
    return qw;
}

SBR_DEV double sbr_terminal(const SbrPar& p, SbrCtl& c, double (&hist)[SBR_KLA_HIST], double (&x)[SBR_NX]) {
    const double t_set = p.t_settle * p.t_cycle;
    double sx[10], sx_eff;
    const double xf = sbr_settle(p, x, t_set, sx);
    const double t_after_draw = (c.t + t_set) + p.t_draw * p.t_cycle;
    const double qw = sbr_draw(p, x, sx, xf, sx_eff);
    // idle: one DO-PID update (So[-1] == So[-2] == x[8] after settle/draw => dcv = 0), then conversion only
    const double e = c.u_do - x[8];
    c.ie_do = c.ie_do + e * p.dt;
    double kla = p.Kc_DO * e + p.KcI_DO * c.ie_do + c.kla_last;
    if (kla > p.Kla_max) { kla = p.Kla_max; c.ie_do = c.ie_do - e * p.dt; }
    if (kla < p.Kla_min) { kla = p.Kla_min; c.ie_do = c.ie_do - e * p.dt; }
    const double span = p.t_cycle - t_after_draw;
    int n = (int)(span / p.dt);                       // 464 for the reference's schedule
    n = n < 0 ? 0 : (n > 100000 ? 100000 : n);        // every wave terminates whatever t holds
    double nold[SBR_NX];
#pragma unroll
    for (int i = 0; i < SBR_NX; ++i) nold[i] = 0.0;
    sbr_rk4<2>(p, x, span, n, kla, 0.0, nold);
    sbr_hist_push(hist, kla);                         // Kla.append in Sim_idle (:2578)
    c.kla_last = kla;
    return qw;
}

// What is left of SbrOS.step (:1011-1273) after the intervals: reward, done test, terminal phases.  Returns the reward;
// xa6 is updated to the pre-settle values when the terminal phases run; qw is written only then.
// OCI = the operating-cost reward (cfg.reward_kind 2) with its running sum(Kla) of the episode's list, ksum: a
// compile-time variant, so that the default kernels carry none of it.
template <bool OCI>
SBR_DEV double sbr_finish_step(const SbrPar& p, SbrCtl& c, double (&hist)[SBR_KLA_HIST], double (&x)[SBR_NX],
                               double (&xa6)[SBR_NXD], double& t_obs, bool& dn, double& qw, double& ksum) {
    sbr_hist_apply(c, hist);
    double r;
    if (OCI) {
        ksum = ksum + c.knew[0];
        if (c.n_new > 1) ksum = ksum + c.knew[1];
        r = sbr_reward_oci(p, 1, c.kla_last, 0.0, 0.0, 0.0);
    } else {
        r = p.reward_kind == 1 ? sbr_reward_g2anet(x) : sbr_reward(p, c, hist, x);     // wave-uniform choice
    }
    t_obs = c.t;
    dn = false;
    if (c.t >= p.T5_end) {                                               // :1122
        dn = true;
        if (p.terminal) {
            sbr_take6(x, xa6);
            const double snh_eff = x[10];            // solubles pass the settler unchanged: eff_component[3] (:2642)
            qw = sbr_terminal(p, c, hist, x);
            if (OCI) {
                ksum = ksum + c.kla_last;            // Sim_idle's Kla.append (:2578)
                r = sbr_reward_oci(p, 2, c.kla_last, ksum, qw, snh_eff);
            }
            t_obs = p.t_cycle;
        }
    }
    return r;
}

// ---------------------------------------------------------------------------------------------------
// Per-cycle env SBR-v2: one PID-controlled phase (sub_phases_FB.py filling.sim_rxn :178-271 / rxn.sim_rxn :406-500).
// Control intervals are the cells of linspace(t_start, t_end, n2), n2 = int((t_end-t_start)/(10 t_delta)); positional
// PID on So with derivative action; interval 0 OVERWRITES Kla[0], which held the incoming bias, so intervals 1.. use the
// controlled (clamped) Kla of interval 0 as bias (:219,:243); the integral restarts in every phase.  t_start/t_end
// are wave-uniform.  Returns the last Kla; ksum = sum(Kla), n_iv = number of intervals.
template <bool FILL>
SBR_DEV double sbr_cycle_phase(const SbrPar& p, double (&x)[SBR_NX], double t_start, double t_end, double sp, double kla_in,
                               const double (&ld)[SBR_NX], double& ksum, int& n_iv) {
    const double t_delta = 0.002 / 24;                           // gym_SBR_env2.py:34
    int n2 = (int)((t_end - t_start) / (t_delta * 10));
    n2 = n2 < 2 ? 2 : (n2 > 100000 ? 100000 : n2);               // every wave terminates
    n_iv = n2 - 1;
    const double step = (t_end - t_start) / (double)(n2 - 1);    // numpy.linspace: i*step + start, last = stop
    double so = x[8], so_prev = x[8], ie = 0.0, bias = kla_in, k = kla_in, sum = 0.0;
    for (int i = 0; i < n_iv; ++i) {
        const double g0 = (double)i * step + t_start;
        const double g1 = (i + 1 == n2 - 1) ? t_end : (double)(i + 1) * step + t_start;
        const double e = sp - so;
        double dcv = 0.0;
        if (i >= 1) { dcv = (so - so_prev) / p.cyc_dt; ie = ie + e * p.cyc_dt; }
        k = p.cyc_Kc * e + p.cyc_KcI * ie + p.cyc_KcD * dcv + bias;
        if (k > p.Kla_max) { k = p.Kla_max; ie = ie - e * p.cyc_dt; }
        if (k < p.Kla_min) { k = p.Kla_min; ie = ie - e * p.cyc_dt; }
        if (i == 0) bias = k;
        sbr_rk4<FILL ? 1 : 2>(p, x, g1 - g0, p.substeps, k, 0.0, ld);
        sum = sum + k;
        so_prev = so; so = x[8];
    }
    ksum = sum;
    return k;
}

// SbrEnv2.step (gym_SBR_env2.py:131-171) = SBR_model_FB.run (SBR_model_FB.py:8-295) + module_reward.py:4-51 for one env:
// x is the start state in, the end-of-cycle state out; ld the influent with ld[0] = Qin/t_phs1; a[3] the clipped action.
// obs3 = [Qeff, COD_eff, Snh_eff/30]; diag (SBR_NCYC_DIAG doubles) may be nullptr.
SBR_DEV double sbr_cycle_env(const SbrPar& p, double (&x)[SBR_NX], const double (&ld)[SBR_NX], double a0, double a1, double a2,
                             double (&obs3)[3], double* diag, int diag_stride) {
    const double t_delta = 0.002 / 24;
    a0 = a0 < 0.0 ? 0.0 : (a0 > 1.0 ? 1.0 : a0); a1 = a1 < 0.0 ? 0.0 : (a1 > 1.0 ? 1.0 : a1);
    a2 = a2 < 0.0 ? 0.0 : (a2 > 1.0 ? 1.0 : a2);
    const double sp3 = a0 * 8, sp5 = a1 * 8, sp8 = a2 * 8;      // DO_setpoints[2], [4], [7]  (:184-186)
    const double qin = p.WV - x[0];
    double ks1, ks2, ks3, ks4, ks5, ks8;
    int n1, n2, n3, n4, n5, n8;
    double t_start = 0.0, t_end = 0.0 + p.t_ph[0];
    double kl = sbr_cycle_phase<true>(p, x, t_start, t_end, 0.0, 0.0, ld, ks1, n1);
    t_start = t_end + t_delta; t_end = t_start + p.t_ph[1];
    kl = sbr_cycle_phase<false>(p, x, t_start, t_end, 0.0, kl, ld, ks2, n2);
    t_start = t_end + t_delta; t_end = t_start + p.t_ph[2];
    kl = sbr_cycle_phase<false>(p, x, t_start, t_end, sp3, kl, ld, ks3, n3);
    t_start = t_end + t_delta; t_end = t_start + p.t_ph[3];
    kl = sbr_cycle_phase<false>(p, x, t_start, t_end, 0.0, kl, ld, ks4, n4);
    t_start = t_end + t_delta; t_end = t_start + p.t_ph[4];
    kl = sbr_cycle_phase<false>(p, x, t_start, t_end, sp5, kl, ld, ks5, n5);
    // settle
    t_start = t_end + t_delta; t_end = t_start + p.t_ph[5];
    double sx[10], sx_eff;
    const double xf = sbr_settle(p, x, t_end - t_start, sx);
    // draw; the effluent composition (cal_eq, sub_phases_FB.py:860-915) uses the PRE-draw state with its particulates
    // scaled by the sludge carried out
    t_start = t_end + t_delta; t_end = t_start + p.t_ph[6];
    const double xi = x[3], xs = x[4], xbh = x[5], xba = x[6], xp = x[7];
    const double qw = sbr_draw(p, x, sx, xf, sx_eff);
    const double exi = xi * (1 / 0.75) * sx_eff / xf, exs = xs * (1 / 0.75) * sx_eff / xf, exbh = xbh * (1 / 0.75) * sx_eff / xf;
    const double exba = xba * (1 / 0.75) * sx_eff / xf, exp_ = xp * (1 / 0.75) * sx_eff / xf;
    const double snkj = x[10] + x[11] + x[12] + 0.08 * (exbh + exba) + 0.06 * (exp_ + exi);
    const double ntot = x[9] + snkj;
    const double ss_ = 0.75 * (exs + exi + exbh + exba + exp_);
    const double bod5 = 0.25 * (x[2] + exs + (1 - 0.08) * (exbh + exba));
    const double cod = x[2] + x[1] + exs + exi + exbh + exba + exp_;
    const double eqi = (2 * ss_ + cod + 30 * snkj + 10 * x[9] + 2 * bod5) * (1.0 / 1000) * 0.66;
    const double snh_eff = x[10], sno_eff = x[9];
    // aerated idle from the drawn reactor, bias = last Kla of phase 5 (SBR_model_FB.py:258)
    t_start = t_end + t_delta; t_end = t_start + p.t_ph[7];
    sbr_cycle_phase<false>(p, x, t_start, t_end, sp8, kl, ld, ks8, n8);
    // reward
    const double td = 0.002 / 24;
    const double ae3 = 1.32 * ks3 * td / ((double)n3 * td), ae5 = 1.32 * ks5 * td / ((double)n5 * td);
    const double ae8 = (1.32 - qw) * ks8 * td / ((double)n8 * td);
    const double ae = p.So_sat / (1.8 * 1000) * (ae3 + ae5 + ae8);
    const double pe = (0.004 * qin + 0.05 * qw + 0.004 * p.Qeff);
    const double me = 0.005 * 1.32 * 24 + 0.005 * 1.32 * 24;
    const double oci = ae + pe + me;
    obs3[0] = p.Qeff; obs3[1] = cod; obs3[2] = snh_eff / 30;
    if (diag) {
        const int st = diag_stride;
        diag[0 * st] = qw; diag[1 * st] = eqi; diag[2 * st] = oci; diag[3 * st] = ntot; diag[4 * st] = cod; diag[5 * st] = snh_eff;
        diag[6 * st] = bod5; diag[7 * st] = sno_eff; diag[8 * st] = ks3 / (double)n3; diag[9 * st] = ks5 / (double)n5;
        diag[10 * st] = ks8 / (double)n8; diag[11 * st] = xf;
    }
    return (5 - oci) + (snh_eff < 4 ? 0.0 : -20.0);
}

// ---------------------------------------------------------------------------------------------------
// Philox4x32-10 (Salmon et al. 2011).  counter = (i, stream, env_lo, env_hi), key = seed.
SBR_DEV void sbr_philox(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c[0];
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * c[2];
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0, n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1, n3 = (uint32_t)p0;
        c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
}

SBR_DEV double sbr_u53(uint32_t hi, uint32_t lo) {     // uniform in (0, 1]
    const uint64_t v = (((uint64_t)hi << 32) | lo) >> 11;
    return ((double)v + 1.0) * (1.0 / 9007199254740992.0);
}

// two standard normals: Box-Muller on Philox block i of stream 0
SBR_DEV void sbr_normal_pair(uint64_t seed, uint64_t env_id, uint32_t i, double& z0, double& z1) {
    uint32_t c[4] = {i, 0u, (uint32_t)env_id, (uint32_t)(env_id >> 32)};
    sbr_philox(c, (uint32_t)seed, (uint32_t)(seed >> 32));
    const double u1 = sbr_u53(c[0], c[1]), u2 = sbr_u53(c[2], c[3]);
    const double rad = sqrt(-2.0 * log(u1)), ang = 6.283185307179586476925286766559 * u2;
    double sn, cs;
    sincos(ang, &sn, &cs);
    z0 = rad * cs; z1 = rad * sn;
}

// uniform random action of call `call` (stream 1), float32 like the action tensors of sbr_step
SBR_DEV void sbr_policy_action(const SbrPar& p, uint64_t seed, uint64_t env_id, uint32_t call, float& a0, float& a1) {
    uint32_t c[4] = {call, 1u, (uint32_t)env_id, (uint32_t)(env_id >> 32)};
    sbr_philox(c, (uint32_t)seed, (uint32_t)(seed >> 32));
    a0 = (float)(sbr_u53(c[0], c[1]) * p.act_DO_max);
    a1 = (float)(sbr_u53(c[2], c[3]) * p.act_EC_max);
}
