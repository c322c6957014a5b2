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
    // kinetics folded into the Monod denominators (sbr_rates): d1 = f1a Ss + f1b = (Ks + Ss) kh/muH,
    // d2 = f2a So + f2b = (Koh + So)/kh, d4 = f4a Snh + f4b = (Knh + Snh)/muA
    double f1a, f1b, f2a, f2b, f3a, f3b, f4a, f4b;   // d3 = f3a Sno + f3b = (Kno + Sno)/(eta_g Koh)
    double KohEtag;      // Koh * eta_g
    double etah_g;       // eta_h / eta_g
    // rho4 + rho5 is carried as s45/bH = Xbh + (bA/bH) Xba, Sno's derivative as k/nu9_3: the factors sit in the coefficients
    double bA_bH, n4_45b, n12_45b, n7_45b;    // bA/bH, nu4_45 bH, nu12_45 bH, nu7_45 bH
    double n9_23, inv_n9_3;   // nu9_2 / nu9_3, 1 / nu9_3
    double t_ph[8];      // phase lengths t_cycle * t_ratio[k]                 (SBR_model_FB.py:18-27)
    double cyc_Kc, cyc_KcI, cyc_KcD, cyc_dt;   // positional PID of the per-cycle env (sub_phases_FB.py:205-243)
    // reciprocals of wave-uniform divisors, taken once on the host (an IEEE f64 division is ~19 issue slots on the device)
    double inv_dt, inv_t_delta, inv_substeps, inv_cyc_dt, h_fill;
    double inv_Koh, inv_Koa;
    double inv_ae_max, inv_ec_max;   // 1 / AE_OCI_max, 1 / EC_OCI_max of module_reward_EQIOCI.py:72, :80 (trajectory export)
    // len(t_range) = int(span/dt) of a control interval (:1339, :1384) is 10 iff span >= rows10_min and 9 iff
    // rows9_min <= span < rows10_min: the smallest doubles whose IEEE quotient by dt reaches 10.0 / 9.0 (found on the
    // host by stepping through neighbouring doubles), so the common case costs two comparisons and stays exact
    double rows10_min, rows9_min;
    // per-cycle env: the schedule of its six PID-controlled phases (fill, 2..5, idle) is wave-uniform, so everything the
    // reference derives from it with a division - n2 = int((t_end - t_start)/(10 t_delta)), linspace's step, the reward's
    // 1/(n td) - is taken once on the host with the reference's own IEEE operations (SBR_model_FB.py:18-27, sub_phases_FB.py:183-184)
    double cyc_t0[6], cyc_t1[6], cyc_step[6], cyc_inv_ntd[6], cyc_inv_n[6];
    double cyc_tset;         // length of the settling phase as the reference forms it (t_end - t_start)
    double inv_t_ph0;        // 1 / t_ph[0]
    double sosat_k;          // So_sat / (1.8 * 1000)
    double inv_T_fill;
    int32_t cyc_n2[6];
    int32_t substeps, terminal, fill_rows, reward_kind, random_scenario;
    int32_t scheme;          // 0: RK4 x substeps per control interval; 1: adaptive Butcher-5 (sbr_b5a)
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
// the same Newton step from a reciprocal that is already within ~1e-7 of 1/d (1/V while carbon is dosed: V moves by
// 3e-8 of itself per substep, so e^3 ~ 1e-23)
SBR_DEV double sbr_rcp_refine(double d, double r) {
    const double e = __builtin_fma(-d, r, 1.0);
    return __builtin_fma(r, __builtin_fma(e, e, e), r);
}

// The components that feed back into the process rates, and therefore have to exist at every RK4 stage:
// Ss Xs Xbh Xba So Sno Snh Snd Xnd.  V, Si, Xi and Salk never enter a rate; Xp does not either.
#define SBR_NA 9
enum { A_SS = 0, A_XS, A_XBH, A_XBA, A_SO, A_SNO, A_SNH, A_SND, A_XND };
SBR_DEV void sbr_gather(const double (&x)[SBR_NX], double (&a)[SBR_NA]) {
    a[A_SS] = x[2]; a[A_XS] = x[4]; a[A_XBH] = x[5]; a[A_XBA] = x[6]; a[A_SO] = x[8]; a[A_SNO] = x[9]; a[A_SNH] = x[10];
    a[A_SND] = x[11]; a[A_XND] = x[12];
}
SBR_DEV void sbr_scatter(const double (&a)[SBR_NA], double (&x)[SBR_NX]) {
    x[2] = a[A_SS]; x[4] = a[A_XS]; x[5] = a[A_XBH]; x[6] = a[A_XBA]; x[8] = a[A_SO]; x[9] = a[A_SNO]; x[10] = a[A_SNH];
    x[11] = a[A_SND]; x[12] = a[A_XND];
}

// What the other derivatives are made of (the known-answer kernel k_rhs rebuilds all 14 from it; the integrator only
// needs s45): rho1, rho2, rho3, rho6 and rho4 + rho5.
struct SbrRho { double rho1, rho2, rho3, rho6, s45b; };        // s45b = (rho4 + rho5)/bH

// Conversion rates of the nine feedback components, incl. aeration.  Process rates :1660-1685, combination :1731-1755.
//
// fp64 VALU issue is the bound of every stepping kernel (one wave per SIMD issues a v_fma_f64 every 5.2 cycles, a
// v_mul/v_add_f64 every 4.3, v_rcp_f64 costs 16: scripts/probes/fp64_issue.hip), so this function is written for
// INSTRUCTION COUNT, 49 + one v_rcp_f64 (the first version: 92 + 7):
//  * the reference's ten quotients need six denominators; everything comes from ONE v_rcp_f64 of their product, from
//    which exactly the four quotients the rates use are formed - 1/(d1 d2), 1/(d4 d5), 1/d3, 1/(d2 d6) -: 5 + 7
//    multiplications, dependency depth 3 + 3;
//  * the leading constants of the rates (muH, muA, kh, eta_g Koh) are folded into the denominators on the host - a
//    denominator (K + x)*c costs one FMA instead of one ADD - so that rho1, rho2, rho3, rho7 come out of the
//    reciprocals already scaled: d1 = (Ks+Ss) kh/muH, d2 = (Koh+So)/kh, d3 = (Kno+Sno)/(eta_g Koh), d4 = (Knh+Snh)/muA;
//  * So/(Koh+So) and Koh/(Koh+So) share 1/d2, which is factored out of the hydrolysis bracket together with 1/d6;
//    (Xs/Xbh)/(Kx + Xs/Xbh) is Xs/(Kx Xbh + Xs), and rho8 = (Xnd/Xs) rho7 needs no 1/Xs because rho7 carries the factor
//    Xs (also the continuous extension at Xs -> 0, where the reference evaluates 0/0);
//  * rho4, rho5, rho6 are never formed: they enter the derivatives through FMAs, decay as (rho4 + rho5)/bH with bH in the
//    coefficients; Sno's derivative is returned divided by nu9_3 (the caller's step constants carry the factor).
// 1/V is deliberately NOT part of the batch (see sbr_rk4): the six reciprocals of a lane are the same whether or not a
// wave-mate doses carbon, so an env's arithmetic does not depend on which envs share its wavefront.
// Parity is to tolerance, not bitwise (RHS known answers within 2e-15 relative).
// Returns k[] = d/dt of the nine components, EXCEPT k[A_SNO] = (d Sno/dt)/nu9_3.
//
// The seven constants in SbrMonod are the ones that are NOT homogeneous of degree one in the concentrations.  ASM1's rates
// are: with a[] = s c (the scaled-mass variables of the dosing integrator, sbr_rk4_dose) and these seven scaled by s = V/V0
// (f3a and ka by 1/s), the function returns s r(c) - the same 49 + 1 instructions.  Every other caller passes the model's
// own constants (sbr_monod: wave-uniform, they stay in SGPRs).
struct SbrMonod { double f1b, f2b, f3a, f4b, Koa, kla_sat, ka; };
SBR_DEV SbrMonod sbr_monod(const SbrPar& p, double kla_sat) { return SbrMonod{p.f1b, p.f2b, p.f3a, p.f4b, p.Koa, kla_sat, p.ka}; }
SBR_DEV void sbr_rates(const SbrPar& p, const SbrMonod& m, const double (&a)[SBR_NA], double kla, double (&k)[SBR_NA],
                       SbrRho& o) {
    const double ss = a[A_SS], xs = a[A_XS], xbh = a[A_XBH], xba = a[A_XBA], so = a[A_SO], sno = a[A_SNO], snh = a[A_SNH],
                 snd = a[A_SND], xnd = a[A_XND];
    const double d1 = __builtin_fma(ss, p.f1a, m.f1b);        // (Ks + Ss) kh/muH
    const double d2 = __builtin_fma(so, p.f2a, m.f2b);        // (Koh + So)/kh
    const double d3 = __builtin_fma(sno, m.f3a, p.f3b);       // (Kno + Sno)/(eta_g Koh)
    const double d4 = __builtin_fma(snh, p.f4a, m.f4b);       // (Knh + Snh)/muA
    const double d5 = m.Koa + so;
    const double d6 = __builtin_fma(p.Kx, xbh, xs);
    // one reciprocal of d1 d2 d3 d6 (d4 d5), then exactly the four quotients the rates use: 5 + 7 multiplications
    const double A = d1 * d2, B = d4 * d5, Cc = d3 * d6, AC = A * Cc;
    const double R = sbr_rcp(AC * B);
    const double rB = R * AC;                                  // muA / ((Knh+Snh)(Koa+So))
    const double rAC = R * B;                                  // 1/(d1 d2 d3 d6)
    const double rA = rAC * Cc;                                // muH / ((Ks+Ss)(Koh+So))
    const double rc = (rAC * A) * d6;                          // eta_g Koh / (Kno+Sno)
    const double rfb = rAC * (d1 * d3);                        // 1/(d2 d6) = kh / ((Kx Xbh + Xs)(Koh+So))
    const double G = (ss * xbh) * rA;
    const double rho1 = G * so;                                // muH Ss/(Ks+Ss) So/(Koh+So) Xbh
    const double kw = sno * rc;                                // eta_g Koh Sno/(Kno+Sno)
    const double rho2 = G * kw;                                // muH Ss/(Ks+Ss) Koh/(Koh+So) Sno/(Kno+Sno) eta_g Xbh
    const double c7 = (rfb * __builtin_fma(p.etah_g, kw, so)) * xbh;   // rho7/Xs = kh [So + eta_h Koh Sno/(Kno+Sno)] Xbh / (d6 (Koh+So))
    const double rho7 = xs * c7, rho8 = xnd * c7;
    const double rho3 = ((snh * so) * rB) * xba;               // muA Snh/(Knh+Snh) So/(Koa+So) Xba
    const double s45b = __builtin_fma(p.bA_bH, xba, xbh);      // (rho4 + rho5)/bH
    const double z = snd * xbh;                                // rho6 / ka
    const double s12 = rho1 + rho2;
    k[A_SS] = __builtin_fma(p.n2_12, s12, rho7);
    k[A_XS] = __builtin_fma(p.n4_45b, s45b, -rho7);
    k[A_XBH] = __builtin_fma(-p.bH, xbh, s12);
    k[A_XBA] = __builtin_fma(-p.bA, xba, rho3);
    k[A_SO] = __builtin_fma(p.n8_1, rho1, __builtin_fma(p.n8_3, rho3, __builtin_fma(-kla, so, m.kla_sat)));   // + kla (So_sat - So)
    k[A_SNO] = __builtin_fma(p.n9_23, rho2, rho3);             // (nu9_2 rho2 + nu9_3 rho3)/nu9_3
    k[A_SNH] = __builtin_fma(p.n10_12, s12, __builtin_fma(p.n10_3, rho3, m.ka * z));
    k[A_SND] = __builtin_fma(-m.ka, z, rho8);
    k[A_XND] = __builtin_fma(p.n12_45b, s45b, -rho8);
    o.rho1 = rho1; o.rho2 = rho2; o.rho3 = rho3; o.rho6 = m.ka * z; o.s45b = s45b;
}

// All 14 derivatives of one state, for the known-answer kernel only.  KIND 0: reaction_dxdt :1658-1787 (dosing ec,
// dilution ec/V), 1: filling_dxdt :1424-1583 at EC = 0 (loading vector ld, ld[0] = inflow), 2: idle_dxdt :2424-2552.
template <int KIND>
SBR_DEV void sbr_rhs(const SbrPar& p, const double (&x)[SBR_NX], double kla, double ec, const double (&ld)[SBR_NX],
                     double (&d)[SBR_NX]) {
    double a[SBR_NA], k[SBR_NA], r[SBR_NX];
    SbrRho o;
    sbr_gather(x, a);
    sbr_rates(p, sbr_monod(p, kla * p.So_sat), a, kla, k, o);
    r[0] = 0.0; r[1] = 0.0; r[3] = 0.0;
    k[A_SNO] = k[A_SNO] * p.n9_3;
    sbr_scatter(k, r);
    r[7] = p.n7_45b * o.s45b;
    r[13] = p.n13_1 * o.rho1 + p.n13_2 * o.rho2 + p.n13_3 * o.rho3 + p.n13_6 * o.rho6;
    if (KIND == 0) {
        const double q = ec * sbr_rcp(x[0]);
        d[0] = ec;
#pragma unroll
        for (int i = 1; i < SBR_NX; ++i) d[i] = r[i] + q * (i == 2 ? (p.EC_conc - x[i]) : (-x[i]));
    } else if (KIND == 1) {
        const double q = ld[0] * sbr_rcp(x[0]);
        d[0] = ld[0];
#pragma unroll
        for (int i = 1; i < SBR_NX; ++i) d[i] = r[i] + q * (ld[i] - x[i]);
    } else {
#pragma unroll
        for (int i = 0; i < SBR_NX; ++i) d[i] = r[i];
    }
}

// Classical RK4, n equal substeps of length h; autonomous inside a call (Kla and the inflow held).
//   FLOW 0: closed reactor - reaction intervals in which this wave doses no carbon, idle phase (idle_dxdt :2424-2552)
//   FLOW 1: carbon dosing - inflow Q = ec of concentration EC_conc in Ss, nothing else (reaction_dxdt :1757-1785):
//           integrated in scaled-mass variables, sbr_rk4_dose below
//   FLOW 2: filling - inflow Q = ld[0] of composition ld[1..13] (filling_dxdt :1555-1581)
// Only the nine feedback components are carried through the stages (x, the running combination and one stage vector:
// 3 x 9 doubles live).  The rest of the state has closed forms that RK4 reproduces to (Q h/V)^5 ~ 1e-38, i.e. exactly:
//   V' = Q                      =>  V advances by h Q per substep; 1/V at the stage times is taken afresh (filling),
//   c' = (Q/V)(c_in - c)        =>  c(t1) = c0 + g (c_in - c0), g = (V1 - V0)/V1, for Si and Xi (no reaction), and for
//   u = Salk - (Snh - Sno)/14       the charge balance u: the alkalinity row of the stoichiometry is (row Snh - row Sno)/14
//                                   for every process (nu13_k = (nu10_k - nu9_k)/14 term by term, :1689-1725), so u only
//                                   dilutes,
//   Xp' = nu7 (rho4 + rho5) + (Q/V)(c_in - Xp): nothing depends on Xp, so without inflow it needs no stage value, only the
//                                   weighted sum of rho4 + rho5 (one FMA per stage).

// Carbon dosing (FLOW 1) in SCALED-MASS variables w_i = c_i V/V0 = c_i s, V0 = the volume at the start of the interval,
// s(t) = V(t)/V0 = 1 + Q t/V0 (round 4; the RK4 layers of the CPU oracle under oracle/ - test infrastructure, never linked
// here - integrate the same system: `rk4_reaction_w` in its C and NumPy files).  In w the dilution terms -(Q/V) c_i of reaction_dxdt (:1757-1785)
// disappear IDENTICALLY:
//     w_i' = s r_i(w/s) + (Q/V0) c_in,i            c_in = EC_conc for Ss, 0 for everything else,
// and because every ASM1 rate is homogeneous of degree one in (concentrations, half-saturation constants) - except
// ammonification rho6 = ka Snd Xbh, which is of degree two - s r(w/s) is sbr_rates evaluated ON w with the additive
// constants of the Monod denominators and the saturation term of the aeration scaled by s, and f3a (it carries 1/Koh: the
// Koh in the numerator of the anoxic switch) and ka scaled by 1/s (SbrMonod).  What the concentration form paid per STAGE -
// nine dilution FMAs, the source term, q = Q/V and its 1/V tracking, the inflow terms of Xp: 68 of 343 instructions per
// substep - becomes 11 instructions per distinct STAGE TIME (two per substep: t + h/2, and t + h which is the next
// substep's t) and two additions for the constant source of Ss (folded into the stage bases: y_j = (a + c_j h src) +
// c_j h k): 299 instructions per substep.
//   * 1/s = 1 - e + e^2 - e^3 with e = s - 1 = Q t/V0: e <= 7e-7 over an interval with the reference's EC_max (sbr_create
//     rejects configurations with EC_max t_delta > 1e-4 IV), so the truncation e^4 is below 1e-16 relative.
//   * w = c at the start of every interval (s = 1), and c = w / s_end at its end: one multiplication per component; Si,
//     Xi and the charge balance u = Salk - (Snh - Sno)/14 have w' = 0, Xp has no source: they need no stage values.
//   * A lane with Q == 0 has e = 0, s = 1/s = 1, every scaled constant equal to the model's own (fma(K, 0, K) = K, K*1 = K)
//     and source 0 (a + 0 = a): it executes, operation for operation, the arithmetic of the FLOW 0 loop on the same
//     numbers - bit for bit the same result whether or not a wave-mate doses (the ballot only selects the code path).
//   * RK4 on w is not RK4 on c: the two discretisations differ by ~(Q h/V) x the local truncation error, 1e-14 relative
//     per substep.  The oracle integrates w as well, so device-vs-oracle stays a rounding-level comparison; both stay
//     pinned to the reference's LSODA trajectories by the same gates as before (tests/test_oracle_golden.py).
struct SbrScaled { SbrMonod m; double rs; };
SBR_DEV SbrScaled sbr_scale_consts(const SbrPar& p, double kla_sat, double e) {
    SbrScaled o;
    const double r1 = __builtin_fma(-e, 1.0, 1.0);             // 1 - e
    const double r2 = __builtin_fma(-e, r1, 1.0);              // 1 - e + e^2
    o.rs = __builtin_fma(-e, r2, 1.0);                         // 1 - e + e^2 - e^3 = 1/(1 + e) - e^4/(1 + e)
    o.m.f1b = __builtin_fma(p.f1b, e, p.f1b); o.m.f2b = __builtin_fma(p.f2b, e, p.f2b); o.m.f4b = __builtin_fma(p.f4b, e, p.f4b);
    o.m.Koa = __builtin_fma(p.Koa, e, p.Koa); o.m.kla_sat = __builtin_fma(kla_sat, e, kla_sat);
    o.m.f3a = p.f3a * o.rs; o.m.ka = p.ka * o.rs;
    return o;
}
SBR_DEV void sbr_rk4_dose(const SbrPar& p, double (&x)[SBR_NX], double h, int n, double kla, double Q) {
    const double h2 = 0.5 * h, h6 = h * (1.0 / 6.0), h3 = h * (1.0 / 3.0);
    const double g1 = h * p.n9_3, g2 = h2 * p.n9_3, g6 = h6 * p.n9_3, g3 = h3 * p.n9_3;      // sbr_rates returns Sno's derivative over nu9_3
    const double p6 = h6 * p.n7_45b, p3 = h3 * p.n7_45b;                                      // Xp' = nu7 bH s45b, no source
    const double kla_sat = kla * p.So_sat;
    const double v0 = x[0];
    const double dl = (h * Q) * sbr_rcp(v0), dl2 = 0.5 * dl;   // growth of s per substep / per half substep
    const double src1 = dl * p.EC_conc, src2 = 0.5 * src1;     // h (Q/V0) EC_conc and half of it: the source of Ss over a (half) substep
    double a[SBR_NA], xp = x[7], e = 0.0;
    sbr_gather(x, a);
    const double n0 = x[10] - x[9];                            // Snh - Sno at the start (charge balance)
    SbrScaled c0 = sbr_scale_consts(p, kla_sat, 0.0);          // the model's own constants, exactly
#pragma unroll 2
    for (int s = 0; s < n; ++s) {
        double k[SBR_NA], y[SBR_NA], acc[SBR_NA], acc7;
        SbrRho o;
        const double ss1 = a[A_SS] + src1, ss2 = a[A_SS] + src2;        // stage bases of Ss with the source folded in
        // stage 1 at (t, s)
        sbr_rates(p, c0.m, a, kla, k, o);
#pragma unroll
        for (int i = 0; i < SBR_NA; ++i) {
            acc[i] = __builtin_fma(i == A_SNO ? g6 : h6, k[i], i == A_SS ? ss1 : a[i]);
            y[i] = __builtin_fma(i == A_SNO ? g2 : h2, k[i], i == A_SS ? ss2 : a[i]);
        }
        acc7 = __builtin_fma(p6, o.s45b, xp);
        // stages 2 and 3 at (t + h/2, s + dl/2)
        const SbrScaled cm = sbr_scale_consts(p, kla_sat, e + dl2);
        sbr_rates(p, cm.m, y, kla, k, o);
        acc7 = __builtin_fma(p3, o.s45b, acc7);
#pragma unroll
        for (int i = 0; i < SBR_NA; ++i) {
            acc[i] = __builtin_fma(i == A_SNO ? g3 : h3, k[i], acc[i]);
            y[i] = __builtin_fma(i == A_SNO ? g2 : h2, k[i], i == A_SS ? ss2 : a[i]);
        }
        sbr_rates(p, cm.m, y, kla, k, o);
        acc7 = __builtin_fma(p3, o.s45b, acc7);
#pragma unroll
        for (int i = 0; i < SBR_NA; ++i) {
            acc[i] = __builtin_fma(i == A_SNO ? g3 : h3, k[i], acc[i]);
            y[i] = __builtin_fma(i == A_SNO ? g1 : h, k[i], i == A_SS ? ss1 : a[i]);
        }
        // stage 4 at (t + h, s + dl): its constants are the next substep's stage-1 constants
        e = e + dl;
        c0 = sbr_scale_consts(p, kla_sat, e);
        sbr_rates(p, c0.m, y, kla, k, o);
#pragma unroll
        for (int i = 0; i < SBR_NA; ++i) a[i] = __builtin_fma(i == A_SNO ? g6 : h6, k[i], acc[i]);
        xp = __builtin_fma(p6, o.s45b, acc7);
    }
    // back to concentrations: c = w / s_end
    const double rs = c0.rs, c14 = 1.0 / 14.0;
#pragma unroll
    for (int i = 0; i < SBR_NA; ++i) a[i] = a[i] * rs;
    sbr_scatter(a, x);
    x[7] = xp * rs;
    x[0] = __builtin_fma(v0, e, v0);                           // V0 s_end
    x[1] = x[1] * rs; x[3] = x[3] * rs;                        // Si, Xi: w' = 0
    const double u = __builtin_fma(-n0, c14, x[13]) * rs;      // charge balance: w' = 0
    x[13] = __builtin_fma(x[10] - x[9], c14, u);
}

template <int FLOW>
SBR_DEV void sbr_rk4(const SbrPar& p, double (&x)[SBR_NX], double h, int n, double kla, double Q,
                     const double (&ld)[SBR_NX]) {
    if constexpr (FLOW == 1) {
        sbr_rk4_dose(p, x, h, n, kla, Q);
        return;
    }
    const double h2 = 0.5 * h, h6 = h * (1.0 / 6.0), h3 = h * (1.0 / 3.0);
    // sbr_rates returns Sno's derivative over nu9_3 and decay over bH: their step constants carry the factors
    const double g1 = h * p.n9_3, g2 = h2 * p.n9_3, g6 = h6 * p.n9_3, g3 = h3 * p.n9_3;
    const double p2 = h2 * p.n7_45b, p1 = h * p.n7_45b, p6 = h6 * p.n7_45b, p3 = h3 * p.n7_45b;     // Xp' = nu7 bH s45b
    const SbrMonod m = sbr_monod(p, kla * p.So_sat);
    double a[SBR_NA], xp = x[7];
    sbr_gather(x, a);
    const double v0 = x[0], n0 = x[10] - x[9];                 // V and Snh - Sno at the start
    double v = v0, rv = FLOW ? sbr_rcp(v0) : 0.0;
    // unrolled by two: the state and the running combination swap registers from one substep to the next, which a rolled
    // loop pays with nine register copies per substep (288 -> 275 instructions per substep, k_step -0.25 us, rollout -3 %)
#pragma unroll 2
    for (int s = 0; s < n; ++s) {
        double k[SBR_NA], y[SBR_NA], acc[SBR_NA], y7 = xp, acc7, q = 0.0, qn = 0.0;
        SbrRho o;
        // what the inflow adds to the stage derivative of w[] (Sno's is carried over nu9_3: qn = q/nu9_3)
        auto flow = [&](const double (&w)[SBR_NA]) {
            if (FLOW == 2) {
                k[A_SS] = __builtin_fma(q, ld[2] - w[A_SS], k[A_SS]); k[A_XS] = __builtin_fma(q, ld[4] - w[A_XS], k[A_XS]);
                k[A_XBH] = __builtin_fma(q, ld[5] - w[A_XBH], k[A_XBH]); k[A_XBA] = __builtin_fma(q, ld[6] - w[A_XBA], k[A_XBA]);
                k[A_SO] = __builtin_fma(q, ld[8] - w[A_SO], k[A_SO]); k[A_SNO] = __builtin_fma(qn, ld[9] - w[A_SNO], k[A_SNO]);
                k[A_SNH] = __builtin_fma(q, ld[10] - w[A_SNH], k[A_SNH]); k[A_SND] = __builtin_fma(q, ld[11] - w[A_SND], k[A_SND]);
                k[A_XND] = __builtin_fma(q, ld[12] - w[A_XND], k[A_XND]);
            }
        };
        // Xp: acc7 += w_s (nu7 (rho4+rho5) + q (c_in - w7)), as two FMAs
        auto xp_in = [&](double w7) { return ld[7] - w7; };
        // stage 1 at (t, V)
        sbr_rates(p, m, a, kla, k, o);
        if (FLOW) { q = Q * rv; qn = q * p.inv_n9_3; flow(a); }
#pragma unroll
        for (int i = 0; i < SBR_NA; ++i) {
            acc[i] = __builtin_fma(i == A_SNO ? g6 : h6, k[i], a[i]); y[i] = __builtin_fma(i == A_SNO ? g2 : h2, k[i], a[i]);
        }
        acc7 = __builtin_fma(p6, o.s45b, xp);
        if (FLOW) { y7 = __builtin_fma(h2 * q, xp_in(xp), __builtin_fma(p2, o.s45b, xp)); acc7 = __builtin_fma(h6 * q, xp_in(xp), acc7); }
        // stages 2 and 3 at (t + h/2, V + h/2 Q)
        if (FLOW) {
            const double vm = __builtin_fma(h2, Q, v);
            rv = sbr_rcp(vm);
            q = Q * rv; qn = q * p.inv_n9_3;
        }
        sbr_rates(p, m, y, kla, k, o);
        if (FLOW) flow(y);
        acc7 = __builtin_fma(p3, o.s45b, acc7);
        if (FLOW) { acc7 = __builtin_fma(h3 * q, xp_in(y7), acc7); y7 = __builtin_fma(h2 * q, xp_in(y7), __builtin_fma(p2, o.s45b, xp)); }
#pragma unroll
        for (int i = 0; i < SBR_NA; ++i) {
            acc[i] = __builtin_fma(i == A_SNO ? g3 : h3, k[i], acc[i]); y[i] = __builtin_fma(i == A_SNO ? g2 : h2, k[i], a[i]);
        }
        sbr_rates(p, m, y, kla, k, o);
        if (FLOW) flow(y);
        acc7 = __builtin_fma(p3, o.s45b, acc7);
        if (FLOW) { acc7 = __builtin_fma(h3 * q, xp_in(y7), acc7); y7 = __builtin_fma(h * q, xp_in(y7), __builtin_fma(p1, o.s45b, xp)); }
#pragma unroll
        for (int i = 0; i < SBR_NA; ++i) {
            acc[i] = __builtin_fma(i == A_SNO ? g3 : h3, k[i], acc[i]); y[i] = __builtin_fma(i == A_SNO ? g1 : h, k[i], a[i]);
        }
        // stage 4 at (t + h, V + h Q)
        if (FLOW) {
            v = __builtin_fma(h, Q, v);
            rv = sbr_rcp(v);
            q = Q * rv; qn = q * p.inv_n9_3;
        }
        sbr_rates(p, m, y, kla, k, o);
        if (FLOW) flow(y);
#pragma unroll
        for (int i = 0; i < SBR_NA; ++i) a[i] = __builtin_fma(i == A_SNO ? g6 : h6, k[i], acc[i]);
        xp = __builtin_fma(p6, o.s45b, acc7);
        if (FLOW) xp = __builtin_fma(h6 * q, xp_in(y7), xp);
    }
    sbr_scatter(a, x);
    x[7] = xp;
    // the components outside the stages
    const double c14 = 1.0 / 14.0;
    double u = __builtin_fma(-n0, c14, x[13]);                 // Salk - (Snh - Sno)/14 at the start
    if (FLOW) {
        const double g = (v - v0) * rv;                        // share of the final volume that flowed in
        const double u_in = __builtin_fma(-(ld[10] - ld[9]), c14, ld[13]);
        x[0] = v;
        x[1] = __builtin_fma(g, ld[1] - x[1], x[1]);
        x[3] = __builtin_fma(g, ld[3] - x[3], x[3]);
        u = __builtin_fma(g, u_in - u, u);
    }
    x[13] = __builtin_fma(x[10] - x[9], c14, u);
}

// ---------------------------------------------------------------------------------------------------
// cfg.scheme = 1 ("B5A", round 5): a control interval by Butcher's six-stage fifth-order Runge-Kutta scheme with a step count
// chosen PER LANE from the env's own state, instead of ten RK4 substeps (the CPU oracle's RK4-mode layers under
// oracle/ - test infrastructure, never linked here - implement the same rule as `b5a_reaction` / `b5a_interval`).
//   * The only stiff mode of reaction_dxdt (:1658-1787) is the relaxation of dissolved oxygen,
//         lam(So) = a1 Koh/(Koh+So)^2 + a3 Koa/(Koa+So)^2 + Kla,  a1 = -nu8_1 muH Ss/(Ks+Ss) Xbh,  a3 = -nu8_3 muA Snh/(Knh+Snh) Xba;
//     every other mode has |lambda| t_delta <= ~1.3, which one or two fifth-order steps resolve far below the parity gate.
//   * slaved: |So| < 1e-9 and an aeration that could not lift it above that within the interval (the anoxic phases once the
//     oxygen is used up; the reference's So is 1e-10 ... 1e-52 there): So is HELD during two steps (its slope multiplied by
//     m_so = 0) and damped afterwards by 1/(1 + lam(0) span);
//   * otherwise n = 1, 2 or 4 steps from z = lam(So_lo) span < 0.3 / < 1.0 / else, So_lo = max(0, min(So, So + So' span)): the
//     lowest So a linear projection reaches (consumption slows as So falls, so it bounds So from below and z from above);
//   * in the last case (the knee, So moving through Koh) n = max(4, floor(lam(0) span / 2.5) + 1): the worst-case lam(0) h stays
//     below 2.5 (Butcher-5 is stable on the real axis up to 3.39); the reference plant needs four steps (five in one golden episode);
//   * a1 and a3 are taken at the projected upper ends of Ss and Snh over the interval, and the count is never below what the
//     movement of the other Monod arguments asks for (zs = max |slope| span / (K + |x|) over Ss, Snh, Sno: < 0.15 -> 1, < 0.5 -> 2,
//     else 4): no effect on the reference's regime, robustness for plants driven far from it (scripts/analysis/plan_probe.py).
// The step count is a function of the lane's own state only, and lanes that take fewer steps than their wave-mates are masked
// out of the later iterations: an env's bits do not depend on which envs share its wavefront (as with the dosing ballot).
// The steps are written in running-sum form - each stage slope is added to the pending stage bases and the result as soon as
// it exists (17 vector FMAs per step, at most five 9-vectors live) - with the step-size products h a_ij per lane in VGPRs.
//   * (round 6: a further floor from the decay rates of the Ss / Snh / Sno modes themselves, n >= floor(j span / 2.5) + 1, was built
//     and measured - without effect on any reference-regime interval, it removes half of the blow-ups of a plant with 4 x faster
//     kinetics - and NOT adopted: three more wave-uniform constants in the plan cost k_step 55 more SGPR spill moves per call,
//     +0.2 us of 11.7 (profiles/r06_notes.md).  The validity domain is stated next to `scheme` in sbr_amd.h instead.)
//   DOSE: the scaled-mass form of sbr_rk4_dose (w = c V/V0; the Monod constants at the five distinct stage times of a step,
//   the constant source of Ss added to every stage slope); a lane with Q == 0 computes the plain form's values.
//   (Round 6 also built a QUASI-STEADY rule for the knee branch - when aeration balances uptake inside the knee nothing moves, and
//   stability alone, max(2, floor(z / 2.5) + 1) steps, would do: applied to control intervals it put the per-cycle env's cycle-end
//   state 0.55 of the gate off (a controller acts on So every interval and feeds the error back); applied to the idle phase of
//   the done call only (Kla held) it was accurate, 0.0036 of the gate, and took 231 -> 204 us off that call = 0.06 us per call of
//   an episode: measured, not adopted.  profiles/r06_notes.md section 2; scripts/analysis/knee_study.py idle.)
// Returns the plan it ran with: step count (<= 64) + SBR_PLAN_SLAVED if dissolved oxygen was held (sbr_amd.h, SBR_C_PLAN).
struct SbrB5C { double a21, a31, a42, a51, a54, a61, a62, a63, a65, b1, b3, b4; };
template <bool DOSE>
SBR_DEV int sbr_b5a(const SbrPar& p, double (&x)[SBR_NX], double span, double kla, double Q) {
    const double kla_sat = kla * p.So_sat;
    const double v0 = x[0], n0 = x[10] - x[9];
    double a[SBR_NA], k[SBR_NA], xp = x[7], e = 0.0, rs = 1.0;
    SbrRho o;
    sbr_gather(x, a);
    const double rv0 = DOSE ? sbr_rcp(v0) : 0.0;
    const double qv = Q * rv0;                                   // Q / V0
    const double srcr = qv * p.EC_conc;                          // the source of w_Ss (a rate)
    SbrMonod m = sbr_monod(p, kla_sat);                          // at s = 1: the model's own constants
    // stage 1 of the first step: does not depend on the step size, and its oxygen slope feeds the plan
    sbr_rates(p, m, a, kla, k, o);
    if (DOSE) k[A_SS] = k[A_SS] + srcr;
    k[A_SNO] = k[A_SNO] * p.n9_3;
    // ---- the plan (oracle: b5a_plan).  Everything the oxygen rate is built from is taken at its upper bound over the interval
    // under a linear projection of the slow variables: Ss and Snh at max(start, start + slope span) (carbon dosing raises Ss
    // within an interval), So at the lowest value the steeper of (its start slope, its slope with the projected substrate levels)
    // reaches.  One reciprocal for the four Monod quotients at the start and projected levels, one for the two oxygen terms.
    const double so = a[A_SO], ss = a[A_SS], snh = a[A_SNH];
    const double p2 = __builtin_fma(k[A_SS], span, ss), p10 = __builtin_fma(k[A_SNH], span, snh);
    const double ss_hi = p2 > ss ? p2 : ss, snh_hi = p10 > snh ? p10 : snh;
    const double e1 = p.Ks + ss, e2 = p.Ks + ss_hi, e3 = p.Knh + snh, e4 = p.Knh + snh_hi;
    const double e12 = e1 * e2, e34 = e3 * e4;
    const double Rm = sbr_rcp(e12 * e34);
    const double r12 = Rm * e34, r34 = Rm * e12;                 // 1/((Ks+Ss)(Ks+Ss_hi)), 1/((Knh+Snh)(Knh+Snh_hi))
    const double m1s = ss * (r12 * e2), m1 = ss_hi * (r12 * e1), m3s = snh * (r34 * e4), m3 = snh_hi * (r34 * e3);
    const double c1 = (-p.n8_1 * p.muH) * a[A_XBH], c3 = (-p.n8_3 * p.muA) * a[A_XBA];
    const double a1 = c1 * m1, a3 = c3 * m3;
    const double g1 = p.Koh + so, g3 = p.Koa + so;
    const double Rg = sbr_rcp(g1 * g3);
    const double slope_hi = __builtin_fma(-(c1 * (m1 - m1s)), so * (Rg * g3), __builtin_fma(-(c3 * (m3 - m3s)), so * (Rg * g1), k[A_SO]));
    const double slope = slope_hi < k[A_SO] ? slope_hi : k[A_SO];
    const double proj = __builtin_fma(slope, span, so);
    const double lo1 = proj < so ? proj : so;
    const double so_lo = lo1 > 0.0 ? lo1 : 0.0;
    const double dh = p.Koh + so_lo, da = p.Koa + so_lo;
    const double dh2 = dh * dh, da2 = da * da;
    const double R = sbr_rcp(dh2 * da2);
    const double z_ub = __builtin_fma(a1 * p.Koh, R * da2, __builtin_fma(a3 * p.Koa, R * dh2, kla)) * span;
    const double lam0 = __builtin_fma(a1, p.inv_Koh, __builtin_fma(a3, p.inv_Koa, kla));
    const bool slaved = (fabs(so) < 1e-9) && (kla_sat * span < 1e-9);
    // n = 2 (slaved) / 1 / 2 / the knee: max(4, floor(lam(0) span / 2.5) + 1), capped at 64 so that every wave terminates
    const double qn = lam0 * span * (1.0 / 2.5);
    // ... for a state INSIDE the model's domain.  The count grows with lam(0) because a1 and a3 are bounded there (the Monod factors of
    // Ss and Snh lie in [0, 1]); outside - Ss or Snh negative towards or beyond its pole, or NaN - the premise is gone, the state is
    // garbage (the reference has no guards either; SBR_ST_NEAR_POLE says so) and the count stays at the knee's four: a launch lasts as
    // long as its slowest wavefront, and ONE such lane at 64 steps made a 65 536-env launch take 60 us instead of 11 (round 6: the
    // uniform policy's late-episode calls 27 us, its done call up to 3 ms).  No in-domain state is affected.
    // (written as two selects on qn, each consuming its comparison at once: a combined lane mask would live in scalar registers across
    // the plan, and the fused kernels' step loops pay for scalar pressure with spill moves.  A NaN factor fails its comparison.)
    const double qn1 = fabs(m1 - 0.5) <= 0.5 ? qn : 0.0;
    const double qn2 = fabs(m3 - 0.5) <= 0.5 ? qn1 : 0.0;
    const int n_knee = qn2 < 4.0 ? 4 : (!(qn2 < 64.0) ? 64 : (int)qn2 + 1);
    const int n_z = slaved ? 2 : (z_ub < 0.3 ? 1 : (z_ub < 1.0 ? 2 : n_knee));
    // ... and never fewer steps than the other Monod arguments ask for: |slope| span / (K + |x|) of Ss, Snh, Sno
    const double f2 = p.Ks + fabs(ss), f10 = p.Knh + fabs(snh), f9 = p.Kno + fabs(a[A_SNO]);
    const double Rs = sbr_rcp((f2 * f10) * f9);
    const double zs2 = fabs(k[A_SS]) * (Rs * (f10 * f9)), zs10 = fabs(k[A_SNH]) * (Rs * (f2 * f9)), zs9 = fabs(k[A_SNO]) * (Rs * (f2 * f10));
    const double zsm = (zs2 > zs10 ? (zs2 > zs9 ? zs2 : zs9) : (zs10 > zs9 ? zs10 : zs9)) * span;
    const int n_s = zsm < 0.15 ? 1 : (zsm < 0.5 ? 2 : 4);
    const int n = n_s > n_z ? n_s : n_z;
    const double m_so = slaved ? 0.0 : 1.0;
    // span / n: exact for 1, 2 and 4 (nearly every step count of the reference plant); one reciprocal for the rest
    const double h = n == 1 ? span : (n == 2 ? span * 0.5 : (n == 4 ? span * 0.25 : span * sbr_rcp((double)n)));
    k[A_SO] = k[A_SO] * m_so;
    SbrB5C c;
    c.a21 = h * 0.25; c.a31 = h * 0.125; c.a42 = h * -0.5; c.a51 = h * (3.0 / 16.0); c.a54 = h * (9.0 / 16.0);
    c.a61 = h * (-3.0 / 7.0); c.a62 = h * (2.0 / 7.0); c.a63 = h * (12.0 / 7.0); c.a65 = h * (8.0 / 7.0);
    c.b1 = h * (7.0 / 90.0); c.b3 = h * (32.0 / 90.0); c.b4 = h * (12.0 / 90.0);
    const double dl = DOSE ? h * qv : 0.0;                       // growth of s = V/V0 per step
    // one stage: slopes of y at scaled-mass time e_st
    auto stage = [&](const double (&y)[SBR_NA], double e_st) {
        if (DOSE) { const SbrScaled cs = sbr_scale_consts(p, kla_sat, e_st); m = cs.m; rs = cs.rs; }
        sbr_rates(p, m, y, kla, k, o);
        if (DOSE) k[A_SS] = k[A_SS] + srcr;
        k[A_SNO] = k[A_SNO] * p.n9_3;
        k[A_SO] = k[A_SO] * m_so;
    };
#pragma unroll 1
    for (int s = 0; s < n; ++s) {
        // running-sum form, each sum formed as late as its terms allow: at most a, k1 and three pending vectors live
        double k1[SBR_NA], y[SBR_NA], p4[SBR_NA], p5[SBR_NA], p6[SBR_NA], pb[SBR_NA];
        if (s > 0) stage(a, e);                                  // stage 1 (the first step's is above)
        double xq = c.b1 * o.s45b;                               // sum_j h b_j (rho4 + rho5)_j / bH: Xp' = nu7 bH s45b
#pragma unroll
        for (int i = 0; i < SBR_NA; ++i) { k1[i] = k[i]; y[i] = __builtin_fma(c.a21, k[i], a[i]); }
        stage(y, __builtin_fma(0.25, dl, e));                    // stage 2 at t + h/4
#pragma unroll
        for (int i = 0; i < SBR_NA; ++i) {
            y[i] = __builtin_fma(c.a31, k[i], __builtin_fma(c.a31, k1[i], a[i]));
            p4[i] = __builtin_fma(c.a42, k[i], a[i]);
            p6[i] = __builtin_fma(c.a62, k[i], __builtin_fma(c.a61, k1[i], a[i]));
        }
        stage(y, __builtin_fma(0.25, dl, e));                    // stage 3 at t + h/4
        xq = __builtin_fma(c.b3, o.s45b, xq);
#pragma unroll
        for (int i = 0; i < SBR_NA; ++i) {
            y[i] = __builtin_fma(h, k[i], p4[i]); p6[i] = __builtin_fma(c.a63, k[i], p6[i]);
            pb[i] = __builtin_fma(c.b3, k[i], __builtin_fma(c.b1, k1[i], a[i]));
            p5[i] = __builtin_fma(c.a51, k1[i], a[i]);
        }
        stage(y, __builtin_fma(0.5, dl, e));                     // stage 4 at t + h/2
        xq = __builtin_fma(c.b4, o.s45b, xq);
#pragma unroll
        for (int i = 0; i < SBR_NA; ++i) {
            y[i] = __builtin_fma(c.a54, k[i], p5[i]); p6[i] = __builtin_fma(-c.a63, k[i], p6[i]); pb[i] = __builtin_fma(c.b4, k[i], pb[i]);
        }
        stage(y, __builtin_fma(0.75, dl, e));                    // stage 5 at t + 3h/4
        xq = __builtin_fma(c.b3, o.s45b, xq);
#pragma unroll
        for (int i = 0; i < SBR_NA; ++i) { y[i] = __builtin_fma(c.a65, k[i], p6[i]); pb[i] = __builtin_fma(c.b3, k[i], pb[i]); }
        e = e + dl;
        stage(y, e);                                             // stage 6 at t + h: its constants are the next step's stage-1 constants
        xp = __builtin_fma(p.n7_45b, __builtin_fma(c.b1, o.s45b, xq), xp);
#pragma unroll
        for (int i = 0; i < SBR_NA; ++i) a[i] = __builtin_fma(c.b1, k[i], pb[i]);
    }
    const double c14 = 1.0 / 14.0;
    if (DOSE) {
        // back to concentrations: c = w / s_end (rs = 1/s at the last stage time = s_end); Si, Xi and the charge balance only dilute
#pragma unroll
        for (int i = 0; i < SBR_NA; ++i) a[i] = a[i] * rs;
        xp = xp * rs;
        x[0] = __builtin_fma(v0, e, v0);
        x[1] = x[1] * rs; x[3] = x[3] * rs;
    }
    a[A_SO] = slaved ? a[A_SO] * sbr_rcp(__builtin_fma(lam0, span, 1.0)) : a[A_SO];
    const double u = DOSE ? __builtin_fma(-n0, c14, x[13]) * rs : __builtin_fma(-n0, c14, x[13]);
    sbr_scatter(a, x);
    x[7] = xp;
    x[13] = __builtin_fma(x[10] - x[9], c14, u);
    return n + (slaved ? SBR_PLAN_SLAVED : 0);
}
// m macro intervals of span/m each, closed reactor (the idle phase of the done call: m = ceil(rows / 10))
SBR_DEV void sbr_b5a_span(const SbrPar& p, double (&x)[SBR_NX], double span, int m, double kla) {
    const double hm = span * sbr_rcp((double)(m > 0 ? m : 1));
#pragma unroll 1
    for (int j = 0; j < m; ++j) (void)sbr_b5a<false>(p, x, hm, kla, 0.0);
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
    int plans;                // scheme 1: plan code of the last interval run (bits 0-7) and of the call's first (bits 8-15); 0 under scheme 0
    double span;              // t_range[-1] - t_range[0] of the last interval
    int rows;                 // len(t_range) of the last interval: 9 or 10
};
// the diagnostics sbr_reward appends to its four lists (module_reward_EQIOCI.py:109-112), before normalisation
struct SbrRewardParts { double eqi2, ae, ec; };

// the six components whose change over the (last) interval the observation reports (:1069-1076)
#define SBR_NXD 6
SBR_DEV void sbr_take6(const double (&x)[SBR_NX], double (&x6)[SBR_NXD]) {
    x6[0] = x[2]; x6[1] = x[5]; x6[2] = x[6]; x6[3] = x[8]; x6[4] = x[9]; x6[5] = x[10];
}
// where the start values of an interval are parked while the RK4 loop runs: registers, or this lane's LDS slots
struct SbrX6Reg {
    static constexpr bool kPark = false;
    SBR_DEV void park(int, double) const {}
    SBR_DEV double unpark(int) const { return 0.0; }
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
template <bool PARK>
struct SbrX6LdsT {         // slot j of the lane lives at base[j * 64] inside its wave's region: conflict-free, 8-byte accesses
    // PARK (the two-waves-per-SIMD build of k_step): what the call keeps across the integration goes to the lane's slots behind
    // the six start values (and the OCI sum) while the step loop runs, so that the loop has the registers
    static constexpr bool kPark = PARK;
    SBR_DEV void park(int j, double v) const { base[(SBR_NXD + 1 + j) * 64] = v; }
    SBR_DEV double unpark(int j) const { return base[(SBR_NXD + 1 + j) * 64]; }
    double* base;
    SBR_DEV void put(const double (&x)[SBR_NX]) {
        base[0 * 64] = x[2]; base[1 * 64] = x[5]; base[2 * 64] = x[6]; base[3 * 64] = x[8]; base[4 * 64] = x[9];
        base[5 * 64] = x[10];
    }
    SBR_DEV void get(double (&o)[SBR_NXD]) const {
#pragma unroll
        for (int j = 0; j < SBR_NXD; ++j) o[j] = base[j * 64];
    }
};
using SbrX6Lds = SbrX6LdsT<false>;

// the small integers of SbrCtl as ONE exactly representable double (the parked build of k_step keeps them in an LDS slot):
// rows (<= 10) | st_new (< 8) << 4 | n_new (<= 2) << 7 | plans (16 bits) << 9
SBR_DEV double sbr_pack_small(const SbrCtl& c) { return (double)(c.rows + (c.st_new << 4) + (c.n_new << 7) + (c.plans << 9)); }
SBR_DEV void sbr_unpack_small(double v, SbrCtl& c) {
    const int pk = (int)v;
    c.rows = pk & 15; c.st_new = (pk >> 4) & 7; c.n_new = (pk >> 7) & 3; c.plans = (pk >> 9) & 0xffff;
}

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
// TR: where the NO3-PID's e / ie / dcv of every interval go (trajectory export; SbrNoTrace for kernels without one).
struct SbrNoTrace { SBR_DEV void pid(int, double, double, double) const {} };
// SCH: cfg.scheme at compile time (a kernel instantiation per scheme: the scheme-1 kernels carry no RK4 loop for the intervals).
template <int SCH, typename X6, typename TR>
SBR_DEV void sbr_interval(const SbrPar& p, SbrCtl& c, double (&x)[SBR_NX], X6& xs6, bool aerobic, const TR& tr) {
    const double t0 = c.t, t1 = t0 + p.t_delta;
    const double span = t1 - t0;
    // len(t_range) = int(span/dt): 9 or 10 with the fp rounding of (t+t_delta)-t (:1339, :1384).  Exact for every span:
    // the two thresholds decide the values a valid t can produce, anything else takes the IEEE division.
    // (Written with selects, like the clamps below: every `if` of this function used to compile to an s_cbranch_execz over a
    // handful of instructions, and a taken branch costs a single resident wave an instruction-buffer refill - more than the
    // instructions it skips.)
    c.rows = span >= p.rows10_min ? 10 : 9;            // min(int(span/dt), SBR_KLA_HIST): the quotient is monotonic in span
    if (__builtin_expect(!(span >= p.rows9_min), 0)) {
        c.rows = (int)(span / p.dt);
        c.rows = c.rows < 2 ? 2 : (c.rows > SBR_KLA_HIST ? SBR_KLA_HIST : c.rows);   // bounded even if t was injected as garbage
    }
    const bool deriv = (p.KcD_DO != 0.0) || (p.KcD_EC != 0.0);            // wave-uniform; tauD = 0 in the reference (:85, :94)
    // DO-PID -> Kla (velocity form: bias is the previous Kla).  In anoxic intervals the output is forced to 0
    // but the integral keeps winding with set-point 0 (:1974-1997).
    const double e = (aerobic ? c.u_do : 0.0) - c.so_m1;
    const double edt = e * p.dt;
    const double ie_do1 = c.ie_do + edt;
    double kla = __builtin_fma(p.Kc_DO, e, p.KcI_DO * ie_do1);
    if (deriv) kla = kla + p.KcD_DO * ((c.so_m1 - c.so_m2) * p.inv_dt);
    kla = aerobic ? kla + c.kla_last : 0.0;
    {   // clamp to [Kla_min, Kla_max]; each clamp that fires takes e dt out of the integral again (:1904-1909: two sequential ifs)
        const bool hi = kla > p.Kla_max;
        const double k1 = hi ? p.Kla_max : kla, i1 = hi ? ie_do1 - edt : ie_do1;
        const bool lo = k1 < p.Kla_min;
        kla = lo ? p.Kla_min : k1;
        c.ie_do = lo ? i1 - edt : i1;
    }
    // NO3-PID -> EC (error sign reversed, :2006); forced to 0 in aerobic intervals while its integral winds (:1918-1937)
    const double e2 = c.sno_m1 - c.u_ec;
    const double e2dt = e2 * p.dt;
    const double dcv2 = (c.sno_m1 - c.sno_m2) * p.inv_dt;
    const double ie_ec1 = c.ie_ec + e2dt;
    double ec = __builtin_fma(p.Kc_EC, e2, p.KcI_EC * ie_ec1);
    if (deriv) ec = ec + p.KcD_EC * dcv2;
    ec = aerobic ? 0.0 : ec + c.ec_last;
    {   // if / elif (:1939-1944, :2030-2035)
        const bool lo = ec < p.EC_min, hi = !lo && ec > p.EC_max;
        ec = lo ? p.EC_min : (hi ? p.EC_max : ec);
        c.ie_ec = (lo || hi) ? ie_ec1 - e2dt : ie_ec1;
    }
    tr.pid(c.n_new, e2, c.ie_ec, dcv2);                // trajectory export only (a no-op type otherwise)
    xs6.put(x);
    // wave-uniform choice of the code path only: a lane with ec == 0 computes the same bits in either (sbr_rk4)
    double nold[SBR_NX];
#pragma unroll
    for (int i = 0; i < SBR_NX; ++i) nold[i] = 0.0;
    // what the rest of the call needs of this interval; the two-waves-per-SIMD build of k_step (X6::kPark) keeps none of it - nor
    // the controller state - in registers while the step loop runs: 13 values go to the lane's LDS slots and come back
    double t1r = t1, klar = kla, ecr = ec, spanr = span;
    if constexpr (X6::kPark) {
        xs6.park(0, t1); xs6.park(1, c.so_m1); xs6.park(2, c.sno_m1); xs6.park(3, c.ie_do); xs6.park(4, c.ie_ec);
        xs6.park(5, c.ec_last); xs6.park(6, ec); xs6.park(7, c.u_do); xs6.park(8, c.u_ec); xs6.park(9, kla);
        xs6.park(10, c.knew[0]); xs6.park(11, span);
        xs6.park(12, sbr_pack_small(c));                  // rows, status bits, interval count, plans: small integers, exact
        asm volatile("" ::: "memory");
    }
    int plan = 0;
    if constexpr (SCH == 1) {
#ifdef SBR_B5_ONE_FORM
        plan = sbr_b5a<true>(p, x, span, kla, ec);
#else
        if (__builtin_amdgcn_ballot_w64(ec != 0.0) == 0ull) plan = sbr_b5a<false>(p, x, span, kla, 0.0);
        else plan = sbr_b5a<true>(p, x, span, kla, ec);
#endif
    } else {
        const double h = span * p.inv_substeps;
        if (__builtin_amdgcn_ballot_w64(ec != 0.0) == 0ull) sbr_rk4<0>(p, x, h, p.substeps, kla, 0.0, nold);
        else sbr_rk4<1>(p, x, h, p.substeps, kla, ec, nold);
    }
    if constexpr (X6::kPark) {
        asm volatile("" ::: "memory");
        t1r = xs6.unpark(0); klar = xs6.unpark(9); ecr = xs6.unpark(6); spanr = xs6.unpark(11);
        sbr_unpack_small(xs6.unpark(12), c);
        c.knew[0] = xs6.unpark(10);
        c.so_m1 = xs6.unpark(1); c.sno_m1 = xs6.unpark(2); c.ie_do = xs6.unpark(3); c.ie_ec = xs6.unpark(4);
        c.ec_last = xs6.unpark(5); c.u_do = xs6.unpark(7); c.u_ec = xs6.unpark(8);
    }
    if (c.n_new == 0) c.knew[0] = klar; else c.knew[1] = klar;    // n_new <= 2, see SbrCtl
    c.plans = c.n_new == 0 ? (plan | (plan << 8)) : ((c.plans & 0xff00) | plan);
    c.n_new += 1;
    c.kla_last = klar;
    c.ec_prev = c.ec_last; c.ec_last = ecr;
    c.so_m2 = c.so_m1; c.so_m1 = x[8];
    c.sno_m2 = c.sno_m1; c.sno_m1 = x[9];
    c.t = t1r; c.span = spanr;
    c.st_new |= sbr_status_bits(p, x);
}

// The phase logic of SbrOS.step (:860-1010): four sequential `if`s on the running time (:860 anoxic, :896 aerobic, :931
// anoxic, :963 aerobic), each with its own clipping of the action; a call that crosses a phase boundary runs a second
// interval (3 times per episode).  A third cannot fire: sbr_create checks that every phase is longer than t_delta.
// The four conditions are mutually exclusive for a given t, so the first one that fires is phase(t); after its interval t
// has advanced, and a LATER test fires in the same call exactly when phase(t) has increased.
// Two equivalent forms, chosen per kernel by measurement (profiles/r02_notes.md, r02_ab_phase_logic.log, r02_ab_loop2.log):
//  LOOP = true   two passes of "run an interval if phase(t) is past the last one run", so that the interval code exists
//                ONCE in the kernel (k_step is sensitive to code size: 14.33 us per launch; a loop over the reference's four
//                tests 14.68; straight-line 14.9);
//  LOOP = false  straight-line, the second interval behind an unlikely branch (k_rollout: 7.7 us per call against 8.0 - in a
//                loop the plant is a loop-carried value, ~60 register copies per call).
// Envs reset together are in lockstep, so the branches are wave-uniform in practice; divergent waves are still correct.
SBR_DEV int sbr_phase(const SbrPar& p, double t) {
    // = t < T3_0 ? 0 : (t <= T3_end ? 1 : (t <= T4_end ? 2 : (t > T4_end ? 3 : -1))), as a sum of comparisons (no branches);
    // every comparison is false for a NaN: -1
    return (t >= p.T3_0 ? 1 : 0) + (t > p.T3_end ? 1 : 0) + (t > p.T4_end ? 1 : 0) - (t == t ? 0 : 1);
}
template <bool LOOP, int SCH, typename X6, typename TR>
SBR_DEV void sbr_run_intervals(const SbrPar& p, SbrCtl& c, double (&x)[SBR_NX], double a0, double a1, X6& xs6, const TR& tr) {
    a0 = a0 < 0.0 ? 0.0 : (a0 > p.act_DO_max ? p.act_DO_max : a0);       // :901-906
    a1 = a1 < 0.0 ? 0.0 : (a1 > p.act_EC_max ? p.act_EC_max : a1);       // :865-870
    c.n_new = 0; c.st_new = 0; c.plans = 0;
    c.knew[0] = c.kla_last; c.knew[1] = c.kla_last;      // defined even if no interval runs (t injected as NaN: sbr_phase = -1)
    if (LOOP) {
        int last = -1;
#pragma unroll 1
        for (int it = 0; it < 2; ++it) {
            const int ph = sbr_phase(p, c.t);
            if (ph > last) {
                const bool aerobic = (ph & 1) != 0;
                c.u_do = aerobic ? a0 : 0.0; c.u_ec = aerobic ? 0.0 : a1;
                sbr_interval<SCH>(p, c, x, xs6, aerobic, tr);
                last = ph;
            }
        }
    } else {
        const int ph = sbr_phase(p, c.t);
        if (ph >= 0) {
            const bool aerobic = (ph & 1) != 0;
            c.u_do = aerobic ? a0 : 0.0; c.u_ec = aerobic ? 0.0 : a1;
            sbr_interval<SCH>(p, c, x, xs6, aerobic, tr);
            const int ph2 = sbr_phase(p, c.t);
            if (__builtin_expect(ph2 > ph, 0)) {
                const bool aerobic2 = (ph2 & 1) != 0;
                c.u_do = aerobic2 ? a0 : 0.0; c.u_ec = aerobic2 ? 0.0 : a1;
                sbr_interval<SCH>(p, c, x, xs6, aerobic2, tr);
            }
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
// Where a kernel keeps the Kla history.  old(i), i = 0..9, is the list's tail BEFORE this call's appends (oldest first,
// old(9) = Kla[-1]); commit_and_window() applies the appends and returns the reward's window sum; push() is Sim_idle's
// extra append.
struct SbrHistReg {                       // the fused kernels: ten registers, shifted in place
    double (&h)[SBR_KLA_HIST];
    SBR_DEV double old(int i) const { return h[i]; }
    SBR_DEV void push(double k) { sbr_hist_push(h, k); }
    // apply this call's appends and return sum(Kla[-rows:-1]) of the list as it then stands (module_reward_EQIOCI.py:70),
    // left to right like python's sum(): the rows-1 values before the current one
    SBR_DEV double commit_and_window(const SbrCtl& c) {
        sbr_hist_apply(c, h);
        double s = (c.rows >= 10) ? h[0] : 0.0;          // 9 previous values if rows == 10, else 8
#pragma unroll
        for (int j = 1; j < SBR_KLA_HIST - 1; ++j) s = s + h[j];
        return s;
    }
};
// k_step's Kla history (round 4): nothing but what a call needs.  The reward sums the 8 or 9 entries before Kla[-1]
// (module_reward_EQIOCI.py:70, sbr_reward); instead of reading the last ten entries every call, the handle keeps
//   w8   = the sum of the 8 entries before Kla[-1]          (row R_W8),
//   last = Kla[-1], the bias of the velocity-form DO-PID     (row R_KLA_LAST),
// and every append moves the window by one: the entry that leaves it - Kla[-9] for the first append of a call, Kla[-8] and
// Kla[-7] for a second (phase-boundary call) and a third (the idle phase of the done call) - comes from the 10-slot ring,
// whose slot depends on the interval count: three loads issued as soon as t has arrived, used seven microseconds later.
// Per call: 5 rows read instead of 10, no parking of the ring in LDS.  A running sum is not python's left-to-right sum():
// the two differ by rounding (<= 1e-10 absolute after a whole episode, 3e-16 in the reward); with equal entries (Kla saturated
// or 0, the common case) every update is exact.
struct SbrHistInc {
    double w8, last;
    double lv[3];
    double idle;
    bool idle_pushed;
    // sum(Kla[-rows:-1]) after this call's n_new appends: rows - 1 = 8 or 9 entries before the new Kla[-1]
    SBR_DEV double commit_and_window(const SbrCtl& c) const {
        const bool nine = c.rows >= 10;
        if (c.n_new <= 0) return w8;                                   // no interval ran (t is NaN): the list is unchanged (rows = 9)
        const double w1 = w8 + last;                                   // the 9 entries before the first append
        if (c.n_new == 1) return nine ? w1 : w1 - lv[0];
        const double w2 = (w1 - lv[0]) + c.knew[0];                    // the 9 entries before the second append
        return nine ? w2 : w2 - lv[1];
    }
    SBR_DEV void push(double k) { idle = k; idle_pushed = true; }      // Sim_idle's Kla.append (:2578)
    // w8 and Kla[-1] as the next call will need them, after all appends of this one
    SBR_DEV void roll(const SbrCtl& c, double& w8_new, double& last_new) const {
        double w = w8, prev = last;
        if (c.n_new > 0) { w = (w - lv[0]) + prev; prev = c.knew[0]; }
        if (c.n_new > 1) { w = (w - lv[1]) + prev; prev = c.knew[1]; }
        if (idle_pushed) {                              // (selects, not a dynamic index: that would put lv[] in scratch memory)
            const double leaving = c.n_new > 1 ? lv[2] : (c.n_new > 0 ? lv[1] : lv[0]);
            w = (w - leaving) + prev; prev = idle;
        }
        w8_new = w; last_new = prev;
    }
};

// module_reward_EQIOCI.py:4-115.  Kla got one append per interval, EC got rows-1:  Kla[-rows:-1] is
// the rows-1 values BEFORE the current one (ksum: SbrHistInc / SbrHistReg commit_and_window), EC[-rows:-1] = last value of the previous interval +
// (rows-2) x current.
SBR_DEV double sbr_reward(const SbrPar& p, const SbrCtl& c, double ksum, const double (&x)[SBR_NX], SbrRewardParts& rp) {
    const double xi = x[3], xs = x[4], xbh = x[5], xba = x[6], xp = x[7];
    const double bio = xbh + xba, part = (xs + xi) + (bio + xp);
    const double snkj = x[10] + x[11] + x[12] + 0.08 * bio + 0.06 * (xp + xi);
    const double bod5 = 0.25 * (x[2] + xs + (1 - 0.08) * bio);
    const double cod = x[2] + x[1] + part;
    // EQI :40-47 with SS = 0.75 part; EQI2 = EQI/10 (:60); the constant divisors are folded (<= 1 ulp each)
    const double eqi2 = (1.5 * part + cod + 30 * snkj + 10 * x[9] + 2 * bod5) * (0.66 / 1000 / 10);
    const double td = 0.002 / 24;
    // AE_OCI = 8/((t1-t0) 1800) 1.32 sum(Kla) td (:70-71), EC_OCI = EC_conc sum(EC) td/((t1-t0) 1000) (:79): one reciprocal
    const double esum = __builtin_fma((double)(c.rows - 2), c.ec_last, c.ec_prev);
    const double aek = (8 * 1.32 * td / 1800) * ksum, eck = (p.EC_conc * td / 1000) * esum, rs = sbr_rcp(c.span);
    const double oci = (aek + eck) * rs;
    rp.eqi2 = eqi2; rp.ae = aek * rs; rp.ec = eck * rs;        // dead code unless the trajectory export is on
    return (1 - __builtin_fma(eqi2, eqi2, oci * oci)) * (1.0 / 473);
}

// module_reward_continuous_G2ANET.py:4-45 (cfg.reward_kind = 1): piecewise-linear in Ss, So, Sno, Snh of the end state
SBR_DEV double sbr_reward_g2anet(const double (&x)[SBR_NX]) {
    const double ss = x[2], so = x[8], sno = x[9], snh = x[10];
    const double r_ec = ss < 0 ? 1.0 : -(ss - 0) * (1.0 / (10 - 0)) + 1;
    const double r_e = so < 1.5 ? 0.0 : -(1 / (8 - 1.5)) * (so - 8) + 0;
    const double r_sno = sno < 4 ? 1.0 : -(sno - 4) * (1.0 / (20 - 4)) + 1;
    const double r_snh = snh < 4 ? 1.0 : -(snh - 4) * (1.0 / (20 - 4)) + 1;
    return (1 * r_ec + 1.5 * r_e + 2 * r_sno + 2 * r_snh) * (1.0 / 10);
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
    // a = vmax / z * t_set with z = V / area; quotients by per-lane values go through sbr_rcp (1 ulp), those by constants are
    // folded: the terminal phases run once per episode, but an IEEE f64 division is ~30 instructions of kernel text each
    const double a = (p.settler_vmax * t_set) * (p.settler_area * sbr_rcp(x[0])), ea = exp(-a);
    // sX[9-j] = Xf e^-a sum_{m<=j} a^m/m!, sX[0] = 10 Xf - sum(others)
    double term = 1.0, partial = 0.0, others = 0.0;
#pragma unroll
    for (int j = 0; j < 9; ++j) { partial += term; sx[9 - j] = xf * ea * partial; term *= a * (1.0 / (double)(j + 1)); }
#pragma unroll
    for (int j = 1; j < 10; ++j) others += sx[j];
    sx[0] = 10.0 * xf - others;
    return xf;
}

// draw + wastage (:2327-2393 / sub_phases_FB.py:780-855): x becomes the reactor after the draw; returns Qw and, in
// sx_eff, the sludge that left with the effluent (sum(sX[-m:-1]*layer_volume), which drops the last layer)
SBR_DEV double sbr_draw(const SbrPar& p, double (&x)[SBR_NX], const double (&sx)[10], double xf, double& sx_eff) {
    const double vs = x[0];
    const double layer_v = vs * 0.1;
    double resid_v = vs - p.Qeff;
    int m = (int)ceil(rint(p.Qeff * sbr_rcp(layer_v)));   // python round() is half-to-even = rint; Qeff/layer_v = 5.0 in the reference's setup
    m = m < 1 ? 1 : (m > 9 ? 9 : m);
    const int nl = 10 - m;                            // layers that stay
    double wsum = 0.0, se = 0.0;
#pragma unroll
    for (int i = 0; i < 9; ++i) { if (i < nl) wsum = wsum + layer_v * sx[i]; else se = se + sx[i] * layer_v; }
    sx_eff = se;
    double waste = wsum - p.biomass_setpoint * resid_v;
    // the wastage loop (:2363-2382) removes whole layers from the top while more than a layer's sludge is to go, then a
    // part qw of the next one: find that layer first, take ONE quotient, then sum what remains in layer order
    int part = -1;                                    // index of the partially wasted layer
    double qnum = __builtin_nan(""), qden = 1.0;
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        if (i < nl && part < 0) {
            const double rest = waste - layer_v * sx[i];
            if (rest > 0) { waste = rest; resid_v -= layer_v; }
            else { part = i; qnum = waste; qden = sx[i] - p.biomass_setpoint; }
        }
    }
    const double qw = qnum * sbr_rcp(qden);
    if (part >= 0) resid_v -= qw;
    double wkeep = 0.0;                               // sum of the weights that remain, in layer order
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        if (i < nl && part >= 0 && i >= part) {
            double w = layer_v * sx[i];
            if (i == part) w = w - qw * sx[i];
            wkeep = wkeep + w;
        }
    }
    const double scale = (1 / 0.75) * (wkeep * sbr_rcp(resid_v)) * sbr_rcp(xf);
    x[0] = resid_v;
#pragma unroll
    for (int i = 3; i <= 7; ++i) x[i] = x[i] * scale;
    return qw;
}

template <int SCH, typename H>
SBR_DEV double sbr_terminal(const SbrPar& p, SbrCtl& c, H& hs, double (&x)[SBR_NX]) {
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
    if constexpr (SCH == 1) {
        sbr_b5a_span(p, x, span, (n + 9) / 10, kla);  // ceil(rows / 10) macro intervals of the adaptive scheme
    } else {
        double nold[SBR_NX];
#pragma unroll
        for (int i = 0; i < SBR_NX; ++i) nold[i] = 0.0;
        sbr_rk4<0>(p, x, span * sbr_rcp((double)(n > 0 ? n : 1)), n, kla, 0.0, nold);
    }
    hs.push(kla);                                     // Kla.append in Sim_idle (:2578)
    c.kla_last = kla;
    return qw;
}

// What is left of SbrOS.step (:1011-1273) after the intervals: reward, done test, terminal phases.  Returns the reward;
// xa6 is updated to the pre-settle values when the terminal phases run; qw is written only then.
// OCI = the operating-cost reward (cfg.reward_kind 2) with its running sum(Kla) of the episode's list, ksum: a
// compile-time variant, so that the default kernels carry none of it.
// TERMINAL_INLINE = false leaves the terminal phases of the done call to the caller (the fused rollout runs them once, after its
// loop over the calls: with them inside the loop the compiler keeps their working set alive across it - 3 % of the rollout's
// time, profiles/r03_notes.md); the reward of a done call does not depend on them unless OCI.
template <bool OCI, int SCH, typename H, bool TERMINAL_INLINE = true, typename PK = SbrX6Reg>
SBR_DEV double sbr_finish_step(const SbrPar& p, SbrCtl& c, H& hs, double (&x)[SBR_NX],
                               double (&xa6)[SBR_NXD], double& t_obs, bool& dn, double& qw, double& ksum, SbrRewardParts& rp,
                               const PK* pk = nullptr) {
    const double kwin = hs.commit_and_window(c);
    double r;
    rp.eqi2 = 0.0; rp.ae = 0.0; rp.ec = 0.0;
    if (OCI) {
        if (c.n_new > 0) ksum = ksum + c.knew[0];
        if (c.n_new > 1) ksum = ksum + c.knew[1];
        r = sbr_reward_oci(p, 1, c.kla_last, 0.0, 0.0, 0.0);
    } else {
        r = p.reward_kind == 1 ? sbr_reward_g2anet(x) : sbr_reward(p, c, kwin, x, rp);     // wave-uniform choice
    }
    t_obs = c.t;
    dn = false;
    if (c.t >= p.T5_end) {                                               // :1122
        dn = true;
        if ((TERMINAL_INLINE || OCI) && p.terminal) {
            const double snh_eff = x[10];            // solubles pass the settler unchanged: eff_component[3] (:2642)
            if constexpr (PK::kPark) {
                // the two-waves-per-SIMD build: nothing of the call is kept in registers across the idle phase's step loop
                SbrX6LdsT<true> q = *pk;
                q.put(x);                            // the caller fetches xa6 after this function
                q.park(0, r); q.park(1, rp.eqi2); q.park(2, rp.ae); q.park(3, rp.ec); q.park(4, c.t); q.park(5, c.so_m1);
                q.park(6, c.so_m2); q.park(7, c.sno_m1); q.park(8, c.sno_m2); q.park(9, c.ie_ec); q.park(10, c.ec_last);
                q.park(11, c.ec_prev); q.park(12, c.u_ec); q.park(13, c.knew[0]); q.park(14, c.knew[1]); q.park(15, c.span);
                q.park(16, sbr_pack_small(c)); q.park(17, snh_eff); q.park(18, ksum);
                asm volatile("" ::: "memory");
                qw = sbr_terminal<SCH>(p, c, hs, x);
                asm volatile("" ::: "memory");
                r = q.unpark(0); rp.eqi2 = q.unpark(1); rp.ae = q.unpark(2); rp.ec = q.unpark(3); c.t = q.unpark(4);
                c.so_m1 = q.unpark(5); c.so_m2 = q.unpark(6); c.sno_m1 = q.unpark(7); c.sno_m2 = q.unpark(8);
                c.ie_ec = q.unpark(9); c.ec_last = q.unpark(10); c.ec_prev = q.unpark(11); c.u_ec = q.unpark(12);
                c.knew[0] = q.unpark(13); c.knew[1] = q.unpark(14); c.span = q.unpark(15);
                sbr_unpack_small(q.unpark(16), c);
                if (OCI) {
                    ksum = q.unpark(18) + c.kla_last;
                    r = sbr_reward_oci(p, 2, c.kla_last, ksum, qw, q.unpark(17));
                }
                t_obs = p.t_cycle;
                return r;
            }
            sbr_take6(x, xa6);
            qw = sbr_terminal<SCH>(p, c, hs, x);
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
template <bool FILL, int SCH>
SBR_DEV double sbr_cycle_phase(const SbrPar& p, double (&x)[SBR_NX], int ph, double sp, double kla_in,
                               const double (&ld)[SBR_NX], double& ksum) {
    const double t_start = p.cyc_t0[ph], t_end = p.cyc_t1[ph], step = p.cyc_step[ph];   // numpy.linspace: i*step + start, last = stop
    const int n2 = p.cyc_n2[ph], n_iv = n2 - 1;                  // host-clamped to [2, 100000]: every wave terminates
    double so = x[8], so_prev = x[8], ie = 0.0, bias = kla_in, k = kla_in, sum = 0.0;
    for (int i = 0; i < n_iv; ++i) {
        const double g0 = (double)i * step + t_start;
        const double g1 = (i + 1 == n2 - 1) ? t_end : (double)(i + 1) * step + t_start;
        const double e = sp - so;
        double dcv = 0.0;
        if (i >= 1) { dcv = (so - so_prev) * p.inv_cyc_dt; ie = ie + e * p.cyc_dt; }
        k = p.cyc_Kc * e + p.cyc_KcI * ie + p.cyc_KcD * dcv + bias;
        if (k > p.Kla_max) { k = p.Kla_max; ie = ie - e * p.cyc_dt; }
        if (k < p.Kla_min) { k = p.Kla_min; ie = ie - e * p.cyc_dt; }
        if (i == 0) bias = k;
        // scheme 1: every interval but the fill phase's (a fill interval is not plannable from its start state: the inflow raises
        // Ss and Snh severalfold WITHIN the interval and the oxygen uptake with them - measured: up to 70 gates, profiles/r05_notes.md)
        if constexpr (SCH == 1 && !FILL) sbr_b5a<false>(p, x, g1 - g0, k, 0.0);
        else sbr_rk4<FILL ? 2 : 0>(p, x, (g1 - g0) * p.inv_substeps, p.substeps, k, FILL ? ld[0] : 0.0, ld);
        sum = sum + k;
        so_prev = so; so = x[8];
    }
    ksum = sum;
    return k;
}

// SbrEnv2.step (gym_SBR_env2.py:131-171) = SBR_model_FB.run (SBR_model_FB.py:8-295) + module_reward.py:4-51 for one env:
// x is the start state in, the end-of-cycle state out; ld the influent with ld[0] = Qin/t_phs1; a[3] the clipped action.
// obs3 = [Qeff, COD_eff, Snh_eff/30]; diag (SBR_NCYC_DIAG doubles) may be nullptr.
template <int SCH>
SBR_DEV double sbr_cycle_env(const SbrPar& p, double (&x)[SBR_NX], const double (&ld)[SBR_NX], double a0, double a1, double a2,
                             double (&obs3)[3], double* diag, int diag_stride) {
    a0 = a0 < 0.0 ? 0.0 : (a0 > 1.0 ? 1.0 : a0); a1 = a1 < 0.0 ? 0.0 : (a1 > 1.0 ? 1.0 : a1);
    a2 = a2 < 0.0 ? 0.0 : (a2 > 1.0 ? 1.0 : a2);
    const double sp3 = a0 * 8, sp5 = a1 * 8, sp8 = a2 * 8;      // DO_setpoints[2], [4], [7]  (:184-186)
    const double qin = p.WV - x[0];
    double ks1, ks2, ks3, ks4, ks5, ks8;
    // phases 1..5 (fill, anoxic, aerobic, anoxic, aerobic): schedule indices 0..4; the aerated idle is index 5
    double kl = sbr_cycle_phase<true, SCH>(p, x, 0, 0.0, 0.0, ld, ks1);
    kl = sbr_cycle_phase<false, SCH>(p, x, 1, 0.0, kl, ld, ks2);
    kl = sbr_cycle_phase<false, SCH>(p, x, 2, sp3, kl, ld, ks3);
    kl = sbr_cycle_phase<false, SCH>(p, x, 3, 0.0, kl, ld, ks4);
    kl = sbr_cycle_phase<false, SCH>(p, x, 4, sp5, kl, ld, ks5);
    // settle
    double sx[10], sx_eff;
    const double xf = sbr_settle(p, x, p.cyc_tset, sx);
    // draw; the effluent composition (cal_eq, sub_phases_FB.py:860-915) uses the PRE-draw state with its particulates
    // scaled by the sludge carried out: one reciprocal of Xf for the five of them (1 ulp; the reference divides five times)
    const double xi = x[3], xs = x[4], xbh = x[5], xba = x[6], xp = x[7];
    const double qw = sbr_draw(p, x, sx, xf, sx_eff);
    const double carry = ((1 / 0.75) * sx_eff) * sbr_rcp(xf);
    const double exi = xi * carry, exs = xs * carry, exbh = xbh * carry, exba = xba * carry, exp_ = xp * carry;
    const double snkj = x[10] + x[11] + x[12] + 0.08 * (exbh + exba) + 0.06 * (exp_ + exi);
    const double ntot = x[9] + snkj;
    const double ss_ = 0.75 * (exs + exi + exbh + exba + exp_);
    const double bod5 = 0.25 * (x[2] + exs + (1 - 0.08) * (exbh + exba));
    const double cod = x[2] + x[1] + exs + exi + exbh + exba + exp_;
    const double eqi = (2 * ss_ + cod + 30 * snkj + 10 * x[9] + 2 * bod5) * (1.0 / 1000) * 0.66;
    const double snh_eff = x[10], sno_eff = x[9];
    // aerated idle from the drawn reactor, bias = last Kla of phase 5 (SBR_model_FB.py:258)
    sbr_cycle_phase<false, SCH>(p, x, 5, sp8, kl, ld, ks8);
    // reward (module_reward.py:4-51): AE_k = 1.32 sum(Kla) td / (n td), the quotient by the wave-uniform n td folded on the host
    const double td = 0.002 / 24;
    const double ae3 = (1.32 * ks3 * td) * p.cyc_inv_ntd[2], ae5 = (1.32 * ks5 * td) * p.cyc_inv_ntd[4];
    const double ae8 = ((1.32 - qw) * ks8 * td) * p.cyc_inv_ntd[5];
    const double ae = p.sosat_k * (ae3 + ae5 + ae8);
    const double pe = (0.004 * qin + 0.05 * qw + 0.004 * p.Qeff);
    const double me = 0.005 * 1.32 * 24 + 0.005 * 1.32 * 24;
    const double oci = ae + pe + me;
    obs3[0] = p.Qeff; obs3[1] = cod; obs3[2] = snh_eff * (1.0 / 30);
    if (diag) {
        const int st = diag_stride;
        diag[0 * st] = qw; diag[1 * st] = eqi; diag[2 * st] = oci; diag[3 * st] = ntot; diag[4 * st] = cod; diag[5 * st] = snh_eff;
        diag[6 * st] = bod5; diag[7 * st] = sno_eff; diag[8 * st] = ks3 * p.cyc_inv_n[2]; diag[9 * st] = ks5 * p.cyc_inv_n[4];
        diag[10 * st] = ks8 * p.cyc_inv_n[5]; diag[11 * st] = xf;
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
