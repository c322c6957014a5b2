// sbr_amd.hip - kernels + C ABI of libsbr_amd.so (gfx950).  See include/sbr_amd.h for the contract.
//
// Kernels (all one-lane-per-env over SoA float64 state, 256-thread workgroups = four wavefronts, one per SIMD of a CU, so
// a launch of N envs is N/64 independent waves that the dispatcher spreads over the 1024 SIMDs):
//   k_reset    influent mix (tables in LDS) + fill phase (252 RK4 substeps under either scheme) + controller init + obs
//   k_step     one SbrOS.step(): phase logic, 2 PIDs, the interval's integration (cfg.scheme 1: adaptive Butcher-5 steps per lane,
//              sbr_b5a; scheme 0: 10 RK4 substeps; x2 at phase boundaries), reward,
//              obs/state, and the terminal phases on the last call of an episode
//   k_rollout  n_steps fused step()s with an on-device Philox policy, plant state stays in VGPRs
//   k_cycle_reset, k_cycle   the per-cycle env SBR-v2: one launch = one whole 12 h cycle (528 control intervals)
//   k_export, k_import, k_m1_explicit   public <-> internal controller layout; implicit So[-1] / Sno[-1] made explicit
//   k_stats    wavefront (DPP) reductions of a per-env vector -> {sum,min,max,count}
//   k_rhs, k_normals, k_scenarios, k_fill   known-answer helpers for the parity tests, the scenario draw, handle init
// What bounds them and how the arithmetic is organised for it: sbr_device.h (sbr_rates, sbr_rk4), DESIGN.md sections 3 and 5.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>

#ifndef SBR_BLOCK
#define SBR_BLOCK 256     // threads per workgroup of the stepping kernels: four waves, one per SIMD.  Measured at N = 65536
                          // (round 1): 64 -> 23.0 us, 128 -> 22.8, 256 -> 22.65 per k_step launch (fewer workgroups to dispatch)
#endif
#include "sbr_device.h"

#define SBR_RESET_BLOCK 256     // k_reset stages 84 KiB of tables in LDS: one block per CU, so make it four waves
static constexpr int kTableDoubles = SBR_NSCEN * SBR_NSERIES * SBR_NSAMP;   // 5376 doubles = 42 KiB (the layout in HBM)
// In LDS a scenario's block of 14 x 48 doubles is padded to 674: 672 doubles are 1344 dwords = 0 mod 64 banks, so lanes
// that hold the eight different scenarios (scenario = env id mod 8) would all hit the same bank pair on every table read
// (8-way conflict); 674 puts scenario s on banks 4s, 4s+1.
#define SBR_TSTRIDE (SBR_NSERIES * SBR_NSAMP + 2)
static constexpr int kLdsTableDoubles = 2 * SBR_NSCEN * SBR_TSTRIDE;        // means then stds: 86 272 B

// ------------------------------------------------------------------------------------------- state I/O
struct SbrBuf {
    double* trace;  // [capacity][SBR_NTRACE][n_trace] or NULL
    int64_t n_trace, trace_cap;
    double* x;      // [14][N]
    double* ctrl;   // [R_NROWS][N]
    double* infl;   // [14][N]
    int64_t n;
    int64_t first_env_id;
#ifdef SBR_STAMPS
    unsigned long long* stamps;   // diagnostic build only (scripts/probes/step_timeline.py): [waves][16] s_memrealtime ticks
#endif
};
// Diagnostic build -DSBR_STAMPS: lane 0 of every wave records the 100 MHz real-time counter at up to sixteen points of k_step.
// The stamps go to a buffer of their own that nothing else reads; the product build contains none of this.
#ifdef SBR_STAMPS
#define SBR_STAMP(k, drain)                                                                                   \
    do {                                                                                                      \
        if (drain) __builtin_amdgcn_s_waitcnt(0);                                                             \
        if (b.stamps != nullptr && (l & 63u) == 0u)                                                           \
            b.stamps[(uint64_t)((i0 + l) >> 6) * 16 + (k)] = __builtin_amdgcn_s_memrealtime();               \
    } while (0)
#else
#define SBR_STAMP(k, drain) do { } while (0)
#endif

// How the stepping kernels' stores leave the CU: 0 plain, 1 agent-scope write-through (sc1), 2 non-temporal (nt), 3 system-scope
// write-through (sc0 sc1).  Plain stores leave ~21 MB dirty in the L2s for the end-of-kernel write-back, which then sits
// between two dependent launches; written through, the bytes drain while other waves still compute.  Measured per k_step
// launch (profiles/r02_notes.md): 65536 envs 16.06 -> 15.5 us, 131072 envs 28.6 -> 23.1 us; nt gains half of that at 65536 and
// nothing at 131072; sc0 sc1 equals sc1.
#ifndef SBR_ST_MODE
#define SBR_ST_MODE 1
#endif
template <typename T>
SBR_DEV void st_out(T* p, T v) {
#if SBR_ST_MODE == 1
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#elif SBR_ST_MODE == 2
    __builtin_nontemporal_store(v, p);
#elif SBR_ST_MODE == 3
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
#else
    *p = v;
#endif
}
typedef unsigned int sbr_u32x4 __attribute__((ext_vector_type(4)));
// The output staging of k_step reuses the wave's LDS parking region (declared double[]) for rows of OutT and reads it back as
// 16-byte chunks: accesses through these may_alias types are visible to alias analysis (round 4 saw the float32
// staging stores hoisted over the parked double loads).  The compiler barriers at the call sites stay as defence in depth.
typedef sbr_u32x4 __attribute__((may_alias)) sbr_u32x4_alias;
typedef float __attribute__((may_alias)) sbr_f32_alias;
typedef double __attribute__((may_alias)) sbr_f64_alias;
template <typename T> struct SbrAliasOf;
template <> struct SbrAliasOf<float> { using type = sbr_f32_alias; };
template <> struct SbrAliasOf<double> { using type = sbr_f64_alias; };
// The sc1 modifier has no builtin for a 16-byte store, hence inline assembly - and inline assembly is invisible to the
// compiler's hazard recognizer: on gfx940+ a VMEM store of more than 64 bits reads its data registers for two more cycles, and a
// VALU instruction that overwrites them inside that window corrupts the stored value (seen in round 4, once the stores were
// issued back to back: the 64-bit address of the NEXT store was formed in the low half of the data registers of the previous
// one).  The s_nop supplies the two wait states the compiler would have inserted for a store it knows.
SBR_DEV void st_out16(sbr_u32x4* p, sbr_u32x4 v) {
#if SBR_ST_MODE == 1
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" : : "v"(p), "v"(v) : "memory");
#elif SBR_ST_MODE == 2
    __builtin_nontemporal_store(v, p);
#elif SBR_ST_MODE == 3
    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" : : "v"(p), "v"(v) : "memory");
#else
    *p = v;
#endif
}

// Addressing.  An env is (i0, l): i0 = first env of the workgroup (wave-uniform, lives in SGPRs), l = threadIdx.x.  Every
// access is written as (uniform pointer advanced to row and workgroup)[l], so the row arithmetic runs on the scalar unit
// and the lane contributes one 32-bit offset (global_load v, v_off, s[base:base+1]).  The first version formed a 64-bit
// address per lane and row: 173 VALU instructions and ~60 VGPRs of k_step.
#define XROW(j) (b.x + ((int64_t)(j) * b.n + i0))[l]
#define CTRL(f) (b.ctrl + ((int64_t)(f) * b.n + i0))[l]
#define INFL(j) (b.infl + ((int64_t)(j) * b.n + i0))[l]

SBR_DEV void load_x(const SbrBuf& b, int64_t i0, uint32_t l, double (&x)[SBR_NX]) {
#pragma unroll
    for (int j = 0; j < SBR_NX; ++j) x[j] = XROW(j);
}
SBR_DEV void store_x(const SbrBuf& b, int64_t i0, uint32_t l, const double (&x)[SBR_NX]) {
#pragma unroll
    for (int j = 0; j < SBR_NX; ++j) st_out(&XROW(j), x[j]);
}
// INTERNAL controller layout [R_NROWS][N] (the public one of sbr_amd.h is produced by k_export / consumed by k_import):
//  * the Kla history is a RING: the value of the j-th interval since reset sits in slot (j-1) % 10, where the interval
//    count k is recovered from the running time, k = round((t - T_fill)/t_delta); a step writes ONE slot, not ten, and
//    (round 4) READS three: next to the ring the handle keeps Kla[-1] and the sum of the eight entries before it, from which
//    the reward's window follows incrementally (SbrHistInc in sbr_device.h);
//  * steps, the plan of the last interval (round 6), two flags, status bits and the done flag share one row (meta = steps*16384 +
//    plan*64 + m1i*32 + idle*16 + status*2 + done, an integer < 2^31 held in a double: steps saturate at 2^17 - 1 calls per
//    episode; plan = SBR_C_PLAN's code, 8 bits);
//  * So[-1] and Sno[-1] are IMPLICIT after an ordinary step (round 4): every interval ends with So.append(x_out[-1][8]),
//    Sno.append(x_out[-1][9]) (:1955-1956, :2043-2044), so after a step the two values ARE x[8] and x[9] and k_step neither
//    stores nor loads their rows (meta's m1i bit says so).  Whoever writes values that are NOT the plant's - the reset (Ss in
//    the Sno memory, :1652), an import, a rollout, the done call (the terminal phases move x afterwards) - stores the rows and
//    leaves the bit clear; the next step then fetches them in a second, dependent load (once per episode);
//  * rows only read when tauD != 0 (So[-2], Sno[-2]), only at the end of an episode (Qw) or only by the operating-cost
//    reward (the running sum of Kla) come last.
// Per env-step the step kernel reads 11 controller rows (t, two integrals, EC[-1], return, meta, Kla[-1], w8, three ring
// slots) and writes 11 (those minus the three slots, plus So[-2], Sno[-2] and one ring slot): 176 B, where the public layout
// would take 20 + 24 rows (352 B) and rounds 2-3 took 18 + 11 (232 B).
enum { R_T = 0, R_SO_M1, R_SNO_M1, R_IE_DO, R_IE_EC, R_EC_LAST, R_RET, R_META, R_KLA_LAST, R_W8, R_RING0,
       R_SO_M2 = R_RING0 + SBR_KLA_HIST, R_SNO_M2, R_QW, R_KSUM, R_NROWS };
#define SBR_MAX_STEPS ((1 << 17) - 1)

SBR_DEV int ring_k(const SbrPar& p, double t) {             // intervals since reset, from the running time
    double q = __builtin_fma(t - p.T_fill, p.inv_t_delta, 0.5);
    if (!(q >= 0.0)) q = 0.0;                                // also catches NaN
    if (q > 2e9) q = 2e9;
    return (int)q;
}
SBR_DEV int ring_wrap(int s) { return s >= SBR_KLA_HIST ? s - SBR_KLA_HIST : s; }       // for 0 <= s < 20
// meta = steps*16384 + plan*64 + m1i*32 + idle*16 + status*2 + done.  `plan`: what cfg.scheme = 1 did in the last control
// interval k_step ran for this env (step count + 128 if dissolved oxygen was held; 0 = none reported: scheme 0, a reset, an
// import, a rollout) - SBR_C_PLAN of the public layout; it rides in this row so that reporting it moves no extra byte.  `idle`: the done call of k_step appended one more Kla than t accounts
// for (Sim_idle's, :2578): the ring's oldest entry then sits one slot further than ring_k(t) says (k_export adds it; a reset,
// an import or a rollout store the ring in plain order and clear the bit).  `m1i`: So[-1], Sno[-1] are x[8], x[9], their rows
// are stale (set by an ordinary k_step call only; every other writer stores the rows and leaves it clear).
#define SBR_META_M1I 32
#define SBR_META_PLAN_SHIFT 6
#define SBR_META_STEPS_SHIFT 14
SBR_DEV double meta_pack(int steps, int status, bool done, bool idle = false, bool m1i = false, int plan = 0) {
    return (double)((steps << SBR_META_STEPS_SHIFT) + ((plan & 0xff) << SBR_META_PLAN_SHIFT) + (m1i ? SBR_META_M1I : 0) + (idle ? 16 : 0) +
                    status * 2 + (done ? 1 : 0));
}
SBR_DEV int meta_plan(double m) { return ((int)m >> SBR_META_PLAN_SHIFT) & 0xff; }
SBR_DEV void meta_unpack(double m, int& steps, int& status, bool& done, bool& idle, bool& m1i) {
    const int v = (int)m;
    done = (v & 1) != 0; status = (v >> 1) & 7; idle = (v & 16) != 0; m1i = (v & SBR_META_M1I) != 0; steps = v >> SBR_META_STEPS_SHIFT;
}
SBR_DEV void meta_unpack(double m, int& steps, int& status, bool& done, bool& idle) { bool m1i; meta_unpack(m, steps, status, done, idle, m1i); }
SBR_DEV void meta_unpack(double m, int& steps, int& status, bool& done) { bool idle; meta_unpack(m, steps, status, done, idle); }
// the two rows k_step keeps next to the ring (SbrHistInc), from a history in logical order (oldest first, hist[9] = Kla[-1])
SBR_DEV double hist_w8(const double (&hist)[SBR_KLA_HIST]) {
    double s = hist[1];
#pragma unroll
    for (int j = 2; j <= 8; ++j) s = s + hist[j];
    return s;
}

// Rows the step consumes BEFORE the integration.  So[-2], Sno[-2] only feed the derivative term (tauD != 0) and the
// trajectory export's dcv_EC: read only then (need_m2, wave-uniform).
// (k_step reads So[-1] / Sno[-1] from the plant when meta says they are implicit; here: the general form, m1i from meta)
SBR_DEV void load_ctl_pre(const SbrBuf& b, int64_t i0, uint32_t l, bool need_m2, bool m1i, const double (&x)[SBR_NX], SbrCtl& c) {
    c.t = CTRL(R_T);
    c.so_m1 = x[8]; c.sno_m1 = x[9];
    if (!m1i) { c.so_m1 = CTRL(R_SO_M1); c.sno_m1 = CTRL(R_SNO_M1); }
    c.ie_do = CTRL(R_IE_DO); c.ie_ec = CTRL(R_IE_EC); c.ec_last = CTRL(R_EC_LAST);
    c.so_m2 = need_m2 ? CTRL(R_SO_M2) : c.so_m1;
    c.sno_m2 = need_m2 ? CTRL(R_SNO_M2) : c.sno_m1;
    c.ec_prev = c.ec_last; c.u_do = 0.0; c.u_ec = 0.0;
    c.n_new = 0; c.st_new = 0; c.span = 0.0; c.rows = 9; c.plans = 0;
}
// rows every step rewrites (the ring slot(s), return and meta are written by the caller); with_m1 = false leaves So[-1] and
// Sno[-1] implicit (the caller then sets meta's m1i bit)
SBR_DEV void store_ctl(const SbrBuf& b, int64_t i0, uint32_t l, const SbrCtl& c, bool with_m1 = true) {
    st_out(&CTRL(R_T), c.t); st_out(&CTRL(R_SO_M2), c.so_m2); st_out(&CTRL(R_SNO_M2), c.sno_m2);
    if (with_m1) { st_out(&CTRL(R_SO_M1), c.so_m1); st_out(&CTRL(R_SNO_M1), c.sno_m1); }
    st_out(&CTRL(R_IE_DO), c.ie_do); st_out(&CTRL(R_IE_EC), c.ie_ec); st_out(&CTRL(R_EC_LAST), c.ec_last);
}
// whole history, logical order (oldest first), for the given interval count (per-lane slot: the general, slower form)
SBR_DEV void load_ring(const SbrBuf& b, int64_t i0, uint32_t l, int k, double (&hist)[SBR_KLA_HIST]) {
    const int kb = k % SBR_KLA_HIST;
#pragma unroll
    for (int j = 0; j < SBR_KLA_HIST; ++j) hist[j] = CTRL(R_RING0 + ring_wrap(kb + j));
}
SBR_DEV void store_ring(const SbrBuf& b, int64_t i0, uint32_t l, int k, const double (&hist)[SBR_KLA_HIST]) {
    const int kb = k % SBR_KLA_HIST;
#pragma unroll
    for (int j = 0; j < SBR_KLA_HIST; ++j) CTRL(R_RING0 + ring_wrap(kb + j)) = hist[j];
}

// public <-> internal translation (sbr_get_state / sbr_set_state / sbr_get_ctrl_row); only_row < 0 = all rows
__global__ __launch_bounds__(SBR_BLOCK) void k_export(SbrPar p, SbrBuf b, double* __restrict__ out, int only_row) {
    const uint32_t l = threadIdx.x;
    const int64_t i0 = (int64_t)blockIdx.x * SBR_BLOCK, i = i0 + l;
    if (i >= b.n) return;
    double v[SBR_NCTRL], hist[SBR_KLA_HIST];
    int steps, status; bool done, idle, m1i;
    const double t = CTRL(R_T);
    meta_unpack(CTRL(R_META), steps, status, done, idle, m1i);
    load_ring(b, i0, l, ring_k(p, t) + (idle ? 1 : 0), hist);
    v[SBR_C_T] = t; v[SBR_C_SO_M1] = m1i ? XROW(8) : CTRL(R_SO_M1); v[SBR_C_SO_M2] = CTRL(R_SO_M2);
    v[SBR_C_SNO_M1] = m1i ? XROW(9) : CTRL(R_SNO_M1); v[SBR_C_SNO_M2] = CTRL(R_SNO_M2);
    v[SBR_C_IE_DO] = CTRL(R_IE_DO); v[SBR_C_IE_EC] = CTRL(R_IE_EC); v[SBR_C_EC_LAST] = CTRL(R_EC_LAST);
#pragma unroll
    for (int j = 0; j < SBR_KLA_HIST; ++j) v[SBR_C_KLA_HIST0 + j] = hist[j];
    v[SBR_C_QW] = CTRL(R_QW); v[SBR_C_RETURN] = CTRL(R_RET); v[SBR_C_STEPS] = (double)steps;
    v[SBR_C_DONE] = done ? 1.0 : 0.0; v[SBR_C_STATUS] = (double)status; v[SBR_C_KLA_SUM] = CTRL(R_KSUM);
    v[SBR_C_PLAN] = (double)meta_plan(CTRL(R_META));
    if (only_row >= 0) {
#pragma unroll
        for (int r = 0; r < SBR_NCTRL; ++r) if (r == only_row) out[i] = v[r];
    } else {
#pragma unroll
        for (int r = 0; r < SBR_NCTRL; ++r) out[(int64_t)r * b.n + i] = v[r];
    }
}
// makes implicit So[-1] / Sno[-1] explicit (sbr_set_state with a plant only)
__global__ __launch_bounds__(SBR_BLOCK) void k_m1_explicit(SbrBuf b) {
    const uint32_t l = threadIdx.x;
    const int64_t i0 = (int64_t)blockIdx.x * SBR_BLOCK;
    if (i0 + l >= b.n) return;
    const int m = (int)CTRL(R_META);
    if ((m & SBR_META_M1I) != 0) {
        CTRL(R_SO_M1) = XROW(8); CTRL(R_SNO_M1) = XROW(9);
        CTRL(R_META) = (double)(m & ~SBR_META_M1I);
    }
}
__global__ __launch_bounds__(SBR_BLOCK) void k_import(SbrPar p, SbrBuf b, const double* __restrict__ in) {
    const uint32_t l = threadIdx.x;
    const int64_t i0 = (int64_t)blockIdx.x * SBR_BLOCK, i = i0 + l;
    if (i >= b.n) return;
#define IN(r) in[(int64_t)(r) * b.n + i]
    const double t = IN(SBR_C_T);
    double hist[SBR_KLA_HIST];
#pragma unroll
    for (int j = 0; j < SBR_KLA_HIST; ++j) hist[j] = IN(SBR_C_KLA_HIST0 + j);
    CTRL(R_T) = t; CTRL(R_SO_M1) = IN(SBR_C_SO_M1); CTRL(R_SO_M2) = IN(SBR_C_SO_M2);
    CTRL(R_SNO_M1) = IN(SBR_C_SNO_M1); CTRL(R_SNO_M2) = IN(SBR_C_SNO_M2);
    CTRL(R_IE_DO) = IN(SBR_C_IE_DO); CTRL(R_IE_EC) = IN(SBR_C_IE_EC); CTRL(R_EC_LAST) = IN(SBR_C_EC_LAST);
    store_ring(b, i0, l, ring_k(p, t), hist);
    CTRL(R_KLA_LAST) = hist[SBR_KLA_HIST - 1]; CTRL(R_W8) = hist_w8(hist);
    CTRL(R_QW) = IN(SBR_C_QW); CTRL(R_RET) = IN(SBR_C_RETURN); CTRL(R_KSUM) = IN(SBR_C_KLA_SUM);
    double st = IN(SBR_C_STEPS);
    st = st >= 0.0 ? (st < (double)SBR_MAX_STEPS ? st : (double)SBR_MAX_STEPS) : 0.0;
    double pl = IN(SBR_C_PLAN);
    pl = pl >= 0.0 ? (pl < 255.0 ? pl : 255.0) : 0.0;       // also catches NaN
    CTRL(R_META) = meta_pack((int)st, (int)IN(SBR_C_STATUS) & 7, IN(SBR_C_DONE) != 0.0, false, false, (int)pl);
#undef IN
}

// scenario of cfg.random_scenario = 1: uniform over the 8 scenarios (np.random.choice(8, 1), gym_SBR_env4.py:107), Philox
// stream 2 keyed by the seed of this reset, subsequence = global env id
SBR_DEV int sbr_scenario_draw(uint64_t seed, uint64_t gid) {
    uint32_t c[4] = {0u, 2u, (uint32_t)gid, (uint32_t)(gid >> 32)};
    sbr_philox(c, (uint32_t)seed, (uint32_t)(seed >> 32));
    return (int)(c[0] & (SBR_NSCEN - 1));
}

// copy the influent tables [2][8][14][48] from HBM into the padded LDS image (whole workgroup; caller synchronises)
template <int BLK = SBR_RESET_BLOCK>
SBR_DEV void stage_tables(double* lds, const double* __restrict__ tables) {
    for (int k = threadIdx.x; k < 2 * kTableDoubles; k += BLK)
        lds[k + 2 * (k / (SBR_NSERIES * SBR_NSAMP))] = tables[k];
}

// influent_mixed for one lane (buffer_tank3.py:68-107): series = mean + std*rnd with ONE rnd vector shared by all series,
// flow-weighted means, sums accumulated in sample order like python's sum().  ld[1..13]; ld[0] is set by the caller.
SBR_DEV void influent_lane(const double* lds, bool need_tables, int s, const double* __restrict__ rnd,
                           const double* __restrict__ influent, uint64_t seed, uint64_t gid, int64_t i, double (&ld)[SBR_NX]) {
    if (need_tables) {
        s = s < 0 ? 0 : (s >= SBR_NSCEN ? SBR_NSCEN - 1 : s);        // never index LDS out of range
        const double* mu = lds + s * SBR_TSTRIDE;
        const double* sd = lds + (SBR_NSCEN + s) * SBR_TSTRIDE;
        double acc[13], sq = 0.0;
#pragma unroll
        for (int j = 0; j < 13; ++j) acc[j] = 0.0;
        for (int kk = 0; kk < SBR_NSAMP; kk += 2) {
            double z[2];
            if (rnd) { z[0] = rnd[i * SBR_NSAMP + kk]; z[1] = rnd[i * SBR_NSAMP + kk + 1]; }
            else sbr_normal_pair(seed, gid, (uint32_t)(kk >> 1), z[0], z[1]);
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int k = kk + u;
                const double q = mu[13 * SBR_NSAMP + k] + sd[13 * SBR_NSAMP + k] * z[u];
                sq = sq + q;
#pragma unroll
                for (int j = 0; j < 13; ++j) acc[j] = acc[j] + (mu[j * SBR_NSAMP + k] + sd[j * SBR_NSAMP + k] * z[u]) * q;
            }
        }
        const double rsq = sbr_rcp(sq);                               // one reciprocal for the 13 flow-weighted means (1 ulp each)
#pragma unroll
        for (int j = 0; j < 13; ++j) ld[1 + j] = acc[j] * rsq;
    } else {
#pragma unroll
        for (int j = 1; j < SBR_NX; ++j) ld[j] = influent[i * SBR_NX + j];
    }
    ld[0] = 0.66;                                                    // buffer_tank3.py:107; callers overwrite it
}
// which scenario an env is reset with: the caller's tensor, else a device draw (cfg.random_scenario), else the fixed one
SBR_DEV int pick_scenario(const SbrPar& p, const int32_t* __restrict__ scenario, int fixed, uint64_t seed, uint64_t gid, int64_t i) {
    if (scenario) return scenario[i];
    return p.random_scenario ? sbr_scenario_draw(seed, gid) : fixed;
}

// ------------------------------------------------------------------------------------------- reset
// SbrOS.reset :168-438.  Influent tables (means, stds: 2 x 42 KiB) are staged in LDS once per
// workgroup; every lane then walks the 48 samples of ITS scenario (same scenario => LDS broadcast).
// BLK = 512 (launches with more wavefronts than the device has SIMDs): the 84 KiB of tables allow one workgroup per CU, so eight
// waves per workgroup are what puts two on a SIMD.
template <typename OutT, bool CARRY, int BLK = SBR_RESET_BLOCK>
__global__ __launch_bounds__(BLK) void k_reset(SbrPar p, SbrBuf b, const double* __restrict__ tables,
                                                    uint64_t seed, const int32_t* __restrict__ scenario,
                                                    const double* __restrict__ rnd, const double* __restrict__ influent,
                                                    const uint8_t* __restrict__ mask, OutT* __restrict__ obs) {
    extern __shared__ __attribute__((aligned(16))) double lds[];   // [2][8][SBR_TSTRIDE]
    const bool need_tables = (influent == nullptr);
    if (need_tables) {
        stage_tables<BLK>(lds, tables);
        __syncthreads();
    }
    const uint32_t l = threadIdx.x;
    const int64_t i0 = (int64_t)blockIdx.x * BLK, i = i0 + l;
    if (i >= b.n) return;
    if (mask != nullptr && mask[i] == 0) return;
    const uint64_t gid = (uint64_t)(b.first_env_id + i);

    double ld[SBR_NX];
    influent_lane(lds, need_tables, pick_scenario(p, scenario, 6 /* :180 */, seed, gid, i), rnd, influent, seed, gid, i, ld);

    // ---- start state: cfg.x0 / cfg.IV (:197-203), or with CARRY this env's own current state (x0_new / IV_new)
    double x[SBR_NX], x0[SBR_NX];
    double iv = p.IV, qin = p.qin;
    if (CARRY) {
        load_x(b, i0, l, x0);
        iv = x0[0]; qin = p.WV - iv;
        ld[0] = qin * p.inv_T_fill;
    } else {
#pragma unroll
        for (int j = 0; j < SBR_NX; ++j) {
            x0[j] = p.x0[j];
            // held in VGPRs from here on: read again from the argument registers after the fill loop (the observation's start
            // values, So[-2], Sno[-2]) they kept 17 SGPRs too many alive across it - spill slots, i.e. a scratch segment
            asm volatile("" : "+v"(x0[j]));
        }
        ld[0] = p.load0;                                             // :287
    }
#pragma unroll
    for (int j = 0; j < SBR_NX; ++j) { x[j] = x0[j]; INFL(j) = ld[j]; }

    // ---- fill phase, Sim_filling :1585-1654.  DO-PID at t_start == 0: ie = 0, dcv = 0, set-point 0
    SbrCtl c;
    double hist[SBR_KLA_HIST];
    const double e = 0.0 - x0[8];
    double ie = 0.0;
    double kla = p.Kc_DO * e + p.KcI_DO * ie;
    if (kla > p.Kla_max) { kla = p.Kla_max; ie = ie - e * p.dt; }
    if (kla < p.Kla_min) { kla = p.Kla_min; ie = ie - e * p.dt; }
    sbr_rk4<2>(p, x, p.h_fill, p.fill_rows, kla, ld[0], ld);        // RK4 under either scheme (see sbr_cycle_phase)
    c.t = p.T_fill;
    c.so_m2 = x0[8]; c.so_m1 = x[8];
    c.sno_m2 = x0[9]; c.sno_m1 = x[2];                               // :1652 stores Ss in the Sno memory
    c.ie_do = ie; c.ie_ec = 0.0; c.ec_last = 0.0; c.ec_prev = 0.0;
    c.u_do = 0.0; c.u_ec = 15.0;                                     // :212-213
#pragma unroll
    for (int j = 0; j < SBR_KLA_HIST; ++j) hist[j] = ((SBR_KLA_HIST - 1 - j) % 2 == 0) ? kla : 0.0;   // [0,k]*126, :323
    c.kla_last = kla;
    store_x(b, i0, l, x);
    store_ctl(b, i0, l, c);
    store_ring(b, i0, l, ring_k(p, c.t), hist);      // k = 0: logical order = slot order
    CTRL(R_KLA_LAST) = hist[SBR_KLA_HIST - 1]; CTRL(R_W8) = hist_w8(hist);
    CTRL(R_RET) = 0.0; CTRL(R_META) = meta_pack(0, sbr_status_bits(p, x), false); CTRL(R_QW) = 0.0;
    double ksum = 0.0;                               // python's sum() over the list [0, k]*126, left to right
    for (int j = 0; j < p.fill_rows / 2; ++j) ksum = ksum + kla;
    CTRL(R_KSUM) = ksum;
    if (obs) {   // volume blend of influent and post-fill state, :346-361
        double xr[SBR_NX];
        const double rwv = sbr_rcp(qin + iv);
#pragma unroll
        for (int j = 0; j < SBR_NX; ++j) xr[j] = (qin * ld[j] + x[j] * iv) * rwv;
        double x06[SBR_NXD];
        sbr_take6(x0, x06);
        sbr_write_obs<OutT>(obs + i * SBR_NOBS, 1, c.t, xr, x06, x);
    }
}

// ------------------------------------------------------------------------------------------- step
// All loads of a lane are issued up front, with addresses that depend on nothing loaded (they are asynchronous; the first
// use waits): ONE exposed memory round trip per launch.  Measured alternatives that were WORSE (profiles/r01_notes.md):
// loading the Kla history and the bookkeeping rows after the integration (+2.4 us: serial round trips).
// The template parameter BLK is the workgroup size.  256 threads (four waves, one per SIMD of a CU; the kernel comes out at
// ~250 VGPRs - tests/test_isa_cpu.py holds it to <= 256 - i.e. two waves per SIMD, which batches above 65536 envs use to overlap the memory phases of one wave with the
// arithmetic of the other and to issue FMAs at 4.43 instead of 5.19 cycles) is the fastest from ~32768 envs up; below that
// 64-thread workgroups win - one wave per workgroup spreads a batch over four times as many CUs, each with its own path to
// memory and its own LDS (profiles/r02_ab_block.log: 1024 .. 16384 envs 12.45 -> 11.6 us per launch, 32768: 12.65 -> 12.1,
// 49152: 13.0 -> 12.8; 65536: 13.85 against 14.03, so 256 from there on).
// Values that only have to SURVIVE the integration (the Kla ring, return, packed steps/status/done, the six xdot start
// values) are parked in LDS, not in VGPRs and not in scratch: a ~100-cycle round trip instead of a trip through L2/HBM, and
// the RK4 loop keeps its registers (keeping them in VGPRs was measured: +0.6 us).  Slot j of lane l of wave w is at
// park[(w*NSLOT + j)*64 + l] (conflict-free).
// LDS slots of a lane in k_step: return, meta, the Kla window sum w8, the six xdot start values (and the OCI running sum);
// the region is sized for the output transposes (64 float64 observation rows: 9216 B = 18 slots).
#define SBR_PK_RET 0
#define SBR_PK_META 1
#define SBR_PK_W8 2
#define SBR_PK_X6 3
#define SBR_PK_KSUM (SBR_PK_X6 + SBR_NXD)
#define SBR_NPARK 20
#define SBR_NPARK_2W 32     // k_step<.., WAVES = 2>: + what the call keeps across the integration (SbrX6LdsT<true>)

// One output row per lane (obs: 18 values, state: 15) -> the caller's row-major tensor.  A full wavefront owns 64
// consecutive rows, i.e. ONE contiguous block of 64 x NV x sizeof(OutT) bytes (4608 B of float32 observations): the rows go
// through the wave's own LDS region and leave as 16-byte-per-lane stores (obs + state: 9 store instructions per lane).
// Written directly, every lane stores NV separate dwords at a 72- or 60-byte stride: 33 store instructions, each touching
// 36 cache lines; with all 1024 waves in their epilogue at once those stores queue behind each other (measured with the
// stamp build: 2.8 us of a 17.6 us launch).  Waves that are not full (last wave of a ragged batch) and misaligned
// destinations take the direct form.  LDS operations of one wave execute in issue order, so the region can be reused
// without a workgroup barrier; the wave barrier keeps the compiler from reordering across it.
template <typename OutT, int NV>
SBR_DEV void store_rows(OutT* __restrict__ rows /* out + i0*NV: first row of the workgroup */, uint32_t l, char* stage,
                        bool wide, const OutT (&v)[NV]) {
    constexpr int RB = NV * (int)sizeof(OutT);                 // bytes per row
    constexpr int CH = 64 * RB / 16;                           // 16-byte chunks per wave
    const uint32_t lane = l & 63u;
    char* wdst = reinterpret_cast<char*>(rows + (size_t)(l & ~63u) * NV);
    if (wide && (reinterpret_cast<uintptr_t>(wdst) & 15u) == 0) {
        asm volatile("" ::: "memory");                         // see store_rows2: the staging stores must not overtake the parked loads
        typename SbrAliasOf<OutT>::type* mine = reinterpret_cast<typename SbrAliasOf<OutT>::type*>(stage + lane * RB);
#pragma unroll
        for (int k = 0; k < NV; ++k) mine[k] = v[k];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        constexpr int NR = (CH + 63) / 64;
        sbr_u32x4 ch[NR];                                      // every chunk into registers first, then the stores (see store_rows2)
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            const int c = r * 64 + (int)lane;
            if ((r + 1) * 64 <= CH || c < CH) ch[r] = *reinterpret_cast<const sbr_u32x4_alias*>(stage + c * 16);
        }
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            const int c = r * 64 + (int)lane;
            if ((r + 1) * 64 <= CH || c < CH) st_out16(reinterpret_cast<sbr_u32x4*>(wdst + c * 16), ch[r]);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    } else {
#pragma unroll
        for (int k = 0; k < NV; ++k) rows[(size_t)l * NV + k] = v[k];
    }
}

// Two row sets at once (observation rows first, state rows behind them in the staging region): stage both, ONE wave barrier,
// read every 16-byte chunk into registers, then issue the stores.  The stores are inline assembly with a memory clobber (the
// sc1 modifier has no builtin for 16 bytes), which the compiler will not move an LDS read across: read-store-read-store cost
// one exposed LDS round trip per chunk (nine per call until round 4).  `stage` must hold 64 (RB_A + RB_B) bytes.
template <typename OutT, int NA, int NB>
SBR_DEV void store_rows2(OutT* __restrict__ rows_a, OutT* __restrict__ rows_b, uint32_t l, char* stage, bool wide,
                         const OutT (&va)[NA], const OutT (&vb)[NB]) {
    constexpr int RA = NA * (int)sizeof(OutT), RBB = NB * (int)sizeof(OutT);
    constexpr int CA = 64 * RA / 16, CB = 64 * RBB / 16;         // 16-byte chunks per wave
    constexpr int NRA = (CA + 63) / 64, NRB = (CB + 63) / 64;
    const uint32_t lane = l & 63u;
    char* wa = reinterpret_cast<char*>(rows_a + (size_t)(l & ~63u) * NA);
    char* wb = reinterpret_cast<char*>(rows_b + (size_t)(l & ~63u) * NB);
    if (wide && ((reinterpret_cast<uintptr_t>(wa) | reinterpret_cast<uintptr_t>(wb)) & 15u) == 0) {
        // The staging region is the wave's parking space, whose slots were read as float64 just before: to the compiler's
        // type-based alias analysis a float32 store and a float64 load never alias, so without this barrier it may hoist the
        // staging stores above those loads (seen in round 4: rows 27.. of the float32 outputs held parked values).
        asm volatile("" ::: "memory");
        char* sa = stage;
        char* sb = stage + 64 * RA;
        typename SbrAliasOf<OutT>::type* ma = reinterpret_cast<typename SbrAliasOf<OutT>::type*>(sa + lane * RA);
        typename SbrAliasOf<OutT>::type* mb = reinterpret_cast<typename SbrAliasOf<OutT>::type*>(sb + lane * RBB);
#pragma unroll
        for (int k = 0; k < NA; ++k) ma[k] = va[k];
#pragma unroll
        for (int k = 0; k < NB; ++k) mb[k] = vb[k];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        sbr_u32x4 ca[NRA], cb[NRB];
#pragma unroll
        for (int r = 0; r < NRA; ++r) {
            const int c = r * 64 + (int)lane;
            if ((r + 1) * 64 <= CA || c < CA) ca[r] = *reinterpret_cast<const sbr_u32x4_alias*>(sa + c * 16);
        }
#pragma unroll
        for (int r = 0; r < NRB; ++r) {
            const int c = r * 64 + (int)lane;
            if ((r + 1) * 64 <= CB || c < CB) cb[r] = *reinterpret_cast<const sbr_u32x4_alias*>(sb + c * 16);
        }
#pragma unroll
        for (int r = 0; r < NRA; ++r) {
            const int c = r * 64 + (int)lane;
            if ((r + 1) * 64 <= CA || c < CA) st_out16(reinterpret_cast<sbr_u32x4*>(wa + c * 16), ca[r]);
        }
#pragma unroll
        for (int r = 0; r < NRB; ++r) {
            const int c = r * 64 + (int)lane;
            if ((r + 1) * 64 <= CB || c < CB) st_out16(reinterpret_cast<sbr_u32x4*>(wb + c * 16), cb[r]);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    } else {
#pragma unroll
        for (int k = 0; k < NA; ++k) rows_a[(size_t)l * NA + k] = va[k];
#pragma unroll
        for (int k = 0; k < NB; ++k) rows_b[(size_t)l * NB + k] = vb[k];
    }
}

// The model constants (SbrPar, 1.2 KB) travel in the kernel-argument segment and the compiler fetches them with scalar
// loads where it needs them - in pieces, because ~60 doubles do not fit the scalar register file at once.  The argument
// segment of a launch is a fresh buffer in device memory that no cache has seen: the first touch of each of its 64-byte lines
// is a miss in the scalar cache that goes out to L2 / memory, and every such piece sits behind its own s_waitcnt ON the
// wave's critical path (PMC, round 4: SQ_WAIT_ANY is what grew when an unrelated code change added two late scalar loads,
// +0.3 us per launch).  So every line of the segment is touched ONCE, all at the same time, at wave start: the misses
// overlap each other and the ~2 us of global-load latency, and every later scalar load of a constant hits the scalar cache.
// The loads target one scratch SGPR whose value is never used.
#ifndef SBR_KERNARG_WARM
#define SBR_KERNARG_WARM 1     // 0 switches the warm-up off (A/B builds only; measured: profiles/r04_ab_kernarg_warm_modes.log)
#endif
#define SBR_WARM_LINES                                                                                                            \
        "s_load_dword %0, %1, 0x40\n s_load_dword %0, %1, 0x80\n s_load_dword %0, %1, 0xc0\n s_load_dword %0, %1, 0x100\n"        \
        "s_load_dword %0, %1, 0x140\n s_load_dword %0, %1, 0x180\n s_load_dword %0, %1, 0x1c0\n s_load_dword %0, %1, 0x200\n"     \
        "s_load_dword %0, %1, 0x240\n s_load_dword %0, %1, 0x280\n s_load_dword %0, %1, 0x2c0\n s_load_dword %0, %1, 0x300\n"     \
        "s_load_dword %0, %1, 0x340\n s_load_dword %0, %1, 0x380\n s_load_dword %0, %1, 0x3c0\n s_load_dword %0, %1, 0x400\n"     \
        "s_load_dword %0, %1, 0x440\n s_load_dword %0, %1, 0x480\n s_load_dword %0, %1, 0x4c0\n s_load_dword %0, %1, 0x500\n"     \
        "s_load_dword %0, %1, 0x538\n s_load_dword %0, %1, 0x548\n"
// k_step's argument segment: four pointers / sizes, the flags word (+ padding), four pointers, SbrPar, SbrBuf.  The touched
// offsets must stay inside it (a scalar load past the segment may fault) and reach its last line.
static constexpr size_t kStepKernargBytes = 72 + sizeof(SbrPar) + sizeof(SbrBuf);
static_assert(kStepKernargBytes >= 0x548 + 4 && kStepKernargBytes <= 0x548 + 56,
              "SbrPar / SbrBuf changed size: adjust the offsets of SBR_WARM_LINES to cover k_step's argument segment");
// In two halves, issue at wave start and wait after the wave's global loads have gone out: the scratch register stays allocated
// (an in/out operand of the second statement) until the loads have landed, so the compiler cannot hand it to anything else
// while they are in flight.  (One statement placed after the global loads was measured too: +0.2 us per launch.)
SBR_DEV uint32_t sbr_warm_kernarg_issue() {
    auto kp = __builtin_amdgcn_kernarg_segment_ptr();
    uint32_t t;
    asm volatile(SBR_WARM_LINES : "=&s"(t) : "s"(kp) : "memory");
    return t;
}
SBR_DEV void sbr_warm_kernarg_wait(uint32_t t) { asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(t) : : "memory"); }

// The leading arguments (14 dwords) are what the first global loads need; the library is built with
// -mllvm -amdgpu-kernarg-preload-count=16, so a wave starts with them in SGPRs and issues its loads without waiting for a
// scalar load of the argument segment (two serial scalar round trips before: n for the bounds test, then the pointers).
// `flags` carries every decision that shapes the LOAD phase (bit 0: the So[-2] / Sno[-2] rows are needed - derivative action
// or a trace buffer), taken on the host: until round 4 that test read two fields of `p` and one of `b0`, and the s_waitcnt
// in front of it held back EVERY global load of the wave for a scalar round trip to the argument segment.
#define SBR_KF_NEED_M2 1u
#define SBR_KF_STAGGER 2u
#ifndef SBR_STAGGER_SLEEP
#define SBR_STAGGER_SLEEP 25        // x 64 cycles
#endif
#ifndef SBR_STAGGER_MIN_ENVS
#define SBR_STAGGER_MIN_ENVS 0      // every launch in 256-thread workgroups (> SBR_SMALL_BATCH envs); A/B builds move the window
#endif
#ifndef SBR_STAGGER_MAX_ENVS
#define SBR_STAGGER_MAX_ENVS (1ll << 62)
#endif
// trajectory export: the NO3-PID's e / ie / dcv of every interval go straight to the call's trace record while the PID runs
// (slot _FIRST for the first interval of the call, the plain slots for the last one run), so that nothing has to be carried
// across the integration for it.  Off (b.trace == NULL, the default) this is one untaken scalar branch per interval.
struct SbrTraceRec {
    const SbrBuf& b;
    const double* meta_lds;       // the lane's parked steps/status/done word
    int64_t env;                  // index of the lane's env in the handle
    SBR_DEV void pid(int iv, double e, double ie, double dcv) const {
        if (__builtin_expect(b.trace != nullptr, 0)) {
            const int64_t steps = (int64_t)((int)(*meta_lds) >> SBR_META_STEPS_SHIFT);      // meta = steps*64 + flags
            if (env < b.n_trace && steps < b.trace_cap) {
                double* rec = b.trace + (steps * SBR_NTRACE) * b.n_trace + env;
                if (iv == 0) {
                    rec[SBR_TR_E_EC_FIRST * b.n_trace] = e; rec[SBR_TR_IE_EC_FIRST * b.n_trace] = ie; rec[SBR_TR_DCV_EC_FIRST * b.n_trace] = dcv;
                }
                rec[SBR_TR_E_EC * b.n_trace] = e; rec[SBR_TR_IE_EC * b.n_trace] = ie; rec[SBR_TR_DCV_EC * b.n_trace] = dcv;
            }
        }
    }
};

#ifndef SBR_STEP_MIN_BLOCKS
#define SBR_STEP_MIN_BLOCKS 1      // A/B builds: 2 caps k_step<.., 256, ..> at 256 registers (two waves per SIMD)
#endif
// WAVES = 2 (scheme 1, launches with more wavefronts than the chip has SIMDs): the register budget of two resident waves per SIMD
// (256).  The scheme-1 step loops need 320 registers with everything the call carries across them; this build parks that - 13
// controller values per interval, 7 of the call, 19 around the idle phase of the done call - in the lane's LDS slots (64 KiB per
// workgroup, two workgroups per CU) and forms the row addresses of its stores again afterwards instead of keeping them.
// Same arithmetic, same bits.  Measured (profiles/r05_ab_two_waves.log): 65 536 envs 12.8 against 12.2 us per call (one wave per
// SIMD either way: the host keeps WAVES = 1 there), 131 072 envs 20.1 against 23.3, 262 144 envs 36.1 against 44.5.
template <typename OutT, typename ActT, int BLK, bool OCI, int SCH, int WAVES = 1>
__global__ __launch_bounds__(BLK, BLK == 256 ? (WAVES == 2 ? 2 : SBR_STEP_MIN_BLOCKS) : 1) void k_step(double* __restrict__ bx, double* __restrict__ bctrl, int64_t bn,
                                                      const ActT* __restrict__ action, uint32_t flags, OutT* __restrict__ obs,
                                                      OutT* __restrict__ state, OutT* __restrict__ reward,
                                                      uint8_t* __restrict__ done, SbrPar p, SbrBuf b0) {
    SbrBuf b = b0;
    b.x = bx; b.ctrl = bctrl; b.n = bn;
    // wave-major: wave w owns park[w][slot][64], 20 slots x 512 B = 10 KiB; the region is reused for the output
    // transpose (64 float64 observation rows take 9216 B, float32 observation + state rows together 8448 B)
    constexpr bool PARK = WAVES == 2;
    constexpr int NSLOT = PARK ? SBR_NPARK_2W : SBR_NPARK;
    // parked values: slots SBR_PK_X6 + SBR_NXD + 1 + j, j < 20 (13 per interval + 7 of the call; 19 around the terminal phases)
    static_assert(!PARK || SBR_NPARK_2W >= SBR_PK_X6 + SBR_NXD + 1 + 20, "the LDS region of the two-waves build is too small for its parked values");
    // ... and its upper bounds: static LDS is limited to 64 KiB per workgroup, and the two-waves build only pays while
    // TWO of its workgroups fit the 160 KiB of a CU - one more __shared__ word would silently halve the residency
    static_assert(sizeof(double) * NSLOT * BLK <= 65536, "k_step's parking region exceeds the 64 KiB static LDS limit");
    static_assert(!PARK || 2 * sizeof(double) * NSLOT * BLK <= 160 * 1024, "two workgroups of the two-waves build must fit a CU's 160 KiB of LDS");
    __shared__ __attribute__((aligned(16))) double park[NSLOT * BLK];
    uint32_t l = threadIdx.x;
    const int64_t i0 = (int64_t)blockIdx.x * BLK;
    if (i0 + l >= b.n) return;
#ifdef SBR_STAMPS
    if (reinterpret_cast<uintptr_t>(b.stamps) == 1) return;   // diagnostic build: the launch period of an EMPTY k_step = the boundary
#endif
    double* wave_lds = park + (l >> 6) * (NSLOT * 64);
    double* my = wave_lds + (l & 63u);        // slot j of this lane: my[j * 64]
    double x[SBR_NX];
    SbrCtl c;
    SBR_STAMP(0, false);
#if SBR_KERNARG_WARM
    const uint32_t warm_token = sbr_warm_kernarg_issue();
#endif
    // Staggered entry (SBR_KF_STAGGER, large batches): every other workgroup of an XCD (workgroup g runs on XCD g % 8, so bit 3
    // of the index alternates inside one) waits ~0.65 us before its loads.  With one wave per SIMD every wave of the chip
    // otherwise loads at the same moment and stores at the same moment; half a microsecond of skew inside each XCD takes the
    // two bursts apart (measured: profiles/r04_notes.md, "staggered entry").
    if ((flags & SBR_KF_STAGGER) != 0u && ((blockIdx.x >> 3) & 1u) != 0u) __builtin_amdgcn_s_sleep(SBR_STAGGER_SLEEP);
    // every load below has an address that depends on nothing loaded: ONE memory round trip (the ring used to be read
    // in logical order, whose rows depend on t: a second, dependent round trip)
    load_x(b, i0, l, x);
    const double meta0 = CTRL(R_META);
    c.t = CTRL(R_T); c.ie_do = CTRL(R_IE_DO); c.ie_ec = CTRL(R_IE_EC); c.ec_last = CTRL(R_EC_LAST);
    c.ec_prev = c.ec_last; c.u_do = 0.0; c.u_ec = 0.0;
    c.n_new = 0; c.st_new = 0; c.span = 0.0; c.rows = 9; c.plans = 0;
    if ((flags & SBR_KF_NEED_M2) != 0u) { c.so_m2 = CTRL(R_SO_M2); c.sno_m2 = CTRL(R_SNO_M2); }
    const ActT* act = action + i0 * 2;
    const double a0 = (double)act[2 * l], a1 = (double)act[2 * l + 1];           // one 8- or 16-byte load per lane
    c.kla_last = CTRL(R_KLA_LAST);
    const double w8_0 = CTRL(R_W8), ret0 = CTRL(R_RET);
#if SBR_KERNARG_WARM
    sbr_warm_kernarg_wait(warm_token);
#endif
    my[SBR_PK_RET * 64] = ret0; my[SBR_PK_META * 64] = meta0; my[SBR_PK_W8 * 64] = w8_0;
    // (The compiler sinks the action load into the `not done` branch below - one global_load behind the branch in the ISA.
    // Pinning it into the batch above was measured SLOWER, profiles/r04_ab_kernarg_warm_and_pinned_loads.log: left alone.)
    SbrX6LdsT<PARK> x6{my + SBR_PK_X6 * 64};
    if (OCI) my[SBR_PK_KSUM * 64] = CTRL(R_KSUM);          // only this reward keeps the running sum of Kla
    x6.put(x);
    // the ring slot of the oldest entry = where this call's first Kla goes; the three entries behind it are the ones that leave
    // the reward's window with this call's appends (SbrHistInc): loads whose address depends on t, issued now, needed after the
    // integration
    const int kb = ring_k(p, c.t) % SBR_KLA_HIST;
    double lv[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) lv[j] = CTRL(R_RING0 + ring_wrap(kb + 1 + j));
    const double kla_before = c.kla_last;
    SBR_STAMP(1, true);                       // every load has returned, the parked values are in LDS
    double t_obs = p.t_cycle, r = 0.0;
    bool dn = true;
    double xa6[SBR_NXD];
    if (((int)meta0 & 1) == 0) {          // not done: a finished env waits for sbr_reset (the reference leaves resetting to the caller)
        double qw = 0.0;
        int steps, status; bool was_done;
        SbrRewardParts rp;
        const double v0 = x[0], si0 = x[1], xi0 = x[3];
        // So[-1], Sno[-1]: the plant's own values after an ordinary step (meta's m1i bit); the first call after a reset, an import
        // or a rollout fetches the rows (a second, dependent load: once per episode)
        c.so_m1 = x[8]; c.sno_m1 = x[9];
        const bool m1_rows = ((int)meta0 & SBR_META_M1I) == 0;
        if (__builtin_amdgcn_ballot_w64(m1_rows) != 0ull) {
            if (m1_rows) { c.so_m1 = CTRL(R_SO_M1); c.sno_m1 = CTRL(R_SNO_M1); }
        }
        if ((flags & SBR_KF_NEED_M2) == 0u) { c.so_m2 = c.so_m1; c.sno_m2 = c.sno_m1; }
        SBR_STAMP(2, false);
        const SbrTraceRec tr{b, my + SBR_PK_META * 64, i0 + l};
#ifndef SBR_STEP_LOOP
#define SBR_STEP_LOOP false     // straight-line: the second interval of a phase-boundary call out of line (247 VGPRs; the loop form needs 278 with the dependent ring loads live across the integration)
#endif
        if constexpr (PARK) {
            x6.park(13, lv[0]); x6.park(14, lv[1]); x6.park(15, lv[2]); x6.park(16, kla_before); x6.park(17, v0); x6.park(18, si0);
            x6.park(19, xi0);
        }
        sbr_run_intervals<SBR_STEP_LOOP, SCH>(p, c, x, a0, a1, x6, tr);
        double kla_before_r = kla_before, v0_r = v0, si0_r = si0, xi0_r = xi0;
        if constexpr (PARK) {
            asm volatile("" : "+v"(l) : : "memory");  // the row addresses of the stores are formed again from here (not kept across the integration)
            lv[0] = x6.unpark(13); lv[1] = x6.unpark(14); lv[2] = x6.unpark(15);
            kla_before_r = x6.unpark(16); v0_r = x6.unpark(17); si0_r = x6.unpark(18); xi0_r = x6.unpark(19);
        }
        SBR_STAMP(3, false);                  // PIDs + RK4 done
        SbrHistInc hs{my[SBR_PK_W8 * 64], kla_before_r, {lv[0], lv[1], lv[2]}, 0.0, false};
        if constexpr (!PARK) x6.get(xa6);
        double ksum = OCI ? my[SBR_PK_KSUM * 64] : 0.0;
        r = sbr_finish_step<OCI, SCH, SbrHistInc, true, SbrX6LdsT<PARK>>(p, c, hs, x, xa6, t_obs, dn, qw, ksum, rp, &x6);
        if constexpr (PARK) x6.get(xa6);              // after the terminal phases, which leave their pre-settle values in the slots
        SBR_STAMP(4, false);                  // reward (and, on the done call, the terminal phases) done
        // everything that reads the three ring entries loaded before the integration comes BEFORE the first store: the memory
        // counter retires in order, so a wait for one of those (long finished) loads placed after the plant stores would wait
        // for the stores' acknowledgements too
        double w8_new, last_new;
        hs.roll(c, w8_new, last_new);
        asm volatile("" : "+v"(w8_new), "+v"(last_new) : : "memory");     // ... and the scheduler is held to that order
        SBR_STAMP(8, false);                  // window rolled: everything that waits for a load is done
        if (OCI) CTRL(R_KSUM) = ksum;
        // plant: V, Si and Xi only change with carbon dosing or in the terminal phases - skip their stores otherwise
        // (wave-uniform test: no lane of the wave changed them)
        const bool inert_moved = (x[0] != v0_r) || (x[1] != si0_r) || (x[3] != xi0_r);
        if (__builtin_amdgcn_ballot_w64(inert_moved) != 0ull) {
            store_x(b, i0, l, x);
        } else {
#pragma unroll
            for (int j = 0; j < SBR_NX; ++j) if (j != 0 && j != 1 && j != 3) st_out(&XROW(j), x[j]);
        }
        SBR_STAMP(9, false);                  // plant stores issued
        // So[-1], Sno[-1] stay implicit unless this was the done call (the terminal phases have moved x away from them)
        store_ctl(b, i0, l, c, false);
        if (__builtin_amdgcn_ballot_w64(dn) != 0ull) {
            if (dn) { CTRL(R_SO_M1) = c.so_m1; CTRL(R_SNO_M1) = c.sno_m1; }
        }
        // the Kla ring is addressed by the interval count: envs reset together share it, so the slot is normally
        // wave-uniform (scalar row arithmetic); a wave whose lanes disagree (masked resets, injected states) takes the
        // per-lane form.  A second (phase-boundary call) and a third append (the idle phase of the done call) are rare.
        const int kb_u = __builtin_amdgcn_readfirstlane(kb);
        const bool ring_uniform = __builtin_amdgcn_ballot_w64(kb != kb_u) == 0ull;
        const bool idle_pushed = hs.idle_pushed;
        const double app0 = c.n_new > 0 ? c.knew[0] : hs.idle;            // the appended values in order: knew[0], knew[1], idle
        const int n_app = c.n_new + (idle_pushed ? 1 : 0);
        if (ring_uniform) {
            if (n_app > 0) st_out(&CTRL(R_RING0 + kb_u), app0);
        } else {
            if (n_app > 0) CTRL(R_RING0 + kb) = app0;
        }
        if (__builtin_amdgcn_ballot_w64(n_app > 1) != 0ull) {
            const double app1 = c.n_new > 1 ? c.knew[1] : hs.idle;
            if (n_app > 1) CTRL(R_RING0 + ring_wrap(kb + 1)) = app1;
            if (n_app > 2) CTRL(R_RING0 + ring_wrap(kb + 2)) = hs.idle;
        }
        st_out(&CTRL(R_KLA_LAST), last_new); st_out(&CTRL(R_W8), w8_new);
        if (dn && p.terminal) CTRL(R_QW) = qw;
        meta_unpack(my[SBR_PK_META * 64], steps, status, was_done);
        st_out(&CTRL(R_RET), my[SBR_PK_RET * 64] + r);
        // the m1i bit ("So[-1] / Sno[-1] are x[8] / x[9]") is claimed only by a call that RAN an interval: a call with
        // n_new == 0 (t injected as NaN) writes nothing to the memories, so it leaves them where they were - rows stay rows
        const bool m1i_new = dn ? false : (c.n_new > 0 ? true : !m1_rows);
        st_out(&CTRL(R_META), meta_pack(steps < SBR_MAX_STEPS ? steps + 1 : steps, status | c.st_new, dn, idle_pushed, m1i_new, c.plans & 0xff));
        SBR_STAMP(10, false);                 // controller stores issued
        if (b.trace != nullptr && i0 + l < b.n_trace && (int64_t)steps < b.trace_cap) {     // trajectory export, off by default
            double* rec = b.trace + ((int64_t)steps * SBR_NTRACE) * b.n_trace + (i0 + l);
            rec[0] = c.t;
#pragma unroll
            for (int j = 0; j < SBR_NX; ++j) rec[(int64_t)(SBR_TR_X0 + j) * b.n_trace] = x[j];
            rec[SBR_TR_KLA * b.n_trace] = c.knew[c.n_new > 1 ? 1 : 0]; rec[SBR_TR_EC * b.n_trace] = c.ec_last;
            rec[SBR_TR_REWARD * b.n_trace] = r; rec[SBR_TR_DONE * b.n_trace] = dn ? 1.0 : 0.0;
            rec[SBR_TR_U_DO * b.n_trace] = c.u_do; rec[SBR_TR_U_EC * b.n_trace] = c.u_ec;
            // (e_EC, ie_EC, dcv_EC of the interval(s): written by SbrTraceRec::pid as the PIDs ran)
            // module_reward_EQIOCI.py:60-112: EQI2, and the cost terms over their maxima (Kla = 240, EC = 0.0005 throughout)
            const double ae2 = rp.ae * p.inv_ae_max, ec2 = rp.ec * p.inv_ec_max;
            rec[SBR_TR_R_EQI * b.n_trace] = rp.eqi2; rec[SBR_TR_R_OCI * b.n_trace] = ae2 + ec2;
            rec[SBR_TR_R_AE * b.n_trace] = ae2; rec[SBR_TR_R_EC * b.n_trace] = ec2;
            // what rebuilding the sub-interval rows needs (sbr_eval_substeps): the intervals run and the first one's Kla / EC
            rec[SBR_TR_N_IV * b.n_trace] = (double)c.n_new; rec[SBR_TR_KLA_FIRST * b.n_trace] = c.knew[0];
            rec[SBR_TR_EC_FIRST * b.n_trace] = c.n_new > 1 ? c.ec_prev : c.ec_last;
            // what cfg.scheme = 1 did (round 6): plan code of the call's last interval and of its first (equal when one ran)
            rec[SBR_TR_PLAN * b.n_trace] = (double)(c.plans & 0xff); rec[SBR_TR_PLAN_FIRST * b.n_trace] = (double)((c.plans >> 8) & 0xff);
        }
    } else {
        x6.get(xa6);
    }
    SBR_STAMP(5, false);                      // state stores issued
    if (reward) st_out(&(reward + i0)[l], (OutT)r);
    if (done) st_out(&(done + i0)[l], (uint8_t)(dn ? 1 : 0));
    SBR_STAMP(11, false);                     // reward / done stores issued
    const bool wide = __builtin_amdgcn_ballot_w64(true) == ~0ull;      // all 64 lanes of the wave hold an env
    char* stage = reinterpret_cast<char*>(wave_lds);
    // both row sets fit the wave's staging region together when they are float32 (64 x (72 + 60) = 8448 of the region's 10240 bytes: 20 slots x 512)
    constexpr bool kBoth = 64 * (SBR_NOBS + SBR_NSTATE) * (int)sizeof(OutT) <= NSLOT * 64 * (int)sizeof(double);
    if (kBoth && obs && state) {
        OutT o[SBR_NOBS], sv[SBR_NSTATE];
        sbr_write_obs<OutT>(o, 1, t_obs, x, xa6, x);
        sbr_write_state<OutT>(sv, 1, t_obs, x);
#ifdef SBR_STAMPS
        asm volatile("" : "+v"(o[0]), "+v"(o[17]), "+v"(sv[14]) : : "memory");
#endif
        SBR_STAMP(12, false);                 // output values formed
        store_rows2<OutT, SBR_NOBS, SBR_NSTATE>(obs + i0 * SBR_NOBS, state + i0 * SBR_NSTATE, l, stage, wide, o, sv);
    } else {
        if (obs) {
            OutT o[SBR_NOBS];
            sbr_write_obs<OutT>(o, 1, t_obs, x, xa6, x);
            store_rows<OutT, SBR_NOBS>(obs + i0 * SBR_NOBS, l, stage, wide, o);
        }
        if (state) {
            OutT sv[SBR_NSTATE];
            sbr_write_state<OutT>(sv, 1, t_obs, x);
            store_rows<OutT, SBR_NSTATE>(state + i0 * SBR_NSTATE, l, stage, wide, sv);
        }
    }
    SBR_STAMP(6, false);                      // output stores issued
    SBR_STAMP(7, true);                       // every store acknowledged
}

// ------------------------------------------------------------------------------------------- rollout
// WAVES: resident waves per SIMD the register budget is cut for.  Scheme 1 keeps more state live (five 9-vectors and the step
// constants of the Butcher-5 steps): capped at 256 registers for two waves per SIMD it spills ~170 B per lane to scratch, which
// pays only where two waves ARE resident - launches above 98 304 envs; up to there the uncapped build runs (measured, round 5:
// 65 536 envs 6.6 against 7.3 us per call, 131 072 envs 11.6 against 10.8).
template <bool OCI, int SCH, int WAVES>
__global__ __launch_bounds__(SBR_BLOCK, WAVES) void k_rollout(SbrPar p, SbrBuf b, int32_t n_steps, uint64_t policy_seed,
                                                      double* __restrict__ returns, float* __restrict__ actions_out) {
    const uint32_t l = threadIdx.x;
    const int64_t i0 = (int64_t)blockIdx.x * SBR_BLOCK, i = i0 + l;
    if (i >= b.n) return;
    const uint64_t gid = (uint64_t)(b.first_env_id + i);
    double x[SBR_NX], xa6[SBR_NXD], hist[SBR_KLA_HIST];
    SbrCtl c;
    // x6 and the ten Kla values stay in registers.  With the terminal phases INSIDE the loop over the calls the kernel needed
    // 268 VGPRs and, capped at 256 for two waves per SIMD, spilled six loop invariants to 52 B of scratch per lane; parking x6 or
    // the history in LDS removed the spill and was 1 - 2 % slower.  Running settle / draw / idle once AFTER the loop (a finished
    // lane skips every later call, so nothing changes in between) gives 243 VGPRs, no scratch and 2 - 3 % less time per call
    // (profiles/r03_ab_rollout_scratch.log, r03_ab_rollout_terminal_after_loop.log).
    SbrX6Reg x6;
    SbrRewardParts rp;
    load_x(b, i0, l, x);
    int steps, status; bool finished, idle_bit, m1i;
    meta_unpack(CTRL(R_META), steps, status, finished, idle_bit, m1i);
    load_ctl_pre(b, i0, l, true, m1i, x, c);
    load_ring(b, i0, l, ring_k(p, c.t) + (idle_bit ? 1 : 0), hist);
    c.kla_last = hist[SBR_KLA_HIST - 1];
    double ret = CTRL(R_RET), qw = CTRL(R_QW), ksum = OCI ? CTRL(R_KSUM) : 0.0;
    double acc = 0.0;
    bool terminal_due = false;        // the done call happened in this launch: its settle / draw / idle run once, after the loop
    for (int32_t s = 0; s < n_steps; ++s) {
        float a0, a1;
        sbr_policy_action(p, policy_seed, gid, (uint32_t)steps, a0, a1);
        if (actions_out) reinterpret_cast<float2*>(actions_out)[(int64_t)s * b.n + i] = make_float2(a0, a1);
        if (finished) continue;
        double t_obs;
        bool dn;
        sbr_run_intervals<false, SCH>(p, c, x, (double)a0, (double)a1, x6, SbrNoTrace{});
        x6.get(xa6);
        SbrHistReg hs{hist};
        const double r = sbr_finish_step<OCI, SCH, SbrHistReg, false>(p, c, hs, x, xa6, t_obs, dn, qw, ksum, rp);
        acc += r; ret += r; status |= c.st_new;
        if (steps < SBR_MAX_STEPS) steps += 1;
        if (dn) { finished = true; terminal_due = !OCI && p.terminal; }
    }
    if (terminal_due) {               // a finished lane skipped every later call, so c, x and hist are as the done call left them
        SbrHistReg hs{hist};
        qw = sbr_terminal<SCH>(p, c, hs, x);
    }
    store_x(b, i0, l, x);
    store_ctl(b, i0, l, c);
    // the ring is addressed by the interval count recovered from t: store the logical history consistently with the final t
    // (the idle phase's extra Kla does not advance t; k_export reads with the same rule)
    store_ring(b, i0, l, ring_k(p, c.t), hist);
    CTRL(R_KLA_LAST) = hist[SBR_KLA_HIST - 1]; CTRL(R_W8) = hist_w8(hist);
    CTRL(R_RET) = ret; CTRL(R_META) = meta_pack(steps, status, finished); CTRL(R_QW) = qw;
    if (OCI) CTRL(R_KSUM) = ksum;                     // like k_step: only this reward maintains the row
    if (returns) returns[i] = acc;
}
// ------------------------------------------------------------------------------------------- per-cycle env (SBR-v2)
// SbrEnv2.reset (gym_SBR_env2.py:69-129): influent draw (scenario 0 by default, :104) and the 3-element observation built
// from the sums of start state and influent.  CARRY keeps each env's current state as the start state (x0_new, :152).
template <typename OutT, bool CARRY>
__global__ __launch_bounds__(SBR_RESET_BLOCK) void k_cycle_reset(SbrPar p, SbrBuf b, const double* __restrict__ tables,
                                                                uint64_t seed, const int32_t* __restrict__ scenario,
                                                                const double* __restrict__ rnd, const double* __restrict__ influent,
                                                                const uint8_t* __restrict__ mask, OutT* __restrict__ obs) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const bool need_tables = (influent == nullptr);
    if (need_tables) {
        stage_tables(lds, tables);
        __syncthreads();
    }
    const uint32_t l = threadIdx.x;
    const int64_t i0 = (int64_t)blockIdx.x * SBR_RESET_BLOCK, i = i0 + l;
    if (i >= b.n) return;
    if (mask != nullptr && mask[i] == 0) return;
    double ld[SBR_NX], x0[SBR_NX];
    const uint64_t gid = (uint64_t)(b.first_env_id + i);
    influent_lane(lds, need_tables, pick_scenario(p, scenario, 0 /* :104 */, seed, gid, i), rnd, influent, seed, gid, i, ld);
    if (CARRY) load_x(b, i0, l, x0);
    else {
#pragma unroll
        for (int j = 0; j < SBR_NX; ++j) x0[j] = p.x0[j];
        store_x(b, i0, l, x0);
    }
#pragma unroll
    for (int j = 0; j < SBR_NX; ++j) INFL(j) = ld[j];
    CTRL(R_T) = 0.0; CTRL(R_RET) = 0.0; CTRL(R_QW) = 0.0;
    CTRL(R_META) = meta_pack(0, sbr_status_bits(p, x0), true);       // inert for sbr_step: this handle runs whole cycles
    if (obs) {
        const double cod = (x0[1] + ld[1]) + (x0[2] + ld[2]) + (x0[3] + ld[3]) + (x0[4] + ld[4]) + (x0[5] + ld[5]) +
                           (x0[6] + ld[6]) + (x0[7] + ld[7]);
        obs[i * 3 + 0] = (OutT)(x0[0] + ld[0]); obs[i * 3 + 1] = (OutT)((cod - 5145) / 10); obs[i * 3 + 2] = (OutT)((x0[10] + ld[10]) / 30);
    }
}

// SbrEnv2.step: one whole 12 h cycle per env (528 control intervals; scheme 1: adaptive Butcher-5 steps on all but the 24 fill
// intervals, scheme 0: x 10 RK4 substeps) in one launch.
template <typename OutT, typename ActT, int SCH, int WAVES>
__global__ __launch_bounds__(SBR_BLOCK, WAVES) void k_cycle(SbrPar p, SbrBuf b, const ActT* __restrict__ action, OutT* __restrict__ obs,
                                                    OutT* __restrict__ reward, double* __restrict__ diag) {
    const uint32_t l = threadIdx.x;
    const int64_t i0 = (int64_t)blockIdx.x * SBR_BLOCK, i = i0 + l;
    if (i >= b.n) return;
    double x[SBR_NX], ld[SBR_NX], o3[3];
    load_x(b, i0, l, x);
#pragma unroll
    for (int j = 0; j < SBR_NX; ++j) ld[j] = INFL(j);
    ld[0] = (p.WV - x[0]) * p.inv_t_ph0;                              // Qin / (t_cycle * t_ratio[0]), gym_SBR_env2.py:144 (host reciprocal: 1 ulp)
    const int st0 = sbr_status_bits(p, x);
    const double r = sbr_cycle_env<SCH>(p, x, ld, (double)action[3 * i], (double)action[3 * i + 1], (double)action[3 * i + 2], o3,
                                   diag ? diag + i * SBR_NCYC_DIAG : nullptr, 1);
    store_x(b, i0, l, x);
    CTRL(R_RET) = CTRL(R_RET) + r; CTRL(R_T) = p.t_cycle;
    CTRL(R_META) = meta_pack(1, st0 | sbr_status_bits(p, x), true);
    if (obs) { obs[i * 3 + 0] = (OutT)o3[0]; obs[i * 3 + 1] = (OutT)o3[1]; obs[i * 3 + 2] = (OutT)o3[2]; }
    if (reward) reward[i] = (OutT)r;
}

// the scenario draw of cfg.random_scenario, for tests and callers that want to know it
__global__ __launch_bounds__(SBR_BLOCK) void k_scenarios(SbrBuf b, uint64_t seed, int32_t* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * SBR_BLOCK + threadIdx.x;
    if (i < b.n) out[i] = sbr_scenario_draw(seed, (uint64_t)(b.first_env_id + i));
}

#undef CTRL
#undef XROW
#undef INFL

// ------------------------------------------------------------------------------------------- stats
// {sum, min, max, count} of a per-env vector: butterfly over the 64 lanes of each wave (DPP/swizzle via
// __shfl_xor), then one atomic per wave.
SBR_DEV double atomic_min_f64(double* addr, double v) {
    unsigned long long* a = reinterpret_cast<unsigned long long*>(addr);
    unsigned long long old = *a, assumed;
    do {
        assumed = old;
        if (__longlong_as_double((long long)assumed) <= v) break;
        old = atomicCAS(a, assumed, (unsigned long long)__double_as_longlong(v));
    } while (assumed != old);
    return __longlong_as_double((long long)old);
}
SBR_DEV double atomic_max_f64(double* addr, double v) {
    unsigned long long* a = reinterpret_cast<unsigned long long*>(addr);
    unsigned long long old = *a, assumed;
    do {
        assumed = old;
        if (__longlong_as_double((long long)assumed) >= v) break;
        old = atomicCAS(a, assumed, (unsigned long long)__double_as_longlong(v));
    } while (assumed != old);
    return __longlong_as_double((long long)old);
}

__global__ __launch_bounds__(SBR_BLOCK) void k_stats_init(double* out4) {
    if (threadIdx.x == 0) { out4[0] = 0.0; out4[1] = INFINITY; out4[2] = -INFINITY; out4[3] = 0.0; }
}

__global__ __launch_bounds__(SBR_BLOCK) void k_stats(const double* __restrict__ v, int64_t n, double* out4) {
    double s = 0.0, mn = INFINITY, mx = -INFINITY, cnt = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * SBR_BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * SBR_BLOCK) {
        const double a = v[i];
        s += a; mn = fmin(mn, a); mx = fmax(mx, a); cnt += 1.0;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        s += __shfl_xor(s, off, 64);
        mn = fmin(mn, __shfl_xor(mn, off, 64));
        mx = fmax(mx, __shfl_xor(mx, off, 64));
        cnt += __shfl_xor(cnt, off, 64);
    }
    if ((threadIdx.x & 63) == 0 && cnt > 0.0) {      // lane 0 of EVERY wave of the workgroup publishes its wave's partials
        atomicAdd(out4 + 0, s);
        atomic_min_f64(out4 + 1, mn);
        atomic_max_f64(out4 + 2, mx);
        atomicAdd(out4 + 3, cnt);
    }
}

// ------------------------------------------------------------------------------------------- helpers
__global__ __launch_bounds__(SBR_BLOCK) void k_fill(double* __restrict__ dst, int64_t n, double v) {
    const int64_t i = (int64_t)blockIdx.x * SBR_BLOCK + threadIdx.x;
    if (i < n) dst[i] = v;
}

__global__ __launch_bounds__(SBR_BLOCK) void k_rhs(SbrPar p, int32_t kind, int64_t n, const double* __restrict__ x,
                                                  const double* __restrict__ kla, const double* __restrict__ ec,
                                                  const double* __restrict__ loading, double* __restrict__ dx) {
    const int64_t i = (int64_t)blockIdx.x * SBR_BLOCK + threadIdx.x;
    if (i >= n) return;
    double xv[SBR_NX], ld[SBR_NX], d[SBR_NX];
#pragma unroll
    for (int j = 0; j < SBR_NX; ++j) { xv[j] = x[i * SBR_NX + j]; ld[j] = loading ? loading[i * SBR_NX + j] : 0.0; }
    if (kind == 0) sbr_rhs<0>(p, xv, kla[i], ec[i], ld, d);
    else if (kind == 1) sbr_rhs<1>(p, xv, kla[i], 0.0, ld, d);
    else sbr_rhs<2>(p, xv, kla[i], 0.0, ld, d);
#pragma unroll
    for (int j = 0; j < SBR_NX; ++j) dx[i * SBR_NX + j] = d[j];
}

// RK4 nodes of one integration span and the right-hand side at each node (sbr_eval_substeps; trajectory export only).
// kind 0: a control interval (reaction_dxdt with Kla and EC held) - substep by substep with the dosing code path, whose flow
// terms are exact no-ops at ec == 0 (sbr_rk4), so the nodes are those sbr_step passes through, to rounding (V, Si, Xi and the
// charge balance are closed per substep here, per interval there); kind 1: the fill phase (filling_dxdt, loading vector);
// kind 2: the idle phase (idle_dxdt); kind 3: settle + draw applied to x0 first (Sim_Settling_Drawing), then the idle phase -
// node 0 is then the reactor after the draw.
__global__ __launch_bounds__(SBR_BLOCK) void k_substeps(SbrPar p, int32_t kind, int64_t n, int32_t n_sub, const double* __restrict__ x0,
                                                       const double* __restrict__ kla, const double* __restrict__ ec,
                                                       const double* __restrict__ loading, const double* __restrict__ hstep,
                                                       double* __restrict__ xs, double* __restrict__ dxs) {
    const int64_t i = (int64_t)blockIdx.x * SBR_BLOCK + threadIdx.x;
    if (i >= n) return;
    double x[SBR_NX], ld[SBR_NX], d[SBR_NX];
#pragma unroll
    for (int j = 0; j < SBR_NX; ++j) { x[j] = x0[i * SBR_NX + j]; ld[j] = (kind == 1) ? loading[i * SBR_NX + j] : 0.0; }
    if (kind == 3) {
        double sx[10], sx_eff;
        const double xf = sbr_settle(p, x, p.t_settle * p.t_cycle, sx);
        (void)sbr_draw(p, x, sx, xf, sx_eff);
    }
    const double h = hstep[i], k = kla[i], e = (kind == 0) ? ec[i] : 0.0;
    for (int s = 0; s <= n_sub; ++s) {
        if (kind == 0) sbr_rhs<0>(p, x, k, e, ld, d);
        else if (kind == 1) sbr_rhs<1>(p, x, k, 0.0, ld, d);
        else sbr_rhs<2>(p, x, k, 0.0, ld, d);
        const int64_t o = (i * ((int64_t)n_sub + 1) + s) * SBR_NX;
#pragma unroll
        for (int j = 0; j < SBR_NX; ++j) { xs[o + j] = x[j]; dxs[o + j] = d[j]; }
        if (s < n_sub) {
            if (kind == 0) sbr_rk4<1>(p, x, h, 1, k, e, ld);
            else if (kind == 1) sbr_rk4<2>(p, x, h, 1, k, ld[0], ld);
            else sbr_rk4<0>(p, x, h, 1, k, 0.0, ld);
        }
    }
}

__global__ __launch_bounds__(SBR_BLOCK) void k_normals(SbrBuf b, uint64_t seed, double* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * SBR_BLOCK + threadIdx.x;
    if (i >= b.n) return;
    for (uint32_t k = 0; k < SBR_NSAMP / 2; ++k) {
        double z0, z1;
        sbr_normal_pair(seed, (uint64_t)(b.first_env_id + i), k, z0, z1);
        out[i * SBR_NSAMP + 2 * k] = z0;
        out[i * SBR_NSAMP + 2 * k + 1] = z1;
    }
}

// =========================================================================================== host / C ABI
struct sbr_env {
    int64_t n = 0;
    int device = 0;
    int64_t first_env_id = 0;
    sbr_config cfg;
    SbrPar par;
    SbrBuf buf{};
    double* tables = nullptr;     // [2][8][14][48] on the device
    int64_t one_wave_envs = 65536; // lanes of one wave per SIMD on this device: CUs x 4 SIMDs x 64 (MI355X: 256 CUs)
    bool have_tables = false;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    std::string err;
};

static thread_local std::string g_create_err;      // creation errors are reported per calling thread (sbr_last_error(NULL))

static int fail(sbr_env* e, int code, const std::string& msg) {
    if (e) e->err = msg; else g_create_err = msg;
    return code;
}
#define HIP_TRY(e, call)                                                                              \
    do {                                                                                              \
        hipError_t _s = (call);                                                                       \
        if (_s != hipSuccess)                                                                         \
            return fail(e, SBR_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(_s));           \
    } while (0)

static inline dim3 grid_for(int64_t n) { return dim3((unsigned)((n + SBR_BLOCK - 1) / SBR_BLOCK)); }

// Entry points run on the handle's device and leave the caller's current device as they found it (a process may drive
// several GPUs, and torch tracks the current device itself).
struct DeviceGuard {
    int prev = -1;
    hipError_t status = hipSuccess;
    explicit DeviceGuard(int device) {
        int cur = -1;
        status = hipGetDevice(&cur);
        if (status == hipSuccess && cur != device) { status = hipSetDevice(device); prev = cur; }
    }
    ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
};
#define ON_DEVICE(e)                      \
    DeviceGuard _guard((e)->device);      \
    HIP_TRY(e, _guard.status)

// smallest double s with (int)(s / dt) >= rows, i.e. whose IEEE quotient by dt reaches rows: start at rows*dt and step
// through the neighbouring doubles (the quotient is monotonic in s; a handful of steps at most)
static double rows_threshold(double dt, int rows) {
    double s = (double)rows * dt;
    for (int it = 0; it < 64 && (int)(s / dt) < rows; ++it) s = std::nextafter(s, INFINITY);
    for (int it = 0; it < 64; ++it) {
        const double below = std::nextafter(s, 0.0);
        if ((int)(below / dt) < rows) break;
        s = below;
    }
    return s;
}

static void derive_params(const sbr_config& c, SbrPar& p) {
    p.muH = c.muH; p.Ks = c.Ks; p.Koh = c.Koh; p.Kno = c.Kno; p.bH = c.bH; p.eta_g = c.eta_g; p.eta_h = c.eta_h;
    p.kh = c.kh; p.Kx = c.Kx; p.muA = c.muA; p.Knh = c.Knh; p.bA = c.bA; p.Koa = c.Koa; p.ka = c.ka;
    const double Yh = c.Yh, Ya = c.Ya, ixb = c.ixb, ixp = c.ixp, fp = c.fp;
    p.n2_12 = -1 / Yh; p.n4_45 = 1 - ixp; p.n7_45 = ixp;
    p.n8_1 = -(1 - Yh) / Yh; p.n8_3 = -(4.57 - Ya) / Ya;
    p.n9_2 = -((1 - Yh) / (2.86 * Yh)); p.n9_3 = 1 / Ya;
    p.n10_12 = -ixb; p.n10_3 = -ixb - 1 / Ya;
    p.n12_45 = ixb - fp * ixp;
    p.n13_1 = -ixb / 14; p.n13_2 = (1 - Yh) / (14 * 2.86 * Yh) - ixb / 14; p.n13_3 = -ixb / 14 - 1 / (7 * Ya);
    p.n13_6 = 1.0 / 14;
    p.WV = c.WV; p.IV = c.IV; p.dt = c.dt; p.t_delta = c.t_delta; p.t_cycle = c.t_cycle;
    p.T_fill = c.T_fill; p.T3_0 = c.T3_0; p.T3_end = c.T3_end; p.T4_end = c.T4_end; p.T5_end = c.T5_end;
    p.t_settle = c.t_settle; p.t_draw = c.t_draw;
    p.qin = c.WV - c.IV; p.load0 = p.qin / c.T_fill;
    p.So_sat = c.So_sat; p.Kla_min = c.Kla_min; p.Kla_max = c.Kla_max;
    p.Kc_DO = c.Kc_DO; p.KcI_DO = c.Kc_DO / c.tauI_DO; p.KcD_DO = c.Kc_DO * c.tauD_DO;
    p.EC_min = c.EC_min; p.EC_max = c.EC_max;
    p.Kc_EC = c.Kc_EC; p.KcI_EC = c.Kc_EC / c.tauI_EC; p.KcD_EC = c.Kc_EC * c.tauD_EC; p.EC_conc = c.EC_conc;
    p.act_DO_max = c.act_DO_max; p.act_EC_max = c.act_EC_max;
    p.biomass_setpoint = c.biomass_setpoint; p.Qeff = c.Qeff; p.settler_area = c.settler_area;
    p.settler_vmax = c.settler_vmax;
    memcpy(p.x0, c.x0, sizeof p.x0);
    p.f1a = c.kh / c.muH; p.f1b = c.Ks * (c.kh / c.muH);
    p.f2a = 1.0 / c.kh; p.f2b = c.Koh / c.kh;
    p.f4a = 1.0 / c.muA; p.f4b = c.Knh / c.muA;
    p.f3a = 1.0 / (c.Koh * c.eta_g); p.f3b = c.Kno / (c.Koh * c.eta_g);
    p.KohEtag = c.Koh * c.eta_g; p.etah_g = c.eta_h / c.eta_g;
    p.bA_bH = c.bA / c.bH; p.n4_45b = p.n4_45 * c.bH; p.n12_45b = p.n12_45 * c.bH; p.n7_45b = p.n7_45 * c.bH;
    p.n9_23 = p.n9_2 / p.n9_3; p.inv_n9_3 = 1.0 / p.n9_3;
    for (int k = 0; k < 8; ++k) p.t_ph[k] = c.t_cycle * c.t_ratio[k];
    p.cyc_Kc = c.cyc_Kc; p.cyc_KcI = c.cyc_Kc / c.cyc_tauI; p.cyc_KcD = c.cyc_Kc * c.cyc_tauD; p.cyc_dt = c.cyc_dt;
    p.substeps = c.substeps; p.terminal = c.terminal; p.scheme = c.scheme;
    p.inv_Koh = 1.0 / c.Koh; p.inv_Koa = 1.0 / c.Koa;
    p.fill_rows = (int)((c.T_fill - 0) / c.dt);      // int((t_end - t_start)/dt) = 252, :1588
    p.reward_kind = c.reward_kind;
    p.random_scenario = c.random_scenario;
    p.inv_dt = 1.0 / c.dt; p.inv_t_delta = 1.0 / c.t_delta; p.inv_substeps = 1.0 / (double)c.substeps;
    p.inv_cyc_dt = 1.0 / c.cyc_dt; p.h_fill = c.T_fill / (double)p.fill_rows;
    {   // module_reward_EQIOCI.py:72, :80 - Kla = 240 and EC = 0.0005 over eleven rows of its own t_delta = 0.002/24, So_sat = 8
        const double td = 0.002 / 24;
        p.inv_ae_max = 1.0 / (1.32 * (240 * 11) * td * (8 / ((td * 11) * 1.8 * 1000)));
        p.inv_ec_max = 1.0 / (c.EC_conc * (0.0005 * 11) * td / ((td * 11) * 1000));
    }
    p.rows10_min = rows_threshold(c.dt, 10); p.rows9_min = rows_threshold(c.dt, 9);
    {   // SBR-v2 phase schedule, with the reference's own operations (SBR_model_FB.py:30-258: t_start = t_end + t_delta,
        // t_end = t_start + t_phs; sub_phases_FB.py:183-184: n2 = int((t_end - t_start)/(t_delta*10)), numpy.linspace's step)
        const double t_delta = 0.002 / 24;                         // gym_SBR_env2.py:34
        double t_start = 0.0, t_end = 0.0 + p.t_ph[0];
        int slot = 0;
        for (int ph = 0; ph < 8; ++ph) {
            if (ph > 0) { t_start = t_end + t_delta; t_end = t_start + p.t_ph[ph]; }
            if (ph == 5) { p.cyc_tset = t_end - t_start; continue; }      // settle
            if (ph == 6) continue;                                         // draw
            int n2 = (int)((t_end - t_start) / (t_delta * 10));
            n2 = n2 < 2 ? 2 : (n2 > 100000 ? 100000 : n2);                 // every wave terminates
            p.cyc_t0[slot] = t_start; p.cyc_t1[slot] = t_end; p.cyc_n2[slot] = n2;
            p.cyc_step[slot] = (t_end - t_start) / (double)(n2 - 1);
            p.cyc_inv_ntd[slot] = 1.0 / ((double)(n2 - 1) * t_delta);      // module_reward.py: .../(len(Kla) * t_delta)
            p.cyc_inv_n[slot] = 1.0 / (double)(n2 - 1);
            ++slot;
        }
        p.inv_t_ph0 = 1.0 / p.t_ph[0];
        p.sosat_k = c.So_sat / (1.8 * 1000);
        p.inv_T_fill = 1.0 / c.T_fill;
    }
}

// up to 1.5 waves per SIMD (MI355X: 98304 envs) the fused scheme-1 kernels run their uncapped-register build
#ifdef SBR_ONE_WAVE_MAX_ENVS
#define SBR_FUSED_ONE_WAVE_ENVS(e) ((int64_t)(SBR_ONE_WAVE_MAX_ENVS))
#else
#define SBR_FUSED_ONE_WAVE_ENVS(e) ((e)->one_wave_envs + (e)->one_wave_envs / 2)
#endif
// A launch with more wavefronts than the device has SIMDs (MI355X: 1024 SIMDs x 64 lanes = 65536 envs; a partitioned device has
// fewer) runs the two-waves-per-SIMD build of the scheme-1 k_step.  A/B builds fix the threshold with -DSBR_STEP_ONE_WAVE_MAX_ENVS=n.
#ifdef SBR_STEP_ONE_WAVE_MAX_ENVS
#define SBR_STEP_ONE_WAVE_ENVS(e) ((int64_t)(SBR_STEP_ONE_WAVE_MAX_ENVS))
#else
#define SBR_STEP_ONE_WAVE_ENVS(e) ((e)->one_wave_envs)
#endif
#ifndef SBR_SMALL_BATCH
#define SBR_SMALL_BATCH 49152       // up to this many envs k_step runs in 64-thread workgroups (measured: profiles/r02_ab_block.log)
#endif
template <typename OutT, typename ActT, bool OCI, int SCH>
static void launch_step_k(sbr_env* e, const void* action, void* obs, void* state, void* reward, uint8_t* done,
                          hipStream_t st) {
    const uint32_t flags = (e->par.KcD_DO != 0.0 || e->par.KcD_EC != 0.0 || e->buf.trace != nullptr) ? SBR_KF_NEED_M2 : 0u;
    if (e->n <= SBR_SMALL_BATCH)
        hipLaunchKernelGGL((k_step<OutT, ActT, 64, OCI, SCH>), dim3((unsigned)((e->n + 63) / 64)), dim3(64), 0, st, e->buf.x, e->buf.ctrl,
                           e->buf.n, (const ActT*)action, flags, (OutT*)obs, (OutT*)state, (OutT*)reward, done, e->par, e->buf);
    else if (SCH == 1 && e->n > SBR_STEP_ONE_WAVE_ENVS(e))
        hipLaunchKernelGGL((k_step<OutT, ActT, 256, OCI, SCH, SCH == 1 ? 2 : 1>), dim3((unsigned)((e->n + 255) / 256)), dim3(256), 0, st,
                           e->buf.x, e->buf.ctrl, e->buf.n, (const ActT*)action,
                           flags | ((e->n >= SBR_STAGGER_MIN_ENVS && e->n <= SBR_STAGGER_MAX_ENVS) ? SBR_KF_STAGGER : 0u), (OutT*)obs,
                           (OutT*)state, (OutT*)reward, done, e->par, e->buf);
    else
        hipLaunchKernelGGL((k_step<OutT, ActT, 256, OCI, SCH>), dim3((unsigned)((e->n + 255) / 256)), dim3(256), 0, st, e->buf.x,
                           e->buf.ctrl, e->buf.n, (const ActT*)action,
                           flags | ((e->n >= SBR_STAGGER_MIN_ENVS && e->n <= SBR_STAGGER_MAX_ENVS) ? SBR_KF_STAGGER : 0u), (OutT*)obs,
                           (OutT*)state, (OutT*)reward, done, e->par, e->buf);
}
template <typename OutT, typename ActT>
static void launch_step(sbr_env* e, const void* action, void* obs, void* state, void* reward, uint8_t* done,
                        hipStream_t st) {
    const bool oci = e->cfg.reward_kind == 2, b5 = e->cfg.scheme == 1;      // one instantiation per reward family and scheme
    if (oci) { if (b5) launch_step_k<OutT, ActT, true, 1>(e, action, obs, state, reward, done, st); else launch_step_k<OutT, ActT, true, 0>(e, action, obs, state, reward, done, st); }
    else { if (b5) launch_step_k<OutT, ActT, false, 1>(e, action, obs, state, reward, done, st); else launch_step_k<OutT, ActT, false, 0>(e, action, obs, state, reward, done, st); }
}

extern "C" {

const char* sbr_version(void) { return "sbr_amd 0.6.0 (gfx950, fp64; scheme 1 = adaptive Butcher-5, scheme 0 = RK4)"; }
int sbr_abi_version(void) { return SBR_ABI_VERSION; }

int sbr_default_config(sbr_config* c) {
    if (!c) return SBR_ERR_INVALID;
    memset(c, 0, sizeof *c);
    // SURVEY.md Appendix A; values asserted against tests/golden/constants.npz by tests/test_capi_cpu.py
    c->Ya = 0.24; c->Yh = 0.67; c->fp = 0.08; c->ixb = 0.08; c->ixp = 0.06;
    c->muH = 4.0; c->Ks = 10.0; c->Koh = 0.2; c->Kno = 0.5; c->bH = 0.3; c->eta_g = 0.8; c->eta_h = 0.8;
    c->kh = 3.0; c->Kx = 0.1; c->muA = 0.5; c->Knh = 1.0; c->bA = 0.05; c->Koa = 0.4; c->ka = 0.05;
    c->WV = 1.32; c->IV = 0.6161484733495801; c->dt = 0.002 / 24; c->t_delta = c->dt * 10; c->t_cycle = 12.0 / 24;
    c->T_fill = 0.021; c->T3_0 = 0.06416666666666668; c->T3_end = 0.2516666666666667;
    c->T4_end = 0.4085000000000001; c->T5_end = 0.40933333333333344;
    c->t_settle = 8.3 / 100; c->t_draw = 2.1 / 100;
    c->So_sat = 8.000000000006622; c->Kla_min = 0; c->Kla_max = 240; c->Kc_DO = 100; c->tauI_DO = 20; c->tauD_DO = 0;
    c->EC_min = 0; c->EC_max = 0.0005; c->Kc_EC = 100; c->tauI_EC = 20; c->tauD_EC = 0; c->EC_conc = 1200000 * 4.0;
    c->act_DO_max = 8; c->act_EC_max = 15;
    c->biomass_setpoint = 2700; c->Qeff = 0.66; c->settler_area = (1.25 / 2) * (1.25 / 2); c->settler_vmax = 474;
    static const double tr[8] = {4.2 / 100, 8.3 / 100, 37.5 / 100, 31.2 / 100, 2.1 / 100, 8.3 / 100, 2.1 / 100, 6.3 / 100};
    memcpy(c->t_ratio, tr, sizeof tr);
    c->cyc_Kc = 5.0; c->cyc_tauI = 0.00035; c->cyc_tauD = 0.005; c->cyc_dt = 0.02 / 24;      // gym_SBR_env2.py:48
    static const double x0[SBR_NX] = {0.6161484733495801, 30, 0.571098000538576, 1440.01157895393, 31.254221999137,
                                      2599.2714348941, 168.915006750837, 551.901552960823, 2.16607843793004,
                                      13.3791460027604, 0.00562880208518134, 0.35996687629947, 1.86916737961228,
                                      3.790463057094611};
    memcpy(c->x0, x0, sizeof x0);
    c->substeps = 10; c->out_f64 = 0; c->terminal = 1; c->reward_kind = 0; c->act_f64 = 0; c->random_scenario = 0;
    c->scheme = 1; c->reserved_ = 0;
    return SBR_OK;
}

int sbr_rows_thresholds(const sbr_config* cfg, double* out2) {
    if (!out2) return SBR_ERR_INVALID;
    sbr_config c;
    if (cfg) c = *cfg; else sbr_default_config(&c);
    if (!(c.dt > 0)) return SBR_ERR_INVALID;
    out2[0] = rows_threshold(c.dt, 9); out2[1] = rows_threshold(c.dt, 10);
    return SBR_OK;
}

int sbr_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return n;
}

int sbr_create(int64_t n_envs, int device_id, int64_t first_env_id, const sbr_config* cfg, sbr_env** out) {
    if (!out) return fail(nullptr, SBR_ERR_INVALID, "sbr_create: out is NULL");
    *out = nullptr;
    if (n_envs <= 0) return fail(nullptr, SBR_ERR_INVALID, "sbr_create: n_envs must be > 0");
    int ndev = sbr_device_count();
    if (ndev <= 0)
        return fail(nullptr, SBR_ERR_NO_DEVICE, "sbr_create: no HIP device visible - this library has no CPU path");
    if (device_id < 0 || device_id >= ndev) return fail(nullptr, SBR_ERR_INVALID, "sbr_create: bad device_id");
    sbr_env* e = new (std::nothrow) sbr_env();
    if (!e) return fail(nullptr, SBR_ERR_ALLOC, "sbr_create: out of host memory");
    e->n = n_envs; e->device = device_id; e->first_env_id = first_env_id;
    if (cfg) e->cfg = *cfg; else sbr_default_config(&e->cfg);
    const sbr_config& c = e->cfg;
    std::string bad;
    if (c.substeps < 1 || c.substeps > 10000) bad = "substeps out of range";
    if (c.scheme < 0 || c.scheme > 1) bad = "scheme must be 0 (RK4 x substeps) or 1 (adaptive Butcher-5)";
    if (c.reserved_ != 0) bad = "reserved_ must be 0 (a caller that leaves it uninitialised could not be told from one using a later meaning)";
    if (!(c.Koa > 0 && c.Koa < 1e300)) bad = "Koa must be positive";
    if (c.reward_kind < 0 || c.reward_kind > 2) bad = "reward_kind must be 0 (EQI/OCI), 1 (G2ANET) or 2 (operating cost)";
    if (!(c.dt > 0) || !(c.t_delta > 0)) bad = "dt and t_delta must be positive";
    else {
        const int rows = (int)(c.t_delta / c.dt + 0.5);
        if (rows != 10) bad = "t_delta must be 10*dt (the reward's Kla look-back is 9 intervals)";
    }
    if (!(c.T_fill > 0) || (int)(c.T_fill / c.dt) < 1) bad = "T_fill/dt must be >= 1";
    // every reaction phase must be longer than one control interval, so that a call runs at most two intervals
    // (the reference's schedule: phases of 46, 190, 171 and 1 intervals; the last phase is open-ended)
    if (!(c.T3_0 - c.T_fill > c.t_delta) || !(c.T3_end - c.T3_0 > c.t_delta) || !(c.T4_end - c.T3_end > c.t_delta))
        bad = "phases 2, 3 and 4 must each be longer than t_delta";
    if (!(c.tauI_DO != 0) || !(c.tauI_EC != 0) || !(c.cyc_tauI != 0) || !(c.cyc_dt > 0)) bad = "tauI must be non-zero, cyc_dt positive";
    // the rate constants that are folded into the Monod denominators (sbr_rates) must be positive and finite
    if (!(c.muH > 0 && c.muH < 1e300) || !(c.muA > 0 && c.muA < 1e300) || !(c.kh > 0 && c.kh < 1e300) ||
        !(c.eta_g > 0 && c.eta_g < 1e300) || !(c.bH > 0 && c.bH < 1e300) || !(c.Ya > 0 && c.Ya < 1e300) ||
        !(c.Koh > 0 && c.Koh < 1e300))
        bad = "muH, muA, kh, eta_g, Koh, bH and Ya must be positive";
    if (c.substeps > (1 << 20)) bad = "substeps out of range";
    // the dosing integrator expands 1/s, s = V/V0, to third order in s - 1 <= EC_max t_delta / V (sbr_rk4_dose): 7e-7 with
    // the reference's EC_max; refuse configurations in which the truncation (s - 1)^4 would reach 1e-16
    if (!(c.IV > 0) || !(c.WV > 0) || !(c.EC_max * c.t_delta <= 1e-4 * (c.IV < c.WV ? c.IV : c.WV)))
        bad = "EC_max * t_delta must be below 1e-4 of the reactor volume (IV, WV > 0)";
    for (int k = 0; k < 8; ++k) if (!(c.t_ratio[k] > 0)) bad = "t_ratio entries must be positive";
    if (!bad.empty()) { delete e; return fail(nullptr, SBR_ERR_INVALID, "sbr_create: " + bad); }
    derive_params(c, e->par);
#define CREATE_TRY(call)                                                                         \
    do {                                                                                         \
        hipError_t _s = (call);                                                                  \
        if (_s != hipSuccess) {                                                                  \
            std::string m = std::string(#call) + ": " + hipGetErrorString(_s);                   \
            sbr_destroy(e);                                                                      \
            return fail(nullptr, _s == hipErrorOutOfMemory ? SBR_ERR_ALLOC : SBR_ERR_HIP, m);    \
        }                                                                                        \
    } while (0)
    DeviceGuard guard(device_id);
    CREATE_TRY(guard.status);
    hipDeviceProp_t prop;
    CREATE_TRY(hipGetDeviceProperties(&prop, device_id));
    if (std::string(prop.gcnArchName).rfind("gfx950", 0) != 0) {
        std::string m = std::string("sbr_create: device is ") + prop.gcnArchName + ", this build targets gfx950 only";
        sbr_destroy(e);
        return fail(nullptr, SBR_ERR_NO_DEVICE, m);
    }
    e->one_wave_envs = (int64_t)prop.multiProcessorCount * 4 * 64;
    const size_t nb = (size_t)n_envs * sizeof(double);
    CREATE_TRY(hipMalloc(&e->buf.x, SBR_NX * nb));
    CREATE_TRY(hipMalloc(&e->buf.ctrl, R_NROWS * nb));
    CREATE_TRY(hipMalloc(&e->buf.infl, SBR_NX * nb));
    CREATE_TRY(hipMalloc(&e->tables, 2 * kTableDoubles * sizeof(double)));
    CREATE_TRY(hipMemset(e->buf.x, 0, SBR_NX * nb));
    CREATE_TRY(hipMemset(e->buf.ctrl, 0, R_NROWS * nb));
    CREATE_TRY(hipMemset(e->buf.infl, 0, SBR_NX * nb));
    // an env is unusable until its first reset: mark everything done so that step() is a no-op until then
    // (meta = steps*64 + m1i*32 + idle*16 + status*2 + done  =>  1.0 = "done"); filled on the device, no host staging buffer
    hipLaunchKernelGGL(k_fill, grid_for(n_envs), dim3(SBR_BLOCK), 0, nullptr, e->buf.ctrl + (size_t)R_META * n_envs, n_envs, 1.0);
    CREATE_TRY(hipGetLastError());
    CREATE_TRY(hipDeviceSynchronize());
    CREATE_TRY(hipEventCreate(&e->ev0));
    CREATE_TRY(hipEventCreate(&e->ev1));
    {
        const int lds_bytes = kLdsTableDoubles * (int)sizeof(double);      // 84 KiB of dynamic LDS: above the 64 KiB default
        const void* fns[8] = {reinterpret_cast<const void*>(&k_reset<float, false>), reinterpret_cast<const void*>(&k_reset<float, true>),
                              reinterpret_cast<const void*>(&k_reset<double, false>), reinterpret_cast<const void*>(&k_reset<double, true>),
                              reinterpret_cast<const void*>(&k_reset<float, false, 512>), reinterpret_cast<const void*>(&k_reset<float, true, 512>),
                              reinterpret_cast<const void*>(&k_reset<double, false, 512>), reinterpret_cast<const void*>(&k_reset<double, true, 512>)};
        for (const void* fn : fns) CREATE_TRY(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
        const void* cfns[4] = {reinterpret_cast<const void*>(&k_cycle_reset<float, false>), reinterpret_cast<const void*>(&k_cycle_reset<float, true>),
                               reinterpret_cast<const void*>(&k_cycle_reset<double, false>), reinterpret_cast<const void*>(&k_cycle_reset<double, true>)};
        for (const void* fn : cfns) CREATE_TRY(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
    }
#undef CREATE_TRY
    e->buf.n = n_envs; e->buf.first_env_id = first_env_id;
    *out = e;
    return SBR_OK;
}

int sbr_destroy(sbr_env* e) {
    if (!e) return SBR_OK;
    DeviceGuard guard(e->device);
    if (e->buf.x) (void)hipFree(e->buf.x);
    if (e->buf.ctrl) (void)hipFree(e->buf.ctrl);
    if (e->buf.infl) (void)hipFree(e->buf.infl);
    if (e->tables) (void)hipFree(e->tables);
    if (e->ev0) (void)hipEventDestroy(e->ev0);
    if (e->ev1) (void)hipEventDestroy(e->ev1);
    delete e;
    return SBR_OK;
}

const char* sbr_last_error(const sbr_env* e) { return e ? e->err.c_str() : g_create_err.c_str(); }

int sbr_query(const sbr_env* e, int32_t what, int64_t* out) {
    if (!e || !out) return SBR_ERR_INVALID;
    const bool b5 = e->cfg.scheme == 1;
    switch (what) {
        case SBR_Q_ONE_WAVE_ENVS: *out = e->one_wave_envs; break;
        case SBR_Q_STEP_SMALL_BATCH_ENVS: *out = SBR_SMALL_BATCH; break;
        case SBR_Q_STEP_BLOCK: *out = e->n <= SBR_SMALL_BATCH ? 64 : 256; break;
        case SBR_Q_STEP_WAVES: *out = (e->n > SBR_SMALL_BATCH && b5 && e->n > SBR_STEP_ONE_WAVE_ENVS(e)) ? 2 : 1; break;
        case SBR_Q_STEP_TWO_WAVES_ABOVE_ENVS: *out = SBR_STEP_ONE_WAVE_ENVS(e) > SBR_SMALL_BATCH ? SBR_STEP_ONE_WAVE_ENVS(e) : SBR_SMALL_BATCH; break;
        case SBR_Q_FUSED_ONE_WAVE_MAX_ENVS: *out = SBR_FUSED_ONE_WAVE_ENVS(e); break;
        case SBR_Q_ROLLOUT_WAVES: *out = (b5 && e->n <= SBR_FUSED_ONE_WAVE_ENVS(e)) ? 1 : 2; break;
        case SBR_Q_RESET_BLOCK: *out = e->n > e->one_wave_envs ? 512 : SBR_RESET_BLOCK; break;
        case SBR_Q_SCHEME: *out = e->cfg.scheme; break;
        default: return SBR_ERR_INVALID;
    }
    return SBR_OK;
}
int64_t sbr_num_envs(const sbr_env* e) { return e ? e->n : 0; }

int sbr_set_influent_tables(sbr_env* e, const double* means, const double* stds) {
    if (!e || !means || !stds) return fail(e, SBR_ERR_INVALID, "sbr_set_influent_tables: NULL argument");
    ON_DEVICE(e);
    HIP_TRY(e, hipMemcpy(e->tables, means, kTableDoubles * sizeof(double), hipMemcpyHostToDevice));
    HIP_TRY(e, hipMemcpy(e->tables + kTableDoubles, stds, kTableDoubles * sizeof(double), hipMemcpyHostToDevice));
    e->have_tables = true;
    return SBR_OK;
}

static int reset_impl(sbr_env* e, bool carry, uint64_t seed, const int32_t* scenario, const double* rnd,
                      const double* influent, const uint8_t* mask, void* obs, void* stream) {
    if (!e) return SBR_ERR_INVALID;
    if (!influent && !e->have_tables)
        return fail(e, SBR_ERR_INVALID, "sbr_reset: no influent given and sbr_set_influent_tables was never called");
    ON_DEVICE(e);
    hipStream_t st = (hipStream_t)stream;
    const size_t lds = influent ? 0 : kLdsTableDoubles * sizeof(double);
    const bool wide = e->n > e->one_wave_envs;           // more waves than SIMDs: 512-thread workgroups, two waves per SIMD
    const int bs = wide ? 512 : SBR_RESET_BLOCK;
    const dim3 grid((unsigned)((e->n + bs - 1) / bs)), blk(bs);
#define RESET_LAUNCH_B(T, C, B) hipLaunchKernelGGL((k_reset<T, C, B>), grid, blk, lds, st, e->par, e->buf, e->tables, seed, scenario, \
                                                   rnd, influent, mask, (T*)obs)
#define RESET_LAUNCH(T, C) do { if (wide) RESET_LAUNCH_B(T, C, 512); else RESET_LAUNCH_B(T, C, SBR_RESET_BLOCK); } while (0)
    if (e->cfg.out_f64) { if (carry) RESET_LAUNCH(double, true); else RESET_LAUNCH(double, false); }
    else { if (carry) RESET_LAUNCH(float, true); else RESET_LAUNCH(float, false); }
#undef RESET_LAUNCH
#undef RESET_LAUNCH_B
    HIP_TRY(e, hipGetLastError());
    return SBR_OK;
}

int sbr_reset(sbr_env* e, uint64_t seed, const int32_t* scenario, const double* rnd, const double* influent,
              const uint8_t* mask, void* obs, void* stream) {
    return reset_impl(e, false, seed, scenario, rnd, influent, mask, obs, stream);
}

int sbr_reset_carry(sbr_env* e, uint64_t seed, const int32_t* scenario, const double* rnd, const double* influent,
                    const uint8_t* mask, void* obs, void* stream) {
    return reset_impl(e, true, seed, scenario, rnd, influent, mask, obs, stream);
}

int sbr_set_trace(sbr_env* e, double* buf, int64_t n_envs, int64_t capacity, int32_t record_width) {
    if (!e || (buf && (n_envs <= 0 || n_envs > e->n || capacity <= 0)))
        return fail(e, SBR_ERR_INVALID, "sbr_set_trace: need 0 < n_envs <= N and capacity > 0");
    if (buf && record_width != SBR_NTRACE)
        return fail(e, SBR_ERR_INVALID, "sbr_set_trace: record_width " + std::to_string(record_width) + " != SBR_NTRACE " +
                                            std::to_string(SBR_NTRACE) + " of this library (caller built against another header?)");
    e->buf.trace = buf; e->buf.n_trace = buf ? n_envs : 0; e->buf.trace_cap = buf ? capacity : 0;
    return SBR_OK;
}

int sbr_step(sbr_env* e, const void* action, void* obs, void* state, void* reward, uint8_t* done, void* stream) {
    if (!e || !action) return fail(e, SBR_ERR_INVALID, "sbr_step: NULL env or action");
    ON_DEVICE(e);
    hipStream_t st = (hipStream_t)stream;
    if (e->cfg.out_f64) {
        if (e->cfg.act_f64) launch_step<double, double>(e, action, obs, state, reward, done, st);
        else launch_step<double, float>(e, action, obs, state, reward, done, st);
    } else {
        if (e->cfg.act_f64) launch_step<float, double>(e, action, obs, state, reward, done, st);
        else launch_step<float, float>(e, action, obs, state, reward, done, st);
    }
    HIP_TRY(e, hipGetLastError());
    return SBR_OK;
}

int sbr_cycle_reset(sbr_env* e, uint64_t seed, const int32_t* scenario, const double* rnd, const double* influent,
                    const uint8_t* mask, int32_t carry_over, void* obs, void* stream) {
    if (!e) return SBR_ERR_INVALID;
    if (!influent && !e->have_tables)
        return fail(e, SBR_ERR_INVALID, "sbr_cycle_reset: no influent given and sbr_set_influent_tables was never called");
    ON_DEVICE(e);
    hipStream_t st = (hipStream_t)stream;
    const size_t lds = influent ? 0 : kLdsTableDoubles * sizeof(double);
    const dim3 grid((unsigned)((e->n + SBR_RESET_BLOCK - 1) / SBR_RESET_BLOCK)), blk(SBR_RESET_BLOCK);
#define CRESET(T, C) hipLaunchKernelGGL((k_cycle_reset<T, C>), grid, blk, lds, st, e->par, e->buf, e->tables, seed, scenario, rnd, \
                                        influent, mask, (T*)obs)
    if (e->cfg.out_f64) { if (carry_over) CRESET(double, true); else CRESET(double, false); }
    else { if (carry_over) CRESET(float, true); else CRESET(float, false); }
#undef CRESET
    HIP_TRY(e, hipGetLastError());
    return SBR_OK;
}

int sbr_cycle_step(sbr_env* e, const void* action, void* obs, void* reward, double* diag, void* stream) {
    if (!e || !action) return fail(e, SBR_ERR_INVALID, "sbr_cycle_step: NULL env or action");
    ON_DEVICE(e);
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid = grid_for(e->n), blk(SBR_BLOCK);
#define CSTEP1(T, A, S, W) hipLaunchKernelGGL((k_cycle<T, A, S, W>), grid, blk, 0, st, e->par, e->buf, (const A*)action, (T*)obs, (T*)reward, diag)
#define CSTEP(T, A) do { if (e->cfg.scheme == 1) { if (e->n <= SBR_FUSED_ONE_WAVE_ENVS(e)) CSTEP1(T, A, 1, 1); else CSTEP1(T, A, 1, 2); } \
                         else CSTEP1(T, A, 0, 2); } while (0)
    if (e->cfg.out_f64) { if (e->cfg.act_f64) CSTEP(double, double); else CSTEP(double, float); }
    else { if (e->cfg.act_f64) CSTEP(float, double); else CSTEP(float, float); }
#undef CSTEP
#undef CSTEP1
    HIP_TRY(e, hipGetLastError());
    return SBR_OK;
}

int sbr_rollout(sbr_env* e, int32_t n_steps, uint64_t policy_seed, double* returns, float* actions_out, void* stream) {
    if (!e || n_steps < 0) return fail(e, SBR_ERR_INVALID, "sbr_rollout: bad argument");
    ON_DEVICE(e);
#define ROLL(O, S, W) hipLaunchKernelGGL((k_rollout<O, S, W>), grid_for(e->n), dim3(SBR_BLOCK), 0, (hipStream_t)stream, e->par, e->buf, \
                                         n_steps, policy_seed, returns, actions_out)
    const bool one_wave = e->cfg.scheme == 1 && e->n <= SBR_FUSED_ONE_WAVE_ENVS(e);
    if (e->cfg.reward_kind == 2) { if (e->cfg.scheme == 1) { if (one_wave) ROLL(true, 1, 1); else ROLL(true, 1, 2); } else ROLL(true, 0, 2); }
    else { if (e->cfg.scheme == 1) { if (one_wave) ROLL(false, 1, 1); else ROLL(false, 1, 2); } else ROLL(false, 0, 2); }
#undef ROLL
    HIP_TRY(e, hipGetLastError());
    return SBR_OK;
}

int sbr_reduce_stats(sbr_env* e, const double* values, int64_t n, double* out4, void* stream) {
    if (!e || !values || !out4 || n < 0) return fail(e, SBR_ERR_INVALID, "sbr_reduce_stats: bad argument");
    ON_DEVICE(e);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(k_stats_init, dim3(1), dim3(SBR_BLOCK), 0, st, out4);
    if (n > 0) {
        int64_t blocks = (n + SBR_BLOCK - 1) / SBR_BLOCK;
        if (blocks > 2048) blocks = 2048;      // grid-stride beyond 8 waves per CU
        hipLaunchKernelGGL(k_stats, dim3((unsigned)blocks), dim3(SBR_BLOCK), 0, st, values, n, out4);
    }
    HIP_TRY(e, hipGetLastError());
    return SBR_OK;
}

int sbr_get_state(sbr_env* e, double* x, double* ctrl, void* stream) {
    if (!e) return SBR_ERR_INVALID;
    ON_DEVICE(e);
    const size_t nb = (size_t)e->n * sizeof(double);
    if (x) HIP_TRY(e, hipMemcpyAsync(x, e->buf.x, SBR_NX * nb, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    if (ctrl) {
        hipLaunchKernelGGL(k_export, grid_for(e->n), dim3(SBR_BLOCK), 0, (hipStream_t)stream, e->par, e->buf, ctrl, -1);
        HIP_TRY(e, hipGetLastError());
    }
    return SBR_OK;
}

int sbr_set_state(sbr_env* e, const double* x, const double* ctrl, void* stream) {
    if (!e) return SBR_ERR_INVALID;
    ON_DEVICE(e);
    const size_t nb = (size_t)e->n * sizeof(double);
    if (x) {
        // a plant without controller rows: So[-1] / Sno[-1] of envs that hold them implicitly (x[8], x[9] of the OLD plant)
        // are written out first, so that replacing the plant does not replace the controllers' memories with it
        if (!ctrl) {
            hipLaunchKernelGGL(k_m1_explicit, grid_for(e->n), dim3(SBR_BLOCK), 0, (hipStream_t)stream, e->buf);
            HIP_TRY(e, hipGetLastError());
        }
        HIP_TRY(e, hipMemcpyAsync(e->buf.x, x, SBR_NX * nb, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    }
    if (ctrl) {
        hipLaunchKernelGGL(k_import, grid_for(e->n), dim3(SBR_BLOCK), 0, (hipStream_t)stream, e->par, e->buf, ctrl);
        HIP_TRY(e, hipGetLastError());
    }
    return SBR_OK;
}

int sbr_get_ctrl_row(sbr_env* e, int32_t row, double* out, void* stream) {
    if (!e || !out || row < 0 || row >= SBR_NCTRL) return fail(e, SBR_ERR_INVALID, "sbr_get_ctrl_row: bad argument");
    ON_DEVICE(e);
    if (row == SBR_C_RETURN) {           // stored as is: a plain device-to-device copy
        HIP_TRY(e, hipMemcpyAsync(out, e->buf.ctrl + (size_t)R_RET * e->n, (size_t)e->n * sizeof(double),
                                  hipMemcpyDeviceToDevice, (hipStream_t)stream));
    } else {
        hipLaunchKernelGGL(k_export, grid_for(e->n), dim3(SBR_BLOCK), 0, (hipStream_t)stream, e->par, e->buf, out, (int)row);
        HIP_TRY(e, hipGetLastError());
    }
    return SBR_OK;
}

int sbr_get_influent(sbr_env* e, double* out, void* stream) {
    if (!e || !out) return fail(e, SBR_ERR_INVALID, "sbr_get_influent: NULL argument");
    ON_DEVICE(e);
    HIP_TRY(e, hipMemcpyAsync(out, e->buf.infl, SBR_NX * (size_t)e->n * sizeof(double), hipMemcpyDeviceToDevice,
                              (hipStream_t)stream));
    return SBR_OK;
}

int sbr_eval_rhs(sbr_env* e, int32_t kind, int64_t n, const double* x, const double* kla, const double* ec,
                 const double* loading, double* dx, void* stream) {
    if (!e || !x || !kla || !ec || !dx || kind < 0 || kind > 2 || (kind == 1 && !loading))
        return fail(e, SBR_ERR_INVALID, "sbr_eval_rhs: bad argument");
    ON_DEVICE(e);
    if (n > 0)
        hipLaunchKernelGGL(k_rhs, grid_for(n), dim3(SBR_BLOCK), 0, (hipStream_t)stream, e->par, kind, n, x, kla, ec,
                           loading, dx);
    HIP_TRY(e, hipGetLastError());
    return SBR_OK;
}

int sbr_eval_substeps(sbr_env* e, int32_t kind, int64_t n, int32_t n_sub, const double* x0, const double* kla, const double* ec,
                      const double* loading, const double* h, double* xs, double* dxs, void* stream) {
    if (!e || !x0 || !kla || !h || !xs || !dxs || n < 0 || n_sub < 1 || n_sub > (1 << 20) || kind < 0 || kind > 3 ||
        (kind == 0 && !ec) || (kind == 1 && !loading))
        return fail(e, SBR_ERR_INVALID, "sbr_eval_substeps: bad argument");
    ON_DEVICE(e);
    if (n > 0)
        hipLaunchKernelGGL(k_substeps, grid_for(n), dim3(SBR_BLOCK), 0, (hipStream_t)stream, e->par, kind, n, n_sub, x0, kla, ec,
                           loading, h, xs, dxs);
    HIP_TRY(e, hipGetLastError());
    return SBR_OK;
}

int sbr_draw_normals(sbr_env* e, uint64_t seed, double* out, void* stream) {
    if (!e || !out) return fail(e, SBR_ERR_INVALID, "sbr_draw_normals: NULL argument");
    ON_DEVICE(e);
    hipLaunchKernelGGL(k_normals, grid_for(e->n), dim3(SBR_BLOCK), 0, (hipStream_t)stream, e->buf, seed, out);
    HIP_TRY(e, hipGetLastError());
    return SBR_OK;
}

int sbr_draw_scenarios(sbr_env* e, uint64_t seed, int32_t* out, void* stream) {
    if (!e || !out) return fail(e, SBR_ERR_INVALID, "sbr_draw_scenarios: NULL argument");
    ON_DEVICE(e);
    hipLaunchKernelGGL(k_scenarios, grid_for(e->n), dim3(SBR_BLOCK), 0, (hipStream_t)stream, e->buf, seed, out);
    HIP_TRY(e, hipGetLastError());
    return SBR_OK;
}

#ifdef SBR_STAMPS
int sbr_set_stamps(sbr_env* e, unsigned long long* buf) { if (!e) return SBR_ERR_INVALID; e->buf.stamps = buf; return SBR_OK; }
#endif

int sbr_synchronize(sbr_env* e, void* stream) {
    if (!e) return SBR_ERR_INVALID;
    ON_DEVICE(e);
    HIP_TRY(e, hipStreamSynchronize((hipStream_t)stream));
    return SBR_OK;
}

int sbr_timer_start(sbr_env* e, void* stream) {
    if (!e) return SBR_ERR_INVALID;
    ON_DEVICE(e);
    HIP_TRY(e, hipEventRecord(e->ev0, (hipStream_t)stream));
    return SBR_OK;
}

int sbr_timer_stop(sbr_env* e, void* stream, float* elapsed_ms) {
    if (!e || !elapsed_ms) return SBR_ERR_INVALID;
    ON_DEVICE(e);
    HIP_TRY(e, hipEventRecord(e->ev1, (hipStream_t)stream));
    HIP_TRY(e, hipEventSynchronize(e->ev1));
    HIP_TRY(e, hipEventElapsedTime(elapsed_ms, e->ev0, e->ev1));
    return SBR_OK;
}

}  // extern "C"
