/*
 * sbr_amd.h - C ABI of the MI355X-native batched SBR environment (libsbr_amd.so).
 *
 * Drop-in boundary for ONE path of SungKu/gym-SBR2: the step-level env `SBROS-v1`
 * (gym_SBR/envs/gym_SBR_oneshot.py::SbrOS; registered at gym_SBR/__init__.py:11).  The
 * reference has no FFI - its boundary is the gym.Env protocol - so each entry point names the
 * Python method it replaces.  All array arguments are DEVICE pointers (HIP, e.g.
 * torch.Tensor.data_ptr()); `stream` is a hipStream_t passed as void* (NULL = default stream).
 * No torch types, no C++ types.  Every call returns 0 on success or a negative sbr_status; the
 * message is available from sbr_last_error().  Nothing here ever falls back to the CPU.
 *
 * Layouts
 *   action  [N][2]  ActT      (u_DO set-point, u_EC set-point)   gym_SBR_oneshot.py:843,862,898
 *                             ActT = float32 (cfg.act_f64 = 0) or float64 (cfg.act_f64 = 1, what the reference's
 *                             step() receives; the NO3 loop's gain makes EC sensitive to set-point rounding)
 *   obs     [N][18] OutT      obs_DO[9] ++ obs_EC[9]             gym_SBR_oneshot.py:1027-1114
 *   state   [N][15] OutT      [t, x0..x13] / x_1_state           gym_SBR_oneshot.py:1020-1025
 *   reward  [N]     OutT                                         module_reward_EQIOCI.py:4-115
 *   done    [N]     uint8                                        gym_SBR_oneshot.py:1122-1124
 *   OutT = float32 (cfg.out_f64 = 0, the RL path) or float64 (cfg.out_f64 = 1, parity checks).
 *   Internal plant/controller state is float64, struct-of-arrays [field][N] (sbr_get_state).
 */
#ifndef SBR_AMD_H
#define SBR_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SBR_NX 14          /* V Si Ss Xi Xs Xbh Xba Xp So Sno Snh Snd Xnd Salk (gym_SBR_oneshot.py:185-188) */
#define SBR_NOBS 18
#define SBR_NSTATE 15
#define SBR_NACT 2
#define SBR_NSCEN 8        /* influent scenarios, buffer_tank3.py:18-1197 */
#define SBR_NSAMP 48       /* samples per influent series */
#define SBR_NSERIES 14     /* 13 concentrations + flow q */
#define SBR_KLA_HIST 10    /* Kla values the reward can look back on (current + 9) */
/* controller/bookkeeping doubles per env exposed by sbr_get_state/sbr_set_state, in this order.  This is the PUBLIC
 * layout; the kernels keep a leaner internal one (Kla history as a ring, packed bookkeeping) and translate. */
#define SBR_NCTRL 25
enum {
    SBR_C_T = 0,           /* running time t (days)                     gym_SBR_oneshot.py:1357 */
    SBR_C_SO_M1, SBR_C_SO_M2, SBR_C_SNO_M1, SBR_C_SNO_M2,   /* So[-1] So[-2] Sno[-1] Sno[-2]  :1959-1961 */
    SBR_C_IE_DO, SBR_C_IE_EC,                               /* PID integrals                  :1893,:1923 */
    SBR_C_EC_LAST,                                          /* EC[-1] */
    SBR_C_KLA_HIST0,                                        /* 10 entries, oldest first; the last is Kla[-1] */
    SBR_C_KLA_LAST = SBR_C_KLA_HIST0 + SBR_KLA_HIST - 1,
    SBR_C_QW,                                               /* wastage flow of the last terminal step :2376 */
    SBR_C_RETURN,                                           /* sum of rewards since reset */
    SBR_C_STEPS,                                            /* step() calls since reset (as double) */
    SBR_C_DONE,                                             /* 1.0 once the episode ended */
    SBR_C_STATUS,                                           /* sticky SBR_ST_* bits since reset (as double) */
    SBR_C_KLA_SUM,                                          /* sum(Kla) of the episode's whole list, in append order:
                                                               the 252 reset entries (:323), one per interval, idle's.
                                                               Advanced only with reward_kind 2 (else: its reset value) */
    SBR_C_PLAN                                              /* (round 6) what cfg.scheme = 1 did in the LAST control interval
                                                               sbr_step ran for this env: its Butcher-5 step count (1 .. 64)
                                                               + SBR_PLAN_SLAVED if dissolved oxygen was held during the
                                                               steps.  0 = nothing to report: cfg.scheme = 0, no sbr_step
                                                               since the reset / sbr_set_state / sbr_rollout.  The reference's
                                                               counterpart is LSODA's infodict (nst, nfe) at its odeint call
                                                               sites (gym_SBR_oneshot.py:1953, :2041).  sbr_set_state keeps
                                                               the value given (0 .. 255).  Costs no memory traffic: it shares
                                                               the internal row of steps / status / done */
};
#define SBR_PLAN_SLAVED 128
#define SBR_PLAN_STEPS(code) ((int)(code) & 127)
/* (The reference's u_DO / u_EC globals and the EC value before EC[-1] are temporaries of one step() call - every
 * interval overwrites them before use - so they are not part of the state.) */

/* Domain-of-validity flags.  The ASM1 rate expressions x/(K+x) have poles at x = -K and the reference has no
 * guards (gym_SBR_oneshot.py:1660-1685): heterotrophic growth takes up ammonia without an ammonia limitation, so an
 * aggressive policy can drive Snh (also So, Sno) below zero and towards a pole, after which the reference - and this
 * library, which reproduces it - returns numbers without physical meaning.  The numerics are NOT altered; these
 * bits, checked at the end of every control interval, only make the condition visible. */
#define SBR_ST_NEGATIVE 1   /* one of Ss, Xs, Xbh, So, Sno, Snh was below -1e-6 */
#define SBR_ST_NEAR_POLE 2  /* Ss, So, Sno or Snh came within 50 % of a Monod pole (x < -K/2): parity to 1e-5 between
                               any two fp64 implementations holds only while this bit is clear (tests/) */
#define SBR_ST_NONFINITE 4  /* a state component is NaN or infinite */

typedef enum {
    SBR_OK = 0,
    SBR_ERR_INVALID = -1,      /* bad argument */
    SBR_ERR_NO_DEVICE = -2,    /* no HIP device / wrong architecture */
    SBR_ERR_HIP = -3,          /* a HIP runtime call failed */
    SBR_ERR_ALLOC = -4
} sbr_status;

/* Every constant of the path.  sbr_default_config() fills in the reference's values
 * (SURVEY.md Appendix A lists file:line for each). */
typedef struct sbr_config {
    /* ASM1 stoichiometry / kinetics                              gym_SBR_oneshot.py:116-119 */
    double Ya, Yh, fp, ixb, ixp;
    double muH, Ks, Koh, Kno, bH, eta_g, eta_h, kh, Kx, muA, Knh, bA, Koa, ka;
    /* plant + time grid                                          gym_SBR_oneshot.py:25-37 */
    double WV, IV, dt, t_delta, t_cycle;
    double T_fill, T3_0, T3_end, T4_end, T5_end;      /* phase scalars, module_batch_time.py:3-116 */
    double t_settle, t_draw;                          /* t_ratio[5], t_ratio[6] (fractions of t_cycle) */
    /* controllers                                                gym_SBR_oneshot.py:80-96 */
    double So_sat, Kla_min, Kla_max, Kc_DO, tauI_DO, tauD_DO;
    double EC_min, EC_max, Kc_EC, tauI_EC, tauD_EC, EC_conc;
    double act_DO_max, act_EC_max;                    /* action clipping :865-870, :901-906 */
    /* terminal phases                                            gym_SBR_oneshot.py:123-124, :2189-2218 */
    double biomass_setpoint, Qeff, settler_area, settler_vmax;
    /* per-cycle env SBR-v2 (gym_SBR_env2.py:32-48, SBR_model_FB.py:18-29): the eight phase fractions of the cycle and
     * the positional DO-PID of sub_phases_FB.py:178-271 (DO_control_par = [Kc, tauI, delt, ..., tauD = [9]]) */
    double t_ratio[8];
    double cyc_Kc, cyc_tauI, cyc_tauD, cyc_dt;
    double x0[SBR_NX];                                /* episode start state :201-203 */
    /* integrator */
    int32_t substeps;          /* RK4 substeps per control interval (10 => h = dt) */
    int32_t out_f64;           /* 0: obs/state/reward are float32; 1: float64 */
    int32_t terminal;          /* 1: run settle/draw/idle on the done step (reference behaviour) */
    int32_t reward_kind;       /* 0: EQI/OCI reward of module_reward_EQIOCI.py (SBROS-v1); 1: the piecewise-linear reward of
                                  module_reward_continuous_G2ANET.py:4-45 (used by the variant gym_SBR_oneshot_copy.py:17,614);
                                  2: the operating-cost reward of module_reward_continuous.py:4-65 (SbrEnv3/SbrEnv4): its
                                  reaction-interval branch (0.5 - AE of the Kla just applied) on every call, and on the
                                  done call, when the terminal phases run, its end-of-cycle branch (sum(Kla) of the whole
                                  episode, pumping of Qw and Qeff, -246 if the effluent ammonia is >= 4 g/m3) */
    int32_t act_f64;           /* 0: sbr_step reads float32 actions; 1: float64.  A finished env ignores step()
                                  until sbr_reset either way (the reference leaves resetting to the caller) */
    int32_t random_scenario;   /* what sbr_reset / sbr_cycle_reset do when `scenario` is NULL: 0 = the fixed scenario of the
                                  reference class (6 for SbrOS :180, 0 for SbrEnv2 :104); 1 = draw one of the 8 scenarios
                                  per env and reset on the device, uniformly, as SbrEnv4.reset does with
                                  np.random.choice(8, 1) (gym_SBR_env4.py:107): Philox4x32-10 keyed by `seed`, stream 2,
                                  subsequence = GLOBAL env id (sbr_draw_scenarios returns the same draw) */
    int32_t scheme;            /* how the control intervals of sbr_step / sbr_rollout (reaction_dxdt :1658-1787, odeint call sites
                                  :1953, :2041), the idle phase of the done call (:2587) and the reaction and idle intervals of
                                  sbr_cycle_step are integrated.  (The fill phase - sbr_reset, :1647, and sbr_cycle_step's first
                                  phase - is integrated with RK4, h = dt, under either scheme.)
                                  0: classical RK4, `substeps` substeps per control interval (40 right-hand-side evaluations).
                                  1 (default, round 5): Butcher's fifth-order scheme with 1, 2 or 4 steps chosen per env and
                                  interval from the env's own state - dissolved oxygen, the system's one stiff mode, is held
                                  where it is slaved to zero, and an env whose oxygen mode is stiffer than the reference plant's
                                  takes more steps, as does one whose substrate, ammonia or nitrate moves across its
                                  half-saturation constant within the interval: ~10 evaluations per interval, and closer to the
                                  reference's trajectories than scheme 0 (closed loop, worst 0.36 of the 1e-5 gate against 0.51).
                                  `substeps` then only sets the fill intervals of sbr_cycle_step; the idle phase is cut into
                                  ceil(rows / 10) macro intervals.  sbr_eval_substeps replays RK4 nodes under either scheme.
                                  VALIDITY DOMAIN of scheme 1 (round 6).  The step count is a start-of-interval plan, not an error
                                  estimate.  What it bounds by construction: the oxygen mode's rate times the step (<= 2.5 of
                                  Butcher-5's real stability limit 3.39, whatever the kinetic constants and the biomass).  What it
                                  ASSUMES: every other mode (uptake of Ss, Snh, Sno; hydrolysis) has |lambda| * t_delta below
                                  ~2.5 - with the reference's constants they stay below 1.0 on every captured interval, 2.2 in
                                  the idle phase's thickened sludge.  Pinned to the reference on 24 fitted + 10 held-out
                                  episodes (closed loop <= 0.373 of the gate), 36 000 intervals of the per-cycle env, and probed
                                  off-regime: on random states of a plant with muH, muA, kh up to 4 x faster it misses the gate
                                  on 134 of 2 836 states (11 beyond 30 gates) where scheme 0 misses 420 (239): scheme 0's fixed
                                  h = dt is unstable once the oxygen rate times dt exceeds 2.785, scheme 1's count grows with
                                  it (tests/test_oracle_golden.py::test_scheme1_under_perturbed_kinetic_constants,
                                  ..._plan_on_states_far_from_the_reference_regime).  A configuration with kinetics several
                                  times faster than the reference's or a longer t_delta is outside what either scheme was
                                  validated on: check it against a fine solution (sbr_eval_substeps with many substeps) first.
                                  OUTSIDE THE MODEL'S DOMAIN (the Monod factor of Ss or Snh outside [0, 1]: a concentration
                                  negative towards or beyond its pole, or NaN - SBR_ST_NEAR_POLE is then set or about to be) the
                                  oxygen rate is no longer bounded and the step count is NOT raised above the knee's four: the
                                  state is garbage either way (the reference has no guards), and a launch lasts as long as its
                                  slowest wavefront - one such env at the cap of 64 steps made a whole 65 536-env batch
                                  several times slower (round 6).  In-domain states are unaffected.
                                  What scheme 1 did is observable per env and call: SBR_C_PLAN, SBR_TR_PLAN. */
    int32_t reserved_;         /* keeps the struct a multiple of 8 bytes; MUST be 0: sbr_create rejects anything else with
                                  SBR_ERR_INVALID (round 6), so that a later round can give the word a meaning */
} sbr_config;

typedef struct sbr_env sbr_env;      /* opaque handle: owns all device state for N envs on one GPU */

/* library / configuration ---------------------------------------------------------------- */
const char* sbr_version(void);
/* Bumped whenever a signature, a struct layout or a record width of this header changes.  A consumer compiled against this
 * header checks sbr_abi_version() == SBR_ABI_VERSION once after loading the library (round 4 = 4: sbr_set_trace takes the
 * record width, SBR_NTRACE = 34; round 5 = 5: sbr_config.scheme; round 6 = 6: SBR_NCTRL = 25 with SBR_C_PLAN, SBR_NTRACE = 36 with
 * SBR_TR_PLAN / SBR_TR_PLAN_FIRST, sbr_query, sbr_config.reserved_ checked). */
#define SBR_ABI_VERSION 6
int sbr_abi_version(void);
int sbr_default_config(sbr_config* cfg);
int sbr_device_count(void);          /* HIP devices visible; 0 if none (never throws) */
/* Host-side helper, no device needed: the reference's control interval has len(t_range) = int(((t + t_delta) - t)/dt) output
 * rows (gym_SBR_oneshot.py:1339, :1384), 9 or 10 depending on the rounding of (t + t_delta) - t, and the reward's look-back
 * depends on it.  The kernels decide it with two comparisons: out2[0] / out2[1] = the smallest doubles s for which the IEEE
 * quotient s/dt reaches 9.0 / 10.0 (rows = 10 iff span >= out2[1], 9 iff out2[0] <= span < out2[1]; any other span takes the
 * division itself).  cfg NULL = defaults. */
int sbr_rows_thresholds(const sbr_config* cfg, double* out2);

/* lifetime: replaces SbrOS.__init__ (gym_SBR_oneshot.py:103-166) for N instances.
 * first_env_id: global id of local env 0 (multi-GPU sharding: RNG streams and scenario
 * assignment depend on the GLOBAL id, so results do not depend on world size). */
int sbr_create(int64_t n_envs, int device_id, int64_t first_env_id, const sbr_config* cfg /* NULL = defaults */,
               sbr_env** out);
int sbr_destroy(sbr_env* env);
const char* sbr_last_error(const sbr_env* env /* NULL = creation errors */);
int64_t sbr_num_envs(const sbr_env* env);

/* What the library decided for this handle from its size and the device it sits on (round 6).  The launch shapes depend on the
 * device's CU count - a partitioned MI355X has fewer than 256 - so a consumer that wants to NAME the kernel a handle runs
 * (bench.py's `config.kernel`, a profiler's filter) asks instead of repeating thresholds.  out receives one integer. */
enum {
    SBR_Q_ONE_WAVE_ENVS = 0,            /* envs that put one wavefront on every SIMD of the device: CUs x 4 x 64 (MI355X: 65536) */
    SBR_Q_STEP_SMALL_BATCH_ENVS,        /* up to this many envs sbr_step launches 64-thread workgroups (49152) */
    SBR_Q_STEP_BLOCK,                   /* workgroup size of THIS handle's sbr_step launches: 64 or 256 */
    SBR_Q_STEP_WAVES,                   /* 1 or 2: which register budget of k_step THIS handle runs (2 = two waves per SIMD:
                                           cfg.scheme = 1 and more envs than SBR_Q_STEP_TWO_WAVES_ABOVE_ENVS) */
    SBR_Q_STEP_TWO_WAVES_ABOVE_ENVS,
    SBR_Q_FUSED_ONE_WAVE_MAX_ENVS,      /* up to this many envs sbr_rollout / sbr_cycle_step run their uncapped-register build */
    SBR_Q_ROLLOUT_WAVES,                /* 1 or 2 for THIS handle's sbr_rollout (and sbr_cycle_step under cfg.scheme = 1) */
    SBR_Q_RESET_BLOCK,                  /* workgroup size of THIS handle's sbr_reset launches: 256 or 512 */
    SBR_Q_SCHEME                        /* cfg.scheme in force */
};
int sbr_query(const sbr_env* env, int32_t what, int64_t* out);

/* influent data: replaces the literals of buffer_tank3.py:18-1197.  means/stds are HOST pointers,
 * [SBR_NSCEN][SBR_NSERIES][SBR_NSAMP] float64; copied to the device once. */
int sbr_set_influent_tables(sbr_env* env, const double* means_host, const double* stds_host);

/* reset: replaces SbrOS.reset() (gym_SBR_oneshot.py:168-438) = influent draw
 * (buffer_tank3.py:68-107) + fill phase (Sim_filling :1585-1654) + reset observation.
 *   scenario  [N] int32 or NULL (NULL = scenario 6 for every env, as the reference :180)
 *   rnd       [N][48] float64 or NULL; NULL => drawn on the device: Philox4x32-10 keyed by `seed`,
 *             subsequence = global env id, Box-Muller (the reference draws np.random.randn(48))
 *   influent  [N][14] float64 or NULL; if given it REPLACES the draw (entry 0 is overwritten by
 *             Qin/T_fill as at :287)
 *   mask      [N] uint8 or NULL; if given only envs with mask != 0 are reset
 *   obs       [N][18] OutT or NULL */
int sbr_reset(sbr_env* env, uint64_t seed, const int32_t* scenario, const double* rnd, const double* influent,
              const uint8_t* mask, void* obs, void* stream);

/* multi-cycle operation: like sbr_reset, but every selected env starts the new cycle from ITS OWN current
 * state - x0 := x (normally the state after the idle phase), inoculum volume IV := x[0], inflow := (WV - IV)/T_fill -
 * instead of cfg.x0 / cfg.IV.  This is what the reference prepares (x0_new, IV_new at gym_SBR_env2.py:152-153) and then
 * leaves disabled (gym_SBR_oneshot.py:260-268: "IV = IV_init  # IV_new"), so there is no reference output to compare
 * with: parity is pinned device-vs-oracle only. */
int sbr_reset_carry(sbr_env* env, uint64_t seed, const int32_t* scenario, const double* rnd, const double* influent,
                    const uint8_t* mask, void* obs, void* stream);

/* trajectory export (replaces the growing lists of SbrOS.trajectory(), gym_SBR_oneshot.py:1275-1288): while a trace
 * buffer is set, every sbr_step call appends one record for each of the first n_envs environments at index = calls since
 * reset (records beyond capacity are dropped).  buf is [capacity][SBR_NTRACE][n_envs] float64, DEVICE pointer, owned by
 * the caller; buf = NULL switches tracing off.  Record (SBR_TR_*): t, x[14] (end of the call), Kla, EC (of the call's last
 * interval), reward, done, the clipped set-points u_DO / u_EC in force (:862-870, :898-906), the NO3-PID's e_EC, ie_EC,
 * dcv_EC (:1918-1926, :2006-2014) and the four diagnostics sbr_reward appends (module_reward_EQIOCI.py:109-112):
 * EQI2, OCI2 = AE_OCI2 + EC_OCI2, AE_OCI2, EC_OCI2; then what a consumer needs to rebuild the reference's sub-interval rows
 * (sbr_eval_substeps): the number of control intervals the call ran (1, or 2 on a phase-boundary call) and the Kla / EC of
 * the FIRST of them (equal to SBR_TR_KLA / SBR_TR_EC when the call ran one); and (round 4) the NO3-PID's e_EC, ie_EC, dcv_EC
 * of that FIRST interval too: the reference appends to these three lists once per INTERVAL (:1918-1926, :2006-2014), so a
 * phase-boundary call contributes two entries each (equal to SBR_TR_E_EC / _IE_EC / _DCV_EC when the call ran one).
 * (Round 6) SBR_TR_PLAN / SBR_TR_PLAN_FIRST: what cfg.scheme = 1 did in the call's last / first interval, coded like SBR_C_PLAN
 * (step count + SBR_PLAN_SLAVED; equal when the call ran one interval; 0 under cfg.scheme = 0).
 * record_width must be SBR_NTRACE of the header the caller was compiled against: the record grew from 28 to 31 to 34 to 36 doubles
 * over the rounds, and a buffer sized for an older width would be overrun silently - a mismatch is SBR_ERR_INVALID. */
#define SBR_NTRACE 36
enum { SBR_TR_T = 0, SBR_TR_X0 = 1, SBR_TR_KLA = 15, SBR_TR_EC, SBR_TR_REWARD, SBR_TR_DONE, SBR_TR_U_DO, SBR_TR_U_EC,
       SBR_TR_E_EC, SBR_TR_IE_EC, SBR_TR_DCV_EC, SBR_TR_R_EQI, SBR_TR_R_OCI, SBR_TR_R_AE, SBR_TR_R_EC,
       SBR_TR_N_IV, SBR_TR_KLA_FIRST, SBR_TR_EC_FIRST, SBR_TR_E_EC_FIRST, SBR_TR_IE_EC_FIRST, SBR_TR_DCV_EC_FIRST,
       SBR_TR_PLAN, SBR_TR_PLAN_FIRST };
int sbr_set_trace(sbr_env* env, double* buf, int64_t n_envs, int64_t capacity, int32_t record_width);

/* step: replaces SbrOS.step(action) (gym_SBR_oneshot.py:843-1273): phase logic, both PIDs,
 * one (at phase boundaries two) control interval(s) of RK4, reward, observations, and on the last
 * call of an episode the settle/draw/idle phases.  Any of obs/state/reward/done may be NULL. */
int sbr_step(sbr_env* env, const void* action, void* obs, void* state, void* reward, uint8_t* done,
             void* stream);

/* ---- per-cycle environment `SBR-v2`: gym_SBR/envs/gym_SBR_env2.py::SbrEnv2, registered at
 * gym_SBR/__init__.py:5.  One step() = one whole 12 h cycle: SBR_model_FB.run (SBR_model_FB.py:8-295) over the phase
 * simulators of sub_phases_FB.py, reward module_reward.py:4-51.  A handle is used EITHER for sbr_reset/sbr_step OR for
 * these two calls.
 *   action [N][3] ActT in [0,1] (clipped): DO set-points of phases 3, 5 and 8 are action*8   gym_SBR_env2.py:133,184-186
 *   obs    [N][3] OutT: reset: [V0 + 0.66, (COD0 + COD_in - 5145)/10, (Snh0 + Snh_in)/30]       :108-119
 *                       step:  [Qeff, COD_eff, Snh_eff/30]                                      :164-169
 *   diag   [N][SBR_NCYC_DIAG] float64 or NULL: Qw, EQI, OCI, effluent Ntot COD Snh BOD5 Sno, mean Kla of phases 3/5/8, Xf */
#define SBR_NCYC_ACT 3
#define SBR_NCYC_OBS 3
#define SBR_NCYC_DIAG 12
/* replaces SbrEnv2.reset(): influent draw (scenario NULL = 0 as at :104; rnd/influent/mask as in sbr_reset) and the reset
 * observation.  carry_over != 0 keeps every env's current state as the start state of the next cycle (x0_new / IV_new,
 * :152-153, disabled upstream at :85-97) instead of cfg.x0. */
int sbr_cycle_reset(sbr_env* env, uint64_t seed, const int32_t* scenario, const double* rnd, const double* influent,
                    const uint8_t* mask, int32_t carry_over, void* obs, void* stream);
/* replaces SbrEnv2.step(action): every call is a complete episode (done is always true, :161). */
int sbr_cycle_step(sbr_env* env, const void* action, void* obs, void* reward, double* diag, void* stream);

/* fused rollout with an on-device uniform random policy (BASELINE.json configs[4]): n_steps step()
 * calls per env in ONE kernel, plant state held in registers; actions ~ U[0,act_DO_max] x
 * U[0,act_EC_max] from Philox keyed by policy_seed, subsequence = global env id, counter = call index.
 *   returns [N] float64 or NULL: sum of the rewards of these n_steps calls
 *   actions_out [n_steps][N][2] float32 or NULL: the sampled actions (for replay through sbr_step) */
int sbr_rollout(sbr_env* env, int32_t n_steps, uint64_t policy_seed, double* returns, float* actions_out,
                void* stream);

/* batch statistics of a per-env float64 vector (e.g. episode returns): wavefront reductions
 * + one atomic per wave.  out4 = {sum, min, max, count} float64, DEVICE pointer. */
int sbr_reduce_stats(sbr_env* env, const double* values, int64_t n, double* out4, void* stream);

/* parity injection / inspection: x is [SBR_NX][N], ctrl is [SBR_NCTRL][N], float64, DEVICE pointers.
 * An injected volume x[0] is taken as it is.  The dosing integrator expands 1/(V/V0) to third order in Q t_delta / V0
 * (sbr_create checks EC_max t_delta <= 1e-4 min(IV, WV), i.e. <= 7e-7 for the reference plant): a state injected with a
 * volume below ~1 % of IV would carry a truncation (Q t_delta / V0)^4 above 1e-16 into its dosing intervals. */
int sbr_get_state(sbr_env* env, double* x, double* ctrl, void* stream);
int sbr_set_state(sbr_env* env, const double* x, const double* ctrl, void* stream);

/* one row of the ctrl block (SBR_C_* index), e.g. SBR_C_RETURN for the episode returns: out is [N] float64, DEVICE
 * pointer; asynchronous on `stream`, no host synchronisation (sbr_get_state translates all SBR_NCTRL rows). */
int sbr_get_ctrl_row(sbr_env* env, int32_t row, double* out, void* stream);

/* the flow-weighted influent each env was reset with (buffer_tank3.py:87-107; entry 0 = Qin/T_fill,
 * gym_SBR_oneshot.py:287): out is [SBR_NX][N] float64, DEVICE pointer. */
int sbr_get_influent(sbr_env* env, double* out, void* stream);

/* standalone right-hand sides for known-answer tests (gym_SBR_oneshot.py:1658-1787, :1424-1583,
 * :2424-2552).  x [n][14], kla [n], ec [n], loading [n][14] or NULL, dx [n][14]; DEVICE pointers.
 * kind: 0 reaction, 1 filling (needs loading), 2 idle. */
int sbr_eval_rhs(sbr_env* env, int32_t kind, int64_t n, const double* x, const double* kla, const double* ec,
                 const double* loading, double* dx, void* stream);

/* Dense output of one integration span, for trajectory export: the reference returns odeint's solution on an output grid
 * of its own - linspace(t, t + t_delta, int(t_delta/dt)) = 9 or 10 points per control interval, 252 points over the fill
 * phase, 463 over the idle phase - and appends those rows to t_t / x_t / So_t ... (gym_SBR_oneshot.py:296-313, :1339, :1369,
 * :1959-1961, :1122-1155); a fixed-step integrator has its own nodes instead.  For n independent spans given by start state
 * x0 [n][14], the held Kla [n] and the substep length h [n] (days), this writes the n_sub + 1 RK4 nodes xs [n][n_sub + 1][14]
 * (node 0 = the start state; with n_sub = cfg.substeps and h = span/n_sub the last node is the state sbr_step ends the
 * interval in, to rounding) and the right-hand side at every node dxs [n][n_sub + 1][14]: values and slopes for cubic-Hermite
 * interpolation onto any grid.  kind 0: a control interval (reaction_dxdt :1658-1787, ec [n] held); 1: the fill phase
 * (filling_dxdt :1424-1583, loading [n][14] with loading[0] = the inflow); 2: the idle phase (idle_dxdt :2424-2552); 3: settle
 * and draw (Sim_Settling_Drawing :2264-2420) applied to x0 first, then the idle phase - node 0 is the reactor after the draw.
 * ec may be NULL unless kind == 0, loading unless kind == 1.  DEVICE pointers.  Not on the stepping path.
 * The nodes are RK4's under EITHER cfg.scheme.  Under cfg.scheme = 0 they are the states sbr_step itself passed through (the last
 * node equals its end state to 1e-12).  Under cfg.scheme = 1 sbr_step integrated the interval with adaptive Butcher-5 steps: the
 * replayed rows are then another discretisation of the same interval from the same start state - inside the parity gate of the
 * reference's rows like the stepped states, but their last node differs from the state sbr_step returned by the two schemes'
 * distance (~1e-2 of the gate; up to 2e-5 relative after the idle phase).  A consumer that needs rows ending exactly on the
 * stepped states pins the last row to the record (gym_sbr2_amd's trajectory(dense=True) does) or runs cfg.scheme = 0. */
int sbr_eval_substeps(sbr_env* env, int32_t kind, int64_t n, int32_t n_sub, const double* x0, const double* kla, const double* ec,
                      const double* loading, const double* h, double* xs, double* dxs, void* stream);

/* device-side normal draws used by sbr_reset when rnd == NULL, exposed for tests: out [N][48]. */
int sbr_draw_normals(sbr_env* env, uint64_t seed, double* out, void* stream);

/* the scenario each env gets from sbr_reset(seed, scenario = NULL) when cfg.random_scenario = 1: out [N] int32, DEVICE
 * pointer (replaces np.random.choice(8, 1), gym_SBR_env4.py:107). */
int sbr_draw_scenarios(sbr_env* env, uint64_t seed, int32_t* out, void* stream);

/* Block the calling host thread until everything queued on `stream` has finished (hipStreamSynchronize on the handle's
 * device).  For host-driven callers: the reference's step() returns finished numbers, so the reference-shaped single env
 * reads its pinned-host outputs after sbr_step + sbr_synchronize - two C calls per step, no other runtime binding needed. */
int sbr_synchronize(sbr_env* env, void* stream);

/* timing helper for bench.py: average device time (ms) per sbr_step launch between two marks,
 * measured with HIP events on `stream` (the stream the kernels are launched on). */
int sbr_timer_start(sbr_env* env, void* stream);
int sbr_timer_stop(sbr_env* env, void* stream, float* elapsed_ms);

#ifdef __cplusplus
}
#endif
#endif /* SBR_AMD_H */
