/* c_abi_demo.c - the drop-in boundary used from plain C: no Python, no torch, no C++.
 *
 * One SBROS-v1 episode of the reference's own configuration (gym_SBR_oneshot.py: scenario 6, numpy seed 0, constant action
 * [2.0, 5.0] - the golden episode tests/golden/sbros_const_2_5.npz) for N replicas through libsbr_amd.so: sbr_create,
 * sbr_reset with the episode's flow-weighted influent, 463 x sbr_step, and prints what the reference prints in its place
 * (SURVEY.md 8c anchors: reward of calls 1, 2, 100, Kla and So at call 100, the episode return, the wastage flow Qw).
 * tests/test_gpu_parity.py builds and runs it on the GPU box and compares the printed numbers with the reference's.
 *
 *   gcc -O2 -D__HIP_PLATFORM_AMD__ -I include -I /opt/rocm/include examples/c_abi_demo.c \
 *       -L gym_sbr2_amd/lib -lsbr_amd -L /opt/rocm/lib -lamdhip64 -Wl,-rpath,$PWD/gym_sbr2_amd/lib -Wl,-rpath,/opt/rocm/lib -o /tmp/c_abi_demo
 */
#include <hip/hip_runtime_api.h>
#include <stdio.h>
#include <stdlib.h>

#include "sbr_amd.h"

#define N 192 /* three wavefronts of identical reactors: every one must give the same numbers */
#define CHECK_HIP(x) do { hipError_t s_ = (x); if (s_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(s_)); return 2; } } while (0)
#define CHECK_SBR(x) do { int s_ = (x); if (s_ != 0) { fprintf(stderr, "%s: %d %s\n", #x, s_, sbr_last_error(env)); return 3; } } while (0)

/* influent_mixed of the episode (buffer_tank3.py:87-107 with np.random.seed(0)); entry 0 (the inflow) is set by the library */
static const double kInfluent[SBR_NX] = {0.0, 30.00000000000001, 71.10697236709676, 42.07346176190449, 170.62242761445307,
                                         23.632836640085962, 0.0, 0.0, 0.0, 0.0, 50.01411289944916, 10.666046512878294,
                                         13.326754715825064, 7.000000000000002};

int main(void) {
    sbr_env* env = NULL;
    sbr_config cfg;
    if (sbr_device_count() < 1) { fprintf(stderr, "no HIP device: this library has no CPU path\n"); return 1; }
    sbr_default_config(&cfg);
    cfg.out_f64 = 1; cfg.act_f64 = 1;                     /* float64 in and out, like the reference's step() */
    if (sbr_create(N, 0, 0, &cfg, &env) != 0) { fprintf(stderr, "sbr_create: %s\n", sbr_last_error(NULL)); return 1; }

    double *d_infl, *d_act, *d_obs, *d_state, *d_reward, *d_ctrl;
    uint8_t* d_done;
    static double h_infl[N * SBR_NX], h_act[N * 2], h_reward[N], h_ctrl[SBR_NCTRL * N], h_state[N * SBR_NSTATE];
    static uint8_t h_done[N];
    for (int i = 0; i < N; ++i) {
        for (int j = 0; j < SBR_NX; ++j) h_infl[i * SBR_NX + j] = kInfluent[j];
        h_act[2 * i] = 2.0; h_act[2 * i + 1] = 5.0;
    }
    CHECK_HIP(hipMalloc((void**)&d_infl, sizeof h_infl)); CHECK_HIP(hipMalloc((void**)&d_act, sizeof h_act));
    CHECK_HIP(hipMalloc((void**)&d_obs, sizeof(double) * N * SBR_NOBS)); CHECK_HIP(hipMalloc((void**)&d_state, sizeof h_state));
    CHECK_HIP(hipMalloc((void**)&d_reward, sizeof h_reward)); CHECK_HIP(hipMalloc((void**)&d_done, sizeof h_done));
    CHECK_HIP(hipMalloc((void**)&d_ctrl, sizeof h_ctrl));
    CHECK_HIP(hipMemcpy(d_infl, h_infl, sizeof h_infl, hipMemcpyHostToDevice));
    CHECK_HIP(hipMemcpy(d_act, h_act, sizeof h_act, hipMemcpyHostToDevice));

    CHECK_SBR(sbr_reset(env, 0, NULL, NULL, d_infl, NULL, d_obs, NULL));
    int calls = 0;
    double ret = 0.0;
    for (;;) {
        CHECK_SBR(sbr_step(env, d_act, d_obs, d_state, d_reward, d_done, NULL));
        CHECK_SBR(sbr_synchronize(env, NULL));
        CHECK_HIP(hipMemcpy(h_reward, d_reward, sizeof h_reward, hipMemcpyDeviceToHost));
        CHECK_HIP(hipMemcpy(h_done, d_done, sizeof h_done, hipMemcpyDeviceToHost));
        ++calls;
        ret += h_reward[0];
        for (int i = 1; i < N; ++i)
            if (h_reward[i] != h_reward[0] || h_done[i] != h_done[0]) { fprintf(stderr, "replica %d differs at call %d\n", i, calls); return 4; }
        if (calls == 1 || calls == 2) printf("reward[%d] %.17g\n", calls, h_reward[0]);
        if (calls == 100) {
            CHECK_SBR(sbr_get_state(env, NULL, d_ctrl, NULL));
            CHECK_HIP(hipMemcpy(h_ctrl, d_ctrl, sizeof h_ctrl, hipMemcpyDeviceToHost));
            CHECK_HIP(hipMemcpy(h_state, d_state, sizeof h_state, hipMemcpyDeviceToHost));
            printf("reward[100] %.17g\nKla[100] %.17g\nSo[100] %.17g\n", h_reward[0], h_ctrl[SBR_C_KLA_LAST * N], h_state[9] * 8.0);
        }
        if (h_done[0] || calls > 1000) break;
    }
    CHECK_SBR(sbr_get_state(env, NULL, d_ctrl, NULL));
    CHECK_HIP(hipMemcpy(h_ctrl, d_ctrl, sizeof h_ctrl, hipMemcpyDeviceToHost));
    printf("calls %d\nreturn %.17g\nreturn_row %.17g\nQw %.17g\n", calls, ret, h_ctrl[SBR_C_RETURN * N], h_ctrl[SBR_C_QW * N]);
    CHECK_SBR(sbr_destroy(env));
    hipFree(d_infl); hipFree(d_act); hipFree(d_obs); hipFree(d_state); hipFree(d_reward); hipFree(d_done); hipFree(d_ctrl);
    return 0;
}
