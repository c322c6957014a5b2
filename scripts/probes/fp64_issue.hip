// fp64_issue.hip - what one wave per SIMD gets out of the fp64 VALU on gfx950: issue cost and dependent latency of
// v_fma_f64 / v_mul_f64 / v_add_f64, the price of v_rcp_f64 and of a float32-seeded reciprocal, and the shader clock the
// chip holds while every SIMD runs such a loop (s_memtime cycles per s_memrealtime tick).
// build: hipcc --offload-arch=gfx950 -O3 -o build/fp64_issue scripts/probes/fp64_issue.hip ; run: build/fp64_issue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

#define ITER 2000
#define UNR 32

template <int TEST>
__global__ __launch_bounds__(256) void k(double a, double b, double* out, unsigned long long* cyc, unsigned long long* rt) {
    double x[8];
    for (int j = 0; j < 8; ++j) x[j] = a + threadIdx.x * 1e-9 + j * 1e-3;
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    const unsigned long long c0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            if (TEST == 0) x[0] = __builtin_fma(x[0], a, b);                                    // 1 dependent chain
            if (TEST == 1) x[u & 1] = __builtin_fma(x[u & 1], a, b);                            // 2 chains
            if (TEST == 2) x[u & 3] = __builtin_fma(x[u & 3], a, b);                            // 4 chains
            if (TEST == 3) x[u & 7] = __builtin_fma(x[u & 7], a, b);                            // 8 chains
            if (TEST == 4) x[u & 7] = x[u & 7] * a;                                             // mul, 8 chains
            if (TEST == 5) x[u & 7] = x[u & 7] + b;                                             // add, 8 chains
            if (TEST == 6) x[u & 7] = __builtin_amdgcn_rcp(x[u & 7]);                           // v_rcp_f64, 8 chains
            if (TEST == 7) x[u & 7] = (double)__builtin_amdgcn_rcpf((float)x[u & 7]);           // cvt + v_rcp_f32 + cvt
            if (TEST == 8) {                                                                     // sbr_rcp, 8 chains
                const double d = x[u & 7], r = __builtin_amdgcn_rcp(d), e = __builtin_fma(-d, r, 1.0);
                x[u & 7] = __builtin_fma(r, __builtin_fma(e, e, e), r);
            }
            if (TEST == 9) {                                                                     // float32-seeded, 8 chains
                const double d = x[u & 7], r = (double)__builtin_amdgcn_rcpf((float)d), e = __builtin_fma(-d, r, 1.0);
                x[u & 7] = __builtin_fma(r, __builtin_fma(e, e, e), r);
            }
            if (TEST == 10) x[0] = x[0] * a;                                                    // mul, dependent
            if (TEST == 11) x[0] = x[0] + b;                                                    // add, dependent
        }
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime();
    const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    double s = 0;
    for (int j = 0; j < 8; ++j) s += x[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) {
        cyc[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = c1 - c0;
        rt[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = r1 - r0;
    }
}

template <int TEST>
void run(const char* name, int insts_per_unr, int blocks) {
    const int waves = blocks * 4;
    double* out; unsigned long long *cyc, *rt;
    hipMalloc(&out, sizeof(double) * blocks * 256); hipMalloc(&cyc, 8 * waves); hipMalloc(&rt, 8 * waves);
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(k<TEST>, dim3(blocks), dim3(256), 0, 0, 1.0000001, 1e-7, out, cyc, rt);
    hipDeviceSynchronize();
    std::vector<unsigned long long> c(waves), r(waves);
    hipMemcpy(c.data(), cyc, 8 * waves, hipMemcpyDeviceToHost); hipMemcpy(r.data(), rt, 8 * waves, hipMemcpyDeviceToHost);
    std::sort(c.begin(), c.end()); std::sort(r.begin(), r.end());
    const double n = (double)ITER * UNR * insts_per_unr;
    printf("%-34s %2d waves/CU-set  cycles/instr %.2f   ns/instr %.3f   clock %.2f GHz\n", name, blocks / 256 * 4,
           c[waves / 2] / n, r[waves / 2] * 10.0 / n, c[waves / 2] / (r[waves / 2] * 10.0));
    hipFree(out); hipFree(cyc); hipFree(rt);
}

int main() {
    for (int blocks : {256, 512}) {          // one and two waves per SIMD
        run<0>("fma_f64, 1 dependent chain", 1, blocks);
        run<1>("fma_f64, 2 chains", 1, blocks);
        run<2>("fma_f64, 4 chains", 1, blocks);
        run<3>("fma_f64, 8 chains", 1, blocks);
        run<4>("mul_f64, 8 chains", 1, blocks);
        run<5>("add_f64, 8 chains", 1, blocks);
        run<10>("mul_f64, dependent", 1, blocks);
        run<11>("add_f64, dependent", 1, blocks);
        run<6>("v_rcp_f64, 8 chains", 1, blocks);
        run<7>("cvt+rcp_f32+cvt (3 instr), 8 chains", 1, blocks);
        run<8>("sbr_rcp (rcp_f64 + 3 fma)", 1, blocks);
        run<9>("f32-seeded rcp (3 + 3 fma)", 1, blocks);
    }
    return 0;
}
