// fma_forms.hip - does the issue cost of v_fma_f64 (5.2 cycles with one wave per SIMD, against 4.3 for v_mul_f64) depend on
// where its operands come from?  Eight independent chains each; operands: VGPR only, one SGPR, inline constants, the
// accumulating form, and a 1:1 mix of multiplies and fused multiply-adds.
// build + run on the GPU box: hipcc --offload-arch=gfx950 -O3 -o /tmp/fma_forms scripts/probes/fma_forms.hip && /tmp/fma_forms
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

#define ITER 2000
#define UNR 32

template <int TEST>
__global__ __launch_bounds__(256) void k(double a, double b, double* out, unsigned long long* rt) {
    double x[8], y[8], z[8];
    for (int j = 0; j < 8; ++j) { x[j] = a + threadIdx.x * 1e-9 + j * 1e-3; y[j] = 1.0 + threadIdx.x * 1e-12 + j * 1e-9; z[j] = b + threadIdx.x * 1e-13; }
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const int j = u & 7;
            if (TEST == 0) x[j] = __builtin_fma(x[j], a, z[j]);            // VGPR * SGPR + VGPR
            if (TEST == 1) x[j] = __builtin_fma(x[j], y[j], z[j]);         // three different VGPR pairs
            if (TEST == 2) x[j] = __builtin_fma(x[j], 2.0, 0.5);           // inline constants
            if (TEST == 3) x[j] = __builtin_fma(y[j], z[j], x[j]);         // accumulating form (v_fmac)
            if (TEST == 4) x[j] = __builtin_fma(x[j], x[j], x[j]);         // one VGPR pair read three times
            if (TEST == 5) { if (u & 8) x[j] = __builtin_fma(x[j], y[j], z[j]); else x[j] = x[j] * y[j]; }   // 1:1 mix
            if (TEST == 6) x[j] = x[j] * y[j];                              // mul, two VGPR pairs
            if (TEST == 7) x[j] = __builtin_fma(a, b, x[j]);               // (folded by the compiler to an add of a*b? check)
        }
    }
    const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    double s = 0;
    for (int j = 0; j < 8; ++j) s += x[j] + y[j] + z[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) rt[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = r1 - r0;
}

template <int TEST>
void run(const char* name, int blocks) {
    const int waves = blocks * 4;
    double* out; unsigned long long* rt;
    hipMalloc(&out, sizeof(double) * blocks * 256); hipMalloc(&rt, 8 * waves);
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(k<TEST>, dim3(blocks), dim3(256), 0, 0, 1.0000001, 1e-7, out, rt);
    hipDeviceSynchronize();
    std::vector<unsigned long long> r(waves);
    hipMemcpy(r.data(), rt, 8 * waves, hipMemcpyDeviceToHost);
    std::sort(r.begin(), r.end());
    printf("%-44s %d wave(s) per SIMD: %.3f ns per instruction and wave\n", name, blocks / 256, r[waves / 2] * 10.0 / ((double)ITER * UNR));
    hipFree(out); hipFree(rt);
}

int main() {
    for (int blocks : {256, 512}) {
        run<0>("fma  VGPR * SGPR + VGPR", blocks);
        run<1>("fma  VGPR * VGPR + VGPR (three pairs)", blocks);
        run<2>("fma  VGPR * 2.0 + 0.5 (inline constants)", blocks);
        run<3>("fma  accumulating: VGPR * VGPR + dst", blocks);
        run<4>("fma  x * x + x (one pair)", blocks);
        run<5>("1:1 mix of mul and fma (VGPR operands)", blocks);
        run<6>("mul  VGPR * VGPR", blocks);
    }
    return 0;
}
