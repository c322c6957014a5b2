// fp64_partial_exec.hip - does a wave64 float64 VALU instruction cost less when part of the wavefront is masked off?
// The 16-lane DP unit takes four passes over 64 lanes; if passes whose 16 lanes are all inactive were skipped, batches that
// cannot fill the chip (configs[1]: 4096 envs = 64 waves; configs[3]'s 32768 envs per GPU = 512 waves on 1024 SIMDs) could be
// spread as half-filled waves over twice as many SIMDs and issue their RK4 loops twice as fast.  Measured here: cycles per
// v_fma_f64 of one wave per SIMD with 64 / 48 / 32 / 16 / 1 active lanes (lanes >= ACTIVE leave the loop's EXEC mask).
// build + run on the GPU box: hipcc --offload-arch=gfx950 -O3 -o /tmp/pe scripts/probes/fp64_partial_exec.hip && /tmp/pe
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
#define ITER 2000
#define UNR 32
template <int ACTIVE>
__global__ __launch_bounds__(64) void k(double a, double b, double* out, unsigned long long* cyc) {
    double x[8];
    for (int j = 0; j < 8; ++j) x[j] = a + threadIdx.x * 1e-9 + j * 1e-3;
    unsigned long long c0 = 0, c1 = 0;
    if ((int)threadIdx.x < ACTIVE) {                    // the loop below runs with EXEC = the low ACTIVE lanes
        c0 = __builtin_amdgcn_s_memtime();
        for (int it = 0; it < ITER; ++it) {
#pragma unroll
            for (int u = 0; u < UNR; ++u) x[u & 7] = __builtin_fma(x[u & 7], a, b);
        }
        c1 = __builtin_amdgcn_s_memtime();
    }
    double s = 0;
    for (int j = 0; j < 8; ++j) s += x[j];
    out[blockIdx.x * 64 + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = c1 - c0;
}
template <int ACTIVE>
void run(int waves) {
    double* out; unsigned long long* cyc;
    (void)hipMalloc(&out, 8 * 64 * waves); (void)hipMalloc(&cyc, 8 * waves);
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(k<ACTIVE>, dim3(waves), dim3(64), 0, 0, 1.0000001, 1e-7, out, cyc);
    (void)hipDeviceSynchronize();
    std::vector<unsigned long long> c(waves);
    (void)hipMemcpy(c.data(), cyc, 8 * waves, hipMemcpyDeviceToHost);
    std::sort(c.begin(), c.end());
    printf("%4d waves, %2d active lanes: %.2f cycles per v_fma_f64 (median wave)\n", waves, ACTIVE, c[waves / 2] / ((double)ITER * UNR));
    (void)hipFree(out); (void)hipFree(cyc);
}
int main() {
    for (int waves : {64, 1024}) { run<64>(waves); run<48>(waves); run<32>(waves); run<16>(waves); run<1>(waves); }
    return 0;
}
