// fma_banks.hip - v_fma_f64 with explicit registers: does its issue cost depend on which VGPRs the three operands sit in
// (register-file banks), on the encoding (VOP3 v_fma_f64 against VOP2 v_fmac_f64), or only on how many VGPR pairs are read?
// Eight independent chains, 32 instructions per loop body, one wave per SIMD (256 workgroups of 256) and two.
// build + run on the GPU box: hipcc --offload-arch=gfx950 -O3 -w -o /tmp/fma_banks scripts/probes/fma_banks.hip && /tmp/fma_banks
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#include <string>

#define ITER 4000

// chain j: accumulator v[D+2j : D+2j+1]; the other operands at B+2j / C+2j (per-chain) or fixed registers
#define STR2(x) #x
#define STR(x) STR2(x)
#define R(base, j) "v[" STR(base) "+" STR(j) "*2:" STR(base) "+" STR(j) "*2+1]"

#define FMA3(D, B, C, j) "v_fma_f64 " R(D, j) ", " R(D, j) ", " R(B, j) ", " R(C, j) "\n"
#define FMAC(D, B, C, j) "v_fmac_f64_e32 " R(D, j) ", " R(B, j) ", " R(C, j) "\n"
#define MUL2(D, B, C, j) "v_mul_f64 " R(D, j) ", " R(D, j) ", " R(B, j) "\n"
#define FMAS(D, B, C, j) "v_fma_f64 " R(D, j) ", " R(D, j) ", s[4:5], " R(C, j) "\n"
#define FMAK(D, B, C, j) "v_fma_f64 " R(D, j) ", " R(D, j) ", " R(B, j) ", 0.5\n"
#define EIGHT(OP, D, B, C) OP(D, B, C, 0) OP(D, B, C, 1) OP(D, B, C, 2) OP(D, B, C, 3) OP(D, B, C, 4) OP(D, B, C, 5) OP(D, B, C, 6) OP(D, B, C, 7)
#define BODY(OP, D, B, C) EIGHT(OP, D, B, C) EIGHT(OP, D, B, C) EIGHT(OP, D, B, C) EIGHT(OP, D, B, C)

#define CLOB "v64","v65","v66","v67","v68","v69","v70","v71","v72","v73","v74","v75","v76","v77","v78","v79","v80","v81","v82","v83", \
    "v84","v85","v86","v87","v88","v89","v90","v91","v92","v93","v94","v95","v96","v97","v98","v99","v100","v101","v102","v103", \
    "v104","v105","v106","v107","v108","v109","v110","v111","v112","v113","v114","v115","v116","v117","v118","v119","v120","v121", \
    "v122","v123","v124","v125","v126","v127","s4","s5"

// all registers v64..v127 are set to 1.0 + tiny before the loop (the chains then stay finite: x <- x * 1 + tiny or x * 1)
#define KERNEL(NAME, OP, D, B, C)                                                                                         \
__global__ __launch_bounds__(256) void NAME(double* out, unsigned long long* cyc, unsigned long long* rt) {                \
    unsigned long long r0, r1, c0, c1;                                                                                     \
    asm volatile("s_mov_b32 s4, 0\n s_mov_b32 s5, 0x3ff00000\n" ::: "s4", "s5");                                          \
    asm volatile(                                                                                                          \
        ".set i, 64\n.rept 32\n v_mov_b32 v[i], 0\n v_mov_b32 v[i+1], 0x3ff00000\n .set i, i+2\n.endr\n" ::: CLOB);       \
    r0 = __builtin_amdgcn_s_memrealtime(); c0 = __builtin_amdgcn_s_memtime();                                              \
    for (int it = 0; it < ITER; ++it) asm volatile(BODY(OP, D, B, C) ::: CLOB);                                            \
    c1 = __builtin_amdgcn_s_memtime(); r1 = __builtin_amdgcn_s_memrealtime();                                              \
    double s; asm volatile("v_add_f64 %0, v[64:65], v[80:81]" : "=v"(s) :: CLOB);                                          \
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;                                                                        \
    if ((threadIdx.x & 63) == 0) { cyc[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = c1 - c0; rt[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = r1 - r0; } \
}

// register bases: 64 = bank pair {0,1}; 66 = {2,3}.  Chains advance by 2 registers, so chain j alternates bank pairs
// unless all three bases advance together - which they do: what matters is the RELATIVE alignment of D, B, C.
KERNEL(k_fma3_same,  FMA3, 64, 80, 96)     // D, B, C all in the same bank pair (bases = 0 mod 4)
KERNEL(k_fma3_b_off, FMA3, 64, 82, 96)     // B in the other bank pair
KERNEL(k_fma3_c_off, FMA3, 64, 80, 98)     // C in the other bank pair
KERNEL(k_fma3_bc_off, FMA3, 64, 82, 98)    // B and C in the other pair than D
KERNEL(k_fmac_same,  FMAC, 64, 80, 96)
KERNEL(k_fmac_b_off, FMAC, 64, 82, 96)
KERNEL(k_fmac_bc_off, FMAC, 64, 82, 98)
KERNEL(k_mul_same,   MUL2, 64, 80, 96)
KERNEL(k_mul_off,    MUL2, 64, 82, 96)
KERNEL(k_fmas_same,  FMAS, 64, 80, 96)     // x * SGPR + VGPR
KERNEL(k_fmas_off,   FMAS, 64, 80, 98)
KERNEL(k_fmak_same,  FMAK, 64, 80, 96)     // x * VGPR + inline constant
KERNEL(k_fmak_off,   FMAK, 64, 82, 96)

typedef void (*kern_t)(double*, unsigned long long*, unsigned long long*);
void run(const char* name, kern_t f, int blocks) {
    const int waves = blocks * 4;
    double* out; unsigned long long *cyc, *rt;
    hipMalloc(&out, sizeof(double) * blocks * 256); hipMalloc(&cyc, 8 * waves); hipMalloc(&rt, 8 * waves);
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(f, dim3(blocks), dim3(256), 0, 0, out, cyc, rt);
    hipDeviceSynchronize();
    std::vector<unsigned long long> c(waves), r(waves);
    hipMemcpy(c.data(), cyc, 8 * waves, hipMemcpyDeviceToHost); hipMemcpy(r.data(), rt, 8 * waves, hipMemcpyDeviceToHost);
    std::sort(c.begin(), c.end()); std::sort(r.begin(), r.end());
    const double n = (double)ITER * 32;
    printf("%-46s %d wave(s)/SIMD: %.2f cycles, %.3f ns per instruction and wave (%.2f GHz)\n", name, blocks / 256,
           c[waves / 2] / n, r[waves / 2] * 10.0 / n, c[waves / 2] / (r[waves / 2] * 10.0));
    hipFree(out); hipFree(cyc); hipFree(rt);
}

int main() {
    for (int blocks : {256, 512}) {
        run("v_fma_f64  D,B,C same bank pair", k_fma3_same, blocks);
        run("v_fma_f64  B in the other pair", k_fma3_b_off, blocks);
        run("v_fma_f64  C in the other pair", k_fma3_c_off, blocks);
        run("v_fma_f64  B and C in the other pair", k_fma3_bc_off, blocks);
        run("v_fmac_f64 D,B,C same bank pair", k_fmac_same, blocks);
        run("v_fmac_f64 B in the other pair", k_fmac_b_off, blocks);
        run("v_fmac_f64 B and C in the other pair", k_fmac_bc_off, blocks);
        run("v_mul_f64  D,B same bank pair", k_mul_same, blocks);
        run("v_mul_f64  B in the other pair", k_mul_off, blocks);
        run("v_fma_f64  x*SGPR+VGPR, same pair", k_fmas_same, blocks);
        run("v_fma_f64  x*SGPR+VGPR, other pair", k_fmas_off, blocks);
        run("v_fma_f64  x*VGPR+0.5, same pair", k_fmak_same, blocks);
        run("v_fma_f64  x*VGPR+0.5, other pair", k_fmak_off, blocks);
    }
    return 0;
}
