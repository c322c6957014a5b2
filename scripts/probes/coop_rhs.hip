// coop_rhs.hip - does splitting ONE environment over several lanes pay for this right-hand side?  (SURVEY.md section 7:
// "one wavefront per env ... decide by measurement"; BASELINE.json north_star.)
//
// Two kernels advance the same 4096 environments through the same chain of dependent RHS evaluations (Euler steps
// x <- x + h f(x) on the nine feedback components: the dependency structure of the RK4 stages) and are timed per evaluation:
//   lane   one lane per env, the product's sbr_rates arithmetic (51 instructions + one v_rcp_f64): 64 wavefronts
//   quad   four lanes per env (256 wavefronts).  Lanes of a wavefront execute ONE instruction stream, so work can only be
//          split where the four lanes do the same operation on different operands.  In this RHS that is the reciprocals:
//          lane q takes 1/(u_q v_q) for (u, v) = (d1, d2), (d4, d5), (d3, 1), (d6, d2) with per-lane constants - four
//          reciprocals in the time of one, no batch inversion (5 + 8 multiplications gone) - and the four results are
//          broadcast inside the quad (DPP quad_perm, two 32-bit moves per double).  Rates, derivatives and the update are
//          then done by every lane (replicated): splitting those too would need every stage value broadcast again (9 values
//          x 2 moves per stage), more than the 12 instructions per stage it could save.
// Also timed: the cross-lane primitives themselves (DPP quad broadcast, ds_bpermute, v_readlane of a double).
// build + run on the GPU box: hipcc --offload-arch=gfx950 -O3 -o /tmp/coop_rhs scripts/probes/coop_rhs.hip && /tmp/coop_rhs
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>

#define DEV __device__ __forceinline__
struct Par {
    double f1a, f1b, f2a, f2b, f4a, f4b, Kno, Koa, Kx, KohEtag, etah_g, bA_bH, bH, bA, ka;
    double n2_12, n4_45b, n8_1, n8_3, n9_23, n10_12, n10_3, n12_45b, So_sat;
};
DEV double rcp(double d) {
    const double r = __builtin_amdgcn_rcp(d), e = __builtin_fma(-d, r, 1.0);
    return __builtin_fma(r, __builtin_fma(e, e, e), r);
}
enum { SS, XS, XBH, XBA, SO, SNO, SNH, SND, XND, NA };

// everything after the four reciprocals (shared by both kernels)
DEV void finish(const Par& p, const double (&a)[NA], double rA, double rB, double rc, double rfb, double kla, double (&k)[NA]) {
    const double G = (a[SS] * a[XBH]) * rA, rho1 = G * a[SO], kw = p.KohEtag * (a[SNO] * rc), rho2 = G * kw;
    const double c7 = (rfb * __builtin_fma(p.etah_g, kw, a[SO])) * a[XBH], rho7 = a[XS] * c7, rho8 = a[XND] * c7;
    const double rho3 = ((a[SNH] * a[SO]) * rB) * a[XBA], s45b = __builtin_fma(p.bA_bH, a[XBA], a[XBH]), z = a[SND] * a[XBH];
    const double s12 = rho1 + rho2;
    k[SS] = __builtin_fma(p.n2_12, s12, rho7); k[XS] = __builtin_fma(p.n4_45b, s45b, -rho7);
    k[XBH] = __builtin_fma(-p.bH, a[XBH], s12); k[XBA] = __builtin_fma(-p.bA, a[XBA], rho3);
    k[SO] = __builtin_fma(p.n8_1, rho1, __builtin_fma(p.n8_3, rho3, __builtin_fma(-kla, a[SO], kla * p.So_sat)));
    k[SNO] = __builtin_fma(p.n9_23, rho2, rho3); k[SNH] = __builtin_fma(p.n10_12, s12, __builtin_fma(p.n10_3, rho3, p.ka * z));
    k[SND] = __builtin_fma(-p.ka, z, rho8); k[XND] = __builtin_fma(p.n12_45b, s45b, -rho8);
}

__global__ __launch_bounds__(256) void k_lane(Par p, int n, int iters, double h, const double* __restrict__ x0, double* __restrict__ out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    double a[NA], k[NA];
    for (int j = 0; j < NA; ++j) a[j] = x0[j * n + i];
    for (int it = 0; it < iters; ++it) {
        const double d1 = __builtin_fma(a[SS], p.f1a, p.f1b), d2 = __builtin_fma(a[SO], p.f2a, p.f2b), d3 = p.Kno + a[SNO];
        const double d4 = __builtin_fma(a[SNH], p.f4a, p.f4b), d5 = p.Koa + a[SO], d6 = __builtin_fma(p.Kx, a[XBH], a[XS]);
        const double A = d1 * d2, B = d4 * d5, Cc = d3 * d6, AB = A * B, R = rcp(AB * Cc), rC = R * AB, rAB = R * Cc;
        finish(p, a, rAB * B, rAB * A, rC * d6, (rC * d3) * ((rAB * B) * d1), 100.0, k);
#pragma unroll
        for (int j = 0; j < NA; ++j) a[j] = __builtin_fma(h, k[j], a[j]);
    }
    for (int j = 0; j < NA; ++j) out[j * n + i] = a[j];
}

// one double from lane `src` of every quad to all four lanes: two v_mov_b32_dpp quad_perm
template <int SRC>
DEV double quad_bcast(double v) {
    const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), SRC * 0x55, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), SRC * 0x55, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}

__global__ __launch_bounds__(256) void k_quad(Par p, int n, int iters, double h, const double* __restrict__ x0, double* __restrict__ out) {
    const int t = blockIdx.x * 256 + threadIdx.x, i = t >> 2, q = t & 3;
    if (i >= n) return;
    double a[NA], k[NA];
    for (int j = 0; j < NA; ++j) a[j] = x0[j * n + i];
    // per-lane constants of u_q = s_u ca + (cb or s_w), v_q = s_v cc + cd
    const double ca = q == 0 ? p.f1a : q == 1 ? p.f4a : q == 2 ? 1.0 : p.Kx, cb = q == 0 ? p.f1b : q == 1 ? p.f4b : q == 2 ? p.Kno : 0.0;
    const double cc = q == 0 ? p.f2a : q == 1 ? 1.0 : q == 2 ? 0.0 : p.f2a, cd = q == 0 ? p.f2b : q == 1 ? p.Koa : q == 2 ? 1.0 : p.f2b;
    for (int it = 0; it < iters; ++it) {
        const double su = q == 0 ? a[SS] : q == 1 ? a[SNH] : q == 2 ? a[SNO] : a[XBH];      // 3 x 2 v_cndmask
        const double u = __builtin_fma(su, ca, q == 3 ? a[XS] : cb);                          // 2 v_cndmask
        const double v = __builtin_fma(a[SO], cc, cd);
        const double r = rcp(u * v);
        finish(p, a, quad_bcast<0>(r), quad_bcast<1>(r), quad_bcast<2>(r), quad_bcast<3>(r), 100.0, k);
#pragma unroll
        for (int j = 0; j < NA; ++j) a[j] = __builtin_fma(h, k[j], a[j]);
    }
    if (q == 0) for (int j = 0; j < NA; ++j) out[j * n + i] = a[j];
}

// cross-lane primitives: a chain of `iters` x 16 exchanges of one double
template <int KIND>
__global__ __launch_bounds__(256) void k_xlane(int iters, double* __restrict__ out) {
    double v = threadIdx.x * 1.0 + 0.5;
    const int lane = threadIdx.x & 63;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            if (KIND == 0) v = quad_bcast<1>(v) + 1.0;
            if (KIND == 1) {
                const int idx = ((lane + 1) & 63) << 2;
                v = __hiloint2double(__builtin_amdgcn_ds_bpermute(idx, __double2hiint(v)), __builtin_amdgcn_ds_bpermute(idx, __double2loint(v))) + 1.0;
            }
            if (KIND == 2) v = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), 5), __builtin_amdgcn_readlane(__double2loint(v), 5)) + 1.0;
            if (KIND == 3) v = v + 1.0;
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = v;
}

int main() {
    const int n = 4096, iters = 4000;
    Par p;
    const double muH = 4, Ks = 10, Koh = 0.2, Kno = 0.5, bH = 0.3, eta_g = 0.8, eta_h = 0.8, kh = 3, Kx = 0.1, muA = 0.5, Knh = 1, bA = 0.05, Koa = 0.4,
                 ka = 0.05, Ya = 0.24, Yh = 0.67, ixb = 0.08, ixp = 0.06, fp = 0.08;
    p.f1a = kh / muH; p.f1b = Ks * kh / muH; p.f2a = 1 / kh; p.f2b = Koh / kh; p.f4a = 1 / muA; p.f4b = Knh / muA; p.Kno = Kno; p.Koa = Koa; p.Kx = Kx;
    p.KohEtag = Koh * eta_g; p.etah_g = eta_h / eta_g; p.bA_bH = bA / bH; p.bH = bH; p.bA = bA; p.ka = ka;
    p.n2_12 = -1 / Yh; p.n4_45b = (1 - ixp) * bH; p.n8_1 = -(1 - Yh) / Yh; p.n8_3 = -(4.57 - Ya) / Ya;
    p.n9_23 = -((1 - Yh) / (2.86 * Yh)) * Ya; p.n10_12 = -ixb; p.n10_3 = -ixb - 1 / Ya; p.n12_45b = (ixb - fp * ixp) * bH; p.So_sat = 8;
    std::vector<double> x0(NA * n), a(NA * n), b(NA * n);
    const double base[NA] = {10, 96, 1250, 79, 0.5, 1.4, 27, 3.6, 7.3};
    for (int j = 0; j < NA; ++j) for (int i = 0; i < n; ++i) x0[j * n + i] = base[j] * (1 + 0.1 * std::sin(0.37 * i + j));
    double *dx, *da, *db;
    hipMalloc(&dx, 8 * NA * n); hipMalloc(&da, 8 * NA * n); hipMalloc(&db, 8 * NA * n * 4);
    hipMemcpy(dx, x0.data(), 8 * NA * n, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const double h = 1e-7;
    float ms_lane = 0, ms_quad = 0, ms;
    for (int rep = 0; rep < 4; ++rep) {     // the last repetition counts (steady clocks)
        hipEventRecord(e0); hipLaunchKernelGGL(k_lane, dim3(n / 256), dim3(256), 0, 0, p, n, iters, h, dx, da); hipEventRecord(e1);
        hipEventSynchronize(e1); hipEventElapsedTime(&ms_lane, e0, e1);
        hipEventRecord(e0); hipLaunchKernelGGL(k_quad, dim3(4 * n / 256), dim3(256), 0, 0, p, n, iters, h, dx, db); hipEventRecord(e1);
        hipEventSynchronize(e1); hipEventElapsedTime(&ms_quad, e0, e1);
    }
    hipMemcpy(a.data(), da, 8 * NA * n, hipMemcpyDeviceToHost); hipMemcpy(b.data(), db, 8 * NA * n, hipMemcpyDeviceToHost);
    double worst = 0;
    for (int j = 0; j < NA * n; ++j) worst = fmax(worst, fabs(a[j] - b[j]) / (fabs(a[j]) + 1e-300));
    printf("%d envs, %d dependent RHS evaluations each\n", n, iters);
    printf("  lane per env  (%4d waves): %8.1f ns per evaluation\n", n / 64, ms_lane * 1e6 / iters);
    printf("  quad per env  (%4d waves): %8.1f ns per evaluation   -> %.2fx of the lane kernel's speed; results agree to %.1e relative\n",
           4 * n / 64, ms_quad * 1e6 / iters, ms_lane / ms_quad, worst);
    double* dout; hipMalloc(&dout, 8 * 256 * 256);
    const char* names[4] = {"DPP quad broadcast of a double + add", "ds_bpermute of a double + add", "v_readlane of a double + add", "add alone"};
    for (int kind = 0; kind < 4; ++kind) {
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0);
            if (kind == 0) hipLaunchKernelGGL(k_xlane<0>, dim3(256), dim3(256), 0, 0, 2000, dout);
            if (kind == 1) hipLaunchKernelGGL(k_xlane<1>, dim3(256), dim3(256), 0, 0, 2000, dout);
            if (kind == 2) hipLaunchKernelGGL(k_xlane<2>, dim3(256), dim3(256), 0, 0, 2000, dout);
            if (kind == 3) hipLaunchKernelGGL(k_xlane<3>, dim3(256), dim3(256), 0, 0, 2000, dout);
            hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
        }
        printf("  %-40s %6.2f ns per exchange (one wave per SIMD, dependent chain)\n", names[kind], ms * 1e6 / (2000.0 * 16));
    }
    return 0;
}
