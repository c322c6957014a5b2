// resident_handoff.hip - what does a RESIDENT stepping kernel behind the unchanged sbr_step ABI cost per call?
// ("Next" item 1, stage A.)  Nothing of the product is in here: only the hand-off skeleton is timed.
//
// k_resident has k_step's footprint (256 workgroups x 256 threads, 53 KiB of LDS each, one wave per SIMD) and stays on the GPU:
// per "call" wave 0 of each workgroup polls a mailbox record {epoch, call index} (sc1 loads + s_sleep, HARD time cap: the kernel
// leaves by itself after cap_ms whatever the host does), the workgroup loads one 8-byte action per lane (sc1), optionally does
// `work` x 8 float64 FMAs per lane (stand-in for the PIDs + RK4), writes 136 B of outputs per lane (sc1 write-through, drained),
// and publishes done[wg] = epoch.  Stream order on the CALLER's stream is kept by one of:
//   bell   a 256-thread doorbell kernel: posts the record, polls the 256 done words (bounded), exits  -> one launch per call
//   cp     hipStreamWriteValue32(record) + hipStreamWaitValue32(signal word) executed by the command processor; the last
//          workgroup to arrive (device-scope counter) stores the epoch to the signal word            -> no launch per call
// Reported per variant: the period per call on the caller's stream (HIP events over the whole sequence), the doorbell kernel's
// own post -> all-done time (s_memrealtime, 100 MHz), and a check that every output word carries the action of ITS call
// (visibility in both directions, consumer caches warm).  "empty" is the same doorbell kernel returning at once = the boundary.
// build + run on the GPU box:  hipcc --offload-arch=gfx950 -O3 -o /tmp/rh scripts/probes/resident_handoff.hip && /tmp/rh
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t s_ = (x); if (s_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(s_)); exit(1); } } while (0)
typedef unsigned u32; typedef unsigned long long u64; typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define DEV __device__ __forceinline__
DEV u32 ld1(const u32* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
DEV void st1(u32* p, u32 v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
DEV u64 now() { return __builtin_amdgcn_s_memrealtime(); }
constexpr int WG = 256, NWG = 256, N = WG * NWG, NOUT = 34;
constexpr u32 EXIT = 0xFFFFFFFFu;
struct Mail { u32* rec; u32* done; u32* arrive; u32* sig; u64* act; float* out; u32* err; int nrep, work, use_sig, policy, outputs; };

__global__ __launch_bounds__(WG) void k_resident(Mail m, u64 cap_ticks) {
    __shared__ double park[6656];                        // 53 KiB: k_step's LDS footprint
    __shared__ u32 sh[2];
    const u32 l = threadIdx.x, wg = blockIdx.x;
    park[l] = 0.0;
    const u32* rec = m.rec + (wg % m.nrep) * 32;        // one record per replica, each on its own 128-byte line
    const u64 t0 = now();
    u32 seen = 0;
    for (;;) {
        if (l == 0) {
            u32 e;
            for (;;) {
                e = ld1(rec);
                if (e != seen) break;
                if (now() - t0 > cap_ticks) { e = EXIT; break; }          // every wave leaves by itself
                __builtin_amdgcn_s_sleep(2);
            }
            sh[0] = e; sh[1] = e == EXIT ? 0u : ld1(rec + 1);               // the call index travels with the epoch
        }
        __syncthreads();
        const u32 e = sh[0], call = sh[1];
        __syncthreads();
        if (e == EXIT) {      // also on cap expiry: release every command-processor wait that is still queued behind this kernel
            if (m.use_sig && wg == 0 && l == 0) __hip_atomic_store(m.sig, 0x7FFFFFFFu, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            break;
        }
        seen = e;
        const u64 a = __hip_atomic_load(m.act + (size_t)(call & 1u) * N + wg * WG + l, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        double x[8];
        for (int j = 0; j < 8; ++j) x[j] = (double)(u32)a + j;
        for (int i = 0; i < m.work; ++i)
            for (int j = 0; j < 8; ++j) x[j] = __builtin_fma(x[j], 1.0000001, 0.5);
        // outputs as the product writes them: a wave's 64 rows of 136 B are one contiguous block, stored 16 bytes per lane (sc1)
        const float av = m.work ? (float)(x[0] * 0.0 + x[3] * 0.0) + (float)(u32)a : (float)(u32)a;
        const float wv = __shfl(av, 0, 64);                                   // (the probe's actions are equal across a wave)
        char* wbase = reinterpret_cast<char*>(m.out) + ((size_t)wg * WG + (l & ~63u)) * NOUT * 4;
        if (m.outputs)
            for (int r = 0; r * 64 < 64 * NOUT * 4 / 16; ++r) {
                const int c = r * 64 + (int)(l & 63u);
                if (c < 64 * NOUT * 4 / 16) {
                    u32x4 v4 = {__float_as_uint(wv), __float_as_uint(wv), __float_as_uint(wv), __float_as_uint(wv)};
                    asm volatile("global_store_dwordx4 %0, %1, off sc1" : : "v"(wbase + c * 16), "v"(v4) : "memory");
                }
            }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (l == 0) {
            st1(m.done + wg, e);
            if (m.use_sig && __hip_atomic_fetch_add(m.arrive, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == e * NWG - 1u)
                __hip_atomic_store(m.sig, e, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}
// the "policy": writes the actions of call k with PLAIN stores, as a caller's kernel would
__global__ __launch_bounds__(WG) void k_policy(u64* act, u32 k) { act[(size_t)(k & 1u) * N + blockIdx.x * WG + threadIdx.x] = 1000u + k; }

__global__ __launch_bounds__(WG) void k_bell(Mail m, u32 k, int empty, u32* ticks) {
    const u32 l = threadIdx.x;
    if (empty || *m.err) return;
    const u64 t0 = now();
    if (l < (u32)m.nrep) { st1(m.rec + l * 32 + 1, k); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); st1(m.rec + l * 32, k); }
    int ok = 0;
    for (int spins = 0; spins < 40000; ++spins) {       // bounded: ~20 ms
        ok = __syncthreads_and(ld1(m.done + l) == k);
        if (ok) break;
        __builtin_amdgcn_s_sleep(1);
    }
    // every output word of this lane's env must carry THIS call's action (read back past the caches, like a later kernel would)
    u32 bad = 0;
    const float want = m.policy ? (float)(1000u + k) : 0.0f;
    if (m.outputs)
        for (int j = 0; j < NOUT; j += 11) bad += __hip_atomic_load(m.out + ((size_t)(l * 97 % NWG) * WG + l) * NOUT + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != want;
    if (!ok || bad) atomicAdd(m.err, ok ? 1u << 16 : 1u);
    if (l == 0) ticks[k] = (u32)(now() - t0);
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 2000;
    Mail m{};
    u32* ticks; hipStream_t sr, su; hipEvent_t e0, e1;
    CK(hipMalloc(&m.rec, 256 * 128)); CK(hipMalloc(&m.done, NWG * 4)); CK(hipMalloc(&m.arrive, 4)); CK(hipMalloc(&m.err, 4));
    CK(hipMalloc(&m.act, 2 * N * 8)); CK(hipMalloc(&m.out, (size_t)N * NOUT * 4)); CK(hipMalloc(&ticks, (iters + 2) * 4));
    CK(hipExtMallocWithFlags((void**)&m.sig, 8, hipMallocSignalMemory));
    CK(hipStreamCreateWithFlags(&sr, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&su, hipStreamNonBlocking));
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    struct V { const char* name; int mode, nrep, work, policy, outputs; };       // mode 0 empty bell, 1 bell, 2 cp
    const V vs[] = {{"empty doorbell kernel (boundary)", 0, 1, 0, 0, 0}, {"bell nrep=1   flags only", 1, 1, 0, 0, 0}, {"bell nrep=8   flags only", 1, 8, 0, 0, 0},
                    {"bell nrep=256 flags only", 1, 256, 0, 0, 0}, {"bell nrep=256 work=0   +outputs", 1, 256, 0, 0, 1}, {"bell nrep=8   work=0   +outputs", 1, 8, 0, 0, 1},
                    {"bell nrep=256 work=300 +outputs", 1, 256, 300, 0, 1}, {"bell nrep=256 work=300 +outputs +policy kernel", 1, 256, 300, 1, 1},
                    {"bell nrep=256 work=0   +outputs +policy kernel", 1, 256, 0, 1, 1}, {"cp   nrep=1   flags only", 2, 1, 0, 0, 0},
                    {"cp   nrep=1   work=0   +outputs", 2, 1, 0, 0, 1}, {"cp   nrep=1   work=300 +outputs", 2, 1, 300, 0, 1}};
    for (const V& v : vs) {
        CK(hipMemset(m.rec, 0, 256 * 128)); CK(hipMemset(m.done, 0, NWG * 4)); CK(hipMemset(m.arrive, 0, 4)); CK(hipMemset(m.err, 0, 4));
        CK(hipMemset(m.sig, 0, 8)); CK(hipMemset(ticks, 0, (iters + 2) * 4)); CK(hipMemset(m.act, 0, 2 * N * 8));
        CK(hipDeviceSynchronize());
        m.nrep = v.nrep; m.work = v.work; m.use_sig = v.mode == 2; m.policy = v.policy; m.outputs = v.outputs;
        if (v.mode) hipLaunchKernelGGL(k_resident, dim3(NWG), dim3(WG), 0, sr, m, (u64)1500 * 100000);     // leaves after 1.5 s at the latest
        for (int pass = 0; pass < 2; ++pass) {          // pass 0 warms up (50 calls), pass 1 is timed
            const int k0 = pass ? 51 : 1, k1 = pass ? 50 + iters : 50;
            CK(hipEventRecord(e0, su));
            for (int k = k0; k <= k1; ++k) {
                if (v.policy) hipLaunchKernelGGL(k_policy, dim3(NWG), dim3(WG), 0, su, m.act, (u32)k);
                if (v.mode == 2) {
                    CK(hipStreamWriteValue32(su, m.rec + 1, k, 0)); CK(hipStreamWriteValue32(su, m.rec, k, 0));
                    CK(hipStreamWaitValue32(su, m.sig, k, hipStreamWaitValueGte, 0xFFFFFFFFu));
                } else hipLaunchKernelGGL(k_bell, dim3(1), dim3(WG), 0, su, m, (u32)k, v.mode == 0, ticks);
            }
            CK(hipEventRecord(e1, su)); CK(hipEventSynchronize(e1));
        }
        float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
        if (v.mode) { u32 ex = EXIT; for (int r = 0; r < 256; ++r) CK(hipMemcpyAsync(m.rec + r * 32, &ex, 4, hipMemcpyHostToDevice, su)); }
        CK(hipDeviceSynchronize());
        std::vector<u32> t(iters + 2); u32 err = 0;
        CK(hipMemcpy(t.data(), ticks, (iters + 2) * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(&err, m.err, 4, hipMemcpyDeviceToHost));
        std::vector<u32> s(t.begin() + 51, t.begin() + 51 + iters); std::sort(s.begin(), s.end());
        printf("%-48s period %7.2f us/call   in-bell post->all-done  p10 %5.2f  median %5.2f  p90 %5.2f us   timeouts %u  stale %u\n", v.name,
               ms * 1e3 / iters, s[iters / 10] * 0.01, s[iters / 2] * 0.01, s[iters * 9 / 10] * 0.01, err & 0xFFFF, err >> 16);
    }
    return 0;
}
