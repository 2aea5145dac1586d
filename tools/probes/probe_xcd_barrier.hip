// What does a barrier between the workgroups of ONE XCD cost, against a barrier across all 8 XCDs?  (diagnostic, not
// shipped; VERDICT r3 item 3c: "probe an XCD-local persistent decode step; build only if the barrier is <= 1 us").
//
// 256 workgroups x 512 threads, one per CU.  Workgroups are dispatched round robin over the XCDs, so blockIdx & 7 is the
// XCD (checked against HW_REG_XCC_ID).  Every workgroup runs N sense-reversing barriers:
//   chip      one counter for all 256 workgroups, agent-scope atomics (what round 3 measured at ~4 us)
//   xcd/agent one counter per XCD (32 workgroups), agent-scope atomics
//   xcd/wg    one counter per XCD, WORKGROUP-scope atomics: the read-modify-write executes in that XCD's L2 and the
//             polling load bypasses only the CU's L1 (sc0) -- legal here because every participant shares that L2
//   + hand-off: before each barrier a workgroup stores 1.2 KB (one activation row) that its neighbour on the XCD reads
//             after it with L1-bypassing loads, and the values are checked (a barrier that does not publish data is useless)
// Timing: s_memrealtime (100 MHz) around the N barriers inside the kernel, max over workgroups; and events around the launch.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

// Returns false when the wait gave up (the caller then stops synchronising: every wave reaches the end of the kernel
// whatever the other workgroups do -- a barrier that does not work must not hang the GPU).
template <int SCOPE>
__device__ __forceinline__ bool barrier_arrive_wait(unsigned* counter, unsigned target, int* gave_up) {
    __shared__ int ok;
    // one thread per workgroup arrives and polls; the rest wait at the workgroup barrier
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELEASE, SCOPE);
        int spins = 0, fine = 1;
        while (__hip_atomic_load(counter, __ATOMIC_ACQUIRE, SCOPE) < target) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > (1 << 15) || __hip_atomic_load(gave_up, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                fine = 0;
                __hip_atomic_store(gave_up, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                break;
            }
        }
        ok = fine;
    }
    __syncthreads();
    const bool r = ok != 0;
    __syncthreads();
    return r;
}

template <int SCOPE, bool PER_XCD, bool HANDOFF>
__global__ __launch_bounds__(512) void k_barriers(unsigned* counters, float* rows, unsigned long long* out, unsigned* xcc_ids,
                                                  int n, int* bad, int* gave_up) {
    const int wg = blockIdx.x, xcd = wg & 7, slot = wg >> 3;            // slot: 0..31 inside the XCD
    unsigned* counter = counters + (PER_XCD ? 64 * xcd : 0);            // 256 B apart: one cache line each
    const unsigned members = PER_XCD ? gridDim.x / 8 : gridDim.x;
    if (threadIdx.x == 0) {
        unsigned x;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
        xcc_ids[wg] = x & 0xf;
    }
    const int nb = ((slot + 1) & 31) * 8 + xcd;                          // neighbour on the same XCD
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    int wrong = 0;
    for (int i = 1; i <= n; ++i) {
        if (HANDOFF) {
            if (threadIdx.x < 300) rows[(size_t)wg * 320 + threadIdx.x] = (float)(i * 1000 + wg);
            __threadfence();                                             // release: the stores are out before the arrive
        }
        if (!barrier_arrive_wait<SCOPE>(counter, members * (unsigned)i, gave_up)) break;
        if (HANDOFF && threadIdx.x < 300) {
            const float v = __builtin_nontemporal_load(rows + (size_t)nb * 320 + threadIdx.x);
            wrong += v != (float)(i * 1000 + nb);
        }
        if (HANDOFF) {
            // a second barrier keeps iteration i + 1's stores behind iteration i's reads (as a real chain would have
            // several phases per layer, every barrier is timed: 2 n in total)
            if (!barrier_arrive_wait<SCOPE>(counter + 16, members * (unsigned)i, gave_up)) break;
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) out[wg] = t1 - t0;
    if (wrong) atomicAdd(bad, 1);
}

template <int SCOPE, bool PER_XCD, bool HANDOFF>
int run(const char* name, int n) {
    unsigned* counters; float* rows; unsigned long long* out; unsigned* ids; int* bad; int* gave_up; int gu;
    CK(hipMalloc(&counters, 8 * 64 * 4)); CK(hipMemset(counters, 0, 8 * 64 * 4));
    CK(hipMalloc(&rows, 256 * 320 * 4)); CK(hipMemset(rows, 0, 256 * 320 * 4));
    CK(hipMalloc(&out, 256 * 8)); CK(hipMalloc(&ids, 256 * 4)); CK(hipMalloc(&bad, 4)); CK(hipMemset(bad, 0, 4));
    CK(hipMalloc(&gave_up, 4)); CK(hipMemset(gave_up, 0, 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((k_barriers<SCOPE, PER_XCD, HANDOFF>), dim3(256), dim3(512), 0, 0, counters, rows, out, ids, n, bad, gave_up);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> t(256); std::vector<unsigned> x(256); int nbad;
    CK(hipMemcpy(t.data(), out, 256 * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(x.data(), ids, 256 * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(&nbad, bad, 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(&gu, gave_up, 4, hipMemcpyDeviceToHost));
    unsigned long long mx = 0; int mism = 0;
    for (int i = 0; i < 256; ++i) { mx = t[i] > mx ? t[i] : mx; mism += (x[i] != (unsigned)(i & 7)); }
    const int nbar = HANDOFF ? 2 * n : n;
    printf("%-34s %6d barriers: %8.3f us each in-kernel (100 MHz clock), %8.3f us by events; XCC_ID != blockIdx&7 for %d of 256 "
           "workgroups; hand-off mismatches %d%s\n", name, nbar, (double)mx / 100.0 / nbar, ms * 1e3 / nbar, mism, nbad,
           gu ? "; A WAIT GAVE UP: this barrier does not work, the times mean nothing" : "");
    fflush(stdout);
    (void)hipFree(counters); (void)hipFree(rows); (void)hipFree(out); (void)hipFree(ids); (void)hipFree(bad); (void)hipFree(gave_up);
    return 0;
}

int main() {
    const int n = 2000;
    for (int rep = 0; rep < 2; ++rep) {
        if (run<__HIP_MEMORY_SCOPE_AGENT, false, false>("chip (256 WGs), agent scope", n)) return 1;
        if (run<__HIP_MEMORY_SCOPE_AGENT, true, false>("per XCD (32 WGs), agent scope", n)) return 1;
        if (run<__HIP_MEMORY_SCOPE_WORKGROUP, true, false>("per XCD (32 WGs), workgroup scope", n)) return 1;
        if (run<__HIP_MEMORY_SCOPE_AGENT, true, true>("per XCD + 1.2 KB hand-off, agent", n)) return 1;
        if (run<__HIP_MEMORY_SCOPE_WORKGROUP, true, true>("per XCD + 1.2 KB hand-off, wg scope", n)) return 1;
    }
    return 0;
}
