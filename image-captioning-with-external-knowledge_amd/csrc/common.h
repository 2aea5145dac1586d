// Shared device/host helpers for libick_amd (gfx950 only; wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "ick_amd.h"

namespace ick {

using f32x4 = __attribute__((ext_vector_type(4))) float;

constexpr int kWave = 64;

inline int ceil_div(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

extern int g_deterministic;     // misc.hip: -1 = not set yet (read ICK_DETERMINISTIC), 0 / 1
inline bool deterministic() {
    if (g_deterministic < 0) {
        const char* e = getenv("ICK_DETERMINISTIC");
        g_deterministic = (e && e[0] && e[0] != '0') ? 1 : 0;
    }
    return g_deterministic == 1;
}

#define ICK_CHECK_ARG(cond)          \
    do {                             \
        if (!(cond)) return ICK_EINVAL; \
    } while (0)

// hipFuncAttributeMaxDynamicSharedMemorySize is a property of (kernel, DEVICE): a process-wide "done" flag would leave a
// second GPU of the process at the 64 KB default and its launches failing.  One bit per device ordinal in an
// atomic mask per call site; setting the attribute twice from two threads is harmless.
struct LdsAttrOnce {
    unsigned long long done = 0;
    int ensure(const void* kernel, int bytes) {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess) return ICK_EINVAL;
        const unsigned long long bit = 1ull << (dev & 63);
        if (__atomic_load_n(&done, __ATOMIC_ACQUIRE) & bit) return ICK_OK;
        hipError_t e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
        if (e != hipSuccess) return (int)e;
        __atomic_fetch_or(&done, bit, __ATOMIC_RELEASE);
        return ICK_OK;
    }
};

#define ICK_LAUNCH_RET()                       \
    do {                                       \
        hipError_t e__ = hipGetLastError();    \
        return e__ == hipSuccess ? ICK_OK : (int)e__; \
    } while (0)

// Reductions over the 64 lanes of a wave, result in every lane.  The four steps inside a row of 16 lanes are DPP
// operands of the add / max itself (quad_perm [1,0,3,2], [2,3,0,1], row_half_mirror, row_mirror: one VALU instruction
// each); the four row results are then read into scalar registers (v_readlane) and combined as (r0 + r1) + (r2 + r3)
// -- the value the xor-16 / xor-32 butterfly gives, without its two ds_bpermute round trips (~100 cycles each).
//
// PRECONDITION: all 64 lanes active (call from wave-uniform control flow only -- every call site in this library
// does: loops / branches around them depend on blockIdx, kernel arguments or wave indices, never on the lane).  A DPP
// read of a disabled lane yields the `old` operand, which is the reduction's identity here (0 for the sum, the lane's own
// value for the maximum), but v_readlane of a disabled lane 0/16/32/48 would return a stale register.
template <int CTRL>
__device__ __forceinline__ float dpp_f32(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, false));
}
template <int CTRL>
__device__ __forceinline__ float dpp_f32_self(float v) {      // a disabled source lane reads as the lane's own value
    const int b = __builtin_bit_cast(int, v);
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(b, b, CTRL, 0xF, 0xF, false));
}
__device__ __forceinline__ float lane_f32(float v, int lane) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), lane));
}
__device__ __forceinline__ float wave_sum(float v) {
    v += dpp_f32<0xB1>(v); v += dpp_f32<0x4E>(v); v += dpp_f32<0x141>(v); v += dpp_f32<0x140>(v);
    return (lane_f32(v, 0) + lane_f32(v, 16)) + (lane_f32(v, 32) + lane_f32(v, 48));
}
__device__ __forceinline__ float wave_max(float v) {
    v = fmaxf(v, dpp_f32_self<0xB1>(v)); v = fmaxf(v, dpp_f32_self<0x4E>(v));
    v = fmaxf(v, dpp_f32_self<0x141>(v)); v = fmaxf(v, dpp_f32_self<0x140>(v));
    return fmaxf(fmaxf(lane_f32(v, 0), lane_f32(v, 16)), fmaxf(lane_f32(v, 32), lane_f32(v, 48)));
}

// Wave priority for the instruction arbiter (s_setprio, 0..3).  The latency-bound chain kernels (a few hundred
// workgroups, one dependent launch after another) raise it: beside a bulk GEMM on the other stream their waves
// otherwise get a round-robin share of the matrix pipe (measured: the 14 us in_proj GEMM of the context encoder
// takes 111 us beside Encoder.conv1), while the bulk kernel hardly notices the few cycles they take.
// ICK_NO_SETPRIO (compile time) turns it off.
#ifndef ICK_CHAIN_PRIO
#define ICK_CHAIN_PRIO 3
#endif
#ifndef ICK_CHAIN_PRIO_BWD
#define ICK_CHAIN_PRIO_BWD ICK_CHAIN_PRIO
#endif
__device__ __forceinline__ void chain_priority() {
#ifndef ICK_NO_SETPRIO
    __builtin_amdgcn_s_setprio(ICK_CHAIN_PRIO);
#endif
}
__device__ __forceinline__ void chain_priority_bwd() {      // backward-pass chain kernels (they run beside the weight gradients)
#ifndef ICK_NO_SETPRIO
    __builtin_amdgcn_s_setprio(ICK_CHAIN_PRIO_BWD);
#endif
}

// Block-wide reductions for blocks of NW waves; scratch must hold NW floats.
template <int NW>
__device__ __forceinline__ float block_sum(float v, float* scratch) {
    v = wave_sum(v);
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    __syncthreads();
    if (l == 0) scratch[w] = v;
    __syncthreads();
    float r = 0.f;
#pragma unroll
    for (int i = 0; i < NW; ++i) r += scratch[i];
    return r;
}
template <int NW>
__device__ __forceinline__ float block_max(float v, float* scratch) {
    v = wave_max(v);
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    __syncthreads();
    if (l == 0) scratch[w] = v;
    __syncthreads();
    float r = scratch[0];
#pragma unroll
    for (int i = 1; i < NW; ++i) r = fmaxf(r, scratch[i]);
    return r;
}

// Counter-based dropout: the keep decision of element `idx` at dropout site `site` of the step whose
// seed is `seed` is a pure function of (seed, site, idx) (murmur3 finaliser), so the backward pass
// regenerates the forward's mask without storing it.  thr = p * 2^32; keep iff hash >= thr.
__device__ __forceinline__ uint32_t drop_hash(uint32_t seed, uint32_t site, uint32_t idx) {
    uint32_t x = idx * 0x9E3779B1u ^ (seed + site * 0x85EBCA6Bu);
    x ^= x >> 16; x *= 0x85EBCA6Bu; x ^= x >> 13; x *= 0xC2B2AE35u; x ^= x >> 16;
    return x;
}
struct Dropout {
    uint32_t thr, seed, site;
    float scale;  // 1 / (1 - p); p == 0 disables
    __device__ __forceinline__ bool on() const { return thr != 0u; }
    __device__ __forceinline__ float mask(uint32_t idx) const { return drop_hash(seed, site, idx) >= thr ? scale : 0.f; }
};
__host__ __device__ __forceinline__ Dropout make_dropout(float p, uint32_t seed, uint32_t site) {
    Dropout d{0u, seed, site, 1.f};
    if (p > 0.f) {
        d.thr = (uint32_t)fminf(p * 4294967296.f, 4294967040.f);   // same IEEE float math on host and device
        d.scale = 1.f / (1.f - p);
    }
    return d;
}
// Kernel-side form: the effective seed is `seed + *epoch` when a device-resident step counter is given, so a
// captured hipGraph draws fresh masks on every replay (the counter is bumped by ick_counter_add in the graph).
//
// epoch_seed(): the counter is read through the SCALAR cache.  As `*epoch` the compiler issued a vector load of the
// uniform address and waited for it on the spot (global_load_dword, s_waitcnt vmcnt(0), v_readfirstlane): one L2 round
// trip at the head of every row-chain and attention launch of a training step, before any other load of the kernel had
// been requested.  The counter is written by an earlier kernel of the stream and a dispatch starts with a clean scalar
// cache, so the scalar read sees it; the wait sits inside the statement because the compiler does not count this load.
__device__ __forceinline__ uint32_t epoch_seed(uint32_t seed, const uint32_t* epoch) {
    if (epoch == nullptr) return seed;      // uniform
#if defined(__HIP_DEVICE_COMPILE__)
    uint32_t v;
    asm volatile("s_load_dword %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(epoch) : "memory");
    return seed + v;
#else
    return seed + *epoch;                   // (the host pass only parses device code)
#endif
}
struct DropArg {
    float p;
    uint32_t seed, site;
    const uint32_t* epoch;
    __device__ __forceinline__ Dropout get() const { return make_dropout(p, epoch_seed(seed, epoch), site); }
};

}  // namespace ick
