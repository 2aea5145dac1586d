// How many cycles does one v_mfma_f32_16x16x32_bf16 take on this MI355X, and at what clock?  (diagnostic, not shipped)
//
// Every ps-GEMM tile shape, ring depth and fragment order of round 4 ended at the same rate: kernel time = (number of
// MFMAs) x 32 cycles / (SIMDs in use x ~2.1 GHz).  Is that the matrix pipe (32 cycles per instruction, half of the 16 the
// spec sheet's 2.5 PFLOP/s implies) or is the pipe half idle?  This probe issues N MFMAs per wave from registers only
// (random bf16 operands, no memory in the loop), with 1 / 4 / 16 accumulator chains, 1 / 2 / 4 waves per SIMD on every CU,
// and reports cycles per MFMA per SIMD by the shader clock (s_memtime) and the clock itself (s_memtime vs the 100 MHz
// s_memrealtime), plus the same for the exact fp32 v_mfma_f32_16x16x4_f32.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <algorithm>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int CHAINS, bool F32>
__global__ __launch_bounds__(1024) void k_mfma(const unsigned* seed, float* out, unsigned long long* clocks, int iters) {
    const int lane = threadIdx.x & 63;
    unsigned s = seed[lane] | 0x3f003f00u;
    const unsigned w0 = s * 2654435761u, w1 = w0 ^ 0x9e3779b9u, w2 = w1 * 40503u, w3 = w2 ^ (w0 >> 3);
    // operands: bit patterns of moderate bf16 numbers (exponents near 1.0), different in every lane
    auto mk = [&](unsigned a, unsigned b, unsigned c, unsigned d) {
        typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
        u32x4 v = {(a & 0x007f007fu) | 0x3f803f80u, (b & 0x007f007fu) | 0x3f003f00u, (c & 0x007f007fu) | 0x3f803f80u,
                   (d & 0x007f007fu) | 0xbf00bf00u};
        return __builtin_bit_cast(bf16x8_t, v);
    };
    const bf16x8_t a0 = mk(w0, w1, w2, w3), b0 = mk(w3, w2, w1, w0), a1 = mk(w1, w3, w0, w2), b1 = mk(w2, w0, w3, w1);
    f32x4 acc[CHAINS];
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) acc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 16 / CHAINS; ++u)
#pragma unroll
            for (int c = 0; c < CHAINS; ++c) {
                if constexpr (F32) {
                    acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(__builtin_bit_cast(float, w0 | 0x3f800000u & 0x3fffffffu),
                                                                   __builtin_bit_cast(float, w1 | 0x3f000000u & 0x3fffffffu), acc[c], 0, 0, 0);
                } else {
                    acc[c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16((u & 1) ? a1 : a0, (c & 1) ? b1 : b0, acc[c], 0, 0, 0);
                }
            }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float sum = 0.f;
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) sum += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = sum;
    // the SIMD serves its oldest wave first: a single wave's own span says nothing about the others, so the workgroup's
    // span is first start to last end over all of its waves
    __shared__ unsigned long long tmin, tmax, rmin, rmax;
    if (threadIdx.x == 0) { tmin = ~0ull; tmax = 0; rmin = ~0ull; rmax = 0; }
    __syncthreads();
    if (lane == 0) {
        atomicMin(&tmin, t0); atomicMax(&tmax, t1); atomicMin(&rmin, r0); atomicMax(&rmax, r1);
    }
    __syncthreads();
    if (threadIdx.x == 0) { clocks[2 * blockIdx.x] = tmax - tmin; clocks[2 * blockIdx.x + 1] = rmax - rmin; }
}

template <int CHAINS, bool F32>
int run(const char* name, int waves_per_simd, int iters) {
    const int nblk = 256, nthr = 256 * waves_per_simd;       // one workgroup per CU, waves_per_simd waves on each SIMD
    unsigned* seed; float* out; unsigned long long* clk;
    CK(hipMalloc(&seed, 64 * 4)); CK(hipMalloc(&out, (size_t)nblk * nthr * 4)); CK(hipMalloc(&clk, nblk * 16));
    std::vector<unsigned> hs(64);
    for (int i = 0; i < 64; ++i) hs[i] = 0x12345u * (i + 7) + 0x9e37u * i * i;
    CK(hipMemcpy(seed, hs.data(), 256, hipMemcpyHostToDevice));
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL((k_mfma<CHAINS, F32>), dim3(nblk), dim3(nthr), 0, 0, seed, out, clk, iters);
    CK(hipDeviceSynchronize());
    std::vector<unsigned long long> c(2 * nblk);
    CK(hipMemcpy(c.data(), clk, nblk * 16, hipMemcpyDeviceToHost));
    std::vector<unsigned long long> cyc(nblk), rt(nblk);
    for (int i = 0; i < nblk; ++i) { cyc[i] = c[2 * i]; rt[i] = c[2 * i + 1]; }
    std::sort(cyc.begin(), cyc.end()); std::sort(rt.begin(), rt.end());
    const double mfma_per_simd = (double)iters * 16 * waves_per_simd;
    const double cycles = (double)cyc[nblk / 2], us = (double)rt[nblk / 2] / 100.0;
    const double flop = F32 ? 16.0 * 16 * 4 * 2 : 16.0 * 16 * 32 * 2;
    printf("%-22s chains %2d waves/SIMD %d: %6.2f shader cycles per MFMA per SIMD, clock %5.2f GHz, %7.1f TFLOP/s chip-wide (1024 SIMDs)\n",
           name, CHAINS, waves_per_simd, cycles / mfma_per_simd, cycles / us / 1e3, mfma_per_simd * flop * 1024 / us / 1e6);
    fflush(stdout);
    (void)hipFree(seed); (void)hipFree(out); (void)hipFree(clk);
    return 0;
}

int main() {
    const int iters = 20000;
    for (int w : {1, 2, 4}) {
        if (run<1, false>("bf16 16x16x32", w, iters)) return 1;
        if (run<4, false>("bf16 16x16x32", w, iters)) return 1;
        if (run<16, false>("bf16 16x16x32", w, iters)) return 1;
    }
    for (int w : {1, 2}) {
        if (run<4, true>("f32 16x16x4", w, iters)) return 1;
        if (run<16, true>("f32 16x16x4", w, iters)) return 1;
    }
    return 0;
}
