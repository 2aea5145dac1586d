// Micro-probes for the latency structure of small kernels on MI355X (diagnostic only, not shipped).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
using f32x4 = __attribute__((ext_vector_type(4))) float;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__global__ void k_empty() {}

template <int NLOAD, bool MFMA, bool LOAD, bool LDSW>
__global__ __launch_bounds__(256) void k_panel(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ C,
                                               int tiles_m, int rowlen /*floats*/) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x;
    const int tm = blockIdx.x % tiles_m, tn = blockIdx.x / tiles_m;
    const int LD = 308;
    const float* a = A + (size_t)tm * 32 * rowlen;
    const float* b = B + (size_t)tn * 32 * rowlen;
    float4 v[NLOAD];
    const int per = 32 * rowlen / 4;  // float4 per panel
#pragma unroll
    for (int j = 0; j < NLOAD; ++j) {
        int idx = tid + 256 * j;
        const float* src = idx < per ? a + 4 * idx : b + 4 * min(idx - per, per - 1);
        if (LOAD) v[j] = *reinterpret_cast<const float4*>(src); else v[j] = make_float4(idx, 1, 2, 3);
    }
    if (LDSW) {
#pragma unroll
        for (int j = 0; j < NLOAD; ++j) {
            int idx = tid + 256 * j;
            int row = idx / 75, c = idx % 75;
            if (row < 64) *reinterpret_cast<float4*>(smem + row * LD + 4 * c) = v[j];
        }
    } else {
        float s = 0; 
#pragma unroll
        for (int j = 0; j < NLOAD; ++j) s += v[j].x + v[j].w;
        if (s == 123.456f) C[tid] = s;
    }
    __syncthreads();
    const int lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1, fi = lane & 15, fq = lane >> 4;
    f32x4 acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
    if (MFMA) {
        const float* ap = smem + (wm * 16 + fi) * LD + 4 * fq;
        const float* bp = smem + (32 + wn * 16 + fi) * LD + 4 * fq;
        for (int t = 0; t < 19; ++t) {
            const float4 x = *reinterpret_cast<const float4*>(ap + 16 * t);
            const float4 y = *reinterpret_cast<const float4*>(bp + 16 * t);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x.x, y.x, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x.y, y.y, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x.z, y.z, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x.w, y.w, acc1, 0, 0, 0);
        }
    }
    const int col = tn * 32 + wn * 16 + fi;
    for (int r = 0; r < 4; ++r) {
        int row = tm * 32 + wm * 16 + fq * 4 + r;
        C[(size_t)row * 320 + col] = acc0[r] + acc1[r];
    }
}

__global__ __launch_bounds__(256) void k_ln(const float* __restrict__ x, float* __restrict__ y, int rows) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    float v[5]; float s = 0;
    for (int j = 0; j < 5; ++j) { int c = lane + 64 * j; v[j] = c < 300 ? x[(size_t)row * 300 + c] : 0.f; s += v[j]; }
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    for (int j = 0; j < 5; ++j) { int c = lane + 64 * j; if (c < 300) y[(size_t)row * 300 + c] = v[j] - s; }
}

template <typename F>
float timeit(F f, int iters, hipStream_t st) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 20; ++i) f();
    hipStreamSynchronize(st);
    hipEventRecord(e0, st);
    for (int i = 0; i < iters; ++i) f();
    hipEventRecord(e1, st);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms * 1000.f / iters;
}

int main() {
    hipStream_t st; CK(hipStreamCreate(&st));
    float *A, *B, *C;
    CK(hipMalloc(&A, 1280 * 300 * 4 + 65536)); CK(hipMalloc(&B, 2048 * 300 * 4 + 65536)); CK(hipMalloc(&C, 1280 * 2048 * 4));
    CK(hipMemset(A, 0, 1280 * 300 * 4)); CK(hipMemset(B, 0, 2048 * 300 * 4));
    const int iters = 500;
    printf("empty kernel back-to-back: %.2f us\n", timeit([&] { hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, st); }, iters, st));
    printf("empty kernel 400x256:      %.2f us\n", timeit([&] { hipLaunchKernelGGL(k_empty, dim3(400), dim3(256), 0, st); }, iters, st));
    printf("ln-like 1280 rows:         %.2f us\n", timeit([&] { hipLaunchKernelGGL(k_ln, dim3(320), dim3(256), 0, st, A, C, 1280); }, iters, st));
    const size_t smem = 64 * 308 * 4;
    auto run = [&](auto kern, const char* name, int tiles_n) {
        hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 159 * 1024);
        float t = timeit([&] { hipLaunchKernelGGL(kern, dim3(40 * tiles_n), dim3(256), smem, st, A, B, C, 40, 300); }, iters, st);
        printf("%-34s tiles_n=%2d blocks=%4d: %.2f us\n", name, tiles_n, 40 * tiles_n, t);
    };
    for (int tn : {10, 29, 57}) {
        run(k_panel<19, true, true, true>, "panel load+lds+mfma", tn);
        run(k_panel<19, false, true, true>, "panel load+lds", tn);
        run(k_panel<19, false, true, false>, "panel load only", tn);
        run(k_panel<19, true, false, true>, "panel lds+mfma (no global load)", tn);
        run(k_panel<19, false, false, true>, "panel lds write only", tn);
    }
    return 0;
}
