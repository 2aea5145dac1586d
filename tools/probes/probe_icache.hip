// Does a kernel pay for fetching its instructions?  Straight-line code (every instruction executed once, like the
// unrolled decode kernels) against the same number of instructions in a short loop.  (diagnostic, not shipped)
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int N>
__global__ __launch_bounds__(512) void k_straight(float* out, float x) {
    float a = x + threadIdx.x, b = x * 2.f;
#pragma unroll
    for (int i = 0; i < N; ++i) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a) : "v"(b));
    if (a == 123.f) out[threadIdx.x] = a;
}
template <int N, int BODY>
__global__ __launch_bounds__(512) void k_loop(float* out, float x, int n) {
    float a = x + threadIdx.x, b = x * 2.f;
    for (int it = 0; it < n; ++it) {
#pragma unroll
        for (int i = 0; i < BODY; ++i) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a) : "v"(b));
    }
    if (a == 123.f) out[threadIdx.x] = a;
}

template <typename F>
static int timeit(const char* name, F launch, int n, hipStream_t s) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < n; ++i) launch(i);
    CK(hipStreamEndCapture(s, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s));
    CK(hipEventRecord(e0, s));
    CK(hipGraphLaunch(ge, s));
    CK(hipEventRecord(e1, s));
    CK(hipStreamSynchronize(s));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-44s %6.2f us/launch\n", name, ms * 1e3 / n);
    return 0;
}

int main() {
    hipStream_t s; CK(hipStreamCreate(&s));
    float* out; CK(hipMalloc(&out, 4096));
    const int n = 200;
    timeit("straight-line  500 v_fma (160 x 512)", [&](int) { hipLaunchKernelGGL(k_straight<500>, dim3(160), dim3(512), 0, s, out, 1.f); }, n, s);
    timeit("straight-line 1000 v_fma", [&](int) { hipLaunchKernelGGL(k_straight<1000>, dim3(160), dim3(512), 0, s, out, 1.f); }, n, s);
    timeit("straight-line 2000 v_fma", [&](int) { hipLaunchKernelGGL(k_straight<2000>, dim3(160), dim3(512), 0, s, out, 1.f); }, n, s);
    timeit("straight-line 4000 v_fma", [&](int) { hipLaunchKernelGGL(k_straight<4000>, dim3(160), dim3(512), 0, s, out, 1.f); }, n, s);
    timeit("loop 20 x 100 = 2000 v_fma", [&](int) { hipLaunchKernelGGL((k_loop<2000, 100>), dim3(160), dim3(512), 0, s, out, 1.f, 20); }, n, s);
    timeit("loop 40 x 100 = 4000 v_fma", [&](int) { hipLaunchKernelGGL((k_loop<4000, 100>), dim3(160), dim3(512), 0, s, out, 1.f, 40); }, n, s);
    timeit("straight 2000, one wave per workgroup", [&](int) { hipLaunchKernelGGL(k_straight<2000>, dim3(160), dim3(64), 0, s, out, 1.f); }, n, s);
    timeit("alternating straight 2000 / 1000 / 4000", [&](int i) {
        if (i % 3 == 0) hipLaunchKernelGGL(k_straight<2000>, dim3(160), dim3(512), 0, s, out, 1.f);
        else if (i % 3 == 1) hipLaunchKernelGGL(k_straight<1000>, dim3(160), dim3(512), 0, s, out, 1.f);
        else hipLaunchKernelGGL(k_straight<4000>, dim3(160), dim3(512), 0, s, out, 1.f); }, n, s);
    return 0;
}
