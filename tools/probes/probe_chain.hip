// What does one link of a chain of small DEPENDENT kernels cost on MI355X?  (diagnostic, not shipped)
// Every variant is launched N times back to back on one stream (eager and as a captured graph) and timed with events;
// kernel i reads the rows kernel i-1 wrote (ping-pong), like the decode step's blocks.
//   empty            nothing
//   rows             every workgroup: G rows x 12 source rows x 1.2 KB read, 1 row written
//   rows+w(shared)   + 108 KB of weights, the same for the 16 workgroups of a "head"
//   rows+w+lds       + 4 barriers with LDS exchanges
//   rows+w(unique)   + 108 KB of weights per workgroup, all different (17 MB)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__global__ void k_empty(const float* in, float* out) {}

template <int NW_LOADS, bool LDS, bool ROWS, int ORDER = 0>
__global__ __launch_bounds__(512) void k_link(const float* __restrict__ in, float* __restrict__ out, const float* __restrict__ W,
                                              size_t w_stride, int heads) {
    __shared__ float4 sh[512];
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int lin = blockIdx.x;
    const int h = lin % heads, g = lin / heads;
    float4 acc = make_float4(0, 0, 0, 0);
    float4 r[12];
    float4 v[NW_LOADS > 0 ? NW_LOADS : 1];
    const float* w = W + (size_t)h * w_stride;
    auto load_rows = [&]() {
        if (ORDER == 2) {
            if (wave < 2) {
                const float* base = in + ((size_t)(2 * g + wave) * 12) * 300;
#pragma unroll
                for (int j = 0; j < 12; ++j) r[j] = *reinterpret_cast<const float4*>(base + j * 300 + 4 * lane);
            }
        } else {
            const float* base = in + ((size_t)(2 * g + (wave >> 2)) * 12 + 3 * (wave & 3)) * 300;
#pragma unroll
            for (int j = 0; j < 3; ++j) r[j] = *reinterpret_cast<const float4*>(base + j * 300 + 4 * lane);
        }
    };
    auto load_w = [&]() {
        if (ORDER == 2) {
            if (wave >= 2) {
#pragma unroll
                for (int j = 0; j < NW_LOADS; ++j) v[j] = *reinterpret_cast<const float4*>(w + 4 * ((tid - 128) + 384 * j));
            }
        } else {
#pragma unroll
            for (int j = 0; j < NW_LOADS; ++j) v[j] = *reinterpret_cast<const float4*>(w + 4 * (tid + 512 * j));
        }
    };
    if (ORDER == 1) { if (NW_LOADS > 0) load_w(); if (ROWS) load_rows(); } else { if (ROWS) load_rows(); if (NW_LOADS > 0) load_w(); }
    if (ROWS) {
        if (ORDER == 2) { if (wave < 2) {
#pragma unroll
            for (int j = 0; j < 12; ++j) { acc.x += r[j].x; acc.y += r[j].y; acc.z += r[j].z; acc.w += r[j].w; } } }
        else {
#pragma unroll
            for (int j = 0; j < 3; ++j) { acc.x += r[j].x; acc.y += r[j].y; acc.z += r[j].z; acc.w += r[j].w; } }
    }
    if (NW_LOADS > 0 && (ORDER != 2 || wave >= 2)) {
#pragma unroll
        for (int j = 0; j < NW_LOADS; ++j) { acc.x += v[j].x; acc.y += v[j].y; acc.z += v[j].z; acc.w += v[j].w; }
    }
    if (LDS) {
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            sh[tid] = acc;
            __syncthreads();
            const float4 o = sh[(tid + 64) & 511];
            acc.x += o.x; acc.y += o.y; acc.z += o.z; acc.w += o.w;
            __syncthreads();
        }
    }
    // one row (300 floats) per (row, head) out: 75 threads
    if (tid < 75) *reinterpret_cast<float4*>(out + ((size_t)(2 * g) * 12 + h) * 300 + 4 * tid) = acc;
    if (tid >= 128 && tid < 203) *reinterpret_cast<float4*>(out + ((size_t)(2 * g + 1) * 12 + h) * 300 + 4 * (tid - 128)) = acc;
}

template <typename F>
static int timeit(const char* name, F launch, int n, hipStream_t s) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 20; ++i) launch(i);
    CK(hipStreamSynchronize(s));
    CK(hipEventRecord(e0, s));
    for (int i = 0; i < n; ++i) launch(i);
    CK(hipEventRecord(e1, s));
    CK(hipStreamSynchronize(s));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    // graph
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
    float msg; CK(hipEventElapsedTime(&msg, e0, e1));
    printf("%-34s eager %6.2f us/launch   graph %6.2f us/launch\n", name, ms * 1e3 / n, msg * 1e3 / n);
    return 0;
}

int main() {
    hipStream_t s; CK(hipStreamCreate(&s));
    const int heads = 10, groups = 16, nwg = heads * groups, n = 300;
    float *a, *b, *W;
    const size_t rows = (size_t)32 * 12 * 300;
    CK(hipMalloc(&a, rows * 4)); CK(hipMalloc(&b, rows * 4));
    const size_t wper = 36 * 512 * 4;            // floats per slice: 27 float4 per thread = 110 KB
    CK(hipMalloc(&W, (size_t)nwg * wper * 4));
    CK(hipMemset(a, 0, rows * 4)); CK(hipMemset(b, 0, rows * 4)); CK(hipMemset(W, 0, (size_t)nwg * wper * 4));
    auto pp = [&](int i, float*& in, float*& out) { in = (i & 1) ? a : b; out = (i & 1) ? b : a; };
    timeit("empty (160 x 512)", [&](int i) { float *in, *out; pp(i, in, out); hipLaunchKernelGGL(k_empty, dim3(nwg), dim3(512), 0, s, in, out); }, n, s);
    timeit("rows only", [&](int i) { float *in, *out; pp(i, in, out); hipLaunchKernelGGL((k_link<0, false, true>), dim3(nwg), dim3(512), 0, s, in, out, W, wper, heads); }, n, s);
    timeit("rows + LDS/barriers", [&](int i) { float *in, *out; pp(i, in, out); hipLaunchKernelGGL((k_link<0, true, true>), dim3(nwg), dim3(512), 0, s, in, out, W, wper, heads); }, n, s);
    timeit("rows + 110 KB weights (shared)", [&](int i) { float *in, *out; pp(i, in, out); hipLaunchKernelGGL((k_link<27, false, true>), dim3(nwg), dim3(512), 0, s, in, out, W, wper, heads); }, n, s);
    timeit("weights first, then rows", [&](int i) { float *in, *out; pp(i, in, out); hipLaunchKernelGGL((k_link<27, false, true, 1>), dim3(nwg), dim3(512), 0, s, in, out, W, wper, heads); }, n, s);
    timeit("rows: waves 0-1; weights: 2-7 (36 ld)", [&](int i) { float *in, *out; pp(i, in, out); hipLaunchKernelGGL((k_link<36, false, true, 2>), dim3(nwg), dim3(512), 0, s, in, out, W, wper, heads); }, n, s);
    timeit("rows + 110 KB weights + LDS", [&](int i) { float *in, *out; pp(i, in, out); hipLaunchKernelGGL((k_link<27, true, true>), dim3(nwg), dim3(512), 0, s, in, out, W, wper, heads); }, n, s);
    timeit("rows + 110 KB weights (unique)", [&](int i) { float *in, *out; pp(i, in, out); hipLaunchKernelGGL((k_link<27, false, true>), dim3(nwg), dim3(512), 0, s, in, out, W, wper, nwg); }, n, s);
    timeit("110 KB weights (shared), no rows", [&](int i) { float *in, *out; pp(i, in, out); hipLaunchKernelGGL((k_link<27, false, false>), dim3(nwg), dim3(512), 0, s, in, out, W, wper, heads); }, n, s);
    timeit("55 KB weights (shared), no rows", [&](int i) { float *in, *out; pp(i, in, out); hipLaunchKernelGGL((k_link<13, false, false>), dim3(nwg), dim3(512), 0, s, in, out, W, wper, heads); }, n, s);
    timeit("14 KB weights (shared), no rows", [&](int i) { float *in, *out; pp(i, in, out); hipLaunchKernelGGL((k_link<3, false, false>), dim3(nwg), dim3(512), 0, s, in, out, W, wper, heads); }, n, s);
    return 0;
}
