// y = LayerNorm(x + res) * gamma + beta, one wave per row (d <= 1024; d = 300 on this path).
// Replaces the residual add + norm1/norm2/norm3 of the post-LN Transformer layers the reference
// builds at geo-aware/models.py:241-244 (torch/nn/modules/transformer.py, norm_first=False).
// HBM-bound elementwise op: a row (1200 B) is read once, kept in registers for the two-pass
// mean / variance (same formulation as torch: biased variance around the mean), written once.
#include "common.h"

namespace ick {
namespace {

constexpr int kMaxPerLane = 16;  // 16 * 64 = 1024

template <int NJ>
__global__ __launch_bounds__(256) void add_layernorm_kernel(const float* __restrict__ x, const float* __restrict__ res,
                                                            const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, float* __restrict__ y,
                                                            int64_t rows, int d, float eps, int64_t x_ld,
                                                            int64_t res_ld, int64_t y_ld, float* save_mean,
                                                            float* save_rstd, DropArg darg) {
    chain_priority();
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const Dropout drop = darg.get();
    const float* xr = x + row * x_ld;
    const float* rr = res ? res + row * res_ld : nullptr;
    float v[NJ], g[NJ], bt[NJ];
    float sum = 0.f;
    // every load of the row (x, residual, gamma, beta) is issued up front: one memory round trip
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int c = lane + 64 * j;
        float t = 0.f;
        g[j] = 0.f;
        bt[j] = 0.f;
        if (c < d) {
            t = xr[c];
            if (drop.on()) t *= drop.mask((uint32_t)row * (uint32_t)d + (uint32_t)c);   // dropout1/2/3 of the layer
            if (rr) t += rr[c];
            g[j] = gamma[c];
            bt[j] = beta[c];
        }
        v[j] = t;
        sum += t;
    }
    const float mean = wave_sum(sum) / (float)d;
    float var = 0.f;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int c = lane + 64 * j;
        const float t = c < d ? v[j] - mean : 0.f;
        var = fmaf(t, t, var);
    }
    const float rstd = rsqrtf(wave_sum(var) / (float)d + eps);
    float* yr = y + row * y_ld;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int c = lane + 64 * j;
        if (c < d) yr[c] = (v[j] - mean) * rstd * g[j] + bt[j];
    }
    if (save_mean && lane == 0) {
        save_mean[row] = mean;
        save_rstd[row] = rstd;
    }
}

}  // namespace
}  // namespace ick

extern "C" int ick_add_layernorm(const float* x, const float* res, const float* gamma, const float* beta, float* y,
                                 int64_t rows, int32_t d, float eps, int64_t x_ld, int64_t res_ld, int64_t y_ld,
                                 float* save_mean, float* save_rstd, float drop_p, uint32_t drop_seed,
                                 uint32_t drop_site, const uint32_t* drop_epoch, void* stream) {
    using namespace ick;
    ICK_CHECK_ARG(x && gamma && beta && y);
    ICK_CHECK_ARG(rows > 0 && d > 0 && d <= 64 * kMaxPerLane);
    ICK_CHECK_ARG((save_mean == nullptr) == (save_rstd == nullptr));
    const DropArg dr{drop_p, drop_seed, drop_site, drop_epoch};
    const dim3 grid(ceil_div(rows, 4));
    hipStream_t s = (hipStream_t)stream;
    if (d <= 320)
        hipLaunchKernelGGL(add_layernorm_kernel<5>, grid, dim3(256), 0, s, x, res, gamma, beta, y, rows, d, eps, x_ld,
                           res_ld, y_ld, save_mean, save_rstd, dr);
    else if (d <= 512)
        hipLaunchKernelGGL(add_layernorm_kernel<8>, grid, dim3(256), 0, s, x, res, gamma, beta, y, rows, d, eps, x_ld,
                           res_ld, y_ld, save_mean, save_rstd, dr);
    else
        hipLaunchKernelGGL(add_layernorm_kernel<16>, grid, dim3(256), 0, s, x, res, gamma, beta, y, rows, d, eps, x_ld,
                           res_ld, y_ld, save_mean, save_rstd, dr);
    ICK_LAUNCH_RET();
}
