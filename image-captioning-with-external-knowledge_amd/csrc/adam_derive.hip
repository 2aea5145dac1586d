// clip_gradient + torch.optim.Adam.step (geo-aware/utils.py:75-85, geo-aware/train.py:287-292) over the flat parameter
// bucket, and -- in the same pass over the same registers -- every re-laid-out copy of the weights that the next step's
// kernels read: the packed row-chain images (forward and transposed, csrc/rowchain.hip), the gathered all-layer cross
// K/V weight / bias, the bf16 hi / mid / lo planes of the large GEMMs' weights (csrc/gemm_ps.hip), the transposed
// predicate weight.  The optimizer is the only writer of the parameters, so nothing else has to re-lay them out: round 4
// spent 3 ick_pack_weights + 3 ick_presplit_weights launches per step on it (126 us of kernel time at cfg2, 30 MB per
// launch at 0.18 of the HBM rate; skipping them took 105 us off the 1.73 ms step, profiles/r05_x_ceilings_train.txt).
//
// One workgroup = one block of the caller's cover of the bucket (include/ick_amd.h, ick_adam_clamp_derive):
//   flat run   <= 1024 float4, the seven 16-byte streams of adam_clamp_vec4_kernel (+ an optional plain copy);
//   tile       64 rows x 64 columns of a 2-D parameter (row segments of 256 contiguous bytes): updated in registers,
//              written back, the new values staged in LDS and read out once per image in that image's own order, so that
//              every image is written in runs of >= 1 KiB (pack, pack_t: [64 rows][4 k] granules; ps, ps_t: 64-byte
//              rows of one plane and K slice).
// The arithmetic is adam_clamp_kernel's, operation for operation (tests/test_adam_derive_gpu.py: bit-identical
// parameters and moments; images bit-identical to ick_pack_weights / ick_presplit_weights of the updated weights).
#include "gemm_common.h"

namespace ick {
namespace {

struct AdamHyper {
    float gscale, clip, b1, b2, c1, c2, eps, step, bc2_sqrt;
};

__device__ __forceinline__ void adam4(const AdamHyper& h, float4& p, float4& g, float4& m, float4& v) {
    float ga[4] = {g.x, g.y, g.z, g.w}, ma[4] = {m.x, m.y, m.z, m.w}, va[4] = {v.x, v.y, v.z, v.w};
    float pa[4] = {p.x, p.y, p.z, p.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) {      // the arithmetic of adam_clamp_kernel (csrc/backward.hip), operation for operation
        float x = ga[k] * h.gscale;
        if (h.clip > 0.f) x = fminf(fmaxf(x, -h.clip), h.clip);
        ga[k] = x;
        ma[k] = h.b1 * ma[k] + h.c1 * x;
        va[k] = h.b2 * va[k] + h.c2 * x * x;
        pa[k] -= h.step * ma[k] / (sqrtf(va[k]) / h.bc2_sqrt + h.eps);
    }
    g = make_float4(ga[0], ga[1], ga[2], ga[3]);
    m = make_float4(ma[0], ma[1], ma[2], ma[3]);
    v = make_float4(va[0], va[1], va[2], va[3]);
    p = make_float4(pa[0], pa[1], pa[2], pa[3]);
}

constexpr int kT = 64;        // tile edge
constexpr int kLd = 68;       // LDS row stride of the tile (16-byte aligned rows, 4 banks of skew per row)

__global__ __launch_bounds__(256) void adam_derive_kernel(float* __restrict__ p, float* __restrict__ g,
                                                          float* __restrict__ m, float* __restrict__ v,
                                                          const ick_adam_item* __restrict__ items,
                                                          const ick_adam_block* __restrict__ blocks, float gscale,
                                                          float clip, float lr, float b1, float b2, float eps, int step0,
                                                          const uint32_t* step_ptr, const float* __restrict__ gscale_den) {
    __shared__ __attribute__((aligned(16))) float tile[kT * kLd];
    const ick_adam_block blk = blocks[blockIdx.x];
    if (gscale_den) {
        // a (global) batch without a single contributing token has no mean loss: parameters, moments and every image stay
        if (!(gscale_den[0] > 0.f)) return;
        gscale = gscale / gscale_den[0];
    }
    AdamHyper h;
    {
        const float t = (float)(step0 + (step_ptr ? (int)*step_ptr : 0));
        const float bc1 = 1.f - powf(b1, t);
        h.bc2_sqrt = sqrtf(1.f - powf(b2, t));
        h.step = lr / bc1;
        h.gscale = gscale; h.clip = clip; h.b1 = b1; h.b2 = b2; h.c1 = 1.f - b1; h.c2 = 1.f - b2; h.eps = eps;
    }
    const int tid = threadIdx.x;

    if (blk.item < 0) {
        // ---- flat run: up to four float4 per thread, every load issued before the first use
        float4* p4 = reinterpret_cast<float4*>(p) + blk.off4;
        float4* g4 = reinterpret_cast<float4*>(g) + blk.off4;
        float4* m4 = reinterpret_cast<float4*>(m) + blk.off4;
        float4* v4 = reinterpret_cast<float4*>(v) + blk.off4;
        float4 pi[4], gi[4], mi[4], vi[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int j = tid + 256 * i;
            if (j < blk.cnt4) { gi[i] = g4[j]; mi[i] = m4[j]; vi[i] = v4[j]; pi[i] = p4[j]; }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int j = tid + 256 * i;
            if (j < blk.cnt4) {
                adam4(h, pi[i], gi[i], mi[i], vi[i]);
                g4[j] = gi[i]; m4[j] = mi[i]; v4[j] = vi[i]; p4[j] = pi[i];
                if (blk.copy) reinterpret_cast<float4*>(blk.copy)[j] = pi[i];
            }
        }
        return;
    }

    // ---- tile (tn, tk) of an item
    const ick_adam_item it = items[blk.item];
    const int n0 = blk.tn * kT, k0 = blk.tk * kT;
    const int K = it.K;
    {
        float4 pi[4], gi[4], mi[4], vi[4];
        int64_t at[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int idx = tid + 256 * i;
            const int row = idx >> 4, c4 = idx & 15;
            const int r = n0 + row - it.drow0, k = k0 + 4 * c4;
            at[i] = (r >= 0 && r < it.rows && k < K) ? it.off + (int64_t)r * K + k : -1;
            if (at[i] >= 0) {
                gi[i] = *reinterpret_cast<const float4*>(g + at[i]);
                mi[i] = *reinterpret_cast<const float4*>(m + at[i]);
                vi[i] = *reinterpret_cast<const float4*>(v + at[i]);
                pi[i] = *reinterpret_cast<const float4*>(p + at[i]);
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int idx = tid + 256 * i;
            const int row = idx >> 4, c4 = idx & 15;
            float4 out = make_float4(0.f, 0.f, 0.f, 0.f);      // outside the item / the matrix: zeros (the images' padding)
            if (at[i] >= 0) {
                adam4(h, pi[i], gi[i], mi[i], vi[i]);
                *reinterpret_cast<float4*>(g + at[i]) = gi[i];
                *reinterpret_cast<float4*>(m + at[i]) = mi[i];
                *reinterpret_cast<float4*>(v + at[i]) = vi[i];
                *reinterpret_cast<float4*>(p + at[i]) = pi[i];
                out = pi[i];
                if (it.copy) *reinterpret_cast<float4*>(it.copy + (int64_t)(n0 + row) * it.copy_ld + k0 + 4 * c4) = pi[i];
            }
            *reinterpret_cast<float4*>(tile + row * kLd + 4 * c4) = out;
        }
    }
    __syncthreads();
    const int rlo = it.drow0 - n0, rhi = it.drow0 + it.rows - n0;      // tile rows [rlo, rhi) belong to the item

    if (it.pack) {
        // dst[slab = n / 64][k4][l = n % 64][4]: lanes <-> 64 rows, 1 KiB per k4
        const int K16 = (K + 15) & ~15;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int u = tid + 256 * i;
            const int j = u >> 6, l = u & 63;
            const int k4 = blk.tk * 16 + j;
            if (l >= rlo && l < rhi && 4 * k4 < K)
                *reinterpret_cast<float4*>(it.pack + (((int64_t)blk.tn * (K16 / 4) + k4) * 64 + l) * 4) =
                    *reinterpret_cast<const float4*>(tile + l * kLd + 4 * j);
        }
    }
    if (it.pack_t) {
        // the image of W^T (K rows, Nd columns): dst[slab = k / 64][n / 4][l = k % 64][n % 4]; lanes <-> 64 k
        const int N16 = (it.Nd + 15) & ~15;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int u = tid + 256 * i;
            const int j = u >> 6, l = u & 63;          // rows 4j .. 4j+3 of the tile, column l
            if (4 * j >= rlo && 4 * j < rhi && k0 + l < K) {
                const float4 x = make_float4(tile[(4 * j) * kLd + l], tile[(4 * j + 1) * kLd + l],
                                             tile[(4 * j + 2) * kLd + l], tile[(4 * j + 3) * kLd + l]);
                *reinterpret_cast<float4*>(it.pack_t + (((int64_t)blk.tk * (N16 / 4) + (n0 >> 2) + j) * 64 + l) * 4) = x;
            }
        }
    }
    if (it.ps) {
        // plane[slice = k / 32][row n < Np][32 k] bf16: a thread splits 8 consecutive k of one row (16 bytes per plane)
        const int64_t np = (int64_t)((it.Nd + 63) / 64) * 64;
        const int K32 = (K + 31) & ~31;
        char* base = reinterpret_cast<char*>(it.ps);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int u = tid + 256 * i;
            const int row = u & 63, c = u >> 6;
            const int k = k0 + 8 * c;
            if (row >= rlo && row < rhi && k < K32) {
                const float4 a = *reinterpret_cast<const float4*>(tile + row * kLd + 8 * c);
                const float4 b = *reinterpret_cast<const float4*>(tile + row * kLd + 8 * c + 4);
                uint32_t hh[4], mm[4], ll[4];
                split3(a.x, a.y, hh[0], mm[0], ll[0]); split3(a.z, a.w, hh[1], mm[1], ll[1]);
                split3(b.x, b.y, hh[2], mm[2], ll[2]); split3(b.z, b.w, hh[3], mm[3], ll[3]);
                char* dst = base + (((int64_t)(k >> 5) * 3) * np + n0 + row) * 64 + (k & 31) * 2;
                *reinterpret_cast<uint4*>(dst) = uint4{hh[0], hh[1], hh[2], hh[3]};
                *reinterpret_cast<uint4*>(dst + np * 64) = uint4{mm[0], mm[1], mm[2], mm[3]};
                *reinterpret_cast<uint4*>(dst + 2 * np * 64) = uint4{ll[0], ll[1], ll[2], ll[3]};
            }
        }
    }
    if (it.ps_t) {
        // the image of W^T: plane[slice = n / 32][row k < Kp][32 n] bf16: a thread splits 8 consecutive n of one k
        const int64_t kp = (int64_t)((K + 63) / 64) * 64;
        char* base = reinterpret_cast<char*>(it.ps_t);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int u = tid + 256 * i;
            const int kk = u & 63, c = u >> 6;
            if (8 * c >= rlo && 8 * c < rhi && k0 + kk < K) {
                float x[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) x[e] = tile[(8 * c + e) * kLd + kk];
                uint32_t hh[4], mm[4], ll[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) split3(x[2 * e], x[2 * e + 1], hh[e], mm[e], ll[e]);
                const int n = n0 + 8 * c;
                char* dst = base + (((int64_t)(n >> 5) * 3) * kp + k0 + kk) * 64 + (n & 31) * 2;
                *reinterpret_cast<uint4*>(dst) = uint4{hh[0], hh[1], hh[2], hh[3]};
                *reinterpret_cast<uint4*>(dst + kp * 64) = uint4{mm[0], mm[1], mm[2], mm[3]};
                *reinterpret_cast<uint4*>(dst + 2 * kp * 64) = uint4{ll[0], ll[1], ll[2], ll[3]};
            }
        }
    }
    if (it.tr) {
        // plain W^T: lanes <-> 64 consecutive n of one k (256 contiguous bytes)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int u = tid + 256 * i;
            const int nl = u & 63, kl = u >> 6;
            if (nl >= rlo && nl < rhi && k0 + kl < K) it.tr[(int64_t)(k0 + kl) * it.tr_ld + n0 + nl] = tile[nl * kLd + kl];
        }
    }
}

}  // namespace
}  // namespace ick

extern "C" int ick_adam_clamp_derive(float* p, float* g, float* m, float* v, const ick_adam_item* items,
                                     const ick_adam_block* blocks, int32_t n_blocks, float gscale, float clip, float lr,
                                     float beta1, float beta2, float eps, int32_t step, const uint32_t* step_ptr,
                                     const float* gscale_den, void* stream) {
    using namespace ick;
    ICK_CHECK_ARG(p && g && m && v && blocks && n_blocks > 0 && (step >= 1 || step_ptr != nullptr));
    auto al16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
    if (!(al16(p) && al16(g) && al16(m) && al16(v))) return ICK_EALIGN;
    hipLaunchKernelGGL(adam_derive_kernel, dim3(n_blocks), dim3(256), 0, (hipStream_t)stream, p, g, m, v, items, blocks,
                       gscale, clip, lr, beta1, beta2, eps, step, step_ptr, gscale_den);
    ICK_LAUNCH_RET();
}
