// fp32 GEMM on the CDNA4 matrix cores (v_mfma_f32_16x16x4_f32): exact fp32 products and
// accumulation, 64 FLOP/clk/SIMD.  C[m,n] = act(alpha * sum_k A(m,k) B(n,k) + bias[n]).
//
// Replaces the nn.Linear / 1x1 nn.Conv2d call sites of the reference's decoder path
// (see include/ick_amd.h).  Design for gfx950:
//   * block = 256 threads = 4 waves (WM x WN); every wave owns TM x TN tiles of 16x16 and keeps
//     them in registers for the whole K loop (TM=TN=4: 64 accumulator VGPRs, 16 independent
//     accumulation chains, so the 40-cycle dependent latency of the 16x16x4 MFMA is hidden);
//   * K is walked in BK=32 slices, global -> registers -> LDS, double buffered: the loads of
//     slice i+1 are issued before the MFMAs of slice i and written to the other LDS buffer
//     after them (one barrier per slice);
//   * an operand may be k-contiguous ([row][k], weights and activations of a Linear) or
//     k-major ([k][row]: the NCHW feature map of Encoder.conv1, and the operands of the
//     weight/data-gradient GEMMs).  k-contiguous tiles sit in LDS as [row][32+4] and are read
//     with one ds_read_b128 per 16x16 k-chunk: lane (i = l&15, q = l>>4) takes k = 16t+4q..+3
//     and MFMA step u consumes k = 16t+4q+u.  Both operands use the same k order, so the sum
//     is the same set of products.  k-major tiles sit as [k][rows+4] and are read with
//     ds_read_b32 in that same k order; both strides are bank-conflict free (<=2-way);
//   * the tile index is remapped so the 8 XCDs (round-robin over blockIdx) each walk a
//     contiguous run of tiles and the smaller operand panel stays in that XCD's L2.
#include "common.h"

namespace ick {
namespace {

constexpr int BK = 32;
constexpr int LDK = BK + 4;  // floats per LDS row of a k-contiguous tile

struct RowMap {  // offset of logical row r:  goff(r / grp) + (r % grp) * rs
    int grp;
    int64_t gs;
    const int32_t* gmap;
    int64_t rs;
    __device__ __forceinline__ int64_t operator()(int r) const {
        if (grp <= 0) return (int64_t)r * rs;
        const int g = r / grp;
        const int i = r - g * grp;
        const int64_t gg = gmap ? (int64_t)gmap[g] : (int64_t)g;
        return gg * gs + (int64_t)i * rs;
    }
};

// Global -> register -> LDS stager for one operand tile of R rows x BK k.
template <int R, bool KM>
struct Stager {
    static constexpr int NP = R / 32;                // float4 per thread
    static constexpr int CH = KM ? R / 4 : BK / 4;   // float4 chunks along the contiguous dim
    static constexpr int LD = KM ? R + 4 : LDK;
    static constexpr int FLOATS = KM ? BK * LD : R * LD;
    const float* base;
    int64_t off[KM ? 1 : NP];  // element offset of this thread's row(s); <0 = out of range
    int64_t ks;                // k stride (1 for k-contiguous)
    int c, r0;                 // chunk index / first row (k-contig) or first k line (k-major)
    bool vec;
    int rows_total;
    int row_first;             // k-major: first of this thread's 4 rows (global index)
    RowMap rm;
    float4 v[NP];

    __device__ __forceinline__ void init(const float* p, const RowMap& m, int64_t kstride, int tile_row0,
                                         int rows, bool vec_ok) {
        base = p; ks = kstride; vec = vec_ok; rows_total = rows; rm = m;
        const int t = threadIdx.x;
        c = t % CH;
        r0 = t / CH;
        if constexpr (KM) {
            row_first = tile_row0 + 4 * c;
            off[0] = row_first < rows ? m(row_first) : -1;
        } else {
#pragma unroll
            for (int j = 0; j < NP; ++j) {
                const int gr = tile_row0 + r0 + 32 * j;
                off[j] = gr < rows ? m(gr) : -1;
            }
        }
    }

    __device__ __forceinline__ void load(int k0, int kend) {
        if constexpr (KM) {
            constexpr int KP = 256 / CH;
#pragma unroll
            for (int j = 0; j < NP; ++j) {
                const int k = k0 + r0 + KP * j;
                float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
                if (k < kend && off[0] >= 0) {
                    if (vec) {
                        x = *reinterpret_cast<const float4*>(base + off[0] + (int64_t)k * ks);
                    } else {
                        float e[4];
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const int gr = row_first + q;
                            e[q] = gr < rows_total ? base[rm(gr) + (int64_t)k * ks] : 0.f;
                        }
                        x = make_float4(e[0], e[1], e[2], e[3]);
                    }
                }
                v[j] = x;
            }
        } else {
            const int k = k0 + 4 * c;
#pragma unroll
            for (int j = 0; j < NP; ++j) {
                float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
                if (off[j] >= 0 && k < kend) {
                    const float* p = base + off[j] + k;
                    if (vec) {
                        x = *reinterpret_cast<const float4*>(p);
                    } else {
                        x.x = p[0];
                        if (k + 1 < kend) x.y = p[1];
                        if (k + 2 < kend) x.z = p[2];
                        if (k + 3 < kend) x.w = p[3];
                    }
                }
                v[j] = x;
            }
        }
    }

    __device__ __forceinline__ void store(float* lds) const {
        if constexpr (KM) {
            constexpr int KP = 256 / CH;
#pragma unroll
            for (int j = 0; j < NP; ++j)
                *reinterpret_cast<float4*>(lds + (r0 + KP * j) * LD + 4 * c) = v[j];
        } else {
#pragma unroll
            for (int j = 0; j < NP; ++j)
                *reinterpret_cast<float4*>(lds + (r0 + 32 * j) * LD + 4 * c) = v[j];
        }
    }
};

// One 16-row fragment for the four MFMA steps of k-chunk t.
template <int R, bool KM>
__device__ __forceinline__ void read_frag(const float* lds, int row0, int t, int i, int q, float (&f)[4]) {
    if constexpr (KM) {
        constexpr int LD = R + 4;
#pragma unroll
        for (int u = 0; u < 4; ++u) f[u] = lds[(16 * t + 4 * q + u) * LD + row0 + i];
    } else {
        const float4 x = *reinterpret_cast<const float4*>(lds + (row0 + i) * LDK + 16 * t + 4 * q);
        f[0] = x.x; f[1] = x.y; f[2] = x.z; f[3] = x.w;
    }
}

template <int WM, int WN, int TM, int TN, bool AKM, bool BKM>
__global__ __launch_bounds__(256) void gemm_kernel(ick_gemm_args p, int tiles_m, int tiles_n, int kchunk) {
    constexpr int BM = WM * TM * 16, BN = WN * TN * 16;
    using SA = Stager<BM, AKM>;
    using SB = Stager<BN, BKM>;
    constexpr int STAGE = SA::FLOATS + SB::FLOATS;
    extern __shared__ __attribute__((aligned(16))) float smem[];

    // XCD-aware tile order (blocks b, b+8, ... share an XCD).
    const int nwg = tiles_m * tiles_n;
    int bid = blockIdx.x;
    {
        const int xcd = bid & 7, qq = nwg >> 3, rr = nwg & 7;
        bid = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + (bid >> 3);
    }
    int tm, tn;
    if (tiles_m <= tiles_n) { tm = bid % tiles_m; tn = bid / tiles_m; }
    else { tn = bid % tiles_n; tm = bid / tiles_n; }
    const int m0 = tm * BM, n0 = tn * BN;
    const int kbeg = blockIdx.z * kchunk;
    const int kend = min(p.K, kbeg + kchunk);

    const RowMap amap{p.a_grp, p.a_gs, p.a_gmap, p.a_rs};
    const RowMap bmap{0, 0, nullptr, p.b_rs};
    const bool avec = (p.flags >> 8) & 1, bvec = (p.flags >> 9) & 1;
    SA sa; SB sb;
    sa.init(p.A, amap, p.a_ks, m0, p.M, avec);
    sb.init(p.B, bmap, p.b_ks, n0, p.N, bvec);

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int fi = lane & 15, fq = lane >> 4;

    f32x4 acc[TM][TN];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nk = (kend - kbeg + BK - 1) / BK;
    if (nk > 0) {
        sa.load(kbeg, kend); sb.load(kbeg, kend);
        sa.store(smem); sb.store(smem + SA::FLOATS);
    }
    __syncthreads();
    for (int it = 0; it < nk; ++it) {
        const float* As = smem + (it & 1) * STAGE;
        const float* Bs = As + SA::FLOATS;
        const int k0 = kbeg + it * BK;
        const bool more = it + 1 < nk;
        if (more) { sa.load(k0 + BK, kend); sb.load(k0 + BK, kend); }
        const int nchunk = (kend - k0 > 16) ? 2 : 1;
        for (int t = 0; t < nchunk; ++t) {
            float af[TM][4], bf[TN][4];
#pragma unroll
            for (int a = 0; a < TM; ++a) read_frag<BM, AKM>(As, (wm * TM + a) * 16, t, fi, fq, af[a]);
#pragma unroll
            for (int b = 0; b < TN; ++b) read_frag<BN, BKM>(Bs, (wn * TN + b) * 16, t, fi, fq, bf[b]);
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int a = 0; a < TM; ++a)
#pragma unroll
                    for (int b = 0; b < TN; ++b)
                        acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[a][u], bf[b][u], acc[a][b], 0, 0, 0);
        }
        if (more) {
            float* An = smem + ((it + 1) & 1) * STAGE;
            sa.store(An); sb.store(An + SA::FLOATS);
        }
        __syncthreads();
    }

    // Epilogue.  C/D map of the 16x16 MFMA: col = lane & 15, row = (lane >> 4) * 4 + reg.
    const RowMap cmap{p.c_grp, p.c_gs, p.c_gmap, p.c_rs};
    const bool relu = p.flags & ICK_GEMM_RELU, accum = p.flags & ICK_GEMM_ACCUM, atomic = p.flags & ICK_GEMM_ATOMIC;
    const bool add_bias = p.bias != nullptr && blockIdx.z == 0;
#pragma unroll
    for (int a = 0; a < TM; ++a) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = m0 + (wm * TM + a) * 16 + fq * 4 + r;
            if (row >= p.M) continue;
            float* crow = p.C + cmap(row);
#pragma unroll
            for (int b = 0; b < TN; ++b) {
                const int col = n0 + (wn * TN + b) * 16 + fi;
                if (col >= p.N) continue;
                float v = acc[a][b][r] * p.alpha;
                if (add_bias) v += p.bias[col];
                if (relu) v = fmaxf(v, 0.f);
                if (atomic) atomicAdd(crow + col, v);
                else if (accum) crow[col] += v;
                else crow[col] = v;
            }
        }
    }
}

template <int WM, int WN, int TM, int TN, bool AKM, bool BKM>
int launch(const ick_gemm_args& a, hipStream_t s) {
    constexpr int BM = WM * TM * 16, BN = WN * TN * 16;
    constexpr int STAGE = Stager<BM, AKM>::FLOATS + Stager<BN, BKM>::FLOATS;
    constexpr size_t smem = 2 * STAGE * sizeof(float);
    const int tiles_m = ceil_div(a.M, BM), tiles_n = ceil_div(a.N, BN);
    int split = a.split_k > 1 ? a.split_k : 1;
    int kchunk = ceil_div(ceil_div(a.K, split), BK) * BK;
    split = ceil_div(a.K, kchunk);
    auto kern = gemm_kernel<WM, WN, TM, TN, AKM, BKM>;
    static bool attr_set = false;
    if (!attr_set && smem > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    hipLaunchKernelGGL(kern, dim3(tiles_m * tiles_n, 1, split), dim3(256), smem, s, a, tiles_m, tiles_n, kchunk);
    ICK_LAUNCH_RET();
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace
}  // namespace ick

extern "C" int ick_gemm(const ick_gemm_args* in, void* stream) {
    using namespace ick;
    if (!in) return ICK_EINVAL;
    ick_gemm_args a = *in;
    ICK_CHECK_ARG(a.A && a.B && a.C);
    ICK_CHECK_ARG(a.M > 0 && a.N > 0 && a.K > 0);
    const bool akm = a.a_rs == 1 && a.a_ks != 1, bkm = a.b_rs == 1 && a.b_ks != 1;
    ICK_CHECK_ARG((a.a_rs == 1) || (a.a_ks == 1));
    ICK_CHECK_ARG((a.b_rs == 1) || (a.b_ks == 1));
    if (a.split_k > 1) ICK_CHECK_ARG(a.flags & ICK_GEMM_ATOMIC);
    if (a.a_grp <= 0) { a.a_grp = 0; a.a_gmap = nullptr; }
    if (a.c_grp <= 0) { a.c_grp = 0; a.c_gmap = nullptr; }
    // 16-byte vector staging is legal when every float4 the stager forms is aligned and in bounds.
    bool avec, bvec;
    if (akm) {
        avec = aligned16(a.A) && a.a_ks % 4 == 0 && a.M % 4 == 0 && (a.a_grp == 0 || (a.a_grp % 4 == 0 && a.a_gs % 4 == 0));
    } else {
        avec = aligned16(a.A) && a.a_rs % 4 == 0 && a.K % 4 == 0 && (a.a_grp == 0 || a.a_gs % 4 == 0);
    }
    if (bkm) bvec = aligned16(a.B) && a.b_ks % 4 == 0 && a.N % 4 == 0;
    else bvec = aligned16(a.B) && a.b_rs % 4 == 0 && a.K % 4 == 0;
    a.flags = (a.flags & 0xff) | (avec ? 0x100 : 0) | (bvec ? 0x200 : 0);
    hipStream_t s = (hipStream_t)stream;

    // Tile selection: 64x64 per wave on large problems; N<=320 keeps 4 waves stacked along M so a
    // 300-wide output costs 5 x 64 columns instead of 3 x 128; small problems use small tiles so
    // that enough workgroups exist to cover the 256 CUs.
    const int64_t tiles_big = (int64_t)ceil_div(a.M, 128) * ceil_div(a.N, 128);
    const int64_t work = (int64_t)a.M * a.N;
#define ICK_DISPATCH(WM, WN, TM, TN)                                            \
    do {                                                                        \
        if (!akm && !bkm) return launch<WM, WN, TM, TN, false, false>(a, s);    \
        if (akm && !bkm) return launch<WM, WN, TM, TN, true, false>(a, s);      \
        if (!akm && bkm) return launch<WM, WN, TM, TN, false, true>(a, s);      \
        return launch<WM, WN, TM, TN, true, true>(a, s);                        \
    } while (0)
    if (work >= (int64_t)256 * 128 * 128 && a.N > 320) ICK_DISPATCH(2, 2, 4, 4);   // 128 x 128
    if (work >= (int64_t)200 * 256 * 64) ICK_DISPATCH(4, 1, 4, 4);                  // 256 x 64
    if (tiles_big >= 96) ICK_DISPATCH(2, 2, 4, 4);
    if (work >= (int64_t)256 * 64 * 64) ICK_DISPATCH(2, 2, 2, 2);                   // 64 x 64
    ICK_DISPATCH(2, 2, 1, 1);                                                       // 32 x 32
#undef ICK_DISPATCH
}
