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
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <type_traits>

#include "common.h"
#include "gemm_common.h"

// Diagnostic build (-DICK_GEMM_STAMPS, tools/debug/gemm_stamps.py): every workgroup of a single-problem launch records
// where it ran (HW_ID) and the realtime clock (100 MHz) at its start, after its first LDS stage, after its K loop and
// at its end.  Compiled out of the product library.
#ifdef ICK_GEMM_STAMPS
__device__ unsigned long long ick_gemm_stamps[16384][8];
#define ICK_GSTAMP(i)                                                                                          \
    do {                                                                                                       \
        if (threadIdx.x == 0 && stamp_id < 16384) ick_gemm_stamps[stamp_id][i] = __builtin_readcyclecounter(); \
    } while (0)
extern "C" int ick_debug_read_gemm_stamps(unsigned long long* out, int n) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(ick_gemm_stamps), sizeof(unsigned long long) * 8 * n);
}
#else
#define ICK_GSTAMP(i)
#endif

namespace ick {

int launch_gemm_ps(const ick_gemm_args& a, bool akm, int tile, int tiles_m, int tiles_n, int kchunk, int split, int a_nt,
                   hipStream_t s);      // gemm_ps.hip
void gemm_ps_tile_dims(int tile, int* bm, int* bn, int* wgs_per_cu);
int gemm_ps_tile_count();

namespace {


// Global -> register -> LDS stager for one operand tile of R rows x BK k.
// VEC: every row offset / K / base pointer is 16-byte friendly.  The tile is fetched with
// buffer_load_dwordx4 through a raw buffer descriptor whose num_records is the operand's byte extent:
// the per-lane byte offset of a row is fixed for the whole K loop (rows outside the matrix get an offset
// beyond the extent), the K position is the scalar offset, and the hardware range check returns zeros
// for anything out of bounds -- no address arithmetic, selects or branches in the loop, so the loads of
// two slices stay in flight behind the MFMAs.  Only the last, partial K slice is masked (at LDS-store time).
// !VEC is the generic element-wise fallback for ragged / unaligned operands.
// k-major planes (bank search): 64-row tiles take (2 R + 64)-byte k lines and swap neighbouring 16-row
// blocks in k lines 8-15 / 24-31; 128-row tiles take (2 R + 32)-byte k lines (the 128 x 64 tile of Encoder.conv1 then
// needs exactly 80 KB: two workgroups per CU) and swap blocks four apart.  kmajor_swap = log2 of the swap distance in
// 4-row units.
constexpr int kmajor_kline(int R) { return R >= 128 ? 2 * R + 32 : 2 * R + 64; }
constexpr int kmajor_swap(int R) { return R >= 128 ? 4 : 2; }
#ifdef ICK_KM_NOSWAP   // A/B: no block swap (k lines q and q + 8 of a transposing read then share banks)
constexpr int kKmSwapOn = 0;
#else
constexpr int kKmSwapOn = 1;
#endif

template <int R, bool KM, bool VEC, int BKT, int NT = 256, bool SPL = false>
struct Stager {
    static constexpr int NP = R * BKT / (4 * NT);    // float4 per thread (NT threads)
    static_assert(NP >= 1, "tile too small for the workgroup");
    static_assert(!SPL || (VEC && BKT == 32), "split operands: vector staging, 32-k slices");
    static constexpr int CH = KM ? R / 4 : BKT / 4;  // float4 chunks along the contiguous dim
    static constexpr int LD = KM ? R + 4 : BKT + 4;
    static constexpr int KLINE = kmajor_kline(R);    // SPL, k-major: bytes per k line of a plane
    static constexpr int PLANE = KM ? BKT * KLINE : R * 64;   // SPL: bytes per bf16 plane
    static constexpr int FLOATS = SPL ? 3 * PLANE / 4 : (KM ? BKT * LD : R * LD);
    static constexpr int KP = NT / CH;               // k-major: k lines covered per pass
    static constexpr int RP = NT / CH;               // k-contiguous: rows covered per pass
    const float* base;
    __amdgpu_buffer_rsrc_t rsrc;
    uint32_t voff[NP];         // VEC: byte offset of this thread's float4 of pass j at k = 0
    int64_t off[KM ? 4 : NP];  // !VEC: clamped element offsets of the rows
    int64_t ks;
    int c, r0;
    uint32_t ok;               // !VEC: validity bits of the rows behind off[]
    float4 v[2][NP];           // two register sets: the loads of slice i+2 fly while slice i+1 waits to be stored
    float4 cs;                 // SPL, k-major: this thread's share of the column sums (rows 4c..4c+3 over its k lines)
    bool want_cs;

    __device__ __forceinline__ void init(const float* p, const RowMap& m, int64_t kstride, int tile_row0, int rows,
                                         int64_t extent) {
        base = p; ks = kstride;
        const int t = threadIdx.x;
        c = t % CH;
        r0 = t / CH;
        ok = 0;
        cs = float4{0.f, 0.f, 0.f, 0.f};
        want_cs = false;
        if constexpr (VEC) {
            rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p), (short)0, (int)(extent * 4), 0x00020000);
            if constexpr (KM) {
                const int gr = tile_row0 + 4 * c;
                const int one[1] = {min(gr, rows - 1)};
                int64_t mo[1];
                map_rows<1>(m, one, mo);
                const uint32_t ro = (uint32_t)(mo[0] * 4);
#pragma unroll
                for (int j = 0; j < NP; ++j)
                    voff[j] = gr < rows ? ro + (uint32_t)((int64_t)(r0 + KP * j) * ks * 4) : kOobOffset;
            } else {
                int grs[NP];
                int64_t mo[NP];
#pragma unroll
                for (int j = 0; j < NP; ++j) grs[j] = min(tile_row0 + r0 + RP * j, rows - 1);
                map_rows<NP>(m, grs, mo);
#pragma unroll
                for (int j = 0; j < NP; ++j)
                    voff[j] = tile_row0 + r0 + RP * j < rows ? (uint32_t)((mo[j] + 4 * c) * 4) : kOobOffset;
            }
        } else if constexpr (KM) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int gr = tile_row0 + 4 * c + q;
                if (gr < rows) ok |= 1u << q;
                off[q] = m(min(gr, rows - 1));
            }
        } else {
#pragma unroll
            for (int j = 0; j < NP; ++j) {
                const int gr = tile_row0 + r0 + RP * j;
                if (gr < rows) ok |= 1u << j;
                off[j] = m(min(gr, rows - 1));
            }
        }
    }

    // Issue the global loads of slice [k0, k0+BK); nothing here depends on loaded data.
    template <int SET>
    __device__ __forceinline__ void load(int k0, int kend) {
        if constexpr (VEC) {
            const int soff = KM ? (int)((int64_t)k0 * ks * 4) : k0 * 4;   // wave-uniform: scalar offset
#pragma unroll
            for (int j = 0; j < NP; ++j)
                v[SET][j] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)voff[j], soff, 0));
        } else if constexpr (KM) {
#pragma unroll
            for (int j = 0; j < NP; ++j) {
                const int k = k0 + r0 + KP * j;
                const int64_t ko = (int64_t)(k < kend ? k : 0) * ks;
                v[SET][j].x = base[off[0] + ko]; v[SET][j].y = base[off[1] + ko];
                v[SET][j].z = base[off[2] + ko]; v[SET][j].w = base[off[3] + ko];
            }
        } else {
            const int k = k0 + 4 * c;
#pragma unroll
            for (int j = 0; j < NP; ++j) {
                v[SET][j].x = base[off[j] + (k + 0 < kend ? k + 0 : 0)];
                v[SET][j].y = base[off[j] + (k + 1 < kend ? k + 1 : 0)];
                v[SET][j].z = base[off[j] + (k + 2 < kend ? k + 2 : 0)];
                v[SET][j].w = base[off[j] + (k + 3 < kend ? k + 3 : 0)];
            }
        }
    }

    // Write the slice to LDS; zero what lies beyond kend (VEC: rows outside the matrix are already zero).
    template <int SET>
    __device__ __forceinline__ void store(float* lds, int k0, int kend) {
        const bool tail = k0 + BKT > kend;   // uniform: only the last slice of the K range needs masking
        if constexpr (KM) {
#pragma unroll
            for (int j = 0; j < NP; ++j) {
                float4 x = v[SET][j];
                if (!VEC || tail) {
                    const bool kv = k0 + r0 + KP * j < kend;
                    const uint32_t m = VEC ? 0xfu : ok;
                    x.x = (kv && (m & 1u)) ? x.x : 0.f; x.y = (kv && (m & 2u)) ? x.y : 0.f;
                    x.z = (kv && (m & 4u)) ? x.z : 0.f; x.w = (kv && (m & 8u)) ? x.w : 0.f;
                }
                if constexpr (SPL) {
                    if (want_cs) { cs.x += x.x; cs.y += x.y; cs.z += x.z; cs.w += x.w; }
                    uint32_t h0, m0, l0, h1, m1, l1;
                    split3(x.x, x.y, h0, m0, l0);
                    split3(x.z, x.w, h1, m1, l1);
                    const int kl = r0 + KP * j;
                    char* at = reinterpret_cast<char*>(lds) + kl * KLINE + 8 * (c ^ ((kKmSwapOn * ((kl >> 3) & 1)) << kmajor_swap(R)));
                    *reinterpret_cast<uint2*>(at) = uint2{h0, h1};
                    *reinterpret_cast<uint2*>(at + PLANE) = uint2{m0, m1};
                    *reinterpret_cast<uint2*>(at + 2 * PLANE) = uint2{l0, l1};
                } else {
                    *reinterpret_cast<float4*>(lds + (r0 + KP * j) * LD + 4 * c) = x;
                }
            }
        } else {
            const int k = k0 + 4 * c;
#pragma unroll
            for (int j = 0; j < NP; ++j) {
                float4 x = v[SET][j];
                if (!VEC || tail) {
                    const bool rv = VEC ? true : (bool)((ok >> j) & 1u);
                    x.x = (rv && k + 0 < kend) ? x.x : 0.f; x.y = (rv && k + 1 < kend) ? x.y : 0.f;
                    x.z = (rv && k + 2 < kend) ? x.z : 0.f; x.w = (rv && k + 3 < kend) ? x.w : 0.f;
                }
                if constexpr (SPL) {
                    uint32_t h0, m0, l0, h1, m1, l1;
                    split3(x.x, x.y, h0, m0, l0);
                    split3(x.z, x.w, h1, m1, l1);
                    const int row = r0 + RP * j;
                    char* at = reinterpret_cast<char*>(lds) + row * 64 + 16 * ((c >> 1) ^ kc_swz(row)) + 8 * (c & 1);
                    *reinterpret_cast<uint2*>(at) = uint2{h0, h1};
                    *reinterpret_cast<uint2*>(at + PLANE) = uint2{m0, m1};
                    *reinterpret_cast<uint2*>(at + 2 * PLANE) = uint2{l0, l1};
                } else {
                    *reinterpret_cast<float4*>(lds + (r0 + RP * j) * LD + 4 * c) = x;
                }
            }
        }
    }
};

// One 16-row fragment for the four MFMA steps of k-chunk t.
template <int R, bool KM, int BKT>
__device__ __forceinline__ void read_frag(const float* lds, int row0, int t, int i, int q, float (&f)[4]) {
    if constexpr (KM) {
        constexpr int LD = R + 4;
#pragma unroll
        for (int u = 0; u < 4; ++u) f[u] = lds[(16 * t + 4 * q + u) * LD + row0 + i];
    } else {
        constexpr int LD = BKT + 4;
        const float4 x = *reinterpret_cast<const float4*>(lds + (row0 + i) * LD + 16 * t + 4 * q);
        f[0] = x.x; f[1] = x.y; f[2] = x.z; f[3] = x.w;
    }
}

// SPL: the three bf16 planes of one 16-row x 32-k MFMA operand (see the layout notes above).
template <int R, bool KM>
__device__ __forceinline__ void read_frag_spl(const float* lds, int row0, int i, int q, bf16x8_t (&f)[3]) {
    const char* base = reinterpret_cast<const char*>(lds);
    if constexpr (KM) {
        constexpr int KLINE = kmajor_kline(R), PLANE = 32 * KLINE;
        const char* at = base + (8 * q + (i >> 2)) * KLINE + ((row0 ^ ((kKmSwapOn * (q & 1)) << (kmajor_swap(R) + 2))) + 4 * (i & 3)) * 2;
#pragma unroll
        for (int p = 0; p < 3; ++p) {
            const s16x4_t lo4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(at + p * PLANE));
            const s16x4_t hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(at + p * PLANE + 4 * KLINE));
            f[p] = __builtin_bit_cast(bf16x8_t, __builtin_shufflevector(lo4, hi4, 0, 1, 2, 3, 4, 5, 6, 7));
        }
    } else {
        constexpr int PLANE = R * 64;
        const int row = row0 + i;
        const char* at = base + row * 64 + 16 * (q ^ kc_swz(row));
#pragma unroll
        for (int p = 0; p < 3; ++p)
            f[p] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const u32x4_t*>(at + p * PLANE));
    }
}

// Split-K problems whose K split count is a multiple of 8 (the weight gradients: small outputs, long reductions):
// workgroups are dispatched round robin over the 8 XCDs in linear order, so XCD x is given the whole tile grid of the K
// splits x, x + 8, ...: every XCD then streams only its own K range of both operands and the re-reads by the other
// tiles of the grid hit its L2 (PMC, tools/pmc_traffic.sh: the cross-K/V weight gradient moved 215-300 MB per launch
// for ~60 MB of operands with the tile-major order).  Returns false (nothing changed) for other split counts.
__device__ __forceinline__ bool split_major(int& bid, int& zid, int nt, int nsplit) {
    if (nsplit < 8 || (nsplit & 7) != 0) return false;
    const int lin = bid + nt * zid;
    const int x = lin & 7, j = lin >> 3;
    const int r = j / nt;
    zid = x + 8 * r;
    bid = j - r * nt;
    return true;
}

// One output tile: workgroup `bid` of the tiles_m x tiles_n grid of problem p, K slice `zid`.
template <int WM, int WN, int TM, int TN, bool AKM, bool BKM, bool VEC, int BKT, bool SPL = false, int LSTG = 2>
__device__ __forceinline__ void gemm_tile(const ick_gemm_args& p, int tiles_m, int tiles_n, int kchunk, int bid,
                                          int zid, float* smem, bool xcd_remap = true) {
    constexpr int BM = WM * TM * 16, BN = WN * TN * 16;
    constexpr int BK = BKT;
    using SA = Stager<BM, AKM, VEC, BKT, WM * WN * 64, SPL>;
    using SB = Stager<BN, BKM, VEC, BKT, WM * WN * 64, SPL>;
    constexpr int STAGE = SA::FLOATS + SB::FLOATS;
#ifdef ICK_GEMM_STAMPS
    const int stamp_id = bid + zid * tiles_m * tiles_n;
    if (threadIdx.x == 0 && stamp_id < 16384) {
        unsigned hw;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        ick_gemm_stamps[stamp_id][7] = ((unsigned long long)xcc << 32) | hw;
    }
#endif
    ICK_GSTAMP(0);

    // XCD-aware tile order (blocks b, b+8, ... share an XCD).
    const int nwg = tiles_m * tiles_n;
    if (xcd_remap) {
        const int xcd = bid & 7, qq = nwg >> 3, rr = nwg & 7;
        bid = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + (bid >> 3);
    }
    int tm, tn;
    if (tiles_m <= tiles_n) { tm = bid % tiles_m; tn = bid / tiles_m; }
    else { tn = bid % tiles_n; tm = bid / tiles_n; }
    const int m0 = tm * BM, n0 = tn * BN;
    const int kbeg = zid * kchunk;
    const int kend = min(p.K, kbeg + kchunk);

    const RowMap amap{p.a_grp, p.a_gs, p.a_gmap, p.a_rs};
    const RowMap bmap{0, 0, nullptr, p.b_rs};
    SA sa; SB sb;
    sa.init(p.A, amap, p.a_ks, m0, p.M, p.a_extent);
    const bool only = (p.flags & ICK_GEMM_COLSUM_ONLY) != 0;   // uniform: column sums of A, no product
    sb.init(only ? p.A : p.B, bmap, p.b_ks, n0, p.N, p.b_extent);

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int fi = lane & 15, fq = lane >> 4;

    f32x4 acc[TM][TN];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nk = (kend - kbeg + BK - 1) / BK;
    const bool colsum = AKM && p.colsum_a != nullptr && tn == 0;   // uniform
    float csum = 0.f;
    if constexpr (SPL && AKM) sa.want_cs = colsum;    // split planes in LDS: the column sums are taken from the registers
    // Software pipeline, two slices deep: slice i is in LDS buffer i&1, slice i+1 is in flight (or landed)
    // in register set (i+1)&1 and is written to the other LDS buffer after the MFMAs of slice i, slice i+2
    // is requested into register set i&1 before them -- every global load has two MFMA phases to land.
    if (nk > 0) {
        sa.template load<0>(kbeg, kend);
        if (!only) sb.template load<0>(kbeg, kend);
        if (nk > 1) { sa.template load<1>(kbeg + BK, kend); if (!only) sb.template load<1>(kbeg + BK, kend); }
        sa.template store<0>(smem, kbeg, kend);
        if (!only) sb.template store<0>(smem + SA::FLOATS, kbeg, kend);
    }
    __syncthreads();
    ICK_GSTAMP(1);
    auto phase = [&](int it, auto cur) {
        constexpr int CUR = decltype(cur)::value, NXT = 1 - CUR;
        // LSTG = 1: one LDS buffer (a second workgroup on the CU overlaps its phases with ours instead of a second
        // buffer); the register sets still hold two slices in flight
        const float* As = smem + (LSTG == 2 ? CUR * STAGE : 0);
        const float* Bs = As + SA::FLOATS;
        const int k0 = kbeg + it * BK;
        if (it + 2 < nk) { sa.template load<CUR>(k0 + 2 * BK, kend); if (!only) sb.template load<CUR>(k0 + 2 * BK, kend); }
        if constexpr (SPL) {
            if (!only) {
                // one 16x16x32 block per tile and slice, six bf16 products (smallest first)
                bf16x8_t af[TM][3];
#pragma unroll
                for (int a = 0; a < TM; ++a) read_frag_spl<BM, AKM>(As, (wm * TM + a) * 16, fi, fq, af[a]);
                if constexpr (TM * TN <= 8) {
                    // every fragment is requested before the first MFMA (reading B column by column exposes the LDS
                    // latency once per column: Encoder.conv1 115 -> 148 us)
                    bf16x8_t bf[TN][3];
#pragma unroll
                    for (int b = 0; b < TN; ++b) read_frag_spl<BN, BKM>(Bs, (wn * TN + b) * 16, fi, fq, bf[b]);
#pragma unroll
                    for (int a = 0; a < TM; ++a)
#pragma unroll
                        for (int b = 0; b < TN; ++b) {
                            f32x4 c = acc[a][b];
                            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[a][0], bf[b][2], c, 0, 0, 0);
                            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[a][2], bf[b][0], c, 0, 0, 0);
                            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[a][1], bf[b][1], c, 0, 0, 0);
                            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[a][0], bf[b][1], c, 0, 0, 0);
                            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[a][1], bf[b][0], c, 0, 0, 0);
                            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[a][0], bf[b][0], c, 0, 0, 0);
                            acc[a][b] = c;
                        }
                } else {
                    // 64 x 64 wave tile: the B fragments of one tile column at a time, the next column requested
                    // before the MFMAs of this one (96 fragment registers otherwise)
                    bf16x8_t bf[2][3];
                    read_frag_spl<BN, BKM>(Bs, (wn * TN) * 16, fi, fq, bf[0]);
#pragma unroll
                    for (int b = 0; b < TN; ++b) {
                        if (b + 1 < TN) read_frag_spl<BN, BKM>(Bs, (wn * TN + b + 1) * 16, fi, fq, bf[(b + 1) & 1]);
#pragma unroll
                        for (int a = 0; a < TM; ++a) {
                            f32x4 c = acc[a][b];
                            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[a][0], bf[b & 1][2], c, 0, 0, 0);
                            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[a][2], bf[b & 1][0], c, 0, 0, 0);
                            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[a][1], bf[b & 1][1], c, 0, 0, 0);
                            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[a][0], bf[b & 1][1], c, 0, 0, 0);
                            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[a][1], bf[b & 1][0], c, 0, 0, 0);
                            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[a][0], bf[b & 1][0], c, 0, 0, 0);
                            acc[a][b] = c;
                        }
                    }
                }
            }
        }
        const int nchunk = (only || SPL) ? 0 : min(BK / 16, (kend - k0 + 15) >> 4);
        for (int t = 0; t < nchunk; ++t) {
            float af[TM][4], bf[TN][4];
#pragma unroll
            for (int a = 0; a < TM; ++a) read_frag<BM, AKM, BKT>(As, (wm * TM + a) * 16, t, fi, fq, af[a]);
#pragma unroll
            for (int b = 0; b < TN; ++b) read_frag<BN, BKM, BKT>(Bs, (wn * TN + b) * 16, t, fi, fq, bf[b]);
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int a = 0; a < TM; ++a)
#pragma unroll
                    for (int b = 0; b < TN; ++b)
                        acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[a][u], bf[b][u], acc[a][b], 0, 0, 0);
        }
        if constexpr (AKM && !SPL) {
            // bias-gradient fusion: column sums of the k-major A operand (dY of a weight-gradient GEMM), taken
            // from the tile already in LDS by the workgroups of the first tile column
            if (colsum && threadIdx.x < BM) {
#pragma unroll 8
                for (int k = 0; k < BK; ++k) csum += As[k * SA::LD + threadIdx.x];
            }
        }
        if constexpr (LSTG == 1) __syncthreads();      // every wave is done with the slice in LDS
        if (it + 1 < nk) {
            float* An = smem + (LSTG == 2 ? NXT * STAGE : 0);
            sa.template store<NXT>(An, k0 + BK, kend);
            if (!only) sb.template store<NXT>(An + SA::FLOATS, k0 + BK, kend);
        }
        __syncthreads();
    };
    for (int it = 0; it < nk; it += 2) {
        phase(it, std::integral_constant<int, 0>{});
        if (it + 1 < nk) phase(it + 1, std::integral_constant<int, 1>{});
    }

    ICK_GSTAMP(2);
    if constexpr (SPL && AKM) {
        if (colsum) {      // uniform.  The threads' shares ([k-line group r0][4 rows]) meet in LDS (free after the last barrier)
            *reinterpret_cast<float4*>(smem + sa.r0 * BM + 4 * sa.c) = sa.cs;
            __syncthreads();
            if (threadIdx.x < BM) {
#pragma unroll 4
                for (int r = 0; r < SA::KP; ++r) csum += smem[r * BM + threadIdx.x];
            }
        }
    }
    if (colsum && threadIdx.x < BM && m0 + (int)threadIdx.x < p.M) atomicAdd(p.colsum_a + m0 + threadIdx.x, csum);
    if (only) return;

    gemm_epilogue<TM, TN>(p, acc, m0, n0, wm, wn, fi, fq, zid);
    ICK_GSTAMP(3);
}

template <int WM, int WN, int TM, int TN, bool AKM, bool BKM, bool VEC, int BKT, bool SPL, int LSTG = 2>
__global__ __launch_bounds__(WM * WN * 64) void gemm_kernel(ick_gemm_args p, int tiles_m, int tiles_n, int kchunk) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    if constexpr (TM == 1) chain_priority();     // 32 x 32 tiles: the chain GEMMs (single launches, not the grouped weight gradients)
    const int nt = tiles_m * tiles_n;
    int bid = blockIdx.x, zid = blockIdx.z;
    const bool by_split = split_major(bid, zid, nt, gridDim.z);
    gemm_tile<WM, WN, TM, TN, AKM, BKM, VEC, BKT, SPL, LSTG>(p, tiles_m, tiles_n, kchunk, bid, zid, smem, !by_split);
}

// Several independent problems of the same kernel configuration in one launch (the weight-gradient GEMMs of a
// layer: each alone is a ~15 us latency-bound launch of a few hundred workgroups).
constexpr int kGroupMax = 8;
struct GroupArgs {
    int count;
    int wg_end[kGroupMax];      // exclusive prefix sums of workgroups (tiles * K splits, each rounded up to 8 so that a
                                // problem's first workgroup lands on XCD 0: the XCD-aware orders assume that)
    int tiles_m[kGroupMax], tiles_n[kGroupMax], kchunk[kGroupMax], split[kGroupMax];
    ick_gemm_args g[kGroupMax];
};

template <int WM, int WN, int TM, int TN, bool AKM, bool BKM, bool VEC, int BKT, bool SPL>
__global__ __launch_bounds__(256) void gemm_group_kernel(GroupArgs ga) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    int gi = 0;
    while (gi + 1 < ga.count && (int)blockIdx.x >= ga.wg_end[gi]) ++gi;
    const int local = blockIdx.x - (gi > 0 ? ga.wg_end[gi - 1] : 0);
    const int nt = ga.tiles_m[gi] * ga.tiles_n[gi];
    if (local >= nt * ga.split[gi]) return;      // padding
    int zid = local / nt, bid = local - zid * nt;
    const bool by_split = split_major(bid, zid, nt, ga.split[gi]);
    gemm_tile<WM, WN, TM, TN, AKM, BKM, VEC, BKT, SPL>(ga.g[gi], ga.tiles_m[gi], ga.tiles_n[gi], ga.kchunk[gi], bid, zid,
                                                       smem, !by_split);
}

// Host-side plan of one problem: validated arguments + kernel configuration.
struct Plan {
    ick_gemm_args a;
    bool akm, bkm, vec;
    bool big;                    // 64 x 64 tiles (else 32 x 32)
    bool wide;                   // 128 x 64 tiles, 8 waves (single launches only)
    bool spl;                    // split-bf16 products (64 x 64 and 128 x 64 tiles of the vector path)
    bool xl;                     // 128 x 128 tiles, 8 waves, split products only (single launches only; implies wide)
    bool ps;                     // B read from its pre-split copy (gemm_ps.hip) ...
    int ps_tile, ps_nt;          // ... on tile shape ps_tile (gemm_ps_tile_dims), A loads non-temporal when ps_nt
    int tiles_m, tiles_n, kchunk, split;
};

template <int WM, int WN, int TM, int TN, bool AKM, bool BKM, bool VEC, bool SPL, int LSTG = 2>
int launch_tile_s(const Plan& pl, hipStream_t s) {
    constexpr int BM = WM * TM * 16, BN = WN * TN * 16, NT = WM * WN * 64;
    constexpr int STAGE = Stager<BM, AKM, VEC, 32, NT, SPL>::FLOATS + Stager<BN, BKM, VEC, 32, NT, SPL>::FLOATS;
    constexpr size_t smem = LSTG * STAGE * sizeof(float);
    static_assert(smem <= 160 * 1024, "tile exceeds the LDS of a CU");
    const size_t lds = smem;
    if (lds > 64 * 1024) {
        static LdsAttrOnce attr;
        if (int e = attr.ensure(reinterpret_cast<const void*>(gemm_kernel<WM, WN, TM, TN, AKM, BKM, VEC, 32, SPL, LSTG>), 160 * 1024))
            return e;
    }
    const int gx = pl.tiles_m * pl.tiles_n;
    hipLaunchKernelGGL((gemm_kernel<WM, WN, TM, TN, AKM, BKM, VEC, 32, SPL, LSTG>), dim3(gx, 1, pl.split),
                       dim3(NT), lds, s, pl.a, pl.tiles_m, pl.tiles_n, pl.kchunk);
    ICK_LAUNCH_RET();
}
template <int WM, int WN, int TM, int TN, bool AKM, bool BKM, bool VEC>
int launch_tile(const Plan& pl, hipStream_t s) {
    if constexpr (VEC && TM > 1) {
        if (pl.spl) return launch_tile_s<WM, WN, TM, TN, AKM, BKM, VEC, true>(pl, s);
    }
    return launch_tile_s<WM, WN, TM, TN, AKM, BKM, VEC, false>(pl, s);
}
template <int TM, int TN, bool AKM, bool BKM, bool VEC>
int launch_one(const Plan& pl, hipStream_t s) { return launch_tile<2, 2, TM, TN, AKM, BKM, VEC>(pl, s); }
template <int TM, int TN, bool AKM, bool BKM, bool VEC>
int launch_wide(const Plan& pl, hipStream_t s) { return launch_tile<4, 2, TM, TN, AKM, BKM, VEC>(pl, s); }   // 8 waves
// 128 x 128, four waves of 64 x 64, one LDS buffer: two workgroups per CU
template <int TM, int TN, bool AKM, bool BKM, bool VEC>
int launch_xl4(const Plan& pl, hipStream_t s) { return launch_tile_s<2, 2, TM, TN, AKM, BKM, VEC, true, 1>(pl, s); }

template <int TM, int TN, bool AKM, bool BKM, bool SPL>
int launch_group_s(const Plan* const* pls, int n, hipStream_t s) {
    constexpr int BM = 2 * TM * 16, BN = 2 * TN * 16;
    constexpr int STAGE = Stager<BM, AKM, true, 32, 256, SPL>::FLOATS + Stager<BN, BKM, true, 32, 256, SPL>::FLOATS;
    constexpr size_t smem = 2 * STAGE * sizeof(float);
    GroupArgs ga;
    ga.count = n;
    int total = 0;
    for (int i = 0; i < n; ++i) {
        total += (pls[i]->tiles_m * pls[i]->tiles_n * pls[i]->split + 7) & ~7;
        ga.wg_end[i] = total;
        ga.tiles_m[i] = pls[i]->tiles_m; ga.tiles_n[i] = pls[i]->tiles_n; ga.kchunk[i] = pls[i]->kchunk;
        ga.split[i] = pls[i]->split;
        ga.g[i] = pls[i]->a;
    }
    for (int i = n; i < kGroupMax; ++i) {
        ga.wg_end[i] = total; ga.tiles_m[i] = ga.tiles_n[i] = 1; ga.kchunk[i] = 32; ga.split[i] = 0; ga.g[i] = pls[0]->a;
    }
    static_assert(smem <= 160 * 1024, "tile exceeds the LDS of a CU");
    const size_t lds = smem;
    if (lds > 64 * 1024) {      // split planes of two k-major operands: 72 KB, above the default dynamic-LDS limit
        static LdsAttrOnce attr;
        if (int e = attr.ensure(reinterpret_cast<const void*>(gemm_group_kernel<2, 2, TM, TN, AKM, BKM, true, 32, SPL>), 160 * 1024))
            return e;
    }
    hipLaunchKernelGGL((gemm_group_kernel<2, 2, TM, TN, AKM, BKM, true, 32, SPL>), dim3(total), dim3(256), lds, s, ga);
    ICK_LAUNCH_RET();
}
template <int TM, int TN, bool AKM, bool BKM>
int launch_group(const Plan* const* pls, int n, hipStream_t s) {
    if constexpr (TM > 1) {
        if (pls[0]->spl) return launch_group_s<TM, TN, AKM, BKM, true>(pls, n, s);
    }
    return launch_group_s<TM, TN, AKM, BKM, false>(pls, n, s);
}

// Split-bf16 products for the 64 x 64 / 128 x 64 / 128 x 128 tiles.  Mode 1 (the DEFAULT since round 4: the reference-pinned
// parity suite runs in every mode, tests/conftest.py gemm_split) takes them where they are faster than the exact fp32 MFMA
// (B operand k-contiguous or pre-split: the forward GEMMs, the feature projection, the vocabulary data gradient); 2 takes
// them for every large-tile problem (the k-major forms gain little: tools/gemm_split_bench.py); ICK_GEMM_SPLIT=0 /
// ick_set_gemm_split(0): every product on v_mfma_f32_16x16x4_f32.
int g_gemm_split = -1;
inline int gemm_split_mode() {
    if (g_gemm_split < 0) {
        const char* e = getenv("ICK_GEMM_SPLIT");
        g_gemm_split = e ? std::min(2, std::max(0, atoi(e))) : 1;
    }
    return g_gemm_split;
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// Validation + kernel selection for one problem (shared by ick_gemm and ick_gemm_grouped).
int make_plan(const ick_gemm_args* in, Plan& pl, int force_big = 0) {
    if (!in) return ICK_EINVAL;
    ick_gemm_args& a = pl.a;
    a = *in;
    if (a.flags & ICK_GEMM_COLSUM_ONLY) {
        // colsum_a[m] += sum_k A(m,k) and nothing else: B / C / N are ignored (one tile column, no product)
        ICK_CHECK_ARG(a.A && a.colsum_a && a.M > 0 && a.K > 0);
        a.B = a.A; a.C = a.colsum_a; a.bias = nullptr; a.gate = nullptr;
        a.N = 1; a.b_rs = 1; a.b_ks = a.a_ks; a.c_rs = 1; a.c_grp = 0; a.hs_dh = 0; a.b_extent = a.a_extent;
        a.drop_p = 0.f;
        a.flags = ICK_GEMM_COLSUM_ONLY | ICK_GEMM_ATOMIC;
    }
    ICK_CHECK_ARG(a.A && a.B && a.C);
    ICK_CHECK_ARG(a.M > 0 && a.N > 0 && a.K > 0);
    const bool akm = a.a_rs == 1 && a.a_ks != 1, bkm = a.b_rs == 1 && a.b_ks != 1;
    ICK_CHECK_ARG((a.a_rs == 1) || (a.a_ks == 1));
    ICK_CHECK_ARG((a.b_rs == 1) || (a.b_ks == 1));
    if (a.split_k > 1) ICK_CHECK_ARG(a.flags & ICK_GEMM_ATOMIC);
    if (a.colsum_a) ICK_CHECK_ARG(akm);       // column sums come from the k-major A tile
    if (a.gate) ICK_CHECK_ARG(a.hs_dh <= 0 && a.c_grp <= 0 && a.split_k <= 1 && !(a.flags & (ICK_GEMM_ATOMIC | ICK_GEMM_ACCUM)));
    if (a.hs_dh > 0) {
        ICK_CHECK_ARG(a.hs_dhp >= a.hs_dh && a.hs_H > 0 && a.hs_S > 0 && a.hs_s0 >= 0);
        ICK_CHECK_ARG(a.N % (a.hs_H * a.hs_dh) == 0);
        ICK_CHECK_ARG(a.hs_s0 + (a.c_grp > 0 ? a.c_grp : a.M) <= a.hs_S);
    } else {
        a.hs_dh = 0;
    }
    if (a.a_grp <= 0) { a.a_grp = 0; a.a_gmap = nullptr; }
    if (a.c_grp <= 0) { a.c_grp = 0; a.c_gmap = nullptr; }
    // 16-byte vector staging is legal when every float4 the stager forms is aligned and in bounds.
    bool avec, bvec;
    if (akm) {
        avec = aligned16(a.A) && a.a_ks % 4 == 0 && a.M % 4 == 0 && (a.a_grp == 0 || (a.a_grp % 4 == 0 && a.a_gs % 4 == 0));
    } else {
        avec = aligned16(a.A) && a.a_rs % 4 == 0 && a.K % 4 == 0 && (a.a_grp == 0 || a.a_gs % 4 == 0);
    }
    if (a.flags & ICK_GEMM_COLSUM_ONLY) bvec = true;
    else if (bkm) bvec = aligned16(a.B) && a.b_ks % 4 == 0 && a.N % 4 == 0;
    else bvec = aligned16(a.B) && a.b_rs % 4 == 0 && a.K % 4 == 0;
    // operand extents (elements addressable from the base pointer) bound the buffer descriptors of the vector
    // path; without a group map they follow from the strides, with one the caller must state them
    auto extent_of = [](int64_t rows, int64_t rs, int64_t K, int64_t ks, int grp, int64_t gs, bool has_map) -> int64_t {
        if (has_map) return 0;
        if (grp <= 0) return (rows - 1) * rs + (K - 1) * ks + 1;
        const int64_t ng = (rows + grp - 1) / grp;
        return (ng - 1) * gs + (std::min<int64_t>(grp, rows) - 1) * rs + (K - 1) * ks + 1;
    };
    if (a.a_extent <= 0) a.a_extent = extent_of(a.M, a.a_rs, a.K, a.a_ks, a.a_grp, a.a_gs, a.a_gmap != nullptr);
    if (a.b_extent <= 0) a.b_extent = extent_of(a.N, a.b_rs, a.K, a.b_ks, 0, 0, false);
    const int64_t kMaxExtent = ((int64_t)1 << 29) - 64;   // < 2 GiB of floats, so kOobOffset is out of range
    // every float4 of the vector path starts at a multiple of 4 floats, so a bound rounded down to one
    // (views into a larger buffer, e.g. the flat parameter bucket) and capped below 2 GiB never cuts a valid one
    a.a_extent = std::min(a.a_extent & ~(int64_t)3, kMaxExtent - 4);
    a.b_extent = std::min(a.b_extent & ~(int64_t)3, kMaxExtent - 4);
    const int64_t a_need = extent_of(a.M, a.a_rs, a.K, a.a_ks, a.a_grp, a.a_gs, a.a_gmap != nullptr);
    const int64_t b_need = extent_of(a.N, a.b_rs, a.K, a.b_ks, 0, 0, false);
    pl.vec = avec && bvec && a.a_extent >= 4 && a.b_extent >= 4 && a_need <= a.a_extent + 3 && b_need <= a.b_extent + 3;
    pl.akm = akm; pl.bkm = bkm;
    a.flags &= 0xff;
    // Tile selection, measured on MI355X (tools/probes/probe_ops, profiles/r01_*): with exact-fp32 MFMA a 64x64
    // wave tile alone needs ~18 us for K = 300, so latency and occupancy favour 64x64 workgroup tiles (32x32 per
    // wave, 4+ waves per SIMD) on every large shape of this path (128x128 / 256x64 tiles were 5-15 % slower, BK =
    // 64 / 128 no faster than 32).  Small outputs (the chain GEMMs of the layers and their data/weight gradients)
    // take 32x32 tiles so that ~1000 workgroups exist: 7.7 vs 11.7 us for 1280x300x300, 18.5 vs 30.6 us for K = 900;
    // from ~500 64x64 tiles on (cross K/V, vocabulary, feature projection) the larger tile wins (175 vs 212 us).
    // Ragged / unaligned operands (element-wise staging) always use the 64x64 shape.
    const int split_req = a.split_k > 1 ? a.split_k : 1;
    const int64_t wgs64 = (int64_t)ceil_div(a.M, 64) * ceil_div(a.N, 64) * split_req;
    pl.big = !pl.vec || wgs64 >= 512;
    if (force_big > 0) pl.big = true;
    // 128 x 64 tiles, 8 waves: measured +4 % on the feature projection (k-major A, K = 2048, N = 300: 157 -> 151 us),
    // -3 % on the K = 300 shapes, so it is reserved for long-K, narrow-N problems
    pl.wide = pl.vec && pl.big && a.M >= 128 && akm && !bkm && a.N <= 320 && a.M >= 4096 && a.K >= 1024 && split_req == 1;
    pl.spl = pl.vec && pl.big && (gemm_split_mode() == 2 || (gemm_split_mode() == 1 && !bkm));
    pl.xl = false;
    if (pl.spl) {
        // 128 x 128 tiles: twice the products per staged (and split) element.  One workgroup per CU (98 KB of LDS), so
        // the problem must bring several rounds of tiles and waste little of its last tile column
        const int64_t t128 = (int64_t)ceil_div(a.M, 128) * ceil_div(a.N, 128);
        const bool fits = split_req == 1 && a.M >= 128 && a.N >= 128 && t128 >= 512 && ceil_div(a.N, 128) * 128 <= a.N + a.N / 8;
        pl.xl = fits;
        if (pl.xl) pl.wide = true;
    }
    const int BMN = pl.big ? 64 : 32;
    pl.tiles_m = ceil_div(a.M, pl.wide ? 128 : BMN); pl.tiles_n = ceil_div(a.N, pl.xl ? 128 : BMN);
    pl.kchunk = ceil_div(ceil_div(a.K, split_req), 32) * 32;
    pl.split = ceil_div(a.K, pl.kchunk);
    // B pre-split by the caller (b_ps): both operands staged by LDS-DMA, only the A fragments split in the kernel
    // (gemm_ps.hip).  Needs the vector conditions on A (16-byte pieces) and a problem large enough for its tiles.
    pl.ps = false; pl.ps_tile = 0; pl.ps_nt = 0;
    if (a.b_ps != nullptr && gemm_split_mode() >= 1 && avec && aligned16(a.b_ps) && a.a_extent >= 4 &&
        a_need <= a.a_extent + 3 && a.colsum_a == nullptr && !(a.flags & ICK_GEMM_COLSUM_ONLY)) {
        static int ps_tile_env = -2;      // ICK_PS_TILE: one tile shape for every pre-split problem (tools/gemm_ps_bench.py sweeps)
        if (ps_tile_env == -2) { const char* e = getenv("ICK_PS_TILE"); ps_tile_env = e ? atoi(e) : -1; }
        const double flop = 2.0 * a.M * a.N * a.K;
        // a one-column-tile problem (N <= 320, no K split) needs ~180 row tiles of 128 to fill the chip with the pre-split
        // kernel's 128-row tiles (Encoder.conv1 at batch 32 -- 98 workgroups -- took as long as at batch 64).  Below that
        // the 64 x 160 four-wave tile (two per CU) keeps the problem on this kernel as long as it brings >= 128 of them:
        // Encoder.conv1 at batch 32 (cfg5's prefill) 99 us on the stager-split 128 x 64 tiles -> ~35 us less per greedy
        // decode (2.12 -> 2.05 ms, profiles/r05_x_ab_conv1_b32.txt; the 128 x 80 tile: 2.06); smaller problems stay on
        // the stager-split kernel's 128 x 64 tiles (five per row panel)
        const bool narrow = a.N <= 320 && split_req == 1;
        const bool narrow_underfilled = narrow && ceil_div(a.M, 128) * ceil_div(a.N, 160) < 180;
        const bool narrow_small_tile = narrow_underfilled && ceil_div(a.M, 64) * ceil_div(a.N, 160) >= 128;
        // the pre-split copy is addressed through one buffer descriptor: it must stay below 2 GiB
        const bool ps_fits = (int64_t)ceil_div(a.K, 32) * 3 * ceil_div(a.N, 64) * 64 * 64 < ((int64_t)1 << 31);
        if (ps_fits && flop >= 1.0e9 && a.M >= 256 && a.N >= 128 && (!narrow_underfilled || narrow_small_tile || ps_tile_env >= 0)) {
            // Tile choice, measured (tools/gemm_ps_bench.py, profiles/r04_e_gemm_ps_tiles.txt: every tile x every shape):
            // 128 x 128 with two workgroups per CU wins wherever the output is wider than 320 columns (cross K/V 100 us
            // against 129-143 on the other tiles, vocabulary 65 against 80-88); outputs at most 320 wide (Encoder.conv1,
            // the vocabulary data gradient) read their A operand once on 128 x 160 (121 us against 139-187).  Larger tiles
            // and deeper rings bought nothing: every variant settles near 120-135 TFLOP/s fp32-equivalent, ~60 % of what
            // the bf16 pipe holds on random data at the clock the chip keeps under MFMA load (MI355X_MICROARCH.md).
            // (split-K problems -- the cross K/V weight gradient, 600 x 300 over 13 824 rows -- also run best on 128 x 128:
            // 62-67 us against 70-77 on 128 x 160 and 87-93 on the exact 64 x 64 tile, profiles/r04_i_gemm_ps_kv_wgrad.txt)
            // Inside the steps Encoder.conv1 runs beside the context-encoder chain of the other stream, whose 49 KB
            // workgroups cannot share a CU with a 139 KB tile: on 128 x 80 (62 KB, two per CU; alone 2-8 % slower than
            // 128 x 160) the cfg2 train step is 1.748 -> 1.722 ms and the forward pass 0.702 -> 0.693
            // (profiles/r04_y_ab_narrow_tile.txt)
            int best = narrow ? (narrow_small_tile ? 8 : 9) : 1;
            // split-K problems at most 320 columns wide (the vocabulary's data gradient 1280 x 300 over K = 10 000, the
            // cross K/V and vocabulary weight gradients): 128 x 80 as well -- the data gradient's 40 x 12 = 480 workgroups
            // fill the chip's two slots per CU once (train step 1.770 -> 1.741 ms), the weight gradients are unchanged
            if (a.N <= 320 && split_req > 1) best = 9;
            if (ps_tile_env >= 0 && ps_tile_env < gemm_ps_tile_count()) best = ps_tile_env;
            int bm, bn, wpc; gemm_ps_tile_dims(best, &bm, &bn, &wpc);
            pl.ps = true; pl.ps_tile = best;
            pl.spl = true; pl.big = true; pl.wide = pl.xl = false;
            pl.tiles_m = ceil_div(a.M, bm);
            pl.tiles_n = ceil_div(a.N, bn);
            pl.ps_nt = 0;    // non-temporal A loads: measured, no gain (same file)
        }
    }
    return ICK_OK;
}

#define ICK_BY_LAYOUT(FN, TM, TN, VECARGS, ...)                                         \
    do {                                                                                \
        if (!pl.akm && !pl.bkm) return FN<TM, TN, false, false VECARGS>(__VA_ARGS__);   \
        if (pl.akm && !pl.bkm) return FN<TM, TN, true, false VECARGS>(__VA_ARGS__);     \
        if (!pl.akm && pl.bkm) return FN<TM, TN, false, true VECARGS>(__VA_ARGS__);     \
        return FN<TM, TN, true, true VECARGS>(__VA_ARGS__);                             \
    } while (0)
#define ICK_COMMA_TRUE , true
#define ICK_COMMA_FALSE , false

int launch_plan(const Plan& pl, hipStream_t s) {
    if (pl.ps) return launch_gemm_ps(pl.a, pl.akm, pl.ps_tile, pl.tiles_m, pl.tiles_n, pl.kchunk, pl.split, pl.ps_nt, s);
    if (!pl.vec) ICK_BY_LAYOUT(launch_one, 2, 2, ICK_COMMA_FALSE, pl, s);
    if (pl.xl) {
        // four waves of 64 x 64 and one LDS buffer: two workgroups per CU overlap their phases (against eight waves, two
        // buffers, one workgroup per CU: cross K/V 140 -> 128 us, train step 1.867 -> 1.839 ms)
        ICK_BY_LAYOUT(launch_xl4, 4, 4, ICK_COMMA_TRUE, pl, s);
    }
    if (pl.wide) ICK_BY_LAYOUT(launch_wide, 2, 2, ICK_COMMA_TRUE, pl, s);
    if (pl.big) ICK_BY_LAYOUT(launch_one, 2, 2, ICK_COMMA_TRUE, pl, s);
    ICK_BY_LAYOUT(launch_one, 1, 1, ICK_COMMA_TRUE, pl, s);
}

int launch_plans_grouped(const Plan* const* pls, int n, hipStream_t s) {
    const Plan& pl = *pls[0];
    if (pl.big) ICK_BY_LAYOUT(launch_group, 2, 2, , pls, n, s);
    ICK_BY_LAYOUT(launch_group, 1, 1, , pls, n, s);
}

}  // namespace
}  // namespace ick

extern "C" int ick_gemm(const ick_gemm_args* in, void* stream) {
    using namespace ick;
    Plan pl;
    if (int rc = make_plan(in, pl)) return rc;
    return launch_plan(pl, (hipStream_t)stream);
}

extern "C" int ick_gemm_plan(const ick_gemm_args* in, ick_gemm_plan_info* out) {
    using namespace ick;
    if (!out) return ICK_EINVAL;
    Plan pl;
    if (int rc = make_plan(in, pl)) return rc;
    const int bmn = pl.big ? 64 : 32;
    out->tile_m = pl.wide ? 128 : bmn; out->tile_n = pl.xl ? 128 : bmn; out->waves = pl.wide ? 8 : 4;   // (xl: 4 or 8, ICK_GEMM_XL4)
    out->presplit = pl.ps;
    if (pl.ps) {
        int wpc;
        gemm_ps_tile_dims(pl.ps_tile, &out->tile_m, &out->tile_n, &wpc);
        out->waves = (pl.ps_tile == 7 || pl.ps_tile == 8) ? 4 : 8;      // the four-wave tiles of csrc/gemm_ps.hip
    }
    out->tiles_m = pl.tiles_m; out->tiles_n = pl.tiles_n; out->split_k = pl.split;
    out->a_kmajor = pl.akm; out->b_kmajor = pl.bkm; out->vec = pl.vec;
    out->split_bf16 = pl.spl;
    return ICK_OK;
}

extern "C" int ick_set_gemm_split(int mode) {
    if (mode < 0 || mode > 2) return ICK_EINVAL;
    ick::g_gemm_split = mode;
    return ICK_OK;
}

extern "C" int ick_get_gemm_split(void) { return ick::gemm_split_mode(); }

extern "C" int ick_gemm_grouped(const ick_gemm_args* problems, int32_t count, void* stream) {
    using namespace ick;
    if (!problems || count <= 0 || count > 64) return ICK_EINVAL;
    Plan plans[64];
    for (int i = 0; i < count; ++i)
        if (int rc = make_plan(problems + i, plans[i])) return rc;
    // weight-gradient problems that are small alone (32x32 tiles) but fill the GPU together take 64x64 tiles: four
    // independent accumulators per wave instead of one dependent chain (train step 2.49 -> 2.44 ms) -- from 512 such
    // tiles on (64 x 64 tiles for EVERY grouped weight gradient: train step 1.742 -> 1.79 ms, round 4)
    {
        int64_t total = 0;
        for (int i = 0; i < count; ++i)
            if (plans[i].vec && !plans[i].big && plans[i].akm && plans[i].bkm && !(plans[i].a.flags & ICK_GEMM_COLSUM_ONLY))
                total += (int64_t)ceil_div(plans[i].a.M, 64) * ceil_div(plans[i].a.N, 64) * plans[i].split;
        if (total >= 512)
            for (int i = 0; i < count; ++i)
                if (plans[i].vec && !plans[i].big && plans[i].akm && plans[i].bkm && !(plans[i].a.flags & ICK_GEMM_COLSUM_ONLY))
                    if (int rc = make_plan(problems + i, plans[i], 1)) return rc;
    }
    hipStream_t s = (hipStream_t)stream;
    bool done[64] = {false};
    for (int i = 0; i < count; ++i) {
        if (done[i]) continue;
        if (!plans[i].vec || plans[i].wide || plans[i].ps) {   // element-wise staging / 8-wave tiles: no grouped instantiation
            done[i] = true;
            if (int rc = launch_plan(plans[i], s)) return rc;
            continue;
        }
        const Plan* grp[kGroupMax];
        int n = 0;
        for (int j = i; j < count && n < kGroupMax; ++j) {
            if (done[j] || !plans[j].vec || plans[j].wide || plans[j].ps || plans[j].akm != plans[i].akm || plans[j].bkm != plans[i].bkm ||
                plans[j].big != plans[i].big || plans[j].spl != plans[i].spl) continue;
            grp[n++] = &plans[j];
            done[j] = true;
        }
        const int rc = n == 1 ? launch_plan(*grp[0], s) : launch_plans_grouped(grp, n, s);
        if (rc) return rc;
    }
    return ICK_OK;
}
