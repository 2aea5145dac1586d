// fp32 GEMM with split-bf16 products whose B operand (an nn.Linear / 1x1 conv weight) arrives PRE-SPLIT.
//
//   C[m, n] = epilogue( sum_k A(m,k) * B(n,k) ),   A fp32 activations,  B = hi + mid + lo bf16 planes (ick_presplit_weights)
//
// Call sites (include/ick_amd.h, ick_gemm with b_ps set): Encoder.conv1 (geo-aware/models.py:32,45), the cross K/V
// projection of the image rows and fc_vocab (geo-aware/models.py:241-242,303), and fc_vocab's data gradient.
//
// Why a kernel of its own (csrc/gemm.hip splits BOTH operands between the global load and the LDS store): there the
// split costs 5.5 VALU instructions per staged element and the vector issue port, not the matrix pipe, bounds the
// kernel.  A weight is the same for the whole step, so it is split once per step by the packing launch; what is left to
// split is the activation operand, and that is done on the 16 x 32 MFMA fragment a wave is about to use (8 elements per
// lane and slice).  Nothing is staged through registers any more:
//   * every tile slice travels global -> LDS by LDS-DMA (buffer_load_dwordx4 ... lds, 1 KiB per wave instruction): the
//     pre-split B planes in the MFMA operand's image (64-byte rows, XOR-swizzled chunks -- the swizzle is applied to
//     the SOURCE address, the LDS side of a DMA is lane-linear), the A tile as raw fp32;
//   * both operands run through rings of D + 1 LDS buffers, D slices in flight ahead of the one being read; the only
//     waits are counted s_waitcnt vmcnt(N) + one raw s_barrier per slice (with D = 2 every wave issues the same number
//     of pieces per slice -- a zero fill into a scratch KiB where a wave has one piece less -- so N is an immediate);
//   * a wave owns TM x 16 rows x (TN x 16) columns: it requests its B fragments (PF blocks ahead, or the whole slice),
//     reads its raw A fragment (k-major: eight ds_read_b32 from a line image rotated by 16 floats per 8 k lines;
//     k-contiguous: two ds_read_b128 from 128-byte rows with XOR-swizzled chunks -- both conflict free: PMC
//     SQ_LDS_BANK_CONFLICT = 0), splits it (44 VALU) and issues 6 x TM x TN v_mfma_f32_16x16x32_bf16;
//   * tile shapes: see launch_gemm_ps below.  Measured (tools/gemm_ps_bench.py, profiles/r04_*_gemm_ps_tiles.txt, PMC
//     profiles/r04_c_pmc_gemm_ps.txt): 128 x 128 with two workgroups per CU is the fastest wherever the output is wider
//     than 320 columns (cross K/V 128 -> 100 us, vocabulary 76 -> 65 us against the stager-split kernel), 128 x 160 for
//     Encoder.conv1 (121 us: level with the stager-split 128 x 64 tile).  Ring depth, fragment prefetch depth, tile
//     size and non-temporal A loads all leave the rate at 120-135 TFLOP/s fp32-equivalent with the matrix pipe ~50 %
//     busy on the CUs in use -- about 60 % of what bf16 MFMA loops hold on random data at the clock the chip keeps under
//     that load (MI355X_MICROARCH.md, DVFS give-back: 1.25 PFLOP/s at 1.9 GHz, i.e. ~208 TFLOP/s fp32-equivalent).
#include <algorithm>

#include "gemm_common.h"

// Diagnostic build (-DICK_PS_STAMPS, tools/debug/gemm_ps_stamps.py): wave 0 of one workgroup in the middle of the grid sums
// the shader-clock ticks it spends in each phase of a slice.  Compiled out of the product library.
#ifdef ICK_PS_STAMPS
__device__ unsigned long long ick_ps_stamps[10];
#define ICK_PSTAMP(var)                                   \
    do {                                                  \
        __builtin_amdgcn_sched_barrier(0);                \
        var = __builtin_amdgcn_s_memtime();               \
        __builtin_amdgcn_sched_barrier(0);                \
    } while (0)
extern "C" int ick_debug_read_ps_stamps(unsigned long long* out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(ick_ps_stamps), sizeof(ick_ps_stamps));
}
#else
#define ICK_PSTAMP(var)
#endif

namespace ick {
namespace {

typedef __attribute__((address_space(3))) void* lds_void_ptr;

inline int64_t ps_rows(int N) { return (int64_t)ceil_div(N, 64) * 64; }
inline int64_t ps_bytes(int N, int K) { return (int64_t)ceil_div(K, 32) * 3 * ps_rows(N) * 64; }

// ---------------------------------------------------------------------------------------------------------------------
// Pre-split copies: dst[slice s][plane p][row n < Np][32 k] bf16, zero beyond N / K.
constexpr int kPsMaxItems = 16;
struct PsBatch {
    int count;
    int first[kPsMaxItems + 1];     // first workgroup of every item
    ick_presplit_item it[kPsMaxItems];
};

__global__ __launch_bounds__(256) void presplit_kernel(PsBatch b) {
    __shared__ float tile[64][33];
    int gi = 0;
    while (gi + 1 < b.count && (int)blockIdx.x >= b.first[gi + 1]) ++gi;
    const ick_presplit_item it = b.it[gi];
    const int local = blockIdx.x - b.first[gi];
    const int blocks_n = (it.N + 63) / 64;
    const int nb = local % blocks_n, s = local / blocks_n;
    const int t = threadIdx.x;
    const float* src = it.src;
    // 32 contiguous source bytes per thread: two 16-byte loads where the view allows (the activation planes of the backward
    // pass -- memory^T, h^T -- are made inside the step: a dword per lane moves a quarter of what a dwordx4 does)
    const bool al16 = (reinterpret_cast<uintptr_t>(src) & 15) == 0;
    if (it.src_cs == 1) {                 // k contiguous: 8 consecutive k of one row per thread
        const int r = t >> 2, kk = (t & 3) * 8;
        const int n = nb * 64 + r;
        const int k0 = 32 * s + kk;
        if (al16 && (it.src_rs & 3) == 0 && n < it.N && k0 + 7 < it.K) {
            const float4* q = reinterpret_cast<const float4*>(src + (int64_t)n * it.src_rs + k0);
            const float4 a = q[0], b = q[1];
            tile[r][kk + 0] = a.x; tile[r][kk + 1] = a.y; tile[r][kk + 2] = a.z; tile[r][kk + 3] = a.w;
            tile[r][kk + 4] = b.x; tile[r][kk + 5] = b.y; tile[r][kk + 6] = b.z; tile[r][kk + 7] = b.w;
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int k = k0 + j;
                tile[r][kk + j] = (n < it.N && k < it.K) ? src[(int64_t)n * it.src_rs + k] : 0.f;
            }
        }
    } else {                              // rows contiguous (a transposed view): 8 consecutive rows at one k per thread
        const int kk = t >> 3, r0 = (t & 7) * 8;
        const int k = 32 * s + kk;
        const int n0 = nb * 64 + r0;
        if (al16 && it.src_rs == 1 && (it.src_cs & 3) == 0 && n0 + 7 < it.N && k < it.K) {
            const float4* q = reinterpret_cast<const float4*>(src + n0 + (int64_t)k * it.src_cs);
            const float4 a = q[0], b = q[1];
            tile[r0 + 0][kk] = a.x; tile[r0 + 1][kk] = a.y; tile[r0 + 2][kk] = a.z; tile[r0 + 3][kk] = a.w;
            tile[r0 + 4][kk] = b.x; tile[r0 + 5][kk] = b.y; tile[r0 + 6][kk] = b.z; tile[r0 + 7][kk] = b.w;
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int n = n0 + j;
                tile[n - nb * 64][kk] = (n < it.N && k < it.K) ? src[(int64_t)n * it.src_rs + (int64_t)k * it.src_cs] : 0.f;
            }
        }
    }
    __syncthreads();
    const int r = t >> 2, c = t & 3;
    uint32_t h[4], m[4], l[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) split3(tile[r][8 * c + 2 * j], tile[r][8 * c + 2 * j + 1], h[j], m[j], l[j]);
    const int64_t np = (int64_t)blocks_n * 64;
    char* dst = reinterpret_cast<char*>(it.dst) + ((int64_t)s * 3 * np + nb * 64 + r) * 64 + c * 16;
    *reinterpret_cast<uint4*>(dst) = uint4{h[0], h[1], h[2], h[3]};
    *reinterpret_cast<uint4*>(dst + np * 64) = uint4{m[0], m[1], m[2], m[3]};
    *reinterpret_cast<uint4*>(dst + 2 * np * 64) = uint4{l[0], l[1], l[2], l[3]};
}

}  // namespace

// ---------------------------------------------------------------------------------------------------------------------
template <int WM_, int WN_, int TM_, int TN_, bool AKM_, int D_, int PF_>
struct PsCfg {
    static constexpr int WM = WM_, WN = WN_, TM = TM_, TN = TN_, D = D_;   // D: slices in flight ahead of the one being read
    static constexpr int PF = PF_ < TN_ ? PF_ : TN_;   // B fragment blocks requested from LDS ahead of the MFMAs that use them
    static constexpr int ST = D + 1;                                         // ring buffers per operand
    static constexpr bool AKM = AKM_;
    static constexpr int BM = WM * TM * 16, BN = WN * TN * 16, NW = WM * WN, NT = NW * 64;
    static constexpr int A_STAGE = BM * 32 * 4;          // bytes: raw fp32 slice of the A tile
    static constexpr int B_PLANE = BN * 64;              // bytes: one bf16 plane of the B tile slice
    static constexpr int B_STAGE = 3 * B_PLANE;
    static constexpr int NPA = BM / 8;                   // 1 KiB DMA pieces of an A slice
    static constexpr int NPB = 3 * BN / 16;              // ... of a B slice (16 rows of one plane each)
    static constexpr int PA_W = NPA / NW;                // pieces per wave
    static constexpr int PB_W = (NPB + NW - 1) / NW;
    // D >= 2: the waits are counted (s_waitcnt vmcnt(N) with N pieces of the younger slices still in flight), so every
    // wave must issue the SAME number of pieces per slice: the waves that have one B piece less send a zero fill into a
    // 1 KiB scratch area instead
    static constexpr bool UNIFORM = D >= 2;
    static constexpr int SCRATCH = UNIFORM && (NPB % NW != 0) ? 1024 : 0;
    static constexpr int LDS = ST * (A_STAGE + B_STAGE) + SCRATCH;
    static constexpr int IN_FLIGHT = (D - 1) * (PA_W + PB_W);   // pieces of the younger slices at the end of an iteration
    static_assert(NPA % NW == 0 && PA_W >= 1, "every wave issues the same number of A pieces (counted vmcnt)");
    static_assert(D == 1 || D == 2, "ring depth");
    static_assert(LDS <= 160 * 1024, "tile exceeds the LDS of a CU");
};

#define ICK_WAIT_VMCNT(N) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory")
// every wave is done reading the slice (its LDS reads have been consumed by MFMAs; the explicit wait covers reads the
// compiler may have left in flight) and its DMA pieces have landed: one barrier publishes both
#define ICK_LDS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

// One slice of the B tile / of the A tile by LDS-DMA: this wave's 1 KiB pieces (see the kernel below).
template <class C>
__device__ __forceinline__ void ps_dma_b(const __amdgpu_buffer_rsrc_t& rsrc, char* Bs, char* scratch, int buf, int wave,
                                         uint32_t bvoff, int slice, bool valid, int n0, int np_rows) {
#pragma unroll
    for (int j = 0; j < C::PB_W; ++j) {
        const int g = wave + C::NW * j;
        if (g < C::NPB) {               // wave-uniform
            const int plane = g / (C::BN / 16), rblk = g % (C::BN / 16);
            const bool in = valid && (n0 + rblk * 16 < np_rows);
            const uint32_t soff = (uint32_t)((((int64_t)slice * 3 + plane) * np_rows + rblk * 16) * 64);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void_ptr)(Bs + buf * C::B_STAGE + plane * C::B_PLANE + rblk * 1024),
                                                     16, in ? bvoff : kOobOffset, in ? soff : 0u, 0, 0);
        } else if (C::UNIFORM) {        // keeps this wave's piece count equal to the others': a zero fill nobody reads
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void_ptr)scratch, 16, kOobOffset, 0u, 0, 0);
        }
    }
}
// nt: the A operand is read once by the whole launch (one column tile): a non-temporal load keeps it from evicting the
// weight planes, which every workgroup of the XCD re-reads, from that XCD's L2
template <class C>
__device__ __forceinline__ void ps_dma_a(const __amdgpu_buffer_rsrc_t& rsrc, char* As, int buf, int wave,
                                         const uint32_t* avoff, const int* akl, int k0, int kend,
                                         uint32_t soff, bool nt) {
#pragma unroll
    for (int j = 0; j < C::PA_W; ++j) {
        const int g = wave + C::NW * j;
        const bool in = k0 + akl[j] < kend;      // also false for every lane of a slice beyond the K range
        lds_void_ptr dst = (lds_void_ptr)(As + buf * C::A_STAGE + g * 1024);
        if (nt) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, dst, 16, in ? avoff[j] : kOobOffset, soff, 0, 2);
        else __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, dst, 16, in ? avoff[j] : kOobOffset, soff, 0, 0);
    }
}

// Piece `idx` of this wave's share of a slice (B pieces first, then A pieces): the kernel issues them one at a time BETWEEN
// its MFMAs.  (All eight waves issuing their pieces together right behind the barrier stalled every wave for ~700 cycles
// per slice: a CU's address unit takes one 1 KiB wave request per 16 cycles -- tools/debug/gemm_ps_stamps.py.)
template <class C>
__device__ __forceinline__ void ps_dma_piece(int idx, const __amdgpu_buffer_rsrc_t& rsrc_a, const __amdgpu_buffer_rsrc_t& rsrc_b,
                                             char* As, char* Bs, char* scratch, int buf, int wave, uint32_t bvoff,
                                             const uint32_t* avoff, const int* akl, int slice, bool valid,
                                             int n0, int np_rows, int k0, int kend, uint32_t soff_a) {
    if (idx < C::PB_W) {
        const int g = wave + C::NW * idx;
        if (g < C::NPB) {               // wave-uniform
            const int plane = g / (C::BN / 16), rblk = g % (C::BN / 16);
            const bool in = valid && (n0 + rblk * 16 < np_rows);
            const uint32_t soff = (uint32_t)((((int64_t)slice * 3 + plane) * np_rows + rblk * 16) * 64);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_b, (lds_void_ptr)(Bs + buf * C::B_STAGE + plane * C::B_PLANE + rblk * 1024),
                                                     16, in ? bvoff : kOobOffset, in ? soff : 0u, 0, 0);
        } else if (C::UNIFORM) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_b, (lds_void_ptr)scratch, 16, kOobOffset, 0u, 0, 0);
        }
    } else {
        const int j = idx - C::PB_W;
        const int g = wave + C::NW * j;
        const bool in = k0 + akl[j] < kend;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_a, (lds_void_ptr)(As + buf * C::A_STAGE + g * 1024), 16,
                                                 in ? avoff[j] : kOobOffset, soff_a, 0, 0);
    }
}

// waves per SIMD the register allocator may plan for: 2 when the tile's LDS footprint leaves room for one workgroup per CU
// (8 waves on 4 SIMDs: 256 VGPRs each, enough to hold a whole slice's B fragments), 4 for the two-workgroup tile
template <int WM_, int WN_, int TM_, int TN_, bool AKM_, int D_, int PF_>
__global__ __launch_bounds__(WM_ * WN_ * 64)
__attribute__((amdgpu_waves_per_eu(1, (PsCfg<WM_, WN_, TM_, TN_, AKM_, D_, PF_>::LDS > 80 * 1024 || WM_ * WN_ <= 4 ? 2 : 4)))) void gemm_ps_kernel(ick_gemm_args p, int np_rows, int64_t bps_bytes, int tiles_m,
                                                                 int tiles_n, int kchunk, int a_nt) {
    using C = PsCfg<WM_, WN_, TM_, TN_, AKM_, D_, PF_>;
    constexpr int BM = C::BM, BN = C::BN, TM = C::TM, TN = C::TN, NW = C::NW, D = C::D, ST = C::ST;
    constexpr bool AKM = C::AKM;
    extern __shared__ __attribute__((aligned(1024))) char smem_ps[];
#ifdef ICK_PS_STAMPS
    unsigned long long t_start = 0, t_loop = 0, t_end = 0;
    ICK_PSTAMP(t_start);
    const unsigned long long rt_start = __builtin_amdgcn_s_memrealtime();      // 100 MHz: the shader clock this workgroup saw
#endif
    char* const As = smem_ps;
    char* const Bs = smem_ps + ST * C::A_STAGE;
    char* const scratch = Bs + ST * C::B_STAGE;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    constexpr int WN = C::WN;
    const int wm = wave / WN, wn = wave % WN;
    const int fi = lane & 15, fq = lane >> 4;

    // tile of this workgroup: XCD-aware order (workgroups are dealt to the 8 XCDs round robin in linear order: b, b + 8, ...
    // share an XCD).  Without a K split each XCD walks a contiguous run of tiles.  With one, the (K slice, tile) pairs are
    // numbered slice-major and each XCD takes a contiguous run of THAT order: an XCD then streams only its own K range of
    // both operands (one to two slices) and the re-reads by the tiles of the grid hit its L2 -- with the tile-major order
    // every XCD read the whole K range of both operands (PMC: the vocabulary's data gradient moved 262 MB for ~70 MB of
    // operands, a cross K/V weight gradient ~400 MB for 59 MB).
    int bid = blockIdx.x;
    int zid = blockIdx.z;
    {
        const int nwg = tiles_m * tiles_n;
        if (gridDim.z > 1) {
            const int total = nwg * (int)gridDim.z, lin = bid + nwg * zid;
            const int xcd = lin & 7, qq = total >> 3, rr = total & 7;
            const int v = xcd * qq + min(xcd, rr) + (lin >> 3);
            zid = v / nwg;
            bid = v - zid * nwg;
        } else {
            const int xcd = bid & 7, qq = nwg >> 3, rr = nwg & 7;
            bid = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + (bid >> 3);
        }
    }
    int tm, tn;
    if (tiles_m <= tiles_n) { tm = bid % tiles_m; tn = bid / tiles_m; }
    else { tn = bid % tiles_n; tm = bid / tiles_n; }
    const int m0 = tm * BM, n0 = tn * BN;
    const int kbeg = zid * kchunk;
    const int kend = min(p.K, kbeg + kchunk);
    const int nk = (kend - kbeg + 31) >> 5;
    const int s0 = kbeg >> 5;

    // ---- A: per-lane source offsets of this wave's pieces (fixed for the whole K loop; the K position is the scalar offset)
    const __amdgpu_buffer_rsrc_t rsrc_a =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.A), (short)0, (int)(p.a_extent * 4), 0x00020000);
    const RowMap amap{p.a_grp, p.a_gs, p.a_gmap, p.a_rs};
    uint32_t avoff[C::PA_W];
    int akl[C::PA_W];                       // k (relative to the slice) of the lane's 16 bytes
#pragma unroll
    for (int j = 0; j < C::PA_W; ++j) {
        const int g = wave + NW * j;
        int row, extra;
        if constexpr (AKM) {
            constexpr int LPK = BM / 4, KPP = 64 / LPK;        // lanes per k line, k lines per piece
            const int kl = g * KPP + lane / LPK, cpos = lane % LPK;
            const int c = (cpos - 4 * ((kl >> 3) & 3)) & (LPK - 1);   // the line image is rotated by 16 floats per 8 k lines
            row = m0 + 4 * c;
            akl[j] = kl;
            extra = 0;
            const int one[1] = {min(row, p.M - 1)};
            int64_t mo[1];
            map_rows<1>(amap, one, mo);
            avoff[j] = row < p.M ? (uint32_t)(mo[0] * 4) + (uint32_t)((int64_t)kl * p.a_ks * 4) : kOobOffset;
        } else {
            const int rt = g * 8 + (lane >> 3), cpos = lane & 7;
            const int c = cpos ^ ((rt >> 1) & 7);
            row = m0 + rt;
            akl[j] = 4 * c;
            extra = 4 * c;
            const int one[1] = {min(row, p.M - 1)};
            int64_t mo[1];
            map_rows<1>(amap, one, mo);
            avoff[j] = row < p.M ? (uint32_t)((mo[0] + extra) * 4) : kOobOffset;
        }
    }
    // ---- B: one per-lane offset; plane / 16-row block / slice are scalar
    const __amdgpu_buffer_rsrc_t rsrc_b =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.b_ps), (short)0, (int)bps_bytes, 0x00020000);
    const uint32_t bvoff = (uint32_t)((n0 + (lane >> 2)) * 64 + (((lane & 3) ^ ((-(lane >> 4)) & 3)) * 16));

    auto dma_b = [&](int it, int buf) { ps_dma_b<C>(rsrc_b, Bs, scratch, buf, wave, bvoff, s0 + it, it < nk, n0, np_rows); };
    auto dma_a = [&](int it, int buf) {
        const int k0 = kbeg + 32 * it;
        const uint32_t soff = it < nk ? (uint32_t)(AKM ? (int64_t)k0 * p.a_ks * 4 : (int64_t)k0 * 4) : 0u;
        ps_dma_a<C>(rsrc_a, As, buf, wave, avoff, akl, k0, kend, soff, a_nt != 0);
    };

    f32x4 acc[TM][TN];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

    // prologue: slices 0 .. D-1 (B then A of each); the first must have landed
#pragma unroll
    for (int i = 0; i < D; ++i) { dma_b(i, i); dma_a(i, i); }
    ICK_WAIT_VMCNT(C::IN_FLIGHT);
    ICK_LDS_BARRIER();

#ifdef ICK_PS_STAMPS
    unsigned long long ph[6] = {0, 0, 0, 0, 0, 0}, ta = 0, tb = 0, tc = 0, td = 0, te = 0, tf = 0;
    ICK_PSTAMP(t_loop);
#endif
    int buf = 0;                             // ring buffer holding slice `it`
    for (int it = 0; it < nk; ++it) {
        ICK_PSTAMP(ta);
        // slice it + D goes into the buffer that was read in iteration it - 1 (slices beyond the range are zero fills that
        // touch no memory: the piece counts stay constant, so the waits below are immediates); its pieces are issued one
        // by one between the MFMAs below
        int nb = buf + D;
        if (nb >= ST) nb -= ST;
        const int k0_nx = kbeg + 32 * (it + D);
        const uint32_t soff_nx = it + D < nk ? (uint32_t)(AKM ? (int64_t)k0_nx * p.a_ks * 4 : (int64_t)k0_nx * 4) : 0u;
        auto piece = [&](int idx) {
            ps_dma_piece<C>(idx, rsrc_a, rsrc_b, As, Bs, scratch, nb, wave, bvoff, avoff, akl, s0 + it + D, it + D < nk, n0,
                            np_rows, k0_nx, kend, soff_nx);
        };
        const int abuf = buf, bbuf = buf;
        ICK_PSTAMP(tb);
        // B fragments PF blocks ahead of their MFMAs (PF = TN: the whole slice is requested first, so the LDS latency is
        // paid once per slice and the 6 x TM x TN MFMAs then issue back to back; the ds_reads return in order)
        const char* bt = Bs + bbuf * C::B_STAGE;
        auto read_b = [&](int b, bf16x8_t (&f)[3]) {
            const int row = (wn * TN + b) * 16 + fi;
            const char* ptr = bt + row * 64 + 16 * (fq ^ kc_swz(row));
#pragma unroll
            for (int pl = 0; pl < 3; ++pl)
                f[pl] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const u32x4_t*>(ptr + pl * C::B_PLANE));
        };
        // The TN column blocks are processed in groups of GB = PF blocks: the next group's fragments are requested while
        // this group's MFMAs issue, and inside a group the six partial products are the OUTER loop -- consecutive MFMAs
        // then go to different accumulators.  (Six MFMAs in a row into one accumulator ran at half rate: every tile shape
        // settled at 32 cycles per v_mfma_f32_16x16x32_bf16, profiles/r04_*_gemm_ps_tiles.txt.)  The order of the six
        // products per accumulator is unchanged (smallest first), so the sums are bit-identical.
        constexpr int GB = C::PF, NG = (TN + GB - 1) / GB;
        bf16x8_t bf[2][GB][3];
#pragma unroll
        for (int j = 0; j < GB; ++j) read_b(j, bf[0][j]);
        // the wave's A fragments (TM x 16 rows x 32 k): raw fp32 -> three bf16 planes each (the split's ~44 VALU
        // instructions per fragment run while the B fragments are on their way)
        bf16x8_t af[TM][3];
        const char* at = As + abuf * C::A_STAGE;
#pragma unroll
        for (int a = 0; a < TM; ++a) {
            float x[8];
            const int r0 = (wm * TM + a) * 16;
            if constexpr (AKM) {
                const int pos = (r0 + fi + 16 * fq) & (BM - 1);
#pragma unroll
                for (int j = 0; j < 8; ++j) x[j] = *reinterpret_cast<const float*>(at + ((8 * fq + j) * BM + pos) * 4);
            } else {
                const int row = r0 + fi, sw = (row >> 1) & 7;
                const float4 u = *reinterpret_cast<const float4*>(at + row * 128 + 16 * ((2 * fq) ^ sw));
                const float4 v = *reinterpret_cast<const float4*>(at + row * 128 + 16 * ((2 * fq + 1) ^ sw));
                x[0] = u.x; x[1] = u.y; x[2] = u.z; x[3] = u.w; x[4] = v.x; x[5] = v.y; x[6] = v.z; x[7] = v.w;
            }
            uint32_t h[4], m[4], l[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) split3(x[2 * j], x[2 * j + 1], h[j], m[j], l[j]);
            af[a][0] = __builtin_bit_cast(bf16x8_t, u32x4_t{h[0], h[1], h[2], h[3]});
            af[a][1] = __builtin_bit_cast(bf16x8_t, u32x4_t{m[0], m[1], m[2], m[3]});
            af[a][2] = __builtin_bit_cast(bf16x8_t, u32x4_t{l[0], l[1], l[2], l[3]});
        }
        // the scheduler must not sink the fragment requests back to their uses (it would, to save registers)
        __builtin_amdgcn_sched_barrier(0);
        ICK_PSTAMP(tc);
        constexpr int pa[6] = {0, 2, 1, 0, 1, 0}, pb[6] = {2, 0, 1, 1, 0, 0};     // (A plane, B plane), smallest products first
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            if (g + 1 < NG) {
#pragma unroll
                for (int j = 0; j < GB; ++j)
                    if ((g + 1) * GB + j < TN) read_b((g + 1) * GB + j, bf[(g + 1) & 1][j]);
            }
#pragma unroll
            for (int q = 0; q < 6; ++q) {
#pragma unroll
                for (int j = 0; j < GB; ++j)
#pragma unroll
                    for (int a = 0; a < TM; ++a) {
                        const int b = g * GB + j;
                        if (b < TN)
                            acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[a][pa[q]], bf[g & 1][j][pb[q]], acc[a][b], 0, 0, 0);
                    }
                // one DMA piece of slice it + D per slot; with D = 1 the pieces go out in the first half of the slots (they
                // must have landed by the end of this iteration), with D >= 2 over all of them
                constexpr int NS = NG * 6, NP = C::PB_W + C::PA_W;
                constexpr int SPAN = D >= 2 ? NS : (NS + 1) / 2;
                constexpr int STRIDE = SPAN / NP > 0 ? SPAN / NP : 1;
                const int slot = g * 6 + q;
                if (slot % STRIDE == 0 && slot / STRIDE < NP) piece(slot / STRIDE);
                if (slot == NS - 1) {            // whatever did not fit the slots
#pragma unroll
                    for (int r = (NS - 1) / STRIDE + 1; r < NP; ++r) piece(r);
                }
            }
        }
        ICK_PSTAMP(td);
        // slice it + 1 has landed (in-order completion: only the pieces of the D - 1 younger slices may remain)
        ICK_WAIT_VMCNT(C::IN_FLIGHT);
        ICK_PSTAMP(te);
        ICK_LDS_BARRIER();
        ICK_PSTAMP(tf);
#ifdef ICK_PS_STAMPS
        ph[0] += tb - ta; ph[1] += tc - tb; ph[2] += td - tc; ph[3] += te - td; ph[4] += tf - te; ph[5] += 1;
#endif
        if (++buf == ST) buf = 0;
    }
#ifdef ICK_PS_STAMPS
    ICK_PSTAMP(t_end);
    const unsigned long long t_loop_end = t_end;
#endif
    ICK_WAIT_VMCNT(0);       // the zero fills of the last iterations write LDS too: none may outlive the workgroup
    gemm_epilogue<TM, TN>(p, acc, m0, n0, wm, wn, fi, fq, zid);
#ifdef ICK_PS_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the stores of the epilogue have been acknowledged
    ICK_PSTAMP(t_end);
    if (threadIdx.x == 0 && blockIdx.x == (gridDim.x >> 1) && blockIdx.z == 0) {
        for (int i = 0; i < 6; ++i) ick_ps_stamps[i] = ph[i];
        ick_ps_stamps[6] = t_loop - t_start;          // set-up + first slices requested and landed
        ick_ps_stamps[7] = t_end - t_loop_end;        // epilogue incl. its stores
        ick_ps_stamps[8] = t_end - t_start;
        ick_ps_stamps[9] = __builtin_amdgcn_s_memrealtime() - rt_start;
    }
#endif
}


namespace {

template <class C>
int launch_ps_cfg(const ick_gemm_args& a, int np_rows, int64_t bytes, int tiles_m, int tiles_n, int kchunk, int split,
                  int a_nt, hipStream_t s) {
    static LdsAttrOnce attr;
    if (int e = attr.ensure(reinterpret_cast<const void*>(gemm_ps_kernel<C::WM, C::WN, C::TM, C::TN, C::AKM, C::D, C::PF>), 160 * 1024))
        return e;
    hipLaunchKernelGGL((gemm_ps_kernel<C::WM, C::WN, C::TM, C::TN, C::AKM, C::D, C::PF>), dim3(tiles_m * tiles_n, 1, split),
                       dim3(C::NT), C::LDS, s, a, np_rows, bytes, tiles_m, tiles_n, kchunk, a_nt);
    ICK_LAUNCH_RET();
}

template <int WM, int WN, int TM, int TN, int D, int PF>
int launch_ps_tile(const ick_gemm_args& a, bool akm, int np, int64_t bytes, int tiles_m, int tiles_n, int kchunk, int split,
                   int a_nt, hipStream_t s) {
    if (akm) return launch_ps_cfg<PsCfg<WM, WN, TM, TN, true, D, PF>>(a, np, bytes, tiles_m, tiles_n, kchunk, split, a_nt, s);
    return launch_ps_cfg<PsCfg<WM, WN, TM, TN, false, D, PF>>(a, np, bytes, tiles_m, tiles_n, kchunk, split, a_nt, s);
}

}  // namespace

// Tile shapes of the pre-split kernel (ick_gemm's plan picks 1 or 2; the others stay selectable with ICK_PS_TILE so that the
// sweep tables under profiles/r04_*_gemm_ps_* can be reproduced): bm x bn, waves as WM x WN, wave tile TM x TN blocks, D
// slices in flight, PF = B fragment blocks requested ahead of their MFMAs.
//   0: 64 x 320   4 x 2 waves of 16 x 160, D 1, 136 KB of LDS
//   1: 128 x 128  8 x 1 waves of 16 x 128, D 1,  80 KB: two workgroups per CU            <- outputs wider than 320 columns, split K
//   2: 128 x 160  8 x 1 waves of 16 x 160, D 2, 139 KB                                   <- Encoder.conv1 (N = 300, one column pair)
//   3: 128 x 320  4 x 2 waves of 32 x 160, D 1, 152 KB
//   4: 128 x 128  as 1 with D 2, 120 KB: one workgroup per CU
//   5: 128 x 128  4 x 2 waves of 32 x 64 (32-row wave tiles read 1.5 x less LDS per product), D 1, two workgroups per CU
//   6: 128 x 160  4 x 2 waves of 32 x 80, D 2
// What was measured on them, and on variants that are no longer compiled (ring depth 3, 64 x 64 / 64 x 128 tiles at four
// workgroups per CU, a ping-pong schedule of two wave groups half a slice apart): DESIGN.md section 3.1c.
//   7: 128 x 128  FOUR waves (4 x 1) of 32 x 128, D 1, 80 KB: two workgroups per CU, one wave of each per SIMD -- the two
//                 workgroups are not coupled by a barrier, so one's fragment phase can run under the other's MFMAs
//   8: 64 x 160   four waves (2 x 2) of 32 x 80, D 1, 76 KB: two per CU
//   9: 128 x 80   8 x 1 waves of 16 x 80, D 1, 62 KB: two workgroups per CU (Encoder.conv1 as 98 x 4 tiles: the two
//                 workgroups of a CU are not barrier-coupled, so their fragment and MFMA phases overlap)
//  10: 128 x 160  as 2 with D 1, 92 KB: leaves room on the CU for a 49 KB row-chain workgroup of the other stream
constexpr int kPsTiles = 11;
void gemm_ps_tile_dims(int tile, int* bm, int* bn, int* wgs_per_cu) {
    static const int dims[kPsTiles][3] = {{64, 320, 1}, {128, 128, 2}, {128, 160, 1}, {128, 320, 1}, {128, 128, 1},
                                          {128, 128, 2}, {128, 160, 1}, {128, 128, 2}, {64, 160, 2}, {128, 80, 2}, {128, 160, 1}};
    *bm = dims[tile][0]; *bn = dims[tile][1]; *wgs_per_cu = dims[tile][2];
}
int gemm_ps_tile_count() { return kPsTiles; }

int launch_gemm_ps(const ick_gemm_args& a, bool akm, int tile, int tiles_m, int tiles_n, int kchunk, int split, int a_nt,
                   hipStream_t s) {
    const int np = (int)ps_rows(a.N);
    const int64_t bytes = ps_bytes(a.N, a.K);
    if (bytes >= ((int64_t)1 << 31)) return ICK_EINVAL;
    switch (tile) {
        case 0: return launch_ps_tile<4, 2, 1, 10, 1, 10>(a, akm, np, bytes, tiles_m, tiles_n, kchunk, split, a_nt, s);
        case 1: return launch_ps_tile<8, 1, 1, 8, 1, 2>(a, akm, np, bytes, tiles_m, tiles_n, kchunk, split, a_nt, s);
        case 2: return launch_ps_tile<8, 1, 1, 10, 2, 10>(a, akm, np, bytes, tiles_m, tiles_n, kchunk, split, a_nt, s);
        case 3: return launch_ps_tile<4, 2, 2, 10, 1, 3>(a, akm, np, bytes, tiles_m, tiles_n, kchunk, split, a_nt, s);
        case 4: return launch_ps_tile<8, 1, 1, 8, 2, 8>(a, akm, np, bytes, tiles_m, tiles_n, kchunk, split, a_nt, s);
        case 5: return launch_ps_tile<4, 2, 2, 4, 1, 2>(a, akm, np, bytes, tiles_m, tiles_n, kchunk, split, a_nt, s);
        case 6: return launch_ps_tile<4, 2, 2, 5, 2, 5>(a, akm, np, bytes, tiles_m, tiles_n, kchunk, split, a_nt, s);
        case 7: return launch_ps_tile<4, 1, 2, 8, 1, 2>(a, akm, np, bytes, tiles_m, tiles_n, kchunk, split, a_nt, s);
        case 8: return launch_ps_tile<2, 2, 2, 5, 1, 5>(a, akm, np, bytes, tiles_m, tiles_n, kchunk, split, a_nt, s);
        case 9: return launch_ps_tile<8, 1, 1, 5, 1, 2>(a, akm, np, bytes, tiles_m, tiles_n, kchunk, split, a_nt, s);
        case 10: return launch_ps_tile<8, 1, 1, 10, 1, 10>(a, akm, np, bytes, tiles_m, tiles_n, kchunk, split, a_nt, s);
    }
    return ICK_EINVAL;
}

}  // namespace ick

extern "C" int ick_presplit_bytes(int32_t N, int32_t K, int64_t* bytes) {
    if (N <= 0 || K <= 0 || !bytes) return ICK_EINVAL;
    *bytes = ick::ps_bytes(N, K);
    return ICK_OK;
}

extern "C" int ick_presplit_weights(const ick_presplit_item* items, int32_t count, void* stream) {
    using namespace ick;
    if (!items || count <= 0) return ICK_EINVAL;
    for (int i0 = 0; i0 < count; i0 += kPsMaxItems) {
        PsBatch b;
        b.count = std::min(kPsMaxItems, count - i0);
        int total = 0;
        for (int i = 0; i < b.count; ++i) {
            const ick_presplit_item& it = items[i0 + i];
            ICK_CHECK_ARG(it.src && it.dst && it.N > 0 && it.K > 0);
            ICK_CHECK_ARG(it.src_cs == 1 || it.src_rs == 1);
            ICK_CHECK_ARG((reinterpret_cast<uintptr_t>(it.dst) & 15) == 0);
            b.it[i] = it;
            b.first[i] = total;
            total += ceil_div(it.N, 64) * ceil_div(it.K, 32);
        }
        b.first[b.count] = total;
        hipLaunchKernelGGL(presplit_kernel, dim3(total), dim3(256), 0, (hipStream_t)stream, b);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return (int)e;
    }
    return ICK_OK;
}
