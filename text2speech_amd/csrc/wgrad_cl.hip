// Weight-gradient GEMM straight from channel-last planes (gfx950 / MI355X).
//
//   P[slab][m][n] = sum over (batch b, time t) in the slab of  A[b][m][t] * B[b][n][t]
//
// i.e. the contraction index of the MFMA is TIME, while every activation / gradient tensor of the WaveGlow path is stored
// channel-last (plane[b][c/32][row][c%32], row = halo + t, split-bf16 hi / lo).  Round 1 fed this GEMM from "time-major"
// copies made by plane_transpose_kernel: 7 transposes per WN layer (661 launches, 11-19 ms of a 96 ms training step), three of
// them only to materialise the dilated taps as shifted copies.  Here the operands are DMA'd as they are - for a 32-step
// K-block, the 32 rows x 64 B of one 32-channel chunk are ONE contiguous 2 KB piece, and a dilated tap is a row offset on the
// source address - and the transpose happens in the LDS read: gfx950's ds_read_b64_tr_b16 hands each 16-lane group a 4-row x
// 16-column block of 16-bit elements column-major, which is exactly the k-contiguous fragment v_mfma_f32_16x16x32_bf16 wants
// (cdna_hip_programming.md T10).  Two such reads (rows 8g..8g+3 and 8g+4..8g+7 for lane group g) make one bf16x8 operand.
//
// Operands are described per 32-channel CHUNK (WgradChunk: hi / lo pointers to row 0 of that chunk for batch 0, with any tap
// shift folded in, and the element stride between batch entries), so one GEMM can take its M rows from two plane sets
// ([d_x ; d_skip]) and its N rows from five ([x tap -1 | x tap 0 | x tap +1 | spect | ones]) without any gather pass.
// Tables are padded to 8 chunks per 256-row tile with a zero chunk; the bias column is a chunk whose channel 0 is 1.
//
// Tile: 256 (M) x 256 (N) per 512-thread workgroup, 8 waves as 2 (M) x 4 (N), 128 x 64 per wave, split-bf16 products
// a_lo*b_hi + a_hi*b_lo + a_hi*b_hi, f32 accumulate; K-step = 32 time steps; two LDS stages of [A_hi | A_lo | B_hi | B_lo] x 16 KB
// filled by global_load_lds_dwordx4.  LDS image of a plane tile: [chunk (8)][t (32)][64 B]; the two 32-byte halves of a row
// are swapped on rows with bit 3 set (on the DMA source address and on the read address), which makes the transposed reads of
// a 32-lane half (two 4-row blocks 8 rows apart) hit disjoint banks.
// Split-K over (batch, time block) flattened; every slab writes its own f32 partial (reduced by wn_backward_kernel).
#include "t2s_common.h"
#include "t2s_kernels.h"

#include <stdlib.h>

#include <type_traits>

namespace {

typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;

constexpr int WG_PLANE = 16384;          // one plane tile: 8 chunks x 32 rows x 64 B
constexpr int WG_STAGE = 4 * WG_PLANE;   // A_hi, A_lo, B_hi, B_lo

__device__ __forceinline__ void wg_glds16(const void* gsrc, void* lds_dst) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_dst, 16, 0, 0);
}

// one k-contiguous MFMA operand (16 rows x 32 time steps) out of a channel-last LDS image: two transposed 8-byte reads
__device__ __forceinline__ bf16x8 tr_frag(const char* lds_addr) {
    typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
    const s16x4 lo4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(lds_addr));
    const s16x4 hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(lds_addr + 4 * 64));      // rows + 4
    const s16x8 v = {lo4[0], lo4[1], lo4[2], lo4[3], hi4[0], hi4[1], hi4[2], hi4[3]};
    return __builtin_bit_cast(bf16x8, v);
}

// The same operand through inline assembly (the ping-pong kernel).  hipcc (ROCm 7.2) puts an s_waitcnt vmcnt(0) in front of
// every __builtin_amdgcn_ds_read_tr16_b64 while an LDS-DMA is outstanding (it cannot tell that the read does not alias the
// DMA's destination), which drains the fill the schedule keeps in flight; an asm statement is invisible to that pass.  The
// caller waits (s_waitcnt lgkmcnt(0)) before the first use.  `addr` = LDS byte address, OFF = immediate byte offset (< 65280).
template <int OFF>
__device__ __forceinline__ bf16x8 tr_frag_asm(unsigned addr) {
    s16x4 lo4, hi4;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(lo4) : "v"(addr), "n"(OFF));
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(hi4) : "v"(addr), "n"(OFF + 4 * 64));
    const s16x8 v = {lo4[0], lo4[1], lo4[2], lo4[3], hi4[0], hi4[1], hi4[2], hi4[3]};
    return __builtin_bit_cast(bf16x8, v);
}

// sum of the 8 bf16 values of a fragment, added to acc (4 v_dot2c_f32_bf16 against (1, 1): exact products, f32 accumulate)
__device__ __forceinline__ float frag_sum(bf16x8 v, float acc) {
#ifndef T2S_SPLIT_F16
    typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
    const bf16x2_t one = {(__bf16)1.0f, (__bf16)1.0f};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const bf16x2_t p = {v[2 * i], v[2 * i + 1]};
        acc = __builtin_amdgcn_fdot2_f32_bf16(p, one, acc, false);
    }
#else       // (diagnostic fp16 build: the training path is not meant to run there; kept compilable)
    typedef __attribute__((ext_vector_type(2))) _Float16 f16x2_t;
    const f16x2_t one = {(_Float16)1.0f, (_Float16)1.0f};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const f16x2_t p = {v[2 * i], v[2 * i + 1]};
        acc = __builtin_amdgcn_fdot2(p, one, acc, false);
    }
#endif
    return acc;
}

}  // namespace

__global__ __launch_bounds__(512) void wgrad_cl_kernel(const WgradClArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;

    // XCD-aware bijective remap, M tiles fastest (they share the B tiles of one slab in that XCD's L2)
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int q8 = nwg >> 3, r8 = nwg & 7, xcd = bid & 7;
    const int logical = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
    const int mt = logical % a.n_mtiles;
    const int rest = logical / a.n_mtiles;
    const int nt = rest % a.n_ntiles;
    const int slab = rest / a.n_ntiles;
    const int kper = a.k1 - a.k0;
    const int kf0 = slab * a.kchunk;
    int nk = min(a.kchunk, a.B * kper - kf0);
    if (nk < 0) nk = 0;

    // ---- DMA sources: call j (0, 1) of a plane tile covers chunks 4j .. 4j+3; this thread's piece: chunk 4j + (tid >> 7),
    // row (tid >> 2) & 31, LDS 16-byte slot tid & 3, which holds the logical slot with the 32-byte half swapped on rows 8..15,
    // 24..31 (the same involution is applied to the read address) ----
    const int row = (tid >> 2) & 31, slot = tid & 3;
    const int lslot = (((slot >> 1) ^ ((row >> 3) & 1)) << 1) | (slot & 1);
    const long thr_el = (long)row * 32 + lslot * 8;                      // u16 elements inside a chunk's K-block
    const u16 *ah[2], *al[2], *bh[2], *bl[2];
    long abs_[2], bbs_[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const WgradChunk ca = a.a_chunks[mt * 8 + 4 * j + (tid >> 7)];
        const WgradChunk cb = a.b_chunks[nt * 8 + 4 * j + (tid >> 7)];
        ah[j] = ca.hi + thr_el; al[j] = ca.lo + thr_el; abs_[j] = ca.bstride;
        bh[j] = cb.hi + thr_el; bl[j] = cb.lo + thr_el; bbs_[j] = cb.bstride;
    }
    char* const lds_wave = smem + wave * 1024;                            // + lane * 16 is implicit in the DMA
    int nbb = kf0 / kper, nkk = a.k0 + kf0 - nbb * kper;                  // (batch entry, K-block) of the next step to issue
    auto issue = [&](int ks, int buf) {
        (void)ks;
        const int bb = nbb, kk = nkk;
        if (++nkk == a.k1) { nkk = a.k0; ++nbb; }
        const long roff = (long)kk * 32 * 32;                            // 32 rows x 32 channels per K-block
        char* dst = lds_wave + buf * WG_STAGE;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            wg_glds16(ah[j] + bb * abs_[j] + roff, dst + j * 8192);
            wg_glds16(al[j] + bb * abs_[j] + roff, dst + WG_PLANE + j * 8192);
            wg_glds16(bh[j] + bb * bbs_[j] + roff, dst + 2 * WG_PLANE + j * 8192);
            wg_glds16(bl[j] + bb * bbs_[j] + roff, dst + 3 * WG_PLANE + j * 8192);
        }
    };

    // ---- transposed fragment addresses.  Lane group g = lane >> 4 owns k = 8g .. 8g+7; inside the group lane 4q + p supplies
    // the address of block row q, columns 4p .. 4p+3 (8 bytes).  16-channel half `c16` of the chunk sits in the 32-byte half
    // c16 ^ (g & 1) of the row (rows 8g .. 8g+7 all have (row >> 3) & 1 == g & 1). ----
    const int g = lane >> 4, qq = (lane >> 2) & 3, pp = lane & 3;
    const int tr_base = (8 * g + qq) * 64 + pp * 8;
    const int half0 = (0 ^ (g & 1)) * 32, half1 = (1 ^ (g & 1)) * 32;
    // A m-tile mi (0..7) of this wave: chunk wr*4 + (mi >> 1), channel half mi & 1;  B n-tile ni (0..3): chunk wc*2 + (ni >> 1)
    const int a_frag = wr * 4 * 2048 + tr_base;
    const int b_frag = 2 * WG_PLANE + wc * 2 * 2048 + tr_base;

    f32x4 acc[8][4];
#pragma unroll
    for (int m = 0; m < 8; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n) acc[m][n] = (f32x4){0.f, 0.f, 0.f, 0.f};

    if (nk > 0) issue(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int ks = 0; ks < nk; ++ks) {
        const char* sb = smem + (ks & 1) * WG_STAGE;
        if (ks + 1 < nk) issue(ks + 1, (ks + 1) & 1);
        bf16x8 fbh[4], fbl[4];
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            const int off = b_frag + (n >> 1) * 2048 + ((n & 1) ? half1 : half0);
            fbh[n] = tr_frag(sb + off);
            fbl[n] = tr_frag(sb + off + WG_PLANE);
        }
        // the A fragments of m-tile m + 1 are fetched while m's 12 MFMAs issue
        bf16x8 fah = tr_frag(sb + a_frag + half0);
        bf16x8 fal = tr_frag(sb + a_frag + half0 + WG_PLANE);
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            bf16x8 fah_n = fah, fal_n = fal;
            if (m + 1 < 8) {
                const int off = a_frag + ((m + 1) >> 1) * 2048 + (((m + 1) & 1) ? half1 : half0);
                fah_n = tr_frag(sb + off);
                fal_n = tr_frag(sb + off + WG_PLANE);
            }
#pragma unroll
            for (int n = 0; n < 4; ++n) acc[m][n] = T2S_MFMA32(fal, fbh[n], acc[m][n], 0, 0, 0);
#pragma unroll
            for (int n = 0; n < 4; ++n) acc[m][n] = T2S_MFMA32(fah, fbl[n], acc[m][n], 0, 0, 0);
#pragma unroll
            for (int n = 0; n < 4; ++n) acc[m][n] = T2S_MFMA32(fah, fbh[n], acc[m][n], 0, 0, 0);
            fah = fah_n;
            fal = fal_n;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }

    // ---- epilogue: C/D map of mfma 16x16: col = lane & 15 (n), row = 4 * (lane >> 4) + reg (m) ----
    float* P = a.P + (size_t)slab * a.M * a.ldp;
    const int ncol = lane & 15, mrow = (lane >> 4) * 4;
#pragma unroll
    for (int m = 0; m < 8; ++m) {
        const int mm = mt * 256 + wr * 128 + m * 16 + mrow;
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            const int nn = nt * 256 + wc * 64 + n * 16 + ncol;
            if (nn >= a.N) continue;
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (mm + e < a.M) P[(size_t)(mm + e) * a.ldp + nn] = acc[m][n][e];
        }
    }
}


// ------------------------------------------------------------------------------------------------------------------------
// The same GEMM on the ping-pong schedule of csrc/gate_gemm_pp.hip (round 3; the kernel above is the round-2 lockstep loop,
// T2S_WGRAD_PP=0).  What carries over unchanged: the 256 x 256 tile, 32-step K-blocks, the LDS image and its half-row swap, the
// DMA pieces, split-bf16 products.  What changes:
//  * waves 0-3 / 4-7 form two groups (one wave of each on every SIMD) that run one barrier apart: in every barrier interval one
//    wave of a SIMD issues its 24 MFMAs while its partner reads LDS (transposed 8-byte reads) and issues the fill;
//  * a wave's 128 x 64 block is 64 rows from each 128-row half of the A tile (chunks 4h + 2 wr, + 1) and 32 columns from each
//    half of the B tile (chunk 4h + wc): a K-step is four phases, each consuming one (A half, B half) pair, and every 8 KB
//    half-plane of LDS is read in exactly one phase (16 / 8 / 16 / 0 ds_read_b64_tr_b16 pairs per phase);
//  * the fill is one half-tile (hi + lo = 2 DMA instructions per wave) per phase, issued five phases ahead of its first read
//    and retired by a counted s_waitcnt vmcnt(6) three phases later - never drained inside the loop (raw s_barrier, no
//    __syncthreads()); the source pointers are running per-thread pointers (one add per issue; a batch boundary is a
//    different addend, selected by a wave-uniform flag) instead of table lookups and 64-bit multiplies per K-step;
//  * the MFMA operands are swapped (B fragment as the matrix-A operand), so a lane's four accumulator registers are four
//    CONSECUTIVE COLUMNS of one output row: the epilogue is 32 16-byte stores per lane instead of 128 4-byte ones (the slab's
//    leading dimension ldp is a multiple of 4 floats for that).
//  * bias_cols (the res/skip weight gradient, N = C exactly two tiles wide): the bias gradient is the row sum of the A operand
//    over K.  As an extra all-ones output column it would open a third column of tiles for one column (+50 % workgroups); here
//    the workgroups of tile column 0 sum their A fragments on the VALU (v_dot2c_f32_bf16 against (1, 1), issued in the shadow of
//    the MFMAs): the four waves that share a block of A rows take the K-steps in turn (ks % 4 == wc), so each adds 32 VALU
//    instructions to one K-step in four, and write four partial sums to columns N .. N+3 of the slab (wn_backward adds them).
// Phase table of K-step ks (LDS buffer ks & 1), as csrc/gate_gemm_pp.hip:
//      p  reads                     MFMAs (x3)                      stages
//      0  A half 0, B half 0        acc[0..3][0..1]                 B half 1 of K-step ks+1
//      1  B half 1                  acc[0..3][2..3]                 A half 1 of K-step ks+1
//      2  A half 1                  acc[4..7][2..3]                 A half 0 of K-step ks+2
//      3  -                         acc[4..7][0..1]                 B half 0 of K-step ks+2
#ifdef T2S_GEMM_STAMPS
__device__ unsigned long long t2s_wgpp_stamps[1024 * 8];
#define WG_STAMP(i) if (tid == 0) t2s_wgpp_stamps[(blockIdx.x & 1023) * 8 + (i)] = __builtin_amdgcn_s_memtime();
#define WG_RSTAMP(i) if (tid == 0) t2s_wgpp_stamps[(blockIdx.x & 1023) * 8 + (i)] = __builtin_amdgcn_s_memrealtime();
extern "C" int t2s_debug_read_wgpp_stamps(unsigned long long* host_out, int n_words) {
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(t2s_wgpp_stamps), sizeof(unsigned long long) * n_words);
}
#else
#define WG_STAMP(i)
#define WG_RSTAMP(i)
#endif

__global__ __launch_bounds__(512) void wgrad_cl_pp_kernel(const WgradClArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2;          // group (SIMD partners are wave w and w + 4) = which 64 rows of each A half
    const int wc = wave & 3;           // which 32 columns of each B half
    WG_STAMP(0)
    WG_RSTAMP(4)

    const int nwg = gridDim.x, bid = blockIdx.x;
    const int q8 = nwg >> 3, r8 = nwg & 7, xcd = bid & 7;
    const int logical = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
    const int mt = logical % a.n_mtiles;
    const int rest = logical / a.n_mtiles;
    const int nt = rest % a.n_ntiles;
    const int slab = rest / a.n_ntiles;
    const int kper = a.k1 - a.k0;
    const int kf0 = slab * a.kchunk;
    int nk = min(a.kchunk, a.B * kper - kf0);
    if (nk < 0) nk = 0;

    // ---- running DMA sources.  Piece of this thread in a half-plane (4 chunks x 32 rows x 64 B): chunk tid >> 7, row
    // (tid >> 2) & 31, 16-byte slot tid & 3 holding the logical slot with the 32-byte half swapped on rows 8..15, 24..31 ----
    const int row = (tid >> 2) & 31, slot = tid & 3;
    const int lslot = (((slot >> 1) ^ ((row >> 3) & 1)) << 1) | (slot & 1);
    const int bb0 = kf0 / kper, kk0 = kf0 - bb0 * kper;          // first K-block of the slab: batch entry, block inside it
    const long thr_b = ((long)row * 32 + lslot * 8) * 2 + (long)(a.k0 + kk0) * 2048;
    const int step_b = 2048;                                      // one K-block = 32 rows x 64 B of a chunk
    const char *pAh[2], *pAl[2], *pBh[2], *pBl[2];
    int jA[2], jB[2];                                             // addend across a batch boundary (last block -> block k0 of b + 1)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const WgradChunk ca = a.a_chunks[mt * 8 + 4 * h + (tid >> 7)];
        const WgradChunk cb = a.b_chunks[nt * 8 + 4 * h + (tid >> 7)];
        pAh[h] = (const char*)ca.hi + (long)bb0 * ca.bstride * 2 + thr_b;
        pAl[h] = (const char*)ca.lo + (long)bb0 * ca.bstride * 2 + thr_b;
        pBh[h] = (const char*)cb.hi + (long)bb0 * cb.bstride * 2 + thr_b;
        pBl[h] = (const char*)cb.lo + (long)bb0 * cb.bstride * 2 + thr_b;
        jA[h] = (int)(ca.bstride * 2 - (long)(kper - 1) * step_b);
        jB[h] = (int)(cb.bstride * 2 - (long)(kper - 1) * step_b);
    }
    char* const lds_wave = smem + wave * 1024;                    // + lane * 16 is implicit in the DMA
    // `edge`: the K-block being staged is the last one of its batch entry (wave-uniform)
    auto stage_a = [&](int ks, int h, bool edge) {
        char* dst = lds_wave + (ks & 1) * WG_STAGE + h * 8192;
        wg_glds16(pAh[h], dst);
        wg_glds16(pAl[h], dst + WG_PLANE);
        const int adv = edge ? jA[h] : step_b;
        pAh[h] += adv;
        pAl[h] += adv;
    };
    auto stage_b = [&](int ks, int h, bool edge) {
        char* dst = lds_wave + (ks & 1) * WG_STAGE + 2 * WG_PLANE + h * 8192;
        wg_glds16(pBh[h], dst);
        wg_glds16(pBl[h], dst + WG_PLANE);
        const int adv = edge ? jB[h] : step_b;
        pBh[h] += adv;
        pBl[h] += adv;
    };

    // ---- transposed fragment addresses (as the lockstep kernel): lane group g = lane >> 4 owns k = 8g .. 8g+7; lane 4q + p of
    // the group supplies block row q, columns 4p .. 4p+3; 16-channel half c16 of a chunk sits in 32-byte half c16 ^ (g & 1) ----
    const int g = lane >> 4, qq = (lane >> 2) & 3, pp = lane & 3;
    const int tr_base = (8 * g + qq) * 64 + pp * 8;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    // per-lane read bases for the two 16-channel halves of a chunk; everything else is an immediate offset:
    // A: + h * 8192 + (mm >> 1) * 2048 (+ WG_PLANE for lo);  B: + h * 8192 (+ WG_PLANE for lo)
    const unsigned a_base[2] = {lds0 + wr * 4096 + tr_base + (0 ^ (g & 1)) * 32, lds0 + wr * 4096 + tr_base + (1 ^ (g & 1)) * 32};
    const unsigned b_base[2] = {lds0 + 2 * WG_PLANE + wc * 2048 + tr_base + (0 ^ (g & 1)) * 32,
                                lds0 + 2 * WG_PLANE + wc * 2048 + tr_base + (1 ^ (g & 1)) * 32};

    f32x4 acc[8][4];
#pragma unroll
    for (int m = 0; m < 8; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n) acc[m][n] = (f32x4){0.f, 0.f, 0.f, 0.f};
    bf16x8 afh[4], afl[4], b0h[2], b0l[2], b1h[2], b1l[2];
    const bool bias_wg = a.bias_cols && nt == 0;
    float rs[8];                        // row sums of this wave's 8 A tiles over its share of the K-steps (bias_wg only)
#pragma unroll
    for (int m = 0; m < 8; ++m) rs[m] = 0.f;

    // position inside the batch entry of K-steps ks + 1 and ks + 2 (scalars)
    int r0 = kk0;
    auto next_r = [&](int r) { return r + 1 == kper ? 0 : r + 1; };
    // ---- prologue: K-step 0 whole, A half 0 / B half 0 of K-step 1 ----
    if (nk > 0) {
        const bool e0 = r0 == kper - 1;
        stage_a(0, 0, e0);
        stage_b(0, 0, e0);
        stage_b(0, 1, e0);
        stage_a(0, 1, e0);
    }
    int r1 = next_r(r0);               // K-step ks + 1
    int r2 = next_r(r1);               // K-step ks + 2
    if (nk > 1) {
        const bool e1 = r1 == kper - 1;
        stage_a(1, 0, e1);
        stage_b(1, 0, e1);
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    if (wr == 1) __builtin_amdgcn_s_barrier();          // group 1 runs one barrier behind group 0

    // swapped operands: D[i][j] = sum_k Bfrag[i][k] * Afrag[j][k]  ->  lane holds columns n = 4 (lane >> 4) + e of row m = lane & 15
#define WG_MFMA(ACC, AH, AL, BH, BL)                                             \
    ACC = T2S_MFMA32(BH, AL, ACC, 0, 0, 0);         \
    ACC = T2S_MFMA32(BL, AH, ACC, 0, 0, 0);         \
    ACC = T2S_MFMA32(BH, AH, ACC, 0, 0, 0);

    auto kstep = [&](int ks, auto main_tag) {
        constexpr bool MAIN = decltype(main_tag)::value;
        const unsigned sbo = (ks & 1) * WG_STAGE;
        const unsigned ab0 = a_base[0] + sbo, ab1 = a_base[1] + sbo, bb0_ = b_base[0] + sbo, bb1_ = b_base[1] + sbo;
        const bool e1 = r1 == kper - 1, e2 = r2 == kper - 1;
        // ------------------------------------------------ phase 0: A half 0 x B half 0
        b0h[0] = tr_frag_asm<0>(bb0_);
        b0l[0] = tr_frag_asm<WG_PLANE>(bb0_);
        b0h[1] = tr_frag_asm<0>(bb1_);
        b0l[1] = tr_frag_asm<WG_PLANE>(bb1_);
        afh[0] = tr_frag_asm<0>(ab0);
        afl[0] = tr_frag_asm<WG_PLANE>(ab0);
        afh[1] = tr_frag_asm<0>(ab1);
        afl[1] = tr_frag_asm<WG_PLANE>(ab1);
        afh[2] = tr_frag_asm<2048>(ab0);
        afl[2] = tr_frag_asm<WG_PLANE + 2048>(ab0);
        afh[3] = tr_frag_asm<2048>(ab1);
        afl[3] = tr_frag_asm<WG_PLANE + 2048>(ab1);
        if (MAIN || ks + 1 < nk) stage_b(ks + 1, 1, e1);
        if (MAIN) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int n = 0; n < 2; ++n) { WG_MFMA(acc[m][n], afh[m], afl[m], b0h[n], b0l[n]) }
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        // ------------------------------------------------ phase 1: A half 0 x B half 1
        b1h[0] = tr_frag_asm<8192>(bb0_);
        b1l[0] = tr_frag_asm<8192 + WG_PLANE>(bb0_);
        b1h[1] = tr_frag_asm<8192>(bb1_);
        b1l[1] = tr_frag_asm<8192 + WG_PLANE>(bb1_);
        if (MAIN || ks + 1 < nk) stage_a(ks + 1, 1, e1);
        if (MAIN) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int n = 0; n < 2; ++n) { WG_MFMA(acc[m][2 + n], afh[m], afl[m], b1h[n], b1l[n]) }
        if (bias_wg && (ks & 3) == wc) {
#pragma unroll
            for (int m = 0; m < 4; ++m) rs[m] = frag_sum(afl[m], frag_sum(afh[m], rs[m]));
        }
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        // ------------------------------------------------ phase 2: A half 1 x B half 1
        afh[0] = tr_frag_asm<8192>(ab0);
        afl[0] = tr_frag_asm<8192 + WG_PLANE>(ab0);
        afh[1] = tr_frag_asm<8192>(ab1);
        afl[1] = tr_frag_asm<8192 + WG_PLANE>(ab1);
        afh[2] = tr_frag_asm<8192 + 2048>(ab0);
        afl[2] = tr_frag_asm<8192 + WG_PLANE + 2048>(ab0);
        afh[3] = tr_frag_asm<8192 + 2048>(ab1);
        afl[3] = tr_frag_asm<8192 + WG_PLANE + 2048>(ab1);
        if (MAIN || ks + 2 < nk) stage_a(ks + 2, 0, e2);
        if (MAIN) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int n = 0; n < 2; ++n) { WG_MFMA(acc[4 + m][2 + n], afh[m], afl[m], b1h[n], b1l[n]) }
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        // ------------------------------------------------ phase 3: A half 1 x B half 0 (fragments still in registers)
        if (MAIN || ks + 2 < nk) stage_b(ks + 2, 0, e2);
        if (MAIN) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int n = 0; n < 2; ++n) { WG_MFMA(acc[4 + m][n], afh[m], afl[m], b0h[n], b0l[n]) }
        if (bias_wg && (ks & 3) == wc) {
#pragma unroll
            for (int m = 0; m < 4; ++m) rs[4 + m] = frag_sum(afl[m], frag_sum(afh[m], rs[4 + m]));
        }
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        r1 = r2;
        r2 = next_r(r2);
    };

    int ks = 0;
    WG_STAMP(1)
    for (; ks + 2 < nk; ++ks) kstep(ks, std::true_type{});
    for (; ks < nk; ++ks) kstep(ks, std::false_type{});
    if (wr == 0) __builtin_amdgcn_s_barrier();          // matches group 1's extra barrier in front of the loop
    WG_STAMP(2)
#undef WG_MFMA

    // ---- epilogue.  D map with the operands swapped: row (n) = 4 * (lane >> 4) + reg, col (m) = lane & 15.
    // m-tile m: output rows (m >> 2) * 128 + wr * 64 + (m & 3) * 16;  n-tile n: output columns (n >> 1) * 128 + wc * 32 + (n & 1) * 16
    float* P = a.P + (size_t)slab * a.M * a.ldp;
    const int mlane = lane & 15, ncol4 = (lane >> 4) * 4;
#pragma unroll
    for (int m = 0; m < 8; ++m) {
        const int mm = mt * 256 + (m >> 2) * 128 + wr * 64 + (m & 3) * 16 + mlane;
        if (mm >= a.M) continue;
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            const int nn = nt * 256 + (n >> 1) * 128 + wc * 32 + (n & 1) * 16 + ncol4;
            if (nn < a.ldp) *(f32x4*)(P + (size_t)mm * a.ldp + nn) = acc[m][n];
        }
    }
    if (bias_wg) {       // lane l holds the sum over k-group l >> 4 of row l & 15: fold the four groups, lanes 0-15 store
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            float v = rs[m];
            v += __shfl_xor(v, 16, 64);
            v += __shfl_xor(v, 32, 64);
            const int mm = mt * 256 + (m >> 2) * 128 + wr * 64 + (m & 3) * 16 + mlane;
            if (lane < 16 && mm < a.M) P[(size_t)mm * a.ldp + a.N + wc] = v;
        }
    }
#ifdef T2S_GEMM_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
    WG_STAMP(3)
    WG_RSTAMP(5)
}

hipError_t t2s_launch_wgrad_cl(const WgradClArgs& a, hipStream_t stream) {
    const int nwg = a.n_mtiles * a.n_ntiles * a.nslab;
    constexpr int lds = 2 * WG_STAGE;
    static const int pp = getenv("T2S_WGRAD_PP") ? atoi(getenv("T2S_WGRAD_PP")) : 1;
    if (a.bias_cols && !(pp && a.ldp % 4 == 0 && a.ldp >= a.N + 4)) return hipErrorInvalidValue;    // only the ping-pong kernel has it
    if (pp && a.ldp % 4 == 0) {
        static std::atomic<unsigned long long> attr_mask_pp{0};
        const hipError_t e = t2s_raise_lds_limit((const void*)wgrad_cl_pp_kernel, lds, attr_mask_pp);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(wgrad_cl_pp_kernel, dim3(nwg), dim3(512), lds, stream, a);
        return hipGetLastError();
    }
    static std::atomic<unsigned long long> attr_mask{0};
    const hipError_t e = t2s_raise_lds_limit((const void*)wgrad_cl_kernel, lds, attr_mask);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(wgrad_cl_kernel, dim3(nwg), dim3(512), lds, stream, a);
    return hipGetLastError();
}
