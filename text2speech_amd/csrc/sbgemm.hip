// Small-batch GEMM on the f32 matrix cores (gfx950): Y[item][row] = sum_k X[item][k] * W[row][k], 9..~100s of items.
//
// At batch 32 (BASELINE configs[1], the Tacotron-2 training step) the wave-per-row kernels of tacotron_ops.hip
// re-read every item's input vector once per weight row out of L2 (655 MB of L2 traffic per LSTMCell) and
// take 56-209 us per launch.  Here a workgroup owns 16 weight rows x 32 items, its 8 waves split K, and the
// products run on v_mfma_f32_16x16x4_f32 (native f32 operands, f32 accumulate - exact f32 products, so the
// recurrence keeps the accuracy of the VALU path):
//   * every weight byte is read once per launch per chip (the floor: 29 + 42 MB per decoder step),
//   * the 32 input vectors are read once per workgroup (2x the weight bytes, from L2),
//   * lane l feeds A[row l%16][k..k+3] and B[k..k+3][item l%16] with k = 16*step + 4*(l/16): one float4 per
//     operand per lane per step, 64 contiguous bytes per row per load instruction.
// Epilogues: plain (bias / activation, GemvArgs semantics) and the fused LSTMCell update (LstmCellArgs semantics,
// reference tacotron.py:366-370,380-385): there the 16 rows of a workgroup are the four gates of four hidden
// units, ordered so that one lane's four accumulators are (i, f, g, o) of one (unit, item).
#include <stdlib.h>
#include "t2s_common.h"
#include "t2s_kernels.h"
#include "tacotron_ops.h"

static __device__ __forceinline__ float sb_sigmoid(float x) { return 1.0f / (1.0f + expf(-x)); }

struct SbOperands {
    const float* W1; int ld1; int k1;
    const float* W2; int ld2;
    const float* x0; const float* x1; const float* x2; int n0, n1, n2; long sx0, sx1, sx2;
    int K;
    int items;
};

#ifdef T2S_SBGEMM_VGPR      // round-1 operand path (register ring), kept for A/B builds
struct SbFrag {
    f32x4 a[4], b0[4], b1[4];
};

// Loads the operands of (up to) four consecutive 16-wide K steps starting at step s (steps >= s_end give zeros).
static __device__ __forceinline__ void sb_load(SbFrag& f, const SbOperands& o, int s, int s_end, size_t grow, bool row_ok,
                                               int item0, int q) {
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int st = s + u;
        if (st < s_end) {                                   // wave-uniform
            const int k = st * 16;
            const float* wp = k < o.k1 ? o.W1 + grow * o.ld1 + k : o.W2 + grow * o.ld2 + (k - o.k1);
            f.a[u] = row_ok ? *(const f32x4*)(wp + q * 4) : zero;
            const float* xp;                                // no dynamic indexing of the segment arrays (scratch)
            long sx;
            if (k < o.n0) { xp = o.x0 + k; sx = o.sx0; }
            else if (k < o.n0 + o.n1) { xp = o.x1 + (k - o.n0); sx = o.sx1; }
            else { xp = o.x2 + (k - o.n0 - o.n1); sx = o.sx2; }
            xp += q * 4;
            f.b0[u] = item0 < o.items ? *(const f32x4*)(xp + (size_t)item0 * sx) : zero;
            f.b1[u] = item0 + 16 < o.items ? *(const f32x4*)(xp + (size_t)(item0 + 16) * sx) : zero;
        } else {
            f.a[u] = zero; f.b0[u] = zero; f.b1[u] = zero;
        }
    }
}

static __device__ __forceinline__ void sb_mma(const SbFrag& f, f32x4& acc0, f32x4& acc1) {
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(f.a[u][j], f.b0[u][j], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(f.a[u][j], f.b1[u][j], acc1, 0, 0, 0);
        }
}

// One workgroup's 16 rows x 32 items; partial sums of the 8 waves end in s_part[wave][half][reg][lane].
template <int KW>
static __device__ __forceinline__ void sb_core(const SbOperands& o, size_t grow, bool row_ok, int item_base,
                                               float (*s_part)[2][4][64], char*, size_t, size_t) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int q = lane >> 4, item0 = item_base + (lane & 15);
    const int nsteps = o.K >> 4;
    const int s0 = (wave * nsteps) >> 3, s1 = ((wave + 1) * nsteps) >> 3;
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    // ring of three operand groups (4 K-steps each): two groups of loads stay in flight behind the MFMAs of the third
    SbFrag f0, f1, f2;
    sb_load(f0, o, s0, s1, grow, row_ok, item0, q);
    sb_load(f1, o, s0 + 4, s1, grow, row_ok, item0, q);
    sb_load(f2, o, s0 + 8, s1, grow, row_ok, item0, q);
    for (int s = s0; s < s1; s += 12) {
        sb_mma(f0, acc0, acc1);
        sb_load(f0, o, s + 12, s1, grow, row_ok, item0, q);
        if (s + 4 < s1) sb_mma(f1, acc0, acc1);
        sb_load(f1, o, s + 16, s1, grow, row_ok, item0, q);
        if (s + 8 < s1) sb_mma(f2, acc0, acc1);
        sb_load(f2, o, s + 20, s1, grow, row_ok, item0, q);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        s_part[wave][0][r][lane] = acc0[r];
        s_part[wave][1][r][lane] = acc1[r];
    }
    __syncthreads();
}

#else
// Operand ring in LDS, filled by LDS-DMA (round 2).  The register ring above never had more than ~5 loads in flight per lane:
// hipcc puts a branch and a `vmcnt(0)` around each conditional operand load (18 full waits for 82 loads), and both rewrites
// with unconditional loads lost more to extra instructions than they won (profiles/r02_summary.md).  `global_load_lds` takes no
// VGPRs and no compiler-inserted waits: every wave owns SB_R slots of 3 KB - the A, B0 and B1 fragments of one 16-wide K-step,
// each exactly the 1 KB a wave-wide 16-byte DMA writes, in lane order, so the fragment read is a conflict-free ds_read_b128 at
// lane * 16 - keeps SB_R - 1 K-steps (12 KB) in flight and waits with a counted vmcnt.  A wave fills and drains its own slots:
// no workgroup barrier in the loop.  8 waves x 3 steps x 3 KB = 72 KB in flight per CU (96 KB of ring; the partial sums of the
// cross-wave reduction reuse each wave's first slot).  Depth is not the limit: 4 / 5 / 6 slots measure 17.98 / 18.37 / 18.54 us -
// the launch moves 126 MB from L2 into 256 CUs (every workgroup re-reads the 32 input vectors: 2/3 of the bytes) at 13 TB/s.  Ablations at 4096 x 2560, B = 32 (tools/microbench/sbgemm_bench.py): 18.3 us =
// 7.2 launch + prologue + epilogue, + 9.5 fill (52 GB/s per CU), + 4.1 MFMA of which 2.5 hide under the fill.
// KW = k values per step and fragment row.  KW = 16: a fragment row is 64 bytes - HALF a 128-byte line, the other half being the
// next step's - and the launch issues one L2 request per half line (207 G requests/s chip-wide at B = 32, 77 % of what the 128 L2
// channels accept).  KW = 32: a fragment row is one whole line (two MFMA sub-steps per fetched step), half the requests.  A wave-wide
// DMA then covers 8 rows x 128 bytes; the eight 16-byte pieces of a row are XOR-permuted by (row & 7) on the SOURCE side so that
// the sub-step's ds_read_b128 (16 rows, one piece each) spreads over the banks.
template <int KW> struct SbCfg;
template <> struct SbCfg<16> { static constexpr int R = 4, SLOT = 3 * 1024; };
template <> struct SbCfg<32> { static constexpr int R = 3, SLOT = 6 * 1024; };
template <int KW> constexpr int sb_ring_bytes() { return 8 * SbCfg<KW>::R * SbCfg<KW>::SLOT; }

static __device__ __forceinline__ void sb_glds16(const void* gsrc, void* lds_dst) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_dst, 16, 0, 0);
}
// the weight rows: read once per launch by ONE workgroup - the textbook case for nt (MI355X_MICROARCH.md nt-weights, aux = 2), and
// measured NEGATIVE here (-DT2S_SB_NT_WEIGHTS, profiles/r04_cache_policy_ab.txt, same box, alternating): teacher-forced forward at
// B = 32 35.98 / 35.86 against 34.35 / 34.54 ms, train step 93.5 / 92.7 against 89.3 / 90.1 ms.  Default policy is what ships.
#ifdef T2S_SB_NT_WEIGHTS
#define SB_W_AUX 2
#else
#define SB_W_AUX 0
#endif
static __device__ __forceinline__ void sb_glds16w(const void* gsrc, void* lds_dst) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_dst, 16, 0, SB_W_AUX);
}

// HALF: the workgroup takes 16 items instead of 32 (no second B fragment): for launches whose row count fills less than half the
// chip - the per-CU fill is what bounds the kernel, and two workgroups with half the input vectors each pull 2/3 of the bytes per CU
template <int KW, bool HALF = false>
static __device__ __forceinline__ void sb_core(const SbOperands& o, size_t grow_of_r16, bool row_ok, int item_base,
                                               float (*s_part)[2][4][64], char* ring, size_t grow_lo8, size_t grow_hi8) {
    constexpr int SB_R = SbCfg<KW>::R, SB_SLOT = SbCfg<KW>::SLOT;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int q = lane >> 4, r16 = lane & 15;
    const int nsteps = o.K / KW;
    const int s0 = (wave * nsteps) >> 3, s1 = ((wave + 1) * nsteps) >> 3;
    // every DMA reads valid memory: rows / items past the end are clamped by the caller / here and zeroed after the fragment read
    const bool v0 = item_base + r16 < o.items, v1 = !HALF && item_base + 16 + r16 < o.items;
    char* const my = ring + wave * (SB_R * SB_SLOT);
    int slot_in = 0;                                        // slot the next issue fills (wave-uniform)
    // DMA lane roles.  KW = 16: row r16, piece q.  KW = 32: rows (lane >> 3) and (lane >> 3) + 8, piece (lane & 7) ^ (row & 7).
    const int dr = KW == 16 ? r16 : (lane >> 3);
    const int dp = KW == 16 ? q : ((lane & 7) ^ (dr & 7));  // (row + 8) & 7 == row & 7
    const size_t ia0 = item_base + dr < o.items ? item_base + dr : o.items - 1;
    const size_t ia1 = item_base + dr + 8 < o.items ? item_base + dr + 8 : o.items - 1;
    const size_t ib0 = item_base + 16 + dr < o.items ? item_base + 16 + dr : o.items - 1;
    const size_t ib1 = item_base + 24 + dr < o.items ? item_base + 24 + dr : o.items - 1;
    auto issue = [&](int st) {
        const int k = st * KW;                              // wave-uniform: a step lies inside one operand segment
        const bool w1 = k < o.k1;
        const float* wbase = w1 ? o.W1 + k : o.W2 + (k - o.k1);
        const size_t wld = w1 ? o.ld1 : o.ld2;
        const float* xp;
        long sx;
        if (k < o.n0) { xp = o.x0 + k; sx = o.sx0; }
        else if (k < o.n0 + o.n1) { xp = o.x1 + (k - o.n0); sx = o.sx1; }
        else { xp = o.x2 + (k - o.n0 - o.n1); sx = o.sx2; }
        char* dst = my + slot_in * SB_SLOT;
        if constexpr (KW == 16) {
            sb_glds16w(wbase + grow_of_r16 * wld + dp * 4, dst);
            sb_glds16(xp + dp * 4 + ia0 * sx, dst + 1024);
            if constexpr (!HALF) sb_glds16(xp + dp * 4 + ib0 * sx, dst + 2048);
        } else {
            sb_glds16w(wbase + grow_lo8 * wld + dp * 4, dst);
            sb_glds16w(wbase + grow_hi8 * wld + dp * 4, dst + 1024);
            sb_glds16(xp + dp * 4 + ia0 * sx, dst + 2048);
            sb_glds16(xp + dp * 4 + ia1 * sx, dst + 3072);
            if constexpr (!HALF) {
                sb_glds16(xp + dp * 4 + ib0 * sx, dst + 4096);
                sb_glds16(xp + dp * 4 + ib1 * sx, dst + 5120);
            }
        }
        slot_in = slot_in + 1 == SB_R ? 0 : slot_in + 1;
    };
    constexpr int NDMA = (KW == 16 ? 3 : 6) - (HALF ? (KW == 16 ? 1 : 2) : 0);
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    const int nmine = s1 - s0;
    int issued = 0;                                         // steps are counted 0 .. nmine - 1
#pragma unroll 1
    for (int i = 0; i < SB_R - 1 && issued < nmine; ++i) issue(s0 + issued++);
    int slot_out = 0;
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    // fragment read offsets of this lane (row r16, logical piece q [+ 4 for the second sub-step])
    const int rd0 = KW == 16 ? lane * 16 : (r16 >> 3) * 1024 + (r16 & 7) * 128 + ((q ^ (r16 & 7)) * 16);
    const int rd1 = KW == 16 ? 0 : (r16 >> 3) * 1024 + (r16 & 7) * 128 + (((4 + q) ^ (r16 & 7)) * 16);
#pragma unroll 1
    for (int s = 0; s < nmine; ++s) {
        // steps s .. issued - 1 are in flight; step s has landed once at most the issued - s - 1 newer ones are outstanding
        if (issued - s == SB_R - 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NDMA * (SB_R - 2)) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // the last SB_R - 2 steps of the range drain
        const char* src = my + slot_out * SB_SLOT;
        constexpr int OPB = KW == 16 ? 1024 : 2048;         // bytes per operand in a slot
        f32x4 fa = *(const f32x4*)(src + rd0);
        f32x4 fb0 = *(const f32x4*)(src + OPB + rd0);
        f32x4 fb1 = zero;
        if constexpr (!HALF) fb1 = *(const f32x4*)(src + 2 * OPB + rd0);
        f32x4 ga = zero, gb0 = zero, gb1 = zero;
        if constexpr (KW == 32) {
            ga = *(const f32x4*)(src + rd1);
            gb0 = *(const f32x4*)(src + OPB + rd1);
            if constexpr (!HALF) gb1 = *(const f32x4*)(src + 2 * OPB + rd1);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        slot_out = slot_out + 1 == SB_R ? 0 : slot_out + 1;
        // the slot read in the PREVIOUS iteration is free (its reads were waited for there): refill it
#if !defined(SB_ABL) || !(SB_ABL & 1)
        if (issued < nmine) issue(s0 + issued++);
#else
        if (issued < nmine) issued++;                      // timing-only ablation: no fill after the prologue (results are wrong)
#endif
        if (!row_ok) { fa = zero; ga = zero; }
        if (!v0) { fb0 = zero; gb0 = zero; }
        if (!v1) { fb1 = zero; gb1 = zero; }
#if !defined(SB_ABL) || !(SB_ABL & 2)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[j], fb0[j], acc0, 0, 0, 0);
            if constexpr (!HALF) acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[j], fb1[j], acc1, 0, 0, 0);
        }
        if constexpr (KW == 32) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(ga[j], gb0[j], acc0, 0, 0, 0);
                if constexpr (!HALF) acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(ga[j], gb1[j], acc1, 0, 0, 0);
            }
        }
#else
        asm volatile("" ::"v"(fa), "v"(fb0), "v"(fb1), "v"(ga), "v"(gb0), "v"(gb1));    // timing-only ablation: no MFMA
#endif
    }
    // this wave's slots are drained (every fragment read above was waited for): its partial sums go into its own first slot
    float* part = (float*)my;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        part[(0 * 4 + r) * 64 + lane] = acc0[r];
        part[(1 * 4 + r) * 64 + lane] = acc1[r];
    }
    __syncthreads();
}
#endif

// tile row r (0..15) of item it (0..31 within the group): D[row][col]: lane = (row/4)*16 + col%16, reg = row%4, half = col/16
#ifdef T2S_SBGEMM_VGPR
template <int KW>
static __device__ __forceinline__ float sb_sum(float (*s_part)[2][4][64], const char*, int r, int it) {
    const int ln = (r >> 2) * 16 + (it & 15), half = it >> 4, reg = r & 3;
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < 8; ++w) s += s_part[w][half][reg][ln];
    return s;
}
#define SB_DECL_PART __shared__ float s_part[8][2][4][64];
#else
template <int KW>
static __device__ __forceinline__ float sb_sum(float (*)[2][4][64], const char* ring, int r, int it) {
    const int ln = (r >> 2) * 16 + (it & 15), half = it >> 4, reg = r & 3;
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < 8; ++w) s += ((const float*)(ring + w * (SbCfg<KW>::R * SbCfg<KW>::SLOT)))[(half * 4 + reg) * 64 + ln];
    return s;
}
#define SB_DECL_PART float (*s_part)[2][4][64] = nullptr;
#endif

// ---- plain epilogue: GemvArgs semantics (no split_row) -------------------------------------------------------
template <int KW, bool HALF = false>
__global__ __launch_bounds__(512) void sbgemm_plain_kernel(const GemvArgs a) {
    SB_DECL_PART
    extern __shared__ __attribute__((aligned(16))) char sb_ring[];
    SbOperands o;
    o.W1 = a.W1; o.ld1 = a.ld1; o.k1 = a.k1; o.W2 = a.W2; o.ld2 = a.ld2;
    o.x0 = a.x1; o.x1 = a.x2; o.x2 = a.x3; o.n0 = a.n1; o.n1 = a.n2; o.n2 = a.n3;
    o.sx0 = a.sx1; o.sx1 = a.sx2; o.sx2 = a.sx3;
    o.K = a.n1 + a.n2 + a.n3; o.items = a.items;
    const int row0 = blockIdx.x * 16, item_base = blockIdx.y * (HALF ? 16 : 32);
    const int lrow = row0 + (threadIdx.x & 15);
    auto clampr = [&](int r) { return (size_t)(r < a.rows ? r : a.rows - 1); };       // DMA sources stay inside the matrix
    const int dr = (threadIdx.x & 63) >> 3;
#ifdef T2S_SBGEMM_VGPR
    sb_core<KW>(o, clampr(lrow), lrow < a.rows, item_base, s_part, sb_ring, clampr(row0 + dr), clampr(row0 + dr + 8));
#else
    sb_core<KW, HALF>(o, clampr(lrow), lrow < a.rows, item_base, s_part, sb_ring, clampr(row0 + dr), clampr(row0 + dr + 8));
#endif
    const int r = threadIdx.x & 15, it = threadIdx.x >> 4;          // 512 threads = 16 rows x 32 items
    const int row = row0 + r, item = item_base + it;
    if (row < a.rows && item < a.items && (!HALF || it < 16)) {
        float y = sb_sum<KW>(s_part, sb_ring, r, it) + (a.bias1 ? a.bias1[row] : 0.f) + (a.bias2 ? a.bias2[row] : 0.f);
        if (a.act == ACT_RELU) y = fmaxf(y, 0.f);
        else if (a.act == ACT_TANH) y = tanhf(y);
        if (a.mask) y *= a.mask[(size_t)item * a.smask_item + row] ? a.mask_scale : 0.f;      // dropout draw (the hoisted prenet)
        a.y[(size_t)item * a.sy_item + (size_t)row * a.sy_row] = y;
    }
}

bool t2s_sbgemm_plain_ok(const GemvArgs& a) {
    const int K = a.n1 + a.n2 + a.n3;
    if (a.items <= 8 || a.split_row > 0 || K < 16) return false;
    if ((K & 15) || (a.n1 & 15) || (a.n2 & 15) || (a.n3 & 15) || (a.k1 & 15)) return false;
    if (a.k1 + a.k2 != K) return false;
    if ((a.ld1 & 3) || (a.W2 && (a.ld2 & 3)) || (a.sx1 & 3) || (a.x2 && (a.sx2 & 3)) || (a.x3 && (a.sx3 & 3))) return false;
    if (((uintptr_t)a.W1 & 15) || ((uintptr_t)a.W2 & 15) || ((uintptr_t)a.x1 & 15) || ((uintptr_t)a.x2 & 15) ||
        ((uintptr_t)a.x3 & 15))
        return false;
    return true;
}

template <int KW, typename Args, int VARIANT = 0>         // VARIANT: distinct kernels of one (KW, Args) each get their own flag
static hipError_t sb_launch(void (*kern)(const Args), const Args& a, dim3 grid, hipStream_t stream) {
#ifdef T2S_SBGEMM_VGPR
    constexpr int lds = 0;
#else
    constexpr int lds = sb_ring_bytes<KW>();
    static std::atomic<unsigned long long> attr_mask{0};         // one per (KW, Args, VARIANT) instantiation = per kernel
    const hipError_t e = t2s_raise_lds_limit((const void*)kern, lds, attr_mask);
    if (e != hipSuccess) return e;
#endif
    hipLaunchKernelGGL(kern, grid, dim3(512), lds, stream, a);
    return hipGetLastError();
}
// whole 128-byte lines per fragment row where every operand segment is a multiple of 32 (T2S_SB_STEP=16 forces the 64-byte form)
static bool sb_wide_ok(int K, int k1, int n0, int n1, int n2) {
    static const int force = getenv("T2S_SB_STEP") ? atoi(getenv("T2S_SB_STEP")) : 0;
    if (force == 16) return false;
    return !((K | k1 | n0 | n1 | n2) & 31);
}

hipError_t t2s_launch_sbgemm_plain(const GemvArgs& a, hipStream_t stream) {
    dim3 grid((a.rows + 15) / 16, (a.items + 31) / 32);
#ifndef T2S_SBGEMM_VGPR
    // 16 items per workgroup where twice the workgroups still fit one round of the chip (T2S_SB_HALF=0: never): the attention
    // cell's transposed GEMM of the BPTT loop (1792 rows = 112 workgroups at 32 items) pulls 512 instead of 768 KB per CU
    static const bool half_ok = !(getenv("T2S_SB_HALF") && atoi(getenv("T2S_SB_HALF")) == 0);
    const bool half = half_ok && !a.no_half && a.items > 16 && (long)grid.x * ((a.items + 15) / 16) <= 256;
    if (half) grid.y = (a.items + 15) / 16;
    if (!a.narrow_ring && sb_wide_ok(a.n1 + a.n2 + a.n3, a.k1, a.n1, a.n2, a.n3))
        return half ? sb_launch<32, GemvArgs, 1>(sbgemm_plain_kernel<32, true>, a, grid, stream) : sb_launch<32>(sbgemm_plain_kernel<32>, a, grid, stream);
    return half ? sb_launch<16, GemvArgs, 1>(sbgemm_plain_kernel<16, true>, a, grid, stream) : sb_launch<16>(sbgemm_plain_kernel<16>, a, grid, stream);
#else
    return sb_launch<16>(sbgemm_plain_kernel<16>, a, grid, stream);
#endif
}

// ---- fused LSTMCell epilogue: LstmCellArgs semantics ---------------------------------------------------------------
// Workgroup = hidden units u0..u0+3; tile row r = unit*4 + gate  ->  weight row gate*H + u0 + unit.
template <int KW>
__global__ __launch_bounds__(512) void sbgemm_lstm_kernel(const LstmCellArgs a) {
    SB_DECL_PART
    extern __shared__ __attribute__((aligned(16))) char sb_ring[];
    if (a.sig_ptr && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0)       // "this launch has started" (LstmCellArgs::sig_ptr)
        __hip_atomic_store(a.sig_ptr, a.sig_val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    SbOperands o;
    const int K1 = a.n1 + a.n2;
    o.W1 = a.W_ih; o.ld1 = K1; o.k1 = K1; o.W2 = a.W_hh; o.ld2 = a.H;
    o.x0 = a.x1; o.x1 = a.x2; o.x2 = a.h_in; o.n0 = a.n1; o.n1 = a.n2; o.n2 = a.H;
    o.sx0 = a.sx1; o.sx1 = a.sx2; o.sx2 = a.H;
    o.K = K1 + a.H; o.items = a.B;
    const int u0 = blockIdx.x * 4, item_base = blockIdx.y * 32;
    auto wrow = [&](int tr) { return (size_t)(tr & 3) * a.H + u0 + (tr >> 2); };       // tile row = unit * 4 + gate
    const int dr = (threadIdx.x & 63) >> 3;
    // partial attention query of this workgroup's four units (LstmCellArgs::q_part, q_dim = 128): its four columns of W_query are
    // requested now, in front of the K loop
    __shared__ float s_hq[4][32];
    f32x4 wq = {0.f, 0.f, 0.f, 0.f};
    if (a.q_part) wq = *(const f32x4*)(a.w_q + (size_t)(threadIdx.x & 127) * a.H + u0);
    // partial pre-activations (LstmCellArgs::pre_a, [B][4H]: the input half of the product, computed for a whole chunk of steps by
    // one GEMM - t2s_taco_decoder::dec_in_part): requested in front of the K loop, added in the epilogue
    float pre[4] = {0.f, 0.f, 0.f, 0.f};
    if (a.pre_a && threadIdx.x < 128) {
        const int item = item_base + (threadIdx.x >> 2), u = u0 + (threadIdx.x & 3);
        if (item < a.B) {
#pragma unroll
            for (int gi = 0; gi < 4; ++gi) pre[gi] = a.pre_a[(size_t)item * 4 * a.H + (size_t)gi * a.H + u];
        }
    }
    sb_core<KW>(o, wrow(threadIdx.x & 15), true, item_base, s_part, sb_ring, wrow(dr), wrow(dr + 8));
    if (threadIdx.x < 128) s_hq[threadIdx.x & 3][threadIdx.x >> 2] = 0.f;
    if (threadIdx.x < 128) {
        const int ul = threadIdx.x & 3, it = threadIdx.x >> 2;
        const int item = item_base + it, u = u0 + ul;
        if (item < a.B) {
            float g[4];
#pragma unroll
            for (int gi = 0; gi < 4; ++gi)
                g[gi] = (sb_sum<KW>(s_part, sb_ring, ul * 4 + gi, it) + pre[gi]) + (a.b_ih[gi * a.H + u] + a.b_hh[gi * a.H + u]);
            const size_t idx = (size_t)item * a.H + u;
            const float gi_ = sb_sigmoid(g[0]), gf = sb_sigmoid(g[1]), gg = tanhf(g[2]), go_ = sb_sigmoid(g[3]);
            const float c2 = gf * a.c[idx] + gi_ * gg;
            float h2 = go_ * tanhf(c2);
            a.c[idx] = c2;
            if (a.gates_out) {
                float* go = a.gates_out + (size_t)item * 4 * a.H + u;
                go[0] = gi_; go[a.H] = gf; go[2 * a.H] = gg; go[3 * a.H] = go_;
                a.c_out[idx] = c2;
            }
            if (a.drop_mask) h2 = a.drop_mask[idx] ? h2 * a.drop_scale : 0.f;
            a.h_out[idx] = h2;
            if (a.h_copy) a.h_copy[(size_t)item * a.s_copy + u] = h2;
            s_hq[ul][it] = h2;
        }
    }
    if (a.q_part) {
        // q_part[workgroup][item][a] = sum over this workgroup's units of W_query[a][u] * h[item][u] (tacotron.py:137 query_layer; summed
        // over the H / 4 workgroups by the energies kernel): the query GEMM leaves the serial chain of the teacher-forced decoder
        __syncthreads();
        const int aq = threadIdx.x & 127, ig = threadIdx.x >> 7;           // 4 groups of 8 items
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int it = ig * 8 + j, item = item_base + it;
            if (item < a.B)
                a.q_part[((size_t)blockIdx.x * a.B + item) * 128 + aq] =
                    (wq[0] * s_hq[0][it] + wq[1] * s_hq[1][it]) + (wq[2] * s_hq[2][it] + wq[3] * s_hq[3][it]);
        }
    }
}

bool t2s_sbgemm_lstm_ok(const LstmCellArgs& a) {
    if (a.B <= 8 || (a.H & 15) || (a.n1 & 15) || (a.n2 & 15) || a.pre_b || !a.h_in || a.ld_ih > 0 || a.w_p2) return false;
    if (a.n1 > 0 && !a.x1) return false;
    if (a.q_part && (a.q_dim != 128 || !a.w_q || ((uintptr_t)a.w_q & 15))) return false;
    if ((a.sx1 & 3) || (a.x2 && (a.sx2 & 3))) return false;
    if (((uintptr_t)a.W_ih & 15) || ((uintptr_t)a.W_hh & 15) || ((uintptr_t)a.x1 & 15) || ((uintptr_t)a.x2 & 15) ||
        ((uintptr_t)a.h_in & 15))
        return false;
    if (a.gates_out && !a.c_out) return false;
    return true;
}

hipError_t t2s_launch_sbgemm_lstm(const LstmCellArgs& a, hipStream_t stream) {
    dim3 grid(a.H / 4, (a.B + 31) / 32);
#ifndef T2S_SBGEMM_VGPR
    if (sb_wide_ok(a.n1 + a.n2 + a.H, a.n1 + a.n2, a.n1, a.n2, a.H)) return sb_launch<32>(sbgemm_lstm_kernel<32>, a, grid, stream);
#endif
    return sb_launch<16>(sbgemm_lstm_kernel<16>, a, grid, stream);
}
