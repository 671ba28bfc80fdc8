// Gate GEMM of the WaveGlow WN layer, "ping-pong" schedule for gfx950 (MI355X).
//
// Same arithmetic, operands and epilogue as conv_gemm_kernel<EPI_GATE, 256> (csrc/conv_gemm.hip: in_layers[i](x) +
// cond_layers[i](spect) as ONE split-bf16 GEMM with K = taps*C + C_cond, tanh * sigmoid gate and the folded WN.end in the
// epilogue; reference glow.py:159-162,172-175) - what changes is how the K loop is scheduled on the CU:
//
//  * the two waves that share a SIMD no longer run the same program in lockstep.  The 8 waves form two groups
//    (waves 0-3 / 4-7: one wave of each group on every SIMD).  A K-step is cut into four phases of
//    {LDS fragment reads + DMA issue | s_barrier | 24 MFMAs | s_barrier}, and group 1 runs one barrier behind group 0, so in
//    every barrier interval one wave of a SIMD feeds the matrix pipe while its partner reads LDS and issues the fill
//    (MI355X_MICROARCH.md "Two waves per SIMD" items 5 and 9; cdna_hip_programming.md section 5, the 256^2 8-phase template).
//  * the global -> LDS fill (global_load_lds_dwordx4) is issued one 16 KB half-tile per phase, five phases ahead of its first
//    read, and is never drained inside the loop: a counted s_waitcnt vmcnt(6) leaves three half-tiles in flight across the
//    raw s_barriers (no __syncthreads(): its fence would emit vmcnt(0)).
//  * a wave's 128 x 64 output block is 64 rows from each 128-row half of the A tile and 32 columns from each 128-column half
//    of the B tile, so every phase consumes exactly one (A half, B half) pair and each LDS half-tile is read in one phase only.
//
// LDS: 2 buffers x [A_hi 16 K | A_lo 16 K | B_hi 16 K | B_lo 16 K]; a half-tile = rows 128h .. 128h+127 of a (hi, lo) pair.
// Per K-step ks (buffer ks & 1), phase p:
//      p  reads (ds_read_b128)        MFMAs (x3 split-bf16 products)        stages (2 x global_load_lds per wave)
//      0  A half 0 (8), B half 0 (4)  acc[0..3][0..1] += A0 . B0            B half 1 of K-step ks+1
//      1  B half 1 (4)                acc[0..3][2..3] += A0 . B1            A half 1 of K-step ks+1
//      2  A half 1 (8)                acc[4..7][2..3] += A1 . B1            A half 0 of K-step ks+2
//      3  -                           acc[4..7][0..1] += A1 . B0            B half 0 of K-step ks+2
// A stage issued in phase q is retired by the vmcnt(6) of phase q+3 (every wave waits for its own pieces, then the barrier),
// and first read in phase q+5; a slot is re-staged no earlier than two phases after its last read (WAR), which also covers
// the one-barrier lag of group 1.
#include "t2s_common.h"
#include "t2s_kernels.h"

#include <stdlib.h>

#include <type_traits>

namespace {

constexpr int PP_PLANE = 16384;          // one plane tile: 256 rows x 64 B
constexpr int PP_HALF = 8192;            // 128 rows
constexpr int PP_BUF = 4 * PP_PLANE;     // A_hi, A_lo, B_hi, B_lo

// 8-byte plane store.  Measured negative (profiles/r02_summary.md): making these stores write-through (sc1, -DT2S_PP_SC1_STORES:
// the bytes leave the XCD's L2 during the epilogue instead of at the end-of-kernel release) costs 1.4 % on the gate GEMM and
// 1.1 % on the forward - 8-byte sc1 stores run at 0.54-0.70 x the 16-byte rate and the residual GEMM that follows no longer
// finds the gate outputs in L2.  Plain stores are the default.
__device__ __forceinline__ void pp_store8(u16* dst, u16x4 v) {
#ifdef T2S_PP_SC1_STORES
    __hip_atomic_store((unsigned long long*)dst, __builtin_bit_cast(unsigned long long, v), __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT);
#else
    *(u16x4*)dst = v;
#endif
}

__device__ __forceinline__ int pp_swz4(int rb) { return (0x78 >> (rb * 2)) & 3; }      // {0,2,3,1}[rb]

__device__ __forceinline__ void pp_glds16(const void* gsrc, void* lds_dst) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_dst, 16, 0, 0);
}

// Operand sources of one K-step.  The conv input is walked channel-chunk-major with the taps innermost: the `taps` tiles of
// one channel chunk are the same rows shifted by dil, and fetched on consecutive K-steps the second and third are L2 hits
// (tap-major they were nk_x / taps K-steps apart -- ~16 MB of other tiles per XCD in between -- and each came from the
// fabric again: 218 MB fetched per launch against 83 MB of operands).  The packed weights stay tap-major, so the A chunk is
// addressed through the cursor (aoff) instead of by the K-step number; then come the conditioning planes.  All state is
// wave-uniform (scalar registers); the per-thread part of the address is inside the base pointers.
// -DT2S_PP_TAP_MAJOR restores the tap-major walk (and with it the round-1 summation order) for A/B runs.
struct BCursor {
    long off;       // bytes from the X (or S) base
    long aoff;      // bytes from the A base: packed K-chunk index * Mpad * 64
    int kc, tap;
    bool in_s;
};

}  // namespace

// ABL: timing-only ablations of the main loop (results are wrong), instantiated only under -DT2S_GEMM_ABLATE and selected with
// env T2S_DBG_GEMM: bit 0 = no DMA, bit 1 = no MFMA, bit 2 = no LDS fragment reads.
#ifdef T2S_GEMM_STAMPS
// Diagnostic build only (cdna_hip_programming.md section 7, in-kernel stamps): s_memtime / s_memrealtime of wave 0 of every
// workgroup at four points, written to a buffer nothing else reads.  [wg][0..3] = shader-clock stamps (kernel entry, loop
// entry, loop exit, kernel exit), [wg][4..5] = 100 MHz real-time stamps at entry / exit.
__device__ unsigned long long t2s_pp_stamps[1024 * 8];
#define PP_STAMP(i)                                                                                        \
    if (tid == 0) t2s_pp_stamps[(blockIdx.x & 1023) * 8 + (i)] = __builtin_amdgcn_s_memtime();
#define PP_RSTAMP(i)                                                                                       \
    if (tid == 0) t2s_pp_stamps[(blockIdx.x & 1023) * 8 + (i)] = __builtin_amdgcn_s_memrealtime();
#else
#define PP_STAMP(i)
#define PP_RSTAMP(i)
#endif

// PH (phase mode, vocoder inference with composed conditioning weights): the 256 tile columns are 256 /
// ph_FT batch entries x ph_FT mel frames of ONE phase phi = t mod ph_P of the hop (plane rows t = ph_P * f + phi, 2 KB apart);
// the conditioning operand is the mel-window planes S[b][sc][ph_Fp][32] (row = frame) against the phase's composed weights
// A2[phi][sc][Mpad][32].  Everything else - tile, schedule, epilogue arithmetic - is the kernel above.
// EPI: EPI_GATE (the WaveGlow gate, above), or - round 3, for the training backward - EPI_RESSKIP in its accumulate form
// (every row a residual row: O (+)= acc, t2s_conv_accumulate: the data gradient of in_layers / cond_layers) and EPI_GATE_BWD
// (t2s_wg_bwd_gate_dgrad).  Those GEMMs have M = C = 512, two 256-row tiles: alone they fill half the chip, but the backward
// runs them next to the weight-gradient stream, and what counts there is the time a CU spends per unit of work - 2.1 us per
// 256 x 256 x 32 step on this schedule against 2 x 1.33 us on the lockstep 128-row tiles.
template <int ABL, bool PH, int EPI = EPI_GATE>
__global__ __launch_bounds__(512) void gate_gemm_pp_kernel(const ConvGemmArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2;          // group: 0 = waves 0-3, 1 = waves 4-7 (SIMD partners are wave w and w + 4)
    const int wc = wave & 3;
    PP_STAMP(0)
    PP_RSTAMP(4)

    // ---- XCD-aware, bijective block remap (as conv_gemm_kernel) ----
    const int nwg = gridDim.x;
    const int bid = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    const int logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    const int mt = logical % a.n_mtiles;
    const int tt_all = logical / a.n_mtiles;
    int b, t0, phi = 0, ft = 0, bt = 0;
    if constexpr (PH) {
        phi = tt_all % a.ph_P;
        const int r2 = tt_all / a.ph_P;
        ft = r2 % a.ph_nft;
        bt = r2 / a.ph_nft;
        b = bt * a.ph_bper;
        t0 = 0;
    } else {
        b = tt_all / a.n_ttiles;
        t0 = (tt_all % a.n_ttiles) * T2S_TILE_N;
    }

    // ---- DMA sources.  LDS slot p = tid (16 B each) of a half-plane: row = tid >> 2, k-chunk slot = tid & 3, which holds
    // logical chunk (tid & 3) ^ s[(row >> 2) & 3] (the swizzle lives on the SOURCE address and on the read address) ----
    const int inrow = ((tid & 3) ^ pp_swz4((tid >> 4) & 3)) * 16;
    const int thr_off = (tid >> 2) * 64 + inrow;
    const size_t a_kstride = (size_t)a.Mpad * 64;
    const size_t x_cstride = (size_t)a.Lp * 64;
    const size_t s_cstride = PH ? (size_t)a.ph_Fp * 64 : x_cstride;
    const char* const A_hi = (const char*)a.A_hi + (size_t)mt * 256 * 64 + thr_off;
    const char* const A_lo = (const char*)a.A_lo + (size_t)mt * 256 * 64 + thr_off;
    // B sources: one (hi) base per 128-row half; the lo plane is a wave-uniform byte distance away
    const long xlo_d = (const char*)a.X_lo - (const char*)a.X_hi;
    const long slo_d = (const char*)a.S_lo - (const char*)a.S_hi;
    const char* Xb[2];
    const char* Sb[2];
    long a2d_hi = 0, a2d_lo = 0;        // PH: byte distance from the packed in-layer weights to this phase's composed weights
    if constexpr (PH) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int rr = (tid >> 2) + h * 128;                  // tile column of this thread's DMA piece
            const int bl = rr / a.ph_FT, fl = rr - bl * a.ph_FT;
            int bb = b + bl;
            bb = bb < a.B ? bb : a.B - 1;                         // columns past the batch / the utterance read valid rows;
            const int f = ft * a.ph_FT + fl;                      // their results are never stored
            int t = a.ph_P * f + phi;
            t = t < a.L ? t : a.L - 1;
            const int fm = f < a.ph_Fp ? f : a.ph_Fp - 1;
            Xb[h] = (const char*)a.X_hi + (((size_t)bb * (a.xbs ? a.xbs : a.xc)) * a.Lp + a.halo + t) * 64 + inrow;
            Sb[h] = (const char*)a.S_hi + (((size_t)bb * a.sc) * a.ph_Fp + fm) * 64 + inrow;
        }
        const long ph_off = (long)phi * (long)a.sc * (long)a_kstride;
        a2d_hi = ((const char*)a.A2_hi + ph_off) - (const char*)a.A_hi;
        a2d_lo = ((const char*)a.A2_lo + ph_off) - (const char*)a.A_lo;
    } else {
        Xb[0] = (const char*)a.X_hi + (((size_t)b * (a.xbs ? a.xbs : a.xc)) * a.Lp + a.halo + t0) * 64 + thr_off;
        Sb[0] = (const char*)a.S_hi + ((size_t)b * a.sc * a.Lp + a.halo + t0) * 64 + thr_off;
        Xb[1] = Xb[0] + PP_HALF;
        Sb[1] = Sb[0] + PP_HALF;
    }
    char* const lds_wave = smem + wave * 1024;            // + lane * 16 is implicit in the DMA

    const int nk = a.nk;
    auto cursor_at = [&](int ks) {
        BCursor c;
        if (ks < a.nk_x) {
#ifdef T2S_PP_TAP_MAJOR
            c.tap = ks / a.xc;
            c.kc = ks - c.tap * a.xc;
#else
            c.kc = ks / a.taps;
            c.tap = ks - c.kc * a.taps;
#endif
            c.in_s = false;
            c.off = (long)c.kc * (long)x_cstride + (long)((c.tap - (a.taps >> 1)) * a.dil) * 64;
            c.aoff = (long)(c.tap * a.xc + c.kc) * (long)a_kstride;
        } else {
            c.tap = a.taps;
            c.kc = ks - a.nk_x;
            c.in_s = true;
            c.off = (long)c.kc * (long)s_cstride;
            c.aoff = (long)(PH ? c.kc : ks) * (long)a_kstride;
        }
        return c;
    };
#ifdef T2S_PP_TAP_MAJOR
    auto advance = [&](BCursor& c) {
        c.kc += 1;
        c.off += c.in_s ? (long)s_cstride : (long)x_cstride;
        c.aoff += (long)a_kstride;
        if (!c.in_s && c.kc == a.xc) {
            c.kc = 0;
            c.tap += 1;
            if (c.tap == a.taps) {
                c.in_s = true;
                c.off = 0;
                if (PH) c.aoff = 0;
            } else {
                c.off = (long)((c.tap - (a.taps >> 1)) * a.dil) * 64;
            }
        }
    };
#else
    const long tap_step = (long)a.dil * 64;
    const long a_tap_step = (long)a.xc * (long)a_kstride;
    auto advance = [&](BCursor& c) {
        if (c.in_s) {
            c.kc += 1;
            c.off += (long)s_cstride;
            c.aoff += (long)a_kstride;
            return;
        }
        c.tap += 1;
        c.off += tap_step;
        c.aoff += a_tap_step;
        if (c.tap == a.taps) {
            c.tap = 0;
            c.kc += 1;
            c.off = (long)c.kc * (long)x_cstride - (long)(a.taps >> 1) * tap_step;
            c.aoff = (long)c.kc * (long)a_kstride;
            if (c.kc == a.xc) {
                c.in_s = true;
                c.tap = a.taps;
                c.kc = 0;
                c.off = 0;
                c.aoff = PH ? 0 : (long)a.nk_x * (long)a_kstride;
            }
        }
    };
#endif
    auto stage_a = [&](int ks, const BCursor& c, int half) {
        char* dst = lds_wave + (ks & 1) * PP_BUF + half * PP_HALF;
        const long off = c.aoff + (long)half * PP_HALF;
        pp_glds16(A_hi + off + ((PH && c.in_s) ? a2d_hi : 0), dst);
        pp_glds16(A_lo + off + ((PH && c.in_s) ? a2d_lo : 0), dst + PP_PLANE);
    };
    auto stage_b = [&](int ks, const BCursor& c, int half) {
        char* dst = lds_wave + (ks & 1) * PP_BUF + 2 * PP_PLANE + half * PP_HALF;
        const char* src = (c.in_s ? Sb[half] : Xb[half]) + c.off;
        pp_glds16(src, dst);
        pp_glds16(src + (c.in_s ? slo_d : xlo_d), dst + PP_PLANE);
    };

    // ---- per-lane fragment read offsets: row = lane & 15, logical k-chunk = lane >> 4 ----
    const int frag_off = (lane & 15) * 64 + (((lane >> 4) ^ pp_swz4((lane >> 2) & 3)) * 16);
    const int a_frag = wr * 4096 + frag_off;                       // + half * 8192 + mm * 1024 (+ PP_PLANE for lo)
    const int b_frag = 2 * PP_PLANE + wc * 2048 + frag_off;        // + half * 8192 + nn * 1024 (+ PP_PLANE for lo)

    f32x4 acc[8][4];
#pragma unroll
    for (int m = 0; m < 8; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n) acc[m][n] = (f32x4){0.f, 0.f, 0.f, 0.f};
    bf16x8 afh[4], afl[4], b0h[2], b0l[2], b1h[2], b1l[2];

    // ---- prologue: K-step 0 whole, A half 0 / B half 0 of K-step 1 (the steady-state lead) ----
    BCursor c1 = cursor_at(0);          // becomes the cursor of K-step ks + 1
    stage_a(0, c1, 0);
    stage_b(0, c1, 0);
    stage_b(0, c1, 1);
    stage_a(0, c1, 1);
    advance(c1);                        // K-step 1
    BCursor c2 = c1;                    // cursor of K-step ks + 2
    if (nk > 1) {
        stage_a(1, c1, 0);
        stage_b(1, c1, 0);
        advance(c2);
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    if (wr == 1) __builtin_amdgcn_s_barrier();          // group 1 runs one barrier behind group 0

#define PP_MFMA(ACC, AH, AL, BH, BL)                                                 \
    if (!(ABL & 2)) {                                                                \
        ACC = T2S_MFMA32(AL, BH, ACC, 0, 0, 0);         \
        ACC = T2S_MFMA32(AH, BL, ACC, 0, 0, 0);         \
        ACC = T2S_MFMA32(AH, BH, ACC, 0, 0, 0);         \
    } else {                                                                         \
        asm volatile("" ::"v"(AH), "v"(AL), "v"(BH), "v"(BL));                       \
    }

    // One phase = read segment, barrier, matrix segment, barrier.  `MAIN` K-steps issue a stage in every phase and wait
    // with a counted vmcnt(6); the last two K-steps have nothing (or less) left to stage and drain with vmcnt(0).
    auto kstep = [&](int ks, auto main_tag) {
        constexpr bool MAIN = decltype(main_tag)::value;
        const char* sb = smem + (ks & 1) * PP_BUF;
        // ------------------------------------------------ phase 0: A half 0 x B half 0
        if (!(ABL & 4) || !MAIN) {
#pragma unroll
        for (int n = 0; n < 2; ++n) {
            b0h[n] = *(const bf16x8*)(sb + b_frag + n * 1024);
            b0l[n] = *(const bf16x8*)(sb + b_frag + PP_PLANE + n * 1024);
        }
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            afh[m] = *(const bf16x8*)(sb + a_frag + m * 1024);
            afl[m] = *(const bf16x8*)(sb + a_frag + PP_PLANE + m * 1024);
        }
        }
        if ((MAIN && !(ABL & 1)) || (!MAIN && ks + 1 < nk)) stage_b(ks + 1, c1, 1);
        if (MAIN && !(ABL & 1)) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int n = 0; n < 2; ++n) { PP_MFMA(acc[m][n], afh[m], afl[m], b0h[n], b0l[n]) }
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        // ------------------------------------------------ phase 1: A half 0 x B half 1
        if (!(ABL & 4) || !MAIN) {
#pragma unroll
        for (int n = 0; n < 2; ++n) {
            b1h[n] = *(const bf16x8*)(sb + b_frag + PP_HALF + n * 1024);
            b1l[n] = *(const bf16x8*)(sb + b_frag + PP_HALF + PP_PLANE + n * 1024);
        }
        }
        if ((MAIN && !(ABL & 1)) || (!MAIN && ks + 1 < nk)) stage_a(ks + 1, c1, 1);
        if (MAIN && !(ABL & 1)) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int n = 0; n < 2; ++n) { PP_MFMA(acc[m][2 + n], afh[m], afl[m], b1h[n], b1l[n]) }
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        // ------------------------------------------------ phase 2: A half 1 x B half 1
        if (!(ABL & 4) || !MAIN) {
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            afh[m] = *(const bf16x8*)(sb + a_frag + PP_HALF + m * 1024);
            afl[m] = *(const bf16x8*)(sb + a_frag + PP_HALF + PP_PLANE + m * 1024);
        }
        }
        if ((MAIN && !(ABL & 1)) || (!MAIN && ks + 2 < nk)) stage_a(ks + 2, c2, 0);
        if (MAIN && !(ABL & 1)) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int n = 0; n < 2; ++n) { PP_MFMA(acc[4 + m][2 + n], afh[m], afl[m], b1h[n], b1l[n]) }
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        // ------------------------------------------------ phase 3: A half 1 x B half 0 (fragments still in registers)
        if ((MAIN && !(ABL & 1)) || (!MAIN && ks + 2 < nk)) stage_b(ks + 2, c2, 0);
        if (MAIN && !(ABL & 1)) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int n = 0; n < 2; ++n) { PP_MFMA(acc[4 + m][n], afh[m], afl[m], b0h[n], b0l[n]) }
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        // cursors: c1 -> K-step ks + 2, c2 -> K-step ks + 3
        c1 = c2;
        advance(c2);
    };

    int ks = 0;
    PP_STAMP(1)
    for (; ks + 2 < nk; ++ks) kstep(ks, std::true_type{});
    for (; ks < nk; ++ks) kstep(ks, std::false_type{});
    if (wr == 0) __builtin_amdgcn_s_barrier();          // matches group 1's extra barrier in front of the loop
    PP_STAMP(2)
#undef PP_MFMA

    // ---- epilogue (conv_gemm_kernel's EPI_GATE, with this kernel's row / column ownership).
    // C/D map of mfma 16x16: col = lane & 15 (time), row = 4 * (lane >> 4) + reg (channel).
    // m-tile m of this wave covers packed rows (m >> 2) * 128 + wr * 64 + (m & 3) * 16 of the M tile: m even = tanh rows,
    // m odd = sigmoid rows of the same 16 channels (PERM_GATE groups of 32 packed rows).
    // n-tile n covers tile columns (n >> 1) * 128 + wc * 32 + (n & 1) * 16.
    const int tcol = lane & 15;
    const int rq = (lane >> 4) * 4;
    // column -> (batch entry, time step) of this lane's four n-tiles; cok = the column exists
    int cb[4], ct[4];
    bool cok[4];
#pragma unroll
    for (int n = 0; n < 4; ++n) {
        const int col = (n >> 1) * 128 + wc * 32 + (n & 1) * 16 + tcol;
        if constexpr (PH) {
            const int bl = col / a.ph_FT, fl = col - bl * a.ph_FT;
            cb[n] = b + bl;
            ct[n] = a.ph_P * (ft * a.ph_FT + fl) + phi;
        } else {
            cb[n] = b;
            ct[n] = t0 + col;
        }
        cok[n] = cb[n] < a.B && ct[n] < a.L;
    }
  if constexpr (EPI == EPI_GATE) {
    // Everything the epilogue reads from memory is requested here in one batch (the fragment registers are free now): the 16
    // bias vectors, the folded-WN.end weights and - for layers after the first - the running fold sums this wave adds to.  One
    // memory round trip instead of a dependent one in front of each (pair, half) block and a read-modify-write at the very end.
    f32x4 bt_all[4], bs_all[4];
#pragma unroll
    for (int mp = 0; mp < 4; ++mp) {
        const int prow = mt * 256 + (mp >> 1) * 128 + wr * 64 + (mp & 1) * 32 + rq;
        bt_all[mp] = *(const f32x4*)(a.bias + prow);
        bs_all[mp] = *(const f32x4*)(a.bias + prow + 16);
    }
    bf16x8 fw_h[2] = {}, fw_l[2] = {};
    if (a.fold_A) {
#pragma unroll
        for (int pair = 0; pair < 2; ++pair) {
            const u16* fa = a.fold_A + ((size_t)(mt * 4 + pair * 2 + wr) * 2 * 64 + lane) * 8;
            fw_h[pair] = *(const bf16x8*)fa;
            fw_l[pair] = *(const bf16x8*)(fa + 64 * 8);
        }
    }
    f32x4 facc[4];
#pragma unroll
    for (int n = 0; n < 4; ++n) facc[n] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (a.fold_A && !a.fold_init && lane < 32) {       // start from the running sums: the final store needs no load then
        const int slot = mt * 2 + wr;
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            const int tc = cok[n] ? ct[n] : 0, bc = cok[n] ? cb[n] : 0;
#pragma unroll
            for (int e = 0; e < 4; ++e) facc[n][e] = a.fold_acc[(((size_t)slot * a.B + bc) * 8 + rq + e) * a.L + tc];
        }
    }
#pragma unroll
    for (int pair = 0; pair < 2; ++pair) {
        u16x4 hv[2][4], lv[2][4];
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const int mp = pair * 2 + half;
            // gate in exp2 / rcp form: E = e^(2x), F = e^(-y);  tanh(x) = (E - 1) / (E + 1),  sigmoid(y) = 1 / (1 + F).
            // The bias is folded into the exponent (one fma per value); x is clamped at 10 (tanh(10) rounds to 1.0f) so that E
            // stays finite; F = inf (y < -88) gives 1 / inf = 0, the correctly rounded limit.  v_exp_f32 / v_rcp_f32 are 1 ulp.
            constexpr float L2E = 1.4426950408889634f;
            const f32x4 bt = bt_all[mp] * (2.0f * L2E), bs = bs_all[mp] * (-L2E);
            const int ch = mt * 128 + pair * 64 + wr * 32 + half * 16 + rq;             // channels ch..ch+3
            const bool chv = ch < a.C;
            const size_t obase = ((size_t)(ch >> 5) * a.Lp + a.halo) * 32 + (ch & 31);
#pragma unroll
            for (int n = 0; n < 4; ++n) {
                u16x4 hi = {0, 0, 0, 0}, lo = {0, 0, 0, 0};
                if (chv && cok[n]) {
                    const size_t o = obase + ((size_t)cb[n] * a.oc * a.Lp + (size_t)ct[n]) * 32;
                    if (a.G_hi) {      // training: sigmoid (and optionally tanh) kept for the backward pass
                        u16x4 thi, tlo, ghi, glo;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float E = __builtin_amdgcn_exp2f(fminf(fmaf(acc[2 * mp][n][e], 2.0f * L2E, bt[e]), 20.0f * L2E));
                            const float F = __builtin_amdgcn_exp2f(fmaf(acc[2 * mp + 1][n][e], -L2E, bs[e]));
                            const float tv = (E - 1.0f) * __builtin_amdgcn_rcpf(E + 1.0f);
                            const float gv = __builtin_amdgcn_rcpf(1.0f + F);
                            u16 h, l;
                            split_bf16(tv * gv, h, l);
                            hi[e] = h;
                            lo[e] = l;
                            split_bf16(tv, h, l);
                            thi[e] = h;
                            tlo[e] = l;
                            split_bf16(gv, h, l);
                            ghi[e] = h;
                            glo[e] = l;
                        }
                        if (a.T_hi) {  // the backward pass rebuilds tanh as acts / sigmoid: these planes are optional
                            pp_store8(a.T_hi + o, thi);
                            pp_store8(a.T_lo + o, tlo);
                        }
                        pp_store8(a.G_hi + o, ghi);
                        pp_store8(a.G_lo + o, glo);
                    } else {           // forward / infer: only the product is needed - one reciprocal for both
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float E = __builtin_amdgcn_exp2f(fminf(fmaf(acc[2 * mp][n][e], 2.0f * L2E, bt[e]), 20.0f * L2E));
                            const float F = __builtin_amdgcn_exp2f(fmaf(acc[2 * mp + 1][n][e], -L2E, bs[e]));
                            const float v = (E - 1.0f) * __builtin_amdgcn_rcpf((E + 1.0f) * (1.0f + F));
                            u16 h, l;
                            split_bf16(v, h, l);
                            hi[e] = h;
                            lo[e] = l;
                        }
                    }
                    pp_store8(a.O_hi + o, hi);
                    pp_store8(a.O_lo + o, lo);
                }
                hv[half][n] = hi;
                lv[half][n] = lo;
            }
        }
        if (a.fold_A) {
            // fold_A blocks are indexed by the 32-channel block of the M tile (endfold_weights_kernel: block = c >> 5)
            const bf16x8 wh = fw_h[pair], wl = fw_l[pair];
#pragma unroll
            for (int n = 0; n < 4; ++n) {
                typedef __attribute__((ext_vector_type(8))) unsigned short u16x8;
                const u16x8 bh8 = {hv[0][n][0], hv[0][n][1], hv[0][n][2], hv[0][n][3],
                                   hv[1][n][0], hv[1][n][1], hv[1][n][2], hv[1][n][3]};
                const u16x8 bl8 = {lv[0][n][0], lv[0][n][1], lv[0][n][2], lv[0][n][3],
                                   lv[1][n][0], lv[1][n][1], lv[1][n][2], lv[1][n][3]};
                const bf16x8 bh = __builtin_bit_cast(bf16x8, bh8);
                const bf16x8 bl = __builtin_bit_cast(bf16x8, bl8);
                facc[n] = T2S_MFMA32(wl, bh, facc[n], 0, 0, 0);
                facc[n] = T2S_MFMA32(wh, bl, facc[n], 0, 0, 0);
                facc[n] = T2S_MFMA32(wh, bh, facc[n], 0, 0, 0);
            }
        }
    }
    if (a.fold_A && lane < 32) {     // D rows j = 4 * (lane >> 4) + reg < 8, col = time
        const int slot = mt * 2 + wr;
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            if (!cok[n]) continue;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                a.fold_acc[(((size_t)slot * a.B + cb[n]) * 8 + rq + e) * a.L + ct[n]] = facc[n][e];
            }
        }
    }
  } else if constexpr (EPI == EPI_RESSKIP) {
    typedef __attribute__((ext_vector_type(8))) unsigned short u16x8_t;
    if (a.pair8) {
      // Rows packed with PERM_PAIR8 (t2s_pack_transposed(pair8 = 1)): the two m-tiles (m0, m0 + 1) of a 32-row group give this lane the 8
      // CONSECUTIVE channels 8 q .. 8 q + 7 of one time step - one 16-byte read-modify-write per plane where the identity order
      // needs two 8-byte ones (a quarter of a 128-byte line per request; the backward GEMMs sit at the L2's request limit in
      // their epilogue just as the forward's residual GEMM did, DESIGN.md section 5 "whole 16-byte pieces").
#pragma unroll
      for (int m0 = 0; m0 < 8; m0 += 2) {
        const int grp = mt * 256 + (m0 >> 2) * 128 + wr * 64 + (m0 & 3) * 16;      // first packed row of the 32-row group
        const bool rok = grp < a.n_res;                                            // (n_res % 32 == 0: whole groups)
        const int chc = rok ? grp + 2 * rq : 0;                                    // channel 8 q of the group
        const f32x4 b0 = *(const f32x4*)(a.bias + chc), b1 = *(const f32x4*)(a.bias + chc + 4);
        const size_t ob = ((size_t)(chc >> 5) * a.Lp + a.halo) * 32 + (chc & 31);
        u16x8_t oh[4], ol[4];
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            oh[n] = (u16x8_t){0, 0, 0, 0, 0, 0, 0, 0};
            ol[n] = oh[n];
            if (rok && cok[n] && !a.res_init) {
                const size_t o = ob + ((size_t)cb[n] * a.oc * a.Lp + (size_t)ct[n]) * 32;
                oh[n] = *(const u16x8_t*)((a.R_hi ? a.R_hi : a.O_hi) + o);
                ol[n] = *(const u16x8_t*)((a.R_lo ? a.R_lo : a.O_lo) + o);
            }
        }
        if (!rok) continue;
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            if (!cok[n]) continue;
            u16x8_t hi, lo;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float add = e < 4 ? acc[m0][n][e & 3] + b0[e & 3] : acc[m0 + 1][n][e & 3] + b1[e & 3];
                u16 h, l;
                split_bf16(join_bf16(oh[n][e], ol[n][e]) + add, h, l);
                hi[e] = h;
                lo[e] = l;
            }
            const size_t o = ob + ((size_t)cb[n] * a.oc * a.Lp + (size_t)ct[n]) * 32;
            *(u16x8_t*)(a.O_hi + o) = hi;
            *(u16x8_t*)(a.O_lo + o) = lo;
        }
      }
    } else {
    // O (+)= acc (+ bias): rows = output channels in identity order, a lane holds channels prow .. prow+3 of one time step.
    // The old values of two m-tiles (8 pieces of 8 bytes per plane) are requested together before any of them is consumed.
#pragma unroll
    for (int m0 = 0; m0 < 8; m0 += 2) {
        u16x4 oh[2][4], ol[2][4];
        f32x4 bv[2];
        size_t ob[2];
        bool rok[2];
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) {
            const int m = m0 + mi;
            const int prow = mt * 256 + (m >> 2) * 128 + wr * 64 + (m & 3) * 16 + rq;
            rok[mi] = prow < a.n_res;
            const int ch = rok[mi] ? prow : 0;
            bv[mi] = *(const f32x4*)(a.bias + ch);
            ob[mi] = ((size_t)(ch >> 5) * a.Lp + a.halo) * 32 + (ch & 31);
#pragma unroll
            for (int n = 0; n < 4; ++n) {
                oh[mi][n] = (u16x4){0, 0, 0, 0};
                ol[mi][n] = (u16x4){0, 0, 0, 0};
                if (rok[mi] && cok[n] && !a.res_init) {
                    const size_t o = ob[mi] + ((size_t)cb[n] * a.oc * a.Lp + (size_t)ct[n]) * 32;
                    oh[mi][n] = *(const u16x4*)((a.R_hi ? a.R_hi : a.O_hi) + o);
                    ol[mi][n] = *(const u16x4*)((a.R_lo ? a.R_lo : a.O_lo) + o);
                }
            }
        }
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) {
            if (!rok[mi]) continue;
#pragma unroll
            for (int n = 0; n < 4; ++n) {
                if (!cok[n]) continue;
                u16x4 hi, lo;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    u16 h, l;
                    split_bf16(join_bf16(oh[mi][n][e], ol[mi][n][e]) + (acc[m0 + mi][n][e] + bv[mi][e]), h, l);
                    hi[e] = h;
                    lo[e] = l;
                }
                const size_t o = ob[mi] + ((size_t)cb[n] * a.oc * a.Lp + (size_t)ct[n]) * 32;
                pp_store8(a.O_hi + o, hi);
                pp_store8(a.O_lo + o, lo);
            }
        }
    }
    }
  } else {
    // EPI_GATE_BWD (csrc/conv_gemm.hip has the derivation): acc = d_acts[c][t]; with the saved a = tanh * sigmoid and g = sigmoid,
    // t = a / g:  d_pre[c] = d_acts g (1 - t^2),  d_pre[C + c] = d_acts a (1 - g).  T_hi / T_lo hold a, G_hi / G_lo hold g (tc =
    // their batch stride in chunks), O planes take 2C channels (oc = their batch stride).
    typedef __attribute__((ext_vector_type(8))) unsigned short u16x8_t;
    if (a.pair8) {
      // PERM_PAIR8 rows (see EPI_RESSKIP above): 8 consecutive acts channels per lane and time step - the saved a / g planes are
      // read and both halves of d_pre written in 16-byte pieces
#pragma unroll
      for (int m0 = 0; m0 < 8; m0 += 2) {
        const int grp = mt * 256 + (m0 >> 2) * 128 + wr * 64 + (m0 & 3) * 16;
        const bool rok = grp < a.C;
        const int ch = rok ? grp + 2 * rq : 0, ch2 = ch + a.C;
        const size_t tgb = ((size_t)(ch >> 5) * a.Lp + a.halo) * 32 + (ch & 31);
        u16x8_t th[4], tl[4], gh[4], gl[4];
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            const size_t o = tgb + ((size_t)(cok[n] ? cb[n] : 0) * a.tc * a.Lp + (size_t)(cok[n] ? ct[n] : 0)) * 32;
            th[n] = *(const u16x8_t*)(a.T_hi + o);
            tl[n] = *(const u16x8_t*)(a.T_lo + o);
            gh[n] = *(const u16x8_t*)(a.G_hi + o);
            gl[n] = *(const u16x8_t*)(a.G_lo + o);
        }
        if (!rok) continue;
        const size_t o1b = ((size_t)(ch >> 5) * a.Lp + a.halo) * 32 + (ch & 31);
        const size_t o2b = ((size_t)(ch2 >> 5) * a.Lp + a.halo) * 32 + (ch2 & 31);
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            if (!cok[n]) continue;
            u16x8_t h1, l1, h2, l2;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float av = join_bf16(th[n][e], tl[n][e]), gv = join_bf16(gh[n][e], gl[n][e]);
                const float tv = gv != 0.0f ? av / gv : 0.0f;
                const float da = e < 4 ? acc[m0][n][e & 3] : acc[m0 + 1][n][e & 3];
                u16 h, l;
                split_bf16(da * gv * (1.0f - tv * tv), h, l);
                h1[e] = h;
                l1[e] = l;
                split_bf16(da * av * (1.0f - gv), h, l);
                h2[e] = h;
                l2[e] = l;
            }
            const size_t bo = ((size_t)cb[n] * a.oc * a.Lp + (size_t)ct[n]) * 32;
            *(u16x8_t*)(a.O_hi + o1b + bo) = h1;
            *(u16x8_t*)(a.O_lo + o1b + bo) = l1;
            *(u16x8_t*)(a.O_hi + o2b + bo) = h2;
            *(u16x8_t*)(a.O_lo + o2b + bo) = l2;
        }
      }
    } else {
#pragma unroll
    for (int m0 = 0; m0 < 8; m0 += 2) {
        u16x4 th[2][4], tl[2][4], gh[2][4], gl[2][4];
        bool rok[2];
        int chs[2];
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) {
            const int m = m0 + mi;
            const int ch = mt * 256 + (m >> 2) * 128 + wr * 64 + (m & 3) * 16 + rq;
            rok[mi] = ch < a.C;
            chs[mi] = rok[mi] ? ch : 0;
            const size_t tgb = ((size_t)(chs[mi] >> 5) * a.Lp + a.halo) * 32 + (chs[mi] & 31);
#pragma unroll
            for (int n = 0; n < 4; ++n) {
                const size_t o = tgb + ((size_t)(cok[n] ? cb[n] : 0) * a.tc * a.Lp + (size_t)(cok[n] ? ct[n] : 0)) * 32;
                th[mi][n] = *(const u16x4*)(a.T_hi + o);
                tl[mi][n] = *(const u16x4*)(a.T_lo + o);
                gh[mi][n] = *(const u16x4*)(a.G_hi + o);
                gl[mi][n] = *(const u16x4*)(a.G_lo + o);
            }
        }
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) {
            if (!rok[mi]) continue;
            const int ch = chs[mi], ch2 = ch + a.C;
            const size_t o1b = ((size_t)(ch >> 5) * a.Lp + a.halo) * 32 + (ch & 31);
            const size_t o2b = ((size_t)(ch2 >> 5) * a.Lp + a.halo) * 32 + (ch2 & 31);
#pragma unroll
            for (int n = 0; n < 4; ++n) {
                if (!cok[n]) continue;
                u16x4 h1, l1, h2, l2;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float av = join_bf16(th[mi][n][e], tl[mi][n][e]), gv = join_bf16(gh[mi][n][e], gl[mi][n][e]);
                    const float tv = gv != 0.0f ? av / gv : 0.0f;
                    const float da = acc[m0 + mi][n][e];
                    u16 h, l;
                    split_bf16(da * gv * (1.0f - tv * tv), h, l);
                    h1[e] = h;
                    l1[e] = l;
                    split_bf16(da * av * (1.0f - gv), h, l);
                    h2[e] = h;
                    l2[e] = l;
                }
                const size_t bo = ((size_t)cb[n] * a.oc * a.Lp + (size_t)ct[n]) * 32;
                pp_store8(a.O_hi + o1b + bo, h1);
                pp_store8(a.O_lo + o1b + bo, l1);
                pp_store8(a.O_hi + o2b + bo, h2);
                pp_store8(a.O_lo + o2b + bo, l2);
            }
        }
    }
    }
  }
#ifdef T2S_GEMM_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
    PP_STAMP(3)
    PP_RSTAMP(5)
}

template <int ABL, bool PH = false, int EPI = EPI_GATE>
static hipError_t launch_pp(const ConvGemmArgs& a, hipStream_t stream) {
    const int nwg = PH ? a.n_mtiles * a.ph_P * a.ph_nft * ((a.B + a.ph_bper - 1) / a.ph_bper) : a.n_mtiles * a.n_ttiles * a.B;
    constexpr int lds = 2 * PP_BUF;
    static std::atomic<unsigned long long> attr_mask{0};
    const hipError_t e = t2s_raise_lds_limit((const void*)gate_gemm_pp_kernel<ABL, PH, EPI>, lds, attr_mask);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((gate_gemm_pp_kernel<ABL, PH, EPI>), dim3(nwg), dim3(512), lds, stream, a);
    return hipGetLastError();
}

// the accumulate / gate-backward GEMMs of the training backward on the ping-pong schedule (256-row tiles); the caller has checked
// t2s_pp_shape_ok(a)
hipError_t t2s_launch_bwd_gemm_pp(const ConvGemmArgs& a, int epi, hipStream_t stream) {
    if (epi == EPI_RESSKIP) return launch_pp<0, false, EPI_RESSKIP>(a, stream);
    if (epi == EPI_GATE_BWD) return launch_pp<0, false, EPI_GATE_BWD>(a, stream);
    return hipErrorInvalidValue;
}
bool t2s_pp_shape_ok(const ConvGemmArgs& a) {
    return a.ksplit <= 1 && a.k0 == 0 && a.kflat == 0 && a.nk == a.nk_x + a.sc && a.nk_x == a.taps * a.xc && a.a_bstride == 0 &&
           (a.taps >> 1) * a.dil <= a.halo && a.ph_P == 0 && a.nk >= 2;
}

#ifdef T2S_GEMM_STAMPS
extern "C" int t2s_debug_read_pp_stamps(unsigned long long* host_out, int n_words) {
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(t2s_pp_stamps), sizeof(unsigned long long) * n_words);
}
#endif

hipError_t t2s_launch_gate_gemm_pp(const ConvGemmArgs& a, hipStream_t stream) {
#ifdef T2S_GEMM_ABLATE
    static const int dbg = getenv("T2S_DBG_GEMM") ? atoi(getenv("T2S_DBG_GEMM")) : 0;
    switch (dbg & 7) {
        case 1: return launch_pp<1>(a, stream);
        case 2: return launch_pp<2>(a, stream);
        case 3: return launch_pp<3>(a, stream);
        case 4: return launch_pp<4>(a, stream);
        case 5: return launch_pp<5>(a, stream);
        case 6: return launch_pp<6>(a, stream);
        case 7: return launch_pp<7>(a, stream);
        default: break;
    }
#endif
    if (a.ph_P > 0) return launch_pp<0, true>(a, stream);
    return launch_pp<0>(a, stream);
}
