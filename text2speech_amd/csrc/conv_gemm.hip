// Split-bf16 conv-as-GEMM for gfx950 (MI355X): the WaveGlow WN hot loop.
//
// Computes, per batch element b and time tile,
//     acc[o][t] = sum_{tap, c} W[o][tap][c] * X[b][c][t + (tap - taps/2) * dil]
//               + sum_{c}      Wc[o][c]     * S[b][c][t]                     (optional)
// which restates reference glow.py:159-161 (in_layers[i](audio) +
// cond_layers[i](spect)) as ONE GEMM with K = taps*C + C_cond, and
// glow.py:164 (res_skip_layers[i](acts)) as a GEMM with K = C, each with its
// element-wise tail fused into the epilogue:
//     EPI_GATE    : acts = tanh(acc[:C]) * sigmoid(acc[C:])   (glow.py:33-40)
//     EPI_RESSKIP : x += acc[:C]; skip (+)= acc[C:]           (glow.py:165-174)
//
// Arithmetic: every operand is a (hi, lo) pair of bf16 planes, and each K-step
// issues three MFMAs  a_lo*b_hi + a_hi*b_lo + a_hi*b_hi  accumulating in f32,
// i.e. ~16-bit-mantissa products with f32 accumulation.  Plain bf16 misses the
// 1e-3 parity budget by 10x; this scheme holds ~1e-5 at
// 3/16 of the cost of the exact-f32 MFMA.
//
// Tiling (one workgroup = 512 threads = 8 waves, 2 per SIMD, one WG per CU):
//   256 (out channels) x 256 (time) output tile, BK = 32 channels per K-step,
//   waves laid out 2 (M) x 4 (N), each wave 128 x 64 = 8 x 4 MFMA 16x16x32 tiles
//   (128 accumulator VGPRs).  Per K-step four 16 KiB plane tiles (A_hi, A_lo,
//   B_hi, B_lo) are DMA'd global->LDS (global_load_lds_dwordx4), double buffered
//   (2 x 64 KiB LDS).  LDS rows are 64 B; the 16-B slots of a row are XOR-permuted
//   by s[(row>>2)&3], s = {0,2,3,1}, applied on the DMA *source* address and on
//   the ds_read_b128 address, which makes every fragment read conflict-free
//   (bank groups of ds_read_b128: MI355X_MICROARCH.md, LDS table).
//
// Grid: (n_mtiles * n_ttiles * B) workgroups, remapped so that the workgroups
// dealt to one XCD (blockIdx % 8) cover contiguous time tiles x all M tiles and
// therefore share their activation tiles in that XCD's L2.
#include "t2s_common.h"
#include "t2s_kernels.h"

#include <stdlib.h>

// Timing-only ablations (results are wrong): build with -DT2S_GEMM_ABLATE and set env T2S_DBG_GEMM to
// 1 (no DMA inside the K loop), 2 (no MFMA) or 3.  Compiled out of the product build.
#ifdef T2S_GEMM_ABLATE
#define T2S_ABLATE(a) ((a).dbg)
#else
#define T2S_ABLATE(a) 0
#endif

#define B_PLANE_BYTES 16384          // one B (activation) plane tile: 256 rows x 64 B

static __device__ __forceinline__ int swz4(int rb) {      // {0,2,3,1}[rb]
    return (0x78 >> (rb * 2)) & 3;                        // 0b01'11'10'00
}

static __device__ __forceinline__ void glds16(const void* gsrc, void* lds_dst) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_dst, 16, 0, 0);
}

// MT = rows of the M (output-channel) tile: 256, or 128 for short GEMMs that would otherwise leave CUs idle.
// WN = waves along N (time): 4 -> 8 waves of 128 x 64 (two per SIMD); 2 -> 4 waves of 128 x 128 (one per SIMD, 256
// accumulator registers): a third fewer LDS fragment bytes per MFMA, the resource this kernel runs out of first.
// BD = B operand direct: the activation fragments go global -> registers (in the plane layout a 16-row x 32-channel fragment
// is one contiguous KiB, so the loads are perfectly coalesced and need no swizzle), one K-step ahead; LDS then only carries A.
// SH = shared B tile (gate GEMM, taps == 3, dilation <= 32): the three taps of a 32-channel block read rows t0-d.., t0..,
// t0+d.. of the SAME activation plane, so one extended tile of 256 + 2d rows is filled once per block and the three K-steps
// read it at row offsets 0, d, 2d - 40 KB of B per block instead of 96 KB.  The K order becomes block-major for the tap part
// (A is indexed tap*xc + c as packed, no repack); the conditioning part keeps the one-tile-per-step scheme.
// EF = early free: a wave reads ALL fragments of the current stage into registers (96 VGPRs), a barrier frees the stage, and
// the DMA for step k+2 goes into it at once - the fill gets two K-steps to land instead of one (2 LDS stages as before).
// NS = LDS stages of the plain main loop: 2, or 3 where the stage is small enough (128-row tiles: 3 x 48 KB) - those GEMMs have
// K-steps of ~0.7 us of MFMA against a ~1.7 us fill turnaround, so the fill must be issued two steps ahead.
template <int EPI, int MT, int WN, bool BD = false, bool SH = false, bool EF = false, int NS = 2>
__global__ __launch_bounds__(128 * WN) void conv_gemm_kernel(const ConvGemmArgs a) {
    constexpr int NTH = 128 * WN;                    // threads per workgroup
    constexpr int NWT = 16 / WN;                     // 16-column MFMA tiles per wave
    constexpr int CALL_BYTES = NTH * 16;             // bytes one workgroup-wide global_load_lds moves
    constexpr int A_PLANE = MT * 64;                 // bytes of one A plane tile
    constexpr int STAGE = 2 * A_PLANE + 2 * B_PLANE_BYTES;
    constexpr int MW = MT / 32;                      // 16-row MFMA tiles per wave (waves are 2 (M) x 4 (N))
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave / WN;          // 0..1  : which half of the M tile
    const int wc = wave % WN;          // which (NWT * 16)-column slice of the N tile

    // ---- XCD-aware, bijective block remap (guide section 5, T1) ----
    const int nwg = gridDim.x;
    const int bid = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    const int logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    const int mt = logical % a.n_mtiles;
    const int tt_all = logical / a.n_mtiles;
    const int bs = tt_all / a.n_ttiles;              // batch entry (x K-split): also the output slab index
    const int t0 = (tt_all % a.n_ttiles) * T2S_TILE_N;
    int b = bs, kbeg = 0, nk_split = a.nk;
    const int kper = a.kend - a.k0;                  // flattened mode: K-steps per batch entry
    int kf0 = 0;
    if (a.kflat > 0) {                               // split-K over (batch entry, time chunk) flattened
        b = 0;
        kf0 = bs * a.kchunk;
        nk_split = min(a.kchunk, a.kflat * kper - kf0);
        if (nk_split < 0) nk_split = 0;
    } else if (a.ksplit > 1 || a.k0 > 0) {           // split-K over time chunks (weight-gradient GEMMs)
        const int sp = a.ksplit > 1 ? bs % a.ksplit : 0;
        b = a.ksplit > 1 ? bs / a.ksplit : bs;
        kbeg = a.k0 + sp * a.kchunk;
        nk_split = min(a.kchunk, a.kend - kbeg);
        if (nk_split < 0) nk_split = 0;
    }

    // ---- per-thread DMA source offsets (bytes) ----
    // LDS linear slot p = j*NTH + tid (16 B each): row = p>>2 = j*NTH/4 + (tid>>2), slot q = tid&3.
    // The slot holds logical k-chunk q ^ s[(row>>2)&3]; (row>>2)&3 == (tid>>4)&3 for every j (NTH/4 is a multiple of 16).
    const int thr_off = (tid >> 2) * 64 + (((tid & 3) ^ swz4((tid >> 4) & 3)) * 16);
    const char* A_hi = (const char*)a.A_hi + (size_t)b * a.a_bstride * 2 + (size_t)mt * MT * 64 + thr_off;
    const char* A_lo = (const char*)a.A_lo + (size_t)b * a.a_bstride * 2 + (size_t)mt * MT * 64 + thr_off;
    A_hi += (size_t)kbeg * a.Mpad * 64;
    A_lo += (size_t)kbeg * a.Mpad * 64;
    const size_t a_kstride = (size_t)a.Mpad * 64;
    const size_t x_cstride = (size_t)a.Lp * 64;            // bytes per 32-channel chunk
    const int xbs = a.xbs ? a.xbs : a.xc;            // batch stride of the X planes in chunks
    const char* X_hi = (const char*)a.X_hi + (((size_t)b * xbs + kbeg) * a.Lp + a.halo + t0) * 64 + thr_off;
    const char* X_lo = (const char*)a.X_lo + (((size_t)b * xbs + kbeg) * a.Lp + a.halo + t0) * 64 + thr_off;
    const char* S_hi = (const char*)a.S_hi + ((size_t)b * a.sc * a.Lp + a.halo + t0) * 64 + thr_off;
    const char* S_lo = (const char*)a.S_lo + ((size_t)b * a.sc * a.Lp + a.halo + t0) * 64 + thr_off;
    char* lds_wave = smem + wave * 1024;                   // + lane*16 is implicit in the DMA

    // B-operand source of K-step ks (tap-major over the conv input, then the conditioning channels)
    auto b_source = [&](int ks, const char*& bh, const char*& bl) {
        if (a.kflat > 0) {
            const int kf = kf0 + ks, bb = kf / kper, kk = a.k0 + kf - bb * kper;
            const long off = ((long)bb * a.xc + kk) * (long)x_cstride;
            bh = X_hi + off;
            bl = X_lo + off;
        } else if (ks < a.nk_x) {
            const int tap = ks / a.xc;
            const int kc = ks - tap * a.xc;
            const long off = (long)kc * (long)x_cstride + (long)((tap - (a.taps >> 1)) * a.dil) * 64;
            bh = X_hi + off;
            bl = X_lo + off;
        } else {
            const long off = (long)(ks - a.nk_x) * (long)x_cstride;
            bh = S_hi + off;
            bl = S_lo + off;
        }
    };
    auto issue = [&](int ks, int buf, const char* bh, const char* bl) {
        char* dst = lds_wave + buf * STAGE;
        size_t a_off = (size_t)ks * a_kstride;
        if (a.kflat > 0) {
            const int kf = kf0 + ks, bb = kf / kper, kk = a.k0 + kf - bb * kper;
            a_off = (size_t)bb * a.a_bstride * 2 + (size_t)kk * a_kstride;
        }
        const char* ah = A_hi + a_off;
        const char* al = A_lo + a_off;
#pragma unroll
        for (int j = 0; j < A_PLANE / CALL_BYTES; ++j) {
            glds16(ah + j * CALL_BYTES, dst + j * CALL_BYTES);
            glds16(al + j * CALL_BYTES, dst + A_PLANE + j * CALL_BYTES);
        }
        if (!BD) {
#pragma unroll
            for (int j = 0; j < B_PLANE_BYTES / CALL_BYTES; ++j) {
                glds16(bh + j * CALL_BYTES, dst + 2 * A_PLANE + j * CALL_BYTES);
                glds16(bl + j * CALL_BYTES, dst + 2 * A_PLANE + B_PLANE_BYTES + j * CALL_BYTES);
            }
        }
    };
    // BD: this lane's 16 bytes of B fragment n of the tile whose DMA source would be (bh, bl)
    const long bd_adj = (long)((wc * (NWT * 16) + (lane & 15)) * 64 + (lane >> 4) * 16) - (long)thr_off;
    auto b_direct = [&](const char* bh, const char* bl, bf16x8 (&rh)[NWT], bf16x8 (&rl)[NWT]) {
#pragma unroll
        for (int n = 0; n < NWT; ++n) {
            rh[n] = *(const bf16x8*)(bh + bd_adj + n * 1024);
            rl[n] = *(const bf16x8*)(bl + bd_adj + n * 1024);
        }
    };

    // ---- per-lane fragment read offset: row = lane&15, logical k-chunk = lane>>4 ----
    const int frag_off = (lane & 15) * 64 + (((lane >> 4) ^ swz4((lane >> 2) & 3)) * 16);
    const int a_frag = wr * (MT / 2) * 64 + frag_off;
    const int b_frag = 2 * A_PLANE + wc * (NWT * 16) * 64 + frag_off;

    f32x4 acc[MW][NWT];
#pragma unroll
    for (int m = 0; m < MW; ++m)
#pragma unroll
        for (int n = 0; n < NWT; ++n) acc[m][n] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // RESSKIP on 128-row tiles (K = 512: a 13 us loop): the residual values the epilogue adds to are requested HERE, in front of
    // the K loop, and ride out their latency under it.  Ablations (tools/res_gemm_study.sh) put 25 of the kernel's 38 us outside
    // MFMA and fill: every CU reading, then writing, its 128 x 256 tile of x at the same moment (33 + 33 MB in one burst).  With the
    // read half moved under the loop only the write burst is left.  64 VGPRs; 256-row tiles have no room for it.
#ifdef T2S_NO_PREX
    constexpr bool PREX = false;
#else
    constexpr bool PREX = EPI == EPI_RESSKIP && MT == 128;
#endif
    u16x4 pre_h[PREX ? MW : 1][PREX ? NWT : 1], pre_l[PREX ? MW : 1][PREX ? NWT : 1];
    typedef __attribute__((ext_vector_type(8))) unsigned short u16x8_t;
    if constexpr (PREX) {
        const int tcol_p = lane & 15, rq_p = (lane >> 4) * 4;
        if (a.pair8) {
            // residual rows packed with PERM_PAIR8: the two m-tiles of a 32-channel group give this lane 8 consecutive channels of a
            // plane row - ONE 16-byte request per plane (the identity order needs two 8-byte ones: a quarter line each)
#pragma unroll
            for (int m = 0; m < MW; m += 2) {
                const int prow = mt * MT + wr * (MT / 2) + m * 16 + rq_p;
                const int ch = (prow & ~31) + 2 * rq_p;
                const bool is_res = ch < a.n_res && !a.res_init;
                const size_t base = (((size_t)b * a.oc + ((is_res ? ch : 0) >> 5)) * a.Lp + a.halo) * 32 + ((is_res ? ch : 0) & 31);
#pragma unroll
                for (int n = 0; n < NWT; ++n) {
                    const int t = t0 + wc * (NWT * 16) + n * 16 + tcol_p;
                    const size_t ro = base + (size_t)(t < a.L ? t : 0) * 32;
                    u16x8_t vh = {0, 0, 0, 0, 0, 0, 0, 0}, vl = {0, 0, 0, 0, 0, 0, 0, 0};
                    if (is_res) {
                        vh = *(const u16x8_t*)((a.R_hi ? a.R_hi : a.O_hi) + ro);
                        vl = *(const u16x8_t*)((a.R_lo ? a.R_lo : a.O_lo) + ro);
                    }
                    pre_h[m][n] = (u16x4){vh[0], vh[1], vh[2], vh[3]};
                    pre_h[m + 1][n] = (u16x4){vh[4], vh[5], vh[6], vh[7]};
                    pre_l[m][n] = (u16x4){vl[0], vl[1], vl[2], vl[3]};
                    pre_l[m + 1][n] = (u16x4){vl[4], vl[5], vl[6], vl[7]};
                }
            }
        } else
#pragma unroll
        for (int m = 0; m < MW; ++m) {
            const int prow = mt * MT + wr * (MT / 2) + m * 16 + rq_p;
            const bool is_res = prow < a.n_res && !a.res_init;
            const int ch = prow < a.n_res ? prow : 0;
            const size_t base = (((size_t)b * a.oc + (ch >> 5)) * a.Lp + a.halo) * 32 + (ch & 31);
#pragma unroll
            for (int n = 0; n < NWT; ++n) {
                const int t = t0 + wc * (NWT * 16) + n * 16 + tcol_p;
                const size_t ro = base + (size_t)(t < a.L ? t : 0) * 32;
                pre_h[m][n] = (u16x4){0, 0, 0, 0};
                pre_l[m][n] = (u16x4){0, 0, 0, 0};
                if (is_res) {
                    pre_h[m][n] = *(const u16x4*)((a.R_hi ? a.R_hi : a.O_hi) + ro);
                    pre_l[m][n] = *(const u16x4*)((a.R_lo ? a.R_lo : a.O_lo) + ro);
                }
            }
        }
    }

    // GATE_BWD on 128-row tiles (K = 32 steps, a 43 us loop): its epilogue reads the layer's saved gate output and sigmoid - 4 planes,
    // 66 MB per launch - and writes 66 MB of d_pre: half of the kernel's time was that read-then-write burst after the loop.  As
    // for RESSKIP above, the reads are requested HERE and ride out their latency under the K loop (128 VGPRs; -DT2S_NO_PREG: off).
#ifdef T2S_NO_PREG
    constexpr bool PREG = false;
#else
    constexpr bool PREG = EPI == EPI_GATE_BWD && MT == 128;
#endif
    u16x4 pg_th[PREG ? MW : 1][PREG ? NWT : 1], pg_tl[PREG ? MW : 1][PREG ? NWT : 1];
    u16x4 pg_gh[PREG ? MW : 1][PREG ? NWT : 1], pg_gl[PREG ? MW : 1][PREG ? NWT : 1];
    if constexpr (PREG) {
        const int tcol_p = lane & 15, rq_p = (lane >> 4) * 4;
#pragma unroll
        for (int m = 0; m < MW; ++m) {
            const int ch = mt * MT + wr * (MT / 2) + m * 16 + rq_p;
            const int chc = ch < a.C ? ch : 0;                        // rows past C read a valid row; the result is never stored
            const size_t tgb = (((size_t)b * a.tc + (chc >> 5)) * a.Lp + a.halo) * 32 + (chc & 31);
#pragma unroll
            for (int n = 0; n < NWT; ++n) {
                const int t = t0 + wc * (NWT * 16) + n * 16 + tcol_p;
                const size_t o = tgb + (size_t)(t < a.L ? t : 0) * 32;
                pg_th[m][n] = *(const u16x4*)(a.T_hi + o);
                pg_tl[m][n] = *(const u16x4*)(a.T_lo + o);
                pg_gh[m][n] = *(const u16x4*)(a.G_hi + o);
                pg_gl[m][n] = *(const u16x4*)(a.G_lo + o);
            }
        }
    }

  if constexpr (EF) {
    const int nk = nk_split;
    const char *nbh = nullptr, *nbl = nullptr;
    if (nk > 0) { b_source(0, nbh, nbl); issue(0, 0, nbh, nbl); }
    if (nk > 1) { b_source(1, nbh, nbl); issue(1, 1, nbh, nbl); }
    if (nk > 2) b_source(2, nbh, nbl);
    for (int ks = 0; ks < nk; ++ks) {
        const int cur = ks & 1;
        const char* sb = smem + cur * STAGE;
        // stage ks has landed once at most the newer DMA group (step ks+1: 8 instructions per wave) is outstanding
        if (ks + 1 < nk) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        bf16x8 fah[MW], fal[MW], fbh[NWT], fbl[NWT];
#pragma unroll
        for (int n = 0; n < NWT; ++n) {
            fbh[n] = *(const bf16x8*)(sb + b_frag + n * 1024);
            fbl[n] = *(const bf16x8*)(sb + b_frag + B_PLANE_BYTES + n * 1024);
        }
#pragma unroll
        for (int m = 0; m < MW; ++m) {
            fah[m] = *(const bf16x8*)(sb + a_frag + m * 1024);
            fal[m] = *(const bf16x8*)(sb + a_frag + A_PLANE + m * 1024);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __syncthreads();                                     // every wave holds its fragments: the stage is free
        if (ks + 2 < nk) {
            issue(ks + 2, cur, nbh, nbl);
            if (ks + 3 < nk) b_source(ks + 3, nbh, nbl);
        }
#pragma unroll
        for (int m = 0; m < MW; ++m) {
#pragma unroll
            for (int n = 0; n < NWT; ++n) acc[m][n] = T2S_MFMA32(fal[m], fbh[n], acc[m][n], 0, 0, 0);
#pragma unroll
            for (int n = 0; n < NWT; ++n) acc[m][n] = T2S_MFMA32(fah[m], fbl[n], acc[m][n], 0, 0, 0);
#pragma unroll
            for (int n = 0; n < NWT; ++n) acc[m][n] = T2S_MFMA32(fah[m], fbh[n], acc[m][n], 0, 0, 0);
        }
    }
  } else if constexpr (!SH) {
    constexpr int DMA_PER_ISSUE = 2 * (A_PLANE / CALL_BYTES) + (BD ? 0 : 2 * (B_PLANE_BYTES / CALL_BYTES));
    const int nk = nk_split;
    const char *nbh = nullptr, *nbl = nullptr;       // B sources of the next stage to issue, computed one step ahead
    bf16x8 bh[NWT], bl[NWT], bhn[NWT], bln[NWT];
    if (nk > 0) {
        b_source(0, nbh, nbl);
        issue(0, 0, nbh, nbl);
        if (BD) b_direct(nbh, nbl, bh, bl);
    }
    if (NS == 3 && nk > 1) {
        b_source(1, nbh, nbl);
        issue(1, 1, nbh, nbl);
    }
    if (nk > NS - 1) b_source(NS - 1, nbh, nbl);
    if (NS == 3 && nk > 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DMA_PER_ISSUE) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    int cur = 0;
    for (int ks = 0; ks < nk; ++ks) {
        const char* sb = smem + cur * STAGE;
        int nxt = cur + (NS - 1);                    // stage that step ks + NS - 1 goes into: freed at the end of step ks - 1
        if (nxt >= NS) nxt -= NS;
        // Order after the barrier: (1) the DMA of the stage NS-1 steps ahead, whose addresses were computed during the
        // previous step (measured: issuing it after the fragment reads instead costs 6 us per launch); (2) the fragment
        // reads the first MFMAs need; (3) the scalar address arithmetic for the next issue, hidden under the MFMAs.  The A
        // fragments of m-tile m+1 are fetched while m's 12 MFMAs issue.
        const bool issued = ks + NS - 1 < nk && !(T2S_ABLATE(a) & 1);
        if (issued) {
            issue(ks + NS - 1, nxt, nbh, nbl);
            if (BD) b_direct(nbh, nbl, bhn, bln);
        }
        bf16x8 ah = *(const bf16x8*)(sb + a_frag);
        bf16x8 al = *(const bf16x8*)(sb + a_frag + A_PLANE);
        if (!BD) {
#pragma unroll
            for (int n = 0; n < NWT; ++n) {
                bh[n] = *(const bf16x8*)(sb + b_frag + n * 1024);
                bl[n] = *(const bf16x8*)(sb + b_frag + B_PLANE_BYTES + n * 1024);
            }
        }
        if (ks + NS < nk) b_source(ks + NS, nbh, nbl);
#pragma unroll
        for (int m = 0; m < MW; ++m) {
            bf16x8 ah_n = ah, al_n = al;
            if (m + 1 < MW) {
                ah_n = *(const bf16x8*)(sb + a_frag + (m + 1) * 1024);
                al_n = *(const bf16x8*)(sb + a_frag + A_PLANE + (m + 1) * 1024);
            }
            if (!(T2S_ABLATE(a) & 2)) {
#pragma unroll
                for (int n = 0; n < NWT; ++n) acc[m][n] = T2S_MFMA32(al, bh[n], acc[m][n], 0, 0, 0);
#pragma unroll
                for (int n = 0; n < NWT; ++n) acc[m][n] = T2S_MFMA32(ah, bl[n], acc[m][n], 0, 0, 0);
#pragma unroll
                for (int n = 0; n < NWT; ++n) acc[m][n] = T2S_MFMA32(ah, bh[n], acc[m][n], 0, 0, 0);
            } else {
                asm volatile("" :: "v"(al), "v"(ah));
            }
            ah = ah_n;
            al = al_n;
        }
        // the next step's stage must have landed; with three stages the group issued in THIS step may still be in flight
        if (NS == 3 && issued) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DMA_PER_ISSUE) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (BD) {
#pragma unroll
            for (int n = 0; n < NWT; ++n) { bh[n] = bhn[n]; bl[n] = bln[n]; }
        }
        __syncthreads();
        cur = cur + 1 == NS ? 0 : cur + 1;
    }
  } else {
    // ------------------------------------------------------------------ shared-B main loop
    constexpr int BX_PLANE = 320 * 64;                     // bytes of one plane of the extended tile (<= 320 rows)
    constexpr int AR = 2 * A_PLANE;                        // one A stage (hi, lo)
    char* const Ar0 = smem;
    char* const Bx0 = smem + 2 * AR;
    const int d = a.dil, xc = a.xc;
    const int n_units = (T2S_TILE_N + 2 * d + 15) >> 4;    // 16-row units of the extended tile
    const int unit_off = (lane >> 2) * 64 + (((lane & 3) ^ swz4((lane >> 4) & 3)) * 16);
    // extended-tile sources without the workgroup-wide DMA thread offset
    const char* Xe_hi = (const char*)a.X_hi + (((size_t)b * xbs) * a.Lp + a.halo + t0 - d) * 64 + unit_off;
    const char* Xe_lo = (const char*)a.X_lo + (((size_t)b * xbs) * a.Lp + a.halo + t0 - d) * 64 + unit_off;
    auto issue_a = [&](int kidx, int buf) {
        char* dst = lds_wave + buf * AR;
        const char* ah = A_hi + (size_t)kidx * a_kstride;
        const char* al = A_lo + (size_t)kidx * a_kstride;
#pragma unroll
        for (int j = 0; j < A_PLANE / CALL_BYTES; ++j) {
            glds16(ah + j * CALL_BYTES, dst + j * CALL_BYTES);
            glds16(al + j * CALL_BYTES, dst + A_PLANE + j * CALL_BYTES);
        }
    };
    auto issue_bunit = [&](int c, int u, int buf) {       // one 16-row unit of block c's extended tile (both planes)
        char* dst = Bx0 + buf * 2 * BX_PLANE + u * 1024;
        const size_t off = (size_t)c * x_cstride + (size_t)u * 1024;
        glds16(Xe_hi + off, dst);
        glds16(Xe_lo + off, dst + BX_PLANE);
    };
    auto issue_cond = [&](int j, int buf) {                // conditioning K-step j: full A and B tiles
        issue_a(a.nk_x + j, buf);
        char* dst = Bx0 + buf * 2 * BX_PLANE + wave * 1024;
        const char* bh = S_hi + (size_t)j * x_cstride;
        const char* bl = S_lo + (size_t)j * x_cstride;
#pragma unroll
        for (int q = 0; q < B_PLANE_BYTES / CALL_BYTES; ++q) {
            glds16(bh + q * CALL_BYTES, dst + q * CALL_BYTES);
            glds16(bl + q * CALL_BYTES, dst + BX_PLANE + q * CALL_BYTES);
        }
    };
    auto compute = [&](const char* sa, const char* sbx, int boff) {
        bf16x8 bh[NWT], bl[NWT];
        bf16x8 ah = *(const bf16x8*)(sa + a_frag);
        bf16x8 al = *(const bf16x8*)(sa + a_frag + A_PLANE);
#pragma unroll
        for (int n = 0; n < NWT; ++n) {
            bh[n] = *(const bf16x8*)(sbx + boff + n * 1024);
            bl[n] = *(const bf16x8*)(sbx + BX_PLANE + boff + n * 1024);
        }
#pragma unroll
        for (int m = 0; m < MW; ++m) {
            bf16x8 ah_n = ah, al_n = al;
            if (m + 1 < MW) {
                ah_n = *(const bf16x8*)(sa + a_frag + (m + 1) * 1024);
                al_n = *(const bf16x8*)(sa + a_frag + A_PLANE + (m + 1) * 1024);
            }
#pragma unroll
            for (int n = 0; n < NWT; ++n) acc[m][n] = T2S_MFMA32(al, bh[n], acc[m][n], 0, 0, 0);
#pragma unroll
            for (int n = 0; n < NWT; ++n) acc[m][n] = T2S_MFMA32(ah, bl[n], acc[m][n], 0, 0, 0);
#pragma unroll
            for (int n = 0; n < NWT; ++n) acc[m][n] = T2S_MFMA32(ah, bh[n], acc[m][n], 0, 0, 0);
            ah = ah_n;
            al = al_n;
        }
    };
    // prologue: A(tap 0, block 0) and the whole extended tile of block 0
    issue_a(0, 0);
    for (int u = wave; u < n_units; u += 2 * WN) issue_bunit(0, u, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const int rbase = wc * (NWT * 16) + (lane & 15);       // this lane's row inside the (unshifted) tile
    const int n_tap_steps = 3 * xc;
    for (int i = 0; i < n_tap_steps; ++i) {
        const int c = i / 3, tap = i - 3 * c;
        // next A tile; at the last tap step the first conditioning stage instead
        if (i + 1 < n_tap_steps) {
            const int c1 = (i + 1) / 3, tap1 = (i + 1) - 3 * c1;
            issue_a(tap1 * xc + c1, (i + 1) & 1);
        } else if (a.sc > 0) {
            issue_cond(0, (i + 1) & 1);
        }
        // a third of the next block's extended tile per step
        if (c + 1 < xc) {
            const int u = tap * (2 * WN) + wave;
            if (u < n_units) issue_bunit(c + 1, u, (c + 1) & 1);
        }
        const int row = rbase + tap * d;
        const int boff = row * 64 + (((lane >> 4) ^ swz4((row >> 2) & 3)) * 16);
        compute(Ar0 + (i & 1) * AR, Bx0 + (c & 1) * 2 * BX_PLANE, boff);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    {
        const int boff0 = rbase * 64 + (((lane >> 4) ^ swz4((rbase >> 2) & 3)) * 16);
        for (int j = 0; j < a.sc; ++j) {
            const int st = (n_tap_steps + j) & 1;
            if (j + 1 < a.sc) issue_cond(j + 1, st ^ 1);
            compute(Ar0 + st * AR, Bx0 + st * 2 * BX_PLANE, boff0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
    }
  }

    // ---- epilogue.  C/D map of mfma 16x16: col = lane&15 (time), row = 4*(lane>>4) + reg (channel) ----
    const int tcol = lane & 15;
    const int rq = (lane >> 4) * 4;
    if (EPI == EPI_GATE) {
        // packed rows of this wave: m even = tanh rows, m odd = sigmoid rows of the same 16 channels.
        // WN.end folded in (a.fold_A): the 8 x C matrix (W_end W_skip,i) times this wave's 64 gate-output channels is
        // itself a small GEMM whose B operand is the gate output as it sits in the accumulator layout (col = time on
        // the lane, 4 consecutive channels per 16-lane group): two 16-channel tiles form one K = 32 step, with the
        // matching K permutation baked into fold_A by endfold_weights_kernel.  3 MFMAs per (pair, n), no shuffles.
        f32x4 facc[NWT];
#pragma unroll
        for (int n = 0; n < NWT; ++n) facc[n] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        // MT = 256: a wave owns 128 packed rows = 64 channels = two 32-channel fold blocks; MT = 128 (small grids: twice the
        // workgroups at half the tile): 64 packed rows = one block
        for (int pair = 0; pair < MW / 4; ++pair) {
            u16x4 hv[2][NWT], lv[2][NWT];
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                const int mp = pair * 2 + half;
                const int prow = mt * MT + wr * (MT / 2) + mp * 32 + rq;          // tanh rows prow..prow+3
                const f32x4 bt = *(const f32x4*)(a.bias + prow);
                const f32x4 bs = *(const f32x4*)(a.bias + prow + 16);
                const int ch = mt * (MT / 2) + wr * (MT / 4) + mp * 16 + rq;       // channels ch..ch+3
                const bool chv = ch < a.C;
                const size_t obase = (((size_t)b * a.oc + (ch >> 5)) * a.Lp + a.halo) * 32 + (ch & 31);
#pragma unroll
                for (int n = 0; n < NWT; ++n) {
                    const int t = t0 + wc * (NWT * 16) + n * 16 + tcol;
                    u16x4 hi = {0, 0, 0, 0}, lo = {0, 0, 0, 0};
                    if (chv && t < a.L) {
                        u16x4 thi, tlo, ghi, glo;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float tv = fast_tanh(acc[2 * mp][n][e] + bt[e]);
                            const float gv = fast_sigmoid(acc[2 * mp + 1][n][e] + bs[e]);
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
                        const size_t o = obase + (size_t)t * 32;
                        *(u16x4*)(a.O_hi + o) = hi;
                        *(u16x4*)(a.O_lo + o) = lo;
                        if (a.T_hi) {      // optional: the tanh values (the backward pass rebuilds them as acts / sigmoid)
                            *(u16x4*)(a.T_hi + o) = thi;
                            *(u16x4*)(a.T_lo + o) = tlo;
                        }
                        if (a.G_hi) {      // training: keep the sigmoid values for the backward pass
                            *(u16x4*)(a.G_hi + o) = ghi;
                            *(u16x4*)(a.G_lo + o) = glo;
                        }
                    }
                    hv[half][n] = hi;
                    lv[half][n] = lo;
                }
            }
            if (a.fold_A) {
                const u16* fa = a.fold_A + ((size_t)(mt * (MT / 64) + wr * (MT / 128) + pair) * 2 * 64 + lane) * 8;
                const bf16x8 wh = *(const bf16x8*)fa;
                const bf16x8 wl = *(const bf16x8*)(fa + 64 * 8);
#pragma unroll
                for (int n = 0; n < NWT; ++n) {
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
        if (a.fold_A && lane < 32) {     // D rows j = 4*(lane>>4) + reg < 8, col = time
            const int slot = mt * 2 + wr;
#pragma unroll
            for (int n = 0; n < NWT; ++n) {
                const int t = t0 + wc * (NWT * 16) + n * 16 + tcol;
                if (t >= a.L) continue;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float* dst = a.fold_acc + ((((size_t)slot * a.B + b) * 8 + rq + e) * a.L + t);
                    *dst = a.fold_init ? facc[n][e] : *dst + facc[n][e];
                }
            }
        }
    } else if (EPI == EPI_RESSKIP && PREX && a.pair8) {
        // t2s_wg_res_only with PERM_PAIR8 rows (residual rows only, values prefetched in front of the K loop): per m-tile pair and
        // column one 16-byte store per plane
      if constexpr (PREX) {
#pragma unroll
        for (int m = 0; m < MW; m += 2) {
            const int prow = mt * MT + wr * (MT / 2) + m * 16 + rq;
            const int ch = (prow & ~31) + 2 * rq;
            if (ch >= a.n_res) continue;
            const f32x4 b0 = *(const f32x4*)(a.bias + prow), b1 = *(const f32x4*)(a.bias + prow + 16);
            const size_t base = (((size_t)b * a.oc + (ch >> 5)) * a.Lp + a.halo) * 32 + (ch & 31);
#pragma unroll
            for (int n = 0; n < NWT; ++n) {
                const int t = t0 + wc * (NWT * 16) + n * 16 + tcol;
                if (t >= a.L) continue;
                u16x8_t hi, lo;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    u16 h, l;
                    split_bf16(join_bf16(pre_h[m][n][e], pre_l[m][n][e]) + (acc[m][n][e] + b0[e]), h, l);
                    hi[e] = h;
                    lo[e] = l;
                    split_bf16(join_bf16(pre_h[m + 1][n][e], pre_l[m + 1][n][e]) + (acc[m + 1][n][e] + b1[e]), h, l);
                    hi[4 + e] = h;
                    lo[4 + e] = l;
                }
                *(u16x8_t*)(a.O_hi + base + (size_t)t * 32) = hi;
                *(u16x8_t*)(a.O_lo + base + (size_t)t * 32) = lo;
            }
        }
      }
    } else if (EPI == EPI_RESSKIP) {
        // The residual / skip values this wave updates in place are requested in batches of m-tiles BEFORE any of them
        // is consumed: with load -> add -> store per tile (same arrays read and written, so hipcc keeps that order) the
        // epilogue was a chain of up to MW * NWT dependent memory round trips.
        constexpr int MB = MW >= 8 ? 2 : 4;        // m-tiles per batch (256-row tiles hold 128 accumulator registers already)
#pragma unroll
        for (int m0 = 0; m0 < MW; m0 += MB) {
            f32x4 bv[MB];
            u16x4 oh[MB][NWT], ol[MB][NWT];
            f32x4 sv[MB][NWT];
#pragma unroll
            for (int mi = 0; mi < MB; ++mi) {
                const int prow = mt * MT + wr * (MT / 2) + (m0 + mi) * 16 + rq;
                bv[mi] = *(const f32x4*)(a.bias + prow);
                const bool is_res = prow < a.n_res;
                const int ch = is_res ? prow : prow - a.n_res;
                const size_t base = (((size_t)b * a.oc + (ch >> 5)) * a.Lp + a.halo) * 32 + (ch & 31);
#pragma unroll
                for (int n = 0; n < NWT; ++n) {
                    const int t = t0 + wc * (NWT * 16) + n * 16 + tcol;
                    const size_t ro = base + (size_t)(t < a.L ? t : 0) * 32;
                    oh[mi][n] = (u16x4){0, 0, 0, 0};
                    ol[mi][n] = (u16x4){0, 0, 0, 0};
                    sv[mi][n] = (f32x4){0.f, 0.f, 0.f, 0.f};
                    if (is_res) {
                        if constexpr (PREX) {
                            oh[mi][n] = pre_h[m0 + mi][n];
                            ol[mi][n] = pre_l[m0 + mi][n];
                        } else if (!a.res_init) {
                            oh[mi][n] = *(const u16x4*)((a.R_hi ? a.R_hi : a.O_hi) + ro);
                            ol[mi][n] = *(const u16x4*)((a.R_lo ? a.R_lo : a.O_lo) + ro);
                        }
                    } else if (ch < a.C && !a.skip_init) {
                        sv[mi][n] = *(const f32x4*)(a.skip + ro);
                    }
                }
            }
#pragma unroll
            for (int mi = 0; mi < MB; ++mi) {
                const int m = m0 + mi;
                const int prow = mt * MT + wr * (MT / 2) + m * 16 + rq;
                if (prow < a.n_res) {
                    const int ch = prow;
                    u16* xhi = a.O_hi + (((size_t)b * a.oc + (ch >> 5)) * a.Lp + a.halo) * 32 + (ch & 31);
                    u16* xlo = a.O_lo + (((size_t)b * a.oc + (ch >> 5)) * a.Lp + a.halo) * 32 + (ch & 31);
#pragma unroll
                    for (int n = 0; n < NWT; ++n) {
                        const int t = t0 + wc * (NWT * 16) + n * 16 + tcol;
                        if (t >= a.L) continue;
                        u16x4 hi, lo;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float v = join_bf16(oh[mi][n][e], ol[mi][n][e]) + (acc[m][n][e] + bv[mi][e]);
                            u16 h, l;
                            split_bf16(v, h, l);
                            hi[e] = h;
                            lo[e] = l;
                        }
                        *(u16x4*)(xhi + (size_t)t * 32) = hi;
                        *(u16x4*)(xlo + (size_t)t * 32) = lo;
                    }
                } else {
                    const int ch = prow - a.n_res;
                    if (ch >= a.C) continue;
                    float* sk = a.skip + (((size_t)b * a.oc + (ch >> 5)) * a.Lp + a.halo) * 32 + (ch & 31);
#pragma unroll
                    for (int n = 0; n < NWT; ++n) {
                        const int t = t0 + wc * (NWT * 16) + n * 16 + tcol;
                        if (t >= a.L) continue;
                        *(f32x4*)(sk + (size_t)t * 32) = acc[m][n] + bv[mi] + sv[mi][n];
                    }
                }
            }
        }
    } else if (EPI == EPI_GATE_BWD) {
        // acc = d_acts[c][t] (rows = acts channel c).  With the saved a = tanh(.) * sigmoid(.) (the layer's gate output, which the
        // weight gradients need anyway) and g = sigmoid(.), t = a / g:
        //   d_pre[c]     = d_acts * g * (1 - t^2)        (tanh half)
        //   d_pre[C + c] = d_acts * a * (1 - g)          (sigmoid half: t * g * (1 - g))     -> planes with 2C channels
        // g = 0 (pre-activation below -88) has a = 0 too and both products vanish whatever t is: t := 0 there.
#pragma unroll
        for (int m = 0; m < MW; ++m) {
            const int ch = mt * MT + wr * (MT / 2) + m * 16 + rq;
            if (ch >= a.C) continue;
            const int ch2 = ch + a.C;
            const size_t tgb = (((size_t)b * a.tc + (ch >> 5)) * a.Lp + a.halo) * 32 + (ch & 31);
            const size_t o1 = (((size_t)b * a.oc + (ch >> 5)) * a.Lp + a.halo) * 32 + (ch & 31);
            const size_t o2 = (((size_t)b * a.oc + (ch2 >> 5)) * a.Lp + a.halo) * 32 + (ch2 & 31);
#pragma unroll
            for (int n = 0; n < NWT; ++n) {
                const int t = t0 + wc * (NWT * 16) + n * 16 + tcol;
                if (t >= a.L) continue;
                u16x4 th, tl, gh, gl;
                if constexpr (PREG) {
                    th = pg_th[m][n]; tl = pg_tl[m][n]; gh = pg_gh[m][n]; gl = pg_gl[m][n];
                } else {
                    th = *(const u16x4*)(a.T_hi + tgb + (size_t)t * 32);
                    tl = *(const u16x4*)(a.T_lo + tgb + (size_t)t * 32);
                    gh = *(const u16x4*)(a.G_hi + tgb + (size_t)t * 32);
                    gl = *(const u16x4*)(a.G_lo + tgb + (size_t)t * 32);
                }
                u16x4 h1, l1, h2, l2;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float av = join_bf16(th[e], tl[e]), gv = join_bf16(gh[e], gl[e]);
                    const float tv = gv != 0.0f ? av / gv : 0.0f;
                    const float da = acc[m][n][e];
                    u16 h, l;
                    split_bf16(da * gv * (1.0f - tv * tv), h, l);
                    h1[e] = h;
                    l1[e] = l;
                    split_bf16(da * av * (1.0f - gv), h, l);
                    h2[e] = h;
                    l2[e] = l;
                }
                *(u16x4*)(a.O_hi + o1 + (size_t)t * 32) = h1;
                *(u16x4*)(a.O_lo + o1 + (size_t)t * 32) = l1;
                *(u16x4*)(a.O_hi + o2 + (size_t)t * 32) = h2;
                *(u16x4*)(a.O_lo + o2 + (size_t)t * 32) = l2;
            }
        }
    } else {   // EPI_BIAS_ACT: out = act(acc + bias) -> planes (and optional f32 [B][C][L] copy)
#pragma unroll
        for (int m = 0; m < MW; ++m) {
            const int ch = mt * MT + wr * (MT / 2) + m * 16 + rq;
            if (ch >= a.C) continue;
            const f32x4 bv = *(const f32x4*)(a.bias + ch);
            u16* ohi = a.O_hi ? a.O_hi + (((size_t)bs * a.oc + (ch >> 5)) * a.Lp + a.halo) * 32 + (ch & 31) : nullptr;
            u16* olo = a.O_lo ? a.O_lo + (((size_t)bs * a.oc + (ch >> 5)) * a.Lp + a.halo) * 32 + (ch & 31) : nullptr;
#pragma unroll
            for (int n = 0; n < NWT; ++n) {
                const int t = t0 + wc * (NWT * 16) + n * 16 + tcol;
                if (t >= a.L) continue;
                u16x4 hi, lo;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float v = acc[m][n][e] + bv[e];
                    if (a.act == ACT_RELU) v = fmaxf(v, 0.f);
                    else if (a.act == ACT_TANH) v = fast_tanh(v);
                    if (a.out_f32) {
                        if (a.f32_cl) a.out_f32[((size_t)bs * a.L + t) * a.C + ch + e] = v;
                        else a.out_f32[((size_t)bs * a.C + ch + e) * a.L + t] = v;
                    }
                    u16 h, l;
                    split_bf16(v, h, l);
                    hi[e] = h;
                    lo[e] = l;
                }
                if (ohi) {
                    *(u16x4*)(ohi + (size_t)t * 32) = hi;
                    *(u16x4*)(olo + (size_t)t * 32) = lo;
                }
            }
        }
    }
}

template <int EPI, int MT, int WN = 4, bool BD = false, bool SH = false, bool EF = false, int NS = 2>
static hipError_t launch_one(const ConvGemmArgs& a, hipStream_t stream) {
    const int nwg = a.n_mtiles * a.n_ttiles * a.B;
    constexpr size_t lds = SH ? 2 * (2 * MT * 64) + 4 * 320 * 64 : NS * (2 * MT * 64 + 2 * B_PLANE_BYTES);
    static std::atomic<unsigned long long> attr_mask{0};        // per instantiation; bit d = raised on device d
    const hipError_t e = t2s_raise_lds_limit((const void*)conv_gemm_kernel<EPI, MT, WN, BD, SH, EF, NS>, (int)lds, attr_mask);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((conv_gemm_kernel<EPI, MT, WN, BD, SH, EF, NS>), dim3(nwg), dim3(128 * WN), lds, stream, a);
    return hipGetLastError();
}

// a.n_mtiles must have been computed for the same tile height `mt_rows` (256 or 128)
hipError_t t2s_launch_conv_gemm(const ConvGemmArgs& a_in, int epi, hipStream_t stream, int mt_rows) {
    static const int dbg = getenv("T2S_DBG_GEMM") ? atoi(getenv("T2S_DBG_GEMM")) : 0;
    ConvGemmArgs a = a_in;
    a.dbg = dbg;
    if (mt_rows == 128) {
        static const int ns3 = getenv("T2S_GEMM_NS3") ? atoi(getenv("T2S_GEMM_NS3")) : 0;     // 1 = three stages (measured slower)
        if (epi == EPI_RESSKIP && ns3) return launch_one<EPI_RESSKIP, 128, 4, false, false, false, 3>(a, stream);
        if (epi == EPI_GATE_BWD && ns3) return launch_one<EPI_GATE_BWD, 128, 4, false, false, false, 3>(a, stream);
        if (epi == EPI_RESSKIP) return launch_one<EPI_RESSKIP, 128>(a, stream);
        if (epi == EPI_GATE_BWD) return launch_one<EPI_GATE_BWD, 128>(a, stream);
        if (epi == EPI_GATE && a.ksplit <= 1 && a.k0 == 0 && a.kflat == 0) return launch_one<EPI_GATE, 128>(a, stream);
        // small grids (Tacotron encoder / postnet at B = 1: 2-8 workgroups walking 80 K-steps): each step is one exposed fill
        // latency with two stages; three stages of the 48 KB tile keep two fills in flight
        if (epi == EPI_BIAS_ACT && a.ksplit <= 1 && a.k0 == 0 && a.kflat == 0)
            return launch_one<EPI_BIAS_ACT, 128, 4, false, false, false, 3>(a, stream);
        return hipErrorInvalidValue;
    }
    // default gate GEMM: the ping-pong schedule (csrc/gate_gemm_pp.hip); T2S_GEMM_PP=0 falls back to the lockstep kernels below
    static const int pp = getenv("T2S_GEMM_PP") ? atoi(getenv("T2S_GEMM_PP")) : 1;
    if (epi == EPI_GATE && pp && a.ksplit <= 1 && a.k0 == 0 && a.kflat == 0 && a.nk == a.nk_x + a.sc && a.nk_x == a.taps * a.xc &&
        a.a_bstride == 0 && (a.taps >> 1) * a.dil <= a.halo)
        return t2s_launch_gate_gemm_pp(a, stream);
    static const int wn2 = getenv("T2S_GEMM_WN2") ? atoi(getenv("T2S_GEMM_WN2")) : 0;
    static const int bdir = getenv("T2S_GEMM_BD") ? atoi(getenv("T2S_GEMM_BD")) : 0;
    if (epi == EPI_GATE && wn2) return launch_one<EPI_GATE, 256, 2>(a, stream);
    if (epi == EPI_GATE && bdir) return launch_one<EPI_GATE, 256, 4, true>(a, stream);
    static const int ef = getenv("T2S_GEMM_EF") ? atoi(getenv("T2S_GEMM_EF")) : 0;
    if (epi == EPI_GATE && ef) return launch_one<EPI_GATE, 256, 4, false, false, true>(a, stream);
    static const int shb = getenv("T2S_GEMM_SH") ? atoi(getenv("T2S_GEMM_SH")) : 1;      // default on; 0 = one B tile per K-step
    if (epi == EPI_GATE && shb && a.taps == 3 && a.dil <= 32 && a.dil <= a.halo && a.nk_x == 3 * a.xc && a.ksplit <= 1 &&
        a.k0 == 0 && a.kflat == 0 && a.nk == a.nk_x + a.sc)
        return launch_one<EPI_GATE, 256, 4, false, true>(a, stream);
    if (epi == EPI_GATE) return launch_one<EPI_GATE, 256>(a, stream);
    if (epi == EPI_RESSKIP) return launch_one<EPI_RESSKIP, 256>(a, stream);
    if (epi == EPI_GATE_BWD) return launch_one<EPI_GATE_BWD, 256>(a, stream);
    return launch_one<EPI_BIAS_ACT, 256>(a, stream);
}
