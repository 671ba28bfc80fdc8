// Tacotron-2 hot path on gfx950: latency- and weight-streaming-bound pieces (exact f32, VALU).
//
// The decoder step (reference tacotron.py:355-393) is 71 MB of LSTM weights per step at <= 16 flop/B:
// it is bound by streaming those weights out of L2 / Infinity Cache, not by math.  Design:
//   * the 4096 gate rows of each LSTMCell are spread over 256 workgroups (4 hidden units each), one
//     wave per hidden unit holding its four gate rows in registers and looping over the batch, so every
//     weight byte is read once per step per chip; the cell update (sigmoid/tanh, c, h) is fused, so one
//     launch per cell and no inter-workgroup hand-off inside a launch;
//   * dependent stages are separate launches (a kernel boundary, ~1.5 us, is cheaper than a grid
//     barrier on this chip - MI355X_MICROARCH.md price list), enqueued back-to-back by a C++ loop
//     (t2s_taco_decode_steps) so the host never sits on the critical path;
//   * softmax / dot-product reductions use wavefront shuffles.
#include "t2s_common.h"
#include "t2s_kernels.h"
#include "tacotron_ops.h"

#include <stdlib.h>

static __device__ __forceinline__ float sigmoid_acc(float x) { return 1.0f / (1.0f + expf(-x)); }

// ------------------------------------------------------------------------------------------------
// Input vector of a GEMV = concatenation of up to three segments (e.g. [prenet_out | context | h]).
// per-lane float4 slot v covers k = (v*64 + lane)*4 .. +3 of the concatenated vector.
//
// The segment pointers / sizes / strides are passed as plain scalars (they stay in SGPRs) and chosen with selects.  They must
// NOT sit in a struct or array that is indexed or address-selected at run time: hipcc then keeps that object in memory,
// promotes it to LDS, and addresses the promoted copy through the flat work-item id, which it derives from the AQL dispatch
// packet - a scalar load from HOST memory over PCIe in the middle of the kernel (measured at B = 1: 5-11 us of the 12.9 us
// the 337 x 1536 projection GEMV took; .amdhsa_user_sgpr_dispatch_ptr in the kernel descriptor is the tell-tale).
template <int NV4>
struct LaneMap {
    const float* xp[NV4];
    long xs[NV4];
    bool valid[NV4];
    __device__ __forceinline__ void init(const float* p0, const float* p1, const float* p2, int n0, int n1, int n2, long s0,
                                         long s1, long s2, int lane) {
        const int K = n0 + n1 + n2;
#pragma unroll
        for (int v = 0; v < NV4; ++v) {
            const int k = (v * 64 + lane) * 4;
            valid[v] = k < K;
            const bool in0 = k < n0, in1 = !in0 && (k - n0) < n1;
            const float* base = in0 ? p0 : (in1 ? p1 : p2);
            const long stride = in0 ? s0 : (in1 ? s1 : s2);
            const int kk = in0 ? k : (in1 ? k - n0 : k - n0 - n1);
            xp[v] = valid[v] ? base + kk : p0;
            xs[v] = valid[v] ? stride : s0;
        }
    }
};

// weight row = [W1 row (k1 floats) | W2 row (k2 floats)]
template <int NV4>
static __device__ __forceinline__ void load_row(f32x4 (&w)[NV4], const float* W1, int ld1, int k1, const float* W2,
                                                int ld2, int k2, int row, int lane) {
    // Every load is issued unconditionally from an in-range address and masked afterwards: a load under a per-slot runtime
    // condition makes hipcc branch around it and wait vmcnt(0) per slot - NV4 dependent memory round trips instead of one
    // (cdna_hip_programming.md section 5, ".s-level traps" (c); measured here: 12.6 us -> a few us for the 337 x 1536 GEMV).
    const float* p[NV4];
    bool ok[NV4];
#pragma unroll
    for (int v = 0; v < NV4; ++v) {
        const int k = (v * 64 + lane) * 4;
        ok[v] = k < k1 + k2;
        const bool in2 = k >= k1 && k2 > 0;
        p[v] = in2 ? W2 + (size_t)row * ld2 + (ok[v] ? k - k1 : 0) : W1 + (size_t)row * ld1 + (k < k1 ? k : 0);
    }
#pragma unroll
    for (int v = 0; v < NV4; ++v) w[v] = *(const f32x4*)p[v];
#pragma unroll
    for (int v = 0; v < NV4; ++v)
        if (!ok[v]) w[v] = (f32x4){0.f, 0.f, 0.f, 0.f};
}

// ------------------------------------------------------------------------------------------------
// y[item][row] = act(bias + W[row] . x[item]) * mask   — one wave per output row, weights in registers.
// Restates every small Linear on the decoder path: query_layer (tacotron.py:137), linear_projection +
// gate_layer (:387-392), Prenet layers (modules.py:19-22), memory_layer (tacotron.py:306).
#ifdef T2S_CLOCK_PROBE
// Diagnostic build only: (kernel id, s_memtime, s_memrealtime) at entry and exit of workgroup 0 of the decoder-chain kernels,
// in a ring nothing else reads (tools/decode_probe.py).  Tells the in-kernel clock and the body time of each launch.
__device__ unsigned long long t2s_probe_buf[1 << 16][12];
__device__ unsigned int t2s_probe_idx;
extern "C" int t2s_debug_read_probe(unsigned long long* host_out, unsigned int* n) {
    hipError_t e = hipMemcpyFromSymbol(n, HIP_SYMBOL(t2s_probe_idx), sizeof(unsigned int));
    if (e != hipSuccess) return (int)e;
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(t2s_probe_buf), sizeof(unsigned long long) * 12 * (1 << 16));
}
#define PROBE_BEGIN(ID)                                                                        \
    unsigned int probe_i = 0;                                                                  \
    if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) {                              \
        probe_i = atomicAdd(&t2s_probe_idx, 1u) & 0xffff;                                      \
        t2s_probe_buf[probe_i][0] = (ID);                                                      \
        t2s_probe_buf[probe_i][1] = __builtin_amdgcn_s_memtime();                              \
        t2s_probe_buf[probe_i][2] = __builtin_amdgcn_s_memrealtime();                          \
    }
#define PROBE_END()                                                                            \
    if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) {                              \
        t2s_probe_buf[probe_i][3] = __builtin_amdgcn_s_memtime();                              \
        t2s_probe_buf[probe_i][4] = __builtin_amdgcn_s_memrealtime();                          \
    }
#define PROBE_MID(J)                                                                           \
    if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) {                              \
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");                            \
        t2s_probe_buf[probe_i][5 + (J)] = __builtin_amdgcn_s_memrealtime();                    \
    }
#else
#define PROBE_BEGIN(ID)
#define PROBE_END()
#define PROBE_MID(J)
#endif

template <int NV4>
static __device__ __forceinline__ void gemv_rows_body(const GemvArgs& a) {
    PROBE_BEGIN(100 + NV4)
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= a.rows) return;          // (never workgroup 0 / thread 0: the probe above always reaches its end)
    f32x4 w[NV4];
    load_row<NV4>(w, a.W1, a.ld1, a.k1, a.W2, a.ld2, a.k2, row, lane);
    PROBE_MID(0)
    LaneMap<NV4> lm;
    lm.init(a.x1, a.x2, a.x3, a.n1, a.n2, a.n3, a.sx1, a.sx2, a.sx3, lane);
    PROBE_MID(3)
    const float bias = (a.bias1 ? a.bias1[row] : 0.f) + (a.bias2 ? a.bias2[row] : 0.f);
    PROBE_MID(4)
    for (int it = blockIdx.y; it < a.items; it += gridDim.y) {
        float acc = 0.f;
        f32x4 x[NV4];
#pragma unroll
        for (int v = 0; v < NV4; ++v) x[v] = *(const f32x4*)(lm.xp[v] + (size_t)it * lm.xs[v]);    // all in flight together
        PROBE_MID(1)
#pragma unroll
        for (int v = 0; v < NV4; ++v) {
            if (!lm.valid[v]) x[v] = (f32x4){0.f, 0.f, 0.f, 0.f};        // a select, not a branch: the weights there are 0 too
            acc += w[v][0] * x[v][0] + w[v][1] * x[v][1] + w[v][2] * x[v][2] + w[v][3] * x[v][3];
        }
        acc = wave_sum(acc);
        PROBE_MID(2)
        if (lane == 0) {
            float y = acc + bias;
            if (a.split_row > 0 && row >= a.split_row) {
                const int r2 = row - a.split_row;
                if (a.act2 == ACT_RELU) y = fmaxf(y, 0.f);
                else if (a.act2 == ACT_TANH) y = tanhf(y);
                if (a.mask2) y *= a.mask2[(size_t)it * a.smask2_item + r2] ? a.mask2_scale : 0.f;
                a.y2[(size_t)it * a.sy2_item + (size_t)r2 * a.sy2_row] = y;
            } else {
                if (a.act == ACT_RELU) y = fmaxf(y, 0.f);
                else if (a.act == ACT_TANH) y = tanhf(y);
                if (a.mask) y *= a.mask[(size_t)it * a.smask_item + row] ? a.mask_scale : 0.f;
                a.y[(size_t)it * a.sy_item + (size_t)row * a.sy_row] = y;
            }
        }
    }
    PROBE_END()
}

template <int NV4>
__global__ __launch_bounds__(256) void gemv_rows_kernel(const GemvArgs a) {
    gemv_rows_body<NV4>(a);
}

// Location term of the NEXT decoder step (LocPreArgs, tacotron_ops.h): P_loc[b][t][a] = sum_f D[a][f] * sum_{c,j} K[f][c][j] cat[c][t + j - pad]
// (tacotron.py:96-107 location_conv + location_dense) for 16 positions of one batch element per workgroup, on the exact-f32 matrix
// cores exactly as the fused attention kernel computes it (same operands, same MFMA order per output): 4 waves, stage 1 = two
// 16 x 16 feature tiles on waves 0 / 1, stage 2 = eight 16 x 16 channel tiles, two per wave, D fragments straight from global.
static __device__ __forceinline__ void loc_pre_role(const LocPreArgs& p, int rb) {
    __shared__ float s_cat[2][16 + 64];
    __shared__ float s_kb[64 * 48];
    __shared__ float s_f[16 * 33];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lr = lane & 15, lq = lane >> 4;
    const int T = p.T, KS = p.loc_ks, pad = KS >> 1, K2 = 2 * KS, KP = (K2 + 3) & ~3;
    const int ntt = (T + 15) >> 4;
    const int b = rb / ntt, t0 = 16 * (rb - b * ntt);
    const int W = 16 + KS - 1;
    constexpr int AD = 128, F = 32;
    // every load first
    float rk[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) {                       // F * K2 <= 32 * 126 = 4032
        const int i = tid + j * 256;
        rk[j] = p.w_loc_conv[i < F * K2 ? i : 0];
    }
    float rc = 0.f;
    {
        const int c = tid >= W ? 1 : 0, j = tid - c * W;
        const int t = t0 + j - pad;
        const bool in = tid < 2 * W && t >= 0 && t < T;
        rc = (c ? p.w_cum : p.w)[(size_t)b * T + (in ? t : 0)];
        if (!in) rc = 0.f;
    }
    float bd[2][8];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int u = 0; u < 8; ++u) bd[i][u] = p.w_loc_denseT[(4 * u + lq) * AD + 16 * (wave + 4 * i) + lr];   // B[k = f][col a] = D[a][f]
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const int i = tid + j * 256;
        if (i < F * K2) {
            const int f = i / K2, k = i - f * K2;        // K[f][c][j] -> B operand [k = c * KS + j][f]
            s_kb[k * 48 + f] = rk[j];
        }
    }
    if (tid < (KP - K2) * F) s_kb[(K2 + tid / F) * 48 + (tid % F)] = 0.f;      // zero rows that pad K to a multiple of 4
    if (tid < 2 * W) s_cat[tid >= W ? 1 : 0][tid >= W ? tid - W : tid] = rc;
    __syncthreads();
    if (wave < 2) {
        const int ft = wave;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        float av[16], bv[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int k = 4 * u + lq;
            const int kc = k < K2 ? k : 0;               // (B rows >= 2 KS are zero)
            const int c = kc >= KS ? 1 : 0, j = kc - c * KS;
            av[u] = s_cat[c][lr + j];
            bv[u] = s_kb[(k < KP ? k : 0) * 48 + 16 * ft + lr];
        }
#pragma unroll
        for (int u = 0; u < 16; ++u)
            if (4 * u < KP) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u], bv[u], acc, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 4; ++r) s_f[(4 * lq + r) * 33 + 16 * ft + lr] = acc[r];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int at = wave + 4 * i;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        float av[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) av[u] = s_f[lr * 33 + 4 * u + lq];
#pragma unroll
        for (int u = 0; u < 8; ++u) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u], bd[i][u], acc, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int t = t0 + 4 * lq + r;
            if (t < T) p.ploc[((size_t)b * T + t) * AD + 16 * at + lr] = acc[r];
        }
    }
}

template <int NV4>
__global__ __launch_bounds__(256) void gemv_rows_loc_kernel(const GemvArgs a, const LocPreArgs lp) {
    if ((int)blockIdx.x >= lp.n_gemv_blocks) {           // whole workgroups: no barrier of either role is skipped
        loc_pre_role(lp, (int)blockIdx.x - lp.n_gemv_blocks);
        return;
    }
    gemv_rows_body<NV4>(a);
}

hipError_t t2s_launch_gemv(const GemvArgs& a, hipStream_t stream) {
    if (t2s_sbgemm_plain_ok(a)) return t2s_launch_sbgemm_plain(a, stream);      // 9+ items: f32 matrix cores
    const int K = a.n1 + a.n2 + a.n3;
    const int nv4 = (K + 255) / 256;
    dim3 grid((a.rows + 3) / 4, a.items < 64 ? 1 : (a.items < 4096 ? 16 : 64));
#define GL(N) hipLaunchKernelGGL(gemv_rows_kernel<N>, grid, dim3(256), 0, stream, a)
    if (nv4 <= 1) GL(1);
    else if (nv4 <= 2) GL(2);
    else if (nv4 <= 4) GL(4);
    else if (nv4 <= 7) GL(7);
    else if (nv4 <= 10) GL(10);
    else if (nv4 <= 16) GL(16);
    else return hipErrorInvalidValue;
#undef GL
    return hipGetLastError();
}

// the same GEMV (wave-per-row form, few items) with the location-term role riding on the launch
hipError_t t2s_launch_gemv_with_loc(const GemvArgs& a, const LocPreArgs& lp_in, hipStream_t stream) {
    const int K = a.n1 + a.n2 + a.n3;
    const int nv4 = (K + 255) / 256;
    if (a.items >= 64 || !lp_in.ploc || !lp_in.w || !lp_in.w_cum || !lp_in.w_loc_conv || !lp_in.w_loc_denseT || lp_in.B <= 0 ||
        lp_in.T <= 0 || lp_in.loc_ks > 31 || !(lp_in.loc_ks & 1))      // (one 64-wide K pass: 2 * KS <= 64)
        return hipErrorInvalidValue;
    LocPreArgs lp = lp_in;
    lp.n_gemv_blocks = (a.rows + 3) / 4;
    dim3 grid(lp.n_gemv_blocks + lp.B * ((lp.T + 15) / 16), 1);
#define GL(N) hipLaunchKernelGGL(gemv_rows_loc_kernel<N>, grid, dim3(256), 0, stream, a, lp)
    if (nv4 <= 1) GL(1);
    else if (nv4 <= 2) GL(2);
    else if (nv4 <= 4) GL(4);
    else if (nv4 <= 7) GL(7);
    else if (nv4 <= 10) GL(10);
    else if (nv4 <= 16) GL(16);
    else return hipErrorInvalidValue;
#undef GL
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// Fused LSTMCell (torch gate order i,f,g,o; reference tacotron.py:366-370,380-385):
//   gates = W_ih [x1|x2] + b_ih + W_hh h + b_hh ;  c' = s(f) c + s(i) tanh(g) ;  h' = s(o) tanh(c')
// One wave per hidden unit u keeps rows {u, H+u, 2H+u, 3H+u} of [W_ih | W_hh] in registers and loops
// over the batch; lane (b mod 64) then does the pointwise update for item b.  h is ping-ponged by the
// caller (h_in read by every workgroup, h_out written by the owner), c is updated in place.
// Work split: a workgroup owns 2 hidden units; each unit's K range is split over 4 waves (8 waves per workgroup,
// two workgroups per CU: 16 waves per CU streaming weights - the cell is latency-bound at B=1 - while the
// 512-thread block keeps a 256-VGPR budget, so nothing spills).
// Wave (unit, kq) keeps float4 slots v = kq, kq+4, ... of its four gate rows in registers; partial sums
// meet in LDS and thread (unit, item) applies the cell update.
// UNITS = hidden units per workgroup (4 -> 1024 threads, 2 -> 512 threads); SAVE = keep gates / cell state (training)
// weight-row loads of the LSTM cell: each row is streamed once per step by ONE workgroup, so they are non-temporal
// (MI355X_MICROARCH.md nt-weights; same box, alternating: 39.26 / 39.01 vs 39.60 / 39.66 us per step at B = 1).
// -DT2S_LSTM_PLAIN_LOADS restores default-policy loads.
#ifdef T2S_LSTM_PLAIN_LOADS
#define T2S_WLOAD(p) (*(p))
#else
#define T2S_WLOAD(p) __builtin_nontemporal_load(p)
#endif
// With the streamed gate partials (LstmCellArgs::pre_a) a cell reads 8-13 MB per step, and those columns are better loaded with the
// default policy while the attention launch's 50 MB stream stays nt: 29.04 / 29.04 vs 29.70 / 29.33 us per step at B = 1, same box
// (profiles/r04_cache_policy_ab.txt; all-plain: 31.4 / 31.3; the unstreamed chain prefers nt in the cells: 37.5-37.7 vs 38.3-38.7).
// -DT2S_CELL_NT_LOADS: nt in the streamed cells too (the A/B's other side).
#ifdef T2S_CELL_NT_LOADS
#define T2S_CLOAD(p) T2S_WLOAD(p)
#else
#define T2S_CLOAD(p) (*(p))
#endif

template <int NVW, int UNITS, bool SAVE, bool STREAMED = false>
__global__ __launch_bounds__(UNITS * 256) void lstm_cell_kernel(const LstmCellArgs a) {
    PROBE_BEGIN(200 + NVW)
    __shared__ float s_part[UNITS][4][4][64];        // [unit][kq][gate][item]
    __shared__ float s_h[UNITS][64];                 // new h of this workgroup's units (for the partial attention query)
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int ul = wave >> 2, kq = wave & 3;
    const int u = blockIdx.x * UNITS + ul;           // H % UNITS == 0 is checked by the caller
    const int K1 = a.n1 + a.n2, K = K1 + (a.h_in ? a.H : 0);
    const int ldi = a.ld_ih > 0 ? a.ld_ih : K1;
    // Everything the tail of the kernel needs from memory is requested now, next to the weight rows, so the pointwise update
    // and the partial query do not add dependent round trips behind the reduction: the gate biases, the cell state of the
    // first item chunk, and this workgroup's UNITS columns of W_query.
    __shared__ float s_bsum[UNITS][4];
    __shared__ float s_wq[UNITS][128];
    float c_pre = 0.f;
    // streamed-gates form: partial pre-activations of the first item chunk.  Loaded raw (clamped addresses, no arithmetic here:
    // an add at this point makes hipcc wait for the loads in front of the weight-row loads - a whole memory round trip at the
    // head of the kernel); summed where they are used.
    float g_pa[4] = {0.f, 0.f, 0.f, 0.f}, g_pb[4] = {0.f, 0.f, 0.f, 0.f};
    {
        const bool mine = kq == 0 && lane < a.B;
        const int itc = mine ? lane : 0;
        c_pre = a.c[(size_t)itc * a.H + u];
        const float* pa = a.pre_a ? a.pre_a : a.c;          // (always a readable address; unused values are dropped below)
        const float* pb = a.pre_b ? a.pre_b : pa;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const size_t i = a.pre_a ? (size_t)itc * 4 * a.H + (size_t)g * a.H + u : (size_t)u;
            g_pa[g] = pa[i];
            g_pb[g] = pb[i];
        }
    }
    f32x4 w[4][NVW];
    const float* xp[NVW];
    long xs[NVW];
    bool valid[NVW];
#pragma unroll
    for (int j = 0; j < NVW; ++j) {
        const int v = kq + 4 * j;
        int k = (v * 64 + lane) * 4;
        valid[j] = k < K;
        // unconditional loads from clamped addresses, masked afterwards (no branch + vmcnt(0) per slot: see load_row)
        const int kc = valid[j] ? k : 0;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const size_t row = (size_t)g * a.H + u;
            const float* wp = kc < K1 ? a.W_ih + row * ldi + kc : a.W_hh + row * a.H + (kc - K1);
            w[g][j] = STREAMED ? T2S_CLOAD((const f32x4*)wp) : T2S_WLOAD((const f32x4*)wp);
        }
#pragma unroll
        for (int g = 0; g < 4; ++g)
            if (!valid[j]) w[g][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (!valid[j]) { xp[j] = a.c; xs[j] = 0; }          // (a readable dummy: x1 is NULL in the folded-prenet form)
        else if (k < a.n1) { xp[j] = a.x1 + k; xs[j] = a.sx1; }
        else if (k < K1) { xp[j] = a.x2 + (k - a.n1); xs[j] = a.sx2; }
        else { xp[j] = a.h_in + (k - K1); xs[j] = a.H; }
    }
    // (behind the weight-row loads in issue order: their wait covers these too)
    if (kq == 0 && lane < 4) s_bsum[ul][lane] = a.b_ih[lane * a.H + u] + a.b_hh[lane * a.H + u];
    if (a.q_part && (int)threadIdx.x < a.q_dim && a.q_dim <= 128) {
#pragma unroll
        for (int i = 0; i < UNITS; ++i) s_wq[i][threadIdx.x] = a.w_q[(size_t)threadIdx.x * a.H + blockIdx.x * UNITS + i];
    }
    for (int b0 = 0; b0 < a.B; b0 += 64) {
        const int bn = min(64, a.B - b0);
        for (int bb = 0; bb < bn; ++bb) {
            const int it = b0 + bb;
            float acc[4] = {0.f, 0.f, 0.f, 0.f};
            f32x4 xv[NVW];
#pragma unroll
            for (int j = 0; j < NVW; ++j) xv[j] = *(const f32x4*)(xp[j] + (size_t)it * xs[j]);
#pragma unroll
            for (int j = 0; j < NVW; ++j) {
                const f32x4 x = valid[j] ? xv[j] : (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    acc[g] += w[g][j][0] * x[0] + w[g][j][1] * x[1] + w[g][j][2] * x[2] + w[g][j][3] * x[3];
            }
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float sum = wave_sum(acc[g]);
                if (lane == 0) s_part[ul][kq][g][bb] = sum;
            }
        }
        __syncthreads();
        if (kq == 0 && lane < bn) {
            const int it = b0 + lane;
            float gsum[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                gsum[g] = (s_part[ul][0][g][lane] + s_part[ul][1][g][lane]) + (s_part[ul][2][g][lane] + s_part[ul][3][g][lane]) +
                          s_bsum[ul][g];
                if (a.pre_a) {
                    const size_t i = (size_t)it * 4 * a.H + (size_t)g * a.H + u;
                    gsum[g] += b0 == 0 ? g_pa[g] + (a.pre_b ? g_pb[g] : 0.f) : a.pre_a[i] + (a.pre_b ? a.pre_b[i] : 0.f);
                }
            }
            const size_t idx = (size_t)it * a.H + u;
            const float c = b0 == 0 ? c_pre : a.c[idx];
            const float c2 = sigmoid_acc(gsum[1]) * c + sigmoid_acc(gsum[0]) * tanhf(gsum[2]);
            float h2 = sigmoid_acc(gsum[3]) * tanhf(c2);
            a.c[idx] = c2;
            if constexpr (SAVE) {      // training: post-activation gates and the new cell state, for the backward pass
                float* go = a.gates_out + (size_t)it * 4 * a.H + u;
                go[0] = sigmoid_acc(gsum[0]); go[a.H] = sigmoid_acc(gsum[1]); go[2 * a.H] = tanhf(gsum[2]);
                go[3 * a.H] = sigmoid_acc(gsum[3]);
                a.c_out[idx] = c2;
            }
            if (a.drop_mask) h2 = a.drop_mask[idx] ? h2 * a.drop_scale : 0.f;
            a.h_out[idx] = h2;
            if (a.h_copy) a.h_copy[(size_t)it * a.s_copy + u] = h2;
            s_h[ul][lane] = h2;
        }
        __syncthreads();
        // partial attention query of this workgroup's units (tacotron.py:137 query_layer, summed over workgroups by
        // att_fused_kernel): q_part[wg][item][a] = sum_u W_query[a][u] * h[item][u]
        if (a.q_part && (int)threadIdx.x < a.q_dim && a.q_dim <= 128) {
            for (int bb = 0; bb < bn; ++bb) {
                float q = 0.f;
#pragma unroll
                for (int i = 0; i < UNITS; ++i) q += s_wq[i][threadIdx.x] * s_h[i][bb];
                a.q_part[((size_t)blockIdx.x * a.B + b0 + bb) * a.q_dim + threadIdx.x] = q;
            }
        }
    }
    PROBE_END()
}

// ------------------------------------------------------------------------------------------------
// Attention LSTMCell of the streamed-gates decode (B <= 8, autoregressive) with the prenet's second layer folded in:
//   x1 = relu(W_pre2 . pre1) * mask * 2      (modules.py:19-22: Linear + ReLU + the always-on dropout), recomputed by EVERY workgroup
//   gates = W_ih [x1 | ctx] + b_ih + b_hh + pre_a            (pre_a = W_hh . h from the previous attention launch's gate-stream role)
// so the prenet GEMV leaves the serial chain (4.4 us per step as a launch of its own).  K = 256 + 512 is exactly three
// 256-float slots, so the workgroup is 12 waves = (4 hidden units) x (3 K slots), 768 threads.
// The folded layer is a SPARSE product: pre1 went through ReLU and a keep-half dropout, so about three quarters of it are exact
// zeros.  With W_pre2 transposed ([k][r]) a nonzero pre1[k] costs one coalesced 1 KB row and no cross-lane reduction at all
// (lane l accumulates outputs 4l .. 4l+3); a wave ballots its 64-entry segment of pre1 and walks the set bits with scalar
// code.  ~64 KB per workgroup out of L2 instead of the dense 256 KB (a CU draws 64 B/clk: 1.9 us for the dense form, measured
// 10.0 us per launch against 6.4 for the plain cell).  Exact: skipped terms are products with 0.0f.
__global__ __launch_bounds__(768) void lstm_cell_p2_kernel(const LstmCellArgs a) {
    PROBE_BEGIN(210)
    __shared__ float s_part[4][3][4][8];             // [unit][kq][gate][item]
    __shared__ float s_h[4][8];
    __shared__ float s_bsum[4][4];
    __shared__ float s_wq[4][128];
    __shared__ __attribute__((aligned(16))) float s_p2[8 * 256];
    __shared__ __attribute__((aligned(16))) float s_red[12][256];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ul = wave / 3, kq = wave - 3 * ul;
    const int u = blockIdx.x * 4 + ul;
    const int ldi = a.ld_ih > 0 ? a.ld_ih : 768;
    // LSTM rows (HBM, non-temporal: read once per step by this wave)
    f32x4 w[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) w[g] = T2S_CLOAD((const f32x4*)(a.W_ih + ((size_t)g * a.H + u) * ldi + kq * 256 + 4 * lane));
    // prenet: this wave's 64-entry segment of pre1 (three waves share a segment and take its entries k = sub, sub + 3, ...)
    const int seg = wave & 3, sub = wave >> 2;
    float p1 = a.p1[seg * 64 + lane];
    unsigned char mkb = tid < 256 ? a.p2_mask[tid] : (unsigned char)0;
    // tail operands: cell state, streamed partials, biases, this workgroup's columns of W_query (raw loads, no arithmetic here)
    const bool mine = kq == 0 && lane < a.B;
    const int itc = mine ? lane : 0;
    const float c_pre = a.c[(size_t)itc * a.H + u];
    float g_pa[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) g_pa[g] = a.pre_a[(size_t)itc * 4 * a.H + (size_t)g * a.H + u];
    float bi = 0.f, bh = 0.f;
    if (kq == 0 && lane < 4) { bi = a.b_ih[lane * a.H + u]; bh = a.b_hh[lane * a.H + u]; }
    float wq[4] = {0.f, 0.f, 0.f, 0.f};
    const bool do_q = a.q_part && tid < a.q_dim && a.q_dim <= 128;
    if (do_q) {
#pragma unroll
        for (int i = 0; i < 4; ++i) wq[i] = a.w_q[(size_t)tid * a.H + blockIdx.x * 4 + i];
    }
    // positions of a segment that belong to this wave: k = sub (mod 3)
    const unsigned long long third = 0x9249249249249249ull << sub;
    for (int b = 0; b < a.B; ++b) {
        unsigned long long mm = __ballot(p1 != 0.f) & third;
        const float* wt = a.w_p2 + (size_t)seg * 64 * 256 + 4 * lane;           // W_pre2^T rows of this segment
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        while (mm) {                                                             // (wave-uniform: scalar loop)
            f32x4 wv[8];
            float pv[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const bool has = mm != 0;
                const int k = has ? (int)__builtin_ctzll(mm) : 0;
                mm = has ? mm & (mm - 1) : 0;
                const float pk = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, p1), k));
                pv[j] = has ? pk : 0.f;
                wv[j] = *(const f32x4*)(wt + (size_t)k * 256);
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) acc += wv[j] * pv[j];
        }
        *(f32x4*)(&s_red[wave][4 * lane]) = acc;
        // the next item's operands are requested before this one's partials are summed
        const unsigned char mk_now = mkb;
        if (b + 1 < a.B) {
            p1 = a.p1[(size_t)(b + 1) * 256 + seg * 64 + lane];
            mkb = tid < 256 ? a.p2_mask[(size_t)(b + 1) * a.s_p2_mask + tid] : (unsigned char)0;
        }
        __syncthreads();
        if (tid < 256) {
            float y = 0.f;
#pragma unroll
            for (int i = 0; i < 12; ++i) y += s_red[i][tid];                    // fixed order
            s_p2[b * 256 + tid] = mk_now ? fmaxf(y, 0.f) * a.p2_scale : 0.f;
        }
        __syncthreads();
    }
    if (kq == 0 && lane < 4) s_bsum[ul][lane] = bi + bh;
    if (do_q) {
#pragma unroll
        for (int i = 0; i < 4; ++i) s_wq[i][tid] = wq[i];
    }
    // ---- gate pre-activations: slot 0 = pre2 (LDS), slots 1, 2 = the context ----
    for (int b = 0; b < a.B; ++b) {
        const f32x4 x = kq == 0 ? *(const f32x4*)(s_p2 + b * 256 + 4 * lane)
                                : *(const f32x4*)(a.x2 + (size_t)b * a.sx2 + (kq - 1) * 256 + 4 * lane);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            float acc = w[g][0] * x[0] + w[g][1] * x[1] + w[g][2] * x[2] + w[g][3] * x[3];
            acc = wave_sum(acc);
            if (lane == 0) s_part[ul][kq][g][b] = acc;
        }
    }
    __syncthreads();
    if (mine) {
        const int it = lane;
        float gsum[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) gsum[g] = (s_part[ul][0][g][it] + s_part[ul][1][g][it]) + s_part[ul][2][g][it] + s_bsum[ul][g] + g_pa[g];
        const size_t idx = (size_t)it * a.H + u;
        const float c2 = sigmoid_acc(gsum[1]) * c_pre + sigmoid_acc(gsum[0]) * tanhf(gsum[2]);
        float h2 = sigmoid_acc(gsum[3]) * tanhf(c2);
        a.c[idx] = c2;
        if (a.drop_mask) h2 = a.drop_mask[idx] ? h2 * a.drop_scale : 0.f;
        a.h_out[idx] = h2;
        if (a.h_copy) a.h_copy[(size_t)it * a.s_copy + u] = h2;
        s_h[ul][it] = h2;
    }
    __syncthreads();
    if (do_q) {
        for (int b = 0; b < a.B; ++b) {
            float q = 0.f;
#pragma unroll
            for (int i = 0; i < 4; ++i) q += s_wq[i][tid] * s_h[i][b];
            a.q_part[((size_t)blockIdx.x * a.B + b) * a.q_dim + tid] = q;
        }
    }
    PROBE_END()
}

hipError_t t2s_launch_lstm_cell(const LstmCellArgs& a, hipStream_t stream) {
    if (t2s_sbgemm_lstm_ok(a)) return t2s_launch_sbgemm_lstm(a, stream);        // 9+ items: f32 matrix cores
    const int K = a.n1 + a.n2 + (a.h_in ? a.H : 0);
    const int nv4 = (K + 255) / 256;
    const int nvw = (nv4 + 3) / 4;
    if (a.H % 4 || (!a.h_in && !a.pre_a)) return hipErrorInvalidValue;
    // eval / inference: 4 units per 1024-thread workgroup (128-VGPR cap, fits without the save code);
    // training saves: 2 units per 512-thread workgroup (256-VGPR budget)
#define LL(N)                                                                                                     \
    do {                                                                                                          \
        if (a.gates_out) hipLaunchKernelGGL((lstm_cell_kernel<N, 2, true>), dim3(a.H / 2), dim3(512), 0, stream, a); \
        else if (a.pre_a) hipLaunchKernelGGL((lstm_cell_kernel<N, 4, false, true>), dim3(a.H / 4), dim3(1024), 0, stream, a); \
        else hipLaunchKernelGGL((lstm_cell_kernel<N, 4, false>), dim3(a.H / 4), dim3(1024), 0, stream, a);          \
    } while (0)
    if (a.w_p2) {           // folded prenet layer 1 (lstm_cell_p2_kernel): K = 256 (pre2) + 512 (context), W_hh . h streamed earlier
        if (a.gates_out || a.c_out || a.n1 != 256 || a.n2 != 512 || a.h_in || !a.pre_a || a.pre_b || a.B > 8 || !a.p1 ||
            !a.p2_mask || !a.x2 || (a.sx2 & 3))
            return hipErrorInvalidValue;
        hipLaunchKernelGGL(lstm_cell_p2_kernel, dim3(a.H / 4), dim3(768), 0, stream, a);
        return hipGetLastError();
    }
    if (nvw <= 1) LL(1);
    else if (nvw <= 2) LL(2);
    else if (nvw <= 3) LL(3);
    else return hipErrorInvalidValue;
#undef LL
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// Location-sensitive attention energies (reference tacotron.py:96-107,124-143):
//   f[t][:]  = conv1d([w ; w_cum], K[32][2][31], pad 15)[:, t]
//   e[t]     = v . tanh(q + D f[t] + processed_memory[t])          (masked to -inf beyond the length)
// One workgroup = 16 time steps of one batch element; a wave handles 4 of them with attention_dim on lanes.
#define ATT_TQ 16
__global__ __launch_bounds__(256) void att_energy_kernel(const AttArgs a) {
    __shared__ float s_cat[2][ATT_TQ + 64];
    __shared__ float s_f[ATT_TQ][33];
    extern __shared__ float s_k[];                       // loc conv weights [F][2][KS]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.y;
    const int t0 = blockIdx.x * ATT_TQ;
    const int KS = a.loc_ks, F = a.loc_f, pad = KS >> 1;
    for (int i = tid; i < F * 2 * KS; i += 256) s_k[i] = a.w_loc_conv[i];
    for (int i = tid; i < 2 * (ATT_TQ + KS - 1); i += 256) {
        const int c = i / (ATT_TQ + KS - 1), j = i - c * (ATT_TQ + KS - 1);
        const int t = t0 + j - pad;
        const float* src = c ? a.w_cum : a.w_prev;
        s_cat[c][j] = (t >= 0 && t < a.T) ? src[(size_t)b * a.T + t] : 0.f;
    }
    __syncthreads();
    for (int i = tid; i < ATT_TQ * F; i += 256) {
        const int tq = i / F, f = i - tq * F;
        float acc = 0.f;
        for (int c = 0; c < 2; ++c)
#pragma unroll 8
            for (int j = 0; j < KS; ++j) acc += s_k[(f * 2 + c) * KS + j] * s_cat[c][tq + j];
        s_f[tq][f] = acc;
    }
    __syncthreads();
    // attention_dim (<= 128) on lanes: a0 = lane, a1 = lane + 64
    const int AD = a.att_dim;
    float d0[32], d1[32];
#pragma unroll
    for (int f = 0; f < 32; ++f) {
        // the transposed copy [F][att_dim] reads coalesced (lanes along att_dim); the row-major weight costs one
        // cache line per lane per load
        if (a.w_loc_denseT) {
            d0[f] = (lane < AD && f < F) ? a.w_loc_denseT[f * AD + lane] : 0.f;
            d1[f] = (lane + 64 < AD && f < F) ? a.w_loc_denseT[f * AD + lane + 64] : 0.f;
        } else {
            d0[f] = (lane < AD && f < F) ? a.w_loc_dense[lane * F + f] : 0.f;
            d1[f] = (lane + 64 < AD && f < F) ? a.w_loc_dense[(lane + 64) * F + f] : 0.f;
        }
    }
    const float q0 = lane < AD ? a.q[(size_t)b * AD + lane] : 0.f;
    const float q1 = lane + 64 < AD ? a.q[(size_t)b * AD + lane + 64] : 0.f;
    const float v0 = lane < AD ? a.w_v[lane] : 0.f;
    const float v1 = lane + 64 < AD ? a.w_v[lane + 64] : 0.f;
    const int len = a.lengths ? a.lengths[b] : a.T;
    float pm0[ATT_TQ / 4], pm1[ATT_TQ / 4];              // processed-memory rows of this wave's steps, fetched up front
#pragma unroll
    for (int r = 0; r < ATT_TQ / 4; ++r) {
        const int t = t0 + wave + 4 * r;
        const float* pm = a.pmem + ((size_t)b * a.T + t) * AD;
        pm0[r] = (t < a.T && lane < AD) ? pm[lane] : 0.f;
        pm1[r] = (t < a.T && lane + 64 < AD) ? pm[lane + 64] : 0.f;
    }
#pragma unroll
    for (int r = 0; r < ATT_TQ / 4; ++r) {
        const int tq = wave + 4 * r;
        const int t = t0 + tq;
        if (t >= a.T) break;
        float p0 = q0, p1 = q1;
#pragma unroll
        for (int f = 0; f < 32; ++f) {
            const float ff = s_f[tq][f];
            p0 += d0[f] * ff;
            p1 += d1[f] * ff;
        }
        float e = 0.f;
        if (lane < AD) e += v0 * tanhf(p0 + pm0[r]);
        if (lane + 64 < AD) e += v1 * tanhf(p1 + pm1[r]);
        e = wave_sum(e);
        if (lane == 0) a.energies[(size_t)b * a.T + t] = t < len ? e : -INFINITY;
    }
}

// softmax over T_in, context = weights . memory, cumulative weights (reference tacotron.py:159-164,379).
// Workgroup = (batch element, 64 context channels): the softmax over the T energies is recomputed by each of the
// enc_dim/64 workgroups of an element (T floats - cheaper than a second launch), chunk 0 publishes the weights.
__global__ __launch_bounds__(256) void att_softmax_ctx_kernel(const AttArgs a) {
    extern __shared__ float s_w[];                       // [T]
    __shared__ float red[4];
    __shared__ __attribute__((aligned(16))) float s_part[4][64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.x;
    const int T = a.T;
    float m = -INFINITY;
    for (int t = tid; t < T; t += 256) {
        const float e = a.energies[(size_t)b * T + t];
        s_w[t] = e;
        m = fmaxf(m, e);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
    if (lane == 0) red[wave] = m;
    __syncthreads();
    m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __syncthreads();
    float sum = 0.f;
    for (int t = tid; t < T; t += 256) {
        const float p = expf(s_w[t] - m);
        s_w[t] = p;
        sum += p;
    }
    sum = wave_sum(sum);
    if (lane == 0) red[wave] = sum;
    __syncthreads();
    const float inv = 1.0f / (red[0] + red[1] + red[2] + red[3]);
    for (int t = tid; t < T; t += 256) {
        const float w = s_w[t] * inv;
        s_w[t] = w;
        if (blockIdx.y == 0) {
            a.w_prev[(size_t)b * T + t] = w;
            const float wc = a.w_cum[(size_t)b * T + t] + w;
            a.w_cum[(size_t)b * T + t] = wc;
            if (a.wcum_save) a.wcum_save[(size_t)b * T + t] = wc;
            if (a.align_out) a.align_out[(size_t)b * a.s_align_b + t] = w;
        }
    }
    __syncthreads();
    const int c = blockIdx.y * 64 + lane;
    if ((a.enc_dim & 63) == 0) {
        // 16 lanes x 16 bytes cover the 64 channels of a row; the four lane groups of the four waves take 16 rows per round, eight
        // rounds of loads in flight (the scalar form below waited for memory once per 16 rows: 16 round trips at T = 256 - 9 of
        // the kernel's 10.5 us, profiles/r03_taco_timeline_fwd.md)
        const int cg = lane & 15, rg = lane >> 4;
        const float* mem = a.memory + (size_t)b * T * a.enc_dim + blockIdx.y * 64 + cg * 4;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        for (int t0 = 4 * wave + rg; t0 < T; t0 += 128) {
            f32x4 m[8];
            float w[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int t = t0 + 16 * u;
                const int tc = t < T ? t : T - 1;
                m[u] = *(const f32x4*)(mem + (size_t)tc * a.enc_dim);
                w[u] = t < T ? s_w[tc] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) acc += w[u] * m[u];
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            acc[i] += __shfl_xor(acc[i], 16, 64);
            acc[i] += __shfl_xor(acc[i], 32, 64);
        }
        if (rg == 0) *(f32x4*)&s_part[wave][cg * 4] = acc;
    } else {
        float a4[4] = {0.f, 0.f, 0.f, 0.f};
        if (c < a.enc_dim) {
            const float* mem = a.memory + (size_t)b * T * a.enc_dim + c;
            int t = wave;
            for (; t + 12 < T; t += 16) {
#pragma unroll
                for (int u = 0; u < 4; ++u) a4[u] += s_w[t + 4 * u] * mem[(size_t)(t + 4 * u) * a.enc_dim];
            }
            for (; t < T; t += 4) a4[0] += s_w[t] * mem[(size_t)t * a.enc_dim];
        }
        s_part[wave][lane] = (a4[0] + a4[1]) + (a4[2] + a4[3]);
    }
    __syncthreads();
    if (wave == 0 && c < a.enc_dim) {
        const float acc = (s_part[0][lane] + s_part[1][lane]) + (s_part[2][lane] + s_part[3][lane]);
        a.ctx[(size_t)b * a.enc_dim + c] = acc;
        if (a.ctx_copy) a.ctx_copy[(size_t)b * a.s_ctx_copy + c] = acc;
    }
}

// The energies on the exact-f32 matrix cores (v_mfma_f32_16x16x4_f32), for the reference's shape (attention_dim 128, 32 location
// filters, kernel <= 31) with the transposed dense weight at hand.  One workgroup = 32 positions of one batch element:
//   F[t][f] = sum_k window[t][k] K[k][f]     [32 x 64] x [64 x 32]   (4 tiles on waves 0-3)
//   P[t][a] = sum_f F[t][f] D^T[f][a]        [32 x 32] x [32 x 128]  (wave w: channels 16 w .. 16 w + 15; B fragments straight from
//                                                                     global memory, coalesced over the channel)
// then e[t] = sum_a v[a] tanh(P + q + pm) by a 16-lane butterfly and an 8-wave sum through LDS.  The VALU kernel above spends its
// time on two LDS reads per MAC of the convolution (profiles/r03_taco_step_kernel_counters_before.json).
#define ATT_MQ 32
// FUSE (teacher-forced chain at 9+ items, AttArgs::xbuf): softmax, cumulative weights and context in the SAME launch.  The n_tiles
// workgroups of a batch element (all on one XCD, above) exchange their 32 energies through tagged 8-byte granules (one sc1 store per
// value, the consumer lane polls its own granule: the encoder recurrence's hand-off, lstm_seq_split_kernel), every workgroup then
// has the whole row, redoes the softmax (T exps), writes the weights / cumulative weights of ITS positions and - the first
// enc_dim / 64 of them - one 64-channel chunk of the context.  The softmax + context launch (7 us at B = 32, T = 256, of which ~5
// are the launch itself) leaves the serial chain for one hand-off.  In-place update of w / w_cum is safe: a workgroup writes only
// after it has every partner's energies, which a partner publishes after its own reads of the window (its halo included).
// xbuf: [B][T] granules + 1 error word, zero before step 0 of a sequence; tag = step + 1.  Waits are bounded.
#define ATT_FUSE_SPIN_MAX (1 << 22)
template <bool FUSE>
__global__ __launch_bounds__(512, 2) void att_energy_mfma_kernel(const AttArgs a) {
    constexpr int AD = 128;
    __shared__ float s_cat[2][ATT_MQ + 64];
    // 13 KB of LDS in all, so that a workgroup fits on a CU next to a small-batch GEMM workgroup (144 KB ring): in training the
    // decoder cells run on a helper stream beside this chain
    // (round 4: the 8 KB of partial-query sums share the conv kernel's buffer - with a buffer of their own the kernel had 21 KB and
    // no longer fitted beside the GEMM: profiles/r04_taco_timeline_paced_before.md, the attention launch ran AFTER the decoder cell)
    __shared__ __attribute__((aligned(16))) float s_kb[2 * 64 * 16];        // conv kernel as B operand, one half per filter tile: [f / 16][k = c * KS + j][f % 16], rows >= 2 KS zero
    __shared__ __attribute__((aligned(16))) float s_f[ATT_MQ * 33];
    float (*s_e)[ATT_MQ] = (float (*)[ATT_MQ])s_kb;      // [8][32] partial energies (after the features: s_kb is free)
    float (*s_qp)[128] = (float (*)[128])s_kb;           // [16][128] partial-query sums (before the conv kernel is put there)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lr = lane & 15, lq = lane >> 4;
    // Block -> (item, tile of 32 positions): all tiles of one item on ONE XCD (workgroups go round the 8 XCDs by block number), so
    // that the item's 128 KB of partial queries cross the fabric once and its other tiles find them in that XCD's L2 - with the
    // plain (tile, item) grid every XCD pulled every item's partials (32 MB per launch at B = 32, T = 256 instead of 4 MB).
    // this kernel is the serial chain; what shares its CUs (a decoder cell's GEMM workgroup on the helper stream) is not: its waves
    // ask for issue priority (T2S_ATT_SETPRIO=0 at build time: -DT2S_ATT_NO_SETPRIO)
#ifndef T2S_ATT_NO_SETPRIO
    __builtin_amdgcn_s_setprio(3);
#endif
    const int n_tiles = (a.T + ATT_MQ - 1) / ATT_MQ;
    const int slot = blockIdx.x >> 3;
    int b = (slot / n_tiles) * 8 + (blockIdx.x & 7), tile = slot - (slot / n_tiles) * n_tiles;
    if (a.tile_major) { b = blockIdx.x / n_tiles; tile = blockIdx.x - b * n_tiles; }
    if (b >= a.B) return;
    const int t0 = tile * ATT_MQ;
    const int T = a.T, KS = a.loc_ks, pad = KS >> 1, K2 = 2 * KS;
    for (int i = tid; i < 2 * (ATT_MQ + KS - 1); i += 512) {
        const int c = i / (ATT_MQ + KS - 1), j = i - c * (ATT_MQ + KS - 1);
        const int t = t0 + j - pad;
        const float* src = c ? a.w_cum : a.w_prev;
        s_cat[c][j] = (t >= 0 && t < T) ? src[(size_t)b * T + t] : 0.f;
    }
    // entry tid + 512 j of the B-operand buffer: (filter tile ft, row k, filter f % 16) <- K[f][k], zero rows k >= 2 KS
    float rk[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int i = tid + j * 512;
        const int ft = i >> 10, k = (i >> 4) & 63, f = ft * 16 + (i & 15);
        rk[j] = k < K2 ? a.w_loc_conv[f * K2 + k] : 0.f;
    }
    const int ach = 16 * wave + lr;
    float bd[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) bd[u] = a.w_loc_denseT[(4 * u + lq) * AD + ach];      // B[k = f][col a] = D[a][f]
    // The query: given (a.q), or the sum of the per-workgroup partials the attention cell's launch left (a.q_part, n_part = 256:
    // thread = (four consecutive channels, one of 16 slices of 16 partials), 16 float4 loads in flight at once next to the loads
    // above - one round trip for the 128 KB -, then a 16-way sum through LDS in a fixed order).
    const bool parts = a.q_part != nullptr;
    if (parts) {
        const int aq = tid & 31, part = tid >> 5;
        const float* qp = a.q_part + ((size_t)part * 16 * a.B + b) * AD + 4 * aq;
        const size_t qs = (size_t)a.B * AD;
        f32x4 pv[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) pv[u] = *(const f32x4*)(qp + (size_t)u * qs);
        f32x4 acc = pv[0];
#pragma unroll
        for (int u = 1; u < 16; ++u) acc += pv[u];
        *(f32x4*)&s_qp[part][4 * aq] = acc;
    }
    float qv = parts ? 0.f : a.q[(size_t)b * AD + ach];
    const float vv = a.w_v[ach];
    float pm[2][4];
#pragma unroll
    for (int tt = 0; tt < 2; ++tt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int t = t0 + 16 * tt + 4 * lq + r;
            pm[tt][r] = t < T ? a.pmem[((size_t)b * T + t) * AD + ach] : 0.f;
        }
    __syncthreads();
    if (parts) {
        float q = 0.f;
#pragma unroll
        for (int p16 = 0; p16 < 16; ++p16) q += s_qp[p16][ach];
        qv = q;
        if (tile == 0 && lq == 0) {
            if (a.q_out) a.q_out[(size_t)b * AD + ach] = q;
            if (a.q_save) a.q_save[(size_t)b * AD + ach] = q;
        }
        __syncthreads();                                     // (uniform) the partial sums are read: their buffer takes the conv kernel
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) s_kb[tid + j * 512] = rk[j];
    __syncthreads();
    if (wave < 4) {
        const int tt = wave >> 1, ft = wave & 1;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        float av[16], bv[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int k = 4 * u + lq;
            const int kc = k < K2 ? k : 0;     // (B rows >= 2 KS are zero)
            const int c = kc >= KS ? 1 : 0, j = kc - c * KS;
            av[u] = s_cat[c][16 * tt + lr + j];
            bv[u] = s_kb[ft * 1024 + k * 16 + lr];
        }
#pragma unroll
        for (int u = 0; u < 16; ++u) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u], bv[u], acc, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 4; ++r) s_f[(16 * tt + 4 * lq + r) * 33 + 16 * ft + lr] = acc[r];
    }
    __syncthreads();
#pragma unroll
    for (int tt = 0; tt < 2; ++tt) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        float av[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) av[u] = s_f[(16 * tt + lr) * 33 + 4 * u + lq];
#pragma unroll
        for (int u = 0; u < 8; ++u) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u], bd[u], acc, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float e = vv * tanhf(acc[r] + qv + pm[tt][r]);
            e += __shfl_xor(e, 1, 64);
            e += __shfl_xor(e, 2, 64);
            e += __shfl_xor(e, 4, 64);
            e += __shfl_xor(e, 8, 64);
            if (lr == 0) s_e[wave][16 * tt + 4 * lq + r] = e;
        }
    }
    __syncthreads();
    if constexpr (!FUSE) {
        if (tid < ATT_MQ) {
            const int t = t0 + tid;
            const int len = a.lengths ? a.lengths[b] : T;
            const float e = ((s_e[0][tid] + s_e[1][tid]) + (s_e[2][tid] + s_e[3][tid])) + ((s_e[4][tid] + s_e[5][tid]) + (s_e[6][tid] + s_e[7][tid]));
            if (t < T) a.energies[(size_t)b * T + t] = t < len ? e : -INFINITY;
        }
    } else {
        extern __shared__ float s_w[];                       // [T] energies, then weights
        __shared__ float s_red[8];
        __shared__ int s_fail;
        float (*s_cpart)[64] = (float (*)[64])s_f;              // [8][64] (the features are consumed by now)
        unsigned long long* xrow = a.xbuf + (size_t)b * T;
        if (tid == 0) s_fail = 0;
        if (tid < ATT_MQ) {
            const int t = t0 + tid;
            const int len = a.lengths ? a.lengths[b] : T;
            const float e = ((s_e[0][tid] + s_e[1][tid]) + (s_e[2][tid] + s_e[3][tid])) + ((s_e[4][tid] + s_e[5][tid]) + (s_e[6][tid] + s_e[7][tid]));
            if (t < T) {
                const float em = t < len ? e : -INFINITY;
                const unsigned long long g = ((unsigned long long)a.tag << 32) | (unsigned long long)__float_as_uint(em);
                __hip_atomic_store(xrow + t, g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                s_w[t] = em;
            }
        }
        // the first 256 rows of this workgroup's context chunk do not depend on the weights: requested now, in flight across the hand-off
        const bool do_ctx = tile * 64 < a.enc_dim;           // (uniform: the first enc_dim / 64 workgroups of the element)
        const int cg = lane & 15, rg = lane >> 4, tb0 = 4 * wave + rg;
        const float* mem = a.memory + (size_t)b * T * a.enc_dim + (do_ctx ? tile * 64 : 0) + cg * 4;
        f32x4 mm0[8];
        if (do_ctx) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int t = tb0 + 32 * u;
                mm0[u] = *(const f32x4*)(mem + (size_t)(t < T ? t : T - 1) * a.enc_dim);
            }
        }
        __syncthreads();                                     // (s_fail = 0 is visible before any waiter may set it)
        bool ok = true;
        for (int t = tid; t < T; t += 512) {
            if (t >= t0 && t < t0 + ATT_MQ) continue;        // own tile: already in s_w
            bool got = false;
            for (int it = 0; it < ATT_FUSE_SPIN_MAX; ++it) {
                const unsigned long long g = __hip_atomic_load(xrow + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if ((unsigned)(g >> 32) == a.tag) { s_w[t] = __uint_as_float((unsigned)g); got = true; break; }
                __builtin_amdgcn_s_sleep(1);
            }
            ok = ok && got;
        }
        if (!ok) s_fail = 1;
        __syncthreads();
        if (s_fail) {
            if (tid == 0) __hip_atomic_store(a.xbuf + (size_t)a.B * T, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return;
        }
        // softmax over the row (reference tacotron.py:159-160), every workgroup of the element for itself
        float m = -INFINITY;
        for (int t = tid; t < T; t += 512) m = fmaxf(m, s_w[t]);
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
        if (lane == 0) s_red[wave] = m;
        __syncthreads();
        m = fmaxf(fmaxf(fmaxf(s_red[0], s_red[1]), fmaxf(s_red[2], s_red[3])), fmaxf(fmaxf(s_red[4], s_red[5]), fmaxf(s_red[6], s_red[7])));
        __syncthreads();
        float sum = 0.f;
        for (int t = tid; t < T; t += 512) {
            const float p = expf(s_w[t] - m);
            s_w[t] = p;
            sum += p;
        }
        sum = wave_sum(sum);
        if (lane == 0) s_red[wave] = sum;
        __syncthreads();
        const float inv = 1.0f / (((s_red[0] + s_red[1]) + (s_red[2] + s_red[3])) + ((s_red[4] + s_red[5]) + (s_red[6] + s_red[7])));
        for (int t = tid; t < T; t += 512) {
            const float w = s_w[t] * inv;
            s_w[t] = w;
            if (t >= t0 && t < t0 + ATT_MQ) {                // this workgroup's positions
                a.w_prev[(size_t)b * T + t] = w;
                const float wc = a.w_cum[(size_t)b * T + t] + w;
                a.w_cum[(size_t)b * T + t] = wc;
                if (a.wcum_save) a.wcum_save[(size_t)b * T + t] = wc;
                if (a.align_out) a.align_out[(size_t)b * a.s_align_b + t] = w;
            }
        }
        __syncthreads();
        if (do_ctx) {
            // context chunk `tile`: 16 lanes x 16 bytes cover its 64 channels of a row; 8 waves x 4 lane groups take 32 rows per
            // round, 8 rounds of loads in flight
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int t = tb0 + 32 * u;
                acc += (t < T ? s_w[t] : 0.f) * mm0[u];
            }
            for (int tb = tb0 + 256; tb < T; tb += 256) {
                f32x4 mm[8];
                float ww[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int t = tb + 32 * u;
                    const int tc = t < T ? t : T - 1;
                    mm[u] = *(const f32x4*)(mem + (size_t)tc * a.enc_dim);
                    ww[u] = t < T ? s_w[tc] : 0.f;
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) acc += ww[u] * mm[u];
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                acc[i] += __shfl_xor(acc[i], 16, 64);
                acc[i] += __shfl_xor(acc[i], 32, 64);
            }
            if (rg == 0) *(f32x4*)&s_cpart[wave][cg * 4] = acc;
            __syncthreads();
            if (wave == 0) {
                const float v = ((s_cpart[0][lane] + s_cpart[1][lane]) + (s_cpart[2][lane] + s_cpart[3][lane])) +
                                ((s_cpart[4][lane] + s_cpart[5][lane]) + (s_cpart[6][lane] + s_cpart[7][lane]));
                const int c = tile * 64 + lane;
                a.ctx[(size_t)b * a.enc_dim + c] = v;
                if (a.ctx_copy) a.ctx_copy[(size_t)b * a.s_ctx_copy + c] = v;
            }
            // fewer tiles than context chunks (short inputs): this workgroup also takes chunks tile + n_tiles, tile + 2 n_tiles, ...
            for (int cj = tile + n_tiles; cj * 64 < a.enc_dim; cj += n_tiles) {      // (uniform)
                __syncthreads();                             // s_cpart is read above
                const float* mem2 = a.memory + (size_t)b * T * a.enc_dim + cj * 64 + cg * 4;
                f32x4 acc2 = {0.f, 0.f, 0.f, 0.f};
                for (int tb = tb0; tb < T; tb += 256) {
                    f32x4 mm[8];
                    float ww[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const int t = tb + 32 * u;
                        const int tc = t < T ? t : T - 1;
                        mm[u] = *(const f32x4*)(mem2 + (size_t)tc * a.enc_dim);
                        ww[u] = t < T ? s_w[tc] : 0.f;
                    }
#pragma unroll
                    for (int u = 0; u < 8; ++u) acc2 += ww[u] * mm[u];
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    acc2[i] += __shfl_xor(acc2[i], 16, 64);
                    acc2[i] += __shfl_xor(acc2[i], 32, 64);
                }
                if (rg == 0) *(f32x4*)&s_cpart[wave][cg * 4] = acc2;
                __syncthreads();
                if (wave == 0) {
                    const float v = ((s_cpart[0][lane] + s_cpart[1][lane]) + (s_cpart[2][lane] + s_cpart[3][lane])) +
                                    ((s_cpart[4][lane] + s_cpart[5][lane]) + (s_cpart[6][lane] + s_cpart[7][lane]));
                    const int c = cj * 64 + lane;
                    a.ctx[(size_t)b * a.enc_dim + c] = v;
                    if (a.ctx_copy) a.ctx_copy[(size_t)b * a.s_ctx_copy + c] = v;
                }
            }
        }
    }
}

// shapes the one-launch form (energies + softmax + context, AttArgs::xbuf) covers: the matrix-core energies kernel's, enc_dim a
// multiple of 64 (workgroup `tile` takes the context chunks tile, tile + n_tiles, ...), a row of weights within 2 KB of LDS (so that
// the workgroup still fits beside a small-batch GEMM workgroup)
bool t2s_att_energy_ctx_ok(const AttArgs& a) {
    static const bool no_mfma = getenv("T2S_ATT_VALU") != nullptr;
    static const bool plain = getenv("T2S_ENERGY_XCD") && atoi(getenv("T2S_ENERGY_XCD")) == 0;
    if (no_mfma || plain || a.att_dim != 128 || a.loc_f != 32 || a.loc_ks > 31 || !a.w_loc_denseT) return false;
    if (a.q_part && a.n_part != 256) return false;
    const int n_tiles = (a.T + ATT_MQ - 1) / ATT_MQ;
    (void)n_tiles;                                           // (fewer tiles than context chunks: a workgroup takes several)
    return a.T <= 512 && (a.enc_dim & 63) == 0 && a.tag != 0;
}
hipError_t t2s_launch_att_energy(const AttArgs& a, hipStream_t stream) {
    // T2S_ATT_VALU set: the VALU kernel (A/B switch, shared with the fused small-batch form)
    static const bool no_mfma = getenv("T2S_ATT_VALU") != nullptr;
    if (!no_mfma && a.att_dim == 128 && a.loc_f == 32 && a.loc_ks <= 31 && a.w_loc_denseT) {
        if (a.q_part && a.n_part != 256) return hipErrorInvalidValue;
        static const bool plain = getenv("T2S_ENERGY_XCD") && atoi(getenv("T2S_ENERGY_XCD")) == 0;
        AttArgs aa = a;
        aa.tile_major = plain ? 1 : 0;
        dim3 grid(8 * ((a.B + 7) / 8) * ((a.T + ATT_MQ - 1) / ATT_MQ));
        if (a.xbuf) {
            if (plain || !t2s_att_energy_ctx_ok(a)) return hipErrorInvalidValue;
            hipLaunchKernelGGL(att_energy_mfma_kernel<true>, grid, dim3(512), (size_t)a.T * sizeof(float), stream, aa);
            return hipGetLastError();
        }
        hipLaunchKernelGGL(att_energy_mfma_kernel<false>, grid, dim3(512), 0, stream, aa);
        return hipGetLastError();
    }
    if (a.q_part) return hipErrorInvalidValue;          // (only the matrix-core kernel sums partial queries)
    dim3 grid((a.T + ATT_TQ - 1) / ATT_TQ, a.B);
    hipLaunchKernelGGL(att_energy_kernel, grid, dim3(256), (size_t)a.loc_f * 2 * a.loc_ks * sizeof(float), stream, a);
    return hipGetLastError();
}
hipError_t t2s_launch_att_softmax_ctx(const AttArgs& a, hipStream_t stream) {
    hipLaunchKernelGGL(att_softmax_ctx_kernel, dim3(a.B, (a.enc_dim + 63) / 64), dim3(256), (size_t)a.T * sizeof(float), stream, a);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// Small-batch form of one attention step: query, location features, energies, softmax, context and the
// cumulative weights in ONE launch, one 1024-thread workgroup per batch element (T <= ATT_FUSED_MAXT).
// Cuts three dependent launches (~1.5-3 us of boundary each plus their serial prologues) out of every
// autoregressive step; the split kernels above remain for large batches, where they fill the chip.
#define ATT_FUSED_MAXT 512
__global__ __launch_bounds__(1024) void att_fused_kernel(const AttArgs a) {
    PROBE_BEGIN(300)
    extern __shared__ float s_dyn[];                 // [T][33] location features, then [T] energies
    __shared__ float s_q[128];
    __shared__ float s_qp[8][128];
    __shared__ float s_d[32 * 128];
    __shared__ float s_cat[2][ATT_FUSED_MAXT + 64];
    __shared__ float s_k[32 * 2 * 63];
    __shared__ float red[16];
    __shared__ float s_ctx[2][512];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.x;
    const int T = a.T, AD = a.att_dim, KS = a.loc_ks, F = a.loc_f, pad = KS >> 1, A = a.att_rnn;
    float* s_f = s_dyn;                               // [T][33]
    float* s_e = s_dyn + (size_t)T * 33;              // [T]
    // ---- every independent global load is issued first and TOGETHER: fixed trip counts and clamped addresses, so the
    // compiler emits one batch of loads and one wait instead of a load -> wait -> LDS store round trip per loop iteration
    // (this kernel is a chain of memory round trips at B = 1; it was 19 of them, ~23 us) ----
    {
        const int nK = F * 2 * KS, nC = 2 * (T + KS - 1), TK = T + KS - 1;
        float rk[4], rd[4], rc[2];
#pragma unroll
        for (int j = 0; j < 4; ++j) {                          // nK <= 32 * 2 * 63 = 4032, s_d is 4096
            const int i = tid + j * 1024;
            rk[j] = a.w_loc_conv[i < nK ? i : 0];
            const int f = i >> 7, ai = i & 127;
            rd[j] = a.w_loc_denseT[(f < F && ai < AD) ? f * AD + ai : 0];
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {                          // nC <= 2 * (512 + 62) = 1148
            const int i = tid + j * 1024;
            const int c = i >= TK ? 1 : 0, jj = i - c * TK, t = jj - pad;
            const bool in = i < nC && t >= 0 && t < T;
            rc[j] = (c ? a.w_cum : a.w_prev)[(size_t)b * T + (in ? t : 0)];
            if (!in) rc[j] = 0.f;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int i = tid + j * 1024;
            if (i < nK) s_k[i] = rk[j];
            const int f = i >> 7, ai = i & 127;
            s_d[i] = (f < F && ai < AD) ? rd[j] : 0.f;
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int i = tid + j * 1024;
            if (i < nC) s_cat[i >= TK ? 1 : 0][i >= TK ? i - TK : i] = rc[j];
        }
    }
    const float v0 = a.w_v[lane < AD ? lane : 0] * (lane < AD ? 1.f : 0.f);
    const float v1 = a.w_v[lane + 64 < AD ? lane + 64 : 0] * (lane + 64 < AD ? 1.f : 0.f);
    PROBE_MID(0)
    // ---- query: sum of the per-workgroup partials the attention LSTM cell just wrote, or W_q h_att ----
    if (a.q_part) {
        const int ai = tid & 127, part = tid >> 7;            // 8 slices of the partial list per output
        float acc = 0.f;
        if (ai < AD) {
            float a8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            int w = part;
            for (; w + 120 < a.n_part; w += 128) {         // 16 independent loads in flight (256 partials: two round trips)
                float pv[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) pv[u] = a.q_part[((size_t)(w + 8 * u) * a.B + b) * AD + ai];
#pragma unroll
                for (int u = 0; u < 16; ++u) a8[u & 7] += pv[u];
            }
            for (; w + 56 < a.n_part; w += 64) {           // 8 independent loads in flight
#pragma unroll
                for (int u = 0; u < 8; ++u) a8[u] += a.q_part[((size_t)(w + 8 * u) * a.B + b) * AD + ai];
            }
            for (; w < a.n_part; w += 8) a8[0] += a.q_part[((size_t)w * a.B + b) * AD + ai];
            acc = ((a8[0] + a8[1]) + (a8[2] + a8[3])) + ((a8[4] + a8[5]) + (a8[6] + a8[7]));
        }
        s_qp[part][ai] = acc;
        __syncthreads();
        if (tid < AD) {
            float q = 0.f;
#pragma unroll
            for (int p8 = 0; p8 < 8; ++p8) q += s_qp[p8][tid];
            s_q[tid] = q;
        }
    } else {
        // loads are unconditional from clamped addresses and masked afterwards (no branch + vmcnt(0) per load: see load_row)
        f32x4 x[4];
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const int k = (v * 64 + lane) * 4;
            x[v] = *(const f32x4*)(a.h_att + (size_t)b * A + (k < A ? k : 0));
        }
#pragma unroll
        for (int v = 0; v < 4; ++v)
            if ((v * 64 + lane) * 4 >= A) x[v] = (f32x4){0.f, 0.f, 0.f, 0.f};
        for (int r0 = wave * 2; r0 < AD; r0 += 32) {       // (fallback path: the decode driver hands over partial queries)
            f32x4 w[2][4];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const int k = (v * 64 + lane) * 4;
                    const int rr = r0 + i < AD ? r0 + i : AD - 1;
                    w[i][v] = *(const f32x4*)(a.w_query + (size_t)rr * A + (k < A ? k : 0));      // x is 0 where k >= A
                }
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                float acc = 0.f;
#pragma unroll
                for (int v = 0; v < 4; ++v)
                    acc += w[i][v][0] * x[v][0] + w[i][v][1] * x[v][1] + w[i][v][2] * x[v][2] + w[i][v][3] * x[v][3];
                acc = wave_sum(acc);
                if (lane == 0 && r0 + i < AD) s_q[r0 + i] = acc;
            }
        }
    }
    __syncthreads();
    PROBE_MID(1)
    if (a.q_save && tid < AD) a.q_save[(size_t)b * AD + tid] = s_q[tid];
    // ---- location features f[t][:] = conv1d([w ; w_cum]) ----
    for (int i = tid; i < T * F; i += 1024) {
        const int t = i / F, f = i - t * F;
        float acc = 0.f;
        for (int c = 0; c < 2; ++c)
#pragma unroll 8
            for (int j = 0; j < KS; ++j) acc += s_k[(f * 2 + c) * KS + j] * s_cat[c][t + j];
        s_f[t * 33 + f] = acc;
    }
    __syncthreads();
    PROBE_MID(2)
    // ---- energies: attention_dim on lanes (2 per lane), one wave per time step ----
    {
        const float q0 = lane < AD ? s_q[lane] : 0.f, q1 = lane + 64 < AD ? s_q[lane + 64] : 0.f;
        const int len = a.lengths ? a.lengths[b] : T;
        for (int tb = wave; tb < T; tb += 64) {             // four time steps per wave per round: their loads go out together
            float pm0[4], pm1[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int t = tb + 16 * r;
                const float* pm = a.pmem + ((size_t)b * T + (t < T ? t : 0)) * AD;
                pm0[r] = pm[lane < AD ? lane : 0];
                pm1[r] = pm[lane + 64 < AD ? lane + 64 : 0];
            }
#pragma unroll 1
            for (int r = 0; r < 4; ++r) {
                const int t = tb + 16 * r;
                if (t >= T) break;
                float p0 = q0, p1 = q1;
#pragma unroll
                for (int f = 0; f < 32; ++f) {
                    const float ff = s_f[t * 33 + f];
                    p0 += s_d[f * 128 + lane] * ff;
                    p1 += s_d[f * 128 + 64 + lane] * ff;
                }
                float e = 0.f;
                if (lane < AD) e += v0 * tanhf(p0 + pm0[r]);
                if (lane + 64 < AD) e += v1 * tanhf(p1 + pm1[r]);
                e = wave_sum(e);
                if (lane == 0) s_e[t] = t < len ? e : -INFINITY;
            }
        }
    }
    __syncthreads();
    PROBE_MID(3)
    // ---- softmax over T ----
    float m = -INFINITY;
    for (int t = tid; t < T; t += 1024) m = fmaxf(m, s_e[t]);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
    if (lane == 0) red[wave] = m;
    __syncthreads();
    m = red[0];
    for (int i = 1; i < 16; ++i) m = fmaxf(m, red[i]);
    __syncthreads();
    float sum = 0.f;
    for (int t = tid; t < T; t += 1024) {
        const float p = expf(s_e[t] - m);
        s_e[t] = p;
        sum += p;
    }
    sum = wave_sum(sum);
    if (lane == 0) red[wave] = sum;
    __syncthreads();
    float tot = 0.f;
    for (int i = 0; i < 16; ++i) tot += red[i];
    const float inv = 1.0f / tot;
    for (int t = tid; t < T; t += 1024) {
        const float w = s_e[t] * inv;
        s_e[t] = w;
        a.w_prev[(size_t)b * T + t] = w;
        const float wc = s_cat[1][t + pad] + w;                 // the cumulative weights were staged in LDS at entry
        a.w_cum[(size_t)b * T + t] = wc;
        if (a.wcum_save) a.wcum_save[(size_t)b * T + t] = wc;
        if (a.align_out) a.align_out[(size_t)b * a.s_align_b + t] = w;
    }
    __syncthreads();
    PROBE_MID(4)
    // ---- context = weights . memory : two threads per channel, each half of the time range ----
    {
        const int c = tid & 511, half = tid >> 9;
        float acc = 0.f;
        if (c < a.enc_dim) {
            const int tb = half ? (T + 1) / 2 : 0, te = half ? T : (T + 1) / 2;
            const float* mem = a.memory + (size_t)b * T * a.enc_dim + c;
            float a4[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            int t = tb;
            for (; t + 16 <= te; t += 16) {       // 16 independent loads in flight per thread (T = 64: two round trips)
                float mv[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) mv[u] = mem[(size_t)(t + u) * a.enc_dim];
#pragma unroll
                for (int u = 0; u < 16; ++u) a4[u & 7] += s_e[t + u] * mv[u];
            }
            for (; t + 8 <= te; t += 8) {
#pragma unroll
                for (int u = 0; u < 8; ++u) a4[u] += s_e[t + u] * mem[(size_t)(t + u) * a.enc_dim];
            }
            for (; t < te; ++t) a4[0] += s_e[t] * mem[(size_t)t * a.enc_dim];
            acc = ((a4[0] + a4[1]) + (a4[2] + a4[3])) + ((a4[4] + a4[5]) + (a4[6] + a4[7]));
        }
        s_ctx[half][c] = acc;
    }
    __syncthreads();
    if (tid < a.enc_dim && tid < 512) {
        const float v = s_ctx[0][tid] + s_ctx[1][tid];
        a.ctx[(size_t)b * a.enc_dim + tid] = v;
        if (a.ctx_copy) a.ctx_copy[(size_t)b * a.s_ctx_copy + tid] = v;
    }
    PROBE_END()
}

// value of lane (l + n) mod 16 within the same 16-lane row (DPP row_ror:n), n in {1, 2, 4, 8}
template <int N>
static __device__ __forceinline__ float row16_ror_c(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x120 + N, 0xf, 0xf, false));
}
static __device__ __forceinline__ float row16_ror(float v, int n) {
    return n == 8 ? row16_ror_c<8>(v) : n == 4 ? row16_ror_c<4>(v) : n == 2 ? row16_ror_c<2>(v) : row16_ror_c<1>(v);
}

// ------------------------------------------------------------------------------------------------
// Small-batch attention, matrix-core form (the default for F = 32 location filters, attention_dim = 128).
// Same operator as att_fused_kernel; the two GEMM-shaped pieces run on the exact-f32 matrix cores
// (v_mfma_f32_16x16x4_f32: f32 operands, f32 accumulate) instead of LDS-bound VALU loops:
//   location features  F[t][f]  = sum_{c,j} cat[c][t + j] * K[f][c][j]       (tacotron.py:96-107 location_conv)
//       = [T x 2*KS] (sliding windows of [w ; w_cum], read straight out of LDS as the A operand) x [2*KS x 32]
//   pre-activation     P[t][a]  = sum_f F[t][f] * D[a][f]                       (location_dense)
//       = [T x 32] x [32 x 128], epilogue  e[t] = sum_a v[a] * tanh(P + q[a] + pmem[t][a])   (tacotron.py:137-143)
// At B = 1, T = 64 the VALU form spent 5.1 us in the convolution and 8.0 us in the energies (124 and 96 ds_read_b32 per
// output and wave: LDS-bound in ONE CU); here both are a few dozen MFMAs per wave.
// LDS operand layouts are padded so that the b32 fragment reads are bank-conflict free: K as [k][48], D as [f][144].
// Gate-stream role of the fused attention launch (GateStreamArgs, tacotron_ops.h): workgroups B .. gridDim.x - 1.
// Unit = one [H = 1024] weight row of one of the three blocks (3 * rows units of 4 KB).  Wave w of the role (NW waves in all)
// takes units w, w + NW, w + 2 NW and - the first (n_units - 3 NW) waves only - w + 3 NW: ONE pass, every weight load of a wave
// (12 or 16 x 16 bytes per lane) requested before the first use, so no workgroup pays a second memory round trip (with 255
// workgroups and 12288 units a strided loop left 48 waves a second pass: 15.6 us per launch instead of ~11).  Non-temporal
// weight loads: each row is read once per step by one wave.  The two input vectors (h_dec(t-1), h_att(t)) are staged in LDS
// once per workgroup.  Requires 3 NW <= n_units <= 4 NW (checked on the host).
static __device__ __forceinline__ void gate_stream_role(const GateStreamArgs& g, int role_block, int role_blocks, float* s_x) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int NW = role_blocks * 16, gw = role_block * 16 + wave;
    const int n_units = 3 * g.rows;
    const bool four = gw + 3 * NW < n_units;                    // wave-uniform
    f32x4 w[4][4];
    float* ob[4];
    int rr[4];
    bool first[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int un = (j < 3 || four) ? gw + j * NW : gw;      // (j = 3 without a fourth unit: a row this wave reads anyway)
        const int blk = un / g.rows, r = un - blk * g.rows;
        const float* W = blk == 0 ? g.W0 : (blk == 1 ? g.W1 : g.W2);
        const int ld = blk == 0 ? g.ld0 : (blk == 1 ? g.ld1 : g.ld2);
        first[j] = blk == 0;
        ob[j] = blk == 0 ? g.out0 : (blk == 1 ? g.out1 : g.out2);
        rr[j] = r;
        const float* wp = W + (size_t)r * ld + 4 * lane;
        if (j < 3 || four) {
#pragma unroll
            for (int v = 0; v < 4; ++v) w[j][v] = T2S_WLOAD((const f32x4*)(wp + 256 * v));
        }
    }
    // input vectors -> LDS: [b][2][H] (h_dec, h_att), 2 * B * H floats <= 64 KB
    for (int i = threadIdx.x; i < g.B * 2 * (g.H / 4); i += 1024) {
        const int b = i / (2 * (g.H / 4)), rem = i - b * 2 * (g.H / 4), which = rem / (g.H / 4), k4 = rem - which * (g.H / 4);
        *(f32x4*)(s_x + (size_t)i * 4) = *(const f32x4*)((which ? g.x12 : g.x0) + (size_t)b * g.H + 4 * k4);
    }
    __syncthreads();
    for (int b = 0; b < g.B; ++b) {
        const float* xd = s_x + (size_t)b * 2 * g.H, *xa = xd + g.H;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (j == 3 && !four) break;
            const float* xs = first[j] ? xd : xa;                  // (wave-uniform: a unit belongs to one block)
            float acc = 0.f;
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const f32x4 x = *(const f32x4*)(xs + 4 * lane + 256 * v);
                acc += w[j][v][0] * x[0] + w[j][v][1] * x[1] + w[j][v][2] * x[2] + w[j][v][3] * x[3];
            }
            acc = wave_sum(acc);
            if (lane == 0) ob[j][(size_t)b * g.rows + rr[j]] = acc;
        }
    }
}

// PLOC: the location term of this step was computed one launch earlier (AttArgs::ploc): no matrix-core stage here.
template <bool STREAM, bool PLOC = false>
__global__ __launch_bounds__(1024) void att_fused_mfma_kernel(const AttArgs a, const GateStreamArgs gs) {
    if constexpr (STREAM) {
        if ((int)blockIdx.x >= a.B) {           // whole workgroups take this branch: no barrier of the attention role is skipped
#ifdef T2S_ATTSTREAM_ABLATE
            if (gs.dbg == 2) return;
#endif
            extern __shared__ __attribute__((aligned(16))) float s_role[];
            gate_stream_role(gs, (int)blockIdx.x - a.B, (int)gridDim.x - a.B, s_role);
            return;
        }
#ifdef T2S_ATTSTREAM_ABLATE
        if (gs.dbg == 1) return;
#endif
    }
    PROBE_BEGIN(301)
    extern __shared__ __attribute__((aligned(16))) float s_dyn[];   // s_f [max(Tp*33, 4096)] | s_e [Tp] | s_ep [8][Tp] | s_kb [KP][48]
    __shared__ float s_q[128];
    __shared__ float s_v[128];
    __shared__ __attribute__((aligned(16))) float s_d[128 * 36];     // location_dense weight D[a][f] (natural layout), rows padded to 36
    __shared__ float s_cat[2][ATT_FUSED_MAXT + 16 + 64];
    __shared__ float red[16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.x;
    const int T = a.T, KS = a.loc_ks, pad = KS >> 1, A = a.att_rnn;
    constexpr int AD = 128, F = 32;
    const int Tp = (T + 15) & ~15, K2 = 2 * KS, KP = (K2 + 3) & ~3, TKp = Tp + KS - 1;
    constexpr int FS = 36;                            // floats per row of s_f / s_d: 16-byte aligned rows for b128 fragment reads
    const int nF = Tp * FS > 4096 ? Tp * FS : 4096;   // s_f doubles as [32][128] / [8][512] reduction scratch
    float* s_f = s_dyn;
    float* s_e = s_f + nF;
    float* s_ep = s_e + Tp;
    float* s_kb = s_ep + 8 * Tp;
    const int lr = lane & 15, lq = lane >> 4;
    // ---- every independent global load first, in one batch (fixed trip counts, clamped addresses) ----
    float pm[3][4];
    // the partial queries the attention cell's launch left (written by 256 other workgroups: they come out of memory, the slowest
    // loads of this kernel) are requested with the first batch, not after it
    const bool q_parts = a.q_part && (a.n_part & 31) == 0 && a.n_part <= 256;
    f32x4 qpv[8];
    f32x4 plv[PLOC ? 4 : 1];
    {
        const int nK = F * K2;
        float rk[4], rd[4], rc[2];
        if constexpr (!PLOC) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {                      // nK <= 32 * 2 * 63 = 4032; D is 32 x 128 = 4096
                const int i = tid + j * 1024;
                rk[j] = a.w_loc_conv[i < nK ? i : 0];
                rd[j] = a.w_loc_dense[i];                      // D[a][f]: i = a * 32 + f
            }
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {                          // 2 * TKp <= 2 * (512 + 62) = 1148
            const int i = tid + j * 1024;
            const int c = i >= TKp ? 1 : 0, t = i - c * TKp - pad;
            const bool in = i < 2 * TKp && t >= 0 && t < T;
            rc[j] = (c ? a.w_cum : a.w_prev)[(size_t)b * T + (in ? t : 0)];
            if (!in) rc[j] = 0.f;
        }
        const float vv = tid < AD ? a.w_v[tid] : 0.f;
        // processed memory of this wave's first three stage-2 tiles (tile id = wave + 16 i: t-tile id >> 3, a-tile id & 7: T <= 96);
        // later tiles are requested one tile ahead inside the loop (two were prefetched at first: beyond 64 encoder
        // positions the later tiles loaded where they were used, one exposed memory round trip each on the launch's critical
        // workgroup)
        if constexpr (!PLOC) {
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const int id = wave + 16 * i, tt = id >> 3, at = id & 7;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int t = 16 * tt + 4 * lq + r;
                    pm[i][r] = a.pmem[((size_t)b * T + (t < T ? t : 0)) * AD + 16 * at + lr];
                }
            }
        } else {
            // location term + processed memory of the first 128 positions (thread = four channels of positions tid / 32 + 32 j)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int t = (tid >> 5) + 32 * j;
                const size_t o = ((size_t)b * T + (t < T ? t : 0)) * AD + 4 * (tid & 31);
                plv[j] = *(const f32x4*)(a.ploc + o);         // (written by the previous launch: out of memory, the slow ones - early)
            }
        }
        // (issued BEHIND the loads above: a counted wait then lets those - L2 hits - be staged while the partials are in flight)
        {
            const int aq = tid & 31, part = tid >> 5;
            const int per = q_parts ? a.n_part >> 5 : 0;         // 8 for 256 partials
            const float* qp = q_parts ? a.q_part + ((size_t)part * per * a.B + b) * AD + 4 * aq : a.w_v;
            const size_t qs = q_parts ? (size_t)a.B * AD : 0;
    #pragma unroll
            for (int u = 0; u < 8; ++u) qpv[u] = *(const f32x4*)(qp + (size_t)(u < per ? u : 0) * qs);
        }
        if constexpr (!PLOC) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int i = tid + j * 1024;
                if (i < nK) {
                    const int f = i / K2, k = i - f * K2;      // K[f][c][j] -> B operand [k = c*KS + j][f]
                    s_kb[k * 48 + f] = rk[j];
                }
                s_d[(i >> 5) * 36 + (i & 31)] = rd[j];
            }
            if (tid < (KP - K2) * F) s_kb[(K2 + tid / F) * 48 + (tid % F)] = 0.f;  // zero rows that pad K to a multiple of 4
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int i = tid + j * 1024;
            if (i < 2 * TKp) s_cat[i >= TKp ? 1 : 0][i >= TKp ? i - TKp : i] = rc[j];
        }
        if (tid < AD) s_v[tid] = vv;
    }
    PROBE_MID(0)
    // ---- query: sum of the per-workgroup partials the attention LSTM cell just wrote, or W_q h_att ----
    if (PLOC || q_parts) {      // (the PLOC form is launched with partial queries only: its W_q . h_att branch is compiled out)
        // thread = (slice of the partial list, four consecutive outputs): the n_part / 32 float4 partials requested at entry, then a
        // 32-way sum through LDS in a fixed order
        const int aq = tid & 31, part = tid >> 5;
        const int per = a.n_part >> 5;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (u < per) acc += qpv[u];
        float* s_q32 = s_f;                                   // [32][128] scratch (s_f is not live yet)
        *(f32x4*)(s_q32 + part * 128 + 4 * aq) = acc;
        __syncthreads();
        if (tid < AD) {
            float q = 0.f;
#pragma unroll
            for (int p8 = 0; p8 < 32; ++p8) q += s_q32[p8 * 128 + tid];
            s_q[tid] = q;
        }
    } else {
        f32x4 x[4];
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const int k = (v * 64 + lane) * 4;
            x[v] = *(const f32x4*)(a.h_att + (size_t)b * A + (k < A ? k : 0));
        }
#pragma unroll
        for (int v = 0; v < 4; ++v)
            if ((v * 64 + lane) * 4 >= A) x[v] = (f32x4){0.f, 0.f, 0.f, 0.f};
        for (int r0 = wave * 2; r0 < AD; r0 += 32) {
            f32x4 w[2][4];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const int k = (v * 64 + lane) * 4;
                    w[i][v] = *(const f32x4*)(a.w_query + (size_t)(r0 + i) * A + (k < A ? k : 0));
                }
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                float acc = 0.f;
#pragma unroll
                for (int v = 0; v < 4; ++v)
                    acc += w[i][v][0] * x[v][0] + w[i][v][1] * x[v][1] + w[i][v][2] * x[v][2] + w[i][v][3] * x[v][3];
                acc = wave_sum(acc);
                if (lane == 0) s_q[r0 + i] = acc;
            }
        }
    }
    __syncthreads();
    PROBE_MID(1)
    if (a.q_save && tid < AD) a.q_save[(size_t)b * AD + tid] = s_q[tid];
    if constexpr (PLOC) {
        // ---- energies from the precomputed location term: e[t] = sum_a v[a] tanh(P_loc[t][a] + q[a] + pmem[t][a]) (tacotron.py:137-143).
        // thread = (four channels, position tid / 32 + 32 j); the 32 lanes of a position meet by shuffles ----
        const int a4 = tid & 31, tr = tid >> 5;
        const f32x4 q4 = *(const f32x4*)&s_q[4 * a4], v4 = *(const f32x4*)&s_v[4 * a4];
        const int len = a.lengths ? a.lengths[b] : T;
        for (int tb0 = 0; tb0 < T; tb0 += 128) {
            f32x4 pmv4[4];                                     // processed memory: L2 hits, requested here (all of a pass together)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int t = tb0 + tr + 32 * j;
                const size_t o = ((size_t)b * T + (t < T ? t : 0)) * AD + 4 * a4;
                if (tb0 > 0) plv[j] = *(const f32x4*)(a.ploc + o);     // (beyond 128 positions)
                pmv4[j] = *(const f32x4*)(a.pmem + o);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int t = tb0 + tr + 32 * j;
                float e = 0.f;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const float x = plv[j][c] + q4[c] + pmv4[j][c];
                    e += v4[c] * (1.0f - 2.0f * __builtin_amdgcn_rcpf(__expf(2.0f * x) + 1.0f));
                }
#pragma unroll
                for (int off = 16; off > 0; off >>= 1) e += __shfl_xor(e, off, 64);
                if (a4 == 0 && t < T) s_e[t] = t < len ? e : -INFINITY;
            }
        }
    } else {
    // ---- stage 1: location features on the matrix cores.  A[t][k] = cat[c][t + j] (k = c*KS + j), B[k][f] = K[f][c][j] ----
    {
        const int n_tiles = (Tp >> 4) * 2;
        for (int id = wave; id < n_tiles; id += 16) {
            const int tt = id >> 1, ft = id & 1;
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            for (int ks0 = 0; ks0 < KP; ks0 += 64) {           // fragments of up to 16 k-steps are read first, then the MFMA chain runs
                float av[16], bv[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) {
                    const int k = ks0 + 4 * u + lq;
                    const int kc = k < K2 ? k : 0;              // (B rows >= 2*KS are zero)
                    const int c = kc >= KS ? 1 : 0, j = kc - c * KS;
                    av[u] = s_cat[c][16 * tt + lr + j];
                    bv[u] = s_kb[(k < KP ? k : 0) * 48 + 16 * ft + lr];
                }
#pragma unroll
                for (int u = 0; u < 16; ++u)
                    if (ks0 + 4 * u < KP) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u], bv[u], acc, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) s_f[(16 * tt + 4 * lq + r) * FS + 16 * ft + lr] = acc[r];
        }
    }
    __syncthreads();
    PROBE_MID(2)
    // ---- stage 2: P = F . D^T on the matrix cores, energies in the epilogue ----
    {
        const int n_tiles = (Tp >> 4) * 8;
        int it = 0;
        float pmn[4] = {0.f, 0.f, 0.f, 0.f};                   // tiles 3, 4, ...: requested during the tile before
        for (int id = wave; id < n_tiles; id += 16, ++it) {
            const int tt = id >> 3, at = id & 7;
            float pmv[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) pmv[r] = it == 0 ? pm[0][r] : it == 1 ? pm[1][r] : it == 2 ? pm[2][r] : pmn[r];
            if (it >= 2 && id + 16 < n_tiles) {
                const int tn = (id + 16) >> 3, an = (id + 16) & 7;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int t = 16 * tn + 4 * lq + r;
                    pmn[r] = a.pmem[((size_t)b * T + (t < T ? t : 0)) * AD + 16 * an + lr];
                }
            }
            // K order of the eight MFMA steps: step u of lane group lq multiplies k = 8 lq + u (any bijection onto 0 .. 31 serves as
            // long as A and B agree), so a lane's eight A values and eight B values are CONTIGUOUS: two ds_read_b128 per operand
            // instead of eight ds_read_b32 with 2- to 4-way bank conflicts (this stage was LDS-bound: 1.3 us per tile, growing
            // linearly with the encoder length on the one workgroup the whole launch waits for)
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            const f32x4 a0 = *(const f32x4*)(s_f + (16 * tt + lr) * FS + 8 * lq), a1 = *(const f32x4*)(s_f + (16 * tt + lr) * FS + 8 * lq + 4);
            const f32x4 b0 = *(const f32x4*)(s_d + (16 * at + lr) * 36 + 8 * lq), b1 = *(const f32x4*)(s_d + (16 * at + lr) * 36 + 8 * lq + 4);
#pragma unroll
            for (int u = 0; u < 4; ++u) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[u], b0[u], acc, 0, 0, 0);
#pragma unroll
            for (int u = 0; u < 4; ++u) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[u], b1[u], acc, 0, 0, 0);
            const float qa = s_q[16 * at + lr], va = s_v[16 * at + lr];
            float ev[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                // tanh(x) = 1 - 2 / (e^(2x) + 1): v_exp_f32 / v_rcp_f32 (1 ulp each), saturates cleanly
                const float x = acc[r] + qa + pmv[r];
                ev[r] = va * (1.0f - 2.0f * __builtin_amdgcn_rcpf(__expf(2.0f * x) + 1.0f));
            }
            // sum over the 16 lanes that hold the 16 attention channels of this tile: DPP row rotations (one VALU
            // instruction each, no LDS round trip as ds_bpermute-based shuffles would need), four independent chains
#pragma unroll
            for (int sh = 8; sh >= 1; sh >>= 1) {
#pragma unroll
                for (int r = 0; r < 4; ++r) ev[r] += row16_ror(ev[r], sh);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (lr == 0) s_ep[at * Tp + 16 * tt + 4 * lq + r] = ev[r];
        }
    }
    __syncthreads();
    {
        const int len = a.lengths ? a.lengths[b] : T;
        for (int t = tid; t < T; t += 1024) {
            float e = 0.f;
#pragma unroll
            for (int at = 0; at < 8; ++at) e += s_ep[at * Tp + t];      // fixed order: bitwise reproducible
            s_e[t] = t < len ? e : -INFINITY;
        }
    }
    }
    __syncthreads();
    PROBE_MID(3)
    // ---- softmax over T ----
    float m = -INFINITY;
    for (int t = tid; t < T; t += 1024) m = fmaxf(m, s_e[t]);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
    if (lane == 0) red[wave] = m;
    __syncthreads();
    m = red[0];
    for (int i = 1; i < 16; ++i) m = fmaxf(m, red[i]);
    __syncthreads();
    float sum = 0.f;
    for (int t = tid; t < T; t += 1024) {
        const float p = expf(s_e[t] - m);
        s_e[t] = p;
        sum += p;
    }
    sum = wave_sum(sum);
    if (lane == 0) red[wave] = sum;
    __syncthreads();
    float tot = 0.f;
    for (int i = 0; i < 16; ++i) tot += red[i];
    const float inv = 1.0f / tot;
    for (int t = tid; t < T; t += 1024) {
        const float w = s_e[t] * inv;
        s_e[t] = w;
        a.w_prev[(size_t)b * T + t] = w;
        const float wc = s_cat[1][t + pad] + w;                 // the cumulative weights were staged in LDS at entry
        a.w_cum[(size_t)b * T + t] = wc;
        if (a.wcum_save) a.wcum_save[(size_t)b * T + t] = wc;
        if (a.align_out) a.align_out[(size_t)b * a.s_align_b + t] = w;
    }
    __syncthreads();
    PROBE_MID(4)
    // ---- context = weights . memory.  thread = (four consecutive channels, one of eight time slices): float4 loads, up to 16
    // of them in flight per thread (T <= 128: ONE memory round trip for the whole [T x enc_dim] block), then an 8-way sum
    // through LDS in a fixed order ----
    {
        const int cq = tid & 127, sl = tid >> 7;               // enc_dim <= 512 = 128 float4
        const int per = (T + 7) >> 3;                          // time steps per slice
        const int tb = sl * per, te = min(T, tb + per);
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        if (4 * cq < a.enc_dim) {
            const float* mem = a.memory + (size_t)b * T * a.enc_dim + 4 * cq;
            int t = tb;
            for (; t + 16 <= te; t += 16) {                     // sixteen float4 loads in flight per thread (T = 128: one round trip)
                f32x4 mv[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) mv[u] = *(const f32x4*)(mem + (size_t)(t + u) * a.enc_dim);
#pragma unroll
                for (int u = 0; u < 16; ++u) acc += mv[u] * s_e[t + u];
            }
            for (; t + 8 <= te; t += 8) {                       // eight float4 loads in flight per thread, no duplicates
                f32x4 mv[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) mv[u] = *(const f32x4*)(mem + (size_t)(t + u) * a.enc_dim);
#pragma unroll
                for (int u = 0; u < 8; ++u) acc += mv[u] * s_e[t + u];
            }
            for (; t < te; ++t) acc += *(const f32x4*)(mem + (size_t)t * a.enc_dim) * s_e[t];
        }
        float* s_c8 = s_f;                                      // [8][512] scratch (the location features are dead by now)
        *(f32x4*)(s_c8 + sl * 512 + 4 * cq) = acc;
    }
    __syncthreads();
    if (tid < a.enc_dim && tid < 512) {
        float v = 0.f;
#pragma unroll
        for (int sl = 0; sl < 8; ++sl) v += s_f[sl * 512 + tid];
        a.ctx[(size_t)b * a.enc_dim + tid] = v;
        if (a.ctx_copy) a.ctx_copy[(size_t)b * a.s_ctx_copy + tid] = v;
    }
    PROBE_END()
}

static bool att_fused_mfma_form(const AttArgs& a) {
    static const bool no_mfma = getenv("T2S_ATT_VALU") != nullptr;       // A/B switch: the VALU form
    return a.loc_f == 32 && a.att_dim == 128 && !no_mfma;
}

// Workgroups of the gate-stream role for a launch with B attention workgroups on this device: one workgroup per CU, and the role's
// ONE pass needs 3 or 4 row units per wave (3 * 16 * blocks <= n_units <= 4 * 16 * blocks).  0: this device has too few CUs - the
// caller keeps the plain chain.
static int gate_stream_blocks(int B, int n_units) {
    static const int n_cu = [] {
        int dev = 0, n = 256;
        if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
        return n > 16 ? n : 256;
    }();
    int blocks = n_cu - B;
    if (3 * 16 * blocks > n_units) blocks = n_units / 48;      // more CUs than the pass has units for: leave the rest idle
    if (blocks <= 0 || 4 * 16 * blocks < n_units) return 0;
    return blocks;
}

// the gate-stream role rides on the matrix-core form of the fused attention launch only; three [4H][H] blocks, H = 1024
bool t2s_att_fused_stream_ok(const AttArgs& a, const GateStreamArgs& g) {
    if (!att_fused_mfma_form(a) || a.B > 8 || a.T > ATT_FUSED_MAXT) return false;
    if (gate_stream_blocks(a.B, 3 * g.rows) == 0) return false;
    if (g.H != 1024 || g.rows != 4 * g.H || g.B != a.B) return false;
    const void* ptrs[] = {g.W0, g.W1, g.W2, g.x0, g.x12, g.out0, g.out1, g.out2};
    for (const void* p : ptrs)
        if (!p || ((uintptr_t)p & 15)) return false;
    if (g.ld0 < g.H || g.ld1 < g.H || g.ld2 < g.H || ((g.ld0 | g.ld1 | g.ld2) & 3)) return false;
    return true;
}

hipError_t t2s_launch_att_fused(const AttArgs& a, hipStream_t stream, const GateStreamArgs* gs) {
    if (a.T > ATT_FUSED_MAXT || a.enc_dim > 512 || a.att_dim > 128 || a.loc_f > 32 || a.loc_ks > 63 || a.att_rnn > 1024)
        return hipErrorInvalidValue;
    if (gs && !t2s_att_fused_stream_ok(a, *gs)) return hipErrorInvalidValue;
    if (att_fused_mfma_form(a)) {
        const int Tp = (a.T + 15) & ~15, KP = (2 * a.loc_ks + 3) & ~3;
        const size_t nF = (size_t)Tp * 36 > 4096 ? (size_t)Tp * 36 : 4096;      // (FS = 36 floats per feature row in the kernel)
        const size_t lds = (nF + Tp + 8 * Tp + (size_t)KP * 48) * sizeof(float);
        if (gs) {
            // one workgroup per CU (1024 threads, > 64 KB of LDS at any T): the attention workgroups take B CUs, the gate-stream
            // role the rest of the chip in ONE round
            static std::atomic<unsigned long long> attr_mask3{0};
            const hipError_t e3 = t2s_raise_lds_limit((const void*)att_fused_mfma_kernel<true>, 120 * 1024, attr_mask3);
            if (e3 != hipSuccess) return e3;
            const int blocks = gate_stream_blocks(a.B, 3 * gs->rows);
            if (blocks == 0) return hipErrorInvalidValue;
            const size_t lds1 = lds > 84 * 1024 ? lds : 84 * 1024;      // more than half a CU's LDS: never two workgroups on one CU
            GateStreamArgs g2 = *gs;
#ifdef T2S_ATTSTREAM_ABLATE
            static const int dbg = getenv("T2S_DBG_ATTSTREAM") ? atoi(getenv("T2S_DBG_ATTSTREAM")) : 0;
            g2.dbg = dbg;
#endif
            if (a.ploc) {
                if (!a.q_part || (a.n_part & 31) || a.n_part > 256) return hipErrorInvalidValue;
                static std::atomic<unsigned long long> attr_mask4{0};
                const hipError_t e4 = t2s_raise_lds_limit((const void*)att_fused_mfma_kernel<true, true>, 120 * 1024, attr_mask4);
                if (e4 != hipSuccess) return e4;
                hipLaunchKernelGGL((att_fused_mfma_kernel<true, true>), dim3(a.B + blocks), dim3(1024), lds1, stream, a, g2);
                return hipGetLastError();
            }
            hipLaunchKernelGGL(att_fused_mfma_kernel<true>, dim3(a.B + blocks), dim3(1024), lds1, stream, a, g2);
            return hipGetLastError();
        }
        static std::atomic<unsigned long long> attr_mask2{0};
        const hipError_t e2 = t2s_raise_lds_limit((const void*)att_fused_mfma_kernel<false>, 120 * 1024, attr_mask2);
        if (e2 != hipSuccess) return e2;
        if (a.ploc) return hipErrorInvalidValue;            // (the precomputed location term rides with the gate-stream form only)
        hipLaunchKernelGGL(att_fused_mfma_kernel<false>, dim3(a.B), dim3(1024), lds, stream, a, GateStreamArgs{});
        return hipGetLastError();
    }
    const size_t lds = ((size_t)a.T * 33 + a.T) * sizeof(float);
    static std::atomic<unsigned long long> attr_mask{0};
    const hipError_t e = t2s_raise_lds_limit((const void*)att_fused_kernel, 96 * 1024, attr_mask);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(att_fused_kernel, dim3(a.B), dim3(1024), lds, stream, a);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// Encoder BiLSTM recurrence with packed-sequence semantics (reference tacotron.py:199-207; nn.LSTM
// bidirectional, H per direction).  gx[b][t][dir*4H + j] already holds W_ih x + b_ih + b_hh (one GEMM
// for all steps).  One workgroup = one direction x BT batch elements; thread j owns gate row j and
// streams W_hh^T (coalesced) each step; h lives in LDS.
// Round 4: part of the matrix stays ON the CU for the whole sequence - thread j keeps W_hh^T[k][j] for k < KR in registers and the
// workgroup keeps rows KR .. KR + KL in LDS (KL * 4 KB), so a step streams H - KR - KL rows instead of H: the kernel is bound by what
// one CU pulls out of L2 (1 MB per step and workgroup before: 14 us per step at B = 32, 8.6 at B = 1).
template <int BT, int KR, int KL>
__global__ __launch_bounds__(1024) void lstm_seq_kernel(const float* __restrict__ gx, const float* __restrict__ whhT_f,
                                                        const float* __restrict__ whhT_r, const int* __restrict__ lengths,
                                                        float* out, int B, int T, int H, int T_out, float* gates_save,
                                                        float* c_save) {
    __shared__ __attribute__((aligned(16))) float s_h[BT][256];
    __shared__ float s_g[BT][1024];
    const int j = threadIdx.x;              // gate row, 4H == blockDim.x
    const int dir = blockIdx.y;
    const int b0 = blockIdx.x * BT;
    const float* WT = dir ? whhT_r : whhT_f;     // [H][4H]
    int len[BT];
    int maxlen = 0;
#pragma unroll
    for (int i = 0; i < BT; ++i) {
        len[i] = (b0 + i < B) ? (lengths ? lengths[b0 + i] : T) : 0;
        maxlen = max(maxlen, len[i]);
    }
    float c[BT];
#pragma unroll
    for (int i = 0; i < BT; ++i) c[i] = 0.f;
    if (j < H)
#pragma unroll
        for (int i = 0; i < BT; ++i) s_h[i][j] = 0.f;
    // resident part of W_hh^T: rows [0, KR) in registers, rows [KR, KR + KL) in LDS (dynamic, KL * 1024 floats)
    extern __shared__ __attribute__((aligned(16))) float s_w[];
    float wr[KR > 0 ? KR : 1];
#pragma unroll
    for (int k = 0; k < KR; ++k) wr[k] = WT[(size_t)k * 4 * H + j];
#pragma unroll 8
    for (int kk = 0; kk < KL; ++kk) s_w[kk * 1024 + j] = WT[(size_t)(KR + kk) * 4 * H + j];
    __syncthreads();
    for (int s = 0; s < maxlen; ++s) {
        float acc[BT];
#pragma unroll
        for (int i = 0; i < BT; ++i) {
            const int t = dir ? len[i] - 1 - s : s;
            acc[i] = (s < len[i]) ? gx[((size_t)(b0 + i) * T + t) * (8 * H) + dir * 4 * H + j] : 0.f;
        }
        // four k per trip: one 16-byte broadcast LDS read per item instead of four 4-byte ones (the loop was LDS-issue
        // bound: 1024 ds_read per thread per step), four independent weight loads in flight
        // (the streamed rows first: their loads are in flight while the resident rows are multiplied)
#pragma unroll 4
        for (int k = KR + KL; k < H; k += 4) {
            const float w0 = WT[(size_t)k * 4 * H + j], w1 = WT[(size_t)(k + 1) * 4 * H + j];
            const float w2 = WT[(size_t)(k + 2) * 4 * H + j], w3 = WT[(size_t)(k + 3) * 4 * H + j];
#pragma unroll
            for (int i = 0; i < BT; ++i) {
                const f32x4 h4 = *(const f32x4*)&s_h[i][k];
                acc[i] += (w0 * h4[0] + w1 * h4[1]) + (w2 * h4[2] + w3 * h4[3]);
            }
        }
#pragma unroll
        for (int k = 0; k < KR; k += 4) {
#pragma unroll
            for (int i = 0; i < BT; ++i) {
                const f32x4 h4 = *(const f32x4*)&s_h[i][k];
                acc[i] += (wr[k] * h4[0] + wr[k + 1] * h4[1]) + (wr[k + 2] * h4[2] + wr[k + 3] * h4[3]);
            }
            if ((k & 15) == 12) __builtin_amdgcn_sched_barrier(0);      // (keeps hipcc from hoisting every LDS read of h in front: spills)
        }
#pragma unroll 4
        for (int kk = 0; kk < KL; kk += 4) {
            const float w0 = s_w[kk * 1024 + j], w1 = s_w[(kk + 1) * 1024 + j], w2 = s_w[(kk + 2) * 1024 + j], w3 = s_w[(kk + 3) * 1024 + j];
#pragma unroll
            for (int i = 0; i < BT; ++i) {
                const f32x4 h4 = *(const f32x4*)&s_h[i][KR + kk];
                acc[i] += (w0 * h4[0] + w1 * h4[1]) + (w2 * h4[2] + w3 * h4[3]);
            }
        }
#pragma unroll
        for (int i = 0; i < BT; ++i) s_g[i][j] = acc[i];
        __syncthreads();
        if (j < H) {
#pragma unroll
            for (int i = 0; i < BT; ++i) {
                if (s < len[i]) {
                    const float gi = s_g[i][j], gf = s_g[i][H + j], gg = s_g[i][2 * H + j], go = s_g[i][3 * H + j];
                    c[i] = sigmoid_acc(gf) * c[i] + sigmoid_acc(gi) * tanhf(gg);
                    const float h = sigmoid_acc(go) * tanhf(c[i]);
                    s_h[i][j] = h;
                    const int t = dir ? len[i] - 1 - s : s;
                    if (gates_save) {      // training: post-activation gates and cell state per step, for the BPTT kernel
                        const size_t gb = (((size_t)(b0 + i) * T + t) * 2 + dir) * 4 * H + j;
                        gates_save[gb] = sigmoid_acc(gi);
                        gates_save[gb + H] = sigmoid_acc(gf);
                        gates_save[gb + 2 * H] = tanhf(gg);
                        gates_save[gb + 3 * H] = sigmoid_acc(go);
                        c_save[(((size_t)(b0 + i) * T + t) * 2 + dir) * H + j] = c[i];
                    }
                    out[((size_t)(b0 + i) * T_out + t) * (2 * H) + dir * H + j] = h;
                }
            }
        }
        __syncthreads();
    }
    // zero the padded tail (pad_packed_sequence semantics)
    if (j < H)
#pragma unroll
        for (int i = 0; i < BT; ++i)
            if (b0 + i < B)
                for (int t = len[i]; t < T_out; ++t) out[((size_t)(b0 + i) * T_out + t) * (2 * H) + dir * H + j] = 0.f;
}

hipError_t t2s_launch_lstm_seq(const float* gx, const float* whhT_f, const float* whhT_r, const int* lengths, float* out,
                               int B, int T, int H, int T_out, float* gates_save, float* c_save, hipStream_t stream) {
    if (4 * H != 1024) return hipErrorInvalidValue;
    // elements per workgroup: the recurrent matrix (1 MB per direction) is re-streamed by every workgroup each step, the
    // FMAs scale with the elements it carries - 2 per workgroup up to 128 elements (<= 128 workgroups), then 4
    // resident rows (T2S_LSTM_SEQ_RESIDENT=0: none, the round-3 kernel): 64 / 24 / 16 in registers (by elements per workgroup: the
    // 128-VGPR cap of a 1024-thread workgroup) + 36 / 36 / 32 in LDS (144 / 128 KB) of the 256
    static const bool resident = !(getenv("T2S_LSTM_SEQ_RESIDENT") && atoi(getenv("T2S_LSTM_SEQ_RESIDENT")) == 0);
#define LSEQ(BT_, KR_, KL_, GRID)                                                                                      \
    do {                                                                                                               \
        static std::atomic<unsigned long long> am{0};                                                                  \
        const hipError_t e_ = t2s_raise_lds_limit((const void*)lstm_seq_kernel<BT_, KR_, KL_>, KL_ * 4096 + 16, am);   \
        if (e_ != hipSuccess) return e_;                                                                               \
        hipLaunchKernelGGL((lstm_seq_kernel<BT_, KR_, KL_>), GRID, dim3(1024), KL_ * 4096, stream, gx, whhT_f, whhT_r, \
                           lengths, out, B, T, H, T_out, gates_save, c_save);                                          \
    } while (0)
    if (B > 128) {
        if (resident) LSEQ(4, 16, 32, dim3((B + 3) / 4, 2)); else LSEQ(4, 0, 0, dim3((B + 3) / 4, 2));
    } else if (B >= 8) {
        if (resident) LSEQ(2, 24, 36, dim3((B + 1) / 2, 2)); else LSEQ(2, 0, 0, dim3((B + 1) / 2, 2));
    } else {
        if (resident) LSEQ(1, 64, 36, dim3(B, 2)); else LSEQ(1, 0, 0, dim3(B, 2));
    }
#undef LSEQ
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// The same recurrence with W_hh RESIDENT: the matrix of one direction (1 MB) does not fit one CU, a quarter of it (256 KB = 64
// registers on each of 1024 threads) does.  Four workgroups share one (batch element, direction): workgroup q owns hidden units
// 64 q .. 64 q + 63 (their four gate rows each), keeps W_hh^T[k][those 256 rows] in registers for the whole sequence and per step
// exchanges its 64 new h values with the other three through 8-byte {value, tag} granules (MI355X_MICROARCH.md handoff-1to1 /
// "R2's granule": one sc1 store per value, the consumer lane polls its own granule with an sc1 load until the tag is this launch's
// step number - no flag, no fence, nothing streamed).  A step is then one hand-off (~1 us) + 64 FMAs per thread instead of 1 MB
// through one CU (12.7 us per step at B = 32).
//   * xbuf: [2 B groups][2 step parities][256 units] granules, caller-owned, zero before its first use; tags carry `epoch` (the
//     caller's launch counter) so that nothing a previous launch left can match.  Slot parity: a workgroup overwrites slot s & 1 at
//     step s + 2, after it has seen every partner's step s + 1, which they publish after reading step s.
//   * block number -> (group, quarter): id = 32 G + 8 m + x is quarter m of group 8 G + x - the four workgroups of a group are
//     within 32 consecutive blocks (they become resident together; a group whose partners are not resident yet spins, complete
//     groups in front of it always finish) and share an XCD under round-robin placement (speed only).
//   * every wait is bounded (LSEQ_SPIN_MAX polls, seconds): on expiry the workgroup raises xbuf's error word and leaves the loop.
#define LSEQ_SPIN_MAX (1 << 22)
static __device__ __forceinline__ void lseq_publish(unsigned long long* slot, float v, unsigned tag) {
    const unsigned long long g = ((unsigned long long)tag << 32) | (unsigned long long)__float_as_uint(v);
    __hip_atomic_store(slot, g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);         // global_store_dwordx2 sc1
}
// polls until the granule carries `tag`; false on expiry
static __device__ __forceinline__ bool lseq_await(const unsigned long long* slot, unsigned tag, float& v) {
    for (int it = 0; it < LSEQ_SPIN_MAX; ++it) {
        const unsigned long long g = __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);    // global_load_dwordx2 sc1
        if ((unsigned)(g >> 32) == tag) { v = __uint_as_float((unsigned)g); return true; }
        __builtin_amdgcn_s_sleep(1);
    }
    return false;
}

__global__ __launch_bounds__(1024) void lstm_seq_split_kernel(const float* __restrict__ gx, const float* __restrict__ whhT_f,
                                                              const float* __restrict__ whhT_r, const int* __restrict__ lengths,
                                                              float* out, int B, int T, int T_out, float* gates_save, float* c_save,
                                                              unsigned long long* xbuf, unsigned epoch) {
    constexpr int H = 256;
    __shared__ __attribute__((aligned(16))) float s_h[H];
    __shared__ float s_part[4][256];
    __shared__ float s_g[256];
    __shared__ int s_fail;
    const int tid = threadIdx.x;
    const int id = blockIdx.x, group = (id >> 5) * 8 + (id & 7), q = (id >> 3) & 3;
    if (group >= 2 * B) return;                              // (whole workgroups)
    const int b = group >> 1, dir = group & 1;
    const int r = tid & 255, g = r >> 6, ul = r & 63, kq = tid >> 8;
    const int j = g * H + 64 * q + ul;                       // this thread's gate row
    const float* WT = dir ? whhT_r : whhT_f;                 // [H][4H]
    const int len = lengths ? lengths[b] : T;
    float wr[64];
#pragma unroll
    for (int i = 0; i < 64; ++i) wr[i] = WT[(size_t)(64 * kq + i) * 4 * H + j];
    if (tid < H) s_h[tid] = 0.f;
    if (tid == 0) s_fail = 0;
    float c = 0.f;
    unsigned long long* xg = xbuf + (size_t)group * 2 * H;
    const unsigned tag0 = epoch << 12;
    // the partner unit this thread fetches each step (threads 64 .. 255: the 192 units of the other three quarters)
    const int pu = tid >= 64 && tid < 256 ? ((tid - 64) < 64 * q ? (tid - 64) : tid) : 0;
    __syncthreads();
    float gxv = 0.f;
    if (tid < 256 && len > 0) gxv = gx[((size_t)b * T + (dir ? len - 1 : 0)) * (8 * H) + dir * 4 * H + j];
    for (int s = 0; s < len; ++s) {
        const int t = dir ? len - 1 - s : s;
        float acc = 0.f;
#pragma unroll
        for (int i = 0; i < 64; i += 4) {
            const f32x4 h4 = *(const f32x4*)&s_h[64 * kq + i];
            acc += (wr[i] * h4[0] + wr[i + 1] * h4[1]) + (wr[i + 2] * h4[2] + wr[i + 3] * h4[3]);
        }
        s_part[kq][r] = acc;
        __syncthreads();
        if (tid < 256) {
            s_g[tid] = gxv + ((s_part[0][tid] + s_part[1][tid]) + (s_part[2][tid] + s_part[3][tid]));
            if (s + 1 < len)                                 // next step's input projection: in flight across the hand-off
                gxv = gx[((size_t)b * T + (dir ? len - 2 - s : s + 1)) * (8 * H) + dir * 4 * H + j];
        }
        __syncthreads();
        if (tid < 64) {
            const int u = 64 * q + tid;
            const float gi = s_g[tid], gf = s_g[64 + tid], gg = s_g[128 + tid], go = s_g[192 + tid];
            c = sigmoid_acc(gf) * c + sigmoid_acc(gi) * tanhf(gg);
            const float h = sigmoid_acc(go) * tanhf(c);
            lseq_publish(xg + (size_t)(s & 1) * H + u, h, tag0 + (unsigned)s + 1u);
            s_h[u] = h;
            if (gates_save) {      // training: post-activation gates and cell state per step, for the BPTT kernel
                const size_t gb = (((size_t)b * T + t) * 2 + dir) * 4 * H + u;
                gates_save[gb] = sigmoid_acc(gi);
                gates_save[gb + H] = sigmoid_acc(gf);
                gates_save[gb + 2 * H] = tanhf(gg);
                gates_save[gb + 3 * H] = sigmoid_acc(go);
                c_save[(((size_t)b * T + t) * 2 + dir) * H + u] = c;
            }
            out[((size_t)b * T_out + t) * (2 * H) + dir * H + u] = h;
        } else if (tid < 256) {
            float v = 0.f;
            const bool ok = s + 1 < len ? lseq_await(xg + (size_t)(s & 1) * H + pu, tag0 + (unsigned)s + 1u, v) : true;
            if (!ok) s_fail = 1;
            s_h[pu] = v;                                     // (the last step's h of the partners is not needed)
        }
        __syncthreads();
        if (s_fail) {
            if (tid == 0) __hip_atomic_store(xbuf + (size_t)2 * B * 2 * H, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            break;
        }
    }
    // zero the padded tail (pad_packed_sequence semantics)
    if (tid < 64)
        for (int t = len; t < T_out; ++t) out[((size_t)b * T_out + t) * (2 * H) + dir * H + 64 * q + tid] = 0.f;
}

hipError_t t2s_launch_lstm_seq_split(const float* gx, const float* whhT_f, const float* whhT_r, const int* lengths, float* out,
                                     int B, int T, int T_out, float* gates_save, float* c_save, unsigned long long* xbuf,
                                     unsigned epoch, hipStream_t stream) {
    if (T >= 4095) return hipErrorInvalidValue;              // (12 tag bits for the step)
    hipLaunchKernelGGL(lstm_seq_split_kernel, dim3(((2 * B + 7) / 8) * 32), dim3(1024), 0, stream, gx, whhT_f, whhT_r, lengths, out,
                       B, T, T_out, gates_save, c_save, xbuf, epoch & 0xFFFFFu);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// Pacing a helper stream by a device word instead of events (an event record between two launches of the serial chain opens a
// ~7 us gap on it; a word stored by the first thread of a chain launch costs nothing there).  One wave polls, bounded.
__global__ void pace_wait_kernel(const unsigned* flag, unsigned val, unsigned long long* err) {
    if (threadIdx.x != 0) return;
    for (int it = 0; it < (1 << 22); ++it) {
        const unsigned v = __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((int)(v - val) >= 0) return;
        __builtin_amdgcn_s_sleep(2);
    }
    if (err) __hip_atomic_store(err, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__global__ void pace_signal_kernel(unsigned* flag, unsigned val) {
    if (threadIdx.x == 0) __hip_atomic_store(flag, val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
hipError_t t2s_launch_pace_wait(const unsigned* flag, unsigned val, unsigned long long* err, hipStream_t stream) {
    hipLaunchKernelGGL(pace_wait_kernel, dim3(1), dim3(64), 0, stream, flag, val, err);
    return hipGetLastError();
}
hipError_t t2s_launch_pace_signal(unsigned* flag, unsigned val, hipStream_t stream) {
    hipLaunchKernelGGL(pace_signal_kernel, dim3(1), dim3(64), 0, stream, flag, val);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// small helpers
__global__ void transpose_kernel(const float* in, float* out, int R, int C) {      // out[c][r] = in[r][c]
    __shared__ float tile[32][33];
    const int c = blockIdx.x * 32 + threadIdx.x, r0 = blockIdx.y * 32;
    for (int i = threadIdx.y; i < 32; i += 8)
        if (r0 + i < R && c < C) tile[i][threadIdx.x] = in[(size_t)(r0 + i) * C + c];
    __syncthreads();
    const int r = r0 + threadIdx.x, c0 = blockIdx.x * 32;
    for (int i = threadIdx.y; i < 32; i += 8)
        if (c0 + i < C && r < R) out[(size_t)(c0 + i) * R + r] = tile[threadIdx.x][i];
}
hipError_t t2s_launch_transpose(const float* in, float* out, int R, int C, hipStream_t stream) {
    hipLaunchKernelGGL(transpose_kernel, dim3((C + 31) / 32, (R + 31) / 32), dim3(32, 8), 0, stream, in, out, R, C);
    return hipGetLastError();
}

// Embedding lookup written straight into (hi, lo) planes: x[b][:, t] = emb[ids[b][t]]  (tacotron.py:40)
__global__ void embed_planes_kernel(const long* ids, const float* emb, int B, int T, int E, int V, int Lp, int halo,
                                    u16* X_hi, u16* X_lo) {
    const int t = blockIdx.x, b = blockIdx.y;
    long id = ids[(size_t)b * T + t];
    if (id < 0 || id >= V) id = 0;
    const int nch = (E + 31) / 32;
    for (int e = threadIdx.x; e < nch * 32; e += blockDim.x) {
        const float v = e < E ? emb[(size_t)id * E + e] : 0.f;
        u16 h, l;
        split_bf16(v, h, l);
        const size_t idx = (((size_t)b * nch + (e >> 5)) * Lp + halo + t) * 32 + (e & 31);
        X_hi[idx] = h;
        X_lo[idx] = l;
    }
}
hipError_t t2s_launch_embed_planes(const long* ids, const float* emb, int B, int T, int E, int V, int Lp, int halo,
                                   u16* X_hi, u16* X_lo, hipStream_t stream) {
    hipLaunchKernelGGL(embed_planes_kernel, dim3(T, B), dim3(256), 0, stream, ids, emb, B, T, E, V, Lp, halo, X_hi, X_lo);
    return hipGetLastError();
}

// [B][C][L] f32 -> planes (the decoder's mel output feeding the postnet, modules.py:131)
__global__ void f32_to_planes_kernel(const float* x, int C, int L, int Lp, int halo, u16* X_hi, u16* X_lo) {
    const int t = blockIdx.x * 64 + (threadIdx.x & 63);
    const int b = blockIdx.z;
    const int nch = (C + 31) / 32;
    if (t >= L) return;
    for (int c = blockIdx.y * 32 + (threadIdx.x >> 6); c < blockIdx.y * 32 + 32; c += 4) {
        const float v = c < C ? x[((size_t)b * C + c) * L + t] : 0.f;
        u16 h, l;
        split_bf16(v, h, l);
        const size_t idx = (((size_t)b * nch + (c >> 5)) * Lp + halo + t) * 32 + (c & 31);
        X_hi[idx] = h;
        X_lo[idx] = l;
    }
}
hipError_t t2s_launch_f32_to_planes(const float* x, int B, int C, int L, int Lp, int halo, u16* X_hi, u16* X_lo,
                                    hipStream_t stream) {
    hipLaunchKernelGGL(f32_to_planes_kernel, dim3((L + 63) / 64, (C + 31) / 32, B), dim3(256), 0, stream, x, C, L, Lp,
                       halo, X_hi, X_lo);
    return hipGetLastError();
}

// Tacotron.parse_output (reference tacotron.py:67-76): beyond each entry's output length the two mel tensors become 0 and the
// gate energies 1e3, in place, one launch.  Row n_mel of the grid is the gate row.
__global__ void parse_output_kernel(float* mel, float* mel_post, float* gate, const int* lengths, int n_mel, int T) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int c = blockIdx.y, b = blockIdx.z;
    if (t >= T || t < lengths[b]) return;
    if (c < n_mel) {
        const size_t i = ((size_t)b * n_mel + c) * T + t;
        mel[i] = 0.f;
        mel_post[i] = 0.f;
    } else {
        gate[(size_t)b * T + t] = 1e3f;
    }
}
hipError_t t2s_launch_parse_output(float* mel, float* mel_post, float* gate, const int* lengths, int B, int n_mel, int T,
                                   hipStream_t stream) {
    hipLaunchKernelGGL(parse_output_kernel, dim3((T + 255) / 256, n_mel + 1, B), dim3(256), 0, stream, mel, mel_post, gate,
                       lengths, n_mel, T);
    return hipGetLastError();
}

// eval-mode BatchNorm folded into the preceding conv: scale[o] = gamma/sqrt(var+eps),
// bias'[o] = (bias[o] - mean[o]) * scale[o] + beta[o]      (tacotron.py:183-184, modules.py:105-129)
__global__ void bn_fold_kernel(const float* gamma, const float* beta, const float* mean, const float* var,
                               const float* conv_bias, float eps, int C, float* scale, float* bias_out) {
    const int o = blockIdx.x * blockDim.x + threadIdx.x;
    if (o >= C) return;
    const float s = gamma[o] / sqrtf(var[o] + eps);
    scale[o] = s;
    bias_out[o] = ((conv_bias ? conv_bias[o] : 0.f) - mean[o]) * s + beta[o];
}
hipError_t t2s_launch_bn_fold(const float* gamma, const float* beta, const float* mean, const float* var,
                              const float* conv_bias, float eps, int C, float* scale, float* bias_out,
                              hipStream_t stream) {
    hipLaunchKernelGGL(bn_fold_kernel, dim3((C + 255) / 256), dim3(256), 0, stream, gamma, beta, mean, var, conv_bias,
                       eps, C, scale, bias_out);
    return hipGetLastError();
}

// Bernoulli(0.5) prenet masks from a counter hash (used when the caller does not inject masks)
__global__ void bernoulli_mask_kernel(unsigned char* mask, size_t n, unsigned long long seed, unsigned long long offset,
                                      unsigned int keep_thr) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    unsigned long long x = (i + offset) * 0x9E3779B97F4A7C15ull + seed;
    x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 27; x *= 0x94D049BB133111EBull; x ^= x >> 31;
    mask[i] = (unsigned char)(((unsigned int)(x >> 40)) < keep_thr);      // 24 random bits vs keep probability
}
hipError_t t2s_launch_bernoulli_mask(unsigned char* mask, size_t n, unsigned long long seed, unsigned long long offset,
                                     float keep_prob, hipStream_t stream) {
    const unsigned int thr = (unsigned int)(keep_prob * 16777216.0f);
    hipLaunchKernelGGL(bernoulli_mask_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, mask, n, seed, offset, thr);
    return hipGetLastError();
}

// first step (per batch element) at which sigmoid(gate) > threshold (reference tacotron.py:455)
// One wave per batch element, the steps of the chunk on the lanes (64 at a time, all loads of a pass in flight together - the
// first version walked them with one thread, 64 dependent loads: 11 us per poll on the decode loop's critical path), first set
// bit of the ballot.
__global__ __launch_bounds__(64) void stop_check_kernel(const float* gate_out, int B, int s_gate_b, int step0, int n, float threshold,
                                                        int* stop_step) {
    const int b = blockIdx.x, lane = threadIdx.x;
    if (stop_step[b] >= 0) return;
    for (int s0 = step0; s0 < step0 + n; s0 += 64) {
        const int s = s0 + lane;
        const bool in = s < step0 + n;
        const float g = gate_out[(size_t)b * s_gate_b + (in ? s : step0)];
        const unsigned long long hit = __ballot(in && 1.0f / (1.0f + expf(-g)) > threshold);
        if (hit) {
            if (lane == 0) stop_step[b] = s0 + (int)__builtin_ctzll(hit);
            return;
        }
    }
}
hipError_t t2s_launch_stop_check(const float* gate_out, int B, int s_gate_b, int step0, int n, float threshold,
                                 int* stop_step, hipStream_t stream) {
    hipLaunchKernelGGL(stop_check_kernel, dim3(B), dim3(64), 0, stream, gate_out, B, s_gate_b, step0, n, threshold, stop_step);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// Training-mode BatchNorm1d after a conv (reference tacotron.py:183-184,193-194; modules.py:105-137):
// batch statistics over (B, T) per channel, affine, activation, dropout mask, result written as planes.
// x: [B][C][T] f32 (the conv output).  stats[c] = (mean, biased var).
__global__ __launch_bounds__(256) void bn_stats_kernel(const float* __restrict__ x, int B, int C, int T, float* mean,
                                                       float* var) {
    __shared__ double red[2][4];
    const int c = blockIdx.x, tid = threadIdx.x;
    double s = 0.0, ss = 0.0;
    for (int b = 0; b < B; ++b) {
        const float* row = x + ((size_t)b * C + c) * T;
        for (int t = tid; t < T; t += 256) {
            const double v = row[t];
            s += v;
            ss += v * v;
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        s += __shfl_xor(s, off, 64);
        ss += __shfl_xor(ss, off, 64);
    }
    if ((tid & 63) == 0) { red[0][tid >> 6] = s; red[1][tid >> 6] = ss; }
    __syncthreads();
    if (tid == 0) {
        const double n = (double)B * T;
        const double m = (red[0][0] + red[0][1] + red[0][2] + red[0][3]) / n;
        const double q = (red[1][0] + red[1][1] + red[1][2] + red[1][3]) / n;
        mean[c] = (float)m;
        var[c] = (float)(q - m * m);
    }
}
// nn.BatchNorm1d's training-mode bookkeeping: running = (1 - momentum) running + momentum batch (the variance unbiased, n / (n - 1)),
// num_batches_tracked += 1 - one launch instead of six [C]-sized eager operators per layer
__global__ void bn_running_update_kernel(const float* mean, const float* var, float* running_mean, float* running_var,
                                         long long* num_batches_tracked, float momentum, float unbias, int C) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c < C) {
        running_mean[c] = running_mean[c] * (1.f - momentum) + momentum * mean[c];
        running_var[c] = running_var[c] * (1.f - momentum) + momentum * (var[c] * unbias);
    }
    if (c == 0 && num_batches_tracked) *num_batches_tracked += 1;
}
hipError_t t2s_launch_bn_running_update(const float* mean, const float* var, float* running_mean, float* running_var,
                                        long long* num_batches_tracked, float momentum, float unbias, int C, hipStream_t stream) {
    hipLaunchKernelGGL(bn_running_update_kernel, dim3((C + 255) / 256), dim3(256), 0, stream, mean, var, running_mean,
                       running_var, num_batches_tracked, momentum, unbias, C);
    return hipGetLastError();
}
// clears n16 16-byte pieces (+ tail bytes): the one fill of a step's accumulator arena
__global__ void zero_fill_kernel(uint4* p, size_t n16, unsigned char* tail, int ntail) {
    const uint4 z = {0u, 0u, 0u, 0u};
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) p[i] = z;
    if (blockIdx.x == 0 && (int)threadIdx.x < ntail) tail[threadIdx.x] = 0;
}
hipError_t t2s_launch_zero_fill(void* p, size_t bytes, hipStream_t stream) {
    const size_t n16 = bytes / 16;
    const int ntail = (int)(bytes - n16 * 16);
    size_t blocks = (n16 + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    if (blocks == 0) blocks = 1;
    hipLaunchKernelGGL(zero_fill_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, (uint4*)p, n16, (unsigned char*)p + n16 * 16, ntail);
    return hipGetLastError();
}
// y = act((x - mean) / sqrt(var + eps) * gamma + beta) * mask * mask_scale -> planes (and optional f32 copy)
// Tile = 64 time steps x one 32-channel chunk, 256 threads: x is read time-major (thread = (channel mod 4, t)), the planes are
// written as 16-byte pieces (thread = (t, 8 channels)) through LDS - they were written 2 bytes per thread at a 64-byte stride.
__global__ __launch_bounds__(256) void bn_apply_kernel(const float* __restrict__ x, const float* __restrict__ mean,
                                                       const float* __restrict__ var, const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, float eps, int act,
                                                       const unsigned char* __restrict__ mask, float mask_scale, int C, int T, int Lp,
                                                       int halo, u16* O_hi, u16* O_lo, float* out_f32) {
    typedef __attribute__((ext_vector_type(8))) unsigned short u16x8;
    __shared__ float s_o[64][33];
    const int t0 = blockIdx.x * 64, chunk = blockIdx.y, b = blockIdx.z, tid = threadIdx.x;
    const int ci = tid >> 6, tl = tid & 63, t = t0 + tl;
    const int nch = (C + 31) / 32;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int cl = ci + 4 * k, c = chunk * 32 + cl;
        float v = 0.f;
        if (c < C && t < T) {
            const size_t i = ((size_t)b * C + c) * T + t;
            v = (x[i] - mean[c]) / sqrtf(var[c] + eps) * gamma[c] + beta[c];
            if (act == ACT_RELU) v = fmaxf(v, 0.f);
            else if (act == ACT_TANH) v = tanhf(v);
            if (mask) v = mask[i] ? v * mask_scale : 0.f;
            if (out_f32) out_f32[i] = v;
        }
        s_o[tl][cl] = v;
    }
    if (!O_hi) return;                               // (uniform)
    __syncthreads();
    const int tw = tid >> 2, q = tid & 3;
    if (t0 + tw < T) {
        u16x8 h, l;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            u16 hh, ll;
            split_bf16(s_o[tw][q * 8 + e], hh, ll);
            h[e] = hh;
            l[e] = ll;
        }
        const size_t idx = (((size_t)b * nch + chunk) * Lp + halo + t0 + tw) * 32 + q * 8;
        *(u16x8*)(O_hi + idx) = h;
        *(u16x8*)(O_lo + idx) = l;
    }
}
hipError_t t2s_launch_bn_train(const float* x, const float* gamma, const float* beta, float eps, int act,
                               const unsigned char* mask, float mask_scale, int B, int C, int T, int Lp, int halo,
                               float* mean, float* var, u16* O_hi, u16* O_lo, float* out_f32, hipStream_t stream) {
    hipLaunchKernelGGL(bn_stats_kernel, dim3(C), dim3(256), 0, stream, x, B, C, T, mean, var);
    hipLaunchKernelGGL(bn_apply_kernel, dim3((T + 63) / 64, (C + 31) / 32, B), dim3(256), 0, stream, x, mean, var, gamma,
                       beta, eps, act, mask, mask_scale, C, T, Lp, halo, O_hi, O_lo, out_f32);
    return hipGetLastError();
}
