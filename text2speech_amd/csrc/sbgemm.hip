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
static __device__ __forceinline__ void sb_core(const SbOperands& o, size_t grow, bool row_ok, int item_base,
                                               float (*s_part)[2][4][64]) {
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

// tile row r (0..15) of item it (0..31 within the group): D[row][col]: lane = (row/4)*16 + col%16, reg = row%4, half = col/16
static __device__ __forceinline__ float sb_sum(float (*s_part)[2][4][64], int r, int it) {
    const int ln = (r >> 2) * 16 + (it & 15), half = it >> 4, reg = r & 3;
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < 8; ++w) s += s_part[w][half][reg][ln];
    return s;
}

// ---- plain epilogue: GemvArgs semantics (no split_row / masks) -------------------------------------------------------
__global__ __launch_bounds__(512) void sbgemm_plain_kernel(const GemvArgs a) {
    __shared__ float s_part[8][2][4][64];
    SbOperands o;
    o.W1 = a.W1; o.ld1 = a.ld1; o.k1 = a.k1; o.W2 = a.W2; o.ld2 = a.ld2;
    o.x0 = a.x1; o.x1 = a.x2; o.x2 = a.x3; o.n0 = a.n1; o.n1 = a.n2; o.n2 = a.n3;
    o.sx0 = a.sx1; o.sx1 = a.sx2; o.sx2 = a.sx3;
    o.K = a.n1 + a.n2 + a.n3; o.items = a.items;
    const int row0 = blockIdx.x * 16, item_base = blockIdx.y * 32;
    const int lrow = row0 + (threadIdx.x & 15);
    sb_core(o, (size_t)lrow, lrow < a.rows, item_base, s_part);
    const int r = threadIdx.x & 15, it = threadIdx.x >> 4;          // 512 threads = 16 rows x 32 items
    const int row = row0 + r, item = item_base + it;
    if (row < a.rows && item < a.items) {
        float y = sb_sum(s_part, r, it) + (a.bias1 ? a.bias1[row] : 0.f) + (a.bias2 ? a.bias2[row] : 0.f);
        if (a.act == ACT_RELU) y = fmaxf(y, 0.f);
        else if (a.act == ACT_TANH) y = tanhf(y);
        a.y[(size_t)item * a.sy_item + (size_t)row * a.sy_row] = y;
    }
}

bool t2s_sbgemm_plain_ok(const GemvArgs& a) {
    const int K = a.n1 + a.n2 + a.n3;
    if (a.items <= 8 || a.split_row > 0 || a.mask || K < 16) return false;
    if ((K & 15) || (a.n1 & 15) || (a.n2 & 15) || (a.n3 & 15) || (a.k1 & 15)) return false;
    if (a.k1 + a.k2 != K) return false;
    if ((a.ld1 & 3) || (a.W2 && (a.ld2 & 3)) || (a.sx1 & 3) || (a.x2 && (a.sx2 & 3)) || (a.x3 && (a.sx3 & 3))) return false;
    if (((uintptr_t)a.W1 & 15) || ((uintptr_t)a.W2 & 15) || ((uintptr_t)a.x1 & 15) || ((uintptr_t)a.x2 & 15) ||
        ((uintptr_t)a.x3 & 15))
        return false;
    return true;
}

hipError_t t2s_launch_sbgemm_plain(const GemvArgs& a, hipStream_t stream) {
    dim3 grid((a.rows + 15) / 16, (a.items + 31) / 32);
    hipLaunchKernelGGL(sbgemm_plain_kernel, grid, dim3(512), 0, stream, a);
    return hipGetLastError();
}

// ---- fused LSTMCell epilogue: LstmCellArgs semantics ---------------------------------------------------------------
// Workgroup = hidden units u0..u0+3; tile row r = unit*4 + gate  ->  weight row gate*H + u0 + unit.
__global__ __launch_bounds__(512) void sbgemm_lstm_kernel(const LstmCellArgs a) {
    __shared__ float s_part[8][2][4][64];
    SbOperands o;
    const int K1 = a.n1 + a.n2;
    o.W1 = a.W_ih; o.ld1 = K1; o.k1 = K1; o.W2 = a.W_hh; o.ld2 = a.H;
    o.x0 = a.x1; o.x1 = a.x2; o.x2 = a.h_in; o.n0 = a.n1; o.n1 = a.n2; o.n2 = a.H;
    o.sx0 = a.sx1; o.sx1 = a.sx2; o.sx2 = a.H;
    o.K = K1 + a.H; o.items = a.B;
    const int u0 = blockIdx.x * 4, item_base = blockIdx.y * 32;
    const int lr = threadIdx.x & 15;
    sb_core(o, (size_t)(lr & 3) * a.H + u0 + (lr >> 2), true, item_base, s_part);
    if (threadIdx.x < 128) {
        const int ul = threadIdx.x & 3, it = threadIdx.x >> 2;
        const int item = item_base + it, u = u0 + ul;
        if (item < a.B) {
            float g[4];
#pragma unroll
            for (int gi = 0; gi < 4; ++gi)
                g[gi] = sb_sum(s_part, ul * 4 + gi, it) + (a.b_ih[gi * a.H + u] + a.b_hh[gi * a.H + u]);
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
        }
    }
}

bool t2s_sbgemm_lstm_ok(const LstmCellArgs& a) {
    if (a.B <= 8 || (a.H & 15) || (a.n1 & 15) || (a.n2 & 15) || a.q_part) return false;
    if ((a.sx1 & 3) || (a.x2 && (a.sx2 & 3))) return false;
    if (((uintptr_t)a.W_ih & 15) || ((uintptr_t)a.W_hh & 15) || ((uintptr_t)a.x1 & 15) || ((uintptr_t)a.x2 & 15) ||
        ((uintptr_t)a.h_in & 15))
        return false;
    if (a.gates_out && !a.c_out) return false;
    return true;
}

hipError_t t2s_launch_sbgemm_lstm(const LstmCellArgs& a, hipStream_t stream) {
    dim3 grid(a.H / 4, (a.B + 31) / 32);
    hipLaunchKernelGGL(sbgemm_lstm_kernel, grid, dim3(512), 0, stream, a);
    return hipGetLastError();
}
