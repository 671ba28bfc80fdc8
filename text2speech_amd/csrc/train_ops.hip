// Backward-pass and optimizer kernels for the WaveGlow training step (reference waveglow/train.py:110-124:
// zero_grad -> forward -> loss -> backward -> Adam.step; autograd through glow.py:207-249).
//
// The heavy lifting (data gradients and weight gradients of the three WN convolutions) reuses
// conv_gemm_kernel: data gradients are convolutions with transposed / tap-flipped weights
// (pack_transposed_kernel), weight gradients are GEMMs whose contraction index is TIME, fed by
// "time-major" planes  tm[b][t/32][row][t%32]  that plane_transpose_kernel builds from the ordinary
// channel-last planes (one dilated tap = one row shift).  Everything else here is bandwidth-bound glue.
#include "t2s_common.h"
#include "t2s_kernels.h"
#include "train_ops.h"

// ------------------------------------------------------------------------------------------------
// planes [b][c/32][row][32 c]  ->  time-major planes [b][row/32][n_off + c][32 t], rows shifted by `shift`
__global__ __launch_bounds__(256) void plane_transpose_kernel(const u16* __restrict__ src_hi, const u16* __restrict__ src_lo,
                                                              int src_chunks, int Lp, int shift, u16* dst_hi, u16* dst_lo,
                                                              int Npad, int n_off, int n_tchunks) {
    __shared__ u16 th[32][34], tl[32][34];
    const int tc = blockIdx.x, cc = blockIdx.y, b = blockIdx.z;
    const int tid = threadIdx.x;
    {
        const int r = tid >> 3, q = tid & 7;
        const int row = tc * 32 + r + shift;
        u16x4 vh = {0, 0, 0, 0}, vl = {0, 0, 0, 0};
        if (row >= 0 && row < Lp) {
            const size_t idx = (((size_t)b * src_chunks + cc) * Lp + row) * 32 + q * 4;
            vh = *(const u16x4*)(src_hi + idx);
            vl = *(const u16x4*)(src_lo + idx);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) { th[r][q * 4 + e] = vh[e]; tl[r][q * 4 + e] = vl[e]; }
    }
    __syncthreads();
    {
        const int ci = tid >> 3, tq = tid & 7;
        u16x4 vh, vl;
#pragma unroll
        for (int e = 0; e < 4; ++e) { vh[e] = th[tq * 4 + e][ci]; vl[e] = tl[tq * 4 + e][ci]; }
        const size_t idx = (((size_t)b * n_tchunks + tc) * Npad + n_off + cc * 32 + ci) * 32 + tq * 4;
        *(u16x4*)(dst_hi + idx) = vh;
        *(u16x4*)(dst_lo + idx) = vl;
    }
}
hipError_t t2s_launch_plane_transpose(const u16* src_hi, const u16* src_lo, int B, int src_chunks, int n_chunks, int Lp,
                                      int shift, u16* dst_hi, u16* dst_lo, int Npad, int n_off, hipStream_t stream) {
    const int n_tchunks = (Lp + 31) / 32;      // rows beyond Lp read as zero
    hipLaunchKernelGGL(plane_transpose_kernel, dim3(n_tchunks, n_chunks, B), dim3(256), 0, stream, src_hi, src_lo,
                       src_chunks, Lp, shift, dst_hi, dst_lo, Npad, n_off, n_tchunks);
    return hipGetLastError();
}

// a row of ones over the valid time range: its weight-gradient column is the bias gradient
__global__ void tm_ones_row_kernel(u16* dst_hi, u16* dst_lo, int Lp, int halo, int L, int Npad, int n_row) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int b = blockIdx.y;
    if (t >= Lp) return;
    const size_t idx = (((size_t)b * ((Lp + 31) / 32) + (t >> 5)) * Npad + n_row) * 32 + (t & 31);
    dst_hi[idx] = (t >= halo && t < halo + L) ? (u16)0x3F80 : (u16)0;
    dst_lo[idx] = 0;
}
hipError_t t2s_launch_tm_ones_row(u16* dst_hi, u16* dst_lo, int B, int Lp, int halo, int L, int Npad, int n_row,
                                  hipStream_t stream) {
    hipLaunchKernelGGL(tm_ones_row_kernel, dim3((Lp + 255) / 256, B), dim3(256), 0, stream, dst_hi, dst_lo, Lp, halo, L,
                       Npad, n_row);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// Data-gradient weights: A[c][koff + tap' * O_pad + o] = scale[o] * v[o][c][flip ? Kt-1-tap' : tap']
// (the transpose of a conv weight, taps mirrored), split to (hi, lo) planes [k/32][Mpad][32].
// pair8: the M rows (= input channels c of the convolution, output channels of the transposed one) in PERM_PAIR8 order - within
// every 32 channels, packed row 16 m + 4 q + e holds channel 8 q + 4 m + e - for the 16-byte epilogues of the backward GEMMs.
__global__ __launch_bounds__(256) void pack_transposed_kernel(const float* __restrict__ v, const float* __restrict__ scale,
                                                              int O, int Cin, int Kt, int flip, int O_pad, int Mpad,
                                                              int koff, u16* A_hi, u16* A_lo, int pair8) {
    __shared__ float tile[32][33];
    const int o0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
    const int tid = threadIdx.x;
    for (int tp = 0; tp < Kt; ++tp) {
        const int tap = flip ? Kt - 1 - tp : tp;
        {
            const int oi = tid >> 3, cq = tid & 7;
            const int o = o0 + oi;
            const float sc = (o < O) ? (scale ? scale[o] : 1.f) : 0.f;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int c = c0 + cq * 4 + e;
                tile[oi][cq * 4 + e] = (o < O && c < Cin) ? v[((size_t)o * Cin + c) * Kt + tap] * sc : 0.f;
            }
        }
        __syncthreads();
        {
            const int ci = tid >> 3, oq = tid & 7;
            const int c = c0 + ci;
            if (c < Cin) {
                u16x4 vh, vl;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    u16 h, l;
                    split_bf16(tile[oq * 4 + e][ci], h, l);
                    vh[e] = h;
                    vl[e] = l;
                }
                const int k = koff + tp * O_pad + o0 + oq * 4;
                const int w = c & 31;
                const int crow = pair8 ? (c & ~31) + ((w >> 2) & 1) * 16 + (w >> 3) * 4 + (w & 3) : c;
                const size_t idx = ((size_t)(k >> 5) * Mpad + crow) * 32 + (k & 31);
                *(u16x4*)(A_hi + idx) = vh;
                *(u16x4*)(A_lo + idx) = vl;
            }
        }
        __syncthreads();
    }
}
hipError_t t2s_launch_pack_transposed(const float* v, const float* scale, int O, int Cin, int Kt, int flip, int O_pad,
                                      int Mpad, int koff, u16* A_hi, u16* A_lo, int pair8, hipStream_t stream) {
    hipLaunchKernelGGL(pack_transposed_kernel, dim3((O_pad + 31) / 32, (Cin + 31) / 32), dim3(256), 0, stream, v, scale, O,
                       Cin, Kt, flip, O_pad, Mpad, koff, A_hi, A_lo, pair8);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// Reduce the split-K partial slabs of a weight-gradient GEMM and apply weight_norm's backward:
//   dW[o][c][tap] = sum_s P[s][row_off + o][col_off + tap*tap_stride + c]
//   w = g v/|v|  =>  dg = <dW, v>/|v| ,  dv = (g/|v|) dW - (g <dW, v>/|v|^3) v        (g == NULL: dv = dW)
//   db[o] = sum_s P[s][row_off + o][col_bias]    (the ones-row column), optional, accumulated if db_accum
__global__ __launch_bounds__(256) void wn_backward_kernel(const WnBwdArgs a) {
    extern __shared__ float s_dw[];          // Cin*Kt
    __shared__ float red[2][4];
    const int o = blockIdx.x, tid = threadIdx.x;
    const int n = a.Cin * a.Kt;
    const float* vrow = a.v + (size_t)o * n;
    float dot = 0.f, ss = 0.f;
    // (tap, c) order: consecutive threads read consecutive slab columns (the slabs are nsplit x the row length).  Where the
    // layout allows, 16 bytes per thread and slab, four slabs in flight: the kernel is a latency-bound stream of 44-67 MB.
    const size_t slab_stride = (size_t)a.Prows * a.Pcols;
    const float* prow = a.P + ((size_t)a.row_off + o) * a.Pcols + a.col_off;
    if (!((a.Cin | a.Pcols | a.col_off | a.tap_stride) & 3) && !((uintptr_t)a.P & 15)) {
        // Two row positions per thread and up to eight slabs of each in flight at once: a row of <= 2048 weights over <= 8 slabs is
        // ONE memory round trip per workgroup (the first form took one per four slabs and per 1024 weights: four dependent round
        // trips for the gate convolution's 7 x 1536, 18-21 us per launch on a stream that runs 288 of them per step).  Slabs are
        // added in ascending order: a fixed order, bitwise reproducible.
        constexpr int SU = 8;
        for (int j0 = 0; j0 < n; j0 += 2048) {
            const float* pj[2];
            bool okj[2];
            int ij[2];
            float vr[2][4];
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int j = j0 + q * 1024 + tid * 4;
                okj[q] = j < n;
                const int jc = okj[q] ? j : 0;
                const int tap = jc / a.Cin, c = jc - tap * a.Cin;
                pj[q] = prow + tap * a.tap_stride + c;
                ij[q] = c * a.Kt + tap;
#pragma unroll
                for (int e = 0; e < 4; ++e) vr[q][e] = vrow[ij[q] + e * a.Kt];
            }
            f32x4 dw4[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
            for (int s0 = 0; s0 < a.nsplit; s0 += SU) {
                f32x4 d[2][SU];
#pragma unroll
                for (int q = 0; q < 2; ++q)
#pragma unroll
                    for (int u = 0; u < SU; ++u) {
                        const int sc = s0 + u < a.nsplit ? s0 + u : a.nsplit - 1;
                        d[q][u] = *(const f32x4*)(pj[q] + (size_t)sc * slab_stride);
                    }
#pragma unroll
                for (int q = 0; q < 2; ++q)
#pragma unroll
                    for (int u = 0; u < SU; ++u)
                        if (s0 + u < a.nsplit) dw4[q] += d[q][u];
            }
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                if (!okj[q]) continue;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    s_dw[ij[q] + e * a.Kt] = dw4[q][e];
                    dot += dw4[q][e] * vr[q][e];
                    ss += vr[q][e] * vr[q][e];
                }
            }
        }
    } else {
        for (int j = tid; j < n; j += 256) {
            const int tap = j / a.Cin, c = j - tap * a.Cin;
            const int i = c * a.Kt + tap;
            float dw = 0.f;
            for (int s = 0; s < a.nsplit; ++s) dw += prow[(size_t)s * slab_stride + tap * a.tap_stride + c];
            s_dw[i] = dw;
            const float vv = vrow[i];
            dot += dw * vv;
            ss += vv * vv;
        }
    }
    dot = wave_sum(dot);
    ss = wave_sum(ss);
    if ((tid & 63) == 0) { red[0][tid >> 6] = dot; red[1][tid >> 6] = ss; }
    __syncthreads();
    dot = red[0][0] + red[0][1] + red[0][2] + red[0][3];
    ss = red[1][0] + red[1][1] + red[1][2] + red[1][3];
    if (a.g) {
        const float nrm = sqrtf(ss), gg = a.g[o];
        const float k1 = gg / nrm, k2 = gg * dot / (nrm * ss);
        for (int i = tid; i < n; i += 256) a.dv[(size_t)o * n + i] = k1 * s_dw[i] - k2 * vrow[i];
        if (tid == 0) a.dg[o] = dot / nrm;
    } else {
        for (int i = tid; i < n; i += 256) a.dv[(size_t)o * n + i] = s_dw[i];
    }
    if (tid == 0 && a.db) {
        float b = 0.f;
        for (int s = 0; s < a.nsplit; ++s)
            for (int j = 0; j < a.n_bias_cols; ++j) b += a.P[((size_t)s * a.Prows + a.row_off + o) * a.Pcols + a.col_bias + j];
        a.db[o] = a.db_accum ? a.db[o] + b : b;
    }
}
hipError_t t2s_launch_wn_backward(const WnBwdArgs& a, hipStream_t stream) {
    hipLaunchKernelGGL(wn_backward_kernel, dim3(a.O), dim3(256), (size_t)a.Cin * a.Kt * sizeof(float), stream, a);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// Affine coupling backward + un-apply (reference glow.py:241-246), elementwise over [B][nh][L]:
//   a1 = (a1' - b) / exp(ls);  d_b = d_a1';  d_ls = d_a1' * a1 * exp(ls) + g_ls;  d_a1 = d_a1' * exp(ls)
// z / dz: [B][G][L], channel c_off + nh + i; wn_out [B][2nh][L] = (b ; ls) saved by the forward;
// d_out [B][2nh][L] receives (d_b ; d_ls).
__global__ void affine_backward_kernel(float* z, float* dz, const float* wn_out, const float* g_ls, int g_ls_scalar, float* d_out,
                                       int G, int c_off, int nh, int L) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int i = blockIdx.y, b = blockIdx.z;
    if (t >= L) return;
    const size_t zi = ((size_t)b * G + c_off + nh + i) * L + t;
    const float bb = wn_out[((size_t)b * 2 * nh + i) * L + t];
    const float ls = wn_out[((size_t)b * 2 * nh + nh + i) * L + t];
    const float e = expf(ls);
    const float a1 = (z[zi] - bb) / e;
    const float d = dz[zi];
    z[zi] = a1;
    dz[zi] = d * e;
    d_out[((size_t)b * 2 * nh + i) * L + t] = d;
    d_out[((size_t)b * 2 * nh + nh + i) * L + t] =
        d * a1 * e + (g_ls ? (g_ls_scalar ? g_ls[0] : g_ls[((size_t)b * nh + i) * L + t]) : 0.f);
}
hipError_t t2s_launch_affine_backward(float* z, float* dz, const float* wn_out, const float* g_ls, int g_ls_scalar, float* d_out,
                                      int B, int G, int c_off, int nh, int L, hipStream_t stream) {
    hipLaunchKernelGGL(affine_backward_kernel, dim3((L + 255) / 256, nh, B), dim3(256), 0, stream, z, dz, wn_out, g_ls,
                       g_ls_scalar, d_out, G, c_off, nh, L);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// Small-matrix weight gradients that contract over (b, t):
//   out[r][j] = sum_{b,t} P[b][r][t] * Q[b][j][t]      r < R channels of a plane set, j < J <= 16 rows of an f32 tensor
// P comes from planes (hi/lo bf16, or f32 planes when P_lo == NULL && P_f32 != NULL), Q is [B][Jtot][L] f32
// at channel offset q_off.  Optionally also rowsum[r] = sum P.  One workgroup per 32-channel chunk.
// Used for dW_end (P = skip sum, Q = d_out; transposed output), dW_start (P = dx, Q = a0).
// grid = (chunks, B * TSPLIT): block (chunk, b, s) covers time steps t = s*8 + tl, stride 8*TSPLIT, and writes a
// partial [17] per channel to scratch; small_wgrad_reduce_kernel sums the partials in a fixed order.
#define SW_TSPLIT 8
__global__ __launch_bounds__(256) void small_wgrad_kernel(const SmallWgradArgs a) {
    __shared__ float s_acc[8][32][17];
    const int tid = threadIdx.x, ci = tid & 31, tl = tid >> 5;
    const int chunk = blockIdx.x;
    const int b = blockIdx.y / SW_TSPLIT, sp = blockIdx.y % SW_TSPLIT;
    float acc[17];
#pragma unroll
    for (int j = 0; j < 17; ++j) acc[j] = 0.f;
    const size_t prow = ((size_t)b * a.chunks + chunk) * a.Lp + a.halo;
    for (int t = sp * 8 + tl; t < a.L; t += 8 * SW_TSPLIT) {
        float p;
        if (a.P_f32) p = a.P_f32[(prow + t) * 32 + ci];
        else p = join_bf16(a.P_hi[(prow + t) * 32 + ci], a.P_lo[(prow + t) * 32 + ci]);
#pragma unroll
        for (int j = 0; j < 16; ++j)
            if (j < a.J) acc[j] += p * a.Q[((size_t)b * a.Jtot + a.q_off + j) * a.L + t];
        acc[16] += p;
    }
#pragma unroll
    for (int j = 0; j < 17; ++j) s_acc[tl][ci][j] = acc[j];
    __syncthreads();
    if (tl == 0) {
        float* dst = a.scratch + (((size_t)blockIdx.y * a.chunks + chunk) * 32 + ci) * 17;
        for (int j = 0; j < 17; ++j) {
            float s = 0.f;
            for (int k = 0; k < 8; ++k) s += s_acc[k][ci][j];
            dst[j] = s;
        }
    }
}
__global__ void small_wgrad_reduce_kernel(const SmallWgradArgs a, int nparts) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= a.R) return;
    float acc[17];
    for (int j = 0; j < 17; ++j) acc[j] = 0.f;
    for (int p = 0; p < nparts; ++p) {
        const float* src = a.scratch + ((size_t)p * a.chunks * 32 + c) * 17;
        for (int j = 0; j < 17; ++j) acc[j] += src[j];
    }
    for (int j = 0; j < a.J; ++j) {
        if (a.out_transposed) a.out[(size_t)j * a.R + c] = acc[j];
        else a.out[(size_t)c * a.J + j] = acc[j];
    }
    if (a.rowsum) a.rowsum[c] = acc[16];
}
hipError_t t2s_launch_small_wgrad(const SmallWgradArgs& a, hipStream_t stream) {
    hipLaunchKernelGGL(small_wgrad_kernel, dim3(a.chunks, a.B * SW_TSPLIT), dim3(256), 0, stream, a);
    hipLaunchKernelGGL(small_wgrad_reduce_kernel, dim3((a.R + 255) / 256), dim3(256), 0, stream, a, a.B * SW_TSPLIT);
    return hipGetLastError();
}
size_t t2s_small_wgrad_scratch_floats(int B, int chunks) { return (size_t)B * SW_TSPLIT * chunks * 32 * 17; }

// colsum[j] = sum_{b,t} Q[b][q_off + j][t]    (bias gradient of WN.end)
__global__ __launch_bounds__(256) void rows_sum_kernel(const float* Q, int B, int Jtot, int q_off, int L, float* out) {
    __shared__ float red[4];
    const int j = blockIdx.x, tid = threadIdx.x;
    float s = 0.f;
    for (int b = 0; b < B; ++b)
        for (int t = tid; t < L; t += 256) s += Q[((size_t)b * Jtot + q_off + j) * L + t];
    s = wave_sum(s);
    if ((tid & 63) == 0) red[tid >> 6] = s;
    __syncthreads();
    if (tid == 0) out[j] = red[0] + red[1] + red[2] + red[3];
}
hipError_t t2s_launch_rows_sum(const float* Q, int B, int Jtot, int q_off, int J, int L, float* out, hipStream_t stream) {
    hipLaunchKernelGGL(rows_sum_kernel, dim3(J), dim3(256), 0, stream, Q, B, Jtot, q_off, L, out);
    return hipGetLastError();
}

// d_z[b][c_off + j][t] += sum_c w[c][j] * dx[c][t]      (data gradient of WN.start; dx from planes)
__global__ __launch_bounds__(256) void start_dgrad_kernel(const u16* __restrict__ X_hi, const u16* __restrict__ X_lo,
                                                          const float* __restrict__ w, float* dz, int G, int c_off, int nh,
                                                          int C, int L, int Lp, int halo) {
    const int tid = threadIdx.x, l32 = tid & 31;
    const int t = blockIdx.x * 8 + (tid >> 5);
    const int b = blockIdx.y;
    const int nchunks = (C + 31) / 32;
    float part[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) part[j] = 0.f;
    const bool tv = t < L;
    if (tv) {
        for (int ck = 0; ck < nchunks; ++ck) {
            const int c = ck * 32 + l32;
            if (c < C) {
                const size_t idx = (((size_t)b * nchunks + ck) * Lp + halo + t) * 32 + l32;
                const float v = join_bf16(X_hi[idx], X_lo[idx]);
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    if (j < nh) part[j] += v * w[c * nh + j];
            }
        }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j)
        if (j < nh) {
#pragma unroll
            for (int off = 16; off > 0; off >>= 1) part[j] += __shfl_xor(part[j], off, 64);
        }
    if (tv && l32 < nh) {
        float v = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j)
            if (j == l32) v = part[j];
        dz[((size_t)b * G + c_off + l32) * L + t] += v;
    }
}
hipError_t t2s_launch_start_dgrad(const u16* X_hi, const u16* X_lo, const float* w, float* dz, int B, int G, int c_off,
                                  int nh, int C, int L, int Lp, int halo, hipStream_t stream) {
    hipLaunchKernelGGL(start_dgrad_kernel, dim3((L + 7) / 8, B), dim3(256), 0, stream, X_hi, X_lo, w, dz, G, c_off, nh, C,
                       L, Lp, halo);
    return hipGetLastError();
}

// dW[i][j] = sum_{b,t} dz[b][c_off+i][t] * zin[b][c_off+j][t]  +  gscale * Winv[j][i]     (reference glow.py:100-101:
// d(B L logdet W)/dW = B L W^-T); one workgroup per row i.
__global__ __launch_bounds__(1024) void convinv_wgrad_kernel(const float* dz, const float* zin, const float* Winv,
                                                             const float* gscale_ptr, float gmul, int B, int G, int c_off,
                                                             int n, int L, float* dW) {
    __shared__ float red[16][16];
    const int i = blockIdx.x, tid = threadIdx.x;
    float acc[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[j] = 0.f;
    for (int b = 0; b < B; ++b)
        for (int t = tid; t < L; t += 1024) {
            const float d = dz[((size_t)b * G + c_off + i) * L + t];
#pragma unroll
            for (int j = 0; j < 16; ++j)
                if (j < n) acc[j] += d * zin[((size_t)b * G + c_off + j) * L + t];
        }
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const float s = wave_sum(acc[j]);
        if ((tid & 63) == 0) red[tid >> 6][j] = s;
    }
    __syncthreads();
    if (tid < n) {
        float s = 0.f;
        for (int k = 0; k < 16; ++k) s += red[k][tid];
        const float gs = gscale_ptr ? gscale_ptr[0] * gmul : 0.f;
        dW[i * n + tid] = s + gs * Winv[tid * n + i];
    }
}
hipError_t t2s_launch_convinv_wgrad(const float* dz, const float* zin, const float* Winv, const float* gscale_ptr,
                                    float gmul, int B, int G, int c_off, int n, int L, float* dW, hipStream_t stream) {
    hipLaunchKernelGGL(convinv_wgrad_kernel, dim3(n), dim3(1024), 0, stream, dz, zin, Winv, gscale_ptr, gmul, B, G, c_off,
                       n, L, dW);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// ConvTranspose1d weight / bias gradient from the conditioning-plane gradient d_s (reference glow.py:215-221):
//   dW[ci][co][k] = sum_{b,f} mel[b][ci][f] * d_up[b][co][stride*f + k],  d_up[b][co][G*t + g] = d_s[b][co*G + g][t]
// One thread per (co, k) keeps CI accumulators (a slice of the input channels) and sweeps (b, f); grid.z walks the slices, so the
// chip sees n_mel / CI times more workgroups than with all 80 accumulators in one thread (320 workgroups, each iteration a
// dependent 2-byte gather plus two barriers, ran 2.15 ms; this form: the slice of mel of one batch entry is staged once per
// b, the gathers of four frames are in flight together).
template <int CI>
__global__ __launch_bounds__(256) void upsample_wgrad_kernel(const u16* __restrict__ D_hi, const u16* __restrict__ D_lo,
                                                             const float* __restrict__ mel, int B, int M, int F, int FT, int ksize,
                                                             int stride, int G, int L, int Lp, int halo, float* dW) {
    extern __shared__ float s_mel[];         // [CI][FT]: a slice of FT frames of the current batch entry
    const int k = blockIdx.x * 256 + threadIdx.x;
    const int co = blockIdx.y;
    const int ci0 = blockIdx.z * CI;
    const int nchunks = (M * G + 31) / 32;
    float acc[CI];
#pragma unroll
    for (int c = 0; c < CI; ++c) acc[c] = 0.f;
    const bool kv = k < ksize;
    // (b, frame) ascending, the frames staged FT at a time: any segment length, the same summation order for every FT
    for (int b = 0; b < B; ++b)
      for (int fb = 0; fb < F; fb += FT) {
        const int fn = min(FT, F - fb);
        __syncthreads();
        for (int i = threadIdx.x; i < CI * fn; i += 256) {
            const int c = i / fn, f = i - c * fn;
            s_mel[c * FT + f] = ci0 + c < M ? mel[((size_t)b * M + ci0 + c) * F + fb + f] : 0.f;
        }
        __syncthreads();
        for (int f0 = 0; f0 < fn; f0 += 4) {
            float d[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int f = fb + f0 + u;
                const int s = stride * f + k;
                const int t = s / G, g = s - t * G;
                const bool ok = kv && f0 + u < fn && t < L;
                const int ch = co * G + g;
                const size_t idx = (((size_t)b * nchunks + (ch >> 5)) * Lp + halo + (ok ? t : 0)) * 32 + (ch & 31);
                const u16 h = D_hi[idx], l = D_lo[idx];          // unconditional: row halo + 0 always exists
                d[u] = ok ? join_bf16(h, l) : 0.f;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (f0 + u >= fn) break;
#pragma unroll
                for (int c = 0; c < CI; ++c) acc[c] += s_mel[c * FT + f0 + u] * d[u];
            }
        }
      }
    if (kv) {
#pragma unroll
        for (int c = 0; c < CI; ++c)
            if (ci0 + c < M) dW[((size_t)(ci0 + c) * M + co) * ksize + k] = acc[c];
    }
}
hipError_t t2s_launch_upsample_wgrad(const u16* D_hi, const u16* D_lo, const float* mel, int B, int M, int F, int ksize,
                                     int stride, int G, int L, int Lp, int halo, float* dW, hipStream_t stream) {
    constexpr int CI = 20;
    const int FT = F < 512 ? (F + 3) / 4 * 4 : 512;        // frames staged per pass: 40 KB of LDS at most, any F
    hipLaunchKernelGGL(upsample_wgrad_kernel<CI>, dim3((ksize + 255) / 256, M, (M + CI - 1) / CI), dim3(256),
                       (size_t)CI * FT * sizeof(float), stream, D_hi, D_lo, mel, B, M, F, FT, ksize, stride, G, L, Lp, halo, dW);
    return hipGetLastError();
}

// db[co] = sum over every valid output sample of d_up[b][co][s]  (each sample counted once)
__global__ __launch_bounds__(256) void upsample_bgrad_kernel(const u16* __restrict__ D_hi, const u16* __restrict__ D_lo,
                                                             int B, int M, int G, int L, int Lp, int halo, float* db) {
    __shared__ float red[4];
    const int co = blockIdx.x, tid = threadIdx.x;
    const int nchunks = (M * G + 31) / 32;
    float s = 0.f;
    for (int b = 0; b < B; ++b)
        for (int i = tid; i < L * G; i += 256) {
            const int t = i / G, g = i - t * G;
            const int ch = co * G + g;
            const size_t idx = (((size_t)b * nchunks + (ch >> 5)) * Lp + halo + t) * 32 + (ch & 31);
            s += join_bf16(D_hi[idx], D_lo[idx]);
        }
    s = wave_sum(s);
    if ((tid & 63) == 0) red[tid >> 6] = s;
    __syncthreads();
    if (tid == 0) db[co] = red[0] + red[1] + red[2] + red[3];
}
hipError_t t2s_launch_upsample_bgrad(const u16* D_hi, const u16* D_lo, int B, int M, int G, int L, int Lp, int halo,
                                     float* db, hipStream_t stream) {
    hipLaunchKernelGGL(upsample_bgrad_kernel, dim3(M), dim3(256), 0, stream, D_hi, D_lo, B, M, G, L, Lp, halo, db);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// Adam (torch.optim.Adam semantics, reference waveglow/train.py:79,124), one launch for every parameter:
//   m = b1 m + (1-b1) g ; v = b2 v + (1-b2) g^2 ; p -= lr/(1-b1^t) * m / (sqrt(v)/sqrt(1-b2^t) + eps)
// gscale multiplies the gradient first (1/world_size after an all-reduce SUM).
__global__ __launch_bounds__(256) void adam_table_kernel(const AdamJob* __restrict__ jobs, int n_jobs, float lr, float b1,
                                                         float b2, float eps, float bc1, float bc2_sqrt, float gscale,
                                                         float weight_decay) {
    const long blk = blockIdx.x;
    int lo = 0, hi = n_jobs - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (jobs[mid].blk_start <= blk) lo = mid; else hi = mid - 1;
    }
    const AdamJob j = jobs[lo];
    const long base = (blk - j.blk_start) * 1024;
    for (int e = 0; e < 4; ++e) {
        const long i = base + e * 256 + threadIdx.x;
        if (i < j.n) {
            float g = j.g[i] * gscale;
            const float p = j.p[i];
            if (weight_decay != 0.f) g += weight_decay * p;
            const float m = b1 * j.m[i] + (1.f - b1) * g;
            const float v = b2 * j.v[i] + (1.f - b2) * g * g;
            j.m[i] = m;
            j.v[i] = v;
            j.p[i] = p - (lr / bc1) * m / (sqrtf(v) / bc2_sqrt + eps);
        }
    }
}
hipError_t t2s_launch_adam_table(const AdamJob* jobs, int n_jobs, long total_blocks, float lr, float b1, float b2,
                                 float eps, float bc1, float bc2_sqrt, float gscale, float weight_decay,
                                 hipStream_t stream) {
    hipLaunchKernelGGL(adam_table_kernel, dim3((unsigned)total_blocks), dim3(256), 0, stream, jobs, n_jobs, lr, b1, b2, eps,
                       bc1, bc2_sqrt, gscale, weight_decay);
    return hipGetLastError();
}
