// Backward kernels of the Tacotron-2 training step (reference: autograd over tacotron/tacotron.py:36-49,
// 355-429 and tacotron/modules.py:19-22,94-137, driven by train.py:219-225).
//
// Weight gradients of every Linear / LSTM matrix are GEMMs that contract over (step, batch) "items": the f32 row
// tensors [items][C] are turned into time-major (hi, lo) planes by rows_to_tm_kernel and fed to the same split-K
// conv_gemm weight-gradient path the WaveGlow backward uses.  What is genuinely sequential - the per-step chain
// through the two LSTM cells and the location-sensitive attention - is the kernels below, one launch per stage.
#include <string.h>
#include "t2s_common.h"
#include "t2s_kernels.h"
#include "taco_bwd_ops.h"

#include <stdlib.h>

static __device__ __forceinline__ float sum3(const float* a, long sa, const float* b, long sb, const float* c, long sc,
                                              int item, int j) {
    // three independent loads, then selects: written as `if (a) v += a[..]` three times, each load sat behind its own branch and
    // a full wait - three (six, for two values per thread) round trips to memory at the head of every kernel that starts with it
    const float* any = a ? a : (b ? b : c);
    if (!any) return 0.f;
    const long sany = a ? sa : (b ? sb : sc);
    const float xa = (a ? a : any)[(size_t)item * (a ? sa : sany) + j];
    const float xb = (b ? b : any)[(size_t)item * (b ? sb : sany) + j];
    const float xc = (c ? c : any)[(size_t)item * (c ? sc : sany) + j];
    float v = 0.f;
    v += a ? xa : 0.f;
    v += b ? xb : 0.f;
    v += c ? xc : 0.f;
    return v;
}

// ------------------------------------------------------------------------------------------------
// f32 rows x[item - shift][c0 + c] (row stride ld)  ->  time-major planes tm[item/32][n_off + c][item%32]
__global__ __launch_bounds__(256) void rows_to_tm_kernel(const float* __restrict__ x, long ld, int items, int shift, int C,
                                                         u16* dst_hi, u16* dst_lo, int Npad, int n_off, long x_bstride,
                                                         long dst_bstride) {
    __shared__ float tile[32][33];
    // channel chunk on the FAST block index: the workgroups in flight at one time read consecutive 128-byte pieces of the same 32
    // rows (whole rows stream) and write one contiguous run of planes.  With the item chunk fastest they touched 128 bytes every
    // `ld` floats across the whole array at once - 1.6 TB/s on the 2 x 419 MB of the decoder cell's gate gradients.
    const int cc = blockIdx.x, ic = blockIdx.y, tid = threadIdx.x;
    x += (size_t)blockIdx.z * x_bstride;            // batched form: one set of planes per blockIdx.z
    dst_hi += (size_t)blockIdx.z * dst_bstride;
    dst_lo += (size_t)blockIdx.z * dst_bstride;
    {
        const int r = tid >> 3, q = tid & 7;
        const int item = ic * 32 + r - shift;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int c = cc * 32 + q * 4 + e;
            tile[r][q * 4 + e] = (item >= 0 && item < items && c < C) ? x[(size_t)item * ld + c] : 0.f;
        }
    }
    __syncthreads();
    {
        const int ci = tid >> 3, tq = tid & 7;
        if (cc * 32 + ci < C) {
            u16x4 vh, vl;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                u16 h, l;
                split_bf16(tile[tq * 4 + e][ci], h, l);
                vh[e] = h;
                vl[e] = l;
            }
            const size_t idx = (((size_t)ic) * Npad + n_off + cc * 32 + ci) * 32 + tq * 4;
            *(u16x4*)(dst_hi + idx) = vh;
            *(u16x4*)(dst_lo + idx) = vl;
        }
    }
}
hipError_t t2s_launch_rows_to_tm(const float* x, long ld, int items, int items_pad, int shift, int C, u16* dst_hi,
                                 u16* dst_lo, int Npad, int n_off, hipStream_t stream) {
    if (items_pad / 32 > 65535) return hipErrorInvalidValue;
    hipLaunchKernelGGL(rows_to_tm_kernel, dim3((C + 31) / 32, items_pad / 32), dim3(256), 0, stream, x, ld, items, shift, C,
                       dst_hi, dst_lo, Npad, n_off, 0L, 0L);
    return hipGetLastError();
}
hipError_t t2s_launch_rows_to_tm_batched(const float* x, long ld, long x_bstride, int items, int items_pad, int shift, int C,
                                         u16* dst_hi, u16* dst_lo, long dst_bstride, int Npad, int n_off, int nb,
                                         hipStream_t stream) {
    if (items_pad / 32 > 65535) return hipErrorInvalidValue;
    hipLaunchKernelGGL(rows_to_tm_kernel, dim3((C + 31) / 32, items_pad / 32, nb), dim3(256), 0, stream, x, ld, items, shift, C,
                       dst_hi, dst_lo, Npad, n_off, x_bstride, dst_bstride);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// LSTMCell backward, pointwise part (torch gate order i,f,g,o):
//   dh = (dh1 + dh2 + dh3) * dropout ;  do = dh tanh(c') ;  dc = dc_carry + dh o (1 - tanh(c')^2)
//   dgates = (dc g i(1-i), dc c f(1-f), dc i (1-g^2), do o(1-o)) ;  dc_carry <- dc f
__global__ void lstm_cell_bwd_kernel(const LstmBwdArgs a) {
    const int u = blockIdx.x * blockDim.x + threadIdx.x;
    const int b = blockIdx.y;
    __shared__ float s_dq[256];
    if (a.wq) {                  // + W_q^T d_q: the query Linear hangs off this cell's output (tacotron.py:137)
        // d_q of this batch element into LDS first (blockIdx.y = b is block-uniform); when the attention backward runs its
        // last part on another stream, the per-chunk partials are summed here
        for (int k = threadIdx.x; k < a.q_dim; k += blockDim.x) {
            float v = 0.f;
            if (a.dq_part) {
                for (int c = 0; c < a.dq_nchunk; ++c) v += a.dq_part[((size_t)b * a.dq_nchunk + c) * a.q_dim + k];
            } else {
                v = a.dq[(size_t)b * a.q_dim + k];
            }
            s_dq[k] = v;
            if (a.dq_part && a.dq_out && blockIdx.x == 0) a.dq_out[(size_t)b * a.q_dim + k] = v;
        }
    }
    if (a.wq) __syncthreads();
    if (u >= a.H) return;
    float dh = sum3(a.dh1, a.s1, a.dh2, a.s2, a.dh3, a.s3, b, u);
    if (a.wq) {
        float e0 = 0.f, e1 = 0.f;
        const float* dq = s_dq;
        int k = 0;
        for (; k + 16 <= a.q_dim; k += 16) {            // sixteen independent loads in flight, then the FMAs
            float w[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) w[i] = a.wq[(size_t)(k + i) * a.H + u];
#pragma unroll
            for (int i = 0; i < 16; i += 2) { e0 += dq[k + i] * w[i]; e1 += dq[k + i + 1] * w[i + 1]; }
        }
        for (; k < a.q_dim; ++k) e0 += dq[k] * a.wq[(size_t)k * a.H + u];
        dh += e0 + e1;
    }
    const size_t idx = (size_t)b * a.H + u;
    if (a.drop_mask) dh = a.drop_mask[idx] ? dh * a.drop_scale : 0.f;
    const float* g4 = a.gates + (size_t)b * 4 * a.H + u;
    const float gi = g4[0], gf = g4[a.H], gg = g4[2 * a.H], go = g4[3 * a.H];
    const float tc = tanhf(a.c_new[idx]);
    const float cp = a.c_prev ? a.c_prev[idx] : 0.f;
    const float dc = a.dc_carry[idx] + dh * go * (1.f - tc * tc);
    float* dg = a.dgates + (size_t)b * 4 * a.H + u;
    dg[0] = dc * gg * gi * (1.f - gi);
    dg[a.H] = dc * cp * gf * (1.f - gf);
    dg[2 * a.H] = dc * gi * (1.f - gg * gg);
    dg[3 * a.H] = dh * tc * go * (1.f - go);
    a.dc_carry[idx] = dc * gf;
}
// The attention cell's form (+ W_q^T d_q): 64 hidden units per workgroup, the query dimension split over the four waves - four
// times the workgroups of the kernel above (512 at H = 1024, B = 32: it left half the CUs idle) and a quarter of the dependent
// load batches per thread.
__global__ __launch_bounds__(256) void lstm_cell_bwd_q_kernel(const LstmBwdArgs a) {
    __shared__ float s_dq[256];
    __shared__ float s_e[4][64];
    const int tid = threadIdx.x, lane = tid & 63, kq = tid >> 6;
    const int u = blockIdx.x * 64 + lane, b = blockIdx.y;
    const int kn = a.q_dim >> 2;                   // query rows per wave
    // this wave's slice of W_q for unit u: requested before anything else
    float w[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) w[i] = i < kn ? a.wq[(size_t)(kq * kn + i) * a.H + u] : 0.f;
    for (int k = tid; k < a.q_dim; k += 256) {
        float v = 0.f;
        if (a.dq_part) {
            for (int c = 0; c < a.dq_nchunk; ++c) v += a.dq_part[((size_t)b * a.dq_nchunk + c) * a.q_dim + k];
            if (a.dq_out && blockIdx.x == 0) a.dq_out[(size_t)b * a.q_dim + k] = v;
        } else {
            v = a.dq[(size_t)b * a.q_dim + k];
        }
        s_dq[k] = v;
    }
    // the pointwise operands of the 64 finishing threads
    const size_t idx = (size_t)b * a.H + u;
    float dh = 0.f, gi = 0.f, gf = 0.f, gg = 0.f, go = 0.f, cn = 0.f, cp = 0.f, dcc = 0.f;
    bool keep = true;
    if (kq == 0) {
        dh = sum3(a.dh1, a.s1, a.dh2, a.s2, a.dh3, a.s3, b, u);
        const float* g4 = a.gates + (size_t)b * 4 * a.H + u;
        gi = g4[0]; gf = g4[a.H]; gg = g4[2 * a.H]; go = g4[3 * a.H];
        cn = a.c_new[idx];
        cp = a.c_prev ? a.c_prev[idx] : 0.f;
        dcc = a.dc_carry[idx];
        if (a.drop_mask) keep = a.drop_mask[idx] != 0;
    }
    __syncthreads();
    float e0 = 0.f, e1 = 0.f;
#pragma unroll
    for (int i = 0; i < 32; i += 2) {
        e0 += s_dq[kq * kn + i] * w[i];            // (w is 0 past kn; s_dq is read inside its 256 entries: kn <= 32)
        e1 += s_dq[kq * kn + i + 1] * w[i + 1];
    }
    s_e[kq][lane] = e0 + e1;
    __syncthreads();
    if (kq != 0) return;
    dh += (s_e[0][lane] + s_e[1][lane]) + (s_e[2][lane] + s_e[3][lane]);
    if (a.drop_mask) dh = keep ? dh * a.drop_scale : 0.f;
    const float tc = tanhf(cn);
    const float dc = dcc + dh * go * (1.f - tc * tc);
    float* dg = a.dgates + (size_t)b * 4 * a.H + u;
    dg[0] = dc * gg * gi * (1.f - gi);
    dg[a.H] = dc * cp * gf * (1.f - gf);
    dg[2 * a.H] = dc * gi * (1.f - gg * gg);
    dg[3 * a.H] = dh * tc * go * (1.f - go);
    a.dc_carry[idx] = dc * gf;
}
hipError_t t2s_launch_lstm_cell_bwd(const LstmBwdArgs& a, hipStream_t stream) {
    if (a.wq && (a.H & 63) == 0 && a.q_dim == 128) {
        hipLaunchKernelGGL(lstm_cell_bwd_q_kernel, dim3(a.H / 64, a.B), dim3(256), 0, stream, a);
        return hipGetLastError();
    }
    hipLaunchKernelGGL(lstm_cell_bwd_kernel, dim3((a.H + 255) / 256, a.B), dim3(256), 0, stream, a);
    return hipGetLastError();
}

// dz = (y > 0) ? scale * dy : 0   (backward of y = relu(z) * mask * scale; y > 0 iff z > 0 and mask == 1)
__global__ void relu_drop_bwd_kernel(const float* dy, const float* y, float scale, size_t n, float* dz) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dz[i] = y[i] > 0.f ? dy[i] * scale : 0.f;
}
hipError_t t2s_launch_relu_drop_bwd(const float* dy, const float* y, float scale, size_t n, float* dz, hipStream_t stream) {
    hipLaunchKernelGGL(relu_drop_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, dy, y, scale, n, dz);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// One step of the location-sensitive attention, backward (reference tacotron.py:124-166,379).  Forward (recomputed here
// from the saved query / weights):
//   f = conv1d([w_prev ; wc_prev], K) ; p = q + D f + pm ; e = v . tanh(p) ; w = softmax(e) ; ctx = w . mem ; wc = wc_prev + w
// Three launches, each over (T_in / 32 chunks) x batch workgroups so the step fills the chip at batch 32:
//   A  att_bwd_dw      d_w[t] = mem[t] . d_ctx + carries ; d_mem[t] += w[t] d_ctx
//   B  att_bwd_energy  softmax backward (every workgroup re-reduces sum w d_w over the whole row: T floats), energies
//                      backward for its 32 positions: d_pmem +=, partial d_q / dv / dD^T ([f][a] per slot), d_f[t][:] = D^T d_pre[t]
//   C  att_bwd_conv    location-conv backward from d_f (with a halo of kernel/2 positions): the carries for step t-1 and
//                      the partial kernel gradient; chunk 0 also folds the partial queries into d_q.
// Parameter-gradient partials live in one slot per (batch element, chunk) and are summed once after the last step.
#define ATTB_CH 32
static __device__ __forceinline__ float attb_wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

__global__ __launch_bounds__(256) void att_bwd_dw_kernel(const AttBwdArgs a) {
    __shared__ __attribute__((aligned(16))) float s_dctx[1024];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.y, T = a.T, E = a.enc_dim;
    for (int c = tid; c < E; c += 256) {
        const float v = sum3(a.dctx1, a.sc1, a.dctx2, a.sc2, a.dctx3, a.sc3, b, c);
        s_dctx[c] = v;
        // deferred form: keep d_ctx of this step; d_memory = sum_t w_t (x) d_ctx_t is then ONE contraction over the decoder
        // steps after the loop instead of a read-modify-write of [T_in][enc] per step
        if (a.dctx_out && blockIdx.x == 0) a.dctx_out[(size_t)b * E + c] = v;
    }
    __syncthreads();
    const int tb = blockIdx.x * ATTB_CH + wave * 8;
    float acc[8], wt[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        acc[r] = 0.f;
        wt[r] = tb + r < T ? a.w_cur[(size_t)b * a.s_wcur + tb + r] : 0.f;
    }
    for (int c = lane * 4; c < E; c += 256) {
        const f32x4 dc = *(const f32x4*)&s_dctx[c];
        f32x4 m[8], dm[8];
#pragma unroll
        for (int r = 0; r < 8; ++r)
            if (tb + r < T) {
                const size_t mo = ((size_t)b * T + tb + r) * E + c;
                m[r] = *(const f32x4*)(a.memory + mo);
                if (a.d_memory) dm[r] = *(const f32x4*)(a.d_memory + mo);
            }
#pragma unroll
        for (int r = 0; r < 8; ++r)
            if (tb + r < T) {
                acc[r] += m[r][0] * dc[0] + m[r][1] * dc[1] + m[r][2] * dc[2] + m[r][3] * dc[3];
                if (a.d_memory) {
                    dm[r] += wt[r] * dc;
                    *(f32x4*)(a.d_memory + ((size_t)b * T + tb + r) * E + c) = dm[r];
                }
            }
    }
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        const float v = attb_wave_sum(acc[r]);
        const int t = tb + r;
        if (lane == 0 && t < T) a.dw_buf[(size_t)b * T + t] = v + a.dw_carry[(size_t)b * T + t] + a.dwc_carry[(size_t)b * T + t];
    }
}

__global__ __launch_bounds__(512) void att_bwd_energy_kernel(const AttBwdArgs a) {
    __shared__ float s_cat[2][ATTB_CH + 64];
    __shared__ float s_k[32 * 128];           // location-conv kernel [F][2][KS] (<= 4032 floats), then D^T [f][a]
    __shared__ float s_f[ATTB_CH][33];
    __shared__ float s_d[128 * 32];           // D [a][f]
    __shared__ float s_dpre[ATTB_CH][129];    // d_pre [t][a]
    __shared__ float s_de[ATTB_CH];
    __shared__ float s_vec[8][2][128];        // per-wave dq, dv
    __shared__ float s_red[8];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.y, chunk = blockIdx.x, t0 = chunk * ATTB_CH;
    const int T = a.T, AD = a.att_dim, KS = a.loc_ks, F = a.loc_f, pad = KS >> 1;
    const int len = a.lengths ? a.lengths[b] : T;
    const size_t slot = (size_t)b * gridDim.x + chunk;
    // ---- loads ----
    for (int i = tid; i < F * 2 * KS; i += 512) s_k[i] = a.w_loc_conv[i];
    for (int i = tid; i < 2 * (ATTB_CH + KS - 1); i += 512) {
        const int c = i / (ATTB_CH + KS - 1), j = i - c * (ATTB_CH + KS - 1);
        const int t = t0 + j - pad;
        const float* src = c ? a.wc_prev : a.w_prev;
        s_cat[c][j] = (src && t >= 0 && t < T) ? src[(size_t)b * (c ? a.s_wcprev : a.s_wprev) + t] : 0.f;
    }
    for (int i = tid; i < 128 * 32; i += 512) {
        const int ai = i >> 5, f = i & 31;
        s_d[i] = (f < F && ai < AD) ? a.w_loc_dense[ai * F + f] : 0.f;
    }
    // this thread's slice of the parameter-gradient slot, fetched early (it is read-modify-write at the very end):
    // dD^T [f][a]: a = tid % 128, f = (tid / 128) * 8 + i
    const int ga = tid & 127, gf0 = (tid >> 7) * 8;
    float dD_old[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) dD_old[i] = (ga < AD && gf0 + i < F) ? a.dD_part[slot * AD * F + (size_t)(gf0 + i) * AD + ga] : 0.f;
    const float dv_old = tid < AD ? a.dv_part[slot * AD + tid] : 0.f;
    // attention_dim on lanes (a0 = lane, a1 = lane + 64); a wave takes 4 positions
    const float q0 = lane < AD ? a.q[(size_t)b * AD + lane] : 0.f, q1 = lane + 64 < AD ? a.q[(size_t)b * AD + lane + 64] : 0.f;
    const float v0 = lane < AD ? a.w_v[lane] : 0.f, v1 = lane + 64 < AD ? a.w_v[lane + 64] : 0.f;
    float pm0[4], pm1[4], dpm0[4], dpm1[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int t = t0 + wave * 4 + r;
        const size_t po = ((size_t)b * T + t) * AD;
        const bool ok0 = t < T && lane < AD, ok1 = t < T && lane + 64 < AD;
        pm0[r] = ok0 ? a.pmem[po + lane] : 0.f;
        pm1[r] = ok1 ? a.pmem[po + lane + 64] : 0.f;
        dpm0[r] = ok0 ? a.d_pmem[po + lane] : 0.f;
        dpm1[r] = ok1 ? a.d_pmem[po + lane + 64] : 0.f;
    }
    // softmax backward needs sum_t w[t] d_w[t] over the whole row
    float part = 0.f;
    for (int t = tid; t < T; t += 512) part += a.w_cur[(size_t)b * a.s_wcur + t] * a.dw_buf[(size_t)b * T + t];
    part = attb_wave_sum(part);
    if (lane == 0) s_red[wave] = part;
    __syncthreads();
    const float sdot = ((s_red[0] + s_red[1]) + (s_red[2] + s_red[3])) + ((s_red[4] + s_red[5]) + (s_red[6] + s_red[7]));
    if (tid < ATTB_CH) {
        const int t = t0 + tid;
        s_de[tid] = (t < T && t < len) ? a.w_cur[(size_t)b * a.s_wcur + t] * (a.dw_buf[(size_t)b * T + t] - sdot) : 0.f;
    }
    // ---- location features of this chunk ----
    for (int i = tid; i < ATTB_CH * F; i += 512) {
        const int tq = i / F, f = i - tq * F;
        float acc = 0.f;
        for (int c = 0; c < 2; ++c)
#pragma unroll 8
            for (int j = 0; j < KS; ++j) acc += s_k[(f * 2 + c) * KS + j] * s_cat[c][tq + j];
        s_f[tq][f] = acc;
    }
    __syncthreads();
    for (int i = tid; i < 32 * 128; i += 512) s_k[i] = s_d[(i & 127) * 32 + (i >> 7)];      // D^T [f][a]: a on lanes below
    __syncthreads();
    // ---- energies backward ----
    float dq0 = 0.f, dq1 = 0.f, dv0 = 0.f, dv1 = 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int tq = wave * 4 + r, t = t0 + tq;
        float dp0 = 0.f, dp1 = 0.f;
        if (t < T) {                                          // wave-uniform
            float p0 = q0 + pm0[r], p1 = q1 + pm1[r];
#pragma unroll 8
            for (int f = 0; f < 32; ++f) {
                const float ff = s_f[tq][f];
                p0 += s_k[f * 128 + lane] * ff;
                p1 += s_k[f * 128 + 64 + lane] * ff;
            }
            const float th0 = tanhf(p0), th1 = tanhf(p1);
            const float de = s_de[tq];
            dp0 = lane < AD ? de * v0 * (1.f - th0 * th0) : 0.f;
            dp1 = lane + 64 < AD ? de * v1 * (1.f - th1 * th1) : 0.f;
            dq0 += dp0; dq1 += dp1;
            dv0 += de * th0; dv1 += de * th1;
            const size_t po = ((size_t)b * T + t) * AD;
            if (lane < AD) a.d_pmem[po + lane] = dpm0[r] + dp0;
            if (lane + 64 < AD) a.d_pmem[po + lane + 64] = dpm1[r] + dp1;
        }
        s_dpre[tq][lane] = dp0;
        s_dpre[tq][lane + 64] = dp1;
        __builtin_amdgcn_sched_barrier(0);                   // keep the four positions' LDS reads from being hoisted together
    }
    s_vec[wave][0][lane] = dq0; s_vec[wave][0][lane + 64] = dq1;
    s_vec[wave][1][lane] = dv0; s_vec[wave][1][lane + 64] = dv1;
    __syncthreads();
    // ---- dD^T[f][a] += sum_t f[t][f] d_pre[t][a]   (thread: one a, eight f) ----
    {
        float acc[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = 0.f;
#pragma unroll 4
        for (int tq = 0; tq < ATTB_CH; ++tq) {
            const float dp = s_dpre[tq][ga];
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] += dp * s_f[tq][gf0 + i];
        }
#pragma unroll
        for (int i = 0; i < 8; ++i)
            if (ga < AD && gf0 + i < F) a.dD_part[slot * AD * F + (size_t)(gf0 + i) * AD + ga] = dD_old[i] + acc[i];
    }
    // ---- d_f[t][f] = sum_a D[a][f] d_pre[t][a]   (thread: one t, f and f + 16) ----
    {
        const int tq = tid >> 4, f = tid & 15;
        float e0 = 0.f, e1 = 0.f;
#pragma unroll 8
        for (int ai = 0; ai < 128; ++ai) {
            const float dp = s_dpre[tq][ai];
            e0 += s_d[ai * 32 + f] * dp;
            e1 += s_d[ai * 32 + 16 + f] * dp;
        }
        if (t0 + tq < T) {
            a.df_buf[((size_t)b * T + t0 + tq) * 32 + f] = e0;
            a.df_buf[((size_t)b * T + t0 + tq) * 32 + 16 + f] = e1;
        }
    }
    if (tid < AD) {
        float sq = 0.f, sv = 0.f;
#pragma unroll
        for (int w = 0; w < 8; ++w) { sq += s_vec[w][0][tid]; sv += s_vec[w][1][tid]; }
        a.dq_part[slot * AD + tid] = sq;
        a.dv_part[slot * AD + tid] = dv_old + sv;
    }
}

// The same stage with its four small matrix products on the exact-f32 matrix cores (v_mfma_f32_16x16x4_f32: f32 operands,
// f32 accumulate), for the reference's shape (attention_dim 128, 32 location filters, kernel <= 31).  The VALU kernel above is
// LDS-bound: ~1200 ds_read_b32 per thread (two per MAC of the location convolution, the energies, dD^T and d_f) - 10 of its
// 22 us at 128 B/clk (profiles/r03_taco_step_kernel_counters_before.json).  Per 32-position chunk the products are
//   F[t][f]     = sum_k cat-window[t][k] K[k][f]          [32 x 64] x [64 x 32]     (location features, recomputed)
//   P[t][a]     = sum_f F[t][f] D^T[f][a]                 [32 x 32] x [32 x 128]    -> tanh, d_pre in the epilogue
//   dD^T[f][a] += sum_t F[t][f] d_pre[t][a]               [32 x 32] x [32 x 128]
//   d_f[t][f]   = sum_a d_pre[t][a] D[a][f]               [32 x 128] x [128 x 32]
// 448 MFMAs per workgroup instead of ~0.6 M LDS-fed FMAs; every fragment read is a conflict-free ds_read_b32 (row strides 33
// for operands read down a column, 144 for operands read along a row; d_pre is kept in both).  A wave owns 16 attention
// channels over all 32 positions of the chunk, so d_q / dv need no cross-wave reduction.
// LDS, 58.9 KB (two workgroups per CU): the convolution kernel's space is reused for d_pre[a][t], D^T's for D.
#define ATTB_LDS_FLOATS (2 * (ATTB_CH + 64) + 128 * 33 + ATTB_CH * 33 + 32 * 144 + ATTB_CH * 144 + ATTB_CH + 8)
__global__ __launch_bounds__(512) void att_bwd_energy_mfma_kernel(const AttBwdArgs a) {
    __shared__ __attribute__((aligned(16))) float s_all[ATTB_LDS_FLOATS];
    constexpr int AD = 128, F = 32;
    float* s_cat = s_all;                          // [2][ATTB_CH + 64]
    float* s_kb = s_cat + 2 * (ATTB_CH + 64);      // [64][48]   conv kernel as B operand [k = c * KS + j][f], rows >= 2 KS zero
    float* s_dpT = s_kb;                           // [128][33]  d_pre[a][t]   (after stage 1)
    float* s_f = s_kb + 128 * 33;                  // [32][33]   F[t][f]
    float* s_dT = s_f + ATTB_CH * 33;              // [32][144]  D^T[f][a]
    float* s_dn = s_dT;                            // [128][33]  D[a][f]       (after stage 2)
    float* s_dp = s_dT + 32 * 144;                 // [32][144]  d_pre[t][a]
    float* s_de = s_dp + ATTB_CH * 144;            // [32]
    float* s_red = s_de + ATTB_CH;                 // [8]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lr = lane & 15, lq = lane >> 4;
    const int b = blockIdx.y, chunk = blockIdx.x, t0 = chunk * ATTB_CH;
    const int T = a.T, KS = a.loc_ks, pad = KS >> 1, K2 = 2 * KS;
    const int len = a.lengths ? a.lengths[b] : T;
    const size_t slot = (size_t)b * gridDim.x + chunk;
    // ---- loads ----
    for (int i = tid; i < 64 * 48; i += 512) s_kb[i] = 0.f;
    for (int i = tid; i < 2 * (ATTB_CH + KS - 1); i += 512) {
        const int c = i / (ATTB_CH + KS - 1), j = i - c * (ATTB_CH + KS - 1);
        const int t = t0 + j - pad;
        const float* src = c ? a.wc_prev : a.w_prev;
        s_cat[c * (ATTB_CH + 64) + j] = (src && t >= 0 && t < T) ? src[(size_t)b * (c ? a.s_wcprev : a.s_wprev) + t] : 0.f;
    }
    float rk[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int i = tid + j * 512;
        rk[j] = i < F * K2 ? a.w_loc_conv[i] : 0.f;
    }
    float rd[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {                  // D [a][f], 4096 floats, coalesced
        const int i = tid + j * 512;
        rd[j] = a.w_loc_dense[i];
        s_dT[(i & 31) * 144 + (i >> 5)] = rd[j];
    }
    // stage-2 / stage-3 ownership of this wave: attention channels 16 wave .. 16 wave + 15 (tile column `wave`)
    const int ach = 16 * wave + lr;
    const float qv = a.q[(size_t)b * AD + ach], vv = a.w_v[ach];
    float pm[2][4], dpm[2][4], dD_old[2][4];
#pragma unroll
    for (int tt = 0; tt < 2; ++tt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int t = t0 + 16 * tt + 4 * lq + r;
            const size_t po = ((size_t)b * T + (t < T ? t : 0)) * AD + ach;
            pm[tt][r] = a.pmem[po];
            dpm[tt][r] = a.d_pmem[po];
            dD_old[tt][r] = a.dD_part[slot * AD * F + (size_t)(16 * tt + 4 * lq + r) * AD + ach];      // (here tt = f tile)
        }
    const float dv_old = a.dv_part[slot * AD + ach];
    // softmax backward needs sum_t w[t] d_w[t] over the whole row
    float part = 0.f;
    for (int t = tid; t < T; t += 512) part += a.w_cur[(size_t)b * a.s_wcur + t] * a.dw_buf[(size_t)b * T + t];
    part = attb_wave_sum(part);
    if (lane == 0) s_red[wave] = part;
    __syncthreads();                               // s_kb zeroed, s_red complete
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int i = tid + j * 512;
        if (i < F * K2) {
            const int f = i / K2, k = i - f * K2;  // K[f][c][j] -> B operand [k = c * KS + j][f]
            s_kb[k * 48 + f] = rk[j];
        }
    }
    const float sdot = ((s_red[0] + s_red[1]) + (s_red[2] + s_red[3])) + ((s_red[4] + s_red[5]) + (s_red[6] + s_red[7]));
    if (tid < ATTB_CH) {
        const int t = t0 + tid;
        s_de[tid] = (t < T && t < len) ? a.w_cur[(size_t)b * a.s_wcur + t] * (a.dw_buf[(size_t)b * T + t] - sdot) : 0.f;
    }
    __syncthreads();
    // ---- stage 1: location features, 4 tiles (t tile, f tile) on waves 0-3 ----
    if (wave < 4) {
        const int tt = wave >> 1, ft = wave & 1;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        float av[16], bv[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int k = 4 * u + lq;
            const int kc = k < K2 ? k : 0;         // (B rows >= 2 KS are zero)
            const int c = kc >= KS ? 1 : 0, j = kc - c * KS;
            av[u] = s_cat[c * (ATTB_CH + 64) + 16 * tt + lr + j];
            bv[u] = s_kb[k * 48 + 16 * ft + lr];
        }
#pragma unroll
        for (int u = 0; u < 16; ++u) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u], bv[u], acc, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 4; ++r) s_f[(16 * tt + 4 * lq + r) * 33 + 16 * ft + lr] = acc[r];
    }
    __syncthreads();
    // ---- stage 2: P = F D^T, energies backward in the epilogue; this wave: channels `ach`, both position tiles ----
    float dq = 0.f, dvs = 0.f;
#pragma unroll
    for (int tt = 0; tt < 2; ++tt) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        float av[8], bv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            av[u] = s_f[(16 * tt + lr) * 33 + 4 * u + lq];
            bv[u] = s_dT[(4 * u + lq) * 144 + ach];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u], bv[u], acc, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int tq = 16 * tt + 4 * lq + r, t = t0 + tq;
            float dp = 0.f;
            if (t < T) {
                const float th = tanhf(acc[r] + qv + pm[tt][r]);
                const float de = s_de[tq];
                dp = de * vv * (1.f - th * th);
                dq += dp;
                dvs += de * th;
                a.d_pmem[((size_t)b * T + t) * AD + ach] = dpm[tt][r] + dp;
            }
            s_dp[tq * 144 + ach] = dp;
            s_dpT[ach * 33 + tq] = dp;
        }
    }
    dq += __shfl_xor(dq, 16, 64);
    dq += __shfl_xor(dq, 32, 64);
    dvs += __shfl_xor(dvs, 16, 64);
    dvs += __shfl_xor(dvs, 32, 64);
    if (lq == 0) {
        a.dq_part[slot * AD + ach] = dq;
        a.dv_part[slot * AD + ach] = dv_old + dvs;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int i = tid + j * 512;
        s_dn[(i >> 5) * 33 + (i & 31)] = rd[j];
    }
    // ---- stage 3: dD^T[f][a] += F^T d_pre; this wave: channels `ach`, both filter tiles ----
#pragma unroll
    for (int ft = 0; ft < 2; ++ft) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        float av[8], bv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            av[u] = s_f[(4 * u + lq) * 33 + 16 * ft + lr];          // A[row f][k = t] = F[t][f]
            bv[u] = s_dp[(4 * u + lq) * 144 + ach];                // B[k = t][col a]
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u], bv[u], acc, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 4; ++r)
            a.dD_part[slot * AD * F + (size_t)(16 * ft + 4 * lq + r) * AD + ach] = dD_old[ft][r] + acc[r];
    }
    __syncthreads();
    // ---- stage 4: d_f[t][f] = d_pre D, 4 tiles (t tile, f tile) on waves 0-3, K = 128 ----
    if (wave < 4) {
        const int tt = wave >> 1, ft = wave & 1;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            float av[16], bv[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const int k = 64 * h + 4 * u + lq;
                av[u] = s_dpT[k * 33 + 16 * tt + lr];               // A[row t][k = a] = d_pre[t][a]
                bv[u] = s_dn[k * 33 + 16 * ft + lr];                // B[k = a][col f] = D[a][f]
            }
#pragma unroll
            for (int u = 0; u < 16; ++u) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u], bv[u], acc, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int t = t0 + 16 * tt + 4 * lq + r;
            if (t < T) a.df_buf[((size_t)b * T + t) * 32 + 16 * ft + lr] = acc[r];
        }
    }
}

__global__ __launch_bounds__(256) void att_bwd_conv_kernel(const AttBwdArgs a) {
    __shared__ float s_cat[2][ATTB_CH + 64];
    __shared__ float s_k[32 * 2 * 63];
    __shared__ float s_df[ATTB_CH + 64][33];
    const int tid = threadIdx.x;
    const int b = blockIdx.y, chunk = blockIdx.x, t0 = chunk * ATTB_CH, nchunk = gridDim.x;
    const int T = a.T, AD = a.att_dim, KS = a.loc_ks, F = a.loc_f, pad = KS >> 1;
    const size_t slot = (size_t)b * nchunk + chunk;
    for (int i = tid; i < F * 2 * KS; i += 256) s_k[i] = a.w_loc_conv[i];
    for (int i = tid; i < 2 * (ATTB_CH + KS - 1); i += 256) {
        const int c = i / (ATTB_CH + KS - 1), j = i - c * (ATTB_CH + KS - 1);
        const int t = t0 + j - pad;
        const float* src = c ? a.wc_prev : a.w_prev;
        s_cat[c][j] = (src && t >= 0 && t < T) ? src[(size_t)b * (c ? a.s_wcprev : a.s_wprev) + t] : 0.f;
    }
    for (int i = tid; i < (ATTB_CH + KS - 1) * 32; i += 256) {
        const int j = i >> 5, f = i & 31;
        const int t = t0 + j - pad;
        s_df[j][f] = (t >= 0 && t < T && f < F) ? a.df_buf[((size_t)b * T + t) * 32 + f] : 0.f;
    }
    __syncthreads();
    // carries: d cat[c][tp] = sum_f sum_j K[f][c][j] d_f[tp - j + pad][f]; 64 outputs x 4 threads (8 filters each)
    {
        const int o = tid >> 2, fg = tid & 3;
        const int c = o >> 5, tl = o & 31;
        float acc = 0.f;
        for (int f = fg * 8; f < fg * 8 + 8 && f < F; ++f)
#pragma unroll 8
            for (int j = 0; j < KS; ++j) acc += s_k[(f * 2 + c) * KS + j] * s_df[tl - j + 2 * pad][f];   // row (tp - j + pad) - (t0 - pad)
        acc += __shfl_xor(acc, 1, 64);
        acc += __shfl_xor(acc, 2, 64);
        const int tp = t0 + tl;
        if (fg == 0 && tp < T) {
            if (c == 0) a.dw_carry[(size_t)b * T + tp] = acc;
            else a.dwc_carry[(size_t)b * T + tp] += acc;
        }
    }
    // kernel gradient partial: dK[f][c][j] += sum_{t in chunk} d_f[t][f] cat[c][t + j - pad]
    for (int i = tid; i < F * 2 * KS; i += 256) {
        const int f = i / (2 * KS), r = i - f * 2 * KS, c = r / KS, j = r - c * KS;
        const float old = a.dK_part[slot * F * 2 * KS + i];
        float acc = 0.f;
#pragma unroll 8
        for (int tl = 0; tl < ATTB_CH; ++tl) acc += s_df[tl + pad][f] * s_cat[c][tl + j];
        a.dK_part[slot * F * 2 * KS + i] = old + acc;
    }
    if (chunk == 0 && tid < AD) {
        float sum = 0.f;
        for (int ch = 0; ch < nchunk; ++ch) sum += a.dq_part[((size_t)b * nchunk + ch) * AD + tid];
        a.d_q[(size_t)b * AD + tid] = sum;
    }
}

// The same stage on the exact-f32 matrix cores, for 32 location filters and kernel <= 31.  Both sums are small matrix products
// once the shift is taken out of the carries (profiles/r03_taco_step_kernel_counters_before.json: the VALU kernel spends a
// third of its LDS cycles in bank conflicts and 22 us per decoder step):
//   G[m = (c, j)][row] = sum_f K[f][c][j] d_f[row][f]            [64 x 32] x [32 x 64]; then d cat[c][tp] = sum_j G[(c, j)][tp - j + 2 pad]
//   dK[f][m = (c, j)] += sum_tl d_f[tl + pad][f] cat[c][tl + j]  [32 x 32] x [32 x 64]  (B read straight from the window: Toeplitz)
__global__ __launch_bounds__(256) void att_bwd_conv_mfma_kernel(const AttBwdArgs a) {
    constexpr int F = 32, SK = 80, SD = 34, SG = 66;
    __shared__ float s_cat[2][ATTB_CH + 64];
    __shared__ float s_k[F * SK];                 // K[f][m], columns >= 2 KS zero
    __shared__ float s_df[64 * SD];               // d_f[row][f], row <-> t0 - pad + row; rows past the window zero
    __shared__ float s_g[64 * SG];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lr = lane & 15, lq = lane >> 4;
    const int b = blockIdx.y, chunk = blockIdx.x, t0 = chunk * ATTB_CH, nchunk = gridDim.x;
    const int T = a.T, AD = a.att_dim, KS = a.loc_ks, pad = KS >> 1, K2 = 2 * KS;
    const size_t slot = (size_t)b * nchunk + chunk;
    for (int i = tid; i < F * SK; i += 256) {
        const int f = i / SK, m = i - f * SK;
        s_k[i] = m < K2 ? a.w_loc_conv[f * K2 + m] : 0.f;
    }
    for (int i = tid; i < 2 * (ATTB_CH + KS - 1); i += 256) {
        const int c = i / (ATTB_CH + KS - 1), j = i - c * (ATTB_CH + KS - 1);
        const int t = t0 + j - pad;
        const float* src = c ? a.wc_prev : a.w_prev;
        s_cat[c][j] = (src && t >= 0 && t < T) ? src[(size_t)b * (c ? a.s_wcprev : a.s_wprev) + t] : 0.f;
    }
    for (int i = tid; i < 64 * 32; i += 256) {
        const int j = i >> 5, f = i & 31;
        const int t = t0 + j - pad;
        s_df[j * SD + f] = (j < ATTB_CH + KS - 1 && t >= 0 && t < T) ? a.df_buf[((size_t)b * T + t) * 32 + f] : 0.f;
    }
    // this wave's columns m = 16 wave + lr of the kernel gradient, fetched early (read-modify-write at the end)
    const int mcol = 16 * wave + lr;
    float dK_old[2][4];
#pragma unroll
    for (int ft = 0; ft < 2; ++ft)
#pragma unroll
        for (int r = 0; r < 4; ++r)
            dK_old[ft][r] = mcol < K2 ? a.dK_part[slot * F * K2 + (size_t)(16 * ft + 4 * lq + r) * K2 + mcol] : 0.f;
    __syncthreads();
    // ---- G: this wave owns rows m = 16 wave .. 16 wave + 15, all four row tiles of d_f ----
    {
        float av[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) av[u] = s_k[(4 * u + lq) * SK + 16 * wave + lr];       // A[row m][k = f]
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            float bv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) bv[u] = s_df[(16 * nt + lr) * SD + 4 * u + lq];    // B[k = f][col row]
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int u = 0; u < 8; ++u) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u], bv[u], acc, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 4; ++r) s_g[(16 * wave + 4 * lq + r) * SG + 16 * nt + lr] = acc[r];
        }
    }
    // ---- kernel gradient partial: this wave owns columns m = 16 wave + lr, both filter tiles ----
    {
        const int mc = mcol < K2 ? mcol : 0;
        const int c = mc >= KS ? 1 : 0, j = mc - c * KS;
        float bv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) bv[u] = s_cat[c][4 * u + lq + j];                       // B[k = tl][col m] = cat[c][tl + j]
#pragma unroll
        for (int ft = 0; ft < 2; ++ft) {
            float av[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) av[u] = s_df[(4 * u + lq + pad) * SD + 16 * ft + lr];   // A[row f][k = tl] = d_f[tl + pad][f]
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int u = 0; u < 8; ++u) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u], bv[u], acc, 0, 0, 0);
            if (mcol < K2)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    a.dK_part[slot * F * K2 + (size_t)(16 * ft + 4 * lq + r) * K2 + mcol] = dK_old[ft][r] + acc[r];
        }
    }
    __syncthreads();
    // ---- carries: 64 outputs x 4 threads (8 taps each) ----
    {
        const int o = tid >> 2, jg = tid & 3;
        const int c = o >> 5, tl = o & 31;
        float acc = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int j = jg * 8 + i;
            if (j < KS) acc += s_g[(c * KS + j) * SG + tl - j + 2 * pad];
        }
        acc += __shfl_xor(acc, 1, 64);
        acc += __shfl_xor(acc, 2, 64);
        const int tp = t0 + tl;
        if (jg == 0 && tp < T) {
            if (c == 0) a.dw_carry[(size_t)b * T + tp] = acc;
            else a.dwc_carry[(size_t)b * T + tp] += acc;
        }
    }
    if (chunk == 0 && tid < AD) {
        float sum = 0.f;
        for (int ch = 0; ch < nchunk; ++ch) sum += a.dq_part[((size_t)b * nchunk + ch) * AD + tid];
        a.d_q[(size_t)b * AD + tid] = sum;
    }
}

// ------------------------------------------------------------------------------------------------
// The three parts above in ONE launch (attention_dim 128, 32 location filters, kernel <= 31, deferred d_memory).  What kept them
// apart were two exchanges across workgroups; both go away:
//   * softmax backward needs sdot = sum_t w[t] d_w[t] over the whole row.  With d_w[t] = mem[t] . d_ctx + carry[t] and the saved
//     context ctx = sum_t w[t] mem[t]:  sdot = ctx . d_ctx + sum_t w[t] carry[t]  - 512 + T multiply-adds, no pass over memory;
//   * the location-conv backward needs d_f on a halo of kernel/2 positions either side of a chunk.  Turned round: a workgroup
//     scatters what ITS 32 rows of d_f contribute to the carries of the 32 + 2 pad positions they reach, into three slots per
//     position - [0] the chunk's own part, [1] the part from the chunk to the right, [2] from the chunk to the left - and the
//     next step's launch adds the slots that exist for a position (a fixed rule, a fixed order: bitwise reproducible, no
//     atomics).  A first version recomputed d_w .. d_f on the halo instead: twice the loads per CU (250 KB at the ~50 GB/s a
//     CU draws from L2) made its prologue 5 of 19 us (profiles/r03_att_bwd_fused_halo_stamps.txt).
// Carry buffers are [3][B][T] and come in two sets: a step reads one and writes the other.
#ifdef T2S_ATTF_STAMPS          // diagnostic build: phase boundaries of workgroup (0, 0), 100 MHz ticks (tools/attf_stamps.py)
__device__ unsigned long long t2s_attf_stamps[16];
#define ATTF_STAMP(i) if (tid == 0 && blockIdx.x == 0 && blockIdx.y == 0) t2s_attf_stamps[i] = __builtin_amdgcn_s_memrealtime();
extern "C" int t2s_debug_read_attf_stamps(unsigned long long* host_out) {
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(t2s_attf_stamps), sizeof(unsigned long long) * 16);
}
#else
#define ATTF_STAMP(i)
#endif
// total carry at position t of batch row b: own slot + the neighbours' slots that exist for it
static __device__ __forceinline__ float attf_carry(const float* c, size_t BT, size_t bt, int t, int T, int pad) {
    const int l = t & (ATTB_CH - 1), t0 = t - l;
    float v = c[bt];
    if (l >= ATTB_CH - pad && t0 + ATTB_CH < T) v += c[BT + bt];
    if (l < pad && t0 > 0) v += c[2 * BT + bt];
    return v;
}
#define ATTF_SP 130
#define ATTF_SPIN_MAX (1 << 22)
#ifdef T2S_ATTB_LB2             // A/B build: two workgroups per CU by registers (128 VGPRs instead of 185, 188 bytes of scratch per thread):
                                // 89.3 / 90.3 against 74.1 / 73.8 ms per train step (profiles/r04_attb_lb2_ab.txt) - the spills cost far more
                                // than sharing CUs with the helper chain's GEMMs could return
#define ATTB_LB __launch_bounds__(512, 4)
#else
#define ATTB_LB __launch_bounds__(512)
#endif
__global__ ATTB_LB void att_bwd_fused_kernel(const AttBwdArgs a, const AttBwdFoldArgs fold) {
    constexpr int AD = 128, F = 32, SP = ATTF_SP, SD = 34, SG = 34;
    __shared__ __attribute__((aligned(16))) float s_all[128 + 64 * 48 + ATTB_CH * 33 + 32 * 144 + ATTB_CH * ATTF_SP + ATTB_CH * 34 + 1024 + 2 * ATTB_CH + 16];
    float* s_cat = s_all;                          // [2][64]    window of [w_prev ; wc_prev]: entry i <-> t0 - pad + i
    float* s_kb = s_cat + 128;                     // [64][48]   conv kernel as B operand [k = c * KS + j][f], rows >= 2 KS zero
    float* s_g = s_kb;                             // [64][34]   G[m][row]    (after the features)
    float* s_f = s_kb + 64 * 48;                   // [32][33]   F[row][f]
    float* s_dT = s_f + ATTB_CH * 33;              // [32][144]  D^T[f][a]
    float* s_dn = s_dT;                            // [128][33]  D[a][f]      (after the energies)
    float* s_dp = s_dT + 32 * 144;                 // [32][130]  d_pre[row][a]
    float* s_df = s_dp + ATTB_CH * SP;             // [32][34]   d_f[row][f]
    float* s_dctx = s_df + ATTB_CH * SD;           // [<= 1024]
    float* s_dw = s_dctx + 1024;                   // [32]
    float* s_de = s_dw + ATTB_CH;                  // [32]
    float* s_red = s_de + ATTB_CH;                 // [16]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lr = lane & 15, lq = lane >> 4;
    // block number -> (batch element, chunk): id = 8 s + x is chunk s % n_chunks of element 8 (s / n_chunks) + x - the chunks of an
    // element share one XCD's L2 (workgroups go round the 8 XCDs by block number) and are neighbours in dispatch order (they wait
    // for one another when the cell backward is folded in, below)
#ifdef T2S_ATTB_SETPRIO
    __builtin_amdgcn_s_setprio(3);                 // (A/B build: this kernel is the backward's serial chain)
#endif
    if (fold.sig_ptr && blockIdx.x == 0 && threadIdx.x == 0)          // "this launch has started"
        __hip_atomic_store(fold.sig_ptr, fold.sig_val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int n_chunks = (a.T + ATTB_CH - 1) / ATTB_CH;
    const int bsl = blockIdx.x >> 3;
    const int b = (bsl / n_chunks) * 8 + (blockIdx.x & 7), chunk = bsl - (bsl / n_chunks) * n_chunks, t0 = chunk * ATTB_CH;
    if (b >= a.B) return;                          // (whole workgroups)
    const int T = a.T, E = a.enc_dim, KS = a.loc_ks, pad = KS >> 1, K2 = 2 * KS;
    const int len = a.lengths ? a.lengths[b] : T;
    const size_t slot = (size_t)b * n_chunks + chunk, BT = (size_t)a.B * T;
    ATTF_STAMP(0)
    // ---- loads: everything the kernel reads from global memory is requested here, unconditionally (clamped addresses, selects
    //      afterwards) and before the first use of any of it: one round trip to memory, not one per dependent group ----
    // (d_ctx sources: a null one reads the first non-null one and is dropped by a select - no branch between the loads)
    const float* dany = a.dctx1 ? a.dctx1 : (a.dctx2 ? a.dctx2 : a.dctx3);
    const long sany = a.dctx1 ? a.sc1 : (a.dctx2 ? a.sc2 : a.sc3);
    float dsrc[2][3];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int j = tid + 512 * h < E ? tid + 512 * h : 0;
        dsrc[h][0] = (a.dctx1 ? a.dctx1 : dany)[(size_t)b * (a.dctx1 ? a.sc1 : sany) + j];
        dsrc[h][1] = (a.dctx2 ? a.dctx2 : dany)[(size_t)b * (a.dctx2 ? a.sc2 : sany) + j];
        dsrc[h][2] = (a.dctx3 ? a.dctx3 : dany)[(size_t)b * (a.dctx3 ? a.sc3 : sany) + j];
    }
    // d_w operands: wave w takes rows 4 w .. 4 w + 3, channels on lanes (registers for enc_dim 512, else a loop below)
    const bool e512 = E == 512;
    f32x4 mrow[2][4];
    if (e512) {
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int t = t0 + 4 * wave + r;
                mrow[h][r] = *(const f32x4*)(a.memory + ((size_t)b * T + (t < T ? t : T - 1)) * E + lane * 4 + 256 * h);
            }
    }
    float ctx_r[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) ctx_r[h] = a.ctx[(size_t)b * a.s_ctx + (tid + 512 * h < E ? tid + 512 * h : 0)];
    // carries: position `tid` of the row (the sdot term) and position t0 + (tid & 31) (the chunk's rows), three slots each
    const int ta = tid < T ? tid : T - 1, tb = t0 + (tid & 31) < T ? t0 + (tid & 31) : T - 1;
    const size_t bta = (size_t)b * T + ta, btb = (size_t)b * T + tb;
    const float wa = a.w_cur[(size_t)b * a.s_wcur + ta], wb = a.w_cur[(size_t)b * a.s_wcur + tb];
    float ca[2][3], cb[2][3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        ca[0][k] = a.dw_carry[k * BT + bta]; ca[1][k] = a.dwc_carry[k * BT + bta];
        cb[0][k] = a.dw_carry[k * BT + btb]; cb[1][k] = a.dwc_carry[k * BT + btb];
    }
    float rk[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int i = tid + j * 512;
        rk[j] = i < F * K2 ? a.w_loc_conv[i] : 0.f;
    }
    float rd[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {                  // D [a][f], 4096 floats, coalesced
        const int i = tid + j * 512;
        rd[j] = a.w_loc_dense[i];
    }
    const int ach = 16 * wave + lr;                // this lane's attention channel in the energies / dD stages
    const float qv = a.q[(size_t)b * AD + ach], vv = a.w_v[ach];
    float pm[2][4], dpm[2][4];
#pragma unroll
    for (int tt = 0; tt < 2; ++tt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int t = t0 + 16 * tt + 4 * lq + r;
            const size_t po = ((size_t)b * T + (t < T ? t : T - 1)) * AD + ach;
            pm[tt][r] = a.pmem[po];
            dpm[tt][r] = a.d_pmem[po];
        }
    float dD_old[2][4];
#pragma unroll
    for (int ft = 0; ft < 2; ++ft)
#pragma unroll
        for (int r = 0; r < 4; ++r) dD_old[ft][r] = a.dD_part[slot * AD * F + (size_t)(16 * ft + 4 * lq + r) * AD + ach];
    const float dv_old = a.dv_part[slot * AD + ach];
    // the conv-backward tiles of this wave: G rows m = 16 (wave >> 1) + lr (all), kernel-gradient filters 16 (wave & 1) ..
    const int mcol = 16 * (wave >> 1) + lr, kft = wave & 1;
    float dK_old[4], ak[8];
#pragma unroll
    for (int r = 0; r < 4; ++r) dK_old[r] = mcol < K2 ? a.dK_part[slot * F * K2 + (size_t)(16 * kft + 4 * lq + r) * K2 + mcol] : 0.f;
#pragma unroll
    for (int u = 0; u < 8; ++u) ak[u] = mcol < K2 ? a.w_loc_conv[(4 * u + lq) * K2 + mcol] : 0.f;       // G stage: A[row m][k = f]
    // (last: hipcc keeps this one behind a branch with a full wait, which here coincides with the wait for everything)
    float cat_r;
    bool cat_ok;
    {
        const int c = (tid >> 6) & 1, j = tid & 63;
        const int t = t0 - pad + j;
        const float* src = c ? a.wc_prev : a.w_prev;                 // null at decoder step 0: read w_cur, drop it
        const long ss = src ? (c ? a.s_wcprev : a.s_wprev) : a.s_wcur;
        cat_r = (src ? src : a.w_cur)[(size_t)b * ss + (t < 0 ? 0 : (t < T ? t : T - 1))];
        cat_ok = src && j < ATTB_CH + KS - 1 && t >= 0 && t < T;     // (applied at the LDS store: no write to a register in flight)
    }
    // ---- first uses ----
    float dctx_r[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        float v = 0.f;                              // (same order of additions as sum3)
        v += a.dctx1 ? dsrc[h][0] : 0.f;
        v += a.dctx2 ? dsrc[h][1] : 0.f;
        v += a.dctx3 ? dsrc[h][2] : 0.f;
        dctx_r[h] = v;
        if (tid + 512 * h < E) s_dctx[tid + 512 * h] = v;
    }
    for (int i = tid; i < 64 * 48; i += 512) s_kb[i] = 0.f;
    if (tid < 128) s_cat[tid] = cat_ok ? cat_r : 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int i = tid + j * 512;
        s_dT[(i & 31) * 144 + (i >> 5)] = rd[j];
    }
    // total carry = own slot + the neighbours' slots that exist for the position
    float wc_r = 0.f, row_w = 0.f, row_c = 0.f, row_dwc = 0.f;
    {
        const int la = ta & (ATTB_CH - 1), lb = tb & (ATTB_CH - 1);
        const bool ra = la >= ATTB_CH - pad && ta - la + ATTB_CH < T, lfa = la < pad && ta - la > 0;
        const bool rb = lb >= ATTB_CH - pad && tb - lb + ATTB_CH < T, lfb = lb < pad && tb - lb > 0;
        const float sa = (ca[0][0] + (ra ? ca[0][1] : 0.f) + (lfa ? ca[0][2] : 0.f)) + (ca[1][0] + (ra ? ca[1][1] : 0.f) + (lfa ? ca[1][2] : 0.f));
        if (tid < T) wc_r = wa * sa;
        for (int t = tid + 512; t < T; t += 512)    // (rows longer than 512)
            wc_r += a.w_cur[(size_t)b * a.s_wcur + t] *
                    (attf_carry(a.dw_carry, BT, (size_t)b * T + t, t, T, pad) + attf_carry(a.dwc_carry, BT, (size_t)b * T + t, t, T, pad));
        if (tid < ATTB_CH && t0 + tid < T) {
            row_w = wb;
            row_dwc = cb[1][0] + (rb ? cb[1][1] : 0.f) + (lfb ? cb[1][2] : 0.f);
            row_c = (cb[0][0] + (rb ? cb[0][1] : 0.f) + (lfb ? cb[0][2] : 0.f)) + row_dwc;
        }
    }
    ATTF_STAMP(1)
    __syncthreads();                               // s_dctx, s_kb zeros
    ATTF_STAMP(2)
    if (chunk == 0) {
#pragma unroll
        for (int h = 0; h < 2; ++h)
            if (tid + 512 * h < E) a.dctx_out[(size_t)b * E + tid + 512 * h] = dctx_r[h];
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int i = tid + j * 512;
        if (i < F * K2) {
            const int f = i / K2, k = i - f * K2;  // K[f][c][j] -> B operand [k = c * KS + j][f]
            s_kb[k * 48 + f] = rk[j];
        }
    }
    // ---- d_w of the chunk's 32 rows (without the carries) ----
    {
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        if (e512) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const f32x4 dc = *(const f32x4*)&s_dctx[lane * 4 + 256 * h];
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    acc[r] += mrow[h][r][0] * dc[0] + mrow[h][r][1] * dc[1] + mrow[h][r][2] * dc[2] + mrow[h][r][3] * dc[3];
            }
        } else {
            for (int c = lane * 4; c < E; c += 256) {
                const f32x4 dc = *(const f32x4*)&s_dctx[c];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int t = t0 + 4 * wave + r;
                    const f32x4 m = *(const f32x4*)(a.memory + ((size_t)b * T + (t < T ? t : T - 1)) * E + c);
                    acc[r] += m[0] * dc[0] + m[1] * dc[1] + m[2] * dc[2] + m[3] * dc[3];
                }
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float v = attb_wave_sum(acc[r]);
            if (lane == 0) s_dw[4 * wave + r] = v;
        }
        float part = wc_r;
#pragma unroll
        for (int h = 0; h < 2; ++h)
            if (tid + 512 * h < E) part += ctx_r[h] * s_dctx[tid + 512 * h];
        part = attb_wave_sum(part);
        if (lane == 0) s_red[wave] = part;
    }
    ATTF_STAMP(3)
    __syncthreads();                               // s_kb, s_dw, s_red
    ATTF_STAMP(4)
    if (tid < ATTB_CH) {
        const float sdot = ((s_red[0] + s_red[1]) + (s_red[2] + s_red[3])) + ((s_red[4] + s_red[5]) + (s_red[6] + s_red[7]));
        const int t = t0 + tid;
        s_de[tid] = (t < T && t < len) ? row_w * (s_dw[tid] + row_c - sdot) : 0.f;
    }
    // ---- location features: 4 tiles (t tile, f tile) on waves 0-3 ----
    if (wave < 4) {
        const int tt = wave >> 1, ft = wave & 1;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        float av[16], bv[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int k = 4 * u + lq;
            const int kc = k < K2 ? k : 0;         // (B rows >= 2 KS are zero)
            const int c = kc >= KS ? 1 : 0, j = kc - c * KS;
            av[u] = s_cat[c * 64 + 16 * tt + lr + j];
            bv[u] = s_kb[k * 48 + 16 * ft + lr];
        }
#pragma unroll
        for (int u = 0; u < 16; ++u) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u], bv[u], acc, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 4; ++r) s_f[(16 * tt + 4 * lq + r) * 33 + 16 * ft + lr] = acc[r];
    }
    ATTF_STAMP(5)
    __syncthreads();                               // s_f, s_de
    ATTF_STAMP(6)
    // ---- energies backward: P = F D^T for channels `ach`, two row tiles ----
    {
        float bT[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) bT[u] = s_dT[(4 * u + lq) * 144 + ach];
        float dq = 0.f, dvs = 0.f;
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            float av[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) av[u] = s_f[(16 * tt + lr) * 33 + 4 * u + lq];
#pragma unroll
            for (int u = 0; u < 8; ++u) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u], bT[u], acc, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = 16 * tt + 4 * lq + r, t = t0 + row;
                const float th = tanhf(acc[r] + qv + pm[tt][r]);
                const float de = s_de[row];        // 0 outside [0, min(T, len))
                const float dp = de * vv * (1.f - th * th);
                s_dp[row * SP + ach] = dp;
                dq += dp;
                dvs += de * th;
                if (t < T) a.d_pmem[((size_t)b * T + t) * AD + ach] = dpm[tt][r] + dp;
            }
        }
        dq += __shfl_xor(dq, 16, 64);
        dq += __shfl_xor(dq, 32, 64);
        dvs += __shfl_xor(dvs, 16, 64);
        dvs += __shfl_xor(dvs, 32, 64);
        if (lq == 0) {
            a.dq_part[slot * AD + ach] = dq;
            a.dv_part[slot * AD + ach] = dv_old + dvs;
            if (fold.xbuf)                         // this chunk's partial d_q for the other chunks of the element (read at the end)
                __hip_atomic_store(fold.xbuf + slot * AD + ach, ((unsigned long long)fold.tag << 32) | (unsigned long long)__float_as_uint(dq),
                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    ATTF_STAMP(7)
    __syncthreads();                               // s_dp complete, s_dT free
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int i = tid + j * 512;
        s_dn[(i >> 5) * 33 + (i & 31)] = rd[j];
    }
    // ---- dD^T[f][a] += sum over the rows F[row][f] d_pre[row][a] ----
#pragma unroll
    for (int ft = 0; ft < 2; ++ft) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        float av[8], bv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            av[u] = s_f[(4 * u + lq) * 33 + 16 * ft + lr];
            bv[u] = s_dp[(4 * u + lq) * SP + ach];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u], bv[u], acc, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 4; ++r)
            a.dD_part[slot * AD * F + (size_t)(16 * ft + 4 * lq + r) * AD + ach] = dD_old[ft][r] + acc[r];
    }
    ATTF_STAMP(8)
    __syncthreads();                               // s_dn
    // ---- d_f[row][f] = d_pre D: tile (wave >> 1 & 1, wave & 1), K = 128 split over the two wave groups (waves 4-7: a >= 64) ----
    {
        const int tt = (wave >> 1) & 1, ft = wave & 1, h = wave >> 2;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        float av[16], bv[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int k = 64 * h + 4 * u + lq;
            av[u] = s_dp[(16 * tt + lr) * SP + k];
            bv[u] = s_dn[k * 33 + 16 * ft + lr];
        }
#pragma unroll
        for (int u = 0; u < 16; ++u) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u], bv[u], acc, 0, 0, 0);
        if (h == 1) {
#pragma unroll
            for (int r = 0; r < 4; ++r) s_df[(16 * tt + 4 * lq + r) * SD + 16 * ft + lr] = acc[r];
        }
        __syncthreads();                           // the upper half's partial sums
        if (h == 0) {
#pragma unroll
            for (int r = 0; r < 4; ++r) s_df[(16 * tt + 4 * lq + r) * SD + 16 * ft + lr] += acc[r];
        }
    }
    ATTF_STAMP(9)
    __syncthreads();                               // s_df
    // ---- location-conv backward.  G[m][row] = sum_f K[f][m] d_f[row][f]: rows m = 16 (wave >> 1) .., row tile wave & 1 ----
    {
        const int nt = wave & 1;
        float bv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) bv[u] = s_df[(16 * nt + lr) * SD + 4 * u + lq];
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int u = 0; u < 8; ++u) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(ak[u], bv[u], acc, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 4; ++r) s_g[(16 * (wave >> 1) + 4 * lq + r) * SG + 16 * nt + lr] = acc[r];
        // kernel gradient: dK[f][m] += sum_tl d_f[tl][f] cat[c][tl + j]
        const int mc = mcol < K2 ? mcol : 0;
        const int c = mc >= KS ? 1 : 0, j = mc - c * KS;
        f32x4 acck = {0.f, 0.f, 0.f, 0.f};
        float av[8], bk[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            av[u] = s_df[(4 * u + lq) * SD + 16 * kft + lr];
            bk[u] = s_cat[c * 64 + 4 * u + lq + j];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) acck = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u], bk[u], acck, 0, 0, 0);
        if (mcol < K2)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                a.dK_part[slot * F * K2 + (size_t)(16 * kft + 4 * lq + r) * K2 + mcol] = dK_old[r] + acck[r];
    }
    ATTF_STAMP(10)
    __syncthreads();                               // s_g
    // ---- what the chunk's rows contribute to the carries of step t - 1: positions q <-> t0 - pad + q, q < 32 + 2 pad; output
    //      (c, q) on 4 threads (8 taps each):  P[c][q] = sum_j G[(c, j)][q - j] over the rows 0 <= q - j < 32 ----
    {
        const int o = tid >> 2, jg = tid & 3;
        const int c = o >= 64, q = o - 64 * c;      // (o < 128; q < 32 + 2 pad <= 62 used)
        float acc = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int j = jg * 8 + i, row = q - j;
            if (j < KS && row >= 0 && row < ATTB_CH) acc += s_g[(c * KS + j) * SG + row];
        }
        acc += __shfl_xor(acc, 1, 64);
        acc += __shfl_xor(acc, 2, 64);
        // the running dwc total of the chunk's own positions sits in threads 0-31 (row_dwc); the threads that write the own slot
        // are spread over all waves, so it travels through LDS (s_dw is free by now)
        if (tid < ATTB_CH) s_dw[tid] = row_dwc;
        __syncthreads();
        const int tp = t0 - pad + q;
        if (jg == 0 && q < ATTB_CH + 2 * pad && tp >= 0 && tp < T) {
            float* dst = c ? a.dwc_carry_out : a.dw_carry_out;
            const size_t bt = (size_t)b * T + tp;
            if (q < pad) dst[BT + bt] = acc;                                     // reaches the chunk to the left: its slot [1]
            else if (q >= pad + ATTB_CH) dst[2 * BT + bt] = acc;                // the chunk to the right: its slot [2]
            else dst[bt] = c ? s_dw[q - pad] + acc : acc;                       // own slot (dwc: + the running total)
        }
    }
    ATTF_STAMP(11)
    // ---- folded in: the attention LSTMCell's pointwise backward (lstm_cell_bwd_q_kernel's arithmetic) for this chunk's share of
    //      the hidden units.  d_q = the sum of every chunk's partial, in chunk order; the partials were published half a kernel ago,
    //      so the wait below is normally over before it starts.  One dependent launch less per step of the BPTT loop. ----
    if (fold.xbuf) {                               // (uniform)
        const LstmBwdArgs& ca = fold.cell;
        __shared__ int s_fail;
        if (tid == 0) s_fail = 0;
        __syncthreads();                           // every stage above is done with the LDS regions reused below
        float* s_x = s_dp;                         // [n_chunks][128] partials (n_chunks <= 16)
        float* s_q = s_x + 16 * AD;                // [128] d_q
        float* s_ee = s_q + AD;                    // [4][128]
        bool ok = true;
        for (int i = tid; i < n_chunks * AD; i += 512) {
            const unsigned long long* g = fold.xbuf + ((size_t)b * n_chunks) * AD + i;
            bool got = false;
            for (int it = 0; it < ATTF_SPIN_MAX; ++it) {
                const unsigned long long v = __hip_atomic_load(g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if ((unsigned)(v >> 32) == fold.tag) { s_x[i] = __uint_as_float((unsigned)v); got = true; break; }
                __builtin_amdgcn_s_sleep(1);
            }
            ok = ok && got;
        }
        if (!ok) s_fail = 1;
        __syncthreads();
        if (s_fail) {
            if (tid == 0) __hip_atomic_store(fold.xbuf + (size_t)a.B * n_chunks * AD, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return;
        }
        if (tid < AD) {
            float qsum = 0.f;
            for (int c = 0; c < n_chunks; ++c) qsum += s_x[c * AD + tid];
            s_q[tid] = qsum;
            if (chunk == 0) a.d_q[(size_t)b * AD + tid] = qsum;
        }
        __syncthreads();
        const int H = ca.H, U = (H + n_chunks - 1) / n_chunks;
        const int u_lo = chunk * U, u_hi = u_lo + U < H ? u_lo + U : H;
        const int ul = tid & 127, kq = tid >> 7;
        for (int ub = u_lo; ub < u_hi; ub += 128) {
            const int u = ub + ul;
            const bool valid = u < u_hi;
            const int uc = valid ? u : u_lo;
            float w[32];
#pragma unroll
            for (int i = 0; i < 32; ++i) w[i] = ca.wq[(size_t)(kq * 32 + i) * H + uc];
            const size_t idx = (size_t)b * H + uc;
            float dh = 0.f, gi = 0.f, gf = 0.f, gg = 0.f, go = 0.f, cn = 0.f, cp = 0.f, dcc = 0.f;
            bool keep = true;
            if (kq == 0) {
                dh = sum3(ca.dh1, ca.s1, ca.dh2, ca.s2, ca.dh3, ca.s3, b, uc);
                const float* g4 = ca.gates + (size_t)b * 4 * H + uc;
                gi = g4[0]; gf = g4[H]; gg = g4[2 * H]; go = g4[3 * H];
                cn = ca.c_new[idx];
                cp = ca.c_prev ? ca.c_prev[idx] : 0.f;
                dcc = ca.dc_carry[idx];
                if (ca.drop_mask) keep = ca.drop_mask[idx] != 0;
            }
            float e0 = 0.f, e1 = 0.f;
#pragma unroll
            for (int i = 0; i < 32; i += 2) {
                e0 += s_q[kq * 32 + i] * w[i];
                e1 += s_q[kq * 32 + i + 1] * w[i + 1];
            }
            s_ee[kq * AD + ul] = e0 + e1;
            __syncthreads();
            if (kq == 0 && valid) {
                dh += (s_ee[ul] + s_ee[AD + ul]) + (s_ee[2 * AD + ul] + s_ee[3 * AD + ul]);
                if (ca.drop_mask) dh = keep ? dh * ca.drop_scale : 0.f;
                const float tc = tanhf(cn);
                const float dc = dcc + dh * go * (1.f - tc * tc);
                float* dg = ca.dgates + (size_t)b * 4 * H + u;
                dg[0] = dc * gg * gi * (1.f - gi);
                dg[H] = dc * cp * gf * (1.f - gf);
                dg[2 * H] = dc * gi * (1.f - gg * gg);
                dg[3 * H] = dh * tc * go * (1.f - go);
                ca.dc_carry[idx] = dc * gf;
            }
            __syncthreads();
        }
    }
}

static bool att_bwd_ok(const AttBwdArgs& a) {
    return !(a.enc_dim > 1024 || (a.enc_dim & 3) || a.att_dim > 128 || a.loc_f > 32 || a.loc_ks > 63 || !(a.loc_ks & 1) ||
             !a.dw_buf || !a.df_buf || !a.dq_part);
}
hipError_t t2s_launch_att_bwd_front(const AttBwdArgs& a, hipStream_t stream) {
    if (!att_bwd_ok(a)) return hipErrorInvalidValue;
    const dim3 grid((a.T + ATTB_CH - 1) / ATTB_CH, a.B);
    hipLaunchKernelGGL(att_bwd_dw_kernel, grid, dim3(256), 0, stream, a);
    // matrix-core form for the reference's shape (T2S_ATTB_VALU bit 0 / bit 1: the VALU energies / convolution kernel, for A/B runs)
    static const int valu = getenv("T2S_ATTB_VALU") ? atoi(getenv("T2S_ATTB_VALU")) : 0;
    if (!(valu & 1) && a.att_dim == 128 && a.loc_f == 32 && a.loc_ks <= 31) {
        hipLaunchKernelGGL(att_bwd_energy_mfma_kernel, grid, dim3(512), 0, stream, a);
    } else {
        hipLaunchKernelGGL(att_bwd_energy_kernel, grid, dim3(512), 0, stream, a);
    }
    return hipGetLastError();
}
hipError_t t2s_launch_att_bwd_conv(const AttBwdArgs& a, hipStream_t stream) {
    if (!att_bwd_ok(a)) return hipErrorInvalidValue;
    const dim3 grid((a.T + ATTB_CH - 1) / ATTB_CH, a.B);
    static const int valu = getenv("T2S_ATTB_VALU") ? atoi(getenv("T2S_ATTB_VALU")) : 0;
    if (!(valu & 2) && a.loc_f == 32 && a.loc_ks <= 31)
        hipLaunchKernelGGL(att_bwd_conv_mfma_kernel, grid, dim3(256), 0, stream, a);
    else
        hipLaunchKernelGGL(att_bwd_conv_kernel, grid, dim3(256), 0, stream, a);
    return hipGetLastError();
}
bool t2s_att_bwd_fused_ok(const AttBwdArgs& a) {
    static const int off = getenv("T2S_ATTB_FUSED") ? !atoi(getenv("T2S_ATTB_FUSED")) : 0;
    return !off && att_bwd_ok(a) && a.att_dim == 128 && a.loc_f == 32 && a.loc_ks <= 31 && a.ctx && a.dw_carry_out &&
           a.dwc_carry_out && a.dctx_out && !a.d_memory && a.dw_carry_out != a.dw_carry && a.dwc_carry_out != a.dwc_carry;
}
hipError_t t2s_launch_att_bwd_fused(const AttBwdArgs& a, hipStream_t stream, const AttBwdFoldArgs* fold) {
    if (!t2s_att_bwd_fused_ok(a)) return hipErrorInvalidValue;
    AttBwdFoldArgs f;
    memset(&f, 0, sizeof(f));
    if (fold) { f.sig_ptr = fold->sig_ptr; f.sig_val = fold->sig_val; }
    if (fold && fold->xbuf) {
        const LstmBwdArgs& c = fold->cell;
        if (a.T > 512 || a.att_dim != 128 || !c.wq || c.q_dim != 128 || c.B != a.B || c.H <= 0 || !c.gates || !c.c_new || !c.dc_carry ||
            !c.dgates || fold->tag == 0)
            return hipErrorInvalidValue;
        f = *fold;
    }
    const dim3 grid(8 * ((a.B + 7) / 8) * ((a.T + ATTB_CH - 1) / ATTB_CH));
    hipLaunchKernelGGL(att_bwd_fused_kernel, grid, dim3(512), 0, stream, a, f);
    return hipGetLastError();
}
hipError_t t2s_launch_att_bwd(const AttBwdArgs& a, hipStream_t stream) {
    hipError_t e = t2s_launch_att_bwd_front(a, stream);
    return e != hipSuccess ? e : t2s_launch_att_bwd_conv(a, stream);
}

// ------------------------------------------------------------------------------------------------
// Training-mode BatchNorm backward fused with the backward of activation + dropout.  With y = act(bn(x)) * mask * s:
//   dy' = d_out * mask * s * act'(.) ;  sums: S1 = sum dy', S2 = sum dy' xhat  over (B, T) per channel
//   dx = gamma inv_std / N (N dy' - S1 - xhat S2) ;  dgamma = S2 ; dbeta = S1
// d_out comes as f32 [B][C][T] or as planes (hi, lo); dx is written as planes (the dgrad / wgrad GEMM operand).
typedef __attribute__((ext_vector_type(8))) unsigned short u16x8;
// dy' of one element given the raw d_out value d (activation + dropout backward)
static __device__ __forceinline__ float bn_dyp(const BnBwdArgs& a, size_t i, int c, float xhat, float d) {
    if (a.mask) d = a.mask[i] ? d * a.mask_scale : 0.f;
    if (a.act != ACT_NONE) {
        const float ybn = xhat * a.gamma[c] + a.beta[c];
        if (a.act == ACT_RELU) d = ybn > 0.f ? d : 0.f;
        else { const float th = tanhf(ybn); d *= 1.f - th * th; }
    }
    return d;
}
// Both kernels work on tiles of 64 time steps x 32 channels (one channel chunk of the planes) and 256 threads.  The planes are
// [t][32 channels] rows of 64 bytes: read / written 16 bytes per thread (thread = (t, 8 channels)) and turned through LDS, while the
// f32 operands ([B][C][T]) are accessed time-major (thread = (channel mod 4, t)).  The first version read and wrote the planes
// 2 bytes per thread at a 64-byte stride: 0.9 TB/s (104 + 113 us per layer at B = 32, 512 channels, 800 frames).
static __device__ __forceinline__ void bn_stage_dout(const BnBwdArgs& a, int b, int chunk, int t0, float (*s_d)[33]) {
    const int tid = threadIdx.x, tl = tid >> 2, q = tid & 3, t = t0 + tl;
    if (!a.dout_f32) {
        const int nch = (a.C + 31) / 32;
        u16x8 h = {0, 0, 0, 0, 0, 0, 0, 0}, l = h;
        if (t < a.T) {
            const size_t pi = (((size_t)b * nch + chunk) * a.Lp + a.halo + t) * 32 + q * 8;
            h = *(const u16x8*)(a.dout_hi + pi);
            l = *(const u16x8*)(a.dout_lo + pi);
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) s_d[tl][q * 8 + e] = join_bf16(h[e], l[e]);
    }
}
// partial[b][c][2] (double): S1, S2 of one batch element; grid (channel chunks, B)
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const BnBwdArgs a, double* partial) {
    __shared__ float s_d[64][33];
    const int chunk = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    const int ci = tid >> 6, tl = tid & 63;
    double s1[8], s2[8];
    float mu[8], istd[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int c = chunk * 32 + ci + 4 * k;
        s1[k] = 0.0; s2[k] = 0.0;
        mu[k] = c < a.C ? a.mean[c] : 0.f;
        istd[k] = c < a.C ? 1.0f / sqrtf(a.var[c] + a.eps) : 0.f;
    }
    for (int t0 = 0; t0 < a.T; t0 += 64) {
        const int t = t0 + tl;
        // the f32 operands of this tile are requested before the plane tile is staged: one memory round trip per tile, not two
        float xv[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int c = chunk * 32 + ci + 4 * k;
            xv[k] = (c < a.C && t < a.T) ? a.x[((size_t)b * a.C + c) * a.T + t] : 0.f;
        }
        bn_stage_dout(a, b, chunk, t0, s_d);
        __syncthreads();
        if (t < a.T) {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int cl = ci + 4 * k, c = chunk * 32 + cl;
                if (c < a.C) {
                    const size_t i = ((size_t)b * a.C + c) * a.T + t;
                    const float xhat = (xv[k] - mu[k]) * istd[k];
                    const float d = bn_dyp(a, i, c, xhat, a.dout_f32 ? a.dout_f32[i] : s_d[tl][cl]);
                    s1[k] += d;
                    s2[k] += d * xhat;
                }
            }
        }
        __syncthreads();
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) { s1[k] += __shfl_xor(s1[k], off, 64); s2[k] += __shfl_xor(s2[k], off, 64); }
        const int c = chunk * 32 + ci + 4 * k;
        if (tl == 0 && c < a.C) {
            partial[((size_t)b * a.C + c) * 2] = s1[k];
            partial[((size_t)b * a.C + c) * 2 + 1] = s2[k];
        }
    }
}
__global__ void bn_bwd_reduce_final_kernel(const BnBwdArgs a, const double* partial) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= a.C) return;
    double s1 = 0.0, s2 = 0.0;
    for (int b = 0; b < a.B; ++b) { s1 += partial[((size_t)b * a.C + c) * 2]; s2 += partial[((size_t)b * a.C + c) * 2 + 1]; }
    a.dbeta[c] = (float)s1;
    a.dgamma[c] = (float)s2;
}
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const BnBwdArgs a) {
    __shared__ float s_d[64][33];
    const int t0 = blockIdx.x * 64, chunk = blockIdx.y, b = blockIdx.z, tid = threadIdx.x;
    const int ci = tid >> 6, tl = tid & 63, t = t0 + tl;
    const int nch = (a.C + 31) / 32;
    const float n = (float)a.B * a.T;
    bn_stage_dout(a, b, chunk, t0, s_d);
    __syncthreads();
    float dx[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int cl = ci + 4 * k, c = chunk * 32 + cl;
        dx[k] = 0.f;
        if (c < a.C && t < a.T) {
            const size_t i = ((size_t)b * a.C + c) * a.T + t;
            const float istd = 1.0f / sqrtf(a.var[c] + a.eps);
            const float xhat = (a.x[i] - a.mean[c]) * istd;
            const float d = bn_dyp(a, i, c, xhat, a.dout_f32 ? a.dout_f32[i] : s_d[tl][cl]);
            dx[k] = a.gamma[c] * istd / n * (n * d - a.dbeta[c] - xhat * a.dgamma[c]);
        }
    }
    __syncthreads();                                 // every read of the staged d_out is done: the tile now carries dx
#pragma unroll
    for (int k = 0; k < 8; ++k) s_d[tl][ci + 4 * k] = dx[k];
    __syncthreads();
    {
        const int tw = tid >> 2, q = tid & 3;
        if (t0 + tw < a.T) {
            u16x8 h, l;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                u16 hh, ll;
                split_bf16(s_d[tw][q * 8 + e], hh, ll);
                h[e] = hh;
                l[e] = ll;
            }
            const size_t idx = (((size_t)b * nch + chunk) * a.Lp + a.halo + t0 + tw) * 32 + q * 8;
            *(u16x8*)(a.dx_hi + idx) = h;
            *(u16x8*)(a.dx_lo + idx) = l;
        }
    }
}
hipError_t t2s_launch_bn_bwd(const BnBwdArgs& a, double* partial, hipStream_t stream) {
    hipLaunchKernelGGL(bn_bwd_reduce_kernel, dim3((a.C + 31) / 32, a.B), dim3(256), 0, stream, a, partial);
    hipLaunchKernelGGL(bn_bwd_reduce_final_kernel, dim3((a.C + 255) / 256), dim3(256), 0, stream, a, (const double*)partial);
    hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3((a.T + 63) / 64, (a.C + 31) / 32, a.B), dim3(256), 0, stream, a);
    return hipGetLastError();
}

// planes (hi, lo) -> f32 [B][C][L]
__global__ void planes_to_f32_kernel(const u16* X_hi, const u16* X_lo, int C, int L, int Lp, int halo, float* out, int accumulate) {
    const int t = blockIdx.x * 64 + (threadIdx.x & 63);
    const int b = blockIdx.z;
    const int nch = (C + 31) / 32;
    if (t >= L) return;
    for (int c = blockIdx.y * 32 + (threadIdx.x >> 6); c < blockIdx.y * 32 + 32 && c < C; c += 4) {
        const size_t idx = (((size_t)b * nch + (c >> 5)) * Lp + halo + t) * 32 + (c & 31);
        const float v = join_bf16(X_hi[idx], X_lo[idx]);
        float* o = out + ((size_t)b * C + c) * L + t;
        *o = accumulate ? *o + v : v;
    }
}
hipError_t t2s_launch_planes_to_f32(const u16* X_hi, const u16* X_lo, int B, int C, int L, int Lp, int halo, float* out,
                                    int accumulate, hipStream_t stream) {
    hipLaunchKernelGGL(planes_to_f32_kernel, dim3((L + 63) / 64, (C + 31) / 32, B), dim3(256), 0, stream, X_hi, X_lo, C, L,
                       Lp, halo, out, accumulate);
    return hipGetLastError();
}

// out[j] = sum_i in[i][j]   (per-batch-element partial parameter gradients -> the gradient)
__global__ void sum_axis0_kernel(const float* in, int n0, int n, float* out) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    float s = 0.f;
    for (int i = 0; i < n0; ++i) s += in[(size_t)i * n + j];
    out[j] = s;
}
hipError_t t2s_launch_sum_axis0(const float* in, int n0, int n, float* out, hipStream_t stream) {
    hipLaunchKernelGGL(sum_axis0_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, in, n0, n, out);
    return hipGetLastError();
}
// out = a + b (+ c)
__global__ void add3_kernel(const float* a, const float* b, const float* c, size_t n, float* out) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = a[i] + (b ? b[i] : 0.f) + (c ? c[i] : 0.f);
}
// out[i] = (in ? in[i] : 1) * scalar[0] * mul: the hand-off of an upstream gradient that lives in device memory (a loss
// Function's backward: d loss / d z = saved gradient x upstream scalar) without a host read or an eager operator
__global__ void scale_by_scalar_kernel(const float* in, size_t n, const float* scalar, float mul, float* out) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = (in ? in[i] : 1.f) * scalar[0] * mul;
}
hipError_t t2s_launch_scale_by_scalar(const float* in, size_t n, const float* scalar, float mul, float* out, hipStream_t stream) {
    hipLaunchKernelGGL(scale_by_scalar_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, in, n, scalar, mul, out);
    return hipGetLastError();
}
hipError_t t2s_launch_add3(const float* a, const float* b, const float* c, size_t n, float* out, hipStream_t stream) {
    hipLaunchKernelGGL(add3_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, a, b, c, n, out);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// Encoder BiLSTM BPTT (reference tacotron.py:199-207), one 1024-thread workgroup per (batch element, direction),
// steps walked in the reverse of the forward order; W_hh ([4H][H], natural layout: coalesced over k) is streamed
// once per step, the recurrent gradient lives in LDS.
__global__ __launch_bounds__(1024) void lstm_seq_bwd_kernel(const float* __restrict__ d_out, const float* __restrict__ out,
                                                            const float* __restrict__ gates, const float* __restrict__ csave,
                                                            const float* __restrict__ whh_f, const float* __restrict__ whh_r,
                                                            const int* __restrict__ lengths, float* dgx, float* hprev, int T,
                                                            int H, int T_out) {
    __shared__ __attribute__((aligned(16))) float s_dg[1024];
    __shared__ float s_dh[256];
    __shared__ float s_part[4][256];
    const int tid = threadIdx.x;
    const int b = blockIdx.x, dir = blockIdx.y;
    const float* W = dir ? whh_r : whh_f;
    const int len = lengths ? lengths[b] : T;
    float dc = 0.f;
    if (tid < H) s_dh[tid] = 0.f;
    __syncthreads();
    for (int s = 0; s < len; ++s) {
        const int t = dir ? s : len - 1 - s;                 // reverse of the forward order
        const int tp = dir ? t + 1 : t - 1;                  // the step the forward pass visited just before t
        const bool has_prev = dir ? (t < len - 1) : (t > 0);
        if (tid < H) {
            const int u = tid;
            const float dh = d_out[((size_t)b * T_out + t) * 2 * H + dir * H + u] + s_dh[u];
            const size_t gb = (((size_t)b * T + t) * 2 + dir) * 4 * H + u;
            const float gi = gates[gb], gf = gates[gb + H], gg = gates[gb + 2 * H], go = gates[gb + 3 * H];
            const float cn = csave[(((size_t)b * T + t) * 2 + dir) * H + u];
            const float cp = has_prev ? csave[(((size_t)b * T + tp) * 2 + dir) * H + u] : 0.f;
            const float tc = tanhf(cn);
            const float dct = dc + dh * go * (1.f - tc * tc);
            const float d0 = dct * gg * gi * (1.f - gi), d1 = dct * cp * gf * (1.f - gf), d2 = dct * gi * (1.f - gg * gg),
                        d3 = dh * tc * go * (1.f - go);
            dc = dct * gf;
            s_dg[u] = d0; s_dg[H + u] = d1; s_dg[2 * H + u] = d2; s_dg[3 * H + u] = d3;
            float* o = dgx + ((size_t)b * T + t) * 8 * H + dir * 4 * H + u;
            o[0] = d0; o[H] = d1; o[2 * H] = d2; o[3 * H] = d3;
            hprev[((size_t)b * T + t) * 2 * H + dir * H + u] = has_prev ? out[((size_t)b * T_out + tp) * 2 * H + dir * H + u] : 0.f;
        }
        __syncthreads();
        {
            const int jq = tid >> 8, k = tid & 255;
            float acc = 0.f;
            if (k < H) {
#pragma unroll 4
                for (int j = jq * H; j < (jq + 1) * H; j += 4) {       // one 16-byte broadcast LDS read per four rows
                    const f32x4 d4 = *(const f32x4*)&s_dg[j];
                    acc += (W[(size_t)j * H + k] * d4[0] + W[(size_t)(j + 1) * H + k] * d4[1]) +
                           (W[(size_t)(j + 2) * H + k] * d4[2] + W[(size_t)(j + 3) * H + k] * d4[3]);
                }
            }
            s_part[jq][k] = acc;
        }
        __syncthreads();
        if (tid < H) s_dh[tid] = (s_part[0][tid] + s_part[1][tid]) + (s_part[2][tid] + s_part[3][tid]);
        __syncthreads();
    }
}
hipError_t t2s_launch_lstm_seq_bwd(const float* d_out, const float* out, const float* gates, const float* csave,
                                   const float* whh_f, const float* whh_r, const int* lengths, float* dgx, float* hprev,
                                   int B, int T, int H, int T_out, hipStream_t stream) {
    if (H != 256) return hipErrorInvalidValue;
    hipLaunchKernelGGL(lstm_seq_bwd_kernel, dim3(B, 2), dim3(1024), 0, stream, d_out, out, gates, csave, whh_f, whh_r, lengths,
                       dgx, hprev, T, H, T_out);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// The same BPTT with W_hh RESIDENT (the forward's lstm_seq_split_kernel, csrc/tacotron_ops.hip, has the scheme): four workgroups per
// (batch element, direction); workgroup q owns the 64 outputs d_h[64 q .. 64 q + 63] and keeps W_hh[:, those 64 columns] (256 KB)
// in registers - thread (jq, kk) the 64 rows 64 jq .. of column 64 q + kk.  Per step every workgroup redoes the pointwise part for
// all 256 units (a few hundred flops; its 8 loads per unit are requested one step ahead), multiplies its quarter, and exchanges 64
// values with the other three through tagged 8-byte granules.  The one-workgroup kernel above streams 1 MB per step through one CU
// (25 us per step at B = 32: 6.5 ms of a train step, behind which the encoder's whole backward waits).
#define LSEQB_SPIN_MAX (1 << 22)
__global__ __launch_bounds__(1024) void lstm_seq_bwd_split_kernel(const float* __restrict__ d_out, const float* __restrict__ out,
                                                                  const float* __restrict__ gates, const float* __restrict__ csave,
                                                                  const float* __restrict__ whh_f, const float* __restrict__ whh_r,
                                                                  const int* __restrict__ lengths, float* dgx, float* hprev, int B,
                                                                  int T, int T_out, unsigned long long* xbuf, unsigned epoch) {
    constexpr int H = 256;
    __shared__ __attribute__((aligned(16))) float s_dg[4 * H];
    __shared__ float s_dh[H];
    __shared__ float s_part[16][64];
    __shared__ int s_fail;
    const int tid = threadIdx.x;
    const int id = blockIdx.x, group = (id >> 5) * 8 + (id & 7), q = (id >> 3) & 3;
    if (group >= 2 * B) return;                              // (whole workgroups)
    const int b = group >> 1, dir = group & 1;
    const int kk = tid & 63, jq = tid >> 6;
    const float* W = dir ? whh_r : whh_f;                    // [4H][H]
    const int len = lengths ? lengths[b] : T;
    float wr[64];
#pragma unroll
    for (int i = 0; i < 64; ++i) wr[i] = W[(size_t)(64 * jq + i) * H + 64 * q + kk];
    if (tid < H) s_dh[tid] = 0.f;
    if (tid == 0) s_fail = 0;
    unsigned long long* xg = xbuf + (size_t)group * 2 * H;
    const unsigned tag0 = epoch << 12;
    const int pu = tid >= 64 && tid < 256 ? ((tid - 64) < 64 * q ? (tid - 64) : tid) : 0;
    const bool mine = tid < H && (tid >> 6) == q;            // the units whose dgx / hprev rows this workgroup writes
    // operands of the pointwise part of step s, requested one step ahead (thread u < 256)
    float p_do = 0.f, p_g[4] = {0.f, 0.f, 0.f, 0.f}, p_cn = 0.f, p_cp = 0.f, p_hp = 0.f;
    auto fetch = [&](int s) {
        const int t = dir ? s : len - 1 - s;
        const int tp = dir ? t + 1 : t - 1;
        const bool has_prev = dir ? (t < len - 1) : (t > 0);
        const int u = tid;
        p_do = d_out[((size_t)b * T_out + t) * 2 * H + dir * H + u];
        const size_t gb = (((size_t)b * T + t) * 2 + dir) * 4 * H + u;
#pragma unroll
        for (int gi = 0; gi < 4; ++gi) p_g[gi] = gates[gb + (size_t)gi * H];
        p_cn = csave[(((size_t)b * T + t) * 2 + dir) * H + u];
        const int tq = has_prev ? tp : t;                    // (clamped address; dropped below)
        p_cp = csave[(((size_t)b * T + tq) * 2 + dir) * H + u];
        p_hp = out[((size_t)b * T_out + tq) * 2 * H + dir * H + u];
    };
    if (tid < H && len > 0) fetch(0);
    float dc = 0.f;
    __syncthreads();
    for (int s = 0; s < len; ++s) {
        const int t = dir ? s : len - 1 - s;                 // reverse of the forward order
        const bool has_prev = dir ? (t < len - 1) : (t > 0);
        if (tid < H) {
            const int u = tid;
            const float dh = p_do + s_dh[u];
            const float gi = p_g[0], gf = p_g[1], gg = p_g[2], go = p_g[3];
            const float cn = p_cn, cp = has_prev ? p_cp : 0.f;
            const float tc = tanhf(cn);
            const float dct = dc + dh * go * (1.f - tc * tc);
            const float d0 = dct * gg * gi * (1.f - gi), d1 = dct * cp * gf * (1.f - gf), d2 = dct * gi * (1.f - gg * gg),
                        d3 = dh * tc * go * (1.f - go);
            dc = dct * gf;
            s_dg[u] = d0; s_dg[H + u] = d1; s_dg[2 * H + u] = d2; s_dg[3 * H + u] = d3;
            if (mine) {
                float* o = dgx + ((size_t)b * T + t) * 8 * H + dir * 4 * H + u;
                o[0] = d0; o[H] = d1; o[2 * H] = d2; o[3 * H] = d3;
                hprev[((size_t)b * T + t) * 2 * H + dir * H + u] = has_prev ? p_hp : 0.f;
            }
            if (s + 1 < len) fetch(s + 1);                   // in flight across the product and the hand-off
        }
        __syncthreads();
        {
            float acc = 0.f;
#pragma unroll
            for (int i = 0; i < 64; i += 4) {
                const f32x4 d4 = *(const f32x4*)&s_dg[64 * jq + i];
                acc += (wr[i] * d4[0] + wr[i + 1] * d4[1]) + (wr[i + 2] * d4[2] + wr[i + 3] * d4[3]);
            }
            s_part[jq][kk] = acc;
        }
        __syncthreads();
        if (tid < 64) {
            float v = 0.f;
#pragma unroll
            for (int w = 0; w < 16; w += 4) v += (s_part[w][tid] + s_part[w + 1][tid]) + (s_part[w + 2][tid] + s_part[w + 3][tid]);
            const unsigned long long g = ((unsigned long long)(tag0 + (unsigned)s + 1u) << 32) | (unsigned long long)__float_as_uint(v);
            __hip_atomic_store(xg + (size_t)(s & 1) * H + 64 * q + tid, g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            s_dh[64 * q + tid] = v;
        } else if (tid < 256) {
            float v = 0.f;
            bool ok = true;
            if (s + 1 < len) {
                ok = false;
                const unsigned want = tag0 + (unsigned)s + 1u;
                const unsigned long long* slot = xg + (size_t)(s & 1) * H + pu;
                for (int it = 0; it < LSEQB_SPIN_MAX; ++it) {
                    const unsigned long long g = __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if ((unsigned)(g >> 32) == want) { v = __uint_as_float((unsigned)g); ok = true; break; }
                    __builtin_amdgcn_s_sleep(1);
                }
            }
            if (!ok) s_fail = 1;
            s_dh[pu] = v;
        }
        __syncthreads();
        if (s_fail) {
            if (tid == 0) __hip_atomic_store(xbuf + (size_t)2 * B * 2 * H, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            break;
        }
    }
}
hipError_t t2s_launch_lstm_seq_bwd_split(const float* d_out, const float* out, const float* gates, const float* csave,
                                         const float* whh_f, const float* whh_r, const int* lengths, float* dgx, float* hprev,
                                         int B, int T, int T_out, unsigned long long* xbuf, unsigned epoch, hipStream_t stream) {
    if (T >= 4095) return hipErrorInvalidValue;
    hipLaunchKernelGGL(lstm_seq_bwd_split_kernel, dim3(((2 * B + 7) / 8) * 32), dim3(1024), 0, stream, d_out, out, gates, csave,
                       whh_f, whh_r, lengths, dgx, hprev, B, T, T_out, xbuf, epoch & 0xFFFFFu);
    return hipGetLastError();
}

// f32 channel-last rows x[b][t][c] -> planes
__global__ void rows_to_planes_kernel(const float* x, int T, int C, int Lp, int halo, u16* X_hi, u16* X_lo) {
    const int t = blockIdx.x, b = blockIdx.y;
    const int nch = (C + 31) / 32;
    for (int c = threadIdx.x; c < nch * 32; c += blockDim.x) {
        const float v = c < C ? x[((size_t)b * T + t) * C + c] : 0.f;
        u16 h, l;
        split_bf16(v, h, l);
        const size_t idx = (((size_t)b * nch + (c >> 5)) * Lp + halo + t) * 32 + (c & 31);
        X_hi[idx] = h;
        X_lo[idx] = l;
    }
}
hipError_t t2s_launch_rows_to_planes(const float* x, int B, int T, int C, int Lp, int halo, u16* X_hi, u16* X_lo,
                                     hipStream_t stream) {
    hipLaunchKernelGGL(rows_to_planes_kernel, dim3(T, B), dim3(256), 0, stream, x, T, C, Lp, halo, X_hi, X_lo);
    return hipGetLastError();
}

// d_emb[v][e] = sum over (b, t) with ids[b][t] == v of d_x[b][e][t]   (d_x as planes).  One workgroup per (symbol, 32-channel
// chunk): deterministic, no atomics (the vocabulary is ~80 symbols).  The ids pass through LDS 2048 at a time; thread
// (position group pg, channel) adds the matches among positions pg, pg + 8, .. in order, the eight groups are summed in a fixed
// order.  (One workgroup per symbol scanning all B x T ids per channel pair took 1.2 ms at B = 32, T = 256.)
#define EMBG_CHUNK 2048
__global__ __launch_bounds__(256) void embedding_grad_kernel(const long* __restrict__ ids, const u16* __restrict__ D_hi,
                                                             const u16* __restrict__ D_lo, int B, int T, int E, int Lp,
                                                             int halo, float* d_emb) {
    __shared__ int s_ids[EMBG_CHUNK];
    __shared__ float s_acc[8][32];
    const int v = blockIdx.x, cc = blockIdx.y;
    const int nch = (E + 31) / 32, n = B * T;
    const int cl = threadIdx.x & 31, pg = threadIdx.x >> 5;
    float acc = 0.f;
    for (int base = 0; base < n; base += EMBG_CHUNK) {
        __syncthreads();
        for (int i = threadIdx.x; i < EMBG_CHUNK; i += 256) s_ids[i] = base + i < n ? (int)ids[base + i] : -1;
        __syncthreads();
        for (int i = pg; i < EMBG_CHUNK; i += 8)
            if (s_ids[i] == v) {
                const int p = base + i, bb = p / T, t = p - bb * T;
                const size_t idx = (((size_t)bb * nch + cc) * Lp + halo + t) * 32 + cl;
                acc += join_bf16(D_hi[idx], D_lo[idx]);
            }
    }
    s_acc[pg][cl] = acc;
    __syncthreads();
    if (pg == 0 && cc * 32 + cl < E) {
        float sum = s_acc[0][cl];
#pragma unroll
        for (int g = 1; g < 8; ++g) sum += s_acc[g][cl];
        d_emb[(size_t)v * E + cc * 32 + cl] = sum;
    }
}
hipError_t t2s_launch_embedding_grad(const long* ids, const u16* D_hi, const u16* D_lo, int B, int T, int E, int V, int Lp,
                                     int halo, float* d_emb, hipStream_t stream) {
    hipLaunchKernelGGL(embedding_grad_kernel, dim3(V, (E + 31) / 32), dim3(256), 0, stream, ids, D_hi, D_lo, B, T, E, Lp, halo, d_emb);
    return hipGetLastError();
}
