// Bandwidth-bound WaveGlow stages around the WN GEMMs (all VALU, no MFMA):
// weight-norm + packing, mel upsample + squeeze, audio squeeze, invertible 1x1
// conv, WN.start, WN.end + affine coupling.  Each kernel cites the reference
// lines it restates.  Layout of the (hi, lo) bf16 planes: t2s_common.h.
#include "t2s_common.h"
#include "t2s_kernels.h"

#include <string.h>

// ------------------------------------------------------------------------------------------------
// weight_norm (w = v * g / ||v||, reference glow.py:123,138,142,151) fused with the packing of the
// effective weight into the GEMM's A operand: split to (hi, lo) bf16, K reordered tap-major, rows
// permuted so that a wave owns the tanh row and the sigmoid row of the same channel.
// One workgroup per output row.
static __device__ __forceinline__ void pack_row(const float* __restrict__ vrow, const float* g, int g_is_scale, int o,
                                                 int Cin, int Kt, int perm, int C_gate, int row_off, int Mpad, int koff,
                                                 int Cin_pad, u16* A_hi, u16* A_lo, const float* bias_in,
                                                 const float* bias_in2, float* bias_out, int bias_accumulate,
                                                 float* scale_out = nullptr) {
    __shared__ float red[4];
    const int tid = threadIdx.x;
    const int n = Cin * Kt;
    float scale = 1.0f;
    if (g && g_is_scale) {
        scale = g[o];
    } else if (g) {
        float ss = 0.f;
        for (int i = tid; i < n; i += 256) {
            const float x = vrow[i];
            ss += x * x;
        }
        ss = wave_sum(ss);
        if ((tid & 63) == 0) red[tid >> 6] = ss;
        __syncthreads();
        ss = red[0] + red[1] + red[2] + red[3];
        scale = g[o] / sqrtf(ss);
    }
    if (scale_out && tid == 0) scale_out[o] = scale;
    int p;
    if (perm == PERM_GATE) {
        const int gate = o >= C_gate;
        const int ch = gate ? o - C_gate : o;
        p = (ch >> 7) * 256 + ((ch >> 6) & 1) * 128 + (((ch >> 4) & 3) * 2 + gate) * 16 + (ch & 15);
    } else {
        p = o + row_off;
    }
    // two consecutive channels per thread -> one 4-byte store per plane
    const int half = (Cin + 1) >> 1;
    for (int i = tid; i < half * Kt; i += 256) {
        const int tap = i / half;
        const int c = (i - tap * half) * 2;
        const float w0 = vrow[c * Kt + tap] * scale;
        const float w1 = (c + 1 < Cin) ? vrow[(c + 1) * Kt + tap] * scale : 0.f;
        const int k = koff + tap * Cin_pad + c;
        const size_t idx = ((size_t)(k >> 5) * Mpad + p) * 32 + (k & 31);
        u16 h0, l0, h1, l1;
        split_bf16(w0, h0, l0);
        split_bf16(w1, h1, l1);
        *(uint32_t*)(A_hi + idx) = h0 | ((uint32_t)h1 << 16);
        *(uint32_t*)(A_lo + idx) = l0 | ((uint32_t)l1 << 16);
    }
    if (tid == 0 && bias_out) {
        float bi = bias_in ? bias_in[o] : 0.f;
        if (bias_in2) bi += bias_in2[o];
        bias_out[p] = bias_accumulate ? bias_out[p] + bi : bi;
    }
}

__global__ __launch_bounds__(256) void pack_kernel(const PackArgs a) {
    const int o = blockIdx.x;
    pack_row(a.v + (size_t)o * a.Cin * a.Kt, a.g, a.g_is_scale, o, a.Cin, a.Kt, a.perm, a.C_gate, a.row_off, a.Mpad,
             a.koff, a.Cin_pad, a.A_hi, a.A_lo, a.bias_in, nullptr, a.bias_out, a.bias_accumulate);
}

// One launch for the whole model.  A workgroup packs 16 consecutive output rows of one job (block -> (job, group) by binary
// search over the jobs' row_start prefix, which counts 16-row groups): the per-row kernel above is latency-bound (three
// dependent round trips for 6 KB of work), here the same chain moves 16x the bytes and the 16 packed rows of a K-chunk are
// one contiguous 1 KB store per plane (rows p..p+15 are adjacent for both permutations).
// Thread = (row r = tid / 16, channel pair cp = tid % 16).  Single pass when the slice fits in registers (Kt in {1, 3},
// <= 20 channel chunks - every WaveGlow convolution): the thread reads the 2*Kt contiguous floats of its channel pair in
// every 32-channel chunk (a row's 16 threads cover 32*Kt contiguous floats), the row's sum of squares is a 16-lane
// shuffle reduction, and the scaled values go out from registers.  Otherwise two passes over the row.
static __device__ __forceinline__ int pack_dst_row(const PackJob& j, int o) {
    if (j.perm == PERM_GATE) {
        const int gate = o >= (int)j.C_gate;
        const int ch = gate ? o - (int)j.C_gate : o;
        return (ch >> 7) * 256 + ((ch >> 6) & 1) * 128 + (((ch >> 4) & 3) * 2 + gate) * 16 + (ch & 15);
    }
    if (j.perm == PERM_PAIR8 && o < (int)j.C_gate) {
        // rows o < C_gate (the residual half), in groups of 32 channels: packed row m * 16 + q * 4 + e holds channel 8 q + 4 m + e, so
        // that the two 16-row MFMA tiles of a group give lane group q the 8 CONSECUTIVE channels 8 q .. 8 q + 7 of a plane row: one
        // 16-byte access per plane where the identity order needs two 8-byte ones (a quarter of a 128-byte line per request)
        const int w = o & 31;
        return (o & ~31) + ((w >> 2) & 1) * 16 + (w >> 3) * 4 + (w & 3) + (int)j.row_off;
    }
    return o + (int)j.row_off;
}
static __device__ __forceinline__ void pack_store2(const PackJob& j, int k, int p, float w0, float w1) {
    const size_t idx = ((size_t)(k >> 5) * (int)j.Mpad + p) * 32 + (k & 31);
    u16 h0, l0, h1, l1;
    split_bf16(w0, h0, l0);
    split_bf16(w1, h1, l1);
    *(uint32_t*)(j.A_hi + idx) = h0 | ((uint32_t)h1 << 16);
    *(uint32_t*)(j.A_lo + idx) = l0 | ((uint32_t)l1 << 16);
}
static __device__ __forceinline__ float sum16(float v) {      // over the 16 lanes that share a row
    v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64); v += __shfl_xor(v, 8, 64);
    return v;
}
#define PACK_MAX_CHUNKS 20
template <int KT>
static __device__ __forceinline__ void pack_rows_regs(const PackJob& j, int o0) {
    const int tid = threadIdx.x, r = tid >> 4, cp = tid & 15;
    const int O = (int)j.O, Cin = (int)j.Cin, Cin_pad = (int)j.Cin_pad, koff = (int)j.koff;
    const int o = o0 + r;
    const bool row_ok = o < O;
    const int n = Cin * KT;
    const int nchunk = (Cin + 31) >> 5;
    const float* vrow = j.v + (size_t)(row_ok ? o : 0) * n;
    float x[PACK_MAX_CHUNKS][2 * KT];
    float ss = 0.f;
#pragma unroll
    for (int cc = 0; cc < PACK_MAX_CHUNKS; ++cc) {
        const int c = cc * 32 + cp * 2;
#pragma unroll
        for (int e = 0; e < 2 * KT; ++e) {
            const int ce = c + e / KT;                       // element e of the pair: channel c or c + 1, tap e % KT
            x[cc][e] = (cc < nchunk && row_ok && ce < Cin) ? vrow[c * KT + e] : 0.f;
            ss += x[cc][e] * x[cc][e];
        }
    }
    float scale = 1.0f;
    if (j.g) {
        if (j.g_is_scale) scale = row_ok ? j.g[o] : 1.f;
        else {
            ss = sum16(ss);
            scale = row_ok ? j.g[o] / sqrtf(ss) : 1.f;
        }
    }
    if (!row_ok) return;
    const int p = pack_dst_row(j, o);
    if (cp == 0) {
        if (j.scale_out) j.scale_out[o] = scale;
        if (j.bias_out) {
            float bi = j.bias_in ? j.bias_in[o] : 0.f;
            if (j.bias_in2) bi += j.bias_in2[o];
            j.bias_out[p] = bi;
        }
    }
#pragma unroll
    for (int cc = 0; cc < PACK_MAX_CHUNKS; ++cc) {
        if (cc < nchunk) {
            const int c = cc * 32 + cp * 2;
#pragma unroll
            for (int tap = 0; tap < KT; ++tap)
                pack_store2(j, koff + tap * Cin_pad + c, p, x[cc][tap] * scale, x[cc][KT + tap] * scale);
        }
    }
}

static __device__ __forceinline__ void pack_rows_two_pass(const PackJob& j, int o0, float* s_scale) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int O = (int)j.O, Cin = (int)j.Cin, Kt = (int)j.Kt, Cin_pad = (int)j.Cin_pad, koff = (int)j.koff;
    const int n = Cin * Kt;
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
        const int r = wave * 4 + rr, o = o0 + r;
        float scale = 1.0f;
        if (o < O && j.g) {
            if (j.g_is_scale) {
                scale = j.g[o];
            } else {
                const float* vrow = j.v + (size_t)o * n;
                float ss = 0.f;
                for (int i = lane; i < n; i += 64) {
                    const float x = vrow[i];
                    ss += x * x;
                }
                ss = wave_sum(ss);
                scale = j.g[o] / sqrtf(ss);
            }
        }
        if (lane == 0) {
            s_scale[r] = scale;
            if (o < O) {
                if (j.scale_out) j.scale_out[o] = scale;
                if (j.bias_out) {
                    float bi = j.bias_in ? j.bias_in[o] : 0.f;
                    if (j.bias_in2) bi += j.bias_in2[o];
                    j.bias_out[pack_dst_row(j, o)] = bi;
                }
            }
        }
    }
    __syncthreads();
    const int r = tid >> 4, cp = tid & 15;
    const int o = o0 + r;
    if (o >= O) return;
    const int p = pack_dst_row(j, o);
    const float scale = s_scale[r];
    const float* vrow = j.v + (size_t)o * n;
    const int nchunk = (Cin + 31) >> 5;
    for (int cc = 0; cc < nchunk; ++cc) {
        const int c = cc * 32 + cp * 2;
        for (int tap = 0; tap < Kt; ++tap) {
            const float w0 = c < Cin ? vrow[c * Kt + tap] * scale : 0.f;
            const float w1 = c + 1 < Cin ? vrow[(c + 1) * Kt + tap] * scale : 0.f;
            pack_store2(j, koff + tap * Cin_pad + c, p, w0, w1);
        }
    }
}

__global__ __launch_bounds__(256) void pack_table_kernel(const PackJob* __restrict__ jobs, int n_jobs) {
    __shared__ float s_scale[16];
    const long blk = blockIdx.x;
    int lo = 0, hi = n_jobs - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (jobs[mid].row_start <= blk) lo = mid; else hi = mid - 1;
    }
    const PackJob j = jobs[lo];
    const int o0 = (int)(blk - j.row_start) * 16;
    const int nchunk = ((int)j.Cin + 31) >> 5;
    // Cin even keeps a thread's 2*Kt floats inside the row; the register path also needs whole channel pairs
    if (nchunk <= PACK_MAX_CHUNKS && !((int)j.Cin & 1) && j.Kt == 3) pack_rows_regs<3>(j, o0);
    else if (nchunk <= PACK_MAX_CHUNKS && !((int)j.Cin & 1) && j.Kt == 1) pack_rows_regs<1>(j, o0);
    else pack_rows_two_pass(j, o0, s_scale);
}
hipError_t t2s_launch_pack_table(const PackJob* jobs, int n_jobs, long total_groups, hipStream_t stream) {
    hipLaunchKernelGGL(pack_table_kernel, dim3((unsigned)total_groups), dim3(256), 0, stream, jobs, n_jobs);
    return hipGetLastError();
}

hipError_t t2s_launch_pack(const PackArgs& a, hipStream_t stream) {
    hipLaunchKernelGGL(pack_kernel, dim3(a.O), dim3(256), 0, stream, a);
    return hipGetLastError();
}

// Effective weight of a small weight-normed conv (WN.start, K = n_half): w[o][k] = v*g/||v||.
__global__ void weightnorm_small_kernel(const float* v, const float* g, int O, int K, float* w) {
    const int o = blockIdx.x * blockDim.x + threadIdx.x;
    if (o >= O) return;
    float ss = 0.f;
    for (int k = 0; k < K; ++k) ss += v[o * K + k] * v[o * K + k];
    const float s = g ? g[o] / sqrtf(ss) : 1.f;
    for (int k = 0; k < K; ++k) w[o * K + k] = v[o * K + k] * s;
}
hipError_t t2s_launch_weightnorm_small(const float* v, const float* g, int O, int K, float* w, hipStream_t stream) {
    hipLaunchKernelGGL(weightnorm_small_kernel, dim3((O + 255) / 256), dim3(256), 0, stream, v, g, O, K, w);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// ConvTranspose1d(n_mel, n_mel, ksize, stride) + trim + squeeze-by-n_group, written straight into
// the conditioning planes (reference glow.py:183-185,215-221; infer: glow.py:252-258).
//   spect[b][co*G + g][t] = bias[co] + sum_{ci,f} mel[b][ci][f] * W[ci][co][G*t + g - stride*f]
// One thread = 8 consecutive squeezed channels of one time step for up to BB batch elements, so each
// weight value fetched is reused BB times; the 8 results are one 16-B store per plane.
template <int BB>
__global__ __launch_bounds__(256) void upsample_squeeze_kernel(const float* __restrict__ mel, const float* __restrict__ W,
                                                               const float* __restrict__ bias, int B, int M, int F,
                                                               int ksize, int stride, int G, int L, int Lp, int halo,
                                                               u16* S_hi, u16* S_lo) {
    const int tid = threadIdx.x;
    const int q = tid & 3;
    const int t = blockIdx.x * 64 + (tid >> 2);
    const int chunk = blockIdx.y;
    const int b0 = blockIdx.z * BB;
    const int c8 = chunk * 32 + q * 8;          // first squeezed channel of this thread
    const int nchan = M * G;
    if (t >= L || c8 >= nchan) return;
    float acc[BB][8];
#pragma unroll
    for (int bb = 0; bb < BB; ++bb)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[bb][j] = 0.f;
    const int nf = ksize / stride;
    if (G == 8) {
        const int co = c8 >> 3;
        const int s0 = 8 * t;
        const int fhi = s0 / stride;
        const int kbase = s0 - fhi * stride;
        for (int ci = 0; ci < M; ++ci) {
            const float* wrow = W + ((size_t)ci * M + co) * ksize + kbase;
            for (int i = 0; i < nf; ++i) {
                const int f = fhi - i;
                if (f < 0 || f >= F) continue;
                const f32x4 w0 = *(const f32x4*)(wrow + i * stride);
                const f32x4 w1 = *(const f32x4*)(wrow + i * stride + 4);
#pragma unroll
                for (int bb = 0; bb < BB; ++bb) {
                    if (b0 + bb >= B) break;
                    const float m = mel[((size_t)(b0 + bb) * M + ci) * F + f];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        acc[bb][j] += m * w0[j];
                        acc[bb][4 + j] += m * w1[j];
                    }
                }
            }
        }
#pragma unroll
        for (int bb = 0; bb < BB; ++bb)
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[bb][j] += bias[co];
    } else {
        for (int j = 0; j < 8; ++j) {
            const int cidx = c8 + j;
            if (cidx >= nchan) break;
            const int co = cidx / G, g = cidx - co * G;
            const int s = G * t + g;
            const int fhi = s / stride;
            for (int ci = 0; ci < M; ++ci)
                for (int i = 0; i < nf; ++i) {
                    const int f = fhi - i;
                    if (f < 0 || f >= F) continue;
                    const float w = W[((size_t)ci * M + co) * ksize + (s - f * stride)];
                    for (int bb = 0; bb < BB; ++bb)
                        if (b0 + bb < B) acc[bb][j] += mel[((size_t)(b0 + bb) * M + ci) * F + f] * w;
                }
            for (int bb = 0; bb < BB; ++bb) acc[bb][j] += bias[co];
        }
    }
    const int nchunks = (nchan + 31) / 32;
#pragma unroll
    for (int bb = 0; bb < BB; ++bb) {
        if (b0 + bb >= B) break;
        const size_t row = ((size_t)(b0 + bb) * nchunks + chunk) * Lp + halo + t;
        u16 hi[8], lo[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float v = (c8 + j < nchan) ? acc[bb][j] : 0.f;
            split_bf16(v, hi[j], lo[j]);
        }
        uint4 ph, pl;
        ph.x = hi[0] | ((uint32_t)hi[1] << 16); ph.y = hi[2] | ((uint32_t)hi[3] << 16);
        ph.z = hi[4] | ((uint32_t)hi[5] << 16); ph.w = hi[6] | ((uint32_t)hi[7] << 16);
        pl.x = lo[0] | ((uint32_t)lo[1] << 16); pl.y = lo[2] | ((uint32_t)lo[3] << 16);
        pl.z = lo[4] | ((uint32_t)lo[5] << 16); pl.w = lo[6] | ((uint32_t)lo[7] << 16);
        *(uint4*)(S_hi + row * 32 + q * 8) = ph;
        *(uint4*)(S_lo + row * 32 + q * 8) = pl;
    }
}

// The same operator as a GEMM on the exact-f32 matrix cores (v_mfma_f32_16x16x4_f32), for the shape WaveGlow uses (n_group 8,
// ksize = 4 * stride, stride a multiple of 64, n_mel a multiple of 4):
//   rows    m = (co, p), p = position inside a hop        A[m][k] = W[ci][co][j * stride + p]
//   columns n = (b, f), f = hop (mel frame) index          B[k][n] = mel[b][ci][f - j]   (0 outside [0, F))
//   K       k = (ci, j), j = 0..3 the four overlapping hops
// 6.6 GFLOP at batch 8 x 16000: 438 us on the vector ALUs, 110 us here (the matrix-core floor is 53 us).  One wave owns 16 positions p of the four output
// channels that make up one 32-channel plane chunk (4 m-tiles) times NT column tiles of 16 frames; one K-step = one input channel
// (lane >> 4 = j).  Operands go global -> register (each A value is used by NT tiles, each B value by 4; W is 26 MB and every
// wave streams its slice once, mel is 160 KB and cache-resident), prefetched one K-step ahead.
// D tile: col = lane & 15 (frame), row = 4 * (lane >> 4) + e -> p = p0 + 4q + e: plane row t = f * (stride / 8) + (p >> 3),
// squeezed channel co * 8 + (p & 7): the lane's four values are four consecutive channels of one plane row, one 8-B store.
typedef float f32x4_up __attribute__((ext_vector_type(4)));
template <int NT, int KB>
__global__ __launch_bounds__(256) void upsample_mfma_kernel(const float* __restrict__ mel, const float* __restrict__ W,
                                                            const float* __restrict__ bias, int B, int M, int F, int ksize,
                                                            int stride, int L, int Lp, int halo, int tiles_per_b, int n_ctiles,
                                                            int n_pairs, int nz, u16* S_hi, u16* S_lo) {
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int r = lane & 15, q = lane >> 4;
    // Workgroups are dealt round-robin to the 8 XCDs: the nz column groups that share one slice of W (a (position block,
    // channel chunk) pair) sit on ONE XCD, so its L2 fetches the slice once (any-order placement had every XCD stream all 26 MB)
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int pair = (slot / nz) * 8 + xcd;
    if (pair >= n_pairs) return;
    const int npb = stride >> 6;                          // position blocks of 64 per hop
    const int p0 = ((pair % npb) * 4 + wave) * 16;
    const int chunk = pair / npb;                         // 4 output channels = 32 squeezed channels
    const int ct0 = (slot % nz) * NT;
    // A: W[(ci * M + co) * ksize + q * stride + p0 + r], co = chunk * 4 + o4; advancing ci adds M * ksize
    const float* wa = W + (size_t)(chunk * 4) * ksize + (size_t)q * stride + p0 + r;
    const size_t wa_step = (size_t)M * ksize;
    // B: mel[(b * M + ci) * F + f - q]; advancing ci adds F
    const float* mb[NT];
    bool mv[NT];
#pragma unroll
    for (int n = 0; n < NT; ++n) {
        const int ct = ct0 + n;
        const int b = ct / tiles_per_b;
        const int f = (ct - b * tiles_per_b) * 16 + r - q;
        mv[n] = ct < n_ctiles && b < B && f >= 0 && f < F;
        mb[n] = mel + ((size_t)(mv[n] ? b : 0) * M) * F + (mv[n] ? f : 0);
    }
    f32x4_up acc[4][NT];
#pragma unroll
    for (int o = 0; o < 4; ++o)
#pragma unroll
        for (int n = 0; n < NT; ++n) acc[o][n] = (f32x4_up){0.f, 0.f, 0.f, 0.f};
    // K runs in batches of KB input channels: the KB * (4 + NT) operand loads of batch i + 1 are in flight under the
    // KB * 4 * NT matrix instructions of batch i
    float a_cur[KB][4], b_cur[KB][NT], a_nxt[KB][4], b_nxt[KB][NT];
#pragma unroll
    for (int kk = 0; kk < KB; ++kk) {
#pragma unroll
        for (int o = 0; o < 4; ++o) a_cur[kk][o] = wa[(size_t)kk * wa_step + (size_t)o * ksize];
#pragma unroll
        for (int n = 0; n < NT; ++n) b_cur[kk][n] = mv[n] ? mb[n][(size_t)kk * F] : 0.f;
    }
    for (int ci = 0; ci < M; ci += KB) {
        const int cn = ci + KB < M ? ci + KB : ci;        // the last batch re-reads its own operands (no branch in the loop)
#pragma unroll
        for (int kk = 0; kk < KB; ++kk) {
#pragma unroll
            for (int o = 0; o < 4; ++o) a_nxt[kk][o] = wa[(size_t)(cn + kk) * wa_step + (size_t)o * ksize];
#pragma unroll
            for (int n = 0; n < NT; ++n) b_nxt[kk][n] = mv[n] ? mb[n][(size_t)(cn + kk) * F] : 0.f;
        }
#pragma unroll
        for (int kk = 0; kk < KB; ++kk)
#pragma unroll
            for (int o = 0; o < 4; ++o)
#pragma unroll
                for (int n = 0; n < NT; ++n)
                    acc[o][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_cur[kk][o], b_cur[kk][n], acc[o][n], 0, 0, 0);
#pragma unroll
        for (int kk = 0; kk < KB; ++kk) {
#pragma unroll
            for (int o = 0; o < 4; ++o) a_cur[kk][o] = a_nxt[kk][o];
#pragma unroll
            for (int n = 0; n < NT; ++n) b_cur[kk][n] = b_nxt[kk][n];
        }
    }
    const int nchunks = (M * 8 + 31) / 32;
    const int tpf = stride >> 3;                          // plane rows per hop
    const int pq = p0 + 4 * q;
#pragma unroll
    for (int n = 0; n < NT; ++n) {
        const int ct = ct0 + n;
        const int b = ct / tiles_per_b;
        const int f = (ct - b * tiles_per_b) * 16 + r;
        const int t = f * tpf + (pq >> 3);
        if (ct >= n_ctiles || b >= B || t >= L) continue;
        const size_t row = (((size_t)b * nchunks + chunk) * Lp + halo + t) * 32 + (pq & 7);
#pragma unroll
        for (int o = 0; o < 4; ++o) {
            const float bo = bias[chunk * 4 + o];
            u16 hi[4], lo[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) split_bf16(acc[o][n][e] + bo, hi[e], lo[e]);
            uint2 ph, pl;
            ph.x = hi[0] | ((uint32_t)hi[1] << 16); ph.y = hi[2] | ((uint32_t)hi[3] << 16);
            pl.x = lo[0] | ((uint32_t)lo[1] << 16); pl.y = lo[2] | ((uint32_t)lo[3] << 16);
            *(uint2*)(S_hi + row + o * 8) = ph;
            *(uint2*)(S_lo + row + o * 8) = pl;
        }
    }
}

hipError_t t2s_launch_upsample_squeeze(const float* mel, const float* W, const float* bias, int B, int n_mel,
                                       int frames, int ksize, int stride, int n_group, int L, int Lp, int halo,
                                       u16* S_hi, u16* S_lo, hipStream_t stream) {
    const int nchunks = (n_mel * n_group + 31) / 32;
    static const bool valu_only = getenv("T2S_UPSAMPLE_VALU") != nullptr;      // A/B switch
    if (!valu_only && n_group == 8 && ksize == 4 * stride && stride % 64 == 0 && n_mel % 4 == 0) {
        // frames that reach plane rows < L: f <= (8 * L - 1) / stride
        const int nf = (8 * L - 1) / stride + 1;
        const int tiles_per_b = (nf + 15) / 16;
        const int n_ctiles = tiles_per_b * B;
        constexpr int NT = 2;
        const int nz = (n_ctiles + NT - 1) / NT;
        const int n_pairs = (stride / 64) * (n_mel / 4);
        dim3 grid(8 * ((n_pairs + 7) / 8) * nz);
        // KB = 1: one K-step of operands in flight per wave, 5 waves per SIMD.  KB = 4 (four K-steps per round trip, 96 VGPRs)
        // measured slower on MI355X (145 us against 114 us), as did NT = 4 (132 us): the kernel wants waves, not depth.
        hipLaunchKernelGGL((upsample_mfma_kernel<NT, 1>), grid, dim3(256), 0, stream, mel, W, bias, B, n_mel, frames, ksize,
                           stride, L, Lp, halo, tiles_per_b, n_ctiles, n_pairs, nz, S_hi, S_lo);
        return hipGetLastError();
    }
    constexpr int BB = 8;
    dim3 grid((L + 63) / 64, nchunks, (B + BB - 1) / BB);
    hipLaunchKernelGGL(upsample_squeeze_kernel<BB>, grid, dim3(256), 0, stream, mel, W, bias, B, n_mel, frames, ksize,
                       stride, n_group, L, Lp, halo, S_hi, S_lo);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// Composed conditioning (vocoder inference).  The upsampled spectrogram at plane row t = P * f + phi is
//   spect[co * G + g][t] = bias[co] + sum_{ci, j} mel[ci][f - j] * W[ci][co][stride * j + G * phi + g],   j = 0 .. ksize / stride - 1,
// a linear map U_phi of the mel WINDOW of frame f (K2 = nlag * n_mel values, k = j * n_mel + ci).  So
//   cond_layers[i](spect)[:, t] = (W_cond,i U_phi) melwindow(f) + W_cond,i bias_expanded:
// (1) upbasis_planes_kernel lays U out as ordinary conditioning planes whose "time" axis is the column (phi, k) - plus one last
//     column holding the expanded bias - so that the existing 1x1 GEMM (t2s_conv_bias_act on the conditioning slice of the packed
//     gate weights) produces W_cond,i U for all phases at once, in packed row order;
// (2) compose_pack_kernel splits that f32 result into the A-operand planes A2[phi][k / 32][row][k % 32] and folds the last column
//     into the layer's bias;
// (3) melwin_planes_kernel builds the mel-window planes M[b][k / 32][f][k % 32] per utterance (160 KB of mel -> 0.6 MB).
__global__ __launch_bounds__(256) void upbasis_planes_kernel(const float* __restrict__ W, const float* __restrict__ bias, int M,
                                                             int ksize, int stride, int G, int P, int K2, int Lp, int halo,
                                                             u16* U_hi, u16* U_lo) {
    const int q = threadIdx.x & 3;
    const int col = blockIdx.x * 64 + (threadIdx.x >> 2);
    const int chunk = blockIdx.y;
    const int ncols = P * K2 + 1;
    if (col >= ncols) return;
    const int c8 = chunk * 32 + q * 8;
    u16 hi[8], lo[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int c = c8 + e;
        float v = 0.f;
        if (c < M * G) {
            const int co = c / G, g = c - co * G;
            if (col == ncols - 1) {
                v = bias[co];
            } else {
                const int phi = col / K2, k = col - phi * K2;
                const int j = k / M, ci = k - j * M;
                v = W[((size_t)ci * M + co) * ksize + (size_t)stride * j + G * phi + g];
            }
        }
        split_bf16(v, hi[e], lo[e]);
    }
    const size_t row = ((size_t)chunk * Lp + halo + col) * 32 + q * 8;
    uint4 ph, pl;
    ph.x = hi[0] | ((uint32_t)hi[1] << 16); ph.y = hi[2] | ((uint32_t)hi[3] << 16);
    ph.z = hi[4] | ((uint32_t)hi[5] << 16); ph.w = hi[6] | ((uint32_t)hi[7] << 16);
    pl.x = lo[0] | ((uint32_t)lo[1] << 16); pl.y = lo[2] | ((uint32_t)lo[3] << 16);
    pl.z = lo[4] | ((uint32_t)lo[5] << 16); pl.w = lo[6] | ((uint32_t)lo[7] << 16);
    *(uint4*)(U_hi + row) = ph;
    *(uint4*)(U_lo + row) = pl;
}
hipError_t t2s_launch_upbasis_planes(const float* W, const float* bias, int n_mel, int ksize, int stride, int n_group, int Lp,
                                     int halo, u16* U_hi, u16* U_lo, hipStream_t stream) {
    const int P = stride / n_group, K2 = (ksize / stride) * n_mel;
    const int ncols = P * K2 + 1;
    dim3 grid((ncols + 63) / 64, (n_mel * n_group + 31) / 32);
    hipLaunchKernelGGL(upbasis_planes_kernel, grid, dim3(256), 0, stream, W, bias, n_mel, ksize, stride, n_group, P, K2, Lp, halo,
                       U_hi, U_lo);
    return hipGetLastError();
}

// tmp [rows][ncols] f32 (row = packed gate row) -> A2[phi][kc][Mpad][32] (hi, lo), bias_out[row] = bias_in[row] + tmp[row][P * K2]
__global__ __launch_bounds__(256) void compose_pack_kernel(const float* __restrict__ tmp, const float* __restrict__ bias_in,
                                                           int rows, int Mpad, int P, int K2, int ld, u16* A2_hi, u16* A2_lo,
                                                           float* bias_out) {
    const int q = threadIdx.x & 3;
    const int m = blockIdx.x * 64 + (threadIdx.x >> 2);
    const int kc = blockIdx.y, phi = blockIdx.z;
    if (m >= Mpad) return;
    u16 hi[8], lo[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int k = kc * 32 + q * 8 + e;
        const float v = (m < rows && k < K2) ? tmp[(size_t)m * ld + (size_t)phi * K2 + k] : 0.f;
        split_bf16(v, hi[e], lo[e]);
    }
    const int mc = (K2 + 31) / 32;
    const size_t o = ((((size_t)phi * mc + kc) * Mpad) + m) * 32 + q * 8;
    uint4 ph, pl;
    ph.x = hi[0] | ((uint32_t)hi[1] << 16); ph.y = hi[2] | ((uint32_t)hi[3] << 16);
    ph.z = hi[4] | ((uint32_t)hi[5] << 16); ph.w = hi[6] | ((uint32_t)hi[7] << 16);
    pl.x = lo[0] | ((uint32_t)lo[1] << 16); pl.y = lo[2] | ((uint32_t)lo[3] << 16);
    pl.z = lo[4] | ((uint32_t)lo[5] << 16); pl.w = lo[6] | ((uint32_t)lo[7] << 16);
    *(uint4*)(A2_hi + o) = ph;
    *(uint4*)(A2_lo + o) = pl;
    if (kc == 0 && phi == 0 && q == 0) bias_out[m] = bias_in[m] + (m < rows ? tmp[(size_t)m * ld + (size_t)P * K2] : 0.f);
}
hipError_t t2s_launch_compose_pack(const float* tmp, const float* bias_in, int rows, int Mpad, int P, int K2, int ld, u16* A2_hi,
                                   u16* A2_lo, float* bias_out, hipStream_t stream) {
    dim3 grid((Mpad + 63) / 64, (K2 + 31) / 32, P);
    hipLaunchKernelGGL(compose_pack_kernel, grid, dim3(256), 0, stream, tmp, bias_in, rows, Mpad, P, K2, ld, A2_hi, A2_lo, bias_out);
    return hipGetLastError();
}

// mel [B][M][F] f32 -> mel-window planes S[b][k / 32][f][k % 32], k = j * M + ci  ->  mel[b][ci][f - j] (0 outside [0, F))
__global__ __launch_bounds__(256) void melwin_planes_kernel(const float* __restrict__ mel, int M, int F, int nlag, int Fp,
                                                            u16* S_hi, u16* S_lo) {
    const int q = threadIdx.x & 3;
    const int f = blockIdx.x * 64 + (threadIdx.x >> 2);
    const int chunk = blockIdx.y, b = blockIdx.z;
    if (f >= Fp) return;
    const int K2 = nlag * M, mc = (K2 + 31) / 32;
    u16 hi[8], lo[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int k = chunk * 32 + q * 8 + e;
        float v = 0.f;
        if (k < K2) {
            const int j = k / M, ci = k - j * M;
            const int ff = f - j;
            if (ff >= 0 && ff < F) v = mel[((size_t)b * M + ci) * F + ff];
        }
        split_bf16(v, hi[e], lo[e]);
    }
    const size_t o = (((size_t)b * mc + chunk) * Fp + f) * 32 + q * 8;
    uint4 ph, pl;
    ph.x = hi[0] | ((uint32_t)hi[1] << 16); ph.y = hi[2] | ((uint32_t)hi[3] << 16);
    ph.z = hi[4] | ((uint32_t)hi[5] << 16); ph.w = hi[6] | ((uint32_t)hi[7] << 16);
    pl.x = lo[0] | ((uint32_t)lo[1] << 16); pl.y = lo[2] | ((uint32_t)lo[3] << 16);
    pl.z = lo[4] | ((uint32_t)lo[5] << 16); pl.w = lo[6] | ((uint32_t)lo[7] << 16);
    *(uint4*)(S_hi + o) = ph;
    *(uint4*)(S_lo + o) = pl;
}
hipError_t t2s_launch_melwin_planes(const float* mel, int B, int n_mel, int frames, int nlag, int Fp, u16* S_hi, u16* S_lo,
                                    hipStream_t stream) {
    dim3 grid((Fp + 63) / 64, (nlag * n_mel + 31) / 32, B);
    hipLaunchKernelGGL(melwin_planes_kernel, grid, dim3(256), 0, stream, mel, n_mel, frames, nlag, Fp, S_hi, S_lo);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// audio [B][T] <-> z [B][G][L], z[b][g][t] = audio[b][G*t + g]  (reference glow.py:223 / 291)
__global__ void audio_squeeze_kernel(const float* audio, float* z, int B, int T, int G, int L, int unsq) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int b = blockIdx.y;
    if (t >= L) return;
    for (int g = 0; g < G; ++g) {
        if (unsq)
            ((float*)audio)[(size_t)b * T + (size_t)G * t + g] = z[((size_t)b * G + g) * L + t];
        else
            z[((size_t)b * G + g) * L + t] = audio[(size_t)b * T + (size_t)G * t + g];
    }
}
hipError_t t2s_launch_audio_squeeze(const float* audio, float* z, int B, int T, int n_group, int L, int unsqueeze,
                                    hipStream_t stream) {
    hipLaunchKernelGGL(audio_squeeze_kernel, dim3((L + 255) / 256, B), dim3(256), 0, stream, audio, z, B, T, n_group,
                       L, unsqueeze);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// Invertible 1x1 conv applied in place to channels [c_off, c_off + n_rem) of z [B][G][L]
// (reference glow.py:82-102; the reverse direction passes W^-1).
__global__ void convinv_kernel(float* z, const float* __restrict__ W, int G, int c_off, int n, int L) {
    __shared__ float w[16 * 16];
    if (threadIdx.x < n * n) w[threadIdx.x] = W[threadIdx.x];
    __syncthreads();
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int b = blockIdx.y;
    if (t >= L) return;
    float* base = z + ((size_t)b * G + c_off) * L + t;
    float v[16], o[16];
    for (int j = 0; j < n; ++j) v[j] = base[(size_t)j * L];
    for (int i = 0; i < n; ++i) {
        float s = 0.f;
        for (int j = 0; j < n; ++j) s += w[i * n + j] * v[j];
        o[i] = s;
    }
    for (int i = 0; i < n; ++i) base[(size_t)i * L] = o[i];
}
hipError_t t2s_launch_convinv(float* z, const float* W, int B, int n_group, int c_off, int n_rem, int L,
                              hipStream_t stream) {
    hipLaunchKernelGGL(convinv_kernel, dim3((L + 255) / 256, B), dim3(256), 0, stream, z, W, n_group, c_off, n_rem, L);
    return hipGetLastError();
}

// log(det W) * scale and (optionally) W^-1 for an n x n matrix, n <= 16, by partial-pivot
// Gauss-Jordan in one thread (reference glow.py:90-91,100: W.inverse(), torch.logdet(W)).
// A negative determinant yields NaN, as torch.logdet does.
__global__ void small_logdet_inv_kernel(const float* W, int n, float scale, float* logdet_out, float* inv_out) {
    __shared__ double a[16][32];        // one thread; LDS (not scratch) keeps the pivot loop fast
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) {
            a[i][j] = W[i * n + j];
            a[i][n + j] = (i == j) ? 1.0 : 0.0;
        }
    double logabs = 0.0;
    int sign = 1;
    for (int c = 0; c < n; ++c) {
        int piv = c;
        double best = fabs(a[c][c]);
        for (int r = c + 1; r < n; ++r)
            if (fabs(a[r][c]) > best) { best = fabs(a[r][c]); piv = r; }
        if (piv != c) {
            for (int j = 0; j < 2 * n; ++j) { double tmp = a[c][j]; a[c][j] = a[piv][j]; a[piv][j] = tmp; }
            sign = -sign;
        }
        const double d = a[c][c];
        if (d < 0) sign = -sign;
        logabs += log(fabs(d));
        const double inv = 1.0 / d;
        for (int j = 0; j < 2 * n; ++j) a[c][j] *= inv;
        for (int r = 0; r < n; ++r) {
            if (r == c) continue;
            const double f = a[r][c];
            if (f == 0.0) continue;
            for (int j = 0; j < 2 * n; ++j) a[r][j] -= f * a[c][j];
        }
    }
    if (logdet_out) *logdet_out = sign > 0 ? (float)(logabs * (double)scale) : __builtin_nanf("");
    if (inv_out)
        for (int i = 0; i < n; ++i)
            for (int j = 0; j < n; ++j) inv_out[i * n + j] = (float)a[i][n + j];
}
// all flows in one launch: block k handles matrix k (device table of pointers / sizes)
static __device__ __forceinline__ void small_logdet_one(const SmallMatJob& j, float scale, double (*a)[32]) {
    const int n = (int)j.n;
    for (int i = 0; i < n; ++i)
        for (int c = 0; c < n; ++c) {
            a[i][c] = j.W[i * n + c];
            a[i][n + c] = (i == c) ? 1.0 : 0.0;
        }
    double logabs = 0.0;
    int sign = 1;
    for (int c = 0; c < n; ++c) {
        int piv = c;
        double best = fabs(a[c][c]);
        for (int r = c + 1; r < n; ++r)
            if (fabs(a[r][c]) > best) { best = fabs(a[r][c]); piv = r; }
        if (piv != c) {
            for (int q = 0; q < 2 * n; ++q) { double tmp = a[c][q]; a[c][q] = a[piv][q]; a[piv][q] = tmp; }
            sign = -sign;
        }
        const double d = a[c][c];
        if (d < 0) sign = -sign;
        logabs += log(fabs(d));
        const double inv = 1.0 / d;
        for (int q = 0; q < 2 * n; ++q) a[c][q] *= inv;
        for (int r = 0; r < n; ++r) {
            if (r == c) continue;
            const double f = a[r][c];
            if (f == 0.0) continue;
            for (int q = 0; q < 2 * n; ++q) a[r][q] -= f * a[c][q];
        }
    }
    if (j.logdet_out) *j.logdet_out = sign > 0 ? (float)(logabs * (double)scale) : __builtin_nanf("");
    if (j.inv_out)
        for (int i = 0; i < n; ++i)
            for (int c = 0; c < n; ++c) j.inv_out[i * n + c] = (float)a[i][n + c];
}
__global__ void small_logdet_batch_kernel(const SmallMatJob* jobs, float scale) {
    __shared__ double a[16][32];
    if (threadIdx.x != 0) return;
    small_logdet_one(jobs[blockIdx.x], scale, a);
}
// the same with the table passed BY VALUE (<= 16 matrices): no device table, so no host -> device copy per forward - a
// pageable copy blocks the host until the stream has drained, and the GPU then idles while the host prepares the next call
struct SmallMatJobs16 { SmallMatJob j[16]; };
__global__ void small_logdet_batch_val_kernel(const SmallMatJobs16 jobs, float scale) {
    __shared__ double a[16][32];
    if (threadIdx.x != 0) return;
    small_logdet_one(jobs.j[blockIdx.x], scale, a);
}
hipError_t t2s_launch_small_logdet_batch(const SmallMatJob* jobs, int n_jobs, float scale, hipStream_t stream) {
    hipLaunchKernelGGL(small_logdet_batch_kernel, dim3(n_jobs), dim3(64), 0, stream, jobs, scale);
    return hipGetLastError();
}
hipError_t t2s_launch_small_logdet_batch_host(const SmallMatJob* host_jobs, int n_jobs, float scale, hipStream_t stream) {
    if (n_jobs > 16) return hipErrorInvalidValue;
    SmallMatJobs16 v;
    memset(&v, 0, sizeof(v));
    for (int i = 0; i < n_jobs; ++i) v.j[i] = host_jobs[i];
    hipLaunchKernelGGL(small_logdet_batch_val_kernel, dim3(n_jobs), dim3(64), 0, stream, v, scale);
    return hipGetLastError();
}

hipError_t t2s_launch_small_logdet_inv(const float* W, int n, float scale, float* logdet_out, float* inv_out,
                                       hipStream_t stream) {
    hipLaunchKernelGGL(small_logdet_inv_kernel, dim3(1), dim3(64), 0, stream, W, n, scale, logdet_out, inv_out);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// WN.start: x[c][t] = bias[c] + sum_j w[c][j] * z[b][c_off + j][t]  -> (hi, lo) planes
// (reference glow.py:122-124,156).  One thread = 8 channels of START_TT time steps (64 apart): its 8 x n_half weights and 8 biases
// are fetched once and stay in registers, each time step is n_half loads and one 16-B store per plane (one time step per thread
// spent 40 loads on 2 stores and ran at 1.5 TB/s of plane writes).
#define START_TT 8
__global__ __launch_bounds__(256) void start_kernel(const float* __restrict__ z, const float* __restrict__ w,
                                                    const float* __restrict__ bias, int G, int c_off, int nh, int C,
                                                    int L, int Lp, int halo, u16* X_hi, u16* X_lo) {
    const int tid = threadIdx.x;
    const int q = tid & 3;
    const int tbase = blockIdx.x * (64 * START_TT) + (tid >> 2);
    const int chunk = blockIdx.y;
    const int b = blockIdx.z;
    const int c8 = chunk * 32 + q * 8;
    // static trip counts with guards (nh <= 8): weights stay in registers
    float wv[8][8], bv[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int c = c8 + e;
        bv[e] = c < C ? bias[c] : 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) wv[e][j] = (c < C && j < nh) ? w[c * nh + j] : 0.f;
    }
    const int nchunks = (C + 31) / 32;
    const size_t row0 = ((size_t)b * nchunks + chunk) * Lp + halo;
#pragma unroll 2
    for (int it = 0; it < START_TT; ++it) {
        const int t = tbase + it * 64;
        if (t >= L) break;
        float a0[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) a0[j] = j < nh ? z[((size_t)b * G + c_off + j) * L + t] : 0.f;
        u16 hi[8], lo[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float v = bv[e];
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (j < nh) v += wv[e][j] * a0[j];
            split_bf16(v, hi[e], lo[e]);
        }
        uint4 ph, pl;
        ph.x = hi[0] | ((uint32_t)hi[1] << 16); ph.y = hi[2] | ((uint32_t)hi[3] << 16);
        ph.z = hi[4] | ((uint32_t)hi[5] << 16); ph.w = hi[6] | ((uint32_t)hi[7] << 16);
        pl.x = lo[0] | ((uint32_t)lo[1] << 16); pl.y = lo[2] | ((uint32_t)lo[3] << 16);
        pl.z = lo[4] | ((uint32_t)lo[5] << 16); pl.w = lo[6] | ((uint32_t)lo[7] << 16);
        *(uint4*)(X_hi + (row0 + t) * 32 + q * 8) = ph;
        *(uint4*)(X_lo + (row0 + t) * 32 + q * 8) = pl;
    }
}
hipError_t t2s_launch_start(const float* z, const float* w, const float* bias, int B, int n_group, int c_off,
                            int n_half, int C, int L, int Lp, int halo, u16* X_hi, u16* X_lo, hipStream_t stream) {
    dim3 grid((L + 64 * START_TT - 1) / (64 * START_TT), (C + 31) / 32, B);
    hipLaunchKernelGGL(start_kernel, grid, dim3(256), 0, stream, z, w, bias, n_group, c_off, n_half, C, L, Lp, halo,
                       X_hi, X_lo);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// WN.end (1x1 conv C -> 2*n_half on the f32 skip planes) fused with the affine coupling
//   forward: a1 = exp(log_s) * a1 + b      (reference glow.py:175,241-246)
//   reverse: a1 = (a1 - b) / exp(s)        (reference glow.py:276-280)
// Half a wave (32 lanes = one 128-B skip row) per time step; the 2*n_half dot products are reduced
// with wavefront shuffles.
__global__ __launch_bounds__(256) void end_affine_kernel(const float* __restrict__ skip, const float* __restrict__ w_end,
                                                         const float* __restrict__ b_end, float* z, float* log_s,
                                                         float* wn_out, int G, int c_off, int nh, int C, int L, int Lp,
                                                         int halo, int reverse) {
    const int tid = threadIdx.x;
    const int l32 = tid & 31;
    const int t = blockIdx.x * 8 + (tid >> 5);
    const int b = blockIdx.y;
    const int nj = 2 * nh;
    const int nchunks = (C + 31) / 32;
    float part[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) part[j] = 0.f;
    const bool tv = t < L;
    if (tv) {
        for (int ck = 0; ck < nchunks; ++ck) {
            const int c = ck * 32 + l32;
            if (c < C) {
                const float v = skip[(((size_t)b * nchunks + ck) * Lp + halo + t) * 32 + l32];
#pragma unroll
                for (int j = 0; j < 16; ++j)
                    if (j < nj) part[j] += v * w_end[j * C + c];
            }
        }
    }
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        if (j < nj) {
#pragma unroll
            for (int off = 16; off > 0; off >>= 1) part[j] += __shfl_xor(part[j], off, 64);
        }
    }
    if (tv && l32 < nh) {
        float bb = 0.f, ls = 0.f;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            if (j == l32) bb = part[j];
            if (j == nh + l32) ls = part[j];
        }
        bb += b_end[l32];
        ls += b_end[nh + l32];
        float* zp = z + ((size_t)b * G + c_off + nh + l32) * L + t;
        const float a1 = *zp;
        *zp = reverse ? (a1 - bb) / expf(ls) : expf(ls) * a1 + bb;
        if (log_s) log_s[((size_t)b * nh + l32) * L + t] = ls;
        if (wn_out) {      // training: keep the coupling network's output (b ; log_s) for the backward pass
            wn_out[((size_t)b * 2 * nh + l32) * L + t] = bb;
            wn_out[((size_t)b * 2 * nh + nh + l32) * L + t] = ls;
        }
    }
}
hipError_t t2s_launch_end_affine(const float* skip, const float* w_end, const float* b_end, float* z, float* log_s,
                                 float* wn_out, int B, int n_group, int c_off, int n_half, int C, int L, int Lp, int halo,
                                 int reverse, hipStream_t stream) {
    hipLaunchKernelGGL(end_affine_kernel, dim3((L + 7) / 8, B), dim3(256), 0, stream, skip, w_end, b_end, z, log_s, wn_out,
                       n_group, c_off, n_half, C, L, Lp, halo, reverse);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// WN.end folded into the skip path (forward / infer without autograd).  The skip sum only ever feeds
// WN.end (reference glow.py:172-175):  end(sum_i skip_i) = sum_i (W_end W_skip,i) acts_i + W_end sum_i b_skip,i + b_end,
// and W_end W_skip,i is only [2*n_half x C].  So the skip half of every res_skip GEMM (9 % of the FLOPs)
// and the f32 skip accumulator (2 x 33 MB of traffic per layer) disappear; the gate GEMM's epilogue
// applies the 8-row product to the gate outputs it already holds in registers.
__global__ __launch_bounds__(256) void endfold_weights_kernel(const EndFoldJob* __restrict__ jobs, int C) {
    extern __shared__ float s_ws[];              // [8][C]: W_end[j][o] * scale[o]
    __shared__ float red[4];
    const EndFoldJob j = jobs[blockIdx.x];
    const int tid = threadIdx.x;
    const int nj = (int)j.nj;
    for (int i = tid; i < 8 * C; i += 256) {
        const int r = i / C, o = i - r * C;
        s_ws[i] = (r < nj) ? j.w_end[(size_t)r * C + o] * (j.scale ? j.scale[o] : 1.f) : 0.f;
    }
    __syncthreads();
    const int c = blockIdx.y * 256 + tid;
    if (c < C) {
        float acc[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) acc[r] = 0.f;
        for (int o = 0; o < C; ++o) {
            const float v = j.v_skip[(size_t)o * C + c];
#pragma unroll
            for (int r = 0; r < 8; ++r) acc[r] += s_ws[r * C + o] * v;
        }
        // scatter into the A-fragment layout the gate epilogue reads: lane = q*16 + row j, element e = half*4 + reg
        const int mt = c >> 7, wr = (c >> 6) & 1, pair = (c >> 5) & 1, half = (c >> 4) & 1, q = (c >> 2) & 3, reg = c & 3;
        u16* base = j.fold_A + ((size_t)((mt * 2 + wr) * 2 + pair) * 2 * 64) * 8;
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            u16 h, l;
            split_bf16(acc[r], h, l);
            const int lane = q * 16 + r;
            base[(size_t)lane * 8 + half * 4 + reg] = h;
            base[(size_t)(64 + lane) * 8 + half * 4 + reg] = l;
        }
    }
    if (blockIdx.y == 0) {
        for (int r = 0; r < 8; ++r) {
            float s = 0.f;
            if (r < nj)
                for (int o = tid; o < C; o += 256) s += j.w_end[(size_t)r * C + o] * j.b_skip[o];
            s = wave_sum(s);
            __syncthreads();
            if ((tid & 63) == 0) red[tid >> 6] = s;
            __syncthreads();
            if (tid == 0) j.bes[r] = red[0] + red[1] + red[2] + red[3];
        }
    }
}
hipError_t t2s_launch_endfold_weights(const EndFoldJob* jobs, int n_jobs, int C, hipStream_t stream) {
    hipLaunchKernelGGL(endfold_weights_kernel, dim3(n_jobs, (C + 255) / 256), dim3(256), (size_t)8 * C * sizeof(float),
                       stream, jobs, C);
    return hipGetLastError();
}

// out[j][t] = b_end[j] + sum_layers bes[layer][j] + sum_slots fold_acc[slot][b][j][t], then the affine coupling
// (reference glow.py:175,241-246; reverse: glow.py:276-280)
__global__ void end_fold_affine_kernel(const float* __restrict__ fold_acc, int nslots, const float* __restrict__ bes,
                                       int n_layers, const float* __restrict__ b_end, float* z, float* log_s, float* wn_out,
                                       int B, int G, int c_off, int nh, int L, int reverse) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int i = blockIdx.y, b = blockIdx.z;
    if (t >= L) return;
    float bb = b_end[i], ls = b_end[nh + i];
    for (int l = 0; l < n_layers; ++l) { bb += bes[l * 8 + i]; ls += bes[l * 8 + nh + i]; }
    for (int s = 0; s < nslots; ++s) {
        bb += fold_acc[(((size_t)s * B + b) * 8 + i) * L + t];
        ls += fold_acc[(((size_t)s * B + b) * 8 + nh + i) * L + t];
    }
    float* zp = z + ((size_t)b * G + c_off + nh + i) * L + t;
    const float a1 = *zp;
    *zp = reverse ? (a1 - bb) / expf(ls) : expf(ls) * a1 + bb;
    if (log_s) log_s[((size_t)b * nh + i) * L + t] = ls;
    if (wn_out) {           // training: (b ; log_s) as WN.end's output, what the coupling's backward reads
        wn_out[((size_t)b * 2 * nh + i) * L + t] = bb;
        wn_out[((size_t)b * 2 * nh + nh + i) * L + t] = ls;
    }
}
hipError_t t2s_launch_end_fold_affine(const float* fold_acc, int nslots, const float* bes, int n_layers,
                                      const float* b_end, float* z, float* log_s, float* wn_out, int B, int n_group, int c_off,
                                      int n_half, int L, int reverse, hipStream_t stream) {
    hipLaunchKernelGGL(end_fold_affine_kernel, dim3((L + 255) / 256, n_half, B), dim3(256), 0, stream, fold_acc, nslots, bes,
                       n_layers, b_end, z, log_s, wn_out, B, n_group, c_off, n_half, L, reverse);
    return hipGetLastError();
}
