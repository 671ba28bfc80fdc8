// Tacotron2Loss (reference tacotron/loss_function.py:3-18): MSE(mel_out, target) + MSE(mel_out_postnet, target) +
// BCEWithLogits(gate_out, gate_target), every term a mean over its elements.  One pass computes the three sums AND the three
// gradients (d loss / d mel_out etc. for an upstream gradient of 1); the reduction is two-stage (per-workgroup partials in
// double, then one workgroup) so the value does not depend on scheduling.
#include "t2s_common.h"
#include "t2s_kernels.h"
#include "taco_bwd_ops.h"

#include <string.h>

#define T2S_LOSS_BLOCKS 256

__global__ __launch_bounds__(256) void taco_loss_partial_kernel(const float* __restrict__ mel, const float* __restrict__ post,
                                                                const float* __restrict__ target, size_t n_mel,
                                                                const float* __restrict__ gate, const float* __restrict__ gate_t,
                                                                size_t n_gate, float* d_mel, float* d_post, float* d_gate,
                                                                double* partial) {
    __shared__ double red[3][4];
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    const float s_mel = 2.0f / (float)n_mel, s_gate = 1.0f / (float)n_gate;
    double a0 = 0.0, a1 = 0.0, a2 = 0.0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_mel; i += stride) {
        const float t = target[i];
        const float e0 = mel[i] - t, e1 = post[i] - t;
        a0 += (double)(e0 * e0);
        a1 += (double)(e1 * e1);
        if (d_mel) d_mel[i] = e0 * s_mel;
        if (d_post) d_post[i] = e1 * s_mel;
    }
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_gate; i += stride) {
        const float x = gate[i], y = gate_t[i];
        // max(x, 0) - x y + log(1 + exp(-|x|)): the numerically stable form torch uses
        a2 += (double)(fmaxf(x, 0.f) - x * y + log1pf(expf(-fabsf(x))));
        if (d_gate) d_gate[i] = (1.0f / (1.0f + expf(-x)) - y) * s_gate;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        a0 += __shfl_xor(a0, off, 64); a1 += __shfl_xor(a1, off, 64); a2 += __shfl_xor(a2, off, 64);
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) { red[0][wave] = a0; red[1][wave] = a1; red[2][wave] = a2; }
    __syncthreads();
    if (threadIdx.x < 3)
        partial[(size_t)blockIdx.x * 3 + threadIdx.x] = (red[threadIdx.x][0] + red[threadIdx.x][1]) + (red[threadIdx.x][2] + red[threadIdx.x][3]);
}

__global__ void taco_loss_final_kernel(const double* partial, int nb, size_t n_mel, size_t n_gate, float* out) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0;
    for (int b = 0; b < nb; ++b) { s0 += partial[b * 3]; s1 += partial[b * 3 + 1]; s2 += partial[b * 3 + 2]; }
    const double mel_loss = s0 / (double)n_mel + s1 / (double)n_mel, gate_loss = s2 / (double)n_gate;
    out[0] = (float)(mel_loss + gate_loss);
    out[1] = (float)mel_loss;
    out[2] = (float)gate_loss;
}

hipError_t t2s_launch_taco_loss(const float* mel, const float* post, const float* target, size_t n_mel, const float* gate,
                                const float* gate_t, size_t n_gate, float* d_mel, float* d_post, float* d_gate, double* partial,
                                float* out, hipStream_t stream) {
    hipLaunchKernelGGL(taco_loss_partial_kernel, dim3(T2S_LOSS_BLOCKS), dim3(256), 0, stream, mel, post, target, n_mel, gate,
                       gate_t, n_gate, d_mel, d_post, d_gate, partial);
    hipLaunchKernelGGL(taco_loss_final_kernel, dim3(1), dim3(64), 0, stream, partial, T2S_LOSS_BLOCKS, n_mel, n_gate, out);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// WaveGlowLoss (reference waveglow/glow.py:43-59): ( sum z^2 / (2 sigma^2) - sum_k sum log_s_k - sum_k log_det_W_k ) / numel(z).
// One pass over z (also writes d loss / d z = z / (sigma^2 N)) and over the n_flows log_s tensors (table by value, <= 16);
// two-stage reduction in double as above.  d loss / d log_s = d loss / d log_det = -1 / N are constants the caller fills in.
struct WgLossJobs { const float* log_s[16]; size_t n[16]; const float* log_det; int n_flows; };

__global__ __launch_bounds__(256) void wg_loss_partial_kernel(const float* __restrict__ z, size_t n_z, WgLossJobs jobs, float dz_scale,
                                                              float* d_z, double* partial) {
    __shared__ double red[2][4];
    const size_t stride = (size_t)gridDim.x * blockDim.x, i0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    double a0 = 0.0, a1 = 0.0;
    for (size_t i = i0; i < n_z; i += stride) {
        const float v = z[i];
        a0 += (double)(v * v);
        if (d_z) d_z[i] = v * dz_scale;
    }
    for (int k = 0; k < jobs.n_flows; ++k) {
        const float* p = jobs.log_s[k];
        for (size_t i = i0; i < jobs.n[k]; i += stride) a1 += (double)p[i];
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { a0 += __shfl_xor(a0, off, 64); a1 += __shfl_xor(a1, off, 64); }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) { red[0][wave] = a0; red[1][wave] = a1; }
    __syncthreads();
    if (threadIdx.x < 2)
        partial[(size_t)blockIdx.x * 2 + threadIdx.x] = (red[threadIdx.x][0] + red[threadIdx.x][1]) + (red[threadIdx.x][2] + red[threadIdx.x][3]);
}
__global__ void wg_loss_final_kernel(const double* partial, int nb, const float* log_det, int n_flows, double inv_2s2, size_t n_z,
                                     float* out) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0;
    for (int b = 0; b < nb; ++b) { s0 += partial[b * 2]; s1 += partial[b * 2 + 1]; }
    for (int k = 0; k < n_flows; ++k) s2 += (double)log_det[k];
    out[0] = (float)((s0 * inv_2s2 - s1 - s2) / (double)n_z);
}
hipError_t t2s_launch_waveglow_loss(const float* z, size_t n_z, const float* const* log_s, const size_t* n_log_s, int n_flows,
                                    const float* log_det, float sigma, float* d_z, double* partial, float* out,
                                    hipStream_t stream) {
    if (n_flows > 16) return hipErrorInvalidValue;
    WgLossJobs j;
    memset(&j, 0, sizeof(j));
    for (int k = 0; k < n_flows; ++k) { j.log_s[k] = log_s[k]; j.n[k] = n_log_s[k]; }
    j.log_det = log_det; j.n_flows = n_flows;
    hipLaunchKernelGGL(wg_loss_partial_kernel, dim3(T2S_LOSS_BLOCKS), dim3(256), 0, stream, z, n_z, j,
                       1.0f / (sigma * sigma * (float)n_z), d_z, partial);
    hipLaunchKernelGGL(wg_loss_final_kernel, dim3(1), dim3(64), 0, stream, partial, T2S_LOSS_BLOCKS, log_det, n_flows,
                       0.5 / ((double)sigma * sigma), n_z, out);
    return hipGetLastError();
}
