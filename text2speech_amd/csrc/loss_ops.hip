// Tacotron2Loss (reference tacotron/loss_function.py:3-18): MSE(mel_out, target) + MSE(mel_out_postnet, target) +
// BCEWithLogits(gate_out, gate_target), every term a mean over its elements.  One pass computes the three sums AND the three
// gradients (d loss / d mel_out etc. for an upstream gradient of 1); the reduction is two-stage (per-workgroup partials in
// double, then one workgroup) so the value does not depend on scheduling.
#include "t2s_common.h"
#include "t2s_kernels.h"
#include "taco_bwd_ops.h"

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
