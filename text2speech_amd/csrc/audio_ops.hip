// Audio front-end / back-end on device: the reference's STFT-as-conv1d (utils/stft.py:72-129),
// TacotronSTFT.mel_spectrogram (utils/layers.py:63-79) and the Denoiser's spectral subtraction (waveglow/denoiser.py:34-40).
//
// The two contractions (frames x windowed Fourier basis, mel basis x magnitudes, and the inverse basis) go through
// t2s_gemv: overlapping frames are just "items" whose stride is the hop, so no frame matrix is ever materialised, and at
// 9+ frames the products run on the f32 matrix cores (sbgemm.hip) - exact f32, which log-mel of quiet bands needs.
// What is left here is the bandwidth-bound glue: reflect padding, magnitude / phase, recombination (with the optional
// bias subtraction of the denoiser fused), overlap-add with the window-sum-square normalisation, log-clamp.
#include "t2s_common.h"
#include "t2s_kernels.h"
#include "audio_ops.h"

// xp[b][i] = x[b][reflect(i - pad)], i in [0, T + 2 pad)   (F.pad(..., mode='reflect'), utils/stft.py:79-83)
__global__ void reflect_pad_kernel(const float* __restrict__ x, int T, int pad, float* __restrict__ xp, long ldp) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int b = blockIdx.y;
    if (i >= T + 2 * pad) return;
    int j = i - pad;
    if (j < 0) j = -j;
    if (j >= T) j = 2 * (T - 1) - j;
    xp[(size_t)b * ldp + i] = x[(size_t)b * T + j];
}

// ft [B][F][ld_ft] (real rows 0..c-1, imaginary rows c..2c-1)  ->  mag, phase [B][c][F]  and  magT [B*F][ld_mt] (zero padded)
__global__ void stft_mag_phase_kernel(const float* __restrict__ ft, int F, int c, long ld_ft, float* __restrict__ mag,
                                      float* __restrict__ phase, float* __restrict__ magT, long ld_mt) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;       // bin (or padding column of magT)
    const int f = blockIdx.y, b = blockIdx.z;
    if (k >= ld_mt && k >= c) return;
    float m = 0.f;
    if (k < c) {
        const float* row = ft + ((size_t)b * F + f) * ld_ft;
        const float re = row[k], im = row[c + k];
        m = sqrtf(re * re + im * im);
        if (mag) mag[((size_t)b * c + k) * F + f] = m;
        if (phase) phase[((size_t)b * c + k) * F + f] = atan2f(im, re);
    }
    if (magT && k < ld_mt) magT[((size_t)b * F + f) * ld_mt + k] = m;
}

// rc[b*F + f][k] = k < c ? m cos(ph) : k < 2c ? m sin(ph) : 0,  m = max(mag - strength * bias[k], 0) when bias is given
// (utils/stft.py:102-103; waveglow/denoiser.py:36-38)
__global__ void stft_recombine_kernel(const float* __restrict__ mag, const float* __restrict__ phase, int F, int c,
                                      const float* __restrict__ bias, float strength, float* __restrict__ rc, long ld_rc) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    const int f = blockIdx.y, b = blockIdx.z;
    if (k >= ld_rc) return;
    float v = 0.f;
    if (k < 2 * c) {
        const int bin = k < c ? k : k - c;
        const size_t i = ((size_t)b * c + bin) * F + f;
        float m = mag[i];
        if (bias) m = fmaxf(m - bias[bin] * strength, 0.f);
        const float ph = phase[i];
        v = k < c ? m * cosf(ph) : m * sinf(ph);
    }
    rc[((size_t)b * F + f) * ld_rc + k] = v;
}

// out[b][n] = scale / wss[p] * sum_f frames[b][f][p - f hop],  p = n + n_fft/2,  wss[p] = sum_f win_sq[p - f hop]
// (conv_transpose1d overlap-add, window-sum-square normalisation where it exceeds tiny, trim: utils/stft.py:105-127)
__global__ void stft_overlap_add_kernel(const float* __restrict__ frames, const float* __restrict__ win_sq, int F, int n_fft,
                                        int hop, float scale, float tiny, float* __restrict__ out, int n_out) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    const int b = blockIdx.y;
    if (n >= n_out) return;
    const int p = n + n_fft / 2;
    int f_lo = (p - n_fft + hop) / hop;            // ceil((p - n_fft + 1) / hop) for p - n_fft + 1 > 0
    if (p - n_fft + 1 <= 0) f_lo = 0;
    int f_hi = p / hop;
    if (f_hi > F - 1) f_hi = F - 1;
    float acc = 0.f, wss = 0.f;
    for (int f = f_lo; f <= f_hi; ++f) {
        const int j = p - f * hop;
        acc += frames[((size_t)b * F + f) * n_fft + j];
        if (win_sq) wss += win_sq[j];
    }
    if (win_sq) {
        if (wss > tiny) acc /= wss;
        acc *= scale;
    }
    out[(size_t)b * n_out + n] = acc;
}

// x <- log(max(x, clip))   (dynamic_range_compression, utils/audio_processing.py:78-84 with C = 1)
__global__ void log_clamp_kernel(float* x, size_t n, float clip) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) x[i] = logf(fmaxf(x[i], clip));
}

hipError_t t2s_launch_reflect_pad(const float* x, int B, int T, int pad, float* xp, long ldp, hipStream_t s) {
    hipLaunchKernelGGL(reflect_pad_kernel, dim3((T + 2 * pad + 255) / 256, B), dim3(256), 0, s, x, T, pad, xp, ldp);
    return hipGetLastError();
}
hipError_t t2s_launch_stft_mag_phase(const float* ft, int B, int F, int c, long ld_ft, float* mag, float* phase, float* magT,
                                     long ld_mt, hipStream_t s) {
    const int cols = (int)(ld_mt > c ? ld_mt : c);
    hipLaunchKernelGGL(stft_mag_phase_kernel, dim3((cols + 255) / 256, F, B), dim3(256), 0, s, ft, F, c, ld_ft, mag, phase, magT,
                       magT ? ld_mt : 0);
    return hipGetLastError();
}
hipError_t t2s_launch_stft_recombine(const float* mag, const float* phase, int B, int F, int c, const float* bias,
                                     float strength, float* rc, long ld_rc, hipStream_t s) {
    hipLaunchKernelGGL(stft_recombine_kernel, dim3((unsigned)((ld_rc + 255) / 256), F, B), dim3(256), 0, s, mag, phase, F, c, bias,
                       strength, rc, ld_rc);
    return hipGetLastError();
}
hipError_t t2s_launch_stft_overlap_add(const float* frames, const float* win_sq, int B, int F, int n_fft, int hop, float scale,
                                       float tiny, float* out, int n_out, hipStream_t s) {
    hipLaunchKernelGGL(stft_overlap_add_kernel, dim3((n_out + 255) / 256, B), dim3(256), 0, s, frames, win_sq, F, n_fft, hop,
                       scale, tiny, out, n_out);
    return hipGetLastError();
}
hipError_t t2s_launch_log_clamp(float* x, size_t n, float clip, hipStream_t s) {
    hipLaunchKernelGGL(log_clamp_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, x, n, clip);
    return hipGetLastError();
}
