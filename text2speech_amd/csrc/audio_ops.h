// Launchers of the audio front-end / back-end kernels (audio_ops.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>

hipError_t t2s_launch_reflect_pad(const float* x, int B, int T, int pad, float* xp, long ldp, hipStream_t s);
hipError_t t2s_launch_stft_mag_phase(const float* ft, int B, int F, int c, long ld_ft, float* mag, float* phase, float* magT,
                                     long ld_mt, hipStream_t s);
hipError_t t2s_launch_stft_recombine(const float* mag, const float* phase, int B, int F, int c, const float* bias,
                                     float strength, float* rc, long ld_rc, hipStream_t s);
hipError_t t2s_launch_stft_overlap_add(const float* frames, const float* win_sq, int B, int F, int n_fft, int hop, float scale,
                                       float tiny, float* out, int n_out, hipStream_t s);
hipError_t t2s_launch_log_clamp(float* x, size_t n, float clip, hipStream_t s);
