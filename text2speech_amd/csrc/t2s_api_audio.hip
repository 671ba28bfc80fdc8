// extern "C" boundary of the audio front-end / back-end (STFT, mel, inverse STFT, denoiser glue).
#include "../../include/t2s_hip.h"
#include "t2s_kernels.h"
#include "audio_ops.h"
#include "tacotron_ops.h"

#include <string.h>

extern "C" int t2s_internal_fail_hip(int e);
#define T2S_CHECK_HIP(expr)                                          \
    do {                                                             \
        hipError_t _e = (expr);                                      \
        if (_e != hipSuccess) return t2s_internal_fail_hip((int)_e); \
    } while (0)

// y[item][row] = sum_k W[row][k] x[item * sx + k]  per batch element, through the GEMV / matrix-core dispatcher
static hipError_t frames_gemm(const float* W, int rows, int K, const float* x, long sx, int items, float* y, long sy_item,
                              long sy_row, hipStream_t s) {
    GemvArgs g;
    memset(&g, 0, sizeof(g));
    g.W1 = W; g.ld1 = K; g.k1 = K; g.x1 = x; g.n1 = K; g.sx1 = sx;
    g.y = y; g.sy_item = sy_item; g.sy_row = sy_row; g.rows = rows; g.items = items; g.mask_scale = 1.f;
    return t2s_launch_gemv(g, s);
}

extern "C" {

int t2s_stft_transform(const float* audio, int B, int T, const float* fwd_basis, int n_fft, int hop, float* xp, long ldp,
                       float* ft, float* mag, float* phase, float* magT, long ld_mt, void* stream) {
    const int c = n_fft / 2 + 1, F = T / hop + 1;
    if (!audio || !fwd_basis || !xp || !ft || B <= 0 || T <= n_fft / 2 || n_fft <= 0 || (n_fft & 15) || hop <= 0 || (hop & 3) ||
        (ldp & 3) || ldp < T + n_fft || n_fft > 4096 || (magT && (ld_mt < c || (ld_mt & 15))))
        return T2S_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    T2S_CHECK_HIP(t2s_launch_reflect_pad(audio, B, T, n_fft / 2, xp, ldp, s));
    for (int b = 0; b < B; ++b)
        T2S_CHECK_HIP(frames_gemm(fwd_basis, 2 * c, n_fft, xp + (size_t)b * ldp, hop, F, ft + (size_t)b * F * 2 * c, 2 * c, 1, s));
    if (mag || phase || magT) T2S_CHECK_HIP(t2s_launch_stft_mag_phase(ft, B, F, c, 2 * c, mag, phase, magT, ld_mt, s));
    return T2S_OK;
}

int t2s_mel_from_mag(const float* magT, long ld_mt, int B, int F, const float* mel_basis_p, int n_mel, float clip, float* mel,
                     void* stream) {
    if (!magT || !mel_basis_p || !mel || B <= 0 || F <= 0 || n_mel <= 0 || (ld_mt & 15)) return T2S_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    for (int b = 0; b < B; ++b)
        T2S_CHECK_HIP(frames_gemm(mel_basis_p, n_mel, (int)ld_mt, magT + (size_t)b * F * ld_mt, ld_mt, F,
                                  mel + (size_t)b * n_mel * F, 1, F, s));
    if (clip > 0.f) T2S_CHECK_HIP(t2s_launch_log_clamp(mel, (size_t)B * n_mel * F, clip, s));
    return T2S_OK;
}

int t2s_stft_inverse(const float* mag, const float* phase, int B, int F, int n_fft, int hop, const float* inv_basis_t, long ld_rc,
                     const float* bias, float strength, const float* win_sq, float tiny, float* rc, float* frames, float* out,
                     void* stream) {
    const int c = n_fft / 2 + 1;
    if (!mag || !phase || !inv_basis_t || !rc || !frames || !out || B <= 0 || F <= 1 || n_fft <= 0 || hop <= 0 ||
        ld_rc < 2 * c || (ld_rc & 15))
        return T2S_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    T2S_CHECK_HIP(t2s_launch_stft_recombine(mag, phase, B, F, c, bias, strength, rc, ld_rc, s));
    for (int b = 0; b < B; ++b)
        T2S_CHECK_HIP(frames_gemm(inv_basis_t, n_fft, (int)ld_rc, rc + (size_t)b * F * ld_rc, ld_rc, F,
                                  frames + (size_t)b * F * n_fft, n_fft, 1, s));
    T2S_CHECK_HIP(t2s_launch_stft_overlap_add(frames, win_sq, B, F, n_fft, hop, (float)n_fft / (float)hop, tiny, out,
                                              hop * (F - 1), s));
    return T2S_OK;
}

}  // extern "C"
