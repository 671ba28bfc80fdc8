/* libt2s_hip — C ABI of the MI355X (gfx950) Tacotron-2 / WaveGlow hot path.
 *
 * The reference (DonggeunYu/Text2Speech) has no FFI of its own: its hot path is
 * the torch op sequence inside waveglow/glow.py and tacotron/tacotron.py.  Each
 * entry point below replaces one such op sequence; the reference lines are cited
 * per function.  The Python side (text2speech_amd/glow.py, .../tacotron.py) binds
 * these with ctypes and keeps the reference's module API.
 *
 * Conventions
 *  - every pointer is a DEVICE pointer owned by the caller (PyTorch allocations);
 *    the library allocates nothing and keeps no global state;
 *  - `stream` is a hipStream_t passed as void*; all work is enqueued on it and
 *    nothing synchronises;
 *  - return value 0 = success; a negative T2S_E* code otherwise (never throws);
 *  - entry points are re-entrant (autograd calls backward from another thread).
 *
 * "Planes": activations on the WN path are channel-last, split-bf16:
 *      x ~= float(hi) + float(lo),   plane[b][c/32][row][c%32] (bf16),
 *      row = halo + t, 0 <= row < Lp,  Lp = t2s_plane_rows(L, halo);
 *    rows outside [halo, halo+L) must be zero (allocate zero-filled once; the
 *    kernels never write them).  f32 "skip" planes use the same indexing.
 */
#ifndef T2S_HIP_H
#define T2S_HIP_H
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif

#define T2S_OK 0
#define T2S_EINVAL (-1)  /* bad argument (null pointer, size, alignment) */
#define T2S_EHIP (-2)    /* HIP runtime reported an error at launch */

#define T2S_PERM_NONE 0
#define T2S_PERM_GATE 1  /* rows o<C: tanh half, o>=C: sigmoid half, interleaved per 16 channels */
#define T2S_ACT_NONE 0
#define T2S_ACT_RELU 1
#define T2S_ACT_TANH 2

int t2s_abi_version(void);
const char* t2s_error_string(int code);
/* last HIP error text seen by a failing entry point on this thread ("" if none) */
const char* t2s_last_hip_error(void);

/* rows per 32-channel chunk of a plane holding L time steps with `halo` zero rows either side */
int t2s_plane_rows(int L, int halo);
/* packed row count (multiple of 256) for `rows` GEMM output rows */
int t2s_padded_rows(int rows);

/* weight_norm + pack one conv weight v[O][Cin][Kt] (gain g[O] or NULL) into the GEMM A planes
 * A_hi/A_lo [nk][Mpad][32] bf16 at packed-K offset `koff` (tap-major, tap stride Cin_pad), and its
 * bias into bias_out[Mpad].  Replaces torch.nn.utils.weight_norm's per-forward recompute
 * (reference glow.py:123,138,142,151). */
int t2s_pack_conv_weight(const float* v, const float* g, const float* bias_in, int O, int Cin, int Kt, int perm,
                         int C_gate, int row_off, int Mpad, int koff, int Cin_pad, void* A_hi, void* A_lo,
                         float* bias_out, int bias_accumulate, void* stream);

/* w[O][K] = v * g / ||v||  for a small weight-normed 1x1 conv (WN.start; reference glow.py:122-124) */
int t2s_weightnorm_small(const float* v, const float* g, int O, int K, float* w, void* stream);

/* ConvTranspose1d(n_mel,n_mel,ksize,stride) + trim to L*n_group samples + squeeze -> conditioning planes
 * S_hi/S_lo [B][ceil(n_mel*n_group/32)][Lp][32]  (reference glow.py:215-221; infer 252-258). */
int t2s_wg_upsample_squeeze(const float* mel, const float* W, const float* bias, int B, int n_mel, int frames,
                            int ksize, int stride, int n_group, int L, int Lp, int halo, void* S_hi, void* S_lo,
                            void* stream);

/* audio[B][T] -> z[B][n_group][L] (unsqueeze=0, reference glow.py:223) or back (unsqueeze=1, glow.py:291) */
int t2s_wg_audio_squeeze(float* audio, float* z, int B, int T, int n_group, int L, int unsqueeze, void* stream);

/* in-place invertible 1x1 conv on channels [c_off, c_off+n_rem) of z[B][n_group][L] (reference glow.py:82-102) */
int t2s_wg_convinv(float* z, const float* W, int B, int n_group, int c_off, int n_rem, int L, void* stream);

/* logdet_out = scale*log(det W) (NaN if det<0) and/or inv_out = W^-1, W is n x n, n<=16
 * (reference glow.py:90-91,100) */
int t2s_small_logdet_inv(const float* W, int n, float scale, float* logdet_out, float* inv_out, void* stream);

/* WN.start: x = w[C][n_half] * z[:, c_off:c_off+n_half] + bias -> planes X_hi/X_lo (reference glow.py:156) */
int t2s_wg_start(const float* z, const float* w, const float* bias, int B, int n_group, int c_off, int n_half, int C,
                 int L, int Lp, int halo, void* X_hi, void* X_lo, void* stream);

/* One WN layer, first half: acts = tanh/sigmoid gate of (dilated conv(x) + 1x1 conv(spect) + bias)
 * (reference glow.py:159-162 with fused_add_tanh_sigmoid_multiply glow.py:33-40).
 * A planes hold [taps*xc + sc] K-steps x Mpad rows packed with T2S_PERM_GATE. */
int t2s_wg_in_cond_gate(const void* A_hi, const void* A_lo, const float* bias, const void* X_hi, const void* X_lo,
                        const void* S_hi, const void* S_lo, void* acts_hi, void* acts_lo, int B, int C, int n_cond,
                        int taps, int dilation, int L, int Lp, int halo, int Mpad, void* stream);

/* One WN layer, second half: rs = 1x1 conv(acts) + bias; x += rs[:n_res]; skip (+)= rs[n_res:]
 * (reference glow.py:164-174).  n_res = C (layers 0..n-2) or 0 (last layer). */
int t2s_wg_res_skip(const void* A_hi, const void* A_lo, const float* bias, const void* acts_hi, const void* acts_lo,
                    void* X_hi, void* X_lo, float* skip, int B, int C, int n_res, int skip_init, int L, int Lp,
                    int halo, int Mpad, void* stream);

/* WN.end + affine coupling on channels [c_off+n_half, c_off+2*n_half) of z; writes log_s[B][n_half][L]
 * if non-NULL (reference glow.py:175,241-246; reverse=1: glow.py:276-280) */
int t2s_wg_end_affine(const float* skip, const float* w_end, const float* b_end, float* z, float* log_s, int B,
                      int n_group, int c_off, int n_half, int C, int L, int Lp, int halo, int reverse, void* stream);

/* Generic split-bf16 conv1d-as-GEMM with bias + activation epilogue:
 *   out[b][o][t] = act(bias[o] + sum_{tap,c} W[o][c][tap] * x[b][c][t + (tap - taps/2)*dil])
 * A planes packed with T2S_PERM_NONE.  Writes planes O_hi/O_lo (may be NULL) and/or out_f32 [B][C][L]
 * (may be NULL).  Used for the Tacotron-2 encoder / postnet convolutions and LSTM input projections
 * (reference tacotron.py:177-194, modules.py:94-137). */
int t2s_conv_bias_act(const void* A_hi, const void* A_lo, const float* bias, const void* X_hi, const void* X_lo,
                      void* O_hi, void* O_lo, float* out_f32, int B, int Cin, int Cout, int taps, int dilation,
                      int act, int L, int Lp, int halo, int Mpad, void* stream);

#ifdef __cplusplus
}
#endif
#endif
