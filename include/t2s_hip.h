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
 *    the library allocates no device memory;
 *  - `stream` is a hipStream_t passed as void*; all work is enqueued on it, or
 *    ordered before the point the call leaves it at, and nothing synchronises
 *    with the host;
 *  - return value 0 = success; a negative T2S_E* code otherwise (never throws);
 *  - entry points are re-entrant and thread-safe (autograd calls backward from
 *    another thread), and work on whichever device is current in the calling
 *    thread.  Library-owned state, all of it per device and created on first use
 *    on that device: (a) t2s_taco_bptt_steps keeps two helper streams + five events
 *    per device behind a per-device mutex (it overlaps dependent chains; on every
 *    exit, error exits included, the caller's stream waits for both helpers; with
 *    t2s_taco_decode_steps borrows the first helper the same way for teacher-forced steps with saves);
 *    (b) the GEMM launchers remember per device that they raised the kernel's
 *    dynamic-LDS limit (an idempotent one-bit flag); (c) t2s_last_hip_error()
 *    is per thread.  Nothing else persists between calls.
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
#define T2S_PERM_PAIR8 2 /* table pack only: rows o < C_gate, per group of 32: packed row 16 m + 4 q + e = channel 8 q + 4 m + e */
#define T2S_ACT_NONE 0
#define T2S_ACT_RELU 1
#define T2S_ACT_TANH 2

int t2s_abi_version(void);
/* ABI v4.  Operand format of the split planes this build computes with: 0 = bf16 hi/lo (the shipped library), 1 = fp16 hi/lo
 * (diagnostic build -DT2S_SPLIT_F16: same three MFMA products per MAC, ~22 instead of ~16 significand bits, but no exponent range
 * to spare - the no-grad WaveGlow forward / infer only; the host side refuses the training path and turns a non-finite result,
 * i.e. an operand plane that overflowed 65504, into an error instead of returning it). */
int t2s_operand_format(void);
/* ABI v4.  sizeof(t2s_taco_decoder) / sizeof(t2s_taco_bptt) as this library was compiled: a binding that mirrors the structs field by
 * field (ctypes, cgo, JNI) checks its own size against these before the first call. */
int t2s_sizeof_taco_decoder(void);
int t2s_sizeof_taco_bptt(void);
const char* t2s_error_string(int code);
/* last HIP error text seen by a failing entry point on this thread ("" if none) */
const char* t2s_last_hip_error(void);

/* rows per 32-channel chunk of a plane holding L time steps with `halo` zero rows either side */
int t2s_plane_rows(int L, int halo);
/* packed row count (multiple of 256) for `rows` GEMM output rows */
int t2s_padded_rows(int rows);

/* weight_norm + pack one conv weight v[O][Cin][Kt] (gain g[O] or NULL) into the GEMM A planes
 * A_hi/A_lo [nk][Mpad][32] bf16 at packed-K offset `koff` (tap-major, tap stride Cin_pad), and its
 * bias into bias_out[Mpad].  g_is_scale=1: g is a plain per-row scale (eval BatchNorm fold) instead of
 * a weight-norm gain.  Replaces torch.nn.utils.weight_norm's per-forward recompute
 * (reference glow.py:123,138,142,151). */
int t2s_pack_conv_weight(const float* v, const float* g, int g_is_scale, const float* bias_in, int O, int Cin, int Kt,
                         int perm, int C_gate, int row_off, int Mpad, int koff, int Cin_pad, void* A_hi, void* A_lo,
                         float* bias_out, int bias_accumulate, void* stream);

/* Table-driven form of t2s_pack_conv_weight: ONE launch packs every listed weight (the per-forward
 * weight-norm recompute of all 288 WN convolutions).  `jobs` is a DEVICE array of t2s_pack_job; a workgroup packs 16
 * consecutive output rows, job i owns workgroups [row_start, row_start + ceil(O/16)), row_start being the running sum of
 * ceil(O/16) and total_groups the grand total.  bias_out[p] = bias_in[o] + bias_in2[o] (either may be NULL); leave bias_out
 * NULL on a job that shares its rows' bias with another. */
typedef struct t2s_pack_job {
    const float *v, *g, *bias_in, *bias_in2;
    void *A_hi, *A_lo;
    float* bias_out;
    long row_start;
    long O, Cin, Kt, perm, C_gate, Mpad, koff, Cin_pad, row_off, g_is_scale;
    float* scale_out; /* optional [O]: per-row factor g/|v| that was applied, kept for the backward pass */
} t2s_pack_job;
int t2s_pack_conv_weight_table(const t2s_pack_job* jobs, int n_jobs, long total_groups, void* stream);

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

/* the same for many small matrices in ONE launch (all 12 flows' 1x1-conv weights): jobs is a DEVICE array */
typedef struct t2s_small_mat_job { const float* W; float* logdet_out; float* inv_out; long n; } t2s_small_mat_job;
int t2s_small_logdet_inv_batch(const t2s_small_mat_job* jobs, int n_jobs, float scale, void* stream);
/* Same, `host_jobs` is a HOST array of at most 16 entries that travels as a kernel argument: no device table to upload. */
int t2s_small_logdet_inv_batch_host(const t2s_small_mat_job* host_jobs, int n_jobs, float scale, void* stream);

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
 * and wn_out[B][2*n_half][L] = (b ; log_s) if non-NULL (reference glow.py:175,241-246; reverse=1: glow.py:276-280) */
int t2s_wg_end_affine(const float* skip, const float* w_end, const float* b_end, float* z, float* log_s,
                      float* wn_out, int B, int n_group, int c_off, int n_half, int C, int L, int Lp, int halo, int reverse, void* stream);

/* ---- WN.end folded into the skip path (no-grad forward / infer) ----
 * end(sum_i skip_i) = sum_i (W_end W_skip,i) acts_i + const (reference glow.py:172-175), and W_end W_skip,i is only
 * [2*n_half x C]: the skip half of res_skip_layers and the f32 skip accumulator are never materialised. */
typedef struct t2s_endfold_job {
    const float *w_end, *v_skip, *scale, *b_skip;   /* [nj][C], [C][C] skip rows of weight_v, their g/|v|, their bias */
    void* fold_A;                                   /* out: MFMA A fragments, 8192*ceil(C/128) bf16, zero-initialised */
    float* bes;                                     /* out: [8] */
    long nj, C;
} t2s_endfold_job;
/* jobs: DEVICE array, one per WN layer; run after the weight packing that produced `scale` */
int t2s_wg_endfold_weights(const t2s_endfold_job* jobs, int n_jobs, int C, void* stream);
/* Tile height (256 or 128 packed rows) the folded gate GEMM uses for this shape: 128 where 256-row tiles would leave at least
 * half the chip without a workgroup (short utterances).  t2s_wg_gate_fold_slots follows it; t2s_wg_in_melwin_gate_fold (phase
 * tiles, always 256 rows) returns T2S_EINVAL for shapes where this is 128. */
int t2s_wg_gate_tile_rows(int B, int C, int L);
/* t2s_wg_in_cond_gate plus fold_acc[slot][b][j][t] (+)= (W_end W_skip,i)[j] . acts[:, t] over the slot's channels; C % 16 == 0.
 * fold_acc holds t2s_wg_gate_fold_slots(B, C, L) slots of [B][8][L] floats: 2*ceil(C/128) with the 256-row tiles, 2*ceil(C/64) when
 * the shape takes 128-row tiles (a grid of 256-row tiles that would leave half of the CUs idle: short utterances at B = 1).
 * t2s_wg_end_fold_affine takes the same count as `nslots`. */
int t2s_wg_gate_fold_slots(int B, int C, int L);
int t2s_wg_in_cond_gate_fold(const void* A_hi, const void* A_lo, const float* bias, const void* X_hi, const void* X_lo,
                             const void* S_hi, const void* S_lo, void* acts_hi, void* acts_lo, const void* fold_A,
                             float* fold_acc, int fold_init, int B, int C, int n_cond, int taps, int dilation, int L,
                             int Lp, int halo, int Mpad, void* stream);
/* residual half only: x += W_res acts + b  (rows [0, C) of the packed res_skip weights; 128-row tiles) */
/* Composed conditioning for the inverse flow (weights packed once).  The ConvTranspose upsampler is linear and only ksize / stride
 * hops overlap, so cond_layers[i](upsample(mel)) at plane row t = P f + phi (P = stride / n_group) is (W_cond,i U_phi) applied to the
 * mel window [mel(f), mel(f-1), ...] of K2 = (ksize / stride) * n_mel values: the conditioning half of the gate GEMM shrinks from
 * n_mel * n_group to K2 columns (glow.py:159-162,215-221 / :252-258; 640 -> 320 at config.json defaults) and the upsampler goes away.
 *  t2s_wg_upsample_basis: U as conditioning planes [1][n_mel*n_group/32][Lp][32] whose time axis is the column (phi, k), plus a last
 *    column with the expanded bias; Lp = t2s_plane_rows(P * K2 + 1, halo).
 *  (caller: t2s_conv_bias_act with A = the conditioning K-chunks of the packed gate weights, X = U, f32 output [2C][ld])
 *  t2s_wg_compose_cond: that f32 result -> A2[phi][K2/32][Mpad][32] (hi, lo); bias_out = bias_in + its last column.
 *  t2s_wg_melwin_planes: mel [B][n_mel][frames] -> M[B][K2/32][Fp][32], row f = the window of frame f (zeros outside the clip).
 *  t2s_wg_in_melwin_gate_fold: t2s_wg_in_cond_gate_fold with (A2, M) in place of the conditioning weights and planes; always the
 *    256-row tiles (fold_acc: 2*ceil(C/128) slots); Fp >= ceil(L / P). */
int t2s_wg_upsample_basis(const float* W, const float* bias, int n_mel, int ksize, int stride, int n_group, int Lp, int halo,
                          void* U_hi, void* U_lo, void* stream);
int t2s_wg_compose_cond(const float* tmp, const float* bias_in, int rows, int Mpad, int P, int K2, long ld, void* A2_hi,
                        void* A2_lo, float* bias_out, void* stream);
int t2s_wg_melwin_planes(const float* mel, int B, int n_mel, int frames, int nlag, int Fp, void* M_hi, void* M_lo, void* stream);
int t2s_wg_in_melwin_gate_fold(const void* A_hi, const void* A_lo, const void* A2_hi, const void* A2_lo, const float* bias,
                               const void* X_hi, const void* X_lo, const void* M_hi, const void* M_lo, void* acts_hi,
                               void* acts_lo, const void* fold_A, float* fold_acc, int fold_init, int B, int C, int K2,
                               int taps, int dilation, int L, int Lp, int halo, int Mpad, int P, int Fp, void* stream);
/* pair8 = 1: the residual rows of A / bias were packed with perm = 2 (T2S_PERM_PAIR8: in every group of 32 channels packed row
 * 16 m + 4 q + e holds channel 8 q + 4 m + e), which lets the epilogue touch x in 16-byte pieces; C % 32 == 0 */
int t2s_wg_res_only(const void* A_hi, const void* A_lo, const float* bias, const void* acts_hi, const void* acts_lo,
                    void* X_hi, void* X_lo, int B, int C, int L, int Lp, int halo, int Mpad, int pair8, void* stream);
/* WN.end output from the folded accumulators + affine coupling (forward or reverse) */
int t2s_wg_end_fold_affine(const float* fold_acc, int nslots, const float* bes, int n_layers, const float* b_end,
                           float* z, float* log_s, float* wn_out, int B, int n_group, int c_off, int n_half, int L, int reverse,
                           void* stream);

/* Generic split-bf16 conv1d-as-GEMM with bias + activation epilogue:
 *   out[b][o][t] = act(bias[o] + sum_{tap,c} W[o][c][tap] * x[b][c][t + (tap - taps/2)*dil])
 * A planes packed with T2S_PERM_NONE.  Writes planes O_hi/O_lo (may be NULL) and/or out_f32 (may be NULL),
 * laid out [B][C][L], or [B][L][C] when f32_channel_last=1.  Used for the Tacotron-2 encoder / postnet convolutions and LSTM input projections
 * (reference tacotron.py:177-194, modules.py:94-137). */
int t2s_conv_bias_act(const void* A_hi, const void* A_lo, const float* bias, const void* X_hi, const void* X_lo,
                      void* O_hi, void* O_lo, float* out_f32, int f32_channel_last, int B, int Cin, int Cout, int taps,
                      int dilation, int act, int L, int Lp, int halo, int Mpad, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Tacotron-2 (reference tacotron/tacotron.py, tacotron/modules.py).  All f32, exact.
 */

/* y[item][row] = act(bias1[row] + bias2[row] + [W1[row] | W2[row]] . [x1 | x2 | x3](item)) * mask * mask_scale
 * One wave per output row (weights in registers), loop over items.  Replaces the small Linear layers:
 * query_layer (tacotron.py:137), linear_projection / gate_layer (:387-392), Prenet (modules.py:19-22),
 * memory_layer (tacotron.py:306).  K = n1+n2+n3 = k1+k2 <= 2560, every n a multiple of 4. */
int t2s_gemv(const float* W1, int ld1, int k1, const float* W2, int ld2, int k2, const float* x1, int n1, long sx1,
             const float* x2, int n2, long sx2, const float* x3, int n3, long sx3, const float* bias1,
             const float* bias2, float* y, long sy_item, long sy_row, int rows, int items, int act,
             const unsigned char* mask, long smask_item, float mask_scale, void* stream);

/* out[c][r] = in[r][c] */
int t2s_transpose(const float* in, float* out, int R, int C, void* stream);

/* embedding lookup -> planes: x[b][:, t] = emb[ids[b][t]]  (tacotron.py:40) */
int t2s_embed_planes(const long* ids, const float* emb, int B, int T, int E, int V, int Lp, int halo, void* X_hi,
                     void* X_lo, void* stream);
/* [B][C][L] f32 -> planes */
int t2s_f32_to_planes(const float* x, int B, int C, int L, int Lp, int halo, void* X_hi, void* X_lo, void* stream);
/* Tacotron.parse_output (reference tacotron.py:67-76): mel[b][:, t] = mel_post[b][:, t] = 0 and gate[b][t] = 1e3 for
 * t >= lengths[b], in place.  mel, mel_post: [B][n_mel][T] f32; gate: [B][T] f32; lengths: [B] int32 in device memory.
 * The reference does this with .data.masked_fill_ AFTER the postnet has run, i.e. on the very tensor the postnet's first
 * convolution saved for its backward pass: its weight gradient sees the masked mel (the training path re-derives that
 * convolution's saved input planes from the masked tensor, text2speech_amd/tacotron/tacotron.py). */
int t2s_taco_parse_output(float* mel, float* mel_post, float* gate, const int* lengths, int B, int n_mel, int T, void* stream);
/* eval BatchNorm folded into the preceding conv: scale = gamma/sqrt(var+eps), bias' = (bias-mean)*scale+beta */
int t2s_bn_fold(const float* gamma, const float* beta, const float* mean, const float* var, const float* conv_bias,
                float eps, int C, float* scale, float* bias_out, void* stream);

/* Training-mode BatchNorm1d on a conv output x[B][C][T]: batch mean / biased variance per channel (written to
 * mean/var), affine, activation, optional {0,1} dropout mask * mask_scale; result as planes and/or f32 [B][C][T]
 * (reference tacotron.py:183-184,193-194; modules.py:105-137 in .train() mode) */
int t2s_bn_train(const float* x, const float* gamma, const float* beta, float eps, int act, const unsigned char* mask,
                 float mask_scale, int B, int C, int T, int Lp, int halo, float* mean, float* var, void* O_hi, void* O_lo,
                 float* out_f32, void* stream);

/* nn.BatchNorm1d's bookkeeping in .train() mode after t2s_bn_train: running_x = (1 - momentum) running_x + momentum batch_x, the
 * variance unbiased by n / (n - 1) (n = B * T); *num_batches_tracked (int64, optional) += 1.  One launch per layer. */
int t2s_bn_running_update(const float* mean, const float* var, float* running_mean, float* running_var,
                          long long* num_batches_tracked, float momentum, long long n, int C, void* stream);
/* clears `bytes` bytes at p (16-byte aligned): the single fill of a training step's accumulator arena */
int t2s_zero_fill(void* p, size_t bytes, void* stream);

/* Encoder BiLSTM recurrence with packed-sequence semantics (tacotron.py:199-207).  gx[B][T][8H] = W_ih x + b_ih + b_hh
 * for both directions (fwd gates then reverse gates), whhT_* = W_hh^T [H][4H]; out[B][T_out][2H]; 4H must be 1024. */
int t2s_taco_encoder_lstm(const float* gx, const float* whhT_fwd, const float* whhT_rev, const int* lengths, float* out,
                          int B, int T, int H, int T_out, float* gates_save /* [B][T][2][4H] or NULL */,
                          float* c_save /* [B][T][2][H] or NULL */, void* stream);
/* ABI v4.  The same recurrence with W_hh resident on the chip: four workgroups per (batch element, direction), each with a quarter
 * of the matrix in registers for the whole sequence, exchange their 64 new h values per step through tagged 8-byte granules in
 * `xbuf` (t2s_taco_lstm_xbuf_bytes(B) bytes, caller-owned, ZERO before its first use, not shared by launches that may overlap in
 * time; the last 8 bytes are an error word the kernel raises if a bounded wait expires - a caller that wants to know reads it
 * after synchronising).  `epoch` is the caller's launch counter for this buffer (any value that differs from the previous launches'
 * in its low 20 bits): tags of an earlier launch never match.  Same results as t2s_taco_encoder_lstm up to summation order.
 * H must be 256, T < 4095. */
int t2s_taco_encoder_lstm_split(const float* gx, const float* whhT_fwd, const float* whhT_rev, const int* lengths, float* out,
                                int B, int T, int H, int T_out, float* gates_save, float* c_save, void* xbuf, unsigned epoch,
                                void* stream);
long t2s_taco_lstm_xbuf_bytes(int B);
/* BPTT of that recurrence: d_out[B][T_out][2H] -> dgx[B][T][8H] (zero beyond each length), hprev[B][T][2H] (the h each
 * step consumed; the X operand of the W_hh weight-gradient GEMM); whh_* are the natural [4H][H] matrices */
int t2s_taco_encoder_lstm_bwd(const float* d_out, const float* out, const float* gates_save, const float* c_save,
                              const float* whh_fwd, const float* whh_rev, const int* lengths, float* dgx, float* hprev,
                              int B, int T, int H, int T_out, void* stream);
/* ABI v4.  ... and its BPTT in the same form (xbuf / epoch as above; a buffer of its own if a forward may be in flight). */
int t2s_taco_encoder_lstm_bwd_split(const float* d_out, const float* out, const float* gates_save, const float* c_save,
                                    const float* whh_fwd, const float* whh_rev, const int* lengths, float* dgx, float* hprev,
                                    int B, int T, int H, int T_out, void* xbuf, unsigned epoch, void* stream);
/* f32 channel-last rows x[b][t][c] -> planes ; embedding gradient d_emb[v][e] = sum over (b,t) with ids == v of planes */
int t2s_rows_to_planes(const float* x, int B, int T, int C, int Lp, int halo, void* X_hi, void* X_lo, void* stream);
int t2s_embedding_grad(const long* ids, const void* D_hi, const void* D_lo, int B, int T, int E, int V, int Lp, int halo,
                       float* d_emb, void* stream);

/* Bernoulli(keep_prob) bytes (0/1) from a counter hash: dropout masks (the always-on prenet dropout,
 * modules.py:21, and the training-mode dropouts) when the caller does not inject them */
int t2s_bernoulli_mask(unsigned char* mask, size_t n, unsigned long long seed, unsigned long long offset, float keep_prob,
                       void* stream);

/* Decoder state + weights for t2s_taco_decode_steps.  Pointers are device pointers; f32 unless noted. */
typedef struct t2s_taco_decoder {
    int B, T_in, n_mel, prenet_dim, enc_dim, att_rnn_dim, dec_rnn_dim, att_dim, loc_filters, loc_kernel;
    int T_cap;           /* time capacity of mel_gate_out / align_out */
    int teacher_forced;  /* 1: step input = pre_all[step]; projection hoisted (hc_all); 0: autoregressive */
    int mask_steps;      /* autoregressive: number of [B][2][prenet] mask slices in prenet_masks */
    const float *att_w_ih, *att_w_hh, *att_b_ih, *att_b_hh;   /* attention_rnn (tacotron.py:366) */
    const float *dec_w_ih, *dec_w_hh, *dec_b_ih, *dec_b_hh;   /* decoder_rnn   (tacotron.py:380) */
    const float *w_query, *w_loc_conv, *w_loc_dense, *w_v;    /* attention_layer (tacotron.py:96-143) */
    const float *w_proj, *b_proj;        /* [n_mel+1][dec+enc]: linear_projection rows, then the gate_layer row */
    const float *w_projpre, *b_projpre;  /* [prenet][dec+enc] = W_prenet0 . W_proj[:n_mel], W_prenet0 . b_proj[:n_mel];
                                          * stored DIRECTLY BEHIND w_proj / b_proj (one [n_mel+1+prenet] row block) */
    const float *w_loc_denseT;           /* [loc_filters][att_dim] (transpose of w_loc_dense) */
    const float *w_pre2;                 /* [prenet][prenet] = prenet.layers.1 */
    const float *memory, *pmem;          /* [B][T_in][enc], [B][T_in][att_dim] */
    const int *mem_lengths;              /* [B] or NULL */
    const float *pre_all;                /* teacher forced: [T+1][B][prenet] */
    const unsigned char *prenet_masks;   /* autoregressive: [mask_steps][B][2][prenet] 0/1, slice s feeds step s */
    const unsigned char *att_drop, *dec_drop; /* training dropout on the LSTM outputs: [T][B][H] 0/1, or NULL */
    float att_drop_scale, dec_drop_scale;
    float *att_h0, *att_h1, *att_c, *dec_h0, *dec_h1, *dec_c;   /* ping-pong h (step parity), c in place */
    float *att_w, *att_wcum, *ctx, *q, *energies, *pre1, *pre2; /* [B][T_in] x3, [B][enc], [B][att_dim], ... */
    float *q_part;                       /* [att_rnn/2][B][att_dim] scratch: per-workgroup partial queries written by the attention
                                          * cell and summed by the fused attention kernel (B <= 8), or NULL (W_query . h there) */
    float *mel_gate_out;                 /* [B][n_mel+1][T_cap] (autoregressive), row n_mel = gate logit */
    float *align_out;                    /* [B][T_cap][T_in] */
    float *hc_all;                       /* teacher forced: [T][B][dec+enc] */
    /* training saves (teacher forced; all NULL otherwise): per step t */
    float *att_gates_all, *att_c_all;    /* [T][B][4H], [T][B][H] attention-LSTM gates (post-activation i,f,g,o) / cell state */
    float *dec_gates_all, *dec_c_all;    /* same for the decoder LSTM */
    float *att_h_all;                    /* [T][B][H] attention-LSTM output after dropout */
    float *q_all, *wcum_all;             /* [T][B][att_dim] queries, [T][B][T_in] cumulative weights after the step */
    /* ABI v4.  [3][B][4H] scratch, ZERO before step 0, or NULL.  Autoregressive decode at B <= 4 (T2S_DECODE_STREAM_MAXB) with
     * att_rnn_dim = dec_rnn_dim = 1024: the fused attention launch of step t also streams W_hh_dec . h_dec(t-1), W_ih_dec[:, :att_rnn] . h_att(t) and
     * W_hh_att . h_att(t) on the CUs the attention leaves idle and keeps the products here; the decoder cell of step t and the
     * attention cell of step t+1 then add them instead of streaming those 50 MB themselves.  State like h / c: carried between
     * calls that continue one utterance.  NULL (or T2S_DECODE_STREAM=0): every cell streams its own weights (ABI v3 behaviour). */
    float *gate_part;
    /* ABI v4.  [prenet][prenet] TRANSPOSE of w_pre2, or NULL.  With gate_part: the prenet's second layer (modules.py:19-22) is
     * folded into the attention cell's launch - every workgroup recomputes its 256 outputs from pre1, as a sparse product (pre1 is
     * ~3/4 exact zeros after ReLU and dropout: one coalesced row of the transpose per nonzero) - instead of a GEMV launch of its own
     * on the serial chain.  NULL (or T2S_DECODE_FOLD_PRE2=0): the separate launch. */
    const float *w_pre2T;
    /* ABI v4.  [B][T_in][att_dim] scratch, ZERO before step 0, or NULL.  With gate_part: the location term of the attention
     * (location_dense o location_conv of the current weights / cumulative weights, tacotron.py:96-107) is computed one launch EARLIER,
     * by extra workgroups of the projection launch of the previous step (which leaves ~170 CUs idle), and the fused attention launch
     * only adds the query and the processed memory to it: the two exact-f32 matrix-core stages leave the one workgroup the whole
     * launch waits for (4 us at 128 encoder positions, growing linearly with T_in).  State like gate_part. */
    float *ploc;
    /* ABI v4.  [16][B][4 * dec_rnn] scratch (not state: written before it is read inside one call), or NULL.  Teacher forced at 9+ items
     * with att_h_all: the decoder cells run a chunk of 16 steps behind the attention chain (helper stream), so the input half of
     * their pre-activations, W_ih_dec . [h_att(s) | ctx(s)], is ONE matrix product per chunk over 16 x B items - its 25 MB of
     * weights read once per 16 steps instead of once per step - and the per-step cell streams only W_hh_dec (17 of 42 MB).
     * Used only with T2S_DECODE_CHUNK_GEMM=1 (measured no faster); otherwise, or NULL: every decoder
     * cell streams all of [W_ih | W_hh]. */
    float *dec_in_part;
    /* ABI v4.  (B * T_in + 1) 8-byte words, ZERO before step 0 of a sequence (state like gate_part: tags are step numbers), or NULL.
     * At 9+ items with attention_dim 128, 32 location filters, T_in <= 512 and enc_dim a multiple of 64, the
     * energies launch also does softmax, cumulative weights and context: the workgroups of a batch element exchange their energies
     * through tagged granules here instead of ending the launch.  The last word is raised if a bounded wait expires.  A buffer
     * serves every step INDEX once per zeroing (the tag of step s is s + 1): callers that revisit step indices pass NULL.
     * NULL (or T2S_ATT_ONE_LAUNCH=0): energies and softmax + context stay two launches. */
    void *att_xbuf;
    /* ABI v4.  16 bytes (a step counter and an error word), ZERO before step 0 of a sequence, or NULL.  Teacher forced at 9+ items with
     * att_h_all: instead of running a chunk of 16 steps behind, the helper stream's decoder cell of step s - 1 is released by a word
     * the attention cell's launch of step s stores as it starts, so that it runs beside the attention launch of step s (whose
     * workgroups share a CU with it) and is over when the next attention cell needs the chip.  NULL (or T2S_DECODE_PACED=0): chunks. */
    void *pace_flag;
} t2s_taco_decoder;

/* Enqueue decoder steps [step0, step0+n_steps) (Decoder.decode, tacotron.py:355-393, plus in autoregressive mode
 * the projection and the prenet of the next step, tacotron.py:447-461) on `stream` without host synchronisation.
 * Teacher-forced with att_h_all and hc_all given (training): the decoder LSTM cells - which feed nothing but the next decoder cell
 * and the projection after the loop - are enqueued on a library-owned helper stream, T2S_DECODE_CHUNK (16) steps behind the
 * attention chain, reading h_att / ctx from those saves; `stream` has been made to wait for the helper when the call returns
 * (T2S_DECODE_SPLIT=0: everything on `stream`). */
int t2s_taco_decode_steps(const t2s_taco_decoder* d, int step0, int n_steps, void* stream);

/* ABI v4.  One call of the location-sensitive attention alone (Attention.forward, tacotron.py:145-166, with the state update of
 * Decoder.decode around it, tacotron.py:371-379): query = w_query . h_att, energies from the location features of (w, w_cum) and
 * the processed memory, masked softmax over T, context.  IN PLACE: w [B][T] holds the previous weights on entry and the new ones on
 * return, w_cum [B][T] is incremented by them, ctx [B][enc] receives the context.  q_scratch [B][att_dim], e_scratch [B][T].
 * lengths [B] int32 or NULL.  Same kernels as t2s_taco_decode_steps (one fused launch up to T2S_ATT_FUSED_MAXB items, else
 * query GEMV + energies + softmax/context). */
int t2s_taco_attention(const float* h_att, const float* memory, const float* pmem, const int* lengths, float* w, float* w_cum,
                       float* ctx, float* q_scratch, float* e_scratch, const float* w_query, const float* w_loc_conv,
                       const float* w_loc_dense, const float* w_loc_denseT, const float* w_v, int B, int T, int att_rnn,
                       int att_dim, int enc_dim, int loc_filters, int loc_kernel, void* stream);

/* stop_step[b] = first step in [step0, step0+n) with sigmoid(gate) > threshold, if still -1 (tacotron.py:455) */
int t2s_taco_stop_check(const float* mel_gate_out, int B, int n_mel, int T_cap, int step0, int n, float threshold,
                        int* stop_step, void* stream);

/* ------------------------------------------------------------------------------------------------
 * WaveGlow training step (reference waveglow/train.py:110-124: forward -> WaveGlowLoss -> backward -> Adam;
 * the reference gets its backward from autograd over glow.py:207-249, these are the hand-written equivalents).
 * "time-major planes": tm[b][t/32][row][t%32] bf16 (hi, lo) - the operands of the weight-gradient GEMMs,
 * whose contraction index is time.
 */

/* training-mode forms of the two WN-layer entry points: also save the sigmoid values (G planes; the tanh planes T are optional,
 * pass NULL: the backward pass rebuilds tanh as acts / sigmoid) for the backward pass, and read the residual from R planes so
 * every layer's input stays resident */
int t2s_wg_in_cond_gate_train(const void* A_hi, const void* A_lo, const float* bias, const void* X_hi, const void* X_lo,
                              const void* S_hi, const void* S_lo, void* acts_hi, void* acts_lo, void* T_hi, void* T_lo,
                              void* G_hi, void* G_lo, int B, int C, int n_cond, int taps, int dilation, int L, int Lp,
                              int halo, int Mpad, void* stream);
int t2s_wg_res_skip_train(const void* A_hi, const void* A_lo, const float* bias, const void* acts_hi,
                          const void* acts_lo, const void* R_hi, const void* R_lo, void* X_hi, void* X_lo, float* skip,
                          int B, int C, int n_res, int skip_init, int L, int Lp, int halo, int Mpad, void* stream);

/* Training forward on the no-grad forward's kernels (round 3).  The skip path only ever feeds WN.end (glow.py:172-175), so the
 * training forward folds it into the gate GEMM's epilogue exactly as the no-grad forward does (t2s_wg_in_cond_gate_fold) and
 * never forms the skip sum; the backward pass, which needs it once per flow for WN.end's weight gradient, rebuilds it with
 * t2s_wg_skip_sum from the saved gate outputs.  What training adds to the forward is only what it saves:
 *   t2s_wg_in_cond_gate_fold_train = t2s_wg_in_cond_gate_fold + the sigmoid planes G (tanh is rebuilt as acts / G) + act_bchunks:
 *     32-channel chunks between batch entries of the acts / G planes (0 = C/32), so that the gate outputs of all layers of a flow
 *     can sit side by side in one plane set;
 *   t2s_wg_res_only_train = t2s_wg_res_only reading the layer input from R and writing the next layer's input to X (both are kept);
 *   t2s_wg_end_fold_affine(wn_out != NULL) also stores (b ; log_s) for the coupling's backward.
 * t2s_wg_skip_sum: skip[b][c][t] = bias[c] + sum_k A[k][c] * acts[b][k][t] over n_k_chunks * 32 K channels - with A the skip rows
 * of a flow's res_skip weights packed one after the other along K (t2s_pack_conv_weight, koff = layer * C) and acts the flow-wide
 * planes, this is sum_i (W_skip,i acts_i + b_skip,i): f32 planes [B][C/32][Lp][32] as t2s_wg_res_skip_train writes them. */
int t2s_wg_in_cond_gate_fold_train(const void* A_hi, const void* A_lo, const float* bias, const void* X_hi, const void* X_lo,
                                   const void* S_hi, const void* S_lo, void* acts_hi, void* acts_lo, void* G_hi, void* G_lo,
                                   int act_bchunks, const void* fold_A, float* fold_acc, int fold_init, int B, int C, int n_cond,
                                   int taps, int dilation, int L, int Lp, int halo, int Mpad, void* stream);
int t2s_wg_res_only_train(const void* A_hi, const void* A_lo, const float* bias, const void* acts_hi, const void* acts_lo,
                          int act_bchunks, const void* R_hi, const void* R_lo, void* X_hi, void* X_lo, int B, int C, int L, int Lp,
                          int halo, int Mpad, int pair8, void* stream);
int t2s_wg_skip_sum(const void* A_hi, const void* A_lo, const float* bias, const void* acts_hi, const void* acts_lo,
                    int n_k_chunks, int act_bchunks, float* skip, int B, int C, int L, int Lp, int halo, int Mpad, void* stream);

/* d_pre = gate'(acts, G) * (W_rs^T [d_x ; d_skip]):  data gradient of res_skip_layers[i] fused with the backward of
 * tanh*sigmoid (glow.py:33-40,164), from the layer's saved gate output acts = tanh * sigmoid and G = sigmoid (tanh = acts / G).
 * A = t2s_pack_transposed(W_rs); DX may be NULL (last layer: skip rows only).
 * DP planes have 2C channels (tanh half, then sigmoid half).  dp_bchunks: 32-channel chunks between batch entries of the DP
 * planes (0 = 2C/32): DP may be a slice of a wider plane set - the training path keeps the d_pre of all layers of a flow side
 * by side so that the conditioning gradient is ONE K-concatenated GEMM per flow.  tg_bchunks: the same for the acts / G planes
 * (0 = C/32): the training forward keeps the gate outputs of a flow's layers side by side too (t2s_wg_skip_sum). */
int t2s_wg_bwd_gate_dgrad(const void* A_hi, const void* A_lo, const float* zero_bias, const void* DX_hi,
                          const void* DX_lo, const void* DS_hi, const void* DS_lo, const void* acts_hi, const void* acts_lo,
                          const void* G_hi, const void* G_lo, int tg_bchunks, void* DP_hi, void* DP_lo, int dp_bchunks, int B, int C,
                          int L, int Lp, int halo, int Mpad, int pair8, void* stream);

/* O (+)= conv(X) with packed (transposed) weights: data gradients of in_layers[i] (dilated, taps mirrored) and
 * cond_layers[i].  init=1 stores, init=0 accumulates into the O planes.  x_bchunks: chunks between batch entries of the X
 * planes (0 = Cin/32; X may be a slice of a wider plane set, see t2s_wg_bwd_gate_dgrad). */
int t2s_conv_accumulate(const void* A_hi, const void* A_lo, const float* zero_bias, const void* X_hi, const void* X_lo,
                        int x_bchunks, void* O_hi, void* O_lo, int B, int Cin, int Cout, int taps, int dilation, int init, int L, int Lp,
                        int halo, int Mpad, int pair8, void* stream);

/* out[b*ksplit + s][m][n] = sum over time chunks [k0,k1) (split s of ksplit) of A_tm[b][t][m] * X_tm[b][t][n]:
 * B*ksplit split-K slabs; chunks outside [k0,k1) (the zero halo) are skipped */
int t2s_wgrad_gemm(const void* A_hi, const void* A_lo, const void* X_hi, const void* X_lo, const float* zero_bias,
                   float* out, int B, int M, int N, int Mpad, int Npad, int n_tchunks, int k0, int k1, int ksplit,
                   void* stream);
/* Same contraction with K flattened over (batch element, time chunk): out = [nsplit][M][N] slabs, slab s covering
 * ceil(B*(k1-k0)/nsplit) consecutive flattened K-steps, so the split count can be chosen to fill the chip in one round
 * (nsplit ~ 256 / (ceil(M/256)*ceil(N/256))) instead of being a multiple of B. */
int t2s_wgrad_gemm_flat(const void* A_hi, const void* A_lo, const void* X_hi, const void* X_lo, const float* zero_bias,
                        float* out, int B, int M, int N, int Mpad, int Npad, int n_tchunks, int k0, int k1, int nsplit,
                        void* stream);

/* The same contraction straight from CHANNEL-LAST planes (no time-major copies): the transpose happens in the LDS read
 * (ds_read_b64_tr_b16).  Operands are lists of 32-channel chunks, 8 per 256-row tile (pad with a chunk of zeros):
 *   hi / lo  = device pointers to plane row 0 of that chunk for batch entry 0 (a dilated tap is a row offset folded into the
 *              pointer: the autograd of in_layers[i], glow.py:134-139, needs x shifted by (tap - 1) * dilation);
 *   bstride  = u16 elements between batch entries (0 for constants such as the all-ones bias chunk).
 * out = [nsplit][M][ldp] f32 slabs (ldp >= N floats per row; columns N .. ldp-1 are scratch) over K-blocks [k0, k1) of 32 plane
 * rows of each of the B batch entries, as t2s_wgrad_gemm_flat.  ldp % 4 == 0 (and out 16-byte aligned) selects the ping-pong
 * kernel, whose epilogue stores 16-byte pieces; any other ldp the round-2 lockstep kernel (also: env T2S_WGRAD_PP=0).
 * Every K-block is a WHOLE block of 32 plane rows starting at row 32 k (+ the shift folded into the pointer): the caller makes
 * sure rows [32 k0 - max shift, 32 k1 + max shift) exist in every plane (halo % 32 == 0 does, text2speech_amd/glow.py geom()).
 * bias_cols != 0 (ping-pong kernel only, ldp >= N + 4): columns N .. N+3 of every slab also receive four partial sums of the
 * M-side operand's rows over the slab's K range - the bias gradient of a convolution, db[m] = sum_t d_out[m][t], without an
 * all-ones column in the N-side table (t2s_wn_backward adds them: n_bias_cols = 4).
 * Both tables live in device memory ([n_tiles * 8] entries).  Replaces 7 t2s_plane_transpose launches per WN layer. */
typedef struct t2s_wgrad_chunk {
    const void* hi;
    const void* lo;
    long bstride;
} t2s_wgrad_chunk;
int t2s_wgrad_cl(const t2s_wgrad_chunk* a_chunks, int n_a_chunks, const t2s_wgrad_chunk* b_chunks, int n_b_chunks, float* out,
                 int B, int M, int N, int ldp, int k0, int k1, int nsplit, int bias_cols, void* stream);

int t2s_plane_transpose(const void* src_hi, const void* src_lo, int B, int src_chunks, int n_chunks, int Lp, int shift,
                        void* dst_hi, void* dst_lo, int Npad, int n_off, void* stream);
int t2s_tm_ones_row(void* dst_hi, void* dst_lo, int B, int Lp, int halo, int L, int Npad, int n_row, void* stream);
/* A[c][koff + tap'*O_pad + o] = scale[o] * v[o][c][flip ? Kt-1-tap' : tap'] -> (hi, lo) [k/32][Mpad][32] */
int t2s_pack_transposed(const float* v, const float* scale, int O, int Cin, int Kt, int flip, int O_pad, int Mpad,
                        int koff, void* A_hi, void* A_lo, int pair8, void* stream);
/* ABI v4: pair8 (t2s_pack_transposed, t2s_wg_bwd_gate_dgrad, t2s_conv_accumulate).  pair8 = 1 packs the M rows of the transposed
 * operand in PERM_PAIR8 order (within every 32 rows, packed row 16 m + 4 q + e = row 8 q + 4 m + e) and tells the two backward GEMMs
 * that their A operand is packed so: their epilogues then move whole 16-byte pieces per plane (8 consecutive channels per lane)
 * instead of 8-byte ones.  Only the 256-row ping-pong kernels have that epilogue: ask t2s_wg_bwd_pair8_ok(B, rows, L) (rows = Cout of
 * t2s_conv_accumulate / C of t2s_wg_bwd_gate_dgrad) before packing; a GEMM call with pair8 = 1 on a shape that takes the 128-row
 * kernels returns T2S_EINVAL.  T2S_BWD_PAIR8=0: t2s_wg_bwd_pair8_ok answers 0. */
int t2s_wg_bwd_pair8_ok(int B, int rows, int L);
/* per-row scale g/|v| of a weight-normed conv (what the forward pack applied), for t2s_pack_transposed */
int t2s_weightnorm_scale(const float* v, const float* g, int O, int K, float* scale, void* stream);
/* reduce split-K slabs P[nsplit][Prows][Pcols] and apply weight_norm's backward (g NULL: plain weight); the bias gradient is
 * the sum over slabs of columns col_bias .. col_bias + n_bias_cols - 1 (n_bias_cols >= 1) */
int t2s_wn_backward(const float* P, int nsplit, int Prows, int Pcols, int row_off, int col_off, int tap_stride,
                    int col_bias, int n_bias_cols, const float* v, const float* g, int O, int Cin, int Kt, float* dv, float* dg,
                    float* db, int db_accum, void* stream);
/* affine coupling backward + un-apply (glow.py:241-246); wn_out = (b ; log_s) [B][2nh][L], d_out gets (d_b ; d_log_s).
 * g_log_s = upstream gradient of this flow's log_s output: [B][nh][L], or - g_log_s_scalar != 0 - ONE float that stands for
 * every element (WaveGlowLoss's gradient is the constant -1/N broadcast, glow.py:52-58), or NULL for none. */
int t2s_wg_affine_backward(float* z, float* dz, const float* wn_out, const float* g_log_s, int g_log_s_scalar, float* d_out, int B,
                           int n_group, int c_off, int n_half, int L, void* stream);
/* out[r][j] (or [j][r]) = sum_{b,t} P[b][r][t] * Q[b][q_off+j][t], rowsum[r] = sum P; P = planes (hi/lo, or f32).
 * scratch: t2s_small_wgrad_scratch(B, chunks) floats of partial sums (reduced in a fixed order: deterministic). */
long t2s_small_wgrad_scratch(int B, int chunks);
int t2s_small_wgrad(const void* P_hi, const void* P_lo, const float* P_f32, const float* Q, float* out, float* rowsum,
                    float* scratch, int B, int chunks, int Lp, int halo, int L, int R, int J, int Jtot, int q_off, int out_transposed,
                    void* stream);
int t2s_rows_sum(const float* Q, int B, int Jtot, int q_off, int J, int L, float* out, void* stream);
/* d_z[:, c_off:c_off+n_half] += W_start^T d_x */
int t2s_wg_start_dgrad(const void* X_hi, const void* X_lo, const float* w, float* dz, int B, int n_group, int c_off,
                       int n_half, int C, int L, int Lp, int halo, void* stream);
/* dW = d_out . z_in^T + (*gscale_ptr * gmul) * W^-T   (glow.py:100-101) */
int t2s_wg_convinv_wgrad(const float* dz, const float* zin, const float* Winv, const float* gscale_ptr, float gmul,
                         int B, int n_group, int c_off, int n, int L, float* dW, void* stream);
/* ConvTranspose1d weight / bias gradient from the conditioning-plane gradient (glow.py:215-221) */
int t2s_wg_upsample_wgrad(const void* D_hi, const void* D_lo, const float* mel, int B, int n_mel, int frames, int ksize,
                          int stride, int n_group, int L, int Lp, int halo, float* dW, float* db, void* stream);

/* Adam over every parameter in one launch (torch.optim.Adam semantics; waveglow/train.py:79,124).  `jobs` is a
 * DEVICE array; job i owns blocks [blk_start, blk_start + ceil(n/1024)).  gscale multiplies the gradient first. */
typedef struct t2s_adam_job {
    float* p; const float* g; float* m; float* v;
    long n;
    long blk_start;
} t2s_adam_job;
int t2s_adam_table(const t2s_adam_job* jobs, int n_jobs, long total_blocks, float lr, float beta1, float beta2,
                   float eps, int step, float gscale, float weight_decay, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Tacotron-2 training step, backward (reference: autograd over tacotron/tacotron.py:36-49,355-429 and
 * tacotron/modules.py:19-22,94-137; loop train.py:219-225).  Weight gradients of every Linear / LSTM matrix contract
 * over (step, batch) items and reuse t2s_wgrad_gemm on time-major planes built by t2s_rows_to_tm.
 */
/* x[item - shift][c] (f32 rows, stride ld) -> time-major planes tm[item/32][n_off + c][item%32]; items_pad % 32 == 0 */
int t2s_rows_to_tm(const float* x, long ld, int items, int items_pad, int shift, int C, void* dst_hi, void* dst_lo,
                   int Npad, int n_off, void* stream);
/* the same for nb sets at once: set z reads x + z * x_bstride (floats) and writes its planes at dst + z * dst_bstride (bf16
 * elements) - e.g. one set per batch element in front of a t2s_wgrad_gemm with B = nb */
int t2s_rows_to_tm_batched(const float* x, long ld, long x_bstride, int items, int items_pad, int shift, int C, void* dst_hi,
                           void* dst_lo, long dst_bstride, int Npad, int n_off, int nb, void* stream);
/* LSTMCell backward, pointwise part: dh = (dh1+dh2+dh3)*dropout -> dgates[B][4H] (i,f,g,o), dc_carry updated in place */
int t2s_lstm_cell_bwd(const float* dh1, long s1, const float* dh2, long s2, const float* dh3, long s3,
                      const unsigned char* drop_mask, float drop_scale, const float* gates, const float* c_new,
                      const float* c_prev, float* dc_carry, float* dgates, int B, int H, void* stream);
/* dz = (y > 0) ? scale*dy : 0  (backward of relu followed by dropout-with-scale, e.g. the prenet) */
int t2s_relu_drop_bwd(const float* dy, const float* y, float scale, size_t n, float* dz, void* stream);

typedef struct t2s_att_bwd {
    const float *dctx1; long sc1; const float *dctx2; long sc2; const float *dctx3; long sc3;
    const float *w_cur; long s_wcur;
    const float *w_prev, *wc_prev; long s_wprev, s_wcprev;
    const float *q, *pmem, *memory; const int *lengths;
    const float *w_loc_conv, *w_loc_dense, *w_v;
    float *dw_carry, *dwc_carry, *d_q, *d_pmem, *d_memory;
    float *dD_part, *dK_part, *dv_part;   /* += ; one slot per (batch element, 32-position chunk): [B*ceil(T/32)][...];
                                             dD slots hold the transposed gradient [loc_f][att_dim] */
    float *dw_buf, *df_buf, *dq_part;     /* scratch: [B][T], [B][T][32], [B][ceil(T/32)][att_dim] */
    float *dctx_out;                      /* optional [B][enc]: d_ctx of this step; with it d_memory may be NULL and the caller
                                           * forms d_memory = sum_t w_t (x) d_ctx_t once after the loop */
    int B, T, att_dim, enc_dim, loc_f, loc_ks;
    /* the one-launch form (all three set; needs dctx_out, d_memory NULL, attention_dim 128, 32 filters, kernel <= 31; d_q is then
     * NOT written - sum dq_part over the chunks): the step's saved context [B][.] (row stride s_ctx) and
     * the carry buffers the step WRITES (it reads dw_carry / dwc_carry); all four are then [3][B][T] (see t2s_taco_bptt) */
    const float *ctx; long s_ctx;
    float *dw_carry_out, *dwc_carry_out;
} t2s_att_bwd;
/* one decoder step of the location-sensitive attention, backward (tacotron.py:124-166,379) */
int t2s_taco_att_bwd(const t2s_att_bwd* a, void* stream);

/* Reversed decoder loop (BPTT through tacotron.py:355-393,418-427): steps t_hi-1 ... t_lo, newest first.  All buffers are the
 * caller's; histories are [T_out][B][...] unless noted.  W_dT = [W_ih | W_hh]^T of decoder_rnn ([A+E+D][4D]), W_aT the same
 * for attention_rnn ([P+E+A][4A]), w_query = query_layer weight as stored ([att_dim][A]).  out_d / out_a receive d[input | h] of the two
 * cells ([T_out][B][A+E+D] and [T_out][B][P+E+A]); dg_d / dg_a / dq_all keep every step's gate / query gradients for the
 * weight-gradient GEMMs that follow the loop. */
typedef struct t2s_taco_bptt {
    int B, T_in, T_out, T_cap;
    int prenet_dim, enc_dim, att_rnn_dim, dec_rnn_dim, att_dim, loc_filters, loc_kernel;
    const float *W_dT, *W_aT, *w_query, *w_loc_conv, *w_loc_dense, *w_v;
    const float *dec_gates_all, *dec_c_all, *att_gates_all, *att_c_all, *q_all, *wcum_all;
    const float *align;                    /* [B][T_cap][T_in] */
    const float *pmem, *memory; const int *lengths;
    const unsigned char *att_drop, *dec_drop; float att_drop_scale, dec_drop_scale;
    const float *d_hc;                     /* [T_out][B][D+E]: gradient of [h_dec | ctx] from projection + gate */
    float *out_d, *out_a, *dg_d, *dg_a, *dq_all;
    float *dc_d, *dc_a, *dw_c, *dwc_c;                  /* carries: [B][D], [B][A], [B][T_in] x2 (zero-initialised) */
    float *d_pmem, *d_memory;                           /* += [B][T_in][att_dim], [B][T_in][enc] */
    float *dD_part, *dK_part, *dv_part, *dw_buf, *df_buf, *dq_part;   /* as in t2s_att_bwd */
    float *dctx_all;                                    /* optional [T_out][B][enc]: every step's d_ctx; then d_memory is not
                                                         * touched by the loop (deferred, see t2s_att_bwd.dctx_out) */
    /* optional, all or none (needs dctx_all, attention_dim 128, 32 location filters, kernel <= 31): the attention backward of a
     * step in ONE launch.  ctx_all = the forward's contexts, step t / item b at ctx_all + t * s_ctx_step + b * s_ctx_item;
     * dw_c2 / dwc_c2 = a second pair of carry buffers, zero-initialised like dw_c / dwc_c (step t reads the pair of parity
     * t & 1 - dw_c for even t - and writes the other).  With these, all four carry buffers are [3][B][T_in]: slot 0 the part
     * a 32-position chunk computes for itself, slots 1 / 2 the parts reaching in from the chunk to the right / left. */
    const float *ctx_all; long s_ctx_step, s_ctx_item;
    float *dw_c2, *dwc_c2;
    /* ABI v4.  (B * ceil(T_in / 32) * 128 + 3) 8-byte words (the last three: error word, pace word, pace error word), ZERO before a BPTT pass, or NULL.  With the one-launch attention backward
     * (ctx_all ...) and T_in <= 512: the attention LSTMCell's pointwise backward (with W_query^T d_q) runs inside that launch - the
     * chunk workgroups of a batch element exchange their partial d_q through tagged granules here (tag = step + 1), each then takes its
     * share of the hidden units.  The last word is raised if a bounded wait expires.  NULL (or T2S_BPTT_FOLD_CELL=0): a launch of its own. */
    void *att_xbuf;
} t2s_taco_bptt;
/* Launch sequencing only.  Unless T2S_BPTT_ONE_STREAM is set, the decoder-cell chain and the location-conv part of the
 * attention backward run on two library-owned non-blocking streams (created on first use on the current device, kept for the
 * life of the process), ordered against `stream` with events; everything this call enqueues is complete, as seen from
 * `stream`, once the call's last wait has been passed.  Not re-entrant across host threads for the same device. */
int t2s_taco_bptt_steps(const t2s_taco_bptt* p, int t_hi, int t_lo, void* stream);

/* Tacotron2Loss (tacotron/loss_function.py:3-18): out[0] = mean((mel-target)^2) + mean((post-target)^2) + mean(BCEWithLogits(gate,
 * gate_target)), out[1] = the two mel terms, out[2] = the gate term; d_mel / d_post / d_gate (each optional) receive the
 * gradients for an upstream gradient of 1.  partial = scratch of 256*3 doubles (two-stage, order-independent reduction). */
int t2s_taco_loss(const float* mel, const float* post, const float* target, size_t n_mel, const float* gate,
                  const float* gate_target, size_t n_gate, float* d_mel, float* d_post, float* d_gate, void* partial,
                  float* out, void* stream);

/* WaveGlowLoss (waveglow/glow.py:43-59): out[0] = (sum z^2 / (2 sigma^2) - sum_k sum log_s[k] - sum_k log_det[k]) / n_z.
 * log_s / n_log_s are HOST arrays of n_flows (<= 16) device pointers / element counts, log_det a device array of n_flows
 * floats.  d_z (optional) receives z / (sigma^2 n_z); the gradients w.r.t. log_s and log_det are the constant -1 / n_z.
 * partial = scratch of 256*2 doubles. */
int t2s_waveglow_loss(const float* z, size_t n_z, const float* const* log_s, const size_t* n_log_s, int n_flows,
                      const float* log_det, float sigma, float* d_z, void* partial, float* out, void* stream);

typedef struct t2s_bn_bwd_args {
    const float *x, *mean, *var, *gamma, *beta; float eps;
    const float *dout_f32; const void *dout_hi, *dout_lo;
    const unsigned char *mask; float mask_scale; int act;
    float *dgamma, *dbeta; void *dx_hi, *dx_lo;
    int B, C, T, Lp, halo;
} t2s_bn_bwd_args;
/* training-mode BatchNorm1d backward fused with the backward of activation + dropout; dx as planes.
 * ABI v4: `partial` = scratch of B * C * 2 doubles (the per-batch-element sums S1, S2, added in a fixed order: bitwise reproducible) */
int t2s_bn_bwd(const t2s_bn_bwd_args* a, void* partial, void* stream);
/* out[j] = sum_i in[i][j] ; out = a (+ b) (+ c): b and c optional, so it is also the stream-ordered copy of an f32 buffer */
int t2s_sum_axis0(const float* in, int n0, int n, float* out, void* stream);
int t2s_add3(const float* a, const float* b, const float* c, size_t n, float* out, void* stream);
/* out[i] = (in ? in[i] : 1) * scalar[0] * mul, scalar in DEVICE memory: how a loss hands its saved gradient times the upstream
 * gradient (a 0-dim device tensor in `loss.backward()`, waveglow/train.py:122) to the model's backward without an eager operator */
int t2s_scale_by_scalar(const float* in, size_t n, const float* scalar, float mul, float* out, void* stream);
/* planes -> f32 [B][C][L] (accumulate=1: +=) */
int t2s_planes_to_f32(const void* X_hi, const void* X_lo, int B, int C, int L, int Lp, int halo, float* out,
                      int accumulate, void* stream);

/* ---- audio front-end / back-end ------------------------------------------------------------------
 * STFT.transform (utils/stft.py:72-99): audio [B][T] -> frames F = T/hop + 1, bins c = n_fft/2 + 1.
 *   fwd_basis [2c][n_fft] (windowed Fourier basis, real rows then imaginary rows); scratch xp [B][ldp] (ldp >= T + n_fft,
 *   ldp % 4 == 0) and ft [B][F][2c]; outputs (each optional) mag / phase [B][c][F] and magT [B*F][ld_mt] (zero padded rows:
 *   the operand of t2s_mel_from_mag, ld_mt % 16 == 0). */
int t2s_stft_transform(const float* audio, int B, int T, const float* fwd_basis, int n_fft, int hop, float* xp, long ldp,
                       float* ft, float* mag, float* phase, float* magT, long ld_mt, void* stream);
/* TacotronSTFT.mel_spectrogram after the STFT (utils/layers.py:76-78): mel [B][n_mel][F] = log(max(mel_basis . mag, clip));
 * mel_basis_p [n_mel][ld_mt] zero padded; clip <= 0 skips the log. */
int t2s_mel_from_mag(const float* magT, long ld_mt, int B, int F, const float* mel_basis_p, int n_mel, float clip, float* mel,
                     void* stream);
/* STFT.inverse (utils/stft.py:101-129) with the Denoiser's spectral subtraction fused when bias != NULL
 * (waveglow/denoiser.py:36-38: m = max(mag - strength * bias[bin], 0)).  inv_basis_t [n_fft][ld_rc] = inverse basis transposed,
 * zero padded to ld_rc >= 2c (% 16 == 0); win_sq [n_fft] squared window or NULL; scratch rc [B*F][ld_rc], frames [B][F][n_fft];
 * out [B][hop * (F - 1)]. */
int t2s_stft_inverse(const float* mag, const float* phase, int B, int F, int n_fft, int hop, const float* inv_basis_t, long ld_rc,
                     const float* bias, float strength, const float* win_sq, float tiny, float* rc, float* frames, float* out,
                     void* stream);

#ifdef __cplusplus
}
#endif
#endif
