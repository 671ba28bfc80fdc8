// extern "C" boundary of the Tacotron-2 kernels: validation, argument blocks, and the decode-step driver.
#include "../../include/t2s_hip.h"
#include "t2s_kernels.h"
#include "tacotron_ops.h"

#include <stdlib.h>
#include <string.h>

extern "C" int t2s_internal_fail_hip(int e);   // defined in t2s_api.hip (records the HIP error text)

#define T2S_CHECK_HIP(expr)                                        \
    do {                                                           \
        hipError_t _e = (expr);                                    \
        if (_e != hipSuccess) return t2s_internal_fail_hip((int)_e); \
    } while (0)

static inline bool al16(const void* p) { return ((uintptr_t)p & 15) == 0; }

static int gemv_args_ok(const GemvArgs& a) {
    if (!a.W1 || !a.x1 || !a.y || a.rows <= 0 || a.items <= 0) return 0;
    const int K = a.n1 + a.n2 + a.n3;
    if (K != a.k1 + a.k2 || K > 4096) return 0;
    if ((a.n1 | a.n2 | a.n3 | a.k1 | a.k2 | a.ld1 | a.ld2) & 3) return 0;
    if ((a.sx1 | a.sx2 | a.sx3) & 3) return 0;
    if (!al16(a.W1) || !al16(a.x1) || (a.W2 && !al16(a.W2)) || (a.x2 && !al16(a.x2)) || (a.x3 && !al16(a.x3))) return 0;
    if ((a.n2 > 0 && !a.x2) || (a.n3 > 0 && !a.x3) || (a.k2 > 0 && !a.W2)) return 0;
    return 1;
}

extern "C" {

int t2s_gemv(const float* W1, int ld1, int k1, const float* W2, int ld2, int k2, const float* x1, int n1, long sx1,
             const float* x2, int n2, long sx2, const float* x3, int n3, long sx3, const float* bias1,
             const float* bias2, float* y, long sy_item, long sy_row, int rows, int items, int act,
             const unsigned char* mask, long smask_item, float mask_scale, void* stream) {
    GemvArgs a;
    memset(&a, 0, sizeof(a));
    a.W1 = W1; a.ld1 = ld1; a.k1 = k1; a.W2 = W2; a.ld2 = ld2; a.k2 = k2;
    a.x1 = x1; a.n1 = n1; a.sx1 = sx1; a.x2 = x2; a.n2 = n2; a.sx2 = sx2; a.x3 = x3; a.n3 = n3; a.sx3 = sx3;
    a.bias1 = bias1; a.bias2 = bias2; a.y = y; a.sy_item = sy_item; a.sy_row = sy_row; a.rows = rows; a.items = items;
    a.act = act; a.mask = mask; a.smask_item = smask_item; a.mask_scale = mask_scale;
    if (!gemv_args_ok(a) || act < 0 || act > 2) return T2S_EINVAL;
    a.split_row = 0; a.y2 = nullptr; a.sy2_item = a.sy2_row = 0; a.act2 = 0; a.mask2 = nullptr; a.smask2_item = 0; a.mask2_scale = 1.f;
    T2S_CHECK_HIP(t2s_launch_gemv(a, (hipStream_t)stream));
    return T2S_OK;
}

int t2s_transpose(const float* in, float* out, int R, int C, void* stream) {
    if (!in || !out || R <= 0 || C <= 0) return T2S_EINVAL;
    T2S_CHECK_HIP(t2s_launch_transpose(in, out, R, C, (hipStream_t)stream));
    return T2S_OK;
}

int t2s_embed_planes(const long* ids, const float* emb, int B, int T, int E, int V, int Lp, int halo, void* X_hi,
                     void* X_lo, void* stream) {
    if (!ids || !emb || !X_hi || !X_lo || B <= 0 || T <= 0 || E <= 0 || V <= 0) return T2S_EINVAL;
    if (Lp < t2s_plane_rows(T, halo)) return T2S_EINVAL;
    T2S_CHECK_HIP(t2s_launch_embed_planes(ids, emb, B, T, E, V, Lp, halo, (u16*)X_hi, (u16*)X_lo, (hipStream_t)stream));
    return T2S_OK;
}

int t2s_f32_to_planes(const float* x, int B, int C, int L, int Lp, int halo, void* X_hi, void* X_lo, void* stream) {
    if (!x || !X_hi || !X_lo || B <= 0 || C <= 0 || L <= 0 || Lp < t2s_plane_rows(L, halo)) return T2S_EINVAL;
    T2S_CHECK_HIP(t2s_launch_f32_to_planes(x, B, C, L, Lp, halo, (u16*)X_hi, (u16*)X_lo, (hipStream_t)stream));
    return T2S_OK;
}

int t2s_taco_parse_output(float* mel, float* mel_post, float* gate, const int* lengths, int B, int n_mel, int T, void* stream) {
    if (!mel || !mel_post || !gate || !lengths || B <= 0 || n_mel <= 0 || T <= 0 || B > 65535 || n_mel >= 65535) return T2S_EINVAL;
    T2S_CHECK_HIP(t2s_launch_parse_output(mel, mel_post, gate, lengths, B, n_mel, T, (hipStream_t)stream));
    return T2S_OK;
}

int t2s_bn_fold(const float* gamma, const float* beta, const float* mean, const float* var, const float* conv_bias,
                float eps, int C, float* scale, float* bias_out, void* stream) {
    if (!gamma || !beta || !mean || !var || !scale || !bias_out || C <= 0) return T2S_EINVAL;
    T2S_CHECK_HIP(t2s_launch_bn_fold(gamma, beta, mean, var, conv_bias, eps, C, scale, bias_out, (hipStream_t)stream));
    return T2S_OK;
}

int t2s_bn_train(const float* x, const float* gamma, const float* beta, float eps, int act, const unsigned char* mask,
                 float mask_scale, int B, int C, int T, int Lp, int halo, float* mean, float* var, void* O_hi, void* O_lo,
                 float* out_f32, void* stream) {
    if (!x || !gamma || !beta || !mean || !var || B <= 0 || C <= 0 || T <= 0 || act < 0 || act > 2) return T2S_EINVAL;
    if (!O_hi && !out_f32) return T2S_EINVAL;
    if (O_hi && (!O_lo || Lp < t2s_plane_rows(T, halo))) return T2S_EINVAL;
    T2S_CHECK_HIP(t2s_launch_bn_train(x, gamma, beta, eps, act, mask, mask_scale, B, C, T, Lp, halo, mean, var,
                                      (u16*)O_hi, (u16*)O_lo, out_f32, (hipStream_t)stream));
    return T2S_OK;
}

int t2s_bn_running_update(const float* mean, const float* var, float* running_mean, float* running_var,
                          long long* num_batches_tracked, float momentum, long long n, int C, void* stream) {
    if (!mean || !var || !running_mean || !running_var || C <= 0 || n <= 0 || !(momentum >= 0.f && momentum <= 1.f)) return T2S_EINVAL;
    const float unbias = n > 1 ? (float)((double)n / (double)(n - 1)) : 1.f;
    T2S_CHECK_HIP(t2s_launch_bn_running_update(mean, var, running_mean, running_var, num_batches_tracked, momentum, unbias, C,
                                               (hipStream_t)stream));
    return T2S_OK;
}

int t2s_zero_fill(void* p, size_t bytes, void* stream) {
    if (!p || ((uintptr_t)p & 15)) return T2S_EINVAL;
    if (bytes == 0) return T2S_OK;
    T2S_CHECK_HIP(t2s_launch_zero_fill(p, bytes, (hipStream_t)stream));
    return T2S_OK;
}

int t2s_taco_encoder_lstm(const float* gx, const float* whhT_fwd, const float* whhT_rev, const int* lengths, float* out,
                          int B, int T, int H, int T_out, float* gates_save, float* c_save, void* stream) {
    if (!gx || !whhT_fwd || !whhT_rev || !out || B <= 0 || T <= 0 || T_out <= 0 || T_out > T || 4 * H != 1024) return T2S_EINVAL;
    if ((gates_save == nullptr) != (c_save == nullptr)) return T2S_EINVAL;
    T2S_CHECK_HIP(t2s_launch_lstm_seq(gx, whhT_fwd, whhT_rev, lengths, out, B, T, H, T_out, gates_save, c_save, (hipStream_t)stream));
    return T2S_OK;
}

long t2s_taco_lstm_xbuf_bytes(int B) { return B > 0 ? ((long)2 * B * 2 * 256 + 1) * 8 : -1; }

int t2s_taco_encoder_lstm_split(const float* gx, const float* whhT_fwd, const float* whhT_rev, const int* lengths, float* out,
                                int B, int T, int H, int T_out, float* gates_save, float* c_save, void* xbuf, unsigned epoch,
                                void* stream) {
    if (!gx || !whhT_fwd || !whhT_rev || !out || !xbuf || B <= 0 || T <= 0 || T >= 4095 || T_out <= 0 || T_out > T || H != 256 ||
        ((uintptr_t)xbuf & 7))
        return T2S_EINVAL;
    if ((gates_save == nullptr) != (c_save == nullptr)) return T2S_EINVAL;
    T2S_CHECK_HIP(t2s_launch_lstm_seq_split(gx, whhT_fwd, whhT_rev, lengths, out, B, T, T_out, gates_save, c_save,
                                            (unsigned long long*)xbuf, epoch, (hipStream_t)stream));
    return T2S_OK;
}

int t2s_bernoulli_mask(unsigned char* mask, size_t n, unsigned long long seed, unsigned long long offset, float keep_prob,
                       void* stream) {
    if (!mask || n == 0 || !(keep_prob > 0.f && keep_prob <= 1.f)) return T2S_EINVAL;
    T2S_CHECK_HIP(t2s_launch_bernoulli_mask(mask, n, seed, offset, keep_prob, (hipStream_t)stream));
    return T2S_OK;
}

int t2s_taco_stop_check(const float* mel_gate_out, int B, int n_mel, int T_cap, int step0, int n, float threshold,
                        int* stop_step, void* stream) {
    if (!mel_gate_out || !stop_step || B <= 0 || step0 < 0 || n <= 0 || step0 + n > T_cap) return T2S_EINVAL;
    T2S_CHECK_HIP(t2s_launch_stop_check(mel_gate_out + (size_t)n_mel * T_cap, B, (n_mel + 1) * T_cap, step0, n, threshold,
                                        stop_step, (hipStream_t)stream));
    return T2S_OK;
}

int t2s_taco_attention(const float* h_att, const float* memory, const float* pmem, const int* lengths, float* w, float* w_cum,
                       float* ctx, float* q_scratch, float* e_scratch, const float* w_query, const float* w_loc_conv,
                       const float* w_loc_dense, const float* w_loc_denseT, const float* w_v, int B, int T, int att_rnn,
                       int att_dim, int enc_dim, int loc_filters, int loc_kernel, void* stream_) {
    if (!h_att || !memory || !pmem || !w || !w_cum || !ctx || !q_scratch || !e_scratch || !w_query || !w_loc_conv || !w_loc_dense ||
        !w_v || B <= 0 || T <= 0 || (att_rnn & 3) || (enc_dim & 3) || att_dim <= 0 || att_dim > 128 || loc_filters <= 0 ||
        loc_filters > 32 || loc_kernel <= 0 || loc_kernel > 63 || !(loc_kernel & 1))
        return T2S_EINVAL;
    hipStream_t stream = (hipStream_t)stream_;
    AttArgs aa;
    memset(&aa, 0, sizeof(aa));
    aa.q = q_scratch; aa.w_loc_conv = w_loc_conv; aa.w_loc_dense = w_loc_dense; aa.w_v = w_v; aa.pmem = pmem; aa.memory = memory;
    aa.lengths = lengths; aa.w_prev = w; aa.w_cum = w_cum; aa.energies = e_scratch; aa.ctx = ctx;
    aa.B = B; aa.T = T; aa.att_dim = att_dim; aa.enc_dim = enc_dim; aa.loc_f = loc_filters; aa.loc_ks = loc_kernel;
    aa.w_query = w_query; aa.h_att = h_att; aa.w_loc_denseT = w_loc_denseT; aa.att_rnn = att_rnn;
    static const int fused_max_b = getenv("T2S_ATT_FUSED_MAXB") ? atoi(getenv("T2S_ATT_FUSED_MAXB")) : 8;
    if (B <= fused_max_b && T <= 512 && w_loc_denseT && enc_dim <= 512 && att_rnn <= 1024) {
        T2S_CHECK_HIP(t2s_launch_att_fused(aa, stream));
        return T2S_OK;
    }
    GemvArgs qa;
    memset(&qa, 0, sizeof(qa));
    qa.W1 = w_query; qa.ld1 = att_rnn; qa.k1 = att_rnn; qa.x1 = h_att; qa.n1 = att_rnn; qa.sx1 = att_rnn;
    qa.y = q_scratch; qa.sy_item = att_dim; qa.sy_row = 1; qa.rows = att_dim; qa.items = B; qa.mask_scale = 1.f;
    if (!gemv_args_ok(qa)) return T2S_EINVAL;
    T2S_CHECK_HIP(t2s_launch_gemv(qa, stream));
    T2S_CHECK_HIP(t2s_launch_att_energy(aa, stream));
    T2S_CHECK_HIP(t2s_launch_att_softmax_ctx(aa, stream));
    return T2S_OK;
}

int t2s_taco_decode_steps(const t2s_taco_decoder* d, int step0, int n_steps, void* stream_) {
    if (!d || step0 < 0 || n_steps <= 0) return T2S_EINVAL;
    hipStream_t stream = (hipStream_t)stream_;
    const int B = d->B, T = d->T_in, P = d->prenet_dim, E = d->enc_dim, A = d->att_rnn_dim, D = d->dec_rnn_dim;
    if (B <= 0 || T <= 0 || A != D || (A & 3) || (P & 3) || (E & 3) || d->att_dim > 128 || d->loc_filters > 32 ||
        d->loc_kernel > 63 || !(d->loc_kernel & 1))
        return T2S_EINVAL;
    if (step0 + n_steps > d->T_cap) return T2S_EINVAL;
    if (!d->att_w_ih || !d->att_w_hh || !d->dec_w_ih || !d->dec_w_hh || !d->w_query || !d->w_loc_conv ||
        !d->w_loc_dense || !d->w_v || !d->memory || !d->pmem || !d->att_h0 || !d->att_h1 || !d->att_c || !d->dec_h0 ||
        !d->dec_h1 || !d->dec_c || !d->att_w || !d->att_wcum || !d->ctx || !d->q || !d->energies || !d->align_out)
        return T2S_EINVAL;
    if (d->teacher_forced) {
        if (!d->pre_all || !d->hc_all) return T2S_EINVAL;
    } else {
        if (!d->w_proj || !d->b_proj || !d->w_projpre || !d->b_projpre || !d->w_pre2 || !d->pre1 || !d->pre2 ||
            !d->mel_gate_out || !d->prenet_masks)
            return T2S_EINVAL;
    }
    // Teacher-forced decoding with the per-step saves at hand (training): the decoder cell of step s feeds only the decoder
    // cell of step s + 1 and the projection after the loop - never the attention chain (attention cell -> query -> energies ->
    // softmax + context), whose next prenet input is given.  So it runs on the library's helper stream from the saved copies
    // of h_att[s] and ctx[s] (att_h_all, hc_all), a chunk of steps behind the attention chain, and drops out of the serial chain.
    // With one event per STEP this measured equal (128.0 vs 128.2 ms per train step, profiles/r03_taco_timeline_fwd_split.md: the
    // record opens a 7 us gap on the critical stream); with one event per 16 steps: 95.7 -> 91.9 ms.  T2S_DECODE_SPLIT=0: off.
    static const bool want_split = !(getenv("T2S_DECODE_SPLIT") && atoi(getenv("T2S_DECODE_SPLIT")) == 0);
    const bool split = d->teacher_forced && d->att_h_all && d->hc_all && want_split;
    T2sHelperStream hs;
    if (split) T2S_CHECK_HIP(t2s_helper_stream_acquire(hs));
    struct Join {           // whatever happens below, the caller's stream waits for the helper before this call returns
        T2sHelperStream& hs; hipStream_t stream; bool on;
        ~Join() {
            if (on && hipEventRecord(hs.ev_join, hs.side) == hipSuccess) (void)hipStreamWaitEvent(stream, hs.ev_join, 0);
        }
    } join{hs, stream, split};
    // Streamed gate partials (ABI v4, t2s_taco_decoder::gate_part): autoregressive small-batch decode only.  T2S_DECODE_STREAM=0: off.
    static const bool want_stream = !(getenv("T2S_DECODE_STREAM") && atoi(getenv("T2S_DECODE_STREAM")) == 0);
    const size_t GP = (size_t)B * 4 * A;                   // one [B][4H] block of gate_part
    // Decoder cells a chunk behind the chain (split): the input half of their pre-activations as ONE product per chunk
    // (t2s_taco_decoder::dec_in_part, 16 steps of scratch).  Built and measured NEGATIVE (profiles/r04_taco_chunk_gemm_ab.txt, same
    // box, alternating: teacher-forced forward at B = 32 35.5 / 35.6 with it against 35.0 / 34.7 ms, train step equal): the small-batch
    // GEMM kernel re-reads its 16 weight rows per group of 32 items and the 32 input vectors per 16 rows (1.2 GB of L2 reads per
    // 512-item chunk), which costs what the per-step cells save.  Off unless T2S_DECODE_CHUNK_GEMM=1.
    static const bool want_chunk_gemm = getenv("T2S_DECODE_CHUNK_GEMM") && atoi(getenv("T2S_DECODE_CHUNK_GEMM")) != 0;
    // Paced decoder cells (t2s_taco_decoder::pace_flag): the helper stream's cell of step s - 1 is released by a word the attention
    // cell's launch of step s stores as it STARTS, i.e. when the attention of step s - 1 is complete.  It is enqueued ~4 us later, finds
    // the chip held by that attention cell, and runs as its workgroups retire - beside the attention launch of step s, whose small
    // workgroups share a CU with it - and is over when the next attention cell needs the CUs.  In bursts of 16 (the chunked form) the
    // helper's cells kept the chain's next attention cell from starting: their time ADDED to the chain's.  T2S_DECODE_PACED=0: chunks.
    static const bool want_paced = !(getenv("T2S_DECODE_PACED") && atoi(getenv("T2S_DECODE_PACED")) == 0);
    const bool paced = split && want_paced && d->pace_flag && B > 8 && !((uintptr_t)d->pace_flag & 7);
    const bool chunk_gemm = split && !paced && want_chunk_gemm && d->dec_in_part && B > 8 && !((A | D | E) & 31);
    int part_c0 = 0;                                       // first step of the chunk whose products dec_in_part holds
    auto body = [&](int s, bool do_att, bool do_dec) -> int {
        float* ah_in = (s & 1) ? d->att_h1 : d->att_h0;
        float* ah_out = (s & 1) ? d->att_h0 : d->att_h1;
        float* dh_in = (s & 1) ? d->dec_h1 : d->dec_h0;
        float* dh_out = (s & 1) ? d->dec_h0 : d->dec_h1;
        GateStreamArgs gs;
        memset(&gs, 0, sizeof(gs));
        bool stream_gates = false, fold_pre2 = false, use_ploc = false;
        if (do_att) {
        // 1. attention LSTMCell on [prenet_out | context]
        LstmCellArgs ca;
        memset(&ca, 0, sizeof(ca));
        ca.W_ih = d->att_w_ih; ca.W_hh = d->att_w_hh; ca.b_ih = d->att_b_ih; ca.b_hh = d->att_b_hh;
        ca.x1 = d->teacher_forced ? d->pre_all + (size_t)s * B * P : d->pre2;
        ca.n1 = P; ca.sx1 = P; ca.x2 = d->ctx; ca.n2 = E; ca.sx2 = E;
        ca.h_in = ah_in; ca.h_out = ah_out; ca.c = d->att_c; ca.B = B; ca.H = A;
        if (d->att_drop) { ca.drop_mask = d->att_drop + (size_t)s * B * A; ca.drop_scale = d->att_drop_scale; }
        // one fused attention launch per step (one workgroup per batch element) up to this batch; beyond it the three-kernel
        // form (query GEMV, energies, softmax + context) fills the chip better (measured at B = 32, T_in = 256: no difference)
        static const int fused_max_b = getenv("T2S_ATT_FUSED_MAXB") ? atoi(getenv("T2S_ATT_FUSED_MAXB")) : 8;
        const bool fused_att = B <= fused_max_b && T <= 512 && d->w_loc_denseT && d->att_dim <= 128;
        if (d->att_gates_all) { ca.gates_out = d->att_gates_all + (size_t)s * B * 4 * A; ca.c_out = d->att_c_all + (size_t)s * B * A; }
        if (d->att_h_all) { ca.h_copy = d->att_h_all + (size_t)s * B * A; ca.s_copy = A; }
        // small batch: the attention cell's workgroups emit partial queries (their own hidden units' columns of W_query), so
        // the fused attention kernel sums 128 KB of partials instead of pulling the 512 KB of W_query through one CU
        const bool q_parts = fused_att && B <= 8 && d->q_part != nullptr;      // (9+ items: the cells run on sbgemm.hip, no partials)
        const int units = ca.gates_out ? 2 : 4;             // hidden units per workgroup of lstm_cell_kernel (training / eval)
        if (q_parts) { ca.w_q = d->w_query; ca.q_part = d->q_part; ca.q_dim = d->att_dim; }
        // large batch (matrix-core cells): the same idea - every cell workgroup (4 hidden units) leaves a partial query and the
        // energies kernel sums the A / 4 = 256 of them - takes the query GEMM off the serial chain.  T2S_QPART_BIG=0: the GEMM.
        static const bool want_qbig = !(getenv("T2S_QPART_BIG") && atoi(getenv("T2S_QPART_BIG")) == 0);
        static const bool att_valu = getenv("T2S_ATT_VALU") != nullptr;
        bool q_big = false;
        if (!fused_att && !q_parts && want_qbig && !att_valu && d->q_part && A == 1024 && d->att_dim == 128 && d->loc_filters == 32 &&
            d->loc_kernel <= 31 && d->w_loc_denseT) {
            ca.w_q = d->w_query; ca.q_part = d->q_part; ca.q_dim = d->att_dim;
            q_big = t2s_sbgemm_lstm_ok(ca);
            if (!q_big) { ca.w_q = nullptr; ca.q_part = nullptr; ca.q_dim = 0; }
        }
        // streamed gates: W_hh_att . h_att(s-1) was left in gate_part[2] by the previous step's attention launch (zero at step 0)
        // (up to 4 items: the role's dot products and reductions are per item - at B = 8 the launch takes longer than the two cells save,
        // 73.9 vs 67.4 us per step; B = 4: 48.2 vs 49.9, B = 2: 36.2 vs 42.0, B = 1: 29.6 vs 37.3.  T2S_DECODE_STREAM_MAXB overrides)
        static const int stream_max_b = getenv("T2S_DECODE_STREAM_MAXB") ? atoi(getenv("T2S_DECODE_STREAM_MAXB")) : 4;
        if (want_stream && d->gate_part && B <= stream_max_b && !d->teacher_forced && fused_att && q_parts && !ca.gates_out && A == 1024 &&
            D == 1024) {
            gs.W0 = d->dec_w_hh; gs.ld0 = D; gs.x0 = dh_in; gs.out0 = d->gate_part;
            gs.W1 = d->dec_w_ih; gs.ld1 = A + E; gs.out1 = d->gate_part + GP;
            gs.W2 = d->att_w_hh; gs.ld2 = A; gs.out2 = d->gate_part + 2 * GP; gs.x12 = ah_out;
            gs.rows = 4 * A; gs.H = A; gs.B = B;
            // (the role must fit this device: one pass of 3-4 row units per wave over one workgroup per CU - else the plain chain)
            AttArgs probe;
            memset(&probe, 0, sizeof(probe));
            probe.B = B; probe.T = T; probe.att_dim = d->att_dim; probe.loc_f = d->loc_filters;
            stream_gates = t2s_att_fused_stream_ok(probe, gs);
        }
        if (stream_gates) { ca.h_in = nullptr; ca.pre_a = d->gate_part + 2 * GP; }
        bool sig_by_kernel = false;
        if (paced) {
            ca.sig_ptr = (unsigned*)d->pace_flag; ca.sig_val = (unsigned)s + 1u;
            sig_by_kernel = t2s_sbgemm_lstm_ok(ca);          // (the matrix-core cell stores the word itself)
            if (!sig_by_kernel) T2S_CHECK_HIP(t2s_launch_pace_signal(ca.sig_ptr, ca.sig_val, stream));
        }
        // ... and with it the prenet's second layer folded into this launch (every workgroup recomputes the 256 outputs from
        // pre1 and W_pre2 out of L2) instead of a GEMV launch of its own at the end of the previous step.  T2S_DECODE_FOLD_PRE2=0: off
        static const bool want_fold = !(getenv("T2S_DECODE_FOLD_PRE2") && atoi(getenv("T2S_DECODE_FOLD_PRE2")) == 0);
        fold_pre2 = stream_gates && want_fold && d->w_pre2T && P == 256 && E == 512 && s < d->mask_steps;
        if (fold_pre2) {
            ca.x1 = nullptr; ca.w_p2 = d->w_pre2T; ca.p1 = d->pre1;
            ca.p2_mask = d->prenet_masks + (size_t)s * B * 2 * P + P; ca.s_p2_mask = 2 * P; ca.p2_scale = 2.0f;
        }
        T2S_CHECK_HIP(t2s_launch_lstm_cell(ca, stream));
        // 2.-4. attention: query, location-sensitive energies, softmax, context, cumulative weights
        AttArgs aa;
        memset(&aa, 0, sizeof(aa));
        aa.q = d->q; aa.w_loc_conv = d->w_loc_conv; aa.w_loc_dense = d->w_loc_dense; aa.w_v = d->w_v;
        aa.pmem = d->pmem; aa.memory = d->memory; aa.lengths = d->mem_lengths;
        aa.w_prev = d->att_w; aa.w_cum = d->att_wcum; aa.energies = d->energies; aa.ctx = d->ctx;
        aa.align_out = d->align_out + (size_t)s * T; aa.s_align_b = (long)d->T_cap * T;
        if (d->teacher_forced) { aa.ctx_copy = d->hc_all + (size_t)s * B * (D + E) + D; aa.s_ctx_copy = D + E; }
        aa.B = B; aa.T = T; aa.att_dim = d->att_dim; aa.enc_dim = E; aa.loc_f = d->loc_filters; aa.loc_ks = d->loc_kernel;
        aa.w_query = d->w_query; aa.h_att = ah_out; aa.w_loc_denseT = d->w_loc_denseT; aa.att_rnn = A;
        if (d->q_all) aa.q_save = d->q_all + (size_t)s * B * d->att_dim;
        if (d->wcum_all) aa.wcum_save = d->wcum_all + (size_t)s * B * T;
        if (q_parts) { aa.q_part = d->q_part; aa.n_part = A / units; }
        if (fused_att) {
            // small batch: one fused launch per step (one workgroup per batch element)
            if (stream_gates && !t2s_att_fused_stream_ok(aa, gs)) return T2S_EINVAL;
            // the location term of this step came out of the previous step's projection launch (zero at step 0).  T2S_DECODE_PLOC=0: off
            static const bool want_ploc = !(getenv("T2S_DECODE_PLOC") && atoi(getenv("T2S_DECODE_PLOC")) == 0);
            use_ploc = stream_gates && want_ploc && d->ploc && d->att_dim == 128 && d->loc_filters == 32 && d->loc_kernel <= 31 && !aa.q_save &&
                       !aa.wcum_save;
            if (use_ploc) aa.ploc = d->ploc;
            T2S_CHECK_HIP(t2s_launch_att_fused(aa, stream, stream_gates ? &gs : nullptr));
        } else {
            GemvArgs qa;
            memset(&qa, 0, sizeof(qa));
            qa.W1 = d->w_query; qa.ld1 = A; qa.k1 = A; qa.x1 = ah_out; qa.n1 = A; qa.sx1 = A;
            qa.y = d->q_all ? d->q_all + (size_t)s * B * d->att_dim : d->q; qa.sy_item = d->att_dim; qa.sy_row = 1;
            qa.rows = d->att_dim; qa.items = B;
            aa.q = qa.y;
            if (q_big) { aa.q_part = d->q_part; aa.n_part = A / 4; aa.q_out = qa.y; aa.q_save = nullptr; }
            else T2S_CHECK_HIP(t2s_launch_gemv(qa, stream));
            // energies, softmax, cumulative weights and context in ONE launch where the shape allows (t2s_taco_decoder::att_xbuf: the
            // tiles of an element exchange their energies through tagged granules).  T2S_ATT_ONE_LAUNCH=0: two launches.
            static const bool want_one = !(getenv("T2S_ATT_ONE_LAUNCH") && atoi(getenv("T2S_ATT_ONE_LAUNCH")) == 0);
            bool one = false;
            if (want_one && d->att_xbuf) {
                aa.xbuf = (unsigned long long*)d->att_xbuf; aa.tag = (unsigned)s + 1u;
                one = t2s_att_energy_ctx_ok(aa);
                if (!one) { aa.xbuf = nullptr; aa.tag = 0; }
            }
            T2S_CHECK_HIP(t2s_launch_att_energy(aa, stream));
            if (!one) T2S_CHECK_HIP(t2s_launch_att_softmax_ctx(aa, stream));
        }
        }
        if (!do_dec) return T2S_OK;
        // 5. decoder LSTMCell on [h_att | context]
        LstmCellArgs cd;
        memset(&cd, 0, sizeof(cd));
        cd.W_ih = d->dec_w_ih; cd.W_hh = d->dec_w_hh; cd.b_ih = d->dec_b_ih; cd.b_hh = d->dec_b_hh;
        cd.x1 = ah_out; cd.n1 = A; cd.sx1 = A; cd.x2 = d->ctx; cd.n2 = E; cd.sx2 = E;
        hipStream_t dstream = stream;
        if (split) {
            cd.x1 = d->att_h_all + (size_t)s * B * A;
            cd.x2 = d->hc_all + (size_t)s * B * (D + E) + D; cd.sx2 = D + E;
            dstream = hs.side;
        }
        cd.h_in = dh_in; cd.h_out = dh_out; cd.c = d->dec_c; cd.B = B; cd.H = D;
        if (stream_gates) {
            // only the context columns of W_ih are left to stream (8.4 MB of 42): W_hh . h_dec(s-1) and W_ih[:, :A] . h_att(s)
            // came out of the attention launch as gate_part[0], gate_part[1]
            cd.W_ih = d->dec_w_ih + A; cd.ld_ih = A + E; cd.x1 = d->ctx; cd.n1 = E; cd.sx1 = E; cd.x2 = nullptr; cd.n2 = 0; cd.sx2 = 0;
            cd.h_in = nullptr; cd.pre_a = d->gate_part; cd.pre_b = d->gate_part + GP;
        }
        if (chunk_gemm) {
            // W_ih . [h_att(s) | ctx(s)] is in dec_in_part[s - part_c0]: only W_hh . h_dec(s-1) is left to stream (17 of 42 MB)
            cd.x1 = nullptr; cd.n1 = 0; cd.sx1 = 0; cd.x2 = nullptr; cd.n2 = 0; cd.sx2 = 0;
            cd.pre_a = d->dec_in_part + (size_t)(s - part_c0) * B * 4 * D;
        }
        if (d->dec_drop) { cd.drop_mask = d->dec_drop + (size_t)s * B * D; cd.drop_scale = d->dec_drop_scale; }
        if (d->teacher_forced) { cd.h_copy = d->hc_all + (size_t)s * B * (D + E); cd.s_copy = D + E; }
        if (d->dec_gates_all) { cd.gates_out = d->dec_gates_all + (size_t)s * B * 4 * D; cd.c_out = d->dec_c_all + (size_t)s * B * D; }
        T2S_CHECK_HIP(t2s_launch_lstm_cell(cd, dstream));
        if (!d->teacher_forced) {
            // 6./7. mel frame + gate logit = W_proj [h_dec | ctx] + b, and (same launch, second row block) layer 0
            //       of the next step's prenet through the precomposed matrix W_pre0 . W_proj (always-on dropout,
            //       modules.py:21), so the mel frame needs no extra hop before the prenet.
            const bool more = s + 1 < d->mask_steps;
            // the projection launch; with use_ploc it also carries the location term of the NEXT step's attention on the CUs the
            // GEMV leaves idle (whatever form the launch takes: the next attention launch reads ploc)
            auto launch_proj = [&](const GemvArgs& g) -> hipError_t {
                if (!use_ploc) return t2s_launch_gemv(g, stream);
                LocPreArgs lp;
                memset(&lp, 0, sizeof(lp));
                lp.w = d->att_w; lp.w_cum = d->att_wcum; lp.w_loc_conv = d->w_loc_conv; lp.w_loc_denseT = d->w_loc_denseT;
                lp.ploc = d->ploc; lp.B = B; lp.T = T; lp.loc_ks = d->loc_kernel;
                return t2s_launch_gemv_with_loc(g, lp, stream);
            };
            const unsigned char* mk = d->prenet_masks + (size_t)(s + 1) * B * 2 * P;
            GemvArgs pa;
            memset(&pa, 0, sizeof(pa));
            pa.W1 = d->w_proj; pa.ld1 = D + E; pa.k1 = D + E;
            pa.x1 = dh_out; pa.n1 = D; pa.sx1 = D; pa.x2 = d->ctx; pa.n2 = E; pa.sx2 = E;
            pa.bias1 = d->b_proj; pa.y = d->mel_gate_out + s; pa.sy_item = (long)(d->n_mel + 1) * d->T_cap;
            pa.sy_row = d->T_cap; pa.rows = d->n_mel + 1; pa.items = B;
            if (more && d->w_projpre == d->w_proj + (size_t)(d->n_mel + 1) * (D + E) && d->b_projpre == d->b_proj + d->n_mel + 1) {
                pa.rows = d->n_mel + 1 + P; pa.split_row = d->n_mel + 1;
                pa.y2 = d->pre1; pa.sy2_item = P; pa.sy2_row = 1; pa.act2 = ACT_RELU;
                pa.mask2 = mk; pa.smask2_item = 2 * P; pa.mask2_scale = 2.0f;
                T2S_CHECK_HIP(launch_proj(pa));
            } else {
                T2S_CHECK_HIP(launch_proj(pa));
                if (more) {
                    GemvArgs p1;
                    memset(&p1, 0, sizeof(p1));
                    p1.W1 = d->w_projpre; p1.ld1 = D + E; p1.k1 = D + E;
                    p1.x1 = dh_out; p1.n1 = D; p1.sx1 = D; p1.x2 = d->ctx; p1.n2 = E; p1.sx2 = E;
                    p1.bias1 = d->b_projpre; p1.y = d->pre1; p1.sy_item = P; p1.sy_row = 1; p1.rows = P; p1.items = B;
                    p1.act = ACT_RELU; p1.mask = mk; p1.smask_item = 2 * P; p1.mask_scale = 2.0f;
                    T2S_CHECK_HIP(t2s_launch_gemv(p1, stream));
                }
            }
            if (more && !fold_pre2) {
                GemvArgs p2;
                memset(&p2, 0, sizeof(p2));
                p2.W1 = d->w_pre2; p2.ld1 = P; p2.k1 = P; p2.x1 = d->pre1; p2.n1 = P; p2.sx1 = P;
                p2.y = d->pre2; p2.sy_item = P; p2.sy_row = 1; p2.rows = P; p2.items = B;
                p2.act = ACT_RELU; p2.mask = mk + P; p2.smask_item = 2 * P; p2.mask_scale = 2.0f;
                T2S_CHECK_HIP(t2s_launch_gemv(p2, stream));
            }
        }
        return T2S_OK;
    };
    if (paced) {
        T2S_CHECK_HIP(hipEventRecord(hs.ev_step, stream));            // everything enqueued so far precedes the helper's first cell
        T2S_CHECK_HIP(hipStreamWaitEvent(hs.side, hs.ev_step, 0));
        unsigned long long* perr = (unsigned long long*)d->pace_flag + 1;
        for (int s = step0; s < step0 + n_steps; ++s) {
            const int rc = body(s, true, false);
            if (rc != T2S_OK) return rc;
            if (s > step0) {
                T2S_CHECK_HIP(t2s_launch_pace_wait((const unsigned*)d->pace_flag, (unsigned)s + 1u, perr, hs.side));
                const int rd = body(s - 1, false, true);
                if (rd != T2S_OK) return rd;
            }
        }
        // the last step's cell: after its attention (the chain is over: one event costs nothing now)
        T2S_CHECK_HIP(hipEventRecord(hs.ev_step, stream));
        T2S_CHECK_HIP(hipStreamWaitEvent(hs.side, hs.ev_step, 0));
        { const int rd = body(step0 + n_steps - 1, false, true); if (rd != T2S_OK) return rd; }
    } else if (split) {
        // the attention chain of `chunk` steps, ONE event, then the decoder cells of those steps on the helper stream while the
        // caller's stream goes on with the next chunk (T2S_DECODE_CHUNK, default 16)
        static const int chunk_env = getenv("T2S_DECODE_CHUNK") ? atoi(getenv("T2S_DECODE_CHUNK")) : 16;
        int chunk = chunk_env > 0 ? chunk_env : 1;
        if (chunk_gemm && chunk > 16) chunk = 16;          // (dec_in_part holds 16 steps)
        for (int c0 = step0; c0 < step0 + n_steps; c0 += chunk) {
            const int c1 = c0 + chunk < step0 + n_steps ? c0 + chunk : step0 + n_steps;
            for (int s = c0; s < c1; ++s) { const int rc = body(s, true, false); if (rc != T2S_OK) return rc; }
            T2S_CHECK_HIP(hipEventRecord(hs.ev_step, stream));        // h_att, ctx of the chunk saved (and all earlier work of the caller)
            T2S_CHECK_HIP(hipStreamWaitEvent(hs.side, hs.ev_step, 0));
            if (chunk_gemm) {
                GemvArgs g;
                memset(&g, 0, sizeof(g));
                g.W1 = d->dec_w_ih; g.ld1 = A + E; g.k1 = A + E;
                g.x1 = d->att_h_all + (size_t)c0 * B * A; g.n1 = A; g.sx1 = A;
                g.x2 = d->hc_all + (size_t)c0 * B * (D + E) + D; g.n2 = E; g.sx2 = D + E;
                g.y = d->dec_in_part; g.sy_item = 4 * D; g.sy_row = 1; g.rows = 4 * D; g.items = (c1 - c0) * B; g.mask_scale = 1.f;
                if (!gemv_args_ok(g)) return T2S_EINVAL;
                T2S_CHECK_HIP(t2s_launch_gemv(g, hs.side));
                part_c0 = c0;
            }
            for (int s = c0; s < c1; ++s) { const int rc = body(s, false, true); if (rc != T2S_OK) return rc; }
        }
    } else {
        for (int s = step0; s < step0 + n_steps; ++s) { const int rc = body(s, true, true); if (rc != T2S_OK) return rc; }
    }
    return T2S_OK;
}

}  // extern "C"
