// extern "C" boundary of the Tacotron-2 backward kernels.
#include "../../include/t2s_hip.h"
#include "t2s_kernels.h"
#include "taco_bwd_ops.h"
#include "tacotron_ops.h"

#include <stdlib.h>
#include <string.h>

#include <mutex>

extern "C" int t2s_internal_fail_hip(int e);
#define T2S_CHECK_HIP(expr)                                          \
    do {                                                             \
        hipError_t _e = (expr);                                      \
        if (_e != hipSuccess) return t2s_internal_fail_hip((int)_e); \
    } while (0)

// The one piece of library-owned state (documented in include/t2s_hip.h): the BPTT driver overlaps three dependent
// chains on three streams.  The caller hands over ONE stream, so the two helpers and their events belong to the library:
// one set per device, created on first use on THAT device, and a per-device mutex held while a call enqueues (two autograd
// threads may call into the same device; the call only enqueues work, it never blocks on the GPU).
namespace {
struct BpttStreams {
    std::mutex mu;
    hipStream_t side = nullptr, side2 = nullptr;
    hipEvent_t ev_main = nullptr, ev_side = nullptr, ev_energy = nullptr, ev_conv = nullptr, ev_join = nullptr;
    bool ready = false;
};
constexpr int kMaxDevices = 64;
BpttStreams g_bptt[kMaxDevices];

hipError_t bptt_streams_init(BpttStreams& s) {      // caller holds s.mu; the current device is the one s belongs to
    if (s.ready) return hipSuccess;
    hipError_t e;
    // T2S_HELPER_PRIO: stream priority of the helpers (HIP: lower number = higher priority; the range is clamped to the device's).
    // The helpers carry the chains nothing waits for per step (decoder cells), so a LOWER priority than the caller's stream lets the
    // serial chain's workgroups go first whenever both have some pending.
    int lo = 0, hi = 0;
    (void)hipDeviceGetStreamPriorityRange(&lo, &hi);         // lo = least (largest number), hi = greatest
    int prio = getenv("T2S_HELPER_PRIO") ? atoi(getenv("T2S_HELPER_PRIO")) : 0;
    prio = prio > lo ? lo : (prio < hi ? hi : prio);
    if ((e = hipStreamCreateWithPriority(&s.side, hipStreamNonBlocking, prio)) != hipSuccess) return e;
    if ((e = hipStreamCreateWithPriority(&s.side2, hipStreamNonBlocking, prio)) != hipSuccess) return e;
    if (getenv("T2S_HELPER_PRIO")) fprintf(stderr, "[t2s] helper streams at priority %d (device range %d .. %d)\n", prio, hi, lo);
    hipEvent_t* evs[] = {&s.ev_main, &s.ev_side, &s.ev_energy, &s.ev_conv, &s.ev_join};
    for (hipEvent_t* ev : evs)
        if ((e = hipEventCreateWithFlags(ev, hipEventDisableTiming)) != hipSuccess) return e;
    s.ready = true;
    return hipSuccess;
}
}  // namespace

// The forward decoder loop in teacher-forced mode borrows the first helper stream (t2s_api_taco.hip): the lock is held while
// the call enqueues, as in the BPTT driver.
hipError_t t2s_helper_stream_acquire(T2sHelperStream& h) {
    int device = 0;
    hipError_t e = hipGetDevice(&device);
    if (e != hipSuccess) return e;
    if (device < 0 || device >= kMaxDevices) return hipErrorInvalidDevice;
    BpttStreams& S = g_bptt[device];
    h.lock = std::unique_lock<std::mutex>(S.mu);
    if ((e = bptt_streams_init(S)) != hipSuccess) return e;
    h.side = S.side; h.ev_step = S.ev_main; h.ev_join = S.ev_join;
    return hipSuccess;
}

extern "C" {

int t2s_rows_to_tm(const float* x, long ld, int items, int items_pad, int shift, int C, void* dst_hi, void* dst_lo,
                   int Npad, int n_off, void* stream) {
    if (!x || !dst_hi || !dst_lo || items <= 0 || items_pad % 32 || items_pad < items || C <= 0 || n_off < 0 ||
        n_off + C > Npad)
        return T2S_EINVAL;
    T2S_CHECK_HIP(t2s_launch_rows_to_tm(x, ld, items, items_pad, shift, C, (u16*)dst_hi, (u16*)dst_lo, Npad, n_off,
                                        (hipStream_t)stream));
    return T2S_OK;
}

int t2s_rows_to_tm_batched(const float* x, long ld, long x_bstride, int items, int items_pad, int shift, int C, void* dst_hi,
                           void* dst_lo, long dst_bstride, int Npad, int n_off, int nb, void* stream) {
    if (!x || !dst_hi || !dst_lo || items <= 0 || items_pad % 32 || items_pad < items || C <= 0 || n_off < 0 ||
        n_off + C > Npad || nb <= 0 || nb > 65535)
        return T2S_EINVAL;
    T2S_CHECK_HIP(t2s_launch_rows_to_tm_batched(x, ld, x_bstride, items, items_pad, shift, C, (u16*)dst_hi, (u16*)dst_lo,
                                                dst_bstride, Npad, n_off, nb, (hipStream_t)stream));
    return T2S_OK;
}

int t2s_lstm_cell_bwd(const float* dh1, long s1, const float* dh2, long s2, const float* dh3, long s3,
                      const unsigned char* drop_mask, float drop_scale, const float* gates, const float* c_new,
                      const float* c_prev, float* dc_carry, float* dgates, int B, int H, void* stream) {
    if (!gates || !c_new || !dc_carry || !dgates || B <= 0 || H <= 0) return T2S_EINVAL;
    LstmBwdArgs a;
    a.dh1 = dh1; a.s1 = s1; a.dh2 = dh2; a.s2 = s2; a.dh3 = dh3; a.s3 = s3;
    a.drop_mask = drop_mask; a.drop_scale = drop_scale; a.gates = gates; a.c_new = c_new; a.c_prev = c_prev;
    a.dc_carry = dc_carry; a.dgates = dgates; a.B = B; a.H = H;
    a.wq = nullptr; a.dq = nullptr; a.q_dim = 0; a.dq_part = nullptr; a.dq_nchunk = 0; a.dq_out = nullptr;
    T2S_CHECK_HIP(t2s_launch_lstm_cell_bwd(a, (hipStream_t)stream));
    return T2S_OK;
}

int t2s_relu_drop_bwd(const float* dy, const float* y, float scale, size_t n, float* dz, void* stream) {
    if (!dy || !y || !dz || n == 0) return T2S_EINVAL;
    T2S_CHECK_HIP(t2s_launch_relu_drop_bwd(dy, y, scale, n, dz, (hipStream_t)stream));
    return T2S_OK;
}

int t2s_taco_att_bwd(const t2s_att_bwd* p, void* stream) {
    if (!p || !p->w_cur || !p->q || !p->pmem || !p->memory || !p->w_loc_conv || !p->w_loc_dense || !p->w_v ||
        !p->dw_carry || !p->dwc_carry || !p->d_q || !p->d_pmem || (!p->d_memory && !p->dctx_out) || !p->dD_part || !p->dK_part ||
        !p->dv_part || !p->dw_buf || !p->df_buf || !p->dq_part || p->B <= 0 || p->T <= 0)
        return T2S_EINVAL;
    static_assert(sizeof(t2s_att_bwd) == sizeof(AttBwdArgs), "t2s_att_bwd layout");
    AttBwdArgs a;
    memcpy(&a, p, sizeof(a));
    if (a.ctx || a.dw_carry_out || a.dwc_carry_out) {      // the one-launch form (the BPTT driver's): d_q stays in dq_part
        if (!t2s_att_bwd_fused_ok(a)) return T2S_EINVAL;
        T2S_CHECK_HIP(t2s_launch_att_bwd_fused(a, (hipStream_t)stream));
        return T2S_OK;
    }
    T2S_CHECK_HIP(t2s_launch_att_bwd(a, (hipStream_t)stream));
    return T2S_OK;
}

// The reversed decoder loop (BPTT through tacotron.py:355-393): per step, newest first,
//   decoder LSTMCell pointwise backward -> [W_ih | W_hh]^T dgates -> attention backward (3 launches) ->
//   attention LSTMCell pointwise backward (+ W_query^T d_q, fused) -> [W_ih | W_hh]^T dgates.
// Pure launch sequencing (pointer arithmetic on the caller's buffers), so the host never sits between the kernels.
int t2s_taco_bptt_steps(const t2s_taco_bptt* p, int t_hi, int t_lo, void* stream_) {
    if (!p || t_lo < 0 || t_hi <= t_lo || t_hi > p->T_out) return T2S_EINVAL;
    hipStream_t stream = (hipStream_t)stream_;
    const int B = p->B, T = p->T_out, Tin = p->T_in, P = p->prenet_dim, E = p->enc_dim, A = p->att_rnn_dim, D = p->dec_rnn_dim;
    const int ad = p->att_dim;
    const int KD = A + E + D, KA = P + E + A, DE = D + E;
    if (B <= 0 || Tin <= 0 || !p->W_dT || !p->W_aT || !p->w_query || !p->w_loc_conv || !p->w_loc_dense || !p->w_v ||
        !p->dec_gates_all || !p->dec_c_all || !p->att_gates_all || !p->att_c_all || !p->q_all || !p->wcum_all || !p->align ||
        !p->pmem || !p->memory || !p->d_hc || !p->out_d || !p->out_a || !p->dg_d || !p->dg_a || !p->dq_all ||
        !p->dc_d || !p->dc_a || !p->dw_c || !p->dwc_c || !p->d_pmem || !p->d_memory || !p->dD_part || !p->dK_part ||
        !p->dv_part || !p->dw_buf || !p->df_buf || !p->dq_part)
        return T2S_EINVAL;
    // Two chains, two streams.  The decoder-cell chain of step t (pointwise backward, then [W_ih | W_hh]^T dgates) needs only
    // the decoder-cell chain of step t+1: h_dec feeds the next decoder cell and the projection, never the attention.  The
    // attention / attention-cell chain of step t consumes its output out_d[t].  So the former runs ahead on a side stream and
    // signals one event per step; its ~27 us per step hide behind the ~78 us of the attention chain.
    // A third stream takes the last part of the attention backward (location-conv backward): it only feeds the NEXT step's
    // carries and the kernel gradient, so it runs beside this step's attention-cell backward and GEMM.
    static const bool two_streams = !getenv("T2S_BPTT_ONE_STREAM");
    // The location-conv backward stays on the caller's stream: with the matrix-core kernels it takes 10 us, while the event
    // record on the critical stream and the wait for a third stream cost ~7 + ~5 us per step on this runtime
    // (profiles/r03_taco_timeline_bwd.md; 128.7 -> 122.4 ms per train step).  T2S_BPTT_CONV_MAIN=0: the round-2 three-stream form.
    static const bool conv_main = !getenv("T2S_BPTT_CONV_MAIN") || atoi(getenv("T2S_BPTT_CONV_MAIN"));
    int device = 0;
    T2S_CHECK_HIP(hipGetDevice(&device));
    if (device < 0 || device >= kMaxDevices) return T2S_EINVAL;
    BpttStreams& S = g_bptt[device];
    std::unique_lock<std::mutex> lock(S.mu);
    if (two_streams) T2S_CHECK_HIP(bptt_streams_init(S));
    hipStream_t side = S.side, side2 = S.side2;
    hipEvent_t ev_main = S.ev_main, ev_side = S.ev_side, ev_energy = S.ev_energy, ev_conv = S.ev_conv;
    // Whatever happens below (including a failed launch), the caller's stream is made to wait for everything already
    // enqueued on the two helper streams before this call returns: the caller may recycle the scratch buffers as soon as
    // ITS stream has passed this point.
    struct Join {
        BpttStreams& S; hipStream_t stream; bool on;
        ~Join() {
            if (!on) return;
            if (hipEventRecord(S.ev_join, S.side) == hipSuccess) (void)hipStreamWaitEvent(stream, S.ev_join, 0);
            if (hipEventRecord(S.ev_join, S.side2) == hipSuccess) (void)hipStreamWaitEvent(stream, S.ev_join, 0);
        }
    } join{S, stream, two_streams};
    bool conv_pending = false;
    hipStream_t dstream = two_streams ? side : stream;
    if (two_streams) {                                   // everything enqueued so far (d_hc, the saves) precedes the side chain
        T2S_CHECK_HIP(hipEventRecord(ev_main, stream));
        T2S_CHECK_HIP(hipStreamWaitEvent(side, ev_main, 0));
    }
    // The decoder-cell chain never waits for the attention chain, so it is enqueued `chunk` steps at a time and the caller's
    // stream waits once per chunk: a cross-stream wait costs ~6 us on the critical stream (profiles/r03_taco_timeline_bwd_fused.md).
    static const int chunk_env = getenv("T2S_BPTT_CHUNK") ? atoi(getenv("T2S_BPTT_CHUNK")) : 16;
    const int chunk = two_streams ? (chunk_env > 0 ? chunk_env : 1) : 1;
    // Two streams, 9+ items: the transposed decoder-cell GEMM [KD = A + E + D rows] x [4 D] of every step is
    // split by consumer - the D rows of d h_dec(t-1), which the next decoder-cell step needs, per step; the A + E rows of d h_att /
    // d ctx, which only the attention chain reads (a chunk of steps later), as ONE launch over the chunk's (steps x batch) items
    // (-1 % of the train step: 88.45 / 88.2 / 87.8 / 88.2 against 89.4 / 89.7 / 89.5 / 89.5 ms in three same-box sessions,
    // profiles/r04_taco_bptt_streams_ab.txt, r04_taco_bptt_narrow_ab.txt, r04_taco_chunk_gemm_ab.txt.  T2S_BPTT_SPLIT_ROWS=0: off)
    static const bool want_split_rows = !(getenv("T2S_BPTT_SPLIT_ROWS") && atoi(getenv("T2S_BPTT_SPLIT_ROWS")) == 0);
    // T2S_BPTT_SIDE_NARROW=1: the decoder-cell chain's GEMM (helper stream) on the 96 KB ring, so that the attention chain's
    // att_bwd_fused workgroups (61 KB of LDS) can share its CUs instead of queueing behind it.  Measured NEGATIVE: 96.6 / 96.3
    // against 89.5 / 90.0 ms per train step (profiles/r04_taco_bptt_narrow_ab.txt) - the 64-byte fragment rows cost the GEMM more
    // CU time than the sharing returns; the loop is the sum of its kernels' CU time.  Off.
    static const bool want_narrow = getenv("T2S_BPTT_SIDE_NARROW") && atoi(getenv("T2S_BPTT_SIDE_NARROW")) != 0;
    const bool side_narrow = want_narrow && two_streams;
    const bool split_rows = want_split_rows && two_streams && chunk > 1 && B > 8 && ((A + E) & 15) == 0 && (D & 15) == 0;
    // Paced helper chain (T2S_BPTT_PACED=1; needs att_xbuf, whose tail holds the word): the decoder-cell chain's step m (counted from
    // t_hi - 1) is released by a word the attention backward's launch of step m - chunk stores as it starts, and its GEMM takes the
    // 96 KB ring - so that it runs BESIDE that launch (61 KB of LDS) instead of holding the CUs the chain's next launch needs.
    // Built after the forward's pacing paid (section 5b of DESIGN.md) and measured NEGATIVE here: 77.8 / 78.0 ms per train step with
    // the 96 KB ring, 78.9 / 78.7 with the 144 KB one, against 74.2 / 73.9 unpaced (profiles/r04_bptt_paced_ab.txt; every gradient test
    // green with it on).  The forward pairs a GEMM with a 14 KB kernel; here the partner is a 61 KB / 185-VGPR kernel and the GEMM has to
    // take the slower ring to fit beside it.  Off.
    static const bool want_bpaced = getenv("T2S_BPTT_PACED") && atoi(getenv("T2S_BPTT_PACED")) != 0;
    static const bool bpaced_narrow = !(getenv("T2S_BPTT_PACED_NARROW") && atoi(getenv("T2S_BPTT_PACED_NARROW")) == 0);
    const bool bpaced = want_bpaced && two_streams && chunk > 1 && p->att_xbuf && p->ctx_all && p->dw_c2 && p->dwc_c2 && conv_main &&
                        Tin <= 512 && ad == 128;
    unsigned* const pace_word = bpaced ? (unsigned*)((unsigned long long*)p->att_xbuf + (size_t)B * ((Tin + 31) / 32) * ad + 1) : nullptr;
    unsigned long long* const pace_err = bpaced ? (unsigned long long*)p->att_xbuf + (size_t)B * ((Tin + 31) / 32) * ad + 2 : nullptr;
    for (int tc = t_hi - 1; tc >= t_lo; tc -= chunk) {
    const int tl = tc - chunk + 1 > t_lo ? tc - chunk + 1 : t_lo;
    for (int t = tc; t >= tl; --t) {
        const bool nxt = t + 1 < T;
        if (bpaced && t_hi - 1 - t >= chunk)
            T2S_CHECK_HIP(t2s_launch_pace_wait(pace_word, (unsigned)(t_hi - 1 - t - chunk) + 1u, pace_err, dstream));
        // decoder LSTMCell: dh = d[h_dec] from the projection + from step t+1's decoder cell (through W_hh)
        LstmBwdArgs cd;
        cd.dh1 = p->d_hc + (size_t)t * B * DE; cd.s1 = DE;
        cd.dh2 = nxt ? p->out_d + (size_t)(t + 1) * B * KD + A + E : nullptr; cd.s2 = KD;
        cd.dh3 = nullptr; cd.s3 = 0;
        cd.drop_mask = p->dec_drop ? p->dec_drop + (size_t)t * B * D : nullptr; cd.drop_scale = p->dec_drop_scale;
        cd.gates = p->dec_gates_all + (size_t)t * B * 4 * D; cd.c_new = p->dec_c_all + (size_t)t * B * D;
        cd.c_prev = t > 0 ? p->dec_c_all + (size_t)(t - 1) * B * D : nullptr;
        cd.dc_carry = p->dc_d; cd.dgates = p->dg_d + (size_t)t * B * 4 * D; cd.B = B; cd.H = D;
        cd.wq = nullptr; cd.dq = nullptr; cd.q_dim = 0; cd.dq_part = nullptr; cd.dq_nchunk = 0; cd.dq_out = nullptr;
        T2S_CHECK_HIP(t2s_launch_lstm_cell_bwd(cd, dstream));
        GemvArgs g;
        memset(&g, 0, sizeof(g));
        g.W1 = p->W_dT; g.ld1 = 4 * D; g.k1 = 4 * D; g.x1 = cd.dgates; g.n1 = 4 * D; g.sx1 = 4 * D;
        g.y = p->out_d + (size_t)t * B * KD; g.sy_item = KD; g.sy_row = 1; g.rows = KD; g.items = B; g.mask_scale = 1.f;
        g.narrow_ring = (side_narrow || (bpaced && bpaced_narrow)) ? 1 : 0;
        static const bool side_full = getenv("T2S_BPTT_SIDE_FULL") && atoi(getenv("T2S_BPTT_SIDE_FULL")) != 0;
        // (A/B: 32 items per workgroup on the helper chain's per-step GEMM - fewer, longer workgroups: 77.1 / 77.6 against
        // 73.5 / 73.7 ms per train step, profiles/r04_bptt_side_full_ab.txt; off)
        g.no_half = (side_full && two_streams) ? 1 : 0;
        if (split_rows) {
            // only the rows the NEXT decoder-cell step reads (d h_dec(t-1) = rows A + E .. KD of [W_ih | W_hh]^T dgates) stay per step;
            // the d h_att / d ctx rows, which the attention chain reads a chunk later, are one GEMM over the chunk's items below
            g.W1 = p->W_dT + (size_t)(A + E) * 4 * D; g.y += A + E; g.rows = D;
        }
        T2S_CHECK_HIP(t2s_launch_gemv(g, dstream));
    }
    if (split_rows) {
        GemvArgs g;
        memset(&g, 0, sizeof(g));
        g.W1 = p->W_dT; g.ld1 = 4 * D; g.k1 = 4 * D; g.x1 = p->dg_d + (size_t)tl * B * 4 * D; g.n1 = 4 * D; g.sx1 = 4 * D;
        g.y = p->out_d + (size_t)tl * B * KD; g.sy_item = KD; g.sy_row = 1; g.rows = A + E; g.items = (tc - tl + 1) * B;
        g.mask_scale = 1.f;
        T2S_CHECK_HIP(t2s_launch_gemv(g, dstream));
    }
    if (two_streams) {       // out_d[tl .. tc] are ready: the wait below binds to THIS record, so one event object serves every chunk
        T2S_CHECK_HIP(hipEventRecord(ev_side, side));
        T2S_CHECK_HIP(hipStreamWaitEvent(stream, ev_side, 0));
    }
    for (int t = tc; t >= tl; --t) {
        const bool nxt = t + 1 < T;
        GemvArgs g;
        // attention: d_ctx = decoder-cell input part + projection part + step t+1's attention-cell input part
        AttBwdArgs ab;
        memset(&ab, 0, sizeof(ab));
        ab.dctx1 = p->out_d + (size_t)t * B * KD + A; ab.sc1 = KD;
        ab.dctx2 = p->d_hc + (size_t)t * B * DE + D; ab.sc2 = DE;
        ab.dctx3 = nxt ? p->out_a + (size_t)(t + 1) * B * KA + P : nullptr; ab.sc3 = KA;
        ab.w_cur = p->align + (size_t)t * Tin; ab.s_wcur = (long)p->T_cap * Tin;
        ab.w_prev = t > 0 ? p->align + (size_t)(t - 1) * Tin : nullptr; ab.s_wprev = (long)p->T_cap * Tin;
        ab.wc_prev = t > 0 ? p->wcum_all + (size_t)(t - 1) * B * Tin : nullptr; ab.s_wcprev = Tin;
        ab.q = p->q_all + (size_t)t * B * ad; ab.pmem = p->pmem; ab.memory = p->memory; ab.lengths = p->lengths;
        ab.w_loc_conv = p->w_loc_conv; ab.w_loc_dense = p->w_loc_dense; ab.w_v = p->w_v;
        ab.dw_carry = p->dw_c; ab.dwc_carry = p->dwc_c; ab.d_q = p->dq_all + (size_t)t * B * ad; ab.d_pmem = p->d_pmem;
        ab.d_memory = p->d_memory; ab.dD_part = p->dD_part; ab.dK_part = p->dK_part; ab.dv_part = p->dv_part;
        ab.dw_buf = p->dw_buf; ab.df_buf = p->df_buf; ab.dq_part = p->dq_part;
        if (p->dctx_all) { ab.dctx_out = p->dctx_all + (size_t)t * B * E; ab.d_memory = nullptr; }
        ab.B = B; ab.T = Tin; ab.att_dim = ad; ab.enc_dim = E; ab.loc_f = p->loc_filters; ab.loc_ks = p->loc_kernel;
        bool fused = false;
        if (p->ctx_all && p->dw_c2 && p->dwc_c2 && (!two_streams || conv_main)) {
            // one launch: reads the carries of parity t, writes those of parity t - 1; d_q stays in per-chunk partials
            ab.ctx = p->ctx_all + (size_t)t * p->s_ctx_step; ab.s_ctx = p->s_ctx_item;
            if (t & 1) { ab.dw_carry = p->dw_c2; ab.dwc_carry = p->dwc_c2; ab.dw_carry_out = p->dw_c; ab.dwc_carry_out = p->dwc_c; }
            else { ab.dw_carry_out = p->dw_c2; ab.dwc_carry_out = p->dwc_c2; }
            fused = t2s_att_bwd_fused_ok(ab);
            if (!fused) {                           // shape not covered (or T2S_ATTB_FUSED=0): the three-launch form, carries in place
                ab.ctx = nullptr; ab.dw_carry = p->dw_c; ab.dwc_carry = p->dwc_c; ab.dw_carry_out = nullptr; ab.dwc_carry_out = nullptr;
            }
        }
        // attention LSTMCell: dh = from the decoder cell input + from the query + from step t+1's attention cell
        LstmBwdArgs ca;
        ca.dh1 = p->out_d + (size_t)t * B * KD; ca.s1 = KD;
        ca.dh2 = nxt ? p->out_a + (size_t)(t + 1) * B * KA + P + E : nullptr; ca.s2 = KA;
        ca.dh3 = nullptr; ca.s3 = 0;
        ca.wq = p->w_query; ca.dq = ab.d_q; ca.q_dim = ad;           // + W_query^T d_q, fused
        ca.dq_part = nullptr; ca.dq_nchunk = 0; ca.dq_out = nullptr;
        if (fused || (two_streams && !conv_main)) {       // d_q is not folded yet: sum the per-chunk partials here
            ca.dq_part = p->dq_part; ca.dq_nchunk = (Tin + 31) / 32;
            if (fused) ca.dq_out = ab.d_q;
        }
        ca.drop_mask = p->att_drop ? p->att_drop + (size_t)t * B * A : nullptr; ca.drop_scale = p->att_drop_scale;
        ca.gates = p->att_gates_all + (size_t)t * B * 4 * A; ca.c_new = p->att_c_all + (size_t)t * B * A;
        ca.c_prev = t > 0 ? p->att_c_all + (size_t)(t - 1) * B * A : nullptr;
        ca.dc_carry = p->dc_a; ca.dgates = p->dg_a + (size_t)t * B * 4 * A; ca.B = B; ca.H = A;
        // ... folded into the one-launch attention backward where it runs (t2s_taco_bptt::att_xbuf; T2S_BPTT_FOLD_CELL=0: a launch of its own)
        static const bool want_fold = !(getenv("T2S_BPTT_FOLD_CELL") && atoi(getenv("T2S_BPTT_FOLD_CELL")) == 0);
        const bool fold_cell = fused && want_fold && p->att_xbuf && Tin <= 512 && ad == 128;
        if (fused) {
            AttBwdFoldArgs fa;
            memset(&fa, 0, sizeof(fa));
            if (fold_cell) { fa.cell = ca; fa.xbuf = (unsigned long long*)p->att_xbuf; fa.tag = (unsigned)t + 1u; }
            if (bpaced) { fa.sig_ptr = pace_word; fa.sig_val = (unsigned)(t_hi - 1 - t) + 1u; }
            T2S_CHECK_HIP(t2s_launch_att_bwd_fused(ab, stream, (fold_cell || bpaced) ? &fa : nullptr));
        } else if (two_streams && conv_main) {
            if (bpaced) T2S_CHECK_HIP(t2s_launch_pace_signal(pace_word, (unsigned)(t_hi - 1 - t) + 1u, stream));
            T2S_CHECK_HIP(t2s_launch_att_bwd(ab, stream));
        } else if (two_streams) {
            if (conv_pending) T2S_CHECK_HIP(hipStreamWaitEvent(stream, ev_conv, 0));   // carries of step t+1 are in place
            T2S_CHECK_HIP(t2s_launch_att_bwd_front(ab, stream));
            T2S_CHECK_HIP(hipEventRecord(ev_energy, stream));
            T2S_CHECK_HIP(hipStreamWaitEvent(side2, ev_energy, 0));
            T2S_CHECK_HIP(t2s_launch_att_bwd_conv(ab, side2));
            T2S_CHECK_HIP(hipEventRecord(ev_conv, side2));
            conv_pending = true;
        } else {
            T2S_CHECK_HIP(t2s_launch_att_bwd(ab, stream));
        }
        if (!fold_cell) T2S_CHECK_HIP(t2s_launch_lstm_cell_bwd(ca, stream));
        memset(&g, 0, sizeof(g));
        g.W1 = p->W_aT; g.ld1 = 4 * A; g.k1 = 4 * A; g.x1 = ca.dgates; g.n1 = 4 * A; g.sx1 = 4 * A;
        g.y = p->out_a + (size_t)t * B * KA; g.sy_item = KA; g.sy_row = 1; g.rows = KA; g.items = B; g.mask_scale = 1.f;
        T2S_CHECK_HIP(t2s_launch_gemv(g, stream));
    }
    }
    if (conv_pending) T2S_CHECK_HIP(hipStreamWaitEvent(stream, ev_conv, 0));       // dq_all, carries, dK complete for the caller
    return T2S_OK;
}

int t2s_taco_loss(const float* mel, const float* post, const float* target, size_t n_mel, const float* gate,
                  const float* gate_target, size_t n_gate, float* d_mel, float* d_post, float* d_gate, void* partial,
                  float* out, void* stream) {
    if (!mel || !post || !target || !gate || !gate_target || !partial || !out || n_mel == 0 || n_gate == 0) return T2S_EINVAL;
    T2S_CHECK_HIP(t2s_launch_taco_loss(mel, post, target, n_mel, gate, gate_target, n_gate, d_mel, d_post, d_gate,
                                       (double*)partial, out, (hipStream_t)stream));
    return T2S_OK;
}

int t2s_waveglow_loss(const float* z, size_t n_z, const float* const* log_s, const size_t* n_log_s, int n_flows,
                      const float* log_det, float sigma, float* d_z, void* partial, float* out, void* stream) {
    if (!z || !log_s || !n_log_s || !log_det || !partial || !out || n_z == 0 || n_flows <= 0 || n_flows > 16 || !(sigma > 0.f))
        return T2S_EINVAL;
    for (int k = 0; k < n_flows; ++k)
        if (!log_s[k]) return T2S_EINVAL;
    T2S_CHECK_HIP(t2s_launch_waveglow_loss(z, n_z, log_s, n_log_s, n_flows, log_det, sigma, d_z, (double*)partial, out,
                                           (hipStream_t)stream));
    return T2S_OK;
}

int t2s_bn_bwd(const t2s_bn_bwd_args* p, void* partial, void* stream) {
    if (!p || !partial || ((uintptr_t)partial & 7) || !p->x || !p->mean || !p->var || !p->gamma || !p->beta || !p->dgamma || !p->dbeta ||
        !p->dx_hi || !p->dx_lo || (!p->dout_f32 && (!p->dout_hi || !p->dout_lo)) || p->B <= 0 || p->C <= 0 || p->T <= 0)
        return T2S_EINVAL;
    static_assert(sizeof(t2s_bn_bwd_args) == sizeof(BnBwdArgs), "t2s_bn_bwd_args layout");
    BnBwdArgs a;
    memcpy(&a, p, sizeof(a));
    T2S_CHECK_HIP(t2s_launch_bn_bwd(a, (double*)partial, (hipStream_t)stream));
    return T2S_OK;
}

int t2s_taco_encoder_lstm_bwd(const float* d_out, const float* out, const float* gates_save, const float* c_save,
                              const float* whh_fwd, const float* whh_rev, const int* lengths, float* dgx, float* hprev,
                              int B, int T, int H, int T_out, void* stream) {
    if (!d_out || !out || !gates_save || !c_save || !whh_fwd || !whh_rev || !dgx || !hprev || B <= 0 || T <= 0 || H != 256 ||
        T_out <= 0 || T_out > T)
        return T2S_EINVAL;
    T2S_CHECK_HIP(t2s_launch_lstm_seq_bwd(d_out, out, gates_save, c_save, whh_fwd, whh_rev, lengths, dgx, hprev, B, T, H,
                                          T_out, (hipStream_t)stream));
    return T2S_OK;
}
int t2s_taco_encoder_lstm_bwd_split(const float* d_out, const float* out, const float* gates_save, const float* c_save,
                                    const float* whh_fwd, const float* whh_rev, const int* lengths, float* dgx, float* hprev,
                                    int B, int T, int H, int T_out, void* xbuf, unsigned epoch, void* stream) {
    if (!d_out || !out || !gates_save || !c_save || !whh_fwd || !whh_rev || !dgx || !hprev || !xbuf || B <= 0 || T <= 0 ||
        T >= 4095 || H != 256 || T_out <= 0 || T_out > T || ((uintptr_t)xbuf & 7))
        return T2S_EINVAL;
    T2S_CHECK_HIP(t2s_launch_lstm_seq_bwd_split(d_out, out, gates_save, c_save, whh_fwd, whh_rev, lengths, dgx, hprev, B, T, T_out,
                                                (unsigned long long*)xbuf, epoch, (hipStream_t)stream));
    return T2S_OK;
}
int t2s_rows_to_planes(const float* x, int B, int T, int C, int Lp, int halo, void* X_hi, void* X_lo, void* stream) {
    if (!x || !X_hi || !X_lo || B <= 0 || T <= 0 || C <= 0 || Lp < t2s_plane_rows(T, halo)) return T2S_EINVAL;
    T2S_CHECK_HIP(t2s_launch_rows_to_planes(x, B, T, C, Lp, halo, (u16*)X_hi, (u16*)X_lo, (hipStream_t)stream));
    return T2S_OK;
}
int t2s_embedding_grad(const long* ids, const void* D_hi, const void* D_lo, int B, int T, int E, int V, int Lp, int halo,
                       float* d_emb, void* stream) {
    if (!ids || !D_hi || !D_lo || !d_emb || B <= 0 || T <= 0 || E <= 0 || V <= 0) return T2S_EINVAL;
    T2S_CHECK_HIP(t2s_launch_embedding_grad(ids, (const u16*)D_hi, (const u16*)D_lo, B, T, E, V, Lp, halo, d_emb,
                                            (hipStream_t)stream));
    return T2S_OK;
}

int t2s_sum_axis0(const float* in, int n0, int n, float* out, void* stream) {
    if (!in || !out || n0 <= 0 || n <= 0) return T2S_EINVAL;
    T2S_CHECK_HIP(t2s_launch_sum_axis0(in, n0, n, out, (hipStream_t)stream));
    return T2S_OK;
}
int t2s_scale_by_scalar(const float* in, size_t n, const float* scalar, float mul, float* out, void* stream) {
    if (!scalar || !out || n == 0) return T2S_EINVAL;
    T2S_CHECK_HIP(t2s_launch_scale_by_scalar(in, n, scalar, mul, out, (hipStream_t)stream));
    return T2S_OK;
}
int t2s_add3(const float* a, const float* b, const float* c, size_t n, float* out, void* stream) {
    if (!a || !out || n == 0) return T2S_EINVAL;
    T2S_CHECK_HIP(t2s_launch_add3(a, b, c, n, out, (hipStream_t)stream));
    return T2S_OK;
}

int t2s_planes_to_f32(const void* X_hi, const void* X_lo, int B, int C, int L, int Lp, int halo, float* out,
                      int accumulate, void* stream) {
    if (!X_hi || !X_lo || !out || B <= 0 || C <= 0 || L <= 0) return T2S_EINVAL;
    T2S_CHECK_HIP(t2s_launch_planes_to_f32((const u16*)X_hi, (const u16*)X_lo, B, C, L, Lp, halo, out, accumulate,
                                           (hipStream_t)stream));
    return T2S_OK;
}

}  // extern "C"
