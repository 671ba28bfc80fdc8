// extern "C" boundary of the Tacotron-2 backward kernels.
#include "../../include/t2s_hip.h"
#include "t2s_kernels.h"
#include "taco_bwd_ops.h"

#include <string.h>

extern "C" int t2s_internal_fail_hip(int e);
#define T2S_CHECK_HIP(expr)                                          \
    do {                                                             \
        hipError_t _e = (expr);                                      \
        if (_e != hipSuccess) return t2s_internal_fail_hip((int)_e); \
    } while (0)

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

int t2s_lstm_cell_bwd(const float* dh1, long s1, const float* dh2, long s2, const float* dh3, long s3,
                      const unsigned char* drop_mask, float drop_scale, const float* gates, const float* c_new,
                      const float* c_prev, float* dc_carry, float* dgates, int B, int H, void* stream) {
    if (!gates || !c_new || !dc_carry || !dgates || B <= 0 || H <= 0) return T2S_EINVAL;
    LstmBwdArgs a;
    a.dh1 = dh1; a.s1 = s1; a.dh2 = dh2; a.s2 = s2; a.dh3 = dh3; a.s3 = s3;
    a.drop_mask = drop_mask; a.drop_scale = drop_scale; a.gates = gates; a.c_new = c_new; a.c_prev = c_prev;
    a.dc_carry = dc_carry; a.dgates = dgates; a.B = B; a.H = H;
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
        !p->dw_carry || !p->dwc_carry || !p->d_q || !p->d_pmem || !p->d_memory || !p->dD_part || !p->dK_part ||
        !p->dv_part || p->B <= 0 || p->T <= 0 || p->T > 256)
        return T2S_EINVAL;
    static_assert(sizeof(t2s_att_bwd) == sizeof(AttBwdArgs), "t2s_att_bwd layout");
    AttBwdArgs a;
    memcpy(&a, p, sizeof(a));
    T2S_CHECK_HIP(t2s_launch_att_bwd(a, (hipStream_t)stream));
    return T2S_OK;
}

int t2s_bn_bwd(const t2s_bn_bwd_args* p, void* stream) {
    if (!p || !p->x || !p->mean || !p->var || !p->gamma || !p->beta || !p->dgamma || !p->dbeta || !p->dx_hi ||
        !p->dx_lo || (!p->dout_f32 && (!p->dout_hi || !p->dout_lo)) || p->B <= 0 || p->C <= 0 || p->T <= 0)
        return T2S_EINVAL;
    static_assert(sizeof(t2s_bn_bwd_args) == sizeof(BnBwdArgs), "t2s_bn_bwd_args layout");
    BnBwdArgs a;
    memcpy(&a, p, sizeof(a));
    T2S_CHECK_HIP(t2s_launch_bn_bwd(a, (hipStream_t)stream));
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
int t2s_add3(const float* a, const float* b, const float* c, size_t n, float* out, void* stream) {
    if (!a || !b || !out || n == 0) return T2S_EINVAL;
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
