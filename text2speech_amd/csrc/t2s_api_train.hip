// extern "C" boundary of the training kernels.
#include "../../include/t2s_hip.h"
#include "t2s_kernels.h"
#include "train_ops.h"

#include <math.h>
#include <string.h>

extern "C" int t2s_internal_fail_hip(int e);

#define T2S_CHECK_HIP(expr)                                          \
    do {                                                             \
        hipError_t _e = (expr);                                      \
        if (_e != hipSuccess) return t2s_internal_fail_hip((int)_e); \
    } while (0)

static inline bool al16(const void* p) { return ((uintptr_t)p & 15) == 0; }
static inline int cdiv(int a, int b) { return (a + b - 1) / b; }
#include <stdlib.h>
// The accumulate / gate-backward GEMMs of the training backward: 256-row tiles on the ping-pong schedule (csrc/gate_gemm_pp.hip)
// once they give at least ~100 workgroups - M = 512 at 8 x 16000 is 128, half the chip, and the rest is taken by the
// weight-gradient stream that runs beside them - else the lockstep kernels on 128-row tiles (twice the workgroups).
// T2S_BWD_PP256=0 restores the round-2 choice for A/B runs.
static bool bwd_pp256(const ConvGemmArgs& a, int rows) {
    static const int on = getenv("T2S_BWD_PP256") ? atoi(getenv("T2S_BWD_PP256")) : 1;
    return on && t2s_pp_shape_ok(a) && (long)cdiv(rows, 256) * a.n_ttiles * a.B >= 100;
}

static int planes_ok(const void* a, const void* b) { return a && b && al16(a) && al16(b); }

// Can the backward GEMM with `rows` output rows over B x L columns take PERM_PAIR8-packed operands (16-byte epilogue pieces)?  Only
// the 256-row ping-pong kernels have that epilogue: the grid rule of bwd_pp256 above, whole 32-row groups.  T2S_BWD_PAIR8=0: never.
extern "C" int t2s_wg_bwd_pair8_ok(int B, int rows, int L) {
    static const int on = getenv("T2S_BWD_PAIR8") ? atoi(getenv("T2S_BWD_PAIR8")) : 1;
    static const int pp = getenv("T2S_BWD_PP256") ? atoi(getenv("T2S_BWD_PP256")) : 1;
    if (!on || !pp || B <= 0 || rows <= 0 || L <= 0 || rows % 32) return 0;
    return (long)cdiv(rows, 256) * cdiv(L, 256) * B >= 100 ? 1 : 0;
}

// rows [row0, row0+1) scale kernel lives in waveglow_ops.hip's weightnorm_small (scale-only form below)
__global__ void weightnorm_scale_kernel(const float* v, const float* g, int O, int K, float* scale) {
    const int o = blockIdx.x * blockDim.x + threadIdx.x;
    if (o >= O) return;
    if (!g) { scale[o] = 1.f; return; }
    float ss = 0.f;
    for (int k = 0; k < K; ++k) ss += v[(size_t)o * K + k] * v[(size_t)o * K + k];
    scale[o] = g[o] / sqrtf(ss);
}

extern "C" {

int t2s_wg_in_cond_gate_train(const void* A_hi, const void* A_lo, const float* bias, const void* X_hi, const void* X_lo,
                              const void* S_hi, const void* S_lo, void* acts_hi, void* acts_lo, void* T_hi, void* T_lo,
                              void* G_hi, void* G_lo, int B, int C, int n_cond, int taps, int dilation, int L, int Lp,
                              int halo, int Mpad, void* stream) {
    if (!planes_ok(A_hi, A_lo) || !planes_ok(X_hi, X_lo) || !planes_ok(acts_hi, acts_lo) || !bias) return T2S_EINVAL;
    if (!planes_ok(G_hi, G_lo) || ((T_hi || T_lo) && !planes_ok(T_hi, T_lo))) return T2S_EINVAL;      // tanh planes are optional
    if (n_cond > 0 && !planes_ok(S_hi, S_lo)) return T2S_EINVAL;
    if (B <= 0 || L <= 0 || C <= 0 || C % 4 || taps <= 0 || !(taps & 1) || dilation <= 0) return T2S_EINVAL;
    if ((taps / 2) * dilation > halo || Lp != t2s_plane_rows(L, halo)) return T2S_EINVAL;
    if (Mpad % 256 || Mpad < cdiv(C, 128) * 256 || !al16(bias)) return T2S_EINVAL;
    ConvGemmArgs a;
    memset(&a, 0, sizeof(a));
    a.A_hi = (const u16*)A_hi; a.A_lo = (const u16*)A_lo;
    a.X_hi = (const u16*)X_hi; a.X_lo = (const u16*)X_lo;
    a.S_hi = (const u16*)S_hi; a.S_lo = (const u16*)S_lo;
    a.bias = bias; a.O_hi = (u16*)acts_hi; a.O_lo = (u16*)acts_lo;
    a.T_hi = (u16*)T_hi; a.T_lo = (u16*)T_lo; a.G_hi = (u16*)G_hi; a.G_lo = (u16*)G_lo;
    a.xc = cdiv(C, 32); a.sc = cdiv(n_cond, 32); a.oc = cdiv(C, 32); a.tc = a.oc;
    a.taps = taps; a.dil = dilation;
    a.nk_x = taps * a.xc; a.nk = a.nk_x + a.sc;
    a.Mpad = Mpad; a.Lp = Lp; a.halo = halo; a.L = L; a.B = B;
    a.n_mtiles = cdiv(C, 128); a.n_ttiles = cdiv(L, 256);
    a.C = C;
    T2S_CHECK_HIP(t2s_launch_conv_gemm(a, EPI_GATE, (hipStream_t)stream));
    return T2S_OK;
}

// gate GEMM tile height: the library's own decision (csrc/t2s_api.hip)
extern "C" int t2s_wg_gate_tile_rows(int B, int C, int L);

int t2s_wg_in_cond_gate_fold_train(const void* A_hi, const void* A_lo, const float* bias, const void* X_hi, const void* X_lo,
                                   const void* S_hi, const void* S_lo, void* acts_hi, void* acts_lo, void* G_hi, void* G_lo,
                                   int act_bchunks, const void* fold_A, float* fold_acc, int fold_init, int B, int C, int n_cond,
                                   int taps, int dilation, int L, int Lp, int halo, int Mpad, void* stream) {
    if (!planes_ok(A_hi, A_lo) || !planes_ok(X_hi, X_lo) || !planes_ok(acts_hi, acts_lo) || !planes_ok(G_hi, G_lo) || !bias)
        return T2S_EINVAL;
    if (!fold_A || !fold_acc || !al16(fold_A) || C % 16) return T2S_EINVAL;
    if (n_cond > 0 && !planes_ok(S_hi, S_lo)) return T2S_EINVAL;
    if (B <= 0 || L <= 0 || C <= 0 || taps <= 0 || !(taps & 1) || dilation <= 0) return T2S_EINVAL;
    if ((taps / 2) * dilation > halo || Lp != t2s_plane_rows(L, halo)) return T2S_EINVAL;
    if (Mpad % 256 || Mpad < cdiv(C, 128) * 256 || !al16(bias)) return T2S_EINVAL;
    if (act_bchunks != 0 && act_bchunks < cdiv(C, 32)) return T2S_EINVAL;
    ConvGemmArgs a;
    memset(&a, 0, sizeof(a));
    a.A_hi = (const u16*)A_hi; a.A_lo = (const u16*)A_lo;
    a.X_hi = (const u16*)X_hi; a.X_lo = (const u16*)X_lo;
    a.S_hi = (const u16*)S_hi; a.S_lo = (const u16*)S_lo;
    a.bias = bias; a.O_hi = (u16*)acts_hi; a.O_lo = (u16*)acts_lo;
    a.G_hi = (u16*)G_hi; a.G_lo = (u16*)G_lo;
    a.fold_A = (const u16*)fold_A; a.fold_acc = fold_acc; a.fold_init = fold_init;
    a.xc = cdiv(C, 32); a.sc = cdiv(n_cond, 32);
    a.oc = act_bchunks ? act_bchunks : cdiv(C, 32); a.tc = a.oc;      // oc = batch stride of the acts / sigmoid planes
    a.taps = taps; a.dil = dilation;
    a.nk_x = taps * a.xc; a.nk = a.nk_x + a.sc;
    a.Mpad = Mpad; a.Lp = Lp; a.halo = halo; a.L = L; a.B = B;
    const int rows = t2s_wg_gate_tile_rows(B, C, L);          // fold_acc holds t2s_wg_gate_fold_slots(B, C, L) slots
    a.n_mtiles = cdiv(C, rows / 2); a.n_ttiles = cdiv(L, 256);
    a.C = C;
    T2S_CHECK_HIP(t2s_launch_conv_gemm(a, EPI_GATE, (hipStream_t)stream, rows));
    return T2S_OK;
}

int t2s_wg_res_only_train(const void* A_hi, const void* A_lo, const float* bias, const void* acts_hi, const void* acts_lo,
                          int act_bchunks, const void* R_hi, const void* R_lo, void* X_hi, void* X_lo, int B, int C, int L, int Lp,
                          int halo, int Mpad, int pair8, void* stream) {
    if (!planes_ok(A_hi, A_lo) || !planes_ok(acts_hi, acts_lo) || !planes_ok(X_hi, X_lo) || !planes_ok(R_hi, R_lo) || !bias)
        return T2S_EINVAL;
    if (B <= 0 || L <= 0 || C <= 0 || C % 4 || Lp != t2s_plane_rows(L, halo) || Mpad % 256 || Mpad < C || !al16(bias)) return T2S_EINVAL;
    if ((act_bchunks != 0 && act_bchunks < cdiv(C, 32)) || (pair8 && C % 32)) return T2S_EINVAL;
    ConvGemmArgs a;
    memset(&a, 0, sizeof(a));
    a.A_hi = (const u16*)A_hi; a.A_lo = (const u16*)A_lo;
    a.X_hi = (const u16*)acts_hi; a.X_lo = (const u16*)acts_lo; a.xbs = act_bchunks;
    a.bias = bias; a.O_hi = (u16*)X_hi; a.O_lo = (u16*)X_lo;
    a.R_hi = (const u16*)R_hi; a.R_lo = (const u16*)R_lo;
    a.xc = cdiv(C, 32); a.sc = 0; a.oc = cdiv(C, 32);
    a.taps = 1; a.dil = 1; a.nk_x = a.xc; a.nk = a.xc;
    a.Mpad = Mpad; a.Lp = Lp; a.halo = halo; a.L = L; a.B = B;
    a.n_mtiles = cdiv(C, 128); a.n_ttiles = cdiv(L, 256);
    a.C = 0; a.n_res = C;
    a.pair8 = pair8 ? 1 : 0;
    T2S_CHECK_HIP(t2s_launch_conv_gemm(a, EPI_RESSKIP, (hipStream_t)stream, 128));
    return T2S_OK;
}

int t2s_wg_skip_sum(const void* A_hi, const void* A_lo, const float* bias, const void* acts_hi, const void* acts_lo,
                    int n_k_chunks, int act_bchunks, float* skip, int B, int C, int L, int Lp, int halo, int Mpad, void* stream) {
    if (!planes_ok(A_hi, A_lo) || !planes_ok(acts_hi, acts_lo) || !bias || !skip || !al16(skip) || !al16(bias)) return T2S_EINVAL;
    if (B <= 0 || L <= 0 || C <= 0 || C % 4 || n_k_chunks <= 0 || Lp != t2s_plane_rows(L, halo) || Mpad % 256 || Mpad < C) return T2S_EINVAL;
    if (act_bchunks != 0 && act_bchunks < n_k_chunks) return T2S_EINVAL;
    ConvGemmArgs a;
    memset(&a, 0, sizeof(a));
    a.A_hi = (const u16*)A_hi; a.A_lo = (const u16*)A_lo;
    a.X_hi = (const u16*)acts_hi; a.X_lo = (const u16*)acts_lo; a.xbs = act_bchunks;
    a.bias = bias; a.skip = skip;
    a.xc = n_k_chunks; a.sc = 0; a.oc = cdiv(C, 32);
    a.taps = 1; a.dil = 1; a.nk_x = a.xc; a.nk = a.xc;
    a.Mpad = Mpad; a.Lp = Lp; a.halo = halo; a.L = L; a.B = B;
    a.n_ttiles = cdiv(L, 256);
    a.C = C; a.n_res = 0; a.skip_init = 1;
    const int mt_rows = cdiv(C, 256) * a.n_ttiles * B < 200 ? 128 : 256;
    a.n_mtiles = cdiv(C, mt_rows);
    T2S_CHECK_HIP(t2s_launch_conv_gemm(a, EPI_RESSKIP, (hipStream_t)stream, mt_rows));
    return T2S_OK;
}

int t2s_wg_res_skip_train(const void* A_hi, const void* A_lo, const float* bias, const void* acts_hi,
                          const void* acts_lo, const void* R_hi, const void* R_lo, void* X_hi, void* X_lo, float* skip,
                          int B, int C, int n_res, int skip_init, int L, int Lp, int halo, int Mpad, void* stream) {
    if (!planes_ok(A_hi, A_lo) || !planes_ok(acts_hi, acts_lo) || !bias || !skip || !al16(skip)) return T2S_EINVAL;
    if (n_res > 0 && (!planes_ok(X_hi, X_lo) || !planes_ok(R_hi, R_lo))) return T2S_EINVAL;
    if (B <= 0 || L <= 0 || C <= 0 || C % 4 || (n_res != 0 && n_res != C)) return T2S_EINVAL;
    if (Lp != t2s_plane_rows(L, halo) || Mpad % 256 || Mpad < n_res + C || !al16(bias)) return T2S_EINVAL;
    ConvGemmArgs a;
    memset(&a, 0, sizeof(a));
    a.A_hi = (const u16*)A_hi; a.A_lo = (const u16*)A_lo;
    a.X_hi = (const u16*)acts_hi; a.X_lo = (const u16*)acts_lo;
    a.bias = bias; a.O_hi = (u16*)X_hi; a.O_lo = (u16*)X_lo; a.skip = skip;
    a.R_hi = (const u16*)R_hi; a.R_lo = (const u16*)R_lo;
    a.xc = cdiv(C, 32); a.sc = 0; a.oc = cdiv(C, 32);
    a.taps = 1; a.dil = 1; a.nk_x = a.xc; a.nk = a.xc;
    a.Mpad = Mpad; a.Lp = Lp; a.halo = halo; a.L = L; a.B = B;
    a.n_mtiles = cdiv(n_res + C, 256); a.n_ttiles = cdiv(L, 256);
    a.C = C; a.n_res = n_res; a.skip_init = skip_init;
    T2S_CHECK_HIP(t2s_launch_conv_gemm(a, EPI_RESSKIP, (hipStream_t)stream));
    return T2S_OK;
}

int t2s_wg_bwd_gate_dgrad(const void* A_hi, const void* A_lo, const float* zero_bias, const void* DX_hi,
                          const void* DX_lo, const void* DS_hi, const void* DS_lo, const void* T_hi, const void* T_lo,
                          const void* G_hi, const void* G_lo, int tg_bchunks, void* DP_hi, void* DP_lo, int dp_bchunks, int B, int C,
                          int L, int Lp, int halo, int Mpad, int pair8, void* stream) {
    if (!planes_ok(A_hi, A_lo) || !planes_ok(DS_hi, DS_lo) || !planes_ok(T_hi, T_lo) || !planes_ok(G_hi, G_lo) ||
        !planes_ok(DP_hi, DP_lo) || !zero_bias)
        return T2S_EINVAL;
    if (DX_hi && !planes_ok(DX_hi, DX_lo)) return T2S_EINVAL;
    if (B <= 0 || L <= 0 || C <= 0 || C % 32 || Lp != t2s_plane_rows(L, halo) || Mpad % 256 || Mpad < C) return T2S_EINVAL;
    if ((dp_bchunks != 0 && dp_bchunks < 2 * (C / 32)) || (tg_bchunks != 0 && tg_bchunks < C / 32)) return T2S_EINVAL;
    ConvGemmArgs a;
    memset(&a, 0, sizeof(a));
    a.A_hi = (const u16*)A_hi; a.A_lo = (const u16*)A_lo;
    const int cc = C / 32;
    if (DX_hi) {   // K = [d_x channels | d_skip channels]
        a.X_hi = (const u16*)DX_hi; a.X_lo = (const u16*)DX_lo; a.xc = cc;
        a.S_hi = (const u16*)DS_hi; a.S_lo = (const u16*)DS_lo; a.sc = cc;
    } else {       // last layer: only skip rows exist
        a.X_hi = (const u16*)DS_hi; a.X_lo = (const u16*)DS_lo; a.xc = cc; a.sc = 0;
    }
    a.bias = zero_bias; a.O_hi = (u16*)DP_hi; a.O_lo = (u16*)DP_lo;
    a.T_hi = (u16*)T_hi; a.T_lo = (u16*)T_lo; a.G_hi = (u16*)G_hi; a.G_lo = (u16*)G_lo;
    a.oc = dp_bchunks ? dp_bchunks : 2 * cc;                 // oc = batch stride of the output planes (a slice of a wider set)
    a.tc = tg_bchunks ? tg_bchunks : cc;                     // the same for the saved gate output / sigmoid planes
    a.taps = 1; a.dil = 1; a.nk_x = a.xc; a.nk = a.xc + a.sc;
    a.Mpad = Mpad; a.Lp = Lp; a.halo = halo; a.L = L; a.B = B;
    a.n_ttiles = cdiv(L, 256);
    a.C = C;
    a.pair8 = pair8 ? 1 : 0;
    if (bwd_pp256(a, C)) {
        a.n_mtiles = cdiv(C, 256);
        T2S_CHECK_HIP(t2s_launch_bwd_gemm_pp(a, EPI_GATE_BWD, (hipStream_t)stream));
        return T2S_OK;
    }
    if (pair8) return T2S_EINVAL;        // PERM_PAIR8 operands need the 256-row ping-pong kernel (t2s_wg_bwd_pair8_ok)
    // 128-row tiles when 256-row tiles would leave half the CUs without a workgroup (C = 512: 2 x 64 tiles)
    const int mt_rows = cdiv(C, 256) * a.n_ttiles * B < 200 ? 128 : 256;
    a.n_mtiles = cdiv(C, mt_rows);
    T2S_CHECK_HIP(t2s_launch_conv_gemm(a, EPI_GATE_BWD, (hipStream_t)stream, mt_rows));
    return T2S_OK;
}

int t2s_conv_accumulate(const void* A_hi, const void* A_lo, const float* zero_bias, const void* X_hi, const void* X_lo,
                        int x_bchunks, void* O_hi, void* O_lo, int B, int Cin, int Cout, int taps, int dilation, int init, int L, int Lp,
                        int halo, int Mpad, int pair8, void* stream) {
    if (!planes_ok(A_hi, A_lo) || !planes_ok(X_hi, X_lo) || !planes_ok(O_hi, O_lo) || !zero_bias) return T2S_EINVAL;
    if (B <= 0 || L <= 0 || Cin <= 0 || Cout <= 0 || Cout % 4 || taps <= 0 || !(taps & 1) || dilation <= 0) return T2S_EINVAL;
    if ((taps / 2) * dilation > halo || Lp != t2s_plane_rows(L, halo) || Mpad % 256 || Mpad < Cout) return T2S_EINVAL;
    if (x_bchunks != 0 && x_bchunks < cdiv(Cin, 32)) return T2S_EINVAL;
    ConvGemmArgs a;
    memset(&a, 0, sizeof(a));
    a.A_hi = (const u16*)A_hi; a.A_lo = (const u16*)A_lo;
    a.X_hi = (const u16*)X_hi; a.X_lo = (const u16*)X_lo;
    a.bias = zero_bias; a.O_hi = (u16*)O_hi; a.O_lo = (u16*)O_lo;
    a.xc = cdiv(Cin, 32); a.sc = 0; a.oc = cdiv(Cout, 32); a.xbs = x_bchunks;
    a.taps = taps; a.dil = dilation; a.nk_x = taps * a.xc; a.nk = a.nk_x;
    a.Mpad = Mpad; a.Lp = Lp; a.halo = halo; a.L = L; a.B = B;
    a.n_ttiles = cdiv(L, 256);
    a.C = 0; a.n_res = Cout; a.res_init = init;      // every row takes the residual branch
    if (pair8 && Cout % 32) return T2S_EINVAL;
    a.pair8 = pair8 ? 1 : 0;
    if (bwd_pp256(a, Cout)) {
        a.n_mtiles = cdiv(Cout, 256);
        T2S_CHECK_HIP(t2s_launch_bwd_gemm_pp(a, EPI_RESSKIP, (hipStream_t)stream));
        return T2S_OK;
    }
    if (pair8) return T2S_EINVAL;        // PERM_PAIR8 operands need the 256-row ping-pong kernel (t2s_wg_bwd_pair8_ok)
    // 128-row tiles when 256-row tiles would leave most CUs without a workgroup
    const int mt_rows = (cdiv(Cout, 256) * a.n_ttiles * B <= 128 || Cout % 256 == 0) && cdiv(Cout, 256) * a.n_ttiles * B < 200 ? 128 : 256;
    a.n_mtiles = cdiv(Cout, mt_rows);
    T2S_CHECK_HIP(t2s_launch_conv_gemm(a, EPI_RESSKIP, (hipStream_t)stream, mt_rows));
    return T2S_OK;
}

int t2s_wgrad_gemm(const void* A_hi, const void* A_lo, const void* X_hi, const void* X_lo, const float* zero_bias,
                   float* out, int B, int M, int N, int Mpad, int Npad, int n_tchunks, int k0, int k1, int ksplit,
                   void* stream) {
    if (!planes_ok(A_hi, A_lo) || !planes_ok(X_hi, X_lo) || !zero_bias || !out) return T2S_EINVAL;
    if (B <= 0 || M <= 0 || N <= 0 || M % 4 || Mpad % 256 || Mpad < M || Npad != cdiv(N, 256) * 256 || n_tchunks <= 0)
        return T2S_EINVAL;
    if (k0 < 0 || k1 > n_tchunks || k0 >= k1 || ksplit < 1 || ksplit > 16) return T2S_EINVAL;
    ConvGemmArgs a;
    memset(&a, 0, sizeof(a));
    a.A_hi = (const u16*)A_hi; a.A_lo = (const u16*)A_lo; a.a_bstride = (long)n_tchunks * Mpad * 32;
    a.X_hi = (const u16*)X_hi; a.X_lo = (const u16*)X_lo;
    a.bias = zero_bias; a.out_f32 = out;
    a.xc = n_tchunks; a.sc = 0; a.oc = cdiv(M, 32);
    a.taps = 1; a.dil = 1; a.nk_x = n_tchunks; a.nk = n_tchunks;
    a.Mpad = Mpad; a.Lp = Npad; a.halo = 0; a.L = N; a.B = B * ksplit;
    a.ksplit = ksplit; a.k0 = k0; a.kend = k1; a.kchunk = cdiv(k1 - k0, ksplit);
    a.n_mtiles = cdiv(M, 256); a.n_ttiles = cdiv(N, 256);
    a.C = M; a.act = ACT_NONE;
    T2S_CHECK_HIP(t2s_launch_conv_gemm(a, EPI_BIAS_ACT, (hipStream_t)stream));
    return T2S_OK;
}

int t2s_wgrad_gemm_flat(const void* A_hi, const void* A_lo, const void* X_hi, const void* X_lo, const float* zero_bias,
                        float* out, int B, int M, int N, int Mpad, int Npad, int n_tchunks, int k0, int k1, int nsplit,
                        void* stream) {
    if (!planes_ok(A_hi, A_lo) || !planes_ok(X_hi, X_lo) || !zero_bias || !out) return T2S_EINVAL;
    if (B <= 0 || M <= 0 || N <= 0 || M % 4 || Mpad % 256 || Mpad < M || Npad != cdiv(N, 256) * 256 || n_tchunks <= 0)
        return T2S_EINVAL;
    if (k0 < 0 || k1 > n_tchunks || k0 >= k1 || nsplit < 1 || nsplit > B * (k1 - k0)) return T2S_EINVAL;
    ConvGemmArgs a;
    memset(&a, 0, sizeof(a));
    a.A_hi = (const u16*)A_hi; a.A_lo = (const u16*)A_lo; a.a_bstride = (long)n_tchunks * Mpad * 32;
    a.X_hi = (const u16*)X_hi; a.X_lo = (const u16*)X_lo;
    a.bias = zero_bias; a.out_f32 = out;
    a.xc = n_tchunks; a.sc = 0; a.oc = cdiv(M, 32);
    a.taps = 1; a.dil = 1; a.nk_x = n_tchunks; a.nk = n_tchunks;
    a.Mpad = Mpad; a.Lp = Npad; a.halo = 0; a.L = N; a.B = nsplit;
    a.ksplit = 1; a.k0 = k0; a.kend = k1; a.kflat = B; a.kchunk = cdiv(B * (k1 - k0), nsplit);
    if ((long)a.kchunk * (nsplit - 1) >= (long)B * (k1 - k0)) return T2S_EINVAL;      // every slab must own >= 1 K-step
    a.n_mtiles = cdiv(M, 256); a.n_ttiles = cdiv(N, 256);
    a.C = M; a.act = ACT_NONE;
    T2S_CHECK_HIP(t2s_launch_conv_gemm(a, EPI_BIAS_ACT, (hipStream_t)stream));
    return T2S_OK;
}

int t2s_wgrad_cl(const t2s_wgrad_chunk* a_chunks, int n_a_chunks, const t2s_wgrad_chunk* b_chunks, int n_b_chunks, float* out,
                 int B, int M, int N, int ldp, int k0, int k1, int nsplit, int bias_cols, void* stream) {
    static_assert(sizeof(t2s_wgrad_chunk) == sizeof(WgradChunk), "t2s_wgrad_chunk layout");
    if (!a_chunks || !b_chunks || !out || B <= 0 || M <= 0 || N <= 0 || k0 < 0 || k0 >= k1 || nsplit < 1) return T2S_EINVAL;
    if (ldp < N || (ldp % 4 == 0 && !al16(out)) || (bias_cols && (ldp % 4 || ldp < N + 4))) return T2S_EINVAL;
    const int n_mtiles = cdiv(M, 256), n_ntiles = cdiv(N, 256);
    if (n_a_chunks != n_mtiles * 8 || n_b_chunks != n_ntiles * 8 || nsplit > B * (k1 - k0)) return T2S_EINVAL;
    WgradClArgs a;
    a.a_chunks = (const WgradChunk*)a_chunks; a.b_chunks = (const WgradChunk*)b_chunks; a.P = out;
    a.M = M; a.N = N; a.ldp = ldp; a.n_mtiles = n_mtiles; a.n_ntiles = n_ntiles; a.B = B; a.k0 = k0; a.k1 = k1;
    a.nslab = nsplit; a.kchunk = cdiv(B * (k1 - k0), nsplit);
    a.bias_cols = bias_cols ? 1 : 0;
    if ((long)a.kchunk * (nsplit - 1) >= (long)B * (k1 - k0)) return T2S_EINVAL;      // every slab must own >= 1 K-block
    T2S_CHECK_HIP(t2s_launch_wgrad_cl(a, (hipStream_t)stream));
    return T2S_OK;
}

int t2s_plane_transpose(const void* src_hi, const void* src_lo, int B, int src_chunks, int n_chunks, int Lp, int shift,
                        void* dst_hi, void* dst_lo, int Npad, int n_off, void* stream) {
    if (!planes_ok(src_hi, src_lo) || !planes_ok(dst_hi, dst_lo) || B <= 0 || n_chunks <= 0 || n_chunks > src_chunks ||
        Lp <= 0 || n_off % 32 || n_off + n_chunks * 32 > Npad)
        return T2S_EINVAL;
    T2S_CHECK_HIP(t2s_launch_plane_transpose((const u16*)src_hi, (const u16*)src_lo, B, src_chunks, n_chunks, Lp, shift,
                                             (u16*)dst_hi, (u16*)dst_lo, Npad, n_off, (hipStream_t)stream));
    return T2S_OK;
}

int t2s_tm_ones_row(void* dst_hi, void* dst_lo, int B, int Lp, int halo, int L, int Npad, int n_row, void* stream) {
    if (!dst_hi || !dst_lo || B <= 0 || Lp <= 0 || n_row < 0 || n_row >= Npad) return T2S_EINVAL;
    T2S_CHECK_HIP(t2s_launch_tm_ones_row((u16*)dst_hi, (u16*)dst_lo, B, Lp, halo, L, Npad, n_row, (hipStream_t)stream));
    return T2S_OK;
}

int t2s_pack_transposed(const float* v, const float* scale, int O, int Cin, int Kt, int flip, int O_pad, int Mpad,
                        int koff, void* A_hi, void* A_lo, int pair8, void* stream) {
    if (!v || !planes_ok(A_hi, A_lo) || O <= 0 || Cin <= 0 || Kt <= 0 || O_pad % 32 || O_pad < O || Mpad % 256 ||
        Mpad < Cin || koff % 32 || (pair8 && Cin % 32))
        return T2S_EINVAL;
    T2S_CHECK_HIP(t2s_launch_pack_transposed(v, scale, O, Cin, Kt, flip, O_pad, Mpad, koff, (u16*)A_hi, (u16*)A_lo, pair8 ? 1 : 0,
                                             (hipStream_t)stream));
    return T2S_OK;
}

int t2s_weightnorm_scale(const float* v, const float* g, int O, int K, float* scale, void* stream) {
    if (!v || !scale || O <= 0 || K <= 0) return T2S_EINVAL;
    hipLaunchKernelGGL(weightnorm_scale_kernel, dim3((O + 255) / 256), dim3(256), 0, (hipStream_t)stream, v, g, O, K, scale);
    T2S_CHECK_HIP(hipGetLastError());
    return T2S_OK;
}

int t2s_wn_backward(const float* P, int nsplit, int Prows, int Pcols, int row_off, int col_off, int tap_stride,
                    int col_bias, int n_bias_cols, const float* v, const float* g, int O, int Cin, int Kt, float* dv, float* dg,
                    float* db, int db_accum, void* stream) {
    if (!P || !v || !dv || (g && !dg) || nsplit <= 0 || O <= 0 || Cin <= 0 || Kt <= 0) return T2S_EINVAL;
    if (n_bias_cols < 1) n_bias_cols = 1;
    if (row_off + O > Prows || col_off + (Kt - 1) * tap_stride + Cin > Pcols || (db && col_bias + n_bias_cols > Pcols)) return T2S_EINVAL;
    if ((size_t)Cin * Kt * sizeof(float) > 48 * 1024) return T2S_EINVAL;
    WnBwdArgs a;
    a.P = P; a.v = v; a.g = g; a.dv = dv; a.dg = dg; a.db = db;
    a.nsplit = nsplit; a.Prows = Prows; a.Pcols = Pcols; a.row_off = row_off; a.col_off = col_off;
    a.tap_stride = tap_stride; a.col_bias = col_bias; a.n_bias_cols = n_bias_cols; a.O = O; a.Cin = Cin; a.Kt = Kt; a.db_accum = db_accum;
    T2S_CHECK_HIP(t2s_launch_wn_backward(a, (hipStream_t)stream));
    return T2S_OK;
}

int t2s_wg_affine_backward(float* z, float* dz, const float* wn_out, const float* g_log_s, int g_log_s_scalar, float* d_out, int B,
                           int n_group, int c_off, int n_half, int L, void* stream) {
    if (!z || !dz || !wn_out || !d_out || B <= 0 || L <= 0 || n_half <= 0 || c_off < 0 || c_off + 2 * n_half > n_group)
        return T2S_EINVAL;
    T2S_CHECK_HIP(t2s_launch_affine_backward(z, dz, wn_out, g_log_s, g_log_s_scalar, d_out, B, n_group, c_off, n_half, L,
                                             (hipStream_t)stream));
    return T2S_OK;
}

long t2s_small_wgrad_scratch(int B, int chunks) { return (long)t2s_small_wgrad_scratch_floats(B, chunks); }

int t2s_small_wgrad(const void* P_hi, const void* P_lo, const float* P_f32, const float* Q, float* out, float* rowsum,
                    float* scratch, int B, int chunks, int Lp, int halo, int L, int R, int J, int Jtot, int q_off, int out_transposed,
                    void* stream) {
    if ((!P_f32 && (!P_hi || !P_lo)) || !Q || !out || !scratch || B <= 0 || chunks <= 0 || L <= 0 || R <= 0 || R > chunks * 32 ||
        J <= 0 || J > 16 || q_off < 0 || q_off + J > Jtot)
        return T2S_EINVAL;
    SmallWgradArgs a;
    a.P_hi = (const u16*)P_hi; a.P_lo = (const u16*)P_lo; a.P_f32 = P_f32; a.Q = Q; a.out = out; a.rowsum = rowsum; a.scratch = scratch;
    a.B = B; a.chunks = chunks; a.Lp = Lp; a.halo = halo; a.L = L; a.R = R; a.J = J; a.Jtot = Jtot; a.q_off = q_off;
    a.out_transposed = out_transposed;
    T2S_CHECK_HIP(t2s_launch_small_wgrad(a, (hipStream_t)stream));
    return T2S_OK;
}

int t2s_rows_sum(const float* Q, int B, int Jtot, int q_off, int J, int L, float* out, void* stream) {
    if (!Q || !out || B <= 0 || J <= 0 || q_off < 0 || q_off + J > Jtot || L <= 0) return T2S_EINVAL;
    T2S_CHECK_HIP(t2s_launch_rows_sum(Q, B, Jtot, q_off, J, L, out, (hipStream_t)stream));
    return T2S_OK;
}

int t2s_wg_start_dgrad(const void* X_hi, const void* X_lo, const float* w, float* dz, int B, int n_group, int c_off,
                       int n_half, int C, int L, int Lp, int halo, void* stream) {
    if (!X_hi || !X_lo || !w || !dz || B <= 0 || L <= 0 || n_half <= 0 || n_half > 8 || c_off < 0 ||
        c_off + n_half > n_group || C <= 0)
        return T2S_EINVAL;
    T2S_CHECK_HIP(t2s_launch_start_dgrad((const u16*)X_hi, (const u16*)X_lo, w, dz, B, n_group, c_off, n_half, C, L, Lp,
                                         halo, (hipStream_t)stream));
    return T2S_OK;
}

int t2s_wg_convinv_wgrad(const float* dz, const float* zin, const float* Winv, const float* gscale_ptr, float gmul,
                         int B, int n_group, int c_off, int n, int L, float* dW, void* stream) {
    if (!dz || !zin || !Winv || !dW || B <= 0 || L <= 0 || n <= 0 || n > 16 || c_off < 0 || c_off + n > n_group) return T2S_EINVAL;
    T2S_CHECK_HIP(t2s_launch_convinv_wgrad(dz, zin, Winv, gscale_ptr, gmul, B, n_group, c_off, n, L, dW, (hipStream_t)stream));
    return T2S_OK;
}

int t2s_wg_upsample_wgrad(const void* D_hi, const void* D_lo, const float* mel, int B, int n_mel, int frames, int ksize,
                          int stride, int n_group, int L, int Lp, int halo, float* dW, float* db, void* stream) {
    if (!D_hi || !D_lo || !mel || !dW || !db || B <= 0 || n_mel <= 0 || n_mel > 80 || frames <= 0 || L <= 0) return T2S_EINVAL;
    T2S_CHECK_HIP(t2s_launch_upsample_wgrad((const u16*)D_hi, (const u16*)D_lo, mel, B, n_mel, frames, ksize, stride,
                                            n_group, L, Lp, halo, dW, (hipStream_t)stream));
    T2S_CHECK_HIP(t2s_launch_upsample_bgrad((const u16*)D_hi, (const u16*)D_lo, B, n_mel, n_group, L, Lp, halo, db,
                                            (hipStream_t)stream));
    return T2S_OK;
}

int t2s_adam_table(const t2s_adam_job* jobs, int n_jobs, long total_blocks, float lr, float beta1, float beta2,
                   float eps, int step, float gscale, float weight_decay, void* stream) {
    if (!jobs || n_jobs <= 0 || total_blocks <= 0 || total_blocks > 0x7fffffffL || step <= 0) return T2S_EINVAL;
    static_assert(sizeof(t2s_adam_job) == sizeof(AdamJob), "t2s_adam_job layout");
    const float bc1 = 1.0f - powf(beta1, (float)step);
    const float bc2s = sqrtf(1.0f - powf(beta2, (float)step));
    T2S_CHECK_HIP(t2s_launch_adam_table((const AdamJob*)jobs, n_jobs, total_blocks, lr, beta1, beta2, eps, bc1, bc2s, gscale,
                                        weight_decay, (hipStream_t)stream));
    return T2S_OK;
}

}  // extern "C"
