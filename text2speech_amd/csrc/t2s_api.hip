// extern "C" boundary: argument validation + launch.  No torch types, no allocation, no sync.
#include "../../include/t2s_hip.h"
#include "t2s_kernels.h"

#include <stdlib.h>
#include <string.h>

static thread_local char g_hip_err[256] = "";

extern "C" int t2s_internal_fail_hip(int e);
static int fail_hip(hipError_t e) { return t2s_internal_fail_hip((int)e); }
extern "C" int t2s_internal_fail_hip(int ei) {
    hipError_t e = (hipError_t)ei;
    strncpy(g_hip_err, hipGetErrorString(e), sizeof(g_hip_err) - 1);
    g_hip_err[sizeof(g_hip_err) - 1] = 0;
    return T2S_EHIP;
}
#define T2S_CHECK_HIP(expr)                   \
    do {                                      \
        hipError_t _e = (expr);               \
        if (_e != hipSuccess) return fail_hip(_e); \
    } while (0)

static inline bool aligned16(const void* p) { return ((uintptr_t)p & 15) == 0; }
static inline int cdiv(int a, int b) { return (a + b - 1) / b; }

extern "C" {

int t2s_abi_version(void) { return 4; }

int t2s_sizeof_taco_decoder(void) { return (int)sizeof(t2s_taco_decoder); }
int t2s_sizeof_taco_bptt(void) { return (int)sizeof(t2s_taco_bptt); }

int t2s_operand_format(void) {
#ifdef T2S_SPLIT_F16
    return 1;
#else
    return 0;
#endif
}

const char* t2s_error_string(int code) {
    switch (code) {
        case T2S_OK: return "ok";
        case T2S_EINVAL: return "invalid argument";
        case T2S_EHIP: return "HIP runtime error";
        default: return "unknown error";
    }
}
const char* t2s_last_hip_error(void) { return g_hip_err; }

int t2s_plane_rows(int L, int halo) { return cdiv(L, 256) * 256 + 2 * halo; }
int t2s_padded_rows(int rows) { return cdiv(rows, 256) * 256; }

int t2s_pack_conv_weight(const float* v, const float* g, int g_is_scale, const float* bias_in, int O, int Cin, int Kt,
                         int perm, int C_gate, int row_off, int Mpad, int koff, int Cin_pad, void* A_hi, void* A_lo,
                         float* bias_out, int bias_accumulate, void* stream) {
    if (!v || !A_hi || !A_lo || O <= 0 || Cin <= 0 || Kt <= 0) return T2S_EINVAL;
    if (Mpad % 256 || koff % 32 || Cin_pad % 32 || Cin_pad < Cin) return T2S_EINVAL;
    if (perm == T2S_PERM_GATE) {
        if (O != 2 * C_gate || cdiv(C_gate, 128) * 256 > Mpad) return T2S_EINVAL;
    } else if (perm == T2S_PERM_NONE) {
        if (row_off < 0 || row_off + O > Mpad) return T2S_EINVAL;
    } else {
        return T2S_EINVAL;
    }
    PackArgs a;
    a.v = v; a.g = g; a.bias_in = bias_in;
    a.A_hi = (u16*)A_hi; a.A_lo = (u16*)A_lo; a.bias_out = bias_out;
    a.O = O; a.Cin = Cin; a.Kt = Kt; a.perm = perm; a.C_gate = C_gate; a.Mpad = Mpad; a.koff = koff;
    a.Cin_pad = Cin_pad; a.bias_accumulate = bias_accumulate; a.row_off = row_off; a.g_is_scale = g_is_scale;
    T2S_CHECK_HIP(t2s_launch_pack(a, (hipStream_t)stream));
    return T2S_OK;
}

int t2s_pack_conv_weight_table(const t2s_pack_job* jobs, int n_jobs, long total_rows, void* stream) {
    if (!jobs || n_jobs <= 0 || total_rows <= 0 || total_rows > 0x7fffffffL) return T2S_EINVAL;
    static_assert(sizeof(t2s_pack_job) == sizeof(PackJob), "t2s_pack_job layout");
    T2S_CHECK_HIP(t2s_launch_pack_table((const PackJob*)jobs, n_jobs, total_rows, (hipStream_t)stream));
    return T2S_OK;
}

int t2s_weightnorm_small(const float* v, const float* g, int O, int K, float* w, void* stream) {
    if (!v || !w || O <= 0 || K <= 0) return T2S_EINVAL;
    T2S_CHECK_HIP(t2s_launch_weightnorm_small(v, g, O, K, w, (hipStream_t)stream));
    return T2S_OK;
}

int t2s_wg_upsample_squeeze(const float* mel, const float* W, const float* bias, int B, int n_mel, int frames,
                            int ksize, int stride, int n_group, int L, int Lp, int halo, void* S_hi, void* S_lo,
                            void* stream) {
    if (!mel || !W || !bias || !S_hi || !S_lo) return T2S_EINVAL;
    if (B <= 0 || n_mel <= 0 || frames <= 0 || L <= 0 || n_group <= 0 || stride <= 0 || ksize % stride) return T2S_EINVAL;
    if (!aligned16(W) || !aligned16(S_hi) || !aligned16(S_lo)) return T2S_EINVAL;
    if (Lp < t2s_plane_rows(L, halo)) return T2S_EINVAL;
    // every squeezed sample must exist in the transposed-conv output (reference glow.py:216 assert)
    if ((long)L * n_group > (long)(frames - 1) * stride + ksize) return T2S_EINVAL;
    if (n_group == 8 && (stride % 8 || ksize % 8)) return T2S_EINVAL;
    T2S_CHECK_HIP(t2s_launch_upsample_squeeze(mel, W, bias, B, n_mel, frames, ksize, stride, n_group, L, Lp, halo,
                                              (u16*)S_hi, (u16*)S_lo, (hipStream_t)stream));
    return T2S_OK;
}

int t2s_wg_audio_squeeze(float* audio, float* z, int B, int T, int n_group, int L, int unsqueeze, void* stream) {
    if (!audio || !z || B <= 0 || L <= 0 || n_group <= 0 || (long)L * n_group > T) return T2S_EINVAL;
    T2S_CHECK_HIP(t2s_launch_audio_squeeze(audio, z, B, T, n_group, L, unsqueeze, (hipStream_t)stream));
    return T2S_OK;
}

int t2s_wg_convinv(float* z, const float* W, int B, int n_group, int c_off, int n_rem, int L, void* stream) {
    if (!z || !W || B <= 0 || L <= 0 || n_rem <= 0 || n_rem > 16 || c_off < 0 || c_off + n_rem > n_group) return T2S_EINVAL;
    T2S_CHECK_HIP(t2s_launch_convinv(z, W, B, n_group, c_off, n_rem, L, (hipStream_t)stream));
    return T2S_OK;
}

int t2s_small_logdet_inv(const float* W, int n, float scale, float* logdet_out, float* inv_out, void* stream) {
    if (!W || n <= 0 || n > 16 || (!logdet_out && !inv_out)) return T2S_EINVAL;
    T2S_CHECK_HIP(t2s_launch_small_logdet_inv(W, n, scale, logdet_out, inv_out, (hipStream_t)stream));
    return T2S_OK;
}

int t2s_small_logdet_inv_batch(const t2s_small_mat_job* jobs, int n_jobs, float scale, void* stream) {
    if (!jobs || n_jobs <= 0) return T2S_EINVAL;
    static_assert(sizeof(t2s_small_mat_job) == sizeof(SmallMatJob), "t2s_small_mat_job layout");
    T2S_CHECK_HIP(t2s_launch_small_logdet_batch((const SmallMatJob*)jobs, n_jobs, scale, (hipStream_t)stream));
    return T2S_OK;
}

int t2s_small_logdet_inv_batch_host(const t2s_small_mat_job* host_jobs, int n_jobs, float scale, void* stream) {
    if (!host_jobs || n_jobs <= 0 || n_jobs > 16) return T2S_EINVAL;
    for (int i = 0; i < n_jobs; ++i)
        if (!host_jobs[i].W || host_jobs[i].n <= 0 || host_jobs[i].n > 16) return T2S_EINVAL;
    T2S_CHECK_HIP(t2s_launch_small_logdet_batch_host((const SmallMatJob*)host_jobs, n_jobs, scale, (hipStream_t)stream));
    return T2S_OK;
}

int t2s_wg_start(const float* z, const float* w, const float* bias, int B, int n_group, int c_off, int n_half, int C,
                 int L, int Lp, int halo, void* X_hi, void* X_lo, void* stream) {
    if (!z || !w || !bias || !X_hi || !X_lo) return T2S_EINVAL;
    if (B <= 0 || L <= 0 || C <= 0 || n_half <= 0 || n_half > 8 || c_off < 0 || c_off + n_half > n_group) return T2S_EINVAL;
    if (!aligned16(X_hi) || !aligned16(X_lo) || Lp < t2s_plane_rows(L, halo)) return T2S_EINVAL;
    T2S_CHECK_HIP(t2s_launch_start(z, w, bias, B, n_group, c_off, n_half, C, L, Lp, halo, (u16*)X_hi, (u16*)X_lo,
                                   (hipStream_t)stream));
    return T2S_OK;
}

static int check_planes(const void* a, const void* b) { return a && b && aligned16(a) && aligned16(b); }

int t2s_wg_in_cond_gate(const void* A_hi, const void* A_lo, const float* bias, const void* X_hi, const void* X_lo,
                        const void* S_hi, const void* S_lo, void* acts_hi, void* acts_lo, int B, int C, int n_cond,
                        int taps, int dilation, int L, int Lp, int halo, int Mpad, void* stream) {
    if (!check_planes(A_hi, A_lo) || !check_planes(X_hi, X_lo) || !check_planes(acts_hi, acts_lo) || !bias) return T2S_EINVAL;
    if (n_cond > 0 && !check_planes(S_hi, S_lo)) return T2S_EINVAL;
    if (B <= 0 || L <= 0 || C <= 0 || C % 4 || taps <= 0 || !(taps & 1) || dilation <= 0) return T2S_EINVAL;
    if ((taps / 2) * dilation > halo || Lp != t2s_plane_rows(L, halo)) return T2S_EINVAL;
    if (Mpad % 256 || Mpad < cdiv(C, 128) * 256 || !aligned16(bias)) return T2S_EINVAL;
    ConvGemmArgs a;
    memset(&a, 0, sizeof(a));
    a.A_hi = (const u16*)A_hi; a.A_lo = (const u16*)A_lo;
    a.X_hi = (const u16*)X_hi; a.X_lo = (const u16*)X_lo;
    a.S_hi = (const u16*)S_hi; a.S_lo = (const u16*)S_lo;
    a.bias = bias; a.O_hi = (u16*)acts_hi; a.O_lo = (u16*)acts_lo;
    a.xc = cdiv(C, 32); a.sc = cdiv(n_cond, 32); a.oc = cdiv(C, 32);
    a.taps = taps; a.dil = dilation;
    a.nk_x = taps * a.xc; a.nk = a.nk_x + a.sc;
    a.Mpad = Mpad; a.Lp = Lp; a.halo = halo; a.L = L; a.B = B;
    a.n_mtiles = cdiv(C, 128); a.n_ttiles = cdiv(L, 256);
    a.C = C;
    T2S_CHECK_HIP(t2s_launch_conv_gemm(a, EPI_GATE, (hipStream_t)stream));
    return T2S_OK;
}

int t2s_wg_res_skip(const void* A_hi, const void* A_lo, const float* bias, const void* acts_hi, const void* acts_lo,
                    void* X_hi, void* X_lo, float* skip, int B, int C, int n_res, int skip_init, int L, int Lp,
                    int halo, int Mpad, void* stream) {
    if (!check_planes(A_hi, A_lo) || !check_planes(acts_hi, acts_lo) || !bias || !skip || !aligned16(skip)) return T2S_EINVAL;
    if (n_res > 0 && !check_planes(X_hi, X_lo)) return T2S_EINVAL;
    if (B <= 0 || L <= 0 || C <= 0 || C % 4 || (n_res != 0 && n_res != C)) return T2S_EINVAL;
    if (Lp != t2s_plane_rows(L, halo) || Mpad % 256 || Mpad < n_res + C || !aligned16(bias)) return T2S_EINVAL;
    ConvGemmArgs a;
    memset(&a, 0, sizeof(a));
    a.A_hi = (const u16*)A_hi; a.A_lo = (const u16*)A_lo;
    a.X_hi = (const u16*)acts_hi; a.X_lo = (const u16*)acts_lo;
    a.bias = bias; a.O_hi = (u16*)X_hi; a.O_lo = (u16*)X_lo; a.skip = skip;
    a.xc = cdiv(C, 32); a.sc = 0; a.oc = cdiv(C, 32);
    a.taps = 1; a.dil = 1;
    a.nk_x = a.xc; a.nk = a.xc;
    a.Mpad = Mpad; a.Lp = Lp; a.halo = halo; a.L = L; a.B = B;
    a.n_mtiles = cdiv(n_res + C, 256); a.n_ttiles = cdiv(L, 256);
    a.C = C; a.n_res = n_res; a.skip_init = skip_init;
    T2S_CHECK_HIP(t2s_launch_conv_gemm(a, EPI_RESSKIP, (hipStream_t)stream));
    return T2S_OK;
}

int t2s_wg_endfold_weights(const t2s_endfold_job* jobs, int n_jobs, int C, void* stream) {
    if (!jobs || n_jobs <= 0 || C <= 0 || (size_t)8 * C * sizeof(float) > 60 * 1024) return T2S_EINVAL;
    static_assert(sizeof(t2s_endfold_job) == sizeof(EndFoldJob), "t2s_endfold_job layout");
    T2S_CHECK_HIP(t2s_launch_endfold_weights((const EndFoldJob*)jobs, n_jobs, C, (hipStream_t)stream));
    return T2S_OK;
}

// Gate GEMM tile height for a shape: 256-row tiles (the ping-pong kernel) unless they leave at least half of the chip's 256 CUs
// without a workgroup - short utterances at B = 1 - where 128-row tiles give twice the workgroups at half the work each.
static int gate_tile_rows(int B, int C, int L) {
    static const int force = getenv("T2S_GATE_TILE") ? atoi(getenv("T2S_GATE_TILE")) : 0;      // 128 / 256: A/B switch
    if (force == 128 || force == 256) return (force == 128 && C % 64 == 0) ? 128 : 256;
    const long wg256 = (long)cdiv(C, 128) * cdiv(L, 256) * B;
    return (wg256 <= 128 && C % 64 == 0) ? 128 : 256;
}

int t2s_wg_gate_tile_rows(int B, int C, int L) {
    if (B <= 0 || C <= 0 || L <= 0) return T2S_EINVAL;
    return gate_tile_rows(B, C, L);
}

int t2s_wg_gate_fold_slots(int B, int C, int L) {
    if (B <= 0 || C <= 0 || L <= 0) return T2S_EINVAL;
    return gate_tile_rows(B, C, L) == 128 ? 2 * cdiv(C, 64) : 2 * cdiv(C, 128);
}

int t2s_wg_in_cond_gate_fold(const void* A_hi, const void* A_lo, const float* bias, const void* X_hi, const void* X_lo,
                             const void* S_hi, const void* S_lo, void* acts_hi, void* acts_lo, const void* fold_A,
                             float* fold_acc, int fold_init, int B, int C, int n_cond, int taps, int dilation, int L,
                             int Lp, int halo, int Mpad, void* stream) {
    if (!check_planes(A_hi, A_lo) || !check_planes(X_hi, X_lo) || !check_planes(acts_hi, acts_lo) || !bias) return T2S_EINVAL;
    if (!fold_A || !fold_acc || !aligned16(fold_A) || C % 16) return T2S_EINVAL;
    if (n_cond > 0 && !check_planes(S_hi, S_lo)) return T2S_EINVAL;
    if (B <= 0 || L <= 0 || C <= 0 || C % 4 || taps <= 0 || !(taps & 1) || dilation <= 0) return T2S_EINVAL;
    if ((taps / 2) * dilation > halo || Lp != t2s_plane_rows(L, halo)) return T2S_EINVAL;
    if (Mpad % 256 || Mpad < cdiv(C, 128) * 256 || !aligned16(bias)) return T2S_EINVAL;
    ConvGemmArgs a;
    memset(&a, 0, sizeof(a));
    a.A_hi = (const u16*)A_hi; a.A_lo = (const u16*)A_lo;
    a.X_hi = (const u16*)X_hi; a.X_lo = (const u16*)X_lo;
    a.S_hi = (const u16*)S_hi; a.S_lo = (const u16*)S_lo;
    a.bias = bias; a.O_hi = (u16*)acts_hi; a.O_lo = (u16*)acts_lo;
    a.fold_A = (const u16*)fold_A; a.fold_acc = fold_acc; a.fold_init = fold_init;
    a.xc = cdiv(C, 32); a.sc = cdiv(n_cond, 32); a.oc = cdiv(C, 32);
    a.taps = taps; a.dil = dilation;
    a.nk_x = taps * a.xc; a.nk = a.nk_x + a.sc;
    a.Mpad = Mpad; a.Lp = Lp; a.halo = halo; a.L = L; a.B = B;
    const int rows = gate_tile_rows(B, C, L);          // fold_acc holds t2s_wg_gate_fold_slots(B, C, L) slots
    a.n_mtiles = cdiv(C, rows / 2); a.n_ttiles = cdiv(L, 256);
    a.C = C;
    T2S_CHECK_HIP(t2s_launch_conv_gemm(a, EPI_GATE, (hipStream_t)stream, rows));
    return T2S_OK;
}

int t2s_wg_upsample_basis(const float* W, const float* bias, int n_mel, int ksize, int stride, int n_group, int Lp, int halo,
                          void* U_hi, void* U_lo, void* stream) {
    if (!W || !bias || !check_planes(U_hi, U_lo)) return T2S_EINVAL;
    if (n_mel <= 0 || n_group <= 0 || stride <= 0 || ksize <= 0 || ksize % stride || stride % n_group) return T2S_EINVAL;
    const int ncols = (stride / n_group) * (ksize / stride) * n_mel + 1;
    if (halo < 0 || Lp != t2s_plane_rows(ncols, halo)) return T2S_EINVAL;
    T2S_CHECK_HIP(t2s_launch_upbasis_planes(W, bias, n_mel, ksize, stride, n_group, Lp, halo, (u16*)U_hi, (u16*)U_lo,
                                            (hipStream_t)stream));
    return T2S_OK;
}

int t2s_wg_compose_cond(const float* tmp, const float* bias_in, int rows, int Mpad, int P, int K2, long ld, void* A2_hi,
                        void* A2_lo, float* bias_out, void* stream) {
    if (!tmp || !bias_in || !bias_out || !check_planes(A2_hi, A2_lo)) return T2S_EINVAL;
    if (rows <= 0 || Mpad % 256 || rows > Mpad || P <= 0 || K2 <= 0 || K2 % 32 || ld < (long)P * K2 + 1) return T2S_EINVAL;
    T2S_CHECK_HIP(t2s_launch_compose_pack(tmp, bias_in, rows, Mpad, P, K2, (int)ld, (u16*)A2_hi, (u16*)A2_lo, bias_out,
                                          (hipStream_t)stream));
    return T2S_OK;
}

int t2s_wg_melwin_planes(const float* mel, int B, int n_mel, int frames, int nlag, int Fp, void* M_hi, void* M_lo, void* stream) {
    if (!mel || !check_planes(M_hi, M_lo)) return T2S_EINVAL;
    if (B <= 0 || n_mel <= 0 || frames <= 0 || nlag <= 0 || Fp < frames) return T2S_EINVAL;
    T2S_CHECK_HIP(t2s_launch_melwin_planes(mel, B, n_mel, frames, nlag, Fp, (u16*)M_hi, (u16*)M_lo, (hipStream_t)stream));
    return T2S_OK;
}

int t2s_wg_in_melwin_gate_fold(const void* A_hi, const void* A_lo, const void* A2_hi, const void* A2_lo, const float* bias,
                               const void* X_hi, const void* X_lo, const void* M_hi, const void* M_lo, void* acts_hi,
                               void* acts_lo, const void* fold_A, float* fold_acc, int fold_init, int B, int C, int K2,
                               int taps, int dilation, int L, int Lp, int halo, int Mpad, int P, int Fp, void* stream) {
    if (!check_planes(A_hi, A_lo) || !check_planes(A2_hi, A2_lo) || !check_planes(X_hi, X_lo) || !check_planes(M_hi, M_lo) ||
        !check_planes(acts_hi, acts_lo) || !bias)
        return T2S_EINVAL;
    if (!fold_A || !fold_acc || !aligned16(fold_A) || C % 16) return T2S_EINVAL;
    if (B <= 0 || L <= 0 || C <= 0 || C % 4 || taps <= 0 || !(taps & 1) || dilation <= 0 || K2 <= 0 || K2 % 32) return T2S_EINVAL;
    if ((taps / 2) * dilation > halo || Lp != t2s_plane_rows(L, halo)) return T2S_EINVAL;
    if (Mpad % 256 || Mpad < cdiv(C, 128) * 256 || !aligned16(bias) || P <= 0) return T2S_EINVAL;
    const int F = cdiv(L, P);
    if (Fp < F) return T2S_EINVAL;
    // the phase tiles are 256 rows high: fold_acc must have been sized (t2s_wg_gate_fold_slots) for that tile height
    if (gate_tile_rows(B, C, L) != 256) return T2S_EINVAL;
    ConvGemmArgs a;
    memset(&a, 0, sizeof(a));
    a.A_hi = (const u16*)A_hi; a.A_lo = (const u16*)A_lo;
    a.A2_hi = (const u16*)A2_hi; a.A2_lo = (const u16*)A2_lo;
    a.X_hi = (const u16*)X_hi; a.X_lo = (const u16*)X_lo;
    a.S_hi = (const u16*)M_hi; a.S_lo = (const u16*)M_lo;
    a.bias = bias; a.O_hi = (u16*)acts_hi; a.O_lo = (u16*)acts_lo;
    a.fold_A = (const u16*)fold_A; a.fold_acc = fold_acc; a.fold_init = fold_init;
    a.xc = cdiv(C, 32); a.sc = K2 / 32; a.oc = cdiv(C, 32);
    a.taps = taps; a.dil = dilation;
    a.nk_x = taps * a.xc; a.nk = a.nk_x + a.sc;
    a.Mpad = Mpad; a.Lp = Lp; a.halo = halo; a.L = L; a.B = B;
    a.n_mtiles = cdiv(C, 128); a.n_ttiles = 0;
    a.C = C;
    a.ph_P = P; a.ph_Fp = Fp;
    a.ph_FT = F <= 64 ? 64 : (F <= 128 ? 128 : 256);
    a.ph_bper = 256 / a.ph_FT;
    a.ph_nft = cdiv(F, a.ph_FT);
    T2S_CHECK_HIP(t2s_launch_gate_gemm_pp(a, (hipStream_t)stream));
    return T2S_OK;
}

int t2s_wg_res_only(const void* A_hi, const void* A_lo, const float* bias, const void* acts_hi, const void* acts_lo,
                    void* X_hi, void* X_lo, int B, int C, int L, int Lp, int halo, int Mpad, int pair8, void* stream) {
    if (!check_planes(A_hi, A_lo) || !check_planes(acts_hi, acts_lo) || !check_planes(X_hi, X_lo) || !bias) return T2S_EINVAL;
    if (B <= 0 || L <= 0 || C <= 0 || C % 4 || Lp != t2s_plane_rows(L, halo) || Mpad % 256 || Mpad < C || !aligned16(bias))
        return T2S_EINVAL;
    ConvGemmArgs a;
    memset(&a, 0, sizeof(a));
    a.A_hi = (const u16*)A_hi; a.A_lo = (const u16*)A_lo;
    a.X_hi = (const u16*)acts_hi; a.X_lo = (const u16*)acts_lo;
    a.bias = bias; a.O_hi = (u16*)X_hi; a.O_lo = (u16*)X_lo;
    a.xc = cdiv(C, 32); a.sc = 0; a.oc = cdiv(C, 32);
    a.taps = 1; a.dil = 1; a.nk_x = a.xc; a.nk = a.xc;
    a.Mpad = Mpad; a.Lp = Lp; a.halo = halo; a.L = L; a.B = B;
    // T2S_RES_TILE=256 (with T2S_RES_PAIR8=0: the 16-byte epilogue exists for 128-row tiles only): 256-row tiles, half the
    // workgroups - the A/B behind DESIGN.md section 8 item 3
    static const int tile_env = getenv("T2S_RES_TILE") ? atoi(getenv("T2S_RES_TILE")) : 0;
    const int rows = (tile_env == 256 && !pair8) ? 256 : 128;
    a.n_mtiles = cdiv(C, rows); a.n_ttiles = cdiv(L, 256);
    a.C = 0; a.n_res = C;
    if (pair8 && C % 32) return T2S_EINVAL;
    a.pair8 = pair8 ? 1 : 0;
    T2S_CHECK_HIP(t2s_launch_conv_gemm(a, EPI_RESSKIP, (hipStream_t)stream, rows));
    return T2S_OK;
}

int t2s_wg_end_fold_affine(const float* fold_acc, int nslots, const float* bes, int n_layers, const float* b_end,
                           float* z, float* log_s, float* wn_out, int B, int n_group, int c_off, int n_half, int L, int reverse,
                           void* stream) {
    if (!fold_acc || !bes || !b_end || !z || nslots <= 0 || n_layers <= 0) return T2S_EINVAL;
    if (B <= 0 || L <= 0 || n_half <= 0 || n_half > 4 || c_off < 0 || c_off + 2 * n_half > n_group) return T2S_EINVAL;
    T2S_CHECK_HIP(t2s_launch_end_fold_affine(fold_acc, nslots, bes, n_layers, b_end, z, log_s, wn_out, B, n_group, c_off, n_half,
                                             L, reverse, (hipStream_t)stream));
    return T2S_OK;
}

int t2s_wg_end_affine(const float* skip, const float* w_end, const float* b_end, float* z, float* log_s,
                      float* wn_out, int B, int n_group, int c_off, int n_half, int C, int L, int Lp, int halo, int reverse, void* stream) {
    if (!skip || !w_end || !b_end || !z) return T2S_EINVAL;
    if (B <= 0 || L <= 0 || C <= 0 || n_half <= 0 || n_half > 8 || c_off < 0 || c_off + 2 * n_half > n_group) return T2S_EINVAL;
    if (Lp < t2s_plane_rows(L, halo)) return T2S_EINVAL;
    T2S_CHECK_HIP(t2s_launch_end_affine(skip, w_end, b_end, z, log_s, wn_out, B, n_group, c_off, n_half, C, L, Lp, halo,
                                        reverse, (hipStream_t)stream));
    return T2S_OK;
}

int t2s_conv_bias_act(const void* A_hi, const void* A_lo, const float* bias, const void* X_hi, const void* X_lo,
                      void* O_hi, void* O_lo, float* out_f32, int f32_channel_last, int B, int Cin, int Cout, int taps,
                      int dilation, int act, int L, int Lp, int halo, int Mpad, void* stream) {
    if (!check_planes(A_hi, A_lo) || !check_planes(X_hi, X_lo) || !bias || !aligned16(bias)) return T2S_EINVAL;
    if ((O_hi || O_lo) && !check_planes(O_hi, O_lo)) return T2S_EINVAL;
    if (!O_hi && !out_f32) return T2S_EINVAL;
    if (B <= 0 || L <= 0 || Cin <= 0 || Cout <= 0 || Cout % 4 || taps <= 0 || !(taps & 1) || dilation <= 0) return T2S_EINVAL;
    if ((taps / 2) * dilation > halo || Lp != t2s_plane_rows(L, halo) || Mpad % 256 || Mpad < Cout) return T2S_EINVAL;
    if (act < T2S_ACT_NONE || act > T2S_ACT_TANH) return T2S_EINVAL;
    ConvGemmArgs a;
    memset(&a, 0, sizeof(a));
    a.A_hi = (const u16*)A_hi; a.A_lo = (const u16*)A_lo;
    a.X_hi = (const u16*)X_hi; a.X_lo = (const u16*)X_lo;
    a.bias = bias; a.O_hi = (u16*)O_hi; a.O_lo = (u16*)O_lo; a.out_f32 = out_f32;
    a.xc = cdiv(Cin, 32); a.sc = 0; a.oc = cdiv(Cout, 32);
    a.taps = taps; a.dil = dilation;
    a.nk_x = taps * a.xc; a.nk = a.nk_x;
    a.Mpad = Mpad; a.Lp = Lp; a.halo = halo; a.L = L; a.B = B;
    // a grid of at most 64 workgroups is latency-bound per K-step: 128-row tiles with three LDS stages (conv_gemm.hip)
    static const int tile_env = getenv("T2S_CONV_TILE") ? atoi(getenv("T2S_CONV_TILE")) : 0;      // 128 / 256: A/B switch
    const int rows = tile_env == 128 || tile_env == 256 ? tile_env : ((long)cdiv(Cout, 256) * cdiv(L, 256) * B <= 64 ? 128 : 256);
    a.n_mtiles = cdiv(Cout, rows); a.n_ttiles = cdiv(L, 256);
    a.C = Cout; a.act = act; a.f32_cl = f32_channel_last;
    T2S_CHECK_HIP(t2s_launch_conv_gemm(a, EPI_BIAS_ACT, (hipStream_t)stream, rows));
    return T2S_OK;
}

}  // extern "C"
