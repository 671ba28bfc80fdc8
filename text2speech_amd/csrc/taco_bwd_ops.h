// Internal argument blocks of the Tacotron-2 backward kernels (taco_bwd_ops.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef unsigned short u16;

struct LstmBwdArgs {
    const float* dh1; long s1;   // up to three gradient sources for h (per-item strides), any may be null
    const float* dh2; long s2;
    const float* dh3; long s3;
    const unsigned char* drop_mask; float drop_scale;   // [B][H] or null
    const float* gates;          // [B][4H] post-activation i,f,g,o
    const float* c_new;          // [B][H]
    const float* c_prev;         // [B][H] or null (zeros)
    float* dc_carry;             // [B][H] in/out
    float* dgates;               // [B][4H] out
    int B, H;
    const float* wq; const float* dq; int q_dim;   // optional: dh += wq^T dq  (wq [q_dim][H], dq [B][q_dim], q_dim <= 256)
    const float* dq_part; int dq_nchunk;           // optional instead of dq: dq[b][k] = sum_c dq_part[(b*dq_nchunk + c)*q_dim + k]
    float* dq_out;                                 // optional with dq_part: the summed d_q [B][q_dim]
};

struct AttBwdArgs {
    const float* dctx1; long sc1; const float* dctx2; long sc2; const float* dctx3; long sc3;   // gradient sources of ctx_t
    const float* w_cur; long s_wcur;      // w_t [B][.] (row stride)
    const float* w_prev; const float* wc_prev; long s_wprev; long s_wcprev;   // w_{t-1}, wc_{t-1} (null: zeros), row strides
    const float* q;              // [B][att_dim] query of this step
    const float* pmem;           // [B][T][att_dim]
    const float* memory;         // [B][T][enc]
    const int* lengths;
    const float* w_loc_conv; const float* w_loc_dense; const float* w_v;
    float* dw_carry;             // [B][T] in: dL/dw_t from step t+1's features; out: same for step t-1
    float* dwc_carry;            // [B][T] in: dL/dwc_t; out: dL/dwc_{t-1}
    float* d_q;                  // [B][att_dim] out
    float* d_pmem;               // [B][T][att_dim] +=
    float* d_memory;             // [B][T][enc] +=
    float* dD_part; float* dK_part; float* dv_part;   // partial parameter gradients, one slot per (batch element, 32-position chunk), +=
    float* dw_buf;               // scratch [B][T]
    float* df_buf;               // scratch [B][T][32]
    float* dq_part;              // scratch [B][ceil(T/32)][att_dim]
    float* dctx_out;             // optional [B][enc]: d_ctx of this step (then d_memory may be null: deferred accumulation)
    int B, T, att_dim, enc_dim, loc_f, loc_ks;
    // one-launch form (att_bwd_fused_kernel): the saved context of this step (row stride s_ctx) and a second pair of carry
    // buffers - the step reads dw_carry / dwc_carry and writes dw_carry_out / dwc_carry_out
    const float* ctx; long s_ctx;
    float* dw_carry_out; float* dwc_carry_out;
};

// Optional second argument of the one-launch attention backward: the attention LSTMCell's pointwise backward (with W_query^T d_q)
// folded into the same launch.  The chunk workgroups of a batch element exchange their partial d_q through tagged 8-byte granules
// (xbuf: [B][ceil(T/32)][128] + 1 error word, zero before a BPTT pass; tag = step + 1), each then does the cell backward of its share
// of the hidden units.  T <= 512.
struct AttBwdFoldArgs {
    LstmBwdArgs cell;
    unsigned long long* xbuf;
    unsigned tag;
    // optional: the first thread of the launch stores sig_val to *sig_ptr as the kernel starts (pacing of the helper stream's
    // decoder-cell chain: t2s_launch_pace_wait, csrc/tacotron_ops.h)
    unsigned* sig_ptr; unsigned sig_val;
};

struct BnBwdArgs {
    const float* x;              // conv output [B][C][T] (pre-BN)
    const float* mean; const float* var; const float* gamma; const float* beta; float eps;
    const float* dout_f32;       // [B][C][T] or null
    const u16* dout_hi; const u16* dout_lo;   // planes or null
    const unsigned char* mask; float mask_scale; int act;
    float* dgamma; float* dbeta; // [C]
    u16* dx_hi; u16* dx_lo;      // planes out
    int B, C, T, Lp, halo;
};

hipError_t t2s_launch_rows_to_tm(const float* x, long ld, int items, int items_pad, int shift, int C, u16* dst_hi,
                                 u16* dst_lo, int Npad, int n_off, hipStream_t stream);
hipError_t t2s_launch_rows_to_tm_batched(const float* x, long ld, long x_bstride, int items, int items_pad, int shift, int C,
                                         u16* dst_hi, u16* dst_lo, long dst_bstride, int Npad, int n_off, int nb,
                                         hipStream_t stream);
hipError_t t2s_launch_lstm_cell_bwd(const LstmBwdArgs& a, hipStream_t stream);
hipError_t t2s_launch_relu_drop_bwd(const float* dy, const float* y, float scale, size_t n, float* dz, hipStream_t stream);
hipError_t t2s_launch_att_bwd(const AttBwdArgs& a, hipStream_t stream);
// the same in two parts: (d_w, energies) and (location-conv backward: carries for step t-1, kernel gradient, d_q fold)
hipError_t t2s_launch_att_bwd_front(const AttBwdArgs& a, hipStream_t stream);
hipError_t t2s_launch_att_bwd_conv(const AttBwdArgs& a, hipStream_t stream);
// the three parts in one launch (needs ctx, the second carry pair, dctx_out; d_q is left as per-chunk partials in dq_part)
bool t2s_att_bwd_fused_ok(const AttBwdArgs& a);
hipError_t t2s_launch_att_bwd_fused(const AttBwdArgs& a, hipStream_t stream, const AttBwdFoldArgs* fold = nullptr);
hipError_t t2s_launch_bn_bwd(const BnBwdArgs& a, double* partial, hipStream_t stream);
hipError_t t2s_launch_planes_to_f32(const u16* X_hi, const u16* X_lo, int B, int C, int L, int Lp, int halo, float* out,
                                    int accumulate, hipStream_t stream);
hipError_t t2s_launch_sum_axis0(const float* in, int n0, int n, float* out, hipStream_t stream);
hipError_t t2s_launch_add3(const float* a, const float* b, const float* c, size_t n, float* out, hipStream_t stream);
hipError_t t2s_launch_scale_by_scalar(const float* in, size_t n, const float* scalar, float mul, float* out, hipStream_t stream);
hipError_t t2s_launch_lstm_seq_bwd_split(const float* d_out, const float* out, const float* gates, const float* csave,
                                         const float* whh_f, const float* whh_r, const int* lengths, float* dgx, float* hprev,
                                         int B, int T, int T_out, unsigned long long* xbuf, unsigned epoch, hipStream_t stream);
hipError_t t2s_launch_lstm_seq_bwd(const float* d_out, const float* out, const float* gates, const float* csave,
                                   const float* whh_f, const float* whh_r, const int* lengths, float* dgx, float* hprev,
                                   int B, int T, int H, int T_out, hipStream_t stream);
hipError_t t2s_launch_rows_to_planes(const float* x, int B, int T, int C, int Lp, int halo, u16* X_hi, u16* X_lo,
                                     hipStream_t stream);
hipError_t t2s_launch_embedding_grad(const long* ids, const u16* D_hi, const u16* D_lo, int B, int T, int E, int V, int Lp,
                                     int halo, float* d_emb, hipStream_t stream);

// loss_ops.hip
hipError_t t2s_launch_taco_loss(const float* mel, const float* post, const float* target, size_t n_mel, const float* gate,
                                const float* gate_t, size_t n_gate, float* d_mel, float* d_post, float* d_gate, double* partial,
                                float* out, hipStream_t stream);
hipError_t t2s_launch_waveglow_loss(const float* z, size_t n_z, const float* const* log_s, const size_t* n_log_s, int n_flows,
                                    const float* log_det, float sigma, float* d_z, double* partial, float* out,
                                    hipStream_t stream);
