// Internal argument blocks of the Tacotron-2 kernels (tacotron_ops.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <mutex>

// Library-owned helper stream (one set per device, t2s_api_taco_bwd.hip): `lock` is held while a call enqueues on it.
struct T2sHelperStream {
    std::unique_lock<std::mutex> lock;
    hipStream_t side = nullptr;
    hipEvent_t ev_step = nullptr, ev_join = nullptr;
};
hipError_t t2s_helper_stream_acquire(T2sHelperStream& h);

struct GemvArgs {
    const float* W1; int ld1; int k1;     // weight row = [W1[row][:k1] | W2[row][:k2]]
    const float* W2; int ld2; int k2;
    const float* x1; int n1; long sx1;    // input vector = [x1 | x2 | x3], per-item strides in floats
    const float* x2; int n2; long sx2;
    const float* x3; int n3; long sx3;
    const float* bias1;
    const float* bias2;
    float* y; long sy_item; long sy_row;  // y[item*sy_item + row*sy_row]
    int rows, items;
    int act;
    const unsigned char* mask; long smask_item; float mask_scale;
    // optional second row block (rows >= split_row, split_row > 0): its own output, activation and mask
    int split_row;
    float* y2; long sy2_item; long sy2_row;
    int act2;
    const unsigned char* mask2; long smask2_item; float mask2_scale;
    // small-batch GEMM only: the 96 KB operand ring (64-byte fragment rows) instead of the 144 KB one, so that a 61 KB workgroup of
    // another stream's kernel (att_bwd_fused_kernel) fits on the same CU - for launches on a helper stream with slack
    int narrow_ring;
    // small-batch GEMM only: never the 16-items-per-workgroup form (helper-stream launches: fewer, longer workgroups cost less CU time)
    int no_half;
};

struct LstmCellArgs {
    const float* W_ih;   // [4H][n1+n2]
    const float* W_hh;   // [4H][H]
    const float* b_ih;
    const float* b_hh;
    const float* x1; int n1; long sx1;
    const float* x2; int n2; long sx2;
    const float* h_in;   // [B][H]
    float* h_out;        // [B][H]
    float* c;            // [B][H], in place
    float* h_copy; long s_copy;           // optional second copy of h_out (stride per item)
    const unsigned char* drop_mask; float drop_scale;   // optional dropout on h_out ([B][H] of 0/1)
    int B, H;
    float* gates_out;    // optional [B][4H]: post-activation i,f,g,o (training: kept for the backward pass)
    float* c_out;        // optional [B][H]: new cell state copy
    // optional: per-workgroup partial attention queries q_part[wg][b][a] = sum_{u in wg} w_q[a][u] * h_out[b][u]
    const float* w_q; float* q_part; int q_dim;
    // streamed-gates form (B <= 8 autoregressive decode, GateStreamArgs below): the products of W_hh (and of some W_ih columns)
    // with vectors that were known one launch earlier arrive as pre-activation partials [B][4H] and are simply added;
    // h_in == NULL then drops the W_hh . h term from this kernel, ld_ih > 0 is the row stride of W_ih when only a column
    // range of it is multiplied here (W_ih then points at the first of those columns)
    const float* pre_a; const float* pre_b; int ld_ih;
    // folded prenet layer 1 (autoregressive B <= 8, with the streamed-gates form): x1 is not read; instead every workgroup
    // recomputes x1[b][r] = relu(sum_k w_p2[k][r] * p1[b][k]) * (p2_mask[b * s_p2_mask + r] ? p2_scale : 0), r < n1 = 256
    // (modules.py:19-22, the prenet's second Linear + ReLU + always-on dropout; w_p2 = the TRANSPOSED weight, walked sparsely:
    // lstm_cell_p2_kernel)
    const float* w_p2; const float* p1; const unsigned char* p2_mask; long s_p2_mask; float p2_scale;
    // matrix-core form only: the first thread of the launch stores sig_val to *sig_ptr (agent scope) as the kernel STARTS - i.e. when
    // everything in front of it on its stream has completed.  A helper stream's t2s_launch_pace_wait on that word then releases work
    // that should run beside what FOLLOWS this launch (the teacher-forced decoder cell beside the attention launch).
    unsigned* sig_ptr; unsigned sig_val;
};
// one wave that polls *flag (bounded) until it has reached `val`; the launches behind it on `stream` start then.  err: a word that is
// raised if the wait expires (may be null)
hipError_t t2s_launch_pace_wait(const unsigned* flag, unsigned val, unsigned long long* err, hipStream_t stream);
hipError_t t2s_launch_pace_signal(unsigned* flag, unsigned val, hipStream_t stream);

// Role-specialised second half of the fused attention launch (B <= 8): while ONE workgroup per batch element runs the
// attention of step t (10.7 us with 255 CUs and HBM idle), the other workgroups stream the three [4H][H] weight blocks whose
// input vectors already exist - W_hh_dec . h_dec(t-1), W_ih_dec[:, :A] . h_att(t) (both for this step's decoder cell) and
// W_hh_att . h_att(t) (next step's attention cell) - and leave the products as partials; the two cell launches then stream
// 8.4 / 12.6 MB instead of 42 / 29.4 MB.  The consumers are the NEXT launches: the kernel boundary stays the only barrier.
struct GateStreamArgs {
    const float* W0; int ld0; const float* x0; float* out0;   // out0[b][r] = W0[r][:H] . x0[b][:H], r < rows
    const float* W1; int ld1; float* out1;                    // out1[b][r] = W1[r][:H] . x12[b][:H]
    const float* W2; int ld2; float* out2;                    // out2[b][r] = W2[r][:H] . x12[b][:H]
    const float* x12;
    int rows, H, B;                                           // rows = 4H of each block, H = 1024
    int dbg;                                                  // -DT2S_ATTSTREAM_ABLATE builds only: 1 = attention role returns at once,
                                                              // 2 = gate-stream role returns at once (timing only, results garbage)
};

struct AttArgs {
    const float* q;            // [B][att_dim]
    const float* w_loc_conv;   // [F][2][KS]
    const float* w_loc_dense;  // [att_dim][F]
    const float* w_v;          // [att_dim]
    const float* pmem;         // [B][T][att_dim]
    const float* memory;       // [B][T][enc_dim]
    const int* lengths;        // [B] or null
    float* w_prev;             // [B][T]  in: previous weights, out: new weights
    float* w_cum;              // [B][T]
    float* energies;           // [B][T] scratch
    float* ctx;                // [B][enc_dim]
    float* ctx_copy; long s_ctx_copy;
    float* align_out; long s_align_b;
    int B, T, att_dim, enc_dim, loc_f, loc_ks;
    // fused single-launch form (small batch): query computed in the same kernel
    const float* w_query;      // [att_dim][att_rnn]
    const float* h_att;        // [B][att_rnn]
    const float* w_loc_denseT; // [F][att_dim]
    int att_rnn;
    const float* q_part;       // [n_part][B][att_dim] partial queries from lstm_cell_kernel (instead of w_query . h_att)
    int n_part;
    float* q_out;              // large-batch form with q_part: the summed query [B][att_dim] is written here (by the first chunk's workgroup)
    float* q_save;             // optional [B][att_dim]: query of this step (training)
    // B <= 8 autoregressive decode (ABI v4 t2s_taco_decoder::ploc): the location term P_loc[b][t][a] = (location_dense o location_conv)
    // of the CURRENT (w, w_cum), computed one launch earlier beside the projection GEMV (LocPreArgs) - the fused attention
    // launch then skips both matrix-core stages (the exact-f32 MFMAs of T x 128 x 94 MACs on ONE CU: 4 us at 128 encoder
    // positions, growing linearly) and only adds q and the processed memory
    const float* ploc;
    float* wcum_save;          // optional [B][T]: cumulative weights after this step (training)
    int tile_major;            // att_energy_mfma_kernel: block -> (tile, item) in launch order instead of item-per-XCD (A/B: T2S_ENERGY_XCD=0)
    // one-launch form of energies + softmax + context (t2s_launch_att_energy with xbuf set; t2s_att_energy_ctx_ok says whether the
    // shape is covered): [B][T] 8-byte granules + 1 error word, zero before step 0; tag = step + 1 (never 0)
    unsigned long long* xbuf;
    unsigned tag;
};
bool t2s_att_energy_ctx_ok(const AttArgs& a);

// Role-specialised second half of the projection launch (B <= 8 autoregressive decode): while 85 workgroups run the 337 x 1536 GEMV,
// T/16 x B more compute the location features of the NEXT step's attention from the weights this step's attention just wrote.
struct LocPreArgs {
    const float* w; const float* w_cum;       // [B][T] attention weights / cumulative weights after this step
    const float* w_loc_conv;                  // [32][2][KS]
    const float* w_loc_denseT;                // [32][128]
    float* ploc;                              // [B][T][128]
    int B, T, loc_ks;
    int n_gemv_blocks;                        // blocks [0, n_gemv_blocks) of the launch run the GEMV
};
hipError_t t2s_launch_gemv(const GemvArgs& a, hipStream_t stream);
hipError_t t2s_launch_gemv_with_loc(const GemvArgs& a, const LocPreArgs& lp, hipStream_t stream);
// sbgemm.hip: the same two operators on the f32 matrix cores for 9+ items (picked inside t2s_launch_gemv / _lstm_cell)
bool t2s_sbgemm_plain_ok(const GemvArgs& a);
hipError_t t2s_launch_sbgemm_plain(const GemvArgs& a, hipStream_t stream);
bool t2s_sbgemm_lstm_ok(const LstmCellArgs& a);
hipError_t t2s_launch_sbgemm_lstm(const LstmCellArgs& a, hipStream_t stream);
hipError_t t2s_launch_lstm_cell(const LstmCellArgs& a, hipStream_t stream);
hipError_t t2s_launch_att_energy(const AttArgs& a, hipStream_t stream);
hipError_t t2s_launch_att_softmax_ctx(const AttArgs& a, hipStream_t stream);
hipError_t t2s_launch_att_fused(const AttArgs& a, hipStream_t stream, const GateStreamArgs* gs = nullptr);
bool t2s_att_fused_stream_ok(const AttArgs& a, const GateStreamArgs& g);
hipError_t t2s_launch_lstm_seq_split(const float* gx, const float* whhT_f, const float* whhT_r, const int* lengths, float* out,
                                     int B, int T, int T_out, float* gates_save, float* c_save, unsigned long long* xbuf,
                                     unsigned epoch, hipStream_t stream);
hipError_t t2s_launch_lstm_seq(const float* gx, const float* whhT_f, const float* whhT_r, const int* lengths, float* out,
                               int B, int T, int H, int T_out, float* gates_save, float* c_save, hipStream_t stream);
hipError_t t2s_launch_transpose(const float* in, float* out, int R, int C, hipStream_t stream);
hipError_t t2s_launch_embed_planes(const long* ids, const float* emb, int B, int T, int E, int V, int Lp, int halo,
                                   unsigned short* X_hi, unsigned short* X_lo, hipStream_t stream);
hipError_t t2s_launch_f32_to_planes(const float* x, int B, int C, int L, int Lp, int halo, unsigned short* X_hi,
                                    unsigned short* X_lo, hipStream_t stream);
hipError_t t2s_launch_parse_output(float* mel, float* mel_post, float* gate, const int* lengths, int B, int n_mel, int T,
                                   hipStream_t stream);
hipError_t t2s_launch_bn_fold(const float* gamma, const float* beta, const float* mean, const float* var,
                              const float* conv_bias, float eps, int C, float* scale, float* bias_out,
                              hipStream_t stream);
hipError_t t2s_launch_bernoulli_mask(unsigned char* mask, size_t n, unsigned long long seed, unsigned long long offset,
                                     float keep_prob, hipStream_t stream);
hipError_t t2s_launch_stop_check(const float* gate_out, int B, int s_gate_b, int step0, int n, float threshold,
                                 int* stop_step, hipStream_t stream);
hipError_t t2s_launch_bn_train(const float* x, const float* gamma, const float* beta, float eps, int act,
                               const unsigned char* mask, float mask_scale, int B, int C, int T, int Lp, int halo,
                               float* mean, float* var, unsigned short* O_hi, unsigned short* O_lo, float* out_f32,
                               hipStream_t stream);
hipError_t t2s_launch_bn_running_update(const float* mean, const float* var, float* running_mean, float* running_var,
                                        long long* num_batches_tracked, float momentum, float unbias, int C, hipStream_t stream);
hipError_t t2s_launch_zero_fill(void* p, size_t bytes, hipStream_t stream);
