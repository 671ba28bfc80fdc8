// Internal launcher prototypes shared by the .hip translation units and the C-ABI layer.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef unsigned short u16;

enum { EPI_GATE = 0, EPI_RESSKIP = 1, EPI_BIAS_ACT = 2, EPI_GATE_BWD = 3 };
enum { ACT_NONE = 0, ACT_RELU = 1, ACT_TANH = 2 };
enum { PERM_NONE = 0, PERM_GATE = 1, PERM_PAIR8 = 2 };

struct ConvGemmArgs {
    const u16* A_hi;   // packed weights [nk][Mpad][32] bf16
    const u16* A_lo;
    const u16* X_hi;   // input planes [B][xc][Lp][32] bf16
    const u16* X_lo;
    const u16* S_hi;   // second (conditioning) input planes [B][sc][Lp][32], or null
    const u16* S_lo;
    const float* bias; // [Mpad], packed row order
    u16* O_hi;         // output planes [B][oc][Lp][32] (GATE: acts; RESSKIP: x, updated in place)
    u16* O_lo;
    float* skip;       // RESSKIP: f32 skip accumulator planes [B][oc][Lp][32]
    float* out_f32;    // BIAS_ACT: optional [B][C][L] f32 copy
    int nk;            // total K-steps = taps*xc + sc
    int nk_x;          // taps*xc
    int xc, sc, oc;    // 32-channel chunks of X, S and the output
    int xbs;           // chunks between batch entries of the X planes (0: xc) - X may be a slice of a wider plane set
    int taps, dil;
    int Mpad, Lp, halo, L, B;
    int n_mtiles, n_ttiles;
    int C;             // valid output channels
    int n_res;         // RESSKIP: packed rows < n_res are the residual half
    int skip_init;     // RESSKIP: 1 = store, 0 = accumulate
    int act;           // BIAS_ACT
    int f32_cl;        // BIAS_ACT: out_f32 is [B][L][C] instead of [B][C][L]
    // training
    u16* T_hi;         // GATE: optional saves of tanh / sigmoid values (planes, C channels); GATE_BWD: inputs
    u16* T_lo;
    u16* G_hi;
    u16* G_lo;
    int tc;            // chunks of the T/G planes
    int pair8;         // RESSKIP: the residual rows were packed with PERM_PAIR8 (a lane's two m-tiles hold 8 consecutive channels)
    int res_init;      // RESSKIP: 1 = residual half stores (x = acc + bias), 0 = accumulates
    const u16* R_hi;   // RESSKIP: residual source planes (null: read the output planes, i.e. in place)
    const u16* R_lo;
    long a_bstride;    // elements between the A operands of consecutive batch entries (0: shared weights)
    int ksplit;        // weight-gradient GEMMs: K-range splits per batch entry (0/1 = none); grid batch index = b*ksplit + s
    int k0, kchunk, kend;   // split s covers K-steps [k0 + s*kchunk, min(k0 + (s+1)*kchunk, kend))
    int kflat;         // >0: K runs over (batch entry, K-step in [k0, kend)) flattened, kflat = number of batch entries; grid batch
                       // index = slab s covering flattened steps [s*kchunk, (s+1)*kchunk): the split count is free of the batch
    // GATE with WN.end folded in: fold_acc[slot][b][j][t] (+)= sum_c fold_w[c][j] * acts[c][t]
    const u16* fold_A;     // (W_end . W_skip_i) as MFMA A fragments [mt][wr][pair][hi,lo][lane][8] (endfold_weights_kernel)
    float* fold_acc;       // [2*n_mtiles][B][8][L]
    int fold_init;         // 1: store, 0: accumulate (first / later layers of a flow)
    // gate GEMM, phase mode (vocoder inference with composed conditioning weights): ph_P > 0 selects it
    const u16* A2_hi;      // composed conditioning weights [ph_P][sc][Mpad][32]
    const u16* A2_lo;
    int ph_P;              // phases = hop / n_group plane rows per mel frame
    int ph_FT;             // frames per batch entry in a 256-column tile (64, 128 or 256)
    int ph_bper;           // batch entries per tile = 256 / ph_FT
    int ph_nft;            // frame tiles per batch entry
    int ph_Fp;             // rows of the mel-window planes S[B][sc][ph_Fp][32]
    int dbg;               // timing-only ablations (env T2S_DBG_GEMM): 1 = no DMA in the K loop, 2 = no MFMA; results are wrong
};

hipError_t t2s_launch_conv_gemm(const ConvGemmArgs& a, int epi, hipStream_t stream, int mt_rows = 256);
// EPI_GATE, 256 x 256 tile, plain K order (nk = taps * xc + sc): the ping-pong schedule of csrc/gate_gemm_pp.hip
hipError_t t2s_launch_gate_gemm_pp(const ConvGemmArgs& a, hipStream_t stream);
// EPI_RESSKIP (accumulate form: n_res = every row, C = 0) / EPI_GATE_BWD on the same schedule; a.n_mtiles counts 256-row tiles
hipError_t t2s_launch_bwd_gemm_pp(const ConvGemmArgs& a, int epi, hipStream_t stream);
bool t2s_pp_shape_ok(const ConvGemmArgs& a);

// csrc/wgrad_cl.hip: weight-gradient GEMM straight from channel-last planes (transposed LDS reads)
struct WgradChunk {            // one 32-channel chunk of an operand; mirrors t2s_wgrad_chunk in include/t2s_hip.h
    const u16* hi;             // row 0 of this chunk for batch entry 0 (chunk index and any tap shift folded in)
    const u16* lo;
    long bstride;              // u16 elements between batch entries (0: the same rows for every batch entry)
};
struct WgradClArgs {
    const WgradChunk* a_chunks;    // [n_mtiles * 8]  M side (output rows)
    const WgradChunk* b_chunks;    // [n_ntiles * 8]  N side (output columns)
    float* P;                      // [nslab][M][ldp]
    int M, N, ldp, n_mtiles, n_ntiles;   // ldp >= N: floats per output row (a multiple of 4 selects the ping-pong kernel)
    int B, k0, k1;                 // K-blocks of 32 plane rows [k0, k1) of every batch entry
    int nslab, kchunk;             // slab s covers flattened (batch, block) steps [s * kchunk, (s + 1) * kchunk)
    int bias_cols;                 // 1: also write the row sums of the M-side operand (4 partial sums) to columns N .. N+3
};
hipError_t t2s_launch_wgrad_cl(const WgradClArgs& a, hipStream_t stream);

struct PackArgs {
    const float* v;        // [O][Cin][Kt]
    const float* g;        // [O] weight-norm gain, or null for a plain weight
    const float* bias_in;  // [O] or null
    u16* A_hi;             // [nk][Mpad][32]
    u16* A_lo;
    float* bias_out;       // [Mpad] packed order, or null
    int O, Cin, Kt;
    int perm, C_gate;      // PERM_GATE: rows o<C_gate are tanh rows, o>=C_gate sigmoid rows
    int Mpad;
    int koff;              // first packed k index of this block (multiple of 32)
    int Cin_pad;           // tap stride in packed k (Cin rounded up to 32)
    int bias_accumulate;   // 1: bias_out[p] += bias_in[o]
    int row_off;           // PERM_NONE: packed row = o + row_off
    int g_is_scale;        // 1: g[o] is a plain per-row scale (BatchNorm fold), 0: weight-norm gain
};
hipError_t t2s_launch_pack(const PackArgs& a, hipStream_t stream);
// table-driven: one launch packs every weight of the model; jobs live in device memory
struct PackJob {               // mirrors t2s_pack_job in include/t2s_hip.h (all 8-byte fields)
    const float* v; const float* g; const float* bias_in; const float* bias_in2;
    u16* A_hi; u16* A_lo; float* bias_out;
    long row_start;            // first workgroup of this job (prefix sum of ceil(O/16): 16 rows per workgroup)
    long O, Cin, Kt, perm, C_gate, Mpad, koff, Cin_pad, row_off, g_is_scale;
    float* scale_out;          // optional [O]: the per-row factor applied (g/|v|), kept for the backward pass
};
hipError_t t2s_launch_pack_table(const PackJob* jobs, int n_jobs, long total_rows, hipStream_t stream);
hipError_t t2s_launch_weightnorm_small(const float* v, const float* g, int O, int K, float* w, hipStream_t stream);

hipError_t t2s_launch_upbasis_planes(const float* W, const float* bias, int n_mel, int ksize, int stride, int n_group, int Lp,
                                     int halo, u16* U_hi, u16* U_lo, hipStream_t stream);
hipError_t t2s_launch_compose_pack(const float* tmp, const float* bias_in, int rows, int Mpad, int P, int K2, int ld, u16* A2_hi,
                                   u16* A2_lo, float* bias_out, hipStream_t stream);
hipError_t t2s_launch_melwin_planes(const float* mel, int B, int n_mel, int frames, int nlag, int Fp, u16* S_hi, u16* S_lo,
                                    hipStream_t stream);
hipError_t t2s_launch_upsample_squeeze(const float* mel, const float* W, const float* bias, int B, int n_mel,
                                       int frames, int ksize, int stride, int n_group, int L, int Lp, int halo,
                                       u16* S_hi, u16* S_lo, hipStream_t stream);
hipError_t t2s_launch_audio_squeeze(const float* audio, float* z, int B, int T, int n_group, int L, int unsqueeze,
                                    hipStream_t stream);
hipError_t t2s_launch_convinv(float* z, const float* W, int B, int n_group, int c_off, int n_rem, int L,
                              hipStream_t stream);
struct SmallMatJob { const float* W; float* logdet_out; float* inv_out; long n; };   // mirrors t2s_small_mat_job
hipError_t t2s_launch_small_logdet_batch(const SmallMatJob* jobs, int n_jobs, float scale, hipStream_t stream);
hipError_t t2s_launch_small_logdet_batch_host(const SmallMatJob* host_jobs, int n_jobs, float scale, hipStream_t stream);
hipError_t t2s_launch_small_logdet_inv(const float* W, int n, float scale, float* logdet_out, float* inv_out,
                                       hipStream_t stream);
hipError_t t2s_launch_start(const float* z, const float* w, const float* bias, int B, int n_group, int c_off,
                            int n_half, int C, int L, int Lp, int halo, u16* X_hi, u16* X_lo, hipStream_t stream);
struct EndFoldJob {        // one WN layer: fold_w[c][j] = sum_o W_end[j][o] * scale[o] * v_skip[o][c]
    const float* w_end;    // [nj][C]
    const float* v_skip;   // [C][C] skip rows of res_skip_layers[i].weight_v (or weight)
    const float* scale;    // [C] per-row weight-norm scale of those rows
    const float* b_skip;   // [C]
    u16* fold_A;           // [C/128][2][2][2][64][8] bf16: MFMA A fragments (rows j, K permuted to the accumulator layout)
    float* bes;            // [8]: W_end . b_skip
    long nj, C;
};
hipError_t t2s_launch_endfold_weights(const EndFoldJob* jobs, int n_jobs, int C, hipStream_t stream);
hipError_t t2s_launch_end_fold_affine(const float* fold_acc, int nslots, const float* bes, int n_layers,
                                      const float* b_end, float* z, float* log_s, float* wn_out, int B, int n_group, int c_off,
                                      int n_half, int L, int reverse, hipStream_t stream);
hipError_t t2s_launch_end_affine(const float* skip, const float* w_end, const float* b_end, float* z, float* log_s,
                                 float* wn_out, int B, int n_group, int c_off, int n_half, int C, int L, int Lp, int halo,
                                 int reverse, hipStream_t stream);
