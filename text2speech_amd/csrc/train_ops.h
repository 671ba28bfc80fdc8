// Internal argument blocks / launchers of the training kernels (train_ops.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef unsigned short u16;

struct WnBwdArgs {
    const float* P;        // partial slabs [nsplit][Prows][Pcols] f32
    const float* v;        // [O][Cin][Kt]
    const float* g;        // [O] or null (plain weight)
    float* dv;             // [O][Cin][Kt]
    float* dg;             // [O] (when g)
    float* db;             // [O] or null
    int nsplit, Prows, Pcols;
    int row_off, col_off, tap_stride, col_bias;
    int O, Cin, Kt;
    int db_accum;
    int n_bias_cols;       // the bias gradient is the sum of this many columns starting at col_bias (>= 1)
};

struct SmallWgradArgs {
    const u16* P_hi; const u16* P_lo; const float* P_f32;   // planes [B][chunks][Lp][32]
    const float* Q;        // [B][Jtot][L]
    float* out;            // [R][J] or [J][R]
    float* rowsum;         // [R] or null
    float* scratch;        // [B*8][chunks*32][17] partial sums
    int B, chunks, Lp, halo, L, R, J, Jtot, q_off, out_transposed;
};

struct AdamJob {           // mirrors t2s_adam_job (8-byte fields)
    float* p; const float* g; float* m; float* v;
    long n;
    long blk_start;        // first block of this job; each block covers 1024 elements
};

hipError_t t2s_launch_plane_transpose(const u16* src_hi, const u16* src_lo, int B, int src_chunks, int n_chunks, int Lp,
                                      int shift, u16* dst_hi, u16* dst_lo, int Npad, int n_off, hipStream_t stream);
hipError_t t2s_launch_tm_ones_row(u16* dst_hi, u16* dst_lo, int B, int Lp, int halo, int L, int Npad, int n_row,
                                  hipStream_t stream);
hipError_t t2s_launch_pack_transposed(const float* v, const float* scale, int O, int Cin, int Kt, int flip, int O_pad,
                                      int Mpad, int koff, u16* A_hi, u16* A_lo, int pair8, hipStream_t stream);
hipError_t t2s_launch_wn_backward(const WnBwdArgs& a, hipStream_t stream);
hipError_t t2s_launch_affine_backward(float* z, float* dz, const float* wn_out, const float* g_ls, int g_ls_scalar, float* d_out,
                                      int B, int G, int c_off, int nh, int L, hipStream_t stream);
hipError_t t2s_launch_small_wgrad(const SmallWgradArgs& a, hipStream_t stream);
size_t t2s_small_wgrad_scratch_floats(int B, int chunks);
hipError_t t2s_launch_rows_sum(const float* Q, int B, int Jtot, int q_off, int J, int L, float* out, hipStream_t stream);
hipError_t t2s_launch_start_dgrad(const u16* X_hi, const u16* X_lo, const float* w, float* dz, int B, int G, int c_off,
                                  int nh, int C, int L, int Lp, int halo, hipStream_t stream);
hipError_t t2s_launch_convinv_wgrad(const float* dz, const float* zin, const float* Winv, const float* gscale_ptr,
                                    float gmul, int B, int G, int c_off, int n, int L, float* dW, hipStream_t stream);
hipError_t t2s_launch_upsample_wgrad(const u16* D_hi, const u16* D_lo, const float* mel, int B, int M, int F, int ksize,
                                     int stride, int G, int L, int Lp, int halo, float* dW, hipStream_t stream);
hipError_t t2s_launch_upsample_bgrad(const u16* D_hi, const u16* D_lo, int B, int M, int G, int L, int Lp, int halo,
                                     float* db, hipStream_t stream);
hipError_t t2s_launch_adam_table(const AdamJob* jobs, int n_jobs, long total_blocks, float lr, float b1, float b2,
                                 float eps, float bc1, float bc2_sqrt, float gscale, float weight_decay,
                                 hipStream_t stream);
