// Probe of v_mfma_scale_f32_16x16x128_f8f6f4 on gfx950 (round 3, numerics study follow-up): operand lane maps with exact integer
// data, the E8M0 scale semantics, and the issue rate against v_mfma_f32_16x16x32_{bf16,f16} - the three facts a kernel built on
// "fp16 main product + two fp8 cross products" (profiles/r03_numerics.md) needs.  Standalone: hipcc -O3 --offload-arch=gfx950.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>

typedef __attribute__((ext_vector_type(8))) int v8i;
typedef __attribute__((ext_vector_type(4))) float v4f;
typedef __attribute__((ext_vector_type(8))) _Float16 h8;
typedef __attribute__((ext_vector_type(8))) __bf16 b8;

// e4m3fn encode of small non-negative integers 0..15 (exact)
static unsigned char f8_of_int(int v) {
    if (v == 0) return 0;
    int e = 0, m = v;
    while (m >= 2) { m >>= 1; ++e; }                  // v = 1.xxx * 2^e
    const int frac = ((v << 3) >> e) & 7;             // 3 mantissa bits (exact for v < 16)
    return (unsigned char)(((e + 7) << 3) | frac);
}

// D[i][n] for the hypothesis: lane l holds row/col (l & 15) and k = 32 * (l >> 4) + j, byte j of its 32 bytes
__global__ void k_probe(const unsigned char* A, const unsigned char* B, float* D, int scale_a, int scale_b) {
    const int l = threadIdx.x;
    v8i a, b;
    const int r = l & 15, g = l >> 4;
    unsigned char ab[32], bb[32];
    for (int j = 0; j < 32; ++j) {
        ab[j] = A[r * 128 + 32 * g + j];              // A[i][k] row-major 16 x 128
        bb[j] = B[(32 * g + j) * 16 + r];             // B[k][n] row-major 128 x 16
    }
    memcpy(&a, ab, 32);
    memcpy(&b, bb, 32);
    v4f c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 0, 0, 0, scale_a, 0, scale_b);
    for (int e = 0; e < 4; ++e) D[(4 * (l >> 4) + e) * 16 + (l & 15)] = c[e];       // C/D: col = lane & 15, row = 4 (lane >> 4) + reg
}

template <int KIND>
__global__ void k_rate(float* out, int iters) {
    v4f acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = (v4f){0.f, 0.f, 0.f, 0.f};
    const int l = threadIdx.x;
    v8i a8, b8_;
    for (int i = 0; i < 8; ++i) { a8[i] = 0x38383838 + l; b8_[i] = 0x30303030 + i; }
    h8 ah, bh;
    b8 ab, bb;
    for (int i = 0; i < 8; ++i) { ah[i] = (_Float16)(0.01f * (l + i)); bh[i] = (_Float16)(0.02f * i); ab[i] = (__bf16)(0.01f * (l + i)); bb[i] = (__bf16)(0.02f * i); }
    const long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (KIND == 0) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ab, bb, acc[i], 0, 0, 0);
            else if (KIND == 1) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, acc[i], 0, 0, 0);
            else acc[i] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a8, b8_, acc[i], 0, 0, 0, 127, 0, 127);
        }
    }
    const long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][3];
    if (threadIdx.x == 0) { out[0] = s; out[1] = (float)(t1 - t0) / (8.0f * iters); }
}

int main() {
    unsigned char hA[16 * 128], hB[128 * 16];
    srand(3);
    for (int i = 0; i < 16 * 128; ++i) hA[i] = f8_of_int(rand() % 8);
    for (int i = 0; i < 128 * 16; ++i) hB[i] = f8_of_int(rand() % 8);
    auto val = [](unsigned char c) { if (!c) return 0.0; const int e = (c >> 3) & 15, m = c & 7; return ldexp(1.0 + m / 8.0, e - 7); };
    double ref[16][16];
    for (int i = 0; i < 16; ++i)
        for (int n = 0; n < 16; ++n) {
            double s = 0;
            for (int k = 0; k < 128; ++k) s += val(hA[i * 128 + k]) * val(hB[k * 16 + n]);
            ref[i][n] = s;
        }
    unsigned char *dA, *dB;
    float* dD;
    hipMalloc(&dA, sizeof(hA)); hipMalloc(&dB, sizeof(hB)); hipMalloc(&dD, 16 * 16 * 4 + 64);
    hipMemcpy(dA, hA, sizeof(hA), hipMemcpyHostToDevice);
    hipMemcpy(dB, hB, sizeof(hB), hipMemcpyHostToDevice);
    float hD[256];
    const int scales[4][2] = {{127, 127}, {126, 127}, {127, 129}, {0x7f7f7f7f, 0x7f7f7f7f}};
    for (int s = 0; s < 4; ++s) {
        hipLaunchKernelGGL(k_probe, dim3(1), dim3(64), 0, 0, dA, dB, dD, scales[s][0], scales[s][1]);
        hipMemcpy(hD, dD, sizeof(hD), hipMemcpyDeviceToHost);
        double worst = 0, ratio = 0;
        for (int i = 0; i < 16; ++i)
            for (int n = 0; n < 16; ++n) {
                worst = fmax(worst, fabs(hD[i * 16 + n] - ref[i][n]));
                if (ref[i][n] != 0) ratio = hD[i * 16 + n] / ref[i][n];
            }
        printf("scale_a=0x%x scale_b=0x%x: max |D - ref| = %g, D/ref (last) = %g, D[0][0]=%g ref=%g\n", scales[s][0], scales[s][1], worst,
               ratio, hD[0], ref[0][0]);
    }
    float hO[2];
    const char* names[3] = {"v_mfma_f32_16x16x32_bf16", "v_mfma_f32_16x16x32_f16", "v_mfma_scale_f32_16x16x128_f8f6f4 (e4m3)"};
    for (int kind = 0; kind < 3; ++kind) {
        for (int rep = 0; rep < 2; ++rep) {
            if (kind == 0) hipLaunchKernelGGL(k_rate<0>, dim3(1), dim3(64), 0, 0, dD, 20000);
            if (kind == 1) hipLaunchKernelGGL(k_rate<1>, dim3(1), dim3(64), 0, 0, dD, 20000);
            if (kind == 2) hipLaunchKernelGGL(k_rate<2>, dim3(1), dim3(64), 0, 0, dD, 20000);
            hipDeviceSynchronize();
        }
        hipMemcpy(hO, dD, 8, hipMemcpyDeviceToHost);
        printf("%s: %.1f cycles per instruction (one wave, 8 independent accumulators)\n", names[kind], hO[1]);
    }
    return 0;
}
