// Microbenchmark: how fast can 256 workgroups x 8 waves pull a [rows][K] f32 weight matrix with
//   P1  the sbgemm operand pattern: per load instruction 16 rows x 64 contiguous bytes, rows K*4 bytes apart
//   P2  a packed layout: per load instruction 1 KB contiguous, consecutive K-steps contiguous
// hot = the same matrix every launch (L2 / Infinity-Cache resident), cold = rotating over > 512 MB of matrices.
// Build: hipcc --offload-arch=gfx950 -O3 tools/microbench/stream_patterns.hip -o /tmp/stream_patterns
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int PACKED, int WITHX>
__global__ __launch_bounds__(512) void pull(const float* __restrict__ W, int rows, int K, float* sink, const float* __restrict__ X) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int tile = blockIdx.x;                 // 16 rows
    const int nsteps = K / 16, s0 = wave * nsteps / 8, s1 = (wave + 1) * nsteps / 8;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int s = s0; s < s1; s += 4) {
        f32x4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int st = s + u;
            if (st < s1) {
                const float* p = PACKED ? W + ((size_t)tile * nsteps + st) * 256 + lane * 4
                                        : W + (size_t)(tile * 16 + (lane & 15)) * K + st * 16 + (lane >> 4) * 4;
                v[u] = *(const f32x4*)p;
                if (WITHX) {     // the 32 input vectors of sbgemm: items lane%16 and 16 + lane%16, same k columns
                    const float* xp = X + (size_t)(lane & 15) * K + st * 16 + (lane >> 4) * 4;
                    v[u] += *(const f32x4*)xp;
                    v[u] += *(const f32x4*)(xp + (size_t)16 * K);
                }
            } else v[u] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) acc += v[u];
    }
    if (acc[0] + acc[1] + acc[2] + acc[3] == 123.456f) sink[0] = acc[0];
}

int main() {
    const int rows = 4096, K = 2560;
    const size_t n = (size_t)rows * K;
    const int NB = 14;                           // 14 x 42 MB = 587 MB
    std::vector<float*> bufs(NB);
    for (auto& b : bufs) { hipMalloc(&b, n * 4); hipMemset(b, 0, n * 4); }
    float* sink; hipMalloc(&sink, 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int packed = 0; packed < 3; ++packed)
        for (int cold = 0; cold < 2; ++cold) {
            float best = 1e9, tot = 0;
            for (int it = 0; it < 40; ++it) {
                const float* W = cold ? bufs[it % NB] : bufs[0];
                hipEventRecord(e0);
                if (packed == 1) hipLaunchKernelGGL((pull<1, 0>), dim3(rows / 16), dim3(512), 0, 0, W, rows, K, sink, bufs[1]);
                else if (packed == 0) hipLaunchKernelGGL((pull<0, 0>), dim3(rows / 16), dim3(512), 0, 0, W, rows, K, sink, bufs[1]);
                else hipLaunchKernelGGL((pull<0, 1>), dim3(rows / 16), dim3(512), 0, 0, W, rows, K, sink, bufs[1]);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                if (it >= 10) { tot += ms; if (ms < best) best = ms; }
            }
            printf("%s %s: avg %.1f us  best %.1f us  -> %.2f TB/s (avg)\n", packed == 2 ? "row-major + X (32 items)" : packed ? "packed  " : "row-major", cold ? "cold" : "hot ",
                   tot / 30 * 1e3, best * 1e3, n * 4 / (tot / 30 * 1e-3) / 1e12);
        }
    return 0;
}
