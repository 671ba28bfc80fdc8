// Microbenchmark: how long does the FIRST dependent load of a small kernel take when the previous kernel on the stream
// was (a) a 64 MB weight-streaming kernel, (b) a tiny kernel?  Classes of address probed, one load -> wait -> stamp each:
//   S_written : a 4 KB buffer of its own (hipMalloc) that the previous kernel just wrote (like h_dec)
//   S_ro      : a 4 KB read-only buffer of its own (like a bias vector)
//   W_tail    : the last 4 KB of the streamed 64 MB allocation (translation certainly warm)
//   W_other   : 4 KB inside a second 64 MB allocation that nobody streams (like W_proj one step ago)
// Build + run on the GPU box:  hipcc --offload-arch=gfx950 -O3 tools/microbench/after_stream_latency.hip -o /tmp/asl && /tmp/asl
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__global__ __launch_bounds__(1024) void stream_kernel(const float4* __restrict__ W, size_t n4, float* out) {
    float acc = 0.f;
    for (size_t i = (size_t)blockIdx.x * 1024 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 1024) {
        const float4 v = W[i];
        acc += v.x + v.y + v.z + v.w;
    }
    // every workgroup writes 4 floats of the small output (like an LSTM cell's hidden units)
    if (threadIdx.x < 4) out[blockIdx.x * 4 + threadIdx.x] = acc;
}
__global__ void tiny_kernel(float* out) {
    if (threadIdx.x < 4) out[blockIdx.x * 4 + threadIdx.x] = 1.0f;
}
__global__ __launch_bounds__(256) void probe_kernel(const float* s_written, const float* s_ro, const float* w_tail,
                                                    const float* w_other, unsigned long long* stamps, float* sink) {
    if (blockIdx.x != 0 || threadIdx.x >= 64) return;
    const int lane = threadIdx.x;
    unsigned long long t[5];
    float acc = 0.f;
    t[0] = __builtin_amdgcn_s_memrealtime();
    acc += s_written[lane * 4];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    t[1] = __builtin_amdgcn_s_memrealtime();
    acc += s_ro[lane * 4];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    t[2] = __builtin_amdgcn_s_memrealtime();
    acc += w_tail[lane * 4];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    t[3] = __builtin_amdgcn_s_memrealtime();
    acc += w_other[lane * 4];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    t[4] = __builtin_amdgcn_s_memrealtime();
    if (lane == 0) for (int i = 0; i < 5; ++i) stamps[i] = t[i];
    sink[lane] = acc;
}

static double med(std::vector<double> v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; }

int main() {
    const size_t wbytes = 64ull << 20;
    float *W, *W2, *Sw, *Sro, *sink;
    unsigned long long* stamps;
    CK(hipMalloc(&W, wbytes)); CK(hipMalloc(&W2, wbytes)); CK(hipMalloc(&Sw, 4096)); CK(hipMalloc(&Sro, 4096));
    CK(hipMalloc(&sink, 4096)); CK(hipMalloc(&stamps, 64));
    CK(hipMemset(W, 0, wbytes)); CK(hipMemset(W2, 0, wbytes)); CK(hipMemset(Sro, 0, 4096));
    const float* w_tail = W + wbytes / 4 - 1024;
    const float* w_other = W2 + (wbytes / 8);
    for (int mode = 0; mode < 2; ++mode) {
        std::vector<double> d[4];
        for (int it = 0; it < 300; ++it) {
            if (mode == 0) hipLaunchKernelGGL(stream_kernel, dim3(256), dim3(1024), 0, 0, (const float4*)W, wbytes / 16, Sw);
            else hipLaunchKernelGGL(tiny_kernel, dim3(256), dim3(64), 0, 0, Sw);
            hipLaunchKernelGGL(probe_kernel, dim3(85), dim3(256), 0, 0, Sw, Sro, w_tail, w_other, stamps, sink);
            unsigned long long h[5];
            CK(hipMemcpy(h, stamps, sizeof(h), hipMemcpyDeviceToHost));
            if (it >= 20) for (int i = 0; i < 4; ++i) d[i].push_back((double)(h[i + 1] - h[i]) * 10.0);
        }
        printf("{\"previous_kernel\": \"%s\", \"first_load_ns_median\": {\"small_buffer_just_written\": %.0f, \"small_read_only_buffer\": %.0f, "
               "\"tail_of_streamed_allocation\": %.0f, \"inside_other_64MB_allocation\": %.0f}}\n",
               mode == 0 ? "64 MB stream (256 x 1024 threads)" : "tiny (256 x 64 threads)", med(d[0]), med(d[1]), med(d[2]), med(d[3]));
    }
    return 0;
}
