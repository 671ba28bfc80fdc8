// Shared device helpers for the gfx950 kernels.
//
// Data layout used everywhere on the WaveGlow path ("planes"):
//   an activation tensor with C channels and L time steps per batch element is
//   stored channel-last in 32-channel chunks as TWO bf16 planes (hi, lo) with
//       x  ~=  float(hi) + float(lo)          (split-bf16, ~16 mantissa bits)
//   plane[b][c / 32][row][c % 32],   row = halo + t,   0 <= row < Lp
//   Rows outside [halo, halo + L) are zero and are never written: they are the
//   zero padding of the dilated convolutions (reference glow.py:134-137).
//   One 32-channel row is 64 B, so a 256-row K-step tile is one contiguous
//   16 KiB block that is fetched straight into LDS by global_load_lds.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef unsigned short u16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned short u16x4;

#define T2S_CHUNK 32            // channels per plane chunk (= one BK step of the GEMM)
#define T2S_TILE_M 256          // output-channel rows per workgroup
#define T2S_TILE_N 256          // time steps per workgroup

#ifndef T2S_SPLIT_F16
// ---- the shipped operand format: split-bf16 ----
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;      // one MFMA operand fragment (8 x 16-bit)
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
#define T2S_MFMA32 __builtin_amdgcn_mfma_f32_16x16x32_bf16
static __device__ __forceinline__ u16 bf16_bits(float x) {
    __bf16 h = (__bf16)x;                      // v_cvt_pk_bf16_f32: RNE, NaN-preserving
    return __builtin_bit_cast(u16, h);
}
static __device__ __forceinline__ float bf16_to_f32(u16 b) {
    return __builtin_bit_cast(float, (uint32_t)b << 16);
}
#else
// ---- DIAGNOSTIC build (-DT2S_SPLIT_F16, `python -m text2speech_amd.build --variant f16x3 -DT2S_SPLIT_F16`): the same three
// products per MAC with fp16 operand planes instead of bf16 (v_mfma_f32_16x16x32_f16: same issue rate, profiles/r03_mfma_f8_probe.txt).
// hi = fp16(x) keeps 11 significand bits instead of 8, so hi + lo carries ~22 bits where both are normal numbers - but fp16 has
// no exponent range to spare: |x| > 65504 overflows to inf and |x| < 6e-5 loses the low plane to subnormals.  The no-grad WaveGlow
// forward / infer only (activations and weight-normed weights sit inside that range; GRADIENT planes do not, and the training path
// also builds bf16 constants on the Python side): tests/test_waveglow_gpu.py::test_stress_weights_* runs it next to the shipped
// library.  The type and function NAMES below stay those of the shipped format so that no kernel source differs between the builds.
typedef __attribute__((ext_vector_type(8))) _Float16 bf16x8;
typedef __attribute__((ext_vector_type(4))) _Float16 bf16x4;
#define T2S_MFMA32 __builtin_amdgcn_mfma_f32_16x16x32_f16
static __device__ __forceinline__ u16 bf16_bits(float x) {
    _Float16 h = (_Float16)x;                  // v_cvt_f16_f32: RNE
    return __builtin_bit_cast(u16, h);
}
static __device__ __forceinline__ float bf16_to_f32(u16 b) {
    return (float)__builtin_bit_cast(_Float16, b);
}
#endif
// x -> (hi, lo) with hi = bf16(x), lo = bf16(x - hi); x - hi is exact in f32.
static __device__ __forceinline__ void split_bf16(float x, u16& hi, u16& lo) {
    hi = bf16_bits(x);
    lo = bf16_bits(x - bf16_to_f32(hi));
}
static __device__ __forceinline__ float join_bf16(u16 hi, u16 lo) {
    return bf16_to_f32(hi) + bf16_to_f32(lo);
}

static __device__ __forceinline__ float fast_sigmoid(float x) {
    return 1.0f / (1.0f + __expf(-x));
}
static __device__ __forceinline__ float fast_tanh(float x) {
    // tanh(x) = 1 - 2 / (exp(2x) + 1); saturates cleanly at +-1 for large |x|
    return 1.0f - 2.0f / (__expf(2.0f * x) + 1.0f);
}

// wave-wide sum (64 lanes)
static __device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// Kernels that use more than 64 KB of dynamic LDS need hipFuncAttributeMaxDynamicSharedMemorySize raised once per device.
// `mask` is one static word per kernel: bit d = done on device d.  Setting the attribute twice is harmless, so a relaxed
// atomic is all the synchronisation two calling threads need.
#include <atomic>
static inline hipError_t t2s_raise_lds_limit(const void* fn, int bytes, std::atomic<unsigned long long>& mask) {
    int device = 0;
    hipError_t e = hipGetDevice(&device);
    if (e != hipSuccess) return e;
    const unsigned long long bit = 1ull << (device & 63);
    if (mask.load(std::memory_order_relaxed) & bit) return hipSuccess;
    e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e == hipSuccess) mask.fetch_or(bit, std::memory_order_relaxed);
    return e;
}
