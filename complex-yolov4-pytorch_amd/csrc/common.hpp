// Shared device/host helpers for libcyolo_hip.so (gfx950 only; wave = 64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "cyolo_hip.h"

typedef _Float16 f16;
typedef f16 f16x8 __attribute__((ext_vector_type(8)));
typedef f16 f16x4 __attribute__((ext_vector_type(4)));
typedef f16 f16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16;
typedef bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

#define CY_WAVE 64
// rows of the shared-bin statistics tables (conv epilogue -> BatchNorm, BatchNorm backward sums): bin = block % CY_STAT_BINS.
// 16 keeps the fold cheap enough to run in every consumer block's prologue (cy_bn_act_fwd_fused) at <= tiles / 16 adds per address.
#define CY_STAT_BINS 16

// hipGetLastError() is per-thread sticky state shared with every other HIP user in the process (PyTorch included):
// clear it on entry so that a launch check only ever reports this call's own launches.
#define CY_ENTER() (void)hipGetLastError()

#define CY_LAUNCH_CHECK()                                   \
    do {                                                    \
        hipError_t e__ = hipGetLastError();                 \
        if (e__ != hipSuccess) return -(1000 + (int)e__);   \
    } while (0)

static inline hipStream_t cy_s(cy_stream_t s) { return (hipStream_t)s; }

// One-time per-kernel set-up (hipFuncSetAttribute of the dynamic LDS size) is per DEVICE: a process that drives several GPUs
// (not the one-process-per-GPU layout bench.py uses, but the C ABI does not forbid it) must repeat it on each.  `mask` is the
// caller's function-static word, bit d = done on device d.  Several host threads may first-launch the same instantiation at
// once (the recorder is thread-local so that a prefetch thread and an eval engine can call the library concurrently): the mask
// is updated with an atomic fetch-or (two threads that both see the bit clear both set the attribute, which is
// idempotent).  hipGetDevice is asked every time: a thread may have switched devices since its last launch.
static inline bool cy_first_use_on_device(unsigned long long& mask) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return true;
    const unsigned long long bit = 1ull << (dev & 63);
    if (__atomic_load_n(&mask, __ATOMIC_RELAXED) & bit) return false;
    return (__atomic_fetch_or(&mask, bit, __ATOMIC_RELAXED) & bit) == 0;
}

// ---- element traits: 16-byte chunk = CH elements ------------------------------------------------
template <typename T>
struct Elem;
template <>
struct Elem<f16> {
    static constexpr int CH = 8;
    static constexpr int DT = CY_F16;
};
template <>
struct Elem<bf16> {
    static constexpr int CH = 8;
    static constexpr int DT = CY_BF16;
};
template <>
struct Elem<float> {
    static constexpr int CH = 4;
    static constexpr int DT = CY_F32;
};

// load/store a 16-byte chunk as CH floats
template <typename T>
__device__ __forceinline__ void chunk_to_f32(const u32x4& v, float* f);
template <>
__device__ __forceinline__ void chunk_to_f32<f16>(const u32x4& v, float* f) {
    const f16x8 h = __builtin_bit_cast(f16x8, v);
#pragma unroll
    for (int i = 0; i < 8; ++i) f[i] = (float)h[i];
}
template <>
__device__ __forceinline__ void chunk_to_f32<bf16>(const u32x4& v, float* f) {
    // bf16 -> f32 is a 16-bit shift: exact, two VALU per pair
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        f[2 * i] = __builtin_bit_cast(float, v[i] << 16);
        f[2 * i + 1] = __builtin_bit_cast(float, v[i] & 0xFFFF0000u);
    }
}
template <>
__device__ __forceinline__ void chunk_to_f32<float>(const u32x4& v, float* f) {
    const f32x4 h = __builtin_bit_cast(f32x4, v);
#pragma unroll
    for (int i = 0; i < 4; ++i) f[i] = h[i];
}
template <typename T>
__device__ __forceinline__ u32x4 f32_to_chunk(const float* f);
template <>
__device__ __forceinline__ u32x4 f32_to_chunk<f16>(const float* f) {
    f16x8 h;
#pragma unroll
    for (int i = 0; i < 8; ++i) h[i] = (f16)f[i];
    return __builtin_bit_cast(u32x4, h);
}
template <>
__device__ __forceinline__ u32x4 f32_to_chunk<bf16>(const float* f) {
    bf16x8 h;
#pragma unroll
    for (int i = 0; i < 8; ++i) h[i] = (bf16)f[i];     // round to nearest even (v_cvt_pk_bf16_f32 on gfx950)
    return __builtin_bit_cast(u32x4, h);
}
template <>
__device__ __forceinline__ u32x4 f32_to_chunk<float>(const float* f) {
    f32x4 h;
#pragma unroll
    for (int i = 0; i < 4; ++i) h[i] = f[i];
    return __builtin_bit_cast(u32x4, h);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// ---- activations -------------------------------------------------------------------------------
// Mish(x) = x*tanh(softplus(x)), softplus threshold 20 as torch (reference darknet2pytorch.py:22-28).
// With n = e^x: tanh(log(1+n)) = n(n+2) / (n(n+2)+2).
// FAST (fp16 storage mode): v_exp_f32 / v_rcp_f32 forms, ~1e-6 relative -- far below the fp16 rounding of the stored
// result, and what keeps these HBM-bound passes from becoming VALU-bound (libm expf + two IEEE divisions cost ~50
// VALU instructions per element).  The f32 parity mode keeps the accurate forms.
template <bool FAST>
__device__ __forceinline__ float cy_exp(float x) { return FAST ? __expf(x) : expf(x); }
template <bool FAST>
__device__ __forceinline__ float cy_div(float a, float b) { return FAST ? a * __builtin_amdgcn_rcpf(b) : a / b; }

// Both functions are written branch-free (argument clamped at the threshold, result selected): an early return compiles to one
// exec-mask region per ELEMENT (s_and_saveexec / s_cbranch_execz around each exp -> rcp chain), which serialises the eight
// elements of a 16-byte chunk and keeps the compiler from pairing them into packed f32 instructions -- the BN-backward reduce
// pass and the dgrad sums epilogue are VALU-bound on exactly this code.  Values are unchanged for every input: below the
// threshold the clamp is the identity, above it the select returns what the early return did, NaN stays NaN (x > 20 is false
// for it, so the clamp passes it through).
template <bool FAST>
__device__ __forceinline__ float mish_f(float x) {
    const bool big = x > 20.f;
    const float xc = big ? 20.f : x;
    const float n = cy_exp<FAST>(xc);
    const float w = n * (n + 2.f);
    const float r = xc * cy_div<FAST>(w, w + 2.f);
    return big ? x : r;
}
template <bool FAST>
__device__ __forceinline__ float mish_grad(float x0) {
    const bool big = x0 > 20.f;
    const float x = big ? 20.f : x0;
    const float n = cy_exp<FAST>(x);
    const float w = n * (n + 2.f);
    if (FAST) {
        // one reciprocal instead of two divisions: with v = w + 2,  tanh(softplus) = w / v,  1 - t^2 = 4 (n + 1)^2 / v^2,
        // sigmoid = n / (n + 1)  =>  mish' = (w v + 4 x n (n + 1)) / v^2   (v^2 <= 5e34 for x <= 20: no overflow).
        // The backward reduce pass is VALU-bound on this function.
        const float v = w + 2.f;
        const float r = (w * v + 4.f * x * n * (n + 1.f)) * __builtin_amdgcn_rcpf(v * v);
        return big ? 1.f : r;
    }
    const float t = cy_div<FAST>(w, w + 2.f);    // tanh(softplus(x))
    const float sg = cy_div<FAST>(n, 1.f + n);   // sigmoid(x)
    const float r = t + x * (1.f - t * t) * sg;
    return big ? 1.f : r;
}
template <int ACT, bool FAST = false>
__device__ __forceinline__ float act_f(float z) {
    if (ACT == CY_ACT_MISH) return mish_f<FAST>(z);
    if (ACT == CY_ACT_LEAKY) return z > 0.f ? z : 0.1f * z;
    return z;
}
template <int ACT, bool FAST = false>
__device__ __forceinline__ float act_grad(float z) {
    if (ACT == CY_ACT_MISH) return mish_grad<FAST>(z);
    if (ACT == CY_ACT_LEAKY) return z > 0.f ? 1.f : 0.1f;
    return 1.f;
}
