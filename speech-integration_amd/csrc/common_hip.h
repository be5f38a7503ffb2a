// Shared device/host helpers for libssi_hip.so (gfx950 only; wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/ssi_hip.h"

typedef __bf16 bf16_t;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

#define SSI_WAVE 64

// thread-local error text (defined in api.hip)
void ssi_set_error(const char* fmt, ...);

#define SSI_CHECK_ARG(cond)                                                                  \
    do {                                                                                     \
        if (!(cond)) {                                                                       \
            ssi_set_error("%s:%d: argument check failed: %s", __FILE__, __LINE__, #cond);    \
            return SSI_ERR_ARG;                                                              \
        }                                                                                    \
    } while (0)

#define SSI_LAUNCH_CHECK()                                                                   \
    do {                                                                                     \
        hipError_t e_ = hipGetLastError();                                                   \
        if (e_ != hipSuccess) {                                                              \
            ssi_set_error("%s:%d: launch failed: %s", __FILE__, __LINE__, hipGetErrorString(e_)); \
            return SSI_ERR_HIP + (int)e_;                                                    \
        }                                                                                    \
    } while (0)

static inline int64_t ssi_cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }
static inline int64_t ssi_align_up(int64_t a, int64_t b) { return ssi_cdiv(a, b) * b; }

// ---- storage <-> fp32 ------------------------------------------------------------------------------------------------
template <typename T> __device__ __forceinline__ float to_f32(T v);
template <> __device__ __forceinline__ float to_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ float to_f32<bf16_t>(bf16_t v) { return (float)v; }
template <typename T> __device__ __forceinline__ T from_f32(float v);
template <> __device__ __forceinline__ float from_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ bf16_t from_f32<bf16_t>(float v) { return (bf16_t)v; }  // RNE, NaN-preserving (v_cvt_pk_bf16_f32)

// logistic function used by every SwiGLU site (elementwise kernels and the fused GEMM epilogues must agree bit for bit).
// bf16 storage: v_exp_f32 / v_rcp_f32 (about 1 ulp each, far below the bf16 rounding that follows); f32 storage: libm.
template <typename T> __device__ __forceinline__ float ssi_sigmoid(float x);
template <> __device__ __forceinline__ float ssi_sigmoid<float>(float x) { return 1.f / (1.f + expf(-x)); }
template <> __device__ __forceinline__ float ssi_sigmoid<bf16_t>(float x) {
    return __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(x * -1.44269504088896f));
}
template <typename T> __device__ __forceinline__ float ssi_silu(float x);
template <> __device__ __forceinline__ float ssi_silu<float>(float x) { return x / (1.f + expf(-x)); }
template <> __device__ __forceinline__ float ssi_silu<bf16_t>(float x) { return x * ssi_sigmoid<bf16_t>(x); }
// SwiGLU backward per element, shared by ssi_swiglu_bwd and the fused GEMM epilogues: contraction is switched off here so
// that every call site rounds identically whatever surrounds it (the fused and unfused paths are tested bit for bit).
template <typename T> __device__ __forceinline__ void ssi_swiglu_bwd_elem(float gf, float uf, float df, float& dgate, float& dup) {
#pragma clang fp contract(off)
    const float sig = ssi_sigmoid<T>(gf);
    const float one_minus = 1.f - sig;
    const float t = 1.f + gf * one_minus;
    dup = df * (gf * sig);
    dgate = (df * uf) * (sig * t);
}

// One RoPE pair, shared by ssi_rope_inplace and the QKV GEMM epilogue (same reason for switching contraction off; it is also
// what the reference's unfused torch expression computes)
__device__ __forceinline__ void ssi_rope_pair(float x0, float x1, float c, float s, float& o0, float& o1) {
#pragma clang fp contract(off)
    o0 = x0 * c - x1 * s;
    o1 = x1 * c + x0 * s;
}

// 16-byte vector of storage elements: 4 floats or 8 bf16
template <typename T> struct Vec16;
template <> struct Vec16<float> {
    static constexpr int N = 4;
    f32x4 v;
    __device__ __forceinline__ float get(int i) const { return v[i]; }
    __device__ __forceinline__ void set(int i, float x) { v[i] = x; }
};
template <> struct Vec16<bf16_t> {
    static constexpr int N = 8;
    bf16x8 v;
    __device__ __forceinline__ float get(int i) const { return (float)v[i]; }
    __device__ __forceinline__ void set(int i, float x) { v[i] = (bf16_t)x; }
};
template <typename T> __device__ __forceinline__ Vec16<T> load16(const T* p) {
    Vec16<T> r;
    r.v = *reinterpret_cast<const decltype(r.v)*>(p);
    return r;
}
template <typename T> __device__ __forceinline__ void store16(T* p, const Vec16<T>& r) {
    *reinterpret_cast<decltype(r.v)*>(p) = r.v;
}

// streaming forms (touched once per launch: keep them out of the way of what the caches could hold)
template <typename T> __device__ __forceinline__ Vec16<T> load16_nt(const T* p) {
    Vec16<T> r;
    r.v = __builtin_nontemporal_load(reinterpret_cast<const decltype(r.v)*>(p));
    return r;
}
template <typename T> __device__ __forceinline__ void store16_nt(T* p, const Vec16<T>& r) {
    __builtin_nontemporal_store(r.v, reinterpret_cast<decltype(r.v)*>(p));
}

// ---- reductions ------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
// block-wide sum for blockDim.x <= 1024; `red` is >= 16 floats of LDS; result broadcast to all threads
__device__ __forceinline__ float block_sum(float v, float* red) {
    v = wave_sum(v);
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63, nw = (blockDim.x + 63) >> 6;
    __syncthreads();
    if (l == 0) red[w] = v;
    __syncthreads();
    float t = (l < nw) ? red[l] : 0.f;
    t = wave_sum(t);
    return t;
}
__device__ __forceinline__ float block_max(float v, float* red) {
    v = wave_max(v);
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63, nw = (blockDim.x + 63) >> 6;
    __syncthreads();
    if (l == 0) red[w] = v;
    __syncthreads();
    float t = (l < nw) ? red[l] : -INFINITY;
    t = wave_max(t);
    return t;
}

#define SSI_DISPATCH_DTYPE(dtype, ...)                           \
    do {                                                         \
        if ((dtype) == SSI_F32) {                                \
            using T = float;                                     \
            __VA_ARGS__;                                         \
        } else if ((dtype) == SSI_BF16) {                        \
            using T = bf16_t;                                    \
            __VA_ARGS__;                                         \
        } else {                                                 \
            ssi_set_error("unsupported dtype %d", (int)(dtype)); \
            return SSI_ERR_ARG;                                  \
        }                                                        \
    } while (0)
