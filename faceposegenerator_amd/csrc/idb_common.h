// Shared device/host helpers for the gfx950 kernels.  wave = 64 lanes everywhere.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include <string.h>
#include "../../include/idb_kernels.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) _Float16 f16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;

#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
#define GLB_PTR(p) ((const __attribute__((address_space(1))) void*)(p))

// Operand-type traits: bf16 and f16 share storage size, MFMA shapes, rate and fragment layouts.
template <typename T> struct Op;
template <> struct Op<__bf16> {
    using v8 = bf16x8;
    using v4 = bf16x4;
    static __device__ __forceinline__ f32x4 mfma16(v8 a, v8 b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
    }
    static __device__ __forceinline__ f32x16 mfma32(v8 a, v8 b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
    }
    static __device__ __forceinline__ v4 ds_read_tr(const void* lds) {
        return __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) v4*)lds);
    }
};
template <> struct Op<_Float16> {
    using v8 = f16x8;
    using v4 = f16x4;
    static __device__ __forceinline__ f32x4 mfma16(v8 a, v8 b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
    }
    static __device__ __forceinline__ f32x16 mfma32(v8 a, v8 b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    }
    static __device__ __forceinline__ v4 ds_read_tr(const void* lds) {
        typedef __fp16 fp16x4_t __attribute__((__vector_size__(4 * sizeof(__fp16))));
        const fp16x4_t t = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fp16x4_t*)lds);
        return __builtin_bit_cast(v4, t);
    }
};

template <typename T> __device__ __forceinline__ float to_f32(T x) { return (float)x; }
template <typename T> __device__ __forceinline__ T from_f32(float x) { return (T)x; }

// x * sigmoid(x) with v_rcp_f32 (1 ulp) instead of the IEEE division sequence (v_div_scale / v_rcp / 4 FMAs / v_div_fmas / v_div_fixup,
// ~10 VALU instructions per element): the GroupNorm+SiLU pass at batch 64 is VALU-co-bound, its result is rounded to 16 bits anyway
__device__ __forceinline__ float silu_f(float x) { return x * __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
// exact-GELU 0.5 x (1 + erf(x / sqrt 2)) with erf from Abramowitz & Stegun 7.1.26 (|abs err| <= 1.5e-7, below fp32
// output resolution of the GEGLU product): 1 exp + 1 rcp + 6 FMAs instead of libm erff's ~40 VALU ops, which made the
// GEGLU epilogue the critical path of the K = C projection GEMMs.
__device__ __forceinline__ float gelu_erf_f(float x) {
    const float z = fabsf(x) * 0.70710678118654752f;
    const float t = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * z);
    float poly = 1.061405429f;
    poly = poly * t - 1.453152027f;
    poly = poly * t + 1.421413741f;
    poly = poly * t - 0.284496736f;
    poly = poly * t + 0.254829592f;
    const float erfc_z = poly * t * __expf(-z * z);          // 1 - erf(z), z >= 0
    const float cdf = x >= 0.f ? 1.0f - 0.5f * erfc_z : 0.5f * erfc_z;
    return x * cdf;
}

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

// ---- host side -----------------------------------------------------------------------------
void idb_set_error(const char* fmt, ...);
const void* idb_zero_page(void);   // >= 256 zero bytes in device memory (padding source for LDS-DMA)

#define IDB_REQUIRE(cond, ...)                  \
    do {                                        \
        if (!(cond)) {                          \
            idb_set_error(__VA_ARGS__);         \
            return IDB_EINVAL;                  \
        }                                       \
    } while (0)

extern unsigned long long idb_launch_counter;   // idb_misc.hip; read through idb_launch_count()

#define IDB_CHECK_LAUNCH(name)                                                     \
    do {                                                                           \
        __atomic_fetch_add(&idb_launch_counter, 1ull, __ATOMIC_RELAXED);           \
        hipError_t e_ = hipGetLastError();                                         \
        if (e_ != hipSuccess) {                                                    \
            idb_set_error("%s: launch failed: %s", name, hipGetErrorString(e_));   \
            return IDB_EHIP;                                                       \
        }                                                                          \
    } while (0)

static inline bool idb_aligned16(const void* p) { return (((uintptr_t)p) & 15) == 0; }
static inline bool idb_is_operand_dtype(int dt) { return dt == IDB_BF16 || dt == IDB_F16; }

// idb_norm.hip: first GroupNorm pass over a dense [batch][hw][c] tensor with one partial per 64-row block (the layout of
// idb_gemm_desc.gn_partials); used by idb_gemm when its own launches cannot produce the statistics.
int idb_launch_gn_stats64(const void* x, int c, int batch, int hw, int groups, float* partial, int dtype, hipStream_t st);
