// Shared host/device helpers for the gfx950 kernels of libunipose_hip.so.
#pragma once

#ifdef UP_EMU
// tests/emu/hip_emu.h (force-included by the CPU emulation build used ONLY by tests) provides the
// HIP surface: threadIdx/blockIdx, __shared__, __syncthreads, hipLaunchKernelGGL, MFMA, shuffles.
#else
#include <hip/hip_runtime.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
#endif

#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/unipose_hip.h"

namespace up {

void set_error(const char* fmt, ...);
int check_launch(const char* what);
void set_bn_rows(int on);   // norm_act.hip: row-strided BatchNorm passes on / off (up_conv_tune "bn_rows")

#define UP_REQUIRE(cond, code, ...)            \
    do {                                       \
        if (!(cond)) {                         \
            ::up::set_error(__VA_ARGS__);      \
            return (code);                     \
        }                                      \
    } while (0)

static inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

// n / d for 0 <= n < 2^31 with one mul-hi and one shift.
struct FastDiv {
    uint32_t mul, shr, d;
};
static inline FastDiv make_fastdiv(int d) {
    FastDiv f;
    f.d = (uint32_t)d;
    if (d <= 1) {
        f.mul = 0;
        f.shr = 0;
        return f;
    }
    uint32_t lg = 0;
    while ((1u << lg) < (uint32_t)d) ++lg;
    uint32_t p = 31 + lg;
    f.mul = (uint32_t)(((1ull << p) + (uint32_t)d - 1) / (uint32_t)d);
    f.shr = p - 32;
    return f;
}
__device__ __forceinline__ uint32_t fdiv(uint32_t n, const FastDiv& f) {
    const uint32_t q = __umulhi(n, f.mul) >> f.shr;   // computed unconditionally: a select, not a branch that would
    return f.d <= 1 ? n : q;                          // split the caller's scheduling region
}

static inline int cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

// ---- bf16 storage (BASELINE configs[4]): activations as raw bf16 bit patterns in HBM, arithmetic in fp32 -------------
typedef uint16_t bf16_t;          // (element-type codes UP_DT_F32 / UP_DT_BF16: include/unipose_hip.h)
#ifndef UP_EMU
// round-to-nearest-even pair conversion (v_cvt_pk_bf16_f32); element 0 in the low half-word
__device__ __forceinline__ uint32_t pack_bf16x2(float lo_elem, float hi_elem) {
    typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
    bf16x2_t v;
    v[0] = (__bf16)lo_elem;
    v[1] = (__bf16)hi_elem;
    return __builtin_bit_cast(uint32_t, v);
}
#endif
__device__ __forceinline__ float bf_lo(uint32_t u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float bf_hi(uint32_t u) { return __uint_as_float(u & 0xffff0000u); }
// four consecutive elements <-> float4 (fp32: one 16-byte access; bf16: one 8-byte access)
template <typename T>
__device__ __forceinline__ float4 ld4(const T* p);
template <>
__device__ __forceinline__ float4 ld4<float>(const float* p) {
    return *reinterpret_cast<const float4*>(p);
}
template <>
__device__ __forceinline__ float4 ld4<bf16_t>(const bf16_t* p) {
    const uint2 u = *reinterpret_cast<const uint2*>(p);
    return make_float4(bf_lo(u.x), bf_hi(u.x), bf_lo(u.y), bf_hi(u.y));
}
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
__device__ __forceinline__ void st4(bf16_t* p, float4 v) {
    uint2 u;
    u.x = pack_bf16x2(v.x, v.y);
    u.y = pack_bf16x2(v.z, v.w);
    *reinterpret_cast<uint2*>(p) = u;
}
__device__ __forceinline__ float ld1(const float* p) { return *p; }
__device__ __forceinline__ float ld1(const bf16_t* p) { return __uint_as_float((uint32_t)*p << 16); }
__device__ __forceinline__ void st1(float* p, float v) { *p = v; }
__device__ __forceinline__ void st1(bf16_t* p, float v) { *p = (bf16_t)(pack_bf16x2(v, 0.f) & 0xffffu); }

}  // namespace up
