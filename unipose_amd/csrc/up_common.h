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

}  // namespace up
