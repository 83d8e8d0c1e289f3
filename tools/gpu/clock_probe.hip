// Effective shader clock under load, measured ON the GPU: one extra wave reads the shader-clock counter (s_memtime) and the
// constant 100 MHz counter (s_memrealtime) 2 ms apart while the chip runs (a) nothing, (b) the register-only fp32 MFMA loop on every
// CU, (c) a streaming copy on every CU, (d) both.  Question (round 5): the K loop of the convolution kernels reaches 142 TFLOP/s
// with its LDS-DMA loads removed and 130 with them issued (not even awaited) — is that the clock, or the CU?
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/gpu/clock_probe.hip -o tools/gpu/clock_probe && tools/gpu/clock_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#pragma clang diagnostic ignored "-Wunused-value"
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ void __launch_bounds__(256) mfma_loop(float* out, int iters) {
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    float a = (float)(threadIdx.x & 7) * 0.125f, b = 1.0f + (float)(threadIdx.x & 3);
    for (int i = 0; i < iters; ++i)
#pragma unroll
        for (int u = 0; u < 16; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    float s = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) s += acc[r];
    if (s == 12345.678f) out[blockIdx.x * 256 + threadIdx.x] = s;
}
__global__ void __launch_bounds__(256) stream_copy(const float4* src, float4* dst, size_t n, int reps) {
    for (int r = 0; r < reps; ++r)
        for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) dst[i] = src[i];
}
__global__ void spin_clock(unsigned long long* out, unsigned long long ref_ticks) {
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    const unsigned long long c0 = __builtin_amdgcn_s_memtime();
    unsigned long long r1 = r0;
    while (r1 - r0 < ref_ticks) {
        __builtin_amdgcn_s_sleep(32);
        r1 = __builtin_amdgcn_s_memrealtime();
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime();
    out[0] = r1 - r0;
    out[1] = c1 - c0;
}

int main() {
    hipStream_t sa, sb, sc;
    hipStreamCreate(&sa);
    hipStreamCreate(&sb);
    hipStreamCreate(&sc);
    float* buf;
    hipMalloc(&buf, 256 * 4 * 256 * sizeof(float));
    const size_t n4 = (size_t)1 << 26;   // 1 GiB per buffer
    float4 *src, *dst;
    hipMalloc(&src, n4 * 16);
    hipMalloc(&dst, n4 * 16);
    hipMemset(src, 1, n4 * 16);
    unsigned long long* out;
    hipHostMalloc(&out, 16);
    const char* names[4] = {"idle", "fp32 MFMA loop on every CU", "streaming copy on every CU", "MFMA loop + streaming copy"};
    for (int sc_i = 0; sc_i < 4; ++sc_i) {
        hipEvent_t e0, e1;
        hipEventCreate(&e0);
        hipEventCreate(&e1);
        const int iters = 60000;   // ~25 ms of MFMA per wave
        hipEventRecord(e0, sa);
        if (sc_i == 1 || sc_i == 3) hipLaunchKernelGGL(mfma_loop, dim3(256 * 2), dim3(256), 0, sa, buf, iters);
        hipEventRecord(e1, sa);
        if (sc_i == 2 || sc_i == 3) hipLaunchKernelGGL(stream_copy, dim3(256 * 2), dim3(256), 0, sb, src, dst, n4, 40);
        hipLaunchKernelGGL(spin_clock, dim3(1), dim3(64), 0, sc, out, 200000ull);   // 2 ms at 100 MHz
        hipDeviceSynchronize();
        float ms = 0.f;
        hipEventElapsedTime(&ms, e0, e1);
        const double mhz = (double)out[1] / ((double)out[0] / 100.0);
        printf("%-32s shader clock %.0f MHz", names[sc_i], mhz);
        if (sc_i == 1 || sc_i == 3) printf("   MFMA loop %.1f TFLOP/s", 256.0 * 2 * 4 * iters * 16.0 * 4096.0 / ms / 1e9);
        printf("\n");
    }
    return 0;
}
