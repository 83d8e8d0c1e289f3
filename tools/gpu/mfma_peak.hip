// Sustained rate of v_mfma_f32_32x32x2_f32 on this chip with NOTHING else in the way (no LDS, no memory): the practical ceiling the
// nominal 157.3 TFLOP/s (256 CUs x 4 SIMDs x 64 FLOP/clk x 2.4 GHz, MI355X_MICROARCH.md) has to be read against.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/gpu/mfma_peak.hip -o tools/gpu/mfma_peak && tools/gpu/mfma_peak
// Variants: waves per SIMD (1 / 2 / 4) x independent accumulators per wave (1 = every MFMA depends on the previous one, like the
// 64x64 tile's wave; 4 = the 128x128 tile's wave).  Each launch runs ~2 ms; the best of five is reported.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#pragma clang diagnostic ignored "-Wunused-value"

typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ void __launch_bounds__(256) mfma_loop(float* out, int iters) {
    f32x16 acc[NACC];
#pragma unroll
    for (int j = 0; j < NACC; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    float a = (float)(threadIdx.x & 7) * 0.125f, b = 1.0f + (float)(threadIdx.x & 3);
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 16 / NACC; ++u)
#pragma unroll
            for (int j = 0; j < NACC; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[j], 0, 0, 0);
    }
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < NACC; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[j][r];
    if (s == 12345.678f) out[blockIdx.x * 256 + threadIdx.x] = s;   // keeps the loop alive
}

template <int NACC>
static double run(int cus, int blocks_per_cu, int iters, float* buf) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    double best = 1e30;
    for (int rep = 0; rep < 6; ++rep) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(mfma_loop<NACC>, dim3(cus * blocks_per_cu), dim3(256), 0, 0, buf, iters);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms = 0.f;
        hipEventElapsedTime(&ms, e0, e1);
        if (rep > 0 && ms < best) best = ms;
    }
    const double flop = (double)cus * blocks_per_cu * 4.0 * iters * 16.0 * (32.0 * 32.0 * 2.0 * 2.0);
    return flop / best / 1e9;   // TFLOP/s
}

int main() {
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    float* buf = nullptr;
    hipMalloc(&buf, (size_t)cus * 8 * 256 * sizeof(float));
    printf("%s: %d CUs, clockRate %d kHz -> nominal %.1f TFLOP/s at that clock\n", p.name, cus, p.clockRate,
           cus * 4.0 * 64.0 * p.clockRate * 1e3 / 1e12);
    const int iters = 30000;   // x 16 MFMAs x 64 clk = 30.7 M clk ~ 13 ms for one wave per SIMD
    for (int bpc = 1; bpc <= 4; bpc *= 2) {
        printf("  %d wave(s) per SIMD: 1 accumulator chain %.1f TF   2 chains %.1f TF   4 chains %.1f TF\n", bpc,
               run<1>(cus, bpc, iters / bpc, buf), run<2>(cus, bpc, iters / bpc, buf), run<4>(cus, bpc, iters / bpc, buf));
    }
    // a long steady run: does the rate sag with time (power / clock management)?
    for (int k = 1; k <= 8; k *= 2) printf("  4 waves per SIMD, 1 chain, %3d ms-class run: %.1f TF\n", 13 * k, run<1>(cus, 4, iters * k / 4, buf));
    return 0;
}
