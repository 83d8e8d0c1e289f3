// What costs the fp32 K loop its last 9 %?  (round 5: 142 TFLOP/s with the LDS-DMA loads removed, 130 with them issued, awaited
// or not.)  The register-only MFMA loop of mfma_peak.hip with the OTHER instructions of the convolution's slice added at the
// convolution's rate (per 16 MFMAs of a wave: 8 ds_read_b128 fragment reads, 4 LDS-DMA loads of 1 KiB from an L2-resident
// buffer), four waves per SIMD like four 64x64 workgroups per CU:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/gpu/mix_probe.hip -o tools/gpu/mix_probe && tools/gpu/mix_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#pragma clang diagnostic ignored "-Wunused-value"
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <bool READS, bool DMA, bool WAIT, int RPER = 1, int DPER = 1, bool AGPR = false>   // AGPR: accumulators in AccVGPRs (inline asm); RPER / DPER: fragment reads / LDS-DMA only every RPER-th / DPER-th iteration
__global__ void __launch_bounds__(256, 4) mix_loop(float* out, const float* src, int iters) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[32768];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    for (int i = tid; i < 8192; i += 256) reinterpret_cast<float*>(smem)[i] = (float)(i & 15) * 0.0625f;
    __syncthreads();
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0, 1 << 22, 0x00020000);
    float a = (float)(tid & 7) * 0.125f, b = 1.0f + (float)(tid & 3);
    const int rd = ((lane & 31) * 128 + (lane >> 5) * 16) & 16383;
    for (int it = 0; it < iters; ++it) {
        const int stage = (it & 1) * 16384;
        if (DMA && (it % DPER) == 0) {
            if constexpr (WAIT) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
            for (int j = 0; j < 4; ++j)   // this wave's share of a 16 KB slice: 4 x 1 KiB, a different 64 KB window per block
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(smem + (stage ^ 16384) + (wave * 4 + j) * 1024),
                                                         16, (uint32_t)(((blockIdx.x & 63) * 65536 + ((it & 3) * 16 + wave * 4 + j) * 1024 + lane * 16)), 0, 0, 0);
        }
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            f32x4 fa = {a, a, a, a}, fb = {b, b, b, b};
            if (READS && (it % RPER) == 0) {
                fa = *reinterpret_cast<const f32x4*>(smem + stage + ((rd + g * 32) & 8191));
                fb = *reinterpret_cast<const f32x4*>(smem + stage + 8192 + ((rd + g * 32) & 8191));
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if constexpr (AGPR) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+a"(acc) : "v"(fa[e]), "v"(fb[e]));
                else acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[e], fb[e], acc, 0, 0, 0);
            }
        }
    }
    float s = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) s += acc[r];
    if (s == 12345.678f) out[blockIdx.x * 256 + tid] = s;
}

template <bool READS, bool DMA, bool WAIT, int RPER = 1, int DPER = 1, bool AGPR = false>
static double run(float* buf, const float* src, int iters) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    double best = 1e30;
    for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL((mix_loop<READS, DMA, WAIT, RPER, DPER, AGPR>), dim3(256 * 4), dim3(256), 0, 0, buf, src, iters);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms = 0.f;
        hipEventElapsedTime(&ms, e0, e1);
        if (rep > 0 && ms < best) best = ms;
    }
    return 256.0 * 4 * 4 * iters * 16.0 * 4096.0 / best / 1e9;
}

// HYBRID: the A operand as in the convolution (LDS-DMA + fragment reads, i.e. half of the slice's LDS traffic), the B operand's
// fragments straight from global memory into registers (one 16-byte load per lane and k-group from 32 rows 128 bytes apart, the
// weight image's layout; L1 / L2 resident), prefetched one iteration ahead.
__global__ void __launch_bounds__(256, 4) hybrid_loop(float* out, const float* src, const float* wsrc, int iters) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[32768];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    for (int i = tid; i < 8192; i += 256) reinterpret_cast<float*>(smem)[i] = (float)(i & 15) * 0.0625f;
    __syncthreads();
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0, 1 << 22, 0x00020000);
    const int rd = ((lane & 31) * 128 + (lane >> 5) * 16) & 16383;
    // B rows of this wave: 32 rows of 32 floats (one K slice) = 4 KB per slice, 64 slices in a 256 KB window per block
    const f32x4* wb = reinterpret_cast<const f32x4*>(wsrc + (size_t)(blockIdx.x & 15) * 65536) + (lane & 31) * 8 + (lane >> 5);
    f32x4 nb[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) nb[g] = wb[g * 2];
    for (int it = 0; it < iters; ++it) {
        const int stage = (it & 1) * 16384;
        f32x4 fb[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) fb[g] = nb[g];
        const f32x4* nxt = wb + (size_t)((it + 1) & 63) * 256;
#pragma unroll
        for (int g = 0; g < 4; ++g) nb[g] = nxt[g * 2];   // next slice's B fragments: in flight during this slice's MFMAs
#pragma unroll
        for (int j = 0; j < 2; ++j)   // A only: 2 x 1 KiB per wave
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(smem + (stage ^ 16384) + (wave * 2 + j) * 1024),
                                                     16, (uint32_t)(((blockIdx.x & 63) * 65536 + ((it & 7) * 8 + wave * 2 + j) * 1024 + lane * 16)), 0, 0, 0);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 fa = *reinterpret_cast<const f32x4*>(smem + stage + ((rd + g * 32) & 8191));
#pragma unroll
            for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[e], fb[g][e], acc, 0, 0, 0);
        }
    }
    float s = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) s += acc[r];
    if (s == 12345.678f) out[blockIdx.x * 256 + tid] = s;
}
static double run_hybrid(float* buf, const float* src, const float* wsrc, int iters) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    double best = 1e30;
    for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(hybrid_loop, dim3(256 * 4), dim3(256), 0, 0, buf, src, wsrc, iters);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms = 0.f;
        hipEventElapsedTime(&ms, e0, e1);
        if (rep > 0 && ms < best) best = ms;
    }
    return 256.0 * 4 * 4 * iters * 16.0 * 4096.0 / best / 1e9;
}

int main() {
    float *buf, *src;
    hipMalloc(&buf, 256 * 4 * 256 * sizeof(float));
    hipMalloc(&src, 1 << 22);
    hipMemset(src, 0, 1 << 22);
    const int iters = 8000;
    printf("four 256-thread workgroups per CU, 16 MFMAs per wave and iteration (= one 64x64 K slice)\n");
    printf("  MFMA only                                      %.1f TFLOP/s\n", run<false, false, false>(buf, src, iters));
    printf("  + 8 ds_read_b128 per iteration                  %.1f\n", run<true, false, false>(buf, src, iters));
    printf("  + 4 LDS-DMA loads per iteration, never awaited  %.1f\n", run<false, true, false>(buf, src, iters));
    printf("  + both                                          %.1f\n", run<true, true, false>(buf, src, iters));
    printf("  + both, vmcnt(0) before the next issue          %.1f\n", run<true, true, true>(buf, src, iters));
    printf("  reads at half rate, LDS-DMA at full rate        %.1f\n", run<true, true, false, 2, 1>(buf, src, iters));
    printf("  reads at full rate, LDS-DMA at half rate        %.1f\n", run<true, true, false, 1, 2>(buf, src, iters));
    printf("  both at half rate (a 128x128 tile on 8 waves)   %.1f\n", run<true, true, false, 2, 2>(buf, src, iters));
    printf("  accumulators in AccVGPRs: MFMA only             %.1f\n", run<false, false, false, 1, 1, true>(buf, src, iters));
    printf("  accumulators in AccVGPRs: + reads               %.1f\n", run<true, false, false, 1, 1, true>(buf, src, iters));
    printf("  accumulators in AccVGPRs: + LDS-DMA             %.1f\n", run<false, true, false, 1, 1, true>(buf, src, iters));
    printf("  accumulators in AccVGPRs: + both                %.1f\n", run<true, true, false, 1, 1, true>(buf, src, iters));
    {
        float* wsrc;
        hipMalloc(&wsrc, 16 * 65536 * sizeof(float));
        hipMemset(wsrc, 0, 16 * 65536 * sizeof(float));
        printf("  HYBRID: A through LDS, B fragments from global  %.1f\n", run_hybrid(buf, src, wsrc, iters));
    }
    printf("  both at quarter rate                            %.1f\n", run<true, true, false, 4, 4>(buf, src, iters));
    return 0;
}
