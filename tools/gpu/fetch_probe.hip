// What one CU can FETCH from L2: LDS-DMA (buffer_load_dwordx4 ... lds, 1 KiB per wave-instruction) against plain 16-byte loads to
// registers, 8 waves per CU, every wave streaming 128-byte rows of an L2-resident buffer (the operand pattern of the bf16-storage
// implicit GEMM: 8 rows x 128 B per instruction).  hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/gpu/fetch_probe.hip -o fetch_probe
// Prints bytes per CU-cycle (2.4 GHz nominal) for: DMA with D instructions in flight per wave, register loads with D in flight,
// both together; with 256 / 64 workgroups (is the limit the CU's or the chip's?).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef __amdgpu_buffer_rsrc_t Rsrc;
__device__ __forceinline__ void dma16(const Rsrc& rs, uint32_t voff, unsigned char* wave_base) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)wave_base, 16, voff, 0, 0, 0);
}
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// MODE 0: LDS-DMA, 1: register loads, 2: one of each per step.  D: instructions in flight per wave.  row_stride: bytes between rows
template <int MODE, int D>
__global__ void __launch_bounds__(512, 1) fetch_kernel(const unsigned char* buf, uint32_t bytes, int iters, int row_stride, unsigned* sink) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[8 * D * 1024 + 16];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const Rsrc rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(buf), 0, bytes, 0x00020000);
    // a wave's instruction covers 8 rows x 128 B; the workgroup walks a region of its own (L2-resident after the first pass)
    const uint32_t region = bytes / gridDim.x & ~1023u;
    const uint32_t base = blockIdx.x * region;
    const uint32_t lane_off = (lane >> 3) * row_stride + (lane & 7) * 16;
    uint32_t pos = wave * 8 * row_stride;
    const uint32_t step = 8 * 8 * row_stride;   // the 8 waves together advance 64 rows per instruction round
    u32x4 acc = {0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
        u32x4 v[D];
#pragma unroll
        for (int d = 0; d < D; ++d) {
            const uint32_t off = base + (pos % (region - step)) + lane_off;
            if (MODE == 0 || (MODE == 2 && (d & 1) == 0)) dma16(rs, off, smem + (wave * D + d) * 1024);
            else v[d] = *reinterpret_cast<const u32x4*>(buf + off);
            pos += step;
        }
        if (MODE != 0) {
#pragma unroll
            for (int d = 0; d < D; ++d)
                if (MODE == 1 || (d & 1)) acc ^= v[d];
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    if (acc[0] == 0x12345678u) sink[0] = acc[1] + smem[threadIdx.x];
}

template <int MODE, int D>
static double run(const unsigned char* buf, uint32_t bytes, int grid, int row_stride, unsigned* sink) {
    const int iters = 2000 / D;
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    hipLaunchKernelGGL((fetch_kernel<MODE, D>), dim3(grid), dim3(512), 0, 0, buf, bytes, iters, row_stride, sink);   // warm L2
    hipEventRecord(a, 0);
    hipLaunchKernelGGL((fetch_kernel<MODE, D>), dim3(grid), dim3(512), 0, 0, buf, bytes, iters, row_stride, sink);
    hipEventRecord(b, 0);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    const double per_cu = (double)iters * D * 8 * 1024;   // bytes one workgroup (CU) fetched
    return per_cu / (ms * 1e-3) / 2.4e9;                  // bytes per 2.4 GHz cycle and CU
}

int main() {
    const uint32_t bytes = 24u << 20;   // 24 MB over 256 workgroups: 96 KB each, L2-resident (4 MB per XCD, 32 workgroups per XCD)
    unsigned char* buf;
    unsigned* sink;
    hipMalloc(&buf, bytes);
    hipMalloc(&sink, 64);
    hipMemset(buf, 1, bytes);
    for (int grid : {256, 64}) {
        for (int rs : {128, 512}) {
            printf("grid %3d row stride %3d B | DMA D=2/4/8: %5.1f %5.1f %5.1f | registers D=2/4/8: %5.1f %5.1f %5.1f | mixed D=4/8: %5.1f %5.1f  B/clk/CU\n", grid, rs,
                   run<0, 2>(buf, bytes, grid, rs, sink), run<0, 4>(buf, bytes, grid, rs, sink), run<0, 8>(buf, bytes, grid, rs, sink),
                   run<1, 2>(buf, bytes, grid, rs, sink), run<1, 4>(buf, bytes, grid, rs, sink), run<1, 8>(buf, bytes, grid, rs, sink),
                   run<2, 4>(buf, bytes, grid, rs, sink), run<2, 8>(buf, bytes, grid, rs, sink));
        }
    }
    return 0;
}
