// Development probe (not part of the library): what does ds_read_b64_tr_b16 deliver on this chip?
//   hipcc --offload-arch=gfx950 -O3 tools/gpu/tr16_probe.hip -o tools/gpu/tr16_probe
// LDS is filled with element ids (16-bit word i holds i); lane l reads 8 bytes at byte address l * 8 (+ a second pattern with a
// 64-byte row stride).  Prints, for every lane, the four element ids it received: the lane <-> (row, column) map of the
// hardware transpose, which unipose_amd/csrc/bf16s_glds.h (wgrad_glds_kernel) relies on.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef short s16x4 __attribute__((ext_vector_type(4)));
__global__ void k(int* out, int mode) {
    __shared__ __attribute__((aligned(16))) unsigned short lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (unsigned short)i;
    __syncthreads();
    const int lane = threadIdx.x;
    // mode 0: lane l reads 4 consecutive 16-bit words at word 4 * l.  mode 1: rows of 32 words (64 bytes): lane l of a 16-lane group
    // reads row l / 4, words 4 * (l % 4) .. + 3, groups 256 words apart.
    const int word = mode == 0 ? 4 * lane : (lane >> 4) * 256 + ((lane & 15) >> 2) * 32 + (lane & 3) * 4;
    s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(lds + word));
    for (int e = 0; e < 4; ++e) out[lane * 4 + e] = (unsigned short)v[e];
}
int main() {
    int* d;
    hipMalloc(&d, 256 * 4);
    for (int mode = 0; mode < 2; ++mode) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, mode);
        int h[256];
        hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf("mode %d (lane: four 16-bit element ids received)\n", mode);
        for (int l = 0; l < 64; ++l) printf("  lane %2d: %4d %4d %4d %4d\n", l, h[4 * l], h[4 * l + 1], h[4 * l + 2], h[4 * l + 3]);
    }
    return 0;
}
