// The network's first convolution (resnet.py:113: 7x7, stride 2, padding 3, 3 -> 64 channels) as a kernel of its own (round 6).
// Included by conv_igemm.hip inside namespace up, after f32_glds.h.
//
// igemm_kernel<128,64,generic> runs this layer at 44-53 TFLOP/s (481 us of the fp32 step at B = 32 / 368^2, 384 us alone on the GPU;
// 0.7 ms at B = 16 / 736^2): with 4 (3 + pad) input channels a 32-wide K slice is eight filter taps, and every tap of every
// gathered row is its own 16-byte load with its own bounds test.  Here:
//  * a tile is 128 CONSECUTIVE output pixels = at most two runs inside an output row (Q >= 128); the input pixels those runs
//    touch — 7 rows x (2 n + 5) columns x 16 B per run, zero-filled outside the image by the buffer descriptor's bounds check —
//    are staged ONCE in LDS (70 KB for both runs, LDS-DMA, one pixel per lane);
//  * the A operand of v_mfma_f32_32x32x2_f32 comes from ONE 16-byte LDS read per row block and k-group (a pixel's four channels) at
//    [lane base + immediate]: no address arithmetic in the loop;
//  * the B operand (196 x 64 weights) lives in REGISTERS, 100 per lane, loaded once per workgroup; a workgroup walks several tiles.
// Same k order and the same MFMA as igemm_kernel: equal bits (tests: op_cases.stem_ab_case).  The epilogue is igemm_epilogue.
//
// MEASURED (profiles/r06_experiments.txt item 15): alone on the GPU 384 -> 335 us = staging 62 + MFMA loop 208 (the pipe's time for
// 20 GFLOP is 176) + epilogue 53 + 14, the three phases NOT overlapping — the two workgroups of a CU run in step; inside the training
// step (input from HBM, not L2) the launch takes 0.58-0.67 ms against 0.48 and the step does not move (60.72-60.93 against
// 60.74-61.16 ms).  OFF by default (up_conv_tune("stem7", 1) / UP_STEM7=1); a second patch buffer (staging behind the MFMAs,
// one workgroup per CU) would be the next form, worth at most 0.1-0.2 ms of the step.
#pragma once

namespace glds {

constexpr int STEM_PW = 320;                       // patch row pitch in pixels: 2 * 128 + 5 columns, rounded up to whole 64-pixel LDS-DMA pieces
constexpr int STEM_SEG_BYTES = 7 * STEM_PW * 16;   // one run's patch
constexpr int STEM_LDS = 2 * STEM_SEG_BYTES;

__global__ void __launch_bounds__(256, 2) stem7_kernel(IgemmArgs a) {
    constexpr int BM = 128, BN = 64;
    __shared__ __attribute__((aligned(16))) unsigned char smem[STEM_LDS];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = uniform(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, lh = lane >> 5;
    const Rsrc rsA = make_rsrc(a.x, a.x_bytes);

    // k order of igemm_kernel (equal bits): K in groups of 8; MFMA e of group g multiplies k = 8 g + e (lanes 0..31) and
    // k = 8 g + 4 + e (lanes 32..63), i.e. lane half lh takes filter tap 2 g + lh, channel e.  25 groups: k >= 196 are zero weights.
    // B fragments: lane (column n = 32 wn + l31, half lh) holds w[n][8 g + 4 lh + e] for the 100 steps s = 4 g + e
    float bw[100];
    {
        const int n = wn * 32 + l31;
        const float* wr = a.w + (size_t)(n < a.Ng ? n : a.Ng - 1) * 196;
#pragma unroll
        for (int s = 0; s < 100; ++s) {
            const int k = 8 * (s >> 2) + 4 * lh + (s & 3);
            bw[s] = k < 196 ? wr[k] : 0.f;
        }
    }
    const int PQ = a.P * a.Q;
    const int ntiles = a.nwg;
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int mt = tile, m0 = tile * BM;
        // the two runs of the tile: pixels m0 .. m0 + n_0 - 1 inside output row (img, p) from column qa, the rest from column 0
        // of the next output row (which may be row 0 of the next image)
        const int img0 = fdiv(m0, a.fPQ);
        const int rem0 = m0 - img0 * PQ;
        const int p0 = fdiv(rem0, a.fQ);
        const int qa = rem0 - p0 * a.Q;
        const int n_0 = min(BM, a.Q - qa);
        int img1 = img0, p1 = p0 + 1;
        if (p1 == a.P) {
            p1 = 0;
            ++img1;
        }
        __syncthreads();   // the previous tile's epilogue is done with the LDS
        // ---- stage the patches: run g, patch row r, 64 pixels per LDS-DMA instruction; (g, r, chunk) dealt over the waves ----
#pragma unroll 1
        for (int job = wave; job < 2 * 7 * 5; job += 4) {
            const int g = job / 35, rr = (job - g * 35) / 5, ck = job - g * 35 - rr * 5;
            const int img = g ? img1 : img0, p = g ? p1 : p0, q_first = g ? 0 : qa, npx = g ? BM - n_0 : n_0;
            const int col = ck * 64 + lane;                 // patch column
            const int h = 2 * p - 3 + rr, w = 2 * q_first - 3 + col;
            const bool ok = npx > 0 && col < 2 * npx + 5 && img < a.M / PQ && (unsigned)h < (unsigned)a.H && (unsigned)w < (unsigned)a.W;
            load16_to_lds(rsA, ok ? (uint32_t)(((img * a.H + h) * a.W + w) * 16) : OOB,
                          smem + g * STEM_SEG_BYTES + (rr * STEM_PW + ck * 64) * 16);
        }
        // lane bases (16-byte aligned: a pixel's four channels) of the two 32-row blocks of this wave: pixel i of the tile -> run,
        // column inside the run.  The upper half-wave
        // reads the NEXT filter tap: one patch column further (baseN), or the first column of the next patch row when the lower
        // half's tap is the last of its filter row (baseW)
        int baseN[2], baseW[2];
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            int i = wm * 64 + b * 32 + l31;
            if (m0 + i >= a.M) i = a.M - 1 - m0;            // rows past the end repeat the last pixel (never stored)
            const int g = i >= n_0 ? 1 : 0;
            const int ql = g ? i - n_0 : i;
            const int pix = g * STEM_SEG_BYTES + 2 * ql * 16;
            baseN[b] = pix + lh * 16;
            baseW[b] = pix + lh * (STEM_PW - 6) * 16;
        }
        f32x16 acc[2][1];
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[b][0][r] = 0.f;
        wait_dma();
        __syncthreads();
        // ---- 25 k-groups: ONE 16-byte LDS read per row block and group (a pixel's four channels) at [lane base + immediate], the
        //      reads of group g + 1 pinned ahead of the eight MFMAs of group g, which alternate between the two row blocks (left
        //      alone the scheduler ran each block's 100 dependent MFMAs as one chain behind read, wait, MFMA x 4) ----
        f32x4 af[2][2];
        auto rd = [&](int g, int slot) {
            const int t = 2 * g;                                       // the lower half-wave's tap
            const int off = ((t / 7) * STEM_PW + (t % 7)) * 16;
            const bool wrap = t % 7 == 6 && g < 24;                    // (group 24's upper half meets zero weights: any finite pixel)
#pragma unroll
            for (int b = 0; b < 2; ++b) af[slot][b] = *reinterpret_cast<const f32x4*>(smem + (wrap ? baseW[b] : baseN[b]) + off);
        };
        rd(0, 0);
#pragma unroll
        for (int g = 0; g < 25; ++g) {
            if (g + 1 < 25) rd(g + 1, (g + 1) & 1);
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int b = 0; b < 2; ++b)
                    acc[b][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[g & 1][b][e], bw[4 * g + e], acc[b][0], 0, 0, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
#pragma unroll
        for (int g = 0; g < 25; ++g) {
            if (g + 1 < 25) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
        }
        __syncthreads();   // every wave is past its last patch read: the LDS becomes the epilogue's scratch
        igemm_epilogue<BM, BN, false>(a, acc, reinterpret_cast<float*>(smem), mt, m0, 0, wm, wn, l31, lh);
        igemm_fold_arrive<BN>(a, mt, 0, smem);
    }
}

// the launch conditions of stem7_kernel
static inline bool stem7_eligible(const IgemmArgs& a) {
    const long long bytes = (long long)(a.M / (a.P * a.Q)) * a.H * a.W * 16;
    return a.taps == 49 && a.S == 7 && a.Cp == 4 && a.ldx == 4 && a.mul == 2 && a.off0 == -3 && a.off0w == -3 && a.tapstep == 1 &&
           a.divshift == 0 && a.Ng == 64 && a.Ktot == 196 && a.Q >= 128 && !a.o_mode && !a.grp_rows && !a.perm && !a.bn_partial && !a.res_bits &&
           a.M % (a.P * a.Q) == 0 && bytes < (1ll << 31) && (reinterpret_cast<uintptr_t>(a.x) & 15) == 0;
}

}  // namespace glds
