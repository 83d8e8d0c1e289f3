// bf16-STORAGE implicit GEMM, second generation (BASELINE configs[4]: 736x736, B = 16, bf16 activations in HBM).
// Included by conv_igemm.hip inside namespace up, after IgemmArgs / xcd_remap / wf_merge.
//
// What differs from igemm_bf16_kernel<..., HS = true> (the round-1/2 register-staged kernel, kept as the fallback):
//  * operands go HBM -> LDS directly (`buffer_load_dwordx4 ... lds`, 1 KiB per wave-instruction): no staging registers,
//    no ds_write pass (the LDS store path needed 1.5x the cycles of the MFMAs it fed, profiles/r02_ae_sq_counters.txt),
//    no VALU zero selects: a row whose filter tap falls into the padding gets a byte offset beyond the buffer
//    descriptor's num_records and the hardware writes zeros into LDS; weight rows >= N are out of range the same way.
//  * K slice = 64 channels (128-byte LDS rows, 16 MFMAs per wave and barrier instead of 8), two LDS stages, ONE barrier
//    per slice: [wait own loads of slice t] -> barrier -> [issue slice t+1 into the other stage] -> MFMAs of slice t.
//  * the LDS image is lane-linear (destination = wave-uniform base + lane * 16), so the XOR swizzle that keeps the
//    ds_read_b128 fragment reads conflict-free is applied to the SOURCE chunk a lane fetches: LDS slot s of row r holds
//    logical 16-byte chunk s ^ ((r >> 1) & 7); the eight lanes of a row still cover one full 128-byte line.
//  * per-row set-up once per tile by ONE thread per row (separable row / column tap tests: R + S comparisons instead of
//    R * S) into an LDS table; tile-level tap skipping (union of the row masks) and tap-sorted rows (PERM) like the fp32 kernel.
//  * epilogue: BatchNorm partials on a straight-line path for full tiles; the tile is transposed through LDS (two
//    bf16 rows per word) and leaves as 16-byte stores of 8 consecutive channels, 4 rows x 256 contiguous bytes per
//    wave-instruction, instead of 64 two-byte stores per thread.
#pragma once

namespace glds {

constexpr int KT = 64;            // channels per K slice
constexpr int ROWB = KT * 2;      // bytes per LDS row
constexpr uint32_t OOB = 0x80000000u;   // byte offset beyond every descriptor's num_records (< 2^31, checked at launch)

#ifdef UP_EMU
struct Rsrc {
    const unsigned char* base;
    uint32_t bytes;
};
__device__ __forceinline__ Rsrc make_rsrc(const void* p, uint32_t bytes) { return Rsrc{static_cast<const unsigned char*>(p), bytes}; }
// one lane of `buffer_load_dwordx4 v, s[rsrc], 0 offen lds`: 16 bytes to wave_base + lane * 16, zeros when out of range
__device__ __forceinline__ void load16_to_lds(const Rsrc& rs, uint32_t voff, unsigned char* wave_base) {
    unsigned char* d = wave_base + (threadIdx.x & 63) * 16;
    if ((uint64_t)voff + 16 <= rs.bytes) memcpy(d, rs.base + voff, 16);
    else memset(d, 0, 16);
}
__device__ __forceinline__ int uniform(int v) { return v; }
__device__ __forceinline__ void wait_dma() {}
__device__ __forceinline__ uint32_t byte_perm(uint32_t hi, uint32_t lo, uint32_t sel) {
    const uint64_t src = ((uint64_t)hi << 32) | lo;
    uint32_t r = 0;
    for (int b = 0; b < 4; ++b) r |= (uint32_t)((src >> (8 * ((sel >> (8 * b)) & 7))) & 0xff) << (8 * b);
    return r;
}
__device__ __forceinline__ void lds_or(unsigned* p, unsigned v) { *p |= v; }   // fibers of a block run on one OS thread
#else
typedef __amdgpu_buffer_rsrc_t Rsrc;
__device__ __forceinline__ Rsrc make_rsrc(const void* p, uint32_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, 0x00020000);
}
__device__ __forceinline__ void load16_to_lds(const Rsrc& rs, uint32_t voff, unsigned char* wave_base) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)wave_base, 16, voff, 0, 0, 0);
}
__device__ __forceinline__ int uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }
// the LDS-DMA writes count on vmcnt; __syncthreads() drains them too, the explicit wait keeps that independent of the compiler
__device__ __forceinline__ void wait_dma() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
__device__ __forceinline__ uint32_t byte_perm(uint32_t hi, uint32_t lo, uint32_t sel) { return __builtin_amdgcn_perm(hi, lo, sel); }
__device__ __forceinline__ void lds_or(unsigned* p, unsigned v) { atomicOr(p, v); }
#endif

struct RowRec {
    int roff;        // byte offset of filter tap (0,0), channel 0 of this GEMM row in the activation tensor (may be negative)
    unsigned mask;   // bit t: tap t reads a real pixel (0 for rows >= M)
    int opix;        // destination pixel of the row, -1 for rows >= M
    int pad;
};

template <int BM, int BN>
struct Geom {
    static constexpr int A_BYTES = BM * ROWB, B_BYTES = BN * ROWB, STAGE = A_BYTES + B_BYTES;
    static constexpr int IMG_BYTES = BM * BN * 4;                        // fp32 epilogue image (addend / residual path)
    static constexpr int MAIN = 2 * STAGE > IMG_BYTES ? 2 * STAGE : IMG_BYTES;
    static constexpr int TAB_OFF = MAIN;                                 // RowRec[BM]
    static constexpr int MASK_OFF = TAB_OFF + BM * (int)sizeof(RowRec);  // tile tap mask (16 bytes reserved)
    static constexpr int STAT_OFF = MASK_OFF + 16;                       // BatchNorm exchange between the two M-waves
    static constexpr int TOTAL = STAT_OFF + BN * 12;
};

// BatchNorm partials of the wave's columns (count, mean, M2 over the tile's rows), exactly the arithmetic of igemm_epilogue;
// FULL: every row of the tile is a real row, so counts are literals and the per-element predicate disappears.
template <int BM, int BN, bool FULL>
__device__ __forceinline__ void tile_stats(const IgemmArgs& a, f32x16 (&acc)[BM / 64][BN / 64], float* xch, int mt, int m0, int n0,
                                           int wm, int wn, int l31, int lh) {
    constexpr int TM = BM / 64, TN = BN / 64;
    const int mrow0 = m0 + wm * (BM / 2) + 4 * lh;
    const int ncol0 = n0 + wn * (BN / 2) + l31;
    float sc[TN], sm[TN], s2[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        float cnt = 0.f, sum = 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = mrow0 + i * 32 + (r & 3) + 8 * (r >> 2);
                if (FULL || m < a.M) {
                    cnt += 1.f;
                    sum += acc[i][j][r];
                }
            }
        if (FULL) cnt = (float)(TM * 16);
        const float mean = cnt > 0.f ? sum / cnt : 0.f;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = mrow0 + i * 32 + (r & 3) + 8 * (r >> 2);
                if (FULL || m < a.M) {
                    const float d = acc[i][j][r] - mean;
                    q += d * d;
                }
            }
        float c1 = cnt, m1 = mean, q1 = q;
        float c2 = __shfl_xor(cnt, 32), m2 = __shfl_xor(mean, 32), q2 = __shfl_xor(q, 32);
        if (FULL) c2 = (float)(TM * 16);
        if (lh) {   // both halves merge in the same order to agree bitwise
            float tc = c2, tm = m2, tq = q2;
            wf_merge(tc, tm, tq, c1, m1, q1);
            c1 = tc;
            m1 = tm;
            q1 = tq;
        } else {
            wf_merge(c1, m1, q1, c2, m2, q2);
        }
        sc[j] = FULL ? (float)(TM * 32) : c1;
        sm[j] = m1;
        s2[j] = q1;
    }
    if (wm == 1 && lh == 0) {
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            float* d = xch + ((wn * TN + j) * 32 + l31) * 3;
            d[0] = sc[j];
            d[1] = sm[j];
            d[2] = s2[j];
        }
    }
    __syncthreads();
    if (wm == 0 && lh == 0) {
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const float* s = xch + ((wn * TN + j) * 32 + l31) * 3;
            wf_merge(sc[j], sm[j], s2[j], FULL ? (float)(TM * 32) : s[0], s[1], s[2]);
            const int n = ncol0 + j * 32;
            if (n < a.Ng) {
                float* o = a.stats + ((size_t)mt * a.Ng + n) * 3;
                o[0] = sc[j];
                o[1] = sm[j];
                o[2] = s2[j];
            }
        }
    }
}

// PERM: GEMM row m is output pixel a.perm[m] (tap-sorted order, see tap_sort_order).
template <int BM, int BN, bool PERM>
__global__ void __launch_bounds__(256, 2) igemm_glds_kernel(IgemmArgs a) {
    using G = Geom<BM, BN>;
    constexpr int TM = BM / 64, TN = BN / 64;
    constexpr int NA = BM / 32, NB = BN / 32;   // LDS-DMA instructions per wave, slice and operand (8 rows x 128 bytes each)
    __shared__ __attribute__((aligned(16))) unsigned char smem[G::TOTAL];
    RowRec* const tab = reinterpret_cast<RowRec*>(smem + G::TAB_OFF);
    unsigned* const tmask_s = reinterpret_cast<unsigned*>(smem + G::MASK_OFF);
    float* const xch = reinterpret_cast<float*>(smem + G::STAT_OFF);

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = uniform(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, lh = lane >> 5;

    const int logical = xcd_remap(blockIdx.x, a.nwg);
    const int mt = fdiv(logical, a.fNtn);
    const int nt = logical - mt * a.ntn;
    const int m0 = mt * BM, n0 = nt * BN;
    const int R = a.taps / a.S;

    if (tid == 0) *tmask_s = 0u;
    __syncthreads();
    if (tid < BM) {   // one thread per GEMM row of the tile
        const int m = m0 + tid;
        const bool in = m < a.M;
        int pix = in ? m : a.M - 1;
        if constexpr (PERM) pix = a.perm[pix];
        const int img = fdiv(pix, a.fPQ);
        const int rem = pix - img * (a.P * a.Q);
        const int p = fdiv(rem, a.fQ);
        const int q = rem - p * a.Q;
        const int hb = p * a.mul + a.off0, wb = q * a.mul + a.off0w;
        unsigned hm = 0, wmk = 0;
        for (int r = 0; r < R; ++r) {
            const int h = hb + r * a.tapstep;
            hm |= (h >= 0 && h < a.H) ? (1u << r) : 0u;
        }
        for (int s = 0; s < a.S; ++s) {
            const int w = wb + s * a.tapstep;
            wmk |= (w >= 0 && w < a.W) ? (1u << s) : 0u;
        }
        unsigned mk = 0;
        for (int r = 0; r < R; ++r) mk |= ((hm >> r) & 1u) ? (wmk << (r * a.S)) : 0u;
        if (!in) mk = 0u;
        RowRec rec;
        rec.roff = ((img * a.H + hb) * a.W + wb) * a.ldx * 2;
        rec.mask = mk;
        rec.opix = in ? pix : -1;
        rec.pad = 0;
        tab[tid] = rec;
        if (mk) lds_or(tmask_s, mk);
    }
    __syncthreads();

    // operand descriptors: the bounds check of the buffer load zero-fills padding rows (offset OOB) and weight rows >= N
    const Rsrc rsA = make_rsrc(a.x, a.x_bytes);
    const Rsrc rsB = make_rsrc(a.w_hi, (uint32_t)a.Ng * (uint32_t)a.Ktot * 2u);

    const int rsub = lane >> 3, slot = lane & 7;
    int roffA[NA];
    unsigned tmA[NA];
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        const int row = (wave + 4 * i) * 8 + rsub;
        roffA[i] = tab[row].roff + ((slot ^ ((row >> 1) & 7)) << 4);
        tmA[i] = tab[row].mask;
    }
    uint32_t woffB[NB];
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        const int row = (wave + 4 * j) * 8 + rsub;
        const int n = n0 + row;
        woffB[j] = n < a.Ng ? (uint32_t)n * (uint32_t)a.Ktot * 2u + (uint32_t)((slot ^ ((row >> 1) & 7)) << 4) : OOB;
    }

    // the K loop visits the slices of the filter taps that are live for at least one row of the tile
    const unsigned all_taps = a.taps >= 32 ? 0xffffffffu : ((1u << a.taps) - 1u);
    unsigned live = (unsigned)uniform((int)*tmask_s);
    if (a.no_tap_skip || live == 0u) live = all_taps;
    const int spt = a.Cp / KT;
    const int nsl = __builtin_popcount(live) * spt;
    unsigned rest = live;
    int tap = __builtin_ctz(rest), cs = 0;

    auto issue = [&](int stage) {
        unsigned char* const As = smem + stage * G::STAGE;
        unsigned char* const Bs = As + G::A_BYTES;
        const int r = fdiv(tap, a.fS);
        const int sx = tap - r * a.S;
        const int delta = ((r * a.tapstep) * a.W + sx * a.tapstep) * a.ldx * 2 + cs * ROWB;
        const uint32_t kb = (uint32_t)(tap * a.Cp + cs * KT) * 2u;
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const bool ok = (tmA[i] >> tap) & 1u;
            load16_to_lds(rsA, ok ? (uint32_t)(roffA[i] + delta) : OOB, As + (wave + 4 * i) * 1024);
        }
#pragma unroll
        for (int j = 0; j < NB; ++j)
            load16_to_lds(rsB, woffB[j] == OOB ? OOB : woffB[j] + kb, Bs + (wave + 4 * j) * 1024);
        if (++cs == spt) {   // next live tap
            cs = 0;
            rest &= rest - 1u;
            tap = rest ? __builtin_ctz(rest) : 0;
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int swz = ((l31 >> 1) & 7) << 4;
    const int a_rd = (wm * (BM / 2) + l31) * ROWB;
    const int b_rd = G::A_BYTES + (wn * (BN / 2) + l31) * ROWB;

    issue(0);
    for (int it = 0; it < nsl; ++it) {
        wait_dma();
        __syncthreads();   // slice `it` has landed for every wave, and every wave is done with the other stage
        if (it + 1 < nsl) issue((it + 1) & 1);
        const unsigned char* base = smem + (it & 1) * G::STAGE;
#pragma unroll
        for (int s = 0; s < KT / 16; ++s) {
            const int col = (((2 * s + lh) << 4) ^ swz);
            bf16x8 af[TM], bf[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) af[i] = *reinterpret_cast<const bf16x8*>(base + a_rd + i * 32 * ROWB + col);
#pragma unroll
            for (int j = 0; j < TN; ++j) bf[j] = *reinterpret_cast<const bf16x8*>(base + b_rd + j * 32 * ROWB + col);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bf[j], acc[i][j], 0, 0, 0);
        }
    }
    __syncthreads();   // every wave is past its last fragment read: the stages become the epilogue image

    // ---- epilogue ----
    const bool full = m0 + BM <= a.M;   // uniform
    if (a.stats) {
        if (full) tile_stats<BM, BN, true>(a, acc, xch, mt, m0, n0, wm, wn, l31, lh);
        else tile_stats<BM, BN, false>(a, acc, xch, mt, m0, n0, wm, wn, l31, lh);
    }
    const bool relu = a.relu != 0;
    const bool affine = a.scale != nullptr || a.bias != nullptr;
    float esc[TN], esh[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = n0 + wn * (BN / 2) + j * 32 + l31;
        const int nn = n < a.Ng ? n : a.Ng - 1;
        esc[j] = a.scale ? a.scale[nn] : 1.f;
        esh[j] = a.scale ? a.shift[nn] : 0.f;
        if (a.bias) esh[j] += a.bias[nn];
    }
    bf16_t* const yo = reinterpret_cast<bf16_t*>(a.y);
    if (!a.residual) {
        // image word [row pair][column] = (row 2rp, row 2rp + 1) of one channel: registers r, r+1 of the accumulator
        uint32_t* const img = reinterpret_cast<uint32_t*>(smem);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; r += 2) {
                    float v0 = acc[i][j][r], v1 = acc[i][j][r + 1];
                    if (affine) {
                        v0 = v0 * esc[j] + esh[j];
                        v1 = v1 * esc[j] + esh[j];
                    }
                    if (relu) {
                        v0 = fmaxf(v0, 0.f);
                        v1 = fmaxf(v1, 0.f);
                    }
                    const int row = wm * (BM / 2) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    img[(row >> 1) * BN + wn * (BN / 2) + j * 32 + l31] = pack_bf16x2(v0, v1);
                }
        __syncthreads();
        constexpr int CQ = BN / 8;                       // 16-byte chunks (8 channels) per row
        constexpr int UNITS = (BM / 2) * CQ / 256;       // (row pair, chunk) units per thread
#pragma unroll
        for (int k = 0; k < UNITS; ++k) {
            const int u = k * 256 + tid;
            const int rp = u / CQ, cq = u - rp * CQ;
            const uint4 w0 = *reinterpret_cast<const uint4*>(img + rp * BN + cq * 8);
            const uint4 w1 = *reinterpret_cast<const uint4*>(img + rp * BN + cq * 8 + 4);
            const int n = n0 + cq * 8;
            if (n >= a.Ng) continue;
            const int p0 = tab[2 * rp].opix, p1 = tab[2 * rp + 1].opix;
            if (p0 >= 0)
                *reinterpret_cast<uint4*>(yo + (size_t)p0 * a.ldy + n) =
                    make_uint4(byte_perm(w0.y, w0.x, 0x05040100u), byte_perm(w0.w, w0.z, 0x05040100u),
                               byte_perm(w1.y, w1.x, 0x05040100u), byte_perm(w1.w, w1.z, 0x05040100u));
            if (p1 >= 0)
                *reinterpret_cast<uint4*>(yo + (size_t)p1 * a.ldy + n) =
                    make_uint4(byte_perm(w0.y, w0.x, 0x07060302u), byte_perm(w0.w, w0.z, 0x07060302u),
                               byte_perm(w1.y, w1.x, 0x07060302u), byte_perm(w1.w, w1.z, 0x07060302u));
        }
    } else {
        // addend / residual: fp32 image [row][column]; the addend is added before the single rounding to bf16
        float* const img = reinterpret_cast<float*>(smem);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float v = acc[i][j][r];
                    if (affine) v = v * esc[j] + esh[j];
                    const int row = wm * (BM / 2) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    img[row * BN + wn * (BN / 2) + j * 32 + l31] = v;
                }
        __syncthreads();
        const bf16_t* const rs = reinterpret_cast<const bf16_t*>(a.residual);
        constexpr int CQ = BN / 8;
        constexpr int UNITS = BM * CQ / 256;
#pragma unroll
        for (int k = 0; k < UNITS; ++k) {
            const int u = k * 256 + tid;
            const int row = u / CQ, cq = u - row * CQ;
            const int n = n0 + cq * 8;
            const int px = tab[row].opix;
            if (n >= a.Ng || px < 0) continue;
            const float4 f0 = *reinterpret_cast<const float4*>(img + row * BN + cq * 8);
            const float4 f1 = *reinterpret_cast<const float4*>(img + row * BN + cq * 8 + 4);
            const uint4 rr = *reinterpret_cast<const uint4*>(rs + (size_t)px * a.ldr + n);
            float v[8] = {f0.x + bf_lo(rr.x), f0.y + bf_hi(rr.x), f0.z + bf_lo(rr.y), f0.w + bf_hi(rr.y),
                          f1.x + bf_lo(rr.z), f1.y + bf_hi(rr.z), f1.z + bf_lo(rr.w), f1.w + bf_hi(rr.w)};
            if (relu) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
            }
            *reinterpret_cast<uint4*>(yo + (size_t)px * a.ldy + n) =
                make_uint4(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7]));
        }
    }
}

}  // namespace glds
