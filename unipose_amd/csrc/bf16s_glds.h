// bf16-STORAGE implicit GEMM, second generation (BASELINE configs[4]: 736x736, B = 16, bf16 activations in HBM).
// Included by conv_igemm.hip inside namespace up, after IgemmArgs / WgradArgs / xcd_remap / wf_merge.
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
#endif

#ifdef UP_EMU
typedef float f32x2 __attribute__((vector_size(8)));
#else
typedef float f32x2 __attribute__((ext_vector_type(2)));
#endif

// geometry of a K slice of KT channels: LDS rows of KT*2 bytes, CH 16-byte chunks per row, RPI rows per LDS-DMA instruction;
// chunk c of row r sits at slot c ^ swz(r): 128-byte rows (r >> 1) & 7, 64-byte rows (r >> 2) & 3 — in both cases the 16 rows of a
// ds_read_b128 lane group land on 16 different 16-byte slots of the 256-byte bank row.
template <int KT>
struct Slice {
    static_assert(KT == 64 || KT == 32, "K slice of 64 or 32 channels");
    static constexpr int ROWB = KT * 2, CH = KT / 8, RPI = 1024 / ROWB;
    __host__ __device__ static constexpr int swz(int r) { return KT == 64 ? ((r >> 1) & 7) : ((r >> 2) & 3); }
};

template <int BM, int BN, int KT, int ST>
struct Geom {
    static constexpr int A_BYTES = BM * Slice<KT>::ROWB, B_BYTES = BN * Slice<KT>::ROWB, STAGE = A_BYTES + B_BYTES;
    static constexpr int IMG_BYTES = BM * BN * 2;                        // epilogue image: bf16 row pairs, or fp32 for half the rows
    static constexpr int MAIN = ST * STAGE > IMG_BYTES ? ST * STAGE : IMG_BYTES;
    static constexpr int STAT_OFF = MAIN;                                // BatchNorm exchange between the two M-waves
    static constexpr int MASK_OFF = STAT_OFF + BN * 12;                  // tap masks of the four waves
    static constexpr int TOTAL = MASK_OFF + 16;
};

// BatchNorm partials of the wave's columns (count, mean, M2 over the tile's rows): two-pass in registers and Welford merges like
// igemm_epilogue.  FULL: every row of the tile is a real row: counts are literals, no per-element predicate, and the two
// passes run on row PAIRS with packed fp32 arithmetic (half the instructions; even / odd rows are summed separately).
template <int BM, int BN, bool FULL>
__device__ __forceinline__ void tile_stats(const IgemmArgs& a, f32x16 (&acc)[BM / 64][BN / 64], float* xch, int mt, int m0, int n0,
                                           int wm, int wn, int l31, int lh) {
    constexpr int TM = BM / 64, TN = BN / 64;
    const int mrow0 = m0 + wm * (BM / 2) + 4 * lh;
    const int ncol0 = n0 + wn * (BN / 2) + l31;
    float sc[TN], sm[TN], s2[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        float cnt, mean, q;
        if constexpr (FULL) {
            // two rows per operation (v_pk_add_f32 / v_pk_fma_f32): registers r, r + 1 hold consecutive rows
            f32x2 s2v = {0.f, 0.f};
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int r = 0; r < 16; r += 2) s2v += f32x2{acc[i][j][r], acc[i][j][r + 1]};
            cnt = (float)(TM * 16);
            mean = (s2v[0] + s2v[1]) / cnt;
            const f32x2 mu = {mean, mean};
            f32x2 q2v = {0.f, 0.f};
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int r = 0; r < 16; r += 2) {
                    const f32x2 d = f32x2{acc[i][j][r], acc[i][j][r + 1]} - mu;
                    q2v += d * d;
                }
            q = q2v[0] + q2v[1];
        } else {
            float sum = 0.f;
            cnt = 0.f;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = mrow0 + i * 32 + (r & 3) + 8 * (r >> 2);
                    if (m < a.M) {
                        cnt += 1.f;
                        sum += acc[i][j][r];
                    }
                }
            mean = cnt > 0.f ? sum / cnt : 0.f;
            q = 0.f;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = mrow0 + i * 32 + (r & 3) + 8 * (r >> 2);
                    if (m < a.M) {
                        const float d = acc[i][j][r] - mean;
                        q += d * d;
                    }
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
                float* o = a.stats + ((size_t)mt * a.Ng + n) * 3;   // (sc1: the last arriver of the fold reads them)
                st_agent(o, sc[j]);
                st_agent(o + 1, sm[j]);
                st_agent(o + 2, s2[j]);
            }
        }
    }
}

// (element offsets into y / the residual fit 31 bits: checked at launch)
// Epilogue of the bf16-storage kernels: BatchNorm partials, optional folded scale / shift / bias / ReLU, then the tile leaves
// through LDS as 16-byte stores of 8 consecutive channels.  `img` (>= BM*BN*2 bytes, 16-byte aligned) must be free: every
// wave is past its last fragment read.  Row r of the tile is output pixel perm[m0 + r] (PERM) or m0 + r.
//   no residual: image word [row pair][column] = bf16 (row 2rp, row 2rp + 1) of one channel (registers r, r + 1 of an accumulator
//                hold consecutive rows); a thread reads 8 words = 8 channels x 2 rows, two byte-permutes per output word
//   residual / addend: fp32 image of half the rows at a time; the addend is added before the single rounding to bf16
//   BNRED (round 4): this launch's output is dz of the layer z = relu(bn(y) (+ res)); the read-out loops also accumulate that
//                layer's BatchNorm-backward sums (a.bn_*: sum g, sum g * (y - mean) per channel, g = dz * [z > 0]) on the rows
//                they are about to store, and the tile's sums go to a.bn_partial[row tile] (f32_glds.h has the fp32 twin)
//   a.res_bits:  the addend is an UNMASKED dz of another layer; its ReLU mask (bit pixel * Ng + channel) is applied here
template <int BM, int BN, bool PERM, bool BNRED = false>
__device__ __forceinline__ void store_tile(const IgemmArgs& a, f32x16 (&acc)[BM / 64][BN / 64], unsigned char* img_mem, float* xch,
                                           int mt, int m0, int n0, int tid, int wm, int wn, int l31, int lh) {
    constexpr int TM = BM / 64, TN = BN / 64;
    static_assert(!BNRED || BM <= 128, "the fused reduction is written for one partial row per tile");
    float s1a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, s2a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    float mu8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    // g = v * [bit set], accumulated with the BatchNorm operand row yy (8 bf16) of the same pixel
    auto bn_acc = [&](const float (&v)[8], const uint4& yy, uint32_t bits8) {
        const float y[8] = {bf_lo(yy.x), bf_hi(yy.x), bf_lo(yy.y), bf_hi(yy.y), bf_lo(yy.z), bf_hi(yy.z), bf_lo(yy.w), bf_hi(yy.w)};
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float g = ((bits8 >> e) & 1u) ? v[e] : 0.f;
            s1a[e] += g;
            s2a[e] += g * (y[e] - mu8[e]);
        }
    };
    if constexpr (BNRED) {   // the batch means of this thread's 8 channels (its chunk column is the same in every unit)
        const int nb = n0 + (tid % (BN / 8)) * 8;
        if (nb < a.Ng) {
            const float4 m0v = *reinterpret_cast<const float4*>(a.bn_mean + nb), m1v = *reinterpret_cast<const float4*>(a.bn_mean + nb + 4);
            mu8[0] = m0v.x; mu8[1] = m0v.y; mu8[2] = m0v.z; mu8[3] = m0v.w;
            mu8[4] = m1v.x; mu8[5] = m1v.y; mu8[6] = m1v.z; mu8[7] = m1v.w;
        }
    }
    // the 8 sign bits of channels n .. n + 7 of pixel p (n is a multiple of 8: one byte of the bit array)
    auto bits8_of = [&](const uint32_t* bits, int C, int p, int n) -> uint32_t {
        const long long b = (long long)p * C + n;
        return (bits[b >> 5] >> (int)(b & 31)) & 0xffu;
    };
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
    auto opix = [&](int row) {   // destination pixel of tile row `row`, -1 past the end
        const int m = m0 + row;
        if (m >= a.M) return -1;
        if constexpr (PERM) return a.perm[m];
        return m;
    };
    bf16_t* const yo = reinterpret_cast<bf16_t*>(a.y);
    constexpr int CQ = BN / 8;   // 16-byte chunks (8 channels) per row
    if (!a.residual) {
        uint32_t* const img = reinterpret_cast<uint32_t*>(img_mem);
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
        constexpr int UNITS = (BM / 2) * CQ / 256;   // (row pair, chunk) units per thread
        // BNRED: the operands of the fused reduction (16 bytes of y and the 8 sign bits per row) are requested HERE, before the
        // barrier, so that their latency overlaps the image write and the barrier instead of sitting in front of every unit
        // (the first version loaded them inside the loop below: the fused launches lost what the removed pass had cost)
        uint4 ya[BNRED ? UNITS : 1], yb[BNRED ? UNITS : 1];
        uint32_t ba[BNRED ? UNITS : 1], bb[BNRED ? UNITS : 1];
        if constexpr (BNRED) {
            const int nq = n0 + (tid % CQ) * 8;
            const int nn = nq < a.Ng ? nq : 0;
            const bf16_t* const ybn = reinterpret_cast<const bf16_t*>(a.bn_y);
#pragma unroll
            for (int k = 0; k < UNITS; ++k) {
                const int rp = (k * 256 + tid) / CQ;
                const int q0 = opix(2 * rp), q1 = opix(2 * rp + 1);
                const int r0 = q0 >= 0 ? q0 : 0, r1 = q1 >= 0 ? q1 : 0;
                ya[k] = *reinterpret_cast<const uint4*>(ybn + (uint32_t)(r0 * a.bn_ld + nn));
                yb[k] = *reinterpret_cast<const uint4*>(ybn + (uint32_t)(r1 * a.bn_ld + nn));
                ba[k] = a.bn_bits ? bits8_of(a.bn_bits, a.bn_C, r0, nn) : 0xffu;
                bb[k] = a.bn_bits ? bits8_of(a.bn_bits, a.bn_C, r1, nn) : 0xffu;
            }
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < UNITS; ++k) {
            const int u = k * 256 + tid;
            const int rp = u / CQ, cq = u - rp * CQ;
            const uint4 w0 = *reinterpret_cast<const uint4*>(img + rp * BN + cq * 8);
            const uint4 w1 = *reinterpret_cast<const uint4*>(img + rp * BN + cq * 8 + 4);
            const int n = n0 + cq * 8;
            if (n >= a.Ng) continue;
            const int p0 = opix(2 * rp), p1 = opix(2 * rp + 1);
            if constexpr (BNRED) {
                const uint32_t wv[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
                if (p0 >= 0) {
                    float v[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = bf_lo(wv[e]);
                    bn_acc(v, ya[k], ba[k]);
                }
                if (p1 >= 0) {
                    float v[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = bf_hi(wv[e]);
                    bn_acc(v, yb[k], bb[k]);
                }
            }
            if (p0 >= 0)
                *reinterpret_cast<uint4*>(yo + (uint32_t)(p0 * a.ldy + n)) =
                    make_uint4(byte_perm(w0.y, w0.x, 0x05040100u), byte_perm(w0.w, w0.z, 0x05040100u),
                               byte_perm(w1.y, w1.x, 0x05040100u), byte_perm(w1.w, w1.z, 0x05040100u));
            if (p1 >= 0)
                *reinterpret_cast<uint4*>(yo + (uint32_t)(p1 * a.ldy + n)) =
                    make_uint4(byte_perm(w0.y, w0.x, 0x07060302u), byte_perm(w0.w, w0.z, 0x07060302u),
                               byte_perm(w1.y, w1.x, 0x07060302u), byte_perm(w1.w, w1.z, 0x07060302u));
        }
    } else {
        float* const img = reinterpret_cast<float*>(img_mem);
        const bf16_t* const rs = reinterpret_cast<const bf16_t*>(a.residual);
        constexpr int UNITS = (BM / 2) * CQ / 256;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            if (half) __syncthreads();   // the first half has been read
            // the addend rows of this half (and the operands of the fused reduction) are requested before the image write
            uint4 radd[UNITS], ybn4[BNRED ? UNITS : 1];
            uint32_t rmask[UNITS], bmask[BNRED ? UNITS : 1];
            {
                const int nq = n0 + (tid % CQ) * 8;
                const int nn = nq < a.Ng ? nq : 0;
#pragma unroll
                for (int k = 0; k < UNITS; ++k) {
                    const int q = opix(half * (BM / 2) + (k * 256 + tid) / CQ);
                    const int r = q >= 0 ? q : 0;
                    radd[k] = *reinterpret_cast<const uint4*>(rs + (uint32_t)(r * a.ldr + nn));
                    rmask[k] = a.res_bits ? bits8_of(a.res_bits, a.Ng, r, nn) : 0xffu;
                    if constexpr (BNRED) {
                        ybn4[k] = *reinterpret_cast<const uint4*>(reinterpret_cast<const bf16_t*>(a.bn_y) + (uint32_t)(r * a.bn_ld + nn));
                        bmask[k] = a.bn_bits ? bits8_of(a.bn_bits, a.bn_C, r, nn) : 0xffu;
                    }
                }
            }
            if (wm == half) {
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            float v = acc[i][j][r];
                            if (affine) v = v * esc[j] + esh[j];
                            const int row = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;   // inside this half
                            img[row * BN + wn * (BN / 2) + j * 32 + l31] = v;
                        }
            }
            __syncthreads();
#pragma unroll
            for (int k = 0; k < UNITS; ++k) {
                const int u = k * 256 + tid;
                const int row = u / CQ, cq = u - row * CQ;
                const int n = n0 + cq * 8;
                const int px = opix(half * (BM / 2) + row);
                if (n >= a.Ng || px < 0) continue;
                const float4 f0 = *reinterpret_cast<const float4*>(img + row * BN + cq * 8);
                const float4 f1 = *reinterpret_cast<const float4*>(img + row * BN + cq * 8 + 4);
                const uint4 rr = radd[k];
                float ad[8] = {bf_lo(rr.x), bf_hi(rr.x), bf_lo(rr.y), bf_hi(rr.y), bf_lo(rr.z), bf_hi(rr.z), bf_lo(rr.w), bf_hi(rr.w)};
                {   // the addend's ReLU mask (all ones without one), applied here
                    const uint32_t mb = rmask[k];
#pragma unroll
                    for (int e = 0; e < 8; ++e) ad[e] = ((mb >> e) & 1u) ? ad[e] : 0.f;
                }
                float v[8] = {f0.x + ad[0], f0.y + ad[1], f0.z + ad[2], f0.w + ad[3], f1.x + ad[4], f1.y + ad[5], f1.z + ad[6], f1.w + ad[7]};
                if (relu) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
                }
                const uint4 packed = make_uint4(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7]));
                if constexpr (BNRED) {
                    // the STORED (rounded) values, like the separate reduction pass reads them back
                    const float vr[8] = {bf_lo(packed.x), bf_hi(packed.x), bf_lo(packed.y), bf_hi(packed.y),
                                         bf_lo(packed.z), bf_hi(packed.z), bf_lo(packed.w), bf_hi(packed.w)};
                    bn_acc(vr, ybn4[k], bmask[k]);
                }
                *reinterpret_cast<uint4*>(yo + (uint32_t)(px * a.ldy + n)) = packed;
            }
        }
    }
    if constexpr (BNRED) {
        // lanes that share a chunk column are CQ apart; then the four waves through LDS (the image is free after one more
        // barrier), summed in a fixed order: deterministic
#pragma unroll
        for (int off = CQ; off < 64; off <<= 1)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                s1a[e] += __shfl_xor(s1a[e], off);
                s2a[e] += __shfl_xor(s2a[e], off);
            }
        __syncthreads();   // every thread is done with the image
        float* const red = reinterpret_cast<float*>(img_mem);
        const int wave = tid >> 6, lane = tid & 63, cq = tid % CQ;
        if (lane < CQ) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                red[(wave * BN + cq * 8 + e) * 2] = s1a[e];
                red[(wave * BN + cq * 8 + e) * 2 + 1] = s2a[e];
            }
        }
        __syncthreads();
        if (tid < 2 * BN) {
            const int ch = tid >> 1, which = tid & 1;
            float t = 0.f;
#pragma unroll
            for (int w = 0; w < 4; ++w) t += red[(w * BN + ch) * 2 + which];
            const int cc = n0 + ch;
            if (cc < a.Ng) {
                if (which) t *= a.bn_invstd[cc];
                st_agent(a.bn_partial + ((size_t)mt * a.Ng + cc) * 2 + which, t);
            }
        }
    }
}

// PERM: GEMM row m is output pixel a.perm[m] (tap-sorted order, see tap_sort_order).
// KT:   channels per K slice (64: 16 MFMAs per wave and barrier, 32 KB per stage of a 128x128 tile; 32: half of both).
// ST:   LDS stages.  2: one barrier per slice, the next slice in flight during the MFMAs.  3: two slices in flight, counted
//       vmcnt + raw s_barrier (a __syncthreads() would drain the LDS-DMA queue).
// EPI:  0 = store_tile (LDS transpose, 16-byte stores), 1 = igemm_epilogue (the register-staged kernel's 2-byte stores; probe).
// DBG:  probe only (tools/gpu/glds_probe.hip): eight 100 MHz time stamps per workgroup.
template <int BM, int BN, bool PERM, int KT = 64, int ST = 2, int OCC = 2, int EPI = 0, int DBG = 0, bool BNRED = false>
__global__ void __launch_bounds__(256, OCC) igemm_glds_kernel(IgemmArgs a) {
    using SL = Slice<KT>;
    using G = Geom<BM, BN, KT, ST>;
    constexpr int TM = BM / 64, TN = BN / 64;
    constexpr int NA = BM / (4 * SL::RPI), NB = BN / (4 * SL::RPI);   // LDS-DMA instructions per wave, slice and operand
    static_assert(NA >= 1 && NB >= 1, "a wave issues at least one LDS-DMA instruction per operand");
    __shared__ __attribute__((aligned(16))) unsigned char smem[G::TOTAL];
    float* const xch = reinterpret_cast<float*>(smem + G::STAT_OFF);
    unsigned* const wmask = reinterpret_cast<unsigned*>(smem + G::MASK_OFF);

    long long stamp[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const int tid = threadIdx.x;
    if (DBG && tid == 0) stamp[0] = wall_clock64();
    const int lane = tid & 63, wave = uniform(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, lh = lane >> 5;

    // (no K-split of tail tiles here: in bf16 storage a cut re-associates fp32 sums batch-size-dependently and one last-bit
    //  difference can flip the rounding of a stored activation; measured 0.15 ms of 40 at best, r03_s — removed in round 4)
    const int logical = xcd_remap(blockIdx.x, a.nwg);
    const int mt = fdiv(logical, a.fNtn);
    const int nt = logical - mt * a.ntn;
    const int m0 = mt * BM, n0 = nt * BN;

    // this lane's rows of the operand tiles: row (wave + 4 i) * RPI + lane / CH, 16-byte slot lane % CH
    const int rsub = lane / SL::CH, slot = lane % SL::CH;
    const int R = a.taps / a.S;
    // operand descriptors: the bounds check of the buffer load zero-fills padding rows (offset OOB) and weight rows >= N
    const Rsrc rsA = make_rsrc(a.x, a.x_bytes);
    const Rsrc rsB = make_rsrc(a.w_hi, (uint32_t)a.Ng * (uint32_t)a.Ktot * 2u);
    uint32_t woffB[NB];
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        const int row = (wave + 4 * j) * SL::RPI + rsub;
        const int n = n0 + row;
        woffB[j] = n < a.Ng ? (uint32_t)n * (uint32_t)a.Ktot * 2u + (uint32_t)((slot ^ SL::swz(row)) << 4) : OOB;
    }

    auto issueB = [&](int stage, int tap_, int cs_) {
        unsigned char* const Bs = smem + stage * G::STAGE + G::A_BYTES;
        const uint32_t kb = (uint32_t)(tap_ * a.Cp + cs_ * KT) * 2u;
#pragma unroll
        for (int j = 0; j < NB; ++j)
            load16_to_lds(rsB, woffB[j] == OOB ? OOB : woffB[j] + kb, Bs + (wave + 4 * j) * 1024);
    };
    // 1x1, stride 1, no padding (two thirds of the launches): the source pixel IS the destination pixel, and the first weight
    // slice does not depend on any row set-up: it is on its way before the rows are looked at
    const bool pointwise = a.taps == 1 && a.mul == 1 && a.off0 == 0 && a.off0w == 0 && a.H == a.P && a.W == a.Q;
    const bool early_b = pointwise;
    if (early_b) issueB(0, 0, 0);

    int roffA[NA];        // byte offset of (filter tap (0,0), this lane's chunk) of the row in the activation tensor
    unsigned tmA[NA];     // bit t: tap t of the row reads a real pixel (0 for rows >= M)
    unsigned tile_taps = 0;
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        const int row = (wave + 4 * i) * SL::RPI + rsub;
        const int m = m0 + row;
        int pix = m < a.M ? m : a.M - 1;
        if constexpr (PERM) pix = a.perm[pix];
        unsigned mk;
        int src;   // source pixel of filter tap (0,0)
        if (pointwise) {
            src = pix;
            mk = 1u;
        } else {
            const int img = fdiv(pix, a.fPQ);
            const int rem = pix - img * (a.P * a.Q);
            const int p = fdiv(rem, a.fQ);
            const int q = rem - p * a.Q;
            const int hb = p * a.mul + a.off0, wb = q * a.mul + a.off0w;
            src = (img * a.H + hb) * a.W + wb;
            // separable test: R + S comparisons instead of R * S
            if (a.taps == 9 && a.S == 3) {
                unsigned hm = 0, wmk = 0;
#pragma unroll
                for (int t = 0; t < 3; ++t) {
                    hm |= ((unsigned)(hb + t * a.tapstep) < (unsigned)a.H) ? (1u << t) : 0u;
                    wmk |= ((unsigned)(wb + t * a.tapstep) < (unsigned)a.W) ? (1u << t) : 0u;
                }
                mk = ((hm & 1u) ? wmk : 0u) | ((hm & 2u) ? (wmk << 3) : 0u) | ((hm & 4u) ? (wmk << 6) : 0u);
            } else {
                unsigned hm = 0, wmk = 0;
                for (int r = 0; r < R; ++r) hm |= ((unsigned)(hb + r * a.tapstep) < (unsigned)a.H) ? (1u << r) : 0u;
                for (int s = 0; s < a.S; ++s) wmk |= ((unsigned)(wb + s * a.tapstep) < (unsigned)a.W) ? (1u << s) : 0u;
                mk = 0u;
                for (int r = 0; r < R; ++r) mk |= ((hm >> r) & 1u) ? (wmk << (r * a.S)) : 0u;
            }
        }
        if (m >= a.M) mk = 0u;
        roffA[i] = src * a.ldx * 2 + ((slot ^ SL::swz(row)) << 4);
        tmA[i] = mk;
        tile_taps |= mk;
    }
    // the K loop visits the slices of the filter taps that are live for at least one row of the tile
    const unsigned all_taps = a.taps >= 32 ? 0xffffffffu : ((1u << a.taps) - 1u);
    unsigned live = all_taps;
    if (a.taps > 1 && !a.no_tap_skip) {
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) tile_taps |= __shfl_xor(tile_taps, off);
        if (lane == 0) wmask[wave] = tile_taps;
        __syncthreads();
        live = (unsigned)uniform((int)(wmask[0] | wmask[1] | wmask[2] | wmask[3]));
        if (live == 0u) live = all_taps;
    }
    const int spt = a.Cp / KT;
    const int nsl = __builtin_popcount(live) * spt;
    unsigned rest = live;
    int tap = __builtin_ctz(rest), cs = 0;

    bool b_issued = early_b;   // the weight slice of the first issue is already on its way
    auto issue = [&](int stage) {
        unsigned char* const As = smem + stage * G::STAGE;
        const int r = fdiv(tap, a.fS);
        const int sx = tap - r * a.S;
        const int delta = ((r * a.tapstep) * a.W + sx * a.tapstep) * a.ldx * 2 + cs * SL::ROWB;
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const bool ok = (tmA[i] >> tap) & 1u;
            load16_to_lds(rsA, ok ? (uint32_t)(roffA[i] + delta) : OOB, As + (wave + 4 * i) * 1024);
        }
        if (!b_issued) issueB(stage, tap, cs);
        b_issued = false;
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

    const int swz = SL::swz(l31) << 4;
    const int a_rd = (wm * (BM / 2) + l31) * SL::ROWB;
    const int b_rd = G::A_BYTES + (wn * (BN / 2) + l31) * SL::ROWB;
    auto mfmas = [&](const unsigned char* base) {
#pragma unroll
        for (int s = 0; s < KT / 16; ++s) {
            const int col = (((2 * s + lh) << 4) ^ swz);
            bf16x8 af[TM], bf[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) af[i] = *reinterpret_cast<const bf16x8*>(base + a_rd + i * 32 * SL::ROWB + col);
#pragma unroll
            for (int j = 0; j < TN; ++j) bf[j] = *reinterpret_cast<const bf16x8*>(base + b_rd + j * 32 * SL::ROWB + col);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bf[j], acc[i][j], 0, 0, 0);
        }
    };

    if (DBG && tid == 0) stamp[1] = wall_clock64();
    static_assert(ST == 2, "two LDS stages (the three-stage form measured no faster and left in round 4)");
    {
        if (nsl > 0) issue(0);
        for (int it = 0; it < nsl; ++it) {
            wait_dma();
            __syncthreads();   // slice `it` has landed for every wave, and every wave is done with the other stage
            if (DBG && tid == 0 && it == 0) stamp[2] = wall_clock64();
            if (it + 1 < nsl) issue((it + 1) & 1);
            mfmas(smem + (it & 1) * G::STAGE);
        }
    }
    __syncthreads();   // every wave is past its last fragment read: the stages become the epilogue image
    if (DBG && tid == 0) stamp[3] = wall_clock64();

    if constexpr (EPI == 1) {
        igemm_epilogue<BM, BN, PERM, bf16_t>(a, acc, reinterpret_cast<float*>(smem), mt, m0, n0, wm, wn, l31, lh);
    } else {
        store_tile<BM, BN, PERM, BNRED>(a, acc, smem, xch, mt, m0, n0, tid, wm, wn, l31, lh);
    }
    igemm_fold_arrive<BN>(a, mt, n0, smem);
    if (DBG && tid == 0) {
        stamp[4] = wall_clock64();
        wait_dma();   // stores of this wave acknowledged
        stamp[5] = wall_clock64();
        long long* o = reinterpret_cast<long long*>(a.dbg) + 8 * (size_t)blockIdx.x;
        for (int k = 0; k < 8; ++k) o[k] = stamp[k];
    }
}

// ======================================================================================================================
// Weight gradient in bf16 storage, second generation.  dW[co][tap*Cp + ci] = sum over pixels of dY[pixel][co] * X[src(pixel, tap)][ci]:
// the reduction index is the PIXEL while both operands are channel-contiguous in HBM.  wgrad_bf16_kernel<HS> transposed 8x8
// half-word blocks in registers (22.8 VALU instructions per MFMA, profiles/r02_ae_sq_counters.txt).  Here both operand slices
// go HBM -> LDS as they are ([pixel][channel] rows, plain 16-byte LDS-DMA copies, hardware zero fill for padding taps and for
// pixels past the end of the split), and the MFMA fragments (8 consecutive pixels of one channel per lane) come out of
// gfx950's transposing LDS read ds_read_b64_tr_b16: a 16-lane group reads a [4 pixels][16 channels] block, lane j' supplies the
// 8-byte address of (pixel j' >> 2, channels 4 * (j' & 3) .. + 3) and receives channel j' & 15, pixels 0..3
// (lane map measured with tools/gpu/tr16_probe.hip, profiles/r03_a_tr16_lane_map.txt).
// LDS rows are BM*2 / BN*2 bytes; the 16-byte chunk c of pixel row p sits at slot c ^ swz(p) so that the four pixel rows of
// a transposing read hit four different 64-byte bank quarters (256-byte rows: swz = 4 * (p & 3); 128-byte rows: 4 * ((p >> 1) & 1)).
// A 64-entry table per slice (pixel -> byte offsets of its dY row and of filter tap (0,0) of its X row, + the tap-(0,0)
// coordinates for the bounds test) is computed by wave 0 one slice ahead, so the other waves only add per-lane constants.
// ======================================================================================================================

struct PixRec {
    uint32_t dyoff;   // byte offset of the pixel's dY row (OOB for pixels >= mend)
    int xoff;         // byte offset of filter tap (0,0) of the pixel's X row (may be negative)
    int h0, w0;       // input coordinates of filter tap (0,0); h0 = -(1 << 20) for pixels >= mend
};

#ifdef UP_EMU
__device__ __forceinline__ bf16x8 lds_read_tr16x2(const unsigned char* lds, int addr0, int addr1) {
    bf16x8 f;
    const int l = threadIdx.x & 63, j = l & 15, g = l & 48;
    for (int h = 0; h < 2; ++h)
        for (int e = 0; e < 4; ++e) {
            const int src = ::emu::shfl(h ? addr1 : addr0, g + 4 * e + (j >> 2));
            memcpy(&f.v[4 * h + e], lds + src + 2 * (j & 3), 2);
        }
    return f;
}
#else
__device__ __forceinline__ bf16x8 lds_read_tr16x2(const unsigned char* lds, int addr0, int addr1) {
    typedef short s16x4 __attribute__((ext_vector_type(4)));
    typedef short s16x8 __attribute__((ext_vector_type(8)));
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(lds + addr0));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(lds + addr1));
    const s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8, v);
}
#endif

#ifndef UP_EMU
// Transposing fragment reads of wgrad_glds_kernel as inline assembly (see the comment in the kernel): one 16-pixel step of
// fragments, the reads of a step, and "wait for this step's reads, then its MFMAs".
typedef short s16x4_t __attribute__((ext_vector_type(4)));
typedef short s16x8_t __attribute__((ext_vector_type(8)));
template <int TM, int TN>
struct TrFrags {
    s16x4_t alo[TM], ahi[TM], blo[TN], bhi[TN];
};
template <int S, int ROWA, int ROWB, int TM, int TN>
__device__ __forceinline__ void tr16_read_step(TrFrags<TM, TN>& f, uint32_t base, const uint32_t (&adA)[TM], const uint32_t (&adB)[TN]) {
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(f.alo[i]) : "v"(base + adA[i]), "n"(16 * S * ROWA));
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(f.ahi[i]) : "v"(base + adA[i]), "n"((16 * S + 4) * ROWA));
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(f.blo[j]) : "v"(base + adB[j]), "n"(16 * S * ROWB));
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(f.bhi[j]) : "v"(base + adB[j]), "n"((16 * S + 4) * ROWB));
    }
}
template <int S, int NS, int ROWA, int ROWB, int TM, int TN>
__device__ __forceinline__ void tr16_mfma_step(TrFrags<TM, TN> (&f)[2], uint32_t base, const uint32_t (&adA)[TM], const uint32_t (&adB)[TN],
                                               f32x16 (&acc)[TM][TN]) {
    constexpr int b = S & 1;
    if constexpr (S + 1 < NS) tr16_read_step<S + 1, ROWA, ROWB>(f[b ^ 1], base, adA, adB);
    constexpr int younger = S + 1 < NS ? 2 * (TM + TN) : 0;
    // the wait is tied to every fragment register of this step, so no MFMA below can be scheduled in front of it
#pragma unroll
    for (int i = 0; i < TM; ++i) asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(f[b].alo[i]), "+v"(f[b].ahi[i]) : "n"(younger));
#pragma unroll
    for (int j = 0; j < TN; ++j) asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(f[b].blo[j]), "+v"(f[b].bhi[j]) : "n"(younger));
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const bf16x8 af = __builtin_bit_cast(bf16x8, (s16x8_t)__builtin_shufflevector(f[b].alo[i], f[b].ahi[i], 0, 1, 2, 3, 4, 5, 6, 7));
            const bf16x8 bf = __builtin_bit_cast(bf16x8, (s16x8_t)__builtin_shufflevector(f[b].blo[j], f[b].bhi[j], 0, 1, 2, 3, 4, 5, 6, 7));
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bf, acc[i][j], 0, 0, 0);
        }
}
#endif

// KP: pixels per K slice (64 | 32), ST: LDS stages (2 | 3, see igemm_glds_kernel), OCC: workgroups per CU the registers allow
template <int BM, int BN, int KP, int ST>
struct WGeom {
    static constexpr int ROWA = BM * 2, ROWB_ = BN * 2;                 // bytes per pixel row of the two slices
    static constexpr int A_BYTES = KP * ROWA, B_BYTES = KP * ROWB_, STAGE = A_BYTES + B_BYTES;
    static constexpr int TAB_OFF = ST * STAGE;                           // PixRec[ST][KP]
    static constexpr int TOTAL = TAB_OFF + ST * KP * (int)sizeof(PixRec);
};

template <int BM, int BN, int KP = 64, int ST = 2, int OCC = 2>
__global__ void __launch_bounds__(256, OCC) wgrad_glds_kernel(WgradArgs a) {
    using G = WGeom<BM, BN, KP, ST>;
    static_assert(KP == 64 || KP == 32, "64 or 32 pixels per slice");
    static_assert(KP / (4 * (1024 / (BM * 2))) >= 1 && KP / (4 * (1024 / (BN * 2))) >= 1, "a wave issues at least one LDS-DMA instruction per operand");
    constexpr int TM = BM / 64, TN = BN / 64;
    constexpr int RPA = 1024 / G::ROWA, RPB = 1024 / G::ROWB_;    // pixel rows per LDS-DMA instruction (4 or 8)
    constexpr int NIA = KP / (4 * RPA), NIB = KP / (4 * RPB);     // instructions per wave, slice and operand (4 or 2)
    constexpr int CHA = G::ROWA / 16, CHB = G::ROWB_ / 16;         // 16-byte chunks per row (16 or 8)
    __shared__ __attribute__((aligned(16))) unsigned char smem[G::TOTAL];
    PixRec* const tab = reinterpret_cast<PixRec*>(smem + G::TAB_OFF);

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = uniform(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31;

    const int logical = xcd_remap(blockIdx.x, a.nwg);
    const int split = fdiv(logical, a.fTiles);
    const int tile = logical - split * (int)a.fTiles.d;
    const int mt = fdiv(tile, a.fNtn);
    const int nt = tile - mt * a.ntn;
    const int co0 = mt * BM, col0 = nt * BN;

    // reduction domain of this workgroup: pixels [mbeg, mend) of an (images x rows x cols) box at (r_pl, r_ql)
    int r_pl = 0, r_ql = 0, r_h = a.P, r_w = a.Q;
    FastDiv r_fhw = a.fPQ, r_fw = a.fQ;
    int mbeg = split * a.rows_per_split;
    int mend = min(a.M, mbeg + a.rows_per_split);
    if (a.rect) {
        const int* rc = a.rect + WGRAD_RECT_INTS * nt;
        r_pl = rc[0];
        r_ql = rc[1];
        r_h = rc[2];
        r_w = rc[3];
        r_fhw = FastDiv{(uint32_t)rc[4], (uint32_t)rc[5], (uint32_t)rc[6]};
        r_fw = FastDiv{(uint32_t)rc[7], (uint32_t)rc[8], (uint32_t)rc[9]};
        mbeg = split * rc[10];
        mend = min(rc[11], mbeg + rc[10]);
    }
    const int r_hw = r_h * r_w;
    const int nsl = mbeg < mend ? (mend - mbeg + KP - 1) / KP : 0;

    const Rsrc rsA = make_rsrc(a.dy, a.dy_bytes);
    const Rsrc rsB = make_rsrc(a.x, a.x_bytes);

    // per-lane constants of the LDS-DMA: pixel row inside the instruction's group and 16-byte slot
    const int prowA = lane / CHA, slotA = lane % CHA;
    const int prowB = lane / CHB, slotB = lane % CHB;
    const int swzA = CHA == 16 ? 4 * (prowA & 3) : 4 * ((prowA >> 1) & 1);
    const int swzB = CHB == 16 ? 4 * (prowB & 3) : 4 * ((prowB >> 1) & 1);
    // A: channels co0 + 8 * (slot ^ swz) .. + 7 of the dY row (channels beyond the row only feed rows of dW that are never stored)
    const uint32_t coffA = (uint32_t)(co0 + 8 * (slotA ^ swzA)) * 2u;
    // B: GEMM columns col0 + 8 * (slot ^ swz) .. + 7 = one filter tap, 8 input channels
    const int colB = col0 + 8 * (slotB ^ swzB);
    const bool colokB = colB < a.Ncols;
    const int tapB = fdiv(colokB ? colB : 0, a.fCp);
    const int ciB = (colokB ? colB : 0) - tapB * a.Cp;
    const int rB = fdiv(tapB, a.fS);
    const int dhB = rB * a.dil, dwB = (tapB - rB * a.S) * a.dil;
    const int deltaB = (dhB * a.W + dwB) * a.ldx * 2 + ciB * 2;
    // columns past the end fail the row test through a huge row offset (with `colokB && ...` in the per-slice predicate the compiler
    // split every B instruction into two exec-masked halves: the lanes with and without a real column)
    const int dhT = colokB ? dhB : (1 << 24);

    auto fill_table = [&](int sl, int slot) {   // wave 0: one lane per pixel of slice sl
        if (KP < 64 && lane >= KP) return;
        const int m = mbeg + sl * KP + lane;
        const bool ok = m < mend;
        const int mm = ok ? m : mbeg;
        const int img = fdiv(mm, r_fhw);
        const int rem = mm - img * r_hw;
        const int pi = fdiv(rem, r_fw);
        const int qi = rem - pi * r_w;
        const int p = r_pl + pi, q = r_ql + qi;
        PixRec rec;
        rec.dyoff = ok ? (uint32_t)(((img * a.P + p) * a.Q + q) * a.ldy) * 2u : OOB;
        rec.h0 = ok ? p * a.stride - a.pad : -(1 << 20);
        rec.w0 = q * a.stride - a.pad;
        rec.xoff = ok ? ((img * a.H + rec.h0) * a.W + rec.w0) * a.ldx * 2 : 0;
        tab[slot * KP + lane] = rec;
    };
    auto issue = [&](int slot) {   // the slice whose table sits in `slot`, into LDS stage `slot`
        unsigned char* const As = smem + slot * G::STAGE;
        unsigned char* const Bs = As + G::A_BYTES;
        const PixRec* const t = tab + slot * KP;
#pragma unroll
        for (int i = 0; i < NIA; ++i) {
            const int grp = wave + 4 * i;
            const PixRec r = t[grp * RPA + prowA];
            load16_to_lds(rsA, r.dyoff == OOB ? OOB : r.dyoff + coffA, As + grp * 1024);
        }
#pragma unroll
        for (int i = 0; i < NIB; ++i) {
            const int grp = wave + 4 * i;
            const PixRec r = t[grp * RPB + prowB];
            const int h = r.h0 + dhT, w = r.w0 + dwB;
            const bool ok = (unsigned)h < (unsigned)a.H && (unsigned)w < (unsigned)a.W;
            load16_to_lds(rsB, ok ? (uint32_t)(r.xoff + deltaB) : OOB, Bs + grp * 1024);
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // transposing fragment reads: lane -> (pixel within the 16-pixel step, channel) of the [4 pixels][16 channels] block
    const int jq = lane & 15, grp16 = lane >> 4;
    const int fpix = 8 * (grp16 >> 1) + (jq >> 2);           // + 4 * h + 16 * s
    const int fch = 16 * (grp16 & 1) + 4 * (jq & 3);         // channel inside the 32-wide MFMA tile

#ifndef UP_EMU
    // Fragment reads as INLINE ASSEMBLY with hand-counted lgkmcnt waits (round 4).  Through the builtin, the compiler's wait-count
    // pass put an `s_waitcnt vmcnt(0)` in front of the first transposing read of every slice: it sees an LDS access while LDS-DMA
    // writes are in flight — the NEXT slice, just issued into the other stage — and cannot tell the stages apart (the plain vector
    // loads of igemm_glds_kernel escape that through their type-based alias info; an intrinsic call carries none).  The kernel
    // therefore waited for slice it + 1 to land before computing slice it: no load / compute overlap inside a workgroup
    // (profiles/r03_z_sq_counters: waves waiting on vmcnt, MFMA pipe 22.5 % busy).  The assembler reads are invisible to that
    // pass; ordering: the reads of 16-pixel step s + 1 are issued before the MFMAs of step s, which wait until only those
    // 2 (TM + TN) younger reads are outstanding (LDS operations return in order).
    const int swA0 = CHA == 16 ? 4 * (fpix & 3) : 4 * ((fpix >> 1) & 1);   // (16 s + 4 h is a multiple of 4: the swizzle term is the lane's)
    const int swB0 = CHB == 16 ? 4 * (fpix & 3) : 4 * ((fpix >> 1) & 1);
    uint32_t adA[TM], adB[TN];   // byte offset inside a stage of this lane's (s = 0, h = 0) read of every 32-channel block
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int ch = wm * (BM / 2) + i * 32 + fch;
        adA[i] = (uint32_t)(fpix * G::ROWA + (((ch >> 3) ^ swA0) << 4) + (ch & 7) * 2);
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int ch = wn * (BN / 2) + j * 32 + fch;
        adB[j] = (uint32_t)(G::A_BYTES + fpix * G::ROWB_ + (((ch >> 3) ^ swB0) << 4) + (ch & 7) * 2);
    }
    const uint32_t lds0 = (uint32_t)(size_t)((__attribute__((address_space(3))) unsigned char*)smem);
    auto mfmas = [&](const unsigned char* As) {
        constexpr int NS = KP / 16;
        static_assert(NS <= 4, "at most four 16-pixel steps per slice");
        const uint32_t base = lds0 + (uint32_t)(As - smem);
        TrFrags<TM, TN> f[2];
        tr16_read_step<0, G::ROWA, G::ROWB_>(f[0], base, adA, adB);
        tr16_mfma_step<0, NS, G::ROWA, G::ROWB_>(f, base, adA, adB, acc);
        if constexpr (NS > 1) tr16_mfma_step<1, NS, G::ROWA, G::ROWB_>(f, base, adA, adB, acc);
        if constexpr (NS > 2) tr16_mfma_step<2, NS, G::ROWA, G::ROWB_>(f, base, adA, adB, acc);
        if constexpr (NS > 3) tr16_mfma_step<3, NS, G::ROWA, G::ROWB_>(f, base, adA, adB, acc);
    };
#else
    auto mfmas = [&](const unsigned char* As) {
        const unsigned char* Bs = As + G::A_BYTES;
#pragma unroll
        for (int s = 0; s < KP / 16; ++s) {
            bf16x8 af[TM], bf[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int ch = wm * (BM / 2) + i * 32 + fch;            // channel inside the tile (multiple of 4)
                int ad[2];
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int p = 16 * s + 4 * h + fpix;
                    const int sw = CHA == 16 ? 4 * (p & 3) : 4 * ((p >> 1) & 1);
                    ad[h] = p * G::ROWA + (((ch >> 3) ^ sw) << 4) + (ch & 7) * 2;
                }
                af[i] = lds_read_tr16x2(As, ad[0], ad[1]);
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int ch = wn * (BN / 2) + j * 32 + fch;
                int ad[2];
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int p = 16 * s + 4 * h + fpix;
                    const int sw = CHB == 16 ? 4 * (p & 3) : 4 * ((p >> 1) & 1);
                    ad[h] = p * G::ROWB_ + (((ch >> 3) ^ sw) << 4) + (ch & 7) * 2;
                }
                bf[j] = lds_read_tr16x2(Bs, ad[0], ad[1]);
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bf[j], acc[i][j], 0, 0, 0);
        }
    };
#endif
    if (nsl > 0) {
        static_assert(ST == 2 && KP == 64, "64-pixel slices, two LDS stages (32 pixels / three stages measured slower and left in round 4)");
        {
            if (wave == 0) fill_table(0, 0);
            __syncthreads();
            issue(0);
            if (wave == 0 && nsl > 1) fill_table(1, 1);
            for (int it = 0; it < nsl; ++it) {
                wait_dma();
                __syncthreads();   // slice `it` has landed, table it+1 is written, everyone is done with the other stage
                if (it + 1 < nsl) issue((it + 1) & 1);
                if (wave == 0 && it + 2 < nsl) fill_table(it + 2, it & 1);
                mfmas(smem + (it & 1) * G::STAGE);
            }
        }
    }

    float* out = a.slab + (size_t)split * a.K * a.Ncols;
    const int lh = lane >> 5;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int col = col0 + wn * (BN / 2) + j * 32 + l31;
        if (col >= a.Ncols) continue;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = co0 + wm * (BM / 2) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (co < a.K) out[(size_t)co * a.Ncols + col] = acc[i][j][r];
            }
    }
}

}  // namespace glds
