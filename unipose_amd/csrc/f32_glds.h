// Exact-fp32 implicit GEMM with operands HBM -> LDS by LDS-DMA (round 4): the forward / data-gradient kernel of the headline
// configuration (BASELINE configs[1]: fp32, 368x368, B = 32) rebuilt on the operand path of bf16s_glds.h.
// Included by conv_igemm.hip inside namespace up, after bf16s_glds.h (Rsrc, load16_to_lds, wait_dma, OOB, uniform).
//
// What differs from igemm_kernel<..., MODE 2> (register-staged: global_load -> VGPR -> zero select -> ds_write_b128):
//  * a K slice of 32 floats per row goes HBM -> LDS with `buffer_load_dwordx4 ... lds`: no staging registers (32 VGPRs of the
//    64x64 two-set loop), no ds_write pass, no per-element zero selects — padding taps and rows >= N are zero-filled by the
//    buffer descriptor's bounds check.  The LDS image is the one igemm_kernel<..., SWZ> reads (128-byte rows, 16-byte chunk c of
//    row r at slot c ^ ((r >> 1) & 7)), so the fragment reads and the MFMA order — hence every result bit — are unchanged.
//  * the epilogue (EPI = 1) leaves through LDS: the accumulator tile is written as an fp32 image (row stride BN + 8 floats: the two
//    half-waves of a ds_write_b32 land on disjoint banks) and read back as float4 rows, so a thread issues BM*BN/1024 16-byte
//    stores (4 on the 64x64 tile) instead of 16 / 64 dword stores, and a residual / addend is read with 16-byte loads.
//  * BNRED: the data-gradient launch that produces dz — the gradient w.r.t. z = relu(bn(y) (+ res)) of the PREVIOUS layer — also
//    reduces that layer's two BatchNorm-backward sums per row tile (sum g, sum g * (y - mean), g = dz * [z > 0]) on the float4 rows
//    it is about to store: the producing layer's backward loses its reduction pass (up_bn_bwd_prereduced_t).  The operands of the
//    reduction (y, sign bits, addend) are fetched with 16-byte loads issued BEFORE the last K slice's MFMAs (64x64 tiles: 4 units
//    per thread), so their latency hides behind MFMA work instead of sitting in the epilogue (the round-2 experiment lost 5.6 ms
//    per step to 2-3 dependent dword loads per output element there, tools/experiments/README.md in the history).
#pragma once

namespace glds {

#ifdef UP_EMU
typedef float f32x4 __attribute__((vector_size(16)));
typedef int i32x4 __attribute__((vector_size(16)));
#else
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
#endif

template <int BM, int BN, int ST, bool BREG = false>
struct GeomF {
    static constexpr int ROWB = 128, CH = 8, RPI = 8;                   // bytes per row slice, 16-byte chunks per row, rows per LDS-DMA instruction
    // BREG: the weight operand never enters LDS (its fragments go global -> registers), a stage holds the activation slice only
    static constexpr int A_BYTES = BM * ROWB, B_BYTES = BREG ? 0 : BN * ROWB, STAGE = A_BYTES + B_BYTES;
    static constexpr int IMG_LD = BN + 8;                               // floats: rows 4 apart land 32 banks apart
    static constexpr int IMG_BYTES = BM * IMG_LD * 4;
    static constexpr int XCH_BYTES = BN * 12;                           // BatchNorm statistics exchange between the two M-waves
    static constexpr int RED_BYTES = 4 * BN * 8;                        // BNRED: per-wave column sums
    static constexpr int EPI_BYTES = IMG_BYTES + XCH_BYTES + RED_BYTES;
    static constexpr int MAIN = ST * STAGE > EPI_BYTES ? ST * STAGE : EPI_BYTES;
    static constexpr int MASK_OFF = MAIN;                               // tap masks of the four waves
    static constexpr int TOTAL = MASK_OFF + 16;
    __host__ __device__ static constexpr int swz(int r) { return (r >> 1) & 7; }
};

// LDS-transposed epilogue: see the header comment.  `smem` must be free (every wave past its last fragment read).
template <int BM, int BN, bool PERM, int ST, bool BNRED, int UNITS_PF, bool BREG = false>
struct Epi32 {
    using G = GeomF<BM, BN, ST, BREG>;
    static constexpr int TM = BM / 64, TN = BN / 64;
    static constexpr int CQ = BN / 4;                    // float4 chunks per tile row
    static constexpr int UNITS = BM * CQ / 256;          // (row, chunk) units per thread
    static constexpr int RSTEP = 256 / CQ;               // rows between two units of a thread
    static constexpr bool PF = UNITS_PF > 0;             // operands prefetched before the last K slice
    static_assert(!PF || UNITS_PF == UNITS, "prefetch holds every unit of the thread");
    static_assert(!BNRED || BN <= 128, "BNRED: one thread per (column, sum) finishes the tile, 2 * BN <= 256 threads");

    float4 res[PF ? UNITS : 1], yb[PF ? UNITS : 1], pmu;
    uint32_t bw[PF ? UNITS : 1], rw[PF ? UNITS : 1];   // sign-bit nibbles: of the layer being reduced / of the layer the addend belongs to
    float pis;           // 1 / std of the column this thread finishes (BNRED)
    int pix[UNITS];      // destination pixel of each unit (-1: row past the end)

    int m_end;           // first row past this tile's group (= a.M without row groups)
    int gmean;           // offset of the tile's group in bn_mean / bn_invstd

    __device__ __forceinline__ void rows(const IgemmArgs& a, int m0, int tid, int m_end_, int gbase, int grp) {
        m_end = m_end_;
        gmean = grp * a.bn_grp_stride;
#pragma unroll
        for (int k = 0; k < UNITS; ++k) {
            const int m = m0 + tid / CQ + k * RSTEP;
            int p = -1;
            if (m < m_end) p = PERM ? gbase + a.perm[m - gbase] : m;
            pix[k] = p;
        }
    }
    // bit mask of the 4 channels n..n+3 of pixel p in the ReLU sign bits of the layer being reduced
    static __device__ __forceinline__ uint32_t bits_word(const uint32_t* bits, int C, int p, int n) {
        const long long quad = ((long long)p * C + n) >> 2;
        return bits[quad >> 3] >> (4 * (int)(quad & 7));
    }
    __device__ __forceinline__ void prefetch(const IgemmArgs& a, int n0, int tid) {
        if constexpr (PF) {
            const int n = n0 + (tid % CQ) * 4;
            const bool nok = n < a.Ng;
            const int nn = nok ? n : 0;
            if (a.residual) {
#pragma unroll
                for (int k = 0; k < UNITS; ++k)
                    res[k] = *reinterpret_cast<const float4*>(a.residual + (size_t)(pix[k] >= 0 ? pix[k] : 0) * a.ldr + nn);
                if (a.res_bits) {   // the addend is dz * [z > 0] of another layer, masked here instead of materialised there
#pragma unroll
                    for (int k = 0; k < UNITS; ++k) rw[k] = bits_word(a.res_bits, a.Ng, pix[k] >= 0 ? pix[k] : 0, nn);
                } else {
#pragma unroll
                    for (int k = 0; k < UNITS; ++k) rw[k] = 0xfu;
                }
            }
            if constexpr (BNRED) {
                pmu = *reinterpret_cast<const float4*>(a.bn_mean + gmean + nn);
                const int cc = n0 + (tid >> 1);
                pis = a.bn_invstd[gmean + (tid < 2 * BN && cc < a.Ng ? cc : 0)];
#pragma unroll
                for (int k = 0; k < UNITS; ++k)
                    yb[k] = *reinterpret_cast<const float4*>(a.bn_y + (size_t)(pix[k] >= 0 ? pix[k] : 0) * a.bn_ld + nn);
                if (a.bn_bits) {
#pragma unroll
                    for (int k = 0; k < UNITS; ++k) bw[k] = bits_word(a.bn_bits, a.bn_C, pix[k] >= 0 ? pix[k] : 0, nn);
                } else {
#pragma unroll
                    for (int k = 0; k < UNITS; ++k) bw[k] = 0xfu;
                }
            }
        }
    }

    __device__ __forceinline__ void run(const IgemmArgs& a, f32x16 (&acc)[BM / 64][BN / 64], unsigned char* smem, int mt, int m0, int n0,
                                        int tid, int wm, int wn, int l31, int lh) {
        float* const img = reinterpret_cast<float*>(smem);
        float* const xch = reinterpret_cast<float*>(smem + G::IMG_BYTES);
        float* const red = reinterpret_cast<float*>(smem + G::IMG_BYTES + G::XCH_BYTES);
        const int mrow0 = m0 + wm * (BM / 2) + 4 * lh;
        const int ncol0 = n0 + wn * (BN / 2) + l31;
        // BatchNorm-forward partials (count, mean, M2) of the raw accumulators, exactly as igemm_epilogue computes them
        float sc[TN], sm[TN], s2[TN];
        if (a.stats) {
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                float cnt = 0.f, sum = 0.f;
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int m = mrow0 + i * 32 + (r & 3) + 8 * (r >> 2);
                        if (m < m_end) {
                            cnt += 1.f;
                            sum += acc[i][j][r];
                        }
                    }
                float mean = cnt > 0.f ? sum / cnt : 0.f;
                float q = 0.f;
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int m = mrow0 + i * 32 + (r & 3) + 8 * (r >> 2);
                        if (m < m_end) {
                            const float d = acc[i][j][r] - mean;
                            q += d * d;
                        }
                    }
                const float c2 = __shfl_xor(cnt, 32), m2 = __shfl_xor(mean, 32), q2 = __shfl_xor(q, 32);
                if (lh) {   // both halves merge in the same order to agree bitwise
                    float tc = c2, tm = m2, tq = q2;
                    wf_merge(tc, tm, tq, cnt, mean, q);
                    cnt = tc;
                    mean = tm;
                    q = tq;
                } else {
                    wf_merge(cnt, mean, q, c2, m2, q2);
                }
                sc[j] = cnt;
                sm[j] = mean;
                s2[j] = q;
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
        }
        // the tile as an fp32 image (folded scale / shift / bias applied per column on the way)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = ncol0 + j * 32;
            const int nn = n < a.Ng ? n : a.Ng - 1;
            const float esc = a.scale ? a.scale[nn] : 1.f;
            float esh = a.scale ? a.shift[nn] : 0.f;
            if (a.bias) esh += a.bias[nn];
            const bool affine = a.scale != nullptr || a.bias != nullptr;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = wm * (BM / 2) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    const float v = acc[i][j][r];
                    img[row * G::IMG_LD + wn * (BN / 2) + j * 32 + l31] = affine ? v * esc + esh : v;
                }
        }
        __syncthreads();
        if (a.stats && wm == 0 && lh == 0) {
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const float* s = xch + ((wn * TN + j) * 32 + l31) * 3;
                wf_merge(sc[j], sm[j], s2[j], s[0], s[1], s[2]);
                const int n = ncol0 + j * 32;
                if (n < a.Ng) {
                    float* o = a.stats + ((size_t)mt * a.Ng + n) * 3;   // (sc1: the last arriver of the fold reads them)
                    st_agent(o, sc[j]);
                    st_agent(o + 1, sm[j]);
                    st_agent(o + 2, s2[j]);
                }
            }
        }
        // read-out: float4 rows; residual / addend, ReLU, the BatchNorm-backward sums of the layer whose dz this is, store.
        // Chunks of up to four units: all operand loads of a chunk first (none when they were prefetched), then the arithmetic,
        // then the stores back to back — with loads, waits and stores interleaved per unit the compiler's wait-count state made
        // every unit wait for the previous unit's store.
        const int cq = tid % CQ, row0 = tid / CQ;
        const int n = n0 + cq * 4;
        const bool nok = n < a.Ng;
        const int nn = nok ? n : 0;
        const bool relu = a.relu != 0;
        float s1a[4] = {0.f, 0.f, 0.f, 0.f}, s2a[4] = {0.f, 0.f, 0.f, 0.f};
        float4 mu = make_float4(0.f, 0.f, 0.f, 0.f);
        if constexpr (BNRED) mu = PF ? pmu : *reinterpret_cast<const float4*>(a.bn_mean + gmean + nn);
        constexpr int CHK = UNITS < 4 ? UNITS : 4;
#pragma unroll
        for (int c0 = 0; c0 < UNITS; c0 += CHK) {
            float4 v[CHK], rr[CHK], yy[CHK];
            uint32_t w[CHK], rm[CHK];
#pragma unroll
            for (int u = 0; u < CHK; ++u) {
                const int k = c0 + u;
                const int p = pix[k] >= 0 ? pix[k] : 0;
                if constexpr (PF) {
                    rr[u] = res[k];
                    rm[u] = rw[k];
                    yy[u] = yb[k];
                    w[u] = bw[k];
                } else {
                    if (a.residual) {
                        rr[u] = *reinterpret_cast<const float4*>(a.residual + (size_t)p * a.ldr + nn);
                        rm[u] = a.res_bits ? bits_word(a.res_bits, a.Ng, p, nn) : 0xfu;
                    }
                    if constexpr (BNRED) {
                        yy[u] = *reinterpret_cast<const float4*>(a.bn_y + (size_t)p * a.bn_ld + nn);
                        w[u] = a.bn_bits ? bits_word(a.bn_bits, a.bn_C, p, nn) : 0xfu;
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < CHK; ++u) v[u] = *reinterpret_cast<const float4*>(img + (row0 + (c0 + u) * RSTEP) * G::IMG_LD + cq * 4);
#pragma unroll
            for (int u = 0; u < CHK; ++u) {
                if (a.residual) {
                    v[u].x += (rm[u] & 1u) ? rr[u].x : 0.f;
                    v[u].y += (rm[u] & 2u) ? rr[u].y : 0.f;
                    v[u].z += (rm[u] & 4u) ? rr[u].z : 0.f;
                    v[u].w += (rm[u] & 8u) ? rr[u].w : 0.f;
                }
                if (relu) {
                    v[u].x = fmaxf(v[u].x, 0.f);
                    v[u].y = fmaxf(v[u].y, 0.f);
                    v[u].z = fmaxf(v[u].z, 0.f);
                    v[u].w = fmaxf(v[u].w, 0.f);
                }
                if constexpr (BNRED) {
                    const bool live = pix[c0 + u] >= 0 && nok;     // rows past the end / columns past N contribute nothing
                    const uint32_t wk = live ? w[u] : 0u;
                    const float g0 = (wk & 1u) ? v[u].x : 0.f, g1 = (wk & 2u) ? v[u].y : 0.f, g2 = (wk & 4u) ? v[u].z : 0.f,
                                g3 = (wk & 8u) ? v[u].w : 0.f;
                    s1a[0] += g0;
                    s1a[1] += g1;
                    s1a[2] += g2;
                    s1a[3] += g3;
                    s2a[0] += g0 * (yy[u].x - mu.x);
                    s2a[1] += g1 * (yy[u].y - mu.y);
                    s2a[2] += g2 * (yy[u].z - mu.z);
                    s2a[3] += g3 * (yy[u].w - mu.w);
                }
            }
#pragma unroll
            for (int u = 0; u < CHK; ++u)
                if (pix[c0 + u] >= 0 && nok) *reinterpret_cast<float4*>(a.y + (size_t)pix[c0 + u] * a.ldy + n) = v[u];
        }
        if constexpr (BNRED) {
            // lanes that share a chunk column are CQ apart; then the four waves through LDS, summed in a fixed order
#pragma unroll
            for (int off = CQ; off < 64; off <<= 1)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    s1a[e] += __shfl_xor(s1a[e], off);
                    s2a[e] += __shfl_xor(s2a[e], off);
                }
            const int wave = tid >> 6, lane = tid & 63;
            if (lane < CQ) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    red[(wave * BN + cq * 4 + e) * 2] = s1a[e];
                    red[(wave * BN + cq * 4 + e) * 2 + 1] = s2a[e];
                }
            }
            __syncthreads();
            if (tid < 2 * BN) {
                const int ch = tid >> 1, which = tid & 1;
                float t = 0.f;   // every wave covers other rows of the tile, whatever the chunk count per row: always all four
#pragma unroll
                for (int w = 0; w < 4; ++w) t += red[(w * BN + ch) * 2 + which];
                const int cc = n0 + ch;
                if (cc < a.Ng) {
                    if (which) t *= PF ? pis : a.bn_invstd[gmean + cc];
                    st_agent(a.bn_partial + ((size_t)mt * a.Ng + cc) * 2 + which, t);
                }
            }
        }
    }
};

// PERM: GEMM row m is output pixel a.perm[m] (tap-sorted order).   ST: LDS stages (2: one barrier per slice, the next slice in
// flight during the MFMAs; 1: single stage, 16 KB per 64x64 workgroup — the co-resident workgroups hide the load).
// EPI: 0 = igemm_epilogue (dword stores from the accumulator layout), 1 = Epi32 (LDS-transposed, 16-byte stores).
// BNRED (EPI 1): BatchNorm-backward sums of the producing layer in the epilogue (a.bn_*).
// WIDE: filters of more than 32 taps (the video head's 11x11, uniposeLSTM.py:43-45).  The per-row validity is kept separably —
//       bit r: filter row r reads a real pixel row, bit 16 + s: filter column s a real column (R, S <= 16) — instead of one bit per
//       tap, every tap is visited (no tile-level skipping: with pad = 5 on 46x46 maps nearly every tap is live for some row of a
//       tile), no tap-sorted rows.  Same slice order (tap-major, 32 channels per slice) as the register-staged per-slice-tap path.
// BREG (round 6, "hybrid" operand path): the B (weight) fragments go global -> registers, one slice ahead of their MFMAs, and
//       only the A (activation) slice goes through LDS-DMA: half the LDS-DMA writes and half the fragment reads per MFMA — the two
//       that collide in the LDS (profiles/r05_z_mix_probe.txt: MFMA + both 126 TF, hybrid 130).  Same values, same k order: bit-
//       identical results.  Weight rows >= N are clamped to the last row (their columns are never stored).  Two LDS stages only.
template <int BM, int BN, bool PERM, int ST = 2, int OCC = 4, int EPI = 1, bool BNRED = false, bool WIDE = false, bool BREG = false>
__global__ void __launch_bounds__(256, OCC) igemm_glds32_kernel(IgemmArgs a) {
    using G = GeomF<BM, BN, ST, BREG>;
    static_assert(!BREG || ST == 2, "the register operand is double-buffered with the two LDS stages");
    static_assert(ST == 1 || ST == 2, "one or two LDS stages");
    static_assert(!BNRED || EPI == 1, "the fused reduction lives in the LDS-transposed epilogue");
    static_assert(!WIDE || (!PERM && !BNRED), "the > 32-tap form has no tap-sorted rows and no fused reduction");
    constexpr int TM = BM / 64, TN = BN / 64;
    constexpr int NA = BM / (4 * G::RPI), NB = BN / (4 * G::RPI);   // LDS-DMA instructions per wave, slice and operand
    constexpr int UNITS = BM * BN / 1024;
    using E = Epi32<BM, BN, PERM, ST, BNRED, (UNITS <= 4 ? UNITS : 0), BREG>;
    __shared__ __attribute__((aligned(16))) unsigned char smem[G::TOTAL];
    unsigned* const wmask = reinterpret_cast<unsigned*>(smem + G::MASK_OFF);

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = uniform(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, lh = lane >> 5;

    // K-split tail tiles exactly as in igemm_kernel ("tail split"): blocks >= full_blocks reduce a 1 / parts share of the live
    // slices of tile full_blocks + tail; the last part of a tile adds the published shares in a fixed order and runs the epilogue
    int logical, part = 0, tail = 0;
    const bool split = (int)blockIdx.x >= a.full_blocks;
    if (!split) {
        logical = xcd_remap(blockIdx.x, a.full_blocks);
    } else {
        const int j = (int)blockIdx.x - a.full_blocks;
        tail = uniform(j / a.parts);
        part = j - tail * a.parts;
        logical = a.full_blocks + tail;
    }
    const int mt = fdiv(logical, a.fNtn);
    const int nt = logical - mt * a.ntn;
    int m0 = mt * BM, m_end = a.M, gbase = 0, grp = 0;
    const int n0 = nt * BN;
    if (a.grp_rows) {   // row groups (see IgemmArgs): tile mt % grp_tiles of group mt / grp_tiles
        grp = fdiv(mt, a.fGrpTiles);
        gbase = grp * a.grp_rows;
        m0 = gbase + (mt - grp * a.grp_tiles) * BM;
        m_end = gbase + a.grp_rows;
    }

    // this lane's rows of the operand tiles: row (wave + 4 i) * RPI + lane / CH, 16-byte slot lane % CH
    const int rsub = lane / G::CH, slot = lane % G::CH;
    const int R = a.taps / a.S;
    const Rsrc rsA = make_rsrc(a.x, a.x_bytes);
    const Rsrc rsB = make_rsrc(a.w, (uint32_t)a.Ng * (uint32_t)a.Ktot * 4u);
    uint32_t woffB[NB];
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        const int row = (wave + 4 * j) * G::RPI + rsub;
        const int n = n0 + row;
        woffB[j] = n < a.Ng ? (uint32_t)n * (uint32_t)a.Ktot * 4u + (uint32_t)((slot ^ G::swz(row)) << 4) : OOB;
    }
    // BREG: this lane's B rows (column n = n0 + wn * BN/2 + 32 j + l31, clamped) as element offsets into the weight image, and the
    // two register sets of B fragments: breg[set][k-group][j] = B[n][slice k0 + (2 g + lh) * 4 .. + 3]
    int browB[TN];
    f32x4 breg[2][4][TN];
    if constexpr (BREG) {
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = n0 + wn * (BN / 2) + j * 32 + l31;
            browB[j] = (n < a.Ng ? n : a.Ng - 1) * a.Ktot + lh * 4;
        }
    }
    auto loadB = [&](auto set_tag, int tap_, int cs_) {
        constexpr int SET = decltype(set_tag)::value;
        const float* const wk = a.w + tap_ * a.Cp + cs_ * 32;
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int j = 0; j < TN; ++j) breg[SET][g][j] = *reinterpret_cast<const f32x4*>(wk + browB[j] + g * 8);
    };
    auto issueB = [&](int stage, int tap_, int cs_) {
        if constexpr (BREG) return;
        unsigned char* const Bs = smem + stage * G::STAGE + G::A_BYTES;
        const uint32_t kb = (uint32_t)(tap_ * a.Cp + cs_ * 32) * 4u;
#pragma unroll
        for (int j = 0; j < NB; ++j)
            load16_to_lds(rsB, woffB[j] == OOB ? OOB : woffB[j] + kb, Bs + (wave + 4 * j) * 1024);
    };
    // 1x1, stride 1, no padding: the source pixel IS the destination pixel, and the first weight slice depends on no row set-up
    const bool pointwise = a.taps == 1 && a.mul == 1 && a.off0 == 0 && a.off0w == 0 && a.H == a.P && a.W == a.Q;
    const bool early_b = pointwise && !split && !BREG;
    if (early_b) issueB(0, 0, 0);

    int roffA[NA];        // byte offset of (filter tap (0,0), this lane's chunk) of the row in the activation tensor
    unsigned tmA[NA];     // bit t: tap t of the row reads a real pixel (0 for rows >= M)
    unsigned tile_taps = 0;
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        const int row = (wave + 4 * i) * G::RPI + rsub;
        const int m = m0 + row;
        int pix = m < m_end ? m : m_end - 1;
        if constexpr (PERM) pix = gbase + a.perm[pix - gbase];
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
            unsigned hm = 0, wmk = 0;   // separable test: R + S comparisons instead of R * S
            for (int r = 0; r < R; ++r) hm |= ((unsigned)(hb + r * a.tapstep) < (unsigned)a.H) ? (1u << r) : 0u;
            for (int s = 0; s < a.S; ++s) wmk |= ((unsigned)(wb + s * a.tapstep) < (unsigned)a.W) ? (1u << s) : 0u;
            mk = 0u;
            if constexpr (WIDE) {
                mk = hm | (wmk << 16);
            } else {
                for (int r = 0; r < R; ++r) mk |= ((hm >> r) & 1u) ? (wmk << (r * a.S)) : 0u;
            }
        }
        if (m >= m_end) mk = 0u;
        roffA[i] = src * a.ldx * 4 + ((slot ^ G::swz(row)) << 4);
        tmA[i] = mk;
        tile_taps |= mk;
    }
    // the K loop visits the slices of the filter taps that are live for at least one row of the tile
    const unsigned all_taps = a.taps >= 32 ? 0xffffffffu : ((1u << a.taps) - 1u);
    unsigned live = all_taps;
    if constexpr (!WIDE) {
        if (a.taps > 1 && !a.no_tap_skip) {
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) tile_taps |= __shfl_xor(tile_taps, off);
            if (lane == 0) wmask[wave] = tile_taps;
            __syncthreads();
            live = (unsigned)uniform((int)(wmask[0] | wmask[1] | wmask[2] | wmask[3]));
            if (live == 0u) live = all_taps;
        }
    }
    const int spt = a.Cp / 32;
    int nsl = (WIDE ? a.taps : __builtin_popcount(live)) * spt;
    unsigned rest = live;
    int tap = WIDE ? 0 : __builtin_ctz(rest), cs = 0;
    if (split) {   // slices [kb, ke) of the tile's live slices
        const int kb = (int)((long long)nsl * part / a.parts), ke = (int)((long long)nsl * (part + 1) / a.parts);
        const int skip = kb / spt;
        if constexpr (WIDE) {
            tap = skip;
        } else {
            for (int t = 0; t < skip; ++t) rest &= rest - 1u;
            tap = rest ? __builtin_ctz(rest) : 0;
        }
        cs = kb - skip * spt;
        nsl = ke - kb;
    }

    bool b_issued = early_b;   // the weight slice of the first issue is already on its way
    auto issue = [&](int stage, auto set_tag) {      // set_tag: BREG's register set of the slice being issued (= its LDS stage)
        unsigned char* const As = smem + stage * G::STAGE;
        const int r = fdiv(tap, a.fS);
        const int sx = tap - r * a.S;
        const int delta = ((r * a.tapstep) * a.W + sx * a.tapstep) * a.ldx * 4 + cs * G::ROWB;
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const bool ok = WIDE ? (((tmA[i] >> r) & (tmA[i] >> (16 + sx))) & 1u) != 0u : ((tmA[i] >> tap) & 1u) != 0u;
            load16_to_lds(rsA, ok ? (uint32_t)(roffA[i] + delta) : OOB, As + (wave + 4 * i) * 1024);
        }
        if constexpr (BREG) loadB(set_tag, tap, cs);
        if (!b_issued) issueB(stage, tap, cs);
        b_issued = false;
        if (++cs == spt) {   // next live tap
            cs = 0;
            if constexpr (WIDE) {
                ++tap;
            } else {
                rest &= rest - 1u;
                tap = rest ? __builtin_ctz(rest) : 0;
            }
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // fragment of k-group g: logical chunk 2g + lh of row (32-multiple + l31) — the lane map and MFMA order of igemm_kernel
    const int swz = G::swz(l31) << 4;
    const int a_rd = (wm * (BM / 2) + l31) * G::ROWB;
    const int b_rd = G::A_BYTES + (wn * (BN / 2) + l31) * G::ROWB;
    auto mfmas = [&](const unsigned char* base, auto set_tag) {
        constexpr int SET = decltype(set_tag)::value;
        // (a plain vector type, not HIP's float4 struct: with the struct's loads the compiler put an `s_waitcnt vmcnt(0)` in
        //  front of the first fragment read of every slice — it could not tell the LDS-DMA writes of the NEXT slice, just
        //  issued into the other stage, from the stage being read — which serialised load and compute)
        f32x4 af[2][TM], bf[2][TN];
        auto frag = [&](int g, int b) {
            const int col = ((2 * g + lh) << 4) ^ swz;
#pragma unroll
            for (int i = 0; i < TM; ++i) af[b][i] = *reinterpret_cast<const f32x4*>(base + a_rd + i * 32 * G::ROWB + col);
            if constexpr (BREG) {
#pragma unroll
                for (int j = 0; j < TN; ++j) bf[b][j] = breg[SET][g][j];
            } else {
#pragma unroll
                for (int j = 0; j < TN; ++j) bf[b][j] = *reinterpret_cast<const f32x4*>(base + b_rd + j * 32 * G::ROWB + col);
            }
        };
        frag(0, 0);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int b = g & 1;
            if (g + 1 < 4) frag(g + 1, b ^ 1);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[b][i][e], bf[b][j][e], acc[i][j], 0, 0, 0);
        }
        // pin the order: the fragment reads of k-group g + 1 BEFORE the MFMAs of group g (left alone, the scheduler issues each
        // read pair right before its use and the wave waits out the LDS latency four times per slice)
        constexpr int NDS = BREG ? TM : TM + TN;      // LDS fragment reads per k-group
        __builtin_amdgcn_sched_group_barrier(0x100, NDS, 0);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            if (g + 1 < 4) __builtin_amdgcn_sched_group_barrier(0x100, NDS, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 4 * TM * TN, 0);
        }
    };

    E epi;
    const bool finisher = !split || part == a.parts - 1;   // this workgroup runs the epilogue
    if constexpr (EPI == 1) {
        if (finisher) epi.rows(a, m0, tid, m_end, gbase, grp);
    }
    using S0 = std::integral_constant<int, 0>;
    using S1 = std::integral_constant<int, 1>;
    if constexpr (ST == 2 && BREG) {
        // as below, unrolled by two so that the register set of the B fragments is static
        auto step = [&](auto cur_tag, auto nxt_tag, int it) {
            wait_dma();
            __syncthreads();
            if (it + 1 < nsl) issue(decltype(nxt_tag)::value, nxt_tag);
            if constexpr (EPI == 1) {
                if (it + 1 == nsl && finisher) epi.prefetch(a, n0, tid);
            }
            mfmas(smem + decltype(cur_tag)::value * G::STAGE, cur_tag);
        };
        if (nsl > 0) issue(0, S0{});
        int it = 0;
        for (; it + 1 < nsl; it += 2) {
            step(S0{}, S1{}, it);
            step(S1{}, S0{}, it + 1);
        }
        if (it < nsl) step(S0{}, S1{}, it);
        __syncthreads();
    } else if constexpr (ST == 2) {
        if (nsl > 0) issue(0, S0{});
        for (int it = 0; it < nsl; ++it) {
            wait_dma();
            __syncthreads();   // slice `it` has landed for every wave, and every wave is done with the other stage
            if (it + 1 < nsl) issue((it + 1) & 1, S0{});
            if constexpr (EPI == 1) {
                if (it + 1 == nsl && finisher) epi.prefetch(a, n0, tid);   // epilogue operands ride behind the last slice's MFMAs
            }
            mfmas(smem + (it & 1) * G::STAGE, S0{});
        }
        __syncthreads();   // every wave is past its last fragment read: the stages become the epilogue image
    } else {
        for (int it = 0; it < nsl; ++it) {
            issue(0, S0{});
            wait_dma();
            __syncthreads();
            if constexpr (EPI == 1) {
                if (it + 1 == nsl && finisher) epi.prefetch(a, n0, tid);
            }
            mfmas(smem, S0{});
            __syncthreads();
        }
    }

    if (split) {
        // Partials are [part][(i*TN+j)*16 + r][256 threads] floats (one coalesced 256-byte row per wave and access), agent-scope
        // accesses; readers have a higher block index than writers (no dispatch deadlock) — igemm_kernel's protocol
        float* pbase = a.partials + (size_t)tail * (a.parts - 1) * (BM * BN);
        int* flag = a.flags + tail * (a.parts - 1);
        if (part < a.parts - 1) {
            float* o = pbase + (size_t)part * (BM * BN) + tid;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) st_agent(o + ((i * TN + j) * 16 + r) * 256, acc[i][j][r]);
            wait_stores();
            __syncthreads();
            if (tid == 0) st_agent_flag(flag + part, 1);
            return;
        }
        for (int pp = 0; pp < a.parts - 1; ++pp) {
            if (tid == 0) spin_until_set(flag + pp);
            __syncthreads();
            const float* o = pbase + (size_t)pp * (BM * BN) + tid;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[i][j][r] += ld_agent(o + ((i * TN + j) * 16 + r) * 256);
        }
        __syncthreads();
        if (tid < a.parts - 1) st_agent_flag(flag + tid, 0);   // consumed: ready for the next launch on this stream
    }

    if constexpr (EPI == 1) {
        epi.run(a, acc, smem, mt, m0, n0, tid, wm, wn, l31, lh);
    } else {
        igemm_epilogue<BM, BN, PERM>(a, acc, reinterpret_cast<float*>(smem), mt, m0, n0, wm, wn, l31, lh);
    }
    igemm_fold_arrive<BN>(a, mt, n0, smem);
}


// ======================================================================================================================
// fp32 weight gradient with both operand slices HBM -> LDS by LDS-DMA.  dW[co][tap*Cp + ci] = sum over pixels of
// dY[pixel][co] * X[src(pixel, tap)][ci]: both operands are channel-contiguous in HBM and the exact-fp32 MFMA takes ONE k per
// lane (lanes 0-31: pixel 2kk, lanes 32-63: pixel 2kk + 1), so the [pixel][channel] rows go to LDS as they are — no transpose,
// no staging registers (32 VGPRs of wgrad_kernel<128,128>), no ds_write pass, no zero selects (padding taps and pixels past the
// end of the split are zero-filled by the buffer descriptor) — and the fragments are the ds_read_b32 pairs wgrad_kernel reads.
// Same pixel order, same 32-pixel slices, same k pairing as wgrad_kernel: every split slab is bit-identical to it.
// A 32-entry table per slice (pixel -> byte offsets of its dY row and of filter tap (0,0) of its X row + the tap-(0,0)
// coordinates) is computed by wave 0 one slice ahead (wgrad_glds_kernel's scheme), so the other waves only add lane constants.
// ST: LDS stages (2: 64 KB per 128x128 workgroup, the next slice in flight during the MFMAs; 1: 32 KB, for grids of more than
//     two workgroups per CU, like wgrad_kernel's single-buffer form).
// ======================================================================================================================
template <int BM, int BN, int ST>
struct WGeomF {
    static constexpr int KP = 32;                                           // pixels per slice
    static constexpr int ROWA = BM * 4, ROWB_ = BN * 4;                     // bytes per pixel row of the two slices
    static constexpr int A_BYTES = KP * ROWA, B_BYTES = KP * ROWB_, STAGE = A_BYTES + B_BYTES;
    static constexpr int TAB_OFF = ST * STAGE;                              // PixRec[2][KP]
    static constexpr int TOTAL = TAB_OFF + 2 * KP * (int)sizeof(PixRec);
};

template <int BM, int BN, int ST = 2, int OCC = 2>
__global__ void __launch_bounds__(256, OCC) wgrad_glds32_kernel(WgradArgs a) {
    using G = WGeomF<BM, BN, ST>;
    static_assert(ST == 1 || ST == 2, "one or two LDS stages");
    constexpr int KP = G::KP;
    constexpr int TM = BM / 64, TN = BN / 64;
    constexpr int RPA = 1024 / G::ROWA, RPB = 1024 / G::ROWB_;    // pixel rows per LDS-DMA instruction (2 or 4)
    constexpr int NIA = KP / (4 * RPA), NIB = KP / (4 * RPB);     // instructions per wave, slice and operand (4 or 2)
    constexpr int CHA = G::ROWA / 16, CHB = G::ROWB_ / 16;         // 16-byte chunks per row (32 or 16)
    __shared__ __attribute__((aligned(16))) unsigned char smem[G::TOTAL];
    PixRec* const tab = reinterpret_cast<PixRec*>(smem + G::TAB_OFF);

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = uniform(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, lh = lane >> 5;

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

    // per-lane constants of the LDS-DMA: pixel row inside the instruction's group and 16-byte slot (= 4 channels / columns)
    const int prowA = lane / CHA, slotA = lane % CHA;
    const int prowB = lane / CHB, slotB = lane % CHB;
    // A: channels co0 + 4 * slot .. + 3 of the dY row (channels beyond K only feed rows of dW that are never stored)
    const uint32_t coffA = (uint32_t)(co0 + 4 * slotA) * 4u;
    // B: GEMM columns col0 + 4 * slot .. + 3 = one filter tap, 4 input channels
    const int colB = col0 + 4 * slotB;
    const bool colokB = colB < a.Ncols;
    const int tapB = fdiv(colokB ? colB : 0, a.fCp);
    const int ciB = (colokB ? colB : 0) - tapB * a.Cp;
    const int rB = fdiv(tapB, a.fS);
    const int dhB = rB * a.dil, dwB = (tapB - rB * a.S) * a.dil;
    const int deltaB = (dhB * a.W + dwB) * a.ldx * 4 + ciB * 4;
    // columns past the end fail the row test below through a huge row offset: with `colokB && ...` in the per-slice predicate the
    // compiler split every B instruction into two exec-masked halves (the lanes with and without a real column)
    const int dhT = colokB ? dhB : (1 << 24);

    auto fill_table = [&](int sl, int slot) {   // wave 0: one lane per pixel of slice sl
        if (lane >= KP) return;
        const int m = mbeg + sl * KP + lane;
        const bool ok = m < mend;
        const int mm = ok ? m : mbeg;
        const int img = fdiv(mm, r_fhw);
        const int rem = mm - img * r_hw;
        const int pi = fdiv(rem, r_fw);
        const int qi = rem - pi * r_w;
        const int p = r_pl + pi, q = r_ql + qi;
        PixRec rec;
        rec.dyoff = ok ? (uint32_t)(((img * a.P + p) * a.Q + q) * a.ldy) * 4u : OOB;
        rec.h0 = ok ? p * a.stride - a.pad : -(1 << 20);
        rec.w0 = q * a.stride - a.pad;
        rec.xoff = ok ? ((img * a.H + rec.h0) * a.W + rec.w0) * a.ldx * 4 : 0;
        tab[slot * KP + lane] = rec;
    };
    auto issue = [&](int stage, int slot) {   // the slice whose table sits in `slot`, into LDS stage `stage`
        unsigned char* const As = smem + stage * G::STAGE;
        unsigned char* const Bs = As + G::A_BYTES;
        const PixRec* const t = tab + slot * KP;
        // all table records first (one 16-byte LDS read each, in flight together), then the LDS-DMA instructions: read -> wait ->
        // issue per record cost eight LDS round trips at the top of every slice
        i32x4 ra[NIA], rb[NIB];
#pragma unroll
        for (int i = 0; i < NIA; ++i) ra[i] = reinterpret_cast<const i32x4*>(t)[(wave + 4 * i) * RPA + prowA];
#pragma unroll
        for (int i = 0; i < NIB; ++i) rb[i] = reinterpret_cast<const i32x4*>(t)[(wave + 4 * i) * RPB + prowB];
#pragma unroll
        for (int i = 0; i < NIA; ++i) {   // PixRec: {dyoff, xoff, h0, w0}
            const uint32_t dyoff = (uint32_t)ra[i][0];
            load16_to_lds(rsA, dyoff == OOB ? OOB : dyoff + coffA, As + (wave + 4 * i) * 1024);
        }
#pragma unroll
        for (int i = 0; i < NIB; ++i) {
            const int h = rb[i][2] + dhT, w = rb[i][3] + dwB;
            const bool ok = (unsigned)h < (unsigned)a.H && (unsigned)w < (unsigned)a.W;
            load16_to_lds(rsB, ok ? (uint32_t)(rb[i][1] + deltaB) : OOB, Bs + (wave + 4 * i) * 1024);
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // fragments of k-step kk: lane (channel l31, pixel 2 kk + lh); the reads of step kk + 1 are issued before the MFMAs of step kk
    const int a_rd = (lh * BM + wm * (BM / 2) + l31) * 4;
    const int b_rd = G::A_BYTES + (lh * BN + wn * (BN / 2) + l31) * 4;
    auto mfmas = [&](const unsigned char* base) {
        float af[2][TM], bf[2][TN];
        auto frag = [&](int b, int kk) {
#pragma unroll
            for (int i = 0; i < TM; ++i) af[b][i] = *reinterpret_cast<const float*>(base + a_rd + 2 * kk * G::ROWA + i * 128);
#pragma unroll
            for (int j = 0; j < TN; ++j) bf[b][j] = *reinterpret_cast<const float*>(base + b_rd + 2 * kk * G::ROWB_ + j * 128);
        };
        frag(0, 0);
#pragma unroll
        for (int kk = 0; kk < KP / 2; ++kk) {
            const int b = kk & 1;
            if (kk + 1 < KP / 2) frag(b ^ 1, kk + 1);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[b][i], bf[b][j], acc[i][j], 0, 0, 0);
        }
        // pin the order (wgrad_kernel's finding: left alone the scheduler emits read, wait, MFMAs per step)
        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
#pragma unroll
        for (int kk = 0; kk < KP / 2; ++kk) {
            if (kk + 1 < KP / 2) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, TM * TN, 0);
        }
    };
    if (nsl > 0) {
        if (wave == 0) fill_table(0, 0);
        __syncthreads();
        if constexpr (ST == 2) {
            issue(0, 0);
            if (wave == 0 && nsl > 1) fill_table(1, 1);
            for (int it = 0; it < nsl; ++it) {
                wait_dma();
                __syncthreads();   // slice `it` has landed, table it+1 is written, everyone is done with the other stage
                if (it + 1 < nsl) issue((it + 1) & 1, (it + 1) & 1);
                if (wave == 0 && it + 2 < nsl) fill_table(it + 2, it & 1);
                mfmas(smem + (it & 1) * G::STAGE);
            }
        } else {
            for (int it = 0; it < nsl; ++it) {
                issue(0, it & 1);
                if (wave == 0 && it + 1 < nsl) fill_table(it + 1, (it + 1) & 1);
                wait_dma();
                __syncthreads();   // slice `it` has landed and table it+1 is written
                mfmas(smem);
                __syncthreads();   // everyone is done with the stage
            }
        }
    }

    float* out = a.slab + (size_t)split * a.K * a.Ncols;
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
