// bf16-STORAGE implicit GEMM, third generation: one 8-wave workgroup per CU on a (32 TM) x 256 tile (round 6).
// Included by conv_igemm.hip inside namespace up, after bf16s_glds.h.
//
// Why: igemm_glds_kernel's 128 x 128 tile moves (128 + 128) * 64 B through the CU's global -> LDS path per 32-channel slice for 8 MFMAs
// per wave, 64 B per MFMA-cycle of the CU — and gets 16-21 B per cycle (profiles/r06_n_bf16_quant.txt: 746 TFLOP/s at any number of
// tiles per CU).  The path itself delivers 37-52 B per cycle to a bare streaming loop (tools/gpu/fetch_probe.hip): the kernel is
// bound by how much it keeps in flight against the latency of operands that the previous kernel left in the Infinity Cache / HBM,
// and by the bytes it asks for per FLOP.  So the tile grows — (BM + 256) * 128 B per 64-channel slice for BM * 256 outputs is
// 2 BM * 256 / (BM + 256) FLOP per byte = 98 / 110 / 128 at BM = 160 / 192 / 256 against 64 — and the pipeline deepens:
//  * 512 threads = 8 wavefronts, wave w owns the 32 output columns n0 + 32 w .. + 31 of ALL BM rows (TM accumulator tiles of
//    32 x 32, <= 128 registers): one B fragment and TM A fragments per 16-channel step; a column's BatchNorm statistics and
//    backward sums never leave the wave.
//  * K slice = 64 channels: 128-byte LDS rows = whole cache lines; stages of (BM + 256) * 128 B, ONE workgroup per CU.  BM = 160
//    keeps THREE stages (156 KB): two slices in flight behind the one computed, per-wave counted `s_waitcnt vmcnt(n)` + a raw
//    s_barrier (a __syncthreads() drains the LDS-DMA queue); BM = 192 / 256: two stages, one slice in flight.
//  * the LDS-DMA pieces of the slice being fetched are issued BETWEEN the MFMAs of the slice being computed (<= 2 per 16-channel
//    step and wave), and the TM + 1 fragment reads of step s + 1 before the MFMAs of step s (two fragment sets, sched_group_barrier
//    pins).  Left alone the scheduler kept four fragment registers and emitted read, read, wait, MFMA, MFMA.
//  * BM = 160 / 192 / 256 per launch (big_tile_rows): the 46 x 46 stage has M = 33 856 rows, 264.5 tiles of 128 — the row count
//    per tile is chosen so that the launch fills whole rounds of the 256 CUs (212 tiles of 160 for N = 256, 708 of 192 for N = 1024).
//  * epilogue without workgroup barriers: every wave transposes its 32 x 32 blocks through 4 KB of its own LDS (bf16 row pairs,
//    or fp32 when an addend comes in before the single rounding) and stores 16 bytes = 8 channels per lane.
// One workgroup per CU leaves one prologue + epilogue per tile exposed (10-15 us): the launch logic sends only reductions of at
// least big_min_k = 1024 here (measured, profiles/r06_experiments.txt item 12).
// Same bits as igemm_glds_kernel for y / dx (same MFMA, same k order, skipped taps contribute exact zeros); BatchNorm partials
// have one row per BM-row tile (merged values agree to fp32 round-off).
#pragma once

namespace glds {

#ifdef UP_EMU
__device__ __forceinline__ void wave_sync() { ::emu::bar_wait(::emu::wave().bar, ::emu::wave_lanes()); }
#else
// LDS operations of one wave execute in order: a wait for the wave's own outstanding LDS operations orders a lane's read behind
// another lane's write; the wave barrier keeps the compiler from moving LDS accesses across it
__device__ __forceinline__ void wave_sync() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
}
#endif

template <int TM, int NS>
struct BigGeom {
    static constexpr int BM = 32 * TM, BN = 256, NT = 512;
    static constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, STAGE = A_BYTES + B_BYTES;
    static constexpr int MAIN = NS * STAGE;      // >= 8 waves x 4 KB of epilogue image, >= FOLD_LDS_BYTES
    static constexpr int MASK_OFF = MAIN;        // tap masks of the eight waves
    // BM % 64 != 0: half of the waves have one LDS-DMA piece of A less.  Two stages: they issue a zero-fill piece into 1 KB of
    // their own instead (the slice stays one basic block); three stages (no room): a uniform branch in front of the pinned block
    static constexpr bool DUMMY = BM % 64 != 0 && NS == 2;
    static constexpr int DUMMY_OFF = MASK_OFF + 32;
    static constexpr int TOTAL = DUMMY_OFF + (DUMMY ? 8 * 1024 : 0);
    static_assert(TOTAL <= 160 * 1024, "LDS of one CU");
};

#ifdef UP_EMU
template <int N>
__device__ __forceinline__ void wait_dma_but() {}
__device__ __forceinline__ void raw_barrier() { __syncthreads(); }
#else
// all but the N youngest LDS-DMA pieces of this wave have landed
template <int N>
__device__ __forceinline__ void wait_dma_but() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
// workgroup barrier that does not drain the LDS-DMA queue (__syncthreads() carries a vmcnt(0))
__device__ __forceinline__ void raw_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
#endif

// TM:    32-row accumulator tiles per wave (BM = 32 TM rows per workgroup tile)
// PERM:  GEMM row m is output pixel a.perm[m] (tap-sorted order)
// BNRED: the launch's output is dz of the layer z = relu(bn(y) (+ res)); the epilogue also reduces that layer's BatchNorm-backward
//        sums over the tile's rows (see store_tile in bf16s_glds.h)
// NS:    LDS stages: 2 (one slice in flight behind the one computed) or 3 (two in flight, counted vmcnt + raw barrier; BM <= 160)
template <int TM, bool PERM, bool BNRED, int NS = (TM <= 5 ? 3 : 2)>
__global__ void __launch_bounds__(512, 1) igemm_big_kernel(IgemmArgs a) {
    using SL = Slice<64>;
    using G = BigGeom<TM, NS>;
    constexpr int BM = G::BM, BN = G::BN;
    constexpr int GA = BM / 8;                 // LDS-DMA instructions (8 rows x 128 B) per A slice
    constexpr int NA = (GA + 7) / 8, NB = 4;   // per wave
    __shared__ __attribute__((aligned(16))) unsigned char smem[G::TOTAL];
    unsigned* const wmask = reinterpret_cast<unsigned*>(smem + G::MASK_OFF);

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = uniform(tid >> 6);
    const int l31 = lane & 31, lh = lane >> 5;

    const int logical = xcd_remap(blockIdx.x, a.nwg);
    const int mt = fdiv(logical, a.fNtn);
    const int nt = logical - mt * a.ntn;
    const int m0 = mt * BM, n0 = nt * BN;

    // this lane's rows of the operand slices: row 8 g + lane / 8 of LDS-DMA group g = wave + 8 i, 16-byte slot lane % 8
    const int rsub = lane >> 3, slot = lane & 7;
    const int R = a.taps / a.S;
    const Rsrc rsA = make_rsrc(a.x, a.x_bytes);
    const Rsrc rsB = make_rsrc(a.w_hi, (uint32_t)a.Ng * (uint32_t)a.Ktot * 2u);
    uint32_t woffB[NB];
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        const int row = (wave + 8 * j) * 8 + rsub;
        const int n = n0 + row;
        woffB[j] = n < a.Ng ? (uint32_t)n * (uint32_t)a.Ktot * 2u + (uint32_t)((slot ^ SL::swz(row)) << 4) : OOB;
    }
    auto issueB = [&](int stage, int tap_, int cs_) {
        unsigned char* const Bs = smem + stage * G::STAGE + G::A_BYTES;
        const uint32_t kb = (uint32_t)(tap_ * a.Cp + cs_ * 64) * 2u;
#pragma unroll
        for (int j = 0; j < NB; ++j)
            load16_to_lds(rsB, woffB[j] == OOB ? OOB : woffB[j] + kb, Bs + (wave + 8 * j) * 1024);
    };
    const bool pointwise = a.taps == 1 && a.mul == 1 && a.off0 == 0 && a.off0w == 0 && a.H == a.P && a.W == a.Q;
    if (pointwise) issueB(0, 0, 0);   // the first weight slice needs no row set-up

    int roffA[NA];
    unsigned tmA[NA];
    unsigned tile_taps = 0;
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        const int g = wave + 8 * i;
        const int row = g * 8 + rsub;
        const int m = m0 + row;
        int pix = m < a.M ? m : a.M - 1;
        if constexpr (PERM) pix = a.perm[pix];
        unsigned mk;
        int src;
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
            unsigned hm = 0, wmk = 0;
            for (int r = 0; r < R; ++r) hm |= ((unsigned)(hb + r * a.tapstep) < (unsigned)a.H) ? (1u << r) : 0u;
            for (int s = 0; s < a.S; ++s) wmk |= ((unsigned)(wb + s * a.tapstep) < (unsigned)a.W) ? (1u << s) : 0u;
            mk = 0u;
            for (int r = 0; r < R; ++r) mk |= ((hm >> r) & 1u) ? (wmk << (r * a.S)) : 0u;
        }
        if (m >= a.M || g >= GA) mk = 0u;
        roffA[i] = src * a.ldx * 2 + ((slot ^ SL::swz(row)) << 4);
        tmA[i] = mk;
        tile_taps |= mk;
    }
    const unsigned all_taps = a.taps >= 32 ? 0xffffffffu : ((1u << a.taps) - 1u);
    unsigned live = all_taps;
    if (a.taps > 1 && !a.no_tap_skip) {
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) tile_taps |= __shfl_xor(tile_taps, off);
        if (lane == 0) wmask[wave] = tile_taps;
        __syncthreads();
        unsigned u = 0;
#pragma unroll
        for (int w = 0; w < 8; ++w) u |= wmask[w];
        live = (unsigned)uniform((int)u);
        if (live == 0u) live = all_taps;
    }
    const int spt = a.Cp / 64;
    const int nsl = __builtin_popcount(live) * spt;
    unsigned rest = live;
    int tap = __builtin_ctz(rest), cs = 0;

    bool b_issued = pointwise;
    // all LDS-DMA pieces of the next slice into `stage` (prologue); more = false: zero-fill pieces (every wave issues the same
    // number of pieces per slice whatever the reduction length, which is what the counted waits of the loop assume)
    auto issue = [&](int stage, bool more) {
        unsigned char* const As = smem + stage * G::STAGE;
        const int r = fdiv(tap, a.fS);
        const int sx = tap - r * a.S;
        const int delta = ((r * a.tapstep) * a.W + sx * a.tapstep) * a.ldx * 2 + cs * 128;
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int g = wave + 8 * i;
            if (GA % 8 == 0 || g < GA) {   // (uniform per wave)
                const bool ok = more && ((tmA[i] >> tap) & 1u);
                load16_to_lds(rsA, ok ? (uint32_t)(roffA[i] + delta) : OOB, As + g * 1024);
            } else if (G::DUMMY) {
                load16_to_lds(rsA, OOB, smem + G::DUMMY_OFF + wave * 1024);
            }
        }
        if (!b_issued) {
            unsigned char* const Bs = As + G::A_BYTES;
            const uint32_t kb = (uint32_t)(tap * a.Cp + cs * 64) * 2u;
#pragma unroll
            for (int j = 0; j < NB; ++j)
                load16_to_lds(rsB, (more && woffB[j] != OOB) ? woffB[j] + kb : OOB, Bs + (wave + 8 * j) * 1024);
        }
        b_issued = false;
        if (++cs == spt) {
            cs = 0;
            rest &= rest - 1u;
            tap = rest ? __builtin_ctz(rest) : 0;
        }
    };

    f32x16 acc[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

    const int swz = SL::swz(l31) << 4;
    const int a_rd = l31 * 128;
    const int b_rd = G::A_BYTES + (wave * 32 + l31) * 128;
    unsigned char* const dummy = smem + G::DUMMY_OFF + wave * 1024;   // landing place of the LDS-DMA pieces a wave has no rows for
    constexpr bool ODD_FIRST = GA % 8 != 0 && !G::DUMMY;

    // One slice of the K loop: the 4 x (TM + 1) fragment reads and 4 x TM MFMAs on `base`, with the LDS-DMA pieces of the NEXT slice
    // (<= 2 per 16-channel step and wave) issued BETWEEN the MFMAs.  Two things the first build (issue everything, then compute) lost:
    //  * the scheduler kept four fragment registers and emitted read, read, wait, MFMA, MFMA — with two waves per SIMD nothing hides
    //    the LDS latency; here the TM + 1 reads of step s + 1 are pinned before the MFMAs of step s (two fragment sets);
    //  * an LDS-DMA instruction stalls its wave 60-180 cycles at issue (MI355X_MICROARCH.md): seven of them in a row after the
    //    barrier, in all eight waves at once, left the MFMA pipe idle for a third of the slice.
    // more = false (last slice): the pieces are issued with out-of-range offsets (zero fill, no memory traffic), so the loop body
    // stays one basic block and the pins hold.
    auto slice = [&](const unsigned char* base, unsigned char* nxt, bool more) {
        const int r = fdiv(tap, a.fS);
        const int sx = tap - r * a.S;
        const int delta = ((r * a.tapstep) * a.W + sx * a.tapstep) * a.ldx * 2 + cs * 128;
        const uint32_t kb = (uint32_t)(tap * a.Cp + cs * 64) * 2u;
        if constexpr (ODD_FIRST) {   // the A piece only half of the waves have: in front of the pinned block
            if (wave + 8 * (NA - 1) < GA) {
                const bool ok = more && ((tmA[NA - 1] >> tap) & 1u);
                load16_to_lds(rsA, ok ? (uint32_t)(roffA[NA - 1] + delta) : OOB, nxt + (wave + 8 * (NA - 1)) * 1024);
            }
        }
        bf16x8 af[2][TM], bf[2];
        auto frag = [&](int s, int b) {
            const int col = (((2 * s + lh) << 4) ^ swz);
#pragma unroll
            for (int i = 0; i < TM; ++i) af[b][i] = *reinterpret_cast<const bf16x8*>(base + a_rd + i * 32 * 128 + col);
            bf[b] = *reinterpret_cast<const bf16x8*>(base + b_rd + col);
        };
        frag(0, 0);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int b = s & 1;
            if (s + 1 < 4) frag(s + 1, b ^ 1);
            if (s < NA && (s < NA - 1 || !ODD_FIRST)) {
                const int g = wave + 8 * s;
                const bool real = GA % 8 == 0 || g < GA;   // (uniform per wave; tmA is 0 for the rows a wave does not have)
                const bool ok = more && ((tmA[s] >> tap) & 1u);
                load16_to_lds(rsA, ok ? (uint32_t)(roffA[s] + delta) : OOB, real ? nxt + g * 1024 : dummy);
            }
            load16_to_lds(rsB, (more && woffB[s] != OOB) ? woffB[s] + kb : OOB, nxt + G::A_BYTES + (wave + 8 * s) * 1024);
#pragma unroll
            for (int i = 0; i < TM; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[b][i], bf[b], acc[i], 0, 0, 0);
        }
        // order: reads of step 0 | per step: first MFMA, reads of step s + 1, MFMAs, piece, MFMAs, piece (only the first TM + 1 reads
        // of a slice have nothing to hide behind)
        __builtin_amdgcn_sched_group_barrier(0x100, TM + 1, 0);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            if (s + 1 < 4) __builtin_amdgcn_sched_group_barrier(0x100, TM + 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, TM / 2 - 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, TM - TM / 2, 0);
            if (s < NA && (s < NA - 1 || !ODD_FIRST)) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
        }
        if (++cs == spt) {   // the slice after the one just issued
            cs = 0;
            rest &= rest - 1u;
            tap = rest ? __builtin_ctz(rest) : 0;
        }
    };

    // NS - 1 slices ahead: slice it + NS - 1 is issued (between the MFMAs) while slice it is computed; its stage is the one slice
    // it - 1 was read from, and every wave has passed this iteration's barrier, i.e. finished those reads
#pragma unroll
    for (int k = 0; k < NS - 1; ++k) issue(k, k < nsl);
    int cur = 0, nx = NS - 1;
    for (int it = 0; it < nsl; ++it) {
        if constexpr (NS == 2) {
            wait_dma();
            __syncthreads();   // slice `it` has landed for every wave, and every wave is done with the other stage
        } else {
            // this wave's pieces of slice `it` have landed: all but the (NS - 2) younger slices' pieces (A pieces it has + 4 of B)
            constexpr int PA_FULL = (GA + 7) / 8, PA_LESS = GA / 8;
            if (GA % 8 == 0 || wave + 8 * (NA - 1) < GA) wait_dma_but<(NS - 2) * (PA_FULL + 4)>();
            else wait_dma_but<(NS - 2) * (PA_LESS + 4)>();
            raw_barrier();
        }
        slice(smem + cur * G::STAGE, smem + nx * G::STAGE, it + NS - 1 < nsl);
        cur = cur + 1 == NS ? 0 : cur + 1;
        nx = nx + 1 == NS ? 0 : nx + 1;
    }
    wait_dma();        // (the zero-fill pieces of the last slices)
    __syncthreads();   // every wave is past its last fragment read: the stages become the waves' epilogue images

    // ---------------------------------------------------------------- epilogue (per wave, no workgroup barrier) ----
    const int ncw = n0 + wave * 32;            // first column of this wave
    const bool full = m0 + BM <= a.M;          // uniform
    if (a.stats) {
        // (count, mean, M2) of this lane's column over the tile's rows: the two half-waves hold the rows 4 lh + ... of every 8
        float s = 0.f, c = 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (full || m < a.M) {
                    s += acc[i][r];
                    c += 1.f;
                }
            }
        s += __shfl_xor(s, 32);
        c += __shfl_xor(c, 32);
        const float mean = c > 0.f ? s / c : 0.f;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (full || m < a.M) {
                    const float d = acc[i][r] - mean;
                    q += d * d;
                }
            }
        q += __shfl_xor(q, 32);
        const int n = ncw + l31;
        if (lh == 0 && n < a.Ng) {
            float* o = a.stats + ((size_t)mt * a.Ng + n) * 3;   // (sc1: the last arriver of the fold reads them)
            st_agent(o, c);
            st_agent(o + 1, mean);
            st_agent(o + 2, q);
        }
    }

    const bool relu = a.relu != 0;
    const bool affine = a.scale != nullptr || a.bias != nullptr;
    float esc = 1.f, esh = 0.f;
    {
        const int n = ncw + l31;
        const int nn = n < a.Ng ? n : a.Ng - 1;
        if (a.scale) {
            esc = a.scale[nn];
            esh = a.shift[nn];
        }
        if (a.bias) esh += a.bias[nn];
    }
    auto opix = [&](int row) {   // destination pixel of tile row `row`, -1 past the end
        const int m = m0 + row;
        if (m >= a.M) return -1;
        if constexpr (PERM) return a.perm[m];
        return m;
    };
    auto bits8_of = [&](const uint32_t* bits, int C, int p, int n) -> uint32_t {
        const long long b = (long long)p * C + n;
        return (bits[b >> 5] >> (int)(b & 31)) & 0xffu;
    };
    bf16_t* const yo = reinterpret_cast<bf16_t*>(a.y);
    unsigned char* const wbuf = smem + wave * 4096;
    const int cq = lane & 3;                   // this lane's 8-channel chunk of the wave's 32 columns
    const int nq = ncw + cq * 8;
    const bool nok = nq < a.Ng;
    const int nn = nok ? nq : 0;

    float s1a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, s2a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    float mu8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if constexpr (BNRED) {
        if (nok) {
            const float4 m0v = *reinterpret_cast<const float4*>(a.bn_mean + nq), m1v = *reinterpret_cast<const float4*>(a.bn_mean + nq + 4);
            mu8[0] = m0v.x; mu8[1] = m0v.y; mu8[2] = m0v.z; mu8[3] = m0v.w;
            mu8[4] = m1v.x; mu8[5] = m1v.y; mu8[6] = m1v.z; mu8[7] = m1v.w;
        }
    }
    auto bn_acc = [&](const float (&v)[8], const uint4& yy, uint32_t bits8) {
        const float y[8] = {bf_lo(yy.x), bf_hi(yy.x), bf_lo(yy.y), bf_hi(yy.y), bf_lo(yy.z), bf_hi(yy.z), bf_lo(yy.w), bf_hi(yy.w)};
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float g = ((bits8 >> e) & 1u) ? v[e] : 0.f;
            s1a[e] += g;
            s2a[e] += g * (y[e] - mu8[e]);
        }
    };
    const bf16_t* const ybn = reinterpret_cast<const bf16_t*>(a.bn_y);

    if (!a.residual) {
        // bf16 row pairs: word [row pair][column] (registers r, r + 1 of an accumulator hold consecutive rows); a lane reads
        // 8 words = 8 channels x 2 rows and byte-permutes them into two 16-byte rows.  Two 2 KB images alternate.
        const int rp = lane >> 2;
        // Destination pixels (tap-sorted order: a load each) and the operands of the fused reduction are requested for CH blocks at
        // a time BEFORE the first of their stores: a load issued between the stores of the previous block could only be waited for
        // together with those stores (loads and stores retire on one counter) — the first build paid a store round trip per block,
        // 112 us against 76 us for the same 3x3 launch without the reduction.
        constexpr int CH = TM <= 6 ? TM : 4;
        int p0[CH], p1[CH];
        uint4 ya[BNRED ? CH : 1], yb[BNRED ? CH : 1];
        uint32_t ba[BNRED ? CH : 1], bb[BNRED ? CH : 1];
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            uint32_t* const img = reinterpret_cast<uint32_t*>(wbuf + (i & 1) * 2048);
            if (i % CH == 0) {
#pragma unroll
                for (int k = 0; k < CH; ++k) {
                    if (i + k >= TM) break;
                    p0[k] = opix((i + k) * 32 + 2 * rp);
                    p1[k] = opix((i + k) * 32 + 2 * rp + 1);
                }
                if constexpr (BNRED) {
#pragma unroll
                    for (int k = 0; k < CH; ++k) {
                        if (i + k >= TM) break;
                        const int r0 = p0[k] >= 0 ? p0[k] : 0, r1 = p1[k] >= 0 ? p1[k] : 0;
                        ya[k] = *reinterpret_cast<const uint4*>(ybn + (uint32_t)(r0 * a.bn_ld + nn));
                        yb[k] = *reinterpret_cast<const uint4*>(ybn + (uint32_t)(r1 * a.bn_ld + nn));
                        ba[k] = a.bn_bits ? bits8_of(a.bn_bits, a.bn_C, r0, nn) : 0xffu;
                        bb[k] = a.bn_bits ? bits8_of(a.bn_bits, a.bn_C, r1, nn) : 0xffu;
                    }
                }
            }
            const int ic = i % CH;
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                float v0 = acc[i][r], v1 = acc[i][r + 1];
                if (affine) {
                    v0 = v0 * esc + esh;
                    v1 = v1 * esc + esh;
                }
                if (relu) {
                    v0 = fmaxf(v0, 0.f);
                    v1 = fmaxf(v1, 0.f);
                }
                const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
                img[(row >> 1) * 32 + l31] = pack_bf16x2(v0, v1);
            }
            wave_sync();
            const uint4 w0 = *reinterpret_cast<const uint4*>(img + rp * 32 + cq * 8);
            const uint4 w1 = *reinterpret_cast<const uint4*>(img + rp * 32 + cq * 8 + 4);
            wave_sync();   // (the image of block i + 2 is written after block i + 1's first wave_sync: ordered behind these reads)
            if (!nok) continue;
            if constexpr (BNRED) {
                const uint32_t wv[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
                if (p0[ic] >= 0) {
                    float v[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = bf_lo(wv[e]);
                    bn_acc(v, ya[ic], ba[ic]);
                }
                if (p1[ic] >= 0) {
                    float v[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = bf_hi(wv[e]);
                    bn_acc(v, yb[ic], bb[ic]);
                }
            }
            if (p0[ic] >= 0)
                *reinterpret_cast<uint4*>(yo + (uint32_t)(p0[ic] * a.ldy + nq)) =
                    make_uint4(byte_perm(w0.y, w0.x, 0x05040100u), byte_perm(w0.w, w0.z, 0x05040100u),
                               byte_perm(w1.y, w1.x, 0x05040100u), byte_perm(w1.w, w1.z, 0x05040100u));
            if (p1[ic] >= 0)
                *reinterpret_cast<uint4*>(yo + (uint32_t)(p1[ic] * a.ldy + nq)) =
                    make_uint4(byte_perm(w0.y, w0.x, 0x07060302u), byte_perm(w0.w, w0.z, 0x07060302u),
                               byte_perm(w1.y, w1.x, 0x07060302u), byte_perm(w1.w, w1.z, 0x07060302u));
        }
    } else {
        // addend (residual / skip gradient): fp32 image of one 32 x 32 block, the addend joins before the single rounding
        float* const img = reinterpret_cast<float*>(wbuf);
        const bf16_t* const rs = reinterpret_cast<const bf16_t*>(a.residual);
        const int rrow = lane >> 2;   // + 16 * pass
        // (addend rows and reduction operands of two blocks at a time, requested before the first block's stores: see above)
        constexpr int CH2 = 2;
        int pxa[CH2][2];
        uint4 radda[CH2][2], ybn4a[BNRED ? CH2 : 1][2];
        uint32_t rmaska[CH2][2], bmaska[BNRED ? CH2 : 1][2];
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            if (i % CH2 == 0) {
#pragma unroll
                for (int k = 0; k < CH2; ++k) {
                    if (i + k >= TM) break;
#pragma unroll
                    for (int ps = 0; ps < 2; ++ps) {
                        pxa[k][ps] = opix((i + k) * 32 + ps * 16 + rrow);
                        const int r = pxa[k][ps] >= 0 ? pxa[k][ps] : 0;
                        radda[k][ps] = *reinterpret_cast<const uint4*>(rs + (uint32_t)(r * a.ldr + nn));
                        rmaska[k][ps] = a.res_bits ? bits8_of(a.res_bits, a.Ng, r, nn) : 0xffu;
                        if constexpr (BNRED) {
                            ybn4a[k][ps] = *reinterpret_cast<const uint4*>(ybn + (uint32_t)(r * a.bn_ld + nn));
                            bmaska[k][ps] = a.bn_bits ? bits8_of(a.bn_bits, a.bn_C, r, nn) : 0xffu;
                        }
                    }
                }
            }
            const int ic = i % CH2;
            const int (&px)[2] = pxa[ic];
            const uint4 (&radd)[2] = radda[ic];
            const uint32_t (&rmask)[2] = rmaska[ic];
            const uint4 (&ybn4)[2] = ybn4a[BNRED ? ic : 0];
            const uint32_t (&bmask)[2] = bmaska[BNRED ? ic : 0];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float v = acc[i][r];
                if (affine) v = v * esc + esh;
                const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
                img[row * 32 + l31] = v;
            }
            wave_sync();
            float4 f0[2], f1[2];
#pragma unroll
            for (int ps = 0; ps < 2; ++ps) {
                f0[ps] = *reinterpret_cast<const float4*>(img + (ps * 16 + rrow) * 32 + cq * 8);
                f1[ps] = *reinterpret_cast<const float4*>(img + (ps * 16 + rrow) * 32 + cq * 8 + 4);
            }
            wave_sync();   // the next block overwrites the image
#pragma unroll
            for (int ps = 0; ps < 2; ++ps) {
                if (!nok || px[ps] < 0) continue;
                const uint4 rr = radd[ps];
                float ad[8] = {bf_lo(rr.x), bf_hi(rr.x), bf_lo(rr.y), bf_hi(rr.y), bf_lo(rr.z), bf_hi(rr.z), bf_lo(rr.w), bf_hi(rr.w)};
                const uint32_t mb = rmask[ps];
#pragma unroll
                for (int e = 0; e < 8; ++e) ad[e] = ((mb >> e) & 1u) ? ad[e] : 0.f;
                float v[8] = {f0[ps].x + ad[0], f0[ps].y + ad[1], f0[ps].z + ad[2], f0[ps].w + ad[3],
                              f1[ps].x + ad[4], f1[ps].y + ad[5], f1[ps].z + ad[6], f1[ps].w + ad[7]};
                if (relu) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
                }
                const uint4 packed = make_uint4(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7]));
                if constexpr (BNRED) {
                    // the STORED (rounded) values, like the separate reduction pass reads them back
                    const float vr[8] = {bf_lo(packed.x), bf_hi(packed.x), bf_lo(packed.y), bf_hi(packed.y),
                                         bf_lo(packed.z), bf_hi(packed.z), bf_lo(packed.w), bf_hi(packed.w)};
                    bn_acc(vr, ybn4[ps], bmask[ps]);
                }
                *reinterpret_cast<uint4*>(yo + (uint32_t)(px[ps] * a.ldy + nq)) = packed;
            }
        }
    }
    if constexpr (BNRED) {
        // the 16 lanes that share a chunk column are 4 apart: fixed butterfly, deterministic; lanes 0..3 publish 8 channels each
#pragma unroll
        for (int off = 4; off < 64; off <<= 1)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                s1a[e] += __shfl_xor(s1a[e], off);
                s2a[e] += __shfl_xor(s2a[e], off);
            }
        if (lane < 4 && nok) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int cc = nq + e;
                float* o = a.bn_partial + ((size_t)mt * a.Ng + cc) * 2;
                st_agent(o, s1a[e]);
                st_agent(o + 1, s2a[e] * a.bn_invstd[cc]);
            }
        }
    }
    igemm_fold_arrive<BN>(a, mt, n0, smem);
}

// Rows per tile of a launch on igemm_big_kernel (0: the launch stays on igemm_glds_kernel): the candidate that needs the fewest
// (rounds of the chip) x (rows per tile); ties go to the taller tile (more FLOP per fetched byte).
static inline int big_tile_rows(long long M, int Ng, int Ktot, int Cp, int cus, int min_k, int only_rows) {
    if (Cp <= 0 || Cp % 64 != 0 || Ng % 256 != 0 || Ktot < min_k || M < 128) return 0;
    const int cand[3] = {256, 192, 160};
    long long best = -1;
    int bm = 0;
    for (int c : cand) {
        if (only_rows && c != only_rows) continue;
        const long long tiles = ((M + c - 1) / c) * (Ng / 256);
        const long long cost = ((tiles + cus - 1) / cus) * c;
        if (best < 0 || cost < best) {
            best = cost;
            bm = c;
        }
    }
    return bm;
}

}  // namespace glds
