// Implicit-GEMM convolution for gfx950 on the exact-fp32 matrix instruction
// v_mfma_f32_32x32x2_f32 (64 FLOP/clk/SIMD, bitwise an fmaf chain).
//
//   forward / data-gradient ("TN" GEMM, both operands k-contiguous in HBM):
//       Y[m][n] = sum_k A[m][k] * B[n][k]
//       m = output pixel (n_img,p,q), k = (tap, channel), A = on-the-fly im2col gather of the
//       NHWC activation (16-B loads along channels), B = re-laid weights [n][tap][channel].
//       The data gradient is the same kernel with source/destination roles swapped and the
//       divisibility test of a strided convolution folded into the gather predicate.
//   weight-gradient ("NT" GEMM, both operands m-contiguous), split-K over pixels:
//       dW[co][tap*Cp+ci] = sum_m dY[m][co] * X[src(m,tap)][ci]
//
// Tiling: 256 threads = 4 wavefronts (2x2), block tile BMxBN (128/64), BK=32, wave tile
// (BM/2)x(BN/2) built from 32x32 MFMA tiles.  One LDS buffer + register prefetch of the next
// K-slice (global loads stay in flight under the MFMAs); 3-4 blocks per CU hide the two barriers
// per slice (MI355X_MICROARCH: one f32 MFMA chain per wave already saturates the pipe).
// LDS rows are 36 dwords (32 + 4 pad): the ds_read_b128 fragment reads are conflict-free because
// 36/4 = 9 is odd (16 lanes of a b128 group hit 16 distinct 16-B slots).  A K-slice of 32 floats is one
// full 128-B line per gathered pixel row.  Pad channels (>= C) must hold zeros (finite), see the header.
// Within each group of 8 k the two half-waves take k = {0..3} / {4..7}: any k permutation is
// legal as long as A and B use the same one, and it turns 4 ds_read_b32 into one ds_read_b128.
#include "up_common.h"
#include "bn_fold.h"

#include <algorithm>
#include <map>
#include <mutex>
#include <type_traits>
#ifdef UP_EMU
#include <sched.h>
#endif

#include <stdlib.h>

#include <vector>

namespace up {

// ---- optional per-launch timing of the MFMA kernels (bench.py roofline leg) -------------------
// up_profile_begin() arms it; every igemm/wgrad launch is then bracketed by two hipEvents on the
// launch stream; up_profile_end() synchronises and returns, per kernel variant, {launches, total ms,
// total algorithmic flops}.  Off by default: no events, no overhead.
constexpr int PROF_VARIANTS = 40;
static const char* const kVariantNames[PROF_VARIANTS] = {
    "igemm_kernel<128,128,aligned>", "igemm_kernel<128,128,generic>", "igemm_kernel<64,128,aligned>",
    "igemm_kernel<64,128,generic>",  "igemm_kernel<128,64,aligned>",  "igemm_kernel<128,64,generic>",
    "igemm_kernel<64,64,aligned>",   "igemm_kernel<64,64,generic>",   "wgrad_kernel<128,128>",
    "wgrad_kernel<128,64>",          "wgrad_kernel<64,128>",          "wgrad_kernel<64,64>",
    "igemm_bf16_kernel<128,128>",    "igemm_bf16_kernel<64,128>",     "igemm_bf16_kernel<128,64>",
    "igemm_bf16_kernel<64,64>",      "wgrad_bf16_kernel<128,128>",    "wgrad_bf16_kernel<128,64>",
    "wgrad_bf16_kernel<64,128>",     "wgrad_bf16_kernel<64,64>",
    // bf16 storage, direct-to-LDS generation (bf16s_glds.h)
    "igemm_glds_kernel<128,128> (bf16)", "igemm_glds_kernel<64,128> (bf16)", "igemm_glds_kernel<128,64> (bf16)",
    "igemm_glds_kernel<64,64> (bf16)",   "wgrad_glds_kernel<128,128> (bf16)", "wgrad_glds_kernel<128,64> (bf16)",
    "wgrad_glds_kernel<64,128> (bf16)",  "wgrad_glds_kernel<64,64> (bf16)",
    // exact fp32, direct-to-LDS generation (f32_glds.h)
    "igemm_glds32_kernel<128,128>", "igemm_glds32_kernel<64,128>", "igemm_glds32_kernel<128,64>", "igemm_glds32_kernel<64,64>",
    "wgrad_glds32_kernel<128,128>", "wgrad_glds32_kernel<128,64>", "wgrad_glds32_kernel<64,128>", "wgrad_glds32_kernel<64,64>",
    // bf16 storage, 8-wave workgroups on (32 TM) x 256 tiles (bf16s_big.h)
    "igemm_big_kernel<256,256> (bf16)", "igemm_big_kernel<192,256> (bf16)", "igemm_big_kernel<160,256> (bf16)",
    // the 7x7 stride-2 first convolution (stem_f32.h)
    "stem7_kernel<128,64>"};
#ifndef UP_EMU
struct ProfRec {
    hipEvent_t a, b;
    int variant;
    double flops;        // nominal 2 M N K (real channels): SURVEY 8d's convention, what `roofline.achieved` is quoted on
    double flops_live;   // the same with a DILATED launch charged for its live (pixel, tap) pairs only (live_tap_share)
    int M, N, K, grid;   // GEMM view of the launch + workgroups (for the per-launch CSV)
};
static bool g_prof_on = false;
static double g_prof_live[64];   // per variant: live-tap FLOP of the last collection (up_profile_live_flops)
static std::vector<ProfRec> g_prof;
// events are recycled: creating two per launch cost a third of the profiled steps' overhead (bench line 62.9 vs 62.2 ms without
// profiling, round 4)
static std::vector<hipEvent_t> g_prof_pool;
static hipEvent_t prof_event() {
    if (!g_prof_pool.empty()) {
        hipEvent_t e = g_prof_pool.back();
        g_prof_pool.pop_back();
        return e;
    }
    hipEvent_t e;
    (void)hipEventCreate(&e);
    return e;
}
struct ProfScope {
    ProfRec r;
    hipStream_t st;
    bool on;
    ProfScope(int variant, double flops, hipStream_t s, int M = 0, int N = 0, int K = 0, int grid = 0, double live_share = 1.0)
        : st(s), on(g_prof_on) {
        if (!on) return;
        r.variant = variant;
        r.flops = flops;
        r.flops_live = flops * live_share;
        r.M = M;
        r.N = N;
        r.K = K;
        r.grid = grid;
        r.a = prof_event();
        r.b = prof_event();
        (void)hipEventRecord(r.a, st);
    }
    ~ProfScope() {
        if (!on) return;
        (void)hipEventRecord(r.b, st);
        g_prof.push_back(r);
    }
};
#else
struct ProfScope {
    ProfScope(int, double, hipStream_t, int = 0, int = 0, int = 0, int = 0, double = 1.0) {}
};
#endif

#ifndef UP_EMU
static inline bool g_prof_on_host() { return g_prof_on; }
#else
static inline bool g_prof_on_host() { return false; }
#endif

constexpr int BK = 32;   // K slice of the weight-gradient kernel (the conv/dgrad kernel takes it as a template parameter)
constexpr int KT_DEFAULT = 32;   // K slice every fp32 forward / data-gradient launch uses

struct IgemmArgs {
    const float* x;
    const float* w;
    const uint16_t* w_hi;   // bf16-operand kernels: packed weights as bf16 planes (hi, and lo = bf16(w - hi))
    const uint16_t* w_lo;
    float* y;
    int M, Ng, Ktot, Cp;
    long long Ktot_real;  // taps * real channels: the algorithmic reduction length (profiling only)
    int H, W, P, Q;  // source H,W; destination P,Q
    int ldx, ldy;
    int S, taps;
    int mul, off0, tapstep;       // src = (dst*mul + off0 + r*tapstep) >> divshift (if divisible)
    int off0w;                    // column offset (== off0 except in the parity classes of a stride-2 data gradient)
    int divshift, divmask;        // 0,0 (forward, stride-1 dgrad) or 1,1 (dgrad of a stride-2 conv)
    int ntn;                      // number of n tiles
    int nwg;
    FastDiv fPQ, fQ, fCp, fS, fNtn, fSpt;   // fSpt: K slices per filter tap (aligned kernels)
    const float* scale;
    const float* shift;
    const float* bias;
    const float* residual;
    int ldr;
    int relu;
    float* stats;
    // output row map of a stride-2 data-gradient parity class: GEMM row m = (img, hc, wc) of the class grid is written
    // to pixel (img, 2*hc + o_ph, 2*wc + o_pw) of the o_H x o_W image (o_mode 0: rows are written linearly)
    int o_mode, o_ph, o_pw, o_H, o_W, o_Wc;
    FastDiv fo_HcWc, fo_Wc;
    void* dbg;   // development probes only (tools/gpu/igemm_probe.hip)
    // tail split (see launch_igemm): blocks [0, full_blocks) compute whole tiles, the rest compute 1/parts of the
    // K range of a tail tile; parts 0..parts-2 publish raw accumulators, the last part adds them and runs the epilogue
    int full_blocks, parts;
    int no_tap_skip;   // UP_TAP_SKIP=0 (A/B runs): visit every filter tap
    float* partials;
    int* flags;
    // tap-sorted row order (igemm_kernel<..., PERM = true>, see TapSort): GEMM row m is output pixel perm[m]; pixels
    // with the same set of live filter taps are contiguous, so the tile-level tap skipping drops (nearly) every dead tap
    const int* perm;
    uint32_t x_bytes;   // bf16s_glds.h / f32_glds.h: bytes of the activation tensor behind x (num_records of its buffer descriptor)
    // f32_glds.h, BNRED: the output of this data-gradient launch is dz of the layer z = relu(bn(y) (+ res)); its epilogue also
    // reduces that layer's BatchNorm-backward sums per row tile: partial[row tile][channel][2] = {sum g, invstd * sum g (y - mean)}
    // masked addend (f32_glds.h / bf16s_glds.h epilogues): the residual / addend is dz of ANOTHER layer whose ReLU mask
    // (bit pixel * Ng + channel) is applied here, so that layer's backward need not materialise dz * [z > 0]
    const uint32_t* res_bits;
    const float* bn_y;
    const uint32_t* bn_bits;   // sign bits of z (bit pixel * bn_C + channel), nullptr: no ReLU
    const float* bn_mean;
    const float* bn_invstd;
    float* bn_partial;
    int bn_ld, bn_C;
    // row GROUPS (f32_glds.h; the video model's batched trunk, ops.bn_groups): the M rows are `M / grp_rows` groups of grp_rows
    // rows (the frames of a clip batch) and every group is tiled on its own — tile mt = (group mt / grp_tiles, tile mt % grp_tiles
    // of that group), rows past the group's end masked — so that no row tile straddles two groups and the per-tile BatchNorm
    // partials (forward statistics, fused backward sums) belong to ONE group.  grp_rows = 0: one group, tiles over all M rows.
    // bn_grp_stride: floats between two groups' bn_mean / bn_invstd vectors.
    int grp_rows, grp_tiles, bn_grp_stride;
    FastDiv fGrpTiles;
    // BatchNorm finalize folded into this launch (bn_fold.h): fold.tickets != nullptr -> every finishing workgroup arrives after its
    // partial row (a.stats in the forward, a.bn_partial in a BNRED data gradient); fold_nv = 3 / 2 says which
    BnFold fold;
    int fold_nv;
};

// end of an implicit-GEMM workgroup that ran the epilogue of tile (mt, n0..n0+BN): the fold's ticket (see bn_fold.h)
template <int BN>
__device__ __forceinline__ void igemm_fold_arrive(const IgemmArgs& a, int mt, int n0, unsigned char* lds) {
    if (a.fold.tickets == nullptr) return;   // uniform
    const int ncols = a.Ng - n0 < BN ? a.Ng - n0 : BN;
    int fg = 0, t = mt;
    if (a.grp_rows) {   // row groups: tile mt % grp_tiles of group mt / grp_tiles
        fg = fdiv(mt, a.fGrpTiles);
        t = mt - fg * a.grp_tiles;
    }
    if (a.fold_nv == 3) bn_fold_arrive<3>(a.fold, a.stats, t, n0, ncols, lds, fg);
    else bn_fold_arrive<2>(a.fold, a.bn_partial, t, n0, ncols, lds, fg);
}

__device__ __forceinline__ void wf_merge(float& n1, float& m1, float& s1, float n2, float m2, float s2) {
    if (n2 == 0.f) return;
    if (n1 == 0.f) {
        n1 = n2;
        m1 = m2;
        s1 = s2;
        return;
    }
    float n = n1 + n2;
    float d = m2 - m1;
    m1 = m1 + d * (n2 / n);
    s1 = s1 + s2 + d * d * (n1 * n2 / n);
    n1 = n;
}

// component-wise select: a `cond ? float4_a : float4_b` on the aggregates becomes an ADDRESS select and forces
// the prefetch registers into scratch memory.
__device__ __forceinline__ float4 keep_or_zero(bool k, float4 v) {
    return make_float4(k ? v.x : 0.f, k ? v.y : 0.f, k ? v.z : 0.f, k ? v.w : 0.f);
}

// agent-scope (device-coherent) accesses for the K-split partials and their ready flags: st_agent / ld_agent / st_agent_flag live
// in bn_fold.h (shared with the BatchNorm fold)
#ifdef UP_EMU
__device__ __forceinline__ void spin_until_set(const int* p) {
    while (__atomic_load_n(p, __ATOMIC_ACQUIRE) == 0) sched_yield();
}
__device__ __forceinline__ void wait_stores() {}
#else
__device__ __forceinline__ void spin_until_set(const int* p) {
    while (__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) __builtin_amdgcn_s_sleep(8);
}
__device__ __forceinline__ void wait_stores() { __builtin_amdgcn_s_waitcnt(0); }   // vmcnt(0): stores acknowledged
#endif

// XCD-aware bijective remap: hardware block b runs on XCD b%8; give every XCD a contiguous run of
// logical tiles so the n-tiles sharing one A row-panel hit the same L2.
__device__ __forceinline__ int xcd_remap(int b, int nwg) {
    int q = nwg >> 3, r = nwg & 7;
    int xcd = b & 7, idx = b >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

template <int BM, int BN, bool RES, bool CHECK, bool OMAP = false, bool PMAP = false, typename TO = float>
__device__ __forceinline__ void igemm_store(const IgemmArgs& a, f32x16 (&acc)[BM / 64][BN / 64], int mrow0, int ncol0) {
    constexpr int TM = BM / 64, TN = BN / 64;
    const bool relu = a.relu != 0;
    TO* const yo = reinterpret_cast<TO*>(a.y);                       // TO = bf16_t: bf16 storage (output and residual)
    const TO* const rs = reinterpret_cast<const TO*>(a.residual);
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = ncol0 + j * 32;
        if (n >= a.Ng) continue;
        const float sc = a.scale ? a.scale[n] : 1.f;
        float sh = a.scale ? a.shift[n] : 0.f;
        if (a.bias) sh += a.bias[n];
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int mb = mrow0 + i * 32;
            float res[16];
            if (RES) {   // all 16 loads first (rows beyond M read a clamped, valid row), then one wait
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = mb + (r & 3) + 8 * (r >> 2);
                    res[r] = ld1(rs + (size_t)(!CHECK || m < a.M ? m : a.M - 1) * a.ldr + n);
                }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = mb + (r & 3) + 8 * (r >> 2);
                float v = acc[i][j][r] * sc + sh;
                if (RES) v += res[r];
                v = relu ? fmaxf(v, 0.f) : v;
                size_t pix = (size_t)m;
                if (OMAP) {   // stride-2 data-gradient parity class: every other row / column of the image
                    const int img = fdiv(m, a.fo_HcWc);
                    const int rem = m - img * (int)a.fo_HcWc.d;
                    const int hc = fdiv(rem, a.fo_Wc);
                    const int wc = rem - hc * a.o_Wc;
                    pix = (size_t)(img * a.o_H + 2 * hc + a.o_ph) * a.o_W + 2 * wc + a.o_pw;
                }
                if (PMAP) pix = (size_t)a.perm[!CHECK || m < a.M ? m : a.M - 1];   // tap-sorted rows: scatter to the pixel
                if (!CHECK || m < a.M) st1(yo + pix * a.ldy + n, v);
            }
        }
    }
}

// DBG selects the K-loop form and the probe instrumentation: bit 5 (32) records a per-workgroup timeline (probe only),
// bit 6 (64) = single LDS buffer with two barriers per slice, bit 7 (128) = double-buffered, branch-free body with the
// refill pinned between the MFMAs (128-wide tiles); 0 = double-buffered with two register staging sets (64x64 tile).
// Shared epilogue of the fp32 and bf16-operand kernels.  C/D map of the 32x32 MFMA (dtype independent):
// column = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5).  `smem` must be free (all waves past their last
// LDS read) and hold >= 2*(BN/64)*32*3 floats.
template <int BM, int BN, bool PERM = false, typename TO = float>
__device__ __forceinline__ void igemm_epilogue(const IgemmArgs& a, f32x16 (&acc)[BM / 64][BN / 64], float* smem,
                                               int mt, int m0, int n0, int wm, int wn, int l31, int lh) {
    constexpr int TM = BM / 64, TN = BN / 64;
    const int mrow0 = m0 + wm * (BM / 2) + 4 * lh;
    const int ncol0 = n0 + wn * (BN / 2) + l31;

    if (a.stats) {
        float sc[TN], sm[TN], s2[TN];
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            float cnt = 0.f, sum = 0.f;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    int m = mrow0 + i * 32 + (r & 3) + 8 * (r >> 2);
                    if (m < a.M) {
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
                    int m = mrow0 + i * 32 + (r & 3) + 8 * (r >> 2);
                    if (m < a.M) {
                        float d = acc[i][j][r] - mean;
                        q += d * d;
                    }
                }
            float c2 = __shfl_xor(cnt, 32), m2 = __shfl_xor(mean, 32), q2 = __shfl_xor(q, 32);
            if (lh) {  // both halves must merge in the same order to agree bitwise
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
        // smem is free: the K loop ended with a barrier after the last LDS read
        if (wm == 1 && lh == 0) {
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                float* d = smem + ((wn * TN + j) * 32 + l31) * 3;
                d[0] = sc[j];
                d[1] = sm[j];
                d[2] = s2[j];
            }
        }
        __syncthreads();
        if (wm == 0 && lh == 0) {
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const float* s = smem + ((wn * TN + j) * 32 + l31) * 3;
                wf_merge(sc[j], sm[j], s2[j], s[0], s[1], s[2]);
                int n = ncol0 + j * 32;
                if (n < a.Ng) {
                    float* o = a.stats + ((size_t)mt * a.Ng + n) * 3;   // (sc1: the last arriver of the fold reads them)
                    st_agent(o, sc[j]);
                    st_agent(o + 1, sm[j]);
                    st_agent(o + 2, s2[j]);
                }
            }
        }
    }

    // Stores.  Row-bound checks and the residual test are hoisted into four straight-line instantiations: with
    // either of them inside the store loop the compiler placed an `s_waitcnt vmcnt(0)` in front of EVERY store
    // (its wait-count state is merged conservatively across the predicated blocks), i.e. each of a thread's 64
    // stores waited for the previous one to be acknowledged (measured: 23 us of a 103 us workgroup lifetime).
    const bool full = m0 + BM <= a.M;   // uniform: only the last row tile is ragged
    if constexpr (PERM) {   // (launched without residual / output row map)
        if (full) igemm_store<BM, BN, false, false, false, true>(a, acc, mrow0, ncol0);
        else igemm_store<BM, BN, false, true, false, true>(a, acc, mrow0, ncol0);
        return;
    }
    if (a.residual) {
        if (full) igemm_store<BM, BN, true, false, false, false, TO>(a, acc, mrow0, ncol0);
        else igemm_store<BM, BN, true, true, false, false, TO>(a, acc, mrow0, ncol0);
    } else if (a.o_mode) {
        if (full) igemm_store<BM, BN, false, false, true, false, TO>(a, acc, mrow0, ncol0);
        else igemm_store<BM, BN, false, true, true, false, TO>(a, acc, mrow0, ncol0);
    } else {
        if (full) igemm_store<BM, BN, false, false, false, false, TO>(a, acc, mrow0, ncol0);
        else igemm_store<BM, BN, false, true, false, false, TO>(a, acc, mrow0, ncol0);
    }
}

// MODE 0: generic (K slices may straddle taps / ragged K: stem, 15-channel LSTM convolutions)
// MODE 1: aligned (Cp % 32 == 0), per-slice bounds arithmetic (strided data gradient, > 32 taps)
// MODE 2: aligned + precomputed per-row offset and per-row tap-validity bit mask: one add + one bit test
//         per gathered row and K slice (the address arithmetic of MODE 1 was ~20 % of the kernel time)
// PERM:    GEMM rows are output pixels in tap-sorted order (a.perm), MODE 2 only.
// (Measured and removed in round 2: a persistent stream-K form of this kernel — one wave of workgroups, each walking a
//  contiguous range of (tile, K slice) units and merging cut tiles through agent-scope partials — was 1-3 % SLOWER per step
//  than the tail-split form in every configuration tried, profiles/r01_j_persistent_vs_default_per_shape.txt.)
// (Measured and removed in round 2: the 64x64 short-reduction kernel compiled for 7 / 8 waves per SIMD — 72 / 64 VGPRs with
//  4 / 10 spills outside the K loop — was 0.2 / 0.7 ms per step SLOWER than the natural 78-VGPR build, profiles/r02_a_knob_ab.txt.)
// SWZ:     LDS rows are 32 floats with NO padding; the eight 16-byte chunks of row r sit at chunk ^ ((r >> 1) & 7).  The
//          16-lane groups of a ds_read_b128 fragment read (rows l&31, same logical chunk) then hit 16 distinct 16-byte
//          slots of the 256-byte bank row, like with the 4-float pad, and a 64x64 double-buffered workgroup takes 32 KB
//          instead of 36.9 KB (64x128: 48 instead of 55.3 KB, three per CU instead of two).  Measured -0.55 ms per step.
//          It does NOT get a fifth 64x64 workgroup onto a CU: the LDS allocator hands out 130 granules of 256 B for a
//          32 KB request, 5 x 130 > 640 (probe timeline: the K-split tail parts still start when a whole tile ends).
//          (Double-buffered forms only: the single-buffer loop is register-, not LDS-limited, and the four per-lane chunk
//          offsets cost it an occupancy step.)
template <int BM, int BN, int MODE, int DBG = 0, int KT = 32, bool PERM = false, bool SWZ = false>
__global__ void __launch_bounds__(256, (BM == 128 && BN == 128) ? 2 : 3) igemm_kernel(IgemmArgs a) {
    static_assert(!SWZ || KT == 32, "the XOR swizzle is written for 8 chunks per row");
    static_assert(!PERM || MODE == 2, "tap-sorted rows: aligned fast path only");
    constexpr bool ALIGNED = MODE >= 1, FAST = MODE == 2;
    constexpr int TM = BM / 64, TN = BN / 64;   // 32x32 MFMA tiles per wave along m / n
    constexpr int Q4 = KT / 4;                   // float4 per K slice row
    constexpr int RPP = 256 / Q4;                // rows staged per pass (32 at KT=32, 16 at KT=64)
    constexpr int PA = BM / RPP, PB = BN / RPP;  // staging passes
    constexpr int LDS_LD = SWZ ? KT : KT + 4;    // (KT+4)/4 odd -> conflict-free ds_read_b128; SWZ: XOR swizzle instead
    constexpr int BK = KT;
    static_assert(PA <= 8 && PB <= 8, "okmask holds 8 row bits");
    constexpr bool DB = (DBG & 64) == 0;         // two LDS buffers, ONE barrier per slice, refill interleaved
                                                 // (probe bit 6 selects the older single-buffer loop)
    constexpr int BUF = (BM + BN) * LDS_LD;
    __shared__ __attribute__((aligned(16))) float smem[(DB ? 2 : 1) * BUF];
    float* As = smem;
    float* Bs = smem + BM * LDS_LD;

    long long dbg_w0 = 0;   // probe bit 5: per-block timeline (start / end of K loop / stores drained, 100 MHz ticks + HW ids)
    if ((DBG & 32) && threadIdx.x == 0) dbg_w0 = wall_clock64();
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, lh = lane >> 5;
    int logical, part = 0, tail = 0;
    const bool split = (int)blockIdx.x >= a.full_blocks;
    if (!split) {
        logical = xcd_remap(blockIdx.x, a.full_blocks);
    } else {   // K-split tail tile: a 1/parts share of the K slices of tile full_blocks + tail
        const int j = (int)blockIdx.x - a.full_blocks;
        tail = j / a.parts;
        part = j - tail * a.parts;
        logical = a.full_blocks + tail;
    }
    const int mt = fdiv(logical, a.fNtn);
    const int nt = logical - mt * a.ntn;
    const int m0 = mt * BM, n0 = nt * BN;

    const int lrow = tid / Q4;  // row within a staging pass
    const int kq = tid % Q4;    // which float4 of the K slice
    // LDS chunk of this thread's float4 (staging passes are 32 rows apart: the swizzle term only depends on lrow)
    const int kqs = SWZ ? (kq ^ ((lrow >> 1) & 7)) : kq;

    // per-thread gather bases for its PA rows.  Rows beyond M only need a SAFE address (their results
    // are never stored); taps that fall into the zero padding are zeroed by a select at LDS-store time.
    int hb[PA], wb[PA], ib[PA];      // MODE 0/1: image row/col of tap (0,0) and image base (pixels)
    int roff[PA];                    // MODE 2: element offset of tap (0,0) (may be negative)
    unsigned tmask[PA];              // MODE 2: bit t set <=> tap t of this row reads a real pixel
#pragma unroll
    for (int i = 0; i < PA; ++i) {
        int m = m0 + i * RPP + lrow;
        int mm = m < a.M ? m : a.M - 1;
        if constexpr (PERM) mm = a.perm[mm];
        int img = fdiv(mm, a.fPQ);
        int rem = mm - img * (a.P * a.Q);
        int p = fdiv(rem, a.fQ);
        int q = rem - p * a.Q;
        hb[i] = p * a.mul + a.off0;
        wb[i] = q * a.mul + a.off0w;
        ib[i] = img * a.H * a.W;
        if (FAST) {
            roff[i] = (ib[i] + hb[i] * a.W + wb[i]) * a.ldx;
            unsigned mk = 0;
            for (int t = 0, r = 0, sx = 0; t < a.taps; ++t) {
                int h = hb[i] + r * a.tapstep, w = wb[i] + sx * a.tapstep;
                mk |= (h >= 0 && w >= 0 && h < a.H && w < a.W) ? (1u << t) : 0u;
                if (++sx == a.S) {
                    sx = 0;
                    ++r;
                }
            }
            tmask[i] = mk;
        }
    }
    // Tap skipping (MODE 2, <= 16 taps): a filter tap that reads padding for EVERY row of this tile contributes
    // nothing, so the K loop only visits the taps of the union of the row masks.  On the dilated convolutions of the
    // 23x23 stages most vertical taps die this way (WASP d = 18: 77 % of the MACs multiply zeros; a 64-row tile spans
    // < 3 image rows).  tapmap packs the surviving taps, 4 bits each; slices are numbered over the surviving taps.
    const int spt = ALIGNED ? (int)a.fSpt.d : 1;
    unsigned long long tapmap = 0xfedcba9876543210ull;   // identity
    const bool mapped = FAST && a.taps <= 16;            // 17..32 taps: every tap is visited
    int nk = (a.Ktot + BK - 1) / BK;
    if (mapped && a.taps > 1 && !a.no_tap_skip) {
        unsigned tm = 0;
#pragma unroll
        for (int i = 0; i < PA; ++i) tm |= tmask[i];
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) tm |= __shfl_xor(tm, off);
        unsigned* su = reinterpret_cast<unsigned*>(smem);
        if (lane == 0) su[wave] = tm;
        __syncthreads();
        tm = su[0] | su[1] | su[2] | su[3];
        __syncthreads();   // smem is about to be overwritten by the first slice
        int nvt = 0;
        tapmap = 0;
        for (int t = 0; t < a.taps; ++t)
            if ((tm >> t) & 1u) {
                tapmap |= (unsigned long long)t << (4 * nvt);
                ++nvt;
            }
        if (nvt == 0) nvt = 1;   // (cannot happen for a tile with a real row; keeps the loop well formed)
        nk = nvt * spt;
    }
    int kb = 0, ke = nk;
    if (split) {
        kb = (int)((long long)nk * part / a.parts);
        ke = (int)((long long)nk * (part + 1) / a.parts);
    }

    const float* wrow[PB];
#pragma unroll
    for (int j = 0; j < PB; ++j) {
        int n = n0 + j * RPP + lrow;
        // rows >= Ng are never stored: clamp.  The kq*4 column offset is folded in only when every K slice is
        // full (ALIGNED); the generic path adds it per slice so that a ragged last slice stays inside the row.
        wrow[j] = a.w + (size_t)(n < a.Ng ? n : a.Ng - 1) * a.Ktot + (ALIGNED ? kq * 4 : 0);
    }

    float4 ra[PA], rb[PB];
    unsigned okmask = 0;   // bit i: ra[i] is real data (else structural zero); bit 31: k slice valid
    // second staging set of the non-pinned double-buffered loop (three slices in flight; 64x64 tile: two rows of A and
    // of B per thread).  Named scalars: as arrays the compiler kept this set in scratch memory.
    float4 ya0, ya1, yb0, yb1;
    unsigned okmask2 = 0;

    // aligned path: the 32-wide slice never straddles a tap; (tap, ci0) advance as scalars
    int tap_c = 0, ci0_c = 0;
    if (ALIGNED && kb > 0) {
        tap_c = fdiv(kb, a.fSpt);
        ci0_c = (kb - tap_c * (int)a.fSpt.d) * BK;
    }

    // Branch-free prefetch: every load is issued unconditionally from a clamped (always valid) address; no
    // wait until lstore().  gprep() does the (mostly scalar) per-slice bookkeeping, gissue(part) issues the
    // loads of one quarter of the slice so that the K loop can spread them between its MFMA groups: issuing
    // all of them back-to-back stalls the wave on the texture-address queue while its MFMAs could run.
    int g_tap = 0, g_ci = 0, g_dh = 0, g_dw = 0, g_delta = 0;
    bool g_kvalid = true;
    size_t g_koff = 0;
    unsigned g_msk = 0;
    auto gprep = [&](int kt) {
        g_kvalid = true;
        int ci0 = 0;   // first channel of the slice within its tap (aligned paths)
        if (ALIGNED && (DBG & 128)) {        // stateless: slice index -> (surviving tap, first channel)
            const int vt = fdiv(kt, a.fSpt);
            ci0 = (kt - vt * spt) * BK;
            g_tap = mapped ? (int)((tapmap >> (4 * vt)) & 15ull) : vt;
            g_ci = ci0 + kq * 4;
        } else if (ALIGNED) {
            ci0 = ci0_c;
            g_tap = mapped ? (int)((tapmap >> (4 * tap_c)) & 15ull) : tap_c;
            g_ci = ci0_c + kq * 4;
        } else {
            int k = kt * BK + kq * 4;
            g_kvalid = k < a.Ktot;
            int kk = g_kvalid ? k : 0;
            g_tap = fdiv(kk, a.fCp);
            g_ci = kk - g_tap * a.Cp;
        }
        int r = fdiv(g_tap, a.fS);
        int s = g_tap - r * a.S;
        g_dh = r * a.tapstep;
        g_dw = s * a.tapstep;
        g_delta = (g_dh * a.W + g_dw) * a.ldx + g_ci;
        g_msk = g_kvalid ? 0x80000000u : 0u;
        g_koff = ALIGNED ? (size_t)g_tap * a.Cp + ci0 : (g_kvalid ? (size_t)(kt * BK + kq * 4) : (size_t)0);
        if (ALIGNED) {
            ci0_c += BK;
            if (ci0_c >= a.Cp) {
                ci0_c = 0;
                tap_c += 1;
            }
        }
    };
    auto gloadA = [&](int i) {
        if (FAST) {
            const bool ok = (tmask[i] >> g_tap) & 1u;
            const int off = ok ? roff[i] + g_delta : 0;
            ra[i] = *reinterpret_cast<const float4*>(a.x + off);
            g_msk |= ok ? (1u << i) : 0u;
        } else {
            int h = hb[i] + g_dh, w = wb[i] + g_dw;
            bool ok = g_kvalid && h >= 0 && w >= 0;
            // data gradient of a stride-2 convolution: only even source offsets hit a real dy sample
            ok = ok && !((h | w) & a.divmask);
            h >>= a.divshift;
            w >>= a.divshift;
            ok = ok && h < a.H && w < a.W;
            size_t off = ok ? (size_t)(ib[i] + h * a.W + w) * a.ldx + g_ci : (size_t)0;
            ra[i] = *reinterpret_cast<const float4*>(a.x + off);
            g_msk |= ok ? (1u << i) : 0u;
        }
    };
    auto gloadB = [&](int j) { rb[j] = *reinterpret_cast<const float4*>(wrow[j] + g_koff); };
    // quarter `part` (0..3) of the slice: PA/4 (or all in part < PA) A rows and likewise B rows
    auto gissue = [&](int part) {
#pragma unroll
        for (int i = 0; i < PA; ++i)
            if ((PA >= 4 ? i * 4 / PA : i) == part) gloadA(i);
#pragma unroll
        for (int j = 0; j < PB; ++j)
            if ((PB >= 4 ? j * 4 / PB : j) == part) gloadB(j);
        if (part == 3) okmask = g_msk;
    };
    auto gload = [&](int kt) {
        gprep(kt);
#pragma unroll
        for (int part = 0; part < 4; ++part) gissue(part);
    };
    auto lstore = [&](int buf = 0) {
#pragma unroll
        for (int i = 0; i < PA; ++i)
            *reinterpret_cast<float4*>(&As[buf * BUF + (i * RPP + lrow) * LDS_LD + kqs * 4]) =
                keep_or_zero((okmask >> i) & 1u, ra[i]);
#pragma unroll
        for (int j = 0; j < PB; ++j)
            *reinterpret_cast<float4*>(&Bs[buf * BUF + (j * RPP + lrow) * LDS_LD + kqs * 4]) =
                ALIGNED ? rb[j] : keep_or_zero(okmask >> 31, rb[j]);
    };
    // the same for the second staging set
    auto gloadY = [&](int i, float4& dst) {
        const bool ok = (tmask[i] >> g_tap) & 1u;
        const int off = ok ? roff[i] + g_delta : 0;
        dst = *reinterpret_cast<const float4*>(a.x + off);
        g_msk |= ok ? (1u << i) : 0u;
    };
    auto gissue2 = [&](int part) {
        if (part == 0) {
            gloadY(0, ya0);
            yb0 = *reinterpret_cast<const float4*>(wrow[0] + g_koff);
        }
        if (part == 1) {
            gloadY(PA - 1, ya1);
            yb1 = *reinterpret_cast<const float4*>(wrow[PB - 1] + g_koff);
        }
        if (part == 3) okmask2 = g_msk;
    };
    auto gload2 = [&](int kt) {
        gprep(kt);
#pragma unroll
        for (int part = 0; part < 4; ++part) gissue2(part);
    };
    auto lstore2 = [&](int buf) {
        *reinterpret_cast<float4*>(&As[buf * BUF + lrow * LDS_LD + kqs * 4]) = keep_or_zero(okmask2 & 1u, ya0);
        *reinterpret_cast<float4*>(&As[buf * BUF + ((PA - 1) * RPP + lrow) * LDS_LD + kqs * 4]) =
            keep_or_zero((okmask2 >> (PA - 1)) & 1u, ya1);
        *reinterpret_cast<float4*>(&Bs[buf * BUF + lrow * LDS_LD + kqs * 4]) = yb0;
        *reinterpret_cast<float4*>(&Bs[buf * BUF + ((PB - 1) * RPP + lrow) * LDS_LD + kqs * 4]) = yb1;
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    gload(kb);
    lstore();
    __syncthreads();

    // fragment of k-group g: logical chunk 2g + lh of row (32-multiple + l31); SWZ: chunk ^ ((l31 >> 1) & 7)
    const float* Ard = As + (wm * (BM / 2) + l31) * LDS_LD + (SWZ ? 0 : lh * 4);
    const float* Brd = Bs + (wn * (BN / 2) + l31) * LDS_LD + (SWZ ? 0 : lh * 4);
    const int swz = (l31 >> 1) & 7;
    auto fcol = [&](int g) { return SWZ ? (((g * 2 + lh) ^ swz) * 4) : g * 8; };   // float offset inside the row

    // (Measured and removed in round 2: a second accumulator for the odd k pairs of the 32x32-per-wave tile, so that
    //  consecutive MFMAs never form one dependent chain — the K loop of a workgroup that is alone on its CU stayed at
    //  53.7 us for 30.7 us of MFMA work (tools/gpu/igemm_probe alone).  What a lone wave loses per slice is the time of the
    //  ~100 non-MFMA instructions the scheduler places in blocks between its 4-MFMA groups)
    auto mfma_group = [&](const float4(&af)[TM], const float4(&bf)[TN]) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].x, bf[j].x, acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].y, bf[j].y, acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].z, bf[j].z, acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].w, bf[j].w, acc[i][j], 0, 0, 0);
            }
    };
    constexpr int G = BK / 8;   // MFMA groups (8 k each) per slice

    if constexpr (DB) {
        // slice kt is in LDS buffer kt&1; slice kt+1 sits in the staging registers (loaded during slice kt-1)
        // and is written to the OTHER buffer in the middle of this slice's MFMAs; the registers are then
        // refilled with slice kt+2.  One barrier per slice, nothing between the MFMAs but LDS/VMEM issue.
        if (ke - kb > 1) gload(kb + 1);
        if constexpr ((DBG & 128) != 0) {
            // Branch-free body: the refill always runs (slice indices are clamped to the last slice, whose
            // reload is harmless), so the whole iteration is ONE scheduling region and the non-MFMA work can be
            // pinned between the MFMAs with sched_group_barrier instead of piling up between 16-MFMA clusters.
            if (ke - kb == 1) gload(kb);
            for (int kt = kb; kt < ke; ++kt) {
                const int cur = (kt - kb) & 1;
                const int k2 = kt + 2 < ke ? kt + 2 : ke - 1;
                float4 af[2][TM], bf[2][TN];
                auto frag = [&](int g, int b) {
#pragma unroll
                    for (int i = 0; i < TM; ++i)
                        af[b][i] = *reinterpret_cast<const float4*>(Ard + cur * BUF + i * 32 * LDS_LD + fcol(g));
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        bf[b][j] = *reinterpret_cast<const float4*>(Brd + cur * BUF + j * 32 * LDS_LD + fcol(g));
                };
                frag(0, 0);
#pragma unroll
                for (int g = 0; g < G; ++g) {
                    const int b = g & 1;
                    if (g + 1 < G) frag(g + 1, b ^ 1);
                    if (g == G / 2 - 1) lstore(cur ^ 1);
                    if (g >= G / 2) {
                        if (g == G / 2) gprep(k2);
#pragma unroll
                        for (int q = 0; q < 8 / G; ++q) gissue((g - G / 2) * (8 / G) + q);
                    }
                    mfma_group(af[b], bf[b]);
                }
#pragma unroll
                for (int q = 0; q < G * 4 * TM * TN; ++q) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // one MFMA ...
                    __builtin_amdgcn_sched_group_barrier(0x080, 1, 0);   // ... then one LDS access,
                    __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);   // a few VALU
                    __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);   // and one global load
                }
                __syncthreads();
            }
        } else {
        // Two staging sets: at the top of slice kt set X holds slice kt+1 and set Y slice kt+2 (X and Y swap roles every
        // slice: the loop is unrolled by two).  Mid-slice the set holding kt+1 goes to the other LDS buffer and is
        // refilled with slice kt+3, so a load has two full slices to land instead of one (FAST path only).
        static_assert(PA == 2 && PB == 2, "the two-set loop is written for the 64x64 tile");
        if (ke - kb > 2) gload2(kb + 2);
        // EARLYBAR (DBG bit 8, probe only): the slice's barrier sits BEFORE its last MFMA group and is followed at once by the
        // fragment reads of the NEXT slice's first group, so a wave never starts a slice by waiting for LDS.  A workgroup
        // alone on its CU gains 10 % (K loop 53.9 -> 48.4 us, tools/gpu/igemm_probe alone), a 1024-tile launch 5 % in the
        // probe — and the network LOSES 1.2-1.4 ms per step with it in both stream modes, forward alone +3.4 % (r02_j):
        // kept out of the library, kept here as the record of the experiment.
        constexpr bool EARLYBAR = (DBG & 256) != 0;
        float4 af[2][TM], bf[2][TN];
        auto fragc = [&](int cur, int g, int b) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
                af[b][i] = *reinterpret_cast<const float4*>(Ard + cur * BUF + i * 32 * LDS_LD + fcol(g));
#pragma unroll
            for (int j = 0; j < TN; ++j)
                bf[b][j] = *reinterpret_cast<const float4*>(Brd + cur * BUF + j * 32 * LDS_LD + fcol(g));
        };
        if constexpr (EARLYBAR) fragc(0, 0, 0);
        auto slice = [&](int kt, auto cur_c, auto use_x_c) {   // compile-time buffer / staging-set choice
            constexpr int cur = decltype(cur_c)::value;
            constexpr bool useX = decltype(use_x_c)::value;
            const bool has1 = kt + 1 < ke, has3 = kt + 3 < ke;
            if constexpr (!EARLYBAR) fragc(cur, 0, 0);
#pragma unroll
            for (int g = 0; g < G; ++g) {
                const int b = g & 1;
                if (g + 1 < G) fragc(cur, g + 1, b ^ 1);
                if (g == G / 2 - 1 && has1) {
                    if constexpr (useX) lstore(cur ^ 1);
                    else lstore2(cur ^ 1);
                }
                if (g >= G / 2 && has3) {
                    if (g == G / 2) gprep(kt + 3);
#pragma unroll
                    for (int q = 0; q < 8 / G; ++q) {
                        if constexpr (useX) gissue((g - G / 2) * (8 / G) + q);
                        else gissue2((g - G / 2) * (8 / G) + q);
                    }
                }
                if constexpr (EARLYBAR) {
                    if (g == G - 1) {
                        __syncthreads();
                        fragc(cur ^ 1, 0, b ^ 1);   // (the last slice reads a buffer nobody stored: the values are never used)
                    }
                }
                mfma_group(af[b], bf[b]);
            }
            if constexpr (!EARLYBAR) __syncthreads();
        };
        for (int kt = kb; kt < ke; kt += 2) {
            slice(kt, std::integral_constant<int, 0>{}, std::true_type{});
            if (kt + 1 < ke) slice(kt + 1, std::integral_constant<int, 1>{}, std::false_type{});
        }
        }
    } else {
    for (int kt = kb; kt < ke; ++kt) {
        const bool more = kt + 1 < ke;
        if (more) gprep(kt + 1);
        // fragments of k-group g+1 are read from LDS while the MFMAs of group g run (static double buffer)
        float4 af[2][TM], bf[2][TN];
        auto frag = [&](int g, int b) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
                af[b][i] = *reinterpret_cast<const float4*>(Ard + i * 32 * LDS_LD + fcol(g));
#pragma unroll
            for (int j = 0; j < TN; ++j)
                bf[b][j] = *reinterpret_cast<const float4*>(Brd + j * 32 * LDS_LD + fcol(g));
        };
        frag(0, 0);
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const int b = g & 1;
            if (g + 1 < G) frag(g + 1, b ^ 1);
            if (more && g % (BK / 32) == 0) gissue(g / (BK / 32));   // a quarter of the next slice's loads
            mfma_group(af[b], bf[b]);
        }
        __syncthreads();
        if (more) {
            lstore();
            __syncthreads();
        }
    }
    }

    long long dbg_w1 = 0;
    auto dbg_record = [&]() {   // probe bit 5
#ifndef UP_EMU
        __builtin_amdgcn_s_waitcnt(0);
        if (threadIdx.x == 0) {
            long long* o = reinterpret_cast<long long*>(a.dbg) + 4 * (size_t)blockIdx.x;
            o[0] = dbg_w0;
            o[1] = dbg_w1;
            o[2] = wall_clock64();
            o[3] = (long long)__builtin_amdgcn_s_getreg((31 << 11) | 4) |                 // HW_REG_HW_ID
                   ((long long)(__builtin_amdgcn_s_getreg((31 << 11) | 20) & 15) << 32) | // HW_REG_XCC_ID
                   ((long long)(__builtin_amdgcn_s_getreg((31 << 11) | 6) & 0x1fffff) << 40);   // HW_REG_LDS_ALLOC
        }
#endif
    };
    if (DBG & 32) dbg_w1 = wall_clock64();
    if (split) {
        // Partials are [part][(i*TN+j)*16 + r][256 threads] floats: every access is one coalesced 256-B row per wave.
        // They are written and read with agent-scope accesses (write-through / cache-bypassing on gfx950), so the
        // flag needs no L2 write-back fence; readers have a higher block index than writers (no dispatch deadlock).
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
            if (DBG & 32) dbg_record();
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

    igemm_epilogue<BM, BN, PERM>(a, acc, smem, mt, m0, n0, wm, wn, l31, lh);
    igemm_fold_arrive<BN>(a, mt, n0, reinterpret_cast<unsigned char*>(smem));
    if (DBG & 32) dbg_record();
}

// ------------------------------------------------------------------------------------------
// bf16-operand variants of the same implicit GEMM on v_mfma_f32_32x32x16_bf16 (16x the fp32 MFMA rate)
//   SPLIT = true : "split-bf16" fp32-equivalent arithmetic.  Every fp32 operand x is carried as hi = bf16(x),
//                  lo = bf16(x - hi) (16 mantissa bits together) and a*b ~= ah*bh + ah*bl + al*bh with fp32
//                  accumulation: relative error per product <= 2^-16, three MFMAs instead of eight fp32 ones.
//   SPLIT = false: plain bf16 operands, fp32 accumulation (BASELINE config 5 arithmetic).
// HBM data stays fp32 (activations) / pre-split bf16 planes (weights, made by up_pack_weights_bf16); the
// activation operand is split while it is staged to LDS.  K slice = 64, LDS rows are 128 B + 16 B pad
// (144/16 = 9 odd: the 16-B fragment reads of 16 lanes hit 16 distinct slots).  A fragment of the MFMA is 8
// consecutive k of one row: lane l -> row l&31, k = 8*(l>>5)..+7 of each 16-wide step.
// ------------------------------------------------------------------------------------------
#ifndef UP_EMU
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#endif
// (HIP's native uint2 / uint4 vector types: hand-made structs moved through reinterpret_cast stayed in scratch)
// (hi, lo) bf16 pairs of two floats; element 0 in the low half-word
__device__ __forceinline__ void split_bf16x2(float a0, float a1, uint32_t& hi, uint32_t& lo) {
    hi = pack_bf16x2(a0, a1);
    float h0 = __uint_as_float(hi << 16), h1 = __uint_as_float(hi & 0xffff0000u);
    lo = pack_bf16x2(a0 - h0, a1 - h1);
}

// Pipeline: K slice = 32.  Slice t is read from LDS buffer t&1 while slice t+1 (held in one of TWO register
// staging sets) is converted and written to the other buffer and slice t+2 is still in flight in the second set;
// the freed set is refilled with slice t+3.  Global loads therefore have two full slices (~2 x 768 MFMA cycles x
// the blocks sharing the SIMD) to land — the bf16 MFMA phase alone (768 cycles) is shorter than the memory
// latency, which made the first single-stage version of this kernel latency-bound.  One barrier per slice.
// Slice indices past the end are clamped (harmless reloads): the loop body is branch-free.
// HS: bf16 STORAGE (BASELINE configs[4]): the activation operand, the residual / addend and the output are bf16 in HBM.  A
// thread stages 16 bytes = 8 channels per row with no conversion at all (the fp32 form: 4 channels + two packs).
// OF32 (with HS): the OUTPUT is fp32 — the network's last convolution in bf16 storage, whose heat-maps leave the library as fp32
// (no residual: it would be read with the output's element type).
template <int BM, int BN, int MODE, bool SPLIT, int KT = 32, bool HS = false, bool OF32 = false>
__global__ void __launch_bounds__(256, 2) igemm_bf16_kernel(IgemmArgs a) {
    static_assert(!OF32 || HS, "OF32 is a bf16-storage form");
    static_assert(MODE == 1 || MODE == 2, "bf16 kernels need channel counts that are multiples of the K slice");
    static_assert(!(HS && SPLIT), "split-bf16 is an fp32-storage arithmetic");
    constexpr bool FAST = MODE == 2;
    constexpr int EA = HS ? 8 : 4;                   // activation elements per 16-byte staging access
    constexpr int TM = BM / 64, TN = BN / 64;
    constexpr int NP = SPLIT ? 2 : 1;
    // LDS row stride in bytes: 80 / 144 = slice + 16 B pad, (RS/16) odd -> the ds_read_b128 of 16 consecutive rows are conflict-free.
    // bf16 storage stages whole 16-byte chunks (4 lanes per 64-byte row), and with the pad its ds_write_b128 hit a 2-way
    // conflict on 3 of every 16 lanes (SQ_LDS_BANK_CONFLICT = 0.34 of the LDS-active cycles, profiles/r02_ac_*) in a kernel
    // whose LDS time exceeds its MFMA time.  SWZB: unpadded 64-byte rows, chunk c of row r at c ^ ((r >> 2) & 3): reads of
    // 16 consecutive rows and writes of 4 rows x 4 chunks both touch 16 distinct 16-byte slots of the 256-byte bank row,
    // and the tile takes 32 KB instead of 40.
    constexpr bool SWZB = HS && KT == 32;
    constexpr int RS = SWZB ? KT * 2 : KT * 2 + 16;
    constexpr int A_PLANE = BM * RS, B_PLANE = BN * RS;
    constexpr int BUF = NP * (A_PLANE + B_PLANE);
    constexpr int Q4 = KT / EA, RPA = 256 / Q4;     // A staging: RPA rows x Q4 16-byte chunks per pass
    constexpr int C8 = KT / 8, RPB = 256 / C8;      // B staging: RPB rows x C8 chunks (8 bf16) per pass and plane
    constexpr int PA = BM / RPA;
    constexpr int PB = BN / RPB;
    static_assert(PA <= 8, "okmask holds 8 row bits");
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * BUF];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, lh = lane >> 5;

    const int logical = xcd_remap(blockIdx.x, a.nwg);
    const int mt = fdiv(logical, a.fNtn);
    const int nt = logical - mt * a.ntn;
    const int m0 = mt * BM, n0 = nt * BN;

    const int lrow = tid / Q4, kq = tid % Q4;       // A: row within pass, float4 within the K slice
    const int brow = tid / C8, bch = tid % C8;      // B: row within pass, 16-byte chunk (8 bf16) within the slice

    int hb[PA], wb[PA], ib[PA], roff[PA];
    unsigned tmask[PA];
#pragma unroll
    for (int i = 0; i < PA; ++i) {
        int m = m0 + i * RPA + lrow;
        int mm = m < a.M ? m : a.M - 1;
        int img = fdiv(mm, a.fPQ);
        int rem = mm - img * (a.P * a.Q);
        int p = fdiv(rem, a.fQ);
        int q = rem - p * a.Q;
        hb[i] = p * a.mul + a.off0;
        wb[i] = q * a.mul + a.off0;
        ib[i] = img * a.H * a.W;
        if (FAST) {
            roff[i] = (ib[i] + hb[i] * a.W + wb[i]) * a.ldx;
            unsigned mk = 0;
            for (int t = 0, r = 0, sx = 0; t < a.taps; ++t) {
                int h = hb[i] + r * a.tapstep, w = wb[i] + sx * a.tapstep;
                mk |= (h >= 0 && w >= 0 && h < a.H && w < a.W) ? (1u << t) : 0u;
                if (++sx == a.S) {
                    sx = 0;
                    ++r;
                }
            }
            tmask[i] = mk;
        }
    }
    size_t wofs[PB];   // bf16-element offset of this thread's chunk in each staged weight row
#pragma unroll
    for (int j = 0; j < PB; ++j) {
        int n = n0 + j * RPB + brow;
        wofs[j] = (size_t)(n < a.Ng ? n : a.Ng - 1) * a.Ktot + bch * 8;
    }

    const int nk = a.Ktot / KT;
    const int spt = a.Cp / KT;                      // K slices per filter tap
    const FastDiv fspt = a.fSpt;

    // two register staging sets (16 bytes per row and thread: four fp32 or eight bf16 channels)
    uint4 raX[PA], raY[PA];
    // B staging registers are named scalars (PB <= 2): as uint4 arrays one set stayed in scratch memory
    static_assert(PB <= 2, "two weight rows per thread at most");
    uint4 rbhX0, rbhX1, rblX0, rblX1, rbhY0, rbhY1, rblY0, rblY1;
    rbhX0 = rbhX1 = rblX0 = rblX1 = rbhY0 = rbhY1 = rblY0 = rblY1 = make_uint4(0u, 0u, 0u, 0u);
    unsigned okX = 0, okY = 0;

    // The two staging sets are addressed by NAME (macro-generated lambdas capturing the arrays directly): passing
    // the arrays to one generic lambda by reference kept them in scratch memory.
#define UP_BF16_STAGE(SFX)                                                                                           \
    auto gload##SFX = [&](int kt_req) {                                                                              \
        const int kt = kt_req < nk ? kt_req : nk - 1;                                                                \
        const int tap = fdiv(kt, fspt);                                                                              \
        const int ci = (kt - tap * spt) * KT + kq * EA;                                                              \
        int r = fdiv(tap, a.fS);                                                                                     \
        int sx = tap - r * a.S;                                                                                      \
        int dh = r * a.tapstep, dw = sx * a.tapstep;                                                                 \
        unsigned msk = 0;                                                                                            \
        if (FAST) {                                                                                                  \
            const int delta = (dh * a.W + dw) * a.ldx + ci;                                                          \
            _Pragma("unroll") for (int i = 0; i < PA; ++i) {                                                         \
                const bool ok = (tmask[i] >> tap) & 1u;                                                              \
                const int off = ok ? roff[i] + delta : 0;                                                            \
                ra##SFX[i] = HS ? *reinterpret_cast<const uint4*>(reinterpret_cast<const bf16_t*>(a.x) + off)        \
                                : *reinterpret_cast<const uint4*>(a.x + off);                                        \
                msk |= ok ? (1u << i) : 0u;                                                                          \
            }                                                                                                        \
        } else {                                                                                                     \
            _Pragma("unroll") for (int i = 0; i < PA; ++i) {                                                         \
                int h = hb[i] + dh, w = wb[i] + dw;                                                                  \
                bool ok = h >= 0 && w >= 0 && !((h | w) & a.divmask);                                                \
                h >>= a.divshift;                                                                                    \
                w >>= a.divshift;                                                                                    \
                ok = ok && h < a.H && w < a.W;                                                                       \
                size_t off = ok ? (size_t)(ib[i] + h * a.W + w) * a.ldx + ci : (size_t)0;                            \
                ra##SFX[i] = HS ? *reinterpret_cast<const uint4*>(reinterpret_cast<const bf16_t*>(a.x) + off)        \
                                : *reinterpret_cast<const uint4*>(a.x + off);                                        \
                msk |= ok ? (1u << i) : 0u;                                                                          \
            }                                                                                                        \
        }                                                                                                            \
        const size_t koff = (size_t)kt * KT;                                                                         \
        rbh##SFX##0 = *reinterpret_cast<const uint4*>(a.w_hi + wofs[0] + koff);                                      \
        if (SPLIT) rbl##SFX##0 = *reinterpret_cast<const uint4*>(a.w_lo + wofs[0] + koff);                           \
        if (PB > 1) {                                                                                                \
            rbh##SFX##1 = *reinterpret_cast<const uint4*>(a.w_hi + wofs[PB - 1] + koff);                             \
            if (SPLIT) rbl##SFX##1 = *reinterpret_cast<const uint4*>(a.w_lo + wofs[PB - 1] + koff);                  \
        }                                                                                                            \
        ok##SFX = msk;                                                                                               \
    };                                                                                                               \
    auto lstore##SFX = [&](int buf) {                                                                                \
        unsigned char* As = smem + buf * BUF;                                                                        \
        unsigned char* Bs = As + NP * A_PLANE;                                                                       \
        _Pragma("unroll") for (int i = 0; i < PA; ++i) {                                                             \
            const bool keep = (ok##SFX >> i) & 1u;                                                                   \
            const uint4 u = ra##SFX[i];                                                                              \
            if constexpr (HS) {                                                                                      \
                *reinterpret_cast<uint4*>(As + (i * RPA + lrow) * RS + (SWZB ? (kq ^ ((lrow >> 2) & 3)) : kq) * 16) =       \
                    make_uint4(keep ? u.x : 0u, keep ? u.y : 0u, keep ? u.z : 0u, keep ? u.w : 0u);                  \
            } else {                                                                                                 \
                const float4 v = keep_or_zero(keep, make_float4(__uint_as_float(u.x), __uint_as_float(u.y),          \
                                                                __uint_as_float(u.z), __uint_as_float(u.w)));        \
                uint2 hi, lo;                                                                                        \
                if (SPLIT) {                                                                                         \
                    split_bf16x2(v.x, v.y, hi.x, lo.x);                                                              \
                    split_bf16x2(v.z, v.w, hi.y, lo.y);                                                              \
                } else {                                                                                             \
                    hi.x = pack_bf16x2(v.x, v.y);                                                                    \
                    hi.y = pack_bf16x2(v.z, v.w);                                                                    \
                }                                                                                                    \
                unsigned char* d = As + (i * RPA + lrow) * RS + kq * 8;                                              \
                *reinterpret_cast<uint2*>(d) = hi;                                                                   \
                if (SPLIT) *reinterpret_cast<uint2*>(d + A_PLANE) = lo;                                              \
            }                                                                                                        \
        }                                                                                                            \
        {                                                                                                            \
            unsigned char* d = Bs + brow * RS + (SWZB ? (bch ^ ((brow >> 2) & 3)) : bch) * 16;                       \
            *reinterpret_cast<uint4*>(d) = rbh##SFX##0;                                                              \
            if (SPLIT) *reinterpret_cast<uint4*>(d + B_PLANE) = rbl##SFX##0;                                         \
            if (PB > 1) {                                                                                            \
                *reinterpret_cast<uint4*>(d + RPB * RS) = rbh##SFX##1;                                               \
                if (SPLIT) *reinterpret_cast<uint4*>(d + RPB * RS + B_PLANE) = rbl##SFX##1;                          \
            }                                                                                                        \
        }                                                                                                            \
    };
    UP_BF16_STAGE(X)
    UP_BF16_STAGE(Y)
#undef UP_BF16_STAGE

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // fragment of 16-wide k step s: logical chunk lh + 2s of row (32-multiple + l31); SWZB: chunk ^ ((l31 >> 2) & 3)
    const int frag_c0 = SWZB ? ((lh ^ ((l31 >> 2) & 3)) * 16) : lh * 16;
    const int a_rd = (wm * (BM / 2) + l31) * RS;
    const int b_rd = NP * A_PLANE + (wn * (BN / 2) + l31) * RS;
    auto fcol = [&](int s) { return SWZB ? (frag_c0 ^ (s * 32)) : frag_c0 + s * 32; };   // byte offset inside the row

    // one K slice: MFMAs from LDS buffer kt&1; in between, slice kt+1 (register set S) goes to the other buffer
    // and S is refilled with slice kt+3
    // one K slice: MFMAs from LDS buffer kt&1; in between, slice kt+1 (register set SFX) goes to the other buffer
    // and the set is refilled with slice kt+3.  (Generated per set: closures passed as arguments also forced the
    // staging arrays into scratch.)
#define UP_BF16_BODY(SFX)                                                                                             \
    auto body##SFX = [&](int kt) {                                                                                    \
        const unsigned char* base = smem + (kt & 1) * BUF;                                                            \
        _Pragma("unroll") for (int s = 0; s < KT / 16; ++s) {                                                         \
            bf16x8 ah[TM], al[TM], bh[TN], bl[TN];                                                                    \
            _Pragma("unroll") for (int i = 0; i < TM; ++i) {                                                          \
                ah[i] = *reinterpret_cast<const bf16x8*>(base + a_rd + i * 32 * RS + fcol(s));                        \
                if (SPLIT) al[i] = *reinterpret_cast<const bf16x8*>(base + a_rd + A_PLANE + i * 32 * RS + fcol(s));   \
            }                                                                                                         \
            _Pragma("unroll") for (int j = 0; j < TN; ++j) {                                                          \
                bh[j] = *reinterpret_cast<const bf16x8*>(base + b_rd + j * 32 * RS + fcol(s));                        \
                if (SPLIT) bl[j] = *reinterpret_cast<const bf16x8*>(base + b_rd + B_PLANE + j * 32 * RS + fcol(s));   \
            }                                                                                                         \
            if (s == 0) lstore##SFX((kt & 1) ^ 1);                                                                    \
            if (s == KT / 16 - 1) gload##SFX(kt + 3);                                                                 \
            _Pragma("unroll") for (int i = 0; i < TM; ++i) _Pragma("unroll") for (int j = 0; j < TN; ++j) {           \
                if (SPLIT) { /* small cross terms first, the dominant product last */                                 \
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);            \
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl[j], acc[i][j], 0, 0, 0);            \
                }                                                                                                     \
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);                \
            }                                                                                                         \
        }                                                                                                             \
        __syncthreads();                                                                                              \
    };
    UP_BF16_BODY(X)
    UP_BF16_BODY(Y)
#undef UP_BF16_BODY

    gloadX(0);
    lstoreX(0);
    gloadX(1);
    gloadY(2);
    __syncthreads();
    for (int kt = 0; kt < nk; kt += 2) {
        bodyX(kt);
        if (kt + 1 < nk) bodyY(kt + 1);
    }
    igemm_epilogue<BM, BN, false, std::conditional_t<HS && !OF32, bf16_t, float>>(a, acc, reinterpret_cast<float*>(smem), mt, m0, n0,
                                                                                  wm, wn, l31, lh);
    igemm_fold_arrive<BN>(a, mt, n0, reinterpret_cast<unsigned char*>(smem));
}

// OIHW fp32 -> bf16 hi / lo planes in the [rows][tap][channel] order of pack_fwd / pack_dgrad
__global__ void __launch_bounds__(256) pack_split_kernel(const float* w, uint16_t* hi, uint16_t* lo, int K, int C,
                                                         int inner_pad, int taps, long long total, int dgrad) {
    long long e = (long long)blockIdx.x * 256 + threadIdx.x;
    if (e >= total) return;
    int in = (int)(e % inner_pad);
    long long t = e / inner_pad;
    int tap = (int)(t % taps);
    int row = (int)(t / taps);
    float v = 0.f;
    if (!dgrad) {
        if (in < C) v = w[((size_t)row * C + in) * taps + tap];          // row = k, in = c
    } else {
        if (in < K) v = w[((size_t)in * C + row) * taps + tap];          // row = c, in = k
    }
    uint32_t h, l;
    split_bf16x2(v, 0.f, h, l);
    hi[e] = (uint16_t)(h & 0xffffu);
    lo[e] = (uint16_t)(l & 0xffffu);
}

// ------------------------------------------------------------------------------------------
// weight gradient
// ------------------------------------------------------------------------------------------
struct WgradArgs {
    const float* x;
    const float* dy;
    float* slab;
    int M, K, Ncols, Cp;
    int H, W, P, Q;
    int ldx, ldy;
    int S;
    int stride, pad, dil;
    int rows_per_split;
    int ntn, nwg;
    FastDiv fPQ, fQ, fCp, fS, fNtn, fTiles;
    void* dbg;   // development probes only
    // RECT form: per column tile 12 ints {p_lo, q_lo, rows, cols, FastDiv(rows*cols), FastDiv(cols), rows per split,
    // pixels} — the rectangle of output pixels (per image) on which ANY filter tap of the tile reads a real input pixel
    const int* rect;
    uint32_t x_bytes, dy_bytes;   // bf16s_glds.h: bytes behind x / dy (num_records of their buffer descriptors)
};
constexpr int WGRAD_RECT_INTS = 12;

// DBG bit 5: per-block timeline (probe); bit 6: the older single-buffer loop (two barriers per slice)
// RECT: the reduction of a column tile runs over the live rectangle of its filter taps only (see WgradRect) instead of
//       all N*P*Q output pixels: the pixels outside multiply structural zeros (77 % of them on the dilation-18 branch).
// (Measured and removed in round 2: 16-pixel K slices in the double-buffered 128x128 form — 32 KB of LDS instead of 64 KB, to
//  let one weight-gradient workgroup share a CU with four data-gradient workgroups of the other stream — +0.5 ms per step.)
template <int BM, int BN, int DBG = 0, bool RECT = false>
__global__ void __launch_bounds__(256, (BM == 128 && BN == 128) ? 2 : 3) wgrad_kernel(WgradArgs a) {
    constexpr int TM = BM / 64, TN = BN / 64;
    constexpr int A4 = BM / 4, B4 = BN / 4;                // float4 per k row
    constexpr int PA = BK * A4 / 256, PB = BK * B4 / 256;  // staging passes
    constexpr int KA = 256 / A4, KB = 256 / B4;            // k rows covered per pass
    constexpr bool DB = (DBG & 64) == 0;                   // two LDS buffers, ONE barrier per slice
    constexpr int BUF = BK * (BM + BN);
    __shared__ __attribute__((aligned(16))) float smem[(DB ? 2 : 1) * BUF];
    float* As = smem;
    float* Bs = smem + BK * BM;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, lh = lane >> 5;

    long long dbg_w0 = 0;
    if ((DBG & 32) && threadIdx.x == 0) dbg_w0 = wall_clock64();
    // 1-D grid of tiles x splits.  Hardware block b runs on XCD b%8: the bijective remap gives every XCD a contiguous
    // run of (split, tile) pairs, i.e. whole splits, so the pixel rows of a split are fetched into ONE L2 instead of
    // all eight (the 2-D grid spread the tiles of a split over every XCD: 139 MB fetched per launch for 35 MB of operands)
    const int logical = xcd_remap(blockIdx.x, a.nwg);
    const int split = fdiv(logical, a.fTiles);
    const int tile = logical - split * (int)a.fTiles.d;
    const int mt = fdiv(tile, a.fNtn);
    const int nt = tile - mt * a.ntn;
    const int co0 = mt * BM, col0 = nt * BN;
    int mbeg = split * a.rows_per_split;
    int mend = min(a.M, mbeg + a.rows_per_split);
    // RECT: reduction index m enumerates (image, row, column) of the tile's live rectangle
    int r_pl = 0, r_ql = 0, r_wr = 0, r_hw = 0;
    FastDiv r_fhw = {0, 0, 0}, r_fw = {0, 0, 0};
    if constexpr (RECT) {
        const int* rc = a.rect + WGRAD_RECT_INTS * nt;
        r_pl = rc[0];
        r_ql = rc[1];
        r_wr = rc[3];
        r_hw = rc[2] * rc[3];
        r_fhw = FastDiv{(uint32_t)rc[4], (uint32_t)rc[5], (uint32_t)rc[6]};
        r_fw = FastDiv{(uint32_t)rc[7], (uint32_t)rc[8], (uint32_t)rc[9]};
        mbeg = split * rc[10];
        mend = min(rc[11], mbeg + rc[10]);
    }
    // (image, output row, output column) of reduction index mm; RECT only
    auto rect_pixel = [&](int mm, int& img, int& pp, int& qq) {
        img = fdiv(mm, r_fhw);
        const int rem = mm - img * r_hw;
        const int i = fdiv(rem, r_fw);
        pp = r_pl + i;
        qq = r_ql + (rem - i * r_wr);
    };

    // A (dY): thread -> (k row within pass, 4 output channels).  Channels >= K are never stored: they only
    // need a valid address.
    const int a_c4 = tid % A4, a_kr = tid / A4;
    const int a_co = (co0 + a_c4 * 4 < a.ldy) ? co0 + a_c4 * 4 : 0;
    // B (X gather): thread -> (k row within pass, 4 columns = one tap, 4 channels); columns >= Ncols likewise
    const int b_c4 = tid % B4, b_kr = tid / B4;
    const int b_col = (col0 + b_c4 * 4 < a.Ncols) ? col0 + b_c4 * 4 : 0;
    const int b_tap = fdiv(b_col, a.fCp);
    const int b_ci = b_col - b_tap * a.Cp;
    const int b_r = fdiv(b_tap, a.fS);
    const int b_dh = b_r * a.dil - a.pad, b_dw = (b_tap - b_r * a.S) * a.dil - a.pad;

    float4 ra[PA], rb[PB];
    unsigned okmask = 0;   // bits 0..PA-1: dY row inside [mbeg,mend); bits 8..8+PB-1: tap inside the image

    // branch-free prefetch (see igemm_kernel): clamped addresses, structural zeros selected when the registers are
    // written to LDS.  One "pass" = one float4 per thread (KA resp. KB pixel rows of the slice).
    auto gloadA = [&](int p, int kbase) {
        int m = kbase + p * KA + a_kr;
        bool ok = m < mend;
        size_t row = (size_t)(ok ? m : mbeg);
        if constexpr (RECT) {
            int img, pp, qq;
            rect_pixel(m, img, pp, qq);
            row = ok ? (size_t)(img * a.P + pp) * a.Q + qq : (size_t)0;   // (an empty rectangle has no valid m at all)
        }
        ra[p] = *reinterpret_cast<const float4*>(a.dy + row * a.ldy + a_co);
        okmask = (okmask & ~(1u << p)) | (ok ? (1u << p) : 0u);
    };
    auto gloadB = [&](int p, int kbase) {
        int m = kbase + p * KB + b_kr;
        int mm = m < mend ? m : mbeg;
        int img, pp, qq;
        if constexpr (RECT) {
            rect_pixel(m, img, pp, qq);
            if (m >= mend) img = pp = qq = 0;
        } else {
            img = fdiv(mm, a.fPQ);
            int rem = mm - img * (a.P * a.Q);
            pp = fdiv(rem, a.fQ);
            qq = rem - pp * a.Q;
        }
        int h = pp * a.stride + b_dh, w = qq * a.stride + b_dw;
        const bool ok = m < mend && h >= 0 && w >= 0 && h < a.H && w < a.W;
        h = h < 0 ? 0 : (h < a.H ? h : a.H - 1);   // clamped, always valid address: no divergent branch around the load
        w = w < 0 ? 0 : (w < a.W ? w : a.W - 1);
        rb[p] = *reinterpret_cast<const float4*>(a.x + (size_t)((img * a.H + h) * a.W + w) * a.ldx + b_ci);
        okmask = (okmask & ~(1u << (8 + p))) | (ok ? (1u << (8 + p)) : 0u);
    };
    auto gload = [&](int kbase) {
#pragma unroll
        for (int p = 0; p < PA; ++p) gloadA(p, kbase);
#pragma unroll
        for (int p = 0; p < PB; ++p) gloadB(p, kbase);
    };
    auto lstoreA = [&](int p, int buf) {
        *reinterpret_cast<float4*>(&As[buf * BUF + (p * KA + a_kr) * BM + a_c4 * 4]) = keep_or_zero((okmask >> p) & 1u, ra[p]);
    };
    auto lstoreB = [&](int p, int buf) {
        *reinterpret_cast<float4*>(&Bs[buf * BUF + (p * KB + b_kr) * BN + b_c4 * 4]) =
            keep_or_zero((okmask >> (8 + p)) & 1u, rb[p]);
    };
    auto lstore = [&](int buf = 0) {
#pragma unroll
        for (int p = 0; p < PA; ++p) lstoreA(p, buf);
#pragma unroll
        for (int p = 0; p < PB; ++p) lstoreB(p, buf);
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    if (mbeg < mend) {
        gload(mbeg);
        lstore();
    }
    __syncthreads();
    const float* Ard = As + lh * BM + wm * (BM / 2) + l31;
    const float* Brd = Bs + lh * BN + wn * (BN / 2) + l31;
    // fragments of k-step kk+1 are read from LDS while the MFMAs of step kk run (the compiler does not pipeline
    // this by itself: it emitted read, s_waitcnt lgkmcnt(0), 4 MFMAs per step — MFMA pipe 74 % busy in the K loop)
    auto mfma_steps = [&](int cur, int k0, int k1) {
        float af[2][TM], bf[2][TN];
        auto frag = [&](int b, int kk) {
#pragma unroll
            for (int i = 0; i < TM; ++i) af[b][i] = Ard[cur * BUF + 2 * kk * BM + i * 32];
#pragma unroll
            for (int j = 0; j < TN; ++j) bf[b][j] = Brd[cur * BUF + 2 * kk * BN + j * 32];
        };
        frag(0, k0);
#pragma unroll
        for (int kk = k0; kk < k1; ++kk) {
            const int b = (kk - k0) & 1;
            if (kk + 1 < k1) frag(b ^ 1, kk + 1);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[b][i], bf[b][j], acc[i][j], 0, 0, 0);
        }
        // pin the order: reads of step kk+1 BEFORE the MFMAs of step kk (left alone, the scheduler folds the two
        // fragment sets into one and issues each read right before its use)
        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
#pragma unroll
        for (int kk = k0; kk < k1; ++kk) {
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, TM * TN, 0);
        }
    };
    if (DB) {
        // slice t sits in LDS buffer t&1, slice t+1 in the staging registers: it is written to the other buffer in
        // the middle of slice t's MFMAs and the registers are refilled with slice t+2 — one barrier per slice and
        // no phase in which the workgroup issues no MFMA (two co-resident workgroups run in lockstep, so the
        // store-and-barrier phase of the single-buffer loop left the MFMA pipe idle: K loop 188 us for 130 us of MFMA)
        // The body is branch-free (slice starts past the end are clamped to the last slice, whose reload and re-store
        // are harmless).  In the second half of the slice every k-step also writes ONE staged row set to the other
        // LDS buffer and refills those registers from global memory, so the store / address arithmetic / load work
        // is spread between the MFMAs instead of forming an MFMA-free phase; sched_barrier keeps the steps apart
        // (left alone the scheduler hoists all stores to the top of the iteration, i.e. waits for the loads at once).
        const int last = mbeg < mend ? mbeg + (mend - mbeg - 1) / BK * BK : mbeg;
        gload(mbeg + BK <= last ? mbeg + BK : last);
        int cur = 0;
        for (int kb = mbeg; kb < mend; kb += BK, cur ^= 1) {
            const int k2 = kb + 2 * BK <= last ? kb + 2 * BK : last;
            float af[2][TM], bf[2][TN];
            auto frag = [&](int b, int kk) {
#pragma unroll
                for (int i = 0; i < TM; ++i) af[b][i] = Ard[cur * BUF + 2 * kk * BM + i * 32];
#pragma unroll
                for (int j = 0; j < TN; ++j) bf[b][j] = Brd[cur * BUF + 2 * kk * BN + j * 32];
            };
            frag(0, 0);
#pragma unroll
            for (int kk = 0; kk < BK / 2; ++kk) {
                const int b = kk & 1;
                if (kk + 1 < BK / 2) frag(b ^ 1, kk + 1);
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[b][i], bf[b][j], acc[i][j], 0, 0, 0);
                const int w = kk - BK / 4;   // staged row set handled by this step
                if (w >= 0 && w < PA) {
                    lstoreA(w, cur ^ 1);
                    gloadA(w, k2);
                } else if (w >= PA && w < PA + PB) {
                    lstoreB(w - PA, cur ^ 1);
                    gloadB(w - PA, k2);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            __syncthreads();
        }
    } else {
        for (int kb = mbeg; kb < mend; kb += BK) {
            const bool more = kb + BK < mend;
            if (more) gload(kb + BK);
            mfma_steps(0, 0, BK / 2);
            __syncthreads();
            if (more) {
                lstore();
                __syncthreads();
            }
        }
    }

    long long dbg_w1 = 0;
    if (DBG & 32) dbg_w1 = wall_clock64();
    float* out = a.slab + (size_t)split * a.K * a.Ncols;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        int col = col0 + wn * (BN / 2) + j * 32 + l31;
        if (col >= a.Ncols) continue;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                int co = co0 + wm * (BM / 2) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (co < a.K) out[(size_t)co * a.Ncols + col] = acc[i][j][r];
            }
    }
#ifndef UP_EMU
    if (DBG & 32) {   // probe: per-block timeline, same record as igemm_kernel
        __builtin_amdgcn_s_waitcnt(0);
        if (threadIdx.x == 0) {
            long long* o = reinterpret_cast<long long*>(a.dbg) + 4 * (size_t)blockIdx.x;
            o[0] = dbg_w0;
            o[1] = dbg_w1;
            o[2] = wall_clock64();
            o[3] = (long long)__builtin_amdgcn_s_getreg((31 << 11) | 4) |
                   ((long long)(__builtin_amdgcn_s_getreg((31 << 11) | 20) & 15) << 32);
        }
    }
#endif
}

// ------------------------------------------------------------------------------------------
// Weight gradient on v_mfma_f32_32x32x16_bf16 (BASELINE configs[4] arithmetic: bf16 operands, fp32 accumulation).
//   dW[co][col] = sum_m dY[m][co] * Xg[m][col],  col = tap * Cp + ci,  m = output pixel (the reduction index).
// The MFMA wants, per lane, EIGHT CONSECUTIVE k of one row; here k is the pixel index while both operands are
// channel-contiguous in HBM, i.e. each operand has to be transposed on its way to the fragment registers.  With fp32
// data in HBM the bf16 conversion does it for free: a thread owns 4 channels x 8 consecutive pixels (eight 16-byte
// loads), and v_cvt_pk_bf16_f32 packs (pixel 2j, pixel 2j+1) of ONE channel from two different load registers, so
// four packs give the 16 bytes "8 pixels of channel c" that one ds_write_b128 puts into the [row][pixel] LDS image.
// LDS image of a 32-pixel slice: (BM + BN) rows of 64 B + 16 B pad (80/16 = 5 odd: the 16-lane groups of the
// ds_read_b128 fragment reads hit 16 distinct 16-B slots).  Tile row R holds channel (R % (BM/4)) * 4 + R / (BM/4):
// with that order the ds_write_b128 of consecutive lanes (consecutive channel quads, same e) go to consecutive rows,
// 20 dwords apart, i.e. to eight disjoint bank quads per 8-lane group instead of two.
// Pipeline: as igemm_bf16_kernel — slice t in LDS buffer t&1, slices t+1 / t+2 in two register staging sets, the set
// holding t+1 is converted and written to the other buffer mid-slice and refilled with t+3; one barrier per slice.
// Reduction range: all N*P*Q pixels split over `splits` workgroups, or (a.rect) the live rectangle of the column
// tile's filter taps; both are walked as (image, row, column) with an incremental carry, two FastDivs per slice.
// Output: fp32 split-K slabs, summed by wgrad_reduce_kernel (unchanged).
// ------------------------------------------------------------------------------------------
// HS = bf16 storage (activations are bf16 in HBM): a staging unit is 8 channels x 8 pixels (eight 16-byte loads again),
// the 8x8 transpose of 16-bit elements is 32 v_perm-class operations (low / high half-word merges of two registers), a
// slice is 64 pixels so that every thread still owns exactly one unit, and tile row R holds channel
// (R % (BM/8)) * 8 + R / (BM/8).
template <bool HS>
struct WbGeom {
    static constexpr int E = HS ? 8 : 4;        // channels per staging unit
    static constexpr int KS = HS ? 64 : 32;     // pixels per slice
    static constexpr int RS = KS * 2 + 16;      // LDS row stride in bytes: 80 / 144, (RS/16) odd
};

template <int BM, int BN, bool HS = false>
__global__ void __launch_bounds__(256, 2) wgrad_bf16_kernel(WgradArgs a) {
    static_assert(BM % 64 == 0 && BN % 64 == 0 && BM + BN <= 256, "one staging unit per thread, wave-uniform roles");
    constexpr int E = WbGeom<HS>::E, WB_KS = WbGeom<HS>::KS, WB_RS = WbGeom<HS>::RS;
    constexpr int TM = BM / 64, TN = BN / 64;
    constexpr int A_BYTES = BM * WB_RS, BUF = (BM + BN) * WB_RS;
    constexpr int QA = BM / E, QB = BN / E;   // channel groups per tile side
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * BUF];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, lh = lane >> 5;

    const int logical = xcd_remap(blockIdx.x, a.nwg);
    const int split = fdiv(logical, a.fTiles);
    const int tile = logical - split * (int)a.fTiles.d;
    const int mt = fdiv(tile, a.fNtn);
    const int nt = tile - mt * a.ntn;
    const int co0 = mt * BM, col0 = nt * BN;

    // reduction domain of this workgroup: pixels [mbeg, mend) of a (images x rows x cols) box at (r_pl, r_ql)
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
    const int nslices = mbeg < mend ? (mend - mbeg + WB_KS - 1) / WB_KS : 0;

    // staging role of this thread: one unit = E channels (columns) x 8 consecutive pixels of the slice
    const bool isA = tid < BM, isB = !isA && tid < BM + BN;
    const int u = isA ? tid : tid - BM;
    const int qg = isA ? u % QA : u % QB;          // channel group within the tile side
    const int pg = isA ? u / QA : u / QB;          // pixel group of the slice
    int ch_off = 0, b_dh = 0, b_dw = 0;
    if (isA) {
        ch_off = (co0 + qg * E < a.ldy) ? co0 + qg * E : 0;      // channels >= K are never stored: any valid address
    } else {
        const int col = (col0 + qg * E < a.Ncols) ? col0 + qg * E : 0;
        const int tap = fdiv(col, a.fCp);
        ch_off = col - tap * a.Cp;
        const int r = fdiv(tap, a.fS);
        b_dh = r * a.dil - a.pad;
        b_dw = (tap - r * a.S) * a.dil - a.pad;
    }
    const float* src = isA ? a.dy : a.x;
    // LDS destination of channel e of the group: row e * Q + qg of the side, pixel group pg
    const int lds_dst = (isA ? 0 : A_BYTES) + qg * WB_RS + pg * 16;
    const int lds_estride = (isA ? QA : QB) * WB_RS;

    uint4 stX[8], stY[8];
    unsigned okX = 0, okY = 0;
#define UP_WB_STAGE(SFX)                                                                                          \
    auto gload##SFX = [&](int slice_req) {                                                                        \
        if (!(isA || isB)) return;                                                                                \
        const int sl = slice_req < nslices ? slice_req : nslices - 1;                                             \
        const int m0 = mbeg + sl * WB_KS + pg * 8;                                                                \
        int img = fdiv(m0, r_fhw);                                                                                \
        const int rem = m0 - img * r_hw;                                                                          \
        int pi = fdiv(rem, r_fw);                                                                                 \
        int qi = rem - pi * r_w;                                                                                  \
        unsigned msk = 0;                                                                                         \
        _Pragma("unroll") for (int j = 0; j < 8; ++j) {                                                           \
            bool ok = m0 + j < mend;                                                                              \
            int off;                                                                                              \
            if (isA) {                                                                                            \
                off = ((img * a.P + r_pl + pi) * a.Q + r_ql + qi) * a.ldy;                                        \
            } else {                                                                                              \
                const int h = (r_pl + pi) * a.stride + b_dh, w = (r_ql + qi) * a.stride + b_dw;                   \
                ok = ok && (unsigned)h < (unsigned)a.H && (unsigned)w < (unsigned)a.W;                            \
                off = ((img * a.H + h) * a.W + w) * a.ldx;                                                        \
            }                                                                                                     \
            off = (ok ? off : 0) + ch_off;                                                                        \
            st##SFX[j] = HS ? *reinterpret_cast<const uint4*>(reinterpret_cast<const bf16_t*>(src) + off)         \
                            : *reinterpret_cast<const uint4*>(src + off);                                         \
            msk |= ok ? (1u << j) : 0u;                                                                           \
            if (++qi == r_w) {                                                                                    \
                qi = 0;                                                                                           \
                if (++pi == r_h) {                                                                                \
                    pi = 0;                                                                                       \
                    ++img;                                                                                        \
                }                                                                                                 \
            }                                                                                                     \
        }                                                                                                         \
        ok##SFX = msk;                                                                                            \
    };                                                                                                            \
    auto lstore##SFX = [&](int buf) {                                                                             \
        if (!(isA || isB)) return;                                                                                \
        uint32_t v[8][4];                                                                                         \
        _Pragma("unroll") for (int j = 0; j < 8; ++j) {                                                           \
            const bool k = (ok##SFX >> j) & 1u;                                                                   \
            v[j][0] = k ? st##SFX[j].x : 0u;                                                                      \
            v[j][1] = k ? st##SFX[j].y : 0u;                                                                      \
            v[j][2] = k ? st##SFX[j].z : 0u;                                                                      \
            v[j][3] = k ? st##SFX[j].w : 0u;                                                                      \
        }                                                                                                         \
        unsigned char* d = smem + buf * BUF + lds_dst;                                                            \
        if constexpr (HS) { /* 8 pixels x 8 channels of bf16: word w of pixel j = channels 2w, 2w+1 */           \
            _Pragma("unroll") for (int w = 0; w < 4; ++w) {                                                       \
                uint4 lo, hi;                                                                                     \
                lo.x = (v[0][w] & 0xffffu) | (v[1][w] << 16);                                                     \
                lo.y = (v[2][w] & 0xffffu) | (v[3][w] << 16);                                                     \
                lo.z = (v[4][w] & 0xffffu) | (v[5][w] << 16);                                                     \
                lo.w = (v[6][w] & 0xffffu) | (v[7][w] << 16);                                                     \
                hi.x = (v[0][w] >> 16) | (v[1][w] & 0xffff0000u);                                                 \
                hi.y = (v[2][w] >> 16) | (v[3][w] & 0xffff0000u);                                                 \
                hi.z = (v[4][w] >> 16) | (v[5][w] & 0xffff0000u);                                                 \
                hi.w = (v[6][w] >> 16) | (v[7][w] & 0xffff0000u);                                                 \
                *reinterpret_cast<uint4*>(d + (2 * w) * lds_estride) = lo;                                        \
                *reinterpret_cast<uint4*>(d + (2 * w + 1) * lds_estride) = hi;                                    \
            }                                                                                                     \
        } else { /* 8 pixels x 4 channels of fp32: the conversion pack does the transpose */                      \
            _Pragma("unroll") for (int e = 0; e < 4; ++e)                                                         \
                *reinterpret_cast<uint4*>(d + e * lds_estride) =                                                  \
                    make_uint4(pack_bf16x2(__uint_as_float(v[0][e]), __uint_as_float(v[1][e])),                   \
                               pack_bf16x2(__uint_as_float(v[2][e]), __uint_as_float(v[3][e])),                   \
                               pack_bf16x2(__uint_as_float(v[4][e]), __uint_as_float(v[5][e])),                   \
                               pack_bf16x2(__uint_as_float(v[6][e]), __uint_as_float(v[7][e])));                  \
        }                                                                                                         \
    };
    UP_WB_STAGE(X)
    UP_WB_STAGE(Y)
#undef UP_WB_STAGE

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int a_rd = (wm * (BM / 2) + l31) * WB_RS + lh * 16;
    const int b_rd = A_BYTES + (wn * (BN / 2) + l31) * WB_RS + lh * 16;
#define UP_WB_BODY(SFX)                                                                                           \
    auto body##SFX = [&](int kt) {                                                                                \
        const unsigned char* base = smem + (kt & 1) * BUF;                                                        \
        _Pragma("unroll") for (int s = 0; s < WB_KS / 16; ++s) {                                                  \
            bf16x8 af[TM], bf[TN];                                                                                \
            _Pragma("unroll") for (int i = 0; i < TM; ++i)                                                        \
                af[i] = *reinterpret_cast<const bf16x8*>(base + a_rd + i * 32 * WB_RS + s * 32);                  \
            _Pragma("unroll") for (int j = 0; j < TN; ++j)                                                        \
                bf[j] = *reinterpret_cast<const bf16x8*>(base + b_rd + j * 32 * WB_RS + s * 32);                  \
            if (s == 0) lstore##SFX((kt & 1) ^ 1);                                                                \
            if (s == WB_KS / 16 - 1) gload##SFX(kt + 3);                                                          \
            _Pragma("unroll") for (int i = 0; i < TM; ++i) _Pragma("unroll") for (int j = 0; j < TN; ++j)         \
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bf[j], acc[i][j], 0, 0, 0);            \
        }                                                                                                         \
        __syncthreads();                                                                                          \
    };
    UP_WB_BODY(X)
    UP_WB_BODY(Y)
#undef UP_WB_BODY

    if (nslices > 0) {
        gloadX(0);
        lstoreX(0);
        gloadX(1);
        gloadY(2);
        __syncthreads();
        for (int kt = 0; kt < nslices; kt += 2) {
            bodyX(kt);
            if (kt + 1 < nslices) bodyY(kt + 1);
        }
    }

    float* out = a.slab + (size_t)split * a.K * a.Ncols;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int cc = wn * (BN / 2) + j * 32 + l31;            // tile column -> GEMM column (see the LDS row order)
        const int col = col0 + (cc % QB) * E + cc / QB;
        if (col >= a.Ncols) continue;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int rr = wm * (BM / 2) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                const int co = co0 + (rr % QA) * E + rr / QA;
                if (co < a.K) out[(size_t)co * a.Ncols + col] = acc[i][j][r];
            }
    }
}

#include "bf16s_glds.h"
#include "bf16s_big.h"
#include "f32_glds.h"
#include "stem_f32.h"

// sum the split-K slabs and scatter into PyTorch OIHW
// (Measured and removed in round 2: the merge folded into the weight-gradient kernels — every split publishes its slab with
//  write-through stores and bumps a per-tile arrival counter, the last arriver adds the slabs in split order and scatters the
//  tile.  Parity held, the step went from 67.4 to 144 ms (736^2 bf16 storage: 52 to 114 ms): the layers with few weight tiles
//  run 32-256 splits per tile, and ONE workgroup per tile then reads all of them, ~1.5 us per 64 KB slab tile, while this pass
//  spreads the same bytes over the whole chip.  profiles/r02_i_knob_ab.txt, r02_m.)
__global__ void __launch_bounds__(256) wgrad_reduce_kernel(const float* slab, float* dw, int splits, int K, int C,
                                                          int Cp, int taps, long long total4 /* K*taps*Cp / 4 */,
                                                          int accumulate /* dw += the sum instead of dw = the sum */) {
    // one thread = 4 consecutive input channels (Cp % 4 == 0): 16-byte slab reads
    long long q = (long long)blockIdx.x * 256 + threadIdx.x;
    if (q >= total4) return;
    const long long e = q * 4;
    const int ci = (int)(e % Cp);
    const long long t = e / Cp;
    const int tap = (int)(t % taps);
    const int co = (int)(t / taps);
    const size_t stride = (size_t)total4 * 4;
    // four splits in flight per step, four partial sums (fixed order: deterministic).  The pass is bound by the number of
    // DEPENDENT load rounds per thread (14-30 splits on the layers that carry the step), not by bytes, so the sums are
    // COMPENSATED (Kahan: the rounding error of every addition is carried along and subtracted; three more VALU operations per
    // element and add, hidden behind the loads).  The weight gradient is a sum with heavy cancellation — BatchNorm's backward makes
    // sum(dy) = sum(dy * xhat) = 0 per channel — and the stem reduces 1 083 392 pixels per weight: with plain additions the merge
    // of its 488 slabs landed 2.3e-3 from the float64 gradient (ATen: 7e-5).
    float4 s0 = make_float4(0.f, 0.f, 0.f, 0.f), s1 = s0, s2 = s0, s3 = s0;
    float4 c0 = s0, c1 = s0, c2 = s0, c3 = s0;
    auto kadd1 = [](float& sum, float& comp, float a) {
        const float y = a - comp;
        const float t = sum + y;
        comp = (t - sum) - y;
        sum = t;
    };
    auto add4 = [&](float4& d, float4& c, const float4& a) {
        kadd1(d.x, c.x, a.x);
        kadd1(d.y, c.y, a.y);
        kadd1(d.z, c.z, a.z);
        kadd1(d.w, c.w, a.w);
    };
    int sp = 0;
    for (; sp + 3 < splits; sp += 4) {
        const float4 a = *reinterpret_cast<const float4*>(slab + (size_t)sp * stride + e);
        const float4 b = *reinterpret_cast<const float4*>(slab + (size_t)(sp + 1) * stride + e);
        const float4 c = *reinterpret_cast<const float4*>(slab + (size_t)(sp + 2) * stride + e);
        const float4 d = *reinterpret_cast<const float4*>(slab + (size_t)(sp + 3) * stride + e);
        add4(s0, c0, a);
        add4(s1, c1, b);
        add4(s2, c2, c);
        add4(s3, c3, d);
    }
    for (; sp < splits; ++sp) add4(s0, c0, *reinterpret_cast<const float4*>(slab + (size_t)sp * stride + e));
    // the four chains into the first one, their pending corrections included
    auto neg = [](const float4& a) { return make_float4(-a.x, -a.y, -a.z, -a.w); };
    add4(s0, c0, s1);
    add4(s0, c0, neg(c1));
    add4(s0, c0, s2);
    add4(s0, c0, neg(c2));
    add4(s0, c0, s3);
    add4(s0, c0, neg(c3));
    const float v[4] = {s0.x - c0.x, s0.y - c0.y, s0.z - c0.z, s0.w - c0.w};
    if (taps == 1 && ci + 3 < C && (C & 3) == 0) {
        float4* o = reinterpret_cast<float4*>(dw + (size_t)co * C + ci);
        float4 r = make_float4(v[0], v[1], v[2], v[3]);
        if (accumulate) {
            const float4 old = *o;
            r = make_float4(old.x + r.x, old.y + r.y, old.z + r.z, old.w + r.w);
        }
        *o = r;
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (ci + j < C) {
                float* o = dw + ((size_t)co * C + ci + j) * taps + tap;
                *o = accumulate ? *o + v[j] : v[j];
            }
    }
}

// Parity classes of the data gradient of a stride-2, dilation-1 convolution.  dx pixel (h, w) only receives filter
// taps with r == (h + pad) mod 2 and s == (w + pad) mod 2, so the four (h%2, w%2) classes are four stride-1-like
// gathers over the dy grid with 1/4 of the taps on average; computing them as ONE gather with a divisibility test
// (the older MODE 1 path) multiplies 4x as many zeros.  Class cls = ph*2 + pw uses taps r = r0 + 2*ri, s = s0 + 2*si
// (ri < Rc, si < Sc); its weight image [C][Rc*Sc][Kp] starts at element base[cls] of the data-gradient image.
struct S2Classes {
    int r0[4], s0[4], Rc[4], Sc[4];
    long long base[4];
};
__host__ __device__ inline S2Classes s2_classes(int R, int S, int pad, int C, int Kp) {
    S2Classes c;
    long long b = 0;
    for (int cls = 0; cls < 4; ++cls) {
        const int ph = cls >> 1, pw = cls & 1;
        c.r0[cls] = (ph + pad) & 1;
        c.s0[cls] = (pw + pad) & 1;
        c.Rc[cls] = c.r0[cls] < R ? (R - c.r0[cls] + 1) / 2 : 0;
        c.Sc[cls] = c.s0[cls] < S ? (S - c.s0[cls] + 1) / 2 : 0;
        c.base[cls] = b;
        b += (long long)C * c.Rc[cls] * c.Sc[cls] * Kp;
    }
    return c;
}
__host__ __device__ inline bool s2_decomposed(int stride, int dil) { return stride == 2 && dil == 1; }
// value of element f of the class-major data-gradient image
__device__ __forceinline__ float s2_dgrad_elem(const float* w, long long f, const S2Classes& c, int K, int Kp, int C, int S,
                                               int taps) {
    int cls = 3;
    while (cls > 0 && f < c.base[cls]) --cls;
    const long long l = f - c.base[cls];
    const int tc = c.Rc[cls] * c.Sc[cls];
    const int k = (int)(l % Kp);
    const long long t = l / Kp;
    const int ti = (int)(t % tc);
    const int ch = (int)(t / tc);
    const int ri = ti / c.Sc[cls], si = ti - ri * c.Sc[cls];
    const int tap = (c.r0[cls] + 2 * ri) * S + c.s0[cls] + 2 * si;
    return k < K ? w[((size_t)k * C + ch) * taps + tap] : 0.f;
}
__global__ void __launch_bounds__(256) pack_dgrad_s2_kernel(const float* w, float* o, int K, int Kp, int C, int R, int S,
                                                            int pad, long long total) {
    long long e = (long long)blockIdx.x * 256 + threadIdx.x;
    if (e >= total) return;
    const S2Classes c = s2_classes(R, S, pad, C, Kp);
    o[e] = s2_dgrad_elem(w, e, c, K, Kp, C, S, R * S);
}

__global__ void __launch_bounds__(256) pack_fwd_kernel(const float* w, float* o, int K, int C, int Cp, int taps,
                                                       long long total) {
    long long e = (long long)blockIdx.x * 256 + threadIdx.x;
    if (e >= total) return;
    int ci = (int)(e % Cp);
    long long t = e / Cp;
    int tap = (int)(t % taps);
    int k = (int)(t / taps);
    o[e] = ci < C ? w[((size_t)k * C + ci) * taps + tap] : 0.f;
}

__global__ void __launch_bounds__(256) pack_dgrad_kernel(const float* w, float* o, int K, int Kp, int C, int taps,
                                                         long long total) {
    long long e = (long long)blockIdx.x * 256 + threadIdx.x;
    if (e >= total) return;
    int k = (int)(e % Kp);
    long long t = e / Kp;
    int tap = (int)(t % taps);
    int c = (int)(t / taps);
    o[e] = k < K ? w[((size_t)k * C + c) * taps + tap] : 0.f;
}

// both images of MANY parameters in one launch (blockIdx.y = job): the optimizer changes every weight once per
// step, and 2 x 115 separate 5-us launches cost more than the packing itself
__global__ void __launch_bounds__(256) pack_batched_kernel(const up_pack_job* jobs) {
    // Round 5: one ROW of an image (fixed (k, tap) forward / (c, tap) data gradient: Cp resp. Kp consecutive outputs) per wave and
    // iteration, 32 consecutive rows per block and round — one 32-bit division per row instead of three 64-bit div / mod pairs per
    // ELEMENT (the re-pack of all 115 weights cost 1.1 ms per step that way: since the optimizer hook of ops.py it IS in every step),
    // and the rows of a block share their source lines (data gradient: the 32 (c, tap) rows behind one 128-byte line of w[k][.][.]).
    const up_pack_job jb = jobs[blockIdx.y];
    const int taps = jb.taps;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    // geometry word: stride | R << 4 | S << 10 | pad << 16 | dil << 24 (0 = plain layout)
    const int gstride = jb.geometry & 15, gR = (jb.geometry >> 4) & 63, gS = (jb.geometry >> 10) & 63;
    const int gpad = (jb.geometry >> 16) & 255, gdil = (jb.geometry >> 24) & 255;
    const bool s2 = jb.geometry != 0 && s2_decomposed(gstride, gdil);
    if (jb.w_fwd) {
        const int rows = jb.K * taps;
        for (int base = blockIdx.x * 32; base < rows; base += gridDim.x * 32) {
            const int end = min(base + 32, rows);
            for (int rt = base + wave; rt < end; rt += 4) {
                const int k = rt / taps, tap = rt - k * taps;
                const float* src = jb.w + (size_t)k * jb.C * taps + tap;
                float* dst = jb.w_fwd + (size_t)rt * jb.Cp;
                // four loads in flight per lane, then their stores: one load -> store per iteration left a wave with ONE memory
                // round trip at a time (the launch is a few dozen dependent round trips long, not bandwidth bound)
                for (int c0 = lane; c0 < jb.Cp; c0 += 256) {
                    float v[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int ci = c0 + 64 * u;
                        v[u] = ci < jb.C ? src[(size_t)ci * taps] : 0.f;
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u)
                        if (c0 + 64 * u < jb.Cp) dst[c0 + 64 * u] = v[u];
                }
            }
        }
    }
    if (!jb.w_dgrad) return;
    if (s2) {   // parity-class-major image of a stride-2 convolution (three weights of the network): element by element
        S2Classes cls = s2_classes(gR, gS, gpad, jb.C, jb.Kp);
        const long long nd = (long long)jb.C * taps * jb.Kp;
        const long long step = (long long)gridDim.x * 256;
        for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < nd; e += step)
            jb.w_dgrad[e] = s2_dgrad_elem(jb.w, e, cls, jb.K, jb.Kp, jb.C, gS, taps);
        return;
    }
    const int rows = jb.C * taps;
    const size_t kstride = (size_t)jb.C * taps;
    // Transposed through LDS: for one output channel k the 32 rows of a block are 32 CONSECUTIVE floats of w[k][.][.] (one 128-byte
    // line), so a wave reads two channels' lines per instruction (lanes = rows), parks them in a [32 rows][64 channels] tile and
    // the tile leaves as 256-byte rows (lanes = channels).  Lanes = channels on the read side touched 64 different lines per load.
    __shared__ float tile[32][65];
    const int r31 = lane & 31, khalf = lane >> 5;
    for (int base = blockIdx.x * 32; base < rows; base += gridDim.x * 32) {
        const int end = min(base + 32, rows);
        for (int k0 = 0; k0 < jb.Kp; k0 += 64) {
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {   // wave w reads channels k0 + 16 w + 2 j + khalf
                const int k = k0 + wave * 16 + 2 * j + khalf;
                v[j] = (k < jb.K && base + r31 < end) ? jb.w[(size_t)k * kstride + base + r31] : 0.f;
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) tile[r31][wave * 16 + 2 * j + khalf] = v[j];
            __syncthreads();
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int rt = base + wave + 4 * j;
                if (rt < end && k0 + lane < jb.Kp) jb.w_dgrad[(size_t)rt * jb.Kp + k0 + lane] = tile[wave + 4 * j][lane];
            }
            __syncthreads();
        }
    }
}

// column sums of a [rows][ld] matrix (bias gradient): one partial row per block, part[block][C]; colsum_finish_kernel adds the
// rows in a fixed order (round 6: the float atomics of rounds 1-5 made the bias gradient — and through the bias every later step —
// differ from run to run in the last bit)
template <typename T>
__global__ void __launch_bounds__(256) colsum_kernel(const T* x, int ld, long long rows, int C, float* part,
                                                     int rows_per_block) {
    // blockDim = 256 = 4 row-lanes x 64 columns
    __shared__ float red[256];
    int c = blockIdx.y * 64 + (threadIdx.x & 63);
    int rl = threadIdx.x >> 6;
    long long r0 = (long long)blockIdx.x * rows_per_block;
    long long r1 = r0 + rows_per_block < rows ? r0 + rows_per_block : rows;
    float s = 0.f;
    if (c < C)
        for (long long r = r0 + rl; r < r1; r += 4) s += ld1(x + r * ld + c);
    red[threadIdx.x] = s;
    __syncthreads();
    if (rl == 0 && c < C) {
        s = red[threadIdx.x] + red[threadIdx.x + 64] + red[threadIdx.x + 128] + red[threadIdx.x + 192];
        part[(size_t)blockIdx.x * C + c] = s;
    }
}
// one wavefront per channel: lane l adds partial rows l, l + 64, ... in order, then a butterfly over the lanes (fixed order)
__global__ void __launch_bounds__(256) colsum_finish_kernel(const float* part, int blocks, int C, float* out, int accumulate) {
    const int c = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    float s = 0.f;
    if (c < C)
        for (int b = lane; b < blocks; b += 64) s += part[(size_t)b * C + c];
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off);
    if (c < C && lane == 0) out[c] = accumulate ? out[c] + s : s;
}

// the same with four channels per thread (16-byte loads, 16 row lanes x 16 channel quads) and 256-row blocks: the 1024-row form above
// ran the bias gradient of the video head (16 928 x 128) on 34 workgroups in 66 us, 31 times per step (profiles/r03_m_kernel_stats_lstm)
template <typename T>
__global__ void __launch_bounds__(256) colsum4_kernel(const T* x, int ld, long long rows, int C, float* part, int rows_per_block) {
    __shared__ float red[16][64];
    const int cq = threadIdx.x & 15, rl = threadIdx.x >> 4;
    const int c = blockIdx.y * 64 + cq * 4;
    const long long r0 = (long long)blockIdx.x * rows_per_block;
    const long long r1 = r0 + rows_per_block < rows ? r0 + rows_per_block : rows;
    float s[4] = {0.f, 0.f, 0.f, 0.f};
    if (c < ld) {   // ld % 4 == 0: a quad that starts inside the row lies inside it (pad channels are zeros)
        long long r = r0 + rl;
        for (; r + 48 < r1; r += 64) {   // four loads in flight
            const float4 a0 = ld4<T>(x + r * ld + c), a1 = ld4<T>(x + (r + 16) * ld + c), a2 = ld4<T>(x + (r + 32) * ld + c),
                         a3 = ld4<T>(x + (r + 48) * ld + c);
            s[0] += (a0.x + a1.x) + (a2.x + a3.x);
            s[1] += (a0.y + a1.y) + (a2.y + a3.y);
            s[2] += (a0.z + a1.z) + (a2.z + a3.z);
            s[3] += (a0.w + a1.w) + (a2.w + a3.w);
        }
        for (; r < r1; r += 16) {
            const float4 a = ld4<T>(x + r * ld + c);
            s[0] += a.x;
            s[1] += a.y;
            s[2] += a.z;
            s[3] += a.w;
        }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) red[rl][cq * 4 + e] = s[e];
    __syncthreads();
    if (threadIdx.x < 64) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) t += red[k][threadIdx.x];
        const int cc = blockIdx.y * 64 + threadIdx.x;
        if (cc < C) part[(size_t)blockIdx.x * C + cc] = t;
    }
}

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------
static int check_desc(const up_conv_desc* d) {
    UP_REQUIRE(d, UP_ERR_INVALID, "conv desc is null");
    UP_REQUIRE(d->N > 0 && d->H > 0 && d->W > 0 && d->C > 0 && d->K > 0 && d->R > 0 && d->S > 0, UP_ERR_INVALID,
               "conv desc: non-positive dimension");
    UP_REQUIRE(d->Cp % 4 == 0 && d->Cp >= d->C && d->ldx % 4 == 0 && d->ldx >= d->Cp, UP_ERR_INVALID,
               "conv desc: Cp=%d ldx=%d must be multiples of 4 with ldx >= Cp >= C=%d", d->Cp, d->ldx, d->C);
    UP_REQUIRE(d->ldy >= d->K, UP_ERR_INVALID, "conv desc: ldy=%d < K=%d", d->ldy, d->K);
    UP_REQUIRE(d->stride >= 1 && d->dil >= 1 && d->pad >= 0, UP_ERR_INVALID, "conv desc: bad stride/dil/pad");
    int P = (d->H + 2 * d->pad - d->dil * (d->R - 1) - 1) / d->stride + 1;
    int Q = (d->W + 2 * d->pad - d->dil * (d->S - 1) - 1) / d->stride + 1;
    UP_REQUIRE(P == d->P && Q == d->Q, UP_ERR_INVALID, "conv desc: P,Q=(%d,%d) but geometry gives (%d,%d)", d->P,
               d->Q, P, Q);
    UP_REQUIRE((int64_t)d->N * d->H * d->W < (1ll << 31) && (int64_t)d->N * d->P * d->Q < (1ll << 31),
               UP_ERR_UNSUPPORTED, "conv: more than 2^31 pixels");
    return UP_OK;
}

struct TileChoice {
    int bm, bn;
};
// knobs of the default form (environment at load time, up_conv_tune at run time)
static int env_int(const char* name, int dflt, int min_ok) {
    const char* e = getenv(name);
    return e && atoi(e) >= min_ok ? atoi(e) : dflt;
}
static int g_tile_want = env_int("UP_TILE_WANT", 1500, 1);   // A/B in the network: 700 / 1000 / 1500 / 2200 / 4300 -> 70.1 / 69.9 / 69.6 / 69.9 / 69.8 ms
static int g_tile_want_bf16 = env_int("UP_TILE_WANT_BF16", 500, 1);
static int g_glds = env_int("UP_GLDS", 1, 0);   // bf16 storage: direct-to-LDS kernels of bf16s_glds.h (0 = the register-staged round-2 kernels)
// bf16 storage, round 6: launches with N % 256 == 0, 64-aligned channels and a reduction of at least big_min_k run one 8-wave
// workgroup per CU on 160/192/256 x 256 tiles (bf16s_big.h); 0 = igemm_glds_kernel everywhere
static int g_big = env_int("UP_GLDS_BIG", 1, 0);
static int g_big_min_k = env_int("UP_BIG_MIN_K", 1024, 1);   // (whole 736^2 step: 256 -> 39.5 ms, 512 -> 38.1, 1024 -> 36.1-36.3, 2048 -> 36.1-36.2 against 36.8-37.0 without; shorter reductions are all prologue + epilogue at one workgroup per CU)
static int g_big_dgrad = env_int("UP_BIG_DGRAD", 1, 0);      // 0: data gradients (tapstep < 0) stay on igemm_glds_kernel
static int g_big_rows = env_int("UP_BIG_ROWS", 0, 0);         // 160 / 192 / 256: only this tile height (0: the rule of big_tile_rows)
static int g_big_stages = env_int("UP_BIG_STAGES", 3, 2);   // LDS stages of the 160-row tiles (3: two slices in flight)
// fp32 forward / data gradient with operands HBM -> LDS by LDS-DMA (f32_glds.h, round 4): glds32 = 0 keeps the register-staged
// igemm_kernel; glds32_epi: 1 = LDS-transposed 16-byte-store epilogue, 0 = igemm_epilogue
static long long g_count_igemm = 0, g_count_glds32 = 0, g_count_glds32_epi1 = 0, g_count_glds32_bnred = 0;   // up_conv_counter
static long long g_count_big = 0;   // launches on igemm_big_kernel (bf16s_big.h)
static long long g_count_wgrad32 = 0, g_count_wgrad32_st1 = 0, g_count_glds32_wide = 0, g_count_glds32_grouped = 0;
static thread_local bool g_extras_dropped = false;   // (per host thread: autograd runs one thread per device) a launch was asked for a masked addend / fused reduction on a kernel without them
static int g_glds32 = env_int("UP_GLDS32", 1, 0);
static int g_glds32_epi = env_int("UP_GLDS32_EPI", 1, 0);

static int g_glds32_wgrad = env_int("UP_GLDS32_WGRAD", 1, 0);   // fp32 weight gradient with LDS-DMA operands (wgrad_glds32_kernel)
static int g_db_min_k = 1024;   // (settled in round 1/2; no longer a run-time knob)
static int g_short_k = 512;            // reductions shorter than this are epilogue-heavy:
static int g_short_k_mult = 4;         // they want g_short_k_mult / 2 times as many workgroups (r04_d sweep: 2 / 4 / 8 within noise)
static int g_tail_split = env_int("UP_TAIL_SPLIT", 1, 0);
static int g_tap_skip = 1;   // (tile-level tap skipping: on since round 1; the probe build switches it for its WASP report)
static int g_wgrad_per_cu = 2;   // workgroups per CU a weight-gradient launch aims for (r04_d: 1 -> +0.7 ms, 3 -> +1.1 ms per step)
static int g_wgrad_rect = env_int("UP_WGRAD_RECT", 1, 0);      // weight-gradient reduction over live rectangles (see WgradRectKey); on since r02_a (-0.65 ms per step)
static int cu_count();
// Largest tile that still yields ~6 workgroups per CU (the tail split evens out the remainder).  Short reductions
// are epilogue-heavy and run better on twice as many, smaller tiles (1x1 256->1024 at 23x23: 64x128 91 TF,
// 128x128 85 TF).
// (math >= UP_MATH_BF16: the plain-bf16 kernels do 16x the MFMA work per cycle and live on operand reuse, they want
//  larger tiles: 736^2 B=16 step in bf16 storage 52.0 ms at 1500, 49.0-49.5 ms anywhere in 300..1000, profiles/r02_t)
// Round 5: reductions of at most tiny_k (K <= 128: the 1x1 64->256 / 128->512 layers of the 92x92 / 46x46 stages and their data
// gradients, two to four K slices per tile) always take 64x64 tiles, whatever the workgroup count: such a tile's life is set-up and
// epilogue, and four small workgroups per CU overlap those better than two large ones.  Whole fp32 step, three alternations
// (profiles/r05_experiments.txt): 0 (the workgroup-count rule alone) 61.95-62.07 ms, 64: 61.72-61.95, 128: 61.58-61.69,
// 256: 61.60-61.68, 512: 61.75-61.81, 576: 61.91-61.96, 1152: 61.76-61.92.  fp32 only: the bf16-storage kernels lose with it
// (736^2 step 37.56 / 37.81 ms without, 37.71-38.27 at 64, 37.90-37.95 at 128, 38.47-38.51 at 256).
static int g_tiny_k = env_int("UP_TINY_K", 128, 0);
static TileChoice choose_tile(int64_t M, int Ng, int Ktot, int math = 0) {
    const int cands[4][2] = {{128, 128}, {64, 128}, {128, 64}, {64, 64}};
    if (math == UP_MATH_F32 && Ktot <= g_tiny_k) return {64, 64};
    const int base_want = math >= UP_MATH_BF16 ? g_tile_want_bf16 : g_tile_want;   // workgroups a launch should at least have
    const int64_t want = Ktot < g_short_k ? (int64_t)g_short_k_mult * base_want / 2 : base_want;
    for (auto& c : cands) {
        if (Ng <= 64 && c[1] == 128) continue;
        int64_t wgs = (int64_t)cdiv(M, c[0]) * cdiv(Ng, c[1]);
        if (wgs >= want) return {c[0], c[1]};
    }
    return {64, 64};
}

// bf16 storage: the tile of a launch described by `a` (pointers may be null: the tile-count queries ask before there are tensors).
// bn = 256 says igemm_big_kernel with bm rows per tile.
static int fill_fwd_args(IgemmArgs& a, const up_conv_desc* d, const float* x, const float* w_fwd, float* y, const up_conv_epilogue* ep);
static int fill_dgrad_args(IgemmArgs& a, const up_conv_desc* d, const float* dy, const float* w_dgrad, float* dx);
static bool glds_eligible(const IgemmArgs& a, bool fast);
static bool igemm_fast(const IgemmArgs& a) {
    return a.taps <= 32 && a.divshift == 0 && (long long)a.H * a.W * a.ldx * ((long long)a.M / (a.P * a.Q) + 1) < (1ll << 31);
}
static TileChoice choose_tile_bf16s(const IgemmArgs& a) {
    if (g_big && (g_big_dgrad || a.tapstep >= 0) && glds_eligible(a, igemm_fast(a))) {
        const int bm = glds::big_tile_rows(a.M, a.Ng, a.Ktot, a.Cp, cu_count(), g_big_min_k, g_big_rows);
        if (bm) return {bm, 256};
    }
    return choose_tile(a.M, a.Ng, a.Ktot, UP_MATH_BF16S);
}

// ---- tail split -------------------------------------------------------------------------------------
// Workgroups are dispatched round-robin over the CUs (tools/gpu/igemm_probe.hip timeline), so a launch of
// T = q*CUs + r tiles leaves r CUs with q+1 tiles and the others idle for a whole tile time at the end: 18 % of
// the launch for the 1060-tile layers of the 23x23 / 46x46 stages (q = 4, r = 36).  When r is small the r tail
// tiles are split along K into p = CUs/r parts, one per CU, so every CU gets q + 1/p tiles.  Parts 0..p-2 publish
// their accumulators to a per-stream scratch, the last part of each tile adds them in a fixed order (the result
// is deterministic) and runs the normal epilogue.  Measured (probe, 3x3 256->256 at 46x46): 108 -> 131 TFLOP/s.
// Tried and dropped: parts FIRST in the grid (no gain: the whole tiles that start late end up alone on their CU
// and a lone workgroup is latency bound), 2-4 parts per CU (slower: longer merge chain), a staggered start of the
// workgroups sharing a CU (no gain).
struct SplitScratch {
    float* partials = nullptr;
    int* flags = nullptr;
    size_t pfloats = 0, nflags = 0;   // capacity
    // BatchNorm fold (bn_fold.h): tickets (zero between launches) + level-1 rows, allocated by the first folding launch on the stream
    int* tickets = nullptr;
    double* part2 = nullptr;
};
constexpr size_t FOLD_TICKETS = 1 << 16;          // ints: columns * (groups + 1) of the largest merge
constexpr size_t FOLD_PART2_BYTES = 8u << 20;     // level-1 rows: groups * C * nv doubles
static int g_cu_override = 0;   // up_conv_tune("cu_count", n): tests shrink the "chip" so that small launches have whole rounds + a tail
static int cu_count() {
    if (g_cu_override > 0) return g_cu_override;
    static const int n = [] {
        if (const char* e = getenv("UP_CU_COUNT")) return atoi(e) > 0 ? atoi(e) : 256;   // tests shrink the "chip"
#ifndef UP_EMU
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess &&
            prop.multiProcessorCount > 0)
            return prop.multiProcessorCount;
#endif
        return 256;
    }();
    return n;
}
static std::mutex g_scratch_mu;
static std::map<hipStream_t, SplitScratch> g_scratch;   // released per stream by up_stream_release
static SplitScratch* split_scratch(hipStream_t st) {
    // at most one partial per CU and launch; launches on one stream are serialised, so one buffer per stream suffices
    std::lock_guard<std::mutex> lock(g_scratch_mu);
    SplitScratch& s = g_scratch[st];
    if (!s.partials) {
        // tail split: at most one 128x128 partial per CU and launch; the all-tiles split of small launches cuts the same
        // bytes into up to four 64x64 partials per CU, one flag each
        const size_t slots = (size_t)cu_count();
        const size_t nflags = 4 * (size_t)cu_count();
        const size_t pbytes = slots * 128 * 128 * sizeof(float), fbytes = nflags * sizeof(int);
#ifdef UP_EMU
        s.partials = static_cast<float*>(malloc(pbytes));
        s.flags = static_cast<int*>(calloc(1, fbytes));
#else
        if (hipMalloc(reinterpret_cast<void**>(&s.partials), pbytes) != hipSuccess ||
            hipMalloc(reinterpret_cast<void**>(&s.flags), fbytes) != hipSuccess ||
            hipMemset(s.flags, 0, fbytes) != hipSuccess) {
            (void)hipGetLastError();
            s.partials = nullptr;
            return nullptr;   // no scratch: the caller falls back to whole tiles
        }
#endif
        s.pfloats = pbytes / sizeof(float);
        s.nflags = nflags;
    }
    return &s;
}
static bool tail_split_enabled() { return g_tail_split != 0; }

// ---- BatchNorm fold (bn_fold.h) -----------------------------------------------------------------------
static int g_bn_fold = env_int("UP_BN_FOLD", 1, 0);   // up_conv_tune("bn_fold", 0): producers leave the finalize to the stand-alone kernel (same bits)
bool bn_fold_enabled() { return g_bn_fold != 0; }
bool bn_fold_scratch(hipStream_t st, int tiles, int C, int nv, BnFold* f, int fgroups) {
    if (tiles < 1 || C < 1 || fgroups < 1) return false;
    const int groups = (tiles + FOLD_G - 1) / FOLD_G;
    const size_t cols = (size_t)(C + FOLD_COLS - 1) / FOLD_COLS, rows2 = (size_t)fgroups * groups;
    if (cols * (rows2 + 1) > FOLD_TICKETS || rows2 * C * nv * sizeof(double) > FOLD_PART2_BYTES) return false;
    std::lock_guard<std::mutex> lock(g_scratch_mu);
    SplitScratch& s = g_scratch[st];
    if (!s.tickets) {
#ifdef UP_EMU
        s.tickets = static_cast<int*>(calloc(FOLD_TICKETS, sizeof(int)));
        s.part2 = static_cast<double*>(malloc(FOLD_PART2_BYTES));
#else
        if (hipMalloc(reinterpret_cast<void**>(&s.tickets), FOLD_TICKETS * sizeof(int)) != hipSuccess ||
            hipMalloc(reinterpret_cast<void**>(&s.part2), FOLD_PART2_BYTES) != hipSuccess ||
            hipMemset(s.tickets, 0, FOLD_TICKETS * sizeof(int)) != hipSuccess) {
            (void)hipGetLastError();
            if (s.tickets) (void)hipFree(s.tickets);
            s.tickets = nullptr;
            return false;
        }
#endif
    }
    f->tickets = s.tickets;
    f->part2 = s.part2;
    f->tiles = tiles;
    f->groups = groups;
    f->C = C;
    f->fgroups = fgroups;
    return true;
}
// what the entry point asked the next launch to fold (per host thread, consumed by launch_igemm / launch_igemm_bf16)
struct FoldRequest {
    up_bn_fold* fwd = nullptr;
    up_bn_reduce_slot* bwd = nullptr;
};
static thread_local FoldRequest g_fold_req;
// `tiles`: row tiles of the launch about to be issued (its partial rows).  Fills a.fold when the launch carries the finalize.
static void apply_fold(IgemmArgs& a, int tiles, hipStream_t st) {
    a.fold.tickets = nullptr;
    a.fold_nv = 0;
    const FoldRequest rq = g_fold_req;
    g_fold_req = FoldRequest();
    if (!g_bn_fold) return;
    const int fgroups = a.grp_rows ? a.M / a.grp_rows : 1;
    if (rq.fwd && a.stats && !a.grp_rows) {
        up_bn_fold* q = rq.fwd;
        if (!bn_fold_scratch(st, tiles, a.Ng, 3, &a.fold)) return;
        a.fold.eps = q->eps;
        a.fold.mom = q->momentum;
        a.fold.rm = q->running_mean;
        a.fold.rv = q->running_var;
        a.fold.gamma = q->gamma;
        a.fold.beta = q->beta;
        a.fold.mean = q->mean;
        a.fold.invstd = q->invstd;
        a.fold.scale = q->scale;
        a.fold.shift = q->shift;
        a.fold_nv = 3;
        q->folded = 1;
    } else if (rq.bwd && a.bn_partial && rq.bwd->dgamma && rq.bwd->dbeta && (fgroups == 1 || rq.bwd->gsum)) {
        // row groups: `tiles` counts all groups' tiles; every group merges its own grp_tiles rows
        if (!bn_fold_scratch(st, a.grp_rows ? a.grp_tiles : tiles, a.Ng, 2, &a.fold, fgroups)) return;
        a.fold.dgamma = rq.bwd->dgamma;
        a.fold.dbeta = rq.bwd->dbeta;
        a.fold.gsum = rq.bwd->gsum;
        a.fold.ostride = 2 * a.Ng;
        a.fold_nv = 2;
        rq.bwd->folded = 1;
    }
}
}  // namespace up
// The library's only per-stream device memory is the K-split scratch above (16 MB + flags, allocated by the first split launch on
// a stream).  A caller that retires a stream — unipose_amd.graph.GraphedForward owns a private capture stream — hands it back
// here once nothing that ran (or was captured) on the stream can execute any more.
extern "C" int up_stream_release(void* stream) {
    using namespace up;
    std::lock_guard<std::mutex> lock(g_scratch_mu);
    auto it = g_scratch.find(as_stream(stream));
    if (it == g_scratch.end()) return UP_OK;
#ifdef UP_EMU
    free(it->second.partials);
    free(it->second.flags);
    free(it->second.tickets);
    free(it->second.part2);
#else
    if (it->second.partials) (void)hipFree(it->second.partials);
    if (it->second.flags) (void)hipFree(it->second.flags);
    if (it->second.tickets) (void)hipFree(it->second.tickets);
    if (it->second.part2) (void)hipFree(it->second.part2);
#endif
    g_scratch.erase(it);
    return UP_OK;
}
namespace up {

// parts each tail tile is split into (1 = no split) for a launch of `tiles` tiles reducing over Ktot
// *all_tiles: every tile is split, not only the tail of the launch
static int g_split_per_cu = env_int("UP_SPLIT_PER_CU", 2, 1);
static int split_parts(int tiles, int Ktot, size_t slots, bool* all_tiles = nullptr) {
    if (all_tiles) *all_tiles = false;
    if (!tail_split_enabled()) return 1;
    const int cus = cu_count(), q = tiles / cus, r = tiles % cus, nk = Ktot / BK;
    // A/B over the whole step: r <= 25 % of the CUs 70.9 ms, 50 % 70.1, 75 % 70.1;  >= 2 / 4 / 8 slices per part 70.1 / 70.1 / 70.3
    int p = 1;
    if (!(r == 0 || r > cus / 2 || q > 12)) {
        p = cus / r;                  // one part per CU (finer cuts of a TAIL measured slower: the merge chain grows)
        if (p > nk / 2) p = nk / 2;   // a part keeps >= 2 K slices
        if (p < 2) p = 1;
    }
    // Small batches (inference at B <= 4): with fewer tiles than CUs every workgroup is alone on its CU, where the K loop
    // runs at 57 % of what four co-resident workgroups reach.  Split EVERY tile so that the launch has g_split_per_cu
    // workgroups per CU (>= 4 K slices per part) when that is a finer cut than the tail rule's.  Measured (tools/gpu/
    // small_batch.py, 368^2 inference forward): B = 1 3.44 -> 3.14 ms, B = 4 4.42 -> 3.79 ms with 2 per CU (3 and 4: the
    // same); launches with one to two tiles per CU (B = 8: 268 tiles) gain nothing from it and keep the tail rule.
    if (g_split_per_cu > 1 && q == 0 && all_tiles) {
        int pn = g_split_per_cu * cus / tiles;
        if (pn > nk / 4) pn = nk / 4;
        if (pn > 16) pn = 16;
        if (pn >= 2 && pn > p && (size_t)tiles * (pn - 1) <= slots) {
            *all_tiles = true;
            return pn;
        }
    }
    return p;
}

// ---- tap-sorted row order -----------------------------------------------------------------------------
// The K loop of a tile skips the filter taps that are dead (read padding) for ALL of its rows.  With rows in image
// order a 64-row tile spans ~3 image rows of a 23x23 map and therefore every column class, so horizontally nothing
// is ever skipped, and tiles that straddle two images keep all taps: the dilation-18 WASP branch still multiplied
// 3.3 of 9 taps per pixel where 2.1 are live (effective MFMA fraction 19 %).  Sorting the GEMM rows by their tap mask
// (classes with more live taps first, image order inside a class) makes the masks of a tile (nearly) uniform: the
// tile-level skipping then drops (nearly) every dead tap — also the border taps of ordinary padded 3x3 convolutions
// (5.7 % of the MACs at 23x23).  The permutation depends only on the geometry; it is built once on the host and
// kept on the device.  Knob "tap_sort" (UP_TAP_SORT).  Measured (profiles/r02_a_*): WASP d = 12 / 18 forward 0.166 -> 0.099 /
// 0.157 -> 0.093 ms, whole step -0.45 ms: on by default.
static int g_tap_sort = env_int("UP_TAP_SORT", 1, 0);
struct TapSortKey {
    int M, H, W, P, Q, taps, S, mul, off0, off0w, tapstep;
    bool operator<(const TapSortKey& o) const { return memcmp(this, &o, sizeof(*this)) < 0; }
};
// tap mask of every pixel of ONE image of the destination grid (bit t: tap t reads a real source pixel)
static std::vector<unsigned> tap_masks(const IgemmArgs& a) {
    std::vector<unsigned> mask((size_t)a.P * a.Q);
    for (int p = 0; p < a.P; ++p)
        for (int q = 0; q < a.Q; ++q) {
            const int hb = p * a.mul + a.off0, wb = q * a.mul + a.off0w;
            unsigned mk = 0;
            for (int t = 0; t < a.taps; ++t) {
                const int h = hb + (t / a.S) * a.tapstep, w = wb + (t % a.S) * a.tapstep;
                if (h >= 0 && w >= 0 && h < a.H && w < a.W) mk |= 1u << t;
            }
            mask[(size_t)p * a.Q + q] = mk;
        }
    return mask;
}
// rows in tap-sorted order: classes with more live taps first, then by mask value; image order inside a class; then the
// 128-row blocks of that order are dealt out over the XCDs (below).  Empty when every pixel has the same mask.
static std::vector<int> tap_sort_order(const IgemmArgs& a, const std::vector<unsigned>& mask) {
    std::vector<unsigned> classes;
    for (unsigned mk : mask) {
        bool seen = false;
        for (unsigned c : classes) seen = seen || c == mk;
        if (!seen) classes.push_back(mk);
    }
    std::vector<int> perm;
    if (classes.size() < 2) return perm;
    std::sort(classes.begin(), classes.end(), [](unsigned x, unsigned y) {
        const int px = __builtin_popcount(x), py = __builtin_popcount(y);
        return px != py ? px > py : x < y;
    });
    const int PQ = a.P * a.Q, imgs = a.M / PQ;
    perm.reserve(a.M);
    for (unsigned c : classes)
        for (int img = 0; img < imgs; ++img)
            for (int e = 0; e < PQ; ++e)
                if (mask[e] == c) perm.push_back(img * PQ + e);
    // Sorted like this, every XCD (a contiguous run of logical tiles, xcd_remap) would hold ONE class: the XCD with the
    // nine-tap rows of the dilation-6 WASP branch ran as long as the dense convolution while the others idled (0.190 vs
    // 0.194 ms, r02_n).  Deal the 128-row blocks of the sorted order out over the eight runs instead, heaviest first, so
    // that every XCD — and, dispatched in order, every CU of it — gets the same mix of long and short tiles.
    constexpr int BLK = 128, XCDS = 8;
    const int nb = a.M / BLK;   // (rows past the last full block keep their place at the end)
    if (nb >= 2 * XCDS) {
        int lo[XCDS + 1];
        for (int x = 0; x <= XCDS; ++x) lo[x] = (int)((long long)x * nb / XCDS);
        std::vector<int> dealt(perm);
        int r = 0;
        for (int slot = 0; r < nb; ++slot)
            for (int x = 0; x < XCDS && r < nb; ++x)
                if (lo[x] + slot < lo[x + 1]) {
                    memcpy(&dealt[(size_t)(lo[x] + slot) * BLK], &perm[(size_t)r * BLK], sizeof(int) * BLK);
                    ++r;
                }
        perm.swap(dealt);
    }
    return perm;
}
static const int* tap_sort_perm(const IgemmArgs& a) {
    static std::mutex mu;
    static std::map<TapSortKey, int*> table;
    TapSortKey key;
    memset(&key, 0, sizeof(key));
    key.M = a.M; key.H = a.H; key.W = a.W; key.P = a.P; key.Q = a.Q; key.taps = a.taps; key.S = a.S;
    key.mul = a.mul; key.off0 = a.off0; key.off0w = a.off0w; key.tapstep = a.tapstep;
    std::lock_guard<std::mutex> lock(mu);
    auto it = table.find(key);
    if (it != table.end()) return it->second;
    // Variable-size inference: the per-geometry tables are bounded by REFUSING new entries, never by freeing old ones — a captured
    // hipGraph (unipose_amd.graph.GraphedForward) carries a.perm as a raw device pointer in its kernel arguments, and a
    // hipDeviceSynchronize / hipFree here would also break a capture in progress.  A geometry beyond the bound
    // keeps the image order (correct, only the tile-level tap skipping is less sharp).
    if (table.size() >= 1024) return nullptr;
    const std::vector<int> perm = tap_sort_order(a, tap_masks(a));
    int* dev = nullptr;
    if ((int)perm.size() == a.M) {
#ifdef UP_EMU
        dev = static_cast<int*>(malloc(sizeof(int) * a.M));
        memcpy(dev, perm.data(), sizeof(int) * a.M);
#else
        if (hipMalloc(reinterpret_cast<void**>(&dev), sizeof(int) * a.M) != hipSuccess ||
            hipMemcpy(dev, perm.data(), sizeof(int) * a.M, hipMemcpyHostToDevice) != hipSuccess) {
            (void)hipGetLastError();
            dev = nullptr;   // no permutation: the launch keeps the image order
        }
#endif
    }
    table[key] = dev;
    return dev;
}
// share of (row tile, filter tap) pairs the K loop of a launch visits, for rows in image order or tap-sorted order;
// *live = share of (pixel, tap) pairs that touch the image at all (what a perfect skip would visit)
static double visited_tap_fraction(const IgemmArgs& a, int bm, bool sorted, double* live) {
    const std::vector<unsigned> mask = tap_masks(a);
    const int PQ = a.P * a.Q;
    std::vector<int> perm;
    if (sorted && a.M % PQ == 0) perm = tap_sort_order(a, mask);
    long long visited = 0, alive = 0;
    const int tiles = cdiv(a.M, bm);
    for (int t = 0; t < tiles; ++t) {
        unsigned u = 0;
        for (int m = t * bm; m < (t + 1) * bm; ++m) {
            const int mm = m < a.M ? m : a.M - 1;   // rows past the end are clamped like in the kernel
            u |= mask[(perm.empty() ? mm : perm[mm]) % PQ];
        }
        visited += __builtin_popcount(u);
    }
    for (unsigned mk : mask) alive += __builtin_popcount(mk);
    if (live) *live = (double)alive / ((double)PQ * a.taps);
    return (double)visited / ((double)tiles * a.taps);
}

// Share of (pixel, filter tap) pairs of a DILATED launch that read a real source pixel: what its per-launch FLOP is charged with in the
// profiler (round 5) — the nominal 2 M N K counted the taps that fall into the padding and that the tile-level skipping / tap-sorted
// rows never multiply (WASP d = 18: 0.229, layer4 d = 8: 0.59); it inflated the dominant variant's TFLOP/s by 3-4 %.
// Per geometry, cached; only evaluated while the profiler is on.
static double live_tap_share(const IgemmArgs& a) {
    // dilated launches only: a dense padded 3x3 keeps the nominal count, the convention behind SURVEY 8d's 31.279 GMAC per image
    if (a.taps <= 1 || a.taps > 32 || a.no_tap_skip || a.divshift != 0 || (a.tapstep >= -1 && a.tapstep <= 1)) return 1.0;
    static std::mutex mu;
    static std::map<TapSortKey, double> cache;
    TapSortKey key;
    memset(&key, 0, sizeof(key));
    key.M = 0; key.H = a.H; key.W = a.W; key.P = a.P; key.Q = a.Q; key.taps = a.taps; key.S = a.S;
    key.mul = a.mul; key.off0 = a.off0; key.off0w = a.off0w; key.tapstep = a.tapstep;
    std::lock_guard<std::mutex> lock(mu);
    auto it = cache.find(key);
    if (it != cache.end()) return it->second;
    long long alive = 0;
    for (unsigned mk : tap_masks(a)) alive += __builtin_popcount(mk);
    const double share = (double)alive / ((double)a.P * a.Q * a.taps);
    if (cache.size() < 4096) cache[key] = share;
    return share;
}

// f32_glds.h: needs the aligned fast path (<= 32 taps, no strided gather), 31-bit BYTE offsets and 16-byte aligned operands
static bool glds32_eligible(IgemmArgs& a, bool fast) {
    if (!g_glds32 || !fast || a.Cp % 32 != 0 || a.M % (a.P * a.Q) != 0) return false;
    const long long a_bytes = (long long)(a.M / (a.P * a.Q)) * a.H * a.W * a.ldx * 4;
    const uintptr_t ptrs = reinterpret_cast<uintptr_t>(a.x) | reinterpret_cast<uintptr_t>(a.w);
    if (a_bytes >= (1ll << 31) || (long long)a.Ng * a.Ktot * 4 >= (1ll << 31) || (ptrs & 15)) return false;
    a.x_bytes = (uint32_t)a_bytes;
    return true;
}
// the LDS-transposed epilogue stores float4 rows: 4-channel granularity everywhere, linear (or tap-sorted) output rows
static bool glds32_epi1_ok(const IgemmArgs& a) {
    const uintptr_t ptrs = reinterpret_cast<uintptr_t>(a.y) | reinterpret_cast<uintptr_t>(a.residual);
    return g_glds32_epi && !a.o_mode && a.Ng % 4 == 0 && a.ldy % 4 == 0 && (!a.residual || a.ldr % 4 == 0) && (ptrs & 15) == 0;
}
// "hybrid" operand path of the direct-to-LDS fp32 kernel (f32_glds.h BREG): weight fragments global -> registers.  0: off,
// 1: pointwise (1x1) launches.  up_conv_tune("breg", v) / UP_BREG.
static int g_breg = env_int("UP_BREG", 0, 0);
static long long g_count_glds32_breg = 0;
template <int BM, int BN>
static auto glds32_kernel(const IgemmArgs& a) -> void (*)(IgemmArgs) {
    constexpr int OCC2 = (BM == 128 && BN == 128) ? 2 : (BM == 64 && BN == 64) ? 4 : 3;
    const bool epi1 = glds32_epi1_ok(a);
    const bool bnred = epi1 && a.bn_partial != nullptr;
    // (the 64x64 form with the fused reduction would spill: 128 VGPRs at four workgroups per CU; it keeps the LDS operand path)
    if (g_breg && a.taps == 1 && !a.perm && !(bnred && BM == 64 && BN == 64)) {
        ++g_count_glds32_breg;
        if (bnred) return glds::igemm_glds32_kernel<BM, BN, false, 2, OCC2, 1, true, false, true>;
        if (epi1) return glds::igemm_glds32_kernel<BM, BN, false, 2, OCC2, 1, false, false, true>;
        return glds::igemm_glds32_kernel<BM, BN, false, 2, OCC2, 0, false, false, true>;
    }
    // (a one-stage form — 21 KB of LDS, six 64x64 workgroups per CU — for reductions shorter than 1024 measured 0.4 ms per step
    //  SLOWER than two stages at four per CU, profiles/r04_a_*, and is not instantiated)
    if (a.perm) {
        if (bnred) return glds::igemm_glds32_kernel<BM, BN, true, 2, OCC2, 1, true>;
        if (epi1) return glds::igemm_glds32_kernel<BM, BN, true, 2, OCC2, 1, false>;
        return glds::igemm_glds32_kernel<BM, BN, true, 2, OCC2, 0, false>;
    }
    if (bnred) return glds::igemm_glds32_kernel<BM, BN, false, 2, OCC2, 1, true>;
    if (epi1) return glds::igemm_glds32_kernel<BM, BN, false, 2, OCC2, 1, false>;
    return glds::igemm_glds32_kernel<BM, BN, false, 2, OCC2, 0, false>;
}

template <int BM, int BN>
static auto glds32_wide_kernel(const IgemmArgs& a) -> void (*)(IgemmArgs) {
    constexpr int OCC2 = (BM == 128 && BN == 128) ? 2 : (BM == 64 && BN == 64) ? 4 : 3;
    if (glds32_epi1_ok(a)) return glds::igemm_glds32_kernel<BM, BN, false, 2, OCC2, 1, false, true>;
    return glds::igemm_glds32_kernel<BM, BN, false, 2, OCC2, 0, false, true>;
}

static thread_local bool g_group_refused = false;   // a launch asked for row groups on a kernel without them (nothing was launched)
template <int BM, int BN>
static void launch_igemm(IgemmArgs& a, bool aligned, hipStream_t st) {
    int ntm = cdiv(a.M, BM);
    if (a.grp_rows) {   // row groups: every group tiled on its own (IgemmArgs)
        a.grp_tiles = cdiv(a.grp_rows, BM);
        a.fGrpTiles = make_fastdiv(a.grp_tiles);
        ntm = (a.M / a.grp_rows) * a.grp_tiles;
    }
    a.ntn = cdiv(a.Ng, BN);
    a.nwg = ntm * a.ntn;
    a.fNtn = make_fastdiv(a.ntn);
    a.fSpt = make_fastdiv(a.Cp >= 32 ? a.Cp / 32 : 1);
    const int vbase = (BM == 128 && BN == 128) ? 0 : (BM == 64 && BN == 128) ? 2 : (BM == 128 && BN == 64) ? 4 : 6;
    // MODE 2 needs <= 32 taps (bit mask), no stride division and 32-bit element offsets
    const bool fast = aligned && a.taps <= 32 && a.divshift == 0 &&
                      (long long)a.H * a.W * a.ldx * ((long long)a.M / (a.P * a.Q) + 1) < (1ll << 31);
    // filters of more than 32 taps (the video head's 11x11): the direct-to-LDS kernel's WIDE form (separable row / column masks)
    const bool wide = aligned && a.taps > 32 && a.S <= 16 && a.taps / a.S <= 16 && a.divshift == 0 &&
                      (long long)a.H * a.W * a.ldx * ((long long)a.M / (a.P * a.Q) + 1) < (1ll << 31);
    const bool use32 = glds32_eligible(a, fast || wide);   // direct-to-LDS generation (f32_glds.h)
    if (a.grp_rows && !(use32 && !wide && glds32_epi1_ok(a) && a.grp_rows % (a.P * a.Q) == 0)) {
        g_group_refused = true;   // only f32_glds.h's LDS-transposed epilogue knows row groups
        return;
    }
    if ((a.bn_partial || a.res_bits) && !(use32 && glds32_epi1_ok(a) && !(wide && a.bn_partial))) {
        // nothing is launched — and nothing is recorded by the profiler (the refusal used to sit behind the ProfScope):
        // dx without the masked addend must never be written
        g_extras_dropped = true;
        return;
    }
    a.no_tap_skip = g_tap_skip ? 0 : 1;
    ProfScope prof(use32 ? 28 + vbase / 2 : vbase + (aligned ? 0 : 1), 2.0 * (double)a.M * (double)a.Ng * (double)a.Ktot_real, st,
                   a.M, a.Ng, a.Ktot, a.nwg, g_prof_on_host() && fast ? live_tap_share(a) : 1.0);
    // double-buffered LDS (one barrier per slice) for long reductions.  In isolation it is 3-5 % faster than the
    // single-buffer loop down to K = 256 (probe), but in the network the rule K >= 1024 is 0.5 % faster per step (A/B in
    // one session, 70.55 vs 70.95 ms): the 73 KB footprint leaves less room for the weight-gradient workgroups of
    // the other stream.  UP_DB_MIN_K overrides the threshold.
    const bool db = a.Ktot >= g_db_min_k;
    // wide tiles additionally pin the refill between the MFMAs (branch-free body + sched_group_barrier): +2..4 %
    // on 128-wide tiles, -3 % on 64x64 (probe, warm)
    constexpr int DB_VARIANT = (BM == 128 || BN == 128) ? 128 : 0;   // 64x64: non-pinned loop with two staging sets
    void (*kernel)(IgemmArgs);
    a.no_tap_skip = g_tap_skip ? 0 : 1;
    a.perm = nullptr;
    if (g_tap_sort && fast && a.taps > 1 && a.taps <= 16 && !a.residual && !a.o_mode && !a.no_tap_skip &&
        a.M % (a.P * a.Q) == 0) {
        if (a.grp_rows) {   // the permutation of ONE group (what a call on that group alone uses), applied group by group
            IgemmArgs ag = a;
            ag.M = a.grp_rows;
            a.perm = tap_sort_perm(ag);
        } else {
            a.perm = tap_sort_perm(a);
        }
    }
    // (the register-staged MODE 2 forms are the fallback of the direct-to-LDS kernels — glds32 = 0, unaligned pointers,
    //  >= 2^31 bytes — and the A/B partner of their tests; the double-buffered ones always use the XOR-swizzled LDS rows)
    if (a.perm && db)
        kernel = igemm_kernel<BM, BN, 2, DB_VARIANT, 32, true, true>;
    else if (fast && db)
        kernel = igemm_kernel<BM, BN, 2, DB_VARIANT, 32, false, true>;
    else if (a.perm)
        kernel = igemm_kernel<BM, BN, 2, 64, 32, true>;
    else if (fast)
        kernel = igemm_kernel<BM, BN, 2, 64>;
    else if (aligned)
        kernel = igemm_kernel<BM, BN, 1, 64>;
    else
        kernel = igemm_kernel<BM, BN, 0, 64>;

    a.full_blocks = a.nwg;
    a.parts = 1;
    a.no_tap_skip = g_tap_skip ? 0 : 1;
    int grid = a.nwg;
    SplitScratch* sc0 = aligned && tail_split_enabled() ? split_scratch(st) : nullptr;
    bool all_tiles = false;
    const size_t slots = sc0 ? std::min(sc0->pfloats / (size_t)(BM * BN), sc0->nflags) : 0;
    const int p = aligned ? split_parts(a.nwg, a.Ktot, slots, &all_tiles) : 1;
    if (p >= 2) {
        if (SplitScratch* sc = sc0) {
            a.full_blocks = all_tiles ? 0 : a.nwg / cu_count() * cu_count();
            a.parts = p;
            a.partials = sc->partials;
            a.flags = sc->flags;
            grid = a.full_blocks + (a.nwg - a.full_blocks) * p;
        }
    }
    if (use32) {
        kernel = wide ? glds32_wide_kernel<BM, BN>(a) : glds32_kernel<BM, BN>(a);
        ++g_count_glds32;
        if (wide) ++g_count_glds32_wide;
        if (a.grp_rows) ++g_count_glds32_grouped;
        if (glds32_epi1_ok(a)) ++g_count_glds32_epi1;
        if (glds32_epi1_ok(a) && a.bn_partial) ++g_count_glds32_bnred;
    } else {
        ++g_count_igemm;
    }
    apply_fold(a, ntm, st);
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(256), 0, st, a);
}

static int g_stem7 = env_int("UP_STEM7", 0, 0);   // 1: the first convolution on stem7_kernel (stem_f32.h: faster alone, not inside the step — off)
static long long g_count_stem7 = 0;
static void launch_stem7(IgemmArgs& a, hipStream_t st) {
    const int ntm = cdiv(a.M, 128);
    a.ntn = 1;
    a.nwg = ntm;
    a.fNtn = make_fastdiv(1);
    a.x_bytes = (uint32_t)((long long)(a.M / (a.P * a.Q)) * a.H * a.W * 16);
    ProfScope prof(39, 2.0 * (double)a.M * (double)a.Ng * (double)a.Ktot_real, st, a.M, a.Ng, a.Ktot, a.nwg);
    apply_fold(a, ntm, st);
    a.full_blocks = a.nwg;
    a.parts = 1;
    ++g_count_stem7;
    const int grid = std::min(ntm, 2 * cu_count());   // two workgroups per CU, each walks its share of the tiles
    hipLaunchKernelGGL(glds::stem7_kernel, dim3(grid), dim3(256), 0, st, a);
}
static void run_igemm(IgemmArgs& a, TileChoice t, hipStream_t st) {
    bool aligned = (a.Cp % BK) == 0;
    if (g_stem7 && t.bm == 128 && t.bn == 64 && !a.residual && glds::stem7_eligible(a)) {
        launch_stem7(a, st);
        return;
    }
    if (t.bm == 128 && t.bn == 128)
        launch_igemm<128, 128>(a, aligned, st);
    else if (t.bm == 64 && t.bn == 128)
        launch_igemm<64, 128>(a, aligned, st);
    else if (t.bm == 128 && t.bn == 64)
        launch_igemm<128, 64>(a, aligned, st);
    else
        launch_igemm<64, 64>(a, aligned, st);
}

}  // namespace up

using namespace up;

extern "C" int up_conv_stats_tiles(const up_conv_desc* d) { return up_conv_stats_tiles_math(d, UP_MATH_F32); }
extern "C" int up_conv_stats_tiles_math(const up_conv_desc* d, int math) {
    if (!d || math < UP_MATH_F32 || math > UP_MATH_BF16S) return UP_ERR_INVALID;
    int64_t M = (int64_t)d->N * d->P * d->Q;
    if (math == UP_MATH_BF16S && !check_desc(d)) {   // (the 8-wave tiles of bf16s_big.h depend on more than M, N, K)
        IgemmArgs a;
        if (fill_fwd_args(a, d, nullptr, nullptr, nullptr, nullptr) == UP_OK) return cdiv(M, choose_tile_bf16s(a).bm);
    }
    return cdiv(M, choose_tile(M, d->K, d->R * d->S * d->Cp, math).bm);
}

extern "C" int up_pack_weights_batched(const up_pack_job* jobs_device, int njobs, void* stream) {
    UP_REQUIRE(jobs_device && njobs > 0 && njobs <= 65535, UP_ERR_INVALID, "pack_weights_batched: bad job table");
    // 144 blocks per job: one round of 32-row groups for the largest weights (512x512x3x3: 4608 rows per image); blocks without rows leave at once
    hipLaunchKernelGGL(pack_batched_kernel, dim3(144, njobs), dim3(256), 0, as_stream(stream), jobs_device);
    return check_launch("pack_weights_batched");
}

extern "C" int up_conv_tune(const char* key, int value) {
    UP_REQUIRE(key, UP_ERR_INVALID, "conv_tune: null key");
    if (!strcmp(key, "tile_want") && value > 0) g_tile_want = value;
    else if (!strcmp(key, "tile_want_bf16") && value > 0) g_tile_want_bf16 = value;
    else if (!strcmp(key, "bn_rows")) set_bn_rows(value);
    else if (!strcmp(key, "glds")) g_glds = value ? 1 : 0;
    else if (!strcmp(key, "stem7")) g_stem7 = value ? 1 : 0;
    else if (!strcmp(key, "glds_big")) g_big = value ? 1 : 0;
    else if (!strcmp(key, "big_min_k") && value > 0) g_big_min_k = value;
    else if (!strcmp(key, "big_stages") && value >= 2) g_big_stages = value;
    else if (!strcmp(key, "big_dgrad")) g_big_dgrad = value ? 1 : 0;
    else if (!strcmp(key, "big_rows") && (value == 0 || value == 160 || value == 192 || value == 256)) g_big_rows = value;
    else if (!strcmp(key, "cu_count") && value >= 0) g_cu_override = value;
    else if (!strcmp(key, "glds32")) g_glds32 = value ? 1 : 0;
    else if (!strcmp(key, "glds32_epi")) g_glds32_epi = value ? 1 : 0;
    else if (!strcmp(key, "glds32_wgrad")) g_glds32_wgrad = value ? 1 : 0;
    else if (!strcmp(key, "tail_split")) g_tail_split = value ? 1 : 0;
    else if (!strcmp(key, "tiny_k") && value >= 0) g_tiny_k = value;
    else if (!strcmp(key, "split_per_cu") && value > 0) g_split_per_cu = value;
    else if (!strcmp(key, "tap_sort")) g_tap_sort = value ? 1 : 0;
    else if (!strcmp(key, "wgrad_rect")) g_wgrad_rect = value ? 1 : 0;
    else if (!strcmp(key, "bn_fold")) g_bn_fold = value ? 1 : 0;
    else if (!strcmp(key, "breg") && value >= 0) g_breg = value;
    else UP_REQUIRE(false, UP_ERR_INVALID, "conv_tune: unknown key '%s'", key);
    return UP_OK;
}

extern "C" long long up_conv_counter(const char* name) {
    if (!name) return -1;
    if (!strcmp(name, "igemm")) return g_count_igemm;
    if (!strcmp(name, "glds32")) return g_count_glds32;
    if (!strcmp(name, "glds32_wide")) return g_count_glds32_wide;
    if (!strcmp(name, "glds32_grouped")) return g_count_glds32_grouped;
    if (!strcmp(name, "glds32_epi1")) return g_count_glds32_epi1;
    if (!strcmp(name, "glds32_bnred")) return g_count_glds32_bnred;
    if (!strcmp(name, "glds32_breg")) return g_count_glds32_breg;
    if (!strcmp(name, "big")) return g_count_big;
    if (!strcmp(name, "stem7")) return g_count_stem7;
    if (!strcmp(name, "wgrad_glds32")) return g_count_wgrad32;
    if (!strcmp(name, "wgrad_glds32_st1")) return g_count_wgrad32_st1;
    return -1;
}

extern "C" int up_conv_split_parts(const up_conv_desc* d) {
    if (!d || check_desc(d)) return UP_ERR_INVALID;
    if (d->Cp % BK) return 1;
    int64_t M = (int64_t)d->N * d->P * d->Q;
    TileChoice t = choose_tile(M, d->K, d->R * d->S * d->Cp);
    bool all_tiles = false;
    return split_parts(cdiv(M, t.bm) * cdiv(d->K, t.bn), d->R * d->S * d->Cp,
                       (size_t)cu_count() * (128 * 128) / (size_t)(t.bm * t.bn), &all_tiles);
}

extern "C" int up_pack_weights(const up_conv_desc* d, const float* w, float* w_fwd, float* w_dgrad, void* stream) {
    if (int e = check_desc(d)) return e;
    UP_REQUIRE(w, UP_ERR_INVALID, "pack_weights: null weight");
    int taps = d->R * d->S;
    if (w_fwd) {
        long long total = (long long)d->K * taps * d->Cp;
        hipLaunchKernelGGL(pack_fwd_kernel, dim3(cdiv(total, 256)), dim3(256), 0, as_stream(stream), w, w_fwd, d->K,
                           d->C, d->Cp, taps, total);
    }
    if (w_dgrad) {
        UP_REQUIRE(d->Kp % 4 == 0 && d->Kp >= d->K, UP_ERR_INVALID, "pack_weights: Kp=%d invalid for K=%d", d->Kp,
                   d->K);
        long long total = (long long)d->C * taps * d->Kp;
        if (s2_decomposed(d->stride, d->dil))   // class-major image, see S2Classes
            hipLaunchKernelGGL(pack_dgrad_s2_kernel, dim3(cdiv(total, 256)), dim3(256), 0, as_stream(stream), w, w_dgrad,
                               d->K, d->Kp, d->C, d->R, d->S, d->pad, total);
        else
            hipLaunchKernelGGL(pack_dgrad_kernel, dim3(cdiv(total, 256)), dim3(256), 0, as_stream(stream), w, w_dgrad,
                               d->K, d->Kp, d->C, taps, total);
    }
    return check_launch("pack_weights");
}

namespace up {
static int fill_fwd_args(IgemmArgs& a, const up_conv_desc* d, const float* x, const float* w_fwd, float* y,
                         const up_conv_epilogue* ep) {
    memset(&a, 0, sizeof(a));
    a.x = x;
    a.w = w_fwd;
    a.y = y;
    a.M = d->N * d->P * d->Q;
    a.Ng = d->K;
    a.Cp = d->Cp;
    a.Ktot = d->R * d->S * d->Cp;
    a.Ktot_real = (long long)d->R * d->S * d->C;
    a.H = d->H;
    a.W = d->W;
    a.P = d->P;
    a.Q = d->Q;
    a.ldx = d->ldx;
    a.ldy = d->ldy;
    a.S = d->S;
    a.taps = d->R * d->S;
    a.mul = d->stride;
    a.off0 = -d->pad;
    a.off0w = -d->pad;
    a.tapstep = d->dil;
    a.divshift = 0;
    a.divmask = 0;
    a.fPQ = make_fastdiv(d->P * d->Q);
    a.fQ = make_fastdiv(d->Q);
    a.fCp = make_fastdiv(d->Cp);
    a.fS = make_fastdiv(d->S);
    if (ep) {
        UP_REQUIRE(!ep->scale || ep->shift, UP_ERR_INVALID, "conv2d_fwd: scale without shift");
        UP_REQUIRE(!ep->residual || ep->ldr >= d->K, UP_ERR_INVALID, "conv2d_fwd: residual stride < K");
        UP_REQUIRE(!ep->stats || !(ep->scale || ep->bias || ep->residual || ep->relu), UP_ERR_INVALID,
                   "conv2d_fwd: stats are taken on the raw accumulator; no other epilogue allowed");
        a.scale = ep->scale;
        a.shift = ep->shift;
        a.bias = ep->bias;
        a.residual = ep->residual;
        a.ldr = ep->ldr;
        a.relu = ep->relu;
        a.stats = ep->stats;
        if (ep->fold) {   // BatchNorm finalize folded into this launch (bn_fold.h); the launch code decides and reports `folded`
            up_bn_fold* q = ep->fold;
            q->folded = 0;
            UP_REQUIRE(ep->stats && q->gamma && q->beta && q->mean && q->invstd && q->scale && q->shift &&
                           (q->running_mean == nullptr) == (q->running_var == nullptr),
                       UP_ERR_INVALID, "conv2d_fwd: fold needs stats, gamma / beta, the four outputs and running statistics in pairs");
            g_fold_req.fwd = q;
        }
    }
    return UP_OK;
}
}  // namespace up

extern "C" int up_conv2d_fwd(const up_conv_desc* d, const float* x, const float* w_fwd, float* y,
                             const up_conv_epilogue* ep, void* stream) {
    if (int e = check_desc(d)) return e;
    UP_REQUIRE(x && w_fwd && y, UP_ERR_INVALID, "conv2d_fwd: null pointer");
    IgemmArgs a;
    g_fold_req = FoldRequest();
    if (int e = fill_fwd_args(a, d, x, w_fwd, y, ep)) return e;
    run_igemm(a, choose_tile(a.M, a.Ng, a.Ktot), as_stream(stream));
    g_fold_req = FoldRequest();
    return check_launch("conv2d_fwd");
}

// Row groups (the video model's batched trunk: `groups` frames stacked along N, BatchNorm statistics per frame).  Every group is
// tiled on its own, so stats is [groups][up_conv_stats_tiles_grouped][K][3] — exactly what up_bn_finalize_groups merges — and no
// extra pass over y is needed.  0 / UP_ERR_UNSUPPORTED when the launch cannot run on the direct-to-LDS fp32 kernel with the
// LDS-transposed epilogue (the caller then keeps up_conv2d_fwd + up_bn_batch_stats_t).
static bool grouped_ok(const up_conv_desc* d, int groups, bool data_gradient) {
    if (!d || check_desc(d) || groups < 1 || d->N % groups) return false;
    if (!g_glds32 || !g_glds32_epi) return false;
    const int cin = data_gradient ? d->Kp : d->Cp, nout = data_gradient ? d->C : d->K, ldo = data_gradient ? d->ldx : d->ldy;
    if (cin % 32 || d->R * d->S > 32 || nout % 4 || ldo % 4 || (data_gradient && d->stride != 1)) return false;
    const long long in_elems = data_gradient ? (long long)d->N * d->P * d->Q * d->ldy : (long long)d->N * d->H * d->W * d->ldx;
    const long long per_img = data_gradient ? (long long)d->P * d->Q * d->ldy : (long long)d->H * d->W * d->ldx;
    if (in_elems * 4 >= (1ll << 31) || per_img * ((long long)d->N + 1) >= (1ll << 31) ||
        (long long)nout * d->R * d->S * cin * 4 >= (1ll << 31))
        return false;
    return true;
}
extern "C" int up_conv_stats_tiles_grouped(const up_conv_desc* d, int groups) {
    if (!grouped_ok(d, groups, false)) return 0;
    const int64_t M = (int64_t)d->N * d->P * d->Q;
    return cdiv(M / groups, choose_tile(M, d->K, d->R * d->S * d->Cp).bm);
}
extern "C" int up_conv2d_fwd_grouped(const up_conv_desc* d, const float* x, const float* w_fwd, float* y, float* stats, int groups,
                                     void* stream) {
    if (int e = check_desc(d)) return e;
    UP_REQUIRE(x && w_fwd && y && stats, UP_ERR_INVALID, "conv2d_fwd_grouped: null pointer");
    UP_REQUIRE(grouped_ok(d, groups, false), UP_ERR_UNSUPPORTED, "conv2d_fwd_grouped: this convolution cannot be tiled per group "
               "(up_conv_stats_tiles_grouped = 0)");
    IgemmArgs a;
    up_conv_epilogue ep;
    memset(&ep, 0, sizeof(ep));
    ep.stats = stats;
    if (int e = fill_fwd_args(a, d, x, w_fwd, y, &ep)) return e;
    a.grp_rows = groups > 1 ? a.M / groups : 0;
    g_group_refused = false;
    run_igemm(a, choose_tile(a.M, a.Ng, a.Ktot), as_stream(stream));
    UP_REQUIRE(!g_group_refused, UP_ERR_UNSUPPORTED, "conv2d_fwd_grouped: the launch did not qualify for the direct-to-LDS kernel "
               "(pointer alignment); nothing was launched");
    return check_launch("conv2d_fwd_grouped");
}

namespace up {
static int fill_dgrad_args(IgemmArgs& a, const up_conv_desc* d, const float* dy, const float* w_dgrad, float* dx) {
    UP_REQUIRE(d->stride <= 2, UP_ERR_UNSUPPORTED, "conv2d_bwd_data: stride %d (only 1 and 2 are implemented)", d->stride);
    UP_REQUIRE(d->Kp % 4 == 0 && d->Kp >= d->K && d->ldy % 4 == 0 && d->ldy >= d->Kp, UP_ERR_INVALID,
               "conv2d_bwd_data: need Kp%%4==0, ldy%%4==0, ldy>=Kp (Kp=%d ldy=%d K=%d)", d->Kp, d->ldy, d->K);
    memset(&a, 0, sizeof(a));
    a.x = dy;
    a.w = w_dgrad;
    a.y = dx;
    a.M = d->N * d->H * d->W;
    a.Ng = d->C;
    a.Cp = d->Kp;
    a.Ktot = d->R * d->S * d->Kp;
    a.Ktot_real = (long long)d->R * d->S * d->K;
    a.H = d->P;
    a.W = d->Q;  // source = dy
    a.P = d->H;
    a.Q = d->W;  // destination = dx
    a.ldx = d->ldy;
    a.ldy = d->ldx;
    a.S = d->S;
    a.taps = d->R * d->S;
    a.mul = 1;
    a.off0 = d->pad;
    a.off0w = d->pad;
    a.tapstep = -d->dil;
    a.divshift = d->stride == 2 ? 1 : 0;
    a.divmask = d->stride == 2 ? 1 : 0;
    a.fPQ = make_fastdiv(d->H * d->W);
    a.fQ = make_fastdiv(d->W);
    a.fCp = make_fastdiv(d->Kp);
    a.fS = make_fastdiv(d->S);
    return UP_OK;
}

// ---- bf16-operand launches ----
// second-generation bf16-storage kernel (bf16s_glds.h): needs 8-channel (16-byte) output rows, 31-bit byte offsets, 16-byte
// aligned pointers and no strided gather
static bool glds_eligible(const IgemmArgs& a, bool fast) {
    const long long a_bytes = (long long)(a.M / (a.P * a.Q)) * a.H * a.W * a.ldx * 2;
    const uintptr_t ptrs = reinterpret_cast<uintptr_t>(a.x) | reinterpret_cast<uintptr_t>(a.y) |
                           reinterpret_cast<uintptr_t>(a.w_hi) | reinterpret_cast<uintptr_t>(a.residual);
    return g_glds && fast && a.M % (a.P * a.Q) == 0 && a.Ng % 8 == 0 && a.ldy % 8 == 0 && a.ldx % 8 == 0 &&
           (!a.residual || a.ldr % 8 == 0) && !a.o_mode && a_bytes < (1ll << 31) && (long long)a.Ng * a.Ktot * 2 < (1ll << 31) &&
           (long long)a.M * a.ldy < (1ll << 31) && (!a.residual || (long long)a.M * a.ldr < (1ll << 31)) && (ptrs & 15) == 0;
}
template <int BM, int BN>
static void launch_igemm_bf16(IgemmArgs& a, int math, hipStream_t st) {
    int ntm = cdiv(a.M, BM);
    a.ntn = cdiv(a.Ng, BN);
    a.nwg = ntm * a.ntn;
    a.fNtn = make_fastdiv(a.ntn);
    const int v = 12 + ((BM == 128 && BN == 128) ? 0 : (BM == 64 && BN == 128) ? 1 : (BM == 128 && BN == 64) ? 2 : 3);
    const bool fast = a.taps <= 32 && a.divshift == 0 &&
                      (long long)a.H * a.W * a.ldx * ((long long)a.M / (a.P * a.Q) + 1) < (1ll << 31);
    const bool of32 = math == UP_MATH_BF16S_F32OUT;
    if (of32) math = UP_MATH_BF16S;
    const bool glds_form = math == UP_MATH_BF16S && !of32 && glds_eligible(a, fast);
    a.no_tap_skip = g_tap_skip ? 0 : 1;
    ProfScope prof(glds_form ? v + 8 : v, 2.0 * (double)a.M * (double)a.Ng * (double)a.Ktot_real, st, a.M, a.Ng, a.Ktot, a.nwg,
                   g_prof_on_host() && fast ? live_tap_share(a) : 1.0);
    apply_fold(a, ntm, st);
    const bool split = math == UP_MATH_BF16X3;
    // K slice: 64 on the 64x64 tile (6 -> 12 MFMAs per wave and barrier) measured no faster than 32 (141.7 vs 145.1 TF)
    constexpr int KT = 32;
    if (math == UP_MATH_BF16S) {   // bf16 storage
        // second-generation kernel (bf16s_glds.h): operands HBM -> LDS directly, 32-channel slices (34 KB of LDS, four
        // workgroups per CU), two LDS stages, 16-byte epilogue stores.  Needs 8-channel (16-byte) output rows, 31-bit byte offsets
        // and no strided gather.  (Round 3 also built 64-channel slices, a per-launch slice rule, three stages, 256 x 128 tiles and
        // a K-split of tail tiles; each measured neutral or slower in the 736^2 step — profiles/r03_p, r03_q, r03_s, r03_u — and
        // left the library in round 4.)
        const long long a_bytes = (long long)(a.M / (a.P * a.Q)) * a.H * a.W * a.ldx * 2;
        if (glds_form) {
            a.no_tap_skip = g_tap_skip ? 0 : 1;
            a.perm = nullptr;
            a.x_bytes = (uint32_t)a_bytes;
            if (g_tap_sort && a.taps > 1 && a.taps <= 16 && !a.residual && !a.no_tap_skip) a.perm = tap_sort_perm(a);
            void (*kernel)(IgemmArgs);
            constexpr int BNRED_OCC = 3;   // (the prefetched reduction operands need ~150 VGPRs: three workgroups per CU, no spills)
            if (a.bn_partial)   // fused BatchNorm-backward reduction of the producing layer (up_conv2d_bwd_data_ex)
                kernel = a.perm ? glds::igemm_glds_kernel<BM, BN, true, 32, 2, BNRED_OCC, 0, 0, true> : glds::igemm_glds_kernel<BM, BN, false, 32, 2, BNRED_OCC, 0, 0, true>;
            else
                kernel = a.perm ? glds::igemm_glds_kernel<BM, BN, true, 32, 2, 4> : glds::igemm_glds_kernel<BM, BN, false, 32, 2, 4>;
            a.full_blocks = a.nwg;
            a.parts = 1;
            hipLaunchKernelGGL(kernel, dim3(a.nwg), dim3(256), 0, st, a);
            return;
        }
        if (a.bn_partial || a.res_bits) {   // (up_conv2d_bwd_data_ex checks eligibility first: cannot happen)
            g_extras_dropped = true;
            return;
        }
        a.fSpt = make_fastdiv(a.Cp / KT);
        if (of32 && fast)
            hipLaunchKernelGGL((igemm_bf16_kernel<BM, BN, 2, false, KT, true, true>), dim3(a.nwg), dim3(256), 0, st, a);
        else if (of32)
            hipLaunchKernelGGL((igemm_bf16_kernel<BM, BN, 1, false, KT, true, true>), dim3(a.nwg), dim3(256), 0, st, a);
        else if (fast)
            hipLaunchKernelGGL((igemm_bf16_kernel<BM, BN, 2, false, KT, true>), dim3(a.nwg), dim3(256), 0, st, a);
        else
            hipLaunchKernelGGL((igemm_bf16_kernel<BM, BN, 1, false, KT, true>), dim3(a.nwg), dim3(256), 0, st, a);
        return;
    }
    if (a.Cp % KT != 0) {   // e.g. Cp = 32: fall back to the 32-wide slice
        a.fSpt = make_fastdiv(a.Cp / 32);
        if (fast && split)
            hipLaunchKernelGGL((igemm_bf16_kernel<BM, BN, 2, true, 32>), dim3(a.nwg), dim3(256), 0, st, a);
        else if (fast)
            hipLaunchKernelGGL((igemm_bf16_kernel<BM, BN, 2, false, 32>), dim3(a.nwg), dim3(256), 0, st, a);
        else if (split)
            hipLaunchKernelGGL((igemm_bf16_kernel<BM, BN, 1, true, 32>), dim3(a.nwg), dim3(256), 0, st, a);
        else
            hipLaunchKernelGGL((igemm_bf16_kernel<BM, BN, 1, false, 32>), dim3(a.nwg), dim3(256), 0, st, a);
        return;
    }
    a.fSpt = make_fastdiv(a.Cp / KT);
    if (fast && split)
        hipLaunchKernelGGL((igemm_bf16_kernel<BM, BN, 2, true, KT>), dim3(a.nwg), dim3(256), 0, st, a);
    else if (fast)
        hipLaunchKernelGGL((igemm_bf16_kernel<BM, BN, 2, false, KT>), dim3(a.nwg), dim3(256), 0, st, a);
    else if (split)
        hipLaunchKernelGGL((igemm_bf16_kernel<BM, BN, 1, true, KT>), dim3(a.nwg), dim3(256), 0, st, a);
    else
        hipLaunchKernelGGL((igemm_bf16_kernel<BM, BN, 1, false, KT>), dim3(a.nwg), dim3(256), 0, st, a);
}
// bf16 storage on igemm_big_kernel (bf16s_big.h): 512 threads, (32 TM) x 256 tiles; the caller has checked glds_eligible
template <int TM>
static void launch_igemm_big(IgemmArgs& a, hipStream_t st) {
    constexpr int BM = 32 * TM;
    const int ntm = cdiv(a.M, BM);
    a.ntn = cdiv(a.Ng, 256);
    a.nwg = ntm * a.ntn;
    a.fNtn = make_fastdiv(a.ntn);
    a.no_tap_skip = g_tap_skip ? 0 : 1;
    ProfScope prof(36 + (TM == 8 ? 0 : TM == 6 ? 1 : 2), 2.0 * (double)a.M * (double)a.Ng * (double)a.Ktot_real, st, a.M, a.Ng, a.Ktot,
                   a.nwg, g_prof_on_host() ? live_tap_share(a) : 1.0);
    apply_fold(a, ntm, st);
    a.perm = nullptr;
    a.x_bytes = (uint32_t)((long long)(a.M / (a.P * a.Q)) * a.H * a.W * a.ldx * 2);
    if (g_tap_sort && a.taps > 1 && a.taps <= 16 && !a.residual && !a.no_tap_skip) a.perm = tap_sort_perm(a);
    void (*kernel)(IgemmArgs);
    constexpr int NS3 = TM <= 5 ? 3 : 2;   // three LDS stages where they fit (160-row tiles)
    if (g_big_stages >= 3 && NS3 == 3) {
        if (a.bn_partial)
            kernel = a.perm ? glds::igemm_big_kernel<TM, true, true, NS3> : glds::igemm_big_kernel<TM, false, true, NS3>;
        else
            kernel = a.perm ? glds::igemm_big_kernel<TM, true, false, NS3> : glds::igemm_big_kernel<TM, false, false, NS3>;
    } else if (a.bn_partial)
        kernel = a.perm ? glds::igemm_big_kernel<TM, true, true, 2> : glds::igemm_big_kernel<TM, false, true, 2>;
    else
        kernel = a.perm ? glds::igemm_big_kernel<TM, true, false, 2> : glds::igemm_big_kernel<TM, false, false, 2>;
    a.full_blocks = a.nwg;
    a.parts = 1;
    ++g_count_big;
    hipLaunchKernelGGL(kernel, dim3(a.nwg), dim3(512), 0, st, a);
}
static int run_igemm_bf16(IgemmArgs& a, int math, hipStream_t st) {
    UP_REQUIRE(math == UP_MATH_BF16X3 || math == UP_MATH_BF16 || math == UP_MATH_BF16S || math == UP_MATH_BF16S_F32OUT, UP_ERR_INVALID,
               "bf16 convolution: math mode %d", math);
    UP_REQUIRE(math != UP_MATH_BF16S_F32OUT || (a.ldx % 8 == 0 && !a.residual && !a.stats && !a.o_mode), UP_ERR_INVALID,
               "bf16-storage convolution with fp32 output: ldx=%d must be a multiple of 8; no residual, statistics or parity-class output",
               a.ldx);
    UP_REQUIRE(math != UP_MATH_BF16S || (a.ldx % 8 == 0 && a.ldy % 2 == 0), UP_ERR_INVALID,
               "bf16-storage convolution: ldx=%d must be a multiple of 8 (16-byte rows)", a.ldx);
    UP_REQUIRE(a.Cp % 32 == 0, UP_ERR_UNSUPPORTED, "bf16 convolution: padded channel count %d is not a multiple of 32",
               a.Cp);
    TileChoice t = choose_tile(a.M, a.Ng, a.Ktot, math);
    if (math == UP_MATH_BF16S) {
        // the tile the host was told (pointer-free query) and the one this launch can run must agree wherever per-tile rows are handed over
        IgemmArgs q = a;
        q.x = nullptr; q.y = nullptr; q.w_hi = nullptr; q.residual = nullptr;
        const TileChoice told = choose_tile_bf16s(q);
        t = choose_tile_bf16s(a);
        UP_REQUIRE((told.bm == t.bm && told.bn == t.bn) || !(a.stats || a.bn_partial), UP_ERR_UNSUPPORTED,
                   "bf16-storage convolution: unaligned pointers on a launch whose per-tile rows were sized for %d x %d tiles", told.bm, told.bn);
    }
    if (t.bn == 256) {
        if (t.bm == 256) launch_igemm_big<8>(a, st);
        else if (t.bm == 192) launch_igemm_big<6>(a, st);
        else launch_igemm_big<5>(a, st);
    } else if (t.bm == 128 && t.bn == 128)
        launch_igemm_bf16<128, 128>(a, math, st);
    else if (t.bm == 64 && t.bn == 128)
        launch_igemm_bf16<64, 128>(a, math, st);
    else if (t.bm == 128 && t.bn == 64)
        launch_igemm_bf16<128, 64>(a, math, st);
    else
        launch_igemm_bf16<64, 64>(a, math, st);
    return UP_OK;
}
}  // namespace up

extern "C" int up_conv_tap_visits(const up_conv_desc* d, int data_gradient, double* image_order, double* tap_sorted,
                                  double* live) {
    if (int e = check_desc(d)) return e;
    UP_REQUIRE(image_order && tap_sorted && live, UP_ERR_INVALID, "conv_tap_visits: null output");
    UP_REQUIRE(!data_gradient || !s2_decomposed(d->stride, d->dil), UP_ERR_UNSUPPORTED,
               "conv_tap_visits: the stride-2 data gradient runs as four parity-class launches");
    IgemmArgs a;
    if (data_gradient) {
        if (int e = fill_dgrad_args(a, d, nullptr, nullptr, nullptr)) return e;
    } else if (int e = fill_fwd_args(a, d, nullptr, nullptr, nullptr, nullptr)) {
        return e;
    }
    UP_REQUIRE(a.taps <= 32 && a.divshift == 0, UP_ERR_UNSUPPORTED, "conv_tap_visits: %d taps / strided gather", a.taps);
    const int bm = choose_tile(a.M, a.Ng, a.Ktot).bm;
    *image_order = visited_tap_fraction(a, bm, false, live);
    *tap_sorted = visited_tap_fraction(a, bm, true, nullptr);
    return UP_OK;
}

extern "C" int up_conv2d_bwd_data(const up_conv_desc* d, const float* dy, const float* w_dgrad, float* dx,
                                  const float* add, int ld_add, void* stream) {
    if (int e = check_desc(d)) return e;
    UP_REQUIRE(dy && w_dgrad && dx, UP_ERR_INVALID, "conv2d_bwd_data: null pointer");
    UP_REQUIRE(!add || ld_add >= d->C, UP_ERR_INVALID, "conv2d_bwd_data: ld_add=%d < C=%d", ld_add, d->C);
    IgemmArgs a;
    if (int e = fill_dgrad_args(a, d, dy, w_dgrad, dx)) return e;
    a.residual = add;   // dx = dgrad + add in the epilogue (a second gradient of the same input)
    a.ldr = ld_add;
    if (s2_decomposed(d->stride, d->dil)) {
        UP_REQUIRE(!add, UP_ERR_UNSUPPORTED, "conv2d_bwd_data: an addend with a stride-2 convolution is not implemented");
        // four parity classes, each a dense stride-1-like gather over the dy grid (see S2Classes): 1/4 of the MACs of
        // the single strided launch.  Classes without taps (1x1 stride 2: three of four) are covered by a memset.
        const S2Classes c = s2_classes(d->R, d->S, d->pad, d->C, d->Kp);
        bool empty = false;
        for (int cls = 0; cls < 4; ++cls) empty = empty || c.Rc[cls] * c.Sc[cls] == 0;
        hipStream_t st = as_stream(stream);
        if (empty && hipMemsetAsync(dx, 0, (size_t)d->N * d->H * d->W * d->ldx * sizeof(float), st) != hipSuccess)
            return check_launch("conv2d_bwd_data memset");
        for (int cls = 0; cls < 4; ++cls) {
            const int ph = cls >> 1, pw = cls & 1, tc = c.Rc[cls] * c.Sc[cls];
            const int Hc = (d->H - ph + 1) / 2, Wc = (d->W - pw + 1) / 2;
            if (!tc || Hc <= 0 || Wc <= 0) continue;
            IgemmArgs b = a;
            b.w = w_dgrad + c.base[cls];
            b.M = d->N * Hc * Wc;
            b.P = Hc;   // destination grid of the class
            b.Q = Wc;
            b.S = c.Sc[cls];
            b.taps = tc;
            b.Ktot = tc * d->Kp;
            b.Ktot_real = (long long)tc * d->K;
            b.mul = 1;
            b.tapstep = -1;
            b.off0 = (ph + d->pad - c.r0[cls]) / 2;
            b.off0w = (pw + d->pad - c.s0[cls]) / 2;
            b.divshift = 0;
            b.divmask = 0;
            b.fPQ = make_fastdiv(Hc * Wc);
            b.fQ = make_fastdiv(Wc);
            b.fS = make_fastdiv(c.Sc[cls]);
            b.o_mode = 1;
            b.o_ph = ph;
            b.o_pw = pw;
            b.o_H = d->H;
            b.o_W = d->W;
            b.o_Wc = Wc;
            b.fo_HcWc = make_fastdiv(Hc * Wc);
            b.fo_Wc = make_fastdiv(Wc);
            run_igemm(b, choose_tile(b.M, b.Ng, b.Ktot), st);
        }
        return check_launch("conv2d_bwd_data");
    }
    run_igemm(a, choose_tile(a.M, a.Ng, a.Ktot), as_stream(stream));
    return check_launch("conv2d_bwd_data");
}

// Row tiles of the data-gradient launch of `d` when that launch runs on a kernel whose epilogue can carry the extras of
// up_dgrad_epilogue (masked addend, fused BatchNorm-backward reduction): fp32 -> f32_glds.h with the LDS-transposed epilogue,
// bf16 storage -> bf16s_glds.h; stride 1, 32-aligned output channels, 4- / 8-aligned input channels.  Else 0.
extern "C" int up_conv2d_bwd_data_tiles_math(const up_conv_desc* d, int math) {
    if (!d || check_desc(d)) return 0;
    if (math != UP_MATH_F32 && math != UP_MATH_BF16S) return 0;
    const int q = math == UP_MATH_BF16S ? 8 : 4, eb = math == UP_MATH_BF16S ? 2 : 4;
    if (math == UP_MATH_F32 && (!g_glds32 || !g_glds32_epi)) return 0;
    if (math == UP_MATH_BF16S && !g_glds) return 0;
    if (d->stride != 1 || d->Kp % 32 != 0 || d->R * d->S > 32 || d->C % q != 0 || d->ldx % q != 0 || d->ldy % q != 0 || d->Kp < d->K)
        return 0;
    const long long M = (long long)d->N * d->H * d->W;
    if ((long long)d->N * d->P * d->Q * d->ldy * eb >= (1ll << 31) || (long long)d->C * d->R * d->S * d->Kp * eb >= (1ll << 31) ||
        (long long)d->P * d->Q * d->ldy * ((long long)d->N + 1) >= (1ll << 31) || M * d->ldx >= (1ll << 31))
        return 0;
    if (math == UP_MATH_BF16S) {
        IgemmArgs a;
        if (fill_dgrad_args(a, d, nullptr, nullptr, nullptr) == UP_OK) return cdiv(M, choose_tile_bf16s(a).bm);
    }
    return cdiv(M, choose_tile(M, d->C, d->R * d->S * d->Kp, math).bm);
}
extern "C" int up_conv2d_bwd_data_tiles(const up_conv_desc* d) { return up_conv2d_bwd_data_tiles_math(d, UP_MATH_F32); }
extern "C" int up_conv2d_bwd_data_tiles_grouped(const up_conv_desc* d, int groups) {
    if (up_conv2d_bwd_data_tiles_math(d, UP_MATH_F32) <= 0 || !grouped_ok(d, groups, true)) return 0;
    const long long M = (long long)d->N * d->H * d->W;
    return cdiv(M / groups, choose_tile(M, d->C, d->R * d->S * d->Kp).bm);
}

// up_conv2d_bwd_data / _bf16 (bf16 storage) with the extended epilogue: see up_dgrad_epilogue in the header.
extern "C" int up_conv2d_bwd_data_ex(const up_conv_desc* d, const void* dy, const void* w_dgrad, void* dx,
                                     const up_dgrad_epilogue* ep, int math, void* stream) {
    if (int e = check_desc(d)) return e;
    UP_REQUIRE(dy && w_dgrad && dx && ep, UP_ERR_INVALID, "conv2d_bwd_data_ex: null pointer");
    UP_REQUIRE(math == UP_MATH_F32 || math == UP_MATH_BF16S, UP_ERR_INVALID, "conv2d_bwd_data_ex: math %d (fp32 or bf16 storage)", math);
    UP_REQUIRE(!ep->add || ep->ld_add >= d->C, UP_ERR_INVALID, "conv2d_bwd_data_ex: ld_add=%d < C=%d", ep->ld_add, d->C);
    UP_REQUIRE(!ep->add_relu_bits || ep->add, UP_ERR_INVALID, "conv2d_bwd_data_ex: add_relu_bits without an addend");
    up_bn_reduce_slot* slot = ep->bn;
    const int q = math == UP_MATH_BF16S ? 8 : 4;
    if (slot) slot->folded = 0;
    UP_REQUIRE(!slot || (slot->dgamma == nullptr) == (slot->dbeta == nullptr), UP_ERR_INVALID,
               "conv2d_bwd_data_ex: dgamma and dbeta of a BatchNorm slot come together");
    UP_REQUIRE(!slot || (slot->y && slot->mean && slot->invstd && slot->partial && slot->C == d->C && slot->ld >= d->C && slot->ld % q == 0),
               UP_ERR_INVALID, "conv2d_bwd_data_ex: bad BatchNorm slot (C=%d vs %d, ld=%d)", slot ? slot->C : 0, d->C, slot ? slot->ld : 0);
    UP_REQUIRE(!(slot || ep->add_relu_bits) || up_conv2d_bwd_data_tiles_math(d, math) > 0, UP_ERR_UNSUPPORTED,
               "conv2d_bwd_data_ex: this launch cannot carry a masked addend / a fused reduction (up_conv2d_bwd_data_tiles_math = 0)");
    const int groups = ep->groups > 1 ? ep->groups : 1;
    UP_REQUIRE(groups == 1 || (math == UP_MATH_F32 && up_conv2d_bwd_data_tiles_grouped(d, groups) > 0), UP_ERR_UNSUPPORTED,
               "conv2d_bwd_data_ex: this launch cannot be tiled per row group (up_conv2d_bwd_data_tiles_grouped = 0)");
    IgemmArgs a;
    if (int e = fill_dgrad_args(a, d, static_cast<const float*>(dy), static_cast<const float*>(w_dgrad), static_cast<float*>(dx))) return e;
    a.residual = static_cast<const float*>(ep->add);
    a.ldr = ep->ld_add;
    a.res_bits = ep->add_relu_bits;
    if (slot) {
        a.bn_y = static_cast<const float*>(slot->y);
        a.bn_ld = slot->ld;
        a.bn_C = slot->C;
        a.bn_bits = slot->relu_bits;
        a.bn_mean = slot->mean;
        a.bn_invstd = slot->invstd;
        a.bn_partial = slot->partial;
        a.bn_grp_stride = groups > 1 ? slot->group_stride : 0;
    }
    a.grp_rows = groups > 1 ? a.M / groups : 0;
    g_group_refused = false;
    const uintptr_t ptrs = reinterpret_cast<uintptr_t>(dy) | reinterpret_cast<uintptr_t>(w_dgrad) | reinterpret_cast<uintptr_t>(dx) |
                           reinterpret_cast<uintptr_t>(ep->add) | (slot ? reinterpret_cast<uintptr_t>(slot->y) : 0);
    UP_REQUIRE(!(slot || ep->add_relu_bits) || ((ptrs & 15) == 0 && (!ep->add || ep->ld_add % q == 0)), UP_ERR_UNSUPPORTED,
               "conv2d_bwd_data_ex: 16-byte alignment");
    g_extras_dropped = false;
    g_fold_req = FoldRequest();
    if (slot && slot->dgamma && (groups == 1 || slot->gsum)) g_fold_req.bwd = slot;   // the launch also finishes dgamma / dbeta (bn_fold.h)
    if (math == UP_MATH_BF16S) {
        UP_REQUIRE(!s2_decomposed(d->stride, d->dil) || !(slot || ep->add_relu_bits), UP_ERR_UNSUPPORTED, "conv2d_bwd_data_ex: stride 2");
        a.w = nullptr;
        a.w_hi = static_cast<const uint16_t*>(w_dgrad);
        a.w_lo = nullptr;
        const int e = run_igemm_bf16(a, math, as_stream(stream));
        g_fold_req = FoldRequest();
        if (e) return e;
    } else {
        UP_REQUIRE(!s2_decomposed(d->stride, d->dil), UP_ERR_UNSUPPORTED, "conv2d_bwd_data_ex: stride-2 convolutions use up_conv2d_bwd_data");
        run_igemm(a, choose_tile(a.M, a.Ng, a.Ktot), as_stream(stream));
        g_fold_req = FoldRequest();
    }
    UP_REQUIRE(!g_group_refused, UP_ERR_UNSUPPORTED, "conv2d_bwd_data_ex: the launch did not qualify for the kernel with row groups; "
               "nothing was launched");
    UP_REQUIRE(!g_extras_dropped, UP_ERR_UNSUPPORTED, "conv2d_bwd_data_ex: the launch did not qualify for a kernel with the requested epilogue extras; nothing was launched");
    return check_launch("conv2d_bwd_data_ex");
}

extern "C" int up_pack_weights_bf16(const up_conv_desc* d, const float* w, uint16_t* fwd_hi, uint16_t* fwd_lo,
                                    uint16_t* dgrad_hi, uint16_t* dgrad_lo, void* stream) {
    if (int e = check_desc(d)) return e;
    UP_REQUIRE(w && (fwd_hi == nullptr) == (fwd_lo == nullptr) && (dgrad_hi == nullptr) == (dgrad_lo == nullptr),
               UP_ERR_INVALID, "pack_weights_bf16: hi/lo planes come in pairs");
    int taps = d->R * d->S;
    if (fwd_hi) {
        long long total = (long long)d->K * taps * d->Cp;
        hipLaunchKernelGGL(pack_split_kernel, dim3(cdiv(total, 256)), dim3(256), 0, as_stream(stream), w, fwd_hi, fwd_lo,
                           d->K, d->C, d->Cp, taps, total, 0);
    }
    if (dgrad_hi) {
        UP_REQUIRE(d->Kp % 4 == 0 && d->Kp >= d->K, UP_ERR_INVALID, "pack_weights_bf16: Kp=%d invalid", d->Kp);
        long long total = (long long)d->C * taps * d->Kp;
        hipLaunchKernelGGL(pack_split_kernel, dim3(cdiv(total, 256)), dim3(256), 0, as_stream(stream), w, dgrad_hi,
                           dgrad_lo, d->K, d->C, d->Kp, taps, total, 1);
    }
    return check_launch("pack_weights_bf16");
}

namespace up {
// up_pack_weights_bf16 for many parameters in one launch (blockIdx.y = job), element order and rounding of pack_split_kernel
__global__ void __launch_bounds__(256) pack_split_batched_kernel(const up_pack_job_bf16* jobs) {
    const up_pack_job_bf16 jb = jobs[blockIdx.y];
    const int taps = jb.taps;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    // (lo planes are only read by the split-bf16 arithmetic, UP_MATH_BF16X3: a job without them — fwd_lo / dgrad_lo NULL — skips
    //  a third of the bytes)
    auto put = [](uint16_t* hi, uint16_t* lo, size_t e, float v) {
        uint32_t h, l;
        split_bf16x2(v, 0.f, h, l);
        hi[e] = (uint16_t)(h & 0xffffu);
        if (lo) lo[e] = (uint16_t)(l & 0xffffu);
    };
    if (jb.fwd_hi) {   // rows (k, tap) of Cp channels — the row scheme of pack_batched_kernel
        const int rows = jb.K * taps;
        for (int base = blockIdx.x * 32; base < rows; base += gridDim.x * 32) {
            const int end = min(base + 32, rows);
            for (int rt = base + wave; rt < end; rt += 4) {
                const int k = rt / taps, tap = rt - k * taps;
                const float* src = jb.w + (size_t)k * jb.C * taps + tap;
                for (int c0 = lane; c0 < jb.Cp; c0 += 256) {   // four loads in flight, then the stores (see pack_batched_kernel)
                    float v[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int ci = c0 + 64 * u;
                        v[u] = ci < jb.C ? src[(size_t)ci * taps] : 0.f;
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u)
                        if (c0 + 64 * u < jb.Cp) put(jb.fwd_hi, jb.fwd_lo, (size_t)rt * jb.Cp + c0 + 64 * u, v[u]);
                }
            }
        }
    }
    if (jb.dgrad_hi) {   // rows (c, tap) of Kp output channels, transposed through LDS like pack_batched_kernel's
        __shared__ float tile[32][65];
        const int rows = jb.C * taps;
        const size_t kstride = (size_t)jb.C * taps;
        const int r31 = lane & 31, khalf = lane >> 5;
        for (int base = blockIdx.x * 32; base < rows; base += gridDim.x * 32) {
            const int end = min(base + 32, rows);
            for (int k0 = 0; k0 < jb.Kp; k0 += 64) {
                float v[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int k = k0 + wave * 16 + 2 * j + khalf;
                    v[j] = (k < jb.K && base + r31 < end) ? jb.w[(size_t)k * kstride + base + r31] : 0.f;
                }
#pragma unroll
                for (int j = 0; j < 8; ++j) tile[r31][wave * 16 + 2 * j + khalf] = v[j];
                __syncthreads();
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int rt = base + wave + 4 * j;
                    if (rt < end && k0 + lane < jb.Kp) put(jb.dgrad_hi, jb.dgrad_lo, (size_t)rt * jb.Kp + k0 + lane, tile[wave + 4 * j][lane]);
                }
                __syncthreads();
            }
        }
    }
}
}  // namespace up

extern "C" int up_pack_weights_bf16_batched(const up_pack_job_bf16* jobs_device, int njobs, void* stream) {
    UP_REQUIRE(jobs_device && njobs > 0 && njobs <= 65535, UP_ERR_INVALID, "pack_weights_bf16_batched: bad job table");
    hipLaunchKernelGGL(pack_split_batched_kernel, dim3(144, njobs), dim3(256), 0, as_stream(stream), jobs_device);
    return check_launch("pack_weights_bf16_batched");
}

extern "C" int up_conv2d_fwd_bf16(const up_conv_desc* d, const float* x, const uint16_t* w_hi, const uint16_t* w_lo,
                                  float* y, const up_conv_epilogue* ep, int math, void* stream) {
    if (int e = check_desc(d)) return e;
    UP_REQUIRE(x && w_hi && y && (w_lo || math != UP_MATH_BF16X3), UP_ERR_INVALID, "conv2d_fwd_bf16: null pointer");
    IgemmArgs a;
    g_fold_req = FoldRequest();
    if (int e = fill_fwd_args(a, d, x, nullptr, y, ep)) return e;
    a.w_hi = w_hi;
    a.w_lo = w_lo;
    const int e = run_igemm_bf16(a, math, as_stream(stream));
    g_fold_req = FoldRequest();
    if (e) return e;
    return check_launch("conv2d_fwd_bf16");
}

extern "C" int up_conv2d_bwd_data_bf16(const up_conv_desc* d, const float* dy, const uint16_t* w_hi,
                                       const uint16_t* w_lo, float* dx, const float* add, int ld_add, int math,
                                       void* stream) {
    if (int e = check_desc(d)) return e;
    UP_REQUIRE(dy && w_hi && dx && (w_lo || math != UP_MATH_BF16X3), UP_ERR_INVALID, "conv2d_bwd_data_bf16: null pointer");
    UP_REQUIRE(!add || ld_add >= d->C, UP_ERR_INVALID, "conv2d_bwd_data_bf16: ld_add=%d < C=%d", ld_add, d->C);
    IgemmArgs a;
    if (int e = fill_dgrad_args(a, d, dy, nullptr, dx)) return e;
    a.residual = add;
    a.ldr = ld_add;
    a.w_hi = w_hi;
    a.w_lo = w_lo;
    if (int e = run_igemm_bf16(a, math, as_stream(stream))) return e;
    return check_launch("conv2d_bwd_data_bf16");
}

namespace up {
#ifdef UP_PROBE
static void* g_wgrad_dbg = nullptr;
static int g_wgrad_grid = 0;
static bool g_wgrad_single = false;
#endif
struct WgradPlan {
    int bm, bn, ntm, ntn, splits, rows_per_split;
};
static WgradPlan plan_wgrad(const up_conv_desc* d, int slice = BK) {
    WgradPlan p;
    int ncols = d->R * d->S * d->Cp;
    // (64x64 tiles everywhere — half the splits and slab traffic, twice the operand traffic, four workgroups per CU —
    //  measured 0.4 ms per step slower, profiles/r02_f_knob_ab.txt)
    p.bm = d->K <= 64 ? 64 : 128;
    p.bn = ncols <= 64 ? 64 : 128;
    p.ntm = cdiv(d->K, p.bm);
    p.ntn = cdiv(ncols, p.bn);
    int64_t M = (int64_t)d->N * d->P * d->Q;
    int tiles = p.ntm * p.ntn;
    // Split K (pixels) so that the launch fills the CUs evenly: workgroups are dispatched round-robin, two are
    // resident per CU, so the launch time follows ceil(WGs / CUs).  Take the fewest splits (least slab traffic for
    // the reduce pass) whose CU utilisation WGs / (CUs * ceil(WGs / CUs)) reaches 95 %, at most 4 workgroups per CU
    // and at least 256 pixel rows per split.  (Measured: 512-workgroup launches 82.5 ms/step, 1024: 85.2, 2048: 87.4.)
    const int cus = cu_count();
    int64_t max_splits = (M + 255) / 256;
    if (max_splits < 1) max_splits = 1;
    int splits = 1;
    double best = 0.0;
    for (int sp = 1; sp <= max_splits && (int64_t)tiles * sp <= 4 * cus; ++sp) {
        const int wgs = tiles * sp;
        // one workgroup per CU leaves nothing to overlap its barriers with (256-workgroup launches: 84.9 ms/step)
        const double util = (double)wgs / ((double)cus * cdiv(wgs, cus)) * (wgs > (g_wgrad_per_cu - 1) * cus ? 1.0 : 0.8);
        if (util > best + 1e-9) {
            best = util;
            splits = sp;
        }
        if (util >= 0.95) break;
    }
    int64_t rps = (M + splits - 1) / splits;
    rps = (rps + slice - 1) / slice * slice;
    p.rows_per_split = (int)rps;
    p.splits = (int)((M + rps - 1) / rps);
    return p;
}

// ---- live rectangles of the weight-gradient column tiles (wgrad_kernel<..., RECT>) ----------------------
// Filter tap (r, s) reads a real input pixel only for output rows p with 0 <= p*stride - pad + r*dil < H (columns
// likewise): a rectangle of the output grid.  A column tile of the weight-gradient GEMM holds one or two taps, so
// its reduction can run over the bounding rectangle of its taps instead of all P x Q pixels of every image — on the
// dilation-18 WASP convolution the corner taps see 5 x 5 of 23 x 23 pixels.  Knob "wgrad_rect" (UP_WGRAD_RECT),
// default off until measured.
struct WgradRectKey {
    int N, H, W, P, Q, R, S, stride, pad, dil, Cp, bn, splits;
    bool operator<(const WgradRectKey& o) const { return memcmp(this, &o, sizeof(*this)) < 0; }
};
// fills `tab` (ntn x WGRAD_RECT_INTS); returns the share of (pixel, column tile) pairs the reduction still visits
static double wgrad_rect_table(const up_conv_desc* d, const WgradPlan& p, std::vector<int>& tab) {
    const int ncols = d->R * d->S * d->Cp;
    tab.assign((size_t)p.ntn * WGRAD_RECT_INTS, 0);
    auto live = [](int size_in, int size_out, int stride, int pad, int off, int& lo, int& hi) {
        // 0 <= o*stride - pad + off <= size_in - 1
        const int a = pad - off;
        lo = a <= 0 ? 0 : (a + stride - 1) / stride;
        const int b = size_in - 1 + pad - off;
        hi = b < 0 ? -1 : b / stride;
        if (hi > size_out - 1) hi = size_out - 1;
    };
    double visited = 0.0;
    for (int nt = 0; nt < p.ntn; ++nt) {
        const int c_lo = nt * p.bn, c_hi = std::min(ncols, (nt + 1) * p.bn) - 1;
        int pl = d->P, ph = -1, ql = d->Q, qh = -1;
        for (int t = c_lo / d->Cp; t <= c_hi / d->Cp; ++t) {
            int lo, hi;
            live(d->H, d->P, d->stride, d->pad, (t / d->S) * d->dil, lo, hi);
            if (lo <= hi) {
                pl = std::min(pl, lo);
                ph = std::max(ph, hi);
            }
            live(d->W, d->Q, d->stride, d->pad, (t % d->S) * d->dil, lo, hi);
            if (lo <= hi) {
                ql = std::min(ql, lo);
                qh = std::max(qh, hi);
            }
        }
        const int hr = ph >= pl ? ph - pl + 1 : 0, wr = qh >= ql ? qh - ql + 1 : 0;
        const int rows = (hr && wr) ? hr : 0, cols = (hr && wr) ? wr : 0;
        const long long mt = (long long)d->N * rows * cols;
        long long rps = (mt + p.splits - 1) / p.splits;
        rps = std::max<long long>(BK, (rps + BK - 1) / BK * BK);
        const FastDiv fhw = make_fastdiv(rows * cols), fw = make_fastdiv(cols);
        int* e = &tab[(size_t)nt * WGRAD_RECT_INTS];
        e[0] = rows ? pl : 0;
        e[1] = rows ? ql : 0;
        e[2] = rows;
        e[3] = cols;
        e[4] = (int)fhw.mul; e[5] = (int)fhw.shr; e[6] = (int)fhw.d;
        e[7] = (int)fw.mul;  e[8] = (int)fw.shr;  e[9] = (int)fw.d;
        e[10] = (int)rps;
        e[11] = (int)mt;
        visited += (double)mt;
    }
    return visited / ((double)p.ntn * d->N * d->P * d->Q);
}
static const int* wgrad_rect_device(const up_conv_desc* d, const WgradPlan& p) {
    static std::mutex mu;
    static std::map<WgradRectKey, int*> table;
    WgradRectKey key;
    memset(&key, 0, sizeof(key));
    key.N = d->N; key.H = d->H; key.W = d->W; key.P = d->P; key.Q = d->Q; key.R = d->R; key.S = d->S;
    key.stride = d->stride; key.pad = d->pad; key.dil = d->dil; key.Cp = d->Cp; key.bn = p.bn; key.splits = p.splits;
    std::lock_guard<std::mutex> lock(mu);
    auto it = table.find(key);
    if (it != table.end()) return it->second;
    if (table.size() >= 1024) return nullptr;   // bounded by refusing new entries (see tap_sort_perm): the plain reduction range
    std::vector<int> tab;
    int* dev = nullptr;
    if (wgrad_rect_table(d, p, tab) < 0.999) {   // every tile sees the whole image (1x1, unpadded): keep the plain form
        const size_t bytes = tab.size() * sizeof(int);
#ifdef UP_EMU
        dev = static_cast<int*>(malloc(bytes));
        memcpy(dev, tab.data(), bytes);
#else
        if (hipMalloc(reinterpret_cast<void**>(&dev), bytes) != hipSuccess ||
            hipMemcpy(dev, tab.data(), bytes, hipMemcpyHostToDevice) != hipSuccess) {
            (void)hipGetLastError();
            dev = nullptr;
        }
#endif
    }
    table[key] = dev;
    return dev;
}
}  // namespace up

extern "C" int up_conv_wgrad_visits(const up_conv_desc* d, double* rect_fraction) {
    if (int e = check_desc(d)) return e;
    UP_REQUIRE(rect_fraction, UP_ERR_INVALID, "conv_wgrad_visits: null output");
    std::vector<int> tab;
    *rect_fraction = wgrad_rect_table(d, plan_wgrad(d), tab);
    return UP_OK;
}

extern "C" size_t up_conv2d_bwd_weight_workspace(const up_conv_desc* d) {
    if (!d || check_desc(d)) return 0;
    const size_t slab = (size_t)d->K * d->R * d->S * d->Cp * sizeof(float);
    const size_t a = (size_t)plan_wgrad(d).splits * slab, b = (size_t)plan_wgrad(d, 64).splits * slab;
    // + the bias gradient's partial rows (one per 256 pixel rows), behind the split-K slabs
    return (a > b ? a : b) + 64 + (size_t)cdiv((int64_t)d->N * d->P * d->Q, 256) * d->K * sizeof(float);
}

namespace up {
static int conv2d_bwd_weight_impl(const up_conv_desc* d, const float* x, const float* dy, float* dw, float* dbias,
                                  void* workspace, size_t workspace_bytes, int bf16, void* stream, int accumulate = 0);
}
extern "C" int up_conv2d_bwd_weight_acc(const up_conv_desc* d, const void* x, const void* dy, float* dw, float* dbias,
                                        void* workspace, size_t workspace_bytes, int math, int accumulate, void* stream) {
    UP_REQUIRE(math == UP_MATH_F32 || math == UP_MATH_BF16 || math == UP_MATH_BF16S, UP_ERR_INVALID,
               "conv2d_bwd_weight_acc: math mode %d", math);
    if (math == UP_MATH_BF16S) {
        UP_REQUIRE(d && d->Cp % 8 == 0 && d->ldx % 8 == 0 && d->ldy % 8 == 0, UP_ERR_INVALID,
                   "conv2d_bwd_weight_acc: bf16 storage needs Cp, ldx and ldy to be multiples of 8 (16-byte channel groups)");
        UP_REQUIRE((int64_t)d->N * d->H * d->W * d->ldx < (1ll << 31) && (int64_t)d->N * d->P * d->Q * d->ldy < (1ll << 31),
                   UP_ERR_UNSUPPORTED, "conv2d_bwd_weight_acc: more than 2^31 elements");
    }
    return conv2d_bwd_weight_impl(d, static_cast<const float*>(x), static_cast<const float*>(dy), dw, dbias, workspace,
                                  workspace_bytes, math == UP_MATH_F32 ? 0 : math == UP_MATH_BF16 ? 1 : 2, stream,
                                  accumulate ? 1 : 0);
}
extern "C" int up_conv2d_bwd_weight(const up_conv_desc* d, const float* x, const float* dy, float* dw, float* dbias,
                                    void* workspace, size_t workspace_bytes, void* stream) {
    return conv2d_bwd_weight_impl(d, x, dy, dw, dbias, workspace, workspace_bytes, 0, stream);
}
extern "C" int up_conv2d_bwd_weight_bf16(const up_conv_desc* d, const float* x, const float* dy, float* dw, float* dbias,
                                         void* workspace, size_t workspace_bytes, void* stream) {
    return conv2d_bwd_weight_impl(d, x, dy, dw, dbias, workspace, workspace_bytes, 1, stream);
}
extern "C" int up_conv2d_bwd_weight_bf16s(const up_conv_desc* d, const void* x, const void* dy, float* dw, float* dbias,
                                          void* workspace, size_t workspace_bytes, void* stream) {
    UP_REQUIRE(d && d->Cp % 8 == 0 && d->ldx % 8 == 0 && d->ldy % 8 == 0, UP_ERR_INVALID,
               "conv2d_bwd_weight_bf16s: Cp, ldx and ldy must be multiples of 8 (16-byte channel groups)");
    UP_REQUIRE((int64_t)d->N * d->H * d->W * d->ldx < (1ll << 31) && (int64_t)d->N * d->P * d->Q * d->ldy < (1ll << 31),
               UP_ERR_UNSUPPORTED, "conv2d_bwd_weight_bf16s: more than 2^31 elements");
    return conv2d_bwd_weight_impl(d, static_cast<const float*>(x), static_cast<const float*>(dy), dw, dbias, workspace,
                                  workspace_bytes, 2, stream);
}
static int up::conv2d_bwd_weight_impl(const up_conv_desc* d, const float* x, const float* dy, float* dw, float* dbias,
                                      void* workspace, size_t workspace_bytes, int bf16, void* stream, int accumulate) {
    if (int e = check_desc(d)) return e;
    UP_REQUIRE(x && dy && dw && workspace, UP_ERR_INVALID, "conv2d_bwd_weight: null pointer");
    UP_REQUIRE(d->ldy % 4 == 0, UP_ERR_INVALID, "conv2d_bwd_weight: ldy=%d must be a multiple of 4", d->ldy);
    WgradPlan p = plan_wgrad(d, bf16 == 2 ? 64 : BK);
    size_t need = (size_t)p.splits * d->K * d->R * d->S * d->Cp * sizeof(float);
    UP_REQUIRE(workspace_bytes >= need, UP_ERR_WORKSPACE, "conv2d_bwd_weight: workspace %zu < %zu", workspace_bytes,
               need);
    hipStream_t st = as_stream(stream);
    WgradArgs a;
    memset(&a, 0, sizeof(a));
    a.x = x;
    a.dy = dy;
    a.slab = (float*)workspace;
    a.M = d->N * d->P * d->Q;
    a.K = d->K;
    a.Cp = d->Cp;
    a.Ncols = d->R * d->S * d->Cp;
    a.H = d->H;
    a.W = d->W;
    a.P = d->P;
    a.Q = d->Q;
    a.ldx = d->ldx;
    a.ldy = d->ldy;
    a.S = d->S;
    a.stride = d->stride;
    a.pad = d->pad;
    a.dil = d->dil;
    a.rows_per_split = p.rows_per_split;
    a.ntn = p.ntn;
    a.fPQ = make_fastdiv(d->P * d->Q);
    a.fQ = make_fastdiv(d->Q);
    a.fCp = make_fastdiv(d->Cp);
    a.fS = make_fastdiv(d->S);
    a.fNtn = make_fastdiv(p.ntn);
    a.fTiles = make_fastdiv(p.ntm * p.ntn);
    a.nwg = p.ntm * p.ntn * p.splits;
    dim3 grid(a.nwg);
    // bf16-operand form (BASELINE configs[4] arithmetic): 32-bit element offsets inside the kernel
    const bool use_bf16 = bf16 && (int64_t)d->N * d->H * d->W * d->ldx < (1ll << 31) &&
                          (int64_t)d->N * d->P * d->Q * d->ldy < (1ll << 31);
    if (use_bf16) {
        const int v = (p.bm == 128 && p.bn == 128) ? 16 : (p.bm == 128 && p.bn == 64) ? 17 : (p.bm == 64 && p.bn == 128) ? 18 : 19;
        const long long xb = (long long)d->N * d->H * d->W * d->ldx * 2, dyb = (long long)d->N * d->P * d->Q * d->ldy * 2;
        const bool glds_form = bf16 == 2 && g_glds && xb < (1ll << 31) && dyb < (1ll << 31) &&
                               ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(dy)) & 15) == 0;
        ProfScope prof(glds_form ? v + 8 : v, 2.0 * (double)a.M * (double)d->K * (double)d->R * d->S * d->C, st, d->K, a.Ncols, a.M,
                       a.nwg);
        a.rect = (g_wgrad_rect && (int64_t)d->N * d->P * d->Q < (1ll << 30)) ? wgrad_rect_device(d, p) : nullptr;
        if (glds_form) {
            // second generation (bf16s_glds.h): [pixel][channel] slices HBM -> LDS directly, transposing fragment reads
            a.x_bytes = (uint32_t)xb;
            a.dy_bytes = (uint32_t)dyb;
            void (*kernel)(WgradArgs);
            if (p.bm == 128 && p.bn == 128) kernel = glds::wgrad_glds_kernel<128, 128, 64, 2, 2>;
            else if (p.bm == 128 && p.bn == 64) kernel = glds::wgrad_glds_kernel<128, 64, 64, 2, 2>;
            else if (p.bm == 64 && p.bn == 128) kernel = glds::wgrad_glds_kernel<64, 128, 64, 2, 2>;
            else kernel = glds::wgrad_glds_kernel<64, 64, 64, 2, 2>;
            hipLaunchKernelGGL(kernel, grid, dim3(256), 0, st, a);
        } else if (bf16 == 2) {
            if (p.bm == 128 && p.bn == 128)
                hipLaunchKernelGGL((wgrad_bf16_kernel<128, 128, true>), grid, dim3(256), 0, st, a);
            else if (p.bm == 128 && p.bn == 64)
                hipLaunchKernelGGL((wgrad_bf16_kernel<128, 64, true>), grid, dim3(256), 0, st, a);
            else if (p.bm == 64 && p.bn == 128)
                hipLaunchKernelGGL((wgrad_bf16_kernel<64, 128, true>), grid, dim3(256), 0, st, a);
            else
                hipLaunchKernelGGL((wgrad_bf16_kernel<64, 64, true>), grid, dim3(256), 0, st, a);
        } else if (p.bm == 128 && p.bn == 128)
            hipLaunchKernelGGL((wgrad_bf16_kernel<128, 128>), grid, dim3(256), 0, st, a);
        else if (p.bm == 128 && p.bn == 64)
            hipLaunchKernelGGL((wgrad_bf16_kernel<128, 64>), grid, dim3(256), 0, st, a);
        else if (p.bm == 64 && p.bn == 128)
            hipLaunchKernelGGL((wgrad_bf16_kernel<64, 128>), grid, dim3(256), 0, st, a);
        else
            hipLaunchKernelGGL((wgrad_bf16_kernel<64, 64>), grid, dim3(256), 0, st, a);
    } else {
        const int v = (p.bm == 128 && p.bn == 128) ? 8 : (p.bm == 128 && p.bn == 64) ? 9 : (p.bm == 64 && p.bn == 128) ? 10 : 11;
        // direct-to-LDS form (f32_glds.h): 31-bit byte offsets, 16-byte aligned operands
        const long long xb = (long long)d->N * d->H * d->W * d->ldx * 4, dyb = (long long)d->N * d->P * d->Q * d->ldy * 4;
        bool glds32_form = g_glds32_wgrad && xb < (1ll << 31) && dyb < (1ll << 31) && (int64_t)d->N * d->P * d->Q < (1ll << 30) &&
                           ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(dy)) & 15) == 0;
#ifdef UP_PROBE
        glds32_form = glds32_form && !g_wgrad_single && !g_wgrad_dbg;
#endif
        ProfScope prof(glds32_form ? v + 24 : v, 2.0 * (double)a.M * (double)d->K * (double)d->R * d->S * d->C, st, d->K, a.Ncols,
                       a.M, a.nwg);
        if (glds32_form) {
            a.x_bytes = (uint32_t)xb;
            a.dy_bytes = (uint32_t)dyb;
            a.rect = g_wgrad_rect ? wgrad_rect_device(d, p) : nullptr;
            const bool single = a.nwg > 2 * cu_count();   // (wgrad_kernel's rule: one LDS stage when more than two workgroups per CU queue up)
            ++g_count_wgrad32;
            if (single) ++g_count_wgrad32_st1;
            void (*kernel)(WgradArgs);
            if (p.bm == 128 && p.bn == 128)
                kernel = single ? glds::wgrad_glds32_kernel<128, 128, 1, 3> : glds::wgrad_glds32_kernel<128, 128, 2, 2>;
            else if (p.bm == 128 && p.bn == 64)
                kernel = single ? glds::wgrad_glds32_kernel<128, 64, 1, 4> : glds::wgrad_glds32_kernel<128, 64, 2, 3>;
            else if (p.bm == 64 && p.bn == 128)
                kernel = single ? glds::wgrad_glds32_kernel<64, 128, 1, 4> : glds::wgrad_glds32_kernel<64, 128, 2, 3>;
            else
                kernel = single ? glds::wgrad_glds32_kernel<64, 64, 1, 4> : glds::wgrad_glds32_kernel<64, 64, 2, 4>;
            hipLaunchKernelGGL(kernel, grid, dim3(256), 0, st, a);
        } else
#ifdef UP_PROBE
        if (g_wgrad_single && !g_wgrad_dbg) {   // the older single-buffer loop, for comparison
            if (p.bm == 128 && p.bn == 128)
                hipLaunchKernelGGL((wgrad_kernel<128, 128, 64>), grid, dim3(256), 0, st, a);
            else if (p.bm == 64 && p.bn == 128)
                hipLaunchKernelGGL((wgrad_kernel<64, 128, 64>), grid, dim3(256), 0, st, a);
            else
                hipLaunchKernelGGL((wgrad_kernel<128, 64, 64>), grid, dim3(256), 0, st, a);
        } else
        if (g_wgrad_dbg) {   // tools/gpu/igemm_probe.hip: per-block timeline
            a.dbg = g_wgrad_dbg;
            g_wgrad_grid = a.nwg;
            if (p.bm == 128 && p.bn == 128)
                hipLaunchKernelGGL((wgrad_kernel<128, 128, 32>), grid, dim3(256), 0, st, a);
            else if (p.bm == 64 && p.bn == 128)
                hipLaunchKernelGGL((wgrad_kernel<64, 128, 32>), grid, dim3(256), 0, st, a);
            else
                hipLaunchKernelGGL((wgrad_kernel<128, 64, 32>), grid, dim3(256), 0, st, a);
        } else
#endif
        {
            // two LDS buffers (64 KB, two workgroups per CU) when the whole grid is resident at once; launches with more
            // workgroups (layers with many weight tiles) keep the 32 KB single-buffer loop and 3-4 per CU
            // (probe, 3x3 512->512: 115 vs 104 TFLOP/s; in the network the double buffer is 0.5 % faster overall)
            const bool single = a.nwg > 2 * cu_count();
            a.rect = (g_wgrad_rect && (int64_t)d->N * d->P * d->Q < (1ll << 30)) ? wgrad_rect_device(d, p) : nullptr;
            if (a.rect && single) {
                if (p.bm == 128 && p.bn == 128)
                    hipLaunchKernelGGL((wgrad_kernel<128, 128, 64, true>), grid, dim3(256), 0, st, a);
                else if (p.bm == 128 && p.bn == 64)
                    hipLaunchKernelGGL((wgrad_kernel<128, 64, 64, true>), grid, dim3(256), 0, st, a);
                else if (p.bm == 64 && p.bn == 128)
                    hipLaunchKernelGGL((wgrad_kernel<64, 128, 64, true>), grid, dim3(256), 0, st, a);
                else
                    hipLaunchKernelGGL((wgrad_kernel<64, 64, 64, true>), grid, dim3(256), 0, st, a);
            } else if (a.rect) {
                if (p.bm == 128 && p.bn == 128)
                    hipLaunchKernelGGL((wgrad_kernel<128, 128, 0, true>), grid, dim3(256), 0, st, a);
                else if (p.bm == 128 && p.bn == 64)
                    hipLaunchKernelGGL((wgrad_kernel<128, 64, 0, true>), grid, dim3(256), 0, st, a);
                else if (p.bm == 64 && p.bn == 128)
                    hipLaunchKernelGGL((wgrad_kernel<64, 128, 0, true>), grid, dim3(256), 0, st, a);
                else
                    hipLaunchKernelGGL((wgrad_kernel<64, 64, 0, true>), grid, dim3(256), 0, st, a);
            } else if (single) {
                if (p.bm == 128 && p.bn == 128)
                    hipLaunchKernelGGL((wgrad_kernel<128, 128, 64>), grid, dim3(256), 0, st, a);
                else if (p.bm == 128 && p.bn == 64)
                    hipLaunchKernelGGL((wgrad_kernel<128, 64, 64>), grid, dim3(256), 0, st, a);
                else if (p.bm == 64 && p.bn == 128)
                    hipLaunchKernelGGL((wgrad_kernel<64, 128, 64>), grid, dim3(256), 0, st, a);
                else
                    hipLaunchKernelGGL((wgrad_kernel<64, 64, 64>), grid, dim3(256), 0, st, a);
            } else if (p.bm == 128 && p.bn == 128)
                hipLaunchKernelGGL((wgrad_kernel<128, 128>), grid, dim3(256), 0, st, a);
            else if (p.bm == 128 && p.bn == 64)
                hipLaunchKernelGGL((wgrad_kernel<128, 64>), grid, dim3(256), 0, st, a);
            else if (p.bm == 64 && p.bn == 128)
                hipLaunchKernelGGL((wgrad_kernel<64, 128>), grid, dim3(256), 0, st, a);
            else
                hipLaunchKernelGGL((wgrad_kernel<64, 64>), grid, dim3(256), 0, st, a);
        }
    }
    long long total4 = (long long)d->K * a.Ncols / 4;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(cdiv(total4, 256)), dim3(256), 0, st, (const float*)workspace, dw,
                       p.splits, d->K, d->C, d->Cp, d->R * d->S, total4, accumulate);
    if (dbias) {
        const bool quads = d->ldy % 4 == 0 && (reinterpret_cast<uintptr_t>(dy) & 15) == 0;
        const int rpb = quads ? 256 : 1024;
        dim3 g(cdiv(a.M, rpb), cdiv(d->K, 64));
        const size_t off = (need + 63) / 64 * 64;
        UP_REQUIRE(workspace_bytes >= off + (size_t)g.x * d->K * sizeof(float), UP_ERR_WORKSPACE,
                   "conv2d_bwd_weight: workspace %zu too small for the bias gradient's partial rows", workspace_bytes);
        float* part = reinterpret_cast<float*>(static_cast<char*>(workspace) + off);
        if (quads && bf16 == 2)
            hipLaunchKernelGGL(colsum4_kernel<bf16_t>, g, dim3(256), 0, st, reinterpret_cast<const bf16_t*>(dy), d->ldy,
                               (long long)a.M, d->K, part, rpb);
        else if (quads)
            hipLaunchKernelGGL(colsum4_kernel<float>, g, dim3(256), 0, st, dy, d->ldy, (long long)a.M, d->K, part, rpb);
        else if (bf16 == 2)
            hipLaunchKernelGGL(colsum_kernel<bf16_t>, g, dim3(256), 0, st, reinterpret_cast<const bf16_t*>(dy), d->ldy,
                               (long long)a.M, d->K, part, rpb);
        else
            hipLaunchKernelGGL(colsum_kernel<float>, g, dim3(256), 0, st, dy, d->ldy, (long long)a.M, d->K, part, rpb);
        hipLaunchKernelGGL(colsum_finish_kernel, dim3(cdiv(d->K, 4)), dim3(256), 0, st, (const float*)part, (int)g.x, d->K, dbias,
                           accumulate);
    }
    return check_launch("conv2d_bwd_weight");
}

// ---- profiling control (see ProfScope above) ----------------------------------------------------
extern "C" int up_profile_variants(void) { return PROF_VARIANTS; }
extern "C" const char* up_profile_variant_name(int i) { return (i >= 0 && i < PROF_VARIANTS) ? kVariantNames[i] : ""; }
extern "C" int up_profile_begin(void) {
#ifndef UP_EMU
    g_prof.clear();
    while (g_prof_pool.size() < 2048) {   // (a training step of the image model brackets ~700 launches; created outside the timed region)
        hipEvent_t e;
        if (hipEventCreate(&e) != hipSuccess) break;
        g_prof_pool.push_back(e);
    }
    g_prof_on = true;
#endif
    return UP_OK;
}
extern "C" int up_profile_enable(int on) {
#ifndef UP_EMU
    g_prof_on = on != 0;   // records collected so far are kept: lets a caller sample some steps of a timed region
#else
    (void)on;
#endif
    return UP_OK;
}
extern "C" int up_profile_end(double* out /* [variants][3] = launches, ms, flops */, int variants) {
    UP_REQUIRE(out && variants == PROF_VARIANTS, UP_ERR_INVALID, "profile_end: expected %d variants", PROF_VARIANTS);
    for (int i = 0; i < variants * 3; ++i) out[i] = 0.0;
#ifndef UP_EMU
    g_prof_on = false;
    for (double& v : g_prof_live) v = 0.0;
    FILE* csv = nullptr;
    if (const char* path = getenv("UP_PROFILE_CSV")) {   // optional per-launch dump for offline analysis: the first
        static int calls = 0;                            // collection goes to `path`, later ones to `path.1`, `path.2` ...
        char name[1024];
        if (calls == 0) snprintf(name, sizeof(name), "%s", path);
        else snprintf(name, sizeof(name), "%s.%d", path, calls);
        ++calls;
        csv = fopen(name, "w");
        if (csv) fprintf(csv, "kernel,M,N,K,workgroups,ms,tflops\n");
    }
    for (auto& r : g_prof) {
        float ms = 0.f;
        (void)hipEventSynchronize(r.b);
        (void)hipEventElapsedTime(&ms, r.a, r.b);
        out[r.variant * 3 + 0] += 1.0;
        out[r.variant * 3 + 1] += ms;
        out[r.variant * 3 + 2] += r.flops;
        g_prof_live[r.variant] += r.flops_live;
        if (csv)
            fprintf(csv, "\"%s\",%d,%d,%d,%d,%.5f,%.2f\n", kVariantNames[r.variant], r.M, r.N, r.K, r.grid, ms,
                    r.flops / ms / 1e9);
        g_prof_pool.push_back(r.a);
        g_prof_pool.push_back(r.b);
    }
    g_prof.clear();
    if (csv) fclose(csv);
#endif
    return UP_OK;
}
// FLOP of the collection up_profile_end just closed with every DILATED forward / data-gradient launch charged for its live
// (pixel, filter tap) pairs only — the taps that fall into the padding for a pixel are never multiplied (tile-level skipping,
// tap-sorted rows).  The nominal figures of up_profile_end stay the ones SURVEY 8d's per-image FLOP count is made of.
extern "C" int up_profile_live_flops(double* out /* [variants] */, int variants) {
    UP_REQUIRE(out && variants == PROF_VARIANTS, UP_ERR_INVALID, "profile_live_flops: expected %d variants", PROF_VARIANTS);
    for (int i = 0; i < variants; ++i) {
#ifndef UP_EMU
        out[i] = g_prof_live[i];
#else
        out[i] = 0.0;
#endif
    }
    return UP_OK;
}
