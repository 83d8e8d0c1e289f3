// BatchNorm finalize folded into the launch that produces the per-tile partials (round 6).
//
// Before: conv (Welford partials per row tile) -> bn_finalize_kernel -> bn_apply, and in the backward
// dgrad (sum partials per row tile) / bn_bwd_reduce -> bn_bwd_finalize_kernel -> bn_bwd_apply: 226 launches of 6-8 us (17 us beside
// the weight-gradient stream) on the critical path of every step, each with a dependent kernel boundary on either side.
// Now the producing launch finishes the job itself: every workgroup publishes its partial row with agent-scope (sc1, write-through)
// stores and takes a TICKET; the last arriver of a group of FOLD_G row tiles sums that group in a fixed order into a double-
// precision level-1 row, takes a second ticket, and the last arriver of the whole channel column sums the level-1 rows (fixed
// order again) and writes the per-channel results (mean / invstd / scale / shift / running statistics, or dgamma / dbeta).
// Which workgroup arrives last varies from run to run; WHAT it computes does not: the summation tree is a function of
// (tiles, channel) only, so results are deterministic, and the stand-alone kernel bn_fold_arrive_kernel (one workgroup per partial
// row doing nothing but arriving) produces the same bits — it is the fallback for producers that cannot carry the ticket and the
// A/B partner of the tests.
//
// Statistics are merged as double-precision power sums about zero: a partial (n, mean, M2) contributes (n, n * mean,
// M2 + n * mean^2); mean = S1 / N, M2 = S2 - S1 * mean.  With 53-bit sums the cancellation in M2 costs |mean|^2 / var * 2^-53
// relative (1e-11 at |mean| / std = 364, the worst channel of this network), i.e. the merge is exact to fp32; the Welford chain of
// rounds 1-5 rounded to fp32 after every pairwise merge.
//
// Visibility protocol (MI355X_MICROARCH.md, "inter-workgroup visibility"): payload stored sc1 -> asm s_waitcnt vmcnt(0) -> barrier
// -> relaxed agent-scope ticket; the last arriver reads the payload with sc1 loads (they bypass its L1, and sc1 stores have left
// the producer's L2), so no fence is needed on either side.  Tickets return to zero inside the launch.
#pragma once

namespace up {

constexpr int FOLD_G = 32;      // row tiles per level-1 group
constexpr int FOLD_COLS = 64;   // channels per ticket column and per pass of the merge (64 channel lanes x 4 row lanes)

struct BnFold {
    int* tickets;        // [columns][groups + 1], zero between launches; nullptr: no fold
    double* part2;       // level-1 rows [groups][C][nv]
    int tiles, groups, C;
    // forward (nv = 3: partial rows are (count, mean, M2) per channel)
    float eps, mom;
    float* rm;           // running statistics (optional pair)
    float* rv;
    const float* gamma;
    const float* beta;
    float* mean;         // outputs [C]
    float* invstd;
    float* scale;
    float* shift;
    // backward (nv = 2: partial rows are (sum g, invstd * sum g (y - mean)) per channel)
    float* dgamma;
    float* dbeta;
    // row groups (the frames of the video model's batched trunk, ops.bn_groups): partial rows are [fgroups][tiles][C][nv], every
    // group is merged on its own, ONE last arriver per channel column finishes all of them in group order — forward: the outputs
    // of group g at + g * ostride floats (coef[g][4][C]: ostride = 4 C), the running statistics take the groups' momentum updates
    // in order; backward: gsum[g][dgamma | dbeta][C] (ostride = 2 C: what group g's data gradient needs) and dgamma / dbeta = the
    // sums over the groups.  fgroups <= 1: one group (gsum unused).
    int fgroups, ostride;
    float* gsum;
};

#ifdef UP_EMU
__device__ __forceinline__ void st_agent(float* p, float v) {
    uint32_t u;
    memcpy(&u, &v, 4);
    __atomic_store_n(reinterpret_cast<uint32_t*>(p), u, __ATOMIC_RELEASE);
}
__device__ __forceinline__ float ld_agent(const float* p) {
    uint32_t u = __atomic_load_n(reinterpret_cast<const uint32_t*>(p), __ATOMIC_ACQUIRE);
    float v;
    memcpy(&v, &u, 4);
    return v;
}
__device__ __forceinline__ void st_agent_f64(double* p, double v) {
    uint64_t u;
    memcpy(&u, &v, 8);
    __atomic_store_n(reinterpret_cast<uint64_t*>(p), u, __ATOMIC_RELEASE);
}
__device__ __forceinline__ double ld_agent_f64(const double* p) {
    uint64_t u = __atomic_load_n(reinterpret_cast<const uint64_t*>(p), __ATOMIC_ACQUIRE);
    double v;
    memcpy(&v, &u, 8);
    return v;
}
__device__ __forceinline__ void st_agent_flag(int* p, int v) { __atomic_store_n(p, v, __ATOMIC_RELEASE); }
__device__ __forceinline__ int ticket_take(int* p) { return __atomic_fetch_add(p, 1, __ATOMIC_ACQ_REL); }
__device__ __forceinline__ void drain_stores() {}
#else
__device__ __forceinline__ void st_agent(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ float ld_agent(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_agent_f64(double* p, double v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ double ld_agent_f64(const double* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_agent_flag(int* p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ int ticket_take(int* p) { return __hip_atomic_fetch_add(p, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// every store of this wave acknowledged by the memory side (inline assembly: the compiler may drop a builtin wait it believes
// redundant, MI355X_MICROARCH.md "compiler hazard")
__device__ __forceinline__ void drain_stores() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
#endif

// Called by ALL threads of a workgroup (256, or 512: igemm_big_kernel — the waves beyond the fourth only keep the barriers), uniformly,
// after it has issued the sc1 stores of partial row `tile` for the channels [c0, c0 + ncols) (c0 a multiple of FOLD_COLS).
// `lds`: >= FOLD_LDS_BYTES of workgroup memory nobody else uses any more.
constexpr int FOLD_LDS_BYTES = 4 * FOLD_COLS * 3 * 8 + 16;
template <int NV>
__device__ __forceinline__ void bn_fold_arrive(const BnFold& f, const float* partial, int tile, int c0, int ncols, unsigned char* lds,
                                               int fg = 0) {
    static_assert(NV == 2 || NV == 3, "backward sums or forward statistics");
    const int FG = f.fgroups > 1 ? f.fgroups : 1;
    partial += (size_t)fg * f.tiles * f.C * NV;      // this row group's partial rows
    double* const red = reinterpret_cast<double*>(lds);
    int* const flag = reinterpret_cast<int*>(lds + 4 * FOLD_COLS * 3 * 8);
    const int tid = threadIdx.x;
    const int cl = tid & (FOLD_COLS - 1), r = tid >> 6;   // channel lane, row lane (0..3)
    drain_stores();
    __syncthreads();      // the whole workgroup's partial row is out (and `lds` is free)
    const int g = tile / FOLD_G;
    int* const tk = f.tickets + (size_t)(c0 / FOLD_COLS) * (FG * f.groups + 1);
    const int row2 = fg * f.groups + g;              // this group's level-1 row (and ticket)
    if (tid == 0) {
        const int expect = f.tiles - g * FOLD_G < FOLD_G ? f.tiles - g * FOLD_G : FOLD_G;
        const int last = ticket_take(tk + row2) == expect - 1;
        if (last) st_agent_flag(tk + row2, 0);
        flag[0] = last;
    }
    __syncthreads();
    if (!flag[0]) return;
    // ---- level 1: the rows of group g, row lane r takes tiles t0 + r, t0 + r + 4, ... in order; lanes summed 0..3 ----
    const int t0 = g * FOLD_G, t1 = t0 + FOLD_G < f.tiles ? t0 + FOLD_G : f.tiles;
    for (int cb = 0; cb < ncols; cb += FOLD_COLS) {
        const int c = c0 + cb + cl;
        const bool cok = cb + cl < ncols && c < f.C && r < 4;
        double s[NV];
#pragma unroll
        for (int k = 0; k < NV; ++k) s[k] = 0.0;
        if (cok) {
            float v[FOLD_G / 4][NV];
#pragma unroll
            for (int u = 0; u < FOLD_G / 4; ++u) {
                const int t = t0 + r + 4 * u;
                const float* p = partial + ((size_t)(t < t1 ? t : t0) * f.C + c) * NV;
#pragma unroll
                for (int k = 0; k < NV; ++k) v[u][k] = ld_agent(p + k);
            }
#pragma unroll
            for (int u = 0; u < FOLD_G / 4; ++u) {
                if (t0 + r + 4 * u >= t1) continue;
                if constexpr (NV == 3) {
                    const double n = (double)v[u][0], m = (double)v[u][1];
                    const double nm = n * m;
                    s[0] += n;
                    s[1] += nm;
                    s[2] += (double)v[u][2] + nm * m;
                } else {
                    s[0] += (double)v[u][0];
                    s[1] += (double)v[u][1];
                }
            }
        }
        if (r < 4) {
#pragma unroll
            for (int k = 0; k < NV; ++k) red[(r * FOLD_COLS + cl) * NV + k] = s[k];
        }
        __syncthreads();
        if (r == 0 && cok) {
#pragma unroll
            for (int k = 0; k < NV; ++k) {
                double t = red[cl * NV + k];
                for (int rr = 1; rr < 4; ++rr) t += red[(rr * FOLD_COLS + cl) * NV + k];
                st_agent_f64(f.part2 + ((size_t)row2 * f.C + c) * NV + k, t);
            }
        }
        __syncthreads();
    }
    drain_stores();
    __syncthreads();
    if (tid == 0) {
        const int last = ticket_take(tk + FG * f.groups) == FG * f.groups - 1;
        if (last) st_agent_flag(tk + FG * f.groups, 0);
        flag[0] = last;
    }
    __syncthreads();
    if (!flag[0]) return;
    // ---- level 2: per row group, row lane r takes level-1 rows r, r + 4, ... in order; lanes summed 0..3; one thread per channel
    //      finishes the groups in order ----
    for (int cb = 0; cb < ncols; cb += FOLD_COLS) {
        const int c = c0 + cb + cl;
        const bool cok = cb + cl < ncols && c < f.C && r < 4;
        // the finishing thread's per-channel operands ride with the level-1 rows (one round trip less on the launch's tail)
        float pg = 0.f, pb = 0.f, prm = 0.f, prv = 0.f;
        if constexpr (NV == 3) {
            if (r == 0 && cok) {
                pg = f.gamma[c];
                pb = f.beta[c];
                if (f.rm) {
                    prm = f.rm[c];
                    prv = f.rv[c];
                }
            }
        }
        double tot0 = 0.0, tot1 = 0.0;      // backward: sums over the row groups
        for (int fgi = 0; fgi < FG; ++fgi) {
            const double* const rows = f.part2 + (size_t)fgi * f.groups * f.C * NV;
            double s[NV];
#pragma unroll
            for (int k = 0; k < NV; ++k) s[k] = 0.0;
            if (cok) {
                for (int gg = r; gg < f.groups; gg += 16) {   // four rows in flight
                    double v[4][NV];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int g2 = gg + 4 * u;
                        const double* p = rows + ((size_t)(g2 < f.groups ? g2 : gg) * f.C + c) * NV;
#pragma unroll
                        for (int k = 0; k < NV; ++k) v[u][k] = ld_agent_f64(p + k);
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u)
                        if (gg + 4 * u < f.groups) {
#pragma unroll
                            for (int k = 0; k < NV; ++k) s[k] += v[u][k];
                        }
                }
            }
            if (r < 4) {
#pragma unroll
                for (int k = 0; k < NV; ++k) red[(r * FOLD_COLS + cl) * NV + k] = s[k];
            }
            __syncthreads();
            if (r == 0 && cok) {
                double t[NV];
#pragma unroll
                for (int k = 0; k < NV; ++k) {
                    t[k] = red[cl * NV + k];
                    for (int rr = 1; rr < 4; ++rr) t[k] += red[(rr * FOLD_COLS + cl) * NV + k];
                }
                const size_t o = (size_t)fgi * f.ostride + c;
                if constexpr (NV == 3) {
                    const double N = t[0];
                    const double md = N > 0.0 ? t[1] / N : 0.0;
                    double m2 = t[2] - t[1] * md;
                    if (m2 < 0.0) m2 = 0.0;
                    const float m = (float)md;
                    const float var = N > 0.0 ? (float)(m2 / N) : 0.f;
                    const float is = 1.0f / sqrtf(var + f.eps);
                    f.mean[o] = m;
                    f.invstd[o] = is;
                    const float sc = pg * is;
                    f.scale[o] = sc;
                    f.shift[o] = pb - m * sc;
                    const float unb = N > 1.0 ? (float)(m2 / (N - 1.0)) : var;
                    prm = (1.f - f.mom) * prm + f.mom * m;      // the groups' momentum updates, in order
                    prv = (1.f - f.mom) * prv + f.mom * unb;
                } else {
                    if (FG > 1) {
                        f.gsum[o] = (float)t[1];                  // dgamma of the group (sum g * xhat)
                        f.gsum[o + f.C] = (float)t[0];            // dbeta of the group (sum g)
                    }
                    tot0 += t[0];
                    tot1 += t[1];
                }
            }
            __syncthreads();
        }
        if (r == 0 && cok) {
            if constexpr (NV == 3) {
                if (f.rm) {
                    f.rm[c] = prm;
                    f.rv[c] = prv;
                }
            } else {
                f.dbeta[c] = (float)tot0;
                f.dgamma[c] = (float)tot1;
            }
        }
    }
}

// scratch of the fold on one stream (conv_igemm.hip owns the per-stream device memory): tickets + level-1 rows for a merge of
// `tiles` partial rows (per row group) over C channels with nv values each; false when it does not fit (the caller runs the
// stand-alone form, which fails loudly) or the fold is switched off
bool bn_fold_scratch(hipStream_t st, int tiles, int C, int nv, BnFold* f, int fgroups = 1);
bool bn_fold_enabled();

}  // namespace up
