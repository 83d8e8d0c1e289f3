// BatchNorm (train statistics merge, apply, backward), ReLU backward, dropout, MSE.
// All of these are HBM-bound streaming kernels: 16-byte accesses along the channel axis of the
// NHWC tensors, grid-stride loops capped at a few blocks per CU, wave-level shuffles + LDS for
// the per-channel reductions.
#include "up_common.h"
#include "bn_fold.h"

#include <stdarg.h>
#include <stdlib.h>

namespace up {

static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("%s: %s", what, hipGetErrorString(e));
        return UP_ERR_LAUNCH;
    }
    return UP_OK;
}

// grid-stride elementwise kernels: workgroup cap from A/B runs of the whole step (2048 / 4096 / 8192 / 16384 workgroups:
// 70.3 / 69.9 / 69.6 / 69.6 ms) — beside the MFMA kernels of the other stream, many short workgroups fill the gaps better
static inline int grid_for(int64_t work_items, int per_block = 256, int cap = 16384) {
    int64_t g = (work_items + per_block - 1) / per_block;
    if (g < 1) g = 1;
    return (int)(g > cap ? cap : g);
}

__device__ __forceinline__ void wf_merge3(float& n1, float& m1, float& s1, float n2, float m2, float s2) {
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

__global__ void __launch_bounds__(256) bn_eval_coeffs_kernel(const float* g, const float* b, const float* rm,
                                                             const float* rv, float eps, int C, float* scale,
                                                             float* shift) {
    int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    float is = 1.0f / sqrtf(rv[c] + eps);
    float sc = g[c] * is;
    scale[c] = sc;
    shift[c] = b[c] - rm[c] * sc;
}

// Row groups (grouped BatchNorm, see up_bn_*_groups): blockIdx.z / blockIdx.y = group; `rows` of a kernel is then the row count
// of ONE group, the group's rows start at g * rows, its per-channel parameters at g * pstride floats (coef[g][4][C]), its
// backward sums at g * gstride floats.  All zero for an ordinary launch.
struct GroupArgs {
    int pstride, gstride;
};

// The stand-alone form of the BatchNorm fold (bn_fold.h): one workgroup per (partial row, 64-channel column) that does nothing but
// ARRIVE — the merge tree, and therefore every result bit, is the one a producing launch that carries the ticket itself computes.
// Used when the producer could not fold (row groups excluded: they keep bn_finalize_kernel's per-group forms), and by the tests.
template <int NV>
__global__ void __launch_bounds__(256) bn_fold_arrive_kernel(const float* partial, BnFold f) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[FOLD_LDS_BYTES];
    const int c0 = blockIdx.y * FOLD_COLS;
    bn_fold_arrive<NV>(f, partial, blockIdx.x, c0, f.C - c0 < FOLD_COLS ? f.C - c0 : FOLD_COLS, lds, blockIdx.z);
}

// One workgroup per channel: thread t merges the partials of row tiles t, t+256, ... (one or two independent loads for
// the layer shapes of this network: 265 tiles at 23x23 / B = 32), then an 8-level merge tree over LDS.  These few-hundred-
// byte kernels sit on the critical path of the forward pass between every convolution and its apply pass, and their
// time is the number of DEPENDENT load rounds (the partials were written by other XCDs: ~2 us per round): one wavefront
// per channel (5 rounds) took 10.5 us per layer, 16 lanes per channel with coalesced 192-byte rows (17 rounds) 23 us.
__global__ void __launch_bounds__(256) bn_finalize_kernel(const float* stats, int tiles, int C, float eps, float mom,
                                                          float* rm, float* rv, const float* gamma,
                                                          const float* beta, float* mean_o, float* invstd_o,
                                                          float* scale, float* shift, GroupArgs grp) {
    __shared__ float red[256][3];
    const int c = blockIdx.x, t0 = threadIdx.x;
    {
        const int g = blockIdx.y;
        stats += (size_t)g * tiles * C * 3;
        mean_o += (size_t)g * grp.pstride;
        invstd_o += (size_t)g * grp.pstride;
        scale += (size_t)g * grp.pstride;
        shift += (size_t)g * grp.pstride;
    }
    float n = 0.f, m = 0.f, q = 0.f;
    for (int t = t0; t < tiles; t += 512) {      // two loads in flight per round
        const float* s = stats + ((size_t)t * C + c) * 3;
        const int t2 = t + 256;
        const float* s2 = stats + ((size_t)(t2 < tiles ? t2 : t) * C + c) * 3;
        const float a0 = s[0], a1 = s[1], a2 = s[2];
        const float b0 = t2 < tiles ? s2[0] : 0.f, b1 = s2[1], b2 = s2[2];
        wf_merge3(n, m, q, a0, a1, a2);
        wf_merge3(n, m, q, b0, b1, b2);
    }
    red[t0][0] = n;
    red[t0][1] = m;
    red[t0][2] = q;
    __syncthreads();
#pragma unroll
    for (int off = 128; off >= 1; off >>= 1) {
        if (t0 < off) {
            wf_merge3(n, m, q, red[t0 + off][0], red[t0 + off][1], red[t0 + off][2]);
            red[t0][0] = n;
            red[t0][1] = m;
            red[t0][2] = q;
        }
        __syncthreads();
    }
    if (t0 == 0) {
        float var = q / n;
        float is = 1.0f / sqrtf(var + eps);
        mean_o[c] = m;
        invstd_o[c] = is;
        float sc = gamma[c] * is;
        scale[c] = sc;
        shift[c] = beta[c] - m * sc;
        if (rm) {
            float unb = n > 1.f ? q / (n - 1.f) : var;
            rm[c] = (1.f - mom) * rm[c] + mom * m;
            rv[c] = (1.f - mom) * rv[c] + mom * unb;
        }
    }
}

// relu_bits (optional): bit (row*C + c) of a dense bit array = (z > 0).  The backward passes read this instead of z:
// 1/32 of the bytes.  One thread produces 4 bits; 8 neighbouring lanes are merged into one 32-bit word.
template <typename T>
__global__ void __launch_bounds__(256) bn_apply_kernel(const T* y, int ldy, const float* scale,
                                                       const float* shift, const float* mean, const T* res, int ldr, int relu,
                                                       T* z, int ldz, uint32_t* relu_bits, int64_t total, int C4,
                                                       FastDiv fC4) {
    const int lane = threadIdx.x & 63;
    for (int64_t i0 = (int64_t)blockIdx.x * 256 + (threadIdx.x & ~63); i0 < total; i0 += (int64_t)gridDim.x * 256) {
        const int64_t i = i0 + lane;
        uint32_t nib = 0;
        if (i < total) {
            uint32_t row = fdiv((uint32_t)i, fC4);
            int c = ((int)i - (int)row * C4) * 4;
            float4 v = ld4<T>(y + (size_t)row * ldy + c);
            float4 s = *reinterpret_cast<const float4*>(scale + c);
            float4 h = *reinterpret_cast<const float4*>(shift + c);
            const float4 mu = mean ? *reinterpret_cast<const float4*>(mean + c) : make_float4(0.f, 0.f, 0.f, 0.f);
            v.x = (v.x - mu.x) * s.x + h.x;      // centred form, see up_bn_apply_centered_t (mu = 0: y * scale + shift exactly)
            v.y = (v.y - mu.y) * s.y + h.y;
            v.z = (v.z - mu.z) * s.z + h.z;
            v.w = (v.w - mu.w) * s.w + h.w;
            if (res) {
                float4 r = ld4<T>(res + (size_t)row * ldr + c);
                v.x += r.x;
                v.y += r.y;
                v.z += r.z;
                v.w += r.w;
            }
            if (relu) {
                v.x = fmaxf(v.x, 0.f);
                v.y = fmaxf(v.y, 0.f);
                v.z = fmaxf(v.z, 0.f);
                v.w = fmaxf(v.w, 0.f);
            }
            st4(z + (size_t)row * ldz + c, v);
            nib = (v.x > 0.f ? 1u : 0u) | (v.y > 0.f ? 2u : 0u) | (v.z > 0.f ? 4u : 0u) | (v.w > 0.f ? 8u : 0u);
        }
        if (relu_bits) {   // wave-uniform
            uint32_t w = nib << (4 * (lane & 7));
            w |= __shfl_xor(w, 1);
            w |= __shfl_xor(w, 2);
            w |= __shfl_xor(w, 4);
            if ((lane & 7) == 0 && i < total) relu_bits[i >> 3] = w;
        }
    }
}

// bf16 storage, C % 8 == 0: EIGHT channels per thread, i.e. the same 16 bytes per access as the fp32 kernel (with four, the
// bf16 passes moved the bytes of the fp32 ones in 1.2-1.4x the time: half as many bytes in flight per thread).
struct F8 {
    float v[8];
};
__device__ __forceinline__ F8 ld8(const bf16_t* p) {
    const uint4 u = *reinterpret_cast<const uint4*>(p);
    F8 r;
    r.v[0] = bf_lo(u.x); r.v[1] = bf_hi(u.x); r.v[2] = bf_lo(u.y); r.v[3] = bf_hi(u.y);
    r.v[4] = bf_lo(u.z); r.v[5] = bf_hi(u.z); r.v[6] = bf_lo(u.w); r.v[7] = bf_hi(u.w);
    return r;
}
__device__ __forceinline__ F8 ld8(const float* p) {
    const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
    F8 r;
    r.v[0] = a.x; r.v[1] = a.y; r.v[2] = a.z; r.v[3] = a.w; r.v[4] = b.x; r.v[5] = b.y; r.v[6] = b.z; r.v[7] = b.w;
    return r;
}
__device__ __forceinline__ void st8(bf16_t* p, const F8& a) {
    uint4 u;
    u.x = pack_bf16x2(a.v[0], a.v[1]);
    u.y = pack_bf16x2(a.v[2], a.v[3]);
    u.z = pack_bf16x2(a.v[4], a.v[5]);
    u.w = pack_bf16x2(a.v[6], a.v[7]);
    *reinterpret_cast<uint4*>(p) = u;
}

__global__ void __launch_bounds__(256) bn_apply8_kernel(const bf16_t* y, int ldy, const float* scale, const float* shift,
                                                        const float* mean, const bf16_t* res, int ldr, int relu, bf16_t* z, int ldz,
                                                        uint32_t* relu_bits, int64_t total, int C8, FastDiv fC8) {
    const int lane = threadIdx.x & 63;
    for (int64_t i0 = (int64_t)blockIdx.x * 256 + (threadIdx.x & ~63); i0 < total; i0 += (int64_t)gridDim.x * 256) {
        const int64_t i = i0 + lane;
        uint32_t byte = 0;
        if (i < total) {
            const uint32_t row = fdiv((uint32_t)i, fC8);
            const int c = ((int)i - (int)row * C8) * 8;
            F8 v = ld8(y + (size_t)row * ldy + c);
            const F8 s = ld8(scale + c), h = ld8(shift + c);
            F8 mu;
#pragma unroll
            for (int e = 0; e < 8; ++e) mu.v[e] = 0.f;
            if (mean) mu = ld8(mean + c);
#pragma unroll
            for (int e = 0; e < 8; ++e) v.v[e] = (v.v[e] - mu.v[e]) * s.v[e] + h.v[e];
            if (res) {
                const F8 r = ld8(res + (size_t)row * ldr + c);
#pragma unroll
                for (int e = 0; e < 8; ++e) v.v[e] += r.v[e];
            }
            if (relu) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v.v[e] = fmaxf(v.v[e], 0.f);
            }
            st8(z + (size_t)row * ldz + c, v);
#pragma unroll
            for (int e = 0; e < 8; ++e) byte |= v.v[e] > 0.f ? (1u << e) : 0u;
        }
        if (relu_bits) {   // wave-uniform; bit (row*C + c) as in bn_apply_kernel: one byte per thread, four lanes per word
            uint32_t w = byte << (8 * (lane & 3));
            w |= __shfl_xor(w, 1);
            w |= __shfl_xor(w, 2);
            if ((lane & 3) == 0 && i < total) relu_bits[i >> 2] = w;
        }
    }
}

// ---- row-strided forms of the two element-wise BatchNorm passes -----------------------------------------------------------
// The kernels above walk a flat (row, channel group) index: every 16-byte group of y re-loads its per-channel parameters
// (2 x 16-32 bytes forward, 5 x 16-32 bytes backward against 16-32 bytes of data) — they are bound by the number of load
// instructions, not by HBM (3.4 TB/s measured on the backward pass at 736x736, r03_f).  When the number of 16-byte channel
// groups per row is a power of two, a thread can keep ONE channel group for its whole life and stride over the rows: the
// parameters are loaded once, the loop body is data loads and stores only, unrolled so four rows are in flight.
// Same arithmetic, same bit layout of relu_bits (bit row * C + c), results bitwise equal to the flat kernels.
template <typename T>
struct Row16 {
    static constexpr int E = 16 / (int)sizeof(T);   // channels per 16-byte access: 4 (fp32) or 8 (bf16)
    float v[E];
};
__device__ __forceinline__ Row16<float> ld16(const float* p) {
    const float4 a = *reinterpret_cast<const float4*>(p);
    return Row16<float>{{a.x, a.y, a.z, a.w}};
}
__device__ __forceinline__ Row16<bf16_t> ld16(const bf16_t* p) {
    const uint4 u = *reinterpret_cast<const uint4*>(p);
    return Row16<bf16_t>{{bf_lo(u.x), bf_hi(u.x), bf_lo(u.y), bf_hi(u.y), bf_lo(u.z), bf_hi(u.z), bf_lo(u.w), bf_hi(u.w)}};
}
__device__ __forceinline__ void st16(float* p, const Row16<float>& a) {
    *reinterpret_cast<float4*>(p) = make_float4(a.v[0], a.v[1], a.v[2], a.v[3]);
}
__device__ __forceinline__ void st16(bf16_t* p, const Row16<bf16_t>& a) {
    uint4 u;
    u.x = pack_bf16x2(a.v[0], a.v[1]);
    u.y = pack_bf16x2(a.v[2], a.v[3]);
    u.z = pack_bf16x2(a.v[4], a.v[5]);
    u.w = pack_bf16x2(a.v[6], a.v[7]);
    *reinterpret_cast<uint4*>(p) = u;
}
template <int E>
__device__ __forceinline__ void ldparam(const float* p, float (&o)[E]) {
#pragma unroll
    for (int e = 0; e < E; e += 4) {
        const float4 a = *reinterpret_cast<const float4*>(p + e);
        o[e] = a.x;
        o[e + 1] = a.y;
        o[e + 2] = a.z;
        o[e + 3] = a.w;
    }
}
constexpr int BN_ROWS_UNROLL = 4;

// lcs = log2(channel lanes per workgroup): lane tid & (LC-1) owns channel group blockIdx.y * LC + lane, the 256 / LC row
// lanes of the workgroup and gridDim.x workgroups stride over the rows
template <typename T>
__global__ void __launch_bounds__(256) bn_apply_rows_kernel(const T* __restrict__ y, int ldy, const float* __restrict__ scale,
                                                            const float* __restrict__ shift, const float* __restrict__ mean,
                                                            const T* __restrict__ res, int ldr,
                                                            int relu, T* __restrict__ z, int ldz, uint32_t* __restrict__ relu_bits,
                                                            int rows, int C, int lcs, GroupArgs grp) {
    constexpr int E = Row16<T>::E, LPW = 32 / E;   // lanes per 32-bit word of relu_bits
    const int lane = threadIdx.x & 63;
    const int64_t grow0 = (int64_t)blockIdx.z * rows;      // first row of this group
    y += grow0 * ldy;
    z += grow0 * ldz;
    if (res) res += grow0 * ldr;
    scale += (size_t)blockIdx.z * grp.pstride;
    // centred form (mean given): `shift` is the BatchNorm bias, one vector for all groups; else the per-group shift row
    shift += mean ? 0 : (size_t)blockIdx.z * grp.pstride;
    const int cl = threadIdx.x & ((1 << lcs) - 1), rl = threadIdx.x >> lcs;
    const int rpb = 256 >> lcs;
    const int c = ((blockIdx.y << lcs) + cl) * E;
    float sc[E], sh[E], mu[E];
    ldparam<E>(scale + c, sc);
    ldparam<E>(shift + c, sh);
#pragma unroll
    for (int e = 0; e < E; ++e) mu[e] = 0.f;
    if (mean) ldparam<E>(mean + (size_t)blockIdx.z * grp.pstride + c, mu);
    const int stride = gridDim.x * rpb;
    for (int base = blockIdx.x * rpb; base < rows; base += BN_ROWS_UNROLL * stride) {   // uniform trip count: the bit merge shuffles
        Row16<T> v[BN_ROWS_UNROLL], r[BN_ROWS_UNROLL];
#pragma unroll
        for (int u = 0; u < BN_ROWS_UNROLL; ++u) {
            const int row = base + u * stride + rl;
            const int rr = row < rows ? row : rows - 1;
            v[u] = ld16(y + (size_t)rr * ldy + c);
            if (res) r[u] = ld16(res + (size_t)rr * ldr + c);
        }
#pragma unroll
        for (int u = 0; u < BN_ROWS_UNROLL; ++u) {
            const int row = base + u * stride + rl;
            const bool ok = row < rows;
            uint32_t bits = 0;
#pragma unroll
            for (int e = 0; e < E; ++e) {
                float t = (v[u].v[e] - mu[e]) * sc[e] + sh[e];
                if (res) t += r[u].v[e];
                if (relu) t = fmaxf(t, 0.f);
                v[u].v[e] = t;
                bits |= t > 0.f ? (1u << e) : 0u;
            }
            if (ok) st16(z + (size_t)row * ldz + c, v[u]);
            if (relu_bits) {   // uniform
                uint32_t w = bits << (E * (lane & (LPW - 1)));
#pragma unroll
                for (int m = 1; m < LPW; m <<= 1) w |= __shfl_xor(w, m);
                if ((lane & (LPW - 1)) == 0 && ok) relu_bits[((grow0 + row) * C + c) >> 5] = w;
            }
        }
    }
}

// Per-channel factors folded once per thread into (k, a, b, mean): dy = k * g + a + (y - mean) * b with k = gamma * invstd,
// a = -k * dbeta / m, b = -k * invstd * dgamma / m.  UNROLL rows in flight per thread; the register budget matters more than
// the depth here: this pass runs beside the weight-gradient kernels of the side stream (two workgroups of ~170 VGPRs per
// SIMD pair), and what is left of the register file decides how many of its waves fit next to them.  ZPATH: ReLU decisions
// taken from the stored output z (no bit mask recorded).
template <typename T, int UNROLL, bool ZPATH>
__global__ void __launch_bounds__(256) bn_bwd_apply_rows_kernel(const T* __restrict__ dz, int lddz, const T* __restrict__ z, int ldz,
                                                                const uint32_t* __restrict__ bits, const T* __restrict__ y, int ldy,
                                                                const float* __restrict__ gamma, const float* __restrict__ mean,
                                                                const float* __restrict__ invstd, const float* __restrict__ dgamma,
                                                                const float* __restrict__ dbeta, int relu, int use_batch, float inv_m,
                                                                T* __restrict__ dy, int lddy, T* __restrict__ dres, int lddres,
                                                                int rows, int C, int lcs, GroupArgs grp) {
    constexpr int E = Row16<T>::E;
    const int64_t grow0 = (int64_t)blockIdx.z * rows;      // first row of this group
    dz += grow0 * lddz;
    y += grow0 * ldy;
    dy += grow0 * lddy;
    if (ZPATH) z += grow0 * ldz;
    if (dres) dres += grow0 * lddres;
    mean += (size_t)blockIdx.z * grp.pstride;
    invstd += (size_t)blockIdx.z * grp.pstride;
    dgamma += (size_t)blockIdx.z * grp.gstride;
    dbeta += (size_t)blockIdx.z * grp.gstride;
    const int cl = threadIdx.x & ((1 << lcs) - 1), rl = threadIdx.x >> lcs;
    const int rpb = 256 >> lcs;
    const int c = ((blockIdx.y << lcs) + cl) * E;
    float kk[E], aa[E], bb[E], mu[E];
    {
        float is[E], t[E];
        ldparam<E>(invstd + c, is);
        ldparam<E>(gamma + c, kk);
        ldparam<E>(mean + c, mu);
#pragma unroll
        for (int e = 0; e < E; ++e) kk[e] *= is[e];
        ldparam<E>(dbeta + c, t);
#pragma unroll
        for (int e = 0; e < E; ++e) aa[e] = use_batch ? -kk[e] * t[e] * inv_m : 0.f;
        ldparam<E>(dgamma + c, t);
#pragma unroll
        for (int e = 0; e < E; ++e) bb[e] = use_batch ? -kk[e] * is[e] * t[e] * inv_m : 0.f;
    }
    const int stride = gridDim.x * rpb;
    for (int base = blockIdx.x * rpb + rl; base < rows; base += UNROLL * stride) {
        Row16<T> g[UNROLL], yv[UNROLL], zz[ZPATH ? UNROLL : 1];
        uint32_t mk[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            const int row = base + u * stride;
            const int rr = row < rows ? row : rows - 1;
            g[u] = ld16(dz + (size_t)rr * lddz + c);
            yv[u] = ld16(y + (size_t)rr * ldy + c);
            mk[u] = 0xffffffffu;
            if (ZPATH) {
                zz[u] = ld16(z + (size_t)rr * ldz + c);
            } else if (relu) {
                const int64_t b = (grow0 + rr) * C + c;
                mk[u] = bits[b >> 5] >> (int)(b & 31);
            }
        }
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            const int row = base + u * stride;
            if (row >= rows) continue;
            Row16<T> o;
#pragma unroll
            for (int e = 0; e < E; ++e) {
                float ge = g[u].v[e];
                const bool pos = ZPATH ? zz[ZPATH ? u : 0].v[e] > 0.f : ((mk[u] >> e) & 1u) != 0u;
                if (!pos) ge = 0.f;
                g[u].v[e] = ge;
                o.v[e] = kk[e] * ge + (aa[e] + (yv[u].v[e] - mu[e]) * bb[e]);
            }
            if (dres) st16(dres + (size_t)row * lddres + c, g[u]);
            st16(dy + (size_t)row * lddy + c, o);
        }
    }
}
template <typename T>
static void launch_bn_bwd_apply_rows(dim3 grid, hipStream_t st, const T* dz, int lddz, const T* z, int ldz, const uint32_t* bits,
                                     const T* y, int ldy, const float* gamma, const float* mean, const float* invstd,
                                     const float* dgamma, const float* dbeta, int relu, int use_batch, float inv_m, T* dy, int lddy,
                                     T* dres, int lddres, int rows, int C, int lcs, GroupArgs grp) {
    // rows in flight per thread: 1 for fp32 (38 VGPRs: four waves per SIMD still fit beside two weight-gradient workgroups;
    // 368^2 B = 32 step 62.3 -> 61.75 ms against 2 or 4, same-session A/B), 2 for bf16 (no difference measured at 736^2)
    constexpr int U = sizeof(T) == 4 ? 1 : 2;
    if (relu && !bits)
        hipLaunchKernelGGL((bn_bwd_apply_rows_kernel<T, U, true>), grid, dim3(256), 0, st, dz, lddz, z, ldz, bits, y, ldy, gamma, mean,
                           invstd, dgamma, dbeta, relu, use_batch, inv_m, dy, lddy, dres, lddres, rows, C, lcs, grp);
    else
        hipLaunchKernelGGL((bn_bwd_apply_rows_kernel<T, U, false>), grid, dim3(256), 0, st, dz, lddz, z, ldz, bits, y, ldy, gamma, mean,
                           invstd, dgamma, dbeta, relu, use_batch, inv_m, dy, lddy, dres, lddres, rows, C, lcs, grp);
}
// launch geometry of the row-strided kernels, or false when the channel-group count is not a power of two (>= 8)
template <typename T>
static bool rows_geometry(int64_t rows, int C, dim3& grid, int& lcs) {
    constexpr int E = 16 / (int)sizeof(T);
    if (C % E) return false;
    const int cgs = C / E;
    if (cgs < 8 || (cgs & (cgs - 1)) || rows >= (1ll << 31)) return false;
    const int lc = cgs < 256 ? cgs : 256;
    lcs = 0;
    while ((1 << lcs) < lc) ++lcs;
    const int rpb = 256 / lc, gy = cgs / lc;
    // ~8 workgroups per CU in total, every thread walks >= BN_ROWS_UNROLL rows when there are enough of them
    int64_t gx = (rows + (int64_t)rpb * BN_ROWS_UNROLL - 1) / ((int64_t)rpb * BN_ROWS_UNROLL);
    const int64_t cap = 2048 / gy;
    if (gx > cap) gx = cap;
    if (gx < 1) gx = 1;
    grid = dim3((unsigned)gx, (unsigned)gy);
    return true;
}
static int g_bn_rows = -1;   // UP_BN_ROWS=0 / up_conv_tune("bn_rows", 0) keeps the flat kernels (A/B)
void set_bn_rows(int on) { g_bn_rows = on ? 1 : 0; }
static bool bn_rows_enabled() {
    if (g_bn_rows < 0) {
        const char* e = getenv("UP_BN_ROWS");
        g_bn_rows = (e && atoi(e) == 0) ? 0 : 1;
    }
    return g_bn_rows != 0;
}

// ---- batch statistics of row GROUPS (video model: the trunk runs ONCE on all T frames of a clip batch, but the reference
// normalises every frame's batch on its own, uniposeLSTM.py:116-133 calls the trunk per frame) ------------------------------
// The convolution epilogue's per-tile partials cannot serve here: a row tile may straddle two frames.  One extra read of y
// instead: partial (count, mean, M2) per (group, 256-row chunk, channel), the format bn_finalize_kernel merges.  A thread
// accumulates shifted sums (shift = its first sample, so there is no cancellation) over its 16 rows, the 16 row lanes are
// merged in a fixed order.
constexpr int BNS_ROWS = 256;
template <typename T>
__global__ void __launch_bounds__(256) bn_batch_stats_kernel(const T* y, int ldy, int rows_per_group, int C, float* stats, int tiles,
                                                             BnFold fold) {
    __shared__ __attribute__((aligned(16))) float red[16][64][3];
    static_assert(sizeof(float) * 16 * 64 * 3 >= FOLD_LDS_BYTES, "the fold's ticket re-uses the reduction buffer");
    const int cq = threadIdx.x & 15, rl = threadIdx.x >> 4;
    const int c = blockIdx.y * 64 + cq * 4;
    const int g = blockIdx.z;
    const int r0 = blockIdx.x * BNS_ROWS, r1 = min(rows_per_group, r0 + BNS_ROWS);
    float n = 0.f, sh[4] = {0.f, 0.f, 0.f, 0.f}, s[4] = {0.f, 0.f, 0.f, 0.f}, q[4] = {0.f, 0.f, 0.f, 0.f};
    if (c < C) {
        const T* base = y + (size_t)g * rows_per_group * ldy + c;
        // shift = the chunk's first row (the same line for all 16 row lanes), so the row loop has no first-iteration branch and
        // four rows are in flight per thread (the loop was one dependent load per iteration: 17 us for a 4 us read, r04_z)
        const float4 s0 = ld4<T>(base + (size_t)r0 * ldy);
        sh[0] = s0.x, sh[1] = s0.y, sh[2] = s0.z, sh[3] = s0.w;
        auto acc = [&](const float4& v) {
            const float x[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float d = x[e] - sh[e];
                s[e] += d;
                q[e] += d * d;
            }
            n += 1.f;
        };
        int r = r0 + rl;
        for (; r + 48 < r1; r += 64) {
            float4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = ld4<T>(base + (size_t)(r + 16 * u) * ldy);
#pragma unroll
            for (int u = 0; u < 4; ++u) acc(v[u]);
        }
        for (; r < r1; r += 16) acc(ld4<T>(base + (size_t)r * ldy));
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float mean = n > 0.f ? sh[e] + s[e] / n : 0.f;
        const float m2 = n > 0.f ? fmaxf(q[e] - s[e] * s[e] / n, 0.f) : 0.f;
        red[rl][cq * 4 + e][0] = n;
        red[rl][cq * 4 + e][1] = mean;
        red[rl][cq * 4 + e][2] = m2;
    }
    __syncthreads();
    if (threadIdx.x < 64) {
        const int ch = threadIdx.x;
        float cn = 0.f, cm = 0.f, cs = 0.f;
        for (int k = 0; k < 16; ++k) wf_merge3(cn, cm, cs, red[k][ch][0], red[k][ch][1], red[k][ch][2]);
        const int cc = blockIdx.y * 64 + ch;
        if (cc < C) {
            float* o = stats + (((size_t)g * tiles + blockIdx.x) * C + cc) * 3;   // (sc1: the fold's last arriver reads them)
            st_agent(o, cn);
            st_agent(o + 1, cm);
            st_agent(o + 2, cs);
        }
    }
    if (fold.tickets) {   // uniform: this launch also finalizes every group (bn_fold.h, row groups)
        const int c0 = blockIdx.y * 64;
        bn_fold_arrive<3>(fold, stats, blockIdx.x, c0, C - c0 < 64 ? C - c0 : 64, reinterpret_cast<unsigned char*>(&red[0][0][0]), g);
    }
}

// Statistics of a SMALL batch (a few rows per channel: the BatchNorm behind the global-average-pool branch sees B values), one
// thread per channel, float64 two-pass: the mean is the correctly rounded mean of the stored values, like ATen's CPU kernel (double
// accumulators).  With few samples nothing averages the round-off of a float sum out, and with |mean| >> std (364 for that layer on
// the G14 input) an ulp of the mean is 2e-5 of the normalised scale.
template <typename T>
__global__ void __launch_bounds__(256) bn_exact_stats_kernel(const T* y, int ldy, int rows_per_group, int C, float* stats) {
    const int c = blockIdx.x * 256 + threadIdx.x, g = blockIdx.y;
    if (c >= C) return;
    const T* base = y + (size_t)g * rows_per_group * ldy + c;
    double sum = 0.0;
    for (int r = 0; r < rows_per_group; ++r) sum += (double)ld1(base + (size_t)r * ldy);
    const double mean = sum / rows_per_group;
    double m2 = 0.0;
    for (int r = 0; r < rows_per_group; ++r) {
        const double d = (double)ld1(base + (size_t)r * ldy) - mean;
        m2 += d * d;
    }
    float* o = stats + ((size_t)g * C + c) * 3;
    o[0] = (float)rows_per_group;
    o[1] = (float)mean;
    o[2] = (float)m2;
}

// ---- BN backward -------------------------------------------------------------------------
// pass 1: partial[chunk][c] = {sum g, sum g*xhat},   g = dz * (z > 0 if relu)
// 256 threads = 16 row lanes x 16 channel quads (64 channels): every access is a 16-byte load of 4
// consecutive channels, rows strided by 16 and unrolled x2 so 6 loads are in flight per thread.
// the ReLU mask of 4 consecutive channels: from the bit array written by bn_apply_kernel when there is one
// (relu_bits: 1/32 of the bytes of z), else from z itself
template <typename T>
__device__ __forceinline__ float4 relu_mask4(const uint32_t* bits, int64_t quad, const T* z, size_t zoff) {
    if (bits) {
        const uint32_t nib = bits[quad >> 3] >> (4 * (int)(quad & 7));
        return make_float4((nib & 1u) ? 1.f : 0.f, (nib & 2u) ? 1.f : 0.f, (nib & 4u) ? 1.f : 0.f, (nib & 8u) ? 1.f : 0.f);
    }
    return ld4<T>(z + zoff);
}

template <typename T>
__global__ void __launch_bounds__(256) bn_bwd_reduce_kernel(const T* dz, int lddz, const T* z, int ldz,
                                                            const uint32_t* bits, const T* y, int ldy,
                                                            const float* mean, const float* invstd, int relu,
                                                            float* partial, int64_t rows, int C, int rows_per_chunk, GroupArgs grp,
                                                            BnFold fold) {
    __shared__ __attribute__((aligned(16))) float red[16][64][2];
    static_assert(sizeof(float) * 16 * 64 * 2 >= FOLD_LDS_BYTES, "the fold's ticket re-uses the reduction buffer");
    const int64_t grow0 = (int64_t)blockIdx.z * rows;      // first row of this group (rows = rows of one group)
    dz += grow0 * lddz;
    y += grow0 * ldy;
    if (z) z += grow0 * ldz;
    mean += (size_t)blockIdx.z * grp.pstride;
    invstd += (size_t)blockIdx.z * grp.pstride;
    float* const partial0 = partial;                       // [groups][chunks][C][2]
    partial += (size_t)blockIdx.z * gridDim.x * C * 2;
    const int64_t bq0 = grow0 * (C >> 2);                  // quad index of the group's first element in the bit array
    const int cq = threadIdx.x & 15, rl = threadIdx.x >> 4;
    const int c = blockIdx.y * 64 + cq * 4;
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_chunk;
    const int64_t r1 = r0 + rows_per_chunk < rows ? r0 + rows_per_chunk : rows;
    float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
    if (c < C) {
        const float4 mu = *reinterpret_cast<const float4*>(mean + c);
        auto acc = [&](float4 g, float4 zz, float4 yy) {
            if (relu) {
                g.x = zz.x > 0.f ? g.x : 0.f;
                g.y = zz.y > 0.f ? g.y : 0.f;
                g.z = zz.z > 0.f ? g.z : 0.f;
                g.w = zz.w > 0.f ? g.w : 0.f;
            }
            s1[0] += g.x;
            s1[1] += g.y;
            s1[2] += g.z;
            s1[3] += g.w;
            s2[0] += g.x * (yy.x - mu.x);
            s2[1] += g.y * (yy.y - mu.y);
            s2[2] += g.z * (yy.z - mu.z);
            s2[3] += g.w * (yy.w - mu.w);
        };
        int64_t r = r0 + rl;
        // rows in flight per thread (2 x 16-byte loads + the mask word each): four, or two in fp32 where this pass shares the
        // CUs with the side stream's weight-gradient workgroups and the register count decides how many of its waves fit
        constexpr int U = sizeof(T) == 4 ? 2 : 4;
        for (; r + 16 * (U - 1) < r1; r += 16 * U) {
            float4 g[U], yv[U], zv[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                g[u] = ld4<T>(dz + (r + 16 * u) * lddz + c);
                yv[u] = ld4<T>(y + (r + 16 * u) * ldy + c);
            }
#pragma unroll
            for (int u = 0; u < U; ++u)
                zv[u] = relu ? relu_mask4(bits, bq0 + (r + 16 * u) * (C >> 2) + (c >> 2), z, (r + 16 * u) * ldz + c) : g[u];
#pragma unroll
            for (int u = 0; u < U; ++u) acc(g[u], zv[u], yv[u]);
        }
        for (; r < r1; r += 16) {
            float4 ga = ld4<T>(dz + r * lddz + c);
            float4 ya = ld4<T>(y + r * ldy + c);
            float4 za = relu ? relu_mask4(bits, bq0 + r * (C >> 2) + (c >> 2), z, r * ldz + c) : ga;
            acc(ga, za, ya);
        }
        const float4 is = *reinterpret_cast<const float4*>(invstd + c);
        s2[0] *= is.x;
        s2[1] *= is.y;
        s2[2] *= is.z;
        s2[3] *= is.w;
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        red[rl][cq * 4 + e][0] = s1[e];
        red[rl][cq * 4 + e][1] = s2[e];
    }
    __syncthreads();
    if (threadIdx.x < 128) {
        const int ch = threadIdx.x >> 1, which = threadIdx.x & 1;
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) t += red[k][ch][which];
        const int cc = blockIdx.y * 64 + ch;
        if (cc < C) st_agent(partial + ((size_t)blockIdx.x * C + cc) * 2 + which, t);   // (sc1: the fold's last arriver reads them)
    }
    if (fold.tickets) {   // uniform: this launch also finishes dgamma / dbeta (bn_fold.h)
        const int c0 = blockIdx.y * 64;
        bn_fold_arrive<2>(fold, partial0, blockIdx.x, c0, C - c0 < 64 ? C - c0 : 64, reinterpret_cast<unsigned char*>(&red[0][0][0]),
                          blockIdx.z);
    }
}
// pass 2: one workgroup per channel, thread t sums chunks t, t+256, ... (four independent loads per round), LDS tree;
// fixed order: deterministic.  (Same reasoning as bn_finalize_kernel: the time is the number of dependent load rounds.)
// acc_dgamma / acc_dbeta (optional): running sums over several calls (a BatchNorm used once per frame of the video unroll):
// the per-call sums still go to dgamma / dbeta, pass 3 needs them.
__global__ void __launch_bounds__(256) bn_bwd_finalize_kernel(const float* partial, int chunks, int C, float* dgamma,
                                                              float* dbeta, float* acc_dgamma, float* acc_dbeta, GroupArgs grp) {
    __shared__ float red[256][2];
    const int c = blockIdx.x, t0 = threadIdx.x;
    partial += (size_t)blockIdx.y * chunks * C * 2;
    dgamma += (size_t)blockIdx.y * grp.gstride;
    dbeta += (size_t)blockIdx.y * grp.gstride;
    float a = 0.f, b = 0.f;
    for (int t = t0; t < chunks; t += 1024) {
        float va[4], vb[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int tt = t + 256 * u;
            const float* v = partial + ((size_t)(tt < chunks ? tt : t) * C + c) * 2;
            va[u] = tt < chunks ? v[0] : 0.f;
            vb[u] = tt < chunks ? v[1] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            a += va[u];
            b += vb[u];
        }
    }
    red[t0][0] = a;
    red[t0][1] = b;
    __syncthreads();
#pragma unroll
    for (int off = 128; off >= 1; off >>= 1) {
        if (t0 < off) {
            a += red[t0 + off][0];
            b += red[t0 + off][1];
            red[t0][0] = a;
            red[t0][1] = b;
        }
        __syncthreads();
    }
    if (t0 == 0) {
        dbeta[c] = a;
        dgamma[c] = b;
        if (acc_dgamma) {
            acc_dbeta[c] += a;
            acc_dgamma[c] += b;
        }
    }
}
// pass 3: dy = gamma*invstd*(g - dbeta/M - xhat*dgamma/M);  dres = g
template <typename T>
__global__ void __launch_bounds__(256) bn_bwd_apply_kernel(const T* dz, int lddz, const T* z, int ldz,
                                                           const uint32_t* bits, const T* y, int ldy,
                                                           const float* gamma,
                                                           const float* mean, const float* invstd,
                                                           const float* dgamma, const float* dbeta, int relu,
                                                           int use_batch, float inv_m, T* dy, int lddy,
                                                           T* dres, int lddres, int64_t total, int C4,
                                                           FastDiv fC4) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        uint32_t row = fdiv((uint32_t)i, fC4);
        int c = ((int)i - (int)row * C4) * 4;
        float4 g4 = ld4<T>(dz + (size_t)row * lddz + c);
        float g[4] = {g4.x, g4.y, g4.z, g4.w};
        if (relu) {
            float4 z4 = relu_mask4(bits, i, z, (size_t)row * ldz + c);
            if (!(z4.x > 0.f)) g[0] = 0.f;
            if (!(z4.y > 0.f)) g[1] = 0.f;
            if (!(z4.z > 0.f)) g[2] = 0.f;
            if (!(z4.w > 0.f)) g[3] = 0.f;
        }
        if (dres) st4(dres + (size_t)row * lddres + c, make_float4(g[0], g[1], g[2], g[3]));
        float4 y4 = ld4<T>(y + (size_t)row * ldy + c);
        float yv[4] = {y4.x, y4.y, y4.z, y4.w};
        float o[4];
        const float4 is4 = *reinterpret_cast<const float4*>(invstd + c), ga4 = *reinterpret_cast<const float4*>(gamma + c);
        const float4 mu4 = *reinterpret_cast<const float4*>(mean + c);
        const float4 dg4 = *reinterpret_cast<const float4*>(dgamma + c), db4 = *reinterpret_cast<const float4*>(dbeta + c);
        const float isv[4] = {is4.x, is4.y, is4.z, is4.w}, gav[4] = {ga4.x, ga4.y, ga4.z, ga4.w};
        const float muv[4] = {mu4.x, mu4.y, mu4.z, mu4.w}, dgv[4] = {dg4.x, dg4.y, dg4.z, dg4.w};
        const float dbv[4] = {db4.x, db4.y, db4.z, db4.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float k = gav[e] * isv[e];
            if (use_batch) {
                float xh = (yv[e] - muv[e]) * isv[e];
                o[e] = k * (g[e] - dbv[e] * inv_m - xh * dgv[e] * inv_m);
            } else {
                o[e] = k * g[e];
            }
        }
        st4(dy + (size_t)row * lddy + c, make_float4(o[0], o[1], o[2], o[3]));
    }
}

// bf16 storage, C % 8 == 0: eight channels per thread (see bn_apply8_kernel)
__global__ void __launch_bounds__(256) bn_bwd_apply8_kernel(const bf16_t* dz, int lddz, const bf16_t* z, int ldz,
                                                            const uint32_t* bits, const bf16_t* y, int ldy,
                                                            const float* gamma, const float* mean, const float* invstd,
                                                            const float* dgamma, const float* dbeta, int relu,
                                                            int use_batch, float inv_m, bf16_t* dy, int lddy, bf16_t* dres,
                                                            int lddres, int64_t total, int C8, FastDiv fC8) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const uint32_t row = fdiv((uint32_t)i, fC8);
        const int c = ((int)i - (int)row * C8) * 8;
        F8 g = ld8(dz + (size_t)row * lddz + c);
        const F8 yv = ld8(y + (size_t)row * ldy + c);
        if (relu) {
            if (bits) {
                const uint32_t byte = (bits[i >> 2] >> (8 * (int)(i & 3))) & 0xffu;
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    if (!((byte >> e) & 1u)) g.v[e] = 0.f;
            } else {
                const F8 zz = ld8(z + (size_t)row * ldz + c);
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    if (!(zz.v[e] > 0.f)) g.v[e] = 0.f;
            }
        }
        if (dres) st8(dres + (size_t)row * lddres + c, g);
        const F8 is = ld8(invstd + c), ga = ld8(gamma + c), mu = ld8(mean + c), dg = ld8(dgamma + c), db = ld8(dbeta + c);
        F8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float k = ga.v[e] * is.v[e];
            if (use_batch) {
                const float xh = (yv.v[e] - mu.v[e]) * is.v[e];
                o.v[e] = k * (g.v[e] - db.v[e] * inv_m - xh * dg.v[e] * inv_m);
            } else {
                o.v[e] = k * g.v[e];
            }
        }
        st8(dy + (size_t)row * lddy + c, o);
    }
}

__global__ void __launch_bounds__(256) relu_bwd_kernel(const float* dz, const float* z, float* dx, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
        dx[i] = z[i] > 0.f ? dz[i] : 0.f;
}

// ---- dropout -----------------------------------------------------------------------------
__device__ __forceinline__ uint32_t mix_hash(uint64_t seed, uint64_t idx) {
    uint64_t v = seed ^ (idx * 0x9E3779B97F4A7C15ull);
    v ^= v >> 30;
    v *= 0xBF58476D1CE4E5B9ull;
    v ^= v >> 27;
    v *= 0x94D049BB133111EBull;
    v ^= v >> 31;
    return (uint32_t)(v >> 40);  // 24 random bits
}
template <typename T>
__global__ void __launch_bounds__(256) dropout_fwd_kernel(const T* x, T* y, uint8_t* mask,
                                                          const float* ext, int64_t n, float p, float inv_keep,
                                                          uint64_t seed, const uint64_t* step_dev) {
    // step_dev (optional): a counter in DEVICE memory mixed into the seed — a captured training step (hipGraph) replays this launch
    // with the same kernel arguments, its masks must still differ from step to step (the graph bumps the counter once per replay)
    if (step_dev) seed += *step_dev * 0x9E3779B97F4A7C15ull;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        bool keep;
        if (ext)
            keep = ext[i] != 0.f;
        else
            keep = (float)mix_hash(seed, (uint64_t)i) * (1.0f / 16777216.0f) >= p;
        mask[i] = keep ? 1 : 0;
        st1(y + i, keep ? ld1(x + i) * inv_keep : 0.f);
    }
}
template <typename T>
__global__ void __launch_bounds__(256) dropout_bwd_kernel(const T* dy, const uint8_t* mask, T* dx, int64_t n,
                                                          float inv_keep) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
        st1(dx + i, mask[i] ? ld1(dy + i) * inv_keep : 0.f);
}

// ---- MSE ---------------------------------------------------------------------------------
__device__ __forceinline__ float block_sum_256(float v, float* red) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_down(v, off);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}
__global__ void __launch_bounds__(256) mse_partial_kernel(const float* y, const float* t, float* ws, int64_t n) {
    __shared__ float red[4];
    float s = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        float d = y[i] - t[i];
        s += d * d;
    }
    s = block_sum_256(s, red);
    if (threadIdx.x == 0) ws[blockIdx.x] = s;
}
__global__ void __launch_bounds__(256) mse_final_kernel(const float* ws, int parts, float inv_n, float* loss) {
    __shared__ float red[4];
    float s = 0.f;
    for (int i = threadIdx.x; i < parts; i += 256) s += ws[i];
    s = block_sum_256(s, red);
    if (threadIdx.x == 0) loss[0] = s * inv_n;
}
__global__ void __launch_bounds__(256) mse_bwd_kernel(const float* y, const float* t, const float* dloss, float* dy,
                                                      int64_t n, float two_inv_n) {
    float k = two_inv_n * dloss[0];
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
        dy[i] = (y[i] - t[i]) * k;
}

constexpr int MSE_PARTS = 512;
constexpr int BNB_ROWS = 128;   // small chunks: enough workgroups (and bytes in flight) to saturate HBM on 23x23 maps
                                // (A/B over the whole step: 64 rows 70.8 ms, 128 69.9, 256 69.9)

}  // namespace up

using namespace up;

extern "C" const char* up_last_error(void) { return g_err; }
// ---- CU-masked streams (round 6 experiment: the weight-gradient side stream on a fixed share of the chip) -----------------------
// A stream created with a compute-unit mask runs its kernels only on the CUs whose bit is set.  `words` 32-bit words, bit i of the
// mask = logical CU i in the runtime's enumeration (up_probe_placement shows where the workgroups of such a stream really run).
extern "C" int up_stream_create_cu_mask(const uint32_t* mask, int words, void** stream) {
    UP_REQUIRE(mask && words > 0 && stream, UP_ERR_INVALID, "stream_create_cu_mask: bad argument");
#ifdef UP_EMU
    *stream = nullptr;
    return UP_OK;
#else
    hipStream_t st = nullptr;
    hipError_t e = hipExtStreamCreateWithCUMask(&st, (uint32_t)words, mask);
    if (e != hipSuccess) {
        set_error("stream_create_cu_mask: %s", hipGetErrorString(e));
        return UP_ERR_LAUNCH;
    }
    *stream = st;
    return UP_OK;
#endif
}
extern "C" int up_stream_destroy(void* stream) {
#ifndef UP_EMU
    if (stream && hipStreamDestroy(as_stream(stream)) != hipSuccess) {
        (void)hipGetLastError();
        return UP_ERR_LAUNCH;
    }
#endif
    return UP_OK;
}
namespace up {
__global__ void __launch_bounds__(64) placement_kernel(int* out) {
#ifndef UP_EMU
    if (threadIdx.x == 0) {
        const unsigned xcc = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11));      // HW_REG_XCC_ID, bits 0..3
        const unsigned hw = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));        // HW_REG_HW_ID: cu_id 8..11, sh 12, se 13..15
        out[2 * blockIdx.x] = (int)(xcc & 15u);
        out[2 * blockIdx.x + 1] = (int)((hw >> 8) & 0xffu);                                      // cu | sh << 4 | se << 5
        for (int i = 0; i < 2000; ++i) __builtin_amdgcn_s_sleep(32);                              // stay resident: later blocks must spread
    }
#else
    if (threadIdx.x == 0) out[2 * blockIdx.x] = out[2 * blockIdx.x + 1] = 0;
#endif
}
}  // namespace up
// where do the workgroups of `stream` run?  out[blocks][2] = (XCC id, CU / SH / SE bits of HW_ID) per workgroup (device memory)
extern "C" int up_probe_placement(int blocks, int* out_device, void* stream) {
    UP_REQUIRE(blocks > 0 && out_device, UP_ERR_INVALID, "probe_placement: bad argument");
    hipLaunchKernelGGL(placement_kernel, dim3(blocks), dim3(64), 0, as_stream(stream), out_device);
    return check_launch("probe_placement");
}

extern "C" int up_abi_version(void) { return 10; }   // 10: BatchNorm finalize folded into the producing launch (up_bn_fold, up_bn_reduce_slot.dgamma, up_bn_bwd_finalized_t); 9: up_pack_weights_bf16_batched; 8: row groups (up_conv2d_fwd_grouped, groups in up_dgrad_epilogue, up_bn_bwd_groups_prereduced_t); 7: up_conv2d_bwd_data_ex / up_bn_bwd_prereduced_t / up_conv_counter; 6: centred BatchNorm; 5: up_conv2d_bwd_weight_acc; 4: element-typed (_t) twins + bf16 storage

extern "C" int up_bn_eval_coeffs(const float* gamma, const float* beta, const float* rm, const float* rv, float eps,
                                 int C, float* scale, float* shift, void* stream) {
    UP_REQUIRE(gamma && beta && rm && rv && scale && shift && C > 0, UP_ERR_INVALID, "bn_eval_coeffs: bad argument");
    hipLaunchKernelGGL(bn_eval_coeffs_kernel, dim3(cdiv(C, 256)), dim3(256), 0, as_stream(stream), gamma, beta, rm,
                       rv, eps, C, scale, shift);
    return check_launch("bn_eval_coeffs");
}

extern "C" int up_bn_finalize(const float* stats, int tiles, int C, float eps, float momentum, float* rm, float* rv,
                              const float* gamma, const float* beta, float* mean, float* invstd, float* scale,
                              float* shift, void* stream) {
    UP_REQUIRE(stats && gamma && beta && mean && invstd && scale && shift && tiles > 0 && C > 0, UP_ERR_INVALID,
               "bn_finalize: bad argument");
    UP_REQUIRE((rm == nullptr) == (rv == nullptr), UP_ERR_INVALID, "bn_finalize: running stats must come in pairs");
    BnFold f;
    memset(&f, 0, sizeof(f));
    UP_REQUIRE(bn_fold_scratch(as_stream(stream), tiles, C, 3, &f), UP_ERR_WORKSPACE,
               "bn_finalize: %d partial rows x %d channels exceed the per-stream merge scratch", tiles, C);
    f.eps = eps;
    f.mom = momentum;
    f.rm = rm;
    f.rv = rv;
    f.gamma = gamma;
    f.beta = beta;
    f.mean = mean;
    f.invstd = invstd;
    f.scale = scale;
    f.shift = shift;
    hipLaunchKernelGGL(bn_fold_arrive_kernel<3>, dim3(tiles, cdiv(C, FOLD_COLS)), dim3(256), 0, as_stream(stream), stats, f);
    return check_launch("bn_finalize");
}

// mean == nullptr: z = y * scale + shift;  else the centred form z = (y - mean) * scale + shift with shift = the BatchNorm bias
static int bn_apply_impl(const void* y, int ldy, const float* mean, const float* scale, const float* shift, const void* res,
                         int ldr, int relu, void* z, int ldz, uint32_t* relu_bits, int64_t rows, int C, int dtype,
                         void* stream) {
    UP_REQUIRE(y && scale && shift && z && rows > 0 && C > 0, UP_ERR_INVALID, "bn_apply: bad argument");
    UP_REQUIRE(C % 4 == 0 && ldy % 4 == 0 && ldz % 4 == 0 && (!res || ldr % 4 == 0), UP_ERR_INVALID,
               "bn_apply: C and strides must be multiples of 4");
    UP_REQUIRE(rows * (C / 4) < (1ll << 31), UP_ERR_UNSUPPORTED, "bn_apply: tensor too large");
    UP_REQUIRE(dtype == UP_DT_F32 || dtype == UP_DT_BF16, UP_ERR_INVALID, "bn_apply: dtype %d", dtype);
    int64_t total = rows * (C / 4);
    {
        dim3 grid;
        int lcs = 0;
        if (bn_rows_enabled() && dtype == UP_DT_BF16 && ldy % 8 == 0 && ldz % 8 == 0 && (!res || ldr % 8 == 0) &&
            rows_geometry<bf16_t>(rows, C, grid, lcs)) {
            hipLaunchKernelGGL(bn_apply_rows_kernel<bf16_t>, grid, dim3(256), 0, as_stream(stream), (const bf16_t*)y, ldy, scale,
                               shift, mean, (const bf16_t*)res, ldr, relu, (bf16_t*)z, ldz, relu_bits, (int)rows, C, lcs, GroupArgs{0, 0});
            return check_launch("bn_apply");
        }
        if (bn_rows_enabled() && dtype == UP_DT_F32 && rows_geometry<float>(rows, C, grid, lcs)) {
            hipLaunchKernelGGL(bn_apply_rows_kernel<float>, grid, dim3(256), 0, as_stream(stream), (const float*)y, ldy, scale,
                               shift, mean, (const float*)res, ldr, relu, (float*)z, ldz, relu_bits, (int)rows, C, lcs, GroupArgs{0, 0});
            return check_launch("bn_apply");
        }
    }
    if (dtype == UP_DT_BF16 && C % 8 == 0 && ldy % 8 == 0 && ldz % 8 == 0 && (!res || ldr % 8 == 0))
        hipLaunchKernelGGL(bn_apply8_kernel, dim3(grid_for(total / 2)), dim3(256), 0, as_stream(stream), (const bf16_t*)y, ldy,
                           scale, shift, mean, (const bf16_t*)res, ldr, relu, (bf16_t*)z, ldz, relu_bits, total / 2, C / 8,
                           make_fastdiv(C / 8));
    else if (dtype == UP_DT_BF16)
        hipLaunchKernelGGL(bn_apply_kernel<bf16_t>, dim3(grid_for(total)), dim3(256), 0, as_stream(stream),
                           (const bf16_t*)y, ldy, scale, shift, mean, (const bf16_t*)res, ldr, relu, (bf16_t*)z, ldz, relu_bits,
                           total, C / 4, make_fastdiv(C / 4));
    else
        hipLaunchKernelGGL(bn_apply_kernel<float>, dim3(grid_for(total)), dim3(256), 0, as_stream(stream), (const float*)y,
                           ldy, scale, shift, mean, (const float*)res, ldr, relu, (float*)z, ldz, relu_bits, total, C / 4,
                           make_fastdiv(C / 4));
    return check_launch("bn_apply");
}
extern "C" int up_bn_apply_t(const void* y, int ldy, const float* scale, const float* shift, const void* res,
                             int ldr, int relu, void* z, int ldz, uint32_t* relu_bits, int64_t rows, int C, int dtype,
                             void* stream) {
    return bn_apply_impl(y, ldy, nullptr, scale, shift, res, ldr, relu, z, ldz, relu_bits, rows, C, dtype, stream);
}
extern "C" int up_bn_apply_centered_t(const void* y, int ldy, const float* mean, const float* scale, const float* beta,
                                      const void* res, int ldr, int relu, void* z, int ldz, uint32_t* relu_bits, int64_t rows,
                                      int C, int dtype, void* stream) {
    UP_REQUIRE(mean, UP_ERR_INVALID, "bn_apply_centered: null mean");
    return bn_apply_impl(y, ldy, mean, scale, beta, res, ldr, relu, z, ldz, relu_bits, rows, C, dtype, stream);
}
extern "C" int up_bn_apply(const float* y, int ldy, const float* scale, const float* shift, const float* res,
                           int ldr, int relu, float* z, int ldz, uint32_t* relu_bits, int64_t rows, int C,
                           void* stream) {
    return up_bn_apply_t(y, ldy, scale, shift, res, ldr, relu, z, ldz, relu_bits, rows, C, UP_DT_F32, stream);
}

extern "C" size_t up_bn_bwd_workspace(int64_t rows, int C) {
    return (size_t)cdiv(rows, BNB_ROWS) * C * 2 * sizeof(float);
}

namespace up {
template <typename T>
static void launch_bn_bwd(const T* dz, int lddz, const T* z, int ldz, const uint32_t* relu_bits, const T* y, int ldy,
                          const float* gamma, const float* mean, const float* invstd, int relu, int use_batch_stats, T* dy,
                          int lddy, T* dres, int lddres, float* dgamma, float* dbeta, float* acc_dgamma, float* acc_dbeta,
                          float* workspace, int64_t rows, int C, hipStream_t st, int prereduced_chunks = 0) {
    // prereduced_chunks > 0: `workspace` already holds that many rows of partial sums [chunk][C][2] — the data-gradient launch
    // that produced dz reduced them in its epilogue (f32_glds.h BNRED) — so pass 1 is skipped;  < 0: dgamma / dbeta are FINAL
    // (that launch, or the reduce pass below, also carried the merge ticket of bn_fold.h): the apply pass alone.
    // Pass 2 is the fold's merge tree in every form (one set of bits whoever runs it); only the accumulating variant of the video
    // unroll (acc_dgamma) keeps bn_bwd_finalize_kernel.
    if (prereduced_chunks >= 0) {
        int chunks = prereduced_chunks > 0 ? prereduced_chunks : cdiv(rows, BNB_ROWS);
        BnFold f;
        memset(&f, 0, sizeof(f));
        const bool tree = !acc_dgamma && bn_fold_scratch(st, chunks, C, 2, &f);
        f.dgamma = dgamma;
        f.dbeta = dbeta;
        const bool carried = tree && prereduced_chunks == 0 && bn_fold_enabled();   // the reduce pass takes the tickets itself
        BnFold none;
        memset(&none, 0, sizeof(none));
        if (prereduced_chunks == 0)
            hipLaunchKernelGGL(bn_bwd_reduce_kernel<T>, dim3(chunks, cdiv(C, 64)), dim3(256), 0, st, dz, lddz, z, ldz, relu_bits,
                               y, ldy, mean, invstd, relu, workspace, rows, C, BNB_ROWS, GroupArgs{0, 0}, carried ? f : none);
        if (tree && !carried)
            hipLaunchKernelGGL(bn_fold_arrive_kernel<2>, dim3(chunks, cdiv(C, FOLD_COLS)), dim3(256), 0, st, (const float*)workspace, f);
        else if (!tree)
            hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(C), dim3(256), 0, st, (const float*)workspace, chunks, C,
                               dgamma, dbeta, acc_dgamma, acc_dbeta, GroupArgs{0, 0});
    }
    int64_t total = rows * (C / 4);
    {
        dim3 grid;
        int lcs = 0;
        constexpr int E = 16 / (int)sizeof(T);
        if (bn_rows_enabled() && lddz % E == 0 && ldy % E == 0 && lddy % E == 0 && (!relu || relu_bits || ldz % E == 0) &&
            (!dres || lddres % E == 0) && rows_geometry<T>(rows, C, grid, lcs)) {
            launch_bn_bwd_apply_rows<T>(grid, st, dz, lddz, z, ldz, relu_bits, y, ldy, gamma, mean, invstd, (const float*)dgamma,
                                        (const float*)dbeta, relu, use_batch_stats, 1.0f / (float)rows, dy, lddy, dres, lddres,
                                        (int)rows, C, lcs, GroupArgs{0, 0});
            return;
        }
    }
    if constexpr (sizeof(T) == 2) {
        if (C % 8 == 0 && lddz % 8 == 0 && ldy % 8 == 0 && lddy % 8 == 0 && (!relu || relu_bits || ldz % 8 == 0) &&
            (!dres || lddres % 8 == 0)) {
            hipLaunchKernelGGL(bn_bwd_apply8_kernel, dim3(grid_for(total / 2)), dim3(256), 0, st, dz, lddz, z, ldz, relu_bits,
                               y, ldy, gamma, mean, invstd, (const float*)dgamma, (const float*)dbeta, relu, use_batch_stats,
                               1.0f / (float)rows, dy, lddy, dres, lddres, total / 2, C / 8, make_fastdiv(C / 8));
            return;
        }
    }
    hipLaunchKernelGGL(bn_bwd_apply_kernel<T>, dim3(grid_for(total)), dim3(256), 0, st, dz, lddz, z, ldz, relu_bits, y,
                       ldy, gamma, mean, invstd, (const float*)dgamma, (const float*)dbeta, relu, use_batch_stats,
                       1.0f / (float)rows, dy, lddy, dres, lddres, total, C / 4, make_fastdiv(C / 4));
}
}  // namespace up

extern "C" int up_bn_bwd_t(const void* dz, int lddz, const void* z, int ldz, const uint32_t* relu_bits, const void* y,
                           int ldy, const float* gamma, const float* mean, const float* invstd, int relu,
                           int use_batch_stats, void* dy, int lddy, void* dres, int lddres, float* dgamma, float* dbeta,
                           float* workspace, size_t workspace_bytes, int64_t rows, int C, int dtype, void* stream) {
    return up_bn_bwd_acc_t(dz, lddz, z, ldz, relu_bits, y, ldy, gamma, mean, invstd, relu, use_batch_stats, dy, lddy, dres,
                           lddres, dgamma, dbeta, nullptr, nullptr, workspace, workspace_bytes, rows, C, dtype, stream);
}
extern "C" int up_bn_bwd_acc_t(const void* dz, int lddz, const void* z, int ldz, const uint32_t* relu_bits, const void* y,
                               int ldy, const float* gamma, const float* mean, const float* invstd, int relu,
                               int use_batch_stats, void* dy, int lddy, void* dres, int lddres, float* dgamma,
                               float* dbeta, float* acc_dgamma, float* acc_dbeta, float* workspace, size_t workspace_bytes,
                               int64_t rows, int C, int dtype, void* stream) {
    UP_REQUIRE(!acc_dgamma == !acc_dbeta, UP_ERR_INVALID, "bn_bwd: acc_dgamma and acc_dbeta come together");
    UP_REQUIRE(dz && y && gamma && mean && invstd && dy && dgamma && dbeta && workspace, UP_ERR_INVALID,
               "bn_bwd: null pointer");
    UP_REQUIRE(!relu || z || relu_bits, UP_ERR_INVALID, "bn_bwd: relu needs the forward output z or its sign bits");
    UP_REQUIRE(C % 4 == 0 && lddz % 4 == 0 && ldy % 4 == 0 && lddy % 4 == 0 && (!relu || relu_bits || ldz % 4 == 0) &&
                   (!dres || lddres % 4 == 0),
               UP_ERR_INVALID, "bn_bwd: C and strides must be multiples of 4");
    UP_REQUIRE(rows * (C / 4) < (1ll << 31), UP_ERR_UNSUPPORTED, "bn_bwd: tensor too large");
    UP_REQUIRE(workspace_bytes >= up_bn_bwd_workspace(rows, C), UP_ERR_WORKSPACE, "bn_bwd: workspace too small");
    UP_REQUIRE(dtype == UP_DT_F32 || dtype == UP_DT_BF16, UP_ERR_INVALID, "bn_bwd: dtype %d", dtype);
    hipStream_t st = as_stream(stream);
    if (dtype == UP_DT_BF16)
        launch_bn_bwd<bf16_t>((const bf16_t*)dz, lddz, (const bf16_t*)z, ldz, relu_bits, (const bf16_t*)y, ldy, gamma, mean,
                              invstd, relu, use_batch_stats, (bf16_t*)dy, lddy, (bf16_t*)dres, lddres, dgamma, dbeta,
                              acc_dgamma, acc_dbeta, workspace, rows, C, st);
    else
        launch_bn_bwd<float>((const float*)dz, lddz, (const float*)z, ldz, relu_bits, (const float*)y, ldy, gamma, mean,
                             invstd, relu, use_batch_stats, (float*)dy, lddy, (float*)dres, lddres, dgamma, dbeta,
                             acc_dgamma, acc_dbeta, workspace, rows, C, st);
    return check_launch("bn_bwd");
}
// BatchNorm backward whose reduction pass already ran inside the data-gradient launch that produced dz
// (up_conv2d_bwd_data_ex): `partial` = [chunks][C][2] from that launch; finalize + apply only.
extern "C" int up_bn_bwd_prereduced_t(const void* dz, int lddz, const uint32_t* relu_bits, const void* y, int ldy,
                                      const float* gamma, const float* mean, const float* invstd, int relu, int use_batch_stats,
                                      void* dy, int lddy, void* dres, int lddres, float* dgamma, float* dbeta, float* acc_dgamma,
                                      float* acc_dbeta, float* partial, int chunks, int64_t rows, int C, int dtype, void* stream) {
    UP_REQUIRE(!acc_dgamma == !acc_dbeta, UP_ERR_INVALID, "bn_bwd_prereduced: acc_dgamma and acc_dbeta come together");
    UP_REQUIRE(dz && y && gamma && mean && invstd && dy && dgamma && dbeta && partial && chunks > 0, UP_ERR_INVALID,
               "bn_bwd_prereduced: null pointer / no partial rows");
    UP_REQUIRE(!relu || relu_bits, UP_ERR_INVALID, "bn_bwd_prereduced: relu needs the sign bits of the forward output");
    UP_REQUIRE(C % 4 == 0 && lddz % 4 == 0 && ldy % 4 == 0 && lddy % 4 == 0 && (!dres || lddres % 4 == 0), UP_ERR_INVALID,
               "bn_bwd_prereduced: C and strides must be multiples of 4");
    UP_REQUIRE(rows * (C / 4) < (1ll << 31), UP_ERR_UNSUPPORTED, "bn_bwd_prereduced: tensor too large");
    UP_REQUIRE(dtype == UP_DT_F32 || dtype == UP_DT_BF16, UP_ERR_INVALID, "bn_bwd_prereduced: dtype %d", dtype);
    if (dtype == UP_DT_BF16)
        launch_bn_bwd<bf16_t>((const bf16_t*)dz, lddz, (const bf16_t*)nullptr, 0, relu_bits, (const bf16_t*)y, ldy, gamma, mean, invstd,
                              relu, use_batch_stats, (bf16_t*)dy, lddy, (bf16_t*)dres, lddres, dgamma, dbeta, acc_dgamma, acc_dbeta,
                              partial, rows, C, as_stream(stream), chunks);
    else
        launch_bn_bwd<float>((const float*)dz, lddz, (const float*)nullptr, 0, relu_bits, (const float*)y, ldy, gamma, mean, invstd, relu,
                             use_batch_stats, (float*)dy, lddy, (float*)dres, lddres, dgamma, dbeta, acc_dgamma, acc_dbeta, partial,
                             rows, C, as_stream(stream), chunks);
    return check_launch("bn_bwd_prereduced");
}
// BatchNorm backward whose dgamma / dbeta are already FINAL: the data-gradient launch that produced dz carried the reduction AND
// its merge ticket (up_bn_reduce_slot.dgamma / dbeta / folded, ABI 10); the apply pass alone.
extern "C" int up_bn_bwd_finalized_t(const void* dz, int lddz, const uint32_t* relu_bits, const void* y, int ldy, const float* gamma,
                                     const float* mean, const float* invstd, int relu, int use_batch_stats, void* dy, int lddy,
                                     void* dres, int lddres, const float* dgamma, const float* dbeta, int64_t rows, int C, int dtype,
                                     void* stream) {
    UP_REQUIRE(dz && y && gamma && mean && invstd && dy && dgamma && dbeta, UP_ERR_INVALID, "bn_bwd_finalized: null pointer");
    UP_REQUIRE(!relu || relu_bits, UP_ERR_INVALID, "bn_bwd_finalized: relu needs the sign bits of the forward output");
    UP_REQUIRE(C % 4 == 0 && lddz % 4 == 0 && ldy % 4 == 0 && lddy % 4 == 0 && (!dres || lddres % 4 == 0), UP_ERR_INVALID,
               "bn_bwd_finalized: C and strides must be multiples of 4");
    UP_REQUIRE(rows * (C / 4) < (1ll << 31), UP_ERR_UNSUPPORTED, "bn_bwd_finalized: tensor too large");
    UP_REQUIRE(dtype == UP_DT_F32 || dtype == UP_DT_BF16, UP_ERR_INVALID, "bn_bwd_finalized: dtype %d", dtype);
    if (dtype == UP_DT_BF16)
        launch_bn_bwd<bf16_t>((const bf16_t*)dz, lddz, (const bf16_t*)nullptr, 0, relu_bits, (const bf16_t*)y, ldy, gamma, mean, invstd,
                              relu, use_batch_stats, (bf16_t*)dy, lddy, (bf16_t*)dres, lddres, const_cast<float*>(dgamma),
                              const_cast<float*>(dbeta), nullptr, nullptr, nullptr, rows, C, as_stream(stream), -1);
    else
        launch_bn_bwd<float>((const float*)dz, lddz, (const float*)nullptr, 0, relu_bits, (const float*)y, ldy, gamma, mean, invstd, relu,
                             use_batch_stats, (float*)dy, lddy, (float*)dres, lddres, const_cast<float*>(dgamma),
                             const_cast<float*>(dbeta), nullptr, nullptr, nullptr, rows, C, as_stream(stream), -1);
    return check_launch("bn_bwd_finalized");
}
extern "C" int up_bn_bwd(const float* dz, int lddz, const float* z, int ldz, const uint32_t* relu_bits,
                         const float* y, int ldy,
                         const float* gamma, const float* mean, const float* invstd, int relu, int use_batch_stats,
                         float* dy, int lddy, float* dres, int lddres, float* dgamma, float* dbeta,
                         float* workspace, size_t workspace_bytes, int64_t rows, int C, void* stream) {
    return up_bn_bwd_t(dz, lddz, z, ldz, relu_bits, y, ldy, gamma, mean, invstd, relu, use_batch_stats, dy, lddy, dres,
                       lddres, dgamma, dbeta, workspace, workspace_bytes, rows, C, UP_DT_F32, stream);
}

// Grouped BatchNorm with up to BN_MAXG row groups: coefficients of every group + the running statistics (the groups' momentum
// updates in order) / the per-group and total backward sums come from ONE merge (bn_fold.h with row groups) — folded into the
// producing launch or stand-alone (bn_fold_arrive_kernel with grid.z = groups); more groups take the per-group kernels below.
constexpr int BN_MAXG = 8;
// running statistics after `groups` batches, in order (one thread per channel): the momentum updates of `groups` module calls
__global__ void __launch_bounds__(256) bn_running_groups_kernel(const float* coef, int groups, int C, float n, float eps, float mom,
                                                                float* rm, float* rv) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    float m = rm[c], v = rv[c];
    for (int g = 0; g < groups; ++g) {
        const float mean = coef[(size_t)g * 4 * C + c], is = coef[(size_t)g * 4 * C + C + c];
        const float var = 1.0f / (is * is) - eps;
        const float unb = n > 1.f ? var * (n / (n - 1.f)) : var;
        m = (1.f - mom) * m + mom * mean;
        v = (1.f - mom) * v + mom * unb;
    }
    rm[c] = m;
    rv[c] = v;
}
// parameter gradients = sums of the per-group sums, in group order (gsum[g] = dgamma | dbeta of group g)
__global__ void __launch_bounds__(256) bn_sum_groups_kernel(const float* gsum, int groups, int C, float* dgamma, float* dbeta) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    float a = 0.f, b = 0.f;
    for (int g = 0; g < groups; ++g) {
        a += gsum[(size_t)g * 2 * C + c];
        b += gsum[(size_t)g * 2 * C + C + c];
    }
    dgamma[c] = a;
    dbeta[c] = b;
}

// ---- grouped BatchNorm: G row groups of equal size in one tensor, each normalised with its own batch statistics ------------
extern "C" int up_bn_batch_stats_tiles(int64_t rows_per_group) { return cdiv(rows_per_group, BNS_ROWS); }
namespace up {
static void launch_bn_batch_stats(const void* y, int ldy, int64_t rows_per_group, int C, int groups, int dtype, float* stats,
                                  const BnFold& fold, hipStream_t st) {
    const int tiles = cdiv(rows_per_group, BNS_ROWS);
    dim3 grid(tiles, cdiv(C, 64), groups);
    if (dtype == UP_DT_BF16)
        hipLaunchKernelGGL(bn_batch_stats_kernel<bf16_t>, grid, dim3(256), 0, st, (const bf16_t*)y, ldy, (int)rows_per_group, C, stats,
                           tiles, fold);
    else
        hipLaunchKernelGGL(bn_batch_stats_kernel<float>, grid, dim3(256), 0, st, (const float*)y, ldy, (int)rows_per_group, C, stats,
                           tiles, fold);
}
}  // namespace up
extern "C" int up_bn_batch_stats_t(const void* y, int ldy, int64_t rows_per_group, int C, int groups, int dtype, float* stats,
                                   void* stream) {
    UP_REQUIRE(y && stats && rows_per_group > 0 && rows_per_group < (1ll << 31) && C > 0 && groups > 0 && groups <= 65535,
               UP_ERR_INVALID, "bn_batch_stats: bad argument");
    UP_REQUIRE(C % 4 == 0 && ldy % 4 == 0, UP_ERR_INVALID, "bn_batch_stats: C and ldy must be multiples of 4");
    UP_REQUIRE(dtype == UP_DT_F32 || dtype == UP_DT_BF16, UP_ERR_INVALID, "bn_batch_stats: dtype %d", dtype);
    BnFold none;
    memset(&none, 0, sizeof(none));
    launch_bn_batch_stats(y, ldy, rows_per_group, C, groups, dtype, stats, none, as_stream(stream));
    return check_launch("bn_batch_stats");
}
// ABI 10: up_bn_batch_stats_t + up_bn_finalize_groups as ONE launch where the fold applies (bn_fold.h with row groups: the statistics
// pass's last workgroup per channel column merges every group's partial rows and writes coef[groups][4][C] + the running
// statistics, the groups' momentum updates in order); else the two launches.
extern "C" int up_bn_stats_groups_t(const void* y, int ldy, int64_t rows_per_group, int C, int groups, int dtype, float* stats,
                                    float eps, float momentum, float* rm, float* rv, const float* gamma, const float* beta, float* coef,
                                    void* stream) {
    UP_REQUIRE(y && stats && gamma && beta && coef && rows_per_group > 0 && rows_per_group < (1ll << 31) && C > 0 && groups > 0 &&
                   groups <= 65535, UP_ERR_INVALID, "bn_stats_groups: bad argument");
    UP_REQUIRE(C % 4 == 0 && ldy % 4 == 0, UP_ERR_INVALID, "bn_stats_groups: C and ldy must be multiples of 4");
    UP_REQUIRE(dtype == UP_DT_F32 || dtype == UP_DT_BF16, UP_ERR_INVALID, "bn_stats_groups: dtype %d", dtype);
    UP_REQUIRE((rm == nullptr) == (rv == nullptr), UP_ERR_INVALID, "bn_stats_groups: running stats must come in pairs");
    const int tiles = cdiv(rows_per_group, BNS_ROWS);
    BnFold f;
    memset(&f, 0, sizeof(f));
    if (bn_fold_enabled() && groups <= BN_MAXG && bn_fold_scratch(as_stream(stream), tiles, C, 3, &f, groups)) {
        f.eps = eps;
        f.mom = momentum;
        f.rm = rm;
        f.rv = rv;
        f.gamma = gamma;
        f.beta = beta;
        f.mean = coef;
        f.invstd = coef + C;
        f.scale = coef + 2 * C;
        f.shift = coef + 3 * C;
        f.ostride = 4 * C;
        launch_bn_batch_stats(y, ldy, rows_per_group, C, groups, dtype, stats, f, as_stream(stream));
        return check_launch("bn_stats_groups");
    }
    if (int e = up_bn_batch_stats_t(y, ldy, rows_per_group, C, groups, dtype, stats, stream)) return e;
    return up_bn_finalize_groups(stats, tiles, C, groups, rows_per_group, eps, momentum, rm, rv, gamma, beta, coef, stream);
}
extern "C" int up_bn_exact_stats_t(const void* y, int ldy, int64_t rows_per_group, int C, int groups, int dtype, float* stats,
                                   void* stream) {
    UP_REQUIRE(y && stats && rows_per_group > 0 && rows_per_group <= 4096 && C > 0 && groups > 0 && groups <= 65535, UP_ERR_INVALID,
               "bn_exact_stats: bad argument (1..4096 rows per group)");
    UP_REQUIRE(dtype == UP_DT_F32 || dtype == UP_DT_BF16, UP_ERR_INVALID, "bn_exact_stats: dtype %d", dtype);
    dim3 grid(cdiv(C, 256), groups);
    if (dtype == UP_DT_BF16)
        hipLaunchKernelGGL(bn_exact_stats_kernel<bf16_t>, grid, dim3(256), 0, as_stream(stream), (const bf16_t*)y, ldy,
                           (int)rows_per_group, C, stats);
    else
        hipLaunchKernelGGL(bn_exact_stats_kernel<float>, grid, dim3(256), 0, as_stream(stream), (const float*)y, ldy,
                           (int)rows_per_group, C, stats);
    return check_launch("bn_exact_stats");
}
// coef: [groups][4][C] = mean, invstd, scale, shift per group.  The groups are finalised IN ORDER on the stream, so the running
// statistics receive the same sequence of momentum updates as `groups` separate forward calls.
extern "C" int up_bn_finalize_groups(const float* stats, int tiles, int C, int groups, int64_t rows_per_group, float eps,
                                     float momentum, float* rm, float* rv, const float* gamma, const float* beta, float* coef,
                                     void* stream) {
    UP_REQUIRE(stats && gamma && beta && coef && tiles > 0 && C > 0 && groups > 0 && groups <= 65535 && rows_per_group > 0,
               UP_ERR_INVALID, "bn_finalize_groups: bad argument");
    UP_REQUIRE((rm == nullptr) == (rv == nullptr), UP_ERR_INVALID, "bn_finalize_groups: running stats must come in pairs");
    // every group's partials are merged by its own workgroups (grid C x groups); the running statistics then take the groups'
    // momentum updates in order (the unbiased variance is recovered from invstd: var = 1 / invstd^2 - eps)
    BnFold f;
    memset(&f, 0, sizeof(f));
    if (groups <= BN_MAXG && bn_fold_scratch(as_stream(stream), tiles, C, 3, &f, groups)) {
        // one launch, the fold's merge tree stand-alone (bn_fold.h with row groups: one workgroup per partial row that only
        // arrives): coefficients of every group + the running statistics — the bits up_bn_stats_groups_t's folded form writes
        f.eps = eps;
        f.mom = momentum;
        f.rm = rm;
        f.rv = rv;
        f.gamma = gamma;
        f.beta = beta;
        f.mean = coef;
        f.invstd = coef + C;
        f.scale = coef + 2 * C;
        f.shift = coef + 3 * C;
        f.ostride = 4 * C;
        hipLaunchKernelGGL(bn_fold_arrive_kernel<3>, dim3(tiles, cdiv(C, FOLD_COLS), groups), dim3(256), 0, as_stream(stream), stats, f);
        return check_launch("bn_finalize_groups");
    }
    hipLaunchKernelGGL(bn_finalize_kernel, dim3(C, groups), dim3(256), 0, as_stream(stream), stats, tiles, C, eps, momentum,
                       (float*)nullptr, (float*)nullptr, gamma, beta, coef, coef + C, coef + 2 * C, coef + 3 * C, GroupArgs{4 * C, 0});
    if (rm)
        hipLaunchKernelGGL(bn_running_groups_kernel, dim3(cdiv(C, 256)), dim3(256), 0, as_stream(stream), (const float*)coef, groups,
                           C, (float)rows_per_group, eps, momentum, rm, rv);
    return check_launch("bn_finalize_groups");
}
extern "C" int up_bn_apply_groups_t(const void* y, int ldy, const float* coef, const float* beta, const void* res, int ldr, int relu,
                                    void* z, int ldz, uint32_t* relu_bits, int64_t rows_per_group, int C, int groups, int dtype,
                                    void* stream) {
    // beta given: centred form (y - mean_g) * scale_g + beta;  else y * scale_g + shift_g
    const float* gmean = beta ? coef : nullptr;
    const float* gshift = beta ? beta : coef + 3 * C;
    UP_REQUIRE(y && z && coef && groups > 0 && groups <= 65535 && rows_per_group > 0, UP_ERR_INVALID, "bn_apply_groups: bad argument");
    UP_REQUIRE(dtype == UP_DT_F32 || dtype == UP_DT_BF16, UP_ERR_INVALID, "bn_apply_groups: dtype %d", dtype);
    UP_REQUIRE(!relu_bits || (rows_per_group * C) % 32 == 0, UP_ERR_UNSUPPORTED,
               "bn_apply_groups: rows_per_group * C must be a multiple of 32 (a group's ReLU bits start on a word)");
    const size_t es = dtype == UP_DT_BF16 ? 2 : 4;
    dim3 grid;
    int lcs = 0;
    const GroupArgs ga{4 * C, 0};
    if (dtype == UP_DT_F32 && ldy % 4 == 0 && ldz % 4 == 0 && (!res || ldr % 4 == 0) && rows_geometry<float>(rows_per_group, C, grid, lcs)) {
        grid.z = groups;   // ONE launch: blockIdx.z = group
        hipLaunchKernelGGL(bn_apply_rows_kernel<float>, grid, dim3(256), 0, as_stream(stream), (const float*)y, ldy, coef + 2 * C,
                           gshift, gmean, (const float*)res, ldr, relu, (float*)z, ldz, relu_bits, (int)rows_per_group, C, lcs, ga);
        return check_launch("bn_apply_groups");
    }
    if (dtype == UP_DT_BF16 && ldy % 8 == 0 && ldz % 8 == 0 && (!res || ldr % 8 == 0) &&
        rows_geometry<bf16_t>(rows_per_group, C, grid, lcs)) {
        grid.z = groups;
        hipLaunchKernelGGL(bn_apply_rows_kernel<bf16_t>, grid, dim3(256), 0, as_stream(stream), (const bf16_t*)y, ldy, coef + 2 * C,
                           gshift, gmean, (const bf16_t*)res, ldr, relu, (bf16_t*)z, ldz, relu_bits, (int)rows_per_group, C, lcs, ga);
        return check_launch("bn_apply_groups");
    }
    for (int g = 0; g < groups; ++g) {       // channel counts without a row-strided geometry: one flat launch per group
        const float* cg = coef + (size_t)g * 4 * C;
        const size_t r0 = (size_t)g * rows_per_group;
        if (int e = bn_apply_impl((const char*)y + r0 * ldy * es, ldy, beta ? cg : nullptr, cg + 2 * C, beta ? beta : cg + 3 * C,
                                  res ? (const char*)res + r0 * ldr * es : nullptr, ldr, relu, (char*)z + r0 * ldz * es, ldz,
                                  relu_bits ? relu_bits + r0 * C / 32 : nullptr, rows_per_group, C, dtype, stream))
            return e;
    }
    return UP_OK;
}
extern "C" size_t up_bn_bwd_groups_workspace(int64_t rows_per_group, int C, int groups) {
    return (size_t)groups * (up_bn_bwd_workspace(rows_per_group, C) + (size_t)2 * C * sizeof(float));
}
namespace up {
template <typename T>
static bool launch_bn_bwd_groups(const T* dz, int lddz, const uint32_t* relu_bits, const T* y, int ldy, const float* gamma,
                                 const float* coef, int relu, T* dy, int lddy, T* dres, int lddres, float* dgamma, float* dbeta,
                                 float* workspace, int64_t rows, int C, int groups, hipStream_t st,
                                 const float* prereduced = nullptr, int prereduced_tiles = 0, const float* finalized_gsum = nullptr) {
    constexpr int E = 16 / (int)sizeof(T);
    dim3 grid;
    int lcs = 0;
    if (!(lddz % E == 0 && ldy % E == 0 && lddy % E == 0 && (!dres || lddres % E == 0) && rows_geometry<T>(rows, C, grid, lcs)))
        return false;
    // prereduced: the data-gradient launch that produced dz (tiled per group, up_conv2d_bwd_data_ex with groups) already wrote
    // [groups][prereduced_tiles][C][2] — pass 1 is skipped;  finalized_gsum: that launch also merged them (bn_fold.h, row groups):
    // gsum / dgamma / dbeta are final, the apply pass alone
    const int chunks = prereduced ? prereduced_tiles : cdiv(rows, BNB_ROWS);
    float* gsum = finalized_gsum ? const_cast<float*>(finalized_gsum) : workspace;      // [groups][dgamma | dbeta]
    const float* partial = prereduced ? prereduced : workspace + (size_t)groups * 2 * C;       // [groups][chunks][C][2]
    const GroupArgs ga{4 * C, 2 * C};
    if (!finalized_gsum) {
        BnFold f;
        memset(&f, 0, sizeof(f));
        const bool tree = groups <= BN_MAXG && bn_fold_scratch(st, chunks, C, 2, &f, groups);
        f.dgamma = dgamma;
        f.dbeta = dbeta;
        f.gsum = gsum;
        f.ostride = 2 * C;
        const bool carried = tree && !prereduced && bn_fold_enabled();     // the reduce pass takes the tickets itself
        BnFold none;
        memset(&none, 0, sizeof(none));
        if (!prereduced)
            hipLaunchKernelGGL(bn_bwd_reduce_kernel<T>, dim3(chunks, cdiv(C, 64), groups), dim3(256), 0, st, dz, lddz, (const T*)nullptr, 0,
                               relu_bits, y, ldy, coef, coef + C, relu, workspace + (size_t)groups * 2 * C, rows, C, BNB_ROWS, ga,
                               carried ? f : none);
        if (carried) {
            // merged inside the reduce pass
        } else if (tree) {      // the same merge tree stand-alone
            hipLaunchKernelGGL(bn_fold_arrive_kernel<2>, dim3(chunks, cdiv(C, FOLD_COLS), groups), dim3(256), 0, st, (const float*)partial, f);
        } else {
            hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(C, groups), dim3(256), 0, st, (const float*)partial, chunks, C, gsum, gsum + C,
                               (float*)nullptr, (float*)nullptr, ga);
            hipLaunchKernelGGL(bn_sum_groups_kernel, dim3(cdiv(C, 256)), dim3(256), 0, st, (const float*)gsum, groups, C, dgamma, dbeta);
        }
    }
    grid.z = groups;
    launch_bn_bwd_apply_rows<T>(grid, st, dz, lddz, (const T*)nullptr, 0, relu_bits, y, ldy, gamma, coef, coef + C, (const float*)gsum,
                                (const float*)(gsum + C), relu, 1, 1.0f / (float)rows, dy, lddy, dres, lddres, (int)rows, C, lcs, ga);
    return true;
}
}  // namespace up
// dgamma / dbeta: sums over ALL groups (the parameter gradients); every group's data gradient uses its own batch sums
extern "C" int up_bn_bwd_groups_t(const void* dz, int lddz, const uint32_t* relu_bits, const void* y, int ldy, const float* gamma,
                                  const float* coef, int relu, void* dy, int lddy, void* dres, int lddres, float* dgamma,
                                  float* dbeta, float* workspace, size_t workspace_bytes, int64_t rows_per_group, int C, int groups,
                                  int dtype, void* stream) {
    UP_REQUIRE(dz && y && dy && gamma && coef && dgamma && dbeta && workspace && groups > 0 && groups <= 65535 && rows_per_group > 0,
               UP_ERR_INVALID, "bn_bwd_groups: bad argument");
    UP_REQUIRE(dtype == UP_DT_F32 || dtype == UP_DT_BF16, UP_ERR_INVALID, "bn_bwd_groups: dtype %d", dtype);
    UP_REQUIRE(!relu || relu_bits, UP_ERR_INVALID, "bn_bwd_groups: relu needs the sign bits of the forward output");
    UP_REQUIRE(C % 4 == 0 && rows_per_group * (C / 4) < (1ll << 31), UP_ERR_UNSUPPORTED, "bn_bwd_groups: C %% 4 or tensor too large");
    UP_REQUIRE(workspace_bytes >= up_bn_bwd_groups_workspace(rows_per_group, C, groups), UP_ERR_WORKSPACE,
               "bn_bwd_groups: workspace too small");
    UP_REQUIRE(!relu_bits || (rows_per_group * C) % 32 == 0, UP_ERR_UNSUPPORTED,
               "bn_bwd_groups: rows_per_group * C must be a multiple of 32 (a group's ReLU bits start on a word)");
    const size_t es = dtype == UP_DT_BF16 ? 2 : 4;
    hipStream_t st = as_stream(stream);
    const bool done = dtype == UP_DT_BF16
        ? launch_bn_bwd_groups<bf16_t>((const bf16_t*)dz, lddz, relu_bits, (const bf16_t*)y, ldy, gamma, coef, relu, (bf16_t*)dy, lddy,
                                       (bf16_t*)dres, lddres, dgamma, dbeta, workspace, rows_per_group, C, groups, st)
        : launch_bn_bwd_groups<float>((const float*)dz, lddz, relu_bits, (const float*)y, ldy, gamma, coef, relu, (float*)dy, lddy,
                                      (float*)dres, lddres, dgamma, dbeta, workspace, rows_per_group, C, groups, st);
    if (done) return check_launch("bn_bwd_groups");
    // channel counts without a row-strided geometry: the three passes group by group
    if (hipMemsetAsync(dgamma, 0, sizeof(float) * C, st) != hipSuccess || hipMemsetAsync(dbeta, 0, sizeof(float) * C, st) != hipSuccess)
        return check_launch("bn_bwd_groups memset");
    float* gsum = workspace;                 // per-group dgamma | dbeta
    float* ws = workspace + 2 * (size_t)C;
    for (int g = 0; g < groups; ++g) {
        const float* cg = coef + (size_t)g * 4 * C;
        const size_t r0 = (size_t)g * rows_per_group;
        if (int e = up_bn_bwd_acc_t((const char*)dz + r0 * lddz * es, lddz, nullptr, 0, relu_bits ? relu_bits + r0 * C / 32 : nullptr,
                                    (const char*)y + r0 * ldy * es, ldy, gamma, cg, cg + C, relu, 1, (char*)dy + r0 * lddy * es, lddy,
                                    dres ? (char*)dres + r0 * lddres * es : nullptr, lddres, gsum, gsum + C, dgamma, dbeta, ws,
                                    workspace_bytes - 2 * (size_t)C * sizeof(float), rows_per_group, C, dtype, stream))
            return e;
    }
    return UP_OK;
}

// up_bn_bwd_groups_t without its first pass: `partial` = [groups][tiles][C][2] written by the data-gradient launch that produced dz
// (up_conv2d_bwd_data_ex with ep->groups = groups, tiles = up_conv2d_bwd_data_tiles_grouped).  fp32, row-strided geometry only.
extern "C" int up_bn_bwd_groups_prereduced_t(const void* dz, int lddz, const uint32_t* relu_bits, const void* y, int ldy,
                                             const float* gamma, const float* coef, int relu, void* dy, int lddy, void* dres,
                                             int lddres, float* dgamma, float* dbeta, float* workspace, size_t workspace_bytes,
                                             const float* partial, int tiles, int64_t rows_per_group, int C, int groups, int dtype,
                                             void* stream) {
    UP_REQUIRE(dz && y && dy && gamma && coef && dgamma && dbeta && workspace && partial && tiles > 0 && groups > 0 && groups <= BN_MAXG &&
               rows_per_group > 0, UP_ERR_INVALID, "bn_bwd_groups_prereduced: bad argument (1..%d groups)", BN_MAXG);
    UP_REQUIRE(dtype == UP_DT_F32, UP_ERR_UNSUPPORTED, "bn_bwd_groups_prereduced: fp32 only");
    UP_REQUIRE(!relu || relu_bits, UP_ERR_INVALID, "bn_bwd_groups_prereduced: relu needs the sign bits of the forward output");
    UP_REQUIRE(C % 4 == 0 && rows_per_group * (C / 4) < (1ll << 31), UP_ERR_UNSUPPORTED, "bn_bwd_groups_prereduced: C %% 4 or tensor too large");
    UP_REQUIRE(workspace_bytes >= (size_t)groups * 2 * C * sizeof(float), UP_ERR_WORKSPACE, "bn_bwd_groups_prereduced: workspace too small");
    UP_REQUIRE(!relu_bits || (rows_per_group * C) % 32 == 0, UP_ERR_UNSUPPORTED,
               "bn_bwd_groups_prereduced: rows_per_group * C must be a multiple of 32");
    const bool done = launch_bn_bwd_groups<float>((const float*)dz, lddz, relu_bits, (const float*)y, ldy, gamma, coef, relu, (float*)dy,
                                                  lddy, (float*)dres, lddres, dgamma, dbeta, workspace, rows_per_group, C, groups,
                                                  as_stream(stream), partial, tiles);
    UP_REQUIRE(done, UP_ERR_UNSUPPORTED, "bn_bwd_groups_prereduced: no row-strided geometry for C = %d (use up_bn_bwd_groups_t)", C);
    return check_launch("bn_bwd_groups_prereduced");
}
// ... and when that launch also MERGED the sums (up_bn_reduce_slot.gsum / dgamma / dbeta, folded = 1): the apply pass alone.
extern "C" int up_bn_bwd_groups_finalized_t(const void* dz, int lddz, const uint32_t* relu_bits, const void* y, int ldy,
                                            const float* gamma, const float* coef, int relu, void* dy, int lddy, void* dres, int lddres,
                                            const float* gsum, int64_t rows_per_group, int C, int groups, int dtype, void* stream) {
    UP_REQUIRE(dz && y && dy && gamma && coef && gsum && groups > 0 && groups <= BN_MAXG && rows_per_group > 0, UP_ERR_INVALID,
               "bn_bwd_groups_finalized: bad argument (1..%d groups)", BN_MAXG);
    UP_REQUIRE(dtype == UP_DT_F32, UP_ERR_UNSUPPORTED, "bn_bwd_groups_finalized: fp32 only");
    UP_REQUIRE(!relu || relu_bits, UP_ERR_INVALID, "bn_bwd_groups_finalized: relu needs the sign bits of the forward output");
    UP_REQUIRE(C % 4 == 0 && rows_per_group * (C / 4) < (1ll << 31), UP_ERR_UNSUPPORTED, "bn_bwd_groups_finalized: C %% 4 or tensor too large");
    UP_REQUIRE(!relu_bits || (rows_per_group * C) % 32 == 0, UP_ERR_UNSUPPORTED,
               "bn_bwd_groups_finalized: rows_per_group * C must be a multiple of 32");
    const bool done = launch_bn_bwd_groups<float>((const float*)dz, lddz, relu_bits, (const float*)y, ldy, gamma, coef, relu, (float*)dy,
                                                  lddy, (float*)dres, lddres, nullptr, nullptr, nullptr, rows_per_group, C, groups,
                                                  as_stream(stream), nullptr, 0, gsum);
    UP_REQUIRE(done, UP_ERR_UNSUPPORTED, "bn_bwd_groups_finalized: no row-strided geometry for C = %d", C);
    return check_launch("bn_bwd_groups_finalized");
}
// does launch_bn_bwd_groups have a row-strided geometry for this shape (else up_bn_bwd_groups_prereduced_t refuses)?
extern "C" int up_bn_bwd_groups_prereduced_ok(int64_t rows_per_group, int C, int groups, int ld) {
    dim3 grid;
    int lcs = 0;
    return groups >= 1 && groups <= BN_MAXG && ld % 4 == 0 && rows_geometry<float>(rows_per_group, C, grid, lcs) ? 1 : 0;
}

extern "C" int up_relu_bwd(const float* dz, const float* z, float* dx, int64_t n, void* stream) {
    UP_REQUIRE(dz && z && dx && n > 0, UP_ERR_INVALID, "relu_bwd: bad argument");
    hipLaunchKernelGGL(relu_bwd_kernel, dim3(grid_for(n)), dim3(256), 0, as_stream(stream), dz, z, dx, n);
    return check_launch("relu_bwd");
}

extern "C" int up_dropout_fwd_step_t(const void* x, void* y, uint8_t* mask, const float* ext_mask, int64_t n, float p,
                                     uint64_t seed, const uint64_t* step_device, int dtype, void* stream) {
    UP_REQUIRE(x && y && mask && n > 0 && p >= 0.f && p < 1.f, UP_ERR_INVALID, "dropout_fwd: bad argument");
    UP_REQUIRE(dtype == UP_DT_F32 || dtype == UP_DT_BF16, UP_ERR_INVALID, "dropout_fwd: dtype %d", dtype);
    if (dtype == UP_DT_BF16)
        hipLaunchKernelGGL(dropout_fwd_kernel<bf16_t>, dim3(grid_for(n)), dim3(256), 0, as_stream(stream), (const bf16_t*)x,
                           (bf16_t*)y, mask, ext_mask, n, p, 1.0f / (1.0f - p), seed, step_device);
    else
        hipLaunchKernelGGL(dropout_fwd_kernel<float>, dim3(grid_for(n)), dim3(256), 0, as_stream(stream), (const float*)x,
                           (float*)y, mask, ext_mask, n, p, 1.0f / (1.0f - p), seed, step_device);
    return check_launch("dropout_fwd");
}
extern "C" int up_dropout_fwd_t(const void* x, void* y, uint8_t* mask, const float* ext_mask, int64_t n, float p,
                                uint64_t seed, int dtype, void* stream) {
    return up_dropout_fwd_step_t(x, y, mask, ext_mask, n, p, seed, nullptr, dtype, stream);
}
extern "C" int up_dropout_fwd(const float* x, float* y, uint8_t* mask, const float* ext_mask, int64_t n, float p,
                              uint64_t seed, void* stream) {
    return up_dropout_fwd_t(x, y, mask, ext_mask, n, p, seed, UP_DT_F32, stream);
}
extern "C" int up_dropout_bwd_t(const void* dy, const uint8_t* mask, void* dx, int64_t n, float p, int dtype,
                                void* stream) {
    UP_REQUIRE(dy && mask && dx && n > 0 && p >= 0.f && p < 1.f, UP_ERR_INVALID, "dropout_bwd: bad argument");
    UP_REQUIRE(dtype == UP_DT_F32 || dtype == UP_DT_BF16, UP_ERR_INVALID, "dropout_bwd: dtype %d", dtype);
    if (dtype == UP_DT_BF16)
        hipLaunchKernelGGL(dropout_bwd_kernel<bf16_t>, dim3(grid_for(n)), dim3(256), 0, as_stream(stream), (const bf16_t*)dy,
                           mask, (bf16_t*)dx, n, 1.0f / (1.0f - p));
    else
        hipLaunchKernelGGL(dropout_bwd_kernel<float>, dim3(grid_for(n)), dim3(256), 0, as_stream(stream), (const float*)dy,
                           mask, (float*)dx, n, 1.0f / (1.0f - p));
    return check_launch("dropout_bwd");
}
extern "C" int up_dropout_bwd(const float* dy, const uint8_t* mask, float* dx, int64_t n, float p, void* stream) {
    return up_dropout_bwd_t(dy, mask, dx, n, p, UP_DT_F32, stream);
}

extern "C" size_t up_mse_workspace(int64_t) { return MSE_PARTS * sizeof(float); }
extern "C" int up_mse_fwd(const float* y, const float* t, float* loss, float* ws, int64_t n, void* stream) {
    UP_REQUIRE(y && t && loss && ws && n > 0, UP_ERR_INVALID, "mse_fwd: bad argument");
    int parts = grid_for(n, 256, MSE_PARTS);
    hipLaunchKernelGGL(mse_partial_kernel, dim3(parts), dim3(256), 0, as_stream(stream), y, t, ws, n);
    hipLaunchKernelGGL(mse_final_kernel, dim3(1), dim3(256), 0, as_stream(stream), (const float*)ws, parts,
                       1.0f / (float)n, loss);
    return check_launch("mse_fwd");
}
extern "C" int up_mse_bwd(const float* y, const float* t, const float* dloss, float* dy, int64_t n, void* stream) {
    UP_REQUIRE(y && t && dloss && dy && n > 0, UP_ERR_INVALID, "mse_bwd: bad argument");
    hipLaunchKernelGGL(mse_bwd_kernel, dim3(grid_for(n)), dim3(256), 0, as_stream(stream), y, t, dloss, dy, n,
                       2.0f / (float)n);
    return check_launch("mse_bwd");
}
