// Layout conversion at the NCHW boundary, max-pool 3/2/1, bilinear (align_corners) resampling,
// global average pool, AvgPool(9,8,1) of the centre map, strided copies (concat/slice), ConvLSTM
// gate math and the wave-reduced heat-map argmax.  All HBM-bound: 16-byte channel-vector accesses on
// NHWC tensors, one thread per (pixel, 4 channels).
#include "up_common.h"

namespace up {

static inline int grid_cap(int64_t work_items, int cap = 8192) {
    int64_t g = (work_items + 255) / 256;
    if (g < 1) g = 1;
    return (int)(g > cap ? cap : g);
}

#define UP_GRID_STRIDE(i, total) \
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < (total); i += (int64_t)gridDim.x * 256)

// ---- layout ------------------------------------------------------------------------------
template <typename TO>
__global__ void __launch_bounds__(256) nchw_to_nhwc_kernel(const float* x, TO* y, int C, int HW, int ld,
                                                           int64_t npix) {
    UP_GRID_STRIDE(i, npix) {
        int64_t n = i / HW;
        int hw = (int)(i - n * HW);
        const float* src = x + n * C * HW + hw;
        TO* dst = y + i * ld;
        for (int c = 0; c < ld; ++c) st1(dst + c, c < C ? src[(int64_t)c * HW] : 0.f);
    }
}
template <typename TI>
__global__ void __launch_bounds__(256) nhwc_to_nchw_kernel(const TI* x, int ld, float* y, int C, int HW,
                                                           int64_t npix) {
    UP_GRID_STRIDE(i, npix) {
        int64_t n = i / HW;
        int hw = (int)(i - n * HW);
        const TI* src = x + i * ld;
        float* dst = y + n * C * HW + hw;
        for (int c = 0; c < C; ++c) dst[(int64_t)c * HW] = ld1(src + c);
    }
}

// ---- 2-D strided copy / add ----------------------------------------------------------------
__global__ void __launch_bounds__(256) copy2d_v4_kernel(const float* s, int lds, float* d, int ldd, int64_t total,
                                                        int C4, FastDiv fC4) {
    UP_GRID_STRIDE(i, total) {
        uint32_t row = fdiv((uint32_t)i, fC4);
        int c = ((int)i - (int)row * C4) * 4;
        *reinterpret_cast<float4*>(d + (size_t)row * ldd + c) = *reinterpret_cast<const float4*>(s + (size_t)row * lds + c);
    }
}
__global__ void __launch_bounds__(256) copy2d_s_kernel(const float* s, int lds, float* d, int ldd, int64_t total,
                                                       int C, FastDiv fC) {
    UP_GRID_STRIDE(i, total) {
        uint32_t row = fdiv((uint32_t)i, fC);
        int c = (int)i - (int)row * C;
        d[(size_t)row * ldd + c] = s[(size_t)row * lds + c];
    }
}
__global__ void __launch_bounds__(256) add2d_kernel(const float* a, int lda, const float* b, int ldb, float* d,
                                                    int ldd, int64_t total, int C, FastDiv fC) {
    UP_GRID_STRIDE(i, total) {
        uint32_t row = fdiv((uint32_t)i, fC);
        int c = (int)i - (int)row * C;
        d[(size_t)row * ldd + c] = a[(size_t)row * lda + c] + b[(size_t)row * ldb + c];
    }
}

// ---- max-pool 3x3 / stride 2 / pad 1 --------------------------------------------------------
template <typename TI, typename TO>
__global__ void __launch_bounds__(256) maxpool_fwd_kernel(const TI* x, int ldx, TO* y, int ldy, uint8_t* idx,
                                                          int H, int W, int C4, int P, int Q, int64_t total,
                                                          FastDiv fC4, FastDiv fQ, FastDiv fP) {
    UP_GRID_STRIDE(i, total) {
        uint32_t pix = fdiv((uint32_t)i, fC4);
        int c = ((int)i - (int)pix * C4) * 4;
        uint32_t t = fdiv(pix, fQ);
        int q = (int)pix - (int)t * Q;
        uint32_t n = fdiv(t, fP);
        int p = (int)t - (int)n * P;
        // The first in-bounds window element is (r0, s0); starting every channel from its code with best = -inf makes the
        // scan branch-free ("first" flag gone): a later element replaces it iff it is larger or NaN, exactly F.max_pool2d's
        // rule (first maximum in scan order wins).  (With a `first` flag in the loop hipcc 7.2 mis-compiled the
        // <bf16, bf16> instantiation — stale code for element 0, values right — tools/gpu/dbg_maxpool.py.)
        const int r0 = 2 * p - 1 < 0 ? 1 : 0, s0 = 2 * q - 1 < 0 ? 1 : 0;
        float best[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
        int bi[4] = {r0 * 3 + s0, r0 * 3 + s0, r0 * 3 + s0, r0 * 3 + s0};
        for (int r = r0; r < 3; ++r) {
            int h = 2 * p - 1 + r;
            if (h >= H) break;
            for (int s = s0; s < 3; ++s) {
                int w = 2 * q - 1 + s;
                if (w >= W) break;
                float4 v4 = ld4<TI>(x + ((size_t)(n * H + h) * W + w) * ldx + c);
                float v[4] = {v4.x, v4.y, v4.z, v4.w};
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (v[e] > best[e] || v[e] != v[e]) {
                        best[e] = v[e];
                        bi[e] = r * 3 + s;
                    }
            }
        }
        st4(y + (size_t)pix * ldy + c, make_float4(best[0], best[1], best[2], best[3]));
        // one 32-bit store of the four window codes
        *reinterpret_cast<uint32_t*>(idx + (size_t)pix * (C4 * 4) + c) =
            (uint32_t)bi[0] | ((uint32_t)bi[1] << 8) | ((uint32_t)bi[2] << 16) | ((uint32_t)bi[3] << 24);
    }
}
template <typename TI, typename TO>
__global__ void __launch_bounds__(256) maxpool_bwd_kernel(const TI* dy, int lddy, const uint8_t* idx, TO* dx,
                                                          int lddx, int H, int W, int C4, int P, int Q,
                                                          int64_t total, FastDiv fC4, FastDiv fW, FastDiv fH) {
    UP_GRID_STRIDE(i, total) {
        uint32_t pix = fdiv((uint32_t)i, fC4);
        int c = ((int)i - (int)pix * C4) * 4;
        uint32_t t = fdiv(pix, fW);
        int w = (int)pix - (int)t * W;
        uint32_t n = fdiv(t, fH);
        int h = (int)t - (int)n * H;
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        for (int r = 0; r < 3; ++r) {
            int tp = h + 1 - r;
            if (tp < 0 || (tp & 1) || (tp >> 1) >= P) continue;
            int p = tp >> 1;
            for (int s = 0; s < 3; ++s) {
                int tq = w + 1 - s;
                if (tq < 0 || (tq & 1) || (tq >> 1) >= Q) continue;
                int q = tq >> 1;
                size_t op = (size_t)(n * P + p) * Q + q;
                const uint8_t* ip = idx + op * (C4 * 4) + c;
                float4 g = ld4<TI>(dy + op * lddy + c);
                int code = r * 3 + s;
                if (ip[0] == code) acc[0] += g.x;
                if (ip[1] == code) acc[1] += g.y;
                if (ip[2] == code) acc[2] += g.z;
                if (ip[3] == code) acc[3] += g.w;
            }
        }
        st4(dx + (size_t)pix * lddx + c, make_float4(acc[0], acc[1], acc[2], acc[3]));
    }
}

// ---- bilinear, align_corners=True ----------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(256) bilinear_fwd_kernel(const T* x, int ldx, T* y, int ldy, int H, int W,
                                                           int C4, int P, int Q, float sh, float sw, int64_t total,
                                                           FastDiv fC4, FastDiv fQ, FastDiv fP) {
    UP_GRID_STRIDE(i, total) {
        uint32_t pix = fdiv((uint32_t)i, fC4);
        int c = ((int)i - (int)pix * C4) * 4;
        uint32_t t = fdiv(pix, fQ);
        int q = (int)pix - (int)t * Q;
        uint32_t n = fdiv(t, fP);
        int p = (int)t - (int)n * P;
        float fh = sh * p, fw = sw * q;
        int h0 = (int)fh, w0 = (int)fw;
        int h1 = h0 < H - 1 ? h0 + 1 : h0, w1 = w0 < W - 1 ? w0 + 1 : w0;
        float lh1 = fh - h0, lw1 = fw - w0, lh0 = 1.f - lh1, lw0 = 1.f - lw1;
        const T* b = x + (size_t)n * H * W * ldx + c;
        float4 a00 = ld4<T>(b + (size_t)(h0 * W + w0) * ldx);
        float4 a01 = ld4<T>(b + (size_t)(h0 * W + w1) * ldx);
        float4 a10 = ld4<T>(b + (size_t)(h1 * W + w0) * ldx);
        float4 a11 = ld4<T>(b + (size_t)(h1 * W + w1) * ldx);
        float4 o;
        o.x = lh0 * (lw0 * a00.x + lw1 * a01.x) + lh1 * (lw0 * a10.x + lw1 * a11.x);
        o.y = lh0 * (lw0 * a00.y + lw1 * a01.y) + lh1 * (lw0 * a10.y + lw1 * a11.y);
        o.z = lh0 * (lw0 * a00.z + lw1 * a01.z) + lh1 * (lw0 * a10.z + lw1 * a11.z);
        o.w = lh0 * (lw0 * a00.w + lw1 * a01.w) + lh1 * (lw0 * a10.w + lw1 * a11.w);
        st4(y + (size_t)pix * ldy + c, o);
    }
}
// weight with which output coordinate o (of `out` samples, scale s) reads input sample `in_i`
__device__ __forceinline__ float bil_weight(int o, int in_i, int in_n, float s) {
    float f = s * o;
    int i0 = (int)f;
    int i1 = i0 < in_n - 1 ? i0 + 1 : i0;
    float l1 = f - i0;
    float w = 0.f;
    if (i0 == in_i) w += 1.f - l1;
    if (i1 == in_i) w += l1;
    return w;
}
template <typename T>
__global__ void __launch_bounds__(256) bilinear_bwd_kernel(const T* dy, int lddy, T* dx, int lddx, int H,
                                                           int W, int C4, int P, int Q, float sh, float sw,
                                                           int64_t total, FastDiv fC4, FastDiv fW, FastDiv fH) {
    UP_GRID_STRIDE(i, total) {
        uint32_t pix = fdiv((uint32_t)i, fC4);
        int c = ((int)i - (int)pix * C4) * 4;
        uint32_t t = fdiv(pix, fW);
        int w = (int)pix - (int)t * W;
        uint32_t n = fdiv(t, fH);
        int h = (int)t - (int)n * H;
        int p_lo = 0, p_hi = P - 1, q_lo = 0, q_hi = Q - 1;
        if (sh > 0.f) {
            p_lo = max(0, (int)floorf((h - 1) / sh) - 1);
            p_hi = min(P - 1, (int)ceilf((h + 1) / sh) + 1);
        }
        if (sw > 0.f) {
            q_lo = max(0, (int)floorf((w - 1) / sw) - 1);
            q_hi = min(Q - 1, (int)ceilf((w + 1) / sw) + 1);
        }
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        for (int p = p_lo; p <= p_hi; ++p) {
            float wh = bil_weight(p, h, H, sh);
            if (wh == 0.f) continue;
            for (int q = q_lo; q <= q_hi; ++q) {
                float ww = bil_weight(q, w, W, sw);
                if (ww == 0.f) continue;
                float4 g = ld4<T>(dy + ((size_t)(n * P + p) * Q + q) * lddy + c);
                float k = wh * ww;
                acc[0] += k * g.x;
                acc[1] += k * g.y;
                acc[2] += k * g.z;
                acc[3] += k * g.w;
            }
        }
        st4(dx + (size_t)pix * lddx + c, make_float4(acc[0], acc[1], acc[2], acc[3]));
    }
}

// Backward of the broadcast of a 1x1 map (the WASP global-pool branch is up-sampled from 1x1 to the feature-map size,
// wasp.py:83): dx[n][c] = sum over all P*Q output pixels of dy.  The generic gather above would give ONE thread the whole
// P*Q loop (868 us at 46x46); here a workgroup owns (image, 64 channels): 16 row lanes x 16 four-channel groups, four rows in
// flight per thread, LDS tree over the row lanes (fixed order: deterministic).
template <typename T>
__global__ void __launch_bounds__(256) bcast_bwd_kernel(const T* dy, int lddy, T* dx, int lddx, int PQ, int C) {
    __shared__ float red[16][64];
    const int n = blockIdx.x;
    const int cq = threadIdx.x & 15, rl = threadIdx.x >> 4;
    const int c = blockIdx.y * 64 + cq * 4;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    if (c < C) {
        const T* base = dy + (size_t)n * PQ * lddy + c;
        int r = rl;
        for (; r + 48 < PQ; r += 64) {
            const float4 a = ld4<T>(base + (size_t)r * lddy), b = ld4<T>(base + (size_t)(r + 16) * lddy);
            const float4 e = ld4<T>(base + (size_t)(r + 32) * lddy), f = ld4<T>(base + (size_t)(r + 48) * lddy);
            s.x += (a.x + b.x) + (e.x + f.x);
            s.y += (a.y + b.y) + (e.y + f.y);
            s.z += (a.z + b.z) + (e.z + f.z);
            s.w += (a.w + b.w) + (e.w + f.w);
        }
        for (; r < PQ; r += 16) {
            const float4 a = ld4<T>(base + (size_t)r * lddy);
            s.x += a.x;
            s.y += a.y;
            s.z += a.z;
            s.w += a.w;
        }
    }
    red[rl][cq * 4 + 0] = s.x;
    red[rl][cq * 4 + 1] = s.y;
    red[rl][cq * 4 + 2] = s.z;
    red[rl][cq * 4 + 3] = s.w;
    __syncthreads();
    if (threadIdx.x < 16 && c < C) {
        float t[4] = {0.f, 0.f, 0.f, 0.f};
        for (int k = 0; k < 16; ++k)
            for (int e = 0; e < 4; ++e) t[e] += red[k][cq * 4 + e];
        st4(dx + (size_t)n * lddx + c, make_float4(t[0], t[1], t[2], t[3]));
    }
}

// ---- global average pool ---------------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(256) gap_fwd_kernel(const T* x, int ldx, T* y, int HW, int C) {
    __shared__ float red[256];
    int n = blockIdx.x;
    int c = blockIdx.y * 64 + (threadIdx.x & 63);
    int rl = threadIdx.x >> 6;
    float s = 0.f;
    if (c < C)
        for (int r = rl; r < HW; r += 4) s += ld1(x + ((size_t)n * HW + r) * ldx + c);
    red[threadIdx.x] = s;
    __syncthreads();
    if (rl == 0 && c < C) {
        int t = threadIdx.x;
        st1(y + (size_t)n * C + c, (red[t] + red[t + 64] + red[t + 128] + red[t + 192]) / (float)HW);
    }
}
template <typename T>
__global__ void __launch_bounds__(256) gap_bwd_kernel(const T* dy, T* dx, int lddx, int HW, int C4,
                                                      float inv, int64_t total, FastDiv fC4, FastDiv fHW) {
    UP_GRID_STRIDE(i, total) {
        uint32_t pix = fdiv((uint32_t)i, fC4);
        int c = ((int)i - (int)pix * C4) * 4;
        uint32_t n = fdiv(pix, fHW);
        float4 g = ld4<T>(dy + (size_t)n * (C4 * 4) + c);
        st4(dx + (size_t)pix * lddx + c, make_float4(g.x * inv, g.y * inv, g.z * inv, g.w * inv));
    }
}

// ---- AvgPool2d(9, 8, 1), count_include_pad -----------------------------------------------------
__global__ void __launch_bounds__(256) avgpool9s8_kernel(const float* x, float* y, int ldy, int coff, int H, int W,
                                                         int P, int Q, int64_t total) {
    UP_GRID_STRIDE(i, total) {
        int q = (int)(i % Q);
        int64_t t = i / Q;
        int p = (int)(t % P);
        int64_t n = t / P;
        int hs = p * 8 - 1, ws = q * 8 - 1;
        int he = min(hs + 9, H + 1), we = min(ws + 9, W + 1);
        float div = (float)((he - hs) * (we - ws));
        hs = max(hs, 0);
        ws = max(ws, 0);
        he = min(he, H);
        we = min(we, W);
        float s = 0.f;
        for (int h = hs; h < he; ++h)
            for (int w = ws; w < we; ++w) s += x[(n * H + h) * W + w];
        y[i * ldy + coff] = s / div;
    }
}

// ---- ConvLSTM gates -------------------------------------------------------------------------
__device__ __forceinline__ float sigm(float v) { return 1.0f / (1.0f + expf(-v)); }

__global__ void __launch_bounds__(256) lstm0_fwd_kernel(const float* G, int ldg, float* cell, float* hide, int ldo,
                                                        int64_t total, int Cg) {
    UP_GRID_STRIDE(i, total) {
        int64_t r = i / Cg;
        int c = (int)(i - r * Cg);
        const float* g = G + r * ldg;
        float gg = tanhf(g[c]), ii = sigm(g[Cg + c]), oo = sigm(g[2 * Cg + c]);
        float cl = tanhf(gg * ii);
        cell[r * ldo + c] = cl;
        hide[r * ldo + c] = oo * cl;
    }
}
__global__ void __launch_bounds__(256) lstm0_bwd_kernel(const float* G, int ldg, const float* dcell,
                                                        const float* dhide, int ldo, float* dG, int64_t total,
                                                        int Cg) {
    UP_GRID_STRIDE(i, total) {
        int64_t r = i / Cg;
        int c = (int)(i - r * Cg);
        const float* g = G + r * ldg;
        float gg = tanhf(g[c]), ii = sigm(g[Cg + c]), oo = sigm(g[2 * Cg + c]);
        float cl = tanhf(gg * ii);
        float dh = dhide[r * ldo + c];
        float dc = dcell[r * ldo + c] + dh * oo;
        float dgi = dc * (1.f - cl * cl);
        float* o = dG + r * ldg;
        o[c] = dgi * ii * (1.f - gg * gg);
        o[Cg + c] = dgi * gg * ii * (1.f - ii);
        o[2 * Cg + c] = dh * cl * oo * (1.f - oo);
    }
}
__global__ void __launch_bounds__(256) lstm_fwd_kernel(const float* G, int ldg, const float* cprev, int ldc,
                                                       float* cell, float* hide, int ldo, int64_t total, int Cg) {
    UP_GRID_STRIDE(i, total) {
        int64_t r = i / Cg;
        int c = (int)(i - r * Cg);
        const float* g = G + r * ldg;
        float gg = tanhf(g[c]), ii = sigm(g[Cg + c]), oo = sigm(g[2 * Cg + c]), ff = sigm(g[3 * Cg + c]);
        float cl = ff * cprev[r * ldc + c] + ii * gg;
        cell[r * ldo + c] = cl;
        hide[r * ldo + c] = oo * tanhf(cl);
    }
}
__global__ void __launch_bounds__(256) lstm_bwd_kernel(const float* G, int ldg, const float* cprev, int ldc,
                                                       const float* cell, const float* dcell, const float* dhide,
                                                       int ldo, float* dG, float* dcprev, int64_t total, int Cg) {
    UP_GRID_STRIDE(i, total) {
        int64_t r = i / Cg;
        int c = (int)(i - r * Cg);
        const float* g = G + r * ldg;
        float gg = tanhf(g[c]), ii = sigm(g[Cg + c]), oo = sigm(g[2 * Cg + c]), ff = sigm(g[3 * Cg + c]);
        float tc = tanhf(cell[r * ldo + c]);
        float dh = dhide[r * ldo + c];
        float dc = dcell[r * ldo + c] + dh * oo * (1.f - tc * tc);
        float* o = dG + r * ldg;
        o[c] = dc * ii * (1.f - gg * gg);
        o[Cg + c] = dc * gg * ii * (1.f - ii);
        o[2 * Cg + c] = dh * tc * oo * (1.f - oo);
        o[3 * Cg + c] = dc * cprev[r * ldc + c] * ff * (1.f - ff);
        dcprev[r * ldo + c] = dc * ff;
    }
}

// ---- heat-map argmax: one wavefront per (b, j) map ------------------------------------------------
__device__ __forceinline__ bool am_better(float v, int i, float bv, int bi) {
    bool vn = v != v, bn = bv != bv;
    if (vn != bn) return vn;        // NaN beats a number (np.argmax semantics)
    if (vn) return i < bi;          // both NaN: first one
    if (v != bv) return v > bv;
    return i < bi;                  // ties: lowest flat index
}
__global__ void __launch_bounds__(256) argmax_kernel(const float* hm, int maps, int HW, int W, int32_t* idx,
                                                     float* preds, float* maxvals) {
    int map = blockIdx.x * 4 + (threadIdx.x >> 6);
    int lane = threadIdx.x & 63;
    if (map >= maps) return;
    const float* p = hm + (size_t)map * HW;
    float bv = -INFINITY;
    int bi = 0x7fffffff;
    for (int i = lane; i < HW; i += 64) {
        float v = p[i];
        if (am_better(v, i, bv, bi)) {
            bv = v;
            bi = i;
        }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        float ov = __shfl_down(bv, off);
        int oi = __shfl_down(bi, off);
        if (am_better(ov, oi, bv, bi)) {
            bv = ov;
            bi = oi;
        }
    }
    if (lane == 0) {
        if (idx) idx[map] = bi;
        float keep = bv > 0.f ? 1.f : 0.f;
        preds[map * 2] = (float)(bi % W) * keep;
        preds[map * 2 + 1] = (float)(bi / W) * keep;
        maxvals[map] = bv;
    }
}

// ---- multi-person decode (the step after the path for the box head; utils/uniPose.py:14-200) -----------------------
// Peaks of a map as the reference finds them: negatives clamped to 0, then `maximum_filter(3x3) == map` XOR the eroded
// zero background (scipy, borders reflected / counted as background) — which is exactly: value > 0 and >= each of its
// in-bounds 8 neighbours (plateaus mark every member).
__global__ void __launch_bounds__(256) peak_mask_kernel(const float* maps, long long total, int H, int W, uint8_t* mask) {
    long long e = (long long)blockIdx.x * 256 + threadIdx.x;
    if (e >= total) return;
    const int w = (int)(e % W);
    const int h = (int)((e / W) % H);
    const float v = maps[e];
    bool peak = v > 0.f;
    for (int dh = -1; dh <= 1; ++dh)
        for (int dw = -1; dw <= 1; ++dw) {
            const int hh = h + dh, ww = w + dw;
            if (hh < 0 || ww < 0 || hh >= H || ww >= W) continue;
            peak = peak && v >= maps[e + (long long)dh * W + dw];
        }
    mask[e] = peak ? 1 : 0;
}
// argmax of channels ch0 .. ch0+nch-1 inside one box per person: one wavefront per (person, channel); position
// relative to the box, first maximum in the row-major order of the BOX (np.argmax of the sliced array)
__global__ void __launch_bounds__(256) box_argmax_kernel(const float* maps, int H, int W, const int32_t* boxes, int P,
                                                         int ch0, int nch, int32_t* out_hw) {
    const int job = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (job >= P * nch) return;
    const int p = job / nch, c = job - p * nch;
    const int r0 = boxes[p * 4], r1 = boxes[p * 4 + 1], c0 = boxes[p * 4 + 2], c1 = boxes[p * 4 + 3];
    const int bw = c1 - c0, n = (r1 - r0) * bw;
    const float* m = maps + (size_t)(ch0 + c) * H * W;
    float bv = -INFINITY;
    int bi = 0x7fffffff;
    for (int i = lane; i < n; i += 64) {
        const int rr = i / bw, cc = i - rr * bw;
        const float v = m[(size_t)(r0 + rr) * W + c0 + cc];
        if (am_better(v, i, bv, bi)) {
            bv = v;
            bi = i;
        }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        float ov = __shfl_down(bv, off);
        int oi = __shfl_down(bi, off);
        if (am_better(ov, oi, bv, bi)) {
            bv = ov;
            bi = oi;
        }
    }
    if (lane == 0) {
        if (bi == 0x7fffffff) bi = 0;   // a box of -inf only: np.argmax returns 0
        out_hw[job * 2] = bi / bw;
        out_hw[job * 2 + 1] = bi - (bi / bw) * bw;
    }
}

// ---- training targets (the step before the path: lsp_lspet_data.py:224-245, mpii_data.py:165-187) ----------------
// Gaussian joint maps exactly as the loaders build them: float64 exp(-D2 / 2.0 / sigma / sigma) on the integer pixel
// grid (utils/utils.py:200-203), clipped to <= 1, values < 0.0099 set to 0, then stored as float32; channel 0 is
// 1 - max over the joint channels (float32).  The joint centre is int(coordinate) / stride like the loaders compute it.
__device__ __forceinline__ float target_gauss(double x, double y, double cx, double cy, double sigma) {
    const double dx = x - cx, dy = y - cy;
    const double d2 = __dadd_rn(__dmul_rn(dx, dx), __dmul_rn(dy, dy));
    double v = exp(-d2 / 2.0 / sigma / sigma);
    if (v > 1.0) v = 1.0;
    if (v < 0.0099) v = 0.0;
    return (float)v;
}
__global__ void __launch_bounds__(256) heatmap_target_kernel(const double* kpt, int K, int H, int W, double stride,
                                                             double sigma, float* out, long long total /* B*H*W */) {
    long long e = (long long)blockIdx.x * 256 + threadIdx.x;
    if (e >= total) return;
    const int x = (int)(e % W);
    const long long t = e / W;
    const int y = (int)(t % H);
    const int b = (int)(t / H);
    float* o = out + (size_t)b * (K + 1) * H * W + (size_t)y * W + x;
    float mx = -INFINITY;
    for (int k = 0; k < K; ++k) {
        const double cx = (double)(long long)kpt[((size_t)b * K + k) * 2] * 1.0 / stride;       // int(kpt) * 1.0 / stride
        const double cy = (double)(long long)kpt[((size_t)b * K + k) * 2 + 1] * 1.0 / stride;
        const float g = target_gauss((double)x, (double)y, cx, cy, sigma);
        o[(size_t)(k + 1) * H * W] = g;
        mx = fmaxf(mx, g);
    }
    o[0] = 1.0f - mx;
}
__global__ void __launch_bounds__(256) gaussian_map_kernel(const double* center, int H, int W, double sigma, float* out,
                                                           long long total /* N*H*W */) {
    long long e = (long long)blockIdx.x * 256 + threadIdx.x;
    if (e >= total) return;
    const int x = (int)(e % W);
    const long long t = e / W;
    const int y = (int)(t % H);
    const int n = (int)(t / H);
    out[e] = target_gauss((double)x, (double)y, center[2 * n], center[2 * n + 1], sigma);
}
// (pixel - mean) / std, HWC -> CHW (Mytransforms.to_tensor + normalize, Mytransforms.py:10-41)
__global__ void __launch_bounds__(256) normalize_image_kernel(const float* img, int C, int HW, float mean, float stdv,
                                                              float* out, long long total /* B*HW*C */) {
    long long e = (long long)blockIdx.x * 256 + threadIdx.x;
    if (e >= total) return;
    const int c = (int)(e % C);
    const long long t = e / C;
    const int p = (int)(t % HW);
    const long long b = t / HW;
    out[((size_t)b * C + c) * HW + p] = (img[e] - mean) / stdv;
}

// ---- PCK / PCKh accuracy on joint coordinates (utils/evaluate.py:5-29,58-172) -------------------------------------
// One workgroup; thread j owns joint j.  Arithmetic types follow the reference under NumPy >= 2: coordinates float32,
// head / torso sizes float32 (np.linalg.norm of float32), thresholds float32 products (python float * np.float32),
// normalised distances float64 (float32 coordinates divided by the float64 [H/10, W/10]).
__device__ __forceinline__ float pck_norm2(float x, float y) { return sqrtf(x * x + y * y); }
__device__ void pck_scales(const float* t0 /* (J,2) of sample 0 */, int dataset, float& head, float& torso) {
    auto X = [&](int j) { return t0[2 * j]; };
    auto Y = [&](int j) { return t0[2 * j + 1]; };
    auto midx = [&](int a, int b) { return (X(a) + X(b)) / 2.f; };
    auto midy = [&](int a, int b) { return (Y(a) + Y(b)) / 2.f; };
    switch (dataset) {
        case UP_DS_LSP:
            head = pck_norm2(X(14) - X(13), Y(14) - Y(13));
            torso = pck_norm2(X(13) - midx(3, 4), Y(13) - midy(3, 4));
            break;
        case UP_DS_COCO:
            head = pck_norm2(X(4) - X(5), Y(4) - Y(5));
            torso = pck_norm2(X(13) - midx(12, 13), Y(13) - midy(12, 13));
            break;
        case UP_DS_PENN_ACTION:
            head = pck_norm2(X(0) - midx(1, 2), Y(0) - midy(1, 2));
            torso = pck_norm2(midx(1, 2) - midx(7, 8), midy(1, 2) - midy(7, 8));
            break;
        case UP_DS_NTID:
            head = 2.f * pck_norm2(X(4) - X(3), Y(4) - Y(3));
            torso = pck_norm2(X(3) - X(1), Y(3) - Y(1));
            break;
        case UP_DS_POSETRACK:
            head = 2.f * pck_norm2(X(1) - X(2), Y(1) - Y(2));
            torso = pck_norm2(midx(12, 13) - midx(6, 7), midy(12, 13) - midy(6, 7));
            break;
        case UP_DS_BBC:   // the reference subtracts the neck POINT from the x coordinate of joint 1 (evaluate.py:147-149)
            head = pck_norm2(X(1) - midx(6, 7), Y(1) - midy(6, 7));
            torso = pck_norm2(3.f * (X(1) - midx(6, 7)), 3.f * (X(1) - midy(6, 7)));
            break;
        default:          // UP_DS_MPII
            head = pck_norm2(X(9) - X(10), Y(9) - Y(10));
            torso = fabsf(X(7) - X(8));
            break;
    }
}
__global__ void __launch_bounds__(256) pck_kernel(const float* pred, const float* tgt, int B, int J, int H, int W,
                                                  int dataset, float thr_pck, float thr_pckh, double* acc, double* pck,
                                                  double* pckh, double* visible, int32_t* cnt) {
    __shared__ double s_acc[256], s_pck[256], s_pckh[256];
    const int j = threadIdx.x;
    float head, torso;
    pck_scales(tgt, dataset, head, torso);
    const double t_h = (double)(thr_pckh * head), t_k = (double)(thr_pck * torso);   // float32 products
    const double nx = (double)H / 10.0, ny = (double)W / 10.0;   // x is divided by H/10, y by W/10 (evaluate.py:68-70)
    if (j < J) {
        int valid = 0, c0 = 0, ch = 0, ck = 0;
        for (int n = 0; n < B; ++n) {
            const float tx = tgt[(n * J + j) * 2], ty = tgt[(n * J + j) * 2 + 1];
            if (tx > 1.f && ty > 1.f) {
                const double dx = (double)pred[(n * J + j) * 2] / nx - (double)tx / nx;
                const double dy = (double)pred[(n * J + j) * 2 + 1] / ny - (double)ty / ny;
                const double d = sqrt(__dadd_rn(__dmul_rn(dx, dx), __dmul_rn(dy, dy)));   // no FMA contraction
                ++valid;
                c0 += d < 0.5;
                ch += d < t_h;
                ck += d < t_k;
            }
        }
        s_acc[j] = valid ? c0 * 1.0 / valid : -1.0;
        s_pckh[j] = valid ? ch * 1.0 / valid : -1.0;
        s_pck[j] = valid ? ck * 1.0 / valid : -1.0;
    }
    __syncthreads();
    if (j == 0) {   // the averages, summed in joint order like the reference's loops
        int c = 0;
        double sa = 0.0, sh = 0.0, sk = 0.0;
        for (int i = 0; i < J; ++i) {
            const bool vis = s_acc[i] >= 0.0;
            visible[i] = vis ? 1.0 : 0.0;
            if (vis) {
                sa += s_acc[i];
                ++c;
            }
            acc[i] = vis ? s_acc[i] : 0.0;
            if (s_pckh[i] >= 0.0) sh += s_pckh[i];
            pckh[i] = s_pckh[i] >= 0.0 ? s_pckh[i] : 0.0;
            if (s_pck[i] >= 0.0) sk += s_pck[i];
            pck[i] = s_pck[i] >= 0.0 ? s_pck[i] : 0.0;
        }
        if (c) {
            acc[0] = sa / c;
            pckh[0] = sh / c;
            pck[0] = sk / c;
        }
        *cnt = c;
    }
}

}  // namespace up

using namespace up;

#define UP_LAUNCH_1D(kernel, total, st, ...) \
    hipLaunchKernelGGL(kernel, dim3(grid_cap(total)), dim3(256), 0, st, __VA_ARGS__)

#define UP_DT_OK(dt) ((dt) == UP_DT_F32 || (dt) == UP_DT_BF16)
extern "C" int up_nchw_to_nhwc_t(const float* x, void* y, int N, int C, int H, int W, int ldy, int dt_out, void* stream) {
    UP_REQUIRE(x && y && N > 0 && C > 0 && H > 0 && W > 0 && ldy >= C && UP_DT_OK(dt_out), UP_ERR_INVALID,
               "nchw_to_nhwc: bad argument");
    int64_t npix = (int64_t)N * H * W;
    if (dt_out == UP_DT_BF16)
        UP_LAUNCH_1D(nchw_to_nhwc_kernel<bf16_t>, npix, as_stream(stream), x, (bf16_t*)y, C, H * W, ldy, npix);
    else
        UP_LAUNCH_1D(nchw_to_nhwc_kernel<float>, npix, as_stream(stream), x, (float*)y, C, H * W, ldy, npix);
    return check_launch("nchw_to_nhwc");
}
extern "C" int up_nchw_to_nhwc(const float* x, float* y, int N, int C, int H, int W, int ldy, void* stream) {
    return up_nchw_to_nhwc_t(x, y, N, C, H, W, ldy, UP_DT_F32, stream);
}
extern "C" int up_nhwc_to_nchw_t(const void* x, int ldx, float* y, int N, int C, int H, int W, int dt_in, void* stream) {
    UP_REQUIRE(x && y && N > 0 && C > 0 && H > 0 && W > 0 && ldx >= C && UP_DT_OK(dt_in), UP_ERR_INVALID,
               "nhwc_to_nchw: bad argument");
    int64_t npix = (int64_t)N * H * W;
    if (dt_in == UP_DT_BF16)
        UP_LAUNCH_1D(nhwc_to_nchw_kernel<bf16_t>, npix, as_stream(stream), (const bf16_t*)x, ldx, y, C, H * W, npix);
    else
        UP_LAUNCH_1D(nhwc_to_nchw_kernel<float>, npix, as_stream(stream), (const float*)x, ldx, y, C, H * W, npix);
    return check_launch("nhwc_to_nchw");
}
extern "C" int up_nhwc_to_nchw(const float* x, int ldx, float* y, int N, int C, int H, int W, void* stream) {
    return up_nhwc_to_nchw_t(x, ldx, y, N, C, H, W, UP_DT_F32, stream);
}

extern "C" int up_copy2d(const float* s, int lds, float* d, int ldd, int64_t rows, int C, void* stream) {
    UP_REQUIRE(s && d && rows > 0 && C > 0 && lds >= C && ldd >= C, UP_ERR_INVALID, "copy2d: bad argument");
    UP_REQUIRE(rows * C < (1ll << 31), UP_ERR_UNSUPPORTED, "copy2d: tensor too large");
    bool v4 = C % 4 == 0 && lds % 4 == 0 && ldd % 4 == 0 && ((uintptr_t)s % 16 == 0) && ((uintptr_t)d % 16 == 0);
    if (v4) {
        int64_t total = rows * (C / 4);
        UP_LAUNCH_1D(copy2d_v4_kernel, total, as_stream(stream), s, lds, d, ldd, total, C / 4, make_fastdiv(C / 4));
    } else {
        int64_t total = rows * C;
        UP_LAUNCH_1D(copy2d_s_kernel, total, as_stream(stream), s, lds, d, ldd, total, C, make_fastdiv(C));
    }
    return check_launch("copy2d");
}
extern "C" int up_add2d(const float* a, int lda, const float* b, int ldb, float* d, int ldd, int64_t rows, int C,
                        void* stream) {
    UP_REQUIRE(a && b && d && rows > 0 && C > 0, UP_ERR_INVALID, "add2d: bad argument");
    UP_REQUIRE(rows * C < (1ll << 31), UP_ERR_UNSUPPORTED, "add2d: tensor too large");
    int64_t total = rows * C;
    UP_LAUNCH_1D(add2d_kernel, total, as_stream(stream), a, lda, b, ldb, d, ldd, total, C, make_fastdiv(C));
    return check_launch("add2d");
}

static int pool_args_ok(int N, int H, int W, int C, int P, int Q, int lda, int ldb) {
    UP_REQUIRE(N > 0 && H > 0 && W > 0 && C > 0 && P > 0 && Q > 0, UP_ERR_INVALID, "spatial op: bad dimension");
    UP_REQUIRE(C % 4 == 0 && lda % 4 == 0 && ldb % 4 == 0 && lda >= C && ldb >= C, UP_ERR_INVALID,
               "spatial op: C and strides must be multiples of 4");
    UP_REQUIRE((int64_t)N * H * W * (C / 4) < (1ll << 31) && (int64_t)N * P * Q * (C / 4) < (1ll << 31),
               UP_ERR_UNSUPPORTED, "spatial op: tensor too large");
    return UP_OK;
}

// (dt_in, dt_out): the max-pool after the stem is where the bf16-storage network leaves fp32 (fp32 in, bf16 out; its
// backward bf16 in, fp32 out); all other uses have equal types
namespace up {
template <typename TI, typename TO>
static void launch_maxpool_fwd(const void* x, int ldx, void* y, int ldy, uint8_t* idx, int H, int W, int C, int P, int Q,
                               int64_t total, hipStream_t st) {
    UP_LAUNCH_1D((maxpool_fwd_kernel<TI, TO>), total, st, (const TI*)x, ldx, (TO*)y, ldy, idx, H, W, C / 4, P, Q, total,
                 make_fastdiv(C / 4), make_fastdiv(Q), make_fastdiv(P));
}
template <typename TI, typename TO>
static void launch_maxpool_bwd(const void* dy, int lddy, const uint8_t* idx, void* dx, int lddx, int H, int W, int C, int P,
                               int Q, int64_t total, hipStream_t st) {
    UP_LAUNCH_1D((maxpool_bwd_kernel<TI, TO>), total, st, (const TI*)dy, lddy, idx, (TO*)dx, lddx, H, W, C / 4, P, Q, total,
                 make_fastdiv(C / 4), make_fastdiv(W), make_fastdiv(H));
}
}  // namespace up
extern "C" int up_maxpool3s2_fwd_t(const void* x, int ldx, void* y, int ldy, uint8_t* idx, int N, int H, int W, int C,
                                   int P, int Q, int dt_in, int dt_out, void* stream) {
    if (int e = pool_args_ok(N, H, W, C, P, Q, ldx, ldy)) return e;
    UP_REQUIRE(x && y && idx && UP_DT_OK(dt_in) && UP_DT_OK(dt_out), UP_ERR_INVALID, "maxpool_fwd: bad argument");
    UP_REQUIRE(P == (H - 1) / 2 + 1 && Q == (W - 1) / 2 + 1, UP_ERR_INVALID, "maxpool_fwd: P,Q mismatch");
    int64_t total = (int64_t)N * P * Q * (C / 4);
    hipStream_t st = as_stream(stream);
    if (dt_in == UP_DT_F32 && dt_out == UP_DT_F32) launch_maxpool_fwd<float, float>(x, ldx, y, ldy, idx, H, W, C, P, Q, total, st);
    else if (dt_in == UP_DT_F32) launch_maxpool_fwd<float, bf16_t>(x, ldx, y, ldy, idx, H, W, C, P, Q, total, st);
    else if (dt_out == UP_DT_F32) launch_maxpool_fwd<bf16_t, float>(x, ldx, y, ldy, idx, H, W, C, P, Q, total, st);
    else launch_maxpool_fwd<bf16_t, bf16_t>(x, ldx, y, ldy, idx, H, W, C, P, Q, total, st);
    return check_launch("maxpool_fwd");
}
extern "C" int up_maxpool3s2_fwd(const float* x, int ldx, float* y, int ldy, uint8_t* idx, int N, int H, int W, int C,
                                 int P, int Q, void* stream) {
    return up_maxpool3s2_fwd_t(x, ldx, y, ldy, idx, N, H, W, C, P, Q, UP_DT_F32, UP_DT_F32, stream);
}
extern "C" int up_maxpool3s2_bwd_t(const void* dy, int lddy, const uint8_t* idx, void* dx, int lddx, int N, int H,
                                   int W, int C, int P, int Q, int dt_in, int dt_out, void* stream) {
    if (int e = pool_args_ok(N, H, W, C, P, Q, lddx, lddy)) return e;
    UP_REQUIRE(dy && idx && dx && UP_DT_OK(dt_in) && UP_DT_OK(dt_out), UP_ERR_INVALID, "maxpool_bwd: bad argument");
    int64_t total = (int64_t)N * H * W * (C / 4);
    hipStream_t st = as_stream(stream);
    if (dt_in == UP_DT_F32 && dt_out == UP_DT_F32) launch_maxpool_bwd<float, float>(dy, lddy, idx, dx, lddx, H, W, C, P, Q, total, st);
    else if (dt_in == UP_DT_F32) launch_maxpool_bwd<float, bf16_t>(dy, lddy, idx, dx, lddx, H, W, C, P, Q, total, st);
    else if (dt_out == UP_DT_F32) launch_maxpool_bwd<bf16_t, float>(dy, lddy, idx, dx, lddx, H, W, C, P, Q, total, st);
    else launch_maxpool_bwd<bf16_t, bf16_t>(dy, lddy, idx, dx, lddx, H, W, C, P, Q, total, st);
    return check_launch("maxpool_bwd");
}
extern "C" int up_maxpool3s2_bwd(const float* dy, int lddy, const uint8_t* idx, float* dx, int lddx, int N, int H,
                                 int W, int C, int P, int Q, void* stream) {
    return up_maxpool3s2_bwd_t(dy, lddy, idx, dx, lddx, N, H, W, C, P, Q, UP_DT_F32, UP_DT_F32, stream);
}

static inline float ac_scale(int in, int out) { return out > 1 ? (float)(in - 1) / (float)(out - 1) : 0.f; }

extern "C" int up_bilinear_fwd_t(const void* x, int ldx, void* y, int ldy, int N, int H, int W, int C, int P, int Q,
                                 int dtype, void* stream) {
    if (int e = pool_args_ok(N, H, W, C, P, Q, ldx, ldy)) return e;
    UP_REQUIRE(x && y && UP_DT_OK(dtype), UP_ERR_INVALID, "bilinear_fwd: bad argument");
    int64_t total = (int64_t)N * P * Q * (C / 4);
    if (dtype == UP_DT_BF16)
        UP_LAUNCH_1D(bilinear_fwd_kernel<bf16_t>, total, as_stream(stream), (const bf16_t*)x, ldx, (bf16_t*)y, ldy, H, W,
                     C / 4, P, Q, ac_scale(H, P), ac_scale(W, Q), total, make_fastdiv(C / 4), make_fastdiv(Q),
                     make_fastdiv(P));
    else
        UP_LAUNCH_1D(bilinear_fwd_kernel<float>, total, as_stream(stream), (const float*)x, ldx, (float*)y, ldy, H, W,
                     C / 4, P, Q, ac_scale(H, P), ac_scale(W, Q), total, make_fastdiv(C / 4), make_fastdiv(Q),
                     make_fastdiv(P));
    return check_launch("bilinear_fwd");
}
extern "C" int up_bilinear_fwd(const float* x, int ldx, float* y, int ldy, int N, int H, int W, int C, int P, int Q,
                               void* stream) {
    return up_bilinear_fwd_t(x, ldx, y, ldy, N, H, W, C, P, Q, UP_DT_F32, stream);
}
extern "C" int up_bilinear_bwd_t(const void* dy, int lddy, void* dx, int lddx, int N, int H, int W, int C, int P,
                                 int Q, int dtype, void* stream) {
    if (int e = pool_args_ok(N, H, W, C, P, Q, lddx, lddy)) return e;
    UP_REQUIRE(dy && dx && UP_DT_OK(dtype), UP_ERR_INVALID, "bilinear_bwd: bad argument");
    int64_t total = (int64_t)N * H * W * (C / 4);
    if (H == 1 && W == 1) {   // broadcast of a 1x1 map: a per-image column sum
        dim3 grid(N, cdiv(C, 64));
        if (dtype == UP_DT_BF16)
            hipLaunchKernelGGL(bcast_bwd_kernel<bf16_t>, grid, dim3(256), 0, as_stream(stream), (const bf16_t*)dy, lddy, (bf16_t*)dx,
                               lddx, P * Q, C);
        else
            hipLaunchKernelGGL(bcast_bwd_kernel<float>, grid, dim3(256), 0, as_stream(stream), (const float*)dy, lddy, (float*)dx, lddx,
                               P * Q, C);
        return check_launch("bilinear_bwd");
    }
    if (dtype == UP_DT_BF16)
        UP_LAUNCH_1D(bilinear_bwd_kernel<bf16_t>, total, as_stream(stream), (const bf16_t*)dy, lddy, (bf16_t*)dx, lddx, H,
                     W, C / 4, P, Q, ac_scale(H, P), ac_scale(W, Q), total, make_fastdiv(C / 4), make_fastdiv(W),
                     make_fastdiv(H));
    else
        UP_LAUNCH_1D(bilinear_bwd_kernel<float>, total, as_stream(stream), (const float*)dy, lddy, (float*)dx, lddx, H, W,
                     C / 4, P, Q, ac_scale(H, P), ac_scale(W, Q), total, make_fastdiv(C / 4), make_fastdiv(W),
                     make_fastdiv(H));
    return check_launch("bilinear_bwd");
}
extern "C" int up_bilinear_bwd(const float* dy, int lddy, float* dx, int lddx, int N, int H, int W, int C, int P,
                               int Q, void* stream) {
    return up_bilinear_bwd_t(dy, lddy, dx, lddx, N, H, W, C, P, Q, UP_DT_F32, stream);
}

extern "C" int up_gap_fwd_t(const void* x, int ldx, void* y, int N, int HW, int C, int dtype, void* stream) {
    UP_REQUIRE(x && y && N > 0 && HW > 0 && C > 0 && ldx >= C && UP_DT_OK(dtype), UP_ERR_INVALID, "gap_fwd: bad argument");
    if (dtype == UP_DT_BF16)
        hipLaunchKernelGGL(gap_fwd_kernel<bf16_t>, dim3(N, cdiv(C, 64)), dim3(256), 0, as_stream(stream), (const bf16_t*)x,
                           ldx, (bf16_t*)y, HW, C);
    else
        hipLaunchKernelGGL(gap_fwd_kernel<float>, dim3(N, cdiv(C, 64)), dim3(256), 0, as_stream(stream), (const float*)x,
                           ldx, (float*)y, HW, C);
    return check_launch("gap_fwd");
}
extern "C" int up_gap_fwd(const float* x, int ldx, float* y, int N, int HW, int C, void* stream) {
    return up_gap_fwd_t(x, ldx, y, N, HW, C, UP_DT_F32, stream);
}
extern "C" int up_gap_bwd_t(const void* dy, void* dx, int lddx, int N, int HW, int C, int dtype, void* stream) {
    UP_REQUIRE(dy && dx && N > 0 && HW > 0 && C > 0 && C % 4 == 0 && lddx % 4 == 0 && lddx >= C && UP_DT_OK(dtype),
               UP_ERR_INVALID, "gap_bwd: bad argument");
    int64_t total = (int64_t)N * HW * (C / 4);
    UP_REQUIRE(total < (1ll << 31), UP_ERR_UNSUPPORTED, "gap_bwd: tensor too large");
    if (dtype == UP_DT_BF16)
        UP_LAUNCH_1D(gap_bwd_kernel<bf16_t>, total, as_stream(stream), (const bf16_t*)dy, (bf16_t*)dx, lddx, HW, C / 4,
                     1.0f / (float)HW, total, make_fastdiv(C / 4), make_fastdiv(HW));
    else
        UP_LAUNCH_1D(gap_bwd_kernel<float>, total, as_stream(stream), (const float*)dy, (float*)dx, lddx, HW, C / 4,
                     1.0f / (float)HW, total, make_fastdiv(C / 4), make_fastdiv(HW));
    return check_launch("gap_bwd");
}
extern "C" int up_gap_bwd(const float* dy, float* dx, int lddx, int N, int HW, int C, void* stream) {
    return up_gap_bwd_t(dy, dx, lddx, N, HW, C, UP_DT_F32, stream);
}

extern "C" int up_avgpool9s8_fwd(const float* x, float* y, int ldy, int coff, int N, int H, int W, int P, int Q,
                                 void* stream) {
    UP_REQUIRE(x && y && N > 0 && H > 0 && W > 0 && coff >= 0 && coff < ldy, UP_ERR_INVALID, "avgpool: bad argument");
    UP_REQUIRE(P == (H + 2 - 9) / 8 + 1 && Q == (W + 2 - 9) / 8 + 1, UP_ERR_INVALID, "avgpool: P,Q mismatch");
    int64_t total = (int64_t)N * P * Q;
    UP_LAUNCH_1D(avgpool9s8_kernel, total, as_stream(stream), x, y, ldy, coff, H, W, P, Q, total);
    return check_launch("avgpool9s8");
}

extern "C" int up_lstm0_fwd(const float* gates, int ldg, float* cell, float* hide, int ldo, int64_t rows, int Cg,
                            void* stream) {
    UP_REQUIRE(gates && cell && hide && rows > 0 && Cg > 0 && ldg >= 3 * Cg && ldo >= Cg, UP_ERR_INVALID,
               "lstm0_fwd: bad argument");
    int64_t total = rows * Cg;
    UP_LAUNCH_1D(lstm0_fwd_kernel, total, as_stream(stream), gates, ldg, cell, hide, ldo, total, Cg);
    return check_launch("lstm0_fwd");
}
extern "C" int up_lstm0_bwd(const float* gates, int ldg, const float* dcell, const float* dhide, int ldo,
                            float* dgates, int64_t rows, int Cg, void* stream) {
    UP_REQUIRE(gates && dcell && dhide && dgates && rows > 0 && Cg > 0 && ldg >= 3 * Cg && ldo >= Cg, UP_ERR_INVALID,
               "lstm0_bwd: bad argument");
    int64_t total = rows * Cg;
    UP_LAUNCH_1D(lstm0_bwd_kernel, total, as_stream(stream), gates, ldg, dcell, dhide, ldo, dgates, total, Cg);
    return check_launch("lstm0_bwd");
}
extern "C" int up_lstm_fwd(const float* gates, int ldg, const float* cprev, int ldc, float* cell, float* hide, int ldo,
                           int64_t rows, int Cg, void* stream) {
    UP_REQUIRE(gates && cprev && cell && hide && rows > 0 && Cg > 0 && ldg >= 4 * Cg && ldo >= Cg && ldc >= Cg,
               UP_ERR_INVALID, "lstm_fwd: bad argument");
    int64_t total = rows * Cg;
    UP_LAUNCH_1D(lstm_fwd_kernel, total, as_stream(stream), gates, ldg, cprev, ldc, cell, hide, ldo, total, Cg);
    return check_launch("lstm_fwd");
}
extern "C" int up_lstm_bwd(const float* gates, int ldg, const float* cprev, int ldc, const float* cell,
                           const float* dcell, const float* dhide, int ldo, float* dgates, float* dcprev,
                           int64_t rows, int Cg, void* stream) {
    UP_REQUIRE(gates && cprev && cell && dcell && dhide && dgates && dcprev && rows > 0 && Cg > 0 && ldg >= 4 * Cg &&
                   ldo >= Cg && ldc >= Cg,
               UP_ERR_INVALID, "lstm_bwd: bad argument");
    int64_t total = rows * Cg;
    UP_LAUNCH_1D(lstm_bwd_kernel, total, as_stream(stream), gates, ldg, cprev, ldc, cell, dcell, dhide, ldo, dgates,
                 dcprev, total, Cg);
    return check_launch("lstm_bwd");
}

extern "C" int up_heatmap_argmax(const float* hm, int B, int J, int H, int W, int32_t* idx, float* preds_xy,
                                 float* maxvals, void* stream) {
    UP_REQUIRE(hm && preds_xy && maxvals && B > 0 && J > 0 && H > 0 && W > 0, UP_ERR_INVALID,
               "heatmap_argmax: bad argument");
    int maps = B * J;
    hipLaunchKernelGGL(argmax_kernel, dim3(cdiv(maps, 4)), dim3(256), 0, as_stream(stream), hm, maps, H * W, W, idx,
                       preds_xy, maxvals);
    return check_launch("heatmap_argmax");
}

extern "C" int up_make_heatmaps(const double* kpt_xy, int B, int K, int H, int W, double stride, double sigma,
                                float* out, void* stream) {
    UP_REQUIRE(kpt_xy && out && B > 0 && K > 0 && H > 0 && W > 0 && stride > 0 && sigma > 0, UP_ERR_INVALID,
               "make_heatmaps: bad argument");
    long long total = (long long)B * H * W;
    hipLaunchKernelGGL(heatmap_target_kernel, dim3(cdiv(total, 256)), dim3(256), 0, as_stream(stream), kpt_xy, K, H, W,
                       stride, sigma, out, total);
    return check_launch("make_heatmaps");
}
extern "C" int up_make_gaussian_maps(const double* center_xy, int N, int H, int W, double sigma, float* out,
                                     void* stream) {
    UP_REQUIRE(center_xy && out && N > 0 && H > 0 && W > 0 && sigma > 0, UP_ERR_INVALID, "make_gaussian_maps: bad argument");
    long long total = (long long)N * H * W;
    hipLaunchKernelGGL(gaussian_map_kernel, dim3(cdiv(total, 256)), dim3(256), 0, as_stream(stream), center_xy, H, W,
                       sigma, out, total);
    return check_launch("make_gaussian_maps");
}
extern "C" int up_normalize_image(const float* img_hwc, int B, int H, int W, int C, float mean, float stdv,
                                  float* out_chw, void* stream) {
    UP_REQUIRE(img_hwc && out_chw && B > 0 && H > 0 && W > 0 && C > 0 && stdv != 0.f, UP_ERR_INVALID,
               "normalize_image: bad argument");
    long long total = (long long)B * H * W * C;
    hipLaunchKernelGGL(normalize_image_kernel, dim3(cdiv(total, 256)), dim3(256), 0, as_stream(stream), img_hwc, C,
                       H * W, mean, stdv, out_chw, total);
    return check_launch("normalize_image");
}

extern "C" int up_peak_mask(const float* maps, int nmaps, int H, int W, uint8_t* mask, void* stream) {
    UP_REQUIRE(maps && mask && nmaps > 0 && H > 0 && W > 0, UP_ERR_INVALID, "peak_mask: bad argument");
    long long total = (long long)nmaps * H * W;
    hipLaunchKernelGGL(peak_mask_kernel, dim3(cdiv(total, 256)), dim3(256), 0, as_stream(stream), maps, total, H, W, mask);
    return check_launch("peak_mask");
}
extern "C" int up_box_argmax(const float* maps, int C, int H, int W, const int32_t* boxes, int P, int ch0, int nch,
                             int32_t* out_hw, void* stream) {
    UP_REQUIRE(maps && boxes && out_hw && C > 0 && H > 0 && W > 0 && P > 0, UP_ERR_INVALID, "box_argmax: bad argument");
    UP_REQUIRE(ch0 >= 0 && nch > 0 && ch0 + nch <= C, UP_ERR_INVALID, "box_argmax: channels %d..%d of %d", ch0,
               ch0 + nch - 1, C);
    hipLaunchKernelGGL(box_argmax_kernel, dim3(cdiv((long long)P * nch, 4)), dim3(256), 0, as_stream(stream), maps, H, W,
                       boxes, P, ch0, nch, out_hw);
    return check_launch("box_argmax");
}

extern "C" int up_pck_accuracy(const float* pred_xy, const float* target_xy, int B, int J, int H, int W, int dataset,
                               double thr_pck, double thr_pckh, double* acc, double* pck, double* pckh, double* visible,
                               int32_t* cnt, void* stream) {
    UP_REQUIRE(pred_xy && target_xy && acc && pck && pckh && visible && cnt && B > 0 && H > 0 && W > 0, UP_ERR_INVALID,
               "pck_accuracy: bad argument");
    UP_REQUIRE(J > 0 && J <= 256, UP_ERR_UNSUPPORTED, "pck_accuracy: %d joints (1..256 supported)", J);
    static const int need[] = {15, 14, 9, 5, 14, 8, 11};   // joints the head / torso formulas of each dataset touch
    UP_REQUIRE(dataset >= UP_DS_LSP && dataset <= UP_DS_MPII, UP_ERR_INVALID, "pck_accuracy: unknown dataset id %d", dataset);
    UP_REQUIRE(J >= need[dataset], UP_ERR_INVALID, "pck_accuracy: dataset %d needs at least %d joint channels, got %d",
               dataset, need[dataset], J);
    hipLaunchKernelGGL(pck_kernel, dim3(1), dim3(256), 0, as_stream(stream), pred_xy, target_xy, B, J, H, W, dataset,
                       (float)thr_pck, (float)thr_pckh, acc, pck, pckh, visible, cnt);
    return check_launch("pck_accuracy");
}
