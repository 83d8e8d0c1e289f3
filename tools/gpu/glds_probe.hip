// Development probe for the bf16-storage kernels of unipose_amd/csrc/bf16s_glds.h (not part of the library):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DUP_PROBE tools/gpu/glds_probe.hip -o tools/gpu/glds_probe
// For the layer shapes that carry the 736x736 B=16 step it times kernel variants side by side (interleaved rounds, best of
// three) and prints the per-workgroup phase breakdown from eight time stamps (100 MHz): set-up, first slice landed, K loop,
// epilogue issued, stores drained — where a tile's lifetime goes.
#include "../../unipose_amd/csrc/conv_igemm.hip"
#include "../../unipose_amd/csrc/norm_act.hip"

#include <algorithm>
#include <vector>

using namespace up;

static up_conv_desc mk(int n, int hw, int c, int k, int r, int pad, int dil) {
    up_conv_desc d;
    memset(&d, 0, sizeof(d));
    d.N = n; d.H = hw; d.W = hw; d.C = c; d.Cp = c; d.ldx = c;
    d.K = k; d.R = r; d.S = r; d.stride = 1; d.pad = pad; d.dil = dil;
    d.P = hw; d.Q = hw; d.Kp = k; d.ldy = k;
    return d;
}

typedef void (*kern_t)(IgemmArgs);
struct Variant {
    const char* name;
    kern_t k;
    int bm, bn;
    bool dbg;
};

static void prep(IgemmArgs& a, int bm, int bn) {
    a.ntn = cdiv(a.Ng, bn);
    a.nwg = cdiv(a.M, bm) * a.ntn;
    a.fNtn = make_fastdiv(a.ntn);
    a.fSpt = make_fastdiv(a.Cp / 32);
    a.full_blocks = a.nwg;
    a.parts = 1;
}

static void shape(const char* name, up_conv_desc d, bool stats, std::vector<Variant> vs) {
    const size_t nx = (size_t)d.N * d.H * d.W * d.ldx, nw = (size_t)d.K * d.R * d.S * d.Cp, ny = (size_t)d.N * d.P * d.Q * d.ldy;
    uint16_t *x, *w, *y;
    float* st;
    hipMalloc(&x, nx * 2);
    hipMalloc(&w, nw * 2);
    hipMalloc(&y, ny * 2);
    const int M = d.N * d.P * d.Q;
    hipMalloc(&st, (size_t)cdiv(M, 64) * d.K * 3 * 4);
    std::vector<uint16_t> h(std::max(nx, nw));
    for (size_t i = 0; i < h.size(); ++i) {   // bf16 in [-1, 1): random sign / exponent 125..126 / mantissa
        uint32_t r = (uint32_t)(i * 2654435761u) >> 7;
        h[i] = (uint16_t)(((r & 1) << 15) | ((125 + ((r >> 1) & 1)) << 7) | ((r >> 2) & 127));
    }
    hipMemcpy(x, h.data(), nx * 2, hipMemcpyHostToDevice);
    hipMemcpy(w, h.data(), nw * 2, hipMemcpyHostToDevice);
    up_conv_epilogue ep;
    memset(&ep, 0, sizeof(ep));
    if (stats) ep.stats = st;
    IgemmArgs a0;
    fill_fwd_args(a0, &d, reinterpret_cast<const float*>(x), nullptr, reinterpret_cast<float*>(y), &ep);
    a0.w_hi = w;
    a0.x_bytes = (uint32_t)(nx * 2);
    long long* dbg;
    hipMalloc(&dbg, (size_t)(1 << 16) * 64);
    a0.dbg = dbg;
    const double fl = 2.0 * a0.M * (double)a0.Ng * a0.Ktot;
    printf("%s: M=%d N=%d K=%d%s  (MFMA floor %.1f us)\n", name, a0.M, a0.Ng, a0.Ktot, stats ? " +stats" : "", fl / 2.5e15 * 1e6);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    std::vector<float> best(vs.size(), 1e9f);
    for (int round = 0; round < 3; ++round)
        for (size_t v = 0; v < vs.size(); ++v) {
            IgemmArgs a = a0;
            prep(a, vs[v].bm, vs[v].bn);
            for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(vs[v].k, dim3(a.nwg), dim3(256), 0, 0, a);
            hipEventRecord(e0, 0);
            for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(vs[v].k, dim3(a.nwg), dim3(256), 0, 0, a);
            hipEventRecord(e1, 0);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            best[v] = std::min(best[v], ms / 20);
        }
    for (size_t v = 0; v < vs.size(); ++v) {
        IgemmArgs a = a0;
        prep(a, vs[v].bm, vs[v].bn);
        printf("  %-34s %4d WGs  %8.2f us  %7.1f TFLOP/s", vs[v].name, a.nwg, best[v] * 1e3, fl / best[v] / 1e9);
        if (vs[v].dbg) {
            hipMemset(dbg, 0, (size_t)a.nwg * 64);
            hipLaunchKernelGGL(vs[v].k, dim3(a.nwg), dim3(256), 0, 0, a);
            hipDeviceSynchronize();
            std::vector<long long> s((size_t)a.nwg * 8);
            hipMemcpy(s.data(), dbg, s.size() * 8, hipMemcpyDeviceToHost);
            double ph[5] = {0, 0, 0, 0, 0};
            long long t0 = s[0], t1 = s[5];
            for (int b = 0; b < a.nwg; ++b) {
                const long long* q = &s[(size_t)b * 8];
                for (int k = 0; k < 5; ++k) ph[k] += (double)(q[k + 1] - q[k]);
                t0 = std::min(t0, q[0]);
                t1 = std::max(t1, q[5]);
            }
            printf("   | per WG (us): set-up %.2f, first slice %.2f, K loop %.2f, epilogue %.2f, drain %.2f; span %.1f us", ph[0] / a.nwg / 100,
                   ph[1] / a.nwg / 100, ph[2] / a.nwg / 100, ph[3] / a.nwg / 100, ph[4] / a.nwg / 100, (t1 - t0) / 100.0);
        }
        printf("\n");
    }
    hipFree(x); hipFree(w); hipFree(y); hipFree(st); hipFree(dbg);
}

#define G(BM, BN, KT, ST, OCC, EPI, DBG) glds::igemm_glds_kernel<BM, BN, false, KT, ST, OCC, EPI, DBG>

int main(int argc, char** argv) {
    if (argc > 1 && !strcmp(argv[1], "forms")) {   // which slice form per layer shape?  (64 | 32 channels, 2 | 3 stages)
        std::vector<Variant> v = {
            {"register-staged <128,128> (round 2)", igemm_bf16_kernel<128, 128, 2, false, 32, true>, 128, 128, false},
            {"glds 128x128 kt32 st2", G(128, 128, 32, 2, 4, 0, 0), 128, 128, false},
            {"glds 128x128 kt32 st3", G(128, 128, 32, 3, 3, 0, 0), 128, 128, false},
            {"glds 128x128 kt64 st2", G(128, 128, 64, 2, 2, 0, 0), 128, 128, false},
        };
        std::vector<Variant> v64 = {
            {"register-staged <128,64> (round 2)", igemm_bf16_kernel<128, 64, 2, false, 32, true>, 128, 64, false},
            {"glds 128x64 kt32 st2", G(128, 64, 32, 2, 4, 0, 0), 128, 64, false},
            {"glds 128x64 kt32 st3", G(128, 64, 32, 3, 3, 0, 0), 128, 64, false},
            {"glds 128x64 kt64 st2", G(128, 64, 64, 2, 2, 0, 0), 128, 64, false},
        };
        shape("layer4 conv2 3x3 512->512 d4 @46", mk(16, 46, 512, 512, 3, 4, 4), true, v);
        shape("layer4 conv1 1x1 2048->512 @46", mk(16, 46, 2048, 512, 1, 0, 1), true, v);
        shape("layer4 conv3 1x1 512->2048 @46", mk(16, 46, 512, 2048, 1, 0, 1), true, v);
        shape("layer4.0 downsample 1x1 1024->2048 @46", mk(16, 46, 1024, 2048, 1, 0, 1), true, v);
        shape("decoder 3x3 320->256 @92", mk(16, 92, 320, 256, 3, 1, 1), true, v);
        shape("decoder 3x3 256->256 @92", mk(16, 92, 256, 256, 3, 1, 1), true, v);
        shape("layer2 conv2 3x3 128->128 @92", mk(16, 92, 128, 128, 3, 1, 1), true, v);
        shape("layer2 conv1 1x1 512->128 @92", mk(16, 92, 512, 128, 1, 0, 1), true, v);
        shape("wasp 3x3 256->256 d6 @46", mk(16, 46, 256, 256, 3, 6, 6), true, v);
        shape("layer1 conv2 3x3 64->64 @184", mk(16, 184, 64, 64, 3, 1, 1), true, v64);
        shape("layer1 conv1 1x1 256->64 @184", mk(16, 184, 256, 64, 1, 0, 1), true, v64);
        return 0;
    }
    std::vector<Variant> v128 = {
        {"register-staged <128,128> (round 2)", igemm_bf16_kernel<128, 128, 2, false, 32, true>, 128, 128, false},
        {"glds 128x128 kt32 st2", G(128, 128, 32, 2, 4, 0, 0), 128, 128, false},
        {"glds 128x128 kt32 st2 +stamps", G(128, 128, 32, 2, 4, 0, 1), 128, 128, true},
        {"glds 128x128 kt32 st2 old epilogue", G(128, 128, 32, 2, 4, 1, 0), 128, 128, false},
        {"glds 128x128 kt32 st3", G(128, 128, 32, 3, 3, 0, 0), 128, 128, false},
        {"glds 128x128 kt64 st2", G(128, 128, 64, 2, 2, 0, 0), 128, 128, false},
        {"glds 128x128 kt64 st2 +stamps", G(128, 128, 64, 2, 2, 0, 1), 128, 128, true},
        {"glds 64x128 kt32 st2", G(64, 128, 32, 2, 4, 0, 0), 64, 128, false},
        {"glds 128x64 kt32 st2", G(128, 64, 32, 2, 4, 0, 0), 128, 64, false},
    };
    shape("layer3 conv3 1x1 256->1024 @46", mk(16, 46, 256, 1024, 1, 0, 1), true, v128);
    shape("layer3 conv1 1x1 1024->256 @46", mk(16, 46, 1024, 256, 1, 0, 1), true, v128);
    shape("layer3 conv2 3x3 256->256 @46", mk(16, 46, 256, 256, 3, 1, 1), true, v128);
    shape("layer3 conv2 3x3 dgrad-like (no stats)", mk(16, 46, 256, 256, 3, 1, 1), false, v128);
    shape("layer1 conv3 1x1 64->256 @184", mk(16, 184, 64, 256, 1, 0, 1), true, v128);
    shape("layer2 conv3 1x1 128->512 @92", mk(16, 92, 128, 512, 1, 0, 1), true, v128);
    shape("layer4 conv2 3x3 512->512 d4 @46", mk(16, 46, 512, 512, 3, 4, 4), true, v128);
    return 0;
}
