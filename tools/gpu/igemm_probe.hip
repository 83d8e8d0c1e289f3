// Ablation probe for the implicit-GEMM kernel (development tool, not part of the library):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/gpu/igemm_probe.hip -o tools/gpu/igemm_probe
// Times igemm_kernel<BM,BN,true,DBG> for DBG ablations on two real layer shapes with hipEvents.
#include "../../unipose_amd/csrc/conv_igemm.hip"
#include "../../unipose_amd/csrc/norm_act.hip"

#include <vector>

using namespace up;

template <int BM, int BN, int DBG, int KT = 32>
static float run(IgemmArgs a, int iters) {
    a.ntn = cdiv(a.Ng, BN);
    a.nwg = cdiv(a.M, BM) * a.ntn;
    a.fNtn = make_fastdiv(a.ntn);
    a.fSpt = make_fastdiv(a.Cp / 32);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((igemm_kernel<BM, BN, 2, DBG, KT>), dim3(a.nwg), dim3(256), 0, 0, a);
    hipEventRecord(e0, 0);
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL((igemm_kernel<BM, BN, 2, DBG, KT>), dim3(a.nwg), dim3(256), 0, 0, a);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    return ms / iters;
}

template <int BM, int BN>
static void sweep(const char* name, up_conv_desc d) {
    size_t nx = (size_t)d.N * d.H * d.W * d.ldx, nw = (size_t)d.K * d.R * d.S * d.Cp, ny = (size_t)d.N * d.P * d.Q * d.ldy;
    float *x, *w, *y;
    hipMalloc(&x, nx * 4);
    hipMalloc(&w, nw * 4);
    hipMalloc(&y, ny * 4);
    std::vector<float> h(nx > nw ? nx : nw);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u >> 8) & 0xffff) / 65536.f - 0.5f;
    hipMemcpy(x, h.data(), nx * 4, hipMemcpyHostToDevice);
    hipMemcpy(w, h.data(), nw * 4, hipMemcpyHostToDevice);
    IgemmArgs a;
    fill_fwd_args(a, &d, x, w, y, nullptr);
    double fl = 2.0 * a.M * a.Ng * a.Ktot;
    float t[6];
    t[0] = run<BM, BN, 0>(a, 20);
    t[1] = run<BM, BN, 1>(a, 20);
    t[2] = run<BM, BN, 3>(a, 20);
    t[3] = run<BM, BN, 7>(a, 20);
    t[4] = run<BM, BN, 15>(a, 20);
    t[3] = run<BM, BN, 64>(a, 20);
    t[4] = run<BM, BN, 128>(a, 20);
    t[5] = run<BM, BN, 0>(a, 20);
    {   // effective shader clock while the kernel runs: full vs no-gload
        long long* dbg;
        hipMalloc(&dbg, 1 << 16);
        for (int mode = 0; mode < 2; ++mode) {
            hipMemset(dbg, 0, 1 << 16);
            IgemmArgs b = a;
            b.bias = nullptr;
            b.ntn = cdiv(b.Ng, BN);
            b.nwg = cdiv(b.M, BM) * b.ntn;
            b.fNtn = make_fastdiv(b.ntn);
            IgemmArgs c = b;
            c.bias = reinterpret_cast<const float*>(dbg);
            for (int it = 0; it < 5; ++it) {
                if (mode == 0) hipLaunchKernelGGL((igemm_kernel<BM, BN, 2, 0>), dim3(b.nwg), dim3(256), 0, 0, b);
                else hipLaunchKernelGGL((igemm_kernel<BM, BN, 2, 1>), dim3(b.nwg), dim3(256), 0, 0, b);
            }
            if (mode == 0) hipLaunchKernelGGL((igemm_kernel<BM, BN, 2, 32>), dim3(c.nwg), dim3(256), 0, 0, c);
            else hipLaunchKernelGGL((igemm_kernel<BM, BN, 2, 33>), dim3(c.nwg), dim3(256), 0, 0, c);
            hipDeviceSynchronize();
            std::vector<long long> h2(64);
            hipMemcpy(h2.data(), dbg, 64 * 8, hipMemcpyDeviceToHost);
            double cs = 0, ws = 0;
            int nb = c.nwg / 97 < 32 ? c.nwg / 97 : 32;
            for (int i = 0; i < nb; ++i) { cs += h2[2 * i]; ws += h2[2 * i + 1]; }
            printf("   %s: block lifetime %.0f shader cycles / %.0f wall ticks(100MHz) -> %.3f GHz\n", mode ? "no-gload" : "full    ",
                   cs / nb, ws / nb, cs / ws * 0.1);
        }
        hipFree(dbg);
    }
    const char* lab[6] = {"full", "no-gload", "no-gload,no-lstore", "single-buffer loop", "DB + pinned interleave", "full again (order check)"};
    printf("%s  tile %dx%d  M=%d N=%d K=%d  WGs=%d\n", name, BM, BN, a.M, a.Ng, a.Ktot, cdiv(a.M, BM) * cdiv(a.Ng, BN));
    for (int i = 0; i < 6; ++i) printf("   %-28s %8.4f ms  %7.1f TFLOP/s\n", lab[i], t[i], fl / t[i] / 1e9);
    hipFree(x);
    hipFree(w);
    hipFree(y);
}

static up_conv_desc mk(int N, int H, int C, int K, int R, int pad, int dil) {
    up_conv_desc d;
    memset(&d, 0, sizeof(d));
    d.N = N; d.H = d.W = H; d.C = d.Cp = d.ldx = C; d.K = K; d.R = d.S = R; d.stride = 1; d.pad = pad; d.dil = dil;
    d.P = d.Q = H; d.ldy = K; d.Kp = K;
    return d;
}

int main(int argc, char**) {
    if (argc > 1) {   // channel-stride experiment: power-of-two pixel stride vs not
        sweep<128, 128>("1x1 256->256 @92^2", mk(32, 92, 256, 256, 1, 0, 1));
        sweep<128, 128>("1x1 288->256 @92^2", mk(32, 92, 288, 256, 1, 0, 1));
        sweep<128, 128>("1x1 1024->256 @46^2", mk(32, 46, 1024, 256, 1, 0, 1));
        sweep<128, 128>("1x1 1056->256 @46^2", mk(32, 46, 1056, 256, 1, 0, 1));
        up_conv_desc d = mk(32, 46, 1024, 256, 1, 0, 1);
        d.ldx = 1056;   // same K, padded pixel stride
        sweep<128, 128>("1x1 1024->256 @46^2 ldx=1056", d);
        return 0;
    }
    sweep<128, 128>("1x1 512->256 @92^2 B32", mk(32, 92, 512, 256, 1, 0, 1));
    sweep<64, 64>("3x3 256->256 @23^2 B32", mk(32, 23, 256, 256, 3, 1, 1));
    sweep<128, 128>("3x3 256->256 @46^2 B32", mk(32, 46, 256, 256, 3, 1, 1));
    sweep<64, 128>("3x3 256->256 @46^2 B32 (64x128)", mk(32, 46, 256, 256, 3, 1, 1));
    return 0;
}
