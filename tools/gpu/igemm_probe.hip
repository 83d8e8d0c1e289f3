// Development probe for the implicit-GEMM kernel (not part of the library):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DUP_PROBE tools/gpu/igemm_probe.hip -o tools/gpu/igemm_probe
// For a few real layer shapes it times the production variant and records a per-workgroup timeline
// (start / end of the K loop / stores drained + XCC, SE, CU ids) to show where the CUs idle.
#include "../../unipose_amd/csrc/conv_igemm.hip"
#include "../../unipose_amd/csrc/norm_act.hip"

#include <algorithm>
#include <map>
#include <vector>

using namespace up;

template <int BM, int BN, int DBG>
static float run(IgemmArgs a, int iters) {
    a.ntn = cdiv(a.Ng, BN);
    a.nwg = cdiv(a.M, BM) * a.ntn;
    a.fNtn = make_fastdiv(a.ntn);
    a.fSpt = make_fastdiv(a.Cp / 32);
    a.full_blocks = a.nwg;
    a.parts = 1;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((igemm_kernel<BM, BN, 2, DBG>), dim3(a.nwg), dim3(256), 0, 0, a);
    hipEventRecord(e0, 0);
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL((igemm_kernel<BM, BN, 2, DBG>), dim3(a.nwg), dim3(256), 0, 0, a);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    return ms / iters;
}

static void analyze(std::vector<long long>& h, int nwg, double mfma_ticks_per_block, int full_blocks = -1, int parts = 1) {
    if (full_blocks >= 0 && full_blocks < nwg && parts > 1) {   // tail blocks: K loop / publish (parts 0..p-2) / merge + epilogue (last part)
        double kl = 0, pub = 0, mrg = 0, klm = 0; int np = 0, nm = 0;
        long long first_start = h[0], tail_first = h[4 * (size_t)full_blocks], tail_last_end = 0;
        for (int b = 0; b < nwg; ++b) first_start = std::min(first_start, h[4 * (size_t)b]);
        for (int b = full_blocks; b < nwg; ++b) {
            const int part = (b - full_blocks) % parts;
            tail_first = std::min(tail_first, h[4 * (size_t)b]);
            tail_last_end = std::max(tail_last_end, h[4 * (size_t)b + 2]);
            if (part < parts - 1) { kl += h[4 * (size_t)b + 1] - h[4 * (size_t)b]; pub += h[4 * (size_t)b + 2] - h[4 * (size_t)b + 1]; ++np; }
            else { klm += h[4 * (size_t)b + 1] - h[4 * (size_t)b]; mrg += h[4 * (size_t)b + 2] - h[4 * (size_t)b + 1]; ++nm; }
        }
        printf("   tail: %d publishing parts: K loop %.2f us, publish %.2f us;  %d merging parts: K loop %.2f us, wait+merge+epilogue %.2f us;"
               "  tail phase %.1f .. %.1f us after the first start\n", np, kl / np / 100.0, pub / np / 100.0, nm, klm / nm / 100.0,
               mrg / nm / 100.0, (tail_first - first_start) / 100.0, (tail_last_end - first_start) / 100.0);
    }
    struct { int nwg; } a{nwg};
    long long t0 = h[0], t1 = h[2];
    for (int b = 0; b < a.nwg; ++b) { t0 = std::min(t0, h[4 * b]); t1 = std::max(t1, h[4 * b + 2]); }
    double span = (double)(t1 - t0);
    std::map<long long, std::vector<int>> by_cu;
    double loop = 0, epi = 0;
    for (int b = 0; b < a.nwg; ++b) {
        long long id = h[4 * b + 3];
        long long key = (((id >> 32) & 15) << 16) | (((id >> 13) & 7) << 8) | ((id >> 8) & 15);   // xcc, se, cu
        by_cu[key].push_back(b);
        loop += h[4 * b + 1] - h[4 * b];
        epi += h[4 * b + 2] - h[4 * b + 1];
    }
    // residency histogram: fraction of (CU x time) with 0,1,2,3+ workgroups resident
    double occ[5] = {0, 0, 0, 0, 0}, gap = 0; long gaps = 0;
    size_t minb = 1 << 30, maxb = 0;
    for (auto& kv : by_cu) {
        std::vector<std::pair<long long, int>> ev;
        for (int b : kv.second) { ev.push_back({h[4 * b], +1}); ev.push_back({h[4 * b + 2], -1}); }
        std::sort(ev.begin(), ev.end());
        long long prev = t0; int cur = 0;
        for (auto& e : ev) { occ[std::min(cur, 4)] += (double)(e.first - prev); prev = e.first; cur += e.second; }
        occ[0] += (double)(t1 - prev);
        minb = std::min(minb, kv.second.size()); maxb = std::max(maxb, kv.second.size());
        // gap: time from a block's end to the next block start on this CU (greedy matching in time order)
        std::vector<long long> ends, starts;
        for (int b : kv.second) { ends.push_back(h[4 * b + 2]); starts.push_back(h[4 * b]); }
        std::sort(ends.begin(), ends.end()); std::sort(starts.begin(), starts.end());
        size_t si = 0;
        for (long long e : ends) {
            while (si < starts.size() && starts[si] < e) ++si;
            if (si < starts.size()) { gap += (double)(starts[si] - e); ++gaps; ++si; }
        }
    }
    double tot = span * by_cu.size();
    printf("   timeline: %zu CUs seen, blocks/CU %zu..%zu, span %.1f us; per block: K loop %.2f us, epilogue+drain %.2f us, "
           "pure-MFMA time %.2f us\n", by_cu.size(), minb, maxb, span / 100.0, loop / a.nwg / 100.0, epi / a.nwg / 100.0,
           mfma_ticks_per_block / 100.0);
    printf("             CU residency: 0 WG %.1f%%  1 WG %.1f%%  2 WG %.1f%%  3 WG %.1f%%  4+ %.1f%%;  end->next start on the "
           "same CU: %.2f us avg (%ld pairs)\n", 100 * occ[0] / tot, 100 * occ[1] / tot, 100 * occ[2] / tot, 100 * occ[3] / tot,
           100 * occ[4] / tot, gaps ? gap / gaps / 100.0 : 0.0, gaps);
    printf("             first blocks of XCD 0 (block:se.cu/ldsbase,ldssize):");
    for (int b = 0, n = 0; b < a.nwg && n < 40; b += 8, ++n) {
        long long id = h[4 * b + 3];
        printf(" %d:%d.%d/%d,%d", b, (int)((id >> 13) & 7), (int)((id >> 8) & 15), (int)((id >> 40) & 0xff), (int)((id >> 52) & 0x1ff));
    }
    printf("\n");
    {   // dispatch ramp: when do the blocks start / end (percentiles, us after the first start)
        std::vector<long long> st, en;
        for (int b = 0; b < a.nwg; ++b) { st.push_back(h[4 * b] - t0); en.push_back(h[4 * b + 2] - t0); }
        std::sort(st.begin(), st.end()); std::sort(en.begin(), en.end());
        auto pc = [&](std::vector<long long>& v, double q) { return v[(size_t)(q * (v.size() - 1))] / 100.0; };
        printf("             starts p10/p50/p90/p100: %.1f %.1f %.1f %.1f us;  ends p0/p10/p50/p90/p100: %.1f %.1f %.1f %.1f %.1f us\n",
               pc(st, .1), pc(st, .5), pc(st, .9), pc(st, 1.0), pc(en, 0), pc(en, .1), pc(en, .5), pc(en, .9), pc(en, 1.0));
    }
    // when do the XCDs finish?
    std::map<int, long long> xend;
    for (int b = 0; b < a.nwg; ++b) { int x = (int)((h[4 * b + 3] >> 32) & 15); xend[x] = std::max(xend[x], h[4 * b + 2]); }
    printf("             XCD finish times (us):");
    for (auto& kv : xend) printf(" %d:%.1f", kv.first, (kv.second - t0) / 100.0);
    printf("\n");
}

template <int BM, int BN, int DBG, bool SWZ = false>
static void timeline(IgemmArgs a, double mfma_ticks_per_block, bool split = false) {
    a.ntn = cdiv(a.Ng, BN);
    a.nwg = cdiv(a.M, BM) * a.ntn;
    a.fNtn = make_fastdiv(a.ntn);
    a.fSpt = make_fastdiv(a.Cp / 32);
    a.full_blocks = a.nwg;
    a.parts = 1;
    const int tiles = a.nwg;
    if (split) {   // the production tail split (launch_igemm)
        SplitScratch* sc = split_scratch(0);
        bool all_tiles = false;
        const int p = split_parts(a.nwg, a.Ktot, std::min(sc->pfloats / (size_t)(BM * BN), sc->nflags), &all_tiles);
        if (p >= 2) {
            a.full_blocks = all_tiles ? 0 : a.nwg / cu_count() * cu_count();
            a.parts = p;
            a.partials = sc->partials;
            a.flags = sc->flags;
            a.nwg = a.full_blocks + (tiles - a.full_blocks) * p;   // blocks launched (timeline records per block)
        }
    }
    long long* dbg;
    hipMalloc(&dbg, (size_t)a.nwg * 32);
    a.dbg = dbg;
    for (int i = 0; i < 2; ++i)
        hipLaunchKernelGGL((igemm_kernel<BM, BN, 2, DBG | 32, 32, false, SWZ>), dim3(a.nwg), dim3(256), 0, 0, a);
    hipDeviceSynchronize();
    std::vector<long long> h((size_t)a.nwg * 4);
    hipMemcpy(h.data(), dbg, h.size() * 8, hipMemcpyDeviceToHost);
    hipFree(dbg);
    analyze(h, a.nwg, mfma_ticks_per_block, a.full_blocks, a.parts);
}

template <int BM, int BN, int DBG>
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
    float ta = run<BM, BN, DBG>(a, 20);
    float tb = run<BM, BN, DBG>(a, 20);   // the first variant timed in a process runs ~10 % slow: time twice
    printf("%s  tile %dx%d  M=%d N=%d K=%d  WGs=%d\n   %.4f / %.4f ms  %.1f / %.1f TFLOP/s\n", name, BM, BN, a.M, a.Ng, a.Ktot,
           cdiv(a.M, BM) * cdiv(a.Ng, BN), ta, tb, fl / ta / 1e9, fl / tb / 1e9);
    // one block's MFMA work alone on a CU: (BM/32)*(BN/32)/4 tiles per wave x K/2 k-steps x 64 cycles @ 2.4 GHz
    double mfma_ticks = (double)(BM / 32) * (BN / 32) / 4.0 * (a.Ktot / 2.0) * 64.0 / 2.4e9 * 1e8;
    if (getenv("PROBE_TIMELINE")) {
        timeline<BM, BN, DBG>(a, mfma_ticks);
        timeline<BM, BN, DBG>(a, mfma_ticks, true);
    }
    {   // the production launch (tile as given, K loop variant and tail split chosen by launch_igemm)
        hipEvent_t e0, e1;
        hipEventCreate(&e0);
        hipEventCreate(&e1);
        for (int i = 0; i < 3; ++i) launch_igemm<BM, BN>(a, true, 0);
        hipEventRecord(e0, 0);
        for (int i = 0; i < 20; ++i) launch_igemm<BM, BN>(a, true, 0);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        ms /= 20;
        printf("   production launch: full blocks %d + %d tail tiles x %d parts: %.4f ms  %.1f TFLOP/s\n", a.full_blocks,
               a.nwg - a.full_blocks, a.parts, ms, fl / ms / 1e9);
    }
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


static void wgrad_sweep(const char* name, up_conv_desc d) {
    size_t nx = (size_t)d.N * d.H * d.W * d.ldx, ny = (size_t)d.N * d.P * d.Q * d.ldy, nw = (size_t)d.K * d.C * d.R * d.S;
    float *x, *dy, *dw;
    void* ws;
    hipMalloc(&x, nx * 4);
    hipMalloc(&dy, ny * 4);
    hipMalloc(&dw, nw * 4);
    g_wgrad_per_cu = 4;
    size_t wsb = up_conv2d_bwd_weight_workspace(&d) * 2;
    g_wgrad_per_cu = 2;
    hipMalloc(&ws, wsb);
    std::vector<float> h(nx > ny ? nx : ny);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u >> 8) & 0xffff) / 65536.f - 0.5f;
    hipMemcpy(x, h.data(), nx * 4, hipMemcpyHostToDevice);
    hipMemcpy(dy, h.data(), ny * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    float ms = 0.f;
    double fl = 2.0 * d.N * d.P * d.Q * (double)d.K * d.R * d.S * d.C;
    float best[4] = {1e9f, 1e9f, 1e9f, 1e9f};
    int wgs[4] = {0, 0, 0, 0};
    for (int round = 0; round < 4; ++round)       // interleaved rounds, best of each: timings drift upward within a process
        for (int v = 0; v < 4; ++v) {
            g_wgrad_single = v >= 1;
            g_wgrad_per_cu = v == 2 ? 3 : v == 3 ? 4 : 2;
            WgradPlan q = plan_wgrad(&d);
            wgs[v] = q.ntm * q.ntn * q.splits;
            for (int i = 0; i < 2; ++i) up_conv2d_bwd_weight(&d, x, dy, dw, nullptr, ws, wsb, nullptr);
            hipEventRecord(e0, 0);
            for (int i = 0; i < 10; ++i) up_conv2d_bwd_weight(&d, x, dy, dw, nullptr, ws, wsb, nullptr);
            hipEventRecord(e1, 0);
            hipEventSynchronize(e1);
            hipEventElapsedTime(&ms, e0, e1);
            ms /= 10;
            if (ms < best[v]) best[v] = ms;
        }
    for (int v = 0; v < 4; ++v)
        printf("   [%4d WGs] %s: %.4f ms  %.1f TFLOP/s\n", wgs[v], v ? "single-buffer loop" : "double-buffered loop", best[v],
               fl / best[v] / 1e9);
    ms = best[0];
    g_wgrad_single = false;
    g_wgrad_per_cu = 2;
    WgradPlan p = plan_wgrad(&d);
    printf("wgrad %s: tile %dx%d, %d tiles x %d splits = %d WGs, %d rows per split: %.4f ms (kernel + reduce)  %.1f TFLOP/s\n",
           name, p.bm, p.bn, p.ntm * p.ntn, p.splits, p.ntm * p.ntn * p.splits, p.rows_per_split, ms, fl / ms / 1e9);
    long long* dbg;
    hipMalloc(&dbg, (size_t)8192 * 32);
    hipMemset(dbg, 0, (size_t)8192 * 32);
    g_wgrad_dbg = dbg;
    up_conv2d_bwd_weight(&d, x, dy, dw, nullptr, ws, wsb, nullptr);
    hipDeviceSynchronize();
    g_wgrad_dbg = nullptr;
    std::vector<long long> hh((size_t)g_wgrad_grid * 4);
    hipMemcpy(hh.data(), dbg, hh.size() * 8, hipMemcpyDeviceToHost);
    double mfma_ticks = (double)(p.bm / 32) * (p.bn / 32) / 4.0 * (p.rows_per_split / 2.0) * 64.0 / 2.4e9 * 1e8;
    analyze(hh, g_wgrad_grid, mfma_ticks);
    hipFree(dbg); hipFree(x); hipFree(dy); hipFree(dw); hipFree(ws);
}
// the WASP dilated 3x3 convolutions (256->256 on 23x23, dilation 6 / 12 / 18 and 24 as used by the video
// variant): nominal and effective (non-padding MACs only) MFMA fraction and algorithmic bytes / time vs the 8 TB/s HBM peak
static void wasp_report(int dil) {
    up_conv_desc d = mk(32, 23, 256, 256, 3, dil, dil);
    size_t nx = (size_t)d.N * d.H * d.W * d.ldx, nw = (size_t)d.K * 9 * d.Cp, ny = (size_t)d.N * d.P * d.Q * d.ldy;
    float *x, *w, *y;
    hipMalloc(&x, nx * 4);
    hipMalloc(&w, nw * 4);
    hipMalloc(&y, ny * 4);
    hipMemset(x, 0, nx * 4);
    hipMemset(w, 0, nw * 4);
    IgemmArgs a;
    fill_fwd_args(a, &d, x, w, y, nullptr);
    float best = 1e9f;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int round = 0; round < 3; ++round) {
        for (int i = 0; i < 3; ++i) launch_igemm<64, 64>(a, true, 0);
        hipEventRecord(e0, 0);
        for (int i = 0; i < 20; ++i) launch_igemm<64, 64>(a, true, 0);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        best = ms / 20 < best ? ms / 20 : best;
    }
    double valid = 0;   // fraction of (pixel, tap) pairs inside the image
    for (int h = 0; h < 23; ++h)
        for (int wv = 0; wv < 23; ++wv)
            for (int r = -1; r <= 1; ++r)
                for (int s2 = -1; s2 <= 1; ++s2) {
                    int hh = h + r * dil, ww = wv + s2 * dil;
                    valid += hh >= 0 && hh < 23 && ww >= 0 && ww < 23;
                }
    valid /= 23.0 * 23.0 * 9.0;
    const double fl = 2.0 * a.M * a.Ng * a.Ktot, bytes = (double)(nx + ny + nw) * 4.0;
    printf("WASP 3x3 256->256 d%-2d @23^2 B32: %.4f ms  nominal %.1f TFLOP/s (%.1f %% of 157.3), effective (%.0f %% of the "
           "MACs touch the image) %.1f TFLOP/s (%.1f %%);  algorithmic %.1f MB -> %.3f TB/s = %.1f %% of 8 TB/s\n", dil, best,
           fl / best / 1e9, fl / best / 1e9 / 1.573, 100 * valid, valid * fl / best / 1e9, valid * fl / best / 1e9 / 1.573,
           bytes / 1e6, bytes / best / 1e9, bytes / best / 1e9 / 8.0 * 100.0);
    hipFree(x); hipFree(w); hipFree(y);
}

int main(int argc, char** argv) {
    if (argc > 1 && !strcmp(argv[1], "loops")) {   // single-buffer (64) vs double-buffered (128 / 0) loop at short reductions
        sweep<128, 128, 64>("1x1 512->256 @92^2 SB", mk(32, 92, 512, 256, 1, 0, 1));
        sweep<128, 128, 128>("1x1 512->256 @92^2 DB", mk(32, 92, 512, 256, 1, 0, 1));
        sweep<64, 128, 64>("1x1 256->1024 @23^2 SB", mk(32, 23, 256, 1024, 1, 0, 1));
        sweep<64, 128, 128>("1x1 256->1024 @23^2 DB", mk(32, 23, 256, 1024, 1, 0, 1));
        sweep<128, 128, 64>("1x1 512->2048 @23^2 SB", mk(32, 23, 512, 2048, 1, 0, 1));
        sweep<128, 128, 128>("1x1 512->2048 @23^2 DB", mk(32, 23, 512, 2048, 1, 0, 1));
        sweep<64, 128, 64>("1x1 512->128 @46^2 SB", mk(32, 46, 512, 128, 1, 0, 1));
        sweep<64, 128, 128>("1x1 512->128 @46^2 DB", mk(32, 46, 512, 128, 1, 0, 1));
        return 0;
    }
    if (argc > 1 && !strcmp(argv[1], "swz")) {   // padded vs XOR-swizzled LDS rows (4 vs 5 resident workgroups per CU), with the tail split
        for (auto cfg : {mk(32, 23, 256, 256, 3, 1, 1), mk(32, 23, 1024, 256, 1, 0, 1)}) {
            size_t nx = (size_t)cfg.N * cfg.H * cfg.W * cfg.ldx, nw = (size_t)cfg.K * cfg.R * cfg.S * cfg.Cp, ny = (size_t)cfg.N * cfg.P * cfg.Q * cfg.ldy;
            float *x, *w, *y;
            hipMalloc(&x, nx * 4); hipMalloc(&w, nw * 4); hipMalloc(&y, ny * 4);
            hipMemset(x, 0, nx * 4); hipMemset(w, 0, nw * 4);
            IgemmArgs a;
            fill_fwd_args(a, &cfg, x, w, y, nullptr);
            double mfma_ticks = (double)(64 / 32) * (64 / 32) / 4.0 * (a.Ktot / 2.0) * 64.0 / 2.4e9 * 1e8;
            printf("K=%d padded rows (36.9 KB):\n", a.Ktot);
            timeline<64, 64, 0, false>(a, mfma_ticks, true);
            printf("K=%d swizzled rows (32 KB):\n", a.Ktot);
            timeline<64, 64, 0, true>(a, mfma_ticks, true);
            hipFree(x); hipFree(w); hipFree(y);
        }
        return 0;
    }
    if (argc > 1 && !strcmp(argv[1], "alone")) {   // how fast is a workgroup that is alone on its CU?  256 / 512 / 1024 / 2048 tiles of 64x64
        sweep<64, 64, 0>("3x3 256->64 @32^2 B16: 1 tile per CU", mk(16, 32, 256, 64, 3, 1, 1));
        sweep<64, 64, 0>("3x3 256->128 @32^2 B16: 2 tiles per CU", mk(16, 32, 256, 128, 3, 1, 1));
        sweep<64, 64, 0>("3x3 256->256 @32^2 B16: 4 tiles per CU", mk(16, 32, 256, 256, 3, 1, 1));
        sweep<64, 64, 0>("3x3 256->512 @32^2 B16: 8 tiles per CU", mk(16, 32, 256, 512, 3, 1, 1));
        sweep<64, 64, 64>("3x3 256->64 @32^2 B16: 1 tile per CU, single-buffer loop", mk(16, 32, 256, 64, 3, 1, 1));
        sweep<64, 64, 64>("3x3 256->256 @32^2 B16: 4 tiles per CU, single-buffer loop", mk(16, 32, 256, 256, 3, 1, 1));
        sweep<64, 64, 0>("1x1 256->64 @32^2 B16: 1 tile per CU, K=256", mk(16, 32, 256, 64, 1, 0, 1));
        sweep<64, 64, 256>("3x3 256->64 @32^2 B16: 1 tile per CU, EARLY BARRIER", mk(16, 32, 256, 64, 3, 1, 1));
        sweep<64, 64, 256>("3x3 256->128 @32^2 B16: 2 tiles per CU, EARLY BARRIER", mk(16, 32, 256, 128, 3, 1, 1));
        sweep<64, 64, 256>("3x3 256->256 @32^2 B16: 4 tiles per CU, EARLY BARRIER", mk(16, 32, 256, 256, 3, 1, 1));
        sweep<64, 64, 256>("3x3 256->512 @32^2 B16: 8 tiles per CU, EARLY BARRIER", mk(16, 32, 256, 512, 3, 1, 1));
        return 0;
    }
    if (argc > 1 && !strcmp(argv[1], "wasp")) {
        for (int dil : {6, 12, 18, 24}) wasp_report(dil);
        return 0;
    }
    if (argc > 1 && !strcmp(argv[1], "wgrad")) {
        wgrad_sweep("3x3 256->256 @23^2", mk(32, 23, 256, 256, 3, 1, 1));
        wgrad_sweep("1x1 1024->256 @23^2", mk(32, 23, 1024, 256, 1, 0, 1));
        wgrad_sweep("1x1 256->1024 @23^2", mk(32, 23, 256, 1024, 1, 0, 1));
        wgrad_sweep("3x3 512->512 d2 @23^2", mk(32, 23, 512, 512, 3, 2, 2));
        wgrad_sweep("3x3 64->64 @92^2", mk(32, 92, 64, 64, 3, 1, 1));
        wgrad_sweep("1x1 64->256 @92^2", mk(32, 92, 64, 256, 1, 0, 1));
        return 0;
    }
    if (argc > 1) {   // tile choice with the tail split available
        sweep<64, 64, 0>("3x3 256->256 @23^2", mk(32, 23, 256, 256, 3, 1, 1));
        sweep<64, 128, 128>("3x3 256->256 @23^2", mk(32, 23, 256, 256, 3, 1, 1));
        sweep<128, 64, 128>("3x3 256->256 @23^2", mk(32, 23, 256, 256, 3, 1, 1));
        sweep<128, 128, 128>("3x3 256->256 @23^2", mk(32, 23, 256, 256, 3, 1, 1));
        sweep<64, 64, 0>("1x1 1024->256 @23^2", mk(32, 23, 1024, 256, 1, 0, 1));
        sweep<64, 128, 128>("1x1 1024->256 @23^2", mk(32, 23, 1024, 256, 1, 0, 1));
        sweep<128, 128, 128>("1x1 1024->256 @23^2", mk(32, 23, 1024, 256, 1, 0, 1));
        sweep<128, 128, 64>("1x1 256->1024 @23^2", mk(32, 23, 256, 1024, 1, 0, 1));
        sweep<64, 128, 64>("1x1 256->1024 @23^2", mk(32, 23, 256, 1024, 1, 0, 1));
        sweep<64, 128, 128>("3x3 512->512 d2 @23^2", mk(32, 23, 512, 512, 3, 2, 2));
        sweep<128, 128, 128>("3x3 512->512 d2 @23^2", mk(32, 23, 512, 512, 3, 2, 2));
        sweep<64, 128, 128>("3x3 128->128 @46^2", mk(32, 46, 128, 128, 3, 1, 1));
        sweep<128, 128, 128>("3x3 128->128 @46^2", mk(32, 46, 128, 128, 3, 1, 1));
        sweep<128, 64, 64>("1x1 256->64 @92^2", mk(32, 92, 256, 64, 1, 0, 1));
        sweep<128, 64, 64>("3x3 64->64 @92^2", mk(32, 92, 64, 64, 3, 1, 1));
        return 0;
    }

    // production variants (launch_igemm): 128-wide tiles, K >= 1024 -> 128; 64x64, K >= 1024 -> 0; K < 1024 -> 64
    sweep<128, 128, 64>("1x1 512->256 @92^2", mk(32, 92, 512, 256, 1, 0, 1));
    sweep<64, 64, 0>("3x3 256->256 @23^2", mk(32, 23, 256, 256, 3, 1, 1));
    sweep<128, 128, 64>("1x1 256->1024 @23^2", mk(32, 23, 256, 1024, 1, 0, 1));
    sweep<64, 64, 0>("1x1 1024->256 @23^2", mk(32, 23, 1024, 256, 1, 0, 1));
    sweep<128, 128, 128>("3x3 256->256 @46^2", mk(32, 46, 256, 256, 3, 1, 1));
    sweep<64, 128, 128>("3x3 512->512 d2 @23^2", mk(32, 23, 512, 512, 3, 2, 2));
    return 0;
}
