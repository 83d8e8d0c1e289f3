// Whole-graph inference entry of the UniPose image network ("whole-graph up_unipose_forward fast path"; ABI 9).
//
// The reference runs its validation / test loops as `heat = model(input)` (unipose.py:150-160); through this entry the same
// forward — ResNet-101 (resnet.py:44-124) + WASP (wasp.py:66-90) + decoder (decoder.py:38-56) with every BatchNorm folded
// into its convolution (checkpoint.fold_batchnorm) — is ONE C call on one stream: no Python, no autograd, no allocation.
// Everything below goes through the PUBLIC C ABI of this library (up_conv2d_fwd, up_maxpool3s2_fwd, up_bilinear_fwd, ...): it is at
// the same time the example of how a C / C++ application drives the kernels.  The plan owns the packed
// weight images and biases (device memory, freed by up_unipose_plan_destroy); activations live in a caller-provided workspace
// whose size up_unipose_plan_workspace() reports (tensor lifetimes are planned, buffers are reused).
//
// The launches, their order, their descriptors and their epilogues are exactly those of the drop-in module's folded inference
// forward (unipose_amd/unipose.py + modules.py after checkpoint.load_folded), so the two produce equal bits
// (tests/test_plan_*.py).  Training has no whole-graph entry: it runs through autograd.
#include <algorithm>
#include <string>
#include <vector>

#include "up_common.h"

namespace up {
namespace plan {

static void* dev_alloc(size_t bytes) {
#ifdef UP_EMU
    return malloc(bytes);
#else
    void* p = nullptr;
    if (hipMalloc(&p, bytes) != hipSuccess) {
        (void)hipGetLastError();
        return nullptr;
    }
    return p;
#endif
}
static void dev_free(void* p) {
    if (!p) return;
#ifdef UP_EMU
    free(p);
#else
    (void)hipFree(p);
#endif
}

static inline int rup4(int c) { return (c + 3) / 4 * 4; }

struct Tensor {
    int n, h, w, c;        // NHWC, c = physical channels
    size_t bytes;
    int first = -1, last = -1;
    size_t off = 0;
};
struct Ref {
    int id = -1;
    size_t elems = 0;      // element offset inside the tensor (row slices of the stacked WASP branches)
};
struct Conv {
    std::string name;      // state_dict prefix: <name>.weight (+ <name>.bias after folding)
    up_conv_desc d;
    int relu;
    bool has_bias;         // the folded network has a bias on every convolution but wasp.conv2
    float* w_fwd = nullptr;
    float* bias = nullptr;
    bool set = false;
};
enum Kind { TO_NHWC, CONV, MAXPOOL, BILINEAR, GAP, COPY, ZERO, TO_NCHW };
struct Op {
    Kind kind;
    Ref in, out, res;
    int conv = -1;
    int a = 0, b = 0, c = 0, e = 0;   // kind-specific integers (see run)
};

struct Plan {
    up_unipose_config cfg;
    std::vector<Tensor> tensors;
    std::vector<Conv> convs;
    std::vector<Op> ops;
    size_t ws_bytes = 0;
    int input_id = -1;

    int tensor(int n, int h, int w, int c, size_t elem = 4) {
        Tensor t;
        t.n = n; t.h = h; t.w = w; t.c = c;
        t.bytes = ((size_t)n * h * w * c * elem + 255) / 256 * 256;
        tensors.push_back(t);
        return (int)tensors.size() - 1;
    }
    void touch(const Ref& r) {
        if (r.id < 0) return;
        Tensor& t = tensors[r.id];
        const int i = (int)ops.size();
        if (t.first < 0) t.first = i;
        t.last = i;
    }
    void push(Op op) {
        touch(op.in);
        touch(op.out);
        touch(op.res);
        ops.push_back(op);
    }
    // conv (+ folded-BatchNorm bias) (+ residual) (+ ReLU) of `images` images at (h, w) with cp physical input channels
    Ref conv(const std::string& name, Ref x, int images, int h, int w, int cp, int c, int k, int r, int stride, int pad, int dil,
             int relu, bool has_bias, Ref res = Ref(), Ref out = Ref()) {
        Conv cv;
        cv.name = name;
        cv.relu = relu;
        cv.has_bias = has_bias;
        up_conv_desc& d = cv.d;
        memset(&d, 0, sizeof(d));
        d.N = images; d.H = h; d.W = w; d.C = c; d.Cp = cp; d.ldx = cp;
        d.K = k; d.R = r; d.S = r; d.stride = stride; d.pad = pad; d.dil = dil;
        d.P = (h + 2 * pad - dil * (r - 1) - 1) / stride + 1;
        d.Q = (w + 2 * pad - dil * (r - 1) - 1) / stride + 1;
        d.Kp = rup4(k);
        d.ldy = d.Kp;
        if (out.id < 0) out.id = tensor(images, d.P, d.Q, d.ldy);
        convs.push_back(cv);
        Op op;
        op.kind = CONV;
        op.in = x; op.out = out; op.res = res;
        op.conv = (int)convs.size() - 1;
        push(op);
        return out;
    }
};

// resnet.py:5-42 — conv1 / conv2 / conv3 (+ downsample) of one Bottleneck, BatchNorm folded, the residual add and the last ReLU in
// conv3's epilogue (modules.Bottleneck.forward under ops.FoldedBatchNorm)
static Ref bottleneck(Plan& p, const std::string& name, Ref x, int n, int& h, int& w, int inplanes, int planes, int stride, int dil,
                      bool down) {
    Ref y = p.conv(name + ".conv1", x, n, h, w, inplanes, inplanes, planes, 1, 1, 0, 1, 1, true);
    y = p.conv(name + ".conv2", y, n, h, w, planes, planes, planes, 3, stride, dil, dil, 1, true);
    const int ho = p.tensors[y.id].h, wo = p.tensors[y.id].w;
    Ref skip = x;
    if (down) skip = p.conv(name + ".downsample.0", x, n, h, w, inplanes, inplanes, planes * 4, 1, stride, 0, 1, 0, true);
    Ref out = p.conv(name + ".conv3", y, n, ho, wo, planes, planes, planes * 4, 1, 1, 0, 1, 1, true, skip);
    h = ho;
    w = wo;
    return out;
}

static int build(Plan& p) {
    const up_unipose_config& c = p.cfg;
    const int n = c.batch;
    int h = c.height, w = c.width;
    int strides[4], dils[4];
    if (c.output_stride == 16) {
        const int s[4] = {1, 2, 2, 1}, d[4] = {1, 1, 1, 2};
        memcpy(strides, s, sizeof(s));
        memcpy(dils, d, sizeof(d));
    } else {
        const int s[4] = {1, 2, 1, 1}, d[4] = {1, 1, 2, 4};
        memcpy(strides, s, sizeof(s));
        memcpy(dils, d, sizeof(d));
    }
    // input NCHW -> NHWC, 3 -> 4 channels (ops.ToNHWC)
    Ref x;
    x.id = p.tensor(n, h, w, 4);
    {
        Op op;
        op.kind = TO_NHWC;
        op.out = x;
        p.push(op);
    }
    // stem: 7x7 s2 + folded bn1 + ReLU, 3x3 s2 max-pool (resnet.py:113-117)
    x = p.conv("backbone.conv1", x, n, h, w, 4, 3, 64, 7, 2, 3, 1, 1, true);
    h = p.tensors[x.id].h;
    w = p.tensors[x.id].w;
    auto maxpool = [&](Ref in, int hh, int ww, int ch) {
        Ref out, idx;
        const int ph = (hh - 1) / 2 + 1, pw = (ww - 1) / 2 + 1;
        out.id = p.tensor(n, ph, pw, ch);
        idx.id = p.tensor(n, ph, pw, ch, 1);
        Op op;
        op.kind = MAXPOOL;
        op.in = in; op.out = out; op.res = idx;
        op.a = hh; op.b = ww; op.c = ch;
        p.push(op);
        return out;
    };
    x = maxpool(x, h, w, 64);
    h = p.tensors[x.id].h;
    w = p.tensors[x.id].w;
    // layer1..4 (resnet.py:76-111): (3, 4, 23, 3) blocks, multi-grid (1, 2, 4) in layer4
    const int planes[4] = {64, 128, 256, 512}, blocks[4] = {3, 4, 23, 3};
    int inplanes = 64;
    Ref low;
    int low_h = 0, low_w = 0;
    for (int l = 0; l < 4; ++l) {
        for (int b = 0; b < blocks[l]; ++b) {
            const int grid = l == 3 ? (b == 0 ? 1 : b == 1 ? 2 : 4) : 1;
            const bool first = b == 0;
            const bool down = first && (strides[l] != 1 || inplanes != planes[l] * 4);
            x = bottleneck(p, "backbone.layer" + std::to_string(l + 1) + "." + std::to_string(b), x, n, h, w, inplanes, planes[l],
                           first ? strides[l] : 1, grid * dils[l], down);
            inplanes = planes[l] * 4;
        }
        if (l == 0) {
            low = x;
            low_h = h;
            low_w = w;
        }
    }
    // WASP (wasp.py:66-90; modules.WASP.forward): the four branch outputs are rows of ONE (4N, h, w, 256) tensor, so that the
    // twice-applied 1x1 convolution runs as two launches over 4 N h w rows
    const int wd[4] = {c.output_stride == 16 ? 24 : 48, c.output_stride == 16 ? 18 : 36, c.output_stride == 16 ? 12 : 24,
                       c.output_stride == 16 ? 6 : 12};
    Ref stack;
    stack.id = p.tensor(4 * n, h, w, 256);
    const size_t branch = (size_t)n * h * w * 256;
    Ref b1 = stack, b2 = stack, b3 = stack, b4 = stack;
    b2.elems = branch;
    b3.elems = 2 * branch;
    b4.elems = 3 * branch;
    p.conv("wasp.aspp1.atrous_conv", x, n, h, w, 2048, 2048, 256, 1, 1, 0, wd[0], 1, true, Ref(), b1);
    p.conv("wasp.aspp2.atrous_conv", b1, n, h, w, 256, 256, 256, 3, 1, wd[1], wd[1], 1, true, Ref(), b2);
    p.conv("wasp.aspp3.atrous_conv", b2, n, h, w, 256, 256, 256, 3, 1, wd[2], wd[2], 1, true, Ref(), b3);
    p.conv("wasp.aspp4.atrous_conv", b3, n, h, w, 256, 256, 256, 3, 1, wd[3], wd[3], 1, true, Ref(), b4);
    Ref y = p.conv("wasp.conv2", stack, 4 * n, h, w, 256, 256, 256, 1, 1, 0, 1, 0, false);
    y = p.conv("wasp.conv2", y, 4 * n, h, w, 256, 256, 256, 1, 1, 0, 1, 0, false);   // the SAME weight again (wasp.py:72-80)
    // global-average-pool branch: GAP -> 1x1 (+ folded BatchNorm) + ReLU -> bilinear 1x1 -> h x w (a broadcast)
    Ref g;
    g.id = p.tensor(n, 1, 1, 2048);
    {
        Op op;
        op.kind = GAP;
        op.in = x; op.out = g;
        op.a = h * w; op.c = 2048;
        p.push(op);
    }
    g = p.conv("wasp.global_avg_pool.1", g, n, 1, 1, 2048, 2048, 256, 1, 1, 0, 1, 1, true);
    auto bilinear = [&](Ref in, int hh, int ww, int ch, int ph, int pw) {
        Ref out;
        out.id = p.tensor(n, ph, pw, ch);
        Op op;
        op.kind = BILINEAR;
        op.in = in; op.out = out;
        op.a = hh; op.b = ww; op.c = ch; op.e = ph * 65536 + pw;
        p.push(op);
        return out;
    };
    g = bilinear(g, 1, 1, 256, h, w);
    Ref cat;
    cat.id = p.tensor(n, h, w, 1280);
    auto copy = [&](Ref src, int lds, Ref dst, int ldd, size_t dst_ch, long long rows, int ch) {
        Op op;
        op.kind = COPY;
        op.in = src; op.out = dst;
        op.out.elems += dst_ch;
        op.a = lds; op.b = ldd; op.c = ch; op.e = (int)rows;
        p.push(op);
    };
    for (int i = 0; i < 4; ++i) {
        Ref s = y;
        s.elems = i * branch;
        copy(s, 256, cat, 1280, (size_t)i * 256, (long long)n * h * w, 256);
    }
    copy(g, 256, cat, 1280, 1024, (long long)n * h * w, 256);
    x = p.conv("wasp.conv1", cat, n, h, w, 1280, 1280, 256, 1, 1, 0, 1, 1, true);   // + folded bn1 + ReLU; dropout is the identity in eval
    // decoder (decoder.py:38-56)
    Ref lw = p.conv("decoder.conv1", low, n, low_h, low_w, 256, 256, 48, 1, 1, 0, 1, 1, true);
    lw = maxpool(lw, low_h, low_w, 48);
    const int dh = p.tensors[lw.id].h, dw = p.tensors[lw.id].w;
    x = bilinear(x, h, w, 256, dh, dw);
    Ref cat2;
    cat2.id = p.tensor(n, dh, dw, 320);      // 256 + 48 = 304 real channels, zero-padded to a multiple of 32
    {
        Op op;
        op.kind = ZERO;
        op.out = cat2;
        p.push(op);
    }
    copy(x, 256, cat2, 320, 0, (long long)n * dh * dw, 256);
    copy(lw, 48, cat2, 320, 256, (long long)n * dh * dw, 48);
    x = p.conv("decoder.last_conv.0", cat2, n, dh, dw, 320, 304, 256, 3, 1, 1, 1, 1, true);
    x = p.conv("decoder.last_conv.4", x, n, dh, dw, 256, 256, 256, 3, 1, 1, 1, 1, true);
    x = p.conv("decoder.last_conv.8", x, n, dh, dw, 256, 256, c.out_channels, 1, 1, 0, 1, 0, true);
    {
        Op op;
        op.kind = TO_NCHW;
        op.in = x;
        op.a = dh; op.b = dw; op.c = c.out_channels; op.e = rup4(c.out_channels);
        p.push(op);
    }
    // workspace layout: first fit over the live ranges, buffers of dead tensors are reused
    struct Block {
        size_t off, bytes;
    };
    std::vector<Block> free_list;
    size_t top = 0;
    for (int i = 0; i < (int)p.ops.size(); ++i) {
        for (int t = 0; t < (int)p.tensors.size(); ++t) {
            Tensor& T = p.tensors[t];
            if (T.first != i) continue;
            bool placed = false;
            for (size_t f = 0; f < free_list.size(); ++f)
                if (free_list[f].bytes >= T.bytes) {
                    T.off = free_list[f].off;
                    free_list[f].off += T.bytes;
                    free_list[f].bytes -= T.bytes;
                    placed = true;
                    break;
                }
            if (!placed) {
                T.off = top;
                top += T.bytes;
            }
        }
        for (int t = 0; t < (int)p.tensors.size(); ++t) {
            Tensor& T = p.tensors[t];
            if (T.last != i) continue;
            free_list.push_back({T.off, T.bytes});
            // merge neighbours
            std::sort(free_list.begin(), free_list.end(), [](const Block& a, const Block& b) { return a.off < b.off; });
            std::vector<Block> merged;
            for (const Block& b : free_list) {
                if (b.bytes == 0) continue;
                if (!merged.empty() && merged.back().off + merged.back().bytes == b.off) merged.back().bytes += b.bytes;
                else merged.push_back(b);
            }
            free_list.swap(merged);
        }
    }
    p.ws_bytes = top;
    return UP_OK;
}

}  // namespace plan
}  // namespace up

using namespace up;
using up::plan::Plan;

struct up_unipose_plan {
    Plan p;
};

extern "C" int up_unipose_plan_create(const up_unipose_config* cfg, up_unipose_plan** out) {
    UP_REQUIRE(cfg && out, UP_ERR_INVALID, "unipose_plan_create: null argument");
    UP_REQUIRE(cfg->batch > 0 && cfg->height >= 32 && cfg->width >= 32 && cfg->out_channels > 0, UP_ERR_INVALID,
               "unipose_plan_create: batch %d, input %dx%d, %d output channels", cfg->batch, cfg->height, cfg->width, cfg->out_channels);
    UP_REQUIRE(cfg->output_stride == 16 || cfg->output_stride == 8, UP_ERR_UNSUPPORTED,
               "unipose_plan_create: output stride %d (the reference builds 16 and 8, resnet.py:49-58)", cfg->output_stride);
    up_unipose_plan* pl = new (std::nothrow) up_unipose_plan();
    UP_REQUIRE(pl, UP_ERR_INVALID, "unipose_plan_create: out of host memory");
    pl->p.cfg = *cfg;
    if (int e = up::plan::build(pl->p)) {
        delete pl;
        return e;
    }
    // plan-owned device memory: one forward weight image (+ bias) per DISTINCT parameter (wasp.conv2 is applied twice)
    for (size_t i = 0; i < pl->p.convs.size(); ++i) {
        plan::Conv& cv = pl->p.convs[i];
        for (size_t j = 0; j < i; ++j)
            if (pl->p.convs[j].name == cv.name) {
                cv.w_fwd = pl->p.convs[j].w_fwd;
                cv.bias = pl->p.convs[j].bias;
            }
        if (cv.w_fwd) continue;
        const size_t nf = (size_t)cv.d.K * cv.d.R * cv.d.S * cv.d.Cp;
        cv.w_fwd = static_cast<float*>(plan::dev_alloc(nf * sizeof(float)));
        if (cv.has_bias) cv.bias = static_cast<float*>(plan::dev_alloc((size_t)cv.d.K * sizeof(float)));
        if (!cv.w_fwd || (cv.has_bias && !cv.bias)) {
            up_unipose_plan_destroy(pl);
            UP_REQUIRE(false, UP_ERR_INVALID, "unipose_plan_create: out of device memory");
        }
    }
    *out = pl;
    return UP_OK;
}

extern "C" void up_unipose_plan_destroy(up_unipose_plan* pl) {
    if (!pl) return;
    for (size_t i = 0; i < pl->p.convs.size(); ++i) {
        plan::Conv& cv = pl->p.convs[i];
        bool shared = false;
        for (size_t j = 0; j < i; ++j) shared = shared || pl->p.convs[j].name == cv.name;
        if (shared) continue;
        plan::dev_free(cv.w_fwd);
        plan::dev_free(cv.bias);
    }
    delete pl;
}

extern "C" int up_unipose_plan_num_convs(const up_unipose_plan* pl) { return pl ? (int)pl->p.convs.size() : UP_ERR_INVALID; }

extern "C" const char* up_unipose_plan_conv_name(const up_unipose_plan* pl, int i) {
    return (pl && i >= 0 && i < (int)pl->p.convs.size()) ? pl->p.convs[i].name.c_str() : "";
}

extern "C" int up_unipose_plan_conv_shape(const up_unipose_plan* pl, int i, int32_t* oihw, int32_t* has_bias) {
    UP_REQUIRE(pl && oihw && i >= 0 && i < (int)pl->p.convs.size(), UP_ERR_INVALID, "unipose_plan_conv_shape: bad argument");
    const plan::Conv& cv = pl->p.convs[i];
    oihw[0] = cv.d.K; oihw[1] = cv.d.C; oihw[2] = cv.d.R; oihw[3] = cv.d.S;
    if (has_bias) *has_bias = cv.has_bias ? 1 : 0;
    return UP_OK;
}

extern "C" int up_unipose_plan_set_conv(up_unipose_plan* pl, int i, const float* w_oihw, const float* bias, void* stream) {
    UP_REQUIRE(pl && w_oihw && i >= 0 && i < (int)pl->p.convs.size(), UP_ERR_INVALID, "unipose_plan_set_conv: bad argument");
    plan::Conv& cv = pl->p.convs[i];
    UP_REQUIRE((bias != nullptr) == cv.has_bias, UP_ERR_INVALID, "unipose_plan_set_conv: %s %s a bias (folded network)", cv.name.c_str(),
               cv.has_bias ? "needs" : "has no");
    if (int e = up_pack_weights(&cv.d, w_oihw, cv.w_fwd, nullptr, stream)) return e;
    if (bias)
        if (int e = up_copy2d(bias, cv.d.K, cv.bias, cv.d.K, 1, cv.d.K, stream)) return e;
    for (plan::Conv& other : pl->p.convs)
        if (other.name == cv.name) other.set = true;
    return UP_OK;
}

extern "C" size_t up_unipose_plan_workspace(const up_unipose_plan* pl) { return pl ? pl->p.ws_bytes : 0; }

extern "C" int up_unipose_forward(up_unipose_plan* pl, const float* x_nchw, float* heat_nchw, void* workspace, size_t ws_bytes,
                                  void* stream) {
    UP_REQUIRE(pl && x_nchw && heat_nchw && workspace, UP_ERR_INVALID, "unipose_forward: null argument");
    for (const plan::Conv& cv : pl->p.convs)
        UP_REQUIRE(cv.set, UP_ERR_INVALID, "unipose_forward: weights of %s were never set (up_unipose_plan_set_conv)", cv.name.c_str());
    UP_REQUIRE(ws_bytes >= pl->p.ws_bytes && (reinterpret_cast<uintptr_t>(workspace) & 255) == 0, UP_ERR_INVALID,
               "unipose_forward: workspace of %zu bytes (256-byte aligned) needed, got %zu", pl->p.ws_bytes, ws_bytes);
    Plan& p = pl->p;
    const up_unipose_config& c = p.cfg;
    unsigned char* const ws = static_cast<unsigned char*>(workspace);
    auto ptr = [&](const plan::Ref& r) -> float* {
        return r.id < 0 ? nullptr : reinterpret_cast<float*>(ws + p.tensors[r.id].off) + r.elems;
    };
    for (const plan::Op& op : p.ops) {
        int e = UP_OK;
        switch (op.kind) {
        case plan::TO_NHWC:
            e = up_nchw_to_nhwc(x_nchw, ptr(op.out), c.batch, 3, c.height, c.width, 4, stream);
            break;
        case plan::CONV: {
            const plan::Conv& cv = p.convs[op.conv];
            up_conv_epilogue ep;
            memset(&ep, 0, sizeof(ep));
            ep.bias = cv.bias;
            ep.residual = ptr(op.res);
            ep.ldr = op.res.id >= 0 ? p.tensors[op.res.id].c : 0;
            ep.relu = cv.relu;
            e = up_conv2d_fwd(&cv.d, ptr(op.in), cv.w_fwd, ptr(op.out), &ep, stream);
            break;
        }
        case plan::MAXPOOL:
            e = up_maxpool3s2_fwd(ptr(op.in), op.c, ptr(op.out), op.c, reinterpret_cast<uint8_t*>(ptr(op.res)), c.batch, op.a, op.b, op.c,
                                  (op.a - 1) / 2 + 1, (op.b - 1) / 2 + 1, stream);
            break;
        case plan::BILINEAR:
            e = up_bilinear_fwd(ptr(op.in), op.c, ptr(op.out), op.c, c.batch, op.a, op.b, op.c, op.e >> 16, op.e & 65535, stream);
            break;
        case plan::GAP:
            e = up_gap_fwd(ptr(op.in), op.c, ptr(op.out), c.batch, op.a, op.c, stream);
            break;
        case plan::COPY:
            e = up_copy2d(ptr(op.in), op.a, ptr(op.out), op.b, op.e, op.c, stream);
            break;
        case plan::ZERO:
            if (hipMemsetAsync(ptr(op.out), 0, p.tensors[op.out.id].bytes, as_stream(stream)) != hipSuccess) e = check_launch("unipose_forward memset");
            break;
        case plan::TO_NCHW:
            e = up_nhwc_to_nchw(ptr(op.in), op.e, heat_nchw, c.batch, op.c, op.a, op.b, stream);
            break;
        }
        if (e) return e;
    }
    return UP_OK;
}
