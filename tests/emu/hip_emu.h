// TEST INFRASTRUCTURE — a minimal CPU emulation of the HIP device surface the kernels in
// unipose_amd/csrc use, so their index/tiling/fragment logic can be exercised without a GPU.
// It is force-included (g++ -include) by tests/emu/build_emu.py when it compiles the *.hip sources
// into tests/emu/libunipose_emu.so.  The product (unipose_amd/_C.py) never loads that library.
//
// Model: a thread block = 256 cooperative fibers on one OS thread (hand-rolled x86-64 stack switch),
// run round-robin; __syncthreads / wave-level exchanges (shuffles, MFMA) are generation barriers that
// yield.  v_mfma_f32_32x32x2_f32 is emulated with the documented operand/result lane maps
// (cdna_hip_programming.md §3): A lane l -> A[i=l&31][k=l>>5], B lane l -> B[k=l>>5][j=l&31],
// D reg r of lane l -> D[(r&3)+8*(r>>2)+4*(l>>5)][l&31], k-ordered fmaf chain.
#pragma once
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define __shared__ static thread_local

typedef float f32x16 __attribute__((vector_size(64)));

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned a = 1, unsigned b = 1, unsigned c = 1) : x(a), y(b), z(c) {}
};
struct float4 {
    float x, y, z, w;
};
static inline float4 make_float4(float a, float b, float c, float d) { return float4{a, b, c, d}; }
struct uint2 {
    uint32_t x, y;
};
struct uint4 {
    uint32_t x, y, z, w;
};
static inline uint4 make_uint4(uint32_t a, uint32_t b, uint32_t c, uint32_t d) { return uint4{a, b, c, d}; }

typedef void* hipStream_t;
typedef int hipError_t;
enum { hipSuccess = 0 };
static inline hipError_t hipGetLastError() { return hipSuccess; }
static inline const char* hipGetErrorString(hipError_t) { return "emu"; }
static inline hipError_t hipMemsetAsync(void* p, int v, size_t n, hipStream_t) {
    memset(p, v, n);
    return hipSuccess;
}

using std::max;
using std::min;

extern "C" void emu_switch(void** save_sp, void* new_sp);

namespace emu {

struct Bar {
    int count = 0, gen = 0;
};
struct Wave {
    Bar bar;
    uint32_t xa[64], xb[64];
};
struct Fiber {
    void* sp = nullptr;
    char* stack = nullptr;
    bool done = true;
    dim3 tid;
    int lin = 0;
};
struct State {
    std::vector<Fiber> fibers;
    std::vector<Wave> waves;
    Bar block_bar;
    int live = 0;
    int nthreads = 0;
    void* main_sp = nullptr;
    Fiber* cur = nullptr;
    dim3 bid, bdim, gdim;
    const std::function<void()>* body = nullptr;
};
inline State& st() {
    static thread_local State s;
    return s;
}
constexpr size_t STACK = 96 * 1024;

inline void yield() {
    State& s = st();
    emu_switch(&s.cur->sp, s.main_sp);
}
inline void bar_wait(Bar& b, int expected) {
    int g = b.gen;
    if (++b.count >= expected) {
        b.count = 0;
        b.gen++;
        return;
    }
    while (b.gen == g) yield();
}
inline void trampoline() {
    State& s = st();
    (*s.body)();
    s.cur->done = true;
    s.live--;
    if (s.block_bar.count > 0 && s.block_bar.count >= s.live) {  // a late exit may complete a barrier
        s.block_bar.count = 0;
        s.block_bar.gen++;
    }
    emu_switch(&s.cur->sp, s.main_sp);
    abort();
}
inline void run_block(dim3 bid, dim3 bdim, dim3 gdim, const std::function<void()>& body) {
    State& s = st();
    int n = (int)(bdim.x * bdim.y * bdim.z);
    if ((int)s.fibers.size() < n) {
        size_t old = s.fibers.size();
        s.fibers.resize(n);
        for (size_t i = old; i < (size_t)n; ++i) s.fibers[i].stack = (char*)aligned_alloc(64, STACK);
    }
    s.waves.assign((n + 63) / 64, Wave());
    s.block_bar = Bar();
    s.nthreads = s.live = n;
    s.bid = bid;
    s.bdim = bdim;
    s.gdim = gdim;
    s.body = &body;
    for (int i = 0; i < n; ++i) {
        Fiber& f = s.fibers[i];
        f.done = false;
        f.lin = i;
        f.tid = dim3(i % bdim.x, (i / bdim.x) % bdim.y, i / (bdim.x * bdim.y));
        uintptr_t top = ((uintptr_t)f.stack + STACK) & ~(uintptr_t)15;
        void** sp = (void**)top;
        *--sp = nullptr;                 // fake return address of the trampoline
        *--sp = (void*)&trampoline;      // popped by `ret` in emu_switch
        for (int r = 0; r < 6; ++r) *--sp = nullptr;  // rbp rbx r12 r13 r14 r15
        f.sp = sp;
    }
    while (s.live > 0) {
        for (int i = 0; i < n; ++i) {
            Fiber& f = s.fibers[i];
            if (f.done) continue;
            s.cur = &f;
            emu_switch(&s.main_sp, f.sp);
        }
    }
}
inline int workers() {
    const char* e = getenv("UP_EMU_THREADS");
    int n = e ? atoi(e) : (int)std::thread::hardware_concurrency();
    return n < 1 ? 1 : n;
}
// Persistent worker pool: fiber stacks (24 MB per worker) are allocated once, not per kernel launch.
struct Pool {
    std::vector<std::thread> th;
    std::mutex mu;
    std::condition_variable cv, done_cv;
    std::function<void(long)> job;
    long total = 0, generation = 0;
    std::atomic<long> next{0};
    int active = 0;
    bool stop = false;
    explicit Pool(int n) {
        for (int i = 0; i < n; ++i) th.emplace_back([this] { loop(); });
    }
    ~Pool() {
        {
            std::lock_guard<std::mutex> l(mu);
            stop = true;
        }
        cv.notify_all();
        for (auto& t : th) t.join();
    }
    void drain() {
        for (;;) {
            long i = next.fetch_add(1);
            if (i >= total) break;
            job(i);
        }
    }
    void loop() {
        long seen = 0;
        for (;;) {
            {
                std::unique_lock<std::mutex> l(mu);
                cv.wait(l, [&] { return stop || generation != seen; });
                if (stop) return;
                seen = generation;
                ++active;
            }
            drain();
            {
                std::lock_guard<std::mutex> l(mu);
                --active;
            }
            done_cv.notify_all();
        }
    }
    void run(long n, std::function<void(long)> f) {
        {
            std::lock_guard<std::mutex> l(mu);
            job = std::move(f);
            total = n;
            next = 0;
            ++generation;
        }
        cv.notify_all();
        drain();   // the launching thread works too
        std::unique_lock<std::mutex> l(mu);
        done_cv.wait(l, [&] { return active == 0 && next.load() >= total; });
        total = 0;  // late wakers find nothing to do
    }
};
inline void launch(dim3 g, dim3 b, const std::function<void()>& body) {
    long total = (long)g.x * g.y * g.z;
    auto one = [&](long i) {
        dim3 bid((unsigned)(i % g.x), (unsigned)((i / g.x) % g.y), (unsigned)(i / ((long)g.x * g.y)));
        run_block(bid, b, g, body);
    };
    static const int nw = workers();
    if (total <= 2 || nw <= 1) {   // tiny grids: run inline, no hand-off latency
        for (long i = 0; i < total; ++i) one(i);
        return;
    }
    static Pool pool(nw - 1);
    pool.run(total, one);
}

inline int lane() { return st().cur->lin & 63; }
inline Wave& wave() { return st().waves[st().cur->lin >> 6]; }
inline int wave_lanes() {
    State& s = st();
    int w = s.cur->lin >> 6;
    return std::min(64, s.nthreads - w * 64);
}
template <class T>
inline T shfl(T v, int src) {
    static_assert(sizeof(T) == 4, "32-bit shuffles only");
    Wave& w = wave();
    int l = lane(), n = wave_lanes();
    memcpy(&w.xa[l], &v, 4);
    bar_wait(w.bar, n);
    T r = v;
    if (src >= 0 && src < n) memcpy(&r, &w.xa[src], 4);
    bar_wait(w.bar, n);
    return r;
}
}  // namespace emu

#define threadIdx (::emu::st().cur->tid)
#define blockIdx (::emu::st().bid)
#define blockDim (::emu::st().bdim)
#define gridDim (::emu::st().gdim)
#define hipLaunchKernelGGL(k, g, b, sh, strm, ...) ::emu::launch((g), (b), [&]() { k(__VA_ARGS__); })

static inline double __dmul_rn(double a, double b) { return a * b; }
static inline double __dadd_rn(double a, double b) { return a + b; }
static inline void __builtin_amdgcn_sched_barrier(int) {}
static inline void __builtin_amdgcn_sched_group_barrier(int, int, int) {}
static inline long long clock64() { return 0; }
static inline long long wall_clock64() { return 0; }
static inline void __syncthreads() { ::emu::bar_wait(::emu::st().block_bar, ::emu::st().live); }
static inline uint32_t __umulhi(uint32_t a, uint32_t b) { return (uint32_t)(((uint64_t)a * b) >> 32); }
template <class T>
static inline T __shfl_xor(T v, int m) {
    return ::emu::shfl(v, ::emu::lane() ^ m);
}
template <class T>
static inline T __shfl_down(T v, int off) {
    return ::emu::shfl(v, ::emu::lane() + off);
}
static inline float atomicAdd(float* p, float v) {
    uint32_t* ip = (uint32_t*)p;
    uint32_t old = __atomic_load_n(ip, __ATOMIC_RELAXED), nw;
    float f;
    do {
        memcpy(&f, &old, 4);
        float s = f + v;
        memcpy(&nw, &s, 4);
    } while (!__atomic_compare_exchange_n(ip, &old, nw, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED));
    return f;
}
static inline unsigned atomicOr(unsigned* p, unsigned v) { return __atomic_fetch_or(p, v, __ATOMIC_RELAXED); }
// ---- bf16 (round-to-nearest-even like v_cvt_pk_bf16_f32) and v_mfma_f32_32x32x16_bf16 ----
struct bf16x8 {
    uint16_t v[8];
};
static inline uint16_t emu_f2bf(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
static inline float emu_bf2f(uint16_t h) {
    uint32_t u = (uint32_t)h << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}
static inline uint32_t pack_bf16x2(float lo_elem, float hi_elem) {
    return (uint32_t)emu_f2bf(lo_elem) | ((uint32_t)emu_f2bf(hi_elem) << 16);
}
static inline float __uint_as_float(uint32_t u) {
    float f;
    memcpy(&f, &u, 4);
    return f;
}
// A: lane l holds A[i = l&31][k = 8*(l>>5) + e], B: lane l holds B[k = 8*(l>>5) + e][j = l&31], e = 0..7
static inline f32x16 __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf16x8 a, bf16x8 b, f32x16 c, int, int, int) {
    static thread_local uint16_t xa[8][64][8], xb[8][64][8];   // up to eight waves per block
    ::emu::Wave& w = ::emu::wave();
    int l = ::emu::lane();
    int wid = ::emu::st().cur->lin >> 6;
    memcpy(xa[wid][l], a.v, 16);
    memcpy(xb[wid][l], b.v, 16);
    ::emu::bar_wait(w.bar, 64);
    const int j = l & 31, hi = l >> 5;
    for (int r = 0; r < 16; ++r) {
        int i = (r & 3) + 8 * (r >> 2) + 4 * hi;
        float acc = c[r];
        for (int h = 0; h < 2; ++h)
            for (int e = 0; e < 8; ++e) acc = fmaf(emu_bf2f(xa[wid][h * 32 + i][e]), emu_bf2f(xb[wid][h * 32 + j][e]), acc);
        c[r] = acc;
    }
    ::emu::bar_wait(w.bar, 64);
    return c;
}
static inline f32x16 __builtin_amdgcn_mfma_f32_32x32x2f32(float a, float b, f32x16 c, int, int, int) {
    ::emu::Wave& w = ::emu::wave();
    int l = ::emu::lane();
    memcpy(&w.xa[l], &a, 4);
    memcpy(&w.xb[l], &b, 4);
    ::emu::bar_wait(w.bar, 64);
    const int j = l & 31, hi = l >> 5;
    for (int r = 0; r < 16; ++r) {
        int i = (r & 3) + 8 * (r >> 2) + 4 * hi;
        float acc = c[r];
        for (int k = 0; k < 2; ++k) {
            float av, bv;
            memcpy(&av, &w.xa[k * 32 + i], 4);
            memcpy(&bv, &w.xb[k * 32 + j], 4);
            acc = fmaf(av, bv, acc);
        }
        c[r] = acc;
    }
    ::emu::bar_wait(w.bar, 64);
    return c;
}
