// Thread-per-lane execution of one workgroup: every lane of the 256-thread block is an OS thread; wave-collective
// operations (MFMA, cross-lane shuffles) and __syncthreads() are barriers.  The MFMA follows the operand / result
// layout the kernels rely on (validated on hardware by k_attn_varlen_hd32, which this harness also runs):
//   A: lane l, element e -> A[m = l % 32][k = 8 (l / 32) + e]     B: lane l, element e -> B[k = 8 (l / 32) + e][n = l % 32]
//   D: lane l, register r -> D[m = (r & 3) + 8 (r >> 2) + 4 (l / 32)][n = l % 32]
#pragma once
#include <atomic>
#include <functional>
#include <memory>
#include <thread>

namespace emul {

struct Idx {
    int x = 0, y = 0, z = 0;
};
struct Tls {
    Idx thread, block;
};
inline thread_local Tls tls;

// Own barrier instead of std::barrier: libstdc++'s waits go through a global, address-hashed waiter pool, which makes
// ThreadSanitizer see happens-before edges between UNRELATED barriers (a cross-wave race then goes unreported).  This one
// synchronises only through its own two atomics, so `-fsanitize=thread` checks the kernels' __syncthreads() placement.
struct Barrier {
    std::atomic<int> count{0}, gen{0};
    int n;
    explicit Barrier(int n_) : n(n_) {}
    void arrive_and_wait() {
        const int g = gen.load(std::memory_order_acquire);
        if (count.fetch_add(1, std::memory_order_acq_rel) + 1 == n) {
            count.store(0, std::memory_order_relaxed);
            gen.fetch_add(1, std::memory_order_release);
        } else {
            while (gen.load(std::memory_order_acquire) == g) std::this_thread::yield();
        }
    }
};

struct Wave {
    Barrier bar{64};
    _Float16 a[64][8], b[64][8];
    float f[64];
};
struct Block {
    int nthreads;
    Barrier bar;
    std::vector<std::unique_ptr<Wave>> waves;
    explicit Block(int n) : nthreads(n), bar(n) {
        for (int i = 0; i < n / 64; ++i) waves.emplace_back(new Wave());
    }
};
inline Block* g_block = nullptr;

inline Wave& my_wave() { return *g_block->waves[tls.thread.x >> 6]; }
inline void syncthreads() { g_block->bar.arrive_and_wait(); }

template <class AB, class C>
inline C mfma_32x32x16(AB a, AB b, C c) {
    Wave& w = my_wave();
    const int lane = tls.thread.x & 63;
    for (int e = 0; e < 8; ++e) {
        w.a[lane][e] = a[e];
        w.b[lane][e] = b[e];
    }
    w.bar.arrive_and_wait();
    const int n = lane & 31, g = lane >> 5;
    C d = c;
    for (int r = 0; r < 16; ++r) {
        const int m = (r & 3) + 8 * (r >> 2) + 4 * g;
        float acc = 0.f;
        for (int k = 0; k < 16; ++k) acc += (float)w.a[m + 32 * (k >> 3)][k & 7] * (float)w.b[n + 32 * (k >> 3)][k & 7];
        d[r] = c[r] + acc;
    }
    w.bar.arrive_and_wait();
    return d;
}

inline float shfl_xor(float v, int mask) {
    Wave& w = my_wave();
    const int lane = tls.thread.x & 63;
    w.f[lane] = v;
    w.bar.arrive_and_wait();
    const float r = w.f[lane ^ mask];
    w.bar.arrive_and_wait();
    return r;
}

inline float med3(float a, float b, float c) { return std::fmax(std::fmin(a, b), std::fmin(std::fmax(a, b), c)); }

// run one workgroup of `nthreads` lanes: body() is the kernel call
inline void run_block(int block_id, int nthreads, const std::function<void()>& body) {
    Block blk(nthreads);
    g_block = &blk;
    std::vector<std::thread> th;
    for (int t = 0; t < nthreads; ++t)
        th.emplace_back([&, t] {
            tls.thread.x = t;
            tls.block.x = block_id;
            body();
        });
    for (auto& x : th) x.join();
    g_block = nullptr;
}

}  // namespace emul
