// Execution engine behind tests/hip_emul/full/hip/hip_runtime.h: a launch runs block after block; inside a block every
// lane is an OS thread.  Collectives:
//   __syncthreads()                       barrier over the block's threads
//   __ballot / __shfl / __shfl_up / __shfl_xor(v, m)       whole-wave exchange (all 64 lanes of the wave must call it --
//                                         true everywhere in csrc: wave-uniform control flow around them)
//   __shfl_xor(v, m, width)               exchange inside an aligned group of `width` lanes (only that group must call it:
//                                         csrc uses width 4 under `if (tid < 4)` and width 16 in the row reductions)
// Synchronisation is done with the kernels' own barriers only (see Barrier), so ThreadSanitizer checks their placement.
#pragma once

namespace emul {

struct Idx {
    unsigned x = 0, y = 0, z = 0;
};
struct Tls {
    Idx thread, block, bdim, gdim;
};
inline thread_local Tls tls;

struct Barrier {
    std::atomic<int> count{0}, gen{0};
    int n;
    explicit Barrier(int n_) : n(n_) {}
    void arrive_and_wait() {
        const int g = gen.load(std::memory_order_acquire);
        if (count.fetch_add(1, std::memory_order_acq_rel) + 1 == n) {
            count.store(0, std::memory_order_relaxed);
            gen.fetch_add(1, std::memory_order_release);
            gen.notify_all();
        } else {
            // a few polite spins (a 4- or 16-lane group meets within microseconds), then sleep on the generation word (futex): with 256-1024
            // lane threads on a handful of cores, yielding waiters would eat the time slices the late arrivers need
            for (int spins = 0; spins < 32; ++spins) {
                if (gen.load(std::memory_order_acquire) != g) return;
                std::this_thread::yield();
            }
            while (gen.load(std::memory_order_acquire) == g) gen.wait(g, std::memory_order_acquire);
        }
    }
};

struct Wave {
    Barrier bar;
    std::vector<std::unique_ptr<Barrier>> g4, g16;  // groups of 4 / 16 lanes
    uint64_t slot[64];
    _Float16 ma[64][8], mb[64][8];  // MFMA operand exchange
    explicit Wave(int lanes) : bar(lanes) {
        for (int i = 0; i < 16; ++i) g4.emplace_back(new Barrier(4));
        for (int i = 0; i < 4; ++i) g16.emplace_back(new Barrier(16));
    }
};
struct Block {
    Barrier bar;
    std::vector<std::unique_ptr<Wave>> waves;
    explicit Block(int n) : bar(n) {
        for (int i = 0; i < (n + 63) / 64; ++i) waves.emplace_back(new Wave(std::min(64, n - 64 * i)));
    }
};
inline Block* g_block = nullptr;
alignas(16) inline unsigned char g_dyn_smem[160 * 1024];
inline unsigned char* dyn_smem() { return g_dyn_smem; }

inline Wave& my_wave() { return *g_block->waves[tls.thread.x >> 6]; }
inline int my_lane() { return (int)(tls.thread.x & 63); }
inline void syncthreads() { g_block->bar.arrive_and_wait(); }
inline void wave_sync() { my_wave().bar.arrive_and_wait(); }

template <class T>
inline uint64_t to_bits(T v) {
    static_assert(sizeof(T) <= 8, "shuffle payload");
    uint64_t b = 0;
    std::memcpy(&b, &v, sizeof(T));
    return b;
}
template <class T>
inline T from_bits(uint64_t b) {
    T v;
    std::memcpy(&v, &b, sizeof(T));
    return v;
}
// whole-wave exchange: returns the value published by lane src(lane); own value when src is out of range
template <class T, class F>
inline T wave_exchange(T v, F src_of) {
    Wave& w = my_wave();
    const int lane = my_lane();
    w.slot[lane] = to_bits(v);
    w.bar.arrive_and_wait();
    const int s = src_of(lane);
    const T r = (s >= 0 && s < w.bar.n) ? from_bits<T>(w.slot[s]) : v;
    w.bar.arrive_and_wait();
    return r;
}
template <class T>
inline T shfl_xor(T v, int mask) { return wave_exchange(v, [mask](int l) { return l ^ mask; }); }
template <class T>
inline T shfl_up(T v, int d) { return wave_exchange(v, [d](int l) { return l - d; }); }
template <class T>
inline T shfl_idx(T v, int src) { return wave_exchange(v, [src](int) { return src; }); }
template <class T>
inline T shfl_xor(T v, int mask, int width) {
    if (width >= 64) return shfl_xor(v, mask);
    Wave& w = my_wave();
    const int lane = my_lane();
    Barrier& b = width == 4 ? *w.g4[lane >> 2] : *w.g16[lane >> 4];  // csrc uses widths 4 and 16 only
    if (width != 4 && width != 16) std::abort();
    w.slot[lane] = to_bits(v);
    b.arrive_and_wait();
    const T r = from_bits<T>(w.slot[lane ^ mask]);
    b.arrive_and_wait();
    return r;
}
// v_mfma_f32_32x32x16_f16: A lane l, e -> A[m = l % 32][k = 8 (l / 32) + e]; B lane l, e -> B[k = 8 (l / 32) + e][n = l % 32];
// D lane l, register r -> D[m = (r & 3) + 8 (r >> 2) + 4 (l / 32)][n = l % 32]
template <class AB, class C>
inline C mfma_32x32x16(AB a, AB b, C c) {
    Wave& w = my_wave();
    const int lane = my_lane();
    for (int e = 0; e < 8; ++e) {
        w.ma[lane][e] = a[e];
        w.mb[lane][e] = b[e];
    }
    w.bar.arrive_and_wait();
    const int n = lane & 31, g = lane >> 5;
    C d = c;
    for (int r = 0; r < 16; ++r) {
        const int m = (r & 3) + 8 * (r >> 2) + 4 * g;
        float acc = 0.f;
        for (int k = 0; k < 16; ++k) acc += (float)w.ma[m + 32 * (k >> 3)][k & 7] * (float)w.mb[n + 32 * (k >> 3)][k & 7];
        d[r] = c[r] + acc;
    }
    w.bar.arrive_and_wait();
    return d;
}
// ds_read_b64_tr_b16 (gfx950): every lane names an 8-byte piece (four halfwords); inside each group of 16 lanes, lane l receives
// halfword l % 4 of the pieces named by lanes l / 4 + 4 k (k = 0..3) of its group.  Restated from a MEASUREMENT: scripts/vopbench.cpp's
// probe on an MI355X, profiles/r5_vopbench_instruction_costs.jsonl (all 64 lanes match this rule).
template <class H4>
inline H4 ds_read_tr16_b64(const void* p) {
    Wave& w = my_wave();
    const int lane = my_lane();
    uint64_t piece;
    std::memcpy(&piece, p, 8);
    w.slot[lane] = piece;
    w.bar.arrive_and_wait();
    H4 r;
    for (int k = 0; k < 4; ++k) {
        const uint64_t src = w.slot[(lane & ~15) + ((lane & 15) >> 2) + 4 * k];
        const uint16_t bits = (uint16_t)(src >> (16 * (lane & 3)));
        _Float16 h;
        std::memcpy(&h, &bits, 2);
        r[k] = h;
    }
    w.bar.arrive_and_wait();
    return r;
}
// v_mfma_f32_32x32x8_f16: A lane l, e (0..3) -> A[m = l % 32][k = 4 (l / 32) + e]; B likewise -> B[k][n = l % 32]; D as the 32x32x16 form
template <class AB, class C>
inline C mfma_32x32x8(AB a, AB b, C c) {
    Wave& w = my_wave();
    const int lane = my_lane();
    for (int e = 0; e < 4; ++e) {
        w.ma[lane][e] = a[e];
        w.mb[lane][e] = b[e];
    }
    w.bar.arrive_and_wait();
    const int n = lane & 31, g = lane >> 5;
    C d = c;
    for (int r = 0; r < 16; ++r) {
        const int m = (r & 3) + 8 * (r >> 2) + 4 * g;
        float acc = 0.f;
        for (int k = 0; k < 8; ++k) acc += (float)w.ma[m + 32 * (k >> 2)][k & 3] * (float)w.mb[n + 32 * (k >> 2)][k & 3];
        d[r] = c[r] + acc;
    }
    w.bar.arrive_and_wait();
    return d;
}
inline float med3(float a, float b, float c) { return std::fmax(std::fmin(a, b), std::fmin(std::fmax(a, b), c)); }

inline unsigned long long ballot(bool p) {
    Wave& w = my_wave();
    const int lane = my_lane();
    w.slot[lane] = p ? 1u : 0u;
    w.bar.arrive_and_wait();
    unsigned long long m = 0;
    for (int i = 0; i < w.bar.n; ++i) m |= (unsigned long long)(w.slot[i] & 1u) << i;
    w.bar.arrive_and_wait();
    return m;
}
template <class T, class V>
inline T atomic_or(T* p, V v) { return __atomic_fetch_or(p, (T)v, __ATOMIC_RELAXED); }
template <class T, class V>
inline T atomic_add(T* p, V v) { return __atomic_fetch_add(p, (T)v, __ATOMIC_RELAXED); }
template <class T, class V>
inline T atomic_min(T* p, V v) {
    T old = __atomic_load_n(p, __ATOMIC_RELAXED);
    while ((T)v < old && !__atomic_compare_exchange_n(p, &old, (T)v, true, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {
    }
    return old;
}
inline long long wall_clock() {
    return std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now().time_since_epoch()).count() / 10;
}

// Persistent lane threads (creating 256 OS threads per block dominated the run time).  Worker i runs when ITS mailbox
// go[i] changes, so a 64-lane block wakes 64 workers; the launch fields are published before the mailboxes (release) and
// read after them (acquire).  Idle workers, lanes waiting at a barrier and the launching thread sleep on the word they wait for
// (C++20 atomic wait = futex) after a few polite spins: with 256-1024 lane threads on a handful of cores, yielding waiters ate the
// time slices the late arrivers needed (whole emulated suite 185 s -> 120 s on 8 cores).
struct Pool {
    static constexpr int MAXT = 1024;
    std::vector<std::thread> th;
    std::unique_ptr<std::atomic<int>[]> go{new std::atomic<int>[MAXT]};
    std::atomic<int> done{0};
    std::atomic<bool> stop{false};
    int gen = 0;
    unsigned bx = 0;
    dim3 block, grid;
    const std::function<void()>* body = nullptr;
    Pool() {
        for (int i = 0; i < MAXT; ++i) go[i].store(0);
    }
    void worker(int i) {
        int seen = 0;
        for (;;) {
            int spins = 0, g;
            while ((g = go[i].load(std::memory_order_acquire)) == seen) {
                if (++spins < 64) std::this_thread::yield();
                else go[i].wait(seen, std::memory_order_acquire);  // asleep on its mailbox (futex): an idle worker costs nothing
            }
            seen = g;
            if (g < 0 || stop.load(std::memory_order_relaxed)) return;  // -1 in the mailbox: this worker is being retired
            tls.thread.x = (unsigned)i;
            tls.block.x = bx;
            tls.bdim.x = block.x;
            tls.gdim.x = grid.x;
            (*body)();
            done.fetch_add(1, std::memory_order_release);
            done.notify_one();
        }
    }
    void run_block(unsigned bx_, dim3 grid_, dim3 block_, const std::function<void()>& f) {
        const int n = (int)block_.x;
        if (n > MAXT) std::abort();
        // (Idle workers sleep on their mailboxes: the 1024 lane threads a 1024-wide workgroup -- the PQ traversal, the provider's length
        // scan -- leaves behind cost nothing while later launches use the first 64 ... 256 of them.)
        while ((int)th.size() < n) {
            const int i = (int)th.size();
            th.emplace_back([this, i] { worker(i); });
        }
        bx = bx_;
        grid = grid_;
        block = block_;
        body = &f;
        done.store(0, std::memory_order_relaxed);
        ++gen;
        for (int i = 0; i < n; ++i) {
            go[i].store(gen, std::memory_order_release);
            go[i].notify_one();
        }
        for (int d; (d = done.load(std::memory_order_acquire)) != n;) done.wait(d, std::memory_order_acquire);
    }
    ~Pool() {
        stop.store(true);
        for (size_t i = 0; i < th.size(); ++i) {
            go[i].store(-1, std::memory_order_release);
            go[i].notify_one();
        }
        for (auto& t : th) t.join();
    }
};
inline Pool& pool() {
    static Pool p;
    return p;
}

// one kernel launch: blocks run one after another, lanes of a block as threads (first_block..: run_block() below)
inline void launch(dim3 grid, dim3 block, size_t /*shmem*/, const std::function<void()>& body, unsigned first_block = 0,
                   unsigned n_blocks = ~0u) {
    const int nthreads = (int)block.x;
    for (unsigned bx = first_block; bx < grid.x && bx - first_block < n_blocks; ++bx) {
        Block blk(nthreads);
        g_block = &blk;
        pool().run_block(bx, grid, block, body);
        g_block = nullptr;
    }
}

// a single workgroup of a (conceptually larger) grid
inline void run_block(int block_id, int nthreads, const std::function<void()>& body) {
    launch(dim3((unsigned)block_id + 1), dim3((unsigned)nthreads), 0, body, (unsigned)block_id, 1);
}

}  // namespace emul
