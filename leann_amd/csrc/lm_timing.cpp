// lm_timing.cpp -- measurement facility: HIP event pairs around the library's own launches of the encoder's MFMA kernels, recorded
// on the stream the kernel is launched on (so they work whoever drives the forward: the one-call C++ forwards, the built-in recompute
// provider inside the search loop, the per-kernel Python path).  Off by default (a disabled scope is one relaxed atomic load).
// bench.py switches the dominant kernel's bit on for the timed region: `roofline.achieved` is that kernel's algorithmic flops over
// the sum of its event-pair durations, measured live on the product's default path (library-side provider, no interpreter in the loop).
// Completed pairs are folded into the accumulators as new ones are recorded (hipEventQuery), so a long run keeps a handful of events.
//
// Events belong to the device that was current when they were created, so the free lists are PER DEVICE and a pair carries its device
// (one process may drive several GPUs); a failed record is dropped and the HIP last-error cleared, so that the launch that follows
// does not report it.  Attention's flops depend on the sequence lengths, which only the device knows: the launcher adds
// 4 H sum(len^2) to a per-device 64-bit counter with a one-workgroup kernel OUTSIDE its event pair (kt_attn_work; only while the
// attention bit is on), lm_kernel_timing_read collects the counters.
#include <algorithm>
#include <atomic>
#include <deque>
#include <map>
#include <mutex>
#include <vector>

#include "lm_internal.h"

namespace lm {
namespace {
struct Pair {
    hipEvent_t a, b;
    int kid, dev;
    double work;
    void* stream;
};
struct Acc {
    int64_t launches = 0;
    double ms = 0, work = 0;
};
struct DevState {
    std::vector<hipEvent_t> free_events;
    // the END event of the last folded pair and its stream: a pair's START event may be stamped while the previous instrumented kernel
    // of the same stream is still running (nothing orders the record behind it), so back-to-back launches of short kernels would be
    // counted twice where they overlap -- round 4's C5 line summed 120.7 s of pairs inside a 115.8 s region.  A pair is therefore
    // charged from max(its start, the previous pair's end).
    hipEvent_t last_end = nullptr;
    void* last_stream = nullptr;
    unsigned long long* d_attn_work = nullptr;  // [2]: sum over launches of H * sum(len^2) (x 4 = flops), added by k_kt_attn_work -- [0] the stand-alone
                                                // attention kernels (LM_KT_ATTN), [1] the fused QKV + attention kernel (LM_KT_QKV_ATTN)
};
std::atomic<unsigned> g_mask{0};
std::mutex g_mu;
std::deque<Pair> g_pending;
std::map<int, DevState> g_dev;
Acc g_acc[LM_KT_COUNT];
const char* const g_names[LM_KT_COUNT] = {"lm::k_layer_tail_h384", "lm::k_gemm_ws_h384", "lm::k_attn_varlen", "lm::k_gemm_f16", "lm::k_qkv_h384",
                                         "lm::k_qkv_attn_h384"};

int current_device() {
    int d = 0;
    if (hipGetDevice(&d) != hipSuccess) (void)hipGetLastError();
    return d;
}
hipEvent_t get_event(int dev) {
    auto& fr = g_dev[dev].free_events;
    if (!fr.empty()) {
        hipEvent_t e = fr.back();
        fr.pop_back();
        return e;
    }
    hipEvent_t e = nullptr;
    if (hipEventCreate(&e) == hipSuccess) return e;
    (void)hipGetLastError();
    return nullptr;
}
void fold(const Pair& p) {
    DevState& ds = g_dev[p.dev];
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) {
        float over = 0.f;  // part of [a, b] that lies before the previous pair's end
        if (ds.last_end && ds.last_stream == p.stream && hipEventElapsedTime(&over, p.a, ds.last_end) == hipSuccess && over > 0.f) ms -= std::min(over, ms);
        (void)hipGetLastError();
        g_acc[p.kid].launches += 1;
        g_acc[p.kid].ms += ms;
        g_acc[p.kid].work += p.work;
    } else {
        (void)hipGetLastError();
    }
    ds.free_events.push_back(p.a);
    if (ds.last_end) ds.free_events.push_back(ds.last_end);
    ds.last_end = p.b;  // kept until the next pair of this device has been folded
    ds.last_stream = p.stream;
}

__global__ __launch_bounds__(256) void k_kt_attn_work(const int32_t* __restrict__ cu, int32_t n_seqs, unsigned long long hidden,
                                                      unsigned long long* __restrict__ acc) {
    __shared__ unsigned long long s_part[4];
    unsigned long long v = 0;
    for (int i = threadIdx.x; i < n_seqs; i += 256) {
        const unsigned long long len = (unsigned long long)(cu[i + 1] - cu[i]);
        v += len * len;
    }
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d);
    if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(acc, (s_part[0] + s_part[1] + s_part[2] + s_part[3]) * hidden);
}
}  // namespace

KtScope::KtScope(int kid_, void* stream, double work_) : kid(kid_), st(stream), work(work_), a(nullptr), dev(0) {
    if (!(g_mask.load(std::memory_order_relaxed) & (1u << kid))) return;
    std::lock_guard<std::mutex> lk(g_mu);
    while (!g_pending.empty() && hipEventQuery(g_pending.front().b) == hipSuccess) {  // completed pairs -> accumulators
        fold(g_pending.front());
        g_pending.pop_front();
    }
    (void)hipGetLastError();  // (hipEventQuery's hipErrorNotReady is a status, not a failure of the caller's launch)
    dev = current_device();
    hipEvent_t e = get_event(dev);
    if (e && hipEventRecord(e, (hipStream_t)st) == hipSuccess) a = e;
    else if (e) {
        (void)hipGetLastError();
        g_dev[dev].free_events.push_back(e);
    }
}
KtScope::~KtScope() {
    if (!a) return;
    std::lock_guard<std::mutex> lk(g_mu);
    hipEvent_t b = get_event(dev);
    if (b && hipEventRecord(b, (hipStream_t)st) == hipSuccess) g_pending.push_back(Pair{(hipEvent_t)a, b, kid, dev, work, st});
    else {
        (void)hipGetLastError();
        g_dev[dev].free_events.push_back((hipEvent_t)a);
        if (b) g_dev[dev].free_events.push_back(b);
    }
}

// attention launchers, in front of their KtScope: the launch's flops / 4 onto the current device's counter (no-op while the bit is off)
void kt_attn_work(const int32_t* d_cu_seqlens, int32_t n_seqs, int32_t hidden, void* stream, int kid) {
    if (!(g_mask.load(std::memory_order_relaxed) & (1u << kid))) return;
    unsigned long long* acc;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        DevState& ds = g_dev[current_device()];
        if (!ds.d_attn_work) {
            if (hipMalloc((void**)&ds.d_attn_work, 16) != hipSuccess || hipMemset(ds.d_attn_work, 0, 16) != hipSuccess) {
                (void)hipGetLastError();
                ds.d_attn_work = nullptr;
                return;
            }
        }
        acc = ds.d_attn_work + (kid == LM_KT_QKV_ATTN ? 1 : 0);
    }
    hipLaunchKernelGGL(k_kt_attn_work, dim3(1), dim3(256), 0, (hipStream_t)stream, d_cu_seqlens, n_seqs, (unsigned long long)hidden, acc);
    (void)hipGetLastError();
}
}  // namespace lm

extern "C" int lm_kernel_timing_enable(uint32_t mask) {
    lm::g_mask.store(mask & ((1u << LM_KT_COUNT) - 1), std::memory_order_relaxed);
    return LM_OK;
}

extern "C" int lm_kernel_timing_read(lm_kernel_time* out, int32_t capacity, int32_t reset) {
    using namespace lm;
    if (!out || capacity < 0) LM_FAIL(LM_EINVAL, "lm_kernel_timing_read: NULL argument / negative capacity");
    std::lock_guard<std::mutex> lk(g_mu);
    while (!g_pending.empty()) {  // waits for what has been recorded so far
        (void)hipEventSynchronize(g_pending.front().b);
        fold(g_pending.front());
        g_pending.pop_front();
    }
    const int before = current_device();
    for (auto& kv : g_dev) {  // attention's device-side flop counters (the kernels that fed them precede the pairs waited for above)
        if (!kv.second.d_attn_work) continue;
        unsigned long long h[2] = {0, 0};
        if (hipSetDevice(kv.first) == hipSuccess && hipMemcpy(h, kv.second.d_attn_work, 16, hipMemcpyDeviceToHost) == hipSuccess) {
            g_acc[LM_KT_ATTN].work += 4.0 * (double)h[0];
            g_acc[LM_KT_QKV_ATTN].work += 4.0 * (double)h[1];
            (void)hipMemset(kv.second.d_attn_work, 0, 16);
        } else {
            (void)hipGetLastError();
        }
    }
    (void)hipSetDevice(before);
    for (int i = 0; i < LM_KT_COUNT; ++i) {
        if (i < capacity) {
            out[i].name = g_names[i];
            out[i].launches = g_acc[i].launches;
            out[i].ms = g_acc[i].ms;
            out[i].work = g_acc[i].work;
        }
        if (reset) g_acc[i] = Acc{};
    }
    return LM_OK;
}
