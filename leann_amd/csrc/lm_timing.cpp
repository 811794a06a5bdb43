// lm_timing.cpp -- measurement facility: HIP event pairs around the library's own launches of the encoder's MFMA kernels, recorded
// on the stream the kernel is launched on (so they work whoever drives the forward: the one-call C++ forwards, the built-in recompute
// provider inside the search loop, the per-kernel Python path).  Off by default (a disabled scope is one relaxed atomic load).
// bench.py switches the dominant kernel's bit on for the timed region: `roofline.achieved` is that kernel's algorithmic flops over
// the sum of its event-pair durations, measured live on the product's default path (library-side provider, no interpreter in the loop).
// Completed pairs are folded into the accumulators as new ones are recorded (hipEventQuery), so a long run keeps a handful of events.
#include <atomic>
#include <deque>
#include <mutex>
#include <vector>

#include "lm_internal.h"

namespace lm {
namespace {
struct Pair {
    hipEvent_t a, b;
    int kid;
    double work;
};
struct Acc {
    int64_t launches = 0;
    double ms = 0, work = 0;
};
std::atomic<unsigned> g_mask{0};
std::mutex g_mu;
std::deque<Pair> g_pending;
std::vector<hipEvent_t> g_free;
Acc g_acc[LM_KT_COUNT];
const char* const g_names[LM_KT_COUNT] = {"lm::k_layer_tail_h384", "lm::k_gemm_ws_h384", "lm::k_attn_varlen", "lm::k_gemm_f16"};

hipEvent_t get_event() {
    if (!g_free.empty()) {
        hipEvent_t e = g_free.back();
        g_free.pop_back();
        return e;
    }
    hipEvent_t e = nullptr;
    return hipEventCreate(&e) == hipSuccess ? e : nullptr;
}
void fold(const Pair& p) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) {
        g_acc[p.kid].launches += 1;
        g_acc[p.kid].ms += ms;
        g_acc[p.kid].work += p.work;
    }
    g_free.push_back(p.a);
    g_free.push_back(p.b);
}
}  // namespace

KtScope::KtScope(int kid_, void* stream, double work_) : kid(kid_), st(stream), work(work_), a(nullptr) {
    if (!(g_mask.load(std::memory_order_relaxed) & (1u << kid))) return;
    std::lock_guard<std::mutex> lk(g_mu);
    while (!g_pending.empty() && hipEventQuery(g_pending.front().b) == hipSuccess) {  // completed pairs -> accumulators
        fold(g_pending.front());
        g_pending.pop_front();
    }
    hipEvent_t e = get_event();
    if (e && hipEventRecord(e, (hipStream_t)st) == hipSuccess) a = e;
    else if (e) g_free.push_back(e);
}
KtScope::~KtScope() {
    if (!a) return;
    std::lock_guard<std::mutex> lk(g_mu);
    hipEvent_t b = get_event();
    if (b && hipEventRecord(b, (hipStream_t)st) == hipSuccess) g_pending.push_back(Pair{(hipEvent_t)a, b, kid, work});
    else {
        g_free.push_back((hipEvent_t)a);
        if (b) g_free.push_back(b);
    }
}
}  // namespace lm

extern "C" int lm_kernel_timing_enable(uint32_t mask) {
    lm::g_mask.store(mask & ((1u << LM_KT_COUNT) - 1), std::memory_order_relaxed);
    return LM_OK;
}

extern "C" int lm_kernel_timing_read(lm_kernel_time* out, int32_t reset) {
    using namespace lm;
    if (!out) LM_FAIL(LM_EINVAL, "lm_kernel_timing_read: NULL argument");
    std::lock_guard<std::mutex> lk(g_mu);
    while (!g_pending.empty()) {  // waits for what has been recorded so far
        (void)hipEventSynchronize(g_pending.front().b);
        fold(g_pending.front());
        g_pending.pop_front();
    }
    for (int i = 0; i < LM_KT_COUNT; ++i) {
        out[i].name = g_names[i];
        out[i].launches = g_acc[i].launches;
        out[i].ms = g_acc[i].ms;
        out[i].work = g_acc[i].work;
        if (reset) g_acc[i] = Acc{};
    }
    return LM_OK;
}
