// lm_internal.h -- shared internals of libleann_mi355x (not part of the public ABI).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <string>
#include <vector>

#include "../../include/leann_mi355x.h"

namespace lm {

void set_error(const std::string& msg);

#define LM_HIP(expr)                                                                       \
    do {                                                                                   \
        hipError_t _e = (expr);                                                            \
        if (_e != hipSuccess) {                                                            \
            lm::set_error(std::string(#expr) + ": " + hipGetErrorString(_e));              \
            return LM_EHIP;                                                                \
        }                                                                                  \
    } while (0)

// opaque to the optimiser: stops loop-invariant code motion of everything computed from `v` (device only; the
// host emulation of tests/hip_emul defines it away)
#ifndef LM_KEEP_LOCAL
#define LM_KEEP_LOCAL(v) asm volatile("" : "+v"(v))
#endif

// kernels that hold ~450 registers per lane by design: tell the compiler not to chase a higher occupancy
#ifndef LM_ONE_WAVE_PER_SIMD
#define LM_ONE_WAVE_PER_SIMD __attribute__((amdgpu_waves_per_eu(1, 1)))
#endif
// kernels budgeted for exactly two waves per SIMD (256 registers each)
#ifndef LM_TWO_WAVES_PER_SIMD
#define LM_TWO_WAVES_PER_SIMD __attribute__((amdgpu_waves_per_eu(2, 2)))
#endif

// Wave-synchronous sections: the 64 lanes of a wave execute in lockstep, so LDS written by one lane is read by another
// lane of the SAME wave in a later instruction without a workgroup barrier.  LM_WAVE_SYNC() marks every such hand-over.
// It expands to nothing in the device build (the validated code is unchanged); the host emulation of tests/hip_emul,
// whose lanes are free-running threads, defines it as a wave barrier -- which is what lets ThreadSanitizer prove that
// these marked points are the ONLY places where the kernels rely on lockstep execution.
#ifndef LM_WAVE_SYNC
#define LM_WAVE_SYNC() ((void)0)
#endif

#define LM_FAIL(code, msg)        \
    do {                          \
        lm::set_error(msg);       \
        return (code);            \
    } while (0)

// hipFuncAttributeMaxDynamicSharedMemorySize is a property of (kernel, DEVICE): a process that searches on a second GPU must raise
// it there too.  One of these per kernel instantiation (a function-local static), keyed by the current device; set again only
// when the request grows.  (Two threads racing on the first launch set the attribute twice: harmless.)
struct DynLdsAttr {
    size_t bytes[32] = {};
};
inline hipError_t ensure_dyn_lds(DynLdsAttr& a, const void* fn, size_t bytes) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0) dev = 0;
    size_t& have = a.bytes[dev & 31];
    if (bytes <= have && dev < 32) return hipSuccess;
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e == hipSuccess) have = bytes;
    return e;
}

// lm_timing.cpp: an event pair around one launch of kernel `kid` (LM_KT_*) on `stream` when that kernel's bit is on in
// lm_kernel_timing_enable's mask; `work` = the launch's algorithmic flops.  Put one in front of the launch: { KtScope kt(...); launch; }
struct KtScope {
    int kid;
    void* st;
    double work;
    void* a;
    int dev;  // device the pair's events belong to
    KtScope(int kid, void* stream, double work);
    ~KtScope();
    KtScope(const KtScope&) = delete;
    KtScope& operator=(const KtScope&) = delete;
};

// attention's algorithmic flops (4 H sum(len^2): the lengths live in device memory) onto the current device's counter; no-op while
// attention's timing bit is off.  Call it in front of the launch's KtScope.
void kt_attn_work(const int32_t* d_cu_seqlens, int32_t n_seqs, int32_t hidden, void* stream, int kid = LM_KT_ATTN);

// ---- query phases (same state machine as oracle/lm_oracle.c) ----
enum : int32_t { PH_SEED = 0, PH_UPPER = 1, PH_BEAM = 2, PH_DONE = 3 };

constexpr uint64_t KEY_NONE = 0xFFFFFFFFFFFFFFFFull;
constexpr uint64_t KEY_EXPANDED = 1ull;

// total-order key (dist, id): NaN -> +inf, -0 -> +0; bit 0 = expanded flag
__host__ __device__ inline uint64_t make_key(float d, int32_t id) {
    if (d != d) d = __builtin_inff();
    if (d == 0.0f) d = 0.0f;
    uint32_t u = __builtin_bit_cast(uint32_t, d);
    u ^= (u >> 31) ? 0xFFFFFFFFu : 0x80000000u;
    return ((uint64_t)u << 32) | ((uint64_t)(uint32_t)id << 1);
}
__host__ __device__ inline float key_dist(uint64_t key) {
    uint32_t u = (uint32_t)(key >> 32);
    u ^= (u >> 31) ? 0x80000000u : 0xFFFFFFFFu;
    return __builtin_bit_cast(float, u);
}
__host__ __device__ inline int32_t key_id(uint64_t key) { return (int32_t)((uint32_t)key >> 1); }

// Parsed compact-CSR file (lm_csr_reader.cpp)
struct HostCsr {
    int64_t ntotal = 0;
    int32_t d = 0;
    int32_t metric = 0;
    int32_t entry_point = -1;
    int32_t max_level = -1;
    std::vector<int32_t> levels;
    std::vector<uint64_t> level_ptr;
    std::vector<uint64_t> node_offsets;
    std::vector<int32_t> neighbors;
    std::vector<float> storage;  // ntotal*d when the file carries flat embeddings
};
int read_csr_file(const char* path, HostCsr& out);  // returns LM_* code

}  // namespace lm

// HBM-resident pre-tokenised passage store (lm_tokens.hip; read by the built-in recompute provider, lm_recompute.hip)
struct lm_tokens {
    int device = 0;
    int64_t n = 0;
    uint64_t total = 0;
    uint16_t* d_tok = nullptr;  // packed token ids of every chunk
    uint64_t* d_off = nullptr;  // n + 1 offsets into d_tok
};

// lm_recompute.hip: the search loop's half of the built-in provider's one-synchronisation round
struct lm_recompute;
namespace lm {
int rc_prepare(lm_recompute* rc, const int32_t* d_ids, const unsigned long long* d_n, int64_t cap, unsigned long long* d_total,
               unsigned long long* d_maxlen, hipStream_t st);
void rc_prepared(lm_recompute* rc, const int32_t* d_ids, int32_t n, int64_t total, int32_t max_len);
int32_t rc_width(const lm_recompute* rc);  // floats per embedding row
// The ONE envelope of the hidden-384 one-call forward (lm_bert_h384_forward_packed, lm_layer_tail_h384_f16) -- shared with
// lm_recompute_create, so that a handle it accepts cannot fail on every search round (round-4 advisor finding).
inline bool bert_h384_envelope_ok(int n_layers, int heads, int ffn) { return n_layers > 0 && heads * 32 == 384 && ffn >= 192 && ffn <= 1728 && ffn % 192 == 0; }
#define LM_BERT_H384_ENVELOPE_TEXT "hidden 384 = heads x 32 and ffn a multiple of 192 in [192, 1728]"
// lm_encoder_forward.cpp: the form of the first half (QKV projection + attention) of a large hidden-384 layer
enum H384FirstHalf : int { H384_FUSED = 0, H384_PAIR_HEAD_MAJOR = 1, H384_PAIR_ROW_MAJOR = 2 };
constexpr int H384_FUSED_MIN_MEAN_LEN = 216;  // mean sequence length from which the fused kernel is the default (see lm_encoder_forward.cpp)
H384FirstHalf h384_first_half_form(int32_t heads, int32_t max_len, int64_t total_tokens, int32_t n_seqs);

}  // namespace lm

// lm_qkv_h384.hip: lm_qkv_h384_f16 with the output layout as a parameter: 0 = [tokens][n_out] (the C ABI's), 1 = head major [n_out / 32][tokens][32]
int lm_qkv_h384_launch(const void* d_x, const void* d_w_img, const float* d_bias, int32_t n_out, void* d_out, int64_t tokens, int32_t head_major, void* stream);

// lm_attn_v3.hip: generation 3 of the head_dim-32 attention kernel (arguments as lm_attn_varlen_hd32_f16; max_len 1..256 checked by the caller).
// total_tokens > 0: qkv is lm_qkv_h384_launch's HEAD-MAJOR layout over that many tokens ([3 x heads][total_tokens][32]); 0 = [tokens][3 x heads x 32]
int lm_attn_v3_launch_hd32(const void* d_qkv, const int32_t* d_cu_seqlens, int32_t n_seqs, int32_t heads, int32_t max_len, void* d_out, int64_t total_tokens,
                           void* stream);

// lm_qkv_attn_h384.hip: the QKV projection fused into attention (hidden 384, 12 heads, lengths 1..256): x [T][384] -> attention output [T][384]
int lm_qkv_attn_h384_launch(const void* d_x, const void* d_wqkv_img, const float* d_bqkv, const int32_t* d_cu_seqlens, int32_t n_seqs, int32_t max_len,
                            int64_t total_tokens, void* d_out, void* stream);

// lm_encoder_ops2.hip: 16-lanes-per-row LayerNorm (opt-in, LEANN_MI355X_LN=2, hidden <= 768); arguments as lm_add_layernorm_f16
int lm_add_layernorm_r16_launch(const void* d_x, const void* d_residual, const void* d_gamma, const void* d_beta, void* d_out,
                                int64_t rows, int32_t hidden, float eps, void* stream);
