// lm_recompute.hip -- the built-in recompute provider: node ids -> tokens (HBM store) -> packed BERT forward -> fp32 embeddings,
// entirely on the library side of the C ABI.
//
// What it replaces in the reference: one round trip of the embedding-recompute server per search hop -- the ZMQ REQ of the node
// ids, PassageManager lookups, the tokeniser and model.encode() (hnsw_embedding_server.py:148-284, leann/api.py:203-215,
// leann/embedding_compute.py:229-239).  Until round 3 the in-process stand-in for that round trip was a Python callable
// (leann_amd/recompute.py: RecomputeProvider.__call__): per round a ctypes callback into the interpreter, a token gather, three
// small torch ops, a device-to-host copy of the cumulative lengths, a packing kernel and the one-call forward -- at one query per
// call (LEANN's real call shape, leann/api.py:644-796) a round recomputes ~5 chunks and that host work was about half of its time,
// with TWO host synchronisations per round (the search loop's counters, the provider's cumulative lengths).
//
// Here a round is: k_rc_lengths_scan (lengths of the round's chunks + exclusive scan, total and longest length next to the search
// loop's own counters, so the loop's ONE device-to-host copy per round carries them) -> k_rc_pack (token store -> packed token ids /
// positions, no padded intermediate) -> lm_bert_h384_forward_packed.  No interpreter, one synchronisation per round.  It is an
// lm_provider_fn (lm_recompute_provider with user = the handle), so every caller of the provider interface -- the HNSW-style search,
// the PQ traversal's deferred rerank -- takes it unchanged.  Sub-batch bounds (forwards of at most max_tokens_per_forward tokens)
// are those of leann_amd/encoder.py: encode_tokens_packed, so both providers hand the same token batches to the same kernels and
// return bit-identical embeddings (tests/test_gpu_native_provider.py, emulated case native_recompute).
//
// Envelope = that of the one-call forwards (csrc/lm_encoder_forward.cpp): hidden 384 = heads x 32 with mean pooling on the fused kernels
// (lm_recompute_create), or any width the general kernels take -- hidden 768: bge-base, contriever, CLS or mean pooling
// (lm_recompute_create_general).  fp16 weights.  A model outside both keeps the Python provider (a GPU path as well, not a fallback).
#include <algorithm>
#include <climits>
#include <cstdlib>
#include <cstring>

#include "lm_device_types.h"
#include "lm_internal.h"

struct lm_recompute {
    int device = 0;
    lm_bert_h384 model{};                    // hidden 384: the fused kernels (lm_recompute_create)
    std::vector<lm_bert_h384_layer> layers;  // model.layers points here
    bool general = false;                    // any supported width on the general kernels (lm_recompute_create_general)
    lm_bert gmodel{};
    std::vector<lm_bert_layer> glayers;
    int32_t hidden = 384;  // width of an output row
    const lm_tokens* tokens = nullptr;
    int32_t T = 0;           // chunks are truncated to T tokens (min(max_seq_length, position table, longest chunk))
    int64_t max_tokens = 0;  // tokens per forward (sub-batch budget)
    // device buffers, grown on demand, freed with the handle
    int32_t* d_cu = nullptr;      // cumulative lengths of the call's chunks [n + 1]
    int32_t* d_cu_sub = nullptr;  // ... of one forward, starting at 0 [n_sub + 1] (second half of the same allocation)
    int64_t cu_cap = 0;
    int32_t* d_tok = nullptr;  // packed token ids (first half) / positions (second half) of one forward
    int64_t tok_cap = 0;
    void* d_ws = nullptr;  // activations of one forward (lm_bert_h384_workspace_bytes)
    size_t ws_bytes = 0;
    float* d_out = nullptr;  // [n][hidden] fp32: valid until the next call on the same stream
    int64_t out_cap = 0;
    unsigned long long* d_meta = nullptr;  // {total tokens, longest chunk}
    unsigned long long* h_meta = nullptr;  // pinned
    std::vector<int32_t> h_cu;
    // lengths already computed for this id list by the search loop (lm::rc_prepare / lm::rc_prepared): no second synchronisation
    const int32_t* prep_ids = nullptr;
    int32_t prep_n = -1;
    int64_t prep_total = 0;
    int32_t prep_maxlen = 0;
    bool prep_launched = false;
    // statistics
    int64_t chunks = 0, tokens_seen = 0, forwards = 0, calls = 0, syncs = 0;
};

namespace lm {

// Lengths of n chunks (n from device memory when d_n is set: the search loop launches this before it knows the round's count) and
// their exclusive scan: cu[0] = 0, cu[i + 1] = cu[i] + min(T, off[id + 1] - off[id]).  ONE workgroup of 1024 lanes walks the list
// 1024 ids at a time (wave scans by __shfl_up, 16 wave sums through LDS); the next block's lengths are requested before the
// current block is scanned, so the two dependent loads (id, offsets) overlap the scan.  A one-query round has ~5-30 ids: one pass.
__global__ __launch_bounds__(1024) void k_rc_lengths_scan(const uint64_t* __restrict__ off, const int32_t* __restrict__ ids, int32_t n_host,
                                                          const unsigned long long* __restrict__ d_n, int64_t cap, int32_t T,
                                                          int32_t* __restrict__ cu, unsigned long long* __restrict__ out_total,
                                                          unsigned long long* __restrict__ out_maxlen) {
    __shared__ int32_t s_wave[16];
    __shared__ int32_t s_max[16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int64_t n = d_n ? (int64_t)*d_n : (int64_t)n_host;
    if (n > cap) n = cap;  // capacity bound of the id list (the host checks the count after its copy)
    int32_t carry = 0, mx = 0;
    auto length_of = [&](int64_t i) -> int32_t {
        if (i >= n) return 0;
        const int32_t id = ids[i];
        const uint64_t b = off[id], e = off[id + 1];
        return (int32_t)min((uint64_t)T, e - b);
    };
    int32_t len = length_of(tid);
    for (int64_t base = 0; base < n; base += 1024) {
        const int32_t nxt = length_of(base + 1024 + tid);
        int32_t v = len;
        mx = max(mx, len);
        for (int d = 1; d < 64; d <<= 1) {
            const int32_t t = __shfl_up(v, d);
            if (lane >= d) v += t;
        }
        if (lane == 63) s_wave[wave] = v;
        __syncthreads();
        int32_t before = 0, block_total = 0;
#pragma unroll
        for (int w = 0; w < 16; ++w) {
            const int32_t s = s_wave[w];
            before += w < wave ? s : 0;
            block_total += s;
        }
        if (base + tid < n) cu[base + tid + 1] = carry + before + v;
        carry += block_total;
        len = nxt;
        __syncthreads();  // s_wave is rewritten by the next block
    }
    for (int d = 32; d >= 1; d >>= 1) mx = max(mx, __shfl_xor(mx, d));
    if (lane == 0) s_max[wave] = mx;
    __syncthreads();
    if (tid == 0) {
        int32_t m = 0;
        for (int w = 0; w < 16; ++w) m = max(m, s_max[w]);
        cu[0] = 0;
        *out_total = (unsigned long long)(uint32_t)carry;
        *out_maxlen = (unsigned long long)(uint32_t)m;
    }
}

// token store -> packed token ids / positions of the chunks [b0, b0 + n_sub) of the call (one wave per chunk: coalesced u16 reads,
// int32 writes), and the cumulative lengths of this forward starting at 0
__global__ __launch_bounds__(256) void k_rc_pack(const uint16_t* __restrict__ tok, const uint64_t* __restrict__ off,
                                                 const int32_t* __restrict__ ids, const int32_t* __restrict__ cu, int32_t b0, int32_t n_sub,
                                                 int32_t* __restrict__ out_tok, int32_t* __restrict__ out_pos, int32_t* __restrict__ cu_sub) {
    const int w = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (w >= n_sub) return;
    const int32_t first = cu[b0], lo = cu[b0 + w], hi = cu[b0 + w + 1];
    const int32_t base = lo - first, len = hi - lo;
    const uint64_t src = off[ids[b0 + w]];
    for (int j = lane; j < len; j += 64) {
        out_tok[base + j] = (int32_t)tok[src + j];
        out_pos[base + j] = j;
    }
    if (lane == 0) {
        cu_sub[w] = base;
        if (w == n_sub - 1) cu_sub[n_sub] = hi - first;
    }
}

template <typename P>
static int rc_grow(lm_recompute* rc, P** p, int64_t* cap, int64_t need, size_t elem, hipStream_t st, int64_t at_least = 1024) {
    if (need <= *cap) return LM_OK;
    if (*p) {
        LM_HIP(hipStreamSynchronize(st));  // earlier work on the stream may still read the old buffer
        rc->syncs++;
        (void)hipFree(*p);
        *p = nullptr;
        *cap = 0;
    }
    const int64_t want = (std::max<int64_t>(need + need / 4, at_least) + 1) & ~(int64_t)1;
    void* v = nullptr;
    LM_HIP(hipMalloc(&v, (size_t)want * elem));
    *p = (P*)v;
    *cap = want;
    return LM_OK;
}

static int rc_ensure_cu(lm_recompute* rc, int64_t n, hipStream_t st) {
    int r = rc_grow(rc, &rc->d_cu, &rc->cu_cap, 2 * (n + 1), 4, st);  // the call's list, then one forward's
    rc->d_cu_sub = rc->d_cu + rc->cu_cap / 2;
    return r;
}

// The search loop's half of the one-synchronisation round: launch the length scan over the round's unique list BEFORE the count is
// known on the host (d_n = the device counter, cap = capacity of the list), total / longest length into the loop's own counter block.
int rc_prepare(lm_recompute* rc, const int32_t* d_ids, const unsigned long long* d_n, int64_t cap, unsigned long long* d_total,
               unsigned long long* d_maxlen, hipStream_t st) {
    rc->prep_ids = nullptr;
    rc->prep_n = -1;
    rc->prep_launched = false;
    if (cap * (int64_t)rc->T > (int64_t)INT32_MAX) return LM_OK;  // cumulative lengths are int32: such a round is left to the provider's own scan
    int r = rc_ensure_cu(rc, cap, st);
    if (r) return r;
    hipLaunchKernelGGL(k_rc_lengths_scan, dim3(1), dim3(1024), 0, st, rc->tokens->d_off, d_ids, 0, d_n, cap, rc->T, rc->d_cu, d_total, d_maxlen);
    LM_HIP(hipGetLastError());
    rc->prep_launched = true;
    return LM_OK;
}

// ... and, after the loop's copy + synchronisation: what the counters said
void rc_prepared(lm_recompute* rc, const int32_t* d_ids, int32_t n, int64_t total, int32_t max_len) {
    if (!rc->prep_launched) return;
    rc->prep_launched = false;
    rc->prep_ids = d_ids;
    rc->prep_n = n;
    rc->prep_total = total;
    rc->prep_maxlen = max_len;
}

static int rc_forward(lm_recompute* rc, const int32_t* d_ids, int32_t b0, int32_t n_sub, int64_t total, int32_t max_len, float* d_out,
                      hipStream_t st) {
    int r;
    if ((r = rc_grow(rc, &rc->d_tok, &rc->tok_cap, 2 * total, 4, st, (int64_t)1 << 18))) return r;  // ids in the first half, positions in the second
    int32_t* d_pos = rc->d_tok + rc->tok_cap / 2;
    const size_t need = rc->general ? lm_bert_workspace_bytes(&rc->gmodel, total) : lm_bert_h384_workspace_bytes(total);
    if (need > rc->ws_bytes) {
        if (rc->d_ws) {
            LM_HIP(hipStreamSynchronize(st));
            rc->syncs++;
            (void)hipFree(rc->d_ws);
            rc->d_ws = nullptr;
            rc->ws_bytes = 0;
        }
        const size_t want = std::max<size_t>(need + need / 4, (size_t)64 << 20);  // 64 MB: forwards of up to ~13k tokens without growth
        LM_HIP(hipMalloc(&rc->d_ws, want));
        rc->ws_bytes = want;
    }
    hipLaunchKernelGGL(k_rc_pack, dim3((unsigned)((n_sub + 3) / 4)), dim3(256), 0, st, rc->tokens->d_tok, rc->tokens->d_off, d_ids, rc->d_cu, b0,
                       n_sub, rc->d_tok, d_pos, rc->d_cu_sub);
    LM_HIP(hipGetLastError());
    rc->forwards++;
    if (rc->general)
        return lm_bert_forward_packed(&rc->gmodel, rc->d_tok, d_pos, rc->d_cu_sub, n_sub, total, max_len, rc->d_ws, rc->ws_bytes, d_out, (void*)st);
    return lm_bert_h384_forward_packed(&rc->model, rc->d_tok, d_pos, rc->d_cu_sub, n_sub, total, max_len, rc->d_ws, rc->ws_bytes, d_out,
                                       (void*)st);
}

// embeddings of n chunks into d_out [n][hidden]
static int rc_embed(lm_recompute* rc, const int32_t* d_ids, int32_t n, float* d_out, hipStream_t st) {
    int r;
    rc->calls++;
    if (n == 0) return LM_OK;
    int64_t total;
    int32_t max_len;
    if (rc->prep_ids == d_ids && rc->prep_n == n) {  // the search loop computed the lengths of exactly this list
        total = rc->prep_total;
        max_len = rc->prep_maxlen;
    } else {
        if ((int64_t)n * rc->T > (int64_t)INT32_MAX) LM_FAIL(LM_EINVAL, "recompute provider: more than 2^31 tokens in one call");
        if ((r = rc_ensure_cu(rc, n, st))) return r;
        hipLaunchKernelGGL(k_rc_lengths_scan, dim3(1), dim3(1024), 0, st, rc->tokens->d_off, d_ids, n, (const unsigned long long*)nullptr,
                           (int64_t)n, rc->T, rc->d_cu, rc->d_meta, rc->d_meta + 1);
        LM_HIP(hipGetLastError());
        LM_HIP(hipMemcpyAsync(rc->h_meta, rc->d_meta, 16, hipMemcpyDeviceToHost, st));
        LM_HIP(hipStreamSynchronize(st));
        rc->syncs++;
        total = (int64_t)rc->h_meta[0];
        max_len = (int32_t)rc->h_meta[1];
    }
    rc->prep_ids = nullptr;
    rc->prep_n = -1;
    rc->chunks += n;
    rc->tokens_seen += total;
    if (total == 0) {  // every chunk empty: mean pooling of nothing is the zero vector
        LM_HIP(hipMemsetAsync(d_out, 0, (size_t)n * rc->hidden * 4, st));
        return LM_OK;
    }
    if (total <= rc->max_tokens) return rc_forward(rc, d_ids, 0, n, total, max_len, d_out, st);
    // more tokens than one forward takes: sub-batch bounds by cumulative token count (encoder.py: encode_tokens_packed) need the
    // cumulative lengths on the host -- one more copy, next to forwards of ~a million tokens each
    rc->h_cu.resize((size_t)n + 1);
    LM_HIP(hipMemcpyAsync(rc->h_cu.data(), rc->d_cu, ((size_t)n + 1) * 4, hipMemcpyDeviceToHost, st));
    LM_HIP(hipStreamSynchronize(st));
    rc->syncs++;
    const int32_t* cs = rc->h_cu.data();
    int32_t b0 = 0;
    while (b0 < n) {
        // first j with cs[j + 1] > cs[b0] + max_tokens (searchsorted(..., side="right")), at least one chunk per forward
        const int64_t limit = (int64_t)cs[b0] + rc->max_tokens;
        int32_t j = (int32_t)(std::upper_bound(cs + 1, cs + n + 1, limit, [](int64_t lim, int32_t v) { return lim < (int64_t)v; }) - (cs + 1));
        j = std::min(std::max(j, b0 + 1), n);
        int32_t ml = 0;
        for (int32_t i = b0; i < j; ++i) ml = std::max(ml, cs[i + 1] - cs[i]);
        const int64_t sub_total = (int64_t)cs[j] - cs[b0];
        if (sub_total > 0) {
            if ((r = rc_forward(rc, d_ids, b0, j - b0, sub_total, ml, d_out + (size_t)b0 * rc->hidden, st))) return r;
        } else {
            LM_HIP(hipMemsetAsync(d_out + (size_t)b0 * rc->hidden, 0, (size_t)(j - b0) * rc->hidden * 4, st));
        }
        b0 = j;
    }
    return LM_OK;
}

int32_t rc_width(const lm_recompute* rc) { return rc->hidden; }

static int rc_finish_create(lm_recompute* rc, const lm_tokens* tokens, int32_t max_seq_len, int64_t max_tokens, lm_recompute** out) {
    rc->device = tokens->device;
    rc->tokens = tokens;
    rc->T = max_seq_len;
    rc->max_tokens = max_tokens;
    if (const char* ft = getenv("LEANN_MI355X_FORWARD_TOKENS"))  // A/B: a forward's token budget whatever the caller asked for (scripts/sessions/r6_29.sh)
        if (atoll(ft) > 0) rc->max_tokens = atoll(ft);
    hipError_t e;
    if ((e = hipMalloc((void**)&rc->d_meta, 16)) != hipSuccess || (e = hipHostMalloc((void**)&rc->h_meta, 16)) != hipSuccess) {
        set_error(std::string("lm_recompute_create: ") + hipGetErrorString(e));
        lm_recompute_free(rc);
        return LM_EHIP;
    }
    *out = rc;
    return LM_OK;
}

}  // namespace lm

extern "C" {

int lm_recompute_create(const lm_bert_h384* model, const lm_tokens* tokens, int32_t max_seq_len, int64_t max_tokens_per_forward,
                        lm_recompute** out) {
    using namespace lm;
    if (!out) LM_FAIL(LM_EINVAL, "out is NULL");
    *out = nullptr;
    if (!model || !model->layers || !tokens) LM_FAIL(LM_EINVAL, "lm_recompute_create: NULL model / token store");
    if (!bert_h384_envelope_ok(model->n_layers, model->heads, model->ffn))
        LM_FAIL(LM_EINVAL, "lm_recompute_create: needs " LM_BERT_H384_ENVELOPE_TEXT " (the one-call forward's envelope; other shapes: lm_recompute_create_general)");
    if (max_seq_len <= 0 || max_seq_len > 256) LM_FAIL(LM_EINVAL, "lm_recompute_create: chunk length limit must be 1..256 tokens");
    if (max_tokens_per_forward <= 0) LM_FAIL(LM_EINVAL, "lm_recompute_create: max_tokens_per_forward must be positive");
    LM_HIP(hipSetDevice(tokens->device));
    lm_recompute* rc = new lm_recompute();
    rc->model = *model;
    rc->layers.assign(model->layers, model->layers + model->n_layers);
    rc->model.layers = rc->layers.data();
    return rc_finish_create(rc, tokens, max_seq_len, max_tokens_per_forward, out);
}

int lm_recompute_create_general(const lm_bert* model, const lm_tokens* tokens, int32_t max_seq_len, int64_t max_tokens_per_forward,
                                lm_recompute** out) {
    using namespace lm;
    if (!out) LM_FAIL(LM_EINVAL, "out is NULL");
    *out = nullptr;
    if (!model || !model->layers || !tokens) LM_FAIL(LM_EINVAL, "lm_recompute_create_general: NULL model / token store");
    if (model->n_layers <= 0 || model->hidden <= 0 || model->hidden % 128 || model->hidden > 768 || model->ffn <= 0 || model->ffn % 128 ||
        model->heads <= 0 || (model->heads * 32 != model->hidden && model->heads * 64 != model->hidden) || (model->pooling != 0 && model->pooling != 1))
        LM_FAIL(LM_EINVAL, "lm_recompute_create_general: outside lm_bert_forward_packed's envelope (hidden % 128 == 0 and <= 768, ffn % 128 == 0, "
                           "head_dim 32 or 64, mean or CLS pooling)");
    if (max_seq_len <= 0 || max_seq_len > (model->heads * 32 == model->hidden ? 256 : 512))
        LM_FAIL(LM_EINVAL, "lm_recompute_create_general: chunk length limit must be 1..256 tokens (head_dim 32) / 1..512 (head_dim 64)");
    if (max_tokens_per_forward <= 0) LM_FAIL(LM_EINVAL, "lm_recompute_create_general: max_tokens_per_forward must be positive");
    LM_HIP(hipSetDevice(tokens->device));
    lm_recompute* rc = new lm_recompute();
    rc->general = true;
    rc->gmodel = *model;
    rc->glayers.assign(model->layers, model->layers + model->n_layers);
    rc->gmodel.layers = rc->glayers.data();
    rc->hidden = model->hidden;
    return rc_finish_create(rc, tokens, max_seq_len, max_tokens_per_forward, out);
}

void lm_recompute_free(lm_recompute* rc) {
    if (!rc) return;
    (void)hipSetDevice(rc->device);
    for (void* p : {(void*)rc->d_cu, (void*)rc->d_tok, rc->d_ws, (void*)rc->d_out, (void*)rc->d_meta})
        if (p) (void)hipFree(p);
    if (rc->h_meta) (void)hipHostFree(rc->h_meta);
    delete rc;
}

// an lm_provider_fn: user = the lm_recompute handle
int lm_recompute_provider(void* user, const int32_t* d_ids, int32_t n, void** d_out, void* stream) {
    using namespace lm;
    lm_recompute* rc = (lm_recompute*)user;
    if (!rc || !d_out || n < 0 || (n > 0 && !d_ids)) LM_FAIL(LM_EINVAL, "lm_recompute_provider: bad arguments");
    LM_HIP(hipSetDevice(rc->device));  // buffers and launches belong to the handle's device, whatever the caller's current one is
    hipStream_t st = (hipStream_t)stream;
    // first allocation: 4096 rows (6 MB) -- the rounds of a small-batch search grow from one chunk to a few hundred, and every
    // re-allocation costs a synchronisation
    int r = rc_grow(rc, &rc->d_out, &rc->out_cap, (int64_t)std::max(n, 1) * rc->hidden, 4, st, (int64_t)4096 * rc->hidden);
    if (r) return r;
    *d_out = rc->d_out;
    return rc_embed(rc, d_ids, n, rc->d_out, st);
}

int lm_recompute_embed(lm_recompute* rc, const int32_t* d_ids, int32_t n, float* d_out, void* stream) {
    using namespace lm;
    if (!rc || n < 0 || (n > 0 && (!d_ids || !d_out))) LM_FAIL(LM_EINVAL, "lm_recompute_embed: bad arguments");
    LM_HIP(hipSetDevice(rc->device));
    return rc_embed(rc, d_ids, n, d_out, (hipStream_t)stream);
}

int lm_recompute_get_stats(const lm_recompute* rc, lm_recompute_stats* out) {
    using namespace lm;
    if (!rc || !out) LM_FAIL(LM_EINVAL, "lm_recompute_get_stats: NULL argument");
    out->calls = rc->calls;
    out->chunks = rc->chunks;
    out->tokens = rc->tokens_seen;
    out->forwards = rc->forwards;
    out->host_syncs = rc->syncs;
    return LM_OK;
}

}  // extern "C"
