// lm_encoder_forward.cpp -- the whole packed BERT forward for hidden 384 as ONE C-ABI call: embedding front end, per layer
// {weight-stationary QKV GEMM, varlen attention, fused layer tail}, mean / CLS pooling.  Host code only: it strings together the
// library's own entry points on one stream, so a search round costs one foreign-function call instead of ~3 L + 2 (at one query per
// call a round recomputes ~5 chunks and the ~20 Python -> ctypes launches are most of its time).  What it replaces in the
// reference: compute_embeddings' model.encode() call (leann/embedding_compute.py:229-239) for sentence-transformers models with
// mean pooling (all-MiniLM-L6-v2 and relatives).  The default launch path of leann_amd/encoder.py since round 3 (MI355X, 200k-chunk
// index: B = 1 p50 57.7 -> 56.3 ms; LEANN_MI355X_ONECALL=0 = one call per kernel); also runs in the thread-per-lane emulation
// (tests/emulated_search_cases.py) and under the real leann.api.LeannSearcher there (tests/real_caller_over_emulation.py).
#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <string>

#include "lm_internal.h"

// tokens up to which the layers run on the general kernels (see lm_bert_h384_layer::wo); LEANN_MI355X_SMALL_TOKENS overrides (0 = never)
static int64_t small_tokens_limit() {  // read per call (one getenv per forward): the Python host reads the same variable per call
    const char* e = getenv("LEANN_MI355X_SMALL_TOKENS");
    return e ? (int64_t)atoll(e) : (int64_t)LM_BERT_SMALL_TOKENS;
}

// Which form the first half of a LARGE hidden-384 layer takes -- decided in ONE place (round-5 advisor: three environment variables were parsed
// inline and the QKV layout depended implicitly on the attention generation):
//   H384_FUSED            lm_qkv_attn_h384_f16: projection fused into attention -- for forwards of LONG sequences (mean length >= 216 of at most 256; 12
//                         heads).  One workgroup per sequence, a 32-row block per wave: its cost per sequence is that of ceil(blocks / 4) rounds, i.e. the same
//                         from 129 to 256 tokens.  Measured (profiles/r6_kbench_fused_qkv_attention_*): 425-436 us against 492-515 us for the pair per 262 k
//                         tokens at length 256, 492-536 against 497-517 at the N(180, 50) lengths of the benchmark corpus -- where the pair stays.
//                         LEANN_MI355X_FUSED_QKV_ATTN=1 forces it at every length, =0 never
//   H384_PAIR_HEAD_MAJOR  lm_qkv_h384 (head-major output) -> lm_attn_v3
//   H384_PAIR_ROW_MAJOR   lm_qkv_h384_f16 / lm_gemm_ws_h384_f16 -> lm_attn_varlen_hd32_f16, [tokens][1152] in between (LEANN_MI355X_QKV_LAYOUT=0, or an
//                         attention generation that reads that layout: LEANN_MI355X_ATTN=2, LEANN_MI355X_ATTN3=9)
lm::H384FirstHalf lm::h384_first_half_form(int32_t heads, int32_t max_len, int64_t total_tokens, int32_t n_seqs) {
    auto is = [](const char* name, char c) {
        const char* e = getenv(name);
        return e && e[0] == c;
    };
    if (heads != 12 || max_len > 256) return H384_PAIR_ROW_MAJOR;
    if (is("LEANN_MI355X_QKV_LAYOUT", '0') || is("LEANN_MI355X_ATTN3", '9') || is("LEANN_MI355X_ATTN", '2')) return H384_PAIR_ROW_MAJOR;
    if (is("LEANN_MI355X_FUSED_QKV_ATTN", '0')) return H384_PAIR_HEAD_MAJOR;
    if (is("LEANN_MI355X_FUSED_QKV_ATTN", '1')) return H384_FUSED;
    return n_seqs > 0 && total_tokens >= (int64_t)H384_FUSED_MIN_MEAN_LEN * n_seqs ? H384_FUSED : H384_PAIR_HEAD_MAJOR;
}
extern "C" int lm_h384_first_half_form(int32_t heads, int32_t max_len, int64_t total_tokens, int32_t n_seqs) {
    return (int)lm::h384_first_half_form(heads, max_len, total_tokens, n_seqs);
}

// tokens up to which a LARGE-form forward takes its QKV projection from the general GEMM (include/leann_mi355x.h: LM_BERT_QKV_GEMM_TOKENS)
static int64_t qkv_gemm_tokens_limit() {
    if (const char* e = getenv("LEANN_MI355X_QKV_GEMM_TOKENS")) return (int64_t)atoll(e);
    const char* s = getenv("LEANN_MI355X_SMALL_TOKENS");
    return s && atoll(s) == 0 ? 0 : (int64_t)LM_BERT_QKV_GEMM_TOKENS;  // SMALL_TOKENS=0: the large-forward kernels at EVERY size (tests, A/B)
}

extern "C" size_t lm_bert_h384_workspace_bytes(int64_t total_tokens) {
    if (total_tokens <= 0) return 0;
    // x, attention output, y: [T][384]; qkv: [T][1152]; small forwards additionally the feed-forward intermediate [T][ffn <= 2560]; fp16
    const size_t small = total_tokens <= std::max<int64_t>(small_tokens_limit(), LM_BERT_SMALL_TOKENS) ? (size_t)total_tokens * 2560 * 2 : 0;
    return (size_t)total_tokens * (3 * 384 + 1152) * 2 + small;
}

extern "C" int lm_bert_h384_forward_packed(const lm_bert_h384* m, const int32_t* d_tok, const int32_t* d_pos, const int32_t* d_cu_seqlens,
                                           int32_t n_seqs, int64_t total_tokens, int32_t max_len, void* d_workspace, size_t workspace_bytes,
                                           float* d_out, void* stream) {
    using namespace lm;
    if (n_seqs == 0 || total_tokens == 0) return LM_OK;
    if (!m || !m->layers || !d_tok || !d_pos || !d_cu_seqlens || !d_workspace || !d_out || n_seqs < 0 || total_tokens < 0)
        LM_FAIL(LM_EINVAL, "lm_bert_h384_forward_packed: bad arguments");
    if (!bert_h384_envelope_ok(m->n_layers, m->heads, m->ffn)) LM_FAIL(LM_EINVAL, "lm_bert_h384_forward_packed: needs " LM_BERT_H384_ENVELOPE_TEXT);
    if (max_len <= 0 || max_len > 256) LM_FAIL(LM_EINVAL, "lm_bert_h384_forward_packed: sequence lengths 1..256");
    if (m->pooling != 0 && m->pooling != 1) LM_FAIL(LM_EINVAL, "lm_bert_h384_forward_packed: pooling 0 (mean) or 1 (CLS)");
    if (workspace_bytes < lm_bert_h384_workspace_bytes(total_tokens)) LM_FAIL(LM_EINVAL, "lm_bert_h384_forward_packed: workspace too small");
    const size_t row = (size_t)total_tokens * 384 * 2;
    unsigned char* ws = (unsigned char*)d_workspace;
    void* x = ws;
    void* a = ws + row;
    void* y = ws + 2 * row;
    void* qkv = ws + 3 * row;
    int rc = lm_embed_layernorm_f16(d_tok, d_pos, m->word, m->pos_table, m->type0, m->emb_gamma, m->emb_beta, x, total_tokens, 384, m->ln_eps, stream);
    if (rc) return rc;
    bool small = total_tokens <= small_tokens_limit() && m->ffn % 128 == 0;  // (every multiple of 192 in the envelope that is one of 128: 384, 768, 1152, 1536, ...)
    for (int l = 0; small && l < m->n_layers; ++l) small = m->layers[l].wo && m->layers[l].w1 && m->layers[l].w2;
    void* hid = ws + 3 * row + (size_t)total_tokens * 1152 * 2;  // [T][ffn], small forwards only (lm_bert_h384_workspace_bytes)
    H384FirstHalf first_half = h384_first_half_form(m->heads, max_len, total_tokens, n_seqs);  // one decision per forward, shared with the Python host's launch path
    bool qkv_gemm = !small && total_tokens <= qkv_gemm_tokens_limit();  // a forward that fills a fraction of the chip: QKV from the general GEMM
    for (int l = 0; qkv_gemm && l < m->n_layers; ++l) qkv_gemm = m->layers[l].wqkv != nullptr;
    {
        const char* f = getenv("LEANN_MI355X_FUSED_QKV_ATTN");
        if (qkv_gemm && f && f[0] == '1') qkv_gemm = false;  // (the fused kernel forced: it has no stand-alone projection)
        else if (qkv_gemm) first_half = H384_PAIR_ROW_MAJOR;
    }
    for (int l = 0; l < m->n_layers; ++l) {
        const lm_bert_h384_layer& L = m->layers[l];
        if (small) {  // every product a grid of small tiles; x -> y (scratch) -> x
            if ((rc = lm_gemm_f16(x, L.wqkv, L.bqkv, nullptr, 0, 1152, 384, qkv, total_tokens, stream))) return rc;
            if ((rc = lm_attn_varlen_hd32_f16(qkv, d_cu_seqlens, n_seqs, m->heads, max_len, a, stream))) return rc;
            if ((rc = lm_gemm_f16(a, L.wo, L.bo, x, 2, 384, 384, y, total_tokens, stream))) return rc;
            if ((rc = lm_add_layernorm_f16(y, nullptr, L.ln1_gamma, L.ln1_beta, x, total_tokens, 384, m->ln_eps, stream))) return rc;
            if ((rc = lm_gemm_f16(x, L.w1, L.b1, nullptr, 1, m->ffn, 384, hid, total_tokens, stream))) return rc;
            if ((rc = lm_gemm_f16(hid, L.w2, L.b2, x, 2, 384, m->ffn, y, total_tokens, stream))) return rc;
            if ((rc = lm_add_layernorm_f16(y, nullptr, L.ln2_gamma, L.ln2_beta, x, total_tokens, 384, m->ln_eps, stream))) return rc;
            continue;
        }
        if (L.wqkv_img && first_half == H384_FUSED) {  // QKV projection fused into attention: Q, K, V never leave the CU (lm_qkv_attn_h384.hip)
            if ((rc = lm_qkv_attn_h384_launch(x, L.wqkv_img, L.bqkv, d_cu_seqlens, n_seqs, max_len, total_tokens, a, stream))) return rc;
        } else if (L.wqkv_img && first_half == H384_PAIR_HEAD_MAJOR) {  // the pair: the projection writes Q / K / V head major, generation 3 of the attention kernel reads contiguous blocks
            if ((rc = lm_qkv_h384_launch(x, L.wqkv_img, L.bqkv, 1152, qkv, total_tokens, 1, stream))) return rc;
            if ((rc = lm_attn_v3_launch_hd32(qkv, d_cu_seqlens, n_seqs, m->heads, max_len, a, total_tokens, stream))) return rc;
        } else {
            if ((rc = qkv_gemm      ? lm_gemm_f16(x, L.wqkv, L.bqkv, nullptr, 0, 1152, 384, qkv, total_tokens, stream)
                      : L.wqkv_img ? lm_qkv_h384_f16(x, L.wqkv_img, L.bqkv, 1152, qkv, total_tokens, stream)
                                   : lm_gemm_ws_h384_f16(x, L.wqkv, L.bqkv, 1152, qkv, total_tokens, stream)))
                return rc;
            if ((rc = lm_attn_varlen_hd32_f16(qkv, d_cu_seqlens, n_seqs, m->heads, max_len, a, stream))) return rc;
        }
        if ((rc = lm_layer_tail_h384_f16(a, x, L.wo_img, L.bo, L.ln1_gamma, L.ln1_beta, m->ln_eps, L.w1_img, L.b1, L.w2_img, L.b2, L.ln2_gamma,
                                         L.ln2_beta, y, total_tokens, m->ffn, m->ln_eps, stream)))
            return rc;
        void* t = x;
        x = y;
        y = t;
    }
    return m->pooling == 1 ? lm_clspool_varlen_f16(x, d_cu_seqlens, n_seqs, 384, m->normalize, d_out, stream)
                           : lm_meanpool_varlen_f16(x, d_cu_seqlens, n_seqs, 384, m->normalize, d_out, stream);
}

// ---- general widths (hidden 768: bge-base, contriever) ---------------------------------------------------------------------------
// The launch sequence of leann_amd/encoder.py: EncoderLayer._forward_packed_general + the pooling kernels, on the C++ side: one call per
// forward for the Python host (B = 1 search on bge-base: ~9 launches per layer x 12 layers through ctypes otherwise) and the forward of
// the built-in recompute provider (lm_recompute_create_general).
static const char* bert_envelope(const lm_bert* m) {
    if (!m || !m->layers || m->n_layers <= 0) return "NULL model / no layers";
    if (m->hidden <= 0 || m->hidden % 128 || m->hidden > 768) return "hidden must be a multiple of 128, <= 768";
    if (m->ffn <= 0 || m->ffn % 128) return "ffn must be a multiple of 128";
    if (m->heads <= 0 || (m->heads * 32 != m->hidden && m->heads * 64 != m->hidden)) return "head_dim = hidden / heads must be 32 or 64";
    if (m->pooling != 0 && m->pooling != 1) return "pooling: 0 = mean, 1 = CLS";
    return nullptr;
}

extern "C" size_t lm_bert_workspace_bytes(const lm_bert* m, int64_t total_tokens) {
    if (!m || total_tokens <= 0) return 0;
    // x, attention output, y: [T][H]; qkv: [T][3H]; feed-forward intermediate: [T][ffn]; fp16
    return (size_t)total_tokens * ((size_t)6 * m->hidden + m->ffn) * 2;
}

extern "C" int lm_bert_forward_packed(const lm_bert* m, const int32_t* d_tok, const int32_t* d_pos, const int32_t* d_cu_seqlens, int32_t n_seqs,
                                      int64_t total_tokens, int32_t max_len, void* d_workspace, size_t workspace_bytes, float* d_out,
                                      void* stream) {
    using namespace lm;
    if (n_seqs == 0 || total_tokens == 0) return LM_OK;
    if (const char* why = bert_envelope(m)) LM_FAIL(LM_EINVAL, std::string("lm_bert_forward_packed: ") + why);
    if (!d_tok || !d_pos || !d_cu_seqlens || !d_workspace || !d_out || n_seqs < 0 || total_tokens < 0)
        LM_FAIL(LM_EINVAL, "lm_bert_forward_packed: bad arguments");
    const int32_t H = m->hidden, hd = H / m->heads;
    if (max_len <= 0 || max_len > (hd == 32 ? 256 : 512)) LM_FAIL(LM_EINVAL, "lm_bert_forward_packed: sequence lengths 1..256 (head_dim 32) / 1..512 (head_dim 64)");
    if (workspace_bytes < lm_bert_workspace_bytes(m, total_tokens)) LM_FAIL(LM_EINVAL, "lm_bert_forward_packed: workspace too small");
    const size_t row = (size_t)total_tokens * H * 2;
    unsigned char* ws = (unsigned char*)d_workspace;
    void *x = ws, *a = ws + row, *y = ws + 2 * row, *qkv = ws + 3 * row, *hid = ws + 6 * row;
    int rc = lm_embed_layernorm_f16(d_tok, d_pos, m->word, m->pos_table, m->type0, m->emb_gamma, m->emb_beta, x, total_tokens, H, m->ln_eps, stream);
    if (rc) return rc;
    for (int l = 0; l < m->n_layers; ++l) {  // x -> y (scratch) -> x
        const lm_bert_layer& L = m->layers[l];
        if ((rc = lm_gemm_f16(x, L.wqkv, L.bqkv, nullptr, 0, 3 * H, H, qkv, total_tokens, stream))) return rc;
        if ((rc = hd == 32 ? lm_attn_varlen_hd32_f16(qkv, d_cu_seqlens, n_seqs, m->heads, max_len, a, stream)
                           : lm_attn_varlen_f16(qkv, d_cu_seqlens, n_seqs, m->heads, hd, max_len, a, stream)))
            return rc;
        if ((rc = lm_gemm_f16(a, L.wo, L.bo, x, 2, H, H, y, total_tokens, stream))) return rc;
        if ((rc = lm_add_layernorm_f16(y, nullptr, L.ln1_gamma, L.ln1_beta, x, total_tokens, H, m->ln_eps, stream))) return rc;
        if ((rc = lm_gemm_f16(x, L.w1, L.b1, nullptr, 1, m->ffn, H, hid, total_tokens, stream))) return rc;
        if ((rc = lm_gemm_f16(hid, L.w2, L.b2, x, 2, H, m->ffn, y, total_tokens, stream))) return rc;
        if ((rc = lm_add_layernorm_f16(y, nullptr, L.ln2_gamma, L.ln2_beta, x, total_tokens, H, m->ln_eps, stream))) return rc;
    }
    return m->pooling == 1 ? lm_clspool_varlen_f16(x, d_cu_seqlens, n_seqs, H, m->normalize, d_out, stream)
                           : lm_meanpool_varlen_f16(x, d_cu_seqlens, n_seqs, H, m->normalize, d_out, stream);
}
