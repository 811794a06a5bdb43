/*
 * leann_mi355x.h -- C ABI of libleann_mi355x.so, the MI355X (gfx950) native replacement for the
 * query-time selective-recompute beam search of LEANN.
 *
 * Every entry point names the reference interface it replaces (paths relative to the LEANN tree,
 * packages/...).  The reference reaches its native code through SWIG (faiss fork) and pybind11
 * (DiskANN fork); both forks are un-vendored submodules, so the binding surface is taken from the
 * Python call sites.  Plain C: pointers + sizes, no torch / C++ types.  All functions return 0 on
 * success or a negative LM_E* code; lm_last_error() gives the thread-local message.  Nothing throws
 * across this boundary.  Pointer arguments named d_* are DEVICE (HBM) pointers, everything else is
 * host memory.  A handle is not re-entrant: one search at a time per lm_index.
 */
#ifndef LEANN_MI355X_H
#define LEANN_MI355X_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LM_OK 0
#define LM_EINVAL (-1)   /* bad argument                 -> Python ValueError      */
#define LM_ENOENT (-2)   /* file missing                 -> FileNotFoundError      */
#define LM_EFORMAT (-3)  /* malformed index file         -> ValueError             */
#define LM_EHIP (-4)     /* HIP runtime failure / no GPU -> RuntimeError           */
#define LM_ESTATE (-5)   /* e.g. recompute requested but no provider attached -> RuntimeError */
#define LM_EPROVIDER (-6)/* embedding provider callback failed                     */

#define LM_METRIC_INNER_PRODUCT 0 /* faiss.METRIC_INNER_PRODUCT: "mips", "cosine" (hnsw_backend.py:25-29) */
#define LM_METRIC_L2 1            /* faiss.METRIC_L2 */

#define LM_DTYPE_F32 0
#define LM_DTYPE_F16 1

typedef struct lm_index lm_index;     /* HBM-resident compact-CSR graph + search workspace */
typedef struct lm_tokens lm_tokens;   /* HBM-resident pre-tokenised passage store           */

/* ---- errors / device ------------------------------------------------------------------- */
const char *lm_last_error(void);
int lm_device_count(void);           /* 0 when no HIP device is visible */
const char *lm_version(void);
/* Revision of THIS interface: bumped whenever an entry point, a struct layout or an enum of this header changes incompatibly (6: round 6 --
 * lm_kernel_timing_read takes the caller's capacity; lm_search_params.batch_size has a meaning; LM_KT_COUNT grew in round 5).  A binding
 * compares lm_abi_revision() with the LM_ABI_REVISION it was written against before it calls anything else. */
#define LM_ABI_REVISION 6
int lm_abi_revision(void);

/* ---- index lifetime ---------------------------------------------------------------------
 * Replaces faiss.read_index(str(index_file), faiss.IO_FLAG_MMAP, HNSWIndexConfig{is_compact,
 * is_recompute})                                   leann_backend_hnsw/hnsw_backend.py:145-151.
 * Parses the compact-CSR file written by convert_to_csr.py:182-237 (or the original IHNf layout,
 * :264-301,439-479) and uploads level_ptr / node_offsets / neighbors / levels to HBM.  If the file
 * carries flat embedding storage it is attached as the stored-embedding table. */
int lm_index_read(const char *path, int device, lm_index **out);

/* Same, from arrays already in host memory (layout of convert_to_csr.py:494-548). */
int lm_index_create_from_csr(int64_t ntotal, int32_t d, int32_t metric,
                             const uint64_t *node_offsets /* ntotal+1 */,
                             const uint64_t *level_ptr, int64_t n_level_ptr,
                             const int32_t *neighbors, int64_t n_neighbors,
                             const int32_t *levels /* ntotal */,
                             int32_t entry_point, int32_t max_level, int device, lm_index **out);
void lm_index_free(lm_index *idx);

typedef struct {
    int64_t ntotal;
    int32_t d;
    int32_t d_padded;       /* row stride of embeddings handed to the kernels (multiple of 64) */
    int32_t metric;
    int32_t entry_point;
    int32_t max_level;
    int32_t max_degree0;    /* largest level-0 neighbour list */
    int32_t max_degree_up;  /* largest upper-level neighbour list */
    int64_t n_neighbors;
    int32_t has_table;      /* stored embeddings attached (non-pruned index) */
    int32_t has_provider;
    int32_t device;
} lm_index_info_t;
int lm_index_info(const lm_index *idx, lm_index_info_t *out);

/* ---- embedding sources --------------------------------------------------------------------
 * Stored embeddings (recompute_embeddings=False on a non-pruned index, hnsw_backend.py:189-193):
 * rows of `table` are gathered straight from HBM by the distance kernel.
 * location 0: host pointer, copied (and zero-padded to d_padded) into library-owned HBM;
 * location 1: device pointer with row stride d_padded, borrowed (caller keeps it alive). */
int lm_index_attach_table(lm_index *idx, const void *table, int32_t dtype, int64_t ntotal, int32_t d,
                          int32_t location);

/* Recompute provider: replaces the per-hop ZMQ REQ of the faiss fork to the embedding server
 * ([[ids],[query]] -> distances / [ids] -> embeddings, hnsw_embedding_server.py:148-284).
 * Called once per search round, on the index's stream, with the SORTED, DE-DUPLICATED node ids
 * of that round in device memory.  (One opt-in exception: with the index option "single_query_direct" a ONE-query pass hands over
 * its new-list as it is -- every id once, in discovery order, not sorted.)  The callback must enqueue (on `stream`) work that produces
 * fp32 embeddings [n][d_padded] (zero padded) in device memory and store that buffer's address
 * in *d_out; the buffer must stay valid until the next callback or the end of the search.
 * Return 0 on success; any other value aborts the search with LM_EPROVIDER. */
typedef int (*lm_provider_fn)(void *user, const int32_t *d_ids, int32_t n, void **d_out, void *stream);
int lm_index_set_provider(lm_index *idx, lm_provider_fn fn, void *user);

/* Hub-embedding cache (LEANN paper section 5: caching the embeddings of the highest-degree ~10 % nodes): the rows
 * of d_embeddings [n][d_padded] (fp32, device) are copied into HBM owned by the index; nodes listed in `ids` (host,
 * unique) are never sent to the provider again.  n = 0 clears the cache.  Cf. num_nodes_to_cache of the DiskANN
 * backend (diskann_backend.py:345). */
int lm_index_set_hub_cache(lm_index *idx, const int32_t *ids, int32_t n, const float *d_embeddings);

/* hipStream_t all kernels of this index are enqueued on (default: the null stream). */
int lm_index_set_stream(lm_index *idx, void *hip_stream);

/* ---- search ---------------------------------------------------------------------------------
 * Mirrors faiss.SearchParametersHNSW as filled in hnsw_backend.py:203-234. */
typedef struct {
    int32_t efSearch;                /* complexity                               :206 */
    int32_t beam_size;               /* beam_width: pops per query per round     :207 */
    int32_t check_relative_distance; /* 0 for OpenAI-cosine models               :209-217 */
    float pq_pruning_ratio;          /* prune_ratio: two-level search (paper Alg. 2), needs lm_pq_attach; only the
                                        fraction 1-ratio of the candidates, ranked by PQ-ADC distance, is
                                        recomputed exactly.  0 = off                                    :220 */
    int32_t local_prune;             /* pruning_strategy == "local": rank this hop's neighbours only    :223-225 */
    float send_neigh_times_ratio;    /* > 1e-6: "proportional": quota from this hop's count, taken from the
                                        per-query approximate queue; else "global": everything unconsumed in
                                        the top (1-ratio) fraction of the approximate queue            :226-231 */
    int32_t batch_size;              /* "Neighbor processing batch size" :163,181,234 -- the paper's dynamic batching (section 4.2):
                                        > 0: after its beam_size pops a query keeps popping its best unexpanded candidate, one at a
                                        time and under the same stop rules, while the round's new-list (unvisited neighbours gathered
                                        so far) is shorter than batch_size -- fewer, fuller recompute forwards per search (a one-query
                                        search at efSearch 64: ~75 rounds of ~10 chunks -> ~20 rounds of ~64 at batch_size 64) for a
                                        few per cent more distance evaluations; the pops are chosen on the pool as it stood at the
                                        start of the round.  0 (the reference's default): off, results unchanged bit for bit.
                                        Rounds batch ACROSS queries as well, whatever this says.  oracle/lm_oracle.c restates it. */
    int32_t zmq_port;                /* accepted, unused: the encoder is in-process   :205 */
    int32_t recompute;               /* 1: use the provider, 0: use the attached table */
    int32_t max_batch;               /* queries in flight per pass (0 = default 4096) */
    int32_t recompute_memo;          /* 1 (default): within ONE pass of a search call (<= max_batch queries) every
                                        node is recomputed at most once -- fresh embeddings stay in HBM until the
                                        pass returns (N x d_padded x 4 bytes at most, grown on demand; results are
                                        identical, only nunique drops); a one-query pass skips it (its visited
                                        set already guarantees it).  0: dedup per lock-step round only, nothing
                                        kept between rounds.  Cf. the reference's own dedup_node_dis ("cache and
                                        reuse distance computations", diskann_backend.py:394,413,463). */
} lm_search_params;
void lm_search_params_default(lm_search_params *p);

/* Replaces index.search(n, swig_ptr(x), k, swig_ptr(D), swig_ptr(I), params)
 *                                                           hnsw_backend.py:241-248.
 * x: n x d fp32 queries (host).  distances: n x k fp32, labels: n x k int64, caller-allocated
 * (hnsw_backend.py:236-238).  l2: squared L2 ascending; inner product: +IP descending.
 * Unfilled slots: label -1, distance +inf (l2) / -inf (ip). */
int lm_index_search(lm_index *idx, int64_t n, const float *x, int32_t k, float *distances,
                    int64_t *labels, const lm_search_params *params);
/* Same with queries and results resident in HBM (no PCIe in the timed region). */
int lm_index_search_device(lm_index *idx, int64_t n, const float *d_x, int32_t k, float *d_distances,
                           int64_t *d_labels, const lm_search_params *params);

typedef struct {
    int64_t ndis;         /* (query,node) distance evaluations of the last search            */
    int64_t nunique;      /* embeddings requested from the provider (after cross-query dedup) */
    int64_t nrounds;      /* lock-step rounds                                                 */
    int64_t nexpand;      /* level-0 expansions                                               */
    int64_t update_launches; /* launches of the fused distance+beam-update kernel            */
    double update_ms;     /* summed HIP-event time of those launches (profiling on)          */
    double expand_ms;     /* summed HIP-event time of the expand(+uniq) kernels              */
    double provider_ms;   /* summed HIP-event time spent in provider work                    */
    int64_t nadc;         /* PQ-ADC evaluations of the two-level search (pq_pruning_ratio > 0) */
    double update_span_ms; /* profiling on: summed execution span (max end - min start over workgroups, device
                              wall clock) of the fused update kernel = what rocprofv3 reports as its duration */
    int64_t update_span_launches;
} lm_search_stats;
int lm_index_get_stats(const lm_index *idx, lm_search_stats *out);
int lm_index_set_profiling(lm_index *idx, int32_t enable); /* HIP events around kernels */
/* Mean event-pair time around an empty kernel (us): the fixed part of every update_ms/launch. */
int lm_index_event_overhead_us(lm_index *idx, double *out_us);
/* Tuning knobs (A/B measurements): "update_variant" 0 = auto (default), 3 / 4 = the fused distance + beam-update kernel with a
 * wave / a workgroup per query.  "persistent_table" 1 (default) = stored-embedding searches run as ONE persistent
 * launch per batch, 0 = lock-step rounds; "persistent_wave" -1 (auto) / 0 / 1 = its workgroup- or wave-per-query form.
 * "pq_threads" 256 / 512 / 1024 (default) = workgroup width of the PQ traversal.  "memo_initial_rows" = first allocation of the
 * per-call recompute memo in rows (0 = default: max(65536, 1024 per query of the pass)); it doubles on demand, never beyond N.
 * "pq_rerank_expanded" 0 (default) / 1: which set the DiskANN-style path's exact rerank ranks -- 0 the final candidate list (what
 * diskann_backend.py:444-449 describes: "fetch embeddings for the final candidate set only"), 1 EVERY node the traversal expanded
 * (upstream DiskANN's full_retset, PQFlashIndex::cached_beam_search: a superset of the final list; up to 4 x complexity <= 8192 nodes
 * per query are recorded, a query that expands more falls back to its final list and is counted in "pq_rerank_overflow").
 * "single_query_direct" 0 (default) / 1: a one-query recompute pass without a memo hands its new-list to the provider as it is (unique ids
 * in discovery order; three launches per round fewer: no request bitmap, no unique-list kernels); same results and counts.  Written after
 * round 4's GPU budget was spent: emulation-validated, not yet timed on hardware.
 * "speculate" S (0 = off, <= 64) / "speculate_max_batch" (default 2): speculative prefetch of a small recompute batch -- a round's forward
 * also embeds the unvisited neighbours of the S best candidates that are not expanded yet, into the per-call memo, so that later rounds
 * find their new nodes there and need no forward (a one-query search is ~100 rounds of ~50 dependent launches: launch latency).  Labels,
 * distances and distance-evaluation counts are those of S = 0; only the provider's request lists differ (more ids in fewer calls).
 * lm_index_get_option reads a value back ("pq_rerank_overflow": queries that fell back since the option was last set). */
int lm_index_set_option(lm_index *idx, const char *name, int64_t value);
int lm_index_get_option(const lm_index *idx, const char *name, int64_t *value);

/* ---- DiskANN-style path: PQ-ADC traversal + deferred exact rerank ---------------------------------
 * Replaces _diskannpy.StaticDiskFloatIndex(metric, prefix, threads, cache, mechanism, zmq_port,
 * pq_prefix, partition_prefix)            leann_backend_diskann/diskann_backend.py:371-380
 * and  .batch_search(query, B, top_k, complexity, beam_width, num_threads, use_deferred_fetch,
 * skip_search_reorder, recompute_neighbors, dedup_node_dis, prune_ratio, batch_recompute,
 * use_global_pruning)                                                  diskann_backend.py:453-467.
 * The graph is the lm_index' level-0 graph entered at entry_point (the medoid for a Vamana graph);
 * lm_pq_attach adds the product quantiser: m sub-quantisers x 256 centroids x (d/m) floats and the
 * N x m code bytes (the role of <prefix>_pq_pivots.bin / _pq_compressed.bin).  Traversal uses PQ
 * distances only; with use_deferred_fetch the final candidate list (<= complexity per query) is
 * re-ranked with exact distances of embeddings fetched ONCE through the provider (:444-449); without
 * it, stored embeddings are used when attached, else the PQ order is returned. */
int lm_pq_attach(lm_index *idx, int32_t m, const float *codebooks, const uint8_t *codes, int64_t ntotal);
/* The same with the sub-quantisers' dimension ranges given explicitly -- the chunking of a stock DiskANN bundle
 * (<prefix>_pq_pivots.bin: 256 full-dimension pivots + centroid + chunk_offsets, diskann_backend.py:151-162): chunk j covers
 * dimensions [chunk_offsets[j], chunk_offsets[j+1]), lengths may differ and may be 0, chunk_offsets[m] <= d (trailing dimensions
 * carry no code: the augmentation coordinate of a MIPS bundle).  codebooks: chunk j's 256 centroids x len_j floats at float
 * offset 256 * chunk_offsets[j].  m % 4 == 0 (pad the code rows with empty chunks).  leann_amd/diskann_files.py converts the files. */
int lm_pq_attach_chunked(lm_index *idx, int32_t m, const int32_t *chunk_offsets, const float *codebooks, const uint8_t *codes,
                         int64_t ntotal);
typedef struct {
    int32_t complexity;          /* L: candidate list size                 :456 */
    int32_t beam_width;          /* W: nodes expanded per iteration (<=64) :457 */
    int32_t num_threads;         /* accepted, unused                       :459 */
    int32_t use_deferred_fetch;  /* = recompute_embeddings                 :450,460 */
    int32_t skip_search_reorder; /* return PQ order / distances            :461 */
    int32_t recompute_neighbors; /* must be 0 -- what the reference passes (:451,462); non-zero: LM_EINVAL */
    /* The four knobs below parametrise the PER-HOP neighbour recomputation of the fork's batch_search (which neighbours' exact distances
     * are recomputed, and how those recomputations are cached / batched).  The reference switches that recomputation OFF for every
     * search ("Do not recompute neighbor distances along the path", :444-451: recompute_neighors = False), so they cannot change its
     * results; here the traversal is PQ-only by construction and they are accepted and have NO effect (the Python host logs that
     * once when a non-default value arrives: leann_amd/backend.py).  The exact rerank is a single batch whatever batch_recompute says. */
    int32_t dedup_node_dis;      /* :463 inert (see above) */
    float prune_ratio;           /* :464 inert */
    int32_t batch_recompute;     /* :465 inert */
    int32_t use_global_pruning;  /* :466 inert */
} lm_pq_search_params;
void lm_pq_search_params_default(lm_pq_search_params *p);
int lm_pq_batch_search(lm_index *idx, int64_t n, const float *x, int32_t k, const lm_pq_search_params *params,
                       int64_t *labels, float *distances);
int lm_pq_batch_search_device(lm_index *idx, int64_t n, const float *d_x, int32_t k,
                              const lm_pq_search_params *params, int64_t *d_labels, float *d_distances);

/* ---- stand-alone kernels (parity tests, shard merge) ----------------------------------------
 * Distances of query row qidx[i] against embedding row ids[i] with the canonical reduction of
 * oracle/lm_oracle.c:orc_dist.  d_table rows have stride d_padded; d_q is nq x d_padded. */
int lm_dist_gather(const void *d_table, int32_t dtype, int32_t d_padded, int32_t metric,
                   const float *d_q, const int32_t *d_qidx, const int32_t *d_ids, int64_t npairs,
                   float *d_out, void *stream);
/* Per-query merge of S per-shard top-k lists (60M-chunk config; ids already global, -1 = empty):
 * in: S x B x k, out: B x k, ordered by (internal distance, id). */
int lm_topk_merge(const int64_t *d_in_ids, const float *d_in_dist, int32_t S, int32_t B, int32_t k,
                  int32_t metric, int64_t *d_out_ids, float *d_out_dist, void *stream);

/* ---- fused encoder elementwise ops ---------------------------------------------------------------
 * out = LayerNorm(x + residual) * gamma + beta over the last dim; fp16 in/out, fp32 arithmetic;
 * residual may be NULL.  Part of the BERT forward inside compute_embeddings
 * (leann/embedding_compute.py:229-239).  The GEMMs and the attention of that forward are the hand-written MFMA kernels below
 * (hidden 384: lm_qkv_h384_f16, lm_attn_varlen_hd32_f16, lm_layer_tail_h384_f16); no library call is on the default path. */
int lm_add_layernorm_f16(const void *d_x, const void *d_residual, const void *d_gamma, const void *d_beta,
                         void *d_out, int64_t rows, int32_t hidden, float eps, void *stream);

/* Fused self-attention for packed variable-length sequences, head_dim 32 (hidden = heads*32), lengths 1..256:
 * d_qkv [total_tokens][3][heads][32] fp16 (the QKV GEMM output), d_cu_seqlens int32[n_seqs+1],
 * d_out [total_tokens][heads*32] fp16.  softmax(QK^T/sqrt(32))V per (sequence, head), MFMA 32x32x16 f16. */
int lm_attn_varlen_hd32_f16(const void *d_qkv, const int32_t *d_cu_seqlens, int32_t n_seqs, int32_t heads,
                            int32_t max_len, void *d_out, void *stream);

/* The same kernel for head_dim 32 or 64 (hidden = heads * head_dim; 64: bge-base / contriever, lengths 1..512; 32: lengths 1..256):
 * d_qkv [total_tokens][3][heads][head_dim] fp16, d_out [total_tokens][heads * head_dim] fp16. */
int lm_attn_varlen_f16(const void *d_qkv, const int32_t *d_cu_seqlens, int32_t n_seqs, int32_t heads, int32_t head_dim,
                       int32_t max_len, void *d_out, void *stream);

/* Embedding front end in one pass: d_out[r] = LayerNorm(half(word[tok[r]] + type0) + pos_table[pos[r]]);
 * tables and output fp16, hidden <= 768.  Replaces the three gathers/adds before the embedding LayerNorm of
 * the BERT forward (leann/embedding_compute.py:229-239).  On the Python host's default path (LEANN_MI355X_EMBED=0 selects
 * the torch ops for A/B runs). */
int lm_embed_layernorm_f16(const int32_t *d_tok, const int32_t *d_pos, const void *d_word, const void *d_pos_table,
                           const void *d_type0, const void *d_gamma, const void *d_beta, void *d_out, int64_t rows,
                           int32_t hidden, float eps, void *stream);

/* Packing front end: padded token ids [n][t] + lengths + cumulative lengths (int32[n+1]) -> packed token ids and
 * positions [total].  Replaces the masked selects before the first encoder layer.  On the default path (LEANN_MI355X_PACK=0 = torch ops, A/B). */
int lm_pack_tokens(const int32_t *d_ids, const int32_t *d_lens, const int32_t *d_cu_seqlens, int32_t n, int32_t t,
                   int32_t *d_tok, int32_t *d_pos, void *stream);

/* Mean pooling over the tokens of each packed sequence (+ optional L2 normalisation), fp16 in, fp32 out
 * [n_seqs][hidden]; fixed summation order (deterministic).  sentence-transformers Pooling as done in
 * leann/embedding_compute.py:323-334.  On the default path (LEANN_MI355X_POOL=0 = torch ops, A/B). */
int lm_meanpool_varlen_f16(const void *d_x, const int32_t *d_cu_seqlens, int32_t n_seqs, int32_t hidden,
                           int32_t normalize, float *d_out, void *stream);

/* The second half of a BERT layer with hidden size 384 in ONE kernel (csrc/lm_layer_tail_h384.hip) -- attention output projection,
 * residual, LayerNorm, then the feed-forward block with its residual and LayerNorm:
 *   x     = LayerNorm(resid + attn W_o^T + b_o) * gamma1 + beta1          (never written to HBM)
 *   d_out = LayerNorm(x + GELU(x W1^T + b1) W2^T + b2) * gamma + beta,    attn / resid / d_out [tokens][384] fp16,
 * biases fp32, GELU = exact erf form, MFMA 32x32x16 f16 with the ffn-wide intermediate held in accumulators (never written to HBM).
 * The three weight matrices are passed as ready-made LDS IMAGES (written once per model by lm_layer_tail_pack_h384), so that the
 * kernel's weight stream is a linear LDS-DMA copy, and the two products of the feed-forward block alternate MFMA by MFMA on single
 * accumulator chains (no partial sums, no stage arithmetic: ~290 instructions per 48 MFMAs on its one wave per SIMD).  ffn a multiple
 * of 192 in [192, 1728]; other shapes: the general GEMM path (lm_gemm_f16).  DEFAULT path of the hidden-384 forward
 * (LEANN_MI355X_TAIL=0 in the Python host = lm_gemm_ws_h384_f16 + lm_add_layernorm_f16 + lm_gemm_f16 x 2 + lm_add_layernorm_f16, A/B).
 * Part of the BERT forward in compute_embeddings (leann/embedding_compute.py:229-239).  (Generations 1-3 of this kernel are history:
 * generation 3 survives only in the diagnosis build, csrc/diag/, as scripts/kbench.cpp's A/B reference.) */
int lm_layer_tail_h384_f16(const void *d_attn, const void *d_resid, const void *d_wo_img, const float *d_bo,
                           const void *d_gamma1, const void *d_beta1, float eps1, const void *d_w1_img, const float *d_b1,
                           const void *d_w2_img, const float *d_b2, const void *d_gamma, const void *d_beta, void *d_out,
                           int64_t tokens, int32_t ffn, float eps, void *stream);
/* d_wo_slabs = W_o as [12][384][32] (slab s = input features 32 s .. 32 s + 31, natural order: leann_amd/encoder.py pack_wo_slabs),
 * d_w1_acc = W1 [ffn][384] with its COLUMNS in accumulator order (pack_w1_acc_order), d_w2_slabs = W2 as [ffn/32][384][32] with the k
 * order of fused_mlp_k_permutation (pack_w2_fused_mlp) -> the images lm_layer_tail_h384_f16 streams (same sizes; XOR-swizzled 16-byte chunks: what makes the
 * kernel's ds_read_b128 fragment reads bank-conflict free).  Device to device, on `stream`. */
int lm_layer_tail_pack_h384(const void *d_wo_slabs, const void *d_w1_acc, const void *d_w2_slabs, int32_t ffn, void *d_wo_img,
                            void *d_w1_img, void *d_w2_img, void *stream);

/* The FIRST half of a BERT layer with hidden size 384 = 12 heads x 32 in ONE kernel (csrc/lm_qkv_attn_h384.hip, round 6): QKV projection fused into
 * self-attention -- d_out[tokens][384] = concat_h softmax(Q_h K_h^T / sqrt(32)) V_h with [Q | K | V] = x W_qkv^T + b_qkv computed per (sequence, head)
 * on chip: Q, K, V never go to HBM (as two kernels they were a 604 MB write + 604 MB read per 262 k tokens around 201 MB of x and 201 MB of output).
 * d_x [tokens][384] fp16 packed sequences, d_cu_seqlens int32[n_seqs + 1], lengths 1..256; d_wqkv_img = lm_qkv_pack_h384's image of the nn.Linear
 * weight [1152][384] (rows W_q | W_k | W_v), d_bqkv fp32[1152].  Replaces the model.encode() call's per-layer QKV GEMM + attention
 * (leann/embedding_compute.py:229-239) on large hidden-384 forwards of long sequences (lm_h384_first_half_form below says when); the stand-alone pair stays
 * the default elsewhere. */
int lm_qkv_attn_h384_f16(const void *d_x, const void *d_wqkv_img, const float *d_bqkv, const int32_t *d_cu_seqlens, int32_t n_seqs, int32_t max_len,
                         int64_t total_tokens, void *d_out, void *stream);
/* Which kernels the first half (QKV projection + attention) of a LARGE hidden-384 layer runs on, for a model with `heads` heads and a forward of n_seqs
 * sequences / total_tokens tokens whose longest sequence is max_len -- the one decision lm_bert_h384_forward_packed and a host that launches kernel by
 * kernel share: 0 = lm_qkv_attn_h384_f16 (forwards whose MEAN length is >= 216: its cost per sequence is the same from 129 to 256 tokens, it wins on
 * long sequences and ties on the benchmark corpus' N(180, 50) lengths; LEANN_MI355X_FUSED_QKV_ATTN=1 forces it, =0 forbids it), 1 = lm_qkv_h384_f16
 * (head-major output) + attention generation 3, 2 = the pair over the [tokens][1152] layout (other head counts / lengths,
 * LEANN_MI355X_QKV_LAYOUT=0, the older attention generations' switches). */
int lm_h384_first_half_form(int32_t heads, int32_t max_len, int64_t total_tokens, int32_t n_seqs);

/* Weight-STREAMING form of the 384-input linear layer (csrc/lm_qkv_h384.hip): d_out[tokens][n_out] = x W^T + b, n_out a multiple of
 * 128 in [256, 6144], d_w_img = lm_qkv_pack_h384's image of the nn.Linear weight [n_out][384] (same size).  x is read once (a wave holds
 * its 32 token rows as MFMA B fragments for the whole kernel), W streams through a four-stage LDS ring shared by eight waves -- two per
 * SIMD, ~230 registers each -- and a slab's 32 x 32 results leave through a per-wave LDS tile as full 128-byte lines.  The QKV projection
 * of the hidden-384 forward (leann/embedding_compute.py:229-239). */
int lm_qkv_h384_f16(const void *d_x, const void *d_w_img, const float *d_bias, int32_t n_out, void *d_out, int64_t tokens,
                    void *stream);
int lm_qkv_pack_h384(const void *d_w, int32_t n_out, void *d_w_img, void *stream); /* device to device, on `stream` */

/* Weight-stationary form of the 384-input linear layer (csrc/lm_gemm_ws_h384.hip): d_out[tokens][n_out] = x W^T + b with
 * d_w = the nn.Linear weight itself, [n_out][384] fp16 row major (no packing), n_out a multiple of 192 (<= 6144).  A
 * workgroup keeps its 192 x 384 weight block resident in LDS and streams token tiles past it (no per-slab barriers, two
 * waves per SIMD).  The QKV projection of lm_bert_h384_forward_packed when the model carries no wqkv_img (and of the Python host under
 * LEANN_MI355X_QKV=0, A/B); with the layer tail switched off also the output projection, followed by lm_add_layernorm_f16. */
int lm_gemm_ws_h384_f16(const void *d_x, const void *d_w, const float *d_bias, int32_t n_out, void *d_out, int64_t tokens,
                        void *stream);

/* General fp16 linear layer (csrc/lm_gemm_f16.hip):  d_out[tokens][n_out] = epi(x[tokens][k_in] W^T + b), d_w = the nn.Linear weight
 * itself ([n_out][k_in] fp16 row major, no packing, < 4 GiB), bias fp32, n_out % 128 == 0, k_in % 128 == 0; any token count.
 * epilogue: 0 = none, 1 = exact-erf GELU, 2 = + d_residual[tokens][n_out] (fp16), 3 = both.  256 x 256 workgroup tiles (128 x 128
 * when n_out % 256 != 0), MFMA 32x32x16 f16, operands staged L2 -> LDS by global_load_lds through two stages.  The GEMMs of the BERT
 * forward in compute_embeddings (leann/embedding_compute.py:229-239) for models whose hidden size is not 384 (bge-base,
 * contriever: 768): QKV projection (epilogue 0), attention output projection (2), feed-forward products (1, 2). */
int lm_gemm_f16(const void *d_x, const void *d_w, const float *d_bias, const void *d_residual, int32_t epilogue, int32_t n_out,
                int32_t k_in, void *d_out, int64_t tokens, void *stream);

/* ---- the whole packed BERT forward (hidden 384, mean or CLS pooling) in one call ---------------------
 * Replaces compute_embeddings' model.encode() (leann/embedding_compute.py:229-239) for sentence-transformers models of the
 * all-MiniLM family: embedding front end, per layer {lm_gemm_ws_h384_f16 (QKV), lm_attn_varlen_hd32_f16,
 * lm_layer_tail_h384_f16}, lm_meanpool_varlen_f16 / lm_clspool_varlen_f16 -- one foreign-function call per recompute round
 * instead of ~3 L + 2.  ffn a multiple of 192 in [192, 1728] (other widths: lm_bert_forward_packed).
 * All pointers are device pointers except `layers` (host array).  Weight layouts as documented at the entry points named above
 * (the three images of lm_layer_tail_pack_h384; leann_amd/encoder.py: pack_tail_images).  d_out: fp32 [n_seqs][384]. */
typedef struct lm_bert_h384_layer {
    const void *wqkv;  /* [1152][384] fp16, nn.Linear layout */
    const float *bqkv; /* [1152] */
    const void *wo_img; /* lm_layer_tail_pack_h384's image of [12][384][32] */
    const float *bo;
    const void *ln1_gamma, *ln1_beta; /* fp16 [384] */
    const void *w1_img;               /* ... of [ffn][384], columns in accumulator order */
    const float *b1;
    const void *w2_img; /* ... of [ffn/32][384][32] */
    const float *b2;
    const void *ln2_gamma, *ln2_beta;
    /* Optional (all three or none): the plain nn.Linear weights [384][384], [ffn][384], [384][ffn] fp16.  With them a forward of at
     * most LM_BERT_SMALL_TOKENS tokens (a one-query search round recomputes ~10 chunks) runs every layer on the general kernels --
     * lm_gemm_f16 x 4 + attention + lm_add_layernorm_f16 x 2: many small workgroups spread over the chip -- instead of the fused
     * layer tail, whose 128-token workgroup is one ~77 us dependency chain however few tokens it holds (MI355X, 200k-chunk index,
     * B = 1 search: p50 59.7 -> 47.2 ms with the limit at 6144 tokens in round 3; round 4 measured the crossover again -- 6144 /
     * 16384 / 32768 tokens: B = 1 p50 44.9 / 39.3 / 40.0 ms, B = 4 70.1 / 56.5 / 56.7 ms, B = 16 115.2 / 107.9 / 108.6 ms -- and moved
     * it to 16384; round 6, with the QKV projection of the next size class on the general GEMM (LM_BERT_QKV_GEMM_TOKENS below) and rounds that
     * dynamic batching makes 5 - 10 x larger: 4096 / 8192 / 16384 tokens: B = 1 p50 37.8 / 37.9 / 38.2 ms at batch_size 0, 13.1 / 13.0 / 14.0 ms at 64,
     * 11.9 / 11.8 / 11.9 ms at 128, B = 16 86.4 / 86.6 / 89.1 ms -- moved to 8192: profiles/r6_latency_b1_forward_size_limits_200k.jsonl).  Same
     * arithmetic up to fp16 rounding of the intermediate activations. */
    const void *wo, *w1, *w2;
    /* Optional: lm_qkv_pack_h384's image of wqkv.  With it the large-forward QKV projection runs on the weight-streaming kernel
     * (lm_qkv_h384_f16: x read once, two waves per SIMD); NULL = the weight-stationary one (lm_gemm_ws_h384_f16 on wqkv). */
    const void *wqkv_img;
} lm_bert_h384_layer;
#define LM_BERT_SMALL_TOKENS 8192
/* Round 6: between the small-forward form and the streaming QKV kernel.  A LARGE-form forward of at most this many tokens takes its QKV projection
 * from the general GEMM (lm_gemm_f16 on wqkv, then attention over the [tokens][1152] layout) instead of the weight-streaming kernel, whose 256-token
 * workgroup is a ~45 us chain however few of the 256 CUs the forward fills: 12,288 / 24,576 / 49,152 tokens: 18.6 / 32.2 / 62.4 us against 45.8 /
 * 53.8 / 65.3 us (profiles/r6_kbench_layer_kernels_at_small_forward_sizes.jsonl) -- the rounds of a small-batch search under dynamic batching
 * (batch_size 64 ... 128: 12 k ... 35 k tokens).  The fused layer tail stays (64 / 80 us at those sizes against ~100 / ~160 for the five launches it
 * replaces).  LEANN_MI355X_QKV_GEMM_TOKENS overrides (0 = never; LEANN_MI355X_SMALL_TOKENS=0, "the large-forward kernels at every size", implies 0). */
#define LM_BERT_QKV_GEMM_TOKENS 45056

typedef struct lm_bert_h384 {
    int32_t n_layers, heads, ffn, normalize;
    int32_t pooling; /* 0 = mean over the tokens (all-MiniLM), 1 = CLS (bge-small) */
    float ln_eps;
    const void *word, *pos_table, *type0, *emb_gamma, *emb_beta; /* fp16 */
    const lm_bert_h384_layer *layers;                           /* host array of n_layers entries */
} lm_bert_h384;

size_t lm_bert_h384_workspace_bytes(int64_t total_tokens); /* sized for either layer form (ffn <= 2560) */
int lm_bert_h384_forward_packed(const lm_bert_h384 *m, const int32_t *d_tok, const int32_t *d_pos, const int32_t *d_cu_seqlens,
                                int32_t n_seqs, int64_t total_tokens, int32_t max_len, void *d_workspace, size_t workspace_bytes,
                                float *d_out, void *stream);

/* ---- the whole packed BERT forward, general widths, in one call (csrc/lm_encoder_forward.cpp) -------
 * The same replacement of compute_embeddings' model.encode() (leann/embedding_compute.py:229-239) for BERT-architecture models the
 * hidden-384 kernels do not cover (bge-base-en-v1.5, contriever: hidden 768, head_dim 64, CLS or mean pooling): embedding front end,
 * per layer {lm_gemm_f16 (QKV) | lm_attn_varlen_f16 | lm_gemm_f16 (+ residual) | lm_add_layernorm_f16 | lm_gemm_f16 (+ GELU) |
 * lm_gemm_f16 (+ residual) | lm_add_layernorm_f16}, lm_meanpool_varlen_f16 / lm_clspool_varlen_f16 -- the launch sequence of
 * leann_amd/encoder.py: EncoderLayer._forward_packed_general, as one foreign-function call.  Weights are the nn.Linear tensors
 * themselves (fp16 [n_out][k_in], biases fp32, LayerNorm parameters fp16).  Envelope: hidden % 128 == 0 and <= 768, ffn % 128 == 0,
 * head_dim = hidden / heads = 32 (chunk lengths <= 256) or 64 (<= 512).  d_out: fp32 [n_seqs][hidden]. */
typedef struct lm_bert_layer {
    const void *wqkv;  /* [3 hidden][hidden] */
    const float *bqkv;
    const void *wo;    /* [hidden][hidden] */
    const float *bo;
    const void *ln1_gamma, *ln1_beta;
    const void *w1;    /* [ffn][hidden] */
    const float *b1;
    const void *w2;    /* [hidden][ffn] */
    const float *b2;
    const void *ln2_gamma, *ln2_beta;
} lm_bert_layer;

typedef struct lm_bert {
    int32_t hidden, n_layers, heads, ffn;
    int32_t pooling;   /* 0 = mean over the tokens, 1 = CLS (first token) */
    int32_t normalize; /* L2-normalise the pooled vector */
    float ln_eps;
    const void *word, *pos_table, *type0, *emb_gamma, *emb_beta; /* fp16 */
    const lm_bert_layer *layers;                                  /* host array of n_layers entries */
} lm_bert;

size_t lm_bert_workspace_bytes(const lm_bert *m, int64_t total_tokens);
int lm_bert_forward_packed(const lm_bert *m, const int32_t *d_tok, const int32_t *d_pos, const int32_t *d_cu_seqlens, int32_t n_seqs,
                           int64_t total_tokens, int32_t max_len, void *d_workspace, size_t workspace_bytes, float *d_out, void *stream);
/* CLS pooling of packed sequences (+ optional L2 normalisation): d_out[s] = float(x[first token of s]); fp16 in, fp32 out. */
int lm_clspool_varlen_f16(const void *d_x, const int32_t *d_cu_seqlens, int32_t n_seqs, int32_t hidden, int32_t normalize,
                          float *d_out, void *stream);

/* ---- token store ---------------------------------------------------------------------------
 * Replaces PassageManager.get_passage (leann/api.py:203-215) + tokenisation inside
 * compute_embeddings (leann/embedding_compute.py:229-239) at query time: passages are tokenised
 * once at load and kept packed in HBM (u16 ids, vocab < 65536). */
int lm_tokens_create(const uint16_t *tokens, const uint64_t *offsets /* n+1 */, int64_t n, int device,
                     lm_tokens **out);
void lm_tokens_free(lm_tokens *t);
/* d_out_ids: n x T int32 (pad_id beyond the chunk length, chunks truncated to T);
 * d_out_len: n int32. */
int lm_tokens_gather(const lm_tokens *t, const int32_t *d_ids, int32_t n, int32_t T, int32_t pad_id,
                     int32_t *d_out_ids, int32_t *d_out_len, void *stream);
int64_t lm_tokens_count(const lm_tokens *t);

/* ---- the built-in recompute provider (csrc/lm_recompute.hip) ------------------------------------
 * One recompute round trip of the reference -- ZMQ REQ of the node ids, PassageManager.get_passage per id, tokeniser,
 * model.encode() (hnsw_embedding_server.py:148-284, leann/api.py:203-215, leann/embedding_compute.py:229-239) -- as library code:
 * ids -> token store -> packed forward -> fp32 [n][hidden], no interpreter in the search loop.  `model` (the struct and its `layers`
 * array are copied; the device weights they point to must outlive the handle) and `tokens` as above; chunks are truncated to
 * max_seq_len (1..256) tokens; a call with more than max_tokens_per_forward tokens runs as several forwards (bounds by cumulative
 * token count, as leann_amd/encoder.py: encode_tokens_packed cuts them).  Envelope: lm_bert_h384_forward_packed's
 * (lm_recompute_create: hidden 384, the fused kernels) or lm_bert_forward_packed's (lm_recompute_create_general: e.g. bge-base,
 * 768-d; max_seq_len up to 512 at head_dim 64); the embeddings are fp32 [n][hidden].
 *   lm_index_set_recompute(idx, rc)  attaches it as the index's embedding provider (NULL detaches) and lets the search loop
 *                                    compute the round's chunk lengths before its own per-round device-to-host copy: ONE host
 *                                    synchronisation per round (lm_index_set_provider with a foreign callback: the callback's own);
 *   lm_recompute_provider            the same object as a plain lm_provider_fn (user = the handle), e.g. for lm_index_set_provider;
 *   lm_recompute_embed               embeddings of n chunk ids into the caller's fp32 [n][hidden] buffer (index build time). */
typedef struct lm_recompute lm_recompute;
typedef struct lm_recompute_stats {
    int64_t calls, chunks, tokens, forwards, host_syncs;
} lm_recompute_stats;
int lm_recompute_create(const lm_bert_h384 *model, const lm_tokens *tokens, int32_t max_seq_len, int64_t max_tokens_per_forward,
                        lm_recompute **out);
int lm_recompute_create_general(const lm_bert *model, const lm_tokens *tokens, int32_t max_seq_len, int64_t max_tokens_per_forward,
                                lm_recompute **out);
void lm_recompute_free(lm_recompute *rc);
int lm_recompute_provider(void *user, const int32_t *d_ids, int32_t n, void **d_out, void *stream);
int lm_recompute_embed(lm_recompute *rc, const int32_t *d_ids, int32_t n, float *d_out, void *stream);
int lm_recompute_get_stats(const lm_recompute *rc, lm_recompute_stats *out);
int lm_index_set_recompute(lm_index *idx, lm_recompute *rc);

/* ---- measurement: event pairs around the library's own launches of the encoder's MFMA kernels (csrc/lm_timing.cpp) ----
 * One HIP event pair per launch, recorded on the stream the kernel is launched on, for the kernels whose bit is set in `mask`
 * (bit LM_KT_*); 0 = off (the default: a disabled launch pays one atomic load).  Works on every launch path -- the one-call forwards,
 * the built-in recompute provider inside lm_index_search*, single entry points.  lm_kernel_timing_read waits for the pairs recorded
 * so far and returns, per kernel, launches / summed milliseconds / summed algorithmic flops since the last reset (attention's
 * flops, 4 H sum(len^2), are summed on the device from the launch's own cumulative lengths).  bench.py's `roofline` is built from these over its timed region. */
#define LM_KT_LAYER_TAIL 0 /* lm_layer_tail_h384_f16 */
#define LM_KT_GEMM_WS 1    /* lm_gemm_ws_h384_f16 */
#define LM_KT_ATTN 2       /* lm_attn_varlen_hd32_f16 / lm_attn_varlen_f16 */
#define LM_KT_GEMM_F16 3   /* lm_gemm_f16 */
#define LM_KT_QKV 4        /* lm_qkv_h384_f16 (the weight-streaming QKV projection of the large hidden-384 forwards) */
#define LM_KT_QKV_ATTN 5   /* lm_qkv_attn_h384_f16 (QKV projection fused into attention: the first half of a hidden-384 layer of the large forwards) */
#define LM_KT_COUNT 6
typedef struct lm_kernel_time {
    const char *name;
    int64_t launches;
    double ms, work;
} lm_kernel_time;
int lm_kernel_timing_enable(uint32_t mask);
/* out[0 .. min(capacity, LM_KT_COUNT)) are written (a caller built against an older, shorter LM_KT_COUNT is never overrun). */
int lm_kernel_timing_read(lm_kernel_time *out, int32_t capacity, int32_t reset);

#ifdef __cplusplus
}
#endif
#endif /* LEANN_MI355X_H */
