/*
 * lm_oracle_pq.c -- CPU ORACLE of the DiskANN-style path: PQ-ADC beam search + deferred exact rerank.
 *
 * TEST INFRASTRUCTURE ONLY (same rules as lm_oracle.c).  PARITY STATUS: "parity unpinned": the
 * reference's implementation lives in the un-vendored fork github.com/yichuan-w/DiskANN
 * (packages/leann-backend-diskann/third_party/DiskANN, empty; .gitmodules:4-6).  Restated from the
 * published DiskANN algorithm (Subramanya et al. 2019: greedy/beam search over a flat Vamana graph
 * with a candidate list of size L, frontier of W nodes per iteration, distances from product-
 * quantised codes via per-query lookup tables) and from the reference's call site / documented
 * strategy, packages/leann-backend-diskann/leann_backend_diskann/diskann_backend.py:
 *   :444-449  "Traversal always uses PQ distances; if recompute_embeddings=True, do a single final
 *              rerank via deferred fetch (fetch embeddings for the final candidate set only); do not
 *              recompute neighbor distances along the path"
 *   :453-467  batch_search(query, B, top_k, complexity(L), beam_width(W), num_threads,
 *              use_deferred_fetch, skip_search_reorder, recompute_neighbors, dedup_node_dis,
 *              prune_ratio, batch_recompute, use_global_pruning)
 * The embedding fetch replaces the protobuf NodeEmbeddingRequest of diskann_embedding_server.py:258-334.
 *
 * Normalised to set semantics under (distance, id), exactly like the HNSW oracle:
 *   list(q)  = the L smallest (pq_dist, id) among all nodes evaluated so far (expanded or not)
 *   iterate  : pop the W smallest unexpanded entries, gather their neighbours in stored order, skip
 *              visited, ADC distance for the rest, merge; stop when no unexpanded entry is left
 *   rerank   : exact canonical distance (orc_dist) of every list entry, sort by (dist,id), top k
 * Canonical PQ arithmetic (bit-exact with the HIP kernel):
 *   LUT[j][c] = sum_t (q[j*dsub+t]-cb[j][c][t])^2  (l2)  |  -sum_t q*cb (ip), sequential fmaf over t
 *   adc(v)    = (p0+p1)+(p2+p3),  p_r = sequential sum over j = r, r+4, r+8, ...  of LUT[j][code[v][j]]
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_METRIC_L2 1

typedef struct {
    int64_t N;
    int32_t D, Dp, max_level, entry_point, metric;
    const uint64_t *node_offsets;
    const uint64_t *level_ptr;
    const int32_t *neighbors;
    const int32_t *levels;
} orc_graph; /* same layout as lm_oracle.c */

typedef int (*orc_provider_fn)(void *user, const int32_t *ids, int32_t n, float *out);
float orc_dist(const float *e, const float *q, int32_t Dp, int32_t metric);

typedef struct {
    int32_t m;    /* sub-quantisers */
    int32_t dsub; /* D / m (uniform chunks; ignored when chunk_off is given) */
    const float *codebooks; /* chunk j: 256 centroids x len_j floats at offset 256 * lo_j (uniform: m x 256 x dsub) */
    const uint8_t *codes;   /* N x m */
    const int32_t *chunk_off; /* NULL = uniform; else m + 1 offsets: chunk j covers dimensions [chunk_off[j], chunk_off[j+1]) --
                                 the public DiskANN pq_pivots layout (chunk_offsets), lengths may differ, zero-length chunks allowed */
} orc_pq;

typedef struct {
    int32_t L, W, k;
    int32_t use_deferred_fetch;  /* rerank with embeddings from the provider */
    int32_t skip_search_reorder; /* return the PQ order/distances */
} orc_pq_params;

typedef struct {
    int64_t n_adc, n_rerank_unique, n_rounds, n_expand;
} orc_pq_stats;

static inline uint64_t pk(float d, int32_t id) {
    if (d != d) d = INFINITY;
    if (d == 0.0f) d = 0.0f;
    uint32_t u;
    memcpy(&u, &d, 4);
    u ^= (u >> 31) ? 0xFFFFFFFFu : 0x80000000u;
    return ((uint64_t)u << 32) | ((uint64_t)(uint32_t)id << 1);
}
static inline float pk_dist(uint64_t key) {
    uint32_t u = (uint32_t)(key >> 32);
    u ^= (u >> 31) ? 0x80000000u : 0xFFFFFFFFu;
    float d;
    memcpy(&d, &u, 4);
    return d;
}
static inline int32_t pk_id(uint64_t key) { return (int32_t)((uint32_t)key >> 1); }

void orc_pq_lut(const orc_pq *pq, const float *q, int32_t metric, float *lut) {
    for (int j = 0; j < pq->m; ++j)
        for (int c = 0; c < 256; ++c) {
            const int lo = pq->chunk_off ? pq->chunk_off[j] : j * pq->dsub;
            const int len = pq->chunk_off ? pq->chunk_off[j + 1] - lo : pq->dsub;
            const float *cb = pq->codebooks + (size_t)256 * lo + (size_t)c * len;
            const float *qs = q + lo;
            float acc = 0.0f;
            if (metric == ORC_METRIC_L2) {
                for (int t = 0; t < len; ++t) {
                    float d = qs[t] - cb[t];
                    acc = fmaf(d, d, acc);
                }
            } else {
                for (int t = 0; t < len; ++t) acc = fmaf(qs[t], cb[t], acc);
                acc = -acc;
            }
            lut[j * 256 + c] = acc;
        }
}

float orc_pq_adc(const orc_pq *pq, const float *lut, int64_t v) {
    const uint8_t *code = pq->codes + (size_t)v * pq->m;
    float p[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    for (int j = 0; j < pq->m; ++j) p[j & 3] = p[j & 3] + lut[j * 256 + code[j]];
    return (p[0] + p[1]) + (p[2] + p[3]);
}

static void list_insert(uint64_t *list, int32_t *n, int32_t L, uint64_t key) {
    if (*n == L) {
        if (key >= list[L - 1]) return;
        (*n)--;
    }
    int32_t i = *n;
    while (i > 0 && list[i - 1] > key) {
        list[i] = list[i - 1];
        --i;
    }
    list[i] = key;
    (*n)++;
}

static int cmp_i32(const void *a, const void *b) {
    int32_t x = *(const int32_t *)a, y = *(const int32_t *)b;
    return (x > y) - (x < y);
}
static int cmp_u64(const void *a, const void *b) {
    uint64_t x = *(const uint64_t *)a, y = *(const uint64_t *)b;
    return (x > y) - (x < y);
}

/* queries: B x Dp (zero padded).  table (N x Dp) or provider supplies exact embeddings for the rerank. */
int orc_pq_search(const orc_graph *g, const orc_pq *pq, const float *table, orc_provider_fn provider, void *user,
                  const float *queries, int32_t B, const orc_pq_params *prm, int64_t *out_ids, float *out_dist,
                  orc_pq_stats *stats) {
    const int32_t k = prm->k, W = prm->W < 1 ? 1 : prm->W;
    const int32_t L = prm->L > k ? prm->L : k;
    const int64_t N = g->N;
    const int32_t Dp = g->Dp;
    const float fill = g->metric == ORC_METRIC_L2 ? INFINITY : -INFINITY;
    orc_pq_stats st = {0, 0, 0, 0};
    for (int64_t i = 0; i < (int64_t)B * k; ++i) {
        out_ids[i] = -1;
        out_dist[i] = fill;
    }
    if (N == 0 || B == 0 || g->entry_point < 0) {
        if (stats) *stats = st;
        return 0;
    }
    const int64_t nw = (N + 31) / 32;
    uint64_t *lists = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)B * L);
    int32_t *nlist = (int32_t *)calloc((size_t)B, 4);
    int64_t n_adc = 0, n_rounds = 0, n_expand = 0;
#pragma omp parallel for schedule(dynamic, 1) reduction(+ : n_adc, n_expand) reduction(max : n_rounds)
    for (int32_t q = 0; q < B; ++q) {
        float *lut = (float *)malloc(sizeof(float) * (size_t)pq->m * 256);
        uint32_t *vis = (uint32_t *)calloc((size_t)nw, 4);
        uint64_t *list = lists + (size_t)q * L;
        int32_t n = 0;
        orc_pq_lut(pq, queries + (size_t)q * Dp, g->metric, lut);
        /* start at the entry point (medoid) */
        int32_t ep = g->entry_point;
        vis[ep >> 5] |= 1u << (ep & 31);
        list_insert(list, &n, L, pk(orc_pq_adc(pq, lut, ep), ep));
        n_adc++;
        int64_t rounds = 0;
        for (;;) {
            int32_t pops[1024];
            int32_t np = 0;
            for (int32_t i = 0; i < n && np < W && np < 1024; ++i)
                if (!(list[i] & 1ull)) {
                    list[i] |= 1ull;
                    pops[np++] = pk_id(list[i]);
                }
            if (np == 0) break;
            rounds++;
            n_expand += np;
            /* evaluate first, insert after: the round is a set operation */
            uint64_t newk[1024 * 8];
            int32_t nn = 0;
            for (int32_t pi = 0; pi < np; ++pi) {
                uint64_t p = g->node_offsets[pops[pi]];
                for (uint64_t j = g->level_ptr[p]; j < g->level_ptr[p + 1]; ++j) {
                    int32_t v = g->neighbors[j];
                    uint32_t bit = 1u << (v & 31);
                    if (vis[v >> 5] & bit) continue;
                    vis[v >> 5] |= bit;
                    if (nn < 1024 * 8) newk[nn++] = pk(orc_pq_adc(pq, lut, v), v);
                }
            }
            n_adc += nn;
            for (int32_t i = 0; i < nn; ++i) list_insert(list, &n, L, newk[i]);
        }
        if (rounds + 1 > n_rounds) n_rounds = rounds + 1;
        nlist[q] = n;
        free(lut);
        free(vis);
    }
    st.n_adc = n_adc;
    st.n_rounds = n_rounds;
    st.n_expand = n_expand;

    int rc = 0;
    const int rerank = !prm->skip_search_reorder && (table || (prm->use_deferred_fetch && provider));
    int32_t *uniq = NULL;
    float *emb = NULL;
    int64_t nu = 0;
    if (rerank && !table) {
        /* ONE deferred fetch for the union of all candidate lists (diskann_backend.py:444-449) */
        uint32_t *bm = (uint32_t *)calloc((size_t)nw, 4);
        for (int32_t q = 0; q < B; ++q)
            for (int32_t i = 0; i < nlist[q]; ++i) {
                int32_t v = pk_id(lists[(size_t)q * L + i]);
                if (!(bm[v >> 5] & (1u << (v & 31)))) {
                    bm[v >> 5] |= 1u << (v & 31);
                    nu++;
                }
            }
        uniq = (int32_t *)malloc(sizeof(int32_t) * (size_t)(nu ? nu : 1));
        emb = (float *)malloc(sizeof(float) * (size_t)(nu ? nu : 1) * Dp);
        int64_t u = 0;
        for (int64_t w = 0; w < nw; ++w)
            for (uint32_t bits = bm[w]; bits; bits &= bits - 1) uniq[u++] = (int32_t)(w * 32 + __builtin_ctz(bits));
        free(bm);
        st.n_rerank_unique = nu;
        if (nu > 0) rc = provider(user, uniq, (int32_t)nu, emb);
    }
    if (rc == 0) {
#pragma omp parallel for schedule(dynamic, 1)
        for (int32_t q = 0; q < B; ++q) {
            uint64_t *list = lists + (size_t)q * L;
            int32_t n = nlist[q];
            if (rerank) {
                for (int32_t i = 0; i < n; ++i) {
                    int32_t v = pk_id(list[i]);
                    const float *row;
                    if (table) row = table + (size_t)v * Dp;
                    else {
                        int32_t *hit = (int32_t *)bsearch(&v, uniq, (size_t)nu, 4, cmp_i32);
                        row = emb + (size_t)(hit - uniq) * Dp;
                    }
                    list[i] = pk(orc_dist(row, queries + (size_t)q * Dp, Dp, g->metric), v);
                }
                qsort(list, (size_t)n, 8, cmp_u64);
            }
            for (int32_t i = 0; i < k && i < n; ++i) {
                float d = pk_dist(list[i]);
                out_ids[(size_t)q * k + i] = pk_id(list[i]);
                out_dist[(size_t)q * k + i] = g->metric == ORC_METRIC_L2 ? d : -d;
            }
        }
    }
    free(uniq);
    free(emb);
    free(lists);
    free(nlist);
    if (stats) *stats = st;
    return rc;
}
