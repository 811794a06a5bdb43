/*
 * lm_oracle_faiss.c -- SECOND, INDEPENDENT CPU oracle: a literal heap-based transcription of upstream faiss'
 * HNSW search (facebookresearch/faiss, faiss/impl/HNSW.cpp + faiss/utils/Heap.h, v1.7.4 .. v1.8 series:
 * HNSW::search, greedy_update_nearest, search_from_candidates, HNSW::MinimaxHeap::{push,pop_min,count_below},
 * heap_push / heap_pop / heap_replace_top / heap_reorder with CMax<float,int>; faiss/impl/ResultHandler.h:
 * HeapBlockResultHandler's add_result; IndexHNSW::search's negation of similarity metrics).
 *
 * THIS FILE IS TEST INFRASTRUCTURE, NOT PRODUCT CODE (same rule as lm_oracle.c: only tests/ load it).
 *
 * Why it exists ("a second, independent pin for the oracle"): lm_oracle.c restates the algorithm in
 * SET semantics (sorted pool, lock-step rounds) -- the form a parallel implementation can match bit for bit.
 * Every traversal check of round 1 compared that restatement with itself or with brute force.  This file keeps
 * the ORIGINAL data structures instead -- a binary max-heap of capacity ef whose popped slots stay in the array
 * with id = -1, pop_min by linear scan, count_below over all slots, a separate result max-heap of size k, one
 * query at a time, neighbours evaluated one by one in stored order -- so that
 *     orc_search(beam = 1)  ==  orcf_search           (ids, distances, #evaluations, #expansions)
 * on tie-free inputs is a statement about two differently shaped programs (tests/test_oracle_faiss.py).
 * The reference's own call into this code: index.search(n, x, k, D, I, params) with
 * SearchParametersHNSW{efSearch, check_relative_distance, ...} -- leann_backend_hnsw/hnsw_backend.py:203-248; the
 * fork (github.com/yichuan-w/faiss, absent: .gitmodules:1-6) adds beam_size / batch_size / PQ pruning on top of
 * this upstream routine, which is therefore the beam_size = 1, no-pruning semantics.
 *
 * The distance computer is lm_oracle.c's canonical orc_dist (declared below), so distances carry identical bits
 * and every comparison takes the same branch in both programs; faiss' own SIMD reduction order is not restated
 * (distances are compared at 1e-4 against the numpy formula elsewhere).
 *
 * Build: oracle/Makefile links this file into liblm_oracle.so.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_METRIC_IP 0
#define ORC_METRIC_L2 1

typedef struct {
    int64_t N;
    int32_t D, Dp, max_level, entry_point, metric;
    const uint64_t *node_offsets;
    const uint64_t *level_ptr;
    const int32_t *neighbors;
    const int32_t *levels;
} orc_graph; /* same layout as lm_oracle.c */

float orc_dist(const float *e, const float *q, int32_t Dp, int32_t metric); /* lm_oracle.c: -IP or squared L2 */

typedef struct {
    int64_t ndis;       /* distance evaluations at level 0 (faiss HNSWStats.ndis) incl. the seed */
    int64_t ndis_upper; /* evaluations of the greedy descent incl. the entry point */
    int64_t nstep;      /* level-0 expansions (faiss nstep / HNSWStats.nhops) */
} orcf_stats;

/* ---- faiss/utils/Heap.h: binary heap on parallel arrays, 1-based inside; CMax::cmp2(a1,a2,b1,b2) =
 *      (a1 > a2) || (a1 == a2 && b1 > b2) --------------------------------------------------------------- */
static inline int cmax_cmp2(float a1, float a2, int32_t b1, int32_t b2) { return (a1 > a2) || (a1 == a2 && b1 > b2); }

static void heap_pop(size_t k, float *bh_val, int32_t *bh_ids) {
    bh_val--; /* 1-based indexing for easier node->child translation */
    bh_ids--;
    float val = bh_val[k];
    int32_t id = bh_ids[k];
    size_t i = 1, i1, i2;
    for (;;) {
        i1 = i << 1;
        i2 = i1 + 1;
        if (i1 > k) break;
        if (i2 == k + 1 || cmax_cmp2(bh_val[i1], bh_val[i2], bh_ids[i1], bh_ids[i2])) {
            if (cmax_cmp2(val, bh_val[i1], id, bh_ids[i1])) break;
            bh_val[i] = bh_val[i1];
            bh_ids[i] = bh_ids[i1];
            i = i1;
        } else {
            if (cmax_cmp2(val, bh_val[i2], id, bh_ids[i2])) break;
            bh_val[i] = bh_val[i2];
            bh_ids[i] = bh_ids[i2];
            i = i2;
        }
    }
    bh_val[i] = bh_val[k];
    bh_ids[i] = bh_ids[k];
}

static void heap_push(size_t k, float *bh_val, int32_t *bh_ids, float val, int32_t id) {
    bh_val--;
    bh_ids--;
    size_t i = k, i_father;
    while (i > 1) {
        i_father = i >> 1;
        if (!cmax_cmp2(val, bh_val[i_father], id, bh_ids[i_father])) break; /* the child is below its father */
        bh_val[i] = bh_val[i_father];
        bh_ids[i] = bh_ids[i_father];
        i = i_father;
    }
    bh_val[i] = val;
    bh_ids[i] = id;
}

static void heap_replace_top(size_t k, float *bh_val, int32_t *bh_ids, float val, int32_t id) {
    bh_val--;
    bh_ids--;
    size_t i = 1, i1, i2;
    for (;;) {
        i1 = i << 1;
        i2 = i1 + 1;
        if (i1 > k) break;
        if (i2 == k + 1 || cmax_cmp2(bh_val[i1], bh_val[i2], bh_ids[i1], bh_ids[i2])) {
            if (cmax_cmp2(val, bh_val[i1], id, bh_ids[i1])) break;
            bh_val[i] = bh_val[i1];
            bh_ids[i] = bh_ids[i1];
            i = i1;
        } else {
            if (cmax_cmp2(val, bh_val[i2], id, bh_ids[i2])) break;
            bh_val[i] = bh_val[i2];
            bh_ids[i] = bh_ids[i2];
            i = i2;
        }
    }
    bh_val[i] = val;
    bh_ids[i] = id;
}

/* heap_reorder: the k-heap (pre-filled with neutral (+inf, -1) entries by heap_heapify) -> ascending array, the
 * entries that are still neutral (id == -1) moved behind the real ones */
static void maxheap_reorder(size_t k, float *bh_val, int32_t *bh_ids) {
    size_t i, ii;
    for (i = 0, ii = 0; i < k; i++) {
        /* top element should be put at the end of the list */
        float val = bh_val[0];
        int32_t id = bh_ids[0];
        /* boundary case: we will over-ride this value if not a true element */
        heap_pop(k - i, bh_val, bh_ids);
        bh_val[k - ii - 1] = val;
        bh_ids[k - ii - 1] = id;
        if (id != -1) ii++;
    }
    memmove(bh_val, bh_val + k - ii, ii * sizeof(*bh_val));
    memmove(bh_ids, bh_ids + k - ii, ii * sizeof(*bh_ids));
    for (; ii < k; ii++) {
        bh_val[ii] = INFINITY; /* CMax::neutral() */
        bh_ids[ii] = -1;
    }
}

/* HeapBlockResultHandler<CMax>::SingleResultHandler::add_result (faiss/impl/ResultHandler.h): the result heap always
 * holds k entries, neutral ones until real ones displace them; threshold = heap top */
static inline void res_add(int k, float *D, int32_t *I, float d, int32_t id) {
    if (d < D[0]) heap_replace_top((size_t)k, D, I, d, id);
}

/* ---- HNSW::MinimaxHeap ---------------------------------------------------------------------------- */
typedef struct {
    int n;      /* capacity */
    int k;      /* slots in use (valid + popped) */
    int nvalid; /* slots with id != -1 */
    int32_t *ids;
    float *dis;
} minimax_heap;

static void mm_push(minimax_heap *h, int32_t i, float v) {
    if (h->k == h->n) {
        if (v >= h->dis[0]) return;
        if (h->ids[0] != -1) --h->nvalid;
        heap_pop((size_t)h->k--, h->dis, h->ids);
    }
    heap_push((size_t)++h->k, h->dis, h->ids, v, i);
    ++h->nvalid;
}

static int32_t mm_pop_min(minimax_heap *h, float *vmin_out) {
    /* returns min. This is an O(n) operation */
    int i = h->k - 1;
    while (i >= 0) {
        if (h->ids[i] != -1) break;
        i--;
    }
    if (i == -1) return -1;
    int imin = i;
    float vmin = h->dis[i];
    i--;
    while (i >= 0) {
        if (h->ids[i] != -1 && h->dis[i] < vmin) {
            vmin = h->dis[i];
            imin = i;
        }
        i--;
    }
    if (vmin_out) *vmin_out = vmin;
    int32_t ret = h->ids[imin];
    h->ids[imin] = -1;
    --h->nvalid;
    return ret;
}

static int mm_count_below(const minimax_heap *h, float thresh) {
    int n_below = 0;
    for (int i = 0; i < h->k; i++)
        if (h->dis[i] < thresh) n_below++;
    return n_below;
}

/* ---- graph access: HNSW::neighbor_range on the compact-CSR arrays (convert_to_csr.py:494-548) ---------- */
static inline void neighbor_range(const orc_graph *g, int32_t no, int level, size_t *begin, size_t *end) {
    uint64_t p = g->node_offsets[no] + (uint64_t)level;
    *begin = (size_t)g->level_ptr[p];
    *end = (size_t)g->level_ptr[p + 1];
}

typedef struct {
    const orc_graph *g;
    const float *table; /* N x Dp */
    const float *q;     /* Dp */
} dist_computer;
static inline float qdis(const dist_computer *dc, int32_t v) {
    return orc_dist(dc->table + (size_t)v * dc->g->Dp, dc->q, dc->g->Dp, dc->g->metric);
}

/* ---- greedy_update_nearest ------------------------------------------------------------------------ */
static void greedy_update_nearest(const dist_computer *dc, int level, int32_t *nearest, float *d_nearest, orcf_stats *st) {
    for (;;) {
        int32_t prev_nearest = *nearest;
        size_t begin, end;
        neighbor_range(dc->g, *nearest, level, &begin, &end);
        for (size_t i = begin; i < end; i++) {
            int32_t v = dc->g->neighbors[i];
            if (v < 0) break;
            float dis = qdis(dc, v);
            st->ndis_upper++;
            if (dis < *d_nearest) {
                *nearest = v;
                *d_nearest = dis;
            }
        }
        if (*nearest == prev_nearest) return;
    }
}

/* ---- search_from_candidates (level 0, bounded queue) ------------------------------------------------ */
static int search_from_candidates(const dist_computer *dc, int k, int32_t *I, float *D, minimax_heap *candidates,
                                  uint8_t *vt, int efSearch, int do_dis_check, orcf_stats *st) {
    for (int i = 0; i < candidates->k; i++) { /* candidates.size() == k here: nothing popped yet */
        int32_t v1 = candidates->ids[i];
        float d = candidates->dis[i];
        res_add(k, D, I, d, v1);
        vt[v1] = 1;
    }
    int nstep = 0;
    while (candidates->nvalid > 0) {
        float d0 = 0;
        int32_t v0 = mm_pop_min(candidates, &d0);
        if (do_dis_check) {
            /* tricky stopping condition: there are more that ef distances that are processed already that are
             * smaller than d0 */
            int n_dis_below = mm_count_below(candidates, d0);
            if (n_dis_below >= efSearch) break;
        }
        size_t begin, end;
        neighbor_range(dc->g, v0, 0, &begin, &end);
        for (size_t j = begin; j < end; j++) {
            int32_t v1 = dc->g->neighbors[j];
            if (v1 < 0) break;
            if (vt[v1]) continue;
            vt[v1] = 1;
            st->ndis++;
            float d = qdis(dc, v1);
            res_add(k, D, I, d, v1);
            mm_push(candidates, v1, d);
        }
        nstep++;
        st->nstep++;
        if (!do_dis_check && nstep > efSearch) break;
    }
    return 0;
}

/* ---- HNSW::search + IndexHNSW::search, one query after the other -------------------------------------
 * table: N x Dp fp32 (zero padded), queries: B x Dp.  out_ids B x k int64 (-1 = unfilled), out_dist B x k
 * (l2: squared L2 ascending; ip: +IP descending, unfilled -inf -- faiss negates back after the search). */
int orcf_search(const orc_graph *g, const float *table, const float *queries, int32_t B, int32_t k, int32_t efSearch,
                int32_t check_relative_distance, int64_t *out_ids, float *out_dist, orcf_stats *stats) {
    orcf_stats st = {0, 0, 0};
    const int ef = efSearch > k ? efSearch : k; /* std::max(efSearch, k) */
    uint8_t *vt = (uint8_t *)calloc((size_t)(g->N > 0 ? g->N : 1), 1);
    int32_t *I = (int32_t *)malloc(sizeof(int32_t) * (size_t)k);
    float *D = (float *)malloc(sizeof(float) * (size_t)k);
    minimax_heap cand;
    cand.n = ef;
    cand.ids = (int32_t *)malloc(sizeof(int32_t) * (size_t)ef);
    cand.dis = (float *)malloc(sizeof(float) * (size_t)ef);
    for (int32_t qi = 0; qi < B; ++qi) {
        for (int i = 0; i < k; ++i) { /* heap_heapify: neutral values */
            D[i] = INFINITY;
            I[i] = -1;
        }
        if (g->N > 0 && g->entry_point >= 0) {
            dist_computer dc = {g, table, queries + (size_t)qi * g->Dp};
            /* greedy search on upper levels */
            int32_t nearest = g->entry_point;
            float d_nearest = qdis(&dc, nearest);
            st.ndis_upper++;
            for (int level = g->max_level; level >= 1; level--) greedy_update_nearest(&dc, level, &nearest, &d_nearest, &st);
            cand.k = cand.nvalid = 0;
            mm_push(&cand, nearest, d_nearest);
            search_from_candidates(&dc, k, I, D, &cand, vt, efSearch, check_relative_distance, &st);
            memset(vt, 0, (size_t)g->N); /* vt.advance() */
        }
        maxheap_reorder((size_t)k, D, I);
        for (int i = 0; i < k; ++i) {
            out_ids[(size_t)qi * k + i] = I[i];
            /* similarity metric: distances were negated for the search; negate back (unfilled: -inf) */
            out_dist[(size_t)qi * k + i] = g->metric == ORC_METRIC_L2 ? D[i] : -D[i];
        }
    }
    free(vt);
    free(I);
    free(D);
    free(cand.ids);
    free(cand.dis);
    if (stats) *stats = st;
    return 0;
}
