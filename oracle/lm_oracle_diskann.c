/*
 * lm_oracle_diskann.c -- SECOND, INDEPENDENT CPU oracle of the DiskANN-style path: a literal transcription of upstream
 * DiskANN's beam search (microsoft/DiskANN, v0.7 series: include/neighbor.h -- struct Neighbor, class NeighborPriorityQueue
 * {insert, closest_unexpanded, has_unexpanded_node}; src/pq_flash_index.cpp -- PQFlashIndex<T>::cached_beam_search: retset of
 * capacity l_search, visited set, frontier of at most beam_width unexpanded nodes per hop, PQ distances of the neighbours of every
 * frontier node, full_retset of EXACT distances of every expanded node, final std::sort of full_retset and the first k_search).
 *
 * THIS FILE IS TEST INFRASTRUCTURE, NOT PRODUCT CODE (same rule as lm_oracle.c: only tests/ load it).
 *
 * Why it exists: lm_oracle_pq.c restates the DiskANN-style traversal in SET semantics (the form the HIP kernels match bit for bit:
 * sorted list, "evaluate the round first, insert after", lock-step over the batch).  This file keeps upstream's ORIGINAL structures
 * -- a sorted array with an expanded flag per entry and a cursor to the first unexpanded one, binary-search insertion with the
 * duplicate-id check, neighbours inserted one by one while the frontier is being processed, one query at a time -- so that
 *     orc_pq_search  ==  orcd_search (rerank_final_list_only = 1)        ids, distances, #expansions, #PQ evaluations
 * on tie-free inputs is a statement about two differently shaped programs (tests/test_oracle_diskann.py), as
 * lm_oracle_faiss.c is for the HNSW path.  PARITY STATUS stays "unpinned": the reference's own implementation is the fork
 * github.com/yichuan-w/DiskANN (packages/leann-backend-diskann/third_party/DiskANN, empty; .gitmodules:4-6), reached through
 * StaticDiskFloatIndex.batch_search (leann_backend_diskann/diskann_backend.py:453-467); what the fork changed on top of upstream is
 * known only from that call site and its comment (:444-449).
 *
 * ONE SEMANTIC DIFFERENCE this transcription makes visible (and the tests quantify).  Upstream ranks the final answer over
 * full_retset = EVERY node the search expanded (their full-precision coordinates came with the sector reads for free).  The
 * reference's deferred fetch is described as "a single final rerank via deferred fetch (fetch embeddings for the final candidate
 * set only)" (diskann_backend.py:444-449); the product and lm_oracle_pq.c rerank the FINAL CANDIDATE LIST (the <= L entries the
 * traversal ends with, all of them expanded).  full_retset is a superset of that list (nodes expanded early and later pushed out of
 * the list stay in it), so upstream's top-k can only be equal or better in exact distance.  rerank_final_list_only selects which of
 * the two sets is ranked: 1 = the product's (must equal lm_oracle_pq.c), 0 = upstream's.
 *
 * Not restated: the node cache, the sector reads / io_limit, use_reorder_data, medoid selection among several (one entry point
 * here), upstream's own SIMD distance kernels (the canonical orc_dist / orc_pq_lut / orc_pq_adc of the other oracle files are
 * used, so distances carry identical bits and every comparison takes the same branch in both programs).
 *
 * Build: oracle/Makefile links this file into liblm_oracle.so.
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

typedef struct {
    int32_t m;
    int32_t dsub;
    const float *codebooks;
    const uint8_t *codes;
    const int32_t *chunk_off;
} orc_pq; /* same layout as lm_oracle_pq.c */

float orc_dist(const float *e, const float *q, int32_t Dp, int32_t metric);    /* lm_oracle.c */
void orc_pq_lut(const orc_pq *pq, const float *q, int32_t metric, float *lut); /* lm_oracle_pq.c */
float orc_pq_adc(const orc_pq *pq, const float *lut, int64_t v);               /* lm_oracle_pq.c */

typedef struct {
    int64_t n_cmps;          /* PQ evaluations of newly visited neighbours (upstream: stats->n_cmps), the medoid's not included */
    int64_t n_hops;          /* iterations of the outer loop, summed over the queries */
    int64_t n_expanded;      /* nodes expanded = entries of full_retset, summed over the queries */
    int64_t max_hops;        /* largest per-query hop count */
    int64_t n_final_differs; /* queries whose final candidate list is a STRICT subset of full_retset */
} orcd_stats;

/* ---- include/neighbor.h ---------------------------------------------------------------------------------------------------- */
typedef struct {
    uint32_t id;
    float distance;
    int expanded;
} Neighbor;

/* bool operator<(const Neighbor &other) const { return distance < other.distance || (distance == other.distance && id < other.id); } */
static inline int nbr_less(const Neighbor *a, const Neighbor *b) {
    return a->distance < b->distance || (a->distance == b->distance && a->id < b->id);
}

typedef struct {
    size_t _size, _capacity, _cur;
    Neighbor *_data; /* _capacity + 1 entries */
} NeighborPriorityQueue;

/* Inserts the item ordered into the set up to the sets capacity.  The item will be dropped if it is the same id as an existing set
 * item or it has a greater distance than the final item in the set. */
static void npq_insert(NeighborPriorityQueue *s, const Neighbor *nbr) {
    if (s->_size == s->_capacity && nbr_less(&s->_data[s->_size - 1], nbr)) return;
    size_t lo = 0, hi = s->_size;
    while (lo < hi) {
        size_t mid = (lo + hi) >> 1;
        if (nbr_less(nbr, &s->_data[mid])) {
            hi = mid;
        } else if (s->_data[mid].id == nbr->id) { /* Make sure the same id isn't inserted into the set */
            return;
        } else {
            lo = mid + 1;
        }
    }
    if (lo < s->_capacity) memmove(&s->_data[lo + 1], &s->_data[lo], (s->_size - lo) * sizeof(Neighbor));
    s->_data[lo].id = nbr->id;
    s->_data[lo].distance = nbr->distance;
    s->_data[lo].expanded = 0;
    if (s->_size < s->_capacity) s->_size++;
    if (lo < s->_cur) s->_cur = lo;
}

static Neighbor npq_closest_unexpanded(NeighborPriorityQueue *s) {
    s->_data[s->_cur].expanded = 1;
    size_t pre = s->_cur;
    while (s->_cur < s->_size && s->_data[s->_cur].expanded) s->_cur++;
    return s->_data[pre];
}

static inline int npq_has_unexpanded_node(const NeighborPriorityQueue *s) { return s->_cur < s->_size; }

static int cmp_neighbor(const void *a, const void *b) {
    const Neighbor *x = (const Neighbor *)a, *y = (const Neighbor *)b;
    return nbr_less(x, y) ? -1 : (nbr_less(y, x) ? 1 : 0);
}

/* ---- src/pq_flash_index.cpp: cached_beam_search, one query ------------------------------------------------------------------- */
static void cached_beam_search(const orc_graph *g, const orc_pq *pq, const float *table, const float *query, int32_t k_search,
                               int32_t l_search, int32_t beam_width, int rerank_final_list_only, float *lut, uint32_t *visited,
                               NeighborPriorityQueue *retset, Neighbor *full_retset, uint32_t *frontier, int64_t *indices,
                               float *distances, orcd_stats *stats) {
    const int32_t Dp = g->Dp;
    /* query <-> PQ chunk centers distances: pq_table.populate_chunk_distances(query_rotated, pq_dists) */
    orc_pq_lut(pq, query, g->metric, lut);
    memset(visited, 0, sizeof(uint32_t) * (size_t)((g->N + 31) / 32));
    retset->_size = 0;
    retset->_cur = 0;
    retset->_capacity = (size_t)l_search; /* retset.reserve(l_search) */
    size_t n_full = 0;

    /* best medoid: one entry point here */
    const uint32_t best_medoid = (uint32_t)g->entry_point;
    {
        Neighbor nn = {best_medoid, orc_pq_adc(pq, lut, best_medoid), 0}; /* compute_dists(&best_medoid, 1, dist_scratch) */
        npq_insert(retset, &nn);
        visited[best_medoid >> 5] |= 1u << (best_medoid & 31); /* visited.insert(best_medoid) */
    }
    uint32_t cmps = 0, hops = 0;

    while (npq_has_unexpanded_node(retset)) {
        /* clear iteration state */
        uint32_t n_frontier = 0, num_seen = 0;
        /* find new beam */
        while (npq_has_unexpanded_node(retset) && n_frontier < (uint32_t)beam_width && num_seen < (uint32_t)beam_width) {
            Neighbor nbr = npq_closest_unexpanded(retset);
            num_seen++;
            frontier[n_frontier++] = nbr.id; /* (no node cache: every beam node is read) */
        }
        /* process each frontier nhood - compute distances to unvisited nodes */
        for (uint32_t f = 0; f < n_frontier; ++f) {
            const uint32_t node = frontier[f];
            /* exact distance of the expanded node from its full-precision coordinates */
            const float cur_expanded_dist = orc_dist(table + (size_t)node * Dp, query, Dp, g->metric);
            full_retset[n_full].id = node;
            full_retset[n_full].distance = cur_expanded_dist;
            full_retset[n_full].expanded = 1;
            n_full++;
            const uint64_t p = g->node_offsets[node]; /* level-0 list of the flat graph */
            for (uint64_t m = g->level_ptr[p]; m < g->level_ptr[p + 1]; ++m) {
                const uint32_t id = (uint32_t)g->neighbors[m];
                const uint32_t bit = 1u << (id & 31);
                if (!(visited[id >> 5] & bit)) { /* visited.insert(id).second */
                    visited[id >> 5] |= bit;
                    cmps++;
                    Neighbor nn = {id, orc_pq_adc(pq, lut, id), 0}; /* dist_scratch[m] */
                    npq_insert(retset, &nn);
                }
            }
        }
        hops++;
    }

    const size_t n_expanded = n_full; /* every frontier entry was expanded exactly once */
    int strict_subset = 0;
    if (rerank_final_list_only) {
        /* the product's deferred fetch: exact distances of the FINAL candidate list only (every entry of it is expanded, hence a
         * member of full_retset: take its exact distance from there) */
        size_t n_keep = 0;
        for (size_t i = 0; i < n_full; ++i) {
            int in_list = 0;
            for (size_t j = 0; j < retset->_size && !in_list; ++j) in_list = retset->_data[j].id == full_retset[i].id;
            if (in_list) full_retset[n_keep++] = full_retset[i];
        }
        strict_subset = n_keep < n_full;
        n_full = n_keep;
    } else {
        strict_subset = retset->_size < n_full;
    }
    /* re-sort by distance */
    qsort(full_retset, n_full, sizeof(Neighbor), cmp_neighbor);
    /* copy k_search values */
    for (int32_t i = 0; i < k_search; ++i) {
        if ((size_t)i < n_full) {
            indices[i] = (int64_t)full_retset[i].id;
            /* internal distance is squared L2 or -IP; inner product goes out as +IP (upstream: distances[i] = -distances[i]) */
            distances[i] = g->metric == ORC_METRIC_L2 ? full_retset[i].distance : -full_retset[i].distance;
        } else {
            indices[i] = -1;
            distances[i] = g->metric == ORC_METRIC_L2 ? INFINITY : -INFINITY;
        }
    }
    stats->n_cmps += cmps;
    stats->n_hops += hops;
    stats->n_expanded += (int64_t)n_expanded;
    if ((int64_t)hops > stats->max_hops) stats->max_hops = hops;
    stats->n_final_differs += strict_subset;
}

/* queries: B x Dp (zero padded); table: N x Dp exact embeddings.  One query at a time, as upstream's per-thread call. */
int orcd_search(const orc_graph *g, const orc_pq *pq, const float *table, const float *queries, int32_t B, int32_t k, int32_t L,
                int32_t beam_width, int32_t rerank_final_list_only, int64_t *out_ids, float *out_dist, orcd_stats *stats) {
    orcd_stats st = {0, 0, 0, 0, 0};
    if (!g || !pq || !table || !queries || !out_ids || !out_dist || k <= 0 || B < 0) return -1;
    const float fill = g->metric == ORC_METRIC_L2 ? INFINITY : -INFINITY;
    for (int64_t i = 0; i < (int64_t)B * k; ++i) {
        out_ids[i] = -1;
        out_dist[i] = fill;
    }
    if (g->N == 0 || B == 0 || g->entry_point < 0) {
        if (stats) *stats = st;
        return 0;
    }
    const int32_t l_search = L > k ? L : k; /* the product / lm_oracle_pq.c: max(complexity, k); upstream requires l_search >= k_search */
    const int32_t bw = beam_width < 1 ? 1 : beam_width;
    float *lut = (float *)malloc(sizeof(float) * (size_t)pq->m * 256);
    uint32_t *visited = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)((g->N + 31) / 32));
    NeighborPriorityQueue retset = {0, 0, 0, (Neighbor *)malloc(sizeof(Neighbor) * ((size_t)l_search + 1))};
    Neighbor *full_retset = (Neighbor *)malloc(sizeof(Neighbor) * (size_t)g->N); /* a node is expanded at most once */
    uint32_t *frontier = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)bw);
    for (int32_t q = 0; q < B; ++q)
        cached_beam_search(g, pq, table, queries + (size_t)q * g->Dp, k, l_search, bw, rerank_final_list_only, lut, visited, &retset,
                           full_retset, frontier, out_ids + (size_t)q * k, out_dist + (size_t)q * k, &st);
    free(lut);
    free(visited);
    free(retset._data);
    free(full_retset);
    free(frontier);
    if (stats) *stats = st;
    return 0;
}
