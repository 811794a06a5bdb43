/*
 * abi_host.c -- a plain C host of libleann_mi355x.so (no Python, no torch): proves that
 * include/leann_mi355x.h is a self-contained C ABI.
 *   - compiles as C11 with gcc against the public header only;
 *   - without a GPU: argument validation and the loud LM_EHIP failure;
 *   - with a GPU: the hand-traced 8-node line graph of tests/test_oracle.py searched in stored-embedding
 *     mode through lm_index_search (host pointers, as faiss' index.search takes them,
 *     hnsw_backend.py:241-248) -- known answer [6, 7, 5, 4] with squared-L2 distances 0.16, 0.36, 1.96, 5.76.
 * Build + run: tests/test_abi.py::test_plain_c_host.  Exit code 0 = pass; prints "GPU-PATH-OK" when the
 * device part ran.
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "leann_mi355x.h"

#define CHECK(cond)                                                                   \
    do {                                                                              \
        if (!(cond)) {                                                                \
            fprintf(stderr, "abi_host: check failed at line %d: %s (last error: %s)\n", __LINE__, #cond, lm_last_error()); \
            return 1;                                                                 \
        }                                                                             \
    } while (0)

int main(void) {
    enum { N = 8, D = 64 };
    /* line graph: level-0 links i <-> i+1; node 3 (entry) and node 6 also live on level 1, linked to each other */
    int32_t levels[N] = {1, 1, 1, 2, 1, 1, 2, 1};
    uint64_t node_offsets[N + 1], level_ptr[N + 2 + N]; /* sum(levels+1) = 8*2 + 2 = 18 */
    int32_t neighbors[2 * (N - 1) + 2];
    int64_t np = 0, ne = 0;
    for (int i = 0; i < N; ++i) {
        node_offsets[i] = (uint64_t)np;
        level_ptr[np++] = (uint64_t)ne;
        if (i > 0) neighbors[ne++] = i - 1;
        if (i < N - 1) neighbors[ne++] = i + 1;
        if (levels[i] == 2) {
            level_ptr[np++] = (uint64_t)ne;
            neighbors[ne++] = (i == 3) ? 6 : 3;
        }
        level_ptr[np++] = (uint64_t)ne;
    }
    node_offsets[N] = (uint64_t)np;
    CHECK(np == 18 && ne == 16);

    lm_search_params prm;
    lm_search_params_default(&prm);
    CHECK(prm.efSearch == 64 && prm.beam_size == 1 && prm.check_relative_distance == 1 && prm.recompute == 1);
    CHECK(strlen(lm_version()) > 0);
    CHECK(lm_abi_revision() == LM_ABI_REVISION); /* the library speaks the header this host was compiled against */
    CHECK(prm.batch_size == 0);                  /* dynamic batching off by default (the reference's default, hnsw_backend.py:163) */

    lm_index *idx = NULL;
    /* argument validation happens before any device use */
    CHECK(lm_index_create_from_csr(N, 0, LM_METRIC_L2, node_offsets, level_ptr, np, neighbors, ne, levels, 3, 1, 0, &idx) == LM_EINVAL);
    CHECK(lm_index_create_from_csr(N, D, 7, node_offsets, level_ptr, np, neighbors, ne, levels, 3, 1, 0, &idx) == LM_EINVAL);
    CHECK(lm_index_create_from_csr(N, D, LM_METRIC_L2, node_offsets, level_ptr, np - 1, neighbors, ne, levels, 3, 1, 0, &idx) == LM_EFORMAT);
    CHECK(lm_index_create_from_csr(N, D, LM_METRIC_L2, node_offsets, level_ptr, np, neighbors, ne, levels, 5, 1, 0, &idx) == LM_EFORMAT);
    CHECK(lm_index_read("/nonexistent/x.index", 0, &idx) == LM_ENOENT);

    int rc = lm_index_create_from_csr(N, D, LM_METRIC_L2, node_offsets, level_ptr, np, neighbors, ne, levels, 3, 1, 0, &idx);
    if (lm_device_count() == 0) {
        CHECK(rc == LM_EHIP && idx == NULL && strstr(lm_last_error(), "no HIP device") != NULL);
        printf("NO-GPU-PATH-OK\n");
        return 0;
    }
    CHECK(rc == LM_OK && idx != NULL);
    lm_index_info_t info;
    CHECK(lm_index_info(idx, &info) == LM_OK);
    CHECK(info.ntotal == N && info.d == D && info.d_padded == 64 && info.max_level == 1 && info.entry_point == 3 &&
          info.max_degree0 == 2 && info.max_degree_up == 1 && info.n_neighbors == 16 && !info.has_table);

    float table[N * D];
    memset(table, 0, sizeof(table));
    for (int i = 0; i < N; ++i) table[i * D] = (float)i; /* 1-d coordinates 0..7 embedded in 64-d */
    float q[D];
    memset(q, 0, sizeof(q));
    q[0] = 6.4f;
    float dist[4];
    int64_t labels[4];
    prm.efSearch = 4;
    prm.recompute = 0;
    /* pruned index without stored embeddings: must refuse, not fall back */
    CHECK(lm_index_search(idx, 1, q, 4, dist, labels, &prm) == LM_ESTATE);
    CHECK(lm_index_attach_table(idx, table, LM_DTYPE_F32, N, D, 0) == LM_OK);
    for (int persistent = 1; persistent >= 0; --persistent) {
        CHECK(lm_index_set_option(idx, "persistent_table", persistent) == LM_OK);
        CHECK(lm_index_search(idx, 1, q, 4, dist, labels, &prm) == LM_OK);
        const int64_t want[4] = {6, 7, 5, 4};
        const float wd[4] = {0.16f, 0.36f, 1.96f, 5.76f};
        for (int i = 0; i < 4; ++i) CHECK(labels[i] == want[i] && fabsf(dist[i] - wd[i]) < 1e-4f);
        lm_search_stats st;
        CHECK(lm_index_get_stats(idx, &st) == LM_OK);
        /* seed + 2 upper-level rounds + pops 6,7,5,4,3(ef=4 keeps 4 best; 3 is evaluated when 4 expands) */
        CHECK(st.ndis >= 6 && st.nexpand >= 4 && st.nrounds >= 6);
    }
    /* recompute requested without a provider: loud error */
    prm.recompute = 1;
    CHECK(lm_index_search(idx, 1, q, 4, dist, labels, &prm) == LM_ESTATE);
    lm_index_free(idx);
    printf("GPU-PATH-OK\n");
    return 0;
}
