// lm_search.hip -- HBM-resident compact-CSR graph + selective-recompute beam search for MI355X (gfx950).
// Kernels (details at each definition):
//   k_init / k_finalize / k_stats   per-query state reset, pool -> (labels, distances), end-of-search totals
//   k_expand      one wave per query: (pop, neighbour) pairs flattened over the lanes -> CSR neighbour gather ->
//                 visited test-and-set (per-query bitmap) -> per-query new-list; marks the round's dedup bitmap
//   k_uniq_count / k_uniq_emit      round bitmap -> SORTED unique node list + per-word ranks
//   k_pq_lut_all / k_prune          two-level search: PQ-ADC ranking of the new-list, approximate queue, selection
//   k_update<NCH,L2,F16,MODE,NT>    fused gather + distance + beam update, one workgroup (or wave) per query:
//                 query slice in registers, 16-lane row dot products with the canonical reduction (bit-exact with
//                 oracle/lm_oracle.c:orc_dist), new keys sorted + rank-merged into the ef-pool in LDS, next pops
//   k_memo_append per-call embedding memo (recompute_memo)
//   k_search_table persistent stored-embedding search: a query's whole traversal in one workgroup, one launch/batch
//   lm_pq_impl.h  DiskANN-style path: k_pq_traverse (persistent PQ-ADC traversal), k_pq_rerank
// Algorithm contract: oracle/lm_oracle.c header (set semantics under the (dist,id) total order).
// Reference call site replaced: index.search(...) leann_backend_hnsw/hnsw_backend.py:241-248.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <type_traits>

#include <hip/hip_fp16.h>

#include "lm_internal.h"

namespace lm {

static thread_local std::string g_err;
void set_error(const std::string& msg) { g_err = msg; }


}  // namespace lm

#include "lm_device_types.h"
#include "lm_beam_common.h"
#include "lm_kernels_expand.h"
#include "lm_kernels_prune.h"
#include "lm_kernels_update.h"
#include "lm_kernels_persist.h"
#include "lm_kernels_misc.h"


// =============================================================================================
// host side
// =============================================================================================
using namespace lm;

struct lm_index {
    int device = 0;
    int64_t N = 0;
    int32_t D = 0, Dp = 0, metric = 0, entry_point = -1, max_level = -1;
    int32_t maxdeg0 = 0, maxdeg_up = 0;
    int64_t n_neighbors = 0, n_level_ptr = 0;
    uint64_t* d_node_offsets = nullptr;
    uint64_t* d_level_ptr = nullptr;
    int32_t* d_neighbors = nullptr;
    L0Range* d_l0 = nullptr;
    // PQ (DiskANN-style path)
    int32_t pq_m = 0;
    float* d_pq_codebooks = nullptr;
    uint8_t* d_pq_codes = nullptr;
    int32_t* d_pq_chunk_off = nullptr;
    // lm_index_search (host pointers): query / result staging, grown on demand, freed with the index
    float* d_stage_x = nullptr;
    float* d_stage_d = nullptr;
    int64_t* d_stage_l = nullptr;
    size_t stage_x_bytes = 0, stage_d_bytes = 0, stage_l_bytes = 0;
    unsigned long long* d_pq_nadc = nullptr;
    int32_t* d_pq_rounds = nullptr;
    int64_t pq_cap = 0;
    float* d_lut = nullptr;  // two-level search: B x m x 256
    unsigned long long* d_tstamp = nullptr;  // profiling: 2 x B stamps + [2] accumulator at the end
    int64_t tstamp_cap = 0;
    double span_ms = 0;
    int64_t span_launches = 0;
    int64_t lut_cap = 0;
    // stored embeddings
    void* d_table = nullptr;
    bool table_owned = false;
    int32_t table_dtype = LM_DTYPE_F32;
    // provider
    lm_provider_fn provider = nullptr;
    void* provider_user = nullptr;
    lm_recompute* native_rc = nullptr;  // lm_index_set_recompute: the built-in provider (provider == lm_recompute_provider, user == this)
    hipStream_t stream = nullptr;
    // workspace
    WsDev ws{};
    int32_t ws_B = 0, ws_ef = 0, ws_W = 0, ws_maxnew = 0, ws_spec = 0;  // (ws_maxnew covers the dynamic-batching target: ensure_ws)
    int64_t ws_ucap = 0;
    int single_query_direct = 0;  // option "single_query_direct": a one-query recompute pass hands its new-list to the provider as it is (no k_uniq_*)
    int speculate = 0;            // option "speculate": candidates whose neighbours a small-batch round embeds ahead of time (k_speculate); 0 = off
    int speculate_max_batch = 2;  // option "speculate_max_batch": ... for calls of at most this many queries (larger rounds are not launch bound)
    int32_t* d_memo_slot = nullptr;
    float* d_memo = nullptr;
    int64_t memo_cap = 0;
    int64_t memo_initial_rows = 0;  // option "memo_initial_rows": first allocation of the per-call memo (0 = max(65536, 1024 per query)); it doubles on demand up to N rows
    int32_t* d_hub_slot_init = nullptr;  // N: slot of every hub node, -1 elsewhere (hub-embedding cache)
    int64_t hub_n = 0;
    std::vector<void*> ws_allocs;
    float* d_qpad = nullptr;
    int64_t qpad_cap = 0;
    unsigned long long* h_counters = nullptr;  // pinned
    // stats / profiling
    lm_search_stats stats{};
    bool profiling = false;
    int update_variant = 0;  // 0: auto, 3: wave (64 lanes) per query, 4: workgroup (256 threads) per query   (1, 2: removed A/B forms)
    int persistent_table = 1;  // stored-embedding mode: one persistent launch per batch (0: lock-step rounds, for A/B)
    int pq_threads = 1024;     // workgroup width of the PQ traversal kernel (option "pq_threads": 256 / 512 / 1024)
    bool pq_rerank_expanded = false;  // option "pq_rerank_expanded": rerank every expanded node (upstream DiskANN's full_retset), not the final list
    int64_t pq_overflow = 0;   // queries (since the option was last set) whose expansions outgrew the record and fell back to the final list
    int persistent_wave = -1;  // persistent search: 1 = one wave per query, 0 = one 256-thread workgroup per query, -1 = auto
    int wave_maxnew = 0;     // auto rule threshold on beam x mean level-0 degree; 0 = never: on the 1M-chunk HNSW graph
                             // (max degree 64, mean 9.3) the workgroup form is 1.5x faster (profiles/r1_bench_default_1M_b2048.json
                             // vs r1_bench_default_1M.json), although the wave form wins on uniform-degree graphs
    double avg_degree0 = 0;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev_update, ev_expand, ev_provider;
    std::vector<hipEvent_t> ev_pool;
};

static int free_ws(lm_index* ix) {
    for (void* p : ix->ws_allocs) (void)hipFree(p);
    ix->ws_allocs.clear();
    ix->ws_B = ix->ws_ef = ix->ws_W = ix->ws_maxnew = ix->ws_spec = 0;
    ix->ws_ucap = 0;
    return 0;
}

template <typename T>
static int ws_alloc(lm_index* ix, T** p, size_t count) {
    void* v = nullptr;
    LM_HIP(hipMalloc(&v, std::max<size_t>(count, 1) * sizeof(T)));
    ix->ws_allocs.push_back(v);
    *p = (T*)v;
    return LM_OK;
}

static int ensure_ws(lm_index* ix, int32_t B, int32_t ef, int32_t W, bool prune = false, int32_t spec = 0, int32_t batch = 0) {
    int32_t maxnew = std::max({W * ix->maxdeg0, ix->maxdeg_up, 1});
    if (batch > 0) maxnew = std::max(maxnew, batch - 1 + ix->maxdeg0);  // dynamic batching: the last extra pop starts below `batch` and adds one list at most
    if (prune) maxnew = std::max(maxnew, (int32_t)AQ_CAP);
    if (B <= ix->ws_B && ef == ix->ws_ef && W == ix->ws_W && maxnew == ix->ws_maxnew && spec == ix->ws_spec) {
        ix->ws.B = B;
        return LM_OK;
    }
    free_ws(ix);
    WsDev& w = ix->ws;
    w.B = B; w.ef = ef; w.W = W; w.maxnew = maxnew;
    w.nw = (ix->N + 31) / 32;
    int rc;
#define A(ptr, cnt) if ((rc = ws_alloc(ix, &w.ptr, (cnt))) != LM_OK) return rc
    A(phase, B); A(level, B); A(cur_key, B); A(nsteps, B); A(npool, B); A(npop, B); A(nnew, B); A(ndis_q, B);
    A(naq, B); A(nadc_q, B); A(aq, (size_t)B * AQ_CAP);
    A(pop, (size_t)B * W); A(newid, (size_t)B * maxnew); A(pool, (size_t)B * ef);
    A(visited, (size_t)B * w.nw);
    A(rbm, w.nw); A(rbm_snap, w.nw); A(word_rank, w.nw);
    int ntiles = (int)((w.nw + UNIQ_TILE - 1) / UNIQ_TILE);
    A(tile_sum, std::max(ntiles, 1));
    ix->ws_ucap = std::min<int64_t>(ix->N, (int64_t)B * (std::max(maxnew, ef) + (int64_t)spec * ix->maxdeg0));  // + the speculative requests
    A(uniq, ix->ws_ucap);
    A(counters, C_NCOUNTERS);
#undef A
    LM_HIP(hipMemsetAsync(w.rbm, 0, w.nw * 4, ix->stream));
    ix->ws_B = B; ix->ws_ef = ef; ix->ws_W = W; ix->ws_maxnew = maxnew; ix->ws_spec = spec;
    return LM_OK;
}

// Rows of the embedding memo (hub cache rows first, then the rows a call appends).  Never more than N are needed: a node enters
// the memo at most once.  Growth keeps the first `keep` rows (geometric: a call that touches u distinct nodes copies < 2u rows).
static int ensure_memo_rows(lm_index* ix, int64_t need, int64_t keep) {
    if (need <= ix->memo_cap) return LM_OK;
    if (need > ix->N) LM_FAIL(LM_ESTATE, "internal: the embedding memo cannot need more rows than the index has nodes");
    const int64_t cap = std::min<int64_t>(ix->N, std::max<int64_t>({need, 2 * ix->memo_cap, (int64_t)65536}));
    float* grown = nullptr;
    if (hipMalloc((void**)&grown, (size_t)cap * ix->Dp * 4) != hipSuccess) {
        (void)hipGetLastError();
        LM_FAIL(LM_EHIP, "out of device memory for the per-call recompute memo (" + std::to_string((size_t)cap * ix->Dp * 4 >> 20) +
                             " MiB); search with recompute_memo = 0 or a smaller max_batch");
    }
    if (ix->d_memo) {
        if (keep > 0) LM_HIP(hipMemcpyAsync(grown, ix->d_memo, (size_t)keep * ix->Dp * 4, hipMemcpyDeviceToDevice, ix->stream));
        LM_HIP(hipStreamSynchronize(ix->stream));  // kernels of earlier rounds still read the old rows
        (void)hipFree(ix->d_memo);
    }
    ix->d_memo = grown;
    ix->memo_cap = cap;
    ix->ws.memo = grown;
    return LM_OK;
}

static hipEvent_t get_event(lm_index* ix) {
    hipEvent_t e;
    if (!ix->ev_pool.empty()) {
        e = ix->ev_pool.back();
        ix->ev_pool.pop_back();
        return e;
    }
    (void)hipEventCreate(&e);
    return e;
}

static double drain_events(lm_index* ix, std::vector<std::pair<hipEvent_t, hipEvent_t>>& v) {
    double ms = 0;
    for (auto& p : v) {
        float t = 0;
        if (hipEventElapsedTime(&t, p.first, p.second) == hipSuccess) ms += t;
        ix->ev_pool.push_back(p.first);
        ix->ev_pool.push_back(p.second);
    }
    v.clear();
    return ms;
}

struct EvScope {
    lm_index* ix;
    std::vector<std::pair<hipEvent_t, hipEvent_t>>* v;
    hipEvent_t a{}, b{};
    EvScope(lm_index* i, std::vector<std::pair<hipEvent_t, hipEvent_t>>* vec) : ix(i), v(vec) {
        if (ix->profiling) {
            a = get_event(ix);
            b = get_event(ix);
            (void)hipEventRecord(a, ix->stream);
        }
    }
    ~EvScope() {
        if (ix->profiling) {
            (void)hipEventRecord(b, ix->stream);
            v->push_back({a, b});
        }
    }
};

static int next_pow2(int x) {
    int p = 1;
    while (p < x) p <<= 1;
    return p;
}

template <bool L2, bool F16>
static int launch_update_nch(lm_index* ix, const UpdateArgs& a, size_t shmem) {
    dim3 grid(ix->ws.B), block(256);
    // one 64-lane wave per query when the per-round new-list is short (low degree x beam): keeps all 16-lane
    // groups busy and quadruples the queries resident per CU; variant 3 forces it, variant 4 forbids it
    // measured (profiles/r1_kernel_ab_wave_vs_wg.txt): the wave form wins only with >= 4096 queries in flight and
    // <= ~48 expected new nodes per query per round (beam x mean level-0 degree)
    const bool wave = ix->update_variant == 3 ||
                      (ix->update_variant == 0 && ix->ws.B >= 4096 && ix->ws.W * ix->avg_degree0 <= (double)ix->wave_maxnew);
    switch (ix->Dp / 64) {
#define CASE(n)                                                                                              \
    case n:                                                                                                  \
        if (a.by_rank == 2) hipLaunchKernelGGL((k_update<n, L2, F16, 2, 256>), grid, block, shmem, ix->stream, ix->ws, a); \
        else if (wave && a.by_rank) hipLaunchKernelGGL((k_update<n, L2, F16, 1, 64>), grid, dim3(64), shmem, ix->stream, ix->ws, a); \
        else if (wave) hipLaunchKernelGGL((k_update<n, L2, F16, 0, 64>), grid, dim3(64), shmem, ix->stream, ix->ws, a); \
        else if (a.by_rank) hipLaunchKernelGGL((k_update<n, L2, F16, 1, 256>), grid, block, shmem, ix->stream, ix->ws, a); \
        else hipLaunchKernelGGL((k_update<n, L2, F16, 0, 256>), grid, block, shmem, ix->stream, ix->ws, a);   \
        break
        CASE(1); CASE(2); CASE(3); CASE(4); CASE(5); CASE(6); CASE(8); CASE(12); CASE(16);
#undef CASE
        default: LM_FAIL(LM_EINVAL, "unsupported padded dimension (supported: 64..384, 512, 768, 1024)");
    }
    LM_HIP(hipGetLastError());
    return LM_OK;
}

static int launch_update(lm_index* ix, const UpdateArgs& a, bool f16) {
    size_t shmem = (size_t)(2 * ix->ws.ef + next_pow2(ix->ws.maxnew)) * sizeof(uint64_t);
    bool l2 = ix->metric == LM_METRIC_L2;
    if (l2) return f16 ? launch_update_nch<true, true>(ix, a, shmem) : launch_update_nch<true, false>(ix, a, shmem);
    return f16 ? launch_update_nch<false, true>(ix, a, shmem) : launch_update_nch<false, false>(ix, a, shmem);
}

// wave (64 threads) or workgroup (256 threads) per query for the persistent stored-embedding search: option "persistent_wave"
// 1 / 0 forces either, -1 (default) = wave when the expected new-list per hop (beam x mean level-0 degree) fits one pass of a
// wave's row groups comfortably (<= 24) and there are enough queries to fill the chip with waves (B >= 2048)
static bool persist_wave_form(const lm_index* ix) {
    if (ix->persistent_wave >= 0) return ix->persistent_wave != 0;
    return ix->ws.B >= 2048 && ix->ws.W * ix->avg_degree0 <= 24.0;
}

template <bool L2, bool F16>
static int launch_persist_nch(lm_index* ix, const PersistArgs& a, const GraphDev& g, size_t shmem) {
    dim3 grid(ix->ws.B);
    const bool wave = persist_wave_form(ix);
    switch (ix->Dp / 64) {
#define CASEP(n)                                                                                                         \
    case n:                                                                                                              \
        if (wave) {                                                                                                      \
            LM_HIP(hipFuncSetAttribute((const void*)k_search_table<n, L2, F16, 64>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem)); \
            hipLaunchKernelGGL((k_search_table<n, L2, F16, 64>), grid, dim3(64), shmem, ix->stream, g, ix->ws, a);        \
        } else {                                                                                                         \
            LM_HIP(hipFuncSetAttribute((const void*)k_search_table<n, L2, F16, 256>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem)); \
            hipLaunchKernelGGL((k_search_table<n, L2, F16, 256>), grid, dim3(256), shmem, ix->stream, g, ix->ws, a);      \
        }                                                                                                                \
        break
        CASEP(1); CASEP(2); CASEP(3); CASEP(4); CASEP(5); CASEP(6); CASEP(8); CASEP(12); CASEP(16);
#undef CASEP
        default: LM_FAIL(LM_EINVAL, "unsupported padded dimension (supported: 64..384, 512, 768, 1024)");
    }
    LM_HIP(hipGetLastError());
    return LM_OK;
}

// stored-embedding mode, no pruning: one persistent launch for the whole batch
static int search_pass_persistent(lm_index* ix, int32_t B, const float* d_q, int32_t k, const lm_search_params& prm,
                                  float* d_dist, int64_t* d_labels) {
    const int32_t ef = std::max(prm.efSearch, k);
    const int32_t W = std::max(prm.beam_size, 1);
    int rc = ensure_ws(ix, B, ef, W);
    if (rc) return rc;
    WsDev& ws = ix->ws;
    ws.efs = prm.efSearch;
    ws.batch = 0;
    ws.check_rel = prm.check_relative_distance;
    hipStream_t st = ix->stream;
    if ((int64_t)B > ix->pq_cap) {
        if (ix->d_pq_nadc) (void)hipFree(ix->d_pq_nadc);
        if (ix->d_pq_rounds) (void)hipFree(ix->d_pq_rounds);
        ix->d_pq_nadc = nullptr;
        ix->d_pq_rounds = nullptr;
        ix->pq_cap = 0;
        LM_HIP(hipMalloc((void**)&ix->d_pq_nadc, (size_t)B * 8));
        LM_HIP(hipMalloc((void**)&ix->d_pq_rounds, (size_t)B * 4));
        ix->pq_cap = B;
    }
    GraphDev g{ix->N, ix->entry_point, ix->max_level, ix->d_node_offsets, ix->d_level_ptr, ix->d_neighbors, ix->d_l0};
    PersistArgs a{};
    a.Q = d_q; a.E = ix->d_table; a.check_rel = prm.check_relative_distance; a.max_level = ix->max_level;
    a.Pmax = next_pow2(ws.maxnew); a.k = k; a.metric = ix->metric; a.labels = d_labels; a.dist = d_dist;
    a.rounds_q = ix->d_pq_rounds;
    const size_t shmem = ((size_t)2 * ef + a.Pmax) * 8 + (size_t)ws.maxnew * 4;
    if (shmem > 150 * 1024) return 1;  // caller falls back to the lock-step path
    LM_HIP(hipMemsetAsync(ws.visited, 0, (size_t)B * ws.nw * 4, st));
    LM_HIP(hipMemsetAsync(ws.counters, 0, C_NCOUNTERS * sizeof(unsigned long long), st));
    {
        EvScope es(ix, &ix->ev_update);
        const bool f16 = ix->table_dtype == LM_DTYPE_F16, l2 = ix->metric == LM_METRIC_L2;
        rc = l2 ? (f16 ? launch_persist_nch<true, true>(ix, a, g, shmem) : launch_persist_nch<true, false>(ix, a, g, shmem))
                : (f16 ? launch_persist_nch<false, true>(ix, a, g, shmem) : launch_persist_nch<false, false>(ix, a, g, shmem));
    }
    if (rc) return rc;
    ix->stats.update_launches++;
    hipLaunchKernelGGL(k_stats, dim3(1), dim3(256), 0, st, ws);
    hipLaunchKernelGGL(k_rounds_max, dim3(1), dim3(256), 0, st, ws, ix->d_pq_rounds);
    unsigned long long* hc = ix->h_counters;
    LM_HIP(hipMemcpyAsync(hc, ws.counters, C_NCOUNTERS * sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
    LM_HIP(hipStreamSynchronize(st));
    ix->stats.nrounds = std::max<int64_t>(ix->stats.nrounds, (int64_t)hc[C_ROUNDS]);
    ix->stats.ndis += (int64_t)hc[C_NDIS];
    ix->stats.nexpand += (int64_t)hc[C_NEXPAND];
    return LM_OK;
}

// one pass over <= max_batch queries; d_q: B x Dp (padded)
static int search_pass(lm_index* ix, int32_t B, const float* d_q, int32_t k, const lm_search_params& prm,
                       float* d_dist, int64_t* d_labels) {
    const int32_t ef = std::max(prm.efSearch, k);  // faiss: max(efSearch, k)
    const int32_t W = std::max(prm.beam_size, 1);
    const bool prune = prm.pq_pruning_ratio > 0.0f;
    if (prune) {
        if (!ix->d_pq_codes) LM_FAIL(LM_ESTATE, "pq_pruning_ratio > 0 needs a product quantiser (lm_pq_attach)");
        if (prm.pq_pruning_ratio >= 1.0f) LM_FAIL(LM_EINVAL, "pq_pruning_ratio must be < 1");
    }
    const bool recompute = prm.recompute != 0;
    // speculative prefetch (k_speculate): small recompute batches with the per-call memo; not with the two-level search (its new-lists are
    // finished by k_prune)
    const int spec = (recompute && prm.recompute_memo != 0 && !prune && B <= ix->speculate_max_batch) ? ix->speculate : 0;
    int rc = ensure_ws(ix, B, ef, W, prune, spec, prm.batch_size);
    if (rc) return rc;
    WsDev& ws = ix->ws;
    ws.efs = prm.efSearch;
    ws.batch = prm.batch_size;
    ws.check_rel = prm.check_relative_distance;
    hipStream_t st = ix->stream;
    PruneArgs pa{};
    size_t prune_shmem = 0;
    if (prune) {
        const int64_t need = (int64_t)B * ix->pq_m * 256;
        if (need > ix->lut_cap) {
            if (ix->d_lut) (void)hipFree(ix->d_lut);
            ix->d_lut = nullptr;
            ix->lut_cap = 0;
            LM_HIP(hipMalloc((void**)&ix->d_lut, (size_t)need * 4));
            ix->lut_cap = need;
        }
        pa.Q = d_q; pa.lut = ix->d_lut; pa.codebooks = ix->d_pq_codebooks; pa.codes = ix->d_pq_codes;
        pa.Dp = ix->Dp; pa.metric = ix->metric; pa.m = ix->pq_m; pa.chunk_off = ix->d_pq_chunk_off;
        pa.keep = 1.0f - prm.pq_pruning_ratio;
        pa.strategy = prm.local_prune ? 1 : (prm.send_neigh_times_ratio > 1e-6f ? 2 : 0);  // hnsw_backend.py:222-231
        pa.Pmax = next_pow2(ws.maxnew);
        prune_shmem = ((size_t)pa.Pmax + 2 * AQ_CAP) * 8;
        hipLaunchKernelGGL(k_pq_lut_all, dim3(B), dim3(256), 0, st, pa);
    }
    const bool hub = recompute && ix->hub_n > 0;
    // Per-call memo (the default): a node's embedding is recomputed at most once per pass and stays in HBM until the pass returns.
    // A one-query pass never meets a node twice (visited set), so it skips the memo's bookkeeping -- unless it prefetches (spec > 0).
    const bool memo_call = recompute && prm.recompute_memo != 0 && (B > 1 || spec > 0);
    const bool memo = memo_call || hub;                                       // rows are addressed through memo_slot
    int64_t memo_used = 0;
    if (memo) {
        if (!ix->d_memo_slot) LM_HIP(hipMalloc((void**)&ix->d_memo_slot, (size_t)ix->N * 4));
        const int64_t first = ix->memo_initial_rows > 0 ? ix->memo_initial_rows : std::max<int64_t>(65536, (int64_t)B * 1024);
        if ((rc = ensure_memo_rows(ix, std::min<int64_t>(ix->N, ix->hub_n + (memo_call ? first : 0)), hub ? ix->hub_n : 0)) != LM_OK) return rc;
        ws.memo_slot = ix->d_memo_slot;
        ws.memo = ix->d_memo;
        if (hub) {
            LM_HIP(hipMemcpyAsync(ix->d_memo_slot, ix->d_hub_slot_init, (size_t)ix->N * 4, hipMemcpyDeviceToDevice, st));
            memo_used = ix->hub_n;
        } else {
            LM_HIP(hipMemsetAsync(ix->d_memo_slot, 0xFF, (size_t)ix->N * 4, st));
        }
    }
    GraphDev g{ix->N, ix->entry_point, ix->max_level, ix->d_node_offsets, ix->d_level_ptr, ix->d_neighbors, ix->d_l0};
    // A ONE-query pass without a memo needs no cross-query dedup: a new-list holds every node once (visited test-and-set; an upper-level
    // list is one neighbour list), so it IS the provider's id list, in discovery order instead of id order -- three launches per round
    // fewer (the live-flag memset, k_uniq_count, k_uniq_emit).  Same labels, distances and counts (a chunk's embedding does not depend on
    // its place in the forward).  Option "single_query_direct", off by default: not yet timed on hardware.
    const bool single = recompute && B == 1 && !memo && !prune && ix->single_query_direct != 0;

    LM_HIP(hipMemsetAsync(ws.visited, 0, (size_t)B * ws.nw * 4, st));
    LM_HIP(hipMemsetAsync(ws.counters, 0, C_NCOUNTERS * sizeof(unsigned long long), st));
    hipLaunchKernelGGL(k_init, dim3((B + 255) / 256), dim3(256), 0, st, ws, ix->max_level);

    UpdateArgs ua{};
    unsigned long long* span_acc = nullptr;
    if (ix->profiling) {
        if (ix->tstamp_cap < B) {
            if (ix->d_tstamp) (void)hipFree(ix->d_tstamp);
            ix->d_tstamp = nullptr;
            ix->tstamp_cap = 0;
            LM_HIP(hipMalloc((void**)&ix->d_tstamp, ((size_t)2 * B + 2) * 8));
            ix->tstamp_cap = B;
        }
        ua.tstamp = ix->d_tstamp;
        span_acc = ix->d_tstamp + (size_t)2 * ix->tstamp_cap;
        LM_HIP(hipMemsetAsync(ix->d_tstamp, 0, ((size_t)2 * ix->tstamp_cap + 2) * 8, st));
    }
    ua.Q = d_q;
    ua.check_rel = prm.check_relative_distance;
    ua.max_level = ix->max_level;
    {
        // dynamic LDS of the update kernel (default launch limit 64 KiB): pool | merged pool | new keys
        const size_t need = ((size_t)2 * ef + next_pow2(ws.maxnew)) * 8;
        if (need > 64 * 1024)
            LM_FAIL(LM_EINVAL, "efSearch / beam_size / batch_size too large for the LDS-resident pool (2*max(efSearch,k) + max(beam*max_degree, batch_size + max_degree) keys must fit 64 KiB)");
    }
    const int ntiles = (int)((ws.nw + UNIQ_TILE - 1) / UNIQ_TILE);
    const int sync_every = recompute ? 1 : 4;
    int64_t rounds = 0;
    unsigned long long* hc = ix->h_counters;

    const int32_t* round_ids = single ? ws.newid : ws.uniq;  // what the provider is asked for
    ua.identity = single ? 1 : 0;
    for (;;) {
        if (!single) LM_HIP(hipMemsetAsync(ws.counters + C_LIVE, 0, sizeof(unsigned long long), st));
        {
            EvScope es(ix, &ix->ev_expand);
            hipLaunchKernelGGL(k_expand, dim3(B), dim3(64), (size_t)ws.maxnew * sizeof(int32_t), st, g, ws, single ? 0 : recompute ? (memo ? 2 : 1) : 0,
                               (int)(rounds + 1), prune ? 1 : 0, single ? 1 : 0);
            if (prune) {
                pa.use_rbm = recompute ? (memo ? 2 : 1) : 0;
                hipLaunchKernelGGL(k_prune, dim3(B), dim3(256), prune_shmem, st, ws, pa);
            }
            if (spec > 0) hipLaunchKernelGGL(k_speculate, dim3(B), dim3(64), 0, st, g, ws, spec, (int)prm.check_relative_distance);
            if (recompute && !single) {
                hipLaunchKernelGGL(k_uniq_count, dim3(ntiles), dim3(256), 0, st, ws);
                hipLaunchKernelGGL(k_uniq_emit, dim3(ntiles), dim3(256), 0, st, ws, ntiles);
            }
        }
        rounds++;
        // built-in provider: lengths of the round's chunks (count still on the device) before the copy below, so that this round
        // needs no second synchronisation inside the provider
        const bool native = recompute && ix->native_rc && ix->provider == lm_recompute_provider && ix->provider_user == (void*)ix->native_rc;
        if (native && (rc = rc_prepare(ix->native_rc, round_ids, ws.counters + C_NUNIQ, single ? (int64_t)ws.maxnew : ix->ws_ucap, ws.counters + C_RC_TOKENS,
                                       ws.counters + C_RC_MAXLEN, st)) != LM_OK)
            return rc;
        bool do_sync = (rounds % sync_every) == 0;
        if (do_sync) {
            LM_HIP(hipMemcpyAsync(hc, ws.counters, C_NCOUNTERS * sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
            LM_HIP(hipStreamSynchronize(st));
            if (hc[C_LIVE] == 0) break;
        }
        if (recompute) {
            int32_t nu = (int32_t)hc[C_NUNIQ];
            void* d_e = nullptr;
            ix->stats.nunique += nu;
            if (native) rc_prepared(ix->native_rc, round_ids, nu, (int64_t)hc[C_RC_TOKENS], (int32_t)hc[C_RC_MAXLEN]);
            if (nu > 0) {
                EvScope es(ix, &ix->ev_provider);
                int prc = ix->provider(ix->provider_user, round_ids, nu, &d_e, (void*)st);
                if (prc != 0 || !d_e) LM_FAIL(LM_EPROVIDER, "embedding provider failed (rc=" + std::to_string(prc) + ")");
            }
            if (memo) {
                if (nu > 0) {
                    if ((rc = ensure_memo_rows(ix, memo_used + nu, memo_used)) != LM_OK) return rc;
                    hipLaunchKernelGGL(k_memo_append, dim3((unsigned)std::min<int64_t>(2048, ((int64_t)nu * (ix->Dp / 4) + 255) / 256)),
                                       dim3(256), 0, st, ws, (const float*)d_e, nu, memo_used, ix->Dp);
                    if (memo_call) memo_used += nu;
                }
                ua.E = ix->d_memo;
                ua.by_rank = 2;
            } else {
                ua.E = d_e;
                ua.by_rank = 1;
            }
            EvScope es(ix, &ix->ev_update);
            rc = launch_update(ix, ua, false);
        } else {
            ua.E = ix->d_table;
            ua.by_rank = 0;
            EvScope es(ix, &ix->ev_update);
            rc = launch_update(ix, ua, ix->table_dtype == LM_DTYPE_F16);
        }
        if (rc) return rc;
        if (hub && !memo_call && recompute && hc[C_NUNIQ] > 0)
            hipLaunchKernelGGL(k_memo_release, dim3((unsigned)std::min<int64_t>(1024, ((int64_t)hc[C_NUNIQ] + 255) / 256)), dim3(256), 0, st, ws,
                               (int32_t)hc[C_NUNIQ]);
        if (span_acc) hipLaunchKernelGGL(k_span, dim3(1), dim3(256), 0, st, ix->d_tstamp, B, span_acc);
        ix->stats.update_launches++;
    }
    hipLaunchKernelGGL(k_finalize, dim3((B * k + 255) / 256), dim3(256), 0, st, ws, k, ix->metric, d_labels, d_dist);
    hipLaunchKernelGGL(k_stats, dim3(1), dim3(256), 0, st, ws);
    LM_HIP(hipGetLastError());
    LM_HIP(hipMemcpyAsync(hc, ws.counters, C_NCOUNTERS * sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
    LM_HIP(hipStreamSynchronize(st));
    ix->stats.nrounds += (int64_t)hc[C_ROUNDS];
    ix->stats.ndis += (int64_t)hc[C_NDIS];
    ix->stats.nexpand += (int64_t)hc[C_NEXPAND];
    ix->stats.nadc += (int64_t)hc[C_NADC];
    if (span_acc) {
        unsigned long long acc[2] = {0, 0};
        LM_HIP(hipMemcpy(acc, span_acc, sizeof(acc), hipMemcpyDeviceToHost));
        int khz = 100000;
        (void)hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, ix->device);
        ix->span_ms += (double)acc[0] / (double)khz;
        ix->span_launches += (int64_t)acc[1];
    }
    return LM_OK;
}

// Query / result staging of the host-pointer entry points (lm_index_search, lm_pq_batch_search): the buffers live with the index and only
// ever grow -- LEANN's real call is one query at a time (leann/api.py:644-796), three hipMalloc / hipFree pairs per call were a
// measurable part of its latency.
static int ensure_stage(lm_index* ix, size_t need_x, size_t need_d, size_t need_l) {
    if (need_x <= ix->stage_x_bytes && need_d <= ix->stage_d_bytes && need_l <= ix->stage_l_bytes) return LM_OK;
    if (ix->d_stage_x) (void)hipFree(ix->d_stage_x);
    if (ix->d_stage_d) (void)hipFree(ix->d_stage_d);
    if (ix->d_stage_l) (void)hipFree(ix->d_stage_l);
    ix->d_stage_x = nullptr; ix->d_stage_d = nullptr; ix->d_stage_l = nullptr;
    ix->stage_x_bytes = ix->stage_d_bytes = ix->stage_l_bytes = 0;
    const size_t gx = std::max(need_x, (size_t)4096), gd = std::max(need_d, (size_t)4096), gl = std::max(need_l, (size_t)4096);
    if (hipMalloc((void**)&ix->d_stage_x, gx) != hipSuccess || hipMalloc((void**)&ix->d_stage_d, gd) != hipSuccess ||
        hipMalloc((void**)&ix->d_stage_l, gl) != hipSuccess)
        LM_FAIL(LM_EHIP, "out of device memory for the query / result staging buffers");
    ix->stage_x_bytes = gx; ix->stage_d_bytes = gd; ix->stage_l_bytes = gl;
    return LM_OK;
}

static int do_search_device(lm_index* ix, int64_t n, const float* d_x, int32_t k, float* d_dist, int64_t* d_labels,
                            const lm_search_params* params) {
    if (!ix || !params || n < 0 || k <= 0) LM_FAIL(LM_EINVAL, "bad search arguments");
    lm_search_params prm = *params;
    if (prm.efSearch <= 0) LM_FAIL(LM_EINVAL, "efSearch must be positive");
    if (prm.batch_size < 0) LM_FAIL(LM_EINVAL, "batch_size must not be negative (0 = no dynamic batching)");
    LM_HIP(hipSetDevice(ix->device));
    ix->stats = lm_search_stats{};
    ix->span_ms = 0;
    ix->span_launches = 0;
    (void)drain_events(ix, ix->ev_update);  // pairs left behind by a call that failed midway
    (void)drain_events(ix, ix->ev_expand);
    (void)drain_events(ix, ix->ev_provider);
    if (n == 0) return LM_OK;
    hipStream_t st = ix->stream;
    if (ix->N == 0 || ix->entry_point < 0) {
        hipLaunchKernelGGL(k_fill_empty, dim3((unsigned)((n * k + 255) / 256)), dim3(256), 0, st, n * (int64_t)k,
                           ix->metric, d_labels, d_dist);
        LM_HIP(hipStreamSynchronize(st));
        return LM_OK;
    }
    if (prm.recompute) {
        if (!ix->provider) LM_FAIL(LM_ESTATE, "recompute requested but no embedding provider is attached");
    } else if (!ix->d_table) {
        LM_FAIL(LM_ESTATE, "index stores no embeddings (pruned): recompute is required");
    }
    // pad queries to Dp if needed
    const float* d_q = d_x;
    if (ix->D != ix->Dp) {
        if (n > ix->qpad_cap) {
            if (ix->d_qpad) (void)hipFree(ix->d_qpad);
            ix->d_qpad = nullptr;
            ix->qpad_cap = 0;
            LM_HIP(hipMalloc((void**)&ix->d_qpad, (size_t)n * ix->Dp * sizeof(float)));
            ix->qpad_cap = n;
        }
        int64_t tot = n * ix->Dp;
        hipLaunchKernelGGL(k_pad_rows, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, d_x, n, ix->D, ix->Dp, ix->d_qpad);
        d_q = ix->d_qpad;
    }
    int64_t maxb = prm.max_batch > 0 ? prm.max_batch : 4096;
    // bound the visited bitmaps to 8 GiB
    int64_t nwbytes = ((ix->N + 31) / 32) * 4;
    maxb = std::max<int64_t>(1, std::min<int64_t>(maxb, (8ll << 30) / std::max<int64_t>(nwbytes, 1)));
    for (int64_t off = 0; off < n; off += maxb) {
        int32_t B = (int32_t)std::min<int64_t>(maxb, n - off);
        int rc = 1;
        // (dynamic batching exists to fill the recompute forward: a stored-embedding search that asks for it runs the lock-step kernels, which implement it)
        if (!prm.recompute && prm.pq_pruning_ratio <= 0.0f && prm.batch_size == 0 && ix->persistent_table && ix->update_variant == 0 && std::max(prm.beam_size, 1) <= 64)
            rc = search_pass_persistent(ix, B, d_q + (size_t)off * ix->Dp, k, prm, d_dist + (size_t)off * k, d_labels + (size_t)off * k);
        if (rc == 1)  // not applicable (or LDS budget exceeded): lock-step rounds
            rc = search_pass(ix, B, d_q + (size_t)off * ix->Dp, k, prm, d_dist + (size_t)off * k, d_labels + (size_t)off * k);
        if (rc) return rc;
    }
    LM_HIP(hipStreamSynchronize(st));
    if (ix->profiling) {
        ix->stats.update_ms = drain_events(ix, ix->ev_update);
        ix->stats.expand_ms = drain_events(ix, ix->ev_expand);
        ix->stats.provider_ms = drain_events(ix, ix->ev_provider);
        ix->stats.update_span_ms = ix->span_ms;
        ix->stats.update_span_launches = ix->span_launches;
        if (ix->span_launches == 0) {  // persistent launches: the event pair around one long kernel is the duration
            ix->stats.update_span_ms = ix->stats.update_ms;
            ix->stats.update_span_launches = ix->stats.update_launches;
        }
    }
    return LM_OK;
}

static void compute_degrees(lm_index* ix, const uint64_t* node_offsets, const uint64_t* level_ptr, std::vector<L0Range>& l0) {
    int32_t m0 = 0, mu = 0;
    l0.resize((size_t)ix->N);
    for (int64_t i = 0; i < ix->N; ++i) {
        uint64_t p0 = node_offsets[i], p1 = node_offsets[i + 1];
        l0[i] = L0Range{p1 > p0 + 1 ? level_ptr[p0] : 0, p1 > p0 + 1 ? (uint32_t)(level_ptr[p0 + 1] - level_ptr[p0]) : 0u, 0u};
        for (uint64_t p = p0; p + 1 < p1; ++p) {
            int32_t deg = (int32_t)(level_ptr[p + 1] - level_ptr[p]);
            if (p == p0) m0 = std::max(m0, deg);
            else mu = std::max(mu, deg);
        }
    }
    ix->maxdeg0 = m0;
    ix->maxdeg_up = mu;
    double e0 = 0;
    for (int64_t i = 0; i < ix->N; ++i) e0 += l0[i].count;
    ix->avg_degree0 = ix->N ? e0 / (double)ix->N : 0.0;
}

extern "C" {

const char* lm_last_error(void) { return g_err.c_str(); }
const char* lm_version(void) { return "leann-mi355x 0.2 (gfx950)"; }
int lm_abi_revision(void) { return LM_ABI_REVISION; }

int lm_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

void lm_search_params_default(lm_search_params* p) {
    if (!p) return;
    std::memset(p, 0, sizeof(*p));
    p->efSearch = 64;
    p->beam_size = 1;
    p->check_relative_distance = 1;
    p->recompute = 1;
    p->max_batch = 0;
    p->recompute_memo = 1;
}

int lm_index_create_from_csr(int64_t ntotal, int32_t d, int32_t metric, const uint64_t* node_offsets,
                             const uint64_t* level_ptr, int64_t n_level_ptr, const int32_t* neighbors,
                             int64_t n_neighbors, const int32_t* levels, int32_t entry_point, int32_t max_level,
                             int device, lm_index** out) {
    if (!out) LM_FAIL(LM_EINVAL, "out is NULL");
    *out = nullptr;
    if (ntotal < 0 || d <= 0 || (metric != LM_METRIC_L2 && metric != LM_METRIC_INNER_PRODUCT))
        LM_FAIL(LM_EINVAL, "bad ntotal / d / metric");
    if (ntotal > 0 && (!node_offsets || !level_ptr || !levels)) LM_FAIL(LM_EINVAL, "NULL CSR array");
    if (ntotal > 0x7fffffff) LM_FAIL(LM_EINVAL, "ntotal exceeds int32 node ids");
    if (ntotal > 0) {
        if ((int64_t)node_offsets[ntotal] != n_level_ptr) LM_FAIL(LM_EFORMAT, "node_offsets[ntotal] != len(level_ptr)");
        if (entry_point < 0 || entry_point >= ntotal) LM_FAIL(LM_EFORMAT, "entry_point out of range");
        if (levels[entry_point] != max_level + 1) LM_FAIL(LM_EFORMAT, "levels[entry_point] != max_level + 1");
        for (int64_t i = 0; i < ntotal; ++i)
            if ((int64_t)(node_offsets[i + 1] - node_offsets[i]) != (int64_t)levels[i] + 1)
                LM_FAIL(LM_EFORMAT, "node_offsets[i+1]-node_offsets[i] != levels[i]+1");
        for (int64_t p = 0; p + 1 < n_level_ptr; ++p)
            if (level_ptr[p + 1] < level_ptr[p]) LM_FAIL(LM_EFORMAT, "level_ptr not monotone");
        if (n_level_ptr > 0 && (int64_t)level_ptr[n_level_ptr - 1] > n_neighbors) LM_FAIL(LM_EFORMAT, "level_ptr past neighbors");
        for (int64_t e = 0; e < n_neighbors; ++e)
            if (neighbors[e] < 0 || neighbors[e] >= ntotal) LM_FAIL(LM_EFORMAT, "neighbor id out of range");
    }
    int ndev = lm_device_count();
    if (ndev <= 0) LM_FAIL(LM_EHIP, "no HIP device visible: libleann_mi355x requires an MI355X (gfx950) GPU");
    if (device < 0 || device >= ndev) LM_FAIL(LM_EINVAL, "device index out of range");
    LM_HIP(hipSetDevice(device));
    lm_index* ix = new lm_index();
    ix->device = device;
    ix->N = ntotal; ix->D = d; ix->Dp = (d + 63) / 64 * 64; ix->metric = metric;
    ix->entry_point = ntotal > 0 ? entry_point : -1;
    ix->max_level = max_level;
    ix->n_neighbors = n_neighbors; ix->n_level_ptr = n_level_ptr;
    if (ntotal > 0) {
        std::vector<L0Range> l0;
        compute_degrees(ix, node_offsets, level_ptr, l0);
        hipError_t e;
        if ((e = hipMalloc((void**)&ix->d_l0, (size_t)ntotal * sizeof(L0Range))) != hipSuccess ||
            (e = hipMemcpy(ix->d_l0, l0.data(), (size_t)ntotal * sizeof(L0Range), hipMemcpyHostToDevice)) != hipSuccess ||
            (e = hipMalloc((void**)&ix->d_node_offsets, (size_t)(ntotal + 1) * 8)) != hipSuccess ||
            (e = hipMalloc((void**)&ix->d_level_ptr, (size_t)std::max<int64_t>(n_level_ptr, 1) * 8)) != hipSuccess ||
            (e = hipMalloc((void**)&ix->d_neighbors, (size_t)std::max<int64_t>(n_neighbors, 1) * 4)) != hipSuccess ||
            (e = hipMemcpy(ix->d_node_offsets, node_offsets, (size_t)(ntotal + 1) * 8, hipMemcpyHostToDevice)) != hipSuccess ||
            (e = hipMemcpy(ix->d_level_ptr, level_ptr, (size_t)n_level_ptr * 8, hipMemcpyHostToDevice)) != hipSuccess ||
            (e = hipMemcpy(ix->d_neighbors, neighbors, (size_t)n_neighbors * 4, hipMemcpyHostToDevice)) != hipSuccess) {
            set_error(std::string("graph upload failed: ") + hipGetErrorString(e));
            lm_index_free(ix);
            return LM_EHIP;
        }
    }
    if (hipHostMalloc((void**)&ix->h_counters, C_NCOUNTERS * sizeof(unsigned long long)) != hipSuccess) {
        set_error("hipHostMalloc failed");
        lm_index_free(ix);
        return LM_EHIP;
    }
    std::memset(ix->h_counters, 0, C_NCOUNTERS * sizeof(unsigned long long));
    *out = ix;
    return LM_OK;
}

int lm_index_read(const char* path, int device, lm_index** out) {
    if (!out || !path) LM_FAIL(LM_EINVAL, "NULL argument");
    *out = nullptr;
    HostCsr h;
    int rc = read_csr_file(path, h);
    if (rc) return rc;
    rc = lm_index_create_from_csr(h.ntotal, h.d, h.metric, h.node_offsets.data(), h.level_ptr.data(),
                                  (int64_t)h.level_ptr.size(), h.neighbors.data(), (int64_t)h.neighbors.size(),
                                  h.levels.data(), h.entry_point, h.max_level, device, out);
    if (rc) return rc;
    if (!h.storage.empty()) {
        rc = lm_index_attach_table(*out, h.storage.data(), LM_DTYPE_F32, h.ntotal, h.d, 0);
        if (rc) {
            lm_index_free(*out);
            *out = nullptr;
        }
    }
    return rc;
}

void lm_index_free(lm_index* ix) {
    if (!ix) return;
    (void)hipSetDevice(ix->device);
    free_ws(ix);
    if (ix->d_node_offsets) (void)hipFree(ix->d_node_offsets);
    if (ix->d_level_ptr) (void)hipFree(ix->d_level_ptr);
    if (ix->d_neighbors) (void)hipFree(ix->d_neighbors);
    if (ix->d_l0) (void)hipFree(ix->d_l0);
    if (ix->d_memo_slot) (void)hipFree(ix->d_memo_slot);
    if (ix->d_memo) (void)hipFree(ix->d_memo);
    if (ix->d_hub_slot_init) (void)hipFree(ix->d_hub_slot_init);
    if (ix->d_pq_codebooks) (void)hipFree(ix->d_pq_codebooks);
    if (ix->d_pq_codes) (void)hipFree(ix->d_pq_codes);
    if (ix->d_pq_chunk_off) (void)hipFree(ix->d_pq_chunk_off);
    if (ix->d_stage_x) (void)hipFree(ix->d_stage_x);
    if (ix->d_stage_d) (void)hipFree(ix->d_stage_d);
    if (ix->d_stage_l) (void)hipFree(ix->d_stage_l);
    if (ix->d_pq_nadc) (void)hipFree(ix->d_pq_nadc);
    if (ix->d_pq_rounds) (void)hipFree(ix->d_pq_rounds);
    if (ix->d_lut) (void)hipFree(ix->d_lut);
    if (ix->d_tstamp) (void)hipFree(ix->d_tstamp);
    if (ix->d_table && ix->table_owned) (void)hipFree(ix->d_table);
    if (ix->d_qpad) (void)hipFree(ix->d_qpad);
    if (ix->h_counters) (void)hipHostFree(ix->h_counters);
    (void)drain_events(ix, ix->ev_update);
    (void)drain_events(ix, ix->ev_expand);
    (void)drain_events(ix, ix->ev_provider);
    for (hipEvent_t e : ix->ev_pool) (void)hipEventDestroy(e);
    delete ix;
}

int lm_index_info(const lm_index* ix, lm_index_info_t* o) {
    if (!ix || !o) LM_FAIL(LM_EINVAL, "NULL argument");
    o->ntotal = ix->N; o->d = ix->D; o->d_padded = ix->Dp; o->metric = ix->metric;
    o->entry_point = ix->entry_point; o->max_level = ix->max_level;
    o->max_degree0 = ix->maxdeg0; o->max_degree_up = ix->maxdeg_up; o->n_neighbors = ix->n_neighbors;
    o->has_table = ix->d_table != nullptr; o->has_provider = ix->provider != nullptr; o->device = ix->device;
    return LM_OK;
}

int lm_index_attach_table(lm_index* ix, const void* table, int32_t dtype, int64_t ntotal, int32_t d, int32_t location) {
    if (!ix || !table) LM_FAIL(LM_EINVAL, "NULL argument");
    if (ntotal != ix->N || d != ix->D) LM_FAIL(LM_EINVAL, "table shape does not match the index");
    if (dtype != LM_DTYPE_F32 && dtype != LM_DTYPE_F16) LM_FAIL(LM_EINVAL, "dtype must be f32 or f16");
    LM_HIP(hipSetDevice(ix->device));
    if (ix->d_table && ix->table_owned) (void)hipFree(ix->d_table);
    ix->d_table = nullptr;
    ix->table_owned = false;
    ix->table_dtype = dtype;
    if (location == 1) {
        ix->d_table = const_cast<void*>(table);  // borrowed, stride Dp
        return LM_OK;
    }
    size_t es = dtype == LM_DTYPE_F16 ? 2 : 4;
    void* dst = nullptr;
    LM_HIP(hipMalloc(&dst, std::max<size_t>((size_t)ntotal * ix->Dp * es, 16)));
    ix->d_table = dst;
    ix->table_owned = true;
    if (ix->D == ix->Dp) {
        LM_HIP(hipMemcpy(dst, table, (size_t)ntotal * d * es, hipMemcpyHostToDevice));
    } else {
        void* tmp = nullptr;
        LM_HIP(hipMalloc(&tmp, std::max<size_t>((size_t)ntotal * d * es, 16)));
        if (hipMemcpy(tmp, table, (size_t)ntotal * d * es, hipMemcpyHostToDevice) != hipSuccess) {
            (void)hipFree(tmp);
            LM_FAIL(LM_EHIP, "table upload failed");
        }
        int64_t tot = ntotal * ix->Dp;
        if (dtype == LM_DTYPE_F32)
            hipLaunchKernelGGL(k_pad_rows, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, 0, (const float*)tmp, ntotal, d, ix->Dp, (float*)dst);
        else
            hipLaunchKernelGGL(k_pad_rows_f16, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, 0, (const __half*)tmp, ntotal, d, ix->Dp, (__half*)dst);
        const hipError_t pe = hipDeviceSynchronize();
        (void)hipFree(tmp);
        LM_HIP(pe);
    }
    return LM_OK;
}

int lm_index_set_hub_cache(lm_index* ix, const int32_t* ids, int32_t n, const float* d_embeddings) {
    if (!ix) LM_FAIL(LM_EINVAL, "NULL index");
    LM_HIP(hipSetDevice(ix->device));
    if (n == 0) {
        ix->hub_n = 0;
        return LM_OK;
    }
    if (n < 0 || !ids || !d_embeddings || n > ix->N) LM_FAIL(LM_EINVAL, "bad hub cache arguments");
    std::vector<int32_t> slot((size_t)ix->N, -1);
    for (int32_t i = 0; i < n; ++i) {
        if (ids[i] < 0 || ids[i] >= ix->N || slot[ids[i]] >= 0) LM_FAIL(LM_EINVAL, "hub ids must be unique and in range");
        slot[ids[i]] = i;
    }
    if (int mrc = ensure_memo_rows(ix, n, 0)) return mrc;
    if (!ix->d_hub_slot_init) LM_HIP(hipMalloc((void**)&ix->d_hub_slot_init, (size_t)ix->N * 4));
    LM_HIP(hipMemcpy(ix->d_hub_slot_init, slot.data(), (size_t)ix->N * 4, hipMemcpyHostToDevice));
    LM_HIP(hipMemcpy(ix->d_memo, d_embeddings, (size_t)n * ix->Dp * 4, hipMemcpyDeviceToDevice));
    ix->hub_n = n;
    return LM_OK;
}

int lm_index_set_provider(lm_index* ix, lm_provider_fn fn, void* user) {
    if (!ix) LM_FAIL(LM_EINVAL, "NULL index");
    ix->provider = fn;
    ix->provider_user = user;
    return LM_OK;
}

int lm_index_set_recompute(lm_index* ix, lm_recompute* rc) {
    if (!ix) LM_FAIL(LM_EINVAL, "NULL index");
    if (rc && ix->Dp != lm::rc_width(rc))
        LM_FAIL(LM_EINVAL, "lm_index_set_recompute: the provider's embeddings are " + std::to_string(lm::rc_width(rc)) + " wide; this index has padded d = " +
                               std::to_string(ix->Dp));
    ix->native_rc = rc;
    ix->provider = rc ? lm_recompute_provider : nullptr;
    ix->provider_user = rc;
    return LM_OK;
}

int lm_index_set_stream(lm_index* ix, void* s) {
    if (!ix) LM_FAIL(LM_EINVAL, "NULL index");
    ix->stream = (hipStream_t)s;
    return LM_OK;
}

int lm_index_set_profiling(lm_index* ix, int32_t enable) {
    if (!ix) LM_FAIL(LM_EINVAL, "NULL index");
    ix->profiling = enable != 0;
    return LM_OK;
}

int lm_index_set_option(lm_index* ix, const char* name, int64_t value) {
    if (!ix || !name) LM_FAIL(LM_EINVAL, "NULL argument");
    if (!std::strcmp(name, "update_variant")) {
        if (value != 0 && value != 3 && value != 4)
            LM_FAIL(LM_EINVAL, "update_variant must be 0 (auto), 3 (wave per query) or 4 (workgroup per query); 1 and 2 were removed");
        ix->update_variant = (int)value;
        return LM_OK;
    }
    if (!std::strcmp(name, "persistent_table")) {
        ix->persistent_table = value != 0;
        return LM_OK;
    }
    if (!std::strcmp(name, "pq_threads")) {
        if (value != 256 && value != 512 && value != 1024) LM_FAIL(LM_EINVAL, "pq_threads must be 256, 512 or 1024");
        ix->pq_threads = (int)value;
        return LM_OK;
    }
    if (!std::strcmp(name, "pq_rerank_expanded")) {
        ix->pq_rerank_expanded = value != 0;
        ix->pq_overflow = 0;
        return LM_OK;
    }
    if (!std::strcmp(name, "persistent_wave")) {
        if (value < -1 || value > 1) LM_FAIL(LM_EINVAL, "persistent_wave must be -1 (auto), 0 or 1");
        ix->persistent_wave = (int)value;
        return LM_OK;
    }
    if (!std::strcmp(name, "wave_maxnew")) {
        ix->wave_maxnew = (int)value;
        return LM_OK;
    }
    if (!std::strcmp(name, "single_query_direct")) {
        ix->single_query_direct = value != 0;
        return LM_OK;
    }
    if (!std::strcmp(name, "speculate")) {  // k_speculate: neighbours of the S best unexpanded candidates are embedded ahead of time (small batches)
        if (value < 0 || value > 64) LM_FAIL(LM_EINVAL, "speculate must be in [0, 64]");
        ix->speculate = (int)value;
        return LM_OK;
    }
    if (!std::strcmp(name, "speculate_max_batch")) {
        if (value < 1) LM_FAIL(LM_EINVAL, "speculate_max_batch must be >= 1");
        ix->speculate_max_batch = (int)std::min<int64_t>(value, 1 << 20);
        return LM_OK;
    }
    if (!std::strcmp(name, "memo_initial_rows")) {  // first allocation of the per-call recompute memo (it doubles on demand); 0 = default
        if (value < 0) LM_FAIL(LM_EINVAL, "memo_initial_rows must not be negative");
        ix->memo_initial_rows = value;
        return LM_OK;
    }
    LM_FAIL(LM_EINVAL, std::string("unknown option: ") + name);
}

int lm_index_get_option(const lm_index* ix, const char* name, int64_t* value) {
    if (!ix || !name || !value) LM_FAIL(LM_EINVAL, "NULL argument");
    if (!std::strcmp(name, "pq_rerank_overflow")) *value = ix->pq_overflow;
    else if (!std::strcmp(name, "pq_rerank_expanded")) *value = ix->pq_rerank_expanded ? 1 : 0;
    else if (!std::strcmp(name, "pq_threads")) *value = ix->pq_threads;
    else if (!std::strcmp(name, "speculate")) *value = ix->speculate;
    else if (!std::strcmp(name, "single_query_direct")) *value = ix->single_query_direct;
    else if (!std::strcmp(name, "speculate_max_batch")) *value = ix->speculate_max_batch;
    else LM_FAIL(LM_EINVAL, std::string("unknown readable option: ") + name);
    return LM_OK;
}

// Mean HIP-event-pair time around an EMPTY kernel on the index' stream: the fixed dispatch + event cost that
// every per-kernel event measurement (lm_search_stats.update_ms) contains.
int lm_index_event_overhead_us(lm_index* ix, double* out_us) {
    if (!ix || !out_us) LM_FAIL(LM_EINVAL, "NULL argument");
    LM_HIP(hipSetDevice(ix->device));
    const int reps = 256;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev(reps);
    int64_t* dl = nullptr;
    float* dd = nullptr;
    LM_HIP(hipMalloc((void**)&dl, 64));
    LM_HIP(hipMalloc((void**)&dd, 64));
    for (int w = 0; w < 2; ++w) {
        for (int i = 0; i < reps; ++i) {
            if (w == 0) {
                (void)hipEventCreate(&ev[i].first);
                (void)hipEventCreate(&ev[i].second);
            }
            (void)hipEventRecord(ev[i].first, ix->stream);
            hipLaunchKernelGGL(k_fill_empty, dim3(1), dim3(64), 0, ix->stream, (int64_t)0, 0, dl, dd);
            (void)hipEventRecord(ev[i].second, ix->stream);
        }
        LM_HIP(hipStreamSynchronize(ix->stream));
    }
    double ms = 0;
    for (auto& p : ev) {
        float t = 0;
        (void)hipEventElapsedTime(&t, p.first, p.second);
        ms += t;
        (void)hipEventDestroy(p.first);
        (void)hipEventDestroy(p.second);
    }
    (void)hipFree(dl);
    (void)hipFree(dd);
    *out_us = 1e3 * ms / reps;
    return LM_OK;
}

int lm_index_get_stats(const lm_index* ix, lm_search_stats* out) {
    if (!ix || !out) LM_FAIL(LM_EINVAL, "NULL argument");
    *out = ix->stats;
    return LM_OK;
}

int lm_index_search_device(lm_index* ix, int64_t n, const float* d_x, int32_t k, float* d_distances, int64_t* d_labels,
                           const lm_search_params* params) {
    if (n > 0 && (!d_x || !d_distances || !d_labels)) LM_FAIL(LM_EINVAL, "NULL buffer");
    return do_search_device(ix, n, d_x, k, d_distances, d_labels, params);
}

int lm_index_search(lm_index* ix, int64_t n, const float* x, int32_t k, float* distances, int64_t* labels,
                    const lm_search_params* params) {
    if (!ix) LM_FAIL(LM_EINVAL, "NULL index");
    if (n < 0 || k <= 0) LM_FAIL(LM_EINVAL, "bad n / k");
    if (n == 0) return LM_OK;
    if (!x || !distances || !labels) LM_FAIL(LM_EINVAL, "NULL buffer");
    LM_HIP(hipSetDevice(ix->device));
    const size_t need_x = (size_t)n * ix->D * 4, need_d = (size_t)n * k * 4, need_l = (size_t)n * k * 8;
    if (int src = ensure_stage(ix, need_x, need_d, need_l)) return src;
    float* d_x = ix->d_stage_x;
    float* d_d = ix->d_stage_d;
    int64_t* d_l = ix->d_stage_l;
    int rc = LM_OK;
    if (hipMemcpyAsync(d_x, x, need_x, hipMemcpyHostToDevice, ix->stream) != hipSuccess) {
        set_error("query upload failed");
        rc = LM_EHIP;
    }
    if (!rc) rc = do_search_device(ix, n, d_x, k, d_d, d_l, params);
    if (!rc && (hipMemcpyAsync(distances, d_d, need_d, hipMemcpyDeviceToHost, ix->stream) != hipSuccess ||
                hipMemcpyAsync(labels, d_l, need_l, hipMemcpyDeviceToHost, ix->stream) != hipSuccess ||
                hipStreamSynchronize(ix->stream) != hipSuccess)) {
        set_error("result copy failed");
        rc = LM_EHIP;
    }
    return rc;
}

int lm_dist_gather(const void* d_table, int32_t dtype, int32_t d_padded, int32_t metric, const float* d_q,
                   const int32_t* d_qidx, const int32_t* d_ids, int64_t npairs, float* d_out, void* stream) {
    if (npairs == 0) return LM_OK;
    if (!d_table || !d_q || !d_qidx || !d_ids || !d_out) LM_FAIL(LM_EINVAL, "NULL buffer");
    if (d_padded <= 0 || d_padded % 64) LM_FAIL(LM_EINVAL, "d_padded must be a positive multiple of 64");
    dim3 grid((unsigned)((npairs + 15) / 16)), block(256);
    hipStream_t st = (hipStream_t)stream;
    const bool l2 = metric == LM_METRIC_L2, f16 = dtype == LM_DTYPE_F16;
#define GO(n)                                                                                                     \
    case n:                                                                                                       \
        if (l2 && f16) hipLaunchKernelGGL((k_dist_pairs<n, true, true>), grid, block, 0, st, d_table, d_q, d_qidx, d_ids, npairs, d_out);        \
        else if (l2) hipLaunchKernelGGL((k_dist_pairs<n, true, false>), grid, block, 0, st, d_table, d_q, d_qidx, d_ids, npairs, d_out);         \
        else if (f16) hipLaunchKernelGGL((k_dist_pairs<n, false, true>), grid, block, 0, st, d_table, d_q, d_qidx, d_ids, npairs, d_out);        \
        else hipLaunchKernelGGL((k_dist_pairs<n, false, false>), grid, block, 0, st, d_table, d_q, d_qidx, d_ids, npairs, d_out);                \
        break
    switch (d_padded / 64) {
        GO(1); GO(2); GO(3); GO(4); GO(5); GO(6); GO(8); GO(12); GO(16);
        default: LM_FAIL(LM_EINVAL, "unsupported padded dimension");
    }
#undef GO
    LM_HIP(hipGetLastError());
    return LM_OK;
}

int lm_topk_merge(const int64_t* d_in_ids, const float* d_in_dist, int32_t S, int32_t B, int32_t k, int32_t metric,
                  int64_t* d_out_ids, float* d_out_dist, void* stream) {
    if (B == 0) return LM_OK;
    if (!d_in_ids || !d_in_dist || !d_out_ids || !d_out_dist || S <= 0 || k <= 0) LM_FAIL(LM_EINVAL, "bad merge arguments");
    int P2 = next_pow2(S * k);
    if (P2 > 2048) LM_FAIL(LM_EINVAL, "S*k too large for the merge kernel (<= 2048)");
    hipLaunchKernelGGL(k_topk_merge, dim3(B), dim3(64), (size_t)P2 * 16, (hipStream_t)stream, d_in_ids, d_in_dist, S, B, k,
                       metric, P2, d_out_ids, d_out_dist);
    LM_HIP(hipGetLastError());
    return LM_OK;
}

}  // extern "C"

#include "lm_pq_impl.h"

