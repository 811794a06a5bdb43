// lm_device_types.h -- device-side views of the graph / workspace, counters, constants.
// Part of lm_search.hip's translation unit (included there, in this order); see its header comment.
#pragma once

namespace lm {

// ---------------------------------------------------------------------------------------------
// device-side views
// ---------------------------------------------------------------------------------------------
struct L0Range {  // derived at load from node_offsets/level_ptr: level-0 list of node i
    uint64_t begin;
    uint32_t count;
    uint32_t pad;
};

struct GraphDev {
    int64_t N;
    int32_t entry_point, max_level;
    const uint64_t* node_offsets;
    const uint64_t* level_ptr;
    const int32_t* neighbors;
    const L0Range* l0;
};

struct WsDev {
    int32_t B, ef, W, maxnew;  // ef = pool capacity max(efSearch, k)
    int32_t efs;               // the caller's efSearch: what both faiss stop rules count against (lm_beam_common.h: select_pops)
    int32_t batch;             // dynamic batching (lm_search_params.batch_size): k_expand goes on popping while a query's new-list is shorter; 0 = off
    int32_t check_rel;         // lm_search_params.check_relative_distance (k_expand's extra pops obey the same stop rules as select_pops)
    int64_t nw;  // visited words per query
    int32_t* phase;
    int32_t* level;
    uint64_t* cur_key;
    int32_t* nsteps;
    int32_t* npool;
    int32_t* npop;
    int32_t* nnew;
    unsigned long long* ndis_q;  // per-query distance evaluations (no same-address atomics in the round kernels)
    int32_t* pop;     // B x W
    int32_t* newid;   // B x maxnew
    uint64_t* pool;   // B x ef
    uint32_t* visited;  // B x nw
    // round dedup (recompute mode)
    uint32_t* rbm;        // nw  (accumulated by k_expand, cleared by k_uniq_emit)
    uint32_t* rbm_snap;   // nw  (this round's bitmap for rank lookups)
    int32_t* word_rank;   // nw
    int32_t* tile_sum;    // ntiles
    int32_t* uniq;        // ucap
    // two-level search (prune_ratio): per-query approximate queue of (PQ-ADC distance, id) keys, bit0 = consumed
    uint64_t* aq;         // B x AQ_CAP
    int32_t* naq;         // B
    unsigned long long* nadc_q;  // B
    // per-call embedding memo (recompute_memo): every node is recomputed at most once per search call
    int32_t* memo_slot;   // N : row in `memo` or -1
    float* memo;          // memo_cap x Dp
    // counters: [0]=live queries this round [1]=n_uniq [2..] stats
    unsigned long long* counters;
};
// C_RC_TOKENS / C_RC_MAXLEN: total tokens and longest chunk of the round's unique list, written by the built-in recompute provider's
// length scan (lm_recompute.hip) so that the loop's one device-to-host copy per round carries them
enum { C_LIVE = 0, C_NUNIQ = 1, C_NDIS = 2, C_NEXPAND = 3, C_ROUNDS = 4, C_RC_TOKENS = 5, C_NADC = 6, C_RC_MAXLEN = 7, C_PQ_OVERFLOW = 8, C_NCOUNTERS = 9 };

constexpr int UNIQ_TILE = 4096;  // words per block in the uniq scan
constexpr int AQ_CAP = 512;       // capacity of the approximate queue (== ORC_AQ_CAP in oracle/lm_oracle.c)


}  // namespace lm
