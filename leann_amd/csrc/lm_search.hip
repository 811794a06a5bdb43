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
//   k_update_sort / k_dist_flat + k_merge   A/B variants (full bitonic sort; split distance / merge kernels)
//   k_memo_append per-call embedding memo (recompute_memo)
//   k_search_table persistent stored-embedding search: a query's whole traversal in one workgroup, one launch/batch
//   lm_pq_impl.h  DiskANN-style path: k_pq_traverse (persistent PQ-ADC traversal), k_pq_rerank
// Algorithm contract: oracle/lm_oracle.c header (set semantics under the (dist,id) total order).
// Reference call site replaced: index.search(...) leann_backend_hnsw/hnsw_backend.py:241-248.
#include <algorithm>
#include <cmath>
#include <cstring>

#include <hip/hip_fp16.h>

#include "lm_internal.h"

namespace lm {

static thread_local std::string g_err;
void set_error(const std::string& msg) { g_err = msg; }

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
    int32_t B, ef, W, maxnew;
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
    // flat (query,node) pair list of the round (split variant): segments allocated by atomicAdd
    int32_t* seg_start;   // B
    int32_t* pair_q;      // B x maxnew
    int32_t* pair_v;      // B x maxnew
    uint64_t* pair_key;   // B x maxnew
    // counters: [0]=live queries this round [1]=n_uniq [2..] stats
    unsigned long long* counters;
};
enum { C_LIVE = 0, C_NUNIQ = 1, C_NDIS = 2, C_NEXPAND = 3, C_ROUNDS = 4, C_NPAIRS = 5, C_NADC = 6, C_NCOUNTERS = 8 };

constexpr int UNIQ_TILE = 4096;  // words per block in the uniq scan
constexpr int AQ_CAP = 512;       // capacity of the approximate queue (== ORC_AQ_CAP in oracle/lm_oracle.c)

// ---------------------------------------------------------------------------------------------
// kernels
// ---------------------------------------------------------------------------------------------
__global__ void k_init(WsDev ws, int32_t max_level) {
    int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= ws.B) return;
    ws.phase[q] = PH_SEED;
    ws.level[q] = max_level;
    ws.cur_key[q] = KEY_NONE;
    ws.nsteps[q] = 0;
    ws.npool[q] = 0;
    ws.npop[q] = 0;
    ws.nnew[q] = 0;
    ws.ndis_q[q] = 0;
    ws.naq[q] = 0;
    ws.nadc_q[q] = 0;
}

__device__ __forceinline__ void nbr_range(const GraphDev& g, int32_t node, int32_t level, uint64_t& b, uint32_t& cnt) {
    // convert_to_csr.py:507-548  p = node_offsets[i] + l ; data[level_ptr[p] : level_ptr[p+1]]
    uint64_t p = g.node_offsets[node] + (uint64_t)level;
    b = g.level_ptr[p];
    cnt = (uint32_t)(g.level_ptr[p + 1] - b);
}

// one wave (64 lanes) per query.  Level-0 expansion is FLATTENED over (pop, neighbour) so that the
// dependent chain is pop -> l0 range -> neighbour ids -> visited atomic, once per 64 neighbours
// instead of once per popped node.  Dynamic LDS: maxnew ints (staging of the new-list).
__global__ __launch_bounds__(64) void k_expand(GraphDev g, WsDev ws, int use_rbm, int round_no, int flat, int defer) {
    // defer != 0: the two-level pruning kernel (k_prune) finishes the new-list: it marks the dedup bitmap and counts
    extern __shared__ int32_t s_new[];
    __shared__ uint32_t s_off[65];
    __shared__ uint64_t s_b[64];
    const int q = blockIdx.x;
    const int lane = threadIdx.x;
    const int ph = ws.phase[q];
    if (ph == PH_DONE) {
        if (lane == 0) ws.nnew[q] = 0;
        return;
    }
    int total = 0;
    if (ph == PH_SEED) {
        if (lane == 0) s_new[0] = g.entry_point;
        total = 1;
    } else if (ph == PH_UPPER) {
        uint64_t b;
        uint32_t cnt;
        nbr_range(g, key_id(ws.cur_key[q]), ws.level[q], b, cnt);
        for (uint32_t j = lane; j < cnt; j += 64) s_new[j] = g.neighbors[b + j];
        total = (int)cnt;
    } else {
        uint32_t* vis = ws.visited + (size_t)q * ws.nw;
        const int npop = ws.npop[q];
        for (int p0 = 0; p0 < npop; p0 += 64) {
            const int np = min(64, npop - p0);
            uint32_t cnt = 0;
            if (lane < np) {
                L0Range r = g.l0[ws.pop[(size_t)q * ws.W + p0 + lane]];
                s_b[lane] = r.begin;
                cnt = r.count;
            }
            uint32_t x = cnt;  // inclusive scan
            for (int d = 1; d < 64; d <<= 1) {
                uint32_t y = __shfl_up(x, d);
                if (lane >= d) x += y;
            }
            if (lane == 0) s_off[0] = 0;
            if (lane < np) s_off[lane + 1] = x;
            const uint32_t totalc = __shfl(x, np - 1);
            __syncthreads();
            for (uint32_t f0 = 0; f0 < totalc; f0 += 64) {
                const uint32_t f = f0 + lane;
                bool fresh = false;
                int32_t v = -1;
                if (f < totalc) {
                    int lo = 0, hi = np - 1;  // largest pi with s_off[pi] <= f
                    while (lo < hi) {
                        int mid = (lo + hi + 1) >> 1;
                        if (s_off[mid] <= f) lo = mid;
                        else hi = mid - 1;
                    }
                    v = g.neighbors[s_b[lo] + (f - s_off[lo])];
                    uint32_t bit = 1u << (v & 31);
                    uint32_t old = atomicOr(&vis[v >> 5], bit);
                    fresh = !(old & bit);
                }
                unsigned long long m = __ballot(fresh);
                if (fresh) s_new[total + __popcll(m & ((1ull << lane) - 1ull))] = v;
                total += __popcll(m);
            }
            __syncthreads();
        }
    }
    __syncthreads();
    int32_t* newid = ws.newid + (size_t)q * ws.maxnew;
    int start = 0;
    if (flat) {
        if (lane == 0) start = (int)atomicAdd(&ws.counters[C_NPAIRS], (unsigned long long)total);
        start = __shfl(start, 0);
    }
    for (int i = lane; i < total; i += 64) {
        const int32_t v = s_new[i];
        newid[i] = v;
        if (!defer && (use_rbm == 1 || (use_rbm == 2 && ws.memo_slot[v] < 0))) atomicOr(&ws.rbm[v >> 5], 1u << (v & 31));
        if (flat) {
            ws.pair_q[start + i] = q;
            ws.pair_v[start + i] = v;
        }
    }
    if (lane == 0) {
        ws.nnew[q] = total;
        if (flat) ws.seg_start[q] = start;
        if (!defer) ws.ndis_q[q] += (unsigned long long)total;
        // plain stores of identical values (benign): a contended same-address atomic costs ~12 ns per
        // workgroup and serialises the launch tail (MI355X_MICROARCH.md, price list row "fanin")
        ws.counters[C_LIVE] = 1ull;
        ws.counters[C_ROUNDS] = (unsigned long long)round_no;
    }
}

// round bitmap -> per-tile popcounts
__global__ __launch_bounds__(256) void k_uniq_count(WsDev ws) {
    __shared__ int red[4];
    const int64_t base = (int64_t)blockIdx.x * UNIQ_TILE;
    int s = 0;
    for (int i = threadIdx.x; i < UNIQ_TILE; i += 256) {
        int64_t w = base + i;
        if (w < ws.nw) s += __popc(ws.rbm[w]);
    }
    for (int m = 32; m >= 1; m >>= 1) s += __shfl_xor(s, m);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) ws.tile_sum[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

// per tile: exclusive ranks, sorted unique ids, snapshot + clear of the round bitmap
__global__ __launch_bounds__(256) void k_uniq_emit(WsDev ws, int ntiles) {
    __shared__ int wsum[4];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    // tile base = sum of previous tiles
    int part = 0;
    for (int t = tid; t < (int)blockIdx.x; t += 256) part += ws.tile_sum[t];
    for (int m = 32; m >= 1; m >>= 1) part += __shfl_xor(part, m);
    if (lane == 0) wsum[wv] = part;
    __syncthreads();
    int run = wsum[0] + wsum[1] + wsum[2] + wsum[3];
    __syncthreads();
    const int64_t base = (int64_t)blockIdx.x * UNIQ_TILE;
    for (int r = 0; r < UNIQ_TILE; r += 256) {
        int64_t w = base + r + tid;
        uint32_t bits = 0;
        if (w < ws.nw) {
            bits = ws.rbm[w];
            ws.rbm_snap[w] = bits;
            if (bits) ws.rbm[w] = 0;
        }
        int c = __popc(bits);
        // inclusive wave scan
        int x = c;
        for (int d = 1; d < 64; d <<= 1) {
            int y = __shfl_up(x, d);
            if (lane >= d) x += y;
        }
        if (lane == 63) wsum[wv] = x;
        __syncthreads();
        int woff = 0;
        for (int i = 0; i < wv; ++i) woff += wsum[i];
        int rowtot = wsum[0] + wsum[1] + wsum[2] + wsum[3];
        int excl = run + woff + x - c;
        if (w < ws.nw) {
            ws.word_rank[w] = excl;
            while (bits) {
                int bpos = __ffs(bits) - 1;
                bits &= bits - 1;
                ws.uniq[excl++] = (int32_t)(w * 32 + bpos);
            }
        }
        run += rowtot;
        __syncthreads();
    }
    if (blockIdx.x == (unsigned)ntiles - 1 && tid == 0) ws.counters[C_NUNIQ] = (unsigned long long)run;
}


// ---- two-level search (paper Alg. 2; prune_ratio / pruning_strategy of hnsw_backend.py:219-231) --------
// Per-query lookup tables for the whole batch: lut[q][j][c]  (canonical: oracle/lm_oracle_pq.c orc_pq_lut)
struct PruneArgs {
    const float* Q;      // B x Dp
    float* lut;          // B x m x 256
    const float* codebooks;
    const uint8_t* codes;
    int32_t Dp, metric, m, dsub;
    float keep;          // a = 1 - prune_ratio
    int32_t strategy;    // 0 global, 1 local, 2 proportional
    int32_t use_rbm;     // 1: mark the dedup bitmap, 2: only nodes without a memo row, 0: stored-embedding mode
    int32_t Pmax;        // pow2 >= maxnew
};

__global__ __launch_bounds__(256) void k_pq_lut_all(PruneArgs a) {
    const int q = blockIdx.x;
    const float* qv = a.Q + (size_t)q * a.Dp;
    float* lut = a.lut + (size_t)q * a.m * 256;
    for (int e = threadIdx.x; e < a.m * 256; e += 256) {
        const int j = e >> 8;
        const float* cb = a.codebooks + (size_t)e * a.dsub;
        const float* qs = qv + j * a.dsub;
        float acc = 0.0f;
        if (a.metric == LM_METRIC_L2) {
            for (int t = 0; t < a.dsub; ++t) {
                float d = qs[t] - cb[t];
                acc = __builtin_fmaf(d, d, acc);
            }
        } else {
            for (int t = 0; t < a.dsub; ++t) acc = __builtin_fmaf(qs[t], cb[t], acc);
            acc = -acc;
        }
        lut[e] = acc;
    }
}

// one workgroup per query: ADC of the fresh list, approximate-queue update, selection of the nodes that get
// an exact (recomputed) distance this round.  dynamic LDS: nk[Pmax] | aq[AQ_CAP] | out[AQ_CAP]  (u64)
__global__ __launch_bounds__(256) void k_prune(WsDev ws, PruneArgs a) {
    extern __shared__ __align__(16) unsigned char smem[];
    __shared__ int s_cnt;
    uint64_t* nk = (uint64_t*)smem;
    uint64_t* aq = nk + a.Pmax;
    uint64_t* out = aq + AQ_CAP;
    const int q = blockIdx.x, tid = threadIdx.x;
    const int ph = ws.phase[q];
    if (ph == PH_DONE) return;
    int32_t* newid = ws.newid + (size_t)q * ws.maxnew;
    const int n = ws.nnew[q];
    auto mark = [&](int32_t v) {
        if (a.use_rbm == 1 || (a.use_rbm == 2 && ws.memo_slot[v] < 0)) atomicOr(&ws.rbm[v >> 5], 1u << (v & 31));
    };
    if (ph != PH_BEAM) {  // seed / upper levels: no pruning, just finish what k_expand deferred
        for (int i = tid; i < n; i += 256) mark(newid[i]);
        if (tid == 0) ws.ndis_q[q] += (unsigned long long)n;
        return;
    }
    // ---- ADC of the fresh nodes: 4 lanes per vector, LUT in global/L2 ----
    const float* lut = a.lut + (size_t)q * a.m * 256;
    const int mw = a.m >> 2;
    int Pn = 1;
    while (Pn < n) Pn <<= 1;
    {
        const int r = tid & 3, gi = tid >> 2;
        for (int i0 = 0; i0 < n; i0 += 64) {
            const int i = i0 + gi;
            const int32_t v = i < n ? newid[i] : newid[0];
            const uint32_t* cw = (const uint32_t*)(a.codes + (size_t)v * a.m);
            float p = 0.0f;
            for (int w = 0; w < mw; ++w) {
                uint32_t word = cw[w];
                p = p + lut[((4 * w + r) << 8) + ((word >> (8 * r)) & 255u)];
            }
            float s01 = p + __shfl_xor(p, 1, 4);
            float tot = s01 + __shfl_xor(s01, 2, 4);
            if (r == 0 && i < n) nk[i] = make_key(tot, v);
        }
        for (int i = n + tid; i < Pn; i += 256) nk[i] = KEY_NONE;
    }
    const int naq0 = ws.naq[q];
    uint64_t* gaq = ws.aq + (size_t)q * AQ_CAP;
    if (a.strategy != 1)
        for (int i = tid; i < naq0; i += 256) aq[i] = gaq[i];
    if (tid == 0) s_cnt = 0;
    __syncthreads();
    for (unsigned k2 = 2; k2 <= (unsigned)Pn; k2 <<= 1)
        for (unsigned j = k2 >> 1; j > 0; j >>= 1) {
            for (unsigned i = tid; i < (unsigned)Pn; i += 256) {
                unsigned ixj = i ^ j;
                if (ixj > i) {
                    uint64_t x = nk[i], y = nk[ixj];
                    bool up = (i & k2) == 0;
                    if ((x > y) == up) {
                        nk[i] = y;
                        nk[ixj] = x;
                    }
                }
            }
            __syncthreads();
        }
    const int quota = (int)ceilf(a.keep * (float)n);
    int nsel = 0;
    if (a.strategy == 1) {  // local: the best of this hop
        nsel = min(quota, n);
        for (int i = tid; i < nsel; i += 256) {
            int32_t v = key_id(nk[i]);
            newid[i] = v;
            mark(v);
        }
    } else {
        // merge the sorted fresh keys into the approximate queue by rank (ids are unique: visited filter)
        const int naq1 = min(AQ_CAP, naq0 + n);
        for (int i = tid; i < naq0; i += 256) {
            uint64_t key = aq[i];
            uint64_t kk = key >> 1;
            int lo = 0, hi = n;
            while (lo < hi) {
                int mid = (lo + hi) >> 1;
                if ((nk[mid] >> 1) < kk) lo = mid + 1;
                else hi = mid;
            }
            if (i + lo < AQ_CAP) out[i + lo] = key;
        }
        for (int j = tid; j < n; j += 256) {
            uint64_t key = nk[j];
            uint64_t kk = key >> 1;
            int lo = 0, hi = naq0;
            while (lo < hi) {
                int mid = (lo + hi) >> 1;
                if ((aq[mid] >> 1) < kk) lo = mid + 1;
                else hi = mid;
            }
            if (j + lo < AQ_CAP) out[j + lo] = key;
        }
        __syncthreads();
        // global: every unconsumed entry inside the top keep-fraction of the queue;
        // proportional: the first `quota` unconsumed entries of the queue
        const int lim = a.strategy == 0 ? min(naq1, (int)ceilf(a.keep * (float)naq1)) : naq1;
        const int cap = a.strategy == 0 ? AQ_CAP : quota;
        if (tid < 64) {
            int found = 0;
            for (int base = 0; base < lim && found < cap; base += 64) {
                int i = base + tid;
                bool un = i < lim && !(out[i] & KEY_EXPANDED);
                unsigned long long m = __ballot(un);
                int r = found + __popcll(m & ((1ull << tid) - 1ull));
                if (un && r < cap) {
                    out[i] |= KEY_EXPANDED;
                    int32_t v = key_id(out[i]);
                    newid[r] = v;
                    mark(v);
                }
                found += __popcll(m);
            }
            if (tid == 0) s_cnt = min(found, cap);
        }
        __syncthreads();
        nsel = s_cnt;
        for (int i = tid; i < naq1; i += 256) gaq[i] = out[i];
        if (tid == 0) ws.naq[q] = naq1;
    }
    if (tid == 0) {
        ws.nnew[q] = nsel;
        ws.ndis_q[q] += (unsigned long long)nsel;
        ws.nadc_q[q] += (unsigned long long)n;
    }
}

// ---- canonical distance: 16 lanes per row, lane t owns float4 chunks t, t+16, ... -------------
template <int NCH, bool L2>
__device__ __forceinline__ float row_reduce(const float4 (&e)[NCH], const float4 (&qv)[NCH]) {
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        if (L2) {
            float d0 = e[i].x - qv[i].x, d1 = e[i].y - qv[i].y, d2 = e[i].z - qv[i].z, d3 = e[i].w - qv[i].w;
            a0 = __builtin_fmaf(d0, d0, a0);
            a1 = __builtin_fmaf(d1, d1, a1);
            a2 = __builtin_fmaf(d2, d2, a2);
            a3 = __builtin_fmaf(d3, d3, a3);
        } else {
            a0 = __builtin_fmaf(e[i].x, qv[i].x, a0);
            a1 = __builtin_fmaf(e[i].y, qv[i].y, a1);
            a2 = __builtin_fmaf(e[i].z, qv[i].z, a2);
            a3 = __builtin_fmaf(e[i].w, qv[i].w, a3);
        }
    }
    float s = (a0 + a1) + (a2 + a3);
    s += __shfl_xor(s, 8, 16);
    s += __shfl_xor(s, 4, 16);
    s += __shfl_xor(s, 2, 16);
    s += __shfl_xor(s, 1, 16);
    return L2 ? s : -s;
}

template <int NCH, bool F16>
__device__ __forceinline__ void load_row(const void* table, int64_t slot, int lane16, float4 (&e)[NCH]) {
    if (F16) {
        const uint2* row = (const uint2*)table + slot * (int64_t)(NCH * 16);
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            uint2 h = row[lane16 + 16 * i];
            __half2 h0 = __builtin_bit_cast(__half2, h.x), h1 = __builtin_bit_cast(__half2, h.y);
            float2 f0 = __half22float2(h0), f1 = __half22float2(h1);
            e[i] = make_float4(f0.x, f0.y, f1.x, f1.y);
        }
    } else {
        const float4* row = (const float4*)table + slot * (int64_t)(NCH * 16);
#pragma unroll
        for (int i = 0; i < NCH; ++i) e[i] = row[lane16 + 16 * i];
    }
}

struct UpdateArgs {
    const float* Q;       // B x Dp
    const void* E;        // embeddings: table (row = node id) or provider output (row = rank)
    int32_t by_rank;      // 1: row index = rank of node in this round's unique list
    int32_t check_rel;
    int32_t max_level;
    int32_t P2;           // pow2 >= ef + maxnew
    unsigned long long* tstamp;  // profiling: 2 x B wall-clock stamps (NULL = off)
};

template <int NCH, bool L2, bool F16>
__global__ __launch_bounds__(256) void k_update_sort(WsDev ws, UpdateArgs a) {
    extern __shared__ __align__(16) unsigned char smem[];
    uint64_t* keys = (uint64_t*)smem;  // P2
    __shared__ unsigned long long s_best;

    const int q = blockIdx.x;
    const int tid = threadIdx.x;
    const int ph = ws.phase[q];
    if (ph == PH_DONE) return;
    const int n = ws.nnew[q];
    const int npool0 = (ph == PH_BEAM) ? ws.npool[q] : 0;
    const int ef = ws.ef;
    uint64_t* pool = ws.pool + (size_t)q * ef;

    // stage: existing pool | new keys (filled below) | padding
    for (int i = tid; i < a.P2; i += 256) keys[i] = (i < npool0) ? pool[i] : KEY_NONE;
    if (tid == 0) s_best = KEY_NONE;

    // query slice in registers: lane t of every 16-lane group holds chunks t, t+16, ...
    const int lane16 = tid & 15, sg = tid >> 4;
    float4 qv[NCH];
    {
        const float4* qrow = (const float4*)(a.Q + (size_t)q * (NCH * 64));
#pragma unroll
        for (int i = 0; i < NCH; ++i) qv[i] = qrow[lane16 + 16 * i];
    }
    __syncthreads();

    const int32_t* newid = ws.newid + (size_t)q * ws.maxnew;
    // two rows in flight per 16-lane group
    for (int i = sg; i < n; i += 32) {
        const int i2 = i + 16;
        const bool has2 = i2 < n;
        int32_t v0 = newid[i];
        int32_t v1 = has2 ? newid[i2] : v0;
        int64_t s0 = v0, s1 = v1;
        if (a.by_rank) {
            s0 = ws.word_rank[v0 >> 5] + __popc(ws.rbm_snap[v0 >> 5] & ((1u << (v0 & 31)) - 1u));
            s1 = ws.word_rank[v1 >> 5] + __popc(ws.rbm_snap[v1 >> 5] & ((1u << (v1 & 31)) - 1u));
        }
        float4 e0[NCH], e1[NCH];
        load_row<NCH, F16>(a.E, s0, lane16, e0);
        load_row<NCH, F16>(a.E, s1, lane16, e1);
        float d0 = row_reduce<NCH, L2>(e0, qv);
        float d1 = row_reduce<NCH, L2>(e1, qv);
        if (lane16 == 0) {
            keys[npool0 + i] = make_key(d0, v0);
            if (has2) keys[npool0 + i2] = make_key(d1, v1);
        }
    }
    __syncthreads();

    if (ph != PH_BEAM) {
        // greedy descent (faiss greedy_update_nearest): best (dist,id) among the neighbours
        for (int i = tid; i < n; i += 256) atomicMin(&s_best, (unsigned long long)keys[i]);
        __syncthreads();
        if (tid == 0) {
            uint64_t best = s_best;
            int level = ws.level[q];
            int phase = ph;
            uint64_t cur = ws.cur_key[q];
            if (ph == PH_SEED) {
                cur = best;
                phase = PH_UPPER;
                level = a.max_level;
            } else {
                if (best != KEY_NONE && best < cur) cur = best;
                else level--;
            }
            if (level <= 0) {
                // faiss HNSW::search: candidates.push(nearest); search_from_candidates(level 0)
                phase = PH_BEAM;
                int32_t c = key_id(cur);
                atomicOr(&ws.visited[(size_t)q * ws.nw + (c >> 5)], 1u << (c & 31));
                pool[0] = cur | KEY_EXPANDED;
                ws.npool[q] = 1;
                ws.pop[(size_t)q * ws.W] = c;
                ws.npop[q] = 1;
                ws.nsteps[q] = 1;
            }
            ws.cur_key[q] = cur;
            ws.level[q] = level;
            ws.phase[q] = phase;
        }
        return;
    }

    // ---- level-0 beam: merge the new keys into the pool (keep the ef smallest) ----
    if (n > 0) {
        for (unsigned k2 = 2; k2 <= (unsigned)a.P2; k2 <<= 1) {
            for (unsigned j = k2 >> 1; j > 0; j >>= 1) {
                for (unsigned i = tid; i < (unsigned)a.P2; i += 256) {
                    unsigned ixj = i ^ j;
                    if (ixj > i) {
                        uint64_t x = keys[i], y = keys[ixj];
                        bool up = (i & k2) == 0;
                        if ((x > y) == up) {
                            keys[i] = y;
                            keys[ixj] = x;
                        }
                    }
                }
                __syncthreads();
            }
        }
    }
    const int npool1 = min(ef, npool0 + n);
    // ---- next pops: the W smallest unexpanded entries (wave 0) ----
    if (tid < 64) {
        const int nsteps = ws.nsteps[q];
        int allowed = ws.W;
        if (!a.check_rel) allowed = min(allowed, max(0, ef + 1 - nsteps));  // faiss: nstep > efSearch -> break
        int found = 0;
        for (int base = 0; base < npool1 && found < allowed; base += 64) {
            int i = base + tid;
            bool un = i < npool1 && !(keys[i] & KEY_EXPANDED);
            unsigned long long m = __ballot(un);
            int r = found + __popcll(m & ((1ull << tid) - 1ull));
            if (un && r < allowed) {
                keys[i] |= KEY_EXPANDED;
                ws.pop[(size_t)q * ws.W + r] = key_id(keys[i]);
            }
            found += __popcll(m);
        }
        found = min(found, allowed);
        if (tid == 0) {
            ws.npop[q] = found;
            ws.nsteps[q] = nsteps + found;
            ws.npool[q] = npool1;
            if (found == 0) ws.phase[q] = PH_DONE;
        }
    }
    __syncthreads();
    for (int i = tid; i < npool1; i += 256) pool[i] = keys[i];
}


// ---- variant 0 (default): sort only the NEW keys, then merge with the (already sorted) pool by rank ----
// LDS: pool[ef] | newk[Pn] | out[ef]   (a.P2 carries ef_lds = ef rounded up to 2, Pn is per block)
template <int NCH, bool L2, bool F16, int MODE, int NT>  // MODE 0: row = node id (table), 1: rank in the round's unique list, 2: memo slot
__device__ __forceinline__ void update_body(const WsDev& ws, const UpdateArgs& a) {
    extern __shared__ __align__(16) unsigned char smem[];
    __shared__ unsigned long long s_best;

    const int q = blockIdx.x;
    const int tid = threadIdx.x;
    const int ph = ws.phase[q];
    if (ph == PH_DONE) return;
    const int n = ws.nnew[q];
    const int ef = ws.ef;
    const int npool0 = (ph == PH_BEAM) ? ws.npool[q] : 0;
    uint64_t* pool = ws.pool + (size_t)q * ef;
    uint64_t* lpool = (uint64_t*)smem;      // ef
    uint64_t* out = lpool + ef;             // ef
    uint64_t* newk = out + ef;              // maxnew rounded up to pow2
    int Pn = 1;
    while (Pn < n) Pn <<= 1;

    for (int i = tid; i < npool0; i += NT) lpool[i] = pool[i];
    for (int i = n + tid; i < Pn; i += NT) newk[i] = KEY_NONE;
    if (tid == 0) s_best = KEY_NONE;

    const int lane16 = tid & 15, sg = tid >> 4;
    float4 qv[NCH];
    {
        const float4* qrow = (const float4*)(a.Q + (size_t)q * (NCH * 64));
#pragma unroll
        for (int i = 0; i < NCH; ++i) qv[i] = qrow[lane16 + 16 * i];
    }
    const int32_t* newid = ws.newid + (size_t)q * ws.maxnew;
    for (int i = sg; i < n; i += NT / 8) {
        const int i2 = i + NT / 16;
        const bool has2 = i2 < n;
        int32_t v0 = newid[i];
        int32_t v1 = has2 ? newid[i2] : v0;
        int64_t s0 = v0, s1 = v1;
        if (MODE == 1) {
            s0 = ws.word_rank[v0 >> 5] + __popc(ws.rbm_snap[v0 >> 5] & ((1u << (v0 & 31)) - 1u));
            s1 = ws.word_rank[v1 >> 5] + __popc(ws.rbm_snap[v1 >> 5] & ((1u << (v1 & 31)) - 1u));
        } else if (MODE == 2) {
            s0 = ws.memo_slot[v0];
            s1 = ws.memo_slot[v1];
        }
        float4 e0[NCH], e1[NCH];
        load_row<NCH, F16>(a.E, s0, lane16, e0);
        load_row<NCH, F16>(a.E, s1, lane16, e1);
        float d0 = row_reduce<NCH, L2>(e0, qv);
        float d1 = row_reduce<NCH, L2>(e1, qv);
        if (lane16 == 0) {
            newk[i] = make_key(d0, v0);
            if (has2) newk[i2] = make_key(d1, v1);
        }
    }
    __syncthreads();

    if (ph != PH_BEAM) {
        for (int i = tid; i < n; i += NT) atomicMin(&s_best, (unsigned long long)newk[i]);
        __syncthreads();
        if (tid == 0) {
            uint64_t best = s_best;
            int level = ws.level[q];
            int phase = ph;
            uint64_t cur = ws.cur_key[q];
            if (ph == PH_SEED) {
                cur = best;
                phase = PH_UPPER;
                level = a.max_level;
            } else {
                if (best != KEY_NONE && best < cur) cur = best;
                else level--;
            }
            if (level <= 0) {
                phase = PH_BEAM;
                int32_t c = key_id(cur);
                atomicOr(&ws.visited[(size_t)q * ws.nw + (c >> 5)], 1u << (c & 31));
                pool[0] = cur | KEY_EXPANDED;
                ws.npool[q] = 1;
                ws.pop[(size_t)q * ws.W] = c;
                ws.npop[q] = 1;
                ws.nsteps[q] = 1;
            }
            ws.cur_key[q] = cur;
            ws.level[q] = level;
            ws.phase[q] = phase;
        }
        return;
    }

    const int npool1 = min(ef, npool0 + n);
    uint64_t* fin = lpool;  // where the merged pool lives
    if (n > 0) {
        // bitonic sort of the new keys only
        for (unsigned k2 = 2; k2 <= (unsigned)Pn; k2 <<= 1) {
            for (unsigned j = k2 >> 1; j > 0; j >>= 1) {
                for (unsigned i = tid; i < (unsigned)Pn; i += NT) {
                    unsigned ixj = i ^ j;
                    if (ixj > i) {
                        uint64_t x = newk[i], y = newk[ixj];
                        bool up = (i & k2) == 0;
                        if ((x > y) == up) {
                            newk[i] = y;
                            newk[ixj] = x;
                        }
                    }
                }
                __syncthreads();
            }
        }
        // merge by rank: (dist,id) pairs are unique across pool U new, compare without the flag bit
        for (int i = tid; i < npool0; i += NT) {
            uint64_t key = lpool[i];
            uint64_t kk = key >> 1;
            int lo = 0, hi = n;
            while (lo < hi) {
                int mid = (lo + hi) >> 1;
                if ((newk[mid] >> 1) < kk) lo = mid + 1;
                else hi = mid;
            }
            int r = i + lo;
            if (r < ef) out[r] = key;
        }
        for (int j = tid; j < n; j += NT) {
            uint64_t key = newk[j];
            uint64_t kk = key >> 1;
            int lo = 0, hi = npool0;
            while (lo < hi) {
                int mid = (lo + hi) >> 1;
                if ((lpool[mid] >> 1) < kk) lo = mid + 1;
                else hi = mid;
            }
            int r = j + lo;
            if (r < ef) out[r] = key;
        }
        __syncthreads();
        fin = out;
    }
    if (tid < 64) {
        const int nsteps = ws.nsteps[q];
        int allowed = ws.W;
        if (!a.check_rel) allowed = min(allowed, max(0, ef + 1 - nsteps));
        int found = 0;
        for (int base = 0; base < npool1 && found < allowed; base += 64) {
            int i = base + tid;
            bool un = i < npool1 && !(fin[i] & KEY_EXPANDED);
            unsigned long long m = __ballot(un);
            int r = found + __popcll(m & ((1ull << tid) - 1ull));
            if (un && r < allowed) {
                fin[i] |= KEY_EXPANDED;
                ws.pop[(size_t)q * ws.W + r] = key_id(fin[i]);
            }
            found += __popcll(m);
        }
        found = min(found, allowed);
        if (tid == 0) {
            ws.npop[q] = found;
            ws.nsteps[q] = nsteps + found;
            ws.npool[q] = npool1;
            if (found == 0) ws.phase[q] = PH_DONE;
        }
    }
    __syncthreads();
    for (int i = tid; i < npool1; i += NT) pool[i] = fin[i];
}

template <int NCH, bool L2, bool F16, int MODE, int NT>
__global__ __launch_bounds__(NT) void k_update(WsDev ws, UpdateArgs a) {
    // profiling: per-workgroup start/end stamps of the constant-rate wall clock; k_span turns them into the
    // launch's execution span (max end - min start), the quantity rocprofv3 reports as the kernel duration
    unsigned long long t0 = 0;
    if (a.tstamp && threadIdx.x == 0) t0 = wall_clock64();
    update_body<NCH, L2, F16, MODE, NT>(ws, a);
    if (a.tstamp && threadIdx.x == 0) {
        a.tstamp[2 * blockIdx.x] = t0;
        a.tstamp[2 * blockIdx.x + 1] = wall_clock64();
    }
}

__global__ __launch_bounds__(256) void k_span(const unsigned long long* tstamp, int nblocks, unsigned long long* acc) {
    __shared__ unsigned long long smin[4], smax[4];
    unsigned long long lo = ~0ull, hi = 0;
    for (int i = threadIdx.x; i < nblocks; i += 256) {
        lo = min(lo, tstamp[2 * i]);
        hi = max(hi, tstamp[2 * i + 1]);
    }
    for (int m = 32; m >= 1; m >>= 1) {
        lo = min(lo, (unsigned long long)__shfl_xor(lo, m));
        hi = max(hi, (unsigned long long)__shfl_xor(hi, m));
    }
    if ((threadIdx.x & 63) == 0) {
        smin[threadIdx.x >> 6] = lo;
        smax[threadIdx.x >> 6] = hi;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        lo = min(min(smin[0], smin[1]), min(smin[2], smin[3]));
        hi = max(max(smax[0], smax[1]), max(smax[2], smax[3]));
        acc[0] += hi - lo;  // ticks
        acc[1] += 1;
    }
}

// ---- variant 2 (split): flat, perfectly balanced distance kernel over the round's pair list ----------
template <int NCH, bool L2, bool F16>
__global__ __launch_bounds__(256) void k_dist_flat(WsDev ws, UpdateArgs a) {
    const int total = (int)ws.counters[C_NPAIRS];
    const int lane16 = threadIdx.x & 15, sg = threadIdx.x >> 4;
    for (int base = blockIdx.x * 32; base < total; base += gridDim.x * 32) {
        const int p0 = base + sg, p1 = p0 + 16;
        if (p0 >= total) continue;
        const bool has2 = p1 < total;
        const int32_t v0 = ws.pair_v[p0], q0 = ws.pair_q[p0];
        const int32_t v1 = has2 ? ws.pair_v[p1] : v0, q1 = has2 ? ws.pair_q[p1] : q0;
        int64_t s0 = v0, s1 = v1;
        if (a.by_rank) {
            s0 = ws.word_rank[v0 >> 5] + __popc(ws.rbm_snap[v0 >> 5] & ((1u << (v0 & 31)) - 1u));
            s1 = ws.word_rank[v1 >> 5] + __popc(ws.rbm_snap[v1 >> 5] & ((1u << (v1 & 31)) - 1u));
        }
        float4 e0[NCH], e1[NCH], qa[NCH], qb[NCH];
        load_row<NCH, F16>(a.E, s0, lane16, e0);
        load_row<NCH, F16>(a.E, s1, lane16, e1);
        const float4* qr0 = (const float4*)(a.Q + (size_t)q0 * (NCH * 64));
        const float4* qr1 = (const float4*)(a.Q + (size_t)q1 * (NCH * 64));
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            qa[i] = qr0[lane16 + 16 * i];
            qb[i] = qr1[lane16 + 16 * i];
        }
        const float d0 = row_reduce<NCH, L2>(e0, qa);
        const float d1 = row_reduce<NCH, L2>(e1, qb);
        if (lane16 == 0) {
            ws.pair_key[p0] = make_key(d0, v0);
            if (has2) ws.pair_key[p1] = make_key(d1, v1);
        }
    }
}

// per-query state update from the keys of k_dist_flat (one 64-lane wave per query)
__global__ __launch_bounds__(64) void k_merge(WsDev ws, UpdateArgs a) {
    extern __shared__ __align__(16) unsigned char smem[];
    __shared__ unsigned long long s_best;
    const int q = blockIdx.x;
    const int tid = threadIdx.x;
    const int ph = ws.phase[q];
    if (ph == PH_DONE) return;
    const int n = ws.nnew[q];
    const int ef = ws.ef;
    const int npool0 = (ph == PH_BEAM) ? ws.npool[q] : 0;
    uint64_t* pool = ws.pool + (size_t)q * ef;
    uint64_t* lpool = (uint64_t*)smem;
    uint64_t* out = lpool + ef;
    uint64_t* newk = out + ef;
    const uint64_t* src = ws.pair_key + ws.seg_start[q];
    int Pn = 1;
    while (Pn < n) Pn <<= 1;
    for (int i = tid; i < npool0; i += 64) lpool[i] = pool[i];
    for (int i = tid; i < Pn; i += 64) newk[i] = i < n ? src[i] : KEY_NONE;
    if (tid == 0) s_best = KEY_NONE;
    __syncthreads();
    if (ph != PH_BEAM) {
        for (int i = tid; i < n; i += 64) atomicMin(&s_best, (unsigned long long)newk[i]);
        __syncthreads();
        if (tid == 0) {
            uint64_t best = s_best;
            int level = ws.level[q];
            int phase = ph;
            uint64_t cur = ws.cur_key[q];
            if (ph == PH_SEED) {
                cur = best;
                phase = PH_UPPER;
                level = a.max_level;
            } else {
                if (best != KEY_NONE && best < cur) cur = best;
                else level--;
            }
            if (level <= 0) {
                phase = PH_BEAM;
                int32_t c = key_id(cur);
                atomicOr(&ws.visited[(size_t)q * ws.nw + (c >> 5)], 1u << (c & 31));
                pool[0] = cur | KEY_EXPANDED;
                ws.npool[q] = 1;
                ws.pop[(size_t)q * ws.W] = c;
                ws.npop[q] = 1;
                ws.nsteps[q] = 1;
            }
            ws.cur_key[q] = cur;
            ws.level[q] = level;
            ws.phase[q] = phase;
        }
        return;
    }
    const int npool1 = min(ef, npool0 + n);
    uint64_t* fin = lpool;
    if (n > 0) {
        for (unsigned k2 = 2; k2 <= (unsigned)Pn; k2 <<= 1) {
            for (unsigned j = k2 >> 1; j > 0; j >>= 1) {
                for (unsigned i = tid; i < (unsigned)Pn; i += 64) {
                    unsigned ixj = i ^ j;
                    if (ixj > i) {
                        uint64_t x = newk[i], y = newk[ixj];
                        bool up = (i & k2) == 0;
                        if ((x > y) == up) {
                            newk[i] = y;
                            newk[ixj] = x;
                        }
                    }
                }
                __syncthreads();
            }
        }
        for (int i = tid; i < npool0; i += 64) {
            uint64_t key = lpool[i];
            uint64_t kk = key >> 1;
            int lo = 0, hi = n;
            while (lo < hi) {
                int mid = (lo + hi) >> 1;
                if ((newk[mid] >> 1) < kk) lo = mid + 1;
                else hi = mid;
            }
            if (i + lo < ef) out[i + lo] = key;
        }
        for (int j = tid; j < n; j += 64) {
            uint64_t key = newk[j];
            uint64_t kk = key >> 1;
            int lo = 0, hi = npool0;
            while (lo < hi) {
                int mid = (lo + hi) >> 1;
                if ((lpool[mid] >> 1) < kk) lo = mid + 1;
                else hi = mid;
            }
            if (j + lo < ef) out[j + lo] = key;
        }
        __syncthreads();
        fin = out;
    }
    {
        const int nsteps = ws.nsteps[q];
        int allowed = ws.W;
        if (!a.check_rel) allowed = min(allowed, max(0, ef + 1 - nsteps));
        int found = 0;
        for (int base = 0; base < npool1 && found < allowed; base += 64) {
            int i = base + tid;
            bool un = i < npool1 && !(fin[i] & KEY_EXPANDED);
            unsigned long long m = __ballot(un);
            int r = found + __popcll(m & ((1ull << tid) - 1ull));
            if (un && r < allowed) {
                fin[i] |= KEY_EXPANDED;
                ws.pop[(size_t)q * ws.W + r] = key_id(fin[i]);
            }
            found += __popcll(m);
        }
        found = min(found, allowed);
        if (tid == 0) {
            ws.npop[q] = found;
            ws.nsteps[q] = nsteps + found;
            ws.npool[q] = npool1;
            if (found == 0) ws.phase[q] = PH_DONE;
        }
    }
    __syncthreads();
    for (int i = tid; i < npool1; i += 64) pool[i] = fin[i];
}


// ---- persistent stored-embedding search: the whole traversal of a query inside ONE workgroup ----------------
// (recompute_embeddings=False path of hnsw_backend.py:189-193; also what the GPU graph builder searches with.)
// No encoder sits between the rounds in this mode, so nothing forces lock-step kernel launches: every workgroup
// walks its own query from the entry point to termination -- greedy descent on the upper levels, then the level-0
// beam with the same pop / visited / merge rules as k_expand + k_update (set semantics => identical results) --
// keeping pool, frontier and new-list in LDS.  One launch per batch, no host round trips, no launch gaps.
// dynamic LDS: lpool[ef] | out[ef] | newk[Pmax] (u64) | s_new[maxnew] (i32)
struct PersistArgs {
    const float* Q;
    const void* E;
    int32_t check_rel, max_level, Pmax, k, metric;
    int64_t* labels;
    float* dist;
    int32_t* rounds_q;
};

template <int NCH, bool L2, bool F16>
__global__ __launch_bounds__(256) void k_search_table(GraphDev g, WsDev ws, PersistArgs a) {
    extern __shared__ __align__(16) unsigned char smem[];
    __shared__ uint32_t s_off[65];
    __shared__ uint64_t s_b[64];
    __shared__ int32_t s_pop[64];
    __shared__ int s_npop, s_wcnt[4];
    __shared__ unsigned long long s_best;
    const int ef = ws.ef;
    uint64_t* lpool = (uint64_t*)smem;
    uint64_t* outp = lpool + ef;
    uint64_t* newk = outp + ef;
    int32_t* s_new = (int32_t*)(newk + a.Pmax);
    const int q = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int lane16 = tid & 15, sg = tid >> 4;
    float4 qv[NCH];
    {
        const float4* qrow = (const float4*)(a.Q + (size_t)q * (NCH * 64));
#pragma unroll
        for (int i = 0; i < NCH; ++i) qv[i] = qrow[lane16 + 16 * i];
    }
    uint32_t* vis = ws.visited + (size_t)q * ws.nw;
    unsigned long long ndis = 0;
    int rounds = 0, nsteps = 0;

    // distances of s_new[0..n) -> newk[0..n)   (two rows in flight per 16-lane group, canonical reduction)
    auto eval_new = [&](int n) {
        for (int i = sg; i < n; i += 32) {
            const int i2 = i + 16;
            const bool has2 = i2 < n;
            const int32_t v0 = s_new[i];
            const int32_t v1 = has2 ? s_new[i2] : v0;
            float4 e0[NCH], e1[NCH];
            load_row<NCH, F16>(a.E, (int64_t)v0, lane16, e0);
            load_row<NCH, F16>(a.E, (int64_t)v1, lane16, e1);
            const float d0 = row_reduce<NCH, L2>(e0, qv);
            const float d1 = row_reduce<NCH, L2>(e1, qv);
            if (lane16 == 0) {
                newk[i] = make_key(d0, v0);
                if (has2) newk[i2] = make_key(d1, v1);
            }
        }
    };

    // ---- seed: distance of the entry point ----
    if (tid == 0) {
        s_new[0] = g.entry_point;
        s_best = KEY_NONE;
    }
    __syncthreads();
    eval_new(1);
    __syncthreads();
    uint64_t cur = newk[0];
    ndis += 1;
    rounds = 1;
    // ---- upper levels: greedy descent (faiss greedy_update_nearest) ----
    for (int level = a.max_level; level > 0;) {
        uint64_t b;
        uint32_t cnt;
        nbr_range(g, key_id(cur), level, b, cnt);
        for (uint32_t j = tid; j < cnt; j += 256) s_new[j] = g.neighbors[b + j];
        if (tid == 0) s_best = KEY_NONE;
        __syncthreads();
        eval_new((int)cnt);
        __syncthreads();
        for (int i = tid; i < (int)cnt; i += 256) atomicMin(&s_best, (unsigned long long)newk[i]);
        __syncthreads();
        const uint64_t best = s_best;
        ndis += cnt;
        rounds++;
        if (best != KEY_NONE && best < cur) cur = best;
        else level--;
        __syncthreads();
    }
    // ---- level 0 ----
    if (tid == 0) {
        const int32_t c = key_id(cur);
        atomicOr(&vis[c >> 5], 1u << (c & 31));
        lpool[0] = cur;
    }
    int npool = 1;
    __syncthreads();
    for (;;) {
        if (tid < 64) {
            int allowed = ws.W;
            if (!a.check_rel) allowed = min(allowed, max(0, ef + 1 - nsteps));
            int found = 0;
            for (int base = 0; base < npool && found < allowed; base += 64) {
                int i = base + tid;
                bool un = i < npool && !(lpool[i] & KEY_EXPANDED);
                unsigned long long m = __ballot(un);
                int r = found + __popcll(m & ((1ull << tid) - 1ull));
                if (un && r < allowed) {
                    lpool[i] |= KEY_EXPANDED;
                    s_pop[r] = key_id(lpool[i]);
                }
                found += __popcll(m);
            }
            found = min(found, allowed);
            uint32_t cnt = 0;
            if (tid < found) {
                L0Range r = g.l0[s_pop[tid]];
                s_b[tid] = r.begin;
                cnt = r.count;
            }
            uint32_t x = cnt;
            for (int d = 1; d < 64; d <<= 1) {
                uint32_t y = __shfl_up(x, d);
                if (tid >= d) x += y;
            }
            if (tid == 0) {
                s_off[0] = 0;
                s_npop = found;
            }
            if (tid < found) s_off[tid + 1] = x;
        }
        __syncthreads();
        const int np = s_npop;
        if (np == 0) break;
        nsteps += np;
        rounds++;
        const uint32_t totalc = s_off[np];
        int total = 0;
        for (uint32_t f0 = 0; f0 < totalc; f0 += 256) {
            const uint32_t f = f0 + tid;
            bool fresh = false;
            int32_t v = -1;
            if (f < totalc) {
                int lo = 0, hi = np - 1;
                while (lo < hi) {
                    int mid = (lo + hi + 1) >> 1;
                    if (s_off[mid] <= f) lo = mid;
                    else hi = mid - 1;
                }
                v = g.neighbors[s_b[lo] + (f - s_off[lo])];
                uint32_t bit = 1u << (v & 31);
                uint32_t old = atomicOr(&vis[v >> 5], bit);
                fresh = !(old & bit);
            }
            unsigned long long m = __ballot(fresh);
            if (lane == 0) s_wcnt[wv] = __popcll(m);
            __syncthreads();
            int woff = 0;
            for (int i = 0; i < wv; ++i) woff += s_wcnt[i];
            if (fresh) s_new[total + woff + __popcll(m & ((1ull << lane) - 1ull))] = v;
            total += s_wcnt[0] + s_wcnt[1] + s_wcnt[2] + s_wcnt[3];
            __syncthreads();
        }
        const int n = total;
        ndis += (unsigned long long)n;
        int Pn = 1;
        while (Pn < n) Pn <<= 1;
        eval_new(n);
        for (int i = n + tid; i < Pn; i += 256) newk[i] = KEY_NONE;
        __syncthreads();
        if (n > 0) {
            for (unsigned k2 = 2; k2 <= (unsigned)Pn; k2 <<= 1) {
                for (unsigned j = k2 >> 1; j > 0; j >>= 1) {
                    for (unsigned i = tid; i < (unsigned)Pn; i += 256) {
                        unsigned ixj = i ^ j;
                        if (ixj > i) {
                            uint64_t x = newk[i], y = newk[ixj];
                            bool up = (i & k2) == 0;
                            if ((x > y) == up) {
                                newk[i] = y;
                                newk[ixj] = x;
                            }
                        }
                    }
                    __syncthreads();
                }
            }
            for (int i = tid; i < npool; i += 256) {
                uint64_t key = lpool[i];
                uint64_t kk = key >> 1;
                int lo = 0, hi = n;
                while (lo < hi) {
                    int mid = (lo + hi) >> 1;
                    if ((newk[mid] >> 1) < kk) lo = mid + 1;
                    else hi = mid;
                }
                if (i + lo < ef) outp[i + lo] = key;
            }
            for (int j = tid; j < n; j += 256) {
                uint64_t key = newk[j];
                uint64_t kk = key >> 1;
                int lo = 0, hi = npool;
                while (lo < hi) {
                    int mid = (lo + hi) >> 1;
                    if ((lpool[mid] >> 1) < kk) lo = mid + 1;
                    else hi = mid;
                }
                if (j + lo < ef) outp[j + lo] = key;
            }
            __syncthreads();
            npool = min(ef, npool + n);
            for (int i = tid; i < npool; i += 256) lpool[i] = outp[i];
            __syncthreads();
        }
    }
    // ---- results (same as k_finalize) + per-query statistics ----
    for (int i = tid; i < a.k; i += 256) {
        const size_t t = (size_t)q * a.k + i;
        if (i < npool) {
            const float d = key_dist(lpool[i]);
            a.labels[t] = key_id(lpool[i]);
            a.dist[t] = a.metric == LM_METRIC_L2 ? d : -d;
        } else {
            a.labels[t] = -1;
            a.dist[t] = a.metric == LM_METRIC_L2 ? __builtin_inff() : -__builtin_inff();
        }
    }
    if (tid == 0) {
        ws.ndis_q[q] = ndis;
        ws.nsteps[q] = nsteps;
        ws.nadc_q[q] = 0;
        a.rounds_q[q] = rounds;
    }
}

__global__ __launch_bounds__(256) void k_rounds_max(WsDev ws, const int32_t* rounds_q) {
    __shared__ int red[4];
    int r = 0;
    for (int q = threadIdx.x; q < ws.B; q += 256) r = max(r, rounds_q[q]);
    for (int m = 32; m >= 1; m >>= 1) r = max(r, __shfl_xor(r, m));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = r;
    __syncthreads();
    if (threadIdx.x == 0) ws.counters[C_ROUNDS] = (unsigned long long)max(max(red[0], red[1]), max(red[2], red[3]));
}

__global__ void k_finalize(WsDev ws, int32_t k, int32_t metric, int64_t* labels, float* dist) {
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= ws.B * k) return;
    int q = t / k, i = t % k;
    if (i < ws.npool[q]) {
        uint64_t key = ws.pool[(size_t)q * ws.ef + i];
        float d = key_dist(key);
        labels[t] = key_id(key);
        dist[t] = metric == LM_METRIC_L2 ? d : -d;
    } else {
        labels[t] = -1;
        dist[t] = metric == LM_METRIC_L2 ? __builtin_inff() : -__builtin_inff();
    }
}

// append this round's fresh embeddings to the per-call memo and publish their slots
__global__ __launch_bounds__(256) void k_memo_append(WsDev ws, const float* e_new, int32_t nu, int64_t base, int32_t Dp) {
    const int64_t nvec = (int64_t)nu * (Dp / 4);
    const float4* src = (const float4*)e_new;
    float4* dst = (float4*)(ws.memo + base * Dp);
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (int64_t)gridDim.x * 256) dst[i] = src[i];
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nu; i += (int64_t)gridDim.x * 256)
        ws.memo_slot[ws.uniq[i]] = (int32_t)(base + i);
}

// hub cache without the per-call memo: forget this round's fresh rows again (their slots go back to -1)
__global__ __launch_bounds__(256) void k_memo_release(WsDev ws, int32_t nu) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nu; i += (int64_t)gridDim.x * 256) ws.memo_slot[ws.uniq[i]] = -1;
}

// end-of-search totals: nexpand = sum of per-query pops (nsteps), ndis = sum of per-query evaluations
__global__ __launch_bounds__(256) void k_stats(WsDev ws) {
    __shared__ unsigned long long red[2][4];
    __shared__ unsigned long long red2[4];
    unsigned long long a = 0, b = 0, c = 0;
    for (int q = threadIdx.x; q < ws.B; q += 256) {
        a += ws.ndis_q[q];
        b += (unsigned long long)ws.nsteps[q];
        c += ws.nadc_q[q];
    }
    for (int m = 32; m >= 1; m >>= 1) {
        a += __shfl_xor(a, m);
        b += __shfl_xor(b, m);
        c += __shfl_xor(c, m);
    }
    if ((threadIdx.x & 63) == 0) {
        red[0][threadIdx.x >> 6] = a;
        red[1][threadIdx.x >> 6] = b;
        red2[threadIdx.x >> 6] = c;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        ws.counters[C_NDIS] = red[0][0] + red[0][1] + red[0][2] + red[0][3];
        ws.counters[C_NEXPAND] = red[1][0] + red[1][1] + red[1][2] + red[1][3];
        ws.counters[C_NADC] = red2[0] + red2[1] + red2[2] + red2[3];
    }
}

__global__ void k_fill_empty(int64_t n, int32_t metric, int64_t* labels, float* dist) {
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    labels[t] = -1;
    dist[t] = metric == LM_METRIC_L2 ? __builtin_inff() : -__builtin_inff();
}

// pad queries [n][D] -> [n][Dp]
__global__ void k_pad_rows(const float* x, int64_t n, int32_t D, int32_t Dp, float* out) {
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n * Dp) return;
    int64_t r = t / Dp;
    int32_t c = (int32_t)(t % Dp);
    out[t] = c < D ? x[r * D + c] : 0.0f;
}

__global__ void k_pad_rows_f16(const __half* x, int64_t n, int32_t D, int32_t Dp, __half* out) {
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n * Dp) return;
    int64_t r = t / Dp;
    int32_t c = (int32_t)(t % Dp);
    out[t] = c < D ? x[r * D + c] : __float2half(0.0f);
}

// stand-alone pair distances (parity tests)
template <int NCH, bool L2, bool F16>
__global__ __launch_bounds__(256) void k_dist_pairs(const void* table, const float* Q, const int32_t* qidx,
                                                    const int32_t* ids, int64_t npairs, float* out) {
    const int lane16 = threadIdx.x & 15;
    int64_t p = (int64_t)blockIdx.x * 16 + (threadIdx.x >> 4);
    if (p >= npairs) return;
    float4 qv[NCH], e[NCH];
    const float4* qrow = (const float4*)(Q + (size_t)qidx[p] * (NCH * 64));
#pragma unroll
    for (int i = 0; i < NCH; ++i) qv[i] = qrow[lane16 + 16 * i];
    load_row<NCH, F16>(table, ids[p], lane16, e);
    float d = row_reduce<NCH, L2>(e, qv);
    if (lane16 == 0) out[p] = d;
}

// per-query merge of S shard lists (one 64-lane block per query, LDS bitonic)
__global__ __launch_bounds__(64) void k_topk_merge(const int64_t* in_ids, const float* in_dist, int S, int B, int k,
                                                   int metric, int P2, int64_t* out_ids, float* out_dist) {
    extern __shared__ __align__(16) unsigned char smem[];
    uint64_t* keys = (uint64_t*)smem;          // P2 : (dist, slot)
    const int q = blockIdx.x, tid = threadIdx.x;
    const int tot = S * k;
    // key = (internal dist, id) cannot hold 63-bit ids: sort by (dist, id) with a 2-word compare
    int64_t* ids = (int64_t*)(keys + P2);       // P2
    for (int i = tid; i < P2; i += 64) {
        if (i < tot) {
            int s = i / k, j = i % k;
            size_t src = ((size_t)s * B + q) * k + j;
            int64_t id = in_ids[src];
            float d = metric == LM_METRIC_L2 ? in_dist[src] : -in_dist[src];
            if (id < 0) {
                keys[i] = KEY_NONE;
                ids[i] = INT64_MAX;
            } else {
                keys[i] = make_key(d, 0) >> 32;  // ordered 32-bit distance
                ids[i] = id;
            }
        } else {
            keys[i] = KEY_NONE;
            ids[i] = INT64_MAX;
        }
    }
    __syncthreads();
    for (unsigned k2 = 2; k2 <= (unsigned)P2; k2 <<= 1)
        for (unsigned j = k2 >> 1; j > 0; j >>= 1) {
            for (unsigned i = tid; i < (unsigned)P2; i += 64) {
                unsigned ixj = i ^ j;
                if (ixj > i) {
                    uint64_t x = keys[i], y = keys[ixj];
                    int64_t xi = ids[i], yi = ids[ixj];
                    bool gt = x > y || (x == y && xi > yi);
                    bool up = (i & k2) == 0;
                    if (gt == up) {
                        keys[i] = y; keys[ixj] = x;
                        ids[i] = yi; ids[ixj] = xi;
                    }
                }
            }
            __syncthreads();
        }
    for (int i = tid; i < k; i += 64) {
        size_t dst = (size_t)q * k + i;
        if (keys[i] == KEY_NONE) {
            out_ids[dst] = -1;
            out_dist[dst] = metric == LM_METRIC_L2 ? __builtin_inff() : -__builtin_inff();
        } else {
            float d = key_dist(keys[i] << 32);
            out_ids[dst] = ids[i];
            out_dist[dst] = metric == LM_METRIC_L2 ? d : -d;
        }
    }
}

}  // namespace lm

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
    hipStream_t stream = nullptr;
    // workspace
    WsDev ws{};
    int32_t ws_B = 0, ws_ef = 0, ws_W = 0, ws_maxnew = 0;
    int64_t ws_ucap = 0;
    int32_t* d_memo_slot = nullptr;
    float* d_memo = nullptr;
    int64_t memo_cap = 0;
    int32_t* d_hub_slot_init = nullptr;  // N: slot of every hub node, -1 elsewhere (hub-embedding cache)
    int64_t hub_n = 0;
    std::vector<void*> ws_allocs;
    float* d_qpad = nullptr;
    int64_t qpad_cap = 0;
    unsigned long long* h_counters = nullptr;  // pinned
    // stats / profiling
    lm_search_stats stats{};
    bool profiling = false;
    int update_variant = 0;  // 0: auto (fused), 1: fused + full bitonic sort, 2: split, 3: fused wave-per-query, 4: fused workgroup-per-query
    int persistent_table = 1;  // stored-embedding mode: one persistent launch per batch (0: lock-step rounds, for A/B)
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
    ix->ws_B = ix->ws_ef = ix->ws_W = ix->ws_maxnew = 0;
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

static int ensure_ws(lm_index* ix, int32_t B, int32_t ef, int32_t W, bool prune = false) {
    int32_t maxnew = std::max({W * ix->maxdeg0, ix->maxdeg_up, 1});
    if (prune) maxnew = std::max(maxnew, (int32_t)AQ_CAP);
    if (B <= ix->ws_B && ef == ix->ws_ef && W == ix->ws_W && maxnew == ix->ws_maxnew) {
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
    ix->ws_ucap = std::min<int64_t>(ix->N, (int64_t)B * std::max(maxnew, ef));
    A(uniq, ix->ws_ucap);
    A(seg_start, B); A(pair_q, (size_t)B * maxnew); A(pair_v, (size_t)B * maxnew); A(pair_key, (size_t)B * maxnew);
    A(counters, C_NCOUNTERS);
#undef A
    LM_HIP(hipMemsetAsync(w.rbm, 0, w.nw * 4, ix->stream));
    ix->ws_B = B; ix->ws_ef = ef; ix->ws_W = W; ix->ws_maxnew = maxnew;
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
    const bool sortv = ix->update_variant == 1;
    // one 64-lane wave per query when the per-round new-list is short (low degree x beam): keeps all 16-lane
    // groups busy and quadruples the queries resident per CU; variant 3 forces it, variant 4 forbids it
    // measured (profiles/r1_kernel_ab_wave_vs_wg.txt): the wave form wins only with >= 4096 queries in flight and
    // <= ~48 expected new nodes per query per round (beam x mean level-0 degree)
    const bool wave = ix->update_variant == 3 ||
                      (ix->update_variant == 0 && ix->ws.B >= 4096 && ix->ws.W * ix->avg_degree0 <= (double)ix->wave_maxnew);
    if (ix->update_variant == 2) {
        // split: flat distance kernel (grid-stride over the pair list) + one-wave-per-query merge
        long cap = (long)ix->ws.B * ix->ws.maxnew;
        dim3 fg((unsigned)std::max<long>(1, std::min<long>((cap + 31) / 32, 256L * 20)));
        switch (ix->Dp / 64) {
#define CASEF(n) case n: hipLaunchKernelGGL((k_dist_flat<n, L2, F16>), fg, block, 0, ix->stream, ix->ws, a); break
            CASEF(1); CASEF(2); CASEF(3); CASEF(4); CASEF(5); CASEF(6); CASEF(8); CASEF(12); CASEF(16);
#undef CASEF
            default: LM_FAIL(LM_EINVAL, "unsupported padded dimension (supported: 64..384, 512, 768, 1024)");
        }
        hipLaunchKernelGGL(k_merge, grid, dim3(64), shmem, ix->stream, ix->ws, a);
        LM_HIP(hipGetLastError());
        return LM_OK;
    }
    switch (ix->Dp / 64) {
#define CASE(n)                                                                                              \
    case n:                                                                                                  \
        if (sortv) hipLaunchKernelGGL((k_update_sort<n, L2, F16>), grid, block, shmem, ix->stream, ix->ws, a); \
        else if (a.by_rank == 2) hipLaunchKernelGGL((k_update<n, L2, F16, 2, 256>), grid, block, shmem, ix->stream, ix->ws, a); \
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
    size_t shmem = ix->update_variant == 1 ? (size_t)a.P2 * sizeof(uint64_t)
                                           : (size_t)(2 * ix->ws.ef + next_pow2(ix->ws.maxnew)) * sizeof(uint64_t);
    bool l2 = ix->metric == LM_METRIC_L2;
    if (l2) return f16 ? launch_update_nch<true, true>(ix, a, shmem) : launch_update_nch<true, false>(ix, a, shmem);
    return f16 ? launch_update_nch<false, true>(ix, a, shmem) : launch_update_nch<false, false>(ix, a, shmem);
}

template <bool L2, bool F16>
static int launch_persist_nch(lm_index* ix, const PersistArgs& a, const GraphDev& g, size_t shmem) {
    dim3 grid(ix->ws.B), block(256);
    switch (ix->Dp / 64) {
#define CASEP(n)                                                                                                         \
    case n:                                                                                                              \
        LM_HIP(hipFuncSetAttribute((const void*)k_search_table<n, L2, F16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem)); \
        hipLaunchKernelGGL((k_search_table<n, L2, F16>), grid, block, shmem, ix->stream, g, ix->ws, a);                   \
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
    hipStream_t st = ix->stream;
    if ((int64_t)B > ix->pq_cap) {
        if (ix->d_pq_nadc) (void)hipFree(ix->d_pq_nadc);
        if (ix->d_pq_rounds) (void)hipFree(ix->d_pq_rounds);
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
        if (ix->update_variant == 2) LM_FAIL(LM_EINVAL, "two-level search is not available with the split update variant");
    }
    int rc = ensure_ws(ix, B, ef, W, prune);
    if (rc) return rc;
    WsDev& ws = ix->ws;
    hipStream_t st = ix->stream;
    const bool recompute = prm.recompute != 0;
    PruneArgs pa{};
    size_t prune_shmem = 0;
    if (prune) {
        const int64_t need = (int64_t)B * ix->pq_m * 256;
        if (need > ix->lut_cap) {
            if (ix->d_lut) (void)hipFree(ix->d_lut);
            ix->d_lut = nullptr;
            LM_HIP(hipMalloc((void**)&ix->d_lut, (size_t)need * 4));
            ix->lut_cap = need;
        }
        pa.Q = d_q; pa.lut = ix->d_lut; pa.codebooks = ix->d_pq_codebooks; pa.codes = ix->d_pq_codes;
        pa.Dp = ix->Dp; pa.metric = ix->metric; pa.m = ix->pq_m; pa.dsub = ix->D / ix->pq_m;
        pa.keep = 1.0f - prm.pq_pruning_ratio;
        pa.strategy = prm.local_prune ? 1 : (prm.send_neigh_times_ratio > 1e-6f ? 2 : 0);  // hnsw_backend.py:222-231
        pa.Pmax = next_pow2(ws.maxnew);
        prune_shmem = ((size_t)pa.Pmax + 2 * AQ_CAP) * 8;
        hipLaunchKernelGGL(k_pq_lut_all, dim3(B), dim3(256), 0, st, pa);
    }
    const bool slots_ok = ix->update_variant != 1 && ix->update_variant != 2;
    const bool hub = recompute && ix->hub_n > 0 && slots_ok;
    const bool memo_call = recompute && prm.recompute_memo != 0 && slots_ok;  // keep rows until the call returns
    const bool memo = memo_call || hub;                                       // rows are addressed through memo_slot
    int64_t memo_used = 0;
    if (memo) {
        if (!ix->d_memo_slot) LM_HIP(hipMalloc((void**)&ix->d_memo_slot, (size_t)ix->N * 4));
        const int64_t want = std::min<int64_t>(ix->N, 16ll << 20);
        if (ix->memo_cap < want) {
            if (hub) LM_FAIL(LM_ESTATE, "internal: memo buffer must be allocated when the hub cache is set");
            if (ix->d_memo) (void)hipFree(ix->d_memo);
            ix->d_memo = nullptr;
            LM_HIP(hipMalloc((void**)&ix->d_memo, (size_t)want * ix->Dp * 4));
            ix->memo_cap = want;
        }
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
    const int flat = ix->update_variant == 2 ? 1 : 0;

    LM_HIP(hipMemsetAsync(ws.visited, 0, (size_t)B * ws.nw * 4, st));
    LM_HIP(hipMemsetAsync(ws.counters, 0, C_NCOUNTERS * sizeof(unsigned long long), st));
    hipLaunchKernelGGL(k_init, dim3((B + 255) / 256), dim3(256), 0, st, ws, ix->max_level);

    UpdateArgs ua{};
    unsigned long long* span_acc = nullptr;
    if (ix->profiling) {
        if (ix->tstamp_cap < B) {
            if (ix->d_tstamp) (void)hipFree(ix->d_tstamp);
            ix->d_tstamp = nullptr;
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
    ua.P2 = next_pow2(ef + ws.maxnew);
    if ((size_t)ua.P2 * 8 > 64 * 1024) LM_FAIL(LM_EINVAL, "efSearch * beam too large for the LDS pool (ef + beam*degree <= 8192)");
    const int ntiles = (int)((ws.nw + UNIQ_TILE - 1) / UNIQ_TILE);
    const int sync_every = recompute ? 1 : 4;
    int64_t rounds = 0;
    unsigned long long* hc = ix->h_counters;

    for (;;) {
        LM_HIP(hipMemsetAsync(ws.counters + C_LIVE, 0, sizeof(unsigned long long), st));
        if (flat) LM_HIP(hipMemsetAsync(ws.counters + C_NPAIRS, 0, sizeof(unsigned long long), st));
        {
            EvScope es(ix, &ix->ev_expand);
            hipLaunchKernelGGL(k_expand, dim3(B), dim3(64), (size_t)ws.maxnew * sizeof(int32_t), st, g, ws, recompute ? (memo ? 2 : 1) : 0,
                               (int)(rounds + 1), flat, prune ? 1 : 0);
            if (prune) {
                pa.use_rbm = recompute ? (memo ? 2 : 1) : 0;
                hipLaunchKernelGGL(k_prune, dim3(B), dim3(256), prune_shmem, st, ws, pa);
            }
            if (recompute) {
                hipLaunchKernelGGL(k_uniq_count, dim3(ntiles), dim3(256), 0, st, ws);
                hipLaunchKernelGGL(k_uniq_emit, dim3(ntiles), dim3(256), 0, st, ws, ntiles);
            }
        }
        rounds++;
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
            if (nu > 0) {
                EvScope es(ix, &ix->ev_provider);
                int prc = ix->provider(ix->provider_user, ws.uniq, nu, &d_e, (void*)st);
                if (prc != 0 || !d_e) LM_FAIL(LM_EPROVIDER, "embedding provider failed (rc=" + std::to_string(prc) + ")");
            }
            if (memo) {
                if (nu > 0) {
                    if (memo_used + nu > ix->memo_cap) LM_FAIL(LM_ESTATE, "recompute memo is full (more than 16M distinct nodes in one call)");
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
        if (span_acc && ix->update_variant == 0) hipLaunchKernelGGL(k_span, dim3(1), dim3(256), 0, st, ix->d_tstamp, B, span_acc);
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

static int do_search_device(lm_index* ix, int64_t n, const float* d_x, int32_t k, float* d_dist, int64_t* d_labels,
                            const lm_search_params* params) {
    if (!ix || !params || n < 0 || k <= 0) LM_FAIL(LM_EINVAL, "bad search arguments");
    lm_search_params prm = *params;
    if (prm.efSearch <= 0) LM_FAIL(LM_EINVAL, "efSearch must be positive");
    LM_HIP(hipSetDevice(ix->device));
    ix->stats = lm_search_stats{};
    ix->span_ms = 0;
    ix->span_launches = 0;
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
        if (!prm.recompute && prm.pq_pruning_ratio <= 0.0f && ix->persistent_table && ix->update_variant == 0 && std::max(prm.beam_size, 1) <= 64)
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
const char* lm_version(void) { return "leann-mi355x 0.1 (gfx950)"; }

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
    if (ix->d_pq_codebooks) (void)hipFree(ix->d_pq_codebooks);
    if (ix->d_pq_codes) (void)hipFree(ix->d_pq_codes);
    if (ix->d_pq_nadc) (void)hipFree(ix->d_pq_nadc);
    if (ix->d_pq_rounds) (void)hipFree(ix->d_pq_rounds);
    if (ix->d_lut) (void)hipFree(ix->d_lut);
    if (ix->d_tstamp) (void)hipFree(ix->d_tstamp);
    if (ix->d_table && ix->table_owned) (void)hipFree(ix->d_table);
    if (ix->d_qpad) (void)hipFree(ix->d_qpad);
    if (ix->h_counters) (void)hipHostFree(ix->h_counters);
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
        LM_HIP(hipMemcpy(tmp, table, (size_t)ntotal * d * es, hipMemcpyHostToDevice));
        int64_t tot = ntotal * ix->Dp;
        if (dtype == LM_DTYPE_F32)
            hipLaunchKernelGGL(k_pad_rows, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, 0, (const float*)tmp, ntotal, d, ix->Dp, (float*)dst);
        else
            hipLaunchKernelGGL(k_pad_rows_f16, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, 0, (const __half*)tmp, ntotal, d, ix->Dp, (__half*)dst);
        LM_HIP(hipDeviceSynchronize());
        (void)hipFree(tmp);
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
    const int64_t want = std::min<int64_t>(ix->N, 16ll << 20);
    if (n > want / 2) LM_FAIL(LM_EINVAL, "hub cache too large (more than half of the memo capacity)");
    if (ix->memo_cap < want) {
        if (ix->d_memo) (void)hipFree(ix->d_memo);
        ix->d_memo = nullptr;
        LM_HIP(hipMalloc((void**)&ix->d_memo, (size_t)want * ix->Dp * 4));
        ix->memo_cap = want;
    }
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
        if (value < 0 || value > 4) LM_FAIL(LM_EINVAL, "update_variant must be 0..4");
        ix->update_variant = (int)value;
        return LM_OK;
    }
    if (!std::strcmp(name, "persistent_table")) {
        ix->persistent_table = value != 0;
        return LM_OK;
    }
    if (!std::strcmp(name, "wave_maxnew")) {
        ix->wave_maxnew = (int)value;
        return LM_OK;
    }
    LM_FAIL(LM_EINVAL, std::string("unknown option: ") + name);
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
    float* d_x = nullptr;
    float* d_d = nullptr;
    int64_t* d_l = nullptr;
    LM_HIP(hipMalloc((void**)&d_x, (size_t)n * ix->D * 4));
    LM_HIP(hipMalloc((void**)&d_d, (size_t)n * k * 4));
    LM_HIP(hipMalloc((void**)&d_l, (size_t)n * k * 8));
    int rc = LM_OK;
    if (hipMemcpyAsync(d_x, x, (size_t)n * ix->D * 4, hipMemcpyHostToDevice, ix->stream) != hipSuccess) rc = LM_EHIP;
    if (!rc) rc = do_search_device(ix, n, d_x, k, d_d, d_l, params);
    if (!rc && (hipMemcpyAsync(distances, d_d, (size_t)n * k * 4, hipMemcpyDeviceToHost, ix->stream) != hipSuccess ||
                hipMemcpyAsync(labels, d_l, (size_t)n * k * 8, hipMemcpyDeviceToHost, ix->stream) != hipSuccess ||
                hipStreamSynchronize(ix->stream) != hipSuccess)) {
        set_error("result copy failed");
        rc = LM_EHIP;
    }
    (void)hipFree(d_x);
    (void)hipFree(d_d);
    (void)hipFree(d_l);
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
