// lm_kernels_persist.h -- k_search_table: persistent stored-embedding search, k_rounds_max.
// Part of lm_search.hip's translation unit (included there, in this order); see its header comment.
#pragma once

namespace lm {

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

// NT = threads per query: 256 (a workgroup of 4 waves; long new-lists: high degree x beam) or 64 (ONE wave per query: the
// traversal of a query is a chain of dependent memory round trips -- pop -> neighbour range -> ids -> visited bits -> rows --
// so throughput is queries in flight / chain latency, and a wave per query puts 4x as many queries on a CU; at beam 1 on a
// pruned graph (~9 neighbours per hop) a wave's 4 row groups x 2 rows in flight cover a hop's new-list in one pass).
// (Round 6 measured a five-waves-per-SIMD register allocation of this kernel -- __launch_bounds__(NT, 5): 96 VGPRs and 20-52 B of scratch instead of the
// compiler's 112-114 VGPRs = four waves -- against it: slower at every batch size and beam, 4.75-5.03 vs 5.08 TB/s at 8192 queries, 6.21 vs 6.50 at 32768
// (profiles/r6_table_mode_queries_in_flight_and_occupancy_sweep.json).  What raises the rate is queries in flight: 5.1 / 6.0 / 6.5 TB/s at 8192 / 16384 / 32768.)
template <int NCH, bool L2, bool F16, int NT>
__global__ __launch_bounds__(NT) void k_search_table(GraphDev g, WsDev ws, PersistArgs a) {
    extern __shared__ __align__(16) unsigned char smem[];
    __shared__ uint32_t s_off[65];
    __shared__ uint64_t s_b[64];
    __shared__ int32_t s_pop[64];
    __shared__ int s_npop, s_wcnt[NT / 64];
    __shared__ unsigned long long s_best;
    const int ef = ws.ef;
    uint64_t* lpool = (uint64_t*)smem;
    uint64_t* outp = lpool + ef;
    uint64_t* newk = outp + ef;
    int32_t* s_new = (int32_t*)(newk + a.Pmax);
    const int q = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int lane16 = tid & 15, sg = tid >> 4;
    float4 qv[NCH];
    load_query<NCH>(a.Q, q, lane16, qv);
    uint32_t* vis = ws.visited + (size_t)q * ws.nw;
    unsigned long long ndis = 0;
    int rounds = 0, nsteps = 0;

    // distances of s_new[0..n) -> newk[0..n)   (two rows in flight per 16-lane group, canonical reduction)
    auto eval_new = [&](int n) {
        for (int i = sg; i < n; i += NT / 8) {
            const int i2 = i + NT / 16;
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
        for (uint32_t j = tid; j < cnt; j += NT) s_new[j] = g.neighbors[b + j];
        if (tid == 0) s_best = KEY_NONE;
        __syncthreads();
        eval_new((int)cnt);
        __syncthreads();
        for (int i = tid; i < (int)cnt; i += NT) atomicMin(&s_best, (unsigned long long)newk[i]);
        __syncthreads();
        const uint64_t best = s_best;
        ndis += cnt;
        rounds++;
        descent_step(PH_UPPER, best, a.max_level, cur, level);
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
            const int found = select_pops(lpool, npool, ws.W, a.check_rel, ws.efs, nsteps, s_pop, tid);
            LM_WAVE_SYNC();  // s_pop[r] written by the popping lanes, read below by lane r
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
        for (uint32_t f0 = 0; f0 < totalc; f0 += NT) {
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
            for (int i = 0; i < NT / 64; ++i) total += s_wcnt[i];
            __syncthreads();
        }
        const int n = total;
        ndis += (unsigned long long)n;
        int Pn = 1;
        while (Pn < n) Pn <<= 1;
        eval_new(n);
        const bool by_count = rank_merge_unsorted_pays<NT>(npool, n);  // (round 6) a hop's few new keys: places by counting, one barrier instead of the sort's
        if (!by_count)
            for (int i = n + tid; i < Pn; i += NT) newk[i] = KEY_NONE;
        __syncthreads();
        if (n > 0) {
            if (by_count) {
                rank_merge_unsorted<NT>(lpool, npool, newk, n, outp, ef, tid);
            } else {
                sort_keys<NT>(newk, Pn, tid);
                rank_merge<NT>(lpool, npool, newk, n, outp, ef, tid);
            }
            npool = min(ef, npool + n);
            for (int i = tid; i < npool; i += NT) lpool[i] = outp[i];
            __syncthreads();
        }
    }
    // ---- results (same as k_finalize) + per-query statistics ----
    for (int i = tid; i < a.k; i += NT) {
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


}  // namespace lm
