// lm_kernels_update.h -- fused update kernel k_update (gather + distance + beam update, one round of the lock-step search), k_span.
// Part of lm_search.hip's translation unit (included there, in this order); see its header comment.
// The full-sort and the split (flat distance kernel + merge kernel) forms of round 1 lost their A/B on the MI355X
// (fused 4.33 vs 4.30 TB/s at beam 1, 5.75 vs 5.22 TB/s at beam 4) and were removed.
#pragma once

namespace lm {

struct UpdateArgs {
    const float* Q;       // B x Dp
    const void* E;        // embeddings: table (row = node id), provider output (row = rank) or memo (row = slot)
    int32_t by_rank;      // 0: row = node id; 1: rank of the node in this round's unique list; 2: memo slot
    int32_t identity;     // by_rank == 1 only: row i = the query's i-th new node (a one-query pass hands its new-list to the provider as it is)
    int32_t check_rel;
    int32_t max_level;
    int32_t P2;           // k_pq_rerank only: pow2 >= candidates per query
    unsigned long long* tstamp;  // profiling: 2 x B wall-clock stamps (NULL = off)
};

// sort only the NEW keys, then merge with the (already sorted) pool by rank
// LDS: pool[ef] | out[ef] | newk[Pmax]
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
    load_query<NCH>(a.Q, q, lane16, qv);
    const int32_t* newid = ws.newid + (size_t)q * ws.maxnew;
    for (int i = sg; i < n; i += NT / 8) {  // two rows in flight per 16-lane group
        const int i2 = i + NT / 16;
        const bool has2 = i2 < n;
        int32_t v0 = newid[i];
        int32_t v1 = has2 ? newid[i2] : v0;
        int64_t s0 = v0, s1 = v1;
        if (MODE == 1 && a.identity) {
            s0 = i;
            s1 = has2 ? i2 : i;
        } else if (MODE == 1) {
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
            uint64_t cur = ws.cur_key[q];
            int level = ws.level[q];
            int phase = PH_UPPER;
            if (descent_step(ph, s_best, a.max_level, cur, level)) {
                phase = PH_BEAM;
                const int32_t c = key_id(cur);
                atomicOr(&ws.visited[(size_t)q * ws.nw + (c >> 5)], 1u << (c & 31));
                // the seed is always the first pop: count_below(d0) = 0 < efSearch and nstep = 0 (efSearch >= 1, host-checked)
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
        if (rank_merge_unsorted_pays<NT>(npool0, n)) {  // (round 6) the usual round: a handful of new keys -- no sort (the padding of newk is then unused)
            rank_merge_unsorted<NT>(lpool, npool0, newk, n, out, ef, tid);
        } else {
            sort_keys<NT>(newk, Pn, tid);
            rank_merge<NT>(lpool, npool0, newk, n, out, ef, tid);
        }
        fin = out;
    }
    if (tid < 64) {
        const int nsteps = ws.nsteps[q];
        const int found = select_pops(fin, npool1, ws.W, a.check_rel, ws.efs, nsteps, ws.pop + (size_t)q * ws.W, tid);
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

}  // namespace lm
