// lm_kernels_expand.h -- k_init, k_expand (flattened CSR gather + visited test-and-set), k_uniq_count/emit (sorted unique list).
// Part of lm_search.hip's translation unit (included there, in this order); see its header comment.
#pragma once

namespace lm {

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
__global__ __launch_bounds__(64) void k_expand(GraphDev g, WsDev ws, int use_rbm, int round_no, int defer, int single) {
    // defer != 0: the two-level pruning kernel (k_prune) finishes the new-list: it marks the dedup bitmap and counts
    // single != 0 (a ONE-query pass, option "single_query_direct"): the new-list is the provider's id list as it is -- no request bitmap, no
    // k_uniq_* launches; this kernel writes the list's length and the live flag itself (its grid is one workgroup)
    extern __shared__ int32_t s_new[];
    __shared__ uint32_t s_off[65];
    __shared__ uint64_t s_b[64];
    const int q = blockIdx.x;
    const int lane = threadIdx.x;
    const int ph = ws.phase[q];
    if (ph == PH_DONE) {
        if (lane == 0) {
            ws.nnew[q] = 0;
            if (single) {
                ws.counters[C_LIVE] = 0ull;
                ws.counters[C_NUNIQ] = 0ull;
            }
        }
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
        // ---- dynamic batching (SearchParametersHNSW.batch_size, hnsw_backend.py:163,181,234; LEANN paper section 4.2: "collect the closest
        //      candidates from the queue until a target batch size is reached"): while this round's new-list is shorter than ws.batch, pop the
        //      best unexpanded pool entry -- one at a time, under select_pops' stop rules, on the pool as k_update left it -- and gather its
        //      unvisited neighbours behind the others.  Word for word oracle/lm_oracle.c ("batching"); ws.batch == 0: nothing happens here.
        //      A one-query search is a chain of ~75 rounds of ~10 chunks each on a 256-CU chip; with batch 64 it is ~20 rounds of ~64. ----
        if (ws.batch > 0 && total < ws.batch) {
            uint64_t* pool = ws.pool + (size_t)q * ws.ef;
            const int npool = ws.npool[q];
            const int scan_n = ws.check_rel ? min(npool, ws.efs) : npool;
            int nsteps = ws.nsteps[q];
            int cursor = 0;  // entries below it are expanded (the pool is scanned in order, and what this loop pops it never looks at again)
            while (total < ws.batch) {
                if (!ws.check_rel && nsteps > ws.efs) break;
                int idx = -1;
                uint64_t key = 0;
                for (int base = cursor & ~63; base < scan_n; base += 64) {
                    const int i = base + lane;
                    const uint64_t mine = i < scan_n ? pool[i] : KEY_NONE;
                    const bool un = i >= cursor && i < scan_n && !(mine & KEY_EXPANDED);
                    const unsigned long long m = __ballot(un);
                    if (m) {
                        const int src = __ffsll((long long)m) - 1;
                        idx = base + src;
                        if (lane == src) pool[i] = mine | KEY_EXPANDED;  // the lane that read the entry marks it: no lane reads another lane's write
                        key = (uint64_t)__shfl((unsigned long long)mine, src);
                        break;
                    }
                }
                if (idx < 0) break;
                cursor = idx + 1;
                ++nsteps;
                const L0Range r = g.l0[key_id(key)];
                for (uint32_t j0 = 0; j0 < r.count; j0 += 64) {
                    const uint32_t j = j0 + lane;
                    bool fresh = false;
                    int32_t v = -1;
                    if (j < r.count) {
                        v = g.neighbors[r.begin + j];
                        const uint32_t bit = 1u << (v & 31);
                        const uint32_t old = atomicOr(&vis[v >> 5], bit);
                        fresh = !(old & bit);
                    }
                    const unsigned long long m = __ballot(fresh);
                    if (fresh) s_new[total + __popcll(m & ((1ull << lane) - 1ull))] = v;
                    total += __popcll(m);
                }
            }
            if (lane == 0) ws.nsteps[q] = nsteps;
        }
    }
    __syncthreads();
    int32_t* newid = ws.newid + (size_t)q * ws.maxnew;
    for (int i = lane; i < total; i += 64) {
        const int32_t v = s_new[i];
        newid[i] = v;
        if (!defer && (use_rbm == 1 || (use_rbm == 2 && ws.memo_slot[v] < 0))) atomicOr(&ws.rbm[v >> 5], 1u << (v & 31));
    }
    if (lane == 0) {
        ws.nnew[q] = total;
        if (!defer) ws.ndis_q[q] += (unsigned long long)total;
        // plain stores of identical values (benign): a contended same-address atomic costs ~12 ns per
        // workgroup and serialises the launch tail (MI355X_MICROARCH.md, price list row "fanin")
        ws.counters[C_LIVE] = 1ull;
        ws.counters[C_ROUNDS] = (unsigned long long)round_no;
        if (single) ws.counters[C_NUNIQ] = (unsigned long long)total;
    }
}

// Speculative prefetch for small batches (option "speculate" = S; needs the per-call memo).  A one-query search is a chain of ~100 rounds
// of ~50 dependent kernel launches each (the recompute forward of ~8 chunks): launch latency, not arithmetic.  The next nodes the search
// will pop are, most of the time, the best candidates of the pool that are not expanded yet -- so this round's forward also embeds THEIR
// unvisited neighbours (the request bitmap gets their bits too), the rows go into the memo, and a later round whose new nodes are all in
// the memo needs no forward at all.  Nothing the search itself reads is touched: visited bits, new-lists, pops, the pool and the
// distance-evaluation counts are exactly those of S = 0, and so are the results; only the provider's request lists differ (more rows in
// fewer calls).  One wave per query; after k_expand of the same round (its visited bits are final), before k_uniq_count.
__global__ __launch_bounds__(64) void k_speculate(GraphDev g, WsDev ws, int S, int check_rel) {
    const int q = blockIdx.x, lane = threadIdx.x;
    if (ws.phase[q] != PH_BEAM) return;
    const uint64_t* pool = ws.pool + (size_t)q * ws.ef;
    const uint32_t* vis = ws.visited + (size_t)q * ws.nw;
    const int npool = ws.npool[q];
    const int scan_n = check_rel ? min(npool, ws.efs) : npool;  // what select_pops will look at
    int found = 0;
    for (int base = 0; base < scan_n && found < S; base += 64) {
        const int i = base + lane;
        const uint64_t key = i < scan_n ? pool[i] : KEY_NONE;
        const bool un = i < scan_n && !(key & KEY_EXPANDED);
        unsigned long long m = __ballot(un);
        const int32_t mine = key_id(key);
        while (m && found < S) {
            const int src = __ffsll((long long)m) - 1;
            m &= m - 1;
            const int32_t c = __shfl(mine, src);
            const L0Range r = g.l0[c];
            for (uint32_t j = lane; j < r.count; j += 64) {
                const int32_t v = g.neighbors[r.begin + j];
                const uint32_t bit = 1u << (v & 31);
                if (!(vis[v >> 5] & bit) && ws.memo_slot[v] < 0) atomicOr(&ws.rbm[v >> 5], bit);
            }
            ++found;
        }
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



}  // namespace lm
