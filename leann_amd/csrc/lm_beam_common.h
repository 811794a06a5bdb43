// lm_beam_common.h -- the pieces of one beam-search step that every traversal kernel shares (k_update, the persistent
// k_search_table / k_search_table_wave, k_pq_traverse): canonical row distance, sort of the fresh keys, rank merge into
// the sorted pool, selection of the next pops (both faiss stop rules), greedy-descent transition.  One definition each:
// the lock-step and the persistent kernels cannot drift apart (round 1 carried four copies of these blocks).
// Part of lm_search.hip's translation unit (included there, in this order); see its header comment.
#pragma once

namespace lm {

// ---- canonical distance: 16 lanes per row, lane t owns float4 chunks t, t+16, ... (== oracle/lm_oracle.c:orc_dist) ----
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

template <int NCH>
__device__ __forceinline__ void load_query(const float* Q, int q, int lane16, float4 (&qv)[NCH]) {
    const float4* qrow = (const float4*)(Q + (size_t)q * (NCH * 64));
#pragma unroll
    for (int i = 0; i < NCH; ++i) qv[i] = qrow[lane16 + 16 * i];
}

// ---- bitonic sort (ascending) of newk[0 .. Pn), Pn a power of two, by NT cooperating threads; ends with a barrier ----
template <int NT>
__device__ __forceinline__ void sort_keys(uint64_t* newk, int Pn, int tid) {
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
}

// ---- merge by rank: out[0 .. min(ef, npool0 + n)) = the smallest keys of lpool[0..npool0) U newk[0..n), both sorted.
//      (dist,id) pairs are unique across pool U new: compare without the flag bit.  Ends with a barrier. ----
template <int NT>
__device__ __forceinline__ void rank_merge(const uint64_t* lpool, int npool0, const uint64_t* newk, int n, uint64_t* out, int ef,
                                           int tid) {
    for (int i = tid; i < npool0; i += NT) {
        const uint64_t key = lpool[i];
        const uint64_t kk = key >> 1;
        int lo = 0, hi = n;
        while (lo < hi) {
            int mid = (lo + hi) >> 1;
            if ((newk[mid] >> 1) < kk) lo = mid + 1;
            else hi = mid;
        }
        if (i + lo < ef) out[i + lo] = key;
    }
    for (int j = tid; j < n; j += NT) {
        const uint64_t key = newk[j];
        const uint64_t kk = key >> 1;
        int lo = 0, hi = npool0;
        while (lo < hi) {
            int mid = (lo + hi) >> 1;
            if ((lpool[mid] >> 1) < kk) lo = mid + 1;
            else hi = mid;
        }
        if (j + lo < ef) out[j + lo] = key;
    }
    __syncthreads();
}

// ---- the same merge WITHOUT sorting newk first: a key's place in the merged list = (new keys below it) + (pool entries below it); the first count by
//      scanning all n new keys (every lane of a wave reads the same LDS word: a broadcast, no bank conflict), the second by binary search in the sorted pool
//      (new key) or its own index (pool entry).  One barrier instead of the bitonic sort's log2(P) (log2(P) + 1) / 2: for a few hundred survivors per hop
//      the sort's barriers were most of a hop of k_pq_traverse.  Same output as sort_keys + rank_merge (keys are unique).  Ends with a barrier. ----
// The limit up to which keys are placed by counting.  On the device: the caller's constant.  In the host emulation (tests/hip_emul) LM_EMUL_COUNTING_MERGE_LIMIT
// overrides it, so that the CPU suite walks BOTH branches of every caller (the small graphs of the emulated cases would never reach the bitonic sort otherwise:
// tests/test_emulated_search.py::test_sort_and_counting_merge_agree).
__device__ __forceinline__ int counting_merge_limit(int dflt) {
#ifdef LM_EMULATED_DEVICE
    static const int v = [] {
        const char* e = getenv("LM_EMUL_COUNTING_MERGE_LIMIT");
        return e ? atoi(e) : -1;
    }();
    return v >= 0 ? v : dflt;
#else
    return dflt;
#endif
}
// when counting beats sorting: every thread's scan is n broadcast reads per item it owns; the bitonic sort is log2(P)(log2(P) + 1) / 2 barrier-separated steps
template <int NT>
__device__ __forceinline__ bool rank_merge_unsorted_pays(int npool0, int n) {
    return n * ((n + npool0 + NT - 1) / NT) <= counting_merge_limit(256);
}
template <int NT>
__device__ __forceinline__ void rank_merge_unsorted(const uint64_t* lpool, int npool0, const uint64_t* newk, int n, uint64_t* out, int ef, int tid) {
    for (int it = tid; it < n + npool0; it += NT) {
        const bool is_new = it < n;
        const uint64_t key = is_new ? newk[it] : lpool[it - n];
        const uint64_t kk = key >> 1;
        int below = 0;
        for (int x = 0; x < n; ++x) below += (newk[x] >> 1) < kk ? 1 : 0;
        int lo = it - n;  // a pool entry's own index
        if (is_new) {
            lo = 0;
            int hi = npool0;
            while (lo < hi) {
                int mid = (lo + hi) >> 1;
                if ((lpool[mid] >> 1) < kk) lo = mid + 1;
                else hi = mid;
            }
        }
        if (below + lo < ef) out[below + lo] = key;
    }
    __syncthreads();
}

// ---- next pops: the W smallest unexpanded pool entries, by ONE wave (lane = 0..63; every lane of the wave calls it and gets
//      the same count back).  Both faiss stop rules (search_from_candidates; pinned by oracle/lm_oracle_faiss.c):
//        check_relative_distance: v0 = pop_min(); if (count_below(d0) >= efSearch) break;   -- count_below(d0) is the
//            sorted-pool index of v0, so entries at index >= efs are never expanded (bites only when k > efSearch:
//            the pool holds max(efSearch, k) entries);
//        otherwise:               nstep++; if (nstep > efSearch) break;                      -- at most efs + 1 pops.
//      Marks the chosen keys expanded and stores their node ids to pop_dst[0 .. found). ----
__device__ __forceinline__ int select_pops(uint64_t* fin, int npool, int W, int check_rel, int efs, int nsteps, int32_t* pop_dst,
                                           int lane) {
    const int scan_n = check_rel ? min(npool, efs) : npool;
    int allowed = W;
    if (!check_rel) allowed = min(allowed, max(0, efs + 1 - nsteps));
    int found = 0;
    for (int base = 0; base < scan_n && found < allowed; base += 64) {
        const int i = base + lane;
        const bool un = i < scan_n && !(fin[i] & KEY_EXPANDED);
        const unsigned long long m = __ballot(un);
        const int r = found + __popcll(m & ((1ull << lane) - 1ull));
        if (un && r < allowed) {
            fin[i] |= KEY_EXPANDED;
            pop_dst[r] = key_id(fin[i]);
        }
        found += __popcll(m);
    }
    return min(found, allowed);
}

// ---- greedy descent (faiss greedy_update_nearest), one step: `best` = smallest key among the neighbours just evaluated
//      (KEY_NONE: none).  SEED: the entry point becomes the current node.  UPPER: move if strictly better, else one level
//      down.  Returns true when level 0 is reached (cur seeds the pool: HNSW::search candidates.push(nearest)). ----
__device__ __forceinline__ bool descent_step(int ph, uint64_t best, int max_level, uint64_t& cur, int& level) {
    if (ph == PH_SEED) {
        cur = best;
        level = max_level;
    } else {
        if (best != KEY_NONE && best < cur) cur = best;
        else level--;
    }
    return level <= 0;
}

}  // namespace lm
