// lm_kernels_prune.h -- two-level search: k_pq_lut_all, k_prune.
// Part of lm_search.hip's translation unit (included there, in this order); see its header comment.
#pragma once

namespace lm {

// ---- two-level search (paper Alg. 2; prune_ratio / pruning_strategy of hnsw_backend.py:219-231) --------
// Per-query lookup tables for the whole batch: lut[q][j][c]  (canonical: oracle/lm_oracle_pq.c orc_pq_lut)
struct PruneArgs {
    const float* Q;      // B x Dp
    float* lut;          // B x m x 256
    const float* codebooks;
    const uint8_t* codes;
    int32_t Dp, metric, m;
    const int32_t* chunk_off;  // m + 1 (see PqDev)
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
        const int j = e >> 8, lo = a.chunk_off[j], len = a.chunk_off[j + 1] - lo;
        const float* cb = a.codebooks + (size_t)256 * lo + (size_t)(e & 255) * len;
        const float* qs = qv + lo;
        float acc = 0.0f;
        if (a.metric == LM_METRIC_L2) {
            for (int t = 0; t < len; ++t) {
                float d = qs[t] - cb[t];
                acc = __builtin_fmaf(d, d, acc);
            }
        } else {
            for (int t = 0; t < len; ++t) acc = __builtin_fmaf(qs[t], cb[t], acc);
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
    sort_keys<256>(nk, Pn, tid);
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


}  // namespace lm
