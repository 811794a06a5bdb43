// lm_pq_impl.h -- DiskANN-style path: PQ-ADC beam search as ONE persistent kernel per batch + a single
// deferred-fetch exact rerank.  Included at the end of lm_search.hip (shares its kernels and lm_index).
//
// Reference surface replaced (fork not in tree): _diskannpy.StaticDiskFloatIndex(...).batch_search(...)
// packages/leann-backend-diskann/leann_backend_diskann/diskann_backend.py:371-380,453-467; strategy
// stated at :444-449 (traversal on PQ distances only, one final rerank via deferred embedding fetch,
// the protobuf NodeEmbeddingRequest of diskann_embedding_server.py:258-334 becomes lm_provider_fn).
//
// MI355X design: a PQ lookup table is m*256 floats (48 KB at m=48).  Re-staging it every lock-step
// round would move more bytes than exact fp32 distances do, so the whole traversal of a query runs
// inside one workgroup that keeps LUT + candidate list in LDS (160 KB/CU) from start to finish:
// no host round trips, no inter-workgroup communication; HBM traffic per evaluation is the m-byte
// code + the neighbour id.  Arithmetic contract: oracle/lm_oracle_pq.c.
#pragma once

namespace lm {

struct PqDev {
    int32_t m;
    const int32_t* chunk_off;  // m + 1: sub-quantiser j covers dimensions [chunk_off[j], chunk_off[j + 1]) (uniform d / m for lm_pq_attach;
                               // the public DiskANN pq_pivots chunking -- unequal, possibly empty chunks -- for lm_pq_attach_chunked)
    const float* codebooks;    // chunk j: 256 centroids x len_j floats at 256 * chunk_off[j]   (uniform: m x 256 x dsub)
    const uint8_t* codes;      // N x m
};

struct PqArgs {
    const float* Q;  // B x Dp
    int32_t Dp, metric, L, W, maxnew, Pmax;
    unsigned long long* n_adc_q;  // per-query ADC evaluations
    int32_t* rounds_q;
    int32_t exp_cap;  // > 0: the rerank set is EVERY expanded node (upstream DiskANN's full_retset), recorded in ws.pool (ws.ef = exp_cap
                      // entries per query); a query that expands more than exp_cap nodes falls back to its final list (counted)
    unsigned long long* n_exp_overflow;
};

// dynamic LDS: lut[m*256] f32 | lpool[L] u64 | out[L] u64 | newk[Pmax] u64 | s_new[maxnew] i32
// NTH = threads per workgroup = per query.  The lookup table alone is m x 1 KB of LDS (96 KB at m = 96): one workgroup per CU, so the
// workgroup's own width is all the latency hiding the CU gets.  Round 2 ran 256 threads (ONE wave per SIMD; 2 % of the HBM rate); with
// 1024 threads (four waves per SIMD) a hop's visited tests, code-row gathers, compactions and the bitonic sort all run four times wider.
#ifndef PQ_RANK_SORT_MAX
#define PQ_RANK_SORT_MAX 512  // survivors of a hop up to which their places in the list are counted instead of sorted (k_pq_traverse)
#endif
template <int NTH>
__global__ __launch_bounds__(NTH) void k_pq_traverse(GraphDev g, PqDev pq, WsDev ws, PqArgs a) {
    constexpr int NWV = NTH / 64;
    extern __shared__ __align__(16) unsigned char smem[];
    __shared__ uint32_t s_off[65];
    __shared__ uint64_t s_b[64];
    __shared__ int32_t s_pop[64];
    __shared__ int s_npop, s_wcnt[NWV];
    float* lut = (float*)smem;
    uint64_t* lpool = (uint64_t*)(lut + pq.m * 256);
    uint64_t* outp = lpool + a.L;
    uint64_t* newk = outp + a.L;
    int32_t* s_new = (int32_t*)(newk + a.Pmax);

    const int q = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const float* qv = a.Q + (size_t)q * a.Dp;
    // ---- lookup table (canonical: sequential fmaf over the sub-vector) ----
    // (round 6) The element-at-a-time loop was compiled to two loads + s_waitcnt vmcnt(0) per element: at m = 96, d = 384 every thread paid 24 entries x 4 elements
    // = 96 dependent round trips before the query's first hop.  Now the loads of FOUR elements (and, for 4-element sub-vectors, of four entries) are requested before
    // the first of them is used; the fmaf chain runs over the elements in the same order: bit-identical table.
    auto lut_value = [&](const float* qs, const float* cb, int len) -> float {
        float acc = 0.0f;
        int t = 0;
        if (a.metric == LM_METRIC_L2) {
            for (; t + 4 <= len; t += 4) {
                const float q0 = qs[t], q1 = qs[t + 1], q2 = qs[t + 2], q3 = qs[t + 3], c0 = cb[t], c1 = cb[t + 1], c2 = cb[t + 2], c3 = cb[t + 3];
                const float d0 = q0 - c0, d1 = q1 - c1, d2 = q2 - c2, d3 = q3 - c3;
                acc = __builtin_fmaf(d0, d0, acc);
                acc = __builtin_fmaf(d1, d1, acc);
                acc = __builtin_fmaf(d2, d2, acc);
                acc = __builtin_fmaf(d3, d3, acc);
            }
            for (; t < len; ++t) {
                float d = qs[t] - cb[t];
                acc = __builtin_fmaf(d, d, acc);
            }
            return acc;
        }
        for (; t + 4 <= len; t += 4) {
            const float q0 = qs[t], q1 = qs[t + 1], q2 = qs[t + 2], q3 = qs[t + 3], c0 = cb[t], c1 = cb[t + 1], c2 = cb[t + 2], c3 = cb[t + 3];
            acc = __builtin_fmaf(q0, c0, acc);
            acc = __builtin_fmaf(q1, c1, acc);
            acc = __builtin_fmaf(q2, c2, acc);
            acc = __builtin_fmaf(q3, c3, acc);
        }
        for (; t < len; ++t) acc = __builtin_fmaf(qs[t], cb[t], acc);
        return -acc;
    };
    {
        const int ne = pq.m * 256;
        constexpr int EU = 4;
        for (int e0 = tid; e0 < ne; e0 += EU * NTH) {
            int lov[EU], lenv[EU];
            bool four = true;
#pragma unroll
            for (int u = 0; u < EU; ++u) {
                const int e = min(e0 + u * NTH, ne - 1);  // (clamped: the tail group recomputes the last entry, stores are guarded)
                lov[u] = pq.chunk_off[e >> 8];
                lenv[u] = pq.chunk_off[(e >> 8) + 1] - lov[u];
                four = four && lenv[u] == 4;
            }
            if (four) {  // the common shape (d = 4 m): sixteen centroid and sixteen query elements in flight per thread
                float qe[EU][4], ce[EU][4];
#pragma unroll
                for (int u = 0; u < EU; ++u) {
                    const int e = min(e0 + u * NTH, ne - 1);
                    const float* cb = pq.codebooks + (size_t)256 * lov[u] + (size_t)(e & 255) * 4;
                    const float* qs = qv + lov[u];
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        qe[u][t] = qs[t];
                        ce[u][t] = cb[t];
                    }
                }
#pragma unroll
                for (int u = 0; u < EU; ++u) {
                    float acc = 0.0f;
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        if (a.metric == LM_METRIC_L2) {
                            const float d = qe[u][t] - ce[u][t];
                            acc = __builtin_fmaf(d, d, acc);
                        } else {
                            acc = __builtin_fmaf(qe[u][t], ce[u][t], acc);
                        }
                    }
                    if (e0 + u * NTH < ne) lut[e0 + u * NTH] = a.metric == LM_METRIC_L2 ? acc : -acc;
                }
            } else {
#pragma unroll
                for (int u = 0; u < EU; ++u) {
                    const int e = e0 + u * NTH;
                    if (e < ne) lut[e] = lut_value(qv + lov[u], pq.codebooks + (size_t)256 * lov[u] + (size_t)(e & 255) * lenv[u], lenv[u]);
                }
            }
        }
    }
    __syncthreads();
    uint32_t* vis = ws.visited + (size_t)q * ws.nw;
    const int mw = pq.m >> 2;  // u32 words per code
    // ADC distance of node v by ONE lane: the canonical arithmetic of oracle/lm_oracle_pq.c:orc_pq_adc -- four partial sums over the
    // sub-quantisers j = r, r + 4, ... (r = 0..3), each in increasing j, combined as (p0 + p1) + (p2 + p3) -- with the code row
    // fetched as 16-byte pieces (m % 16 == 0) or dwords.  Round 2 spread one vector over 4 lanes (every lane re-loading all m / 4
    // dwords): 64 vectors per workgroup pass and 100-byte gathers one dword at a time made the kernel latency bound (2 % of the HBM
    // rate); one lane per vector puts 256 independent gathers in flight per pass and needs no cross-lane reduction.
    auto word = [&](float& p0, float& p1, float& p2, float& p3, int i, uint32_t w) {
        const float* l4 = lut + ((4 * i) << 8);
        p0 = p0 + l4[w & 255u];
        p1 = p1 + l4[256 + ((w >> 8) & 255u)];
        p2 = p2 + l4[512 + ((w >> 16) & 255u)];
        p3 = p3 + l4[768 + (w >> 24)];
    };
    // (round 6) the row's NQ 16-byte pieces are ALL requested before the first lookup, NQ a compile-time constant of the call: the loop form below was compiled to
    // load / s_waitcnt vmcnt(0) / 16 lookups per piece -- six DEPENDENT global round trips per evaluation at m = 96, the longest chain of a hop.  Same additions in the
    // same order: bit-identical distances.
    typedef unsigned pq_u32x4 __attribute__((ext_vector_type(4)));
    auto adc_pieces = [&](auto NQ, int32_t v) -> float {
        constexpr int nq = decltype(NQ)::value;
        const pq_u32x4* c4 = (const pq_u32x4*)(pq.codes + (size_t)v * pq.m);
        pq_u32x4 c[nq];
#pragma unroll
        for (int i = 0; i < nq; ++i) c[i] = c4[i];
        float p0 = 0.0f, p1 = 0.0f, p2 = 0.0f, p3 = 0.0f;
#pragma unroll
        for (int i = 0; i < nq; ++i) {
            word(p0, p1, p2, p3, 4 * i, c[i][0]);
            word(p0, p1, p2, p3, 4 * i + 1, c[i][1]);
            word(p0, p1, p2, p3, 4 * i + 2, c[i][2]);
            word(p0, p1, p2, p3, 4 * i + 3, c[i][3]);
        }
        return (p0 + p1) + (p2 + p3);
    };
    auto adc1 = [&](int32_t v) -> float {
        using std::integral_constant;
        switch (pq.m) {  // (wave uniform)
            case 16: return adc_pieces(integral_constant<int, 1>{}, v);
            case 32: return adc_pieces(integral_constant<int, 2>{}, v);
            case 48: return adc_pieces(integral_constant<int, 3>{}, v);
            case 64: return adc_pieces(integral_constant<int, 4>{}, v);
            case 96: return adc_pieces(integral_constant<int, 6>{}, v);
            case 128: return adc_pieces(integral_constant<int, 8>{}, v);
            default: break;
        }
        const uint32_t* cw = (const uint32_t*)(pq.codes + (size_t)v * pq.m);
        float p0 = 0.0f, p1 = 0.0f, p2 = 0.0f, p3 = 0.0f;
        if ((pq.m & 15) == 0) {
            const uint4* c4 = (const uint4*)cw;
            for (int i = 0; i < (mw >> 2); ++i) {
                const uint4 w4 = c4[i];
                word(p0, p1, p2, p3, 4 * i, w4.x);
                word(p0, p1, p2, p3, 4 * i + 1, w4.y);
                word(p0, p1, p2, p3, 4 * i + 2, w4.z);
                word(p0, p1, p2, p3, 4 * i + 3, w4.w);
            }
        } else {
            for (int i = 0; i < mw; ++i) word(p0, p1, p2, p3, i, cw[i]);
        }
        return (p0 + p1) + (p2 + p3);
    };
    // ---- seed with the entry point (medoid) ----
    int npool = 0;
    if (tid == 0) {
        const int32_t ep = g.entry_point;
        atomicOr(&vis[ep >> 5], 1u << (ep & 31));
        lpool[0] = make_key(adc1(ep), ep);
    }
    npool = 1;
    unsigned long long n_adc = 1;
    int rounds = 0, nexp = 0;
    __syncthreads();

    for (;;) {
        // ---- pops: the W smallest unexpanded (wave 0) ----
        if (tid < 64) {
            // DiskANN's rule: the W closest unexpanded entries of the L-list, until none is left (no step cap)
            const int found = select_pops(lpool, npool, a.W, 1, a.L, 0, s_pop, tid);
            LM_WAVE_SYNC();  // s_pop[r] written by the popping lanes, read below by lane r
            // neighbour ranges of the pops + inclusive scan of their degrees
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
        rounds++;
        if (a.exp_cap > 0 && tid < np && nexp + tid < a.exp_cap)  // the expanded nodes, in expansion order (s_pop is stable until the next pop selection)
            ws.pool[(size_t)q * ws.ef + nexp + tid] = make_key(0.0f, s_pop[tid]);
        nexp += np;
        const uint32_t totalc = s_off[np];
        // ---- flattened expansion over the workgroup, visited test-and-set, ordered compaction ----
        // (round 6) EP passes of the workgroup in flight at a time: their neighbour ids are requested together (clamped indices: unconditional loads), then their
        // test-and-sets (a lane past the end ORs a zero into a word of its own), then the ordered compactions -- one global round trip per EP * NTH
        // neighbours for each of the two, where the pass-at-a-time form paid both per NTH.  Which of two duplicates of a node counts as the fresh one may differ
        // from that form; the SET of fresh nodes, hence every key the hop produces, does not.
        int total = 0;
        constexpr int EP = 4;
        for (uint32_t f0 = 0; f0 < totalc; f0 += EP * NTH) {
            int32_t vv[EP];
            uint32_t bitp[EP], oldw[EP];
#pragma unroll
            for (int p = 0; p < EP; ++p) {
                const uint32_t f = f0 + p * NTH + tid;
                const bool valid = f < totalc;
                const uint32_t fc = valid ? f : totalc - 1;
                int lo = 0, hi = np - 1;
                while (lo < hi) {
                    int mid = (lo + hi + 1) >> 1;
                    if (s_off[mid] <= fc) lo = mid;
                    else hi = mid - 1;
                }
                vv[p] = g.neighbors[s_b[lo] + (fc - s_off[lo])];
                bitp[p] = valid ? 1u : 0u;
            }
#pragma unroll
            for (int p = 0; p < EP; ++p) {  // a lane past the end ORs a zero into a word of its own (tid mod nw): the same word for all of them would serialise in the L2
                const bool valid = bitp[p] != 0u;
                bitp[p] <<= (vv[p] & 31);
                oldw[p] = atomicOr(&vis[valid ? (uint32_t)(vv[p] >> 5) : (uint32_t)tid % (uint32_t)ws.nw], bitp[p]);
            }
#pragma unroll
            for (int p = 0; p < EP; ++p) {
                if (f0 + p * NTH < totalc) {  // (workgroup uniform)
                    const bool fresh = bitp[p] != 0u && !(oldw[p] & bitp[p]);
                    unsigned long long m = __ballot(fresh);
                    if (lane == 0) s_wcnt[wv] = __popcll(m);
                    __syncthreads();
                    int woff = 0;
                    for (int i = 0; i < wv; ++i) woff += s_wcnt[i];
                    if (fresh) s_new[total + woff + __popcll(m & ((1ull << lane) - 1ull))] = vv[p];
#pragma unroll
                    for (int w2 = 0; w2 < NWV; ++w2) total += s_wcnt[w2];
                    __syncthreads();
                }
            }
        }
        const int n = total;
        n_adc += (unsigned long long)n;
        // ---- ADC distances: one lane per fresh node, NTH gathers in flight per pass ----
        // A key that is not below the worst entry of a FULL list can never enter it (keys are unique: (distance, id)): such keys are
        // dropped before the sort -- in steady state most of a hop's candidates -- so the bitonic sort runs over the survivors only.
        const uint64_t thr = npool >= a.L ? lpool[a.L - 1] : KEY_NONE;
        int kept = 0;
        for (int i0 = 0; i0 < n; i0 += NTH) {
            const int i = i0 + tid;
            uint64_t key = KEY_NONE;
            if (i < n) {
                const int32_t v = s_new[i];
                key = make_key(adc1(v), v);
            }
            const bool keep = key < thr;  // KEY_NONE (idle lanes) is never below thr
            unsigned long long mk = __ballot(keep);
            if (lane == 0) s_wcnt[wv] = __popcll(mk);
            __syncthreads();
            int woff = 0;
            for (int w2 = 0; w2 < wv; ++w2) woff += s_wcnt[w2];
            if (keep) newk[kept + woff + __popcll(mk & ((1ull << lane) - 1ull))] = key;
#pragma unroll
            for (int w2 = 0; w2 < NWV; ++w2) kept += s_wcnt[w2];
            __syncthreads();
        }
        if (kept > 0 && kept <= counting_merge_limit(PQ_RANK_SORT_MAX)) {  // (round 6) few survivors -- the steady state: their places by counting, one barrier (lm_beam_common.h)
            rank_merge_unsorted<NTH>(lpool, npool, newk, kept, outp, a.L, tid);
            npool = min(a.L, npool + kept);
            for (int i = tid; i < npool; i += NTH) lpool[i] = outp[i];
            __syncthreads();
        } else if (kept > 0) {
            int Pn = 1;
            while (Pn < kept) Pn <<= 1;
            for (int i = kept + tid; i < Pn; i += NTH) newk[i] = KEY_NONE;
            __syncthreads();
            sort_keys<NTH>(newk, Pn, tid);
            rank_merge<NTH>(lpool, npool, newk, kept, outp, a.L, tid);
            npool = min(a.L, npool + kept);
            for (int i = tid; i < npool; i += NTH) lpool[i] = outp[i];
            __syncthreads();
        }
    }
    // ---- rerank set -> global pool: the final candidate list, or (exp_cap > 0) the expanded nodes already recorded there ----
    uint64_t* pool = ws.pool + (size_t)q * ws.ef;
    const bool expanded_set = a.exp_cap > 0 && nexp <= a.exp_cap;
    if (!expanded_set)
        for (int i = tid; i < npool; i += NTH) pool[i] = lpool[i];
    if (tid == 0) {
        if (a.exp_cap > 0 && !expanded_set) atomicAdd(a.n_exp_overflow, 1ull);
        if (expanded_set) npool = nexp;
        ws.npool[q] = npool;
        ws.nsteps[q] = nexp;
        a.n_adc_q[q] = n_adc;
        a.rounds_q[q] = rounds + 1;
    }
}

// set the dedup bitmap for every candidate of every query
__global__ void k_pq_mark(WsDev ws) {
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= ws.B * ws.ef) return;
    int q = t / ws.ef, i = t % ws.ef;
    if (i < ws.npool[q]) {
        int32_t v = key_id(ws.pool[(size_t)q * ws.ef + i]);
        atomicOr(&ws.rbm[v >> 5], 1u << (v & 31));
    }
}

// exact canonical distances of the candidates, then per-query sort (one workgroup per query)
template <int NCH, bool L2, bool F16>
__global__ __launch_bounds__(256) void k_pq_rerank(WsDev ws, UpdateArgs a) {
    extern __shared__ __align__(16) unsigned char smem[];
    uint64_t* keys = (uint64_t*)smem;  // P2 >= ef
    const int q = blockIdx.x, tid = threadIdx.x;
    const int n = ws.npool[q];
    uint64_t* pool = ws.pool + (size_t)q * ws.ef;
    for (int i = n + tid; i < a.P2; i += 256) keys[i] = KEY_NONE;
    const int lane16 = tid & 15, sg = tid >> 4;
    float4 qv[NCH];
    load_query<NCH>(a.Q, q, lane16, qv);
    for (int i = sg; i < n; i += 32) {
        const int i2 = i + 16;
        const bool has2 = i2 < n;
        int32_t v0 = key_id(pool[i]);
        int32_t v1 = has2 ? key_id(pool[i2]) : v0;
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
            keys[i] = make_key(d0, v0);
            if (has2) keys[i2] = make_key(d1, v1);
        }
    }
    __syncthreads();
    sort_keys<256>(keys, a.P2, tid);
    for (int i = tid; i < n; i += 256) pool[i] = keys[i];
}

__global__ __launch_bounds__(256) void k_pq_stats(WsDev ws, PqArgs a) {
    __shared__ unsigned long long red[2][4];
    __shared__ int redr[4];
    unsigned long long x = 0, y = 0;
    int r = 0;
    for (int q = threadIdx.x; q < ws.B; q += 256) {
        x += a.n_adc_q[q];
        y += (unsigned long long)ws.nsteps[q];
        r = max(r, a.rounds_q[q]);
    }
    for (int m = 32; m >= 1; m >>= 1) {
        x += __shfl_xor(x, m);
        y += __shfl_xor(y, m);
        r = max(r, __shfl_xor(r, m));
    }
    if ((threadIdx.x & 63) == 0) {
        red[0][threadIdx.x >> 6] = x;
        red[1][threadIdx.x >> 6] = y;
        redr[threadIdx.x >> 6] = r;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        ws.counters[C_NDIS] = red[0][0] + red[0][1] + red[0][2] + red[0][3];
        ws.counters[C_NEXPAND] = red[1][0] + red[1][1] + red[1][2] + red[1][3];
        ws.counters[C_ROUNDS] = (unsigned long long)max(max(redr[0], redr[1]), max(redr[2], redr[3]));
    }
}

}  // namespace lm

template <bool L2, bool F16>
static int launch_rerank_nch(lm_index* ix, const UpdateArgs& a) {
    dim3 grid(ix->ws.B), block(256);
    size_t shmem = (size_t)a.P2 * 8;
    switch (ix->Dp / 64) {
#define CASER(n) case n: hipLaunchKernelGGL((k_pq_rerank<n, L2, F16>), grid, block, shmem, ix->stream, ix->ws, a); break
        CASER(1); CASER(2); CASER(3); CASER(4); CASER(5); CASER(6); CASER(8); CASER(12); CASER(16);
#undef CASER
        default: LM_FAIL(LM_EINVAL, "unsupported padded dimension");
    }
    LM_HIP(hipGetLastError());
    return LM_OK;
}

static int pq_search_pass(lm_index* ix, int32_t B, const float* d_q, int32_t k, const lm_pq_search_params& prm,
                          float* d_dist, int64_t* d_labels) {
    const int32_t L = std::max(prm.complexity, k);
    const int32_t W = std::max(prm.beam_width, 1);
    if (W > 64) LM_FAIL(LM_EINVAL, "beam_width > 64 is not supported by the PQ traversal kernel");
    // option "pq_rerank_expanded": the exact rerank ranks every EXPANDED node (upstream DiskANN: full_retset) instead of the final
    // candidate list; the record holds up to 4 L (<= 8192: the rerank kernel's sort) entries per query
    // The record replaces the candidate list as what the traversal leaves in ws.pool (keys in expansion order, distance 0), so it is kept
    // ONLY when an exact rerank follows: with skip_search_reorder, or with neither a table nor a provider, the PQ-ordered list is the result.
    const bool rerank = !prm.skip_search_reorder && ((prm.use_deferred_fetch && ix->provider) || ix->d_table != nullptr);
    const int32_t exp_cap = (ix->pq_rerank_expanded && rerank) ? std::min<int32_t>(8192, 4 * L) : 0;
    if (exp_cap && L > 8192) LM_FAIL(LM_EINVAL, "pq_rerank_expanded: complexity <= 8192");
    int rc = ensure_ws(ix, B, exp_cap ? exp_cap : L, W);
    if (rc) return rc;
    WsDev& ws = ix->ws;
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
    PqDev pq{ix->pq_m, ix->d_pq_chunk_off, ix->d_pq_codebooks, ix->d_pq_codes};
    PqArgs pa{};
    pa.Q = d_q; pa.Dp = ix->Dp; pa.metric = ix->metric; pa.L = L; pa.W = W; pa.maxnew = ws.maxnew;
    pa.Pmax = next_pow2(ws.maxnew);
    pa.n_adc_q = ix->d_pq_nadc; pa.rounds_q = ix->d_pq_rounds;
    pa.exp_cap = exp_cap;
    pa.n_exp_overflow = ws.counters + C_PQ_OVERFLOW;
    size_t shmem = (size_t)ix->pq_m * 256 * 4 + (size_t)2 * L * 8 + (size_t)pa.Pmax * 8 + (size_t)ws.maxnew * 4;
    if (shmem > 158 * 1024)  // 160 KiB per workgroup minus the kernel's ~1.1 KiB of static LDS
        LM_FAIL(LM_EINVAL, "PQ search state does not fit the 160 KB LDS (reduce m, complexity or beam_width)");
    if (ix->pq_threads == 256) LM_HIP(hipFuncSetAttribute((const void*)k_pq_traverse<256>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
    else if (ix->pq_threads == 512) LM_HIP(hipFuncSetAttribute((const void*)k_pq_traverse<512>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
    else LM_HIP(hipFuncSetAttribute((const void*)k_pq_traverse<1024>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
    LM_HIP(hipMemsetAsync(ws.visited, 0, (size_t)B * ws.nw * 4, st));
    LM_HIP(hipMemsetAsync(ws.counters, 0, C_NCOUNTERS * sizeof(unsigned long long), st));
    {
        EvScope es(ix, &ix->ev_update);
        // option "pq_threads" (256 / 512 / 1024; default 1024): workgroup width of the traversal (A/B; identical results)
        if (ix->pq_threads == 256) hipLaunchKernelGGL(k_pq_traverse<256>, dim3(B), dim3(256), shmem, st, g, pq, ws, pa);
        else if (ix->pq_threads == 512) hipLaunchKernelGGL(k_pq_traverse<512>, dim3(B), dim3(512), shmem, st, g, pq, ws, pa);
        else hipLaunchKernelGGL(k_pq_traverse<1024>, dim3(B), dim3(1024), shmem, st, g, pq, ws, pa);
    }
    LM_HIP(hipGetLastError());
    ix->stats.update_launches++;
    hipLaunchKernelGGL(k_pq_stats, dim3(1), dim3(256), 0, st, ws, pa);
    unsigned long long* hc = ix->h_counters;
    if (rerank) {
        UpdateArgs ua{};
        ua.Q = d_q;
        ua.P2 = next_pow2(exp_cap ? exp_cap : L);
        if ((size_t)ua.P2 * 8 > 64 * 1024) LM_FAIL(LM_EINVAL, "complexity too large for the rerank kernel (<= 8192 candidates per query)");
        if (prm.use_deferred_fetch && ix->provider) {
            // ONE deferred fetch for the union of all candidate lists
            const int ntiles = (int)((ws.nw + UNIQ_TILE - 1) / UNIQ_TILE);
            hipLaunchKernelGGL(k_pq_mark, dim3((unsigned)(((int64_t)B * ws.ef + 255) / 256)), dim3(256), 0, st, ws);
            hipLaunchKernelGGL(k_uniq_count, dim3(ntiles), dim3(256), 0, st, ws);
            hipLaunchKernelGGL(k_uniq_emit, dim3(ntiles), dim3(256), 0, st, ws, ntiles);
            LM_HIP(hipMemcpyAsync(hc, ws.counters, C_NCOUNTERS * sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
            LM_HIP(hipStreamSynchronize(st));
            int32_t nu = (int32_t)hc[C_NUNIQ];
            void* d_e = nullptr;
            ix->stats.nunique += nu;
            if (nu > 0) {
                EvScope es(ix, &ix->ev_provider);
                int prc = ix->provider(ix->provider_user, ws.uniq, nu, &d_e, (void*)st);
                if (prc != 0 || !d_e) LM_FAIL(LM_EPROVIDER, "embedding provider failed (rc=" + std::to_string(prc) + ")");
            }
            ua.E = d_e;
            ua.by_rank = 1;
            rc = ix->metric == LM_METRIC_L2 ? launch_rerank_nch<true, false>(ix, ua) : launch_rerank_nch<false, false>(ix, ua);
        } else {
            ua.E = ix->d_table;
            ua.by_rank = 0;
            const bool f16 = ix->table_dtype == LM_DTYPE_F16, l2 = ix->metric == LM_METRIC_L2;
            rc = l2 ? (f16 ? launch_rerank_nch<true, true>(ix, ua) : launch_rerank_nch<true, false>(ix, ua))
                    : (f16 ? launch_rerank_nch<false, true>(ix, ua) : launch_rerank_nch<false, false>(ix, ua));
        }
        if (rc) return rc;
    }
    hipLaunchKernelGGL(k_finalize, dim3((B * k + 255) / 256), dim3(256), 0, st, ws, k, ix->metric, d_labels, d_dist);
    LM_HIP(hipGetLastError());
    LM_HIP(hipMemcpyAsync(hc, ws.counters, C_NCOUNTERS * sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
    LM_HIP(hipStreamSynchronize(st));
    ix->stats.ndis += (int64_t)hc[C_NDIS];
    ix->stats.nexpand += (int64_t)hc[C_NEXPAND];
    ix->stats.nrounds = std::max<int64_t>(ix->stats.nrounds, (int64_t)hc[C_ROUNDS]);
    ix->pq_overflow += (int64_t)hc[C_PQ_OVERFLOW];
    return LM_OK;
}

extern "C" {

int lm_pq_attach_chunked(lm_index* ix, int32_t m, const int32_t* chunk_offsets, const float* codebooks, const uint8_t* codes, int64_t ntotal) {
    if (!ix || !chunk_offsets || !codebooks || !codes) LM_FAIL(LM_EINVAL, "NULL argument");
    if (ntotal != ix->N) LM_FAIL(LM_EINVAL, "code count does not match the index");
    if (m <= 0 || m % 4 || m > 4096) LM_FAIL(LM_EINVAL, "m must be a positive multiple of 4 (pad the codes with empty chunks)");
    if (chunk_offsets[0] != 0) LM_FAIL(LM_EINVAL, "chunk_offsets[0] must be 0");
    for (int j = 0; j < m; ++j)
        if (chunk_offsets[j + 1] < chunk_offsets[j]) LM_FAIL(LM_EINVAL, "chunk_offsets must not decrease");
    if (chunk_offsets[m] > ix->D) LM_FAIL(LM_EINVAL, "chunk_offsets[m] exceeds the index dimension");
    LM_HIP(hipSetDevice(ix->device));
    if (ix->d_pq_codebooks) (void)hipFree(ix->d_pq_codebooks);
    if (ix->d_pq_codes) (void)hipFree(ix->d_pq_codes);
    if (ix->d_pq_chunk_off) (void)hipFree(ix->d_pq_chunk_off);
    ix->d_pq_codebooks = nullptr;
    ix->d_pq_codes = nullptr;
    ix->d_pq_chunk_off = nullptr;
    const size_t cb_bytes = (size_t)256 * chunk_offsets[m] * 4, code_bytes = (size_t)ntotal * m;
    LM_HIP(hipMalloc((void**)&ix->d_pq_codebooks, std::max<size_t>(cb_bytes, 16)));
    LM_HIP(hipMalloc((void**)&ix->d_pq_codes, std::max<size_t>(code_bytes, 16)));
    LM_HIP(hipMalloc((void**)&ix->d_pq_chunk_off, (size_t)(m + 1) * 4));
    LM_HIP(hipMemcpy(ix->d_pq_codebooks, codebooks, cb_bytes, hipMemcpyHostToDevice));
    LM_HIP(hipMemcpy(ix->d_pq_codes, codes, code_bytes, hipMemcpyHostToDevice));
    LM_HIP(hipMemcpy(ix->d_pq_chunk_off, chunk_offsets, (size_t)(m + 1) * 4, hipMemcpyHostToDevice));
    ix->pq_m = m;
    return LM_OK;
}

int lm_pq_attach(lm_index* ix, int32_t m, const float* codebooks, const uint8_t* codes, int64_t ntotal) {
    if (!ix) LM_FAIL(LM_EINVAL, "NULL argument");
    if (m <= 0 || m % 4 || ix->D % m) LM_FAIL(LM_EINVAL, "m must be a multiple of 4 that divides d");
    std::vector<int32_t> off((size_t)m + 1);
    for (int j = 0; j <= m; ++j) off[j] = j * (ix->D / m);  // uniform chunks: the same code path, the same arithmetic
    return lm_pq_attach_chunked(ix, m, off.data(), codebooks, codes, ntotal);
}

void lm_pq_search_params_default(lm_pq_search_params* p) {
    if (!p) return;
    std::memset(p, 0, sizeof(*p));
    p->complexity = 64;
    p->beam_width = 1;
    p->use_global_pruning = 1;
    p->num_threads = 8;
}

static int pq_search_device(lm_index* ix, int64_t n, const float* d_x, int32_t k, const lm_pq_search_params* params,
                            int64_t* d_labels, float* d_dist) {
    if (!ix || !params || n < 0 || k <= 0) LM_FAIL(LM_EINVAL, "bad search arguments");
    if (params->complexity <= 0) LM_FAIL(LM_EINVAL, "complexity must be positive");
    if (params->recompute_neighbors)
        LM_FAIL(LM_EINVAL, "recompute_neighbors != 0 is not supported: the traversal runs on PQ distances only, as the reference runs it (diskann_backend.py:444-451)");
    if (!ix->d_pq_codes) LM_FAIL(LM_ESTATE, "no PQ codes attached (lm_pq_attach)");
    if (params->use_deferred_fetch && !ix->provider && !ix->d_table)
        LM_FAIL(LM_ESTATE, "deferred fetch requested but neither an embedding provider nor stored embeddings are attached");
    LM_HIP(hipSetDevice(ix->device));
    ix->stats = lm_search_stats{};
    if (n == 0) return LM_OK;
    hipStream_t st = ix->stream;
    if (ix->N == 0 || ix->entry_point < 0) {
        hipLaunchKernelGGL(k_fill_empty, dim3((unsigned)((n * k + 255) / 256)), dim3(256), 0, st, n * (int64_t)k, ix->metric,
                           d_labels, d_dist);
        LM_HIP(hipStreamSynchronize(st));
        return LM_OK;
    }
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
    int64_t maxb = 4096;
    int64_t nwbytes = ((ix->N + 31) / 32) * 4;
    maxb = std::max<int64_t>(1, std::min<int64_t>(maxb, (8ll << 30) / std::max<int64_t>(nwbytes, 1)));
    for (int64_t off = 0; off < n; off += maxb) {
        int32_t B = (int32_t)std::min<int64_t>(maxb, n - off);
        int rc = pq_search_pass(ix, B, d_q + (size_t)off * ix->Dp, k, *params, d_dist + (size_t)off * k, d_labels + (size_t)off * k);
        if (rc) return rc;
    }
    LM_HIP(hipStreamSynchronize(st));
    if (ix->profiling) {
        ix->stats.update_ms = drain_events(ix, ix->ev_update);
        ix->stats.provider_ms = drain_events(ix, ix->ev_provider);
    }
    return LM_OK;
}

int lm_pq_batch_search_device(lm_index* ix, int64_t n, const float* d_x, int32_t k, const lm_pq_search_params* params,
                              int64_t* d_labels, float* d_distances) {
    if (n > 0 && (!d_x || !d_labels || !d_distances)) LM_FAIL(LM_EINVAL, "NULL buffer");
    return pq_search_device(ix, n, d_x, k, params, d_labels, d_distances);
}

int lm_pq_batch_search(lm_index* ix, int64_t n, const float* x, int32_t k, const lm_pq_search_params* params, int64_t* labels,
                       float* distances) {
    if (!ix) LM_FAIL(LM_EINVAL, "NULL index");
    if (n < 0 || k <= 0) LM_FAIL(LM_EINVAL, "bad n / k");
    if (n == 0) return LM_OK;
    if (!x || !labels || !distances) LM_FAIL(LM_EINVAL, "NULL buffer");
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
    if (!rc) rc = pq_search_device(ix, n, d_x, k, params, d_l, d_d);
    if (!rc && (hipMemcpyAsync(distances, d_d, need_d, hipMemcpyDeviceToHost, ix->stream) != hipSuccess ||
                hipMemcpyAsync(labels, d_l, need_l, hipMemcpyDeviceToHost, ix->stream) != hipSuccess ||
                hipStreamSynchronize(ix->stream) != hipSuccess)) {
        set_error("result copy failed");
        rc = LM_EHIP;
    }
    return rc;
}

}  // extern "C"
