// lm_kernels_update.h -- canonical distance primitives, fused update kernel k_update (+ A/B variants k_update_sort, k_dist_flat/k_merge), k_span.
// Part of lm_search.hip's translation unit (included there, in this order); see its header comment.
#pragma once

namespace lm {

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



}  // namespace lm
