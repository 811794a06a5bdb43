// lm_kernels_misc.h -- k_finalize, memo append/release, k_stats, padding, k_dist_pairs, k_topk_merge.
// Part of lm_search.hip's translation unit (included there, in this order); see its header comment.
#pragma once

namespace lm {

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
