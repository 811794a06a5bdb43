// lm_tokens.hip -- HBM-resident pre-tokenised passage store + gather kernel.
// Replaces, at query time, PassageManager.get_passage (leann/api.py:203-215: open+seek+readline+
// json.loads per id) and the tokeniser call inside compute_embeddings
// (leann/embedding_compute.py:229-239): passages are tokenised once at load and kept packed
// (u16 ids, u64 offsets) in HBM; a round's unique node ids are turned into a padded
// [n][T] int32 batch by one coalesced kernel.
#include "lm_internal.h"

using namespace lm;

namespace lm {
// one wave per chunk: coalesced u16 reads, int32 writes (pad beyond the length)
__global__ __launch_bounds__(256) void k_tokens_gather(const uint16_t* tok, const uint64_t* off, const int32_t* ids,
                                                       int32_t n, int32_t T, int32_t pad_id, int32_t* out, int32_t* out_len) {
    const int w = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (w >= n) return;
    const int32_t id = ids[w];
    const uint64_t b = off[id];
    int32_t len = (int32_t)min((uint64_t)T, off[id + 1] - b);
    int32_t* row = out + (size_t)w * T;
    for (int j = lane; j < T; j += 64) row[j] = j < len ? (int32_t)tok[b + j] : pad_id;
    if (lane == 0) out_len[w] = len;
}
}  // namespace lm

extern "C" {

int lm_tokens_create(const uint16_t* tokens, const uint64_t* offsets, int64_t n, int device, lm_tokens** out) {
    if (!out) LM_FAIL(LM_EINVAL, "out is NULL");
    *out = nullptr;
    if (n < 0 || !offsets || (n > 0 && offsets[n] > 0 && !tokens)) LM_FAIL(LM_EINVAL, "bad token store arguments");
    for (int64_t i = 0; i < n; ++i)
        if (offsets[i + 1] < offsets[i]) LM_FAIL(LM_EINVAL, "token offsets not monotone");
    if (lm_device_count() <= 0) LM_FAIL(LM_EHIP, "no HIP device visible");
    LM_HIP(hipSetDevice(device));
    lm_tokens* t = new lm_tokens();
    t->device = device;
    t->n = n;
    t->total = offsets[n];
    hipError_t e;
    if ((e = hipMalloc((void**)&t->d_tok, std::max<size_t>(t->total * 2, 16))) != hipSuccess ||
        (e = hipMalloc((void**)&t->d_off, (size_t)(n + 1) * 8)) != hipSuccess ||
        (t->total && (e = hipMemcpy(t->d_tok, tokens, t->total * 2, hipMemcpyHostToDevice)) != hipSuccess) ||
        (e = hipMemcpy(t->d_off, offsets, (size_t)(n + 1) * 8, hipMemcpyHostToDevice)) != hipSuccess) {
        set_error(std::string("token store upload failed: ") + hipGetErrorString(e));
        lm_tokens_free(t);
        return LM_EHIP;
    }
    *out = t;
    return LM_OK;
}

void lm_tokens_free(lm_tokens* t) {
    if (!t) return;
    (void)hipSetDevice(t->device);
    if (t->d_tok) (void)hipFree(t->d_tok);
    if (t->d_off) (void)hipFree(t->d_off);
    delete t;
}

int64_t lm_tokens_count(const lm_tokens* t) { return t ? t->n : 0; }

int lm_tokens_gather(const lm_tokens* t, const int32_t* d_ids, int32_t n, int32_t T, int32_t pad_id, int32_t* d_out_ids,
                     int32_t* d_out_len, void* stream) {
    if (!t) LM_FAIL(LM_EINVAL, "NULL token store");
    if (n == 0) return LM_OK;
    if (n < 0 || T <= 0 || !d_ids || !d_out_ids || !d_out_len) LM_FAIL(LM_EINVAL, "bad gather arguments");
    hipLaunchKernelGGL(k_tokens_gather, dim3((n + 3) / 4), dim3(256), 0, (hipStream_t)stream, t->d_tok, t->d_off, d_ids, n, T,
                       pad_id, d_out_ids, d_out_len);
    LM_HIP(hipGetLastError());
    return LM_OK;
}

}  // extern "C"
