// kbench.cpp -- stand-alone micro-benchmark + reference check of the hand-written encoder kernels through the C ABI
// (no Python, no torch: a run costs seconds of GPU time, not the minute a torch import takes on a fresh box).
//
//   hipcc --offload-arch=gfx950 -O2 -std=c++17 scripts/kbench.cpp -Iinclude -Lleann_amd/lib -lleann_mi355x -lrocblas
//         -Wl,-rpath,$PWD/leann_amd/lib -o gpurun_out/kbench            (scripts/build_kbench.sh)
//   ./kbench [tokens=262144] [reps=20] [what=all|bw|linear|qkv|fusedqa|wsgemm|tail|tail4|attn|a3stamps|attn64|ln|gemmf16|gemmstamp]
//
// Every kernel is checked against a plain fp32 GPU reference of the same op on the first and last 192 tokens (incl. the
// ragged tail: tokens is deliberately not a multiple of 128) and timed with HIP events on the launch stream.  One JSON
// object per line.  rocBLAS fp16 GEMMs of the same shapes are timed next to them (the library number to beat).
#include <hip/hip_fp16.h>
#include <hip/hip_runtime.h>
#include <rocblas/rocblas.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <random>
#include <string>
#include <vector>

#include "leann_mi355x.h"
extern "C" int lm_qa_stamps_read(unsigned long long* out, int64_t max_words, int reset);       // diagnosis library only (csrc/lm_qkv_attn_h384.hip)
extern "C" int lm_attn_v3_stamps_read(unsigned long long* out, int64_t max_words, int reset);  // diagnosis library only (csrc/lm_attn_v3.hip)

// generation 3 of the fused layer tail: only in the diagnosis build of the library (csrc/diag/lm_mlp_fused_v3.hip), as the `tail` / `tail4`
// modes' A/B reference; operands: W_o as [12][384][32] slabs, W1 with its columns in accumulator order, W2 as [ffn/32][384][32] slabs
extern "C" int lm_attn_out_mlp_fused_h384_f16(const void* d_attn, const void* d_resid, const void* d_wo_p, const float* d_bo, const void* d_gamma1,
                                              const void* d_beta1, float eps1, const void* d_w1acc, const float* d_b1, const void* d_w2p,
                                              const float* d_b2, const void* d_gamma, const void* d_beta, void* d_out, int64_t tokens, int32_t ffn,
                                              float eps, void* stream);

// internal (C++-linkage) entry points of the library: the head-major QKV projection + generation-3 attention pair of round 5's large forwards
int lm_qkv_h384_launch(const void* d_x, const void* d_w_img, const float* d_bias, int32_t n_out, void* d_out, int64_t tokens, int32_t head_major, void* stream);
int lm_attn_v3_launch_hd32(const void* d_qkv, const int32_t* d_cu_seqlens, int32_t n_seqs, int32_t heads, int32_t max_len, void* d_out, int64_t total_tokens,
                           void* stream);

#define CK(e)                                                                                  \
    do {                                                                                       \
        hipError_t _e = (e);                                                                   \
        if (_e != hipSuccess) {                                                                \
            fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #e, hipGetErrorString(_e)); \
            exit(2);                                                                           \
        }                                                                                      \
    } while (0)
#define LM(e)                                                                         \
    do {                                                                              \
        int _r = (e);                                                                 \
        if (_r != 0) {                                                                \
            fprintf(stderr, "%s:%d %s -> %d: %s\n", __FILE__, __LINE__, #e, _r, lm_last_error()); \
            exit(3);                                                                  \
        }                                                                             \
    } while (0)

static constexpr int H = 384;

template <class T>
struct Dev {
    T* p = nullptr;
    size_t n = 0;
    explicit Dev(size_t n_) : n(n_) { CK(hipMalloc((void**)&p, std::max<size_t>(n * sizeof(T), 16))); }
    Dev(const std::vector<T>& h) : Dev(h.size()) { CK(hipMemcpy(p, h.data(), n * sizeof(T), hipMemcpyHostToDevice)); }
    ~Dev() { (void)hipFree(p); }
    std::vector<T> host() const {
        std::vector<T> h(n);
        CK(hipMemcpy(h.data(), p, n * sizeof(T), hipMemcpyDeviceToHost));
        return h;
    }
};

static std::vector<__half> rand_half(size_t n, float scale, uint32_t seed) {
    std::mt19937 g(seed);
    std::normal_distribution<float> d(0.f, scale);
    std::vector<__half> v(n);
    for (auto& x : v) x = __float2half(d(g));
    return v;
}
static std::vector<float> rand_float(size_t n, float scale, uint32_t seed) {
    std::mt19937 g(seed);
    std::normal_distribution<float> d(0.f, scale);
    std::vector<float> v(n);
    for (auto& x : v) x = d(g);
    return v;
}

// ---- plain references (one thread per output element, fp32 accumulate; rows = a list of token indices) ----
__global__ void ref_linear(const __half* x, const __half* w, const float* b, const int* rows, int nrows, int N, float* out) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nrows * N) return;
    int r = rows[i / N], c = i % N;
    float acc = b[c];
    for (int k = 0; k < H; ++k) acc += __half2float(x[(size_t)r * H + k]) * __half2float(w[(size_t)c * H + k]);
    out[i] = acc;
}
// device-side operand fill: the host std::normal_distribution path costs a minute per GEMM shape at 262k tokens (GPU-box minutes)
__global__ void fill_half_normal(__half* p, size_t n, float scale, uint32_t seed) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint64_t z = (i + 1) * 0x9E3779B97F4A7C15ull + ((uint64_t)seed << 32);
    float acc = 0.f;
    for (int k = 0; k < 4; ++k) {  // four uniforms -> roughly normal (Irwin-Hall), unit variance after scaling
        z ^= z >> 30; z *= 0xBF58476D1CE4E5B9ull; z ^= z >> 27; z *= 0x94D049BB133111EBull; z ^= z >> 31;
        acc += (float)(z & 0xFFFFFF) * (1.0f / 16777216.0f) - 0.5f;
    }
    p[i] = __float2half(acc * 1.7320508f * scale);
}
static void dev_fill(__half* p, size_t n, float scale, uint32_t seed, hipStream_t st) {
    hipLaunchKernelGGL(fill_half_normal, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, p, n, scale, seed);
}

// general shape: out[i] = epi(x[rows[r]] . w[c] + b[c]) (+ res), the result rounded the way lm_gemm_f16 rounds it
__global__ void ref_linear_k(const __half* x, const __half* w, const float* b, const __half* res, const int* rows, int nrows, int N, int K, int epi,
                             float* out) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nrows * N) return;
    int r = rows[i / N], c = i % N;
    float acc = b[c];
    for (int k = 0; k < K; ++k) acc += __half2float(x[(size_t)r * K + k]) * __half2float(w[(size_t)c * K + k]);
    if (epi & 1) acc = 0.5f * acc * (1.0f + erff(acc * 0.70710678f));
    acc = __half2float(__float2half(acc));
    if (epi & 2) acc = __half2float(__float2half(acc + __half2float(res[(size_t)r * N + c])));
    out[i] = acc;
}
__global__ void ref_gelu_fc1(const __half* x, const __half* w1, const float* b1, const int* rows, int nrows, int F, float* out) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nrows * F) return;
    int r = rows[i / F], c = i % F;
    float acc = b1[c];
    for (int k = 0; k < H; ++k) acc += __half2float(x[(size_t)r * H + k]) * __half2float(w1[(size_t)c * H + k]);
    float ge = 0.5f * acc * (1.0f + erff(acc * 0.70710678f));
    out[i] = __half2float(__float2half(ge));  // the fused kernel feeds fp16 activations to the second product
}
__global__ void ref_fc2(const float* hid, const __half* w2, const float* b2, int nrows, int F, float* out) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nrows * H) return;
    int r = i / H, c = i % H;
    float acc = b2[c];
    for (int k = 0; k < F; ++k) acc += hid[(size_t)r * F + k] * __half2float(w2[(size_t)c * F + k]);
    out[i] = acc;
}
// out[r] = LayerNorm(z[r] + res[rows[r]]) * gamma + beta
__global__ void ref_add_ln(const float* z, const __half* res, const int* rows, int nrows, const __half* gamma, const __half* beta, float eps,
                           float* out) {
    int r = blockIdx.x;
    __shared__ float sh[H];
    __shared__ float stat[2];
    for (int c = threadIdx.x; c < H; c += blockDim.x) sh[c] = z[(size_t)r * H + c] + (res ? __half2float(res[(size_t)rows[r] * H + c]) : 0.f);
    __syncthreads();
    if (threadIdx.x == 0) {
        double m = 0, v = 0;
        for (int c = 0; c < H; ++c) m += sh[c];
        m /= H;
        for (int c = 0; c < H; ++c) v += (sh[c] - m) * (sh[c] - m);
        stat[0] = (float)m;
        stat[1] = (float)(1.0 / sqrt(v / H + eps));
    }
    __syncthreads();
    for (int c = threadIdx.x; c < H; c += blockDim.x)
        out[(size_t)r * H + c] = (sh[c] - stat[0]) * stat[1] * __half2float(gamma[c]) + __half2float(beta[c]);
}
__global__ void ref_attn(const __half* qkv, const int* cu, int heads, float* out) {  // one thread per (token, head)
    int s = blockIdx.x, h = blockIdx.y;
    int a = cu[s], b = cu[s + 1], Hh = heads * 32;
    for (int t = a + threadIdx.x; t < b; t += blockDim.x) {
        float q[32], o[32], mx = -1e30f, den = 0.f;
        for (int d = 0; d < 32; ++d) {
            q[d] = __half2float(qkv[(size_t)t * 3 * Hh + h * 32 + d]);
            o[d] = 0.f;
        }
        for (int u = a; u < b; ++u) {
            float sc = 0.f;
            for (int d = 0; d < 32; ++d) sc += q[d] * __half2float(qkv[(size_t)u * 3 * Hh + Hh + h * 32 + d]);
            sc *= 0.17677669529663687f;
            float nm = fmaxf(mx, sc), corr = expf(mx - nm), p = expf(sc - nm);
            den = den * corr + p;
            for (int d = 0; d < 32; ++d) o[d] = o[d] * corr + p * __half2float(qkv[(size_t)u * 3 * Hh + 2 * Hh + h * 32 + d]);
            mx = nm;
        }
        for (int d = 0; d < 32; ++d) out[(size_t)t * Hh + h * 32 + d] = o[d] / den;
    }
}

__global__ void f32_to_f16(const float* a, __half* b, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) b[i] = __float2half(a[i]);
}

static double max_err_rows(const std::vector<__half>& got, int ncols, int col0, int width, const std::vector<int>& rows, const std::vector<float>& ref) {
    double m = 0;
    for (size_t i = 0; i < rows.size(); ++i)
        for (int c = 0; c < width; ++c) {
            double d = fabs((double)__half2float(got[(size_t)rows[i] * ncols + col0 + c]) - (double)ref[i * width + c]);
            if (!(d <= m)) m = d;  // NaN propagates
        }
    return m;
}

static float time_us(hipStream_t st, int reps, const std::function<void()>& f) {
    for (int i = 0; i < 3; ++i) f();
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    CK(hipEventRecord(a, st));
    for (int i = 0; i < reps; ++i) f();
    CK(hipEventRecord(b, st));
    CK(hipEventSynchronize(b));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, a, b));
    CK(hipEventDestroy(a));
    CK(hipEventDestroy(b));
    return 1e3f * ms / reps;
}

// W2 [384][F] -> [F/32][384][32] with the k permutation of leann_amd/encoder.py: fused_mlp_k_permutation
static std::vector<__half> pack_w2(const std::vector<__half>& w2, int F) {
    int perm[32];
    for (int pos = 0; pos < 32; ++pos) {
        int u = pos / 16, g = (pos % 16) / 8, e = pos % 8;
        perm[pos] = e < 4 ? 16 * u + 4 * g + e : 16 * u + 8 + 4 * g + e - 4;
    }
    std::vector<__half> p(w2.size());
    for (int s = 0; s < F / 32; ++s)
        for (int f = 0; f < H; ++f)
            for (int pos = 0; pos < 32; ++pos) p[((size_t)s * H + f) * 32 + pos] = w2[(size_t)f * F + s * 32 + perm[pos]];
    return p;
}

// W_o [384][384] -> [12][384][32], natural k order (leann_amd/encoder.py: pack_wo_slabs)
static std::vector<__half> pack_wo(const std::vector<__half>& w) {
    std::vector<__half> p(w.size());
    for (int s = 0; s < 12; ++s)
        for (int f = 0; f < H; ++f)
            for (int c = 0; c < 32; ++c) p[((size_t)s * H + f) * 32 + c] = w[(size_t)f * H + s * 32 + c];
    return p;
}
// W1 [F][384] with its columns in accumulator order (leann_amd/encoder.py: pack_w1_acc_order)
static std::vector<__half> pack_w1_acc(const std::vector<__half>& w1, int F) {
    int perm[32];
    for (int pos = 0; pos < 32; ++pos) {
        int u = pos / 16, g = (pos % 16) / 8, e = pos % 8;
        perm[pos] = e < 4 ? 16 * u + 4 * g + e : 16 * u + 8 + 4 * g + e - 4;
    }
    std::vector<__half> p(w1.size());
    for (int f = 0; f < F; ++f)
        for (int j = 0; j < H / 32; ++j)
            for (int pos = 0; pos < 32; ++pos) p[(size_t)f * H + 32 * j + pos] = w1[(size_t)f * H + 32 * j + perm[pos]];
    return p;
}

// ---- raw HBM bandwidth probes (what bounds a kernel whose output is 3x its input) ----
__global__ void bw_write(uint4* p, size_t n) {
    const uint4 v = {1u, 2u, 3u, (unsigned)threadIdx.x};
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v;
}
__global__ void bw_read(const uint4* p, size_t n, unsigned* sink) {
    unsigned acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        uint4 v = p[i];
        acc ^= v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345u) *sink = acc;
}
__global__ void bw_copy(const uint4* a, uint4* b, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) b[i] = a[i];
}
// the GEMM epilogue's pattern: a wave writes 32 rows x 32 B (row stride `stride` bytes)
__global__ void bw_write_rows32(unsigned char* p, int rows, int stride, int width) {
    const int lane = threadIdx.x & 63, wv = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, nw = (gridDim.x * blockDim.x) >> 6;
    const uint4 v = {1u, 2u, 3u, 4u};
    for (int r0 = wv * 32; r0 < rows; r0 += nw * 32)
        for (int c = 0; c < width; c += 32) {
            const int row = r0 + (lane & 31);
            if (row < rows) *(uint4*)(p + (size_t)row * stride + c + 16 * (lane >> 5)) = v;
        }
}

int main(int argc, char** argv) {
    const int T = argc > 1 ? atoi(argv[1]) : 262144 - 37;
    const int reps = argc > 2 ? atoi(argv[2]) : 20;
    const std::string what = argc > 3 ? argv[3] : "all";
    auto want = [&](const char* k) { return what == "all" || what.find(k) != std::string::npos; };
    hipStream_t st;
    CK(hipStreamCreate(&st));
    printf("{\"kbench\": \"%s\", \"tokens\": %d, \"reps\": %d}\n", lm_version(), T, reps);

    std::vector<int> rows;
    for (int i = 0; i < 192 && i < T; ++i) rows.push_back(i);
    for (int i = std::max(192, T - 192); i < T; ++i) rows.push_back(i);
    const int nr = (int)rows.size();
    Dev<int> d_rows(rows);
    auto hx = rand_half((size_t)T * H, 1.0f, 1);
    Dev<__half> x(hx);
    auto hres = rand_half((size_t)T * H, 1.0f, 2);
    Dev<__half> res(hres);
    Dev<__half> gamma(rand_half(H, 0.1f, 3)), beta(rand_half(H, 0.1f, 4));
    {  // gamma ~ 1
        auto g = gamma.host();
        for (auto& v : g) v = __float2half(1.0f + __half2float(v));
        CK(hipMemcpy(gamma.p, g.data(), H * sizeof(__half), hipMemcpyHostToDevice));
    }
    rocblas_handle rb;
    rocblas_create_handle(&rb);
    rocblas_set_stream(rb, st);
    const float one = 1.f, zero = 0.f;
    auto lib_gemm = [&](const __half* W, int N, int K, const __half* A, __half* C) {  // C[T][N] = A[T][K] W[N][K]^T (column-major view)
        rocblas_gemm_ex(rb, rocblas_operation_transpose, rocblas_operation_none, N, T, K, &one, W, rocblas_datatype_f16_r, K, A,
                        rocblas_datatype_f16_r, K, &zero, C, rocblas_datatype_f16_r, N, C, rocblas_datatype_f16_r, N, rocblas_datatype_f32_r,
                        rocblas_gemm_algo_standard, 0, 0);
    };

    if (want("bw")) {
        const size_t bytes = (size_t)T * 1152 * 2, n16 = bytes / 16;
        Dev<unsigned char> A(bytes), B(bytes);
        Dev<unsigned> sink(1);
        CK(hipMemset(A.p, 1, bytes));
        for (int blocks : {1024, 4096, 16384}) {
            float us = time_us(st, reps, [&] { hipLaunchKernelGGL(bw_write, dim3(blocks), dim3(256), 0, st, (uint4*)A.p, n16); });
            printf("{\"probe\": \"write %zu MB, 16 B/lane coalesced, %d blocks\", \"us\": %.1f, \"GBps\": %.0f}\n", bytes >> 20, blocks, us, bytes / us * 1e-3);
            us = time_us(st, reps, [&] { hipLaunchKernelGGL(bw_read, dim3(blocks), dim3(256), 0, st, (const uint4*)A.p, n16, sink.p); });
            printf("{\"probe\": \"read  %zu MB, 16 B/lane coalesced, %d blocks\", \"us\": %.1f, \"GBps\": %.0f}\n", bytes >> 20, blocks, us, bytes / us * 1e-3);
            us = time_us(st, reps, [&] { hipLaunchKernelGGL(bw_copy, dim3(blocks), dim3(256), 0, st, (const uint4*)A.p, (uint4*)B.p, n16); });
            printf("{\"probe\": \"copy  %zu MB -> %zu MB, %d blocks\", \"us\": %.1f, \"GBps_read_plus_write\": %.0f}\n", bytes >> 20, bytes >> 20, blocks, us, 2.0 * bytes / us * 1e-3);
        }
        float us = time_us(st, reps, [&] { hipLaunchKernelGGL(bw_write_rows32, dim3(2048), dim3(256), 0, st, A.p, T, 2304, 2304); });
        printf("{\"probe\": \"write %zu MB as 32 rows x 32 B per wave store (the GEMM epilogue pattern)\", \"us\": %.1f, \"GBps\": %.0f}\n", bytes >> 20, us, bytes / us * 1e-3);
        us = time_us(st, reps, [&] { CK(hipMemsetAsync(A.p, 0, bytes, st)); });
        printf("{\"probe\": \"hipMemsetAsync %zu MB\", \"us\": %.1f, \"GBps\": %.0f}\n", bytes >> 20, us, bytes / us * 1e-3);
        fflush(stdout);
    }
    if (want("linear")) {
        for (int mode = 0; mode < 2; ++mode) {
            const int N = mode == 0 ? 3 * H : H;
            auto hw = rand_half((size_t)N * H, 0.05f, 10 + mode);
            auto hb = rand_float(N, 0.2f, 20 + mode);
            Dev<__half> w(hw);
            Dev<float> b(hb);
            Dev<__half> out((size_t)T * N);
            Dev<float> zref((size_t)nr * N), lref((size_t)nr * H);
            hipLaunchKernelGGL(ref_linear, dim3((nr * N + 255) / 256), dim3(256), 0, st, x.p, w.p, b.p, d_rows.p, nr, N, zref.p);
            if (mode == 1) hipLaunchKernelGGL(ref_add_ln, dim3(nr), dim3(128), 0, st, zref.p, res.p, d_rows.p, nr, gamma.p, beta.p, 1e-12f, lref.p);
            CK(hipStreamSynchronize(st));
            auto ref = mode == 0 ? zref.host() : lref.host();
            const double gflop = 2.0 * T * (double)N * H * 1e-9;
            {   // weight-stationary form (plain weight layout); the LN variant is GEMM + lm_add_layernorm_f16
                Dev<__half> tmp((size_t)T * N);
                CK(hipMemsetAsync(out.p, 0xFF, out.n * sizeof(__half), st));
                auto run = [&] {
                    if (mode == 0) LM(lm_gemm_ws_h384_f16(x.p, w.p, b.p, N, out.p, T, st));
                    else {
                        LM(lm_gemm_ws_h384_f16(x.p, w.p, b.p, N, tmp.p, T, st));
                        LM(lm_add_layernorm_f16(tmp.p, res.p, gamma.p, beta.p, out.p, T, H, 1e-12f, st));
                    }
                };
                run();
                CK(hipStreamSynchronize(st));
                const double err = max_err_rows(out.host(), N, 0, N, rows, ref);
                const float us = time_us(st, reps, run);
                printf("{\"kernel\": \"lm_gemm_ws_h384_f16%s\", \"mode\": \"%s\", \"us\": %.1f, \"TFLOPs\": %.3f, \"max_abs_err\": %.3g}\n",
                       mode ? " + lm_add_layernorm_f16" : "", mode ? "out-proj+res+LN (N=384)" : "QKV (N=1152)", us, gflop / us, err);
                fflush(stdout);
            }
            const float us = time_us(st, reps, [&] { lib_gemm(w.p, N, H, x.p, out.p); });
            printf("{\"kernel\": \"rocblas_gemm_ex f16 (no bias / LN)\", \"mode\": \"N=%d K=384\", \"us\": %.1f, \"TFLOPs\": %.1f}\n", N, us, gflop / us);
            fflush(stdout);
        }
    }
    if (want("qkv")) {  // QKV projection: the weight-stationary kernel (x read six times) vs the weight-streaming one with two waves per SIMD (lm_qkv_h384.hip)
        const int N = 3 * H;
        auto hw = rand_half((size_t)N * H, 0.05f, 60);
        Dev<__half> w(hw), wimg((size_t)N * H), outa((size_t)T * N), outb((size_t)T * N);
        Dev<float> b(rand_float(N, 0.2f, 61));
        Dev<float> zref((size_t)nr * N);
        LM(lm_qkv_pack_h384(w.p, N, wimg.p, st));
        hipLaunchKernelGGL(ref_linear, dim3((nr * N + 255) / 256), dim3(256), 0, st, x.p, w.p, b.p, d_rows.p, nr, N, zref.p);
        auto runa = [&] { LM(lm_gemm_ws_h384_f16(x.p, w.p, b.p, N, outa.p, T, st)); };
        auto runb = [&] { LM(lm_qkv_h384_f16(x.p, wimg.p, b.p, N, outb.p, T, st)); };
        CK(hipMemsetAsync(outb.p, 0xFF, outb.n * sizeof(__half), st));
        runa();
        runb();
        CK(hipStreamSynchronize(st));
        auto ref = zref.host();
        double dmax = 0;
        {
            auto a = outa.host(), c = outb.host();
            for (size_t i = 0; i < a.size(); ++i) {
                double d = fabs((double)__half2float(a[i]) - (double)__half2float(c[i]));
                if (!(d <= dmax)) dmax = d;
            }
            printf("{\"kernel\": \"lm_qkv_h384_f16\", \"max_abs_err\": %.3g, \"ws_max_abs_err\": %.3g, \"max_abs_diff_all_rows\": %.3g}\n", max_err_rows(c, N, 0, N, rows, ref),
                   max_err_rows(a, N, 0, N, rows, ref), dmax);
        }
        for (int round = 0; round < 3; ++round) {
            const float ua = time_us(st, reps, runa), ub = time_us(st, reps, runb);
            printf("{\"kernel\": \"lm_gemm_ws_h384_f16 (weight stationary)\", \"mode\": \"N=1152\", \"round\": %d, \"us\": %.1f, \"TFLOPs\": %.1f}\n", round, ua, 2.0 * T * N * H / ua * 1e-6);
            printf("{\"kernel\": \"lm_qkv_h384_f16 (weight streaming, 2 waves per SIMD)\", \"mode\": \"N=1152\", \"round\": %d, \"us\": %.1f, \"TFLOPs\": %.1f}\n", round, ub, 2.0 * T * N * H / ub * 1e-6);
            fflush(stdout);
        }
    }
    if (want("wsgemm")) {  // weight-stationary GEMM alone: timing + where a wave's cycles go (LEANN_MI355X_ABLATE=64 / 65: phase stamps)
        for (int N : {3 * H, H}) {
            auto hw = rand_half((size_t)N * H, 0.05f, 50 + N);
            Dev<__half> w(hw), out((size_t)T * N);
            Dev<float> b(rand_float(N, 0.2f, 51));
            auto run = [&] { LM(lm_gemm_ws_h384_f16(x.p, w.p, b.p, N, out.p, T, st)); };
            const float us = time_us(st, reps, run);
            printf("{\"kernel\": \"lm_gemm_ws_h384_f16\", \"mode\": \"N=%d\", \"us\": %.1f, \"TFLOPs\": %.3f}\n", N, us, 2.0 * T * N * H / us * 1e-6);
            for (const char* ab : {"64", "65"}) {
                setenv("LEANN_MI355X_ABLATE", ab, 1);
                for (int rep = 0; rep < 2; ++rep) run();
                CK(hipStreamSynchronize(st));
                const float us2 = time_us(st, reps, run);
                std::vector<unsigned long long> t(256 * 8 * 4);
                CK(hipMemcpy(t.data(), out.p, t.size() * 8, hipMemcpyDeviceToHost));
                double sum[3] = {0, 0, 0}, tiles = 0;
                const int nblk = N / 192, used = (32 / nblk) * nblk;
                for (int blk = 0; blk < 256; ++blk) {
                    if ((blk >> 3) >= used) continue;
                    for (int wv = 0; wv < 8; ++wv) {
                        const unsigned long long* q = &t[(size_t)(blk * 8 + wv) * 4];
                        sum[0] += (double)q[0]; sum[1] += (double)q[1]; sum[2] += (double)q[2]; tiles += (double)q[3];
                    }
                }
                printf("{\"kernel\": \"lm_gemm_ws_h384_f16 stamps\", \"mode\": \"N=%d, ablate %s (64: loads waited in full; 65: + stores drained)\", \"us\": %.1f, "
                       "\"mean_cycles_per_wave_tile\": {\"row loads (issue + wait)\": %.0f, \"144 MFMAs\": %.0f, \"stores\": %.0f}, \"wave_tiles\": %.0f}\n",
                       N, ab, us2, sum[0] / tiles, sum[1] / tiles, sum[2] / tiles, tiles);
                fflush(stdout);
            }
            unsetenv("LEANN_MI355X_ABLATE");
        }
    }
    if (want("tail")) {  // second half of a layer: out-projection + LN + feed-forward block + LN -- three kernels vs the fused one
        const int F = 1536;
        auto hwo = rand_half((size_t)H * H, 0.05f, 40), hw1 = rand_half((size_t)F * H, 0.05f, 41), hw2 = rand_half((size_t)H * F, 0.03f, 42);
        Dev<__half> wo(hwo), wop(pack_wo(hwo)), w1(hw1), w1a(pack_w1_acc(hw1, F)), w2(hw2), w2p(pack_w2(hw2, F));
        Dev<float> bo(rand_float(H, 0.2f, 43)), b1(rand_float(F, 0.2f, 44)), b2(rand_float(H, 0.2f, 45));
        Dev<__half> gamma1(rand_half(H, 0.1f, 46)), beta1(rand_half(H, 0.1f, 47));
        {
            auto g = gamma1.host();
            for (auto& v : g) v = __float2half(1.0f + __half2float(v));
            CK(hipMemcpy(gamma1.p, g.data(), H * sizeof(__half), hipMemcpyHostToDevice));
        }
        Dev<__half> y0((size_t)T * H), x1((size_t)T * H), out3((size_t)T * H), outf((size_t)T * H), x1h((size_t)nr * H);
        Dev<float> z0((size_t)nr * H), x1f((size_t)nr * H), hid((size_t)nr * F), z((size_t)nr * H), lref((size_t)nr * H);
        std::vector<int> ident(nr);
        for (int i = 0; i < nr; ++i) ident[i] = i;
        Dev<int> d_ident(ident);
        // reference on the sampled rows: x = attention output, res = the layer's input
        hipLaunchKernelGGL(ref_linear, dim3((nr * H + 255) / 256), dim3(256), 0, st, x.p, wo.p, bo.p, d_rows.p, nr, H, z0.p);
        hipLaunchKernelGGL(ref_add_ln, dim3(nr), dim3(128), 0, st, z0.p, res.p, d_rows.p, nr, gamma1.p, beta1.p, 1e-12f, x1f.p);
        hipLaunchKernelGGL(f32_to_f16, dim3((nr * H + 255) / 256), dim3(256), 0, st, x1f.p, x1h.p, (size_t)nr * H);
        hipLaunchKernelGGL(ref_gelu_fc1, dim3((nr * F + 255) / 256), dim3(256), 0, st, x1h.p, w1.p, b1.p, d_ident.p, nr, F, hid.p);
        hipLaunchKernelGGL(ref_fc2, dim3((nr * H + 255) / 256), dim3(256), 0, st, hid.p, w2.p, b2.p, nr, F, z.p);
        hipLaunchKernelGGL(ref_add_ln, dim3(nr), dim3(128), 0, st, z.p, x1h.p, d_ident.p, nr, gamma.p, beta.p, 1e-12f, lref.p);
        CK(hipStreamSynchronize(st));
        auto ref = lref.host();
        const double gflop = (4.0 * F * H + 2.0 * H * H) * T * 1e-9;
        Dev<__half> hmid((size_t)T * F);
        auto run3 = [&] {  // the unfused form (LEANN_MI355X_TAIL=0 in the Python host): five launches, the intermediate through HBM
            LM(lm_gemm_ws_h384_f16(x.p, wo.p, bo.p, H, y0.p, T, st));
            LM(lm_add_layernorm_f16(y0.p, res.p, gamma1.p, beta1.p, x1.p, T, H, 1e-12f, st));
            LM(lm_gemm_f16(x1.p, w1.p, b1.p, nullptr, 1, F, H, hmid.p, T, st));
            LM(lm_gemm_f16(hmid.p, w2.p, b2.p, x1.p, 2, H, F, y0.p, T, st));
            LM(lm_add_layernorm_f16(y0.p, nullptr, gamma.p, beta.p, out3.p, T, H, 1e-12f, st));
        };
        auto runf = [&] {
            LM(lm_attn_out_mlp_fused_h384_f16(x.p, res.p, wop.p, bo.p, gamma1.p, beta1.p, 1e-12f, w1a.p, b1.p, w2p.p, b2.p, gamma.p, beta.p, outf.p, T,
                                              F, 1e-12f, st));
        };
        CK(hipMemsetAsync(out3.p, 0xFF, out3.n * sizeof(__half), st));
        CK(hipMemsetAsync(outf.p, 0xFF, outf.n * sizeof(__half), st));
        run3();
        runf();
        CK(hipStreamSynchronize(st));
        const double e3 = max_err_rows(out3.host(), H, 0, H, rows, ref), ef = max_err_rows(outf.host(), H, 0, H, rows, ref);
        double dmax = 0;
        {
            auto a = out3.host(), b = outf.host();
            for (size_t i = 0; i < a.size(); ++i) {
                double d = fabs((double)__half2float(a[i]) - (double)__half2float(b[i]));
                if (!(d <= dmax)) dmax = d;
            }
        }
        for (int round = 0; round < 2; ++round) {  // interleaved A/B
            const float us3 = time_us(st, reps, run3), usf = time_us(st, reps, runf);
            printf("{\"kernel\": \"lm_gemm_ws_h384_f16 + lm_add_layernorm_f16 + lm_gemm_f16 x 2 + lm_add_layernorm_f16\", \"mode\": \"out-proj+res+LN+MLP+res+LN, five launches\", \"round\": %d, \"us\": %.1f, \"TFLOPs\": %.3f, \"max_abs_err\": %.3g}\n", round, us3, gflop / us3 * 1e-3, e3);
            printf("{\"kernel\": \"lm_attn_out_mlp_fused_h384_f16 (generation 3, diagnosis library)\", \"mode\": \"out-proj+res+LN+MLP+res+LN, one launch\", \"round\": %d, \"us\": %.1f, \"TFLOPs\": %.3f, \"max_abs_err\": %.3g, \"max_abs_diff_vs_five_launches_all_rows\": %.3g}\n", round, usf, gflop / usf * 1e-3, ef, dmax);
            fflush(stdout);
        }
        if (want("tail4")) {  // generation 4 (lm_layer_tail_h384.hip): every schedule variant of the diagnosis library, interleaved with generation 3
            Dev<__half> woi((size_t)H * H), w1i((size_t)F * H), w2i((size_t)H * F), out4((size_t)T * H);
            LM(lm_layer_tail_pack_h384(wop.p, w1a.p, w2p.p, F, woi.p, w1i.p, w2i.p, st));
            auto run4 = [&] {
                LM(lm_layer_tail_h384_f16(x.p, res.p, woi.p, bo.p, gamma1.p, beta1.p, 1e-12f, w1i.p, b1.p, w2i.p, b2.p, gamma.p, beta.p, out4.p, T, F,
                                          1e-12f, st));
            };
            // LEANN_MI355X_TAIL4 = 1000 DM + 100 RD/4 + 10 GF + WM (DM: DMA pieces 0 at slot 0 / 1 three groups / 2 singles; RD 4 / 8; GF 1 C, 4 asm; WM 1 batched LDS waits)
            // KBENCH_TAIL4_ONLY=1: the product instance alone, no stamps (the PMC passes: scripts/pmc_sq.sh, scripts/pmc_tail.sh)
            const bool only0 = getenv("KBENCH_TAIL4_ONLY") != nullptr;
            std::vector<const char*> vars = {"0", "110", "1110", "2110", "210", "1210", "2210", "211", "1211", "2211", "2140", "2241", "1241"};
            if (only0) vars = {"0"};
            for (const char* v : vars) {  // correctness of every variant first (sampled rows against the fp32 reference, all rows against generation 3)
                setenv("LEANN_MI355X_TAIL4", v, 1);
                CK(hipMemsetAsync(out4.p, 0xFF, out4.n * sizeof(__half), st));
                run4();
                CK(hipStreamSynchronize(st));
                auto a = outf.host(), b = out4.host();
                double d4 = 0;
                for (size_t i = 0; i < a.size(); ++i) {
                    double d = fabs((double)__half2float(a[i]) - (double)__half2float(b[i]));
                    if (!(d <= d4)) d4 = d;
                }
                printf("{\"kernel\": \"lm_layer_tail_h384_f16\", \"variant\": \"%s\", \"max_abs_err\": %.3g, \"max_abs_diff_vs_generation_3_all_rows\": %.3g}\n", v,
                       max_err_rows(b, H, 0, H, rows, ref), d4);
                fflush(stdout);
            }
            for (int round = 0; round < 3; ++round) {
                const float usf = time_us(st, reps, runf);
                printf("{\"kernel\": \"lm_attn_out_mlp_fused_h384_f16 (generation 3)\", \"round\": %d, \"us\": %.1f, \"TFLOPs\": %.3f}\n", round, usf, gflop / usf * 1e-3);
                for (const char* v : vars) {
                        setenv("LEANN_MI355X_TAIL4", v, 1);
                    const float us4 = time_us(st, reps, run4);
                    printf("{\"kernel\": \"lm_layer_tail_h384_f16\", \"variant\": \"%s\", \"round\": %d, \"us\": %.1f, \"TFLOPs\": %.3f}\n", v, round, us4, gflop / us4 * 1e-3);
                    fflush(stdout);
                }
            }
            for (const char* v : {"10110", "12110", "12211", "11211", "12241"}) {  // stamps (10000 +)
                if (only0 && !(getenv("KBENCH_TAIL4_STAMP") && !strcmp(v, "12110"))) continue;  // KBENCH_TAIL4_STAMP: the stamps of the product schedule too
                setenv("LEANN_MI355X_TAIL4", v, 1);
                const int nwg = (T + 127) / 128;
                for (int rep = 0; rep < 3; ++rep) run4();
                CK(hipStreamSynchronize(st));
                const float us = time_us(st, reps, run4);
                auto ho = out4.host();
                double sum[10] = {0};
                for (int b = 0; b < nwg; ++b) {
                    unsigned long long tt[10];
                    memcpy(tt, (const char*)ho.data() + (size_t)b * 128 * H * 2, 80);
                    // stamp order in time: 0 start, 1 prologue done, 8 out-projection done, 9 LayerNorm 1 done, 2 first product, 3 iteration 0, 4 (s = 19), 5 steady done, 6 final, 7 end
                    const int ord[10] = {0, 1, 8, 9, 2, 3, 4, 5, 6, 7};
                    for (int i = 1; i < 10; ++i) sum[i] += (double)(tt[ord[i]] - tt[ord[i - 1]]);
                }
                printf("{\"kernel\": \"lm_layer_tail_h384_f16 stamps\", \"variant\": \"%s\", \"us\": %.1f, \"mean_cycles\": {\"prologue\": %.0f, \"out-projection (12 slabs)\": %.0f, \"LayerNorm 1 in registers\": %.0f, "
                       "\"first product of slab 0\": %.0f, \"iteration 0\": %.0f, \"per steady iteration s = 1..18\": %.0f, \"per steady iteration s = 19..46\": %.0f, "
                       "\"last iteration + final second product\": %.0f, \"epilogue\": %.0f}}\n",
                       v, us, sum[1] / nwg, sum[2] / nwg, sum[3] / nwg, sum[4] / nwg, sum[5] / nwg, sum[6] / nwg / 18, sum[7] / nwg / 28, sum[8] / nwg, sum[9] / nwg);
                fflush(stdout);
            }
            unsetenv("LEANN_MI355X_TAIL4");
        }
        fflush(stdout);
    }
    if (want("fusedqa")) {  // round 6: the QKV projection fused into attention (lm_qkv_attn_h384.hip) against the pair it replaces, same operands, interleaved
        const int heads = 12, N = 3 * H;
        std::mt19937 g(5);
        std::normal_distribution<float> d(180.f, 50.f);
        const char* fl = getenv("KBENCH_FIXED_LEN");  // every sequence this long instead of N(180, 50) clipped to [16, 256]
        std::vector<int> cu{0};
        while (true) {
            int len = fl ? atoi(fl) : std::min(256, std::max(16, (int)lroundf(d(g))));
            if (cu.back() + len > T) break;
            cu.push_back(cu.back() + len);
        }
        const int ns = (int)cu.size() - 1, tot = cu.back();
        Dev<int> dcu(cu);
        auto hw = rand_half((size_t)N * H, 0.05f, 60);
        Dev<__half> w(hw), wimg((size_t)N * H), qkv((size_t)tot * N), qkvr((size_t)tot * N), outp((size_t)tot * H), outr((size_t)tot * H), outf((size_t)tot * H);
        Dev<float> b(rand_float(N, 0.2f, 61));
        LM(lm_qkv_pack_h384(w.p, N, wimg.p, st));
        auto run_q = [&] { LM(lm_qkv_h384_launch(x.p, wimg.p, b.p, N, qkv.p, tot, 1, st)); };
        auto run_a = [&] { LM(lm_attn_v3_launch_hd32(qkv.p, dcu.p, ns, heads, 256, outp.p, tot, st)); };
        auto run_f = [&] { LM(lm_qkv_attn_h384_f16(x.p, wimg.p, b.p, dcu.p, ns, 256, tot, outf.p, st)); };
        // reference: the public row-major pair, then the fp32 attention reference on ITS fp16 projection (first sequences)
        LM(lm_qkv_h384_f16(x.p, wimg.p, b.p, N, qkvr.p, tot, st));
        LM(lm_attn_varlen_hd32_f16(qkvr.p, dcu.p, ns, heads, 256, outr.p, st));
        const int nchk = std::min(ns, 6);
        Dev<float> ref((size_t)cu[nchk] * H);
        hipLaunchKernelGGL(ref_attn, dim3(nchk, heads), dim3(64), 0, st, qkvr.p, dcu.p, heads, ref.p);
        CK(hipMemsetAsync(outf.p, 0xff, (size_t)tot * H * 2, st));
        run_q(); run_a(); run_f();
        CK(hipStreamSynchronize(st));
        {
            auto href = ref.host();
            std::vector<int> arows;
            for (int i = 0; i < cu[nchk]; ++i) arows.push_back(i);
            auto hp = outp.host(), hf = outf.host(), hr = outr.host();
            double dpf = 0, dpr = 0;
            size_t nan_f = 0;
            for (size_t i = 0; i < hp.size(); ++i) {
                const double vf = (double)__half2float(hf[i]);
                if (!(vf == vf)) ++nan_f;
                dpf = std::max(dpf, fabs((double)__half2float(hp[i]) - vf));
                dpr = std::max(dpr, fabs((double)__half2float(hp[i]) - (double)__half2float(hr[i])));
            }
            printf("{\"kernel\": \"lm_qkv_attn_h384_f16\", \"sequences\": %d, \"tokens\": %d, \"max_abs_err_fused_vs_fp32_attention_of_fp16_qkv\": %.3g, \"max_abs_err_pair\": %.3g, "
                   "\"max_abs_diff_fused_vs_pair_all_rows\": %.3g, \"max_abs_diff_head_major_pair_vs_row_major_pair\": %.3g, \"nan_in_fused\": %zu}\n",
                   ns, tot, max_err_rows(hf, H, 0, H, arows, href), max_err_rows(hp, H, 0, H, arows, href), dpf, dpr, nan_f);
        }
        double flops_attn = 0;
        for (int i = 0; i < ns; ++i) flops_attn += 4.0 * (double)(cu[i + 1] - cu[i]) * (cu[i + 1] - cu[i]) * H;
        const double flops = flops_attn + 2.0 * tot * N * H;
        if (getenv("KBENCH_QA_STAMPS")) {  // where a wave's cycles go (s_memtime stamps, one record per wave of ONE launch)
            setenv("LEANN_MI355X_QA_ABLATE", "16", 1);
            run_f();
            CK(hipStreamSynchronize(st));
            const size_t NW = (size_t)1 << 15, WORDS = 16;
            std::vector<unsigned long long> z(NW * WORDS);
            LM(lm_qa_stamps_read(z.data(), (int64_t)z.size(), 1));
            run_f();
            CK(hipStreamSynchronize(st));
            LM(lm_qa_stamps_read(z.data(), (int64_t)z.size(), 1));
            const float us = time_us(st, reps, run_f);
            unsetenv("LEANN_MI355X_QA_ABLATE");
            const char* names[14] = {"prologue_issue", "prologue_own_landed", "prologue_barrier", "head_start_wait_barrier", "q_slab", "q_epilogue", "k_slab", "k_epilogue", "v_slab",
                                     "v_epilogue_lds_retired", "barrier_before_tiles", "tile_loop", "normalise_store", "lifetime"};
            for (int kind = 1; kind <= 2; ++kind) {  // 1 = active waves, 2 = waves past the sequence end
                double sum[14] = {};
                double n = 0, lens = 0;
                for (size_t w = 0; w < NW; ++w) {
                    const unsigned long long* r = z.data() + w * WORDS;
                    if (r[15] != (unsigned long long)kind) continue;
                    for (int i = 0; i < 14; ++i) sum[i] += (double)r[i];
                    lens += (double)r[14];
                    n += 1;
                }
                if (n == 0) continue;
                printf("{\"kernel\": \"lm_qkv_attn_h384_f16 stamps\", \"waves\": \"%s\", \"n_waves\": %.0f, \"mean_sequence_length\": %.1f, \"us_stamped_build\": %.1f, \"mean_cycles_per_wave\": {", kind == 1 ? "active" : "past the sequence end",
                       n, lens / n, us);
                for (int i = 0; i < 14; ++i) printf("\"%s\": %.0f%s", names[i], sum[i] / n, i < 13 ? ", " : "}}\n");
            }
            fflush(stdout);
        }
        if (const char* abl = getenv("KBENCH_QA_ABLATIONS")) {  // diagnosis library: what each ingredient of the fused kernel costs (timing only, garbage results)
            for (const char* v : {"0", "1", "3", "7", "15", "2", "9"}) {
                if (v[0] != '0') setenv("LEANN_MI355X_QA_ABLATE", v, 1);
                else unsetenv("LEANN_MI355X_QA_ABLATE");
                run_f();
                CK(hipStreamSynchronize(st));
                const float uf = time_us(st, reps, run_f);
                printf("{\"kernel\": \"lm_qkv_attn_h384_f16 ablation\", \"bits\": \"%s\", \"meaning\": \"1 no tile loop, 2 no slab barriers / waits, 4 no DMA, 8 no fragment reads\", \"us\": %.1f}\n", v, uf);
                fflush(stdout);
            }
            unsetenv("LEANN_MI355X_QA_ABLATE");
            (void)abl;
        }
        for (int round = 0; round < 3; ++round) {
            const float uq = time_us(st, reps, run_q), ua = time_us(st, reps, run_a), uf = time_us(st, reps, run_f);
            printf("{\"kernel\": \"pair: lm_qkv_h384 (head major) + lm_attn_v3\", \"round\": %d, \"us_qkv\": %.1f, \"us_attn\": %.1f, \"us\": %.1f, \"TFLOPs\": %.1f}\n", round, uq, ua, uq + ua,
                   flops / (uq + ua) * 1e-6);
            printf("{\"kernel\": \"lm_qkv_attn_h384_f16 (fused)\", \"round\": %d, \"us\": %.1f, \"TFLOPs\": %.1f, \"us_per_262144_tokens\": %.1f}\n", round, uf, flops / uf * 1e-6,
                   uf * 262144.0 / tot);
            fflush(stdout);
        }
    }
    if (want("attn")) {
        const int heads = 12;
        std::mt19937 g(5);
        std::normal_distribution<float> d(180.f, 50.f);
        std::vector<int> cu{0};
        while (true) {
            int len = std::min(256, std::max(16, (int)lroundf(d(g))));
            if (cu.back() + len > T) break;
            cu.push_back(cu.back() + len);
        }
        const int ns = (int)cu.size() - 1, tot = cu.back();
        Dev<int> dcu(cu);
        Dev<__half> qkv(rand_half((size_t)tot * 3 * H, 1.0f, 40)), out((size_t)tot * H);
        const int nchk = std::min(ns, 6);
        Dev<float> ref((size_t)cu[nchk] * H);
        hipLaunchKernelGGL(ref_attn, dim3(nchk, heads), dim3(64), 0, st, qkv.p, dcu.p, heads, ref.p);
        CK(hipStreamSynchronize(st));
        auto href = ref.host();
        std::vector<int> arows;
        for (int i = 0; i < cu[nchk]; ++i) arows.push_back(i);
        double flops = 0;
        for (int i = 0; i < ns; ++i) flops += 4.0 * (double)(cu[i + 1] - cu[i]) * (cu[i + 1] - cu[i]) * H;
        // generation 2 (lm_attn_v2.hip) and generation 3 (lm_attn_v3.hip), two rounds each (interleaved: clocks / box drift);
        // KBENCH_ATTN_ONLY=<variant digit, 9 = generation 2> restricts the run to one kernel (PMC passes: one population per kernel name)
        const char* only = getenv("KBENCH_ATTN_ONLY");
        for (int round = 0; round < (only ? 1 : 2); ++round)
            for (const char* rev : {"9", "0"}) {  // generation 2 (LEANN_MI355X_ATTN3=9 routes to it), generation 3
                if (only && (only[0] != rev[0] || rev[1])) continue;
                setenv("LEANN_MI355X_ATTN3", std::string(1, rev[0]).c_str(), 1);
                setenv("LEANN_MI355X_ATTN_XCD", rev[1] ? "0" : "1", 1);
                auto run = [&] { LM(lm_attn_varlen_hd32_f16(qkv.p, dcu.p, ns, heads, 256, out.p, st)); };
                CK(hipMemsetAsync(out.p, 0xff, (size_t)tot * H * 2, st));
                run();
                CK(hipStreamSynchronize(st));
                const double err = max_err_rows(out.host(), H, 0, H, arows, href);
                const float us = time_us(st, reps, run);
                printf("{\"kernel\": \"lm_attn_varlen_hd32_f16\", \"mode\": \"%s%s, %d sequences, %d tokens\", \"round\": %d, \"us\": %.1f, \"TFLOPs\": %.1f, \"GBps_qkv_plus_out\": %.0f, \"max_abs_err\": %.3g}\n",
                       rev[0] == '9' ? "generation 2" : "generation 3", "", ns, tot, round, us, flops / us * 1e-6, (double)tot * H * 8 / us * 1e-3, err);
                fflush(stdout);
            }
        unsetenv("LEANN_MI355X_ATTN3");
        unsetenv("LEANN_MI355X_ATTN_XCD");
    }
    if (want("a3stamps")) {
        // s_memtime stamps of lm_attn_v3.hip (diagnosis library, LEANN_MI355X_ATTN3=4: the kernel + stamps; NOTE the launch time printed here comes from a 5-launch burst and reads 10 % low against a 20-launch measurement of the same kernel -- clocks): where a wave's time goes
        const int heads = 12;
        std::mt19937 g(5);
        std::normal_distribution<float> d(180.f, 50.f);
        std::vector<int> cu{0};
        while (true) {
            int len = std::min(256, std::max(16, (int)lroundf(d(g))));
            if (cu.back() + len > T) break;
            cu.push_back(cu.back() + len);
        }
        const int ns = (int)cu.size() - 1, tot = cu.back();
        Dev<int> dcu(cu);
        Dev<__half> qkv(rand_half((size_t)tot * 3 * H, 1.0f, 40)), out((size_t)tot * H);
        setenv("LEANN_MI355X_ATTN3", "4", 1);
        auto run = [&] { LM(lm_attn_varlen_hd32_f16(qkv.p, dcu.p, ns, heads, 256, out.p, st)); };
        run();
        CK(hipStreamSynchronize(st));
        const size_t NW = (size_t)1 << 17, WORDS = 12;
        std::vector<unsigned long long> z(NW * WORDS);
        LM(lm_attn_v3_stamps_read(z.data(), (int64_t)z.size(), 1));
        run();  // ONE stamped launch (records are per wave, overwritten per launch)
        CK(hipStreamSynchronize(st));
        LM(lm_attn_v3_stamps_read(z.data(), (int64_t)z.size(), 1));
        const float us = time_us(st, reps, run);
        unsetenv("LEANN_MI355X_ATTN3");
        // means over all waves, and over the waves grouped by their number of query blocks (0 = a wave with nothing to do: staging + barrier only)
        double sum[3][12] = {};
        double cnt[3] = {};
        unsigned long long tmin = ~0ull, tmax = 0;
        for (size_t w = 0; w < NW; ++w) {
            const unsigned long long* r = &z[w * WORDS];
            if (!r[8]) continue;
            const int grp = (int)std::min<unsigned long long>(r[6], 2);
            cnt[grp] += 1;
            for (int k = 0; k < 12; ++k) sum[grp][k] += (double)r[k];
            tmin = std::min(tmin, r[11]);
            tmax = std::max(tmax, r[11] + r[9]);
        }
        for (int grp = 0; grp < 3; ++grp) {
            if (cnt[grp] == 0) continue;
            const double c = cnt[grp], qb = std::max(sum[grp][6], 1.0), tl = std::max(sum[grp][7], 1.0);
            printf("{\"kernel\": \"lm_attn_v3 stamps\", \"variant\": \"%s\", \"us_stamped_build\": %.1f, \"waves_with_query_blocks\": %d, \"waves\": %.0f, \"mean_sequence_length\": %.1f, "
                   "\"mean_cycles_per_wave\": {\"lifetime\": %.0f, \"entry_to_requests_issued\": %.0f, \"own_dma_landed\": %.0f, \"barrier\": %.0f, \"query_block_setup_total\": %.0f, "
                   "\"tile_loops_total\": %.0f, \"epilogues_total\": %.0f}, \"mean_cycles\": {\"per_query_block_setup\": %.0f, \"per_tile\": %.0f, \"per_query_block_epilogue\": %.0f}, "
                   "\"launch_span_cycles_first_start_to_last_end\": %llu}\n",
                   "4", us, grp, c, sum[grp][10] / c, sum[grp][9] / c, sum[grp][0] / c, sum[grp][1] / c, sum[grp][2] / c, sum[grp][3] / c, sum[grp][4] / c, sum[grp][5] / c,
                   sum[grp][3] / qb, sum[grp][4] / tl, sum[grp][5] / qb, tmax - tmin);
        }
        fflush(stdout);
    }
    if (want("ln")) {
        Dev<__half> out((size_t)T * H);
        for (const char* rev : {"1", "2"}) {
            setenv("LEANN_MI355X_LN", rev, 1);
            const float us = time_us(st, reps, [&] { LM(lm_add_layernorm_f16(x.p, res.p, gamma.p, beta.p, out.p, T, H, 1e-12f, st)); });
            printf("{\"kernel\": \"lm_add_layernorm_f16\", \"mode\": \"revision %s\", \"us\": %.1f, \"GBps\": %.0f}\n", rev, us, (double)T * H * 6 / us * 1e-3);
        }
        unsetenv("LEANN_MI355X_LN");
    }
    if (want("gemmf16")) {
        // lm_gemm_f16 (csrc/lm_gemm_f16.hip) on the encoder's GEMM shapes, against rocBLAS (no bias / epilogue) on the same operands
        struct Shape { int N, K, epi; const char* what; };
        const Shape shapes[] = {{1152, 384, 0, "MiniLM QKV"}, {2304, 768, 0, "bge-base QKV"}, {768, 768, 2, "bge-base out-proj + residual"},
                                {3072, 768, 1, "bge-base fc1 + GELU"}, {768, 3072, 2, "bge-base fc2 + residual"}, {1536, 384, 1, "MiniLM fc1 + GELU"}};
        for (const Shape& sh : shapes) {
            const int N = sh.N, K = sh.K;
            Dev<__half> xa((size_t)T * K), wa((size_t)N * K), ra((size_t)T * N);
            dev_fill(xa.p, (size_t)T * K, 1.0f, 31, st);
            dev_fill(wa.p, (size_t)N * K, 0.05f, 32, st);
            dev_fill(ra.p, (size_t)T * N, 1.0f, 33, st);
            Dev<float> ba(rand_float(N, 0.2f, 34));
            Dev<__half> out((size_t)T * N), outl((size_t)T * N);
            Dev<float> ref((size_t)nr * N);
            hipLaunchKernelGGL(ref_linear_k, dim3((nr * N + 255) / 256), dim3(256), 0, st, xa.p, wa.p, ba.p, ra.p, d_rows.p, nr, N, K, sh.epi, ref.p);
            LM(lm_gemm_f16(xa.p, wa.p, ba.p, ra.p, sh.epi, N, K, out.p, T, st));
            CK(hipStreamSynchronize(st));
            double err = 0;
            {  // the sampled rows are the first and the last 192 of the output: copy those two slabs only
                std::vector<__half> got((size_t)T * N);  // sparse use; rows outside the slabs stay zero
                const size_t head = std::min<size_t>(192, T), tail0 = std::max<size_t>(192, T > 192 ? T - 192 : 0);
                CK(hipMemcpy(got.data(), out.p, head * N * sizeof(__half), hipMemcpyDeviceToHost));
                if (tail0 < (size_t)T) CK(hipMemcpy(got.data() + tail0 * N, out.p + tail0 * N, ((size_t)T - tail0) * N * sizeof(__half), hipMemcpyDeviceToHost));
                err = max_err_rows(got, N, 0, N, rows, ref.host());
            }
            const double gflop = 2.0 * T * N * K * 1e-9;
            for (int round = 0; round < 2; ++round) {
                const float us = time_us(st, reps, [&] { LM(lm_gemm_f16(xa.p, wa.p, ba.p, ra.p, sh.epi, N, K, out.p, T, st)); });
                const float usl = time_us(st, reps, [&] { lib_gemm(wa.p, N, K, xa.p, outl.p); });
                printf("{\"kernel\": \"lm_gemm_f16\", \"mode\": \"%s (N=%d K=%d epilogue %d)\", \"round\": %d, \"us\": %.1f, \"TFLOPs\": %.1f, \"max_abs_err\": %.3g, "
                       "\"rocblas_us\": %.1f, \"rocblas_TFLOPs\": %.1f}\n", sh.what, N, K, sh.epi, round, us, gflop / us * 1e3, err, usl, gflop / usl * 1e3);
                if (K == 384 && sh.epi == 0) {
                    const float usw = time_us(st, reps, [&] { LM(lm_gemm_ws_h384_f16(xa.p, wa.p, ba.p, N, out.p, T, st)); });
                    printf("{\"kernel\": \"lm_gemm_ws_h384_f16\", \"mode\": \"same operands\", \"round\": %d, \"us\": %.1f, \"TFLOPs\": %.1f}\n", round, usw, gflop / usw * 1e3);
                }
                fflush(stdout);
            }
        }
    }
    if (want("gemmstamp")) {
        // cycle stamps of lm_gemm_f16 (diagnosis variant 7, run with LEANN_MI355X_GEMM_VARIANT=7): per-workgroup means of wave 0's phases
        struct Shape { int N, K; const char* what; };
        for (const Shape& sh : {Shape{2304, 768, "bge-base QKV"}, Shape{3072, 768, "bge-base fc1 (no GELU here)"}, Shape{768, 3072, "bge-base fc2 (no residual here)"}, Shape{1152, 384, "MiniLM QKV"}}) {
            Dev<__half> xa((size_t)T * sh.K), wa((size_t)sh.N * sh.K), out((size_t)T * sh.N);
            dev_fill(xa.p, (size_t)T * sh.K, 1.0f, 31, st);
            dev_fill(wa.p, (size_t)sh.N * sh.K, 0.05f, 32, st);
            Dev<float> ba(rand_float(sh.N, 0.2f, 34));
            Dev<unsigned long long> dbg(8);
            for (int rep = 0; rep < 2; ++rep) {
                CK(hipMemsetAsync(dbg.p, 0, 64, st));
                LM(lm_gemm_f16(xa.p, wa.p, ba.p, dbg.p, 0, sh.N, sh.K, out.p, T, st));
                CK(hipStreamSynchronize(st));
            }
            auto h = dbg.host();
            const double nt = (double)std::max<unsigned long long>(h[5], 1), nwg = (double)std::max<unsigned long long>(h[6], 1);
            printf("{\"kernel\": \"lm_gemm_f16 stamps\", \"mode\": \"%s (N=%d K=%d)\", \"workgroups\": %llu, \"tiles\": %llu, \"cycles\": {\"per_workgroup_until_first_k_tile_landed\": %.0f, "
                   "\"per_tile_main_loop\": %.0f, \"per_tile_bias_and_lds_tile_writes\": %.0f, \"per_tile_readback_and_store_issue\": %.0f}, \"k_tiles\": %d}\n",
                   sh.what, sh.N, sh.K, h[6], h[5], h[0] / nwg, h[1] / nt, h[2] / nt, h[3] / nt, sh.K / 64);
            fflush(stdout);
        }
    }
    if (want("attn64")) {
        // head_dim 64 attention (bge-base: 12 heads x 64) on sequences of the synthetic corpus' length distribution
        const int heads = 12, Hh = heads * 64;
        std::mt19937 g(7);
        std::normal_distribution<float> nd(180.f, 50.f);
        std::vector<int> cu{0};
        while (cu.back() < T) {
            int len = std::min(256, std::max(16, (int)lroundf(nd(g))));
            cu.push_back(std::min(T, cu.back() + len));
        }
        const int ns = (int)cu.size() - 1;
        Dev<int> dcu(cu);
        Dev<__half> qkv((size_t)T * 3 * Hh), out((size_t)T * Hh);
        dev_fill(qkv.p, (size_t)T * 3 * Hh, 1.0f, 41, st);
        double flop = 0;
        for (int i = 0; i < ns; ++i) flop += 4.0 * (cu[i + 1] - cu[i]) * (double)(cu[i + 1] - cu[i]) * Hh;
        for (int round = 0; round < 2; ++round) {
            const float us = time_us(st, reps, [&] { LM(lm_attn_varlen_f16(qkv.p, dcu.p, ns, heads, 64, 256, out.p, st)); });
            printf("{\"kernel\": \"lm_attn_varlen_f16 head_dim 64\", \"mode\": \"%d sequences, %d tokens\", \"round\": %d, \"us\": %.1f, \"TFLOPs\": %.1f, \"GBps_qkv_plus_out\": %.0f}\n",
                   ns, T, round, us, flop / us * 1e-6, (double)T * Hh * 8 / us * 1e-3);
        }
        fflush(stdout);
    }
    rocblas_destroy_handle(rb);
    return 0;
}
