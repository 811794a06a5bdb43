// Host execution of the MFMA kernels of leann_amd/csrc (the real source files, compiled for x86 through the stubs
// in this directory) against plain reference implementations.  Test infrastructure only; built and run by
// tests/test_hip_emulation.py:
//     clang++ -std=c++20 -O1 -pthread -Itests/hip_emul/full -Iinclude tests/hip_emul/run_kernels.cpp -o run_kernels
// (The hidden-384 GEMM kernels -- layer tail, QKV, weight-stationary -- run in the emulated LIBRARY instead: tests/hip_emul/build_emul_lib.py.)
#define LM_HOST_EMULATION 1  // skip the launchers: this harness calls the kernels directly
#include <hip/hip_runtime.h>  // tests/hip_emul/full/hip/hip_runtime.h

#include <algorithm>
#include <cstdlib>
#include <random>

namespace lm {
__attribute__((aligned(16))) unsigned char smem[160 * 1024];  // the dynamic LDS of the one workgroup that runs at a time
}

#include "../../leann_amd/csrc/lm_encoder_ops.hip"
#include "../../leann_amd/csrc/lm_attn_v2.hip"
// lm_encoder_ops2.hip declares its LDS arrays as static __shared__ locals (no dynamic LDS): one array per workgroup
#undef __shared__
#define __shared__ static
#include "../../leann_amd/csrc/lm_encoder_ops2.hip"
#undef __shared__
#define __shared__

using h16 = _Float16;
static std::mt19937 rng(12345);
static float rnd(float s) { return s * std::normal_distribution<float>(0.f, 1.f)(rng); }
static void fill(std::vector<h16>& v, float s) {
    for (auto& x : v) x = (h16)rnd(s);
}
static void fillf(std::vector<float>& v, float s) {
    for (auto& x : v) x = rnd(s);
}
static int failures = 0;
static void report(const char* name, double err, double tol) {
    std::printf("%-46s max_abs_err=%.3e tol=%.1e %s\n", name, err, tol, err <= tol ? "ok" : "FAIL");
    if (!(err <= tol)) failures++;
}

// ---------------------------------------------------------------- attention
static void test_attention(int heads, const std::vector<int>& lens) {
    const int H = heads * 32, nseq = (int)lens.size();
    std::vector<int32_t> cu(nseq + 1, 0);
    int maxlen = 0;
    for (int i = 0; i < nseq; ++i) {
        cu[i + 1] = cu[i] + lens[i];
        maxlen = std::max(maxlen, lens[i]);
    }
    const int tot = cu[nseq];
    std::vector<h16> qkv((size_t)tot * 3 * H), out((size_t)tot * H, (h16)0);
    fill(qkv, 1.5f);
    const int nt = (maxlen + 31) / 32;
    const float scale_log2e = 1.4426950408889634f / std::sqrt(32.0f);
    {
        // the kernel deals the (sequence, head) units out per XCD: its grid is padded to a multiple of 8 workgroups
        const int n_units = nseq * heads, grid = (n_units + 7) / 8 * 8;
        emul::launch(dim3((unsigned)grid), dim3(256), 0, [&] {
            const __half* q = (const __half*)qkv.data();
            __half* o = (__half*)out.data();
#define RUN(n)                                                                                  \
    case n:                                                                                     \
        lm::k_attn_varlen_hd32_v2<n>(q, cu.data(), o, heads, scale_log2e, n_units);             \
        break
            switch (nt) { RUN(1); RUN(2); RUN(3); RUN(4); RUN(5); RUN(6); RUN(7); RUN(8); }
#undef RUN
        });
    }
    double err = 0;
    for (int s = 0; s < nseq; ++s)
        for (int h = 0; h < heads; ++h)
            for (int i = 0; i < lens[s]; ++i) {
                std::vector<double> p(lens[s]);
                double mx = -1e300, sum = 0;
                for (int j = 0; j < lens[s]; ++j) {
                    double d = 0;
                    for (int e = 0; e < 32; ++e)
                        d += (double)qkv[(size_t)(cu[s] + i) * 3 * H + h * 32 + e] * (double)qkv[(size_t)(cu[s] + j) * 3 * H + H + h * 32 + e];
                    p[j] = d / std::sqrt(32.0);
                    mx = std::max(mx, p[j]);
                }
                for (auto& x : p) {
                    x = std::exp(x - mx);
                    sum += x;
                }
                for (int e = 0; e < 32; ++e) {
                    double o = 0;
                    for (int j = 0; j < lens[s]; ++j) o += p[j] / sum * (double)qkv[(size_t)(cu[s] + j) * 3 * H + 2 * H + h * 32 + e];
                    err = std::max(err, std::fabs(o - (double)out[(size_t)(cu[s] + i) * H + h * 32 + e]));
                }
            }
    char name[96];
    std::snprintf(name, sizeof name, "attention heads=%d maxlen=%d nseq=%d", heads, maxlen, nseq);
    report(name, err, 4e-3);
}

static void layernorm_ref(std::vector<double>& z, const std::vector<h16>& gamma, const std::vector<h16>& beta) {
    double mu = 0, var = 0;
    for (double v : z) mu += v;
    mu /= z.size();
    for (double v : z) var += (v - mu) * (v - mu);
    var /= z.size();
    for (size_t f = 0; f < z.size(); ++f) z[f] = (z[f] - mu) / std::sqrt(var + 1e-12) * (double)gamma[f] + (double)beta[f];
}

// ---------------------------------------------------------------- LayerNorm (16 lanes per row), embedding front end, mean pooling
static void test_ln_pool() {
    const int H = 384, rows = 50;  // 3 full groups of 16 rows + a tail of 2 (whole 16-lane groups leave early)
    std::vector<h16> x((size_t)rows * H), r((size_t)rows * H), gamma(H), beta(H), out((size_t)rows * H, (h16)0);
    fill(x, 1.0f);
    fill(r, 3.0f);
    for (auto& v : gamma) v = (h16)(1.0f + rnd(0.1f));
    fill(beta, 0.1f);
    for (int b = 0; b < (rows + 15) / 16; ++b)
        emul::run_block(b, 256, [&] {
            lm::k_add_layernorm_f16_r16<3, true, true>((const __half*)x.data(), (const __half*)r.data(), (const __half*)gamma.data(),
                                                       (const __half*)beta.data(), (__half*)out.data(), rows, H, 1e-12f);
        });
    double err = 0;
    for (int t = 0; t < rows; ++t) {
        std::vector<double> z(H);
        for (int f = 0; f < H; ++f) z[f] = (double)x[(size_t)t * H + f] + (double)r[(size_t)t * H + f];
        layernorm_ref(z, gamma, beta);
        for (int f = 0; f < H; ++f) err = std::max(err, std::fabs(z[f] - (double)out[(size_t)t * H + f]));
    }
    report("LayerNorm 16 lanes/row (exact variant) H=384", err, 4e-3);
    // non-exact variant: H = 320 (NV = 3, last vector partly out of range)
    {
        const int H2 = 320;
        std::vector<h16> x2((size_t)rows * H2), o2((size_t)rows * H2, (h16)0), g2(H2), b2(H2);
        fill(x2, 1.0f);
        for (auto& v : g2) v = (h16)(1.0f + rnd(0.1f));
        fill(b2, 0.1f);
        for (int b = 0; b < (rows + 15) / 16; ++b)
            emul::run_block(b, 256, [&] {
                lm::k_add_layernorm_f16_r16<3, false, false>((const __half*)x2.data(), nullptr, (const __half*)g2.data(),
                                                             (const __half*)b2.data(), (__half*)o2.data(), rows, H2, 1e-12f);
            });
        double e2 = 0;
        for (int t = 0; t < rows; ++t) {
            std::vector<double> z(H2);
            for (int f = 0; f < H2; ++f) z[f] = (double)x2[(size_t)t * H2 + f];
            layernorm_ref(z, g2, b2);
            for (int f = 0; f < H2; ++f) e2 = std::max(e2, std::fabs(z[f] - (double)o2[(size_t)t * H2 + f]));
        }
        report("LayerNorm 16 lanes/row (clamped variant) H=320", e2, 4e-3);
    }
    // embedding front end
    {
        const int V = 500, Pn = 64;
        std::vector<h16> word((size_t)V * H), posw((size_t)Pn * H), type0(H), o3((size_t)rows * H, (h16)0);
        fill(word, 1.0f);
        fill(posw, 1.0f);
        fill(type0, 0.5f);
        std::vector<int32_t> tok(rows), pos(rows);
        for (int t = 0; t < rows; ++t) {
            tok[t] = (int)(rng() % V);
            pos[t] = (int)(rng() % Pn);
        }
        for (int b = 0; b < (rows + 15) / 16; ++b)
            emul::run_block(b, 256, [&] {
                lm::k_embed_layernorm_f16<3, true>(tok.data(), pos.data(), (const __half*)word.data(), (const __half*)posw.data(),
                                                   (const __half*)type0.data(), (const __half*)gamma.data(), (const __half*)beta.data(),
                                                   (__half*)o3.data(), rows, H, 1e-12f);
            });
        double e3 = 0;
        for (int t = 0; t < rows; ++t) {
            std::vector<double> z(H);
            for (int f = 0; f < H; ++f)
                z[f] = (double)(h16)((float)word[(size_t)tok[t] * H + f] + (float)type0[f]) + (double)posw[(size_t)pos[t] * H + f];
            layernorm_ref(z, gamma, beta);
            for (int f = 0; f < H; ++f) e3 = std::max(e3, std::fabs(z[f] - (double)o3[(size_t)t * H + f]));
        }
        report("embedding gather + LayerNorm H=384", e3, 4e-3);
    }
    // mean pooling
    for (int normalize = 0; normalize < 2; ++normalize) {
        const std::vector<int> lens = {7, 1, 256, 33};
        std::vector<int32_t> cu(lens.size() + 1, 0);
        for (size_t i = 0; i < lens.size(); ++i) cu[i + 1] = cu[i] + lens[i];
        std::vector<h16> xs((size_t)cu.back() * H);
        fill(xs, 1.0f);
        std::vector<float> po(lens.size() * H, 0.f);
        for (int b = 0; b < (int)lens.size(); ++b)
            emul::run_block(b, 256, [&] { lm::k_meanpool_varlen_f16((const __half*)xs.data(), cu.data(), po.data(), H, normalize); });
        double e4 = 0;
        for (size_t s = 0; s < lens.size(); ++s) {
            std::vector<double> m(H, 0.0);
            for (int t = cu[s]; t < cu[s + 1]; ++t)
                for (int f = 0; f < H; ++f) m[f] += (double)xs[(size_t)t * H + f];
            double nn = 0;
            for (auto& v : m) {
                v /= lens[s];
                nn += v * v;
            }
            for (int f = 0; f < H; ++f) e4 = std::max(e4, std::fabs((normalize ? m[f] / std::sqrt(nn) : m[f]) - (double)po[s * H + f]));
        }
        report(normalize ? "mean pooling + L2 normalise H=384" : "mean pooling H=384", e4, 1e-5);
    }
}

int main(int argc, char** argv) {
    const std::string what = argc > 1 ? argv[1] : "all";
    if (what == "all" || what == "attention") {
        test_attention(2, {70, 1, 33, 64, 2, 69});
        test_attention(1, {256, 255, 200, 129});
        for (int ml : {20, 40, 90, 100, 150, 170, 210})  // every NT instantiation (1..7; 8 above), ragged tails
            test_attention(1, {ml, ml - 1, 1, ml / 2 + 1});
    }
    if (what == "all" || what == "elementwise") test_ln_pool();
    std::printf("%s\n", failures ? "FAILED" : "ALL OK");
    return failures ? 1 : 0;
}
