// hnsw_build.cpp -- host-side HNSW graph construction (index build time, NOT the query hot path).
//
// The reference builds its graph with faiss.IndexHNSWFlat(dim, M=32, metric), efConstruction=200
// (packages/leann-backend-hnsw/leann_backend_hnsw/hnsw_backend.py:54-55,83-90) and then converts
// it to compact CSR (convert_to_csr.py).  faiss is an un-vendored submodule, so this file implements
// the published HNSW insertion algorithm (Malkov & Yashunin 2016, Alg. 1-4; the variant faiss uses:
// level-0 degree 2M, upper levels M, neighbour selection by the diversity heuristic, overflowing
// lists re-pruned with the same heuristic) and emits exactly the CSR arrays of convert_to_csr.py:494-548.
//
// Plain C ABI (called through ctypes by leann_amd/hnsw_builder.py).  g++ -O3 -fopenmp.
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <queue>
#include <random>
#include <vector>
#ifdef _OPENMP
#include <omp.h>
#endif

namespace {

struct SpinLock {
    std::atomic_flag f = ATOMIC_FLAG_INIT;
    void lock() { while (f.test_and_set(std::memory_order_acquire)) {} }
    void unlock() { f.clear(std::memory_order_release); }
};

struct Builder {
    const float* X;
    int64_t N;
    int D, metric, M, efC;
    std::vector<int32_t> levels;              // #levels per node
    std::vector<std::vector<int32_t>> links;  // links[i] = concatenated fixed-capacity slots per level
    std::vector<std::vector<uint16_t>> cnt;   // cnt[i][l] = used entries at level l
    std::vector<SpinLock> locks;
    int32_t entry = -1;
    int32_t max_level = -1;
    SpinLock global;

    int cap(int level) const { return level == 0 ? 2 * M : M; }
    int32_t* slot(int32_t i, int level) { return links[i].data() + (level == 0 ? 0 : 2 * M + (level - 1) * M); }

    float dist(const float* a, const float* b) const {
        float s = 0.f;
        if (metric == 1) {
            for (int i = 0; i < D; ++i) { float d = a[i] - b[i]; s += d * d; }
            return s;
        }
        for (int i = 0; i < D; ++i) s += a[i] * b[i];
        return -s;
    }
    const float* vec(int32_t i) const { return X + (size_t)i * D; }

    typedef std::pair<float, int32_t> DI;

    // Alg. 2 SEARCH-LAYER: ef nearest of q at `level`, starting from ep
    void search_layer(const float* q, int32_t ep, float epd, int ef, int level, std::vector<DI>& out,
                      std::vector<uint32_t>& vis, uint32_t tag) {
        std::priority_queue<DI, std::vector<DI>, std::greater<DI>> cand;  // min-heap
        std::priority_queue<DI> top;                                      // max-heap (worst on top)
        cand.push({epd, ep});
        top.push({epd, ep});
        vis[ep] = tag;
        std::vector<int32_t> nb;
        while (!cand.empty()) {
            DI c = cand.top();
            if (c.first > top.top().first && (int)top.size() >= ef) break;
            cand.pop();
            locks[c.second].lock();
            int n = cnt[c.second][level];
            nb.assign(slot(c.second, level), slot(c.second, level) + n);
            locks[c.second].unlock();
            for (int32_t v : nb) {
                if (vis[v] == tag) continue;
                vis[v] = tag;
                float d = dist(q, vec(v));
                if ((int)top.size() < ef || d < top.top().first) {
                    cand.push({d, v});
                    top.push({d, v});
                    if ((int)top.size() > ef) top.pop();
                }
            }
        }
        out.clear();
        while (!top.empty()) { out.push_back(top.top()); top.pop(); }
        std::reverse(out.begin(), out.end());  // ascending
    }

    // Alg. 4 SELECT-NEIGHBORS-HEURISTIC on an ascending candidate list
    void select(const std::vector<DI>& cands, int maxn, std::vector<DI>& out) {
        out.clear();
        for (const DI& c : cands) {
            if ((int)out.size() >= maxn) break;
            bool good = true;
            for (const DI& s : out)
                if (dist(vec(c.second), vec(s.second)) < c.first) { good = false; break; }
            if (good) out.push_back(c);
        }
    }

    void connect(int32_t a, int32_t b, float dab, int level) {
        // add b to a's list at `level`; re-prune with the heuristic when full
        locks[a].lock();
        int32_t* s = slot(a, level);
        int n = cnt[a][level];
        for (int i = 0; i < n; ++i)
            if (s[i] == b) { locks[a].unlock(); return; }
        if (n < cap(level)) {
            s[n] = b;
            cnt[a][level] = (uint16_t)(n + 1);
            locks[a].unlock();
            return;
        }
        std::vector<DI> cands;
        cands.reserve(n + 1);
        cands.push_back({dab, b});
        for (int i = 0; i < n; ++i) cands.push_back({dist(vec(a), vec(s[i])), s[i]});
        std::sort(cands.begin(), cands.end());
        std::vector<DI> keep;
        select(cands, cap(level), keep);
        for (size_t i = 0; i < keep.size(); ++i) s[i] = keep[i].second;
        cnt[a][level] = (uint16_t)keep.size();
        locks[a].unlock();
    }

    void insert(int32_t i, std::vector<uint32_t>& vis, uint32_t& tag) {
        const int lv = levels[i] - 1;
        global.lock();
        int32_t ep = entry;
        int32_t ml = max_level;
        if (ep < 0) {
            entry = i;
            max_level = lv;
            global.unlock();
            return;
        }
        bool hold = lv > ml;  // this node raises the top level: keep the global lock (as hnswlib does)
        if (!hold) global.unlock();
        const float* q = vec(i);
        float epd = dist(q, vec(ep));
        for (int level = ml; level > lv; --level) {
            bool changed = true;
            while (changed) {
                changed = false;
                locks[ep].lock();
                int n = cnt[ep][level];
                std::vector<int32_t> nb(slot(ep, level), slot(ep, level) + n);
                locks[ep].unlock();
                for (int32_t v : nb) {
                    float d = dist(q, vec(v));
                    if (d < epd) { epd = d; ep = v; changed = true; }
                }
            }
        }
        std::vector<DI> found, sel;
        for (int level = std::min(lv, ml); level >= 0; --level) {
            if (++tag == 0) { std::fill(vis.begin(), vis.end(), 0u); tag = 1; }
            search_layer(q, ep, epd, efC, level, found, vis, tag);
            select(found, M, sel);  // new node links to <= M neighbours on every level (faiss add_links)
            locks[i].lock();
            int32_t* s = slot(i, level);
            for (size_t j = 0; j < sel.size(); ++j) s[j] = sel[j].second;
            cnt[i][level] = (uint16_t)sel.size();
            locks[i].unlock();
            for (const DI& n : sel) connect(n.second, i, n.first, level);
            ep = found[0].second;
            epd = found[0].first;
        }
        if (hold) {
            entry = i;
            max_level = lv;
            global.unlock();
        }
    }
};

}  // namespace

extern "C" {

// Returns an opaque builder handle (NULL on bad arguments).  metric: 0 = inner product, 1 = L2.
void* lm_hnsw_build(const float* X, int64_t N, int32_t D, int32_t metric, int32_t M, int32_t efC,
                    uint64_t seed, int32_t nthreads) {
    if (!X || N < 0 || D <= 0 || M < 2 || efC < 1) return nullptr;
    Builder* b = new Builder();
    b->X = X; b->N = N; b->D = D; b->metric = metric; b->M = M; b->efC = std::max(efC, M);
    b->levels.resize(N);
    b->links.resize(N);
    b->cnt.resize(N);
    b->locks = std::vector<SpinLock>(N);
    std::mt19937_64 rng(seed);
    std::uniform_real_distribution<double> U(0.0, 1.0);
    const double mult = 1.0 / std::log((double)M);
    for (int64_t i = 0; i < N; ++i) {
        double u = U(rng);
        if (u < 1e-300) u = 1e-300;
        int lv = (int)(-std::log(u) * mult);
        if (lv > 30) lv = 30;
        b->levels[i] = lv + 1;
        b->links[i].assign((size_t)2 * M + (size_t)lv * M, -1);
        b->cnt[i].assign(lv + 1, 0);
    }
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#endif
    // the first few nodes serially so that the entry point / top levels exist
    int64_t serial = std::min<int64_t>(N, 1024);
    {
        std::vector<uint32_t> vis(N, 0);
        uint32_t tag = 0;
        for (int64_t i = 0; i < serial; ++i) b->insert((int32_t)i, vis, tag);
    }
#pragma omp parallel
    {
        std::vector<uint32_t> vis(N, 0);
        uint32_t tag = 0;
#pragma omp for schedule(dynamic, 64)
        for (int64_t i = serial; i < N; ++i) b->insert((int32_t)i, vis, tag);
    }
    return b;
}

void lm_hnsw_build_sizes(void* h, int64_t* n_level_ptr, int64_t* n_edges, int32_t* entry, int32_t* max_level) {
    Builder* b = (Builder*)h;
    int64_t np = 0, ne = 0;
    for (int64_t i = 0; i < b->N; ++i) {
        np += b->levels[i] + 1;
        for (int l = 0; l < b->levels[i]; ++l) ne += b->cnt[i][l];
    }
    *n_level_ptr = np; *n_edges = ne; *entry = b->entry; *max_level = b->max_level;
}

// Emit the compact-CSR arrays (convert_to_csr.py:494-548 semantics).
void lm_hnsw_build_export(void* h, int32_t* levels, uint64_t* node_offsets, uint64_t* level_ptr, int32_t* neighbors) {
    Builder* b = (Builder*)h;
    uint64_t p = 0, e = 0;
    for (int64_t i = 0; i < b->N; ++i) {
        levels[i] = b->levels[i];
        node_offsets[i] = p;
        for (int l = 0; l < b->levels[i]; ++l) {
            level_ptr[p++] = e;
            const int32_t* s = b->slot((int32_t)i, l);
            for (int j = 0; j < b->cnt[i][l]; ++j) neighbors[e++] = s[j];
        }
        level_ptr[p++] = e;
    }
    node_offsets[b->N] = p;
}

void lm_hnsw_build_free(void* h) { delete (Builder*)h; }

}  // extern "C"
