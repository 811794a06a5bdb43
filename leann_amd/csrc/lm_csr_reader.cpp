// lm_csr_reader.cpp -- parser of the LEANN HNSW index file (compact CSR and original layout).
// Field order: packages/leann-backend-hnsw/leann_backend_hnsw/convert_to_csr.py:196-237 (writer),
// :264-301,411-479 (original layout reader).  Replaces the file-parsing half of
// faiss.read_index(path, IO_FLAG_MMAP, HNSWIndexConfig) (hnsw_backend.py:145-151).
#include <cstdio>
#include <cstring>
#include <memory>

#include "lm_internal.h"

namespace lm {
namespace {

struct File {
    FILE* f = nullptr;
    ~File() { if (f) fclose(f); }
    bool rd(void* p, size_t n) { return fread(p, 1, n, f) == n; }
    template <typename T> bool val(T& v) { return rd(&v, sizeof(T)); }
    template <typename T> bool vec(std::vector<T>& v) {
        uint64_t cnt;
        if (!val(cnt) || cnt > (1ull << 40)) return false;
        v.resize(cnt);
        return cnt == 0 || rd(v.data(), cnt * sizeof(T));
    }
};

constexpr uint32_t fourcc(const char (&s)[5]) {
    return (uint32_t)(uint8_t)s[0] | ((uint32_t)(uint8_t)s[1] << 8) | ((uint32_t)(uint8_t)s[2] << 16) | ((uint32_t)(uint8_t)s[3] << 24);
}

int read_flat_storage(File& f, uint32_t fcc, int64_t ntotal, std::vector<float>& out) {
    if (fcc != fourcc("IxFI") && fcc != fourcc("IxF2") && fcc != fourcc("IxFl")) LM_FAIL(LM_EFORMAT, "unsupported storage fourcc");
    int32_t d, mt; int64_t n, dummy; uint8_t trained; float marg; uint64_t cnt;
    if (!f.val(d) || !f.val(n) || !f.val(dummy) || !f.val(dummy) || !f.val(trained) || !f.val(mt)) LM_FAIL(LM_EFORMAT, "truncated storage header");
    if (mt > 1 && !f.val(marg)) LM_FAIL(LM_EFORMAT, "truncated storage header");
    if (!f.val(cnt)) LM_FAIL(LM_EFORMAT, "truncated storage");
    if (n != ntotal || (cnt != (uint64_t)n * d && cnt != (uint64_t)n * d * 4)) LM_FAIL(LM_EFORMAT, "flat storage size mismatch");
    out.resize((size_t)n * d);
    if (n * d > 0 && !f.rd(out.data(), out.size() * 4)) LM_FAIL(LM_EFORMAT, "truncated storage payload");
    return LM_OK;
}

}  // namespace

int read_csr_file(const char* path, HostCsr& o) {
    File f;
    f.f = fopen(path, "rb");
    if (!f.f) LM_FAIL(LM_ENOENT, std::string("HNSW index file not found at ") + path);
    uint32_t fcc; int64_t dummy; uint8_t trained; float marg;
    if (!f.val(fcc) || fcc != fourcc("IHNf")) LM_FAIL(LM_EFORMAT, "not an IHNf index file");
    if (!f.val(o.d) || !f.val(o.ntotal) || !f.val(dummy) || !f.val(dummy) || !f.val(trained) || !f.val(o.metric))
        LM_FAIL(LM_EFORMAT, "truncated header");
    if (o.metric > 1 && !f.val(marg)) LM_FAIL(LM_EFORMAT, "truncated header");
    std::vector<double> probas;
    std::vector<int32_t> cum;
    if (!f.vec(probas) || !f.vec(cum) || !f.vec(o.levels)) LM_FAIL(LM_EFORMAT, "truncated HNSW vectors");
    o.ntotal = (int64_t)o.levels.size();
    long pos = ftell(f.f);
    uint8_t flag = 0xff;
    bool have = f.val(flag);
    int32_t efc, efs, ub;
    uint32_t sfcc = fourcc("null");
    if (have && flag == 1) {
        if (!f.vec(o.level_ptr) || !f.vec(o.node_offsets)) LM_FAIL(LM_EFORMAT, "truncated compact pointers");
        if (!f.val(o.entry_point) || !f.val(o.max_level) || !f.val(efc) || !f.val(efs) || !f.val(ub)) LM_FAIL(LM_EFORMAT, "truncated scalars");
        if (!f.val(sfcc)) LM_FAIL(LM_EFORMAT, "missing storage fourcc");
        if (!f.vec(o.neighbors)) LM_FAIL(LM_EFORMAT, "truncated neighbors");
    } else {
        // original layout: optional 0x00 byte, offsets, -1 padded neighbours (convert_to_csr.py:411-479)
        if (!(have && flag == 0)) fseek(f.f, pos, SEEK_SET);
        std::vector<uint64_t> offsets;
        std::vector<int32_t> nb;
        if (!f.vec(offsets) || !f.vec(nb)) LM_FAIL(LM_EFORMAT, "truncated original-layout arrays");
        if ((int64_t)offsets.size() != o.ntotal + 1) LM_FAIL(LM_EFORMAT, "offsets size mismatch");
        if (!f.val(o.entry_point) || !f.val(o.max_level) || !f.val(efc) || !f.val(efs) || !f.val(ub)) LM_FAIL(LM_EFORMAT, "truncated scalars");
        auto cum_at = [&](int level) -> int64_t {
            if (level < 0 || cum.empty()) return 0;
            return level < (int)cum.size() ? cum[level] : cum.back();
        };
        o.node_offsets.assign(o.ntotal + 1, 0);
        for (int64_t i = 0; i < o.ntotal; ++i) {
            o.node_offsets[i] = o.level_ptr.size();
            for (int l = 0; l < o.levels[i]; ++l) {
                o.level_ptr.push_back(o.neighbors.size());
                int64_t b = std::min<int64_t>((int64_t)offsets[i] + cum_at(l), (int64_t)nb.size());
                int64_t e = std::min<int64_t>(std::max<int64_t>(b, (int64_t)offsets[i] + cum_at(l + 1)), (int64_t)nb.size());
                for (int64_t j = b; j < e; ++j)
                    if (nb[j] >= 0) o.neighbors.push_back(nb[j]);
            }
            o.level_ptr.push_back(o.neighbors.size());
        }
        o.node_offsets[o.ntotal] = o.level_ptr.size();
        if (!f.val(sfcc)) sfcc = fourcc("null");
    }
    if (sfcc != fourcc("null")) {
        int rc = read_flat_storage(f, sfcc, o.ntotal, o.storage);
        if (rc) return rc;
    }
    if ((int64_t)o.node_offsets.size() != o.ntotal + 1) LM_FAIL(LM_EFORMAT, "node_offsets size mismatch");
    return LM_OK;
}

}  // namespace lm
