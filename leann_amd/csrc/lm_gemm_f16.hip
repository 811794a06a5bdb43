// lm_gemm_f16.hip -- the GENERAL fp16 linear layer of the encoder:   out[T][N] = epi(x[T][K] W[N][K]^T + b)
//
//   epi = identity                      QKV projection                         (N = 3 H)
//       = exact-erf GELU                first feed-forward product             (N = ffn)
//       = (.) + residual[T][N]          attention output projection, second feed-forward product (N = H; LayerNorm follows as
//                                       lm_add_layernorm_f16 with no residual)
//
// Why it exists: the hidden-384 kernels (lm_gemm_ws_h384 / lm_mlp_fused_v3) keep a whole 384-wide row of accumulators per wave; at
// hidden 768 (bge-base, contriever: BASELINE.json configs[4]) that does not fit a wave, and round 2 ran those models on library
// GEMMs.  This is the hand-written path for them -- any K % 128 == 0, N % 128 == 0 -- and the QKV projection of the 384 models can
// take it as well.  x and W are both K-contiguous ("B^T input"), so an MFMA fragment is 8 consecutive halfs of a row for either.
//
// Shape (MI355X_MICROARCH.md / cdna_hip_programming.md section 5, "glds, 2 LDS buffers, BK = 64"):
//   * workgroup tile 256 tokens x 256 features (BIG) -- 128 flop per byte moved L2 -> LDS: a 128 x 128 tile at full MFMA rate would
//     need 64 B/clk/CU from the L2, more than it delivers -- or 128 x 128 (SMALL: N % 256 != 0 and short launches); 8 / 4 waves,
//     each 128 features x 64 tokens (BIG) as 4 x 2 MFMA tiles of 32x32x16 (128 accumulator registers), two waves per SIMD;
//   * K in tiles of 64: one LDS stage = [256 W rows | 256 x rows] x 128 B = 64 KB, two stages; every row of a stage is one full
//     128-byte line fetched by 8 lanes of one global_load_lds_dwordx4 (1 KB per wave-instruction, 8 per wave per K-tile);
//   * LDS image: lane-linear per DMA piece (the hardware writes base + 16 lane), made conflict-free for the ds_read_b128 fragment
//     reads by permuting the SOURCE chunks: position (row, c') holds chunk c' ^ ((row >> 1) & 7) of the row.  A fragment read has
//     lanes 0..31 on 32 consecutive rows at one chunk: within each of the hardware's 16-lane groups ({0-3, 12-15, 20-27}, ...) the
//     eight row pairs have eight different (row >> 1) & 7, the two rows of a pair sit in different halves of the 256-byte bank row;
//   * one barrier per K-tile: wait own DMA (vmcnt(0)) -> barrier -> 32 MFMAs with the next K-tile's DMA pieces issued behind the first
//     two fragment requests (measured best placement: profiles/r3_session3_gemm_loop_variants.txt);
//     (round 6: the residual epilogues run a rotated form of this loop -- see GM_LOOP_FOR below -- because its registers hold a tile's residual rows ahead of time);
//   * PERSISTENT: one workgroup per CU walks a list of output tiles.  s_memtime stamps of the one-tile-per-workgroup form (QKV
//     shape, 12 K-tiles): 3.4 k cycles until the first K-tile has landed + 34.0 k main loop + 5.6 k epilogue + 0.5 k store drain,
//     and a workgroup swap on top.  Here the NEXT tile's first K-tile is requested before the epilogue starts (it lands under the
//     epilogue's LDS round trip and stores), and no workgroup is ever re-dispatched;
//   * MFMA orientation: A = W rows (M = features), B = x rows (N = tokens): a lane ends up with 4 consecutive features of ONE
//     token per accumulator quad -> packed to fp16 and written token-major into a per-wave LDS tile of 32 token rows (8-byte writes,
//     rows padded by 8 B: conflict free; one tile per 32-token MFMA column block, behind the first stage so that the next tile's
//     prefetch can use that stage), read back as whole rows and stored as 256-byte (BIG) row segments: full lines, 16 B per lane.
//     Bias and GELU are applied on the way into the tile (fp32), the residual on the way out (fp16 add, as torch's half `+` does);
//   * the tiles that share an x row block run back to back on ONE XCD (workgroup b -> XCD b % 8 is today's dispatch; any other
//     mapping costs speed only): x comes from HBM once and from that XCD's L2 afterwards, W stays L2 / Infinity-Cache resident.
// Role in the reference: the GEMMs of compute_embeddings' BERT forward (leann/embedding_compute.py:229-239) for models whose
// hidden size is not 384.
#include <algorithm>
#include <cstdlib>
#include <cstring>

#include "lm_h384_common.h"

namespace lm {

#ifdef LM_EMULATED_DEVICE
#define GM_WAIT_VM0() ((void)0)
#define GM_WAIT_VM(n) ((void)0)
#define GM_WAIT_LGKM0() ((void)0)
#define GM_WAIT_LGKM0_SEEN() ((void)0)
#define GM_WAIT_VM0_SEEN() ((void)0)
#define GM_BARRIER() __syncthreads()
#define GM_UNIFORM(v) (v)
#else
#define GM_UNIFORM(v) __builtin_amdgcn_readfirstlane(v)  // the wave index: keeps everything derived from it in SGPRs
#define GM_WAIT_VM0() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#define GM_WAIT_VM(n) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n) : "memory")
#define GM_WAIT_LGKM0() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
// the same wait as an instruction the compiler's own wait-count bookkeeping SEES (gfx9 encoding: vmcnt 63, expcnt 7, lgkmcnt 0): behind it the fragments requested
// earlier count as landed, so MFMAs that use them are not made to wait for LDS reads issued later (an asm wait is invisible to that bookkeeping)
#define GM_WAIT_LGKM0_SEEN() __builtin_amdgcn_s_waitcnt(0xC07F)
#define GM_WAIT_VM0_SEEN() __builtin_amdgcn_s_waitcnt(0x0F70)  // vmcnt 0, expcnt 7, lgkmcnt 15
#define GM_BARRIER() __builtin_amdgcn_s_barrier()
#endif

constexpr int GM_EPI_GELU = 1, GM_EPI_RESID = 2;

// exact-erf GELU, fp32, 8.5 instructions and ONE transcendental per value (derivation and error bounds: lm_mlp_fused_v3.hip, FORM 1):
//   gelu(x) = max(x, 0) - |x| 2^(-1 - u q(u)),  u = |x|,  q = degree-4 fit of -log2(erfc(u / sqrt2)) / u;  |error| < 1e-6 absolute
__device__ __forceinline__ float gm_gelu(float x) {
    const float u = fabsf(x);
    float p = fmaf(u, -0.0004881171917077154f, 0.007198805455118418f);
    p = fmaf(p, u, -0.052146803587675095f);
    p = fmaf(p, u, -0.4595957100391388f);
    p = fmaf(p, u, -1.1510006189346313f);
    const float w = __builtin_amdgcn_exp2f(fmaf(p, u, -1.0f));
    return fmaf(-u, w, fmaxf(x, 0.0f));
}

template <int WF_, int WT_, int TF_, int TT_>
struct GemmShape {
    static constexpr int WF = WF_, WT = WT_, TF = TF_, TT = TT_;  // waves along features / tokens, MFMA tiles per wave along each
    static constexpr int NW = WF * WT, THREADS = 64 * NW;
    static constexpr int BN = WF * TF * 32, BM = WT * TT * 32;     // features / tokens per workgroup tile
    static constexpr int STAGE = (BN + BM) * 128;                  // bytes of one K-tile (64 halfs per row)
    static constexpr int PIECES = (BN + BM) / 8 / NW;              // 1 KB DMA pieces per wave per K-tile
    static constexpr int RS = TF * 64 + 8;                         // bytes per token row of a wave's output tile (+ 8: conflict-free writes)
    static constexpr int OUT_TILE = 32 * RS;                       // per wave: one 32-token column block at a time
    static constexpr int OUT_OFF = STAGE;                          // the output tiles sit behind stage 0 (which the next tile's prefetch fills)
    static constexpr int BIAS_OFF = 2 * STAGE > OUT_OFF + NW * OUT_TILE ? 2 * STAGE : OUT_OFF + NW * OUT_TILE;  // this tile's BN bias values (fp32)
    static constexpr int LDS = BIAS_OFF + BN * 4;
    static_assert((BN + BM) % (8 * NW) == 0 && PIECES % 2 == 0, "DMA pieces must divide evenly over the waves and two k-steps");
    static_assert(LDS <= 160 * 1024, "LDS budget");
};
using GemmBig = GemmShape<2, 4, 4, 2>;    // 256 x 256, 512 threads, 128 accumulator registers per lane
using GemmSmall = GemmShape<2, 2, 2, 2>;  // 128 x 128, 256 threads, 64 accumulator registers per lane

// VAR (diagnosis builds only, -DLM_DIAG + LEANN_MI355X_GEMM_VARIANT; the product library holds variant 0 alone):
//   7 = s_memtime stamps: wave 0 of every workgroup adds its cycle counts {until the first K-tile has landed, main loops, bias + LDS
//       tile writes, row read-back + store issue, (unused), tiles} to the u64 words at `resid` (epilogue 0 only: the residual pointer
//       is then a debug buffer; scripts/kbench.cpp "gemmstamp").
// Launch shape (diagnosis: LEANN_MI355X_GEMM_GRID=tiles): the persistent grid is one workgroup per CU; a grid of one workgroup per
// tile runs the same code with tile lists of length one (the round-3 session-2..4 form, for A/B).
constexpr int GM_VAR_DEFAULT = 0;
// LOOP (round 6): 0 = the round-5 tile loop (barrier in front of a K-tile's first fragment reads; bias slice staged in LDS and added in the epilogue; a residual
// row loaded where it is added).  3 = the ROTATED loop (barrier in front of a K-tile's LAST k-step; the accumulators start at the bias) + all residual rows of a tile
// requested ahead with one wait.  Measured on the encoder's shapes at 65,536 tokens, three interleaved rounds (GPU session 15): the rotation by itself is worth nothing
// (QKV of a 768-wide model 241-244 us in form 0, 250-251 in form 3; first feed-forward product + GELU 361 / 367-372; MiniLM QKV 82-84 / 86-95) -- the barrier release and
// the first LDS round trip of a K-tile were NOT what the 8-wave shape loses against the vendor library's bare product (204 / 272 us); the residual rows ahead are worth
// 3-7 % (out-projection + residual 111-115 -> 104-106 us, second feed-forward product + residual 302-303 -> 291-294): every residual load of form 0 sits in the branch of
// its store behind its own s_waitcnt vmcnt(0) -- 16 serial round trips per 256 x 256 tile.  Form 0's registers do not hold the rows of form 3's epilogue (30 VGPRs of
// scratch), so: the residual epilogues run form 3, the others form 0.  Also built, measured there and deleted: the next K-tile's DMA pieces spread one behind each MFMA
// of the last k-step (equal), and a four-wave shape of 128 x 128 per wave (one wave per SIMD, a third less LDS read traffic: 20-60 % SLOWER -- a wave alone on its
// SIMD hides neither its own DMA issue nor the barrier).
#define GM_LOOP_FOR(EPI) (((EPI) & 2) != 0 ? 3 : 0)

#if defined(LM_DIAG) && !defined(LM_EMULATED_DEVICE)
#define GM_STAMP(t)                                     \
    do {                                                \
        if constexpr (VAR == 7) {                       \
            __builtin_amdgcn_sched_barrier(0);          \
            t = __builtin_amdgcn_s_memtime();           \
            __builtin_amdgcn_sched_barrier(0);          \
        }                                               \
    } while (0)
#else
#define GM_STAMP(t) ((void)0)
#endif

// Tile order: XCD x (workgroup b runs on XCD b % 8 today; any other mapping costs speed only) owns the row blocks rb = x (mod 8) and
// walks its tiles i = 0, 1, ... as (rb = (i / NC) * 8 + x, column tile i % NC): the NC tiles of a row block are neighbours in time on
// ONE XCD, so x is fetched from HBM once and re-read from that XCD's L2.  Workgroup b = (slot = b / 8, x = b % 8) takes the tiles
// i = slot, slot + gridDim.x / 8, ... of its XCD.
template <class S, int EPI, int VAR, int LOOP>
__global__ __launch_bounds__(S::THREADS) LM_TWO_WAVES_PER_SIMD void k_gemm_f16(
    const __half* __restrict__ x, const __half* __restrict__ w, const float* __restrict__ bias, const __half* __restrict__ resid,
    __half* __restrict__ out, int T, int N, int K) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wv = GM_UNIFORM(tid >> 6);
    const int r31 = lane & 31, g = lane >> 5;
    const int NC = (N + S::BN - 1) / S::BN;  // the last column tile may be partial (N % 128 == 0): its surplus rows of W re-read row N - 1
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, nslot = gridDim.x >> 3;
    const int nrb = (T + S::BM - 1) / S::BM;
    const int tiles_x = ((nrb - xcd + 7) >> 3) * NC;  // tiles of this XCD (row blocks xcd, xcd + 8, ... < nrb)
    if (slot >= tiles_x) return;
    const int wf = wv % S::WF, wt = wv / S::WF;
    [[maybe_unused]] unsigned long long ts0 = 0, ts1 = 0, tacc[4] = {0, 0, 0, 0}, ntile = 0;
    GM_STAMP(ts0);

    // ---- DMA plan: piece p = wv + NW * i covers stage rows 8 p .. 8 p + 7 (W rows first, then x rows); lane = (row 8 p + lane / 8,
    //      position c' = lane % 8) fetches chunk c' ^ ((row >> 1) & 7) of the row's 128-byte line.  Source = wave-uniform base +
    //      per-lane 32-bit offset (recomputed per tile). ----
    unsigned voff[S::PIECES];
    auto plan = [&](int t0, int n0) {
#pragma unroll
        for (int i = 0; i < S::PIECES; ++i) {
            const int row = 8 * (wv + S::NW * i) + (lane >> 3);
            unsigned grow;
            if (row < S::BN) {
                grow = (unsigned)(n0 + row < N ? n0 + row : N - 1);  // features past N (partial last column tile): never stored
            } else {
                const int tok = t0 + row - S::BN;
                grow = (unsigned)(tok < T ? tok : T - 1);  // rows past the end re-read the last token; their results are never stored
            }
            voff[i] = grow * (unsigned)(K * 2) + (((lane & 7) ^ ((row >> 1) & 7)) << 4);
        }
    };
    auto issue_piece = [&](int kt, int stage, int i) {
        const int p = wv + S::NW * i;  // wave uniform
        const unsigned char* base = (const unsigned char*)(8 * p < S::BN ? (const void*)w : (const void*)x) + (size_t)kt * 128;
        lm_dma16_sv(base, voff[i], smem + stage * S::STAGE + p * 1024);
    };
    // ---- fragment addresses: row (tile base + r31), k-step kk (16 halfs): chunk 2 kk + g at position (2 kk + g) ^ ((r31 >> 1) & 7)
    //      (tile bases are multiples of 32 rows, so the row term of the permutation depends on r31 only) ----
    int fo[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) fo[kk] = r31 * 128 + ((((2 * kk + g) ^ (r31 >> 1)) & 7) << 4);
    const int a_base = wf * S::TF * 32 * 128;                 // W rows of this wave
    const int b_base = S::BN * 128 + wt * S::TT * 32 * 128;   // x rows of this wave
    const int nk = K / 64;  // even (K % 128 == 0)

    float16v acc[S::TF][S::TT];
    // One K-tile: four k-steps of TF x TT MFMAs.  The fragments of step kk + 1 are requested BEFORE the MFMAs of step kk are issued (a
    // second register set: the matrix pipe never waits for an LDS round trip inside a tile); the NEXT K-tile's DMA pieces go behind the
    // first two fragment requests.  The scheduling barriers pin that order: the register budget (256 per wave) leaves the compiler no
    // room to find it by itself.
    auto ktile = [&](int stage, int next_kt, bool prefetch) {
        const unsigned char* sb = smem + stage * S::STAGE;
        half8 af[2][S::TF], bf[2][S::TT];
#pragma unroll
        for (int i = 0; i < S::TF; ++i) af[0][i] = *(const half8*)(sb + a_base + i * 4096 + fo[0]);
#pragma unroll
        for (int j = 0; j < S::TT; ++j) bf[0][j] = *(const half8*)(sb + b_base + j * 4096 + fo[0]);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const int c = kk & 1, n = c ^ 1;
            if (kk < 3) {
#pragma unroll
                for (int i = 0; i < S::TF; ++i) af[n][i] = *(const half8*)(sb + a_base + i * 4096 + fo[kk + 1]);
#pragma unroll
                for (int j = 0; j < S::TT; ++j) bf[n][j] = *(const half8*)(sb + b_base + j * 4096 + fo[kk + 1]);
            }
            // (Measured and dropped, GPU session r3-10: letting the second half of the waves issue its pieces BEHIND the k-step's MFMAs so
            // that one wave's DMA issue sits under its SIMD partner's MFMAs -- 1017 vs 962 us on the QKV shape.)
            if (prefetch && kk < 2) {
#pragma unroll
                for (int i = 0; i < S::PIECES / 2; ++i) issue_piece(next_kt, stage ^ 1, kk * (S::PIECES / 2) + i);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < S::TF; ++i)
#pragma unroll
                for (int j = 0; j < S::TT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[c][i], bf[c][j], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    int ti = slot;
    int t0 = ((ti / NC) * 8 + xcd) * S::BM, n0 = (ti % NC) * S::BN;
    plan(t0, n0);
#pragma unroll
    for (int i = 0; i < S::PIECES; ++i) issue_piece(0, 0, i);
    for (;;) {
        if constexpr ((LOOP & 1) != 0) {
            // the accumulators START at the bias (register 4 q + e of tile (i, j) <-> feature 32 i + 8 q + 4 g + e of the wave's slice, whatever the token):
            // no bias slice in LDS, no add in the epilogue (as the hidden-384 kernels do it: the bias is the first MFMA's C operand)
            const int fb = n0 + wf * S::TF * 32 + 4 * g;
#pragma unroll
            for (int i = 0; i < S::TF; ++i)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int f = fb + 32 * i + 8 * q;
                    const float4v bv = *(const float4v*)(bias + (f < N ? f : N - 4));  // (features past N -- a partial last column tile -- are never stored)
#pragma unroll
                    for (int j = 0; j < S::TT; ++j)
#pragma unroll
                        for (int e = 0; e < 4; ++e) acc[i][j][4 * q + e] = bv[e];
                }
        } else {
#pragma unroll
            for (int i = 0; i < S::TF; ++i)
#pragma unroll
                for (int j = 0; j < S::TT; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        }
        [[maybe_unused]] unsigned long long tm0 = 0, tm1 = 0, tm2 = 0, tm3 = 0;
        if constexpr ((LOOP & 1) != 0) {
            // ROTATED main loop (round 6).  The barrier of a K-tile sits in front of its LAST k-step's MFMAs instead of in front of its first fragment
            // reads: when a wave reaches it, that step's fragments are already in registers (lgkmcnt(0): the wave is done reading the stage) and its pieces
            // of the next K-tile have landed (vmcnt(0)).  Behind the barrier the next K-tile's first fragments are requested from the other stage and the
            // K-tile after that is requested into the stage that just fell free -- both UNDER the last step's MFMAs: the matrix pipe no longer idles through
            // a barrier release + an LDS round trip once per K-tile (the form above: wait, barrier, then the first fragment reads with nothing to issue).
            half8 af[2][S::TF], bf[2][S::TT];
            auto rd = [&](const unsigned char* sb, int kk, half8 (&a)[S::TF], half8 (&b)[S::TT]) {
#pragma unroll
                for (int i = 0; i < S::TF; ++i) a[i] = *(const half8*)(sb + a_base + i * 4096 + fo[kk]);
#pragma unroll
                for (int j = 0; j < S::TT; ++j) b[j] = *(const half8*)(sb + b_base + j * 4096 + fo[kk]);
            };
            auto mm = [&](const half8 (&a)[S::TF], const half8 (&b)[S::TT]) {
#pragma unroll
                for (int i = 0; i < S::TF; ++i)
#pragma unroll
                    for (int j = 0; j < S::TT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i], b[j], acc[i][j], 0, 0, 0);
            };
            GM_WAIT_VM0_SEEN();  // this wave's pieces of the tile's first K-tile have landed (its stores of the previous tile's epilogue are out, the bias values are in) ...
            GM_BARRIER();        // ... and everybody else's; all waves are done with the previous tile's epilogue (output tiles = stage 1's memory)
            GM_STAMP(tm0);
            if (ntile == 0) ts1 = tm0;
            rd(smem, 0, af[0], bf[0]);
#pragma unroll
            for (int p = 0; p < S::PIECES; ++p) issue_piece(1, 1, p);  // (nk >= 2)
            for (int kt = 0; kt < nk; kt += 2) {
#pragma unroll
                for (int h = 0; h < 2; ++h) {  // K-tile kt + h out of stage h
                    const unsigned char* sb = smem + h * S::STAGE;
                    const unsigned char* so = smem + (h ^ 1) * S::STAGE;
                    rd(sb, 1, af[1], bf[1]);
                    __builtin_amdgcn_sched_barrier(0);
                    mm(af[0], bf[0]);
                    __builtin_amdgcn_sched_barrier(0);
                    rd(sb, 2, af[0], bf[0]);
                    __builtin_amdgcn_sched_barrier(0);
                    mm(af[1], bf[1]);
                    __builtin_amdgcn_sched_barrier(0);
                    rd(sb, 3, af[1], bf[1]);
                    __builtin_amdgcn_sched_barrier(0);
                    mm(af[0], bf[0]);
                    __builtin_amdgcn_sched_barrier(0);
                    GM_WAIT_LGKM0_SEEN();  // step 3's fragments are in registers: this wave's last read of stage h is complete
                    GM_WAIT_VM0();    // this wave's pieces of K-tile kt + h + 1 have landed
                    GM_BARRIER();     // everybody's: stage h ^ 1 is complete, stage h is free
                    const bool more_k = kt + h + 1 < nk;
                    if (more_k) rd(so, 0, af[0], bf[0]);
                    __builtin_amdgcn_sched_barrier(0);
                    if (kt + h + 2 < nk) {
#pragma unroll
                        for (int p = 0; p < S::PIECES; ++p) issue_piece(kt + h + 2, h, p);
                    }
                    mm(af[1], bf[1]);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            // (the loop's last barrier: every wave is done with both stages)
        } else {
        for (int kt = 0; kt < nk; kt += 2) {
            GM_WAIT_VM0();   // this wave's pieces of K-tile kt have landed (and its stores of the previous tile's epilogue are out) ...
            GM_BARRIER();    // ... and so have everybody else's; all waves are done with stage 1 (K-tile kt - 1 / the previous output tiles)
            float bias_v = 0.f;
            if (kt == 0) {
                GM_STAMP(tm0);
                if (ntile == 0) ts1 = tm0;
                // this tile's bias slice -> LDS (read in the epilogue with ds_read: a vector-memory load there would have to wait for
                // the NEXT tile's prefetched K-tile, which is older in the wave's in-order vmcnt queue -- stamps: 9.1 k cycles for the
                // bias + tile-write phase instead of 3.0 k).  Requested here, written behind the wait that the K-tile needs anyway.
                if (tid < S::BN) bias_v = bias[n0 + tid < N ? n0 + tid : N - 1];
            }
            ktile(0, kt + 1, true);
            GM_WAIT_LGKM0();  // own fragment reads of stage 0 are complete before anybody's DMA may overwrite it
            GM_WAIT_VM0();
            if (kt == 0 && tid < S::BN) ((float*)(smem + S::BIAS_OFF))[tid] = bias_v;
            GM_BARRIER();
            ktile(1, kt + 2, kt + 2 < nk);
            GM_WAIT_LGKM0();
        }
        GM_BARRIER();  // every wave is done with both stages
        }
        GM_STAMP(tm1);
        // ---- the NEXT tile's first K-tile goes into stage 0 now: it lands while this tile's epilogue runs ----
        const int ti_next = ti + nslot;
        const bool more = ti_next < tiles_x;
        const int t0_cur = t0, n0_cur = n0;
        if (more) {
            t0 = ((ti_next / NC) * 8 + xcd) * S::BM;
            n0 = (ti_next % NC) * S::BN;
            plan(t0, n0);
#pragma unroll
            for (int i = 0; i < S::PIECES; ++i) issue_piece(0, 0, i);
        }
        // ---- epilogue, one 32-token column block (j) at a time.  (1) + bias (, GELU), fp16, token-major into this wave's LDS tile:
        //      acc[i][j][4 q + e] = feature 32 i + 8 q + 4 g + e of token 32 j + r31 (relative to the wave's sub-tile).  (2) whole rows
        //      out: LPR lanes cover one token's TF * 32 features (16 B each), 64 / LPR rows per instruction ----
        unsigned char* ot = smem + S::OUT_OFF + wv * S::OUT_TILE;
        const float* bl = (const float*)(smem + S::BIAS_OFF) + wf * S::TF * 32 + 4 * g;  // staged at the tile's first K-tile
        constexpr int LPR = S::TF * 4, RPI = 64 / LPR;
        const int lr = lane / LPR, lc = lane % LPR;
        const int64_t col = n0_cur + wf * S::TF * 32 + lc * 8;
        // Residual rows (LOOP & 2, round 6): ALL of the tile's rows requested up front with CLAMPED addresses (behind the next tile's prefetch), ONE wait for them
        // behind the first column block's LDS writes -- instead of a load inside the `if (tok < T && col < N)` of the store loop, where every one of the
        // 32 / RPI * TT loads got its own s_waitcnt vmcnt(0): 16 serial round trips per 256 x 256 tile, each of them also waiting for the previous row's
        // store (the out-projection of a 768-wide model: 116 us against the vendor library's 75 us per 65 k tokens).  The wait is one the compiler's
        // bookkeeping sees: the stores that follow (conditional, hence "maybe outstanding" to that bookkeeping) then never stand between a load and its use.
        constexpr int NIT = 32 / RPI;
        typedef unsigned gm_u32x4 __attribute__((ext_vector_type(4)));
        [[maybe_unused]] gm_u32x4 rr[S::TT][NIT];
        if constexpr ((EPI & GM_EPI_RESID) != 0 && (LOOP & 2) != 0) {
            const int64_t colc = col < N ? col : (int64_t)N - 8;
#pragma unroll
            for (int j = 0; j < S::TT; ++j)
#pragma unroll
                for (int it = 0; it < NIT; ++it) {
                    const int tok = t0_cur + wt * S::TT * 32 + 32 * j + it * RPI + lr;
                    rr[j][it] = *(const gm_u32x4*)((const _Float16*)resid + (int64_t)(tok < T ? tok : T - 1) * N + colc);
                }
        }
#pragma unroll
        for (int j = 0; j < S::TT; ++j) {
            GM_STAMP(tm2);
#pragma unroll
            for (int i = 0; i < S::TF; ++i)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float4v bb = {0.f, 0.f, 0.f, 0.f};
                    if constexpr ((LOOP & 1) == 0) bb = *(const float4v*)(bl + 32 * i + 8 * q);
                    half4 h;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float v = acc[i][j][4 * q + e];
                        if constexpr ((LOOP & 1) == 0) v += bb[e];
                        if constexpr ((EPI & GM_EPI_GELU) != 0) v = gm_gelu(v);
                        h[e] = (_Float16)v;
                    }
                    *(half4*)(ot + r31 * S::RS + (32 * i + 8 * q + 4 * g) * 2) = h;
                }
            if constexpr ((EPI & GM_EPI_RESID) != 0 && (LOOP & 2) != 0) {
                if (j == 0) {
                    GM_WAIT_VM0_SEEN();  // the residual rows (and the next tile's first K-tile, requested before them) are in
#ifndef LM_EMULATED_DEVICE
#pragma unroll
                    for (int j2 = 0; j2 < S::TT; ++j2)
#pragma unroll
                        for (int it = 0; it < NIT; ++it) asm volatile("" : "+v"(rr[j2][it]));  // (opaque from here on: no load may sink below this point into a store branch)
#endif
                }
            }
            LM_WAVE_SYNC();  // the tile is read back by the wave that wrote it: no workgroup barrier
            GM_STAMP(tm3);
            if constexpr (VAR == 7) tacc[2] += tm3 - tm2;
            if constexpr ((LOOP & 2) != 0) {  // all rows of the block read back first, then the stores: one LDS wait per block instead of one per row
                half8 y[NIT];
#pragma unroll
                for (int it = 0; it < NIT; ++it) {
                    const unsigned char* src = ot + (it * RPI + lr) * S::RS + lc * 16;  // 8-byte aligned (RS = 8 mod 16): two ds_read_b64
                    const half4 lo = *(const half4*)src, hi = *(const half4*)(src + 8);
                    y[it] = half8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                }
#pragma unroll
                for (int it = 0; it < NIT; ++it) {
                    const int tok = t0_cur + wt * S::TT * 32 + 32 * j + it * RPI + lr;
                    if constexpr ((EPI & GM_EPI_RESID) != 0) {
                        const half8 rv = __builtin_bit_cast(half8, rr[j][it]);
#pragma unroll
                        for (int e = 0; e < 8; ++e) y[it][e] = (_Float16)((float)y[it][e] + (float)rv[e]);
                    }
                    if (tok < T && col < N) *(half8*)((_Float16*)out + (int64_t)tok * N + col) = y[it];
                }
            } else {
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int row = it * RPI + lr;
                const int tok = t0_cur + wt * S::TT * 32 + 32 * j + row;
                const unsigned char* src = ot + row * S::RS + lc * 16;  // 8-byte aligned (RS = 8 mod 16): two ds_read_b64
                const half4 lo = *(const half4*)src, hi = *(const half4*)(src + 8);
                half8 y = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                if (tok < T && col < N) {
                    if constexpr ((EPI & GM_EPI_RESID) != 0) {
                        const half8 rv = *(const half8*)((const _Float16*)resid + (int64_t)tok * N + col);
#pragma unroll
                        for (int e = 0; e < 8; ++e) y[e] = (_Float16)((float)y[e] + (float)rv[e]);
                    }
                    *(half8*)((_Float16*)out + (int64_t)tok * N + col) = y;
                }
            }
            }
            LM_WAVE_SYNC();  // the read-back is done before the next column block overwrites the tile
            GM_STAMP(tm2);
            if constexpr (VAR == 7) tacc[3] += tm2 - tm3;
        }
        if constexpr (VAR == 7) {
            tacc[1] += tm1 - tm0;
            ntile += 1;
        }
        if (!more) break;
        ti = ti_next;
    }
#if defined(LM_DIAG) && !defined(LM_EMULATED_DEVICE)
    if constexpr (VAR == 7) {
        if (tid == 0) {
            unsigned long long* dbg = (unsigned long long*)resid;
            atomicAdd(dbg + 0, ts1 - ts0);
            atomicAdd(dbg + 1, tacc[1]);
            atomicAdd(dbg + 2, tacc[2]);
            atomicAdd(dbg + 3, tacc[3]);
            atomicAdd(dbg + 5, ntile);
            atomicAdd(dbg + 6, 1ull);
        }
    }
#endif
}

template <class S, int EPI, int VAR, int LOOP>
static int gemm_launch_var(const void* d_x, const void* d_w, const float* d_bias, const void* d_resid, void* d_out, int64_t tokens, int32_t n_out,
                           int32_t k_in, hipStream_t st) {
    const int64_t rbs = (tokens + S::BM - 1) / S::BM;
    const int64_t per_xcd = (int64_t)((n_out + S::BN - 1) / S::BN) * ((rbs + 7) / 8);  // tiles of the fullest XCD
    if (8 * per_xcd > 0x7fffffff) LM_FAIL(LM_EINVAL, "lm_gemm_f16: too many tiles for one launch");
    // persistent grid: one workgroup per CU (32 per XCD on the MI355X), fewer when there are fewer tiles
#ifdef LM_EMULATED_DEVICE
    static const int cus_per_xcd = 1;  // host emulation: tiny grids, so that every test walks multi-tile lists
#else
    static int cus_by_dev[32] = {};  // per device: a process may search on more than one GPU
    int dev_now = 0;
    if (hipGetDevice(&dev_now) != hipSuccess || dev_now < 0) dev_now = 0;
    int& cus_cached = cus_by_dev[dev_now & 31];
    if (cus_cached == 0) {
        int cus = 0;
        cus_cached = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev_now) == hipSuccess && cus >= 8 ? cus : 256;
    }
    const int cus_per_xcd = cus_cached / 8;
#endif
    int64_t slots = std::min<int64_t>(per_xcd, cus_per_xcd * (S::LDS <= 80 * 1024 ? 2 : 1));
#ifdef LM_DIAG
    static const bool one_per_tile = [] { const char* v = getenv("LEANN_MI355X_GEMM_GRID"); return v && !strcmp(v, "tiles"); }();
    if (one_per_tile) slots = per_xcd;
#endif
    static DynLdsAttr attr;
    LM_HIP(ensure_dyn_lds(attr, (const void*)k_gemm_f16<S, EPI, VAR, LOOP>, S::LDS));
    hipLaunchKernelGGL((k_gemm_f16<S, EPI, VAR, LOOP>), dim3((unsigned)(8 * slots)), dim3(S::THREADS), S::LDS, st, (const __half*)d_x, (const __half*)d_w, d_bias,
                       (const __half*)d_resid, (__half*)d_out, (int)tokens, n_out, k_in);
    LM_HIP(hipGetLastError());
    return LM_OK;
}

template <class S, int EPI>
static int gemm_launch(const void* d_x, const void* d_w, const float* d_bias, const void* d_resid, void* d_out, int64_t tokens, int32_t n_out,
                       int32_t k_in, hipStream_t st) {
#ifdef LM_DIAG
    static const int var = [] { const char* v = getenv("LEANN_MI355X_GEMM_VARIANT"); return v ? atoi(v) : GM_VAR_DEFAULT; }();
    if (var == 7) return gemm_launch_var<S, EPI, 7, 0>(d_x, d_w, d_bias, d_resid, d_out, tokens, n_out, k_in, st);
#endif
    // which form of the tile loop an epilogue runs (GM_LOOP_FOR above): measured, profiles/r6_kbench_gemm_f16_forms_*.jsonl
    return gemm_launch_var<S, EPI, GM_VAR_DEFAULT, GM_LOOP_FOR(EPI)>(d_x, d_w, d_bias, d_resid, d_out, tokens, n_out, k_in, st);
}

}  // namespace lm

extern "C" int lm_gemm_f16(const void* d_x, const void* d_w, const float* d_bias, const void* d_residual, int32_t epilogue, int32_t n_out,
                           int32_t k_in, void* d_out, int64_t tokens, void* stream) {
    using namespace lm;
    if (tokens == 0) return LM_OK;
    if (!d_x || !d_w || !d_bias || !d_out || tokens < 0 || tokens > 0x7fffffff) LM_FAIL(LM_EINVAL, "bad linear arguments");
    if (n_out <= 0 || n_out % 128 || k_in <= 0 || k_in % 128) LM_FAIL(LM_EINVAL, "lm_gemm_f16: n_out and k_in must be positive multiples of 128");
    if (epilogue < 0 || epilogue > 3) LM_FAIL(LM_EINVAL, "lm_gemm_f16: epilogue is a combination of 1 (GELU) and 2 (+ residual)");
    if ((epilogue & GM_EPI_RESID) && !d_residual) LM_FAIL(LM_EINVAL, "lm_gemm_f16: residual epilogue without a residual");
    if ((uint64_t)n_out * (uint64_t)k_in * 2 >= (1ull << 32)) LM_FAIL(LM_EINVAL, "lm_gemm_f16: a weight matrix of 4 GiB or more");
    hipStream_t st = (hipStream_t)stream;
#define GM_GO(S)                                                                                                             \
    switch (epilogue) {                                                                                                      \
        case 0: return gemm_launch<S, 0>(d_x, d_w, d_bias, d_residual, d_out, tokens, n_out, k_in, st);                       \
        case 1: return gemm_launch<S, 1>(d_x, d_w, d_bias, d_residual, d_out, tokens, n_out, k_in, st);                       \
        case 2: return gemm_launch<S, 2>(d_x, d_w, d_bias, d_residual, d_out, tokens, n_out, k_in, st);                       \
        default: return gemm_launch<S, 3>(d_x, d_w, d_bias, d_residual, d_out, tokens, n_out, k_in, st);                      \
    }
    // DMA sources are addressed as wave-uniform base + 32-bit per-lane byte offset: an x operand of 4 GiB or more (1M tokens x 3072
    // features is 6.4 GB) goes as several launches over token ranges of < 4 GiB each (multiples of 256 tokens)
    const int64_t max_rows = (int64_t)(((1ull << 32) - 1) / ((uint64_t)k_in * 2)) / 256 * 256;
    if (tokens > max_rows) {
        for (int64_t r0 = 0; r0 < tokens; r0 += max_rows) {
            const int64_t nr = std::min<int64_t>(max_rows, tokens - r0);
            const int rc = lm_gemm_f16((const unsigned char*)d_x + (size_t)r0 * k_in * 2, d_w, d_bias,
                                       d_residual ? (const unsigned char*)d_residual + (size_t)r0 * n_out * 2 : nullptr, epilogue, n_out, k_in,
                                       (unsigned char*)d_out + (size_t)r0 * n_out * 2, nr, stream);
            if (rc) return rc;
        }
        return LM_OK;
    }
    // (the timing pair sits BEHIND the split above: until round 5 it was opened in front of it, so a split launch -- fc2 of a 768-wide model on more
    // than 699,050 tokens -- was booked twice, once whole and once per part: round 4's C5 line summed more kernel time than its timed region held)
    KtScope kt(LM_KT_GEMM_F16, stream, 2.0 * (double)tokens * n_out * k_in);
    // 256-wide tiles also when the last column tile is half empty, as long as that wastes <= 1/8 of the matrix-pipe work (N >= 896):
    // the 128 x 128 shape is L2-bandwidth bound at a third of the big shape's rate (QKV of the 384-wide models: N = 1152)
    const bool big = tokens > 128 && (n_out % 256 == 0 || n_out >= 896);
    if (big) {
        GM_GO(GemmBig)
    }
    GM_GO(GemmSmall)
#undef GM_GO
}
