// simpool (eval): all-pairs  score[q, v] = max_{l < len_v} < q_hat[q], g_hat[v, l] >  for 1-2 branches
// and their weighted fusion.  Replaces DLDKD.get_sim_scores (reference method/model.py:307-329) as
// driven by compute_query2ctx_info (method/eval.py:200-208) and the fusion at method/eval.py:254.
//
// MI355X design (DESIGN.md section "K1"):
//   * The contraction is [clips x 384] . [384 x queries] on bf16 MFMA 16x16x32 with CLIPS as the MFMA
//     row index and QUERIES on the lanes, so the fp32 result has one query per lane and the clip rows in
//     the accumulator registers: the key-clip max-pool is in-register v_max plus two cross-lane swaps.
//     The (Nq, L, Nv) clip tensor of the reference is never materialised.
//   * GALLERY-STATIONARY IN REGISTERS: a wave owns one (video, branch): all 128 clips x 384 dims
//     (96 KiB) live in 384 of the wave's 512 registers for the wave's whole life; one wave per SIMD,
//     4 videos per CU.  The gallery is read from HBM exactly once.
//   * The queries are pre-packed into MFMA B-fragment order and streamed L2 -> LDS by LDS-DMA in 24 KiB
//     tiles (32 queries) shared by the workgroup's 4 waves, 3-slot ring, one barrier per tile.
//     Each B fragment read from LDS feeds up to 8 MFMAs (0.125 KiB of LDS read per MFMA).
//   * ragged videos: only ceil(len/16) row tiles are computed (wave-uniform template dispatch); videos
//     are visited in a caller-given order (descending length balances the 4 waves of a workgroup).
//   * QUERY SPLIT: the grid is [query range s][branch][group of 4 videos]; a workgroup streams only the
//     query tiles of its range.  Small galleries (one rank's shard of ActivityNet: 615 videos = 308
//     workgroups on 256 CUs) then still fill the chip: the host picks the split from a round-count model.
//     Ranges are dispatched in order (range-major block index) and, on request, every workgroup bumps a
//     per-range arrival counter after an agent-scope release of its scores, so a consumer stream
//     (hipStreamWaitValue32) can finish / all-gather range s while ranges s+1.. are still being scored.
//   (The 32x32x16 scorer v1, the half-video scorer v3 and the row-stream scorer v4 of round 1 were measured
//    dead ends - profiles/r01/ablation_simpool_v2.md - and left the product in round 2; git 952aeba has them.)
#include <stdlib.h>

#include <type_traits>

#include "common.hpp"

#ifndef DLDKD_FIN_SWZ
#define DLDKD_FIN_SWZ 1
#endif
#ifndef K1_ROT
#define K1_ROT 1     // query tiles visited in an order rotated by the chip-wide clock (see score_stream16): A/B switch, tools/r05_ab_k1_rot.sh
#endif
#ifndef K1_NT
#define K1_NT 0      // A/B switch (tools/r05_ab_k1_nt.sh): 1 = non-temporal plane stores, 2 = non-temporal gallery loads, 3 = both
#endif
namespace dldkd {

constexpr int kQTile = 32;                     // queries per LDS tile (two 16-query sub-tiles)
constexpr int kQTileBytes = kQTile * kHidden * 2;   // 24 KiB: [2 sub-tiles][12 k-steps][64 lanes][8 bf16]
constexpr int kRowBf16x8 = kHidden / 8;        // 48 16-byte chunks per gallery row

__host__ __device__ inline int round_up(int x, int m) { return (x + m - 1) / m * m; }

// ----------------------------------------------------------------------------------------------
// packers (fp32 -> normalised bf16).  Tiny, bandwidth-bound, vectorised 16-byte stores.
// ----------------------------------------------------------------------------------------------
// Packed query layout: [tile = q/32][sub = (q%32)/16][k-step ks of 32][lane = 16*kg + q%16][8 bf16 = k 32ks + 8kg ..+8]
// = the B operand of mfma_f32_16x16x32_bf16 (cdna_hip_programming.md section 3 lane maps).
__global__ __launch_bounds__(256) void pack_queries_kernel(const float* __restrict__ q, int nq, int nq_pad,
                                                           int normalize, bf16x8* __restrict__ out,
                                                           float* __restrict__ bad) {
    const int lane = threadIdx.x & 63;
    const int qi = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (qi >= nq_pad) return;
    float v[8];
    float ss = 0.f;
    const bool act = lane < kRowBf16x8 && qi < nq;
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = 0.f;
    if (act) {
        const f32x4* src = reinterpret_cast<const f32x4*>(q + (size_t)qi * kHidden + lane * 8);
        f32x4 a = src[0], b = src[1];
#pragma unroll
        for (int j = 0; j < 4; ++j) { v[j] = a[j]; v[4 + j] = b[j]; }
#pragma unroll
        for (int j = 0; j < 8; ++j) ss += v[j] * v[j];
    }
    float scale = 1.f;
    ss = wave_sum(ss);
    if (normalize) scale = 1.f / fmaxf(sqrtf(ss), 1e-12f);   // F.normalize eps, model.py:318
    // a query vector with a NaN / Inf component (a diverged model): the scorer's v_max pool would DROP the NaN products and
    // hand every video the same sentinel score - a tie that ranks the ground truth first.  Flag it instead; the finish /
    // rank kernels turn a flagged query's scores into NaN, which ranks last (rank.hip NaN policy).
    if (bad != nullptr && lane == 0 && qi < nq && !(ss < INFINITY)) bad[qi] = 1.f;
    if (lane < kRowBf16x8) {
        bf16x8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = (short)f32_to_bf16_bits(v[j] * scale);
        const int ks = lane >> 2, kg = lane & 3, sub = (qi >> 4) & 1;
        out[(((size_t)(qi >> 5) * 2 + sub) * (kHidden / 32) + ks) * 64 + kg * 16 + (qi & 15)] = o;
    }
}

// Gallery blob: row-major bf16 [nv][Lp][384], Lp = round_up(L, 32).  Rows l >= len_v inside the video's last 16-row tile
// REPLICATE its last valid clip (their scores equal a real clip's, so the scorer's max-pool needs no padding mask);
// rows beyond that tile are zero.
// `out` points at the first video of this call; L is the row count of the SOURCE (g, mask), Lp the destination's.
__global__ __launch_bounds__(256) void pack_gallery_kernel(const float* __restrict__ g, const float* __restrict__ mask,
                                                           int nv, int L, int Lp, int normalize,
                                                           bf16x8* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= (long)nv * Lp) return;
    const int v = (int)(row / Lp), l = (int)(row % Lp);
    int len = L;                                  // masks are prefix masks (data_provider.py:81-84)
    if (mask != nullptr) {
        float c = 0.f;
        for (int k = lane; k < L; k += 64) c += mask[(size_t)v * L + k] > 0.f ? 1.f : 0.f;
        len = (int)wave_sum(c);
    }
    const int ls = l < len ? l : (len > 0 && l < ((len + 15) & ~15) ? len - 1 : -1);   // source clip, -1 = zero row
    float x[8];
    float ss = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) x[j] = 0.f;
    if (ls >= 0 && lane < kRowBf16x8) {
        const f32x4* src = reinterpret_cast<const f32x4*>(g + ((size_t)v * L + ls) * kHidden + lane * 8);
        f32x4 a = src[0], b = src[1];
#pragma unroll
        for (int j = 0; j < 4; ++j) { x[j] = a[j]; x[4 + j] = b[j]; }
#pragma unroll
        for (int j = 0; j < 8; ++j) ss += x[j] * x[j];
    }
    float scale = 1.f;
    if (normalize) {
        ss = wave_sum(ss);
        scale = 1.f / fmaxf(sqrtf(ss), 1e-12f);   // model.py:319
    }
    if (lane < kRowBf16x8) {
        bf16x8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = (short)f32_to_bf16_bits(x[j] * scale);
        out[(size_t)row * kRowBf16x8 + lane] = o;
    }
}

// lens[v] = number of mask entries > 0 (compute_kl_loss counts them the same way, model.py:192;
// masks are prefix masks, data_provider.py:81-84).
__global__ __launch_bounds__(256) void mask_lens_kernel(const float* __restrict__ mask, int nv, int L,
                                                        int32_t* __restrict__ lens) {
    const int lane = threadIdx.x & 63;
    const int v = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (v >= nv) return;
    float c = 0.f;
    for (int l = lane; l < L; l += 64) c += (mask == nullptr || mask[(size_t)v * L + l] > 0.f) ? 1.f : 0.f;
    c = wave_sum(c);
    if (lane == 0) lens[v] = (int32_t)c;
}

// Visiting order of the scorer: videos by DESCENDING length, equal lengths in index order (a stable counting sort over the
// 129 possible lengths; one workgroup: the gallery has ~2e4 videos).  order[pos] = video, inv[video] = pos.
// (torch.argsort(stable, descending) took 22 ms for 21,793 lengths - as long as encoding the whole gallery.)
__global__ __launch_bounds__(1024) void order_by_len_desc_kernel(const int32_t* __restrict__ lens, int nv,
                                                                   int32_t* __restrict__ order, int32_t* __restrict__ inv) {
    constexpr int NB = DLDKD_MAX_CLIPS + 1;
    __shared__ int base[NB + 3];
    __shared__ int wcnt[16][NB + 3];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < NB; i += 1024) base[i] = 0;
    __syncthreads();
    for (int v = tid; v < nv; v += 1024) atomicAdd(&base[min(max(lens[v], 0), NB - 1)], 1);
    __syncthreads();
    if (tid == 0) {                                  // exclusive prefix, longest first
        int run = 0;
        for (int l = NB - 1; l >= 0; --l) { const int c = base[l]; base[l] = run; run += c; }
    }
    __syncthreads();
    for (int c0 = 0; c0 < nv; c0 += 1024) {
        for (int i = tid; i < 16 * (NB + 3); i += 1024) (&wcnt[0][0])[i] = 0;
        __syncthreads();
        const int v = c0 + tid;
        const int l = v < nv ? min(max(lens[v], 0), NB - 1) : -1;
        int rank = 0;
        unsigned long long todo = __ballot(l >= 0);
        while (todo) {                               // one round per distinct length in the wave
            const int lead = __ffsll((long long)todo) - 1;
            const int L = __shfl(l, lead);
            const unsigned long long m = __ballot(l == L);
            if (l == L) rank = __popcll(m & ((1ull << lane) - 1ull));
            if (lane == lead) wcnt[wave][L] = __popcll(m);
            todo &= ~m;
        }
        __syncthreads();
        if (l >= 0) {
            int off = base[l];
            for (int w = 0; w < wave; ++w) off += wcnt[w][l];
            order[off + rank] = v;
            inv[v] = off + rank;
        }
        __syncthreads();
        if (tid < NB) {
            int add = 0;
            for (int w = 0; w < 16; ++w) add += wcnt[w][tid];
            base[tid] += add;
        }
        __syncthreads();
    }
}

// ----------------------------------------------------------------------------------------------
// the scorer
// ----------------------------------------------------------------------------------------------
struct SimpoolEvalArgs {
    const bf16x8* q[2];      // packed queries per branch
    const bf16x8* g[2];      // gallery blobs per branch
    const int32_t* lens;     // [nv]
    const int32_t* order;    // [nv] visiting order (sorted position -> video id)
    float* part;             // [n_branches][nv (sorted position)][nq_pad] partial pooled scores
    int32_t* done;           // [n_qsplit] arrival counters (one increment per workgroup of the range) or null
    int nq_pad, nv, Lp, n_qtiles, n_groups;
    int n_wg0;               // workgroups per query range = n_groups * n_branches
    int tiles_per_range;     // query tiles of every range but the last
    int ablate;              // diagnostic builds only
};

// ----------------------------------------------------------------------------------------------
// mfma_f32_16x16x32_bf16, gallery-stationary:
//   * 16-clip row tiles: ragged videos waste < 16 padded rows;
//   * a 16-query sub-tile needs only 4 accumulator registers per row tile, so TWO accumulator sets fit:
//     the max-pool VALU of sub-tile i is issued in slices between the MFMAs of sub-tile i+1;
//   * 3-slot LDS ring: tile t+1 is already visible while tile t is computed, so the B-fragment
//     prefetch ring runs across tile boundaries and the only per-tile cost left is the barrier itself.
// ----------------------------------------------------------------------------------------------
constexpr int kKSteps16 = kHidden / 32;   // 12
constexpr int kRing = 3;
#ifndef K1_GLDS_OFF
#define K1_GLDS_OFF 1
#endif

__device__ __forceinline__ float xor16_max(float m) {
    const unsigned u = __builtin_bit_cast(unsigned, m);
    auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);   // rows 1<->0', 3<->2'
    return fmaxf(__builtin_bit_cast(float, (unsigned)r[0]), __builtin_bit_cast(float, (unsigned)r[1]));
}
__device__ __forceinline__ float xor32_max(float m) {
    const unsigned u = __builtin_bit_cast(unsigned, m);
    auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);   // upper half <-> lower half'
    return fmaxf(__builtin_bit_cast(float, (unsigned)r[0]), __builtin_bit_cast(float, (unsigned)r[1]));
}

// lanes 0..31: max(a[l], a[l + 32]); lanes 32..63: max(b[l - 32], b[l])  - the xor-32 step of two reductions in one swap
__device__ __forceinline__ float halves_max2(float a, float b) {
    auto r = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, a), __builtin_bit_cast(unsigned, b), false, false);
    return fmaxf(__builtin_bit_cast(float, (unsigned)r[0]), __builtin_bit_cast(float, (unsigned)r[1]));
}

// LDS fragment read the compiler does not track (cdna_hip_programming.md 5.7 form (iii)): hipcc turned the
// 4-deep B ring into {issue read, s_waitcnt lgkmcnt(0)} pairs, i.e. a full LDS round trip every few k-steps
// (22 % of wave cycles parked, PMC SQ_WAIT_ANY).  With the reads in asm the waits are hand-counted: exactly
// one read is issued per k-step, so "all but the 3 newest" = the fragment this k-step consumes.
__device__ __forceinline__ void lds_read_frag(bf16x8& dst, uint32_t lds_addr, int byte_off) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(lds_addr), "i"(0) : "memory");
    (void)byte_off;
}
template <int OFF>
__device__ __forceinline__ void lds_read_frag_off(bf16x8& dst, uint32_t lds_addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(lds_addr), "i"(OFF) : "memory");
}

// Rotation of a workgroup's query-tile order (K1_ROT): workgroups are dispatched as CUs free up, i.e. at times spread over a
// workgroup's whole life, and each streams its range's tiles from the first one - at any moment the 32 workgroups of an XCD are
// spread over the whole 8.4-MB query blob, twice its 4-MiB L2: 6.8 GB of query tiles per launch came from beyond L2
// (profiles/r04/pmc_simpool).  Starting at tile (clock / time per tile) mod T instead puts every workgroup of the chip on the same
// tile at the same time, whenever it was dispatched.  s_memrealtime is the chip-wide constant 100-MHz counter; a tile of nrt row tiles
// takes ~27 ticks per row tile (2.15 us at 8).  One lane reads the clock, the workgroup shares it through LDS.
__device__ __forceinline__ int pick_rot(int T, int nrt_hint) {
#if K1_ROT
    __shared__ int s_rot;
    if (threadIdx.x == 0) {
        const unsigned long long now = __builtin_amdgcn_s_memrealtime();     // (NOT s_memtime: that one counts core clocks, per XCD)
        const unsigned per = (K1_ROT == 3 ? 23u : 27u) * (unsigned)(nrt_hint < 1 ? 1 : nrt_hint);
        // ... minus a lag of 0..31 tiles by the workgroup's index within its XCD (blockIdx % 8 = XCD): with every workgroup of an
        // XCD on the SAME tile at the same instant all 32 miss together and the L2 fetches the tile 32 times (measured: 31 GB per
        // launch instead of 9); staggered, the first one misses and the others find the tile in L2 a few microseconds later
        const unsigned lag = K1_ROT == 2 ? 0u : (blockIdx.x >> 3) & 31u;
        s_rot = T > 1 ? (int)((now / per + (unsigned long long)T - lag % (unsigned)T) % (unsigned long long)T) : 0;
    }
    __syncthreads();
    return __builtin_amdgcn_readfirstlane(s_rot);
#else
    (void)T; (void)nrt_hint;
    return 0;
#endif
}

// ABL (diagnostic builds only, DLDKD_SIMPOOL_ABLATE): 1 = no max-pool / stores, 2 = no in-loop LDS-DMA staging,
// 4 = no per-tile barrier, 8 = no B-fragment LDS reads (ring registers reused).  Results are wrong by design;
// only the timing matters (cdna_hip_programming.md section 7, "Ablate").
template <int NRT, int ABL = 0>
__device__ __forceinline__ void score_stream16(const bf16x8 (&a)[8][kKSteps16], const SimpoolEvalArgs& p, int branch,
                                               int vs, int t0, int T, char* smem, int rot = 0) {
    // this workgroup streams query tiles [t0, t0 + T) of the packed blob
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const char* qsrc = reinterpret_cast<const char*>(p.q[branch]) + (size_t)t0 * kQTileBytes;

    auto stage = [&](int t, int slot) {
        char* dst = smem + slot * kQTileBytes;
        const char* src = qsrc + (size_t)t * kQTileBytes;
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const int piece = wave * 6 + i;
            glds16(src + piece * 1024 + lane * 16, dst + piece * 1024);
        }
    };

    // The range's tiles are visited in an order ROTATED by `rot` (the same for the four waves): iteration i works on tile tau(i).
    // Every tile's scores are stored where they belong, so any rotation gives the same planes; the kernels pick rot from the
    // chip-wide clock so that workgroups dispatched at different times stream the SAME query tiles at the same time (K1_ROT).
    auto tau = [&](int i) { const int x = i + rot; return x >= T ? x - T : x; };
    stage(tau(0), 0);
    if (T > 1) stage(tau(1), 1);

    if constexpr (NRT == 0) {   // padding wave: staging + barriers only
        int slot2 = 2;
        for (int t = 0; t < T; ++t) {
            __syncthreads();
            if (t + 2 < T) stage(tau(t + 2), slot2);
            slot2 = slot2 == kRing - 1 ? 0 : slot2 + 1;
        }
    } else {
        // no padding mask: rows of the last tile beyond `len` replicate the video's last valid clip (pack_gallery_kernel),
        // so they can never change the maximum (same-box A/B against the masked form: 19.07 -> 18.94 ms)
        float* outp = p.part + ((size_t)branch * p.nv + vs) * p.nq_pad + (size_t)t0 * kQTile + (lane & 15);
        constexpr int V = NRT * 4;              // accumulator values per lane per sub-tile
        constexpr int kFoldSteps = 9;           // k-steps 0..8 fold the values, 9/10 cross lanes, 11 stores
        constexpr int kPer = (V + kFoldSteps - 1) / kFoldSteps;
        constexpr int kPF = 4;                  // B-fragment ring depth (divides 12); lgkmcnt(3) below = kPF - 1

        f32x4 accA[NRT], accB[NRT];
#pragma unroll
        for (int rt = 0; rt < NRT; ++rt) accB[rt] = f32x4{0.f, 0.f, 0.f, 0.f};
        bf16x8 b[kPF];
        float m = 0.f;

        // one 16-query sub-tile: MFMAs into `cur`, max-pool of `prev` sliced between the k-steps
        auto subtile = [&](auto sub_c, f32x4 (&cur)[NRT], const f32x4 (&prev)[NRT], uint32_t cbase, uint32_t nbase,
                           float* prev_out, const char* st_src, char* st_dst) {
            constexpr int S = decltype(sub_c)::value;
            auto step = [&](auto ks_c) {
                constexpr int ks = decltype(ks_c)::value;
                // the ring holds kPF reads in flight, issued one per k-step: all but the 3 newest have landed
                // (the wait does not name b[]: the MFMAs below are compiler-visible and are kept behind it by the scheduling fence
                // on the next line; naming the fragment - "+v"(b[ks % kPF]) - measured +0.5 % on the same box, 19.33-19.38 vs
                // 19.21-19.26 ms)
                if constexpr (!(ABL & 8)) asm volatile("s_waitcnt lgkmcnt(3)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int rt = 0; rt < NRT; ++rt) {
                    if (ks == 0) {
                        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
                        cur[rt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[rt][ks], b[ks % kPF], z, 0, 0, 0);
                    } else {
                        cur[rt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[rt][ks], b[ks % kPF], cur[rt], 0, 0, 0);
                    }
                }
                // (no fence here: the MFMAs, the ring read and the pool slice below form one scheduling region, see the
                // sched_group_barrier sequence at its end)
                // keep the B ring kPF k-steps ahead, across the sub-tile and the tile boundary (past the last
                // tile this reads a stale ring slot: harmless, never consumed)
                constexpr int idx = S * kKSteps16 + kPF + ks;
                if constexpr (ABL & 8) { asm volatile("" : "+v"(b[ks % kPF])); }
                else if constexpr (idx < 2 * kKSteps16) lds_read_frag_off<idx * 1024>(b[ks % kPF], cbase);
                else lds_read_frag_off<(idx - 2 * kKSteps16) * 1024>(b[ks % kPF], nbase);
                // slice of the previous sub-tile's key-clip max-pool
                if constexpr (ABL & 1) {
                    if constexpr (ks == 11) {   // keep every accumulator (hence every MFMA) live: rule 17
#pragma unroll
                        for (int rt = 0; rt < NRT; ++rt) asm volatile("" :: "v"(prev[rt]));
                    }
                } else if constexpr (ks < kFoldSteps) {
                    if (ks == 0) m = -3.0e38f;
#pragma unroll
                    for (int i = ks * kPer; i < (ks + 1) * kPer && i < V; ++i) {
                        float x = prev[i >> 2][i & 3];
                        m = fmaxf(m, x);
                    }
                } else if constexpr (ks == 9) {
                    m = xor16_max(m);
                } else if constexpr (ks == 10) {
                    m = xor32_max(m);
                } else {
                    // after the two swaps all four 16-lane groups hold the same maximum and prev_out is indexed by lane & 15:
                    // every lane stores (four identical writes per address) - no exec masking, no branch around the store
                    // (same-box A/B: 19.59 -> 19.38 ms)
                    if constexpr (K1_NT & 1) __builtin_nontemporal_store(m, prev_out); else *prev_out = m;
                }
                // one 1-KiB LDS-DMA piece of tile t+2 per k-step of sub-tile 0 (6 per wave) instead of all six at the top
                // of the tile, where the MFMA pipe waited for their issue (same-box A/B: 20.27-20.42 -> 20.03-20.06 ms).
                // They stay older than the tile's two result stores, so the vmcnt(2) before the next barrier still means
                // "this DMA has landed".
                if constexpr (S == 0 && ks < 6 && !(ABL & 2)) {
                    // (one base and one M0 per four pieces - the immediate reaches 4095: the scorer is issue-bound)
                    if constexpr (K1_GLDS_OFF) glds16_off<(ks & 3) * 1024>(st_src + (ks >> 2) * 4096, st_dst + (ks >> 2) * 4096);
                    else glds16(st_src + ks * 1024, st_dst + ks * 1024);
                }
                // Place this k-step's pool slice (VALU) in the shadow of its MFMAs: groups of {2 MFMA, 2 VALU}.  With a hard
                // fence between the 8 MFMAs and the slice (first version) the VALU issued after the last MFMA and only
                // its 16-cycle shadow was free.  Same-box A/B at C2: 20.55-20.74 ms fenced, 20.07-20.20 ms {2,2};
                // {1 MFMA, 1 VALU} x 8 gives nothing (20.70), {2,1} x 4 is equal to {2,2} (20.19).
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            };
            step(std::integral_constant<int, 0>{}); step(std::integral_constant<int, 1>{});
            step(std::integral_constant<int, 2>{}); step(std::integral_constant<int, 3>{});
            step(std::integral_constant<int, 4>{}); step(std::integral_constant<int, 5>{});
            step(std::integral_constant<int, 6>{}); step(std::integral_constant<int, 7>{});
            step(std::integral_constant<int, 8>{}); step(std::integral_constant<int, 9>{});
            step(std::integral_constant<int, 10>{}); step(std::integral_constant<int, 11>{});
        };

        int slot = 0, slot2 = 2;
        for (int t = 0; t < T; ++t) {
            // Tile t+1 (and t) must have landed before the barrier; every wave is then done with tile t-1.
            // VMEM ops retire in order: at t >= 1 the only ops younger than tile t+1's DMA are this wave's two
            // result stores of iteration t-1, so vmcnt(2) waits for the DMA but not for the stores (a plain
            // __syncthreads() = vmcnt(0) lgkmcnt(0) also waits for those stores and drains the B-fragment ring).
            // Measured alternatives that did NOT help (kept out): staging the tile through registers
            // (global_load + ds_write spread over k-steps) instead of LDS-DMA: 21.1 ms vs 20.4 ms at C2.
            if (t == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else if constexpr (!(ABL & 3)) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (t == 0 || !(ABL & 4)) __builtin_amdgcn_s_barrier();
            // staging of tile t+2 is spread over the k-steps of sub-tile 0; past the end it re-stages tile T-1 into a free slot
            const int t2 = tau(t + 2 < T ? t + 2 : T - 1);
            const char* st_src = qsrc + (size_t)t2 * kQTileBytes + (size_t)wave * 6 * 1024 + lane * 16;
            char* st_dst = smem + slot2 * kQTileBytes + wave * 6 * 1024;
            const int nslot = slot == kRing - 1 ? 0 : slot + 1;
            const uint32_t smem_lds = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(smem));
            const uint32_t cbase = smem_lds + slot * kQTileBytes + lane * 16;
            const uint32_t nbase = smem_lds + nslot * kQTileBytes + lane * 16;
            if (t == 0) {
                lds_read_frag_off<0>(b[0], cbase);
                lds_read_frag_off<1024>(b[1], cbase);
                lds_read_frag_off<2048>(b[2], cbase);
                lds_read_frag_off<3072>(b[3], cbase);
                __builtin_amdgcn_sched_barrier(0);
            }
            // t == 0: there is no previous sub-tile; its (garbage) result goes to queries 0..15, which the
            // next sub-tile's store - later in program order, same lanes, same addresses - overwrites.
            // (the previous sub-tile is the second half of tile tau(t - 1); at t == 0 the garbage goes where this tile's first half
            // is stored next)
            const size_t o_cur = (size_t)tau(t) * kQTile;
            subtile(std::integral_constant<int, 0>{}, accA, accB, cbase, nbase, outp + (t > 0 ? (size_t)tau(t - 1) * kQTile + 16 : o_cur), st_src, st_dst);
            subtile(std::integral_constant<int, 1>{}, accB, accA, cbase, nbase, outp + o_cur, st_src, st_dst);
            slot = nslot;
            slot2 = slot2 == kRing - 1 ? 0 : slot2 + 1;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // retire the (unused) tail of the ring
        // drain: max-pool of the very last sub-tile
        m = -3.0e38f;
#pragma unroll
        for (int i = 0; i < V; ++i) {
            float x = accB[i >> 2][i & 3];
            m = fmaxf(m, x);
        }
        m = xor32_max(xor16_max(m));
        if (lane < 16) outp[(size_t)tau(T - 1) * kQTile + 16] = m;
    }
}

__global__ __launch_bounds__(256, 1) void simpool_eval16_kernel(const SimpoolEvalArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // block index = [query range][branch][group]: ranges are dispatched in order; inside a range the resident workgroups are
    // kept on the same query tiles by the clock-rotated tile order (pick_rot), which keeps the tiles L2-hot
    const int range = blockIdx.x / p.n_wg0;
    const int b0 = blockIdx.x - range * p.n_wg0;
    const int branch = b0 / p.n_groups;
    const int vs = (b0 - branch * p.n_groups) * 4 + wave;
    const int t0 = range * p.tiles_per_range;
    const int T = min(p.tiles_per_range, p.n_qtiles - t0);
    int len = 0, v = 0;
    if (vs < p.nv) {
        v = p.order[vs];
        len = p.lens[v];
    }
    len = __builtin_amdgcn_readfirstlane(len);
    const int nrt = (len + 15) >> 4;

    // stationary operand: lane l holds clip (16rt + l%16), k 32ks + 8(l/16) ..+8
    bf16x8 a[8][kKSteps16];
    const bf16x8* gv = p.g[branch] + (size_t)v * p.Lp * kRowBf16x8 + (lane & 15) * kRowBf16x8 + (lane >> 4);
#pragma unroll
    for (int rt = 0; rt < 8; ++rt) {
        if (rt < nrt) {
#pragma unroll
            for (int ks = 0; ks < kKSteps16; ++ks) a[rt][ks] = (K1_NT & 2) ? __builtin_nontemporal_load(gv + (size_t)rt * 16 * kRowBf16x8 + ks * 4) : gv[(size_t)rt * 16 * kRowBf16x8 + ks * 4];
        } else {
#pragma unroll
            for (int ks = 0; ks < kKSteps16; ++ks) a[rt][ks] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
        }
    }
    // Pin register classes: 64 fragments in the accumulator half of the unified file, 32 in arch VGPRs.
    // Without this hipcc allocates all of them as VGPR-class values, "spills" the overflow to AGPRs and
    // copies each back before its MFMA (160 v_accvgpr_mov per 96 MFMAs).  The file is built with
    // -mllvm -amdgpu-mfma-vgpr-form=1 so the MFMA results stay in arch VGPRs, where the max-pool VALU
    // reads them directly (AGPR-form results cost one v_accvgpr_read per value).
#pragma unroll
    for (int rt = 0; rt < 8; ++rt)
#pragma unroll
        for (int ks = 0; ks < kKSteps16; ++ks) {
            if (rt * kKSteps16 + ks < 64) asm volatile("" : "+a"(a[rt][ks]));
            else asm volatile("" : "+v"(a[rt][ks]));
        }

    const int rot = pick_rot(T, nrt);                         // (every wave of the workgroup, padding waves included)
#ifdef DLDKD_DIAG_ABLATE   // `make DIAG=1`: co-compiled variants perturb the shipped one's codegen (rule 19)
    if (p.ablate && nrt == 8) {   // diagnostic timing builds, full-length videos only
        switch (p.ablate) {
            case 1: score_stream16<8, 1>(a, p, branch, vs, t0, T, smem, rot); return;
            case 2: score_stream16<8, 2>(a, p, branch, vs, t0, T, smem, rot); return;
            case 4: score_stream16<8, 4>(a, p, branch, vs, t0, T, smem, rot); return;
            case 8: score_stream16<8, 8>(a, p, branch, vs, t0, T, smem, rot); return;
            case 15: score_stream16<8, 15>(a, p, branch, vs, t0, T, smem, rot); return;
            default: break;
        }
    }
#endif
    switch (nrt) {
        case 8: score_stream16<8>(a, p, branch, vs, t0, T, smem, rot); break;
        case 7: score_stream16<7>(a, p, branch, vs, t0, T, smem, rot); break;
        case 6: score_stream16<6>(a, p, branch, vs, t0, T, smem, rot); break;
        case 5: score_stream16<5>(a, p, branch, vs, t0, T, smem, rot); break;
        case 4: score_stream16<4>(a, p, branch, vs, t0, T, smem, rot); break;
        case 3: score_stream16<3>(a, p, branch, vs, t0, T, smem, rot); break;
        case 2: score_stream16<2>(a, p, branch, vs, t0, T, smem, rot); break;
        case 1: score_stream16<1>(a, p, branch, vs, t0, T, smem, rot); break;
        default:
            score_stream16<0>(a, p, branch, vs, t0, T, smem, rot);
            // a real video with no valid clip: the reference's masked maximum is exactly -1e10 (mask_logits, model.py:444)
            if (vs < p.nv) {
                float* row = p.part + ((size_t)branch * p.nv + vs) * p.nq_pad + (size_t)t0 * kQTile;
                for (int q = lane; q < T * kQTile; q += 64) row[q] = -1e10f;
            }
            break;
    }
    if (p.done != nullptr) {
        // publish this workgroup's scores of the range: every wave drains its stores, the workgroup meets, ONE lane
        // releases at agent scope (write-back of the XCD's L2) and only then bumps the range's arrival counter
        // (cdna_hip_programming.md Guideline 16; the asm wait after the fence is the ROCm 7.2 pitfall-12 guard).
        // A stream parked on the counter (hipStreamWaitValue32) then launches kernels that read the scores from memory.
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __hip_atomic_fetch_add(p.done + range, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

typedef float f32x2 __attribute__((ext_vector_type(2)));

// Two videos per wave (pair waves, simpool_eval16p_kernel): all 128 rows of the wave are in use - rows [0, cA) are video A (cA = its
// length rounded up to 4), rows cA.. video B (its last clip replicated to row 127).  KB = cA / 16 is a template parameter: the
// tiles below KB fold into mA and the tiles above it into mB exactly like the one-video kernel's running maximum (max3 chains, 2 VALU
// per tile); only tile KB can hold both videos, and there a lane group (lane >> 4) owns rows 4 g .. 4 g + 3, all of ONE video
// because the boundary is a multiple of 4: the tile's in-lane maximum goes to both sides through one packed add of the lane's
// (0, -3e38) / (-3e38, 0) routing pair (x + 0 is exact; x - 3e38 loses to every real score).  19 VALU per sub-tile instead of 16.
// The kernel is issue-bound at one wave per SIMD (every VALU instruction shows up in the time: measured, ablation_simpool_ragged.md),
// so the two cross-lane reductions are ONE: permlane32_swap(mA, mB) puts the halves of mA side by side in lanes 0..31 and those of
// mB in lanes 32..63; one max and one xor-16 step later lanes 0..31 hold video A's scores and lanes 32..63 video B's, and a single
// store with a per-lane row pointer writes both.  Instruction for instruction the one-video loop + 3 VALU.
template <int KB>
__device__ __forceinline__ void score_stream16p(const bf16x8 (&a)[8][kKSteps16], const SimpoolEvalArgs& p, int branch,
                                                int posA, int posB, int cA, int t0, int T, char* smem, int rot = 0) {
    constexpr int ABL = 0;
    constexpr int NRT = 8;
    // this workgroup streams query tiles [t0, t0 + T) of the packed blob
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const char* qsrc = reinterpret_cast<const char*>(p.q[branch]) + (size_t)t0 * kQTileBytes;

    auto stage = [&](int t, int slot) {
        char* dst = smem + slot * kQTileBytes;
        const char* src = qsrc + (size_t)t * kQTileBytes;
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const int piece = wave * 6 + i;
            glds16(src + piece * 1024 + lane * 16, dst + piece * 1024);
        }
    };

    // The range's tiles are visited in an order ROTATED by `rot` (the same for the four waves): iteration i works on tile tau(i).
    // Every tile's scores are stored where they belong, so any rotation gives the same planes; the kernels pick rot from the
    // chip-wide clock so that workgroups dispatched at different times stream the SAME query tiles at the same time (K1_ROT).
    auto tau = [&](int i) { const int x = i + rot; return x >= T ? x - T : x; };
    stage(tau(0), 0);
    if (T > 1) stage(tau(1), 1);

    if constexpr (NRT == 0) {   // padding wave: staging + barriers only
        int slot2 = 2;
        for (int t = 0; t < T; ++t) {
            __syncthreads();
            if (t + 2 < T) stage(tau(t + 2), slot2);
            slot2 = slot2 == kRing - 1 ? 0 : slot2 + 1;
        }
    } else {
        // no padding mask: rows of the last tile beyond `len` replicate the video's last valid clip (pack_gallery_kernel),
        // so they can never change the maximum (same-box A/B against the masked form: 19.07 -> 18.94 ms)
        float* outp = p.part + ((size_t)branch * p.nv + (lane < 32 ? posA : posB)) * p.nq_pad + (size_t)t0 * kQTile + (lane & 15);
        const int rowg = 4 * (lane >> 4);
        constexpr int kPF = 4;                  // B-fragment ring depth (divides 12); lgkmcnt(3) below = kPF - 1

        f32x4 accA[NRT], accB[NRT];
#pragma unroll
        for (int rt = 0; rt < NRT; ++rt) accB[rt] = f32x4{0.f, 0.f, 0.f, 0.f};
        bf16x8 b[kPF];
        float m = 0.f, m2 = 0.f;
        const bool first = rowg < cA - 16 * KB;
        f32x2 route = {first ? 0.f : -3.0e38f, first ? -3.0e38f : 0.f};
        asm volatile("" : "+v"(route));

        // one 16-query sub-tile: MFMAs into `cur`, max-pool of `prev` sliced between the k-steps
        auto subtile = [&](auto sub_c, f32x4 (&cur)[NRT], const f32x4 (&prev)[NRT], uint32_t cbase, uint32_t nbase,
                           float* prev_out, const char* st_src, char* st_dst) {
            constexpr int S = decltype(sub_c)::value;
            auto step = [&](auto ks_c) {
                constexpr int ks = decltype(ks_c)::value;
                // the ring holds kPF reads in flight, issued one per k-step: all but the 3 newest have landed
                // (the wait does not name b[]: the MFMAs below are compiler-visible and are kept behind it by the scheduling fence
                // on the next line; naming the fragment - "+v"(b[ks % kPF]) - measured +0.5 % on the same box, 19.33-19.38 vs
                // 19.21-19.26 ms)
                if constexpr (!(ABL & 8)) asm volatile("s_waitcnt lgkmcnt(3)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int rt = 0; rt < NRT; ++rt) {
                    if (ks == 0) {
                        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
                        cur[rt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[rt][ks], b[ks % kPF], z, 0, 0, 0);
                    } else {
                        cur[rt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[rt][ks], b[ks % kPF], cur[rt], 0, 0, 0);
                    }
                }
                // (no fence here: the MFMAs, the ring read and the pool slice below form one scheduling region, see the
                // sched_group_barrier sequence at its end)
                // keep the B ring kPF k-steps ahead, across the sub-tile and the tile boundary (past the last
                // tile this reads a stale ring slot: harmless, never consumed)
                constexpr int idx = S * kKSteps16 + kPF + ks;
                if constexpr (ABL & 8) { asm volatile("" : "+v"(b[ks % kPF])); }
                else if constexpr (idx < 2 * kKSteps16) lds_read_frag_off<idx * 1024>(b[ks % kPF], cbase);
                else lds_read_frag_off<(idx - 2 * kKSteps16) * 1024>(b[ks % kPF], nbase);
                // slice of the previous sub-tile's key-clip max-pool
                if constexpr (ABL & 1) {
                    if constexpr (ks == 11) {   // keep every accumulator (hence every MFMA) live: rule 17
#pragma unroll
                        for (int rt = 0; rt < NRT; ++rt) asm volatile("" :: "v"(prev[rt]));
                    }
                } else if constexpr (ks < 9) {
                    if (ks == 0) { m = -3.0e38f; m2 = -3.0e38f; }
                    if constexpr (ks < KB) {
                        m = fmaxf(fmaxf(m, prev[ks][0]), prev[ks][1]);
                        m = fmaxf(fmaxf(m, prev[ks][2]), prev[ks][3]);
                    } else if constexpr (ks == KB) {
                        const float tm = fmaxf(fmaxf(prev[ks][0], prev[ks][1]), fmaxf(prev[ks][2], prev[ks][3]));
                        const f32x2 r = f32x2{tm, tm} + route;
                        m = fmaxf(m, r[0]);
                        m2 = fmaxf(m2, r[1]);
                    } else if constexpr (ks < NRT) {
                        m2 = fmaxf(fmaxf(m2, prev[ks][0]), prev[ks][1]);
                        m2 = fmaxf(fmaxf(m2, prev[ks][2]), prev[ks][3]);
                    }
                } else if constexpr (ks == 9) {
                    m = halves_max2(m, m2);
                } else if constexpr (ks == 10) {
                    m = xor16_max(m);
                } else {
                    if constexpr (K1_NT & 1) __builtin_nontemporal_store(m, prev_out); else *prev_out = m;
                }
                // one 1-KiB LDS-DMA piece of tile t+2 per k-step of sub-tile 0 (6 per wave) instead of all six at the top
                // of the tile, where the MFMA pipe waited for their issue (same-box A/B: 20.27-20.42 -> 20.03-20.06 ms).
                // They stay older than the tile's two result stores, so the vmcnt(2) before the next barrier still means
                // "this DMA has landed".
                if constexpr (S == 0 && ks < 6 && !(ABL & 2)) {
                    // (one base and one M0 per four pieces - the immediate reaches 4095: the scorer is issue-bound)
                    if constexpr (K1_GLDS_OFF) glds16_off<(ks & 3) * 1024>(st_src + (ks >> 2) * 4096, st_dst + (ks >> 2) * 4096);
                    else glds16(st_src + ks * 1024, st_dst + ks * 1024);
                }
                // Place this k-step's pool slice (VALU) in the shadow of its MFMAs: groups of {2 MFMA, 2 VALU}.  With a hard
                // fence between the 8 MFMAs and the slice (first version) the VALU issued after the last MFMA and only
                // its 16-cycle shadow was free.  Same-box A/B at C2: 20.55-20.74 ms fenced, 20.07-20.20 ms {2,2};
                // {1 MFMA, 1 VALU} x 8 gives nothing (20.70), {2,1} x 4 is equal to {2,2} (20.19).
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            };
            step(std::integral_constant<int, 0>{}); step(std::integral_constant<int, 1>{});
            step(std::integral_constant<int, 2>{}); step(std::integral_constant<int, 3>{});
            step(std::integral_constant<int, 4>{}); step(std::integral_constant<int, 5>{});
            step(std::integral_constant<int, 6>{}); step(std::integral_constant<int, 7>{});
            step(std::integral_constant<int, 8>{}); step(std::integral_constant<int, 9>{});
            step(std::integral_constant<int, 10>{}); step(std::integral_constant<int, 11>{});
        };

        int slot = 0, slot2 = 2;
        for (int t = 0; t < T; ++t) {
            // Tile t+1 (and t) must have landed before the barrier; every wave is then done with tile t-1.
            // VMEM ops retire in order: at t >= 1 the only ops younger than tile t+1's DMA are this wave's two
            // result stores of iteration t-1, so vmcnt(2) waits for the DMA but not for the stores (a plain
            // __syncthreads() = vmcnt(0) lgkmcnt(0) also waits for those stores and drains the B-fragment ring).
            // Measured alternatives that did NOT help (kept out): staging the tile through registers
            // (global_load + ds_write spread over k-steps) instead of LDS-DMA: 21.1 ms vs 20.4 ms at C2.
            if (t == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else if constexpr (!(ABL & 3)) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (t == 0 || !(ABL & 4)) __builtin_amdgcn_s_barrier();
            // staging of tile t+2 is spread over the k-steps of sub-tile 0; past the end it re-stages tile T-1 into a free slot
            const int t2 = tau(t + 2 < T ? t + 2 : T - 1);
            const char* st_src = qsrc + (size_t)t2 * kQTileBytes + (size_t)wave * 6 * 1024 + lane * 16;
            char* st_dst = smem + slot2 * kQTileBytes + wave * 6 * 1024;
            const int nslot = slot == kRing - 1 ? 0 : slot + 1;
            const uint32_t smem_lds = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(smem));
            const uint32_t cbase = smem_lds + slot * kQTileBytes + lane * 16;
            const uint32_t nbase = smem_lds + nslot * kQTileBytes + lane * 16;
            if (t == 0) {
                lds_read_frag_off<0>(b[0], cbase);
                lds_read_frag_off<1024>(b[1], cbase);
                lds_read_frag_off<2048>(b[2], cbase);
                lds_read_frag_off<3072>(b[3], cbase);
                __builtin_amdgcn_sched_barrier(0);
            }
            // t == 0: there is no previous sub-tile; its (garbage) result goes to queries 0..15, which the
            // next sub-tile's store - later in program order, same lanes, same addresses - overwrites.
            // (the previous sub-tile is the second half of tile tau(t - 1); at t == 0 the garbage goes where this tile's first half
            // is stored next)
            const size_t o_cur = (size_t)tau(t) * kQTile;
            subtile(std::integral_constant<int, 0>{}, accA, accB, cbase, nbase, outp + (t > 0 ? (size_t)tau(t - 1) * kQTile + 16 : o_cur), st_src, st_dst);
            subtile(std::integral_constant<int, 1>{}, accB, accA, cbase, nbase, outp + o_cur, st_src, st_dst);
            slot = nslot;
            slot2 = slot2 == kRing - 1 ? 0 : slot2 + 1;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // retire the (unused) tail of the ring
        // drain: max-pool of the very last sub-tile
        m = -3.0e38f;
        m2 = -3.0e38f;
#pragma unroll
        for (int rt = 0; rt < NRT; ++rt) {
            const float tm = fmaxf(fmaxf(accB[rt][0], accB[rt][1]), fmaxf(accB[rt][2], accB[rt][3]));
            if (rt < KB) m = fmaxf(m, tm);
            else if (rt > KB) m2 = fmaxf(m2, tm);
            else { m = fmaxf(m, tm + route[0]); m2 = fmaxf(m2, tm + route[1]); }
        }
        m = xor16_max(halves_max2(m, m2));
        if ((lane & 31) < 16) outp[(size_t)tau(T - 1) * kQTile + 16] = m;
    }
}

struct SimpoolPairArgs {
    SimpoolEvalArgs e;
    const int32_t* pairs;    // [n_waves][2]: sorted positions (posA, posB or -1) of the videos a wave scores
    int n_waves;
};

// K1 with TWO videos per wave where they fill it: a pair wave's 128 register-resident rows
// hold video A and, behind it on a 4-row boundary, video B - chosen by the host (scoring.pair_waves) so that
// 112 < round_up(len A, 4) + len B <= 128.  The ragged TVR gallery (U{24..128} clips) needs ~13.7 k waves instead of 21.8 k and
// ~5 % fewer 16-row MFMA tiles, every one of them amortising the per-sub-tile costs (B-fragment reads, staging, barrier, pooling
// tail) over 8 row tiles.  A wave with one video runs the one-video loop unchanged.  Scores are bit-identical to
// simpool_eval16_kernel's: the same MFMA accumulation per clip row, the same maxima over the same clips.
__global__ __launch_bounds__(256, 1) void simpool_eval16p_kernel(const SimpoolPairArgs pp) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const SimpoolEvalArgs& p = pp.e;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int range = blockIdx.x / p.n_wg0;
    const int b0 = blockIdx.x - range * p.n_wg0;
    const int branch = b0 / p.n_groups;
    const int w = (b0 - branch * p.n_groups) * 4 + wave;
    const int t0 = range * p.tiles_per_range;
    const int T = min(p.tiles_per_range, p.n_qtiles - t0);
    int posA = 0, posB = -1, vA = 0, vB = 0, lenA = 0, lenB = 0;
    if (w < pp.n_waves) {
        posA = pp.pairs[2 * w];
        posB = pp.pairs[2 * w + 1];
        vA = p.order[posA];
        lenA = p.lens[vA];
        if (posB >= 0) { vB = p.order[posB]; lenB = p.lens[vB]; }
    }
    posA = __builtin_amdgcn_readfirstlane(posA); posB = __builtin_amdgcn_readfirstlane(posB);
    lenA = __builtin_amdgcn_readfirstlane(lenA); lenB = __builtin_amdgcn_readfirstlane(lenB);
    const bool pair = posB >= 0 && lenA >= 1 && lenB >= 1 && ((lenA + 3) & ~3) + lenB <= 128;   // (a plan that breaks the rule: A only)
    const int cA = pair ? (lenA + 3) & ~3 : 128;              // rows [0, cA): video A (its last clip replicated past lenA)
    const int nrt = pair ? 8 : (lenA + 15) >> 4;

    // stationary operand: lane l holds row (16 rt + l % 16) of the wave, k 32 ks + 8 (l / 16) .. + 8; a row past its video's
    // length reads the video's last clip (it can tie, never beat, a real clip: no padding mask in the pool)
    bf16x8 a[8][kKSteps16];
#pragma unroll
    for (int rt = 0; rt < 8; ++rt) {
        if (rt < nrt) {
            const int r = 16 * rt + (lane & 15);
            const bool inA = r < cA;
            const int v = inA ? vA : vB;
            const int row = inA ? min(r, max(lenA - 1, 0)) : min(r - cA, lenB - 1);
            const bf16x8* gv = p.g[branch] + ((size_t)v * p.Lp + row) * kRowBf16x8 + (lane >> 4);
#pragma unroll
            for (int ks = 0; ks < kKSteps16; ++ks) a[rt][ks] = (K1_NT & 2) ? __builtin_nontemporal_load(gv + ks * 4) : gv[ks * 4];
        } else {
#pragma unroll
            for (int ks = 0; ks < kKSteps16; ++ks) a[rt][ks] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
        }
    }
#pragma unroll
    for (int rt = 0; rt < 8; ++rt)
#pragma unroll
        for (int ks = 0; ks < kKSteps16; ++ks) {
            if (rt * kKSteps16 + ks < 64) asm volatile("" : "+a"(a[rt][ks]));
            else asm volatile("" : "+v"(a[rt][ks]));
        }
    const int rot = pick_rot(T, pair ? 8 : nrt);              // (every wave of the workgroup, padding waves included)
    switch (pair ? 8 + (cA >> 4) : nrt) {
        case 15: score_stream16p<7>(a, p, branch, posA, posB, cA, t0, T, smem, rot); break;
        case 14: score_stream16p<6>(a, p, branch, posA, posB, cA, t0, T, smem, rot); break;
        case 13: score_stream16p<5>(a, p, branch, posA, posB, cA, t0, T, smem, rot); break;
        case 12: score_stream16p<4>(a, p, branch, posA, posB, cA, t0, T, smem, rot); break;
        case 11: score_stream16p<3>(a, p, branch, posA, posB, cA, t0, T, smem, rot); break;
        case 10: score_stream16p<2>(a, p, branch, posA, posB, cA, t0, T, smem, rot); break;
        case 9: score_stream16p<1>(a, p, branch, posA, posB, cA, t0, T, smem, rot); break;
        case 8:
            if (pair) score_stream16p<0>(a, p, branch, posA, posB, cA, t0, T, smem, rot);
            else score_stream16<8>(a, p, branch, posA, t0, T, smem, rot);
            break;
        case 7: score_stream16<7>(a, p, branch, posA, t0, T, smem, rot); break;
        case 6: score_stream16<6>(a, p, branch, posA, t0, T, smem, rot); break;
        case 5: score_stream16<5>(a, p, branch, posA, t0, T, smem, rot); break;
        case 4: score_stream16<4>(a, p, branch, posA, t0, T, smem, rot); break;
        case 3: score_stream16<3>(a, p, branch, posA, t0, T, smem, rot); break;
        case 2: score_stream16<2>(a, p, branch, posA, t0, T, smem, rot); break;
        case 1: score_stream16<1>(a, p, branch, posA, t0, T, smem, rot); break;
        default:
            score_stream16<0>(a, p, branch, 0, t0, T, smem, rot);           // staging + barriers only
            // a real video with no valid clip (always a wave of its own): the reference's masked maximum is exactly -1e10
            if (w < pp.n_waves) {
                float* row = p.part + ((size_t)branch * p.nv + posA) * p.nq_pad + (size_t)t0 * kQTile;
                for (int q = lane; q < T * kQTile; q += 64) row[q] = -1e10f;
            }
            break;
    }
    if (p.done != nullptr) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __hip_atomic_fetch_add(p.done + range, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// fused[q, v] = w0 * part[0][pos(v)][q] + w1 * part[1][pos(v)][q]  (eval.py:254), plus per-branch copies, for the
// queries [q_lo, q_hi) (q_lo a multiple of 64); output row = q - q_lo.
// 64 x 64 tiles through LDS: every gathered part row is read as 256 contiguous bytes (float4 per lane) and every output
// row segment is written as 256 contiguous bytes (32 x 32 tiles moved 128-byte segments and reached 3.0 TB/s; this form is
// measured in DESIGN.md).
__global__ __launch_bounds__(256) void simpool_finish64_kernel(const float* __restrict__ part, const int32_t* __restrict__ inv,
                                                               int q_lo, int q_hi, int nq_pad, int nv, int n_branches, float w0,
                                                               float w1, const float* __restrict__ q_bad, float* __restrict__ fused,
                                                               float* __restrict__ s0, float* __restrict__ s1) {
    __shared__ float t0[64][65];
    __shared__ float t1[64][65];
#if DLDKD_FIN_SWZ
    // 1-D grid, XCD-aware (workgroup id % 8 = XCD): all video tiles of query tile qt run on XCD qt % 8, consecutively - an output row
    // has 4 * nv bytes, so the 256-byte segments two neighbouring video tiles write share a 128-byte line at their seam; with the
    // neighbours on one XCD, back to back, the two halves of such a line meet in ONE L2 instead of leaving two as partial lines
    const int n_vt = (nv + 63) >> 6, xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int qt = (slot / n_vt) * 8 + xcd, vt = slot % n_vt;
    const int q0 = q_lo + qt * 64, v0 = vt * 64;
    if (q0 >= q_hi) return;
#else
    const int q0 = q_lo + blockIdx.x * 64, v0 = blockIdx.y * 64;
#endif
    {
        const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;   // 16 float4 along q x 16 video rows per pass
        const int q = q0 + 4 * tx;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int vl = ty + 16 * i, vv = v0 + vl;
            f32x4 a = {0.f, 0.f, 0.f, 0.f}, b = a;
            if (vv < nv && q < nq_pad) {
                const size_t row = (size_t)inv[vv] * nq_pad + q;
                a = *reinterpret_cast<const f32x4*>(part + row);
                if (n_branches > 1) b = *reinterpret_cast<const f32x4*>(part + (size_t)nv * nq_pad + row);
                if (q_bad != nullptr) {                      // queries flagged by pack_queries_kernel: NaN scores (rank last)
                    const f32x4 f = *reinterpret_cast<const f32x4*>(q_bad + q);
#pragma unroll
                    for (int e = 0; e < 4; ++e) if (f[e] != 0.f) a[e] = b[e] = __builtin_nanf("");
                }
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                t0[vl][4 * tx + e] = a[e];
                t1[vl][4 * tx + e] = b[e];
            }
        }
    }
    __syncthreads();
    const int vl = threadIdx.x & 63, vv = v0 + vl;
    if (vv >= nv) return;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int ql = (threadIdx.x >> 6) + 4 * i, qq = q0 + ql;
        if (qq < q_hi) {
            const float a = t0[vl][ql], b = t1[vl][ql];
            const size_t o = (size_t)(qq - q_lo) * nv + vv;
            if (fused) fused[o] = n_branches > 1 ? fuse2(w0, a, w1, b) : a;
            if (s0) s0[o] = a;
            if (s1) s1[o] = b;
        }
    }
}

}  // namespace dldkd

using namespace dldkd;

// Query split of a launch: n_wg0 workgroups per range (one per CU: 512 registers per wave), n_qtiles query tiles.
// Cost model in units of "one query tile on one CU": a workgroup pays a fixed prologue (its 4 videos = 384 KiB from HBM at
// ~25 GB/s per CU = ~15 us = ~7 tile times of 2.15 us) plus its tiles; the chip runs ceil(workgroups / 256) rounds.
// The split with the lowest modelled time wins; ranges keep >= 16 tiles so the prologue stays amortised.
static int pick_q_split(int n_wg0, int n_qtiles, int min_split) {
    constexpr int kCUs = 256, kPrologueTiles = 7, kMinTiles = 16, kMaxSplit = 64;
    int best = 1;
    long best_cost = -1;
    for (int s = 1; s <= kMaxSplit && s <= n_qtiles; ++s) {
        const int tiles = (n_qtiles + s - 1) / s;
        if (s > 1 && tiles < kMinTiles && s > min_split) break;
        const int ranges = (n_qtiles + tiles - 1) / tiles;          // the last range may vanish when tiles rounds up
        if (ranges < min_split && s < n_qtiles) continue;
        const long rounds = ((long)n_wg0 * ranges + kCUs - 1) / kCUs;
        const long cost = rounds * (kPrologueTiles + tiles);
        if (best_cost < 0 || cost < best_cost) { best_cost = cost; best = ranges; }
    }
    return best;
}

extern "C" {

size_t dldkd_packed_queries_bytes(int nq) {
    return (size_t)round_up(nq < 1 ? 1 : nq, kQTile) * kHidden * 2;
}
size_t dldkd_packed_gallery_bytes(int nv, int L) {
    return (size_t)(nv < 1 ? 1 : nv) * round_up(L < 1 ? 1 : L, 32) * kHidden * 2;
}
size_t dldkd_simpool_eval_workspace_bytes(int nq, int nv, int n_branches) {
    return (size_t)n_branches * (nv < 1 ? 1 : nv) * round_up(nq < 1 ? 1 : nq, kQTile) * sizeof(float);
}

int dldkd_pack_queries_bf16(const float* q, int nq, int normalize, void* q_packed, float* bad_flags, void* stream) {
    if (nq < 0 || (nq > 0 && (!q || !q_packed))) { set_error("pack_queries: bad arguments"); return DLDKD_EINVAL; }
    if (nq == 0) return DLDKD_OK;
    const int nq_pad = round_up(nq, kQTile);
    DLDKD_LAUNCH(pack_queries_kernel, dim3((nq_pad + 3) / 4), dim3(256), 0, (hipStream_t)stream, q, nq, nq_pad,
                       normalize, (bf16x8*)q_packed, bad_flags);
    return check_launch("pack_queries");
}

int dldkd_pack_gallery_bf16(const float* g, const float* mask, int nv, int L, int normalize, void* g_packed,
                            int32_t* lens, void* stream) {
    if (nv < 0 || L < 1 || L > DLDKD_MAX_CLIPS || (nv > 0 && (!g || !g_packed || !lens))) {
        set_error("pack_gallery: bad arguments (nv=%d L=%d, L must be 1..%d)", nv, L, DLDKD_MAX_CLIPS);
        return DLDKD_EINVAL;
    }
    if (nv == 0) return DLDKD_OK;
    const int Lp = round_up(L, 32);
    const long rows = (long)nv * Lp;
    DLDKD_LAUNCH(pack_gallery_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, g, mask,
                       nv, L, Lp, normalize, (bf16x8*)g_packed);
    DLDKD_LAUNCH(mask_lens_kernel, dim3((nv + 3) / 4), dim3(256), 0, (hipStream_t)stream, mask, nv, L, lens);
    return check_launch("pack_gallery");
}

int dldkd_mask_lens_f32(const float* mask, int n, int L, int32_t* lens, void* stream) {
    if (n < 0 || L < 1 || (n > 0 && (!mask || !lens))) { set_error("mask_lens: bad arguments"); return DLDKD_EINVAL; }
    if (n == 0) return DLDKD_OK;
    DLDKD_LAUNCH(mask_lens_kernel, dim3((n + 3) / 4), dim3(256), 0, (hipStream_t)stream, mask, n, L, lens);
    return check_launch("mask_lens");
}

int dldkd_pack_gallery_chunk_bf16(const float* g, const float* mask, int nv_chunk, int L_chunk, int normalize,
                                  void* g_packed, int32_t* lens, int v0, int nv_total, int L_total, void* stream) {
    if (nv_chunk < 0 || v0 < 0 || nv_total < 0 || (long)v0 + nv_chunk > nv_total || L_chunk < 1 || L_total < L_chunk ||
        L_total > DLDKD_MAX_CLIPS || (nv_chunk > 0 && (!g || !g_packed || !lens))) {
        set_error("pack_gallery_chunk: bad arguments (chunk %d videos x %d clips at %d into %d x %d, max %d clips)", nv_chunk,
                  L_chunk, v0, nv_total, L_total, DLDKD_MAX_CLIPS);
        return DLDKD_EINVAL;
    }
    if (nv_chunk == 0) return DLDKD_OK;
    const int Lp = round_up(L_total, 32);
    const long rows = (long)nv_chunk * Lp;
    bf16x8* dst = (bf16x8*)g_packed + (size_t)v0 * Lp * kRowBf16x8;
    DLDKD_LAUNCH(pack_gallery_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, g, mask,
                       nv_chunk, L_chunk, Lp, normalize, dst);
    DLDKD_LAUNCH(mask_lens_kernel, dim3((nv_chunk + 3) / 4), dim3(256), 0, (hipStream_t)stream, mask, nv_chunk, L_chunk,
                       lens + v0);
    return check_launch("pack_gallery_chunk");
}

int dldkd_order_by_len_desc(const int32_t* lens, int nv, int32_t* order, int32_t* inv_order, void* stream) {
    if (nv < 0) { set_error("order_by_len_desc: nv must be >= 0"); return DLDKD_EINVAL; }
    if (nv == 0) return DLDKD_OK;
    if (!lens || !order || !inv_order) { set_error("order_by_len_desc: null pointer"); return DLDKD_EINVAL; }
    DLDKD_LAUNCH(order_by_len_desc_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, lens, nv, order, inv_order);
    return check_launch("order_by_len_desc");
}

int dldkd_simpool_eval_plan(int nq, int nv, int n_branches, int min_split, int* n_ranges, int* queries_per_range) {
    if (nq < 0 || nv < 0 || n_branches < 1 || n_branches > 2 || min_split < 0 || !n_ranges || !queries_per_range) {
        set_error("simpool_eval_plan: bad arguments");
        return DLDKD_EINVAL;
    }
    const int n_qtiles = round_up(nq < 1 ? 1 : nq, kQTile) / kQTile;
    const int n_wg0 = ((nv < 1 ? 1 : nv) + 3) / 4 * n_branches;
    const int split = pick_q_split(n_wg0, n_qtiles, min_split < 1 ? 1 : min_split);
    const int tiles = (n_qtiles + split - 1) / split;
    *n_ranges = (n_qtiles + tiles - 1) / tiles;
    *queries_per_range = tiles * kQTile;
    return DLDKD_OK;
}

int dldkd_simpool_eval_bf16(const void* const* q_packed, const void* const* g_packed, const int32_t* lens,
                            const int32_t* order, int nq, int nv, int L, int n_branches, int q_split, int32_t* done,
                            void* workspace, void* stream) {
    if (nq < 0 || nv < 0 || L < 1 || L > DLDKD_MAX_CLIPS || n_branches < 1 || n_branches > 2 || q_split < 0) {
        set_error("simpool_eval: bad sizes nq=%d nv=%d L=%d branches=%d q_split=%d", nq, nv, L, n_branches, q_split);
        return DLDKD_EINVAL;
    }
    if (nq == 0 || nv == 0) return DLDKD_OK;
    if (!q_packed || !g_packed || !lens || !order || !workspace || !q_packed[0] || !g_packed[0] ||
        (n_branches == 2 && (!q_packed[1] || !g_packed[1]))) {
        set_error("simpool_eval: null pointer");
        return DLDKD_EINVAL;
    }
    SimpoolEvalArgs p;
    for (int b = 0; b < 2; ++b) {
        p.q[b] = (const bf16x8*)q_packed[b < n_branches ? b : 0];
        p.g[b] = (const bf16x8*)g_packed[b < n_branches ? b : 0];
    }
    p.lens = lens;
    p.order = order;
    p.part = (float*)workspace;
    p.done = done;
    p.nq_pad = round_up(nq, kQTile);
    p.nv = nv;
    p.Lp = round_up(L, 32);
    p.n_qtiles = p.nq_pad / kQTile;
    p.n_groups = (nv + 3) / 4;
    p.n_wg0 = p.n_groups * n_branches;
    const int split = q_split > 0 ? (q_split < p.n_qtiles ? q_split : p.n_qtiles) : pick_q_split(p.n_wg0, p.n_qtiles, 1);
    p.tiles_per_range = (p.n_qtiles + split - 1) / split;
    const int n_ranges = (p.n_qtiles + p.tiles_per_range - 1) / p.tiles_per_range;
    p.ablate = 0;
#ifdef DLDKD_DIAG_ABLATE
    static const int ablate_env = [] { const char* e = getenv("DLDKD_SIMPOOL_ABLATE"); return e ? atoi(e) : 0; }();
    p.ablate = ablate_env;
#endif
    static const bool attr_ok = [] {
        return hipFuncSetAttribute((const void*)simpool_eval16_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                   kRing * kQTileBytes) == hipSuccess;
    }();
    (void)attr_ok;
    DLDKD_LAUNCH(simpool_eval16_kernel, dim3((unsigned)p.n_wg0 * n_ranges), dim3(256), kRing * kQTileBytes,
                       (hipStream_t)stream, p);
    return check_launch("simpool_eval");
}

int dldkd_simpool_eval_pairs_bf16(const void* const* q_packed, const void* const* g_packed, const int32_t* lens,
                                  const int32_t* order, const int32_t* pairs, int n_waves, int nq, int nv, int L, int n_branches,
                                  int q_split, int32_t* done, void* workspace, void* stream) {
    if (nq < 0 || nv < 0 || n_waves < 0 || n_waves > nv || L < 1 || L > DLDKD_MAX_CLIPS || n_branches < 1 || n_branches > 2 || q_split < 0) {
        set_error("simpool_eval_pairs: bad sizes nq=%d nv=%d waves=%d L=%d branches=%d q_split=%d", nq, nv, n_waves, L, n_branches, q_split);
        return DLDKD_EINVAL;
    }
    if (nq == 0 || nv == 0) return DLDKD_OK;
    if (!q_packed || !g_packed || !lens || !order || !pairs || !workspace || !q_packed[0] || !g_packed[0] ||
        (n_branches == 2 && (!q_packed[1] || !g_packed[1]))) {
        set_error("simpool_eval_pairs: null pointer");
        return DLDKD_EINVAL;
    }
    SimpoolPairArgs pp;
    SimpoolEvalArgs& p = pp.e;
    for (int b = 0; b < 2; ++b) {
        p.q[b] = (const bf16x8*)q_packed[b < n_branches ? b : 0];
        p.g[b] = (const bf16x8*)g_packed[b < n_branches ? b : 0];
    }
    p.lens = lens;
    p.order = order;
    p.part = (float*)workspace;
    p.done = done;
    p.nq_pad = round_up(nq, kQTile);
    p.nv = nv;
    p.Lp = round_up(L, 32);
    p.n_qtiles = p.nq_pad / kQTile;
    p.n_groups = (n_waves + 3) / 4;
    p.n_wg0 = p.n_groups * n_branches;
    const int split = q_split > 0 ? (q_split < p.n_qtiles ? q_split : p.n_qtiles) : pick_q_split(p.n_wg0, p.n_qtiles, 1);
    p.tiles_per_range = (p.n_qtiles + split - 1) / split;
    const int n_ranges = (p.n_qtiles + p.tiles_per_range - 1) / p.tiles_per_range;
    p.ablate = 0;
    pp.pairs = pairs;
    pp.n_waves = n_waves;
    static const bool attr_ok = [] {
        return hipFuncSetAttribute((const void*)simpool_eval16p_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                   kRing * kQTileBytes) == hipSuccess;
    }();
    (void)attr_ok;
    DLDKD_LAUNCH(simpool_eval16p_kernel, dim3((unsigned)p.n_wg0 * n_ranges), dim3(256), kRing * kQTileBytes, (hipStream_t)stream, pp);
    return check_launch("simpool_eval_pairs");
}

int dldkd_simpool_finish_range(const void* workspace, const int32_t* inv_order, int nq, int nv, int n_branches, float w0,
                               float w1, int q_lo, int q_hi, const float* q_bad, float* fused, float* s0, float* s1, void* stream) {
    if (nq < 0 || nv < 0 || n_branches < 1 || n_branches > 2 || q_lo < 0 || q_hi < q_lo || q_hi > nq || (q_lo & 3)) {
        set_error("simpool_finish: bad sizes nq=%d nv=%d branches=%d range [%d, %d)", nq, nv, n_branches, q_lo, q_hi);
        return DLDKD_EINVAL;
    }
    if (q_hi == q_lo || nv == 0 || (!fused && !s0 && !s1)) return DLDKD_OK;
    if (!workspace || !inv_order) { set_error("simpool_finish: null pointer"); return DLDKD_EINVAL; }
    const int nq_pad = round_up(nq, kQTile);
#if DLDKD_FIN_SWZ
    const unsigned n_qt = (unsigned)((q_hi - q_lo + 63) / 64), n_vt = (unsigned)((nv + 63) / 64);
    const dim3 grid(8u * n_vt * ((n_qt + 7u) / 8u));
#else
    const dim3 grid((q_hi - q_lo + 63) / 64, (nv + 63) / 64);
#endif
    DLDKD_LAUNCH(simpool_finish64_kernel, grid, dim3(256), 0, (hipStream_t)stream,
                       (const float*)workspace, inv_order, q_lo, q_hi, nq_pad, nv, n_branches, w0, w1, q_bad, fused, s0, s1);
    return check_launch("simpool_finish");
}

int dldkd_simpool_finish(const void* workspace, const int32_t* inv_order, int nq, int nv, int n_branches, float w0,
                         float w1, float* fused, float* s0, float* s1, void* stream) {
    return dldkd_simpool_finish_range(workspace, inv_order, nq, nv, n_branches, w0, w1, 0, nq < 0 ? 0 : nq, nullptr, fused, s0, s1, stream);
}

int dldkd_stream_wait_counter(void* stream, int32_t* counter, int32_t at_least) {
    if (!counter) { set_error("stream_wait_counter: null pointer"); return DLDKD_EINVAL; }
    const hipError_t e = hipStreamWaitValue32((hipStream_t)stream, counter, (uint32_t)at_least, hipStreamWaitValueGte, 0xFFFFFFFFu);
    if (e != hipSuccess) { set_error("stream_wait_counter: %s", hipGetErrorString(e)); return DLDKD_ELAUNCH; }
    return DLDKD_OK;
}

}  // extern "C"
