// simpool scorer v4 ("row stream").  Same gallery-stationary bf16 contraction + in-register key-clip max-pool as
// scorer v2 (simpool_eval.hip), but the unit of work is no longer "one video per wave":
//
//   v2 gives every video its own wave and computes ceil(len/16) row tiles, so a ragged gallery wastes up to 15 rows
//   per video (C2, len ~ U{24..128}: 90.5 % of the MFMA rows are real clips) and waves of one workgroup run at the
//   pace of the longest video.  Here the valid clips of ALL videos are laid end to end in one row stream and wave w
//   owns stream rows [128 w, 128 w + 128): every wave runs 8 full 16-row tiles of real clips, all waves do identical
//   work (no length sorting), and a video that straddles a wave boundary simply yields two partial maxima ("units")
//   that the finish kernel combines with max().
//
//   The max-pool therefore has to respect segment boundaries inside a wave.  The host planner (dldkd_simpool_plan_stream)
//   guarantees at most ONE segment end per 16-row tile (a video that would end in the same tile as its predecessor is
//   pushed to the next tile boundary; the skipped rows are zero "gap" rows that belong to no segment), so the per-tile
//   metadata is two wave-uniform scalars: tile_end (0 = no end; e = the segment ends after e rows of this tile;
//   +256 = the rest of the tile is gap) and tile_unit (where that segment's maximum goes).  The common tile costs
//   what it costs in v2 (two v_max3 per sub-tile); a boundary tile adds 8 selects, the cross-lane reduce and one store.
#include <stdlib.h>
#include <string.h>

#include <type_traits>

#include "common.hpp"

namespace dldkd {

constexpr int kSK = kHidden / 32;             // 12 k-steps of mfma_f32_16x16x32_bf16
constexpr int kSQTileBytes = 2 * kSK * 1024;  // 24 KiB: 32 queries = 2 sub-tiles of 16 (same packed layout as v2)
constexpr int kSRing = 3;
constexpr int kSRow = kHidden / 8;            // 48 16-byte chunks per gallery row
constexpr int kGap = 256;

struct SimpoolSArgs {
    const bf16x8* q[2];
    const bf16x8* g[2];
    const int32_t* rowsrc;      // [n_waves * 128] gallery blob row (v * Lp + clip) or -1 (zero row)
    const int32_t* tile_end;    // [n_waves * 8]
    const int32_t* tile_unit;   // [n_waves * 8]
    const int32_t* tail_unit;   // [n_waves] unit of the segment still open after the wave's last row, or -1
    float* part;                // [n_branches][n_units][nq_pad]
    int nq_pad, n_waves, n_units, n_qtiles, n_groups;
};

template <int OFF>
__device__ __forceinline__ void s_lds_read(bf16x8& dst, uint32_t lds_addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(lds_addr), "i"(OFF) : "memory");
}
__device__ __forceinline__ float s_xor16_max(float m) {
    const unsigned u = __builtin_bit_cast(unsigned, m);
    auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    return fmaxf(__builtin_bit_cast(float, (unsigned)r[0]), __builtin_bit_cast(float, (unsigned)r[1]));
}
__device__ __forceinline__ float s_xor32_max(float m) {
    const unsigned u = __builtin_bit_cast(unsigned, m);
    auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    return fmaxf(__builtin_bit_cast(float, (unsigned)r[0]), __builtin_bit_cast(float, (unsigned)r[1]));
}

constexpr float kNegBig = -3.0e38f;

template <bool ACTIVE>
__device__ __forceinline__ void stream_wave(const bf16x8 (&a)[8][kSK], const SimpoolSArgs& p, int branch, int wv, char* smem) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const char* qsrc = reinterpret_cast<const char*>(p.q[branch]);
    const int T = p.n_qtiles;

    auto stage = [&](int t, int slot) {
        char* dst = smem + slot * kSQTileBytes;
        const char* src = qsrc + (size_t)t * kSQTileBytes;
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const int piece = wave * 6 + i;
            glds16(src + piece * 1024 + lane * 16, dst + piece * 1024);
        }
    };
    stage(0, 0);
    if (T > 1) stage(1, 1);

    if constexpr (!ACTIVE) {   // padding wave of the last workgroup: staging + barriers only
        int slot2 = 2;
        for (int t = 0; t < T; ++t) {
            __syncthreads();
            if (t + 2 < T) stage(t + 2, slot2);
            slot2 = slot2 == kSRing - 1 ? 0 : slot2 + 1;
        }
    } else {
        // wave-uniform segment metadata (scalar registers)
        int te[8], tu[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            te[i] = __builtin_amdgcn_readfirstlane(p.tile_end[(size_t)wv * 8 + i]);
            tu[i] = __builtin_amdgcn_readfirstlane(p.tile_unit[(size_t)wv * 8 + i]);
        }
        const int tail = __builtin_amdgcn_readfirstlane(p.tail_unit[wv]);
        float* part_b = p.part + (size_t)branch * p.n_units * p.nq_pad + (lane & 15);
        const int rr0 = 4 * (lane >> 4);            // this lane's first row inside a 16-row tile
        constexpr int kPF = 4;

        f32x4 accA[8], accB[8];
#pragma unroll
        for (int rt = 0; rt < 8; ++rt) accB[rt] = f32x4{0.f, 0.f, 0.f, 0.f};
        bf16x8 b[kPF];
        float m = kNegBig;

        // pool slice of row tile RT of the previous sub-tile (queries qoff .. qoff+15)
        auto pool_tile = [&](auto rt_c, const f32x4 (&prev)[8], size_t qoff) {
            constexpr int RT = decltype(rt_c)::value;
            const f32x4 x = prev[RT];
            if (__builtin_expect(te[RT] == 0, 1)) {
                m = fmaxf(fmaxf(m, fmaxf(x[0], x[1])), fmaxf(x[2], x[3]));
            } else {
                int e = te[RT] & 31;
                asm volatile("" : "+s"(e));   // keep the row masks of the rare path out of the loop-invariant set (64 SGPRs)
                float lo = kNegBig, hi = kNegBig;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const bool below = rr0 + i < e;
                    lo = fmaxf(lo, below ? x[i] : kNegBig);
                    hi = fmaxf(hi, below ? kNegBig : x[i]);
                }
                float fin = s_xor32_max(s_xor16_max(fmaxf(m, lo)));
                if (lane < 16) part_b[(size_t)tu[RT] * p.nq_pad + qoff] = fin;
                m = (te[RT] & kGap) ? kNegBig : hi;
            }
        };
        auto pool_tail = [&](size_t qoff) {
            if (tail >= 0) {
                float fin = s_xor32_max(s_xor16_max(m));
                if (lane < 16) part_b[(size_t)tail * p.nq_pad + qoff] = fin;
            }
        };

        auto subtile = [&](auto sub_c, f32x4 (&cur)[8], const f32x4 (&prev)[8], uint32_t cbase, uint32_t nbase, size_t prev_qoff) {
            constexpr int S = decltype(sub_c)::value;
            auto step = [&](auto ks_c) {
                constexpr int ks = decltype(ks_c)::value;
                asm volatile("s_waitcnt lgkmcnt(3)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int rt = 0; rt < 8; ++rt) {
                    if (ks == 0) {
                        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
                        cur[rt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[rt][ks], b[ks % kPF], z, 0, 0, 0);
                    } else {
                        cur[rt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[rt][ks], b[ks % kPF], cur[rt], 0, 0, 0);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
                constexpr int idx = S * kSK + kPF + ks;
                if constexpr (idx < 2 * kSK) s_lds_read<idx * 1024>(b[ks % kPF], cbase);
                else s_lds_read<(idx - 2 * kSK) * 1024>(b[ks % kPF], nbase);
                if constexpr (ks == 0) m = kNegBig;
                if constexpr (ks < 8) pool_tile(std::integral_constant<int, ks>{}, prev, prev_qoff);
                else if constexpr (ks == 8) pool_tail(prev_qoff);
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
            // Only the LDS-DMA of tile t+1 must have landed before the barrier; result stores may stay in flight.  The
            // number of stores per sub-tile varies per wave here (one per segment end), so wait for everything but
            // a bounded tail is not expressible - drain VMEM: stores are few and were issued >= one sub-tile ago.
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            if (t + 2 < T) stage(t + 2, slot2);
            const int nslot = slot == kSRing - 1 ? 0 : slot + 1;
            const uint32_t smem_lds = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(smem));
            const uint32_t cbase = smem_lds + slot * kSQTileBytes + lane * 16;
            const uint32_t nbase = smem_lds + nslot * kSQTileBytes + lane * 16;
            if (t == 0) {
                s_lds_read<0>(b[0], cbase);
                s_lds_read<1024>(b[1], cbase);
                s_lds_read<2048>(b[2], cbase);
                s_lds_read<3072>(b[3], cbase);
                __builtin_amdgcn_sched_barrier(0);
            }
            // t == 0: no previous sub-tile; its garbage goes to queries 0..15 of the units, which the next sub-tile's
            // stores (later in program order, same lanes, same addresses) overwrite
            subtile(std::integral_constant<int, 0>{}, accA, accB, cbase, nbase, t > 0 ? (size_t)t * 32 - 16 : 0);
            subtile(std::integral_constant<int, 1>{}, accB, accA, cbase, nbase, (size_t)t * 32);
            slot = nslot;
            slot2 = slot2 == kSRing - 1 ? 0 : slot2 + 1;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        // drain: pool of the very last sub-tile
        m = kNegBig;
        const size_t qlast = (size_t)(T - 1) * 32 + 16;
        pool_tile(std::integral_constant<int, 0>{}, accB, qlast); pool_tile(std::integral_constant<int, 1>{}, accB, qlast);
        pool_tile(std::integral_constant<int, 2>{}, accB, qlast); pool_tile(std::integral_constant<int, 3>{}, accB, qlast);
        pool_tile(std::integral_constant<int, 4>{}, accB, qlast); pool_tile(std::integral_constant<int, 5>{}, accB, qlast);
        pool_tile(std::integral_constant<int, 6>{}, accB, qlast); pool_tile(std::integral_constant<int, 7>{}, accB, qlast);
        pool_tail(qlast);
    }
}

__global__ __launch_bounds__(256, 1) void simpool_eval_stream_kernel(const SimpoolSArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int branch = blockIdx.x / p.n_groups;
    const int wv = (blockIdx.x % p.n_groups) * 4 + wave;
    const bool active = wv < p.n_waves;

    // stationary operand: lane l holds stream row 16 rt + l%16, k 32 ks + 8 (l/16) .. +8
    bf16x8 a[8][kSK];
#pragma unroll
    for (int rt = 0; rt < 8; ++rt) {
        const int src = active ? p.rowsrc[(size_t)wv * 128 + rt * 16 + (lane & 15)] : -1;
        if (src >= 0) {
            const bf16x8* gr = p.g[branch] + (size_t)src * kSRow + (lane >> 4);
#pragma unroll
            for (int ks = 0; ks < kSK; ++ks) a[rt][ks] = gr[ks * 4];
        } else {
#pragma unroll
            for (int ks = 0; ks < kSK; ++ks) a[rt][ks] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
        }
    }
#pragma unroll
    for (int rt = 0; rt < 8; ++rt)
#pragma unroll
        for (int ks = 0; ks < kSK; ++ks) {
            if (rt * kSK + ks < 64) asm volatile("" : "+a"(a[rt][ks]));
            else asm volatile("" : "+v"(a[rt][ks]));
        }
    if (active) stream_wave<true>(a, p, branch, wv, smem);
    else stream_wave<false>(a, p, branch, wv, smem);
}

}  // namespace dldkd

using namespace dldkd;

static inline int s_round_up(int x, int m) { return (x + m - 1) / m * m; }

extern "C" {

int dldkd_simpool_plan_stream(const int32_t* lens, int nv, int Lp, int max_waves, int32_t* rowsrc, int32_t* tile_end,
                              int32_t* tile_unit, int32_t* tail_unit, int32_t* video_unit0, int32_t* video_unit1, int* n_waves,
                              int* n_units) {
    if (nv < 0 || Lp < 1 || max_waves < 1 || !lens || !rowsrc || !tile_end || !tile_unit || !tail_unit || !video_unit0 ||
        !video_unit1 || !n_waves || !n_units) {
        set_error("simpool_plan_stream: bad arguments");
        return DLDKD_EINVAL;
    }
    const long cap = (long)max_waves * 128;
    for (long i = 0; i < cap; ++i) rowsrc[i] = -1;
    memset(tile_end, 0, sizeof(int32_t) * (size_t)max_waves * 8);
    for (long i = 0; i < (long)max_waves * 8; ++i) tile_unit[i] = -1;
    for (int i = 0; i < max_waves; ++i) tail_unit[i] = -1;
    long pos = 0, prev_end_tile = -1;
    int units = 0;
    for (int v = 0; v < nv; ++v) {
        const int len = lens[v];
        if (len < 1 || len > 128 || len > Lp) { set_error("simpool_plan_stream: video %d has %d valid clips (need 1..%d)", v, len, Lp < 128 ? Lp : 128); return DLDKD_EINVAL; }
        long start = pos;
        if (prev_end_tile >= 0 && (start + len - 1) / 16 == prev_end_tile) {   // would be the tile's second segment end
            start = (prev_end_tile + 1) * 16;
            tile_end[prev_end_tile] |= kGap;
        }
        const long end = start + len;
        if (end > cap) { set_error("simpool_plan_stream: max_waves %d too small", max_waves); return DLDKD_EINVAL; }
        for (int l = 0; l < len; ++l) rowsrc[start + l] = v * Lp + l;
        const long T = (end - 1) / 16;
        tile_end[T] = (int32_t)(end - 16 * T);
        const long w0 = start / 128, w1 = (end - 1) / 128;
        if (w0 != w1) {
            tail_unit[w0] = units;
            video_unit0[v] = units++;
            tile_unit[T] = units;
            video_unit1[v] = units++;
        } else {
            tile_unit[T] = units;
            video_unit0[v] = units++;
            video_unit1[v] = -1;
        }
        pos = end;
        prev_end_tile = T;
    }
    if (prev_end_tile >= 0 && pos % 16) tile_end[prev_end_tile] |= kGap;   // rows after the last clip are zero rows
    *n_waves = (int)((pos + 127) / 128);
    *n_units = units;
    return DLDKD_OK;
}

int dldkd_simpool_eval_stream_bf16(const void* const* q_packed, const void* const* g_packed, const int32_t* rowsrc,
                                   const int32_t* tile_end, const int32_t* tile_unit, const int32_t* tail_unit, int nq, int n_waves,
                                   int n_units, int n_branches, void* workspace, void* stream) {
    if (nq < 0 || n_waves < 0 || n_units < 0 || n_branches < 1 || n_branches > 2) { set_error("simpool_eval_stream: bad sizes"); return DLDKD_EINVAL; }
    if (nq == 0 || n_waves == 0) return DLDKD_OK;
    if (!q_packed || !g_packed || !rowsrc || !tile_end || !tile_unit || !tail_unit || !workspace || !q_packed[0] || !g_packed[0] ||
        (n_branches == 2 && (!q_packed[1] || !g_packed[1]))) {
        set_error("simpool_eval_stream: null pointer");
        return DLDKD_EINVAL;
    }
    SimpoolSArgs p;
    for (int b = 0; b < 2; ++b) {
        p.q[b] = (const bf16x8*)q_packed[b < n_branches ? b : 0];
        p.g[b] = (const bf16x8*)g_packed[b < n_branches ? b : 0];
    }
    p.rowsrc = rowsrc; p.tile_end = tile_end; p.tile_unit = tile_unit; p.tail_unit = tail_unit;
    p.part = (float*)workspace;
    p.nq_pad = s_round_up(nq, 32);
    p.n_waves = n_waves;
    p.n_units = n_units;
    p.n_qtiles = p.nq_pad / 32;
    p.n_groups = (n_waves + 3) / 4;
    static const bool attr_ok = [] {
        return hipFuncSetAttribute((const void*)simpool_eval_stream_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                   kSRing * kSQTileBytes) == hipSuccess;
    }();
    (void)attr_ok;
    hipLaunchKernelGGL(simpool_eval_stream_kernel, dim3(p.n_groups * n_branches), dim3(256), kSRing * kSQTileBytes, (hipStream_t)stream, p);
    return check_launch("simpool_eval_stream");
}

}  // extern "C"
