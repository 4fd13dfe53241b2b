// simpool scorer v3 ("half-video units"): the same gallery-stationary bf16 contraction + in-register key-clip
// max-pool as simpool_eval.hip, restructured for TWO waves per SIMD.
//
// v2 keeps a whole 128-clip video (384 registers) in one wave, so each SIMD hosts a single wave and every
// instruction that is not an MFMA (max-pool VALU, LDS-DMA issue, fragment reads, barrier) is time the MFMA pipe of
// that SIMD idles (ablation: profiles/r01/ablation_simpool_v2.md).  Here the unit of work is a HALF video:
// 64 clips x 384 dims = 192 registers (all in the accumulator half of the unified file), so a wave fits in 256
// registers, two waves share each SIMD, and one wave's bookkeeping hides under the other's MFMAs.  A video longer
// than 64 clips is two independent units whose partial maxima are combined by the finish kernel.
// Workgroup = 8 waves = 8 units sharing the query stream (3-slot LDS ring, 3 LDS-DMA pieces per wave per tile).
#include <stdlib.h>

#include <type_traits>

#include "common.hpp"

namespace dldkd {

constexpr int kHK = kHidden / 32;             // 12 k-steps of mfma_f32_16x16x32_bf16
constexpr int kHQTileBytes = 2 * kHK * 1024;  // 24 KiB: 32 queries = 2 sub-tiles of 16
constexpr int kHRing = 3;
constexpr int kHRow = kHidden / 8;            // 48 16-byte chunks per gallery row

struct SimpoolHArgs {
    const bf16x8* q[2];
    const bf16x8* g[2];
    const int32_t* unit_video;   // [n_units] video id of the unit
    const int32_t* unit_row0;    // [n_units] first clip of the unit (0 or 64)
    const int32_t* unit_rows;    // [n_units] valid clips in the unit (1..64)
    float* part;                 // [n_branches][n_units][nq_pad]
    int nq_pad, n_units, Lp, n_qtiles, n_groups;
};

template <int OFF>
__device__ __forceinline__ void h_lds_read(bf16x8& dst, uint32_t lds_addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(lds_addr), "i"(OFF) : "memory");
}
__device__ __forceinline__ float h_xor16_max(float m) {
    const unsigned u = __builtin_bit_cast(unsigned, m);
    auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    return fmaxf(__builtin_bit_cast(float, (unsigned)r[0]), __builtin_bit_cast(float, (unsigned)r[1]));
}
__device__ __forceinline__ float h_xor32_max(float m) {
    const unsigned u = __builtin_bit_cast(unsigned, m);
    auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    return fmaxf(__builtin_bit_cast(float, (unsigned)r[0]), __builtin_bit_cast(float, (unsigned)r[1]));
}

template <int NRT>   // 16-clip row tiles of the unit, 0..4
__device__ __forceinline__ void unit_stream(const bf16x8 (&a)[4][kHK], const SimpoolHArgs& p, int branch, int unit, int rows,
                                            char* smem) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const char* qsrc = reinterpret_cast<const char*>(p.q[branch]);
    const int T = p.n_qtiles;

    auto stage = [&](int t, int slot) {   // 24 pieces per tile, 3 per wave
        char* dst = smem + slot * kHQTileBytes;
        const char* src = qsrc + (size_t)t * kHQTileBytes;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int piece = wave * 3 + i;
            glds16(src + piece * 1024 + lane * 16, dst + piece * 1024);
        }
    };
    stage(0, 0);
    if (T > 1) stage(1, 1);

    if constexpr (NRT == 0) {
        int slot2 = 2;
        for (int t = 0; t < T; ++t) {
            __syncthreads();
            if (t + 2 < T) stage(t + 2, slot2);
            slot2 = slot2 == kHRing - 1 ? 0 : slot2 + 1;
        }
    } else {
        const int lim = rows - 16 * (NRT - 1) - 4 * (lane >> 4);
        const bool ok0 = 0 < lim, ok1 = 1 < lim, ok2 = 2 < lim, ok3 = 3 < lim;
        float* outp = p.part + ((size_t)branch * p.n_units + unit) * p.nq_pad + (lane & 15);
        constexpr int kPF = 4;
        bf16x8 b[kPF];
        f32x4 acc[NRT];

        auto subtile = [&](auto sub_c, uint32_t cbase, uint32_t nbase, float* out) {
            constexpr int S = decltype(sub_c)::value;
            auto step = [&](auto ks_c) {
                constexpr int ks = decltype(ks_c)::value;
                asm volatile("s_waitcnt lgkmcnt(3)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int rt = 0; rt < NRT; ++rt) {
                    if (ks == 0) {
                        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
                        acc[rt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[rt][ks], b[ks % kPF], z, 0, 0, 0);
                    } else {
                        acc[rt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[rt][ks], b[ks % kPF], acc[rt], 0, 0, 0);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
                constexpr int idx = S * kHK + kPF + ks;
                if constexpr (idx < 2 * kHK) h_lds_read<idx * 1024>(b[ks % kPF], cbase);
                else h_lds_read<(idx - 2 * kHK) * 1024>(b[ks % kPF], nbase);
                __builtin_amdgcn_sched_barrier(0);
            };
            step(std::integral_constant<int, 0>{}); step(std::integral_constant<int, 1>{});
            step(std::integral_constant<int, 2>{}); step(std::integral_constant<int, 3>{});
            step(std::integral_constant<int, 4>{}); step(std::integral_constant<int, 5>{});
            step(std::integral_constant<int, 6>{}); step(std::integral_constant<int, 7>{});
            step(std::integral_constant<int, 8>{}); step(std::integral_constant<int, 9>{});
            step(std::integral_constant<int, 10>{}); step(std::integral_constant<int, 11>{});
            // key-clip max-pool of this sub-tile: the sibling wave on this SIMD issues MFMAs meanwhile
            float m = -3.0e38f;
#pragma unroll
            for (int rt = 0; rt < NRT; ++rt)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float x = acc[rt][j];
                    if (rt == NRT - 1) {
                        const bool ok = j == 0 ? ok0 : j == 1 ? ok1 : j == 2 ? ok2 : ok3;
                        x = ok ? x : -3.0e38f;
                    }
                    m = fmaxf(m, x);
                }
            m = h_xor32_max(h_xor16_max(m));
            if (lane < 16) *out = m;
        };

        int slot = 0, slot2 = 2;
        for (int t = 0; t < T; ++t) {
            // tile t+1's DMA is older than this wave's two result stores of iteration t-1 (see simpool_eval.hip)
            if (t == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            if (t + 2 < T) stage(t + 2, slot2);
            const int nslot = slot == kHRing - 1 ? 0 : slot + 1;
            const uint32_t smem_lds = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(smem));
            const uint32_t cbase = smem_lds + slot * kHQTileBytes + lane * 16;
            const uint32_t nbase = smem_lds + nslot * kHQTileBytes + lane * 16;
            if (t == 0) {
                h_lds_read<0>(b[0], cbase);
                h_lds_read<1024>(b[1], cbase);
                h_lds_read<2048>(b[2], cbase);
                h_lds_read<3072>(b[3], cbase);
                __builtin_amdgcn_sched_barrier(0);
            }
            subtile(std::integral_constant<int, 0>{}, cbase, nbase, outp + (size_t)t * 32);
            subtile(std::integral_constant<int, 1>{}, cbase, nbase, outp + (size_t)t * 32 + 16);
            slot = nslot;
            slot2 = slot2 == kHRing - 1 ? 0 : slot2 + 1;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
}

__global__ __launch_bounds__(512, 2) void simpool_eval_h_kernel(const SimpoolHArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int branch = blockIdx.x / p.n_groups;
    const int unit = (blockIdx.x % p.n_groups) * 8 + wave;
    int rows = 0, v = 0, row0 = 0;
    if (unit < p.n_units) {
        v = p.unit_video[unit];
        row0 = p.unit_row0[unit];
        rows = p.unit_rows[unit];
    }
    rows = __builtin_amdgcn_readfirstlane(rows);
    const int nrt = (rows + 15) >> 4;

    bf16x8 a[4][kHK];
    const bf16x8* gv = p.g[branch] + ((size_t)v * p.Lp + row0 + (lane & 15)) * kHRow + (lane >> 4);
#pragma unroll
    for (int rt = 0; rt < 4; ++rt) {
        if (rt < nrt) {
#pragma unroll
            for (int ks = 0; ks < kHK; ++ks) a[rt][ks] = gv[(size_t)rt * 16 * kHRow + ks * 4];
        } else {
#pragma unroll
            for (int ks = 0; ks < kHK; ++ks) a[rt][ks] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
        }
    }
    // 256 registers per wave = 128 accumulator-half + 128 arch: 32 fragments pinned to AGPRs, 16 to VGPRs.  (All 48
    // pinned "+a" made hipcc spill 16 of them to VGPRs and copy each back before its MFMA: 2 v_mov_b64 per MFMA.)
#pragma unroll
    for (int rt = 0; rt < 4; ++rt)
#pragma unroll
        for (int ks = 0; ks < kHK; ++ks) {
            if (rt * kHK + ks < 32) asm volatile("" : "+a"(a[rt][ks]));
            else asm volatile("" : "+v"(a[rt][ks]));
        }

    switch (nrt) {
        case 4: unit_stream<4>(a, p, branch, unit, rows, smem); break;
        case 3: unit_stream<3>(a, p, branch, unit, rows, smem); break;
        case 2: unit_stream<2>(a, p, branch, unit, rows, smem); break;
        case 1: unit_stream<1>(a, p, branch, unit, rows, smem); break;
        default: unit_stream<0>(a, p, branch, unit, rows, smem); break;
    }
}

// fused[q, v] = w0 * max_u part[0][u][q] + w1 * max_u part[1][u][q] over the 1-2 units u of video v
__global__ __launch_bounds__(256) void simpool_finish_units_kernel(const float* __restrict__ part, const int32_t* __restrict__ u0,
                                                                   const int32_t* __restrict__ u1, int nq, int nq_pad, int nv,
                                                                   int n_units, int n_branches, float w0, float w1,
                                                                   float* __restrict__ fused, float* __restrict__ s0,
                                                                   float* __restrict__ s1) {
    __shared__ float t0[32][33];
    __shared__ float t1[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int q0 = blockIdx.x * 32, v0 = blockIdx.y * 32;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int vv = v0 + ty + 8 * i;
        float a = 0.f, b = 0.f;
        if (vv < nv) {
            const int ua = u0[vv], ub = u1[vv];
            const size_t ra = (size_t)ua * nq_pad + q0 + tx;
            a = part[ra];
            if (n_branches > 1) b = part[(size_t)n_units * nq_pad + ra];
            if (ub >= 0) {
                const size_t rb = (size_t)ub * nq_pad + q0 + tx;
                a = fmaxf(a, part[rb]);
                if (n_branches > 1) b = fmaxf(b, part[(size_t)n_units * nq_pad + rb]);
            }
        }
        t0[ty + 8 * i][tx] = a;
        t1[ty + 8 * i][tx] = b;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int qq = q0 + ty + 8 * i, vv = v0 + tx;
        if (qq < nq && vv < nv) {
            const float a = t0[tx][ty + 8 * i], b = t1[tx][ty + 8 * i];
            const size_t o = (size_t)qq * nv + vv;
            if (fused) fused[o] = n_branches > 1 ? w0 * a + w1 * b : a;
            if (s0) s0[o] = a;
            if (s1) s1[o] = b;
        }
    }
}

}  // namespace dldkd

using namespace dldkd;

static inline int h_round_up(int x, int m) { return (x + m - 1) / m * m; }

extern "C" {

size_t dldkd_simpool_units_workspace_bytes(int nq, int n_units, int n_branches) {
    return (size_t)n_branches * (n_units < 1 ? 1 : n_units) * h_round_up(nq < 1 ? 1 : nq, 32) * sizeof(float);
}

int dldkd_simpool_eval_units_bf16(const void* const* q_packed, const void* const* g_packed, const int32_t* unit_video,
                                  const int32_t* unit_row0, const int32_t* unit_rows, int nq, int n_units, int L,
                                  int n_branches, void* workspace, void* stream) {
    if (nq < 0 || n_units < 0 || L < 1 || L > DLDKD_MAX_CLIPS || n_branches < 1 || n_branches > 2) {
        set_error("simpool_eval_units: bad sizes");
        return DLDKD_EINVAL;
    }
    if (nq == 0 || n_units == 0) return DLDKD_OK;
    if (!q_packed || !g_packed || !unit_video || !unit_row0 || !unit_rows || !workspace || !q_packed[0] || !g_packed[0] ||
        (n_branches == 2 && (!q_packed[1] || !g_packed[1]))) {
        set_error("simpool_eval_units: null pointer");
        return DLDKD_EINVAL;
    }
    SimpoolHArgs p;
    for (int b = 0; b < 2; ++b) {
        p.q[b] = (const bf16x8*)q_packed[b < n_branches ? b : 0];
        p.g[b] = (const bf16x8*)g_packed[b < n_branches ? b : 0];
    }
    p.unit_video = unit_video;
    p.unit_row0 = unit_row0;
    p.unit_rows = unit_rows;
    p.part = (float*)workspace;
    p.nq_pad = h_round_up(nq, 32);
    p.n_units = n_units;
    p.Lp = h_round_up(L, 32);
    p.n_qtiles = p.nq_pad / 32;
    p.n_groups = (n_units + 7) / 8;
    static const bool attr_ok = [] {
        return hipFuncSetAttribute((const void*)simpool_eval_h_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                   kHRing * kHQTileBytes) == hipSuccess;
    }();
    (void)attr_ok;
    hipLaunchKernelGGL(simpool_eval_h_kernel, dim3(p.n_groups * n_branches), dim3(512), kHRing * kHQTileBytes, (hipStream_t)stream, p);
    return check_launch("simpool_eval_units");
}

int dldkd_simpool_finish_units(const void* workspace, const int32_t* video_unit0, const int32_t* video_unit1, int nq, int nv,
                               int n_units, int n_branches, float w0, float w1, float* fused, float* s0, float* s1,
                               void* stream) {
    if (nq < 0 || nv < 0 || n_units < 0 || n_branches < 1 || n_branches > 2) { set_error("simpool_finish_units: bad sizes"); return DLDKD_EINVAL; }
    if (nq == 0 || nv == 0 || (!fused && !s0 && !s1)) return DLDKD_OK;
    if (!workspace || !video_unit0 || !video_unit1) { set_error("simpool_finish_units: null pointer"); return DLDKD_EINVAL; }
    const int nq_pad = h_round_up(nq, 32);
    hipLaunchKernelGGL(simpool_finish_units_kernel, dim3(nq_pad / 32, (nv + 31) / 32), dim3(256), 0, (hipStream_t)stream,
                       (const float*)workspace, video_unit0, video_unit1, nq, nq_pad, nv, n_units, n_branches, w0, w1, fused, s0, s1);
    return check_launch("simpool_finish_units");
}

}  // extern "C"
