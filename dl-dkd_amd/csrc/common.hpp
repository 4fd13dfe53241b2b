// Shared device helpers for the gfx950 kernels (wave64, MFMA, bf16).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/dldkd_hip.h"

namespace dldkd {

constexpr int kHidden = DLDKD_HIDDEN;   // 384 = 24 k-steps of 16 (mfma 32x32x16) = 12 of 32 (16x16x32)
constexpr int kWave = 64;

typedef short bf16x8 __attribute__((ext_vector_type(8)));   // 8 bf16 = one MFMA A/B fragment (4 VGPRs)
typedef float f32x16 __attribute__((ext_vector_type(16)));  // 32x32 accumulator fragment
typedef float f32x4 __attribute__((ext_vector_type(4)));    // 16x16 accumulator fragment / 16-byte fp32 vector

// fp32 -> bf16 bits, round-to-nearest-even, NaN preserving (plain cast -> v_cvt_pk_bf16_f32 on gfx950,
// MI355X_MICROARCH.md "Correctness boundaries").
__device__ __forceinline__ unsigned short f32_to_bf16_bits(float x) {
    return __builtin_bit_cast(unsigned short, static_cast<__bf16>(x));
}
__device__ __forceinline__ float bf16_bits_to_f32(unsigned short b) {
    return __builtin_bit_cast(float, static_cast<unsigned int>(b) << 16);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}

// LDS-DMA: one 16-byte piece per lane, 1 KiB per wave-instruction.  The LDS destination is the
// wave-uniform base + lane*16 (cdna_hip_programming.md section 5 caveat); the global source is per lane.
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void global_cvoid;
__device__ __forceinline__ void glds16(const void* gsrc_lane, void* lds_base_uniform) {
    __builtin_amdgcn_global_load_lds(
        reinterpret_cast<global_cvoid*>(reinterpret_cast<uintptr_t>(gsrc_lane)),
        reinterpret_cast<lds_void*>(static_cast<uint32_t>(reinterpret_cast<uintptr_t>(lds_base_uniform))),
        16, 0, 0);
}


// Epilogue of the 128x128-block / 64x64-per-wave MFMA GEMMs: C = act(alpha * acc + bias).  The 32x32 accumulator layout puts
// 32 consecutive COLUMNS of one row on 32 lanes (128-byte row segments as 4-byte stores).  When the wave's 64 columns are
// all in range and rows are 16-byte aligned, each 32 x 64 half is parked in the wave's private LDS region (the operand
// tiles are dead after the k-loop's last barrier) and written back as float4: 16 lanes cover 256 contiguous bytes of a row.
// Ragged edges keep the element-wise path.  Split-K launches write their partial plane the same way (C then points into
// the split-K workspace, one plane per blockIdx.z) and a reduce pass sums the planes.
template <typename Args>
__device__ __forceinline__ void gemm_store_tile(const f32x16 (&acc)[2][2], const Args& p, int m0, int n0, int wm, int wn, int lane,
                                                float* stg /* 32 x 72 floats, private to the wave */) {
    constexpr int SP = 72;
    const bool fast = !(p.ldc & 3) && !((uintptr_t)p.C & 15) && n0 + wn + 64 <= p.N;   // split-K planes are plain stores too
    if (fast) {
        float bias[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) bias[j] = p.bias ? p.bias[n0 + wn + 32 * j + (lane & 31)] : 0.f;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float v = acc[i][j][r] * p.alpha + bias[j];
                    if (p.relu) v = fmaxf(v, 0.f);
                    stg[((r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * SP + 32 * j + (lane & 31)] = v;
                }
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                const int idx = lane + 64 * it, rl = idx >> 4, c4 = idx & 15;
                const int m = m0 + wm + 32 * i + rl;
                const f32x4 v = *reinterpret_cast<const f32x4*>(stg + rl * SP + 4 * c4);
                if (m < p.M) *reinterpret_cast<f32x4*>(p.C + (size_t)m * p.ldc + n0 + wn + 4 * c4) = v;
            }
        }
        return;
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int n = n0 + wn + 32 * j + (lane & 31);
        if (n >= p.N) continue;
        const float bias = p.bias ? p.bias[n] : 0.f;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (m < p.M) {
                    float v = acc[i][j][r] * p.alpha + bias;
                    if (p.relu) v = fmaxf(v, 0.f);
                    p.C[(size_t)m * p.ldc + n] = v;
                }
            }
    }
}

// split-K plans of the three tiled GEMMs (number of k-slices, 1 = none), shared with dldkd_gemm_workspace_bytes
int gemm_f32_split_plan(int M, int N, int K, int a_kmajor, int b_kmajor, int* k_tiles_per_split);
int gemm_f32x3_split_plan(int M, int N, int K, int a_kmajor, int b_kmajor, int* k_tiles_per_split);
int gemm_bf16_split_plan(int M, int N, int K, int a_kmajor, int b_kmajor, int* k_tiles_per_split);
int launch_splitk_reduce(const float* ws, float* out, int split, long n, hipStream_t s);
void set_error(const char* fmt, ...);
int check_launch(const char* what);

}  // namespace dldkd
