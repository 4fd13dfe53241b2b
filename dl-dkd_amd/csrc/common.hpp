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

void set_error(const char* fmt, ...);
int check_launch(const char* what);

}  // namespace dldkd
