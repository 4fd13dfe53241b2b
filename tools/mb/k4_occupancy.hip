// Microbenchmark for the next K4 design step: does the input projection's k-step instruction mix run closer to the MFMA rate with
// TWO waves per SIMD (each owning 128 rows x 96 columns = 24 MFMAs per k-step) than with ONE (128 rows x 192 columns = 48 MFMAs,
// today's in_proj_rows128_kernel)?  No global memory in the loop: x (fp32) and W' (bf16 fragments) are read from LDS images that are
// never refilled, so what is measured is the issue side only - per 32-k step and wave: 16 ds_read_b128 of fp32 x + 32
// v_cvt_pk_bf16_f32 (8 A fragments), the wave's B fragments (12 / 6 ds_read_b128), the MFMAs, one s_barrier.
//   variant 1: 4 waves per workgroup, 4 x 6 tiles per wave (accumulators: 4 x 4 tiles by the compiler + the mix of K4 is not
//              reproduced - all accumulators here are compiler-allocated, 384 registers -> launch_bounds(256, 1))
//   variant 2: 8 waves per workgroup, 4 x 3 tiles per wave (192 accumulator registers, two waves per SIMD)
// Output: cycles per k-step per workgroup, MFMA-pipe utilisation (48 x 32 cycles of MFMA per SIMD and k-step in both variants),
// and the TFLOP/s of the chip at the measured wall time.
// Measured (MI355X, round 3): one wave 2649 cycles per k-step at 2.37 GHz = 58 % MFMA busy = 1441 TFLOP/s; two waves 2212 cycles
// at 2.35 GHz = 69 % = 1714 TFLOP/s - compiler-scheduled code, no memory traffic.  K4 itself runs its k-step in 1937 cycles (79 %
// busy, hand-scheduled) but at the 1.38 GHz the chip holds under its HBM + LDS-DMA + MFMA load: 945 TFLOP/s.  The issue side is
// not what separates K4 from 0.45 of the HBM peak; the clock is.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ bf16x8 cvt8(const f32x4& lo, const f32x4& hi) {
    unsigned u0, u1, u2, u3;
    asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(u0) : "v"(lo[0]), "v"(lo[1]));
    asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(u1) : "v"(lo[2]), "v"(lo[3]));
    asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(u2) : "v"(hi[0]), "v"(hi[1]));
    asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(u3) : "v"(hi[2]), "v"(hi[3]));
    u32x4 r = {u0, u1, u2, u3};
    return __builtin_bit_cast(bf16x8, r);
}

// 384 accumulators exceed the 256 AGPRs and hipcc puts every MFMA of a kernel in one register class: like K4, the one-wave
// variant issues its MFMAs as asm, column tiles 0-3 in AGPRs, 4-5 in arch VGPRs
__device__ __forceinline__ void mfma_a(f32x16& acc, const bf16x8& a, const bf16x8& b) {
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
}
__device__ __forceinline__ void mfma_v(f32x16& acc, const bf16x8& a, const bf16x8& b) {
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
}

__device__ __forceinline__ void keep_a(const f32x16& acc) { asm volatile("" :: "a"(acc)); }
__device__ __forceinline__ void keep_v(const f32x16& acc) { asm volatile("" :: "v"(acc)); }

// NCT column tiles per wave (6 or 3); WAVES = 4 or 8
template <int NCT, int WAVES>
__global__ __launch_bounds__(64 * WAVES, 1) void kstep(int iters, unsigned long long* out, float* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // x image: 128 rows x 32 fp32 (16 KiB), XOR-swizzled like K4's; W' image: [NCT * WAVES column tiles][2 kk][64 lanes][16 B]
    const char* xs = smem;
    const char* ws = smem + 16384 + wave * (NCT * 2 * 1024);
    for (int i = threadIdx.x; i < (16384 + WAVES * NCT * 2048) / 4; i += blockDim.x) reinterpret_cast<float*>(smem)[i] = 1.0f + (i & 7) * 0.125f;
    __syncthreads();
    f32x16 acc[4][NCT];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < NCT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const int row = lane & 31, half = lane >> 5;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        bf16x8 a[4][2];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                const int r = 32 * i + row;
                const int c0 = (4 * kk + 2 * half) ^ ((r >> 1) & 7), c1 = (4 * kk + 2 * half + 1) ^ ((r >> 1) & 7);
                const f32x4 lo = *reinterpret_cast<const f32x4*>(xs + r * 128 + c0 * 16);
                const f32x4 hi = *reinterpret_cast<const f32x4*>(xs + r * 128 + c1 * 16);
                a[i][kk] = cvt8(lo, hi);
            }
#pragma unroll
        for (int j = 0; j < NCT; ++j)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                const bf16x8 b = *reinterpret_cast<const bf16x8*>(ws + ((j * 2 + kk) * 64 + lane) * 16);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    if constexpr (NCT == 6) { if (j < 4) mfma_a(acc[i][j], a[i][kk], b); else mfma_v(acc[i][j], a[i][kk], b); }
                    else acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][kk], b, acc[i][j], 0, 0, 0);
                }
            }
        __builtin_amdgcn_s_barrier();
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < NCT; ++j) {
            if constexpr (NCT == 6) { if (j < 4) keep_a(acc[i][j]); else keep_v(acc[i][j]); }
            else {
#pragma unroll
                for (int r = 0; r < 16; ++r) s += acc[i][j][r];
            }
        }
    if (s == 12345.678f) sink[0] = s;
}

template <int NCT, int WAVES>
static void run(const char* name, int iters, unsigned long long* out, float* sink) {
    const int nwg = 256;
    const size_t lds = 16384 + WAVES * NCT * 2048;
    hipFuncSetAttribute((const void*)kstep<NCT, WAVES>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    for (int rep = 0; rep < 3; ++rep) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        hipLaunchKernelGGL((kstep<NCT, WAVES>), dim3(nwg), dim3(64 * WAVES), lds, 0, iters, out, sink);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        std::vector<unsigned long long> h(nwg);
        hipMemcpy(h.data(), out, nwg * 8, hipMemcpyDeviceToHost);
        std::sort(h.begin(), h.end());
        const double cyc = (double)h[nwg / 2] / iters;                 // s_memtime counts shader clocks: cycles per k-step (median workgroup)
        const double flop = 2.0 * 128 * 768 * 32 * (double)iters * nwg;
        if (rep == 2)
            printf("%s: %.3f ms for %d k-steps x %d workgroups = %.0f TFLOP/s; %.0f cycles per k-step at %.2f GHz: MFMA pipe busy %.0f %% "
                   "(48 MFMAs x 32 cycles per SIMD and k-step)\n", name, ms, iters, nwg, flop / ms / 1e9, cyc,
                   cyc * iters / (ms * 1e6), 100.0 * 1536.0 / cyc);
    }
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 20000;
    unsigned long long* out; float* sink;
    hipMalloc(&out, 256 * 8); hipMalloc(&sink, 64);
    run<6, 4>("1 wave / SIMD, 128 x 192 per wave (48 MFMA + 32 cvt + 28 LDS reads per k-step)", iters, out, sink);
    run<3, 8>("2 waves / SIMD, 128 x 96 per wave (24 MFMA + 32 cvt + 22 LDS reads per k-step)", iters, out, sink);
    return 0;
}
