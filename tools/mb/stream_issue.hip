// Microbenchmark: what does it cost a compute wave to keep a weight stream going beside its MFMAs (one wave per SIMD, 4 waves per
// workgroup, one workgroup per CU, all CUs)?  The loop is K5's projection loop in miniature: per MFMA (v_mfma_f32_32x32x16_f16) one
// 1-KiB fragment read from an LDS ring (ds_read_b128, 3 deep, counted lgkmcnt) and, every 4th MFMA, ONE 1-KiB piece of the stream
// (per wave: 8 pieces per 32 MFMAs, as tower_seq.hip) - issued as
//   mode 0  nothing (the fragment reads + MFMAs alone)
//   mode 1  LDS-DMA: s_mov m0 + global_load_lds_dwordx4 (what K5 / K4b / K1 do)
//   mode 2  register staging: global_load_dwordx4 into a ring of D register quads, ds_write_b128 of the piece loaded D pieces earlier
//   mode 3  as 2 with the ring in AGPRs (load -> a[...], ds_write from a[...])
// Reports cycles per MFMA (s_memtime over the loop, median over workgroups).  The stream source is L2-resident (1.5 MB per XCD group).
// usage: stream_issue [D = 4]
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int kIters = 400;        // x 32 MFMAs

template <int MODE, int D>
__global__ __launch_bounds__(256, 1) void k(const char* src, unsigned long long* out, float* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t lds = (uint32_t)(uintptr_t)smem, lane16 = lane * 16;
    // ring of 96 fragment slots (96 KiB) as in K5; the stream writes slot (32 c + 8 wave + i), the reads walk all 96
    for (int i = threadIdx.x; i < 96 * 1024 / 16; i += 256) reinterpret_cast<u32x4*>(smem)[i] = u32x4{0x3c003c00u, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u};
    __syncthreads();
    const char* base = src + (size_t)(blockIdx.x & 7) * (1536 * 1024);       // 1.5 MB per group: L2-resident
    f32x16 acc[2] = {};
    f16x8 xb;
#pragma unroll
    for (int j = 0; j < 8; ++j) xb[j] = (_Float16)(0.001f * (lane + j));
    f16x8 fr[4];
    u32x4 ring[D];
#pragma unroll
    for (int d = 0; d < D; ++d) ring[d] = u32x4{0, 0, 0, 0};
    if constexpr (MODE == 3) {
#pragma unroll
        for (int d = 0; d < D; ++d) asm volatile("" : "+a"(ring[d]));
    }
    unsigned long long t0;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
    for (int it = 0; it < kIters; ++it) {
        const uint32_t rd = lds + lane16 + (uint32_t)((it % 3) * 32 * 1024);
#pragma unroll
        for (int n = 0; n < 32; ++n) {
            if (n == 0) {
#pragma unroll
                for (int d = 0; d < 3; ++d) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fr[d]) : "v"(rd), "i"(d * 1024) : "memory");
            }
            if (n % 4 == 0) {                      // one stream piece
                const int i = n / 4;
                const char* g = base + (size_t)((it * 32 + wave * 8 + i) % 1408) * 1024;          // K5's piece order: chunk, wave, piece
                const uint32_t dst = lds + (uint32_t)((((it + 2) % 3) * 32 + wave * 8 + i) * 1024);
                if constexpr (MODE == 1) {
                    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(lane16), "s"(g), "s"(dst) : "memory");
                } else if constexpr (MODE == 2 || MODE == 3) {
                    // write the piece loaded D pieces ago, then reuse its registers for the new load
                    u32x4& r = ring[(n / 4) % D];
                    if constexpr (MODE == 2) {
                        asm volatile("s_waitcnt vmcnt(%2)\n\tds_write_b128 %1, %0" : "+v"(r) : "v"(dst + lane16), "n"(D - 1) : "memory");
                        asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(r) : "v"(lane16), "s"(g) : "memory");
                    } else {
                        asm volatile("s_waitcnt vmcnt(%2)\n\tds_write_b128 %1, %0" : "+a"(r) : "v"(dst + lane16), "n"(D - 1) : "memory");
                        asm volatile("global_load_dwordx4 %0, %1, %2" : "=a"(r) : "v"(lane16), "s"(g) : "memory");
                    }
                }
            }
            if (n + 3 < 32) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fr[(n + 3) % 4]) : "v"(rd), "i"(0) : "memory");
            // (the offset is immaterial for timing; all reads hit the ring)
            {
                f16x8& f = fr[n % 4];
                // reads in flight behind this one: min(3, 31 - n); stream writes issued after it are counted too (mode 2 / 3: <= 1)
                // LDS operations issued behind the read of fragment n: reads n + 1 .. n + 3, and (modes 2 / 3) the stream's ds_write of
                // steps n - 2 .. n (one of them is a multiple of 4 unless n % 4 == 3)
                if (n + 3 >= 32) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f) :: "memory");
                else if ((MODE == 2 || MODE == 3) && n % 4 != 3) asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(f) :: "memory");
                else asm volatile("s_waitcnt lgkmcnt(3)" : "+v"(f) :: "memory");
                acc[n & 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f, xb, acc[n & 1], 0, 0, 0);
            }
        }
        if constexpr (MODE == 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        asm volatile("s_barrier" ::: "memory");
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    unsigned long long t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
    float s = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) s += acc[0][r] + acc[1][r];
    if (s == 12345.678f) sink[0] = s + smem[lane];
}

template <int MODE, int D>
static void run(const char* name, const char* src, unsigned long long* out, float* sink) {
    hipFuncSetAttribute((const void*)k<MODE, D>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
    std::vector<unsigned long long> h(256);
    double best = 1e30;
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL((k<MODE, D>), dim3(256), dim3(256), 96 * 1024, 0, src, out, sink);
        hipDeviceSynchronize();
        hipMemcpy(h.data(), out, 256 * 8, hipMemcpyDeviceToHost);
        std::sort(h.begin(), h.end());
        best = std::min(best, (double)h[128] / (kIters * 32.0));
    }
    // s_memtime ticks at 100 MHz on this part: report ticks per MFMA and the ratio to mode 0 instead of trusting a clock
    std::printf("%-58s %.4f ticks per MFMA\n", name, best);
}

int main(int argc, char** argv) {
    char* src; unsigned long long* out; float* sink;
    hipMalloc(&src, 16u << 20); hipMemset(src, 0x3c, 16u << 20);
    hipMalloc(&out, 256 * 8); hipMalloc(&sink, 64);
    run<0, 4>("mode 0: fragment reads + MFMAs only", src, out, sink);
    run<1, 4>("mode 1: + LDS-DMA piece every 4th MFMA", src, out, sink);
    run<2, 2>("mode 2: + global_load / ds_write_b128 via VGPR ring, D = 2", src, out, sink);
    run<2, 4>("mode 2: + global_load / ds_write_b128 via VGPR ring, D = 4", src, out, sink);
    run<2, 8>("mode 2: + global_load / ds_write_b128 via VGPR ring, D = 8", src, out, sink);
    run<3, 4>("mode 3: + global_load / ds_write_b128 via AGPR ring, D = 4", src, out, sink);
    run<0, 4>("mode 0 again", src, out, sink);
    return 0;
}
