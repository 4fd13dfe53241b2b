// Micro-benchmark: what bf16 MFMA rate does this chip sustain with operands in registers only?
//   mode 0: all-zero operands (minimal switching power)   mode 1: random bf16 operands
// One wave per SIMD (256 threads per CU, 1 workgroup per CU x 256 CUs x WG_PER_CU), 8 independent accumulators,
// v_mfma_f32_16x16x32_bf16 (the scorer's instruction) or v_mfma_f32_32x32x16_bf16.
// Build: hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_peak.hip -o tools/micro/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int KIND>
__global__ __launch_bounds__(256, 1) void mfma_loop(const bf16x8* __restrict__ src, float* __restrict__ out, int iters) {
    bf16x8 a[8], b[2];
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = src[(threadIdx.x + 64 * i) & 1023];
    b[0] = src[(threadIdx.x * 3 + 1) & 1023];
    b[1] = src[(threadIdx.x * 5 + 2) & 1023];
    float sum = 0.f;
    if (KIND == 16) {
        f32x4 acc[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int k = 0; k < 12; ++k)
#pragma unroll
                for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[k & 1], acc[i], 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) sum += acc[i][0] + acc[i][3];
    } else {
        f32x16 acc[4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int k = 0; k < 12; ++k)
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[k & 1], acc[i], 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) sum += acc[i][0] + acc[i][7];
    }
    if (sum == 12345.678f) out[0] = sum;   // keep the chain live
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 20000;
    std::vector<unsigned short> h(1024 * 8);
    bf16x8* d;
    float* o;
    hipMalloc(&d, h.size() * 2);
    hipMalloc(&o, 4);
    for (int mode = 0; mode < 2; ++mode) {
        for (auto& x : h) {
            if (!mode) { x = 0; continue; }
            const float f = (rand() / (float)RAND_MAX - 0.5f) * 0.25f;   // small values: the accumulators stay finite
            unsigned u; memcpy(&u, &f, 4); x = (unsigned short)(u >> 16);
        }
        hipMemcpy(d, h.data(), h.size() * 2, hipMemcpyHostToDevice);
        for (int kind : {16, 32}) {
            hipEvent_t e0, e1;
            hipEventCreate(&e0); hipEventCreate(&e1);
            const dim3 grid(256 * 4), block(256);
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0);
                if (kind == 16) hipLaunchKernelGGL(mfma_loop<16>, grid, block, 0, 0, d, o, iters);
                else hipLaunchKernelGGL(mfma_loop<32>, grid, block, 0, 0, d, o, iters);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
            }
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double mfmas = (double)grid.x * 4 * iters * 12 * (kind == 16 ? 8 : 4);
            const double flops = mfmas * (kind == 16 ? 16384.0 : 32768.0);
            printf("%s operands, mfma_%s: %.2f ms  %.0f TFLOP/s (%.1f%% of 2500)\n", mode ? "random" : "zero  ",
                   kind == 16 ? "16x16x32" : "32x32x16", ms, flops / ms / 1e9, flops / ms / 1e9 / 25.0);
        }
    }
    return 0;
}
