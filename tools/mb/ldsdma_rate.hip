// Microbenchmark: L2 -> LDS (LDS-DMA) and L2 -> register load rate per CU, 4 waves per CU (one per SIMD), all CUs at once.
// usage: ldsdma_rate [footprint_KiB per workgroup, default 64]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void global_cvoid;
typedef float f32x4 __attribute__((ext_vector_type(4)));

// MODE 0: contiguous 1 KiB per instruction.  MODE r (1, 2, 4, 8): one instruction = r rows x (1024 / r) B of a row-major
// [128 rows][12 KiB] tile per workgroup (the x operand of K4 at K = 3072); column window cycles within the first `win` bytes of the
// rows (win = 512: L2-resident).  skew: workgroup b starts its window at (b * skew * 128) % win' (different column phase per CU).
template <int MODE>
__global__ __launch_bounds__(256, 1) void k(const char* src, size_t per_wg, int iters, unsigned long long* out, float* sink, int win, int skew) {
    extern __shared__ char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const char* base = src + (size_t)blockIdx.x * per_wg;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            size_t o;
            const int t = wave * 16 + q;                                      // 64 pieces of 1 KiB per iteration
            if (MODE == 0) o = ((size_t)it * 65536) % per_wg + t * 1024 + lane * 16;
            else {
                constexpr int R = MODE ? MODE : 1, CB = 1024 / R, LPR = 64 / R;           // rows, bytes per row, lanes per row
                // iteration covers 128 rows x 512 B: piece t -> rows (t * R) % 128 .., column block (t * R / 128) * CB
                const int row = (t * R) % 128 + lane / LPR;
                const int col = ((t * R) / 128) * CB + (lane % LPR) * 16;
                const size_t cw = ((size_t)it * 512 + (size_t)blockIdx.x * skew * 128 + col) % win;
                o = (size_t)row * 12288 + cw;
            }
            __builtin_amdgcn_global_load_lds((global_cvoid*)(base + o), (lds_void*)(uint32_t)(uintptr_t)(smem + t * 1024), 16, 0, 0);
            asm volatile("s_waitcnt vmcnt(15)" ::: "memory");
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
    if (iters < 0) sink[0] = smem[lane];
}

int main(int argc, char** argv) {
    const int win = argc > 1 ? atoi(argv[1]) : 512, skew = argc > 2 ? atoi(argv[2]) : 0;
    const size_t per_wg = 128 * 12288;
    const int nwg = 256, iters = 2000;
    char* src; unsigned long long* out; float* sink;
    hipMalloc(&src, per_wg * nwg + (1 << 20)); hipMemset(src, 1, per_wg * nwg + (1 << 20));
    hipMalloc(&out, nwg * 8); hipMalloc(&sink, 64);
    const int modes[5] = {0, 8, 4, 2, 1};
    for (int mi = 0; mi < 5; ++mi) {
        const int mode = modes[mi];
        for (int rep = 0; rep < 2; ++rep) {
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            hipEventRecord(e0);
            if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(nwg), dim3(256), 65536, 0, src, per_wg, iters, out, sink, win, skew);
            if (mode == 8) hipLaunchKernelGGL(k<8>, dim3(nwg), dim3(256), 65536, 0, src, per_wg, iters, out, sink, win, skew);
            if (mode == 4) hipLaunchKernelGGL(k<4>, dim3(nwg), dim3(256), 65536, 0, src, per_wg, iters, out, sink, win, skew);
            if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(nwg), dim3(256), 65536, 0, src, per_wg, iters, out, sink, win, skew);
            if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(nwg), dim3(256), 65536, 0, src, per_wg, iters, out, sink, win, skew);
            hipEventRecord(e1);
            if (hipDeviceSynchronize() != hipSuccess) { printf("mode %d failed\n", mode); return 1; }
            float ms; hipEventElapsedTime(&ms, e0, e1);
            std::vector<unsigned long long> h(nwg); hipMemcpy(h.data(), out, nwg * 8, hipMemcpyDeviceToHost);
            std::sort(h.begin(), h.end());
            const double bytes = 65536.0 * iters;
            if (rep) printf("rows/instr %d win %d skew %d: %.3f ms  %.1f GB/s per CU, %.2f TB/s chip; median %.1f B/clk/CU\n", mode, win, skew, ms,
                            bytes / ms / 1e6, bytes * nwg / ms / 1e9, bytes / h[nwg / 2]);
        }
    }
    return 0;
}
