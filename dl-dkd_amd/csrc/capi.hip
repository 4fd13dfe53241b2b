// Error plumbing + ABI version for libdldkd_hip.so (see include/dldkd_hip.h).
#include <stdarg.h>
#include <stdio.h>

#include "common.hpp"

namespace dldkd {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("%s: %s", what, hipGetErrorString(e));
        return DLDKD_ELAUNCH;
    }
    return DLDKD_OK;
}

// ---- split-K support shared by the three tiled GEMMs: partial planes in a CALLER-PROVIDED workspace + one reduce pass
// (no fp32 atomics: no same-address contention, and the weight gradients are bitwise reproducible).  The library never
// allocates, frees or synchronises: every entry point only enqueues (include/dldkd_hip.h, Conventions).
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ ws, float* __restrict__ out, int split, long n4,
                                                            long stride4) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    const f32x4* w = reinterpret_cast<const f32x4*>(ws) + i;
    f32x4 a = w[0];
#pragma unroll 8
    for (int z = 1; z < split; ++z) { const f32x4 b = w[(size_t)z * stride4]; a += b; }
    reinterpret_cast<f32x4*>(out)[i] = a;
}

int launch_splitk_reduce(const float* ws, float* out, int split, long n, hipStream_t s) {
    // n = M * N is a multiple of 4 (N of the split shapes is a multiple of 128) and both buffers are 16-byte aligned
    if ((n & 3) || ((uintptr_t)out & 15)) { set_error("split-K reduce: unaligned output"); return DLDKD_EINVAL; }
    DLDKD_LAUNCH(splitk_reduce_kernel, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, s, ws, out, split, n / 4, n / 4);
    return check_launch("splitk_reduce");
}

}  // namespace dldkd

extern "C" {
size_t dldkd_gemm_workspace_bytes(int precision, int M, int N, int K, int a_kmajor, int b_kmajor) {
    if (M <= 0 || N <= 0 || K <= 0) return 0;
    int per = 0, split = 1;
    switch (precision) {
        case DLDKD_GEMM_F32: split = dldkd::gemm_f32_split_plan(M, N, K, a_kmajor, b_kmajor, &per); break;
        case DLDKD_GEMM_F32X3: split = dldkd::gemm_f32x3_split_plan(M, N, K, a_kmajor, b_kmajor, &per); break;
        case DLDKD_GEMM_BF16: split = dldkd::gemm_bf16_split_plan(M, N, K, a_kmajor, b_kmajor, &per); break;
        default: return 0;
    }
    return split > 1 ? (size_t)split * M * N * sizeof(float) : 0;
}
int dldkd_abi_version(void) { return 7; }
const char* dldkd_last_error(void) { return dldkd::g_err; }
}
