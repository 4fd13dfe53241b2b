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

// ---- split-K support shared by the three tiled GEMMs: partial planes in a cached workspace + one reduce pass
// (replaces fp32 atomics: no same-address contention, and the weight gradients become bitwise reproducible).
static float* g_splitk_ws = nullptr;
static size_t g_splitk_floats = 0;

float* splitk_workspace(size_t floats) {
    if (floats > g_splitk_floats) {
        if (g_splitk_ws) {
            (void)hipDeviceSynchronize();     // rare (growth only): earlier launches may still read the old buffer
            (void)hipFree(g_splitk_ws);
        }
        const size_t want = floats < (size_t)8 << 20 ? (size_t)8 << 20 : floats;     // >= 32 MiB
        if (hipMalloc(&g_splitk_ws, want * sizeof(float)) != hipSuccess) {
            g_splitk_ws = nullptr;
            g_splitk_floats = 0;
            set_error("split-K workspace: hipMalloc of %zu bytes failed", want * sizeof(float));
            return nullptr;
        }
        g_splitk_floats = want;
    }
    return g_splitk_ws;
}

__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ ws, float* __restrict__ out, int split, long n4,
                                                            long stride4) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    const f32x4* w = reinterpret_cast<const f32x4*>(ws) + i;
    f32x4 a = w[0];
    for (int z = 1; z < split; ++z) { const f32x4 b = w[(size_t)z * stride4]; a += b; }
    reinterpret_cast<f32x4*>(out)[i] = a;
}

int launch_splitk_reduce(const float* ws, float* out, int split, long n, hipStream_t s) {
    // n = M * N is a multiple of 4 (N of the split shapes is a multiple of 128) and both buffers are 16-byte aligned
    if ((n & 3) || ((uintptr_t)out & 15)) { set_error("split-K reduce: unaligned output"); return DLDKD_EINVAL; }
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, s, ws, out, split, n / 4, n / 4);
    return check_launch("splitk_reduce");
}

}  // namespace dldkd

extern "C" {
int dldkd_abi_version(void) { return 1; }
const char* dldkd_last_error(void) { return dldkd::g_err; }
}
