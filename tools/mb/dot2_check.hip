// does v_dot2c_f32_bf16 (the builtin fdot2_f32_bf16) sum a packed bf16 pair against ones as expected?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
__global__ void k(const unsigned* in, float* out) {
    const unsigned u = in[threadIdx.x];
    float a = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf2, u), __builtin_bit_cast(bf2, 0x3f803f80u), 0.5f, false);
    float b = 0.5f + __builtin_bit_cast(float, u << 16) + __builtin_bit_cast(float, u & 0xffff0000u);
    out[2 * threadIdx.x] = a; out[2 * threadIdx.x + 1] = b;
}
int main() {
    unsigned h[64]; for (int i = 0; i < 64; ++i) h[i] = (0x3f80u + 7 * i) | ((0xbf00u + 13 * i) << 16);
    unsigned* d; float* o; hipMalloc(&d, 256); hipMalloc(&o, 512); hipMemcpy(d, h, 256, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, o); float r[128]; hipMemcpy(r, o, 512, hipMemcpyDeviceToHost);
    for (int i = 0; i < 4; ++i) printf("%d: dot2 %.6f manual %.6f\n", i, r[2 * i], r[2 * i + 1]);
    return 0;
}
