// Fused self-attention on bf16 MFMA (throughput mode of BertSelfAttention.forward, model_components.py:398-436):
// the same "swapped" structure as attention_fwd_f32 (encoder_f32.hip) - S^T = K Q^T with keys on accumulator
// registers and queries on lanes, softmax over keys in registers, the probabilities fed back as the B operand of
// O^T = V^T P^T - with v_mfma_f32_32x32x16_bf16 instead of the fp32-input 32x32x2 (16x fewer MFMA cycles).
//   * Q, K are staged as bf16 [row][d] (208-byte rows: conflict-free ds_read_b128 fragments); V is staged
//     TRANSPOSED, bf16 [d][key], because the second product contracts over keys (8 consecutive keys per lane);
//   * an accumulator lane half holds keys {4h..4h+3, 8+4h..} of each 16-key step; one v_permlane32_swap per packed
//     register pair turns that into the 8 consecutive keys {8h..8h+7} the B fragment wants;
//   * softmax statistics and the output stay fp32; operands (Q, K, V, P) are rounded to bf16.
#include "common.hpp"

namespace dldkd {

constexpr int aHeads = 4, aDh = 96, aLmax = 128;
constexpr int aQKP = aDh + 8;        // Q/K pitch (bf16 elements)

__device__ __forceinline__ float a_half_swap_max(float m) {
    const unsigned u = __builtin_bit_cast(unsigned, m);
    auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    return fmaxf(__builtin_bit_cast(float, (unsigned)r[0]), __builtin_bit_cast(float, (unsigned)r[1]));
}
__device__ __forceinline__ float a_half_swap_sum(float m) {
    const unsigned u = __builtin_bit_cast(unsigned, m);
    auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    return __builtin_bit_cast(float, (unsigned)r[0]) + __builtin_bit_cast(float, (unsigned)r[1]);
}
__device__ __forceinline__ unsigned pack2(float a, float b) {
    return (unsigned)f32_to_bf16_bits(a) | ((unsigned)f32_to_bf16_bits(b) << 16);
}

template <int NKT, bool QKV16>   // key tiles of 32; QKV16: q|k|v arrive as bf16 (written by linear_rows with out_bf16)
__device__ __forceinline__ void attention_bf16_body(const void* __restrict__ qkv_v, const float* __restrict__ mask,
                                                    float* __restrict__ out, int L, char* smem) {
    constexpr int LP = NKT * 32;
    constexpr int VP = LP + 8;       // V^T pitch (keys per d row)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = blockIdx.x / aHeads, head = blockIdx.x % aHeads;
    unsigned short* Qs = reinterpret_cast<unsigned short*>(smem);   // [LP][aQKP]
    unsigned short* Ks = Qs + LP * aQKP;                            // [LP][aQKP]
    unsigned short* Vt = Ks + LP * aQKP;                            // [96][VP]
    float* Ms = reinterpret_cast<float*>(Vt + aDh * VP);            // [LP] additive key mask
    for (int i = tid; i < LP * 24; i += 256) {
        const int row = i / 24, c = i % 24;
        uint2 pq = {0u, 0u}, pk = pq, pv = pq;
        if (row < L) {
            const size_t off = ((size_t)n * L + row) * (3 * 384) + head * aDh + c * 4;
            if constexpr (QKV16) {
                const unsigned short* r = reinterpret_cast<const unsigned short*>(qkv_v) + off;
                pq = *reinterpret_cast<const uint2*>(r);
                pk = *reinterpret_cast<const uint2*>(r + 384);
                pv = *reinterpret_cast<const uint2*>(r + 768);
            } else {
                const float* r = reinterpret_cast<const float*>(qkv_v) + off;
                const f32x4 q = *reinterpret_cast<const f32x4*>(r);
                const f32x4 k = *reinterpret_cast<const f32x4*>(r + 384);
                const f32x4 v = *reinterpret_cast<const f32x4*>(r + 768);
                pq.x = pack2(q[0], q[1]); pq.y = pack2(q[2], q[3]);
                pk.x = pack2(k[0], k[1]); pk.y = pack2(k[2], k[3]);
                pv.x = pack2(v[0], v[1]); pv.y = pack2(v[2], v[3]);
            }
        }
        *reinterpret_cast<uint2*>(Qs + row * aQKP + c * 4) = pq;
        *reinterpret_cast<uint2*>(Ks + row * aQKP + c * 4) = pk;
        Vt[(c * 4 + 0) * VP + row] = (unsigned short)(pv.x & 0xFFFFu);
        Vt[(c * 4 + 1) * VP + row] = (unsigned short)(pv.x >> 16);
        Vt[(c * 4 + 2) * VP + row] = (unsigned short)(pv.y & 0xFFFFu);
        Vt[(c * 4 + 3) * VP + row] = (unsigned short)(pv.y >> 16);
    }
    for (int i = tid; i < LP; i += 256)
        Ms[i] = i < L ? (mask ? (1.f - mask[(size_t)n * L + i]) * -10000.f : 0.f) : -INFINITY;
    __syncthreads();

    const int q0 = wave * 32;
    if (q0 >= L) return;
    // S^T[key][query]: A = K rows (keys), B = Q rows (queries), contraction over d in 6 steps of 16
    f32x16 s[NKT];
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
        for (int r = 0; r < 16; ++r) s[kt][r] = 0.f;
    const unsigned short* qrow = Qs + (q0 + (lane & 31)) * aQKP + (lane >> 5) * 8;
    const unsigned short* krow = Ks + (lane & 31) * aQKP + (lane >> 5) * 8;
#pragma unroll
    for (int ks = 0; ks < aDh / 16; ++ks) {
        const bf16x8 b = *reinterpret_cast<const bf16x8*>(qrow + ks * 16);
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt) {
            const bf16x8 a = *reinterpret_cast<const bf16x8*>(krow + kt * 32 * aQKP + ks * 16);
            s[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, s[kt], 0, 0, 0);
        }
    }
    const float scale = 0.10206207261596577f;   // 1/sqrt(96), model_components.py:419
    float mx = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            s[kt][r] = s[kt][r] * scale + Ms[key];
            mx = fmaxf(mx, s[kt][r]);
        }
    mx = a_half_swap_max(mx);
    float sum = 0.f;
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            s[kt][r] = expf(s[kt][r] - mx);
            sum += s[kt][r];
        }
    sum = a_half_swap_sum(sum);
    const float inv = 1.f / sum;

    // O^T[d][query] = sum_key V^T[d][key] P[key][query]
    f32x16 o[3];
#pragma unroll
    for (int dt = 0; dt < 3; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[dt][r] = 0.f;
    const unsigned short* vrow = Vt + (lane & 31) * VP + (lane >> 5) * 8;
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
        for (int k2 = 0; k2 < 2; ++k2) {
            // registers 8*k2 .. 8*k2+7 of tile kt: this lane half holds keys {4h..4h+3} (first four) and {8+4h..} (last
            // four) of the 16-key step; after the swap the lower half holds keys 0..7, the upper half keys 8..15
            unsigned x0 = pack2(s[kt][8 * k2 + 0], s[kt][8 * k2 + 1]), x1 = pack2(s[kt][8 * k2 + 2], s[kt][8 * k2 + 3]);
            unsigned y0 = pack2(s[kt][8 * k2 + 4], s[kt][8 * k2 + 5]), y1 = pack2(s[kt][8 * k2 + 6], s[kt][8 * k2 + 7]);
            auto r0 = __builtin_amdgcn_permlane32_swap(x0, y0, false, false);   // x' = (x.lo, y.lo), y' = (x.hi, y.hi)
            auto r1 = __builtin_amdgcn_permlane32_swap(x1, y1, false, false);
            union { unsigned u[4]; bf16x8 v; } pb;
            pb.u[0] = r0[0]; pb.u[1] = r1[0]; pb.u[2] = r0[1]; pb.u[3] = r1[1];
#pragma unroll
            for (int dt = 0; dt < 3; ++dt) {
                const bf16x8 a = *reinterpret_cast<const bf16x8*>(vrow + dt * 32 * VP + kt * 32 + k2 * 16);
                o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, pb.v, o[dt], 0, 0, 0);
            }
        }
    const int q = q0 + (lane & 31);
    if (q < L) {
        float* orow = out + ((size_t)n * L + q) * 384 + head * aDh;
#pragma unroll
        for (int dt = 0; dt < 3; ++dt)
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = o[dt][r4 * 4 + e] * inv;
                *reinterpret_cast<f32x4*>(orow + dt * 32 + 8 * r4 + 4 * (lane >> 5)) = v;
            }
    }
}

template <bool QKV16>
__global__ __launch_bounds__(256) void attention_fwd_bf16_kernel(const void* __restrict__ qkv, const float* __restrict__ mask,
                                                                 float* __restrict__ out, int L) {
    extern __shared__ __attribute__((aligned(16))) char smem_a[];
    const int nkt = (L + 31) >> 5;
    switch (nkt) {
        case 1: attention_bf16_body<1, QKV16>(qkv, mask, out, L, smem_a); break;
        case 2: attention_bf16_body<2, QKV16>(qkv, mask, out, L, smem_a); break;
        case 3: attention_bf16_body<3, QKV16>(qkv, mask, out, L, smem_a); break;
        default: attention_bf16_body<4, QKV16>(qkv, mask, out, L, smem_a); break;
    }
}

}  // namespace dldkd

using namespace dldkd;

extern "C" int dldkd_attention_fwd_bf16(const void* qkv, const float* mask, float* out, int N, int L, int qkv_is_bf16, void* stream) {
    if (N < 0 || L < 1 || L > aLmax) { set_error("attention_bf16: bad sizes N=%d L=%d (L <= %d)", N, L, aLmax); return DLDKD_EINVAL; }
    if (N == 0) return DLDKD_OK;
    if (!qkv || !out) { set_error("attention_bf16: null pointer"); return DLDKD_EINVAL; }
    const int LP = ((L + 31) / 32) * 32;
    const size_t lds = (size_t)(2 * LP * aQKP + aDh * (LP + 8)) * 2 + (size_t)LP * 4;
    static const bool attr_ok = [] {
        constexpr int mx = (2 * aLmax * aQKP + aDh * (aLmax + 8)) * 2 + aLmax * 4;
        return hipFuncSetAttribute((const void*)attention_fwd_bf16_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, mx) == hipSuccess &&
               hipFuncSetAttribute((const void*)attention_fwd_bf16_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, mx) == hipSuccess;
    }();
    (void)attr_ok;
    if (qkv_is_bf16) DLDKD_LAUNCH(attention_fwd_bf16_kernel<true>, dim3(N * aHeads), dim3(256), lds, (hipStream_t)stream, qkv, mask, out, L);
    else DLDKD_LAUNCH(attention_fwd_bf16_kernel<false>, dim3(N * aHeads), dim3(256), lds, (hipStream_t)stream, qkv, mask, out, L);
    return check_launch("attention_fwd_bf16");
}
