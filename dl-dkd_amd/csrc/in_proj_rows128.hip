// K4, second generation: y = ReLU( LayerNorm(x) . W^T + b ) for the raw clip / word features, both branches (768 output
// columns) in ONE pass over the fp32 rows.  Replaces LinearLayer.forward (reference method/model_components.py:305-312) on the
// inference path, like rows_linear_bf16_kernel<1, true> (in_proj_bf16.hip), whose limits this kernel is built around
// (profiles/r01/ablation_k4_in_proj.md): both MFMA operands came from LDS (0.58 fragment reads per MFMA: an LDS
// operand-delivery ceiling of ~834 TFLOP/s raw, no memory traffic needed to hit it) and 8 waves marched in lock-step around one
// barrier per k-tile.  The kernel needs ~1.1 PFLOP/s to move 3.6 TB/s (307 flop per byte).
//
//   * 4 waves per workgroup, ONE per SIMD, NO barrier in the k-loop.  Wave w owns ALL 128 rows of the workgroup x columns
//     [192 w, 192 w + 192): 4 x 6 tiles of mfma_f32_32x32x16_bf16 = 384 accumulator registers - more than the 256 AGPRs, and
//     hipcc allocates every MFMA of a kernel in one register class (a test compile copied 128 registers in and out of the AGPR
//     half every k-step).  So the MFMAs are inline asm: the accumulators of column tiles 0-3 are constrained to AGPRs ("+a"),
//     those of column tiles 4-5 to arch VGPRs ("+v"): 256 + 128, no copies.
//   * the x operand never touches LDS: each lane loads the 8 consecutive fp32 of its A-fragment row straight from global memory
//     (two dwordx4), converts them to bf16 in registers (v_cvt_pk_bf16_f32) and keeps them for the 6 column tiles of the k-step.
//     The four waves load the same rows (L1 / L2 serve three of the four); rows are loaded ONE k-step ahead into a second
//     register set, the loads hidden from hipcc in asm so that it does not drain the whole VMEM queue around the W' LDS-DMA
//     (cdna_hip_programming.md section 5, trap (b)) - one hand-counted s_waitcnt per k-step.
//   * W' (LayerNorm-folded weights, bf16, MFMA B-fragment order) streams L2 -> LDS by LDS-DMA into a PRIVATE 3-slot ring per
//     wave (12 KiB per k-step: exactly the 6 column tiles the wave uses), two k-steps ahead; each 1-KiB fragment read from LDS
//     feeds 4 MFMAs (0.25 fragment reads per MFMA).
//   * LayerNorm is folded as before: out = rstd (x.W'^T - mean colsum(W')) + (W.beta + b); wave w accumulates sum / sum of
//     squares of rows 32 w .. 32 w + 31 from the fp32 values it converts anyway; one barrier after the k-loop publishes them.
//   * epilogue: 32 x 192 tiles staged through the wave's (now idle) LDS ring and written as float4 rows.
#include <type_traits>

#include "common.hpp"

namespace dldkd {

constexpr int RM = 128, RK = 32, RWC = 192;
constexpr int RSLOT = 6 * 2 * 1024;       // bytes of W' per wave per k-step: [6 column tiles][2 kk][64 lanes][16 B]
constexpr int RRING = 3;
constexpr int RSP = 200;                  // epilogue staging pitch (floats): the two lane halves (rows +4) hit disjoint banks
constexpr int RW_TILE = 768 * RK * 2;     // bytes of W' per k-step for all 24 column tiles

struct Rows128Args {
    const float* x;
    const char* Wf;        // [k-step][24 column tiles][2][64][8] bf16
    const float* cs;       // [768] colsum of W'
    const float* bb;       // [768] W.beta + b
    float* y[2];           // columns [0, 384) -> y[0], [384, 768) -> y[1]; row stride 384
    long M;
    int K;
    float eps;
    int relu;
};

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2n __attribute__((ext_vector_type(2)));
// 8 fp32 -> one bf16 MFMA fragment with 4 v_cvt_pk_bf16_f32 (element-wise casts made hipcc convert singly and pack with v_perm)
__device__ __forceinline__ bf16x8 pack8(const f32x4& lo, const f32x4& hi) {
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    u32x4 u;
    u[0] = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{lo[0], lo[1]}, bf16x2n));
    u[1] = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{lo[2], lo[3]}, bf16x2n));
    u[2] = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{hi[0], hi[1]}, bf16x2n));
    u[3] = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{hi[2], hi[3]}, bf16x2n));
    return __builtin_bit_cast(bf16x8, u);
}

template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

__device__ __forceinline__ void mfma_agpr(f32x16& acc, const bf16x8& a, const bf16x8& b) {
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
}
__device__ __forceinline__ void mfma_vgpr(f32x16& acc, const bf16x8& a, const bf16x8& b) {
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
}
template <int OFF>
__device__ __forceinline__ void lds_frag(bf16x8& dst, uint32_t addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "i"(OFF) : "memory");
}
template <int OFF>
__device__ __forceinline__ void xload(f32x4& dst, uint32_t voff, const float* base) {
    asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" : "=v"(dst) : "v"(voff), "s"(base), "i"(OFF) : "memory");
}

__global__ __launch_bounds__(256, 1) void in_proj_rows128_kernel(const Rows128Args p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const long m0 = (long)blockIdx.x * RM;
    const int nk = p.K / RK;
    char* ring = smem + wave * (RRING * RSLOT);                       // this wave's private W' ring
    float* s_mean = reinterpret_cast<float*>(smem + 4 * RRING * RSLOT);
    float* s_rstd = s_mean + RM;
    const uint32_t ring_lds = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(ring)) + lane * 16;
    const char* wsrc = p.Wf + (size_t)wave * RSLOT + lane * 16;       // + k-step * RW_TILE + piece * 1024

    // per-lane byte offsets of the 4 row tiles (rows past M are clamped: they feed accumulator rows that are never stored)
    uint32_t voff[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        long row = m0 + 32 * i + (lane & 31);
        if (row > p.M - 1) row = p.M - 1;
        voff[i] = (uint32_t)((row - m0) * p.K * 4 + (lane >> 5) * 32);
    }
    const float* xb = p.x + m0 * p.K;                                  // wave-uniform, advanced by 32 floats per k-step

    f32x16 acc[4][6];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 6; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    bf16x8 a[4][2];          // A fragments of the current k-step: row tile i, kk
    f32x4 raw[4][2][2];      // fp32 of the NEXT k-step in flight: row tile i, kk, half of the 8 floats
    float sum = 0.f, sq = 0.f;

    auto stage = [&](int kt) {                                        // 12 x 1 KiB LDS-DMA pieces into slot kt % 3
        const char* src = wsrc + (size_t)kt * RW_TILE;
        char* dst = ring + (kt % RRING) * RSLOT;
#pragma unroll
        for (int i = 0; i < 12; ++i) glds16(src + i * 1024, dst + i * 1024);
    };
    auto issue_x = [&]() {
        static_for<0, 4>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            xload<0>(raw[i][0][0], voff[i], xb);
            xload<16>(raw[i][0][1], voff[i], xb);
            xload<64>(raw[i][1][0], voff[i], xb);
            xload<80>(raw[i][1][1], voff[i], xb);
        });
    };
    auto convert = [&]() {                                            // raw -> bf16 fragments (+ LayerNorm sums of the wave's own row tile)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                a[i][kk] = pack8(raw[i][kk][0], raw[i][kk][1]);
                if (wave == i) {                                      // wave-uniform
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const float v = raw[i][kk][e >> 2][e & 3];
                        sum += v;
                        sq += v * v;
                    }
                }
            }
    };
    auto wait_x = [&]() {          // every VMEM operation issued so far has landed (x of the next k-step, W' two steps ahead)
        asm volatile("s_waitcnt vmcnt(0)"
                     : "+v"(raw[0][0][0]), "+v"(raw[0][0][1]), "+v"(raw[0][1][0]), "+v"(raw[0][1][1]), "+v"(raw[1][0][0]),
                       "+v"(raw[1][0][1]), "+v"(raw[1][1][0]), "+v"(raw[1][1][1]), "+v"(raw[2][0][0]), "+v"(raw[2][0][1]),
                       "+v"(raw[2][1][0]), "+v"(raw[2][1][1]), "+v"(raw[3][0][0]), "+v"(raw[3][0][1]), "+v"(raw[3][1][0]),
                       "+v"(raw[3][1][1])
                     :
                     : "memory");
    };

    // prologue
    stage(0);
    if (nk > 1) stage(1);
    issue_x();
    xb += RK;
    wait_x();
    convert();

    for (int kt = 0; kt < nk; ++kt) {
        if (kt + 2 < nk) stage(kt + 2);
        if (kt + 1 < nk) { issue_x(); xb += RK; }
        const uint32_t slot = ring_lds + (kt % RRING) * RSLOT;
        // 12 B fragments (column tile j, kk) in order f = 2 j + kk (their order in the slot); fragment f + 1 is read while the
        // 4 MFMAs of fragment f run (128 cycles against ~100 of LDS latency); a 3-deep ring cost 4 registers the kernel lacks
        bf16x8 b[2];
        lds_frag<0>(b[0], slot);
        static_for<0, 12>([&](auto fc) {
            constexpr int f = decltype(fc)::value;
            constexpr int j = f >> 1, kk = f & 1;
            if constexpr (f + 1 < 12) {
                lds_frag<(f + 1) * 1024>(b[(f + 1) & 1], slot);
                asm volatile("s_waitcnt lgkmcnt(1)" : "+v"(b[f & 1]) : : "memory");
            } else {
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(b[f & 1]) : : "memory");
            }
            if constexpr (j < 4) {
                mfma_agpr(acc[0][j], a[0][kk], b[f & 1]);
                mfma_agpr(acc[1][j], a[1][kk], b[f & 1]);
                mfma_agpr(acc[2][j], a[2][kk], b[f & 1]);
                mfma_agpr(acc[3][j], a[3][kk], b[f & 1]);
            } else {
                mfma_vgpr(acc[0][j], a[0][kk], b[f & 1]);
                mfma_vgpr(acc[1][j], a[1][kk], b[f & 1]);
                mfma_vgpr(acc[2][j], a[2][kk], b[f & 1]);
                mfma_vgpr(acc[3][j], a[3][kk], b[f & 1]);
            }
        });
        if (kt + 1 < nk) {
            wait_x();
            convert();
            asm volatile("s_nop 1");        // VALU-written fragments -> MFMA operands (cdna_hip_programming.md 5.7 item 2)
        }
    }
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");               // last MFMA results -> VALU readers below

    // LayerNorm statistics of the wave's own row tile: the two lane halves hold the two 8-float chunks of every 16 k
    sum += __shfl_xor(sum, 32);
    sq += __shfl_xor(sq, 32);
    if (lane < 32) {
        const float mean = sum / p.K;
        const float var = fmaxf(sq / p.K - mean * mean, 0.f);
        s_mean[32 * wave + lane] = mean;
        s_rstd[32 * wave + lane] = rsqrtf(var + p.eps);
    }
    __syncthreads();

    // epilogue: wave w writes columns [192 w, 192 w + 192) = branch w / 2, columns (w & 1) * 192 ..
    float* stg = reinterpret_cast<float*>(ring);                     // 32 x RSP floats = 25.6 KiB of the wave's 36 KiB
    float* outb = p.y[wave >> 1] + (wave & 1) * RWC;
    float csn[6], bbn[6];
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        const int n = wave * RWC + 32 * j + (lane & 31);
        csn[j] = p.cs[n];
        bbn[j] = p.bb[n];
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int j = 0; j < 6; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int rl = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                const int ml = 32 * i + rl;
                float v = s_rstd[ml] * (acc[i][j][r] - s_mean[ml] * csn[j]) + bbn[j];
                if (p.relu) v = fmaxf(v, 0.f);
                stg[rl * RSP + 32 * j + (lane & 31)] = v;
            }
        // the wave's own staging region: no workgroup barrier needed, only the wave's own LDS writes must have landed
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int it = 0; it < 24; ++it) {
            const int idx = lane + 64 * it;                          // 32 rows x 48 float4
            const int rl = idx / 48, c4 = idx % 48;
            const long mrow = m0 + 32 * i + rl;
            const f32x4 v = *reinterpret_cast<const f32x4*>(stg + rl * RSP + 4 * c4);
            if (mrow < p.M) *reinterpret_cast<f32x4*>(outb + (size_t)mrow * kHidden + 4 * c4) = v;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");           // reads done before the next tile overwrites the staging
    }
}

}  // namespace dldkd

using namespace dldkd;

extern "C" int dldkd_in_proj_bf16_rows128(const float* x, const void* Wfrag, const float* cs, const float* bb, float* y0, float* y1,
                                          long M, int K, float eps, int relu, void* stream) {
    if (M < 0 || K < RK || (K % RK) || (long)127 * K * 4 + 64 > 0xFFFFFFFFL) {
        set_error("in_proj_bf16_rows128: K must be a multiple of %d (M=%ld K=%d)", RK, M, K);
        return DLDKD_EINVAL;
    }
    if (M == 0) return DLDKD_OK;
    if (!x || !Wfrag || !cs || !bb || !y0 || !y1) { set_error("in_proj_bf16_rows128: null pointer"); return DLDKD_EINVAL; }
    if (((uintptr_t)x | (uintptr_t)y0 | (uintptr_t)y1) & 15) { set_error("in_proj_bf16_rows128: unaligned buffer"); return DLDKD_EINVAL; }
    Rows128Args p{x, (const char*)Wfrag, cs, bb, {y0, y1}, M, K, eps, relu};
    constexpr int lds = 4 * RRING * RSLOT + 2 * RM * 4;
    static const bool ok = hipFuncSetAttribute((const void*)in_proj_rows128_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds) == hipSuccess;
    (void)ok;
    DLDKD_LAUNCH(in_proj_rows128_kernel, dim3((unsigned)((M + RM - 1) / RM)), dim3(256), lds, (hipStream_t)stream, p);
    return check_launch("in_proj_bf16_rows128");
}
