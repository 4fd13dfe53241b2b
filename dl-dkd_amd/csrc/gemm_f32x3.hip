// gemm_f32x3: fp32-grade GEMM on the bf16 matrix cores.  Same contract as gemm_f32 (three operand layouts, fused
// bias + ReLU, strided batching, split-K), but every fp32 operand is split on the way to LDS into three bf16 planes
//     x = h + m + l,   h = bf16(x),  m = bf16(x - h),  l = bf16(x - h - m)      (24 mantissa bits, both subtractions exact)
// and each product is assembled from the six plane products of order <= 2
//     a*b ~= a_h b_h + (a_h b_m + a_m b_h) + (a_h b_l + a_l b_h + a_m b_m)       (dropped: a_m b_l, a_l b_m, a_l b_l <= 2^-24 |ab|)
// with v_mfma_f32_32x32x16_bf16 accumulating in fp32.  Each plane product is exact in fp32 (8 x 8 significant bits),
// so the result carries the same ~2^-24 relative error per product as a true fp32 multiply, while 6 bf16 MFMAs
// (6 x 32 cycles per 16 k) replace 8 fp32-input MFMAs (8 x 64 cycles): the fp32-input pipe of CDNA4 peaks at 157 TFLOP/s,
// the bf16 pipe at 2500, so the "fp32-equivalent" MFMA ceiling moves to 2500/6 = 417 TFLOP/s.
// Measured: 105-139 TFLOP/s fp32-equivalent on the tower shapes (= 630-834 TFLOP/s of raw bf16 MFMA) against 68-88 for
// the true fp32-input MFMA kernel.  The ceiling is LDS operand delivery, not memory or VALU: with ALL global loads
// removed the rate is unchanged, interleaving the split VALU with the MFMAs (sched_group_barrier) changes nothing, and
// 834 TFLOP/s raw is exactly where the K4 kernel (same 0.5 fragment reads per MFMA) also stops.
// This is the PARITY-grade GEMM of the towers and of the training step (losses within 1e-4, gradients 2e-3 of the
// fp64 oracle are gated by the same tests as before); gemm_f32 (true fp32 MFMA) stays available as "fp32_exact".
#include <stdlib.h>

#include <type_traits>

#include "common.hpp"

namespace dldkd {

constexpr int XBM = 128, XBN = 128, XBK = 16;
constexpr int XPITCH = XBK + 8;                 // 48-byte rows: conflict-free ds_read_b128 fragments
constexpr int XPLANE = XBM * XPITCH;            // bf16 elements per plane
constexpr int XKV = XBK / 4;                    // float4 per tile row (4)
constexpr int XRPP = 256 / XKV;                 // rows per pass of the k-minor loader (64)
constexpr int XNPASS = XBM / XRPP;              // 2
constexpr int XNREG = XNPASS * 4;               // 8 staging floats per thread per operand
constexpr int XKPT = XBK / 2;                   // k per thread of the k-major loader (8)

struct GemmXArgs {
    const float* A;
    const float* B;
    const float* bias;
    float* C;
    int M, N, K, lda, ldb, ldc, relu;
    int a_vec, b_vec;
    int batch_inner;
    long sAo, sAi, sBo, sBi, sCo, sCi;
    float alpha;
    int split_k, k_tiles_per_split;
    const unsigned char* mflags;   // per 32 rows of A / C (k-minor A, M % 128 == 0) or null: 0 = rows of the padding - not multiplied
    const unsigned char* kflags;   // per 32 contraction rows (dW layout) or null: 0 = zero rows - the k-tiles are skipped
};

// x -> (h, m, l) bf16 bit patterns
__device__ __forceinline__ void split3(float x, unsigned short& h, unsigned short& m, unsigned short& l) {
    h = f32_to_bf16_bits(x);
    const float r1 = x - bf16_bits_to_f32(h);
    m = f32_to_bf16_bits(r1);
    const float r2 = r1 - bf16_bits_to_f32(m);
    l = f32_to_bf16_bits(r2);
}

template <bool KMAJOR>
struct TileX {
    static __device__ __forceinline__ void load(const float* __restrict__ P, int ld, int row0, int nrows, int k0, int K,
                                                int tid, bool vec, float (&r)[XNREG]) {
        if constexpr (!KMAJOR) {
#pragma unroll
            for (int j = 0; j < XNPASS; ++j) {
                const int row = row0 + tid / XKV + XRPP * j;
                const int k = k0 + (tid % XKV) * 4;
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (row < nrows) {
                    const float* src = P + (size_t)row * ld + k;
                    if (vec && k + 3 < K) v = *reinterpret_cast<const f32x4*>(src);
                    else {
#pragma unroll
                        for (int e = 0; e < 4; ++e) if (k + e < K) v[e] = src[e];
                    }
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) r[4 * j + e] = v[e];
            }
        } else {
            const int row = row0 + (tid & 127);
            const int kb = k0 + (tid >> 7) * XKPT;
            const float* src = P + (size_t)kb * ld + row;
            const bool rok = row < nrows;
#pragma unroll
            for (int i = 0; i < XKPT; ++i) r[i] = (rok && kb + i < K) ? src[(size_t)i * ld] : 0.f;
        }
    }
    // interior tile: branch-free; rows past the end are clamped (they only feed accumulators that are never stored)
    static __device__ __forceinline__ void load_fast(const float* __restrict__ P, int ld, int row0, int nrows, int k0,
                                                     int tid, float (&r)[XNREG]) {
        if constexpr (!KMAJOR) {
#pragma unroll
            for (int j = 0; j < XNPASS; ++j) {
                const int row = min(row0 + tid / XKV + XRPP * j, nrows - 1);
                const f32x4 v = *reinterpret_cast<const f32x4*>(P + (size_t)row * ld + k0 + (tid % XKV) * 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) r[4 * j + e] = v[e];
            }
        } else {
            const int row = min(row0 + (tid & 127), nrows - 1);
            const float* src = P + (size_t)(k0 + (tid >> 7) * XKPT) * ld + row;
#pragma unroll
            for (int i = 0; i < XKPT; ++i) r[i] = src[(size_t)i * ld];
        }
    }
    // registers -> three (NPL = 2: two) bf16 planes in LDS ([plane][row][k])
    template <int NPL = 3>
    static __device__ __forceinline__ void store(unsigned short* __restrict__ S, int tid, const float (&r)[XNREG]) {
        if constexpr (!KMAJOR) {
#pragma unroll
            for (int j = 0; j < XNPASS; ++j) {
                unsigned short h[4], m[4], l[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) split3(r[4 * j + e], h[e], m[e], l[e]);
                unsigned short* dst = S + (tid / XKV + XRPP * j) * XPITCH + (tid % XKV) * 4;
                *reinterpret_cast<uint2*>(dst) = uint2{(unsigned)h[0] | ((unsigned)h[1] << 16), (unsigned)h[2] | ((unsigned)h[3] << 16)};
                *reinterpret_cast<uint2*>(dst + XPLANE) = uint2{(unsigned)m[0] | ((unsigned)m[1] << 16), (unsigned)m[2] | ((unsigned)m[3] << 16)};
                if constexpr (NPL == 3) *reinterpret_cast<uint2*>(dst + 2 * XPLANE) = uint2{(unsigned)l[0] | ((unsigned)l[1] << 16), (unsigned)l[2] | ((unsigned)l[3] << 16)};
            }
        } else {
            unsigned short h[8], m[8], l[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) split3(r[e], h[e], m[e], l[e]);
            unsigned short* dst = S + (tid & 127) * XPITCH + (tid >> 7) * XKPT;
            auto pk = [](const unsigned short (&v)[8]) {
                return uint4{(unsigned)v[0] | ((unsigned)v[1] << 16), (unsigned)v[2] | ((unsigned)v[3] << 16),
                             (unsigned)v[4] | ((unsigned)v[5] << 16), (unsigned)v[6] | ((unsigned)v[7] << 16)};
            };
            *reinterpret_cast<uint4*>(dst) = pk(h);
            *reinterpret_cast<uint4*>(dst + XPLANE) = pk(m);
            if constexpr (NPL == 3) *reinterpret_cast<uint4*>(dst + 2 * XPLANE) = pk(l);
        }
    }
};

// EPI: 0 = store C, 1 = training simpool max-pool (PoolArgs), 2 = LayerNorm parameter gradients (LnGradArgs)
// NPL = 2 ("fp32x2", the forward pass of the "mixed" training precision): two planes per operand, x ~= h + m (16 mantissa bits,
// residual <= 2^-17 |x|), products a_h b_h + a_h b_m + a_m b_h (dropped: a_m b_m <= 2^-16 |ab|): three MFMAs instead of six, error
// ~2^-16 per product against the three-plane scheme's 2^-24 - measured on the step's seven losses: tests/test_train_mode_gpu.py.
template <bool A_KMAJOR, bool B_KMAJOR, int EPI, typename EArgs, int NPL = 3>
__device__ __forceinline__ void gemm_f32x3_body(GemmXArgs p, const EArgs* pa) {
    __shared__ __attribute__((aligned(16))) unsigned short lds[2][2][3 * XPLANE];   // [stage][A|B][plane][row][k]: 72 KiB (the epilogues stage through all of it)
    const Tile3 bid = xcd_tile_order();              // column tiles of a row block (split-K: row tiles of a B slab) behind one L2: gemm_bf16.hip
    if (p.split_k > 1) {
        p.C += (size_t)bid.z * p.M * p.ldc;      // this split's partial plane in the workspace
    } else {
        const int zo = bid.z / p.batch_inner, zi = bid.z % p.batch_inner;
        p.A += zo * p.sAo + zi * p.sAi;
        p.B += zo * p.sBo + zi * p.sBi;
        p.C += zo * p.sCo + zi * p.sCi;
    }
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;
    const int m0 = bid.y * XBM, n0 = bid.x * XBN;
    const int nk_all = (p.K + XBK - 1) / XBK;
    const int kt0 = p.split_k > 1 ? bid.z * p.k_tiles_per_split : 0;
    const int nk = p.split_k > 1 ? min(nk_all - kt0, p.k_tiles_per_split) : nk_all;
    if (nk <= 0) return;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // rows of the padding (gemm_bf16.hip has the same two filters): a wave's 32-row tiles whose group is flagged 0 are not
    // multiplied (k-minor A: six MFMAs per product saved; the rows are still loaded - they hold zeros), and in the dW layout the
    // k-tiles of flagged-0 contraction groups are neither loaded nor multiplied
    unsigned pm = 0xFu;
    if constexpr (!A_KMAJOR) {
        if (p.mflags != nullptr) {
            const unsigned w4 = *reinterpret_cast<const unsigned*>(p.mflags + (m0 >> 5));
            pm = ((w4 & 0xffu) ? 1u : 0u) | ((w4 & 0xff00u) ? 2u : 0u) | ((w4 & 0xff0000u) ? 4u : 0u) | ((w4 & 0xff000000u) ? 8u : 0u);
            pm = __builtin_amdgcn_readfirstlane(pm);
        }
    }
    const bool rt_ok[2] = {((pm >> (wm / 32)) & 1u) != 0, ((pm >> (wm / 32 + 1)) & 1u) != 0};
    const int g0 = kt0 >> 1;                               // first 32-row contraction group of this workgroup's k-range (XBK = 16)
    unsigned long long km0 = ~0ull, km1 = ~0ull;
    if (p.kflags != nullptr && ((kt0 + nk - 1) >> 1) - g0 < 128) {
        const int ng = ((kt0 + nk - 1) >> 1) - g0 + 1;
        km0 = __ballot(lane < ng ? p.kflags[g0 + lane] != 0 : false);
        km1 = __ballot(64 + lane < ng ? p.kflags[g0 + 64 + lane] != 0 : false);
    }
    auto tile_ok = [&](int kt) {
        const int g = ((kt0 + kt) >> 1) - g0;
        return g < 64 ? ((km0 >> g) & 1ull) != 0 : g < 128 ? ((km1 >> (g - 64)) & 1ull) != 0 : true;
    };
    // operand tiles are prefetched TWO k-tiles ahead in a ring of two register sets (the loop is unrolled by 2 so the
    // ring index is static): at ~2 us of loaded HBM/L2 latency a one-tile-ahead pipeline paid a round trip per k-tile
    float ra[2][XNREG], rb[2][XNREG];
    const bool fa = A_KMAJOR || p.a_vec, fb = B_KMAJOR || p.b_vec;
    auto load_tiles = [&](int k0, auto set_c) {
        constexpr int S = decltype(set_c)::value;
        if (k0 + XBK <= p.K && fa && fb) {
            TileX<A_KMAJOR>::load_fast(p.A, p.lda, m0, p.M, k0, tid, ra[S]);
            TileX<B_KMAJOR>::load_fast(p.B, p.ldb, n0, p.N, k0, tid, rb[S]);
        } else {
            TileX<A_KMAJOR>::load(p.A, p.lda, m0, p.M, k0, p.K, tid, p.a_vec, ra[S]);
            TileX<B_KMAJOR>::load(p.B, p.ldb, n0, p.N, k0, p.K, tid, p.b_vec, rb[S]);
        }
    };
    using S0 = std::integral_constant<int, 0>;
    using S1 = std::integral_constant<int, 1>;
    auto iter = [&](int kt, auto set_c) {           // tile kt is in LDS stage kt & 1; its register set (S) is free
        constexpr int S = decltype(set_c)::value;
        const int cur = kt & 1;
        if (kt + 2 < nk && tile_ok(kt + 2)) load_tiles((kt0 + kt + 2) * XBK, set_c);
        const bool this_ok = tile_ok(kt);
        const unsigned short* As = lds[cur][0] + (wm + (lane & 31)) * XPITCH + (lane >> 5) * 8;
        const unsigned short* Bs = lds[cur][1] + (wn + (lane & 31)) * XPITCH + (lane >> 5) * 8;
        bf16x8 a[2][3], b[2][3];
        if (this_ok) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int pl = 0; pl < NPL; ++pl) {
                    if (rt_ok[i]) a[i][pl] = *reinterpret_cast<const bf16x8*>(As + pl * XPLANE + 32 * i * XPITCH);
                    b[i][pl] = *reinterpret_cast<const bf16x8*>(Bs + pl * XPLANE + 32 * i * XPITCH);
                }
        }
        // smallest terms first
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                if (!this_ok || !rt_ok[i]) continue;
                f32x16 c = acc[i][j];
                if constexpr (NPL == 3) {
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][2], b[j][0], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][2], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][1], b[j][1], c, 0, 0, 0);
                }
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][1], b[j][0], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][1], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][0], c, 0, 0, 0);
                acc[i][j] = c;
            }
        if (kt + 1 < nk && tile_ok(kt + 1)) {
            TileX<A_KMAJOR>::template store<NPL>(lds[cur ^ 1][0], tid, ra[1 - S]);
            TileX<B_KMAJOR>::template store<NPL>(lds[cur ^ 1][1], tid, rb[1 - S]);
        }
        __syncthreads();
    };
    if (tile_ok(0)) load_tiles(kt0 * XBK, S0{});
    if (nk > 1 && tile_ok(1)) load_tiles((kt0 + 1) * XBK, S1{});
    if (tile_ok(0)) {
        TileX<A_KMAJOR>::template store<NPL>(lds[0][0], tid, ra[0]);
        TileX<B_KMAJOR>::template store<NPL>(lds[0][1], tid, rb[0]);
    }
    __syncthreads();
    for (int kt = 0; kt < nk; kt += 2) {
        iter(kt, S0{});
        if (kt + 1 < nk) iter(kt + 1, S1{});
    }
    // (the k-loop ended with a barrier: every wave is done with the operand tiles, the LDS is free for staging)
    if constexpr (EPI == 1) gemm_pool_tile(acc, p, *pa, bid.z, n0, wm, wn, lane, wave, reinterpret_cast<float*>(&lds[0][0][0]));
    else if constexpr (EPI == 2) gemm_lngrad_tile(acc, p, *pa, m0, n0, wm, wn, lane, wave, reinterpret_cast<float*>(&lds[0][0][0]));
    else gemm_store_tile(acc, p, m0, n0, wm, wn, lane, reinterpret_cast<float*>(&lds[0][0][0]) + wave * (32 * 72));
}

template <bool A_KMAJOR, bool B_KMAJOR>
__global__ __launch_bounds__(256) void gemm_f32x3_kernel(GemmXArgs p) {
    gemm_f32x3_body<A_KMAJOR, B_KMAJOR, 0, PoolArgs>(p, nullptr);
}
// training simpool: one video per blockIdx.z, max-pool epilogue (common.hpp, gemm_pool_tile)
__global__ __launch_bounds__(256) void gemm_f32x3_pool_kernel(GemmXArgs p, PoolArgs pa) {
    gemm_f32x3_body<false, false, 1, PoolArgs>(p, &pa);
}
// the two-plane ("fp32x2") forms: the forward layout and the pooled simpool product
__global__ __launch_bounds__(256) void gemm_f32x2_kernel(GemmXArgs p) {
    gemm_f32x3_body<false, false, 0, PoolArgs, 2>(p, nullptr);
}
__global__ __launch_bounds__(256) void gemm_f32x2_pool_kernel(GemmXArgs p, PoolArgs pa) {
    gemm_f32x3_body<false, false, 1, PoolArgs, 2>(p, &pa);
}
// dz' = dY W (the dX layout) with the LayerNorm-parameter-gradient epilogue: the parity-mode twin of gemm_bf16_lngrad_kernel
__global__ __launch_bounds__(256) void gemm_f32x3_lngrad_kernel(GemmXArgs p, LnGradArgs la) {
    gemm_f32x3_body<false, true, 2, LnGradArgs>(p, &la);
}

static int launch_gemm_x(GemmXArgs p, int batch, int a_kmajor, int b_kmajor, void* stream) {
    const dim3 grid((p.N + XBN - 1) / XBN, (p.M + XBM - 1) / XBM, batch), block(256);
    hipStream_t s = (hipStream_t)stream;
    if (!a_kmajor && !b_kmajor) DLDKD_LAUNCH((gemm_f32x3_kernel<false, false>), grid, block, 0, s, p);
    else if (!a_kmajor && b_kmajor) DLDKD_LAUNCH((gemm_f32x3_kernel<false, true>), grid, block, 0, s, p);
    else if (a_kmajor && b_kmajor) DLDKD_LAUNCH((gemm_f32x3_kernel<true, true>), grid, block, 0, s, p);
    else DLDKD_LAUNCH((gemm_f32x3_kernel<true, false>), grid, block, 0, s, p);
    return check_launch("gemm_f32x3");
}

int launch_linear_lngrad_x3(const float* dy, const float* W, long M, int N, int K, const LnGradArgs& la, void* stream,
                            const unsigned char* row_flags) {
    const int a_vec = !(N & 3) && !((uintptr_t)dy & 15);
    GemmXArgs p{dy, W, nullptr, nullptr, (int)M, K, N, N, K, K, 0, a_vec, 0, 1, 0, 0, 0, 0, 0, 0, 1.0f, 1, 0};
    p.mflags = (M % XBM == 0 && !((uintptr_t)row_flags & 3)) ? row_flags : nullptr;
    DLDKD_LAUNCH(gemm_f32x3_lngrad_kernel, dim3((K + XBN - 1) / XBN, (unsigned)((M + XBM - 1) / XBM), 1), dim3(256), 0,
                 (hipStream_t)stream, p, la);
    return check_launch("linear_lngrad (fp32x3)");
}

int launch_simpool_pool_x3(const float* g, const float* q, int nv, int L, int nq, int D, const PoolArgs& pa, void* stream, int planes) {
    const bool al = !(D & 3) && !((uintptr_t)g & 15), bl = !(D & 3) && !((uintptr_t)q & 15);
    GemmXArgs p{g, q, nullptr, nullptr, L, nq, D, D, D, nq, 0, al, bl, 1, (long)L * D, 0, 0, 0, 0, 0, 1.0f, 1, 0};
    if (planes == 2) DLDKD_LAUNCH(gemm_f32x2_pool_kernel, dim3((nq + XBN - 1) / XBN, 1, nv), dim3(256), 0, (hipStream_t)stream, p, pa);
    else DLDKD_LAUNCH(gemm_f32x3_pool_kernel, dim3((nq + XBN - 1) / XBN, 1, nv), dim3(256), 0, (hipStream_t)stream, p, pa);
    return check_launch("simpool_train_fwd (fp32x3 / fp32x2)");
}

}  // namespace dldkd

using namespace dldkd;

int dldkd::gemm_f32x3_split_plan(int M, int N, int K, int a_kmajor, int b_kmajor, int* k_tiles_per_split) {
    const int tiles = ((N + XBN - 1) / XBN) * ((M + XBM - 1) / XBM);
    const int nk = (K + XBK - 1) / XBK;
    *k_tiles_per_split = nk;
    // never for the forward layout: the forward pass - hence the losses - stays one k-ordered accumulation per element
    if (!(a_kmajor || b_kmajor) || (((long)M * N) & 3) || tiles >= 128 || nk < 32) return 1;
    int split = (256 + tiles - 1) / tiles;
    if (split > nk / 8) split = nk / 8;
    if (split <= 1) return 1;
    *k_tiles_per_split = (nk + split - 1) / split;
    return (nk + *k_tiles_per_split - 1) / *k_tiles_per_split;
}

static int gemm_f32x3_impl(const float* A, const float* B, const float* bias, float* C, int M, int N, int K, int lda,
                           int ldb, int ldc, int a_kmajor, int b_kmajor, int relu, void* workspace, size_t workspace_bytes,
                           const unsigned char* flags, void* stream);

extern "C" int dldkd_gemm_f32x3(const float* A, const float* B, const float* bias, float* C, int M, int N, int K, int lda,
                                int ldb, int ldc, int a_kmajor, int b_kmajor, int relu, void* workspace, size_t workspace_bytes,
                                void* stream) {
    return gemm_f32x3_impl(A, B, bias, C, M, N, K, lda, ldb, ldc, a_kmajor, b_kmajor, relu, workspace, workspace_bytes, nullptr, stream);
}

extern "C" int dldkd_gemm_f32x3_flags(const float* A, const float* B, const float* bias, float* C, int M, int N, int K, int lda,
                                      int ldb, int ldc, int a_kmajor, int b_kmajor, int relu, void* workspace, size_t workspace_bytes,
                                      const unsigned char* flags, void* stream) {
    return gemm_f32x3_impl(A, B, bias, C, M, N, K, lda, ldb, ldc, a_kmajor, b_kmajor, relu, workspace, workspace_bytes, flags, stream);
}

extern "C" int dldkd_gemm_f32x2(const float* A, const float* B, const float* bias, float* C, int M, int N, int K, int lda, int ldb, int ldc,
                                int relu, const unsigned char* row_flags, void* stream) {
    if (M < 0 || N < 0 || K < 0 || lda < 1 || ldb < 1 || ldc < N) { set_error("gemm_f32x2: bad sizes"); return DLDKD_EINVAL; }
    if (M == 0 || N == 0) return DLDKD_OK;
    if (!A || !B || !C) { set_error("gemm_f32x2: null pointer"); return DLDKD_EINVAL; }
    const int a_vec = !(lda & 3) && !((uintptr_t)A & 15), b_vec = !(ldb & 3) && !((uintptr_t)B & 15);
    GemmXArgs p{A, B, bias, C, M, N, K, lda, ldb, ldc, relu, a_vec, b_vec, 1, 0, 0, 0, 0, 0, 0, 1.0f, 1, 0};
    if (row_flags != nullptr && M % XBM == 0 && !((uintptr_t)row_flags & 3)) p.mflags = row_flags;
    DLDKD_LAUNCH(gemm_f32x2_kernel, dim3((N + XBN - 1) / XBN, (M + XBM - 1) / XBM, 1), dim3(256), 0, (hipStream_t)stream, p);
    return check_launch("gemm_f32x2");
}

static int gemm_f32x3_impl(const float* A, const float* B, const float* bias, float* C, int M, int N, int K, int lda,
                           int ldb, int ldc, int a_kmajor, int b_kmajor, int relu, void* workspace, size_t workspace_bytes,
                           const unsigned char* flags, void* stream) {
    if (M < 0 || N < 0 || K < 0 || lda < 1 || ldb < 1 || ldc < N) { set_error("gemm_f32x3: bad sizes"); return DLDKD_EINVAL; }
    if (M == 0 || N == 0) return DLDKD_OK;
    if (!A || !B || !C) { set_error("gemm_f32x3: null pointer"); return DLDKD_EINVAL; }
    const int a_vec = !(lda & 3) && !((uintptr_t)A & 15), b_vec = !(ldb & 3) && !((uintptr_t)B & 15);
    GemmXArgs p{A, B, bias, C, M, N, K, lda, ldb, ldc, relu, a_vec, b_vec, 1, 0, 0, 0, 0, 0, 0, 1.0f, 1, 0};
    // flags: one byte per 32 rows of the activation operand - A's rows when A is k-minor (forward, dX; M % 128 == 0), the
    // contraction index when both operands are k-major (dW)
    if (flags != nullptr && !a_kmajor && M % XBM == 0 && !((uintptr_t)flags & 3)) p.mflags = flags;
    if (flags != nullptr && a_kmajor && b_kmajor) p.kflags = flags;
    int per = 0;
    const int split = (!bias && !relu && ldc == N && !((uintptr_t)C & 15)) ? gemm_f32x3_split_plan(M, N, K, a_kmajor, b_kmajor, &per) : 1;
    if (split > 1 && workspace && !((uintptr_t)workspace & 15) && workspace_bytes >= (size_t)split * M * N * sizeof(float)) {
        p.k_tiles_per_split = per;
        p.split_k = split;
        p.C = (float*)workspace;
        const int rc = launch_gemm_x(p, p.split_k, a_kmajor, b_kmajor, stream);
        if (rc != DLDKD_OK) return rc;
        return launch_splitk_reduce((const float*)workspace, C, p.split_k, (long)M * N, (hipStream_t)stream);
    }
    return launch_gemm_x(p, 1, a_kmajor, b_kmajor, stream);
}
