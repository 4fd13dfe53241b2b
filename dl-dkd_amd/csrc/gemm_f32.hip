// gemm_f32: C[M,N] = act( sum_k A(m,k) * B(n,k) + bias[n] ) on the fp32-input MFMA
// (v_mfma_f32_32x32x2_f32: exact fp32 fmaf chains, 157 TFLOP/s peak on MI355X).
//
// This is the parity-grade building block of the encoder towers: every nn.Linear of the reference path
// (LinearLayer.net.1 model_components.py:302, BertSelfAttention.query/key/value :388-390,
// BertSelfOutput.dense :442, out_mapping_linear model.py:39) and, through the operand-layout flags, both
// of its gradients without explicit transposes:
//     forward  Y  = X . W^T      A = X  [M][K]   (a_kmajor 0)   B = W  [N][K]   (b_kmajor 0)
//     dX       = dY . W          A = dY [M][N']  (a_kmajor 0)   B = W  [N'][K'] (b_kmajor 1)
//     dW       = dY^T . X        A = dY [M'][N'] (a_kmajor 1)   B = X  [M'][K'] (b_kmajor 1)
// "kmajor" = the contraction index is the slow (row) index of that operand in memory.
//
// Tiling: 128x128 block, BK = 16, 4 waves (2x2), each wave 64x64 = 2x2 MFMA tiles (64 accumulator
// registers).  Operands are staged global -> registers -> LDS one k-tile ahead (double-buffered LDS, one
// barrier per k-tile).  LDS tiles are stored [row][k] with a 17-word pitch (k-minor operands) or [k][row]
// with a 132-word pitch (k-major operands) so that each MFMA fragment read (32 consecutive rows at one k)
// is bank-conflict free.
#include "common.hpp"

namespace dldkd {

constexpr int BM = 128, BN = 128, BK = 16;
constexpr int LD_KMINOR = BK + 1;     // [row][k] pitch (words)
constexpr int LD_KMAJOR = BM + 4;     // [k][row] pitch (words)
constexpr int TILE_WORDS = (BM * LD_KMINOR > BK * LD_KMAJOR) ? BM * LD_KMINOR : BK * LD_KMAJOR;

struct GemmArgs {
    const float* A;
    const float* B;
    const float* bias;
    float* C;
    int M, N, K, lda, ldb, ldc, relu;
    int a_vec, b_vec;   // operand is 16-byte aligned with a leading dimension divisible by 4 -> float4 loads
    // strided batch: blockIdx.z = zo * batch_inner + zi; operand offset = zo * s?o + zi * s?i (elements)
    int batch_inner;
    long sAo, sAi, sBo, sBi, sCo, sCi;
    float alpha;
    // split-K (unbatched launches only): blockIdx.z owns k-tiles [z * k_tiles_per_split, ...) and adds its
    // partial product into a zero-initialised C with fp32 atomics.  For dW-shaped problems (384 x 384 outputs,
    // K = 16k rows) the plain grid has 9 workgroups on a 256-CU chip.
    int split_k, k_tiles_per_split;
};

// One operand tile (128 rows x 16 k) per k-tile; each thread moves 8 floats as two float4.
template <bool KMAJOR>
struct TileIO {
    // global -> registers.  rows >= nrows or k >= K read as zero.
    static __device__ __forceinline__ void load(const float* __restrict__ P, int ld, int row0, int nrows, int k0, int K,
                                                int tid, bool vec, f32x4 (&r)[2]) {
        if constexpr (!KMAJOR) {
            // memory [row][k]: 4 threads cover one row's 16 k (64 B), 64 rows per pass
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int row = row0 + (tid >> 2) + 64 * j;
                const int k = k0 + (tid & 3) * 4;
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (row < nrows) {
                    const float* src = P + (size_t)row * ld + k;
                    if (vec && k + 3 < K) v = *reinterpret_cast<const f32x4*>(src);
                    else {
#pragma unroll
                        for (int e = 0; e < 4; ++e) if (k + e < K) v[e] = src[e];
                    }
                }
                r[j] = v;
            }
        } else {
            // memory [k][row]: 32 threads cover one k's 128 rows (512 B), 8 k per pass
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int k = k0 + (tid >> 5) + 8 * j;
                const int row = row0 + (tid & 31) * 4;
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (k < K) {
                    const float* src = P + (size_t)k * ld + row;
                    if (vec && row + 3 < nrows) v = *reinterpret_cast<const f32x4*>(src);
                    else {
#pragma unroll
                        for (int e = 0; e < 4; ++e) if (row + e < nrows) v[e] = src[e];
                    }
                }
                r[j] = v;
            }
        }
    }
    // Interior tile (all 16 k in range, 16-byte aligned): branch-free.  k-minor: rows past the end are clamped to
    // the last row (they only feed accumulator rows that are never stored); k-major: the caller guarantees that
    // all 128 rows of the tile exist.
    static __device__ __forceinline__ void load_fast(const float* __restrict__ P, int ld, int row0, int nrows, int k0,
                                                     int tid, f32x4 (&r)[2]) {
        if constexpr (!KMAJOR) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int row = min(row0 + (tid >> 2) + 64 * j, nrows - 1);
                r[j] = *reinterpret_cast<const f32x4*>(P + (size_t)row * ld + k0 + (tid & 3) * 4);
            }
        } else {
#pragma unroll
            for (int j = 0; j < 2; ++j)
                r[j] = *reinterpret_cast<const f32x4*>(P + (size_t)(k0 + (tid >> 5) + 8 * j) * ld + row0 + (tid & 31) * 4);
        }
    }
    // registers -> LDS
    static __device__ __forceinline__ void store(float* __restrict__ S, int tid, const f32x4 (&r)[2]) {
        if constexpr (!KMAJOR) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                float* dst = S + ((tid >> 2) + 64 * j) * LD_KMINOR + (tid & 3) * 4;
#pragma unroll
                for (int e = 0; e < 4; ++e) dst[e] = r[j][e];
            }
        } else {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                float* dst = S + ((tid >> 5) + 8 * j) * LD_KMAJOR + (tid & 31) * 4;
                *reinterpret_cast<f32x4*>(dst) = r[j];
            }
        }
    }
    // MFMA fragment: element (row, k) of the tile
    static __device__ __forceinline__ float frag(const float* __restrict__ S, int row, int k) {
        if constexpr (!KMAJOR) return S[row * LD_KMINOR + k];
        else return S[k * LD_KMAJOR + row];
    }
};

template <bool A_KMAJOR, bool B_KMAJOR>
__global__ __launch_bounds__(256) void gemm_f32_kernel(GemmArgs p) {
    if (p.split_k > 1) {
        p.C += (size_t)blockIdx.z * p.M * p.ldc;      // this split's partial plane in the workspace
    } else {
        const int zo = blockIdx.z / p.batch_inner, zi = blockIdx.z % p.batch_inner;
        p.A += zo * p.sAo + zi * p.sAi;
        p.B += zo * p.sBo + zi * p.sBi;
        p.C += zo * p.sCo + zi * p.sCi;
    }
    // [buffer][A|B] operand tiles; sized to also hold the epilogue staging (4 waves x 32 x 72 floats)
    constexpr int kLdsWords = 4 * TILE_WORDS > 4 * 32 * 72 ? 4 * TILE_WORDS : 4 * 32 * 72;
    __shared__ __attribute__((aligned(16))) float lds_raw[kLdsWords];
    float (*lds)[2][TILE_WORDS] = reinterpret_cast<float (*)[2][TILE_WORDS]>(lds_raw);
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    const int nk_all = (p.K + BK - 1) / BK;
    const int kt0 = p.split_k > 1 ? blockIdx.z * p.k_tiles_per_split : 0;
    const int nk = p.split_k > 1 ? min(nk_all - kt0, p.k_tiles_per_split) : nk_all;
    if (nk <= 0) return;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    f32x4 ra[2], rb[2];
    const bool fast = p.a_vec && p.b_vec && (!A_KMAJOR || m0 + BM <= p.M) && (!B_KMAJOR || n0 + BN <= p.N);
    auto load_tiles = [&](int k0) {
        if (fast && k0 + BK <= p.K) {
            TileIO<A_KMAJOR>::load_fast(p.A, p.lda, m0, p.M, k0, tid, ra);
            TileIO<B_KMAJOR>::load_fast(p.B, p.ldb, n0, p.N, k0, tid, rb);
        } else {
            TileIO<A_KMAJOR>::load(p.A, p.lda, m0, p.M, k0, p.K, tid, p.a_vec, ra);
            TileIO<B_KMAJOR>::load(p.B, p.ldb, n0, p.N, k0, p.K, tid, p.b_vec, rb);
        }
    };
    load_tiles(kt0 * BK);
    TileIO<A_KMAJOR>::store(lds[0][0], tid, ra);
    TileIO<B_KMAJOR>::store(lds[0][1], tid, rb);
    __syncthreads();

    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) load_tiles((kt0 + kt + 1) * BK);   // next tile's global loads fly under this tile's MFMAs
        const float* As = lds[cur][0];
        const float* Bs = lds[cur][1];
#pragma unroll
        for (int kk = 0; kk < BK; kk += 2) {
            const int k = kk + (lane >> 5);
            float a[2], b[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) a[i] = TileIO<A_KMAJOR>::frag(As, wm + 32 * i + (lane & 31), k);
#pragma unroll
            for (int j = 0; j < 2; ++j) b[j] = TileIO<B_KMAJOR>::frag(Bs, wn + 32 * j + (lane & 31), k);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        if (kt + 1 < nk) {
            TileIO<A_KMAJOR>::store(lds[cur ^ 1][0], tid, ra);
            TileIO<B_KMAJOR>::store(lds[cur ^ 1][1], tid, rb);
        }
        __syncthreads();
    }

    // (the k-loop ended with a barrier: every wave is done with the operand tiles, the LDS is free for staging)
    gemm_store_tile(acc, p, m0, n0, wm, wn, lane, lds_raw + wave * (32 * 72));
}

}  // namespace dldkd

using namespace dldkd;

static int launch_gemm(GemmArgs p, int batch, int a_kmajor, int b_kmajor, void* stream) {
    const dim3 grid((p.N + BN - 1) / BN, (p.M + BM - 1) / BM, batch), block(256);
    hipStream_t s = (hipStream_t)stream;
    if (!a_kmajor && !b_kmajor) DLDKD_LAUNCH((gemm_f32_kernel<false, false>), grid, block, 0, s, p);
    else if (!a_kmajor && b_kmajor) DLDKD_LAUNCH((gemm_f32_kernel<false, true>), grid, block, 0, s, p);
    else if (a_kmajor && b_kmajor) DLDKD_LAUNCH((gemm_f32_kernel<true, true>), grid, block, 0, s, p);
    else DLDKD_LAUNCH((gemm_f32_kernel<true, false>), grid, block, 0, s, p);
    return check_launch("gemm_f32");
}

// Split-K plan (weight gradients, dq of the clip scores: the output grid cannot fill the chip and K is long).  Never for
// the forward layout: the forward pass - hence the losses - stays one k-ordered accumulation per element.
int dldkd::gemm_f32_split_plan(int M, int N, int K, int a_kmajor, int b_kmajor, int* k_tiles_per_split) {
    const int tiles = ((N + BN - 1) / BN) * ((M + BM - 1) / BM);
    const int nk = (K + BK - 1) / BK;
    *k_tiles_per_split = nk;
    if (!(a_kmajor || b_kmajor) || (((long)M * N) & 3) || tiles >= 128 || nk < 32) return 1;
    int split = (512 + tiles - 1) / tiles;
    if (split > nk / 8) split = nk / 8;
    if (split <= 1) return 1;
    *k_tiles_per_split = (nk + split - 1) / split;
    return (nk + *k_tiles_per_split - 1) / *k_tiles_per_split;
}

extern "C" int dldkd_gemm_f32(const float* A, const float* B, const float* bias, float* C, int M, int N, int K, int lda,
                              int ldb, int ldc, int a_kmajor, int b_kmajor, int relu, void* workspace, size_t workspace_bytes,
                              void* stream) {
    if (M < 0 || N < 0 || K < 0 || lda < 1 || ldb < 1 || ldc < N) {
        set_error("gemm_f32: bad sizes M=%d N=%d K=%d lda=%d ldb=%d ldc=%d", M, N, K, lda, ldb, ldc);
        return DLDKD_EINVAL;
    }
    if (M == 0 || N == 0) return DLDKD_OK;
    if (!A || !B || !C) { set_error("gemm_f32: null pointer"); return DLDKD_EINVAL; }
    const int a_vec = !(lda & 3) && !((uintptr_t)A & 15), b_vec = !(ldb & 3) && !((uintptr_t)B & 15);
    GemmArgs p{A, B, bias, C, M, N, K, lda, ldb, ldc, relu, a_vec, b_vec, 1, 0, 0, 0, 0, 0, 0, 1.0f, 1, 0};
    int per = 0;
    const int split = (!bias && !relu && ldc == N && !((uintptr_t)C & 15)) ? gemm_f32_split_plan(M, N, K, a_kmajor, b_kmajor, &per) : 1;
    if (split > 1 && workspace && !((uintptr_t)workspace & 15) && workspace_bytes >= (size_t)split * M * N * sizeof(float)) {
        p.k_tiles_per_split = per;
        p.split_k = split;
        p.C = (float*)workspace;
        const int rc = launch_gemm(p, p.split_k, a_kmajor, b_kmajor, stream);
        if (rc != DLDKD_OK) return rc;
        return launch_splitk_reduce((const float*)workspace, C, p.split_k, (long)M * N, (hipStream_t)stream);
    }
    return launch_gemm(p, 1, a_kmajor, b_kmajor, stream);
}
