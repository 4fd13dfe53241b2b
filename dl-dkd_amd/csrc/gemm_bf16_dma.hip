// gemm_bf16_nt: C[M, N] = act(A[M, K] . B[N, K]^T + bias), fp32 operands in memory (both k-minor: the Linear forward layout, and
// dX once the weight has been transposed), bf16 MFMA, fp32 accumulation - the throughput-mode GEMM of the towers' 384- and
// 1152-wide layers (reference model_components.py:302,388-390,442) where gemm_bf16_kernel is latency-bound: at 16,384 x 384 x
// 384 a workgroup of that kernel makes 12 dependent round trips (global load -> registers -> convert -> LDS -> barrier, one
// k-tile of prefetch) for 2 us of MFMA work, 23 us in all against ~10 us for moving the 50 MB once.
// Here the operand tiles go HBM / L2 -> LDS by LDS-DMA as fp32 (no staging registers, no conversion on the way in), a whole
// ring of NST k-tiles per operand is in flight from the first instruction, and the fragments are read from LDS as fp32 and
// rounded to bf16 (v_cvt_pk_bf16_f32, RNE - the rounding gemm_bf16_kernel applies on its way INTO the LDS: same products, same
// k order per element, so the results are bit-identical to gemm_bf16_kernel's).
//   * 128 x 128 tile, 4 waves of 64 x 64 (2 x 2 mfma_f32_32x32x16_bf16 tiles), k-tiles of 32: 16 KiB per operand and stage.
//   * LDS image of a k-tile = K4's (in_proj_rows128.hip): 128-byte rows, 16-byte chunk c of row r at chunk c ^ ((r >> 1) & 7),
//     produced by permuting the per-lane SOURCE address of the DMA; fragment reads (lane = row, 2 x ds_read_b128) are
//     conflict-free.
//   * NST = 2 stages, 64 KiB per workgroup, two workgroups per CU (a 16,384 x 384 GEMM is 384 workgroups: one round).
//     The DMA of tile t + 1 is issued before the fragment reads of tile t; its wait (asm vmcnt(0): the compiler does not see
//     the DMA and so does not drain the queue before unrelated LDS reads) sits in front of the tile's closing barrier.
//   * epilogue: gemm_store_tile (bias, ReLU, float4 rows through the wave's LDS), XCD-aware tile order.
#include "common.hpp"

namespace dldkd {
namespace gdma {

constexpr int BM = 128, BN = 128, BK = 32, NST = 2;
constexpr int TILE_B = BM * BK * 4;                 // 16 KiB: one operand's k-tile as fp32
constexpr int STAGE_B = 2 * TILE_B;                 // A then B

struct Args {
    const float* A;
    const float* B;
    const float* bias;
    float* C;
    int M, N, K, lda, ldb, ldc, relu;
    float alpha;
    const unsigned char* mflags;      // per 32 rows of A / C (M % 128 == 0), or null: 0 = rows of the padding (zero operand rows or
                                      // don't-care outputs) - not loaded, not multiplied; their C rows come out as act(bias)
};

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void glds16_s(uint32_t voff, const char* sbase, uint32_t lds_base) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(voff), "s"(sbase), "s"(lds_base) : "memory");
}
__device__ __forceinline__ bf16x8 cvt8(const f32x4& lo, const f32x4& hi) {
    u32x4 u;
    asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(u[0]) : "v"(lo[0]), "v"(lo[1]));
    asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(u[1]) : "v"(lo[2]), "v"(lo[3]));
    asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(u[2]) : "v"(hi[0]), "v"(hi[1]));
    asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(u[3]) : "v"(hi[2]), "v"(hi[3]));
    return __builtin_bit_cast(bf16x8, u);
}

__global__ __launch_bounds__(256, 2) void gemm_bf16_nt_kernel(const Args p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const Tile3 bid = xcd_tile_order();
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;
    const int m0 = bid.y * BM, n0 = bid.x * BN;
    const int nk = p.K / BK;
    const uint32_t smem_lds = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(smem));

    // DMA: piece t = 4 wave + q of a tile = rows 8 t .. 8 t + 7 (1 KiB); lane -> LDS chunk 64 t + lane = row 8 t + (lane >> 3),
    // position lane & 7, which holds global chunk (lane & 7) ^ ((row >> 1) & 7) of that row.  Rows past the end are clamped
    // (they feed accumulator rows / columns that are never stored).
    uint32_t voa[4], vob[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int row = 8 * (4 * wave + q) + (lane >> 3);
        const int ch = ((lane & 7) ^ ((row >> 1) & 7)) << 4;
        const int ra = min(m0 + row, p.M - 1) - m0, rb = min(n0 + row, p.N - 1) - n0;     // may be negative only if the tile is empty (never launched)
        voa[q] = (uint32_t)(ra * p.lda * 4 + ch);
        vob[q] = (uint32_t)(rb * p.ldb * 4 + ch);
    }
    const char* abase = reinterpret_cast<const char*>(p.A + (size_t)m0 * p.lda);
    const char* bbase = reinterpret_cast<const char*>(p.B + (size_t)n0 * p.ldb);
    // row groups: wave w's four A pieces ARE group w of the tile (rows 32 w .. 32 w + 31)
    unsigned pm = 0xFu;
    if (p.mflags != nullptr) {
        const unsigned w4 = *reinterpret_cast<const unsigned*>(p.mflags + (m0 >> 5));
        pm = ((w4 & 0xffu) ? 1u : 0u) | ((w4 & 0xff00u) ? 2u : 0u) | ((w4 & 0xff0000u) ? 4u : 0u) | ((w4 & 0xff000000u) ? 8u : 0u);
        pm = __builtin_amdgcn_readfirstlane(pm);
    }
    const bool load_a = ((pm >> wave) & 1u) != 0;
    const bool rt_ok[2] = {((pm >> (wm / 32)) & 1u) != 0, ((pm >> (wm / 32 + 1)) & 1u) != 0};
    auto issue = [&](int kt, int stage) {
        const char* as = abase + (size_t)kt * (BK * 4);
        const char* bs = bbase + (size_t)kt * (BK * 4);
        const uint32_t dst = smem_lds + stage * STAGE_B + (4 * wave) * 1024;
        if (load_a) {
#pragma unroll
            for (int q = 0; q < 4; ++q) glds16_s(voa[q], as, dst + q * 1024);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) glds16_s(vob[q], bs, dst + TILE_B + q * 1024);
    };

    // fragment reads: lane (r = lane & 31, h = lane >> 5) takes chunks 4 kk + 2 h + e (e = 0, 1) of row `base + r`
    int fo[2][2];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int r = lane & 31, c = 4 * kk + 2 * (lane >> 5) + e;
            fo[kk][e] = r * 128 + ((c ^ ((r >> 1) & 7)) << 4);        // (rows 32 i + r: (r >> 1) & 7 is unchanged by + 32 i)
        }

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    issue(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int st = kt & 1;
        if (kt + 1 < nk) issue(kt + 1, st ^ 1);
        const char* As = smem + st * STAGE_B + wm * 128;
        const char* Bs = smem + st * STAGE_B + TILE_B + wn * 128;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            bf16x8 a[2], b[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const f32x4 lo = *reinterpret_cast<const f32x4*>(Bs + j * 4096 + fo[kk][0]);
                const f32x4 hi = *reinterpret_cast<const f32x4*>(Bs + j * 4096 + fo[kk][1]);
                b[j] = cvt8(lo, hi);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                if (!rt_ok[i]) continue;                          // rows of the padding: never loaded
                const f32x4 lo = *reinterpret_cast<const f32x4*>(As + i * 4096 + fo[kk][0]);
                const f32x4 hi = *reinterpret_cast<const f32x4*>(As + i * 4096 + fo[kk][1]);
                a[i] = cvt8(lo, hi);
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the next tile has landed (this wave's pieces; the barrier covers the others)
        __syncthreads();                                       // and everyone is done reading this one
    }
    gemm_store_tile(acc, p, m0, n0, wm, wn, lane, reinterpret_cast<float*>(smem) + wave * (32 * 72));
}

// ---------------------------------------------------------------------------------------------------------------------------------
// The same kernel with BOTH operands bf16 in memory (the training input projection: A = the LayerNorm-dropout rows that
// dldkd_layernorm_dropout_bf16 writes, B = the weight cast once per step): a k-tile is 64 elements = the same 128-byte rows, the
// same DMA and the same XOR image, but a 16-byte chunk now IS one lane's fragment (8 bf16) - no conversion, one ds_read_b128 per
// fragment - and a tile carries four k-steps = 16 MFMAs per wave per barrier instead of 8, half as many dependent round trips
// for a given K (16,384 x 384 x 3,072: 48 instead of 96 - this GEMM was 150 us in the register-staged kernel for 40 us of MFMA work).
// ---------------------------------------------------------------------------------------------------------------------------------
constexpr int BK16 = 64;

struct Args16 {
    const unsigned short* A;
    const unsigned short* B;
    const float* bias;
    float* C;
    int M, N, K, lda, ldb, ldc, relu;
    float alpha;
    const unsigned char* mflags;
    // plane segments (round 6, the forward GEMMs of the "mixed" training precision): the contraction runs nseg times over K, segment s
    // reading A from byte offset aoff[s] and B from boff[s] - with the operands stored as bf16 PLANES (x = h + m: dldkd_split2_bf16,
    // dldkd_layernorm_ex_f32) the three segments (m, h), (h, m), (h, h) are the two-plane product of gemm_f32x3.hip (NPL = 2) at THIS
    // kernel's rate: no split on the way to LDS, tiles by LDS-DMA.  nseg = 0 / 1: one pass, offsets ignored.  nseg = 3: am / bm = the
    // byte offsets of the m planes (segment 0 reads A's m plane, segment 1 B's, segment 2 both h planes).
    int nseg;
    long am, bm;
};

__global__ __launch_bounds__(256, 2) void gemm_bf16_nt16_kernel(const Args16 p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const Tile3 bid = xcd_tile_order();
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;
    const int m0 = bid.y * BM, n0 = bid.x * BN;
    const int nk1 = p.K / BK16, nseg = p.nseg > 1 ? p.nseg : 1;
    const int nk = nk1 * nseg;
    const uint32_t smem_lds = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(smem));
    uint32_t voa[4], vob[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int row = 8 * (4 * wave + q) + (lane >> 3);
        const int ch = ((lane & 7) ^ ((row >> 1) & 7)) << 4;
        const int ra = min(m0 + row, p.M - 1) - m0, rb = min(n0 + row, p.N - 1) - n0;
        voa[q] = (uint32_t)(ra * p.lda * 2 + ch);
        vob[q] = (uint32_t)(rb * p.ldb * 2 + ch);
    }
    const char* abase = reinterpret_cast<const char*>(p.A + (size_t)m0 * p.lda);
    const char* bbase = reinterpret_cast<const char*>(p.B + (size_t)n0 * p.ldb);
    unsigned pm = 0xFu;
    if (p.mflags != nullptr) {
        const unsigned w4 = *reinterpret_cast<const unsigned*>(p.mflags + (m0 >> 5));
        pm = ((w4 & 0xffu) ? 1u : 0u) | ((w4 & 0xff00u) ? 2u : 0u) | ((w4 & 0xff0000u) ? 4u : 0u) | ((w4 & 0xff000000u) ? 8u : 0u);
        pm = __builtin_amdgcn_readfirstlane(pm);
    }
    const bool load_a = ((pm >> wave) & 1u) != 0;
    const bool rt_ok[2] = {((pm >> (wm / 32)) & 1u) != 0, ((pm >> (wm / 32 + 1)) & 1u) != 0};
    int iseg = 0, ikk = 0;                                 // (segment, k-tile inside it) of the NEXT tile to be issued (tiles are issued in order)
    auto issue = [&](int, int stage) {
        const char* as = abase + ((p.nseg > 1 && iseg == 0) ? p.am : 0) + (size_t)ikk * (BK16 * 2);
        const char* bs = bbase + ((p.nseg > 1 && iseg == 1) ? p.bm : 0) + (size_t)ikk * (BK16 * 2);
        if (++ikk == nk1) { ikk = 0; ++iseg; }
        const uint32_t dst = smem_lds + stage * STAGE_B + (4 * wave) * 1024;
        if (load_a) {
#pragma unroll
            for (int q = 0; q < 4; ++q) glds16_s(voa[q], as, dst + q * 1024);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) glds16_s(vob[q], bs, dst + TILE_B + q * 1024);
    };
    // fragment of k-step kk (16 k = chunks 2 kk, 2 kk + 1): lane (r = lane & 31, h = lane >> 5) reads chunk 2 kk + h of its row
    int fo[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
        const int r = lane & 31, c = 2 * kk + (lane >> 5);
        fo[kk] = r * 128 + ((c ^ ((r >> 1) & 7)) << 4);
    }
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    issue(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int st = kt & 1;
        if (kt + 1 < nk) issue(kt + 1, st ^ 1);
        const char* As = smem + st * STAGE_B + wm * 128;
        const char* Bs = smem + st * STAGE_B + TILE_B + wn * 128;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            bf16x8 a[2], b[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) b[j] = *reinterpret_cast<const bf16x8*>(Bs + j * 4096 + fo[kk]);
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                if (!rt_ok[i]) continue;
                a[i] = *reinterpret_cast<const bf16x8*>(As + i * 4096 + fo[kk]);
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    gemm_store_tile(acc, p, m0, n0, wm, wn, lane, reinterpret_cast<float*>(smem) + wave * (32 * 72));
}

// Training simpool with bf16 operands (the queries and the gallery clips cast once by dldkd_row_invnorm2_cast_f32, the pass that
// reads every row for its norm anyway): S_v = G_v Q^T, one video per blockIdx.z on the tile rows (L <= 128 clips), 128 queries on the
// columns, the key-clip max-pool epilogue of gemm_pool_tile instead of a store.  Same DMA tiles and fragments as the kernel above;
// row tiles past the video's LENGTH are neither loaded nor multiplied (their clips are masked in the epilogue).  Replaces the
// register-staged fp32-operand kernel (gemm_bf16_pool_kernel: 93 us for 8 GFLOP at the TVR batch, converting both operands on
// their way to LDS 32 k at a time).
__global__ __launch_bounds__(256, 2) void gemm_bf16_nt16_pool_kernel(const Args16 p, const PoolArgs pa) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;
    const int v = blockIdx.z, n0 = blockIdx.x * BN;
    const int nk1 = p.K / BK16, nseg = p.nseg > 1 ? p.nseg : 1;
    const int nk = nk1 * nseg;
    const int len = min(pa.lens[v], p.M);
    const uint32_t smem_lds = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(smem));
    uint32_t voa[4], vob[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int row = 8 * (4 * wave + q) + (lane >> 3);
        const int ch = ((lane & 7) ^ ((row >> 1) & 7)) << 4;
        const int ra = min(row, p.M - 1), rb = min(n0 + row, p.N - 1) - n0;
        voa[q] = (uint32_t)(ra * p.lda * 2 + ch);
        vob[q] = (uint32_t)(rb * p.ldb * 2 + ch);
    }
    const char* abase = reinterpret_cast<const char*>(p.A + (size_t)v * p.M * p.lda);
    const char* bbase = reinterpret_cast<const char*>(p.B + (size_t)n0 * p.ldb);
    const int tiles = (len + 31) >> 5;
    const unsigned pm = __builtin_amdgcn_readfirstlane(tiles >= 4 ? 0xFu : ((1u << tiles) - 1u));
    const bool load_a = ((pm >> wave) & 1u) != 0;
    const bool rt_ok[2] = {((pm >> (wm / 32)) & 1u) != 0, ((pm >> (wm / 32 + 1)) & 1u) != 0};
    int iseg = 0, ikk = 0;                                 // as in gemm_bf16_nt16_kernel
    auto issue = [&](int, int stage) {
        const char* as = abase + ((p.nseg > 1 && iseg == 0) ? p.am : 0) + (size_t)ikk * (BK16 * 2);
        const char* bs = bbase + ((p.nseg > 1 && iseg == 1) ? p.bm : 0) + (size_t)ikk * (BK16 * 2);
        if (++ikk == nk1) { ikk = 0; ++iseg; }
        const uint32_t dst = smem_lds + stage * STAGE_B + (4 * wave) * 1024;
        if (load_a) {
#pragma unroll
            for (int q = 0; q < 4; ++q) glds16_s(voa[q], as, dst + q * 1024);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) glds16_s(vob[q], bs, dst + TILE_B + q * 1024);
    };
    int fo[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
        const int r = lane & 31, c = 2 * kk + (lane >> 5);
        fo[kk] = r * 128 + ((c ^ ((r >> 1) & 7)) << 4);
    }
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    if (pm != 0u) {                                   // (a video without clips: nothing to multiply, the epilogue writes the constants)
        issue(0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        for (int kt = 0; kt < nk; ++kt) {
            const int st = kt & 1;
            if (kt + 1 < nk) issue(kt + 1, st ^ 1);
            const char* As = smem + st * STAGE_B + wm * 128;
            const char* Bs = smem + st * STAGE_B + TILE_B + wn * 128;
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                bf16x8 a[2], b[2];
#pragma unroll
                for (int j = 0; j < 2; ++j) b[j] = *reinterpret_cast<const bf16x8*>(Bs + j * 4096 + fo[kk]);
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    if (!rt_ok[i]) continue;
                    a[i] = *reinterpret_cast<const bf16x8*>(As + i * 4096 + fo[kk]);
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
                }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
    }
    gemm_pool_tile(acc, p, pa, v, n0, wm, wn, lane, wave, reinterpret_cast<float*>(smem));
}

// x (n) fp32 -> bf16 (round to nearest even): the weight operand of gemm_bf16_nt16, once per optimizer step
__global__ __launch_bounds__(256) void cast_bf16_kernel(const float* __restrict__ x, unsigned short* __restrict__ y, long n4) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    const f32x4 v = reinterpret_cast<const f32x4*>(x)[i];
    uint2 pk;
    pk.x = (unsigned)f32_to_bf16_bits(v[0]) | ((unsigned)f32_to_bf16_bits(v[1]) << 16);
    pk.y = (unsigned)f32_to_bf16_bits(v[2]) | ((unsigned)f32_to_bf16_bits(v[3]) << 16);
    reinterpret_cast<uint2*>(y)[i] = pk;
}

}  // namespace gdma
}  // namespace dldkd

using namespace dldkd;

extern "C" int dldkd_cast_bf16(const float* x, void* y, long n, void* stream) {
    if (n < 0 || (n & 3)) { set_error("cast_bf16: n must be a multiple of 4"); return DLDKD_EINVAL; }
    if (n == 0) return DLDKD_OK;
    if (!x || !y || ((uintptr_t)x & 15) || ((uintptr_t)y & 7)) { set_error("cast_bf16: null or unaligned pointer"); return DLDKD_EINVAL; }
    DLDKD_LAUNCH(gdma::cast_bf16_kernel, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, (unsigned short*)y, n / 4);
    return check_launch("cast_bf16");
}

extern "C" int dldkd_gemm_bf16_nt16_ok(int M, int N, int K, int lda, int ldb) {
    return M > 0 && N > 0 && K >= gdma::BK16 && (K % gdma::BK16) == 0 && !(lda & 7) && !(ldb & 7) &&
           (long)127 * lda * 2 + 128 <= 0x7fffffffL && (long)127 * ldb * 2 + 128 <= 0x7fffffffL;
}

extern "C" int dldkd_gemm_bf16_nt16(const void* A, const void* B, const float* bias, float* C, int M, int N, int K, int lda, int ldb,
                                    int ldc, int relu, const unsigned char* row_flags, void* stream) {
    if (M < 0 || N < 0 || K < 0 || lda < K || ldb < K || ldc < N) { set_error("gemm_bf16_nt16: bad sizes"); return DLDKD_EINVAL; }
    if (M == 0 || N == 0) return DLDKD_OK;
    if (!A || !B || !C) { set_error("gemm_bf16_nt16: null pointer"); return DLDKD_EINVAL; }
    if (!dldkd_gemm_bf16_nt16_ok(M, N, K, lda, ldb) || (((uintptr_t)A | (uintptr_t)B) & 15)) {
        set_error("gemm_bf16_nt16: needs K %% 64 == 0, lda / ldb %% 8 == 0 and 16-byte aligned operands (M=%d N=%d K=%d)", M, N, K);
        return DLDKD_EINVAL;
    }
    gdma::Args16 p{(const unsigned short*)A, (const unsigned short*)B, bias, C, M, N, K, lda, ldb, ldc, relu != 0, 1.0f,
                   (M % gdma::BM == 0 && !((uintptr_t)row_flags & 3)) ? row_flags : nullptr, 1, 0, 0};
    constexpr int lds = gdma::NST * gdma::STAGE_B;
    static const bool ok = hipFuncSetAttribute((const void*)gdma::gemm_bf16_nt16_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds) == hipSuccess;
    (void)ok;
    const dim3 grid((N + gdma::BN - 1) / gdma::BN, (M + gdma::BM - 1) / gdma::BM, 1);
    DLDKD_LAUNCH(gdma::gemm_bf16_nt16_kernel, grid, dim3(256), lds, (hipStream_t)stream, p);
    return check_launch("gemm_bf16_nt16");
}

// C = act(A B^T + bias) with both fp32 operands given as TWO bf16 planes each (x = h + m: h = bf16(x), m = bf16(x - h)): the
// two-plane product a_h b_h + a_h b_m + a_m b_h of gemm_f32x3.hip (NPL = 2), smallest term first, as ONE pass of the LDS-DMA kernel
// over three K-long segments.  A_planes = [2][M][lda], B_planes = [2][N][ldb] (plane strides in elements).
extern "C" int dldkd_gemm_bf16_nt16_planes(const void* A_planes, const void* B_planes, const float* bias, float* C, int M, int N, int K, int lda,
                                           int ldb, int ldc, int relu, const unsigned char* row_flags, long a_plane_stride, long b_plane_stride,
                                           void* stream) {
    if (M < 0 || N < 0 || K < 0 || lda < K || ldb < K || ldc < N || a_plane_stride < 0 || b_plane_stride < 0) { set_error("gemm_bf16_nt16_planes: bad sizes"); return DLDKD_EINVAL; }
    if (M == 0 || N == 0) return DLDKD_OK;
    if (!A_planes || !B_planes || !C) { set_error("gemm_bf16_nt16_planes: null pointer"); return DLDKD_EINVAL; }
    if (!dldkd_gemm_bf16_nt16_ok(M, N, K, lda, ldb) || (((uintptr_t)A_planes | (uintptr_t)B_planes) & 15) || ((a_plane_stride | b_plane_stride) & 7)) {
        set_error("gemm_bf16_nt16_planes: needs K %% 64 == 0, lda / ldb / plane strides %% 8 == 0 and 16-byte aligned operands (M=%d N=%d K=%d)", M, N, K);
        return DLDKD_EINVAL;
    }
    gdma::Args16 p{(const unsigned short*)A_planes, (const unsigned short*)B_planes, bias, C, M, N, K, lda, ldb, ldc, relu != 0, 1.0f,
                   (M % gdma::BM == 0 && !((uintptr_t)row_flags & 3)) ? row_flags : nullptr, 3,
                   2 * a_plane_stride, 2 * b_plane_stride};                              // (m, h), (h, m), (h, h): byte offsets of the m planes
    constexpr int lds = gdma::NST * gdma::STAGE_B;
    static const bool ok = hipFuncSetAttribute((const void*)gdma::gemm_bf16_nt16_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds) == hipSuccess;
    (void)ok;
    const dim3 grid((N + gdma::BN - 1) / gdma::BN, (M + gdma::BM - 1) / gdma::BM, 1);
    DLDKD_LAUNCH(gdma::gemm_bf16_nt16_kernel, grid, dim3(256), lds, (hipStream_t)stream, p);
    return check_launch("gemm_bf16_nt16_planes");
}

// g16 (nv, L, D) / q16 (nq, D) bf16 -> the PoolArgs outputs (simpool_train.hip: dldkd_simpool_train_fwd_bf16in)
namespace dldkd {
int launch_simpool_pool_bf16_dma(const void* g16, const void* q16, int nv, int L, int nq, int D, const PoolArgs& pa, void* stream,
                                 long g_plane_stride, long q_plane_stride) {
    if (L < 1 || L > gdma::BM || !dldkd_gemm_bf16_nt16_ok(L, nq, D, D, D) || (((uintptr_t)g16 | (uintptr_t)q16) & 15)) {
        set_error("simpool_train_fwd_bf16in: needs L <= 128, D %% 64 == 0 and 16-byte aligned operands (L=%d nq=%d D=%d)", L, nq, D);
        return DLDKD_EINVAL;
    }
    gdma::Args16 p{(const unsigned short*)g16, (const unsigned short*)q16, nullptr, nullptr, L, nq, D, D, D, nq, 0, 1.0f, nullptr, 1, 0, 0};
    if (g_plane_stride > 0 && q_plane_stride > 0) {       // two bf16 planes per operand: the two-plane product in three K-long segments
        if ((g_plane_stride | q_plane_stride) & 7) { set_error("simpool_train_fwd_planes: plane strides must be multiples of 8 elements"); return DLDKD_EINVAL; }
        p.nseg = 3;
        p.am = 2 * g_plane_stride; p.bm = 2 * q_plane_stride;
    }
    constexpr int lds = gdma::NST * gdma::STAGE_B;
    static const bool ok = hipFuncSetAttribute((const void*)gdma::gemm_bf16_nt16_pool_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds) == hipSuccess;
    (void)ok;
    DLDKD_LAUNCH(gdma::gemm_bf16_nt16_pool_kernel, dim3((nq + gdma::BN - 1) / gdma::BN, 1, nv), dim3(256), lds, (hipStream_t)stream, p, pa);
    return check_launch("simpool_train_fwd (bf16 operands)");
}
}  // namespace dldkd

extern "C" int dldkd_gemm_bf16_nt_ok(int M, int N, int K, int lda, int ldb) {
    return M > 0 && N > 0 && K >= gdma::BK && (K % gdma::BK) == 0 && !(lda & 3) && !(ldb & 3) &&
           (long)127 * lda * 4 + 128 <= 0x7fffffffL && (long)127 * ldb * 4 + 128 <= 0x7fffffffL;
}

extern "C" int dldkd_gemm_bf16_nt(const float* A, const float* B, const float* bias, float* C, int M, int N, int K, int lda, int ldb,
                                  int ldc, int relu, const unsigned char* row_flags, void* stream) {
    if (M < 0 || N < 0 || K < 0 || lda < K || ldb < K || ldc < N) { set_error("gemm_bf16_nt: bad sizes"); return DLDKD_EINVAL; }
    if (M == 0 || N == 0) return DLDKD_OK;
    if (!A || !B || !C) { set_error("gemm_bf16_nt: null pointer"); return DLDKD_EINVAL; }
    if (!dldkd_gemm_bf16_nt_ok(M, N, K, lda, ldb) || (((uintptr_t)A | (uintptr_t)B) & 15)) {
        set_error("gemm_bf16_nt: needs K %% 32 == 0, lda / ldb %% 4 == 0 and 16-byte aligned operands (M=%d N=%d K=%d)", M, N, K);
        return DLDKD_EINVAL;
    }
    gdma::Args p{A, B, bias, C, M, N, K, lda, ldb, ldc, relu != 0, 1.0f,
                 (M % gdma::BM == 0 && !((uintptr_t)row_flags & 3)) ? row_flags : nullptr};
    constexpr int lds = gdma::NST * gdma::STAGE_B;
    static const bool ok = hipFuncSetAttribute((const void*)gdma::gemm_bf16_nt_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds) == hipSuccess;
    (void)ok;
    const dim3 grid((N + gdma::BN - 1) / gdma::BN, (M + gdma::BM - 1) / gdma::BM, 1);
    DLDKD_LAUNCH(gdma::gemm_bf16_nt_kernel, grid, dim3(256), lds, (hipStream_t)stream, p);
    return check_launch("gemm_bf16_nt");
}
