// gemm_bf16_tn: C[i, j] = sum_r A[r, i] * B[r, j] - the weight-gradient layout of the training step (dW = dY^T X: reference
// model_components.py:302,388-390,442 backward, both operands saved bf16 rows whose ROW index is the contraction index) with both
// operands brought HBM / L2 -> LDS by LDS-DMA exactly as they lie in memory and transposed on the way OUT of the LDS by gfx950's
// ds_read_b64_tr_b16 (cdna_hip_programming.md T10).  Replaces the register-staged kernels of gemm_bf16.hip for these products
// (gemm_bf16_dw_group_kernel 74 us, gemm_bf16_dw_dual_kernel 96 us per tower at the TVR batch): those load a bf16 PAIR per lane and
// instruction (the contraction index is the slow one in memory), convert through fp32 registers and transpose with their LDS
// writes, 32 rows per barrier.
//   * workgroup = 128 x 128 outputs, 4 waves of 64 x 64 (2 x 2 v_mfma_f32_32x32x16_bf16 tiles), row tiles of 64 contraction rows:
//     16 KiB per operand and stage, 2 stages, two workgroups per CU.
//   * LDS image of a row tile = [64 rows][128 columns] bf16 with 256-byte rows, 16-byte chunk ch of row r at position
//     ch ^ f(r), f(r) = ((r & 3) << 2) | ((r >> 2) & 3) (image (b) of T10), produced by permuting the per-lane SOURCE address
//     of the DMA; the transposed reads (a 16-lane group takes 4 rows x 16 columns) are conflict-free.
//   * one MFMA k-step = 16 contraction rows = two transposed reads per 32-column fragment: lane (c = lane & 31, h = lane >> 5)
//     receives rows 8 h .. 8 h + 7 of column c - the k order of the register-staged kernel, so with the same split plan the
//     planes are bit-identical to it.
//   * row flags (one per 32 rows, from the input projection's LayerNorm kernel): a half tile of padding is neither loaded nor
//     multiplied (its rows are never written by the row kernels: they may hold anything).
//   * DUAL (the training input projection's backward pass, gemm_bf16.hip "inproj_bwd"): a second accumulator set takes the same A
//     fragments against the 0 / 1 mask [B != 0], made from the B fragment in registers (3 packed VALU per dword).
//   * bias gradient: column sums of A from the A fragments (v_dot2c_f32_bf16 against ones) by the workgroups of the first column
//     tile.
#include "common.hpp"

namespace dldkd {
namespace gtn {

typedef unsigned short u16;
typedef short v4s __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int BI = 128, BJ = 128;
constexpr int kMaxGroups = 256;                     // 32-row groups per workgroup (4 x 64 flag bits): 8,192 rows per k-slice

struct Args {
    // (the names gemm_store_tile reads)
    float* C;
    int M, N, ldc;                                  // outputs: M = columns of A taken, N = columns of B taken
    float alpha;
    const float* bias;
    int relu;
    const u16* A;
    const u16* B;
    int R, lda, ldb;
    int split, tiles_per_split;                     // row tiles (64 rows) per k-slice; split > 1: C = [split][DUAL ? 2 : 1][M][ldc] planes
    const unsigned char* rflags;                    // per 32 rows, or null
    float* a_colsum;                                // [M] += column sums of A over the rows visited (workgroups of column tile 0), or null
};

__device__ __forceinline__ void glds16_s(uint32_t voff, const char* sbase, uint32_t lds_base) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(voff), "s"(sbase), "s"(lds_base) : "memory");
}
__device__ __forceinline__ int swz(int row) { return ((row & 3) << 2) | ((row >> 2) & 3); }

__device__ __forceinline__ bf16x8 tr_frag(const char* base, int off0, int off1) {
    typedef __attribute__((address_space(3))) v4s* lp;
    const v4s lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)(base + off0));
    const v4s hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)(base + off1));
    return bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}
// 1.0 where the bf16 element is not +-0, else 0 (the register-staged kernel's `value != 0.f ? 1.f : 0.f`)
__device__ __forceinline__ bf16x8 nonzero_mask(const bf16x8& b) {
    u32x4 u = __builtin_bit_cast(u32x4, b);
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        unsigned t = u[d] & 0x7fff7fffu, one;
        asm("v_pk_min_u16 %0, %1, %2" : "=v"(one) : "v"(t), "v"(0x00010001u));
        asm("v_pk_mul_lo_u16 %0, %1, %2" : "=v"(t) : "v"(one), "v"(0x3f803f80u));
        u[d] = t;
    }
    return __builtin_bit_cast(bf16x8, u);
}

template <bool DUAL, int NST, int BR>
__device__ __forceinline__ void tn_body(Args p, const int bx, const int by, const int bz, char* smem) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;
    constexpr int TILE_B = BR * 256;                    // one operand's row tile: BR rows x 128 bf16 (16 KiB at BR = 64)
    constexpr int STAGE_B = 2 * TILE_B;                 // A then B
    constexpr int G = BR / 32;                          // flag groups per tile
    constexpr int PPW = BR / 16;                        // 1-KiB pieces (4 rows) per wave, operand and tile
    const int i0 = by * BI, j0 = bx * BJ;
    const int nt_all = (p.R + BR - 1) / BR;
    const int t0 = p.split > 1 ? bz * p.tiles_per_split : 0;
    const int nt = p.split > 1 ? min(nt_all - t0, p.tiles_per_split) : nt_all;
    if (nt <= 0) return;
    if (p.split > 1) p.C += (size_t)bz * p.M * p.ldc * (DUAL ? 2 : 1);
    const uint32_t smem_lds = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(smem));

    // group flags of this workgroup's rows: bit (G t + h) of km = 32-row group h of tile t is wanted (in range and not padding)
    unsigned long long km[4];
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        const int hidx = 64 * w + lane;                                   // group index inside the workgroup's range
        const long row = ((long)t0 * G + hidx) * 32;
        bool on = hidx < G * nt && row < p.R;
        if (on && p.rflags != nullptr) on = p.rflags[(long)t0 * G + hidx] != 0;
        km[w] = __ballot(on);
    }
    auto half_on = [&](int t, int h) -> bool {
        const int idx = G * t + h, w = idx >> 6;
        const unsigned long long m = w == 0 ? km[0] : w == 1 ? km[1] : w == 2 ? km[2] : km[3];
        return ((m >> (idx & 63)) & 1ull) != 0;
    };

    // DMA: wave w brings rows 4 PPW w .. + 4 PPW - 1 of a tile (pieces of 4 rows = 1 KiB); lane -> row 4 q + (lane >> 4) of them, LDS
    // position lane & 15, which holds the row's chunk (lane & 15) ^ f(row)
    constexpr int wgroup_shift = BR == 64 ? 1 : 2;      // the wave's rows lie in flag group wave >> shift of the tile
    uint32_t voa[PPW], vob[PPW];
#pragma unroll
    for (int q = 0; q < PPW; ++q) {
        const int row = 4 * PPW * wave + 4 * q + (lane >> 4);
        const int ch = ((lane & 15) ^ swz(row)) << 4;
        voa[q] = (uint32_t)(row * p.lda * 2 + ch);
        vob[q] = (uint32_t)(row * p.ldb * 2 + ch);
    }
    const char* abase = reinterpret_cast<const char*>(p.A + (size_t)t0 * BR * p.lda + i0);
    const char* bbase = reinterpret_cast<const char*>(p.B + (size_t)t0 * BR * p.ldb + j0);
    const size_t astep = (size_t)BR * p.lda * 2, bstep = (size_t)BR * p.ldb * 2;
    auto issue = [&](int t, int stage) {
        if (!half_on(t, wave >> wgroup_shift)) return;
        const char* as = abase + (size_t)t * astep;
        const char* bs = bbase + (size_t)t * bstep;
        const uint32_t dst = smem_lds + stage * STAGE_B + (PPW * wave) * 1024;
#pragma unroll
        for (int q = 0; q < PPW; ++q) glds16_s(voa[q], as, dst + q * 1024);
#pragma unroll
        for (int q = 0; q < PPW; ++q) glds16_s(vob[q], bs, dst + TILE_B + q * 1024);
    };

    // transposed fragment reads of one k-step (16 rows): 16-lane group g = lane >> 4 takes rows 8 (g >> 1) + 4 e .. + 3 (e = 0, 1: the
    // two reads), columns 16 (g & 1) .. + 15 of the fragment's 32; lane 4 q + pp of the group supplies the address of row q, columns
    // 4 pp .. 4 pp + 3
    int offa[2][2], offb[2][2];
    {
        const int g = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3;
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int row = 8 * (g >> 1) + 4 * e + q;
            const int f = swz(row);
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int cha = ((wm + 32 * i) >> 3) + 2 * (g & 1) + (pp >> 1);
                const int chb = ((wn + 32 * i) >> 3) + 2 * (g & 1) + (pp >> 1);
                offa[i][e] = 256 * row + 16 * (cha ^ f) + 8 * (pp & 1);
                offb[i][e] = 256 * row + 16 * (chb ^ f) + 8 * (pp & 1) + TILE_B;
            }
        }
    }

    f32x16 acc[2][2], acc2[DUAL ? 2 : 1][DUAL ? 2 : 1];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                acc[i][j][r] = 0.f;
                if constexpr (DUAL) acc2[i][j][r] = 0.f;
            }
    const bool want_cs = p.a_colsum != nullptr && bx == 0 && (wave & 1) == 0;
    float cs[2] = {0.f, 0.f};

    // NST stages: tiles t + 1 .. t + NST - 1 are in flight while tile t is multiplied.  One barrier per tile: behind it every wave's
    // pieces of tile t have landed and every wave is done with tile t - 1, whose stage the DMA of tile t + NST - 1 then overwrites.
    // The counted wait leaves this wave's pieces of the younger tiles outstanding (8 per tile it brings, none for a half of padding).
#pragma unroll
    for (int d = 0; d < NST - 1; ++d)
        if (d < nt) issue(d, d);
    for (int t = 0; t < nt; ++t) {
        const int st = t % NST;
        if constexpr (NST == 2) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {
            // this wave's pieces of the tiles behind tile t stay in flight: 2 PPW per tile it brings
            int younger = 0;
#pragma unroll
            for (int d = 1; d < NST - 1; ++d) younger += (t + d < nt && half_on(t + d, wave >> wgroup_shift)) ? 1 : 0;
            static_assert(NST <= 4, "counted waits for at most two younger tiles");
            if (younger == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else if (younger == 1) asm volatile("s_waitcnt vmcnt(%0)" : : "n"(2 * PPW) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" : : "n"(4 * PPW) : "memory");
        }
        __syncthreads();
        if (t + NST - 1 < nt) issue(t + NST - 1, (t + NST - 1) % NST);
        const char* S = smem + st * STAGE_B;
#pragma unroll
        for (int kk = 0; kk < BR / 16; ++kk) {
            if (!half_on(t, kk >> 1)) continue;
            bf16x8 a[2], b[2], bm[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                b[j] = tr_frag(S + kk * 4096, offb[j][0], offb[j][1]);
                if constexpr (DUAL) bm[j] = nonzero_mask(b[j]);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                a[i] = tr_frag(S + kk * 4096, offa[i][0], offa[i][1]);
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
                    if constexpr (DUAL) acc2[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], bm[j], acc2[i][j], 0, 0, 0);
                }
                if (want_cs) {
                    const u32x4 u = __builtin_bit_cast(u32x4, a[i]);
#pragma unroll
                    for (int d = 0; d < 4; ++d)
                        asm("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(cs[i]) : "v"(0x3f803f80u), "v"(u[d]));      // (the builtin, fed the four
                                                                                                             // dwords of a fragment, read the first one four times)
                }
            }
        }
    }
    __syncthreads();                                           // (the epilogue stages its tiles through the operand region)
    if (want_cs) {
#pragma unroll
        for (int i = 0; i < 2; ++i) atomicAdd(p.a_colsum + i0 + wm + 32 * i + (lane & 31), cs[i]);      // (both lane halves: 8 rows each of every k-step)
    }
    gemm_store_tile(acc, p, i0, j0, wm, wn, lane, reinterpret_cast<float*>(smem) + wave * (32 * 72));
    if constexpr (DUAL) {
        p.C += (size_t)p.M * p.ldc;
        gemm_store_tile(acc2, p, i0, j0, wm, wn, lane, reinterpret_cast<float*>(smem) + wave * (32 * 72));
    }
}

template <bool DUAL, int NST, int BR>
__global__ __launch_bounds__(256, NST * BR <= 128 ? 2 : 1) void gemm_bf16_tn_kernel(const Args p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const Tile3 bid = xcd_tile_order();
    tn_body<DUAL, NST, BR>(p, bid.x, bid.y, bid.z, smem);
}

// the weight gradients of one fused training tower as ONE launch: output rows are 384-row blocks, each with its own operands
struct GroupArgs {
    Args base;                   // M = 384 n_blocks, N = 384
    const u16* A[5];             // block b: columns acol[b] .. + 383 of A[b] (R, lda[b])
    const u16* B[5];             // (R, 384)
    int lda[5], acol[5];
};
template <int NST, int BR>
__global__ __launch_bounds__(256, NST * BR <= 128 ? 2 : 1) void gemm_bf16_tn_group_kernel(const GroupArgs g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const Tile3 bid = xcd_tile_order();
    Args p = g.base;
    const int blk = bid.y / (kHidden / BI);
    // global output row m = 384 blk + c is column acol + c of this block's A: shift the base pointer instead of the index
    p.A = g.A[blk] + ((long)g.acol[blk] - (long)kHidden * blk);
    p.lda = g.lda[blk];
    p.B = g.B[blk];
    tn_body<false, NST, BR>(p, bid.x, bid.y, bid.z, smem);
}

}  // namespace gtn

// Row tiles per k-slice and the number of slices: ~384 workgroups (measured at the C3 step, tools/r05_ab_dw_tn_target.sh: 256 / 384 / 512 / 768
// -> dual 69 / 56 / 62 / 64 us, grouped 53 / 42 / 50 / 46 us, more slices = more plane traffic for the reduce), at least 4 tiles (256 rows) per
// slice, at most kMaxTiles
// the operand ring: (stages, rows per tile) = (2, 64) - 64 KiB, two workgroups per CU, one tile in flight behind the one being
// multiplied - or (4, 32): the same 64 KiB as four half-size tiles, two in flight (DLDKD_TN_RING=432; measured: see plan below)
static int tn_ring() {
    static const int n = [] { const char* e = getenv("DLDKD_TN_RING"); const int v = e ? atoi(e) : 0; return (v == 432 || v == 364 || v == 332) ? v : 264; }();
    return n;
}
static int tn_rows() { return tn_ring() % 100; }
int gemm_bf16_tn_plan(int M, int N, long R, int* tiles_per_split) {
    const int BR = tn_rows();
    const int tiles = (M / gtn::BI) * (N / gtn::BJ);
    const int nt = (int)((R + BR - 1) / BR);
    static const int target = [] { const char* e = getenv("DLDKD_TN_TARGET"); const int v = e ? atoi(e) : 0; return v > 0 ? v : 384; }();
    int split = (target + tiles - 1) / tiles;
    if (split > nt / 4) split = nt / 4;
    if (split < 1) split = 1;
    int per = (nt + split - 1) / split;
    if (per > gtn::kMaxGroups * 32 / BR) per = gtn::kMaxGroups * 32 / BR;
    *tiles_per_split = per;
    return (nt + per - 1) / per;
}

template <typename K, typename A>
static void tn_launch(K kernel, dim3 grid, int lds, hipStream_t stream, const A& args) {
    // the LDS limit of an instance is raised once per thread that launches it (instances share this function's type: a table, not a static)
    thread_local const void* seen[8] = {};
    bool known = false;
    for (const void* k : seen) known |= k == (const void*)kernel;
    if (!known) {
        (void)hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        for (auto& k : seen)
            if (k == nullptr) { k = (const void*)kernel; break; }
    }
    DLDKD_LAUNCH(kernel, grid, dim3(256), lds, stream, args);
}

bool gemm_bf16_tn_enabled() {
    static const bool on = [] { const char* e = getenv("DLDKD_DW_TN"); return !(e && e[0] == '0'); }();
    return on;
}

// shapes the kernel takes: whole 128-column tiles, whole 32-row flag groups, 16-byte aligned row pieces
bool gemm_bf16_tn_ok(int M, int N, long R, int lda, int ldb) {
    return M > 0 && N > 0 && R >= 32 && !(M % gtn::BI) && !(N % gtn::BJ) && !(R % 32) && !(lda & 7) && !(ldb & 7) &&
           (long)63 * lda * 2 + 256 <= 0x7fffffffL && (long)63 * ldb * 2 + 256 <= 0x7fffffffL && R <= 0x7fffffffL;
}

size_t gemm_bf16_tn_planes_bytes(int M, int N, long R, int dual) {
    int per = 0;
    const int split = gemm_bf16_tn_plan(M, N, R, &per);
    return (size_t)(split > 1 ? split : (dual ? 1 : 0)) * (dual ? 2 : 1) * M * N * sizeof(float);
}

// C (M, N) [DUAL: planes C, C + M N, always through `planes`] = A^T B.  planes: gemm_bf16_tn_planes_bytes.  Returns the number of
// k-slices written to `planes` (0: the result went straight to C; DUAL: >= 1) or a negative error code.
int launch_gemm_bf16_tn(const void* A, const void* B, float* C, int M, int N, long R, int lda, int ldb, int dual, void* planes,
                        const unsigned char* rflags, float* a_colsum, hipStream_t stream) {
    int per = 0;
    const int split = gemm_bf16_tn_plan(M, N, R, &per);
    gtn::Args p{C, M, N, N, 1.0f, nullptr, 0, (const gtn::u16*)A, (const gtn::u16*)B, (int)R, lda, ldb, split, per, rflags, a_colsum};
    if (split > 1 || dual) p.C = (float*)planes;
    const int ring = tn_ring(), lds = (ring / 100) * 2 * (ring % 100) * 256;
    const dim3 grid(N / gtn::BJ, M / gtn::BI, split);
#define DLDKD_TN_CASE(R_, NST_, BR_) case R_: if (dual) tn_launch(gtn::gemm_bf16_tn_kernel<true, NST_, BR_>, grid, lds, stream, p); \
                                              else tn_launch(gtn::gemm_bf16_tn_kernel<false, NST_, BR_>, grid, lds, stream, p); break;
    switch (ring) { DLDKD_TN_CASE(432, 4, 32) DLDKD_TN_CASE(332, 3, 32) DLDKD_TN_CASE(364, 3, 64) default: DLDKD_TN_CASE(264, 2, 64) }
#undef DLDKD_TN_CASE
    const int rc = check_launch("gemm_bf16_tn");
    return rc != DLDKD_OK ? rc : (split > 1 || dual ? split : 0);
}

// the tower's grouped weight gradients: blocks of 384 output rows; every A / B bf16.  Returns the k-slices written to `planes`
// (0: straight to dW) or a negative error code.
int launch_gemm_bf16_tn_group(const void* const* A, const int* lda, const int* acol, const void* const* B, int n_blocks, long R, float* dW,
                              void* planes, const unsigned char* rflags, float* a_colsum, hipStream_t stream) {
    const int M = kHidden * n_blocks;
    int per = 0;
    const int split = gemm_bf16_tn_plan(M, kHidden, R, &per);
    gtn::GroupArgs g{};
    g.base = gtn::Args{split > 1 ? (float*)planes : dW, M, kHidden, kHidden, 1.0f, nullptr, 0, nullptr, nullptr, (int)R, 0, kHidden, split, per, rflags, a_colsum};
    for (int b = 0; b < n_blocks; ++b) {
        g.A[b] = (const gtn::u16*)A[b]; g.B[b] = (const gtn::u16*)B[b]; g.lda[b] = lda[b]; g.acol[b] = acol[b];
    }
    const int ring = tn_ring(), lds = (ring / 100) * 2 * (ring % 100) * 256;
    const dim3 grid(kHidden / gtn::BJ, M / gtn::BI, split);
    switch (ring) {
        case 432: tn_launch(gtn::gemm_bf16_tn_group_kernel<4, 32>, grid, lds, stream, g); break;
        case 332: tn_launch(gtn::gemm_bf16_tn_group_kernel<3, 32>, grid, lds, stream, g); break;
        case 364: tn_launch(gtn::gemm_bf16_tn_group_kernel<3, 64>, grid, lds, stream, g); break;
        default: tn_launch(gtn::gemm_bf16_tn_group_kernel<2, 64>, grid, lds, stream, g); break;
    }
    const int rc = check_launch("gemm_bf16_tn (tower weight gradients)");
    return rc != DLDKD_OK ? rc : (split > 1 ? split : 0);
}

}  // namespace dldkd
