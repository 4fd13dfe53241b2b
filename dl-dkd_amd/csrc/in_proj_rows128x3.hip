// Parity-grade input projection: y = ReLU( LayerNorm(x) . W^T + b ) for the raw clip / word features, both branches (768 output
// columns) in one pass, with fp32-grade products on the bf16 matrix cores (the three-plane scheme of gemm_f32x3.hip).  Replaces
// LinearLayer.forward (reference method/model_components.py:305-312) on the inference path of PARITY mode, where
// LayerNorm (1.7 ms) + two gemm_f32x3 launches (5.1 ms) per 1024-video batch were 70 % of the gallery encode.
//
//   out[m][n] = relu( sum_k xhat[m][k] W'[n][k] + bb[n] ),   xhat = (x - mean[m]) * rstd[m]   (what the LayerNorm kernel computes),
//   W' = gamma (.) W in fp32, bb = W.beta + b    (dldkd_fold_ln_linear_planes; the reference rounds xhat*gamma+beta, this rounds
//   gamma*W and beta.W: the same 2^-24 class of error, no cancellation - unlike the bf16 kernel's rstd (x.W' - mean colsum) fold).
//   mean / rstd come from dldkd_row_meanrstd_f32 (the LayerNorm kernel's own two-pass reduction, bit for bit).
//
//   xhat = h + m + l and W' = h + m + l (bf16 planes, both splits exact); each product is the six plane products of order <= 2,
//   smallest first, accumulated in fp32 by v_mfma_f32_32x32x16_bf16: 6 MFMAs where the bf16 kernel has one.
//
// Structure (the measurements behind it: profiles/r02/ablation_k4_rows128.md):
//   * 4 waves per workgroup, one per SIMD; wave w owns all 128 rows x columns [192 w, 192 w + 192): 384 accumulator registers
//     (inline-asm MFMAs: column tiles 0-3 in AGPRs, 4-5 in VGPRs).  Template JT = 3: N = 384, 96 columns per wave - the
//     384-wide linears of the towers (dldkd_linear_f32x3_rows, no normalisation).
//   * a half-step = 16 k.  Row tile outermost: for row tile i, column tile j: 6 MFMAs on acc[i][j].  Only TWO row tiles' A planes
//     are live (24 registers): the planes of the next row tile (or of row tile 0 of the next half-step) are normalised, split and
//     converted piecewise in the MFMA shadow of the current one, each piece pinned between two groups of MFMAs by an empty asm.
//     The B planes of a column tile (3 KiB) are re-read from LDS for every row tile: 70 B/clk of the LDS's 256.
//   * W' planes (fragment order, split once per weight version) stream L2 -> LDS by LDS-DMA into a private ring per wave of
//     36 1-KiB fragments = two half-steps; a column tile's three slots are refilled (two half-steps ahead) right after row tile 3
//     has read them.  x streams HBM -> LDS as fp32 in half tiles (128 rows x 64 B, one per kk), XOR-swizzled by permuting the
//     DMA's source addresses; the half tile for half-step h + 2 is requested when half-step h has read its own (one barrier per
//     half-step hands the slot over and publishes the other half).  All 160 KiB of LDS; hand-counted vmcnt / lgkmcnt.
//   * epilogue: + bias, ReLU, staged through the (now idle) LDS and written as float4 rows.
#include <type_traits>

#include "common.hpp"

namespace dldkd {

// JT = 32-column tiles per wave: 6 (N = 768: two 384-wide outputs) or 3 (N = 384)
constexpr int YM = 128;
constexpr int YXHALF = YM * 64;             // x half tile: 128 rows x 16 fp32 = 8 KiB
template <int JT>
struct YCfg {
    static constexpr int NF = 3 * JT;               // W' fragments per wave per half-step: [JT column tiles][3 planes]
    static constexpr int WC = 32 * JT;              // columns per wave
    static constexpr int HALF = NF * 1024;          // bytes of them
    static constexpr int WRING = 2 * HALF;          // ring per wave: two half-steps (36 / 18 KiB)
    static constexpr int XBASE = 4 * WRING;         // x halves behind the four W' rings
    static constexpr int LDS = XBASE + 2 * YXHALF;  // 160 / 88 KiB
    static constexpr int W_STEP = 2 * 4 * HALF;     // bytes of W' planes per 32 k: [kk 2][wave 4][column tile JT][plane 3][1 KiB]
    static constexpr int SP = WC + 8;               // epilogue staging pitch (floats): the two lane halves hit disjoint banks
};

struct Rows128X3Args {
    const float* x;
    const float* mean;     // [M]
    const float* rstd;     // [M]
    const char* Wp;        // [K / 32][W_STEP]
    const float* bb;       // [N]
    float* y[2];           // columns [0, 384) -> y[0], [384, 768) -> y[1]; row stride ldy
    long M;
    int K;
    int relu;
    int ldy;
};

typedef float f32x2y __attribute__((ext_vector_type(2)));
typedef unsigned u32x4y __attribute__((ext_vector_type(4)));

template <int I, int N, typename F>
__device__ __forceinline__ void static_for_y(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for_y<I + 1, N>(f);
    }
}

__device__ __forceinline__ void mfma_a(f32x16& acc, const u32x4y& a, const bf16x8& b) {
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
}
__device__ __forceinline__ void mfma_v(f32x16& acc, const u32x4y& a, const bf16x8& b) {
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
}
__device__ __forceinline__ void y_m0(uint32_t lds_base) { asm volatile("s_mov_b32 m0, %0" : : "s"(lds_base) : "memory"); }
template <int OFF>                      // M0 set at least one instruction earlier; the immediate moves BOTH addresses
__device__ __forceinline__ void y_glds(uint32_t voff, const char* sbase) {
    asm volatile("global_load_lds_dwordx4 %0, %1 offset:%2" : : "v"(voff), "s"(sbase), "i"(OFF) : "memory");
}
template <int OFF, typename T>
__device__ __forceinline__ void y_lds16(T& dst, uint32_t addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "i"(OFF) : "memory");
}

// two fp32 -> (h, m, l) bf16 pairs; x = h + m + l to 24 bits, both subtractions exact (gemm_f32x3.hip)
__device__ __forceinline__ void split_pair(float v0, float v1, unsigned& h, unsigned& m, unsigned& l) {
    typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
    h = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2y{v0, v1}, bf2));
    const float r0 = v0 - __builtin_bit_cast(float, h << 16), r1 = v1 - __builtin_bit_cast(float, h & 0xffff0000u);
    m = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2y{r0, r1}, bf2));
    const float s0 = r0 - __builtin_bit_cast(float, m << 16), s1 = r1 - __builtin_bit_cast(float, m & 0xffff0000u);
    l = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2y{s0, s1}, bf2));
}

template <int JT>
__global__ __launch_bounds__(256, 1) void in_proj_rows128x3_kernel(const Rows128X3Args p) {
    using C = YCfg<JT>;
    constexpr int YHALF = C::HALF, YWRING = C::WRING, YXBASE = C::XBASE, YW_STEP = C::W_STEP, YSP = C::SP, YWC = C::WC, NF = C::NF;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const long m0 = (long)blockIdx.x * YM;
    const int nk = p.K / 32;
    const uint32_t smem_lds = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(smem));
    const uint32_t ring_base = smem_lds + wave * YWRING;               // wave-uniform
    const uint32_t ring_lds = ring_base + lane * 16;
    const uint32_t wlane = lane * 16;
    const char* wsrc_w = p.Wp + (size_t)wave * YHALF;                  // + k-step * YW_STEP + kk * 4 * YHALF + fragment * 1024

    // x LDS-DMA: a half tile is 8 pieces of 1 KiB (16 rows x 64 B); wave w issues pieces 2 w, 2 w + 1.  Lane -> row 16 t + lane / 4,
    // LDS chunk lane % 4; it fetches the global chunk that belongs there in the swizzled image (chunk c of row r at c ^ ((r >> 2) & 3)).
    // Rows past M are clamped into the tile (they feed accumulator rows that are never stored).
    const long rv = p.M - m0 < YM ? p.M - m0 : YM;
    uint32_t voffx[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        int row = 16 * (2 * wave + q) + (lane >> 2);
        const int chunk = (lane & 3) ^ ((row >> 2) & 3);
        if (row > rv - 1) row = (int)rv - 1;
        voffx[q] = (uint32_t)((long)row * p.K * 4 + chunk * 16);
    }
    const char* xsrc = reinterpret_cast<const char*>(p.x + m0 * p.K);   // + k-step * 128 + kk * 64 bytes
    // A-fragment reads: lane (r = lane & 31, hh = lane >> 5) takes chunks 2 hh + e of row 32 i + r of the half tile
    uint32_t va[2];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        const int r = lane & 31, c = 2 * (lane >> 5) + e;
        va[e] = smem_lds + YXBASE + r * 64 + ((c ^ ((r >> 2) & 3)) << 4);    // + kk * YXHALF + i * 2048
    }
    // LayerNorm statistics of the lane's row in each row tile
    float mean_[4], rstd_[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        long r = m0 + 32 * i + (lane & 31);
        if (r > p.M - 1) r = p.M - 1;
        mean_[i] = p.mean ? p.mean[r] : 0.f;         // no statistics: a plain linear layer ((v - 0) * 1 is exact)
        rstd_[i] = p.rstd ? p.rstd[r] : 1.f;
    }

    f32x16 acc[4][JT];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < JT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    u32x4y ap[2][3];         // A planes (h, m, l) of the row tile in use [i & 1] and of the one being converted
    f32x4 raw[2];            // its 8 fp32 on their way from LDS
    bf16x8 b[2][3];          // B planes of the column tile in use and of the next one

    // k-steps in natural order for every workgroup (the bf16 kernel rotates them per XCD against memory-channel conflicts; here a
    // row's result must not depend on which tile of which batch it falls into - parity mode is batch-invariant bit for bit - and
    // with six MFMAs per product the memory system has six times longer)
    auto rot = [&](int k) { return k; };
    auto wrap = [&](int k) { return k < nk ? k : k - nk; };

    // normalise + split the 8 floats in raw[] (row tile ti) into the three planes of dst
    auto conv_half = [&](u32x4y (&dst)[3], int e, const f32x4& v, float mu, float rs) {
        unsigned h0, m0_, l0, h1, m1, l1;
        split_pair((v[0] - mu) * rs, (v[1] - mu) * rs, h0, m0_, l0);
        split_pair((v[2] - mu) * rs, (v[3] - mu) * rs, h1, m1, l1);
        dst[0][2 * e] = h0, dst[0][2 * e + 1] = h1;
        dst[1][2 * e] = m0_, dst[1][2 * e + 1] = m1;
        dst[2][2 * e] = l0, dst[2][2 * e + 1] = l1;
    };

    // prologue: both x halves of k-step 0, W' of half-steps 0 and 1
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
        const char* src = xsrc + (size_t)rot(0) * 128 + kk * 64;
        y_m0(smem_lds + YXBASE + kk * YXHALF + (2 * wave) * 1024);
        asm volatile("s_nop 0");
        y_glds<0>(voffx[0], src);
        y_m0(smem_lds + YXBASE + kk * YXHALF + (2 * wave + 1) * 1024);
        asm volatile("s_nop 0");
        y_glds<0>(voffx[1], src);
    }
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
        const char* src = wsrc_w + (size_t)rot(0) * YW_STEP + kk * 4 * YHALF;
        static_for_y<0, NF>([&](auto fc) {
            constexpr int f = decltype(fc)::value;
            if constexpr ((f & 3) == 0) { y_m0(ring_base + kk * YHALF + (f >> 2) * 4096); asm volatile("s_nop 0"); }
            y_glds<(f & 3) * 1024>(wlane, src + (f >> 2) * 4096);
        });
    }
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    y_lds16<0>(raw[0], va[0]);
    y_lds16<0>(raw[1], va[1]);
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(raw[0]), "+v"(raw[1]) : : "memory");
    conv_half(ap[0], 0, raw[0], mean_[0], rstd_[0]);
    conv_half(ap[0], 1, raw[1], mean_[0], rstd_[0]);
    y_lds16<0>(b[0][0], ring_lds);
    y_lds16<1024>(b[0][1], ring_lds);
    y_lds16<2048>(b[0][2], ring_lds);

    // One half-step (16 k), KK = h & 1 (compile time: ring half, x half).  Blocks (i, j) = 6 MFMAs on acc[i][j]; at the top of a
    // block the B planes of the NEXT block are requested, then one counted wait covers this block's planes.
    //   VMEM queue per half-step (all in row tile 3): 2 x pieces, then 3 refills in each block (3, j): 2 + 3 JT operations.  A
    //   column tile c >= 1 refilled two half-steps ago has 3 (JT - 1 - c) + 2 + 3 JT operations behind it when block (0, c - 1)
    //   asks for it; column tile 0 is asked for at block (3, JT - 1) of the half-step before: 3 (JT - 1) + 2 + 3 (JT - 1).
    //   LDS queue: 3 plane reads per block (+ 2 raw reads at blocks (i, 0)), so "this block's planes have landed" is
    //   lgkmcnt(5) at j = 0 and lgkmcnt(3) otherwise (at j = 1 that also covers the raw reads).
    auto half_step = [&](auto kkc, int kt) {
        constexpr int KK = decltype(kkc)::value;
        // refills go two half-steps ahead = next k-step, same kk; x: next k-step, same kk (this half is free after row tile 2)
        const char* wnext = wsrc_w + (size_t)rot(wrap(kt + 1)) * YW_STEP + KK * 4 * YHALF;
        const char* xnext = xsrc + (size_t)rot(wrap(kt + 1)) * 128 + KK * 64;
        // raw reads of row tile i + 1 (same half), or of row tile 0 of the other half for the next half-step
        static_for_y<0, 4>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            constexpr int CUR = i & 1, NXT = CUR ^ 1;
            static_for_y<0, JT>([&](auto jc) {
                constexpr int j = decltype(jc)::value;
                constexpr int BC = (i * JT + j) & 1, BN = BC ^ 1;     // 4 JT blocks per half-step (even): the parity carries over
                // ---- top of the block: requests for the next block
                if constexpr (i == 3 && j == 0) {
                    // own x pieces of the OTHER half (requested a half-step ago) have landed: NF refills were issued after them
                    asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" : : "n"(NF) : "memory");
                    y_m0(smem_lds + YXBASE + KK * YXHALF + (2 * wave) * 1024);
                }
                if constexpr (j < JT - 1) {
                    // planes of (i, j + 1): for i = 0 they were refilled two half-steps ago, with 3 (JT - 1 - c) + (2 + NF)
                    // operations issued since (c = j + 1)
                    if constexpr (i == 0) asm volatile("s_waitcnt vmcnt(%0)" : : "n"(6 * JT - 1 - 3 * (j + 1)) : "memory");
                    y_lds16<KK * YHALF + (3 * (j + 1) + 0) * 1024>(b[BN][0], ring_lds);
                    y_lds16<KK * YHALF + (3 * (j + 1) + 1) * 1024>(b[BN][1], ring_lds);
                    y_lds16<KK * YHALF + (3 * (j + 1) + 2) * 1024>(b[BN][2], ring_lds);
                } else if constexpr (i < 3) {
                    y_lds16<KK * YHALF + 0>(b[BN][0], ring_lds);       // (i + 1, 0)
                    y_lds16<KK * YHALF + 1024>(b[BN][1], ring_lds);
                    y_lds16<KK * YHALF + 2048>(b[BN][2], ring_lds);
                } else {
                    // (0, 0) of the next half-step, the other ring half: 3 (JT - 1) + 2 + 3 (JT - 1) operations since its refill
                    asm volatile("s_waitcnt vmcnt(%0)" : : "n"(6 * JT - 4) : "memory");
                    y_lds16<(KK ^ 1) * YHALF + 0>(b[BN][0], ring_lds);
                    y_lds16<(KK ^ 1) * YHALF + 1024>(b[BN][1], ring_lds);
                    y_lds16<(KK ^ 1) * YHALF + 2048>(b[BN][2], ring_lds);
                }
                if constexpr (j == 0) {                                // fp32 of the row tile to convert during this one
                    if constexpr (i < 3) {
                        y_lds16<KK * YXHALF + (i + 1) * 2048>(raw[0], va[0]);
                        y_lds16<KK * YXHALF + (i + 1) * 2048>(raw[1], va[1]);
                    } else {
                        y_lds16<(KK ^ 1) * YXHALF>(raw[0], va[0]);
                        y_lds16<(KK ^ 1) * YXHALF>(raw[1], va[1]);
                    }
                    asm volatile("s_waitcnt lgkmcnt(5)" : "+v"(b[BC][0]), "+v"(b[BC][1]), "+v"(b[BC][2]) : : "memory");
                } else if constexpr (j == 1) {
                    asm volatile("s_waitcnt lgkmcnt(3)"
                                 : "+v"(b[BC][0]), "+v"(b[BC][1]), "+v"(b[BC][2]), "+v"(raw[0]), "+v"(raw[1])
                                 :
                                 : "memory");
                } else {
                    asm volatile("s_waitcnt lgkmcnt(3)" : "+v"(b[BC][0]), "+v"(b[BC][1]), "+v"(b[BC][2]) : : "memory");
                }
                // ---- 6 MFMAs, smallest terms first; the conversion of the next row tile rides between them (blocks 1-4: one
                // quarter each, pinned by the empty asm statements that name its inputs and outputs)
                constexpr int ni = (i + 1) & 3;
                auto M = [&](const u32x4y& a, const bf16x8& bb_) {
                    if constexpr (j < 4) mfma_a(acc[i][j], a, bb_);
                    else mfma_v(acc[i][j], a, bb_);
                };
                M(ap[CUR][2], b[BC][0]);
                M(ap[CUR][0], b[BC][2]);
                auto quarter = [&](auto qc) {                          // a quarter of the next row tile's conversion
                    constexpr int q = decltype(qc)::value, e = q >> 1, hpart = q & 1;   // float4 e, its first / second pair
                    asm volatile("" : "+v"(raw[e]) : : "memory");
                    unsigned h, m, l;
                    const f32x4 v = raw[e];
                    split_pair((v[2 * hpart] - mean_[ni]) * rstd_[ni], (v[2 * hpart + 1] - mean_[ni]) * rstd_[ni], h, m, l);
                    ap[NXT][0][2 * e + hpart] = h;
                    ap[NXT][1][2 * e + hpart] = m;
                    ap[NXT][2][2 * e + hpart] = l;
                    asm volatile("" : "+v"(ap[NXT][0]), "+v"(ap[NXT][1]), "+v"(ap[NXT][2]) : : "memory");
                };
                constexpr int QPB = JT >= 5 ? 1 : 2;                   // quarters per block, from block 1 on
                constexpr int q0 = (j - 1) * QPB;
                if constexpr (j >= 1 && q0 < 4) quarter(std::integral_constant<int, q0>{});
                if constexpr (i == 3 && j == 0) {                      // this half's x for the next k-step (M0 set after the barrier)
                    y_glds<0>(voffx[0], xnext);
                    y_m0(smem_lds + YXBASE + KK * YXHALF + (2 * wave + 1) * 1024);
                }
                M(ap[CUR][1], b[BC][1]);
                if constexpr (i == 3 && j == 0) y_glds<0>(voffx[1], xnext);
                M(ap[CUR][1], b[BC][0]);
                if constexpr (QPB == 2 && j >= 1 && q0 + 1 < 4) quarter(std::integral_constant<int, q0 + 1>{});
                if constexpr (i == 3) {                                // this column tile's three ring slots are free: refill
                    // slots 3 j .. 3 j + 2 of this ring half: M0 base per 4 fragments (the immediate spans 4 KiB)
                    static_for_y<0, 3>([&](auto pc) {
                        constexpr int f = 3 * j + decltype(pc)::value;
                        y_m0(ring_base + KK * YHALF + (f >> 2) * 4096);
                        asm volatile("s_nop 0");
                        y_glds<(f & 3) * 1024>(wlane, wnext + (f >> 2) * 4096);
                    });
                }
                M(ap[CUR][0], b[BC][1]);
                M(ap[CUR][0], b[BC][0]);
            });
        });
    };

    for (int kt = 0; kt < nk; ++kt) {
        half_step(std::integral_constant<int, 0>{}, kt);
        half_step(std::integral_constant<int, 1>{}, kt);
    }
    // the look-ahead requests past the end are still landing in the rings, which become staging space below
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier\n\ts_nop 15\n\ts_nop 15" ::: "memory");

    // epilogue: wave w writes columns [192 w, 192 w + 192) = branch w / 2, columns (w & 1) * 192 .., one 32-row tile at a time
    // through its own LDS region (LDS operations of one wave execute in order: no barrier, no wait)
    float* stg = reinterpret_cast<float*>(smem + wave * YWRING);      // 32 x YSP floats = 25.6 KiB of the wave's 36
    const int col0 = wave * YWC;                                       // 384 is a multiple of YWC: one output per wave
    float* outb = (col0 >= kHidden ? p.y[1] : p.y[0]) + (col0 % kHidden) + (size_t)m0 * p.ldy;
    const bool relu = p.relu, full = m0 + YM <= p.M;
    const int hrow = 4 * (lane >> 5);
    float* wr = stg + hrow * YSP + (lane & 31);
    float bbn[JT];
#pragma unroll
    for (int j = 0; j < JT; ++j) bbn[j] = p.bb[wave * YWC + 32 * j + (lane & 31)];
    constexpr int F4R = 8 * JT;                                        // float4 per row of the wave's column range
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int j = 0; j < JT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float v = acc[i][j][r] + bbn[j];
                if (relu) v = fmaxf(v, 0.f);
                wr[((r & 3) + 8 * (r >> 2)) * YSP + 32 * j] = v;
            }
        f32x4 o[4 * JT];
#pragma unroll
        for (int it = 0; it < 4 * JT; ++it) {
            const int idx = lane + 64 * it;
            o[it] = *reinterpret_cast<const f32x4*>(stg + (idx / F4R) * YSP + 4 * (idx % F4R));
        }
#pragma unroll
        for (int it = 0; it < 4 * JT; ++it) {
            const int idx = lane + 64 * it;
            const long row = m0 + 32 * i + idx / F4R;
            if (full || row < p.M) *reinterpret_cast<f32x4*>(outb + (size_t)(32 * i + idx / F4R) * p.ldy + 4 * (idx % F4R)) = o[it];
        }
    }
}

// W' = gamma (.) W split into three bf16 planes in the kernel's fragment order; bb = W.beta + b.  One wave per output column.
__global__ __launch_bounds__(256) void fold_ln_linear_planes_kernel(const float* __restrict__ W, const float* __restrict__ bias,
                                                                    const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                    int N, int K, int n_offset, int jt, unsigned short* __restrict__ Wp,
                                                                    float* __restrict__ bb) {
    const int lane = threadIdx.x & 63;
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= N) return;
    const int ng = n_offset + n, ct = ng >> 5, col = ng & 31;          // global column tile = jt * wave + j
    const int w = ct / jt, j = ct % jt;
    float t = 0.f;
    for (int k = lane; k < K; k += 64) {
        const float wv = W[(size_t)n * K + k];
        const float wf = gamma ? wv * gamma[k] : wv;
        const unsigned short h = f32_to_bf16_bits(wf);
        const float r1 = wf - bf16_bits_to_f32(h);
        const unsigned short m = f32_to_bf16_bits(r1);
        const unsigned short l = f32_to_bf16_bits(r1 - bf16_bits_to_f32(m));
        const int kt = k >> 5, kk = (k >> 4) & 1, half = (k >> 3) & 1, e = k & 7;
        // [kt][kk][wave][j][plane][lane = 32 half + col][8]
        const size_t base = ((((size_t)kt * 2 + kk) * 4 + w) * jt + j) * 3;
        Wp[((base + 0) * 64 + half * 32 + col) * 8 + e] = h;
        Wp[((base + 1) * 64 + half * 32 + col) * 8 + e] = m;
        Wp[((base + 2) * 64 + half * 32 + col) * 8 + e] = l;
        if (beta) t += wv * beta[k];
    }
    t = wave_sum(t);
    if (lane == 0) bb[ng] = t + (bias ? bias[n] : 0.f);
}

// mean and rstd of every row exactly as layernorm_kernel (encoder_f32.hip) computes them: two passes over registers.
template <int MAXV>
__global__ __launch_bounds__(256) void row_meanrstd_kernel(const float* __restrict__ x, float* __restrict__ mean_out,
                                                           float* __restrict__ rstd_out, long M, int D, float eps) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const int nv = D >> 2;
    const f32x4* xr = reinterpret_cast<const f32x4*>(x + row * D);
    f32x4 v[MAXV];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int c = lane + 64 * i;
        v[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (c < nv) {
            v[i] = xr[c];
            s += v[i][0] + v[i][1] + v[i][2] + v[i][3];
        }
    }
    const float mean = wave_sum(s) / D;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int c = lane + 64 * i;
        if (c < nv) {
#pragma unroll
            for (int e = 0; e < 4; ++e) { const float d = v[i][e] - mean; q += d * d; }
        }
    }
    const float rstd = rsqrtf(wave_sum(q) / D + eps);
    if (lane == 0) { mean_out[row] = mean; rstd_out[row] = rstd; }
}

}  // namespace dldkd

using namespace dldkd;

extern "C" int dldkd_in_proj_f32x3_rows128_ok(int K) { return K >= 64 && K % 32 == 0 && K <= 16 * 256 && (long)127 * K * 4 + 128 <= 0xFFFFFFFFL; }

extern "C" int dldkd_row_meanrstd_f32(const float* x, float* mean, float* rstd, long M, int D, float eps, void* stream) {
    if (M < 0 || D < 4 || (D & 3) || D > 16 * 256) { set_error("row_meanrstd: D must be a multiple of 4, at most 4096 (M=%ld D=%d)", M, D); return DLDKD_EINVAL; }
    if (M == 0) return DLDKD_OK;
    if (!x || !mean || !rstd || ((uintptr_t)x & 15)) { set_error("row_meanrstd: null or unaligned pointer"); return DLDKD_EINVAL; }
    const dim3 grid((unsigned)((M + 3) / 4)), block(256);
    const int nv = D >> 2;
    hipStream_t s = (hipStream_t)stream;
    if (nv <= 2 * 64) DLDKD_LAUNCH(row_meanrstd_kernel<2>, grid, block, 0, s, x, mean, rstd, M, D, eps);
    else if (nv <= 4 * 64) DLDKD_LAUNCH(row_meanrstd_kernel<4>, grid, block, 0, s, x, mean, rstd, M, D, eps);
    else if (nv <= 8 * 64) DLDKD_LAUNCH(row_meanrstd_kernel<8>, grid, block, 0, s, x, mean, rstd, M, D, eps);
    else DLDKD_LAUNCH(row_meanrstd_kernel<16>, grid, block, 0, s, x, mean, rstd, M, D, eps);
    return check_launch("row_meanrstd");
}

extern "C" int dldkd_pack_linear_planes(const float* W, const float* bias, const float* gamma, const float* beta, int N, int K,
                                       int n_offset, int n_total, void* Wplanes, float* bb, void* stream) {
    if (N < 1 || K < 32 || (K % 32) || (n_total != 384 && n_total != 768) || n_offset < 0 || n_offset + N > n_total || (n_offset % 32) ||
        (N % 32)) {
        set_error("pack_linear_planes: need K a multiple of 32, n_total 384 or 768 and whole 32-column tiles inside it");
        return DLDKD_EINVAL;
    }
    if (!W || !Wplanes || !bb || (gamma == nullptr) != (beta == nullptr)) { set_error("pack_linear_planes: null pointer"); return DLDKD_EINVAL; }
    DLDKD_LAUNCH(fold_ln_linear_planes_kernel, dim3((N + 3) / 4), dim3(256), 0, (hipStream_t)stream, W, bias, gamma, beta, N, K,
                 n_offset, n_total / 128, (unsigned short*)Wplanes, bb);
    return check_launch("pack_linear_planes");
}

extern "C" int dldkd_fold_ln_linear_planes(const float* W, const float* bias, const float* gamma, const float* beta, int N, int K,
                                           int n_offset, void* Wplanes, float* bb, void* stream) {
    if (!gamma || !beta) { set_error("fold_ln_linear_planes: null pointer"); return DLDKD_EINVAL; }
    return dldkd_pack_linear_planes(W, bias, gamma, beta, N, K, n_offset, 768, Wplanes, bb, stream);
}

extern "C" int dldkd_linear_f32x3_rows(const float* x, const float* mean, const float* rstd, const void* Wplanes, const float* bb,
                                       float* y0, float* y1, long M, int N, int K, int ldy, int relu, void* stream) {
    if (M < 0 || (N != 384 && N != 768) || !dldkd_in_proj_f32x3_rows128_ok(K) || ldy < 384 || (ldy & 3)) {
        set_error("linear_f32x3_rows: N must be 384 or 768, K a multiple of 32 in [64, 4096], ldy >= 384 (M=%ld N=%d K=%d ldy=%d)", M, N, K, ldy);
        return DLDKD_EINVAL;
    }
    if (M == 0) return DLDKD_OK;
    if (!x || !Wplanes || !bb || !y0 || (N == 768 && !y1) || (mean == nullptr) != (rstd == nullptr)) {
        set_error("linear_f32x3_rows: null pointer");
        return DLDKD_EINVAL;
    }
    if (((uintptr_t)x | (uintptr_t)y0 | (uintptr_t)y1 | (uintptr_t)Wplanes) & 15) { set_error("linear_f32x3_rows: unaligned buffer"); return DLDKD_EINVAL; }
    Rows128X3Args p{x, mean, rstd, (const char*)Wplanes, bb, {y0, y1}, M, K, relu != 0, ldy};
    const dim3 grid((unsigned)((M + YM - 1) / YM));
    if (N == 768) {
        static const bool ok = hipFuncSetAttribute((const void*)in_proj_rows128x3_kernel<6>, hipFuncAttributeMaxDynamicSharedMemorySize, YCfg<6>::LDS) == hipSuccess;
        (void)ok;
        DLDKD_LAUNCH(in_proj_rows128x3_kernel<6>, grid, dim3(256), YCfg<6>::LDS, (hipStream_t)stream, p);
    } else {
        static const bool ok = hipFuncSetAttribute((const void*)in_proj_rows128x3_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize, YCfg<3>::LDS) == hipSuccess;
        (void)ok;
        DLDKD_LAUNCH(in_proj_rows128x3_kernel<3>, grid, dim3(256), YCfg<3>::LDS, (hipStream_t)stream, p);
    }
    return check_launch("linear_f32x3_rows");
}

extern "C" int dldkd_in_proj_f32x3_rows128(const float* x, const float* mean, const float* rstd, const void* Wplanes, const float* bb,
                                           float* y0, float* y1, long M, int K, int relu, void* stream) {
    if (!mean || !rstd) { set_error("in_proj_f32x3_rows128: null pointer"); return DLDKD_EINVAL; }
    return dldkd_linear_f32x3_rows(x, mean, rstd, Wplanes, bb, y0, y1, M, 768, K, 384, relu, stream);
}
