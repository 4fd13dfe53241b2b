// NOTE (round 5): the 16-bit MFMA operands of this file are h16 = IEEE fp16, not bf16 (common.hpp says why; the text below and the
// identifiers still say "bf16" where they mean "the 16-bit operand": bf16x8 is the 8 x 16-bit container, whatever the format).
// K4b: the two-branch input projection y = ReLU( LayerNorm(x) . W^T + b ) (LinearLayer.forward, reference
// method/model_components.py:305-312) on rows that are ALREADY stored the way the MFMA consumes them: bf16 features plus
// the row's fp32 LayerNorm statistics, both written once by the ingest pass (dldkd_rows_to_h16_stats, ingest.hip) when a
// dataset's raw features become device-resident.  in_proj_rows128_kernel (K4) reads fp32 rows and spends its issue slots on
// what this kernel no longer does: 16 fp32 fragment reads + 32 v_cvt_pk_bf16_f32 + the branch-free LayerNorm sums per k-step,
// against 48 MFMAs (profiles/r02/ablation_k4_rows128.md: issue-bound at 1.38 GHz).  Here a k-step is 48 MFMAs + 8 A-fragment
// reads + 12 B-fragment reads + 12 W' LDS-DMAs (+ 4 x pieces every second step), half the HBM and LDS bytes per row, and the
// numbers are the same: K4 rounds x to bf16 (RNE) before the MFMA and takes the statistics from the fp32 values - so does
// the ingest pass.
//
// Structure = K4's (in_proj_rows128.hip; read its header first): 4 waves, wave w owns all 128 rows x columns [192 w, 192 w + 192)
// = 4 x 6 tiles of mfma_f32_32x32x16_bf16 (AGPR + VGPR accumulators), W' (LayerNorm-folded, fragment order: the SAME blob as
// K4's) through a private 24-fragment LDS-DMA ring per wave, one persistent workgroup per CU, k-steps rotated per XCD,
// hand-counted waits.  What differs:
//   * an x ring slot (16 KiB) is 128 rows x 64 k of bf16 = TWO k-steps: the same 128-byte rows and XOR swizzle as K4's fp32
//     k-tile, filled by the same 4 pieces per wave, every second k-step (even steps, groups 8-11) three x tiles ahead;
//   * an A fragment is ONE ds_read_b128 (8 bf16 of one row), read a k-step ahead straight into the other fragment set;
//   * mean / rstd of the tile's rows come from memory at the tile boundary (through the same free x slot K4 uses).
#include <type_traits>

#include "common.hpp"

namespace dldkd {
namespace k4b {

constexpr int RM = 128, RK = 32, RWC = 192;
constexpr int RSLOT = 6 * 2 * 1024;       // bytes of W' per wave per k-step: [6 column tiles][2 kk][64 lanes][16 B]
constexpr int XSLOT = RM * 64 * 2;        // one x tile: 128 rows x 64 k bf16 = 16 KiB = two k-steps
constexpr int WREGION = 2 * RSLOT + XSLOT;  // LDS per wave: [W' ring, 2 k-steps][x ring slot number `wave`] = 40 KiB
constexpr int RW_TILE = 768 * RK * 2;     // bytes of W' per k-step for all 24 column tiles

struct Args {
    const unsigned short* x;   // (M, K) bf16
    const float* mean;         // (M)
    const float* rstd;         // (M)
    const char* Wf;            // [k-step][24 column tiles][2][64][8] bf16 (dldkd_fold_ln_linear_h16_frag)
    const float* cs;           // [768] colsum of W'
    const float* bb;           // [768] W.beta + b
    float* y[2];               // columns [0, 384) -> y[0], [384, 768) -> y[1]; row stride 384
    int y16;                   // y[] are bf16 rows (row stride 384 bf16) - what the fused tower reads (dldkd_tower_seq_h16_rows16)
    long M;
    int K;
    int relu;
    const int32_t* grp;        // row-group table or null (as in K4: group i of tile t starts at row grp[4 t + i] of x, mean, rstd AND y)
    long n_tiles;
};

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

__device__ __forceinline__ void mfma_agpr(f32x16& acc, const u32x4& a, const bf16x8& b) {
    asm volatile(DLDKD_H16_MFMA32 " %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
}
__device__ __forceinline__ void mfma_vgpr(f32x16& acc, const u32x4& a, const bf16x8& b) {
    asm volatile(DLDKD_H16_MFMA32 " %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
}
__device__ __forceinline__ void glds_m0(uint32_t lds_base) { asm volatile("s_mov_b32 m0, %0" : : "s"(lds_base) : "memory"); }
template <int OFF>                      // M0 set at least one instruction earlier by glds_m0
__device__ __forceinline__ void glds16_m(uint32_t voff, const char* sbase) {
    asm volatile("global_load_lds_dwordx4 %0, %1 offset:%2" : : "v"(voff), "s"(sbase), "i"(OFF) : "memory");
}
template <int OFF>
__device__ __forceinline__ void glds16_s(uint32_t voff, const char* sbase, uint32_t lds_base) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 offset:%3"
                 :
                 : "v"(voff), "s"(sbase), "s"(lds_base), "i"(OFF)
                 : "memory");
}
template <int OFF, typename T>
__device__ __forceinline__ void lds_read16(T& dst, uint32_t addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "i"(OFF) : "memory");
}
template <int OFF>
__device__ __forceinline__ void gstore32(uint32_t voff, float v, const char* sbase) {
    asm volatile("global_store_dword %0, %1, %2 offset:%3" : : "v"(voff), "v"(v), "s"(sbase), "i"(OFF) : "memory");
}

template <bool Y16>
__global__ __launch_bounds__(256, 1) void in_proj_rows128b_kernel(const Args p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nk = p.K / RK;                                          // k-steps of 32 (even, >= 8: entry point)
    const long ntiles = p.n_tiles;
    const uint32_t smem_lds = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(smem));
    const uint32_t region_lds = smem_lds + wave * WREGION;            // this wave's W' ring (+ x slot `wave` behind it)
    const uint32_t ring_lds = region_lds + lane * 16;
    const uint32_t wlane = lane * 16;
    const char* wsrc_w = p.Wf + (size_t)wave * RSLOT;                  // wave-uniform: + k-step * RW_TILE + fragment * 1024

    // x LDS-DMA: piece 4 wave + q of an x tile = rows 8 q .. 8 q + 7 of the wave's 32-row group (1 KiB, lane -> LDS chunk
    // 64 (4 wave + q) + lane); the lane fetches the 16-byte chunk (8 k) that belongs at that position of the swizzled image
    // (chunk c of row r sits at chunk c ^ ((r >> 1) & 7)).  Rows past M: the byte offset is clamped to the group's last valid chunk.
    uint32_t voffx[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int row = 8 * q + (lane >> 3);
        voffx[q] = (uint32_t)((long)row * p.K * 2 + (((lane & 7) ^ ((row >> 1) & 7)) << 4));
    }
    auto group_row = [&](long t, int i) -> long {                    // first row of group i of tile t, clamped into [0, M)
        const long g = p.grp != nullptr ? (long)p.grp[4 * t + i] : t * RM + 32 * i;
        return g < p.M ? g : p.M - 1;
    };
    auto tile_src = [&](long t) { return reinterpret_cast<const char*>(p.x + group_row(t, wave) * p.K); };
    auto tile_maxoff = [&](long t) {
        const long left = p.M - group_row(t, wave);
        const long rv = left < 32 ? left : 32;
        return (uint32_t)((rv - 1) * p.K * 2 + 112);
    };
    // A-fragment reads: lane (r = lane & 31, h = lane >> 5) takes chunk 4 half + 2 kk + h of row 32 i + r (half = which k-step of
    // the x tile)
    uint32_t va[2][2];
#pragma unroll
    for (int hf = 0; hf < 2; ++hf)
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const int r = lane & 31, c = 4 * hf + 2 * kk + (lane >> 5);
            va[hf][kk] = smem_lds + 2 * RSLOT + r * 128 + ((c ^ ((r >> 1) & 7)) << 4);   // + slot * WREGION + row tile * 4096
        }

    f32x16 acc[4][6];
    u32x4 a[2][4][2];        // A fragments: [k-step parity][row tile i][kk]; the other parity is being read

    // k-steps in rotated order, starting (workgroup % 8) eighths of the way through K (even: an x tile is two steps)
    const int kt0 = ((int)(blockIdx.x & 7) * nk / 8) & ~1;
    auto rot = [&](int k) { const int r = k + kt0; return r < nk ? r : r - nk; };

    // every wave numbers the row tiles from its own group (tile i' = rows 32 ((i' + wave) % 4) ..), as in K4: the epilogue code
    // is the same instruction stream in all four waves
    uint32_t roff[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) roff[i] = ((i + wave) & 3) * 4096;

    long tile = blockIdx.x;
    const char* xsrc = tile_src(tile);
    uint32_t xmax = tile_maxoff(tile);
    {
#pragma unroll
        for (int s = 0; s < 3; ++s) {                                 // x tiles 0, 1, 2 -> slots 0, 1, 2
            const char* src = xsrc + (size_t)rot(2 * s) * (RK * 2);
            const uint32_t dst = smem_lds + s * WREGION + 2 * RSLOT + (4 * wave) * 1024;
#pragma unroll
            for (int q = 0; q < 4; ++q) glds16_s<0>(voffx[q] < xmax ? voffx[q] : xmax, src, dst + q * 1024);
        }
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const char* src = wsrc_w + (size_t)rot(s) * RW_TILE;
            static_for<0, 12>([&](auto fc) {
                constexpr int f = decltype(fc)::value;
                glds16_s<(f & 3) * 1024>(wlane, src + (f >> 2) * 4096, region_lds + s * RSLOT + (f >> 2) * 4096);
            });
        }
        asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
        static_for<0, 8>([&](auto pc) {                               // k-step 0 of the first tile -> a[0]
            constexpr int pr = decltype(pc)::value, i = pr >> 1, kk = pr & 1;
            lds_read16<0>(a[0][i][kk], va[0][kk] + roff[i]);
        });
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    bf16x8 b[3];
    lds_read16<0>(b[0], ring_lds);                                    // B fragments 0, 1 of k-step 0
    lds_read16<1024>(b[1], ring_lds);
    int nflat = 0;                                                    // k-steps done by this workgroup, mod 8: the x ring phase (even)

    // One k-step (32 k).  PAR = kt & 1: the W' ring half, the A-fragment set in use and the half of the x tile.  12 groups:
    // B fragment g (column tile g / 2, kk = g & 1) x the 4 row tiles = 4 MFMAs; the gaps carry
    //   gap 1  groups 0-7: read A fragment (row tile g / 2, kk = g & 1) of the NEXT k-step into the other set;
    //          groups 8-11, even steps: one of the wave's 4 x pieces, three x tiles ahead, into the slot whose tile was last
    //          read two steps ago;
    //   gap 2  refill fragment g's ring slot for k-step kt + 2;
    //   gap 3  read B fragment g + 2 (groups 10, 11: fragments 0, 1 of the next step).
    // ONE hand-counted wait per group, at its top (both queues complete in order): kVm[PAR][g] = VMEM operations issued after the
    // refill of the ring slot that gap 3 reads (even steps issue 12 W' + 4 x operations, odd steps 12); kLg[g] = LDS operations
    // issued after the read of fragment g.
    const char* xsrc_n = xsrc;
    uint32_t xmax_n = xmax;
    auto step = [&](auto parc, int kt) {
        constexpr int PAR = decltype(parc)::value;
        constexpr int kVm[2][12] = {{25, 25, 25, 25, 25, 25, 24, 23, 22, 22, 23, 24}, {25, 25, 25, 25, 25, 25, 25, 25, 25, 25, 25, 25}};
        constexpr int kLg[12] = {1, 2, 2, 2, 2, 2, 2, 2, 2, 1, 1, 1};
        asm volatile("s_barrier" ::: "memory");   // all waves are done with the previous step's reads; every quarter of the tile read next has landed
        const int n = nflat + kt;                 // parity of n == PAR
        const int T = n >> 1;                     // x tile of this step (mod 4 = its slot)
        const uint32_t xslot = (uint32_t)((PAR ? T + 1 : T) & 3) * WREGION;      // slot of the NEXT step's half
        const uint32_t xa0 = va[PAR ^ 1][0] + xslot, xa1 = va[PAR ^ 1][1] + xslot;
        const uint32_t wdst = region_lds + PAR * RSLOT;
        const char* wnext = wsrc_w + (size_t)rot(kt + 2 < nk ? kt + 2 : kt + 2 - nk) * RW_TILE;
        const char* wn1 = wnext + 4096;
        const char* wn2 = wnext + 8192;
        // even steps: x tile kt / 2 + 3 of the flat sequence (the next row tile's once this one's run out) -> slot (T + 3) & 3
        const bool over = kt + 6 >= nk;
        const char* xnext = (over ? xsrc_n : xsrc) + (size_t)rot(over ? kt + 6 - nk : kt + 6) * (RK * 2);
        const uint32_t xm = over ? xmax_n : xmax;
        const uint32_t xdst = smem_lds + (uint32_t)((T + 3) & 3) * WREGION + 2 * RSLOT + (4 * wave) * 1024;
        static_for<0, 12>([&](auto gc) {
            constexpr int g = decltype(gc)::value;
            constexpr int j = g >> 1, kk = g & 1;
            asm volatile("s_waitcnt vmcnt(%1) lgkmcnt(%2)" : "+v"(b[g % 3]) : "n"(kVm[PAR][g]), "n"(kLg[g]) : "memory");
            if constexpr (j < 4) mfma_agpr(acc[0][j], a[PAR][0][kk], b[g % 3]); else mfma_vgpr(acc[0][j], a[PAR][0][kk], b[g % 3]);
            if constexpr (g >= 8) {
                if constexpr (PAR == 0) glds_m0(xdst + (g - 8) * 1024);
                else if constexpr (g == 8) glds_m0(wdst + 2 * 4096);
            }
            if constexpr (j < 4) mfma_agpr(acc[1][j], a[PAR][1][kk], b[g % 3]); else mfma_vgpr(acc[1][j], a[PAR][1][kk], b[g % 3]);
            if constexpr (g < 8) {                                   // gap 1
                if constexpr ((g & 3) == 0) glds_m0(wdst + (g >> 2) * 4096);
                lds_read16<0>(a[PAR ^ 1][g >> 1][kk], (kk ? xa1 : xa0) + roff[g >> 1]);
            } else if constexpr (PAR == 0) {
                glds16_m<0>(voffx[g - 8] < xm ? voffx[g - 8] : xm, xnext);
                glds_m0(wdst + 2 * 4096);
            }
            if constexpr (j < 4) mfma_agpr(acc[2][j], a[PAR][2][kk], b[g % 3]); else mfma_vgpr(acc[2][j], a[PAR][2][kk], b[g % 3]);
            glds16_m<(g & 3) * 1024>(wlane, (g >> 2) == 0 ? wnext : (g >> 2) == 1 ? wn1 : wn2);       // gap 2
            if constexpr (j < 4) mfma_agpr(acc[3][j], a[PAR][3][kk], b[g % 3]); else mfma_vgpr(acc[3][j], a[PAR][3][kk], b[g % 3]);
            if constexpr (g < 10) lds_read16<PAR * RSLOT + (g + 2) * 1024>(b[(g + 2) % 3], ring_lds);    // gap 3
            else lds_read16<(PAR ^ 1) * RSLOT + (g - 10) * 1024>(b[(g + 2) % 3], ring_lds);
        });
    };

    const bool relu = p.relu;
    const int hrow = 4 * (lane >> 5);
    uint32_t vo[4];                      // output byte offsets of the lane's 4 rows of an accumulator register group
#pragma unroll
    for (int e = 0; e < 4; ++e) vo[e] = (uint32_t)((hrow + e) * (kHidden * 4) + ((wave & 1) * RWC + (lane & 31)) * 4);
    const uint32_t coff = (wave * RWC + (lane & 31)) * 4;

    for (;;) {
        const long tnext = tile + gridDim.x;
        const bool more = tnext < ntiles;
        xsrc_n = more ? tile_src(tnext) : xsrc;
        xmax_n = more ? tile_maxoff(tnext) : xmax;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 6; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        for (int kt = 0; kt < nk; kt += 2) {
            step(std::integral_constant<int, 0>{}, kt);
            step(std::integral_constant<int, 1>{}, kt + 1);
        }
        nflat = (nflat + nk) & 7;

        // ---- tile boundary.  The rings hold (or are receiving) the next tile's first k-steps, its k-step 0 fragments are read.
        // x slot (nflat / 2 + 3) & 3 is free until group 8 of the next k-step: the row statistics reach the waves through it.
        float csn[6], bbn[6], mrow, rrow;
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            asm volatile("global_load_dword %0, %1, %2 offset:%3" : "=v"(csn[j]) : "v"(coff), "s"(p.cs), "i"(128 * j) : "memory");
            asm volatile("global_load_dword %0, %1, %2 offset:%3" : "=v"(bbn[j]) : "v"(coff), "s"(p.bb), "i"(128 * j) : "memory");
        }
        {
            long srow = group_row(tile, wave) + (lane & 31);         // the wave's own group: rows 32 wave .. of the tile
            srow = srow < p.M ? srow : p.M - 1;
            mrow = p.mean[srow];
            rrow = p.rstd[srow];
        }
        float* s_mean = reinterpret_cast<float*>(smem + (((nflat >> 1) + 3) & 3) * WREGION + 2 * RSLOT);
        float* s_rstd = s_mean + RM;
        asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier"
                     : "+v"(csn[0]), "+v"(csn[1]), "+v"(csn[2]), "+v"(csn[3]), "+v"(csn[4]), "+v"(csn[5]), "+v"(bbn[0]), "+v"(bbn[1]),
                       "+v"(bbn[2]), "+v"(bbn[3]), "+v"(bbn[4]), "+v"(bbn[5]), "+v"(mrow), "+v"(rrow)
                     :
                     : "memory");
        if (lane < 32) {
            s_mean[32 * wave + lane] = mrow;
            s_rstd[32 * wave + lane] = rrow;
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier\n\ts_nop 15\n\ts_nop 15" ::: "memory");

        // epilogue (K4's): wave w writes columns [192 w, 192 w + 192) = branch w / 2, columns (w & 1) * 192 .. straight from the
        // accumulators: a register holds one column of 2 x 4 rows; 32 lanes = one 128-byte row segment per store.
        long grow[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) grow[i] = group_row(tile, i);
        const bool full = grow[0] + 32 <= p.M && grow[1] + 32 <= p.M && grow[2] + 32 <= p.M && grow[3] + 32 <= p.M;
        const char* ybranch = reinterpret_cast<const char*>((wave >> 1) ? p.y[1] : p.y[0]);
        auto epilogue = [&](auto fullc) {
            constexpr bool FULL = decltype(fullc)::value;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int ti = (i + wave) & 3;                       // the row tile behind the wave's accumulators acc[i][..]
                const long m0 = ti == 0 ? grow[0] : ti == 1 ? grow[1] : ti == 2 ? grow[2] : grow[3];
                const char* yt = ybranch + (size_t)m0 * (kHidden * 4);
                static_for<0, 4>([&](auto gc) {
                    constexpr int g = decltype(gc)::value;
                    const char* yg = yt + g * 8 * (kHidden * 4);
                    const f32x4 mean4 = *reinterpret_cast<const f32x4*>(s_mean + 32 * ti + 8 * g + hrow);
                    const f32x4 rstd4 = *reinterpret_cast<const f32x4*>(s_rstd + 32 * ti + 8 * g + hrow);
                    static_for<0, 4>([&](auto ec) {
                        constexpr int e = decltype(ec)::value;
                        float v[6];
#pragma unroll
                        for (int j = 0; j < 6; ++j) {
                            float t;
                            if (j < 4) asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(t) : "a"(acc[i][j][4 * g + e]));
                            else t = acc[i][j][4 * g + e];
                            v[j] = rstd4[e] * (t - mean4[e] * csn[j]) + bbn[j];
                            if (relu) v[j] = fmaxf(v[j], 0.f);
                        }
                        if (FULL || m0 + 8 * g + hrow + e < p.M) {
                            gstore32<0>(vo[e], v[0], yg);
                            gstore32<128>(vo[e], v[1], yg);
                            gstore32<256>(vo[e], v[2], yg);
                            gstore32<384>(vo[e], v[3], yg);
                            gstore32<512>(vo[e], v[4], yg);
                            gstore32<640>(vo[e], v[5], yg);
                        }
                    });
                });
            }
        };
        // the same as bf16 rows: a lane holds ONE column of 4 rows; lanes 2 c', 2 c' + 1 trade a value (DPP quad_perm [1,0,3,2]) so
        // that the even lane owns columns (c, c + 1) of row e and the odd lane the same columns of row e + 1: one
        // v_cvt_pk_bf16_f32 and one dword store each - 16 lanes = a 64-byte row segment, half the store instructions of the fp32 form
        auto epilogue16 = [&](auto fullc) {
            constexpr bool FULL = decltype(fullc)::value;
            const bool odd = lane & 1;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int ti = (i + wave) & 3;
                const long m0 = ti == 0 ? grow[0] : ti == 1 ? grow[1] : ti == 2 ? grow[2] : grow[3];
                const char* yt = ybranch + (size_t)m0 * (kHidden * 2);
                static_for<0, 4>([&](auto gc) {
                    constexpr int g = decltype(gc)::value;
                    const char* yg = yt + g * 8 * (kHidden * 2);
                    const f32x4 mean4 = *reinterpret_cast<const f32x4*>(s_mean + 32 * ti + 8 * g + hrow);
                    const f32x4 rstd4 = *reinterpret_cast<const f32x4*>(s_rstd + 32 * ti + 8 * g + hrow);
                    static_for<0, 2>([&](auto pc) {
                        constexpr int e0 = 2 * decltype(pc)::value;
                        // (rows e0 + lane parity, the lane pair's two columns)
                        const uint32_t vo16 = (uint32_t)((hrow + e0 + (lane & 1)) * (kHidden * 2) + ((wave & 1) * RWC + (lane & 30)) * 2);
                        const bool inside = FULL || m0 + 8 * g + hrow + e0 + (odd ? 1 : 0) < p.M;
                        static_for<0, 6>([&](auto jc) {
                            constexpr int j = decltype(jc)::value;
                            float t0, t1;
                            if constexpr (j < 4) {
                                asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(t0) : "a"(acc[i][j][4 * g + e0]));
                                asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(t1) : "a"(acc[i][j][4 * g + e0 + 1]));
                            } else {
                                t0 = acc[i][j][4 * g + e0];
                                t1 = acc[i][j][4 * g + e0 + 1];
                            }
                            float v0 = rstd4[e0] * (t0 - mean4[e0] * csn[j]) + bbn[j];
                            float v1 = rstd4[e0 + 1] * (t1 - mean4[e0 + 1] * csn[j]) + bbn[j];
                            if (relu) { v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); }
                            const float give = odd ? v0 : v1;                       // what the partner lane needs
                            const float got = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, give), 0xB1, 0xF, 0xF, true));
                            const unsigned lo = f32_to_h16_bits(odd ? got : v0), hi = f32_to_h16_bits(odd ? v1 : got);
                            if (inside) gstore32<64 * j>(vo16, __builtin_bit_cast(float, lo | (hi << 16)), yg);
                        });
                    });
                });
            }
        };
        if constexpr (Y16) {
            if (full) epilogue16(std::true_type{});
            else epilogue16(std::false_type{});
        } else {
            if (full) epilogue(std::true_type{});
            else epilogue(std::false_type{});
        }
        if (!more) break;
        tile = tnext;
        xsrc = xsrc_n;
        xmax = xmax_n;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                  // LDS-DMAs of the look-ahead must not outlive the workgroup
}

}  // namespace k4b
}  // namespace dldkd

using namespace dldkd;

extern "C" int dldkd_in_proj_h16_rows128b_ok(int K) { return K >= 8 * k4b::RK && K % (2 * k4b::RK) == 0 && (long)127 * K * 2 + 128 <= 0xFFFFFFFFL; }

static int rows128b_launch(const void* x_bf16, const float* mean, const float* rstd, const void* Wfrag, const float* cs,
                           const float* bb, void* y0, void* y1, int y16, long M, int K, int relu, const int32_t* groups,
                           long n_groups, void* stream);

extern "C" int dldkd_in_proj_h16_rows128b(const void* x_bf16, const float* mean, const float* rstd, const void* Wfrag, const float* cs,
                                           const float* bb, float* y0, float* y1, long M, int K, int relu, const int32_t* groups,
                                           long n_groups, void* stream) {
    return rows128b_launch(x_bf16, mean, rstd, Wfrag, cs, bb, y0, y1, 0, M, K, relu, groups, n_groups, stream);
}

extern "C" int dldkd_in_proj_h16_rows128b_out16(const void* x_bf16, const float* mean, const float* rstd, const void* Wfrag, const float* cs,
                                                 const float* bb, void* y0_bf16, void* y1_bf16, long M, int K, int relu,
                                                 const int32_t* groups, long n_groups, void* stream) {
    return rows128b_launch(x_bf16, mean, rstd, Wfrag, cs, bb, y0_bf16, y1_bf16, 1, M, K, relu, groups, n_groups, stream);
}

static int rows128b_launch(const void* x_bf16, const float* mean, const float* rstd, const void* Wfrag, const float* cs,
                           const float* bb, void* y0, void* y1, int y16, long M, int K, int relu, const int32_t* groups,
                           long n_groups, void* stream) {
    if (M < 0 || !dldkd_in_proj_h16_rows128b_ok(K)) {
        set_error("in_proj_bf16_rows128b: K must be a multiple of %d, at least %d (M=%ld K=%d)", 2 * k4b::RK, 8 * k4b::RK, M, K);
        return DLDKD_EINVAL;
    }
    if (M == 0) return DLDKD_OK;
    if (!x_bf16 || !mean || !rstd || !Wfrag || !cs || !bb || !y0 || !y1) { set_error("in_proj_bf16_rows128b: null pointer"); return DLDKD_EINVAL; }
    if (((uintptr_t)x_bf16 | (uintptr_t)y0 | (uintptr_t)y1 | (uintptr_t)Wfrag) & 15) { set_error("in_proj_bf16_rows128b: unaligned buffer"); return DLDKD_EINVAL; }
    if (groups != nullptr && (n_groups < 0 || (n_groups & 3))) { set_error("in_proj_bf16_rows128b: the group table must hold a multiple of 4 groups"); return DLDKD_EINVAL; }
    if (groups != nullptr && n_groups == 0) return DLDKD_OK;
    const long ntiles = groups != nullptr ? n_groups / 4 : (M + k4b::RM - 1) / k4b::RM;
    k4b::Args p{(const unsigned short*)x_bf16, mean, rstd, (const char*)Wfrag, cs, bb, {(float*)y0, (float*)y1}, y16, M, K, relu != 0, groups, ntiles};
    constexpr int lds = 4 * k4b::WREGION;       // all 160 KiB
    static int n_cu = 0;                        // one persistent workgroup per CU
    if (!n_cu) {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
        n_cu = v;
    }
    const dim3 grid((unsigned)(ntiles < n_cu ? ntiles : n_cu));
    static const bool ok = hipFuncSetAttribute((const void*)k4b::in_proj_rows128b_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds) == hipSuccess &&
                           hipFuncSetAttribute((const void*)k4b::in_proj_rows128b_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds) == hipSuccess;
    (void)ok;
    if (y16) DLDKD_LAUNCH(k4b::in_proj_rows128b_kernel<true>, grid, dim3(256), lds, (hipStream_t)stream, p);
    else DLDKD_LAUNCH(k4b::in_proj_rows128b_kernel<false>, grid, dim3(256), lds, (hipStream_t)stream, p);
    return check_launch("in_proj_bf16_rows128b");
}
