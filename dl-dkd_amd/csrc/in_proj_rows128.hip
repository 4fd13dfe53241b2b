// NOTE (round 5): the 16-bit MFMA operands of this file are h16 = IEEE fp16, not bf16 (common.hpp says why; the text below and the
// identifiers still say "bf16" where they mean "the 16-bit operand": bf16x8 is the 8 x 16-bit container, whatever the format).
// K4, second generation: y = ReLU( LayerNorm(x) . W^T + b ) for the raw clip / word features, both branches (768 output
// columns) in ONE pass over the fp32 rows.  Replaces LinearLayer.forward (reference method/model_components.py:305-312) on the
// inference path, like rows_linear_bf16_kernel<1, true> (in_proj_h16.hip), whose limits this kernel is built around
// (profiles/r01/ablation_k4_in_proj.md): 0.58 LDS fragment reads per MFMA and 8 waves in lock-step around a barrier with one
// k-tile of loads in flight.  The kernel needs ~1.1 PFLOP/s to move 3.6 TB/s (307 flop per byte), i.e. the MFMA pipe busy
// ~45 % of the time with ONE wave per SIMD - so every load has to be in flight long before it is needed, and nothing that is
// in flight may cost registers:
//
//   * 4 waves per workgroup, one per SIMD.  Wave w owns ALL 128 rows of the workgroup x columns [192 w, 192 w + 192):
//     4 x 6 tiles of mfma_f32_32x32x16_bf16 = 384 accumulator registers - more than the 256 AGPRs, and hipcc allocates every
//     MFMA of a kernel in one register class.  So the MFMAs are inline asm: column tiles 0-3 accumulate in AGPRs ("+a"),
//     column tiles 4-5 in arch VGPRs ("+v"): 256 + 128, no copies.  A fragment feeds 6 MFMAs, a B fragment 4.
//   * x streams HBM -> LDS by LDS-DMA (global_load_lds_dwordx4) as fp32 into a 4-slot ring of 16-KiB k-tiles (128 rows x 32
//     floats) shared by the four waves (each issues a quarter): three k-steps (48 KiB per CU) in flight, costing no registers.
//     The DMA's per-lane SOURCE addresses are permuted so that the image is XOR-swizzled (16-B chunk c of row r sits at chunk
//     c ^ ((r >> 1) & 7)): the A-fragment reads (ds_read_b128, lane = row) are conflict-free.  Each wave reads its fragments as
//     fp32 one k-step ahead, converts with v_cvt_pk_bf16_f32 and keeps them for the step's 6 column tiles.
//     (The first version of this kernel loaded x straight into registers, one k-step ahead: 1775 GB/s, every step waited ~2 us
//     for HBM.)
//   * W' (LayerNorm-folded weights, bf16, MFMA B-fragment order) streams L2 -> LDS by LDS-DMA into a PRIVATE ring per wave of
//     24 1-KiB fragments = two k-steps of the wave's 6 column tiles.  A fragment's slot is refilled (two k-steps ahead) as soon
//     as its ds_read has returned, so the DMA issue is spread over the step and each fragment has two full steps to land.
//   * one s_barrier per k-step (x slot hand-over); all other waits are hand-counted: the VMEM queue of a wave holds, in issue
//     order, 4 x pieces + 12 W' fragments per step, so "fragment f of this step has landed" is a fixed s_waitcnt vmcnt(26..29)
//     (table kVm at the k-step).  Loads and waits are asm the compiler cannot see through, so it does not drain the queue around
//     the LDS-DMA (cdna_hip_programming.md section 5, trap (b)); every register an asm load fills is named by the asm wait that
//     covers it, or plain C++ readers could be scheduled above the wait.
//   * LayerNorm is folded as before: out = rstd (x.W'^T - mean colsum(W')) + (W.beta + b); wave w accumulates sum / sum of
//     squares of rows 32 w .. 32 w + 31 from the fp32 values it converts anyway.  No branch in the k-loop: every wave numbers
//     the row tiles from its own, so "the wave's own tile" is tile 0 of the same instruction stream in all four waves.
//   * the k-tiles are visited in an order rotated per XCD (workgroup b starts (b % 8) / 8 of the way through K): with a row
//     stride of 3 x 4 KiB, workgroups in the same k phase kept hitting the same memory channels.
//   * one persistent workgroup per CU: the LDS-DMA streams run through the tile boundary (the x look-ahead moves on to the next
//     tile's rows), the row statistics cross the waves through the x slot that is free for one barrier, and the epilogue stores
//     the accumulators straight to global memory (128-byte row segments per 32 lanes) while the next tile's k-tiles land.
#include <type_traits>

#include "common.hpp"

namespace dldkd {

constexpr int RM = 128, RK = 32, RWC = 192;
constexpr int RSLOT = 6 * 2 * 1024;       // bytes of W' per wave per k-step: [6 column tiles][2 kk][64 lanes][16 B]
constexpr int XSLOT = RM * RK * 4;        // one k-tile of x as fp32: 16 KiB
constexpr int WREGION = 2 * RSLOT + XSLOT;  // LDS per wave: [W' ring, 2 k-steps][x ring slot number `wave`] = 40 KiB
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
    unsigned long long* stamps;   // diagnostics only (dldkd_debug_in_proj_rows128_timeline): 9 words per workgroup, else null
    const int32_t* grp;    // row-group table or null.  A 128-row tile is four GROUPS of 32 consecutive rows; with a table, group i of
                           // tile t starts at row grp[4 t + i] (of x AND of y): only the groups that hold valid clips of a padded
                           // (n, L, K) batch are visited.  null: group i of tile t = rows 128 t + 32 i.
    long n_tiles;          // ceil(#groups / 4)
};

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
// 8 fp32 -> one h16 (fp16: common.hpp) MFMA fragment: 4 v_cvt_pk_f16_f32, pinned where they are written (between two MFMAs)
__device__ __forceinline__ void cvt8(u32x4& dst, const f32x4& lo, const f32x4& hi) {
    unsigned u0, u1, u2, u3;
    asm volatile(DLDKD_H16_CVT_PK " %0, %1, %2" : "=v"(u0) : "v"(lo[0]), "v"(lo[1]));
    asm volatile(DLDKD_H16_CVT_PK " %0, %1, %2" : "=v"(u1) : "v"(lo[2]), "v"(lo[3]));
    asm volatile(DLDKD_H16_CVT_PK " %0, %1, %2" : "=v"(u2) : "v"(hi[0]), "v"(hi[1]));
    asm volatile(DLDKD_H16_CVT_PK " %0, %1, %2" : "=v"(u3) : "v"(hi[2]), "v"(hi[3]));
    dst = u32x4{u0, u1, u2, u3};
}

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
// LDS-DMA, 16 B per lane: global address = scalar base + 32-bit lane offset + immediate (no 64-bit VALU add in the MFMA shadow),
// LDS destination = wave-uniform lds_base (-> M0) + the SAME immediate + lane * 16.  hipcc treats M0 as reserved and re-materialises it before each
// of its own uses, so the asm may overwrite it.
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

template <bool STAMP>
__global__ __launch_bounds__(256, 1) void in_proj_rows128_kernel(const Rows128Args p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nk = p.K / RK;
    const long ntiles = p.n_tiles;
    const uint32_t smem_lds = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(smem));
    const uint32_t region_lds = smem_lds + wave * WREGION;            // this wave's W' ring (+ x slot `wave` behind it)
    const uint32_t ring_lds = region_lds + lane * 16;
    const uint32_t wlane = lane * 16;
    const char* wsrc_w = p.Wf + (size_t)wave * RSLOT;                  // wave-uniform: + k-step * RW_TILE + fragment * 1024

    // x LDS-DMA: piece t = 4 wave + q of a k-tile = rows 8 t .. 8 t + 7 (1 KiB, lane -> LDS chunk 64 t + lane).  The lane
    // fetches the global chunk that belongs at that position of the swizzled image.  In a ragged last tile the byte offset is
    // clamped to the tile's last valid chunk (rows past M feed accumulator rows that are never stored).
    // (a wave's four pieces are rows 32 wave .. 32 wave + 31 of the tile = ONE row group: its global source is the group's own
    // base, so a tile can be made of any four 32-row groups; offsets are relative to the group)
    uint32_t voffx[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int row = 8 * q + (lane >> 3);                         // row inside the wave's group
        voffx[q] = (uint32_t)((long)row * p.K * 4 + (((lane & 7) ^ ((row >> 1) & 7)) << 4));
    }
    auto group_row = [&](long t, int i) -> long {                    // first row of group i of tile t, clamped into [0, M)
        const long g = p.grp != nullptr ? (long)p.grp[4 * t + i] : t * RM + 32 * i;
        return g < p.M ? g : p.M - 1;
    };
    auto tile_src = [&](long t) { return reinterpret_cast<const char*>(p.x + group_row(t, wave) * p.K); };
    auto tile_maxoff = [&](long t) {
        const long left = p.M - group_row(t, wave);
        const long rv = left < 32 ? left : 32;
        return (uint32_t)((rv - 1) * p.K * 4 + 112);
    };
    // A-fragment reads: lane (r = lane & 31, h = lane >> 5) takes chunks 4 kk + 2 h + e of row 32 i + r
    uint32_t va[2][2];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int r = lane & 31, c = 4 * kk + 2 * (lane >> 5) + e;
            va[kk][e] = smem_lds + 2 * RSLOT + r * 128 + ((c ^ ((r >> 1) & 7)) << 4);   // + slot * WREGION + row tile * 4096
        }

    unsigned long long ts[10];
    if constexpr (STAMP) { ts[0] = __builtin_amdgcn_s_memtime(); ts[1] = __builtin_amdgcn_s_memrealtime(); }
    f32x16 acc[4][6];
    u32x4 a[2][4][2];        // A fragments (bf16 pairs): [k-step parity][row tile i][kk]; the other parity is being filled
    f32x4 tmp[2][2];         // fp32 of one (row tile, kk) pair of the NEXT k-step on its way from LDS to bf16, double-buffered
    f32x2 sum2 = {0.f, 0.f}, sq2 = {0.f, 0.f};      // LayerNorm sums of the k-tiles converted so far (wave's own row tile)
    f32x2 fsum2 = {0.f, 0.f}, fsq2 = {0.f, 0.f};    // ... of the tile being finished (taken before its last k-step, see below)

    // k-steps are taken in rotated order, starting at kt0 = (workgroup % 8) eighths of the way: at K = 3072 the row stride is
    // 3 x 4 KiB, and workgroups marching through the same k-tile at the same time kept hitting the same few HBM channels
    // (12 % of the kernel's time).  Consecutive workgroups go to different XCDs, so the CUs that share an L2 still share W'.
    const int kt0 = (int)(blockIdx.x & 7) * nk / 8;
    auto rot = [&](int k) { const int r = k + kt0; return r < nk ? r : r - nk; };
    auto stats = [&](const f32x4& lo, const f32x4& hi) {
        const f32x2 v0 = {lo[0], lo[1]}, v1 = {lo[2], lo[3]}, v2 = {hi[0], hi[1]}, v3 = {hi[2], hi[3]};
        sum2 += (v0 + v1) + (v2 + v3);
        sq2 += v0 * v0 + v1 * v1 + v2 * v2 + v3 * v3;
    };

    // Every wave numbers the row tiles from its own: its tile i' is rows 32 ((i' + wave) % 4) ..  The wave's own tile (whose
    // LayerNorm sums it accumulates) is then i' = 0 in every wave: the same instruction stream for all four, no branch in the loop
    // (four specialised copies of the loop made hipcc spill).  The price is one scalar-operand add per A-fragment read.
    uint32_t roff[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) roff[i] = ((i + wave) & 3) * 4096;

    // The workgroup is persistent: tiles blockIdx.x, + gridDim.x, ..  The two LDS-DMA streams never drain between tiles: the
    // k-tile sequence is flat across them (x: four k-tiles ahead, from the NEXT tile's rows once the current tile's run out;
    // W': two ahead, wrapping around).  The prologue below happens once per workgroup.
    long tile = blockIdx.x;
    const char* xsrc = tile_src(tile);
    uint32_t xmax = tile_maxoff(tile);
    {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const char* src = xsrc + (size_t)rot(s) * (RK * 4);
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
        static_for<0, 8>([&](auto pc) {                               // k-tile 0 of the first tile -> a[0], unpipelined
            constexpr int pr = decltype(pc)::value, i = pr >> 1, kk = pr & 1;
            lds_read16<0>(tmp[0][0], va[kk][0] + roff[i]);
            lds_read16<0>(tmp[0][1], va[kk][1] + roff[i]);
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(tmp[0][0]), "+v"(tmp[0][1]) : : "memory");
            cvt8(a[0][i][kk], tmp[0][0], tmp[0][1]);
            if constexpr (i == 0) stats(tmp[0][0], tmp[0][1]);
        });
    }
    uint32_t x00 = va[0][0], x01 = va[0][1], x10 = va[1][0], x11 = va[1][1];    // slot 0; every k-step advances them first
    bf16x8 b[3];
    lds_read16<0>(b[0], ring_lds);                                    // B fragments 0, 1 of k-step 0
    lds_read16<1024>(b[1], ring_lds);
    int nflat = 0;                                                    // k-steps done by this workgroup, mod 4: the x ring phase

    // One k-step.  PAR = kt & 1: the W' ring half and the A-fragment set in use.  12 groups: B fragment g (column tile g / 2,
    // kk = g & 1) x the 4 row tiles = 4 MFMAs.  One wave per SIMD: whatever else the wave issues has to fit the MFMA shadow
    // (an MFMA keeps the issue port 8 of its 32 cycles; every instruction after it costs 4-16 more; measured: the un-hidden
    // rest cost a third of the loop), branches cost a refetch, and nothing may be waited for less than ~250 cycles after it was
    // issued.  So there is no branch in the loop and the four gaps of a group carry one job each:
    //   gap 0  groups 2-9: convert the pair read two groups earlier into the other A-fragment set (+ the LayerNorm sums for the
    //          wave's own row tile: groups 2, 3);
    //   gap 1  groups 0-7: read pair g = (row tile g / 2, kk = g & 1) of the NEXT k-tile as fp32 (2-deep ring);
    //          groups 8-11: one of the wave's 4 x LDS-DMA pieces, four k-tiles ahead;
    //   gap 2  refill fragment g's ring slot for k-step kt + 2 (fragment g is in registers since the top of the group);
    //   gap 3  read B fragment g + 2 (groups 10, 11: fragments 0, 1 of the next step) into the 3-deep register ring.
    // ONE hand-counted wait per group, at its top (both queues complete in order): kVm[g] = VMEM operations issued after the
    // ring slot that gap 3 will read was refilled; kLgTop[g] = LDS operations issued after fragment g (the pair for gap 0 is
    // older).  Operations the count does not know (the epilogue's stores and loads) only make a wait longer, never too short:
    // loads complete in order.  M0 (LDS-DMA destination) is written a gap ahead of its use instead of padding with s_nop.
    const char* xsrc_n = xsrc;           // rows of this workgroup's next tile (its own again when there is none: harmless re-reads)
    uint32_t xmax_n = xmax;
    auto step = [&](auto parc, int kt) {
        constexpr int PAR = decltype(parc)::value;
        constexpr int kVm[12] = {29, 29, 29, 29, 29, 29, 28, 27, 26, 26, 27, 28};
        constexpr int kLgTop[12] = {1, 3, 3, 3, 3, 3, 3, 3, 3, 1, 1, 1};
        asm volatile("s_barrier" ::: "memory");   // all waves are done with x slot n & 3; every quarter of the next k-tile has landed
        const int n = nflat + kt;
        {   // A-fragment read addresses move on to ring slot (n + 1) & 3
            const int adv = ((n + 1) & 3) ? WREGION : -3 * WREGION;
            x00 += adv, x01 += adv, x10 += adv, x11 += adv;
        }
        const uint32_t wdst = region_lds + PAR * RSLOT;
        const char* wnext = wsrc_w + (size_t)rot(kt + 2 < nk ? kt + 2 : kt + 2 - nk) * RW_TILE;
        const char* wn1 = wnext + 4096;
        const char* wn2 = wnext + 8192;
        const bool over = kt + 4 >= nk;
        const char* xnext = (over ? xsrc_n : xsrc) + (size_t)rot(over ? kt + 4 - nk : kt + 4) * (RK * 4);
        const uint32_t xm = over ? xmax_n : xmax;
        const uint32_t xdst = smem_lds + (n & 3) * WREGION + 2 * RSLOT + (4 * wave) * 1024;
        static_for<0, 12>([&](auto gc) {
            constexpr int g = decltype(gc)::value;
            constexpr int j = g >> 1, kk = g & 1;
            // (the pair converted in gap 0 is named too: its plain-C++ readers - the LayerNorm sums - must not be scheduled above the wait)
            asm volatile("s_waitcnt vmcnt(%3) lgkmcnt(%4)"
                         : "+v"(b[g % 3]), "+v"(tmp[g & 1][0]), "+v"(tmp[g & 1][1])
                         : "n"(kVm[g]), "n"(kLgTop[g])
                         : "memory");
            if constexpr (j < 4) mfma_agpr(acc[0][j], a[PAR][0][kk], b[g % 3]); else mfma_vgpr(acc[0][j], a[PAR][0][kk], b[g % 3]);
            if constexpr (g >= 2 && g <= 9) {                        // gap 0
                constexpr int pr = g - 2, pi = pr >> 1, pk = pr & 1;
                cvt8(a[PAR ^ 1][pi][pk], tmp[pr & 1][0], tmp[pr & 1][1]);
                if constexpr (pi == 0) stats(tmp[pr & 1][0], tmp[pr & 1][1]);
            }
            if constexpr (g >= 8) glds_m0(xdst + (g - 8) * 1024);
            if constexpr (j < 4) mfma_agpr(acc[1][j], a[PAR][1][kk], b[g % 3]); else mfma_vgpr(acc[1][j], a[PAR][1][kk], b[g % 3]);
            if constexpr (g < 8) {                                   // gap 1
                if constexpr ((g & 3) == 0) glds_m0(wdst + (g >> 2) * 4096);
                lds_read16<0>(tmp[g & 1][0], (kk ? x10 : x00) + roff[g >> 1]);
                lds_read16<0>(tmp[g & 1][1], (kk ? x11 : x01) + roff[g >> 1]);
            } else {
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
    bool first = true;

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
        if constexpr (STAMP) if (first) { ts[2] = __builtin_amdgcn_s_memtime(); ts[3] = __builtin_amdgcn_s_memrealtime(); }
        for (int kt = 0; kt < nk; kt += 2) {      // nk is even (entry point)
            step(std::integral_constant<int, 0>{}, kt);
            // The conversions of the tile's LAST k-step belong to the next tile (its k-tile 0): the sums of this tile are taken
            // before it.  (Its own k-tile 0 was converted by the previous tile's last step, or by the prologue.)
            const bool last = kt + 2 >= nk;
            fsum2 = last ? sum2 : fsum2, fsq2 = last ? sq2 : fsq2;
            sum2 = last ? f32x2{0.f, 0.f} : sum2, sq2 = last ? f32x2{0.f, 0.f} : sq2;
            step(std::integral_constant<int, 1>{}, kt + 1);
        }
        nflat = (nflat + nk) & 3;
        if constexpr (STAMP) if (first) { ts[4] = __builtin_amdgcn_s_memtime(); ts[5] = __builtin_amdgcn_s_memrealtime(); }

        // ---- tile boundary.  The rings already hold (or are receiving) the next tile's first k-tiles; the A fragments of its
        // k-step 0 are converted.  x slot nflat & 3 is free from the barrier below until group 8 of the next k-step: the row
        // statistics cross the waves through it.
        float csn[6], bbn[6];
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            asm volatile("global_load_dword %0, %1, %2 offset:%3" : "=v"(csn[j]) : "v"(coff), "s"(p.cs), "i"(128 * j) : "memory");
            asm volatile("global_load_dword %0, %1, %2 offset:%3" : "=v"(bbn[j]) : "v"(coff), "s"(p.bb), "i"(128 * j) : "memory");
        }
        float* s_mean = reinterpret_cast<float*>(smem + (nflat & 3) * WREGION + 2 * RSLOT);
        float* s_rstd = s_mean + RM;
        float sum = fsum2[0] + fsum2[1], sq = fsq2[0] + fsq2[1];
        sum += __shfl_xor(sum, 32);       // the two lane halves hold the two 8-float chunks of every 16 k
        sq += __shfl_xor(sq, 32);
        asm volatile("s_barrier" ::: "memory");                       // every wave has read the slot's k-tile
        if (lane < 32) {
            const float mean = sum / p.K;
            const float var = fmaxf(sq / p.K - mean * mean, 0.f);
            s_mean[32 * wave + lane] = mean;
            s_rstd[32 * wave + lane] = rsqrtf(var + p.eps);
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier\n\ts_nop 15\n\ts_nop 15"
                     : "+v"(csn[0]), "+v"(csn[1]), "+v"(csn[2]), "+v"(csn[3]), "+v"(csn[4]), "+v"(csn[5]), "+v"(bbn[0]), "+v"(bbn[1]),
                       "+v"(bbn[2]), "+v"(bbn[3]), "+v"(bbn[4]), "+v"(bbn[5])
                     :
                     : "memory");

        // epilogue: wave w writes columns [192 w, 192 w + 192) = branch w / 2, columns (w & 1) * 192 .. straight from the
        // accumulators: a register holds one column of 2 x 4 rows; 32 lanes = one 128-byte row segment per store.  No LDS (it is
        // full of the next tile), no wait inside (one wave per SIMD: nothing would hide it).
        long grow[4];                                                // first output row of the tile's four groups
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
                    // rows 32 ti + 8 g + hrow + {0..3} = accumulator registers 4 g .. 4 g + 3 (all lanes of a half read one address)
                    const f32x4 mean4 = *reinterpret_cast<const f32x4*>(s_mean + 32 * ti + 8 * g + hrow);
                    const f32x4 rstd4 = *reinterpret_cast<const f32x4*>(s_rstd + 32 * ti + 8 * g + hrow);
                    static_for<0, 4>([&](auto ec) {
                        constexpr int e = decltype(ec)::value;
                        float v[6];
#pragma unroll
                        for (int j = 0; j < 6; ++j) {
                            float t;     // AGPR accumulators are read where they are used (left to itself hipcc copied 238 of them ahead of the loop and spilled)
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
        if (full) epilogue(std::true_type{});
        else epilogue(std::false_type{});
        if constexpr (STAMP) if (first) { ts[6] = __builtin_amdgcn_s_memtime(); ts[7] = __builtin_amdgcn_s_memrealtime(); }
        first = false;
        if (!more) break;
        tile = tnext;
        xsrc = xsrc_n;
        xmax = xmax_n;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                  // LDS-DMAs of the look-ahead must not outlive the workgroup
    if constexpr (STAMP) {      // s_memtime (shader clock) / s_memrealtime (100 MHz): start, first loop start / end, first epilogue end, end
        ts[8] = __builtin_amdgcn_s_memtime(); ts[9] = __builtin_amdgcn_s_memrealtime();
        if (tid == 0) {
            unsigned long long* o = p.stamps + (size_t)blockIdx.x * 12;
            for (int i = 0; i < 10; ++i) o[i] = ts[i];
            unsigned xcc;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
            o[10] = xcc & 0xf;
            o[11] = (ntiles - blockIdx.x + gridDim.x - 1) / gridDim.x;
        }
    }
}

}  // namespace dldkd

using namespace dldkd;

static int launch_rows128(const float* x, const void* Wfrag, const float* cs, const float* bb, float* y0, float* y1, long M, int K,
                          float eps, int relu, unsigned long long* stamps, void* stream, const int32_t* groups = nullptr, long n_groups = 0) {
    if (M < 0 || K < 4 * RK || (K % (2 * RK)) || (long)127 * K * 4 + 128 > 0xFFFFFFFFL) {
        set_error("in_proj_bf16_rows128: K must be a multiple of %d, at least %d (M=%ld K=%d)", 2 * RK, 4 * RK, M, K);
        return DLDKD_EINVAL;
    }
    if (M == 0) return DLDKD_OK;
    if (!x || !Wfrag || !cs || !bb || !y0 || !y1) { set_error("in_proj_bf16_rows128: null pointer"); return DLDKD_EINVAL; }
    if (((uintptr_t)x | (uintptr_t)y0 | (uintptr_t)y1 | (uintptr_t)Wfrag) & 15) { set_error("in_proj_bf16_rows128: unaligned buffer"); return DLDKD_EINVAL; }
    if (groups != nullptr && (n_groups < 0 || (n_groups & 3))) { set_error("in_proj_bf16_rows128: the group table must hold a multiple of 4 groups"); return DLDKD_EINVAL; }
    if (groups != nullptr && n_groups == 0) return DLDKD_OK;
    const long ntiles = groups != nullptr ? n_groups / 4 : (M + RM - 1) / RM;
    Rows128Args p{x, (const char*)Wfrag, cs, bb, {y0, y1}, M, K, eps, relu != 0, stamps, groups, ntiles};
    constexpr int lds = 4 * WREGION;            // all 160 KiB
    static int n_cu = 0;                        // one persistent workgroup per CU (all 160 KiB of LDS)
    if (!n_cu) {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
        n_cu = v;
    }
    const dim3 grid((unsigned)(ntiles < n_cu ? ntiles : n_cu));
    if (stamps) {
        static const bool ok = hipFuncSetAttribute((const void*)in_proj_rows128_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds) == hipSuccess;
        (void)ok;
        DLDKD_LAUNCH(in_proj_rows128_kernel<true>, grid, dim3(256), lds, (hipStream_t)stream, p);
    } else {
        static const bool ok = hipFuncSetAttribute((const void*)in_proj_rows128_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds) == hipSuccess;
        (void)ok;
        DLDKD_LAUNCH(in_proj_rows128_kernel<false>, grid, dim3(256), lds, (hipStream_t)stream, p);
    }
    return check_launch("in_proj_bf16_rows128");
}

extern "C" int dldkd_in_proj_h16_rows128(const float* x, const void* Wfrag, const float* cs, const float* bb, float* y0, float* y1,
                                          long M, int K, float eps, int relu, void* stream) {
    return launch_rows128(x, Wfrag, cs, bb, y0, y1, M, K, eps, relu, nullptr, stream);
}

extern "C" int dldkd_in_proj_h16_rows128_groups(const float* x, const void* Wfrag, const float* cs, const float* bb, float* y0, float* y1,
                                                 long M, int K, float eps, int relu, const int32_t* groups, long n_groups, void* stream) {
    if (!groups) { set_error("in_proj_bf16_rows128_groups: null group table"); return DLDKD_EINVAL; }
    return launch_rows128(x, Wfrag, cs, bb, y0, y1, M, K, eps, relu, nullptr, stream, groups, n_groups);
}

extern "C" int dldkd_in_proj_h16_rows128_ok(int K) { return K >= 4 * RK && K % (2 * RK) == 0 && (long)127 * K * 4 + 128 <= 0xFFFFFFFFL; }

// Diagnostics: the same kernel with clock stamps; stamps = 12 x u64 per workgroup (tools/k4_timeline.py).
extern "C" int dldkd_debug_in_proj_rows128_timeline(const float* x, const void* Wfrag, const float* cs, const float* bb, float* y0,
                                                    float* y1, long M, int K, float eps, int relu, unsigned long long* stamps,
                                                    void* stream) {
    if (!stamps) { set_error("in_proj_rows128_timeline: stamps is null"); return DLDKD_EINVAL; }
    return launch_rows128(x, Wfrag, cs, bb, y0, y1, M, K, eps, relu, stamps, stream);
}
