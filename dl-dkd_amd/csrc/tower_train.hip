// Fused training towers, throughput mode (bf16 MFMA, fp32 statistics): everything of a clip-level encoder tower behind the
// input projection - TrainablePositionalEncoding (reference method/model_components.py:277-284), BertSelfAttention's three
// projections (:398-410), BertSelfOutput (:446-450) and out_mapping_linear (method/model.py:219) - as TWO row kernels forward and
// TWO backward around the fused attention kernels (attention_train_bf16.hip, bf16 in / out), instead of the chain of ~9 forward and
// ~20 backward launches per tower (LayerNorm, dropout, GEMM, bias / ReLU / mask passes, each with an fp32 round trip through HBM).
//
//   F1  y0 (+ position rows) -> LayerNorm -> dropout -> h1d (bf16, saved) -> q | k | v = h1d W^T + b (bf16)
//   F3  ctx -> dense -> dropout -> + h1d -> LayerNorm -> xh2 (normalised rows, bf16, saved), h2 -> [out mapping -> g (fp32)]
//   B3  dg -> [dh2 = dg Wo] -> LayerNorm backward (dgamma / dbeta by atomics) -> d(dense out) (bf16), dres (bf16), dctx = . Wd (bf16)
//   B1  dqkv -> dh1d = dqkv Wqkv + dres -> dropout mask -> LayerNorm backward -> dx1, dy0 = dx1 (.) [y0 > 0]
//
// Layout of every product: TRANSPOSED, Y^T = W X^T on v_mfma_f32_32x32x16_bf16 with the weights as the A operand (packed once per
// step in fragment order: one fragment = 1 KiB contiguous, read straight from L2) and the wave's 32 rows as the B operand, so an
// accumulator tile holds 32 output features on its registers and the wave's 32 ROWS ON ITS LANES:
//   * a lane owns half a row (192 features; its partner lane + 32 the other half): LayerNorm statistics, forward and backward, are
//     in-register sums + one v_permlane32_swap; dropout indices, row statistics and row validity are per-lane scalars;
//   * the accumulators of one product, rounded to bf16 in place, ARE the B operand of the next one (cdna_hip_programming.md section 3,
//     "An accumulator tile as the next MFMA's operand"): the weights of such a product are packed in the permuted k order
//     feature(ks, h, j) = 16 ks + 8 (j >> 2) + 4 h + (j & 3) - no LDS round trip between dense -> out mapping and
//     d(out) -> d(dense);
//   * a wave is independent of the other three of its workgroup: no LDS operand staging, no barrier.  The 32-row groups of a padded
//     batch that hold no valid clip (flags from the input projection's LayerNorm kernel) return at once: the padding is never
//     computed, read or written (the attention kernels take the sequences' lengths).
// Only the parameter gradients that sum over ROWS (LayerNorm gamma / beta) cross lanes: 64 features at a time through a wave-private
// LDS scratch, one atomic per feature and wave.  Weight and bias gradients are row contractions = GEMMs over the saved bf16 rows
// (gemm_bf16.hip) and column sums (colsum16 below).
// Dropout masks: Philox4x32-10 on the flat index of the (N, L, 384) tensor exactly as layernorm_kernel / dropout_fwd_kernel draw
// them, recomputed in the backward pass (no keep bytes are stored).
#include "common.hpp"

namespace dldkd {
namespace tt {

constexpr int kD = kHidden;               // 384
constexpr int kT = 12;                    // 32-feature accumulator tiles of a row
constexpr int kKS = 24;                   // 16-wide k-steps over 384
constexpr int kMatFrags = kKS * kT;       // fragments (1 KiB) of one 384 x 384 operand
#ifndef TT_WPB
#define TT_WPB 4       // waves (32-row groups) per workgroup of the four row kernels: 4, 2 or 1 (make TT_WPB=n)
#endif
#ifndef TT_ABL
#define TT_ABL 0       // diagnostic builds only (make TT_ABL=n): 1 no Philox, 2 no weight loads, 4 no LayerNorm-gradient column sums, 8 no MFMAs
#endif

typedef unsigned short u16;

struct Drop {
    unsigned long long seed, off;
    const unsigned long long* state;
    unsigned thresh;
    float scale;
    int on;
};

__device__ __forceinline__ float hswap_sum(float m) {
    const unsigned u = __builtin_bit_cast(unsigned, m);
    auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    return __builtin_bit_cast(float, (unsigned)r[0]) + __builtin_bit_cast(float, (unsigned)r[1]);
}

// keep flags (bits 0..3) of the four consecutive elements at flat index idx (a multiple of 4)
__device__ __forceinline__ unsigned keep4(const Drop& d, unsigned long long seed, unsigned long long off, size_t idx) {
    if (TT_ABL & 1) return 0xFu;
    // (no branch on d.on: p = 0 has thresh = 0 and keeps everything; a wave-uniform branch around 48 unrolled Philox calls made
    // hipcc spill the whole accumulator array)
    const unsigned long long c = off + (idx >> 2);
    unsigned rnd[4];
    philox4x32_10((unsigned)c, (unsigned)(c >> 32), 0u, 0u, (unsigned)seed, (unsigned)(seed >> 32), rnd);
    return (rnd[0] >= d.thresh ? 1u : 0u) | (rnd[1] >= d.thresh ? 2u : 0u) | (rnd[2] >= d.thresh ? 4u : 0u) | (rnd[3] >= d.thresh ? 8u : 0u);
}

__device__ __forceinline__ void unpack4(uint2 v, float (&o)[4]) {
    o[0] = __builtin_bit_cast(float, v.x << 16); o[1] = __builtin_bit_cast(float, v.x & 0xffff0000u);
    o[2] = __builtin_bit_cast(float, v.y << 16); o[3] = __builtin_bit_cast(float, v.y & 0xffff0000u);
}
__device__ __forceinline__ uint2 pack4(float a, float b, float c, float d) {
    uint2 pk;
    pk.x = (unsigned)f32_to_bf16_bits(a) | ((unsigned)f32_to_bf16_bits(b) << 16);
    pk.y = (unsigned)f32_to_bf16_bits(c) | ((unsigned)f32_to_bf16_bits(d) << 16);
    return pk;
}

// 8 consecutive accumulator registers (k-step s of the tile: features 16 s + 8 (j >> 2) + 4 h + (j & 3)) -> a B fragment
__device__ __forceinline__ bf16x8 pack8(const f32x16& a, int s) {
    bf16x8 b;
#pragma unroll
    for (int j = 0; j < 8; ++j) b[j] = (short)f32_to_bf16_bits(a[8 * s + j]);
    return b;
}

// acc[t][4 g + e] = vec[32 t + 8 g + 4 h + e]: a product's bias as its INITIAL accumulator (16-byte LDS reads straight into the
// tiles) instead of 192 adds behind the chain - the row kernels issue from one wave per SIMD, every VALU instruction shows
__device__ __forceinline__ void bias_acc(f32x16 (&acc)[kT], const float* lds_vec_plus_4h) {
#pragma unroll
    for (int t = 0; t < kT; ++t)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 b = *reinterpret_cast<const f32x4*>(lds_vec_plus_4h + 32 * t + 8 * g);
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[t][4 * g + e] = b[e];
        }
}
__device__ __forceinline__ void zero_acc(f32x16 (&acc)[kT]) {
#pragma unroll
    for (int t = 0; t < kT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
}

// acc[t] += sum_ks W[ks][t] (A operand: 32 output features x 16 k) * xb[ks] (B operand: 16 k x the wave's 32 rows).
// Wp: fragments [ks][t][lane] of 16 bytes = 288 KiB-fragments read straight from L2 through a 24-deep REGISTER RING of inline-asm
// loads: fragment i + 24 is requested the moment MFMA i has been issued, so ~23 loads (two k-steps, 736 MFMA cycles) are always in
// flight.  Written as plain C++ (two fragment arrays, loads one k-step ahead) hipcc kept ONE fragment register and emitted
// load / s_waitcnt vmcnt(0) / MFMA 288 times per product: an L2 round trip per MFMA, 260 us for the q | k | v kernel of a TVR batch.
// The compiler does not count asm loads in its own s_waitcnt bookkeeping; the waits here are explicit and tied to the register they
// release ("+v"), and any VMEM instruction the compiler adds in between only makes them more conservative (vmcnt is in order).
template <int N>
__device__ __forceinline__ void wait_vm(bf16x8& r) { asm volatile("s_waitcnt vmcnt(%1)" : "+v"(r) : "n"(N)); }
__device__ __forceinline__ void wait_vm_n(bf16x8& r, int n) {      // n folds to a constant: the callers are fully unrolled
    switch (n) {
#define DLDKD_W(k) case k: wait_vm<k>(r); break;
        DLDKD_W(0) DLDKD_W(1) DLDKD_W(2) DLDKD_W(3) DLDKD_W(4) DLDKD_W(5) DLDKD_W(6) DLDKD_W(7) DLDKD_W(8) DLDKD_W(9) DLDKD_W(10) DLDKD_W(11)
        DLDKD_W(12) DLDKD_W(13) DLDKD_W(14) DLDKD_W(15) DLDKD_W(16) DLDKD_W(17) DLDKD_W(18) DLDKD_W(19) DLDKD_W(20) DLDKD_W(21) DLDKD_W(22)
        default: wait_vm<23>(r); break;
#undef DLDKD_W
    }
}
__device__ __forceinline__ void ld_frag(bf16x8& r, const bf16x8* p) { asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(r) : "v"(p)); }

constexpr int kRing = 24;
__device__ __forceinline__ void gemm24(f32x16 (&acc)[kT], const bf16x8* __restrict__ Wp, const bf16x8 (&xb)[kKS], int lane) {
    const bf16x8* w = Wp + lane;
    bf16x8 ring[kRing];
    if (TT_ABL & 2) {
#pragma unroll
        for (int i = 0; i < kRing; ++i) ring[i] = w[i * 64];
#pragma unroll
        for (int i = 0; i < kMatFrags; ++i)
            if (!(TT_ABL & 8)) acc[i % kT] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ring[i % kRing], xb[i / kT], acc[i % kT], 0, 0, 0);
        return;
    }
#pragma unroll
    for (int i = 0; i < kRing; ++i) ld_frag(ring[i], w + i * 64);
#pragma unroll
    for (int i = 0; i < kMatFrags; ++i) {
        const int ks = i / kT, t = i % kT, slot = i % kRing;
        wait_vm_n(ring[slot], i + kRing <= kMatFrags ? kRing - 1 : kMatFrags - 1 - i);
        if (!(TT_ABL & 8)) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ring[slot], xb[ks], acc[t], 0, 0, 0);
        if (i + kRing < kMatFrags) ld_frag(ring[slot], w + (i + kRing) * 64);
    }
}

// Column sums over the wave's 32 rows (= the 32 lanes of a lane half) of per-lane values in the accumulator layout, for ONE tile:
// 16 registers per lane -> out[32 t + feature] += sum over rows.  Cross-lane adds by DPP (quad_perm xor 1 / xor 2, row_half_mirror,
// row_mirror: one v_add_f32_dpp each) + v_permlane16_swap for the two 16-lane rows of a half; lane r of each half keeps sum r and
// ONE atomic instruction (32 active lanes) adds the tile's 32 sums.  7 VALU per value; the first version went through a wave-private
// LDS scratch (64 writes + 32 reads + two lgkmcnt(0) waits per 64 features): 15 us of the 68-us b3 kernel, 25 of b1's 104.
template <int CTRL>
__device__ __forceinline__ float dpp_add(float v) {
    const int x = __builtin_bit_cast(int, v);
    return v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(x, x, CTRL, 0xF, 0xF, false));
}
__device__ __forceinline__ float half_rows_sum(float v) {        // sum over the 32 lanes of this lane's half, in every lane of it
    v = dpp_add<0xB1>(v);                                         // quad_perm [1,0,3,2]
    v = dpp_add<0x4E>(v);                                         // quad_perm [2,3,0,1]
    v = dpp_add<0x141>(v);                                        // row_half_mirror
    v = dpp_add<0x140>(v);                                        // row_mirror
    const unsigned u = __builtin_bit_cast(unsigned, v);
    auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    return __builtin_bit_cast(float, (unsigned)r[0]) + __builtin_bit_cast(float, (unsigned)r[1]);
}
__device__ __forceinline__ void colsum_tile(const f32x16& v, float* out_tile /* out + 32 t */, int lane) {
    if (TT_ABL & 4) return;
    const int r32 = lane & 31, h = lane >> 5;
    float keep = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const float sum = half_rows_sum(v[r]);
        keep = r32 == r ? sum : keep;
    }
    if (r32 < 16) atomicAdd(out_tile + (r32 & 3) + 8 * (r32 >> 2) + 4 * h, keep);
}

struct Wave {
    int lane, r32, h;
    long row, rowc;
    bool valid, run;
};

// the wave's 32-row group; run = false: nothing to do (past the end / a group of padding)
__device__ __forceinline__ Wave locate(long M, const unsigned char* flags) {
    Wave w;
    w.lane = threadIdx.x & 63;
    w.r32 = w.lane & 31;
    w.h = w.lane >> 5;
    const long g = (long)blockIdx.x * TT_WPB + (threadIdx.x >> 6);
    const long row0 = g * 32;
    w.run = row0 < M && (flags == nullptr || flags[g] != 0);
    w.row = row0 + w.r32;
    w.valid = w.row < M;
    w.rowc = w.valid ? w.row : (M - 1);
    return w;
}

// Per-feature vectors (biases, LayerNorm gains) are the same for every row: the workgroup copies them into LDS once and the sweeps
// read them from there (a lane half reads one address: a broadcast).  Per-ROW operands of a sweep (residual, normalised rows) are
// loaded as ONE batch of 48 loads per lane behind a scheduling barrier before the sweep starts: left to itself hipcc sinks every
// load next to its use and waits for it there - 48 to 96 exposed L2 round trips per sweep, more than the products themselves.
__device__ __forceinline__ void stage_vec(float* dst, const float* src) {
    for (int i = threadIdx.x; i < kD; i += 64 * TT_WPB) dst[i] = src[i];
}
__device__ __forceinline__ f32x4 vec4(const float* lds_vec_plus_4h, int fo) { return *reinterpret_cast<const f32x4*>(lds_vec_plus_4h + fo); }

// the four 8-byte pieces a lane holds of tile t of a bf16 row (accumulator layout); the sweeps request tile t + 1 before they work
// on tile t (one tile = 4 Philox calls + ~100 VALU: about one L2 round trip) - holding a whole row half (96 registers) next to the
// 192 accumulator values a sweep touches spilled 200-400 registers
// Memory side of a tile: a lane's four 8-byte pieces are features 8 g + 4 h .. + 3 of its row - as 8-byte accesses every instruction
// touches 32 rows with 16 contiguous bytes each.  Widened (cdna_hip_programming.md T21): the lane halves exchange one piece of each
// group pair through v_permlane32_swap, so lanes 0-31 access features 16 j .. 16 j + 7 and lanes 32-63 features 16 j + 8 .. + 15
// as ONE 16-byte access: half the memory instructions for the same bytes.
struct Tile4 { uint2 v[4]; };
__device__ __forceinline__ void swap_pair(uint2& a, uint2& b) {      // lanes 32-63 of a <-> lanes 0-31 of b (an involution)
    auto r0 = __builtin_amdgcn_permlane32_swap(a.x, b.x, false, false);
    auto r1 = __builtin_amdgcn_permlane32_swap(a.y, b.y, false, false);
    a.x = r0[0]; b.x = r0[1]; a.y = r1[0]; b.y = r1[1];
}
__device__ __forceinline__ Tile4 load_tile16(const u16* row_plus_4h, int t) {
    const u16* base = row_plus_4h + 32 * t + 4 * (threadIdx.x & 32 ? 1 : 0);     // row + 8 h (row_plus_4h carries 4 h)
    Tile4 r;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const uint4 w = *reinterpret_cast<const uint4*>(base + 16 * j);
        r.v[2 * j] = uint2{w.x, w.y};
        r.v[2 * j + 1] = uint2{w.z, w.w};
        swap_pair(r.v[2 * j], r.v[2 * j + 1]);
    }
    return r;
}
__device__ __forceinline__ void store_tile16(u16* row_plus_4h, int t, Tile4 r, bool valid) {
    u16* base = row_plus_4h + 32 * t + 4 * (threadIdx.x & 32 ? 1 : 0);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        swap_pair(r.v[2 * j], r.v[2 * j + 1]);                  // (every lane takes part in the exchange; only the store is predicated)
        if (valid) *reinterpret_cast<uint4*>(base + 16 * j) = uint4{r.v[2 * j].x, r.v[2 * j].y, r.v[2 * j + 1].x, r.v[2 * j + 1].y};
    }
}

// ------------------------------------------------------------------------------------------------------------------------- F1
struct F1Args {
    const float* y0;             // (M, 384) input projection output (after its ReLU)
    const float* pos;            // (L, 384) position rows
    int L;
    const float* gamma;
    const float* beta;
    float eps;
    Drop drop;
    const bf16x8* wqkv;          // natural pack of q | k | v: [3][24][12][64]
    const float* bias[3];
    const unsigned char* flags;
    long M;
    u16* h1d;                    // (M, 384) bf16
    u16* xh1;                    // (M, 384) bf16 normalised rows of y0 + pos (round to nearest even: all 16 bits are value bits)
    float* stats;                // [2][M] mean, rstd of y0 + pos
    u16* qkv;                    // (M, 1152) bf16
    unsigned char* relu_bits;    // (M, 48) bytes: bit (c % 8) of byte c / 8 of a row = [y0[row, c] > 0] - the ReLU mask of the input
                                 // projection for b1_kernel, in a plane of its own (round 6; until then bit 0 of every xh1 element, which
                                 // cost the LayerNorm backward a mantissa bit: ADVICE r04)
};

__global__ __launch_bounds__(64 * TT_WPB) __attribute__((amdgpu_waves_per_eu(1, 1))) void f1_kernel(const F1Args p) {
    __shared__ __attribute__((aligned(16))) float vs[5][kD];      // gamma, beta, bq, bk, bv
    stage_vec(vs[0], p.gamma); stage_vec(vs[1], p.beta);
    stage_vec(vs[2], p.bias[0]); stage_vec(vs[3], p.bias[1]); stage_vec(vs[4], p.bias[2]);
    __syncthreads();
    const Wave w = locate(p.M, p.flags);
    if (!w.run) return;
    const int h = w.h, lane = w.lane;
    unsigned long long seed = p.drop.seed, off = p.drop.off;
    if (p.drop.on && p.drop.state != nullptr) { seed = p.drop.state[0]; off += p.drop.state[1]; }

    // x = y0 + pos: this lane's 192 features (k-steps x 8) in two batches of 48 16-byte loads; bit ks*8+e of pos_bits: y0 > 0
    float x[kKS][8];
    unsigned relu[kKS];
    float s = 0.f;
    {
        const float* xr = p.y0 + w.rowc * kD + 8 * h;
        const float* pr = p.pos + (w.rowc % p.L) * kD + 8 * h;
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            f32x4 ld[kKS / 2][4];
#pragma unroll
            for (int k = 0; k < kKS / 2; ++k) {
                const int ks = b * (kKS / 2) + k;
                ld[k][0] = *reinterpret_cast<const f32x4*>(xr + 16 * ks); ld[k][1] = *reinterpret_cast<const f32x4*>(xr + 16 * ks + 4);
                ld[k][2] = *reinterpret_cast<const f32x4*>(pr + 16 * ks); ld[k][3] = *reinterpret_cast<const f32x4*>(pr + 16 * ks + 4);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int k = 0; k < kKS / 2; ++k) {
                const int ks = b * (kKS / 2) + k;
                unsigned m = 0;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    m |= (ld[k][0][e] > 0.f ? 1u : 0u) << e | (ld[k][1][e] > 0.f ? 1u : 0u) << (4 + e);
                    x[ks][e] = ld[k][0][e] + ld[k][2][e];
                    x[ks][4 + e] = ld[k][1][e] + ld[k][3][e];
                }
                relu[ks] = m;
#pragma unroll
                for (int e = 0; e < 8; ++e) s += x[ks][e];
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const float mean = hswap_sum(s) * (1.f / kD);
    float q = 0.f;
#pragma unroll
    for (int ks = 0; ks < kKS; ++ks)
#pragma unroll
        for (int e = 0; e < 8; ++e) { const float d = x[ks][e] - mean; q += d * d; }
    const float rstd = rsqrtf(hswap_sum(q) * (1.f / kD) + p.eps);
    if (w.valid && h == 0) { p.stats[w.row] = mean; p.stats[p.M + w.row] = rstd; }
    bf16x8 xb[kKS];
    const float* gm = vs[0] + 8 * h;
    const float* bt = vs[1] + 8 * h;
    const size_t didx = (size_t)w.row * kD + 8 * h;
#pragma unroll
    for (int ks = 0; ks < kKS; ++ks) {
        const f32x4 g0 = *reinterpret_cast<const f32x4*>(gm + 16 * ks), g1 = *reinterpret_cast<const f32x4*>(gm + 16 * ks + 4);
        const f32x4 b0 = *reinterpret_cast<const f32x4*>(bt + 16 * ks), b1 = *reinterpret_cast<const f32x4*>(bt + 16 * ks + 4);
        const unsigned k0 = keep4(p.drop, seed, off, didx + 16 * ks), k1 = keep4(p.drop, seed, off, didx + 16 * ks + 4);
        bf16x8 xn;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float xh = (x[ks][e] - mean) * rstd;
            const float v = xh * (e < 4 ? g0[e] : g1[e - 4]) + (e < 4 ? b0[e] : b1[e - 4]);
            const unsigned kb = e < 4 ? (k0 >> e) & 1u : (k1 >> (e - 4)) & 1u;
            xb[ks][e] = (short)f32_to_bf16_bits(kb ? v * p.drop.scale : 0.f);
            xn[e] = (short)f32_to_bf16_bits(xh);
        }
        if (w.valid) {
            *reinterpret_cast<bf16x8*>(p.h1d + w.row * kD + 8 * h + 16 * ks) = xb[ks];
            *reinterpret_cast<bf16x8*>(p.xh1 + w.row * kD + 8 * h + 16 * ks) = xn;
        }
    }
    {
        // the ReLU mask as a bit plane: this lane's relu[ks] is byte 2 ks + h of its row (columns 16 ks + 8 h ..); one
        // v_permlane32_swap pairs it with the other half-row lane's byte (r[0] is the h = 0 lane's value on BOTH lanes, r[1] the
        // h = 1 lane's), and lane h stores ushorts 12 h .. 12 h + 11 of the row's 24: 24 contiguous bytes per lane
        unsigned wd[6];
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            unsigned lo[2], hi[2];
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
                auto a = __builtin_amdgcn_permlane32_swap(relu[12 * hh + 2 * j], relu[12 * hh + 2 * j], false, false);
                auto b = __builtin_amdgcn_permlane32_swap(relu[12 * hh + 2 * j + 1], relu[12 * hh + 2 * j + 1], false, false);
                lo[hh] = ((unsigned)a[0] & 0xffu) | (((unsigned)a[1] & 0xffu) << 8);
                hi[hh] = ((unsigned)b[0] & 0xffu) | (((unsigned)b[1] & 0xffu) << 8);
            }
            wd[j] = h ? (lo[1] | (hi[1] << 16)) : (lo[0] | (hi[0] << 16));
        }
        if (w.valid && p.relu_bits != nullptr) {
            unsigned char* rb = p.relu_bits + (size_t)w.row * 48 + 24 * h;
            *reinterpret_cast<uint2*>(rb) = uint2{wd[0], wd[1]};
            *reinterpret_cast<uint2*>(rb + 8) = uint2{wd[2], wd[3]};
            *reinterpret_cast<uint2*>(rb + 16) = uint2{wd[4], wd[5]};
        }
    }
#pragma unroll 1
    for (int c = 0; c < 3; ++c) {
        f32x16 acc[kT];
        bias_acc(acc, vs[2 + c] + 4 * h);
        gemm24(acc, p.wqkv + (size_t)c * kMatFrags * 64, xb, lane);
        u16* orow = p.qkv + w.row * (3 * kD) + c * kD + 4 * h;
#pragma unroll
        for (int t = 0; t < kT; ++t) {
            Tile4 o;
#pragma unroll
            for (int g = 0; g < 4; ++g) o.v[g] = pack4(acc[t][4 * g], acc[t][4 * g + 1], acc[t][4 * g + 2], acc[t][4 * g + 3]);
            store_tile16(orow, t, o, w.valid);
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------------- F3
struct F3Args {
    const u16* ctx;              // (M, 384) bf16 attention output
    const u16* h1d;              // (M, 384) bf16 residual
    const bf16x8* wd;            // natural pack of the dense weight
    const float* bd;
    Drop drop;
    const float* gamma;
    const float* beta;
    float eps;
    const bf16x8* wo;            // permuted pack of the out mapping, or null (query towers)
    const float* bo;
    const unsigned char* flags;
    long M;
    u16* xh2;                    // (M, 384) bf16 normalised rows (LayerNorm backward)
    float* rstd2;                // [M]
    u16* h2_16;                  // (M, 384) bf16 LayerNorm output (operand of the out mapping's weight gradient) or null
    float* h2_32;                // (M, 384) fp32 LayerNorm output (query towers: modular pooling) or null
    float* g;                    // (M, 384) fp32 out mapping output or null
};

// VIDEO: h2 as bf16 + the out mapping; otherwise (query towers) h2 as fp32 rows for the modular pooling
template <bool VIDEO>
__global__ __launch_bounds__(64 * TT_WPB) __attribute__((amdgpu_waves_per_eu(1, 1))) void f3_kernel(const F3Args p) {
    __shared__ __attribute__((aligned(16))) float vs[4][kD];      // bd, gamma, beta, bo
    stage_vec(vs[0], p.bd); stage_vec(vs[1], p.gamma); stage_vec(vs[2], p.beta);
    if constexpr (VIDEO) stage_vec(vs[3], p.bo);
    __syncthreads();
    const Wave w = locate(p.M, p.flags);
    if (!w.run) return;
    const int h = w.h, lane = w.lane;
    unsigned long long seed = p.drop.seed, off = p.drop.off;
    if (p.drop.on && p.drop.state != nullptr) { seed = p.drop.state[0]; off += p.drop.state[1]; }
    bf16x8 xb[kKS];
    {
        const u16* cr = p.ctx + w.rowc * kD + 8 * h;
#pragma unroll
        for (int ks = 0; ks < kKS; ++ks) xb[ks] = *reinterpret_cast<const bf16x8*>(cr + 16 * ks);
    }
    f32x16 acc[kT];
    bias_acc(acc, vs[0] + 4 * h);                               // dense bias as the initial accumulator
    gemm24(acc, p.wd, xb, lane);
    const u16* hres = p.h1d + w.rowc * kD + 4 * h;
    float s = 0.f;
    const size_t didx = (size_t)w.row * kD + 4 * h;
    Tile4 nxt = load_tile16(hres, 0);
#pragma unroll
    for (int t = 0; t < kT; ++t) {
        const Tile4 cur = nxt;
        if (t + 1 < kT) nxt = load_tile16(hres, t + 1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int fo = 32 * t + 8 * g;
            const unsigned kb = keep4(p.drop, seed, off, didx + fo);
            float r4[4];
            unpack4(cur.v[g], r4);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float v = acc[t][4 * g + e];
                const float y = (((kb >> e) & 1u) ? v * p.drop.scale : 0.f) + r4[e];
                acc[t][4 * g + e] = y;
                s += y;
            }
            // pin this group's arithmetic HERE: the sweep is one basic block and hipcc otherwise runs the 48 groups' Philox rounds
            // first and parks their results in scratch (200+ registers spilled) until the accumulators are read
            asm volatile("" : "+v"(s));
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const float mean = hswap_sum(s) * (1.f / kD);
    float q = 0.f;
#pragma unroll
    for (int t = 0; t < kT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) { const float d = acc[t][r] - mean; q += d * d; }
    const float rstd = rsqrtf(hswap_sum(q) * (1.f / kD) + p.eps);
    if (w.valid && h == 0) p.rstd2[w.row] = rstd;
    const float* gmp = vs[1] + 4 * h;
    const float* btp = vs[2] + 4 * h;
    u16* xh2r = p.xh2 + w.row * kD + 4 * h;
    u16* h16r = p.h2_16 + w.row * kD + 4 * h;
    float* h32r = p.h2_32 + w.row * kD + 4 * h;
    (void)h16r; (void)h32r;
#pragma unroll
    for (int t = 0; t < kT; ++t) {
        Tile4 oxh, oh2;
        (void)oh2;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int fo = 32 * t + 8 * g;
            const f32x4 gm = vec4(gmp, fo), bt = vec4(btp, fo);
            float xh[4], y[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                xh[e] = (acc[t][4 * g + e] - mean) * rstd;
                y[e] = xh[e] * gm[e] + bt[e];
                acc[t][4 * g + e] = y[e];
            }
            oxh.v[g] = pack4(xh[0], xh[1], xh[2], xh[3]);
            if constexpr (VIDEO) oh2.v[g] = pack4(y[0], y[1], y[2], y[3]);
            else if (w.valid) *reinterpret_cast<f32x4*>(h32r + fo) = f32x4{y[0], y[1], y[2], y[3]};
        }
        store_tile16(xh2r, t, oxh, w.valid);
        if constexpr (VIDEO) store_tile16(h16r, t, oh2, w.valid);
        // h2 of this tile, rounded, is the out mapping's B operand for k-steps 2 t, 2 t + 1 (permuted k order)
        if constexpr (VIDEO) {
            xb[2 * t] = pack8(acc[t], 0);
            xb[2 * t + 1] = pack8(acc[t], 1);
        }
    }
    if constexpr (!VIDEO) return;
    bias_acc(acc, vs[3] + 4 * h);
    gemm24(acc, p.wo, xb, lane);
    float* gr = p.g + w.row * kD + 4 * h;
#pragma unroll
    for (int t = 0; t < kT; ++t)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int fo = 32 * t + 8 * g;
            if (w.valid)
                *reinterpret_cast<f32x4*>(gr + fo) = f32x4{acc[t][4 * g], acc[t][4 * g + 1], acc[t][4 * g + 2], acc[t][4 * g + 3]};
        }
}

// ------------------------------------------------------------------------------------------------------------------------- B3
struct B3Args {
    const float* dg;             // (M, 384) fp32: gradient of the out mapping's output (wot != null) or of h2 itself (wot == null)
    const bf16x8* wot;           // natural pack of Wo^T or null
    const u16* xh2;
    const float* rstd2;
    const float* gamma;
    Drop drop;
    const bf16x8* wdt;           // permuted pack of Wd^T
    const unsigned char* flags;
    long M;
    u16* ddo;                    // (M, 384) bf16 gradient of the dense output (before its dropout)
    u16* dctx;                   // (M, 384) bf16 gradient of the attention output
    u16* dres;                   // (M, 384) bf16 gradient reaching h1d through the residual
    float* dgamma;               // [384] += (zeroed by the caller)
    float* dbeta;                // [384] +=
    u16* dg16;                   // (M, 384) bf16 copy of dg for the out mapping's weight-gradient GEMM (wot != null), or null
    u16* dh2_16;                 // SUMS = false, wot != null: (M, 384) bf16 gradient of the LayerNorm output h2 (wot == null: it is dg)
};

// SUMS: the LayerNorm parameter gradients (column sums over the batch rows of dh2 and dh2 (.) xh2) inside this kernel - DPP
// reductions + one atomic per feature and wave: 24 of this kernel's 75 us at the TVR batch, 36 of b1_kernel's 116
// (tools/r05_abl_tower2.sh).  false: the kernel leaves dh2 as bf16 rows and the column sums to the launch that reduces the
// weight gradients' split-K planes (gemm_bf16.hip, dw_finish_kernel: plain coalesced row sweeps).
template <bool SUMS>
__global__ __launch_bounds__(64 * TT_WPB) __attribute__((amdgpu_waves_per_eu(1, 1))) void b3_kernel(const B3Args p) {
    __shared__ __attribute__((aligned(16))) float vs[kD];          // gamma
    stage_vec(vs, p.gamma);
    __syncthreads();
    const Wave w = locate(p.M, p.flags);
    if (!w.run) return;
    const int h = w.h, lane = w.lane;
    unsigned long long seed = p.drop.seed, off = p.drop.off;
    if (p.drop.on && p.drop.state != nullptr) { seed = p.drop.state[0]; off += p.drop.state[1]; }
    f32x16 acc[kT];
    bf16x8 xb[kKS];
    if (p.wot != nullptr) {
        const float* gr = p.dg + w.rowc * kD + 8 * h;
#pragma unroll
        for (int b = 0; b < 2; ++b) {                              // two batches of 24 16-byte loads
            f32x4 ld[kKS / 2][2];
#pragma unroll
            for (int k = 0; k < kKS / 2; ++k) {
                const int ks = b * (kKS / 2) + k;
                ld[k][0] = *reinterpret_cast<const f32x4*>(gr + 16 * ks);
                ld[k][1] = *reinterpret_cast<const f32x4*>(gr + 16 * ks + 4);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int k = 0; k < kKS / 2; ++k) {
                const int ks = b * (kKS / 2) + k;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    xb[ks][e] = (short)f32_to_bf16_bits(w.valid ? ld[k][0][e] : 0.f);
                    xb[ks][4 + e] = (short)f32_to_bf16_bits(w.valid ? ld[k][1][e] : 0.f);
                }
                if (p.dg16 != nullptr && w.valid) *reinterpret_cast<bf16x8*>(p.dg16 + w.row * kD + 8 * h + 16 * ks) = xb[ks];
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        zero_acc(acc);
        gemm24(acc, p.wot, xb, lane);
    } else {
        const float* dgr = p.dg + w.rowc * kD + 4 * h;
#pragma unroll
        for (int t = 0; t < kT; ++t)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(dgr + 32 * t + 8 * g);
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[t][4 * g + e] = w.valid ? v[e] : 0.f;
            }
    }
    // acc = dh2 (rows past the end: zero).  LayerNorm backward: dyg = dh2 gamma, dx = rstd (dyg - mean(dyg) - xh mean(dyg xh))
    const u16* xrow = p.xh2 + w.rowc * kD + 4 * h;
    float m1 = 0.f, m2 = 0.f;
    const float* gmp = vs + 4 * h;
    const size_t didx = (size_t)w.row * kD + 4 * h;
    Tile4 nx0 = load_tile16(xrow, 0), nx1 = load_tile16(xrow, 1);
    u16* dh2r = p.dh2_16 + w.row * kD + 4 * h;
    const bool put_dh2 = !SUMS && p.wot != nullptr;
#pragma unroll
    for (int pr = 0; pr < kT / 2; ++pr) {
        f32x16 pg[2];
        const Tile4 cu[2] = {nx0, nx1};
        if (pr + 1 < kT / 2) { nx0 = load_tile16(xrow, 2 * pr + 2); nx1 = load_tile16(xrow, 2 * pr + 3); }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
            const int t = 2 * pr + tt;
            Tile4 odh;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 gm = vec4(gmp, 32 * t + 8 * g);
                float xh[4];
                unpack4(cu[tt].v[g], xh);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float dy = acc[t][4 * g + e];
                    if constexpr (SUMS) pg[tt][4 * g + e] = dy * xh[e];
                    m1 += dy * gm[e];
                    m2 += dy * gm[e] * xh[e];
                }
                if constexpr (!SUMS) odh.v[g] = pack4(acc[t][4 * g], acc[t][4 * g + 1], acc[t][4 * g + 2], acc[t][4 * g + 3]);
                asm volatile("" : "+v"(m1), "+v"(m2));      // (pins the group's arithmetic here: see f3_kernel)
                __builtin_amdgcn_sched_barrier(0);
            }
            if constexpr (!SUMS) {
                if (put_dh2) store_tile16(dh2r, t, odh, w.valid);
            }
        }
        if constexpr (SUMS) {
            colsum_tile(pg[0], p.dgamma + 64 * pr, lane);
            colsum_tile(pg[1], p.dgamma + 64 * pr + 32, lane);
            colsum_tile(acc[2 * pr], p.dbeta + 64 * pr, lane);
            colsum_tile(acc[2 * pr + 1], p.dbeta + 64 * pr + 32, lane);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    m1 = hswap_sum(m1) * (1.f / kD);
    m2 = hswap_sum(m2) * (1.f / kD);
    const float rstd = p.rstd2[w.rowc];
    u16* dresr = p.dres + w.row * kD + 4 * h;
    u16* ddor = p.ddo + w.row * kD + 4 * h;
    Tile4 nxt = load_tile16(xrow, 0);
#pragma unroll
    for (int t = 0; t < kT; ++t) {
        const Tile4 cur = nxt;
        if (t + 1 < kT) nxt = load_tile16(xrow, t + 1);
        __builtin_amdgcn_sched_barrier(0);
        Tile4 ores, oddo;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int fo = 32 * t + 8 * g;
            const f32x4 gm = vec4(gmp, fo);
            float xh[4], dx[4], dd[4];
            unpack4(cur.v[g], xh);
            const unsigned kb = keep4(p.drop, seed, off, didx + fo);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                dx[e] = rstd * (acc[t][4 * g + e] * gm[e] - m1 - xh[e] * m2);
                dd[e] = ((kb >> e) & 1u) ? dx[e] * p.drop.scale : 0.f;
                acc[t][4 * g + e] = dd[e];
            }
            ores.v[g] = pack4(dx[0], dx[1], dx[2], dx[3]);
            oddo.v[g] = pack4(dd[0], dd[1], dd[2], dd[3]);
        }
        store_tile16(dresr, t, ores, w.valid);
        store_tile16(ddor, t, oddo, w.valid);
        xb[2 * t] = pack8(acc[t], 0);
        xb[2 * t + 1] = pack8(acc[t], 1);
        __builtin_amdgcn_sched_barrier(0);
    }
    zero_acc(acc);
    gemm24(acc, p.wdt, xb, lane);
    u16* dctxr = p.dctx + w.row * kD + 4 * h;
#pragma unroll
    for (int t = 0; t < kT; ++t) {
        Tile4 o;
#pragma unroll
        for (int g = 0; g < 4; ++g) o.v[g] = pack4(acc[t][4 * g], acc[t][4 * g + 1], acc[t][4 * g + 2], acc[t][4 * g + 3]);
        store_tile16(dctxr, t, o, w.valid);
    }
}

// ------------------------------------------------------------------------------------------------------------------------- B1
struct B1Args {
    const u16* dqkv;             // (M, 1152) bf16
    const u16* dres;             // (M, 384) bf16
    const bf16x8* wqkvt;         // natural pack of [Wq; Wk; Wv]^T: [72][12][64]
    const u16* xh1;              // (M, 384) bf16 normalised rows of y0 + pos (f1_kernel): all the LayerNorm backward pass needs of them
    const unsigned char* relu_bits;   // (M, 48) bytes: the input projection's ReLU mask [y0 > 0], one bit per element (f1_kernel; needed
                                 // when relu_mask) - neither y0 nor the position rows are read again
    const float* stats;          // [2][M]: rstd = stats[M + row]
    const float* gamma;
    Drop drop;
    const unsigned char* flags;
    long M;
    int relu_mask;               // dy0 = dx1 (.) [y0 > 0] (the input projection's ReLU); 0: dy0 = dx1
    float* dy0;                  // (M, 384) fp32 (rows of skipped groups: zeros)
    float* dx1;                  // (M, 384) fp32 gradient of y0 + pos (position-table gradient = its sum over sequences), or null
    float* dgamma;
    float* dbeta;
    u16* dz16;                   // SUMS = false: (M, 384) bf16 gradient of the first LayerNorm's output (rows of skipped groups: not written)
    u16* dy16;                   // (M, 384) bf16 copy of dy0 for the input projection's weight-gradient GEMM (rows of skipped groups: zeros,
                                 // as in dy0), or null
};

template <bool DX1, bool SUMS>      // SUMS: see b3_kernel
__global__ __launch_bounds__(64 * TT_WPB) __attribute__((amdgpu_waves_per_eu(1, 1))) void b1_kernel(const B1Args p) {
    __shared__ __attribute__((aligned(16))) float vs[kD];          // gamma
    stage_vec(vs, p.gamma);
    __syncthreads();
    const Wave w = locate(p.M, nullptr);
    if (!w.run) return;
    const int h = w.h, lane = w.lane;
    if (p.flags != nullptr && p.flags[(long)blockIdx.x * TT_WPB + (threadIdx.x >> 6)] == 0) {
        // a group of padding: its gradients are exact zeros (the weight-gradient GEMMs skip these rows, plain column sums read them)
        if (w.valid) {
#pragma unroll
            for (int i = 0; i < 48; ++i) {
                *reinterpret_cast<f32x4*>(p.dy0 + w.row * kD + 192 * h + 4 * i) = f32x4{0.f, 0.f, 0.f, 0.f};
                if constexpr (DX1) *reinterpret_cast<f32x4*>(p.dx1 + w.row * kD + 192 * h + 4 * i) = f32x4{0.f, 0.f, 0.f, 0.f};
            }
            if (p.dy16 != nullptr) {
#pragma unroll
                for (int i = 0; i < 24; ++i) *reinterpret_cast<uint4*>(p.dy16 + w.row * kD + 192 * h + 8 * i) = uint4{0u, 0u, 0u, 0u};
            }
        }
        return;
    }
    unsigned long long seed = p.drop.seed, off = p.drop.off;
    if (p.drop.on && p.drop.state != nullptr) { seed = p.drop.state[0]; off += p.drop.state[1]; }
    f32x16 acc[kT];
    zero_acc(acc);
#pragma unroll 1
    for (int c = 0; c < 3; ++c) {
        bf16x8 xb[kKS];
        const u16* dr = p.dqkv + w.rowc * (3 * kD) + c * kD + 8 * h;
#pragma unroll
        for (int ks = 0; ks < kKS; ++ks) xb[ks] = *reinterpret_cast<const bf16x8*>(dr + 16 * ks);
        gemm24(acc, p.wqkvt + (size_t)c * kMatFrags * 64, xb, lane);
    }
    const float rstd = p.stats[p.M + w.rowc];
    const u16* xrow = p.xh1 + w.rowc * kD + 4 * h;
    const u16* rrow = p.dres + w.rowc * kD + 4 * h;
    const float* gmp = vs + 4 * h;
    const size_t didx = (size_t)w.row * kD + 4 * h;
    float* dy0r = p.dy0 + w.row * kD + 4 * h;
    float* dx1r = p.dx1 + w.row * kD + 4 * h;
    (void)dx1r;
    float m1 = 0.f, m2 = 0.f;
    Tile4 nx[2] = {load_tile16(xrow, 0), load_tile16(xrow, 1)}, nr[2] = {load_tile16(rrow, 0), load_tile16(rrow, 1)};
    u16* dzr = p.dz16 + w.row * kD + 4 * h;
#pragma unroll
    for (int pr = 0; pr < kT / 2; ++pr) {
        f32x16 pg[2];
        const Tile4 cx[2] = {nx[0], nx[1]}, cr[2] = {nr[0], nr[1]};
        if (pr + 1 < kT / 2) {
            nx[0] = load_tile16(xrow, 2 * pr + 2); nx[1] = load_tile16(xrow, 2 * pr + 3);
            nr[0] = load_tile16(rrow, 2 * pr + 2); nr[1] = load_tile16(rrow, 2 * pr + 3);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
            const int t = 2 * pr + tt;
            Tile4 odz;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 gm = vec4(gmp, 32 * t + 8 * g);
                float res[4], xh[4];
                unpack4(cr[tt].v[g], res);
                unpack4(cx[tt].v[g], xh);
                const unsigned kb = keep4(p.drop, seed, off, didx + 32 * t + 8 * g);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float dh = acc[t][4 * g + e] + res[e];
                    dh = (w.valid && ((kb >> e) & 1u)) ? dh * p.drop.scale : 0.f;
                    acc[t][4 * g + e] = dh;
                    if constexpr (SUMS) pg[tt][4 * g + e] = dh * xh[e];
                    m1 += dh * gm[e];
                    m2 += dh * gm[e] * xh[e];
                }
                if constexpr (!SUMS) odz.v[g] = pack4(acc[t][4 * g], acc[t][4 * g + 1], acc[t][4 * g + 2], acc[t][4 * g + 3]);
                asm volatile("" : "+v"(m1), "+v"(m2));      // (pins the group's arithmetic here: see f3_kernel)
                __builtin_amdgcn_sched_barrier(0);
            }
            if constexpr (!SUMS) store_tile16(dzr, t, odz, w.valid);
        }
        if constexpr (SUMS) {
            colsum_tile(pg[0], p.dgamma + 64 * pr, lane);
            colsum_tile(pg[1], p.dgamma + 64 * pr + 32, lane);
            colsum_tile(acc[2 * pr], p.dbeta + 64 * pr, lane);
            colsum_tile(acc[2 * pr + 1], p.dbeta + 64 * pr + 32, lane);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    m1 = hswap_sum(m1) * (1.f / kD);
    m2 = hswap_sum(m2) * (1.f / kD);
    Tile4 nxt = load_tile16(xrow, 0);
    u16* dy16r = p.dy16 + w.row * kD + 4 * h;
    // the ReLU bits of this lane's columns 32 t + 8 g + 4 h ..+4: nibble h of byte 4 t + g of the row = one dword per tile t
    const unsigned* rbits = reinterpret_cast<const unsigned*>(p.relu_bits + (size_t)w.rowc * 48);
    unsigned nbits = p.relu_mask ? rbits[0] : 0xffffffffu;
#pragma unroll
    for (int t = 0; t < kT; ++t) {
        const Tile4 cur = nxt;
        const unsigned cbits = nbits;
        if (t + 1 < kT) {
            nxt = load_tile16(xrow, t + 1);
            if (p.relu_mask) nbits = rbits[t + 1];
        }
        __builtin_amdgcn_sched_barrier(0);
        Tile4 ody;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int fo = 32 * t + 8 * g;
            const f32x4 gm = vec4(gmp, fo);
            float xh[4];
            const uint2 xw = cur.v[g];
            unpack4(xw, xh);
            const unsigned pos4 = (cbits >> (8 * g + 4 * h)) & 0xfu;
            f32x4 dx, dy;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                dx[e] = rstd * (acc[t][4 * g + e] * gm[e] - m1 - xh[e] * m2);
                dy[e] = (!p.relu_mask || ((pos4 >> e) & 1u)) ? dx[e] : 0.f;
            }
            ody.v[g] = pack4(dy[0], dy[1], dy[2], dy[3]);
            if (w.valid) {
                *reinterpret_cast<f32x4*>(dy0r + fo) = dy;
                if constexpr (DX1) *reinterpret_cast<f32x4*>(dx1r + fo) = dx;
            }
        }
        if (p.dy16 != nullptr) store_tile16(dy16r, t, ody, w.valid);
        __builtin_amdgcn_sched_barrier(0);
    }
}

// ----------------------------------------------------------------------------------------------------------------- weight packs
// mode 0: natural, forward   A[o][k = in]  = W[o][in]             fragments [src][ks][t], t = 32-output tile
// mode 1: permuted, forward  the same with in = 32 (ks >> 1) + 16 (ks & 1) + 8 (j >> 2) + 4 h + (j & 3)
// mode 2: natural, transposed A[i][k = out] = W[out][i]           fragments [ks over all sources' outputs][t], t = 32-input tile
// mode 3: permuted, transposed
// mode 4: plain cast of src[0] (nsrc groups of 8 consecutive elements) to bf16 - the input projection's weight, which the tower's
//         first GEMM (gemm_bf16_nt16) reads as bf16: one launch per tower and step prepares every weight operand
struct PackJob {
    const float* src[3];         // (384, 384) row-major weights (nn.Linear: [out][in])
    int nsrc, mode;
    bf16x8* out;
};
struct PackArgs {
    PackJob job[7];
    int njobs;
    const float* mask;           // optional extra job (blockIdx.y == njobs): lens[n] = number of mask[n, :L] entries > 0, one wave per sequence
    int n_seq, L;
    int32_t* lens;
};

__global__ __launch_bounds__(256) void pack_kernel(const PackArgs a) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if ((int)blockIdx.y == a.njobs) {
        const int n = idx >> 6, lane = idx & 63;
        if (n >= a.n_seq) return;
        float c = 0.f;
        for (int l = lane; l < a.L; l += 64) c += a.mask[(size_t)n * a.L + l] > 0.f ? 1.f : 0.f;
        c = wave_sum(c);
        if (lane == 0) a.lens[n] = (int32_t)c;
        return;
    }
    const PackJob& jb = a.job[blockIdx.y];
    if (jb.mode == 4) {
        if (idx >= jb.nsrc) return;
        const f32x4 lo = reinterpret_cast<const f32x4*>(jb.src[0])[2 * (size_t)idx], hi = reinterpret_cast<const f32x4*>(jb.src[0])[2 * (size_t)idx + 1];
        bf16x8 v;
#pragma unroll
        for (int j = 0; j < 4; ++j) { v[j] = (short)f32_to_bf16_bits(lo[j]); v[4 + j] = (short)f32_to_bf16_bits(hi[j]); }
        jb.out[idx] = v;
        return;
    }
    const int frag = idx >> 6, lane = idx & 63, r32 = lane & 31, h = lane >> 5;
    if (frag >= jb.nsrc * kMatFrags) return;
    bf16x8 v;
    if (jb.mode < 2) {
        const int c = frag / kMatFrags, rem = frag - c * kMatFrags, ks = rem / kT, t = rem - ks * kT;
        const float* Wr = jb.src[c] + (size_t)(32 * t + r32) * kD;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k = jb.mode == 0 ? 16 * ks + 8 * h + j : 32 * (ks >> 1) + 16 * (ks & 1) + 8 * (j >> 2) + 4 * h + (j & 3);
            v[j] = (short)f32_to_bf16_bits(Wr[k]);
        }
    } else {
        const int ksg = frag / kT, t = frag - ksg * kT, c = ksg / kKS, ks = ksg - c * kKS;
        const float* Wc = jb.src[c] + 32 * t + r32;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k = jb.mode == 2 ? 16 * ks + 8 * h + j : 32 * (ks >> 1) + 16 * (ks & 1) + 8 * (j >> 2) + 4 * h + (j & 3);
            v[j] = (short)f32_to_bf16_bits(Wc[(size_t)k * kD]);
        }
    }
    jb.out[idx] = v;
}

// ---------------------------------------------------------------------------------------------------------------- column sums
// out[c] += sum over the rows of x (M, ld) bf16 whose 32-row group is flagged (flags null: all rows); N columns from column c0.
// Grid (ceil(N / 64), slabs): a workgroup sums 64 columns over its slab of rows, 4 waves = 4 row phases, one atomic per column.
__global__ __launch_bounds__(256) void colsum16_kernel(const u16* __restrict__ x, int ld, int c0, int N, long M,
                                                       const unsigned char* __restrict__ flags, float* __restrict__ out, int rows_per_slab) {
    __shared__ float red[3][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + lane;
    const long r0 = (long)blockIdx.y * rows_per_slab, r1 = min(M, r0 + rows_per_slab);
    float s = 0.f;
    if (c < N) {
        for (long g0 = r0 + 32 * wave; g0 < r1; g0 += 128) {          // (rows_per_slab is a multiple of 128: whole groups per wave)
            if (flags != nullptr && flags[g0 >> 5] == 0) continue;
            const long e = min(r1, g0 + 32);
            for (long r = g0; r < e; ++r) s += bf16_bits_to_f32(x[r * ld + c0 + c]);
        }
    }
    if (wave > 0) red[wave - 1][lane] = s;
    __syncthreads();
    if (wave == 0 && c < N) atomicAdd(out + c, s + red[0][lane] + red[1][lane] + red[2][lane]);
}

static Drop make_drop(float p, unsigned long long seed, unsigned long long off, const unsigned long long* state) {
    Drop d{};
    d.on = p > 0.f;
    d.seed = seed;
    d.off = off;
    d.state = state;
    const double t = (double)p * 4294967296.0;
    d.thresh = t >= 4294967295.0 ? 4294967295u : (unsigned)t;
    d.scale = 1.0f / (1.0f - p);
    return d;
}

static bool bad_p(float p) { return !(p >= 0.f && p < 1.f); }
static unsigned grid_of(long M) { return (unsigned)((M + 32 * TT_WPB - 1) / (32 * TT_WPB)); }

}  // namespace tt
}  // namespace dldkd

using namespace dldkd;

extern "C" {

size_t dldkd_tower_train_pack_bytes(int n_mats) { return (size_t)n_mats * tt::kMatFrags * 1024; }

static int tower_train_pack_impl(const float* const* host_src, const int* host_nsrc, const int* host_mode, void* const* host_out, int njobs,
                                 const float* mask, int n_seq, int L, int32_t* lens, void* stream) {
    if (njobs < 1 || njobs > 7 || !host_src || !host_nsrc || !host_mode || !host_out) { set_error("tower_train_pack: 1..7 jobs"); return DLDKD_EINVAL; }
    tt::PackArgs a{};
    a.njobs = njobs;
    int max_src = 1;
    long max_idx = 0;
    for (int j = 0; j < njobs; ++j) {
        const int ns = host_nsrc[j];
        if (host_mode[j] == 4) {           // plain cast: ns = groups of 8 elements
            if (ns < 1 || !host_src[3 * j] || ((uintptr_t)host_src[3 * j] & 15) || !host_out[j] || ((uintptr_t)host_out[j] & 15)) {
                set_error("tower_train_pack: job %d: cast needs >= 1 group and 16-byte aligned pointers", j);
                return DLDKD_EINVAL;
            }
            a.job[j].src[0] = host_src[3 * j];
            a.job[j].nsrc = ns;
            a.job[j].mode = 4;
            a.job[j].out = (bf16x8*)host_out[j];
            if (ns > max_idx) max_idx = ns;
            continue;
        }
        if (ns < 1 || ns > 3 || host_mode[j] < 0 || host_mode[j] > 3 || !host_out[j] || ((uintptr_t)host_out[j] & 15)) {
            set_error("tower_train_pack: job %d: 1..3 sources, mode 0..3, 16-byte aligned output", j);
            return DLDKD_EINVAL;
        }
        for (int c = 0; c < ns; ++c) {
            if (!host_src[3 * j + c]) { set_error("tower_train_pack: null weight"); return DLDKD_EINVAL; }
            a.job[j].src[c] = host_src[3 * j + c];
        }
        a.job[j].nsrc = ns;
        a.job[j].mode = host_mode[j];
        a.job[j].out = (bf16x8*)host_out[j];
        if (ns > max_src) max_src = ns;
    }
    if ((long)max_src * tt::kMatFrags * 64 > max_idx) max_idx = (long)max_src * tt::kMatFrags * 64;
    const bool want_lens = mask != nullptr && n_seq > 0;
    if (want_lens) {
        a.mask = mask; a.n_seq = n_seq; a.L = L; a.lens = lens;
        if ((long)n_seq * 64 > max_idx) max_idx = (long)n_seq * 64;
    }
    DLDKD_LAUNCH(tt::pack_kernel, dim3((unsigned)((max_idx + 255) / 256), (unsigned)(njobs + (want_lens ? 1 : 0))), dim3(256), 0, (hipStream_t)stream, a);
    return check_launch("tower_train_pack");
}

int dldkd_tower_train_pack(const float* const* host_src, const int* host_nsrc, const int* host_mode, void* const* host_out, int njobs,
                           void* stream) {
    return tower_train_pack_impl(host_src, host_nsrc, host_mode, host_out, njobs, nullptr, 0, 0, nullptr, stream);
}

int dldkd_tower_train_prepare(const float* const* host_src, const int* host_nsrc, const int* host_mode, void* const* host_out, int njobs,
                              const float* mask, int n_seq, int L, int32_t* lens, void* stream) {
    if (n_seq < 0 || L < 1 || (n_seq > 0 && (!mask || !lens))) { set_error("tower_train_prepare: bad mask arguments"); return DLDKD_EINVAL; }
    return tower_train_pack_impl(host_src, host_nsrc, host_mode, host_out, njobs, mask, n_seq, L, lens, stream);
}

int dldkd_tower_train_f1(const float* y0, const float* pos, int L, const float* gamma, const float* beta, float eps, float p_drop,
                         unsigned long long seed, unsigned long long offset, const unsigned long long* state, const void* wqkv_pack,
                         const float* bq, const float* bk, const float* bv, const unsigned char* flags, long M, void* h1d, void* xh1,
                         float* stats, void* qkv, void* relu_bits, void* stream) {
    if (M < 0 || L < 1 || tt::bad_p(p_drop)) { set_error("tower_train_f1: bad sizes"); return DLDKD_EINVAL; }
    if ((uintptr_t)relu_bits & 7) { set_error("tower_train_f1: relu_bits must be 8-byte aligned"); return DLDKD_EINVAL; }
    if (M == 0) return DLDKD_OK;
    if (!y0 || !pos || !gamma || !beta || !wqkv_pack || !bq || !bk || !bv || !h1d || !xh1 || !stats || !qkv) { set_error("tower_train_f1: null pointer"); return DLDKD_EINVAL; }
    if (((uintptr_t)y0 | (uintptr_t)pos | (uintptr_t)gamma | (uintptr_t)beta | (uintptr_t)bq | (uintptr_t)bk | (uintptr_t)bv | (uintptr_t)h1d |
         (uintptr_t)xh1 | (uintptr_t)qkv | (uintptr_t)wqkv_pack) & 15) { set_error("tower_train_f1: 16-byte alignment"); return DLDKD_EINVAL; }
    tt::F1Args a{y0, pos, L, gamma, beta, eps, tt::make_drop(p_drop, seed, offset, state), (const bf16x8*)wqkv_pack, {bq, bk, bv}, flags, M,
                 (tt::u16*)h1d, (tt::u16*)xh1, stats, (tt::u16*)qkv, (unsigned char*)relu_bits};
    DLDKD_LAUNCH(tt::f1_kernel, dim3(tt::grid_of(M)), dim3(64 * TT_WPB), 0, (hipStream_t)stream, a);
    return check_launch("tower_train_f1");
}

int dldkd_tower_train_f3(const void* ctx, const void* h1d, const void* wd_pack, const float* bd, float p_drop, unsigned long long seed,
                         unsigned long long offset, const unsigned long long* state, const float* gamma, const float* beta, float eps,
                         const void* wo_pack, const float* bo, const unsigned char* flags, long M, void* xh2, float* rstd2, void* h2_bf16,
                         float* h2_f32, float* g, void* stream) {
    if (M < 0 || tt::bad_p(p_drop)) { set_error("tower_train_f3: bad sizes"); return DLDKD_EINVAL; }
    if (M == 0) return DLDKD_OK;
    if (!ctx || !h1d || !wd_pack || !bd || !gamma || !beta || !xh2 || !rstd2 || (wo_pack && (!bo || !g))) { set_error("tower_train_f3: null pointer"); return DLDKD_EINVAL; }
    if (((uintptr_t)ctx | (uintptr_t)h1d | (uintptr_t)wd_pack | (uintptr_t)bd | (uintptr_t)gamma | (uintptr_t)beta | (uintptr_t)wo_pack |
         (uintptr_t)bo | (uintptr_t)xh2 | (uintptr_t)h2_bf16 | (uintptr_t)h2_f32 | (uintptr_t)g) & 15) { set_error("tower_train_f3: 16-byte alignment"); return DLDKD_EINVAL; }
    tt::F3Args a{(const tt::u16*)ctx, (const tt::u16*)h1d, (const bf16x8*)wd_pack, bd, tt::make_drop(p_drop, seed, offset, state), gamma, beta, eps,
                 (const bf16x8*)wo_pack, bo, flags, M, (tt::u16*)xh2, rstd2, (tt::u16*)h2_bf16, h2_f32, g};
    if (wo_pack != nullptr) {
        if (!h2_bf16) { set_error("tower_train_f3: the out mapping needs the bf16 h2 buffer"); return DLDKD_EINVAL; }
        DLDKD_LAUNCH(tt::f3_kernel<true>, dim3(tt::grid_of(M)), dim3(64 * TT_WPB), 0, (hipStream_t)stream, a);
    } else {
        if (!h2_f32) { set_error("tower_train_f3: without the out mapping h2 is written as fp32 rows"); return DLDKD_EINVAL; }
        DLDKD_LAUNCH(tt::f3_kernel<false>, dim3(tt::grid_of(M)), dim3(64 * TT_WPB), 0, (hipStream_t)stream, a);
    }
    return check_launch("tower_train_f3");
}

int dldkd_tower_train_b3(const float* dg, const void* wot_pack, const void* xh2, const float* rstd2, const float* gamma, float p_drop,
                         unsigned long long seed, unsigned long long offset, const unsigned long long* state, const void* wdt_pack,
                         const unsigned char* flags, long M, void* ddo, void* dctx, void* dres, float* dgamma, float* dbeta, void* dg_bf16,
                         void* dh2_bf16, void* stream) {
    if (M < 0 || tt::bad_p(p_drop)) { set_error("tower_train_b3: bad sizes"); return DLDKD_EINVAL; }
    if (M == 0) return DLDKD_OK;
    if (!dg || !xh2 || !rstd2 || !gamma || !wdt_pack || !ddo || !dctx || !dres) { set_error("tower_train_b3: null pointer"); return DLDKD_EINVAL; }
    const bool sums = dgamma != nullptr;
    if (sums ? !dbeta : (dbeta != nullptr || (wot_pack != nullptr && !dh2_bf16))) {
        set_error("tower_train_b3: dgamma and dbeta together (column sums in the kernel), or neither and dh2_bf16 under an out mapping");
        return DLDKD_EINVAL;
    }
    if (((uintptr_t)dg | (uintptr_t)wot_pack | (uintptr_t)xh2 | (uintptr_t)gamma | (uintptr_t)wdt_pack | (uintptr_t)ddo | (uintptr_t)dctx |
         (uintptr_t)dres | (uintptr_t)dg_bf16 | (uintptr_t)dh2_bf16) & 15) { set_error("tower_train_b3: 16-byte alignment"); return DLDKD_EINVAL; }
    tt::B3Args a{dg, (const bf16x8*)wot_pack, (const tt::u16*)xh2, rstd2, gamma, tt::make_drop(p_drop, seed, offset, state), (const bf16x8*)wdt_pack,
                 flags, M, (tt::u16*)ddo, (tt::u16*)dctx, (tt::u16*)dres, dgamma, dbeta, (tt::u16*)dg_bf16, (tt::u16*)dh2_bf16};
    if (sums) DLDKD_LAUNCH(tt::b3_kernel<true>, dim3(tt::grid_of(M)), dim3(64 * TT_WPB), 0, (hipStream_t)stream, a);
    else DLDKD_LAUNCH(tt::b3_kernel<false>, dim3(tt::grid_of(M)), dim3(64 * TT_WPB), 0, (hipStream_t)stream, a);
    return check_launch("tower_train_b3");
}

int dldkd_tower_train_b1(const void* dqkv, const void* dres, const void* wqkvt_pack, const void* xh1, const void* relu_bits,
                         const float* stats, const float* gamma, float p_drop, unsigned long long seed, unsigned long long offset,
                         const unsigned long long* state, const unsigned char* flags, long M, int relu_mask, float* dy0, float* dx1,
                         float* dgamma, float* dbeta, void* dz_bf16, void* dy_bf16, void* stream) {
    if (M < 0 || tt::bad_p(p_drop)) { set_error("tower_train_b1: bad sizes"); return DLDKD_EINVAL; }
    if (M == 0) return DLDKD_OK;
    if (!dqkv || !dres || !wqkvt_pack || !xh1 || !stats || !gamma || !dy0) { set_error("tower_train_b1: null pointer"); return DLDKD_EINVAL; }
    if (relu_mask && (!relu_bits || ((uintptr_t)relu_bits & 3))) { set_error("tower_train_b1: relu_mask needs the 4-byte aligned bit plane f1 wrote"); return DLDKD_EINVAL; }
    const bool sums = dgamma != nullptr;
    if (sums ? !dbeta : (dbeta != nullptr || !dz_bf16)) {
        set_error("tower_train_b1: dgamma and dbeta together (column sums in the kernel), or neither and dz_bf16");
        return DLDKD_EINVAL;
    }
    if (((uintptr_t)dqkv | (uintptr_t)dres | (uintptr_t)wqkvt_pack | (uintptr_t)xh1 | (uintptr_t)gamma | (uintptr_t)dy0 |
         (uintptr_t)dx1 | (uintptr_t)dz_bf16 | (uintptr_t)dy_bf16) & 15) { set_error("tower_train_b1: 16-byte alignment"); return DLDKD_EINVAL; }
    tt::B1Args a{(const tt::u16*)dqkv, (const tt::u16*)dres, (const bf16x8*)wqkvt_pack, (const tt::u16*)xh1, (const unsigned char*)relu_bits, stats, gamma,
                 tt::make_drop(p_drop, seed, offset, state), flags, M, relu_mask, dy0, dx1, dgamma, dbeta, (tt::u16*)dz_bf16, (tt::u16*)dy_bf16};
    const dim3 grid(tt::grid_of(M));
    hipStream_t st = (hipStream_t)stream;
    if (dx1 != nullptr) {
        if (sums) DLDKD_LAUNCH((tt::b1_kernel<true, true>), grid, dim3(64 * TT_WPB), 0, st, a);
        else DLDKD_LAUNCH((tt::b1_kernel<true, false>), grid, dim3(64 * TT_WPB), 0, st, a);
    } else {
        if (sums) DLDKD_LAUNCH((tt::b1_kernel<false, true>), grid, dim3(64 * TT_WPB), 0, st, a);
        else DLDKD_LAUNCH((tt::b1_kernel<false, false>), grid, dim3(64 * TT_WPB), 0, st, a);
    }
    return check_launch("tower_train_b1");
}

int dldkd_colsum_bf16(const void* x, int ld, int c0, int N, long M, const unsigned char* flags, float* out, void* stream) {
    if (M < 0 || N < 1 || ld < 1 || c0 < 0 || c0 + N > ld) { set_error("colsum_bf16: bad sizes"); return DLDKD_EINVAL; }
    if (M == 0) return DLDKD_OK;
    if (!x || !out) { set_error("colsum_bf16: null pointer"); return DLDKD_EINVAL; }
    const int slabs = (int)((M + 511) / 512);
    const int rows_per_slab = (int)(((M + slabs - 1) / slabs + 127) / 128 * 128);
    DLDKD_LAUNCH(tt::colsum16_kernel, dim3((unsigned)((N + 63) / 64), (unsigned)((M + rows_per_slab - 1) / rows_per_slab)), dim3(256), 0,
                 (hipStream_t)stream, (const tt::u16*)x, ld, c0, N, M, flags, out, rows_per_slab);
    return check_launch("colsum_bf16");
}

}  // extern "C"
