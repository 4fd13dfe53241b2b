// Shared device helpers for the gfx950 kernels (wave64, MFMA, bf16).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/dldkd_hip.h"

namespace dldkd {

constexpr int kHidden = DLDKD_HIDDEN;   // 384 = 24 k-steps of 16 (mfma 32x32x16) = 12 of 32 (16x16x32)
constexpr int kWave = 64;

typedef short bf16x8 __attribute__((ext_vector_type(8)));   // 8 bf16 = one MFMA A/B fragment (4 VGPRs)
typedef float f32x16 __attribute__((ext_vector_type(16)));  // 32x32 accumulator fragment
typedef float f32x4 __attribute__((ext_vector_type(4)));    // 16x16 accumulator fragment / 16-byte fp32 vector

// fp32 -> bf16 bits, round-to-nearest-even, NaN preserving (plain cast -> v_cvt_pk_bf16_f32 on gfx950,
// MI355X_MICROARCH.md "Correctness boundaries").
__device__ __forceinline__ unsigned short f32_to_bf16_bits(float x) {
    return __builtin_bit_cast(unsigned short, static_cast<__bf16>(x));
}
__device__ __forceinline__ float bf16_bits_to_f32(unsigned short b) {
    return __builtin_bit_cast(float, static_cast<unsigned int>(b) << 16);
}

// ---- h16: the 16-bit operand format of the EVAL-path towers (input projection K4 / K4b, fused tower K5, the resident feature
// table and the h0 rows between K4b and K5) is IEEE fp16, not bf16.  Same MFMA rate (v_mfma_f32_32x32x16_f16), three more mantissa
// bits: rounding an operand costs 2^-12 relative instead of 2^-9.  Measured on the trained TVR-dims model (tools/rk_stage_probe.py,
// tools/emu_operand_format.py; profiles/r05/): with bf16 operands the K = 3072 input projection alone moved the fused scores by
// 1.8e-4 (mean) and the whole throughput path by 2.4e-4 - 13 of 8,192 queries crossed the R@100 cut, a coin flip away from
// north_star's +-0.1 gate; with fp16 operands the path sits at 7.7e-5, the bf16 SCORER's own rounding (K1 keeps bf16 operands:
// BASELINE's named dtype, rows L2-normalised in fp32 first).  Range: every operand of these kernels is a LayerNorm output, a
// weight, a probability or an L2-normalised feature (|x| <= ~20 measured, fp16 holds 65504; below 6e-5 precision tapers off
// where the values no longer matter).  Training keeps bf16 (gradients need the exponent range).  The containers stay `short`
// vectors (bf16x8): only the conversions and the MFMA opcode differ.
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16_ __attribute__((ext_vector_type(16)));
#ifndef DLDKD_H16_IS_BF16            // (make H16_BF16=1: the bf16 form of the same kernels, for same-box A/B runs only - tools/r05_ab_h16.sh)
__device__ __forceinline__ unsigned short f32_to_h16_bits(float x) {       // round to nearest even (v_cvt_f16_f32 / v_cvt_pk_f16_f32)
    return __builtin_bit_cast(unsigned short, static_cast<_Float16>(x));
}
__device__ __forceinline__ float h16_bits_to_f32(unsigned short b) { return static_cast<float>(__builtin_bit_cast(_Float16, b)); }
#define DLDKD_H16_MFMA32 "v_mfma_f32_32x32x16_f16"
#define DLDKD_H16_CVT_PK "v_cvt_pk_f16_f32"
template <class A, class B>
__device__ __forceinline__ f32x16_ h16_mfma32(const A& a, const B& b, const f32x16_& c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}
#else
typedef short s16x8_ __attribute__((ext_vector_type(8)));
__device__ __forceinline__ unsigned short f32_to_h16_bits(float x) { return __builtin_bit_cast(unsigned short, static_cast<__bf16>(x)); }
__device__ __forceinline__ float h16_bits_to_f32(unsigned short b) { return __builtin_bit_cast(float, static_cast<unsigned int>(b) << 16); }
#define DLDKD_H16_MFMA32 "v_mfma_f32_32x32x16_bf16"
#define DLDKD_H16_CVT_PK "v_cvt_pk_bf16_f32"
template <class A, class B>
__device__ __forceinline__ f32x16_ h16_mfma32(const A& a, const B& b, const f32x16_& c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(s16x8_, a), __builtin_bit_cast(s16x8_, b), c, 0, 0, 0);
}
#endif

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}

// fused = w0 * s0 + w1 * s1 (reference method/eval.py:254), written ONE way - fma(w0, a, w1 * b) - wherever it is computed (the
// finish kernel that writes the fused matrix and the rank-from-partials kernel that never writes it), so that both see
// bit-identical fused scores whatever contraction the compiler would pick per call site.
__device__ __forceinline__ float fuse2(float w0, float a, float w1, float b) { return __builtin_fmaf(w0, a, w1 * b); }

// Philox4x32-10 (Salmon et al., SC'11): the counter-based generator of every dropout mask (train_f32.hip, attention_train.hip)
__device__ __forceinline__ void philox4x32_10(unsigned c0, unsigned c1, unsigned c2, unsigned c3, unsigned k0, unsigned k1,
                                              unsigned (&out)[4]) {
#ifndef DLDKD_PHILOX_ROUNDS
#define DLDKD_PHILOX_ROUNDS 10     // (diagnostic builds: make PHILOX_ROUNDS=7 measures what the generator's last three rounds cost)
#endif
#pragma unroll
    for (int r = 0; r < DLDKD_PHILOX_ROUNDS; ++r) {
        const unsigned long long p0 = 0xD2511F53ull * c0, p1 = 0xCD9E8D57ull * c2;
        const unsigned n0 = (unsigned)(p1 >> 32) ^ c1 ^ k0, n2 = (unsigned)(p0 >> 32) ^ c3 ^ k1;
        c1 = (unsigned)p1; c3 = (unsigned)p0; c0 = n0; c2 = n2;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// LDS-DMA: one 16-byte piece per lane, 1 KiB per wave-instruction.  The LDS destination is the
// wave-uniform base + lane*16 (cdna_hip_programming.md section 5 caveat); the global source is per lane.
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void global_cvoid;
__device__ __forceinline__ void glds16(const void* gsrc_lane, void* lds_base_uniform) {
    __builtin_amdgcn_global_load_lds(
        reinterpret_cast<global_cvoid*>(reinterpret_cast<uintptr_t>(gsrc_lane)),
        reinterpret_cast<lds_void*>(static_cast<uint32_t>(reinterpret_cast<uintptr_t>(lds_base_uniform))),
        16, 0, 0);
}

// The same with an instruction offset: OFF moves the global AND the LDS address, so a run of pieces shares one address register pair
// and one M0 value (no 64-bit add and no M0 arithmetic per piece)
template <int OFF>
__device__ __forceinline__ void glds16_off(const void* gsrc_lane, void* lds_base_uniform) {
    __builtin_amdgcn_global_load_lds(
        reinterpret_cast<global_cvoid*>(reinterpret_cast<uintptr_t>(gsrc_lane)),
        reinterpret_cast<lds_void*>(static_cast<uint32_t>(reinterpret_cast<uintptr_t>(lds_base_uniform))),
        16, OFF, 0);
}

// XCD-aware order of a 3-D tile grid (cdna_hip_programming.md section 5 "XCD swizzle must be bijective", T1): the workgroups with
// equal launch id % 8 share an XCD (round-robin placement: a speed assumption only) and are handed a contiguous run of tile ids,
// x fastest.  Returns the tile this workgroup computes in place of blockIdx.
struct Tile3 { int x, y, z; };
__device__ __forceinline__ Tile3 xcd_tile_order() {
    const unsigned gx = gridDim.x, gy = gridDim.y;
    const unsigned nwg = gx * gy * gridDim.z;
    const unsigned orig = blockIdx.x + gx * (blockIdx.y + gy * blockIdx.z);
    unsigned t = orig;
    if (nwg >= 16) {
        const unsigned q = nwg >> 3, r = nwg & 7, xcd = orig & 7;
        t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
    }
    Tile3 o;
    o.x = (int)(t % gx);
    const unsigned yz = t / gx;
    o.y = (int)(yz % gy);
    o.z = (int)(yz / gy);
    return o;
}

// Epilogue of the 128x128-block / 64x64-per-wave MFMA GEMMs: C = act(alpha * acc + bias).  The 32x32 accumulator layout puts
// 32 consecutive COLUMNS of one row on 32 lanes (128-byte row segments as 4-byte stores).  When the wave's 64 columns are
// all in range and rows are 16-byte aligned, each 32 x 64 half is parked in the wave's private LDS region (the operand
// tiles are dead after the k-loop's last barrier) and written back as float4: 16 lanes cover 256 contiguous bytes of a row.
// Ragged edges keep the element-wise path.  Split-K launches write their partial plane the same way (C then points into
// the split-K workspace, one plane per blockIdx.z) and a reduce pass sums the planes.
template <typename Args>
__device__ __forceinline__ void gemm_store_tile(const f32x16 (&acc)[2][2], const Args& p, int m0, int n0, int wm, int wn, int lane,
                                                float* stg /* 32 x 72 floats, private to the wave */) {
    constexpr int SP = 72;
    const bool fast = !(p.ldc & 3) && !((uintptr_t)p.C & 15) && n0 + wn + 64 <= p.N;   // split-K planes are plain stores too
    if (fast) {
        float bias[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) bias[j] = p.bias ? p.bias[n0 + wn + 32 * j + (lane & 31)] : 0.f;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float v = acc[i][j][r] * p.alpha + bias[j];
                    if (p.relu) v = fmaxf(v, 0.f);
                    stg[((r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * SP + 32 * j + (lane & 31)] = v;
                }
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                const int idx = lane + 64 * it, rl = idx >> 4, c4 = idx & 15;
                const int m = m0 + wm + 32 * i + rl;
                const f32x4 v = *reinterpret_cast<const f32x4*>(stg + rl * SP + 4 * c4);
                if (m < p.M) *reinterpret_cast<f32x4*>(p.C + (size_t)m * p.ldc + n0 + wn + 4 * c4) = v;
            }
        }
        return;
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int n = n0 + wn + 32 * j + (lane & 31);
        if (n >= p.N) continue;
        const float bias = p.bias ? p.bias[n] : 0.f;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (m < p.M) {
                    float v = acc[i][j][r] * p.alpha + bias;
                    if (p.relu) v = fmaxf(v, 0.f);
                    p.C[(size_t)m * p.ldc + n] = v;
                }
            }
    }
}

// Epilogue of the input projection's backward pass in training (LinearLayer: LayerNorm -> Dropout -> Linear -> ReLU on RAW
// features, method/model_components.py:294-312).  The raw features need no gradient, so of the Linear's input gradient
// dz' = dY W only LayerNorm's parameter gradients are wanted:
//     dz = dz' (.) keep / (1 - p),   dgamma[k] = sum_m dz[m, k] xhat[m, k],   dbeta[k] = sum_m dz[m, k].
// The tile of dz' stays in the accumulators: each lane multiplies its 64 values by the keep byte and the normalised feature
// (x read once, coalesced: 32 lanes = 128 B of a row) and sums over its rows; lane halves, then the two waves that share the
// columns, are combined and the workgroup writes ONE partial row per sum (part[row tile][k]) - no dz' in memory (201 MB written
// and read back at the TVR batch), no LayerNorm backward pass over x and dz'.  A column sum over the row tiles finishes.
struct LnGradArgs {
    const float* x;              // [M, ldx] raw features
    const unsigned char* keep;   // [M, ldx] dropout keep bytes or null
    const float* mean;           // [M]
    const float* rstd;           // [M]
    float* part_g;               // [row tiles][ldp]
    float* part_b;               // [row tiles][ldp]
    int ldx;
    float keep_scale;
    int ldp;                     // row stride of the partial sums (N, or 2 N with part_b = part_g + N: ONE column sum finishes both)
};

template <typename Args>
__device__ __forceinline__ void gemm_lngrad_tile(const f32x16 (&acc)[2][2], const Args& p, const LnGradArgs& la, int m0, int n0, int wm,
                                                 int wn, int lane, int wave, float* lds_f) {
    // Each 32 x 64 half of the wave's tile is parked in the wave's private LDS region (like gemm_store_tile) and read back
    // row-wise: lane = (row rl = lane >> 4 (+ 4 per pass), 4 consecutive columns c4 = lane & 15), so x is read as float4 (16 lanes =
    // 256 contiguous bytes of a row) and the keep bytes as one dword - the first version read them in the accumulator layout
    // (4-byte and 1-byte accesses, 128 VMEM instructions per lane) and spent 165 us of 280 in this epilogue.
    constexpr int SP = 72;
    float* stg = lds_f + wave * (32 * SP);
    float* red = lds_f + 4 * (32 * SP);                      // [4 waves][2 sums][64 columns] behind the staging regions
    const int rl0 = lane >> 4, c4 = lane & 15;
    const int col = n0 + wn + 4 * c4;
    const bool vec = !(la.ldx & 3) && !((uintptr_t)la.x & 15) && col + 3 < p.N && (la.keep == nullptr || !((uintptr_t)la.keep & 3));
    f32x4 sg = {0.f, 0.f, 0.f, 0.f}, sb = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) stg[((r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * SP + 32 * j + (lane & 31)] = acc[i][j][r];
        f32x4 xv[8];
        unsigned kw[8];
        float mu[8], rs[8];
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int row = m0 + wm + 32 * i + rl0 + 4 * it;
            const int rowc = row < p.M ? row : p.M - 1;
            const size_t at = (size_t)rowc * la.ldx + col;
            mu[it] = la.mean[rowc];
            rs[it] = row < p.M ? la.rstd[rowc] : 0.f;        // rows past the end: xhat = 0 and (below) dz = 0
            kw[it] = 0x01010101u;
            xv[it] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (rs[it] == 0.f) {
                // rstd == 0 marks a row of the padding (dldkd_layernorm_dropout_bf16 with a row mask; a real row has rstd > 0): its
                // features are not read and it contributes nothing
                kw[it] = 0u;
            } else if (vec) {
                xv[it] = *reinterpret_cast<const f32x4*>(la.x + at);
                if (la.keep != nullptr) kw[it] = *reinterpret_cast<const unsigned*>(la.keep + at);
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const bool ok = col + e < p.N;
                    xv[it][e] = ok ? la.x[at + e] : 0.f;
                    if (la.keep != nullptr) kw[it] = (kw[it] & ~(0xffu << (8 * e))) | ((ok && la.keep[at + e] ? 1u : 0u) << (8 * e));
                    else if (!ok) kw[it] &= ~(0xffu << (8 * e));
                }
            }
        }
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const bool rok = m0 + wm + 32 * i + rl0 + 4 * it < p.M;
            const f32x4 a4 = *reinterpret_cast<const f32x4*>(stg + (rl0 + 4 * it) * SP + 4 * c4);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const bool on = rok && ((kw[it] >> (8 * e)) & 0xffu);
                const float dz = on ? a4[e] * la.keep_scale : 0.f;
                sg[e] += on ? dz * ((xv[it][e] - mu[it]) * rs[it]) : 0.f;
                sb[e] += dz;
            }
        }
    }
    // lanes l, l + 16, l + 32, l + 48 hold the same four columns (rows rl0 + 4 it): combine, then the wave that shares the columns
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        sg[e] += __shfl_xor(sg[e], 16); sg[e] += __shfl_xor(sg[e], 32);
        sb[e] += __shfl_xor(sb[e], 16); sb[e] += __shfl_xor(sb[e], 32);
    }
    float* mine = red + wave * 128;
    if (lane < 16) {
        *reinterpret_cast<f32x4*>(mine + 4 * c4) = sg;
        *reinterpret_cast<f32x4*>(mine + 64 + 4 * c4) = sb;
    }
    __syncthreads();
    if (wm == 0 && lane < 16) {
        const float* other = red + (wave + 2) * 128;         // wave + 2: same columns, rows 64..127
        const f32x4 og = *reinterpret_cast<const f32x4*>(other + 4 * c4), ob = *reinterpret_cast<const f32x4*>(other + 64 + 4 * c4);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if (col + e >= p.N) continue;
            la.part_g[(size_t)(m0 / 128) * la.ldp + col + e] = sg[e] + og[e];
            la.part_b[(size_t)(m0 / 128) * la.ldp + col + e] = sb[e] + ob[e];
        }
    }
}

// ---- training simpool epilogue (simpool_train.hip): the batched GEMM S_v = G_v Q^T (one video per blockIdx.z, M = L <= 128 clips on
// the tile rows, queries on the columns) ends in a key-clip max-pool instead of a store, so the (Nq, Nv, L) clip tensor of
// get_sim_scores / get_unnormalized_sim_scores (reference method/model.py:307-350) is never written:
//   pooled_raw[n, v] = max_{l < len_v} S[l, n]                      arg_raw = its clip            (model.py:344-349)
//   pooled_cos[n, v] = rq[n] * max_{l < len_v} S[l, n] * rg[v, l]   arg_cos = its clip            (model.py:318-327)
//   clip_pos[n, l]   = S[l, n] * rg[v, l] * rq[n]  (l < len_v, else -1e10)  only where labels[n] == v: the one column of
//                      the clip-level tensor that compute_kl_loss reads (model.py:183-197)
// rq / rg = 1 / max(|row|, 1e-12) (F.normalize): cos = <q, g> / (|q| |g|) from the SAME raw product, half the GEMM work.
struct PoolArgs {
    const float* rg;        // [nv * L]
    const float* rq;        // [nq]
    const int32_t* lens;    // [nv]
    const int32_t* labels;  // [nq]
    float* pooled_raw;      // [nq, nv]
    float* pooled_cos;      // [nq, nv]
    int32_t* arg_raw;       // [nq, nv]
    int32_t* arg_cos;       // [nq, nv]
    float* clip_pos;        // [nq, L] or null
    int nv, L;
};

// Does candidate (v, i) replace the running best (bv, bi) of a max-pool with arg-max?  torch.max semantics (model.py:327,349):
// the first maximum wins, and a NaN wins over everything (the first NaN): a diverged step then shows as a NaN loss, as in the
// reference - with `s > best` alone an all-NaN column left the sentinel index in arg_* and the backward gather read 2^31 rows
// past the gallery.  bi starts at 0x7fffffff with bv = -inf, so the first valid clip is always taken.
__device__ __forceinline__ bool pool_better(float v, int i, float bv, int bi) {
    const bool vn = v != v, bn = bv != bv;
    if (vn || bn) return vn && (!bn || i < bi);
    return v > bv || (v == bv && i < bi);
}

// acc: the 128 x 128 block's accumulators in the layout of the three tiled GEMMs (wave (wm, wn) owns rows wm..wm+63, columns
// wn..wn+63 as 2 x 2 tiles of 32 x 32: column = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)).
// scratch: >= 4 * 64 * 4 floats of LDS, free after the k-loop's last barrier.
template <typename Args>
__device__ __forceinline__ void gemm_pool_tile(const f32x16 (&acc)[2][2], const Args& p, const PoolArgs& pa, int v, int n0, int wm,
                                               int wn, int lane, int wave, float* scratch) {
    const int len = pa.lens[v];
    const int half = lane >> 5;
    float braw[2] = {-INFINITY, -INFINITY}, bcos[2] = {-INFINITY, -INFINITY};
    int iraw[2] = {0x7fffffff, 0x7fffffff}, icos[2] = {0x7fffffff, 0x7fffffff};
    const float* rgv = pa.rg + (size_t)v * pa.L;
    int lab[2];
    float rqn[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int n = n0 + wn + 32 * j + (lane & 31);
        lab[j] = n < p.N ? pa.labels[n] : -1;
        rqn[j] = n < p.N ? pa.rq[n] : 0.f;
    }
    const bool any_pos = pa.clip_pos != nullptr && __any(lab[0] == v || lab[1] == v);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = wm + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * half;     // increasing in (i, r): first maximum wins
            const bool valid = row < len;
            const float rgl = valid ? rgv[row] : 0.f;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const float s = acc[i][j][r] * p.alpha;
                const float c = s * rgl;
                if (valid && pool_better(s, row, braw[j], iraw[j])) { braw[j] = s; iraw[j] = row; }
                if (valid && pool_better(c, row, bcos[j], icos[j])) { bcos[j] = c; icos[j] = row; }
                if (any_pos && lab[j] == v && row < pa.L)
                    pa.clip_pos[(size_t)(n0 + wn + 32 * j + (lane & 31)) * pa.L + row] = valid ? c * rqn[j] : -1e10f;
            }
        }
    // other half of the wave (rows + 4), then the wave that holds the other 64 rows
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const float ob = __shfl_xor(braw[j], 32), oc = __shfl_xor(bcos[j], 32);
        const int oi = __shfl_xor(iraw[j], 32), oj = __shfl_xor(icos[j], 32);
        if (pool_better(ob, oi, braw[j], iraw[j])) { braw[j] = ob; iraw[j] = oi; }
        if (pool_better(oc, oj, bcos[j], icos[j])) { bcos[j] = oc; icos[j] = oj; }
    }
    float* mine = scratch + wave * 256;
    if (half == 0) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            float* e = mine + (32 * j + lane) * 4;
            e[0] = braw[j]; e[1] = __int_as_float(iraw[j]); e[2] = bcos[j]; e[3] = __int_as_float(icos[j]);
        }
    }
    __syncthreads();
    if (wm == 0 && half == 0) {
        const float* other = scratch + (wave + 2) * 256;       // wave + 2: same columns, rows 64..127
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int n = n0 + wn + 32 * j + lane;
            if (n >= p.N) continue;
            const float* e = other + (32 * j + lane) * 4;
            const float ob = e[0], oc = e[2];
            const int oi = __float_as_int(e[1]), oj = __float_as_int(e[3]);
            if (pool_better(ob, oi, braw[j], iraw[j])) { braw[j] = ob; iraw[j] = oi; }
            if (pool_better(oc, oj, bcos[j], icos[j])) { bcos[j] = oc; icos[j] = oj; }
            const size_t o = (size_t)n * pa.nv + v;
            const bool none = len <= 0;                          // no valid clip: the reference's masked maximum (-1e10, clip 0)
            pa.pooled_raw[o] = none ? -1e10f : braw[j];
            pa.pooled_cos[o] = none ? -1e10f : bcos[j] * rqn[j];
            pa.arg_raw[o] = none ? 0 : min(iraw[j], len - 1);    // always a clip of the video (the backward pass gathers it)
            pa.arg_cos[o] = none ? 0 : min(icos[j], len - 1);
        }
    }
}

// split-K plans of the three tiled GEMMs (number of k-slices, 1 = none), shared with dldkd_gemm_workspace_bytes
int gemm_f32_split_plan(int M, int N, int K, int a_kmajor, int b_kmajor, int* k_tiles_per_split);
int gemm_f32x3_split_plan(int M, int N, int K, int a_kmajor, int b_kmajor, int* k_tiles_per_split);
int gemm_bf16_split_plan(int M, int N, int K, int a_kmajor, int b_kmajor, int* k_tiles_per_split);
// pooled batched GEMMs (gemm_f32x3.hip / gemm_bf16.hip): g (nv, L, D), q (nq, D) -> the PoolArgs outputs
int launch_simpool_pool_x3(const float* g, const float* q, int nv, int L, int nq, int D, const PoolArgs& pa, void* stream, int planes = 3);
int launch_simpool_pool_bf16_dma(const void* g16, const void* q16, int nv, int L, int nq, int D, const PoolArgs& pa, void* stream,
                                 long g_plane_stride = 0, long q_plane_stride = 0);
// dz' = dy W with the LayerNorm-parameter-gradient epilogue (gemm_lngrad_tile), per precision mode
int launch_linear_lngrad_bf16(const float* dy, const float* W, long M, int N, int K, const LnGradArgs& la, void* stream,
                              const unsigned char* row_flags = nullptr);
int launch_linear_lngrad_x3(const float* dy, const float* W, long M, int N, int K, const LnGradArgs& la, void* stream,
                            const unsigned char* row_flags = nullptr);
int launch_simpool_pool_bf16(const float* g, const float* q, int nv, int L, int nq, int D, const PoolArgs& pa, void* stream);
int launch_splitk_reduce(const float* ws, float* out, int split, long n, hipStream_t s);
// gemm_bf16_tn.hip: C = A^T B over bf16 rows (LDS-DMA tiles, transposed LDS reads) - the weight-gradient products of the training step
bool gemm_bf16_tn_enabled();
bool gemm_bf16_tn_ok(int M, int N, long R, int lda, int ldb);
size_t gemm_bf16_tn_planes_bytes(int M, int N, long R, int dual);
int launch_gemm_bf16_tn(const void* A, const void* B, float* C, int M, int N, long R, int lda, int ldb, int dual, void* planes,
                        const unsigned char* rflags, float* a_colsum, hipStream_t stream);
int launch_gemm_bf16_tn_group(const void* const* A, const int* lda, const int* acol, const void* const* B, int n_blocks, long R, float* dW,
                              void* planes, const unsigned char* rflags, float* a_colsum, hipStream_t stream);
// Every kernel launch of the library: drop whatever error another library left in the runtime's sticky per-thread slot
// (torch's caching allocator probes the runtime while a hipGraph is being captured and leaves hipErrorInvalidValue behind),
// so that check_launch() reports THIS launch only.
#define DLDKD_LAUNCH(...)                \
    do {                                 \
        (void)hipGetLastError();         \
        hipLaunchKernelGGL(__VA_ARGS__); \
    } while (0)

void set_error(const char* fmt, ...);
int check_launch(const char* what);

}  // namespace dldkd
