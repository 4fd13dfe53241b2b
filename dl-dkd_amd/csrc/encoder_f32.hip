// Encoder-tower forward kernels, fp32 (parity grade).  Together with gemm_f32 they implement
// DLDKD.encode_input / encode_context / encode_query (reference method/model.py:199-258) without any
// library call:
//   layernorm_f32      LayerNorm(x [+ add]) : LinearLayer.LayerNorm (model_components.py:308), the
//                      "+ position rows, LayerNorm" of TrainablePositionalEncoding (:277-284) and the
//                      "+ residual, LayerNorm" of BertSelfOutput (:446-450)
//   attention_fwd_f32  BertSelfAttention.forward (:398-436): softmax(QK^T/sqrt(96) + (1-mask)*-1e4) V,
//                      4 heads x 96, whole sequence (L <= 128) resident in LDS, no L x L matrix in HBM
//   modpool_fwd_f32    get_modularized_queries (model.py:245-258)
#include "common.hpp"

namespace dldkd {

// ----------------------------------------------------------------------------------------------
// LayerNorm over the last dim (eps inside the sqrt, biased variance: nn.LayerNorm).  One wave per row;
// the row is cached in registers (two-pass mean / variance like ATen, not E[x^2]-mean^2).
// add_mod == 0: add[row]; add_mod > 0: add[row % add_mod] (position table); add == NULL: none.
// ----------------------------------------------------------------------------------------------
template <int MAXV>
__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ x, const float* __restrict__ add,
                                                        int add_mod, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, float* __restrict__ out, long M,
                                                        int D, float eps, unsigned char* __restrict__ keep, unsigned thresh,
                                                        float dscale, unsigned long long seed, unsigned long long off,
                                                        const unsigned long long* __restrict__ state,
                                                        unsigned short* __restrict__ out16, float* __restrict__ stats,
                                                        const float* __restrict__ row_mask, unsigned char* __restrict__ gflags,
                                                        const unsigned char* __restrict__ gin, unsigned short* __restrict__ planes) {
    // planes != null: the row also as TWO bf16 planes [2][M][D] (h = bf16(o), m = bf16(o - h)): the operand form of the two-plane
    // GEMM dldkd_gemm_bf16_nt16_planes ("mixed" training precision); plane 0 is the bf16 row the backward pass keeps
    // out16 != null: the row is written as bf16 (round to nearest even) instead of fp32 - the operand form of the bf16 GEMMs that
    // consume it (dldkd_gemm_bf16_mixed); stats != null: mean -> stats[row], rstd -> stats[M + row] (kept for the backward pass)
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    // dropout is on when a keep mask is asked for OR the threshold is non-zero (p > 0 without a mask buffer: the masks are
    // Philox4x32-10 on the flat element index and whoever needs a bit later draws it again - gemm_bf16.hip, inproj_bwd_final_kernel)
    const bool drop = keep != nullptr || thresh != 0u;
    if (drop && state != nullptr) { seed = state[0]; off += state[1]; }
    const int nv = D >> 2;
    if (row_mask != nullptr || gin != nullptr) {
        // rows of the padding (row_mask[row] == 0: clips past a video's length in a padded batch) are not read: their output row
        // is zeros (keep bytes 0, statistics 0) - no loss term depends on them and their gradients are exactly zero, so the
        // GEMMs behind this kernel may skip them (gflags[g] = 1 when ANY of the 32 rows of group g is valid - the wave of the
        // group's first row reads the 32 mask values; a mask need not be a prefix: ADVICE r03 - and M % 32 == 0, entry points)
        // (gin: the group flags an earlier kernel of the tower wrote - validity per 32-row group instead of per row)
        const bool valid = row_mask != nullptr ? row_mask[row] > 0.f : gin[row >> 5] != 0;
        if (gflags != nullptr && (row & 31) == 0) {
            const bool any = row_mask != nullptr ? __ballot(lane < 32 && row_mask[row + (lane & 31)] > 0.f) != 0ull : valid;
            if (lane == 0) gflags[row >> 5] = any ? 1 : 0;
        }
        if (!valid) {
            if (stats != nullptr && lane == 0) { stats[row] = 0.f; stats[M + row] = 0.f; }
            for (int c = lane; c < nv; c += 64) {
                if (out16 != nullptr) reinterpret_cast<uint2*>(out16 + row * D)[c] = uint2{0u, 0u};
                if (out != nullptr) reinterpret_cast<f32x4*>(out + row * D)[c] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (planes != nullptr) {
                    reinterpret_cast<uint2*>(planes + row * D)[c] = uint2{0u, 0u};
                    reinterpret_cast<uint2*>(planes + (M + row) * D)[c] = uint2{0u, 0u};
                }
                if (keep != nullptr) reinterpret_cast<uchar4*>(keep + row * D)[c] = uchar4{0, 0, 0, 0};
            }
            return;
        }
    }
    const f32x4* xr = reinterpret_cast<const f32x4*>(x + row * D);
    const f32x4* ar = add ? reinterpret_cast<const f32x4*>(add + (add_mod > 0 ? row % add_mod : row) * D) : nullptr;
    f32x4 v[MAXV];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int c = lane + 64 * i;
        v[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (c < nv) {
            v[i] = xr[c];
            if (ar) { const f32x4 a = ar[c]; v[i] += a; }
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
    if (stats != nullptr && lane == 0) { stats[row] = mean; stats[M + row] = rstd; }
    f32x4* orow = reinterpret_cast<f32x4*>(out + row * D);
    uint2* orow16 = reinterpret_cast<uint2*>(out16 + row * D);
    const f32x4* g4 = reinterpret_cast<const f32x4*>(gamma);
    const f32x4* b4 = reinterpret_cast<const f32x4*>(beta);
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int c = lane + 64 * i;
        if (c < nv) {
            const f32x4 g = g4[c], b = b4[c];
            f32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = (v[i][e] - mean) * rstd * g[e] + b[e];
            if (drop) {                 // fused inverted dropout on the normalised row: the masks dldkd_dropout_fwd_f32 would draw
                const unsigned long long ctr = off + (unsigned long long)(row * nv + c);
                unsigned rnd[4];
                philox4x32_10((unsigned)ctr, (unsigned)(ctr >> 32), 0u, 0u, (unsigned)seed, (unsigned)(seed >> 32), rnd);
                uchar4 k;
                k.x = rnd[0] >= thresh; k.y = rnd[1] >= thresh; k.z = rnd[2] >= thresh; k.w = rnd[3] >= thresh;
                o[0] = k.x ? o[0] * dscale : 0.f; o[1] = k.y ? o[1] * dscale : 0.f;
                o[2] = k.z ? o[2] * dscale : 0.f; o[3] = k.w ? o[3] * dscale : 0.f;
                if (keep != nullptr) reinterpret_cast<uchar4*>(keep + row * D)[c] = k;
            }
            if (out16 != nullptr) {
                uint2 pk;
                pk.x = (unsigned)f32_to_bf16_bits(o[0]) | ((unsigned)f32_to_bf16_bits(o[1]) << 16);
                pk.y = (unsigned)f32_to_bf16_bits(o[2]) | ((unsigned)f32_to_bf16_bits(o[3]) << 16);
                orow16[c] = pk;
            }
            if (out != nullptr) orow[c] = o;           // (both: "mixed" training keeps fp32 rows for its forward GEMM and bf16 rows for the backward)
            if (planes != nullptr) {
                unsigned short hb[4], mb[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    hb[e] = f32_to_bf16_bits(o[e]);
                    mb[e] = f32_to_bf16_bits(o[e] - bf16_bits_to_f32(hb[e]));
                }
                reinterpret_cast<uint2*>(planes + row * D)[c] = uint2{(unsigned)hb[0] | ((unsigned)hb[1] << 16), (unsigned)hb[2] | ((unsigned)hb[3] << 16)};
                reinterpret_cast<uint2*>(planes + (M + row) * D)[c] = uint2{(unsigned)mb[0] | ((unsigned)mb[1] << 16), (unsigned)mb[2] | ((unsigned)mb[3] << 16)};
            }
        }
    }
}

// ----------------------------------------------------------------------------------------------
// The LayerNorm (+ inverted dropout) of BOTH branches' input projections over the SAME raw rows in one pass (round 6): the two
// video towers of the training step normalise the same (Nv, L, Dv) student features with their own gamma / beta and their own
// dropout draws (method/model.py:229-243 -> LinearLayer.forward, model_components.py:305-310, once per branch).  Two launches of
// layernorm_kernel on two streams read the 201-MB TVR batch twice, beside each other: 204 + 206 us against 87 alone
// (profiles/r05/step_timeline_bf16_graph.txt).  Here a row is read once, its mean / rstd taken once, and written twice as bf16
// rows; masks = what two dldkd_layernorm_dropout_bf16 calls at (seed, off0) and (seed, off1) draw, bit for bit.
// ----------------------------------------------------------------------------------------------
template <int MAXV>
__global__ __launch_bounds__(256) void layernorm_dual_bf16_kernel(const float* __restrict__ x, const float* __restrict__ gamma0,
                                                                  const float* __restrict__ beta0, const float* __restrict__ gamma1,
                                                                  const float* __restrict__ beta1, unsigned short* __restrict__ out0,
                                                                  unsigned short* __restrict__ out1, float* __restrict__ stats, long M,
                                                                  int D, float eps, unsigned thresh, float dscale, unsigned long long seed,
                                                                  unsigned long long off0, unsigned long long off1,
                                                                  const unsigned long long* __restrict__ state,
                                                                  const float* __restrict__ row_mask, unsigned char* __restrict__ gflags,
                                                                  int planes) {
    // planes: out0 / out1 are [2][M][D] - the second bf16 plane m = bf16(o - bf16(o)) follows the first (the two-plane GEMM operands
    // of the "mixed" training precision)
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const bool drop = thresh != 0u;
    if (drop && state != nullptr) { seed = state[0]; off0 += state[1]; off1 += state[1]; }
    const int nv = D >> 2;
    if (row_mask != nullptr) {
        const bool valid = row_mask[row] > 0.f;
        if (gflags != nullptr && (row & 31) == 0) {
            const bool any = __ballot(lane < 32 && row_mask[row + (lane & 31)] > 0.f) != 0ull;
            if (lane == 0) gflags[row >> 5] = any ? 1 : 0;
        }
        if (!valid) {
            if (lane == 0) { stats[row] = 0.f; stats[M + row] = 0.f; }
            for (int c = lane; c < nv; c += 64) {
                reinterpret_cast<uint2*>(out0 + row * D)[c] = uint2{0u, 0u};
                reinterpret_cast<uint2*>(out1 + row * D)[c] = uint2{0u, 0u};
                if (planes) {
                    reinterpret_cast<uint2*>(out0 + (M + row) * D)[c] = uint2{0u, 0u};
                    reinterpret_cast<uint2*>(out1 + (M + row) * D)[c] = uint2{0u, 0u};
                }
            }
            return;
        }
    }
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
    if (lane == 0) { stats[row] = mean; stats[M + row] = rstd; }
#pragma unroll
    for (int br = 0; br < 2; ++br) {
        const f32x4* g4 = reinterpret_cast<const f32x4*>(br ? gamma1 : gamma0);
        const f32x4* b4 = reinterpret_cast<const f32x4*>(br ? beta1 : beta0);
        uint2* orow16 = reinterpret_cast<uint2*>((br ? out1 : out0) + row * D);
        const unsigned long long off = br ? off1 : off0;
#pragma unroll
        for (int i = 0; i < MAXV; ++i) {
            const int c = lane + 64 * i;
            if (c < nv) {
                const f32x4 g = g4[c], b = b4[c];
                f32x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = (v[i][e] - mean) * rstd * g[e] + b[e];
                if (drop) {
                    const unsigned long long ctr = off + (unsigned long long)(row * nv + c);
                    unsigned rnd[4];
                    philox4x32_10((unsigned)ctr, (unsigned)(ctr >> 32), 0u, 0u, (unsigned)seed, (unsigned)(seed >> 32), rnd);
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = rnd[e] >= thresh ? o[e] * dscale : 0.f;
                }
                unsigned short hb[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) hb[e] = f32_to_bf16_bits(o[e]);
                orow16[c] = uint2{(unsigned)hb[0] | ((unsigned)hb[1] << 16), (unsigned)hb[2] | ((unsigned)hb[3] << 16)};
                if (planes) {
                    unsigned short mb[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) mb[e] = f32_to_bf16_bits(o[e] - bf16_bits_to_f32(hb[e]));
                    orow16[(size_t)M * nv + c] = uint2{(unsigned)mb[0] | ((unsigned)mb[1] << 16), (unsigned)mb[2] | ((unsigned)mb[3] << 16)};
                }
            }
        }
    }
}

// ----------------------------------------------------------------------------------------------
// Fused self-attention, one workgroup (4 waves) per (sequence, head); wave w owns queries 32w..32w+31.
// Computed "swapped" (S^T = K Q^T) so that keys sit on MFMA rows = accumulator registers and queries on
// lanes: the softmax over keys is in-register (+ one permlane32 swap) and the probabilities are already
// the B operand of the second product O^T = V^T P^T (cdna_hip_programming.md section 3, "An accumulator
// tile as the next MFMA's operand"; T12).  fp32-input MFMA 32x32x2.
// ----------------------------------------------------------------------------------------------
constexpr int kHeads = 4, kDh = 96, kLmax = 128;
constexpr int kLdQK = kDh + 1;   // [row][d] pitch: 32 consecutive rows at one d hit 32 banks

__device__ __forceinline__ float half_swap_max(float m) {
    const unsigned u = __builtin_bit_cast(unsigned, m);
    auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    return fmaxf(__builtin_bit_cast(float, (unsigned)r[0]), __builtin_bit_cast(float, (unsigned)r[1]));
}
__device__ __forceinline__ float half_swap_sum(float m) {
    const unsigned u = __builtin_bit_cast(unsigned, m);
    auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    return __builtin_bit_cast(float, (unsigned)r[0]) + __builtin_bit_cast(float, (unsigned)r[1]);
}

template <int NKT>   // key tiles of 32 (ceil(L/32))
__device__ __forceinline__ void attention_body(const float* __restrict__ qkv, const float* __restrict__ mask,
                                               float* __restrict__ out, int L, float* smem) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = blockIdx.x / kHeads, head = blockIdx.x % kHeads;
    float* Qs = smem;                              // [NKT*32][97]
    float* Ks = Qs + NKT * 32 * kLdQK;             // [NKT*32][97]
    float* Vs = Ks + NKT * 32 * kLdQK;             // [NKT*32][96]
    float* Ms = Vs + NKT * 32 * kDh;               // [NKT*32] additive key mask
    constexpr int LP = NKT * 32;
    const float* base = qkv + (size_t)n * L * (3 * kHidden) + head * kDh;
    // stage Q, K, V rows (zero beyond L): 24 float4 per row per matrix
    for (int i = tid; i < LP * 24; i += 256) {
        const int row = i / 24, c = i % 24;
        f32x4 q = {0.f, 0.f, 0.f, 0.f}, k = q, v = q;
        if (row < L) {
            const float* r = base + (size_t)row * (3 * kHidden) + c * 4;
            q = *reinterpret_cast<const f32x4*>(r);
            k = *reinterpret_cast<const f32x4*>(r + kHidden);
            v = *reinterpret_cast<const f32x4*>(r + 2 * kHidden);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            Qs[row * kLdQK + c * 4 + e] = q[e];
            Ks[row * kLdQK + c * 4 + e] = k[e];
        }
        *reinterpret_cast<f32x4*>(Vs + row * kDh + c * 4) = v;
    }
    for (int i = tid; i < LP; i += 256) {
        // keys >= L do not exist; masked keys get the reference's additive -10000 (model_components.py:422)
        Ms[i] = i < L ? (mask ? (1.f - mask[(size_t)n * L + i]) * -10000.f : 0.f) : -INFINITY;
    }
    __syncthreads();

    const int q0 = wave * 32;
    if (q0 < L) {
        // S^T[key][query] = sum_d K[key][d] Q[query][d]
        f32x16 s[NKT];
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
            for (int r = 0; r < 16; ++r) s[kt][r] = 0.f;
        const float* qrow = Qs + (q0 + (lane & 31)) * kLdQK + (lane >> 5);
        const float* krow = Ks + (lane & 31) * kLdQK + (lane >> 5);
#pragma unroll 4
        for (int d = 0; d < kDh; d += 2) {
            const float b = qrow[d];
#pragma unroll
            for (int kt = 0; kt < NKT; ++kt)
                s[kt] = __builtin_amdgcn_mfma_f32_32x32x2f32(krow[kt * 32 * kLdQK + d], b, s[kt], 0, 0, 0);
        }
        // softmax over keys: registers, then the other lane half
        const float scale = 0.10206207261596577f;   // 1/sqrt(96), model_components.py:419
        float mx = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                s[kt][r] = s[kt][r] * scale + Ms[key];
                mx = fmaxf(mx, s[kt][r]);
            }
        mx = half_swap_max(mx);
        float sum = 0.f;
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                s[kt][r] = expf(s[kt][r] - mx);
                sum += s[kt][r];
            }
        sum = half_swap_sum(sum);
        const float inv = 1.f / sum;
        // O^T[d][query] = sum_key V[key][d] P[key][query]; register r of key tile kt holds keys
        // (kr, kr+4) on the two lane halves = one k-step of the 32x32x2 MFMA
        f32x16 o[3];
#pragma unroll
        for (int dt = 0; dt < 3; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[dt][r] = 0.f;
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                const float* vrow = Vs + key * kDh + (lane & 31);
#pragma unroll
                for (int dt = 0; dt < 3; ++dt)
                    o[dt] = __builtin_amdgcn_mfma_f32_32x32x2f32(vrow[dt * 32], s[kt][r], o[dt], 0, 0, 0);
            }
        // write the context rows: out[n, q, head*96 + d]; lane = query, registers = d
        const int q = q0 + (lane & 31);
        if (q < L) {
            float* orow = out + ((size_t)n * L + q) * kHidden + head * kDh;
#pragma unroll
            for (int dt = 0; dt < 3; ++dt)
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4) {
                    f32x4 v;
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = o[dt][r4 * 4 + e] * inv;
                    // registers 4*r4..4*r4+3 are 4 consecutive d: d = dt*32 + 8*r4 + 4*(lane>>5) + e
                    *reinterpret_cast<f32x4*>(orow + dt * 32 + 8 * r4 + 4 * (lane >> 5)) = v;
                }
        }
    }
}

__global__ __launch_bounds__(256) void attention_fwd_kernel(const float* __restrict__ qkv, const float* __restrict__ mask,
                                                            float* __restrict__ out, int L) {
    extern __shared__ __attribute__((aligned(16))) float smem_f[];
    const int nkt = (L + 31) >> 5;
    switch (nkt) {
        case 1: attention_body<1>(qkv, mask, out, L, smem_f); break;
        case 2: attention_body<2>(qkv, mask, out, L, smem_f); break;
        case 3: attention_body<3>(qkv, mask, out, L, smem_f); break;
        default: attention_body<4>(qkv, mask, out, L, smem_f); break;
    }
}

// ----------------------------------------------------------------------------------------------
// Modular query pooling: logits_l = h_l . w, masked words -> exactly -1e10 (mask_logits, model.py:444),
// softmax over words, out = sum_l a_l h_l.  One WORKGROUP per query (L <= 64 words; config max_desc_l = 30): wave w takes the
// words l = w (mod 4) and issues the loads of 8 of them before the first reduction; the weighted sum is one column (+ one of the
// last 128) per thread over the words in order.  (Round 4: one wave per query walked the words one after the other - a load
// round trip per word, twice: 30 us for 640 queries, on the query towers' serial chains.  Same per-word dot products, same
// per-column summation order: the results are bit-identical.)
// ----------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void modpool_fwd_kernel(const float* __restrict__ h, const float* __restrict__ mask,
                                                          const float* __restrict__ w, float* __restrict__ out,
                                                          float* __restrict__ attn, int N, int L) {
    __shared__ float lg[64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, tid = threadIdx.x;
    const int n = blockIdx.x;
    const float* hn = h + (size_t)n * L * kHidden;
    float wv[6];
#pragma unroll
    for (int j = 0; j < 6; ++j) wv[j] = w[lane + 64 * j];
    for (int i0 = 0; wave + 4 * i0 < L; i0 += 8) {
        float x[8][6];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int l = wave + 4 * (i0 + i);
#pragma unroll
            for (int j = 0; j < 6; ++j) x[i][j] = l < L ? hn[(size_t)l * kHidden + lane + 64 * j] : 0.f;
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int l = wave + 4 * (i0 + i);
            if (l < L) {                                   // (uniform per wave)
                float d = 0.f;
#pragma unroll
                for (int j = 0; j < 6; ++j) d += x[i][j] * wv[j];
                d = wave_sum(d);
                const float m = mask[(size_t)n * L + l];
                d = d * m + (1.f - m) * -1e10f;
                if (lane == 0) lg[l] = d;
            }
        }
    }
    __syncthreads();
    const float my_logit = lane < L ? lg[lane] : -INFINITY;   // lane l keeps word l's logit (every wave, redundantly)
    const float mx = wave_max(my_logit);
    const float e = lane < L ? expf(my_logit - mx) : 0.f;
    const float a = e / wave_sum(e);
    if (attn && wave == 0 && lane < L) attn[(size_t)n * L + lane] = a;
    const bool two = tid < kHidden - 256;
    float acc0 = 0.f, acc1 = 0.f;
    for (int l0 = 0; l0 < L; l0 += 8) {                // 8 words' loads in flight (a partial unroll of a loop with a shuffle in it is refused)
        float x0[8], x1[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int l = l0 + i;
            x0[i] = l < L ? hn[(size_t)l * kHidden + tid] : 0.f;
            x1[i] = (two && l < L) ? hn[(size_t)l * kHidden + 256 + tid] : 0.f;
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int l = l0 + i;
            if (l < L) {
                const float al = __shfl(a, l);
                acc0 += al * x0[i];
                acc1 += al * x1[i];
            }
        }
    }
    out[(size_t)n * kHidden + tid] = acc0;
    if (two) out[(size_t)n * kHidden + 256 + tid] = acc1;
}

}  // namespace dldkd

using namespace dldkd;

extern "C" {

static int launch_layernorm(const float* x, const float* add, int add_mod, const float* gamma, const float* beta, float* out,
                            long M, int D, float eps, unsigned char* keep, float p_drop, unsigned long long seed,
                            unsigned long long offset, const unsigned long long* state, void* stream,
                            unsigned short* out16 = nullptr, float* stats = nullptr, const float* row_mask = nullptr,
                            unsigned char* gflags = nullptr, const unsigned char* gin = nullptr, unsigned short* planes = nullptr) {
    if (M < 0 || D < 4 || (D & 3) || D > 4096 || add_mod < 0 || !(p_drop >= 0.f && p_drop < 1.f)) {
        set_error("layernorm: bad sizes M=%ld D=%d (D must be a multiple of 4, <= 4096) or p=%f", M, D, (double)p_drop);
        return DLDKD_EINVAL;
    }
    if (M == 0) return DLDKD_OK;
    if (!x || !gamma || !beta || (!out && !out16 && !planes)) { set_error("layernorm: null pointer"); return DLDKD_EINVAL; }
    const dim3 grid((unsigned)((M + 3) / 4)), block(256);
    hipStream_t s = (hipStream_t)stream;
    const int nv = (D / 4 + 63) / 64;
    const double t = (double)p_drop * 4294967296.0;
    const unsigned thresh = t >= 4294967295.0 ? 4294967295u : (unsigned)t;
    const float ds = 1.0f / (1.0f - p_drop);
    if (nv <= 2) DLDKD_LAUNCH(layernorm_kernel<2>, grid, block, 0, s, x, add, add_mod, gamma, beta, out, M, D, eps, keep, thresh, ds, seed, offset, state, out16, stats, row_mask, gflags, gin, planes);
    else if (nv <= 4) DLDKD_LAUNCH(layernorm_kernel<4>, grid, block, 0, s, x, add, add_mod, gamma, beta, out, M, D, eps, keep, thresh, ds, seed, offset, state, out16, stats, row_mask, gflags, gin, planes);
    else if (nv <= 8) DLDKD_LAUNCH(layernorm_kernel<8>, grid, block, 0, s, x, add, add_mod, gamma, beta, out, M, D, eps, keep, thresh, ds, seed, offset, state, out16, stats, row_mask, gflags, gin, planes);
    // (3072-wide rows - the TVR video features - are 12 float4 per lane: the <16> instance holds 64 row registers, 142 VGPRs = 3 waves per
    // SIMD; <12> 110 = 4 waves: 108 -> 94 us at 16,384 rows with dropout, tools/r05_ab_ln12.sh)
    else if (nv <= 12) DLDKD_LAUNCH(layernorm_kernel<12>, grid, block, 0, s, x, add, add_mod, gamma, beta, out, M, D, eps, keep, thresh, ds, seed, offset, state, out16, stats, row_mask, gflags, gin, planes);
    else DLDKD_LAUNCH(layernorm_kernel<16>, grid, block, 0, s, x, add, add_mod, gamma, beta, out, M, D, eps, keep, thresh, ds, seed, offset, state, out16, stats, row_mask, gflags, gin, planes);
    return check_launch("layernorm");
}

int dldkd_layernorm_f32(const float* x, const float* add, int add_mod, const float* gamma, const float* beta, float* out,
                        long M, int D, float eps, void* stream) {
    return launch_layernorm(x, add, add_mod, gamma, beta, out, M, D, eps, nullptr, 0.f, 0, 0, nullptr, stream);
}

int dldkd_layernorm_dropout_f32(const float* x, const float* add, int add_mod, const float* gamma, const float* beta, float* out,
                                unsigned char* keep, long M, int D, float eps, float p_drop, unsigned long long seed,
                                unsigned long long offset, const unsigned long long* state, void* stream) {
    if (!keep || ((uintptr_t)keep & 3)) { set_error("layernorm_dropout: keep mask missing or unaligned"); return DLDKD_EINVAL; }
    return launch_layernorm(x, add, add_mod, gamma, beta, out, M, D, eps, keep, p_drop, seed, offset, state, stream);
}

int dldkd_layernorm_groups_f32(const float* x, const float* add, int add_mod, const float* gamma, const float* beta, float* out,
                               unsigned char* keep, long M, int D, float eps, float p_drop, unsigned long long seed,
                               unsigned long long offset, const unsigned long long* state, const unsigned char* group_flags, void* stream) {
    if (p_drop > 0.f && (!keep || ((uintptr_t)keep & 3))) { set_error("layernorm_groups: keep mask missing or unaligned"); return DLDKD_EINVAL; }
    if (group_flags && (M & 31)) { set_error("layernorm_groups: M %% 32 != 0"); return DLDKD_EINVAL; }
    return launch_layernorm(x, add, add_mod, gamma, beta, out, M, D, eps, p_drop > 0.f ? keep : nullptr, p_drop, seed, offset, state, stream,
                            nullptr, nullptr, nullptr, nullptr, group_flags);
}

int dldkd_layernorm_dropout_rows_f32(const float* x, const float* gamma, const float* beta, float* out, unsigned char* keep, float* stats,
                                     long M, int D, float eps, float p_drop, unsigned long long seed, unsigned long long offset,
                                     const unsigned long long* state, const float* row_mask, unsigned char* group_flags, void* stream) {
    if (p_drop > 0.f && (!keep || ((uintptr_t)keep & 3))) { set_error("layernorm_dropout_rows: keep mask missing or unaligned"); return DLDKD_EINVAL; }
    if (group_flags && (!row_mask || (M & 31))) { set_error("layernorm_dropout_rows: group flags need a row mask and M %% 32 == 0"); return DLDKD_EINVAL; }
    return launch_layernorm(x, nullptr, 0, gamma, beta, out, M, D, eps, p_drop > 0.f ? keep : nullptr, p_drop, seed, offset, state, stream,
                            nullptr, stats, row_mask, group_flags);
}

int dldkd_layernorm_dropout_bf16(const float* x, const float* gamma, const float* beta, void* out_bf16, unsigned char* keep, float* stats,
                                 long M, int D, float eps, float p_drop, unsigned long long seed, unsigned long long offset,
                                 const unsigned long long* state, const float* row_mask, unsigned char* group_flags, void* stream) {
    if (!out_bf16 || ((uintptr_t)out_bf16 & 7)) { set_error("layernorm_dropout_bf16: output missing or unaligned"); return DLDKD_EINVAL; }
    if (keep && ((uintptr_t)keep & 3)) { set_error("layernorm_dropout_bf16: keep mask unaligned"); return DLDKD_EINVAL; }     // (keep == NULL with p > 0: no mask written)
    if (group_flags && (!row_mask || (M & 31))) { set_error("layernorm_dropout_bf16: group flags need a row mask and M %% 32 == 0"); return DLDKD_EINVAL; }
    return launch_layernorm(x, nullptr, 0, gamma, beta, nullptr, M, D, eps, p_drop > 0.f ? keep : nullptr, p_drop, seed, offset, state, stream,
                            (unsigned short*)out_bf16, stats, row_mask, group_flags);
}

int dldkd_layernorm_dropout_bf16_dual(const float* x, const float* gamma0, const float* beta0, const float* gamma1, const float* beta1,
                                      void* out0_bf16, void* out1_bf16, float* stats, long M, int D, float eps, float p_drop,
                                      unsigned long long seed, unsigned long long offset0, unsigned long long offset1,
                                      const unsigned long long* state, const float* row_mask, unsigned char* group_flags, int planes,
                                      void* stream) {
    if (M < 0 || D < 4 || (D & 3) || D > 4096 || !(p_drop >= 0.f && p_drop < 1.f)) {
        set_error("layernorm_dropout_bf16_dual: bad sizes M=%ld D=%d (D must be a multiple of 4, <= 4096) or p=%f", M, D, (double)p_drop);
        return DLDKD_EINVAL;
    }
    if (M == 0) return DLDKD_OK;
    if (!x || !gamma0 || !beta0 || !gamma1 || !beta1 || !out0_bf16 || !out1_bf16 || !stats || (((uintptr_t)out0_bf16 | (uintptr_t)out1_bf16) & 7)) {
        set_error("layernorm_dropout_bf16_dual: null pointer or unaligned output");
        return DLDKD_EINVAL;
    }
    if (group_flags && (!row_mask || (M & 31))) { set_error("layernorm_dropout_bf16_dual: group flags need a row mask and M %% 32 == 0"); return DLDKD_EINVAL; }
    const dim3 grid((unsigned)((M + 3) / 4)), block(256);
    hipStream_t s = (hipStream_t)stream;
    const int nv = (D / 4 + 63) / 64;
    const double t = (double)p_drop * 4294967296.0;
    const unsigned thresh = t >= 4294967295.0 ? 4294967295u : (unsigned)t;
    const float ds = 1.0f / (1.0f - p_drop);
    unsigned short *o0 = (unsigned short*)out0_bf16, *o1 = (unsigned short*)out1_bf16;
    if (nv <= 4) DLDKD_LAUNCH(layernorm_dual_bf16_kernel<4>, grid, block, 0, s, x, gamma0, beta0, gamma1, beta1, o0, o1, stats, M, D, eps, thresh, ds, seed, offset0, offset1, state, row_mask, group_flags, planes ? 1 : 0);
    else if (nv <= 8) DLDKD_LAUNCH(layernorm_dual_bf16_kernel<8>, grid, block, 0, s, x, gamma0, beta0, gamma1, beta1, o0, o1, stats, M, D, eps, thresh, ds, seed, offset0, offset1, state, row_mask, group_flags, planes ? 1 : 0);
    else if (nv <= 12) DLDKD_LAUNCH(layernorm_dual_bf16_kernel<12>, grid, block, 0, s, x, gamma0, beta0, gamma1, beta1, o0, o1, stats, M, D, eps, thresh, ds, seed, offset0, offset1, state, row_mask, group_flags, planes ? 1 : 0);
    else DLDKD_LAUNCH(layernorm_dual_bf16_kernel<16>, grid, block, 0, s, x, gamma0, beta0, gamma1, beta1, o0, o1, stats, M, D, eps, thresh, ds, seed, offset0, offset1, state, row_mask, group_flags, planes ? 1 : 0);
    return check_launch("layernorm_dropout_bf16_dual");
}

int dldkd_layernorm_ex_f32(const float* x, const float* add, int add_mod, const float* gamma, const float* beta, float* out_f32,
                           void* out_bf16, void* out_planes, unsigned char* keep, float* stats, long M, int D, float eps, float p_drop,
                           unsigned long long seed, unsigned long long offset, const unsigned long long* state, const float* row_mask,
                           unsigned char* group_flags_out, const unsigned char* group_flags_in, void* stream) {
    if ((!out_f32 && !out_bf16 && !out_planes) || (((uintptr_t)out_bf16 | (uintptr_t)out_planes) & 7) || ((uintptr_t)keep & 3)) {
        set_error("layernorm_ex: no output, or an unaligned bf16 output / keep mask");
        return DLDKD_EINVAL;
    }
    if ((group_flags_out && (!row_mask || (M & 31))) || (group_flags_in && ((M & 31) || row_mask))) {
        set_error("layernorm_ex: group flags out need a row mask, group flags in exclude one, both need M %% 32 == 0");
        return DLDKD_EINVAL;
    }
    return launch_layernorm(x, add, add_mod, gamma, beta, out_f32, M, D, eps, p_drop > 0.f ? keep : nullptr, p_drop, seed, offset, state, stream,
                            (unsigned short*)out_bf16, stats, row_mask, group_flags_out, group_flags_in, (unsigned short*)out_planes);
}

int dldkd_attention_fwd_f32(const float* qkv, const float* mask, float* out, int N, int L, void* stream) {
    if (N < 0 || L < 1 || L > kLmax) { set_error("attention: bad sizes N=%d L=%d (L <= %d)", N, L, kLmax); return DLDKD_EINVAL; }
    if (N == 0) return DLDKD_OK;
    if (!qkv || !out) { set_error("attention: null pointer"); return DLDKD_EINVAL; }
    const int LP = ((L + 31) / 32) * 32;
    const size_t lds = (size_t)(2 * LP * kLdQK + LP * kDh + LP) * sizeof(float);
    static const bool attr_ok = [] {
        return hipFuncSetAttribute((const void*)attention_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                   (2 * kLmax * kLdQK + kLmax * kDh + kLmax) * (int)sizeof(float)) == hipSuccess;
    }();
    (void)attr_ok;
    DLDKD_LAUNCH(attention_fwd_kernel, dim3(N * kHeads), dim3(256), lds, (hipStream_t)stream, qkv, mask, out, L);
    return check_launch("attention_fwd");
}

int dldkd_modpool_fwd_f32(const float* h, const float* mask, const float* w, float* out, float* attn, int N, int L,
                          void* stream) {
    if (N < 0 || L < 1 || L > 64) { set_error("modpool: bad sizes N=%d L=%d (L <= 64)", N, L); return DLDKD_EINVAL; }
    if (N == 0) return DLDKD_OK;
    if (!h || !mask || !w || !out) { set_error("modpool: null pointer"); return DLDKD_EINVAL; }
    DLDKD_LAUNCH(modpool_fwd_kernel, dim3(N), dim3(256), 0, (hipStream_t)stream, h, mask, w, out, attn, N, L);
    return check_launch("modpool_fwd");
}

}  // extern "C"
