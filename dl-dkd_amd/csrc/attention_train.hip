// Training form of BertSelfAttention (reference method/model_components.py:398-436): forward with dropout on the attention
// probabilities, and the whole backward, as THREE fused kernels per encoder instead of six strided-batched GEMMs + row softmax
// forward/backward + dropout forward/backward (ten launches whose 128 x 128 GEMM tiles were 94 % padding on the 30-word query
// sequences).  One workgroup per (sequence, head), L <= 128, 4 heads x 96; exact fp32 products on the fp32-input MFMA
// (v_mfma_f32_32x32x2_f32: a k-ordered fmaf chain, cdna_hip_programming.md section 3) - the attention products are ~5 % of the
// step's flops, so the parity (1e-4 losses) and throughput modes share these kernels.
//
//   forward   S^T = K Q^T "swapped" (keys on MFMA rows = accumulator registers, queries on lanes) so the softmax over keys is
//             in-register and the probabilities are already the B operand of O^T = V^T Pd^T (encoder_f32.hip, attention_body).
//             Writes P (before dropout; the backward needs it) and the context layer.
//   bwd_q     query-parallel, same orientation: dPd^T = V dO^T, dP = dPd (.) keep / (1-p), dS = P (.) (dP - rowsum(dP (.) P)) / sqrt(96),
//             dQ^T = K^T dS^T.  Writes dQ, dS, and overwrites P with Pd = P (.) keep / (1-p) (what bwd_kv needs).
//   bwd_kv    key-parallel: dK^T = Q^T dS, dV^T = dO^T Pd with keys on lanes; dS / Pd rows stream from L2 as B operands.
//
// Dropout masks are Philox4x32-10 on the flat index of P (N, 4, L, L) exactly as dldkd_dropout_fwd_f32 draws them for that
// tensor (counter = offset + idx / 4, word idx % 4), so the fused path reproduces the unfused one bit for bit in the masks and
// a hipGraph-captured step can refresh (seed, offset) from device memory (`state`).
#include "common.hpp"

namespace dldkd {

constexpr int kHeadsT = 4, kDhT = 96, kLmaxT = 128;
constexpr int kPitch = kDhT + 1;   // [row][d] images read with 32 consecutive ROWS at one d: 97-float rows hit 32 banks

struct AttnTrainArgs {
    const float* qkv;      // (N, L, 1152)
    const float* mask;     // (N, L) or null
    float* P;              // (N, 4, L, L): probabilities (forward: written; bwd_q: read, overwritten with the dropped ones)
    float* out;            // forward: context (N, L, 384)
    const float* dout;     // backward: gradient of the context (N, L, 384)
    float* dS;             // backward: (N, 4, L, L)
    float* dqkv;           // backward: (N, L, 1152)
    int N, L;
    unsigned thresh;       // keep iff Philox word >= thresh
    float dscale;          // 1 / (1 - p)
    int dropout;           // 0: no dropout
    unsigned long long seed, offset;
    const unsigned long long* state;   // null, or {seed, base offset} in device memory (hipGraph-captured step)
};

__device__ __forceinline__ float hswap_max(float m) {
    const unsigned u = __builtin_bit_cast(unsigned, m);
    auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    return fmaxf(__builtin_bit_cast(float, (unsigned)r[0]), __builtin_bit_cast(float, (unsigned)r[1]));
}
__device__ __forceinline__ float hswap_sum(float m) {
    const unsigned u = __builtin_bit_cast(unsigned, m);
    auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    return __builtin_bit_cast(float, (unsigned)r[0]) + __builtin_bit_cast(float, (unsigned)r[1]);
}

// keep-factors (0 or 1/(1-p)) of the four consecutive keys kb..kb+3 of query row `rowidx` (flat index of P[.., q, kb])
__device__ __forceinline__ void keep4(const AttnTrainArgs& p, unsigned long long seed, unsigned long long off, size_t idx0,
                                      float (&k)[4]) {
    if (!p.dropout) { k[0] = k[1] = k[2] = k[3] = 1.f; return; }
    unsigned rnd[4];
    if ((idx0 & 3) == 0) {
        const unsigned long long c = off + (idx0 >> 2);
        philox4x32_10((unsigned)c, (unsigned)(c >> 32), 0u, 0u, (unsigned)seed, (unsigned)(seed >> 32), rnd);
#pragma unroll
        for (int e = 0; e < 4; ++e) k[e] = rnd[e] >= p.thresh ? p.dscale : 0.f;
    } else {                                   // L not a multiple of 4: the four keys straddle two Philox calls
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const size_t idx = idx0 + e;
            const unsigned long long c = off + (idx >> 2);
            philox4x32_10((unsigned)c, (unsigned)(c >> 32), 0u, 0u, (unsigned)seed, (unsigned)(seed >> 32), rnd);
            k[e] = rnd[idx & 3] >= p.thresh ? p.dscale : 0.f;
        }
    }
}

// registers 4g..4g+3 of an accumulator tile kt hold, on this lane, the keys kt*32 + 8g + 4*(lane>>5) + 0..3 of query lane&31
template <int NKT, typename F>
__device__ __forceinline__ void for_key_groups(int lane, F&& f) {
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
        for (int g = 0; g < 4; ++g) f(kt, g, kt * 32 + 8 * g + 4 * (lane >> 5));
}

// ---------------------------------------------------------------------------------------------------------------- forward
template <int NKT>
__device__ __forceinline__ void attn_train_fwd_body(const AttnTrainArgs& p, float* smem) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nthr = blockDim.x;
    const int n = blockIdx.x / kHeadsT, head = blockIdx.x % kHeadsT, L = p.L;
    constexpr int LP = NKT * 32;
    float* Qs = smem;                     // [LP][97]
    float* Ks = Qs + LP * kPitch;         // [LP][97]
    float* Vs = Ks + LP * kPitch;         // [LP][96]
    float* Ms = Vs + LP * kDhT;           // [LP] additive key mask
    const float* base = p.qkv + (size_t)n * L * (3 * kHidden) + head * kDhT;
    for (int i = tid; i < LP * 24; i += nthr) {
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
            Qs[row * kPitch + c * 4 + e] = q[e];
            Ks[row * kPitch + c * 4 + e] = k[e];
        }
        *reinterpret_cast<f32x4*>(Vs + row * kDhT + c * 4) = v;
    }
    for (int i = tid; i < LP; i += nthr)      // masked keys: the reference's additive -10000 (model_components.py:422)
        Ms[i] = i < L ? (p.mask ? (1.f - p.mask[(size_t)n * L + i]) * -10000.f : 0.f) : -INFINITY;
    __syncthreads();
    const int q0 = wave * 32;
    if (q0 >= L) return;
    unsigned long long seed = p.seed, off = p.offset;
    if (p.state != nullptr) { seed = p.state[0]; off += p.state[1]; }

    f32x16 s[NKT];
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
        for (int r = 0; r < 16; ++r) s[kt][r] = 0.f;
    const float* qrow = Qs + (q0 + (lane & 31)) * kPitch + (lane >> 5);
    const float* krow = Ks + (lane & 31) * kPitch + (lane >> 5);
#pragma unroll 4
    for (int d = 0; d < kDhT; d += 2) {
        const float b = qrow[d];
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt) s[kt] = __builtin_amdgcn_mfma_f32_32x32x2f32(krow[kt * 32 * kPitch + d], b, s[kt], 0, 0, 0);
    }
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
    mx = hswap_max(mx);
    float sum = 0.f;
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
        for (int r = 0; r < 16; ++r) { s[kt][r] = expf(s[kt][r] - mx); sum += s[kt][r]; }
    const float inv = 1.f / hswap_sum(sum);
    // probabilities out (before dropout), dropout in registers
    const int q = q0 + (lane & 31);
    const size_t prow = (((size_t)n * kHeadsT + head) * L + (q < L ? q : 0)) * L;
    const bool vec = !(L & 3);
    for_key_groups<NKT>(lane, [&](int kt, int g, int kb) {
        float k4[4];
        keep4(p, seed, off, prow + kb, k4);
        f32x4 pv;
#pragma unroll
        for (int e = 0; e < 4; ++e) { pv[e] = s[kt][4 * g + e] * inv; s[kt][4 * g + e] = pv[e] * k4[e]; }
        if (q < L && p.P != nullptr) {            // (P == null: a forward whose backward recomputes the probabilities - "mixed" training)
            if (vec) { if (kb < L) *reinterpret_cast<f32x4*>(p.P + prow + kb) = pv; }
            else {
#pragma unroll
                for (int e = 0; e < 4; ++e) if (kb + e < L) p.P[prow + kb + e] = pv[e];
            }
        }
    });
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
            const float* vrow = Vs + key * kDhT + (lane & 31);
#pragma unroll
            for (int dt = 0; dt < 3; ++dt) o[dt] = __builtin_amdgcn_mfma_f32_32x32x2f32(vrow[dt * 32], s[kt][r], o[dt], 0, 0, 0);
        }
    if (q < L) {
        float* orow = p.out + ((size_t)n * L + q) * kHidden + head * kDhT;
#pragma unroll
        for (int dt = 0; dt < 3; ++dt)
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = o[dt][r4 * 4 + e];
                *reinterpret_cast<f32x4*>(orow + dt * 32 + 8 * r4 + 4 * (lane >> 5)) = v;
            }
    }
}

__global__ __launch_bounds__(256) void attn_train_fwd_kernel(const AttnTrainArgs p) {
    extern __shared__ __attribute__((aligned(16))) float smem_f[];
    switch ((p.L + 31) >> 5) {
        case 1: attn_train_fwd_body<1>(p, smem_f); break;
        case 2: attn_train_fwd_body<2>(p, smem_f); break;
        case 3: attn_train_fwd_body<3>(p, smem_f); break;
        default: attn_train_fwd_body<4>(p, smem_f); break;
    }
}

// ------------------------------------------------------------------------------------------------- backward, query-parallel
template <int NKT>
__device__ __forceinline__ void attn_train_bwd_q_body(const AttnTrainArgs& p, float* smem) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nthr = blockDim.x;
    const int n = blockIdx.x / kHeadsT, head = blockIdx.x % kHeadsT, L = p.L;
    constexpr int LP = NKT * 32;
    float* Ks = smem;                     // [LP][96]  read as K^T: 32 consecutive d of one key
    float* Vs = Ks + LP * kDhT;           // [LP][97]  read as rows: 32 consecutive keys at one d
    float* Gs = Vs + LP * kPitch;         // [LP][97]  dO rows (queries)
    const float* base = p.qkv + (size_t)n * L * (3 * kHidden) + head * kDhT;
    const float* gbase = p.dout + (size_t)n * L * kHidden + head * kDhT;
    for (int i = tid; i < LP * 24; i += nthr) {
        const int row = i / 24, c = i % 24;
        f32x4 k = {0.f, 0.f, 0.f, 0.f}, v = k, g = k;
        if (row < L) {
            const float* r = base + (size_t)row * (3 * kHidden) + c * 4;
            k = *reinterpret_cast<const f32x4*>(r + kHidden);
            v = *reinterpret_cast<const f32x4*>(r + 2 * kHidden);
            g = *reinterpret_cast<const f32x4*>(gbase + (size_t)row * kHidden + c * 4);
        }
        *reinterpret_cast<f32x4*>(Ks + row * kDhT + c * 4) = k;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            Vs[row * kPitch + c * 4 + e] = v[e];
            Gs[row * kPitch + c * 4 + e] = g[e];
        }
    }
    __syncthreads();
    const int q0 = wave * 32;
    if (q0 >= L) return;
    unsigned long long seed = p.seed, off = p.offset;
    if (p.state != nullptr) { seed = p.state[0]; off += p.state[1]; }
    // dPd^T[key][q] = sum_d V[key][d] dO[q][d]
    f32x16 dp[NKT];
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
        for (int r = 0; r < 16; ++r) dp[kt][r] = 0.f;
    const float* grow = Gs + (q0 + (lane & 31)) * kPitch + (lane >> 5);
    const float* vrow = Vs + (lane & 31) * kPitch + (lane >> 5);
#pragma unroll 4
    for (int d = 0; d < kDhT; d += 2) {
        const float b = grow[d];
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt) dp[kt] = __builtin_amdgcn_mfma_f32_32x32x2f32(vrow[kt * 32 * kPitch + d], b, dp[kt], 0, 0, 0);
    }
    const int q = q0 + (lane & 31);
    const size_t prow = (((size_t)n * kHeadsT + head) * L + (q < L ? q : 0)) * L;
    const bool vec = !(L & 3);
    // P (saved by the forward pass) in the accumulator layout; dP = dPd * keep; delta = sum_key dP * P
    f32x16 pr[NKT];
    float delta = 0.f;
    for_key_groups<NKT>(lane, [&](int kt, int g, int kb) {
        f32x4 pv = {0.f, 0.f, 0.f, 0.f};
        if (q < L) {
            if (vec) { if (kb < L) pv = *reinterpret_cast<const f32x4*>(p.P + prow + kb); }
            else {
#pragma unroll
                for (int e = 0; e < 4; ++e) if (kb + e < L) pv[e] = p.P[prow + kb + e];
            }
        }
        float k4[4];
        keep4(p, seed, off, prow + kb, k4);
        f32x4 pd;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            pr[kt][4 * g + e] = pv[e];
            dp[kt][4 * g + e] *= k4[e];
            delta += dp[kt][4 * g + e] * pv[e];
            pd[e] = pv[e] * k4[e];
        }
        if (q < L && p.dropout) {               // P -> Pd in place: bwd_kv reads the dropped probabilities
            if (vec) { if (kb < L) *reinterpret_cast<f32x4*>(p.P + prow + kb) = pd; }
            else {
#pragma unroll
                for (int e = 0; e < 4; ++e) if (kb + e < L) p.P[prow + kb + e] = pd[e];
            }
        }
    });
    delta = hswap_sum(delta);
    const float scale = 0.10206207261596577f;
    for_key_groups<NKT>(lane, [&](int kt, int g, int kb) {
        f32x4 ds;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            ds[e] = pr[kt][4 * g + e] * (dp[kt][4 * g + e] - delta) * scale;
            dp[kt][4 * g + e] = ds[e];
        }
        if (q < L) {
            if (vec) { if (kb < L) *reinterpret_cast<f32x4*>(p.dS + prow + kb) = ds; }
            else {
#pragma unroll
                for (int e = 0; e < 4; ++e) if (kb + e < L) p.dS[prow + kb + e] = ds[e];
            }
        }
    });
    // dQ^T[d][q] = sum_key K[key][d] dS^T[key][q]
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
            const float* krow = Ks + key * kDhT + (lane & 31);
#pragma unroll
            for (int dt = 0; dt < 3; ++dt) o[dt] = __builtin_amdgcn_mfma_f32_32x32x2f32(krow[dt * 32], dp[kt][r], o[dt], 0, 0, 0);
        }
    if (q < L) {
        float* orow = p.dqkv + ((size_t)n * L + q) * (3 * kHidden) + head * kDhT;
#pragma unroll
        for (int dt = 0; dt < 3; ++dt)
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = o[dt][r4 * 4 + e];
                *reinterpret_cast<f32x4*>(orow + dt * 32 + 8 * r4 + 4 * (lane >> 5)) = v;
            }
    }
}

__global__ __launch_bounds__(256) void attn_train_bwd_q_kernel(const AttnTrainArgs p) {
    extern __shared__ __attribute__((aligned(16))) float smem_f[];
    switch ((p.L + 31) >> 5) {
        case 1: attn_train_bwd_q_body<1>(p, smem_f); break;
        case 2: attn_train_bwd_q_body<2>(p, smem_f); break;
        case 3: attn_train_bwd_q_body<3>(p, smem_f); break;
        default: attn_train_bwd_q_body<4>(p, smem_f); break;
    }
}

// --------------------------------------------------------------------------------------------------- backward, key-parallel
// wave w owns keys 32w..32w+31 (lanes); dK^T[d][key] = sum_q Q[q][d] dS[q][key], dV^T[d][key] = sum_q dO[q][d] Pd[q][key]:
// A = Q^T / dO^T from LDS (32 consecutive d of one query), B = a row segment of dS / Pd straight from L2.
__global__ __launch_bounds__(256) void attn_train_bwd_kv_kernel(const AttnTrainArgs p) {
    extern __shared__ __attribute__((aligned(16))) float smem_f[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nthr = blockDim.x;
    const int n = blockIdx.x / kHeadsT, head = blockIdx.x % kHeadsT, L = p.L;
    const int LP = ((L + 1) >> 1) << 1;   // queries are consumed in pairs (one 32x32x2 k-step)
    float* Qs = smem_f;                   // [LP][96]
    float* Gs = Qs + (size_t)LP * kDhT;   // [LP][96]
    const float* base = p.qkv + (size_t)n * L * (3 * kHidden) + head * kDhT;
    const float* gbase = p.dout + (size_t)n * L * kHidden + head * kDhT;
    for (int i = tid; i < LP * 24; i += nthr) {
        const int row = i / 24, c = i % 24;
        f32x4 q = {0.f, 0.f, 0.f, 0.f}, g = q;
        if (row < L) {
            q = *reinterpret_cast<const f32x4*>(base + (size_t)row * (3 * kHidden) + c * 4);
            g = *reinterpret_cast<const f32x4*>(gbase + (size_t)row * kHidden + c * 4);
        }
        *reinterpret_cast<f32x4*>(Qs + row * kDhT + c * 4) = q;
        *reinterpret_cast<f32x4*>(Gs + row * kDhT + c * 4) = g;
    }
    __syncthreads();
    const int k0 = wave * 32;
    if (k0 >= L) return;
    const int key = k0 + (lane & 31);
    const bool kok = key < L;
    const size_t pbase = ((size_t)n * kHeadsT + head) * L * L + (kok ? key : 0);
    f32x16 dk[3], dv[3];
#pragma unroll
    for (int dt = 0; dt < 3; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) { dk[dt][r] = 0.f; dv[dt][r] = 0.f; }
    for (int q2 = 0; q2 < LP; q2 += 2) {
        const int q = q2 + (lane >> 5);
        const bool ok = kok && q < L;
        const float bs = ok ? p.dS[pbase + (size_t)q * L] : 0.f;
        const float bp = ok ? p.P[pbase + (size_t)q * L] : 0.f;
        const float* qrow = Qs + q * kDhT + (lane & 31);
        const float* grow = Gs + q * kDhT + (lane & 31);
#pragma unroll
        for (int dt = 0; dt < 3; ++dt) {
            dk[dt] = __builtin_amdgcn_mfma_f32_32x32x2f32(qrow[dt * 32], bs, dk[dt], 0, 0, 0);
            dv[dt] = __builtin_amdgcn_mfma_f32_32x32x2f32(grow[dt * 32], bp, dv[dt], 0, 0, 0);
        }
    }
    if (kok) {
        float* krow = p.dqkv + ((size_t)n * L + key) * (3 * kHidden) + kHidden + head * kDhT;
#pragma unroll
        for (int dt = 0; dt < 3; ++dt)
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                f32x4 a, b;
#pragma unroll
                for (int e = 0; e < 4; ++e) { a[e] = dk[dt][r4 * 4 + e]; b[e] = dv[dt][r4 * 4 + e]; }
                *reinterpret_cast<f32x4*>(krow + dt * 32 + 8 * r4 + 4 * (lane >> 5)) = a;
                *reinterpret_cast<f32x4*>(krow + kHidden + dt * 32 + 8 * r4 + 4 * (lane >> 5)) = b;
            }
    }
}

}  // namespace dldkd

using namespace dldkd;

static int attn_train_args(AttnTrainArgs& a, int N, int L, float p_drop, unsigned long long seed, unsigned long long offset,
                           const unsigned long long* state, const char* what) {
    if (N < 0 || L < 1 || L > kLmaxT || !(p_drop >= 0.f && p_drop < 1.f) || N * kHeadsT > 65535 * 8) {
        set_error("%s: bad sizes N=%d L=%d p=%f (L <= %d)", what, N, L, (double)p_drop, kLmaxT);
        return DLDKD_EINVAL;
    }
    a.N = N;
    a.L = L;
    a.dropout = p_drop > 0.f;
    const double t = (double)p_drop * 4294967296.0;
    a.thresh = t >= 4294967295.0 ? 4294967295u : (unsigned)t;
    a.dscale = 1.0f / (1.0f - p_drop);
    a.seed = seed;
    a.offset = offset;
    a.state = state;
    return DLDKD_OK;
}

extern "C" {

int dldkd_attention_train_fwd_f32(const float* qkv, const float* mask, float* probs, float* out, int N, int L, float p_drop,
                                  unsigned long long seed, unsigned long long offset, const unsigned long long* state,
                                  void* stream) {
    AttnTrainArgs a{};
    const int rc = attn_train_args(a, N, L, p_drop, seed, offset, state, "attention_train_fwd");
    if (rc != DLDKD_OK) return rc;
    if (N == 0) return DLDKD_OK;
    if (!qkv || !out) { set_error("attention_train_fwd: null pointer"); return DLDKD_EINVAL; }
    a.qkv = qkv; a.mask = mask; a.P = probs; a.out = out;
    const int LP = ((L + 31) / 32) * 32;
    const size_t lds = (size_t)(2 * LP * kPitch + LP * kDhT + LP) * sizeof(float);
    static const bool attr_ok = hipFuncSetAttribute((const void*)attn_train_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                                    (2 * kLmaxT * kPitch + kLmaxT * kDhT + kLmaxT) * (int)sizeof(float)) == hipSuccess;
    (void)attr_ok;
    DLDKD_LAUNCH(attn_train_fwd_kernel, dim3(N * kHeadsT), dim3(64 * (LP / 32)), lds, (hipStream_t)stream, a);
    return check_launch("attention_train_fwd");
}

int dldkd_attention_train_bwd_f32(const float* qkv, const float* dout, float* probs, float* dS, float* dqkv, int N, int L,
                                  float p_drop, unsigned long long seed, unsigned long long offset,
                                  const unsigned long long* state, void* stream) {
    AttnTrainArgs a{};
    const int rc = attn_train_args(a, N, L, p_drop, seed, offset, state, "attention_train_bwd");
    if (rc != DLDKD_OK) return rc;
    if (N == 0) return DLDKD_OK;
    if (!qkv || !dout || !probs || !dS || !dqkv) { set_error("attention_train_bwd: null pointer"); return DLDKD_EINVAL; }
    a.qkv = qkv; a.dout = dout; a.P = probs; a.dS = dS; a.dqkv = dqkv;
    const int LP = ((L + 31) / 32) * 32;
    hipStream_t s = (hipStream_t)stream;
    static const bool attr_ok = [] {
        bool ok = hipFuncSetAttribute((const void*)attn_train_bwd_q_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                      (kLmaxT * kDhT + 2 * kLmaxT * kPitch) * (int)sizeof(float)) == hipSuccess;
        ok &= hipFuncSetAttribute((const void*)attn_train_bwd_kv_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                  2 * kLmaxT * kDhT * (int)sizeof(float)) == hipSuccess;
        return ok;
    }();
    (void)attr_ok;
    const dim3 grid(N * kHeadsT), block(64 * (LP / 32));
    DLDKD_LAUNCH(attn_train_bwd_q_kernel, grid, block, (size_t)(LP * kDhT + 2 * LP * kPitch) * sizeof(float), s, a);
    DLDKD_LAUNCH(attn_train_bwd_kv_kernel, grid, block, (size_t)(2 * (((L + 1) >> 1) << 1) * kDhT) * sizeof(float), s, a);
    return check_launch("attention_train_bwd");
}

}  // extern "C"
