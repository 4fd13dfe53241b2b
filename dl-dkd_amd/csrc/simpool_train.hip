// simpool (training): cosine AND raw key-clip max-pooled scores of every (query, video) pair plus the clip-level cosine
// scores of every query's OWN video, forward and backward, without ever forming the (Nq, Nv, L) clip tensor.
// Replaces, inside DLDKD.forward (reference method/model.py:113-129), the pairs
//     get_sim_scores(q, ctx, mask)            model.py:307-329   (F.normalize, einsum, mask_logits, max)
//     get_unnormalized_sim_scores(q, ctx, mask) model.py:331-350
// and the [i, :, label_i] read of compute_kl_loss (model.py:183-197).  Round 1 ran, per (query set, gallery) pair, two
// normalisations, TWO (Nq x Nv L x D) GEMMs that wrote 42 MB clip tensors, two pooling passes, and in the backward pass
// two more GEMMs per operand over a clip-gradient tensor that is zero except at the arg-max clips.
//
//   forward   one batched MFMA GEMM S_v = G_v Q^T per video with the pooling epilogue of common.hpp (gemm_pool_tile); the
//             cosine comes from the same raw product: cos = <q, g> rq rg, rq / rg = 1 / max(|.|, 1e-12) (row_invnorm_kernel).
//   backward  the gradient of a max-pool is a scatter to the arg-max clip:
//       dq[n]   = sum_v [ a_nv g[v, l_raw] + b_nv rq_n (rg g)[v, l_cos] ]  + sum_l e_nl rq_n (rg g)[lab_n, l]  -  rq_n^2 q_n ( sum_v b_nv cos_nv + sum_l e_nl c_nl )
//       dg[v,l] = sum_n [ a_nv 1(l = l_raw) q_n + b_nv 1(l = l_cos) rg_vl rq_n q_n ] + sum_{n: lab_n = v} e_nl rg_vl rq_n q_n  -  rg_vl^2 g_vl ( ... the matching cos-weighted sums )
//     with a = d pooled_raw, b = d pooled_cos, e = d clip_pos: two gather kernels (fp32 VALU, L2-resident operands).
#include "common.hpp"

namespace dldkd {

// inv[r] = 1 / max(|x[r, :]|, 1e-12)   (F.normalize's clamp, model.py:318-319); one wave per row
// y16 != null: the row is also written as bf16 (round to nearest even) - the operand form of the pooled GEMM with bf16 operands
// y16: the row as bf16 beside its norm; mstride > 0 (elements): the second bf16 plane m = bf16(x - bf16(x)) too, at y16 + mstride
// (the two-plane operands of dldkd_simpool_train_fwd_planes)
__device__ __forceinline__ void row_invnorm_row(const float* __restrict__ x, float* __restrict__ inv, long r, int D, int lane,
                                                unsigned short* __restrict__ y16 = nullptr, long mstride = 0) {
    float ss = 0.f;
    const float* row = x + r * D;
    if (!(D & 3) && !((uintptr_t)x & 15)) {
        for (int c = lane * 4; c < D; c += 256) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(row + c);
            ss += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
            if (y16 != nullptr) {
                uint2 pk;
                pk.x = (unsigned)f32_to_bf16_bits(v[0]) | ((unsigned)f32_to_bf16_bits(v[1]) << 16);
                pk.y = (unsigned)f32_to_bf16_bits(v[2]) | ((unsigned)f32_to_bf16_bits(v[3]) << 16);
                *reinterpret_cast<uint2*>(y16 + r * D + c) = pk;
                if (mstride > 0) {
                    unsigned short m[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) m[e] = f32_to_bf16_bits(v[e] - bf16_bits_to_f32(f32_to_bf16_bits(v[e])));
                    *reinterpret_cast<uint2*>(y16 + mstride + r * D + c) = uint2{(unsigned)m[0] | ((unsigned)m[1] << 16), (unsigned)m[2] | ((unsigned)m[3] << 16)};
                }
            }
        }
    } else {
        for (int c = lane; c < D; c += 64) {
            ss += row[c] * row[c];
            if (y16 != nullptr) {
                y16[r * D + c] = f32_to_bf16_bits(row[c]);
                if (mstride > 0) y16[mstride + r * D + c] = f32_to_bf16_bits(row[c] - bf16_bits_to_f32(f32_to_bf16_bits(row[c])));
            }
        }
    }
    ss = wave_sum(ss);
    if (lane == 0) inv[r] = 1.f / fmaxf(sqrtf(ss), 1e-12f);
}

__global__ __launch_bounds__(256) void row_invnorm_kernel(const float* __restrict__ x, float* __restrict__ inv, long M, int D) {
    const long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r < M) row_invnorm_row(x, inv, r, D, threadIdx.x & 63);
}

// the two operands of one scored pair (queries, gallery clips) in one launch: workgroups [0, ceil(M0 / 4)) take x0's rows
__global__ __launch_bounds__(256) void row_invnorm2_kernel(const float* __restrict__ x0, float* __restrict__ inv0, long M0,
                                                           const float* __restrict__ x1, float* __restrict__ inv1, long M1, int D,
                                                           unsigned short* __restrict__ y0, unsigned short* __restrict__ y1, int planes) {
    const long b0 = (M0 + 3) / 4;
    const bool first = (long)blockIdx.x < b0;
    const long r = ((long)blockIdx.x - (first ? 0 : b0)) * 4 + (threadIdx.x >> 6);
    if (r < (first ? M0 : M1))
        row_invnorm_row(first ? x0 : x1, first ? inv0 : inv1, r, D, threadIdx.x & 63, first ? y0 : y1, planes ? (first ? M0 : M1) * D : 0);
}

struct SimpoolBwdArgs {
    const float* q;        // [nq, D]
    const float* g;        // [nv, L, D]
    const float* rq;       // [nq]
    const float* rg;       // [nv * L]
    const int32_t* lens;   // [nv]
    const int32_t* labels; // [nq]
    const int32_t* arg_raw;
    const int32_t* arg_cos;   // [nq, nv]
    const float* pooled_cos;  // [nq, nv]  cos at the arg-max
    const float* clip_pos;    // [nq, L]   cos of the positive column (or null)
    const float* d_raw;       // [nq, nv] or null
    const float* d_cos;       // [nq, nv] or null
    const float* d_clip;      // [nq, L] or null
    float* dq;                // [nq, D]
    float* dg;                // [nv, L, D]
    int nq, nv, L, D;
};

// dq: one workgroup per query: 4 groups of 128 threads (x float4 = up to 512 columns) split the 2 nv (+ len) gathered rows
// between them, then one LDS reduction.  Every gather is an L2 round trip; its address and coefficient come from five index /
// gradient arrays.  Those are staged FIRST - thread v of the workgroup loads video v's entry (coalesced) and leaves (coefficients,
// clip indices) in LDS - so that the gather loop has no load in front of its addresses and runs 8 videos = 16 independent row
// loads deep (round 4: the loop with the index loads inside it, unrolled by 4, took 87 us for 640 queries x 128 videos - 0.25 GB of
// L2-resident rows - twice the time of everything else in the branch's backward pass).
constexpr int kDqGroups = 4;
constexpr int kDqStage = 512;            // videos staged per pass (= the workgroup's threads)
__global__ __launch_bounds__(128 * kDqGroups) void simpool_bwd_dq_kernel(const SimpoolBwdArgs p) {
    __shared__ float red[kDqGroups][516];
    __shared__ float st_a[kDqStage], st_b[kDqStage], st_p[kDqStage];
    __shared__ int st_l[kDqStage];       // l_raw | l_cos << 16
    const int n = blockIdx.x, grp = threadIdx.x >> 7, t = threadIdx.x & 127, c = 4 * t;
    const bool act = c < p.D;
    const float rq = p.rq[n];
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    float proj = 0.f;                                  // sum of (gradient x cosine): the projection onto q
    const size_t rowq = (size_t)n * p.nv;
    auto grow = [&](int v, int l) -> f32x4 {
        return act ? *reinterpret_cast<const f32x4*>(p.g + ((size_t)v * p.L + l) * p.D + c) : f32x4{0.f, 0.f, 0.f, 0.f};
    };
    for (int v0 = 0; v0 < p.nv; v0 += kDqStage) {
        const int nst = min(kDqStage, p.nv - v0);
        if (v0) __syncthreads();                       // the previous pass has been consumed
        if ((int)threadIdx.x < nst) {
            const int v = v0 + threadIdx.x;
            const bool has = p.lens[v] > 0;            // a video without clips pooled to the constant -1e10: no gradient
            const float a = (p.d_raw && has) ? p.d_raw[rowq + v] : 0.f;
            const float b = (p.d_cos && has) ? p.d_cos[rowq + v] : 0.f;
            const int lr = min(max(p.arg_raw[rowq + v], 0), p.L - 1), lc = min(max(p.arg_cos[rowq + v], 0), p.L - 1);   // never outside the video
            st_a[threadIdx.x] = a;
            st_b[threadIdx.x] = b * rq * p.rg[(size_t)v * p.L + lc];
            st_p[threadIdx.x] = b * p.pooled_cos[rowq + v];
            st_l[threadIdx.x] = lr | (lc << 16);
        }
        __syncthreads();
#pragma unroll 8
        for (int i = grp; i < nst; i += kDqGroups) {
            const int lrc = st_l[i];
            const f32x4 xr = grow(v0 + i, lrc & 0xffff), xc = grow(v0 + i, lrc >> 16);
            acc += st_a[i] * xr;
            acc += st_b[i] * xc;
            proj += st_p[i];
        }
    }
    if (p.d_clip) {
        const int v = p.labels[n], len = p.lens[v];
#pragma unroll 4
        for (int l = grp; l < len; l += kDqGroups) {
            const float e = p.d_clip[(size_t)n * p.L + l];
            const f32x4 x = grow(v, l);
            acc += (e * rq * p.rg[(size_t)v * p.L + l]) * x;
            proj += e * p.clip_pos[(size_t)n * p.L + l];
        }
    }
    if (act) *reinterpret_cast<f32x4*>(&red[grp][c]) = acc;
    if (t == 0) red[grp][512] = proj;
    __syncthreads();
    if (grp == 0 && act) {
#pragma unroll
        for (int g = 1; g < kDqGroups; ++g) {
            const f32x4 o = *reinterpret_cast<const f32x4*>(&red[g][c]);
            acc += o;
        }
        proj = red[0][512] + red[1][512] + red[2][512] + red[3][512];
        const f32x4 qv = *reinterpret_cast<const f32x4*>(p.q + (size_t)n * p.D + c);
        *reinterpret_cast<f32x4*>(p.dq + (size_t)n * p.D + c) = acc - (proj * rq * rq) * qv;
    }
}

// dg: one workgroup per (video, 128-column slab).  No accumulator in memory and no atomics: the (query, coefficient)
// contributions of the video are first bucketed BY CLIP in LDS (a counting sort in query order, so the sums have a fixed
// order), then every wave walks the clips it owns (clip % 4 == wave), accumulates a clip's gradient row in registers over
// the clip's bucket - independent, prefetchable query-row loads - and writes it once.
//   contribution of query n to clip l:  a_nv q_n  (l = arg_raw)   +   b_nv rq_n rg_l q_n  (l = arg_cos)
//                                       + e_nl rq_n rg_l q_n  (n a caption of this video, every l < len)
//   and the projection  - rg_l^2 g_l * sum( b_nv cos_nv [l = arg_cos] + e_nl c_nl )
constexpr int kDgCols = 128, kDgThreads = 1024, kDgWaves = kDgThreads / 64;
__global__ __launch_bounds__(kDgThreads) void simpool_bwd_dg_kernel(const SimpoolBwdArgs p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int v = blockIdx.x, c0 = blockIdx.y * kDgCols, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int L = p.L, nq = p.nq, len = p.lens[v];
    float* sc = lds;                                           // [nq][4]: a, b * rq, l_raw, l_cos
    float* bcos = sc + 4 * (size_t)nq;                         // [nq]     b * cos
    int* cnt = reinterpret_cast<int*>(bcos + nq);              // [L + 1]  bucket starts
    float* rowproj = reinterpret_cast<float*>(cnt + L + 1);    // [L]
    int* pos = reinterpret_cast<int*>(rowproj + L);            // [nq + 1] captions of this video, pos[nq] = their number
    int* wcnt = pos + nq + 1;                                  // [kDgWaves] scratch of the ordered compaction
    float2* ent = reinterpret_cast<float2*>(wcnt + kDgWaves);  // [2 nq] (query, coefficient); word offset 6 nq + 2 L + 2 + 16: even
    if (tid == 0) pos[nq] = 0;
    __syncthreads();
    for (int n0 = 0; n0 < nq; n0 += kDgThreads) {
        const int n = n0 + tid;
        bool mine = false;
        if (n < nq) {
            const size_t o = (size_t)n * p.nv + v;
            const float b = (p.d_cos && len > 0) ? p.d_cos[o] : 0.f;
            sc[4 * n + 0] = (p.d_raw && len > 0) ? p.d_raw[o] : 0.f;
            sc[4 * n + 1] = b * p.rq[n];
            sc[4 * n + 2] = __int_as_float(p.arg_raw[o]);
            sc[4 * n + 3] = __int_as_float(p.arg_cos[o]);
            bcos[n] = b * p.pooled_cos[o];
            mine = p.d_clip != nullptr && len > 0 && p.labels[n] == v;
        }
        // captions of this video, in query order: wave ballots + a per-wave prefix (ordered, hence deterministic sums later)
        const unsigned long long m = __ballot(mine);
        if (lane == 0) wcnt[wave] = __popcll(m);
        __syncthreads();
        int base = pos[nq];
        for (int w = 0; w < wave; ++w) base += wcnt[w];
        if (mine) pos[base + __popcll(m & ((1ull << lane) - 1ull))] = n;
        __syncthreads();
        if (tid == 0) { int t = 0; for (int w = 0; w < kDgWaves; ++w) t += wcnt[w]; pos[nq] += t; }
        __syncthreads();
    }
    // Stable counting sort of the 2 nq (query, coefficient) entries by clip, element e = 2 n + {0: raw, 1: cos}, in e order within
    // a clip - the order the first version produced with ONE thread per clip walking all nq queries twice (2 x 640 dependent LDS
    // broadcasts: ~30 us per workgroup, three workgroups per video).  Here: integer LDS atomics count the clips' entries, one
    // thread prefixes the 128 counts, and the placement ranks an element among the equal keys of its wave with ballots (one round
    // per distinct key in the wave) and adds the counts of the waves before it: same positions, a few microseconds.
    float* rgs = reinterpret_cast<float*>(ent + 2 * (size_t)nq);   // [L] rg of this video's clips (0 past its length)
    float* entp = rgs + L;                                        // [2 nq] b cos of the cos-type entries, 0 for raw ones
    int* run = reinterpret_cast<int*>(entp + 2 * (size_t)nq);      // [L] entries of a clip placed by earlier chunks
    int* wk = run + L;                                             // [kDgWaves][L] per-wave counts of the chunk
    for (int i = tid; i <= L; i += kDgThreads) cnt[i] = 0;
    for (int i = tid; i < L; i += kDgThreads) { run[i] = 0; rgs[i] = i < len ? p.rg[(size_t)v * L + i] : 0.f; }
    __syncthreads();
    const int ne = 2 * nq;
    for (int e = tid; e < ne; e += kDgThreads) {
        const int k = __float_as_int(sc[4 * (e >> 1) + 2 + (e & 1)]);
        if (k >= 0 && k < L) atomicAdd(&cnt[k + 1], 1);
    }
    __syncthreads();
    if (tid == 0) for (int l = 0; l < L; ++l) cnt[l + 1] += cnt[l];
    for (int e0 = 0; e0 < ne; e0 += kDgThreads) {
        for (int i = tid; i < kDgWaves * L; i += kDgThreads) wk[i] = 0;
        __syncthreads();
        const int e = e0 + tid;
        int key = -1, rank = 0;
        if (e < ne) {
            const int k = __float_as_int(sc[4 * (e >> 1) + 2 + (e & 1)]);
            if (k >= 0 && k < L) key = k;
        }
        unsigned long long todo = __ballot(key >= 0);
        while (todo) {                                 // one round per distinct clip in the wave
            const int lead = __ffsll((long long)todo) - 1;
            const int K = __shfl(key, lead);
            const unsigned long long m = __ballot(key == K);
            if (key == K) rank = __popcll(m & ((1ull << lane) - 1ull));
            if (lane == lead) wk[wave * L + K] = __popcll(m);
            todo &= ~m;
        }
        __syncthreads();
        if (key >= 0) {
            int at = cnt[key] + run[key] + rank;
            for (int w = 0; w < wave; ++w) at += wk[w * L + key];
            const int n = e >> 1;
            const bool is_cos = e & 1;
            ent[at] = float2{__int_as_float(n), is_cos ? sc[4 * n + 1] * rgs[key] : sc[4 * n]};
            entp[at] = is_cos ? bcos[n] : 0.f;
        }
        __syncthreads();
        for (int l = tid; l < L; l += kDgThreads) {
            int t = 0;
            for (int w = 0; w < kDgWaves; ++w) t += wk[w * L + l];
            run[l] += t;
        }
        __syncthreads();
    }
    if (tid < L) {
        float pr = 0.f;
        for (int k = cnt[tid]; k < cnt[tid + 1]; ++k) pr += entp[k];       // e order = the query order of the first version
        const int np = pos[nq];
        if (tid < len)
            for (int i = 0; i < np; ++i) { const size_t o = (size_t)pos[i] * L + tid; pr += p.d_clip[o] * p.clip_pos[o]; }
        rowproj[tid] = pr;
    }
    __syncthreads();
#if defined(DG_ABLATE) && DG_ABLATE == 1       // (measurement only: the set-up phases without the gather)
    return;
#endif
    const int c = c0 + 2 * lane;
    if (c >= p.D) return;
    const int np = pos[nq];
    // 16 waves, one clip row each at a time: the rows' gather chains (bucket entry -> query row, an L2 round trip) overlap
    for (int l = wave; l < L; l += kDgWaves) {
        float2 acc = {0.f, 0.f};
        const size_t o = ((size_t)v * L + l) * p.D + c;
        if (l < len) {
            const int e0 = cnt[l], e1 = cnt[l + 1];
#pragma unroll 4
            for (int e = e0; e < e1; ++e) {
                const float2 en = ent[e];
                const float2 qv = *reinterpret_cast<const float2*>(p.q + (size_t)__float_as_int(en.x) * p.D + c);
                acc.x += en.y * qv.x;
                acc.y += en.y * qv.y;
            }
            const float rgl = p.rg[(size_t)v * L + l];
            for (int i = 0; i < np; ++i) {
                const int n = pos[i];
                const float w = p.d_clip[(size_t)n * L + l] * p.rq[n] * rgl;
                const float2 qv = *reinterpret_cast<const float2*>(p.q + (size_t)n * p.D + c);
                acc.x += w * qv.x;
                acc.y += w * qv.y;
            }
            const float k = rowproj[l] * rgl * rgl;
            const float2 gv = *reinterpret_cast<const float2*>(p.g + o);
            acc.x -= k * gv.x;
            acc.y -= k * gv.y;
        }
        *reinterpret_cast<float2*>(p.dg + o) = acc;
    }
}

}  // namespace dldkd

using namespace dldkd;

extern "C" {

int dldkd_row_invnorm_f32(const float* x, float* inv, long M, int D, void* stream) {
    if (M < 0 || D < 1) { set_error("row_invnorm: bad sizes"); return DLDKD_EINVAL; }
    if (M == 0) return DLDKD_OK;
    if (!x || !inv) { set_error("row_invnorm: null pointer"); return DLDKD_EINVAL; }
    DLDKD_LAUNCH(row_invnorm_kernel, dim3((unsigned)((M + 3) / 4)), dim3(256), 0, (hipStream_t)stream, x, inv, M, D);
    return check_launch("row_invnorm");
}

int dldkd_row_invnorm2_f32(const float* x0, float* inv0, long M0, const float* x1, float* inv1, long M1, int D, void* stream) {
    if (M0 < 0 || M1 < 0 || D < 1) { set_error("row_invnorm2: bad sizes"); return DLDKD_EINVAL; }
    if (M0 + M1 == 0) return DLDKD_OK;
    if ((M0 && (!x0 || !inv0)) || (M1 && (!x1 || !inv1))) { set_error("row_invnorm2: null pointer"); return DLDKD_EINVAL; }
    DLDKD_LAUNCH(row_invnorm2_kernel, dim3((unsigned)((M0 + 3) / 4 + (M1 + 3) / 4)), dim3(256), 0, (hipStream_t)stream, x0, inv0, M0, x1, inv1,
                 M1, D, (unsigned short*)nullptr, (unsigned short*)nullptr, 0);
    return check_launch("row_invnorm2");
}

int dldkd_row_invnorm2_cast_f32(const float* x0, float* inv0, void* y0_bf16, long M0, const float* x1, float* inv1, void* y1_bf16, long M1,
                                int D, void* stream) {
    if (M0 < 0 || M1 < 0 || D < 1) { set_error("row_invnorm2_cast: bad sizes"); return DLDKD_EINVAL; }
    if (M0 + M1 == 0) return DLDKD_OK;
    if ((M0 && (!x0 || !inv0 || !y0_bf16)) || (M1 && (!x1 || !inv1 || !y1_bf16)) || (((uintptr_t)y0_bf16 | (uintptr_t)y1_bf16) & 7)) {
        set_error("row_invnorm2_cast: null or unaligned pointer");
        return DLDKD_EINVAL;
    }
    DLDKD_LAUNCH(row_invnorm2_kernel, dim3((unsigned)((M0 + 3) / 4 + (M1 + 3) / 4)), dim3(256), 0, (hipStream_t)stream, x0, inv0, M0, x1, inv1,
                 M1, D, (unsigned short*)y0_bf16, (unsigned short*)y1_bf16, 0);
    return check_launch("row_invnorm2_cast");
}

int dldkd_row_invnorm2_planes_f32(const float* x0, float* inv0, void* y0_planes, long M0, const float* x1, float* inv1, void* y1_planes, long M1,
                                  int D, void* stream) {
    if (M0 < 0 || M1 < 0 || D < 1 || (D & 7)) { set_error("row_invnorm2_planes: bad sizes (D must be a multiple of 8)"); return DLDKD_EINVAL; }
    if (M0 + M1 == 0) return DLDKD_OK;
    if ((M0 && (!x0 || !inv0 || !y0_planes)) || (M1 && (!x1 || !inv1 || !y1_planes)) || (((uintptr_t)y0_planes | (uintptr_t)y1_planes) & 15)) {
        set_error("row_invnorm2_planes: null or unaligned pointer");
        return DLDKD_EINVAL;
    }
    DLDKD_LAUNCH(row_invnorm2_kernel, dim3((unsigned)((M0 + 3) / 4 + (M1 + 3) / 4)), dim3(256), 0, (hipStream_t)stream, x0, inv0, M0, x1, inv1,
                 M1, D, (unsigned short*)y0_planes, (unsigned short*)y1_planes, 1);
    return check_launch("row_invnorm2_planes");
}

int dldkd_simpool_train_fwd_bf16in(const void* q_bf16, const void* g_bf16, const float* rq, const float* rg, const int32_t* lens,
                                   const int32_t* labels, int nq, int nv, int L, int D, float* pooled_cos, float* pooled_raw,
                                   int32_t* arg_cos, int32_t* arg_raw, float* clip_pos, void* stream) {
    if (nq < 0 || nv < 0 || L < 1 || L > DLDKD_MAX_CLIPS || D < 1 || nv > 65535) {
        set_error("simpool_train_fwd_bf16in: bad sizes nq=%d nv=%d L=%d D=%d", nq, nv, L, D);
        return DLDKD_EINVAL;
    }
    if (nq == 0 || nv == 0) return DLDKD_OK;
    if (!q_bf16 || !g_bf16 || !rq || !rg || !lens || !labels || !pooled_cos || !pooled_raw || !arg_cos || !arg_raw) {
        set_error("simpool_train_fwd_bf16in: null pointer");
        return DLDKD_EINVAL;
    }
    PoolArgs pa{rg, rq, lens, labels, pooled_raw, pooled_cos, arg_raw, arg_cos, clip_pos, nv, L};
    return launch_simpool_pool_bf16_dma(g_bf16, q_bf16, nv, L, nq, D, pa, stream);
}

/* dldkd_simpool_train_fwd_bf16in with both operands as TWO bf16 planes (dldkd_row_invnorm2_planes_f32: [2][nq][D] / [2][nv L][D]):
 * the two-plane fp32-grade product (three K-long segments of the LDS-DMA kernel) - the pooled scores of the "mixed" training precision. */
int dldkd_simpool_train_fwd_planes(const void* q_planes, const void* g_planes, const float* rq, const float* rg, const int32_t* lens,
                                   const int32_t* labels, int nq, int nv, int L, int D, float* pooled_cos, float* pooled_raw,
                                   int32_t* arg_cos, int32_t* arg_raw, float* clip_pos, void* stream) {
    if (nq < 0 || nv < 0 || L < 1 || L > DLDKD_MAX_CLIPS || D < 1 || nv > 65535) {
        set_error("simpool_train_fwd_planes: bad sizes nq=%d nv=%d L=%d D=%d", nq, nv, L, D);
        return DLDKD_EINVAL;
    }
    if (nq == 0 || nv == 0) return DLDKD_OK;
    if (!q_planes || !g_planes || !rq || !rg || !lens || !labels || !pooled_cos || !pooled_raw || !arg_cos || !arg_raw) {
        set_error("simpool_train_fwd_planes: null pointer");
        return DLDKD_EINVAL;
    }
    PoolArgs pa{rg, rq, lens, labels, pooled_raw, pooled_cos, arg_raw, arg_cos, clip_pos, nv, L};
    return launch_simpool_pool_bf16_dma(g_planes, q_planes, nv, L, nq, D, pa, stream, (long)nv * L * D, (long)nq * D);
}

int dldkd_simpool_train_fwd_f32(int precision, const float* q, const float* g, const float* rq, const float* rg,
                                const int32_t* lens, const int32_t* labels, int nq, int nv, int L, int D, float* pooled_cos,
                                float* pooled_raw, int32_t* arg_cos, int32_t* arg_raw, float* clip_pos, void* stream) {
    if (nq < 0 || nv < 0 || L < 1 || L > DLDKD_MAX_CLIPS || D < 1 || nv > 65535) {
        set_error("simpool_train_fwd: bad sizes nq=%d nv=%d L=%d D=%d", nq, nv, L, D);
        return DLDKD_EINVAL;
    }
    if (nq == 0 || nv == 0) return DLDKD_OK;
    if (!q || !g || !rq || !rg || !lens || !labels || !pooled_cos || !pooled_raw || !arg_cos || !arg_raw) {
        set_error("simpool_train_fwd: null pointer");
        return DLDKD_EINVAL;
    }
    PoolArgs pa{rg, rq, lens, labels, pooled_raw, pooled_cos, arg_raw, arg_cos, clip_pos, nv, L};
    if (precision == DLDKD_GEMM_BF16) return launch_simpool_pool_bf16(g, q, nv, L, nq, D, pa, stream);
    if (precision == DLDKD_GEMM_F32X3) return launch_simpool_pool_x3(g, q, nv, L, nq, D, pa, stream);
    if (precision == DLDKD_GEMM_F32X2) return launch_simpool_pool_x3(g, q, nv, L, nq, D, pa, stream, 2);
    set_error("simpool_train_fwd: precision %d has no pooled kernel (DLDKD_GEMM_F32X3 or DLDKD_GEMM_BF16)", precision);
    return DLDKD_EINVAL;
}

int dldkd_simpool_train_bwd_f32(const float* q, const float* g, const float* rq, const float* rg, const int32_t* lens,
                                const int32_t* labels, const int32_t* arg_cos, const int32_t* arg_raw, const float* pooled_cos,
                                const float* clip_pos, const float* d_cos, const float* d_raw, const float* d_clip, int nq, int nv,
                                int L, int D, float* dq, float* dg, void* stream) {
    if (nq < 0 || nv < 0 || L < 1 || L > DLDKD_MAX_CLIPS || D < 4 || (D & 3) || D > 512) {
        set_error("simpool_train_bwd: bad sizes nq=%d nv=%d L=%d D=%d (D a multiple of 4, <= 512)", nq, nv, L, D);
        return DLDKD_EINVAL;
    }
    if (nq == 0 || nv == 0) return DLDKD_OK;
    if (!q || !g || !rq || !rg || !lens || !labels || !arg_cos || !arg_raw || !pooled_cos || (d_clip && !clip_pos)) {
        set_error("simpool_train_bwd: null pointer");
        return DLDKD_EINVAL;
    }
    if (((uintptr_t)q | (uintptr_t)g | (uintptr_t)dq | (uintptr_t)dg) & 15) { set_error("simpool_train_bwd: unaligned buffer"); return DLDKD_EINVAL; }
    SimpoolBwdArgs p{q, g, rq, rg, lens, labels, arg_raw, arg_cos, pooled_cos, clip_pos, d_raw, d_cos, d_clip, dq, dg, nq, nv, L, D};
    hipStream_t s = (hipStream_t)stream;
    if (dq) DLDKD_LAUNCH(simpool_bwd_dq_kernel, dim3(nq), dim3(128 * kDqGroups), 0, s, p);
    if (dg) {
        // sc 4 nq, bcos nq, cnt L + 1, rowproj L, pos nq + 1, wcnt, ent 4 nq, rgs L, entp 2 nq, run L, per-wave counts 16 L
        const size_t lds = ((size_t)5 * nq + 2 * L + 2 + nq + 4 + kDgWaves + 2 + (size_t)4 * nq + 2 * L + (size_t)2 * nq +
                            (size_t)kDgWaves * L) * sizeof(float);
        if (lds > 160 * 1024) { set_error("simpool_train_bwd: %d queries need %zu bytes of LDS (max ~3000 queries per batch)", nq, lds); return DLDKD_EINVAL; }
        static const bool attr_ok = hipFuncSetAttribute((const void*)simpool_bwd_dg_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess;
        (void)attr_ok;
        DLDKD_LAUNCH(simpool_bwd_dg_kernel, dim3(nv, (D + kDgCols - 1) / kDgCols), dim3(kDgThreads), lds, s, p);
    }
    return check_launch("simpool_train_bwd");
}

}  // extern "C"
