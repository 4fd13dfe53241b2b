// rank_gt: rank of the ground-truth video(s) of every query in a (Nq, Nv) score matrix, on the GPU.
// Replaces the per-query np.argsort + np.where of eval_q2m (reference method/eval.py:69-83) and the
// pure-Python list walk of t2v_map (method/eval.py:97-111): rank = 1 + #(scores strictly greater than the
// ground-truth score) (ties counted optimistically; the reference's unstable argsort breaks them
// arbitrarily).  HBM-bound: one read of the score matrix, one workgroup per query.
// NaN policy: "above" is !(s <= gt), so a NaN score counts as above and a NaN ground-truth score ranks last (nv + 1).  With
// the plain s > gt every comparison against NaN is false and a diverged model (all-NaN scores) would report rank 1 for
// every query, R@K = 100 - and be saved as the best checkpoint.  (The reference's argsort leaves a NaN row in index order.)
#include "common.hpp"

namespace dldkd {

__global__ __launch_bounds__(256) void rank_gt_kernel(const float* __restrict__ scores, int nv,
                                                      const int32_t* __restrict__ gt_ptr,
                                                      const int32_t* __restrict__ gt_idx, int32_t* __restrict__ rank_best,
                                                      int32_t* __restrict__ rank_first) {
    __shared__ int red[2][4];
    const int q = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* row = scores + (size_t)q * nv;
    const int g0 = gt_ptr[q], g1 = gt_ptr[q + 1];
    // thresholds: best (highest-scoring) GT video -> eval_q2m's min rank; first listed GT -> t2v_map
    float best = -INFINITY, first = INFINITY;
    if (g1 > g0) first = row[gt_idx[g0]];
    for (int g = g0; g < g1; ++g) best = fmaxf(best, row[gt_idx[g]]);       // fmaxf drops NaNs: all-NaN ground truth leaves -inf
    int cb = 0, cf = 0;
    // 16-byte loads once the row pointer is aligned
    const int head = (int)(((16 - ((uintptr_t)row & 15)) & 15) / 4);
    const int nhead = head < nv ? head : nv;
    if (tid < nhead) { const float s = row[tid]; cb += !(s <= best); cf += !(s <= first); }
    const int nvec = (nv - nhead) / 4;
    const f32x4* r4 = reinterpret_cast<const f32x4*>(row + nhead);
    for (int i = tid; i < nvec; i += 256) {
        const f32x4 s = r4[i];
#pragma unroll
        for (int e = 0; e < 4; ++e) { cb += !(s[e] <= best); cf += !(s[e] <= first); }
    }
    const int tail0 = nhead + nvec * 4;
    if (tail0 + tid < nv) { const float s = row[tail0 + tid]; cb += !(s <= best); cf += !(s <= first); }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { cb += __shfl_xor(cb, o); cf += __shfl_xor(cf, o); }
    if (lane == 0) { red[0][wave] = cb; red[1][wave] = cf; }
    __syncthreads();
    if (tid == 0) {
        const bool has = g1 > g0;
        // counts can reach nv (NaN ground truth: its own entry is "above" too): clamp to the worst rank nv + 1
        rank_best[q] = has ? min(1 + red[0][0] + red[0][1] + red[0][2] + red[0][3], nv + 1) : nv + 1;   // eval.py:76
        if (rank_first) rank_first[q] = has ? min(1 + red[1][0] + red[1][1] + red[1][2] + red[1][3], nv + 1) : nv + 1;
    }
}


// ---------------------------------------------------------------------------------------------------------------------
// Ranks straight from the scorer's partial planes (part[b][pos(v)][q], simpool_eval.hip), for eval_epoch (eval.py:237-263), which
// needs R@K of the inheritance, exploration and fused scores but never the (Nq, Nv) matrices themselves: no finish pass
// (read 2 planes, write up to 3 matrices) and no three rank passes over them - one read of the two planes in their native,
// query-contiguous layout.  Kernel 1: the ground-truth thresholds of every query (best / first GT video, per score kind);
// kernel 2: counts of !(score <= threshold) per query, video chunks in parallel, integer atomics (exact, order-free).
// Score kinds k: 0 = branch 0, 1 = branch 1, 2 = fused (fuse2, the finish kernel's expression).  thr / cnt: [kind][best|first][nq].
// ---------------------------------------------------------------------------------------------------------------------
struct RankPartArgs {
    const float* part;
    const int32_t* inv;      // [nv] video -> sorted position
    const int32_t* gt_ptr;
    const int32_t* gt_idx;
    const float* q_bad;      // [nq] or null: queries flagged by pack_queries_kernel (NaN / Inf vector) rank last
    float* thr;              // [3][2][nq]
    int32_t* cnt;            // [3][2][nq]
    int nq, nq_pad, nv, nb;
    float w0, w1;
    // gallery sharded by video (dist.sharded_ranks_from_partials): the CSR lists only THIS shard's ground-truth videos (local
    // indices); first_local[q] != 0 iff the query's first listed GT video is one of them.  Thresholds then come out NaN-free
    // (-inf where this shard holds nothing: the all-reduce(MAX) across shards picks the owner's value) and nan_flag[3][2][nq]
    // says where the owner's value was NaN (all-reduced too: that query ranks last, like a NaN threshold does unsharded).
    const int32_t* first_local;
    float* nan_flag;
};

__global__ __launch_bounds__(256) void rank_part_thr_kernel(const RankPartArgs p) {
    const int q = blockIdx.x * 256 + threadIdx.x;
    if (q >= p.nq) return;
    const int g0 = p.gt_ptr[q];
    const int g1 = (p.q_bad != nullptr && p.q_bad[q] != 0.f) ? g0 : p.gt_ptr[q + 1];     // flagged: as if without ground truth
    const float nan = __builtin_nanf("");
    float best[3] = {-INFINITY, -INFINITY, -INFINITY}, first[3] = {nan, nan, nan};   // NaN threshold = "everything is above": rank nv + 1
    for (int g = g0; g < g1; ++g) {
        const size_t row = (size_t)p.inv[p.gt_idx[g]] * p.nq_pad + q;
        float s[3];
        s[0] = p.part[row];
        s[1] = p.nb > 1 ? p.part[(size_t)p.nv * p.nq_pad + row] : s[0];
        s[2] = p.nb > 1 ? fuse2(p.w0, s[0], p.w1, s[1]) : s[0];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            if (g == g0) first[k] = s[k];
            best[k] = fmaxf(best[k], s[k]);           // drops NaNs; an all-NaN ground truth leaves -inf: everything finite is above
        }
    }
    if (p.nan_flag != nullptr) {         // shard mode
        const bool own_first = g1 > g0 && p.first_local[q] != 0;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const bool fn = own_first && first[k] != first[k];
            p.thr[(size_t)(2 * k) * p.nq + q] = best[k];                                  // -inf without a local GT video
            p.thr[(size_t)(2 * k + 1) * p.nq + q] = (own_first && !fn) ? first[k] : -INFINITY;
            p.nan_flag[(size_t)(2 * k) * p.nq + q] = 0.f;                                 // best drops NaNs (fmaxf), as unsharded
            p.nan_flag[(size_t)(2 * k + 1) * p.nq + q] = fn ? 1.f : 0.f;
        }
        return;
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        p.thr[(size_t)(2 * k) * p.nq + q] = g1 > g0 ? best[k] : nan;
        p.thr[(size_t)(2 * k + 1) * p.nq + q] = first[k];
        // the counters the next kernel adds into start at 0 (a kernel store, not a memset node: see optim.hip zero_f32_kernel)
        p.cnt[(size_t)(2 * k) * p.nq + q] = 0;
        p.cnt[(size_t)(2 * k + 1) * p.nq + q] = 0;
    }
}

__global__ __launch_bounds__(256) void zero_i32_kernel(int32_t* __restrict__ x, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) x[i] = 0;
}

constexpr int kRankVChunk = 512;
__global__ __launch_bounds__(256) void rank_part_count_kernel(const RankPartArgs p) {
    __shared__ int red[4][6][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int q = blockIdx.x * 64 + lane;
    const int v0 = blockIdx.y * kRankVChunk, v1 = min(v0 + kRankVChunk, p.nv);
    const bool act = q < p.nq;
    float t[6];
    int c[6] = {0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int k = 0; k < 6; ++k) t[k] = act ? p.thr[(size_t)k * p.nq + q] : 0.f;
    const int qq = act ? q : 0;
    // rows are visited in SORTED position order (any order counts the same): 256 contiguous bytes per wave and row
#pragma unroll 4
    for (int pos = v0 + wave; pos < v1; pos += 4) {
        const size_t row = (size_t)pos * p.nq_pad + qq;
        const float a = p.part[row];
        const float b = p.nb > 1 ? p.part[(size_t)p.nv * p.nq_pad + row] : a;
        const float f = p.nb > 1 ? fuse2(p.w0, a, p.w1, b) : a;
        c[0] += !(a <= t[0]); c[1] += !(a <= t[1]);
        c[2] += !(b <= t[2]); c[3] += !(b <= t[3]);
        c[4] += !(f <= t[4]); c[5] += !(f <= t[5]);
    }
#pragma unroll
    for (int k = 0; k < 6; ++k) red[wave][k][lane] = c[k];
    __syncthreads();
    if (wave == 0 && act) {
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            const int s = red[0][k][lane] + red[1][k][lane] + red[2][k][lane] + red[3][k][lane];
            if (s) atomicAdd(p.cnt + (size_t)k * p.nq + q, s);
        }
    }
}

}  // namespace dldkd

using namespace dldkd;

extern "C" int dldkd_rank_gt(const float* scores, int nq, int nv, const int32_t* gt_ptr, const int32_t* gt_idx,
                             int32_t* rank_best, int32_t* rank_first, void* stream) {
    if (nq < 0 || nv < 1) { set_error("rank_gt: bad sizes nq=%d nv=%d", nq, nv); return DLDKD_EINVAL; }
    if (nq == 0) return DLDKD_OK;
    if (!scores || !gt_ptr || !gt_idx || !rank_best) { set_error("rank_gt: null pointer"); return DLDKD_EINVAL; }
    DLDKD_LAUNCH(rank_gt_kernel, dim3(nq), dim3(256), 0, (hipStream_t)stream, scores, nv, gt_ptr, gt_idx, rank_best,
                       rank_first);
    return check_launch("rank_gt");
}

extern "C" int dldkd_simpool_rank_partials_thr(const void* workspace, const int32_t* inv_order, int nq, int nv, int n_branches, float w0,
                                               float w1, const int32_t* gt_ptr, const int32_t* gt_idx, const int32_t* first_local,
                                               const float* q_bad, float* thr, float* nan_flag, void* stream) {
    if (nq < 0 || nv < 0 || n_branches < 1 || n_branches > 2) { set_error("rank_partials_thr: bad sizes nq=%d nv=%d", nq, nv); return DLDKD_EINVAL; }
    if (nq == 0) return DLDKD_OK;
    if (!gt_ptr || !gt_idx || !first_local || !thr || !nan_flag || (nv > 0 && (!workspace || !inv_order))) { set_error("rank_partials_thr: null pointer"); return DLDKD_EINVAL; }
    RankPartArgs p{(const float*)workspace, inv_order, gt_ptr, gt_idx, q_bad, thr, nullptr, nq, (nq + 31) / 32 * 32, nv, n_branches, w0, w1,
                   first_local, nan_flag};
    DLDKD_LAUNCH(rank_part_thr_kernel, dim3((nq + 255) / 256), dim3(256), 0, (hipStream_t)stream, p);
    return check_launch("rank_partials_thr");
}

extern "C" int dldkd_simpool_rank_partials_count(const void* workspace, int nq, int nv, int n_branches, float w0, float w1, const float* thr,
                                                 int32_t* counts, void* stream) {
    if (nq < 0 || nv < 0 || n_branches < 1 || n_branches > 2) { set_error("rank_partials_count: bad sizes nq=%d nv=%d", nq, nv); return DLDKD_EINVAL; }
    if (nq == 0) return DLDKD_OK;
    if (!thr || !counts || (nv > 0 && !workspace)) { set_error("rank_partials_count: null pointer"); return DLDKD_EINVAL; }
    hipStream_t s = (hipStream_t)stream;
    DLDKD_LAUNCH(zero_i32_kernel, dim3((6 * nq + 255) / 256), dim3(256), 0, s, counts, 6 * nq);
    if (nv == 0) return check_launch("rank_partials_count");
    RankPartArgs p{(const float*)workspace, nullptr, nullptr, nullptr, nullptr, const_cast<float*>(thr), counts, nq, (nq + 31) / 32 * 32, nv,
                   n_branches, w0, w1, nullptr, nullptr};
    DLDKD_LAUNCH(rank_part_count_kernel, dim3((nq + 63) / 64, (nv + kRankVChunk - 1) / kRankVChunk), dim3(256), 0, s, p);
    return check_launch("rank_partials_count");
}

extern "C" int dldkd_simpool_rank_partials(const void* workspace, const int32_t* inv_order, int nq, int nv, int n_branches, float w0,
                                           float w1, const int32_t* gt_ptr, const int32_t* gt_idx, const float* q_bad,
                                           float* thr_scratch, int32_t* counts, void* stream) {
    if (nq < 0 || nv < 1 || n_branches < 1 || n_branches > 2) { set_error("rank_partials: bad sizes nq=%d nv=%d", nq, nv); return DLDKD_EINVAL; }
    if (nq == 0) return DLDKD_OK;
    if (!workspace || !inv_order || !gt_ptr || !gt_idx || !thr_scratch || !counts) { set_error("rank_partials: null pointer"); return DLDKD_EINVAL; }
    RankPartArgs p{(const float*)workspace, inv_order, gt_ptr, gt_idx, q_bad, thr_scratch, counts, nq, (nq + 31) / 32 * 32, nv, n_branches, w0, w1,
                   nullptr, nullptr};
    hipStream_t s = (hipStream_t)stream;
    DLDKD_LAUNCH(rank_part_thr_kernel, dim3((nq + 255) / 256), dim3(256), 0, s, p);
    DLDKD_LAUNCH(rank_part_count_kernel, dim3((nq + 63) / 64, (nv + kRankVChunk - 1) / kRankVChunk), dim3(256), 0, s, p);
    return check_launch("rank_partials");
}
