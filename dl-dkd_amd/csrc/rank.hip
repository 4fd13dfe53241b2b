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
