// Fused multi-tensor BertAdam step (reference method/optimization.py:278-343) and the threshold count used
// by gather-free sharded ranking.  Both are bandwidth-bound elementwise / reduction kernels.
//
// BertAdam semantics kept exactly: per-TENSOR gradient clip to max_grad_norm (torch clip_grad_norm_:
// coef = min(1, max_norm / (norm + 1e-6))), moments without bias correction, decoupled weight decay
// added to the update, lr already multiplied by the schedule on the host (0 at step 0 for warmup_linear).
// Layout: all parameters of the model live in ONE flat fp32 buffer (and so do grads and both moments);
// tensor t occupies [start[t], start[t] + numel[t]) with start[t] a multiple of 256, so a 256-element chunk
// never straddles two tensors and chunk_tensor[chunk] names its tensor.
#include <atomic>

#include "common.hpp"

namespace dldkd {
static std::atomic<int> g_zero_by_memset{0};

constexpr int kSumsqChunks = 16;

__global__ __launch_bounds__(256) void adam_sumsq_kernel(const float* __restrict__ g, const int32_t* __restrict__ chunk_tensor,
                                                         const int32_t* __restrict__ t_start, const int32_t* __restrict__ t_numel,
                                                         float* __restrict__ norm2, int n_chunks) {
    // kSumsqChunks consecutive 256-element chunks per workgroup; consecutive chunks of one tensor are summed locally and
    // flushed with ONE atomic (per-chunk atomics were 4,608 same-address adds for the 3072 x 384 projection alone)
    __shared__ float red[4];
    const int c0 = blockIdx.x * kSumsqChunks;
    int cur = -1;
    float acc = 0.f;
    for (int c = c0; c < c0 + kSumsqChunks && c < n_chunks; ++c) {
        const int t = chunk_tensor[c];
        const long i = (long)c * 256 + threadIdx.x;
        float v = 0.f;
        if (i < (long)t_start[t] + t_numel[t]) { v = g[i]; v *= v; }
        v = wave_sum(v);
        __syncthreads();                       // red[] of the previous chunk has been consumed
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
        __syncthreads();
        if (threadIdx.x == 0) {
            if (t != cur && cur >= 0) { atomicAdd(norm2 + cur, acc); acc = 0.f; }
            cur = t;
            acc += red[0] + red[1] + red[2] + red[3];
        }
    }
    if (threadIdx.x == 0 && cur >= 0) atomicAdd(norm2 + cur, acc);
}

// norm2 <- 0 as a KERNEL, not hipMemsetAsync: a memset NODE of a replayed hipGraph left every fourth word of this 296-byte
// buffer unzeroed (stale 0x510c7186-like words) whenever the stream was idle at launch - ROCm 7.0.2, seen as per-tensor clip
// coefficients of ~0 in replayed steps only (round 3; tests/test_train_loop_gpu.py variable-length test).
__global__ __launch_bounds__(256) void zero_f32_kernel(float* __restrict__ x, int n) {
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) x[i] = 0.f;
}

__global__ __launch_bounds__(256) void adam_update_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                          float* __restrict__ v, const int32_t* __restrict__ chunk_tensor,
                                                          const int32_t* __restrict__ t_start, const int32_t* __restrict__ t_numel,
                                                          const float* __restrict__ norm2, const float* __restrict__ t_wd,
                                                          const float* __restrict__ t_lr, const float* __restrict__ t_active,
                                                          float b1, float b2, float eps, float max_norm) {
    const int t = chunk_tensor[blockIdx.x];
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long)t_start[t] + t_numel[t]) return;
    // a parameter without a gradient is skipped entirely: no moment decay, no weight decay (`if p.grad is None: continue`,
    // optimization.py:294-295)
    if (t_active != nullptr && t_active[t] == 0.f) return;
    float coef = 1.f;
    if (max_norm > 0.f) coef = fminf(max_norm / (sqrtf(norm2[t]) + 1e-6f), 1.f);
    const float gr = g[i] * coef;
    const float mi = m[i] * b1 + (1.f - b1) * gr;
    const float vi = v[i] * b2 + (1.f - b2) * gr * gr;
    float upd = mi / (sqrtf(vi) + eps);
    const float wd = t_wd[t];
    if (wd > 0.f) upd += wd * p[i];
    m[i] = mi;
    v[i] = vi;
    p[i] -= t_lr[t] * upd;
}

// counts[q] = #{ v < nv : !(scores[q, v] <= thr[q]) }   (one workgroup per query row; NaN scores count as above, rank.hip)
__global__ __launch_bounds__(256) void count_above_kernel(const float* __restrict__ scores, const float* __restrict__ thr,
                                                          int nv, int ld, int32_t* __restrict__ counts) {
    __shared__ int red[4];
    const int q = blockIdx.x;
    const float* row = scores + (size_t)q * ld;
    const float t = thr[q];
    int c = 0;
    for (int i = threadIdx.x; i < nv; i += 256) c += !(row[i] <= t);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) counts[q] = red[0] + red[1] + red[2] + red[3];
}

}  // namespace dldkd

using namespace dldkd;

extern "C" {

/* Zero n floats on `stream` the way the captured step's entry points zero their scratch buffers: by a kernel (default) or, after
 * dldkd_set_zero_by_memset(1), by hipMemsetAsync - a MEMSET node under capture.  Round 3 found such a node defective on ROCm 7.0.2
 * (a replayed memset node left every fourth word of BertAdam's 296-byte norm scratch stale whenever the stream was idle at launch:
 * garbage clip coefficients in replayed steps only); staging.memset_node_defect captures THIS call in a one-node graph and replays
 * it over a poisoned buffer to find out what the runtime at hand does (allocation, synchronisation and the read-back are the
 * caller's: this library only enqueues). */
int dldkd_zero_scratch_f32(float* x, int n, void* stream) {
    if (n < 0 || (n > 0 && !x)) { set_error("zero_scratch: bad arguments"); return DLDKD_EINVAL; }
    if (n == 0) return DLDKD_OK;
    hipStream_t s = (hipStream_t)stream;
    if (g_zero_by_memset.load(std::memory_order_relaxed)) {
        if (hipMemsetAsync(x, 0, (size_t)n * sizeof(float), s) != hipSuccess) return check_launch("zero_scratch (memset)");
        return DLDKD_OK;
    }
    DLDKD_LAUNCH(zero_f32_kernel, dim3((n + 255) / 256), dim3(256), 0, s, x, n);
    return check_launch("zero_scratch");
}

/* How the entry points that zero a small scratch buffer inside a captured step do it (dldkd_bert_adam_step_f32's norm scratch):
 * 0 = a kernel (default, immune to the defect above), 1 = hipMemsetAsync (a memset node under capture).  Returns the previous
 * setting.  train.GraphedTrainStep sets it from the probe's result. */
int dldkd_set_zero_by_memset(int on) { return g_zero_by_memset.exchange(on ? 1 : 0); }

int dldkd_bert_adam_step_f32(float* p, const float* g, float* m, float* v, const int32_t* chunk_tensor, int n_chunks,
                             const int32_t* t_start, const int32_t* t_numel, int n_tensors, float* norm2_scratch,
                             const float* t_wd, const float* t_lr, const float* t_active, float b1, float b2, float eps,
                             float max_grad_norm, void* stream) {
    if (n_chunks < 0 || n_tensors < 0) { set_error("bert_adam: bad sizes"); return DLDKD_EINVAL; }
    if (n_chunks == 0) return DLDKD_OK;
    if (!p || !g || !m || !v || !chunk_tensor || !t_start || !t_numel || !norm2_scratch || !t_wd || !t_lr) {
        set_error("bert_adam: null pointer");
        return DLDKD_EINVAL;
    }
    hipStream_t s = (hipStream_t)stream;
    if (max_grad_norm > 0.f) {
        // zeroed by a kernel unless the start-up probe (staging.memset_node_defect) found this runtime's memset nodes clean and the
        // caller switched them on (dldkd_set_zero_by_memset)
        if (g_zero_by_memset.load(std::memory_order_relaxed)) {
            if (hipMemsetAsync(norm2_scratch, 0, (size_t)n_tensors * sizeof(float), s) != hipSuccess) return check_launch("bert_adam (memset)");
        } else {
            DLDKD_LAUNCH(zero_f32_kernel, dim3((n_tensors + 255) / 256), dim3(256), 0, s, norm2_scratch, n_tensors);
        }
        DLDKD_LAUNCH(adam_sumsq_kernel, dim3((n_chunks + kSumsqChunks - 1) / kSumsqChunks), dim3(256), 0, s, g, chunk_tensor, t_start,
                           t_numel, norm2_scratch, n_chunks);
    }
    DLDKD_LAUNCH(adam_update_kernel, dim3(n_chunks), dim3(256), 0, s, p, g, m, v, chunk_tensor, t_start, t_numel,
                       norm2_scratch, t_wd, t_lr, t_active, b1, b2, eps, max_grad_norm);
    return check_launch("bert_adam");
}

int dldkd_count_above_f32(const float* scores, const float* thr, int nq, int nv, int ld, int32_t* counts, void* stream) {
    if (nq < 0 || nv < 0 || ld < nv) { set_error("count_above: bad sizes"); return DLDKD_EINVAL; }
    if (nq == 0) return DLDKD_OK;
    if (!scores || !thr || !counts) { set_error("count_above: null pointer"); return DLDKD_EINVAL; }
    DLDKD_LAUNCH(count_above_kernel, dim3(nq), dim3(256), 0, (hipStream_t)stream, scores, thr, nv, ld, counts);
    return check_launch("count_above");
}

}  // extern "C"
