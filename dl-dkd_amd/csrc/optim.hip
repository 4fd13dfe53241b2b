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

// One tower's gradients (wherever autograd left them) -> their ranges of the flat gradient buffer, and the per-tensor sums of squares the
// clip needs, in ONE pass: a gradient is read once between the kernel that produced it and the update (the multi-tensor copy + the
// separate sum-of-squares pass read it twice and wrote it once, as two launches at the serial end of the step).
constexpr int kGatherMax = 32;          // tensors per launch
constexpr int kGatherSpan = 4096;       // elements per workgroup: 4 float4 per thread
struct GatherArgs {
    const float* src[kGatherMax];
    int start[kGatherMax], numel[kGatherMax], tensor[kGatherMax], vec[kGatherMax];
    int blk0[kGatherMax + 1];
    int n;
    float* flat;
    float* norm2;                       // null: copy only
};

__global__ __launch_bounds__(256) void gather_sumsq_kernel(const GatherArgs a) {
    __shared__ float red[4];
    int j = 0;
    while (j + 1 < a.n && (int)blockIdx.x >= a.blk0[j + 1]) ++j;
    const float* __restrict__ src = a.src[j];
    float* __restrict__ dst = a.flat + a.start[j];
    const int n = a.numel[j], base = ((int)blockIdx.x - a.blk0[j]) * kGatherSpan;
    float acc = 0.f;
    if (a.vec[j]) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int i = base + (k * 256 + (int)threadIdx.x) * 4;
            if (i + 3 < n) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(src + i);
                *reinterpret_cast<f32x4*>(dst + i) = v;
                acc += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
            } else {
                for (int e = i; e < n; ++e) { const float v = src[e]; dst[e] = v; acc += v * v; }
            }
        }
    } else {
        for (int i = base + (int)threadIdx.x; i < base + kGatherSpan && i < n; i += 256) { const float v = src[i]; dst[i] = v; acc += v * v; }
    }
    if (a.norm2 == nullptr) return;
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(a.norm2 + a.tensor[j], (red[0] + red[1]) + (red[2] + red[3]));
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

int dldkd_gather_sumsq_f32(const float* const* host_src, const int* host_start, const int* host_numel, const int* host_tensor, int n,
                           float* flat_grad, float* norm2, void* stream) {
    if (n < 0 || n > kGatherMax) { set_error("gather_sumsq: 0..%d tensors per call", kGatherMax); return DLDKD_EINVAL; }
    if (n == 0) return DLDKD_OK;
    if (!host_src || !host_start || !host_numel || !host_tensor || !flat_grad || ((uintptr_t)flat_grad & 15)) {
        set_error("gather_sumsq: null or unaligned pointer");
        return DLDKD_EINVAL;
    }
    GatherArgs a{};
    int blocks = 0, used = 0;
    for (int j = 0; j < n; ++j) {
        if (host_numel[j] < 0 || host_start[j] < 0 || host_tensor[j] < 0 || (host_numel[j] > 0 && !host_src[j])) {
            set_error("gather_sumsq: tensor %d: bad range or null source", j);
            return DLDKD_EINVAL;
        }
        if (host_numel[j] == 0) continue;
        a.src[used] = host_src[j]; a.start[used] = host_start[j]; a.numel[used] = host_numel[j]; a.tensor[used] = host_tensor[j];
        a.vec[used] = !((uintptr_t)host_src[j] & 15) && !(host_start[j] & 3);
        a.blk0[used] = blocks;
        blocks += (host_numel[j] + kGatherSpan - 1) / kGatherSpan;
        ++used;
    }
    if (used == 0) return DLDKD_OK;
    a.blk0[used] = blocks;
    a.n = used; a.flat = flat_grad; a.norm2 = norm2;
    DLDKD_LAUNCH(gather_sumsq_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a);
    return check_launch("gather_sumsq");
}

static int bert_adam_impl(float* p, const float* g, float* m, float* v, const int32_t* chunk_tensor, int n_chunks,
                          const int32_t* t_start, const int32_t* t_numel, int n_tensors, float* norm2_scratch,
                          const float* t_wd, const float* t_lr, const float* t_active, float b1, float b2, float eps,
                          float max_grad_norm, bool norms_ready, void* stream) {
    if (n_chunks < 0 || n_tensors < 0) { set_error("bert_adam: bad sizes"); return DLDKD_EINVAL; }
    if (n_chunks == 0) return DLDKD_OK;
    if (!p || !g || !m || !v || !chunk_tensor || !t_start || !t_numel || !norm2_scratch || !t_wd || !t_lr) {
        set_error("bert_adam: null pointer");
        return DLDKD_EINVAL;
    }
    hipStream_t s = (hipStream_t)stream;
    if (max_grad_norm > 0.f && !norms_ready) {
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

int dldkd_bert_adam_step_f32(float* p, const float* g, float* m, float* v, const int32_t* chunk_tensor, int n_chunks,
                             const int32_t* t_start, const int32_t* t_numel, int n_tensors, float* norm2_scratch,
                             const float* t_wd, const float* t_lr, const float* t_active, float b1, float b2, float eps,
                             float max_grad_norm, void* stream) {
    return bert_adam_impl(p, g, m, v, chunk_tensor, n_chunks, t_start, t_numel, n_tensors, norm2_scratch, t_wd, t_lr, t_active, b1, b2, eps,
                          max_grad_norm, false, stream);
}

int dldkd_bert_adam_update_f32(float* p, const float* g, float* m, float* v, const int32_t* chunk_tensor, int n_chunks,
                               const int32_t* t_start, const int32_t* t_numel, int n_tensors, const float* norm2,
                               const float* t_wd, const float* t_lr, const float* t_active, float b1, float b2, float eps,
                               float max_grad_norm, void* stream) {
    return bert_adam_impl(p, g, m, v, chunk_tensor, n_chunks, t_start, t_numel, n_tensors, const_cast<float*>(norm2), t_wd, t_lr, t_active, b1,
                          b2, eps, max_grad_norm, true, stream);
}

int dldkd_count_above_f32(const float* scores, const float* thr, int nq, int nv, int ld, int32_t* counts, void* stream) {
    if (nq < 0 || nv < 0 || ld < nv) { set_error("count_above: bad sizes"); return DLDKD_EINVAL; }
    if (nq == 0) return DLDKD_OK;
    if (!scores || !thr || !counts) { set_error("count_above: null pointer"); return DLDKD_EINVAL; }
    DLDKD_LAUNCH(count_above_kernel, dim3(nq), dim3(256), 0, (hipStream_t)stream, scores, thr, nv, ld, counts);
    return check_launch("count_above");
}

}  // extern "C"
