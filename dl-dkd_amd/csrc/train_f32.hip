// fp32 training-side kernels of the encoder towers and the clip max-pool (forward pieces that must keep
// intermediates, and every backward piece).  The heavy contractions go through gemm_f32 (plain and
// strided-batched); what is here is bandwidth / latency bound row-wise work, one wave per row.
//   softmax_rows_{fwd,bwd}   attention probabilities, BertSelfAttention.forward model_components.py:417-426
//   layernorm_bwd            nn.LayerNorm backward (input grad + gamma/beta grads)
//   colsum                   bias / position-table gradients (sum over rows)
//   relu_bwd, axpy, mul      LinearLayer ReLU (:310-311), residual adds, dropout masks
//   normalize_rows_{fwd,bwd} F.normalize of get_sim_scores (model.py:318-319)
//   clip_pool_{fwd,bwd}      mask_logits + torch.max over clips (model.py:325-327,347-349)
//   modpool_bwd              get_modularized_queries backward (model.py:245-258)
#include <stdlib.h>

#include "common.hpp"

namespace dldkd {

// ---------------------------------------------------------------- LayerNorm backward
// y = (x+add - mean) * rstd * gamma + beta.  dx (optional) = rstd * (g - mean(g) - xhat * mean(g * xhat)),
// g = dy * gamma; dgamma += dy * xhat, dbeta += dy (atomics, one per column per workgroup).
// WAVES waves per workgroup: narrow rows (D <= 512) are latency-bound (4 dependent wave reductions per row), so they
// run 16 waves x 4 rows per workgroup - the same 64 rows and the same atomic count as 4 waves x 16 rows, with 4x the
// resident waves per SIMD to overlap the reductions.
template <int MAXV, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void layernorm_bwd_kernel(const float* __restrict__ x, const float* __restrict__ add,
                                                            int add_mod, const float* __restrict__ gamma,
                                                            const float* __restrict__ dy, float* __restrict__ dx,
                                                            float* __restrict__ dgamma, float* __restrict__ dbeta, long M,
                                                            int D, float eps, int rows_per_wave,
                                                            const unsigned char* __restrict__ keep, float keep_scale,
                                                            const unsigned char* __restrict__ gin) {
    const int lane = threadIdx.x & 63;
    const int nv = D >> 2;
    const f32x4* g4 = reinterpret_cast<const f32x4*>(gamma);
    f32x4 ag[MAXV], ab[MAXV];
#pragma unroll
    for (int i = 0; i < MAXV; ++i) { ag[i] = f32x4{0.f, 0.f, 0.f, 0.f}; ab[i] = ag[i]; }
    const long row0 = ((long)blockIdx.x * WAVES + (threadIdx.x >> 6)) * rows_per_wave;   // may be >= M: the wave then only joins the reduction
    for (long row = row0; row < row0 + rows_per_wave && row < M; ++row) {
        if (gin != nullptr && gin[row >> 5] == 0) {
            // a row of the padding (32-row group flagged 0 by the tower's first kernel): dy is zero there - no contribution to the
            // parameter gradients, a zero input gradient - and neither x nor dy is read
            if (dx) {
                f32x4* dxr = reinterpret_cast<f32x4*>(dx + row * D);
                for (int c = lane; c < nv; c += 64) dxr[c] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
            continue;
        }
        const f32x4* xr = reinterpret_cast<const f32x4*>(x + row * D);
        const f32x4* ar = add ? reinterpret_cast<const f32x4*>(add + (add_mod > 0 ? row % add_mod : row) * D) : nullptr;
        const f32x4* dyr = reinterpret_cast<const f32x4*>(dy + row * D);
        f32x4 v[MAXV], d[MAXV];
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < MAXV; ++i) {
            const int c = lane + 64 * i;
            v[i] = f32x4{0.f, 0.f, 0.f, 0.f};
            d[i] = v[i];
            if (c < nv) {
                v[i] = xr[c];
                if (ar) { const f32x4 a = ar[c]; v[i] += a; }
                d[i] = dyr[c];
                if (keep != nullptr) {      // the forward pass dropped the normalised row (dldkd_layernorm_dropout_f32): mask dy first
                    const uchar4 k = reinterpret_cast<const uchar4*>(keep + row * D)[c];
                    d[i][0] = k.x ? d[i][0] * keep_scale : 0.f; d[i][1] = k.y ? d[i][1] * keep_scale : 0.f;
                    d[i][2] = k.z ? d[i][2] * keep_scale : 0.f; d[i][3] = k.w ? d[i][3] * keep_scale : 0.f;
                }
                s += v[i][0] + v[i][1] + v[i][2] + v[i][3];
            }
        }
        const float mean = wave_sum(s) / D;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < MAXV; ++i)
            if (lane + 64 * i < nv) {
#pragma unroll
                for (int e = 0; e < 4; ++e) { const float t = v[i][e] - mean; q += t * t; }
            }
        const float rstd = rsqrtf(wave_sum(q) / D + eps);
        float sg = 0.f, sgx = 0.f;
#pragma unroll
        for (int i = 0; i < MAXV; ++i) {
            const int c = lane + 64 * i;
            if (c < nv) {
                const f32x4 g = g4[c];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float xh = (v[i][e] - mean) * rstd;
                    v[i][e] = xh;
                    ag[i][e] += d[i][e] * xh;
                    ab[i][e] += d[i][e];
                    d[i][e] *= g[e];
                    sg += d[i][e];
                    sgx += d[i][e] * xh;
                }
            }
        }
        if (dx) {
            const float mg = wave_sum(sg) / D, mgx = wave_sum(sgx) / D;
            f32x4* dxr = reinterpret_cast<f32x4*>(dx + row * D);
#pragma unroll
            for (int i = 0; i < MAXV; ++i) {
                const int c = lane + 64 * i;
                if (c < nv) {
                    f32x4 o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = rstd * (d[i][e] - mg - v[i][e] * mgx);
                    dxr[c] = o;
                }
            }
        }
    }
    // dgamma / dbeta: combine the block's 4 waves in LDS, then ONE atomic per column per block (per-wave atomics
    // were 1.5M same-address adds per call: the contended case MI355X_MICROARCH.md measures at 14x slower).
    extern __shared__ float ln_red[];            // [WAVES][2][D]
    float* mine = ln_red + (size_t)(threadIdx.x >> 6) * 2 * D;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int c = lane + 64 * i;
        if (c < nv) {
            *reinterpret_cast<f32x4*>(mine + c * 4) = ag[i];
            *reinterpret_cast<f32x4*>(mine + D + c * 4) = ab[i];
        }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < 2 * D; c += 64 * WAVES) {
        float v = 0.f;
#pragma unroll
        for (int w = 0; w < WAVES; ++w) v += ln_red[(size_t)w * 2 * D + c];
        atomicAdd((c < D ? dgamma : dbeta - D) + c, v);
    }
}

// ---------------------------------------------------------------- column sums: out[c] += sum_r x[r, c]
// (measured alternative, dropped: waves streaming whole rows as float4 with an in-block LDS combine - 46 us at
// 16384 x 384 against 22 us for this column-per-thread form: its strided per-lane atomics cost more than it saves)
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ x, float* __restrict__ out, long M, long N,
                                                     int rows_per_block) {
    const long c = (long)blockIdx.x * 256 + threadIdx.x;
    if (c >= N) return;
    const long r0 = (long)blockIdx.y * rows_per_block;
    float s = 0.f;
    for (long r = r0; r < r0 + rows_per_block && r < M; ++r) s += x[r * N + c];
    atomicAdd(out + c, s);
}

// Wave-per-64-columns form: a wave reads 256 contiguous bytes of a row, 4 waves take interleaved rows of a 128-row slab,
// combine in LDS, and issue ONE atomic per column per slab on 64 consecutive addresses.
__global__ __launch_bounds__(256) void colsum64_kernel(const float* __restrict__ x, float* __restrict__ out, long M, long N) {
    __shared__ float red[3][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long c = (long)blockIdx.x * 64 + lane;
    const long r0 = (long)blockIdx.y * 128;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (c < N) {
        const long r1 = r0 + 128 < M ? r0 + 128 : M;
        long r = r0 + wave;
        for (; r + 12 < r1; r += 16) {          // 4 independent loads in flight per lane
            s0 += x[r * N + c];
            s1 += x[(r + 4) * N + c];
            s2 += x[(r + 8) * N + c];
            s3 += x[(r + 12) * N + c];
        }
        for (; r < r1; r += 4) s0 += x[r * N + c];
    }
    const float s = (s0 + s1) + (s2 + s3);
    if (wave > 0) red[wave - 1][lane] = s;
    __syncthreads();
    if (wave == 0 && c < N) atomicAdd(out + c, s + red[0][lane] + red[1][lane] + red[2][lane]);
}

__global__ __launch_bounds__(256) void relu_bwd_kernel(float* __restrict__ dy, const float* __restrict__ y, long n) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < n && !(y[i] > 0.f)) dy[i] = 0.f;
}
__global__ __launch_bounds__(256) void axpy_kernel(float* __restrict__ a, const float* __restrict__ b, float alpha, long n) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) a[i] += alpha * b[i];
}
__global__ __launch_bounds__(256) void mul_kernel(const float* __restrict__ a, const float* __restrict__ m, float scale,
                                                  float* __restrict__ out, long n) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = a[i] * m[i] * scale;
}

// ---------------------------------------------------------------- dropout: counter-based RNG + mask + scale in one pass
// Philox4x32-10 (Salmon et al., SC'11) keyed by `seed`, counter = (offset + i/4): four 32-bit draws per call, one
// per element of a float4.  keep[i] = u32 >= p * 2^32;  out = keep ? x * scale : 0.  The byte mask is what the
// backward pass re-reads (1 B/element instead of a 4 B float mask + a separate multiply kernel).
__global__ __launch_bounds__(256) void dropout_fwd_kernel(const float* __restrict__ x, float* __restrict__ out,
                                                          unsigned char* __restrict__ keep, long n, unsigned thresh, float scale,
                                                          unsigned long long seed, unsigned long long offset,
                                                          const unsigned long long* __restrict__ state) {
    const long q = (long)blockIdx.x * 256 + threadIdx.x;      // group of 4 elements
    const long i = q * 4;
    if (i >= n) return;
    if (state != nullptr) {       // hipGraph-captured step: (seed, base offset) live in device memory, refreshed per replay
        seed = state[0];
        offset += state[1];
    }
    unsigned rnd[4];
    const unsigned long long ctr = offset + (unsigned long long)q;
    philox4x32_10((unsigned)ctr, (unsigned)(ctr >> 32), 0u, 0u, (unsigned)seed, (unsigned)(seed >> 32), rnd);
    if (i + 3 < n) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(x + i);
        f32x4 o;
        uchar4 k;
        k.x = rnd[0] >= thresh; k.y = rnd[1] >= thresh; k.z = rnd[2] >= thresh; k.w = rnd[3] >= thresh;
        o[0] = k.x ? v[0] * scale : 0.f; o[1] = k.y ? v[1] * scale : 0.f;
        o[2] = k.z ? v[2] * scale : 0.f; o[3] = k.w ? v[3] * scale : 0.f;
        *reinterpret_cast<f32x4*>(out + i) = o;
        *reinterpret_cast<uchar4*>(keep + i) = k;
    } else {
        for (int e = 0; i + e < n; ++e) {
            const unsigned char k = rnd[e] >= thresh;
            keep[i + e] = k;
            out[i + e] = k ? x[i + e] * scale : 0.f;
        }
    }
}
__global__ __launch_bounds__(256) void mask_scale_kernel(const float* __restrict__ a, const unsigned char* __restrict__ keep,
                                                         float scale, float* __restrict__ out, long n) {
    const long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i >= n) return;
    if (i + 3 < n) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(a + i);
        const uchar4 k = *reinterpret_cast<const uchar4*>(keep + i);
        f32x4 o;
        o[0] = k.x ? v[0] * scale : 0.f; o[1] = k.y ? v[1] * scale : 0.f;
        o[2] = k.z ? v[2] * scale : 0.f; o[3] = k.w ? v[3] * scale : 0.f;
        *reinterpret_cast<f32x4*>(out + i) = o;
    } else {
        for (int e = 0; i + e < n; ++e) out[i + e] = keep[i + e] ? a[i + e] * scale : 0.f;
    }
}

// ---------------------------------------------------------------- F.normalize rows (eps 1e-12)
__global__ __launch_bounds__(256) void normalize_rows_fwd_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                                 float* __restrict__ inv, long M, int D) {
    const int lane = threadIdx.x & 63;
    const long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= M) return;
    float ss = 0.f;
    for (int c = lane; c < D; c += 64) { const float v = x[r * D + c]; ss += v * v; }
    const float s = 1.f / fmaxf(sqrtf(wave_sum(ss)), 1e-12f);
    for (int c = lane; c < D; c += 64) y[r * D + c] = x[r * D + c] * s;
    if (lane == 0) inv[r] = s;
}
// dx = inv * (dy - y * <y, dy>)   (norm > eps; rows of zeros have y = 0 -> dx = inv * dy like ATen's clamp path)
__global__ __launch_bounds__(256) void normalize_rows_bwd_kernel(const float* __restrict__ y, const float* __restrict__ inv,
                                                                 const float* __restrict__ dy, float* __restrict__ dx,
                                                                 long M, int D) {
    const int lane = threadIdx.x & 63;
    const long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= M) return;
    float dot = 0.f;
    for (int c = lane; c < D; c += 64) dot += y[r * D + c] * dy[r * D + c];
    dot = wave_sum(dot);
    const float s = inv[r];
    for (int c = lane; c < D; c += 64) dx[r * D + c] = s * (dy[r * D + c] - y[r * D + c] * dot);
}

// ---------------------------------------------------------------- clip max-pool over (Nq, Nv, L) scores
// S[q, v, l] for l >= lens[v] becomes exactly -1e10 (mask_logits, model.py:444-445); pooled = max_l, arg = first
// index of the max.  One wave per (q, v).
__global__ __launch_bounds__(256) void clip_pool_fwd_kernel(float* __restrict__ S, const int32_t* __restrict__ lens,
                                                            float* __restrict__ pooled, int32_t* __restrict__ arg, long pairs,
                                                            int nv, int L) {
    const int lane = threadIdx.x & 63;
    const long pr = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (pr >= pairs) return;
    const int len = lens[pr % nv];
    float* row = S + pr * L;
    float best = -INFINITY;
    int bi = 0x7fffffff;
    for (int c = lane; c < L; c += 64) {
        float v = row[c];
        if (c >= len) { v = -1e10f; row[c] = v; }
        if (v > best) { best = v; bi = c; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ob = __shfl_xor(best, o);
        const int oi = __shfl_xor(bi, o);
        if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
    }
    if (lane == 0) { pooled[pr] = best; arg[pr] = bi; }
}
// dS[q, v, arg] += dpooled[q, v]   (valid clips only: the mask multiply zeroes the gradient of padding)
__global__ __launch_bounds__(256) void clip_pool_bwd_kernel(const float* __restrict__ dpooled, const int32_t* __restrict__ arg,
                                                            const int32_t* __restrict__ lens, float* __restrict__ dS,
                                                            long pairs, int nv, int L) {
    const long pr = (long)blockIdx.x * 256 + threadIdx.x;
    if (pr >= pairs) return;
    const int a = arg[pr];
    if (a < lens[pr % nv]) dS[pr * L + a] += dpooled[pr];
}

// ---------------------------------------------------------------- modular pooling backward
// out = sum_l a_l h_l, a = softmax(logit), logit_l = m_l * (h_l . w) + (1 - m_l) * -1e10
// One 256-thread group per query, like the forward kernel - da_l = dout . h_l by the group's wave l (mod 4), 8 words' loads in
// flight; then one column (+ one of the last 128) per thread over the words in order: dh and the column's part of dw - and FOUR
// groups per workgroup, which add their dw parts in LDS: one atomic per column and four queries (640 queries hitting the same 384
// addresses was most of this kernel's time).
constexpr int kModGroups = 4;
__global__ __launch_bounds__(256 * kModGroups) void modpool_bwd_kernel(const float* __restrict__ h, const float* __restrict__ mask,
                                                                       const float* __restrict__ w, const float* __restrict__ attn,
                                                                       const float* __restrict__ dout, float* __restrict__ dh,
                                                                       float* __restrict__ dw, int N, int L) {
    __shared__ float da_s[kModGroups][64];
    __shared__ float wred[kModGroups - 1][kHidden];
    const int grp = threadIdx.x >> 8, tid = threadIdx.x & 255;
    const int lane = tid & 63, wave = tid >> 6;
    const int n = blockIdx.x * kModGroups + grp;
    const bool valid = n < N;
    const int nn = valid ? n : N - 1;                      // (a group past the last query reads the last one and writes nothing)
    float* da = da_s[grp];
    const float* hn = h + (size_t)nn * L * kHidden;
    float* dhn = dh + (size_t)nn * L * kHidden;
    float dv[6];
#pragma unroll
    for (int j = 0; j < 6; ++j) dv[j] = dout[(size_t)nn * kHidden + lane + 64 * j];
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
            if (l < L) {
                float d = 0.f;
#pragma unroll
                for (int j = 0; j < 6; ++j) d += x[i][j] * dv[j];
                d = wave_sum(d);
                if (lane == 0) da[l] = d;
            }
        }
    }
    __syncthreads();
    const float a = lane < L ? attn[(size_t)nn * L + lane] : 0.f;
    const float my_da = lane < L ? da[lane] : 0.f;     // lane l keeps da_l = dout . h_l
    const float dot = wave_sum(a * my_da);
    const float dlogit = a * (my_da - dot) * (lane < L ? mask[(size_t)nn * L + lane] : 0.f);
    const bool two = tid < kHidden - 256;
    const float d0 = dout[(size_t)nn * kHidden + tid], w0 = w[tid];
    const float d1 = two ? dout[(size_t)nn * kHidden + 256 + tid] : 0.f, w1 = two ? w[256 + tid] : 0.f;
    float wacc0 = 0.f, wacc1 = 0.f;
    for (int l0 = 0; l0 < L; l0 += 8) {                // 8 words' loads in flight
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
                const float al = __shfl(a, l), dl = __shfl(dlogit, l);
                if (valid) dhn[(size_t)l * kHidden + tid] = al * d0 + dl * w0;
                wacc0 += dl * x0[i];
                if (two) {
                    if (valid) dhn[(size_t)l * kHidden + 256 + tid] = al * d1 + dl * w1;
                    wacc1 += dl * x1[i];
                }
            }
        }
    }
    if (!valid) { wacc0 = 0.f; wacc1 = 0.f; }
    if (grp > 0) {
        wred[grp - 1][tid] = wacc0;
        if (two) wred[grp - 1][256 + tid] = wacc1;
    }
    __syncthreads();
    if (grp == 0) {
#pragma unroll
        for (int g = 0; g < kModGroups - 1; ++g) {
            wacc0 += wred[g][tid];
            if (two) wacc1 += wred[g][256 + tid];
        }
        atomicAdd(dw + tid, wacc0);
        if (two) atomicAdd(dw + 256 + tid, wacc1);
    }
}


// ----------------------------------------------------------------------------------------------
// "mixed" training mode (round 6): the tower's forward pass runs on the fp32-grade kernels (exact loss values); this kernel then
// writes, from the fp32 intermediates, the bf16 rows the FUSED bf16 backward kernels (tower_train.hip: b3 / attention / b1 / the
// grouped weight-gradient GEMM) read - the set tt::f1 / f3 save in throughput mode:
//   xh1  = ((y0 + pos) - mean1) rstd1,   relu_bits = [y0 > 0] one bit per element   (b1's LayerNorm backward + ReLU mask)
//   h1d  = bf16(h1)       qkv16 = bf16(qkv)       ctx16 = bf16(ctx)             (operands of the weight-gradient blocks / attention)
//   xh2  = ((dd + h1) - mean2) rstd2,   rstd2,   h2_16 = bf16(h2) (video towers)
// One thread per float4 column chunk of a row; rows of 32-row groups flagged 0 (padding) are skipped like the fused kernels skip them.
// ----------------------------------------------------------------------------------------------
// x (fp32) -> two bf16 planes h = bf16(x), m = bf16(x - h) (x = h + m to 16 mantissa bits: the operands of dldkd_gemm_bf16_nt16_planes),
// or a plain fp32 copy (kind 1: gathers scattered parameters - the three attention biases - into one vector) - up to 12 jobs in one launch
struct SplitJobs {
    const float* src[12];
    void* dh[12];
    unsigned short* dm[12];
    long n4[12];               // float4 groups of the job
    long first[13];            // first workgroup of job j (prefix sums, 256 groups per workgroup)
    int kind[12];
    int njobs;
};

__global__ __launch_bounds__(256) void split2_jobs_kernel(const SplitJobs p) {
    int j = 0;
#pragma unroll 1
    while (j + 1 < p.njobs && (long)blockIdx.x >= p.first[j + 1]) ++j;
    const long i = ((long)blockIdx.x - p.first[j]) * 256 + threadIdx.x;
    if (i >= p.n4[j]) return;
    const f32x4 v = reinterpret_cast<const f32x4*>(p.src[j])[i];
    if (p.kind[j] == 1) { reinterpret_cast<f32x4*>(p.dh[j])[i] = v; return; }
    unsigned short h[4], m[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        h[e] = f32_to_bf16_bits(v[e]);
        m[e] = f32_to_bf16_bits(v[e] - bf16_bits_to_f32(h[e]));
    }
    reinterpret_cast<uint2*>(p.dh[j])[i] = uint2{(unsigned)h[0] | ((unsigned)h[1] << 16), (unsigned)h[2] | ((unsigned)h[3] << 16)};
    reinterpret_cast<uint2*>(p.dm[j])[i] = uint2{(unsigned)m[0] | ((unsigned)m[1] << 16), (unsigned)m[2] | ((unsigned)m[3] << 16)};
}

struct EmitArgs {
    const float *y0, *pos, *stats1, *h1, *qkv, *ctx, *dd, *stats2, *h2;
    const unsigned char* flags;
    unsigned short *xh1, *h1d, *qkv16, *ctx16, *xh2, *h2_16;
    unsigned char* relu_bits;
    float* rstd2;
    long M;
    int L;
};

__device__ __forceinline__ uint2 pack_bf16x4(const f32x4 v) {
    return uint2{(unsigned)f32_to_bf16_bits(v[0]) | ((unsigned)f32_to_bf16_bits(v[1]) << 16),
                 (unsigned)f32_to_bf16_bits(v[2]) | ((unsigned)f32_to_bf16_bits(v[3]) << 16)};
}

__global__ __launch_bounds__(256) void tower_train_emit_kernel(const EmitArgs p) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    const long row = idx / 96;
    const int c4 = (int)(idx % 96);
    if (row >= p.M) return;
    if (p.flags != nullptr && p.flags[row >> 5] == 0) return;
    const size_t o = (size_t)row * 96 + c4;                          // float4 / bf16x4 index in a (M, 384) tensor
    const f32x4 y = reinterpret_cast<const f32x4*>(p.y0)[o];
    const f32x4 ps = reinterpret_cast<const f32x4*>(p.pos)[(size_t)(row % p.L) * 96 + c4];
    const float m1 = p.stats1[row], r1 = p.stats1[p.M + row];
    f32x4 x1;
    unsigned nib = 0;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        x1[e] = (y[e] + ps[e] - m1) * r1;
        nib |= (y[e] > 0.f ? 1u : 0u) << e;
    }
    reinterpret_cast<uint2*>(p.xh1)[o] = pack_bf16x4(x1);
    // the ReLU mask [y0 > 0] as tt::f1_kernel leaves it: bit c % 8 of byte c / 8 of the row; threads of even c4 write the byte (their
    // neighbour lane holds the other nibble: idx = 96 row + c4 has c4's parity, and a wave starts at an even idx)
    const unsigned other = __shfl_down(nib, 1);
    if (!(c4 & 1)) p.relu_bits[(size_t)row * 48 + (c4 >> 1)] = (unsigned char)(nib | (other << 4));
    const f32x4 h1 = reinterpret_cast<const f32x4*>(p.h1)[o];
    if (p.h1d != nullptr) reinterpret_cast<uint2*>(p.h1d)[o] = pack_bf16x4(h1);            // (null: the caller holds them already - plane 0 of
    if (p.ctx16 != nullptr) reinterpret_cast<uint2*>(p.ctx16)[o] = pack_bf16x4(reinterpret_cast<const f32x4*>(p.ctx)[o]);   //  the two-plane GEMM operands)
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const size_t q = (size_t)row * 288 + 96 * j + c4;
        reinterpret_cast<uint2*>(p.qkv16)[q] = pack_bf16x4(reinterpret_cast<const f32x4*>(p.qkv)[q]);
    }
    const f32x4 dd = reinterpret_cast<const f32x4*>(p.dd)[o];
    const float m2 = p.stats2[row], r2 = p.stats2[p.M + row];
    f32x4 x2;
#pragma unroll
    for (int e = 0; e < 4; ++e) x2[e] = (dd[e] + h1[e] - m2) * r2;
    reinterpret_cast<uint2*>(p.xh2)[o] = pack_bf16x4(x2);
    if (c4 == 0) p.rstd2[row] = r2;
    if (p.h2_16 != nullptr) reinterpret_cast<uint2*>(p.h2_16)[o] = pack_bf16x4(reinterpret_cast<const f32x4*>(p.h2)[o]);
}

}  // namespace dldkd

using namespace dldkd;

#define LAUNCH1D(kernel, n, per_block, ...) \
    DLDKD_LAUNCH(kernel, dim3((unsigned)(((n) + (per_block) - 1) / (per_block))), dim3(256), 0, (hipStream_t)stream, __VA_ARGS__)

extern "C" {

static int launch_layernorm_bwd(const float* x, const float* add, int add_mod, const float* gamma, const float* dy, float* dx,
                                float* dgamma, float* dbeta, long M, int D, float eps, const unsigned char* keep, float keep_scale,
                                const unsigned char* gin, void* stream);

int dldkd_layernorm_bwd_f32(const float* x, const float* add, int add_mod, const float* gamma, const float* dy, float* dx,
                            float* dgamma, float* dbeta, long M, int D, float eps, const unsigned char* keep, float keep_scale,
                            void* stream) {
    return launch_layernorm_bwd(x, add, add_mod, gamma, dy, dx, dgamma, dbeta, M, D, eps, keep, keep_scale, nullptr, stream);
}

int dldkd_layernorm_bwd_groups_f32(const float* x, const float* add, int add_mod, const float* gamma, const float* dy, float* dx,
                                   float* dgamma, float* dbeta, long M, int D, float eps, const unsigned char* keep, float keep_scale,
                                   const unsigned char* group_flags, void* stream) {
    if (group_flags && (M & 31)) { set_error("layernorm_bwd_groups: M %% 32 != 0"); return DLDKD_EINVAL; }
    return launch_layernorm_bwd(x, add, add_mod, gamma, dy, dx, dgamma, dbeta, M, D, eps, keep, keep_scale, group_flags, stream);
}

static int launch_layernorm_bwd(const float* x, const float* add, int add_mod, const float* gamma, const float* dy, float* dx,
                                float* dgamma, float* dbeta, long M, int D, float eps, const unsigned char* keep, float keep_scale,
                                const unsigned char* gin, void* stream) {
    if (M < 0 || D < 4 || (D & 3) || D > 4096) { set_error("layernorm_bwd: bad sizes"); return DLDKD_EINVAL; }
    if (M == 0) return DLDKD_OK;
    if (!x || !gamma || !dy || !dgamma || !dbeta) { set_error("layernorm_bwd: null pointer"); return DLDKD_EINVAL; }
    const int nv = (D / 4 + 63) / 64;
    hipStream_t s = (hipStream_t)stream;
    if (nv <= 2) {           // 16 waves x 4 rows per workgroup
        const long waves = (M + 3) / 4;
        DLDKD_LAUNCH((layernorm_bwd_kernel<2, 16>), dim3((unsigned)((waves + 15) / 16)), dim3(1024), (size_t)32 * D * sizeof(float), s,
                           x, add, add_mod, gamma, dy, dx, dgamma, dbeta, M, D, eps, 4, keep, keep_scale, gin);
        return check_launch("layernorm_bwd");
    }
    const int rpw = 8;       // 4 waves x 8 rows
    const long waves = (M + rpw - 1) / rpw;
    const dim3 grid((unsigned)((waves + 3) / 4)), block(256);
    const size_t lds = (size_t)8 * D * sizeof(float);
    static const bool attr_ok = [] {
        return hipFuncSetAttribute((const void*)layernorm_bwd_kernel<16, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 8 * 4096 * 4) == hipSuccess;
    }();
    (void)attr_ok;
    if (nv <= 4) DLDKD_LAUNCH((layernorm_bwd_kernel<4, 4>), grid, block, lds, s, x, add, add_mod, gamma, dy, dx, dgamma, dbeta, M, D, eps, rpw, keep, keep_scale, gin);
    // (no MAXV = 8 instantiation: hipcc spilled 248 registers in it; rows of 1028..2048 floats use the 16-wide form)
    else DLDKD_LAUNCH((layernorm_bwd_kernel<16, 4>), grid, block, lds, s, x, add, add_mod, gamma, dy, dx, dgamma, dbeta, M, D, eps, rpw, keep, keep_scale, gin);
    return check_launch("layernorm_bwd");
}
int dldkd_colsum_f32(const float* x, float* out, long M, long N, void* stream) {
    if (M < 0 || N < 0) { set_error("colsum: bad sizes"); return DLDKD_EINVAL; }
    if (M == 0 || N == 0) return DLDKD_OK;
    if (M >= 512 && (M + 127) / 128 <= 65535) {
        DLDKD_LAUNCH(colsum64_kernel, dim3((unsigned)((N + 63) / 64), (unsigned)((M + 127) / 128)), dim3(256), 0,
                           (hipStream_t)stream, x, out, M, N);
        return check_launch("colsum64");
    }
    // few rows (the per-row-tile partial sums of dldkd_linear_lngrad: 128 x 3072): 64 rows per block left 24 workgroups
    // adding 64 dependent loads each (18 us); 8 rows per block is 192 workgroups
    const int rpb = M <= 256 ? 8 : 64;
    DLDKD_LAUNCH(colsum_kernel, dim3((unsigned)((N + 255) / 256), (unsigned)((M + rpb - 1) / rpb)), dim3(256), 0,
                       (hipStream_t)stream, x, out, M, N, rpb);
    return check_launch("colsum");
}
int dldkd_relu_bwd_f32(float* dy, const float* y, long n, void* stream) {
    if (n <= 0) return n < 0 ? DLDKD_EINVAL : DLDKD_OK;
    LAUNCH1D(relu_bwd_kernel, n, 256, dy, y, n);
    return check_launch("relu_bwd");
}
int dldkd_axpy_f32(float* a, const float* b, float alpha, long n, void* stream) {
    if (n <= 0) return n < 0 ? DLDKD_EINVAL : DLDKD_OK;
    LAUNCH1D(axpy_kernel, n, 256, a, b, alpha, n);
    return check_launch("axpy");
}
int dldkd_mul_f32(const float* a, const float* m, float scale, float* out, long n, void* stream) {
    if (n <= 0) return n < 0 ? DLDKD_EINVAL : DLDKD_OK;
    LAUNCH1D(mul_kernel, n, 256, a, m, scale, out, n);
    return check_launch("mul");
}
int dldkd_dropout_fwd_f32(const float* x, float* out, unsigned char* keep, long n, float p, unsigned long long seed,
                          unsigned long long offset, const unsigned long long* state, void* stream) {
    if (n < 0 || !(p >= 0.f && p < 1.f)) { set_error("dropout_fwd: bad n=%ld or p=%f", n, (double)p); return DLDKD_EINVAL; }
    if (n == 0) return DLDKD_OK;
    if (!x || !out || !keep) { set_error("dropout_fwd: null pointer"); return DLDKD_EINVAL; }
    if (((uintptr_t)x | (uintptr_t)out) & 15 || ((uintptr_t)keep & 3)) { set_error("dropout_fwd: unaligned buffer"); return DLDKD_EINVAL; }
    const double t = (double)p * 4294967296.0;
    const unsigned thresh = t >= 4294967295.0 ? 4294967295u : (unsigned)t;
    LAUNCH1D(dropout_fwd_kernel, (n + 3) / 4, 256, x, out, keep, n, thresh, 1.0f / (1.0f - p), seed, offset, state);
    return check_launch("dropout_fwd");
}
int dldkd_mask_scale_f32(const float* a, const unsigned char* keep, float scale, float* out, long n, void* stream) {
    if (n <= 0) return n < 0 ? DLDKD_EINVAL : DLDKD_OK;
    if (!a || !out || !keep) { set_error("mask_scale: null pointer"); return DLDKD_EINVAL; }
    if (((uintptr_t)a | (uintptr_t)out) & 15 || ((uintptr_t)keep & 3)) { set_error("mask_scale: unaligned buffer"); return DLDKD_EINVAL; }
    LAUNCH1D(mask_scale_kernel, (n + 3) / 4, 256, a, keep, scale, out, n);
    return check_launch("mask_scale");
}
int dldkd_normalize_rows_fwd_f32(const float* x, float* y, float* inv, long M, int D, void* stream) {
    if (M < 0 || D < 1) { set_error("normalize_rows_fwd: bad sizes"); return DLDKD_EINVAL; }
    if (M == 0) return DLDKD_OK;
    LAUNCH1D(normalize_rows_fwd_kernel, M, 4, x, y, inv, M, D);
    return check_launch("normalize_rows_fwd");
}
int dldkd_normalize_rows_bwd_f32(const float* y, const float* inv, const float* dy, float* dx, long M, int D, void* stream) {
    if (M < 0 || D < 1) { set_error("normalize_rows_bwd: bad sizes"); return DLDKD_EINVAL; }
    if (M == 0) return DLDKD_OK;
    LAUNCH1D(normalize_rows_bwd_kernel, M, 4, y, inv, dy, dx, M, D);
    return check_launch("normalize_rows_bwd");
}
int dldkd_clip_pool_fwd_f32(float* S, const int32_t* lens, float* pooled, int32_t* arg, int nq, int nv, int L, void* stream) {
    if (nq < 0 || nv < 0 || L < 1) { set_error("clip_pool_fwd: bad sizes"); return DLDKD_EINVAL; }
    const long pairs = (long)nq * nv;
    if (pairs == 0) return DLDKD_OK;
    LAUNCH1D(clip_pool_fwd_kernel, pairs, 4, S, lens, pooled, arg, pairs, nv, L);
    return check_launch("clip_pool_fwd");
}
int dldkd_clip_pool_bwd_f32(const float* dpooled, const int32_t* arg, const int32_t* lens, float* dS, int nq, int nv, int L,
                            void* stream) {
    if (nq < 0 || nv < 0 || L < 1) { set_error("clip_pool_bwd: bad sizes"); return DLDKD_EINVAL; }
    const long pairs = (long)nq * nv;
    if (pairs == 0) return DLDKD_OK;
    LAUNCH1D(clip_pool_bwd_kernel, pairs, 256, dpooled, arg, lens, dS, pairs, nv, L);
    return check_launch("clip_pool_bwd");
}
int dldkd_modpool_bwd_f32(const float* h, const float* mask, const float* w, const float* attn, const float* dout, float* dh,
                          float* dw, int N, int L, void* stream) {
    if (N < 0 || L < 1 || L > 64) { set_error("modpool_bwd: bad sizes"); return DLDKD_EINVAL; }
    if (N == 0) return DLDKD_OK;
    DLDKD_LAUNCH(modpool_bwd_kernel, dim3((unsigned)((N + kModGroups - 1) / kModGroups)), dim3(256 * kModGroups), 0, (hipStream_t)stream, h, mask, w, attn, dout, dh,
                 dw, N, L);
    return check_launch("modpool_bwd");
}

int dldkd_split2_bf16_jobs(const float* const* host_src, void* const* host_dst_h, void* const* host_dst_m, const long* host_n,
                           const int* host_kind, int njobs, void* stream) {
    if (njobs < 0 || njobs > 12 || (njobs && (!host_src || !host_dst_h || !host_dst_m || !host_n || !host_kind))) {
        set_error("split2_bf16_jobs: 0 .. 12 jobs with their host tables");
        return DLDKD_EINVAL;
    }
    if (njobs == 0) return DLDKD_OK;
    SplitJobs a{};
    a.njobs = njobs;
    long blocks = 0;
    for (int j = 0; j < njobs; ++j) {
        const int kind = host_kind[j];
        if (host_n[j] < 0 || (host_n[j] & 3) || (kind != 0 && kind != 1) || (host_n[j] && (!host_src[j] || !host_dst_h[j] || (kind == 0 && !host_dst_m[j]))) ||
            (((uintptr_t)host_src[j] | (kind == 1 ? (uintptr_t)host_dst_h[j] : 0)) & 15) || (kind == 0 && (((uintptr_t)host_dst_h[j] | (uintptr_t)host_dst_m[j]) & 7))) {
            set_error("split2_bf16_jobs: job %d: n must be a multiple of 4, fp32 pointers 16-byte and plane pointers 8-byte aligned, kind 0 (split) or 1 (copy)", j);
            return DLDKD_EINVAL;
        }
        a.src[j] = host_src[j]; a.dh[j] = host_dst_h[j]; a.dm[j] = (unsigned short*)host_dst_m[j]; a.n4[j] = host_n[j] / 4; a.kind[j] = kind;
        a.first[j] = blocks;
        blocks += (a.n4[j] + 255) / 256;
    }
    a.first[njobs] = blocks;
    if (blocks == 0) return DLDKD_OK;
    if (blocks > 0x7fffffffL) { set_error("split2_bf16_jobs: too many elements"); return DLDKD_EINVAL; }
    DLDKD_LAUNCH(split2_jobs_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a);
    return check_launch("split2_bf16_jobs");
}

int dldkd_tower_train_emit(const float* y0, const float* pos, int L, const float* stats1, const float* h1, const float* qkv,
                           const float* ctx, const float* dd, const float* stats2, const float* h2, const unsigned char* flags, long M,
                           void* xh1, void* relu_bits, void* h1d, void* qkv16, void* ctx16, void* xh2, float* rstd2, void* h2_16, void* stream) {
    if (M < 0 || L < 1 || (flags && (M & 31))) { set_error("tower_train_emit: bad sizes M=%ld L=%d", M, L); return DLDKD_EINVAL; }
    if (M == 0) return DLDKD_OK;
    if (!y0 || !pos || !stats1 || !h1 || !qkv || !dd || !stats2 || !xh1 || !relu_bits || !qkv16 || !xh2 || !rstd2 || (ctx16 && !ctx) ||
        (h2_16 && !h2)) {
        set_error("tower_train_emit: null pointer");
        return DLDKD_EINVAL;
    }
    if (((uintptr_t)y0 | (uintptr_t)pos | (uintptr_t)h1 | (uintptr_t)qkv | (uintptr_t)ctx | (uintptr_t)dd | (uintptr_t)h2) & 15 ||
        ((uintptr_t)xh1 | (uintptr_t)h1d | (uintptr_t)qkv16 | (uintptr_t)ctx16 | (uintptr_t)xh2 | (uintptr_t)h2_16) & 7) {
        set_error("tower_train_emit: unaligned buffer");
        return DLDKD_EINVAL;
    }
    EmitArgs a{y0, pos, stats1, h1, qkv, ctx, dd, stats2, h2, flags, (unsigned short*)xh1, (unsigned short*)h1d, (unsigned short*)qkv16,
               (unsigned short*)ctx16, (unsigned short*)xh2, (unsigned short*)h2_16, (unsigned char*)relu_bits, rstd2, M, L};
    LAUNCH1D(tower_train_emit_kernel, M * 96, 256, a);
    return check_launch("tower_train_emit");
}

}  // extern "C"
