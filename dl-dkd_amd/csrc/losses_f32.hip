// Loss kernels of DLDKD.forward (reference method/model.py:131-158), fp32, forward values and gradients.
// Latency-bound: the matrices are (Nq, Nv) <= ~1024 x 1024 and (Nq, L <= 128); each replaces a Python
// loop of the reference (640 host syncs + ~7k launches per step at the TVR batch).
//   kl_frame      compute_kl_loss(mode='frame_score')        model.py:183-197
//   nce_rows/cols clip_nce_soft.forward / clip_nce.forward   model_components.py:126-199 / :216-234
//   trip_t2v/v2t  get_clip_triplet_loss                      model.py:353-387
#include "common.hpp"

namespace dldkd {

// ---------------------------------------------------------------------------------------------
// KL( softmax(t/temp) || softmax(p/temp) ) over the first len clips of the query's own video, per query.
// Sp / St: (Nq, Nv, L) clip scores (student cosine / teacher cosine).  One wave per query, L <= 128.
// dSp (optional): += g * (softmax(p/temp) - softmax(t/temp)) / temp at [q, label_q, l < len].
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void kl_frame_body(int bid, const float* __restrict__ Sp, const float* __restrict__ St,
                                              const int32_t* __restrict__ labels, const int32_t* __restrict__ lens,
                                              float temp, int nq, int nv, int L, float* __restrict__ out,
                                              float* __restrict__ dSp, float g) {
    const int lane = threadIdx.x & 63;
    const int q = bid * 4 + (threadIdx.x >> 6);
    if (q >= nq) return;
    const int v = labels[q];
    const int n = lens[v];
    // nv == 0: the compact form - Sp / St are (nq, L), already the positive column of every query (simpool_train.hip)
    const size_t base = nv > 0 ? ((size_t)q * nv + v) * L : (size_t)q * L;
    float p[2], t[2];
    float mp = -INFINITY, mt = -INFINITY;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int c = lane + 64 * i;
        p[i] = c < n ? Sp[base + c] / temp : -INFINITY;
        t[i] = c < n ? St[base + c] / temp : -INFINITY;
        mp = fmaxf(mp, p[i]);
        mt = fmaxf(mt, t[i]);
    }
    mp = wave_max(mp);
    mt = wave_max(mt);
    float sp = 0.f, st = 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        if (lane + 64 * i < n) { sp += expf(p[i] - mp); st += expf(t[i] - mt); }
    }
    const float lsp = mp + logf(wave_sum(sp)), lst = mt + logf(wave_sum(st));
    float kl = 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int c = lane + 64 * i;
        if (c < n) {
            const float logp = p[i] - lsp, logt = t[i] - lst, tt = expf(logt);
            if (tt > 0.f) kl += tt * (logt - logp);          // xlogy: 0 where the target is 0
            if (dSp) dSp[base + c] += g * (expf(logp) - tt) / temp;
        }
    }
    kl = wave_sum(kl);
    if (lane == 0 && out) out[q] = kl;
}
__global__ __launch_bounds__(256) void kl_frame_kernel(const float* __restrict__ Sp, const float* __restrict__ St,
                                                       const int32_t* __restrict__ labels, const int32_t* __restrict__ lens,
                                                       float temp, int nq, int nv, int L, float* __restrict__ out,
                                                       float* __restrict__ dSp, const float* __restrict__ gp) {
    // upstream gradient: a device scalar (no host read-back, capturable)
    kl_frame_body(blockIdx.x, Sp, St, labels, lens, temp, nq, nv, L, out, dSp, gp ? *gp : 0.f);
}

// ---------------------------------------------------------------------------------------------
// Symmetric InfoNCE, row pass (text -> video).  One wave per query row.
//   IQ[q,:] = onehot(label_q)                                            (hard rows / clip_nce)
//           = (1-beta) * softmax(T[q,:]) + beta * onehot                 (soft rows: q >= hardQ)
//   term[q] = cq[q] * sum_v IQ[q,v] * (LSE(S[q,:]) - S[q,v])
//   dS[q,v]  = g*cq[q] * (sum(IQ) * softmax(S[q,:])[v] - IQ[q,v])
//   dT[q,u]  = g*cq[q] * (1-beta) * smT[u] * (c[u] - sum_v smT[v] c[v]),  c[v] = LSE - S[q,v]   (soft rows)
// ---------------------------------------------------------------------------------------------
// FOLD_T: the soft targets are the scores themselves (exploration branch, model.py:149-150): the gradient through the targets is added
// to dS instead of written to dT
template <bool FOLD_T = false>
__device__ __forceinline__ void nce_rows_body(int bid, const float* __restrict__ S, const float* __restrict__ T,
                                              const int32_t* __restrict__ labels, const float* __restrict__ cq,
                                              int hardQ, float beta, int nq, int nv, float* __restrict__ terms,
                                              float* __restrict__ dS, float* __restrict__ dT, float g) {
    const int lane = threadIdx.x & 63;
    const int q = bid * 4 + (threadIdx.x >> 6);
    if (q >= nq) return;
    const float* s = S + (size_t)q * nv;
    const bool soft = T != nullptr && q >= hardQ;
    const float* t = soft ? T + (size_t)q * nv : nullptr;
    const int lab = labels[q];
    float ms = -INFINITY, mt = -INFINITY;
    for (int v = lane; v < nv; v += 64) { ms = fmaxf(ms, s[v]); if (soft) mt = fmaxf(mt, t[v]); }
    ms = wave_max(ms);
    if (soft) mt = wave_max(mt);
    float zs = 0.f, zt = 0.f;
    for (int v = lane; v < nv; v += 64) { zs += expf(s[v] - ms); if (soft) zt += expf(t[v] - mt); }
    const float lse = ms + logf(wave_sum(zs));
    const float izt = soft ? 1.f / wave_sum(zt) : 0.f;
    float acc = 0.f, sumiq = 0.f, cbar = 0.f;
    for (int v = lane; v < nv; v += 64) {
        const float oh = v == lab ? 1.f : 0.f;
        const float sm = soft ? expf(t[v] - mt) * izt : 0.f;
        const float iq = soft ? fmaxf((1.f - beta) * sm + beta * oh, 0.f) : oh;
        const float c = lse - s[v];
        acc += iq * c;
        sumiq += iq;
        cbar += sm * c;
    }
    acc = wave_sum(acc);
    sumiq = wave_sum(sumiq);
    cbar = wave_sum(cbar);
    const float coef = cq[q];
    if (lane == 0 && terms) terms[q] = coef * acc;
    if (dS) {
        for (int v = lane; v < nv; v += 64) {
            const float oh = v == lab ? 1.f : 0.f;
            const float sm = soft ? expf(t[v] - mt) * izt : 0.f;
            const float iq = soft ? fmaxf((1.f - beta) * sm + beta * oh, 0.f) : oh;
            const float ds = g * coef * (sumiq * expf(s[v] - lse) - iq);
            const float dt = soft ? g * coef * (1.f - beta) * sm * ((lse - s[v]) - cbar) : 0.f;
            if constexpr (FOLD_T) {
                dS[(size_t)q * nv + v] = ds + dt;
            } else {
                dS[(size_t)q * nv + v] = ds;
                if (dT) dT[(size_t)q * nv + v] = dt;
            }
        }
    }
}
__global__ __launch_bounds__(256) void nce_rows_kernel(const float* __restrict__ S, const float* __restrict__ T,
                                                       const int32_t* __restrict__ labels, const float* __restrict__ cq,
                                                       int hardQ, float beta, int nq, int nv, float* __restrict__ terms,
                                                       float* __restrict__ dS, float* __restrict__ dT, const float* __restrict__ gp) {
    nce_rows_body<false>(blockIdx.x, S, T, labels, cq, hardQ, beta, nq, nv, terms, dS, dT, gp ? *gp : 0.f);
}

// ---------------------------------------------------------------------------------------------
// Column pass (video -> text).  One wave per video column; videos without a query contribute nothing.
//   IV[q] = onehot                                                       (hard columns / clip_nce)
//         = (1-beta) * softmax_q(T[:,v]) + beta * onehot                 (soft columns: v >= hardV)
//   term[v] = cv[v] * ( LSE_q S[q,v] - LSE_q( log(IV[q] + eps) + S[q,v] ) )
//   dS[q,v] += g*cv * (softmax_q(S[:,v])[q] - w[q]),  w = softmax_q(log(IV+eps) + S)
//   dT[u,v] += g*cv * (1-beta) * smT[u] * (-r[u] + sum_q smT[q] r[q]),  r = w / (IV + eps)       (soft columns)
// eps = 1e-12 for the soft loss (model_components.py:171), 0 for clip_nce (LSE over the positives only).
// ---------------------------------------------------------------------------------------------
// REG (nq <= 64 * kNceMaxQ): the lane's S / T / label entries of the column are loaded ONCE into registers - the column is read
// with a stride of nv floats (one cache line per lane), and the passes below depend on each other through wave reductions, so
// re-reading it in each of the 5-6 passes was 20-36 us of exposed latency per call at the TVR batch (640 x 128).  Same
// arithmetic in the same order as the memory form (the fallback for longer columns).
constexpr int kNceMaxQ = 16;
template <bool REG, bool FOLD_T = false>
__device__ __forceinline__ void nce_cols_body(int bid, const float* __restrict__ S, const float* __restrict__ T,
                                              const int32_t* __restrict__ labels, const float* __restrict__ cv,
                                              int hardV, float beta, float eps, int nq, int nv,
                                              float* __restrict__ terms, float* __restrict__ dS,
                                              float* __restrict__ dT, float g) {
    const int lane = threadIdx.x & 63;
    const int v = bid * 4 + (threadIdx.x >> 6);
    if (v >= nv) return;
    const bool soft = T != nullptr && v >= hardV;
    float sreg[REG ? kNceMaxQ : 1], treg[REG ? kNceMaxQ : 1];
    unsigned ohm = 0;                                   // bit i: labels[lane + 64 i] == v
    if constexpr (REG) {
#pragma unroll
        for (int i = 0; i < kNceMaxQ; ++i) {
            const int q = lane + 64 * i;
            sreg[i] = treg[i] = 0.f;
            if (q < nq) {
                sreg[i] = S[(size_t)q * nv + v];
                if (soft) treg[i] = T[(size_t)q * nv + v];
                ohm |= (labels[q] == v ? 1u : 0u) << i;
            }
        }
    }
    const int npass = REG ? kNceMaxQ : (nq + 63) / 64;
    auto Sv = [&](int i, int q) { if constexpr (REG) return sreg[i]; else return S[(size_t)q * nv + v]; };
    auto Tv = [&](int i, int q) { if constexpr (REG) return treg[i]; else return T[(size_t)q * nv + v]; };
    auto Oh = [&](int i, int q) { if constexpr (REG) return (ohm >> i) & 1u ? 1.f : 0.f; else return labels[q] == v ? 1.f : 0.f; };
    float ms = -INFINITY, mt = -INFINITY;
    int cnt = 0;
#pragma unroll
    for (int i = 0; i < npass; ++i) {
        const int q = lane + 64 * i;
        if (q < nq) {
            ms = fmaxf(ms, Sv(i, q));
            if (soft) mt = fmaxf(mt, Tv(i, q));
            cnt += Oh(i, q) != 0.f;
        }
    }
    ms = wave_max(ms);
    if (soft) mt = wave_max(mt);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o);
    if (cnt == 0) {   // label_dict has no entry for this video (model_components.py:169-180)
        if (lane == 0 && terms) terms[v] = 0.f;
        return;
    }
    float zs = 0.f, zt = 0.f;
#pragma unroll
    for (int i = 0; i < npass; ++i) {
        const int q = lane + 64 * i;
        if (q < nq) {
            zs += expf(Sv(i, q) - ms);
            if (soft) zt += expf(Tv(i, q) - mt);
        }
    }
    const float lse = ms + logf(wave_sum(zs));
    const float izt = soft ? 1.f / wave_sum(zt) : 0.f;
    // nominator LSE: max first
    float mn = -INFINITY;
#pragma unroll
    for (int i = 0; i < npass; ++i) {
        const int q = lane + 64 * i;
        if (q < nq) {
            const float oh = Oh(i, q);
            const float sm = soft ? expf(Tv(i, q) - mt) * izt : 0.f;
            const float iv = soft ? fmaxf((1.f - beta) * sm + beta * oh, 0.f) : oh;
            const float a = iv + eps;
            if (a > 0.f) mn = fmaxf(mn, logf(a) + Sv(i, q));
        }
    }
    mn = wave_max(mn);
    float zn = 0.f;
#pragma unroll
    for (int i = 0; i < npass; ++i) {
        const int q = lane + 64 * i;
        if (q < nq) {
            const float oh = Oh(i, q);
            const float sm = soft ? expf(Tv(i, q) - mt) * izt : 0.f;
            const float iv = soft ? fmaxf((1.f - beta) * sm + beta * oh, 0.f) : oh;
            const float a = iv + eps;
            if (a > 0.f) zn += expf(logf(a) + Sv(i, q) - mn);
        }
    }
    const float nom = mn + logf(wave_sum(zn));
    const float coef = cv[v];
    if (lane == 0 && terms) terms[v] = coef * (lse - nom);
    if (dS) {
        float rbar = 0.f;
        if (soft && (dT || FOLD_T)) {
#pragma unroll
            for (int i = 0; i < npass; ++i) {
                const int q = lane + 64 * i;
                if (q < nq) {
                    const float oh = Oh(i, q);
                    const float sm = expf(Tv(i, q) - mt) * izt;
                    const float a = fmaxf((1.f - beta) * sm + beta * oh, 0.f) + eps;
                    const float w = expf(logf(a) + Sv(i, q) - nom);
                    rbar += sm * (w / a);
                }
            }
            rbar = wave_sum(rbar);
        }
#pragma unroll
        for (int i = 0; i < npass; ++i) {
            const int q = lane + 64 * i;
            if (q < nq) {
                const size_t ix = (size_t)q * nv + v;
                const float oh = Oh(i, q);
                const float sq = Sv(i, q);
                const float sm = soft ? expf(Tv(i, q) - mt) * izt : 0.f;
                const float iv = soft ? fmaxf((1.f - beta) * sm + beta * oh, 0.f) : oh;
                const float a = iv + eps;
                const float w = a > 0.f ? expf(logf(a) + sq - nom) : 0.f;
                const float ds = g * coef * (expf(sq - lse) - w);
                if constexpr (FOLD_T) {
                    dS[ix] += ds + (soft ? g * coef * (1.f - beta) * sm * (-(w / a) + rbar) : 0.f);
                } else {
                    dS[ix] += ds;
                    if (soft && dT) dT[ix] += g * coef * (1.f - beta) * sm * (-(w / a) + rbar);
                }
            }
        }
    }
}
template <bool REG>
__global__ __launch_bounds__(256) void nce_cols_kernel(const float* __restrict__ S, const float* __restrict__ T,
                                                       const int32_t* __restrict__ labels, const float* __restrict__ cv,
                                                       int hardV, float beta, float eps, int nq, int nv,
                                                       float* __restrict__ terms, float* __restrict__ dS,
                                                       float* __restrict__ dT, const float* __restrict__ gp) {
    nce_cols_body<REG, false>(blockIdx.x, S, T, labels, cv, hardV, beta, eps, nq, nv, terms, dS, dT, gp ? *gp : 0.f);
}

// ---------------------------------------------------------------------------------------------
// Triplet, text -> video (model.py:372-385).  One wave per query: positive C[q,label]; negative = the
// r-th largest (1-based r = rsel[q]) of the OTHER entries of the row (the reference forces the positive
// to rank 0 with 999 and indexes the sorted row at r).  Selection by rank counting, no sort.
// term[q] = max(0, margin + neg - pos) * scale;  dC[q,neg] += g*scale, dC[q,label] -= g*scale when active.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void trip_t2v_body(int bid, float* sm, const float* __restrict__ C, const int32_t* __restrict__ labels,
                                              const int32_t* __restrict__ rsel, float margin, float scale, int nq,
                                              int nv, float* __restrict__ terms, float* __restrict__ dC, float g) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int q = bid * 4 + wave;
    float* row = sm + (size_t)wave * nv;
    if (q < nq) for (int v = lane; v < nv; v += 64) row[v] = C[(size_t)q * nv + v];
    __syncthreads();
    if (q >= nq) return;
    const int lab = labels[q];
    const int want = rsel[q] - 1;               // 0-based rank among the others
    const bool vec = (nv & 3) == 0;
    int found = -1;
    for (int v = lane; v < nv; v += 64) {
        if (v == lab) continue;
        const float x = row[v];
        int rank = 0;
        int u = 0;
        for (; u + 4 <= nv; u += 4) {           // 16-byte LDS reads (the row is 16-byte aligned: nv * 4 bytes per wave, nv % 4 == 0 checked by `vec`)
            if (!vec) break;
            const f32x4 y4 = *reinterpret_cast<const f32x4*>(row + u);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float y = y4[e];
                rank += (u + e != lab) && ((y > x) || (y == x && u + e < v));
            }
        }
        for (; u < nv; ++u) {
            if (u == lab) continue;
            const float y = row[u];
            rank += (y > x) || (y == x && u < v);
        }
        if (rank == want) found = v;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) found = max(found, __shfl_xor(found, o));
    if (lane == 0) {
        const float loss = found >= 0 ? margin + row[found] - row[lab] : 0.f;
        const bool act = loss > 0.f;
        if (terms) terms[q] = act ? loss * scale : 0.f;
        if (dC && act) {
            atomicAdd(dC + (size_t)q * nv + found, g * scale);
            atomicAdd(dC + (size_t)q * nv + lab, -g * scale);
        }
    }
}
__global__ __launch_bounds__(256) void trip_t2v_kernel(const float* __restrict__ C, const int32_t* __restrict__ labels,
                                                       const int32_t* __restrict__ rsel, float margin, float scale, int nq,
                                                       int nv, float* __restrict__ terms, float* __restrict__ dC, const float* __restrict__ gp) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    trip_t2v_body(blockIdx.x, sm, C, labels, rsel, margin, scale, nq, nv, terms, dC, gp ? *gp : 0.f);
}

// ---------------------------------------------------------------------------------------------
// Triplet, video -> text (model.py:360-369).  One workgroup per video i: positive = mean of C[q in i, i];
// negative = max (hard) or the rsel[i]-th largest (0-based) of C[q not in i, i].
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void trip_v2t_body(int bid, float* sm, const float* __restrict__ C, const int32_t* __restrict__ labels,
                                              const int32_t* __restrict__ rsel, int hard, float margin, float scale,
                                              int nq, int nv, float* __restrict__ terms, float* __restrict__ dC, float g) {
    __shared__ float red_s[4];
    __shared__ int red_i[4];
    __shared__ int sel;
    float* col = sm;
    const int i = bid, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int q = tid; q < nq; q += 256) col[q] = C[(size_t)q * nv + i];
    if (tid == 0) sel = -1;
    __syncthreads();
    float ps = 0.f;
    int pc = 0;
    for (int q = tid; q < nq; q += 256) if (labels[q] == i) { ps += col[q]; ++pc; }
    ps = wave_sum(ps);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) pc += __shfl_xor(pc, o);
    if (lane == 0) { red_s[wave] = ps; red_i[wave] = pc; }
    __syncthreads();
    const float psum = red_s[0] + red_s[1] + red_s[2] + red_s[3];
    const int pcnt = red_i[0] + red_i[1] + red_i[2] + red_i[3];
    if (hard) {
        // hardest negative = plain arg-max over the other queries' scores (ties: lowest index, like the rank rule)
        float best = -INFINITY;
        int bi = 0x7fffffff;
        for (int q = tid; q < nq; q += 256) {
            if (labels[q] == i) continue;
            const float x = col[q];
            if (x > best || (x == best && q < bi)) { best = x; bi = q; }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float ob = __shfl_xor(best, o);
            const int oi = __shfl_xor(bi, o);
            if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
        }
        __syncthreads();                       // red_s / red_i were read above by every thread
        if (lane == 0) { red_s[wave] = best; red_i[wave] = bi; }
        __syncthreads();
        if (tid == 0) {
            float b = red_s[0]; int k = red_i[0];
            for (int w = 1; w < 4; ++w) if (red_s[w] > b || (red_s[w] == b && red_i[w] < k)) { b = red_s[w]; k = red_i[w]; }
            sel = k == 0x7fffffff ? -1 : k;
        }
    } else {
        const int want = rsel[i];
        for (int q = tid; q < nq; q += 256) {
            if (labels[q] == i) continue;
            const float x = col[q];
            int rank = 0;
            for (int u = 0; u < nq; ++u) {
                if (labels[u] == i) continue;
                const float y = col[u];
                rank += (y > x) || (y == x && u < q);
            }
            if (rank == want) sel = q;
        }
    }
    __syncthreads();
    if (tid == 0) {
        const int n = sel;
        const float pos = psum / (float)pcnt;       // NaN when the video has no caption, like torch.mean([])
        const float loss = n >= 0 ? margin + col[n] - pos : 0.f;
        const bool act = loss > 0.f;
        if (terms) terms[i] = act ? loss * scale : (loss != loss ? loss : 0.f);
        if (dC && act) atomicAdd(dC + (size_t)n * nv + i, g * scale);
        red_i[0] = act ? 1 : 0;
    }
    __syncthreads();
    if (dC && red_i[0]) {
        for (int q = tid; q < nq; q += 256)
            if (labels[q] == i) atomicAdd(dC + (size_t)q * nv + i, -g * scale / (float)pcnt);
    }
}
__global__ __launch_bounds__(256) void trip_v2t_kernel(const float* __restrict__ C, const int32_t* __restrict__ labels,
                                                       const int32_t* __restrict__ rsel, int hard, float margin, float scale,
                                                       int nq, int nv, float* __restrict__ terms, float* __restrict__ dC,
                                                       const float* __restrict__ gp) {
    extern __shared__ float sm[];
    trip_v2t_body(blockIdx.x, sm, C, labels, rsel, hard, margin, scale, nq, nv, terms, dC, gp ? *gp : 0.f);
}

// ---------------------------------------------------------------------------------------------
// One branch's losses (model.py:137-155) with their gradients, three launches instead of ~18 + ATen glue:
//   A  [triplet t2v | triplet v2t | InfoNCE rows | KL]   independent of each other: one grid, the block index picks the part
//   B  [InfoNCE columns]                                   adds into the rows' dS
//   C  the three sums (fixed order), loss weights applied
// Values AND gradients in the same pass: the loss weights (inher / explore_nce_weight, kl_intra_weight * weight(epoch)) are host
// scalars of the step, the gradients are written for an upstream gradient of 1 and scaled by the actual one in the backward
// pass (branch_scale_kernel).
// ---------------------------------------------------------------------------------------------
struct BranchArgs {
    const float* C;              // (nq, nv) pooled cosines
    const float* S;              // (nq, nv) pooled raw scores
    const float* T;              // (nq, nv) soft-label scores (teacher) or null: hard labels / T = S (fold_t)
    const float* clip_p;         // (nq, L) positive-column clip cosines, student; null: no KL term
    const float* clip_t;         // (nq, L) teacher
    const int32_t* labels;
    const int32_t* lens;
    const int32_t* r_t2v;
    const int32_t* r_v2t;
    const float* cq;
    const float* cv;
    int nq, nv, L, hard, hardQ, hardV, fold_t;
    float margin, beta, eps, temp, w_nce, w_kl;
    float* terms;                // [nq + nv | nq + nv | nq]: triplet, InfoNCE, KL terms
    float* dC;                   // (nq, nv) zeroed by the caller (atomics)
    float* dS;                   // (nq, nv) written
    float* dclip;                // (nq, L) zeroed by the caller
    float* out;                  // [3]: triplet, w_nce * InfoNCE, w_kl * KL
    const int32_t* sched;        // null, or the per-step scalars read from the device: {hardQ, hardV, bits(beta), bits(w_kl), nq_valid}
    int nq_valid;                // rows [nq_valid, nq) of C / S / T / clip_* belong to padding queries (<= 0: none)
};

// The scalars an epoch's schedule moves (alpha -> hardQ / hardV and the coefficient vectors, belta, the KD weight: train.py:66-113).
// By value they are baked into a captured launch; behind `sched` a replayed graph follows the schedule without a new capture
// (the coefficient vectors cq / cv are rewritten in place by the same caller).
// nq_valid: a batch whose query axis was padded to a bucket (variable caption counts - Charades, ActivityNet - would otherwise give
// every batch its own graph): the matrices are row-major with row stride nv, so the bodies simply run over the first nq_valid rows
// (bounds, normalisers, the columns' softmax over queries); the padding rows' dS is written as zero, their terms are not summed.
struct SchedWords { int hardQ, hardV; float beta, w_kl; int nqv; };
__device__ __forceinline__ SchedWords sched_words(const BranchArgs& p) {
    SchedWords w{p.hardQ, p.hardV, p.beta, p.w_kl, p.nq_valid};
    if (p.sched != nullptr) {
        w.hardQ = p.sched[0]; w.hardV = p.sched[1];
        w.beta = __int_as_float(p.sched[2]); w.w_kl = __int_as_float(p.sched[3]);
        w.nqv = p.sched[4];
    }
    if (w.nqv <= 0 || w.nqv > p.nq) w.nqv = p.nq;
    return w;
}

__global__ __launch_bounds__(256) void branch_loss_a_kernel(const BranchArgs p) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int nb_q = (p.nq + 3) / 4;
    int b = blockIdx.x;
    const SchedWords w = sched_words(p);
    const int nqv = w.nqv;
    if (b < nb_q) { trip_t2v_body(b, sm, p.C, p.labels, p.r_t2v, p.margin, 1.f / nqv, nqv, p.nv, p.terms, p.dC, 1.f); return; }
    b -= nb_q;
    if (b < p.nv) { trip_v2t_body(b, sm, p.C, p.labels, p.r_v2t, p.hard, p.margin, 1.f / p.nv, nqv, p.nv, p.terms + p.nq, p.dC, 1.f); return; }
    b -= p.nv;
    float* nterms = p.terms + p.nq + p.nv;
    if (b < nb_q) {
        const int q = b * 4 + (threadIdx.x >> 6);
        if (q >= nqv) {                                     // a padding query's row of dS: zero (simpool's backward reads every row)
            if (q < p.nq) for (int v = threadIdx.x & 63; v < p.nv; v += 64) p.dS[(size_t)q * p.nv + v] = 0.f;
            return;
        }
        if (p.fold_t) nce_rows_body<true>(b, p.S, p.S, p.labels, p.cq, w.hardQ, w.beta, nqv, p.nv, nterms, p.dS, nullptr, p.w_nce);
        else nce_rows_body<false>(b, p.S, p.T, p.labels, p.cq, w.hardQ, w.beta, nqv, p.nv, nterms, p.dS, nullptr, p.w_nce);
        return;
    }
    b -= nb_q;
    if (p.clip_p != nullptr) kl_frame_body(b, p.clip_p, p.clip_t, p.labels, p.lens, p.temp, nqv, 0, p.L, p.terms + 2 * (p.nq + p.nv), p.dclip, w.w_kl);
}

template <bool REG>
__global__ __launch_bounds__(256) void branch_loss_b_kernel(const BranchArgs p) {
    float* nterms = p.terms + p.nq + p.nv + p.nq;
    const SchedWords w = sched_words(p);
    if (p.fold_t) nce_cols_body<REG, true>(blockIdx.x, p.S, p.S, p.labels, p.cv, w.hardV, w.beta, p.eps, w.nqv, p.nv, nterms, p.dS, nullptr, p.w_nce);
    else nce_cols_body<REG, false>(blockIdx.x, p.S, p.T, p.labels, p.cv, w.hardV, w.beta, p.eps, w.nqv, p.nv, nterms, p.dS, nullptr, p.w_nce);
}

// block k sums segment k of the terms in a fixed order (a segment = [nq per-query terms | nv per-video terms]; the terms of padding
// queries were never written and are skipped)
__global__ __launch_bounds__(256) void branch_loss_c_kernel(const BranchArgs p) {
    __shared__ float red[4];
    const int k = blockIdx.x, n2 = p.nq + p.nv;
    const SchedWords w = sched_words(p);
    const float* x = p.terms + (k == 0 ? 0 : k == 1 ? n2 : 2 * n2);
    const long n = k == 2 ? (p.clip_p != nullptr ? p.nq : 0) : n2;
    float s = 0.f;
    for (long i = threadIdx.x; i < n; i += 256)
        if (i < w.nqv || i >= p.nq) s += x[i];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) p.out[k] = (red[0] + red[1] + red[2] + red[3]) * (k == 0 ? 1.f : k == 1 ? p.w_nce : w.w_kl);
}

// dC *= g[0], dS *= g[1], dclip *= g[2] (g: the upstream gradients of the three loss terms, device scalars)
__global__ __launch_bounds__(256) void branch_scale_kernel(float* dC, float* dS, long n, float* dclip, long nk, const float* g0,
                                                           const float* g1, const float* g2) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) { dC[i] *= *g0; dS[i] *= *g1; }
    if (dclip != nullptr && i < nk) dclip[i] *= *g2;
}

// out[0] = sum x[0..n)   (single workgroup, deterministic order)
__global__ __launch_bounds__(256) void sum_kernel(const float* __restrict__ x, long n, float* __restrict__ out) {
    __shared__ float red[4];
    float s = 0.f;
    for (long i = threadIdx.x; i < n; i += 256) s += x[i];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) out[0] = red[0] + red[1] + red[2] + red[3];
}

}  // namespace dldkd

namespace dldkd {
struct ScalarPtrs { const float* p[8]; };
__global__ void sum_scalars_kernel(const ScalarPtrs a, int n, float* __restrict__ out) {
    if (threadIdx.x == 0) {
        float s = a.p[0][0];
        for (int i = 1; i < n; ++i) s += a.p[i][0];
        out[0] = s;
    }
}
}  // namespace dldkd

using namespace dldkd;

extern "C" {

int dldkd_kl_frame_f32(const float* Sp, const float* St, const int32_t* labels, const int32_t* lens, float temp, int nq,
                       int nv, int L, float* out, float* dSp, const float* g, void* stream) {
    if (nq < 0 || nv < 0 || L < 1 || L > 128 || temp <= 0.f) { set_error("kl_frame: bad sizes"); return DLDKD_EINVAL; }
    if (nq == 0) return DLDKD_OK;
    if (!Sp || !St || !labels || !lens) { set_error("kl_frame: null pointer"); return DLDKD_EINVAL; }
    DLDKD_LAUNCH(kl_frame_kernel, dim3((nq + 3) / 4), dim3(256), 0, (hipStream_t)stream, Sp, St, labels, lens, temp, nq,
                       nv, L, out, dSp, g);
    return check_launch("kl_frame");
}

int dldkd_nce_f32(const float* S, const float* T, const int32_t* labels, const float* cq, const float* cv, int hardQ,
                  int hardV, float beta, float eps, int nq, int nv, float* terms, float* dS, float* dT, const float* g,
                  void* stream) {
    if (nq < 1 || nv < 1) { set_error("nce: bad sizes"); return DLDKD_EINVAL; }
    if (!S || !labels || !cq || !cv) { set_error("nce: null pointer"); return DLDKD_EINVAL; }
    DLDKD_LAUNCH(nce_rows_kernel, dim3((nq + 3) / 4), dim3(256), 0, (hipStream_t)stream, S, T, labels, cq, hardQ, beta, nq,
                       nv, terms, dS, dT, g);
    if (nq <= 64 * kNceMaxQ)
        DLDKD_LAUNCH(nce_cols_kernel<true>, dim3((nv + 3) / 4), dim3(256), 0, (hipStream_t)stream, S, T, labels, cv, hardV, beta, eps,
                           nq, nv, terms ? terms + nq : nullptr, dS, dT, g);
    else
        DLDKD_LAUNCH(nce_cols_kernel<false>, dim3((nv + 3) / 4), dim3(256), 0, (hipStream_t)stream, S, T, labels, cv, hardV, beta, eps,
                           nq, nv, terms ? terms + nq : nullptr, dS, dT, g);
    return check_launch("nce");
}

int dldkd_triplet_f32(const float* C, const int32_t* labels, const int32_t* r_t2v, const int32_t* r_v2t, int hard,
                      float margin, int nq, int nv, float* terms, float* dC, const float* g, void* stream) {
    if (nq < 1 || nv < 1) { set_error("triplet: bad sizes"); return DLDKD_EINVAL; }
    if (!C || !labels || !r_t2v || (!hard && !r_v2t)) { set_error("triplet: null pointer"); return DLDKD_EINVAL; }
    if ((size_t)nv * 4 * sizeof(float) > 64 * 1024 || (size_t)nq * sizeof(float) > 64 * 1024) {
        set_error("triplet: batch too large for the LDS row/column buffers");
        return DLDKD_EINVAL;
    }
    DLDKD_LAUNCH(trip_t2v_kernel, dim3((nq + 3) / 4), dim3(256), (size_t)nv * 4 * sizeof(float), (hipStream_t)stream, C,
                       labels, r_t2v, margin, 1.f / nq, nq, nv, terms, dC, g);
    DLDKD_LAUNCH(trip_v2t_kernel, dim3(nv), dim3(256), (size_t)nq * sizeof(float), (hipStream_t)stream, C, labels, r_v2t,
                       hard, margin, 1.f / nv, nq, nv, terms ? terms + nq : nullptr, dC, g);
    return check_launch("triplet");
}

int dldkd_branch_losses_f32(const float* C, const float* S, const float* T, const float* clip_p, const float* clip_t,
                            const int32_t* labels, const int32_t* lens, const int32_t* r_t2v, const int32_t* r_v2t, const float* cq,
                            const float* cv, int nq, int nv, int L, int hard, int hardQ, int hardV, int fold_t, float margin, float beta,
                            float eps, float temp, float w_nce, float w_kl, float* terms, float* dC, float* dS, float* dclip, float* out,
                            int nq_valid, const int32_t* sched, void* stream) {
    if (nq < 1 || nv < 1 || (clip_p && (L < 1 || L > 128)) || temp <= 0.f) { set_error("branch_losses: bad sizes"); return DLDKD_EINVAL; }
    if (!C || !S || !labels || !r_t2v || (!hard && !r_v2t) || !cq || !cv || !terms || !dC || !dS || !out || (clip_p && (!clip_t || !lens || !dclip))) {
        set_error("branch_losses: null pointer");
        return DLDKD_EINVAL;
    }
    if ((size_t)nv * 4 * sizeof(float) > 64 * 1024 || (size_t)nq * sizeof(float) > 64 * 1024) {
        set_error("branch_losses: batch too large for the LDS row/column buffers");
        return DLDKD_EINVAL;
    }
    BranchArgs p{C, S, T, clip_p, clip_t, labels, lens, r_t2v, r_v2t, cq, cv, nq, nv, L, hard, hardQ, hardV, fold_t, margin, beta, eps,
                 temp, w_nce, w_kl, terms, dC, dS, dclip, out, sched, nq_valid};
    const int nb_q = (nq + 3) / 4;
    const size_t lds = sizeof(float) * (size_t)((nv * 4 > nq) ? nv * 4 : nq);
    hipStream_t s = (hipStream_t)stream;
    DLDKD_LAUNCH(branch_loss_a_kernel, dim3(nb_q + nv + nb_q + (clip_p ? nb_q : 0)), dim3(256), lds, s, p);
    if (nq <= 64 * kNceMaxQ) DLDKD_LAUNCH(branch_loss_b_kernel<true>, dim3((nv + 3) / 4), dim3(256), 0, s, p);
    else DLDKD_LAUNCH(branch_loss_b_kernel<false>, dim3((nv + 3) / 4), dim3(256), 0, s, p);
    DLDKD_LAUNCH(branch_loss_c_kernel, dim3(3), dim3(256), 0, s, p);
    return check_launch("branch_losses");
}

int dldkd_branch_losses_scale_f32(float* dC, float* dS, long n, float* dclip, long n_clip, const float* g_trip, const float* g_nce,
                                  const float* g_kl, void* stream) {
    if (n < 0 || n_clip < 0 || !dC || !dS || !g_trip || !g_nce || (dclip && !g_kl)) { set_error("branch_losses_scale: bad arguments"); return DLDKD_EINVAL; }
    const long m = n > n_clip ? n : n_clip;
    if (m == 0) return DLDKD_OK;
    DLDKD_LAUNCH(branch_scale_kernel, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, (hipStream_t)stream, dC, dS, n, dclip, n_clip,
                 g_trip, g_nce, g_kl);
    return check_launch("branch_losses_scale");
}

/* out[0] = ((((x0 + x1) + x2) + x3) + ...) of n <= 8 device scalars, left to right in fp32 - the order in which the reference adds
 * its loss terms (loss = inher_trip + inher_nce + kl + explore_trip + explore_nce, model.py:157-160); one launch instead of n - 1. */
int dldkd_sum_scalars_f32(const float* const* host_ptrs, int n, float* out, void* stream) {
    if (!host_ptrs || !out || n < 1 || n > 8) { set_error("sum_scalars: 1..8 scalars"); return DLDKD_EINVAL; }
    ScalarPtrs a{};
    for (int i = 0; i < n; ++i) {
        if (!host_ptrs[i]) { set_error("sum_scalars: null scalar"); return DLDKD_EINVAL; }
        a.p[i] = host_ptrs[i];
    }
    DLDKD_LAUNCH(sum_scalars_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, a, n, out);
    return check_launch("sum_scalars");
}

int dldkd_sum_f32(const float* x, long n, float* out, void* stream) {
    if (n < 0 || !out) { set_error("sum: bad arguments"); return DLDKD_EINVAL; }
    DLDKD_LAUNCH(sum_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, x, n, out);
    return check_launch("sum");
}

}  // extern "C"
