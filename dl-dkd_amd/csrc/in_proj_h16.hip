// NOTE (round 5): the 16-bit MFMA operands of this file are h16 = IEEE fp16, not bf16 (common.hpp says why; the text below and the
// identifiers still say "bf16" where they mean "the 16-bit operand": bf16x8 is the 8 x 16-bit container, whatever the format).
// K4 in_proj: y = ReLU( LayerNorm(x) . W^T + b ) for the raw clip / word features, bf16 MFMA.
// Replaces LinearLayer.forward (reference method/model_components.py:305-312) on the inference path; it is
// the only stage that touches the raw fp32 features (Dv = 3072 for TVR i3d: 1.57 MB per 128-clip video),
// so it is priced against HBM bandwidth, not MFMA.
//
//   * LayerNorm is FOLDED: with W' = gamma (.) W,  LN(x).W^T + b = rstd * (x.W'^T - mean * colsum(W')) + (W.beta + b).
//     The kernel contracts the RAW rows with W' and applies mean / rstd in the epilogue; mean and E[x^2] are
//     accumulated in fp32 from the very tiles that feed the MFMAs, so x is read from HBM exactly once
//     (a separate LayerNorm pass would read and write it again).
//   * fp32 -> bf16 conversion happens in registers on the way to LDS (v_cvt_pk_bf16_f32).
//   * Both branches' weights are concatenated along N (768 outputs): the six 128-column tiles of one row
//     block are adjacent in blockIdx, run together and share the x tile through L2.
//   * 128x128x32 block tile, 4 waves (2x2) x 64x64, mfma_f32_32x32x16_bf16, global -> registers -> LDS one
//     k-tile ahead, LDS rows padded to 80 B so every ds_read_b128 fragment read is conflict-free.
//   Measured (MI355X, M = 400k rows, K = 3072, beyond the Infinity Cache): 4.9 ms = 1252 GB/s algorithmic (15.6 % of
//   8 TB/s), 384 TFLOP/s, 10x the fp32 parity path.  It is bound by neither HBM nor MFMA yet but by VALU: every one
//   of the six column-tile workgroups of a row block converts the same fp32 x tile to bf16 and re-accumulates the
//   LayerNorm sums (~2.5 VALU ops per element, six times).  A 128x256x64 retile (x converted 3x) was slower
//   (5.6 ms: one workgroup per CU).  Next: one workgroup owns all 768 columns of its rows (x converted once, W'
//   streamed from L2 in MFMA-fragment order by LDS-DMA).
#include "common.hpp"

namespace dldkd {

constexpr int PBM = 128, PBK = 32;
constexpr int PITCH = PBK + 8;   // bf16 elements per LDS row (80 B): ds_read_b128 fragment reads are conflict-free

// one wave per output row n: Wf[n,:] = bf16(gamma * W[n,:]); cs[n] = sum_k float(Wf[n,k]); bb[n] = W[n,:].beta + b[n]
__global__ __launch_bounds__(256) void fold_ln_linear_kernel(const float* __restrict__ W, const float* __restrict__ bias,
                                                             const float* __restrict__ gamma, const float* __restrict__ beta,
                                                             int N, int K, unsigned short* __restrict__ Wf,
                                                             float* __restrict__ cs, float* __restrict__ bb) {
    const int lane = threadIdx.x & 63;
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= N) return;
    float s = 0.f, t = 0.f;
    for (int k = lane; k < K; k += 64) {
        const float w = W[(size_t)n * K + k];
        const unsigned short h = f32_to_h16_bits(w * gamma[k]);
        Wf[(size_t)n * K + k] = h;
        s += h16_bits_to_f32(h);
        t += w * beta[k];
    }
    s = wave_sum(s);
    t = wave_sum(t);
    if (lane == 0) { cs[n] = s; bb[n] = t + (bias ? bias[n] : 0.f); }
}

struct InProjArgs {
    const float* x;
    const bf16x8* Wf;      // [N][K] bf16
    const float* cs;
    const float* bb;
    float* y[2];           // outputs: columns [0,384) -> y[0], [384,768) -> y[1]
    long M;
    int N, K;
    float eps;
    int relu;
    int ldy;               // row stride of y[*] in elements (full-row kernels; the tiled kernel writes 384-wide rows)
    int out_bf16;          // full-row kernels: y[*] are bf16 buffers (q|k|v for attention_fwd_bf16, which rounds them anyway)
};

// BN = 256 (two branches, N = 768: three column tiles per row block) or 128 (one branch, N = 384).
// 4 waves as 2 (rows) x 2 (cols); wave tile 64 x BN/2.
template <int BN>
__global__ __launch_bounds__(256) void in_proj_bf16_kernel(const InProjArgs p) {
    extern __shared__ __attribute__((aligned(16))) unsigned short lds_u16[];
    constexpr int NJ = BN / 64;                       // 32-column MFMA tiles per wave
    constexpr int WROWS = BN / 128;                   // W rows staged per thread pair... (BN=256: each thread a whole row)
    unsigned short* As0 = lds_u16;                    // [2][PBM * PITCH]
    unsigned short* Bs0 = lds_u16 + 2 * PBM * PITCH;  // [2][BN * PITCH]
    __shared__ float s_mean[PBM], s_rstd[PBM];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = (wave >> 1) * 64, wn = (wave & 1) * (BN / 2);
    const long m0 = (long)blockIdx.y * PBM;
    const int n0 = blockIdx.x * BN;
    const int nk = p.K / PBK;
    const int lrow = tid >> 1, lhalf = tid & 1;       // x: row lrow, k [16*lhalf, +16)

    f32x16 acc[2][NJ];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const bool row_ok = m0 + lrow < p.M;
    const float* xrow = p.x + (size_t)(m0 + (row_ok ? lrow : 0)) * p.K + lhalf * (PBK / 2);
    // W: BN = 256 -> thread t stages the whole k-tile row t; BN = 128 -> row t/2, half t%2
    const int wrow_i = BN == 256 ? tid : (tid >> 1);
    const int wk0 = BN == 256 ? 0 : (tid & 1) * (PBK / 2);
    constexpr int WV = (BN == 256 ? PBK : PBK / 2) / 8;   // 16-byte pieces per thread
    constexpr int XV = PBK / 8;                            // float4 pieces of x per thread
    const bf16x8* wrow = p.Wf + ((size_t)(n0 + wrow_i) * p.K + wk0) / 8;
    f32x4 rx[XV];
    bf16x8 rw[WV];
    float sum = 0.f, sq = 0.f;
    (void)WROWS;

    auto gload = [&](int kt) {
        const f32x4* src = reinterpret_cast<const f32x4*>(xrow + (size_t)kt * PBK);
#pragma unroll
        for (int i = 0; i < XV; ++i) rx[i] = row_ok ? src[i] : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < WV; ++i) rw[i] = wrow[(size_t)kt * (PBK / 8) + i];
    };
    auto lstore = [&](int buf) {
        unsigned short* a = As0 + buf * PBM * PITCH + lrow * PITCH + lhalf * (PBK / 2);
#pragma unroll
        for (int g = 0; g < XV / 2; ++g) {
            bf16x8 h;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float v = rx[2 * g + (e >> 2)][e & 3];
                sum += v;
                sq += v * v;
                h[e] = (short)f32_to_h16_bits(v);
            }
            *reinterpret_cast<bf16x8*>(a + 8 * g) = h;
        }
        unsigned short* b = Bs0 + buf * BN * PITCH + wrow_i * PITCH + wk0;
#pragma unroll
        for (int i = 0; i < WV; ++i) *reinterpret_cast<bf16x8*>(b + 8 * i) = rw[i];
    };

    gload(0);
    lstore(0);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) gload(kt + 1);
        const unsigned short* A = As0 + cur * PBM * PITCH;
        const unsigned short* B = Bs0 + cur * BN * PITCH;
#pragma unroll
        for (int kk = 0; kk < PBK / 16; ++kk) {
            bf16x8 a[2], b[NJ];
#pragma unroll
            for (int i = 0; i < 2; ++i)
                a[i] = *reinterpret_cast<const bf16x8*>(A + (wm + 32 * i + (lane & 31)) * PITCH + kk * 16 + (lane >> 5) * 8);
#pragma unroll
            for (int j = 0; j < NJ; ++j)
                b[j] = *reinterpret_cast<const bf16x8*>(B + (wn + 32 * j + (lane & 31)) * PITCH + kk * 16 + (lane >> 5) * 8);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j) acc[i][j] = h16_mfma32(a[i], b[j], acc[i][j]);
        }
        if (kt + 1 < nk) lstore(cur ^ 1);
        __syncthreads();
    }

    // LayerNorm statistics of the block's rows (two threads per row)
    sum += __shfl_xor(sum, 1);
    sq += __shfl_xor(sq, 1);
    if (lhalf == 0) {
        const float mean = sum / p.K;
        const float var = fmaxf(sq / p.K - mean * mean, 0.f);
        s_mean[lrow] = mean;
        s_rstd[lrow] = rsqrtf(var + p.eps);
    }
    __syncthreads();

#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int n = n0 + wn + 32 * j + (lane & 31);       // a 32-column tile never straddles the 384 boundary
        float* out = p.y[n / kHidden] + (n % kHidden);
        const float csn = p.cs[n], bbn = p.bb[n];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ml = wm + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (m0 + ml < p.M) {
                    float v = s_rstd[ml] * (acc[i][j][r] - s_mean[ml] * csn) + bbn;
                    if (p.relu) v = fmaxf(v, 0.f);
                    out[(size_t)(m0 + ml) * kHidden] = v;
                }
            }
    }
}


// ----------------------------------------------------------------------------------------------
// Full-row variant (two branches, N = 768): one 8-wave workgroup owns ALL output columns of its 128 rows.
//   * x is converted to bf16 and its LayerNorm sums are accumulated ONCE per row (the column-tiled kernel above
//     repeats both in each of its six column-tile workgroups and is VALU-bound by it);
//   * the folded weights are stored in MFMA B-fragment order ([k-tile][32-col tile][kk][lane][8]) so a k-tile of
//     W' (48 KiB) is 48 linear 1-KiB LDS-DMA pieces (6 per wave) and every fragment read is base + lane*16;
//   * wave w computes rows 0..127 x columns [96w, 96w+96): 4x3 MFMA tiles, 192 accumulator registers, two waves
//     per SIMD; per k-tile 24 MFMAs against 14 fragment reads.
//   Measured 3.02 ms at 400k x 3072 (2034 GB/s, 624 TFLOP/s).  Tried and dropped: a third W' slot with the DMA two
//   k-tiles ahead behind a counted vmcnt (x loads hidden in inline asm so hipcc does not drain the ring): 3.03 ms,
//   i.e. the L2 round trip of the W' tile is not what limits this kernel.
// ----------------------------------------------------------------------------------------------
constexpr int FBM = 128, FBK = 32, FN = 768;
constexpr int FPITCH = FBK + 8;                 // x image: 80-byte rows
constexpr int FW_TILE_BYTES = FN * FBK * 2;     // 48 KiB of W' per k-tile

__global__ __launch_bounds__(256) void fold_ln_linear_frag_kernel(const float* __restrict__ W, const float* __restrict__ bias,
                                                                  const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                  int N, int K, int n_offset, int fn, unsigned short* __restrict__ Wfrag,
                                                                  float* __restrict__ cs, float* __restrict__ bb) {
    const int lane = threadIdx.x & 63;
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= N) return;
    const int ng = n_offset + n, ct = ng >> 5, col = ng & 31;
    float s = 0.f, t = 0.f;
    for (int k = lane; k < K; k += 64) {
        const float w = W[(size_t)n * K + k];
        const unsigned short h = f32_to_h16_bits(gamma ? w * gamma[k] : w);
        const int kt = k >> 5, kk = (k >> 4) & 1, half = (k >> 3) & 1, j = k & 7;
        Wfrag[((((size_t)kt * (fn / 32) + ct) * 2 + kk) * 64 + half * 32 + col) * 8 + j] = h;
        s += h16_bits_to_f32(h);
        if (beta) t += w * beta[k];
    }
    s = wave_sum(s);
    t = wave_sum(t);
    if (lane == 0) { if (cs) cs[ng] = s; bb[ng] = t + (bias ? bias[n] : 0.f); }
}

// WR = 1: wave w owns rows 0..127 x columns [96w, 96w + 96) of N = 768 (the two-branch input projection).
// WR = 2: waves are 2 row groups x 4 column groups of N = 384 (64 rows x 96 columns each): the 384-wide linears of the
//         towers.  LNFOLD = false: plain y = x W^T + b (no LayerNorm statistics), bb = bias.
template <int WR, bool LNFOLD>
__global__ __launch_bounds__(512, 2) void rows_linear_bf16_kernel(const InProjArgs p) {
    constexpr int WCN = 8 / WR;                      // column groups
    constexpr int FN_ = 96 * WCN;                    // 768 or 384
    constexpr int W_TILE = FN_ * FBK * 2;            // bytes of W' per k-tile
    constexpr int PIECES = W_TILE / 1024 / 8;        // 1-KiB LDS-DMA pieces per wave
    constexpr int RT = 4 / WR;                       // 32-row tiles per wave
    extern __shared__ __attribute__((aligned(16))) char lds_full[];
    char* Wl = lds_full;                                            // [2][W_TILE]
    unsigned short* Al = reinterpret_cast<unsigned short*>(lds_full + 2 * W_TILE);   // [2][128 * FPITCH]
    float* s_mean = reinterpret_cast<float*>(lds_full + 2 * W_TILE + 2 * FBM * FPITCH * 2);
    float* s_rstd = s_mean + FBM;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave / WCN, wc = wave % WCN;
    const long m0 = (long)blockIdx.x * FBM;
    const int nk = p.K / FBK;
    const int xrow = tid >> 2, xq = tid & 3;                        // x: row xrow, floats [8*xq, 8*xq + 8) of the k-tile
    const bool row_ok = m0 + xrow < p.M;
    const float* xsrc = p.x + (size_t)(m0 + (row_ok ? xrow : 0)) * p.K + xq * 8;
    const char* wsrc = reinterpret_cast<const char*>(p.Wf);

    f32x16 acc[RT][3];
#pragma unroll
    for (int i = 0; i < RT; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    f32x4 rx[2];
    float sum = 0.f, sq = 0.f;
    auto wstage = [&](int kt, int buf) {                            // W_TILE / 1 KiB pieces, PIECES per wave
        const char* src = wsrc + (size_t)kt * W_TILE;
        char* dst = Wl + buf * W_TILE;
#pragma unroll
        for (int i = 0; i < PIECES; ++i) {
            const int piece = wave * PIECES + i;
            glds16(src + piece * 1024 + lane * 16, dst + piece * 1024);
        }
    };
    auto xload = [&](int kt) {
        const f32x4* src = reinterpret_cast<const f32x4*>(xsrc + (size_t)kt * FBK);
        rx[0] = row_ok ? src[0] : f32x4{0.f, 0.f, 0.f, 0.f};
        rx[1] = row_ok ? src[1] : f32x4{0.f, 0.f, 0.f, 0.f};
    };
    auto xstore = [&](int buf) {
        bf16x8 h;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float v = rx[e >> 2][e & 3];
            if constexpr (LNFOLD) {
                sum += v;
                sq += v * v;
            }
            h[e] = (short)f32_to_h16_bits(v);
        }
        *reinterpret_cast<bf16x8*>(Al + buf * FBM * FPITCH + xrow * FPITCH + xq * 8) = h;
    };

    wstage(0, 0);
    xload(0);
    xstore(0);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) { wstage(kt + 1, cur ^ 1); xload(kt + 1); }
        const unsigned short* A = Al + cur * FBM * FPITCH + wr * (FBM / WR) * FPITCH;
        const char* B = Wl + cur * W_TILE + lane * 16;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            bf16x8 b[3];
#pragma unroll
            for (int j = 0; j < 3; ++j)
                b[j] = *reinterpret_cast<const bf16x8*>(B + ((wc * 3 + j) * 2 + kk) * 1024);
#pragma unroll
            for (int i = 0; i < RT; ++i) {   // one A fragment live at a time: 192 accumulators leave little room
                const bf16x8 a = *reinterpret_cast<const bf16x8*>(A + (32 * i + (lane & 31)) * FPITCH + kk * 16 + (lane >> 5) * 8);
#pragma unroll
                for (int j = 0; j < 3; ++j) acc[i][j] = h16_mfma32(a, b[j], acc[i][j]);
            }
        }
        if (kt + 1 < nk) xstore(cur ^ 1);
        __syncthreads();
    }

    if constexpr (LNFOLD) {
        sum += __shfl_xor(sum, 1);
        sq += __shfl_xor(sq, 1);
        sum += __shfl_xor(sum, 2);
        sq += __shfl_xor(sq, 2);
        if (xq == 0) {
            const float mean = sum / p.K;
            const float var = fmaxf(sq / p.K - mean * mean, 0.f);
            s_mean[xrow] = mean;
            s_rstd[xrow] = rsqrtf(var + p.eps);
        }
        __syncthreads();
    }

    // Epilogue through LDS: the accumulator layout puts 32 consecutive COLUMNS of one row on 32 lanes, i.e. 128-byte
    // row segments as 4-byte stores.  Each wave instead parks one 32-row x 96-column tile at a time in its private LDS
    // region (the W' / x buffers are dead after the last barrier) and writes it back as float4: 24 lanes cover the
    // 384 contiguous bytes a row has in this wave's column range.
    constexpr int SP = 104;                                  // staging pitch (floats): the two lane halves hit disjoint banks
    float* stg = reinterpret_cast<float*>(lds_full) + wave * (32 * SP);
    const int n0w = wc * 96;                                 // 96 divides 384: a wave's columns belong to one output
    float* outb = p.y[n0w / kHidden] + (n0w % kHidden);
    float csn[3], bbn[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const int n = n0w + 32 * j + (lane & 31);
        csn[j] = LNFOLD ? p.cs[n] : 0.f;
        bbn[j] = p.bb[n];
    }
#pragma unroll
    for (int i = 0; i < RT; ++i) {
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int rl = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);       // row inside the 32-row tile
                float v;
                if constexpr (LNFOLD) {
                    const int ml = wr * (FBM / WR) + 32 * i + rl;
                    v = s_rstd[ml] * (acc[i][j][r] - s_mean[ml] * csn[j]) + bbn[j];
                } else {
                    v = acc[i][j][r] + bbn[j];
                }
                if (p.relu) v = fmaxf(v, 0.f);
                stg[rl * SP + 32 * j + (lane & 31)] = v;
            }
        if (p.out_bf16) {                                     // 8 columns = one 16-byte store per lane
            unsigned short* outh = reinterpret_cast<unsigned short*>(p.y[n0w / kHidden]) + (n0w % kHidden);
#pragma unroll
            for (int it = 0; it < 6; ++it) {
                const int idx = lane + 64 * it;              // 384 chunks of 8 columns per tile
                const int rl = idx / 12, c8 = idx % 12;
                const long mrow = m0 + wr * (FBM / WR) + 32 * i + rl;
                const f32x4 lo = *reinterpret_cast<const f32x4*>(stg + rl * SP + 8 * c8);
                const f32x4 hi = *reinterpret_cast<const f32x4*>(stg + rl * SP + 8 * c8 + 4);
                bf16x8 h;
#pragma unroll
                for (int e = 0; e < 4; ++e) { h[e] = (short)f32_to_bf16_bits(lo[e]); h[4 + e] = (short)f32_to_bf16_bits(hi[e]); }
                if (mrow < p.M) *reinterpret_cast<bf16x8*>(outh + (size_t)mrow * p.ldy + 8 * c8) = h;
            }
        } else {
#pragma unroll
            for (int it = 0; it < 12; ++it) {
                const int idx = lane + 64 * it;              // 768 float4 per tile
                const int rl = idx / 24, c4 = idx % 24;
                const long mrow = m0 + wr * (FBM / WR) + 32 * i + rl;
                const f32x4 v = *reinterpret_cast<const f32x4*>(stg + rl * SP + 4 * c4);
                if (mrow < p.M) *reinterpret_cast<f32x4*>(outb + (size_t)mrow * p.ldy + 4 * c4) = v;
            }
        }
    }
}

}  // namespace dldkd

using namespace dldkd;

extern "C" {

int dldkd_fold_ln_linear_h16(const float* W, const float* bias, const float* gamma, const float* beta, int N, int K,
                              void* Wf, float* cs, float* bb, void* stream) {
    if (N < 1 || K < 1) { set_error("fold_ln_linear: bad sizes"); return DLDKD_EINVAL; }
    if (!W || !gamma || !beta || !Wf || !cs || !bb) { set_error("fold_ln_linear: null pointer"); return DLDKD_EINVAL; }
    DLDKD_LAUNCH(fold_ln_linear_kernel, dim3((N + 3) / 4), dim3(256), 0, (hipStream_t)stream, W, bias, gamma, beta, N, K,
                       (unsigned short*)Wf, cs, bb);
    return check_launch("fold_ln_linear");
}

int dldkd_in_proj_h16(const float* x, const void* Wf, const float* cs, const float* bb, float* y0, float* y1, long M, int N,
                       int K, float eps, int relu, void* stream) {
    if (M < 0 || (N != kHidden && N != 2 * kHidden) || K < PBK || (K % PBK)) {
        set_error("in_proj_h16: need N = 384 or 768 and K a multiple of %d (got M=%ld N=%d K=%d)", PBK, M, N, K);
        return DLDKD_EINVAL;
    }
    if (M == 0) return DLDKD_OK;
    if (!x || !Wf || !cs || !bb || !y0 || (N == 2 * kHidden && !y1)) { set_error("in_proj_h16: null pointer"); return DLDKD_EINVAL; }
    InProjArgs p{x, (const bf16x8*)Wf, cs, bb, {y0, y1}, M, N, K, eps, relu, kHidden, 0};
    const unsigned rows = (unsigned)((M + PBM - 1) / PBM);
    // 128-column tiles (6 for two branches).  Measured at M = 400k, K = 3072: BN 128 / BK 32 = 4.9 ms (1252 GB/s,
    // 384 TFLOP/s); BN 256 / BK 64 = 5.6 ms (one workgroup per CU at 255 VGPRs).
    constexpr int lds = (2 * PBM * PITCH + 2 * 128 * PITCH) * 2;
    DLDKD_LAUNCH(in_proj_bf16_kernel<128>, dim3(N / 128, rows), dim3(256), lds, (hipStream_t)stream, p);
    return check_launch("in_proj_h16");
}


int dldkd_fold_ln_linear_h16_frag(const float* W, const float* bias, const float* gamma, const float* beta, int N, int K,
                                   int n_offset, void* Wfrag, float* cs, float* bb, void* stream) {
    if (N < 1 || K < FBK || (K % FBK) || n_offset < 0 || n_offset + N > FN || (n_offset % 32)) {
        set_error("fold_ln_linear_frag: need K a multiple of %d and columns inside [0, %d)", FBK, FN);
        return DLDKD_EINVAL;
    }
    if (!W || !gamma || !beta || !Wfrag || !cs || !bb) { set_error("fold_ln_linear_frag: null pointer"); return DLDKD_EINVAL; }
    DLDKD_LAUNCH(fold_ln_linear_frag_kernel, dim3((N + 3) / 4), dim3(256), 0, (hipStream_t)stream, W, bias, gamma, beta, N,
                       K, n_offset, FN, (unsigned short*)Wfrag, cs, bb);
    return check_launch("fold_ln_linear_frag");
}

int dldkd_pack_linear_h16_frag(const float* W, const float* bias, int N, int K, int n_offset, int n_total, void* Wfrag, float* bb,
                                void* stream) {
    if (N < 1 || K < FBK || (K % FBK) || (n_total != 384 && n_total != 768) || n_offset < 0 || n_offset + N > n_total ||
        (n_offset % 32)) {
        set_error("pack_linear_frag: need K a multiple of %d, n_total 384 or 768 and columns inside it (N=%d K=%d off=%d)", FBK, N, K,
                  n_offset);
        return DLDKD_EINVAL;
    }
    if (!W || !Wfrag || !bb) { set_error("pack_linear_frag: null pointer"); return DLDKD_EINVAL; }
    DLDKD_LAUNCH(fold_ln_linear_frag_kernel, dim3((N + 3) / 4), dim3(256), 0, (hipStream_t)stream, W, bias,
                       (const float*)nullptr, (const float*)nullptr, N, K, n_offset, n_total, (unsigned short*)Wfrag,
                       (float*)nullptr, bb);
    return check_launch("pack_linear_frag");
}

int dldkd_linear_rows_h16(const float* x, const void* Wfrag, const float* bb, void* y0, void* y1, int ldy, long M, int N, int K,
                           int relu, int out_bf16, void* stream) {
    if (M < 0 || (N != 384 && N != 768) || K < FBK || (K % FBK) || ldy < 384 || (out_bf16 && (ldy & 7))) {
        set_error("linear_rows_bf16: need N = 384 or 768, K a multiple of %d, ldy >= 384 (M=%ld N=%d K=%d ldy=%d)", FBK, M, N, K, ldy);
        return DLDKD_EINVAL;
    }
    if (M == 0) return DLDKD_OK;
    if (!x || !Wfrag || !bb || !y0 || (N == 768 && !y1)) { set_error("linear_rows_bf16: null pointer"); return DLDKD_EINVAL; }
    InProjArgs p{x, (const bf16x8*)Wfrag, nullptr, bb, {(float*)y0, (float*)y1}, M, N, K, 0.f, relu, ldy, out_bf16};
    const dim3 grid((unsigned)((M + FBM - 1) / FBM));
    if (N == 768) {
        constexpr int lds = 2 * FW_TILE_BYTES + 2 * FBM * FPITCH * 2 + 2 * FBM * 4;
        static const bool ok = [] { return hipFuncSetAttribute((const void*)rows_linear_bf16_kernel<1, false>,
                                                               hipFuncAttributeMaxDynamicSharedMemorySize, lds) == hipSuccess; }();
        (void)ok;
        DLDKD_LAUNCH((rows_linear_bf16_kernel<1, false>), grid, dim3(512), lds, (hipStream_t)stream, p);
    } else {
        constexpr int lds = 8 * 32 * 104 * 4;     // the epilogue staging (104 KiB) exceeds the k-loop's 68 KiB
        static const bool ok = [] { return hipFuncSetAttribute((const void*)rows_linear_bf16_kernel<2, false>,
                                                               hipFuncAttributeMaxDynamicSharedMemorySize, lds) == hipSuccess; }();
        (void)ok;
        DLDKD_LAUNCH((rows_linear_bf16_kernel<2, false>), grid, dim3(512), lds, (hipStream_t)stream, p);
    }
    return check_launch("linear_rows_bf16");
}

int dldkd_in_proj_h16_full(const float* x, const void* Wfrag, const float* cs, const float* bb, float* y0, float* y1, long M,
                            int K, float eps, int relu, void* stream) {
    if (M < 0 || K < FBK || (K % FBK)) { set_error("in_proj_bf16_full: K must be a multiple of %d", FBK); return DLDKD_EINVAL; }
    if (M == 0) return DLDKD_OK;
    if (!x || !Wfrag || !cs || !bb || !y0 || !y1) { set_error("in_proj_bf16_full: null pointer"); return DLDKD_EINVAL; }
    InProjArgs p{x, (const bf16x8*)Wfrag, cs, bb, {y0, y1}, M, FN, K, eps, relu, kHidden, 0};
    constexpr int lds = 2 * FW_TILE_BYTES + 2 * FBM * FPITCH * 2 + 2 * FBM * 4;
    static const bool ok = [] { return hipFuncSetAttribute((const void*)rows_linear_bf16_kernel<1, true>,
                                                           hipFuncAttributeMaxDynamicSharedMemorySize, lds) == hipSuccess; }();
    (void)ok;
    DLDKD_LAUNCH((rows_linear_bf16_kernel<1, true>), dim3((unsigned)((M + FBM - 1) / FBM)), dim3(512), lds, (hipStream_t)stream, p);
    return check_launch("in_proj_bf16_full");
}

}  // extern "C"
