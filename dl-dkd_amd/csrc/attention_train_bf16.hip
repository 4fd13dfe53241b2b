// Throughput-mode training attention (reference method/model_components.py:398-436): the forward pass with dropout on the
// probabilities and the WHOLE backward pass as one kernel each, every product on the bf16 matrix cores
// (v_mfma_f32_32x32x16_bf16, fp32 accumulation) instead of the exact-fp32 MFMA of attention_train.hip (1/16 of the rate), and
// nothing but q|k|v, dO and the results crossing HBM: the probabilities are recomputed in the backward pass (two 24-MFMA
// products per 32 x 128 tile) instead of saved (33 MB fp32 per video tower at the TVR batch, written once and read twice), and
// dS never leaves the workgroup.
//
// One workgroup per (sequence, head) with one wave per 32 rows (L <= 128); for L <= 32 FOUR (sequence, head) pairs per
// workgroup, one per wave (the 30-word query towers were 2,560 single-wave workgroups).  q, k, v, dO of the head sit in LDS as
// bf16 [row][96] images (208-byte rows: the 16-byte fragment reads of 32 consecutive rows are conflict-free).
//
//   orientation A (forward, backward phase 1): S^T = K Q^T, keys on accumulator registers, queries on lanes: softmax over keys is
//     in-register, and the 8 consecutive accumulator registers of a lane ARE the B operand of the next product over keys
//     (O^T = V^T Pd^T, dQ^T = K^T dS^T) once its A operand is gathered in the same permuted key order
//     key(ks, h, j) = 16 ks + 8 (j >> 2) + 4 h + (j & 3)   (ks: 16-key step, h: lane half, j: element of the bf16x8)
//     - 2-byte LDS reads of a [row][d] image at one d, no transposed copies.
//   orientation B (backward phase 2): S = Q K^T, queries on registers, keys on lanes, the wave owns 32 keys: dS / Pd are the B
//     operands of dK^T = Q^T dS and dV^T = dO^T Pd (contraction over queries, same trick).  Row statistics (max, 1/sum, delta)
//     and the dropout keep bits come from phase 1 through LDS: Philox runs once per element, in orientation A where one call
//     serves the lane's four consecutive keys.
//
// Dropout masks: Philox4x32-10 on the flat index of P (N, 4, L, L) exactly as attention_train.hip and dldkd_dropout_fwd_f32
// draw them, so the two precision modes train with the same masks.
#include "common.hpp"

namespace dldkd {
namespace atb {

constexpr int kHeads = 4, kDh = 96, kLmax = 128;
constexpr int kPitchB = 208;                    // bytes per image row (96 bf16 + 16 B pad)
constexpr float kScale = 0.10206207261596577f;  // 1/sqrt(96), model_components.py:419

struct Args {
    const float* qkv;      // (N, L, 1152)   (IO16 kernels: bf16, as every tensor below but the mask)
    const float* mask;     // (N, L) or null
    float* out;            // forward: context (N, L, 384)
    const float* dout;     // backward: gradient of the context
    float* dqkv;           // backward: (N, L, 1152)
    const int* lens;       // IO16 kernels: valid rows per sequence or null (= L): rows past it are never read (the fused training
                           // towers do not write the rows of all-padding 32-row groups) and enter as zero rows
    int N, L, n_items;     // n_items = N * 4
    unsigned thresh;
    float dscale;
    int dropout;
    unsigned long long seed, offset;
    const unsigned long long* state;
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
__device__ __forceinline__ unsigned hswap_or(unsigned u) {
    auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    return (unsigned)r[0] | (unsigned)r[1];
}

// keep flags of the four consecutive keys kb..kb+3 of one query row (flat index idx0 of P[.., q, kb]) as bits 0..3
__device__ __forceinline__ unsigned keep_bits4(const Args& p, unsigned long long seed, unsigned long long off, size_t idx0) {
    if (!p.dropout) return 0xFu;
    unsigned rnd[4], bits = 0;
    if ((idx0 & 3) == 0) {
        const unsigned long long c = off + (idx0 >> 2);
        philox4x32_10((unsigned)c, (unsigned)(c >> 32), 0u, 0u, (unsigned)seed, (unsigned)(seed >> 32), rnd);
#pragma unroll
        for (int e = 0; e < 4; ++e) bits |= (rnd[e] >= p.thresh ? 1u : 0u) << e;
    } else {                                   // L not a multiple of 4: the four keys straddle two Philox calls
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const size_t idx = idx0 + e;
            const unsigned long long c = off + (idx >> 2);
            philox4x32_10((unsigned)c, (unsigned)(c >> 32), 0u, 0u, (unsigned)seed, (unsigned)(seed >> 32), rnd);
            bits |= (rnd[idx & 3] >= p.thresh ? 1u : 0u) << e;
        }
    }
    return bits;
}

// one head's rows of NIMG (N, L, ld) fp32 tensors -> bf16 [LP][96] images (rows >= L zero); `nthr` threads starting at `t`,
// LP * 24 / nthr = 12 16-byte pieces per thread and image whatever the tile count.  All loads of the call are issued before the
// first conversion: with one load per loop trip the fill was a chain of 48 exposed HBM round trips (100 us of the first version's
// 110-us backward kernels).
template <int NIMG>
__device__ __forceinline__ void fill_images(char* const (&img)[NIMG], const float* const (&src)[NIMG], const int (&ld)[NIMG], int L,
                                            int t, int nthr) {
    f32x4 v[NIMG][12];
#pragma unroll
    for (int j = 0; j < 12; ++j) {
        const int i = t + j * nthr, row = i / 24, c = i - row * 24;
#pragma unroll
        for (int m = 0; m < NIMG; ++m) {
            v[m][j] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (row < L) v[m][j] = *reinterpret_cast<const f32x4*>(src[m] + (size_t)row * ld[m] + c * 4);
        }
    }
#pragma unroll
    for (int j = 0; j < 12; ++j) {
        const int i = t + j * nthr, row = i / 24, c = i - row * 24;
#pragma unroll
        for (int m = 0; m < NIMG; ++m) {
            typedef unsigned short us4 __attribute__((ext_vector_type(4)));
            const us4 h = {f32_to_bf16_bits(v[m][j][0]), f32_to_bf16_bits(v[m][j][1]), f32_to_bf16_bits(v[m][j][2]),
                           f32_to_bf16_bits(v[m][j][3])};
            *reinterpret_cast<us4*>(img[m] + row * kPitchB + c * 8) = h;
        }
    }
}

// the same for tensors that are bf16 in memory (the fused training towers, tower_train.hip): 16-byte pieces, LP * 12 / nthr = 6 per
// thread and image; rows >= Lv (the sequence's valid rows) are zero rows and are not read
template <int NIMG>
__device__ __forceinline__ void fill_images16(char* const (&img)[NIMG], const unsigned short* const (&src)[NIMG], const int (&ld)[NIMG],
                                              int Lv, int t, int nthr) {
    bf16x8 v[NIMG][6];
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        const int i = t + j * nthr, row = i / 12, c = i - row * 12;
#pragma unroll
        for (int m = 0; m < NIMG; ++m) {
            v[m][j] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
            if (row < Lv) v[m][j] = *reinterpret_cast<const bf16x8*>(src[m] + (size_t)row * ld[m] + c * 8);
        }
    }
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        const int i = t + j * nthr, row = i / 12, c = i - row * 12;
#pragma unroll
        for (int m = 0; m < NIMG; ++m) *reinterpret_cast<bf16x8*>(img[m] + row * kPitchB + c * 16) = v[m][j];
    }
}

__device__ __forceinline__ bf16x8 frag(const char* img, int row, int ks, int h) {
    return *reinterpret_cast<const bf16x8*>(img + row * kPitchB + (16 * ks + 8 * h) * 2);
}

// A operand of a product over ROWS of an image (keys or queries) in the permuted order of the accumulator registers:
// element j = img[row0 + 16 ks + 8 (j >> 2) + 4 h + (j & 3)][d]
__device__ __forceinline__ bf16x8 gather(const char* img, int row0, int ks, int h, int d) {
    const char* b = img + (row0 + 16 * ks + 4 * h) * kPitchB + d * 2;
    bf16x8 a;
#pragma unroll
    for (int j = 0; j < 8; ++j) a[j] = *reinterpret_cast<const short*>(b + (8 * (j >> 2) + (j & 3)) * kPitchB);
    return a;
}

// 8 consecutive accumulator registers (one 16-row step of the tile) -> the B operand of the next product
__device__ __forceinline__ bf16x8 pack8(const f32x16& s, int ks) {
    bf16x8 b;
#pragma unroll
    for (int j = 0; j < 8; ++j) b[j] = (short)f32_to_bf16_bits(s[8 * ks + j]);
    return b;
}

__device__ __forceinline__ void store_rows_t(float* dst_row, const f32x16 (&o)[3], int h) {
    // o[dt] register r holds d = 32 dt + (r & 3) + 8 (r >> 2) + 4 h of this lane's row
#pragma unroll
    for (int dt = 0; dt < 3; ++dt)
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) {
            f32x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = o[dt][r4 * 4 + e];
            *reinterpret_cast<f32x4*>(dst_row + dt * 32 + 8 * r4 + 4 * h) = v;
        }
}

__device__ __forceinline__ void store_rows_t16(unsigned short* dst_row, const f32x16 (&o)[3], int h) {
#pragma unroll
    for (int dt = 0; dt < 3; ++dt)
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) {
            uint2 pk;
            pk.x = (unsigned)f32_to_bf16_bits(o[dt][r4 * 4]) | ((unsigned)f32_to_bf16_bits(o[dt][r4 * 4 + 1]) << 16);
            pk.y = (unsigned)f32_to_bf16_bits(o[dt][r4 * 4 + 2]) | ((unsigned)f32_to_bf16_bits(o[dt][r4 * 4 + 3]) << 16);
            *reinterpret_cast<uint2*>(dst_row + dt * 32 + 8 * r4 + 4 * h) = pk;
        }
}

struct Item {
    int n, head, active;
    char* lds;             // this item's LDS region
    int wave_in_item, t, nthr;
};

// NKT > 1: one item per workgroup, wave = 32-row tile.  NKT == 1: four items per workgroup, one per wave.
template <int NKT>
__device__ __forceinline__ Item locate(const Args& p, char* smem, int item_bytes) {
    Item it;
    const int wave = threadIdx.x >> 6;
    int item;
    if constexpr (NKT == 1) {
        item = blockIdx.x * 4 + wave;
        it.lds = smem + wave * item_bytes;
        it.wave_in_item = 0;
        it.t = threadIdx.x & 63;
        it.nthr = 64;
    } else {
        item = blockIdx.x;
        it.lds = smem;
        it.wave_in_item = wave;
        it.t = threadIdx.x;
        it.nthr = blockDim.x;
    }
    it.active = item < p.n_items;
    if (!it.active) item = p.n_items - 1;
    it.n = item / kHeads;
    it.head = item - it.n * kHeads;
    return it;
}

// S^T tiles of orientation A: s[kt] = K[kt] Q[q-tile]^T (fp32 accumulators), then softmax over keys in registers.
// Returns the row maximum and 1 / sum; s holds exp(. - max) (NOT yet divided).
template <int NKT>
__device__ __forceinline__ void scores_a(const char* Qi, const char* Ki, const float* Ms, int q0, int lane, f32x16 (&s)[NKT],
                                         float& mx, float& inv) {
    const int r32 = lane & 31, h = lane >> 5;
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
        for (int r = 0; r < 16; ++r) s[kt][r] = 0.f;
#pragma unroll
    for (int ks = 0; ks < 6; ++ks) {
        const bf16x8 b = frag(Qi, q0 + r32, ks, h);
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt) s[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag(Ki, kt * 32 + r32, ks, h), b, s[kt], 0, 0, 0);
    }
    mx = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            s[kt][r] = s[kt][r] * kScale + Ms[key];
            mx = fmaxf(mx, s[kt][r]);
        }
    mx = hswap_max(mx);
    float sum = 0.f;
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
        for (int r = 0; r < 16; ++r) { s[kt][r] = __expf(s[kt][r] - mx); sum += s[kt][r]; }
    inv = 1.f / hswap_sum(sum);
}

// ---------------------------------------------------------------------------------------------------------------- forward
template <int NKT, bool IO16 = false>
__device__ __forceinline__ void fwd_body(const Args& p, char* smem) {
    constexpr int LP = NKT * 32;
    constexpr int kImg = LP * kPitchB;
    constexpr int kItem = 3 * kImg + LP * 4;
    const Item it = locate<NKT>(p, smem, kItem);
    const int lane = threadIdx.x & 63, r32 = lane & 31, h = lane >> 5, L = p.L;
    char* Qi = it.lds;
    char* Ki = Qi + kImg;
    char* Vi = Ki + kImg;
    float* Ms = reinterpret_cast<float*>(Vi + kImg);
    const int Lv = (IO16 && p.lens != nullptr) ? min(p.lens[it.n], L) : L;
    if constexpr (IO16) {
        const unsigned short* base = reinterpret_cast<const unsigned short*>(p.qkv) + (size_t)it.n * L * (3 * kHidden) + it.head * kDh;
        char* const imgs[3] = {Qi, Ki, Vi};
        const unsigned short* const srcs[3] = {base, base + kHidden, base + 2 * kHidden};
        const int lds_[3] = {3 * kHidden, 3 * kHidden, 3 * kHidden};
        fill_images16<3>(imgs, srcs, lds_, Lv, it.t, it.nthr);
    } else {
        const float* base = p.qkv + (size_t)it.n * L * (3 * kHidden) + it.head * kDh;
        char* const imgs[3] = {Qi, Ki, Vi};
        const float* const srcs[3] = {base, base + kHidden, base + 2 * kHidden};
        const int lds_[3] = {3 * kHidden, 3 * kHidden, 3 * kHidden};
        fill_images<3>(imgs, srcs, lds_, L, it.t, it.nthr);
    }
    for (int i = it.t; i < LP; i += it.nthr)      // masked keys: the reference's additive -10000 (model_components.py:422)
        Ms[i] = i < L ? (p.mask ? (1.f - p.mask[(size_t)it.n * L + i]) * -10000.f : 0.f) : -INFINITY;
    __syncthreads();
    const int q0 = it.wave_in_item * 32;
    if (!it.active || q0 >= Lv) return;           // (IO16: the rows of a query tile past the sequence are not written)
    unsigned long long seed = p.seed, off = p.offset;
    if (p.state != nullptr) { seed = p.state[0]; off += p.state[1]; }

    f32x16 s[NKT];
    float mx, inv;
    scores_a<NKT>(Qi, Ki, Ms, q0, lane, s, mx, inv);
    const int q = q0 + r32;
    const size_t prow = (((size_t)it.n * kHeads + it.head) * L + (q < L ? q : 0)) * L;
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const unsigned kb = keep_bits4(p, seed, off, prow + kt * 32 + 8 * g + 4 * h);
#pragma unroll
            for (int e = 0; e < 4; ++e) s[kt][4 * g + e] *= ((kb >> e) & 1u) ? inv * p.dscale : 0.f;
        }
    // O^T[d][q] = sum_key V^T[d][key] Pd^T[key][q]
    f32x16 o[3];
#pragma unroll
    for (int dt = 0; dt < 3; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[dt][r] = 0.f;
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const bf16x8 b = pack8(s[kt], ks);
#pragma unroll
            for (int dt = 0; dt < 3; ++dt)
                o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(gather(Vi, kt * 32, ks, h, dt * 32 + r32), b, o[dt], 0, 0, 0);
        }
    if constexpr (IO16) {
        if (q < L) store_rows_t16(reinterpret_cast<unsigned short*>(p.out) + ((size_t)it.n * L + q) * kHidden + it.head * kDh, o, h);
    } else {
        if (q < L) store_rows_t(p.out + ((size_t)it.n * L + q) * kHidden + it.head * kDh, o, h);
    }
}

template <int NKT, bool IO16 = false>
__global__ __launch_bounds__(NKT == 1 ? 256 : 64 * NKT) void attn_bf16_fwd_kernel(const Args p) {
    extern __shared__ __attribute__((aligned(16))) char smem_c[];
    fwd_body<NKT, IO16>(p, smem_c);
}

// --------------------------------------------------------------------------------------------------------------- backward
template <int NKT, bool IO16 = false>
__device__ __forceinline__ void bwd_body(const Args& p, char* smem) {
    constexpr int LP = NKT * 32;
    constexpr int kImg = LP * kPitchB;
    constexpr int kItem = 4 * kImg + LP * 4 * 4 + LP * NKT * 4;    // images, Ms / max / inv / delta, keep bits
    const Item it = locate<NKT>(p, smem, kItem);
    const int lane = threadIdx.x & 63, r32 = lane & 31, h = lane >> 5, L = p.L;
    char* Qi = it.lds;
    char* Ki = Qi + kImg;
    char* Vi = Ki + kImg;
    char* Gi = Vi + kImg;                                          // dO
    float* Ms = reinterpret_cast<float*>(Gi + kImg);
    float* Mx = Ms + LP;
    float* Iv = Mx + LP;
    float* Dl = Iv + LP;
    unsigned* Kb = reinterpret_cast<unsigned*>(Dl + LP);           // [LP queries][NKT key tiles]: keep bit of key 32 kt + b
    const int Lv = (IO16 && p.lens != nullptr) ? min(p.lens[it.n], L) : L;
    if constexpr (IO16) {
        const unsigned short* base = reinterpret_cast<const unsigned short*>(p.qkv) + (size_t)it.n * L * (3 * kHidden) + it.head * kDh;
        char* const imgs[4] = {Qi, Ki, Vi, Gi};
        const unsigned short* const srcs[4] = {base, base + kHidden, base + 2 * kHidden,
                                               reinterpret_cast<const unsigned short*>(p.dout) + (size_t)it.n * L * kHidden + it.head * kDh};
        const int lds_[4] = {3 * kHidden, 3 * kHidden, 3 * kHidden, kHidden};
        fill_images16<4>(imgs, srcs, lds_, Lv, it.t, it.nthr);
    } else {
        const float* base = p.qkv + (size_t)it.n * L * (3 * kHidden) + it.head * kDh;
        char* const imgs[4] = {Qi, Ki, Vi, Gi};
        const float* const srcs[4] = {base, base + kHidden, base + 2 * kHidden, p.dout + (size_t)it.n * L * kHidden + it.head * kDh};
        const int lds_[4] = {3 * kHidden, 3 * kHidden, 3 * kHidden, kHidden};
        fill_images<4>(imgs, srcs, lds_, L, it.t, it.nthr);
    }
    for (int i = it.t; i < LP; i += it.nthr)
        Ms[i] = i < L ? (p.mask ? (1.f - p.mask[(size_t)it.n * L + i]) * -10000.f : 0.f) : -INFINITY;
    __syncthreads();
    const int t0 = it.wave_in_item * 32;          // this wave's query tile (phase 1) and key tile (phase 2)
    // IO16: tiles past the sequence do nothing - their dO rows are zero rows (dQ = dK = dV = 0 there) and the rows of dqkv they
    // would write are never read (tower_train.hip visits the same row groups)
    const bool work = it.active && t0 < Lv;
    unsigned long long seed = p.seed, off = p.offset;
    if (p.state != nullptr) { seed = p.state[0]; off += p.state[1]; }
    float* drow_base = p.dqkv + (size_t)it.n * L * (3 * kHidden) + it.head * kDh;
    unsigned short* drow_base16 = reinterpret_cast<unsigned short*>(p.dqkv) + (size_t)it.n * L * (3 * kHidden) + it.head * kDh;
    (void)drow_base; (void)drow_base16;

    // ---- phase 1: orientation A, queries t0.. on lanes
    if (work) {
        f32x16 s[NKT], dp[NKT];
        float mx, inv;
        scores_a<NKT>(Qi, Ki, Ms, t0, lane, s, mx, inv);
        // dPd^T[key][q] = sum_d V[key][d] dO[q][d]
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
            for (int r = 0; r < 16; ++r) dp[kt][r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < 6; ++ks) {
            const bf16x8 b = frag(Gi, t0 + r32, ks, h);
#pragma unroll
            for (int kt = 0; kt < NKT; ++kt)
                dp[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag(Vi, kt * 32 + r32, ks, h), b, dp[kt], 0, 0, 0);
        }
        const int q = t0 + r32;
        const size_t prow = (((size_t)it.n * kHeads + it.head) * L + (q < L ? q : 0)) * L;
        float delta = 0.f;
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt) {
            unsigned word = 0;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const unsigned kb = keep_bits4(p, seed, off, prow + kt * 32 + 8 * g + 4 * h);
                word |= kb << (8 * g + 4 * h);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float pv = s[kt][4 * g + e] * inv;               // P
                    s[kt][4 * g + e] = pv;
                    dp[kt][4 * g + e] *= ((kb >> e) & 1u) ? p.dscale : 0.f;  // dP = dPd (.) keep / (1 - p)
                    delta += dp[kt][4 * g + e] * pv;
                }
            }
            word = hswap_or(word);
            if (h == 0) Kb[q * NKT + kt] = word;
        }
        delta = hswap_sum(delta);
        if (h == 0) { Mx[q] = mx; Iv[q] = inv; Dl[q] = delta; }
        // dS^T = P (.) (dP - delta) / sqrt(96);  dQ^T[d][q] = sum_key K^T[d][key] dS^T[key][q]
        f32x16 o[3];
#pragma unroll
        for (int dt = 0; dt < 3; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[dt][r] = 0.f;
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt) {
#pragma unroll
            for (int r = 0; r < 16; ++r) s[kt][r] = s[kt][r] * (dp[kt][r] - delta) * kScale;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const bf16x8 b = pack8(s[kt], ks);
#pragma unroll
                for (int dt = 0; dt < 3; ++dt)
                    o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(gather(Ki, kt * 32, ks, h, dt * 32 + r32), b, o[dt], 0, 0, 0);
            }
        }
        if constexpr (IO16) { if (q < L) store_rows_t16(drow_base16 + (size_t)q * (3 * kHidden), o, h); }
        else { if (q < L) store_rows_t(drow_base + (size_t)q * (3 * kHidden), o, h); }
    }
    __syncthreads();
    if (!work) return;

    // ---- phase 2: orientation B, keys t0.. on lanes, queries tile by tile on registers
    const int key = t0 + r32;
    const float mkey = Ms[key];
    f32x16 dk[3], dv[3];
#pragma unroll
    for (int dt = 0; dt < 3; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) { dk[dt][r] = 0.f; dv[dt][r] = 0.f; }
    const int nqt = (Lv + 31) >> 5;               // (query tiles past the sequence: dO = 0 there, they add nothing)
    for (int qt = 0; qt < nqt; ++qt) {
        f32x16 s, dp;
#pragma unroll
        for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
        const int qrow = qt * 32 + r32;
#pragma unroll
        for (int ks = 0; ks < 6; ++ks) {
            s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag(Qi, qrow, ks, h), frag(Ki, key, ks, h), s, 0, 0, 0);
            dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag(Gi, qrow, ks, h), frag(Vi, key, ks, h), dp, 0, 0, 0);
        }
        // register r of this lane: query qt*32 + (r & 3) + 8 (r >> 2) + 4 h, key = this lane's
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int qb = qt * 32 + 8 * g + 4 * h;
            const f32x4 mx4 = *reinterpret_cast<const f32x4*>(Mx + qb);
            const f32x4 iv4 = *reinterpret_cast<const f32x4*>(Iv + qb);
            const f32x4 dl4 = *reinterpret_cast<const f32x4*>(Dl + qb);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const unsigned kw = Kb[(qb + e) * NKT + it.wave_in_item];
                const float keep = ((kw >> r32) & 1u) ? p.dscale : 0.f;
                const float pv = __expf(s[4 * g + e] * kScale + mkey - mx4[e]) * iv4[e];
                s[4 * g + e] = pv * (dp[4 * g + e] * keep - dl4[e]) * kScale;      // dS
                dp[4 * g + e] = pv * keep;                                          // Pd
            }
        }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const bf16x8 bs = pack8(s, ks), bp = pack8(dp, ks);
#pragma unroll
            for (int dt = 0; dt < 3; ++dt) {
                dk[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(gather(Qi, qt * 32, ks, h, dt * 32 + r32), bs, dk[dt], 0, 0, 0);
                dv[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(gather(Gi, qt * 32, ks, h, dt * 32 + r32), bp, dv[dt], 0, 0, 0);
            }
        }
    }
    if (key < L) {
        if constexpr (IO16) {
            unsigned short* krow = drow_base16 + (size_t)key * (3 * kHidden) + kHidden;
            store_rows_t16(krow, dk, h);
            store_rows_t16(krow + kHidden, dv, h);
        } else {
            float* krow = drow_base + (size_t)key * (3 * kHidden) + kHidden;
            store_rows_t(krow, dk, h);
            store_rows_t(krow + kHidden, dv, h);
        }
    }
}

template <int NKT, bool IO16 = false>
__global__ __launch_bounds__(NKT == 1 ? 256 : 64 * NKT) void attn_bf16_bwd_kernel(const Args p) {
    extern __shared__ __attribute__((aligned(16))) char smem_c[];
    bwd_body<NKT, IO16>(p, smem_c);
}

template <int NKT, bool IO16 = false>
static int launch(const Args& a, bool backward, hipStream_t s) {
    constexpr int LP = NKT * 32, kImg = LP * kPitchB;
    constexpr int per_item = (NKT == 1 ? 4 : 1);
    constexpr int lds_f = per_item * (3 * kImg + LP * 4);
    constexpr int lds_b = per_item * (4 * kImg + LP * 4 * 4 + LP * NKT * 4);
    const dim3 grid((a.n_items + per_item - 1) / per_item), block(NKT == 1 ? 256 : 64 * NKT);
    static const bool ok = [] {
        bool r = hipFuncSetAttribute((const void*)attn_bf16_fwd_kernel<NKT, IO16>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_f) == hipSuccess;
        r &= hipFuncSetAttribute((const void*)attn_bf16_bwd_kernel<NKT, IO16>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_b) == hipSuccess;
        return r;
    }();
    (void)ok;
    if (backward) { DLDKD_LAUNCH((attn_bf16_bwd_kernel<NKT, IO16>), grid, block, lds_b, s, a); }
    else { DLDKD_LAUNCH((attn_bf16_fwd_kernel<NKT, IO16>), grid, block, lds_f, s, a); }
    return DLDKD_OK;
}

static int fill_args(Args& a, int N, int L, float p_drop, unsigned long long seed, unsigned long long offset,
                     const unsigned long long* state, const char* what) {
    if (N < 0 || L < 1 || L > kLmax || !(p_drop >= 0.f && p_drop < 1.f)) {
        set_error("%s: bad sizes N=%d L=%d p=%f (L <= %d)", what, N, L, (double)p_drop, kLmax);
        return DLDKD_EINVAL;
    }
    a.N = N;
    a.L = L;
    a.n_items = N * kHeads;
    a.dropout = p_drop > 0.f;
    const double t = (double)p_drop * 4294967296.0;
    a.thresh = t >= 4294967295.0 ? 4294967295u : (unsigned)t;
    a.dscale = 1.0f / (1.0f - p_drop);
    a.seed = seed;
    a.offset = offset;
    a.state = state;
    return DLDKD_OK;
}

template <bool IO16 = false>
static int dispatch(const Args& a, bool backward, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    switch ((a.L + 31) >> 5) {
        case 1: launch<1, IO16>(a, backward, s); break;
        case 2: launch<2, IO16>(a, backward, s); break;
        case 3: launch<3, IO16>(a, backward, s); break;
        default: launch<4, IO16>(a, backward, s); break;
    }
    return check_launch(backward ? "attention_train_bwd_bf16" : "attention_train_fwd_bf16");
}

}  // namespace atb
}  // namespace dldkd

using namespace dldkd;

extern "C" {

int dldkd_attention_train_fwd_bf16(const float* qkv, const float* mask, float* out, int N, int L, float p_drop,
                                   unsigned long long seed, unsigned long long offset, const unsigned long long* state,
                                   void* stream) {
    atb::Args a{};
    const int rc = atb::fill_args(a, N, L, p_drop, seed, offset, state, "attention_train_fwd_bf16");
    if (rc != DLDKD_OK) return rc;
    if (N == 0) return DLDKD_OK;
    if (!qkv || !out) { set_error("attention_train_fwd_bf16: null pointer"); return DLDKD_EINVAL; }
    a.qkv = qkv; a.mask = mask; a.out = out;
    return atb::dispatch<false>(a, false, stream);
}

int dldkd_attention_train_bwd_bf16(const float* qkv, const float* mask, const float* dout, float* dqkv, int N, int L, float p_drop,
                                   unsigned long long seed, unsigned long long offset, const unsigned long long* state,
                                   void* stream) {
    atb::Args a{};
    const int rc = atb::fill_args(a, N, L, p_drop, seed, offset, state, "attention_train_bwd_bf16");
    if (rc != DLDKD_OK) return rc;
    if (N == 0) return DLDKD_OK;
    if (!qkv || !dout || !dqkv) { set_error("attention_train_bwd_bf16: null pointer"); return DLDKD_EINVAL; }
    a.qkv = qkv; a.mask = mask; a.dout = dout; a.dqkv = dqkv;
    return atb::dispatch<false>(a, true, stream);
}

// The same kernels with every tensor but the mask stored as bf16 (the fused training towers, tower_train.hip) and the sequences'
// valid lengths: rows past lens[n] are not read (they enter as zero rows) and the query / key tiles past it are skipped.
int dldkd_attention_train_fwd_bf16io(const void* qkv, const float* mask, const int* lens, void* out, int N, int L, float p_drop,
                                     unsigned long long seed, unsigned long long offset, const unsigned long long* state,
                                     void* stream) {
    atb::Args a{};
    const int rc = atb::fill_args(a, N, L, p_drop, seed, offset, state, "attention_train_fwd_bf16io");
    if (rc != DLDKD_OK) return rc;
    if (N == 0) return DLDKD_OK;
    if (!qkv || !out || ((uintptr_t)qkv & 15) || ((uintptr_t)out & 7)) { set_error("attention_train_fwd_bf16io: null or unaligned pointer"); return DLDKD_EINVAL; }
    a.qkv = (const float*)qkv; a.mask = mask; a.lens = lens; a.out = (float*)out;
    return atb::dispatch<true>(a, false, stream);
}

int dldkd_attention_train_bwd_bf16io(const void* qkv, const float* mask, const int* lens, const void* dout, void* dqkv, int N, int L,
                                     float p_drop, unsigned long long seed, unsigned long long offset,
                                     const unsigned long long* state, void* stream) {
    atb::Args a{};
    const int rc = atb::fill_args(a, N, L, p_drop, seed, offset, state, "attention_train_bwd_bf16io");
    if (rc != DLDKD_OK) return rc;
    if (N == 0) return DLDKD_OK;
    if (!qkv || !dout || !dqkv || ((uintptr_t)qkv & 15) || ((uintptr_t)dout & 15) || ((uintptr_t)dqkv & 7)) {
        set_error("attention_train_bwd_bf16io: null or unaligned pointer");
        return DLDKD_EINVAL;
    }
    a.qkv = (const float*)qkv; a.mask = mask; a.lens = lens; a.dout = (const float*)dout; a.dqkv = (float*)dqkv;
    return atb::dispatch<true>(a, true, stream);
}

}  // extern "C"
