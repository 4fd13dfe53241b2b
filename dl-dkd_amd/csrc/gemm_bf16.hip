// gemm_bf16: C[M,N] = act(alpha * sum_k A(m,k) * B(n,k) + bias[n]) with fp32 operands in memory, bf16 MFMA
// (v_mfma_f32_32x32x16_bf16) and fp32 accumulation: the throughput-mode twin of gemm_f32 (same argument
// meaning, same three operand layouts for Linear forward / dX / dW, same strided batching and split-K).
// Operands are converted fp32 -> bf16 in registers on the way to LDS; the LDS image is always [row][k]
// (80-byte rows: conflict-free ds_read_b128 fragments), so an operand whose contraction index is its row
// index in memory ("k-major": dX's W, both operands of dW) is transposed by its LDS write.
#include "common.hpp"

namespace dldkd {

#ifndef DLDKD_GEMM_BF16_BK
#define DLDKD_GEMM_BF16_BK 32   // 64 measured slower: 276-320 VGPRs -> one workgroup per CU (fwd 384x384: 46 vs 27 us)
#endif
constexpr int HBM_ = 128, HBN_ = 128, HBK_ = DLDKD_GEMM_BF16_BK;
constexpr int KV_ = HBK_ / 4;            // float4 per tile row
constexpr int RPP_ = 256 / KV_;          // rows per pass of the k-minor loader
constexpr int NPASS_ = HBM_ / RPP_;      // passes
constexpr int NREG_ = NPASS_ * 4;        // staging floats per thread per operand (= HBM_ * HBK_ / 256)
constexpr int KPT_ = HBK_ / 2;           // k per thread of the k-major loader (one row, KPT_ consecutive k)
constexpr int HPITCH = HBK_ + 8;

struct GemmHArgs {
    const float* A;
    const float* B;
    const float* bias;
    float* C;
    int M, N, K, lda, ldb, ldc, relu;
    int a_vec, b_vec;
    int batch_inner;
    long sAo, sAi, sBo, sBi, sCo, sCi;
    float alpha;
    int split_k, k_tiles_per_split;
    const unsigned char* mflags;   // per 32 rows of A / C (k-minor A only, M % 128 == 0): 0 = rows of the padding - not loaded, not
                                   // multiplied; their accumulators stay zero (C rows = act(bias), LayerNorm-gradient terms 0).  null: all
    const unsigned char* kflags;   // per k-tile (32 consecutive k): 0 = every operand row of the tile is padding (exact zeros in one
                                   // operand): the tile is skipped.  null: all tiles.  Used by the dW layout only
    float* a_colsum;               // k-major A only (the dW layout, A = dY): a_colsum[m] += sum_k A[k, m] over the k-tiles this workgroup
                                   // visits (atomics; zeroed by the caller) - the bias gradient, taken from the A tiles on their way to
                                   // LDS by the workgroups of the first column tile.  null: not wanted
};

// One operand tile: 128 rows x 32 k, 16 floats per thread.
//   k-minor memory [row][k]: 4 float4 loads (8 threads cover a row's 128 B), packed 8-byte LDS writes.
//   k-major memory [k][row]: the thread owns ONE row and 16 consecutive k (dword loads, coalesced across the wave
//     along the row index), so the transposed LDS write is two 16-byte stores - no 2-byte scatter, no conflicts.
// SRC16: the operand sits in memory as bf16 already (the LayerNorm-dropout output of the training input projection,
// dldkd_layernorm_dropout_bf16): half the bytes per tile, no rounding here (bf16 -> fp32 -> bf16 is exact).
//   k-minor bf16: same thread -> (row, 4 k) map, 8-byte loads.
//   k-major bf16: the thread owns TWO adjacent rows and 8 consecutive k (dword loads of a bf16 pair, coalesced along the row
//     index; two 16-byte LDS stores); needs an even ld and row count.
template <bool KMAJOR, bool SRC16 = false>
struct TileH {
    // pm (k-minor operands): bit j = pass j (rows 32 j .. 32 j + 31 of the tile) is wanted; the others are neither loaded nor stored
    static __device__ __forceinline__ void load(const float* __restrict__ P, int ld, int row0, int nrows, int k0, int K,
                                                int tid, bool vec, float (&r)[NREG_], unsigned pm = 0xFu) {
        if constexpr (SRC16) {
            const unsigned short* P16 = reinterpret_cast<const unsigned short*>(P);
            if constexpr (!KMAJOR) {
#pragma unroll
                for (int j = 0; j < NPASS_; ++j) {
                    if (!((pm >> j) & 1u)) continue;
                    const int row = row0 + tid / KV_ + RPP_ * j;
                    const int k = k0 + (tid % KV_) * 4;
#pragma unroll
                    for (int e = 0; e < 4; ++e) r[4 * j + e] = (row < nrows && k + e < K) ? bf16_bits_to_f32(P16[(size_t)row * ld + k + e]) : 0.f;
                }
            } else {
                const int row = row0 + 2 * (tid & 63);
                const int kb = k0 + (tid >> 6) * 8;
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const bool kok = kb + i < K;
                    r[i] = (kok && row < nrows) ? bf16_bits_to_f32(P16[(size_t)(kb + i) * ld + row]) : 0.f;
                    r[8 + i] = (kok && row + 1 < nrows) ? bf16_bits_to_f32(P16[(size_t)(kb + i) * ld + row + 1]) : 0.f;
                }
            }
        } else if constexpr (!KMAJOR) {
#pragma unroll
            for (int j = 0; j < NPASS_; ++j) {
                if (!((pm >> j) & 1u)) continue;
                const int row = row0 + tid / KV_ + RPP_ * j;
                const int k = k0 + (tid % KV_) * 4;
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (row < nrows) {
                    const float* src = P + (size_t)row * ld + k;
                    if (vec && k + 3 < K) v = *reinterpret_cast<const f32x4*>(src);
                    else {
#pragma unroll
                        for (int e = 0; e < 4; ++e) if (k + e < K) v[e] = src[e];
                    }
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) r[4 * j + e] = v[e];
            }
        } else {
            const int row = row0 + (tid & 127);
            const int kb = k0 + (tid >> 7) * KPT_;
            const float* src = P + (size_t)kb * ld + row;
            const bool rok = row < nrows;
#pragma unroll
            for (int i = 0; i < KPT_; ++i) r[i] = (rok && kb + i < K) ? src[(size_t)i * ld] : 0.f;
        }
    }
    // Interior tile (all k in range, 16-byte aligned rows): branch-free.  Rows past the end are CLAMPED to the
    // last row instead of zeroed - they only feed accumulator rows/columns that are never stored.
    static __device__ __forceinline__ void load_fast(const float* __restrict__ P, int ld, int row0, int nrows, int k0,
                                                     int tid, float (&r)[NREG_], unsigned pm = 0xFu) {
        if constexpr (SRC16) {
            const unsigned short* P16 = reinterpret_cast<const unsigned short*>(P);
            if constexpr (!KMAJOR) {
#pragma unroll
                for (int j = 0; j < NPASS_; ++j) {
                    if (!((pm >> j) & 1u)) continue;
                    const int row = min(row0 + tid / KV_ + RPP_ * j, nrows - 1);
                    const uint2 v = *reinterpret_cast<const uint2*>(P16 + (size_t)row * ld + k0 + (tid % KV_) * 4);
                    r[4 * j + 0] = __builtin_bit_cast(float, v.x << 16); r[4 * j + 1] = __builtin_bit_cast(float, v.x & 0xffff0000u);
                    r[4 * j + 2] = __builtin_bit_cast(float, v.y << 16); r[4 * j + 3] = __builtin_bit_cast(float, v.y & 0xffff0000u);
                }
            } else {
                const int row = min(row0 + 2 * (tid & 63), nrows - 2);          // (nrows even, >= 2: entry point)
                const unsigned short* src = P16 + (size_t)(k0 + (tid >> 6) * 8) * ld + row;
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const unsigned v = *reinterpret_cast<const unsigned*>(src + (size_t)i * ld);
                    r[i] = __builtin_bit_cast(float, v << 16);
                    r[8 + i] = __builtin_bit_cast(float, v & 0xffff0000u);
                }
            }
        } else if constexpr (!KMAJOR) {
#pragma unroll
            for (int j = 0; j < NPASS_; ++j) {
                if (!((pm >> j) & 1u)) continue;
                const int row = min(row0 + tid / KV_ + RPP_ * j, nrows - 1);
                const f32x4 v = *reinterpret_cast<const f32x4*>(P + (size_t)row * ld + k0 + (tid % KV_) * 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) r[4 * j + e] = v[e];
            }
        } else {
            const int row = min(row0 + (tid & 127), nrows - 1);
            const float* src = P + (size_t)(k0 + (tid >> 7) * KPT_) * ld + row;
#pragma unroll
            for (int i = 0; i < KPT_; ++i) r[i] = src[(size_t)i * ld];
        }
    }
    static __device__ __forceinline__ void store(unsigned short* __restrict__ S, int tid, const float (&r)[NREG_], unsigned pm = 0xFu) {
        if constexpr (SRC16 && KMAJOR) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {                    // rows 2 p and 2 p + 1, k (tid >> 6) * 8 .. + 7
                unsigned short* dst = S + (2 * (tid & 63) + h) * HPITCH + (tid >> 6) * 8;
                uint4 pk;
                pk.x = (__builtin_bit_cast(unsigned, r[8 * h + 0]) >> 16) | (__builtin_bit_cast(unsigned, r[8 * h + 1]) & 0xffff0000u);
                pk.y = (__builtin_bit_cast(unsigned, r[8 * h + 2]) >> 16) | (__builtin_bit_cast(unsigned, r[8 * h + 3]) & 0xffff0000u);
                pk.z = (__builtin_bit_cast(unsigned, r[8 * h + 4]) >> 16) | (__builtin_bit_cast(unsigned, r[8 * h + 5]) & 0xffff0000u);
                pk.w = (__builtin_bit_cast(unsigned, r[8 * h + 6]) >> 16) | (__builtin_bit_cast(unsigned, r[8 * h + 7]) & 0xffff0000u);
                *reinterpret_cast<uint4*>(dst) = pk;
            }
        } else if constexpr (!KMAJOR) {
#pragma unroll
            for (int j = 0; j < NPASS_; ++j) {
                if (!((pm >> j) & 1u)) continue;
                unsigned short* dst = S + (tid / KV_ + RPP_ * j) * HPITCH + (tid % KV_) * 4;
                uint2 pk;
                pk.x = (unsigned)f32_to_bf16_bits(r[4 * j]) | ((unsigned)f32_to_bf16_bits(r[4 * j + 1]) << 16);
                pk.y = (unsigned)f32_to_bf16_bits(r[4 * j + 2]) | ((unsigned)f32_to_bf16_bits(r[4 * j + 3]) << 16);
                *reinterpret_cast<uint2*>(dst) = pk;
            }
        } else {
            unsigned short* dst = S + (tid & 127) * HPITCH + (tid >> 7) * KPT_;
#pragma unroll
            for (int h = 0; h < KPT_ / 8; ++h) {
                uint4 pk;
                pk.x = (unsigned)f32_to_bf16_bits(r[8 * h + 0]) | ((unsigned)f32_to_bf16_bits(r[8 * h + 1]) << 16);
                pk.y = (unsigned)f32_to_bf16_bits(r[8 * h + 2]) | ((unsigned)f32_to_bf16_bits(r[8 * h + 3]) << 16);
                pk.z = (unsigned)f32_to_bf16_bits(r[8 * h + 4]) | ((unsigned)f32_to_bf16_bits(r[8 * h + 5]) << 16);
                pk.w = (unsigned)f32_to_bf16_bits(r[8 * h + 6]) | ((unsigned)f32_to_bf16_bits(r[8 * h + 7]) << 16);
                *reinterpret_cast<uint4*>(dst + 8 * h) = pk;
            }
        }
    }
};

// EPI: 0 = store C (bias / ReLU / split-K plane), 1 = training simpool max-pool (PoolArgs), 2 = LayerNorm parameter gradients (LnGradArgs)
// DUAL (the dW layout of the training input projection, inproj_bwd below): a second accumulator set takes the SAME A tiles against
// the 0 / 1 mask [B != 0] of the B tiles (made while the tile is staged: one more LDS tile, no memory traffic) and goes to a second
// plane behind C's (C + M ldc).
template <bool A_KMAJOR, bool B_KMAJOR, int EPI, typename EArgs, bool A16 = false, bool B16 = false, bool DUAL = false>
__device__ __forceinline__ void gemm_bf16_body(GemmHArgs p, const EArgs* pa) {
    extern __shared__ __attribute__((aligned(16))) unsigned short lds_raw[];
    unsigned short (*lds)[2][HBM_ * HPITCH] = reinterpret_cast<unsigned short (*)[2][HBM_ * HPITCH]>(lds_raw);
    unsigned short (*ldm)[HBM_ * HPITCH] = reinterpret_cast<unsigned short (*)[HBM_ * HPITCH]>(lds_raw + 2 * 2 * HBM_ * HPITCH);   // DUAL: [2]
    // XCD-aware tile order (cdna_hip_programming.md T1, bijective form).  Workgroups are dealt round-robin over the 8 XCDs, each
    // with its own L2: in launch order the 3 (N = 384) or 9 (N = 1152) column tiles that share a 128-row block of A ran on
    // different XCDs and every one of them pulled the block through the fabric again (16,384 x 384 x 384: 100 MB moved for 50 MB
    // of operands + result, 16,384 x 1152 x 384: 300 MB for 100 MB - their measured times at ~5 TB/s).  Remapped, the workgroups
    // that share an XCD (equal id % 8: a group label, never used for correctness) take CONSECUTIVE tiles, column tile fastest, so
    // a row block's tiles - and for split-K launches the row tiles that share a B slab - run side by side behind one L2.
    const Tile3 bid = xcd_tile_order();
    if (p.split_k > 1) {
        p.C += (size_t)bid.z * p.M * p.ldc * (DUAL ? 2 : 1);      // this split's partial plane(s) in the workspace
    } else {
        const int zo = bid.z / p.batch_inner, zi = bid.z % p.batch_inner;
        p.A += zo * p.sAo + zi * p.sAi;
        p.B += zo * p.sBo + zi * p.sBi;
        p.C += zo * p.sCo + zi * p.sCi;
    }
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;
    const int m0 = bid.y * HBM_, n0 = bid.x * HBN_;
    const int nk_all = (p.K + HBK_ - 1) / HBK_;
    const int kt0 = p.split_k > 1 ? bid.z * p.k_tiles_per_split : 0;
    const int nk = p.split_k > 1 ? min(nk_all - kt0, p.k_tiles_per_split) : nk_all;
    if (nk <= 0) return;

    f32x16 acc[2][2], acc2[DUAL ? 2 : 1][DUAL ? 2 : 1];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                acc[i][j][r] = 0.f;
                if constexpr (DUAL) acc2[i][j][r] = 0.f;
            }
    auto store_mask = [&](unsigned short* S, const float (&rb_)[NREG_]) {
        if constexpr (DUAL) {
            float rm[NREG_];
#pragma unroll
            for (int i = 0; i < NREG_; ++i) rm[i] = rb_[i] != 0.f ? 1.f : 0.f;
            TileH<B_KMAJOR, B16>::store(S, tid, rm);
        }
    };

    float ra[NREG_], rb[NREG_];
#pragma unroll
    for (int i = 0; i < NREG_; ++i) ra[i] = 0.f;
    // which 32-row groups of this tile's A rows hold anything (wave-uniform 4-bit mask; all without flags)
    unsigned pm = 0xFu;
    if constexpr (!A_KMAJOR) {
        if (p.mflags != nullptr) {
            const unsigned w = *reinterpret_cast<const unsigned*>(p.mflags + (m0 >> 5));         // 4 flag bytes (M % 128 == 0: entry point)
            pm = ((w & 0xffu) ? 1u : 0u) | ((w & 0xff00u) ? 2u : 0u) | ((w & 0xff0000u) ? 4u : 0u) | ((w & 0xff000000u) ? 8u : 0u);
            pm = __builtin_amdgcn_readfirstlane(pm);
        }
    }
    // k-minor operands need 16-byte aligned rows for the fast path; k-major ones use dword loads (always fine)
    const bool fa = A_KMAJOR || p.a_vec, fb = B_KMAJOR || p.b_vec;
    auto load_tiles = [&](int k0) {
        if (k0 + HBK_ <= p.K && fa && fb) {
            TileH<A_KMAJOR, A16>::load_fast(p.A, p.lda, m0, p.M, k0, tid, ra, pm);
            TileH<B_KMAJOR, B16>::load_fast(p.B, p.ldb, n0, p.N, k0, tid, rb);
        } else {
            TileH<A_KMAJOR, A16>::load(p.A, p.lda, m0, p.M, k0, p.K, tid, p.a_vec, ra, pm);
            TileH<B_KMAJOR, B16>::load(p.B, p.ldb, n0, p.N, k0, p.K, tid, p.b_vec, rb);
        }
    };
    const bool rt_ok[2] = {((pm >> (wm / 32)) & 1u) != 0, ((pm >> (wm / 32 + 1)) & 1u) != 0};      // this wave's two 32-row tiles
    // k-tiles whose flag is 0 are skipped (neither loaded nor multiplied).  The flags of this workgroup's k-range become two
    // wave-uniform 64-bit masks up front (one coalesced byte load per lane), ranges of more than 128 tiles are not filtered.
    unsigned long long km0 = ~0ull, km1 = ~0ull;
    if (p.kflags != nullptr && nk <= 128) {
        km0 = __ballot(lane < nk ? p.kflags[kt0 + lane] != 0 : false);
        km1 = __ballot(64 + lane < nk ? p.kflags[kt0 + 64 + lane] != 0 : false);
    }
    auto tile_ok = [&](int kt) { return kt < 64 ? ((km0 >> kt) & 1ull) != 0 : kt < 128 ? ((km1 >> (kt - 64)) & 1ull) != 0 : true; };
    // bias gradient (a_colsum): every thread sums the A values it stages - k-major loaders own one (fp32) or two (bf16) columns of
    // A and 16 / 8 consecutive k of every tile
    float cs0 = 0.f, cs1 = 0.f;
    const bool want_cs = A_KMAJOR && p.a_colsum != nullptr && bid.x == 0;
    auto add_cs = [&]() {
        if constexpr (A_KMAJOR) {
            if (want_cs) {
                if constexpr (A16) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) { cs0 += ra[i]; cs1 += ra[8 + i]; }
                } else {
#pragma unroll
                    for (int i = 0; i < NREG_; ++i) cs0 += ra[i];
                }
            }
        }
    };
    bool cur_ok = tile_ok(0);
    if (cur_ok) {
        load_tiles(kt0 * HBK_);
        add_cs();
        TileH<A_KMAJOR, A16>::store(lds[0][0], tid, ra, pm);
        TileH<B_KMAJOR, B16>::store(lds[0][1], tid, rb);
        store_mask(ldm[0], rb);
    }
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        const bool nxt_ok = kt + 1 < nk && tile_ok(kt + 1);
        if (nxt_ok) { load_tiles((kt0 + kt + 1) * HBK_); add_cs(); }
        if (cur_ok) {
            const unsigned short* As = lds[cur][0];
            const unsigned short* Bs = lds[cur][1];
#pragma unroll
            for (int kk = 0; kk < HBK_ / 16; ++kk) {
                bf16x8 a[2], b[2], bm[2];
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    b[j] = *reinterpret_cast<const bf16x8*>(Bs + (wn + 32 * j + (lane & 31)) * HPITCH + kk * 16 + (lane >> 5) * 8);
                    if constexpr (DUAL) bm[j] = *reinterpret_cast<const bf16x8*>(ldm[cur] + (wn + 32 * j + (lane & 31)) * HPITCH + kk * 16 + (lane >> 5) * 8);
                }
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    if (!rt_ok[i]) continue;                 // a 32-row tile of padding: its LDS rows were never written
                    a[i] = *reinterpret_cast<const bf16x8*>(As + (wm + 32 * i + (lane & 31)) * HPITCH + kk * 16 + (lane >> 5) * 8);
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
                        if constexpr (DUAL) acc2[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], bm[j], acc2[i][j], 0, 0, 0);
                    }
                }
            }
        }
        if (nxt_ok) {
            TileH<A_KMAJOR, A16>::store(lds[cur ^ 1][0], tid, ra, pm);
            TileH<B_KMAJOR, B16>::store(lds[cur ^ 1][1], tid, rb);
            store_mask(ldm[cur ^ 1], rb);
        }
        __syncthreads();
        cur_ok = nxt_ok;
    }
    if constexpr (A_KMAJOR) {
        if (want_cs) {
            // (rows past M were clamped to the last row by the fast loader or zeroed by the slow one: only real columns are added)
            if constexpr (A16) {
                const int m = m0 + 2 * (tid & 63);
                if (m < p.M) atomicAdd(p.a_colsum + m, cs0);
                if (m + 1 < p.M) atomicAdd(p.a_colsum + m + 1, cs1);
            } else {
                const int m = m0 + (tid & 127);
                if (m < p.M) atomicAdd(p.a_colsum + m, cs0);
            }
        }
    }
    // (the k-loop ended with a barrier: every wave is done with the operand tiles, the LDS is free for staging)
    if constexpr (EPI == 1) gemm_pool_tile(acc, p, *pa, bid.z, n0, wm, wn, lane, wave, reinterpret_cast<float*>(lds_raw));
    else if constexpr (EPI == 2) gemm_lngrad_tile(acc, p, *pa, m0, n0, wm, wn, lane, wave, reinterpret_cast<float*>(lds_raw));
    else {
        gemm_store_tile(acc, p, m0, n0, wm, wn, lane, reinterpret_cast<float*>(lds_raw) + wave * (32 * 72));
        if constexpr (DUAL) {
            p.C += (size_t)p.M * p.ldc;
            gemm_store_tile(acc2, p, m0, n0, wm, wn, lane, reinterpret_cast<float*>(lds_raw) + wave * (32 * 72));
        }
    }
}

template <bool A_KMAJOR, bool B_KMAJOR>
__global__ __launch_bounds__(256) void gemm_bf16_kernel(GemmHArgs p) {
    gemm_bf16_body<A_KMAJOR, B_KMAJOR, 0, PoolArgs>(p, nullptr);
}
// operands that are bf16 in memory: forward (A = the bf16 LayerNorm-dropout rows) and dW (B = the same rows, k-major)
__global__ __launch_bounds__(256) void gemm_bf16_a16_kernel(GemmHArgs p) {
    gemm_bf16_body<false, false, 0, PoolArgs, true, false>(p, nullptr);
}
__global__ __launch_bounds__(256) void gemm_bf16_dw_b16_kernel(GemmHArgs p) {
    gemm_bf16_body<true, true, 0, PoolArgs, false, true>(p, nullptr);
}
// both operands bf16 and k-major: the weight gradients of the fused training towers (tower_train.hip: dY and X are saved bf16 rows)
__global__ __launch_bounds__(256) void gemm_bf16_dw_a16b16_kernel(GemmHArgs p) {
    gemm_bf16_body<true, true, 0, PoolArgs, true, true>(p, nullptr);
}
// The weight gradients of one fused training tower as ONE product (tower_train.hip): the output rows are 384-row blocks [Wo | Wd | Wq |
// Wk | Wv] (or without Wo), every block with its own operands - A = the gradient of that layer's output (fp32 or bf16 rows, a column
// range of them), B = that layer's saved input rows (bf16) - and all of them contracted over the same batch rows with the same
// split-K plan and k-tile filter: one launch + one reduce instead of three of each, and 45 output tiles per k-slice to fill the chip.
struct DwGroupArgs {
    GemmHArgs base;              // M = 384 n_blocks, N = 384, K = batch rows, ldb = ldc = 384
    const void* A[5];
    const void* B[5];
    int lda[5], acol[5], a16[5];
};
__global__ __launch_bounds__(256) void gemm_bf16_dw_group_kernel(DwGroupArgs g) {
    GemmHArgs p = g.base;
    const int blk = xcd_tile_order().y / (kHidden / HBM_);
    // global output row m = 384 blk + j is column acol + j of this block's A: shift the base pointer instead of the index
    const long shift = (long)g.acol[blk] - (long)kHidden * blk;
    p.lda = g.lda[blk];
    p.B = (const float*)g.B[blk];
    if (g.a16[blk]) {
        p.A = (const float*)((const unsigned short*)g.A[blk] + shift);
        gemm_bf16_body<true, true, 0, PoolArgs, true, true>(p, nullptr);
    } else {
        p.A = (const float*)g.A[blk] + shift;
        gemm_bf16_body<true, true, 0, PoolArgs, false, true>(p, nullptr);
    }
}
// Backward pass of the training input projection, bf16 mode (LinearLayer on raw features: LayerNorm -> Dropout -> Linear,
// method/model_components.py:294-312; the features need no gradient).  With z = keep s (xhat gamma + beta) the saved bf16 rows,
// dY the output gradient and mask = [z != 0]:
//     dW[n, k]   = sum_m dY[m, n] z[m, k]                                  (the weight gradient, as before)
//     H[n, k]    = s sum_m dY[m, n] mask[m, k]
//     dbeta[k]   = sum_m (dY W)[m, k] s mask[m, k]         = sum_n W[n, k] H[n, k]
//     dgamma[k]  = sum_m (dY W)[m, k] s mask[m, k] xhat[m, k] = (sum_n W[n, k] dW[n, k] - beta[k] dbeta[k]) / gamma[k]
// i.e. the (M x K) product dY W of dldkd_linear_lngrad - a GEMM with a 384-long contraction whose 50 M results were reduced on the
// spot, 191 us at the TVR batch - is REASSOCIATED into a second accumulator of the weight-gradient GEMM (same dY tiles, the mask
// made from the z tile on its way to LDS): one launch with a 16,384-long contraction does both.  A column with |gamma| below
// kSmallGamma cannot be recovered from z (it holds beta only): the finishing kernel recomputes those columns from x exactly.
__global__ __launch_bounds__(256, 2) void gemm_bf16_dw_dual_kernel(GemmHArgs p) {
    gemm_bf16_body<true, true, 0, PoolArgs, false, true, true>(p, nullptr);
}
constexpr float kSmallGamma = 0.05f;
struct InprojFinishArgs {
    const float* part;           // [split][2][N][K] partial planes (or split = 1: the planes themselves)
    const float* W;              // [N][K]
    float* dW;                   // [N][K]
    float* dg;                   // [K] accumulators (zeroed by the caller): sum_n W dW
    float* db;                   // [K]                                      sum_n W H  (unscaled)
    int split, N, K;
};
// grid (K / 256, N / 16): a workgroup = 256 columns x 16 rows n; lane = FOUR consecutive columns (16-byte accesses: the first version
// read 4 bytes per load and ran at 1.6 TB/s), wave w = rows 4 w .. 4 w + 3 of the 16: reduces the split planes, writes dW,
// accumulates the two dot products, which the four waves combine in LDS - ONE atomic per column and workgroup (the version with a
// workgroup per 4 rows issued 96 same-address atomics per column: 590 k of them at the TVR projection, most of its 26 us).
__global__ __launch_bounds__(256) void inproj_bwd_reduce_kernel(const InprojFinishArgs a) {
    __shared__ f32x4 red[2][3][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int k = (blockIdx.x * 64 + lane) * 4;
    const bool act = k < a.K;
    const size_t plane = (size_t)a.N * a.K;
    f32x4 sg = {0.f, 0.f, 0.f, 0.f}, sb = {0.f, 0.f, 0.f, 0.f};
    const int n_lo = blockIdx.y * 16 + wave * 4, n_hi = min(a.N, n_lo + 4);
    if (act) {
        for (int n = n_lo; n < n_hi; ++n) {
            const size_t at = (size_t)n * a.K + k;
            f32x4 dw = {0.f, 0.f, 0.f, 0.f}, h = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 8
            for (int s = 0; s < a.split; ++s) {
                dw += *reinterpret_cast<const f32x4*>(a.part + (size_t)s * 2 * plane + at);
                h += *reinterpret_cast<const f32x4*>(a.part + (size_t)s * 2 * plane + plane + at);
            }
            *reinterpret_cast<f32x4*>(a.dW + at) = dw;
            const f32x4 w = *reinterpret_cast<const f32x4*>(a.W + at);
            sg += w * dw;
            sb += w * h;
        }
    }
    if (wave > 0) { red[0][wave - 1][lane] = sg; red[1][wave - 1][lane] = sb; }
    __syncthreads();
    if (wave == 0 && act) {
#pragma unroll
        for (int w = 0; w < 3; ++w) { sg += red[0][w][lane]; sb += red[1][w][lane]; }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            atomicAdd(a.dg + k + e, sg[e]);
            atomicAdd(a.db + k + e, sb[e]);
        }
    }
}
struct InprojFinalArgs {
    float* dg;                   // in: sum_n W dW, out: dgamma
    float* db;                   // in: sum_n W H (unscaled), out: dbeta
    const float* gamma;
    const float* beta;
    float keep_scale;
    int K;
    // the exact path for columns with a small gamma
    const float* dy;             // [M][N]
    const float* W;              // [N][K]
    const float* x;              // [M][K]
    const unsigned char* keep;   // [M][K] or null
    const float* mean;
    const float* rstd;           // (0 for rows of the padding)
    long M;
    int N;
    // keep == null and thresh != 0: the dropout bits are drawn again - Philox4x32-10 on the flat element index exactly as
    // layernorm_kernel drew them (counter = offset + (row K + col) / 4, lane of the counter = col % 4; seed / offset from `state`
    // when given): the forward pass then writes no keep byte per element (50 MB per video tower at the TVR batch)
    unsigned thresh;
    unsigned long long seed, off;
    const unsigned long long* state;
};
__global__ __launch_bounds__(256) void inproj_bwd_final_kernel(const InprojFinalArgs a) {
    __shared__ float wcol[kHidden];
    __shared__ float red[2][4];
    const int k = blockIdx.x * 256 + threadIdx.x;
    unsigned long long seed = a.seed, off = a.off;
    if (a.state != nullptr) { seed = a.state[0]; off += a.state[1]; }
    bool small = false;
    if (k < a.K) {
        const float g = a.gamma[k], dbeta = a.keep_scale * a.db[k];
        small = fabsf(g) < kSmallGamma;
        a.db[k] = dbeta;
        if (!small) a.dg[k] = (a.dg[k] - a.beta[k] * dbeta) / g;
    }
    // (rare) columns whose gamma is too small to divide by: dz'[m] = dY[m, :] . W[:, k] row by row, against x and the keep bytes
    for (int w = 0; w < 4; ++w) {
        unsigned long long todo = __ballot(small);
        __shared__ unsigned long long todo_s[4];
        if ((threadIdx.x & 63) == 0) todo_s[threadIdx.x >> 6] = todo;
        __syncthreads();
        todo = todo_s[w];
        while (todo) {
            const int bit = __builtin_ctzll(todo);
            todo &= todo - 1;
            const int kc = blockIdx.x * 256 + 64 * w + bit;
            for (int n = threadIdx.x; n < a.N && n < kHidden; n += 256) wcol[n] = a.W[(size_t)n * a.K + kc];
            __syncthreads();
            float sg = 0.f, sb = 0.f;
            for (long m = threadIdx.x; m < a.M; m += 256) {
                const float rs = a.rstd[m];
                if (rs == 0.f) continue;
                float kp = a.keep_scale;
                if (a.keep != nullptr) {
                    kp = a.keep[(size_t)m * a.K + kc] ? a.keep_scale : 0.f;
                } else if (a.thresh != 0u) {
                    const unsigned long long ctr = off + ((unsigned long long)m * (unsigned)(a.K >> 2) + (unsigned)(kc >> 2));
                    unsigned rnd[4];
                    philox4x32_10((unsigned)ctr, (unsigned)(ctr >> 32), 0u, 0u, (unsigned)seed, (unsigned)(seed >> 32), rnd);
                    kp = rnd[kc & 3] >= a.thresh ? a.keep_scale : 0.f;
                }
                if (kp == 0.f) continue;
                float dz = 0.f;
                const float* dyr = a.dy + (size_t)m * a.N;
                for (int n = 0; n < a.N; ++n) dz += dyr[n] * wcol[n];
                dz *= kp;
                sb += dz;
                sg += dz * (a.x[(size_t)m * a.K + kc] - a.mean[m]) * rs;
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) { sg += __shfl_xor(sg, o); sb += __shfl_xor(sb, o); }
            if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = sg; red[1][threadIdx.x >> 6] = sb; }
            __syncthreads();
            if (threadIdx.x == 0) {
                a.dg[kc] = red[0][0] + red[0][1] + red[0][2] + red[0][3];
                a.db[kc] = red[1][0] + red[1][1] + red[1][2] + red[1][3];
            }
            __syncthreads();
        }
        __syncthreads();
    }
}
// training simpool: one video per blockIdx.z, max-pool epilogue (common.hpp, gemm_pool_tile)
__global__ __launch_bounds__(256) void gemm_bf16_pool_kernel(GemmHArgs p, PoolArgs pa) {
    gemm_bf16_body<false, false, 1, PoolArgs>(p, &pa);
}
// dz' = dY W (the dX layout) with the LayerNorm-parameter-gradient epilogue
__global__ __launch_bounds__(256, 2) void gemm_bf16_lngrad_kernel(GemmHArgs p, LnGradArgs la) {
    gemm_bf16_body<false, true, 2, LnGradArgs>(p, &la);
}

static int launch_gemm_h(GemmHArgs p, int batch, int a_kmajor, int b_kmajor, void* stream) {
    const dim3 grid((p.N + HBN_ - 1) / HBN_, (p.M + HBM_ - 1) / HBM_, batch), block(256);
    hipStream_t s = (hipStream_t)stream;
    constexpr size_t lds = sizeof(unsigned short) * 2 * 2 * HBM_ * HPITCH;
    static const bool attr_ok = [] {
        bool ok = true;
        ok &= hipFuncSetAttribute((const void*)gemm_bf16_kernel<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds) == hipSuccess;
        ok &= hipFuncSetAttribute((const void*)gemm_bf16_kernel<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds) == hipSuccess;
        ok &= hipFuncSetAttribute((const void*)gemm_bf16_kernel<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds) == hipSuccess;
        ok &= hipFuncSetAttribute((const void*)gemm_bf16_kernel<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds) == hipSuccess;
        return ok;
    }();
    (void)attr_ok;
    if (!a_kmajor && !b_kmajor) DLDKD_LAUNCH((gemm_bf16_kernel<false, false>), grid, block, lds, s, p);
    else if (!a_kmajor && b_kmajor) DLDKD_LAUNCH((gemm_bf16_kernel<false, true>), grid, block, lds, s, p);
    else if (a_kmajor && b_kmajor) DLDKD_LAUNCH((gemm_bf16_kernel<true, true>), grid, block, lds, s, p);
    else DLDKD_LAUNCH((gemm_bf16_kernel<true, false>), grid, block, lds, s, p);
    return check_launch("gemm_bf16");
}

static int launch_gemm_h_mixed(GemmHArgs p, int batch, int dw, void* stream) {
    const dim3 grid((p.N + HBN_ - 1) / HBN_, (p.M + HBM_ - 1) / HBM_, batch), block(256);
    constexpr size_t lds = sizeof(unsigned short) * 2 * 2 * HBM_ * HPITCH;
    static const bool attr_ok = [] {
        bool ok = hipFuncSetAttribute((const void*)gemm_bf16_a16_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds) == hipSuccess;
        ok &= hipFuncSetAttribute((const void*)gemm_bf16_dw_b16_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds) == hipSuccess;
        ok &= hipFuncSetAttribute((const void*)gemm_bf16_dw_a16b16_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds) == hipSuccess;
        return ok;
    }();
    (void)attr_ok;
    if (dw == 3) DLDKD_LAUNCH(gemm_bf16_dw_a16b16_kernel, grid, block, lds, (hipStream_t)stream, p);
    else if (dw) DLDKD_LAUNCH(gemm_bf16_dw_b16_kernel, grid, block, lds, (hipStream_t)stream, p);
    else DLDKD_LAUNCH(gemm_bf16_a16_kernel, grid, block, lds, (hipStream_t)stream, p);
    return check_launch("gemm_bf16_mixed");
}

int launch_simpool_pool_bf16(const float* g, const float* q, int nv, int L, int nq, int D, const PoolArgs& pa, void* stream) {
    const bool al = !(D & 3) && !((uintptr_t)g & 15), bl = !(D & 3) && !((uintptr_t)q & 15);
    GemmHArgs p{g, q, nullptr, nullptr, L, nq, D, D, D, nq, 0, al, bl, 1, (long)L * D, 0, 0, 0, 0, 0, 1.0f, 1, 0};
    constexpr size_t lds = sizeof(unsigned short) * 2 * 2 * HBM_ * HPITCH;
    static const bool attr_ok = hipFuncSetAttribute((const void*)gemm_bf16_pool_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds) == hipSuccess;
    (void)attr_ok;
    DLDKD_LAUNCH(gemm_bf16_pool_kernel, dim3((nq + HBN_ - 1) / HBN_, 1, nv), dim3(256), lds, (hipStream_t)stream, p, pa);
    return check_launch("simpool_train_fwd (bf16)");
}

}  // namespace dldkd

using namespace dldkd;

// Split-K plan of the unbatched entry point (backward layouts only: the forward pass, hence every loss, stays one
// k-ordered accumulation per element).  Returns the number of k-slices (1 = no split) and the k-tiles per slice.
int dldkd::gemm_bf16_split_plan(int M, int N, int K, int a_kmajor, int b_kmajor, int* k_tiles_per_split) {
    const int tiles = ((N + HBN_ - 1) / HBN_) * ((M + HBM_ - 1) / HBM_);
    const int nk = (K + HBK_ - 1) / HBK_;
    *k_tiles_per_split = nk;
    if (!(a_kmajor || b_kmajor) || (((long)M * N) & 3) || tiles >= 128 || nk < 16) return 1;
    // blocks to aim for: two workgroups fit a CU; more splits mean more partial planes to write and reduce.  Measured
    // (dW shapes of the C3 step): 384 is best for outputs of >= 400k elements (in-proj dW 158 -> 121 us), 256 below.
    const int target = (long)M * N >= 400000 ? 384 : 256;
    int split = (target + tiles - 1) / tiles;
    if (split > nk / 4) split = nk / 4;
    if (split <= 1) return 1;
    *k_tiles_per_split = (nk + split - 1) / split;
    return (nk + *k_tiles_per_split - 1) / *k_tiles_per_split;
}

extern "C" int dldkd_colsum_f32(const float* x, float* out, long M, long N, void* stream);

int dldkd::launch_linear_lngrad_bf16(const float* dy, const float* W, long M, int N, int K, const LnGradArgs& la, void* stream,
                                     const unsigned char* row_flags) {
    // C[M, K] = dy[M, N] . W[N, K]: contraction over the Linear's outputs, B in the k-major (dX) layout
    const int a_vec = !(N & 3) && !((uintptr_t)dy & 15);
    GemmHArgs p{dy, W, nullptr, nullptr, (int)M, K, N, N, K, K, 0, a_vec, 0, 1, 0, 0, 0, 0, 0, 0, 1.0f, 1, 0};
    p.mflags = (M % HBM_ == 0 && !((uintptr_t)row_flags & 3)) ? row_flags : nullptr;
    constexpr size_t lds = sizeof(unsigned short) * 2 * 2 * HBM_ * HPITCH;
    static const bool attr_ok = hipFuncSetAttribute((const void*)gemm_bf16_lngrad_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds) == hipSuccess;
    (void)attr_ok;
    DLDKD_LAUNCH(gemm_bf16_lngrad_kernel, dim3((K + HBN_ - 1) / HBN_, (unsigned)((M + HBM_ - 1) / HBM_), 1), dim3(256), lds,
                 (hipStream_t)stream, p, la);
    return check_launch("linear_lngrad (bf16)");
}

extern "C" int dldkd_colsum_f32(const float* x, float* out, long M, long N, void* stream);

extern "C" int dldkd_linear_lngrad(int precision, const float* dy, const float* W, const float* x, const unsigned char* keep,
                                   float keep_scale, const float* mean, const float* rstd, float* workspace, size_t workspace_bytes,
                                   float* dgamma, float* dbeta, long M, int N, int K, const unsigned char* row_flags, void* stream) {
    if (M < 0 || N < 1 || K < 1 || M > 0x7fffffffL) { set_error("linear_lngrad: bad sizes"); return DLDKD_EINVAL; }
    if (precision != DLDKD_GEMM_BF16 && precision != DLDKD_GEMM_F32X3) {
        set_error("linear_lngrad: precision %d has no such kernel (DLDKD_GEMM_F32X3 or DLDKD_GEMM_BF16)", precision);
        return DLDKD_EINVAL;
    }
    if (M == 0) return DLDKD_OK;                        // dgamma / dbeta are zero-initialised by the caller
    if (!dy || !W || !x || !mean || !rstd || !workspace || !dgamma || !dbeta) { set_error("linear_lngrad: null pointer"); return DLDKD_EINVAL; }
    const long tiles = (M + 127) / 128;
    if (workspace_bytes < (size_t)2 * tiles * K * sizeof(float)) { set_error("linear_lngrad: workspace too small"); return DLDKD_EINVAL; }
    // dgamma | dbeta contiguous (the caller's (2, K) buffer): the partial rows are laid out [tile][dgamma K | dbeta K] and ONE column
    // sum over 2 K columns finishes both; otherwise two planes and two column sums
    const bool joined = dbeta == dgamma + K;
    LnGradArgs la{x, keep, mean, rstd, workspace, joined ? workspace + K : workspace + (size_t)tiles * K, K, keep_scale, joined ? 2 * K : K};
    int rc = precision == DLDKD_GEMM_BF16 ? launch_linear_lngrad_bf16(dy, W, M, N, K, la, stream, row_flags)
                                          : launch_linear_lngrad_x3(dy, W, M, N, K, la, stream, row_flags);
    if (rc != DLDKD_OK) return rc;
    if (joined) return dldkd_colsum_f32(la.part_g, dgamma, tiles, 2L * K, stream);
    rc = dldkd_colsum_f32(la.part_g, dgamma, tiles, K, stream);
    if (rc != DLDKD_OK) return rc;
    return dldkd_colsum_f32(la.part_b, dbeta, tiles, K, stream);
}

extern "C" int dldkd_gemm_bf16(const float* A, const float* B, const float* bias, float* C, int M, int N, int K, int lda,
                               int ldb, int ldc, int a_kmajor, int b_kmajor, int relu, void* workspace, size_t workspace_bytes,
                               void* stream) {
    if (M < 0 || N < 0 || K < 0 || lda < 1 || ldb < 1 || ldc < N) { set_error("gemm_bf16: bad sizes"); return DLDKD_EINVAL; }
    if (M == 0 || N == 0) return DLDKD_OK;
    if (!A || !B || !C) { set_error("gemm_bf16: null pointer"); return DLDKD_EINVAL; }
    const int a_vec = !(lda & 3) && !((uintptr_t)A & 15), b_vec = !(ldb & 3) && !((uintptr_t)B & 15);
    GemmHArgs p{A, B, bias, C, M, N, K, lda, ldb, ldc, relu, a_vec, b_vec, 1, 0, 0, 0, 0, 0, 0, 1.0f, 1, 0};
    int per = 0;
    const int split = (!bias && !relu && ldc == N && !((uintptr_t)C & 15)) ? gemm_bf16_split_plan(M, N, K, a_kmajor, b_kmajor, &per) : 1;
    if (split > 1 && workspace && !((uintptr_t)workspace & 15) && workspace_bytes >= (size_t)split * M * N * sizeof(float)) {
        // partial planes [split][M][N] in the caller's workspace (plain stores), then one reduce pass into C
        p.k_tiles_per_split = per;
        p.split_k = split;
        p.C = (float*)workspace;
        const int rc = launch_gemm_h(p, p.split_k, a_kmajor, b_kmajor, stream);
        if (rc != DLDKD_OK) return rc;
        return launch_splitk_reduce((const float*)workspace, C, p.split_k, (long)M * N, (hipStream_t)stream);
    }
    return launch_gemm_h(p, 1, a_kmajor, b_kmajor, stream);
}

// The two GEMMs of the training input projection whose activation operand is stored as bf16 (dldkd_layernorm_dropout_bf16):
//   dw == 0  forward:  C[M, N] = act(A16[M, K] . B[N, K]^T + bias)    A16 bf16 row-major (lda elements), B fp32 (N, K)
//   dw != 0  dW:       C[M, N] = sum_k A[k, m] B16[k, n]              A fp32 (K, M) = dy, B16 bf16 (K, N) = the saved rows;
//                      k_flags (one byte per 32 consecutive k, or NULL): 0 = the tile's rows are padding (zero rows) - skipped;
//   dw == 2  the same with B fp32 (K, N): dldkd_gemm_bf16's dW layout plus the k-tile filter
//   dw == 3  the same with A bf16 (K, M) too (even lda and M): the fused training towers' weight gradients
//                                                                     split-K as dldkd_gemm_bf16 (workspace from
//                                                                     dldkd_gemm_workspace_bytes(DLDKD_GEMM_BF16, M, N, K, 1, 1))
static int gemm_bf16_mixed_impl(int dw, const void* A, const void* B, const float* bias, float* C, int M, int N, int K, int lda,
                                int ldb, int ldc, int relu, void* workspace, size_t workspace_bytes, const unsigned char* k_flags,
                                float* a_colsum, void* stream);

extern "C" int dldkd_cast_bf16(const float* x, void* y, long n, void* stream);
// planes of the register-staged kernel or of gemm_bf16_tn (whichever is larger), then room for the bf16 copy of ONE fp32 A block
static size_t tower_dw_planes_bytes(int n_blocks, long rows) {
    int per = 0;
    const int split = gemm_bf16_split_plan(kHidden * n_blocks, kHidden, (int)rows, 1, 1, &per);
    size_t b = split > 1 ? (size_t)split * kHidden * n_blocks * kHidden * sizeof(float) : 0;
    if (gemm_bf16_tn_enabled() && gemm_bf16_tn_ok(kHidden * n_blocks, kHidden, rows, kHidden, kHidden)) {
        const size_t t = gemm_bf16_tn_planes_bytes(kHidden * n_blocks, kHidden, rows, 0);
        b = t > b ? t : b;
    }
    return (b + 255) & ~(size_t)255;
}
extern "C" size_t dldkd_tower_train_dw_workspace_bytes(int n_blocks, long rows) {
    if (n_blocks < 1 || rows < 1) return 0;
    return tower_dw_planes_bytes(n_blocks, rows) + (size_t)rows * kHidden * sizeof(unsigned short);
}

// What is left of a tower's parameter gradients once the grouped weight-gradient GEMM has run, as ONE launch of three kinds of workgroups:
//   [0, red_blocks)                 the split-K reduce of the GEMM's planes;
//   the next col_blocks * row_blocks  the position table's gradient: dpos[c] += sum over the n_seq sequences of dx1[n, c], c < cols = L * 384
//                                     (colsum_kernel's form; one atomic per column and workgroup);
//   the next n_ln * ln_blocks        the two LayerNorms' parameter gradients: dgamma[c] += sum_r a[r, c] xh[r, c], dbeta[c] += sum_r a[r, c]
//                                     over kLnRows rows per workgroup (a: the gradient of that LayerNorm's output as the row kernels left
//                                     it - bf16, or the fp32 rows the loss handed in; xh: the saved normalised rows, bf16), 32-row groups
//                                     of padding skipped by the row flags.  Thread = 8 columns of every fifth row: 16-byte loads, a row's
//                                     48 threads cover its 768 bytes.
#ifndef DLDKD_LN_ROWS
#define DLDKD_LN_ROWS 64
#endif
constexpr int kLnRows = DLDKD_LN_ROWS;      // rows per workgroup of the LayerNorm sums (a multiple of 32)
struct LnJob {
    const void* a;
    const unsigned short* xh;
    float* dgamma;
    float* dbeta;
    int a16;                     // a is bf16 (else fp32)
    int clear_bit0;              // xh carries a flag in bit 0 of every element (tower_train.hip f1_kernel): cleared before use
};
struct DwFinishArgs {
    const float* ws; float* out; int split; long n4; int red_blocks;
    const float* x; float* csum; long n_seq, cols; int rows_per_block, col_blocks, pos_blocks;
    LnJob ln[2]; int n_ln, ln_blocks; long rows; const unsigned char* rflags;
};
__global__ __launch_bounds__(256) void dw_finish_kernel(const DwFinishArgs a) {
    if ((int)blockIdx.x < a.red_blocks) {
        const long i = (long)blockIdx.x * 256 + threadIdx.x;
        if (i >= a.n4) return;
        const f32x4* w = reinterpret_cast<const f32x4*>(a.ws) + i;
        f32x4 acc = w[0];
#pragma unroll 8
        for (int z = 1; z < a.split; ++z) { const f32x4 b = w[(size_t)z * a.n4]; acc += b; }
        reinterpret_cast<f32x4*>(a.out)[i] = acc;
        return;
    }
    int b = (int)blockIdx.x - a.red_blocks;
    if (b < a.pos_blocks) {
        const int bx = b % a.col_blocks, by = b / a.col_blocks;
        const long c = (long)bx * 256 + threadIdx.x;
        if (c >= a.cols) return;
        const long r0 = (long)by * a.rows_per_block, r1 = r0 + a.rows_per_block < a.n_seq ? r0 + a.rows_per_block : a.n_seq;
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
        long r = r0;
        for (; r + 3 < r1; r += 4) {            // 4 independent loads in flight per lane
            s0 += a.x[r * a.cols + c];
            s1 += a.x[(r + 1) * a.cols + c];
            s2 += a.x[(r + 2) * a.cols + c];
            s3 += a.x[(r + 3) * a.cols + c];
        }
        for (; r < r1; ++r) s0 += a.x[r * a.cols + c];
        atomicAdd(a.csum + c, (s0 + s1) + (s2 + s3));
        return;
    }
    b -= a.pos_blocks;
    const LnJob j = a.ln[b / a.ln_blocks];
    const long row0 = (long)(b % a.ln_blocks) * kLnRows;
    __shared__ float part[5][48][17];
    const int tid = threadIdx.x, rsub = tid / 48, c8 = tid % 48;      // thread = 8 columns of every fifth row (240 of the 256 threads)
    float sg[8], sb[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { sg[e] = 0.f; sb[e] = 0.f; }
    if (tid < 240) {
        const unsigned xm = j.clear_bit0 ? 0xfffefffeu : 0xffffffffu;
#pragma unroll 1
        for (int g = 0; g < kLnRows / 32; ++g) {                          // the 32-row groups of this workgroup's rows
            const long rg = row0 + 32 * g;
            if (rg >= a.rows) break;
            if (a.rflags != nullptr && a.rflags[rg >> 5] == 0) continue;
            const long rend = rg + 32 < a.rows ? rg + 32 : a.rows;
#pragma unroll 7
            for (long r = rg + rsub; r < rend; r += 5) {
                const uint4 xw = *reinterpret_cast<const uint4*>(j.xh + r * kHidden + 8 * c8);
                const unsigned xu[4] = {xw.x & xm, xw.y & xm, xw.z & xm, xw.w & xm};
                float v[8];
                if (j.a16) {
                    const uint4 aw = *reinterpret_cast<const uint4*>((const unsigned short*)j.a + r * kHidden + 8 * c8);
                    const unsigned au[4] = {aw.x, aw.y, aw.z, aw.w};
#pragma unroll
                    for (int d = 0; d < 4; ++d) { v[2 * d] = __builtin_bit_cast(float, au[d] << 16); v[2 * d + 1] = __builtin_bit_cast(float, au[d] & 0xffff0000u); }
                } else {
                    const f32x4 lo = *reinterpret_cast<const f32x4*>((const float*)j.a + r * kHidden + 8 * c8);
                    const f32x4 hi = *reinterpret_cast<const f32x4*>((const float*)j.a + r * kHidden + 8 * c8 + 4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) { v[e] = lo[e]; v[4 + e] = hi[e]; }
                }
#pragma unroll
                for (int d = 0; d < 4; ++d) {
                    sg[2 * d] += v[2 * d] * __builtin_bit_cast(float, xu[d] << 16);
                    sg[2 * d + 1] += v[2 * d + 1] * __builtin_bit_cast(float, xu[d] & 0xffff0000u);
                    sb[2 * d] += v[2 * d]; sb[2 * d + 1] += v[2 * d + 1];
                }
            }
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) { part[rsub][c8][e] = sg[e]; part[rsub][c8][8 + e] = sb[e]; }
    }
    __syncthreads();
    // one atomic per feature and workgroup, consecutive lanes on consecutive addresses: an atomic instruction is served per 128-byte
    // line it touches (the first version - a thread adding its own 8 sums, 16 bytes apart - touched 8 lines per instruction and made
    // this launch 40 us longer at the TVR batch)
    for (int i = tid; i < 2 * kHidden; i += 256) {
        const int which = i >= kHidden ? 1 : 0, col = i - which * kHidden;
        float v = 0.f;
#pragma unroll
        for (int q = 0; q < 5; ++q) v += part[q][col >> 3][8 * which + (col & 7)];
        atomicAdd((which ? j.dbeta : j.dgamma) + col, v);
    }
}

extern "C" int dldkd_colsum_f32(const float* x, float* out, long M, long N, void* stream);

static int tower_train_dw_impl(const void* const* host_A, const int* host_lda, const int* host_acol, const int* host_a16,
                               const void* const* host_B, int n_blocks, long rows, float* dW, float* dbias, void* workspace,
                               size_t workspace_bytes, const unsigned char* k_flags, const float* dx1, float* dpos, long n_seq, long cols,
                               const LnJob* ln, int n_ln, void* stream) {
    if (n_blocks < 1 || n_blocks > 5 || rows < 0 || rows > 0x7fffffffL || !host_A || !host_lda || !host_acol || !host_a16 || !host_B || !dW) {
        set_error("tower_train_dw: bad arguments");
        return DLDKD_EINVAL;
    }
    if (rows == 0) return DLDKD_OK;
    DwGroupArgs g{};
    const int M = kHidden * n_blocks;
    for (int b = 0; b < n_blocks; ++b) {
        if (!host_A[b] || !host_B[b] || host_lda[b] < kHidden || (host_lda[b] & 1) || (host_acol[b] & 1) || host_acol[b] + kHidden > host_lda[b] ||
            ((uintptr_t)host_A[b] & 3) || ((uintptr_t)host_B[b] & 3)) {
            set_error("tower_train_dw: block %d: operands need even strides / column offsets, 4-byte alignment, a 384-column range", b);
            return DLDKD_EINVAL;
        }
        g.A[b] = host_A[b]; g.B[b] = host_B[b]; g.lda[b] = host_lda[b]; g.acol[b] = host_acol[b]; g.a16[b] = host_a16[b];
    }
    g.base = GemmHArgs{nullptr, nullptr, nullptr, dW, M, kHidden, (int)rows, 0, kHidden, kHidden, 0, 0, 0, 1, 0, 0, 0, 0, 0, 0, 1.0f, 1, 0};
    g.base.kflags = k_flags;
    g.base.a_colsum = dbias;
    int per = 0;
    int split = ((uintptr_t)dW & 15) ? 1 : gemm_bf16_split_plan(M, kHidden, (int)rows, 1, 1, &per);
    bool use_split = split > 1 && workspace && !((uintptr_t)workspace & 15) && workspace_bytes >= (size_t)split * M * kHidden * sizeof(float);
    int rc = DLDKD_OK;
    // gemm_bf16_tn.hip (LDS-DMA tiles, transposed LDS reads) when every operand is 16-byte aligned bf16 - at most one fp32 A block (the
    // loss's gradient under the out mapping) is cast into the workspace first
    bool tn = gemm_bf16_tn_enabled() && gemm_bf16_tn_ok(M, kHidden, rows, kHidden, kHidden) && workspace && !((uintptr_t)workspace & 15) &&
              !((uintptr_t)dW & 15) && workspace_bytes >= dldkd_tower_train_dw_workspace_bytes(n_blocks, rows);
    int n32 = 0;
    for (int b = 0; b < n_blocks && tn; ++b) {
        n32 += host_a16[b] ? 0 : 1;
        tn = !(host_lda[b] & 7) && !(host_acol[b] & 7) && !((uintptr_t)host_B[b] & 15) && !((uintptr_t)host_A[b] & 15) && n32 <= 1;
    }
    if (tn) {
        const void* A16[5];
        char* cast = (char*)workspace + tower_dw_planes_bytes(n_blocks, rows);
        for (int b = 0; b < n_blocks; ++b) {
            A16[b] = host_A[b];
            if (!host_a16[b]) {
                if (host_lda[b] != kHidden || host_acol[b] != 0) { tn = false; break; }
                rc = dldkd_cast_bf16((const float*)host_A[b], cast, rows * kHidden, stream);
                if (rc != DLDKD_OK) return rc;
                A16[b] = cast;
            }
        }
        if (tn) {
            split = launch_gemm_bf16_tn_group(A16, host_lda, host_acol, host_B, n_blocks, rows, dW, workspace, k_flags, dbias, (hipStream_t)stream);
            if (split < 0) return split;
            use_split = split > 1;
        }
    }
    if (!tn) {
        if (use_split) { g.base.split_k = split; g.base.k_tiles_per_split = per; g.base.C = (float*)workspace; }
        constexpr size_t lds = sizeof(unsigned short) * 2 * 2 * HBM_ * HPITCH;
        static const bool attr_ok = hipFuncSetAttribute((const void*)gemm_bf16_dw_group_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds) == hipSuccess;
        (void)attr_ok;
        DLDKD_LAUNCH(gemm_bf16_dw_group_kernel, dim3(kHidden / HBN_, M / HBM_, use_split ? split : 1), dim3(256), lds, (hipStream_t)stream, g);
        rc = check_launch("tower_train_dw");
        if (rc != DLDKD_OK) return rc;
    }
    const bool pos = dx1 != nullptr && dpos != nullptr && n_seq > 0 && cols > 0;
    DwFinishArgs f{};
    if (use_split) {
        f.ws = (const float*)workspace; f.out = dW; f.split = split; f.n4 = (long)M * kHidden / 4;
        f.red_blocks = (int)((f.n4 + 255) / 256);
    }
    long blocks = f.red_blocks;
    if (pos) {
        const int rpb = n_seq <= 256 ? 8 : 32;
        const long col_blocks = (cols + 255) / 256, row_blocks = (n_seq + rpb - 1) / rpb;
        if (col_blocks * row_blocks > 0x3fffffffL) { set_error("tower_train_dw: position table too large"); return DLDKD_EINVAL; }
        f.x = dx1; f.csum = dpos; f.n_seq = n_seq; f.cols = cols; f.rows_per_block = rpb; f.col_blocks = (int)col_blocks;
        f.pos_blocks = (int)(col_blocks * row_blocks);
        blocks += f.pos_blocks;
    }
    if (n_ln > 0) {
        f.n_ln = n_ln; f.ln_blocks = (int)((rows + kLnRows - 1) / kLnRows); f.rows = rows; f.rflags = k_flags;
        for (int i = 0; i < n_ln; ++i) f.ln[i] = ln[i];
        blocks += (long)n_ln * f.ln_blocks;
    }
    if (blocks == 0) return DLDKD_OK;
    if (blocks > 0x7fffffffL) { set_error("tower_train_dw: too many workgroups"); return DLDKD_EINVAL; }
    DLDKD_LAUNCH(dw_finish_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, f);
    return check_launch("tower_train_dw (reduce + position sums + LayerNorm sums)");
}

extern "C" int dldkd_tower_train_dw(const void* const* host_A, const int* host_lda, const int* host_acol, const int* host_a16,
                                    const void* const* host_B, int n_blocks, long rows, float* dW, float* dbias, void* workspace,
                                    size_t workspace_bytes, const unsigned char* k_flags, void* stream) {
    return tower_train_dw_impl(host_A, host_lda, host_acol, host_a16, host_B, n_blocks, rows, dW, dbias, workspace, workspace_bytes, k_flags,
                               nullptr, nullptr, 0, 0, nullptr, 0, stream);
}

extern "C" int dldkd_tower_train_dw_pos(const void* const* host_A, const int* host_lda, const int* host_acol, const int* host_a16,
                                        const void* const* host_B, int n_blocks, long rows, float* dW, float* dbias, void* workspace,
                                        size_t workspace_bytes, const unsigned char* k_flags, const float* dx1, float* dpos, long n_seq,
                                        long cols, void* stream) {
    if (!dx1 || !dpos || n_seq < 0 || cols < 1) { set_error("tower_train_dw_pos: bad position-gradient arguments"); return DLDKD_EINVAL; }
    return tower_train_dw_impl(host_A, host_lda, host_acol, host_a16, host_B, n_blocks, rows, dW, dbias, workspace, workspace_bytes, k_flags,
                               dx1, dpos, n_seq, cols, nullptr, 0, stream);
}

extern "C" int dldkd_tower_train_dw_ln(const void* const* host_A, const int* host_lda, const int* host_acol, const int* host_a16,
                                       const void* const* host_B, int n_blocks, long rows, float* dW, float* dbias, void* workspace,
                                       size_t workspace_bytes, const unsigned char* k_flags, const float* dx1, float* dpos, long n_seq,
                                       long cols, const void* dz1_bf16, const void* xh1, const void* dh2, int dh2_is_bf16, const void* xh2,
                                       float* ln_grads, void* stream) {
    if ((dx1 != nullptr) != (dpos != nullptr) || (dx1 && (n_seq < 0 || cols < 1))) { set_error("tower_train_dw_ln: bad position-gradient arguments"); return DLDKD_EINVAL; }
    if (!dz1_bf16 || !xh1 || !dh2 || !xh2 || !ln_grads || (((uintptr_t)dz1_bf16 | (uintptr_t)xh1 | (uintptr_t)xh2) & 7) ||
        ((uintptr_t)dh2 & (dh2_is_bf16 ? 7 : 15)) || ((uintptr_t)ln_grads & 3)) {
        set_error("tower_train_dw_ln: null or unaligned LayerNorm operands");
        return DLDKD_EINVAL;
    }
    // ln_grads = [dgamma2 | dbeta2 | dgamma1 | dbeta1] (the order of the row kernels' accumulators)
    const LnJob ln[2] = {{dh2, (const unsigned short*)xh2, ln_grads, ln_grads + kHidden, dh2_is_bf16 ? 1 : 0, 0},
                         {dz1_bf16, (const unsigned short*)xh1, ln_grads + 2 * kHidden, ln_grads + 3 * kHidden, 1, 0}};   // (xh1 carries no flag bit since round 6)
    return tower_train_dw_impl(host_A, host_lda, host_acol, host_a16, host_B, n_blocks, rows, dW, dbias, workspace, workspace_bytes, k_flags,
                               dx1, dpos, n_seq, cols, ln, 2, stream);
}


static size_t inproj_planes_bytes(int N, int K, long M) {
    int per = 0;
    const int split = gemm_bf16_split_plan(N, K, (int)M, 1, 1, &per);
    size_t b = (size_t)(split > 1 ? split : 1) * 2 * N * K * sizeof(float);
    if (gemm_bf16_tn_enabled() && gemm_bf16_tn_ok(N, K, M, N, K)) {
        const size_t t = gemm_bf16_tn_planes_bytes(N, K, M, 1);
        b = t > b ? t : b;
    }
    return (b + 255) & ~(size_t)255;
}
// the [split][dW | H] planes, then room for the bf16 copy of dY
extern "C" size_t dldkd_inproj_bwd_workspace_bytes(int N, int K, long M) {
    if (N < 1 || K < 1 || M < 1) return 0;
    return inproj_planes_bytes(N, K, M) + (size_t)M * N * sizeof(unsigned short);
}

extern "C" int dldkd_inproj_bwd_bf16(const float* dy, const void* z_bf16, const float* W, const float* gamma, const float* beta,
                                     float keep_scale, const float* x, const unsigned char* keep, float p_drop, unsigned long long seed,
                                     unsigned long long offset, const unsigned long long* state, const float* mean, const float* rstd,
                                     float* dW, float* dbias, float* dgamma, float* dbeta, long M, int N, int K, void* workspace,
                                     size_t workspace_bytes, const unsigned char* k_flags, const void* dy_bf16, void* stream) {
    if (M < 0 || M > 0x7fffffffL || N < 1 || N > kHidden || K < 2 || (K & 3) || (N & 1)) {
        set_error("inproj_bwd: bad sizes (M=%ld N=%d K=%d; N <= 384, K a multiple of 4)", M, N, K);
        return DLDKD_EINVAL;
    }
    if (M == 0) return DLDKD_OK;
    if (!dy || !z_bf16 || !W || !gamma || !beta || !x || !mean || !rstd || !dW || !dgamma || !dbeta || !workspace) {
        set_error("inproj_bwd: null pointer");
        return DLDKD_EINVAL;
    }
    if (((uintptr_t)z_bf16 & 3) || ((uintptr_t)workspace & 15) || ((uintptr_t)dW & 15) || ((uintptr_t)W & 15) || workspace_bytes < dldkd_inproj_bwd_workspace_bytes(N, K, M)) {
        set_error("inproj_bwd: unaligned buffer or workspace too small");
        return DLDKD_EINVAL;
    }
    // C[n, k] over the batch rows: A = dY (rows, N) fp32 k-major, B = z (rows, K) bf16 k-major; planes [split][dW | H] in the workspace
    GemmHArgs p{dy, (const float*)z_bf16, nullptr, (float*)workspace, N, K, (int)M, N, K, K, 0, 0, 0, 1, 0, 0, 0, 0, 0, 0, 1.0f, 1, 0};
    p.kflags = k_flags;
    p.a_colsum = dbias;
    int per = 0;
    int split = gemm_bf16_split_plan(N, K, (int)M, 1, 1, &per);
    if (split > 1) { p.split_k = split; p.k_tiles_per_split = per; } else split = 1;
    hipStream_t s = (hipStream_t)stream;
    int rc = DLDKD_OK;
    if (gemm_bf16_tn_enabled() && gemm_bf16_tn_ok(N, K, M, N, K) && !((uintptr_t)z_bf16 & 15) && !((uintptr_t)dy & 15)) {
        // gemm_bf16_tn.hip: dY cast to bf16 once (the register-staged kernel rounds the same values on their way to LDS)
        const void* dy16 = dy_bf16;
        if (dy16 == nullptr || ((uintptr_t)dy16 & 15)) {
            dy16 = (char*)workspace + inproj_planes_bytes(N, K, M);
            rc = dldkd_cast_bf16(dy, (void*)dy16, M * N, stream);
            if (rc != DLDKD_OK) return rc;
        }
        split = launch_gemm_bf16_tn(dy16, z_bf16, nullptr, N, K, M, N, K, 1, workspace, k_flags, dbias, s);
        if (split < 0) return split;
    } else {
        constexpr size_t lds = sizeof(unsigned short) * 3 * 2 * HBM_ * HPITCH;
        static const bool attr_ok = hipFuncSetAttribute((const void*)gemm_bf16_dw_dual_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds) == hipSuccess;
        (void)attr_ok;
        DLDKD_LAUNCH(gemm_bf16_dw_dual_kernel, dim3((K + HBN_ - 1) / HBN_, (N + HBM_ - 1) / HBM_, split), dim3(256), lds, s, p);
        rc = check_launch("inproj_bwd (dual dW)");
        if (rc != DLDKD_OK) return rc;
    }
    const InprojFinishArgs f{(const float*)workspace, W, dW, dgamma, dbeta, split, N, K};
    DLDKD_LAUNCH(inproj_bwd_reduce_kernel, dim3((K + 255) / 256, (N + 15) / 16), dim3(256), 0, s, f);
    rc = check_launch("inproj_bwd (reduce)");
    if (rc != DLDKD_OK) return rc;
    if (!(p_drop >= 0.f && p_drop < 1.f)) { set_error("inproj_bwd: p_drop must be in [0, 1)"); return DLDKD_EINVAL; }
    const double tq = (double)p_drop * 4294967296.0;                              // the threshold dldkd_layernorm_dropout_bf16 used
    const unsigned thresh = tq >= 4294967295.0 ? 4294967295u : (unsigned)tq;
    const InprojFinalArgs g{dgamma, dbeta, gamma, beta, keep_scale, K, dy, W, x, keep, mean, rstd, M, N, keep ? 0u : thresh, seed, offset, state};
    DLDKD_LAUNCH(inproj_bwd_final_kernel, dim3((K + 255) / 256), dim3(256), 0, s, g);
    return check_launch("inproj_bwd (final)");
}

extern "C" int dldkd_gemm_bf16_mixed(int dw, const void* A, const void* B, const float* bias, float* C, int M, int N, int K, int lda,
                                     int ldb, int ldc, int relu, void* workspace, size_t workspace_bytes, const unsigned char* k_flags,
                                     void* stream) {
    return gemm_bf16_mixed_impl(dw, A, B, bias, C, M, N, K, lda, ldb, ldc, relu, workspace, workspace_bytes, k_flags, nullptr, stream);
}

// The dW layouts (dw = 1: A fp32, dw = 3: A bf16; B bf16) with the bias gradient on the side: a_colsum[m] += sum_k A[k, m] over the
// k-tiles that are not skipped (zeroed by the caller; fp32 atomics).
extern "C" int dldkd_gemm_bf16_dw_bias(int dw, const void* A, const void* B, float* C, int M, int N, int K, int lda, int ldb,
                                       void* workspace, size_t workspace_bytes, const unsigned char* k_flags, float* a_colsum,
                                       void* stream) {
    if (dw != 1 && dw != 3) { set_error("gemm_bf16_dw_bias: dw must be 1 or 3"); return DLDKD_EINVAL; }
    return gemm_bf16_mixed_impl(dw, A, B, nullptr, C, M, N, K, lda, ldb, N, 0, workspace, workspace_bytes, k_flags, a_colsum, stream);
}

static int gemm_bf16_mixed_impl(int dw, const void* A, const void* B, const float* bias, float* C, int M, int N, int K, int lda,
                                int ldb, int ldc, int relu, void* workspace, size_t workspace_bytes, const unsigned char* k_flags,
                                float* a_colsum, void* stream) {
    if (M < 0 || N < 0 || K < 0 || lda < 1 || ldb < 1 || ldc < N) { set_error("gemm_bf16_mixed: bad sizes"); return DLDKD_EINVAL; }
    if (M == 0 || N == 0) return DLDKD_OK;
    if (!A || !B || !C) { set_error("gemm_bf16_mixed: null pointer"); return DLDKD_EINVAL; }
    if (!dw) {
        if ((lda & 3) || ((uintptr_t)A & 7)) { set_error("gemm_bf16_mixed: bf16 A needs lda %% 4 == 0 and 8-byte alignment"); return DLDKD_EINVAL; }
        const int b_vec = !(ldb & 3) && !((uintptr_t)B & 15);
        GemmHArgs p{(const float*)A, (const float*)B, bias, C, M, N, K, lda, ldb, ldc, relu, 1, b_vec, 1, 0, 0, 0, 0, 0, 0, 1.0f, 1, 0};
        p.mflags = (M % HBM_ == 0 && !((uintptr_t)k_flags & 3)) ? k_flags : nullptr;     // forward: the flags are per 32 ROWS of A / C
        return launch_gemm_h_mixed(p, 1, 0, stream);
    }
    if (bias || relu) { set_error("gemm_bf16_mixed: the dW layout takes no bias / ReLU"); return DLDKD_EINVAL; }
    if (dw == 2) {          // both operands fp32 and k-major (the plain dW of dldkd_gemm_bf16) with the k-tile filter
        GemmHArgs p{(const float*)A, (const float*)B, nullptr, C, M, N, K, lda, ldb, ldc, 0, 0, 0, 1, 0, 0, 0, 0, 0, 0, 1.0f, 1, 0};
        p.kflags = k_flags;
        int per = 0;
        const int split = (ldc == N && !((uintptr_t)C & 15)) ? gemm_bf16_split_plan(M, N, K, 1, 1, &per) : 1;
        if (split > 1 && workspace && !((uintptr_t)workspace & 15) && workspace_bytes >= (size_t)split * M * N * sizeof(float)) {
            p.k_tiles_per_split = per;
            p.split_k = split;
            p.C = (float*)workspace;
            const int rc = launch_gemm_h(p, p.split_k, 1, 1, stream);
            if (rc != DLDKD_OK) return rc;
            return launch_splitk_reduce((const float*)workspace, C, p.split_k, (long)M * N, (hipStream_t)stream);
        }
        return launch_gemm_h(p, 1, 1, 1, stream);
    }
    if ((ldb & 1) || (N & 1) || N < 2 || ((uintptr_t)B & 3)) { set_error("gemm_bf16_mixed: bf16 k-major B needs even ldb and N"); return DLDKD_EINVAL; }
    if (dw == 3 && ((lda & 1) || (M & 1) || M < 2 || ((uintptr_t)A & 3))) { set_error("gemm_bf16_mixed: bf16 k-major A needs even lda and M"); return DLDKD_EINVAL; }
    static_assert(HBK_ == 32, "k_flags are per 32 contraction rows");
    GemmHArgs p{(const float*)A, (const float*)B, nullptr, C, M, N, K, lda, ldb, ldc, 0, 0, 0, 1, 0, 0, 0, 0, 0, 0, 1.0f, 1, 0};
    p.kflags = k_flags;            // (K + 31) / 32 bytes or NULL
    p.a_colsum = a_colsum;
    int per = 0;
    const int split = (ldc == N && !((uintptr_t)C & 15)) ? gemm_bf16_split_plan(M, N, K, 1, 1, &per) : 1;
    if (split > 1 && workspace && !((uintptr_t)workspace & 15) && workspace_bytes >= (size_t)split * M * N * sizeof(float)) {
        p.k_tiles_per_split = per;
        p.split_k = split;
        p.C = (float*)workspace;
        const int rc = launch_gemm_h_mixed(p, p.split_k, dw == 3 ? 3 : 1, stream);
        if (rc != DLDKD_OK) return rc;
        return launch_splitk_reduce((const float*)workspace, C, p.split_k, (long)M * N, (hipStream_t)stream);
    }
    return launch_gemm_h_mixed(p, 1, dw == 3 ? 3 : 1, stream);
}
