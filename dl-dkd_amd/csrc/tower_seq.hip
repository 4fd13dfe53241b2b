// NOTE (round 5): the 16-bit MFMA operands of this file are h16 = IEEE fp16, not bf16 (common.hpp says why; the text below and the
// identifiers still say "bf16" where they mean "the 16-bit operand": bf16x8 is the 8 x 16-bit container, whatever the format).
// K5: the whole clip / word tower AFTER the input projection as ONE kernel, one workgroup per (sequence, branch):
//     h1 = LN(h0 + pos)                                   TrainablePositionalEncoding.forward, model_components.py:277-284
//     q|k|v = h1 W^T + b ; P = softmax(q k^T / sqrt(96) + key mask) ; ctx = P v      BertSelfAttention.forward, :398-436
//     h2 = LN(ctx Wd^T + bd + h1)                          BertSelfOutput.forward, :446-450 (BertAttention.forward :345-353)
//     y  = h2 Wo^T + bo                                    out_mapping_linear, model.py:219 (video towers only)
//     gallery row = bf16(y / max(|y|, 1e-12))              F.normalize of get_sim_scores, model.py:319, in the scorer's packed layout
// (throughput mode: bf16 MFMA operands, fp32 accumulation / statistics; the parity towers keep their fp32-grade kernels).
// Before this kernel the same work was 7+ launches per branch with fp32 activations round-tripping HBM between them
// (profiles/r02/enc_fast_kernel_stats.csv: 1.64 ms per 1024 videos for ~0.2 ms of arithmetic).
//
// MI355X design
//   * TRANSPOSED ACTIVATIONS IN REGISTERS.  Wave w owns rows 32 w .. 32 w + 31 of the sequence for the whole kernel.  Every
//     linear layer is computed as Y^T = W X^T with mfma_f32_32x32x16_bf16: the weights are the A operand, the activations the B
//     operand, so an accumulator tile holds 32 OUTPUT FEATURES on its 16 registers x 2 lane halves and the wave's 32 sequence
//     rows on its lanes.  A following layer contracts over exactly that register index, so the accumulators - rounded to bf16
//     in place - ARE its B operand (cdna_hip_programming.md section 3, "An accumulator tile as the next MFMA's operand"): no LDS
//     round trip, no lane movement between layers.  The k order inside such a fragment is permuted (element j of lane half h of
//     k-step ks is feature 16 ks + 8 (j >> 2) + 4 h + (j & 3)); the weights are packed ONCE in that order (tower_pack_kernel).
//     LayerNorm / L2-norm statistics run over the register index: in-register sums + one v_permlane32_swap.
//   * ATTENTION WITHOUT TRANSPOSES.  Q^T and K^T come out transposed ([d][row]); V is computed in the normal orientation
//     ([row][d]: the same bf16 registers of h1 serve as the A operand).  Then S^T = K Q^T takes K^T's accumulators as the A
//     operand ("X^T B" form) and Q^T's as B; softmax runs over the accumulator registers (keys), P^T is the B operand of
//     O^T = V^T P^T whose A operand is V's accumulators - every product sums over the previous one's register index.  Only
//     K and V cross waves (a wave needs all 128 keys): 12 KiB per wave and head through LDS, written and read as raw 1-KiB
//     fragments (ds_write_b128 / ds_read_b128 at base + lane * 16: conflict-free).
//   * WEIGHTS STREAM L2 -> LDS -> all four waves: 1.44 MB of fragments per (sequence, branch) in consumption order, by LDS-DMA
//     (global_load_lds_dwordx4, 1 KiB per wave-instruction) into a 96-slot ring, 32-fragment chunks, two chunks in flight
//     behind a counted vmcnt, one raw s_barrier per chunk; fragment reads run 3 deep behind hand-counted lgkmcnt (hipcc turns
//     such a ring into read + lgkmcnt(0) pairs).  Blocks of one branch share an XCD (blockIdx & 4), so an L2 holds one branch's
//     weights (1.47 MB of 4 MiB).
//   * per-feature vectors (biases, gamma, beta) sit in LDS in fragment order (14 KiB); padded keys are masked through the
//     INITIAL ACCUMULATOR of S^T (0 / -inf), padded key tiles are skipped.
//   * A WORKGROUP IS FOUR 32-ROW SLOTS, not one sequence: the host packs the tiles of short sequences side by side (two
//     64-clip videos, four 30-word queries) - a wave only ever talks to the slots of its own sequence (K / V).
#include <type_traits>

#include "common.hpp"

namespace dldkd {
namespace tw {

constexpr int kNKS = 24;                       // k-steps of 16 over 384 features
constexpr int kChunk = 32;                     // ring: fragments per chunk
#ifndef TW_PERS_EXP
#define TW_PERS_EXP 0
#endif
#ifndef TW_FOLD_LN2
#define TW_FOLD_LN2 1
#endif
#ifndef TW_SPREAD
#define TW_SPREAD 1
#endif
#ifndef TW_DEPTH
#define TW_DEPTH 3
#endif
constexpr int kDepth = TW_DEPTH, kFr = kDepth < 4 ? 4 : 8;   // fragment reads in flight; fragment registers (a ring of kFr x 4 VGPRs)
constexpr int kHeadFrags = 3 * 3 * kNKS;       // Q | K | V of one head: 3 tiles x 24 k-steps each
constexpr int kQKVFrags = 4 * kHeadFrags;      // 864
constexpr int kSqFrags = 12 * kNKS;            // a 384 x 384 linear: 288
constexpr int P_G1 = 0, P_B1 = 384, P_BQ = 768, P_BK = 1152, P_BV = 1536, P_BD = 1920, P_G2 = 2304, P_B2 = 2688, P_BO = 3072,
              P_TOTAL = 3456;
constexpr int kLdsKV = 96 * 1024;              // [ring 96 KiB][K/V exchange, later output staging: 50 KiB][params][key mask]
constexpr int kKVBytes = 51200;
constexpr int kLdsPar = kLdsKV + kKVBytes;
constexpr int P_WG = P_BO;                     // query towers (no out mapping): modular weight x gamma2 in the out-mapping bias slot
constexpr int P_TAIL = 4;                      // ... followed by c1 = sum w gamma2, c2 = sum w beta2 (+ 2 unused words)
constexpr int kLdsTotal = kLdsPar + P_TOTAL * 4 + 512;      // 163,840 = all 160 KiB
constexpr int kLdsSink = kLdsPar + P_TOTAL * 4 + 256;       // 256 bytes nobody reads: where the L2 touch loads of the persistent kernel land
constexpr int kStgPitch = 400;                 // output staging: 32 rows x 384 payload bytes per wave and pass
constexpr int kPosTile = kNKS * 64 * 8;        // floats of one 32-position tile of the position table in fragment order
constexpr int kInStage = 32 * 1024;            // prologue: h0 half-rows are staged at [32 KiB + wave * 24 KiB, +24 KiB) of LDS

typedef unsigned u32x4v __attribute__((ext_vector_type(4)));

template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

// LDS-DMA of one 1-KiB piece: global address = scalar base + lane * 16, LDS destination = lds_base (-> M0) + lane * 16
__device__ __forceinline__ void glds_piece(uint32_t voff, const char* sbase, uint32_t lds_base) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(voff), "s"(sbase), "s"(lds_base) : "memory");
}
template <int OFF>
__device__ __forceinline__ void lds_read16(bf16x8& dst, uint32_t addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "i"(OFF) : "memory");
}
__device__ __forceinline__ float half_swap_max(float m) {
    const unsigned u = __builtin_bit_cast(unsigned, m);
    auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    return fmaxf(__builtin_bit_cast(float, (unsigned)r[0]), __builtin_bit_cast(float, (unsigned)r[1]));
}
__device__ __forceinline__ float half_swap_sum(float m) {
    const unsigned u = __builtin_bit_cast(unsigned, m);
    auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    return __builtin_bit_cast(float, (unsigned)r[0]) + __builtin_bit_cast(float, (unsigned)r[1]);
}
__device__ __forceinline__ bf16x8 pack8(const float (&v)[8]) {
    bf16x8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = (short)f32_to_h16_bits(v[j]);          // h16 = fp16 operands (common.hpp)
    return o;
}

struct TowerArgs {
    const float* h0[2];       // per branch: input-projection output, rows (., 384) fp32 ...
    const unsigned short* h0b[2];   // ... or (H16 kernels) the same rows as bf16, written so by K4b (dldkd_in_proj_h16_rows128b_out16)
    const char* blob[2];      // per branch: weight fragments in stream order, the parameter table, the position table in
                              // fragment order
    const int32_t* row0;      // [n_seq] first row of the sequence in h0 / out; null: seq * seq_rows
    const int32_t* lens;      // [n_seq] valid rows (> 0 for every scheduled sequence)
    const int32_t* items;     // [n_items][4]: what the four waves of a workgroup work on: (seq << 10) | (tile << 8) | length, or -1 (idle);
                              // the 32-row tiles of one sequence sit in consecutive slots, in order.  null: workgroup i = sequence i
    int n_items, n_seq, n_branches;
    float* out[2];            // OUTMODE 0: fp32 rows (., 384), indexed like h0
    int seq_rows;             // OUTMODE 0: rows allotted per sequence, 0 = ragged.  Rows len .. seq_rows - 1: computed like the
                              // reference does (no item table) or written as zeros (with an item table)
    char* gal[2];             // OUTMODE 1: gallery blobs bf16 [nv][Lp][384]
    int v0, Lp;               // OUTMODE 1: gallery index of sequence 0, rows per video
    int32_t* lens_out;        // OUTMODE 1: lens of the whole gallery (or null)
    float* pooled[2];         // OUTMODE 2: (n_seq, 384) fp32 modular query vectors
    unsigned long long* stamps;   // diagnostics only (dldkd_debug_tower_seq_timeline): 24 words per workgroup, else null
    int skip_zero_rows;       // OUTMODE 1: rows >= ceil(len / 16) * 16 of a video are NOT written (the scorers never read them: only
                              // ceil(len / 16) row tiles are loaded) - for a gallery buffer whose padding is already zero (zero-filled
                              // once, same lengths every epoch): 4.29 -> 2.56 GB of HBM writes per TVR gallery
    int32_t* nonfinite;       // or null (OUTMODE 0 / 1): set to 1 when the second LayerNorm of a VALID row sees a mean that is not finite -
                              // an fp16 operand overflowed somewhere upstream (h0 from K4 / K4b, q | k | v, the context).  OUTMODE 2
                              // (query towers; that kernel has no register to spare) leaves it alone: a non-finite query vector is
                              // flagged by dldkd_pack_queries_bf16 (bad_flags)
};

// OUTMAP: the stream ends with the 384 x 384 out_mapping_linear (video towers); OUTMODE 0: fp32 rows, 1: packed bf16 gallery,
// 2 (query towers, sequences of at most 32 words, one per wave): the modular attention pooling of get_modularized_queries
// (method/model.py:245-258) on top: softmax_l(mask_logits(w . h2_l)) -> sum_l a_l h2_l, one 384-vector per sequence.
template <bool OUTMAP, int OUTMODE, bool STAMP = false, bool H16 = false>
__global__ __launch_bounds__(256, 1) void tower_seq_kernel(const TowerArgs p) {
    unsigned long long ts[16], tp[8], t_dma = 0, t_bar = 0;
    int n_tp = 0;
    int n_ts = 0;
    auto stamp = [&]() {      // phase boundaries only: no hand-counted read is in flight there (cdna_hip_programming.md section 7)
        if constexpr (STAMP) {
            __builtin_amdgcn_sched_barrier(0);
            unsigned long long t;
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
            __builtin_amdgcn_sched_barrier(0);
            ts[n_ts < 15 ? n_ts : 15] = t;
            ++n_ts;
        }
    };
    auto pstamp = [&]() {     // finer stamps inside the prologue (diagnostic build only)
        if constexpr (STAMP) {
            __builtin_amdgcn_sched_barrier(0);
            unsigned long long t;
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
            __builtin_amdgcn_sched_barrier(0);
            tp[n_tp < 7 ? n_tp : 7] = t;
            ++n_tp;
        }
    };
    stamp();
    constexpr int NFRAG = kQKVFrags + kSqFrags + (OUTMAP ? kSqFrags : 0);
    constexpr int NCH = NFRAG / kChunk;
    static_assert(NFRAG % kChunk == 0, "whole chunks");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // (not const: the persistent form passes them through an empty asm at the top of every item - see there)
    const int tid = threadIdx.x;
    int lane = tid & 63;
    int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int r = lane & 31, h = lane >> 5;
    // PERS (the bf16-h0 gallery kernel): a PERSISTENT workgroup - one per CU - walks items item, item + stride, ...; the next item's
    // slot entry, row0 and h0 rows and the first weight chunks are fetched under the current item's tail, so that an item's
    // prologue no longer waits for three dependent scalar loads and an HBM round trip (25 k of a workgroup's 158 k cycles in the
    // stamped build) and its row stores drain under the next item's products.
    constexpr bool PERS = H16 && !STAMP && !(TW_PERS_EXP & 4);
    int branch = 0, item = blockIdx.x;
    if (p.n_branches == 2) {        // blocks b and b + 8 share an XCD: blocks with (b & 4) equal share a branch's weights in L2
        branch = (blockIdx.x >> 2) & 1;
        item = (blockIdx.x >> 3) * 4 + (blockIdx.x & 3);
    }
    const int first_item = item;
    const int stride = p.n_branches == 2 ? (int)(gridDim.x >> 1) : (int)gridDim.x;   // PERS: items between two of this workgroup's
    if (item >= p.n_items) return;
    uint32_t smem_lds = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(smem));
    uint32_t lane16 = lane * 16;
    const char* wsrc = p.blob[branch];
    float* par = reinterpret_cast<float*>(smem + kLdsPar);
    auto issue_chunk = [&](auto cc) {
        constexpr int c = decltype(cc)::value;
        const char* src = wsrc + (size_t)(c * kChunk + wave * 8) * 1024;
        const uint32_t dst = smem_lds + ((c % 3) * kChunk + wave * 8) * 1024;
#pragma unroll
        for (int i = 0; i < 8; ++i) glds_piece(lane16, src + i * 1024, dst + i * 1024);
    };
    auto issue_piece = [&](auto cc, auto ic) {
        constexpr int c = decltype(cc)::value, i = decltype(ic)::value;
        const char* src = wsrc + (size_t)(c * kChunk + wave * 8 + i) * 1024;
        const uint32_t dst = smem_lds + ((c % 3) * kChunk + wave * 8 + i) * 1024;
        glds_piece(lane16, src, dst);
    };
    // the first weight chunk depends on the branch only: in flight before the (dependent, scalar) loads that say which sequence
    // this wave works on
    issue_chunk(std::integral_constant<int, 0>{});
    if constexpr (PERS) issue_chunk(std::integral_constant<int, 1>{});
    // PERS: what the NEXT iteration works on, fetched one iteration ahead (ent: the slot entry, row0: its sequence's first row)
    int ent_cur = 0, row0_cur = 0, ent_nxt = 0, row0_nxt = 0, ent_nn = 0;     // this item, the next, the one after
    auto slot_entry = [&](int it) {
        int e = p.items[it * 4 + wave];
        if (e < 0) e = p.items[it * 4] | (int)0x80000000;          // idle slot: slot 0's tile, marked (entries are < 2^31)
        return __builtin_amdgcn_readfirstlane(e);
    };
    // (PERS) pull a slot's 32 h0 rows (24 KiB = 192 lines of 128 B: three loads per lane, results discarded) into L2 one item
    // ahead: the prologue's row loads then meet L2, not HBM under load.  Keeping the rows themselves in registers across the
    // item boundary was tried first: the register allocator answered with 96 registers of scratch and a wait behind every load.
    auto touch_rows = [&](int e, int row0_) {
        const int len_ = e & 255, tile_ = (e >> 8) & 3;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int line = lane + 64 * i, row = line / 6, part = line - 6 * row;
            const int l = 32 * tile_ + row, lrow = l < len_ ? l : len_ - 1;
            const char* a = reinterpret_cast<const char*>(p.h0b[branch] + ((size_t)row0_ + lrow) * kHidden) + part * 128;
            // an LDS-DMA into 256 spare bytes behind the parameter table: a load with a REGISTER destination would write it when the
            // data arrives - long after the compiler, which sees an unused result, has given that register to a live value (the
            // first build did exactly that: every item but a workgroup's last came out wrong)
            asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dword %0, off" : : "v"(a), "s"(smem_lds + kLdsSink) : "memory");
        }
    };
    if constexpr (PERS) {
        ent_cur = slot_entry(item);
        row0_cur = __builtin_amdgcn_readfirstlane(p.row0[(ent_cur & 0x7fffffff) >> 10]);
        if (item + stride < p.n_items) ent_nxt = slot_entry(item + stride);
        {   // parameter table: once per workgroup (the branch is fixed)
            const f32x4* psrc = reinterpret_cast<const f32x4*>(wsrc + (size_t)(kQKVFrags + 2 * kSqFrags) * 1024);
            f32x4* dst = reinterpret_cast<f32x4*>(par);
            for (int i = tid; i < (P_TOTAL + P_TAIL) / 4; i += 256) dst[i] = psrc[i];
        }
    }
    for (;;) {          // (not PERS: one pass)
    // The loop makes every address the body forms from these seven values loop-invariant, and LLVM hoists them all in front of it:
    // 360 64-bit weight-chunk pointers, hundreds of per-lane LDS addresses - 765 SGPR and 330 VGPR spills in the first build.
    // Opaque per iteration, they are recomputed where they are used, as in the one-pass kernel.
    // (the wave-uniform ones go through a VGPR and v_readfirstlane: LLVM takes the result of an asm for divergent and then puts
    // "s" operands of later asm statements into VGPRs - `s_mov_b32 m0, v1` - which only the assembler notices)
    if constexpr (PERS) {
        int w_ = wave;
        uint32_t sl_ = smem_lds, lo_ = (uint32_t)reinterpret_cast<uint64_t>(wsrc), hi_ = (uint32_t)(reinterpret_cast<uint64_t>(wsrc) >> 32);
        asm volatile("" : "+v"(lane), "+v"(w_), "+v"(sl_), "+v"(lo_), "+v"(hi_));
        r = lane & 31;                                                 // (re-derived: the compiler may rematerialise them from `lane`)
        h = lane >> 5;
        lane16 = lane * 16;
        wave = __builtin_amdgcn_readfirstlane(w_);
        smem_lds = (uint32_t)__builtin_amdgcn_readfirstlane((int)sl_);
        wsrc = reinterpret_cast<const char*>(((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)hi_) << 32) |
                                             (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)lo_));
    }
    // this wave's slot: a 32-row tile of some sequence.  An idle slot re-computes slot 0's tile and stores nothing (the
    // instruction stream, with its barriers and its share of the weight DMA, is the same for every wave).
    int seq, tile;
    bool live = true;
    int len_item = -1;
    if constexpr (PERS) {                              // fetched one iteration ahead (entry point: an item table and row0 are given)
        live = ent_cur >= 0;
        const int ent = ent_cur & 0x7fffffff;
        seq = ent >> 10;
        tile = (ent >> 8) & 3;
        len_item = ent & 255;
    } else if (p.items != nullptr) {
        int ent = p.items[item * 4 + wave];                      // (seq << 10) | (tile << 8) | length: ONE scalar load per wave
        if (ent < 0) { live = false; ent = p.items[item * 4]; }
        seq = ent >> 10;
        tile = (ent >> 8) & 3;
        len_item = ent & 255;
    } else if constexpr (OUTMODE == 2) {
        seq = item * 4 + wave;                         // four single-tile sequences per workgroup
        tile = 0;
        if (seq >= p.n_seq) { live = false; seq = item * 4; }
    } else {
        // without an item table the workgroup is sequence `item`.  OUTMODE 0 then computes ALL seq_rows rows: a clip past the
        // length is a query like any other (only KEYS are masked, model_components.py:422), so its row comes out as the
        // reference computes it from the zero features - don't-care values, but the same ones.
        seq = item;
        tile = wave;
        const int rows = OUTMODE == 0 && p.seq_rows > p.lens[item] ? p.seq_rows : p.lens[item];
        if (wave > 0 && 32 * wave >= rows) { live = false; tile = 0; }
    }
    const int len_raw = len_item >= 0 ? len_item : p.lens[seq];
    const int len = OUTMODE == 2 ? (len_raw < 1 ? 1 : len_raw > 32 ? 32 : len_raw) : len_raw;   // (query mode: host contract 1..32)
    const int nrows = OUTMODE == 0 && p.items == nullptr && p.seq_rows > len ? p.seq_rows : len;   // rows computed and stored
    const int first = (live || (OUTMODE == 2 && p.items == nullptr)) ? wave - tile : 0;   // slot of the sequence's tile 0 (K / V of key tile kt: slot first + kt)
    const int row0 = PERS ? row0_cur : p.row0 != nullptr ? p.row0[seq] : seq * p.seq_rows;
    if (!PERS && OUTMODE != 2 && len <= 0) {     // only without an item table
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");               // (an LDS-DMA must not outlive its workgroup) (the host never schedules an empty sequence): the workgroup IS the sequence
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
        if constexpr (OUTMODE == 1) {
            char* g = p.gal[branch] + (size_t)(p.v0 + seq) * p.Lp * (kHidden * 2);
            for (int i = tid; i < p.Lp * 48; i += 256) *reinterpret_cast<f32x4*>(g + (size_t)i * 16) = z;
            if (p.lens_out != nullptr && branch == 0 && tid == 0) p.lens_out[p.v0 + seq] = 0;
        } else {
            float* o = p.out[branch] + (size_t)row0 * kHidden;
            for (int i = tid; i < p.seq_rows * 96; i += 256) *reinterpret_cast<f32x4*>(o + (size_t)i * 4) = z;
        }
        return;
    }
    const int ntiles = (len + 31) >> 5;                // 32-row tiles of the sequence (wave-uniform)
    const bool has_next = PERS && item + stride < p.n_items;
    if constexpr (PERS) {
        // two independent fetches, issued here and complete at the prologue's vmcnt(0) - i.e. before the hand-counted stream begins:
        // the next item's row0 (its slot entry came in one iteration ago) and the slot entry of the item after it
        if (has_next) row0_nxt = __builtin_amdgcn_readfirstlane(p.row0[(ent_nxt & 0x7fffffff) >> 10]);
        if (item + 2 * stride < p.n_items) ent_nn = slot_entry(item + 2 * stride);
    }

    // ---- weight ring --------------------------------------------------------------------------------------------
    bf16x8 fr[kFr];
    const uint32_t ring_a = smem_lds + lane16, ring_b = ring_a + 48 * 1024;
    auto ring_read = [&](auto nc) {
        constexpr int n = decltype(nc)::value, slot = n % 96;
        if constexpr (slot < 48) lds_read16<slot * 1024>(fr[n % kFr], ring_a);
        else lds_read16<(slot - 48) * 1024>(fr[n % kFr], ring_b);
    };
    // everything that has to happen before the MFMA that consumes fragment n.  Fragment reads run kDepth ahead INSIDE a phase
    // (one head's Q | K | V projection, the dense layer, the out mapping) and never across a phase end: between phases the
    // compiler schedules its own code (attention, LayerNorm) and must not find asm loads in flight in registers it may move.
    auto pre = [&](auto nc) {
        constexpr int n = decltype(nc)::value;
        constexpr int pbeg = n < kQKVFrags ? n / kHeadFrags * kHeadFrags : n < kQKVFrags + kSqFrags ? kQKVFrags : kQKVFrags + kSqFrags;
        constexpr int pend = n < kQKVFrags ? pbeg + kHeadFrags : pbeg + kSqFrags;
        if constexpr ((n + kDepth) % kChunk == 0 && n + kDepth < NFRAG) {
            constexpr int c = (n + kDepth) / kChunk;           // chunk c is about to be read
            // (PERS: the stream runs on into the next item's chunks 0 and 1 - issued below whether or not there is a next item: a
            // wasted 64 KiB of L2 traffic at the end of a workgroup's life buys one form of the counted wait)
            unsigned long long tb0 = 0, tb1 = 0;               // (stamped build: cycles parked at this chunk's wait + barrier)
            if constexpr (STAMP) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tb0) :: "memory");
            if constexpr (c + 1 < NCH || (PERS && !(TW_PERS_EXP & 1))) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");   // all but chunk c + 1's pieces
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if constexpr (STAMP) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tb1) :: "memory");
            asm volatile("s_barrier" ::: "memory");            // every wave's pieces of chunk c landed; chunk c - 1 is consumed
            if constexpr (STAMP) {
                unsigned long long tb2;
                asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tb2) :: "memory");
                t_dma += tb1 - tb0;
                t_bar += tb2 - tb1;
            }
            if constexpr (!TW_SPREAD) {
                if constexpr (c + 2 < NCH) issue_chunk(std::integral_constant<int, c + 2>{});
                else if constexpr (PERS && !(TW_PERS_EXP & 1)) issue_chunk(std::integral_constant<int, c + 2 - NCH>{});
            }
        }
        if constexpr (TW_SPREAD && n + kDepth >= kChunk && n + kDepth < NFRAG && (n + kDepth) % 4 == 0) {
            // TW_SPREAD: the 8 pieces of chunk c + 2 one by one, every fourth product of chunk c's period (all eight are out before
            // the next chunk wait, so the counted waits keep their meaning) instead of back to back behind the barrier: an LDS-DMA
            // issued among MFMAs costs the wave 100-185 cycles where eight of them queue up (MI355X_MICROARCH.md, kernel-cost table)
            constexpr int c = (n + kDepth) / kChunk, i = ((n + kDepth) % kChunk) / 4;
            if constexpr (c + 2 < NCH) issue_piece(std::integral_constant<int, c + 2>{}, std::integral_constant<int, i>{});
            else if constexpr (PERS && !(TW_PERS_EXP & 1)) issue_piece(std::integral_constant<int, c + 2 - NCH>{}, std::integral_constant<int, i>{});
        }
        if constexpr (n == pbeg) static_for<0, kDepth>([&](auto dc) { ring_read(std::integral_constant<int, pbeg + decltype(dc)::value>{}); });
        if constexpr (n + kDepth < pend) ring_read(std::integral_constant<int, n + kDepth>{});
        constexpr int after = pend - 1 - n < kDepth ? pend - 1 - n : kDepth;
        bf16x8& f = fr[n % kFr];                                 // (named first: an asm operand alone does not capture `fr`)
        asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(f) : "n"(after) : "memory");
    };

    // ---- prologue: parameter table, h1 = LN(h0 + pos) ---------------------------------------------------------------
    // h0 rows are 1.5 KiB of fp32 each; a lane needs 16-byte pieces of ONE row (fragment-shaped loads from HBM cost a third of
    // the kernel: 32 lines per instruction, four serialised round trips).  Instead the wave's 32 rows come by LDS-DMA, whole
    // half-rows (768 B) at a time, into 24 KiB of the LDS the weight ring and the K / V exchange do not use yet; the image is
    // ROTATED by the row (16-byte position p of row r holds chunk (p + r) mod 48 - done on the per-lane SOURCE address, the
    // LDS side of an LDS-DMA is linear), so the ds_read_b128 of 16 lanes = 16 different rows hit 16 different bank groups.
    bf16x8 X1[kNKS];     // h1^T (later h2^T) as MFMA operand fragments: lane = (row r, half h), k-step ks, 8 features
    float x[kNKS][8];    // h0 + pos for the lane's 192 features, parked in the accumulator half of the register file (x is needed
                         // twice: for the statistics and for the normalisation)
    float s = 0.f, q = 0.f;
    if constexpr (H16) {
        // bf16 h0 (K4b writes it so: half the bytes both ways): a row is 768 B, the lane pair (r, 0), (r, 1) reads 32 contiguous
        // bytes of row r per k-step - features 16 ks + 8 h .. + 7 - and one v_permlane32_swap per dword pair turns that into the
        // fragment's feature set {16 ks + 4 h + 0..3, 16 ks + 8 + 4 h + 0..3}.  All 24 loads (96 VGPRs) fly at once: ONE HBM round
        // trip, no LDS staging, no rotation.
        const int l = 32 * tile + r, lrow = l < nrows ? l : nrows - 1;                          // rows past the sequence: a finite copy
        const u32x4v* src = reinterpret_cast<const u32x4v*>(p.h0b[branch] + ((size_t)row0 + lrow) * kHidden + 8 * h);
        u32x4v raw[kNKS];
#pragma unroll
        for (int ks = 0; ks < kNKS; ++ks) raw[ks] = __builtin_nontemporal_load(src + 2 * ks);
        pstamp();                                                      // p0: row loads issued
        if constexpr (!PERS) {
            // parameter table
            const f32x4* psrc = reinterpret_cast<const f32x4*>(wsrc + (size_t)NFRAG * 1024);
            f32x4* dst = reinterpret_cast<f32x4*>(par);
            for (int i = tid; i < (P_TOTAL + P_TAIL) / 4; i += 256) dst[i] = psrc[i];
        }
        const float* posf = reinterpret_cast<const float*>(wsrc + (size_t)NFRAG * 1024 + (P_TOTAL + P_TAIL) * 4) +
                            (size_t)tile * kPosTile + lane * 8;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            f32x4 pz[12][2];
#pragma unroll
            for (int k2 = 0; k2 < 12; ++k2) {
                pz[k2][0] = *reinterpret_cast<const f32x4*>(posf + (12 * half + k2) * 512);
                pz[k2][1] = *reinterpret_cast<const f32x4*>(posf + (12 * half + k2) * 512 + 4);
            }
#pragma unroll
            for (int k2 = 0; k2 < 12; ++k2) {
                const int ks = 12 * half + k2;
                const auto s0 = __builtin_amdgcn_permlane32_swap(raw[ks][0], raw[ks][2], false, false);
                const auto s1 = __builtin_amdgcn_permlane32_swap(raw[ks][1], raw[ks][3], false, false);
                const unsigned w[4] = {(unsigned)s0[0], (unsigned)s1[0], (unsigned)s0[1], (unsigned)s1[1]};   // feature pairs in order
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float hv = h16_bits_to_f32((unsigned short)((e & 1) ? (w[e >> 1] >> 16) : (w[e >> 1] & 0xffffu)));
                    const float v = hv + pz[k2][e >> 2][e & 3];
                    s += v;
                    q += v * v;
                    x[ks][e] = v;
                    asm volatile("" : "+a"(x[ks][e]));
                }
            }
            if (half == 0) pstamp();                                   // p1: rows landed, first half combined
        }
        pstamp(); pstamp(); pstamp();                                  // (p2 .. p4: stages of the fp32 prologue only)
    } else {
        const uint32_t stage = smem_lds + kInStage + wave * (24 * 1024);
        // wave-uniform (one sequence per wave); made provably so for the asm's "s" operand (cdna_hip_programming.md T20)
        const uint64_t hs64 = reinterpret_cast<uint64_t>(p.h0[branch] + (size_t)row0 * kHidden);
        const uint32_t hs_hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(hs64 >> 32));
        const uint32_t hs_lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)hs64);      // (the builtin returns a SIGNED int)
        const char* hsrc = reinterpret_cast<const char*>(((uint64_t)hs_hi << 32) | (uint64_t)hs_lo);
        uint32_t vsrc[24];                                             // per piece: this lane's source offset inside the sequence's rows
#pragma unroll
        for (int j = 0; j < 24; ++j) {
            const int t48 = (64 * j) % 48 + lane, d48 = (t48 >= 48) + (t48 >= 96);              // (no integer division by 48 per lane)
            const int rl = (64 * j) / 48 + d48, pp = t48 - 48 * d48;                            // LDS slot 64 j + lane -> (row, position)
            const int l = 32 * tile + rl, lrow = l < nrows ? l : nrows - 1;                     // rows past the sequence: a finite copy
            vsrc[j] = (uint32_t)(lrow * (kHidden * 4) + ((pp + rl) % 48) * 16);
        }
        auto dma_half = [&](int half) {
#pragma unroll
            for (int j = 0; j < 24; ++j) glds_piece(vsrc[j], hsrc + half * 768, stage + j * 1024);
        };
        dma_half(0);
        pstamp();                                                      // p0: first half's DMA issued
        {   // parameter table (compiler-managed loads: their wait coincides with the wait for the first half)
            const f32x4* src = reinterpret_cast<const f32x4*>(wsrc + (size_t)NFRAG * 1024);
            f32x4* dst = reinterpret_cast<f32x4*>(par);
            for (int i = tid; i < (P_TOTAL + P_TAIL) / 4; i += 256) dst[i] = src[i];
        }
        const float* posf = reinterpret_cast<const float*>(wsrc + (size_t)NFRAG * 1024 + (P_TOTAL + P_TAIL) * 4) +
                            (size_t)tile * kPosTile + lane * 8;
        const int rot = 48 - r;                                        // (c - r) mod 48 = (c + rot) mod 48, c < 48
        const char* stg_lane = smem + kInStage + wave * (24 * 1024) + r * 768;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            f32x4 pz[12][2];
#pragma unroll
            for (int k2 = 0; k2 < 12; ++k2) {
                pz[k2][0] = *reinterpret_cast<const f32x4*>(posf + (12 * half + k2) * 512);
                pz[k2][1] = *reinterpret_cast<const f32x4*>(posf + (12 * half + k2) * 512 + 4);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // this half's rows (and the parameter / position loads) landed
            pstamp();                                                  // p1 / p3: half landed
#pragma unroll
            for (int k2 = 0; k2 < 12; ++k2)
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    int pos16 = 4 * k2 + 2 * c + h + rot;
                    pos16 = pos16 >= 48 ? pos16 - 48 : pos16;
                    const f32x4 a = *reinterpret_cast<const f32x4*>(stg_lane + pos16 * 16);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float v = a[e] + pz[k2][c][e];
                        s += v;
                        q += v * v;
                        x[12 * half + k2][4 * c + e] = v;
                        asm volatile("" : "+a"(x[12 * half + k2][4 * c + e]));
                    }
                }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");         // the staging reads are done before the next half overwrites them
            pstamp();                                                  // p2 / p4: half read
            if (half == 0) dma_half(1);
        }
    }
    {
        const float mean = half_swap_sum(s) * (1.f / kHidden);
        const float rstd = rsqrtf(fmaxf(half_swap_sum(q) * (1.f / kHidden) - mean * mean, 0.f) + 1e-5f);
        const float nmr = -mean * rstd;
        // (PERS: also the previous item's row stores and the look-ahead chunks 0 and 1; the barrier below is the item boundary -
        // every wave is past the previous item's last chunk and its output staging in the K / V region)
        if constexpr (H16) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // weight chunk 0 landed (older than the row loads)
        __syncthreads();                                               // parameter table visible; every wave is done with its staging
        pstamp();                                                      // p5: workgroup met
        if constexpr (PERS && (TW_PERS_EXP & 1)) {                     // (experiment: no look-ahead chunks)
            if (item != first_item) { issue_chunk(std::integral_constant<int, 0>{}); issue_chunk(std::integral_constant<int, 1>{}); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __syncthreads(); }
        }
        if constexpr (!PERS) issue_chunk(std::integral_constant<int, 1>{});    // (chunks 1 and 2 land where the staging was)
        issue_chunk(std::integral_constant<int, 2>{});                 // (PERS: chunk 1 is in flight since the previous item's tail;
                                                                       // slot 2 is free now: every wave is past that item's last chunk)
#pragma unroll
        for (int ks = 0; ks < kNKS; ++ks) {
            const f32x4* g = reinterpret_cast<const f32x4*>(par + P_G1 + ks * 16 + h * 8);
            const f32x4* b = reinterpret_cast<const f32x4*>(par + P_B1 + ks * 16 + h * 8);
            const f32x4 g0 = g[0], g1 = g[1], b0 = b[0], b1 = b[1];
            float v[8];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                v[e] = (x[ks][e] * rstd + nmr) * g0[e] + b0[e];
                v[4 + e] = (x[ks][4 + e] * rstd + nmr) * g1[e] + b1[e];
            }
            X1[ks] = pack8(v);
            asm volatile("" : "+a"(X1[ks]));                           // MFMA operand for the rest of the kernel: accumulator half
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    asm volatile("; TW_STREAM_BEGIN" ::: "memory");
    stamp();                                                           // [1] prologue done

    // T-product: acc[feature][row] += W fragment (A) x activation fragment (B); N-product: acc[row][feature] (V)
    auto tprod = [&](auto n0c, f32x16& acc, const bf16x8 (&X)[kNKS]) {
        static_for<0, kNKS>([&](auto ksc) {
            constexpr int ks = decltype(ksc)::value, n = decltype(n0c)::value + ks;
            pre(std::integral_constant<int, n>{});
            acc = h16_mfma32(fr[n % kFr], X[ks], acc);
        });
    };
    auto nprod = [&](auto n0c, f32x16& acc, const bf16x8 (&X)[kNKS]) {
        static_for<0, kNKS>([&](auto ksc) {
            constexpr int ks = decltype(ksc)::value, n = decltype(n0c)::value + ks;
            pre(std::integral_constant<int, n>{});
            acc = h16_mfma32(X[ks], fr[n % kFr], acc);
        });
    };
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    // The per-feature vector of a T-product (bias) goes in as the INITIAL accumulator - four 16-byte LDS reads straight into the
    // tile's registers (fragment order: register 8 s + j of lane half h is feature 32 t + 16 s + 8 h' ...) - instead of 16 adds behind
    // the chain: the kernel is issue-bound, every VALU instruction between two MFMA chains shows.
    auto bias_tile = [&](int tab, int t) {
        f32x16 a;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const f32x4* b = reinterpret_cast<const f32x4*>(par + tab + (2 * t + s) * 16 + h * 8);
            const f32x4 b0 = b[0], b1 = b[1];
#pragma unroll
            for (int e = 0; e < 4; ++e) { a[8 * s + e] = b0[e]; a[8 * s + 4 + e] = b1[e]; }
        }
        return a;
    };
    // accumulator tile (features 32 t .., bias already in) -> the two operand fragments of k-steps 2t, 2t+1
    auto tile_pack = [&](const f32x16& acc, bf16x8& f0, bf16x8& f1) {
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = acc[8 * s + e];
            (s == 0 ? f0 : f1) = pack8(v);
        }
    };

    bf16x8* Kl = reinterpret_cast<bf16x8*>(smem + kLdsKV);             // [slot 4][k-step 6][lane]
    bf16x8* Vl = Kl + 4 * 6 * 64;                                      // [slot 4][d tile 3][s 2][lane]
    bf16x8 Cf[kNKS];                                                   // ctx^T fragments (all heads)
    const int krem = len - 32 * (ntiles - 1) - 4 * h;                  // valid keys of the last key tile, from this lane half's first

    static_for<0, 4>([&](auto hdc) {
        constexpr int hd = decltype(hdc)::value, n0 = hd * kHeadFrags;
        bf16x8 Qf[6], Kf[6], Vf[6];
        static_for<0, 3>([&](auto dtc) {                               // Q^T (pre-scaled by log2(e) / sqrt(96) in the blob)
            constexpr int dt = decltype(dtc)::value;
            f32x16 a = bias_tile(P_BQ, 3 * hd + dt);
            tprod(std::integral_constant<int, n0 + dt * kNKS>{}, a, X1);
            tile_pack(a, Qf[2 * dt], Qf[2 * dt + 1]);
        });
        static_for<0, 3>([&](auto dtc) {                               // K^T
            constexpr int dt = decltype(dtc)::value;
            f32x16 a = bias_tile(P_BK, 3 * hd + dt);
            tprod(std::integral_constant<int, n0 + 72 + dt * kNKS>{}, a, X1);
            tile_pack(a, Kf[2 * dt], Kf[2 * dt + 1]);
        });
        static_for<0, 3>([&](auto dtc) {                               // V (rows on registers, d on lanes)
            constexpr int dt = decltype(dtc)::value;
            f32x16 a = zero16;
            nprod(std::integral_constant<int, n0 + 144 + dt * kNKS>{}, a, X1);
            // (the value bias is not added here: softmax rows sum to 1, so P (V + 1 bv^T) = P V + bv^T - it enters the context as the
            // addend of the normalising multiply below, which becomes an FMA: 16 VALU per tile less)
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                float v[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = a[8 * s + j];
                Vf[2 * dt + s] = pack8(v);
            }
        });
        stamp();                                                       // [2 + 2 hd] this head's q | k | v projection done
        __syncthreads();                                               // every wave is done with the previous head's K / V
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            Kl[(wave * 6 + i) * 64 + lane] = Kf[i];
            Vl[(wave * 6 + i) * 64 + lane] = Vf[i];
        }
        __syncthreads();
        // S^T[key][query] (log2 domain), keys on registers; the keys past the sequence (only in its last key tile) enter as
        // -inf through the INITIAL accumulator; key tiles past the sequence are skipped
        f32x16 sa[4];
        static_for<0, 4>([&](auto ktc) {
            constexpr int kt = decltype(ktc)::value;
            if (kt < ntiles) {
#pragma unroll
                for (int e = 0; e < 16; ++e) sa[kt][e] = (kt == ntiles - 1 && (e & 3) + 8 * (e >> 2) >= krem) ? -INFINITY : 0.f;
#pragma unroll
                for (int i = 0; i < 6; ++i)
                    sa[kt] = h16_mfma32(Kl[((first + kt) * 6 + i) * 64 + lane], Qf[i], sa[kt]);
            }
        });
        // (key tiles past the sequence - 4 - ntiles of them, wave-uniform - take no part: no -inf fill, no max / exp / sum over them:
        // on the ragged gallery a third of the softmax's VALU)
        float mx = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
            if (kt < ntiles) {
#pragma unroll
                for (int e = 0; e < 16; ++e) mx = fmaxf(mx, sa[kt][e]);
            }
        mx = half_swap_max(mx);
        float sum = 0.f;
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
            if (kt < ntiles) {
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    sa[kt][e] = __builtin_amdgcn_exp2f(sa[kt][e] - mx);
                    sum += sa[kt][e];
                }
            }
        const float inv = 1.f / half_swap_sum(sum);
        f32x16 oa[3] = {zero16, zero16, zero16};
        static_for<0, 4>([&](auto ktc) {
            constexpr int kt = decltype(ktc)::value;
            if (kt < ntiles) {
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    float v[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[j] = sa[kt][8 * s + j];
                    const bf16x8 pf = pack8(v);
#pragma unroll
                    for (int dt = 0; dt < 3; ++dt)
                        oa[dt] = h16_mfma32(Vl[((first + kt) * 6 + 2 * dt + s) * 64 + lane], pf, oa[dt]);
                }
            }
        });
#pragma unroll
        for (int dt = 0; dt < 3; ++dt)
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                // context feature of register 8 s + j, lane half h: 96 hd + 32 dt + 16 s + 8 (j >> 2) + 4 h + (j & 3) (accumulator rows)
                const float* bvp = par + P_BV + 96 * hd + 32 * dt + 16 * s + 4 * h;
                const f32x4 bv0 = *reinterpret_cast<const f32x4*>(bvp), bv1 = *reinterpret_cast<const f32x4*>(bvp + 8);
                float v[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = oa[dt][8 * s + j] * inv + (j < 4 ? bv0[j & 3] : bv1[j & 3]);
                Cf[6 * hd + 2 * dt + s] = pack8(v);
                asm volatile("" : "+a"(Cf[6 * hd + 2 * dt + s]));
            }
        stamp();                                                       // [3 + 2 hd] this head's attention done
    });

    // ---- dense + residual + LayerNorm -> h2^T (fp32 in `val`, bf16 fragments back into X1) ---------------------------
    float val[12][16];
    float s1 = 0.f, s2 = 0.f, s3 = 0.f;
    static_for<0, 12>([&](auto otc) {
        constexpr int ot = decltype(otc)::value;
        f32x16 a = bias_tile(P_BD, ot);
        tprod(std::integral_constant<int, kQKVFrags + ot * kNKS>{}, a, Cf);
#pragma unroll
        for (int s = 0; s < 2; ++s) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float v = a[8 * s + j] + h16_bits_to_f32((unsigned short)X1[2 * ot + s][j]);
                s1 += v;
                s2 += v * v;
                val[ot][8 * s + j] = v;
            }
            if constexpr (OUTMODE == 2) {          // sum_f (w gamma2)_f v_f: the modular logit, up to the row's LayerNorm statistics
                const f32x4* wg = reinterpret_cast<const f32x4*>(par + P_WG + (2 * ot + s) * 16 + h * 8);
                const f32x4 w0 = wg[0], w1 = wg[1];
#pragma unroll
                for (int j = 0; j < 8; ++j) s3 += val[ot][8 * s + j] * (j < 4 ? w0[j & 3] : w1[j & 3]);
            }
        }
    });
    stamp();                                                           // [10] dense done
    if constexpr (OUTMODE == 2) {
        // h2_l = (v_l rstd_l + nmr_l) gamma + beta is never formed.  With u_l = a_l rstd_l and U = sum_l a_l nmr_l (sum_l a_l = 1):
        //   logit_l = rstd_l (w gamma . v_l) + nmr_l c1 + c2         pooled_f = gamma_f (sum_l u_l v_l[f] + U) + beta_f
        asm volatile("; TW_STREAM_END" ::: "memory");
        const float mean = half_swap_sum(s1) * (1.f / kHidden);
        const float rstd = rsqrtf(fmaxf(half_swap_sum(s2) * (1.f / kHidden) - mean * mean, 0.f) + 1e-5f);
        const float nmr = -mean * rstd;
        float logit = rstd * half_swap_sum(s3) + nmr * par[P_TOTAL] + par[P_TOTAL + 1];
        if (r >= len) logit = -1e10f;                                  // mask_logits, model.py:444-445 (exactly -1e10)
        float mxl = logit;
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) mxl = fmaxf(mxl, __shfl_xor(mxl, o));
        const float e = __expf(logit - mxl);
        float den = e;
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) den += __shfl_xor(den, o);
        const float a = e / den;
        float U = a * nmr;
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) U += __shfl_xor(U, o);
        const float u = a * rstd;
        if (!live) return;                                             // (no barrier below this line)
        char* stg = smem + kLdsKV + wave * (32 * kStgPitch);
        float* o = p.pooled[branch] + (size_t)seq * kHidden;
#pragma unroll
        for (int pass = 0; pass < 4; ++pass) {
#pragma unroll
            for (int t = 0; t < 3; ++t)
#pragma unroll
                for (int q4 = 0; q4 < 4; ++q4) {
                    const f32x4 v = {val[3 * pass + t][4 * q4] * u, val[3 * pass + t][4 * q4 + 1] * u, val[3 * pass + t][4 * q4 + 2] * u,
                                     val[3 * pass + t][4 * q4 + 3] * u};
                    *reinterpret_cast<f32x4*>(stg + r * kStgPitch + (32 * t + 8 * q4 + 4 * h) * 4) = v;
                }
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                const int c = lane + 64 * half;                        // feature of this pass summed by this lane
                if (c < 96) {
                    float acc = 0.f;
#pragma unroll
                    for (int row = 0; row < 32; ++row) acc += *reinterpret_cast<const float*>(stg + row * kStgPitch + 4 * c);
                    const int f = 96 * pass + c, w16 = f & 15;
                    const int ti = (f >> 4) * 16 + ((w16 >> 2) & 1) * 8 + ((w16 >> 3) << 2) + (w16 & 3);   // f in fragment order
                    o[f] = len_raw >= 1 ? par[P_G2 + ti] * (acc + U) + par[P_B2 + ti] : 0.f;
                }
            }
        }
        return;
    }
    unsigned long long row_bad = 0ull;
    {
        const float mean = half_swap_sum(s1) * (1.f / kHidden);
        row_bad = __builtin_amdgcn_ballot_w64(!(fabsf(mean) <= 3.0e38f) && r < len);      // fp16 overflow guard: stored behind the stream
        const float rstd = rsqrtf(fmaxf(half_swap_sum(s2) * (1.f / kHidden) - mean * mean, 0.f) + 1e-5f);
        const float nmr = -mean * rstd;
#pragma unroll
        for (int t = 0; t < 12; ++t)
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                float v[8];
                if constexpr (OUTMAP && TW_FOLD_LN2) {
                    // gamma2 / beta2 live in the out mapping's weights and bias (folded at pack time: y = xhat2 (Wo diag(gamma2))^T +
                    // (bo + Wo beta2)): the operand is the normalised row itself - one FMA per element, no vector reads
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[j] = val[t][8 * s + j] * rstd + nmr;
                    X1[2 * t + s] = pack8(v);
                    asm volatile("" : "+a"(X1[2 * t + s]));
                } else {
                    const f32x4* g = reinterpret_cast<const f32x4*>(par + P_G2 + (2 * t + s) * 16 + h * 8);
                    const f32x4* b = reinterpret_cast<const f32x4*>(par + P_B2 + (2 * t + s) * 16 + h * 8);
                    const f32x4 g0 = g[0], g1 = g[1], b0 = b[0], b1 = b[1];
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        v[j] = (val[t][8 * s + j] * rstd + nmr) * (j < 4 ? g0[j & 3] : g1[j & 3]) + (j < 4 ? b0[j & 3] : b1[j & 3]);
                        val[t][8 * s + j] = v[j];
                    }
                    if constexpr (OUTMAP) {
                        X1[2 * t + s] = pack8(v);
                        asm volatile("" : "+a"(X1[2 * t + s]));
                    }
                }
            }
    }
    stamp();                                                           // [11] LayerNorm done
    if constexpr (OUTMAP) {
        static_for<0, 12>([&](auto otc) {
            constexpr int ot = decltype(otc)::value;
            f32x16 a = bias_tile(P_BO, ot);
            tprod(std::integral_constant<int, kQKVFrags + kSqFrags + ot * kNKS>{}, a, X1);
#pragma unroll
            for (int e = 0; e < 16; ++e) val[ot][e] = a[e];
        });
    }
    asm volatile("; TW_STREAM_END" ::: "memory");
    stamp();                                                           // [12] out mapping done
    if (row_bad != 0ull && p.nonfinite != nullptr && lane == 0) *p.nonfinite = 1;   // (TowerArgs.nonfinite; behind the counted stream)
    if constexpr (PERS) {
        if (has_next && !(TW_PERS_EXP & 2)) touch_rows(ent_nxt, row0_nxt);                   // the next item's h0 rows on their way to L2 under the row stores below
    }
    if (!PERS && !live) return;                                        // (no barrier below this line)
    if (live) {

    // ---- output: val[t][e] = y^T[feature 32 t + (e & 3) + 8 (e >> 2) + 4 h][row r] ----------------------------------
    // (the K / V region is free: the dense / out-mapping chunks put barriers between the last head and here; each wave stages
    // through its own 12.5 KiB of it).  The wave of a sequence's LAST tile also writes the zero rows behind the sequence.
    char* stg = smem + kLdsKV + wave * (32 * kStgPitch);
    const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
    if constexpr (OUTMODE == 1) {
        float ss = 0.f;
#pragma unroll
        for (int t = 0; t < 12; ++t)
#pragma unroll
            for (int e = 0; e < 16; ++e) ss += val[t][e] * val[t][e];
        const float scale = 1.f / fmaxf(sqrtf(half_swap_sum(ss)), 1e-12f);       // F.normalize, model.py:319
        const int len16 = (len + 15) & ~15;
        const int lastr = (len - 1) & 31;                                          // replicated into its 16-row tile's padding
        char* g = p.gal[branch] + (size_t)(p.v0 + seq) * p.Lp * (kHidden * 2);
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) {
#pragma unroll
            for (int t = 0; t < 6; ++t)
#pragma unroll
                for (int q4 = 0; q4 < 4; ++q4) {
                    uint2 w2;
                    w2.x = (unsigned)f32_to_bf16_bits(val[6 * pass + t][4 * q4 + 0] * scale) |
                           ((unsigned)f32_to_bf16_bits(val[6 * pass + t][4 * q4 + 1] * scale) << 16);
                    w2.y = (unsigned)f32_to_bf16_bits(val[6 * pass + t][4 * q4 + 2] * scale) |
                           ((unsigned)f32_to_bf16_bits(val[6 * pass + t][4 * q4 + 3] * scale) << 16);
                    *reinterpret_cast<uint2*>(stg + r * kStgPitch + (32 * t + 8 * q4 + 4 * h) * 2) = w2;
                }
#pragma unroll
            for (int it = 0; it < 12; ++it) {
                const int idx = lane + 64 * it, row = idx / 24, c = idx % 24;
                const int l = 32 * tile + row;
                if (l >= p.Lp || (p.skip_zero_rows && l >= len16)) continue;
                f32x4 v = z4;
                if (l < len16) v = *reinterpret_cast<const f32x4*>(stg + (l < len ? row : lastr) * kStgPitch + 16 * c);
                *reinterpret_cast<f32x4*>(g + (size_t)l * (kHidden * 2) + pass * 384 + 16 * c) = v;
            }
        }
        if (tile == ntiles - 1 && !p.skip_zero_rows)
            for (int i = 32 * ntiles * 48 + lane; i < p.Lp * 48; i += 64) *reinterpret_cast<f32x4*>(g + (size_t)i * 16) = z4;
        if (p.lens_out != nullptr && branch == 0 && tile == 0 && lane == 0) p.lens_out[p.v0 + seq] = len;
    } else {
        float* o = p.out[branch] + (size_t)row0 * kHidden;
#pragma unroll
        for (int pass = 0; pass < 4; ++pass) {
#pragma unroll
            for (int t = 0; t < 3; ++t)
#pragma unroll
                for (int q4 = 0; q4 < 4; ++q4) {
                    const f32x4 v = {val[3 * pass + t][4 * q4], val[3 * pass + t][4 * q4 + 1], val[3 * pass + t][4 * q4 + 2],
                                     val[3 * pass + t][4 * q4 + 3]};
                    *reinterpret_cast<f32x4*>(stg + r * kStgPitch + (32 * t + 8 * q4 + 4 * h) * 4) = v;
                }
#pragma unroll
            for (int it = 0; it < 12; ++it) {
                const int idx = lane + 64 * it, row = idx / 24, c = idx % 24;
                const int l = 32 * tile + row;
                if (l >= nrows && l >= p.seq_rows) continue;
                f32x4 v = z4;
                if (l < nrows) v = *reinterpret_cast<const f32x4*>(stg + row * kStgPitch + 16 * c);
                *reinterpret_cast<f32x4*>(o + (size_t)l * kHidden + pass * 96 + 4 * c) = v;
            }
        }
        if (p.items != nullptr && tile == ntiles - 1)                  // with an item table: zero rows behind the sequence
            for (int i = 32 * ntiles * 96 + lane; i < p.seq_rows * 96; i += 64) *reinterpret_cast<f32x4*>(o + (size_t)i * 4) = z4;
    }
    }   // if (live)
    if constexpr (STAMP) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        stamp();                                                       // [13] rows stored
        if (tid == 0 && p.stamps != nullptr) {
            for (int i = 0; i < 16; ++i) p.stamps[(size_t)blockIdx.x * 24 + i] = i < n_ts ? ts[i] : 0ull;
            p.stamps[(size_t)blockIdx.x * 24 + 14] = t_dma;           // wave 0: cycles in the 44 chunk waits (its own LDS-DMA pieces) ...
            p.stamps[(size_t)blockIdx.x * 24 + 15] = t_bar;           // ... and in the barriers behind them (the other waves' pieces)
            for (int i = 0; i < 8; ++i) p.stamps[(size_t)blockIdx.x * 24 + 16 + i] = i < n_tp ? tp[i] : 0ull;
        }
    }
    if (!has_next) break;
    item += stride;
    ent_cur = ent_nxt;
    row0_cur = row0_nxt;
    ent_nxt = ent_nn;
    }   // for (;;): the items of a persistent workgroup
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                  // (PERS: the look-ahead LDS-DMAs must not outlive the workgroup)
}

// ---- weight / parameter packing -----------------------------------------------------------------------------------
struct PackArgs {
    const float *g1, *b1, *wq, *bq, *wk, *bk, *wv, *bv, *wd, *bd, *g2, *b2, *wo, *bo, *mw, *pos;
    int max_pos;
    unsigned short* frags;
    float* par;
    int nfrag;
};
// fragment n (stream order), lane, element j  <-  W[row base + (lane & 31)][16 ks + 8 (j >> 2) + 4 (lane >> 5) + (j & 3)]
__global__ __launch_bounds__(256) void tower_pack_kernel(const PackArgs a) {
    const float qscale = 1.4426950408889634f * 0.10206207261596577f;     // log2(e) / sqrt(96): softmax runs on exp2
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < (long)a.nfrag * 512) {
        const int n = (int)(i >> 9), lane = (int)(i >> 3) & 63, j = (int)i & 7;
        const float* W;
        int rowbase, ks;
        float sc = 1.f;
        if (n < kQKVFrags) {
            const int hd = n / kHeadFrags, m = n % kHeadFrags, which = m / 72, dt = (m % 72) / kNKS;
            ks = m % kNKS;
            W = which == 0 ? a.wq : which == 1 ? a.wk : a.wv;
            if (which == 0) sc = qscale;
            rowbase = 96 * hd + 32 * dt;
        } else {
            const int m = (n - kQKVFrags) % kSqFrags;
            W = n < kQKVFrags + kSqFrags ? a.wd : a.wo;
            rowbase = 32 * (m / kNKS);
            ks = m % kNKS;
        }
        const int f = 16 * ks + 8 * (j >> 2) + 4 * (lane >> 5) + (j & 3);
        if (TW_FOLD_LN2 && n >= kQKVFrags + kSqFrags) sc = a.g2[f];             // out mapping: LayerNorm 2's gamma folded in (its beta: the bias below)
        a.frags[i] = f32_to_h16_bits(W[(size_t)(rowbase + (lane & 31)) * kHidden + f] * sc);
    }
    if (i < P_TOTAL) {
        const int tab = (int)i / kHidden, x = (int)i % kHidden;
        const int ks = x >> 4, hh = (x >> 3) & 1, j = x & 7;
        const int f = 16 * ks + 8 * (j >> 2) + 4 * hh + (j & 3);             // fragment order: [ks][half][j]
        float v;
        switch (tab) {
            case 0: v = a.g1[f]; break;
            case 1: v = a.b1[f]; break;
            case 2: v = a.bq[f] * qscale; break;
            case 3: v = a.bk[f]; break;
            case 4: v = a.bv[x]; break;                                        // V's features sit on lanes: natural order
            case 5: v = a.bd[f]; break;
            case 6: v = a.g2[f]; break;
            case 7: v = a.b2[f]; break;
            default:
                if (a.bo) {                                                    // bo + Wo beta2 (see the out mapping's fragments)
                    v = a.bo[f];
                    for (int c = 0; TW_FOLD_LN2 && c < kHidden; ++c) v += a.wo[(size_t)f * kHidden + c] * a.b2[c];
                } else {
                    v = a.mw ? a.mw[f] * a.g2[f] : 0.f;
                }
                break;
        }
        a.par[i] = v;
    }
    if (i < 4L * kPosTile) {          // position table, 4 tiles of 32 positions in fragment order [tile][ks][lane][8]
        const int t = (int)(i / kPosTile), rem = (int)(i % kPosTile), ks = rem / 512, lane = (rem >> 3) & 63, j = rem & 7;
        const int l = 32 * t + (lane & 31), f = 16 * ks + 8 * (j >> 2) + 4 * (lane >> 5) + (j & 3);
        a.par[P_TOTAL + P_TAIL + i] = l < a.max_pos ? a.pos[(size_t)l * kHidden + f] : 0.f;
    }
    if (i == 0) {                     // query towers: the two constants of the folded modular logit
        float c1 = 0.f, c2 = 0.f;
        if (a.mw)
            for (int f = 0; f < kHidden; ++f) { c1 += a.mw[f] * a.g2[f]; c2 += a.mw[f] * a.b2[f]; }
        a.par[P_TOTAL] = c1; a.par[P_TOTAL + 1] = c2; a.par[P_TOTAL + 2] = 0.f; a.par[P_TOTAL + 3] = 0.f;
    }
}

}  // namespace tw
}  // namespace dldkd

using namespace dldkd;

extern "C" {

size_t dldkd_tower_blob_bytes(int with_out_map) {
    const size_t nfrag = tw::kQKVFrags + tw::kSqFrags + (with_out_map ? tw::kSqFrags : 0);
    return nfrag * 1024 + (size_t)(tw::P_TOTAL + tw::P_TAIL + 4 * tw::kPosTile) * 4;
}

int dldkd_tower_pack_h16(const float* ln1_g, const float* ln1_b, const float* wq, const float* bq, const float* wk, const float* bk,
                          const float* wv, const float* bv, const float* wd, const float* bd, const float* ln2_g, const float* ln2_b,
                          const float* wo, const float* bo, const float* mod_w, const float* pos, int max_pos, void* blob, void* stream) {
    if (!ln1_g || !ln1_b || !wq || !bq || !wk || !bk || !wv || !bv || !wd || !bd || !ln2_g || !ln2_b || !blob || !pos || (!wo != !bo) ||
        (!wo == !mod_w) || max_pos < 1) {
        set_error("tower_pack: null pointer (a video tower has wo / bo, a query tower has mod_w) or max_pos < 1");
        return DLDKD_EINVAL;
    }
    if ((uintptr_t)blob & 15) { set_error("tower_pack: blob must be 16-byte aligned"); return DLDKD_EINVAL; }
    const int nfrag = tw::kQKVFrags + tw::kSqFrags + (wo ? tw::kSqFrags : 0);
    tw::PackArgs a{ln1_g, ln1_b, wq, bq, wk, bk, wv, bv, wd, bd, ln2_g, ln2_b, wo, bo, mod_w, pos, max_pos, (unsigned short*)blob,
                   (float*)((char*)blob + (size_t)nfrag * 1024), nfrag};
    DLDKD_LAUNCH(tw::tower_pack_kernel, dim3((unsigned)(((long)nfrag * 512 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a);
    return check_launch("tower_pack");
}

static int tower_seq_launch(const void* const* h0, int h16, const void* const* blob, const int32_t* row0,
                            const int32_t* lens, const int32_t* items, int n_items, int n_seq, int n_branches,
                            int out_mode, float* const* out_rows, int seq_rows, void* const* gallery, int v0, int Lp, int32_t* lens_out,
                            int32_t* nonfinite_flag, void* stream, int skip_zero_rows = 0);

int dldkd_tower_seq_h16(const float* const* h0, const void* const* blob, const int32_t* row0,
                         const int32_t* lens, const int32_t* items, int n_items, int n_seq, int n_branches,
                         int out_mode, float* const* out_rows, int seq_rows, void* const* gallery, int v0, int Lp, int32_t* lens_out,
                         int32_t* nonfinite_flag, void* stream) {
    return tower_seq_launch((const void* const*)h0, 0, blob, row0, lens, items, n_items, n_seq, n_branches, out_mode, out_rows, seq_rows,
                            gallery, v0, Lp, lens_out, nonfinite_flag, stream);
}

int dldkd_tower_seq_h16_rows16(const void* const* h0_bf16, const void* const* blob, const int32_t* row0,
                             const int32_t* lens, const int32_t* items, int n_items, int n_seq, int n_branches,
                             void* const* gallery, int v0, int Lp, int32_t* lens_out, int32_t* nonfinite_flag, int skip_zero_rows,
                             void* stream) {
    return tower_seq_launch(h0_bf16, 1, blob, row0, lens, items, n_items, n_seq, n_branches, 1, nullptr, 0, gallery, v0, Lp, lens_out,
                            nonfinite_flag, stream, skip_zero_rows);
}

static int tower_seq_launch(const void* const* h0, int h16, const void* const* blob, const int32_t* row0,
                            const int32_t* lens, const int32_t* items, int n_items, int n_seq, int n_branches,
                            int out_mode, float* const* out_rows, int seq_rows, void* const* gallery, int v0, int Lp, int32_t* lens_out,
                            int32_t* nonfinite_flag, void* stream, int skip_zero_rows) {
    if (n_items < 0 || n_seq < 0 || (n_branches != 1 && n_branches != 2) || out_mode < 0 || out_mode > 2 || seq_rows < 0 ||
        seq_rows > 128 || (!row0 && seq_rows < 1) || (out_mode == 1 && (Lp < 32 || Lp > 128 || (Lp & 31) || v0 < 0)) ||
        (out_mode == 2 && !items && n_items != (n_seq + 3) / 4)) {
        set_error("tower_seq: bad arguments (n_items=%d n_seq=%d n_branches=%d out_mode=%d seq_rows=%d Lp=%d)", n_items, n_seq,
                  n_branches, out_mode, seq_rows, Lp);
        return DLDKD_EINVAL;
    }
    if (n_items == 0) return DLDKD_OK;
    if (!h0 || !blob || !lens || (out_mode != 1 && !out_rows) || (out_mode == 1 && !gallery)) {
        set_error("tower_seq: null pointer");
        return DLDKD_EINVAL;
    }
    tw::TowerArgs p{};
    for (int b = 0; b < n_branches; ++b) {
        p.h0[b] = (const float*)h0[b]; p.h0b[b] = (const unsigned short*)h0[b]; p.blob[b] = (const char*)blob[b];
        if (out_mode == 0) p.out[b] = out_rows[b]; else if (out_mode == 2) p.pooled[b] = out_rows[b]; else p.gal[b] = (char*)gallery[b];
        if (!p.h0[b] || !p.blob[b] || (out_mode == 1 ? !p.gal[b] : !out_rows[b])) { set_error("tower_seq: null branch pointer"); return DLDKD_EINVAL; }
        if (((uintptr_t)p.h0[b] | (uintptr_t)p.blob[b] | (uintptr_t)p.out[b] | (uintptr_t)p.gal[b] | (uintptr_t)p.pooled[b]) & 15) {
            set_error("tower_seq: buffers must be 16-byte aligned");
            return DLDKD_EINVAL;
        }
    }
    p.row0 = row0; p.lens = lens; p.items = items; p.n_items = n_items; p.n_seq = n_seq; p.n_branches = n_branches;
    p.seq_rows = seq_rows; p.v0 = v0; p.Lp = Lp; p.lens_out = lens_out; p.nonfinite = nonfinite_flag; p.skip_zero_rows = skip_zero_rows ? 1 : 0;
    dim3 grid(n_branches == 2 ? 8u * (unsigned)((n_items + 3) / 4) : (unsigned)n_items);
    if (h16 && !(TW_PERS_EXP & 4)) {
        // persistent workgroups, one per CU (a multiple of 8 so that a workgroup's branch = its XCD half stays put): workgroup w walks
        // items w', w' + grid / 2, ... of its branch
        if (!items || !row0) { set_error("tower_seq_h16: needs the slot table and row0"); return DLDKD_EINVAL; }
        static int n_cu = 0;
        if (!n_cu) {
            int dev = 0, v = 0;
            if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v < 8) v = 256;
            n_cu = v & ~7;
        }
        if (grid.x > (unsigned)n_cu) grid.x = (unsigned)n_cu;
    }
    static const bool lds_ok = [] {           // once per process: the attribute call is a driver round trip
        return hipFuncSetAttribute((const void*)tw::tower_seq_kernel<true, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, tw::kLdsTotal) == hipSuccess &&
               hipFuncSetAttribute((const void*)tw::tower_seq_kernel<true, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, tw::kLdsTotal) == hipSuccess &&
               hipFuncSetAttribute((const void*)tw::tower_seq_kernel<true, 1, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, tw::kLdsTotal) == hipSuccess &&
               hipFuncSetAttribute((const void*)tw::tower_seq_kernel<false, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, tw::kLdsTotal) == hipSuccess;
    }();
    if (!lds_ok) {
        (void)hipGetLastError();
        set_error("tower_seq: cannot reserve %d bytes of LDS", tw::kLdsTotal);
        return DLDKD_ELAUNCH;
    }
    if (h16) DLDKD_LAUNCH((tw::tower_seq_kernel<true, 1, false, true>), grid, dim3(256), tw::kLdsTotal, (hipStream_t)stream, p);
    else if (out_mode == 1) DLDKD_LAUNCH((tw::tower_seq_kernel<true, 1>), grid, dim3(256), tw::kLdsTotal, (hipStream_t)stream, p);
    else if (out_mode == 0) DLDKD_LAUNCH((tw::tower_seq_kernel<true, 0>), grid, dim3(256), tw::kLdsTotal, (hipStream_t)stream, p);
    else DLDKD_LAUNCH((tw::tower_seq_kernel<false, 2>), grid, dim3(256), tw::kLdsTotal, (hipStream_t)stream, p);
    return check_launch("tower_seq");
}

/* Diagnostics: the gallery-mode kernel with s_memtime stamps at its phase boundaries; stamps = 16 x u64 per workgroup
 * (tools/tower_timeline.py): [0] start, [1] prologue, [2 + 2h] head h projected, [3 + 2h] head h attended, [10] dense,
 * [11] LayerNorm, [12] out mapping, [13] rows stored. */
int dldkd_debug_tower_seq_timeline(const float* const* h0, const void* const* blob, const int32_t* lens,
                                   const int32_t* items, int n_items, int n_seq, int seq_rows, void* const* gallery, int Lp,
                                   unsigned long long* stamps, int h16, void* stream) {
    if (!h0 || !blob || !lens || !gallery || !stamps || n_items < 1 || seq_rows < 1 || Lp < 32 || (Lp & 31)) {
        set_error("tower_seq_timeline: bad arguments");
        return DLDKD_EINVAL;
    }
    tw::TowerArgs p{};
    for (int b = 0; b < 2; ++b) { p.h0[b] = h0[b]; p.h0b[b] = (const unsigned short*)h0[b]; p.blob[b] = (const char*)blob[b]; p.gal[b] = (char*)gallery[b]; }
    p.lens = lens; p.items = items; p.n_items = n_items; p.n_seq = n_seq; p.n_branches = 2;
    p.seq_rows = seq_rows; p.v0 = 0; p.Lp = Lp; p.stamps = stamps;
    static const bool ok = hipFuncSetAttribute((const void*)tw::tower_seq_kernel<true, 1, true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                               tw::kLdsTotal) == hipSuccess &&
                           hipFuncSetAttribute((const void*)tw::tower_seq_kernel<true, 1, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                               tw::kLdsTotal) == hipSuccess;
    if (!ok) { (void)hipGetLastError(); set_error("tower_seq_timeline: cannot reserve LDS"); return DLDKD_ELAUNCH; }
    if (h16) DLDKD_LAUNCH((tw::tower_seq_kernel<true, 1, true, true>), dim3(8u * (unsigned)((n_items + 3) / 4)), dim3(256), tw::kLdsTotal, (hipStream_t)stream, p);
    else DLDKD_LAUNCH((tw::tower_seq_kernel<true, 1, true>), dim3(8u * (unsigned)((n_items + 3) / 4)), dim3(256), tw::kLdsTotal, (hipStream_t)stream, p);
    return check_launch("tower_seq_timeline");
}

}  // extern "C"
