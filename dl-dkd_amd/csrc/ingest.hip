// Feature ingest on the GPU: temporal down-sampling of raw frame features to at most
// max_ctx_l clips by segment means + per-clip L2 normalisation + zero padding, i.e. what
// VisDataSet4DLDKD.__getitem__ -> uniform_feature_sampling -> l2_normalize_np_array -> cat_videos do on the CPU
// one clip at a time (reference method/data_provider.py:52-86,283-309).  HBM-bound: every frame row is read once,
// one wave per output clip, 16-byte accesses.
#include "common.hpp"

namespace dldkd {

// out row r (of n_rows = B * Lmax): seg_start[r] < 0 -> padding (zeros); seg_start < seg_end -> mean of frames
// [seg_start, seg_end); seg_start == seg_end -> frame seg_start.  Then x / (||x|| + eps).
__global__ __launch_bounds__(256) void segment_mean_l2norm_kernel(const float* __restrict__ frames,
                                                                  const int32_t* __restrict__ seg_start,
                                                                  const int32_t* __restrict__ seg_end,
                                                                  float* __restrict__ out, long n_rows, int D, float eps) {
    const int lane = threadIdx.x & 63;
    const long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= n_rows) return;
    const int nv = D >> 2;
    f32x4* o = reinterpret_cast<f32x4*>(out + r * D);
    const int s = seg_start[r];
    if (s < 0) {
        for (int c = lane; c < nv; c += 64) o[c] = f32x4{0.f, 0.f, 0.f, 0.f};
        return;
    }
    int e = seg_end[r];
    if (e <= s) e = s + 1;
    const float inv_n = 1.f / (float)(e - s);
    float ss = 0.f;
    for (int c = lane; c < nv; c += 64) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        for (int f = s; f < e; ++f) acc += reinterpret_cast<const f32x4*>(frames + (size_t)f * D)[c];
        acc *= inv_n;
        o[c] = acc;
        ss += acc[0] * acc[0] + acc[1] * acc[1] + acc[2] * acc[2] + acc[3] * acc[3];
    }
    const float scale = 1.f / (sqrtf(wave_sum(ss)) + eps);
    for (int c = lane; c < nv; c += 64) { f32x4 v = o[c]; v *= scale; o[c] = v; }
}

// Small host-produced tables (slot / row-group tables of the fused gallery encode) -> device by a KERNEL that reads the pinned
// host buffer over the bus: the upload stays on the compute queue.  An asynchronous hipMemcpy of the same 4-12 KB goes through
// the copy engine, and the two cross-engine hand-offs per upload were measured at ~2.4 ms each on a loaded host (bench.py after
// its CPU baseline: gallery encode 0.107 s instead of 0.023 s, every kernel unchanged) against ~4 us for this kernel.
__global__ __launch_bounds__(256) void upload_words_kernel(const int32_t* __restrict__ src, int32_t* __restrict__ dst, long n) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) dst[i] = __builtin_nontemporal_load(src + i);
}

// Batch assembly from device-resident ragged tables (data.DeviceTrainSet): out[b, l, :] = src[row_start[items[b]] + l, :] for
// l < lens[items[b]], zeros beyond, mask[b, l] = 1 / 0 - what the reference's collate (pad to the longest item + mask,
// data_provider.py:75-86,111-136) produces on the host, followed by an H2D copy of the padded batch.  One wave per output row.
__global__ __launch_bounds__(256) void gather_pad_rows_kernel(const float* __restrict__ src, const long long* __restrict__ row_start,
                                                              const int32_t* __restrict__ lens, const int32_t* __restrict__ items,
                                                              float* __restrict__ out, float* __restrict__ mask, long n_rows, int Lmax,
                                                              int D) {
    const int lane = threadIdx.x & 63;
    const long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= n_rows) return;
    const int b = (int)(r / Lmax), l = (int)(r - (long)b * Lmax);
    const int it = items[b];
    const bool valid = l < lens[it];
    if (mask != nullptr && lane == 0) mask[r] = valid ? 1.f : 0.f;
    f32x4* o = reinterpret_cast<f32x4*>(out + r * D);
    const int nv = D >> 2;
    if (valid) {
        const f32x4* s = reinterpret_cast<const f32x4*>(src + (size_t)(row_start[it] + l) * D);
        for (int c = lane; c < nv; c += 64) o[c] = s[c];
    } else {
        for (int c = lane; c < nv; c += 64) o[c] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
}


// Raw feature rows -> the resident form K4b (in_proj_rows128b.hip) consumes: bf16 (round to nearest even: what K4's
// v_cvt_pk_bf16_f32 does to the same values on their way to the MFMA) + the row's fp32 LayerNorm statistics, taken from the
// fp32 values (two passes: mean, then the centred second moment - torch's LayerNorm, method/model_components.py:297,308).
// Valid rows of a padded (n_items, L, K) batch are APPENDED to a ragged table: item b's row l < lens[b] goes to table row
// dst_row0[b] + l.  One wave per source row; the second pass re-reads the row from L1 / L2.
__global__ __launch_bounds__(256) void rows_to_h16_stats_kernel(const float* __restrict__ src, const int32_t* __restrict__ lens,
                                                                 const long long* __restrict__ dst_row0, unsigned short* __restrict__ xb,
                                                                 float* __restrict__ mean, float* __restrict__ rstd, long n_rows, int L,
                                                                 int K, float eps) {
    const int lane = threadIdx.x & 63;
    const long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= n_rows) return;
    const int b = (int)(r / L), l = (int)(r - (long)b * L);
    if (lens != nullptr && l >= lens[b]) return;
    const long d = dst_row0 != nullptr ? (long)dst_row0[b] + l : r;
    const f32x4* s = reinterpret_cast<const f32x4*>(src + (size_t)r * K);
    const int nv = K >> 2;
    float sum = 0.f;
    for (int c = lane; c < nv; c += 64) { const f32x4 v = s[c]; sum += (v[0] + v[1]) + (v[2] + v[3]); }
    const float mu = wave_sum(sum) / (float)K;
    float sq = 0.f;
    uint2* o = reinterpret_cast<uint2*>(xb + (size_t)d * K);
    for (int c = lane; c < nv; c += 64) {
        const f32x4 v = s[c];
        const float d0 = v[0] - mu, d1 = v[1] - mu, d2 = v[2] - mu, d3 = v[3] - mu;
        sq += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
        uint2 pk;
        pk.x = (unsigned)f32_to_h16_bits(v[0]) | ((unsigned)f32_to_h16_bits(v[1]) << 16);     // h16 = fp16 (common.hpp)
        pk.y = (unsigned)f32_to_h16_bits(v[2]) | ((unsigned)f32_to_h16_bits(v[3]) << 16);
        o[c] = pk;
    }
    const float var = wave_sum(sq) / (float)K;
    if (lane == 0) {
        mean[d] = mu;
        rstd[d] = rsqrtf(var + eps);
    }
}

}  // namespace dldkd

using namespace dldkd;

extern "C" int dldkd_rows_to_h16_stats(const float* src, const int32_t* lens, const long long* dst_row0, int n_items, int L, int K,
                                        float eps, void* x_bf16, float* mean, float* rstd, void* stream) {
    if (n_items < 0 || L < 0 || K < 4 || (K & 3)) { set_error("rows_to_h16_stats: bad sizes (K must be a multiple of 4)"); return DLDKD_EINVAL; }
    const long n_rows = (long)n_items * L;
    if (n_rows == 0) return DLDKD_OK;
    if (!src || !x_bf16 || !mean || !rstd) { set_error("rows_to_h16_stats: null pointer"); return DLDKD_EINVAL; }
    if (((uintptr_t)src & 15) || ((uintptr_t)x_bf16 & 7)) { set_error("rows_to_h16_stats: unaligned buffer"); return DLDKD_EINVAL; }
    DLDKD_LAUNCH(rows_to_h16_stats_kernel, dim3((unsigned)((n_rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, src, lens, dst_row0,
                       (unsigned short*)x_bf16, mean, rstd, n_rows, L, K, eps);
    return check_launch("rows_to_h16_stats");
}

extern "C" int dldkd_gather_pad_rows_f32(const float* src, const long long* row_start, const int32_t* lens, const int32_t* items,
                                         int n_items, int Lmax, int D, float* out, float* mask, void* stream) {
    if (n_items < 0 || Lmax < 0 || D < 4 || (D & 3)) { set_error("gather_pad_rows: bad sizes (D must be a multiple of 4)"); return DLDKD_EINVAL; }
    const long n_rows = (long)n_items * Lmax;
    if (n_rows == 0) return DLDKD_OK;
    if (!src || !row_start || !lens || !items || !out) { set_error("gather_pad_rows: null pointer"); return DLDKD_EINVAL; }
    if (((uintptr_t)src | (uintptr_t)out) & 15) { set_error("gather_pad_rows: unaligned buffer"); return DLDKD_EINVAL; }
    DLDKD_LAUNCH(gather_pad_rows_kernel, dim3((unsigned)((n_rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, src, row_start, lens,
                       items, out, mask, n_rows, Lmax, D);
    return check_launch("gather_pad_rows");
}

extern "C" int dldkd_upload_words(const int32_t* pinned_src, int32_t* dst, long n_words, void* stream) {
    if (n_words < 0) { set_error("upload_words: bad size"); return DLDKD_EINVAL; }
    if (n_words == 0) return DLDKD_OK;
    if (!pinned_src || !dst) { set_error("upload_words: null pointer"); return DLDKD_EINVAL; }
    const long nb = (n_words + 255) / 256;
    DLDKD_LAUNCH(upload_words_kernel, dim3((unsigned)(nb < 64 ? nb : 64)), dim3(256), 0, (hipStream_t)stream, pinned_src, dst, n_words);
    return check_launch("upload_words");
}

extern "C" int dldkd_segment_mean_l2norm_f32(const float* frames, const int32_t* seg_start, const int32_t* seg_end, float* out,
                                             long n_rows, int D, float eps, void* stream) {
    if (n_rows < 0 || D < 4 || (D & 3)) { set_error("segment_mean_l2norm: bad sizes (D must be a multiple of 4)"); return DLDKD_EINVAL; }
    if (n_rows == 0) return DLDKD_OK;
    if (!frames || !seg_start || !seg_end || !out) { set_error("segment_mean_l2norm: null pointer"); return DLDKD_EINVAL; }
    DLDKD_LAUNCH(segment_mean_l2norm_kernel, dim3((unsigned)((n_rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, frames,
                       seg_start, seg_end, out, n_rows, D, eps);
    return check_launch("segment_mean_l2norm");
}
