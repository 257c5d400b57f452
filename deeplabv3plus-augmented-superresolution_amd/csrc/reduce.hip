// Output-processing (OPM), thresholding and IoU counting kernels (gfx950).
//
// Reference (paths in /root/reference):
//   utils.py:115-119                                    create_mask (argmax)
//   superresolution_scripts/augmentation_utils.py:80-115  OPM modes argmax / slice / slice_max
//   superresolution_scripts/superres_utils.py:56-62,118-139  min_max_normalization, threshold_image
//   utils.py:180-204                                    single_class_IOU (integer counts)
// All of these are HBM-bound single passes; per-segment reductions use wave shuffles and one
// block per segment so the results are deterministic.
#include "asr_common.h"

namespace {

// ---- per-segment min / max ------------------------------------------------------------------
// A segment is split over several workgroups (a single 1024-thread workgroup per segment left 255 CUs idle on the
// one-segment calls of the hot path: 190 us for 1.6 M floats); partial results meet in out[] through integer atomics
// on the float bit patterns (min / max are order-independent, so the result is exact).
__device__ __forceinline__ void atomic_min_float(float* addr, float v) {
    v += 0.0f;                                     // -0.0 -> +0.0: the sign test below must agree with the bit pattern
    if (v >= 0.0f) atomicMin(reinterpret_cast<int*>(addr), __float_as_int(v));
    else atomicMax(reinterpret_cast<unsigned int*>(addr), __float_as_uint(v));
}
__device__ __forceinline__ void atomic_max_float(float* addr, float v) {
    v += 0.0f;
    if (v >= 0.0f) atomicMax(reinterpret_cast<int*>(addr), __float_as_int(v));
    else atomicMin(reinterpret_cast<unsigned int*>(addr), __float_as_uint(v));
}

__global__ __launch_bounds__(256) void minmax_init_kernel(float* __restrict__ out, int segments) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < segments) { out[2 * i] = INFINITY; out[2 * i + 1] = -INFINITY; }
}

__global__ __launch_bounds__(1024) void minmax_kernel(const float* __restrict__ x, float* __restrict__ out,
                                                      int64_t per_seg) {
    __shared__ float smin[16], smax[16];
    const float* p = x + (int64_t)blockIdx.y * per_seg;
    float mn = INFINITY, mx = -INFINITY;
    for (int64_t i = (int64_t)blockIdx.x * 1024 + threadIdx.x; i < per_seg; i += (int64_t)gridDim.x * 1024) {
        const float v = p[i];
        mn = fminf(mn, v);
        mx = fmaxf(mx, v);
    }
    mn = asr_wave_min(mn);
    mx = asr_wave_max(mx);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0) { smin[wave] = mn; smax[wave] = mx; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int i = 1; i < 16; ++i) { mn = fminf(mn, smin[i]); mx = fmaxf(mx, smax[i]); }
        atomic_min_float(out + blockIdx.y * 2 + 0, mn);
        atomic_max_float(out + blockIdx.y * 2 + 1, mx);
    }
}

// ---- argmax over the class axis (first maximum wins, like tf.argmax) --------------------------
__device__ __forceinline__ int argmax_row(const float* __restrict__ row, int classes) {
    float best = row[0];
    int arg = 0;
    for (int c = 1; c < classes; ++c) {
        const float v = row[c];
        if (v > best) { best = v; arg = c; }
    }
    return arg;
}

// The class rows are 4 * classes bytes apart (84 for 21 classes): a thread walking its own row makes every wave-load a
// 64-way strided gather that pulls whole cache lines for 4 bytes each (PMC: 7.3x the logits' bytes fetched).  A
// workgroup therefore copies its 256 consecutive rows into LDS with coalesced loads and the threads walk the LDS rows
// (stride of `classes` words: conflict-free for odd class counts such as 21).  kMaxStageClasses bounds the LDS tile.
constexpr int kMaxStageClasses = 32;

__device__ __forceinline__ const float* stage_rows(const float* __restrict__ logits, int64_t first, int64_t pixels, int classes,
                                                   float* __restrict__ tile) {
    const int64_t rows = min((int64_t)256, pixels - first);
    const int64_t n = rows * classes;
    const float* src = logits + first * classes;
    for (int64_t i = threadIdx.x; i < n; i += 256) tile[i] = src[i];
    __syncthreads();
    return tile + (int64_t)threadIdx.x * classes;
}

__global__ __launch_bounds__(256) void argmax_kernel(const float* __restrict__ logits, int32_t* __restrict__ out,
                                                     int64_t pixels, int classes) {
    __shared__ float tile[256 * kMaxStageClasses];
    for (int64_t first = (int64_t)blockIdx.x * 256; first < pixels; first += (int64_t)gridDim.x * 256) {
        const float* row = stage_rows(logits, first, pixels, classes, tile);
        const int64_t p = first + threadIdx.x;
        if (p < pixels) out[p] = argmax_row(row, classes);
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void opm_argmax_kernel(const float* __restrict__ logits, float* __restrict__ out,
                                                         int64_t pixels, int classes, int class_id) {
    __shared__ float tile[256 * kMaxStageClasses];
    for (int64_t first = (int64_t)blockIdx.x * 256; first < pixels; first += (int64_t)gridDim.x * 256) {
        const float* row = stage_rows(logits, first, pixels, classes, tile);
        const int64_t p = first + threadIdx.x;
        if (p < pixels) out[p] = (argmax_row(row, classes) == class_id) ? (float)class_id : 0.0f;
        __syncthreads();
    }
}

// same walks straight from global memory, for class counts beyond the LDS tile
__global__ __launch_bounds__(256) void argmax_direct_kernel(const float* __restrict__ logits, int32_t* __restrict__ out,
                                                            int64_t pixels, int classes) {
    for (int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x; p < pixels; p += (int64_t)gridDim.x * 256)
        out[p] = argmax_row(logits + p * classes, classes);
}

__global__ __launch_bounds__(256) void opm_argmax_direct_kernel(const float* __restrict__ logits, float* __restrict__ out,
                                                                int64_t pixels, int classes, int class_id) {
    for (int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x; p < pixels; p += (int64_t)gridDim.x * 256)
        out[p] = (argmax_row(logits + p * classes, classes) == class_id) ? (float)class_id : 0.0f;
}

__global__ __launch_bounds__(256) void opm_slice_max_kernel(const float* __restrict__ logits, float* __restrict__ cls,
                                                            float* __restrict__ mx, int64_t pixels, int classes,
                                                            int class_id) {
    for (int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x; p < pixels; p += (int64_t)gridDim.x * 256) {
        const float* row = logits + p * classes;
        float best = -INFINITY;
        for (int c = 0; c < classes; ++c)
            if (c != class_id) best = fmaxf(best, row[c]);
        cls[p] = row[class_id];
        mx[p] = best;
    }
}

// slice OPM: class logit min-max normalised by the per-copy global min/max (seg_minmax[copy])
__global__ __launch_bounds__(256) void opm_slice_kernel(const float* __restrict__ logits, float* __restrict__ out,
                                                        const float* __restrict__ seg_minmax, int64_t pixels_per_copy,
                                                        int classes, int class_id, float new_min, float new_max) {
    const int copy = blockIdx.y;
    const float mn = seg_minmax[copy * 2 + 0], mxv = seg_minmax[copy * 2 + 1];
    const float den = ((mxv - mn) != 0.0f) ? (mxv - mn) : 1.0f;
    const float span = new_max - new_min;
    const float* base = logits + (int64_t)copy * pixels_per_copy * classes;
    float* o = out + (int64_t)copy * pixels_per_copy;
    for (int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x; p < pixels_per_copy; p += (int64_t)gridDim.x * 256) {
        const float num = (base[p * classes + class_id] - mn) * span;
        o[p] = new_min + num / den;
    }
}

// ---- min_max_normalization of a stack with its own global minimum / maximum (load_SR_data, superres_utils.py:183-206) ----
__global__ __launch_bounds__(256) void minmax_normalize_kernel(const float* __restrict__ x, float* __restrict__ out,
                                                               const float* __restrict__ seg_minmax, int64_t per_seg,
                                                               float new_min, float new_max) {
    const int seg = blockIdx.y;
    const float mn = seg_minmax[seg * 2 + 0], mxv = seg_minmax[seg * 2 + 1];
    const float den = ((mxv - mn) != 0.0f) ? (mxv - mn) : 1.0f;
    const float span = new_max - new_min;
    const float* p = x + (int64_t)seg * per_seg;
    float* o = out + (int64_t)seg * per_seg;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < per_seg; i += (int64_t)gridDim.x * 256)
        o[i] = new_min + ((p[i] - mn) * span) / den;
}

// ---- threshold_image -----------------------------------------------------------------------------
__global__ __launch_bounds__(256) void threshold_kernel(const float* __restrict__ img, const float* __restrict__ th_mask,
                                                        const float* __restrict__ seg_minmax, int32_t* __restrict__ out,
                                                        int64_t per_seg, float th_factor, int th_value) {
    const int seg = blockIdx.y;
    const float* p = img + (int64_t)seg * per_seg;
    int32_t* o = out + (int64_t)seg * per_seg;
    if (th_mask) {
        const float* t = th_mask + (int64_t)seg * per_seg;
        for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < per_seg; i += (int64_t)gridDim.x * 256)
            o[i] = (p[i] >= t[i]) ? th_value : 0;
    } else {
        const float th = seg_minmax[seg * 2 + 1] * th_factor;
        for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < per_seg; i += (int64_t)gridDim.x * 256)
            o[i] = (p[i] > th) ? th_value : 0;
    }
}

// ---- IoU counts: counts[seg] = {inter_c, union_c, inter_bg, union_bg} ------------------------------
__global__ __launch_bounds__(256) void iou_counts_kernel(const int32_t* __restrict__ truth, const int32_t* __restrict__ pred,
                                                         unsigned long long* __restrict__ counts, int64_t per_seg,
                                                         int64_t truth_stride, int class_id, int include_bg) {
    const int seg = blockIdx.y;
    const int32_t* t = truth + (int64_t)seg * truth_stride;          // truth_stride 0: every segment against one label map
    const int32_t* q = pred + (int64_t)seg * per_seg;
    unsigned int ic = 0, uc = 0, ib = 0, ub = 0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < per_seg; i += (int64_t)gridDim.x * 256) {
        int tv = t[i];
        const int pv = q[i];
        if (include_bg && tv != class_id) tv = 0;  // utils.py:188-190
        const bool tc = tv == class_id, pc = pv == class_id;
        ic += (tc && pc); uc += (tc || pc);
        const bool tb = tv == 0, pb = pv == 0;
        ib += (tb && pb); ub += (tb || pb);
    }
    auto wsum = [](unsigned int v) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
        return v;
    };
    ic = wsum(ic); uc = wsum(uc); ib = wsum(ib); ub = wsum(ub);
    // one set of 64-bit atomics per workgroup (per-wave atomics on four shared addresses serialised the launch)
    __shared__ unsigned int part[4][4];
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { part[wave][0] = ic; part[wave][1] = uc; part[wave][2] = ib; part[wave][3] = ub; }
    __syncthreads();
    if (threadIdx.x < 4) {
        const unsigned long long v = (unsigned long long)part[0][threadIdx.x] + part[1][threadIdx.x] + part[2][threadIdx.x] +
                                     part[3][threadIdx.x];
        if (v) atomicAdd(counts + seg * 4 + threadIdx.x, v);
    }
}

// ---- per-label counts for Mean_IOU (utils.py:151-177) -----------------------------------------------------------
// counts[seg][0][l] = |truth == l|, [1][l] = |pred == l|, [2][l] = |truth == l & pred == l| for labels 0..255
// (union = [0] + [1] - [2]); labels outside 0..255 are not counted.
__global__ __launch_bounds__(256) void class_counts_kernel(const int32_t* __restrict__ truth, const int32_t* __restrict__ pred,
                                                           unsigned long long* __restrict__ counts, int64_t per_seg) {
    __shared__ unsigned int hist[3 * 256];
    const int seg = blockIdx.y;
    for (int i = threadIdx.x; i < 3 * 256; i += 256) hist[i] = 0;
    __syncthreads();
    const int32_t* t = truth + (int64_t)seg * per_seg;
    const int32_t* q = pred + (int64_t)seg * per_seg;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < per_seg; i += (int64_t)gridDim.x * 256) {
        const int tv = t[i], pv = q[i];
        if (tv >= 0 && tv < 256) atomicAdd(hist + tv, 1u);
        if (pv >= 0 && pv < 256) atomicAdd(hist + 256 + pv, 1u);
        if (tv == pv && tv >= 0 && tv < 256) atomicAdd(hist + 512 + tv, 1u);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 3 * 256; i += 256)
        if (hist[i]) atomicAdd(counts + (int64_t)seg * 768 + i, (unsigned long long)hist[i]);
}

// ---- last_activation: softmax / sigmoid over the class axis (model.py:124-125) ----------------------
__device__ __forceinline__ void activate_row(const float* row, float* o, int classes, int kind) {     // o may be row
    if (kind == 1) {                       // softmax: exp(x - max) / sum
        float mx = row[0];
        for (int c = 1; c < classes; ++c) mx = fmaxf(mx, row[c]);
        float sum = 0.f;
        for (int c = 0; c < classes; ++c) sum += expf(row[c] - mx);
        for (int c = 0; c < classes; ++c) o[c] = expf(row[c] - mx) / sum;
    } else {                               // sigmoid
        for (int c = 0; c < classes; ++c) o[c] = 1.0f / (1.0f + expf(-row[c]));
    }
}

// Rows of `classes` floats (84 bytes for 21 classes) walked by one thread each are 64-way strided gathers and scatters
// (measured 0.44 TB/s on the [100,128,128,21] logits of BASELINE configs[2]); like argmax_kernel, a workgroup moves its
// 256 rows through LDS with coalesced loads and stores and the threads work on the LDS rows, in place.
__global__ __launch_bounds__(256) void class_activation_kernel(const float* __restrict__ logits, float* __restrict__ out,
                                                              int64_t pixels, int classes, int kind) {
    __shared__ float tile[256 * kMaxStageClasses];
    for (int64_t first = (int64_t)blockIdx.x * 256; first < pixels; first += (int64_t)gridDim.x * 256) {
        const int64_t rows = min((int64_t)256, pixels - first);
        const int64_t n = rows * classes;
        const float* src = logits + first * classes;
        for (int64_t i = threadIdx.x; i < n; i += 256) tile[i] = src[i];
        __syncthreads();
        if ((int64_t)threadIdx.x < rows) {
            float* row = tile + (int64_t)threadIdx.x * classes;
            activate_row(row, row, classes, kind);            // every o[c] is written after the last read of row[c'] it needs:
        }                                                     // softmax re-reads row[c] right before overwriting it
        __syncthreads();
        float* dst = out + first * classes;
        for (int64_t i = threadIdx.x; i < n; i += 256) dst[i] = tile[i];
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void class_activation_direct_kernel(const float* __restrict__ logits, float* __restrict__ out,
                                                                     int64_t pixels, int classes, int kind) {
    for (int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x; p < pixels; p += (int64_t)gridDim.x * 256)
        activate_row(logits + p * classes, out + p * classes, classes, kind);
}

int stream_grid(int64_t n) {
    int64_t g = asr_cdiv(n, 256);
    return (int)(g < 2048 ? (g > 0 ? g : 1) : 2048);
}

}  // namespace

extern "C" int asr_minmax_f32(const float* x, float* out_minmax, int64_t per_segment, int segments, asr_stream_t stream) {
    ASR_REQUIRE(x && out_minmax, "asr_minmax_f32: null pointer");
    ASR_REQUIRE(per_segment > 0 && segments > 0, "asr_minmax_f32: empty input (per_segment=%lld segments=%d)",
                (long long)per_segment, segments);
    ASR_REQUIRE(segments <= 65535, "asr_minmax_f32: more than 65535 segments");
    hipLaunchKernelGGL(minmax_init_kernel, dim3((unsigned)asr_cdiv(segments, 256)), dim3(256), 0, asr_stream(stream), out_minmax, segments);
    ASR_LAUNCH_CHECK();
    // ~16 K elements per workgroup, at most ~1024 workgroups in the launch
    long long per_wg = asr_cdiv(per_segment, 16384);
    const long long cap = segments >= 1024 ? 1 : 1024 / segments;
    if (per_wg > cap) per_wg = cap;
    hipLaunchKernelGGL(minmax_kernel, dim3((unsigned)per_wg, (unsigned)segments), dim3(1024), 0, asr_stream(stream), x, out_minmax,
                       per_segment);
    ASR_LAUNCH_CHECK();
    return ASR_OK;
}

extern "C" int asr_argmax_i32(const float* logits, int32_t* out, int64_t pixels, int classes, asr_stream_t stream) {
    ASR_REQUIRE(logits && out, "asr_argmax_i32: null pointer");
    ASR_REQUIRE(pixels > 0 && classes > 0, "asr_argmax_i32: bad shape");
    if (classes <= kMaxStageClasses)
        hipLaunchKernelGGL(argmax_kernel, dim3(stream_grid(pixels)), dim3(256), 0, asr_stream(stream), logits, out, pixels, classes);
    else
        hipLaunchKernelGGL(argmax_direct_kernel, dim3(stream_grid(pixels)), dim3(256), 0, asr_stream(stream), logits, out, pixels, classes);
    ASR_LAUNCH_CHECK();
    return ASR_OK;
}

extern "C" int asr_opm_argmax_f32(const float* logits, float* class_mask, int64_t pixels, int classes, int class_id,
                                  asr_stream_t stream) {
    ASR_REQUIRE(logits && class_mask, "asr_opm_argmax_f32: null pointer");
    ASR_REQUIRE(pixels > 0 && classes > 0 && class_id >= 0 && class_id < classes, "asr_opm_argmax_f32: bad shape/class");
    if (classes <= kMaxStageClasses)
        hipLaunchKernelGGL(opm_argmax_kernel, dim3(stream_grid(pixels)), dim3(256), 0, asr_stream(stream), logits, class_mask,
                           pixels, classes, class_id);
    else
        hipLaunchKernelGGL(opm_argmax_direct_kernel, dim3(stream_grid(pixels)), dim3(256), 0, asr_stream(stream), logits,
                           class_mask, pixels, classes, class_id);
    ASR_LAUNCH_CHECK();
    return ASR_OK;
}

extern "C" int asr_opm_slice_max_f32(const float* logits, float* class_mask, float* max_mask, int64_t pixels, int classes,
                                     int class_id, asr_stream_t stream) {
    ASR_REQUIRE(logits && class_mask && max_mask, "asr_opm_slice_max_f32: null pointer");
    ASR_REQUIRE(pixels > 0 && classes > 1 && class_id >= 0 && class_id < classes, "asr_opm_slice_max_f32: bad shape/class");
    hipLaunchKernelGGL(opm_slice_max_kernel, dim3(stream_grid(pixels)), dim3(256), 0, asr_stream(stream), logits,
                       class_mask, max_mask, pixels, classes, class_id);
    ASR_LAUNCH_CHECK();
    return ASR_OK;
}

extern "C" int asr_opm_slice_f32(const float* logits, float* class_mask, float* minmax_ws, int copies,
                                 int64_t pixels_per_copy, int classes, int class_id, float new_min, float new_max,
                                 asr_stream_t stream) {
    ASR_REQUIRE(logits && class_mask && minmax_ws, "asr_opm_slice_f32: null pointer");
    ASR_REQUIRE(copies > 0 && copies <= 65535 && pixels_per_copy > 0 && classes > 0 && class_id >= 0 && class_id < classes,
                "asr_opm_slice_f32: bad shape/class");
    int rc = asr_minmax_f32(logits, minmax_ws, pixels_per_copy * classes, copies, stream);
    if (rc != ASR_OK) return rc;
    hipLaunchKernelGGL(opm_slice_kernel, dim3(stream_grid(pixels_per_copy), copies), dim3(256), 0, asr_stream(stream),
                       logits, class_mask, minmax_ws, pixels_per_copy, classes, class_id, new_min, new_max);
    ASR_LAUNCH_CHECK();
    return ASR_OK;
}

extern "C" int asr_threshold_f32(const float* image, const float* th_mask, float* minmax_ws, int32_t* out,
                                 int64_t per_segment, int segments, float th_factor, int th_value,
                                 asr_stream_t stream) {
    ASR_REQUIRE(image && out, "asr_threshold_f32: null pointer");
    ASR_REQUIRE(per_segment > 0 && segments > 0 && segments <= 65535, "asr_threshold_f32: bad shape");
    if (!th_mask) {
        ASR_REQUIRE(minmax_ws, "asr_threshold_f32: minmax workspace required without th_mask");
        int rc = asr_minmax_f32(image, minmax_ws, per_segment, segments, stream);
        if (rc != ASR_OK) return rc;
    }
    hipLaunchKernelGGL(threshold_kernel, dim3(stream_grid(per_segment), segments), dim3(256), 0, asr_stream(stream),
                       image, th_mask, minmax_ws, out, per_segment, th_factor, th_value);
    ASR_LAUNCH_CHECK();
    return ASR_OK;
}

static int iou_counts_common(const char* fn, const int32_t* truth, const int32_t* pred, int64_t* counts, int64_t per_segment,
                             int64_t truth_stride, int segments, int class_id, int include_bg, asr_stream_t stream) {
    ASR_REQUIRE(truth && pred && counts, "%s: null pointer", fn);
    ASR_REQUIRE(per_segment > 0 && segments > 0 && segments <= 65535, "%s: bad shape", fn);
    hipStream_t s = asr_stream(stream);
    ASR_HIP_CHECK(hipMemsetAsync(counts, 0, sizeof(int64_t) * 4 * segments, s));
    hipLaunchKernelGGL(iou_counts_kernel, dim3(stream_grid(per_segment) > 64 ? 64 : stream_grid(per_segment), segments),
                       dim3(256), 0, s, truth, pred, reinterpret_cast<unsigned long long*>(counts), per_segment, truth_stride,
                       class_id, include_bg);
    ASR_LAUNCH_CHECK();
    return ASR_OK;
}

extern "C" int asr_iou_counts_i32(const int32_t* truth, const int32_t* pred, int64_t* counts, int64_t per_segment,
                                  int segments, int class_id, int include_bg, asr_stream_t stream) {
    return iou_counts_common("asr_iou_counts_i32", truth, pred, counts, per_segment, per_segment, segments, class_id, include_bg,
                             stream);
}

extern "C" int asr_iou_counts_shared_truth_i32(const int32_t* truth, const int32_t* preds, int64_t* counts, int64_t pixels,
                                               int num_preds, int class_id, int include_bg, asr_stream_t stream) {
    return iou_counts_common("asr_iou_counts_shared_truth_i32", truth, preds, counts, pixels, 0, num_preds, class_id,
                             include_bg, stream);
}

extern "C" int asr_minmax_normalize_f32(const float* x, float* out, float* minmax_ws, int64_t per_segment, int segments,
                                        float new_min, float new_max, asr_stream_t stream) {
    ASR_REQUIRE(x && out && minmax_ws, "asr_minmax_normalize_f32: null pointer");
    int rc = asr_minmax_f32(x, minmax_ws, per_segment, segments, stream);
    if (rc != ASR_OK) return rc;
    hipLaunchKernelGGL(minmax_normalize_kernel, dim3(stream_grid(per_segment), segments), dim3(256), 0, asr_stream(stream), x, out,
                       minmax_ws, per_segment, new_min, new_max);
    ASR_LAUNCH_CHECK();
    return ASR_OK;
}

extern "C" int asr_class_counts_i32(const int32_t* truth, const int32_t* pred, int64_t* counts, int64_t per_segment,
                                    int segments, asr_stream_t stream) {
    ASR_REQUIRE(truth && pred && counts, "asr_class_counts_i32: null pointer");
    ASR_REQUIRE(per_segment > 0 && segments > 0 && segments <= 65535, "asr_class_counts_i32: bad shape");
    hipStream_t s = asr_stream(stream);
    ASR_HIP_CHECK(hipMemsetAsync(counts, 0, sizeof(int64_t) * 768 * (size_t)segments, s));
    const int grid = stream_grid(per_segment) > 64 ? 64 : stream_grid(per_segment);
    hipLaunchKernelGGL(class_counts_kernel, dim3(grid, segments), dim3(256), 0, s, truth, pred,
                       reinterpret_cast<unsigned long long*>(counts), per_segment);
    ASR_LAUNCH_CHECK();
    return ASR_OK;
}

extern "C" int asr_class_activation_f32(const float* logits, float* out, int64_t pixels, int classes, int kind,
                                        asr_stream_t stream) {
    ASR_REQUIRE(logits && out, "asr_class_activation_f32: null pointer");
    ASR_REQUIRE(pixels > 0 && classes > 0 && (kind == 1 || kind == 2), "asr_class_activation_f32: bad arguments (kind 1 = softmax, 2 = sigmoid)");
    if (classes <= kMaxStageClasses)
        hipLaunchKernelGGL(class_activation_kernel, dim3(stream_grid(pixels)), dim3(256), 0, asr_stream(stream), logits, out, pixels,
                           classes, kind);
    else
        hipLaunchKernelGGL(class_activation_direct_kernel, dim3(stream_grid(pixels)), dim3(256), 0, asr_stream(stream), logits, out,
                           pixels, classes, kind);
    ASR_LAUNCH_CHECK();
    return ASR_OK;
}
