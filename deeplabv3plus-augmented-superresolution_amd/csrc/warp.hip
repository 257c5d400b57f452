// Affine / projective warps of NHWC float32 images (gfx950).
//   asr_warp_affine_f32    : one ImageProjectiveTransformV3 pass (BILINEAR, zero fill)
//   asr_augment_copies_f32 : translate(rotate(tile(image))) fused, never materialising the
//                            tiled or rotated stacks (augmentation_utils.py:11-27)
// HBM-bound gather kernels: the shared source image stays cache resident, each output
// element is written exactly once with lane-contiguous stores.
#include "asr_warp_device.h"

namespace {

constexpr int kThreads = 256;

// ---- generic single-stage warp ----------------------------------------------------------
// One thread per output pixel, all C channels (runtime C).  NEAREST: ImageProjectiveTransformV3 with interpolation
// "NEAREST" (the label maps of check_robustness.py:45-50): I(round(in_y), round(in_x)), std::round, 0 outside.
template <bool NEAREST>
__global__ __launch_bounds__(kThreads) void warp_affine_kernel(
    const float* __restrict__ src, float* __restrict__ dst, const float* __restrict__ tfs,
    int n, int src_batched, int tf_batched, int h_in, int w_in, int h_out, int w_out, int c) {
    const int64_t total = (int64_t)n * h_out * w_out;
    for (int64_t p = (int64_t)blockIdx.x * kThreads + threadIdx.x; p < total;
         p += (int64_t)gridDim.x * kThreads) {
        const int x = (int)(p % w_out);
        const int y = (int)((p / w_out) % h_out);
        const int b = (int)(p / ((int64_t)w_out * h_out));
        const AsrTf8 t = asr_load_tf(tfs + (tf_batched ? (int64_t)b * 8 : 0));
        const float* img = src + (src_batched ? (int64_t)b * h_in * w_in * c : 0);
        float ix, iy;
        const bool ok = asr_tf_map(t, (float)x, (float)y, ix, iy);
        float* o = dst + p * c;
        if (NEAREST) {
            const float ry = roundf(iy), rx = roundf(ix);
            const bool in = ok && ry >= 0.0f && ry < (float)h_in && rx >= 0.0f && rx < (float)w_in;
            const float* q = img + ((int64_t)(in ? (int)ry : 0) * w_in + (in ? (int)rx : 0)) * c;
            for (int ch = 0; ch < c; ++ch) o[ch] = in ? q[ch] : 0.0f;
            continue;
        }
        for (int ch = 0; ch < c; ++ch) {
            auto rd = [&](int yy, int xx) -> float {
                return (yy >= 0 && yy < h_in && xx >= 0 && xx < w_in)
                           ? img[((int64_t)yy * w_in + xx) * c + ch]
                           : 0.0f;
            };
            o[ch] = ok ? asr_tf_bilinear(rd, ix, iy) : 0.0f;
        }
    }
}

// ---- fused rotate -> translate from one shared source -----------------------------------
// out[n,y,x,:] = bilinear_T( R_n ) where R_n(yr,xr,:) = bilinear_R(image) for in-bounds
// integer (yr,xr) and 0 outside: exactly the two sequential resamplings of the reference.
// All C channels of a tap travel together (one 12-byte load per tap for RGB instead of three 4-byte loads; the
// coordinate arithmetic is shared): per channel the operations and their order are those of asr_tf_bilinear.
template <int C>
struct PixC {
    float v[C];
};
typedef float asr_f3_a4 __attribute__((ext_vector_type(3), aligned(4)));   // an RGB pixel: 12 bytes, 4-byte aligned

template <int C, class Read>
__device__ __forceinline__ PixC<C> bilinear_c(Read rd, float ix, float iy) {
    const float xf = floorf(ix), yf = floorf(iy);
    const float xc = xf + 1.0f, yc = yf + 1.0f;
    const int x0 = asr_coord_to_int(xf), y0 = asr_coord_to_int(yf);
    const PixC<C> v00 = rd(y0, x0), v01 = rd(y0, x0 + 1);
    const PixC<C> v10 = rd(y0 + 1, x0), v11 = rd(y0 + 1, x0 + 1);
    const float wxl = xc - ix, wxh = ix - xf;
    PixC<C> o;
#pragma unroll
    for (int ch = 0; ch < C; ++ch) {
        const float vyf = wxl * v00.v[ch] + wxh * v01.v[ch];
        const float vyc = wxl * v10.v[ch] + wxh * v11.v[ch];
        o.v[ch] = (yc - iy) * vyf + (iy - yf) * vyc;
    }
    return o;
}

template <int C>
__global__ __launch_bounds__(kThreads) void augment_copies_kernel(
    const float* __restrict__ image, float* __restrict__ copies,
    const float* __restrict__ rot_tf, const float* __restrict__ trans_tf, int n, int h, int w) {
    // grid = (ceil(w / kThreads), h, n): row and copy come from the block index -- no 64-bit divisions per pixel, and
    // the copy's two transforms are block-uniform (scalar loads)
    PixC<C> zero;
#pragma unroll
    for (int ch = 0; ch < C; ++ch) zero.v[ch] = 0.0f;
    const int x = blockIdx.x * kThreads + threadIdx.x;
    const int y = blockIdx.y;
    const int b = blockIdx.z;
    if (x < w) {
        const int64_t p = ((int64_t)b * h + y) * w + x;
        const AsrTf8 tr = asr_load_tf(rot_tf + (int64_t)b * 8);
        const AsrTf8 tt = asr_load_tf(trans_tf + (int64_t)b * 8);
        float ix, iy;
        const bool ok = asr_tf_map(tt, (float)x, (float)y, ix, iy);
        auto rd_img = [&](int yy, int xx) -> PixC<C> {
            const bool in = (yy >= 0 && yy < h && xx >= 0 && xx < w);
            const float* q = image + ((int64_t)(in ? yy : 0) * w + (in ? xx : 0)) * C;   // clamped address
            PixC<C> r;
            if (C == 3) {
                const asr_f3_a4 t = *reinterpret_cast<const asr_f3_a4*>(q);
                r.v[0] = in ? t.x : 0.0f; r.v[1 % C] = in ? t.y : 0.0f; r.v[2 % C] = in ? t.z : 0.0f;
            } else {
#pragma unroll
                for (int ch = 0; ch < C; ++ch) {
                    const float t = q[ch];
                    r.v[ch] = in ? t : 0.0f;
                }
            }
            return r;
        };
        auto rd_rot = [&](int yr, int xr) -> PixC<C> {       // branch-free: all 16 taps of a pixel can be in flight together
            const bool inr = (yr >= 0 && yr < h && xr >= 0 && xr < w);
            float rx, ry;
            const bool okr = asr_tf_map(tr, (float)xr, (float)yr, rx, ry);
            const PixC<C> v = bilinear_c<C>(rd_img, rx, ry);
            PixC<C> r;
#pragma unroll
            for (int ch = 0; ch < C; ++ch) r.v[ch] = (inr && okr) ? v.v[ch] : 0.0f;
            return r;
        };
        const PixC<C> out = ok ? bilinear_c<C>(rd_rot, ix, iy) : zero;
        float* o = copies + p * C;
#pragma unroll
        for (int ch = 0; ch < C; ++ch) o[ch] = out.v[ch];
    }
}

int grid_for(int64_t total) {
    int64_t g = asr_cdiv(total, kThreads);
    const int64_t cap = 256 * 16;  // 256 CUs x 16 resident 256-thread blocks, grid-stride beyond
    return (int)(g < cap ? g : cap);
}

}  // namespace

extern "C" int asr_warp_affine_f32(const float* src, float* dst, const float* transforms, int n,
                                   int src_batched, int tf_batched, int h_in, int w_in, int h_out,
                                   int w_out, int c, asr_stream_t stream) {
    ASR_REQUIRE(src && dst && transforms, "asr_warp_affine_f32: null pointer");
    ASR_REQUIRE(n > 0 && h_in > 0 && w_in > 0 && h_out > 0 && w_out > 0 && c > 0,
                "asr_warp_affine_f32: bad shape n=%d in=%dx%d out=%dx%d c=%d", n, h_in, w_in, h_out,
                w_out, c);
    const int64_t total = (int64_t)n * h_out * w_out;
    hipLaunchKernelGGL(warp_affine_kernel<false>, dim3(grid_for(total)), dim3(kThreads), 0, asr_stream(stream),
                       src, dst, transforms, n, src_batched, tf_batched, h_in, w_in, h_out, w_out, c);
    ASR_LAUNCH_CHECK();
    return ASR_OK;
}

extern "C" int asr_warp_affine_nearest_f32(const float* src, float* dst, const float* transforms, int n,
                                           int src_batched, int tf_batched, int h_in, int w_in, int h_out,
                                           int w_out, int c, asr_stream_t stream) {
    ASR_REQUIRE(src && dst && transforms, "asr_warp_affine_nearest_f32: null pointer");
    ASR_REQUIRE(n > 0 && h_in > 0 && w_in > 0 && h_out > 0 && w_out > 0 && c > 0,
                "asr_warp_affine_nearest_f32: bad shape n=%d in=%dx%d out=%dx%d c=%d", n, h_in, w_in, h_out,
                w_out, c);
    const int64_t total = (int64_t)n * h_out * w_out;
    hipLaunchKernelGGL(warp_affine_kernel<true>, dim3(grid_for(total)), dim3(kThreads), 0, asr_stream(stream),
                       src, dst, transforms, n, src_batched, tf_batched, h_in, w_in, h_out, w_out, c);
    ASR_LAUNCH_CHECK();
    return ASR_OK;
}

extern "C" int asr_augment_copies_f32(const float* image, float* copies, const float* rot_tf,
                                      const float* trans_tf, int n, int h, int w, int c,
                                      asr_stream_t stream) {
    ASR_REQUIRE(image && copies && rot_tf && trans_tf, "asr_augment_copies_f32: null pointer");
    ASR_REQUIRE(n > 0 && h > 0 && w > 0, "asr_augment_copies_f32: bad shape n=%d %dx%d", n, h, w);
    ASR_UNSUPPORTED(n > 65535 || h > 65535, "asr_augment_copies_f32: n and h must not exceed 65535 (grid dimensions)");
    const dim3 grid((unsigned)((w + kThreads - 1) / kThreads), (unsigned)h, (unsigned)n), block(kThreads);
    hipStream_t s = asr_stream(stream);
    switch (c) {
        case 1: hipLaunchKernelGGL(augment_copies_kernel<1>, grid, block, 0, s, image, copies, rot_tf, trans_tf, n, h, w); break;
        case 3: hipLaunchKernelGGL(augment_copies_kernel<3>, grid, block, 0, s, image, copies, rot_tf, trans_tf, n, h, w); break;
        default:
            asr_set_error("asr_augment_copies_f32: channels must be 1 or 3, got %d", c);
            return ASR_ERR_UNSUPPORTED;
    }
    ASR_LAUNCH_CHECK();
    return ASR_OK;
}
