// Depthwise 3x3 convolution, NHWC float32, with the BatchNorm that follows it folded into the
// weights/bias and the ReLUs on either side fused (gfx950).
//
// Reference: model.py:463-508 (_SepConv_BN): [ZeroPadding2D if stride != 1] -> [ReLU if not
// depth_activation] -> DepthwiseConv2D(3x3, stride, dilation) -> BN -> [ReLU if depth_activation].
//
// HBM-bound (2.1 flop/byte).  Channels are the fastest axis, so a lane owns 4 consecutive
// channels (one 16-byte access) and neighbouring lanes neighbouring channels: every global
// access of a wave covers whole 256-byte pixel rows.
//  * dw_tiled_kernel (stride 1, rate 1 or 2 -- 90 % of the depthwise bytes): a 64-channel x
//    (8+2R)x(16+2R) input tile is staged once through LDS (pre-ReLU applied on the way in);
//    each lane then reads its 9 taps with conflict-free ds_read_b128 and writes 8 output rows.
//  * dw_direct_kernel: any stride / rate (stride-2 block ends, ASPP rates 6/12/18 whose halo
//    exceeds the 32x32 map), taps straight from L1/L2.
#include "asr_common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

struct DwArgs {
    const float* x;
    const float* w;     // [3][3][C], BN scale folded
    const float* bias;  // [C]
    float* y;
    int batch, h_in, w_in, c, h_out, w_out;
    int stride, rate, pad_top, pad_left;
    int pre_relu, post_relu;
    int ldx, ldy;  // channel strides of one pixel (>= c)
};

__device__ __forceinline__ f32x4 relu4(f32x4 v) {
    v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
    return v;
}

constexpr int TH = 8, TW = 16, CB = 64;

template <int R>
__global__ __launch_bounds__(256) void dw_tiled_kernel(DwArgs p, int tiles_x) {
    constexpr int ROWS = TH + 2 * R, COLS = TW + 2 * R;
    __shared__ __attribute__((aligned(16))) float tile[ROWS * COLS * CB];
    const int tid = threadIdx.x;
    const int tx = blockIdx.x % tiles_x, ty = blockIdx.x / tiles_x;
    const int cb = blockIdx.y * CB;
    const int b = blockIdx.z;
    const int oy0 = ty * TH, ox0 = tx * TW;
    const int iy0 = oy0 - p.pad_top, ix0 = ox0 - p.pad_left;
    const float* xin = p.x + (long long)b * p.h_in * p.w_in * p.ldx;

    // stage the input tile (zero padded, pre-ReLU applied once)
    for (int idx = tid; idx < ROWS * COLS * (CB / 4); idx += 256) {
        const int c4 = idx & 15, pix = idx >> 4;
        const int pr = pix / COLS, pc = pix - pr * COLS;
        const int iy = iy0 + pr, ix = ix0 + pc, ch = cb + c4 * 4;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (iy >= 0 && iy < p.h_in && ix >= 0 && ix < p.w_in && ch < p.c) {
            v = *reinterpret_cast<const f32x4*>(xin + ((long long)iy * p.w_in + ix) * p.ldx + ch);
            if (p.pre_relu) v = relu4(v);
        }
        *reinterpret_cast<f32x4*>(tile + idx * 4) = v;
    }
    __syncthreads();

    const int c4 = tid & 15, col = tid >> 4;
    const int ch = cb + c4 * 4;
    const int ox = ox0 + col;
    if (ch >= p.c || ox >= p.w_out) return;
    f32x4 wk[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) wk[t] = *reinterpret_cast<const f32x4*>(p.w + (long long)t * p.c + ch);
    const f32x4 bv = *reinterpret_cast<const f32x4*>(p.bias + ch);
    float* yout = p.y + (long long)b * p.h_out * p.w_out * p.ldy;
#pragma unroll
    for (int r = 0; r < TH; ++r) {
        const int oy = oy0 + r;
        if (oy >= p.h_out) break;
        f32x4 acc = bv;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(tile + (((r + ky * R) * COLS + col + kx * R) * (CB / 4) + c4) * 4);
                acc += v * wk[ky * 3 + kx];
            }
        if (p.post_relu) acc = relu4(acc);
        *reinterpret_cast<f32x4*>(yout + ((long long)oy * p.w_out + ox) * p.ldy + ch) = acc;
    }
}

__global__ __launch_bounds__(256) void dw_direct_kernel(DwArgs p) {
    const int c4n = p.c >> 2;
    const long long total = (long long)p.batch * p.h_out * p.w_out * c4n;
    for (long long o = (long long)blockIdx.x * 256 + threadIdx.x; o < total; o += (long long)gridDim.x * 256) {
        const int c4 = (int)(o % c4n);
        long long t = o / c4n;
        const int ox = (int)(t % p.w_out); t /= p.w_out;
        const int oy = (int)(t % p.h_out);
        const long long b = t / p.h_out;
        const int ch = c4 * 4;
        const float* xin = p.x + b * p.h_in * p.w_in * p.ldx;
        f32x4 acc = *reinterpret_cast<const f32x4*>(p.bias + ch);
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int iy = oy * p.stride - p.pad_top + ky * p.rate;
            if (iy < 0 || iy >= p.h_in) continue;
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int ix = ox * p.stride - p.pad_left + kx * p.rate;
                if (ix < 0 || ix >= p.w_in) continue;
                f32x4 v = *reinterpret_cast<const f32x4*>(xin + ((long long)iy * p.w_in + ix) * p.ldx + ch);
                if (p.pre_relu) v = relu4(v);
                acc += v * *reinterpret_cast<const f32x4*>(p.w + (long long)(ky * 3 + kx) * p.c + ch);
            }
        }
        if (p.post_relu) acc = relu4(acc);
        *reinterpret_cast<f32x4*>(p.y + ((b * p.h_out + oy) * p.w_out + ox) * p.ldy + ch) = acc;
    }
}

}  // namespace

extern "C" int asr_dwconv3x3_nhwc_f32(const float* x, const float* w, const float* bias, float* y, int batch, int h_in,
                                      int w_in, int c, int stride, int rate, int pad_top, int pad_left, int h_out,
                                      int w_out, int ldx, int ldy, int pre_relu, int post_relu, int force_direct,
                                      asr_stream_t stream) {
    ASR_REQUIRE(x && w && bias && y, "asr_dwconv3x3_nhwc_f32: null pointer");
    ASR_REQUIRE(batch > 0 && h_in > 0 && w_in > 0 && c > 0 && h_out > 0 && w_out > 0 && stride > 0 && rate > 0 &&
                    pad_top >= 0 && pad_left >= 0,
                "asr_dwconv3x3_nhwc_f32: bad geometry");
    ASR_REQUIRE(ldx >= c && ldy >= c, "asr_dwconv3x3_nhwc_f32: ldx/ldy < c");
    ASR_REQUIRE((long long)(h_out - 1) * stride - pad_top <= h_in - 1 && (long long)(w_out - 1) * stride - pad_left <= w_in - 1,
                "asr_dwconv3x3_nhwc_f32: output %dx%d does not fit input %dx%d (stride %d)", h_out, w_out, h_in, w_in, stride);
    ASR_UNSUPPORTED((c & 3) || (ldx & 3) || (ldy & 3), "asr_dwconv3x3_nhwc_f32: c, ldx, ldy must be multiples of 4");
    ASR_UNSUPPORTED((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(w) |
                     reinterpret_cast<uintptr_t>(bias)) & 15,
                    "asr_dwconv3x3_nhwc_f32: pointers must be 16-byte aligned");
    DwArgs p{x, w, bias, y, batch, h_in, w_in, c, h_out, w_out, stride, rate, pad_top, pad_left, pre_relu, post_relu, ldx, ldy};
    hipStream_t s = asr_stream(stream);
    if (!force_direct && stride == 1 && (rate == 1 || rate == 2) && batch <= 65535) {
        const int tiles_x = (int)asr_cdiv(w_out, TW), tiles_y = (int)asr_cdiv(h_out, TH);
        const dim3 grid(tiles_x * tiles_y, (unsigned)asr_cdiv(c, CB), batch);
        if (rate == 1) hipLaunchKernelGGL(dw_tiled_kernel<1>, grid, dim3(256), 0, s, p, tiles_x);
        else hipLaunchKernelGGL(dw_tiled_kernel<2>, grid, dim3(256), 0, s, p, tiles_x);
    } else {
        const long long total = (long long)batch * h_out * w_out * (c >> 2);
        const long long g = asr_cdiv(total, 256);
        hipLaunchKernelGGL(dw_direct_kernel, dim3((unsigned)(g < 8192 ? g : 8192)), dim3(256), 0, s, p);
    }
    ASR_LAUNCH_CHECK();
    return ASR_OK;
}
