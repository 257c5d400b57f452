// Device-side restatement of ImageProjectiveTransformV3 (BILINEAR, fill CONSTANT 0) sampling.
// Files including this header are compiled with -ffp-contract=off so that every multiply
// and add rounds separately, exactly like the TF CPU kernel (and like oracle/tf_ops.py).
//
// Reference call sites this arithmetic serves (paths in /root/reference):
//   superresolution_scripts/augmentation_utils.py:22-25  (rotate, translate of image copies)
//   superresolution_scripts/superresolution.py:61-64     (forward model inside the SR loss)
//   superresolution_scripts/superresolution.py:142-147   (inverse warps of max/mean SR)
#pragma once
#include "asr_common.h"

// (x, y) = output pixel -> source coordinates, TF ProjectiveGenerator order of operations.
__device__ __forceinline__ bool asr_tf_map(const AsrTf8& t, float x, float y, float& ix, float& iy) {
    const float nx = t.a0 * x + t.a1 * y + t.a2;
    const float ny = t.b0 * x + t.b1 * y + t.b2;
    if (t.c0 == 0.0f && t.c1 == 0.0f) {  // affine (every transform the reference builds): k == 1, v / 1 == v
        ix = nx;
        iy = ny;
        return true;
    }
    const float proj = t.c0 * x + t.c1 * y + 1.0f;
    ix = nx / proj;
    iy = ny / proj;
    return proj != 0.0f;
}

// Bilinear read with TF's weight order; rd(yi, xi) must return 0 for out-of-bounds taps.
template <class Read>
__device__ __forceinline__ float asr_tf_bilinear(Read rd, float ix, float iy) {
    const float xf = floorf(ix), yf = floorf(iy);
    const float xc = xf + 1.0f, yc = yf + 1.0f;
    const int x0 = asr_coord_to_int(xf), y0 = asr_coord_to_int(yf);
    const float v00 = rd(y0, x0), v01 = rd(y0, x0 + 1);
    const float v10 = rd(y0 + 1, x0), v11 = rd(y0 + 1, x0 + 1);
    const float wxl = xc - ix, wxh = ix - xf;
    const float vyf = wxl * v00 + wxh * v01;
    const float vyc = wxl * v10 + wxh * v11;
    return (yc - iy) * vyf + (iy - yf) * vyc;
}

// Sample rd through transform t at output pixel (x, y).
template <class Read>
__device__ __forceinline__ float asr_tf_sample(const AsrTf8& t, Read rd, int x, int y) {
    float ix, iy;
    if (!asr_tf_map(t, (float)x, (float)y, ix, iy)) return 0.0f;
    return asr_tf_bilinear(rd, ix, iy);
}

// TF2 half-pixel bilinear resize tap set for one output coordinate:
// in = (o + 0.5) * scale - 0.5 ; lower = max(floor(in), 0) ; upper = min(ceil(in), n - 1).
struct AsrLerp {
    int lo, hi;
    float t;
};
__device__ __forceinline__ AsrLerp asr_half_pixel(int o, float scale, int in_size) {
    const float pos = ((float)o + 0.5f) * scale - 0.5f;
    const float fl = floorf(pos);
    AsrLerp r;
    r.lo = max((int)fl, 0);
    r.hi = min((int)ceilf(pos), in_size - 1);
    r.t = pos - fl;
    return r;
}
