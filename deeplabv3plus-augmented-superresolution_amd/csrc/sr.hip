// Iterative TV-regularised augmented super-resolution (ASR) solver and the max/mean realign
// fusions, as fused gather kernels for gfx950.
//
// Reference (paths in /root/reference):
//   superresolution_scripts/superresolution.py:44-100   loss_function
//   superresolution_scripts/superresolution.py:102-137  augmented_superresolution (the loop)
//   superresolution_scripts/superresolution.py:139-161  max_/mean_superresolution
//   superresolution_scripts/optimizer.py:37-52          Adam / AMSGrad + ExponentialDecay
//
// The TF formulation materialises ~7 tensors of [N,H,W,1] per iteration (0.7-1.4 GB of HBM
// traffic at N=100, 512x512).  Here one iteration is two launches that touch only the
// algorithmic minimum: y / residual [N,h,w] and x + Adam state [H,W]:
//   K_fwd : residual[n,i,j] = D(T_n(R_n(x)))[i,j] - y[n,i,j]   (two-stage bilinear kept exact)
//   K_bwd : g = sum_n InvRot_n(InvTrans_n(D^T(2*lambda*residual_n)))  -- TensorFlow's registered
//           gradient of ImageProjectiveTransformV3 (inverse warp of the upstream gradient,
//           NOT the scatter adjoint) -- + TV/L2/L1 prior gradients, then the Keras
//           Adam/AMSGrad update, all in one pass over the HR pixels.
#include "asr_warp_device.h"

namespace {

constexpr int kTileX = 32, kTileY = 8;  // 256 threads = 4 waves, 2 HR rows per wave

struct SrDims {
    int batch, n, H, W, h, w, f;  // f = H / h = W / w (even)
};

// ---- x0 = tf.image.resize(y[b,0], (H,W))  (superresolution.py:112-113) -------------------
__global__ __launch_bounds__(256) void sr_init_kernel(const float* __restrict__ y, float* __restrict__ x,
                                                      SrDims d, float scale_y, float scale_x) {
    const int X = blockIdx.x * kTileX + threadIdx.x;
    const int Y = blockIdx.y * kTileY + threadIdx.y;
    const int b = blockIdx.z;
    if (X >= d.W || Y >= d.H) return;
    const float* src = y + (int64_t)b * d.n * d.h * d.w;  // copy 0 of image b
    const AsrLerp ly = asr_half_pixel(Y, scale_y, d.h);
    const AsrLerp lx = asr_half_pixel(X, scale_x, d.w);
    const float tl = src[ly.lo * d.w + lx.lo], tr = src[ly.lo * d.w + lx.hi];
    const float bl = src[ly.hi * d.w + lx.lo], br = src[ly.hi * d.w + lx.hi];
    const float top = tl + (tr - tl) * lx.t;
    const float bot = bl + (br - bl) * lx.t;
    x[((int64_t)b * d.H + Y) * d.W + X] = top + (bot - top) * ly.t;
}

// ---- zero-bordered planes ---------------------------------------------------------------------------
// The solver keeps the images its bilinear gathers sample (the current x for K_fwd, the per-copy gradient planes G_R for
// the backward gather) with a zero border: row stride W + 64 (32 zero columns each side), 2 zero rows above and below.  A
// 2 x 2 tap block whose floor coordinate is clamped to [-2, W] x [-2, H] then reads zeros exactly where the unclamped
// taps fall outside the image (TF's CONSTANT-0 fill), so a sample is two unaligned 8-byte loads with no per-tap bounds
// test and no select; the weights still come from the unclamped coordinate.  Same products and sums as
// asr_tf_bilinear over a bounds-checked reader -> bit-identical.
// 32 border columns = one 128-byte line: every row's payload then starts on a line (with 4 columns -- enough for the taps --
// each 256-byte row segment a wave writes straddled three lines: K_gt + K_bwd 68 -> 62 us per iteration at N = 100).
#if defined(ASR_DIAG_KFWD_CHECK) || defined(ASR_DIAG_KFWD_NOPK)
// unpacked arithmetic written out in asm (diagnostic builds only): the compiler can neither pair nor reorder these
__device__ __forceinline__ float dg_mul(float x, float y) { float z; asm volatile("v_mul_f32 %0, %1, %2" : "=v"(z) : "v"(x), "v"(y)); return z; }
__device__ __forceinline__ float dg_add(float x, float y) { float z; asm volatile("v_add_f32 %0, %1, %2" : "=v"(z) : "v"(x), "v"(y)); return z; }
__device__ __forceinline__ float dg_sub(float x, float y) { float z; asm volatile("v_sub_f32 %0, %1, %2" : "=v"(z) : "v"(x), "v"(y)); return z; }
#endif
#ifndef ASR_DIAG_KFWD_NOPK
#define ASR_DIAG_KFWD_NOPK 0      // bits: 1 the coordinate map, 2 the bilinear sample, 4 the translate blend + D -- that stage of K_fwd in unpacked asm
#endif
constexpr int kGrPadX = 32, kGrPadY = 2;
__host__ __device__ inline size_t sr_gr_plane_elems(int H, int W) { return (size_t)(H + 2 * kGrPadY) * (size_t)(W + 2 * kGrPadX); }
typedef float asr_f2u __attribute__((ext_vector_type(2), aligned(4)));

// plane = first element of the bordered plane (its top-left border corner), WP = W + 2 * kGrPadX.  The element offset is
// non-negative, so the loads take the scalar plane base + a 32-bit lane offset.
__device__ __forceinline__ float sr_bilinear_bordered(const float* __restrict__ plane, int WP, int H, int W, float ix, float iy) {
    const float xf = floorf(ix), yf = floorf(iy);
    // min(max(asr_coord_to_int(f), -2), size) with the clamp done on the float (exact: f is an integer, the bounds are small):
    // one v_med3_f32 + one conversion per coordinate instead of four instructions; a NaN gives -2 either way (v_med3 returns
    // the minimum of the other two, asr_coord_to_int -1e9).  The gathers are bound by VALU issue.
    const int x0 = (int)__builtin_amdgcn_fmed3f(xf, -2.0f, (float)W), y0 = (int)__builtin_amdgcn_fmed3f(yf, -2.0f, (float)H);
    // byte offset in 32 bits: the loads take the scalar plane base + this lane offset (no 64-bit vector arithmetic)
    // (y0 + pad) * row bytes + (x0 + pad) * 4 as one 24-bit multiply-add (|y0|, row bytes < 2^23) and one shift-add
    const unsigned off = (unsigned)(__mul24(y0, WP * 4) + (kGrPadY * WP * 4 + kGrPadX * 4) + x0 * 4);
    const char* const base = reinterpret_cast<const char*>(plane);
    const char* const base_next = base + WP * 4;               // the next row through a second SCALAR base, not a lane add
    const asr_f2u top = *reinterpret_cast<const asr_f2u*>(base + off);
    const asr_f2u bot = *reinterpret_cast<const asr_f2u*>(base_next + off);
#ifdef ASR_DIAG_KFWD_WAIT      // diagnostic: every sample waits for its own two loads before any arithmetic on them (no loads in flight
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // of THIS wave while its packed ops execute)
#endif
#if ASR_DIAG_KFWD_NOPK & 2
    const float wxl = dg_sub(dg_add(xf, 1.0f), ix), wxh = dg_sub(ix, xf);
    const float vyf = dg_add(dg_mul(wxl, top.x), dg_mul(wxh, top.y));
    const float vyc = dg_add(dg_mul(wxl, bot.x), dg_mul(wxh, bot.y));
    return dg_add(dg_mul(dg_sub(dg_add(yf, 1.0f), iy), vyf), dg_mul(dg_sub(iy, yf), vyc));
#else
    const float wxl = (xf + 1.0f) - ix, wxh = ix - xf;
    const float vyf = wxl * top.x + wxh * top.y;
    const float vyc = wxl * bot.x + wxh * bot.y;
    return ((yf + 1.0f) - iy) * vyf + (iy - yf) * vyc;
#endif
}

// asr_tf_map for a transform known to be affine (c0 == c1 == 0): the same two expressions without the (uniform) branch
// on the projective terms -- a branch inside a sampling loop makes the compiler wait for each sample's loads in turn.
__device__ __forceinline__ void sr_map_affine(const AsrTf8& t, float x, float y, float& ix, float& iy) {
#if ASR_DIAG_KFWD_NOPK & 1
    ix = dg_add(dg_add(dg_mul(t.a0, x), dg_mul(t.a1, y)), t.a2);
    iy = dg_add(dg_add(dg_mul(t.b0, x), dg_mul(t.b1, y)), t.b2);
#else
    ix = t.a0 * x + t.a1 * y + t.a2;
    iy = t.b0 * x + t.b1 * y + t.b2;
#endif
}

#ifdef ASR_DIAG_KFWD_CHECK
// Diagnostic build only (tools/build_hazard_variants.py "kfwd_check"): K_fwd re-derives every intermediate of its fast path with
// UNPACKED arithmetic written out in asm (v_mul_f32 / v_add_f32 / v_sub_f32 never went wrong) and counts, stage by stage, where
// the compiler's (packed) code disagrees -- which of its packed-f32 sequences fails beside co-resident MFMA waves.
//   counters: 0 map ix   1 map iy   2 bilinear (weights / blend)   3 translate blend Tq   4 final D   5 waves with any mismatch
//             8 + lane (64): mismatching lanes   80 + 4 * a + c... (9): which of the 3 x 3 rotation samples (map stage)
//   records (first 16 mismatches): stage, lane, inputs, got, expected
__device__ unsigned g_kfwd_cnt[128];
__device__ float g_kfwd_rec[16][12];
__device__ __forceinline__ bool dg_ne(float a, float b) { return __float_as_int(a) != __float_as_int(b); }
__device__ __forceinline__ void dg_record(int stage, float i0, float i1, float i2, float i3, float i4, float i5, float got, float want) {
    atomicAdd(&g_kfwd_cnt[stage], 1u);
    atomicAdd(&g_kfwd_cnt[8 + ((threadIdx.y * kTileX + threadIdx.x) & 63)], 1u);
    const unsigned k = atomicAdd(&g_kfwd_cnt[7], 1u);
    if (k < 16) {
        float* r = g_kfwd_rec[k];
        r[0] = (float)stage; r[1] = (float)((threadIdx.y * kTileX + threadIdx.x) & 63);
        r[2] = i0; r[3] = i1; r[4] = i2; r[5] = i3; r[6] = i4; r[7] = i5; r[8] = got; r[9] = want;
        r[10] = (float)blockIdx.z; r[11] = (float)(blockIdx.y * 1000 + blockIdx.x);
    }
}
#endif

// ---- K_fwd --------------------------------------------------------------------------------
// One thread per LR residual element (b, n, i, j).  BORDERED: x is the solver's zero-bordered copy [batch, H+4, W+64].
#ifdef ASR_DIAG_KFWD_MAX_VGPR
#define ASR_KFWD_ATTR __attribute__((amdgpu_num_vgpr(ASR_DIAG_KFWD_MAX_VGPR)))
#else
#define ASR_KFWD_ATTR
#endif
template <bool BORDERED>
__global__ __launch_bounds__(256) ASR_KFWD_ATTR void sr_forward_residual_kernel(
    const float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ rot_tf,
    const float* __restrict__ trans_tf, float* __restrict__ resid, SrDims d) {
#ifdef ASR_DIAG_KFWD_TOP_VGPR
    ASR_DIAG_TOUCH_VGPR(ASR_DIAG_KFWD_TOP_VGPR);
#endif
    const int j = blockIdx.x * kTileX + threadIdx.x;
    const int i = blockIdx.y * kTileY + threadIdx.y;
    const int bn = blockIdx.z;  // b * n + copy
    if (j >= d.w || i >= d.h) return;
    const int b = bn / d.n;
    const int H = d.H, W = d.W, WP = W + 2 * kGrPadX;
    const float* img = BORDERED ? x + (size_t)b * sr_gr_plane_elems(H, W) : x + (int64_t)b * H * W;
    const AsrTf8 tr = asr_load_tf(rot_tf + (int64_t)bn * 8);
    const AsrTf8 tt = asr_load_tf(trans_tf + (int64_t)bn * 8);

    // Branch-free taps: clamp the index, load unconditionally, select afterwards -- the 64 image taps of one
    // residual element are then issued back to back instead of one exec-masked branch + wait each.
    auto rd_x = [&](int yy, int xx) -> float {
        const bool ok = (yy >= 0) & (yy < H) & (xx >= 0) & (xx < W);
        const float v = img[ok ? yy * W + xx : 0];
        return ok ? v : 0.0f;
    };
    auto rd_rot = [&](int yr, int xr) -> float {
        const bool ok = (yr >= 0) & (yr < H) & (xr >= 0) & (xr < W);
        float v;
        if (BORDERED) {
            float ix, iy;
            v = asr_tf_map(tr, (float)xr, (float)yr, ix, iy) ? sr_bilinear_bordered(img, WP, H, W, ix, iy) : 0.0f;
        } else {
            v = asr_tf_sample(tr, rd_x, xr, yr);
        }
        return ok ? v : 0.0f;
    };
    auto T = [&](int yt, int xt) -> float { return asr_tf_sample(tt, rd_rot, xt, yt); };

    // D: half-pixel bilinear, even integer factor f -> taps (f*o + f/2 - 1, +1), lerp 0.5
    const int y0 = d.f * i + d.f / 2 - 1, x0 = d.f * j + d.f / 2 - 1;
    float tl, trv, bl, br;
    // Pure translation (always, for tfa.image.translate): the four T pixels read a 3 x 3 block of R pixels
    // (2 x 2 taps each, shifted by one); evaluate those 9 rotation samples once instead of 16 times.  Same
    // per-tap arithmetic as the generic path, so the results are bit-identical; the generic path stays for
    // other transforms and for the (float-rounding) case where the two tap columns / rows do not abut.
    const bool pure_translation = (tt.a0 == 1.0f) & (tt.a1 == 0.0f) & (tt.b0 == 0.0f) & (tt.b1 == 1.0f) &
                                  (tt.c0 == 0.0f) & (tt.c1 == 0.0f);
    const float jx0 = (float)x0 + tt.a2, jx1 = (float)(x0 + 1) + tt.a2;
    const float jy0 = (float)y0 + tt.b2, jy1 = (float)(y0 + 1) + tt.b2;
    const float fx0 = floorf(jx0), fx1 = floorf(jx1), fy0 = floorf(jy0), fy1 = floorf(jy1);
    const int cx0 = asr_coord_to_int(fx0), cx1 = asr_coord_to_int(fx1);
    const int cy0 = asr_coord_to_int(fy0), cy1 = asr_coord_to_int(fy1);
    const bool rot_affine = (tr.c0 == 0.0f) & (tr.c1 == 0.0f);
    if (pure_translation && cx1 == cx0 + 1 && cy1 == cy0 + 1) {
        float rv[3][3];
        if (BORDERED && rot_affine) {   // no branch inside: the 18 loads of the 9 samples go out together
#pragma unroll
            for (int a = 0; a < 3; ++a)
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const int yr = cy0 + a, xr = cx0 + c;
                    float ix, iy;
                    sr_map_affine(tr, (float)xr, (float)yr, ix, iy);
                    const float v = sr_bilinear_bordered(img, WP, H, W, ix, iy);
                    rv[a][c] = ((yr >= 0) & (yr < H) & (xr >= 0) & (xr < W)) ? v : 0.0f;
#ifdef ASR_DIAG_KFWD_CHECK
                    {
                        const float fx = (float)xr, fy = (float)yr;
                        const float ex = dg_add(dg_add(dg_mul(tr.a0, fx), dg_mul(tr.a1, fy)), tr.a2);
                        const float ey = dg_add(dg_add(dg_mul(tr.b0, fx), dg_mul(tr.b1, fy)), tr.b2);
                        if (dg_ne(ix, ex)) { dg_record(0, fx, fy, tr.a0, tr.a1, tr.a2, (float)(3 * a + c), ix, ex); atomicAdd(&g_kfwd_cnt[80 + 3 * a + c], 1u); }
                        if (dg_ne(iy, ey)) { dg_record(1, fx, fy, tr.b0, tr.b1, tr.b2, (float)(3 * a + c), iy, ey); atomicAdd(&g_kfwd_cnt[80 + 3 * a + c], 1u); }
                        // the bilinear sample from the EXPECTED coordinates, unpacked (same loads: clamped floor, two 8-byte taps)
                        const float xf = floorf(ex), yf = floorf(ey);
                        const int x0 = (int)__builtin_amdgcn_fmed3f(xf, -2.0f, (float)W), y0 = (int)__builtin_amdgcn_fmed3f(yf, -2.0f, (float)H);
                        const float* q = img + (size_t)(y0 + kGrPadY) * WP + (x0 + kGrPadX);
                        const float t0 = q[0], t1 = q[1], b0 = q[WP], b1 = q[WP + 1];
                        const float wxl = dg_sub(dg_add(xf, 1.0f), ex), wxh = dg_sub(ex, xf);
                        const float vyf = dg_add(dg_mul(wxl, t0), dg_mul(wxh, t1)), vyc = dg_add(dg_mul(wxl, b0), dg_mul(wxh, b1));
                        const float ev = dg_add(dg_mul(dg_sub(dg_add(yf, 1.0f), ey), vyf), dg_mul(dg_sub(ey, yf), vyc));
                        if (!dg_ne(ix, ex) && !dg_ne(iy, ey) && dg_ne(v, ev)) dg_record(2, ex, ey, t0, t1, b0, b1, v, ev);
                    }
#endif
                }
        } else {
#pragma unroll
            for (int a = 0; a < 3; ++a)
#pragma unroll
                for (int c = 0; c < 3; ++c) rv[a][c] = rd_rot(cy0 + a, cx0 + c);
        }
#if ASR_DIAG_KFWD_NOPK & 4
        const float wxl0 = dg_sub(dg_add(fx0, 1.0f), jx0), wxh0 = dg_sub(jx0, fx0), wxl1 = dg_sub(dg_add(fx1, 1.0f), jx1), wxh1 = dg_sub(jx1, fx1);
        const float wyl0 = dg_sub(dg_add(fy0, 1.0f), jy0), wyh0 = dg_sub(jy0, fy0), wyl1 = dg_sub(dg_add(fy1, 1.0f), jy1), wyh1 = dg_sub(jy1, fy1);
        auto Tq = [&](int a, int c, float wxl, float wxh, float wyl, float wyh) -> float {
            const float vyf = dg_add(dg_mul(wxl, rv[a][c]), dg_mul(wxh, rv[a][c + 1]));
            const float vyc = dg_add(dg_mul(wxl, rv[a + 1][c]), dg_mul(wxh, rv[a + 1][c + 1]));
            return dg_add(dg_mul(wyl, vyf), dg_mul(wyh, vyc));
        };
#else
        const float wxl0 = (fx0 + 1.0f) - jx0, wxh0 = jx0 - fx0, wxl1 = (fx1 + 1.0f) - jx1, wxh1 = jx1 - fx1;
        const float wyl0 = (fy0 + 1.0f) - jy0, wyh0 = jy0 - fy0, wyl1 = (fy1 + 1.0f) - jy1, wyh1 = jy1 - fy1;
        auto Tq = [&](int a, int c, float wxl, float wxh, float wyl, float wyh) -> float {
            const float vyf = wxl * rv[a][c] + wxh * rv[a][c + 1];
            const float vyc = wxl * rv[a + 1][c] + wxh * rv[a + 1][c + 1];
            return wyl * vyf + wyh * vyc;
        };
#endif
        tl = Tq(0, 0, wxl0, wxh0, wyl0, wyh0);
        trv = Tq(0, 1, wxl1, wxh1, wyl0, wyh0);
        bl = Tq(1, 0, wxl0, wxh0, wyl1, wyh1);
        br = Tq(1, 1, wxl1, wxh1, wyl1, wyh1);
#ifdef ASR_DIAG_KFWD_CHECK
        {
            auto eTq = [&](int a, int c, float jx, float fxx, float jy, float fyy) -> float {
                const float wl = dg_sub(dg_add(fxx, 1.0f), jx), wh = dg_sub(jx, fxx), yl = dg_sub(dg_add(fyy, 1.0f), jy), yh = dg_sub(jy, fyy);
                const float vf = dg_add(dg_mul(wl, rv[a][c]), dg_mul(wh, rv[a][c + 1]));
                const float vc = dg_add(dg_mul(wl, rv[a + 1][c]), dg_mul(wh, rv[a + 1][c + 1]));
                return dg_add(dg_mul(yl, vf), dg_mul(yh, vc));
            };
            const float e0 = eTq(0, 0, jx0, fx0, jy0, fy0), e1 = eTq(0, 1, jx1, fx1, jy0, fy0), e2 = eTq(1, 0, jx0, fx0, jy1, fy1),
                        e3 = eTq(1, 1, jx1, fx1, jy1, fy1);
            if (dg_ne(tl, e0)) dg_record(3, jx0, jy0, rv[0][0], rv[0][1], rv[1][0], rv[1][1], tl, e0);
            if (dg_ne(trv, e1)) dg_record(3, jx1, jy0, rv[0][1], rv[0][2], rv[1][1], rv[1][2], trv, e1);
            if (dg_ne(bl, e2)) dg_record(3, jx0, jy1, rv[1][0], rv[1][1], rv[2][0], rv[2][1], bl, e2);
            if (dg_ne(br, e3)) dg_record(3, jx1, jy1, rv[1][1], rv[1][2], rv[2][1], rv[2][2], br, e3);
        }
#endif
    } else {
        tl = T(y0, x0); trv = T(y0, x0 + 1);
        bl = T(y0 + 1, x0); br = T(y0 + 1, x0 + 1);
    }
    const float top = tl + (trv - tl) * 0.5f;
    const float bot = bl + (br - bl) * 0.5f;
    const float dval = top + (bot - top) * 0.5f;
    const int64_t o = ((int64_t)bn * d.h + i) * d.w + j;
    resid[o] = dval - y[o];
#ifdef ASR_DIAG_KFWD_CHECK
    {
        const float et = dg_add(tl, dg_mul(dg_sub(trv, tl), 0.5f)), eb = dg_add(bl, dg_mul(dg_sub(br, bl), 0.5f));
        const float ed = dg_add(et, dg_mul(dg_sub(eb, et), 0.5f));
        if (dg_ne(dval, ed)) dg_record(4, tl, trv, bl, br, top, bot, dval, ed);
    }
#endif
}

// ---- K_bwd --------------------------------------------------------------------------------
// Update rule and prior of one solver step (from asr_sr_config, include/asr_hip.h).
struct SrStep {
    int optimizer, flag;      // ASR_OPT_*, amsgrad / nesterov
    float c0, c1, c2;         // hyper-parameters, meaning per optimizer (asr_hip.h)
    int prior, btv_shift;     // ASR_PRIOR_*, bilateral-TV window
    float btv_w[9];           // alpha^k, k = |h| + |v| (float32 pow like tf.pow)
};

// One axis of the inverse-translate stage for one integer G_R coordinate c (pure translation:
// the map is c + shift exactly, so x and y separate).  Bitwise the same values the generic
// asr_tf_sample path produces, computed once per axis instead of once per tap.
struct SrAxisTap {
    float wl, wh;   // (c_ceil - pos), (pos - c_floor)
    int l0, l1;     // LR index of the two G_T taps, or -1 when that tap is structurally zero
    bool inb;       // c itself inside the image (else G_R(c) = 0)
};

template <int LOG2F>
__device__ __forceinline__ int sr_gt_index(int t, int size, int f, int ph0, int ph1) {
    // G_T = ResizeBilinearGrad(g_D) is non-zero only on the 2x2 centre of every f x f block
    if (t < 0 || t >= size) return -1;
    const int ph = (LOG2F > 0) ? (t & (f - 1)) : (t % f);
    if (ph != ph0 && ph != ph1) return -1;
    return (LOG2F > 0) ? (t >> LOG2F) : (t / f);
}

template <int LOG2F>
__device__ __forceinline__ SrAxisTap sr_axis_tap(int c, float shift, int size, int f, int ph0, int ph1) {
    SrAxisTap a;
    a.inb = (c >= 0 && c < size);
    const float pos = (float)c + shift;
    const float fl = floorf(pos);
    a.wl = (fl + 1.0f) - pos;
    a.wh = pos - fl;
    const int t0 = asr_coord_to_int(fl);
    a.l0 = sr_gt_index<LOG2F>(t0, size, f, ph0, ph1);
    a.l1 = sr_gt_index<LOG2F>(t0 + 1, size, f, ph0, ph1);
    return a;
}

// Everything after the data-term gradient of one HR pixel: prior gradients and the optimiser update (shared by the
// fused and the gather backward kernels).
__device__ __forceinline__ void sr_prior_and_update(const float* __restrict__ x, float* __restrict__ x_new, float* __restrict__ m,
                                                    float* __restrict__ v, float* __restrict__ vhat,
                                                    const float* __restrict__ alphas, float* __restrict__ grad_out,
                                                    const SrDims& d, int b, int X, int Y, float g_df, float lambda_tv,
                                                    float two_lambda_l2, float lambda_l1, const SrStep& st,
                                                    float* __restrict__ x_bordered_out = nullptr) {
    const int H = d.H, W = d.W;
    // priors (superresolution.py:81-98): TV (forward differences, last row/col 0) or bilateral TV, L2, L1
    const float* img = x + (int64_t)b * H * W;
    const float xc = img[Y * W + X];
    auto sgn = [](float a) -> float { return (a > 0.0f) ? 1.0f : ((a < 0.0f) ? -1.0f : 0.0f); };
    float g_tv;
    if (st.prior == ASR_PRIOR_BTV) {
        // bilateral_tv (superresolution.py:8-23): sum over pairs p = (h, v), h in [-s, s], v in [0, s], of
        // alpha^(|h|+|v|) * || x - translate(x, p) ||_1 with translate(x, p)(q) = x(q - p), zero outside.
        // x enters twice: as the minuend (Abs grad = sign) and through the translate, whose registered gradient
        // samples the upstream at q + p.  Both sums run in the pair order of the reference's list comprehension.
        const int sh = st.btv_shift;
        float ga = 0.0f, gb = 0.0f;
        for (int hh = -sh; hh <= sh; ++hh)
            for (int vv = 0; vv <= sh; ++vv) {
                const float c = lambda_tv * st.btv_w[abs(hh) + vv];
                const int xs = X - hh, ys = Y - vv;
                const float sv = (xs >= 0 && xs < W && ys >= 0 && ys < H) ? img[ys * W + xs] : 0.0f;
                ga += sgn(xc - sv) * c;
            }
        for (int hh = -sh; hh <= sh; ++hh)
            for (int vv = 0; vv <= sh; ++vv) {
                const float c = lambda_tv * st.btv_w[abs(hh) + vv];
                const int xt = X + hh, yt = Y + vv;
                if (xt >= 0 && xt < W && yt >= 0 && yt < H) gb -= sgn(img[yt * W + xt] - xc) * c;
            }
        g_tv = ga + gb;
    } else {
        const float sy = (Y < H - 1) ? sgn(img[(Y + 1) * W + X] - xc) * lambda_tv : 0.0f;
        const float sx = (X < W - 1) ? sgn(img[Y * W + X + 1] - xc) * lambda_tv : 0.0f;
        g_tv = -sy - sx;
        if (Y > 0) g_tv += sgn(xc - img[(Y - 1) * W + X]) * lambda_tv;
        if (X > 0) g_tv += sgn(xc - img[Y * W + X - 1]) * lambda_tv;
    }
    float g = g_df + g_tv + two_lambda_l2 * xc;
    if (lambda_l1 > 0.0f) g = g + lambda_l1 * sgn(xc);

    const int64_t o = ((int64_t)b * H + Y) * W + X;
    if (grad_out) grad_out[o] = g;
    if (!x_new) return;

    // apply_gradients: the dense CPU kernels of tensorflow/core/kernels/training_ops.cc that Keras 2.7 dispatches to
    // (operation order as in their Eigen expressions; alpha = the per-step scalar the host prepared)
    const float alpha = alphas[b];
    float xn;
    switch (st.optimizer) {
        case ASR_OPT_SGD: {                                 // ApplyGradientDescent / ApplyKerasMomentum
            if (st.c0 == 0.0f) {
                xn = xc - g * alpha;
            } else {
                const float acc = m[o] * st.c0 - g * alpha;
                m[o] = acc;
                xn = st.flag ? xc + (acc * st.c0 - g * alpha) : xc + acc;
            }
            break;
        }
        case ASR_OPT_ADAGRAD: {                             // ApplyAdagradV2
            const float acc = v[o] + g * g;
            v[o] = acc;
            xn = xc - (g * alpha) / (sqrtf(acc) + st.c2);
            break;
        }
        case ASR_OPT_ADADELTA: {                            // ApplyAdadelta: c0 = rho, c1 = 1 - rho
            const float acc = v[o] * st.c0 + (g * g) * st.c1;
            v[o] = acc;
            const float au = m[o];
            const float upd = sqrtf(au + st.c2) * (1.0f / sqrtf(acc + st.c2)) * g;
            xn = xc - upd * alpha;
            m[o] = au * st.c0 + (upd * upd) * st.c1;
            break;
        }
        case ASR_OPT_ADAMAX: {                              // ApplyAdaMax: c0 = 1 - beta1, c1 = beta2
            float mm = m[o];
            mm = mm + (g - mm) * st.c0;
            const float vv = fmaxf(st.c1 * v[o], fabsf(g));
            m[o] = mm;
            v[o] = vv;
            xn = xc - alpha * (mm / (vv + st.c2));
            break;
        }
        default: {                                          // ApplyAdam[WithAmsgrad]: c0 = 1 - beta1, c1 = 1 - beta2
            float mm = m[o], vv = v[o];
            mm = mm + (g - mm) * st.c0;
            vv = vv + (g * g - vv) * st.c1;
            m[o] = mm;
            v[o] = vv;
            float denom;
            if (st.flag) {
                const float vh = fmaxf(vhat[o], vv);
                vhat[o] = vh;
                denom = sqrtf(vh) + st.c2;
            } else {
                denom = sqrtf(vv) + st.c2;
            }
            xn = xc - (mm * alpha) / denom;
        }
    }
    x_new[o] = xn;
    if (x_bordered_out)   // the solver's zero-bordered copy for the next K_fwd
        x_bordered_out[(size_t)b * sr_gr_plane_elems(H, W) + (size_t)(Y + kGrPadY) * (W + 2 * kGrPadX) + (X + kGrPadX)] = xn;
}

// Work split: a workgroup owns 64 HR pixels (32 x 2) and NSPLIT = 4 waves; wave g evaluates the copies
// n = g, g + 4, ... for those pixels (the copy index is wave-uniform, so the transforms are scalar loads)
// and parks each contribution in LDS as contrib[n][pixel]; after one barrier wave 0 adds the N
// contributions IN COPY ORDER (the summation order of the oracle, so results stay bit-identical) and
// applies the priors and the Adam update.  4x the exposed parallelism of a one-thread-per-pixel loop for a
// latency-bound gather, N * 256 bytes of LDS.
constexpr int kBwdSplit = 4, kBwdPixX = 32, kBwdPixY = 2, kBwdPix = kBwdPixX * kBwdPixY;

// Two-kernel form of the data-term gradient (asr_sr_solve_*): G_R(n, c) -- the gradient that reaches the rotation stage
// at the integer HR position c of copy n, i.e. the registered gradient of the translate applied to G_T -- depends on
// (n, c) only, yet the fused kernel re-derives it for each of the 4 rotation taps of every (pixel, copy) pair (16 nested
// taps, ~250 VALU instructions per pair; the kernel is VALU-bound).  sr_grad_translate_kernel evaluates it ONCE per
// (n, c) into a zero-bordered [batch*n, H + 4, W + 64] plane (written and re-read through L2 / the Infinity Cache: 119 MB
// at N = 100, 512^2) and sr_backward_gather_kernel takes the rotation's 4 taps from that plane.  Same products and sums
// on the same operands as the fused kernel (x blend, then y blend, then the rotation's bilinear weights) -> bit-identical.
//   Plane geometry: row stride W + 64 (32 zero columns each side), 2 zero rows above and below.  A tap pair whose floor
//   coordinate is clamped to [-2, W] x [-2, H] reads only zeros wherever the unclamped taps are out of the image, so the
//   gather needs no per-tap bounds test and no select.
constexpr int kGrRows = 16;   // HR rows per wave of sr_grad_translate_kernel

// One wave: 64 columns x kGrRows rows of one copy.  The x taps are per lane (its column); the y taps of the wave's rows
// are computed by lanes 0..kGrRows-1 in ONE pass and fetched per row with v_readlane (they are wave-uniform: scalar row
// pointers, scalar validity branches).  The x blend of an LR row, h(L) = wl * G(L, l0) + wh * G(L, l1), serves the up
// to 4 HR rows that tap L: a two-entry cache keyed by the (uniform) LR row index keeps the last two.
template <int LOG2F>
__global__ __launch_bounds__(256) void sr_grad_translate_kernel(const float* __restrict__ resid,
                                                                const float* __restrict__ inv_trans_tf,
                                                                float* __restrict__ gr_out, SrDims d, float two_lambda_df,
                                                                int n0, int cn) {
    // blockIdx.z = b * cn + local: copy n0 + local of image b, written to plane slot blockIdx.z of the chunk's planes
    const int slot = blockIdx.z;
    const int bn = (slot / cn) * d.n + n0 + (slot % cn);
    const int lane = threadIdx.x;
    const int cx = blockIdx.x * 64 + lane;
    const int cy0 = (blockIdx.y * 4 + __builtin_amdgcn_readfirstlane(threadIdx.y)) * kGrRows;
    const int H = d.H, W = d.W, f = d.f, lw = d.w, lh = d.h;
    const int ph0 = f / 2 - 1, ph1 = f / 2;
    const int WP = W + 2 * kGrPadX;
    const float* r = resid + (int64_t)bn * lh * lw;
    const AsrTf8 it = asr_load_tf(inv_trans_tf + (int64_t)bn * 8);
    float* const out = gr_out + (size_t)slot * sr_gr_plane_elems(H, W) + (size_t)kGrPadY * WP + kGrPadX;
    const bool col_ok = cx < W;
    const bool pure_translation = (it.a0 == 1.0f && it.a1 == 0.0f && it.b0 == 0.0f && it.b1 == 1.0f &&
                                   it.c0 == 0.0f && it.c1 == 0.0f);
    if (pure_translation) {
        const SrAxisTap ax = sr_axis_tap<LOG2F>(col_ok ? cx : 0, it.a2, W, f, ph0, ph1);
        const int c0 = max(ax.l0, 0), c1 = max(ax.l1, 0);
        const bool ok0 = ax.l0 >= 0, ok1 = ax.l1 >= 0;
        const SrAxisTap ayv = sr_axis_tap<LOG2F>(min(cy0 + lane, H - 1), it.b2, H, f, ph0, ph1);   // lane j: row cy0 + j
        auto xblend = [&](int L) -> float {                         // L >= 0, wave-uniform
            const float* row = r + L * lw;
            const float a = (two_lambda_df * row[c0]) * 0.25f, b = (two_lambda_df * row[c1]) * 0.25f;
            return ax.wl * (ok0 ? a : 0.0f) + ax.wh * (ok1 ? b : 0.0f);
        };
        int la = -1, lb = -1;          // cached LR rows (lb the more recent)
        float va = 0.0f, vb = 0.0f;
        auto h_of = [&](int L) -> float {
            if (L < 0) return 0.0f;
            if (L == lb) return vb;
            if (L == la) return va;
            la = lb; va = vb;
            lb = L; vb = xblend(L);
            return vb;
        };
        const int rows = min(kGrRows, H - cy0);
        for (int j = 0; j < rows; ++j) {
            const float wl = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, ayv.wl), j));
            const float wh = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, ayv.wh), j));
            const int l0 = __builtin_amdgcn_readlane(ayv.l0, j), l1 = __builtin_amdgcn_readlane(ayv.l1, j);
            const float h0 = h_of(l0), h1 = h_of(l1);
            if (col_ok) out[(cy0 + j) * WP + cx] = wl * h0 + wh * h1;
        }
    } else {   // generic projective inverse transforms (never produced by the reference's translate)
        auto gt_at = [&](int ly, int lx) -> float {
            const bool ok = (ly >= 0) & (lx >= 0);
            const float v = (two_lambda_df * r[ok ? ly * lw + lx : 0]) * 0.25f;
            return ok ? v : 0.0f;
        };
        auto rd_gt = [&](int yt, int xt) -> float {
            return gt_at(sr_gt_index<LOG2F>(yt, H, f, ph0, ph1), sr_gt_index<LOG2F>(xt, W, f, ph0, ph1));
        };
        for (int j = 0; j < kGrRows; ++j) {
            const int cy = cy0 + j;
            if (cy < H && col_ok) out[cy * WP + cx] = asr_tf_sample(it, rd_gt, cx, cy);
        }
    }
}

template <int LOG2F>
__global__ __launch_bounds__(256) void sr_backward_adam_kernel(
    const float* __restrict__ x, float* __restrict__ x_new, const float* __restrict__ resid,
    const float* __restrict__ inv_rot_tf, const float* __restrict__ inv_trans_tf,
    float* __restrict__ m, float* __restrict__ v, float* __restrict__ vhat,
    const float* __restrict__ alphas /* [batch] for this iteration */, float* __restrict__ grad_out,
    SrDims d, float two_lambda_df, float lambda_tv, float two_lambda_l2, float lambda_l1, SrStep st) {
    extern __shared__ float contrib[];                     // [n][kBwdPix]
    const int X = blockIdx.x * kBwdPixX + threadIdx.x;
    const int Y = blockIdx.y * kBwdPixY + threadIdx.y;
    const int b = blockIdx.z;
    const int grp = threadIdx.z;                           // copy group == wave index
    const int pix = threadIdx.y * kBwdPixX + threadIdx.x;
    const bool in_image = (X < d.W) & (Y < d.H);
    const int H = d.H, W = d.W, f = d.f, lw = d.w, lh = d.h;
    const int ph0 = f / 2 - 1, ph1 = f / 2;

    // Two copies per trip: two independent instruction streams for the compiler to interleave (and to pair in packed
    // f32 math where it can) in a loop that is bound by VALU issue, not by memory.
    auto contribution = [&](int n) -> float {
        float g_df = 0.0f;
        const int bn = b * d.n + n;
        const AsrTf8 ir = asr_load_tf(inv_rot_tf + (int64_t)bn * 8);
        const float* r = resid + (int64_t)bn * lh * lw;
        const AsrTf8 it = asr_load_tf(inv_trans_tf + (int64_t)bn * 8);
        auto gt_at = [&](int ly, int lx) -> float {   // branch-free: clamped index, unconditional load, select
            const bool ok = (ly >= 0) & (lx >= 0);
            const float v = (two_lambda_df * r[ok ? ly * lw + lx : 0]) * 0.25f;
            return ok ? v : 0.0f;
        };
        const bool pure_translation = (it.a0 == 1.0f && it.a1 == 0.0f && it.b0 == 0.0f && it.b1 == 1.0f &&
                                       it.c0 == 0.0f && it.c1 == 0.0f);
        if (pure_translation) {
            // rotation stage (generic affine), then the separable translate stage
            float ix, iy;
            asr_tf_map(ir, (float)X, (float)Y, ix, iy);
            const float xf = floorf(ix), yf = floorf(iy);
            const int x0 = asr_coord_to_int(xf), y0 = asr_coord_to_int(yf);
            const SrAxisTap ax0 = sr_axis_tap<LOG2F>(x0, it.a2, W, f, ph0, ph1);
            const SrAxisTap ax1 = sr_axis_tap<LOG2F>(x0 + 1, it.a2, W, f, ph0, ph1);
            const SrAxisTap ay0 = sr_axis_tap<LOG2F>(y0, it.b2, H, f, ph0, ph1);
            const SrAxisTap ay1 = sr_axis_tap<LOG2F>(y0 + 1, it.b2, H, f, ph0, ph1);
            // The 4 x 4 G_T taps under this pixel sit on 3 consecutive HR positions per axis, i.e. on at most
            // 2 x 2 distinct LR residual cells: load those four values once and let every tap select its own
            // (16 gathers -> 4; the texture-address path, not the VALU, bounded this kernel).
            const int big = 0x3fffffff;
            auto lo_of = [&](int a, int b2, int c, int e) {
                return min(min(a >= 0 ? a : big, b2 >= 0 ? b2 : big), min(c >= 0 ? c : big, e >= 0 ? e : big));
            };
            const int cxa = lo_of(ax0.l0, ax0.l1, ax1.l0, ax1.l1), cxb = max(max(ax0.l0, ax0.l1), max(ax1.l0, ax1.l1));
            const int cya = lo_of(ay0.l0, ay0.l1, ay1.l0, ay1.l1), cyb = max(max(ay0.l0, ay0.l1), max(ay1.l0, ay1.l1));
            auto in2 = [](int l, int a, int b2) { return (l < 0) | (l == a) | (l == b2); };
            const bool two_cells = in2(ax0.l0, cxa, cxb) & in2(ax0.l1, cxa, cxb) & in2(ax1.l0, cxa, cxb) & in2(ax1.l1, cxa, cxb) &
                                   in2(ay0.l0, cya, cyb) & in2(ay0.l1, cya, cyb) & in2(ay1.l0, cya, cyb) & in2(ay1.l1, cya, cyb);
            const float wxl = (xf + 1.0f) - ix, wxh = ix - xf;
            if (two_cells) {
                const int xa = (cxa == big) ? 0 : cxa, xb = max(cxb, 0), ya = (cya == big) ? 0 : cya, yb = max(cyb, 0);
                const float saa = (two_lambda_df * r[ya * lw + xa]) * 0.25f, sab = (two_lambda_df * r[ya * lw + xb]) * 0.25f;
                const float sba = (two_lambda_df * r[yb * lw + xa]) * 0.25f, sbb = (two_lambda_df * r[yb * lw + xb]) * 0.25f;
                // The 16 taps select among the four cell values; selection and the x blend commute with the row choice,
                // so the x blend is done ONCE per (tap column pair, cell row) and the taps only pick a row afterwards --
                // the same products and sums on the same operands as tap-by-tap selection (an invalid tap contributes
                // w * 0 = +0 either way), a third of the v_cndmask / compare instructions this VALU-bound loop spent.
                auto cell = [&](int lx, float va, float vb) -> float {           // value of column lx in one cell row
                    const float v = (lx == cxa) ? va : vb;
                    return (lx >= 0) ? v : 0.0f;
                };
                auto hx = [&](const SrAxisTap& ax, float va, float vb) -> float {   // x blend of a tap column pair in one cell row
                    return ax.wl * cell(ax.l0, va, vb) + ax.wh * cell(ax.l1, va, vb);
                };
                const float h0a = hx(ax0, saa, sab), h0b = hx(ax0, sba, sbb);
                const float h1a = hx(ax1, saa, sab), h1b = hx(ax1, sba, sbb);
                auto row = [&](int ly, float ha, float hb) -> float {              // the x-blended value of tap row ly
                    const float v = (ly == cya) ? ha : hb;
                    return (ly >= 0) ? v : 0.0f;
                };
                auto gr = [&](const SrAxisTap& ay, const SrAxisTap& ax, float ha, float hb) -> float {   // G_R at one integer position
                    const float v = ay.wl * row(ay.l0, ha, hb) + ay.wh * row(ay.l1, ha, hb);
                    return (ay.inb & ax.inb) ? v : 0.0f;
                };
                const float vyf = wxl * gr(ay0, ax0, h0a, h0b) + wxh * gr(ay0, ax1, h1a, h1b);
                const float vyc = wxl * gr(ay1, ax0, h0a, h0b) + wxh * gr(ay1, ax1, h1a, h1b);
                g_df += ((yf + 1.0f) - iy) * vyf + (iy - yf) * vyc;
            } else {   // > 2 distinct cells on an axis: only through float rounding at a binade edge; 16 direct gathers
                auto gr = [&](const SrAxisTap& ay, const SrAxisTap& ax) -> float {
                    const float vyf = ax.wl * gt_at(ay.l0, ax.l0) + ax.wh * gt_at(ay.l0, ax.l1);
                    const float vyc = ax.wl * gt_at(ay.l1, ax.l0) + ax.wh * gt_at(ay.l1, ax.l1);
                    const float v = ay.wl * vyf + ay.wh * vyc;
                    return (ay.inb & ax.inb) ? v : 0.0f;
                };
                const float vyf = wxl * gr(ay0, ax0) + wxh * gr(ay0, ax1);
                const float vyc = wxl * gr(ay1, ax0) + wxh * gr(ay1, ax1);
                g_df += ((yf + 1.0f) - iy) * vyf + (iy - yf) * vyc;
            }
        } else {
            // generic projective inverse transforms (never produced by the reference's translate)
            auto rd_gt = [&](int yt, int xt) -> float {
                return gt_at(sr_gt_index<LOG2F>(yt, H, f, ph0, ph1), sr_gt_index<LOG2F>(xt, W, f, ph0, ph1));
            };
            auto rd_gr = [&](int yr, int xr) -> float {
                if (!(yr >= 0 && yr < H && xr >= 0 && xr < W)) return 0.0f;
                return asr_tf_sample(it, rd_gt, xr, yr);
            };
            g_df += asr_tf_sample(ir, rd_gr, X, Y);
        }
        return g_df;
    };
    int n = grp;
    for (; n + kBwdSplit < d.n; n += 2 * kBwdSplit) {
        const float g0 = contribution(n), g1 = contribution(n + kBwdSplit);
        contrib[n * kBwdPix + pix] = g0;
        contrib[(n + kBwdSplit) * kBwdPix + pix] = g1;
    }
    if (n < d.n) contrib[n * kBwdPix + pix] = contribution(n);
    __syncthreads();
    if (grp != 0 || !in_image) return;
    float g_df = 0.0f;
    for (int n = 0; n < d.n; ++n) g_df += contrib[n * kBwdPix + pix];   // fixed order n = 0..N-1

    sr_prior_and_update(x, x_new, m, v, vhat, alphas, grad_out, d, b, X, Y, g_df, lambda_tv, two_lambda_l2, lambda_l1, st);
}

// Backward, gather form: one thread per HR pixel walks the copies in order (the oracle's summation order) and takes
// the rotation's 2 x 2 taps of copy n from the zero-bordered G_R plane: two unaligned 8-byte loads, no bounds tests.
// The copy index is uniform, so the transforms are scalar loads.  The copies are taken K at a time: when all K rotations
// are affine (always, for the reference) the chunk is branch-free, so its 2K loads are in flight together and only the
// final adds are a dependent chain.
template <int K, bool AFFINE>
__device__ __forceinline__ float sr_gather_chunk(float g_df, const float* __restrict__ inv_rot_tf, const float* __restrict__ planes,
                                                 size_t plane, int WP, int H, int W, float fx, float fy) {
    AsrTf8 t[K];
#pragma unroll
    for (int u = 0; u < K; ++u) t[u] = asr_load_tf(inv_rot_tf + u * 8);
    float c[K];
    if (AFFINE) {
#pragma unroll
        for (int u = 0; u < K; ++u) {
            float ix, iy;
            sr_map_affine(t[u], fx, fy, ix, iy);
            c[u] = sr_bilinear_bordered(planes + u * plane, WP, H, W, ix, iy);
        }
    } else {
#pragma unroll
        for (int u = 0; u < K; ++u) {
            float ix, iy;
            c[u] = asr_tf_map(t[u], fx, fy, ix, iy) ? sr_bilinear_bordered(planes + u * plane, WP, H, W, ix, iy) : 0.0f;
        }
    }
#pragma unroll
    for (int u = 0; u < K; ++u) g_df += c[u];
    return g_df;
}

// flags[b] = 1 when every inverse rotation of image b is affine (c0 == c1 == 0: always, for the reference).  Evaluated once
// per solve -- the transforms do not change between the iterations -- instead of by every wave for every chunk of copies
// (two vector compares per copy and pixel in a kernel that is bound by VALU issue).
__global__ __launch_bounds__(64) void sr_affine_flags_kernel(const float* __restrict__ inv_rot_tf, int* __restrict__ flags, int n) {
    const int b = blockIdx.x;
    bool ok = true;
    for (int i = threadIdx.x; i < n; i += 64) {
        const float* t = inv_rot_tf + ((int64_t)b * n + i) * 8;
        ok = ok && t[6] == 0.0f && t[7] == 0.0f;
    }
    const bool all = __builtin_amdgcn_ballot_w64(!ok) == 0;
    if (threadIdx.x == 0) flags[b] = all ? 1 : 0;
}

// The copies are taken in chunks [n0, n0 + cn) (one launch per chunk, the chunk's planes written by the K_gt launch before
// it): the running sum of the data-term gradient crosses the launches through `acc` [batch, H, W] as a float32 -- the same
// sequence of float32 additions, in copy order, as one pass over all copies, so the result does not depend on the chunking.
// The last chunk adds the priors and applies the update.
__global__ __launch_bounds__(256) void sr_backward_gather_kernel(
    const float* __restrict__ x, float* __restrict__ x_new, const float* __restrict__ gr_planes,
    const float* __restrict__ inv_rot_tf, float* __restrict__ m, float* __restrict__ v, float* __restrict__ vhat,
    const float* __restrict__ alphas, float* __restrict__ grad_out, SrDims d, float lambda_tv, float two_lambda_l2,
    float lambda_l1, SrStep st, float* __restrict__ x_bordered_out, float* __restrict__ acc, int n0, int cn,
    const int* __restrict__ affine_flags) {
    const int X = blockIdx.x * 64 + threadIdx.x;
    const int Y = blockIdx.y * 4 + threadIdx.y;
    const int b = blockIdx.z;
    const int H = d.H, W = d.W, WP = W + 2 * kGrPadX;
    if (X >= W || Y >= H) return;
    const size_t plane = sr_gr_plane_elems(H, W);
    const float* const planes = gr_planes + (size_t)b * cn * plane;
    const float* const tfs = inv_rot_tf + ((int64_t)b * d.n + n0) * 8;
    const float fx = (float)X, fy = (float)Y;
    const int64_t o = ((int64_t)b * H + Y) * W + X;
    float g_df = (n0 == 0) ? 0.0f : acc[o];
    int n = 0;
    if (affine_flags[b]) {       // the copies K at a time, branch-free: 2 K loads in flight, only the final adds are a chain
        for (; n + 8 <= cn; n += 8) g_df = sr_gather_chunk<8, true>(g_df, tfs + n * 8, planes + n * plane, plane, WP, H, W, fx, fy);
        if (n + 4 <= cn) { g_df = sr_gather_chunk<4, true>(g_df, tfs + n * 8, planes + n * plane, plane, WP, H, W, fx, fy); n += 4; }
        for (; n < cn; ++n) g_df = sr_gather_chunk<1, true>(g_df, tfs + n * 8, planes + n * plane, plane, WP, H, W, fx, fy);
    } else {
        for (; n < cn; ++n) g_df = sr_gather_chunk<1, false>(g_df, tfs + n * 8, planes + n * plane, plane, WP, H, W, fx, fy);
    }
    if (n0 + cn < d.n) { acc[o] = g_df; return; }
    sr_prior_and_update(x, x_new, m, v, vhat, alphas, grad_out, d, b, X, Y, g_df, lambda_tv, two_lambda_l2, lambda_l1, st,
                        x_bordered_out);
}

typedef void (*SrBwdKernel)(const float*, float*, const float*, const float*, const float*, float*, float*, float*,
                            const float*, float*, SrDims, float, float, float, float, SrStep);

SrBwdKernel sr_backward_kernel_for(int f) {
    switch (f) {
        case 2: return sr_backward_adam_kernel<1>;
        case 4: return sr_backward_adam_kernel<2>;
        case 8: return sr_backward_adam_kernel<3>;
        default: return sr_backward_adam_kernel<0>;   // any other even factor: integer division
    }
}

typedef void (*SrGradTranslateKernel)(const float*, const float*, float*, SrDims, float, int, int);
SrGradTranslateKernel sr_grad_translate_kernel_for(int f) {
    switch (f) {
        case 2: return sr_grad_translate_kernel<1>;
        case 4: return sr_grad_translate_kernel<2>;
        case 8: return sr_grad_translate_kernel<3>;
        default: return sr_grad_translate_kernel<0>;
    }
}

// ---- loss terms (reporting only): out[b] = {df, tv, l2, l1} in float64 -------------------------
__global__ __launch_bounds__(256) void sr_loss_df_kernel(const float* __restrict__ resid, double* __restrict__ out,
                                                         int64_t per_image) {
    const int b = blockIdx.y;
    const float* r = resid + (int64_t)b * per_image;
    double acc = 0.0;
    for (int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x; p < per_image; p += (int64_t)gridDim.x * 256) {
        const float t = r[p];
        acc += (double)(t * t);
    }
    acc = asr_wave_sum(acc);
    if ((threadIdx.x & 63) == 0) atomicAdd(out + (int64_t)b * 4 + 0, acc);
}

__global__ __launch_bounds__(256) void sr_loss_prior_kernel(const float* __restrict__ x, double* __restrict__ out,
                                                            int H, int W, SrStep st) {
    const int b = blockIdx.y;
    const float* img = x + (int64_t)b * H * W;
    double tv = 0.0, l2 = 0.0, l1 = 0.0;
    const int64_t total = (int64_t)H * W;
    for (int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x; p < total; p += (int64_t)gridDim.x * 256) {
        const int X = (int)(p % W), Y = (int)(p / W);
        const float xc = img[p];
        if (st.prior == ASR_PRIOR_BTV) {
            for (int hh = -st.btv_shift; hh <= st.btv_shift; ++hh)
                for (int vv = 0; vv <= st.btv_shift; ++vv) {
                    const int xs = X - hh, ys = Y - vv;
                    const float sv = (xs >= 0 && xs < W && ys >= 0 && ys < H) ? img[(int64_t)ys * W + xs] : 0.0f;
                    tv += (double)st.btv_w[abs(hh) + vv] * (double)fabsf(xc - sv);
                }
        } else {
            const float dy = (Y < H - 1) ? img[p + W] - xc : 0.0f;
            const float dx = (X < W - 1) ? img[p + 1] - xc : 0.0f;
            tv += (double)(fabsf(dy) + fabsf(dx));
        }
        l2 += (double)(xc * xc);
        l1 += (double)fabsf(xc);
    }
    tv = asr_wave_sum(tv);
    l2 = asr_wave_sum(l2);
    l1 = asr_wave_sum(l1);
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(out + (int64_t)b * 4 + 1, tv);
        atomicAdd(out + (int64_t)b * 4 + 2, l2);
        atomicAdd(out + (int64_t)b * 4 + 3, l1);
    }
}

// ---- realign: max / mean over copies of InvWarp(upsample(y_n)) (superresolution.py:139-161) --
// MODE 0: mean -> out_a; 1: max -> out_a; 2: both in one pass (max -> out_a, mean -> out_b): the reference calls
// max_superresolution and mean_superresolution on the same copies (SR_single_class.py:103-110), and the per-copy value
// is the expensive part.
template <int MODE>
__global__ __launch_bounds__(256) void sr_realign_kernel(const float* __restrict__ y, float* __restrict__ out_a,
                                                         float* __restrict__ out_b,
                                                         const float* __restrict__ trans_tf /* translate(-s) */,
                                                         const float* __restrict__ rot_tf /* rotate(-theta) */,
                                                         SrDims d, float scale_y, float scale_x) {
    const int X = blockIdx.x * kTileX + threadIdx.x;
    const int Y = blockIdx.y * kTileY + threadIdx.y;
    const int b = blockIdx.z;
    if (X >= d.W || Y >= d.H) return;
    const int H = d.H, W = d.W, lh = d.h, lw = d.w;
    float acc_sum = 0.0f, acc_max = 0.0f;
    for (int n = 0; n < d.n; ++n) {
        const int bn = b * d.n + n;
        const float* src = y + (int64_t)bn * lh * lw;
        const AsrTf8 tt = asr_load_tf(trans_tf + (int64_t)bn * 8);
        const AsrTf8 tr = asr_load_tf(rot_tf + (int64_t)bn * 8);
        auto rd_up = [&](int yu, int xu) -> float {  // tf.image.resize(y_n, (H,W)) at integer (yu,xu)
            if (!(yu >= 0 && yu < H && xu >= 0 && xu < W)) return 0.0f;
            const AsrLerp ly = asr_half_pixel(yu, scale_y, lh);
            const AsrLerp lx = asr_half_pixel(xu, scale_x, lw);
            const float tl = src[ly.lo * lw + lx.lo], trv = src[ly.lo * lw + lx.hi];
            const float bl = src[ly.hi * lw + lx.lo], br = src[ly.hi * lw + lx.hi];
            const float top = tl + (trv - tl) * lx.t;
            const float bot = bl + (br - bl) * lx.t;
            return top + (bot - top) * ly.t;
        };
        auto rd_tr = [&](int yt, int xt) -> float {
            if (!(yt >= 0 && yt < H && xt >= 0 && xt < W)) return 0.0f;
            return asr_tf_sample(tt, rd_up, xt, yt);
        };
        float val;
        // Pure translation (always, for tfa.image.translate): the 2 x 2 translate-stage pixels under the rotation sample
        // read a 3 x 3 block of upsampled pixels (2 x 2 taps each, shifted by one); evaluate those 9 resize samples once
        // instead of 16 times.  Same per-tap arithmetic as the generic path (bit-identical); the generic path stays for
        // other transforms and for the float-rounding case where the two tap columns / rows do not abut.
        float ix, iy;
        const bool ok = asr_tf_map(tr, (float)X, (float)Y, ix, iy);
        const float xf = floorf(ix), yf = floorf(iy);
        const int x0 = asr_coord_to_int(xf), y0 = asr_coord_to_int(yf);
        const bool pure_translation = (tt.a0 == 1.0f) & (tt.a1 == 0.0f) & (tt.b0 == 0.0f) & (tt.b1 == 1.0f) &
                                      (tt.c0 == 0.0f) & (tt.c1 == 0.0f);
        const float jx0 = (float)x0 + tt.a2, jx1 = (float)(x0 + 1) + tt.a2;
        const float jy0 = (float)y0 + tt.b2, jy1 = (float)(y0 + 1) + tt.b2;
        const float fx0 = floorf(jx0), fx1 = floorf(jx1), fy0 = floorf(jy0), fy1 = floorf(jy1);
        const int cx0 = asr_coord_to_int(fx0), cx1 = asr_coord_to_int(fx1);
        const int cy0 = asr_coord_to_int(fy0), cy1 = asr_coord_to_int(fy1);
        if (ok && pure_translation && cx1 == cx0 + 1 && cy1 == cy0 + 1) {
            float uv[3][3];
#pragma unroll
            for (int a = 0; a < 3; ++a)
#pragma unroll
                for (int c = 0; c < 3; ++c) uv[a][c] = rd_up(cy0 + a, cx0 + c);
            const float wxl0 = (fx0 + 1.0f) - jx0, wxh0 = jx0 - fx0, wxl1 = (fx1 + 1.0f) - jx1, wxh1 = jx1 - fx1;
            const float wyl0 = (fy0 + 1.0f) - jy0, wyh0 = jy0 - fy0, wyl1 = (fy1 + 1.0f) - jy1, wyh1 = jy1 - fy1;
            auto Tq = [&](int a, int c, float wxl, float wxh, float wyl, float wyh) -> float {
                const float vyf = wxl * uv[a][c] + wxh * uv[a][c + 1];
                const float vyc = wxl * uv[a + 1][c] + wxh * uv[a + 1][c + 1];
                return wyl * vyf + wyh * vyc;
            };
            const bool vx0 = x0 >= 0 && x0 < W, vx1 = x0 + 1 >= 0 && x0 + 1 < W;
            const bool vy0 = y0 >= 0 && y0 < H, vy1 = y0 + 1 >= 0 && y0 + 1 < H;
            const float v00 = (vy0 && vx0) ? Tq(0, 0, wxl0, wxh0, wyl0, wyh0) : 0.0f;
            const float v01 = (vy0 && vx1) ? Tq(0, 1, wxl1, wxh1, wyl0, wyh0) : 0.0f;
            const float v10 = (vy1 && vx0) ? Tq(1, 0, wxl0, wxh0, wyl1, wyh1) : 0.0f;
            const float v11 = (vy1 && vx1) ? Tq(1, 1, wxl1, wxh1, wyl1, wyh1) : 0.0f;
            const float wxl = (xf + 1.0f) - ix, wxh = ix - xf;
            const float vyf = wxl * v00 + wxh * v01;
            const float vyc = wxl * v10 + wxh * v11;
            val = ((yf + 1.0f) - iy) * vyf + (iy - yf) * vyc;
        } else {
            val = asr_tf_sample(tr, rd_tr, X, Y);
        }
        if (MODE != 0) acc_max = (n == 0) ? val : fmaxf(acc_max, val);
        if (MODE != 1) acc_sum += val;
    }
    const int64_t o = ((int64_t)b * H + Y) * W + X;
    if (MODE == 0) out_a[o] = acc_sum / (float)d.n;
    if (MODE == 1) out_a[o] = acc_max;
    if (MODE == 2) { out_a[o] = acc_max; out_b[o] = acc_sum / (float)d.n; }
}

int check_dims(const char* fn, int batch, int n, int H, int W, int h, int w, SrDims* d) {
    ASR_REQUIRE(batch > 0 && n > 0 && H > 0 && W > 0 && h > 0 && w > 0, "%s: bad shape batch=%d n=%d %dx%d <- %dx%d",
                fn, batch, n, H, W, h, w);
    ASR_REQUIRE((int64_t)batch * n <= 65535, "%s: batch*n=%lld exceeds 65535 (grid.z)", fn, (long long)batch * n);
    const int f = H / h;
    ASR_UNSUPPORTED(f * h != H || f * w != W || f < 2 || (f & 1),
                    "%s: output %dx%d must be an even integer multiple of the feature size %dx%d", fn, H, W, h, w);
    d->batch = batch; d->n = n; d->H = H; d->W = W; d->h = h; d->w = w; d->f = f;
    return ASR_OK;
}

int prepare_backward(const SrDims& d) {
    const size_t lds = sizeof(float) * (size_t)d.n * 64;
    ASR_UNSUPPORTED(lds > 160 * 1024, "asr_sr: num_aug=%d needs %zu bytes of LDS in the backward kernel (max 640 copies)", d.n, lds);
    static AsrDeviceOnce once[4];
    const int idx = d.f == 2 ? 1 : (d.f == 4 ? 2 : (d.f == 8 ? 3 : 0));
    if (lds > 64 * 1024)
        ASR_HIP_CHECK(asr_allow_dynamic_lds(once[idx], reinterpret_cast<const void*>(sr_backward_kernel_for(d.f)), 160 * 1024));
    return ASR_OK;
}
dim3 gr_grid(const SrDims& d, int cn) {
    return dim3((unsigned)asr_cdiv(d.W, 64), (unsigned)asr_cdiv(d.H, 4 * kGrRows), (unsigned)(d.batch * cn));
}
// Copies whose gradient planes are alive at once.  The planes (1.07 MB each at 512 x 512) are written by K_gt and read
// back by K_bwd in the same iteration: 107 MB at N = 100, 214 MB at N = 200.  Cutting them into chunks of <= 32 copies
// (<= 35 MB live) was measured SLOWER on every shape and lane count (profiles/r03_sr_plane_chunk_experiment.txt: N = 100
// 99 -> 127 us per iteration, N = 200 314 -> 348 us, two solves in flight 614 -> 694 us): each extra launch pair costs
// more than the Infinity Cache misses it avoids, and the fabric counters see the planes either way.  The library default
// therefore keeps all copies in one chunk while the planes stay under kPlaneBytesMax per call and splits evenly beyond
// (a bound on the workspace, not a speed-up); cfg->plane_chunk overrides.
constexpr size_t kPlaneBytesMax = (size_t)1 << 30;
int sr_plane_chunk(int batch, int n, int H, int W, int requested) {
    if (requested > 0) return requested < n ? requested : n;
    const size_t per_copy = sizeof(float) * (size_t)batch * sr_gr_plane_elems(H, W);
    const size_t fit = kPlaneBytesMax / per_copy > 0 ? kPlaneBytesMax / per_copy : 1;
    if ((size_t)n <= fit) return n;
    const int chunks = (int)(((size_t)n + fit - 1) / fit);
    return (n + chunks - 1) / chunks;
}
size_t sr_workspace_bytes(int batch, int n, int H, int W, int h, int w, int chunk) {
    // residuals [batch*n, h, w] + the ping-pong x [batch, H, W] + the running data-term sum [batch, H, W] + the
    // zero-bordered planes [H + 4, W + 64]: G_R per copy of a chunk, x per image
    // + one int per image: "all inverse rotations affine"
    return sizeof(float) * ((size_t)batch * n * h * w + 2 * (size_t)batch * H * W +
                            ((size_t)batch * chunk + batch) * sr_gr_plane_elems(H, W)) + sizeof(int) * (size_t)batch;
}
dim3 gather_grid(const SrDims& d) {
    return dim3((unsigned)asr_cdiv(d.W, 64), (unsigned)asr_cdiv(d.H, 4), (unsigned)d.batch);
}
dim3 bwd_grid(const SrDims& d) { return dim3((unsigned)asr_cdiv(d.W, kBwdPixX), (unsigned)asr_cdiv(d.H, kBwdPixY), (unsigned)d.batch); }
const dim3 kBwdBlock(kBwdPixX, kBwdPixY, kBwdSplit);
size_t bwd_lds(const SrDims& d) { return sizeof(float) * (size_t)d.n * kBwdPix; }
dim3 hr_grid(const SrDims& d) { return dim3((unsigned)asr_cdiv(d.W, kTileX), (unsigned)asr_cdiv(d.H, kTileY), (unsigned)d.batch); }
dim3 lr_grid(const SrDims& d) { return dim3((unsigned)asr_cdiv(d.w, kTileX), (unsigned)asr_cdiv(d.h, kTileY), (unsigned)(d.batch * d.n)); }
const dim3 kBlock(kTileX, kTileY);

}  // namespace

extern "C" int asr_sr_init_target_f32(const float* y, float* x, int batch, int n, int H, int W, int h, int w,
                                      asr_stream_t stream) {
    ASR_REQUIRE(y && x, "asr_sr_init_target_f32: null pointer");
    SrDims d;
    ASR_REQUIRE(batch > 0 && n > 0 && H > 0 && W > 0 && h > 0 && w > 0, "asr_sr_init_target_f32: bad shape");
    d.batch = batch; d.n = n; d.H = H; d.W = W; d.h = h; d.w = w; d.f = 0;
    const float sy = (float)h / (float)H, sx = (float)w / (float)W;
    hipLaunchKernelGGL(sr_init_kernel, hr_grid(d), kBlock, 0, asr_stream(stream), y, x, d, sy, sx);
    ASR_LAUNCH_CHECK();
    return ASR_OK;
}

extern "C" int asr_sr_forward_residual_f32(const float* x, const float* y, const float* rot_tf, const float* trans_tf,
                                           float* resid, int batch, int n, int H, int W, int h, int w,
                                           asr_stream_t stream) {
    ASR_REQUIRE(x && y && rot_tf && trans_tf && resid, "asr_sr_forward_residual_f32: null pointer");
    SrDims d;
    int rc = check_dims("asr_sr_forward_residual_f32", batch, n, H, W, h, w, &d);
    if (rc != ASR_OK) return rc;
    hipLaunchKernelGGL(sr_forward_residual_kernel<false>, lr_grid(d), kBlock, 0, asr_stream(stream), x, y, rot_tf, trans_tf,
                       resid, d);
    ASR_LAUNCH_CHECK();
    return ASR_OK;
}

namespace {

// asr_sr_config -> kernel argument; validates the host-side choices
int make_step(const char* fn, const asr_sr_config* cfg, bool need_update, const float* m, const float* v, const float* vhat,
              SrStep* st) {
    ASR_REQUIRE(cfg, "%s: null config", fn);
    ASR_REQUIRE(cfg->optimizer >= ASR_OPT_ADAM && cfg->optimizer <= ASR_OPT_ADAMAX, "%s: unknown optimizer %d", fn, cfg->optimizer);
    ASR_REQUIRE(cfg->prior == ASR_PRIOR_TV || cfg->prior == ASR_PRIOR_BTV, "%s: unknown prior %d", fn, cfg->prior);
    st->optimizer = cfg->optimizer; st->flag = cfg->flag; st->c0 = cfg->c0; st->c1 = cfg->c1; st->c2 = cfg->c2;
    st->prior = cfg->prior; st->btv_shift = 0;
    for (int k = 0; k < 9; ++k) st->btv_w[k] = 0.0f;
    if (cfg->prior == ASR_PRIOR_BTV) {
        ASR_REQUIRE(cfg->btv_shift >= 1 && cfg->btv_shift <= 4, "%s: btv_shift %d outside [1, 4]", fn, cfg->btv_shift);
        st->btv_shift = cfg->btv_shift;
        for (int k = 0; k <= 2 * cfg->btv_shift; ++k) st->btv_w[k] = powf(cfg->btv_alpha, (float)k);   // tf.pow in float32
    }
    if (need_update) {
        bool ok = true;
        switch (cfg->optimizer) {
            case ASR_OPT_ADAM: ok = m && v && (vhat || !cfg->flag); break;
            case ASR_OPT_SGD: ok = m || cfg->c0 == 0.0f; break;
            case ASR_OPT_ADAGRAD: ok = v != nullptr; break;
            default: ok = m && v; break;
        }
        ASR_REQUIRE(ok, "%s: optimizer %d needs slot buffers that were not given", fn, cfg->optimizer);
    }
    return ASR_OK;
}

asr_sr_config adam_tv_config(float one_minus_beta1, float one_minus_beta2, float epsilon, int amsgrad) {
    asr_sr_config c{};
    c.optimizer = ASR_OPT_ADAM; c.flag = amsgrad; c.c0 = one_minus_beta1; c.c1 = one_minus_beta2; c.c2 = epsilon;
    c.prior = ASR_PRIOR_TV;
    c.plane_chunk = 0;
    return c;
}

}  // namespace

extern "C" int asr_sr_backward_cfg_f32(const float* x, float* x_new, const float* resid, const float* inv_rot_tf,
                                       const float* inv_trans_tf, float* m, float* v, float* vhat, const float* alphas,
                                       float* grad_out, int batch, int n, int H, int W, int h, int w, float lambda_df,
                                       float lambda_tv, float lambda_l2, float lambda_l1, const asr_sr_config* cfg,
                                       asr_stream_t stream) {
    ASR_REQUIRE(x && resid && inv_rot_tf && inv_trans_tf, "asr_sr_backward_cfg_f32: null pointer");
    ASR_REQUIRE(x_new || grad_out, "asr_sr_backward_cfg_f32: need x_new and/or grad_out");
    ASR_REQUIRE(!x_new || alphas, "asr_sr_backward_cfg_f32: alphas required with x_new");
    ASR_REQUIRE(x != x_new, "asr_sr_backward_cfg_f32: x_new must not alias x (the priors read neighbours)");
    SrDims d;
    int rc = check_dims("asr_sr_backward_cfg_f32", batch, n, H, W, h, w, &d);
    if (rc != ASR_OK) return rc;
    SrStep st;
    rc = make_step("asr_sr_backward_cfg_f32", cfg, x_new != nullptr, m, v, vhat, &st);
    if (rc != ASR_OK) return rc;
    rc = prepare_backward(d);
    if (rc != ASR_OK) return rc;
    hipLaunchKernelGGL(sr_backward_kernel_for(d.f), bwd_grid(d), kBwdBlock, bwd_lds(d), asr_stream(stream), x, x_new, resid, inv_rot_tf,
                       inv_trans_tf, m, v, vhat, alphas, grad_out, d, 2.0f * lambda_df, lambda_tv, 2.0f * lambda_l2,
                       lambda_l1, st);
    ASR_LAUNCH_CHECK();
    return ASR_OK;
}

extern "C" int asr_sr_backward_adam_f32(const float* x, float* x_new, const float* resid, const float* inv_rot_tf,
                                        const float* inv_trans_tf, float* m, float* v, float* vhat,
                                        const float* alphas, float* grad_out, int batch, int n, int H, int W, int h,
                                        int w, float lambda_df, float lambda_tv, float lambda_l2, float lambda_l1,
                                        float one_minus_beta1, float one_minus_beta2, float epsilon, int amsgrad,
                                        asr_stream_t stream) {
    const asr_sr_config c = adam_tv_config(one_minus_beta1, one_minus_beta2, epsilon, amsgrad);
    return asr_sr_backward_cfg_f32(x, x_new, resid, inv_rot_tf, inv_trans_tf, m, v, vhat, alphas, grad_out, batch, n, H, W,
                                   h, w, lambda_df, lambda_tv, lambda_l2, lambda_l1, &c, stream);
}

extern "C" int asr_sr_loss_terms_cfg_f64(const float* x, const float* resid, double* terms, int batch, int n, int H, int W,
                                         int h, int w, const asr_sr_config* cfg, asr_stream_t stream) {
    ASR_REQUIRE(x && resid && terms, "asr_sr_loss_terms_cfg_f64: null pointer");
    ASR_REQUIRE(batch > 0 && batch <= 65535, "asr_sr_loss_terms_cfg_f64: bad batch %d", batch);
    SrStep st;
    int rc = make_step("asr_sr_loss_terms_cfg_f64", cfg, false, nullptr, nullptr, nullptr, &st);
    if (rc != ASR_OK) return rc;
    hipStream_t s = asr_stream(stream);
    ASR_HIP_CHECK(hipMemsetAsync(terms, 0, sizeof(double) * 4 * batch, s));
    const int64_t per_image = (int64_t)n * h * w;
    hipLaunchKernelGGL(sr_loss_df_kernel, dim3(64, batch), dim3(256), 0, s, resid, terms, per_image);
    ASR_LAUNCH_CHECK();
    hipLaunchKernelGGL(sr_loss_prior_kernel, dim3(64, batch), dim3(256), 0, s, x, terms, H, W, st);
    ASR_LAUNCH_CHECK();
    return ASR_OK;
}

extern "C" int asr_sr_loss_terms_f64(const float* x, const float* resid, double* terms, int batch, int n, int H, int W,
                                     int h, int w, asr_stream_t stream) {
    const asr_sr_config c = adam_tv_config(0.f, 0.f, 0.f, 0);
    return asr_sr_loss_terms_cfg_f64(x, resid, terms, batch, n, H, W, h, w, &c, stream);
}

extern "C" size_t asr_sr_solve_workspace_bytes(int batch, int n, int H, int W, int h, int w) {
    return sr_workspace_bytes(batch, n, H, W, h, w, sr_plane_chunk(batch, n, H, W, 0));
}

extern "C" size_t asr_sr_solve_workspace_bytes_cfg(int batch, int n, int H, int W, int h, int w, const asr_sr_config* cfg) {
    return sr_workspace_bytes(batch, n, H, W, h, w, sr_plane_chunk(batch, n, H, W, cfg ? cfg->plane_chunk : 0));
}

// The whole optimisation loop of augmented_superresolution (superresolution.py:120-135) as one
// host call: num_iter x {K_fwd, K_bwd}, ping-ponging x between the caller's buffer and the
// workspace.  alphas[it*batch + b] = the per-step scalar of image b at its own global optimiser
// step (device array, prepared by the host wrapper so that the schedule and the persistent step
// counter follow optimizer.py exactly).
extern "C" int asr_sr_solve_cfg_f32(float* x, const float* y, const float* rot_tf, const float* trans_tf,
                                    const float* inv_rot_tf, const float* inv_trans_tf, float* m, float* v, float* vhat,
                                    const float* alphas, int num_iter, double* last_loss_terms, void* workspace,
                                    size_t workspace_bytes, int batch, int n, int H, int W, int h, int w,
                                    float lambda_df, float lambda_tv, float lambda_l2, float lambda_l1,
                                    const asr_sr_config* cfg, asr_stream_t stream) {
    ASR_REQUIRE(x && y && rot_tf && trans_tf && inv_rot_tf && inv_trans_tf && alphas && workspace,
                "asr_sr_solve_cfg_f32: null pointer");
    ASR_REQUIRE(num_iter >= 0, "asr_sr_solve_cfg_f32: num_iter < 0");
    SrDims d;
    int rc = check_dims("asr_sr_solve_cfg_f32", batch, n, H, W, h, w, &d);
    if (rc != ASR_OK) return rc;
    SrStep st;
    rc = make_step("asr_sr_solve_cfg_f32", cfg, true, m, v, vhat, &st);
    if (rc != ASR_OK) return rc;
    ASR_REQUIRE(cfg->plane_chunk >= 0, "asr_sr_solve_cfg_f32: plane_chunk %d < 0", cfg->plane_chunk);
    const int chunk = sr_plane_chunk(batch, n, H, W, cfg->plane_chunk);
    ASR_REQUIRE((int64_t)batch * chunk <= 65535, "asr_sr_solve_cfg_f32: batch*chunk=%lld exceeds 65535 (grid.z)", (long long)batch * chunk);
    const size_t need = sr_workspace_bytes(batch, n, H, W, h, w, chunk);
    if (workspace_bytes < need) {
        asr_set_error("asr_sr_solve_cfg_f32: workspace %zu < required %zu bytes", workspace_bytes, need);
        return ASR_ERR_WORKSPACE;
    }
    hipStream_t s = asr_stream(stream);
    float* resid = static_cast<float*>(workspace);
    float* x_alt = resid + (size_t)batch * n * h * w;
    float* cur = x;
    float* nxt = x_alt;
    float* const acc = x_alt + (size_t)batch * H * W;   // running data-term sum between the chunks of an iteration
    float* const gr = acc + (size_t)batch * H * W;
    const SrGradTranslateKernel gr_kernel = sr_grad_translate_kernel_for(d.f);
    float* const xb = gr + (size_t)batch * chunk * sr_gr_plane_elems(H, W);   // bordered copy of the current x
    int* const affine_flags = reinterpret_cast<int*>(xb + (size_t)batch * sr_gr_plane_elems(H, W));
    if (num_iter > 0) {
        hipLaunchKernelGGL(sr_affine_flags_kernel, dim3((unsigned)batch), dim3(64), 0, s, inv_rot_tf, affine_flags, n);
        ASR_LAUNCH_CHECK();
    }
    if (num_iter > 0) {   // the borders stay zero for the whole solve; the interiors are rewritten every iteration
        const size_t pe = sr_gr_plane_elems(H, W);
        ASR_HIP_CHECK(hipMemsetAsync(gr, 0, sizeof(float) * ((size_t)batch * chunk + batch) * pe, s));
        const size_t wp = (size_t)W + 2 * kGrPadX;
        for (int b = 0; b < batch; ++b)
            ASR_HIP_CHECK(hipMemcpy2DAsync(xb + b * pe + kGrPadY * wp + kGrPadX, wp * sizeof(float), x + (size_t)b * H * W,
                                           (size_t)W * sizeof(float), (size_t)W * sizeof(float), (size_t)H,
                                           hipMemcpyDeviceToDevice, s));
    }
    for (int it = 0; it < num_iter; ++it) {
        hipLaunchKernelGGL(sr_forward_residual_kernel<true>, lr_grid(d), kBlock, 0, s, xb, y, rot_tf, trans_tf, resid, d);
        ASR_LAUNCH_CHECK();
        if (last_loss_terms && it == num_iter - 1) {
            rc = asr_sr_loss_terms_cfg_f64(cur, resid, last_loss_terms, batch, n, H, W, h, w, cfg, stream);
            if (rc != ASR_OK) return rc;
        }
        for (int n0 = 0; n0 < n; n0 += chunk) {
            const int cn = n - n0 < chunk ? n - n0 : chunk;
            hipLaunchKernelGGL(gr_kernel, gr_grid(d, cn), dim3(64, 4), 0, s, resid, inv_trans_tf, gr, d, 2.0f * lambda_df, n0, cn);
            ASR_LAUNCH_CHECK();
            hipLaunchKernelGGL(sr_backward_gather_kernel, gather_grid(d), dim3(64, 4), 0, s, cur, nxt, gr, inv_rot_tf, m, v, vhat,
                               alphas + (size_t)it * batch, (float*)nullptr, d, lambda_tv, 2.0f * lambda_l2, lambda_l1, st, xb,
                               acc, n0, cn, affine_flags);
            ASR_LAUNCH_CHECK();
        }
        float* t = cur; cur = nxt; nxt = t;
    }
    if (cur != x) ASR_HIP_CHECK(hipMemcpyAsync(x, cur, sizeof(float) * (size_t)batch * H * W, hipMemcpyDeviceToDevice, s));
    return ASR_OK;
}

extern "C" int asr_sr_solve_f32(float* x, const float* y, const float* rot_tf, const float* trans_tf,
                                const float* inv_rot_tf, const float* inv_trans_tf, float* m, float* v, float* vhat,
                                const float* alphas, int num_iter, double* last_loss_terms, void* workspace,
                                size_t workspace_bytes, int batch, int n, int H, int W, int h, int w,
                                float lambda_df, float lambda_tv, float lambda_l2, float lambda_l1,
                                float one_minus_beta1, float one_minus_beta2, float epsilon, int amsgrad,
                                asr_stream_t stream) {
    const asr_sr_config c = adam_tv_config(one_minus_beta1, one_minus_beta2, epsilon, amsgrad);
    return asr_sr_solve_cfg_f32(x, y, rot_tf, trans_tf, inv_rot_tf, inv_trans_tf, m, v, vhat, alphas, num_iter,
                                last_loss_terms, workspace, workspace_bytes, batch, n, H, W, h, w, lambda_df, lambda_tv,
                                lambda_l2, lambda_l1, &c, stream);
}

static int realign_common(int mode, const float* y, float* out_a, float* out_b, const float* trans_tf, const float* rot_tf,
                          int batch, int n, int H, int W, int h, int w, asr_stream_t stream) {
    ASR_REQUIRE(y && out_a && trans_tf && rot_tf && (mode != 2 || out_b), "asr_realign: null pointer");
    ASR_REQUIRE(batch > 0 && batch <= 65535 && n > 0 && H > 0 && W > 0 && h > 0 && w > 0, "asr_realign: bad shape");
    SrDims d;
    d.batch = batch; d.n = n; d.H = H; d.W = W; d.h = h; d.w = w; d.f = 0;
    const float sy = (float)h / (float)H, sx = (float)w / (float)W;
    hipStream_t s = asr_stream(stream);
    if (mode == 1) hipLaunchKernelGGL(sr_realign_kernel<1>, hr_grid(d), kBlock, 0, s, y, out_a, out_b, trans_tf, rot_tf, d, sy, sx);
    else if (mode == 0) hipLaunchKernelGGL(sr_realign_kernel<0>, hr_grid(d), kBlock, 0, s, y, out_a, out_b, trans_tf, rot_tf, d, sy, sx);
    else hipLaunchKernelGGL(sr_realign_kernel<2>, hr_grid(d), kBlock, 0, s, y, out_a, out_b, trans_tf, rot_tf, d, sy, sx);
    ASR_LAUNCH_CHECK();
    return ASR_OK;
}

extern "C" int asr_realign_max_f32(const float* y, float* out, const float* trans_tf, const float* rot_tf, int batch,
                                   int n, int H, int W, int h, int w, asr_stream_t stream) {
    return realign_common(1, y, out, nullptr, trans_tf, rot_tf, batch, n, H, W, h, w, stream);
}

extern "C" int asr_realign_mean_f32(const float* y, float* out, const float* trans_tf, const float* rot_tf, int batch,
                                    int n, int H, int W, int h, int w, asr_stream_t stream) {
    return realign_common(0, y, out, nullptr, trans_tf, rot_tf, batch, n, H, W, h, w, stream);
}

extern "C" int asr_realign_max_mean_f32(const float* y, float* out_max, float* out_mean, const float* trans_tf,
                                        const float* rot_tf, int batch, int n, int H, int W, int h, int w,
                                        asr_stream_t stream) {
    return realign_common(2, y, out_max, out_mean, trans_tf, rot_tf, batch, n, H, W, h, w, stream);
}

#ifdef ASR_DIAG_KFWD_CHECK
// diag builds only: copy out (and clear) K_fwd's self-check counters [128] and records [16][12]
extern "C" int asr_diag_kfwd_read(unsigned* counters, float* records) {
    ASR_HIP_CHECK(hipDeviceSynchronize());
    ASR_HIP_CHECK(hipMemcpyFromSymbol(counters, HIP_SYMBOL(g_kfwd_cnt), sizeof(unsigned) * 128));
    ASR_HIP_CHECK(hipMemcpyFromSymbol(records, HIP_SYMBOL(g_kfwd_rec), sizeof(float) * 16 * 12));
    static unsigned zero[128];
    ASR_HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(g_kfwd_cnt), zero, sizeof(zero)));
    return ASR_OK;
}
#endif
