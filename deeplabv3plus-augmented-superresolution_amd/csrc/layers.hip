// Remaining DeepLabV3+ layers as NHWC float32 kernels (gfx950):
//   asr_conv3x3_direct_f32   dense 3x3 for tiny Cin (entry_flow_conv1_1: 3 -> 32, stride 2, model.py:150-151)
//   asr_conv3x3_stem_f16x3   the same layer as a split-f16 MFMA implicit GEMM (precision f16x3)
//   asr_gap_f32              GlobalAveragePooling2D(keepdims) (model.py:196-197)
//   asr_resize_bilinear_f32  Resizing(bilinear) = tf.image.resize half-pixel (model.py:109-110,204-205,241-242)
// All HBM-bound; lanes own 4 consecutive channels so stores are whole 16-byte pieces of a pixel row.
#include "asr_common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

// ---- dense 3x3, small cin: weights [3][3][cin][cout] in LDS, thread = (pixel, 4 output channels) ----
__global__ __launch_bounds__(256) void conv3x3_direct_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                             const float* __restrict__ bias, float* __restrict__ y,
                                                             int batch, int h_in, int w_in, int cin, int cout, int stride,
                                                             int pad_top, int pad_left, int h_out, int w_out, int ldx,
                                                             int ldy, int relu) {
    extern __shared__ __attribute__((aligned(16))) float sw[];  // 9*cin*cout
    const int nw = 9 * cin * cout;
    for (int i = threadIdx.x; i < nw; i += 256) sw[i] = w[i];
    __syncthreads();
    const int co4n = cout >> 2;
    const long long total = (long long)batch * h_out * w_out * co4n;
    for (long long o = (long long)blockIdx.x * 256 + threadIdx.x; o < total; o += (long long)gridDim.x * 256) {
        const int co4 = (int)(o % co4n);
        long long t = o / co4n;
        const int ox = (int)(t % w_out); t /= w_out;
        const int oy = (int)(t % h_out);
        const long long b = t / h_out;
        const float* xin = x + b * h_in * w_in * ldx;
        f32x4 acc = *reinterpret_cast<const f32x4*>(bias + co4 * 4);
        for (int ky = 0; ky < 3; ++ky) {
            const int iy = oy * stride - pad_top + ky;
            if (iy < 0 || iy >= h_in) continue;
            for (int kx = 0; kx < 3; ++kx) {
                const int ix = ox * stride - pad_left + kx;
                if (ix < 0 || ix >= w_in) continue;
                const float* px = xin + ((long long)iy * w_in + ix) * ldx;
                const float* pw = sw + ((ky * 3 + kx) * cin) * cout + co4 * 4;
                for (int ci = 0; ci < cin; ++ci) {
                    const float v = px[ci];
                    acc += v * *reinterpret_cast<const f32x4*>(pw + ci * cout);
                }
            }
        }
        if (relu) { acc.x = fmaxf(acc.x, 0.f); acc.y = fmaxf(acc.y, 0.f); acc.z = fmaxf(acc.z, 0.f); acc.w = fmaxf(acc.w, 0.f); }
        if (relu == 2) { acc.x = fminf(acc.x, 6.f); acc.y = fminf(acc.y, 6.f); acc.z = fminf(acc.z, 6.f); acc.w = fminf(acc.w, 6.f); }
        *reinterpret_cast<f32x4*>(y + ((b * h_out + oy) * w_out + ox) * ldy + co4 * 4) = acc;
    }
}

// ---- the stem: dense 3x3 on RGB, 32 output channels -------------------------------------------------------------
// thread = (4 output channels, SP consecutive output pixels); the 8 lanes that share a pixel group each fetch 4 of a
// pixel's 27 input values (instead of all 27: the texture-address path, 27 dword loads per 16 output bytes, bounded the
// generic kernel at 1.0 TB/s) and hand them round with ds_bpermute; a weight quad read from LDS serves SP pixels.
constexpr int SP = 4;

__global__ __launch_bounds__(256) ASR_PK_F32 void conv3x3_stem_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                           const float* __restrict__ bias, float* __restrict__ y, int batch,
                                                           int h_in, int w_in, int stride, int pad_top, int pad_left, int h_out,
                                                           int w_out, int ldx, int ldy, int relu) {
    constexpr int COUT = 32, NE = 27;                        // 3 input channels x 9 taps
    __shared__ __attribute__((aligned(16))) float sw[NE * COUT];
    for (int i = threadIdx.x; i < NE * COUT; i += 256) sw[i] = w[i];
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int co4 = lane & 7;
    const int groups_x = (w_out + SP - 1) / SP;
    const long long total = (long long)batch * h_out * groups_x;          // pixel groups
    const f32x4 bv = *reinterpret_cast<const f32x4*>(bias + co4 * 4);
    // every lane of a wave takes part in the exchanges: the loop bound is wave-uniform, late groups are masked
    const long long waves_total = (total + 7) / 8;
    for (long long wv = (long long)blockIdx.x * 4 + (threadIdx.x >> 6); wv < waves_total; wv += (long long)gridDim.x * 4) {
        const long long g = wv * 8 + (lane >> 3);
        const bool g_ok = g < total;
        const long long gc = g_ok ? g : total - 1;
        const int gx = (int)(gc % groups_x);
        long long t = gc / groups_x;
        const int oy = (int)(t % h_out);
        const long long b = t / h_out;
        const float* xin = x + b * h_in * w_in * ldx;
        // slot s of lane co4 holds input element e = s * 8 + co4 of each of the SP pixels (e = (ky * 3 + kx) * 3 + ci)
        float val[SP][4];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int e = s * 8 + co4;
            const int tap = e / 3, ci = e - tap * 3, ky = tap / 3, kx = tap - ky * 3;
            const int iy = oy * stride - pad_top + ky;
#pragma unroll
            for (int j = 0; j < SP; ++j) {
                const int ox = gx * SP + j;
                const int ix = ox * stride - pad_left + kx;
                const bool in = e < NE && iy >= 0 && iy < h_in && ix >= 0 && ix < w_in;
                const float v = xin[((long long)(in ? iy : 0) * w_in + (in ? ix : 0)) * ldx + (in ? ci : 0)];
                val[j][s] = in ? v : 0.0f;
            }
        }
        f32x4 acc[SP];
#pragma unroll
        for (int j = 0; j < SP; ++j) acc[j] = bv;
        const int grp_base = lane & ~7;
#pragma unroll
        for (int e = 0; e < NE; ++e) {
            const f32x4 wq = *reinterpret_cast<const f32x4*>(sw + e * COUT + co4 * 4);
#pragma unroll
            for (int j = 0; j < SP; ++j) acc[j] += __shfl(val[j][e >> 3], grp_base | (e & 7), 64) * wq;
        }
        if (g_ok) {
#pragma unroll
            for (int j = 0; j < SP; ++j) {
                const int ox = gx * SP + j;
                if (ox < w_out) {
                    f32x4 a = acc[j];
                    if (relu) { a.x = fmaxf(a.x, 0.f); a.y = fmaxf(a.y, 0.f); a.z = fmaxf(a.z, 0.f); a.w = fmaxf(a.w, 0.f); }
                    if (relu == 2) { a.x = fminf(a.x, 6.f); a.y = fminf(a.y, 6.f); a.z = fminf(a.z, 6.f); a.w = fminf(a.w, 6.f); }
                    *reinterpret_cast<f32x4*>(y + ((b * h_out + oy) * w_out + ox) * ldy + co4 * 4) = a;
                }
            }
        }
    }
}

// ---- entry_flow_conv1_1 on the matrix cores (asr_conv3x3_stem_f16x3) ---------------------------------------------
// 3 -> 32 channels, 3 x 3: an implicit GEMM with K = 27 (padded to 32), N = 32.  One wave = 32 consecutive output
// pixels of a row: it gathers its A fragment straight from the image (lane = pixel, 16 of the 32 k slots each, scalar
// dword loads with clamped addresses + select -- the 27 inputs of a pixel are three 36-byte runs), splits it to
// hi/lo f16 in registers and issues 2 k-steps x {lo*hi, hi*lo, hi*hi} v_mfma_f32_32x32x16_f16; the weights' B fragments
// are split once per wave and stay in registers.  The accumulator has the 32 output channels on the lanes, so every
// store instruction writes two whole 128-byte pixel rows.  HBM-bound (3.15 MB in + 8.4 MB out per 512^2 image) where the
// VALU kernel above was bound by its 27 ds_bpermute + 108 FMA per 4 outputs.
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// 16 values -> hi / lo halves as two f16x8 each (element order preserved), in a kernel that has called
// asr_enable_f16_saturation(): packed saturating conversions instead of clamp + convert per value (asr_common.h)
__device__ __forceinline__ void split16_f16(const float (&a)[16], f16x8 (&hi)[2], f16x8 (&lo)[2]) {
    typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        u32x4_t h, l;
        unsigned int h0, h1, l0, l1;
        asr_split4_f16_saturating_mode(a[8 * s + 0], a[8 * s + 1], a[8 * s + 2], a[8 * s + 3], h0, h1, l0, l1);
        h.x = h0; h.y = h1; l.x = l0; l.y = l1;
        asr_split4_f16_saturating_mode(a[8 * s + 4], a[8 * s + 5], a[8 * s + 6], a[8 * s + 7], h0, h1, l0, l1);
        h.z = h0; h.w = h1; l.z = l0; l.w = l1;
        hi[s] = __builtin_bit_cast(f16x8, h);
        lo[s] = __builtin_bit_cast(f16x8, l);
    }
}


__global__ __launch_bounds__(256) ASR_PK_F32 void conv3x3_stem_mfma_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                                const float* __restrict__ bias, float* __restrict__ y, int batch,
                                                                int h_in, int w_in, int stride, int pad_top, int pad_left,
                                                                int h_out, int w_out, int ldx, int ldy, int relu) {
    asr_enable_f16_saturation();                              // the splits below convert with the hardware's f16 clamp
    constexpr int COUT = 32, NE = 27;
    const int lane = threadIdx.x & 63, l32 = lane & 31, hh = lane >> 5;
    // k slot (s, j) of this lane half holds input element e = 16 s + 8 hh + j, e = (ky * 3 + kx) * 3 + ci
    f16x8 bh[2], bl[2];
    int off[16], kyx[16];                                     // element offset from the window origin; ky | kx << 2 | valid << 4
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const int e = 16 * (k >> 3) + 8 * hh + (k & 7);
        const bool ok = e < NE;
        const int ec = ok ? e : 0;
        const int tap = ec / 3, ci = ec - tap * 3, ky = tap / 3, kx = tap - ky * 3;
        off[k] = (ky * w_in + kx) * ldx + ci;
        kyx[k] = ky | (kx << 2) | ((int)ok << 4);
        const float wv = ok ? w[ec * COUT + l32] : 0.0f;
        const _Float16 hi = (_Float16)wv;
        bh[k >> 3][k & 7] = hi;
        bl[k >> 3][k & 7] = (_Float16)(wv - (float)hi);
    }
    const float bv = bias[l32];
    const int groups_x = (w_out + 31) / 32;
    const long long total = (long long)batch * h_out * groups_x;
    for (long long g = (long long)blockIdx.x * 4 + (threadIdx.x >> 6); g < total; g += (long long)gridDim.x * 4) {
        const int gx = (int)(g % groups_x);
        const long long t = g / groups_x;
        const int oy = (int)(t % h_out);
        const long long b = t / h_out;
        const int ox = gx * 32 + l32;
        const int iy0 = oy * stride - pad_top, ix0 = ox * stride - pad_left;
        const float* xin = x + b * h_in * w_in * ldx;
        const long long org = ((long long)iy0 * w_in + ix0) * ldx;
        float a[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int iy = iy0 + (kyx[k] & 3), ix = ix0 + ((kyx[k] >> 2) & 3);
            const bool in = (kyx[k] & 16) && ox < w_out && iy >= 0 && iy < h_in && ix >= 0 && ix < w_in;
            const float v = xin[in ? org + off[k] : 0];
            a[k] = in ? v : 0.0f;
        }
        f16x8 ah[2], al[2];
        split16_f16(a, ah, al);
        f32x16 acc;
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] = 0.0f;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[s], bh[s], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[s], bl[s], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[s], bh[s], acc, 0, 0, 0);
        }
        // C/D map: column (output channel) = lane & 31, row (pixel) = (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5)
        float* yrow = y + ((b * h_out + oy) * w_out + (long long)gx * 32) * ldy + l32;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int r = (e & 3) + 8 * (e >> 2) + 4 * hh;
            float v = acc[e] + bv;
            if (relu) v = fmaxf(v, 0.f);
            if (relu == 2) v = fminf(v, 6.f);
            if (gx * 32 + r < w_out) yrow[(long long)r * ldy] = v;
        }
    }
}

// ---- entry_flow_conv1_1 + entry_flow_conv1_2 in one kernel (asr_entry_stem_f16x3) ---------------------------------
// y = relu(conv3x3_same(relu(conv3x3_s2_same(x, w1) + b1), w2) + b2): 3 -> 32 channels at stride 2, then 32 -> 64.
// A persistent workgroup (8 waves) owns 16 x 16 output pixels at a time:
//   stage 1  the 18 x 18 conv1_1 outputs under the tile (halo 1; positions outside the map are conv1_2's zero padding)
//            are computed on the matrix cores exactly like conv3x3_stem_mfma_kernel, but with the MFMA operands swapped
//            (weights as A) so that a lane holds one pixel and 4 consecutive channels per register quad, and are written
//            to LDS already split into hi/lo f16: one 128-byte line per pixel [hi c0-31 | lo c0-31] in 16-byte slots,
//            slot XOR (pixel >> 1) & 7;
//   stage 2  conv1_2 as an implicit GEMM whose A fragments come straight from that LDS image (tap (ky, kx) = the line of
//            pixel (y + ky, x + kx)) and whose weights (9 taps x 32 x 64, hi + lo: 72 KB) stay resident in LDS for the
//            kernel's lifetime: 108 MFMAs per wave and tile, no global A traffic at all.
// The 32-channel intermediate (839 MB per 100 copies, written once and re-read nine times through L2 by the implicit
// GEMM of the two-kernel form) never leaves the CU.
#ifndef ASR_DIAG_STEM_SKIP
#define ASR_DIAG_STEM_SKIP 0                                   // ablation bits of tools/build_hazard_variants.py ("stem_skip*"); 0 in the product
#endif
constexpr int ES_T = 16, ES_H = ES_T + 2, ES_NPIX = ES_H * ES_H, ES_GROUPS = (ES_NPIX + 31) / 32;
// Row stride of the LDS image in lines.  Stage 2 reads, per ds_read_b128 phase of 16 lanes, 8 lines of one tile row and 8 of
// the next; with the natural stride 18 two of the 16 land on the same bank group as two others (28 % of the LDS cycles
// were bank conflicts, profiles/r02_pmc_sq.json); a stride that is a multiple of 16 keeps the XOR swizzle's 16 residues
// distinct across the two rows.
constexpr int ES_LS = 32;
constexpr int ES_T1_BYTES = ES_H * ES_LS * 128, ES_B_BYTES = 36 * 64 * 16, ES_LDS_BYTES = ES_T1_BYTES + 2 * ES_B_BYTES;   // 144 KB
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(512) ASR_PK_F32 void entry_stem_fused_kernel(const float* __restrict__ x, const float* __restrict__ w1,
                                                               const float* __restrict__ b1, const _Float16* __restrict__ w2p,
                                                               const float* __restrict__ b2, float* __restrict__ y, int batch,
                                                               int h_in, int w_in, int h1, int w1d, int ldx, int ldy, int npad2) {
    asr_enable_f16_saturation();                              // the splits below convert with the hardware's f16 clamp
#ifdef ASR_DIAG_STEM_TOP_VGPR
    ASR_DIAG_TOUCH_VGPR(ASR_DIAG_STEM_TOP_VGPR);
#endif
    extern __shared__ __attribute__((aligned(16))) char es_lds[];
    char* const T1 = es_lds;
    char* const Bh = es_lds + ES_T1_BYTES;
    char* const Bl = Bh + ES_B_BYTES;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l32 = lane & 31, hh = lane >> 5;
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    {   // conv1_2's weights: packed planes [36 octets][npad2][8 halfs] (hi, then lo) -> LDS [36][64][8]
        const long long plane = (long long)288 * npad2;
        for (int i = tid; i < 36 * 64; i += 512) {
            const int oct = i >> 6, col = i & 63;
            const _Float16* src = w2p + ((long long)oct * npad2 + col) * 8;
            *reinterpret_cast<u32x4*>(Bh + i * 16) = *reinterpret_cast<const u32x4*>(src);
            *reinterpret_cast<u32x4*>(Bl + i * 16) = *reinterpret_cast<const u32x4*>(src + plane);
        }
    }
    // stage-1 constants: k slot (s, j) of this lane half = input element e = 16 s + 8 hh + j = (ky * 3 + kx) * 3 + ci
    f16x8 wh[2], wl[2];
    int off[16], kyx[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const int e = 16 * (k >> 3) + 8 * hh + (k & 7);
        const bool ok = e < 27;
        const int ec = ok ? e : 0;
        const int tap = ec / 3, ci = ec - tap * 3, ky = tap / 3, kx = tap - ky * 3;
        off[k] = (ky * w_in + kx) * ldx + ci;
        kyx[k] = ky | (kx << 2) | ((int)ok << 4);
        const float wv = ok ? w1[ec * 32 + l32] : 0.0f;
        const _Float16 hi = (_Float16)wv;
        wh[k >> 3][k & 7] = hi;
        wl[k >> 3][k & 7] = (_Float16)(wv - (float)hi);
    }
    float b1v[16];                                            // bias of the channel each accumulator register holds
#pragma unroll
    for (int e = 0; e < 16; ++e) b1v[e] = b1[(e & 3) + 8 * (e >> 2) + 4 * hh];
    const float b2v0 = b2[l32], b2v1 = b2[32 + l32];

    const int tiles_x = (w1d + ES_T - 1) / ES_T, tiles_y = (h1 + ES_T - 1) / ES_T;
    const long long total = (long long)batch * tiles_y * tiles_x;

    // Stage 1 of one 32-pixel group: gather() issues the lane's 16 image loads, finish() turns them into conv1_1 lines.
    // (Software-pipelining the two stages over the tiles -- the next tile's gather in flight under stage 2, a second
    // LDS image -- measured no faster: 982 vs 938 us; the tile time is set by stage 2's MFMAs and the 64 KB of stores.)
    struct Gathered {
        float a[16];
        int t, line;                                           // pixel of the 18 x 18 halo tile, its line in the LDS image
        bool in_map;
    };
    auto gather = [&](long long tile, int g) -> Gathered {
        Gathered r;
        const int tx0 = (int)(tile % tiles_x) * ES_T;
        const long long tt = tile / tiles_x;
        const int ty0 = (int)(tt % tiles_y) * ES_T;
        const float* xin = x + (tt / tiles_y) * h_in * w_in * ldx;
        r.t = 32 * g + l32;
        const int tyy = r.t / ES_H, txx = r.t - tyy * ES_H;
        r.line = tyy * ES_LS + txx;
        const int oy1 = ty0 - 1 + tyy, ox1 = tx0 - 1 + txx;
        r.in_map = r.t < ES_NPIX && oy1 >= 0 && oy1 < h1 && ox1 >= 0 && ox1 < w1d;
        const int iy0 = 2 * oy1, ix0 = 2 * ox1;                // 'same' at stride 2 on an even input: pad bottom / right only
        const long long org = ((long long)iy0 * w_in + ix0) * ldx;
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int iy = iy0 + (kyx[k] & 3), ix = ix0 + ((kyx[k] >> 2) & 3);
            const bool in = (kyx[k] & 16) && r.in_map && iy < h_in && ix < w_in;
#if ASR_DIAG_STEM_SKIP & 1                                      // ablation: no image loads
            const float v = 0.25f * (float)k;
#else
            const float v = xin[in ? org + off[k] : 0];
#endif
            r.a[k] = in ? v : 0.0f;
        }
        return r;
    };
    auto finish = [&](const Gathered& r, char* T1) {
        f16x8 xh[2], xl[2];
        split16_f16(r.a, xh, xl);
        f32x16 acc;
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] = 0.0f;
#pragma unroll
        for (int s = 0; s < 2; ++s) {                          // D[channel][pixel]: weights as the A operand
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[s], xl[s], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl[s], xh[s], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[s], xh[s], acc, 0, 0, 0);
        }
        if (r.t < ES_NPIX) {
            char* const line = T1 + r.line * 128;
            const int swz = (r.line >> 1) & 7;
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {                   // registers 4 g4 .. 4 g4 + 3 = channels 8 g4 + 4 hh + 0..3
                float v[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] = r.in_map ? fmaxf(acc[4 * g4 + i] + b1v[4 * g4 + i], 0.0f) : 0.0f;
                typedef unsigned int u32x2_t __attribute__((ext_vector_type(2)));
                u32x2_t hi, lo;
                unsigned int h0, h1, l0, l1;
                asr_split4_f16_saturating_mode(v[0], v[1], v[2], v[3], h0, h1, l0, l1);
                hi.x = h0; hi.y = h1; lo.x = l0; lo.y = l1;
                *reinterpret_cast<u32x2_t*>(line + ((g4 ^ swz) << 4) + hh * 8) = hi;
                *reinterpret_cast<u32x2_t*>(line + (((4 + g4) ^ swz) << 4) + hh * 8) = lo;
            }
        }
    };
    const bool two_groups = wave + 8 < ES_GROUPS;              // waves 0..2 own a second group of the 11

    for (long long tile = blockIdx.x; tile < total; tile += gridDim.x) {
        __syncthreads();                                       // the previous tile's stage 2 is done with the LDS image
        // ---- stage 1: conv1_1 (+ bias, ReLU) of the 18 x 18 halo tile -> split-f16 lines in LDS ----
#if ASR_DIAG_STEM_SKIP & 8                                      // ablation: no stage 1 at all (stage 2 reads whatever the LDS holds)
        if (tile < 0) finish(gather(tile, wave), T1);
#else
        finish(gather(tile, wave), T1);
        if (two_groups) finish(gather(tile, wave + 8), T1);
#endif
        __syncthreads();
        const int tx0 = (int)(tile % tiles_x) * ES_T;
        const long long tt = tile / tiles_x;
        const int ty0 = (int)(tt % tiles_y) * ES_T;
        const long long b = tt / tiles_y;
        // ---- stage 2: conv1_2 from the LDS image; this wave's 32 output pixels = 2 rows x 16 columns ----
        const int o = 32 * wave + l32, yy = o >> 4, xx = o & 15;
        f32x16 acc2[2];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc2[j][e] = 0.0f;
#pragma unroll 3                                               // fully unrolled, the hoisted LDS reads spill
        for (int tap = 0; tap < ((ASR_DIAG_STEM_SKIP & 2) ? 1 : 9); ++tap) {      // (ablation bit 2: one tap instead of nine)
            const int ky = (tap * 11) >> 5;                    // tap / 3 for tap < 9
            const int tp = (yy + ky) * ES_LS + xx + (tap - 3 * ky);
            const char* const line = T1 + tp * 128;
            const int swz = (tp >> 1) & 7;
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                const int oct = 2 * kk + hh;
                const f16x8 ah = *reinterpret_cast<const f16x8*>(line + ((oct ^ swz) << 4));
                const f16x8 al = *reinterpret_cast<const f16x8*>(line + (((4 + oct) ^ swz) << 4));
                const int og = tap * 4 + oct;
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const f16x8 bh = *reinterpret_cast<const f16x8*>(Bh + (og * 64 + j * 32 + l32) * 16);
                    const f16x8 bl = *reinterpret_cast<const f16x8*>(Bl + (og * 64 + j * 32 + l32) * 16);
                    acc2[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc2[j], 0, 0, 0);
                    acc2[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc2[j], 0, 0, 0);
                    acc2[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc2[j], 0, 0, 0);
                }
            }
        }
        float* const ybase = y + (b * h1 * w1d) * ldy + l32;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int r = 32 * wave + (e & 3) + 8 * (e >> 2) + 4 * hh;       // pixel of accumulator row e
            const int oy = ty0 + (r >> 4), ox = tx0 + (r & 15);
            if (oy < h1 && ox < w1d && !((ASR_DIAG_STEM_SKIP & 4) && acc2[0][e] != 12345.678f)) {      // (ablation bit 4: no stores)
                float* const q = ybase + ((long long)oy * w1d + ox) * ldy;
                q[0] = fmaxf(acc2[0][e] + b2v0, 0.0f);
                q[32] = fmaxf(acc2[1][e] + b2v1, 0.0f);
            }
        }
    }
}

// ---- global average pool: block = 64 channel-quads x 4 pixel groups ----------------------------
__global__ __launch_bounds__(256) void gap_kernel(const float* __restrict__ x, float* __restrict__ y, int hw, int c, int ldx) {
    __shared__ f32x4 part[4][64];
    const int lane = threadIdx.x & 63, grp = threadIdx.x >> 6;
    const int c4 = blockIdx.x * 64 + lane;
    const int b = blockIdx.y;
    const bool ok = c4 * 4 < c;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    if (ok) {
        const float* p = x + (long long)b * hw * ldx + c4 * 4;
        for (int i = grp; i < hw; i += 4) acc += *reinterpret_cast<const f32x4*>(p + (long long)i * ldx);
    }
    part[grp][lane] = acc;
    __syncthreads();
    if (grp == 0 && ok) {
        f32x4 s = part[0][lane] + part[1][lane];
        s += part[2][lane];
        s += part[3][lane];
        const float inv = 1.0f / (float)hw;
        *reinterpret_cast<f32x4*>(y + (long long)b * c + c4 * 4) = s * inv;
    }
}

// ---- bilinear resize, TF2 half-pixel centres ------------------------------------------------------
// grid = (ceil(w_out * c / 4 / 256), h_out, batch): the output row and image come from the block index (the generic
// flat index cost three 64-bit divisions per 16 output bytes -- the kernel was VALU-bound at 3 TB/s on the decoder's
// 32^2 -> 128^2 x 256-channel upsample), the row's y taps are block-uniform.
__global__ __launch_bounds__(256) void resize_bilinear_kernel(const float* __restrict__ x, float* __restrict__ y, int batch,
                                                              int h_in, int w_in, int c, int h_out, int w_out, int ldx,
                                                              int ldy, float scale_y, float scale_x) {
    const unsigned c4n = (unsigned)c >> 2;
    const unsigned idx = blockIdx.x * 256u + threadIdx.x;
    const unsigned ox = idx / c4n, c4 = idx - ox * c4n;
    if (ox >= (unsigned)w_out) return;
    const int oy = blockIdx.y;
    const long long b = blockIdx.z;
    const float py = ((float)oy + 0.5f) * scale_y - 0.5f, px = ((float)ox + 0.5f) * scale_x - 0.5f;
    const float fy = floorf(py), fx = floorf(px);
    const int ylo = max((int)fy, 0), yhi = min((int)ceilf(py), h_in - 1);
    const int xlo = max((int)fx, 0), xhi = min((int)ceilf(px), w_in - 1);
    const float ty = py - fy, tx = px - fx;
    const float* base = x + b * h_in * w_in * ldx + c4 * 4;
    const f32x4 tl = *reinterpret_cast<const f32x4*>(base + ((long long)ylo * w_in + xlo) * ldx);
    const f32x4 tr = *reinterpret_cast<const f32x4*>(base + ((long long)ylo * w_in + xhi) * ldx);
    const f32x4 bl = *reinterpret_cast<const f32x4*>(base + ((long long)yhi * w_in + xlo) * ldx);
    const f32x4 br = *reinterpret_cast<const f32x4*>(base + ((long long)yhi * w_in + xhi) * ldx);
    const f32x4 top = tl + (tr - tl) * tx;
    const f32x4 bot = bl + (br - bl) * tx;
    *reinterpret_cast<f32x4*>(y + ((b * h_out + oy) * w_out + ox) * ldy + c4 * 4) = top + (bot - top) * ty;
}

// 4x horizontal upsampling (the decoder's Resizing 32 -> 128, model.py:241-242): the four outputs 4k .. 4k + 3 read input
// columns k - 1, k, k + 1 only, so a thread produces all four from 6 loads instead of 16 -- the L1 / texture-address path,
// not HBM, bounded the one-output-per-thread form (1.68 GB written at 3 TB/s).  Same expressions per output -> bit-identical.
__global__ __launch_bounds__(256) void resize_bilinear_x4_kernel(const float* __restrict__ x, float* __restrict__ y, int h_in,
                                                                 int w_in, int c, int h_out, int ldx, int ldy, float scale_y) {
    const unsigned c4n = (unsigned)c >> 2;
    const unsigned idx = blockIdx.x * 256u + threadIdx.x;
    const unsigned k = idx / c4n, c4 = idx - k * c4n;
    if (k >= (unsigned)w_in) return;
    const int oy = blockIdx.y;
    const long long b = blockIdx.z;
    const float py = ((float)oy + 0.5f) * scale_y - 0.5f;
    const float fy = floorf(py);
    const int ylo = max((int)fy, 0), yhi = min((int)ceilf(py), h_in - 1);
    const float ty = py - fy;
    const int cm = max((int)k - 1, 0), cp = min((int)k + 1, w_in - 1);
    const float* base = x + b * h_in * w_in * ldx + c4 * 4;
    const float* rlo = base + (long long)ylo * w_in * ldx;
    const float* rhi = base + (long long)yhi * w_in * ldx;
    const f32x4 tA = *reinterpret_cast<const f32x4*>(rlo + (long long)cm * ldx), tB = *reinterpret_cast<const f32x4*>(rlo + (long long)k * ldx),
                tC = *reinterpret_cast<const f32x4*>(rlo + (long long)cp * ldx);
    const f32x4 bA = *reinterpret_cast<const f32x4*>(rhi + (long long)cm * ldx), bB = *reinterpret_cast<const f32x4*>(rhi + (long long)k * ldx),
                bC = *reinterpret_cast<const f32x4*>(rhi + (long long)cp * ldx);
    float* out = y + ((b * h_out + oy) * (long long)(4 * w_in) + 4 * k) * ldy + c4 * 4;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const float px = ((float)(4 * k + t) + 0.5f) * 0.25f - 0.5f;      // the generic kernel's expression with scale_x = 0.25
        const float tx = px - floorf(px);
        const f32x4 tl = t < 2 ? tA : tB, tr = t < 2 ? tB : tC, bl = t < 2 ? bA : bB, br = t < 2 ? bB : bC;
        const f32x4 top = tl + (tr - tl) * tx;
        const f32x4 bot = bl + (br - bl) * tx;
        *reinterpret_cast<f32x4*>(out + (long long)t * ldy) = top + (bot - top) * ty;
    }
}

// Standard-output mask (generate_standard_output.py:52-65 with the model's final Resizing, model.py:108-111): bilinear
// upsample of one logits map (half-pixel, the arithmetic of resize_bilinear_kernel), argmax over the classes (first
// maximum), keep class_id else 0 -- one pass, the upsampled logits never exist.
__global__ __launch_bounds__(256) void standard_mask_kernel(const float* __restrict__ logits, int* __restrict__ out, int h_in,
                                                            int w_in, int classes, int h_out, int w_out, float scale_y,
                                                            float scale_x, int class_id) {
    const int ox = blockIdx.x * 256 + threadIdx.x, oy = blockIdx.y;
    if (ox >= w_out) return;
    const float py = ((float)oy + 0.5f) * scale_y - 0.5f, px = ((float)ox + 0.5f) * scale_x - 0.5f;
    const float fy = floorf(py), fx = floorf(px);
    const int ylo = max((int)fy, 0), yhi = min((int)ceilf(py), h_in - 1);
    const int xlo = max((int)fx, 0), xhi = min((int)ceilf(px), w_in - 1);
    const float ty = py - fy, tx = px - fx;
    const float* tl = logits + ((long long)ylo * w_in + xlo) * classes;
    const float* tr = logits + ((long long)ylo * w_in + xhi) * classes;
    const float* bl = logits + ((long long)yhi * w_in + xlo) * classes;
    const float* br = logits + ((long long)yhi * w_in + xhi) * classes;
    float best = 0.0f;
    int arg = 0;
    for (int c = 0; c < classes; ++c) {
        const float top = tl[c] + (tr[c] - tl[c]) * tx;
        const float bot = bl[c] + (br[c] - bl[c]) * tx;
        const float v = top + (bot - top) * ty;
        if (c == 0 || v > best) { best = v; arg = c; }
    }
    out[(long long)oy * w_out + ox] = (arg == class_id) ? class_id : 0;
}

int cap_grid(long long total) {
    const long long g = asr_cdiv(total, 256);
    return (int)(g < 8192 ? (g > 0 ? g : 1) : 8192);
}

}  // namespace

extern "C" int asr_conv3x3_direct_f32(const float* x, const float* w, const float* bias, float* y, int batch, int h_in,
                                      int w_in, int cin, int cout, int stride, int pad_top, int pad_left, int h_out,
                                      int w_out, int ldx, int ldy, int relu, asr_stream_t stream) {
    ASR_REQUIRE(x && w && bias && y, "asr_conv3x3_direct_f32: null pointer");
    ASR_REQUIRE(batch > 0 && h_in > 0 && w_in > 0 && cin > 0 && cout > 0 && stride > 0 && h_out > 0 && w_out > 0 &&
                    pad_top >= 0 && pad_left >= 0 && ldx >= cin && ldy >= cout,
                "asr_conv3x3_direct_f32: bad geometry");
    ASR_UNSUPPORTED((cout & 3) || (ldy & 3) || 9 * cin * cout > 12288,
                    "asr_conv3x3_direct_f32: cout %% 4 == 0, ldy %% 4 == 0 and 9*cin*cout <= 12288 required (cin=%d cout=%d)", cin, cout);
    ASR_UNSUPPORTED((reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(bias)) & 15,
                    "asr_conv3x3_direct_f32: y and bias must be 16-byte aligned");
    if (cin == 3 && cout == 32) {
        const long long waves = asr_cdiv((long long)batch * h_out * asr_cdiv(w_out, SP), 8);
        hipLaunchKernelGGL(conv3x3_stem_kernel, dim3(cap_grid(waves * 64)), dim3(256), 0, asr_stream(stream), x, w, bias, y,
                           batch, h_in, w_in, stride, pad_top, pad_left, h_out, w_out, ldx, ldy, relu);
        ASR_LAUNCH_CHECK();
        return ASR_OK;
    }
    const long long total = (long long)batch * h_out * w_out * (cout >> 2);
    hipLaunchKernelGGL(conv3x3_direct_kernel, dim3(cap_grid(total)), dim3(256), sizeof(float) * 9 * cin * cout,
                       asr_stream(stream), x, w, bias, y, batch, h_in, w_in, cin, cout, stride, pad_top, pad_left, h_out,
                       w_out, ldx, ldy, relu);
    ASR_LAUNCH_CHECK();
    return ASR_OK;
}

extern "C" int asr_conv3x3_stem_f16x3(const float* x, const float* w, const float* bias, float* y, int batch, int h_in, int w_in,
                                      int cin, int cout, int stride, int pad_top, int pad_left, int h_out, int w_out, int ldx,
                                      int ldy, int relu, asr_stream_t stream) {
    ASR_REQUIRE(x && w && bias && y, "asr_conv3x3_stem_f16x3: null pointer");
    ASR_REQUIRE(batch > 0 && h_in > 0 && w_in > 0 && stride > 0 && h_out > 0 && w_out > 0 && pad_top >= 0 && pad_left >= 0 &&
                    ldx >= cin && ldy >= cout,
                "asr_conv3x3_stem_f16x3: bad geometry");
    ASR_UNSUPPORTED(cin != 3 || cout != 32, "asr_conv3x3_stem_f16x3: cin = 3, cout = 32 only (got %d, %d)", cin, cout);
    ASR_REQUIRE((h_out - 1) * stride - pad_top + 2 < h_in + 2 && (w_out - 1) * stride - pad_left + 2 < w_in + 2,
                "asr_conv3x3_stem_f16x3: output %dx%d does not fit input %dx%d", h_out, w_out, h_in, w_in);
    ASR_UNSUPPORTED((long long)h_in * w_in * ldx > 0x7fffffffLL, "asr_conv3x3_stem_f16x3: image too large");
    const long long waves = (long long)batch * h_out * asr_cdiv(w_out, 32);
    hipLaunchKernelGGL(conv3x3_stem_mfma_kernel, dim3(cap_grid(waves * 64)), dim3(256), 0, asr_stream(stream), x, w, bias, y,
                       batch, h_in, w_in, stride, pad_top, pad_left, h_out, w_out, ldx, ldy, relu);
    ASR_LAUNCH_CHECK();
    return ASR_OK;
}

extern "C" int asr_entry_stem_f16x3(const float* x, const float* w1, const float* b1, const void* w2_packed, const float* b2,
                                    float* y, int batch, int h_in, int w_in, int ldx, int ldy, asr_stream_t stream) {
    ASR_REQUIRE(x && w1 && b1 && w2_packed && b2 && y, "asr_entry_stem_f16x3: null pointer");
    ASR_REQUIRE(batch > 0 && h_in > 0 && w_in > 0 && ldx >= 3 && ldy >= 64, "asr_entry_stem_f16x3: bad geometry");
    ASR_UNSUPPORTED((h_in | w_in) & 1, "asr_entry_stem_f16x3: even input size required (got %dx%d)", h_in, w_in);
    ASR_UNSUPPORTED((long long)h_in * w_in * ldx > 0x7fffffffLL, "asr_entry_stem_f16x3: image too large");
    ASR_UNSUPPORTED(reinterpret_cast<uintptr_t>(w2_packed) & 15, "asr_entry_stem_f16x3: w2_packed must be 16-byte aligned");
    const int h1 = h_in / 2, w1d = w_in / 2;
    const int cus = asr_device_cu_count();
    ASR_REQUIRE(cus > 0, "asr_entry_stem_f16x3: cannot query the device");
    static AsrDeviceOnce once;
    ASR_HIP_CHECK(asr_allow_dynamic_lds(once, reinterpret_cast<const void*>(entry_stem_fused_kernel), ES_LDS_BYTES));
    const long long tiles = (long long)batch * asr_cdiv(h1, ES_T) * asr_cdiv(w1d, ES_T);
    const int grid = (int)(tiles < cus ? tiles : cus);
    hipLaunchKernelGGL(entry_stem_fused_kernel, dim3(grid), dim3(512), ES_LDS_BYTES, asr_stream(stream), x, w1, b1,
                       reinterpret_cast<const _Float16*>(w2_packed), b2, y, batch, h_in, w_in, h1, w1d, ldx, ldy, 128);
    ASR_LAUNCH_CHECK();
    return ASR_OK;
}

extern "C" int asr_gap_f32(const float* x, float* y, int batch, int hw, int c, int ldx, asr_stream_t stream) {
    ASR_REQUIRE(x && y, "asr_gap_f32: null pointer");
    ASR_REQUIRE(batch > 0 && batch <= 65535 && hw > 0 && c > 0 && ldx >= c, "asr_gap_f32: bad shape");
    ASR_UNSUPPORTED((c & 3) || (ldx & 3) || ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) & 15),
                    "asr_gap_f32: c, ldx multiples of 4 and 16-byte aligned pointers required");
    hipLaunchKernelGGL(gap_kernel, dim3((unsigned)asr_cdiv(c, 256), batch), dim3(256), 0, asr_stream(stream), x, y, hw, c, ldx);
    ASR_LAUNCH_CHECK();
    return ASR_OK;
}

extern "C" int asr_resize_bilinear_f32(const float* x, float* y, int batch, int h_in, int w_in, int c, int h_out, int w_out,
                                       int ldx, int ldy, asr_stream_t stream) {
    ASR_REQUIRE(x && y, "asr_resize_bilinear_f32: null pointer");
    ASR_REQUIRE(batch > 0 && h_in > 0 && w_in > 0 && c > 0 && h_out > 0 && w_out > 0 && ldx >= c && ldy >= c,
                "asr_resize_bilinear_f32: bad shape");
    ASR_UNSUPPORTED((c & 3) || (ldx & 3) || (ldy & 3) || ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) & 15),
                    "asr_resize_bilinear_f32: c, ldx, ldy multiples of 4 and 16-byte aligned pointers required");
    const float sy = (float)h_in / (float)h_out, sx = (float)w_in / (float)w_out;
    ASR_UNSUPPORTED(batch > 65535 || h_out > 65535, "asr_resize_bilinear_f32: batch and h_out must not exceed 65535 (grid dimensions)");
    if (w_out == 4 * w_in && w_in >= 2) {
        hipLaunchKernelGGL(resize_bilinear_x4_kernel, dim3((unsigned)asr_cdiv((long long)w_in * (c >> 2), 256), (unsigned)h_out, (unsigned)batch),
                           dim3(256), 0, asr_stream(stream), x, y, h_in, w_in, c, h_out, ldx, ldy, sy);
        ASR_LAUNCH_CHECK();
        return ASR_OK;
    }
    hipLaunchKernelGGL(resize_bilinear_kernel, dim3((unsigned)asr_cdiv((long long)w_out * (c >> 2), 256), (unsigned)h_out, (unsigned)batch),
                       dim3(256), 0, asr_stream(stream), x, y, batch, h_in, w_in, c, h_out, w_out, ldx, ldy, sy, sx);
    ASR_LAUNCH_CHECK();
    return ASR_OK;
}

extern "C" int asr_standard_mask_i32(const float* logits, int32_t* mask, int h_in, int w_in, int classes, int h_out, int w_out,
                                     int class_id, asr_stream_t stream) {
    ASR_REQUIRE(logits && mask, "asr_standard_mask_i32: null pointer");
    ASR_REQUIRE(h_in > 0 && w_in > 0 && classes > 0 && h_out > 0 && w_out > 0 && h_out <= 65535 && class_id >= 0 &&
                    class_id < classes,
                "asr_standard_mask_i32: bad shape / class");
    const float sy = (float)h_in / (float)h_out, sx = (float)w_in / (float)w_out;
    hipLaunchKernelGGL(standard_mask_kernel, dim3((unsigned)asr_cdiv(w_out, 256), (unsigned)h_out), dim3(256), 0, asr_stream(stream),
                       logits, mask, h_in, w_in, classes, h_out, w_out, sy, sx, class_id);
    ASR_LAUNCH_CHECK();
    return ASR_OK;
}
