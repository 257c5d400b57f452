// Fused separable convolution for the high-resolution entry-flow layers (gfx950):
//   asr_sepconv_fused_f16x3   [ReLU ->] DepthwiseConv2D 3x3 (stride 1, rate 1) + BN [-> ReLU] -> Conv2D 1x1 + BN [-> ReLU]
//                             for cin in {64, 128} -> 128 (entry_flow_block1_separable_conv1 / conv2 at 256 x 256) and
//                             cin % 16 == 0 -> 256 (entry_flow_block2, decoder_conv0 / conv1 at 128 x 128)
//                             (model.py:157-158 through _Xception_block / _SepConv_BN, model.py:381-424, 463-508).
// At that resolution both halves of a separable conv are HBM-bound and the depthwise output (3.4 GB per 100 copies) is
// written by one kernel only to be read back by the next.  Here a persistent workgroup (8 waves) owns 8 x 14 output
// pixels at a time:
//   stage 1  depthwise: a 16-lane DPP row is 16 consecutive input columns (the tile's 14 + one halo column each side), so
//            every input element is loaded ONCE per thread and the left / right neighbours come from DPP row shifts
//            -- no 3x redundant tap loads --; a thread owns 4 channels and marches down the rows with the 3 x 3 window
//            in registers.  The result goes to LDS already split into hi/lo f16, one line per pixel
//            [hi c0..cin-1 | lo c0..cin-1] in 16-byte slots, slot XOR (line & 15) within 256-byte groups;
//   stage 2  pointwise: a 128 (lines) x 128 (channels) x cin GEMM on split-f16 MFMA whose A fragments are read from
//            that LDS image and whose weights (cin x 128, hi + lo) stay resident in LDS for the kernel's lifetime.
// Same depthwise arithmetic (bias first, taps in (ky, kx) order) and the same MFMA sequence per accumulator as the
// two-kernel form, so the results are bit-identical to asr_dwconv3x3_nhwc_f32 + asr_pwconv_mfma_f16x3.
#include "asr_common.h"

// 1: the stage-1 split uses the packed saturating conversions of asr_common.h (2.4 - 3 % faster than clamp + convert per
// value).  Round 3 switched it off because with it (208 instead of 212 registers) a wave of the SR solver fits beside two of
// this kernel's waves on a SIMD, whose MFMAs then corrupted the solver's op_sel:[0,1] packed-f32 instructions;
// since round 4 no kernel of the library contains that instruction form (csrc/isa_guard.py).
#ifndef ASR_SEPCONV_PACKED_SPLIT
#define ASR_SEPCONV_PACKED_SPLIT 1
#endif

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

namespace {

constexpr int SF_TH = 8, SF_TW = 14, SF_VALID = SF_TH * SF_TW, SF_M = 128, SF_N = 128;

template <int CTRL>
__device__ __forceinline__ f32x4 sf_dpp4(f32x4 v) {           // a DPP row shift of all four channels (0 shifted in at the row's end)
    f32x4 r;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float f = v[i];          // (hipcc 7.2 miscompiles __builtin_bit_cast(int, v[i]) on a vector element: it reads v[0])
        r[i] = __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(f), CTRL, 0xF, 0xF, true));
    }
    return r;
}

__device__ __forceinline__ int sf_swz(int slot, int line) { return (slot & ~15) | ((slot ^ line) & 15); }

#ifdef ASR_DIAG_SEPCONV_MAX_VGPR
#define ASR_SEPCONV_ATTR __attribute__((amdgpu_num_vgpr(ASR_DIAG_SEPCONV_MAX_VGPR)))
#else
#define ASR_SEPCONV_ATTR
#endif
template <int CIN>
__global__ __launch_bounds__(512) ASR_PK_F32 ASR_SEPCONV_ATTR void sepconv_fused_kernel(const float* __restrict__ x, const float* __restrict__ wd,
                                                            const float* __restrict__ bd, const _Float16* __restrict__ wp,
                                                            const float* __restrict__ bp, float* __restrict__ y, int batch, int h,
                                                            int w, int ldx, int ldy, int npad, int pre_relu, int dw_relu,
                                                            int out_relu) {
#if ASR_SEPCONV_PACKED_SPLIT
    asr_enable_f16_saturation();                              // the stage-1 split converts with the hardware's f16 clamp
#endif
#ifdef ASR_DIAG_SEPCONV_TOP_VGPR
    ASR_DIAG_TOUCH_VGPR(ASR_DIAG_SEPCONV_TOP_VGPR);
#endif
    constexpr int QUADS = CIN / 4;                            // channel quads: 32 or 16
    constexpr int HALVES = 32 / QUADS;                        // row halves of the tile handled by different thread slots
    constexpr int RPT = SF_TH / HALVES;                       // output rows per thread: 8 or 4
    constexpr int LB = CIN * 4;                               // bytes of one LDS line (hi + lo halfs of all channels)
    constexpr int OCTS = CIN / 8;
    constexpr int A_BYTES = SF_M * LB, B_BYTES = OCTS * SF_N * 16;
    extern __shared__ __attribute__((aligned(16))) char sf_lds[];
    char* const A = sf_lds;
    char* const Bh = sf_lds + A_BYTES;
    char* const Bl = Bh + B_BYTES;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l32 = lane & 31, hh = lane >> 5;
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    {   // resident pointwise weights: packed planes [OCTS][npad][8 halfs] (hi, then lo) -> LDS [OCTS][128][8]
        const long long plane = (long long)CIN * npad;
        for (int i = tid; i < OCTS * SF_N; i += 512) {
            const int oct = i >> 7, col = i & 127;
            const _Float16* src = wp + ((long long)oct * npad + col) * 8;
            *reinterpret_cast<u32x4*>(Bh + i * 16) = *reinterpret_cast<const u32x4*>(src);
            *reinterpret_cast<u32x4*>(Bl + i * 16) = *reinterpret_cast<const u32x4*>(src + plane);
        }
    }
    // ---- stage-1 role: (input column lane16 of the 16-wide strip, channel quad q, row half) ----
    const int lane16 = lane & 15, slot = wave * 4 + (lane >> 4);
    const int q = slot % QUADS, ry0 = (slot / QUADS) * RPT, ch = q * 4;
    f32x4 wk[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) wk[t] = *reinterpret_cast<const f32x4*>(wd + t * CIN + ch);
    const f32x4 bv = *reinterpret_cast<const f32x4*>(bd + ch);
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    // ---- stage-2 role: 32 lines x 64 output channels ----
    const int wm = wave & 3, wn = wave >> 2;
    const float bp0 = bp[wn * 64 + l32], bp1 = bp[wn * 64 + 32 + l32];

    const int tiles_x = (w + SF_TW - 1) / SF_TW, tiles_y = (h + SF_TH - 1) / SF_TH;
    const long long total = (long long)batch * tiles_y * tiles_x;
    // The RPT + 2 input rows of a tile for this thread (clamped addresses, no branch).  They are requested one tile ahead:
    // right after stage 1 has consumed the current rows, so they fly under stage 2 (MFMAs, stores) of the current tile.
    f32x4 in[RPT + 2];
    auto request_rows = [&](long long t) {
        const int tx0 = (int)(t % tiles_x) * SF_TW;
        const long long tt = t / tiles_x;
        const int ty0 = (int)(tt % tiles_y) * SF_TH;
        const float* xin = x + (tt / tiles_y) * h * w * ldx + ch;
        const int ixc = min(max(tx0 - 1 + lane16, 0), w - 1);
#pragma unroll
        for (int r = 0; r < RPT + 2; ++r) {
            const int iy = min(max(ty0 + ry0 - 1 + r, 0), h - 1);
            in[r] = *reinterpret_cast<const f32x4*>(xin + ((long long)iy * w + ixc) * ldx);
        }
    };
    // Every workgroup walks ONE contiguous run of tiles (row-major over the image, images one after the other) instead of
    // every gridDim-th tile: the halo columns of a tile are then re-read by the same workgroup one tile later and its halo
    // rows one tile row (19 tiles, 1.5 MB of traffic) later -- L2 hits in its own XCD -- where the strided walk had the
    // neighbours of a tile running at the same time on other XCDs, each fetching the shared lines from HBM for itself
    // (PMC, round 2: 1.2x the algorithmic bytes).  Same time (the kernel is not bound by its reads), less HBM traffic.
    const long long per_wg = (total + gridDim.x - 1) / gridDim.x;
    const long long first = (long long)blockIdx.x * per_wg, last = min(first + per_wg, total);
    if (first < last) request_rows(first);
    for (long long tile = first; tile < last; ++tile) {
        const int x0 = (int)(tile % tiles_x) * SF_TW;
        const long long tt = tile / tiles_x;
        const int y0 = (int)(tt % tiles_y) * SF_TH;
        const long long b = tt / tiles_y;
        __syncthreads();                                      // the previous tile's stage 2 is done with the LDS image
        // ---- stage 1 ----
        const int ix = x0 - 1 + lane16;
        const bool col_ok = ix >= 0 && ix < w;
        f32x4 win[3][3];                                       // [window row][left, centre, right]
#pragma unroll
        for (int r = 0; r < RPT + 2; ++r) {
            const int iy = y0 + ry0 - 1 + r;
            f32x4 c = in[r];
            if (pre_relu) {
#pragma unroll
                for (int i = 0; i < 4; ++i) c[i] = fmaxf(c[i], 0.f);
            }
            c = (col_ok && iy >= 0 && iy < h) ? c : zero;      // zero padding of the 'same' convolution
            win[2][1] = c;
            win[2][0] = sf_dpp4<0x111>(c);                     // row_shr:1 -> the value of lane - 1 = input column ix - 1
            win[2][2] = sf_dpp4<0x101>(c);                     // row_shl:1 -> the value of lane + 1 = input column ix + 1
            if (r >= 2) {
                f32x4 acc = bv;
#pragma unroll
                for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx) acc += win[ky][kx] * wk[ky * 3 + kx];
                if (dw_relu) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) acc[i] = fmaxf(acc[i], 0.f);
                }
                typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
#if ASR_SEPCONV_PACKED_SPLIT
                unsigned int h01, h23, l01, l23;                // the split with saturating packed conversions (asr_common.h)
                asr_split4_f16_saturating_mode(acc[0], acc[1], acc[2], acc[3], h01, h23, l01, l23);
                u32x2 h2 = {h01, h23}, l2 = {l01, l23};
#else
                f16x4 hi, lo;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    _Float16 hf, lf;
                    asr_split_f16(acc[i], hf, lf);
                    hi[i] = hf;
                    lo[i] = lf;
                }
                u32x2 h2 = __builtin_bit_cast(u32x2, hi), l2 = __builtin_bit_cast(u32x2, lo);
#endif
                // Lanes 16 apart hold neighbouring channel quads (q even / odd) of the same pixel: they trade halves
                // (v_permlane16_swap: row r of 16 lanes <-> row r ^ 1) so that the even quad's lane holds the 8 hi halfs of
                // both quads and the odd quad's lane their 8 lo halfs -- ONE 16-byte LDS store per lane into a whole
                // swizzled slot (conflict-free: 8 consecutive lines per store phase) instead of two 8-byte stores that
                // hit each bank pair twice (20 % of the LDS cycles were bank conflicts, profiles/r02_pmc_sq.json).  Same
                // bytes at the same LDS addresses as before.
                {   // swap(a = hi, b = lo): a's odd rows <-> b's even rows
                    const auto sx = __builtin_amdgcn_permlane16_swap(h2.x, l2.x, false, false);
                    const auto sy = __builtin_amdgcn_permlane16_swap(h2.y, l2.y, false, false);
                    h2.x = sx[0]; l2.x = sx[1];
                    h2.y = sy[0]; l2.y = sy[1];
                }
                // even quad: {hi(q), hi(q + 1)} -> hi slot q / 2; odd quad: {lo(q - 1), lo(q)} -> lo slot q / 2
                const u32x4 out = {h2.x, h2.y, l2.x, l2.y};
                if (lane16 >= 1 && lane16 <= SF_TW) {          // lanes 0 and 15 are the halo columns
                    const int line = (ry0 + r - 2) * SF_TW + lane16 - 1;
                    *reinterpret_cast<u32x4*>(A + line * LB + sf_swz(((q & 1) ? OCTS : 0) + (q >> 1), line) * 16) = out;
                }
            }
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                win[0][k] = win[1][k];
                win[1][k] = win[2][k];
            }
        }
        request_rows(min(tile + 1, last - 1));                // next tile's rows (the last tile re-requests its own)
        __syncthreads();
        // ---- stage 2: out[line][n] = sum_c A[line][c] * W[c][n] on split-f16 MFMA, A from the LDS image ----
        f32x16 acc2[2];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc2[j][e] = 0.0f;
        const int m = wm * 32 + l32;
        const char* const aline = A + m * LB;
#pragma unroll
        for (int kk = 0; kk < CIN / 16; ++kk) {
            const int oct = 2 * kk + hh;
            const f16x8 ah = *reinterpret_cast<const f16x8*>(aline + sf_swz(oct, m) * 16);
            const f16x8 al = *reinterpret_cast<const f16x8*>(aline + sf_swz(OCTS + oct, m) * 16);
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int col = wn * 64 + j * 32 + l32;
                const f16x8 bh = *reinterpret_cast<const f16x8*>(Bh + (oct * SF_N + col) * 16);
                const f16x8 bl = *reinterpret_cast<const f16x8*>(Bl + (oct * SF_N + col) * 16);
                acc2[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc2[j], 0, 0, 0);
                acc2[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc2[j], 0, 0, 0);
                acc2[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc2[j], 0, 0, 0);
            }
        }
        // C/D map: column (output channel) = lane & 31, row (line) = (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5)
        float* const ybase = y + (b * h * w) * ldy + wn * 64 + l32;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int line = wm * 32 + (e & 3) + 8 * (e >> 2) + 4 * hh;
            const int ty = line / SF_TW, tx = line - ty * SF_TW;
            const int oy = y0 + ty, ox = x0 + tx;
            if (line < SF_VALID && oy < h && ox < w) {
                float* const o = ybase + ((long long)oy * w + ox) * ldy;
                float v0 = acc2[0][e] + bp0, v1 = acc2[1][e] + bp1;
                if (out_relu) { v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); }
                o[0] = v0;
                o[32] = v1;
            }
        }
    }
}

template <int CIN>
int launch_sepconv_fused(const float* x, const float* wd, const float* bd, const void* wp, const float* bp, float* y, int batch,
                         int h, int w, int ldx, int ldy, int npad, int pre_relu, int dw_relu, int out_relu, hipStream_t s) {
    constexpr int lds = SF_M * CIN * 4 + 2 * (CIN / 8) * SF_N * 16;        // 128 KB (cin 128) / 64 KB (cin 64)
    const int cus = asr_device_cu_count();
    ASR_REQUIRE(cus > 0, "asr_sepconv_fused_f16x3: cannot query the device");
    static AsrDeviceOnce once;
    ASR_HIP_CHECK(asr_allow_dynamic_lds(once, reinterpret_cast<const void*>(sepconv_fused_kernel<CIN>), lds));
    const long long tiles = (long long)batch * asr_cdiv(h, SF_TH) * asr_cdiv(w, SF_TW);
    const long long want = (long long)cus * (lds <= 80 * 1024 ? 2 : 1);
    const int grid = (int)(tiles < want ? tiles : want);
    hipLaunchKernelGGL(sepconv_fused_kernel<CIN>, dim3(grid), dim3(512), lds, s, x, wd, bd, reinterpret_cast<const _Float16*>(wp), bp, y,
                       batch, h, w, ldx, ldy, npad, pre_relu, dw_relu, out_relu);
    ASR_LAUNCH_CHECK();
    return ASR_OK;
}

}  // namespace

extern "C" int asr_sepconv_fused_f16x3(const float* x, const float* w_dw, const float* bias_dw, const void* w_pw_packed,
                                       const float* bias_pw, float* y, int batch, int h, int w, int cin, int cout, int ldx,
                                       int ldy, int pre_relu, int dw_relu, int out_relu, asr_stream_t stream) {
    ASR_REQUIRE(x && w_dw && bias_dw && w_pw_packed && bias_pw && y, "asr_sepconv_fused_f16x3: null pointer");
    ASR_REQUIRE(batch > 0 && h > 0 && w > 0 && ldx >= cin && ldy >= cout, "asr_sepconv_fused_f16x3: bad geometry");
    ASR_UNSUPPORTED(!((cin == 64 || cin == 128) && cout == 128),
                    "asr_sepconv_fused_f16x3: cin in {64, 128} -> 128 only (got %d -> %d)", cin, cout);
    ASR_UNSUPPORTED((ldx & 3) || ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(w_dw) |
                                   reinterpret_cast<uintptr_t>(bias_dw) | reinterpret_cast<uintptr_t>(w_pw_packed)) & 15),
                    "asr_sepconv_fused_f16x3: ldx %% 4 == 0 and 16-byte aligned x / weights required");
    ASR_UNSUPPORTED((long long)h * w * ldx > 0x7fffffffLL, "asr_sepconv_fused_f16x3: image too large");
    hipStream_t s = asr_stream(stream);
    const int npad = 128;
    if (cin == 64)
        return launch_sepconv_fused<64>(x, w_dw, bias_dw, w_pw_packed, bias_pw, y, batch, h, w, ldx, ldy, npad, pre_relu, dw_relu, out_relu, s);
    return launch_sepconv_fused<128>(x, w_dw, bias_dw, w_pw_packed, bias_pw, y, batch, h, w, ldx, ldy, npad, pre_relu, dw_relu, out_relu, s);
}
