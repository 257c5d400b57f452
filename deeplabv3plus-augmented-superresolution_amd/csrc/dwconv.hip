// Depthwise 3x3 convolution, NHWC float32, with the BatchNorm that follows it folded into the
// weights/bias and the ReLUs on either side fused (gfx950).
//
// Reference: model.py:463-508 (_SepConv_BN): [ZeroPadding2D if stride != 1] -> [ReLU if not
// depth_activation] -> DepthwiseConv2D(3x3, stride, dilation) -> BN -> [ReLU if depth_activation];
// model.py:212-221 for the three ASPP branches (rates 6/12/18 on the same input).
//
// HBM-bound (2.1 flop/byte).  Channels are the fastest axis, so a lane owns 4 consecutive
// channels (one 16-byte access) and neighbouring lanes neighbouring channels: every global
// access of a wave covers whole 256-byte pieces of pixel rows.  Kernels (an LDS-staged tile kernel was
// measured at 3.5-4.0 TB/s against 3.8-5.6 TB/s for the streaming forms and is not built):
//  * dw_stream_kernel<R,S,ROWS> (default for rate <= 2, stride 1 or 2 -- 93 % of the depthwise
//    bytes): register sliding window.  A thread owns (4 channels, one output column) and marches
//    down a strip of output rows keeping the (2R+1) x 3 taps in registers; the next input rows are
//    loaded before the current row's FMAs; no barrier, >= 4 waves per SIMD.
//  * aspp_dw3_phase_kernel: the three dilated ASPP depthwise convs fused -- the plane is cut into its residue
//    classes modulo gcd(rates) (on which the dilated taps close), each staged in LDS once and all three rates computed
//    from it, so the input is read from HBM once instead of three times, on planes of any size.
//  * dw_direct_kernel: any stride / rate, taps straight from L1/L2 (fallback).
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
// post-activation selector of the C ABI: 0 none, 1 ReLU, 2 ReLU6 (MobileNetV2)
__device__ __forceinline__ f32x4 post_act4(f32x4 v, int mode) {
    if (mode) v = relu4(v);
    if (mode == 2) { v.x = fminf(v.x, 6.f); v.y = fminf(v.y, 6.f); v.z = fminf(v.z, 6.f); v.w = fminf(v.w, 6.f); }
    return v;
}

// ---------------------------------------------------------------------------------------------
// register sliding window (stride S, rate R)
// ---------------------------------------------------------------------------------------------
constexpr int SCOLS = 16;

template <int R, int S, int SROWS>
__global__ __launch_bounds__(256) void dw_stream_kernel(DwArgs p, int tiles_x) {
    constexpr int WIN = 2 * R + 1;
    const int tid = threadIdx.x;
    const int c4 = tid & 15, col = tid >> 4;
    const int tx = blockIdx.x % tiles_x, cbk = blockIdx.x / tiles_x;
    const int ch = (cbk * 16 + c4) * 4;
    const int ox = tx * SCOLS + col;
    const int oy0 = blockIdx.y * SROWS;
    const int b = blockIdx.z;
    if (ch >= p.c || ox >= p.w_out) return;
    const float* xin = p.x + (long long)b * p.h_in * p.w_in * p.ldx + ch;
    float* yout = p.y + (long long)b * p.h_out * p.w_out * p.ldy + ch;
    f32x4 wk[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) wk[t] = *reinterpret_cast<const f32x4*>(p.w + (long long)t * p.c + ch);
    const f32x4 bv = *reinterpret_cast<const f32x4*>(p.bias + ch);
    const int ixl = ox * S - p.pad_left, ixc = ixl + R, ixr = ixl + 2 * R;
    const bool vl = ixl >= 0 && ixl < p.w_in, vc = ixc >= 0 && ixc < p.w_in, vr = ixr >= 0 && ixr < p.w_in;
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};

    auto load_row = [&](int iy, f32x4& l, f32x4& c, f32x4& r) {
        l = zero; c = zero; r = zero;
        if (iy >= 0 && iy < p.h_in) {
            const float* row = xin + (long long)iy * p.w_in * p.ldx;
            if (vl) l = *reinterpret_cast<const f32x4*>(row + (long long)ixl * p.ldx);
            if (vc) c = *reinterpret_cast<const f32x4*>(row + (long long)ixc * p.ldx);
            if (vr) r = *reinterpret_cast<const f32x4*>(row + (long long)ixr * p.ldx);
        }
    };
    // The pre-activation is applied when a row ENTERS the window, not when it is loaded: touching the registers of a
    // load right after issuing it makes the compiler wait for it there (s_waitcnt vmcnt(0)) and the "prefetch" of the
    // next rows would be synchronous on every pre-ReLU layer (all of the middle flow).
    auto act = [&](const f32x4& v) { return p.pre_relu ? relu4(v) : v; };

    // window rows [base, base + WIN) with base = oy * S - pad_top; it advances by S rows per output row
    f32x4 win[WIN][3], nxt[S][3];
    const int base0 = oy0 * S - p.pad_top;
#pragma unroll
    for (int k = 0; k < WIN - S; ++k) {
        load_row(base0 + k, win[S + k][0], win[S + k][1], win[S + k][2]);
        win[S + k][0] = act(win[S + k][0]); win[S + k][1] = act(win[S + k][1]); win[S + k][2] = act(win[S + k][2]);
    }
#pragma unroll
    for (int j = 0; j < S; ++j) load_row(base0 + WIN - S + j, nxt[j][0], nxt[j][1], nxt[j][2]);
    const int rows = min(SROWS, p.h_out - oy0);
    for (int r = 0; r < rows; ++r) {
#pragma unroll
        for (int k = 0; k < WIN - S; ++k) { win[k][0] = win[k + S][0]; win[k][1] = win[k + S][1]; win[k][2] = win[k + S][2]; }
#pragma unroll
        for (int j = 0; j < S; ++j) { win[WIN - S + j][0] = act(nxt[j][0]); win[WIN - S + j][1] = act(nxt[j][1]); win[WIN - S + j][2] = act(nxt[j][2]); }
        if (r + 1 < rows) {  // prefetch the S input rows the next output row adds
#pragma unroll
            for (int j = 0; j < S; ++j) load_row(base0 + r * S + WIN + j, nxt[j][0], nxt[j][1], nxt[j][2]);
        }
        f32x4 acc = bv;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) acc += win[ky * R][kx] * wk[ky * 3 + kx];
        acc = post_act4(acc, p.post_relu);
        *reinterpret_cast<f32x4*>(yout + ((long long)(oy0 + r) * p.w_out + ox) * p.ldy) = acc;
    }
}

// The same register window for strips that are all full (h_out % SROWS == 0: every layer of the net), written so that
// the compiler can count outstanding memory operations exactly: no branch around any load or store (rows and columns
// outside the image are CLAMPED to a valid address and zeroed when the row enters the window), PF output rows of
// input in flight per thread, held in a queue indexed at compile time (the strip loop is unrolled PF-fold, no register
// of an outstanding load is ever copied).  s_waitcnt vmcnt(N) then waits for exactly the row that enters the window
// and leaves the younger loads and the previous stores in flight; with a branch or a copy in the way the wait
// degenerates to vmcnt(0) and every "prefetch" is synchronous.
// SPLIT: the output is written in the split-f16 A-operand format of pw_gemm_f16x3_pre_kernel instead of f32 -- per pixel
// and per chunk of 32 channels one 128-byte line [hi(32 halfs) | lo(32 halfs)], hi = f16(v), lo = f16(v - hi): exactly
// the split the pointwise GEMM would apply to the f32 value on its way into LDS (same bytes per element, bit-identical
// GEMM result), which lets that GEMM take its A tiles by LDS-DMA without staging registers or conversion work.  ldy is
// then the number of chunks per pixel; channels c .. 32 * ldy - 1 are written as zeros (the GEMM reads whole chunks).
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

// The split of four output values into the hi / lo dwords of the A-operand line.  1 (default): the packed saturating
// conversions of asr_common.h in a wave that has called asr_enable_f16_saturation() -- 10 - 12 vector instructions instead of
// 24 for the four values, in kernels that are bound by VALU issue; 0: clamp + convert per value (rounds 1 - 3).  Same halves
// for every finite input.
#ifndef ASR_DIAG_DW
#define ASR_DIAG_DW 0                                          // ablation bits (tools/build_hazard_variants.py: dw_skip*); 0 in the product
#endif
#ifndef ASR_DW_PACKED_SPLIT
#define ASR_DW_PACKED_SPLIT 1
#endif
typedef unsigned int dw_u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void dw_split4(const f32x4& acc, dw_u32x2& h2, dw_u32x2& l2) {
#if ASR_DW_PACKED_SPLIT
    unsigned int h01, h23, l01, l23;
    asr_split4_f16_saturating_mode(acc[0], acc[1], acc[2], acc[3], h01, h23, l01, l23);
    h2 = dw_u32x2{h01, h23};
    l2 = dw_u32x2{l01, l23};
#else
    f16x4 hi, lo;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        _Float16 h, l;
        asr_split_f16(acc[e], h, l);
        hi[e] = h;
        lo[e] = l;
    }
    h2 = __builtin_bit_cast(dw_u32x2, hi);
    l2 = __builtin_bit_cast(dw_u32x2, lo);
#endif
}

// ACT: 0 = activations from the runtime flags, 1 = pre-ReLU only (the sepconvs without depth activation), 2 = post-ReLU
// only (with depth activation) -- at ~5 TB/s these kernels are bound by VALU issue (3 waves per SIMD, ~150 instructions
// per output row), and a runtime flag costs a v_max + v_cndmask per loaded element instead of one v_max.
// Streaming (non-temporal) output stores: the depthwise output is consumed by the NEXT kernel, never by this one, and a
// normal store leaves 300 MB marching through L2; same-box A/B over all depthwise launches: 12.75 -> 12.1-12.7 ms (-DASR_DW_NT=0 to compare).
#ifndef ASR_DW_NT
#define ASR_DW_NT 1
#endif
constexpr bool kNtStores = ASR_DW_NT != 0;
// Maps of up to this many output rows are walked in 16-row strips with 4 rows of input in flight and the activation
// pattern compiled in; taller ones in 32-row strips with one row in flight (measured per map size).
#ifndef ASR_DW_SMALL_MAX
#define ASR_DW_SMALL_MAX 64
#endif
template <int R, int S, int SROWS, int PF, bool SPLIT, int ACT>
__global__ __launch_bounds__(256) ASR_PK_F32 void dw_stream_full_kernel(DwArgs p, int tiles_x) {
    constexpr int WIN = 2 * R + 1;
    static_assert(SROWS % PF == 0, "strip length must be a multiple of the prefetch depth");
#if ASR_DW_PACKED_SPLIT
    if (SPLIT) asr_enable_f16_saturation();                   // dw_split4 converts with the hardware's f16 clamp
#endif
    const int tid = threadIdx.x;
    const int c4 = tid & 15, col = tid >> 4;
    const int tx = blockIdx.x % tiles_x, cbk = blockIdx.x / tiles_x;
    const int ch = (cbk * 16 + c4) * 4;
    const int ox = tx * SCOLS + col;
    const int oy0 = blockIdx.y * SROWS;
    const int b = blockIdx.z;
    _Float16* ysplit = nullptr;
    long long split_row_stride = 0;
    if (SPLIT) {
        ysplit = reinterpret_cast<_Float16*>(p.y) + (((long long)b * p.h_out * p.w_out + ox) * p.ldy + (ch >> 5)) * 64 + (ch & 31);
        split_row_stride = (long long)p.w_out * p.ldy * 64;
        if (ox < p.w_out && ch >= p.c && ch < p.ldy * 32) {      // padding channels of the last chunk: zeros, no loads
            const f16x4 z = {(_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f};
            for (int r = 0; r < SROWS; ++r) {
                _Float16* o = ysplit + (long long)(oy0 + r) * split_row_stride;
                *reinterpret_cast<f16x4*>(o) = z;
                *reinterpret_cast<f16x4*>(o + 32) = z;
            }
        }
    }
    if (ch >= p.c || ox >= p.w_out) return;
    const float* xin = p.x + (long long)b * p.h_in * p.w_in * p.ldx + ch;
    float* yout = p.y + ((long long)b * p.h_out * p.w_out + ox) * p.ldy + ch;
    const int ixl = ox * S - p.pad_left, ixc = ixl + R, ixr = ixl + 2 * R;
    const bool vl = ixl >= 0 && ixl < p.w_in, vc = ixc >= 0 && ixc < p.w_in, vr = ixr >= 0 && ixr < p.w_in;
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    // a tap column outside the image (left / right zero padding) is switched off in the WEIGHTS, once per thread, instead
    // of zeroing the (clamped, finite) loaded values row by row
    f32x4 wk[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const bool col_ok = (t % 3 == 0) ? vl : ((t % 3 == 1) ? vc : vr);
        const f32x4 wv = *reinterpret_cast<const f32x4*>(p.w + (long long)t * p.c + ch);
        wk[t] = col_ok ? wv : zero;
    }
    const f32x4 bv = *reinterpret_cast<const f32x4*>(p.bias + ch);
    const long long ofl = (long long)min(max(ixl, 0), p.w_in - 1) * p.ldx, ofc = (long long)min(max(ixc, 0), p.w_in - 1) * p.ldx,
                    ofr = (long long)min(max(ixr, 0), p.w_in - 1) * p.ldx;
    const long long row_stride = (long long)p.w_in * p.ldx;
    const bool pre = ACT == 0 ? p.pre_relu != 0 : ACT == 1;

    auto issue = [&](int iy, f32x4 (&d)[3]) {                 // three unconditional loads from a clamped row
        const float* row = xin + (long long)min(max(iy, 0), p.h_in - 1) * row_stride;
#if ASR_DIAG_DW & 4                                             // ablation: the centre column only (one load per row instead of three)
        d[1] = *reinterpret_cast<const f32x4*>(row + ofc);
        d[0] = d[1]; d[2] = d[1];
#else
        d[0] = *reinterpret_cast<const f32x4*>(row + ofl);     // (non-temporal LOADS are 20 % slower: the two neighbour
        d[1] = *reinterpret_cast<const f32x4*>(row + ofc);     //  columns are L1 hits of other lanes' centre loads)
        d[2] = *reinterpret_cast<const f32x4*>(row + ofr);
#endif
    };
    auto enter = [&](int iy, const f32x4 (&d)[3], f32x4 (&w)[3]) {   // top / bottom zero padding + pre-activation at window entry
        // the row index is the same for the whole workgroup: a scalar branch in the (rare) padding rows, nothing otherwise
        const int vy = __builtin_amdgcn_readfirstlane((int)(iy >= 0 && iy < p.h_in));
#pragma unroll
        for (int k = 0; k < 3; ++k) w[k] = (ACT == 1) ? relu4(d[k]) : ((ACT == 2) ? d[k] : (pre ? relu4(d[k]) : d[k]));
        if (!vy) {
            asm volatile("" ::: "memory");                     // keeps this a branch (if-conversion would put 12 v_cndmask on every row)
            w[0] = zero; w[1] = zero; w[2] = zero;
        }
    };

    f32x4 win[WIN][3], q[PF][S][3];
    const int base0 = oy0 * S - p.pad_top;
#pragma unroll
    for (int k = 0; k < WIN - S; ++k) {
        f32x4 d[3];
        issue(base0 + k, d);
        enter(base0 + k, d, win[S + k]);
    }
#pragma unroll
    for (int j = 0; j < PF; ++j)
#pragma unroll
        for (int t = 0; t < S; ++t) issue(base0 + WIN - S + j * S + t, q[j][t]);
#pragma unroll
    for (int r0 = 0; r0 < SROWS; r0 += PF) {       // fully unrolled: window "shifts" become register renaming
#pragma unroll
        for (int j = 0; j < PF; ++j) {
            const int r = r0 + j;
#pragma unroll
            for (int k = 0; k < WIN - S; ++k) { win[k][0] = win[k + S][0]; win[k][1] = win[k + S][1]; win[k][2] = win[k + S][2]; }
#pragma unroll
            for (int t = 0; t < S; ++t) enter(base0 + r * S + WIN - S + t, q[j][t], win[WIN - S + t]);
#pragma unroll
            for (int t = 0; t < S; ++t) issue(base0 + (r + PF) * S + WIN - S + t, q[j][t]);   // past the strip: clamped re-read, unused
            f32x4 acc = bv;
#if ASR_DIAG_DW & 1                                             // ablation: one tap instead of nine (loads stay)
            acc += win[R][1] * wk[4] + win[0][0] + win[2 * R][2];
#else
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) acc += win[ky * R][kx] * wk[ky * 3 + kx];
#endif
            if (ACT == 2) acc = relu4(acc);
            else if (ACT == 0) acc = post_act4(acc, p.post_relu);
#if ASR_DIAG_DW & 2                                             // ablation: f32 bytes to the split operand's address (no split, no lane trade)
            if (SPLIT) {
                _Float16* o = ysplit + (long long)(oy0 + r) * split_row_stride + ((c4 & 1) ? 32 - 4 : 0);
                __builtin_nontemporal_store(acc, reinterpret_cast<f32x4*>(o));
            } else
#endif
            if (SPLIT) {
                // Lane pairs (channel quads 2j, 2j+1; c % 8 == 0) trade halves through DPP so that each lane issues ONE
                // 16-byte store -- the even lane the 8 hi halfs of both quads, the odd lane their 8 lo halfs -- instead of
                // two 8-byte stores per lane (which cost the split variant 15 % of the kernel's bandwidth).
                typedef dw_u32x2 u32x2;
                typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
                u32x2 h2, l2;
                dw_split4(acc, h2, l2);
                const bool even = (c4 & 1) == 0;
                const u32x2 give = even ? l2 : h2;
                u32x2 got;
                // quad_perm [1,0,3,2] (v_mov_b32_dpp, no LDS crossbar trip: __shfl_xor compiles to ds_bpermute + a wait)
                got.x = (unsigned int)__builtin_amdgcn_mov_dpp((int)give.x, 0xB1, 0xF, 0xF, true);
                got.y = (unsigned int)__builtin_amdgcn_mov_dpp((int)give.y, 0xB1, 0xF, 0xF, true);
                const u32x4 out = even ? u32x4{h2.x, h2.y, got.x, got.y} : u32x4{got.x, got.y, l2.x, l2.y};
                // even lane: hi of channels ch .. ch+7 at its own hi slot; odd lane: lo of channels ch-4 .. ch+3 at the lo slot
                _Float16* o = ysplit + (long long)(oy0 + r) * split_row_stride + (even ? 0 : 32 - 4);
                if (kNtStores) __builtin_nontemporal_store(out, reinterpret_cast<u32x4*>(o)); else *reinterpret_cast<u32x4*>(o) = out;
            } else {
                if (kNtStores) __builtin_nontemporal_store(acc, reinterpret_cast<f32x4*>(yout + (long long)(oy0 + r) * p.w_out * p.ldy)); else *reinterpret_cast<f32x4*>(yout + (long long)(oy0 + r) * p.w_out * p.ldy) = acc;
            }
        }
    }
}


// ---------------------------------------------------------------------------------------------
// fused ASPP: three dilation rates from one LDS-resident phase of the plane
// ---------------------------------------------------------------------------------------------
// The three ASPP rates are multiples of g = gcd(rates) (6 / 12 / 18 at OS 16, 12 / 24 / 36 at OS 8): an output pixel
// (y, x) only ever taps pixels (y + a g, x + b g), i.e. pixels of ITS OWN residue class (y mod g, x mod g).  The plane
// therefore falls apart into g x g independent sub-grids ("phases") on which the three convs are ordinary dilated 3x3
// convs of rates rate / g = 1, 2, 3 -- no halo, no redundancy: every input line is read from HBM exactly once and every
// output line written once, whatever the plane size.  A workgroup owns (32 channels = one 128-byte line per pixel, one
// row phase py, one group of gs = g / nxg adjacent column phases, one image): ceil(H / g) rows x ~W / nxg columns of
// lines in LDS -- 24 KB on the 32 x 32 map of the 512 x 512 inputs (6 workgroups per CU: the staging loads of one overlap
// the tap loops and the stores of the others), 46 KB on the 64 x 64 map of the 1024 x 1024 inputs (nxg = 2), where the
// whole-plane form of round 2 (128 KB at 32 x 32, one workgroup per CU, load and compute phases strictly alternating)
// did not fit at all and the net fell back to three dw_direct launches.
// Taps outside the image read one zero line of LDS (the select is on the ADDRESS: one v_cndmask per tap, no branch).
constexpr int ACB = 32;        // channels per workgroup (128-byte pieces of a pixel row)
constexpr int ATHREADS = 256;  // 8 lanes per line, 32 lines per pass
constexpr int AMAXCOLS = 128;  // columns of one workgroup (x-table size)

struct AsppArgs {
    const float* x;
    const float* w;     // [3 branches][3][3][C]
    const float* bias;  // [3][C]
    float* y[3];
    int batch, h, w_, c;
    int k[3];           // rate / g per branch
    int g, nxg, gs;     // phase period, column-phase groups per row phase, phases per group (g = nxg * gs)
    int ldx, ldy;
    int pre_relu, post_relu;
};

// SPLIT: the three outputs in the split-f16 operand format (see dw_stream_full_kernel); a workgroup's 32 channels are
// exactly one chunk, ldy = chunks per pixel; c % 32 == 0.
template <bool SPLIT>
__global__ __launch_bounds__(ATHREADS) void aspp_dw3_phase_kernel(AsppArgs p) {
#if ASR_DW_PACKED_SPLIT
    if (SPLIT) asr_enable_f16_saturation();                   // dw_split4 converts with the hardware's f16 clamp
#endif
    extern __shared__ __attribute__((aligned(16))) float lds[];  // [rows * cols + 1 zero line][ACB] floats, then int xtab[cols]
    const int tid = threadIdx.x;
    const int c4 = tid & 7, slot = tid >> 3;                     // 8 lanes x 16 bytes = one line; 32 line slots
    const int cb = blockIdx.x * ACB;
    const int py = blockIdx.y / p.nxg, xg = blockIdx.y - py * p.nxg;
    const int b = blockIdx.z;
    const int g = p.g, gs = p.gs, x0 = xg * gs;
    const int rows = (p.h - py + g - 1) / g;                     // image rows y = py + i g
    // columns x = q g + x0 + r (r < gs), ascending in j = q gs + r: the valid ones are the first `cols`
    const int qfull = (p.w_ - x0) / g, rem = (p.w_ - x0) - qfull * g;
    const int cols = (p.w_ > x0) ? qfull * gs + min(max(rem, 0), gs) : 0;
    const int lines = rows * cols;
    if (lines <= 0) return;
    int* xtab = reinterpret_cast<int*>(lds + (size_t)(lines + 1) * ACB);
    for (int j = tid; j < cols; j += ATHREADS) xtab[j] = (j / gs) * g + x0 + (j % gs);
    if (tid < 8) *reinterpret_cast<f32x4*>(lds + (size_t)lines * ACB + tid * 4) = f32x4{0.f, 0.f, 0.f, 0.f};
    const float* xin = p.x + (long long)b * p.h * p.w_ * p.ldx;
    const bool ch_ok = cb + c4 * 4 < p.c;
    {   // stage the phase: UN unconditional loads in flight per thread (clamped line index), then the LDS stores
        constexpr int UN = 8;
        const float* src = xin + (ch_ok ? cb + c4 * 4 : 0);
        const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
        for (int base = slot; base < lines; base += 32 * UN) {
            f32x4 v[UN];
#pragma unroll
            for (int u = 0; u < UN; ++u) {
                const int l = min(base + u * 32, lines - 1);
                const int i = l / cols, j = l - i * cols;
                const int xx = (j / gs) * g + x0 + (j % gs);
                v[u] = *reinterpret_cast<const f32x4*>(src + ((long long)(py + i * g) * p.w_ + xx) * p.ldx);
            }
#pragma unroll
            for (int u = 0; u < UN; ++u) {
                const int l = base + u * 32;
                if (l < lines) *reinterpret_cast<f32x4*>(lds + (size_t)l * ACB + c4 * 4) = ch_ok ? (p.pre_relu ? relu4(v[u]) : v[u]) : zero;
            }
        }
    }
    __syncthreads();
    const int ch = cb + c4 * 4;
    if (ch >= p.c) return;
    const int zoff = lines * ACB + c4 * 4;
#pragma unroll 1
    for (int br = 0; br < 3; ++br) {
        const int k = p.k[br], kj = k * gs, kx_img = k * g;
        const float* wb = p.w + (long long)br * 9 * p.c + ch;
        f32x4 wk[9];
#pragma unroll
        for (int t = 0; t < 9; ++t) wk[t] = *reinterpret_cast<const f32x4*>(wb + (long long)t * p.c);
        const f32x4 bv = *reinterpret_cast<const f32x4*>(p.bias + (long long)br * p.c + ch);
        int i = slot / cols, j = slot - i * cols;
        for (int l = slot; l < lines; l += 32) {
            const int xx = xtab[j];
            const int center = (i * cols + j) * ACB + c4 * 4;
            const bool vyu = i - k >= 0, vyd = i + k < rows, vxl = xx - kx_img >= 0, vxr = xx + kx_img < p.w_;
            const int up = -k * cols * ACB, dn = k * cols * ACB, lf = -kj * ACB, rt = kj * ACB;
            f32x4 acc = bv;
            acc += *reinterpret_cast<const f32x4*>(lds + ((vyu && vxl) ? center + up + lf : zoff)) * wk[0];
            acc += *reinterpret_cast<const f32x4*>(lds + (vyu ? center + up : zoff)) * wk[1];
            acc += *reinterpret_cast<const f32x4*>(lds + ((vyu && vxr) ? center + up + rt : zoff)) * wk[2];
            acc += *reinterpret_cast<const f32x4*>(lds + (vxl ? center + lf : zoff)) * wk[3];
            acc += *reinterpret_cast<const f32x4*>(lds + center) * wk[4];
            acc += *reinterpret_cast<const f32x4*>(lds + (vxr ? center + rt : zoff)) * wk[5];
            acc += *reinterpret_cast<const f32x4*>(lds + ((vyd && vxl) ? center + dn + lf : zoff)) * wk[6];
            acc += *reinterpret_cast<const f32x4*>(lds + (vyd ? center + dn : zoff)) * wk[7];
            acc += *reinterpret_cast<const f32x4*>(lds + ((vyd && vxr) ? center + dn + rt : zoff)) * wk[8];
            acc = post_act4(acc, p.post_relu);
            const long long pix = (long long)b * p.h * p.w_ + (long long)(py + i * g) * p.w_ + xx;
            if (SPLIT) {
                typedef dw_u32x2 u32x2;
                typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
                u32x2 h2, l2;
                dw_split4(acc, h2, l2);
                const bool even = (c4 & 1) == 0;
                const u32x2 give = even ? l2 : h2;
                u32x2 got;
                got.x = (unsigned int)__builtin_amdgcn_mov_dpp((int)give.x, 0xB1, 0xF, 0xF, true);
                got.y = (unsigned int)__builtin_amdgcn_mov_dpp((int)give.y, 0xB1, 0xF, 0xF, true);
                const u32x4 out = even ? u32x4{h2.x, h2.y, got.x, got.y} : u32x4{got.x, got.y, l2.x, l2.y};
                _Float16* o = reinterpret_cast<_Float16*>(p.y[br]) + (pix * p.ldy + (cb >> 5)) * 64 + (even ? c4 * 4 : 32 + (c4 - 1) * 4);
                if (kNtStores) __builtin_nontemporal_store(out, reinterpret_cast<u32x4*>(o)); else *reinterpret_cast<u32x4*>(o) = out;
            } else {
                *reinterpret_cast<f32x4*>(p.y[br] + pix * p.ldy + ch) = acc;
            }
            j += 32;
            while (j >= cols) { j -= cols; ++i; }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// direct (fallback)
// ---------------------------------------------------------------------------------------------
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
        acc = post_act4(acc, p.post_relu);
        *reinterpret_cast<f32x4*>(p.y + ((b * p.h_out + oy) * p.w_out + ox) * p.ldy + ch) = acc;
    }
}

template <int R, int S>
int launch_stream(const DwArgs& p, hipStream_t s, bool split = false) {
    const int tiles_x = (int)asr_cdiv(p.w_out, SCOLS), chunks = (int)asr_cdiv(p.c, 64);
    const int srows = p.h_out <= ASR_DW_SMALL_MAX ? 16 : 32;
    const dim3 grid(tiles_x * chunks, (unsigned)asr_cdiv(p.h_out, srows), p.batch);
    // full strips (every layer of the net): the branch-free kernel; 4 rows in flight on the small OS16 maps (few waves
    // per image), 1 on the large ones (measured); the two activation patterns of the Xception sepconvs are
    // compiled in for the 16-row strips (the middle flow).  Ragged strips take the generic kernel below.
    if (p.h_out % srows == 0) {
        const int act = (srows == 16) ? ((p.pre_relu == 1 && p.post_relu == 0) ? 1 : ((p.pre_relu == 0 && p.post_relu == 1) ? 2 : 0)) : 0;
#define ASR_DW_FULL(SR_, PF_, ACT_)                                                                                              \
    do {                                                                                                                         \
        if (split) hipLaunchKernelGGL((dw_stream_full_kernel<R, S, SR_, PF_, true, ACT_>), grid, dim3(256), 0, s, p, tiles_x);    \
        else hipLaunchKernelGGL((dw_stream_full_kernel<R, S, SR_, PF_, false, ACT_>), grid, dim3(256), 0, s, p, tiles_x);         \
    } while (0)
        if (srows == 32) ASR_DW_FULL(32, 1, 0);
        else if (act == 1) ASR_DW_FULL(16, 4, 1);
        else if (act == 2) ASR_DW_FULL(16, 4, 2);
        else ASR_DW_FULL(16, 4, 0);
#undef ASR_DW_FULL
        return ASR_OK;
    }
    ASR_UNSUPPORTED(split, "asr_dwconv3x3_nhwc_split_f16: needs h_out to be a multiple of the strip height (%d)", srows);
    if (srows == 16) hipLaunchKernelGGL((dw_stream_kernel<R, S, 16>), grid, dim3(256), 0, s, p, tiles_x);
    else hipLaunchKernelGGL((dw_stream_kernel<R, S, 32>), grid, dim3(256), 0, s, p, tiles_x);
    return ASR_OK;
}

}  // namespace

// mode: 0 = auto, 1 = direct, 2 = streaming (register window)
extern "C" int asr_dwconv3x3_nhwc_f32(const float* x, const float* w, const float* bias, float* y, int batch, int h_in,
                                      int w_in, int c, int stride, int rate, int pad_top, int pad_left, int h_out,
                                      int w_out, int ldx, int ldy, int pre_relu, int post_relu, int mode,
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
    ASR_REQUIRE(mode >= 0 && mode <= 2, "asr_dwconv3x3_nhwc_f32: mode must be 0 (auto), 1 (direct) or 2 (streaming)");
    DwArgs p{x, w, bias, y, batch, h_in, w_in, c, h_out, w_out, stride, rate, pad_top, pad_left, pre_relu, post_relu, ldx, ldy};
    hipStream_t s = asr_stream(stream);
    const bool can_stream = batch <= 65535 && ((stride == 1 && (rate == 1 || rate == 2)) || (stride == 2 && rate == 1));
    // (an LDS-tiled kernel and a flat lane mapping were measured at or below the streaming form on every layer of the net,
    // DESIGN.md 4.2, and are not built)
    if (mode == 0) mode = can_stream ? 2 : 1;
    if (mode == 2) {
        ASR_UNSUPPORTED(!can_stream, "asr_dwconv3x3_nhwc_f32: streaming kernel needs (stride 1, rate 1|2) or (stride 2, rate 1)");
        if (stride == 1 && rate == 1) launch_stream<1, 1>(p, s);
        else if (stride == 1) launch_stream<2, 1>(p, s);
        else launch_stream<1, 2>(p, s);
    } else {
        const long long total = (long long)batch * h_out * w_out * (c >> 2);
        const long long g = asr_cdiv(total, 256);
        hipLaunchKernelGGL(dw_direct_kernel, dim3((unsigned)(g < 8192 ? g : 8192)), dim3(256), 0, s, p);
    }
    ASR_LAUNCH_CHECK();
    return ASR_OK;
}

// Depthwise 3x3 whose output feeds asr_pwconv_mfma_f16x3_presplit: y is written as split-f16 chunks (see
// dw_stream_full_kernel), ldy_chunks = ceil(c / 32) chunks of 128 bytes per pixel.
extern "C" int asr_dwconv3x3_nhwc_split_f16(const float* x, const float* w, const float* bias, void* y_split, int batch, int h_in,
                                            int w_in, int c, int stride, int rate, int pad_top, int pad_left, int h_out,
                                            int w_out, int ldx, int ldy_chunks, int pre_relu, int post_relu,
                                            asr_stream_t stream) {
    ASR_REQUIRE(x && w && bias && y_split, "asr_dwconv3x3_nhwc_split_f16: null pointer");
    ASR_REQUIRE(batch > 0 && batch <= 65535 && h_in > 0 && w_in > 0 && c > 0 && h_out > 0 && w_out > 0 && stride > 0 && rate > 0 &&
                    pad_top >= 0 && pad_left >= 0,
                "asr_dwconv3x3_nhwc_split_f16: bad geometry");
    ASR_REQUIRE(ldx >= c && ldy_chunks * 32 >= c, "asr_dwconv3x3_nhwc_split_f16: ldx < c or ldy_chunks * 32 < c");
    ASR_REQUIRE((long long)(h_out - 1) * stride - pad_top <= h_in - 1 && (long long)(w_out - 1) * stride - pad_left <= w_in - 1,
                "asr_dwconv3x3_nhwc_split_f16: output %dx%d does not fit input %dx%d (stride %d)", h_out, w_out, h_in, w_in, stride);
    ASR_UNSUPPORTED((c & 7) || (ldx & 3), "asr_dwconv3x3_nhwc_split_f16: c must be a multiple of 8 and ldx of 4");
    ASR_UNSUPPORTED(ldy_chunks * 32 > ((c + 63) / 64) * 64, "asr_dwconv3x3_nhwc_split_f16: ldy_chunks beyond ceil64(c) / 32");
    ASR_UNSUPPORTED((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(w) | reinterpret_cast<uintptr_t>(bias)) & 15 ||
                        (reinterpret_cast<uintptr_t>(y_split) & 127),
                    "asr_dwconv3x3_nhwc_split_f16: x, w, bias must be 16-byte and y_split 128-byte aligned");
    ASR_UNSUPPORTED(!((stride == 1 && (rate == 1 || rate == 2)) || (stride == 2 && rate == 1)),
                    "asr_dwconv3x3_nhwc_split_f16: needs (stride 1, rate 1|2) or (stride 2, rate 1)");
    DwArgs p{x, w, bias, reinterpret_cast<float*>(y_split), batch, h_in, w_in, c, h_out, w_out, stride, rate, pad_top, pad_left,
             pre_relu, post_relu, ldx, ldy_chunks};
    hipStream_t s = asr_stream(stream);
    int rc;
    if (stride == 1 && rate == 1) rc = launch_stream<1, 1>(p, s, true);
    else if (stride == 1) rc = launch_stream<2, 1>(p, s, true);
    else rc = launch_stream<1, 2>(p, s, true);
    if (rc != ASR_OK) return rc;
    ASR_LAUNCH_CHECK();
    return ASR_OK;
}

static int asr_gcd(int a, int b) { while (b) { const int t = a % b; a = b; b = t; } return a; }

// Geometry of the phase decomposition: g = gcd of the rates; the column phases are split into nxg groups (a divisor of g)
// until a workgroup's lines fit 48 KB of LDS (>= 3 workgroups per CU), or as far as g allows.
static int aspp_geometry(int h, int w, int rate0, int rate1, int rate2, int* g_out, int* nxg_out, size_t* lds_out) {
    const int g = asr_gcd(asr_gcd(rate0, rate1), rate2);
    const int rows = (h + g - 1) / g;
    int nxg = 1;
    size_t lds = 0;
    for (;;) {
        const int gs = g / nxg;
        const int cols = ((w + g - 1) / g) * gs;                     // upper bound over the column-phase groups
        lds = sizeof(float) * ((size_t)rows * cols + 1) * ACB + sizeof(int) * (size_t)cols;
        if ((lds <= 48 * 1024 && cols <= AMAXCOLS) || nxg == g) break;
        int next = nxg + 1;
        while (g % next) ++next;
        nxg = next;
    }
    *g_out = g; *nxg_out = nxg; *lds_out = lds;
    return ((w + g - 1) / g) * (g / nxg) <= AMAXCOLS && lds <= 160 * 1024;
}

extern "C" int asr_aspp_dwconv3_supported(int h, int w, int rate0, int rate1, int rate2) {
    if (h <= 0 || w <= 0 || rate0 <= 0 || rate1 <= 0 || rate2 <= 0) return 0;
    int g = 0, nxg = 0;
    size_t lds = 0;
    return aspp_geometry(h, w, rate0, rate1, rate2, &g, &nxg, &lds) && (long long)g * nxg <= 65535 ? 1 : 0;
}

static int aspp_common(bool split, const float* x, const float* w3, const float* bias3, void* y0, void* y1, void* y2, int batch,
                       int h, int w, int c, int rate0, int rate1, int rate2, int ldx, int ldy, int pre_relu, int post_relu,
                       asr_stream_t stream) {
    ASR_REQUIRE(x && w3 && bias3 && y0 && y1 && y2, "asr_aspp_dwconv3: null pointer");
    ASR_REQUIRE(batch > 0 && batch <= 65535 && h > 0 && w > 0 && c > 0 && rate0 > 0 && rate1 > 0 && rate2 > 0 && ldx >= c &&
                    (split ? ldy * 32 : ldy) >= c,
                "asr_aspp_dwconv3: bad geometry");
    ASR_UNSUPPORTED((c & 3) || (ldx & 3) || (!split && (ldy & 3)), "asr_aspp_dwconv3: c, ldx, ldy must be multiples of 4");
    ASR_UNSUPPORTED(split && (c & 31), "asr_aspp_dwconv3_nhwc_split_f16: c must be a multiple of 32 (got %d)", c);
    ASR_UNSUPPORTED((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(w3) | reinterpret_cast<uintptr_t>(bias3)) & 15,
                    "asr_aspp_dwconv3: x, w3, bias3 must be 16-byte aligned");
    ASR_UNSUPPORTED((reinterpret_cast<uintptr_t>(y0) | reinterpret_cast<uintptr_t>(y1) | reinterpret_cast<uintptr_t>(y2)) & (split ? 127 : 15),
                    "asr_aspp_dwconv3: outputs must be 16-byte (split: 128-byte) aligned");
    int g = 1, nxg = 1;
    size_t lds = 0;
    const int fits = aspp_geometry(h, w, rate0, rate1, rate2, &g, &nxg, &lds);
    ASR_UNSUPPORTED(!fits, "asr_aspp_dwconv3: a phase of the %dx%d plane at rates %d/%d/%d (period %d) needs %zu B of LDS (max 160 KB); "
                    "use asr_dwconv3x3_nhwc_f32 per branch", h, w, rate0, rate1, rate2, g, lds);
    ASR_UNSUPPORTED((long long)g * nxg > 65535, "asr_aspp_dwconv3: too many phases (%d x %d)", g, nxg);
    static AsrDeviceOnce once_f32, once_split;
    if (lds > 64 * 1024) {
        ASR_HIP_CHECK(asr_allow_dynamic_lds(once_f32, reinterpret_cast<const void*>(aspp_dw3_phase_kernel<false>), 160 * 1024));
        ASR_HIP_CHECK(asr_allow_dynamic_lds(once_split, reinterpret_cast<const void*>(aspp_dw3_phase_kernel<true>), 160 * 1024));
    }
    AsppArgs p{};
    p.x = x; p.w = w3; p.bias = bias3;
    p.y[0] = static_cast<float*>(y0); p.y[1] = static_cast<float*>(y1); p.y[2] = static_cast<float*>(y2);
    p.batch = batch; p.h = h; p.w_ = w; p.c = c; p.k[0] = rate0 / g; p.k[1] = rate1 / g; p.k[2] = rate2 / g;
    p.g = g; p.nxg = nxg; p.gs = g / nxg;
    p.ldx = ldx; p.ldy = ldy; p.pre_relu = pre_relu; p.post_relu = post_relu;
    const dim3 grid((unsigned)asr_cdiv(c, ACB), (unsigned)(g * nxg), batch);
    if (split) hipLaunchKernelGGL(aspp_dw3_phase_kernel<true>, grid, dim3(ATHREADS), lds, asr_stream(stream), p);
    else hipLaunchKernelGGL(aspp_dw3_phase_kernel<false>, grid, dim3(ATHREADS), lds, asr_stream(stream), p);
    ASR_LAUNCH_CHECK();
    return ASR_OK;
}

extern "C" int asr_aspp_dwconv3_nhwc_f32(const float* x, const float* w3, const float* bias3, float* y0, float* y1,
                                         float* y2, int batch, int h, int w, int c, int rate0, int rate1, int rate2,
                                         int ldx, int ldy, int pre_relu, int post_relu, asr_stream_t stream) {
    return aspp_common(false, x, w3, bias3, y0, y1, y2, batch, h, w, c, rate0, rate1, rate2, ldx, ldy, pre_relu, post_relu, stream);
}

// asr_aspp_dwconv3_nhwc_f32 with the three outputs as split-f16 GEMM operands (ldy_chunks = c / 32 chunks per pixel).
extern "C" int asr_aspp_dwconv3_nhwc_split_f16(const float* x, const float* w3, const float* bias3, void* y0, void* y1, void* y2,
                                               int batch, int h, int w, int c, int rate0, int rate1, int rate2, int ldx,
                                               int ldy_chunks, int pre_relu, int post_relu, asr_stream_t stream) {
    return aspp_common(true, x, w3, bias3, y0, y1, y2, batch, h, w, c, rate0, rate1, rate2, ldx, ldy_chunks, pre_relu, post_relu,
                       stream);
}
