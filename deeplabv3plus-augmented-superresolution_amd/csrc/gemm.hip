// Pointwise (1x1) convolution and small dense KxK convolution as an LDS-tiled FP32 MFMA GEMM
// for gfx950:  Y[m, n] = act( sum_k A[m, k] * W[k, n] + bias[n] ) (+ residual[m, n]).
//
// Reference layers served (paths in /root/reference): every Conv2D(.., (1,1)) of model.py --
// _SepConv_BN pointwise :500-503, _conv2d_same shortcuts :529-541, aspp0 / image_pooling /
// concat_projection :195-231, feature_projection0 :244-247, logits :303-304 -- and the
// 3x3 stem conv entry_flow_conv1_2 :155-156 (implicit GEMM, one tap per K-tile).
//
// Design (MI355X): v_mfma_f32_32x32x2_f32 (exact f32, 64 FLOP/clk/SIMD).  Block = 4 waves,
// 128 x {128,64,32} output tile, BK = 32, double-buffered LDS, one barrier per K-tile, next
// tile's global loads in flight behind the current tile's 64 MFMAs per wave.
//  * A tile [BM][32] is XOR-swizzled on its 16-byte slots so the per-lane ds_read_b128 of
//    one row's k-quad is bank-conflict free.
//  * W is pre-packed once per model into [K/4][Npad][4] (k-quads interleaved) so a lane reads
//    four consecutive k of one output column with one ds_read_b128; K is padded to 32 and N
//    to 128 with zeros so the B path needs no predicates.
//  * The k -> (mfma step, lane half) assignment is a permutation shared by A and B: lane half
//    h of step t in quad-pair kk holds k = 8*kk + 4*h + t.
//  * blockIdx -> tile mapping is XCD-aware: the tiles that share an A row-panel get
//    consecutive logical ids, and logical ids are dealt so that consecutive ones share an XCD
//    (private L2), using the bijective remap.
#include "gemm_common.h"

namespace {

template <int WM, int WN, int TM, int TN, bool CONV>
__global__ __launch_bounds__(256) void pw_gemm_kernel(PwArgs p) {
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
    constexpr int A_F4 = BM / 32;  // float4 staged per thread per K-tile
    constexpr int B_F4 = BN / 32;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    // ONE staging buffer (32 KB at 128 x 128): five workgroups = five waves per SIMD hide the two staging barriers better
    // than a double-buffered form at two workgroups per CU (measured +9 % on the whole forward pass).
    float* const sA = smem;                   // [BM*32]
    float* const sB = smem + BM * BK;         // [32*BN]

    // XCD-aware bijective remap of the workgroup id (speed only)
    const int nwg = gridDim.x, orig = blockIdx.x;
    const int q8 = nwg >> 3, r8 = nwg & 7, xcd = orig & 7;
    const int lid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (orig >> 3);
    const int tile_m = lid / p.tiles_n, tile_n = lid % p.tiles_n;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int l32 = lane & 31, hh = lane >> 5;

    // ---- per-thread A staging coordinates (fixed over the K loop) --------------------------
    const int a_c4 = tid & 7;
    const float* a_base[A_F4];
    int a_oy[A_F4], a_ox[A_F4];
#pragma unroll
    for (int i = 0; i < A_F4; ++i) {
        const int row = (tid >> 3) + 32 * i;
        const long long m = (long long)tile_m * BM + row;
        a_base[i] = nullptr;
        a_oy[i] = 0;
        a_ox[i] = 0;
        if (m < p.M) {
            if (p.h_out > 0) {  // spatial mapping
                const int ox = (int)(m % p.w_out);
                const long long t = m / p.w_out;
                const int oy = (int)(t % p.h_out);
                const long long b = t / p.h_out;
                if (CONV) {
                    a_oy[i] = oy * p.stride - p.pad;
                    a_ox[i] = ox * p.stride - p.pad;
                    a_base[i] = p.x + b * (long long)p.h_in * p.w_in * p.ldx;
                } else {
                    a_base[i] = p.x + ((b * p.h_in + (long long)oy * p.stride) * p.w_in + (long long)ox * p.stride) * p.ldx;
                }
            } else {
                a_base[i] = p.x + m * p.ldx;
            }
        }
    }
    const float* b_base = p.wp + ((long long)tile_n * BN) * 4;

    f32x4 ra[A_F4], rb[B_F4];

    auto load_tile = [&](int kt) {
        const int k0 = kt * BK;
#pragma unroll
        for (int i = 0; i < A_F4; ++i) {
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (CONV) {
                const int tap = k0 / p.cin, coff = k0 - tap * p.cin;
                const int dy = tap / 3, dx = tap - dy * 3;
                const int iy = a_oy[i] + dy * p.dil, ix = a_ox[i] + dx * p.dil;
                const int k = coff + a_c4 * 4;
                if (a_base[i] && iy >= 0 && iy < p.h_in && ix >= 0 && ix < p.w_in && k < p.cin)
                    v = *reinterpret_cast<const f32x4*>(a_base[i] + ((long long)iy * p.w_in + ix) * p.ldx + k);
            } else {
                const int k = k0 + a_c4 * 4;
                if (a_base[i] && k < p.K) v = *reinterpret_cast<const f32x4*>(a_base[i] + k);
            }
            ra[i] = v;
        }
#pragma unroll
        for (int i = 0; i < B_F4; ++i) {
            const int qq = tid + 256 * i;
            const int kq = qq / BN, n = qq % BN;
            rb[i] = *reinterpret_cast<const f32x4*>(b_base + ((long long)(k0 / 4 + kq) * p.Npad + n) * 4);
        }
    };
    auto store_tile = [&]() {
        float* dA = sA;
        float* dB = sB;
#pragma unroll
        for (int i = 0; i < A_F4; ++i) {
            const int row = (tid >> 3) + 32 * i;
            *reinterpret_cast<f32x4*>(dA + (row * 8 + (a_c4 ^ ((row >> 1) & 7))) * 4) = ra[i];
        }
#pragma unroll
        for (int i = 0; i < B_F4; ++i) {
            const int qq = tid + 256 * i;
            *reinterpret_cast<f32x4*>(dB + qq * 4) = rb[i];
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int KT = (p.K + BK - 1) / BK;
    load_tile(0);
    store_tile();
    __syncthreads();

    for (int kt = 0; kt < KT; ++kt) {
        if (kt + 1 < KT) load_tile(kt + 1);
        const float* cA = sA;
        const float* cB = sB;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const int slot = 2 * kk + hh;
            f32x4 a[TM], b[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int row = (wm * TM + i) * 32 + l32;
                a[i] = *reinterpret_cast<const f32x4*>(cA + (row * 8 + (slot ^ ((row >> 1) & 7))) * 4);
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int col = (wn * TN + j) * 32 + l32;
                b[j] = *reinterpret_cast<const f32x4*>(cB + (slot * BN + col) * 4);
            }
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][t], b[j][t], acc[i][j], 0, 0, 0);
        }
        if (kt + 1 < KT) {
            __syncthreads();      // every wave has finished reading the tile
            store_tile();
            __syncthreads();
        }
    }

    // ---- epilogue: bias, ReLU, residual, store ------------------------------------------------
    // C/D map of the 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
    __syncthreads();                               // all waves are done reading the last K-tile
    static_assert(WM * WN * 32 * TN * 32 <= BM * BK + BK * BN, "epilogue staging does not fit the LDS");
    pw_epilogue<WM, WN, TM, TN>(p, acc, smem, tile_m, tile_n, wave, lane);
}

// =================================================================================================
// Split-f16 ("f16x3") variant: x = hi + lo with hi = f16(x), lo = f16(x - hi) carries 22 mantissa
// bits, and x*w ~= hi_x*hi_w + hi_x*lo_w + lo_x*hi_w (the dropped lo*lo term is < 2^-22 relative)
// runs on v_mfma_f32_32x32x16_f16 with f32 accumulation: 3 MFMAs at 16x the f32-MFMA rate = 5.3x
// less matrix-pipe time for f32-grade results (max error ~1e-6 relative to sum |a||w|, the size of
// f32 summation-order noise).  Activations stay f32 in HBM; the split happens on the way into LDS.
// 128x128 tile, BK = 32, 4 waves, same epilogue.  LDS: A_hi/A_lo [128][32] halfs (16-byte slots
// XOR-swizzled by (row >> 2) & 3), B_hi/B_lo [4 octets][128][8] halfs from the pre-split packed
// weights ([K/8][Npad][8] per plane) -- 32 KB, single buffer, 2 barriers per K-tile.
// =================================================================================================

// in a kernel that has called asr_enable_f16_saturation()
__device__ __forceinline__ void split_f16x8(const f32x4& a, const f32x4& b, f16x8& hi, f16x8& lo) {
    typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
    u32x4_t h, l;
    unsigned int h0, h1, l0, l1;
    asr_split4_f16_saturating_mode(a.x, a.y, a.z, a.w, h0, h1, l0, l1);
    h.x = h0; h.y = h1; l.x = l0; l.y = l1;
    asr_split4_f16_saturating_mode(b.x, b.y, b.z, b.w, h0, h1, l0, l1);
    h.z = h0; h.w = h1; l.z = l0; l.w = l1;
    hi = __builtin_bit_cast(f16x8, h);
    lo = __builtin_bit_cast(f16x8, l);
}

template <int WM, int WN, int TM, int TN, bool CONV>
__global__ __launch_bounds__(WM * WN * 64) void pw_gemm_f16x3_kernel(PwArgs p) {
    constexpr int NT = WM * WN * 64, BM = WM * TM * 32, BN = WN * TN * 32;
    constexpr int PA = BM * 4 / NT;                    // (row, 32-byte k-slot) items of the A tile per thread
    constexpr int PB = BN * 4 / NT;                    // (k-octet, column) items of the B tile per thread and plane
    static_assert((BM * 4) % NT == 0 && (BN * 4) % NT == 0 && (BN & (BN - 1)) == 0, "tile / thread-count mismatch");
    asr_enable_f16_saturation();                       // the A split's conversions saturate in hardware (split_f16x8)
    // dynamic LDS = max(stages, epilogue staging of WM*WN waves x 32 x TN*32 floats), sized by the launcher
    extern __shared__ __attribute__((aligned(16))) float smem[];
    _Float16* const sAh = reinterpret_cast<_Float16*>(smem);   // [BM rows][32]          (+ stage offset)
    _Float16* const sAl = sAh + BM * BK;
    _Float16* const sBh = sAl + BM * BK;                        // [4 octets][BN cols][8]
    _Float16* const sBl = sBh + BK * BN;

    const int nwg = gridDim.x, orig = blockIdx.x;
    const int q8 = nwg >> 3, r8 = nwg & 7, xcd = orig & 7;
    const int lid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (orig >> 3);
    const int tile_m = lid / p.tiles_n, tile_n = lid % p.tiles_n;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int l32 = lane & 31, hh = lane >> 5;

    // A staging: thread owns k-slot (tid % 4) of rows (tid / 4) + i * (NT / 4)
    const int a_oct = tid & 3;
    const float* a_base[PA];
    int a_oy[PA], a_ox[PA];                                     // CONV: top-left input pixel of the 3x3 window
#pragma unroll
    for (int i = 0; i < PA; ++i) {
        const int row = (tid >> 2) + (NT / 4) * i;
        const long long m = (long long)tile_m * BM + row;
        a_base[i] = nullptr;
        a_oy[i] = 0;
        a_ox[i] = 0;
        if (m < p.M) {
            if (p.h_out > 0) {
                const int ox = (int)(m % p.w_out);
                const long long t = m / p.w_out;
                const int oy = (int)(t % p.h_out);
                const long long b = t / p.h_out;
                if (CONV) {
                    a_oy[i] = oy * p.stride - p.pad;
                    a_ox[i] = ox * p.stride - p.pad;
                    a_base[i] = p.x + b * (long long)p.h_in * p.w_in * p.ldx;
                } else {
                    a_base[i] = p.x + ((b * p.h_in + (long long)oy * p.stride) * p.w_in + (long long)ox * p.stride) * p.ldx;
                }
            } else {
                a_base[i] = p.x + m * p.ldx;
            }
        }
    }
    const long long plane = (long long)p.Kpad * p.Npad;          // halfs per weight plane
    const _Float16* const wh = reinterpret_cast<const _Float16*>(p.wp) + (long long)tile_n * BN * 8;
    f32x4 ra[PA][2];
    f16x8 rbh[PB], rbl[PB];
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};

    auto load_tile = [&](int kt) {
        const int k = kt * BK + a_oct * 8;
        if (CONV) {                                            // implicit im2col: K-tile kt lies inside one tap (cin % 32 == 0)
            const int tap = (kt * BK) / p.cin, coff = kt * BK - tap * p.cin + a_oct * 8;
            const int dy = tap / 3, dx = tap - dy * 3;
#pragma unroll
            for (int i = 0; i < PA; ++i) {
                const int iy = a_oy[i] + dy * p.dil, ix = a_ox[i] + dx * p.dil;
                const bool in = a_base[i] && iy >= 0 && iy < p.h_in && ix >= 0 && ix < p.w_in;
                const float* src = a_base[i] + ((long long)iy * p.w_in + ix) * p.ldx + coff;
                ra[i][0] = in ? *reinterpret_cast<const f32x4*>(src) : zero4;
                ra[i][1] = in ? *reinterpret_cast<const f32x4*>(src + 4) : zero4;
            }
        } else {
#pragma unroll
            for (int i = 0; i < PA; ++i) {
                ra[i][0] = (a_base[i] && k < p.K) ? *reinterpret_cast<const f32x4*>(a_base[i] + k) : zero4;
                ra[i][1] = (a_base[i] && k + 4 < p.K) ? *reinterpret_cast<const f32x4*>(a_base[i] + k + 4) : zero4;
            }
        }
#pragma unroll
        for (int i = 0; i < PB; ++i) {
            const int slot = tid + NT * i;
            const int oct = slot / BN, n = slot & (BN - 1);
            const long long off = ((long long)(kt * 4 + oct) * p.Npad + n) * 8;
            rbh[i] = *reinterpret_cast<const f16x8*>(wh + off);
            rbl[i] = *reinterpret_cast<const f16x8*>(wh + plane + off);
        }
    };
    auto store_tile = [&]() {
        constexpr int so = 0;
#pragma unroll
        for (int i = 0; i < PA; ++i) {
            const int row = (tid >> 2) + (NT / 4) * i;
            f16x8 hi, lo;
            split_f16x8(ra[i][0], ra[i][1], hi, lo);
            const int off = so + (row * 4 + (a_oct ^ ((row >> 2) & 3))) * 8;
            *reinterpret_cast<f16x8*>(sAh + off) = hi;
            *reinterpret_cast<f16x8*>(sAl + off) = lo;
        }
#pragma unroll
        for (int i = 0; i < PB; ++i) {
            const int slot = tid + NT * i;
            *reinterpret_cast<f16x8*>(sBh + so + slot * 8) = rbh[i];
            *reinterpret_cast<f16x8*>(sBl + so + slot * 8) = rbl[i];
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int KT = (p.K + BK - 1) / BK;
#ifdef ASR_GEMM_PHASE_PROFILE
    long long ph[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    long long tprev = (long long)__builtin_readcyclecounter();
#endif
    load_tile(0);
    store_tile();
    __syncthreads();
    PHASE_MARK(0);                                             // prologue: first tile in, LDS filled
    for (int kt = 0; kt < KT; ++kt) {
        constexpr int so = 0;
        // All loads of the next tile go out before the MFMAs: a wave stalls ~250 cycles in the vector-memory issue queue
        // per 1 KiB load while every wave of the CU is loading, and spreading the loads between MFMA groups (tried) only
        // spreads that stall over the matrix pipe's busy phase (-15 %).
        if (kt + 1 < KT) load_tile(kt + 1);
        PHASE_MARK(1);                                         // global loads issued
#pragma unroll
        for (int s = 0; s < BK / 16; ++s) {
            const int oct = 2 * s + hh;
            f16x8 ah[TM], al[TM];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int row = (wm * TM + i) * 32 + l32;
                const int off = so + (row * 4 + (oct ^ ((row >> 2) & 3))) * 8;
                ah[i] = *reinterpret_cast<const f16x8*>(sAh + off);
                al[i] = *reinterpret_cast<const f16x8*>(sAl + off);
            }
            if (TN <= 2) {                                 // all fragments first, then a row-major MFMA sweep
                f16x8 bh[TN], bl[TN];
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const int col = (wn * TN + j) * 32 + l32;
                    bh[j] = *reinterpret_cast<const f16x8*>(sBh + so + (oct * BN + col) * 8);
                    bl[j] = *reinterpret_cast<const f16x8*>(sBl + so + (oct * BN + col) * 8);
                }
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i], bh[j], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bl[j], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bh[j], acc[i][j], 0, 0, 0);
                    }
            } else {                                       // wide wave tile: B fragments one column block at a time
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const int col = (wn * TN + j) * 32 + l32;
                    const f16x8 bh = *reinterpret_cast<const f16x8*>(sBh + so + (oct * BN + col) * 8);
                    const f16x8 bl = *reinterpret_cast<const f16x8*>(sBl + so + (oct * BN + col) * 8);
#pragma unroll
                    for (int i = 0; i < TM; ++i) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i], bh, acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bl, acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bh, acc[i][j], 0, 0, 0);
                    }
                }
            }
        }
        PHASE_MARK(2);                                         // LDS fragment reads + MFMA issue
        if (kt + 1 < KT) {
            PHASE_WAIT_VM();
            PHASE_MARK(3);                                     // next tile's global loads landed
            __syncthreads();
            PHASE_MARK(4);                                     // every wave done reading the tile
            store_tile();
            PHASE_WAIT_LGKM();
            PHASE_MARK(5);                                     // split to f16 + LDS stores
            __syncthreads();
            PHASE_MARK(6);                                     // stores of every wave visible
        }
    }
    __syncthreads();
    pw_epilogue<WM, WN, TM, TN>(p, acc, smem, tile_m, tile_n, wave, lane);
#ifdef ASR_GEMM_PHASE_PROFILE
    PHASE_MARK(7);                                             // epilogue
    if (tid == 0 && orig < ASR_PHASE_BLOCKS)
        for (int i = 0; i < 8; ++i) g_phase_cycles[orig * 16 + i] = ph[i];
#endif
}

// =================================================================================================
// Pre-split A operand (asr_pwconv_mfma_f16x3_presplit): the producer of the activations -- the depthwise kernel of a
// separable conv, asr_dwconv3x3_nhwc_split_f16 -- already wrote them as split-f16 chunks, per row and 32-deep K chunk
// one 128-byte line [hi(32) | lo(32)].  Both operands then reach LDS by LDS-DMA (global_load_lds_dwordx4): no staging
// registers, no conversion VALU, no ds_write, which is what lets a 256 x 256 tile (half the staged bytes per flop of
// 128 x 128: the CU takes in only ~20-30 B/clk from L2 under load) run out of LDS alone.  BK = 32.
//
// The workgroup's twelve waves have FIXED ROLES: waves 0-7 only read fragments and issue MFMAs (no vector-memory
// instruction inside the K loop), each 64 x 128 of the tile; waves 8-11 -- one per SIMD -- only request.
//
// LDS = a RING of five 32 KB units (all 160 KB of the CU).  A K-step needs two units: A_k = [256 rows][8 slots of 16 B:
// hi oct 0-3, lo oct 0-3], slot XOR-swizzled by (row >> 1) & 7 (applied on the per-lane SOURCE address, the DMA
// destination is lane-linear), and B_k = B_hi, B_lo [4 oct][256 col][8 halfs].  Unit u (A_k = 2k, B_k = 2k + 1) lives in
// slot u mod 5.  During K-step k the MFMA waves read units 2k and 2k + 1 while the loaders request B_{k+1} (unit 2k + 3)
// and then A_{k+2} (unit 2k + 4) -- the five live units 2k .. 2k + 4 never share a slot -- and wait with
// s_waitcnt vmcnt(8): everything but the 8 youngest pieces (A_{k+2}) has landed, i.e. both operands of step k + 1.  A's
// pieces so have two K-steps to land and B's, requested first, ~1.5.  Round 2's form of this kernel (two 64 KB stages,
// the whole next stage requested and waited for inside ONE K-step: csrc/diag/gemm_diag.hip) had its loaders' last piece
// landing at ~2950 of a 3620-cycle K-step whose MFMAs need 3072: every hiccup of the memory system was a stall of all
// twelve waves at the barrier.
// Three waves per SIMD leave 168 registers per wave: the MFMA waves hold the 128 accumulators and walk their 64 x 128
// tile in two 32-row halves (the A fragments of two row tiles at a time, the B fragments double-buffered one column
// tile ahead; scheduling barriers keep the compiler from hoisting every read, which would spill).  Same MFMA sequence
// per accumulator as every earlier form => bit-identical results.
// =================================================================================================
__global__ __launch_bounds__(768, 3) ASR_PK_F32 void pw_gemm_f16x3_pre_ring_kernel(PwArgs p) {
    constexpr int BM = 256, BN = 256, RT = 4, CT = 8;
    constexpr int UNIT = 32 * 1024, RING = 5, B_PLANE = 4 * BN * 16;      // B unit = hi plane (16 KB) + lo plane
    constexpr int PIECES = 8;                                  // 1 KiB pieces per loader wave and unit (4 loaders x 8 = 32 KB)
    static_assert(BM * 128 == UNIT && 2 * B_PLANE == UNIT, "unit size");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    char* const lds = reinterpret_cast<char*>(smem);

    const int nwg = gridDim.x, orig = blockIdx.x;
    const int q8 = nwg >> 3, r8 = nwg & 7, xcd = orig & 7;
    const int lid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (orig >> 3);
    const int tile_m = lid / p.tiles_n, tile_n = lid % p.tiles_n;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int KT = p.Kpad / BK;

    if (wave >= 8) {
        // ---- loader wave w: pieces 8 w .. 8 w + 7 of every unit ------------------------------------------------------
        const int first = (wave - 8) * PIECES;
        // Source address of a piece = a base that is the same for the whole wave (scalar registers; advanced per K-step by
        // scalar adds) + a 32-bit byte offset per lane, fixed for the whole tile: the request loop then holds no vector
        // instruction at all (a 64-bit vector add per piece before) -- vector instructions of the loader waves take issue
        // slots from the MFMA waves of their SIMD (profiles/r03_gemm_loader_valu_experiment.txt).
        unsigned off_a[PIECES], off_b[PIECES];
        const long long plane_bytes = (long long)p.Kpad * p.Npad * 2;
        const long long kstep_b = (long long)4 * p.Npad * 16;     // byte advance per K-step (A: 128)
        const long long m0 = (long long)tile_m * BM;
        const char* const base_a = reinterpret_cast<const char*>(p.x) + m0 * p.ldx * 128;
        const char* const base_b = reinterpret_cast<const char*>(p.wp) + (long long)tile_n * BN * 16;
#pragma unroll
        for (int j = 0; j < PIECES; ++j) {
            const int pi = first + j;
            {   // A: (row q >> 3, LDS slot q & 7 holding global slot ^ swizzle)
                const int q = pi * 64 + lane, row = q >> 3, slot = (q & 7) ^ ((row >> 1) & 7);
                long long r = row;
                if (m0 + r >= p.M) r = p.M - 1 - m0;             // rows past the end re-read the last row; never stored
                off_a[j] = (unsigned)(r * p.ldx * 128 + slot * 16);
            }
            {   // B: plane (hi, lo), k-octet, column
                const int qb = pi * 64 + lane, plane = qb >> 10, r = qb & 1023, oct = r >> 8, col = r & 255;
                off_b[j] = (unsigned)(plane * plane_bytes + ((long long)oct * p.Npad + col) * 16);
            }
        }
        auto issue_a = [&](int kt, int slot) {
            char* const dst = lds + slot * UNIT + first * 1024;
            const char* const src = base_a + (long long)kt * 128;
#pragma unroll
            for (int j = 0; j < PIECES; ++j) glds16_sbase(src, off_a[j], dst + j * 1024);
        };
        auto issue_b = [&](int kt, int slot) {
            char* const dst = lds + slot * UNIT + first * 1024;
            const char* const src = base_b + kt * kstep_b;
#pragma unroll
            for (int j = 0; j < PIECES; ++j) glds16_sbase(src, off_b[j], dst + j * 1024);
        };
        issue_a(0, 0);
        issue_b(0, 1);
        if (KT > 1) {
            issue_a(1, 2);
            asm volatile("s_waitcnt vmcnt(8)" ::: "memory");    // A_0, B_0 landed; A_1 may still be in flight
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
#ifdef ASR_GEMM_PHASE_PROFILE
        long long ph[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        long long tprev = (long long)__builtin_readcyclecounter();
#endif
        int slot_b = 3, slot_a = 4;                              // slots of units 2 kt + 3 and 2 kt + 4
        // Epilogue operands by LDS-DMA (see the MFMA waves' epilogue below): in the LAST K-step, which requests no more
        // units, the three slots that hold no live unit take the bias row of the tile (1 KB) and the first row tile of the
        // residual; row tiles 1 - 3 follow one barrier each behind the K loop, alternating two 64 KB buffers.
        // A residual row tile = rows wm * 64 + rt * 16 + (0..15) of the four row blocks wm = 0..3 = 64 rows x 1 KB;
        // row r = wm * 16 + i of it goes to half r >> 5 of its buffer, a piece (one wave instruction) = one row, with the
        // 16-byte slots XOR-ed by (r & 15) on the SOURCE side (conflict-free ds_read_b128 of 16 rows x one column quad).
        const bool ep_fast = pw_ep_fast(p);
        const bool res_dma = ep_fast && p.res != nullptr;
        auto issue_res = [&](int rt, int slot_h0, int slot_h1) {
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const int r = (wave - 8) * 16 + j;                     // row of the 64-row staging buffer: wm = r >> 4, i = r & 15
                long long m = (long long)tile_m * BM + (r >> 4) * 64 + rt * 16 + (r & 15);
                if (m >= p.M) m = p.M - 1;                             // never used
                int col = tile_n * BN + ((lane ^ (r & 15)) << 2);
                if (col + 3 >= p.N) col = 0;                           // columns past N: any valid quad, never used
                glds16_sbase(p.res + m * p.ldres, (unsigned)col * 4u, lds + ((r >> 5) ? slot_h1 : slot_h0) * UNIT + (r & 31) * 1024);   // row base: scalar
            }
        };
        for (int kt = 0; kt < KT; ++kt) {
            // both slots were last read in K-step kt - 1, which every MFMA wave left through the previous barrier
            if (kt + 1 < KT) issue_b(kt + 1, slot_b);
            if (kt + 2 < KT) {
                issue_a(kt + 2, slot_a);
                PHASE_MARK(0);
                asm volatile("s_waitcnt vmcnt(8)" ::: "memory");   // all but A_{kt+2}: the operands of step kt + 1 are in
            } else {
                if (kt + 1 == KT && ep_fast) {
                    if (p.bias && wave == 8) {
                        int col = tile_n * BN + lane * 4;
                        if (col + 3 >= p.N) col = 0;
                        glds16_sbase(p.bias, (unsigned)col * 4u, lds + slot_a * UNIT);
                    }
                    if (res_dma) issue_res(0, slot_b - 1 < 0 ? RING - 1 : slot_b - 1, slot_b);
                }
                PHASE_MARK(0);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            PHASE_MARK(1);
            __builtin_amdgcn_s_barrier();
            PHASE_MARK(2);
            if (kt + 1 < KT) {
                slot_b = slot_b + 2 >= RING ? slot_b + 2 - RING : slot_b + 2;
                slot_a = slot_a + 2 >= RING ? slot_a + 2 - RING : slot_a + 2;
            }
        }
        if (res_dma) {
            // slot_b / slot_a still name the free slots of the last K-step: buffer 0 = (slot_b - 1, slot_b), bias = slot_a,
            // buffer 1 = the two slots of the last K-step's own units (slot_b + 2, slot_b + 3), read for the last time
            // before the barrier above
            const int b0h0 = slot_b - 1 < 0 ? RING - 1 : slot_b - 1, b0h1 = slot_b;
            const int b1h0 = (slot_b + 2) % RING, b1h1 = (slot_b + 3) % RING;
#pragma unroll 1
            for (int rt = 1; rt < 4; ++rt) {
                if (rt & 1) issue_res(rt, b1h0, b1h1); else issue_res(rt, b0h0, b0h1);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
            }
        }
#ifdef ASR_GEMM_PHASE_PROFILE
        if (tid == 512 && orig < ASR_PHASE_BLOCKS)
            for (int i = 0; i < 3; ++i) g_phase_cycles[orig * 16 + 8 + i] = ph[i];
#endif
        return;
    }

    // ---- MFMA wave ------------------------------------------------------------------------------------------------
    // Waves w and w + 4 share a SIMD: they get the two column halves of the same row block, so that a padded last N-tile
    // (N = 728: 2 of the 8 column tiles of the right half are pure padding, and are skipped) shortens every SIMD's K-step alike.
    const int wm = wave & 3, wn = wave >> 2;
    const int ct_valid = min(CT, max(0, (p.N - (tile_n * BN + wn * (CT * 16)) + 15) >> 4));   // column tiles holding real columns
    f32x4 acc[RT][CT];
#pragma unroll
    for (int i = 0; i < RT; ++i)
#pragma unroll
        for (int j = 0; j < CT; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[i][j][e] = 0.f;

#ifdef ASR_GEMM_PHASE_PROFILE
    long long ph[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    long long tprev = (long long)__builtin_readcyclecounter();
#endif
    __builtin_amdgcn_s_barrier();                              // units 0 and 1 landed (the loaders waited for them)
    PHASE_MARK(0);
#ifdef ASR_GEMM_PHASE_PROFILE
    const long long loop_c0 = (long long)__builtin_readcyclecounter(), loop_r0 = (long long)__builtin_amdgcn_s_memrealtime();
#endif
    int slot = 0;                                              // slot of unit 2 kt (A); B sits in the next slot of the ring
    auto kloop = [&](auto CTV_) {
        constexpr int CTV = decltype(CTV_)::value;             // column tiles computed by this wave (even; 8 = all)
        for (int kt = 0; kt < KT; ++kt) {
            const int off_a = slot * UNIT, off_b = (slot + 1 >= RING ? 0 : slot + 1) * UNIT;    // wave-uniform
            slot = slot + 2 >= RING ? slot + 2 - RING : slot + 2;
            // The three lane-dependent byte offsets of the fragment reads (A hi, A lo, B) are RE-DERIVED from the lane id in
            // every K-step (a dozen VALU instructions against ~3000 cycles of MFMAs): the accumulators and fragments leave
            // no registers to keep them in, and the compiler would otherwise hoist them out of the loop and spill them --
            // three scratch reloads with their waits at the head of every K-step.  The empty asm makes the lane id opaque.
            int ln = lane;
            asm volatile("" : "+v"(ln));
            const int l16 = ln & 15, oct = ln >> 4;            // A: row = l16, k = 8 oct ..; B: column = l16, same k
            // row tile t of this wave starts at row (wm * RT + t) * 16: (row >> 1) & 7 == (l16 >> 1) & 7 for every t
            const int swz = (l16 >> 1) & 7;
            const int a_row = off_a + (wm * RT * 16 + l16) * 128;
            const int a_hi = a_row + ((oct ^ swz) << 4), a_lo = a_row + (((4 + oct) ^ swz) << 4);
            const int b_col = off_b + (oct * BN + wn * CT * 16 + l16) * 16;
            // 2 halves x CTV column tiles = 2 CTV groups of 6 MFMAs (16 groups for a full tile); the fragments of group g + 1 are requested before the MFMAs of
            // group g (one B double buffer, the A pair of the second half is requested under the last group of the first),
            // and a scheduling barrier per group keeps the compiler from hoisting every read to the top (it would need 160
            // fragment registers and spill).
            f16x8 ah[2], al[2], bh[2], bl[2];
            auto read_a = [&](int half, int i) {                   // hi / lo fragments of row tile 2 * half + i
                ah[i] = *reinterpret_cast<const f16x8*>(lds + a_hi + (2 * half + i) * 2048);
                al[i] = *reinterpret_cast<const f16x8*>(lds + a_lo + (2 * half + i) * 2048);
            };
            auto read_b = [&](int j, int buf) {
                bh[buf] = *reinterpret_cast<const f16x8*>(lds + b_col + j * 256);
                bl[buf] = *reinterpret_cast<const f16x8*>(lds + b_col + j * 256 + B_PLANE);
            };
            read_a(0, 0);
            read_a(0, 1);
            read_b(0, 0);
            auto group = [&](auto G) {
                constexpr int g = decltype(G)::value, half = g / CTV, j = g % CTV;
                constexpr bool last_of_half0 = g == CTV - 1;
                if (g + 1 < 2 * CTV) read_b((g + 1) % CTV, (g + 1) & 1);
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    f32x4& a4 = acc[2 * half + i][j];
                    // the WEIGHTS fragment is the first operand: the MFMA then yields the transposed 16 x 16 tile -- lane
                    // (l16, q4) holds row l16 and the four CONSECUTIVE columns 4 q4 .. 4 q4 + 3 -- which the epilogue
                    // stores with one 16-byte store per register quad, no LDS transposition.  Same products (x * w
                    // commutes), same k order, same accumulation sequence as the untransposed form: bit-identical.
                    a4 = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh[g & 1], al[i], a4, 0, 0, 0);
                    a4 = __builtin_amdgcn_mfma_f32_16x16x32_f16(bl[g & 1], ah[i], a4, 0, 0, 0);
                    a4 = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh[g & 1], ah[i], a4, 0, 0, 0);
                    if (last_of_half0) read_a(1, i);               // the second half's row tile into the registers just consumed
                }
                // order inside the group: the next group's fragment reads first, then the MFMAs (the A refills behind their rows)
                if (g + 1 < 2 * CTV) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
                if (last_of_half0) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
                if (last_of_half0) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                __builtin_amdgcn_sched_barrier(0);
            };
            asr_static_for<2 * CTV>(group);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");    // this wave's reads of the two units are done
            PHASE_MARK(2);
            __builtin_amdgcn_s_barrier();
            PHASE_MARK(6);
        }
    };
    if (ct_valid > 6) kloop(std::integral_constant<int, 8>{});
    else if (ct_valid > 4) kloop(std::integral_constant<int, 6>{});
    else if (ct_valid > 2) kloop(std::integral_constant<int, 4>{});
    else kloop(std::integral_constant<int, 2>{});
#ifdef ASR_GEMM_PHASE_PROFILE
    if (tid == 0 && orig < ASR_PHASE_BLOCKS) {
        g_phase_cycles[orig * 16 + 11] = (long long)__builtin_readcyclecounter() - loop_c0;
        g_phase_cycles[orig * 16 + 12] = (long long)__builtin_amdgcn_s_memrealtime() - loop_r0;
    }
#endif
    // ---- epilogue: bias, ReLU, residual, store -- no vector-memory LOAD and no LDS transposition -------------------------
    // The transposed accumulators go straight out: per register quad one 16-byte store, 64 contiguous bytes per row and
    // instruction (the next column tile completes the 128-byte line).  What the epilogue has to READ -- the bias row and,
    // in the last layer of a block, the residual tile -- was put into LDS by the loader waves (above), so the only waits of
    // these waves are lgkmcnt waits on ds_reads: on gfx9 one vmcnt counter counts loads AND stores in issue order, and a
    // wave that waits for a residual load also waits for every older store to reach a memory system that all 256 CUs are
    // writing to at once (the direct-load form of this epilogue measured 1.16x the staged one with a residual,
    // profiles/r02_gemm_epilogue_experiments.txt #4; without one 0.95x).  `slot` = the first free slot of the last K-step.
    pw_epilogue16_ring<RT, CT>(p, acc, lds, slot, UNIT, RING, (long long)tile_m * BM + wm * (RT * 16), tile_n * BN + wn * (CT * 16),
                              wm, wn);
#ifdef ASR_GEMM_PHASE_PROFILE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    PHASE_MARK(7);
    if (tid == 0 && orig < ASR_PHASE_BLOCKS)
        for (int i = 0; i < 8; ++i) g_phase_cycles[orig * 16 + i] = ph[i];
#endif
}


// The same kernel as a PERSISTENT walk for launches without a residual (with one, the epilogue's staging owns the free slots:
// built and measured equal to the one-tile kernel, 0 ... -2 %, not kept): a workgroup takes tiles wg, wg + G, wg + 2 G, ...
// and the ring runs on across the tile boundary -- unit u of the next tile lives where unit 2 KT + u of the current one
// would -- so that the next tile's units 0 and 1 are requested in the current tile's LAST K-step (whose three free slots
// held only the bias row) and have landed long before its epilogue ends: no prologue (and no workgroup dispatch) between
// tiles.  One barrier after the epilogue tells the loaders that the bias row is no longer read, then unit 2 of the next
// tile takes its slot.  Same MFMA sequence per accumulator, same epilogue: bit-identical to the one-tile kernel.
__global__ __launch_bounds__(768, 3) ASR_PK_F32 void pw_gemm_f16x3_pre_ring_persist_kernel(PwArgs p, int ntiles) {
    constexpr int BM = 256, BN = 256, RT = 4, CT = 8;
    constexpr int UNIT = 32 * 1024, RING = 5, B_PLANE = 4 * BN * 16;      // B unit = hi plane (16 KB) + lo plane
    constexpr int PIECES = 8;                                  // 1 KiB pieces per loader wave and unit (4 loaders x 8 = 32 KB)
    static_assert(BM * 128 == UNIT && 2 * B_PLANE == UNIT, "unit size");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    char* const lds = reinterpret_cast<char*>(smem);

    const int G = gridDim.x, wg = blockIdx.x;                 // G % 8 == 0: every tile of a workgroup has its XCD (wg & 7)
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int KT = p.Kpad / BK;
    auto tile_of = [&](int t, int& tile_m, int& tile_n) {     // the bijective XCD remap of the one-tile kernel, over all tiles
        const int q8 = ntiles >> 3, r8 = ntiles & 7, xcd = t & 7;
        const int lid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (t >> 3);
        tile_m = lid / p.tiles_n;
        tile_n = lid % p.tiles_n;
    };

    if (wave >= 8) {
        // ---- loader wave w: pieces 8 w .. 8 w + 7 of every unit ------------------------------------------------------
        const int first = (wave - 8) * PIECES;
        unsigned off_a[PIECES], off_b[PIECES];
        const long long plane_bytes = (long long)p.Kpad * p.Npad * 2;
        const long long kstep_b = (long long)4 * p.Npad * 16;     // byte advance per K-step (A: 128)
        const char* base_a = nullptr;
        const char* base_b = nullptr;
        int tile_m = 0, tile_n = 0;
        auto setup = [&](int t) {                                 // scalar bases + 32-bit lane offsets of tile t (see the one-tile kernel)
            tile_of(t, tile_m, tile_n);
            const long long m0 = (long long)tile_m * BM;
            base_a = reinterpret_cast<const char*>(p.x) + m0 * p.ldx * 128;
            base_b = reinterpret_cast<const char*>(p.wp) + (long long)tile_n * BN * 16;
#pragma unroll
            for (int j = 0; j < PIECES; ++j) {
                const int pi = first + j;
                {
                    const int q = pi * 64 + lane, row = q >> 3, slot = (q & 7) ^ ((row >> 1) & 7);
                    long long r = row;
                    if (m0 + r >= p.M) r = p.M - 1 - m0;
                    off_a[j] = (unsigned)(r * p.ldx * 128 + slot * 16);
                }
                {
                    const int qb = pi * 64 + lane, plane = qb >> 10, r = qb & 1023, oct = r >> 8, col = r & 255;
                    off_b[j] = (unsigned)(plane * plane_bytes + ((long long)oct * p.Npad + col) * 16);
                }
            }
        };
        auto issue_a = [&](int kt, int slot) {
            char* const dst = lds + slot * UNIT + first * 1024;
            const char* const src = base_a + (long long)kt * 128;
#pragma unroll
            for (int j = 0; j < PIECES; ++j) glds16_sbase(src, off_a[j], dst + j * 1024);
        };
        auto issue_b = [&](int kt, int slot) {
            char* const dst = lds + slot * UNIT + first * 1024;
            const char* const src = base_b + kt * kstep_b;
#pragma unroll
            for (int j = 0; j < PIECES; ++j) glds16_sbase(src, off_b[j], dst + j * 1024);
        };
        auto ring = [&](int v) { return v % RING; };
        const bool ep_fast = pw_ep_fast(p);
        int t = wg, s0 = 0;                                      // unit u of the current tile lives in slot (s0 + u) % RING
        setup(t);
        issue_a(0, 0);                                           // only the FIRST tile has a prologue
        issue_b(0, 1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        for (; t < ntiles; t += G) {
            const bool has_next = t + G < ntiles;
            int slot_b = ring(s0 + 3), slot_a = ring(s0 + 4);    // slots of units 2 kt + 3 and 2 kt + 4
            for (int kt = 0; kt < KT; ++kt) {
                // A_1 is requested here, not with A_0 / B_0: its slot held the previous tile's bias row until the barrier
                // that ended that tile's epilogue
                if (kt == 0) issue_a(1, ring(s0 + 2));
                if (kt + 1 < KT) issue_b(kt + 1, slot_b);
                if (kt + 2 < KT) {
                    issue_a(kt + 2, slot_a);
                    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");   // all but A_{kt+2}: the operands of step kt + 1 are in
                } else {
                    if (kt + 1 == KT) {
                        // last K-step: no more units of this tile.  Of the three free slots, (2 KT + 2) takes the bias row as in
                        // the one-tile kernel, and (2 KT), (2 KT + 1) take units 0 and 1 of the NEXT tile: they land under this
                        // tile's last MFMAs and its epilogue, the next tile has no prologue.
                        if (ep_fast && p.bias && wave == 8) {
                            int col = tile_n * BN + lane * 4;
                            if (col + 3 >= p.N) col = 0;
                            glds16_sbase(p.bias, (unsigned)col * 4u, lds + slot_a * UNIT);
                        }
                        if (has_next) {
                            setup(t + G);
                            issue_a(0, slot_b - 1 < 0 ? RING - 1 : slot_b - 1);
                            issue_b(0, slot_b);
                            // the epilogue may start when this tile's operands and the bias row are in: everything but the
                            // 16 youngest requests (the next tile's units, waited for below)
                            asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
                        } else {
                            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                        }
                    } else {
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    }
                }
                __builtin_amdgcn_s_barrier();
                if (kt + 1 < KT) {
                    slot_b = slot_b + 2 >= RING ? slot_b + 2 - RING : slot_b + 2;
                    slot_a = slot_a + 2 >= RING ? slot_a + 2 - RING : slot_a + 2;
                }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // units 0 and 1 of the next tile are in
            __builtin_amdgcn_s_barrier();                        // the MFMA waves are done with the bias row (epilogue)
            s0 = ring(s0 + 2 * KT);
        }
        return;
    }

    // ---- MFMA wave ------------------------------------------------------------------------------------------------
    // Waves w and w + 4 share a SIMD: they get the two column halves of the same row block, so that a padded last N-tile
    // (N = 728: 2 of the 8 column tiles of the right half are pure padding, and are skipped) shortens every SIMD's K-step alike.
    const int wm = wave & 3, wn = wave >> 2;
    __builtin_amdgcn_s_barrier();                              // units 0 and 1 of the first tile landed (the loaders waited for them)
    int slot = 0;                                              // slot of unit 2 kt (A); B sits in the next slot of the ring; runs on across tiles
    for (int t = wg; t < ntiles; t += G) {
    int tile_m, tile_n;
    tile_of(t, tile_m, tile_n);
    const int ct_valid = min(CT, max(0, (p.N - (tile_n * BN + wn * (CT * 16)) + 15) >> 4));   // column tiles holding real columns
    f32x4 acc[RT][CT];
#pragma unroll
    for (int i = 0; i < RT; ++i)
#pragma unroll
        for (int j = 0; j < CT; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[i][j][e] = 0.f;

    auto kloop = [&](auto CTV_) {
        constexpr int CTV = decltype(CTV_)::value;             // column tiles computed by this wave (even; 8 = all)
        for (int kt = 0; kt < KT; ++kt) {
            const int off_a = slot * UNIT, off_b = (slot + 1 >= RING ? 0 : slot + 1) * UNIT;    // wave-uniform
            slot = slot + 2 >= RING ? slot + 2 - RING : slot + 2;
            // The three lane-dependent byte offsets of the fragment reads (A hi, A lo, B) are RE-DERIVED from the lane id in
            // every K-step (a dozen VALU instructions against ~3000 cycles of MFMAs): the accumulators and fragments leave
            // no registers to keep them in, and the compiler would otherwise hoist them out of the loop and spill them --
            // three scratch reloads with their waits at the head of every K-step.  The empty asm makes the lane id opaque.
            int ln = lane;
            asm volatile("" : "+v"(ln));
            const int l16 = ln & 15, oct = ln >> 4;            // A: row = l16, k = 8 oct ..; B: column = l16, same k
            // row tile t of this wave starts at row (wm * RT + t) * 16: (row >> 1) & 7 == (l16 >> 1) & 7 for every t
            const int swz = (l16 >> 1) & 7;
            const int a_row = off_a + (wm * RT * 16 + l16) * 128;
            const int a_hi = a_row + ((oct ^ swz) << 4), a_lo = a_row + (((4 + oct) ^ swz) << 4);
            const int b_col = off_b + (oct * BN + wn * CT * 16 + l16) * 16;
            // 2 halves x CTV column tiles = 2 CTV groups of 6 MFMAs (16 groups for a full tile); the fragments of group g + 1 are requested before the MFMAs of
            // group g (one B double buffer, the A pair of the second half is requested under the last group of the first),
            // and a scheduling barrier per group keeps the compiler from hoisting every read to the top (it would need 160
            // fragment registers and spill).
            f16x8 ah[2], al[2], bh[2], bl[2];
            auto read_a = [&](int half, int i) {                   // hi / lo fragments of row tile 2 * half + i
                ah[i] = *reinterpret_cast<const f16x8*>(lds + a_hi + (2 * half + i) * 2048);
                al[i] = *reinterpret_cast<const f16x8*>(lds + a_lo + (2 * half + i) * 2048);
            };
            auto read_b = [&](int j, int buf) {
                bh[buf] = *reinterpret_cast<const f16x8*>(lds + b_col + j * 256);
                bl[buf] = *reinterpret_cast<const f16x8*>(lds + b_col + j * 256 + B_PLANE);
            };
            read_a(0, 0);
            read_a(0, 1);
            read_b(0, 0);
            auto group = [&](auto G) {
                constexpr int g = decltype(G)::value, half = g / CTV, j = g % CTV;
                constexpr bool last_of_half0 = g == CTV - 1;
                if (g + 1 < 2 * CTV) read_b((g + 1) % CTV, (g + 1) & 1);
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    f32x4& a4 = acc[2 * half + i][j];
                    // the WEIGHTS fragment is the first operand: the MFMA then yields the transposed 16 x 16 tile -- lane
                    // (l16, q4) holds row l16 and the four CONSECUTIVE columns 4 q4 .. 4 q4 + 3 -- which the epilogue
                    // stores with one 16-byte store per register quad, no LDS transposition.  Same products (x * w
                    // commutes), same k order, same accumulation sequence as the untransposed form: bit-identical.
                    a4 = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh[g & 1], al[i], a4, 0, 0, 0);
                    a4 = __builtin_amdgcn_mfma_f32_16x16x32_f16(bl[g & 1], ah[i], a4, 0, 0, 0);
                    a4 = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh[g & 1], ah[i], a4, 0, 0, 0);
                    if (last_of_half0) read_a(1, i);               // the second half's row tile into the registers just consumed
                }
                // order inside the group: the next group's fragment reads first, then the MFMAs (the A refills behind their rows)
                if (g + 1 < 2 * CTV) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
                if (last_of_half0) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
                if (last_of_half0) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                __builtin_amdgcn_sched_barrier(0);
            };
            asr_static_for<2 * CTV>(group);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");    // this wave's reads of the two units are done
            __builtin_amdgcn_s_barrier();
        }
    };
    if (ct_valid > 6) kloop(std::integral_constant<int, 8>{});
    else if (ct_valid > 4) kloop(std::integral_constant<int, 6>{});
    else if (ct_valid > 2) kloop(std::integral_constant<int, 4>{});
    else kloop(std::integral_constant<int, 2>{});
    // ---- epilogue: bias, ReLU, residual, store -- no vector-memory LOAD and no LDS transposition -------------------------
    // The transposed accumulators go straight out: per register quad one 16-byte store, 64 contiguous bytes per row and
    // instruction (the next column tile completes the 128-byte line).  What the epilogue has to READ -- the bias row and,
    // in the last layer of a block, the residual tile -- was put into LDS by the loader waves (above), so the only waits of
    // these waves are lgkmcnt waits on ds_reads: on gfx9 one vmcnt counter counts loads AND stores in issue order, and a
    // wave that waits for a residual load also waits for every older store to reach a memory system that all 256 CUs are
    // writing to at once (the direct-load form of this epilogue measured 1.16x the staged one with a residual,
    // profiles/r02_gemm_epilogue_experiments.txt #4; without one 0.95x).  `slot` = the first free slot of the last K-step.
    pw_epilogue16_ring<RT, CT>(p, acc, lds, slot, UNIT, RING, (long long)tile_m * BM + wm * (RT * 16), tile_n * BN + wn * (CT * 16),
                              wm, wn);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // this wave's reads of the bias row are done
    __builtin_amdgcn_s_barrier();                              // ... and everybody's: the loaders may overwrite its slot (A_1 of the next tile)
    }
}



// w [K][N] f32 -> two half planes [Kpad/8][Npad][8] (hi then lo), zero padded
__global__ __launch_bounds__(256) void pack_weights_f16x3_kernel(const float* __restrict__ w, _Float16* __restrict__ wp, int K,
                                                                 int N, int Kpad, int Npad) {
    const long long total = (long long)Kpad * Npad;
    for (long long o = (long long)blockIdx.x * 256 + threadIdx.x; o < total; o += (long long)gridDim.x * 256) {
        const int kr = (int)(o & 7);
        const long long t = o >> 3;
        const int n = (int)(t % Npad);
        const int k = (int)(t / Npad) * 8 + kr;
        const float v = (k < K && n < N) ? w[(long long)k * N + n] : 0.f;
        const _Float16 h = (_Float16)v;
        wp[o] = h;
        wp[total + o] = (_Float16)(v - (float)h);
    }
}

__global__ __launch_bounds__(256) void pack_weights_kernel(const float* __restrict__ w, float* __restrict__ wp, int K, int N,
                                                           int Kpad, int Npad) {
    const long long total = (long long)Kpad * Npad;
    for (long long o = (long long)blockIdx.x * 256 + threadIdx.x; o < total; o += (long long)gridDim.x * 256) {
        const int kr = (int)(o & 3);
        const long long t = o >> 2;
        const int n = (int)(t % Npad);
        const int k = (int)(t / Npad) * 4 + kr;
        wp[o] = (k < K && n < N) ? w[(long long)k * N + n] : 0.f;
    }
}

template <int WM, int WN, int TM, int TN, bool CONV>
int launch(const PwArgs& a, hipStream_t s) {
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
    PwArgs p = a;
    p.tiles_n = (int)asr_cdiv(p.N, BN);
    const long long tiles_m = asr_cdiv(p.M, BM);
    const long long nwg = tiles_m * p.tiles_n;
    if (nwg > 0x7fffffffLL) {
        asr_set_error("asr_pwconv_mfma_f32: grid too large (%lld workgroups)", nwg);
        return ASR_ERR_INVALID_ARG;
    }
    const size_t lds = sizeof(float) * (BM * BK + BK * BN);
    auto kern = pw_gemm_kernel<WM, WN, TM, TN, CONV>;
    static AsrDeviceOnce once;
    ASR_HIP_CHECK(asr_allow_dynamic_lds(once, reinterpret_cast<const void*>(kern), (int)lds));
    hipLaunchKernelGGL(kern, dim3((unsigned)nwg), dim3(256), lds, s, p);
    ASR_LAUNCH_CHECK();
    return ASR_OK;
}

template <bool CONV>
int dispatch(const PwArgs& a, hipStream_t s) {
    if (a.N <= 32) return launch<4, 1, 1, 1, CONV>(a, s);
    if (a.N <= 64) return launch<2, 2, 2, 1, CONV>(a, s);
    return launch<2, 2, 2, 2, CONV>(a, s);
}

}  // namespace

extern "C" size_t asr_pwconv_packed_floats(int k, int n) {
    if (k <= 0 || n <= 0) return 0;
    return (size_t)round_up(k, BK) * (size_t)round_up(n, 128);
}

extern "C" int asr_pwconv_pack_weights_f32(const float* w_kn, float* w_packed, int k, int n, asr_stream_t stream) {
    ASR_REQUIRE(w_kn && w_packed, "asr_pwconv_pack_weights_f32: null pointer");
    ASR_REQUIRE(k > 0 && n > 0, "asr_pwconv_pack_weights_f32: bad shape k=%d n=%d", k, n);
    const int Kpad = round_up(k, BK), Npad = round_up(n, 128);
    const long long total = (long long)Kpad * Npad;
    const int grid = (int)(asr_cdiv(total, 256) < 4096 ? asr_cdiv(total, 256) : 4096);
    hipLaunchKernelGGL(pack_weights_kernel, dim3(grid), dim3(256), 0, asr_stream(stream), w_kn, w_packed, k, n, Kpad, Npad);
    ASR_LAUNCH_CHECK();
    return ASR_OK;
}

static int check_common(const char* fn, const float* x, const float* wp, float* y, long long m, int k, int n, int ldx, int ldy,
                        const float* res, int ldres) {
    ASR_REQUIRE(x && wp && y, "%s: null pointer", fn);
    ASR_REQUIRE(m > 0 && k > 0 && n > 0, "%s: bad shape m=%lld k=%d n=%d", fn, m, k, n);
    ASR_REQUIRE(ldy >= n && (!res || ldres >= n), "%s: ldy/ldres smaller than n", fn);
    ASR_UNSUPPORTED((ldx & 3) || (reinterpret_cast<uintptr_t>(x) & 15) || (reinterpret_cast<uintptr_t>(wp) & 15),
                    "%s: x rows must be 16-byte aligned (ldx %% 4 == 0, base aligned)", fn);
    return ASR_OK;
}

extern "C" int asr_pwconv_mfma_f32(const float* x, const float* w_packed, const float* bias, const float* residual, float* y,
                                   int64_t m, int k, int n, int ldx, int ldy, int ldres, int relu, int sub_stride,
                                   int h_in, int w_in, asr_stream_t stream) {
    int rc = check_common("asr_pwconv_mfma_f32", x, w_packed, y, m, k, n, ldx, ldy, residual, ldres);
    if (rc != ASR_OK) return rc;
    ASR_REQUIRE(ldx >= k, "asr_pwconv_mfma_f32: ldx < k");
    ASR_UNSUPPORTED(k & 3, "asr_pwconv_mfma_f32: k must be a multiple of 4 (got %d)", k);
    PwArgs a{};
    a.x = x; a.wp = w_packed; a.bias = bias; a.res = residual; a.y = y;
    a.M = m; a.K = k; a.N = n; a.Npad = round_up(n, 128);
    a.ldx = ldx; a.ldy = ldy; a.ldres = ldres; a.relu = relu;
    a.taps = 1; a.cin = k; a.stride = 1; a.pad = 0; a.dil = 1;
    if (sub_stride > 1) {
        ASR_REQUIRE(h_in > 0 && w_in > 0, "asr_pwconv_mfma_f32: h_in/w_in required with sub_stride");
        a.h_in = h_in; a.w_in = w_in; a.stride = sub_stride;
        a.h_out = (h_in + sub_stride - 1) / sub_stride;
        a.w_out = (w_in + sub_stride - 1) / sub_stride;
        ASR_REQUIRE(m % ((long long)a.h_out * a.w_out) == 0, "asr_pwconv_mfma_f32: m is not a whole number of %dx%d maps",
                    a.h_out, a.w_out);
    }
    return dispatch<false>(a, asr_stream(stream));
}

extern "C" int asr_conv3x3_mfma_f32(const float* x, const float* w_packed, const float* bias, float* y, int batch, int h_in,
                                    int w_in, int cin, int cout, int stride, int pad, int dil, int h_out, int w_out, int ldx,
                                    int ldy, int relu, asr_stream_t stream) {
    const long long m = (long long)batch * h_out * w_out;
    int rc = check_common("asr_conv3x3_mfma_f32", x, w_packed, y, m, 9 * cin, cout, ldx, ldy, nullptr, 0);
    if (rc != ASR_OK) return rc;
    ASR_REQUIRE(batch > 0 && h_in > 0 && w_in > 0 && h_out > 0 && w_out > 0 && stride > 0 && dil > 0 && pad >= 0,
                "asr_conv3x3_mfma_f32: bad geometry");
    ASR_UNSUPPORTED(cin % BK, "asr_conv3x3_mfma_f32: cin must be a multiple of %d (got %d)", BK, cin);
    ASR_REQUIRE(ldx >= cin, "asr_conv3x3_mfma_f32: ldx < cin");
    PwArgs a{};
    a.x = x; a.wp = w_packed; a.bias = bias; a.res = nullptr; a.y = y;
    a.M = m; a.K = 9 * cin; a.N = cout; a.Npad = round_up(cout, 128);
    a.ldx = ldx; a.ldy = ldy; a.ldres = 0; a.relu = relu;
    a.taps = 9; a.cin = cin; a.h_in = h_in; a.w_in = w_in; a.h_out = h_out; a.w_out = w_out;
    a.stride = stride; a.pad = pad; a.dil = dil;
    return dispatch<true>(a, asr_stream(stream));
}

// ---- split-f16 entry points ---------------------------------------------------------------------------
extern "C" size_t asr_pwconv_packed_floats_f16x3(int k, int n) {
    if (k <= 0 || n <= 0) return 0;
    return (size_t)round_up(k, BK) * (size_t)round_up(n, 128);   // two half planes = one float per (k, n)
}

extern "C" int asr_pwconv_pack_weights_f16x3(const float* w_kn, float* w_packed, int k, int n, asr_stream_t stream) {
    ASR_REQUIRE(w_kn && w_packed, "asr_pwconv_pack_weights_f16x3: null pointer");
    ASR_REQUIRE(k > 0 && n > 0, "asr_pwconv_pack_weights_f16x3: bad shape k=%d n=%d", k, n);
    const int Kpad = round_up(k, BK), Npad = round_up(n, 128);
    const long long total = (long long)Kpad * Npad;
    const int grid = (int)(asr_cdiv(total, 256) < 4096 ? asr_cdiv(total, 256) : 4096);
    hipLaunchKernelGGL(pack_weights_f16x3_kernel, dim3(grid), dim3(256), 0, asr_stream(stream), w_kn,
                       reinterpret_cast<_Float16*>(w_packed), k, n, Kpad, Npad);
    ASR_LAUNCH_CHECK();
    return ASR_OK;
}

// Launch of the in-kernel-split kernels (shared by the pointwise and the 3x3 implicit-GEMM entry points): 128 x 128 tile
// (4 waves of 64 x 64, one LDS stage, 3 workgroups per CU); conv = 0 pointwise, 1 implicit 3x3 GEMM, 2 the same with a
// 128 x 64 tile (cout <= 64).  Larger tiles of this kernel (256 x 256, 128 x 256) were measured within +-3 % and 10-15 %
// behind on the 728-channel layers and are not built.
template <int WM, int WN, int TM, int TN, bool CONV>
static int launch_f16x3_shape(PwArgs& a, long long& nwg, asr_stream_t stream) {
    constexpr int bm = WM * TM * 32, bn = WN * TN * 32;
    constexpr size_t lds_stage = (size_t)2 * BK * (bm + bn) * sizeof(_Float16);
    constexpr size_t lds_epi = (size_t)WM * WN * 32 * TN * 32 * sizeof(float);
    constexpr size_t lds = lds_stage > lds_epi ? lds_stage : lds_epi;
    a.tiles_n = (int)asr_cdiv(a.N, bn);
    nwg = asr_cdiv(a.M, bm) * a.tiles_n;
    ASR_REQUIRE(nwg <= 0x7fffffffLL, "asr_pwconv_mfma_f16x3: grid too large");
    auto kern = pw_gemm_f16x3_kernel<WM, WN, TM, TN, CONV>;
    static AsrDeviceOnce once;
    ASR_HIP_CHECK(asr_allow_dynamic_lds(once, reinterpret_cast<const void*>(kern), (int)lds));
    hipLaunchKernelGGL(kern, dim3((unsigned)nwg), dim3(WM * WN * 64), lds, asr_stream(stream), a);
    ASR_LAUNCH_CHECK();
    return ASR_OK;
}

static int launch_f16x3(PwArgs a, int conv, asr_stream_t stream) {
    long long nwg = 0;
    const int rc = conv == 0 ? (a.N <= 64 ? launch_f16x3_shape<2, 2, 2, 1, false>(a, nwg, stream)     // 128 x 64 tile
                                          : launch_f16x3_shape<2, 2, 2, 2, false>(a, nwg, stream))
                 : conv == 1 ? launch_f16x3_shape<2, 2, 2, 2, true>(a, nwg, stream)
                             : launch_f16x3_shape<2, 2, 2, 1, true>(a, nwg, stream);
    if (rc != ASR_OK) return rc;
#ifdef ASR_GEMM_PHASE_PROFILE
    {
        static long long host[ASR_PHASE_BLOCKS * 16];
        ASR_HIP_CHECK(hipDeviceSynchronize());
        ASR_HIP_CHECK(hipMemcpyFromSymbol(host, HIP_SYMBOL(g_phase_cycles), sizeof(host)));
        const long long nb = nwg < ASR_PHASE_BLOCKS ? nwg : ASR_PHASE_BLOCKS;
        double mean[16] = {0};
        for (long long b = 0; b < nb; ++b)
            for (int i = 0; i < 16; ++i) mean[i] += (double)host[b * 16 + i] / (double)nb;
        const int kt = (a.K + 31) / 32;
        const int ktm = kt > 1 ? kt - 1 : 1;                     // phases that run between K-steps (none when K <= 32)
        fprintf(stderr, "[phase] M=%lld K=%d N=%d blocks=%lld ksteps=%d | prologue %.0f | per k-step: issue %.0f  lds+mfma %.0f  vmwait %.0f  "
                        "barrier1 %.0f  split+store %.0f  barrier2 %.0f | epilogue %.0f | block total %.0f cycles\n",
                (long long)a.M, a.K, a.N, nwg, kt, mean[0], mean[1] / kt, mean[2] / kt, mean[3] / ktm, mean[4] / ktm, mean[5] / ktm,
                mean[6] / ktm, mean[7], mean[0] + mean[1] + mean[2] + mean[3] + mean[4] + mean[5] + mean[6] + mean[7]);
    }
#endif
    return ASR_OK;
}

extern "C" int asr_pwconv_mfma_f16x3(const float* x, const float* w_packed, const float* bias, const float* residual, float* y,
                                     int64_t m, int k, int n, int ldx, int ldy, int ldres, int relu, int sub_stride,
                                     int h_in, int w_in, asr_stream_t stream) {
    int rc = check_common("asr_pwconv_mfma_f16x3", x, w_packed, y, m, k, n, ldx, ldy, residual, ldres);
    if (rc != ASR_OK) return rc;
    ASR_REQUIRE(ldx >= k, "asr_pwconv_mfma_f16x3: ldx < k");
    ASR_UNSUPPORTED(k & 3, "asr_pwconv_mfma_f16x3: k must be a multiple of 4 (got %d)", k);
    PwArgs a{};
    a.x = x; a.wp = w_packed; a.bias = bias; a.res = residual; a.y = y;
    a.M = m; a.K = k; a.N = n; a.Npad = round_up(n, 128); a.Kpad = round_up(k, BK);
    a.ldx = ldx; a.ldy = ldy; a.ldres = ldres; a.relu = relu;
    a.taps = 1; a.cin = k; a.stride = 1; a.pad = 0; a.dil = 1;
    if (sub_stride > 1) {
        ASR_REQUIRE(h_in > 0 && w_in > 0, "asr_pwconv_mfma_f16x3: h_in/w_in required with sub_stride");
        a.h_in = h_in; a.w_in = w_in; a.stride = sub_stride;
        a.h_out = (h_in + sub_stride - 1) / sub_stride;
        a.w_out = (w_in + sub_stride - 1) / sub_stride;
        ASR_REQUIRE(m % ((long long)a.h_out * a.w_out) == 0, "asr_pwconv_mfma_f16x3: m is not a whole number of %dx%d maps",
                    a.h_out, a.w_out);
    }
    return launch_f16x3(a, 0, stream);
}

extern "C" int asr_pwconv_mfma_f16x3_presplit(const void* x_split, const float* w_packed, const float* bias,
                                              const float* residual, float* y, int64_t m, int k, int n, int ldx_chunks, int ldy,
                                              int ldres, int relu, asr_stream_t stream) {
    ASR_REQUIRE(x_split && w_packed && y, "asr_pwconv_mfma_f16x3_presplit: null pointer");
    ASR_REQUIRE(m > 0 && k > 0 && n > 0, "asr_pwconv_mfma_f16x3_presplit: bad shape m=%lld k=%d n=%d", (long long)m, k, n);
    ASR_REQUIRE(ldy >= n && (!residual || ldres >= n), "asr_pwconv_mfma_f16x3_presplit: ldy / ldres < n");
    PwArgs a{};
    a.x = reinterpret_cast<const float*>(x_split); a.wp = w_packed; a.bias = bias; a.res = residual; a.y = y;
    a.M = m; a.K = k; a.N = n; a.Npad = round_up(n, 128); a.Kpad = round_up(k, BK);
    a.ldx = ldx_chunks; a.ldy = ldy; a.ldres = ldres; a.relu = relu;
    a.taps = 1; a.cin = k; a.stride = 1; a.pad = 0; a.dil = 1;
    ASR_REQUIRE(ldx_chunks * BK >= a.Kpad, "asr_pwconv_mfma_f16x3_presplit: ldx_chunks * 32 < ceil32(k)");
    ASR_UNSUPPORTED(a.Npad % 256 != 0, "asr_pwconv_mfma_f16x3_presplit: ceil128(n) must be a multiple of 256 (n=%d)", n);
    ASR_UNSUPPORTED(reinterpret_cast<uintptr_t>(x_split) & 127, "asr_pwconv_mfma_f16x3_presplit: x_split must be 128-byte aligned");
    // the loader waves address a tile's operands by 32-bit byte offsets from a per-tile base
    ASR_UNSUPPORTED((long long)ldx_chunks * 128 * 256 >= (1LL << 32) || (long long)a.Kpad * a.Npad * 4 + (long long)a.Npad * 64 >= (1LL << 32),
                    "asr_pwconv_mfma_f16x3_presplit: operands too large for 32-bit tile offsets (k=%d n=%d ldx_chunks=%d)", k, n, ldx_chunks);
    constexpr int bm = 256, bn = 256;
    a.tiles_n = (int)asr_cdiv(n, bn);
    const long long nwg = asr_cdiv(m, bm) * a.tiles_n;
    ASR_REQUIRE(nwg <= 0x7fffffffLL, "asr_pwconv_mfma_f16x3_presplit: grid too large");
    constexpr size_t lds = 5 * 32 * 1024;                      // the five-unit ring: all of the CU's LDS
    // Without a residual (whose staging needs the ring's free slots in the epilogue) and with more tiles than CUs the
    // tiles are walked by one persistent workgroup per CU: the next tile's first units land under the current epilogue.
    const int cus = asr_device_cu_count();
    ASR_REQUIRE(cus > 0, "asr_pwconv_mfma_f16x3_presplit: no device");
    const int cu_count = cus >= 8 ? cus / 8 * 8 : cus;        // a multiple of 8: a workgroup's tiles stay on its XCD
    // The persistent walk is the product path since round 4.  Round 3 held it back because, with two lanes in flight, it moved
    // the forward pass's kernels to where the other lane's SR solves met the fused entry-flow kernels on a SIMD -- an MI355X
    // erratum of packed-f32 op_sel beside MFMA that hit the SOLVER's kernels (csrc/isa_guard.py keeps the
    // instruction form out of every kernel).  This kernel itself was never part of it: bit-identical to the one-tile kernel
    // (tests/test_gpu_layers.py), and its twelve waves leave no room for a co-resident wave.
#ifndef ASR_PERSISTENT_WALK
#define ASR_PERSISTENT_WALK 1       // 0: the one-tile kernel for every launch (A/B builds: ASR_EXTRA_HIPFLAGS=-DASR_PERSISTENT_WALK=0)
#endif
    constexpr bool kPersistentWalk = ASR_PERSISTENT_WALK != 0;
    if (kPersistentWalk && !residual && a.Kpad / BK >= 4 && nwg > cu_count) {
        auto kern = pw_gemm_f16x3_pre_ring_persist_kernel;
        static AsrDeviceOnce once;
        ASR_HIP_CHECK(asr_allow_dynamic_lds(once, reinterpret_cast<const void*>(kern), (int)lds));
        hipLaunchKernelGGL(kern, dim3((unsigned)cu_count), dim3(768), lds, asr_stream(stream), a, (int)nwg);
        ASR_LAUNCH_CHECK();
        return ASR_OK;
    }
    auto kern = pw_gemm_f16x3_pre_ring_kernel;
    static AsrDeviceOnce once;
    ASR_HIP_CHECK(asr_allow_dynamic_lds(once, reinterpret_cast<const void*>(kern), (int)lds));
    hipLaunchKernelGGL(kern, dim3((unsigned)nwg), dim3(768), lds, asr_stream(stream), a);
    ASR_LAUNCH_CHECK();
#ifdef ASR_GEMM_PHASE_PROFILE
    {
        static long long host[ASR_PHASE_BLOCKS * 16];
        ASR_HIP_CHECK(hipDeviceSynchronize());
        ASR_HIP_CHECK(hipMemcpyFromSymbol(host, HIP_SYMBOL(g_phase_cycles), sizeof(host)));
        const long long nb = nwg < ASR_PHASE_BLOCKS ? nwg : ASR_PHASE_BLOCKS;
        double mean[16] = {0};
        for (long long b = 0; b < nb; ++b)
            for (int i = 0; i < 16; ++i) mean[i] += (double)host[b * 16 + i] / (double)nb;
        const int kt = a.Kpad / 32;
        fprintf(stderr, "[phase-pre] M=%lld K=%d N=%d blocks=%lld ksteps=%d | mfma wave 0: prologue %.0f | per k-step: reads+mfma %.0f  barrier %.0f | "
                        "epilogue %.0f || loader wave 8 per k-step: issue %.0f  landing wait %.0f  barrier %.0f || K loop %.0f cycles = %.2f us -> "
                        "%.3f GHz in-kernel, per k-step %.0f\n", (long long)a.M, a.K, a.N, nwg, kt, mean[0], mean[2] / kt, mean[6] / kt, mean[7],
                mean[8] / kt, mean[9] / kt, mean[10] / kt, mean[11], mean[12] / 100.0, mean[12] > 0 ? mean[11] / (mean[12] * 10.0) : 0.0,
                mean[11] / kt);
    }
#endif
    return ASR_OK;
}

extern "C" int asr_conv3x3_mfma_f16x3(const float* x, const float* w_packed, const float* bias, float* y, int batch, int h_in,
                                      int w_in, int cin, int cout, int stride, int pad, int dil, int h_out, int w_out, int ldx,
                                      int ldy, int relu, asr_stream_t stream) {
    const long long m = (long long)batch * h_out * w_out;
    int rc = check_common("asr_conv3x3_mfma_f16x3", x, w_packed, y, m, 9 * cin, cout, ldx, ldy, nullptr, 0);
    if (rc != ASR_OK) return rc;
    ASR_REQUIRE(batch > 0 && h_in > 0 && w_in > 0 && h_out > 0 && w_out > 0 && stride > 0 && dil > 0 && pad >= 0,
                "asr_conv3x3_mfma_f16x3: bad geometry");
    ASR_UNSUPPORTED(cin % BK, "asr_conv3x3_mfma_f16x3: cin must be a multiple of %d (got %d)", BK, cin);
    ASR_REQUIRE(ldx >= cin, "asr_conv3x3_mfma_f16x3: ldx < cin");
    PwArgs a{};
    a.x = x; a.wp = w_packed; a.bias = bias; a.res = nullptr; a.y = y;
    a.M = m; a.K = 9 * cin; a.N = cout; a.Npad = round_up(cout, 128); a.Kpad = round_up(9 * cin, BK);
    a.ldx = ldx; a.ldy = ldy; a.ldres = 0; a.relu = relu;
    a.taps = 9; a.cin = cin; a.h_in = h_in; a.w_in = w_in; a.h_out = h_out; a.w_out = w_out;
    a.stride = stride; a.pad = pad; a.dil = dil;
    return launch_f16x3(a, cout <= 64 ? 2 : 1, stream);
}
