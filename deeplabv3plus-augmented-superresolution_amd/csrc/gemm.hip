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
#include "asr_common.h"
#include <utility>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int BK = 32;

struct PwArgs {
    const float* x;
    const float* wp;
    const float* bias;
    const float* res;
    float* y;
    long long M;
    int K, N, Npad, Kpad;
    int ldx, ldy, ldres;
    int relu;
    int tiles_n;
    // row mapping: output row m = (b, oy, ox) of an h_out x w_out map
    int taps;  // 1 = pointwise (optionally spatially subsampled), 9 = 3x3 implicit GEMM
    int cin;   // channels per tap
    int h_in, w_in, h_out, w_out, stride, pad, dil;
};

// ---- epilogue shared by the f32 and the split-f16 kernels ------------------------------------------
// C/D map of every 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5).
// The accumulators hold columns on lanes; writing them straight out would cost 16 dword stores per
// 32x32 tile, each touching two 128-byte row pieces.  Instead every wave transposes its
// (32 x TN*32) half-tile through its own slice of the (now idle) staging LDS and stores whole
// 16-byte pieces: 16 lanes cover one 256-byte row segment, 4x fewer store instructions.
// The caller guarantees (barrier) that no wave still reads the staging tiles.
template <int WM, int WN, int TM, int TN, bool RES_AHEAD = false>
__device__ __forceinline__ void pw_epilogue(const PwArgs& p, f32x16 (&acc)[TM][TN], float* smem, int tile_m, int tile_n,
                                            int wave, int lane) {
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
    constexpr int WCOLS = TN * 32;                 // columns of the wave's sub-tile
    constexpr int LPR = WCOLS / 4;                 // lanes per row in the read-back
    constexpr int RPI = 64 / LPR;                  // rows per wave-instruction
    const int wm = wave / WN, wn = wave % WN;
    const int l32 = lane & 31, hh = lane >> 5;
    float* const stage = smem + wave * (32 * WCOLS);
    const bool vec_ok = ((p.N & 3) == 0) && ((p.ldy & 3) == 0) && ((reinterpret_cast<uintptr_t>(p.y) & 15) == 0) &&
                        (!p.res || (((p.ldres & 3) == 0) && ((reinterpret_cast<uintptr_t>(p.res) & 15) == 0)));
    const int n_wave = tile_n * BN + wn * WCOLS;
    const int c4 = lane % LPR, r_in = lane / LPR;
    const int n = n_wave + c4 * 4;
    const bool res_vec = vec_ok && p.res != nullptr;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const long long m_base = (long long)tile_m * BM + (wm * TM + i) * 32;
        // RES_AHEAD (kernels with registers to spare: the 256 x 256 LDS-DMA kernel): the residual pieces of this
        // 32-row slab are requested before the accumulators go through the LDS (clamped addresses, no branch around the
        // loads), so they arrive during the transposition instead of one waited-for round trip per piece in the store loop.
        f32x4 rv[RES_AHEAD ? 32 / RPI : 1];
        if (RES_AHEAD && res_vec) {
            const float* rbase = p.res + (n < p.N ? n : 0);
#pragma unroll
            for (int q = 0; q < 32 / RPI; ++q) {
                const long long m = m_base + q * RPI + r_in;
                rv[q] = *reinterpret_cast<const f32x4*>(rbase + (m < p.M ? m : p.M - 1) * p.ldres);
            }
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int nj = n_wave + j * 32 + l32;
            const float bv = (p.bias && nj < p.N) ? p.bias[nj] : 0.f;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                float v = acc[i][j][e] + bv;
                if (p.relu) v = fmaxf(v, 0.f);
                if (p.relu == 2) v = fminf(v, 6.f);          // ReLU6 (MobileNetV2 blocks)
                stage[((e & 3) + 8 * (e >> 2) + 4 * hh) * WCOLS + j * 32 + l32] = v;
            }
        }
#pragma unroll
        for (int q = 0; q < 32 / RPI; ++q) {
            const int r = q * RPI + r_in;
            const long long m = m_base + r;
            f32x4 v = *reinterpret_cast<const f32x4*>(stage + r * WCOLS + c4 * 4);
            if (m < p.M && n < p.N) {
                if (vec_ok) {
                    if (p.res) {
                        if (RES_AHEAD) v += rv[q];
                        else v += *reinterpret_cast<const f32x4*>(p.res + m * p.ldres + n);
                    }
                    *reinterpret_cast<f32x4*>(p.y + m * p.ldy + n) = v;
                } else {
#pragma unroll
                    for (int t = 0; t < 4; ++t)
                        if (n + t < p.N) {
                            float o = v[t];
                            if (p.res) o += p.res[m * p.ldres + n + t];
                            p.y[m * p.ldy + n + t] = o;
                        }
                }
            }
        }
    }
}

// pw_epilogue for accumulators of v_mfma_f32_16x16x32_f16: acc[rt][ct] (f32x4) holds rows rt * 16 + 4 * (lane >> 4) + r
// (r = 0..3), column ct * 16 + (lane & 15) of the wave's (RT * 16) x (CT * 16) sub-tile.  Same LDS transposition, residual
// prefetch and 16-byte stores as pw_epilogue<..., RES_AHEAD = true>; only the staging map differs.
// RES_DEPTH: residual pieces (of the 16 per 32-row slab) requested ahead of their use.  16 = the whole slab before its
// accumulators go through the LDS (kernels with registers to spare); a smaller depth keeps a rolling queue -- the piece of
// row group q + RES_DEPTH is requested when the piece of row group q has been added -- for kernels at 168 registers per wave.
template <int WM, int WN, int RT, int CT, int RES_DEPTH = 16>
__device__ __forceinline__ void pw_epilogue16(const PwArgs& p, f32x4 (&acc)[RT][CT], float* smem, int tile_m, int tile_n, int wave, int lane) {
    constexpr int BM = WM * RT * 16, BN = WN * CT * 16, WCOLS = CT * 16;
    constexpr int LPR = WCOLS / 4, RPI = 64 / LPR;
    const int wm = wave / WN, wn = wave % WN;
    const int l16 = lane & 15, q4 = lane >> 4;
    float* const stage = smem + wave * (32 * WCOLS);
    const bool vec_ok = ((p.N & 3) == 0) && ((p.ldy & 3) == 0) && ((reinterpret_cast<uintptr_t>(p.y) & 15) == 0) &&
                        (!p.res || (((p.ldres & 3) == 0) && ((reinterpret_cast<uintptr_t>(p.res) & 15) == 0)));
    const int n_wave = tile_n * BN + wn * WCOLS;
    const int c4 = lane % LPR, r_in = lane / LPR;
    const int n = n_wave + c4 * 4;
    const bool res_vec = vec_ok && p.res != nullptr;
    float bv[CT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
        const int nj = n_wave + ct * 16 + l16;
        bv[ct] = (p.bias && nj < p.N) ? p.bias[nj] : 0.f;
    }
#pragma unroll
    for (int i = 0; i < RT / 2; ++i) {                         // 32-row slabs
        const long long m_base = (long long)tile_m * BM + (wm * RT + 2 * i) * 16;
        constexpr int NQ = 32 / RPI;
        static_assert(RES_DEPTH >= 1 && RES_DEPTH <= NQ, "RES_DEPTH out of range");
        f32x4 rv[RES_DEPTH];
        const float* const rbase = res_vec ? p.res + (n < p.N ? n : 0) : nullptr;
        auto request = [&](int q) {                            // clamped address, no branch around the load
            const long long m = m_base + q * RPI + r_in;
            rv[q % RES_DEPTH] = *reinterpret_cast<const f32x4*>(rbase + (m < p.M ? m : p.M - 1) * p.ldres);
        };
        constexpr bool kRequestFirst = RES_DEPTH == NQ;        // a short queue is requested behind the staging stores, when
        if (res_vec && kRequestFirst) {                        // this slab's accumulators no longer occupy registers
#pragma unroll
            for (int q = 0; q < RES_DEPTH; ++q) request(q);
        }
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2)
#pragma unroll
            for (int ct = 0; ct < CT; ++ct)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float v = acc[2 * i + h2][ct][r] + bv[ct];
                    if (p.relu) v = fmaxf(v, 0.f);
                    if (p.relu == 2) v = fminf(v, 6.f);
                    stage[(h2 * 16 + 4 * q4 + r) * WCOLS + ct * 16 + l16] = v;
                }
        if (res_vec && !kRequestFirst) {
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int q = 0; q < RES_DEPTH; ++q) request(q);
        }
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int r = q * RPI + r_in;
            const long long m = m_base + r;
            f32x4 v = *reinterpret_cast<const f32x4*>(stage + r * WCOLS + c4 * 4);
            f32x4 rq = {0.f, 0.f, 0.f, 0.f};
            if (res_vec) {
                rq = rv[q % RES_DEPTH];
                if (q + RES_DEPTH < NQ) request(q + RES_DEPTH);
            }
            if (m < p.M && n < p.N) {
                if (vec_ok) {
                    if (p.res) v += rq;
                    *reinterpret_cast<f32x4*>(p.y + m * p.ldy + n) = v;
                } else {
#pragma unroll
                    for (int t = 0; t < 4; ++t)
                        if (n + t < p.N) {
                            float o = v[t];
                            if (p.res) o += p.res[m * p.ldres + n + t];
                            p.y[m * p.ldy + n + t] = o;
                        }
                }
            }
        }
    }
}

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
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ void split_f16x8(const f32x4& a, const f32x4& b, f16x8& hi, f16x8& lo) {
    const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        _Float16 h, l;
        asr_split_f16(v[j], h, l);
        hi[j] = h;
        lo[j] = l;
    }
}

// Phase timing of the K loop (tools/gemm_phase_profile.sh builds with -DASR_GEMM_PHASE_PROFILE; never in the product build):
// wave 0 of every block accumulates shader-clock deltas per phase and the launcher prints their means.
#ifdef ASR_GEMM_PHASE_PROFILE
#define ASR_PHASE_BLOCKS 8192
__device__ long long g_phase_cycles[ASR_PHASE_BLOCKS * 16];    // per block: 0-7 wave 0's phases, 8-10 loader wave, 11/12 K-loop cycles / 100 MHz ticks
#define PHASE_MARK(i)                                              \
    do {                                                           \
        const long long now_ = (long long)__builtin_readcyclecounter(); \
        ph[i] += now_ - tprev;                                     \
        tprev = now_;                                              \
    } while (0)
#define PHASE_WAIT_VM() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#define PHASE_WAIT_LGKM() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
#else
#define PHASE_MARK(i)
#define PHASE_WAIT_VM()
#define PHASE_WAIT_LGKM()
#endif

template <int WM, int WN, int TM, int TN, bool CONV>
__global__ __launch_bounds__(WM * WN * 64) void pw_gemm_f16x3_kernel(PwArgs p) {
    constexpr int NT = WM * WN * 64, BM = WM * TM * 32, BN = WN * TN * 32;
    constexpr int PA = BM * 4 / NT;                    // (row, 32-byte k-slot) items of the A tile per thread
    constexpr int PB = BN * 4 / NT;                    // (k-octet, column) items of the B tile per thread and plane
    static_assert((BM * 4) % NT == 0 && (BN * 4) % NT == 0 && (BN & (BN - 1)) == 0, "tile / thread-count mismatch");
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
// 128 x 128: the CU takes in only ~20-30 B/clk from L2 under load, DESIGN.md "GEMM phase profile") run with two LDS
// stages.  8 MFMA waves, each 64 x 128 of the tile; BK = 32.
//   LDS stage (64 KB): A [256 rows][8 slots of 16 B: hi oct 0-3, lo oct 0-3], slot XOR-swizzled by (row >> 1) & 7 --
//   applied on the per-lane SOURCE address, the DMA destination is lane-linear --, then B_hi, B_lo [4 oct][256 col][8].
// =================================================================================================
typedef __attribute__((address_space(3))) void* asr_lds_ptr;
typedef const __attribute__((address_space(1))) void* asr_gbl_ptr;

__device__ __forceinline__ void glds16(const void* g, void* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((asr_gbl_ptr)g, (asr_lds_ptr)lds_wave_base, 16, 0, 0);
}

// The workgroup's twelve waves have FIXED ROLES.  Round 1's form of this kernel (8 waves that each requested their own 8
// pieces of the next stage and then issued their MFMAs; now diagnostic-only, below) left the matrix pipe idle for the ~1000
// cycles per K-step in which all waves sat in the vector-memory issue queue: a wave pays ~100-150 cycles per 1 KiB LDS-DMA
// piece whoever issues it, and a wave that is issuing cannot issue MFMAs.  Here waves 0-7 only read fragments and issue
// MFMAs (no vector-memory instruction inside the K loop); waves 8-11 -- one per SIMD -- only request: 16 pieces each per
// K-step (~1700 cycles of issue + ~1300 until the last piece has landed, under the ~3100 cycles the SIMD's two MFMA waves
// need), then meet the MFMA waves at the K-step's barrier.  Three waves per SIMD leave 168 registers per wave: the MFMA
// waves hold the 128 accumulators and walk their 64 x 128 tile in two 32-row halves (the A fragments of two row tiles at a
// time, the B fragments double-buffered one column tile ahead; scheduling barriers keep the compiler from hoisting every
// read, which would spill).  Same MFMA sequence per accumulator as the 8-wave form => bit-identical results; K-step 4100 ->
// 3600 cycles, 4-6 % less wall time launch for launch (tools/ab_presplit_lw.py: the two forms interleaved in one process;
// timed in separate processes the chip's clock drift hides the difference), profiles/r02_gemm_loader_wave_experiment.txt.
// =================================================================================================
template <typename F, int... I>
__device__ __forceinline__ void asr_static_for_impl(F& f, std::integer_sequence<int, I...>) {
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, typename F>
__device__ __forceinline__ void asr_static_for(F& f) {
    asr_static_for_impl(f, std::make_integer_sequence<int, N>{});
}

// MOCK_FUSED (diagnostic library only, tools/ab_presplit_lw.py "experiment"): the four loader waves behave like the producer
// waves of a fused depthwise -> pointwise layer WOULD -- per K-step each requests 8 B pieces by LDS-DMA, loads its share of
// the 40 KB of f32 input rows (8 image rows + 2 halo rows of a 32-wide map, 32 channels) into registers two K-steps ahead,
// issues 416 VALU instructions on them and writes its 8 KB of the A stage with ds_write_b128.  The A stage then holds
// garbage: TIMING ONLY, an upper bound on what the fusion could reach.
template <int PIECES_PER_LOADER = 16, bool MOCK_FUSED = false>
__global__ __launch_bounds__(768, 3) void pw_gemm_f16x3_pre_lw_kernel(PwArgs p) {
    constexpr int BM = 256, BN = 256, WN = 2, RT = 4, CT = 8;
    constexpr int A_BYTES = BM * 128, B_BYTES = 4 * BN * 16, STAGE_BYTES = A_BYTES + 2 * B_BYTES;
    constexpr int A_PIECES = A_BYTES / 1024;                   // 32 pieces of 64 lanes x 16 B; then 32 B pieces (hi plane, lo plane)
    static_assert(4 * PIECES_PER_LOADER * 1024 == STAGE_BYTES, "four loader waves cover one stage");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    char* const lds = reinterpret_cast<char*>(smem);

    const int nwg = gridDim.x, orig = blockIdx.x;
    const int q8 = nwg >> 3, r8 = nwg & 7, xcd = orig & 7;
    const int lid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (orig >> 3);
    const int tile_m = lid / p.tiles_n, tile_n = lid % p.tiles_n;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int KT = p.Kpad / BK;

#ifdef ASR_DIAG_KERNELS
    if (MOCK_FUSED && wave >= 8) {
        const int w = wave - 8;
        const long long plane_bytes = (long long)p.Kpad * p.Npad * 2;
        const char* bsrc[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int qb = (w * 8 + j) * 64 + lane, plane = qb >> 10, r = qb & 1023, oct = r >> 8, col = r & 255;
            bsrc[j] = reinterpret_cast<const char*>(p.wp) + plane * plane_bytes + (((long long)oct * p.Npad) + (long long)tile_n * BN + col) * 16;
        }
        const char* rsrc[10];
#pragma unroll
        for (int r = 0; r < 10; ++r) {
            long long px = (long long)tile_m * BM - 32 + w * 80 + r * 8 + (lane >> 3);
            px = px < 0 ? 0 : (px >= p.M ? p.M - 1 : px);
            rsrc[r] = reinterpret_cast<const char*>(p.x) + (px * p.ldx) * 128 + (lane & 7) * 16;
        }
        auto issue_b = [&](int kt, int stage) {
            char* const st = lds + stage * STAGE_BYTES + A_BYTES + w * 8 * 1024;
#pragma unroll
            for (int j = 0; j < 8; ++j) glds16(bsrc[j] + (long long)kt * 4 * p.Npad * 16, st + j * 1024);
        };
        f32x4 rows[2][10];
        auto load_rows = [&](int kt, int buf) {
#pragma unroll
            for (int r = 0; r < 10; ++r) rows[buf][r] = *reinterpret_cast<const f32x4*>(rsrc[r] + (long long)(kt < KT ? kt : KT - 1) * 128);
        };
        auto produce = [&](int stage, int buf) {                // 416 VALU instructions on the rows, then 8 x ds_write_b128
#pragma unroll
            for (int it = 0; it < 13; ++it)
#pragma unroll
                for (int r = 0; r < 8; ++r)
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(rows[buf][r][e]) : "v"(rows[buf][(r + 1) % 10][e]), "v"(rows[buf][9][(e + 1) & 3]));
            char* const st = lds + stage * STAGE_BYTES + w * 8 * 1024 + lane * 16;
#pragma unroll
            for (int j = 0; j < 8; ++j) *reinterpret_cast<f32x4*>(st + j * 1024) = rows[buf][j];
        };
        issue_b(0, 0);
        load_rows(0, 0);
        load_rows(1, 1);
        asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
        produce(0, 0);
        asm volatile("s_waitcnt vmcnt(10)\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        for (int kt = 0; kt < KT; kt += 2) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {                       // K-step kt + h: produce the stage of kt + h + 1 from rows[(h + 1) & 1]
                const int k = kt + h;
                if (k < KT) {
                    if (k + 1 < KT) issue_b(k + 1, (k + 1) & 1);
                    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");       // the rows of k + 1 (requested a K-step ago) are in
                    if (k + 1 < KT) produce((k + 1) & 1, (h + 1) & 1);
                    load_rows(k + 2, h & 1);                                 // two K-steps ahead, into the buffer just consumed
                    asm volatile("s_waitcnt vmcnt(10)\n\ts_waitcnt lgkmcnt(0)" ::: "memory");   // B pieces landed, A stores done
                    __builtin_amdgcn_s_barrier();
                }
            }
        }
        return;
    }
#endif
    if (wave >= 8) {
        // ---- loader wave: pieces (wave - 8) * 16 .. + 15 of every stage ------------------------------------------
        const int first = (wave - 8) * PIECES_PER_LOADER;
        const char* src[PIECES_PER_LOADER];
        long long kstep[2];                                     // byte advance per K-step: A pieces, B pieces
        kstep[0] = 128;
        kstep[1] = (long long)4 * p.Npad * 16;
        const long long plane_bytes = (long long)p.Kpad * p.Npad * 2;
#pragma unroll
        for (int j = 0; j < PIECES_PER_LOADER; ++j) {
            const int pi = first + j;
            if (pi < A_PIECES) {                                // A: (row q >> 3, LDS slot q & 7 holding global slot ^ swizzle)
                const int q = pi * 64 + lane, row = q >> 3, slot = (q & 7) ^ ((row >> 1) & 7);
                long long m = (long long)tile_m * BM + row;
                if (m >= p.M) m = p.M - 1;                      // rows past the end re-read the last row; never stored
                src[j] = reinterpret_cast<const char*>(p.x) + (m * p.ldx) * 128 + slot * 16;
            } else {                                            // B: plane (hi, lo), k-octet, column
                const int qb = (pi - A_PIECES) * 64 + lane, plane = qb >> 10, r = qb & 1023, oct = r >> 8, col = r & 255;
                src[j] = reinterpret_cast<const char*>(p.wp) + plane * plane_bytes +
                         (((long long)oct * p.Npad) + (long long)tile_n * BN + col) * 16;
            }
        }
        auto issue = [&](int kt, int stage) {
            char* const st = lds + stage * STAGE_BYTES + first * 1024;
#pragma unroll
            for (int j = 0; j < PIECES_PER_LOADER; ++j)
                glds16(src[j] + kt * kstep[(first + j) < A_PIECES ? 0 : 1], st + j * 1024);
        };
        issue(0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
#ifdef ASR_GEMM_PHASE_PROFILE
        long long ph[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        long long tprev = (long long)__builtin_readcyclecounter();
#endif
        for (int kt = 0; kt < KT; ++kt) {
            if (kt + 1 < KT) issue(kt + 1, (kt + 1) & 1);      // the other stage: last read before the previous barrier
            PHASE_MARK(0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // landed before anyone is released to read it
            PHASE_MARK(1);
            __builtin_amdgcn_s_barrier();
            PHASE_MARK(2);
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
    const int wave_e = wm * WN + wn;                           // the epilogue's (row block, column half) numbering
    const int ct_valid = min(CT, max(0, (p.N - (tile_n * BN + wn * (CT * 16)) + 15) >> 4));   // column tiles holding real columns
    const int l16 = lane & 15, oct = lane >> 4;                // A: row = l16, k = 8 oct ..; B: column = l16, same k
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
    __builtin_amdgcn_s_barrier();                              // stage 0 landed (the loaders waited for it)
    PHASE_MARK(0);
#ifdef ASR_GEMM_PHASE_PROFILE
    const long long loop_c0 = (long long)__builtin_readcyclecounter(), loop_r0 = (long long)__builtin_amdgcn_s_memrealtime();
#endif
    auto kloop = [&](auto CTV_) {
        constexpr int CTV = decltype(CTV_)::value;             // column tiles computed by this wave (even; 8 = all)
        for (int kt = 0; kt < KT; ++kt) {
            const char* const st = lds + (kt & 1) * STAGE_BYTES;
            // 2 halves x CTV column tiles = 2 CTV groups of 6 MFMAs (16 groups for a full tile); the fragments of group g + 1 are requested before the MFMAs of
            // group g (one B double buffer, the A pair of the second half is requested under the last group of the first),
            // and a scheduling barrier per group keeps the compiler from hoisting every read to the top (it would need 160
            // fragment registers and spill).
            f16x8 ah[2], al[2], bh[2], bl[2];
            auto read_a = [&](int half, int i) {                   // hi / lo fragments of row tile 2 * half + i
                const int row = (wm * RT + 2 * half + i) * 16 + l16, swz = (row >> 1) & 7;
                ah[i] = *reinterpret_cast<const f16x8*>(st + row * 128 + ((oct ^ swz) << 4));
                al[i] = *reinterpret_cast<const f16x8*>(st + row * 128 + (((4 + oct) ^ swz) << 4));
            };
            auto read_b = [&](int j, int buf) {
                const int col = (wn * CT + j) * 16 + l16;
                bh[buf] = *reinterpret_cast<const f16x8*>(st + A_BYTES + (oct * BN + col) * 16);
                bl[buf] = *reinterpret_cast<const f16x8*>(st + A_BYTES + B_BYTES + (oct * BN + col) * 16);
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
                    a4 = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[i], bh[g & 1], a4, 0, 0, 0);
                    a4 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[i], bl[g & 1], a4, 0, 0, 0);
                    a4 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[i], bh[g & 1], a4, 0, 0, 0);
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
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");    // this wave's reads of the stage are done
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
    pw_epilogue16<4, WN, RT, CT, 4>(p, acc, smem, tile_m, tile_n, wave_e, lane);
#ifdef ASR_GEMM_PHASE_PROFILE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    PHASE_MARK(7);
    if (tid == 0 && orig < ASR_PHASE_BLOCKS)
        for (int i = 0; i < 8; ++i) g_phase_cycles[orig * 16 + i] = ph[i];
#endif
}

#ifdef ASR_DIAG_KERNELS
// ---- diagnostic build only (csrc/build.py, ASR_BUILD_VARIANT=diag), never part of libasr_hip.so: round 1's form of this
//      kernel -- 8 waves that each request their own 8 pieces and then issue their MFMAs -- kept so that
//      tools/ab_presplit_lw.py can A/B the two in one process -------------------------------------------------------------
template <int WM, int WN, int TM, int TN>
__global__ __launch_bounds__(WM * WN * 64, 2) void pw_gemm_f16x3_pre_kernel(PwArgs p) {
    constexpr int NT = WM * WN * 64, BM = WM * TM * 32, BN = WN * TN * 32;
    constexpr int A_BYTES = BM * 128, B_BYTES = 4 * BN * 16, STAGE_BYTES = A_BYTES + 2 * B_BYTES;
    constexpr int PA = BM * 8 / NT, PB = 4 * BN / NT;          // 16-byte DMA pieces per thread: A, B (per plane)
    static_assert((BM * 8) % NT == 0 && (4 * BN) % NT == 0, "tile / thread-count mismatch");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    char* const lds = reinterpret_cast<char*>(smem);

    const int nwg = gridDim.x, orig = blockIdx.x;
    const int q8 = nwg >> 3, r8 = nwg & 7, xcd = orig & 7;
    const int lid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (orig >> 3);
    const int tile_m = lid / p.tiles_n, tile_n = lid % p.tiles_n;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;

    // per-thread DMA sources: A piece q = tid + NT * i -> (row q >> 3, LDS slot q & 7, holding global slot ^ swizzle)
    const char* a_src[PA];
#pragma unroll
    for (int i = 0; i < PA; ++i) {
        const int q = tid + NT * i, row = q >> 3, slot = (q & 7) ^ ((row >> 1) & 7);
        long long m = (long long)tile_m * BM + row;
        if (m >= p.M) m = p.M - 1;                             // rows past the end re-read the last row; never stored
        a_src[i] = reinterpret_cast<const char*>(p.x) + (m * p.ldx) * 128 + slot * 16;   // + kt * 128 per K chunk
    }
    const long long plane_bytes = (long long)p.Kpad * p.Npad * 2;
    const char* b_src[PB];
#pragma unroll
    for (int i = 0; i < PB; ++i) {
        const int q = tid + NT * i, oct = q / BN, col = q % BN;
        b_src[i] = reinterpret_cast<const char*>(p.wp) + (((long long)oct * p.Npad) + (long long)tile_n * BN + col) * 16;  // + kt * 4 * Npad * 16
    }
    const long long b_kstep = (long long)4 * p.Npad * 16;

    auto issue_tile = [&](int kt, int stage) {
        char* const st = lds + stage * STAGE_BYTES;
#pragma unroll
        for (int i = 0; i < PA; ++i) glds16(a_src[i] + (long long)kt * 128, st + (wave * 64 + NT * i) * 16);
#pragma unroll
        for (int i = 0; i < PB; ++i) {
            glds16(b_src[i] + kt * b_kstep, st + A_BYTES + (wave * 64 + NT * i) * 16);
            glds16(b_src[i] + kt * b_kstep + plane_bytes, st + A_BYTES + B_BYTES + (wave * 64 + NT * i) * 16);
        }
    };

    // v_mfma_f32_16x16x32_f16: one MFMA spans the whole 32-deep K-step.  Same flops per cycle on paper as the 32x32x16
    // shape, but under the chip's power management it sustains ~1.18x the rate (profiles/r01_gemm_phase_profile.txt:
    // 1.55-1.82 vs 1.86-2.20 PFLOP/s in bare loops) and has four independent accumulator chains per column tile.
    constexpr int RT = 2 * TM, CT = 2 * TN;                    // 16 x 16 tiles of the wave's (RT * 16) x (CT * 16) sub-tile
    f32x4 acc[RT][CT];
#pragma unroll
    for (int i = 0; i < RT; ++i)
#pragma unroll
        for (int j = 0; j < CT; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[i][j][e] = 0.f;

    const int KT = p.Kpad / BK;
#ifdef ASR_GEMM_PHASE_PROFILE
    long long ph[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    long long tprev = (long long)__builtin_readcyclecounter();
#endif
    issue_tile(0, 0);
    __syncthreads();                                           // drains the DMA (vmcnt(0)) and publishes stage 0
    PHASE_MARK(0);
#ifdef ASR_GEMM_PHASE_PROFILE
    const long long loop_c0 = (long long)__builtin_readcyclecounter(), loop_r0 = (long long)__builtin_amdgcn_s_memrealtime();
#endif
    const int l16 = lane & 15, oct = lane >> 4;                // A: row = l16, k = 8 oct ..; B: column = l16, same k
    for (int kt = 0; kt < KT; ++kt) {
        const char* const st = lds + (kt & 1) * STAGE_BYTES;
        // All 8 DMA pieces of the next tile go out before the MFMAs (spreading them between the MFMA groups, letting half of
        // the waves request, or giving the requests to four dedicated loader waves all measured equal in wall clock:
        // DESIGN.md 4.1, profiles/r02_gemm_loader_wave_experiment.txt).
        if (kt + 1 < KT) issue_tile(kt + 1, (kt + 1) & 1);     // the other stage: last read before the previous barrier
        PHASE_MARK(1);
        f16x8 ah[RT], al[RT];
#pragma unroll
        for (int i = 0; i < RT; ++i) {
            const int row = (wm * RT + i) * 16 + l16, swz = (row >> 1) & 7;
            ah[i] = *reinterpret_cast<const f16x8*>(st + row * 128 + ((oct ^ swz) << 4));
            al[i] = *reinterpret_cast<const f16x8*>(st + row * 128 + (((4 + oct) ^ swz) << 4));
        }
#pragma unroll
        for (int j = 0; j < CT; ++j) {
            const int col = (wn * CT + j) * 16 + l16;
            const f16x8 bh = *reinterpret_cast<const f16x8*>(st + A_BYTES + (oct * BN + col) * 16);
            const f16x8 bl = *reinterpret_cast<const f16x8*>(st + A_BYTES + B_BYTES + (oct * BN + col) * 16);
#pragma unroll
            for (int i = 0; i < RT; ++i) {
                f32x4& a4 = acc[i][j];
                a4 = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[i], bh, a4, 0, 0, 0);
                a4 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[i], bl, a4, 0, 0, 0);
                a4 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[i], bh, a4, 0, 0, 0);
            }
        }
        PHASE_MARK(2);
        PHASE_WAIT_VM();
        PHASE_MARK(3);
        __syncthreads();                                       // next stage landed (vmcnt(0)) and everyone is done reading this one
        PHASE_MARK(6);
    }
#ifdef ASR_GEMM_PHASE_PROFILE
    if (tid == 0 && orig < ASR_PHASE_BLOCKS) {
        g_phase_cycles[orig * 16 + 11] = (long long)__builtin_readcyclecounter() - loop_c0;
        g_phase_cycles[orig * 16 + 12] = (long long)__builtin_amdgcn_s_memrealtime() - loop_r0;
    }
#endif
    pw_epilogue16<WM, WN, RT, CT>(p, acc, smem, tile_m, tile_n, wave, lane);
#ifdef ASR_GEMM_PHASE_PROFILE
    PHASE_WAIT_VM();
    PHASE_MARK(7);
    if (tid == 0 && orig < ASR_PHASE_BLOCKS)
        for (int i = 0; i < 8; ++i) g_phase_cycles[orig * 16 + i] = ph[i];
#endif
}




#endif  // ASR_DIAG_KERNELS

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

inline int round_up(int v, int m) { return (v + m - 1) / m * m; }

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
    const int rc = conv == 0 ? launch_f16x3_shape<2, 2, 2, 2, false>(a, nwg, stream)
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
    constexpr int bm = 256, bn = 256;
    a.tiles_n = (int)asr_cdiv(n, bn);
    const long long nwg = asr_cdiv(m, bm) * a.tiles_n;
    ASR_REQUIRE(nwg <= 0x7fffffffLL, "asr_pwconv_mfma_f16x3_presplit: grid too large");
    constexpr size_t lds = 2 * (bm * 128 + 2 * 4 * bn * 16);
    auto kern = pw_gemm_f16x3_pre_lw_kernel<16>;
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

#ifdef ASR_DIAG_KERNELS
// timing-only mock of a fused depthwise -> pointwise layer (pw_gemm_f16x3_pre_lw_kernel<16, true>); results are garbage
extern "C" int asr_diag_pwconv_presplit_exp(const void* x_split, const float* w_packed, const float* bias, const float* residual,
                                            float* y, int64_t m, int k, int n, int ldx_chunks, int ldy, int ldres, int relu,
                                            asr_stream_t stream) {
    ASR_REQUIRE(x_split && w_packed && y && m > 0 && k > 0 && n > 0, "asr_diag_pwconv_presplit_exp: bad arguments");
    PwArgs a{};
    a.x = reinterpret_cast<const float*>(x_split); a.wp = w_packed; a.bias = bias; a.res = residual; a.y = y;
    a.M = m; a.K = k; a.N = n; a.Npad = round_up(n, 128); a.Kpad = round_up(k, BK);
    a.ldx = ldx_chunks; a.ldy = ldy; a.ldres = ldres; a.relu = relu;
    a.taps = 1; a.cin = k; a.stride = 1; a.pad = 0; a.dil = 1;
    ASR_UNSUPPORTED(a.Npad % 256 != 0, "asr_diag_pwconv_presplit_exp: ceil128(n) must be a multiple of 256");
    a.tiles_n = (int)asr_cdiv(n, 256);
    const long long nwg = asr_cdiv(m, 256) * a.tiles_n;
    constexpr size_t lds = 2 * (256 * 128 + 2 * 4 * 256 * 16);
    auto kern = pw_gemm_f16x3_pre_lw_kernel<16, true>;
    static AsrDeviceOnce once;
    ASR_HIP_CHECK(asr_allow_dynamic_lds(once, reinterpret_cast<const void*>(kern), (int)lds));
    hipLaunchKernelGGL(kern, dim3((unsigned)nwg), dim3(768), lds, asr_stream(stream), a);
    ASR_LAUNCH_CHECK();
    return ASR_OK;
}

// asr_pwconv_mfma_f16x3_presplit's arguments on round 1's 8-wave kernel (diagnostic library only; not in include/asr_hip.h)
extern "C" int asr_diag_pwconv_presplit_8w(const void* x_split, const float* w_packed, const float* bias, const float* residual,
                                           float* y, int64_t m, int k, int n, int ldx_chunks, int ldy, int ldres, int relu,
                                           asr_stream_t stream) {
    ASR_REQUIRE(x_split && w_packed && y && m > 0 && k > 0 && n > 0, "asr_diag_pwconv_presplit_8w: bad arguments");
    PwArgs a{};
    a.x = reinterpret_cast<const float*>(x_split); a.wp = w_packed; a.bias = bias; a.res = residual; a.y = y;
    a.M = m; a.K = k; a.N = n; a.Npad = round_up(n, 128); a.Kpad = round_up(k, BK);
    a.ldx = ldx_chunks; a.ldy = ldy; a.ldres = ldres; a.relu = relu;
    a.taps = 1; a.cin = k; a.stride = 1; a.pad = 0; a.dil = 1;
    ASR_UNSUPPORTED(a.Npad % 256 != 0, "asr_diag_pwconv_presplit_8w: ceil128(n) must be a multiple of 256");
    a.tiles_n = (int)asr_cdiv(n, 256);
    const long long nwg = asr_cdiv(m, 256) * a.tiles_n;
    constexpr size_t lds = 2 * (256 * 128 + 2 * 4 * 256 * 16);
    auto kern8 = pw_gemm_f16x3_pre_kernel<4, 2, 2, 4>;
    static AsrDeviceOnce once;
    ASR_HIP_CHECK(asr_allow_dynamic_lds(once, reinterpret_cast<const void*>(kern8), (int)lds));
    hipLaunchKernelGGL(kern8, dim3((unsigned)nwg), dim3(512), lds, asr_stream(stream), a);
    ASR_LAUNCH_CHECK();
    return ASR_OK;
}
#endif
