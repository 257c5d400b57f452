// Shared pieces of the pointwise GEMM kernels (csrc/gemm.hip: the product kernels; csrc/diag/gemm_diag.hip: earlier forms of
// the pre-split kernel kept for in-process A/Bs, built into the diagnostic library only): launch arguments, the two epilogues,
// the LDS-DMA helper and the phase-profile macros.  Everything lives in an anonymous namespace of the including file.
#pragma once
#include "asr_common.h"
#include <utility>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

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


// ---- epilogue of the ring kernel (csrc/gemm.hip: pw_gemm_f16x3_pre_ring_kernel) ----------------------------------------
// Conditions under which a tile takes the 16-byte path (kernel-uniform: the loader waves stage the epilogue's operands in
// LDS exactly when this holds): 16-byte aligned rows of y / residual / bias, N a multiple of 4.
__device__ __forceinline__ bool pw_ep_fast(const PwArgs& p) {
    return ((p.N & 3) == 0) && ((p.ldy & 3) == 0) && ((reinterpret_cast<uintptr_t>(p.y) & 15) == 0) &&
           ((reinterpret_cast<uintptr_t>(p.bias) & 15) == 0) &&
           (!p.res || (((p.ldres & 3) == 0) && ((reinterpret_cast<uintptr_t>(p.res) & 15) == 0)));
}

// acc[rt][ct] (f32x4) of lane (l16 = lane & 15, q4 = lane >> 4) holds row rt * 16 + l16 and columns ct * 16 + 4 q4 + (0..3)
// of the wave's (RT * 16) x (CT * 16) sub-tile (MFMAs issued with the weights fragment first).  m_wave / n_wave: first row /
// column of the sub-tile; wm / wn: the wave's row block and column half.  LDS (filled by the loader waves): slot f + 2 holds
// the tile's 256 bias values; the residual row tile rt sits in buffer rt & 1 (buffer 0 = slots f, f + 1; buffer 1 = slots
// f + 3, f + 4; all modulo `ring`), row wm * 16 + l16 of 64, 16-byte slots XOR-ed with l16.  With a staged residual the
// workgroup meets at one barrier per row tile (3 in all) -- every wave calls this function and takes them, whatever path
// its own rows take.
template <int RT, int CT>
__device__ __forceinline__ void pw_epilogue16_ring(const PwArgs& p, f32x4 (&acc)[RT][CT], const char* lds, int f, int unit, int ring,
                                                   long long m_wave, int n_wave, int wm, int wn) {
    int lane;
    {   // the lane index is re-derived (mbcnt of an opaque zero): no per-lane register of the epilogue is live across the K loop
        int z = 0;
        asm volatile("" : "+v"(z));
        lane = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, (unsigned)z));
    }
    const int l16 = lane & 15, q4 = lane >> 4;
    const bool fast = pw_ep_fast(p);                            // kernel-uniform
    const bool res_dma = fast && p.res != nullptr;
    const bool rows_ok = m_wave + RT * 16 <= p.M;               // wave-uniform: every row of the sub-tile exists
    auto slot_of = [&](int k) { const int v = f + k; return v >= ring ? v - ring : v; };
    const int n0 = n_wave + 4 * q4;
    if (fast && rows_ok) {
        const char* const bias_lds = lds + slot_of(2) * unit + (wn * (CT * 16) + 4 * q4) * 4;
        const unsigned yoff = ((unsigned)l16 * (unsigned)p.ldy + (unsigned)n0) * 4u;
        const int relu = p.relu;
        const bool has_bias = p.bias != nullptr;
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            char* const yrow = reinterpret_cast<char*>(p.y + (m_wave + rt * 16) * p.ldy);
            // residual quad of (row l16, column tile ct): staging row r = wm * 16 + l16, global slot s = (n_wave - tile col 0) / 4 + ct * 4 + q4
            const int r = wm * 16 + l16;
            const char* const res_lds = lds + slot_of((rt & 1) * 3 + (r >> 5)) * unit + (r & 31) * 1024;
            const int s0 = wn * (CT * 4) + q4;
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {
                if (n0 + ct * 16 < p.N) {                       // N % 4 == 0: a quad is inside or outside as a whole
                    f32x4 v = acc[rt][ct];
                    if (has_bias) v += *reinterpret_cast<const f32x4*>(bias_lds + ct * 64);
                    if (relu) {
#pragma unroll
                        for (int t = 0; t < 4; ++t) v[t] = fmaxf(v[t], 0.f);
                    }
                    if (relu == 2) {
#pragma unroll
                        for (int t = 0; t < 4; ++t) v[t] = fminf(v[t], 6.f);
                    }
                    if (res_dma) v += *reinterpret_cast<const f32x4*>(res_lds + (((s0 + ct * 4) ^ l16) << 4));
                    *reinterpret_cast<f32x4*>(yrow + (yoff + ct * 64)) = v;
                }
            }
            if (res_dma && rt + 1 < RT) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // done reading this row tile's buffer
                __builtin_amdgcn_s_barrier();                          // the next row tile has landed (the loaders waited for it)
            }
        }
        return;
    }
    if (res_dma) {                                              // the workgroup's barriers, also for a wave on the slow path
#pragma unroll 1
        for (int rt = 0; rt + 1 < RT; ++rt) __builtin_amdgcn_s_barrier();
    }
    // ragged last row tile or unaligned operands: element by element, operands straight from global memory
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
        const int n = n0 + ct * 16;
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            const long long m = m_wave + rt * 16 + l16;
            if (m >= p.M) continue;
#pragma unroll
            for (int t = 0; t < 4; ++t)
                if (n + t < p.N) {
                    float o = acc[rt][ct][t] + (p.bias ? p.bias[n + t] : 0.f);
                    if (p.relu) o = fmaxf(o, 0.f);
                    if (p.relu == 2) o = fminf(o, 6.f);
                    if (p.res) o += p.res[m * p.ldres + n + t];
                    p.y[m * p.ldy + n + t] = o;
                }
        }
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

typedef __attribute__((address_space(3))) void* asr_lds_ptr;
typedef const __attribute__((address_space(1))) void* asr_gbl_ptr;

__device__ __forceinline__ void glds16(const void* g, void* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((asr_gbl_ptr)g, (asr_lds_ptr)lds_wave_base, 16, 0, 0);
}

// The same request with the source as a wave-uniform base (scalar register pair) + a 32-bit byte offset per lane: the
// compiler's builtin always takes a 64-bit vector address (one v_lshl_add_u64 per request).  M0 = LDS byte address of the
// wave's destination; lane l lands at M0 + 16 l.
__device__ __forceinline__ void glds16_sbase(const void* uniform_base, unsigned lane_byte_offset, void* lds_wave_base) {
    const unsigned m0v = (unsigned)(unsigned long long)(asr_lds_ptr)lds_wave_base;
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"
                 :
                 : "v"(lane_byte_offset), "s"(uniform_base), "s"(m0v)
                 : "memory", "m0");
}

template <typename F, int... I>
__device__ __forceinline__ void asr_static_for_impl(F& f, std::integer_sequence<int, I...>) {
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, typename F>
__device__ __forceinline__ void asr_static_for(F& f) {
    asr_static_for_impl(f, std::make_integer_sequence<int, N>{});
}

inline int round_up(int v, int m) { return (v + m - 1) / m * m; }

}  // namespace
