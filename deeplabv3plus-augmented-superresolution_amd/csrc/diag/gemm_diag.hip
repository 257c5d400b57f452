// Diagnostic library only (csrc/build.py, ASR_BUILD_VARIANT=diag -> libasr_hip_diag.so; never part of libasr_hip.so, never shipped
// to the GPU box prebuilt): earlier forms of the pre-split pointwise GEMM, kept so that tools/ab_presplit_lw.py can A/B them
// against the product kernel IN ONE PROCESS (the chip's clock drifts by 10-20 % between processes), and a timing-only mock.
//   asr_diag_pwconv_presplit_8w   round 1: 8 waves that each request their own 8 pieces and then issue their MFMAs
//   asr_diag_pwconv_presplit_lw   round 2's product kernel: 8 MFMA waves + 4 loader waves, two 64 KB stages, the next
//                                 stage requested and waited for inside one K-step
//   asr_diag_pwconv_presplit_exp  MOCK of a fused depthwise -> pointwise layer (results are garbage: timing only)
// The product kernel (csrc/gemm.hip: pw_gemm_f16x3_pre_ring_kernel) differs from the round-2 form in the loaders' protocol
// only (a five-unit LDS ring, requests 1.5 - 2 K-steps ahead); all three real kernels issue the same MFMA sequence per
// accumulator and are bit-identical.
#include "../gemm_common.h"

namespace {

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




}  // namespace

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

// asr_pwconv_mfma_f16x3_presplit's arguments on round 2's loader-wave kernel (two 64 KB stages)
extern "C" int asr_diag_pwconv_presplit_lw(const void* x_split, const float* w_packed, const float* bias, const float* residual,
                                           float* y, int64_t m, int k, int n, int ldx_chunks, int ldy, int ldres, int relu,
                                           asr_stream_t stream) {
    ASR_REQUIRE(x_split && w_packed && y && m > 0 && k > 0 && n > 0, "asr_diag_pwconv_presplit_lw: bad arguments");
    PwArgs a{};
    a.x = reinterpret_cast<const float*>(x_split); a.wp = w_packed; a.bias = bias; a.res = residual; a.y = y;
    a.M = m; a.K = k; a.N = n; a.Npad = round_up(n, 128); a.Kpad = round_up(k, BK);
    a.ldx = ldx_chunks; a.ldy = ldy; a.ldres = ldres; a.relu = relu;
    a.taps = 1; a.cin = k; a.stride = 1; a.pad = 0; a.dil = 1;
    ASR_UNSUPPORTED(a.Npad % 256 != 0, "asr_diag_pwconv_presplit_lw: ceil128(n) must be a multiple of 256");
    a.tiles_n = (int)asr_cdiv(n, 256);
    const long long nwg = asr_cdiv(m, 256) * a.tiles_n;
    constexpr size_t lds = 2 * (256 * 128 + 2 * 4 * 256 * 16);
    auto kern = pw_gemm_f16x3_pre_lw_kernel<16, false>;
    static AsrDeviceOnce once;
    ASR_HIP_CHECK(asr_allow_dynamic_lds(once, reinterpret_cast<const void*>(kern), (int)lds));
    hipLaunchKernelGGL(kern, dim3((unsigned)nwg), dim3(768), lds, asr_stream(stream), a);
    ASR_LAUNCH_CHECK();
    return ASR_OK;
}
