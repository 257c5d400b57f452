// Synthetic reproduction of the lanes-48-63 interaction, gfx950 only.
//   hipcc -O2 --offload-arch=gfx950 tools/ubench_pk_sgpr_hazard.hip -o /tmp/ub && /tmp/ub [trials]
// Round 4's variant matrix (profiles/r04_hazard_matrix.txt) showed that neither the SGPR operands of the victim's packed-f32
// instructions nor the saturating split / MODE write of the aggressor matter: what the aggressors have in common is that two of
// their waves leave just enough of a SIMD's 512 vector registers for ONE victim wave at the top of the file.  This program
// tests that directly:
//   VICTIM   256-thread workgroups whose waves allocate 96 registers and execute v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32 (vector
//            operands only) and the unpacked v_mul_f32 on explicitly numbered registers (low: v8.., high: v84..), comparing
//            every packed result, lane by lane, with the unpacked arithmetic.  A wave that sees a mismatch records its
//            HW_REG_GPR_ALLOC (physical register base / size) and HW_REG_HW_ID.
//   AGGRESSOR  one 512-thread workgroup per CU (120 KB of LDS keeps a second one out), NV registers per wave, plain v_fma
//            loop -- no MODE write, no conversions, no MFMA, no LDS traffic.  Two of its waves per SIMD occupy registers
//            0 .. 2 NV - 1, so a victim wave lands at base 2 NV when 512 - 2 NV >= 96.
// Output per aggressor NV: mismatches per instruction form, the lanes, and the register bases of the waves that failed / of all.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

typedef float f2 __attribute__((ext_vector_type(2)));

// bad[form * 64 + lane]; NFORMS instruction forms, each checked lane by lane against unpacked arithmetic written out in asm:
//   0 pk_mul low registers      1 pk_mul high registers     2 pk_add high        3 pk_fma high       4 unpacked v_mul (control)
//   5 pk_mul IN PLACE (dst = src0)                          6 pk_mul in place with the halves of src0 swapped (op_sel:[1,0] op_sel_hi:[0,1])
//   7 pk_add in place with neg_lo / neg_hi on src1          8 pk_mul with op_sel_hi:[1,0] (low half of src1 for both)
//   9 a chain of four dependent in-place packed ops (mul, add, mul swapped, add neg)
//  10 pk_add in place, src1 halves swapped + negated (op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1])   } the two forms that
//  11 pk_add in place, src0 halves swapped, src1 negated (op_sel:[1,0] op_sel_hi:[0,1] neg_lo:[0,1] neg_hi:[0,1]) } ONLY the failing
//  12 K_fwd's weight sequence: (a + 1.0) [op_sel_hi:[1,0] on the constant], then form 10 on it, then form 11       } builds of K_fwd hold
//  13 form 10 NOT in place
// alloc_all[base granule 0..63], alloc_bad[base granule]: histogram of VGPR_BASE (units of 8 registers) over all / failing waves
constexpr int NFORMS = 14;
// LOADS: eight 8-byte global loads per lane are in flight (into v60..v75, not otherwise used) while the packed ops execute --
// K_fwd of the SR solver issues its 18 tap loads and computes the next coordinates behind them.
template <bool LOADS>
__global__ __launch_bounds__(256) void victim(const float* __restrict__ src, unsigned* __restrict__ bad, unsigned* __restrict__ alloc_all,
                                              unsigned* __restrict__ alloc_bad, unsigned* __restrict__ first_bad, int iters) {
    const int lane = threadIdx.x & 63;
    f2 a = {src[lane] + 0.01f * (blockIdx.x & 63), src[64 + lane]}, b = {src[128 + lane], src[192 + lane] - 0.003f * (blockIdx.x & 31)};
    unsigned nbad[NFORMS];
#pragma unroll
    for (int f = 0; f < NFORMS; ++f) nbad[f] = 0;
    for (int it = 0; it < iters; ++it) {
        float r[NFORMS][2];
        if (LOADS) {
            const float* q = src + ((lane * 2 + it * 64) & 255);
            asm volatile(
                "global_load_dwordx2 v[60:61], %0, off\n global_load_dwordx2 v[62:63], %0, off offset:256\n"
                "global_load_dwordx2 v[64:65], %0, off offset:512\n global_load_dwordx2 v[66:67], %0, off offset:768\n"
                "global_load_dwordx2 v[68:69], %0, off offset:8\n global_load_dwordx2 v[70:71], %0, off offset:264\n"
                "global_load_dwordx2 v[72:73], %0, off offset:520\n global_load_dwordx2 v[74:75], %0, off offset:776\n"
                :: "v"(q) : "v60", "v61", "v62", "v63", "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73", "v74", "v75", "memory");
        }
        asm volatile(
            "v_mov_b32 v8, %28\n v_mov_b32 v9, %29\n v_mov_b32 v10, %30\n v_mov_b32 v11, %31\n"
            "v_mov_b32 v84, %28\n v_mov_b32 v85, %29\n v_mov_b32 v86, %30\n v_mov_b32 v87, %31\n"
            "v_pk_mul_f32 v[12:13], v[8:9], v[10:11]\n"
            "v_pk_mul_f32 v[88:89], v[84:85], v[86:87]\n"
            "v_pk_add_f32 v[90:91], v[84:85], v[86:87]\n"
            "v_pk_fma_f32 v[92:93], v[84:85], v[86:87], v[84:85]\n"
            "v_mul_f32 v94, v84, v86\n v_mul_f32 v95, v85, v87\n"
            "v_mov_b32 %0, v12\n v_mov_b32 %1, v13\n v_mov_b32 %2, v88\n v_mov_b32 %3, v89\n v_mov_b32 %4, v90\n v_mov_b32 %5, v91\n"
            "v_mov_b32 %6, v92\n v_mov_b32 %7, v93\n v_mov_b32 %8, v94\n v_mov_b32 %9, v95\n"
            // 5: in place
            "v_mov_b32 v88, v84\n v_mov_b32 v89, v85\n"
            "v_pk_mul_f32 v[88:89], v[88:89], v[86:87]\n"
            "v_mov_b32 %10, v88\n v_mov_b32 %11, v89\n"
            // 6: in place, halves of src0 swapped
            "v_mov_b32 v90, v84\n v_mov_b32 v91, v85\n"
            "v_pk_mul_f32 v[90:91], v[90:91], v[86:87] op_sel:[1,0] op_sel_hi:[0,1]\n"
            "v_mov_b32 %12, v90\n v_mov_b32 %13, v91\n"
            // 7: in place add with negated src1
            "v_mov_b32 v92, v84\n v_mov_b32 v93, v85\n"
            "v_pk_add_f32 v[92:93], v[92:93], v[86:87] neg_lo:[0,1] neg_hi:[0,1]\n"
            "v_mov_b32 %14, v92\n v_mov_b32 %15, v93\n"
            // 8: low half of src1 for both results
            "v_pk_mul_f32 v[94:95], v[84:85], v[86:87] op_sel_hi:[1,0]\n"
            "v_mov_b32 %16, v94\n v_mov_b32 %17, v95\n"
            // 9: chain of four dependent in-place ops
            "v_mov_b32 v88, v84\n v_mov_b32 v89, v85\n"
            "v_pk_mul_f32 v[88:89], v[88:89], v[86:87]\n"
            "v_pk_add_f32 v[88:89], v[88:89], v[84:85]\n"
            "v_pk_mul_f32 v[88:89], v[88:89], v[86:87] op_sel:[1,0] op_sel_hi:[0,1]\n"
            "v_pk_add_f32 v[88:89], v[88:89], v[86:87] neg_lo:[0,1] neg_hi:[0,1]\n"
            "v_mov_b32 %18, v88\n v_mov_b32 %19, v89\n"
            // 10: in place, src1 halves swapped and negated
            "v_mov_b32 v90, v84\n v_mov_b32 v91, v85\n"
            "v_pk_add_f32 v[90:91], v[90:91], v[86:87] op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n"
            "v_mov_b32 %20, v90\n v_mov_b32 %21, v91\n"
            // 11: in place, src0 halves swapped, src1 negated
            "v_mov_b32 v92, v84\n v_mov_b32 v93, v85\n"
            "v_pk_add_f32 v[92:93], v[92:93], v[86:87] op_sel:[1,0] op_sel_hi:[0,1] neg_lo:[0,1] neg_hi:[0,1]\n"
            "v_mov_b32 %22, v92\n v_mov_b32 %23, v93\n"
            // 12: K_fwd's weight sequence
            "v_mov_b32 v94, v86\n v_mov_b32 v95, v87\n"
            "v_pk_add_f32 v[88:89], v[84:85], 1.0 op_sel_hi:[1,0]\n"
            "v_pk_add_f32 v[88:89], v[88:89], v[94:95] op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n"
            "v_pk_add_f32 v[94:95], v[94:95], v[84:85] op_sel:[1,0] op_sel_hi:[0,1] neg_lo:[0,1] neg_hi:[0,1]\n"
            "v_add_f32 v88, v88, v94\n v_add_f32 v89, v89, v95\n"
            "v_mov_b32 %24, v88\n v_mov_b32 %25, v89\n"
            // 13: form 10 not in place
            "v_pk_add_f32 v[90:91], v[84:85], v[86:87] op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n"
            "v_mov_b32 %26, v90\n v_mov_b32 %27, v91\n"
            : "=&v"(r[0][0]), "=&v"(r[0][1]), "=&v"(r[1][0]), "=&v"(r[1][1]), "=&v"(r[2][0]), "=&v"(r[2][1]), "=&v"(r[3][0]), "=&v"(r[3][1]),
              "=&v"(r[4][0]), "=&v"(r[4][1]), "=&v"(r[5][0]), "=&v"(r[5][1]), "=&v"(r[6][0]), "=&v"(r[6][1]), "=&v"(r[7][0]), "=&v"(r[7][1]),
              "=&v"(r[8][0]), "=&v"(r[8][1]), "=&v"(r[9][0]), "=&v"(r[9][1]), "=&v"(r[10][0]), "=&v"(r[10][1]), "=&v"(r[11][0]), "=&v"(r[11][1]),
              "=&v"(r[12][0]), "=&v"(r[12][1]), "=&v"(r[13][0]), "=&v"(r[13][1])
            : "v"(a.x), "v"(a.y), "v"(b.x), "v"(b.y)
            : "v8", "v9", "v10", "v11", "v12", "v13", "v84", "v85", "v86", "v87", "v88", "v89", "v90", "v91", "v92", "v93", "v94", "v95",
              "v60", "v61", "v62", "v63", "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73", "v74", "v75");   // (loads may be landing there)
        if (LOADS) asm volatile("s_waitcnt vmcnt(0)" ::: "v60", "v61", "v62", "v63", "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73", "v74", "v75", "memory");
        // the unpacked arithmetic, written out so that the compiler cannot pair it
        auto mul = [](float x, float y) { float z; asm volatile("v_mul_f32 %0, %1, %2" : "=v"(z) : "v"(x), "v"(y)); return z; };
        auto add = [](float x, float y) { float z; asm volatile("v_add_f32 %0, %1, %2" : "=v"(z) : "v"(x), "v"(y)); return z; };
        auto sub = [](float x, float y) { float z; asm volatile("v_sub_f32 %0, %1, %2" : "=v"(z) : "v"(x), "v"(y)); return z; };
        auto fma = [](float x, float y, float w) { float z; asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(z) : "v"(x), "v"(y), "v"(w)); return z; };
        float e[NFORMS][2];
        e[0][0] = mul(a.x, b.x); e[0][1] = mul(a.y, b.y);
        e[1][0] = e[0][0]; e[1][1] = e[0][1];
        e[2][0] = add(a.x, b.x); e[2][1] = add(a.y, b.y);
        e[3][0] = fma(a.x, b.x, a.x); e[3][1] = fma(a.y, b.y, a.y);
        e[4][0] = e[0][0]; e[4][1] = e[0][1];
        e[5][0] = e[0][0]; e[5][1] = e[0][1];
        e[6][0] = mul(a.y, b.x); e[6][1] = mul(a.x, b.y);
        e[7][0] = sub(a.x, b.x); e[7][1] = sub(a.y, b.y);
        e[8][0] = e[0][0]; e[8][1] = mul(a.y, b.x);
        {
            float c0 = add(e[0][0], a.x), c1 = add(e[0][1], a.y);         // after mul, add
            float d0 = mul(c1, b.x), d1 = mul(c0, b.y);                   // swapped mul
            e[9][0] = sub(d0, b.x); e[9][1] = sub(d1, b.y);
        }
        e[10][0] = sub(a.x, b.y); e[10][1] = sub(a.y, b.x);           // lo = src0.lo - src1.hi, hi = src0.hi - src1.lo
        e[11][0] = sub(a.y, b.x); e[11][1] = sub(a.x, b.y);           // lo = src0.hi - src1.lo, hi = src0.lo - src1.hi
        {
            const float p0 = add(a.x, 1.0f), p1 = add(a.y, 1.0f);     // (a + 1)
            const float q0 = sub(p0, b.y), q1 = sub(p1, b.x);         // form 10 on it with src1 = b
            const float w0 = sub(b.y, a.x), w1 = sub(b.x, a.y);       // form 11: src0 = b (halves swapped), src1 = a negated
            e[12][0] = add(q0, w0); e[12][1] = add(q1, w1);
        }
        e[13][0] = e[10][0]; e[13][1] = e[10][1];
#pragma unroll
        for (int f = 0; f < NFORMS; ++f)
            nbad[f] += (__float_as_int(r[f][0]) != __float_as_int(e[f][0])) | (__float_as_int(r[f][1]) != __float_as_int(e[f][1]));
        a.x += 0.5f;
        b.y -= 0.25f;
    }
    unsigned any = 0;
#pragma unroll
    for (int f = 0; f < NFORMS; ++f) {
        if (nbad[f]) atomicAdd(&bad[f * 64 + lane], nbad[f]);
        any |= nbad[f];
    }
    const unsigned alloc = __builtin_amdgcn_s_getreg(5 | (0 << 6) | (31 << 11));     // HW_REG_GPR_ALLOC, all 32 bits
    const unsigned hwid = __builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11));      // HW_REG_HW_ID
    const bool wave_bad = __builtin_amdgcn_ballot_w64(any != 0) != 0;
    if (lane == 0) {
        atomicAdd(&alloc_all[alloc & 63], 1u);
        if (wave_bad) {
            atomicAdd(&alloc_bad[alloc & 63], 1u);
            if (atomicAdd(&first_bad[0], 1u) < 8) {
                const unsigned k = atomicAdd(&first_bad[1], 1u);
                if (k < 8) { first_bad[2 + 2 * k] = alloc; first_bad[3 + 2 * k] = hwid; }
            }
        }
    }
}

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef unsigned u4 __attribute__((ext_vector_type(4)));

// MODE: 0 v_fma only | 1 v_mfma_f32_16x16x32_f16 | 2 v_mfma_f32_32x32x16_f16 | 3 LDS 16-byte writes + reads | 4 DPP row shifts
//       5 v_permlane16_swap | 6 global 16-byte loads + stores | 7 the packed saturating split (v_cvt_pk_f16_f32, SDWA, FP16_OVFL)
//       8 all of them in one loop
template <int NV, int MODE>
__global__ __launch_bounds__(512) void aggressor(const float* __restrict__ src, float* __restrict__ sink, int iters) {
    extern __shared__ char lds[];
    const int tid = threadIdx.x;
    float x = src[tid & 255], acc = 0.f;
    if (NV == 200) asm volatile("v_mov_b32 v199, 0" ::: "v199");
    if (NV == 208) asm volatile("v_mov_b32 v207, 0" ::: "v207");
    if (NV == 216) asm volatile("v_mov_b32 v215, 0" ::: "v215");
    if (NV == 128) asm volatile("v_mov_b32 v127, 0" ::: "v127");
    if (NV == 160) asm volatile("v_mov_b32 v159, 0" ::: "v159");
    if (MODE == 7 || MODE == 8) __builtin_amdgcn_s_setreg(1 | (23 << 6), 1);
    f4 m4 = {0.f, 0.f, 0.f, 0.f};
    f16v m16;
#pragma unroll
    for (int i = 0; i < 16; ++i) m16[i] = 0.f;
    h8 fa;
#pragma unroll
    for (int i = 0; i < 8; ++i) fa[i] = (_Float16)(0.01f * (tid & 63) + 0.1f * i);
    u4 q = {(unsigned)tid, 1u, 2u, 3u};
    unsigned uacc = 0;
    const u4* gsrc = reinterpret_cast<const u4*>(src);
    u4* gdst = reinterpret_cast<u4*>(sink + (1 << 20)) + blockIdx.x * 512 + tid;      // sink holds 4 M floats: [1 M, 1 M + grid * 2048)
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0 || MODE == 8) {
            acc = acc * 1.0001f + x;
            x += 0.001f;
        }
        if (MODE == 1 || MODE == 8) m4 = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa, fa, m4, 0, 0, 0);
        if (MODE == 2 || MODE == 8) m16 = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa, fa, m16, 0, 0, 0);
        if (MODE == 3 || MODE == 8) {
            reinterpret_cast<u4*>(lds)[tid + 512 * (it & 7)] = q;
            __builtin_amdgcn_s_barrier();
            q += reinterpret_cast<u4*>(lds)[((tid * 5) & 511) + 512 * (it & 7)];
        }
        if (MODE == 4 || MODE == 8) {
            q.x += __builtin_amdgcn_mov_dpp((int)q.y, 0x111, 0xF, 0xF, true);
            q.y += __builtin_amdgcn_mov_dpp((int)q.x, 0x101, 0xF, 0xF, true);
        }
        if (MODE == 5 || MODE == 8) {
            const auto sw = __builtin_amdgcn_permlane16_swap(q.z, q.w, false, false);
            q.z = sw[0] + 1u;
            q.w = sw[1] + 3u;
        }
        if (MODE == 6 || MODE == 8) {
            const u4 g = gsrc[(tid + it) & 127];
            uacc += g.x + g.w;
            if ((it & 15) == 0) *gdst = q;
        }
        if (MODE == 7 || MODE == 8) {
            f2 a = {x + (float)it, acc + 1.5f}, b = {x * 3.f, (float)it * 0.37f};
            typedef _Float16 h2 __attribute__((ext_vector_type(2)));
            const h2 ha = __builtin_convertvector(a, h2), hb = __builtin_convertvector(b, h2);
            const f2 ra = a - __builtin_convertvector(ha, f2), rb = b - __builtin_convertvector(hb, f2);
            const h2 la = __builtin_convertvector(ra, h2), lb = __builtin_convertvector(rb, h2);
            uacc += __builtin_bit_cast(unsigned, ha) + 3u * __builtin_bit_cast(unsigned, hb) + 5u * __builtin_bit_cast(unsigned, la) +
                    7u * __builtin_bit_cast(unsigned, lb);
            x += 0.001f;
        }
    }
    float r = acc + (float)uacc + m4[0] + m4[1] + m4[2] + m4[3] + (float)(q.x + q.y + q.z + q.w);
#pragma unroll
    for (int i = 0; i < 16; ++i) r += m16[i];
    sink[blockIdx.x * 512 + tid] = r;
}

template <int NV, int MODE>
static void launch_aggr(hipStream_t s, const float* src, float* sink, int grid, int iters) {
    const int ldsb = 120 * 1024;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(aggressor<NV, MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, ldsb));
    hipLaunchKernelGGL((aggressor<NV, MODE>), dim3(grid), dim3(512), ldsb, s, src, sink, iters);
}

int main(int argc, char** argv) {
    const int trials = argc > 1 ? atoi(argv[1]) : 6;
    hipStream_t sa, sb;
    CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
    float *src, *sink;
    unsigned* stats;                                            // bad[NFORMS * 64] | alloc_all[64] | alloc_bad[64] | first_bad[18]
    CK(hipMalloc(&src, 512 * 4)); CK(hipMalloc(&sink, (size_t)4 << 22)); CK(hipMalloc(&stats, 2048 * 4));
    std::vector<float> hs(512);
    for (int i = 0; i < 512; ++i) hs[i] = 0.0137f * i - 1.3f;
    CK(hipMemcpy(src, hs.data(), 2048, hipMemcpyHostToDevice));
    const int vgrid = 6400, viters = 32;                        // 25 600 victim waves per launch, like one K_fwd launch
    const int agrid = 256;
    struct Aggr { void (*fn)(hipStream_t, const float*, float*, int, int); int nv, iters; const char* what; };
    const Aggr aggr[] = {
        {launch_aggr<200, 0>, 200, 400000, "v_fma only"},
        {launch_aggr<200, 1>, 200, 800000, "v_mfma_f32_16x16x32_f16"},
        {launch_aggr<200, 2>, 200, 400000, "v_mfma_f32_32x32x16_f16"},
        {launch_aggr<200, 8>, 200, 60000, "all ingredients"},
        {launch_aggr<216, 1>, 216, 800000, "v_mfma_f32_16x16x32_f16 (victims cannot co-reside)"},
    };
    const int naggr = sizeof(aggr) / sizeof(aggr[0]);
    const char* forms[NFORMS] = {"pk_mul v8..", "pk_mul v84..", "pk_add v84..", "pk_fma v84..", "v_mul v84.. (unpacked)", "pk_mul in place", "pk_mul in place, src0 halves swapped",
                                 "pk_add in place, neg src1", "pk_mul op_sel_hi:[1,0]", "chain of 4 in-place packed ops",
                                 "pk_add in place, src1 swapped + neg", "pk_add in place, src0 swapped, src1 neg", "K_fwd weight sequence", "pk_add src1 swapped + neg, not in place"};
    for (int with_loads = 0; with_loads < 2; ++with_loads)
    for (int m = -1; m < naggr; ++m) {
        CK(hipMemset(stats, 0, 2048 * 4));
        hipEvent_t e0, e1, v0, v1;
        CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&v0)); CK(hipEventCreate(&v1));
        float ms_a = 0.f, ms_v = 0.f;
        for (int t = 0; t < trials; ++t) {
            if (m >= 0) {
                CK(hipEventRecord(e0, sa));
                aggr[m].fn(sa, src, sink, agrid, aggr[m].iters);
                CK(hipEventRecord(e1, sa));
            }
            CK(hipEventRecord(v0, sb));
            for (int k = 0; k < 8; ++k) {
                if (with_loads) hipLaunchKernelGGL(victim<true>, dim3(vgrid), dim3(256), 0, sb, src, stats, stats + NFORMS * 64, stats + NFORMS * 64 + 64, stats + NFORMS * 64 + 128, viters);
                else hipLaunchKernelGGL(victim<false>, dim3(vgrid), dim3(256), 0, sb, src, stats, stats + NFORMS * 64, stats + NFORMS * 64 + 64, stats + NFORMS * 64 + 128, viters);
            }
            CK(hipEventRecord(v1, sb));
            CK(hipGetLastError());
            CK(hipDeviceSynchronize());
            if (m >= 0) CK(hipEventElapsedTime(&ms_a, e0, e1));
            CK(hipEventElapsedTime(&ms_v, v0, v1));
        }
        unsigned h[2048];
        CK(hipMemcpy(h, stats, 8192, hipMemcpyDeviceToHost));
        const int A0 = NFORMS * 64, B0 = A0 + 64, F0 = A0 + 128;
        printf("[victim %s loads in flight] ", with_loads ? "WITH" : "without");
        if (m < 0) printf("no aggressor (victims %.2f ms per 8 launches):\n", ms_v);
        else printf("aggressor: %s, %d registers per wave (%.2f ms per launch; victims %.2f ms per 8 launches):\n", aggr[m].what, aggr[m].nv, ms_a, ms_v);
        for (int f = 0; f < NFORMS; ++f) {
            unsigned long long tot = 0; int lo = 64, hi = -1;
            for (int l = 0; l < 64; ++l) if (h[f * 64 + l]) { tot += h[f * 64 + l]; lo = l < lo ? l : lo; hi = l > hi ? l : hi; }
            if (tot) printf("    %-38s %llu mismatching results, lanes %d..%d\n", forms[f], tot, lo, hi);
            else printf("    %-38s 0\n", forms[f]);
        }
        printf("    victim waves by VGPR base (registers: waves [failing]):");
        for (int g = 0; g < 64; ++g) if (h[A0 + g]) printf(" %d: %u [%u]", g * 8, h[A0 + g], h[B0 + g]);
        printf("\n");
        for (unsigned k = 0; k < 8 && k < h[F0 + 1]; ++k) printf("    failing wave: GPR_ALLOC 0x%08x HW_ID 0x%08x\n", h[F0 + 2 + 2 * k], h[F0 + 3 + 2 * k]);
        fflush(stdout);
    }
    return 0;
}
