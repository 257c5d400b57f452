// The minimal trigger pair of the "lanes 48-63" defect, gfx950 / MI355X.
//   hipcc -O2 --offload-arch=gfx950 tools/ubench_pk_opsel_erratum.hip -o /tmp/ubo && /tmp/ubo [trials]
// VICTIM: waves of 96 registers that execute ONE packed-f32 instruction form at a time on fixed registers (sources rewritten by
//   v_mov_b32 before, result read by v_mov_b32 after) and compare both halves, lane by lane, with unpacked arithmetic written out in
//   asm.  The forms vary the operand-select (op_sel / op_sel_hi: which 32-bit half of a 64-bit source feeds the low / high result)
//   and negate modifiers of v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32.
// AGGRESSOR (another stream): one 512-thread workgroup per CU (120 KB of LDS), 200 registers per wave -- two of its waves leave
//   112 of a SIMD's 512 registers, so exactly one victim wave is co-resident per SIMD -- running a bare v_mfma_f32_16x16x32_f16 loop,
//   or a plain v_fma loop (control), or nothing.
// Round 4 result (profiles/r04_hazard_matrix.txt): beside the MFMA loop, and only there, the forms whose SRC1 halves are swapped
// return wrong values in lanes 48-63.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

constexpr int NFORMS = 21;
static const char* kForms[NFORMS] = {
    "pk_add plain (control)",
    "pk_add src1 swapped            op_sel:[0,1] op_sel_hi:[1,0]",
    "pk_add src1 swapped, negated   op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]",
    "pk_mul src1 swapped            op_sel:[0,1] op_sel_hi:[1,0]",
    "pk_add src0 swapped, negated   op_sel:[1,0] op_sel_hi:[0,1] neg_lo:[1,0] neg_hi:[1,0]",
    "pk_add src1 low half for both, negated   op_sel_hi:[1,0] neg",
    "pk_add src1 high half for both, negated  op_sel:[0,1] neg",
    "pk_fma src1 swapped            op_sel:[0,1,0] op_sel_hi:[1,0,1]",
    "pk_add src1 swapped, neg_lo only",
    "pk_add src1 swapped, neg_hi only",
    "pk_add src1 = SGPR pair, swapped, negated",
    "pk_add BOTH sources swapped    op_sel:[1,1] op_sel_hi:[0,0]",
    "pk_mul src1 swapped, negated",
    "pk_add src0 swapped            op_sel:[1,0] op_sel_hi:[0,1]",
    "pk_mul src0 swapped            op_sel:[1,0] op_sel_hi:[0,1]",
    "pk_fma src2 swapped            op_sel:[0,0,1] op_sel_hi:[1,1,0]",
    "v_pk_mov_b32 op_sel:[0,1]      (low = src0.lo, high = src1.hi)",
    "v_pk_mov_b32 op_sel:[1,0]      (low = src0.hi, high = src1.lo)",
    "v_pk_mov_b32 op_sel:[1,1]",
    "v_pk_add_f16 src1 swapped      op_sel:[0,1] op_sel_hi:[1,0]   (packed f16: halves are 16-bit)",
    "v_pk_fma_f32 op_sel:[0,1,1] op_sel_hi:[1,0,0]   (src1 AND src2 swapped)",
};

// one form: v84:85 = a, v86:87 = b, v88:89 = c (fma addend / scratch), result in v90:91
#define FORM(ASM, OUT0, OUT1)                                                                                              \
    asm volatile("v_mov_b32 v84, %2\n v_mov_b32 v85, %3\n v_mov_b32 v86, %4\n v_mov_b32 v87, %5\n v_mov_b32 v88, %2\n v_mov_b32 v89, %5\n" \
                 ASM "\n v_mov_b32 %0, v90\n v_mov_b32 %1, v91\n"                                                            \
                 : "=&v"(OUT0), "=&v"(OUT1) : "v"(a.x), "v"(a.y), "v"(b.x), "v"(b.y), "s"(sb)                               \
                 : "v84", "v85", "v86", "v87", "v88", "v89", "v90", "v91", "v95")

__global__ __launch_bounds__(256) void victim(const float* __restrict__ src, unsigned* __restrict__ bad, unsigned* __restrict__ waves, int iters) {
    const int lane = threadIdx.x & 63;
    asm volatile("v_mov_b32 v95, 0" ::: "v95");                  // 96 registers per wave, like K_fwd with packed-f32
    f2 a = {src[lane] + 0.01f * (blockIdx.x & 63), src[64 + lane]}, b = {src[128 + lane], src[192 + lane] - 0.003f * (blockIdx.x & 31)};
    f2 sb = {src[256 + (blockIdx.x & 7)], src[264 + (blockIdx.x & 7)]};       // wave-uniform pair -> SGPRs (form 10)
    sb.x = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(sb.x)));
    sb.y = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(sb.y)));
    auto mul = [](float x, float y) { float z; asm volatile("v_mul_f32 %0, %1, %2" : "=v"(z) : "v"(x), "v"(y)); return z; };
    auto add = [](float x, float y) { float z; asm volatile("v_add_f32 %0, %1, %2" : "=v"(z) : "v"(x), "v"(y)); return z; };
    auto sub = [](float x, float y) { float z; asm volatile("v_sub_f32 %0, %1, %2" : "=v"(z) : "v"(x), "v"(y)); return z; };
    auto fma = [](float x, float y, float w) { float z; asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(z) : "v"(x), "v"(y), "v"(w)); return z; };
    unsigned nbad[NFORMS];
#pragma unroll
    for (int f = 0; f < NFORMS; ++f) nbad[f] = 0;
    for (int it = 0; it < iters; ++it) {
        float r[NFORMS][2], e[NFORMS][2];
        const float cx = a.x, cy = b.y;                          // v88:89 = (a.x, b.y)
        FORM("v_pk_add_f32 v[90:91], v[84:85], v[86:87]", r[0][0], r[0][1]);
        e[0][0] = add(a.x, b.x); e[0][1] = add(a.y, b.y);
        FORM("v_pk_add_f32 v[90:91], v[84:85], v[86:87] op_sel:[0,1] op_sel_hi:[1,0]", r[1][0], r[1][1]);
        e[1][0] = add(a.x, b.y); e[1][1] = add(a.y, b.x);
        FORM("v_pk_add_f32 v[90:91], v[84:85], v[86:87] op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]", r[2][0], r[2][1]);
        e[2][0] = sub(a.x, b.y); e[2][1] = sub(a.y, b.x);
        FORM("v_pk_mul_f32 v[90:91], v[84:85], v[86:87] op_sel:[0,1] op_sel_hi:[1,0]", r[3][0], r[3][1]);
        e[3][0] = mul(a.x, b.y); e[3][1] = mul(a.y, b.x);
        FORM("v_pk_add_f32 v[90:91], v[84:85], v[86:87] op_sel:[1,0] op_sel_hi:[0,1] neg_lo:[1,0] neg_hi:[1,0]", r[4][0], r[4][1]);
        e[4][0] = sub(b.x, a.y); e[4][1] = sub(b.y, a.x);
        FORM("v_pk_add_f32 v[90:91], v[84:85], v[86:87] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]", r[5][0], r[5][1]);
        e[5][0] = sub(a.x, b.x); e[5][1] = sub(a.y, b.x);
        FORM("v_pk_add_f32 v[90:91], v[84:85], v[86:87] op_sel:[0,1] neg_lo:[0,1] neg_hi:[0,1]", r[6][0], r[6][1]);
        e[6][0] = sub(a.x, b.y); e[6][1] = sub(a.y, b.y);
        FORM("v_pk_fma_f32 v[90:91], v[84:85], v[86:87], v[88:89] op_sel:[0,1,0] op_sel_hi:[1,0,1]", r[7][0], r[7][1]);
        e[7][0] = fma(a.x, b.y, cx); e[7][1] = fma(a.y, b.x, cy);
        FORM("v_pk_add_f32 v[90:91], v[84:85], v[86:87] op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]", r[8][0], r[8][1]);
        e[8][0] = sub(a.x, b.y); e[8][1] = add(a.y, b.x);
        FORM("v_pk_add_f32 v[90:91], v[84:85], v[86:87] op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]", r[9][0], r[9][1]);
        e[9][0] = add(a.x, b.y); e[9][1] = sub(a.y, b.x);
        FORM("v_pk_add_f32 v[90:91], v[84:85], %6 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]", r[10][0], r[10][1]);
        e[10][0] = sub(a.x, sb.y); e[10][1] = sub(a.y, sb.x);
        FORM("v_pk_add_f32 v[90:91], v[84:85], v[86:87] op_sel:[1,1] op_sel_hi:[0,0]", r[11][0], r[11][1]);
        e[11][0] = add(a.y, b.y); e[11][1] = add(a.x, b.x);
        FORM("v_pk_mul_f32 v[90:91], v[84:85], v[86:87] op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]", r[12][0], r[12][1]);
        e[12][0] = mul(a.x, -b.y); e[12][1] = mul(a.y, -b.x);
        FORM("v_pk_add_f32 v[90:91], v[84:85], v[86:87] op_sel:[1,0] op_sel_hi:[0,1]", r[13][0], r[13][1]);
        e[13][0] = add(a.y, b.x); e[13][1] = add(a.x, b.y);
        FORM("v_pk_mul_f32 v[90:91], v[84:85], v[86:87] op_sel:[1,0] op_sel_hi:[0,1]", r[14][0], r[14][1]);
        e[14][0] = mul(a.y, b.x); e[14][1] = mul(a.x, b.y);
        FORM("v_pk_fma_f32 v[90:91], v[84:85], v[86:87], v[88:89] op_sel:[0,0,1] op_sel_hi:[1,1,0]", r[15][0], r[15][1]);
        e[15][0] = fma(a.x, b.x, cy); e[15][1] = fma(a.y, b.y, cx);
        FORM("v_pk_mov_b32 v[90:91], v[84:85], v[86:87] op_sel:[0,1]", r[16][0], r[16][1]);
        e[16][0] = a.x; e[16][1] = b.y;
        FORM("v_pk_mov_b32 v[90:91], v[84:85], v[86:87] op_sel:[1,0]", r[17][0], r[17][1]);
        e[17][0] = a.y; e[17][1] = b.x;
        FORM("v_pk_mov_b32 v[90:91], v[84:85], v[86:87] op_sel:[1,1]", r[18][0], r[18][1]);
        e[18][0] = a.y; e[18][1] = b.y;
        {   // packed f16 add on the low dwords: v90 = {lo: a.lo16 + b.hi16, hi: a.hi16 + b.lo16}; compared with the unswapped add of b rotated
            float rr, ee;
            asm volatile("v_mov_b32 v84, %2\n v_mov_b32 v86, %3\n v_pk_add_f16 v90, v84, v86 op_sel:[0,1] op_sel_hi:[1,0]\n v_mov_b32 %0, v90\n"
                         "v_alignbit_b32 v87, v86, v86, 16\n v_pk_add_f16 v91, v84, v87\n v_mov_b32 %1, v91\n"
                         : "=&v"(rr), "=&v"(ee) : "v"(a.x), "v"(b.x) : "v84", "v86", "v87", "v90", "v91");
            r[19][0] = rr; r[19][1] = 0.f; e[19][0] = ee; e[19][1] = 0.f;
        }
        FORM("v_pk_fma_f32 v[90:91], v[84:85], v[86:87], v[88:89] op_sel:[0,1,1] op_sel_hi:[1,0,0]", r[20][0], r[20][1]);
        e[20][0] = fma(a.x, b.y, cy); e[20][1] = fma(a.y, b.x, cx);
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
    const unsigned alloc = __builtin_amdgcn_s_getreg(5 | (0 << 6) | (31 << 11));     // HW_REG_GPR_ALLOC: VGPR base in units of 8
    const bool wave_bad = __builtin_amdgcn_ballot_w64(any != 0) != 0;
    if (lane == 0) {
        atomicAdd(&waves[2 * (alloc & 63)], 1u);
        if (wave_bad) atomicAdd(&waves[2 * (alloc & 63) + 1], 1u);
    }
}

template <int MODE>      // 0: v_fma loop, 1: v_mfma_f32_16x16x32_f16 loop; 200 registers per wave
__global__ __launch_bounds__(512) void aggressor(const float* __restrict__ src, float* __restrict__ sink, int iters) {
    extern __shared__ char lds[];
    const int tid = threadIdx.x;
    asm volatile("v_mov_b32 v199, 0" ::: "v199");
    float x = src[tid & 255], acc = 0.f;
    f4 m4 = {0.f, 0.f, 0.f, 0.f};
    h8 fa;
#pragma unroll
    for (int i = 0; i < 8; ++i) fa[i] = (_Float16)(0.01f * (tid & 63) + 0.1f * i);
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) { acc = acc * 1.0001f + x; x += 0.001f; }
        else m4 = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa, fa, m4, 0, 0, 0);
    }
    if (tid == 0 && iters < 0) lds[0] = 1;
    sink[blockIdx.x * 512 + tid] = acc + m4[0] + m4[1] + m4[2] + m4[3];
}

int main(int argc, char** argv) {
    const int trials = argc > 1 ? atoi(argv[1]) : 3;
    hipStream_t sa, sb;
    CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
    float *src, *sink;
    unsigned* stats;                                            // bad[NFORMS * 64] | waves[64][2]
    CK(hipMalloc(&src, 512 * 4)); CK(hipMalloc(&sink, 256 * 512 * 4)); CK(hipMalloc(&stats, (NFORMS * 64 + 128) * 4));
    float hs[512];
    for (int i = 0; i < 512; ++i) hs[i] = 0.0137f * i - 1.3f;
    CK(hipMemcpy(src, hs, 2048, hipMemcpyHostToDevice));
    const int ldsb = 120 * 1024;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(aggressor<0>), hipFuncAttributeMaxDynamicSharedMemorySize, ldsb));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(aggressor<1>), hipFuncAttributeMaxDynamicSharedMemorySize, ldsb));
    const char* names[3] = {"no aggressor", "aggressor: v_fma loop, 200 registers per wave", "aggressor: v_mfma_f32_16x16x32_f16 loop, 200 registers per wave"};
    for (int m = 0; m < 3; ++m) {
        CK(hipMemset(stats, 0, (NFORMS * 64 + 128) * 4));
        for (int t = 0; t < trials; ++t) {
            if (m == 1) hipLaunchKernelGGL(aggressor<0>, dim3(256), dim3(512), ldsb, sa, src, sink, 400000);
            if (m == 2) hipLaunchKernelGGL(aggressor<1>, dim3(256), dim3(512), ldsb, sa, src, sink, 800000);
            for (int k = 0; k < 4; ++k) hipLaunchKernelGGL(victim, dim3(6400), dim3(256), 0, sb, src, stats, stats + NFORMS * 64, 16);
            CK(hipGetLastError());
            CK(hipDeviceSynchronize());
        }
        unsigned h[NFORMS * 64 + 128];
        CK(hipMemcpy(h, stats, sizeof(h), hipMemcpyDeviceToHost));
        unsigned long long w_all = 0, w_co = 0, w_bad = 0;
        for (int g = 0; g < 64; ++g) { w_all += h[NFORMS * 64 + 2 * g]; w_bad += h[NFORMS * 64 + 2 * g + 1]; if (g * 8 >= 400) w_co += h[NFORMS * 64 + 2 * g]; }
        printf("%s: %llu victim waves, %llu of them at register base >= 400 (beside two aggressor waves), %llu waves with a mismatch\n", names[m], w_all, w_co, w_bad);
        for (int f = 0; f < NFORMS; ++f) {
            unsigned long long tot = 0; int lo = 64, hi = -1;
            for (int l = 0; l < 64; ++l) if (h[f * 64 + l]) { tot += h[f * 64 + l]; lo = l < lo ? l : lo; hi = l > hi ? l : hi; }
            if (tot) printf("    [%2d] %-88s %10llu WRONG, lanes %d..%d\n", f, kForms[f], tot, lo, hi);
            else printf("    [%2d] %-88s          0\n", f, kForms[f]);
        }
        fflush(stdout);
    }
    return 0;
}
