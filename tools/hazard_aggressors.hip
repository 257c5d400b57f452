// Synthetic aggressors for tools/diag_sr_stages_under_stem.py (DIAG_REPLAY=synthetic:<mode>:<registers>:<iterations>): the real
// K_fwd of the SR solver as the victim, beside workgroups that do ONE thing.  One 512-thread workgroup per CU
// (120 KB of LDS keeps a second one out), NV registers per wave, bounded loops.
//   hipcc -O2 --offload-arch=gfx950 -shared -fPIC tools/hazard_aggressors.hip -o /tmp/libhazard_aggressors.so
#include <hip/hip_runtime.h>
typedef float f2 __attribute__((ext_vector_type(2)));
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


static float* g_src = nullptr;
static float* g_sink = nullptr;

template <int NV, int MODE>
static int launch(hipStream_t s, int iters) {
    const int ldsb = 120 * 1024;
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(aggressor<NV, MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, ldsb) != hipSuccess) return -2;
    hipLaunchKernelGGL((aggressor<NV, MODE>), dim3(256), dim3(512), ldsb, s, g_src, g_sink, iters);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

extern "C" int hazard_aggressor(int mode, int registers, int iters, void* stream) {
    if (!g_src) {
        if (hipMalloc(&g_src, 512 * 4) != hipSuccess || hipMalloc(&g_sink, (size_t)4 << 22) != hipSuccess) return -1;
        hipMemset(g_src, 0, 512 * 4);
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
#define CASE(NV, M) if (registers == NV && mode == M) return launch<NV, M>(s, iters);
    CASE(200, 0) CASE(200, 1) CASE(200, 2) CASE(200, 3) CASE(200, 4) CASE(200, 5) CASE(200, 6) CASE(200, 7) CASE(200, 8)
    CASE(216, 0) CASE(216, 1) CASE(216, 8) CASE(208, 0) CASE(208, 1) CASE(208, 8) CASE(160, 1) CASE(128, 1)
#undef CASE
    return -4;
}
