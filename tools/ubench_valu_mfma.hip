// Micro-benchmark: how many VALU instructions can a third wave of a SIMD issue per "K-step" while the two MFMA waves of
// that SIMD keep the matrix pipe busy?  (Feasibility of a fused depthwise -> pointwise kernel: the depthwise producer wave
// shares the issue port of its SIMD with the MFMA waves.)
//   workgroup = 12 waves: waves 0-7 issue 96 MFMAs each per K-step (2 per SIMD -> 192 per SIMD, the product kernel's count),
//   waves 8-11 issue V VALU instructions per K-step (v_pk_fma_f32 / v_fma_f32 / v_mov_dpp mix), one s_barrier per K-step.
//   Prints cycles per K-step for V = 0 .. 640 and both MFMA shapes.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_valu_mfma.hip -o /tmp/ubv && /tmp/ubv
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// 16 independent VALU instructions of the given kind
template <int KIND>
__device__ __forceinline__ void valu16(f32x2 (&r)[16], const f32x2& w) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        if (KIND == 0) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(r[i]) : "v"(r[(i + 5) & 15]), "v"(w));
        else if (KIND == 1) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(r[i][0]) : "v"(r[(i + 5) & 15][1]), "v"(w[0]));
        else if (KIND == 2) asm volatile("v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(r[i][0]) : "v"(r[(i + 5) & 15][1]));
        else asm volatile("v_cvt_f16_f32 %0, %1" : "+v"(r[i][0]) : "v"(r[(i + 5) & 15][1]));
    }
}

template <int SHAPE, int KIND>
__global__ __launch_bounds__(768) void k(float* __restrict__ sink, long long* __restrict__ cyc, int ksteps, int v16, float seed) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float s = 0.f;
    const long long t0 = (long long)__builtin_readcyclecounter();
    if (wave < 8) {
        f16x8 a, b;
        for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(seed * (float)((lane * 7 + i * 3) % 13 - 6)); b[i] = (_Float16)(seed * (float)((lane * 5 + i) % 11 - 5)); }
        if (SHAPE == 32) {
            f32x16 acc[8];
            for (int j = 0; j < 8; ++j) for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
            for (int it = 0; it < ksteps; ++it) {
#pragma unroll
                for (int m = 0; m < 48; ++m) acc[m & 7] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[m & 7], 0, 0, 0);
                __builtin_amdgcn_s_barrier();
            }
            for (int j = 0; j < 8; ++j) s += acc[j][0] + acc[j][9];
        } else {
            f32x4 acc[32];
            for (int j = 0; j < 32; ++j) for (int e = 0; e < 4; ++e) acc[j][e] = 0.f;
            for (int it = 0; it < ksteps; ++it) {
#pragma unroll
                for (int m = 0; m < 96; ++m) acc[m & 31] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[m & 31], 0, 0, 0);
                __builtin_amdgcn_s_barrier();
            }
            for (int j = 0; j < 32; ++j) s += acc[j][0] + acc[j][3];
        }
    } else {
        f32x2 r[16], w;
        for (int i = 0; i < 16; ++i) { r[i][0] = seed * (float)(lane + i); r[i][1] = seed * (float)(lane - i); }
        w[0] = 0.5f; w[1] = 0.25f;
        for (int it = 0; it < ksteps; ++it) {
            for (int j = 0; j < v16; ++j) valu16<KIND>(r, w);
            __builtin_amdgcn_s_barrier();
        }
        for (int i = 0; i < 16; ++i) s += r[i][0] + r[i][1];
    }
    const long long t1 = (long long)__builtin_readcyclecounter();
    if (s == 123.456f) sink[0] = s;
    if (lane == 0) cyc[blockIdx.x * 12 + wave] = t1 - t0;
}

template <int SHAPE, int KIND>
void run(float* sink, long long* cyc, const char* name) {
    const int blocks = 256, ksteps = 400;
    printf("%s:", name);
    for (int v16 : {0, 8, 16, 20, 24, 28, 32, 40}) {
        hipLaunchKernelGGL((k<SHAPE, KIND>), dim3(blocks), dim3(768), 0, 0, sink, cyc, ksteps, v16, 0.01f);
        hipLaunchKernelGGL((k<SHAPE, KIND>), dim3(blocks), dim3(768), 0, 0, sink, cyc, ksteps, v16, 0.01f);
        hipDeviceSynchronize();
        std::vector<long long> h(blocks * 12);
        hipMemcpy(h.data(), cyc, sizeof(long long) * blocks * 12, hipMemcpyDeviceToHost);
        double c = 0;
        for (int b = 0; b < blocks; ++b) c += h[b * 12];
        printf("  V=%d: %.0f", v16 * 16, c / blocks / ksteps);
    }
    printf("   cycles per K-step (s_memtime ticks: 100 MHz x ? -- compare within a row)\n");
}

int main() {
    float* sink; long long* cyc;
    hipMalloc(&sink, 64); hipMalloc(&cyc, sizeof(long long) * 256 * 12);
    run<16, 0>(sink, cyc, "16x16x32 + v_pk_fma_f32 ");
    run<16, 1>(sink, cyc, "16x16x32 + v_fma_f32    ");
    run<16, 2>(sink, cyc, "16x16x32 + v_mov_dpp    ");
    run<16, 3>(sink, cyc, "16x16x32 + v_cvt_f16_f32");
    run<32, 0>(sink, cyc, "32x32x16 + v_pk_fma_f32 ");
    run<32, 1>(sink, cyc, "32x32x16 + v_fma_f32    ");
    return 0;
}
