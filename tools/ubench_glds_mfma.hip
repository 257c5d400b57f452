// Micro-benchmark: does LDS-DMA (global_load_lds_dwordx4) issue by one wave of a SIMD slow down the MFMAs of the other wave
// of the same SIMD?  One workgroup of 8 waves per CU: waves 0-3 ("loaders", one per SIMD) request PIECES 1 KiB pieces per
// iteration from an L2-resident buffer, waves 4-7 ("math") issue MFMAS independent v_mfma_f32_32x32x16_f16 per iteration.
// Modes: 1 = math only, 2 = loaders only, 3 = both.  Prints mean cycles per iteration of wave 4 (math) and wave 0 (loader).
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_glds_mfma.hip -o /tmp/ubench && /tmp/ubench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) void* lds_ptr;
typedef const __attribute__((address_space(1))) void* gbl_ptr;

template <int PIECES, int MFMAS>
__global__ __launch_bounds__(512) void k(const char* __restrict__ src, float* __restrict__ sink, long long* __restrict__ cyc, int iters, int mode) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool loader = wave < 4;
    f32x16 acc[8];
    for (int j = 0; j < 8; ++j) for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
    f16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(0.001f * (lane + i)); b[i] = (_Float16)(0.002f * (lane - i)); }
    const char* g = src + ((size_t)blockIdx.x * 65536 + (size_t)wave * 8192 + lane * 16);
    __syncthreads();
    const long long t0 = (long long)__builtin_readcyclecounter();
    if (loader) {                                              // the role branch is OUTSIDE the loops: two clean loop bodies
        if (mode & 2)
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int p = 0; p < PIECES; ++p)
                    __builtin_amdgcn_global_load_lds((gbl_ptr)(g + ((it * PIECES + p) & 7) * 1024), (lds_ptr)(lds + (wave * PIECES + p) * 1024), 16, 0, 0);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
    } else if (mode & 1) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int m = 0; m < MFMAS; ++m) acc[m & 7] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[m & 7], 0, 0, 0);
        }
    }
    const long long t1 = (long long)__builtin_readcyclecounter();
    float s = 0.f;
    for (int j = 0; j < 8; ++j) s += acc[j][0] + acc[j][7];
    if (s == 123.456f) sink[0] = s + lds[tid];
    if (lane == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}

template <int PIECES, int MFMAS>
void run(const char* src, float* sink, long long* cyc, int blocks) {
    const int iters = 2000;
    for (int mode = 1; mode <= 3; ++mode) {
        hipLaunchKernelGGL((k<PIECES, MFMAS>), dim3(blocks), dim3(512), 65536, 0, src, sink, cyc, iters, mode);
        hipDeviceSynchronize();
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL((k<PIECES, MFMAS>), dim3(blocks), dim3(512), 65536, 0, src, sink, cyc, iters, mode);
        hipEventRecord(e1, 0);
        hipDeviceSynchronize();
        float ms = 0.f;
        hipEventElapsedTime(&ms, e0, e1);
        std::vector<long long> h(blocks * 8);
        hipMemcpy(h.data(), cyc, sizeof(long long) * blocks * 8, hipMemcpyDeviceToHost);
        double ld = 0, mm = 0;
        for (int b = 0; b < blocks; ++b) { ld += h[b * 8 + 0]; mm += h[b * 8 + 4]; }
        printf("pieces/iter %2d mfma/iter %2d mode %d (%s): loader wave %8.1f ticks/iter   math wave %8.1f ticks/iter   kernel %7.1f us = %6.1f ns/iter\n",
               PIECES, MFMAS, mode, mode == 1 ? "math only " : (mode == 2 ? "loads only" : "both      "), ld / blocks / iters,
               mm / blocks / iters, ms * 1e3, ms * 1e6 / iters);
    }
}

int main() {
    const int blocks = 256;
    char* src; float* sink; long long* cyc;
    hipMalloc(&src, (size_t)blocks * 65536 + 65536);
    hipMemset(src, 1, (size_t)blocks * 65536 + 65536);
    hipMalloc(&sink, 4096);
    hipMalloc(&cyc, sizeof(long long) * blocks * 8);
    hipFuncSetAttribute(reinterpret_cast<const void*>(k<16, 48>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipFuncSetAttribute(reinterpret_cast<const void*>(k<8, 48>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipFuncSetAttribute(reinterpret_cast<const void*>(k<4, 24>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    run<16, 48>(src, sink, cyc, blocks);
    run<8, 48>(src, sink, cyc, blocks);
    run<4, 24>(src, sink, cyc, blocks);
    return 0;
}
