// Micro-benchmark: sustained f16 MFMA rate of the two dense shapes under the chip's power management, all SIMDs busy.
//   v_mfma_f32_32x32x16_f16 (32768 flop) against v_mfma_f32_16x16x32_f16 (16384 flop), 1 or 2 waves per SIMD, operands in
//   registers.  Prints cycles (s_memtime) per MFMA, wall time, the implied clock and TFLOP/s.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_mfma_shapes.hip -o /tmp/ubs && /tmp/ubs
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int SHAPE>
__global__ __launch_bounds__(512) void k(float* __restrict__ sink, long long* __restrict__ cyc, int iters, float seed) {
    const int tid = threadIdx.x, lane = tid & 63;
    f16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(seed * (float)((lane * 7 + i * 3) % 13 - 6)); b[i] = (_Float16)(seed * (float)((lane * 5 + i) % 11 - 5)); }
    float s = 0.f;
    const long long t0 = (long long)__builtin_readcyclecounter();
    if (SHAPE == 32) {
        f32x16 acc[8];
        for (int j = 0; j < 8; ++j) for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int m = 0; m < 48; ++m) acc[m & 7] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[m & 7], 0, 0, 0);
        }
        for (int j = 0; j < 8; ++j) s += acc[j][0] + acc[j][9];
    } else {
        f32x4 acc[32];
        for (int j = 0; j < 32; ++j) for (int e = 0; e < 4; ++e) acc[j][e] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int m = 0; m < 96; ++m) acc[m & 31] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[m & 31], 0, 0, 0);
        }
        for (int j = 0; j < 32; ++j) s += acc[j][0] + acc[j][3];
    }
    const long long t1 = (long long)__builtin_readcyclecounter();
    if (s == 123.456f) sink[0] = s;
    if (lane == 0) cyc[blockIdx.x * 8 + (tid >> 6)] = t1 - t0;
}

template <int SHAPE>
void run(float* sink, long long* cyc, int threads) {
    const int blocks = 256, iters = 4000;
    const double flop_per_iter_wave = 48.0 * 32768.0;                       // both loops do the same flops per iteration
    hipLaunchKernelGGL((k<SHAPE>), dim3(blocks), dim3(threads), 0, 0, sink, cyc, iters / 4, 0.01f);
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((k<SHAPE>), dim3(blocks), dim3(threads), 0, 0, sink, cyc, iters, 0.01f);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> h(blocks * 8);
    hipMemcpy(h.data(), cyc, sizeof(long long) * blocks * 8, hipMemcpyDeviceToHost);
    double c = 0;
    for (int b = 0; b < blocks; ++b) c += h[b * 8];
    c /= blocks;
    const int waves = threads / 64;
    const double tflops = flop_per_iter_wave * iters * waves * blocks / (ms * 1e-3) / 1e12;
    printf("%s  %d waves/SIMD: %7.1f cycles per iteration of wave 0 (%5.1f per 32768 flop), kernel %8.1f us, clock %.2f GHz, %7.1f TFLOP/s\n",
           SHAPE == 32 ? "32x32x16" : "16x16x32", waves / 4, c / iters, c / iters / 48.0, ms * 1e3, c / (ms * 1e6), tflops);
}

int main() {
    float* sink; long long* cyc;
    hipMalloc(&sink, 4096);
    hipMalloc(&cyc, sizeof(long long) * 256 * 8);
    for (int rep = 0; rep < 2; ++rep) {
        run<32>(sink, cyc, 256);
        run<16>(sink, cyc, 256);
        run<32>(sink, cyc, 512);
        run<16>(sink, cyc, 512);
    }
    return 0;
}
