// Micro-benchmark: how fast can a CU get a 256 KB output tile out of its registers?  G workgroups (one per CU) of 8 waves;
// every wave issues 32 global_store_dwordx4 (1 KB per instruction) -- 16 rows x 64 B (the transposed-MFMA epilogue's
// pattern), 8 rows x 128 B (whole lines: two column tiles per store) or 2 rows x 512 B (the LDS-transposed epilogue's
// pattern), at a row stride of 728 floats (2912 B, not a multiple of the 128-byte line) and of 736 (what the engine
// allocates) -- then waits for them.  Round 3, G = 256: 13.6 K cycles at stride 728 against 8.7 K at 736 for the 64 B and
// the 128 B segments alike (30 B/clk per CU), 21.4 K / 12.4 K for the 512 B rows: the alignment of the rows matters, the
// segment length does not.  Prints the
// mean cycles per workgroup until issued and until drained, for G = 8 .. 256: a per-CU limit shows at every G, a shared
// (fabric / HBM) limit only at large G.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_store_burst.hip -o ubs && ./ubs
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int PATTERN>
__global__ __launch_bounds__(512) void k(float* __restrict__ y, long long* __restrict__ cyc, int ldy, int rounds) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    f32x4 v = {(float)lane, 1.f, 2.f, 3.f};
    long long t_issue = 0, t_drain = 0;
    for (int r = 0; r < rounds; ++r) {
        // the workgroup's tile: 256 rows x 256 columns at rows (blockIdx * rounds + r) * 256
        float* const tile = y + ((long long)(blockIdx.x * rounds + r) * 256) * ldy;
        const int wm = wave & 3, wn = wave >> 2;                // wave: 64 rows x 128 columns
        __builtin_amdgcn_s_barrier();
        const long long t0 = (long long)__builtin_readcyclecounter();
        if (PATTERN == 0) {                                     // 16 rows x 64 B per instruction
            const int l16 = lane & 15, q4 = lane >> 4;
#pragma unroll
            for (int ct = 0; ct < 8; ++ct)
#pragma unroll
                for (int rt = 0; rt < 4; ++rt)
                    *reinterpret_cast<f32x4*>(tile + (long long)(wm * 64 + rt * 16 + l16) * ldy + wn * 128 + ct * 16 + q4 * 4) = v;
        } else if (PATTERN == 2) {                              // 8 rows x 128 B per instruction (two column tiles per store)
            const int l8 = lane & 7, q8 = lane >> 3;
#pragma unroll
            for (int ct = 0; ct < 8; ct += 2)
#pragma unroll
                for (int rt = 0; rt < 8; ++rt)
                    *reinterpret_cast<f32x4*>(tile + (long long)(wm * 64 + rt * 8 + l8) * ldy + wn * 128 + ct * 16 + q8 * 4) = v;
        } else {                                                // 2 rows x 512 B per instruction
            const int c4 = lane & 31, r_in = lane >> 5;
#pragma unroll
            for (int q = 0; q < 32; ++q)
                *reinterpret_cast<f32x4*>(tile + (long long)(wm * 64 + q * 2 + r_in) * ldy + wn * 128 + c4 * 4) = v;
        }
        const long long t1 = (long long)__builtin_readcyclecounter();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const long long t2 = (long long)__builtin_readcyclecounter();
        t_issue += t1 - t0;
        t_drain += t2 - t0;
    }
    if (tid == 0) { cyc[blockIdx.x * 2] = t_issue / rounds; cyc[blockIdx.x * 2 + 1] = t_drain / rounds; }
}

int main() {
    const int rounds = 8;
    float* y; long long* cyc;
    hipMalloc(&y, (size_t)256 * rounds * 256 * 736 * 4 + 4096);
    hipMalloc(&cyc, sizeof(long long) * 512);
    for (int ldy : {728, 736}) {
    printf("row stride %d floats\n", ldy);
    for (int pattern = 0; pattern < 3; ++pattern)
        for (int g : {8, 64, 256}) {
            for (int rep = 0; rep < 2; ++rep) {
                if (pattern == 0) hipLaunchKernelGGL(k<0>, dim3(g), dim3(512), 0, 0, y, cyc, ldy, rounds);
                else if (pattern == 2) hipLaunchKernelGGL(k<2>, dim3(g), dim3(512), 0, 0, y, cyc, ldy, rounds);
                else hipLaunchKernelGGL(k<1>, dim3(g), dim3(512), 0, 0, y, cyc, ldy, rounds);
                hipDeviceSynchronize();
            }
            std::vector<long long> h(512);
            hipMemcpy(h.data(), cyc, sizeof(long long) * 512, hipMemcpyDeviceToHost);
            double a = 0, b = 0;
            for (int i = 0; i < g; ++i) { a += h[2 * i]; b += h[2 * i + 1]; }
            printf("%s  G=%3d workgroups: issued after %6.0f cycles, drained after %6.0f cycles  (%.1f B/clk per CU)\n",
                   pattern == 0 ? "16 rows x 64 B " : (pattern == 2 ? "8 rows x 128 B " : "2 rows x 512 B "), g, a / g, b / g, 262144.0 / (b / g));
        }
    }
    return 0;
}
