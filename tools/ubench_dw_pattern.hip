// How much of the HBM copy rate does the depthwise kernels' ACCESS PATTERN leave?  (gfx950)
//   hipcc -O3 --offload-arch=gfx950 tools/ubench_dw_pattern.hip -o /tmp/ubdw && /tmp/ubdw
// The middle-flow depthwise launch moves 100 x 32 x 32 pixels of 736 floats in and the same out (599 MB) in ~125 us = 4.8 TB/s;
// torch's linear copy of the same bytes reaches ~6.3 TB/s.  Every kernel here does one 16-byte load and one 16-byte
// (non-temporal) store per thread and output row, nothing else, over the same tensor:
//   linear    thread i copies quad i, i + stride, ...                                          (the ceiling)
//   dw        the depthwise mapping: workgroup = 16 columns x 64 channels, 16-row strips, grid (2 x 12, 2, 100)
//   dw3       the same with the three column taps loaded (left / centre / right; two of them L1 hits)
//   pixel     workgroup = 4 columns x all 736 channels (768 threads), 16-row strips: whole 2944-byte pixels contiguous
//   dw_xcd    the dw mapping with the workgroup index remapped so that the 12 channel blocks x 2 column tiles of a strip run on
//             ONE XCD (their 256-byte pieces of a pixel then meet in one L2)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));
constexpr int B = 100, H = 32, W = 32, C = 728, LD = 736, ROWS = 16;

__global__ __launch_bounds__(256) void k_linear(const f4* __restrict__ x, f4* __restrict__ y, long long n) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) __builtin_nontemporal_store(x[i], y + i);
}

template <int TAPS, bool XCD>
__global__ __launch_bounds__(256) void k_dw(const float* __restrict__ x, float* __restrict__ y) {
    int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
    if (XCD) {      // flat id -> (xcd, slot): consecutive slots of one XCD walk the 24 x-blocks of a strip
        const int flat = (bz * gridDim.y + by) * gridDim.x + bx, total = gridDim.x * gridDim.y * gridDim.z;
        const int xcd = flat & 7, slot = flat >> 3, per = total >> 3;
        const int id = xcd * per + slot;                       // total is a multiple of 8 here
        bx = id % gridDim.x; by = (id / gridDim.x) % gridDim.y; bz = id / (gridDim.x * gridDim.y);
    }
    const int c4 = threadIdx.x & 15, col = threadIdx.x >> 4, tx = bx % 2, cbk = bx / 2;
    const int ch = (cbk * 16 + c4) * 4, ox = tx * 16 + col, oy0 = by * ROWS;
    if (ch >= C) return;
    const float* xin = x + (long long)bz * H * W * LD + ch;
    float* yo = y + ((long long)bz * H * W + ox) * LD + ch;
    const int xl = max(ox - 1, 0), xr = min(ox + 1, W - 1);
#pragma unroll 4
    for (int r = 0; r < ROWS; ++r) {
        const float* row = xin + (long long)(oy0 + r) * W * LD;
        f4 v = *reinterpret_cast<const f4*>(row + (long long)ox * LD);
        if (TAPS == 3) {
            v += *reinterpret_cast<const f4*>(row + (long long)xl * LD);
            v += *reinterpret_cast<const f4*>(row + (long long)xr * LD);
        }
        __builtin_nontemporal_store(v, reinterpret_cast<f4*>(yo + (long long)(oy0 + r) * W * LD));
    }
}

__global__ __launch_bounds__(768) void k_pixel(const float* __restrict__ x, float* __restrict__ y) {
    const int q = threadIdx.x % 192, col = threadIdx.x / 192;          // 184 of 192 lanes hold a channel quad
    const int ox = blockIdx.x * 4 + col, oy0 = blockIdx.y * ROWS, b = blockIdx.z;
    if (q >= LD / 4) return;
    const long long base = ((long long)b * H * W + ox) * LD + q * 4;
#pragma unroll 4
    for (int r = 0; r < ROWS; ++r) {
        const long long o = base + (long long)(oy0 + r) * W * LD;
        __builtin_nontemporal_store(*reinterpret_cast<const f4*>(x + o), reinterpret_cast<f4*>(y + o));
    }
}

int main() {
    const long long n = (long long)B * H * W * LD;
    float *x, *y;
    CK(hipMalloc(&x, n * 4)); CK(hipMalloc(&y, n * 4));
    CK(hipMemset(x, 0, n * 4)); CK(hipMemset(y, 0, n * 4));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const char* names[5] = {"linear", "dw (16 col x 64 ch)", "dw3 (three column taps)", "pixel (4 col x 736 ch)", "dw, XCD-grouped"};
    float best[5] = {1e9f, 1e9f, 1e9f, 1e9f, 1e9f};
    for (int rep = 0; rep < 12; ++rep)
        for (int k = 0; k < 5; ++k) {                            // interleaved: the clock drifts
            CK(hipEventRecord(e0));
            if (k == 0) hipLaunchKernelGGL(k_linear, dim3(256 * 16), dim3(256), 0, 0, (const f4*)x, (f4*)y, n / 4);
            if (k == 1) hipLaunchKernelGGL((k_dw<1, false>), dim3(24, 2, B), dim3(256), 0, 0, x, y);
            if (k == 2) hipLaunchKernelGGL((k_dw<3, false>), dim3(24, 2, B), dim3(256), 0, 0, x, y);
            if (k == 3) hipLaunchKernelGGL(k_pixel, dim3(8, 2, B), dim3(768), 0, 0, x, y);
            if (k == 4) hipLaunchKernelGGL((k_dw<3, true>), dim3(24, 2, B), dim3(256), 0, 0, x, y);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep >= 2 && ms < best[k]) best[k] = ms;
        }
    for (int k = 0; k < 5; ++k) printf("%-28s %8.1f us  %7.0f GB/s (read + written)\n", names[k], best[k] * 1e3, 2.0 * n * 4 / best[k] / 1e6);
    return 0;
}
