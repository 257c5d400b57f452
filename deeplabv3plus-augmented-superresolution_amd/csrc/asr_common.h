// Shared host/device helpers for libasr_hip.so (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include <atomic>
#include "../../include/asr_hip.h"

// ---- error plumbing -------------------------------------------------------------------
void asr_set_error(const char* fmt, ...);

#define ASR_REQUIRE(cond, ...)                      \
    do {                                            \
        if (!(cond)) {                              \
            asr_set_error(__VA_ARGS__);             \
            return ASR_ERR_INVALID_ARG;             \
        }                                           \
    } while (0)

#define ASR_UNSUPPORTED(cond, ...)                  \
    do {                                            \
        if (cond) {                                 \
            asr_set_error(__VA_ARGS__);             \
            return ASR_ERR_UNSUPPORTED;             \
        }                                           \
    } while (0)

#define ASR_HIP_CHECK(expr)                                                           \
    do {                                                                              \
        hipError_t _e = (expr);                                                       \
        if (_e != hipSuccess) {                                                       \
            asr_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e),      \
                          __FILE__, __LINE__);                                        \
            return ASR_ERR_HIP;                                                       \
        }                                                                             \
    } while (0)

#define ASR_LAUNCH_CHECK() ASR_HIP_CHECK(hipGetLastError())

// ---- per-device once-initialisation ----------------------------------------------------
// The library keeps no unsynchronised global state: a process may drive several devices from several host threads.  What is
// cached (a kernel's dynamic-LDS allowance, a device's CU count) is cached PER DEVICE in atomics; the cached calls are
// idempotent, so a race only repeats one.
struct AsrDeviceOnce {
    std::atomic<unsigned long long> mask{0};                   // bit d: done on device d (devices >= 64 are never cached)
};

static inline hipError_t asr_allow_dynamic_lds(AsrDeviceOnce& once, const void* kernel, int bytes) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    const unsigned long long bit = (dev >= 0 && dev < 64) ? (1ull << dev) : 0ull;
    if (bit && (once.mask.load(std::memory_order_acquire) & bit)) return hipSuccess;
    e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e == hipSuccess && bit) once.mask.fetch_or(bit, std::memory_order_release);
    return e;
}

// multiProcessorCount of the CURRENT device (core.cpp; cached per device), <= 0 on error
int asr_device_cu_count();

static inline hipStream_t asr_stream(asr_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

static inline int64_t asr_cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

// ---- device helpers -------------------------------------------------------------------
#ifdef __HIPCC__

// Packed-f32 instructions (v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32) are OFF for every translation unit of the library
// (csrc/build.py: NO_PK_F32) and switched back on, kernel by kernel, with this attribute.  MI355X erratum (DESIGN.md 4.5,
// tools/ubench_pk_opsel_erratum.hip, profiles/r04_hazard_matrix.txt): a packed-f32 instruction whose LOW result takes the low
// half of src0 and the HIGH half of a vector-register src1 (VOP3P op_sel = [0,1]) returns wrong values in lanes 48-63 while an
// MFMA instruction of another wave is in flight on the same SIMD.  The compiler picks op_sel by itself, so packed-f32 is only
// allowed where it pays and where its output has been checked: csrc/isa_guard.py disassembles the linked library after every
// build and rejects the form anywhere, and packed-f32 of any form outside the kernels that carry this attribute (it DID emit
// the form in the SR solver, whose two-lane runs next to the forward pass's MFMA kernels found the erratum).  Same IEEE
// operations packed or not: results are bit-identical either way.
// (An opt-IN, because the inliner only merges a callee whose target features are a subset of its caller's: helpers and
// lambdas compiled without the feature inline into a kernel that has it, not the other way round.)
#if defined(__HIP_DEVICE_COMPILE__)
#define ASR_PK_F32 __attribute__((target("packed-fp32-ops")))
#else
#define ASR_PK_F32                                             // (the host pass of hipcc does not know the gfx950 feature)
#endif

// One flat projective transform (ImageProjectiveTransformV3 parameter vector).
struct AsrTf8 {
    float a0, a1, a2, b0, b1, b2, c0, c1;
};

__device__ __forceinline__ AsrTf8 asr_load_tf(const float* __restrict__ t) {
    AsrTf8 r;
    r.a0 = t[0]; r.a1 = t[1]; r.a2 = t[2];
    r.b0 = t[3]; r.b1 = t[4]; r.b2 = t[5];
    r.c0 = t[6]; r.c1 = t[7];
#ifdef ASR_TF_IN_VGPR
    // The coefficients are wave-uniform and arrive in SGPRs; pin copies in vector registers, so that whatever packed-f32
    // arithmetic the compiler forms from them takes VECTOR operands only (packed-f32 with SGPR sources).
    asm volatile("" : "+v"(r.a0), "+v"(r.a1), "+v"(r.a2), "+v"(r.b0), "+v"(r.b1), "+v"(r.b2), "+v"(r.c0), "+v"(r.c1));
#endif
    return r;
}

// float -> int for a floor()ed coordinate; clamps so that far-out-of-range (or NaN)
// coordinates become an index every bounds check rejects.
__device__ __forceinline__ int asr_coord_to_int(float f) {
    f = fminf(fmaxf(f, -1.0e9f), 1.0e9f);
    return (f == f) ? (int)f : -1000000000;
}

// Split-f16 operand halves of an ACTIVATION: v ~= hi + lo, hi = f16(v), lo = f16(v - hi).  Saturating: |v| beyond f16's
// largest finite value gives hi = +-65504 (and the remainder clamped likewise) instead of an infinity, so an out-of-range
// activation degrades the result (finite, inexact) rather than poisoning everything downstream with inf / NaN.  In range
// the clamps are the identity.  Supported range and what the host does about layers outside it: DESIGN.md 4.1.
__device__ __forceinline__ void asr_split_f16(float v, _Float16& hi, _Float16& lo) {
    hi = (_Float16)__builtin_fminf(__builtin_fmaxf(v, -65504.0f), 65504.0f);
    lo = (_Float16)__builtin_fminf(__builtin_fmaxf(v - (float)hi, -65504.0f), 65504.0f);
}

// The same split for four values at once in a wave that has switched on the hardware's f16 overflow clamp
// (asr_enable_f16_saturation, once at kernel entry): the conversions themselves saturate at +-65504, so the two clamps per
// value go, and the packed converts take two values per instruction -- 10 vector instructions for four values instead of
// 24 (28 when the values come straight from memory: the compiler first quiets a possible signalling NaN).  It matters where
// the split shares a SIMD with MFMAs: vector instructions of co-resident waves take issue slots from the matrix pipe
// (profiles/r03_gemm_loader_valu_experiment.txt).  Same halves as asr_split_f16 for every finite input; a true infinity
// stays one (MODE.FP16_OVFL preserves INF), where asr_split_f16 gives 65504.
typedef float asr_f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 asr_f16x2 __attribute__((ext_vector_type(2)));
// ASR_DIAG_TOUCH_VGPR(n): raise the kernel's vector-register allocation to at least n + 1 by naming register n once (diagnostic
// builds only: how many waves of which kernels share a SIMD's 512 registers was one of the variables of round 4's matrix).
#define ASR_DIAG_STR2(x) #x
#define ASR_DIAG_STR(x) ASR_DIAG_STR2(x)
#define ASR_DIAG_TOUCH_VGPR(n) asm volatile("v_mov_b32 v" ASR_DIAG_STR(n) ", 0" ::: "v" ASR_DIAG_STR(n))
// Diagnostic switches (tools/build_hazard_variants.py; never defined in the product build): ASR_DIAG_NO_SETREG leaves MODE
// alone (the packed conversions then overflow to infinity), ASR_DIAG_SCALAR_SPLIT keeps the MODE write but splits value by
// value with clamps like asr_split_f16.
__device__ __forceinline__ void asr_enable_f16_saturation() {
#ifndef ASR_DIAG_NO_SETREG
    __builtin_amdgcn_s_setreg(1 | (23 << 6), 1);   // hwreg(HW_REG_MODE, 23, 1) = FP16_OVFL: an overflowing f16 result clamps to +-MAX
#endif
}
// hi01, hi23, lo01, lo23: two halves per dword, element order preserved
__device__ __forceinline__ void asr_split4_f16_saturating_mode(float v0, float v1, float v2, float v3, unsigned int& hi01,
                                                               unsigned int& hi23, unsigned int& lo01, unsigned int& lo23) {
#ifdef ASR_DIAG_SCALAR_SPLIT
    _Float16 h[4], l[4];
    asr_split_f16(v0, h[0], l[0]); asr_split_f16(v1, h[1], l[1]); asr_split_f16(v2, h[2], l[2]); asr_split_f16(v3, h[3], l[3]);
    const asr_f16x2 ha = {h[0], h[1]}, hb = {h[2], h[3]}, la = {l[0], l[1]}, lb = {l[2], l[3]};
#else
    const asr_f32x2 a = {v0, v1}, b = {v2, v3};
    const asr_f16x2 ha = __builtin_convertvector(a, asr_f16x2), hb = __builtin_convertvector(b, asr_f16x2);
    const asr_f32x2 ra = a - __builtin_convertvector(ha, asr_f32x2), rb = b - __builtin_convertvector(hb, asr_f32x2);
    const asr_f16x2 la = __builtin_convertvector(ra, asr_f16x2), lb = __builtin_convertvector(rb, asr_f16x2);
#endif
    hi01 = __builtin_bit_cast(unsigned int, ha); hi23 = __builtin_bit_cast(unsigned int, hb);
    lo01 = __builtin_bit_cast(unsigned int, la); lo23 = __builtin_bit_cast(unsigned int, lb);
}

// wave64 sum via DPP-free shuffles.
__device__ __forceinline__ float asr_wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}
__device__ __forceinline__ double asr_wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}
__device__ __forceinline__ float asr_wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_down(v, o, 64));
    return v;
}
__device__ __forceinline__ float asr_wave_min(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_down(v, o, 64));
    return v;
}

#endif  // __HIPCC__
