// Error reporting and library identity for libasr_hip.so.
#include "asr_common.h"
#include <string.h>

static thread_local char g_asr_error[512] = "";

void asr_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_asr_error, sizeof(g_asr_error), fmt, ap);
    va_end(ap);
}

extern "C" const char* asr_last_error(void) { return g_asr_error; }

extern "C" int asr_abi_version(void) { return ASR_ABI_VERSION; }

extern "C" const char* asr_target_arch(void) { return "gfx950"; }

// multiProcessorCount of the current device, cached per device (atomics: several host threads / devices may call)
int asr_device_cu_count() {
    static std::atomic<int> cached[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return -1;
    if (dev >= 0 && dev < 64) {
        const int c = cached[dev].load(std::memory_order_acquire);
        if (c > 0) return c;
    }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return -1;
    const int cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    if (dev >= 0 && dev < 64) cached[dev].store(cus, std::memory_order_release);
    return cus;
}
