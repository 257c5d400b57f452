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
