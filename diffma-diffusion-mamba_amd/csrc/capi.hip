// C-ABI glue: error string, version, build info.  See include/diffma_hip.h.
#include "dm_common.h"
#include <cstdarg>
#include <cstdio>

namespace dm {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
static thread_local const void* g_mix_second = nullptr;
static thread_local bool g_mix_taken = false;
const void* mix_peek() { return g_mix_second; }
void mix_announce(const void* second) { g_mix_second = second; g_mix_taken = false; }
bool mix_was_taken() { return g_mix_taken; }
void mix_take() { g_mix_second = nullptr; g_mix_taken = true; }
}  // namespace dm

extern "C" int dm_abi_version(void) { return DM_ABI_VERSION; }
extern "C" const char* dm_last_error(void) { return dm::g_err; }
extern "C" const char* dm_build_info(void) {
#define DM_STR2(x) #x
#define DM_STR(x) DM_STR2(x)
    return "libdiffma_hip gfx950 (CDNA4, wave64) built " __DATE__ " " __TIME__ " hip " DM_STR(HIP_VERSION_MAJOR) "." DM_STR(
        HIP_VERSION_MINOR) "." DM_STR(HIP_VERSION_PATCH);
}
