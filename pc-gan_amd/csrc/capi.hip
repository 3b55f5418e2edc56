// Status / diagnostics part of the C-ABI (include/pcgan_hip.h).
#include "common.h"
#include <stdarg.h>

namespace pcgan {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
}  // namespace pcgan

extern "C" const char* pcgan_last_error(void) { return pcgan::g_err; }
extern "C" int pcgan_version(void) { return 100; }

extern "C" int pcgan_device_info(int* cu_count, char* arch, int arch_len) {
    hipDeviceProp_t prop;
    hipError_t e = hipGetDeviceProperties(&prop, 0);
    if (e != hipSuccess) {
        pcgan::set_error("hipGetDeviceProperties failed: %s", hipGetErrorString(e));
        return 1;
    }
    if (cu_count) *cu_count = prop.multiProcessorCount;
    if (arch && arch_len > 0) {
        strncpy(arch, prop.gcnArchName, arch_len - 1);
        arch[arch_len - 1] = 0;
    }
    return 0;
}
