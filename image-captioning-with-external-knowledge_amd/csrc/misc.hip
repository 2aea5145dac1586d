// Library / device probe.
#include <cstring>

#include "common.h"

extern "C" int ick_version(void) { return 100; }

extern "C" int ick_device_info(int* num_cu, int* wave_size, char* arch_name, int arch_name_len) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return (int)e;
    hipDeviceProp_t prop;
    e = hipGetDeviceProperties(&prop, dev);
    if (e != hipSuccess) return (int)e;
    if (num_cu) *num_cu = prop.multiProcessorCount;
    if (wave_size) *wave_size = prop.warpSize;
    if (arch_name && arch_name_len > 0) {
        std::strncpy(arch_name, prop.gcnArchName, (size_t)arch_name_len - 1);
        arch_name[arch_name_len - 1] = 0;
    }
    return ICK_OK;
}
