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

// Device-side time stamp (100 MHz constant clock): a one-lane kernel that drops wall_clock64() into *out when the
// stream reaches it.  Diagnostic only (ICK_TIMESTAMPS=1 in training.py): rocprofv3 slows hipGraph launches enough to
// distort how the branches of a captured graph overlap, these stamps do not.
namespace {
__global__ void timestamp_kernel(unsigned long long* out) { *out = wall_clock64(); }
}  // namespace

extern "C" int ick_timestamp(unsigned long long* out, void* stream) {
    if (!out) return ICK_EINVAL;
    hipLaunchKernelGGL(timestamp_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, out);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? ICK_OK : (int)e;
}
