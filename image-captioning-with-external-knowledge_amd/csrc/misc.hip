// Library / device probe.
#include <cstring>

#include "common.h"

extern "C" int ick_version(void) { return 100; }

// Deterministic mode (ICK_DETERMINISTIC=1 in the environment, or ick_set_deterministic): every gradient reduction that
// otherwise lets float atomics decide the order of its terms runs in a fixed order instead (backward.hip); the Python
// side then keeps every GEMM unsplit and the step on one stream.  Slower; bit-reproducible.
namespace ick { int g_deterministic = -1; }
extern "C" int ick_set_deterministic(int on) { ick::g_deterministic = on ? 1 : 0; return ICK_OK; }
extern "C" int ick_get_deterministic(void) { return ick::deterministic() ? 1 : 0; }

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

// Several device-to-device copies in one launch (the inputs of a captured graph are refreshed every step: five
// separate ~5 us copy kernels otherwise).  16-byte chunks where both pointers and the size allow, bytes otherwise.
namespace {
constexpr int kCopyMax = 8;
struct CopyBatch {
    const unsigned char* src[kCopyMax];
    unsigned char* dst[kCopyMax];
    long long bytes[kCopyMax];
    int n;
};
__global__ __launch_bounds__(256) void copy_batch_kernel(CopyBatch cb) {
    const int which = blockIdx.y;
    if (which >= cb.n) return;
    const unsigned char* s = cb.src[which];
    unsigned char* d = cb.dst[which];
    const long long nb = cb.bytes[which];
    const long long stride = (long long)gridDim.x * 256;
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if ((((unsigned long long)s | (unsigned long long)d | (unsigned long long)nb) & 15) == 0) {
        const uint4* s4 = reinterpret_cast<const uint4*>(s);
        uint4* d4 = reinterpret_cast<uint4*>(d);
        for (long long i = t; i < nb / 16; i += stride) d4[i] = s4[i];
    } else {
        for (long long i = t; i < nb; i += stride) d[i] = s[i];
    }
}
}  // namespace

extern "C" int ick_copy_batch(const void* const* src, void* const* dst, const long long* bytes, int n, void* stream) {
    if (!src || !dst || !bytes || n <= 0 || n > kCopyMax) return ICK_EINVAL;
    CopyBatch cb;
    long long mx = 0;
    for (int i = 0; i < n; ++i) {
        if (!src[i] || !dst[i] || bytes[i] < 0) return ICK_EINVAL;
        cb.src[i] = static_cast<const unsigned char*>(src[i]);
        cb.dst[i] = static_cast<unsigned char*>(dst[i]);
        cb.bytes[i] = bytes[i];
        mx = bytes[i] > mx ? bytes[i] : mx;
    }
    cb.n = n;
    const long long per_block = 256LL * 16 * 4;     // four 16-byte chunks per thread at most
    long long gx = (mx + per_block - 1) / per_block;
    gx = gx < 1 ? 1 : (gx > 2048 ? 2048 : gx);
    hipLaunchKernelGGL(copy_batch_kernel, dim3((unsigned)gx, (unsigned)n), dim3(256), 0, (hipStream_t)stream, cb);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? ICK_OK : (int)e;
}
