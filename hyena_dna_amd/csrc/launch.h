// launch.h -- kernel launch shim shared by the host translation units (fftconv.hip, onchip.hip): hipLaunchKernelGGL for
// the product, tests/hipemu's CPU launcher with -DHIPEMU (tests only).
#pragma once
#ifdef HIPEMU
#include <tuple>
#define HY_LAUNCH(kernel, grid, block, smem, stream, ...)                              \
    do {                                                                                \
        auto _args = std::make_tuple(__VA_ARGS__);                                      \
        hipemu::launch(grid, block, smem, [&] { std::apply(kernel, _args); });          \
    } while (0)
static inline int hy_launch_error() { return 0; }
#else
#define HY_LAUNCH(kernel, grid, block, smem, stream, ...) \
    hipLaunchKernelGGL(kernel, grid, block, smem, (hipStream_t)(stream), __VA_ARGS__)
static inline int hy_launch_error() { return hipGetLastError() != hipSuccess; }
#endif

// Kernels that want more than the 64 KiB of LDS a launch may ask for by default (per device: the attribute is re-applied
// whenever the calling thread's current device changes).
template <typename K>
static inline void hy_allow_lds(K kernel, size_t bytes, int* device_done) {
#ifndef HIPEMU
    if (bytes <= 65536) return;
    int dev = -1;
    (void)hipGetDevice(&dev);
    if (*device_done == dev) return;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    *device_done = dev;
#else
    (void)kernel; (void)bytes; (void)device_done;
#endif
}
