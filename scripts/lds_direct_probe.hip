// Probe of gfx950's global_load_lds_dwordx4 (LDS-direct 16-byte loads): where a lane's 16 bytes land (M0 base + 16 * lane) and that
// per-lane global addresses are free (permuted pieces).   hipcc --offload-arch=gfx950 -O3 scripts/lds_direct_probe.hip -o /tmp/ldsprobe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

__global__ void probe(const char* g, char* out) {
    extern __shared__ char smem[];
    const int wave = threadIdx.x >> 6;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g + (size_t)(threadIdx.x ^ 5) * 16),
                                     (__attribute__((address_space(3))) void*)(smem + wave * 1024), 16, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    reinterpret_cast<uint4*>(out)[threadIdx.x] = reinterpret_cast<uint4*>(smem)[threadIdx.x];
}

int main() {
    const int n = 256 * 16;
    std::vector<unsigned char> h(n), o(n);
    for (int i = 0; i < n; ++i) h[i] = (unsigned char)(i * 7 + (i >> 8));
    char *dg, *dout;
    hipMalloc(&dg, n); hipMalloc(&dout, n);
    hipMemcpy(dg, h.data(), n, hipMemcpyHostToDevice);
    hipMemset(dout, 0, n);
    hipLaunchKernelGGL(probe, dim3(1), dim3(256), 4096, 0, dg, dout);
    hipMemcpy(o.data(), dout, n, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int t = 0; t < 256; ++t) if (memcmp(&o[t * 16], &h[(t ^ 5) * 16], 16) != 0) ++bad;
    printf("lds-direct dwordx4: %d of 256 lanes wrong (0 = lane l's 16 bytes land at M0 + 16 l, global addresses per lane)\n", bad);
    return bad != 0;
}
