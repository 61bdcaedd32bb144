// Micro-benchmark: does separating HBM reads from HBM writes in time (chip-wide) beat the mixed copy rate (~5.2 TB/s)?
// 256 persistent workgroups of 1024 threads (one per CU); each iteration a workgroup loads NB x 16 B per thread
// (NB = 16: 256 KB per workgroup, 64 MB chip-wide) and then stores them.
//   SYNC 0: no coordination;  SYNC 1: a chip-wide epoch barrier (atomic counter) between the load and the store phase.
//   hipcc --offload-arch=gfx950 -O3 scripts/phase_bw.hip -o build/phase_bw
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned int u4 __attribute__((ext_vector_type(4)));
typedef __amdgpu_buffer_rsrc_t rsrc_t;
__device__ __forceinline__ rsrc_t make_rsrc(const void* p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, 0x00020000);
}
__device__ __forceinline__ void grid_sync(unsigned* ctr, unsigned target) {
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        int spin = 0;
        while (__hip_atomic_load(ctr, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target && ++spin < (1 << 22)) __builtin_amdgcn_s_sleep(1);
    }
    __syncthreads();
}
template <int NB, int SYNC, int PL, int PS>
__global__ __launch_bounds__(1024) void cp(const char* p, char* q, int iters, unsigned* ctr) {
    const unsigned chunk = 1024u * 16u * NB;            // bytes per workgroup and iteration
    unsigned epoch = 0;
    for (int it = 0; it < iters; ++it) {
        const size_t off = ((size_t)it * gridDim.x + blockIdx.x) * chunk;
        const rsrc_t r = make_rsrc(p + off, chunk), w = make_rsrc(q + off, chunk);
        u4 v[NB];
#pragma unroll
        for (int i = 0; i < NB; ++i) v[i] = __builtin_amdgcn_raw_buffer_load_b128(r, (int)(threadIdx.x * 16 + i * 16384), 0, PL);
        if (SYNC) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); epoch += gridDim.x; grid_sync(ctr, epoch); }
#pragma unroll
        for (int i = 0; i < NB; ++i) __builtin_amdgcn_raw_buffer_store_b128(v[i], w, (int)(threadIdx.x * 16 + i * 16384), 0, PS);
        if (SYNC) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); epoch += gridDim.x; grid_sync(ctr, epoch); }
    }
}
static hipEvent_t e0, e1;
template <int NB, int SYNC, int PL, int PS> static void run(const char* p, char* q, size_t bytes, unsigned* ctr, int grid) {
    const size_t chunk = 1024u * 16u * NB;
    const int iters = (int)(bytes / (chunk * grid));
    float ms, best = 1e9f;
    for (int rep = 0; rep < 4; ++rep) {
        hipMemset(ctr, 0, 4);
        hipEventRecord(e0); cp<NB, SYNC, PL, PS><<<grid, 1024>>>(p, q, iters, ctr); hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1); if (rep && ms < best) best = ms;
    }
    printf("NB %2d (%4zu KB/wg) sync %d  pol(ld %2d, st %2d) grid %4d : %7.0f GB/s (%.3f ms)\n", NB, chunk >> 10, SYNC, PL, PS, grid,
           2.0 * chunk * grid * iters / best / 1e6, best);
}
int main() {
    hipEventCreate(&e0); hipEventCreate(&e1);
    const size_t bytes = (size_t)2 << 30;
    char *p, *q; unsigned* ctr; hipMalloc(&p, bytes); hipMalloc(&q, bytes); hipMalloc(&ctr, 64);
    hipMemset(p, 1, bytes); hipMemset(q, 2, bytes);
    for (int grid : {256, 512}) {
        run<4, 0, 0, 0>(p, q, bytes, ctr, grid);
        run<16, 0, 0, 0>(p, q, bytes, ctr, grid);
        run<16, 0, 2, 0>(p, q, bytes, ctr, grid);
        run<16, 0, 2, 2>(p, q, bytes, ctr, grid);
        run<28, 0, 2, 0>(p, q, bytes, ctr, grid);
    }
    run<4, 1, 0, 0>(p, q, bytes, ctr, 256);
    run<16, 1, 0, 0>(p, q, bytes, ctr, 256);
    run<16, 1, 2, 0>(p, q, bytes, ctr, 256);
    run<28, 1, 2, 0>(p, q, bytes, ctr, 256);
    run<28, 1, 0, 0>(p, q, bytes, ctr, 256);
    return 0;
}
