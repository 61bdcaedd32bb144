// Micro-benchmark for the planned per-XCD persistent pipeline:
//   1. which XCD does workgroup i run on?  (expected: i % 8)
//   2. how fast can the 64 workgroups of one XCD exchange a W-like slab -- written as 128-byte column pieces, read back
//      as 8 KB rows after an intra-XCD barrier -- when the slab fits the XCD's 4 MB L2, the Infinity Cache, or neither?
// The barrier is hand-rolled (no agent-scope fence: that would write back / invalidate the L2): stores complete
// (s_waitcnt), one relaxed atomic per workgroup on a counter that lives in the L2, the consumers invalidate their L1.
// Spins are bounded: a grid that is not co-resident reports an error instead of hanging the device.
//   hipcc --offload-arch=gfx950 -O3 scripts/xcd_exchange.hip -o build/xcd_exchange && ./build/xcd_exchange
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define NXCD 8
#define WG_PER_XCD 64
#define THREADS 256
#define SPIN_LIMIT 2000000

__global__ void probe_xcc(int* out) {
    if (threadIdx.x == 0) out[blockIdx.x] = (int)__builtin_amdgcn_s_getreg(20 | (3 << 11));   // HW_REG_XCC_ID[3:0]
}

#ifndef BAR_SCOPE
#define BAR_SCOPE __HIP_MEMORY_SCOPE_AGENT
#endif
template <int SLEEP>
__device__ __forceinline__ bool xcd_barrier(unsigned* counter, unsigned target, int* err) {
    __builtin_amdgcn_s_waitcnt(0);                       // this lane's stores have reached the L2
    __syncthreads();
    bool ok = true;
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, BAR_SCOPE);
        unsigned spins = 0;
        while (__hip_atomic_fetch_add(counter, 0u, __ATOMIC_RELAXED, BAR_SCOPE) < target) {   // RMW: always served by the L2
            if (++spins > SPIN_LIMIT) { ok = false; atomicExch(err, 1); break; }
            if (SLEEP) __builtin_amdgcn_s_sleep(SLEEP);
        }
    }
    __syncthreads();
    asm volatile("buffer_inv sc1" ::: "memory");         // drop stale L1 lines; the L2 keeps its data
    return ok;
}

// barrier latency alone: `iters` empty phases
template <int SLEEP>
__global__ void __launch_bounds__(THREADS) barrier_only(int iters, unsigned* counters, int* err) {
    unsigned* counter = counters + (blockIdx.x % NXCD) * 32;
    for (int it = 1; it <= iters; ++it)
        if (!xcd_barrier<SLEEP>(counter, WG_PER_XCD * it, err)) return;
}

// slab of one XCD: rows x 512 float4 (8 KB rows).  Workgroup r owns float4 columns [8 r, 8 r + 8) (128 bytes of a row).
__global__ void __launch_bounds__(THREADS) exchange(float4* slabs, size_t slab_f4, int rows, int iters, unsigned* counters,
                                                    int* err, float* sink, int use_swizzle) {
    const int bx = blockIdx.x;
    const int xcd = use_swizzle ? bx % NXCD : bx / WG_PER_XCD;          // "wrong" mapping for comparison
    const int r = use_swizzle ? bx / NXCD : bx % WG_PER_XCD;
    float4* slab = slabs + (size_t)xcd * slab_f4;
    unsigned* counter = counters + xcd * 32;                             // one counter per 128-byte line
    const int t = threadIdx.x;
    float acc = 0.f;
    unsigned phase = 0;
    for (int it = 0; it < iters; ++it) {
        // column phase: 8 lanes cover the 128-byte piece, 32 rows per pass
        for (int row = t >> 3; row < rows; row += THREADS / 8) {
            const float v = (float)(it + row + r);
            slab[(size_t)row * 512 + 8 * r + (t & 7)] = make_float4(v, v, v, v);
        }
        if (!xcd_barrier<1>(counter, WG_PER_XCD * ++phase, err)) return;
        // row phase: rows r, r + 64, ... read whole (512 float4 = 2 per thread)
        for (int row = r; row < rows; row += WG_PER_XCD) {
            const float4 a = slab[(size_t)row * 512 + t], b = slab[(size_t)row * 512 + 256 + t];
            acc += a.x + b.x - 2.f * (float)(it + row) - (float)(t >> 3) - (float)((256 + t) >> 3);
        }
        if (!xcd_barrier<1>(counter, WG_PER_XCD * ++phase, err)) return;
    }
    if (acc != 0.f) atomicExch(err, 2);                                  // every value read back must be the one written
    if (acc == 1234.5f) sink[0] = acc;
}

int main() {
    const int grid = NXCD * WG_PER_XCD;
    int* d_xcc; hipMalloc(&d_xcc, grid * sizeof(int));
    probe_xcc<<<grid, 64>>>(d_xcc);
    std::vector<int> xcc(grid);
    hipMemcpy(xcc.data(), d_xcc, grid * sizeof(int), hipMemcpyDeviceToHost);
    int match = 0;
    for (int i = 0; i < grid; ++i) match += (xcc[i] == i % NXCD);
    printf("XCC_ID of workgroup i == i %% 8 for %d of %d workgroups (first 16:", match, grid);
    for (int i = 0; i < 16; ++i) printf(" %d", xcc[i]);
    printf(")\n");

    unsigned* counters; hipMalloc(&counters, NXCD * 32 * sizeof(unsigned));
    int* err; hipMalloc(&err, sizeof(int));
    float* sink; hipMalloc(&sink, 64);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    {
        const int iters = 2000;
        float ms;
        hipMemset(counters, 0, NXCD * 32 * sizeof(unsigned)); hipMemset(err, 0, sizeof(int));
        hipEventRecord(e0); barrier_only<0><<<grid, THREADS>>>(iters, counters, err); hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1); printf("intra-XCD barrier alone (64 workgroups, busy spin): %.2f us\n", ms * 1e3 / iters);
        hipMemset(counters, 0, NXCD * 32 * sizeof(unsigned));
        hipEventRecord(e0); barrier_only<1><<<grid, THREADS>>>(iters, counters, err); hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1); printf("intra-XCD barrier alone (s_sleep 1 in the spin):    %.2f us\n", ms * 1e3 / iters);
        hipMemset(counters, 0, NXCD * 32 * sizeof(unsigned));
        hipEventRecord(e0); barrier_only<8><<<grid, THREADS>>>(iters, counters, err); hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1); printf("intra-XCD barrier alone (s_sleep 8 in the spin):    %.2f us\n", ms * 1e3 / iters);
    }
    const int row_counts[] = {128, 256, 384, 512, 1024, 2048, 16384};    // x 8 KB: 1, 2, 3, 4, 8, 16, 128 MB per XCD
    printf("%12s %10s %14s %14s\n", "MB per XCD", "mapping", "exchange GB/s", "us per phase");
    for (int rows : row_counts) {
        const size_t slab_f4 = (size_t)rows * 512;
        float4* slabs;
        if (hipMalloc(&slabs, slab_f4 * 16 * NXCD) != hipSuccess) { printf("alloc failed\n"); break; }
        hipMemset(slabs, 0, slab_f4 * 16 * NXCD);
        for (int swz = 1; swz >= 0; --swz) {
            const int iters = rows <= 1024 ? 200 : (rows <= 2048 ? 50 : 8);
            float best = 1e30f;
            int herr = 0;
            for (int rep = 0; rep < 3; ++rep) {
                hipMemset(counters, 0, NXCD * 32 * sizeof(unsigned));
                hipMemset(err, 0, sizeof(int));
                hipEventRecord(e0);
                exchange<<<grid, THREADS>>>(slabs, slab_f4, rows, iters, counters, err, sink, swz);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                hipMemcpy(&herr, err, sizeof(int), hipMemcpyDeviceToHost);
                if (herr) break;
                if (ms < best) best = ms;
            }
            if (herr) { printf("%12.1f %10s   ERROR %d (1 = barrier timed out, 2 = stale data)\n", rows * 8.0 / 1024, swz ? "i%8" : "i/64", herr); continue; }
            const double bytes = (double)slab_f4 * 16 * NXCD * 2 * iters;                   // written once + read once per iteration
            printf("%12.1f %10s %14.0f %14.2f\n", rows * 8.0 / 1024, swz ? "i%8" : "i/64", bytes / best / 1e6,
                   best * 1e3 / (2.0 * iters));
        }
        hipFree(slabs);
    }
    return 0;
}
