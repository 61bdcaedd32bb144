// Micro-benchmark: write-then-read bandwidth of a buffer as a function of its size (is the 256 MiB Infinity
// Cache a write-back home for intermediates?).  hipcc --offload-arch=gfx950 -O3 scripts/mall_bw.hip -o build/mall_bw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void wr(float4* p, size_t n, float s) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x, st = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += st) p[i] = make_float4(s, s + 1, s + 2, (float)i);
}
__global__ void rd(const float4* p, size_t n, float* out) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x, st = (size_t)gridDim.x * blockDim.x;
    float a = 0;
    for (; i < n; i += st) { float4 v = p[i]; a += v.x + v.y + v.z + v.w; }
    if (a == 1234.5f) out[0] = a;
}
__global__ void cp(const float4* p, float4* q, size_t n) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x, st = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += st) q[i] = p[i];
}
int main() {
    float* out; hipMalloc(&out, 64);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    size_t sizes[] = {8u << 20, 16u << 20, 32u << 20, 64u << 20, 128u << 20, 192u << 20, 256u << 20, 384u << 20, 512u << 20, 1024u << 20, 2048u << 20};
    printf("%10s %12s %12s %12s %12s\n", "MiB", "write GB/s", "read GB/s", "wr+rd GB/s", "copy GB/s");
    for (size_t bytes : sizes) {
        float4 *p, *q; hipMalloc(&p, bytes); hipMalloc(&q, bytes);
        size_t n = bytes / 16; int grid = 256 * 8, it = 20; float ms;
        wr<<<grid, 256>>>(p, n, 1.f); rd<<<grid, 256>>>(p, n, out); hipDeviceSynchronize();
        hipEventRecord(e0); for (int i = 0; i < it; ++i) wr<<<grid, 256>>>(p, n, (float)i); hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1); double w = bytes * (double)it / ms / 1e6;
        hipEventRecord(e0); for (int i = 0; i < it; ++i) rd<<<grid, 256>>>(p, n, out); hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1); double r = bytes * (double)it / ms / 1e6;
        hipEventRecord(e0); for (int i = 0; i < it; ++i) { wr<<<grid, 256>>>(p, n, (float)i); rd<<<grid, 256>>>(p, n, out); } hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1); double wrd = 2.0 * bytes * it / ms / 1e6;
        hipEventRecord(e0); for (int i = 0; i < it; ++i) cp<<<grid, 256>>>(p, q, n); hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1); double c = 2.0 * bytes * it / ms / 1e6;
        printf("%10zu %12.0f %12.0f %12.0f %12.0f\n", bytes >> 20, w, r, wrd, c);
        hipFree(p); hipFree(q);
    }
    return 0;
}
