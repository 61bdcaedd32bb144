// Micro-benchmark: HBM write / read / copy rate of a 2 GiB (beyond the 256 MiB Infinity Cache) buffer as a function of the
// cache-policy bits of the buffer instructions (gfx950: aux bit 0 = sc0, bit 1 = nt, bit 4 = sc1).
//   hipcc --offload-arch=gfx950 -O3 scripts/cpol_bw.hip -o build/cpol_bw
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned int u4 __attribute__((ext_vector_type(4)));
typedef __amdgpu_buffer_rsrc_t rsrc_t;

__device__ __forceinline__ rsrc_t make_rsrc(const void* p) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, 1 << 20, 0x00020000);
}

// every workgroup streams its own contiguous 1 MiB pieces (256 threads x 16 B x 256 iterations), pieces dealt round-robin
template <int POL>
__global__ void wr(char* p, size_t pieces, float s) {
    for (size_t pc = blockIdx.x; pc < pieces; pc += gridDim.x) {
        char* base = p + (pc << 20);
        const rsrc_t r = make_rsrc(base);
        u4 v = {(unsigned)s, (unsigned)pc, threadIdx.x, 7u};
#pragma unroll 8
        for (int i = 0; i < 256; ++i) __builtin_amdgcn_raw_buffer_store_b128(v, r, (int)(threadIdx.x * 16 + i * 4096), 0, POL);
    }
}
template <int POL>
__global__ void rd(const char* p, size_t pieces, unsigned* out) {
    unsigned a = 0;
    for (size_t pc = blockIdx.x; pc < pieces; pc += gridDim.x) {
        const rsrc_t r = make_rsrc(p + (pc << 20));
#pragma unroll 8
        for (int i = 0; i < 256; ++i) {
            const u4 v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)(threadIdx.x * 16 + i * 4096), 0, POL);
            a += v.x ^ v.y ^ v.z ^ v.w;
        }
    }
    if (a == 0x12345u) out[0] = a;
}
template <int PL, int PS>
__global__ void cp(const char* p, char* q, size_t pieces) {
    for (size_t pc = blockIdx.x; pc < pieces; pc += gridDim.x) {
        const rsrc_t r = make_rsrc(p + (pc << 20)), w = make_rsrc(q + (pc << 20));
#pragma unroll 8
        for (int i = 0; i < 256; ++i) {
            const u4 v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)(threadIdx.x * 16 + i * 4096), 0, PL);
            __builtin_amdgcn_raw_buffer_store_b128(v, w, (int)(threadIdx.x * 16 + i * 4096), 0, PS);
        }
    }
}

static hipEvent_t e0, e1;
template <class F> static double rate(F f, double bytes) {
    const int it = 10; float ms;
    f(); hipDeviceSynchronize();
    hipEventRecord(e0); for (int i = 0; i < it; ++i) f(); hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
    return bytes * it / ms / 1e6;
}

template <int POL> static void row(char* p, char* q, unsigned* out, size_t bytes, int grid) {
    const size_t pieces = bytes >> 20;
    const double w = rate([&] { wr<POL><<<grid, 256>>>(p, pieces, 1.f); }, (double)bytes);
    const double r = rate([&] { rd<POL><<<grid, 256>>>(p, pieces, out); }, (double)bytes);
    const double c = rate([&] { cp<POL, POL><<<grid, 256>>>(p, q, pieces); }, 2.0 * bytes);
    const double c0 = rate([&] { cp<0, POL><<<grid, 256>>>(p, q, pieces); }, 2.0 * bytes);
    const double c1 = rate([&] { cp<POL, 0><<<grid, 256>>>(p, q, pieces); }, 2.0 * bytes);
    printf("%6d %6d %12.0f %12.0f %12.0f %14.0f %14.0f\n", POL, grid, w, r, c, c0, c1);
}

int main() {
    unsigned* out; hipMalloc(&out, 64);
    hipEventCreate(&e0); hipEventCreate(&e1);
    const size_t bytes = (size_t)2 << 30;
    char *p, *q; hipMalloc(&p, bytes); hipMalloc(&q, bytes);
    printf("2 GiB buffers; GB/s.  pol: bit0 sc0, bit1 nt, bit4 sc1\n%6s %6s %12s %12s %12s %14s %14s\n", "pol", "grid", "write", "read", "copy(pol,pol)",
           "copy(0,pol)", "copy(pol,0)");
    for (int grid : {1024, 2048, 4096}) {
        row<0>(p, q, out, bytes, grid);
        row<1>(p, q, out, bytes, grid);
        row<2>(p, q, out, bytes, grid);
        row<3>(p, q, out, bytes, grid);
        row<16>(p, q, out, bytes, grid);
        row<17>(p, q, out, bytes, grid);
        row<18>(p, q, out, bytes, grid);
        row<19>(p, q, out, bytes, grid);
    }
    return 0;
}
