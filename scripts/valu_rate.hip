// fp32 VALU issue rate on gfx950: v_fma_f32 vs v_pk_fma_f32 (two fp32 lanes per instruction), independent chains.
// Decides whether packed complex arithmetic can buy anything in the FFT kernels.
//   hipcc --offload-arch=gfx950 -O3 scripts/valu_rate.hip -o build/valu_rate && ./build/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));

template <int WAVES_PER_SIMD>
__global__ void __launch_bounds__(256 * WAVES_PER_SIMD / 1) fma_scalar(float* out, int iters, float a, float b) {
    float x[16];
    for (int i = 0; i < 16; ++i) x[i] = (float)(threadIdx.x + i);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(x[i]) : "v"(a), "v"(b));
    }
    float s = 0; for (int i = 0; i < 16; ++i) s += x[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int WAVES_PER_SIMD>
__global__ void __launch_bounds__(256 * WAVES_PER_SIMD / 1) fma_packed(float* out, int iters, float a, float b) {
    f2 x[16], av = {a, a}, bv = {b, b};
    for (int i = 0; i < 16; ++i) x[i] = f2{(float)(threadIdx.x + i), (float)i};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(x[i]) : "v"(av), "v"(bv));
    }
    float s = 0; for (int i = 0; i < 16; ++i) s += x[i].x + x[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int WAVES_PER_SIMD>
__global__ void __launch_bounds__(256 * WAVES_PER_SIMD / 1) add_packed(float* out, int iters, float a, float b) {
    f2 x[16], av = {a, a};
    for (int i = 0; i < 16; ++i) x[i] = f2{(float)(threadIdx.x + i), (float)i};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) asm volatile("v_pk_add_f32 %0, %1, %0" : "+v"(x[i]) : "v"(av));
    }
    float s = 0; for (int i = 0; i < 16; ++i) s += x[i].x + x[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int WAVES_PER_SIMD>
__global__ void __launch_bounds__(256 * WAVES_PER_SIMD / 1) add_scalar(float* out, int iters, float a, float b) {
    float x[16];
    for (int i = 0; i < 16; ++i) x[i] = (float)(threadIdx.x + i);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) asm volatile("v_add_f32 %0, %1, %0" : "+v"(x[i]) : "v"(a));
    }
    float s = 0; for (int i = 0; i < 16; ++i) s += x[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <typename K>
void run(const char* name, K kern, int threads, double flop_per_instr_lane) {
    float* out; hipMalloc(&out, 256 * 8 * 1024 * 4);
    const int iters = 20000, grid = 256 * 4;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    kern<<<grid, threads>>>(out, 100, 1.0001f, 0.5f);
    hipEventRecord(e0); kern<<<grid, threads>>>(out, iters, 1.0001f, 0.5f); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double instr = (double)grid * (threads / 64) * iters * 16;            // wave-instructions
    printf("%-28s %3d thr/WG: %7.2f G wave-instr/s = %.2f cycles per instr per SIMD at 2.4 GHz, %.1f TFLOP/s\n", name, threads,
           instr / ms / 1e6, 256.0 * 4 * 2.4e9 / (instr / (ms * 1e-3)), instr * 64 * flop_per_instr_lane / ms / 1e9);
    hipFree(out);
}
int main() {
    run("v_fma_f32", fma_scalar<1>, 256, 2);
    run("v_fma_f32", fma_scalar<4>, 1024, 2);
    run("v_pk_fma_f32", fma_packed<1>, 256, 4);
    run("v_pk_fma_f32", fma_packed<4>, 1024, 4);
    run("v_add_f32", add_scalar<4>, 1024, 1);
    run("v_pk_add_f32", add_packed<4>, 1024, 2);
    return 0;
}
