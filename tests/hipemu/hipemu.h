// hipemu.h -- TEST INFRASTRUCTURE ONLY.  Never shipped, never loaded by hyena_dna_amd.
//
// A tiny x86-64 host emulation of the HIP execution model, just enough to run the kernels of
// hyena_dna_amd/csrc/*.h on the CPU inside `pytest -m "not gpu"` so that index math, LDS
// exchange patterns, wave shuffles and barriers are checked in the build container (which has
// no GPU) before a kernel ever reaches an MI355X.  One fibre per HIP thread, one block at a
// time per OS thread (blocks are spread over OpenMP threads), cooperative switching at
// __syncthreads() / wave shuffles.  It is NOT a fallback for the product: the product library
// (libhyena_fftconv.so, built by hipcc for gfx950) contains none of this, and the Python
// package refuses to run without the real HIP library.
#pragma once
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <functional>
#include <vector>

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
typedef void* hipStream_t;
typedef int hipError_t;
#define hipSuccess 0

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)

namespace hipemu {

enum { WAVE = 64 };

struct Fiber {
    void* sp;        // saved stack pointer
    char* stack;
    int done;
    int yield_kind;  // 0 none, 1 barrier, 2 shuffle
};

struct State {
    dim3 tid, bid, bdim, gdim;
    char* smem;
    std::vector<Fiber> fibers;
    void* sched_sp;
    int cur;
    const std::function<void()>* body;
    std::vector<uint32_t> xchg;   // shuffle exchange slots, one per thread
    std::vector<uint32_t> xchg2;  // second operand slot (MFMA B operand)
};

extern thread_local State S;

extern "C" void hipemu_switch(void** save_sp, void* load_sp);

inline void yield(int kind) {
    Fiber& f = S.fibers[S.cur];
    f.yield_kind = kind;
    hipemu_switch(&f.sp, S.sched_sp);
}

inline void syncthreads() { yield(1); }

inline uint32_t shfl_u32(uint32_t v, int src_lane) {
    int t = S.cur;
    S.xchg[t] = v;
    yield(2);
    int wave_base = (t / WAVE) * WAVE;
    uint32_t r = S.xchg[wave_base + (src_lane & (WAVE - 1))];
    yield(2);
    return r;
}

// v_mfma_f32_32x32x2_f32: D = A(32x2) * B(2x32) + C, one wave.  Lane l supplies A[l & 31][l >> 5] and
// B[l >> 5][l & 31]; it owns C/D[(r & 3) + 8 * (r >> 2) + 4 * (l >> 5)][l & 31] for r in [0, 16).
typedef float floatx16 __attribute__((vector_size(64)));
inline floatx16 mfma_f32_32x32x2f32(float a, float b, floatx16 c) {
    int t = S.cur, lane = t & (WAVE - 1), base = t - lane;
    memcpy(&S.xchg[t], &a, 4);
    memcpy(&S.xchg2[t], &b, 4);
    yield(2);
    for (int r = 0; r < 16; ++r) {
        int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5), col = lane & 31;
        float acc = c[r];
        for (int k = 0; k < 2; ++k) {
            float av, bv;
            memcpy(&av, &S.xchg[base + row + 32 * k], 4);
            memcpy(&bv, &S.xchg2[base + col + 32 * k], 4);
            acc = fmaf(av, bv, acc);
        }
        c[r] = acc;
    }
    yield(2);
    return c;
}

void launch(dim3 grid, dim3 block, size_t smem_bytes, const std::function<void()>& body);

}  // namespace hipemu

#define threadIdx (hipemu::S.tid)
#define blockIdx (hipemu::S.bid)
#define blockDim (hipemu::S.bdim)
#define gridDim (hipemu::S.gdim)
#define __syncthreads() hipemu::syncthreads()
