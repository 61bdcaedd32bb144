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
    std::vector<uint32_t> xq, xq2;  // four words per thread: the 16-byte A / B operands of the 16-bit MFMAs
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

// v_mfma_f32_32x32x16_{bf16,f16} (gfx950): D = A(32x16) * B(16x32) + C, one wave.  Lane l supplies the eight 16-bit values
// A[l & 31][8 (l >> 5) + j] and B[8 (l >> 5) + j][l & 31], j = 0..7 (16 bytes each), and owns the same C/D elements as above.
// Products are exact in fp32; the sum is taken as a k-ordered fmaf chain here -- the hardware's internal order may differ, so
// GPU results are compared with a tolerance, not bitwise.
struct u32x4 { uint32_t w[4]; };
template <bool BF16>
inline floatx16 mfma_f32_32x32x16_h(u32x4 a, u32x4 b, floatx16 c) {
    int t = S.cur, lane = t & (WAVE - 1), base = t - lane;
    memcpy(&S.xq[4 * t], a.w, 16);
    memcpy(&S.xq2[4 * t], b.w, 16);
    yield(2);
    auto dec = [](uint16_t h) -> float {
        if (BF16) { uint32_t u = (uint32_t)h << 16; float f; memcpy(&f, &u, 4); return f; }
        uint32_t sign = (uint32_t)(h >> 15) << 31, e = (h >> 10) & 31, m = h & 1023, u;
        if (e == 0) {
            if (m == 0) u = sign;
            else { int sh = 0; while (!(m & 1024)) { m <<= 1; ++sh; } u = sign | ((uint32_t)(113 - sh) << 23) | ((m & 1023) << 13); }
        } else if (e == 31) u = sign | 0x7F800000u | (m << 13);
        else u = sign | ((e + 112) << 23) | (m << 13);
        float f; memcpy(&f, &u, 4); return f;
    };
    for (int r = 0; r < 16; ++r) {
        int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5), col = lane & 31;
        float acc = c[r];
        for (int k = 0; k < 16; ++k) {
            const uint16_t* ap = reinterpret_cast<const uint16_t*>(&S.xq[4 * (base + row + 32 * (k >> 3))]);
            const uint16_t* bp = reinterpret_cast<const uint16_t*>(&S.xq2[4 * (base + col + 32 * (k >> 3))]);
            acc = fmaf(dec(ap[k & 7]), dec(bp[k & 7]), acc);
        }
        c[r] = acc;
    }
    yield(2);
    return c;
}

// v_mfma_f32_16x16x32_{bf16,f16} (gfx950): D = A(16x32) * B(32x16) + C.  Lane l supplies A[l & 15][8 (l >> 4) + j] and
// B[8 (l >> 4) + j][l & 15], j = 0..7, and owns C/D[4 (l >> 4) + r][l & 15] for r in [0, 4).
typedef float floatx4 __attribute__((vector_size(16)));
template <bool BF16>
inline floatx4 mfma_f32_16x16x32_h(u32x4 a, u32x4 b, floatx4 c) {
    int t = S.cur, lane = t & (WAVE - 1), base = t - lane;
    memcpy(&S.xq[4 * t], a.w, 16);
    memcpy(&S.xq2[4 * t], b.w, 16);
    yield(2);
    auto dec = [](uint16_t h) -> float {
        if (BF16) { uint32_t u = (uint32_t)h << 16; float f; memcpy(&f, &u, 4); return f; }
        uint32_t sign = (uint32_t)(h >> 15) << 31, e = (h >> 10) & 31, m = h & 1023, u;
        if (e == 0) {
            if (m == 0) u = sign;
            else { int sh = 0; while (!(m & 1024)) { m <<= 1; ++sh; } u = sign | ((uint32_t)(113 - sh) << 23) | ((m & 1023) << 13); }
        } else if (e == 31) u = sign | 0x7F800000u | (m << 13);
        else u = sign | ((e + 112) << 23) | (m << 13);
        float f; memcpy(&f, &u, 4); return f;
    };
    for (int r = 0; r < 4; ++r) {
        int row = 4 * (lane >> 4) + r, col = lane & 15;
        float acc = c[r];
        for (int k = 0; k < 32; ++k) {
            const uint16_t* ap = reinterpret_cast<const uint16_t*>(&S.xq[4 * (base + row + 16 * (k >> 3))]);
            const uint16_t* bp = reinterpret_cast<const uint16_t*>(&S.xq2[4 * (base + col + 16 * (k >> 3))]);
            acc = fmaf(dec(ap[k & 7]), dec(bp[k & 7]), acc);
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
