// Micro-benchmark (round 6, second probe): what does the LENGTH of the contiguous piece a wavefront instruction touches in each row of a
// channel-major tensor cost?  The matrix-core projection kernels write (in_proj: xT, vg) or read (out_proj: y, x0) channel-major rows 2 MB apart,
// one 64-position tile = 128 bytes per row at a time (a 64-lane dwordx4 instruction = 8 rows x 128 B); the streaming shell kernels touch
// 4 KB of ONE row per workgroup step and run at 5+ TB/s.  Here every 16-byte lane access is coalesced within its row (lane = (row, 16-byte column),
// columns fastest), only the piece length varies: 128 / 256 / 512 / 1024 bytes per row and instruction.
//   modes:  W  stores only        R  loads only        C  position-major coalesced loads (1 KB per instruction) feeding channel-major stores
//   hipcc --offload-arch=gfx950 -O3 scripts/rowpiece_probe.hip -o build/rowpiece_probe && build/rowpiece_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned u4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

constexpr int ROWS = 1024;                 // channel rows (3 D + D at d_model 256)
constexpr int RPW = 64;                    // rows a wavefront owns (in_proj: 16 channels x (x0, x1, v) + 16 rows of vg)

// A wavefront owns RPW rows and walks a run of tiles of PIECE bytes along the positions; per tile it touches every one of its rows once:
// RPW * PIECE / 1024 instructions of 1 KB.  grid = (ROWS / RPW row groups) x runs; the row groups of a run sit on one XCD (as the kernels' do).
template <int PIECE, int MODE, bool NT>
__global__ void __launch_bounds__(256) probe(const char* __restrict__ src, char* __restrict__ dst, const char* __restrict__ pm, size_t cs,
                                             size_t row_bytes, int tiles_per_wave, unsigned* sink) {
    constexpr int LPR = PIECE / 16, RPI = 64 / LPR;                 // lanes per row, rows per instruction
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, r = lane / LPR, c = lane % LPR;
    const int wg = blockIdx.x, xcd = wg & 7, seq = wg >> 3;
    constexpr int NG = ROWS / RPW / 4;                              // row-group quads: a workgroup's 4 wavefronts own 4 x RPW rows
    const int rg = seq % NG, run = (seq / NG) * 8 + xcd;
    const size_t row0 = (size_t)(rg * 4 + wave) * RPW;
    const size_t t0 = (size_t)run * tiles_per_wave;
    u4 acc = {0, 0, 0, 0};
    for (int t = 0; t < tiles_per_wave; ++t) {
        const size_t col = (t0 + t) * PIECE + c * 16;
        if (col >= row_bytes) break;
        if (MODE == 2) {                                            // position-major operand: PIECE / 2 positions x 512 B, fully coalesced 1 KB loads
#pragma unroll
            for (int i = 0; i < PIECE / 2 * 512 / 1024 / 4; ++i)    // (a quarter of the tile per wavefront: the 4 wavefronts of a workgroup share it)
                acc ^= *reinterpret_cast<const u4*>(pm + ((t0 + t) * (PIECE / 2) * 512 + (size_t)(wave * (PIECE / 2 * 512 / 4)) + i * 1024 + lane * 16));
        }
#pragma unroll
        for (int i = 0; i < RPW / RPI; ++i) {
            const size_t off = (row0 + i * RPI + r) * cs + col;
            if (MODE == 1) acc ^= *reinterpret_cast<const u4*>(src + off);
            else {
                u4 v = acc; v.x += i + t;
                if (NT) __builtin_nontemporal_store(v, reinterpret_cast<u4*>(dst + off)); else *reinterpret_cast<u4*>(dst + off) = v;
            }
        }
    }
    if (MODE == 1 && acc.x == 0x12345u) sink[0] = acc.y;
}

static hipEvent_t e0, e1;
template <class F> static double ms_of(F f) {
    const int it = 10; float ms;
    f(); f(); CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0)); for (int i = 0; i < it; ++i) f(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&ms, e0, e1));
    return ms / it;
}

template <int PIECE, int MODE, bool NT>
static void run(const char* src, char* dst, const char* pm, size_t cs, size_t row_bytes, unsigned* sink, int wgs_per_cu) {
    constexpr int NG = ROWS / RPW / 4;
    const int runs = 256 * wgs_per_cu / NG;                         // one round of resident workgroups
    const int tiles = (int)(row_bytes / PIECE), tpw = (tiles + runs - 1) / runs;
    const int grid = ((runs + 7) / 8) * 8 * NG;
    const double ms = ms_of([&] { probe<PIECE, MODE, NT><<<grid, 256>>>(src, dst, pm, cs, row_bytes, tpw, sink); });
    const double bytes = (double)ROWS * row_bytes * (MODE == 2 ? 1.0 + 0.25 : 1.0);
    printf("%s  piece %4d B  %d WG/CU  %s: %7.1f us  %6.0f GB/s\n", MODE == 0 ? "W stores        " : MODE == 1 ? "R loads         " : "C pm loads+store",
           PIECE, wgs_per_cu, NT ? "nt" : "  ", ms * 1e3, bytes / ms / 1e6);
}

int main() {
    const size_t row_bytes = (size_t)1 << 21;                       // 2^20 positions x 2 B
    const size_t cs = row_bytes;
    char *a, *b, *pm;
    unsigned* sink;
    CK(hipMalloc(&a, ROWS * cs)); CK(hipMalloc(&b, ROWS * cs)); CK(hipMalloc(&pm, ((size_t)1 << 20) * 512)); CK(hipMalloc(&sink, 64));
    CK(hipMemset(a, 1, ROWS * cs)); CK(hipMemset(b, 0, ROWS * cs)); CK(hipMemset(pm, 2, ((size_t)1 << 20) * 512));
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    printf("1024 channel-major rows of 2 MB (2.15 GB), rows 2 MB apart; a wavefront owns 64 rows and touches PIECE bytes of each per tile\n");
    for (int w : {2, 4, 8}) {
        if (w == 2) { run<128, 0, true>(a, b, pm, cs, row_bytes, sink, w); run<256, 0, true>(a, b, pm, cs, row_bytes, sink, w); run<512, 0, true>(a, b, pm, cs, row_bytes, sink, w); run<1024, 0, true>(a, b, pm, cs, row_bytes, sink, w); }
        if (w == 4) { run<128, 0, true>(a, b, pm, cs, row_bytes, sink, w); run<256, 0, true>(a, b, pm, cs, row_bytes, sink, w); run<512, 0, true>(a, b, pm, cs, row_bytes, sink, w); run<1024, 0, true>(a, b, pm, cs, row_bytes, sink, w); }
        if (w == 8) { run<128, 0, true>(a, b, pm, cs, row_bytes, sink, w); run<256, 0, true>(a, b, pm, cs, row_bytes, sink, w); run<512, 0, true>(a, b, pm, cs, row_bytes, sink, w); run<1024, 0, true>(a, b, pm, cs, row_bytes, sink, w); }
    }
    run<128, 0, false>(a, b, pm, cs, row_bytes, sink, 4); run<512, 0, false>(a, b, pm, cs, row_bytes, sink, 4);
    for (int w : {2, 4, 8}) {
        if (w == 2) { run<128, 1, false>(a, b, pm, cs, row_bytes, sink, w); run<256, 1, false>(a, b, pm, cs, row_bytes, sink, w); run<512, 1, false>(a, b, pm, cs, row_bytes, sink, w); }
        if (w == 4) { run<128, 1, false>(a, b, pm, cs, row_bytes, sink, w); run<256, 1, false>(a, b, pm, cs, row_bytes, sink, w); run<512, 1, false>(a, b, pm, cs, row_bytes, sink, w); }
        if (w == 8) { run<128, 1, false>(a, b, pm, cs, row_bytes, sink, w); run<256, 1, false>(a, b, pm, cs, row_bytes, sink, w); run<512, 1, false>(a, b, pm, cs, row_bytes, sink, w); }
    }
    for (int w : {2, 4}) {
        if (w == 2) { run<128, 2, true>(a, b, pm, cs, row_bytes, sink, w); run<256, 2, true>(a, b, pm, cs, row_bytes, sink, w); run<512, 2, true>(a, b, pm, cs, row_bytes, sink, w); }
        if (w == 4) { run<128, 2, true>(a, b, pm, cs, row_bytes, sink, w); run<256, 2, true>(a, b, pm, cs, row_bytes, sink, w); run<512, 2, true>(a, b, pm, cs, row_bytes, sink, w); }
    }
    return 0;
}
