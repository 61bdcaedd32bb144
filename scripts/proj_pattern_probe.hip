// Micro-benchmark (round 6): the MEMORY ACCESS PATTERN of an in_proj-shaped kernel without its arithmetic -- what rate does the part sustain for
// "read a position-major (P, K) 16-bit operand, write 3K + K channel-major rows of P positions" as a function of
//   * the positions a wavefront owns per tile (T = 32: 64-byte row pieces per store instruction, T = 64: 128-byte pieces),
//   * who owns neighbouring tiles (MODE 0: a wavefront walks its own run of consecutive tiles; MODE 1: the W wavefronts of a workgroup take W
//     adjacent tiles per step, so a workgroup writes W * T * 2 contiguous bytes of every row per step),
//   * the wavefronts per CU (threads per workgroup, one workgroup per CU: LDS is claimed as the weights would claim it).
// Stores depend on the loads (a tile's loads are reduced into the stored value), loads are issued back to back like the real kernel's operand
// fetch.  The current weights-stationary kernel's pattern for comparison: scripts/build_variant.sh's PJ_DBG_NO_* builds.
//   hipcc --offload-arch=gfx950 -O3 scripts/proj_pattern_probe.hip -o build/proj_pattern_probe && build/proj_pattern_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned u4 __attribute__((ext_vector_type(4)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

constexpr int K = 256, KB = K * 2;          // bytes per operand row
constexpr int QROWS = 256;                  // output rows a workgroup owns: 64 channels x (x0, x1, v) + 64 rows of vg

// grid: 256 workgroups = 64 runs x 4 channel quarters; the quarters of one run share an XCD (workgroups are dealt to XCDs round-robin)
template <int T, int MODE, int WAVES, bool NT>
__global__ void __launch_bounds__(WAVES * 64) probe(const char* __restrict__ u, char* __restrict__ out, size_t cs, int P, int lds_claim) {
    extern __shared__ char smem[];
    if (lds_claim < 0) smem[threadIdx.x] = 0;                    // (never true: keeps the dynamic LDS allocation alive)
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, j = lane & 15, kq = lane >> 4;
    const int wg = blockIdx.x, xcd = wg & 7, seq = wg >> 3, q = seq & 3, run = (seq >> 2) * 8 + xcd;
    const int runs = gridDim.x / 4;
    const int tiles_per_run = P / T / runs, tiles_per_wave = tiles_per_run / WAVES;
    char* const obase = out + (size_t)q * QROWS * cs;
    for (int s = 0; s < tiles_per_wave; ++s) {
        const int tile = run * tiles_per_run + (MODE == 0 ? wave * tiles_per_wave + s : s * WAVES + wave);
        const size_t p0 = (size_t)tile * T;
        u4 acc = {0, 0, 0, 0};
        // operand fetch: T / 16 row tiles x 8 k-steps, lane (j, kq) reads 16 bytes of row j at 64 ks + 16 kq (the v_mfma_f32_16x16x32 A fragment)
#pragma unroll
        for (int rt = 0; rt < T / 16; ++rt) {
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {
                const u4 v = *reinterpret_cast<const u4*>(u + (p0 + rt * 16 + j) * KB + ks * 64 + kq * 16);
                acc ^= v;
            }
        }
        if constexpr (T == 32) {
            // stores: 16 rows x 64 bytes per instruction (lane j -> row, kq -> 16-byte piece of the row's 32 positions)
#pragma unroll
            for (int i = 0; i < QROWS / 16; ++i) {
                u4 v = acc; v.x += i;
                u4* dst = reinterpret_cast<u4*>(obase + (size_t)(i * 16 + j) * cs + p0 * 2 + kq * 16);
                if (NT) __builtin_nontemporal_store(v, dst); else *dst = v;
            }
        } else {
            // T = 64: 8 rows x 128 bytes per instruction
            const int r8 = lane >> 3, pc = lane & 7;
#pragma unroll
            for (int i = 0; i < QROWS / 8; ++i) {
                u4 v = acc; v.x += i;
                u4* dst = reinterpret_cast<u4*>(obase + (size_t)(i * 8 + r8) * cs + p0 * 2 + pc * 16);
                if (NT) __builtin_nontemporal_store(v, dst); else *dst = v;
            }
        }
    }
}

static hipEvent_t e0, e1;
template <class F> static double ms_of(F f) {
    const int it = 10; float ms;
    f(); f(); CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0)); for (int i = 0; i < it; ++i) f(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&ms, e0, e1));
    return ms / it;
}

template <int T, int MODE, int WAVES, bool NT>
static void run(const char* u, char* out, size_t cs, int P, int lds) {
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(probe<T, MODE, WAVES, NT>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    const double ms = ms_of([&] { probe<T, MODE, WAVES, NT><<<256, WAVES * 64, lds>>>(u, out, cs, P, 0); });
    const double bytes = (double)P * KB + 1024.0 * P * 2;       // operand once (its three re-reads by the other quarters are L2 hits) + every row once
    printf("T=%2d  %-9s  waves/CU=%2d  %s stores : %7.1f us   %6.0f GB/s  (operand %.2f GB read x4 through L2, %.2f GB written)\n", T,
           MODE == 0 ? "run-walk" : "adjacent", WAVES, NT ? "nt" : "  ", ms * 1e3, bytes / ms / 1e6, (double)P * KB / 1e9, 1024.0 * P * 2 / 1e9);
}

int main() {
    const int P = 1 << 20;
    const size_t cs = (size_t)P * 2;
    char *u, *out;
    CK(hipMalloc(&u, (size_t)P * KB));
    CK(hipMalloc(&out, 1024 * cs));
    CK(hipMemset(u, 1, (size_t)P * KB));
    CK(hipMemset(out, 0, 1024 * cs));
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int lds = 96 * 1024;                                  // the 64 x 3 weight rows of a channel quarter: one workgroup per CU
    printf("in_proj-shaped traffic, P = %d positions, K = %d: read 0.54 GB (+ 3 L2 re-reads), write 2.15 GB; one workgroup per CU (96 KB LDS claimed)\n", P, K);
    run<32, 0, 16, true>(u, out, cs, P, lds);
    run<32, 1, 16, true>(u, out, cs, P, lds);
    run<32, 0, 12, true>(u, out, cs, P, lds);        // (tiles per run 512 / 12 waves: 42 each, the last 8 tiles of a run unwritten -- 1.6 % fewer bytes)
    run<32, 1, 12, true>(u, out, cs, P, lds);
    run<32, 0, 8, true>(u, out, cs, P, lds);
    run<32, 1, 8, true>(u, out, cs, P, lds);
    run<64, 0, 16, true>(u, out, cs, P, lds);
    run<64, 1, 16, true>(u, out, cs, P, lds);
    run<64, 0, 8, true>(u, out, cs, P, lds);
    run<64, 1, 8, true>(u, out, cs, P, lds);
    run<32, 1, 16, false>(u, out, cs, P, lds);
    run<64, 1, 16, false>(u, out, cs, P, lds);
    return 0;
}
