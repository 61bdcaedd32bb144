// Round 4, the ONE follow-up VERDICT r3 item 3 asked for on the L2-resident exchange: the flag polls of scripts/xcd_flags.hip (sc1 loads)
// DO observe fresh data, the sc1 data loads of the same kernel returned stale values -- what differs?  Three suspects, one run:
//   (1) placement: the experiment ASSUMES workgroup i runs on XCD i % 8 (so that producer and consumer share an L2).  Every workgroup
//       records HW_REG_XCC_ID; the host counts the workgroups for which the assumption is wrong and classifies every stale value by
//       "producer on my XCD" / "producer on another XCD";
//   (2) the L1: a line the consumer read in an EARLIER iteration may still sit in its L1 -- FRESH = 1 gives every iteration its own slab
//       (first touch: nothing to be stale in the L1), FRESH = 0 re-reads the same addresses;
//   (3) the access width: 4-byte vs 16-byte sc1 loads.
// Control: buffer_inv sc1 + plain loads (the mode that worked).
//   hipcc --offload-arch=gfx950 -O3 scripts/xcd_stale_probe.hip -o build/xcd_stale_probe && ./build/xcd_stale_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define NXCD 8
#define WG_PER_XCD 64
#define THREADS 256
#define ROWS 128                 // 8 KB rows: 1 MB per XCD-group slab
#define ITERS 4
#define SPIN_LIMIT 4000000

__device__ __forceinline__ bool flag_barrier(unsigned* flags, int me, unsigned epoch, int* err, bool inv) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(flags + me, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    bool ok = true;
    if (threadIdx.x < 64) {
        unsigned spins = 0;
        for (;;) {
            const unsigned f = __hip_atomic_load(flags + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (__builtin_amdgcn_ballot_w64(f >= epoch) == ~0ull) break;
            if (++spins > SPIN_LIMIT) { ok = false; if (threadIdx.x == 0) atomicExch(err, 1); break; }
            __builtin_amdgcn_s_sleep(1);
        }
        if (inv) asm volatile("buffer_inv sc1" ::: "memory");
    }
    __syncthreads();
    return ok;
}

typedef unsigned u4 __attribute__((ext_vector_type(4)));
// MODE 0: global_load_dwordx4 sc1;  1: global_load_dword sc1 (x component only);  2: plain load (after buffer_inv sc1)
template <int MODE>
__device__ __forceinline__ float ld_x(const float4* p) {
    if (MODE == 0) {
        u4 r;
        asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(r) : "v"(p) : "memory");
        return __uint_as_float(r.x);
    } else if (MODE == 1) {
        unsigned r;
        asm volatile("global_load_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(r) : "v"(p) : "memory");
        return __uint_as_float(r);
    }
    return p->x;
}

// slab of one XCD group: ROWS x 512 float4.  Workgroup r of a group owns float4 columns [8 r, 8 r + 8) of every row.
// stale[bx][producer r] counts the values workgroup bx read from producer r's columns that were not this iteration's.
template <int MODE, int FRESH>
__global__ void __launch_bounds__(THREADS) probe(float4* slabs, unsigned* flags, int* err, unsigned* xcc_of, unsigned* stale) {
    const int bx = blockIdx.x, xcd = bx % NXCD, r = bx / NXCD, t = threadIdx.x;
    if (t == 0) {
        unsigned x;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
        xcc_of[bx] = x & 0xf;
    }
    unsigned* fl = flags + xcd * WG_PER_XCD;
    const size_t slab_f4 = (size_t)ROWS * 512;
    unsigned phase = 0;
    for (int it = 0; it < ITERS; ++it) {
        float4* slab = slabs + ((size_t)(FRESH ? it : 0) * NXCD + xcd) * slab_f4;
        for (int row = t >> 3; row < ROWS; row += THREADS / 8) {
            const float v = (float)(1000 * (it + 1) + r);
            slab[(size_t)row * 512 + 8 * r + (t & 7)] = make_float4(v, v, v, v);
        }
        if (!flag_barrier(fl, r, ++phase, err, MODE == 2)) return;
        for (int row = r; row < ROWS; row += WG_PER_XCD) {
            for (int half = 0; half < 2; ++half) {
                const int col = half * 256 + t;
                const float got = ld_x<MODE>(slab + (size_t)row * 512 + col);
                if (got != (float)(1000 * (it + 1) + (col >> 3))) atomicAdd(stale + ((size_t)it * NXCD * WG_PER_XCD + bx) * WG_PER_XCD + (col >> 3), 1u);
            }
        }
        if (!flag_barrier(fl, r, ++phase, err, false)) return;
    }
}

// Part 2: the exchange RATE without an invalidate.  Same traffic pattern as scripts/xcd_flags.hip (128-byte column pieces written, whole
// 8 KB rows read back, two flag barriers per iteration), slabs of 1 ... 4 MB per XCD, every value checked.
template <int MODE>
__global__ void __launch_bounds__(THREADS) timed(float4* slabs, size_t slab_f4, int rows, int iters, unsigned* flags, int* err) {
    const int bx = blockIdx.x, xcd = bx % NXCD, r = bx / NXCD, t = threadIdx.x;
    float4* slab = slabs + (size_t)xcd * slab_f4;
    unsigned* fl = flags + xcd * WG_PER_XCD;
    unsigned phase = 0, bad = 0;
    for (int it = 0; it < iters; ++it) {
        for (int row = t >> 3; row < rows; row += THREADS / 8) {
            const float v = (float)(it + row + r);
            slab[(size_t)row * 512 + 8 * r + (t & 7)] = make_float4(v, v, v, v);
        }
        if (!flag_barrier(fl, r, ++phase, err, MODE == 2)) return;
        for (int row = r; row < rows; row += WG_PER_XCD) {
            for (int half = 0; half < 2; ++half) {
                const int col = half * 256 + t;
                bad += ld_x<MODE>(slab + (size_t)row * 512 + col) != (float)(it + row + (col >> 3));
            }
        }
        if (!flag_barrier(fl, r, ++phase, err, false)) return;
    }
    if (bad) atomicExch(err, 2);
}

int main() {
    const int grid = NXCD * WG_PER_XCD;
    unsigned *flags, *xcc_of, *stale;
    int* err;
    float4* slabs;
    const size_t slab_f4 = (size_t)ROWS * 512;
    hipMalloc(&flags, grid * sizeof(unsigned));
    hipMalloc(&xcc_of, grid * sizeof(unsigned));
    hipMalloc(&stale, (size_t)ITERS * grid * WG_PER_XCD * sizeof(unsigned));
    hipMalloc(&err, sizeof(int));
    hipMalloc(&slabs, slab_f4 * 16 * NXCD * ITERS);
    std::vector<unsigned> hx(grid), hs((size_t)ITERS * grid * WG_PER_XCD);
    const char* mname[] = {"sc1 dwordx4 loads, no invalidate", "sc1 dword loads, no invalidate", "buffer_inv sc1 + plain loads (control)"};
    for (int mode = 0; mode < 3; ++mode) {
        for (int fresh = 0; fresh < 2; ++fresh) {
            hipMemset(flags, 0, grid * sizeof(unsigned));
            hipMemset(err, 0, sizeof(int));
            hipMemset(stale, 0, hs.size() * sizeof(unsigned));
            hipMemset(slabs, 0, slab_f4 * 16 * NXCD * ITERS);
            hipDeviceSynchronize();
#define RUN(M, F) probe<M, F><<<grid, THREADS>>>(slabs, flags, err, xcc_of, stale)
            if (mode == 0 && !fresh) RUN(0, 0); else if (mode == 0) RUN(0, 1); else if (mode == 1 && !fresh) RUN(1, 0);
            else if (mode == 1) RUN(1, 1); else if (!fresh) RUN(2, 0); else RUN(2, 1);
            hipDeviceSynchronize();
            int herr = 0;
            hipMemcpy(&herr, err, sizeof(int), hipMemcpyDeviceToHost);
            hipMemcpy(hx.data(), xcc_of, grid * sizeof(unsigned), hipMemcpyDeviceToHost);
            hipMemcpy(hs.data(), stale, hs.size() * sizeof(unsigned), hipMemcpyDeviceToHost);
            int wrong = 0;
            for (int bx = 0; bx < grid; ++bx) wrong += (int)(hx[bx] != (unsigned)(bx % NXCD));
            // does every group at least sit on ONE XCD (a consistent permutation), even if not XCD bx % 8?
            int split_groups = 0;
            for (int g = 0; g < NXCD; ++g) {
                bool same = true;
                for (int r = 1; r < WG_PER_XCD; ++r) same = same && hx[r * NXCD + g] == hx[g];
                split_groups += same ? 0 : 1;
            }
            printf("%s, %s slab per iteration%s: %d of %d workgroups NOT on XCD (blockIdx %% 8); %d of 8 groups spread over several XCDs\n",
                   mname[mode], fresh ? "a FRESH" : "the SAME", herr == 1 ? " [BARRIER TIMED OUT]" : "", wrong, grid, split_groups);
            for (int it = 0; it < ITERS; ++it) {
                unsigned long same_x = 0, diff_x = 0, total = 0;
                for (int bx = 0; bx < grid; ++bx)
                    for (int pr = 0; pr < WG_PER_XCD; ++pr) {
                        const unsigned n = hs[((size_t)it * grid + bx) * WG_PER_XCD + pr];
                        const int producer = pr * NXCD + bx % NXCD;
                        (hx[producer] == hx[bx] ? same_x : diff_x) += n;
                        total += n;
                    }
                printf("   iteration %d: %lu stale values of %d read (producer on the reader's XCD: %lu, on another XCD: %lu)\n", it, total,
                       grid * (ROWS / WG_PER_XCD) * 2 * THREADS, same_x, diff_x);
            }
        }
    }
    // ---- part 2 ----
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int row_counts[] = {128, 256, 384, 512, 1024};          // x 8 KB: 1, 2, 3, 4, 8 MB per XCD
    for (int mode = 0; mode < 3; mode += 2) {
        printf("exchange rate, consumer: %s\n%12s %14s %14s\n", mname[mode], "MB per XCD", "exchange GB/s", "us per phase");
        for (int rows : row_counts) {
            const size_t f4 = (size_t)rows * 512;
            float4* sl;
            if (hipMalloc(&sl, f4 * 16 * NXCD) != hipSuccess) break;
            hipMemset(sl, 0, f4 * 16 * NXCD);
            const int iters = 200;
            float best = 1e30f;
            int herr = 0;
            for (int rep = 0; rep < 3 && !herr; ++rep) {
                hipMemset(flags, 0, grid * sizeof(unsigned));
                hipMemset(err, 0, sizeof(int));
                hipEventRecord(e0);
                if (mode == 0) timed<0><<<grid, THREADS>>>(sl, f4, rows, iters, flags, err);
                else timed<2><<<grid, THREADS>>>(sl, f4, rows, iters, flags, err);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                hipMemcpy(&herr, err, sizeof(int), hipMemcpyDeviceToHost);
                if (ms < best) best = ms;
            }
            if (herr) printf("%12.1f   ERROR %d (1 = barrier timed out, 2 = stale data)\n", rows * 8.0 / 1024, herr);
            else printf("%12.1f %14.0f %14.2f\n", rows * 8.0 / 1024, (double)f4 * 16 * NXCD * 2 * iters / best / 1e6, best * 1e3 / (2.0 * iters));
            hipFree(sl);
        }
    }
    return 0;
}
