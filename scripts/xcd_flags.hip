// Round-2 redo of scripts/xcd_exchange.hip ("weak" item 10): the 20 us intra-XCD barrier measured there was an
// artefact of its design -- 64 workgroups polling ONE counter with read-modify-write atomics.  Here every workgroup owns
// a flag word (one relaxed agent-scope store after its data stores have completed), and a consumer polls all 64 flags of
// its XCD with ONE 64-lane load (sc1: served by the L2) + a wave ballot:
//     producer : data stores -> s_waitcnt vmcnt(0) -> __syncthreads -> lane 0: flag[xcd][wg] = epoch   (relaxed, agent)
//     consumer : wave 0: do { f = load(flag[xcd][lane]) } while (!all(f >= epoch)); -> __syncthreads -> buffer_inv sc1
// Measured: (1) the barrier alone, (2) a W-like slab exchange (128-byte column pieces written, 8 KB rows read back)
// for slabs of 1 ... 128 MB per XCD, values checked.  Spins are bounded (a non-resident grid reports an error).
//   hipcc --offload-arch=gfx950 -O3 scripts/xcd_flags.hip -o build/xcd_flags && ./build/xcd_flags
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define NXCD 8
#define WG_PER_XCD 64
#define THREADS 256
#define SPIN_LIMIT 4000000

// INV: how the consumer side drops stale L1 lines after the barrier.  2 = every wavefront issues buffer_inv sc1 (first
// attempt: 21 us per barrier -- each invalidate costs ~2 us and the 8 wavefronts of a CU's two workgroups serialise);
// 1 = wavefront 0 only; 0 = none (the consumer then reads the exchanged data with sc1 loads, which bypass the L1).
template <int SLEEP, int INV>
__device__ __forceinline__ bool flag_barrier(unsigned* flags /* [WG_PER_XCD] of this XCD */, int me, unsigned epoch, int* err) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // this wave's stores have reached the L2
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(flags + me, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    bool ok = true;
    if (threadIdx.x < 64) {
        unsigned spins = 0;
        for (;;) {
            const unsigned f = __hip_atomic_load(flags + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (__builtin_amdgcn_ballot_w64(f >= epoch) == ~0ull) break;
            if (++spins > SPIN_LIMIT) { ok = false; if (threadIdx.x == 0) atomicExch(err, 1); break; }
            if (SLEEP) __builtin_amdgcn_s_sleep(SLEEP);
        }
    }
    if (INV == 1 && threadIdx.x < 64) asm volatile("buffer_inv sc1" ::: "memory");
    __syncthreads();
    if (INV == 2) asm volatile("buffer_inv sc1" ::: "memory");           // drop stale L1 lines; the L2 keeps its data
    return ok;
}

template <int SLEEP, int INV>
__global__ void __launch_bounds__(THREADS) barrier_only(int iters, unsigned* flags, int* err) {
    const int xcd = blockIdx.x % NXCD, me = blockIdx.x / NXCD;
    for (int it = 1; it <= iters; ++it)
        if (!flag_barrier<SLEEP, INV>(flags + xcd * WG_PER_XCD, me, (unsigned)it, err)) return;
}

typedef unsigned u4 __attribute__((ext_vector_type(4)));
// 16-byte loads with an explicit cache policy.  MODE 0: sc1 (agent scope; round 2: returned stale lines), 3: sc0 sc1 (system scope),
// 4: nt, 5: sc0 sc1 nt -- round 3's one follow-up: can the consumer skip the L1 invalidate with a load that
// cannot hit the L1?
// (Both loads AND the s_waitcnt live in ONE asm statement: with the wait in a separate statement nothing stops the compiler from
// scheduling the consumers of the loaded registers above it -- the first version of this experiment, and round 2's, read the
// registers before the data had arrived and reported every policy as "stale".)
template <int MODE>
__device__ __forceinline__ void ld_pol2(const float4* pa, const float4* pb, float4& a, float4& b) {
    u4 ra, rb;
    if (MODE == 0) asm volatile("global_load_dwordx4 %0, %2, off sc1\n\tglobal_load_dwordx4 %1, %3, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(ra), "=&v"(rb) : "v"(pa), "v"(pb) : "memory");
    else if (MODE == 3) asm volatile("global_load_dwordx4 %0, %2, off sc0 sc1\n\tglobal_load_dwordx4 %1, %3, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(ra), "=&v"(rb) : "v"(pa), "v"(pb) : "memory");
    else if (MODE == 4) asm volatile("global_load_dwordx4 %0, %2, off nt\n\tglobal_load_dwordx4 %1, %3, off nt\n\ts_waitcnt vmcnt(0)" : "=&v"(ra), "=&v"(rb) : "v"(pa), "v"(pb) : "memory");
    else asm volatile("global_load_dwordx4 %0, %2, off sc0 sc1 nt\n\tglobal_load_dwordx4 %1, %3, off sc0 sc1 nt\n\ts_waitcnt vmcnt(0)" : "=&v"(ra), "=&v"(rb) : "v"(pa), "v"(pb) : "memory");
    a = make_float4(__uint_as_float(ra.x), __uint_as_float(ra.y), __uint_as_float(ra.z), __uint_as_float(ra.w));
    b = make_float4(__uint_as_float(rb.x), __uint_as_float(rb.y), __uint_as_float(rb.z), __uint_as_float(rb.w));
}
// producer side of the same question: stores at agent scope (PST = 1: `sc1`) instead of plain write-through stores
template <int PST>
__device__ __forceinline__ void st_pol(float4* p, float v) {
    if (PST == 0) { *p = make_float4(v, v, v, v); return; }
    u4 d; d.x = d.y = d.z = d.w = __float_as_uint(v);
    asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(p), "v"(d) : "memory");
}

// slab of one XCD: rows x 512 float4 (8 KB rows).  Workgroup r owns float4 columns [8 r, 8 r + 8) (128 bytes of a row).
template <int INV, int PST>
__global__ void __launch_bounds__(THREADS) exchange(float4* slabs, size_t slab_f4, int rows, int iters, unsigned* flags, int* err,
                                                    float* sink) {
    const int bx = blockIdx.x;
    const int xcd = bx % NXCD, r = bx / NXCD;
    float4* slab = slabs + (size_t)xcd * slab_f4;
    unsigned* fl = flags + xcd * WG_PER_XCD;
    const int t = threadIdx.x;
    float acc = 0.f;
    unsigned phase = 0;
    for (int it = 0; it < iters; ++it) {
        for (int row = t >> 3; row < rows; row += THREADS / 8) {
            const float v = (float)(it + row + r);
            st_pol<PST>(slab + (size_t)row * 512 + 8 * r + (t & 7), v);
        }
        if (!flag_barrier<1, (INV == 1 ? 1 : 0)>(fl, r, ++phase, err)) return;
        for (int row = r; row < rows; row += WG_PER_XCD) {
            float4 a, b;
            if (INV != 1) ld_pol2<INV>(slab + (size_t)row * 512 + t, slab + (size_t)row * 512 + 256 + t, a, b);
            else { a = slab[(size_t)row * 512 + t]; b = slab[(size_t)row * 512 + 256 + t]; }
            acc += a.x + b.x - 2.f * (float)(it + row) - (float)(t >> 3) - (float)((256 + t) >> 3);
        }
        if (!flag_barrier<1, 0>(fl, r, ++phase, err)) return;      // write-after-read: nothing to invalidate
    }
    if (acc != 0.f) atomicExch(err, 2);
    if (acc == 1234.5f) sink[0] = acc;
}

int main() {
    const int grid = NXCD * WG_PER_XCD;
    unsigned* flags; hipMalloc(&flags, NXCD * WG_PER_XCD * sizeof(unsigned));
    int* err; hipMalloc(&err, sizeof(int));
    float* sink; hipMalloc(&sink, 64);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    {
        const int iters = 2000;
        float ms; int herr;
#define RUN_BAR(S, I, label)                                                                                         \
        hipMemset(flags, 0, NXCD * WG_PER_XCD * sizeof(unsigned)); hipMemset(err, 0, sizeof(int));                \
        hipEventRecord(e0); barrier_only<S, I><<<grid, THREADS>>>(iters, flags, err); hipEventRecord(e1);            \
        hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);                                                 \
        hipMemcpy(&herr, err, sizeof(int), hipMemcpyDeviceToHost);                                                 \
        printf("flag barrier alone, 64 workgroups per XCD x 8 XCDs (%s): %.2f us%s\n", label, ms * 1e3 / iters, herr ? "  [TIMED OUT]" : "");
        RUN_BAR(1, 2, "s_sleep 1, buffer_inv by all 4 wavefronts")
        RUN_BAR(1, 1, "s_sleep 1, buffer_inv by wavefront 0")
        RUN_BAR(1, 0, "s_sleep 1, no invalidate")
        RUN_BAR(0, 0, "busy poll, no invalidate")
    }
    const int row_counts[] = {128, 256, 384, 512, 1024, 2048, 16384};    // x 8 KB: 1, 2, 3, 4, 8, 16, 128 MB per XCD
    const int modes[][2] = {{1, 0}, {0, 0}, {3, 0}, {4, 0}, {5, 0}, {0, 1}, {3, 1}};
    const char* names[] = {"?", "buffer_inv sc1 by wavefront 0 + plain loads", "", "no invalidate, sc0 sc1 loads", "no invalidate, nt loads",
                           "no invalidate, sc0 sc1 nt loads"};
    for (const auto& md : modes) {
    const int inv = md[0], pst = md[1];
    printf("consumer: %s; producer: %s stores\n%12s %14s %14s\n", inv == 0 ? "no invalidate, sc1 loads" : names[inv], pst ? "sc1" : "plain",
           "MB per XCD", "exchange GB/s", "us per phase");
    for (int rows : row_counts) {
        const size_t slab_f4 = (size_t)rows * 512;
        float4* slabs;
        if (hipMalloc(&slabs, slab_f4 * 16 * NXCD) != hipSuccess) { printf("alloc failed\n"); break; }
        hipMemset(slabs, 0, slab_f4 * 16 * NXCD);
        const int iters = rows <= 1024 ? 200 : (rows <= 2048 ? 50 : 8);
        float best = 1e30f;
        int herr = 0;
        for (int rep = 0; rep < 3; ++rep) {
            hipMemset(flags, 0, NXCD * WG_PER_XCD * sizeof(unsigned));
            hipMemset(err, 0, sizeof(int));
            hipEventRecord(e0);
#define XRUN(I, P) exchange<I, P><<<grid, THREADS>>>(slabs, slab_f4, rows, iters, flags, err, sink)
            if (inv == 1) XRUN(1, 0); else if (inv == 0 && !pst) XRUN(0, 0); else if (inv == 3 && !pst) XRUN(3, 0);
            else if (inv == 4) XRUN(4, 0); else if (inv == 5) XRUN(5, 0); else if (inv == 0) XRUN(0, 1); else XRUN(3, 1);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            hipMemcpy(&herr, err, sizeof(int), hipMemcpyDeviceToHost);
            if (herr) break;
            if (ms < best) best = ms;
        }
        if (herr) { printf("%12.1f   ERROR %d (1 = barrier timed out, 2 = stale data)\n", rows * 8.0 / 1024, herr); hipFree(slabs); continue; }
        const double bytes = (double)slab_f4 * 16 * NXCD * 2 * iters;
        printf("%12.1f %14.0f %14.2f\n", rows * 8.0 / 1024, bytes / best / 1e6, best * 1e3 / (2.0 * iters));
        hipFree(slabs);
    }
    }
    return 0;
}
