// proj_kernels.h -- the operator's input projection on the matrix cores, with the front of the element-wise shell in its epilogue.
//
// Reference: src/models/sequence/hyena.py:391-404,420 (order 2):
//     u  = in_proj(u)                       (B, L, D) x (3D, D)^T                      hyena.py:391      nn.Linear(D, 3D)
//     u  = rearrange(u, 'b l d -> b d l');  uc = short_filter(u)[..., :l_filter]       hyena.py:392-394  Conv1d(3D, 3D, 3, groups = 3D, padding = 2)
//     *x, v = uc.split(D, dim = 1);  v = v * x[1]                                      hyena.py:404,420
// Until round 3 this was a library GEMM writing xT (3D, B, L) followed by cm_pre_fwd (cm_kernels.h) reading two thirds of it
// back.  Here ONE kernel produces both tensors the rest of the layer needs:
//     xT [c, b, l] = sum_k W[c, k] u[b, l, k]                    (3D, B, Lx), 16-bit, WITHOUT the bias (as before: the shell
//                                                                 kernels add it on load and return its gradient)
//     vg [b, d, l] = xc[2D + d, b, l] xc[D + d, b, l],  l < Lc   (B, D, Lc),  xc = short conv of (xT + b_in) as in cm_kernels.h
//
// Shape of the GEMM: K = D in {128, 256}, 3D output rows, B L ~ 10^6 columns: 200 flop per byte moved -- HBM-bound on an MI355X
// (2.5 PFLOP/s bf16 against 8 TB/s) even at a fraction of the MFMA peak.  So the design is WEIGHTS-STATIONARY: a wavefront owns 16
// channels of each of the three groups (x0, x1, v) and keeps their 48 weight rows as MFMA A-fragments in registers for the whole
// kernel (48 x K 16-bit values = 96 VGPRs at K = 256, v_mfma_f32_16x16x32); the four wavefronts of a workgroup (64 channels) walk
// the SAME run of positions, 64 at a time: the u tile (64 x K, contiguous in memory) is staged once through LDS, every wavefront
// reads its B-fragments from it (one 16-byte ds_read per 3 MFMAs), and the accumulators (position on lane, channel on register)
// are rounded to the storage type, parked in a wavefront-private LDS tile [channel][position] and leave from there as 16-byte row
// pieces -- xT as is, vg after the 3-tap window + gate, whose two-position halo is the tail of the previous tile (a workgroup's
// first tile is preceded by one warm-up tile whose results are dropped).  A workgroup needs 63 KB of LDS and < 256 registers per
// wavefront: TWO workgroups share a CU (2 wavefronts per SIMD) -- a lone wavefront per SIMD (the first version: 32 channels per
// wavefront on 32x32x16) was bound by its own instruction issue.  The short-conv arithmetic is the one of cm_kernels.h::cm_sc, FMA by
// FMA, on the ROUNDED xT values: vg is what cm_pre_fwd computes from the stored xT (and what the backward, which recomputes the window
// from xT, sees).
//
// Compiled by hipcc for gfx950 (product) and, with -DHIPEMU, by g++ against tests/hipemu (tests only).
#pragma once
#define HY_HELPERS_ONLY
#include "fftconv_kernels.h"
#include "block_kernels.h"            // blk_load / blk_store / wave_sum: the add + LayerNorm arithmetic of the LN-fused out_proj epilogue

namespace hyena {
namespace pj {

enum { PJ_THREADS = 256, PJ_WAVES = 4, PJ_NT = 64 /* positions per tile */,
       PJ_EW = PJ_NT + 8 /* row of the epilogue tile: 8 halo slots (the last two used) + the tile; 144 B: 16-byte aligned, conflict-free */ };

#ifdef HIPEMU
#define HY_WAVE_SYNC_PJ() hipemu::yield(2)
#else
// LDS operations of one wavefront execute in order: lanes of a wavefront exchange data through LDS without a workgroup barrier;
// this only stops the compiler from moving LDS accesses across the exchange point.
#define HY_WAVE_SYNC_PJ() __builtin_amdgcn_wave_barrier()
#endif

// -DPJ_PROFILE (profiling builds only, FULL=1 scripts/build_variant.sh): per-wavefront time (s_memtime, shader clock) spent between the
// phase boundaries of mlp_kernel, summed over its tiles and written by lane 0 to a buffer the host hands over through hyena_pj_prof_set
// (proj.hip); scripts/pj_phase_profile.py prints the table.  The product build contains none of it.
#if defined(PJ_PROFILE) && !defined(HIPEMU)
__device__ unsigned long long* pj_prof_buf = nullptr;
#define PJ_NOW(v)                                                                                       \
    do {                                                                                                \
        __builtin_amdgcn_sched_barrier(0);                                                              \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v)::"memory");                       \
        __builtin_amdgcn_sched_barrier(0);                                                              \
    } while (0)
#define PJ_MARK(i) do { unsigned long long now_; PJ_NOW(now_); pj_d[i] += now_ - pj_t; pj_t = now_; } while (0)
#else
#define PJ_MARK(i) do {} while (0)
#endif

struct Frag { uint32_t w[4]; };                       // eight 16-bit values: one A or B operand of v_mfma_f32_32x32x16
#ifdef HIPEMU
typedef hipemu::floatx16 acc_t;
template <int DT>
__device__ __forceinline__ acc_t mfma(const Frag& a, const Frag& b, acc_t c) {
    hipemu::u32x4 x, y;
    __builtin_memcpy(x.w, a.w, 16);
    __builtin_memcpy(y.w, b.w, 16);
    return hipemu::mfma_f32_32x32x16_h<DT == DT_BF16>(x, y, c);
}
#else
typedef float acc_t __attribute__((ext_vector_type(16)));
template <int DT>
__device__ __forceinline__ acc_t mfma(const Frag& a, const Frag& b, acc_t c) {
    if constexpr (DT == DT_BF16) {
        typedef __bf16 v8 __attribute__((ext_vector_type(8)));
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(v8, a), __builtin_bit_cast(v8, b), c, 0, 0, 0);
    } else {
        typedef _Float16 v8 __attribute__((ext_vector_type(8)));
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(v8, a), __builtin_bit_cast(v8, b), c, 0, 0, 0);
    }
}
#endif

// 16-byte moves.  Global rows start at any even byte offset (odd B L): the vector type carries 2-byte alignment, gfx950 global
// memory takes unaligned dwordx4 accesses.
#ifdef HIPEMU
__device__ __forceinline__ Frag ld16(const void* p) { Frag f; __builtin_memcpy(f.w, p, 16); return f; }
__device__ __forceinline__ void st16(void* p, const Frag& f) { __builtin_memcpy(p, f.w, 16); }
__device__ __forceinline__ Frag lds_ld16(const HY_LDS char* p) { return ld16(p); }
__device__ __forceinline__ void lds_st16(HY_LDS char* p, const Frag& f) { st16(p, f); }
#else
typedef unsigned pj_vec16 __attribute__((ext_vector_type(4), aligned(2)));
typedef unsigned pj_lvec16 __attribute__((ext_vector_type(4)));
// PJ_DBG_* (profiling builds only, scripts/build_variant.sh; results are wrong by construction): leave out the global stores or
// the global loads to see what each side of the memory traffic costs
__device__ __forceinline__ Frag ld16(const void* p) {
#ifdef PJ_DBG_NO_LD
    Frag f; f.w[0] = f.w[1] = f.w[2] = f.w[3] = (uint32_t)(size_t)p; return f;
#else
    return __builtin_bit_cast(Frag, *reinterpret_cast<const pj_vec16*>(p));
#endif
}
#ifndef PJ_POL_ST
#define PJ_POL_ST 1                                                      // 1 = non-temporal stores (outputs are written once, read by a later launch)
#endif
__device__ __forceinline__ void st16(void* p, const Frag& f) {
#ifdef PJ_DBG_NO_ST
    if (f.w[0] == 0x12345678u && f.w[3] == 0x9abcdef0u) *reinterpret_cast<pj_vec16*>(p) = __builtin_bit_cast(pj_vec16, f);
#else
    if (PJ_POL_ST) __builtin_nontemporal_store(__builtin_bit_cast(pj_vec16, f), reinterpret_cast<pj_vec16*>(p));
    else *reinterpret_cast<pj_vec16*>(p) = __builtin_bit_cast(pj_vec16, f);
#endif
}
__device__ __forceinline__ Frag lds_ld16(const HY_LDS char* p) { return __builtin_bit_cast(Frag, *reinterpret_cast<const HY_LDS pj_lvec16*>(p)); }
__device__ __forceinline__ void lds_st16(HY_LDS char* p, const Frag& f) { *reinterpret_cast<HY_LDS pj_lvec16*>(p) = __builtin_bit_cast(pj_lvec16, f); }
#endif

#ifdef HIPEMU
// The test double models the memory pipeline's queue the way the kernels' counted waits assume it: operations retire in issue order, an
// LDS-direct load delivers its 16 bytes only when it retires, `s_waitcnt vmcnt(n)` retires the oldest operations until
// n are left.  A wait that names too many younger operations therefore leaves the data stale under the emulator as well -- on the hardware such
// a mistake is a race that may or may not show.  (One queue per thread: the stack of its fibre.  Stores are performed at issue and only counted.)
struct EmuVmq {
    struct Op { const void* src; void* dst; };
    Op q[128];
    int head = 0, n = 0;
    void push(const void* src, void* dst) { q[(head + n++) & 127] = Op{src, dst}; }
    void retire_to(int keep) {
        while (n > keep) {
            const Op& o = q[head];
            if (o.dst != nullptr) __builtin_memcpy(o.dst, o.src, 16);
            head = (head + 1) & 127;
            --n;
        }
    }
    ~EmuVmq() { retire_to(0); }
};
#define PJ_VMQ_DECL EmuVmq pj_vmq
#define PJ_VMQ_PARAM , EmuVmq& pj_vmq
#define PJ_VMQ_ARG , pj_vmq
__device__ __forceinline__ void glds16(const char* base, uint32_t voff, HY_LDS char* wave_base, int lane, EmuVmq& q) { q.push(base + voff, wave_base + 16 * lane); }
#define PJ_VMWAIT(n) pj_vmq.retire_to(n)
#define PJ_LGKMWAIT() do {} while (0)
#define PJ_BARRIER() __syncthreads()
#define PJ_ST16(p, f) do { st16((p), (f)); pj_vmq.push(nullptr, nullptr); } while (0)
#else
// global_load_lds_dwordx4 voffset, sbase: 16 bytes from base + voff (per lane) to LDS byte M0 + 16 * lane.  Written as inline assembly on
// purpose: hipcc tracks the built-in's LDS writes on vmcnt and, having no alias information for them, drains EVERY outstanding memory
// operation before any later LDS read of the kernel -- the wavefront tile's reads in the epilogue would wait for the next operand tile and
// for each preceding global store.  Here the kernel's own counted waits (PJ_VMWAIT) are the only ones; the compiler's waits for the memory
// operations it knows can only come out stricter for the extra ones in the queue, never laxer.
#define PJ_VMQ_DECL do {} while (0)
#define PJ_VMQ_PARAM
#define PJ_VMQ_ARG
#define PJ_ST16(p, f) st16((p), (f))
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
__device__ __forceinline__ void glds16(const char* base, uint32_t voff, HY_LDS char* wave_base, int lane) {
    (void)lane;
#ifdef PJ_DBG_NO_LD
    return;                                  // (profiling builds only: the operand tile is never fetched -- results are wrong by construction)
#endif
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"
                 :: "v"(voff), "s"(base), "s"((uint32_t)(size_t)wave_base) : "memory", "m0");
}
#pragma clang diagnostic pop
#define PJ_VMWAIT(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")
#define PJ_LGKMWAIT() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
#define PJ_BARRIER() do { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); } while (0)
#endif

struct True_ { static constexpr bool value = true; };      // compile-time flags handed to the kernels' generic lambdas
struct False_ { static constexpr bool value = false; };

// The operand tile [64 positions][K] of the weights-stationary kernels, brought to LDS by LDS-direct loads: load i of wavefront `wave` fills
// the 1 KB chunk i * 4 + wave; lane -> 16-byte slot S = 64 chunk + lane = (position S / PCS, slot S mod PCS), which holds the source piece
// (slot ^ position mod PCS), PCS = K / 8 pieces per row -- a fragment read of 16 neighbouring rows at one piece index then touches 16
// different bank groups.  FULL = false: rows of positions >= P are not fetched (they keep older values; their results are never stored).
template <int K, bool FULL>
__device__ __forceinline__ void issue_operand_tile(const char* xbase, unsigned p0, unsigned P, HY_LDS char* ubuf, int wave, int lane PJ_VMQ_PARAM) {
    constexpr int PCS = K / 8, NX = PJ_NT * K * 2 / 1024 / PJ_WAVES;
    const char* const tb = HY_UNIFORM_PTR(const char, xbase + (size_t)p0 * K * 2);
    HY_OPAQUE(lane);                             // the NX per-lane offsets are recomputed per tile (a handful of operations) instead of being held in NX registers
    HY_UNROLL
    for (int i = 0; i < NX; ++i) {
        const int chunk = i * PJ_WAVES + wave, S = chunk * 64 + lane, pos = S / PCS, c = (S % PCS) ^ (pos % PCS);
        if (FULL || p0 + (unsigned)pos < P) glds16(tb, (uint32_t)(pos * K * 2 + c * 16), ubuf + chunk * 1024, lane PJ_VMQ_ARG);
    }
}

template <int K> struct PjCfg {
    static_assert(K == 128 || K == 256, "d_model of the HyenaDNA models");
    static constexpr int KS = K / 16;                         // MFMA steps over the contraction
    static constexpr int UROW = (K + 8) * 2;                  // bytes per staged u row: +16 so that the 16-byte fragment reads of 8 neighbouring lanes hit 32 different banks
    static constexpr int UBUF = PJ_NT * UROW;                 // one staging buffer
    static constexpr int CH = PJ_NT * K * 2 / 16 / PJ_THREADS;   // 16-byte chunks of a u tile per thread (8 / 4)
};

struct InProjArgs {
    const void* u;       // (B, Lx, K) 16-bit
    const void* W;       // (3D, K) 16-bit, K = D
    const float* bin;    // (3D,) in_proj bias or null
    const float* w;      // (3D, 3) short-filter taps
    const float* b;      // (3D,) short-filter bias
    void* xT;            // (3D, B, Lx): row (c, b) starts at element c csx + b bsx
    void* vg;            // (B, D, Lc), row pitch ldv: row (b, d) starts at element (b D + d) ldv
    int B, Lx, Lc, D;
    long csx; int bsx;   // bsx >= Lx, csx >= (B - 1) bsx + Lx (packed: B Lx, Lx)
    int ldv;             // >= Lc (packed: Lc)
    int tiles;           // ceil(B Lx / 64)
    int tiles_per_wg;
};

// rows of the C/D operand: register r of a lane in half-wave hb holds channel (r & 3) + 8 (r >> 2) + 4 hb of the 32
__device__ __forceinline__ int pj_row(int r, int hb) { return (r & 3) + 8 * (r >> 2) + 4 * hb; }

// v_mfma_f32_16x16x32: lane l = (j = l & 15, kq = l >> 4) supplies A[j][8 kq ...] and B[8 kq ...][j], owns D[4 kq + r][j], r < 4
#ifdef HIPEMU
typedef hipemu::floatx4 acc4_t;
template <int DT>
__device__ __forceinline__ acc4_t mfma16(const Frag& a, const Frag& b, acc4_t c) {
    hipemu::u32x4 x, y;
    __builtin_memcpy(x.w, a.w, 16);
    __builtin_memcpy(y.w, b.w, 16);
    return hipemu::mfma_f32_16x16x32_h<DT == DT_BF16>(x, y, c);
}
#else
typedef float acc4_t __attribute__((ext_vector_type(4)));
template <int DT>
__device__ __forceinline__ acc4_t mfma16(const Frag& a, const Frag& b, acc4_t c) {
    if constexpr (DT == DT_BF16) {
        typedef __bf16 v8 __attribute__((ext_vector_type(8)));
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(v8, a), __builtin_bit_cast(v8, b), c, 0, 0, 0);
    } else {
        typedef _Float16 v8 __attribute__((ext_vector_type(8)));
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(v8, a), __builtin_bit_cast(v8, b), c, 0, 0, 0);
    }
}
#endif

// The in_proj kernel proper.  A wavefront owns IP_CB = 16 channels of each of the three groups (48 weight rows x K = 96 VGPRs at
// K = 256) on v_mfma_f32_16x16x32 and stays under 256 registers; a workgroup (4 wavefronts, 64 channels) needs 63 KB of LDS -- ONE
// staging buffer, the wavefronts' epilogue tiles and taps -- so TWO workgroups share a CU: 2 wavefronts per SIMD.  (The first
// version -- 32 channels per wavefront on 32x32x16, ~350 registers, one wavefront per SIMD -- was bound by the lone wavefront's
// instruction issue: 0.77 ms at L = 2^20 against 0.55 ms of memory time.)
enum { IP_CB = 16, IP_NT4 = PJ_NT / 16 };
template <int K> struct IpCfg {
    static constexpr int KS = K / 32;                         // v_mfma_f32_16x16x32 steps over the contraction
    static constexpr int PCS = K / 8;                         // 16-byte pieces per row of the u tile
    static constexpr int UROWB = K * 2, UBUF = PJ_NT * UROWB; // the u tile, unpadded: filled by LDS-direct loads, source pieces permuted (issue_operand_tile)
    static constexpr int EROW = PJ_EW * 2;
    static constexpr int EBUF = 3 * IP_CB * EROW;             // per wavefront: [group][channel][PJ_EW]
    static constexpr int TAPS = 2 * IP_CB * 5 * 4;
    static constexpr size_t LDS = (size_t)UBUF + PJ_WAVES * ((size_t)EBUF + TAPS);
};

template <int K, int DT>
__global__ void __launch_bounds__(PJ_THREADS, 2) inproj_pre_fwd_kernel(InProjArgs a) {
    typedef IpCfg<K> C;
    typedef typename Elem<DT>::type elem_t;
    static_assert(sizeof(elem_t) == 2, "16-bit element types only");
    HY_SMEM(smem);
    PJ_VMQ_DECL;
    const int tid = (int)threadIdx.x, wave = HY_SGPR(tid >> 6), lane = tid & 63, j = lane & 15, kq = lane >> 4;
    const int D = a.D;
    const unsigned P = (unsigned)a.B * (unsigned)a.Lx;                       // flattened positions (< 2^31, checked by the host)
    // workgroup -> (channel group of 64, run of tiles).  Workgroups are dealt to the 8 XCDs round-robin; the channel groups of
    // one run of positions get slots of ONE XCD, so its L2 serves all but the first read of the u tiles.
    const int ncg = D / (PJ_WAVES * IP_CB);
    int cg, run;
    {
        const int wg = blockIdx.x, xcd = wg & 7, seq = wg >> 3;
        cg = seq % ncg;
        run = (seq / ncg) * 8 + xcd;
    }
    const int t_begin = run * a.tiles_per_wg;
    if (t_begin >= a.tiles) return;
    const int t_end = (t_begin + a.tiles_per_wg < a.tiles) ? t_begin + a.tiles_per_wg : a.tiles;
    const int d0 = cg * PJ_WAVES * IP_CB + wave * IP_CB;                   // first channel of this wavefront

    HY_LDS char* const ubuf = HY_LDS_CAST(char, smem);
    HY_LDS char* const ebuf = HY_LDS_CAST(char, smem) + C::UBUF + wave * C::EBUF;
    HY_LDS float* const taps = HY_LDS_CAST(float, smem + C::UBUF + PJ_WAVES * C::EBUF + wave * C::TAPS);

    // stationary operand: the 48 weight rows of this wavefront, as A fragments (row = channel j of group g, k = 32 ks + 8 kq ...)
    Frag wf[3][C::KS];
    HY_UNROLL
    for (int g = 0; g < 3; ++g) {
        const char* row = reinterpret_cast<const char*>(a.W) + ((size_t)(g * D + d0 + j) * K + 8 * kq) * 2;
        HY_UNROLL
        for (int ks = 0; ks < C::KS; ++ks) wf[g][ks] = ld16(row + ks * 64);
    }
    // short-filter taps of the x1 and v channels, (w0, w1, w2, b_sc, b_in) per channel
    if (lane < 2 * IP_CB) {
        const int c = (1 + (lane >> 4)) * D + d0 + (lane & 15);
        HY_LDS float* t = taps + lane * 5;
        t[0] = a.w[c * 3]; t[1] = a.w[c * 3 + 1]; t[2] = a.w[c * 3 + 2]; t[3] = a.b[c];
        t[4] = a.bin != nullptr ? a.bin[c] : 0.f;
    }
    // Piece m of a lane: id = lane + 64 m -> group m >> 1, channel (lane >> 3) + 8 (m & 1), piece lane & 7: the m-dependent part
    // of every address is wave-uniform (scalar registers), only piece 0's offset is held per lane.
    const size_t CS = (size_t)a.csx;                                         // elements between the same position of consecutive xT channels
    const size_t xoff0 = (size_t)(d0 + (lane >> 3)) * CS + 8u * (unsigned)(lane & 7);
    const unsigned voff0 = (unsigned)(d0 + (lane >> 3)) * (unsigned)a.ldv + 8u * (unsigned)(lane & 7);
    const char* const ubase = reinterpret_cast<const char*>(a.u);

    const int t_first = t_begin > 0 ? t_begin - 1 : t_begin;                // warm-up tile: provides the halo of tile t_begin
    // (sequence, position within it) of the first position of the current tile, carried along instead of divided out per tile
    unsigned sb = ((unsigned)t_first * PJ_NT) / (unsigned)a.Lx;
    int sl0 = (int)((unsigned)t_first * PJ_NT - sb * (unsigned)a.Lx);
    // The u tile of positions [64 t, 64 t + 64) -- one contiguous block of 64 K 2 bytes -- reaches LDS by LDS-direct loads (round 4; until
    // then global -> registers -> LDS with the wait for it on every tile's critical path: the staged tile, the weights and the accumulators
    // fill the register file, there was nothing to prefetch into).  Tile t + 1 is requested when every wavefront has read tile t's last
    // fragment and lands behind tile t's epilogue; the wait at the top names the stores that may stay in flight behind it.
    const int t_whole = (int)(P / PJ_NT);                                    // tiles below this one are whole
    if (t_first < t_whole) issue_operand_tile<K, true>(ubase, (unsigned)t_first * PJ_NT, P, ubuf, wave, lane PJ_VMQ_ARG);
    else issue_operand_tile<K, false>(ubase, (unsigned)t_first * PJ_NT, P, ubuf, wave, lane PJ_VMQ_ARG);
    bool counted = false;                                                    // the previous tile left exactly IP_ST stores behind the loads
    for (int t = t_first; t < t_end; ++t, sl0 += PJ_NT) {
        while (sl0 >= a.Lx) { sl0 -= a.Lx; ++sb; }
        const unsigned p0 = (unsigned)t * PJ_NT;
        if (counted) PJ_VMWAIT(8);                                           // (IP_ST = 8: six xT stores, two vg stores)
        else PJ_VMWAIT(0);
        PJ_BARRIER();                                                        // everybody's share of the tile has landed
        acc4_t acc[3][IP_NT4];
        HY_UNROLL
        for (int g = 0; g < 3; ++g) {
            HY_UNROLL
            for (int nt = 0; nt < IP_NT4; ++nt) {
                HY_UNROLL
                for (int r = 0; r < 4; ++r) acc[g][nt][r] = 0.f;
            }
        }
        // B fragment of (position tile nt, step ks): 16 bytes of row 16 nt + j, piece 4 ks + kq -> slot (4 ks + kq) ^ ((16 nt + j) mod PCS)
        // = (kq ^ j) ^ (4 ks ^ (16 nt mod PCS)): a per-lane part and a compile-time part
        const int ua = j * C::UROWB, ux = (kq ^ j) * 16;
        HY_UNROLL
        for (int ks = 0; ks < C::KS; ++ks) {
            HY_UNROLL
            for (int nt = 0; nt < IP_NT4; ++nt) {
                const Frag bf = lds_ld16(ubuf + nt * 16 * C::UROWB + ua + (ux ^ (((4 * ks) ^ ((16 * nt) % C::PCS)) * 16)));
                HY_UNROLL
                for (int g = 0; g < 3; ++g) acc[g][nt] = mfma16<DT>(wf[g][ks], bf, acc[g][nt]);
            }
        }
        PJ_BARRIER();                                                        // every wavefront has read its last fragment
        if (t + 1 < t_end) {
            if (t + 1 < t_whole) issue_operand_tile<K, true>(ubase, p0 + PJ_NT, P, ubuf, wave, lane PJ_VMQ_ARG);
            else issue_operand_tile<K, false>(ubase, p0 + PJ_NT, P, ubuf, wave, lane PJ_VMQ_ARG);
        }
        counted = false;
        // ---- epilogue, wavefront-private ---------------------------------------------------------------------------
        // (1) the previous tile's last two positions become this tile's halo (slots 6, 7: one dword per row)
        if (lane < 3 * IP_CB) {
            HY_LDS uint32_t* row = reinterpret_cast<HY_LDS uint32_t*>(ebuf + lane * C::EROW);
            row[3] = row[3 + PJ_NT / 2];
        }
        HY_WAVE_SYNC_PJ();
        // (2) accumulators -> storage type -> [group][channel 4 kq + r][8 + position 16 nt + j], group by group; on the straight-line path
        //     of interior tiles a group's rows are read back while the next group is converted and stored after it -- the stores leave one
        //     or two at a time between the element-wise work instead of as a burst of eight at the end of the tile.
        // Interior tiles -- whole, inside one sequence, at least two positions into it, inside the convolved length -- take a
        // straight-line path: no per-element predicates, no divisions, addresses from offsets hoisted out of the loop.
        const bool fast = t >= t_begin && p0 + PJ_NT <= P && sl0 >= 2 && sl0 + PJ_NT <= a.Lc;
        auto park = [&](int g) {
            HY_UNROLL
            for (int nt = 0; nt < IP_NT4; ++nt) {
                HY_UNROLL
                for (int r = 0; r < 4; ++r) {
                    HY_LDS elem_t* e = reinterpret_cast<HY_LDS elem_t*>(ebuf + (g * IP_CB + 4 * kq + r) * C::EROW);
                    e[8 + nt * 16 + j] = Elem<DT>::cvt(acc[g][nt][r]);
                }
            }
        };
        if (fast) {
            counted = t + 1 < t_whole;                                       // eight stores, no branch around any of them; the next tile's loads whole
            const size_t xpos = (size_t)sb * (size_t)a.bsx + (size_t)sl0;   // the tile's first position inside an xT row (the tile lies in ONE sequence)
            // (3) xT: 3 x 16 rows x 8 pieces of 8 positions; piece m of a lane: group m >> 1, channel (lane >> 3) + 8 (m & 1), piece lane & 7
            Frag ra, rb;
            HY_UNROLL
            for (int g = 0; g < 3; ++g) {
                park(g);
                if (g > 0) {
                    const size_t rowm = (size_t)((g - 1) * D) * CS + xpos;                            // wave-uniform
                    PJ_ST16(reinterpret_cast<elem_t*>(a.xT) + (xoff0 + rowm), ra);
                    PJ_ST16(reinterpret_cast<elem_t*>(a.xT) + (xoff0 + rowm + (size_t)8 * CS), rb);
                }
                HY_WAVE_SYNC_PJ();
                ra = lds_ld16(ebuf + (g * IP_CB + (lane >> 3)) * C::EROW + 16 + (lane & 7) * 16);
                rb = lds_ld16(ebuf + (g * IP_CB + (lane >> 3) + 8) * C::EROW + 16 + (lane & 7) * 16);
            }
            // (4) vg = shortconv(v) * shortconv(x1): 16 channels x 8 pieces
            const size_t vbase = (size_t)sb * D * a.ldv + (size_t)sl0;
            Frag vf[IP_CB * 8 / 64];
            HY_UNROLL
            for (int m = 0; m < IP_CB * 8 / 64; ++m) {
                const int ch = (lane >> 3) + 8 * m, pc = lane & 7;
                float prod[8];
                HY_UNROLL
                for (int gi = 0; gi < 2; ++gi) {                    // gi = 0: x1 (group 1), gi = 1: v (group 2)
                    const HY_LDS char* row = ebuf + ((1 + gi) * IP_CB + ch) * C::EROW + pc * 16;
                    const Frag lo = lds_ld16(row), hi = lds_ld16(row + 16);
                    elem_t pl[8], ph[8];
                    __builtin_memcpy(pl, lo.w, 16);
                    __builtin_memcpy(ph, hi.w, 16);
                    float xs[10];
                    const HY_LDS float* tp = taps + (gi * IP_CB + ch) * 5;
                    const float w0 = tp[0], w1 = tp[1], w2 = tp[2], bsc = tp[3], bin = tp[4];
                    xs[0] = Elem<DT>::dec(pl[6]) + bin; xs[1] = Elem<DT>::dec(pl[7]) + bin;
                    HY_UNROLL
                    for (int i = 0; i < 8; ++i) xs[2 + i] = Elem<DT>::dec(ph[i]) + bin;
                    HY_UNROLL
                    for (int i = 0; i < 8; ++i) {
                        const float c = __builtin_fmaf(w2, xs[i + 2], __builtin_fmaf(w1, xs[i + 1], __builtin_fmaf(w0, xs[i], bsc)));   // = cm_sc
                        prod[i] = gi == 0 ? c : prod[i] * c;
                    }
                }
                elem_t out[8];
                HY_UNROLL
                for (int i = 0; i < 8; ++i) out[i] = Elem<DT>::cvt(prod[i]);
                __builtin_memcpy(vf[m].w, out, 16);
                if (m == 0) {                                       // the v rows of xT, behind the first half of the window arithmetic
                    const size_t rowm = (size_t)(2 * D) * CS + xpos;
                    PJ_ST16(reinterpret_cast<elem_t*>(a.xT) + (xoff0 + rowm), ra);
                    PJ_ST16(reinterpret_cast<elem_t*>(a.xT) + (xoff0 + rowm + (size_t)8 * CS), rb);
                } else {
                    PJ_ST16(reinterpret_cast<elem_t*>(a.vg) + (vbase + voff0 + (unsigned)(8 * (m - 1)) * (unsigned)a.ldv), vf[m - 1]);
                }
            }
            PJ_ST16(reinterpret_cast<elem_t*>(a.vg) + (vbase + voff0 + (unsigned)(8 * (IP_CB * 8 / 64 - 1)) * (unsigned)a.ldv), vf[IP_CB * 8 / 64 - 1]);
        } else {
            HY_UNROLL
            for (int g = 0; g < 3; ++g) park(g);
            HY_WAVE_SYNC_PJ();
        }
        if (t >= t_begin && !fast) {
            {
                // generic path: ragged tiles, sequence boundaries inside the tile, the first two positions of a sequence,
                // positions beyond the convolved length
                HY_UNROLL
                for (int m = 0; m < 3 * IP_CB * 8 / 64; ++m) {
                    const int g = m >> 1, ch = (lane >> 3) + 8 * (m & 1), pc = lane & 7;
                    const unsigned p = p0 + 8u * (unsigned)pc;
                    const Frag v = lds_ld16(ebuf + (g * IP_CB + ch) * C::EROW + 16 + pc * 16);
                    if (p >= P) continue;
                    const unsigned xb_ = p / (unsigned)a.Lx;                     // the piece's first position: sequence, position within it
                    const int xl = (int)(p - xb_ * (unsigned)a.Lx);
                    elem_t* const crow = reinterpret_cast<elem_t*>(a.xT) + (size_t)(g * D + d0 + ch) * CS;
                    if (xl + 8 <= a.Lx) PJ_ST16(crow + (size_t)xb_ * a.bsx + xl, v);
                    else {                                                       // the piece runs into the next sequence's row (or past the last one)
                        elem_t sv[8];
                        __builtin_memcpy(sv, v.w, 16);
                        for (int i = 0; i < 8; ++i) {
                            int li = xl + i;
                            unsigned bi = xb_;
                            if (li >= a.Lx) { li -= a.Lx; ++bi; }
                            if (p + i < P) crow[(size_t)bi * a.bsx + li] = sv[i];
                        }
                    }
                }
                HY_UNROLL
                for (int m = 0; m < IP_CB * 8 / 64; ++m) {
                    const int ch = (lane >> 3) + 8 * m, pc = lane & 7;
                    const unsigned p = p0 + 8u * (unsigned)pc;
                    if (p >= P) continue;
                    float prod[8];
                    const unsigned b = p / (unsigned)a.Lx;
                    const int l = (int)(p - b * (unsigned)a.Lx);
                    HY_UNROLL
                    for (int gi = 0; gi < 2; ++gi) {
                        const HY_LDS char* row = ebuf + ((1 + gi) * IP_CB + ch) * C::EROW + pc * 16;
                        const Frag lo = lds_ld16(row), hi = lds_ld16(row + 16);
                        elem_t pl[8], ph[8];
                        __builtin_memcpy(pl, lo.w, 16);
                        __builtin_memcpy(ph, hi.w, 16);
                        float xs[10];
                        xs[0] = Elem<DT>::dec(pl[6]); xs[1] = Elem<DT>::dec(pl[7]);
                        HY_UNROLL
                        for (int i = 0; i < 8; ++i) xs[2 + i] = Elem<DT>::dec(ph[i]);
                        const HY_LDS float* tp = taps + (gi * IP_CB + ch) * 5;
                        const float w0 = tp[0], w1 = tp[1], w2 = tp[2], bsc = tp[3], bin = tp[4];
                        HY_UNROLL
                        for (int i = 0; i < 8; ++i) {
                            int li = l + i;                             // position within its sequence (the piece may cross into the next one)
                            if (li >= a.Lx) li -= a.Lx;
                            const float x0 = li >= 2 ? xs[i] + bin : 0.f, x1 = li >= 1 ? xs[i + 1] + bin : 0.f, x2 = xs[i + 2] + bin;
                            const float c = __builtin_fmaf(w2, x2, __builtin_fmaf(w1, x1, __builtin_fmaf(w0, x0, bsc)));   // = cm_sc
                            prod[i] = gi == 0 ? c : prod[i] * c;
                        }
                    }
                    elem_t out[8];
                    HY_UNROLL
                    for (int i = 0; i < 8; ++i) out[i] = Elem<DT>::cvt(prod[i]);
                    elem_t* vrow = reinterpret_cast<elem_t*>(a.vg) + ((size_t)b * D + d0 + ch) * a.ldv;
                    if (l + 8 <= a.Lc) {
                        Frag f;
                        __builtin_memcpy(f.w, out, 16);
                        PJ_ST16(vrow + l, f);
                    } else {
                        for (int i = 0; i < 8; ++i) {
                            int li = l + i;
                            unsigned bi = b;
                            if (li >= a.Lx) { li -= a.Lx; ++bi; }
                            if (p + i < P && li < a.Lc) reinterpret_cast<elem_t*>(a.vg)[((size_t)bi * D + d0 + ch) * a.ldv + li] = out[i];
                        }
                    }
                }
            }
        }
        HY_WAVE_SYNC_PJ();                        // the tile is re-written in the next round
    }
}

// =============================================================================================================================
// The block's MLP (flash_attn.modules.mlp.Mlp; simple_lm.py:191-211, long_conv_lm.py:117-123: fc1 -> tanh-GELU -> fc2): the two
// products whose contraction is d_model (K = 128 / 256) on the same weights-stationary scheme, POSITION-major, with the
// element-wise passes of the reference in their epilogues:
//     mlp_fc1_gelu_kernel :  a = x W1^T + b1   (P, N);   h = gelu_tanh(a)   (P, N)         forward   (replaces GEMM + bias + GELU pass)
//     mlp_dh_dgelu_kernel :  da = (dy W2) * gelu_tanh'(a)   (P, N);   partial column sums of da (-> d b1)
//                                                                                            backward  (replaces GEMM + GELU-backward
//                                                                                                       pass + bias-gradient pass)
// Roles are swapped with respect to the in_proj kernel: positions are the ROWS of the MFMA (A fragments from the staged
// activation tile), the stationary weights are its COLUMNS (B fragments in registers: 64 hidden units x K per wavefront =
// 128 VGPRs at K = 256), so that an accumulator register holds 32 neighbouring hidden units of ONE position across the lanes
// and rows of the position-major outputs leave as 128-byte pieces (through a wavefront-private LDS tile, 16 bytes per lane).
// A workgroup = 4 wavefronts = 256 hidden units over a run of positions; the N / 256 workgroups of a run share an XCD's L2.
// Numerics follow the autocast graph they replace: a is the rounded GEMM result (bias added in fp32 before the one rounding, as
// the library's epilogue does), h = round(gelu(a)) from the ROUNDED a, da = round(round(dh) * gelu'(a)).
// =============================================================================================================================
enum { PM_UW = 64 /* hidden units per wavefront */ };

// Round 4: the operand tile (and MODE 1's a tile) reach LDS by gfx950's LDS-direct 16-byte loads (global_load_lds_dwordx4: no registers, no
// ds_write pass, asynchronous).  An instruction fills 1 KB of LDS lane by lane (wave-uniform base + 16 * lane) from per-lane global
// addresses, so the tiles are unpadded and a bank-conflict-free image is obtained by permuting the SOURCE pieces:
//     operand tile [position][K]:   piece c of position pos at slot c ^ (pos mod PCS)          (PCS = K / 8 pieces per row)
//     wavefront tile [position][64 units]:  piece c at slot c ^ (4 * bit 2 of pos)
// What this buys: tile t + 1 is requested as soon as every wavefront has read tile t's last fragment and lands behind tile t's whole
// epilogue (s_memtime stamps of the round-3 kernel, profiles/r4j_mlp_phases.txt: of 9.5 us per tile and wavefront 2.3 us were spent issuing
// the operand's loads into registers and waiting for them -- 250 of 256 registers hold weights, accumulators and the staged tile, there is
// no room to prefetch through registers).  vmcnt counts loads and stores in issue order: the waits name how many younger operations may
// stay in flight (a tile's own stores), never a blanket vmcnt(0) inside the steady state.
template <int K> struct PmCfg {
    static constexpr int KS = K / 16;
    static constexpr int PCS = K / 8;                           // 16-byte pieces per operand row (32 / 16)
    static constexpr int UROWB = K * 2;                         // bytes per operand row
    static constexpr int UBUF = PJ_NT * UROWB;                  // the operand tile: 32 / 16 KB
    static constexpr int NX = UBUF / 1024 / PJ_WAVES;           // LDS-direct loads per wavefront and operand tile (8 / 4)
    static constexpr int EROW = PM_UW * 2;                      // bytes per row of a wavefront's [position][unit] chunk image
    // a wavefront's LDS: two 1 KB chunk images (8 positions x 64 units), double-buffered; MODE 1: + the tile of a values (64 positions x 64 units)
    static constexpr int IMG = 2048, ATILE = PJ_NT * EROW;
    static constexpr int ebuf(int mode) { return mode == 0 ? IMG : IMG + ATILE; }
    // 40 / 72 KB at K = 256: two workgroups share a CU (2 wavefronts per SIMD, <= 256 registers each); the second one fills the first one's
    // dependent chains (LDS and transcendental latencies), barriers and store back-pressure.
    static constexpr size_t lds(int mode) { return (size_t)UBUF + PJ_WAVES * (size_t)ebuf(mode); }
};

struct MlpArgs {
    const void* x;        // fc1: x (P, K);  dh: dy (P, K)
    const void* W;        // fc1: W1 (N, K);  dh: W2^T (N, K)        [K contiguous]
    const float* bias;    // fc1: b1 (N,) fp32 (already rounded to the element type by the caller) or null
    const void* a_in;     // dh: a (P, N)
    void* o0;             // fc1: a (P, N);  dh: da (P, N)
    void* o1;             // fc1: h (P, N)
    float* part;          // dh: [runs][N] partial column sums of da
    unsigned P;
    int N;
    int tiles, tiles_per_wg;
};

// The tanh approximation PyTorch's F.gelu(approximate="tanh") evaluates (aten/src/ATen/native/cuda/ActivationGeluKernel.cu):
//     gelu(x) = 0.5 x (1 + tanh(z)),  z = sqrt(2 / pi) (x + 0.044715 x^3),  in fp32.
// 0.5 (1 + tanh(z)) = sigmoid(2 z) = 1 / (1 + exp(-2 z)): no cancellation in the negative tail (where 1 + tanh loses its digits), exact
// limits 0 and 1.  exp(-2 z) = exp2(x (c1 + c2 x^2)) with the constants folded (c1 = -2 sqrt(2 / pi) log2(e), c2 = 0.044715 c1): five
// fp32 operations and two transcendentals per value -- the element-wise part is what these kernels' wavefronts spend their time on
// (profiles/r4j_mlp_phases.txt).
#define PM_C1 (-2.302208198f)                 /* -2 * 0.7978845608028654 * 1.4426950408889634 */
#define PM_C2 (-0.1029432396f)                /* 0.044715 * PM_C1 */
__device__ __forceinline__ float pm_exp2(float w) {
#ifdef HIPEMU
    return exp2f(w);
#else
    return __builtin_amdgcn_exp2f(w);                 // v_exp_f32
#endif
}
__device__ __forceinline__ float pm_rcp(float d) {
#ifdef HIPEMU
    return 1.0f / d;
#else
    return __builtin_amdgcn_rcpf(d);                  // v_rcp_f32 (1 ulp): a correctly rounded division costs ten instructions
#endif
}
__device__ __forceinline__ float pm_gelu(float x) {
    const float e = pm_exp2(x * __builtin_fmaf(x * x, PM_C2, PM_C1));
    return x * pm_rcp(1.0f + e);
}
// its derivative: sg + x sg (1 - sg) (2 z'),  sg = sigmoid(2 z),  2 z' = 2 sqrt(2 / pi) (1 + 3 * 0.044715 x^2)
__device__ __forceinline__ float pm_dgelu(float x) {
    const float x2 = x * x;
    const float sg = pm_rcp(1.0f + pm_exp2(x * __builtin_fmaf(x2, PM_C2, PM_C1)));
    const float q = x * __builtin_fmaf(x2, 0.2140644488f /* 6 * 0.044715 * sqrt(2 / pi) */, 1.5957691216f /* 2 sqrt(2 / pi) */);
    return __builtin_fmaf(q * sg, 1.0f - sg, sg);
}

// MODE 0: fc1 + bias + GELU (outputs a, h);  MODE 1: dh = dy W2, da = dh gelu'(a) (+ partial column sums)
template <int K, int DT, int MODE>
__global__ void __launch_bounds__(PJ_THREADS, 2) mlp_kernel(MlpArgs a) {
    typedef PmCfg<K> C;
    typedef typename Elem<DT>::type elem_t;
    HY_SMEM(smem);
    PJ_VMQ_DECL;
    const int tid = (int)threadIdx.x, wave = HY_SGPR(tid >> 6), lane = tid & 63, j = lane & 31, hb = lane >> 5;
    const int N = a.N;
    const unsigned P = a.P;
    const int ncg = N / (PJ_WAVES * PM_UW);
    int cg, run;
    {
        const int wg = blockIdx.x, xcd = wg & 7, seq = wg >> 3;
        cg = seq % ncg;
        run = (seq / ncg) * 8 + xcd;
    }
    const int t_begin = run * a.tiles_per_wg;
    if (t_begin >= a.tiles) return;
    const int t_end = (t_begin + a.tiles_per_wg < a.tiles) ? t_begin + a.tiles_per_wg : a.tiles;
    const int n0 = cg * PJ_WAVES * PM_UW + wave * PM_UW;                   // first hidden unit of this wavefront

    HY_LDS char* const ubuf = HY_LDS_CAST(char, smem);
    HY_LDS char* const et = HY_LDS_CAST(char, smem) + C::UBUF + wave * C::ebuf(MODE);       // this wavefront's chunk images [position][unit] (+ MODE 1: its a tile)

    const char* const xbase = reinterpret_cast<const char*>(a.x);
    auto issue_x = [&](int t, auto full_c) {
        issue_operand_tile<K, decltype(full_c)::value>(xbase, (unsigned)t * PJ_NT, P, ubuf, wave, lane PJ_VMQ_ARG);
    };
    // A wavefront's tile: 16-byte slot S = 64 m + lane = (position 8 m + lane / 8, slot lane mod 8) <-> piece (lane mod 8) ^ 4 hb of the global
    // row (bit 2 of the position is hb for every m).  Same map for the a tile coming in (MODE 1) and the results going out.
    const uint32_t eoff0 = (uint32_t)(((lane >> 3) * N + n0 + 8 * ((lane & 7) ^ (hb << 2))) * 2);          // bytes; + (p0 + 8 m) N 2: wave-uniform
    // MODE 1: the a values arrive as the same row pieces by LDS-direct loads into a wavefront-private tile (slot 64 m + lane = the piece this
    // lane turns into a piece of da), half a tile (chunks 4 h .. 4 h + 3) at a time: a half is requested as soon as the previous tile's row
    // phases have read it out, a whole half-tile of epilogue before it is needed.  (They first came straight into registers, two chunks
    // ahead; the queue retires in order, so every wait for such a load also waited for the next tile's operand loads issued before it --
    // the emulator's queue model showed the operand prefetch being drained two chunks into the epilogue.)
    HY_LDS char* const atile = et + C::IMG;
    auto issue_a_half = [&](int t, int h, auto full_c) {
        constexpr bool FULL = decltype(full_c)::value;
        const unsigned p0 = (unsigned)t * PJ_NT;
        HY_UNROLL
        for (int m = 4 * h; m < 4 * h + 4; ++m) {
            const char* const rb = HY_UNIFORM_PTR(const char, reinterpret_cast<const char*>(a.a_in) + ((size_t)p0 + 8u * m) * N * 2);
            if (FULL || p0 + (unsigned)(lane >> 3) + 8u * m < P) glds16(rb, eoff0, atile + m * 1024, lane PJ_VMQ_ARG);
        }
    };
    const int t_whole = (int)(P / PJ_NT) < t_end ? (int)(P / PJ_NT) : t_end;                 // tiles [t_begin, t_whole) are whole
    if (t_begin < t_whole) { issue_x(t_begin, True_()); if (MODE == 1) { issue_a_half(t_begin, 0, True_()); issue_a_half(t_begin, 1, True_()); } }
    else { issue_x(t_begin, False_()); if (MODE == 1) { issue_a_half(t_begin, 0, False_()); issue_a_half(t_begin, 1, False_()); } }

    // stationary operand: 64 weight rows as B fragments (column = unit j of unit tile ut, k = 16 ks + 8 hb ...)
    Frag wf[2][C::KS];
    HY_UNROLL
    for (int ut = 0; ut < 2; ++ut) {
        const char* row = reinterpret_cast<const char*>(a.W) + ((size_t)(n0 + ut * 32 + j) * K + 8 * hb) * 2;
        HY_UNROLL
        for (int ks = 0; ks < C::KS; ++ks) wf[ut][ks] = ld16(row + ks * 32);
    }
    float bias[2] = {0.f, 0.f};
    if (MODE == 0 && a.bias != nullptr) { bias[0] = a.bias[n0 + j]; bias[1] = a.bias[n0 + 32 + j]; }
    float colsum[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};                 // MODE 1: units n0 + 8 ((lane & 7) ^ 4 hb) + i of the positions this lane's row pieces cover

    // A fragment of (position tile pt, step ks): 16 bytes at row pt 32 + j, piece (2 ks + hb) ^ (j mod PCS) = (2 ks) ^ (hb ^ j mod PCS)
    const int ua = j * C::UROWB, ux = (hb ^ (j % C::PCS)) * 16;

#if defined(PJ_PROFILE) && !defined(HIPEMU)
    unsigned long long pj_d[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, pj_t, pj_start, pj_rt0, pj_rt1;
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(pj_rt0)::"memory");
    PJ_NOW(pj_t);
    pj_start = pj_t;
#endif
    // one tile; FULL (compile time): whole tiles carry no per-position predicate and no branch around a memory instruction, so that the
    // number of operations in flight behind a given one is known
    // FULL: a whole tile; NEXT: the next tile exists and is whole (both compile time: see above)
    auto tile = [&](int t, auto full_c, auto next_c) {
        constexpr bool full = decltype(full_c)::value, next_full = decltype(next_c)::value;
        const unsigned p0 = (unsigned)t * PJ_NT;
        // (1) this tile's operand is in LDS: my share has landed (younger: the previous tile's 16 stores -- MODE 1: 8 stores and the 8 loads
        //     of this tile's a values), then everybody's
        if (t == t_begin || !full) PJ_VMWAIT(0);
        else PJ_VMWAIT(16);
        PJ_MARK(0);
        PJ_BARRIER();
        PJ_MARK(1);
        acc_t acc[2][2];                           // [position tile][unit tile]
        HY_UNROLL
        for (int pt = 0; pt < 2; ++pt) {
            HY_UNROLL
            for (int ut = 0; ut < 2; ++ut) {
                HY_UNROLL
                for (int r = 0; r < 16; ++r) acc[pt][ut][r] = 0.f;
            }
        }
        HY_UNROLL
        for (int ks = 0; ks < C::KS; ++ks) {
            HY_UNROLL
            for (int pt = 0; pt < 2; ++pt) {
                const Frag af = lds_ld16(ubuf + pt * 32 * C::UROWB + ua + (ux ^ (ks * 32)));
                HY_UNROLL
                for (int ut = 0; ut < 2; ++ut) acc[pt][ut] = mfma<DT>(af, wf[ut][ks], acc[pt][ut]);
            }
        }
        PJ_MARK(2);
        // (2) every wavefront has read its last fragment: the next tile may land, behind this tile's epilogue
        PJ_BARRIER();
        PJ_MARK(3);
        if (next_full) issue_x(t + 1, True_());
        else if (t + 1 < t_end) issue_x(t + 1, False_());
        PJ_MARK(4);
        // ---- epilogue, wavefront-private, in 8 chunks of 8 positions x 64 units (1 KB = one 16-byte row piece per lane) ----
        // Register r of lane (j, hb) = position pt 32 + pj_row(r, hb), unit ut 32 + j: chunk m = positions 8 m .. 8 m + 7 is registers
        // 4 (m mod 4) + i, i < 4, of position tile m / 4, both unit tiles -- 8 values per lane.  The GEMM result of a chunk (MODE 0: a =
        // round(acc + bias); MODE 1: round(dh)) is parked in LDS in the [position][unit] image (byte q 128 + (ut ^ hb) 64 + 2 j, q = i + 4 hb)
        // and read back as row pieces: from there on a lane holds 8 neighbouring units of ONE position, exactly what one 16-byte piece of
        // the outputs (and of MODE 1's a operand) is, and the element-wise part runs in that layout -- eight independent chains per lane,
        // nothing but the GEMM result ever crosses LDS (the first chunked version parked h as well and MODE 1 picked its a values out of
        // an LDS tile one 2-byte read at a time, each with its latency exposed: 30 % of a wavefront's life, profiles/r4l_mlp_sq_counters.csv).
        // Row phase r = m - 1 runs between chunk m's parking and the request for its rows, which are used behind the parking of chunk m + 1: the
        // read-back's latency and the stores' back-pressure hide behind arithmetic.
        {
            Frag rp;
            rp.w[0] = rp.w[1] = rp.w[2] = rp.w[3] = 0u;
            char* const dst0 = reinterpret_cast<char*>(a.o0) + (size_t)p0 * N * 2 + eoff0;
            char* const dst1 = MODE == 0 ? reinterpret_cast<char*>(a.o1) + (size_t)p0 * N * 2 + eoff0 : nullptr;
            HY_UNROLL
            for (int m = 0; m <= 8; ++m) {
                if (m < 8) {
                    HY_LDS char* const img = et + (m & 1) * 1024;
                    HY_UNROLL
                    for (int ut = 0; ut < 2; ++ut) {
                        HY_UNROLL
                        for (int i = 0; i < 4; ++i) {
                            const float v = acc[m >> 2][ut][4 * (m & 3) + i];
                            HY_LDS elem_t* slot = reinterpret_cast<HY_LDS elem_t*>(img + (i + 4 * hb) * C::EROW + (ut ^ hb) * 64 + 2 * j);
                            *slot = Elem<DT>::cvt(MODE == 0 ? v + bias[ut] : v);                     // MODE 1: the rounding of the unfused dh tensor
                        }
                    }
                    HY_WAVE_SYNC_PJ();
                }
                if (m > 0) {
                    const int r = m - 1;
                    const bool on = full || p0 + (unsigned)(lane >> 3) + 8u * r < P;
                    elem_t ge[8], oe[8];
                    __builtin_memcpy(ge, rp.w, 16);
                    if (MODE == 0) {
                        HY_UNROLL
                        for (int i = 0; i < 8; ++i) oe[i] = Elem<DT>::cvt(pm_gelu(Elem<DT>::dec(ge[i])));
                        Frag rh;
                        __builtin_memcpy(rh.w, oe, 16);
                        if (on) {
                            PJ_ST16(dst0 + (size_t)(8 * r) * N * 2, rp);
                            PJ_ST16(dst1 + (size_t)(8 * r) * N * 2, rh);
                        }
                    } else {
                        // the a values of this half tile have landed (requested half a tile of epilogue ago; younger in the queue: 4 stores, the 4
                        // loads of the other half and the next tile's operand loads)
                        if ((r & 3) == 0) {
                            if (!full || !next_full) PJ_VMWAIT(0);
                            else if (C::NX == 8) PJ_VMWAIT(16);
                            else PJ_VMWAIT(12);
                        }
                        const Frag af = lds_ld16(atile + r * 1024 + lane * 16);
                        elem_t ae[8];
                        __builtin_memcpy(ae, af.w, 16);
                        Frag rd;
                        HY_UNROLL
                        for (int i = 0; i < 8; ++i) {
                            oe[i] = Elem<DT>::cvt(Elem<DT>::dec(ge[i]) * pm_dgelu(Elem<DT>::dec(ae[i])));
                            if (on) colsum[i] += Elem<DT>::dec(oe[i]);
                            if (i & 1) rd.w[i >> 1] = (uint32_t)oe[i - 1] | ((uint32_t)oe[i] << 16);      // packed as they come: eight loose 16-bit results and the fp16 kernel spills
                            if (i == 3) HY_SCHED_FENCE();          // four chains at a time
                        }
                        if (on) PJ_ST16(dst0 + (size_t)(8 * r) * N * 2, rd);
                        if ((r & 3) == 3) {                                  // this half of the a tile is read out: the next tile's may land in it
                            PJ_LGKMWAIT();
                            if (next_full) issue_a_half(t + 1, r >> 2, True_());
                            else if (t + 1 < t_end) issue_a_half(t + 1, r >> 2, False_());
                        }
                    }
                }
                // chunk m's rows: requested now, used behind the parking of chunk m + 1
                if (m < 8) rp = lds_ld16(et + (m & 1) * 1024 + lane * 16);
                HY_WAVE_SYNC_PJ();
            }
            PJ_MARK(5);                                          // accumulators (waits for the matrix cores) -> element-wise -> rows stored
        }
    };
    {
        int t = t_begin;
        for (; t + 1 < t_whole; ++t) tile(t, True_(), True_());
        if (t < t_whole) { tile(t, True_(), False_()); ++t; }                  // the last whole tile: nothing, or a ragged tile, behind it
        if (t < t_end) tile(t, False_(), False_());
    }
#if defined(PJ_PROFILE) && !defined(HIPEMU)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    PJ_MARK(9);                                                  // last stores acknowledged
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(pj_rt1)::"memory");
    if (pj_prof_buf != nullptr && lane == 0) {
        unsigned long long* o = pj_prof_buf + ((size_t)blockIdx.x * PJ_WAVES + wave) * 16;
        HY_UNROLL
        for (int i = 0; i < 10; ++i) o[i] = pj_d[i];
        o[10] = (unsigned long long)(t_end - t_begin);
        o[11] = pj_t - pj_start;
        o[12] = pj_rt1 - pj_rt0;
        unsigned hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        o[13] = ((unsigned long long)xcc << 32) | hw;
        o[14] = pj_start;
    }
#endif
    if (MODE == 1) {
        // column sums of da over this run: the 8 lanes that share a piece index hold different positions of the same 8 units -- through
        // LDS ([row lane >> 3][unit]: 2 KB), added in row order
        HY_LDS float* red = reinterpret_cast<HY_LDS float*>(et);
        HY_WAVE_SYNC_PJ();
        HY_UNROLL
        for (int i = 0; i < 8; ++i) red[(lane >> 3) * 64 + 8 * ((lane & 7) ^ (hb << 2)) + i] = colsum[i];
        HY_WAVE_SYNC_PJ();
        float sum = 0.f;
        HY_UNROLL
        for (int q = 0; q < 8; ++q) sum += red[q * 64 + lane];
        a.part[(size_t)run * N + n0 + lane] = sum;
    }
}

// =============================================================================================================================
// Column sums of a position-major 16-bit matrix (P, N) in fp32: the bias gradient of a linear layer, d b = sum over the positions
// of d y (out_proj, fc2: projection.py, lm.py).  One streaming pass with 16-byte loads at the memory rate (the generic reduction
// kernel PyTorch picks for this shape reaches a third of it); per-workgroup partial sums, added in workgroup order by a second
// small kernel: bitwise reproducible, no atomics.
// =============================================================================================================================
struct ColsumArgs {
    const void* x;       // (P, N) 16-bit
    float* part;         // [G][N]
    float* out;          // (N,)
    unsigned P;
    int N, G, rows_per_wg;
};

template <int DT>
__global__ void __launch_bounds__(PJ_THREADS) colsum_kernel(ColsumArgs a) {
    typedef typename Elem<DT>::type elem_t;
    HY_SMEM(smem);
    const int tid = (int)threadIdx.x;
    const int cpr = a.N / 8;                              // 16-byte pieces per row; divides 256 (checked by the host)
    const int pc = tid % cpr, ro = tid / cpr, rstep = PJ_THREADS / cpr;
    const unsigned r0 = (unsigned)blockIdx.x * (unsigned)a.rows_per_wg;
    const unsigned r1 = r0 + (unsigned)a.rows_per_wg < a.P ? r0 + (unsigned)a.rows_per_wg : a.P;
    float acc[8];
    HY_UNROLL
    for (int i = 0; i < 8; ++i) acc[i] = 0.f;
    const elem_t* base = reinterpret_cast<const elem_t*>(a.x) + 8 * pc;
    unsigned r = r0 + (unsigned)ro;
    // eight rows' loads in flight per thread, added in row order (one load per trip left the pass at 3.4 TB/s: 158 us for 0.54 GB)
    for (; r + 7u * (unsigned)rstep < r1; r += 8u * (unsigned)rstep) {
        Frag f[8];
        HY_UNROLL
        for (int u = 0; u < 8; ++u) f[u] = ld16(base + (size_t)(r + (unsigned)(u * rstep)) * a.N);
        HY_UNROLL
        for (int u = 0; u < 8; ++u) {
            elem_t e[8];
            __builtin_memcpy(e, f[u].w, 16);
            HY_UNROLL
            for (int i = 0; i < 8; ++i) acc[i] += Elem<DT>::dec(e[i]);
        }
    }
    for (; r < r1; r += (unsigned)rstep) {
        const Frag f = ld16(base + (size_t)r * a.N);
        elem_t e[8];
        __builtin_memcpy(e, f.w, 16);
        HY_UNROLL
        for (int i = 0; i < 8; ++i) acc[i] += Elem<DT>::dec(e[i]);
    }
    // the rstep threads of a column piece add up through LDS in a fixed order
    HY_LDS float* red = HY_LDS_CAST(float, smem);
    HY_UNROLL
    for (int i = 0; i < 8; ++i) red[(ro * cpr + pc) * 8 + i] = acc[i];
    __syncthreads();
    if (ro == 0) {
        for (int o = 1; o < rstep; ++o) {
            HY_UNROLL
            for (int i = 0; i < 8; ++i) acc[i] += red[(o * cpr + pc) * 8 + i];
        }
        HY_UNROLL
        for (int i = 0; i < 8; ++i) a.part[(size_t)blockIdx.x * a.N + 8 * pc + i] = acc[i];
    }
}

// 64 columns per workgroup; the G partial rows are split over the workgroup's 4 thread rows (each adds its rows in order), the four
// sums are added in thread-row order: a fixed order, G / 4 loads per thread
__global__ void __launch_bounds__(PJ_THREADS) colsum_final_kernel(ColsumArgs a) {
    HY_SMEM(smem);
    const int col = (int)blockIdx.x * 64 + ((int)threadIdx.x & 63), q = (int)threadIdx.x >> 6;
    const int per = (a.G + 3) / 4;
    float acc = 0.f;
    if (col < a.N) {
        const int g1 = (q + 1) * per < a.G ? (q + 1) * per : a.G;
        int g = q * per;
        for (; g + 8 <= g1; g += 8) {                                // eight independent loads in flight, added in the same order
            float v[8];
            HY_UNROLL
            for (int u = 0; u < 8; ++u) v[u] = a.part[(size_t)(g + u) * a.N + col];
            HY_UNROLL
            for (int u = 0; u < 8; ++u) acc += v[u];
        }
        for (; g < g1; ++g) acc += a.part[(size_t)g * a.N + col];
    }
    HY_LDS float* red = HY_LDS_CAST(float, smem);
    red[threadIdx.x] = acc;
    __syncthreads();
    if (q == 0 && col < a.N) a.out[col] = ((red[threadIdx.x] + red[threadIdx.x + 64]) + red[threadIdx.x + 128]) + red[threadIdx.x + 192];
}


// =============================================================================================================================
// The output projection with the second gate on its operand load (round 4).  hyena.py:432-440 (order 2):
//     y = y * x[0]                                   (b, d, l) element-wise, x[0] = short_filter(in_proj(u))[0:D]      hyena.py:432
//     y = rearrange(y, 'b d l -> b l d');  y = self.out_proj(y)                                                        hyena.py:439-440
// Until round 4: cm_post_fwd (cm_kernels.h) wrote zT = y x0 channel-major and a library GEMM read it back.  Here ONE kernel reads the
// long convolution's output y (B, D, L) and the x0 third of xT (3D, B, Lx) -- both rows contiguous along l -- and writes
// out (B, L, N) = z W^T + b; zT (D, B, L) is written on the side only when the caller keeps it for the weight gradient.
//
// The product contracts over the channels, the operands arrive position-contiguous: the 64 x K tile of z has to be transposed on its
// way to the matrix cores.  A wavefront fetches 8 channel rows x 128 bytes per access (lane = (row r, 16-byte piece c): every access
// is 8 whole 128-byte lines), forms z = round(y * shortconv(x0)) for its 8 positions (the arithmetic of cm_post_fwd, FMA by FMA), swaps
// half of them with the lane holding the neighbouring channel (lane ^ 8) so that it owns (k, k + 1) pairs for 4 positions, and writes
// 4-byte pairs into the LDS tile.  Tile layout: chunk-major, swizzled --
//     element (pos, k) at byte  (k / 8) * 1024 + (pos ^ (pos >> 3)) * 16 + (k % 8) * 2
// i.e. for every 8 channels one 1 KB block of 64 sixteen-byte slots, the slot of a position XOR-ed with its octet: an MFMA A-fragment
// (8 consecutive channels of one position) is ONE aligned 16-byte read, conflict-free for the 16-lane groups of ds_read_b128 (checked
// group by group for both 32-position halves), and the 4-byte writes of a wavefront hit every bank exactly twice (free for ds_write_b32).
// No padding: 64 x K x 2 bytes.  The rest is the weights-stationary scheme of mlp_kernel: a wavefront keeps N / 4 output channels x K
// weights as B fragments, the four wavefronts share the z tile, accumulators leave through a wavefront-private [position][channel]
// tile as 16-byte row pieces.  76 KB of LDS at K = 256: two workgroups per CU.
// =============================================================================================================================
#ifndef OP_NB256
#define OP_NB256 4
#endif
template <int K> struct OpCfg {
    static_assert(K == 128 || K == 256, "d_model of the HyenaDNA models");
    static constexpr int KS = K / 16;
    static constexpr int UW = K / PJ_WAVES;                   // output channels per wavefront (N = K)
    static constexpr int UT = UW / 32;                        // 32-channel MFMA tiles per wavefront
    static constexpr int RND = K / 32;                        // staging rounds: 4 wavefronts x 8 rows each
    static constexpr int NB = K == 256 ? OP_NB256 : 2;        // ... fetched and transposed in NB batches (register budget, see the kernel)
    static constexpr int ZBUF = PJ_NT * K * 2;                // the z tile
    static constexpr int TAPB = K * 8 * 4;                    // per channel 8 floats: w0 w1 w2 b_sc b_in - - -
    static constexpr int EROW = (UW + 8) * 2;                 // epilogue tile row [position][channel], +16 bytes: conflict-free 16-byte reads
    static constexpr int EBUF = 32 * EROW;                    // 32 positions at a time
    static constexpr int PCS = UW / 8;                        // 16-byte pieces per position and wavefront
    static constexpr size_t LDS = (size_t)ZBUF + TAPB + PJ_WAVES * (size_t)EBUF;
    // LN = true (the residual add + LayerNorm in the epilogue): the four wavefronts park their 32 x UW blocks in ONE [32 positions][K] tile per
    // half tile (rows padded by 16 B: the halves of a wavefront land 16 banks apart), double-buffered so that one workgroup barrier per half
    // tile suffices; a wavefront then owns 8 whole rows of it
    static constexpr int SROW = (K + 8) * 2;
    static constexpr int SBUF = 32 * SROW;
    static constexpr size_t LDS_LN = (size_t)ZBUF + TAPB + 2 * (size_t)SBUF;
};

struct OutProjArgs {
    const void* y;        // (B, D, L) long-convolution output
    const void* xT;       // (3D, B, Lx) in_proj output without its bias; rows [0, D) are used
    const float* bin;     // (3D,) in_proj bias or null
    const float* w;       // (3D, 3) short-filter taps
    const float* b;       // (3D,) short-filter bias
    const void* W;        // (N, K) out_proj weight, N = K = D
    const float* bias;    // (N,) fp32 (values already rounded to the element type) or null
    void* out;            // (B, L, N)
    void* zT;             // (D, B, L) or null
    // LN = true: out (B, L, N) = LayerNorm(residual'), residual' = round(z W^T + bias) + residual_in  (block_kernels.h's add_norm_fwd_kernel, bit for bit)
    const float* res_in;  // (B L, N) fp32 or null
    const float* ln_w;    // (N,)
    const float* ln_b;    // (N,)
    float* res_out;       // (B L, N) fp32: residual'
    float* mean;          // (B L,)
    float* rstd;          // (B L,)
    float eps;
    int B, L, Lx, D;
    long csx; int bsx;    // xT: row (c, b) at c csx + b bsx
    long csz; int bsz;    // zT: row (d, b) at d csz + b bsz
    int lda;              // y: row (b, d) at (b D + d) lda
    int tiles_per_seq, tiles, tiles_per_wg;
};

__device__ __forceinline__ int op_slot(int pos) { return pos ^ (pos >> 3); }

template <int K, int DT, bool LN = false>
__global__ void __launch_bounds__(PJ_THREADS, 2) outproj_gate_fwd_kernel(OutProjArgs a) {
    typedef OpCfg<K> C;
    typedef typename Elem<DT>::type elem_t;
    HY_SMEM(smem);
    const int tid = (int)threadIdx.x, wave = tid >> 6, lane = tid & 63, j = lane & 31, hb = lane >> 5;
    const int r = lane >> 3, c = lane & 7;                    // staging role: row of the round, 16-byte piece (8 positions)
    const int run = blockIdx.x;
    const int t_begin = run * a.tiles_per_wg;
    if (t_begin >= a.tiles) return;
    const int t_end = (t_begin + a.tiles_per_wg < a.tiles) ? t_begin + a.tiles_per_wg : a.tiles;
    const int n0 = wave * C::UW;                              // first output channel of this wavefront
    const int N = K;

    HY_LDS char* const zt = HY_LDS_CAST(char, smem);
    HY_LDS float* const taps = reinterpret_cast<HY_LDS float*>(HY_LDS_CAST(char, smem) + C::ZBUF);
    HY_LDS char* const et = HY_LDS_CAST(char, smem) + C::ZBUF + C::TAPB + wave * C::EBUF;

    // short-filter taps of the x0 channels -> LDS (read per round below; 5 of 8 floats per channel)
    for (int k = tid; k < K; k += PJ_THREADS) {
        taps[k * 8 + 0] = a.w[k * 3]; taps[k * 8 + 1] = a.w[k * 3 + 1]; taps[k * 8 + 2] = a.w[k * 3 + 2];
        taps[k * 8 + 3] = a.b[k];
        taps[k * 8 + 4] = a.bin != nullptr ? a.bin[k] : 0.f;
        if constexpr (LN) { taps[k * 8 + 5] = a.ln_w[k]; taps[k * 8 + 6] = a.ln_b[k]; }      // (N = K: the LayerNorm's weight / bias ride in the spare slots)
    }
    // stationary operand: UW weight rows as B fragments (column = output channel j of tile ut, k = 16 ks + 8 hb ...)
    Frag wf[C::UT][C::KS];
    HY_UNROLL
    for (int ut = 0; ut < C::UT; ++ut) {
        const char* row = reinterpret_cast<const char*>(a.W) + ((size_t)(n0 + ut * 32 + j) * K + 8 * hb) * 2;
        HY_UNROLL
        for (int ks = 0; ks < C::KS; ++ks) wf[ut][ks] = ld16(row + ks * 32);
    }
    float bias[C::UT];
    HY_UNROLL
    for (int ut = 0; ut < C::UT; ++ut) bias[ut] = a.bias != nullptr ? a.bias[n0 + ut * 32 + j] : 0.f;

    const elem_t* const yb = reinterpret_cast<const elem_t*>(a.y);
    const elem_t* const xb = reinterpret_cast<const elem_t*>(a.xT);
    elem_t* const zb = reinterpret_cast<elem_t*>(a.zT);
    elem_t* const ob = reinterpret_cast<elem_t*>(a.out);

    for (int t = t_begin; t < t_end; ++t) {
        // a sequence's last tile is pulled back to end at L when L is not a multiple of 64: it recomputes (and rewrites, identically) up to
        // 63 positions of its neighbour instead of taking a ragged path -- every tile is whole, and rows may start at any even byte offset
        const int b = t / a.tiles_per_seq, l0raw = (t - b * a.tiles_per_seq) * PJ_NT;
        const int l0 = l0raw + PJ_NT <= a.L ? l0raw : a.L - PJ_NT;
        const int lp = l0 + 8 * c;                                    // this lane's first position
        // The rounds go in NB batches (fetch a batch, transpose it into the tile, fetch the next): all 8 rounds of K = 256 in flight at
        // once are 72 registers next to the 128 that hold the weights -- hipcc spilled 61 - 126 registers; 14 with two batches, none
        // with four.  The CU's other workgroup covers the extra round trips.
        HY_UNROLL
        for (int half = 0; half < C::NB; ++half) {
            constexpr int HR = C::RND / C::NB;
            // ---- global -> registers: y and x0 pieces of this half's rounds (+ the two positions before the tile for piece 0) ----
            Frag yr[HR], xr[HR];
            uint32_t halo[HR];
            HY_UNROLL
            for (int ii = 0; ii < HR; ++ii) {
                const int i = half * HR + ii;
                const int k = 32 * i + 8 * wave + r;
                const elem_t* yrow = yb + ((size_t)b * a.D + k) * a.lda;
                const elem_t* xrow = xb + ((size_t)k * (size_t)a.csx + (size_t)b * a.bsx);
                yr[ii] = ld16(yrow + lp);                           // (every tile is whole; gfx950 global memory takes under-aligned 16-byte accesses)
                xr[ii] = ld16(xrow + lp);
                uint32_t h2 = 0u;
                if (c == 0 && l0 >= 2) __builtin_memcpy(&h2, xrow + (l0 - 2), 4);
                else if (c == 0 && l0 == 1) { uint16_t h1; __builtin_memcpy(&h1, xrow, 2); h2 = (uint32_t)h1 << 16; }   // L = 65: the pulled-back tile starts at 1
                halo[ii] = h2;
            }
            if (half == 0) __syncthreads();                          // every wavefront is done with the previous z tile (first tile: the taps are in LDS)
            // ---- z = round(y * shortconv(x0)) -> pairs of channels -> the swizzled tile ----
            HY_UNROLL
            for (int ii = 0; ii < HR; ++ii) {
                const int i = half * HR + ii;
                const int k = 32 * i + 8 * wave + r;
                const uint32_t prev = HY_SHFL_U32(xr[ii].w[3], lane - 1);        // positions lp - 2, lp - 1 (lane - 1 = same row, piece c - 1)
                const uint32_t hw = c == 0 ? halo[ii] : prev;
                elem_t xe[10], ye[8];
                __builtin_memcpy(xe, &hw, 4);
                __builtin_memcpy(xe + 2, xr[ii].w, 16);
                __builtin_memcpy(ye, yr[ii].w, 16);
                const float w0 = taps[k * 8], w1 = taps[k * 8 + 1], w2 = taps[k * 8 + 2], bsc = taps[k * 8 + 3], bin = taps[k * 8 + 4];
                elem_t ze[8];
                HY_UNROLL
                for (int e = 0; e < 8; ++e) {
                    const int l = lp + e;
                    const float x0 = l >= 2 ? Elem<DT>::dec(xe[e]) + bin : 0.f, x1 = l >= 1 ? Elem<DT>::dec(xe[e + 1]) + bin : 0.f,
                                x2 = Elem<DT>::dec(xe[e + 2]) + bin;
                    const float c0 = __builtin_fmaf(w2, x2, __builtin_fmaf(w1, x1, __builtin_fmaf(w0, x0, bsc)));   // = cm_sc
                    ze[e] = Elem<DT>::cvt(Elem<DT>::dec(ye[e]) * c0);                                                  // = cm_post_fwd
                }
                Frag zp;
                __builtin_memcpy(zp.w, ze, 16);
                if (zb != nullptr) st16(zb + ((size_t)k * (size_t)a.csz + (size_t)b * a.bsz) + lp, zp);
                // lanes r (even) and r + 1 (= lane ^ 8) hold channels k, k + 1 for the same 8 positions: the even one takes positions
                // 0..3 of both, the odd one positions 4..7
                const bool odd = (r & 1) != 0;
                const uint32_t s0 = odd ? zp.w[0] : zp.w[2], s1 = odd ? zp.w[1] : zp.w[3];
                const uint32_t g0 = HY_SHFL_U32(s0, lane ^ 8), g1 = HY_SHFL_U32(s1, lane ^ 8);
                const uint32_t m0 = odd ? zp.w[2] : zp.w[0], m1 = odd ? zp.w[3] : zp.w[1];      // my own channel's 4 positions
                const uint32_t lo0 = odd ? g0 : m0, lo1 = odd ? g1 : m1;                         // even channel (k & ~1)
                const uint32_t hi0 = odd ? m0 : g0, hi1 = odd ? m1 : g1;                         // odd channel
                const uint32_t pr[4] = {(lo0 & 0xffffu) | (hi0 << 16), (lo0 >> 16) | (hi0 & 0xffff0000u),
                                        (lo1 & 0xffffu) | (hi1 << 16), (lo1 >> 16) | (hi1 & 0xffff0000u)};
                const int kk = k & ~1, pp = 8 * c + (odd ? 4 : 0);
                HY_LDS char* const base = zt + (kk >> 3) * 1024 + (kk & 7) * 2;
                HY_UNROLL
                for (int e = 0; e < 4; ++e) *reinterpret_cast<HY_LDS uint32_t*>(base + op_slot(pp + e) * 16) = pr[e];
                HY_SCHED_FENCE();                                   // one round at a time: interleaved, their temporaries spill
            }
        }
        __syncthreads();
        // ---- out tile = z W^T on the matrix cores, 32 positions at a time (both halves' accumulators at once -- 64 registers next
        //      to the 128 of the weights -- made hipcc spill a quarter of the weights) ----
        HY_UNROLL
        for (int pt = 0; pt < 2; ++pt) {
            acc_t acc[C::UT];
            HY_UNROLL
            for (int ut = 0; ut < C::UT; ++ut) {
                HY_UNROLL
                for (int q = 0; q < 16; ++q) acc[ut][q] = 0.f;
            }
            HY_UNROLL
            for (int ks = 0; ks < C::KS; ++ks) {
                const Frag af = lds_ld16(zt + (2 * ks + hb) * 1024 + op_slot(pt * 32 + j) * 16);
                HY_UNROLL
                for (int ut = 0; ut < C::UT; ++ut) acc[ut] = mfma<DT>(af, wf[ut][ks], acc[ut]);
            }
            if constexpr (LN) {
                // ---- the residual add + LayerNorm of the block, on whole rows (round 5): the out_proj output never reaches memory.  The four
                //      wavefronts' rounded blocks meet in ONE [32][K] tile (buffer pt: the other buffer may still be read by a slower wavefront);
                //      after the barrier a wavefront owns positions 8 wave .. 8 wave + 7, lane l channels E l .. E l + E - 1 -- the layout and,
                //      operation for operation, the arithmetic of add_norm_fwd_kernel (block_kernels.h): residual', out, mean and rstd are its bits.
                constexpr int E = K / 64;
                HY_LDS char* const sh = HY_LDS_CAST(char, smem) + C::ZBUF + C::TAPB + pt * C::SBUF;
                HY_UNROLL
                for (int ut = 0; ut < C::UT; ++ut) {
                    HY_UNROLL
                    for (int q = 0; q < 16; ++q) {
                        const int pos = pj_row(q, hb), un = n0 + ut * 32 + j;
                        *(reinterpret_cast<HY_LDS elem_t*>(sh + pos * C::SROW) + un) = Elem<DT>::cvt(acc[ut][q] + bias[ut]);
                    }
                }
                __syncthreads();
                const int c0 = lane * E;
                const float inv_d = 1.f / (float)K;
                constexpr int RB = 2;                            // rows per batch: their residual loads in flight together (four: 49 spilled registers)
                for (int g4 = 0; g4 < 8 / RB; ++g4) {
                    float r[RB][E];
                    HY_UNROLL
                    for (int i = 0; i < RB; ++i) {
                        const int pos = 8 * wave + RB * g4 + i;
                        const size_t off = ((size_t)b * a.L + l0 + pt * 32 + pos) * N + c0;
                        if (a.res_in != nullptr) blk_load<DT_F32, E>(a.res_in, off, r[i]);
                        else {
                            HY_UNROLL
                            for (int e = 0; e < E; ++e) r[i][e] = 0.f;
                        }
                    }
                    HY_UNROLL
                    for (int i = 0; i < RB; ++i) {
                        const int pos = 8 * wave + RB * g4 + i;
                        const size_t row = (size_t)b * a.L + l0 + pt * 32 + pos, off = row * N + c0;
                        elem_t xe[E];
                        __builtin_memcpy(xe, sh + pos * C::SROW + c0 * 2, sizeof(xe));
                        float v[E];
                        HY_UNROLL
                        for (int e = 0; e < E; ++e) v[e] = Elem<DT>::dec(xe[e]);
                        if (a.res_in != nullptr) {
                            HY_UNROLL
                            for (int e = 0; e < E; ++e) v[e] += r[i][e];
                        }
                        float sm = 0.f;
                        HY_UNROLL
                        for (int e = 0; e < E; ++e) sm += v[e];
                        const float mean = wave_sum(sm) * inv_d;
                        float vs = 0.f;
                        HY_UNROLL
                        for (int e = 0; e < E; ++e) vs += (v[e] - mean) * (v[e] - mean);
                        const float rstd = 1.f / sqrtf(wave_sum(vs) * inv_d + a.eps);
                        float o[E];
                        HY_UNROLL
                        for (int e = 0; e < E; ++e) o[e] = (v[e] - mean) * rstd * taps[(c0 + e) * 8 + 5] + taps[(c0 + e) * 8 + 6];
                        blk_store<DT, E>(a.out, off, o);
                        blk_store<DT_F32, E>(a.res_out, off, v);
                        if (lane == 0) { a.mean[row] = mean; a.rstd[row] = rstd; }
                    }
                }
                HY_SCHED_FENCE();
            } else {
            // epilogue, wavefront-private: register q of lane (j, hb) = position pt 32 + pj_row(q, hb), channel ut 32 + j
            HY_WAVE_SYNC_PJ();
            HY_UNROLL
            for (int ut = 0; ut < C::UT; ++ut) {
                HY_UNROLL
                for (int q = 0; q < 16; ++q) {
                    const int pos = pj_row(q, hb), un = ut * 32 + j;
                    *(reinterpret_cast<HY_LDS elem_t*>(et + pos * C::EROW) + un) = Elem<DT>::cvt(acc[ut][q] + bias[ut]);
                }
            }
            HY_WAVE_SYNC_PJ();
            HY_UNROLL
            for (int m = 0; m < C::PCS / 2; ++m) {
                const int pos = lane / C::PCS + (64 / C::PCS) * m, pc = lane % C::PCS;        // 32 positions x PCS pieces = 64 lanes x PCS / 2
                st16(ob + (((size_t)b * a.L + l0 + pt * 32 + pos) * N + n0 + 8 * pc), lds_ld16(et + pos * C::EROW + pc * 16));
            }
            HY_SCHED_FENCE();
            }
        }
        HY_WAVE_SYNC_PJ();
    }
}


// =============================================================================================================================
// out_proj's input gradient with the second gate's backward in its epilogue (round 5).  hyena.py:432-440, backwards:
//     dz^T = W_out^T dy^T                                                   (D, B, L)   the gradient of z = y * x0 (transposed GEMM)
//     d y_conv = dz * x0c;   g = dz * y  ->  through the short filter of the x0 channels:
//     d xT[d, m] = w2 g[m] + w1 g[m + 1] + w0 g[m + 2]   (+ the partial sums of d w, d b_sc, d b_in)
// Until round 5: a library GEMM wrote dz^T (560 us for 1 GB at L = 2^20, d = 256 -- the one product of the layer the library runs far below
// the memory rate) and cm_post_bwd (cm_kernels.h) read it back with y and xT.  Here the in_proj kernel's scheme runs the other way round:
// weights-stationary v_mfma_f32_16x16x32 (A = 32 rows of W_out^T per wavefront, B = the staged dy tile, positions on the lanes), accumulators
// rounded to the storage type (dz^T is a 16-bit tensor in the unfused graph: same rounding) and parked as [channel][position] rows in a
// wavefront-private LDS tile; each lane then takes 8 positions of a channel, loads the matching 16-byte pieces of y and of the x0 row of xT,
// and forms everything cm_post_bwd forms -- expression for expression (cm_sc, cm_sc_bwd) -- so dz^T never exists in memory.
// The transposed short convolution needs g at the two positions AFTER a piece: from the neighbouring lane, and across tiles from the tile
// above -- so a workgroup walks its run of tiles DOWNWARDS (a warm-up tile above the run supplies the first two values, its results are
// dropped), the mirror image of the in_proj kernel's halo.  Tiles never cross a sequence (tiles_per_seq per sequence, the last one ragged):
// with per-sequence pitched rows every 16-byte access of the epilogue is aligned whatever L and B are.
// Partial sums (d w0..2, d b_sc, d b_in per channel) stay in registers over the run, are reduced over the 8 lanes of a channel by shuffles
// and leave as one record per (channel, run): part[c][run][8], summed over the runs by the host in a fixed order -- deterministic.
// =============================================================================================================================
enum { DG_CB = 32 /* channels per wavefront: two 16-row MFMA groups */, DG_G = DG_CB / 16, DG_M = DG_CB * 8 / 64 /* (channel, piece) pairs per lane */ };
template <int K> struct DgCfg {
    static_assert(K == 128 || K == 256, "d_model of the HyenaDNA models");
    static constexpr int KS = K / 32;
    static constexpr int PCS = K / 8;
    static constexpr int UROWB = K * 2, UBUF = PJ_NT * UROWB;   // the dy tile [64 positions][K], filled by LDS-direct loads (issue_operand_tile)
    static constexpr int EROW = PJ_EW * 2;                      // [channel][64 positions] + 16 B: conflict-free 16-byte reads
    static constexpr int EBUF = DG_CB * EROW;
    static constexpr size_t LDS = (size_t)UBUF + PJ_WAVES * (size_t)EBUF;
};

struct DgArgs {
    const void* dy;       // (B L, N) 16-bit: gradient of out_proj's output, N = K = D
    const void* Wt;       // (D, N) 16-bit: out_proj.weight TRANSPOSED (row d = the weights that multiply dy[:, n] into channel d)
    const void* y;        // (B, D, L) long-convolution output, row pitch lda
    const void* xT;       // (3D, B, Lx) in_proj output without its bias, row (c, b) at c csx + b bsx; rows [0, D) are read
    const float* bin;     // (3D,) or null
    const float* w;       // (3D, 3)
    const float* b;       // (3D,)
    void* dyc;            // (B, D, L) out: gradient of the long convolution's output, row pitch lda
    void* dxT;            // (3D, B, Lx) out: rows [0, D), positions < L, xT's layout
    float* part;          // out: [D][nrec][8] (dw0, dw1, dw2, db_sc, db_in, -, -, -), record = run
    int B, L, Lx, D;
    long csx; int bsx;    // xT / dxT: row (c, b) at c csx + b bsx
    int lda;              // y / dyc: row (b, d) at (b D + d) lda
    int tiles_per_seq, tiles, tiles_per_wg, nrec;
};

#ifndef DG_WGS
#define DG_WGS 2                 // workgroups per CU the kernel is compiled for.  3 (<= 168 registers, 16 of them spilled) measured 756 vs 800 us at
                                 // L = 2^20 but 250 vs 190 us at 32768 x 8 and 266 vs 224 us at 160000 x 2 (profiles/r5e_outproj_dgrad.txt)
#endif
template <int K, int DT>
__global__ void __launch_bounds__(PJ_THREADS, DG_WGS) outproj_dgrad_gate_bwd_kernel(DgArgs a) {
    typedef DgCfg<K> C;
    typedef typename Elem<DT>::type elem_t;
    static_assert(sizeof(elem_t) == 2, "16-bit element types only");
    HY_SMEM(smem);
    PJ_VMQ_DECL;
    const int tid = (int)threadIdx.x, wave = HY_SGPR(tid >> 6), lane = tid & 63, j = lane & 15, kq = lane >> 4;
    const int D = a.D;
    const unsigned P = (unsigned)a.B * (unsigned)a.L;
    const int ncg = D / (PJ_WAVES * DG_CB);
    int cg, run;
    {
        const int wg = blockIdx.x, xcd = wg & 7, seq = wg >> 3;       // the channel groups of one run of positions share an XCD's L2 (the dy tiles)
        cg = seq % ncg;
        run = (seq / ncg) * 8 + xcd;
    }
    const int t_begin = run * a.tiles_per_wg;
    if (t_begin >= a.tiles) return;
    const int t_end = (t_begin + a.tiles_per_wg < a.tiles) ? t_begin + a.tiles_per_wg : a.tiles;
    const int d0 = cg * PJ_WAVES * DG_CB + wave * DG_CB;             // first channel of this wavefront

    HY_LDS char* const ubuf = HY_LDS_CAST(char, smem);
    HY_LDS char* const ebuf = HY_LDS_CAST(char, smem) + C::UBUF + wave * C::EBUF;

    // stationary operand: 32 rows of W_out^T as A fragments (row = channel 16 g + j, k = 32 ks + 8 kq ...)
    Frag wf[DG_G][C::KS];
    HY_UNROLL
    for (int g = 0; g < DG_G; ++g) {
        const char* row = reinterpret_cast<const char*>(a.Wt) + ((size_t)(d0 + 16 * g + j) * K + 8 * kq) * 2;
        HY_UNROLL
        for (int ks = 0; ks < C::KS; ++ks) wf[g][ks] = ld16(row + ks * 64);
    }
    // this lane's DG_M (channel, piece) pairs: channel (lane >> 3) + 8 m, positions 8 (lane & 7) .. + 7 of the tile
    const int pc = lane & 7;
    // the channels' short-filter taps live in the 16 spare bytes behind each row of the epilogue tile (w0, w1, w2, b_sc), the in_proj bias in a register
    float tbi[DG_M];
    HY_UNROLL
    for (int m = 0; m < DG_M; ++m) {
        const int c = d0 + (lane >> 3) + 8 * m;
        tbi[m] = a.bin != nullptr ? a.bin[c] : 0.f;
    }
    if (lane < DG_CB) {
        HY_LDS float* tp = reinterpret_cast<HY_LDS float*>(ebuf + lane * C::EROW + PJ_NT * 2);
        const int c = d0 + lane;
        tp[0] = a.w[c * 3]; tp[1] = a.w[c * 3 + 1]; tp[2] = a.w[c * 3 + 2]; tp[3] = a.b[c];
    }
    HY_WAVE_SYNC_PJ();
    float sdw0[DG_M], sdw1[DG_M], sdw2[DG_M], sdbs[DG_M], sdbi[DG_M];      // partial sums of the run
    float cr0[DG_M], cr1[DG_M];                                             // g at the first two positions of the tile above (held by the piece-0 lanes)
    HY_UNROLL
    for (int m = 0; m < DG_M; ++m) { sdw0[m] = sdw1[m] = sdw2[m] = sdbs[m] = sdbi[m] = 0.f; cr0[m] = cr1[m] = 0.f; }

    const char* const dbase = reinterpret_cast<const char*>(a.dy);
    const elem_t* const yb = reinterpret_cast<const elem_t*>(a.y);
    const elem_t* const xb = reinterpret_cast<const elem_t*>(a.xT);
    elem_t* const ob = reinterpret_cast<elem_t*>(a.dyc);
    elem_t* const gb = reinterpret_cast<elem_t*>(a.dxT);

    // the walk goes DOWN: tile t_end (if it belongs to the sequence of tile t_end - 1) is the warm-up tile
    const bool warm = t_end < a.tiles && (t_end % a.tiles_per_seq) != 0;
    const int t_first = warm ? t_end : t_end - 1;
    auto tile_p0 = [&](int t) -> unsigned { const int sb = t / a.tiles_per_seq; return (unsigned)sb * (unsigned)a.L + (unsigned)(t - sb * a.tiles_per_seq) * PJ_NT; };
    {
        const unsigned p0 = tile_p0(t_first);
        if (p0 + PJ_NT <= P) issue_operand_tile<K, true>(dbase, p0, P, ubuf, wave, lane PJ_VMQ_ARG);
        else issue_operand_tile<K, false>(dbase, p0, P, ubuf, wave, lane PJ_VMQ_ARG);
    }
    for (int t = t_first; t >= t_begin; --t) {
        const int sb = t / a.tiles_per_seq, ti = t - sb * a.tiles_per_seq, l0 = ti * PJ_NT;
        PJ_VMWAIT(0);
        PJ_BARRIER();                                                        // everybody's share of the tile has landed
        acc4_t acc[DG_G][IP_NT4];
        HY_UNROLL
        for (int g = 0; g < DG_G; ++g) {
            HY_UNROLL
            for (int nt = 0; nt < IP_NT4; ++nt) {
                HY_UNROLL
                for (int r = 0; r < 4; ++r) acc[g][nt][r] = 0.f;
            }
        }
        const int ua = j * C::UROWB, ux = (kq ^ j) * 16;
        HY_UNROLL
        for (int ks = 0; ks < C::KS; ++ks) {
            HY_UNROLL
            for (int nt = 0; nt < IP_NT4; ++nt) {
                const Frag bf = lds_ld16(ubuf + nt * 16 * C::UROWB + ua + (ux ^ (((4 * ks) ^ ((16 * nt) % C::PCS)) * 16)));
                HY_UNROLL
                for (int g = 0; g < DG_G; ++g) acc[g][nt] = mfma16<DT>(wf[g][ks], bf, acc[g][nt]);
            }
        }
        PJ_BARRIER();                                                        // every wavefront has read its last fragment
        if (t - 1 >= t_begin) {
            const unsigned p0 = tile_p0(t - 1);
            if (p0 + PJ_NT <= P) issue_operand_tile<K, true>(dbase, p0, P, ubuf, wave, lane PJ_VMQ_ARG);
            else issue_operand_tile<K, false>(dbase, p0, P, ubuf, wave, lane PJ_VMQ_ARG);
        }
        // ---- epilogue, wavefront-private -------------------------------------------------------------------------------------
        // (1) dz, rounded to the storage type, as [channel][position] rows; positions beyond the sequence are zero (a ragged last tile
        //     multiplied rows of the next sequence, or stale ones)
        HY_UNROLL
        for (int g = 0; g < DG_G; ++g) {
            HY_UNROLL
            for (int nt = 0; nt < IP_NT4; ++nt) {
                HY_UNROLL
                for (int r = 0; r < 4; ++r) {
                    HY_LDS elem_t* e = reinterpret_cast<HY_LDS elem_t*>(ebuf + (g * 16 + 4 * kq + r) * C::EROW);
                    e[nt * 16 + j] = l0 + nt * 16 + j < a.L ? Elem<DT>::cvt(acc[g][nt][r]) : (elem_t)0;
                }
            }
        }
        HY_WAVE_SYNC_PJ();
        if (ti == a.tiles_per_seq - 1) {                                     // a sequence's last tile: nothing above it
            HY_UNROLL
            for (int m = 0; m < DG_M; ++m) { cr0[m] = 0.f; cr1[m] = 0.f; }
        }
        const bool keep = t < t_end;                                         // (the warm-up tile only supplies cr0 / cr1)
        const int l = l0 + 8 * pc;                                           // this lane's first position in the sequence
        const bool whole = l + 8 <= a.L;
        HY_UNROLL
        for (int m = 0; m < DG_M; ++m) {
            const int ch = (lane >> 3) + 8 * m, c = d0 + ch;
            elem_t dze[8], ye[8], xe[10];
            {
                const Frag f = lds_ld16(ebuf + ch * C::EROW + pc * 16);
                __builtin_memcpy(dze, f.w, 16);
            }
            const elem_t* yrow = yb + ((size_t)sb * D + c) * a.lda;
            const elem_t* xrow = xb + ((size_t)c * (size_t)a.csx + (size_t)sb * a.bsx);
            if (whole) {
                const Frag fy = ld16(yrow + l), fx = ld16(xrow + l);
                __builtin_memcpy(ye, fy.w, 16);
                __builtin_memcpy(xe + 2, fx.w, 16);
            } else {
                HY_UNROLL
                for (int i = 0; i < 8; ++i) {
                    const bool ok = l + i < a.L;
                    ye[i] = ok ? yrow[ok ? l + i : 0] : (elem_t)0;
                    xe[2 + i] = ok ? xrow[ok ? l + i : 0] : (elem_t)0;
                }
            }
            // the two raw x values before the piece: the previous lane's last pair, or (piece 0) straight from the row
            {
                uint32_t own;
                __builtin_memcpy(&own, xe + 8, 4);
                uint32_t hw = HY_SHFL_U32(own, lane - 1);
                if (pc == 0) {
                    hw = 0u;
                    if (l0 >= 2) __builtin_memcpy(&hw, xrow + (l0 - 2), 4);
                }
                __builtin_memcpy(xe, &hw, 4);
            }
            float dz[8], g8[10], c0[8];
            HY_UNROLL
            for (int i = 0; i < 8; ++i) dz[i] = Elem<DT>::dec(dze[i]);
            const HY_LDS float* tp = reinterpret_cast<const HY_LDS float*>(ebuf + ch * C::EROW + PJ_NT * 2);
            const float tw0 = tp[0], tw1 = tp[1], tw2 = tp[2], tbs = tp[3];
            HY_UNROLL
            for (int i = 0; i < 8; ++i) {
                const int li = l + i;
                const float x0 = li >= 2 ? Elem<DT>::dec(xe[i]) + tbi[m] : 0.f, x1 = li >= 1 ? Elem<DT>::dec(xe[i + 1]) + tbi[m] : 0.f,
                            x2 = Elem<DT>::dec(xe[i + 2]) + tbi[m];
                c0[i] = __builtin_fmaf(tw2, x2, __builtin_fmaf(tw1, x1, __builtin_fmaf(tw0, x0, tbs)));       // = cm_sc
                g8[i] = dz[i] * Elem<DT>::dec(ye[i]);                                                                  // zero beyond L: dz is
                if (keep) {
                    sdw0[m] += g8[i] * x0; sdw1[m] += g8[i] * x1; sdw2[m] += g8[i] * x2; sdbs[m] += g8[i];            // = cm_sc_bwd
                }
            }
            // g at the two positions after the piece: the next lane's first two, or (piece 7) the tile above's
            {
                const float n0 = u2f(HY_SHFL_U32(f2u(g8[0]), lane + 1)), n1 = u2f(HY_SHFL_U32(f2u(g8[1]), lane + 1));
                const float u0 = u2f(HY_SHFL_U32(f2u(cr0[m]), lane - 7)), u1 = u2f(HY_SHFL_U32(f2u(cr1[m]), lane - 7));
                g8[8] = pc == 7 ? u0 : n0;
                g8[9] = pc == 7 ? u1 : n1;
                cr0[m] = g8[0]; cr1[m] = g8[1];                               // (meaningful in the piece-0 lanes: the tile below reads them there)
            }
            if (keep) {
                elem_t oe[8], de[8];
                HY_UNROLL
                for (int i = 0; i < 8; ++i) {
                    oe[i] = Elem<DT>::cvt(dz[i] * c0[i]);                                                                              // = cm_post_bwd's dy
                    const float dx = __builtin_fmaf(tw0, g8[i + 2], __builtin_fmaf(tw1, g8[i + 1], tw2 * g8[i]));         // = cm_sc_bwd's dx
                    if (l + i < a.L) sdbi[m] += dx;
                    de[i] = Elem<DT>::cvt(dx);
                }
                elem_t* orow = ob + ((size_t)sb * D + c) * a.lda;
                elem_t* drow = gb + ((size_t)c * (size_t)a.csx + (size_t)sb * a.bsx);
                if (whole) {
                    Frag fo, fd;
                    __builtin_memcpy(fo.w, oe, 16);
                    __builtin_memcpy(fd.w, de, 16);
                    st16(orow + l, fo);
                    st16(drow + l, fd);
                } else {
                    HY_UNROLL
                    for (int i = 0; i < 8; ++i) {
                        if (l + i < a.L) { orow[l + i] = oe[i]; drow[l + i] = de[i]; }
                    }
                }
            }
        }
        HY_WAVE_SYNC_PJ();                        // the tile is re-written in the next round
    }
    // ---- the run's partial sums: over the 8 lanes (pieces) of a channel, fixed order; one record per (channel, run) ----
    HY_UNROLL
    for (int m = 0; m < DG_M; ++m) {
        float v[5] = {sdw0[m], sdw1[m], sdw2[m], sdbs[m], sdbi[m]};
        HY_UNROLL
        for (int f = 0; f < 5; ++f) {
            HY_UNROLL
            for (int o = 1; o < 8; o <<= 1) v[f] += u2f(HY_SHFL_U32(f2u(v[f]), lane ^ o));
        }
        if (pc == 0) {
            float* rec = a.part + ((size_t)(d0 + (lane >> 3) + 8 * m) * a.nrec + run) * 8;
            HY_UNROLL
            for (int f = 0; f < 5; ++f) rec[f] = v[f];
        }
    }
}

}  // namespace pj
}  // namespace hyena
