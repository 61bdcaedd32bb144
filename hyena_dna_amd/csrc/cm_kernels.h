// cm_kernels.h -- the element-wise shell of the Hyena operator in CHANNEL-MAJOR layout.
//
// Reference: src/models/sequence/hyena.py:391-440 (order 2, one head / block, inner factor 1, dropout 0, activation id).
// The reference computes x = in_proj(u) as (B, L, 3D) and immediately rearranges it to (B, 3D, L) for the depthwise
// short convolution, the gates and the long convolution, then rearranges back for out_proj -- two transposing copies.
// mixer_kernels.h (round 1) fuses those copies into its four kernels through 64 x 64 LDS tiles.  Here the projections
// themselves produce / consume the transposed tensors (the GEMM library takes transposed operands at no cost --
// measured in profiles/gemm_layout_r2.txt), so every tensor between in_proj and out_proj is a set of rows contiguous
// along L and the shell is four 1-D streaming kernels with 16-byte accesses, no LDS tiles, no transposes:
//
//     xT  = W_in u^T                               (3D, B, Lx)    [GEMM; the bias b_in is added here, on load]
//     xc[c, b, t] = b_sc[c] + sum_j w[c, j] (xT[c, b, t-2+j] + b_in[c])  for t-2+j >= 0      hyena.py:394 (Conv1d k=3, padding 2)
//     vg[b, d, t] = xc[2D+d, b, t] xc[D+d, b, t]                     cm_pre_fwd            hyena.py:404,420
//     y           = fftconv(vg, k, bias)                             (hyena_fftconv_*)      hyena.py:423
//     zT[d, b, t] = y[b, d, t] xc[d, b, t]                           cm_post_fwd           hyena.py:432
//     out         = zT^T W_out^T + b_out                             [GEMM]                hyena.py:440
// and the gradients: cm_post_bwd (dzT, y, xT -> dy, dxT rows [0, D)), cm_pre_bwd (dvg, xT -> dxT rows [D, 3D)); both leave
// per-workgroup partial sums of the short filter's weight / bias gradients and of db_in (deterministic two-stage sums).
//
// A workgroup = 256 threads x 8 consecutive positions = one 2048-position tile of one (channel, batch) row.
#pragma once
#define HY_HELPERS_ONLY
#include "fftconv_kernels.h"

namespace hyena {

#ifndef HY_CM_V
#define HY_CM_V 8
#endif
enum { CM_THREADS = 256, CM_V = HY_CM_V, CM_TILE = CM_THREADS * CM_V, CM_NP = 8 /* floats per partial record */ };

struct CmArgs {
    const void* xT;     // (3D, B, Lx) in_proj output WITHOUT its bias, elements of DT
    const float* bin;   // (3D,) in_proj bias (fp32), added on load
    const float* w;     // (3D, 3) short-filter taps
    const float* b;     // (3D,)   short-filter bias
    const void* a0;     // pre_fwd: -           post_fwd: y (B, D, L)   post_bwd: y (B, D, L)     pre_bwd: dvg (B, D, L)
    const void* a1;     // post_bwd: dzT (D, B, L)
    void* o0;           // pre_fwd: vg (B, D, L)   post_fwd: zT (D, B, L)   post_bwd: dy (B, D, L)
    void* dxT;          // bwd: (3D, B, Lx) gradient of xT (post_bwd writes rows [0, D), pre_bwd rows [D, 3D)); positions >= L untouched
    float* part;        // bwd: [3D][B * tiles][CM_NP] partial sums (dw0, dw1, dw2, db_sc, db_in, -, -, -): the host reads [:5]
    int B, L, D, Lx;
    long csx; int bsx;  // xT and dxT: row (c, b) starts at element c csx + b bsx (packed: csx = B Lx, bsx = Lx; per-sequence pitch ld: csx = B ld, bsx = ld;
                        // channel rows pitched over the FLATTENED positions -- what a library GEMM takes as one matrix -- csx >= B Lx, bsx = Lx)
    long csz; int bsz;  // zT / dzT (d, b) likewise
    int lda;            // row pitch (elements, >= L) of the (B, D, L) tensors vg / y / dy / dvg: row (b, d) at (b D + d) lda
    int rpw;            // (channel, batch) rows per workgroup: 1, or 2 / 4 / 8 for sequences of at most CM_TILE / rpw positions (round 6: a 2048-position
                        // tile is half empty at L = 1024 -- the shipped experiment's 256 x 1023 x 128 -- and every workgroup pays its reduction and
                        // record; the rows of one workgroup share the channel and differ in b: grid.z = ceil(B / rpw), one record per workgroup)
};
// thread -> (batch item, first position); dead threads (beyond B) get l0 = L: they load zeros, store nothing and add zeros to the sums
struct CmPos { int b, l0; };
__device__ __forceinline__ CmPos cm_pos(const CmArgs& a) {
    CmPos p;
    if (a.rpw <= 1) {
        p.b = blockIdx.z;
        p.l0 = (int)(blockIdx.x * CM_THREADS + threadIdx.x) * CM_V;
        return p;
    }
    const int tpr = CM_THREADS / a.rpw, r = (int)threadIdx.x / tpr, tx = (int)threadIdx.x % tpr;
    const int b = (int)blockIdx.z * a.rpw + r;
    p.b = b < a.B ? b : a.B - 1;
    p.l0 = b < a.B ? tx * CM_V : a.L + CM_V;             // (grid.x = 1: rpw > 1 only when a row fits CM_TILE / rpw positions)
    return p;
}

// v[i] = row[l0 + i] for i in [LO, HI), zero outside [0, L).  Interior vectors move as 16-byte (8 x 16-bit) or 2 x 16-byte
// accesses (rows of odd length start under-aligned: gfx950 global memory handles that).
// HY_POL_CM: cache policy of the shell's output streams (2 = nt; every output is written once and read by a later launch)
#ifndef HY_POL_CM
#define HY_POL_CM 2
#endif
#if !defined(HIPEMU)
typedef unsigned cm_vec16 __attribute__((ext_vector_type(4), aligned(2)));   // rows start at any even byte offset
#endif
template <int DT, int N>
__device__ __forceinline__ void cm_ld(const void* row, int l0, int L, float (&v)[N]) {
    typedef typename Elem<DT>::type elem_t;
    const elem_t* p = reinterpret_cast<const elem_t*>(row);
    if (l0 >= 0 && l0 + N <= L) {
        elem_t raw[N];
        __builtin_memcpy(raw, p + l0, sizeof(raw));
        HY_UNROLL
        for (int i = 0; i < N; ++i) v[i] = Elem<DT>::dec(raw[i]);
    } else {
        HY_UNROLL
        for (int i = 0; i < N; ++i) {
            const int l = l0 + i;
            const bool ok = l >= 0 && l < L;
            const float f = Elem<DT>::ld(p + (ok ? l : 0));
            v[i] = ok ? f : 0.f;
        }
    }
}
template <int DT, int N>
__device__ __forceinline__ void cm_st(void* row, int l0, int L, const float (&v)[N]) {
    typedef typename Elem<DT>::type elem_t;
    elem_t* p = reinterpret_cast<elem_t*>(row);
    if (l0 + N <= L) {
        elem_t raw[N];
        HY_UNROLL
        for (int i = 0; i < N; ++i) raw[i] = Elem<DT>::cvt(v[i]);
#if !defined(HIPEMU)
        if constexpr ((HY_POL_CM & 2) != 0 && sizeof(raw) % 16 == 0) {      // write-once stream: non-temporal 16-byte stores
            HY_UNROLL
            for (int j = 0; j < (int)(sizeof(raw) / 16); ++j) {
                cm_vec16 t;
                __builtin_memcpy(&t, reinterpret_cast<const char*>(raw) + 16 * j, 16);
                __builtin_nontemporal_store(t, reinterpret_cast<cm_vec16*>(p + l0) + j);
            }
            return;
        }
#endif
        __builtin_memcpy(p + l0, raw, sizeof(raw));
    } else {
        HY_UNROLL
        for (int i = 0; i < N; ++i)
            if (l0 + i < L) Elem<DT>::st(p + l0 + i, v[i]);
    }
}

struct CmTap { float w0, w1, w2, bsc, bin; };
__device__ __forceinline__ CmTap cm_tap(const CmArgs& a, int c) {
    CmTap t;
    t.w0 = a.w[c * 3]; t.w1 = a.w[c * 3 + 1]; t.w2 = a.w[c * 3 + 2];
    t.bsc = a.b[c];
    t.bin = a.bin != nullptr ? a.bin[c] : 0.f;
    return t;
}
// xs[i] = raw xT[l0 - 2 + i]; out[i] = short-conv output at l0 + i (taps that fall before position 0 are zero padding)
template <int N>
__device__ __forceinline__ void cm_sc(const float (&xs)[N + 2], int l0, const CmTap& t, float (&out)[N]) {
    HY_UNROLL
    for (int i = 0; i < N; ++i) {
        const int l = l0 + i;
        const float x0 = l >= 2 ? xs[i] + t.bin : 0.f, x1 = l >= 1 ? xs[i + 1] + t.bin : 0.f, x2 = xs[i + 2] + t.bin;
        // explicit fused multiply-adds in a fixed order: the projection kernel's epilogue (proj_kernels.h) evaluates the same
        // expression and must give the same bits -- left to the compiler, the contraction into FMAs differs between kernels
        out[i] = __builtin_fmaf(t.w2, x2, __builtin_fmaf(t.w1, x1, __builtin_fmaf(t.w0, x0, t.bsc)));
    }
}
__device__ __forceinline__ const char* cm_row(const void* base, size_t row, int len, size_t es) {
    return reinterpret_cast<const char*>(base) + row * (size_t)len * es;
}
// row (c, b) of a channel-major (C, B, .) tensor with channel stride cs and sequence stride bs (elements)
__device__ __forceinline__ const char* cm_cb(const void* base, int c, int b, long cs, int bs, size_t es) {
    return reinterpret_cast<const char*>(base) + ((size_t)c * (size_t)cs + (size_t)b * (size_t)bs) * es;
}
template <int DT> struct CmEs { static constexpr size_t V = (DT == DT_F32) ? 4 : 2; };

// vg[b, d, :] = xc[2D + d, b, :] * xc[D + d, b, :]                grid (tiles, D, B)
template <int DT>
__global__ void __launch_bounds__(CM_THREADS) cm_pre_fwd_kernel(CmArgs a) {
    constexpr size_t ES = CmEs<DT>::V;
    const int d = blockIdx.y;
    const CmPos ps = cm_pos(a);
    const int b = ps.b, l0 = ps.l0;
    if (l0 >= a.L) return;
    const CmTap t1 = cm_tap(a, a.D + d), tv = cm_tap(a, 2 * a.D + d);
    float x1[CM_V + 2], xv[CM_V + 2], c1[CM_V], cv[CM_V], o[CM_V];
    cm_ld<DT, CM_V + 2>(cm_cb(a.xT, a.D + d, b, a.csx, a.bsx, ES), l0 - 2, a.Lx, x1);
    cm_ld<DT, CM_V + 2>(cm_cb(a.xT, 2 * a.D + d, b, a.csx, a.bsx, ES), l0 - 2, a.Lx, xv);
    cm_sc<CM_V>(x1, l0, t1, c1);
    cm_sc<CM_V>(xv, l0, tv, cv);
    HY_UNROLL
    for (int i = 0; i < CM_V; ++i) o[i] = c1[i] * cv[i];
    cm_st<DT, CM_V>(const_cast<char*>(cm_row(a.o0, (size_t)b * a.D + d, a.lda, ES)), l0, a.L, o);
}

// zT[d, b, :] = y[b, d, :] * xc[d, b, :]
template <int DT>
__global__ void __launch_bounds__(CM_THREADS) cm_post_fwd_kernel(CmArgs a) {
    constexpr size_t ES = CmEs<DT>::V;
    const int d = blockIdx.y;
    const CmPos ps = cm_pos(a);
    const int b = ps.b, l0 = ps.l0;
    if (l0 >= a.L) return;
    const CmTap t0 = cm_tap(a, d);
    float x0[CM_V + 2], c0[CM_V], y[CM_V], o[CM_V];
    cm_ld<DT, CM_V + 2>(cm_cb(a.xT, d, b, a.csx, a.bsx, ES), l0 - 2, a.Lx, x0);
    cm_ld<DT, CM_V>(cm_row(a.a0, (size_t)b * a.D + d, a.lda, ES), l0, a.L, y);
    cm_sc<CM_V>(x0, l0, t0, c0);
    HY_UNROLL
    for (int i = 0; i < CM_V; ++i) o[i] = y[i] * c0[i];
    cm_st<DT, CM_V>(const_cast<char*>(cm_cb(a.o0, d, b, a.csz, a.bsz, ES)), l0, a.L, o);
}

// wavefront sum (xor butterfly: every lane ends with the total, fixed order)
__device__ __forceinline__ float cm_wave_sum(float v) {
    HY_UNROLL
    for (int off = 32; off > 0; off >>= 1) v += u2f(HY_SHFL_U32(f2u(v), (threadIdx.x & 63) ^ off));
    return v;
}

// Back through one short-conv channel.  da[i] = gradient w.r.t. the conv OUTPUT at l0 + i, i in [0, V + 2) (zero beyond L);
// xs[i] = raw xT[l0 - 2 + i], i in [0, V + 2).  Writes dx[l0 .. l0 + V) and returns this thread's partial sums.
struct CmPart { float dw0, dw1, dw2, dbsc, dbin; };
template <int DT>
__device__ __forceinline__ CmPart cm_sc_bwd(const float (&da)[CM_V + 2], const float (&xs)[CM_V + 2], int l0, int L, const CmTap& t,
                                            void* dx_row, int Lx) {
    CmPart p = {0.f, 0.f, 0.f, 0.f, 0.f};
    float dx[CM_V];
    HY_UNROLL
    for (int i = 0; i < CM_V; ++i) {
        const int m = l0 + i;
        // x[m] feeds outputs m (tap 2), m + 1 (tap 1), m + 2 (tap 0); da is already zero beyond L
        // (explicit fused multiply-adds in a fixed order: outproj_dgrad_gate_bwd_kernel, proj_kernels.h, evaluates the same expression and must give the same bits)
        dx[i] = __builtin_fmaf(t.w0, da[i + 2], __builtin_fmaf(t.w1, da[i + 1], t.w2 * da[i]));
        if (m < L) p.dbin += dx[i];
        // the outputs this thread owns (l = m): dw[j] += da[l] x_true[l - 2 + j]
        const float x0 = m >= 2 ? xs[i] + t.bin : 0.f, x1 = m >= 1 ? xs[i + 1] + t.bin : 0.f, x2 = xs[i + 2] + t.bin;
        p.dw0 += da[i] * x0;
        p.dw1 += da[i] * x1;
        p.dw2 += da[i] * x2;
        p.dbsc += da[i];
    }
    cm_st<DT, CM_V>(dx_row, l0, L < Lx ? L : Lx, dx);
    return p;
}
// The workgroup's sums of NC channels' partial records -> part[c][rec][:], ONE barrier for all 5 NC values: wavefront sums
// by shuffles, the four wavefronts' results through LDS, added in a fixed order (deterministic).
template <int NC>
__device__ __forceinline__ void cm_store_parts(const CmArgs& a, const int (&c)[NC], int rec, int nrec, const CmPart (&p)[NC],
                                               HY_LDS float* red) {
    float v[NC * 5];
    HY_UNROLL
    for (int i = 0; i < NC; ++i) {
        v[i * 5] = cm_wave_sum(p[i].dw0); v[i * 5 + 1] = cm_wave_sum(p[i].dw1); v[i * 5 + 2] = cm_wave_sum(p[i].dw2);
        v[i * 5 + 3] = cm_wave_sum(p[i].dbsc); v[i * 5 + 4] = cm_wave_sum(p[i].dbin);
    }
    if ((threadIdx.x & 63) == 0) {
        HY_UNROLL
        for (int i = 0; i < NC * 5; ++i) red[i * 4 + (threadIdx.x >> 6)] = v[i];
    }
    __syncthreads();
    if (threadIdx.x < NC * 5) {
        const int i = threadIdx.x, ch = i / 5, f = i % 5;
        const float s = (red[i * 4] + red[i * 4 + 1]) + (red[i * 4 + 2] + red[i * 4 + 3]);
        a.part[((size_t)c[ch] * nrec + rec) * CM_NP + f] = s;
    }
}

// dy[b, d, :] = dzT[d, b, :] * xc[d, b, :];   g = dzT * y  -> dxT[d, b, :] and the partials of channel d
template <int DT>
__global__ void __launch_bounds__(CM_THREADS) cm_post_bwd_kernel(CmArgs a) {
    constexpr size_t ES = CmEs<DT>::V;
    HY_SMEM(smem);
    HY_LDS float* red = HY_LDS_CAST(float, smem);
    const int d = blockIdx.y;
    const CmPos pos = cm_pos(a);
    const int b = pos.b, l0 = pos.l0;                                  // (threads beyond L still take part in the sums)
    const CmTap t0 = cm_tap(a, d);
    float x0[CM_V + 2], dz[CM_V + 2], y[CM_V + 2], c0[CM_V], dy[CM_V], da[CM_V + 2];
    cm_ld<DT, CM_V + 2>(cm_cb(a.xT, d, b, a.csx, a.bsx, ES), l0 - 2, a.Lx, x0);
    cm_ld<DT, CM_V + 2>(cm_cb(a.a1, d, b, a.csz, a.bsz, ES), l0, a.L, dz);
    cm_ld<DT, CM_V + 2>(cm_row(a.a0, (size_t)b * a.D + d, a.lda, ES), l0, a.L, y);
    cm_sc<CM_V>(x0, l0, t0, c0);
    HY_UNROLL
    for (int i = 0; i < CM_V; ++i) dy[i] = dz[i] * c0[i];
    HY_UNROLL
    for (int i = 0; i < CM_V + 2; ++i) da[i] = dz[i] * y[i];            // zero beyond L: dz is
    if (l0 < a.L) cm_st<DT, CM_V>(const_cast<char*>(cm_row(a.o0, (size_t)b * a.D + d, a.lda, ES)), l0, a.L, dy);
    CmPart p = {0.f, 0.f, 0.f, 0.f, 0.f};
    if (l0 < a.L)
        p = cm_sc_bwd<DT>(da, x0, l0, a.L, t0, const_cast<char*>(cm_cb(a.dxT, d, b, a.csx, a.bsx, ES)), a.Lx);
    const int cs[1] = {d};
    const CmPart ps[1] = {p};
    cm_store_parts<1>(a, cs, blockIdx.z * gridDim.x + blockIdx.x, gridDim.z * gridDim.x, ps, red);
}

// dvg -> dxT rows D + d (through x1c: gradient dvg * vc) and 2D + d (through vc: gradient dvg * x1c), and their partials
template <int DT>
__global__ void __launch_bounds__(CM_THREADS) cm_pre_bwd_kernel(CmArgs a) {
    constexpr size_t ES = CmEs<DT>::V;
    HY_SMEM(smem);
    HY_LDS float* red = HY_LDS_CAST(float, smem);
    const int d = blockIdx.y;
    const CmPos pos = cm_pos(a);
    const int b = pos.b, l0 = pos.l0;
    const CmTap t1 = cm_tap(a, a.D + d), tv = cm_tap(a, 2 * a.D + d);
    // conv outputs are needed at l0 .. l0 + V + 1, hence raw inputs at l0 - 2 .. l0 + V + 1
    float x1[CM_V + 4], xv[CM_V + 4], g[CM_V + 2], c1[CM_V + 2], cv[CM_V + 2], da1[CM_V + 2], dav[CM_V + 2];
    cm_ld<DT, CM_V + 4>(cm_cb(a.xT, a.D + d, b, a.csx, a.bsx, ES), l0 - 2, a.Lx, x1);
    cm_ld<DT, CM_V + 4>(cm_cb(a.xT, 2 * a.D + d, b, a.csx, a.bsx, ES), l0 - 2, a.Lx, xv);
    cm_ld<DT, CM_V + 2>(cm_row(a.a0, (size_t)b * a.D + d, a.lda, ES), l0, a.L, g);
    cm_sc<CM_V + 2>(x1, l0, t1, c1);
    cm_sc<CM_V + 2>(xv, l0, tv, cv);
    HY_UNROLL
    for (int i = 0; i < CM_V + 2; ++i) { da1[i] = g[i] * cv[i]; dav[i] = g[i] * c1[i]; }      // zero beyond L: dvg is
    float xs1[CM_V + 2], xsv[CM_V + 2];
    HY_UNROLL
    for (int i = 0; i < CM_V + 2; ++i) { xs1[i] = x1[i]; xsv[i] = xv[i]; }
    CmPart p1 = {0.f, 0.f, 0.f, 0.f, 0.f}, pv = {0.f, 0.f, 0.f, 0.f, 0.f};
    if (l0 < a.L) {
        p1 = cm_sc_bwd<DT>(da1, xs1, l0, a.L, t1, const_cast<char*>(cm_cb(a.dxT, a.D + d, b, a.csx, a.bsx, ES)), a.Lx);
        pv = cm_sc_bwd<DT>(dav, xsv, l0, a.L, tv, const_cast<char*>(cm_cb(a.dxT, 2 * a.D + d, b, a.csx, a.bsx, ES)), a.Lx);
    }
    const int cs[2] = {a.D + d, 2 * a.D + d};
    const CmPart ps[2] = {p1, pv};
    cm_store_parts<2>(a, cs, blockIdx.z * gridDim.x + blockIdx.x, gridDim.z * gridDim.x, ps, red);
}

}  // namespace hyena
