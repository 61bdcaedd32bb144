// mixer_kernels.h -- the element-wise shell of the Hyena operator around the long convolution, fused:
// the order-2 short depthwise convolution (k = 3, causal), the two multiplicative gates and the two layout changes
// between the projections' (B, L, C) and the convolution's (B, D, L).
//
// Reference (src/models/sequence/hyena.py:388-444, order 2, one head/block, inner factor 1, dropout 0, activation id):
//     x  = in_proj(u)                                     (B, L, 3D)          hyena.py:391
//     xc = short_filter(x^T)[..., :L']                    (B, 3D, L')         hyena.py:394   xc[c,t] = b[c] + sum_i w[c,i] x[t-2+i,c]
//     x0, x1, v = xc.split(D, dim=1)                                          hyena.py:404
//     vg = v * x1                                                             hyena.py:420
//     y  = fftconv(vg, k, bias)                           (B, D, L')          hyena.py:423   (hyena_fftconv_*)
//     z  = (y * x0)^T                                     (B, L', D)          hyena.py:432-439
//     out = out_proj(z)                                                       hyena.py:440
// The reference spends one transposing copy inside conv1d, the conv, three splits/multiplies and one
// rearranging copy on this -- eight passes over (B, L, D)-sized tensors; here it is two kernels forward
// (pre: x -> vg;  post: y, x -> z) and two backward, each a single streaming pass.
//
// Structure of all four kernels: a workgroup is ONE wavefront that owns 64 channels (one per lane, so accesses to the
// channel-minor (B, L, 3D) / (B, L, D) tensors are coalesced) and a run of NT x 64 positions which it walks
// sequentially, carrying the 3-tap windows in registers; the (B, D, L)-side tensor of each 64 x 64 tile is transposed
// through a 64 x 65-float LDS tile so that its global accesses are coalesced along L.  No atomics: the per-run partial
// sums of the short filter's gradients go to a small buffer that the host reduces (deterministic).
#pragma once
#include "fftconv_kernels.h"

namespace hyena {

enum { MIX_T = 64, MIX_NT = 16, MIX_RUN = MIX_T * MIX_NT };

struct MixArgs {
    const void* x;      // (B, Lx, 3D) in_proj output, elements of DT
    const float* w;     // (3D, 3) short-filter taps:  xc[c,t] = b[c] + w[c,0] x[t-2,c] + w[c,1] x[t-1,c] + w[c,2] x[t,c]
    const float* b;     // (3D,)
    void* a0;           // pre_fwd: vg out (B, D, L)      post_fwd: y in (B, D, L)     post_bwd: y in       pre_bwd: dvg in (B, D, L)
    void* a1;           // post_fwd: z out (B, L, D)      post_bwd: dz in (B, L, D)
    void* a2;           // post_bwd: dy out (B, D, L)
    void* dx;           // bwd: (B, Lx, 3D) gradient of x (post_bwd writes channels [0, D), pre_bwd [D, 3D))
    float* part;        // bwd: partial sums [B][nruns][3D][4] = (dw0, dw1, dw2, db) per run
    int B, L, D, Lx;    // L = positions processed (= min(Lx, l_max)); Lx = positions of x
};

template <int DT>
__device__ __forceinline__ float mix_ld(const void* base, size_t idx, bool ok) {
    typedef typename Elem<DT>::type elem_t;
    // unconditional load from a clamped (always valid) index, then a select: a conditional load costs a branch and an
    // s_waitcnt vmcnt(0) of its own, which serialises the 64 loads of a tile
    const float v = Elem<DT>::ld(reinterpret_cast<const elem_t*>(base) + (ok ? idx : (size_t)0));
    return ok ? v : 0.f;
}
template <int DT>
__device__ __forceinline__ void mix_st(void* base, size_t idx, bool ok, float v) {
    typedef typename Elem<DT>::type elem_t;
    if (ok) Elem<DT>::st(reinterpret_cast<elem_t*>(base) + idx, v);
}

// vg[b, d, t] = xc[b, 2D + d, t] * xc[b, D + d, t]
template <int DT>
__global__ void __launch_bounds__(64) mixer_pre_fwd_kernel(MixArgs a) {
    HY_SMEM(smem);
    HY_LDS float* tile = HY_LDS_CAST(float, smem);                 // [64 channels][65]
    const int lane = threadIdx.x;
    const int c0 = blockIdx.y * 64, c = c0 + lane, b = blockIdx.z;
    const bool cv = c < a.D;
    const int D3 = 3 * a.D;
    const int cs = cv ? c : 0;
    float w1[3], w2[3];
    HY_UNROLL
    for (int i = 0; i < 3; ++i) { w1[i] = a.w[(a.D + cs) * 3 + i]; w2[i] = a.w[(2 * a.D + cs) * 3 + i]; }
    const float b1 = a.b[a.D + cs], b2 = a.b[2 * a.D + cs];
    const size_t xb = (size_t)b * a.Lx * D3;
    const int t_begin = blockIdx.x * MIX_RUN;
    float x1m2 = mix_ld<DT>(a.x, xb + (size_t)(t_begin - 2) * D3 + a.D + cs, cv && t_begin >= 2);
    float x1m1 = mix_ld<DT>(a.x, xb + (size_t)(t_begin - 1) * D3 + a.D + cs, cv && t_begin >= 1);
    float vm2 = mix_ld<DT>(a.x, xb + (size_t)(t_begin - 2) * D3 + 2 * a.D + cs, cv && t_begin >= 2);
    float vm1 = mix_ld<DT>(a.x, xb + (size_t)(t_begin - 1) * D3 + 2 * a.D + cs, cv && t_begin >= 1);
    for (int st = 0; st < MIX_NT; ++st) {
        const int t0 = t_begin + st * MIX_T;
        if (t0 >= a.L) break;
        HY_UNROLL
        for (int p = 0; p < MIX_T; ++p) {
            const int t = t0 + p;
            const bool ok = cv && t < a.L;
            const float x1t = mix_ld<DT>(a.x, xb + (size_t)t * D3 + a.D + cs, ok);
            const float vt = mix_ld<DT>(a.x, xb + (size_t)t * D3 + 2 * a.D + cs, ok);
            const float x1c = b1 + w1[0] * x1m2 + w1[1] * x1m1 + w1[2] * x1t;
            const float vc = b2 + w2[0] * vm2 + w2[1] * vm1 + w2[2] * vt;
            tile[lane * 65 + p] = x1c * vc;
            x1m2 = x1m1; x1m1 = x1t; vm2 = vm1; vm1 = vt;
        }
        __syncthreads();
        const int t = t0 + lane;
        for (int cc = 0; cc < 64; ++cc)
            mix_st<DT>(a.a0, ((size_t)b * a.D + c0 + cc) * a.L + t, t < a.L && c0 + cc < a.D, tile[cc * 65 + lane]);
        __syncthreads();
    }
}

// z[b, t, d] = y[b, d, t] * xc[b, d, t]
template <int DT>
__global__ void __launch_bounds__(64) mixer_post_fwd_kernel(MixArgs a) {
    HY_SMEM(smem);
    HY_LDS float* tile = HY_LDS_CAST(float, smem);
    const int lane = threadIdx.x;
    const int c0 = blockIdx.y * 64, c = c0 + lane, b = blockIdx.z;
    const bool cv = c < a.D;
    const int D3 = 3 * a.D;
    const int cs = cv ? c : 0;
    float w0[3];
    HY_UNROLL
    for (int i = 0; i < 3; ++i) w0[i] = a.w[cs * 3 + i];
    const float b0 = a.b[cs];
    const size_t xb = (size_t)b * a.Lx * D3;
    const int t_begin = blockIdx.x * MIX_RUN;
    float xm2 = mix_ld<DT>(a.x, xb + (size_t)(t_begin - 2) * D3 + cs, cv && t_begin >= 2);
    float xm1 = mix_ld<DT>(a.x, xb + (size_t)(t_begin - 1) * D3 + cs, cv && t_begin >= 1);
    for (int st = 0; st < MIX_NT; ++st) {
        const int t0 = t_begin + st * MIX_T;
        if (t0 >= a.L) break;
        {
            const int t = t0 + lane;
            for (int cc = 0; cc < 64; ++cc)
                tile[cc * 65 + lane] = mix_ld<DT>(a.a0, ((size_t)b * a.D + c0 + cc) * a.L + t, t < a.L && c0 + cc < a.D);
        }
        __syncthreads();
        HY_UNROLL
        for (int p = 0; p < MIX_T; ++p) {
            const int t = t0 + p;
            const bool ok = cv && t < a.L;
            const float xt = mix_ld<DT>(a.x, xb + (size_t)t * D3 + cs, ok);
            const float x0c = b0 + w0[0] * xm2 + w0[1] * xm1 + w0[2] * xt;
            mix_st<DT>(a.a1, ((size_t)b * a.L + t) * a.D + cs, ok, x0c * tile[lane * 65 + p]);
            xm2 = xm1; xm1 = xt;
        }
        __syncthreads();
    }
}

// Given dz: dy = dz^T * x0c;  g = dz^T * y (gradient of x0c);  dx[.., 0:D] = conv^T(g);  partial (dw, db) of group 0.
template <int DT>
__global__ void __launch_bounds__(64) mixer_post_bwd_kernel(MixArgs a) {
    HY_SMEM(smem);
    HY_LDS float* ty = HY_LDS_CAST(float, smem);                   // y tile      [64][65]
    HY_LDS float* td = ty + 64 * 65;                               // dy tile     [64][65]
    const int lane = threadIdx.x;
    const int c0 = blockIdx.y * 64, c = c0 + lane, b = blockIdx.z;
    const bool cv = c < a.D;
    const int D3 = 3 * a.D;
    const int cs = cv ? c : 0;
    float w0[3];
    HY_UNROLL
    for (int i = 0; i < 3; ++i) w0[i] = a.w[cs * 3 + i];
    const float b0 = a.b[cs];
    const size_t xb = (size_t)b * a.Lx * D3;
    const int t_begin = blockIdx.x * MIX_RUN;
    const int t_end = (t_begin + MIX_RUN < a.L) ? t_begin + MIX_RUN : a.L;     // own positions [t_begin, t_end)
    float xm2 = mix_ld<DT>(a.x, xb + (size_t)(t_begin - 2) * D3 + cs, cv && t_begin >= 2);
    float xm1 = mix_ld<DT>(a.x, xb + (size_t)(t_begin - 1) * D3 + cs, cv && t_begin >= 1);
    float gm1 = 0.f, gm2 = 0.f;
    float dw[3] = {0.f, 0.f, 0.f}, db = 0.f;
    // NT tiles of own positions, then 2 halo positions whose g completes dx[t_end-2], dx[t_end-1]
    for (int st = 0; st <= MIX_NT; ++st) {
        const int t0 = t_begin + st * MIX_T;
        if (t0 >= t_end + 2 || t0 >= a.L + 2) break;
        const int npos = (st < MIX_NT) ? MIX_T : 2;
        {
            const int t = t0 + lane;
            for (int cc = 0; cc < 64; ++cc)
                ty[cc * 65 + lane] = mix_ld<DT>(a.a0, ((size_t)b * a.D + c0 + cc) * a.L + t, t < a.L && c0 + cc < a.D);
        }
        __syncthreads();
        for (int p = 0; p < npos; ++p) {
            const int t = t0 + p;
            const bool in_l = cv && t < a.L;
            const bool own = t < t_end;
            const float dzv = mix_ld<DT>(a.a1, ((size_t)b * a.L + t) * a.D + cs, in_l);
            const float xt = mix_ld<DT>(a.x, xb + (size_t)t * D3 + cs, in_l);
            const float x0c = b0 + w0[0] * xm2 + w0[1] * xm1 + w0[2] * xt;
            td[lane * 65 + p] = dzv * x0c;
            const float g = dzv * ty[lane * 65 + p];
            if (own) { dw[0] += g * xm2; dw[1] += g * xm1; dw[2] += g * xt; db += g; }
            // dx[t-2] = w0 g(t) + w1 g(t-1) + w2 g(t-2); emitted for t-2 in [t_begin, t_end)
            const int td2 = t - 2;
            mix_st<DT>(a.dx, xb + (size_t)td2 * D3 + cs, cv && td2 >= t_begin && td2 < t_end,
                       w0[0] * g + w0[1] * gm1 + w0[2] * gm2);
            gm2 = gm1; gm1 = g; xm2 = xm1; xm1 = xt;
        }
        __syncthreads();
        if (st < MIX_NT) {
            const int t = t0 + lane;
            for (int cc = 0; cc < 64; ++cc)
                mix_st<DT>(a.a2, ((size_t)b * a.D + c0 + cc) * a.L + t, t < t_end && c0 + cc < a.D, td[cc * 65 + lane]);
        }
        __syncthreads();
    }
    if (cv) {
        float* pp = a.part + (((size_t)b * gridDim.x + blockIdx.x) * D3 + c) * 4;
        pp[0] = dw[0]; pp[1] = dw[1]; pp[2] = dw[2]; pp[3] = db;
    }
}

// Given dvg (gradient of vg = v * x1): g1 = dvg * vc, g2 = dvg * x1c;  dx[.., D:3D] = conv^T(g1 | g2);  partials.
template <int DT>
__global__ void __launch_bounds__(64) mixer_pre_bwd_kernel(MixArgs a) {
    HY_SMEM(smem);
    HY_LDS float* tg = HY_LDS_CAST(float, smem);                   // dvg tile [64][65]
    const int lane = threadIdx.x;
    const int c0 = blockIdx.y * 64, c = c0 + lane, b = blockIdx.z;
    const bool cv = c < a.D;
    const int D3 = 3 * a.D;
    const int cs = cv ? c : 0;
    float w1[3], w2[3];
    HY_UNROLL
    for (int i = 0; i < 3; ++i) { w1[i] = a.w[(a.D + cs) * 3 + i]; w2[i] = a.w[(2 * a.D + cs) * 3 + i]; }
    const float b1 = a.b[a.D + cs], b2 = a.b[2 * a.D + cs];
    const size_t xb = (size_t)b * a.Lx * D3;
    const int t_begin = blockIdx.x * MIX_RUN;
    const int t_end = (t_begin + MIX_RUN < a.L) ? t_begin + MIX_RUN : a.L;
    float x1m2 = mix_ld<DT>(a.x, xb + (size_t)(t_begin - 2) * D3 + a.D + cs, cv && t_begin >= 2);
    float x1m1 = mix_ld<DT>(a.x, xb + (size_t)(t_begin - 1) * D3 + a.D + cs, cv && t_begin >= 1);
    float vm2 = mix_ld<DT>(a.x, xb + (size_t)(t_begin - 2) * D3 + 2 * a.D + cs, cv && t_begin >= 2);
    float vm1 = mix_ld<DT>(a.x, xb + (size_t)(t_begin - 1) * D3 + 2 * a.D + cs, cv && t_begin >= 1);
    float g1m1 = 0.f, g1m2 = 0.f, g2m1 = 0.f, g2m2 = 0.f;
    float dw1[3] = {0.f, 0.f, 0.f}, db1 = 0.f, dw2[3] = {0.f, 0.f, 0.f}, db2 = 0.f;
    for (int st = 0; st <= MIX_NT; ++st) {
        const int t0 = t_begin + st * MIX_T;
        if (t0 >= t_end + 2 || t0 >= a.L + 2) break;
        const int npos = (st < MIX_NT) ? MIX_T : 2;
        {
            const int t = t0 + lane;
            for (int cc = 0; cc < 64; ++cc)
                tg[cc * 65 + lane] = mix_ld<DT>(a.a0, ((size_t)b * a.D + c0 + cc) * a.L + t, t < a.L && c0 + cc < a.D);
        }
        __syncthreads();
        for (int p = 0; p < npos; ++p) {
            const int t = t0 + p;
            const bool in_l = cv && t < a.L;
            const bool own = t < t_end;
            const float x1t = mix_ld<DT>(a.x, xb + (size_t)t * D3 + a.D + cs, in_l);
            const float vt = mix_ld<DT>(a.x, xb + (size_t)t * D3 + 2 * a.D + cs, in_l);
            const float x1c = b1 + w1[0] * x1m2 + w1[1] * x1m1 + w1[2] * x1t;
            const float vc = b2 + w2[0] * vm2 + w2[1] * vm1 + w2[2] * vt;
            const float dv = in_l ? tg[lane * 65 + p] : 0.f;
            const float g1 = dv * vc, g2 = dv * x1c;
            if (own) {
                dw1[0] += g1 * x1m2; dw1[1] += g1 * x1m1; dw1[2] += g1 * x1t; db1 += g1;
                dw2[0] += g2 * vm2; dw2[1] += g2 * vm1; dw2[2] += g2 * vt; db2 += g2;
            }
            const int td2 = t - 2;
            const bool emit = cv && td2 >= t_begin && td2 < t_end;
            mix_st<DT>(a.dx, xb + (size_t)td2 * D3 + a.D + cs, emit, w1[0] * g1 + w1[1] * g1m1 + w1[2] * g1m2);
            mix_st<DT>(a.dx, xb + (size_t)td2 * D3 + 2 * a.D + cs, emit, w2[0] * g2 + w2[1] * g2m1 + w2[2] * g2m2);
            g1m2 = g1m1; g1m1 = g1; g2m2 = g2m1; g2m1 = g2;
            x1m2 = x1m1; x1m1 = x1t; vm2 = vm1; vm1 = vt;
        }
        __syncthreads();
    }
    if (cv) {
        float* p1 = a.part + (((size_t)b * gridDim.x + blockIdx.x) * D3 + a.D + c) * 4;
        float* p2 = a.part + (((size_t)b * gridDim.x + blockIdx.x) * D3 + 2 * a.D + c) * 4;
        p1[0] = dw1[0]; p1[1] = dw1[1]; p1[2] = dw1[2]; p1[3] = db1;
        p2[0] = dw2[0]; p2[1] = dw2[1]; p2[2] = dw2[2]; p2[3] = db2;
    }
}

// =====================================================================================================================
// Wide-access versions (used when D is a multiple of 64): workgroups of 256 threads stage 64-position x 64-channel tiles
// through LDS with 16-byte global accesses on BOTH layouts (8 bf16 / 4 fp32 per lane; rows of 128 / 256 bytes), compute
// from LDS with one thread per (channel, every 4th position), and keep the short filter's gradient partials in
// registers over a run of MIX_NT tiles.  The single-wavefront kernels above (2-byte accesses, sequential windows)
// reached 0.7-2.6 TB/s; they remain the general-D fallback.
// =====================================================================================================================
enum { MW_TP = 64, MW_TC = 64, MW_THREADS = 256, MW_CS = MW_TC + 1 };     // CS: LDS row stride of channel-minor tiles

template <int DT> struct MixVec { enum { N = (DT == DT_F32) ? 4 : 8 }; };

template <int DT>
__device__ __forceinline__ void vec_load(const typename Elem<DT>::type* p, float (&out)[MixVec<DT>::N]) {
    struct __attribute__((aligned(4))) Raw { uint32_t w[4]; } r;
    __builtin_memcpy(&r, p, 16);                                   // one 16-byte access (may be under-aligned)
    if constexpr (DT == DT_F32) {
        HY_UNROLL
        for (int e = 0; e < 4; ++e) out[e] = u2f(r.w[e]);
    } else {
        HY_UNROLL
        for (int e = 0; e < 4; ++e) {
            const c32 v = Pair<DT>::cvt(r.w[e]);
            out[2 * e] = v.x;
            out[2 * e + 1] = v.y;
        }
    }
}
template <int DT>
__device__ __forceinline__ void vec_store(typename Elem<DT>::type* p, const float (&in)[MixVec<DT>::N]) {
    struct __attribute__((aligned(4))) Raw { uint32_t w[4]; } r;
    if constexpr (DT == DT_F32) {
        HY_UNROLL
        for (int e = 0; e < 4; ++e) r.w[e] = f2u(in[e]);
    } else {
        HY_UNROLL
        for (int e = 0; e < 4; ++e) r.w[e] = Pair<DT>::pack(mk(in[2 * e], in[2 * e + 1]));
    }
    __builtin_memcpy(p, &r, 16);
}

// store rows (positions t_first ..) of a channel-minor LDS tile, only positions in [t_lo, t_hi)
template <int DT>
__device__ __forceinline__ void store_cm_tile(const HY_LDS float* src, void* base, size_t boff, int ld, int coff, int t_first,
                                              int nrows, int t_lo, int t_hi, int tid) {
    typedef typename Elem<DT>::type elem_t;
    constexpr int N = MixVec<DT>::N, VPR = MW_TC / N;
    elem_t* dst = reinterpret_cast<elem_t*>(base) + boff + coff;
    HY_UNROLL
    for (int i = tid; i < nrows * VPR; i += MW_THREADS) {
        const int row = i / VPR, vl = i % VPR, t = t_first + row;
        if (t >= t_lo && t < t_hi) {
            float v[N];
            HY_UNROLL
            for (int e = 0; e < N; ++e) v[e] = src[row * MW_CS + vl * N + e];
            vec_store<DT>(dst + (size_t)t * ld + vl * N, v);
        }
    }
}
// (B, D, L)-side tile store: rows = 64 channels of batch item b (row0 = b * D + c0), positions t0 .. t0 + npos - 1 from
// LDS src[c * ds + p]; only positions < t_hi are written.
template <int DT>
__device__ __forceinline__ void store_dl_tile(const HY_LDS float* src, int ds, void* base, size_t row0, int L, int t0, int npos,
                                              int t_hi, int tid) {
    typedef typename Elem<DT>::type elem_t;
    constexpr int N = MixVec<DT>::N;
    const int vpr = npos / N;
    elem_t* dst = reinterpret_cast<elem_t*>(base);
    HY_UNROLL
    for (int i = tid; i < MW_TC * vpr; i += MW_THREADS) {
        const int c = i / vpr, vl = i % vpr, t = t0 + vl * N;
        elem_t* rp = dst + (row0 + c) * (size_t)L;
        float v[N];
        HY_UNROLL
        for (int e = 0; e < N; ++e) v[e] = src[c * ds + vl * N + e];
        if (t + N <= t_hi) {
            vec_store<DT>(rp + t, v);
        } else {
            HY_UNROLL
            for (int e = 0; e < N; ++e)
                if (t + e < t_hi) Elem<DT>::st(rp + t + e, v[e]);
        }
    }
}

// ---- two-phase tile movement: fetch = global -> registers (raw 16-byte vectors, every load issued back to back),
// commit = registers -> LDS as fp32.  A kernel fetches tile s+1 right after committing tile s, so the loads are in
// flight while it computes and stores tile s (the single-phase movers above expose the full load latency per tile).
struct RawVec { uint32_t w[4]; };

template <int DT>
__device__ __forceinline__ void raw_to_f32(const RawVec& r, float (&out)[MixVec<DT>::N]) {
    if constexpr (DT == DT_F32) {
        HY_UNROLL
        for (int e = 0; e < 4; ++e) out[e] = u2f(r.w[e]);
    } else {
        HY_UNROLL
        for (int e = 0; e < 4; ++e) {
            const c32 v = Pair<DT>::cvt(r.w[e]);
            out[2 * e] = v.x;
            out[2 * e + 1] = v.y;
        }
    }
}

template <int DT, int NROWS>
struct CmFetch {
    static constexpr int N = MixVec<DT>::N, VPR = MW_TC / N, TOTAL = NROWS * VPR, CNT = (TOTAL + MW_THREADS - 1) / MW_THREADS;
    RawVec v[CNT];
};
template <int DT, int NROWS>
__device__ __forceinline__ void fetch_cm(CmFetch<DT, NROWS>& f, const void* base, size_t boff, int ld, int coff, int t_first,
                                         int tlim, int tid) {
    typedef typename Elem<DT>::type elem_t;
    typedef CmFetch<DT, NROWS> F;
    const elem_t* src = reinterpret_cast<const elem_t*>(base) + boff + coff;
    HY_UNROLL
    for (int j = 0; j < F::CNT; ++j) {
        const int i = tid + j * MW_THREADS;
        const int row = i / F::VPR, vl = i % F::VPR, t = t_first + row;
        const bool ok = i < F::TOTAL && t >= 0 && t < tlim;
        __builtin_memcpy(&f.v[j], src + (size_t)(ok ? t : 0) * ld + vl * F::N, 16);
    }
}
template <int DT, int NROWS>
__device__ __forceinline__ void commit_cm(HY_LDS float* dst, const CmFetch<DT, NROWS>& f, int t_first, int tlim, int tid) {
    typedef CmFetch<DT, NROWS> F;
    HY_UNROLL
    for (int j = 0; j < F::CNT; ++j) {
        const int i = tid + j * MW_THREADS;
        const int row = i / F::VPR, vl = i % F::VPR, t = t_first + row;
        if (i < F::TOTAL) {
            const bool ok = t >= 0 && t < tlim;
            float v[F::N];
            raw_to_f32<DT>(f.v[j], v);
            HY_UNROLL
            for (int e = 0; e < F::N; ++e) dst[row * MW_CS + vl * F::N + e] = ok ? v[e] : 0.f;
        }
    }
}

template <int DT, int NPOS>
struct DlFetch {
    static constexpr int N = MixVec<DT>::N, VPR = NPOS / N, TOTAL = MW_TC * VPR, CNT = (TOTAL + MW_THREADS - 1) / MW_THREADS;
    RawVec v[CNT];
};
template <int DT, int NPOS>
__device__ __forceinline__ void fetch_dl(DlFetch<DT, NPOS>& f, const void* base, size_t row0, int L, int t0, int tid) {
    typedef typename Elem<DT>::type elem_t;
    typedef DlFetch<DT, NPOS> F;
    const elem_t* src = reinterpret_cast<const elem_t*>(base);
    HY_UNROLL
    for (int j = 0; j < F::CNT; ++j) {
        const int i = tid + j * MW_THREADS;
        const int c = (i / F::VPR) & (MW_TC - 1), vl = i % F::VPR, t = t0 + vl * F::N;
        const elem_t* rp = src + (row0 + c) * (size_t)L;
        // a vector that would straddle the end of the row is fetched from the last full vector of the row instead and
        // re-aligned at commit time (rows shorter than one vector take the element-wise path there)
        const int ts = t + F::N <= L ? t : (L >= F::N ? L - F::N : 0);
        if (L >= F::N) __builtin_memcpy(&f.v[j], rp + ts, 16);
        else {
            HY_UNROLL
            for (int e = 0; e < 4; ++e) f.v[j].w[e] = 0u;
        }
    }
}
template <int DT, int NPOS>
__device__ __forceinline__ void commit_dl(HY_LDS float* dst, int ds, const DlFetch<DT, NPOS>& f, const void* base, size_t row0,
                                          int L, int t0, int tid) {
    typedef typename Elem<DT>::type elem_t;
    typedef DlFetch<DT, NPOS> F;
    HY_UNROLL
    for (int j = 0; j < F::CNT; ++j) {
        const int i = tid + j * MW_THREADS;
        const int c = i / F::VPR, vl = i % F::VPR, t = t0 + vl * F::N;
        if (i < F::TOTAL) {
            float v[F::N];
            raw_to_f32<DT>(f.v[j], v);
            if (t + F::N <= L) {
                HY_UNROLL
                for (int e = 0; e < F::N; ++e) dst[c * ds + vl * F::N + e] = v[e];
            } else if (L >= F::N) {
                // fetched from [L - N, L): element e of the wanted vector (position t + e) sits at index t + e - (L - N)
                const int sh = t - (L - F::N);
                HY_UNROLL
                for (int e = 0; e < F::N; ++e) {
                    float x = 0.f;
                    HY_UNROLL
                    for (int q = 0; q < F::N; ++q) x = (q == e + sh) ? v[q] : x;
                    dst[c * ds + vl * F::N + e] = (t + e < L) ? x : 0.f;
                }
            } else {
                const elem_t* rp = reinterpret_cast<const elem_t*>(base) + (row0 + c) * (size_t)L;
                HY_UNROLL
                for (int e = 0; e < F::N; ++e) dst[c * ds + vl * F::N + e] = (t + e < L) ? Elem<DT>::ld(rp + t + e) : 0.f;
            }
        }
    }
}

// short-conv value at tile row p (row index = position - first_position_of_tile) from a channel-minor LDS tile whose
// row r holds position (first + r): xc(t) = b + w0 x(t-2) + w1 x(t-1) + w2 x(t), with x(t) in row `r2` and x(t-2) in r2-2
__device__ __forceinline__ float sconv(const HY_LDS float* xs, int r2, int c, const float (&w)[3], float b) {
    return b + w[0] * xs[(r2 - 2) * MW_CS + c] + w[1] * xs[(r2 - 1) * MW_CS + c] + w[2] * xs[r2 * MW_CS + c];
}

// tiles of 64 positions in the run of MIX_RUN positions that starts at t_begin
__device__ __forceinline__ int mw_tiles(int L, int t_begin) {
    const int n = (L - t_begin + MW_TP - 1) / MW_TP;
    return n < MIX_NT ? n : MIX_NT;
}

template <int DT>
__global__ void __launch_bounds__(MW_THREADS) mixer_pre_fwd_wide_kernel(MixArgs a) {
    HY_SMEM(smem);
    HY_LDS float* xs1 = HY_LDS_CAST(float, smem);                  // [66][65]  x1 rows t0-2 .. t0+63
    HY_LDS float* xs2 = xs1 + (MW_TP + 2) * MW_CS;                 // [66][65]  v
    HY_LDS float* vt = xs2 + (MW_TP + 2) * MW_CS;                  // [64][65]  vg tile, [c][p]
    const int tid = threadIdx.x, c = tid & 63, pq = tid >> 6;
    const int c0 = blockIdx.y * 64, b = blockIdx.z, D3 = 3 * a.D;
    float w1[3], w2[3];
    HY_UNROLL
    for (int i = 0; i < 3; ++i) { w1[i] = a.w[(a.D + c0 + c) * 3 + i]; w2[i] = a.w[(2 * a.D + c0 + c) * 3 + i]; }
    const float b1 = a.b[a.D + c0 + c], b2 = a.b[2 * a.D + c0 + c];
    const size_t xb = (size_t)b * a.Lx * D3;
    const int t_begin = blockIdx.x * MIX_RUN;
    const int nst = mw_tiles(a.L, t_begin);
    CmFetch<DT, MW_TP + 2> f1, f2;
    fetch_cm(f1, a.x, xb, D3, a.D + c0, t_begin - 2, a.L, tid);
    fetch_cm(f2, a.x, xb, D3, 2 * a.D + c0, t_begin - 2, a.L, tid);
    for (int st = 0; st < nst; ++st) {
        const int t0 = t_begin + st * MW_TP;
        commit_cm(xs1, f1, t0 - 2, a.L, tid);
        commit_cm(xs2, f2, t0 - 2, a.L, tid);
        __syncthreads();
        if (st + 1 < nst) {
            fetch_cm(f1, a.x, xb, D3, a.D + c0, t0 + MW_TP - 2, a.L, tid);
            fetch_cm(f2, a.x, xb, D3, 2 * a.D + c0, t0 + MW_TP - 2, a.L, tid);
        }
        HY_UNROLL
        for (int i = 0; i < MW_TP / 4; ++i) {
            const int p = pq + 4 * i;
            vt[c * MW_CS + p] = sconv(xs1, p + 2, c, w1, b1) * sconv(xs2, p + 2, c, w2, b2);
        }
        __syncthreads();
        store_dl_tile<DT>(vt, MW_CS, a.a0, (size_t)b * a.D + c0, a.L, t0, MW_TP, a.L, tid);
    }
}

template <int DT>
__global__ void __launch_bounds__(MW_THREADS) mixer_post_fwd_wide_kernel(MixArgs a) {
    HY_SMEM(smem);
    HY_LDS float* xs0 = HY_LDS_CAST(float, smem);                  // [66][65]
    HY_LDS float* yt = xs0 + (MW_TP + 2) * MW_CS;                  // [64][65]  y tile [c][p]
    HY_LDS float* zs = yt + MW_TC * MW_CS;                         // [64][65]  z tile [p][c]
    const int tid = threadIdx.x, c = tid & 63, pq = tid >> 6;
    const int c0 = blockIdx.y * 64, b = blockIdx.z, D3 = 3 * a.D;
    float w0[3];
    HY_UNROLL
    for (int i = 0; i < 3; ++i) w0[i] = a.w[(c0 + c) * 3 + i];
    const float b0 = a.b[c0 + c];
    const size_t xb = (size_t)b * a.Lx * D3;
    const int t_begin = blockIdx.x * MIX_RUN;
    const int nst = mw_tiles(a.L, t_begin);
    const size_t yrow = (size_t)b * a.D + c0;
    CmFetch<DT, MW_TP + 2> f0;
    DlFetch<DT, MW_TP> fy;
    fetch_cm(f0, a.x, xb, D3, c0, t_begin - 2, a.L, tid);
    fetch_dl(fy, a.a0, yrow, a.L, t_begin, tid);
    for (int st = 0; st < nst; ++st) {
        const int t0 = t_begin + st * MW_TP;
        commit_cm(xs0, f0, t0 - 2, a.L, tid);
        commit_dl(yt, MW_CS, fy, a.a0, yrow, a.L, t0, tid);
        __syncthreads();
        if (st + 1 < nst) {
            fetch_cm(f0, a.x, xb, D3, c0, t0 + MW_TP - 2, a.L, tid);
            fetch_dl(fy, a.a0, yrow, a.L, t0 + MW_TP, tid);
        }
        HY_UNROLL
        for (int i = 0; i < MW_TP / 4; ++i) {
            const int p = pq + 4 * i;
            zs[p * MW_CS + c] = sconv(xs0, p + 2, c, w0, b0) * yt[c * MW_CS + p];
        }
        __syncthreads();
        store_cm_tile<DT>(zs, a.a1, (size_t)b * a.L * a.D, a.D, c0, t0, MW_TP, 0, a.L, tid);
    }
}

// reduce the per-thread partials of a channel over its 4 position-phases and write them
__device__ __forceinline__ void write_partials(HY_LDS float* red, float* dst, const float (&dw)[3], float db, int c, int pq) {
    red[(pq * 64 + c) * 4 + 0] = dw[0];
    red[(pq * 64 + c) * 4 + 1] = dw[1];
    red[(pq * 64 + c) * 4 + 2] = dw[2];
    red[(pq * 64 + c) * 4 + 3] = db;
    __syncthreads();
    if (pq == 0) {
        HY_UNROLL
        for (int k = 0; k < 4; ++k)
            dst[c * 4 + k] = ((red[(0 * 64 + c) * 4 + k] + red[(1 * 64 + c) * 4 + k]) + red[(2 * 64 + c) * 4 + k]) + red[(3 * 64 + c) * 4 + k];
    }
    __syncthreads();
}

enum { MW_YS = 72 + 1 };   // LDS row stride of a (B, D, L)-side tile that carries 2 halo positions (72 = 66 rounded up to a vector)

template <int DT>
__global__ void __launch_bounds__(MW_THREADS) mixer_post_bwd_wide_kernel(MixArgs a) {
    HY_SMEM(smem);
    HY_LDS float* xs0 = HY_LDS_CAST(float, smem);                  // [68][65]  x0 rows t0-2 .. t0+65; later the dx tile
    HY_LDS float* gs = xs0 + (MW_TP + 4) * MW_CS;                  // [66][65]  dz rows t0 .. t0+65, then g = dz * y
    HY_LDS float* ys = gs + (MW_TP + 2) * MW_CS;                   // [64][73]  y tile [c][p], then dy
    const int tid = threadIdx.x, c = tid & 63, pq = tid >> 6;
    const int c0 = blockIdx.y * 64, b = blockIdx.z, D3 = 3 * a.D;
    float w0[3];
    HY_UNROLL
    for (int i = 0; i < 3; ++i) w0[i] = a.w[(c0 + c) * 3 + i];
    const float b0 = a.b[c0 + c];
    const size_t xb = (size_t)b * a.Lx * D3;
    const int t_begin = blockIdx.x * MIX_RUN;
    float dw[3] = {0.f, 0.f, 0.f}, db = 0.f;
    const int nst = mw_tiles(a.L, t_begin);
    const size_t yrow = (size_t)b * a.D + c0, zoff = (size_t)b * a.L * a.D;
    CmFetch<DT, MW_TP + 4> f0;
    CmFetch<DT, MW_TP + 2> fz;
    DlFetch<DT, 72> fy;
    fetch_cm(f0, a.x, xb, D3, c0, t_begin - 2, a.L, tid);
    fetch_cm(fz, a.a1, zoff, a.D, c0, t_begin, a.L, tid);
    fetch_dl(fy, a.a0, yrow, a.L, t_begin, tid);
    for (int st = 0; st < nst; ++st) {
        const int t0 = t_begin + st * MW_TP;
        commit_cm(xs0, f0, t0 - 2, a.L, tid);
        commit_cm(gs, fz, t0, a.L, tid);
        commit_dl(ys, MW_YS, fy, a.a0, yrow, a.L, t0, tid);
        __syncthreads();
        if (st + 1 < nst) {
            fetch_cm(f0, a.x, xb, D3, c0, t0 + MW_TP - 2, a.L, tid);
            fetch_cm(fz, a.a1, zoff, a.D, c0, t0 + MW_TP, a.L, tid);
            fetch_dl(fy, a.a0, yrow, a.L, t0 + MW_TP, tid);
        }
        for (int p = pq; p < MW_TP + 2; p += 4) {
            const float dzv = gs[p * MW_CS + c];
            const float g = dzv * ys[c * MW_YS + p];
            gs[p * MW_CS + c] = g;
            if (p < MW_TP) {                                       // own position: dy, dw, db
                ys[c * MW_YS + p] = dzv * sconv(xs0, p + 2, c, w0, b0);
                dw[0] += g * xs0[p * MW_CS + c];
                dw[1] += g * xs0[(p + 1) * MW_CS + c];
                dw[2] += g * xs0[(p + 2) * MW_CS + c];
                db += g;
            }
        }
        __syncthreads();
        HY_UNROLL
        for (int i = 0; i < MW_TP / 4; ++i) {                      // dx(t) = w0 g(t+2) + w1 g(t+1) + w2 g(t)
            const int p = pq + 4 * i;
            xs0[p * MW_CS + c] = w0[0] * gs[(p + 2) * MW_CS + c] + w0[1] * gs[(p + 1) * MW_CS + c] + w0[2] * gs[p * MW_CS + c];
        }
        __syncthreads();
        store_dl_tile<DT>(ys, MW_YS, a.a2, (size_t)b * a.D + c0, a.L, t0, MW_TP, a.L, tid);
        store_cm_tile<DT>(xs0, a.dx, xb, D3, c0, t0, MW_TP, 0, a.L, tid);
        __syncthreads();
    }
    write_partials(gs, a.part + (((size_t)b * gridDim.x + blockIdx.x) * D3 + c0) * 4, dw, db, c, pq);
}

template <int DT>
__global__ void __launch_bounds__(MW_THREADS) mixer_pre_bwd_wide_kernel(MixArgs a) {
    HY_SMEM(smem);
    HY_LDS float* xs1 = HY_LDS_CAST(float, smem);                  // [68][65]  x1 rows t0-2 .. t0+65; later dx1
    HY_LDS float* xs2 = xs1 + (MW_TP + 4) * MW_CS;                 // [68][65]  v; later dx2
    HY_LDS float* ds = xs2 + (MW_TP + 4) * MW_CS;                  // [64][73]  dvg tile [c][p]; g1 overwrites it in place
    HY_LDS float* g2s = ds + MW_TC * MW_YS;                        // [64][73]  g2 [c][p]
    const int tid = threadIdx.x, c = tid & 63, pq = tid >> 6;
    const int c0 = blockIdx.y * 64, b = blockIdx.z, D3 = 3 * a.D;
    float w1[3], w2[3];
    HY_UNROLL
    for (int i = 0; i < 3; ++i) { w1[i] = a.w[(a.D + c0 + c) * 3 + i]; w2[i] = a.w[(2 * a.D + c0 + c) * 3 + i]; }
    const float b1 = a.b[a.D + c0 + c], b2 = a.b[2 * a.D + c0 + c];
    const size_t xb = (size_t)b * a.Lx * D3;
    const int t_begin = blockIdx.x * MIX_RUN;
    float dw1[3] = {0.f, 0.f, 0.f}, db1 = 0.f, dw2[3] = {0.f, 0.f, 0.f}, db2 = 0.f;
    const int nst = mw_tiles(a.L, t_begin);
    const size_t drow = (size_t)b * a.D + c0;
    CmFetch<DT, MW_TP + 4> f1, f2;
    DlFetch<DT, 72> fd;
    fetch_cm(f1, a.x, xb, D3, a.D + c0, t_begin - 2, a.L, tid);
    fetch_cm(f2, a.x, xb, D3, 2 * a.D + c0, t_begin - 2, a.L, tid);
    fetch_dl(fd, a.a0, drow, a.L, t_begin, tid);
    for (int st = 0; st < nst; ++st) {
        const int t0 = t_begin + st * MW_TP;
        commit_cm(xs1, f1, t0 - 2, a.L, tid);
        commit_cm(xs2, f2, t0 - 2, a.L, tid);
        commit_dl(ds, MW_YS, fd, a.a0, drow, a.L, t0, tid);
        __syncthreads();
        if (st + 1 < nst) {
            fetch_cm(f1, a.x, xb, D3, a.D + c0, t0 + MW_TP - 2, a.L, tid);
            fetch_cm(f2, a.x, xb, D3, 2 * a.D + c0, t0 + MW_TP - 2, a.L, tid);
            fetch_dl(fd, a.a0, drow, a.L, t0 + MW_TP, tid);
        }
        for (int p = pq; p < MW_TP + 2; p += 4) {
            const float dv = ds[c * MW_YS + p];                    // zero for positions >= L
            const float g1 = dv * sconv(xs2, p + 2, c, w2, b2);    // gradient of x1c = dvg * vc
            const float g2 = dv * sconv(xs1, p + 2, c, w1, b1);    // gradient of vc  = dvg * x1c
            ds[c * MW_YS + p] = g1;
            g2s[c * MW_YS + p] = g2;
            if (p < MW_TP) {
                dw1[0] += g1 * xs1[p * MW_CS + c]; dw1[1] += g1 * xs1[(p + 1) * MW_CS + c]; dw1[2] += g1 * xs1[(p + 2) * MW_CS + c];
                db1 += g1;
                dw2[0] += g2 * xs2[p * MW_CS + c]; dw2[1] += g2 * xs2[(p + 1) * MW_CS + c]; dw2[2] += g2 * xs2[(p + 2) * MW_CS + c];
                db2 += g2;
            }
        }
        __syncthreads();
        HY_UNROLL
        for (int i = 0; i < MW_TP / 4; ++i) {
            const int p = pq + 4 * i;
            xs1[p * MW_CS + c] = w1[0] * ds[c * MW_YS + p + 2] + w1[1] * ds[c * MW_YS + p + 1] + w1[2] * ds[c * MW_YS + p];
            xs2[p * MW_CS + c] = w2[0] * g2s[c * MW_YS + p + 2] + w2[1] * g2s[c * MW_YS + p + 1] + w2[2] * g2s[c * MW_YS + p];
        }
        __syncthreads();
        store_cm_tile<DT>(xs1, a.dx, xb, D3, a.D + c0, t0, MW_TP, 0, a.L, tid);
        store_cm_tile<DT>(xs2, a.dx, xb, D3, 2 * a.D + c0, t0, MW_TP, 0, a.L, tid);
        __syncthreads();
    }
    float* base = a.part + ((size_t)b * gridDim.x + blockIdx.x) * D3 * 4;
    write_partials(g2s, base + (size_t)(a.D + c0) * 4, dw1, db1, c, pq);
    write_partials(g2s, base + (size_t)(2 * a.D + c0) * 4, dw2, db2, c, pq);
}

}  // namespace hyena
