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

}  // namespace hyena
