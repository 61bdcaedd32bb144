// onchip_kernels.h -- workspace-free long convolution for M <= 32768: one workgroup per (b, d) row, the whole
// transform lives in registers and LDS, HBM traffic = the operator's algorithmic bytes (+ the filter spectrum).
//
// What is computed: the same fftconv_ref (src/models/sequence/hyena.py:59-88) as fftconv_kernels.h, for rows short
// enough to fit one CU (the reference's own fused kernel has this shape -- one block per (b, h) row, grid (B, H),
// csrc/fftconv/fftconv_cuda.cu:805 -- but stops at fft_size 16384 / L 8192 and needs even L; nothing of its cuFFTDx
// structure is used here).
//
// Transform: "right-angle" (negacyclic) convolution.  For a real row x of length L <= M put
//     c[n] = x[n] e^(-i pi n / 2M),  n < M          (zero beyond L),         X = FFT_M(c)
// X[m] is bin 2m of the odd-frequency DFT of the zero-padded length-2M real row; the remaining (odd) bins are complex
// conjugates, so X alone carries the whole spectrum and a negacyclic convolution of length 2M -- which IS the linear
// convolution here, because 2L - 1 <= 2M never wraps -- is the plain elementwise product
//     out = Re( IFFT_M( X_u .* X_k ) e^(+i pi n / 2M) ),      du, dk: the same with conj(X_k), conj(X_u).
// There is no real/complex unpacking step, hence no partner index (M - k), no self-paired rows and no exchange for
// the product: every thread multiplies the 32 spectrum values it already holds.  (Numerically checked against
// numpy in scratch models and by tests/ against the oracle.)
//
// Decomposition M = 32 x 32 x R (R = 1, 2, 4, ..., 32), T = 32 R threads per row, 32 points per thread in every pass:
//     pass 1   thread t:          radix-32 DFT over s of c[t + T s]            -> index ka;  x w_M^(t (ka + 1/4))
//     exchange 1 (LDS, workgroup)
//     pass 2   thread (ka, t'):   radix-32 DFT over s' of y[ka][R s' + t']     -> index kb1; x w_T^(t' kb1)
//     exchange 2 (LDS, inside groups of R adjacent lanes: no workgroup barrier)
//     pass 3   thread (ka, g):    32/R radix-R DFTs over t' for kb1 = g + R i  -> index kb2
// Result X[ka + 32 kb1 + 1024 kb2] sits in register q = i R + kb2 of thread tau = ka R + g; spectra in memory use
// exactly that order (element q T + tau), so filter loads are coalesced and need no permutation.  The inverse runs
// the three passes backwards with conjugated twiddles.  The e^(-i pi n / 2M) twist costs nothing in pass 1: its
// t-part is folded into the twiddle (the "+ 1/4"), its s-part is a compile-time constant per register.
//
// LDS: exchange buffer of 32 x 33 R floats per row (one plane at a time -- real parts, then imaginary parts: 135 KB at
// R = 32), padded so that every access is bank-conflict free (see x1p / x2p).
//
// Compiled by hipcc for gfx950 (product) and, with -DHIPEMU, by g++ against tests/hipemu (tests only).
#pragma once
#define HY_HELPERS_ONLY
#include "fftconv_kernels.h"

namespace hyena {
namespace oc {

#ifdef HIPEMU
#define HY_WAVE_SYNC() hipemu::yield(2)
#else
// LDS operations of one wavefront execute in order, so data exchanged between lanes of the same wavefront needs no
// workgroup barrier; this only stops the compiler from moving LDS accesses across the exchange point.
#define HY_WAVE_SYNC() __builtin_amdgcn_wave_barrier()
#endif

// -DOC_PROFILE (profiling builds only, scripts/build_variant.sh): wavefront time stamps (s_memtime: shader clock) at the phase boundaries of
// conv_kernel, written by lane 0 of every wavefront to a buffer the host hands over through hyena_oc_prof_set (onchip.hip);
// scripts/oc_phase_profile.py turns them into a per-phase timeline.  The product build contains none of it.
#if defined(OC_PROFILE) && !defined(HIPEMU)
__device__ unsigned long long* oc_prof_buf = nullptr;
struct Prof { unsigned long long t[32]; };
#define OC_MARK(c, i)                                                                                               \
    do {                                                                                                            \
        __builtin_amdgcn_sched_barrier(0);                                                                          \
        if ((c).prof != nullptr) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"((c).prof->t[i])::"memory"); \
        __builtin_amdgcn_sched_barrier(0);                                                                          \
    } while (0)
#else
#define OC_MARK(c, i) do {} while (0)
#endif

// cos / sin (2 pi j / 256): the per-register part of the input twist, e^(-2 pi i s phi / 32) with phi = PHI8 / 8
HY_CONST_TABLE float OC_COS256[256] = {1.000000000e+00f, 9.996988177e-01f, 9.987954497e-01f, 9.972904325e-01f, 9.951847196e-01f, 9.924795628e-01f, 9.891765118e-01f, 9.852776527e-01f, 9.807852507e-01f, 9.757021070e-01f, 9.700312614e-01f, 9.637760520e-01f, 9.569403529e-01f, 9.495281577e-01f, 9.415440559e-01f, 9.329928160e-01f, 9.238795042e-01f, 9.142097831e-01f, 9.039893150e-01f, 8.932242990e-01f, 8.819212914e-01f, 8.700869679e-01f, 8.577286005e-01f, 8.448535800e-01f, 8.314695954e-01f, 8.175848126e-01f, 8.032075167e-01f, 7.883464098e-01f, 7.730104327e-01f, 7.572088242e-01f, 7.409511209e-01f, 7.242470980e-01f, 7.071067691e-01f, 6.895405650e-01f, 6.715589762e-01f, 6.531728506e-01f, 6.343932748e-01f, 6.152315736e-01f, 5.956993103e-01f, 5.758081675e-01f, 5.555702448e-01f, 5.349976420e-01f, 5.141027570e-01f, 4.928981960e-01f, 4.713967443e-01f, 4.496113360e-01f, 4.275550842e-01f, 4.052413106e-01f, 3.826834261e-01f, 3.598950505e-01f, 3.368898630e-01f, 3.136817515e-01f, 2.902846634e-01f, 2.667127550e-01f, 2.429801822e-01f, 2.191012353e-01f, 1.950903237e-01f, 1.709618866e-01f, 1.467304677e-01f, 1.224106774e-01f, 9.801714122e-02f, 7.356456667e-02f, 4.906767607e-02f, 2.454122901e-02f, 6.123234263e-17f, -2.454122901e-02f, -4.906767607e-02f, -7.356456667e-02f, -9.801714122e-02f, -1.224106774e-01f, -1.467304677e-01f, -1.709618866e-01f, -1.950903237e-01f, -2.191012353e-01f, -2.429801822e-01f, -2.667127550e-01f, -2.902846634e-01f, -3.136817515e-01f, -3.368898630e-01f, -3.598950505e-01f, -3.826834261e-01f, -4.052413106e-01f, -4.275550842e-01f, -4.496113360e-01f, -4.713967443e-01f, -4.928981960e-01f, -5.141027570e-01f, -5.349976420e-01f, -5.555702448e-01f, -5.758081675e-01f, -5.956993103e-01f, -6.152315736e-01f, -6.343932748e-01f, -6.531728506e-01f, -6.715589762e-01f, -6.895405650e-01f, -7.071067691e-01f, -7.242470980e-01f, -7.409511209e-01f, -7.572088242e-01f, -7.730104327e-01f, -7.883464098e-01f, -8.032075167e-01f, -8.175848126e-01f, -8.314695954e-01f, -8.448535800e-01f, -8.577286005e-01f, -8.700869679e-01f, -8.819212914e-01f, -8.932242990e-01f, -9.039893150e-01f, -9.142097831e-01f, -9.238795042e-01f, -9.329928160e-01f, -9.415440559e-01f, -9.495281577e-01f, -9.569403529e-01f, -9.637760520e-01f, -9.700312614e-01f, -9.757021070e-01f, -9.807852507e-01f, -9.852776527e-01f, -9.891765118e-01f, -9.924795628e-01f, -9.951847196e-01f, -9.972904325e-01f, -9.987954497e-01f, -9.996988177e-01f, -1.000000000e+00f, -9.996988177e-01f, -9.987954497e-01f, -9.972904325e-01f, -9.951847196e-01f, -9.924795628e-01f, -9.891765118e-01f, -9.852776527e-01f, -9.807852507e-01f, -9.757021070e-01f, -9.700312614e-01f, -9.637760520e-01f, -9.569403529e-01f, -9.495281577e-01f, -9.415440559e-01f, -9.329928160e-01f, -9.238795042e-01f, -9.142097831e-01f, -9.039893150e-01f, -8.932242990e-01f, -8.819212914e-01f, -8.700869679e-01f, -8.577286005e-01f, -8.448535800e-01f, -8.314695954e-01f, -8.175848126e-01f, -8.032075167e-01f, -7.883464098e-01f, -7.730104327e-01f, -7.572088242e-01f, -7.409511209e-01f, -7.242470980e-01f, -7.071067691e-01f, -6.895405650e-01f, -6.715589762e-01f, -6.531728506e-01f, -6.343932748e-01f, -6.152315736e-01f, -5.956993103e-01f, -5.758081675e-01f, -5.555702448e-01f, -5.349976420e-01f, -5.141027570e-01f, -4.928981960e-01f, -4.713967443e-01f, -4.496113360e-01f, -4.275550842e-01f, -4.052413106e-01f, -3.826834261e-01f, -3.598950505e-01f, -3.368898630e-01f, -3.136817515e-01f, -2.902846634e-01f, -2.667127550e-01f, -2.429801822e-01f, -2.191012353e-01f, -1.950903237e-01f, -1.709618866e-01f, -1.467304677e-01f, -1.224106774e-01f, -9.801714122e-02f, -7.356456667e-02f, -4.906767607e-02f, -2.454122901e-02f, -1.836970147e-16f, 2.454122901e-02f, 4.906767607e-02f, 7.356456667e-02f, 9.801714122e-02f, 1.224106774e-01f, 1.467304677e-01f, 1.709618866e-01f, 1.950903237e-01f, 2.191012353e-01f, 2.429801822e-01f, 2.667127550e-01f, 2.902846634e-01f, 3.136817515e-01f, 3.368898630e-01f, 3.598950505e-01f, 3.826834261e-01f, 4.052413106e-01f, 4.275550842e-01f, 4.496113360e-01f, 4.713967443e-01f, 4.928981960e-01f, 5.141027570e-01f, 5.349976420e-01f, 5.555702448e-01f, 5.758081675e-01f, 5.956993103e-01f, 6.152315736e-01f, 6.343932748e-01f, 6.531728506e-01f, 6.715589762e-01f, 6.895405650e-01f, 7.071067691e-01f, 7.242470980e-01f, 7.409511209e-01f, 7.572088242e-01f, 7.730104327e-01f, 7.883464098e-01f, 8.032075167e-01f, 8.175848126e-01f, 8.314695954e-01f, 8.448535800e-01f, 8.577286005e-01f, 8.700869679e-01f, 8.819212914e-01f, 8.932242990e-01f, 9.039893150e-01f, 9.142097831e-01f, 9.238795042e-01f, 9.329928160e-01f, 9.415440559e-01f, 9.495281577e-01f, 9.569403529e-01f, 9.637760520e-01f, 9.700312614e-01f, 9.757021070e-01f, 9.807852507e-01f, 9.852776527e-01f, 9.891765118e-01f, 9.924795628e-01f, 9.951847196e-01f, 9.972904325e-01f, 9.987954497e-01f, 9.996988177e-01f};
HY_CONST_TABLE float OC_SIN256[256] = {0.000000000e+00f, 2.454122901e-02f, 4.906767607e-02f, 7.356456667e-02f, 9.801714122e-02f, 1.224106774e-01f, 1.467304677e-01f, 1.709618866e-01f, 1.950903237e-01f, 2.191012353e-01f, 2.429801822e-01f, 2.667127550e-01f, 2.902846634e-01f, 3.136817515e-01f, 3.368898630e-01f, 3.598950505e-01f, 3.826834261e-01f, 4.052413106e-01f, 4.275550842e-01f, 4.496113360e-01f, 4.713967443e-01f, 4.928981960e-01f, 5.141027570e-01f, 5.349976420e-01f, 5.555702448e-01f, 5.758081675e-01f, 5.956993103e-01f, 6.152315736e-01f, 6.343932748e-01f, 6.531728506e-01f, 6.715589762e-01f, 6.895405650e-01f, 7.071067691e-01f, 7.242470980e-01f, 7.409511209e-01f, 7.572088242e-01f, 7.730104327e-01f, 7.883464098e-01f, 8.032075167e-01f, 8.175848126e-01f, 8.314695954e-01f, 8.448535800e-01f, 8.577286005e-01f, 8.700869679e-01f, 8.819212914e-01f, 8.932242990e-01f, 9.039893150e-01f, 9.142097831e-01f, 9.238795042e-01f, 9.329928160e-01f, 9.415440559e-01f, 9.495281577e-01f, 9.569403529e-01f, 9.637760520e-01f, 9.700312614e-01f, 9.757021070e-01f, 9.807852507e-01f, 9.852776527e-01f, 9.891765118e-01f, 9.924795628e-01f, 9.951847196e-01f, 9.972904325e-01f, 9.987954497e-01f, 9.996988177e-01f, 1.000000000e+00f, 9.996988177e-01f, 9.987954497e-01f, 9.972904325e-01f, 9.951847196e-01f, 9.924795628e-01f, 9.891765118e-01f, 9.852776527e-01f, 9.807852507e-01f, 9.757021070e-01f, 9.700312614e-01f, 9.637760520e-01f, 9.569403529e-01f, 9.495281577e-01f, 9.415440559e-01f, 9.329928160e-01f, 9.238795042e-01f, 9.142097831e-01f, 9.039893150e-01f, 8.932242990e-01f, 8.819212914e-01f, 8.700869679e-01f, 8.577286005e-01f, 8.448535800e-01f, 8.314695954e-01f, 8.175848126e-01f, 8.032075167e-01f, 7.883464098e-01f, 7.730104327e-01f, 7.572088242e-01f, 7.409511209e-01f, 7.242470980e-01f, 7.071067691e-01f, 6.895405650e-01f, 6.715589762e-01f, 6.531728506e-01f, 6.343932748e-01f, 6.152315736e-01f, 5.956993103e-01f, 5.758081675e-01f, 5.555702448e-01f, 5.349976420e-01f, 5.141027570e-01f, 4.928981960e-01f, 4.713967443e-01f, 4.496113360e-01f, 4.275550842e-01f, 4.052413106e-01f, 3.826834261e-01f, 3.598950505e-01f, 3.368898630e-01f, 3.136817515e-01f, 2.902846634e-01f, 2.667127550e-01f, 2.429801822e-01f, 2.191012353e-01f, 1.950903237e-01f, 1.709618866e-01f, 1.467304677e-01f, 1.224106774e-01f, 9.801714122e-02f, 7.356456667e-02f, 4.906767607e-02f, 2.454122901e-02f, 1.224646853e-16f, -2.454122901e-02f, -4.906767607e-02f, -7.356456667e-02f, -9.801714122e-02f, -1.224106774e-01f, -1.467304677e-01f, -1.709618866e-01f, -1.950903237e-01f, -2.191012353e-01f, -2.429801822e-01f, -2.667127550e-01f, -2.902846634e-01f, -3.136817515e-01f, -3.368898630e-01f, -3.598950505e-01f, -3.826834261e-01f, -4.052413106e-01f, -4.275550842e-01f, -4.496113360e-01f, -4.713967443e-01f, -4.928981960e-01f, -5.141027570e-01f, -5.349976420e-01f, -5.555702448e-01f, -5.758081675e-01f, -5.956993103e-01f, -6.152315736e-01f, -6.343932748e-01f, -6.531728506e-01f, -6.715589762e-01f, -6.895405650e-01f, -7.071067691e-01f, -7.242470980e-01f, -7.409511209e-01f, -7.572088242e-01f, -7.730104327e-01f, -7.883464098e-01f, -8.032075167e-01f, -8.175848126e-01f, -8.314695954e-01f, -8.448535800e-01f, -8.577286005e-01f, -8.700869679e-01f, -8.819212914e-01f, -8.932242990e-01f, -9.039893150e-01f, -9.142097831e-01f, -9.238795042e-01f, -9.329928160e-01f, -9.415440559e-01f, -9.495281577e-01f, -9.569403529e-01f, -9.637760520e-01f, -9.700312614e-01f, -9.757021070e-01f, -9.807852507e-01f, -9.852776527e-01f, -9.891765118e-01f, -9.924795628e-01f, -9.951847196e-01f, -9.972904325e-01f, -9.987954497e-01f, -9.996988177e-01f, -1.000000000e+00f, -9.996988177e-01f, -9.987954497e-01f, -9.972904325e-01f, -9.951847196e-01f, -9.924795628e-01f, -9.891765118e-01f, -9.852776527e-01f, -9.807852507e-01f, -9.757021070e-01f, -9.700312614e-01f, -9.637760520e-01f, -9.569403529e-01f, -9.495281577e-01f, -9.415440559e-01f, -9.329928160e-01f, -9.238795042e-01f, -9.142097831e-01f, -9.039893150e-01f, -8.932242990e-01f, -8.819212914e-01f, -8.700869679e-01f, -8.577286005e-01f, -8.448535800e-01f, -8.314695954e-01f, -8.175848126e-01f, -8.032075167e-01f, -7.883464098e-01f, -7.730104327e-01f, -7.572088242e-01f, -7.409511209e-01f, -7.242470980e-01f, -7.071067691e-01f, -6.895405650e-01f, -6.715589762e-01f, -6.531728506e-01f, -6.343932748e-01f, -6.152315736e-01f, -5.956993103e-01f, -5.758081675e-01f, -5.555702448e-01f, -5.349976420e-01f, -5.141027570e-01f, -4.928981960e-01f, -4.713967443e-01f, -4.496113360e-01f, -4.275550842e-01f, -4.052413106e-01f, -3.826834261e-01f, -3.598950505e-01f, -3.368898630e-01f, -3.136817515e-01f, -2.902846634e-01f, -2.667127550e-01f, -2.429801822e-01f, -2.191012353e-01f, -1.950903237e-01f, -1.709618866e-01f, -1.467304677e-01f, -1.224106774e-01f, -9.801714122e-02f, -7.356456667e-02f, -4.906767607e-02f, -2.454122901e-02f};

template <int PHI8>
__device__ __forceinline__ c32 twist_const(int s) {          // e^(-2 pi i s PHI8 / 256)
    const int j = (s * PHI8) & 255;
    return mk(OC_COS256[j], -OC_SIN256[j]);
}

template <int R> struct Cfg {
    static constexpr int T = 32 * R;                 // threads per row
    static constexpr int M = 1024 * R;               // complex points per row
    static constexpr int NB = 32 / R;                // radix-R butterflies per thread in pass 3
    // The exchanges move one float plane at a time (real parts, then imaginary parts).  At R = 32 there is no choice (256 KB
    // of row data vs 160 KB of LDS); below that it halves the LDS per row and, measured, the registers the compiler needs
    // around an exchange -- the kernels then fit 128 VGPRs = 4 wavefronts per SIMD, worth 13-21 % at M = 4096 ... 16384
    // over complex64 exchanges at 2 wavefronts per SIMD (gpurun_out/r2v).
    static constexpr int XE = 32 * 33 * R;           // exchange buffer elements (floats) per row
    static constexpr size_t XBYTES = (size_t)XE * 4;
    static constexpr int ROW1 = T + R;               // exchange 1: position (ka, t) at ka * ROW1 + t
    static constexpr int GRP2 = 33 * R;              // exchange 2: group ka at ka * GRP2, (t', kb1) at t' * 33 + kb1
    // twiddle table layout (c32 units): tw1[11][T] | tw2[10][R]
    static constexpr int TW1 = 11 * T;
    static constexpr int TWN = TW1 + 10 * R;
};

// ---------------------------------------------------------------------------------------------
// twiddles.  Pass 1: w_M^(t (ka + phi)), ka = 8 a + b, as tA[a] * tB[b] with tB[b] = w_M^(t (b + phi)) (b = 0..7) and
// tA[a] = w_M^(8 t a) (a = 1..3): 11 table loads per thread instead of 32, one extra rounding in 24 of them.
// Pass 2: w_T^(t' kb1) likewise from tB[b] = w_T^(t' b) (b = 1..7), tA[a] = w_T^(8 t' a).
// ---------------------------------------------------------------------------------------------
struct Tw {
    c32 tB[8];
    c32 tA[4];   // [0] unused
};
template <int R>
__device__ __forceinline__ void load_tw1(Tw& w, GBuf tab, int t) {
    constexpr int T = Cfg<R>::T;
    // opaque offset: the forward and the inverse transform of a kernel must not share one set of loaded twiddles
    // (kept alive across the product they cost 42 VGPRs -- spills at T = 1024)
    unsigned o = (unsigned)t * 8u;
    HY_OPAQUE(o);
#if defined(OC_DBG_NO_TW) || defined(OC_DBG_NO_TW1)
    HY_UNROLL
    for (int b = 0; b < 8; ++b) { w.tB[b] = mk(1.f - 1e-6f * (float)(o + b), 1e-3f * (float)b); HY_OPAQUE(w.tB[b].x); }
    HY_UNROLL
    for (int a = 1; a < 4; ++a) { w.tA[a] = mk(1.f - 1e-6f * (float)(o + a), 1e-3f * (float)a); HY_OPAQUE(w.tA[a].x); }
    return;
#endif
    HY_UNROLL
    for (int b = 0; b < 8; ++b) w.tB[b] = gb_ld(tab, o, (unsigned)(b * T) * 8u);
    HY_UNROLL
    for (int a = 1; a < 4; ++a) w.tA[a] = gb_ld(tab, o, (unsigned)((7 + a) * T) * 8u);
}
template <int R>
__device__ __forceinline__ void load_tw2(Tw& w, GBuf tab, int tp) {
    constexpr int O = Cfg<R>::TW1;
    unsigned o = (unsigned)tp * 8u;
    HY_OPAQUE(o);
#if defined(OC_DBG_NO_TW) || defined(OC_DBG_NO_TW2)
    HY_UNROLL
    for (int b = 1; b < 8; ++b) { w.tB[b] = mk(1.f - 1e-6f * (float)(o + b), 1e-3f * (float)b); HY_OPAQUE(w.tB[b].x); }
    HY_UNROLL
    for (int a = 1; a < 4; ++a) { w.tA[a] = mk(1.f - 1e-6f * (float)(o + a), 1e-3f * (float)a); HY_OPAQUE(w.tA[a].x); }
    return;
#endif
    HY_UNROLL
    for (int b = 1; b < 8; ++b) w.tB[b] = gb_ld(tab, o, (unsigned)(O + (b - 1) * R) * 8u);
    HY_UNROLL
    for (int a = 1; a < 4; ++a) w.tA[a] = gb_ld(tab, o, (unsigned)(O + (6 + a) * R) * 8u);
}
// v[s] *= w^(s + phi) (PHI: tB[0] is the phi twist) or w^s (no PHI: s = 0 untouched); INV conjugates.
template <bool INV, bool PHI>
__device__ __forceinline__ void apply_tw(c32 (&v)[32], const Tw& w) {
    HY_UNROLL
    for (int s = 0; s < 32; ++s) {
        const int a = s >> 3, b = s & 7;
        if (!PHI && s == 0) continue;
        // (no opaque copies needed here: the table LOADS are opaque, so no product can be hoisted out of a batch loop or
        // shared between the forward and the inverse pass -- and the copies cost 90 v_mov per pass)
        const c32 ta = w.tA[a & 3], tb = w.tB[b];
        const c32 f = (a == 0) ? tb : (!PHI && b == 0) ? ta : cmul(ta, tb);
        v[s] = INV ? cmulc(v[s], f) : cmul(v[s], f);
    }
}

// ---------------------------------------------------------------------------------------------
// LDS exchanges.  `xb` = this row's buffer (lc32 or float plane), tid = thread within the row, ka = tid / R,
// tp = tid % R.  Bank analysis (32 lanes of a group must hit 32 different 8-byte / 4-byte slots mod 32):
//   exchange 1  write  q ROW1 + tid                      lanes -> consecutive slots
//               read   ka ROW1 + tp + R s                = ka T + (ka R + tp) + R s = tid + const (mod 32)
//   exchange 2  write  ka GRP2 + tp 33 + kb1             = tid 33 + kb1: stride 33
//               read   ka GRP2 + tp + 33 t'' + R i       = ka R + tp + const (mod 32) = tid + const
// ---------------------------------------------------------------------------------------------
template <int T>
__device__ __forceinline__ void row_sync() {
    if constexpr (T <= 64) HY_WAVE_SYNC();           // the row lives in one wavefront
    else __syncthreads();
}

// (every exchange moves one float plane at a time: real parts, then imaginary parts)
template <int R, bool INV, class CTX>
__device__ __forceinline__ void x1p_planes(c32 (&v)[32], HY_LDS float* xb, int tid, int ka, int tp, const CTX& c) {
    constexpr int T = Cfg<R>::T, ROW1 = Cfg<R>::ROW1;
    constexpr int MB = INV ? 17 : 3;                    // (profiling builds: stamps MB .. MB + 3 after the four barriers)
    (void)c;
    HY_LDS float* const pa = xb + tid;
    HY_LDS float* const pb = xb + ka * ROW1 + tp;
    HY_UNROLL
    for (int part = 0; part < 2; ++part) {
        if (!INV) {
            HY_UNROLL
            for (int q = 0; q < 32; ++q) pa[q * ROW1] = part ? v[q].y : v[q].x;
            row_sync<T>();
            if (part) OC_MARK(c, MB + 2); else OC_MARK(c, MB + 0);
            HY_UNROLL
            for (int s = 0; s < 32; ++s) { if (part) v[s].y = pb[R * s]; else v[s].x = pb[R * s]; }
        } else {
            HY_UNROLL
            for (int s = 0; s < 32; ++s) pb[R * s] = part ? v[s].y : v[s].x;
            row_sync<T>();
            if (part) OC_MARK(c, MB + 2); else OC_MARK(c, MB + 0);
            HY_UNROLL
            for (int q = 0; q < 32; ++q) { if (part) v[q].y = pa[q * ROW1]; else v[q].x = pa[q * ROW1]; }
        }
        row_sync<T>();
        if (part) OC_MARK(c, MB + 3); else OC_MARK(c, MB + 1);
    }
}
// OC_X1_HALVES (opt-in, per translation unit; measured and NOT adopted: profiles/r4f_x1_halves_not_kept.txt).  A thread that reads its 32 new
// values in round 0 still holds 16 unsent old ones: 96 data registers live -- 32 - 43 spilled registers under the 128 of the conv / spectrum
// kernels (+34 % time), none under the 256 of the dk / one-launch kernels, where it changes nothing (they are not LDS bound).
#ifndef OC_X1_HALVES
#define OC_X1_HALVES 0
#endif
#ifndef OC_DK_PRESYNC
#define OC_DK_PRESYNC 1
#endif
// Exchange 1 in COMPLEX halves (round 4).  The same LDS footprint as one float plane -- 16 rows of T + R complex slots -- holds half
// of the registers (q = 16 h ... 16 h + 15) as complex values: round h = every thread writes those 16 values (ds_write_b64), barrier,
// the threads whose ka lies in that half read all their 32 values (ds_read_b64), barrier.  Against the two float planes: the same 32
// 8-byte writes per thread (6 LDS cycles per wavefront instruction either way), but the reads are 32 ds_read_b64 (2 cycles each) where
// the planes need 32 ds_read2_b32 (4 cycles each) -- exchange 1 is LDS-throughput bound (profiles/r4b_conv_phase_timeline.txt: its four
// phases take exactly the cycles the LDS needs for their instructions), so a quarter of its time goes.  Still four barriers.  The halves
// are wave-uniform for T >= 128 (a wavefront's ka values lie in one half); below that the other half's lanes sit the reads out.
// Bank check (64 banks = 32 slots for b64): writes go to consecutive slots; a reading half-wave covers 32 / R rows of R slots at stride
// T + R = R mod 32 slots -- all 32 distinct.
template <int R, bool INV, class CTX>
__device__ __forceinline__ void x1p(c32 (&v)[32], HY_LDS float* xbf, int tid, int ka, int tp, const CTX& c) {
#if !OC_X1_HALVES
    x1p_planes<R, INV>(v, xbf, tid, ka, tp, c);
#else
    if constexpr (R < 2) {                              // T = 32: both halves inside every half-wave -- the planes stay (the halves spill)
        x1p_planes<R, INV>(v, xbf, tid, ka, tp, c);
        return;
    }
    constexpr int T = Cfg<R>::T, ROW1 = Cfg<R>::ROW1;
    constexpr int MB = INV ? 17 : 3;
    (void)c;
    HY_LDS lc32* const xb = HY_LDS_CAST(lc32, xbf);
    HY_LDS lc32* const pa = xb + tid;                                   // (q, tid) of the half at (q - 16 h) ROW1 + tid
    HY_LDS lc32* const pb = xb + (ka & 15) * ROW1 + tp;                 // (ka, R s + tp)
    const int myhalf = ka >> 4;
    if (!INV) {
        // round 0: everyone writes q < 16, the ka < 16 threads read their 32 values -- into `w`: their own q >= 16 are still to be written
        c32 w[32];
        HY_UNROLL
        for (int q = 0; q < 16; ++q) lds_st(pa + q * ROW1, v[q]);
        row_sync<T>();
        OC_MARK(c, MB + 0);
        if (myhalf == 0) {
            HY_UNROLL
            for (int s = 0; s < 32; ++s) w[s] = lds_ld(pb + R * s);
        }
        row_sync<T>();
        OC_MARK(c, MB + 1);
        HY_UNROLL
        for (int q = 0; q < 16; ++q) lds_st(pa + q * ROW1, v[16 + q]);
        row_sync<T>();
        OC_MARK(c, MB + 2);
        if (myhalf == 0) {
            HY_UNROLL
            for (int s = 0; s < 32; ++s) v[s] = w[s];
        } else {
            HY_UNROLL
            for (int s = 0; s < 32; ++s) v[s] = lds_ld(pb + R * s);
        }
        row_sync<T>();
        OC_MARK(c, MB + 3);
    } else {
        // round 0: the ka < 16 threads write their 32 values, everyone reads q < 16 -- into `w`: the ka >= 16 threads still hold theirs
        c32 w[16];
        if (myhalf == 0) {
            HY_UNROLL
            for (int s = 0; s < 32; ++s) lds_st(pb + R * s, v[s]);
        }
        row_sync<T>();
        OC_MARK(c, MB + 0);
        HY_UNROLL
        for (int q = 0; q < 16; ++q) w[q] = lds_ld(pa + q * ROW1);
        row_sync<T>();
        OC_MARK(c, MB + 1);
        if (myhalf != 0) {
            HY_UNROLL
            for (int s = 0; s < 32; ++s) lds_st(pb + R * s, v[s]);
        }
        row_sync<T>();
        OC_MARK(c, MB + 2);
        HY_UNROLL
        for (int q = 0; q < 16; ++q) { v[q] = w[q]; v[16 + q] = lds_ld(pa + q * ROW1); }
        row_sync<T>();
        OC_MARK(c, MB + 3);
    }
#endif
}
template <int R, bool INV>
__device__ __forceinline__ void x2p(c32 (&v)[32], HY_LDS float* xb, int ka, int tp) {
    constexpr int NB = Cfg<R>::NB, GRP2 = Cfg<R>::GRP2;
    HY_LDS float* const pa = xb + ka * GRP2 + tp * 33;
    HY_LDS float* const pb = xb + ka * GRP2 + tp;
    HY_UNROLL
    for (int part = 0; part < 2; ++part) {
        if (!INV) {
            HY_UNROLL
            for (int q = 0; q < 32; ++q) pa[q] = part ? v[q].y : v[q].x;
            HY_WAVE_SYNC();
            HY_UNROLL
            for (int i = 0; i < NB; ++i) {
                HY_UNROLL
                for (int t2 = 0; t2 < R; ++t2) {
                    const float f = pb[33 * t2 + R * i];
                    if (part) v[i * R + t2].y = f; else v[i * R + t2].x = f;
                }
            }
        } else {
            HY_UNROLL
            for (int i = 0; i < NB; ++i) {
                HY_UNROLL
                for (int t2 = 0; t2 < R; ++t2) pb[33 * t2 + R * i] = part ? v[i * R + t2].y : v[i * R + t2].x;
            }
            HY_WAVE_SYNC();
            HY_UNROLL
            for (int q = 0; q < 32; ++q) { const float f = pa[q]; if (part) v[q].y = f; else v[q].x = f; }
        }
        HY_WAVE_SYNC();
    }
}

// radix-R butterflies of pass 3 on the 32/R register groups
template <int R, bool INV>
__device__ __forceinline__ void pass3(c32 (&v)[32]) {
    if constexpr (R > 1) {
        HY_UNROLL
        for (int i = 0; i < 32 / R; ++i) {
            c32 y[R];
            HY_UNROLL
            for (int q = 0; q < R; ++q) y[q] = v[i * R + q];
            dft_reg<R, INV>(y);
            HY_UNROLL
            for (int q = 0; q < R; ++q) v[i * R + q] = y[q];
        }
    }
}

// Per-row context of a transform: where the row's exchange buffer and twiddle tables are, who the thread is.
// scheduling fences between the phases of a transform (exchange | butterflies | twiddles)
#ifdef OC_NO_PHASE_FENCES
#define OC_FENCE() do {} while (0)
#else
#define OC_FENCE() HY_SCHED_FENCE()
#endif

struct Ctx {
    HY_LDS char* xb;        // exchange buffer of this row
    GBuf tab;               // twiddle tables of this transform size (tw1 | tw2)
    int tid, ka, tp;
#if defined(OC_PROFILE) && !defined(HIPEMU)
    Prof* prof;             // time stamps of this wavefront (profiling builds)
#endif
};

// v[s] = c[tid + T s] e^(+2 pi i tid phi / M) on entry (i.e. the raw samples times the per-register twist constants);
// spectrum in the order described at the top on exit.
// OC_DBG_* (profiling builds only, scripts/build_variant.sh; results are wrong by construction): leave out the
// workgroup-wide exchange, the wave-local exchange or the butterflies to see what each costs on the hardware
#ifdef OC_DBG_NO_X1
#define OC_X1(...) do {} while (0)
#else
#define OC_X1(...) __VA_ARGS__
#endif
#ifdef OC_DBG_NO_X2
#define OC_X2(...) do {} while (0)
#else
#define OC_X2(...) __VA_ARGS__
#endif
#ifdef OC_DBG_NO_DFT
#define OC_DFT(...) do {} while (0)
#else
#define OC_DFT(...) __VA_ARGS__
#endif

// (Issuing the twiddle-table loads of a pass ahead of the butterflies that precede their use was tried -- no gain on the
// conv kernels, more spills in the 256-register dk kernels; profiles/r2_attribution.txt.)
// PRESYNC: this transform follows ANOTHER transform on the same exchange buffer without a workgroup barrier in between (dk: u, then dout,
// then the next item's u).  Exchange 1 writes anywhere in the buffer while exchange 2 of the previous transform is wave-local, so a fast
// wavefront could in principle overwrite the exchange-2 slice a slow one is still reading (it would have to be a whole transform's worth of
// instructions ahead; never observed -- 1 042 + 4 816 double-run stress cases bitwise reproducible -- but nothing in the code ruled it out
// until round 4, and a wavefront CAN be held up that long on its own: a retried page fault under XNACK with migratable memory).  One
// barrier before exchange 1 does (fft_inv has always had it).  Measured cost: dk 302 -> 311 us at 32768 x 8, 111.5 -> 116 us at 16384 x 8
// (profiles/r4h_dk_presync_ab.txt); -DOC_DK_PRESYNC=0 builds without it.
template <int R, bool PRESYNC = false>
__device__ __forceinline__ void fft_fwd(c32 (&v)[32], const Ctx& c) {
    OC_DFT((dft_reg<32, false>(v)));
    OC_FENCE();
    OC_MARK(c, 1);
    {
        Tw w;
        load_tw1<R>(w, c.tab, c.tid);
        apply_tw<false, true>(v, w);
    }
    OC_FENCE();
    OC_MARK(c, 2);
    if constexpr (PRESYNC && OC_DK_PRESYNC) row_sync<Cfg<R>::T>();
    OC_X1(x1p<R, false>(v, HY_LDS_CAST(float, c.xb), c.tid, c.ka, c.tp, c));
    OC_FENCE();
    OC_DFT((dft_reg<32, false>(v)));
    OC_FENCE();
    OC_MARK(c, 7);
    if constexpr (R > 1) {
        {
            Tw w;
            load_tw2<R>(w, c.tab, c.tp);
            apply_tw<false, false>(v, w);
        }
        OC_FENCE();
        OC_MARK(c, 8);
        OC_X2(x2p<R, false>(v, HY_LDS_CAST(float, c.xb), c.ka, c.tp));
        OC_FENCE();
        OC_MARK(c, 9);
        OC_DFT((pass3<R, false>(v)));
        OC_FENCE();
        OC_MARK(c, 10);
    }
}
// inverse (unnormalised); on exit v[s] = result[tid + T s] e^(-2 pi i (tid + T s) phi / M) e^(+2 pi i s phi / 32),
// i.e. the caller still multiplies by conj(twist_const(s)).
template <int R>
__device__ __forceinline__ void fft_inv(c32 (&v)[32], const Ctx& c) {
    if constexpr (R > 1) {
        OC_DFT((pass3<R, true>(v)));
        OC_FENCE();
        OC_MARK(c, 12);
        OC_X2(x2p<R, true>(v, HY_LDS_CAST(float, c.xb), c.ka, c.tp));
        OC_FENCE();
        OC_MARK(c, 13);
        Tw w;
        load_tw2<R>(w, c.tab, c.tp);
        apply_tw<true, false>(v, w);
    }
    OC_FENCE();
    OC_MARK(c, 14);
    OC_DFT((dft_reg<32, true>(v)));
    OC_FENCE();
    OC_MARK(c, 15);
    // exchange 1 writes anywhere in the buffer: every wavefront must be done with its exchange-2 region
    if constexpr (R > 1) row_sync<Cfg<R>::T>();
    OC_MARK(c, 16);
    OC_X1(x1p<R, true>(v, HY_LDS_CAST(float, c.xb), c.tid, c.ka, c.tp, c));
    OC_FENCE();
    {
        Tw w;
        load_tw1<R>(w, c.tab, c.tid);
        apply_tw<true, true>(v, w);
    }
    OC_FENCE();
    OC_MARK(c, 21);
    OC_DFT((dft_reg<32, true>(v)));
    OC_FENCE();
    OC_MARK(c, 22);
}

// ---------------------------------------------------------------------------------------------
// row I/O.  A workgroup's rows are RPW consecutive rows of the (B, D, L) tensor; the buffer descriptor covers them
// (RPW = 1: hardware bounds checking clips n >= L for free; RPW = 2 -- two rows per wavefront at T = 32 -- adds an
// explicit predicate).  Element n of the thread's register s is n = tid + T s.
// ---------------------------------------------------------------------------------------------
// Element access is specialised on the element SIZE only (template parameter HALF: 16-bit vs fp32); bf16 vs fp16 is a
// per-element select on the run-time dtype.  (A run-time switch over three typed store loops behind the last butterfly
// stage made hipcc spill 50 registers at T = 1024.)
__device__ __forceinline__ float half_to_f32(uint16_t h, bool bf) { return bf ? bf16_to_f32(h) : f16_to_f32(h); }
__device__ __forceinline__ uint16_t f32_to_half(float f, bool bf) { return bf ? f32_to_bf16(f) : f32_to_f16(f); }
#ifdef HIPEMU
template <bool HALF>
__device__ __forceinline__ float io_ld(GBuf b, unsigned byte_off, unsigned, bool bf) {
    if (HALF) return half_to_f32(*reinterpret_cast<const uint16_t*>(b.p + byte_off), bf);
    return *reinterpret_cast<const float*>(b.p + byte_off);
}
template <bool HALF>
__device__ __forceinline__ void io_st(GBuf b, unsigned byte_off, float v, bool bf) {
    if (HALF) *reinterpret_cast<uint16_t*>(b.p + byte_off) = f32_to_half(v, bf);
    else *reinterpret_cast<float*>(b.p + byte_off) = v;
}
#else
#ifndef OC_POL_LD
#define OC_POL_LD 0
#endif
#ifndef OC_POL_ST
#define OC_POL_ST 2          // outputs are written once: non-temporal stores, 1 - 2.5 % at every size (gpurun_out/r2z_oc)
#endif
template <bool HALF>
__device__ __forceinline__ float io_ld(GBuf b, unsigned voff, unsigned soff, bool bf) {
    if constexpr (!HALF) return u2f(__builtin_amdgcn_raw_buffer_load_b32(b.r, voff, soff, OC_POL_LD));
    else return half_to_f32(__builtin_amdgcn_raw_buffer_load_b16(b.r, voff, soff, OC_POL_LD), bf);
}
template <bool HALF>
__device__ __forceinline__ void io_st(GBuf b, unsigned voff, float v, bool bf) {
    // no scalar-offset field on stores (see gb_st)
    if constexpr (!HALF) __builtin_amdgcn_raw_buffer_store_b32(f2u(v), b.r, voff, 0, OC_POL_ST);
    else __builtin_amdgcn_raw_buffer_store_b16(f32_to_half(v, bf), b.r, voff, 0, OC_POL_ST);
}
#endif

// x[s] = sample tid + T s (+ extra) of the row at byte offset row_off of the descriptor, zero beyond L
template <int T, bool HALF, bool PRED>
__device__ __forceinline__ void load_raw(float (&x)[32], GBuf xb, bool bf, int tid, unsigned row_off, int L, int extra) {
    constexpr unsigned ES = HALF ? 2u : 4u;
#ifdef HIPEMU
    HY_UNROLL
    for (int s = 0; s < 32; ++s) {
        const int n = tid + T * s + extra;
        x[s] = n < L ? io_ld<HALF>(xb, row_off + (unsigned)n * ES, 0u, bf) : 0.f;
    }
#else
#ifdef OC_DBG_NO_IO
    HY_UNROLL
    for (int s = 0; s < 32; ++s) { x[s] = (float)(tid * 3 + s); HY_OPAQUE(x[s]); }
#else
    HY_UNROLL
    for (int s = 0; s < 32; ++s) x[s] = io_ld<HALF>(xb, row_off + (unsigned)(tid + extra) * ES, (unsigned)(T * s) * ES, bf);
#endif
    if constexpr (PRED) {            // several rows under one descriptor: its bounds check cannot clip n >= L
        HY_UNROLL
        for (int s = 0; s < 32; ++s) x[s] = (tid + T * s + extra < L) ? x[s] : 0.f;
    }
#endif
}
// the row's samples times the per-register input twist: v[s] = x[tid + T s] twist(s)
template <int R, int PHI8, bool HALF, bool PRED>
__device__ __forceinline__ void load_row(c32 (&v)[32], GBuf xb, bool bf, int tid, unsigned row_off, int L) {
    float x[32];
    load_raw<Cfg<R>::T, HALF, PRED>(x, xb, bf, tid, row_off, L, 0);
    HY_UNROLL
    for (int s = 0; s < 32; ++s) {
        v[s] = rmul(x[s], twist_const<PHI8>(s));
    }
}
// y[s] -> sample tid + T s of the row.  One running address register (32 precomputed voffsets cost 32 VGPRs); with one
// row per descriptor the hardware bounds check drops n >= L.
template <int T, bool HALF, bool PRED>
__device__ __forceinline__ void store_row(GBuf ob, bool bf, int tid, unsigned row_off, int L, const float (&y)[32]) {
    constexpr unsigned ES = HALF ? 2u : 4u;
#ifdef HIPEMU
    HY_UNROLL
    for (int s = 0; s < 32; ++s) {
        const int n = tid + T * s;
        if (n < L) io_st<HALF>(ob, row_off + (unsigned)n * ES, y[s], bf);
    }
#else
    unsigned vo = row_off + (unsigned)tid * ES;
#ifdef OC_DBG_NO_IO
    float acc = 0.f;
    HY_UNROLL
    for (int s = 0; s < 32; ++s) acc += y[s];
    io_st<HALF>(ob, vo, acc, bf);
#else
    HY_UNROLL
    for (int s = 0; s < 32; ++s) {
        if (!PRED || tid + T * s < L) io_st<HALF>(ob, vo, y[s], bf);
        vo += (unsigned)T * ES;
        HY_OPAQUE(vo);
    }
#endif
#endif
}

// ---------------------------------------------------------------------------------------------
// kernels
// ---------------------------------------------------------------------------------------------
struct SpecArgs {          // row spectrum: H[d] = (FFT(c_k) + bias) / M -- of the filter rows (fp32), or, for dk at B = 1, of the u rows
    const void* k;         // (D, L) fp32 / 16-bit, row pitch ld
    const float* bias;     // (D,) or null
    c32* H;                // [D][M], register order
    const c32* tab;
    int D, L, dtype;
    int ld;                // elements between the starts of consecutive rows of k (>= L)
};
struct ConvArgs {          // out = IFFT(FFT(x) .* H) (conj_sign = +1) or .* conj(H) (conj_sign = -1)
    const void* x;         // (B, D, L)
    void* out;             // (B, D, L), the type of x -- or fp32 (conv_kernel<R, HALF, true>: dk at B = 1)
    const c32* H;          // [D][M]
    const c32* tab;
    int B, D, L, dtype;
    float conj_sign;
    int ldx, ldo;          // row pitch (elements, >= L) of x and of out: row (b, d) starts at element (b D + d) ld
};
struct DkArgs {            // dk[d] = sum_b corr(dout[b, d], u[b, d]);  dbias[d] = dk[d][0]
    const void* dout;
    const void* u;
    float* dk;             // (D, L) fp32
    float* dbias;          // (D,) or null
    const c32* tab;        // tables of the (sub-)transform; NP = 2: the two parity sets of size 16384, one after the other
    float* part;           // S > 1: [S][D][L] fp32, the slices' partial dk rows (summed in slice order by dk_sum_kernel)
    int B, D, L, dtype;
    int S, nb;             // batch slices per channel (grid.y) and batch items per slice: slice s owns b in [s nb, min(B, (s + 1) nb))
    int ldx, ldk;          // row pitch (elements, >= L) of dout / u and of dk (`part` rows are packed: pitch L)
    // dk_kernel<..., DU = true> (round 6): du from the SAME transform of dout -- du[b, d] = corr(dout[b, d], H[d]) as conv_kernel computes it
    void* du;              // (B, D, L), the type of dout, row pitch ldx
    const c32* H;          // [D][M] filter spectrum (spec_kernel's)
};

// Wavefronts per SIMD the conv / spectrum kernels are compiled for (-> at most 128 VGPRs): a wavefront issues one VALU
// instruction per ~4.4 cycles, so a SIMD needs 3-4 of them to stay busy (profiles/r2d_pmc_sq_onchip.csv)
#ifndef OC_MINW
#define OC_MINW 4
#endif

template <int R> struct WgCfg {
    static constexpr int T = Cfg<R>::T;
    static constexpr int WGT = T < 64 ? 64 : T;          // workgroup threads of the conv / spectrum kernels
    static constexpr int RPW = WGT / T;                  // rows per workgroup
    static constexpr size_t LDS = Cfg<R>::XBYTES * RPW;
};

template <int R>
__device__ __forceinline__ Ctx make_ctx(HY_LDS char* smem, int rg, int tid, const c32* tab) {
    Ctx c;
    c.xb = smem + (size_t)rg * Cfg<R>::XBYTES;
    c.tab = make_gbuf(tab, (unsigned)Cfg<R>::TWN * 8u);
    c.tid = tid;
    c.ka = tid / R;
    c.tp = tid % R;
#if defined(OC_PROFILE) && !defined(HIPEMU)
    c.prof = nullptr;
#endif
    return c;
}

template <int R, bool HALF = false>
__global__ void __launch_bounds__(WgCfg<R>::WGT, OC_MINW) spec_kernel(SpecArgs a) {
    typedef Cfg<R> C;
    constexpr int T = C::T, RPW = WgCfg<R>::RPW;
    constexpr unsigned ES = HALF ? 2u : 4u;
    HY_SMEM(smem);
    const bool bf = a.dtype == DT_BF16;
    const int rg = RPW == 1 ? 0 : (int)threadIdx.x / T, tid = RPW == 1 ? (int)threadIdx.x : (int)threadIdx.x % T;
    const int d0 = blockIdx.x * RPW;
    const int d_raw = d0 + rg;
    const bool valid = d_raw < a.D;
    const int d = valid ? d_raw : a.D - 1;
    const Ctx c = make_ctx<R>(HY_LDS_CAST(char, smem), rg, tid, a.tab);
    const int nrows = (a.D - d0) < RPW ? (a.D - d0) : RPW;
    const GBuf kb = make_gbuf(reinterpret_cast<const char*>(a.k) + (size_t)d0 * a.ld * ES, ((unsigned)(nrows - 1) * (unsigned)a.ld + (unsigned)a.L) * ES);
    c32 v[32];
    load_row<R, 2, HALF, (RPW > 1)>(v, kb, bf, tid, (unsigned)(d - d0) * (unsigned)a.ld * ES, a.L);
    fft_fwd<R>(v, c);
    const float bias = (a.bias != nullptr) ? a.bias[d] : 0.f;
    const float sc = 1.0f / (float)C::M;
    if (valid) {
        c32* Hd = a.H + (size_t)d * C::M + tid;
        HY_UNROLL
        for (int q = 0; q < 32; ++q) Hd[q * T] = mk((v[q].x + bias) * sc, v[q].y * sc);
    }
}

template <int R, bool HALF, bool OUTF32 = false>
__global__ void __launch_bounds__(WgCfg<R>::WGT, OC_MINW) conv_kernel(ConvArgs a) {
    typedef Cfg<R> C;
    constexpr int T = C::T, RPW = WgCfg<R>::RPW;
    constexpr unsigned ES = HALF ? 2u : 4u;
    constexpr bool OHALF = HALF && !OUTF32;                  // element size of the output rows
    constexpr unsigned EO = OHALF ? 2u : 4u;
    HY_SMEM(smem);
    const bool bf = a.dtype == DT_BF16;
    const int rg = RPW == 1 ? 0 : (int)threadIdx.x / T, tid = RPW == 1 ? (int)threadIdx.x : (int)threadIdx.x % T;
    const int rows = a.B * a.D;
    // Row of this workgroup.  One row per workgroup: workgroups are dealt to the 8 XCDs round-robin, so the B rows of a
    // channel (which read the same 8 M bytes of H) are given consecutive slots of ONE XCD's sequence -- its L2 then
    // serves B - 1 of the B reads of H.
    int r0;
    if (RPW == 1 && (a.D & 7) == 0) {
        const int w = blockIdx.x, xcd = w & 7, seq = w >> 3;
        const int cs = seq / a.B, b = seq - cs * a.B;
        r0 = HY_SGPR(b * a.D + cs * 8 + xcd);
    } else {
        r0 = blockIdx.x * RPW;
    }
    const int r_raw = r0 + rg;
    const bool valid = r_raw < rows;
    const int r = valid ? r_raw : rows - 1;
    const int d = RPW == 1 ? HY_SGPR(r % a.D) : r % a.D;
    const Ctx c = make_ctx<R>(HY_LDS_CAST(char, smem), rg, tid, a.tab);
    const int nrows = (rows - r0) < RPW ? (rows - r0) : RPW;
    const GBuf xb = make_gbuf(reinterpret_cast<const char*>(a.x) + (size_t)r0 * a.ldx * ES, ((unsigned)(nrows - 1) * (unsigned)a.ldx + (unsigned)a.L) * ES);
    const GBuf ob = make_gbuf(reinterpret_cast<char*>(a.out) + (size_t)r0 * a.ldo * EO, ((unsigned)(nrows - 1) * (unsigned)a.ldo + (unsigned)a.L) * EO);
    const unsigned row_off = (unsigned)(r - r0) * (unsigned)a.ldx * ES, orow_off = (unsigned)(r - r0) * (unsigned)a.ldo * EO;
    const GBuf hb = make_gbuf(a.H, (unsigned)a.D * (unsigned)C::M * 8u);
#if defined(OC_PROFILE) && !defined(HIPEMU)
    Prof prof;
    HY_UNROLL
    for (int i = 0; i < 32; ++i) prof.t[i] = 0;
    const_cast<Ctx&>(c).prof = &prof;
    unsigned long long rt0, rt1;
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(rt0)::"memory");       // 100 MHz reference: calibrates s_memtime
    OC_MARK(c, 0);
#endif
    c32 v[32];
    load_row<R, 2, HALF, (RPW > 1)>(v, xb, bf, tid, row_off, a.L);
    fft_fwd<R>(v, c);
    {
        const unsigned ho = ((unsigned)d * (unsigned)C::M + (unsigned)tid) * 8u;
        HY_UNROLL
        for (int q0 = 0; q0 < 32; q0 += 8) {          // 8 at a time: 32 filter values at once do not fit 128 VGPRs at T = 1024
            c32 h[8];
#ifdef OC_DBG_NO_H
            HY_UNROLL
            for (int q = 0; q < 8; ++q) { h[q] = mk((float)(tid + q), 0.5f); HY_OPAQUE(h[q].x); }
#else
            HY_UNROLL
            for (int q = 0; q < 8; ++q) h[q] = gb_ld(hb, ho, (unsigned)((q0 + q) * T) * 8u);
#endif
            HY_UNROLL
            for (int q = 0; q < 8; ++q) v[q0 + q] = cmul(v[q0 + q], mk(h[q].x, a.conj_sign * h[q].y));
            HY_SCHED_FENCE();
        }
    }
    OC_MARK(c, 11);
    fft_inv<R>(v, c);
    float y[32];
    HY_UNROLL
    for (int s = 0; s < 32; ++s) {
        const c32 w = twist_const<2>(s);
        y[s] = v[s].x * w.x + v[s].y * w.y;             // Re(v conj(twist))
    }
    // (one row per workgroup: the grid is exactly B D blocks, every block is valid; a conditional epilogue costs hipcc 36
    // spilled registers at T = 1024)
    if (RPW == 1 || valid) store_row<T, OHALF, (RPW > 1)>(ob, bf, tid, orow_off, a.L, y);
#if defined(OC_PROFILE) && !defined(HIPEMU)
    OC_MARK(c, 23);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    OC_MARK(c, 24);
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(rt1)::"memory");
    if (oc_prof_buf != nullptr && (threadIdx.x & 63) == 0) {
        unsigned long long* o = oc_prof_buf + ((size_t)blockIdx.x * (WgCfg<R>::WGT / 64) + (threadIdx.x >> 6)) * 32;
        HY_UNROLL
        for (int i = 0; i < 25; ++i) o[i] = prof.t[i];
        // which CU / XCD ran this workgroup (HW_ID register: CU id bits 8-11, SE 13-15 ...; XCC_ID separately)
        unsigned hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        o[25] = ((unsigned long long)xcc << 32) | hw;
        o[26] = rt0;
        o[27] = rt1;
    }
#endif
}

// dbias[d] = dk[d][0] (after a dk that came out of conv_kernel)
template <int UNUSED = 0>
__global__ void __launch_bounds__(256) dk_bias_kernel(const float* dk, float* dbias, int D, int ld) {
    const int d = (int)(blockIdx.x * 256 + threadIdx.x);
    if (d < D) dbias[d] = dk[(size_t)d * ld];
}

// dk.  A workgroup owns a channel; its BP row groups (T threads each) take the batch items b = g, g + BP, ... and keep
// their partial sum of G .* conj(U) in 32 registers; the groups' sums are added through LDS in group order and ONE
// inverse transform per channel produces dk (bitwise reproducible: fixed order, no atomics).
// NP = 2 (M = 32768: two spectra + the accumulator would need 768 KB of registers): the transform is split by one
// radix-2 decimation-in-frequency step into the even and the odd bins, two 16384-point problems
//     y_e[n] = x[n] + (-1)^e e^(-i pi / 4) x[n + M/2],  twist phi_e = (1 + 4 e) / 8,
// run as two launches (E0 = 0, 1: both parities unrolled into one kernel made hipcc spill 126 registers); with h_e the
// untwisted inverse of bin set e,
//     dk[n] = Re(h_0 + h_1) / M,     dk[n + M/2] = Re(e^(i pi / 4) (h_0 - h_1)) / M,
// the e = 0 halves are parked in the dk row itself and the e = 1 launch adds to them.
// Table layout for NP = 2 (R = 16): [tw1(phi = 1/8) | tw2] [tw1(phi = 5/8) | tw2].
template <int R, int NP> struct DkCfg {
    static constexpr int T = Cfg<R>::T;
    static constexpr int BP = (512 / T > 16) ? 16 : 512 / T;       // row groups per workgroup (T BP <= 512 threads: ~192 VGPRs each)
    static constexpr int WGT = T * BP;
    static_assert(WGT == 512 && (NP == 1 || BP == 1), "dk workgroups are 512 threads");
    static constexpr size_t LDS_X = Cfg<R>::XBYTES * BP;
    static constexpr size_t LDS_RED = BP > 1 ? (size_t)BP * Cfg<R>::M * 8 : 0;
    // NP = 2 (two spectra + the accumulator = 192 of the 256 registers): the upper half of the accumulator lives in LDS (64 KB next to
    // the 68 KB exchange buffer; 16 b64 reads + writes per batch item against the 512 LDS accesses of its two transforms)
    static constexpr int PARKQ = NP == 2 ? 16 : 0;
    static constexpr size_t PARK = (size_t)PARKQ * WGT * 8;
    static constexpr size_t LDS = (LDS_X > LDS_RED ? LDS_X : LDS_RED) + PARK;
};

// spectrum input of sub-problem e (NP = 2: PHI8 = 1 + 4 e, sigma = +-1) or of the whole row (NP = 1)
template <int R, int NP, bool HALF, int PHI8>
__device__ __forceinline__ void dk_load(c32 (&v)[32], GBuf xb, bool bf, int tid, unsigned row_off, int L, float sigma) {
    if constexpr (NP == 1) {
        // T = 32: two row groups share a wavefront, hence one descriptor (the whole tensor) and a per-lane row offset
        load_row<R, PHI8, HALF, (Cfg<R>::T < 64)>(v, xb, bf, tid, row_off, L);
    } else {
        constexpr int T = Cfg<R>::T, MS = Cfg<R>::M;
        const float r = 0.70710678118654752440f * sigma;
        // all 64 loads of the thread in one batch (one memory round trip; 8-pair batches measured 4x slower on MI355X:
        // with 2 wavefronts per SIMD nothing hides the latency between batches)
        float xa[32], xc[32];
        load_raw<T, HALF, false>(xa, xb, bf, tid, 0u, L, 0);
        load_raw<T, HALF, false>(xc, xb, bf, tid, 0u, L, MS);
        HY_UNROLL
        for (int s = 0; s < 32; ++s) {
            const c32 y = mk(xa[s] + r * xc[s], -r * xc[s]);           // x[n] + sigma e^(-i pi/4) x[n + M/2]
            v[s] = cmul(y, twist_const<PHI8>(s));
        }
    }
}

template <int R, int NP, bool HALF, int E0, bool DU = false>
__global__ void __launch_bounds__((DkCfg<R, NP>::WGT)) dk_kernel(DkArgs a) {
    static_assert(!DU || NP == 1, "du rides on dk only where the row is one transform (M <= 16384)");
    typedef Cfg<R> C;
    typedef DkCfg<R, NP> K;
    constexpr int T = C::T, BP = K::BP;
    constexpr unsigned ES = HALF ? 2u : 4u;
    HY_SMEM(smem);
    const bool bf = a.dtype == DT_BF16;
    const int rg = BP == 1 ? 0 : (int)threadIdx.x / T, tid = BP == 1 ? (int)threadIdx.x : (int)threadIdx.x % T;
    const int d = blockIdx.x;
    // With fewer channels than CUs a channel's batch is cut into S slices (blockIdx.y), each a workgroup of its own that leaves
    // its partial dk row in `part`; the transform is linear, so the rows are simply added afterwards (dk_sum_kernel, slice order).
    const int sl = blockIdx.y;
    const int b_lo = sl * a.nb, b_hi = (b_lo + a.nb < a.B) ? b_lo + a.nb : a.B;
    const bool sliced = a.S > 1;
    Ctx c = make_ctx<R>(HY_LDS_CAST(char, smem), rg, tid, a.tab);
    const unsigned rowbytes = (unsigned)a.L * ES;
    const float sc = 1.0f / (float)(C::M * NP);
    float* dkrow = sliced ? a.part + ((size_t)sl * a.D + d) * a.L : a.dk + (size_t)d * a.ldk;
    float* dbias = sliced ? nullptr : a.dbias;
    {
        constexpr int e = E0;
        constexpr int PHI_A = NP == 2 ? 1 : 2, PHI_B = 5;       // e = 0 / e = 1
        const float sigma = e ? -1.f : 1.f;
        if constexpr (NP == 2) c.tab = make_gbuf(a.tab + e * C::TWN, (unsigned)C::TWN * 8u);
        c32 acc[32];
        HY_UNROLL
        for (int q = 0; q < 32; ++q) acc[q] = mk(0.f, 0.f);
        constexpr int NREG = 32 - K::PARKQ;                  // accumulator entries kept in registers; the rest at park[(q - NREG) WGT]
        HY_LDS lc32* const park = HY_LDS_CAST(lc32, HY_LDS_CAST(char, smem) + (K::LDS - K::PARK)) + threadIdx.x;
        HY_UNROLL
        for (int q = NREG; q < 32; ++q) lds_st(park + (q - NREG) * K::WGT, mk(0.f, 0.f));
        for (int b0 = b_lo; b0 < b_hi; b0 += BP) {    // uniform trip count: the transforms contain workgroup barriers
            const bool live = b0 + rg < b_hi;
            const int b = live ? b0 + rg : b_hi - 1;
            const size_t row = ((size_t)b * a.D + d) * a.ldx * ES;
            // one descriptor per row (hardware bounds check clips n >= L) -- except at T = 32, where the two row groups of
            // a wavefront need a common one: the whole tensor (< 4 GB, checked by the host) + a per-lane row offset
            constexpr bool WHOLE = T < 64;
            const unsigned whole = (unsigned)((((size_t)a.B * a.D - 1) * a.ldx + a.L) * ES);
            const GBuf gb = WHOLE ? make_gbuf(a.dout, whole) : make_gbuf(reinterpret_cast<const char*>(a.dout) + row, rowbytes);
            const GBuf ub = WHOLE ? make_gbuf(a.u, whole) : make_gbuf(reinterpret_cast<const char*>(a.u) + row, rowbytes);
            const unsigned row_off = WHOLE ? (unsigned)row : 0u;
            c32 u[32], v[32];
            if (e == 0) dk_load<R, NP, HALF, PHI_A>(u, ub, bf, tid, row_off, a.L, sigma);
            else dk_load<R, NP, HALF, PHI_B>(u, ub, bf, tid, row_off, a.L, sigma);
            fft_fwd<R, true>(u, c);
            HY_SCHED_FENCE();
            if (e == 0) dk_load<R, NP, HALF, PHI_A>(v, gb, bf, tid, row_off, a.L, sigma);
            else dk_load<R, NP, HALF, PHI_B>(v, gb, bf, tid, row_off, a.L, sigma);
            fft_fwd<R, true>(v, c);
            HY_SCHED_FENCE();
            const float lv = live ? 1.f : 0.f;
            HY_UNROLL
            for (int q = 0; q < 32; ++q) {
                const c32 p = cmulc(v[q], u[q]);                                             // G conj(U)
                if (q < NREG) acc[q] = mk(acc[q].x + lv * p.x, acc[q].y + lv * p.y);
                else {                                                                       // (own slots: no synchronisation)
                    const c32 o = lds_ld(park + (q - NREG) * K::WGT);
                    lds_st(park + (q - NREG) * K::WGT, mk(o.x + lv * p.x, o.y + lv * p.y));
                }
            }
            if constexpr (DU) {
                // du of this batch item from the transform of dout that is in registers anyway: G conj(H) -> inverse -> the row (conv_kernel's
                // arithmetic; the separate launch transforms dout a second time: 6.1 -> 5 transforms per row of a forward + backward step)
                HY_SCHED_FENCE();
                const GBuf hb = make_gbuf(a.H, (unsigned)a.D * (unsigned)C::M * 8u);
                const unsigned ho = ((unsigned)d * (unsigned)C::M + (unsigned)tid) * 8u;
                HY_UNROLL
                for (int q0 = 0; q0 < 32; q0 += 8) {
                    c32 h[8];
                    HY_UNROLL
                    for (int q = 0; q < 8; ++q) h[q] = gb_ld(hb, ho, (unsigned)((q0 + q) * T) * 8u);
                    HY_UNROLL
                    for (int q = 0; q < 8; ++q) v[q0 + q] = cmul(v[q0 + q], mk(h[q].x, -h[q].y));
                    HY_SCHED_FENCE();
                }
                fft_inv<R>(v, c);
                float y[32];
                HY_UNROLL
                for (int s = 0; s < 32; ++s) {
                    const c32 w = twist_const<2>(s);
                    y[s] = v[s].x * w.x + v[s].y * w.y;         // Re(v conj(twist))
                }
                // (no branch around the stores: a dead row group -- the batch ran out -- stores nothing because its descriptor is empty / its length 0)
                const GBuf ob = WHOLE ? make_gbuf(a.du, whole) : make_gbuf(reinterpret_cast<char*>(a.du) + row, live ? rowbytes : 0u);
                store_row<T, HALF, WHOLE>(ob, bf, tid, row_off, live ? a.L : 0, y);
                HY_SCHED_FENCE();
            }
        }
        HY_UNROLL
        for (int q = NREG; q < 32; ++q) acc[q] = lds_ld(park + (q - NREG) * K::WGT);
        if constexpr (BP > 1) {
            // sum of the groups' partial spectra, in group order, through LDS (the exchange buffers are idle now)
            __syncthreads();
            HY_LDS lc32* red = HY_LDS_CAST(lc32, smem);
            if (rg > 0) {
                HY_UNROLL
                for (int q = 0; q < 32; ++q) lds_st(red + rg * C::M + q * T + tid, acc[q]);
            }
            __syncthreads();
            if (rg == 0) {
                const int ng = (b_hi - b_lo) < BP ? (b_hi - b_lo) : BP;
                for (int o = 1; o < ng; ++o) {
                    HY_UNROLL
                    for (int q = 0; q < 32; ++q) acc[q] = cadd(acc[q], lds_ld(red + o * C::M + q * T + tid));
                }
            }
            __syncthreads();
        }
        // Only group 0 holds the sum, but every group runs the inverse: its barriers are workgroup-wide (the other
        // groups transform their stale partial sums inside their own exchange buffers and store nothing).
        fft_inv<R>(acc, c);
        if (rg == 0) {
            if constexpr (NP == 1) {
                HY_UNROLL
                for (int s = 0; s < 32; ++s) {
                    const c32 w = twist_const<2>(s);
                    const int n = tid + T * s;
                    const float val = (acc[s].x * w.x + acc[s].y * w.y) * sc;
                    if (n < a.L) dkrow[n] = val;
                    if (n == 0 && dbias != nullptr) dbias[d] = val;
                }
            } else {
                const float rr = 0.70710678118654752440f;
                HY_UNROLL
                for (int s = 0; s < 32; ++s) {
                    const c32 h = cmulc(acc[s], e ? twist_const<PHI_B>(s) : twist_const<PHI_A>(s));          // untwisted h_e[n]
                    const int n = tid + T * s;
                    const float lo = h.x * sc, hi = (h.x - h.y) * rr * sc;   // Re(h), Re(e^(i pi/4) h)
                    if (e == 0) {
                        if (n < a.L) dkrow[n] = lo;
                        if (n + C::M < a.L) dkrow[n + C::M] = hi;
                    } else {
                        if (n < a.L) {
                            const float val = dkrow[n] + lo;
                            dkrow[n] = val;
                            if (n == 0 && dbias != nullptr) dbias[d] = val;
                        }
                        if (n + C::M < a.L) dkrow[n + C::M] -= hi;
                    }
                }
            }
        }
    }
}

// dk[d][n] = sum over the S batch slices of part[s][d][n], in slice order (deterministic); dbias[d] = dk[d][0]
template <int UNUSED = 0>     // (a template only so that the two translation units including this header do not both define it)
__global__ void __launch_bounds__(256) dk_sum_kernel(DkArgs a) {
    const int d = blockIdx.y;
    const int n = (int)(blockIdx.x * 256 + threadIdx.x);
    if (n >= a.L) return;
    float acc = a.part[(size_t)d * a.L + n];
    for (int s = 1; s < a.S; ++s) acc += a.part[((size_t)s * a.D + d) * a.L + n];
    a.dk[(size_t)d * a.ldk + n] = acc;
    if (n == 0 && a.dbias != nullptr) a.dbias[d] = acc;
}


// ---------------------------------------------------------------------------------------------
// Short rows (M <= 2048) with a small batch: ONE launch per direction.  A hyenadna-tiny-1k step (L = 1024, B = 8, D = 128)
// is 1.5 us of memory traffic; as spec + conv (forward) and conv + dk (+ dk_sum) (backward) it is four to five launches
// of ~5-10 us each.  Here a workgroup owns a channel and its G = 256 / T row groups the batch items b = g, g + G, ...; what bounds a
// launch is then the DEPENDENT CHAIN of 1024-point transforms one wavefront walks (a row lives in one wavefront: no workgroup barrier
// inside a transform, but also nothing to overlap it with), so the independent transforms of a row are given to different wavefronts:
//   small_fwd_kernel : 4 row wavefronts + 1 filter wavefront.  The filter wavefront transforms the filter row (H = (FFT(c_k) + bias) / M)
//                      while the row wavefronts transform their rows; H crosses through LDS at ONE workgroup barrier (and goes to the
//                      saved-spectrum buffer for the backward if asked to).  Chain: 2 transforms (was 3: every row group transformed
//                      the filter itself).
//   small_bwd_kernel : 4 "g" wavefronts (G = FFT(dout)) + 4 "u" wavefronts (U = FFT(u)), side by side; U crosses through LDS, the g
//                      wavefront of a row forms G conj(U) (summed over its batch items in registers), leaves the sum in LDS, and runs
//                      G conj(H) -> inverse -> du WHILE the first u wavefront adds the groups' sums in group order (bitwise
//                      reproducible) and runs the ONE inverse per channel that gives dk, dbias = dk[0].  Chain: 2 transforms + the
//                      sum (was 4 + the sum).  One transform of dout serves both gradients -- the shape of the reference's fused
//                      backward (csrc/fftconv/fftconv_cuda.cu:945-1266).
// At T = 32 two row groups share a wavefront (one whole-tensor descriptor + per-lane row offsets + an explicit n < L predicate, as in
// dk_kernel).  Every thread of a workgroup passes the same number of workgroup barriers (1 forward, 2 per batch step backward).
// ---------------------------------------------------------------------------------------------
struct SmallFwdArgs {
    const void* x;         // (B, D, L)
    void* out;             // (B, D, L)
    const float* k;        // (D, L) fp32
    const float* bias;     // (D,) or null
    c32* Hout;             // [D][M] or null: where the backward finds the filter spectrum
    const c32* tab;
    int B, D, L, dtype;
    int ldx, ldk;          // row pitch (elements, >= L) of x / out and of k
};
struct SmallBwdArgs {
    const void* dout;      // (B, D, L)
    const void* u;         // (B, D, L); read only if dk != null
    void* du;              // (B, D, L) or null
    float* dk;             // (D, L) fp32 or null
    float* dbias;          // (D,) or null
    const c32* H;          // [D][M] from the forward
    const c32* tab;
    int B, D, L, dtype;
    int ldx, ldk;          // row pitch (elements, >= L) of dout / u / du and of dk
};

template <int R> struct SmallCfg {
    static_assert(R <= 2, "row groups must not contain workgroup barriers");
    static constexpr int T = Cfg<R>::T;
    static constexpr int G = 256 / T;                                   // row groups per workgroup (4 wavefronts)
    static constexpr int FG = 64 / T;                                   // groups of the filter wavefront (T = 32: two, the second one redundant)
    static constexpr int WGT_FWD = 256 + 64;
    static constexpr int WGT_BWD = 512;                                 // G "g" groups, then G "u" groups
    static constexpr size_t LDS_XF = Cfg<R>::XBYTES * (G + FG);
    static constexpr size_t LDS_FWD = LDS_XF + (size_t)Cfg<R>::M * 8;   // + H
    static constexpr size_t LDS_XB = Cfg<R>::XBYTES * 2 * G;
    static constexpr size_t LDS_BWD = LDS_XB + (size_t)G * Cfg<R>::M * 8;   // + per group: U, later the group's sum of G conj(U)
};

template <int R, bool HALF>
__global__ void __launch_bounds__(SmallCfg<R>::WGT_FWD) small_fwd_kernel(SmallFwdArgs a) {
    typedef Cfg<R> C;
    typedef SmallCfg<R> S;
    constexpr int T = C::T, G = S::G;
    constexpr unsigned ES = HALF ? 2u : 4u;
    HY_SMEM(smem);
    const bool bf = a.dtype == DT_BF16;
    const int grp = (int)threadIdx.x / T, tid = (int)threadIdx.x % T;
    const int d = blockIdx.x;
    const Ctx c = make_ctx<R>(HY_LDS_CAST(char, smem), grp, tid, a.tab);
    HY_LDS lc32* const Hs = HY_LDS_CAST(lc32, HY_LDS_CAST(char, smem) + S::LDS_XF) + tid;
    if (grp >= G) {
        // the filter wavefront: H = (FFT(c_k) + bias) / M -> LDS (and the saved-spectrum buffer)
        c32 h[32];
        const GBuf kb = make_gbuf(a.k, ((unsigned)(a.D - 1) * (unsigned)a.ldk + (unsigned)a.L) * 4u);
        load_row<R, 2, false, true>(h, kb, false, tid, (unsigned)d * (unsigned)a.ldk * 4u, a.L);
        fft_fwd<R>(h, c);
        const float bias = (a.bias != nullptr) ? a.bias[d] : 0.f;
        const float sc = 1.0f / (float)C::M;
        HY_UNROLL
        for (int q = 0; q < 32; ++q) h[q] = mk((h[q].x + bias) * sc, h[q].y * sc);
        if (grp == G) {
            HY_UNROLL
            for (int q = 0; q < 32; ++q) lds_st(Hs + q * T, h[q]);
            if (a.Hout != nullptr) {
                c32* Hd = a.Hout + (size_t)d * C::M + tid;
                HY_UNROLL
                for (int q = 0; q < 32; ++q) Hd[q * T] = h[q];
            }
        }
        __syncthreads();
        return;
    }
    const unsigned total = a.B > 0 ? (((unsigned)a.B * (unsigned)a.D - 1u) * (unsigned)a.ldx + (unsigned)a.L) * ES : 0u;
    const GBuf xb = make_gbuf(a.x, total);
    const GBuf ob = make_gbuf(a.out, total);
    c32 h[32];
    bool have_h = false;
    for (int b0 = 0; b0 < a.B; b0 += G) {
        const bool live = b0 + grp < a.B;
        const int b = live ? b0 + grp : a.B - 1;
        const unsigned row_off = ((unsigned)b * (unsigned)a.D + (unsigned)d) * (unsigned)a.ldx * ES;
        c32 v[32];
        load_row<R, 2, HALF, true>(v, xb, bf, tid, row_off, a.L);
        fft_fwd<R>(v, c);
        if (!have_h) {
            __syncthreads();                              // H is in LDS
            HY_UNROLL
            for (int q = 0; q < 32; ++q) h[q] = lds_ld(Hs + q * T);
            have_h = true;
        }
        HY_UNROLL
        for (int q = 0; q < 32; ++q) v[q] = cmul(v[q], h[q]);
        fft_inv<R>(v, c);
        float y[32];
        HY_UNROLL
        for (int s = 0; s < 32; ++s) {
            const c32 w = twist_const<2>(s);
            y[s] = v[s].x * w.x + v[s].y * w.y;             // Re(v conj(twist))
        }
        // (a dead row group of the last step stores nothing: L = 0 fails every n < L predicate)
        store_row<T, HALF, true>(ob, bf, tid, row_off, live ? a.L : 0, y);
    }
    if (!have_h) __syncthreads();                         // empty batch (the backward's "transform the filter only" call): still one barrier
}

template <int R, bool HALF>
__global__ void __launch_bounds__(SmallCfg<R>::WGT_BWD) small_bwd_kernel(SmallBwdArgs a) {
    typedef Cfg<R> C;
    typedef SmallCfg<R> S;
    constexpr int T = C::T, G = S::G;
    constexpr unsigned ES = HALF ? 2u : 4u;
    HY_SMEM(smem);
    const bool bf = a.dtype == DT_BF16;
    const int grp = (int)threadIdx.x / T, tid = (int)threadIdx.x % T;
    const bool is_u = grp >= G;
    const int rg = is_u ? grp - G : grp;
    const int d = blockIdx.x;
    const Ctx c = make_ctx<R>(HY_LDS_CAST(char, smem), grp, tid, a.tab);
    const unsigned total = a.B > 0 ? (((unsigned)a.B * (unsigned)a.D - 1u) * (unsigned)a.ldx + (unsigned)a.L) * ES : 0u;
    const GBuf gb = make_gbuf(a.dout, total);
    const GBuf ub = make_gbuf(a.u != nullptr ? a.u : a.dout, total);
    const GBuf ob = make_gbuf(a.du != nullptr ? a.du : const_cast<void*>(a.dout), total);
    const GBuf hb = make_gbuf(a.H, (unsigned)a.D * (unsigned)C::M * 8u);
    const unsigned ho = ((unsigned)d * (unsigned)C::M + (unsigned)tid) * 8u;
    const bool want_dk = a.dk != nullptr, want_du = a.du != nullptr;
    // slot of row group rg in LDS, element q T + tid: first U of the group's batch item, after the last step the group's sum of
    // G conj(U) (each thread touches only its own 32 entries of its group's slot)
    HY_LDS lc32* const slots = HY_LDS_CAST(lc32, HY_LDS_CAST(char, smem) + S::LDS_XB);
    HY_LDS lc32* const mine = slots + rg * C::M + tid;
    c32 acc[32];
    HY_UNROLL
    for (int q = 0; q < 32; ++q) acc[q] = mk(0.f, 0.f);
    for (int b0 = 0; b0 < a.B; b0 += G) {                 // uniform trip count: two workgroup barriers per step for every thread
        const bool last = b0 + G >= a.B;
        const bool live = b0 + rg < a.B;
        const int b = live ? b0 + rg : a.B - 1;
        const unsigned row_off = ((unsigned)b * (unsigned)a.D + (unsigned)d) * (unsigned)a.ldx * ES;
        if (is_u) {
            if (want_dk) {
                c32 u[32];
                load_row<R, 2, HALF, true>(u, ub, bf, tid, row_off, a.L);
                fft_fwd<R>(u, c);
                HY_UNROLL
                for (int q = 0; q < 32; ++q) lds_st(mine + q * T, u[q]);
            }
            __syncthreads();                              // (A) U is in LDS
            __syncthreads();                              // (B) the g groups have read it (last step: their sums are in its place)
            // at T = 32 group 1 shares group 0's wavefront and runs the inverse along with it (on its own exchange buffer, storing nothing)
            if (last && want_dk && rg < (T < 64 ? 2 : 1)) {
                // sum of the groups' spectra in group order (bitwise reproducible), then ONE inverse per channel
                c32 sum[32];
                HY_UNROLL
                for (int q = 0; q < 32; ++q) sum[q] = lds_ld(slots + q * T + tid);
                for (int o = 1; o < G; ++o) {
                    HY_UNROLL
                    for (int q = 0; q < 32; ++q) sum[q] = cadd(sum[q], lds_ld(slots + o * C::M + q * T + tid));
                }
                fft_inv<R>(sum, c);
                if (rg == 0) {
                    const float sc = 1.0f / (float)C::M;
                    float* dkrow = a.dk + (size_t)d * a.ldk;
                    HY_UNROLL
                    for (int s = 0; s < 32; ++s) {
                        const c32 w = twist_const<2>(s);
                        const int n = tid + T * s;
                        const float val = (sum[s].x * w.x + sum[s].y * w.y) * sc;
                        if (n < a.L) dkrow[n] = val;
                        if (n == 0 && a.dbias != nullptr) a.dbias[d] = val;
                    }
                }
            }
        } else {
            c32 g[32];
            load_row<R, 2, HALF, true>(g, gb, bf, tid, row_off, a.L);
            fft_fwd<R>(g, c);
            __syncthreads();                              // (A)
            if (want_dk) {
                const float lv = live ? 1.f : 0.f;
                HY_UNROLL
                for (int q = 0; q < 32; ++q) {
                    const c32 p = cmulc(g[q], lds_ld(mine + q * T));                         // G conj(U)
                    acc[q] = mk(acc[q].x + lv * p.x, acc[q].y + lv * p.y);
                }
                if (last) {
                    HY_UNROLL
                    for (int q = 0; q < 32; ++q) lds_st(mine + q * T, acc[q]);
                }
            }
            __syncthreads();                              // (B)
            if (want_du) {
                HY_UNROLL
                for (int q0 = 0; q0 < 32; q0 += 8) {
                    c32 h[8];
                    HY_UNROLL
                    for (int q = 0; q < 8; ++q) h[q] = gb_ld(hb, ho, (unsigned)((q0 + q) * T) * 8u);
                    HY_UNROLL
                    for (int q = 0; q < 8; ++q) g[q0 + q] = cmulc(g[q0 + q], h[q]);              // G conj(H)
                }
                fft_inv<R>(g, c);
                float y[32];
                HY_UNROLL
                for (int s = 0; s < 32; ++s) {
                    const c32 w = twist_const<2>(s);
                    y[s] = g[s].x * w.x + g[s].y * w.y;
                }
                store_row<T, HALF, true>(ob, bf, tid, row_off, live ? a.L : 0, y);
            }
        }
    }
}

}  // namespace oc
}  // namespace hyena
