// fftconv_kernels.h -- device code of the MI355X-native Hyena long convolution.
//
// What is computed (reference: src/models/sequence/hyena.py:59-88, fftconv_ref):
//     out = irfft(rfft(u, N) * rfft(k, N) / N, norm="forward")[..., :L] + u * bias,    N >= 2L
// i.e. a causal linear convolution of every (b, d) row of u with the per-channel filter k[d],
// all arithmetic in fp32, plus its gradients.
//
// How (nothing here follows csrc/fftconv, which tops out at L = 8192 -- SURVEY.md 0.1):
//   * packed real transform: z[n] = u[2n] + i u[2n+1], a complex FFT of M = N/2 points per row;
//   * four-step decomposition M = M1 x 1024 with the intermediate W[k1][n2] held in a workspace sized
//     to stay in the Infinity Cache:
//         col_fwd : strided size-M1 FFTs over n1 (+ twiddle w_M^(n2 k1)), HBM -> W, coalesced on n2
//         row_*   : contiguous size-1024 FFTs over n2 on row pairs (k1, M1-k1), the pointwise product
//                   with the filter spectrum in the packed domain, and the inverse size-1024 FFTs
//         col_inv : inverse strided size-M1 FFTs, W -> HBM
//   * every size-1024 / size-M1 transform is two Stockham stages of radix-32 register butterflies
//     with ONE LDS exchange between them (64-wide wavefronts: a 1024-point row is 32 lanes x 32 points,
//     a wavefront carries the two rows of a (k1, M1-k1) pair, the partner element of lane t is lane 63-t);
//   * the bias term rides in the filter spectrum (Ke += bias), dbias falls out of dk[0]: no extra pass.
//
// Packed-domain product.  With Z = FFT_M(z), H = FFT_M(packed filter), partner index p = (M - k) mod M,
//     He = (H[k] + conj(H[p])) / 2,  Ho = (H[k] - conj(H[p])) / (2i),  w = exp(-2 pi i k / M)
//     conv:  Ke = He (+bias), Ko = Ho          corr:  Ke = conj(He) (+bias), Ko = conj(w Ho)
//     Z'[k] = (Ke + (i/2)(1 - w) Ko) Z[k] + ((i/2)(1 + w) Ko) conj(Z[p])
// and IFFT_M(Z') is the packed output.  (Checked against numpy in scratch models and by tests/.)
//
// This header is compiled by hipcc for gfx950 (the product) and, with -DHIPEMU, by g++ against
// tests/hipemu (CPU emulation used only by the `not gpu` tests).
#pragma once

#ifdef HIPEMU
#include "hipemu.h"
#define HY_SMEM(name) char* name = hipemu::S.smem
#define HY_SHFL_U32(v, lane) hipemu::shfl_u32((v), (lane))
#define HY_UNROLL
#define HY_NOUNROLL
#define HY_SCHED_FENCE() do {} while (0)
#define HY_UNIFORM_PTR(T, p) (p)
#define HY_SGPR(x) (x)
#else
#include <hip/hip_runtime.h>
#define HY_SMEM(name) extern __shared__ __attribute__((aligned(16))) char name[]
#define HY_SHFL_U32(v, lane) ((uint32_t)__shfl((int)(v), (lane), 64))
#define HY_UNROLL _Pragma("unroll")
#define HY_NOUNROLL _Pragma("nounroll")
// stops hipcc from hoisting every load of an unrolled loop to its top (which costs hundreds of VGPRs)
#define HY_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
// pins a wave-uniform pointer into SGPRs (and hides it from loop strength reduction, which otherwise turns
// base + lane offset into 32 loop-carried 64-bit VGPR address pairs)
#define HY_UNIFORM_PTR(T, p) hyena::uniform_ptr<T>(p)
// pins a wave-uniform 32-bit value into an SGPR (integer divisions by run-time values otherwise leave their result in VGPRs)
#define HY_SGPR(x) __builtin_amdgcn_readfirstlane(x)
#endif

#ifdef HIPEMU
#define HY_CONST_TABLE static const
#else
#define HY_CONST_TABLE __device__ static const
#endif

#include <stdint.h>

namespace hyena {

#ifndef HIPEMU
template <typename T>
__device__ __forceinline__ T* uniform_ptr(T* p) {
    const unsigned long long v = reinterpret_cast<unsigned long long>(p);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v);
    const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return reinterpret_cast<T*>(((unsigned long long)hi << 32) | lo);
}
#endif

struct __attribute__((aligned(8))) c32 {
    float x, y;
};

__device__ __forceinline__ c32 mk(float x, float y) { c32 r; r.x = x; r.y = y; return r; }
#if defined(HIPEMU) || !defined(HY_PACKED_F32)
__device__ __forceinline__ c32 cadd(c32 a, c32 b) { return mk(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ c32 csub(c32 a, c32 b) { return mk(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ c32 cmul(c32 a, c32 b) { return mk(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
// a * conj(b)
__device__ __forceinline__ c32 cmulc(c32 a, c32 b) { return mk(a.x * b.x + a.y * b.y, a.y * b.x - a.x * b.y); }
#else
// Packed fp32 VALU (v_pk_add/mul/fma_f32: two fp32 lanes of one 64-bit register pair per instruction, the 157 TFLOP/s
// path of CDNA4).  A complex value IS a register pair, so a complex add is one instruction and a complex multiply two:
//     t = (a.x w.x, a.y w.x)                     v_pk_mul_f32, second source broadcast from its low half
//     r = (-a.y w.y + t.x, a.x w.y + t.y)        v_pk_fma_f32, first source with swapped halves (op_sel) and its low
//                                                 lane negated (neg_lo), second source broadcast from its high half
// Written as inline asm: hipcc's own SLP packing of the scalar formulation scatters the halves over unpaired registers
// (v_mov chains, spills -- hence -fno-slp-vectorize).
// OPT-IN (-DHY_PACKED_F32), off by default: parity-green on MI355X (tests/test_gpu_parity.py) but it changes nothing --
// 6.63 vs 6.64 ms at L = 2^20, 1.545 vs 1.542 ms at L = 32768 x 8 (profiles/README.md, r1ae).  The kernels are bound by
// HBM (B = 1) or by per-wavefront latency at 2 waves per SIMD (B = 8), not by VALU issue, even though their arithmetic
// rate (23-33 TFLOP/s) looks close to the unpacked one-flop-per-lane ceiling.
typedef float hy_f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ hy_f2 tov(c32 a) { hy_f2 v; v.x = a.x; v.y = a.y; return v; }
__device__ __forceinline__ c32 toc(hy_f2 v) { return mk(v.x, v.y); }
__device__ __forceinline__ c32 cadd(c32 a, c32 b) {
    hy_f2 r;
    asm("v_pk_add_f32 %0, %1, %2" : "=v"(r) : "v"(tov(a)), "v"(tov(b)));
    return toc(r);
}
__device__ __forceinline__ c32 csub(c32 a, c32 b) {
    hy_f2 r;
    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(tov(a)), "v"(tov(b)));
    return toc(r);
}
__device__ __forceinline__ c32 cmul(c32 a, c32 b) {
    hy_f2 t, r;
    const hy_f2 av = tov(a), bv = tov(b);
    asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(t) : "v"(av), "v"(bv));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0]" : "=v"(r) : "v"(av), "v"(bv), "v"(t));
    return toc(r);
}
// a * conj(b) = (a.x b.x + a.y b.y, a.y b.x - a.x b.y)
__device__ __forceinline__ c32 cmulc(c32 a, c32 b) {
    hy_f2 t, r;
    const hy_f2 av = tov(a), bv = tov(b);
    asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(t) : "v"(av), "v"(bv));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_hi:[1,0,0]" : "=v"(r) : "v"(av), "v"(bv), "v"(t));
    return toc(r);
}
// -i (a - b) (forward) / +i (a - b) (inverse): the quarter-turn twiddle of a butterfly rides on the subtraction's source modifiers
template <bool INV>
__device__ __forceinline__ c32 csub_rot(c32 a, c32 b) {
    hy_f2 r;
    if (!INV) asm("v_pk_add_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[0,0] neg_lo:[0,1] neg_hi:[1,0]" : "=v"(r) : "v"(tov(a)), "v"(tov(b)));
    else asm("v_pk_add_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[0,0] neg_lo:[1,0] neg_hi:[0,1]" : "=v"(r) : "v"(tov(a)), "v"(tov(b)));
    return toc(r);
}
// a * (c + i s) (CONJ: a * (c - i s)) with a compile-time constant held in a scalar register pair
template <bool CONJ>
__device__ __forceinline__ c32 cmul_k(c32 a, float c, float s) {
    hy_f2 t, r, kv;
    kv.x = c; kv.y = s;
    const hy_f2 av = tov(a);
    asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(t) : "v"(av), "s"(kv));
    if (!CONJ) asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0]" : "=v"(r) : "v"(av), "s"(kv), "v"(t));
    else asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_hi:[1,0,0]" : "=v"(r) : "v"(av), "s"(kv), "v"(t));
    return toc(r);
}
#endif
#if defined(HIPEMU) || !defined(HY_PACKED_F32)
template <bool INV>
__device__ __forceinline__ c32 csub_rot(c32 a, c32 b) {
    const c32 d = csub(a, b);
    return INV ? mk(-d.y, d.x) : mk(d.y, -d.x);
}
#endif
__device__ __forceinline__ c32 rmul(float x, c32 w) { return mk(x * w.x, x * w.y); }          // real x complex (two multiplies by literals)
__device__ __forceinline__ c32 cconj(c32 a) { return mk(a.x, -a.y); }
__device__ __forceinline__ c32 cscale(c32 a, float s) { return mk(a.x * s, a.y * s); }

__device__ __forceinline__ uint32_t f2u(float f) { union { float f; uint32_t u; } c; c.f = f; return c.u; }
__device__ __forceinline__ float u2f(uint32_t u) { union { float f; uint32_t u; } c; c.u = u; return c.f; }

// LDS pointers carry their address space explicitly.  Left generic, hipcc loses track of it once a pointer is
// passed through a function or selected in a loop, and falls back to FLAT instructions (64-bit address pairs,
// spills, slower).  lc32 is a built-in vector type because a struct cannot be copied through an
// address-space-qualified reference.
#ifdef HIPEMU
#define HY_LDS
typedef c32 lc32;
__device__ __forceinline__ c32 lds_ld(const lc32* p) { return *p; }
__device__ __forceinline__ void lds_st(lc32* p, c32 v) { *p = v; }
#define HY_LDS_CAST(T, p) reinterpret_cast<T*>(p)
#else
#define HY_LDS __attribute__((address_space(3)))
typedef float lc32 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ c32 lds_ld(const HY_LDS lc32* p) { const lc32 t = *p; return mk(t.x, t.y); }
__device__ __forceinline__ void lds_st(HY_LDS lc32* p, c32 v) { lc32 t; t.x = v.x; t.y = v.y; *p = t; }
#define HY_LDS_CAST(T, p) ((HY_LDS T*)(p))
#endif

// Global access as wave-uniform base + 32-bit BYTE offset: lets hipcc use the SGPR-base/VGPR-offset form of
// global_load/store instead of materialising a 64-bit address pair per access (which spills).
// HY_POL_LD / HY_POL_ST: cache-policy bits of the accesses to the workspace W (gfx950: 1 = sc0, 2 = nt, 16 = sc1).
// Non-temporal STORES are worth 3 % at every two-level length (write-once streams stop displacing the lines the
// next kernel reads); non-temporal LOADS read 14 % faster in isolation (7.2 vs 6.3 TB/s) but gain < 1 % here at
// L = 2^20 and lose 3 % at L = 160000, where part of W is served by the Infinity Cache
// (profiles/cpol_bw_r2.txt, gpurun_out/r2z_pol2).
#ifndef HY_POL_LD
#define HY_POL_LD 0
#endif
#ifndef HY_POL_ST
#define HY_POL_ST 2
#endif
#if !defined(HIPEMU)
typedef float hy_f2 __attribute__((ext_vector_type(2)));
#endif
__device__ __forceinline__ c32 ldg(const c32* base, unsigned idx) {
#if !defined(HIPEMU)
    if constexpr ((HY_POL_LD & 2) != 0) {
        const hy_f2 t = __builtin_nontemporal_load(reinterpret_cast<const hy_f2*>(reinterpret_cast<const char*>(base) + (size_t)(idx * 8u)));
        return mk(t.x, t.y);
    }
#endif
    return *reinterpret_cast<const c32*>(reinterpret_cast<const char*>(base) + (size_t)(idx * 8u));
}
__device__ __forceinline__ void stg(c32* base, unsigned idx, c32 v) {
#if !defined(HIPEMU)
    if constexpr ((HY_POL_ST & 2) != 0) {
        hy_f2 t; t.x = v.x; t.y = v.y;
        __builtin_nontemporal_store(t, reinterpret_cast<hy_f2*>(reinterpret_cast<char*>(base) + (size_t)(idx * 8u)));
        return;
    }
#endif
    *reinterpret_cast<c32*>(reinterpret_cast<char*>(base) + (size_t)(idx * 8u)) = v;
}

// Buffer-addressed global access for the row kernels: address = descriptor base (SGPRs) + one per-lane VGPR byte
// offset + a scalar byte offset.  All 32 accesses of a lane share ONE address VGPR (hipcc otherwise keeps a
// 64-bit address pair per access alive across the batch loop and spills); out-of-range accesses are dropped by
// the hardware bounds check.
#ifdef HIPEMU
struct GBuf { char* p; };
// Like the hardware descriptor (built from SGPRs via readfirstlane), the emulated one takes the base of the wavefront's
// FIRST lane for every lane: a kernel that hands per-lane bases to make_gbuf fails here as it would on the GPU.
__device__ __forceinline__ GBuf make_gbuf(const void* base, unsigned) {
    const uintptr_t v = (uintptr_t)base;
    const uint32_t lo = hipemu::shfl_u32((uint32_t)v, 0), hi = hipemu::shfl_u32((uint32_t)(v >> 32), 0);
    GBuf b;
    b.p = (char*)(((uintptr_t)hi << 32) | lo);
    return b;
}
__device__ __forceinline__ c32 gb_ld(GBuf b, unsigned voff, unsigned soff) { return *reinterpret_cast<const c32*>(b.p + voff + soff); }
__device__ __forceinline__ void gb_st(GBuf b, unsigned voff, unsigned soff, c32 v) { *reinterpret_cast<c32*>(b.p + voff + soff) = v; }
__device__ __forceinline__ c32 gb_ldt(GBuf b, unsigned voff, unsigned soff) { return gb_ld(b, voff, soff); }
struct __attribute__((aligned(16))) c32x2 { c32 a, b; };
__device__ __forceinline__ c32x2 gb_ld2(GBuf b, unsigned voff, unsigned soff) { return *reinterpret_cast<const c32x2*>(b.p + voff + soff); }
__device__ __forceinline__ void gb_st2(GBuf b, unsigned voff, unsigned soff, c32x2 v) { *reinterpret_cast<c32x2*>(b.p + voff + soff) = v; }
#define HY_OPAQUE(x) do {} while (0)
#else
struct GBuf { __amdgpu_buffer_rsrc_t r; };
typedef unsigned hy_u2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ GBuf make_gbuf(const void* base, unsigned bytes) {
    GBuf b;
    b.r = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr(const_cast<void*>(base)), 0, bytes, 0x00020000);
    return b;
}
__device__ __forceinline__ c32 gb_ld(GBuf b, unsigned voff, unsigned soff) {
    const hy_u2 w = __builtin_amdgcn_raw_buffer_load_b64(b.r, voff, soff, HY_POL_LD);
    return mk(u2f(w.x), u2f(w.y));
}
__device__ __forceinline__ c32 gb_ldt(GBuf b, unsigned voff, unsigned soff) {      // tables: re-used, default policy
    const hy_u2 w = __builtin_amdgcn_raw_buffer_load_b64(b.r, voff, soff, 0);
    return mk(u2f(w.x), u2f(w.y));
}
__device__ __forceinline__ void gb_st(GBuf b, unsigned voff, unsigned soff, c32 v) {
    // STORES never use the scalar-offset field: on gfx950 a buffer_store with an SGPR soffset was observed to read
    // its data VGPRs late -- a VALU instruction issued a few slots later that re-used a data register (here: the
    // next iteration's LDS index) leaked into the stored value for lanes 12-15/28-31 of either half-wave, rarely
    // and timing-dependently (hipcc's hazard recogniser exempts stores that have a register soffset).  With the
    // offset folded into voffset/immediate the store is an ordinary one.  Found by
    // tests/test_gpu_parity.py::test_properties_at_baseline_sizes (causality / determinism at D = 256).
    hy_u2 w; w.x = f2u(v.x); w.y = f2u(v.y);
    __builtin_amdgcn_raw_buffer_store_b64(w, b.r, voff + soff, 0, HY_POL_ST);
}
struct __attribute__((aligned(16))) c32x2 { c32 a, b; };
typedef unsigned hy_u4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ c32x2 gb_ld2(GBuf b, unsigned voff, unsigned soff) {
    const hy_u4 w = __builtin_amdgcn_raw_buffer_load_b128(b.r, voff, soff, HY_POL_LD);
    c32x2 r; r.a = mk(u2f(w.x), u2f(w.y)); r.b = mk(u2f(w.z), u2f(w.w)); return r;
}
__device__ __forceinline__ void gb_st2(GBuf b, unsigned voff, unsigned soff, c32x2 v) {
    gb_st(b, voff, soff, v.a);          // two 8-byte stores (see gb_st for why not one dwordx4 with soffset)
    gb_st(b, voff + 8u, soff, v.b);
}
// makes a value opaque to the optimiser at this point (keeps "lane base + immediate" LDS addressing intact)
#define HY_OPAQUE(x) asm volatile("" : "+v"(x))
#endif

__device__ __forceinline__ c32 shfl_c32(c32 v, int lane) {
    return mk(u2f(HY_SHFL_U32(f2u(v.x), lane)), u2f(HY_SHFL_U32(f2u(v.y), lane)));
}

// ---------------------------------------------------------------------------------------------
// element types
// ---------------------------------------------------------------------------------------------
enum { DT_F32 = 0, DT_BF16 = 1, DT_F16 = 2 };

__device__ __forceinline__ float bf16_to_f32(uint16_t h) { return u2f(((uint32_t)h) << 16); }
#ifdef HIPEMU
__device__ __forceinline__ uint16_t f32_to_bf16(float f) {   // round to nearest even, quiet NaN
    uint32_t u = f2u(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x0040u);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
#else
// gfx950 converts in hardware (v_cvt_pk_bf16_f32: round to nearest even, quiet NaN) -- one instruction instead of six
__device__ __forceinline__ uint16_t f32_to_bf16(float f) {
    const __bf16 h = (__bf16)f;
    uint16_t u;
    __builtin_memcpy(&u, &h, 2);
    return u;
}
#endif
#ifdef HIPEMU
__device__ __forceinline__ float f16_to_f32(uint16_t h) {
    uint32_t s = ((uint32_t)h & 0x8000u) << 16, e = (h >> 10) & 0x1fu, m = h & 0x3ffu;
    if (e == 0) {
        if (m == 0) return u2f(s);
        float v = (float)m * 5.9604644775390625e-08f;   // 2^-24
        return s ? -v : v;
    }
    if (e == 31) return u2f(s | 0x7f800000u | (m << 13));
    return u2f(s | ((e + 112u) << 23) | (m << 13));
}
__device__ __forceinline__ uint16_t f32_to_f16(float f) {   // round to nearest even
    uint32_t u = f2u(f), s = (u >> 16) & 0x8000u, a = u & 0x7fffffffu;
    if (a > 0x7f800000u) return (uint16_t)(s | 0x7e00u);
    if (a >= 0x47800000u) return (uint16_t)(s | 0x7c00u);       // >= 65536 -> inf (65520 rounds to inf below)
    if (a < 0x33000000u) return (uint16_t)s;                    // < 2^-25 -> 0
    if (a < 0x38800000u) {                                       // subnormal half
        float v = u2f(a) * 16777216.0f;                          // * 2^24 -> integer part is the mantissa
        uint32_t m = (uint32_t)v;
        float r = v - (float)m;
        if (r > 0.5f || (r == 0.5f && (m & 1u))) ++m;
        return (uint16_t)(s | m);
    }
    uint32_t m = a & 0x7fffffu, e = (a >> 23) - 112u;
    uint32_t h = (e << 10) | (m >> 13);
    uint32_t rem = m & 0x1fffu;
    if (rem > 0x1000u || (rem == 0x1000u && (h & 1u))) ++h;      // may carry into the exponent (and to inf): correct
    return (uint16_t)(s | h);
}
#else
__device__ __forceinline__ float f16_to_f32(uint16_t h) {
    union { uint16_t u; _Float16 f; } c; c.u = h; return (float)c.f;
}
__device__ __forceinline__ uint16_t f32_to_f16(float f) {
    union { uint16_t u; _Float16 f; } c; c.f = (_Float16)f; return c.u;
}
#endif

// HY_POL_IO: cache policy of the operator's own tensors (u, k, dout read once; out, du, dk written once)
#ifndef HY_POL_IO
#define HY_POL_IO 2
#endif
template <class T> __device__ __forceinline__ T io_ldp(const T* p) {
#if !defined(HIPEMU)
    if constexpr ((HY_POL_IO & 2) != 0) return __builtin_nontemporal_load(p);
#endif
    return *p;
}
template <class T> __device__ __forceinline__ void io_stp(T* p, T v) {
#if !defined(HIPEMU)
    if constexpr ((HY_POL_IO & 2) != 0) { __builtin_nontemporal_store(v, p); return; }
#endif
    *p = v;
}
template <int DT> struct Elem;
template <> struct Elem<DT_F32> {
    typedef float type;
    static __device__ __forceinline__ float cvt(float v) { return v; }
    static __device__ __forceinline__ float dec(float r) { return r; }
    static __device__ __forceinline__ float ld(const float* p) { return io_ldp(p); }
    static __device__ __forceinline__ void st(float* p, float v) { io_stp(p, v); }
#if defined(HIPEMU)
    static __device__ __forceinline__ c32 ld2(const float* p) { return *reinterpret_cast<const c32*>(p); }
    static __device__ __forceinline__ void st2(float* p, c32 v) { *reinterpret_cast<c32*>(p) = v; }
#else
    static __device__ __forceinline__ c32 ld2(const float* p) { const hy_f2 t = io_ldp(reinterpret_cast<const hy_f2*>(p)); return mk(t.x, t.y); }
    static __device__ __forceinline__ void st2(float* p, c32 v) { hy_f2 t; t.x = v.x; t.y = v.y; io_stp(reinterpret_cast<hy_f2*>(p), t); }
#endif
};
template <> struct Elem<DT_BF16> {
    typedef uint16_t type;
    static __device__ __forceinline__ uint16_t cvt(float v) { return f32_to_bf16(v); }
    static __device__ __forceinline__ float dec(uint16_t r) { return bf16_to_f32(r); }
    static __device__ __forceinline__ float ld(const uint16_t* p) { return bf16_to_f32(io_ldp(p)); }
    static __device__ __forceinline__ void st(uint16_t* p, float v) { io_stp(p, f32_to_bf16(v)); }
    static __device__ __forceinline__ c32 ld2(const uint16_t* p) {
        uint32_t w = io_ldp(reinterpret_cast<const uint32_t*>(p));
        return mk(u2f(w << 16), u2f(w & 0xffff0000u));
    }
    static __device__ __forceinline__ void st2(uint16_t* p, c32 v) {
        io_stp(reinterpret_cast<uint32_t*>(p), (uint32_t)f32_to_bf16(v.x) | ((uint32_t)f32_to_bf16(v.y) << 16));
    }
};
template <> struct Elem<DT_F16> {
    typedef uint16_t type;
    static __device__ __forceinline__ uint16_t cvt(float v) { return f32_to_f16(v); }
    static __device__ __forceinline__ float dec(uint16_t r) { return f16_to_f32(r); }
    static __device__ __forceinline__ float ld(const uint16_t* p) { return f16_to_f32(io_ldp(p)); }
    static __device__ __forceinline__ void st(uint16_t* p, float v) { io_stp(p, f32_to_f16(v)); }
    static __device__ __forceinline__ c32 ld2(const uint16_t* p) {
        uint32_t w = io_ldp(reinterpret_cast<const uint32_t*>(p));
        return mk(f16_to_f32((uint16_t)(w & 0xffffu)), f16_to_f32((uint16_t)(w >> 16)));
    }
    static __device__ __forceinline__ void st2(uint16_t* p, c32 v) {
        io_stp(reinterpret_cast<uint32_t*>(p), (uint32_t)f32_to_f16(v.x) | ((uint32_t)f32_to_f16(v.y) << 16));
    }
};

// Pair access z[n] = (x[2n], x[2n+1]) of a real row.  A pair is moved as ONE 4-byte (16-bit types) or 8-byte (fp32)
// access that may be under-aligned (odd L: odd rows start 2 / 4 bytes off) -- gfx950 global memory handles that, and
// __builtin_memcpy lets hipcc emit a single global_load/store_dword(x2).  Keeping the raw word and converting
// later lets a thread issue all its loads back to back (branchy element-wise loads were serialised by hipcc with
// an s_waitcnt vmcnt(0) after every one).
template <int DT> struct Pair;
template <> struct Pair<DT_F32> {
    typedef c32 raw_t;
    static __device__ __forceinline__ c32 cvt(raw_t w) { return w; }
    static __device__ __forceinline__ raw_t pack(c32 v) { return v; }
};
template <> struct Pair<DT_BF16> {
    typedef uint32_t raw_t;
    static __device__ __forceinline__ c32 cvt(raw_t w) { return mk(u2f(w << 16), u2f(w & 0xffff0000u)); }
    static __device__ __forceinline__ raw_t pack(c32 v) { return (uint32_t)f32_to_bf16(v.x) | ((uint32_t)f32_to_bf16(v.y) << 16); }
};
template <> struct Pair<DT_F16> {
    typedef uint32_t raw_t;
    static __device__ __forceinline__ c32 cvt(raw_t w) { return mk(f16_to_f32((uint16_t)(w & 0xffffu)), f16_to_f32((uint16_t)(w >> 16))); }
    static __device__ __forceinline__ raw_t pack(c32 v) { return (uint32_t)f32_to_f16(v.x) | ((uint32_t)f32_to_f16(v.y) << 16); }
};
template <int DT>
__device__ __forceinline__ typename Pair<DT>::raw_t load_raw_pair(const typename Elem<DT>::type* row, int n) {
    typename Pair<DT>::raw_t w;
    __builtin_memcpy(&w, row + 2 * n, sizeof(w));
    return w;
}
template <int DT>
__device__ __forceinline__ void store_raw_pair(typename Elem<DT>::type* row, int n, typename Pair<DT>::raw_t w) {
    __builtin_memcpy(row + 2 * n, &w, sizeof(w));
}

// ---------------------------------------------------------------------------------------------
// register butterflies: N-point DFT (N = 1..32, power of two) of v[0..N), natural order in and out.
// Decimation in frequency with compile-time twiddles; the final bit reversal is register renaming.
// ---------------------------------------------------------------------------------------------
__host__ __device__ constexpr float tw32_cos(int t) {   // cos(2 pi t / 32), t = 0..15
    return t == 0 ? 1.0f : t == 1 ? 0.98078528040323044913f : t == 2 ? 0.92387953251128675613f
         : t == 3 ? 0.83146961230254523708f : t == 4 ? 0.70710678118654752440f : t == 5 ? 0.55557023301960222474f
         : t == 6 ? 0.38268343236508977173f : t == 7 ? 0.19509032201612826785f : t == 8 ? 0.0f
         : t == 9 ? -0.19509032201612826785f : t == 10 ? -0.38268343236508977173f : t == 11 ? -0.55557023301960222474f
         : t == 12 ? -0.70710678118654752440f : t == 13 ? -0.83146961230254523708f : t == 14 ? -0.92387953251128675613f
         : -0.98078528040323044913f;
}
__host__ __device__ constexpr float tw32_sin(int t) {   // sin(2 pi t / 32), t = 0..15
    return t == 0 ? 0.0f : t == 1 ? 0.19509032201612826785f : t == 2 ? 0.38268343236508977173f
         : t == 3 ? 0.55557023301960222474f : t == 4 ? 0.70710678118654752440f : t == 5 ? 0.83146961230254523708f
         : t == 6 ? 0.92387953251128675613f : t == 7 ? 0.98078528040323044913f : t == 8 ? 1.0f
         : t == 9 ? 0.98078528040323044913f : t == 10 ? 0.92387953251128675613f : t == 11 ? 0.83146961230254523708f
         : t == 12 ? 0.70710678118654752440f : t == 13 ? 0.55557023301960222474f : t == 14 ? 0.38268343236508977173f
         : 0.19509032201612826785f;
}
__host__ __device__ constexpr int ilog2(int n) { return n <= 1 ? 0 : 1 + ilog2(n >> 1); }
__host__ __device__ constexpr int brev(int q, int bits) {
    int r = 0;
    for (int i = 0; i < bits; ++i) r |= ((q >> i) & 1) << (bits - 1 - i);
    return r;
}

// d * w_32^t (forward) or d * conj(w_32^t) (inverse); t is a compile-time constant after unrolling.
template <bool INV>
__device__ __forceinline__ c32 mul_tw32(c32 d, int t) {
    const float r = 0.70710678118654752440f;
    if (t == 0) return d;
#if !defined(HIPEMU) && defined(HY_PACKED_F32)
    if (t != 8) return cmul_k<!INV>(d, tw32_cos(t), tw32_sin(t));       // two packed instructions (t = 8 never gets here: csub_rot)
#endif
    if (t == 8) return INV ? mk(-d.y, d.x) : mk(d.y, -d.x);
    if (t == 4) return INV ? mk((d.x - d.y) * r, (d.x + d.y) * r) : mk((d.x + d.y) * r, (d.y - d.x) * r);
    if (t == 12) return INV ? mk(-(d.x + d.y) * r, (d.x - d.y) * r) : mk((d.y - d.x) * r, -(d.x + d.y) * r);
    const float c = tw32_cos(t), s = tw32_sin(t);
    return INV ? mk(d.x * c - d.y * s, d.y * c + d.x * s) : mk(d.x * c + d.y * s, d.y * c - d.x * s);
}

template <int N, bool INV>
__device__ __forceinline__ void dft_reg(c32 (&v)[N]) {
    constexpr int LG = ilog2(N);
    HY_UNROLL
    for (int st = 0; st < LG; ++st) {
        const int len = N >> st, half = len >> 1, tstep = 32 / len;
        HY_UNROLL
        for (int base = 0; base < N; base += len) {
            HY_UNROLL
            for (int j = 0; j < half; ++j) {
                const c32 a = v[base + j], b = v[base + j + half];
                v[base + j] = cadd(a, b);
                v[base + j + half] = (j * tstep == 8) ? csub_rot<INV>(a, b) : mul_tw32<INV>(csub(a, b), j * tstep);
            }
        }
    }
    HY_UNROLL
    for (int q = 0; q < N; ++q) {
        const int p = brev(q, LG);
        if (q < p) { const c32 t = v[q]; v[q] = v[p]; v[p] = t; }
    }
}

// 5-point DFT (the odd factor of M1 = 160 = 32 x 5, which serves L = 160000 with N = 327680 instead of 524288)
template <bool INV>
__device__ __forceinline__ void dft5_reg(c32 (&v)[5]) {
    const float c1 = 0.30901699437494742410f, c2 = -0.80901699437494742410f;     // cos(2 pi / 5), cos(4 pi / 5)
    const float s1 = 0.95105651629515357212f, s2 = 0.58778525229247312917f;      // sin(2 pi / 5), sin(4 pi / 5)
    const c32 t1 = cadd(v[1], v[4]), t2 = cadd(v[2], v[3]), t3 = csub(v[1], v[4]), t4 = csub(v[2], v[3]);
    const c32 m1 = mk(v[0].x + c1 * t1.x + c2 * t2.x, v[0].y + c1 * t1.y + c2 * t2.y);
    const c32 m2 = mk(v[0].x + c2 * t1.x + c1 * t2.x, v[0].y + c2 * t1.y + c1 * t2.y);
    const c32 n1 = mk(s1 * t3.x + s2 * t4.x, s1 * t3.y + s2 * t4.y);
    const c32 n2 = mk(s2 * t3.x - s1 * t4.x, s2 * t3.y - s1 * t4.y);
    v[0] = cadd(v[0], cadd(t1, t2));
    // forward: y1 = m1 - i n1, y4 = m1 + i n1, y2 = m2 - i n2, y3 = m2 + i n2;  inverse: signs swapped
    const c32 a1 = mk(m1.x + n1.y, m1.y - n1.x), b1 = mk(m1.x - n1.y, m1.y + n1.x);
    const c32 a2 = mk(m2.x + n2.y, m2.y - n2.x), b2 = mk(m2.x - n2.y, m2.y + n2.x);
    v[1] = INV ? b1 : a1;
    v[4] = INV ? a1 : b1;
    v[2] = INV ? b2 : a2;
    v[3] = INV ? a2 : b2;
}
// 7-point DFT and, on top of it, the 14-point DFT (one radix-2 DIF step + two 7-point DFTs): the odd factor of
// M1 = 448 = 32 x 14, which serves 262144 < L <= 458752 (L = 450560: N = 917504 instead of 1048576)
template <bool INV>
__device__ __forceinline__ void dft7_reg(c32 (&v)[7]) {
    const float c1 = 0.62348980185873353053f, c2 = -0.22252093395631440429f, c3 = -0.90096886790241912624f;
    const float s1 = 0.78183148246802980871f, s2 = 0.97492791218182360702f, s3 = 0.43388373911755812048f;
    const c32 t1 = cadd(v[1], v[6]), t2 = cadd(v[2], v[5]), t3 = cadd(v[3], v[4]);
    const c32 d1 = csub(v[1], v[6]), d2 = csub(v[2], v[5]), d3 = csub(v[3], v[4]);
    const c32 x0 = v[0];
    v[0] = cadd(x0, cadd(t1, cadd(t2, t3)));
    const c32 a1 = mk(x0.x + c1 * t1.x + c2 * t2.x + c3 * t3.x, x0.y + c1 * t1.y + c2 * t2.y + c3 * t3.y);
    const c32 a2 = mk(x0.x + c2 * t1.x + c3 * t2.x + c1 * t3.x, x0.y + c2 * t1.y + c3 * t2.y + c1 * t3.y);
    const c32 a3 = mk(x0.x + c3 * t1.x + c1 * t2.x + c2 * t3.x, x0.y + c3 * t1.y + c1 * t2.y + c2 * t3.y);
    const c32 b1 = mk(s1 * d1.x + s2 * d2.x + s3 * d3.x, s1 * d1.y + s2 * d2.y + s3 * d3.y);
    const c32 b2 = mk(s2 * d1.x - s3 * d2.x - s1 * d3.x, s2 * d1.y - s3 * d2.y - s1 * d3.y);
    const c32 b3 = mk(s3 * d1.x - s1 * d2.x + s2 * d3.x, s3 * d1.y - s1 * d2.y + s2 * d3.y);
    // forward: X[k] = a_k - i b_k, X[7 - k] = a_k + i b_k;  inverse: signs swapped
    const c32 m1 = mk(a1.x + b1.y, a1.y - b1.x), p1 = mk(a1.x - b1.y, a1.y + b1.x);
    const c32 m2 = mk(a2.x + b2.y, a2.y - b2.x), p2 = mk(a2.x - b2.y, a2.y + b2.x);
    const c32 m3 = mk(a3.x + b3.y, a3.y - b3.x), p3 = mk(a3.x - b3.y, a3.y + b3.x);
    v[1] = INV ? p1 : m1; v[6] = INV ? m1 : p1;
    v[2] = INV ? p2 : m2; v[5] = INV ? m2 : p2;
    v[3] = INV ? p3 : m3; v[4] = INV ? m3 : p3;
}
template <bool INV>
__device__ __forceinline__ void dft14_reg(c32 (&v)[14]) {
    const float cw[7] = {1.0f, 0.90096886790241912624f, 0.62348980185873353053f, 0.22252093395631440429f,
                         -0.22252093395631440429f, -0.62348980185873353053f, -0.90096886790241912624f};
    const float sw[7] = {0.0f, 0.43388373911755812048f, 0.78183148246802980871f, 0.97492791218182360702f,
                         0.97492791218182360702f, 0.78183148246802980871f, 0.43388373911755812048f};
    c32 a[7], b[7];
    HY_UNROLL
    for (int m = 0; m < 7; ++m) {
        a[m] = cadd(v[m], v[m + 7]);
        const c32 d = csub(v[m], v[m + 7]);
        // d * w_14^m (forward) / d * conj(w_14^m) (inverse)
        b[m] = INV ? mk(d.x * cw[m] - d.y * sw[m], d.y * cw[m] + d.x * sw[m]) : mk(d.x * cw[m] + d.y * sw[m], d.y * cw[m] - d.x * sw[m]);
    }
    dft7_reg<INV>(a);
    dft7_reg<INV>(b);
    HY_UNROLL
    for (int k = 0; k < 7; ++k) { v[2 * k] = a[k]; v[2 * k + 1] = b[k]; }
}
template <bool INV>
__device__ __forceinline__ void dft3_reg(c32 (&v)[3]) {
    const float s = 0.86602540378443864676f;                  // sin(2 pi / 3); cos = -1/2
    const c32 t = cadd(v[1], v[2]), d = csub(v[1], v[2]);
    const c32 m = mk(v[0].x - 0.5f * t.x, v[0].y - 0.5f * t.y);
    const c32 b = mk(s * d.x, s * d.y);
    v[0] = cadd(v[0], t);
    const c32 lo = mk(m.x + b.y, m.y - b.x), hi = mk(m.x - b.y, m.y + b.x);      // m - i b, m + i b
    v[1] = INV ? hi : lo;
    v[2] = INV ? lo : hi;
}
template <int P, bool INV>
__device__ __forceinline__ void dft_odd_reg(c32 (&v)[P]) {
    if constexpr (P == 3) dft3_reg<INV>(v);
    else if constexpr (P == 5) dft5_reg<INV>(v);
    else dft7_reg<INV>(v);
}
__host__ __device__ constexpr int odd_part(int n) { return (n & 1) ? n : odd_part(n >> 1); }
__host__ __device__ constexpr bool is_pow2(int n) { return (n & (n - 1)) == 0; }

// N = 2^A * P points (P = 3, 5 or 7): A radix-2 decimation-in-frequency steps with twiddles w_N^j = tw[j * tw_step]
// (an LDS table of w_M1^j; conjugated for the inverse), then 2^A P-point DFTs; block b of the result holds the outputs
// k = bitrev_A(b) + 2^A k', which the final permutation puts in natural order.  Serves the column sizes that are not
// powers of two (M1 = 3 ... 28 in one stage, 32 x {3, 6, 7, 10, 12, 20, 24} as the second stage).
template <int N, bool INV>
__device__ __forceinline__ void dft_mixed(c32 (&v)[N], const HY_LDS lc32* tw, int tw_step) {
    constexpr int P = odd_part(N), NBLK = N / P, A = ilog2(NBLK);
    HY_UNROLL
    for (int st = 0; st < A; ++st) {
        const int len = N >> st, half = len >> 1;
        HY_UNROLL
        for (int base = 0; base < N; base += len) {
            HY_UNROLL
            for (int j = 0; j < half; ++j) {
                const c32 a = v[base + j], b = v[base + j + half];
                v[base + j] = cadd(a, b);
                const c32 d = csub(a, b);
                if (j == 0) v[base + j + half] = d;
                else {
                    const c32 w = lds_ld(tw + j * (N / len) * tw_step);
                    v[base + j + half] = INV ? cmulc(d, w) : cmul(d, w);
                }
            }
        }
    }
    c32 o[N];
    HY_UNROLL
    for (int b = 0; b < NBLK; ++b) {
        c32 blk[P];
        HY_UNROLL
        for (int k = 0; k < P; ++k) blk[k] = v[b * P + k];
        dft_odd_reg<P, INV>(blk);
        HY_UNROLL
        for (int k = 0; k < P; ++k) o[brev(b, A) + NBLK * k] = blk[k];
    }
    HY_UNROLL
    for (int q = 0; q < N; ++q) v[q] = o[q];
}
// N-point register DFT of a column stage; `tw[j * tw_step]` = w_N^j (needed for the sizes that are not powers of two)
template <int N, bool INV>
__device__ __forceinline__ void dft_any(c32 (&y)[N], const HY_LDS lc32* tw, int tw_step) {
    if constexpr (is_pow2(N)) dft_reg<N, INV>(y);
    else if constexpr (N == 3 || N == 5 || N == 7) dft_odd_reg<N, INV>(y);
    else if constexpr (N == 14) dft14_reg<INV>(y);
    else dft_mixed<N, INV>(y, tw, tw_step);
}

#ifndef HY_HELPERS_ONLY      // (onchip_kernels.h takes the helpers above and none of the two-level kernels below)
// ---------------------------------------------------------------------------------------------
// 1024-point row transform on a half-wavefront: lane j (0..31) holds v[s] = x[j + 32 s] on entry and
// X[j + 32 q] on exit (Stockham 32 x 32, natural order).  `xb` = this half's LDS exchange buffer of
// ROW_LDS c32 (index p -> p + p/32: conflict-free ds_write_b64 / ds_read_b64).  `twT[s*32 + j]` =
// w_1024^(j s).  All 64 lanes of the workgroup must call it (it contains workgroup barriers).
// ---------------------------------------------------------------------------------------------
enum { ROW_N = 1024, ROW_LDS = 1024 + 32 };

__device__ __forceinline__ int row_idx(int p) { return p + (p >> 5); }

// Stage twiddles w_1024^(j s), s = 8a + b, as the product tA[a] * tB[b] of ten table entries per lane
// (tB[b] = w^(j b), b = 1..7; tA[a] = w^(8 j a), a = 1..3): 10 loads (kept in 20 VGPRs for every transform of the
// kernel) instead of 31, at the price of one extra rounding in 21 of the 31 twiddles.
struct RowTw {
    c32 tB[8];   // [0] unused
    c32 tA[4];   // [0] unused
};
__device__ __forceinline__ void load_row_tw(RowTw& t, GBuf twT, int j) {
    HY_UNROLL
    for (int b = 1; b < 8; ++b) t.tB[b] = gb_ldt(twT, (unsigned)j * 8u, (unsigned)b * 256u);
    HY_UNROLL
    for (int a = 1; a < 4; ++a) t.tA[a] = gb_ldt(twT, (unsigned)j * 8u, (unsigned)(8 * a) * 256u);
}

template <bool INV>
__device__ __forceinline__ void row_fft1024(c32 (&v)[32], HY_LDS lc32* xb, int j, const RowTw& t) {
    dft_reg<32, INV>(v);
    HY_UNROLL
    for (int q = 0; q < 32; ++q) lds_st(xb + j * 33 + q, v[q]);     // position j*32 + q
    __syncthreads();
    HY_UNROLL
    for (int s = 0; s < 32; ++s) v[s] = lds_ld(xb + j + 33 * s);    // position j + 32 s
    __syncthreads();
    HY_UNROLL
    for (int s = 1; s < 32; ++s) {
        const int a = s >> 3, b = s & 7;
        c32 ta = t.tA[a & 3], tb = t.tB[b];
        // opaque copies: otherwise hipcc hoists all 21 products out of the batch loop / shares them between the
        // transforms of a kernel, i.e. keeps 31 twiddles (62 VGPRs) alive and spills
        HY_OPAQUE(ta.x); HY_OPAQUE(ta.y); HY_OPAQUE(tb.x); HY_OPAQUE(tb.y);
        const c32 w = (a == 0) ? tb : (b == 0) ? ta : cmul(ta, tb);
        v[s] = INV ? cmulc(v[s], w) : cmul(v[s], w);
    }
    dft_reg<32, INV>(v);
}

// ---------------------------------------------------------------------------------------------
// tables (device memory, built by the host in double precision)
//   tw_lo[p]   = w_M^p,            p < 1024
//   tw_hi[p]   = w_M^(1024 p) = w_M1^p,   p < M1      (stored right behind tw_lo: one contiguous copy to LDS)
//   tw_row[p]  = w_1024^p,         p < 1024
//   tw_rowT[s*32 + j] = w_1024^(j s)
// ---------------------------------------------------------------------------------------------
struct Tables {
    const c32* tw_lo;
    const c32* tw_hi;
    const c32* tw_row;
    const c32* tw_rowT;
};

// ---------------------------------------------------------------------------------------------
// column kernels.  One workgroup = C adjacent columns n2 of one row; thread (c, r) with r < T,
// T = max(1, M1/32).  Each thread owns E = min(M1, 32) points of its column.
// ---------------------------------------------------------------------------------------------
#ifndef HY_COLW
#define HY_COLW(m1) (((m1) == 28 || (m1) == 768) ? 3 : 4)
#endif
template <int M1> struct ColCfg {
    static constexpr int T = M1 >= 32 ? M1 / 32 : 1;
    static constexpr int E = M1 >= 32 ? 32 : M1;
    static_assert(M1 < 32 || M1 % 32 == 0, "two-stage column sizes are 32 x T");
    static constexpr bool POW2 = is_pow2(T);                 // T = 3, 5, 6, 7, 10, 12, 14, 20, 24: the odd-factor sizes
    // (32 columns per workgroup -- 256-byte row pieces, one 1024-thread workgroup per CU -- measured slower at M1 = 1024:
    // 7.3 vs 6.7 ms per step)
    static constexpr int C = T == 1 ? 256 : POW2 ? ((256 / T) > 16 ? (256 / T) : 16)
                           : (T == 3 || T == 5 || T == 7) ? 64 : (T == 6 || T == 10 || T == 14) ? 32 : 16;
    static constexpr int THREADS = C * T;
    static_assert(THREADS % 64 == 0, "whole wavefronts");
    static constexpr int NB = (32 + T - 1) / T;              // stage-2 butterflies per thread (T > 1); for T = 5 the
                                                             // 32 of them split 7/7/6/6/6 (jj = r + T i < 32)
    static constexpr int TO = (T + 1) / 2;                   // stage-2 outputs q < TO can lie below M1/2
    static constexpr int EH = E >= 2 ? (E + 1) / 2 : 1;      // single-stage sizes: inputs / outputs n1 < EH can lie below M1/2
    // waves per SIMD the register budget is set for: 4 (<= 128 VGPRs) everywhere except M1 = 28 and 768, which measured
    // faster at 3 (28000 x 8: 1.32 vs 1.37 ms; 700000: 6.31 vs 6.51 ms) -- M1 = 640 prefers 4 with a 52-byte spill
    // (600000: 5.20 vs 5.39 ms)
    static constexpr int MIN_WAVES = HY_COLW(M1);
    static constexpr size_t LDS_TABLES = (1024 + (size_t)M1) * sizeof(c32);
    // (M1 = 1024: 64 KB plane + 16 KB tables = exactly half a CU's LDS.  The plane's writes have a 4-way bank conflict at
    // C = 16; both cures measured worse: padding pushes it to one workgroup per CU (8.24 vs 6.64 ms per step), an XOR row
    // swizzle turns the immediate-offset ds_writes into per-lane address arithmetic in a VALU-tight kernel (7.51 ms))
    static constexpr size_t LDS_PLANE = T > 1 ? (size_t)M1 * C * sizeof(float) : 0;
    static constexpr size_t LDS = LDS_TABLES + LDS_PLANE;
};

struct ColArgs {
    const void* x;      // real rows (col_fwd input / col_inv output)
    c32* W;             // [rows][M1][1024]
    Tables tab;
    int L;              // real samples per row
    int inner;          // rows per outer index (channels in the chunk)
    long outer_stride;  // elements between consecutive outer indices (D * L)
    long inner_stride;  // elements between consecutive inner indices (L)
    float* aux0;        // col_inv only: if non-null, aux0[row] = real part of output sample 0
    int w_bstride;      // rows of W between consecutive outer (batch) indices: `inner` for a chunk-local workspace,
                        // D when W is the persistent saved-spectrum buffer [B][D][M]
    const void* x2;     // col_fwd only: second input tensor of the same shape (blockIdx.z == 1) ...
    c32* W2;            // ... and where its transform goes; lets dout and u share one launch in the backward
};

__device__ __forceinline__ c32 outer_tw(const HY_LDS lc32* tlo, const HY_LDS lc32* thi, int n2, int k1) {
    const int p = n2 * k1;
    return cmul(lds_ld(tlo + (p & 1023)), lds_ld(thi + (p >> 10)));
}

// copy tw_lo | tw_hi (contiguous, 1024 + M1 entries) into LDS: all loads first, then all LDS writes
template <int M1, int THREADS>
__device__ __forceinline__ void stage_col_tables(HY_LDS lc32* tab, const c32* __restrict__ src, int tid) {
    constexpr int NTAB = 1024 + M1, NT = (NTAB + THREADS - 1) / THREADS;
    c32 tt[NT];
    HY_UNROLL
    for (int i = 0; i < NT; ++i) {
        const int idx = tid + i * THREADS;
        tt[i] = src[idx < NTAB ? idx : NTAB - 1];
    }
    HY_UNROLL
    for (int i = 0; i < NT; ++i) {
        const int idx = tid + i * THREADS;
        if (idx < NTAB) lds_st(tab + idx, tt[i]);
    }
}

// LDS exchange of the two Stockham stages of a column transform, one float plane at a time:
// position p of column c lives at plane[p*C + c]; thread (c, r) writes p = r*32 + q, reads p = (r + T i) + 32 s.
template <int T, int C, int NB>
__device__ __forceinline__ void col_exchange(const c32 (&v)[32], c32 (&x2)[NB * T], HY_LDS float* plane, int c, int r) {
    HY_UNROLL
    for (int q = 0; q < 32; ++q) plane[(r * 32 + q) * C + c] = v[q].x;
    __syncthreads();
    HY_UNROLL
    for (int i = 0; i < NB; ++i) {
        const int jj = (NB * T == 32 || r + T * i < 32) ? r + T * i : 0;       // T = 5: the last butterfly of r >= 2 is idle
        HY_UNROLL
        for (int s = 0; s < T; ++s) x2[i * T + s].x = plane[(jj + 32 * s) * C + c];
    }
    __syncthreads();
    HY_UNROLL
    for (int q = 0; q < 32; ++q) plane[(r * 32 + q) * C + c] = v[q].y;
    __syncthreads();
    HY_UNROLL
    for (int i = 0; i < NB; ++i) {
        const int jj = (NB * T == 32 || r + T * i < 32) ? r + T * i : 0;
        HY_UNROLL
        for (int s = 0; s < T; ++s) x2[i * T + s].y = plane[(jj + 32 * s) * C + c];
    }
}

// Column group of a workgroup.  Workgroups are dealt to the 8 XCDs round-robin on their flattened index, so with the
// plain mapping (group = blockIdx.x) the 8 neighbouring column groups of a row -- 8 x 128 B of one W row -- land in 8
// different L2s.  Here each XCD owns a contiguous eighth of the columns instead (1 KB of every W row at M1 = 1024):
// measured 6.92 -> 6.58 ms per step at L = 2^20, every column kernel 8-20 % faster (profiles/r1ab); runs of 2 or 4
// adjacent groups per XCD instead of the whole eighth: 7.00 / 6.92 ms.
__device__ __forceinline__ int col_group(int groups) {
    const int bx = blockIdx.x;
    return (groups & 7) == 0 ? (bx & 7) * (groups >> 3) + (bx >> 3) : bx;
}

template <int M1, int DT>
__global__ void __launch_bounds__(ColCfg<M1>::THREADS, ColCfg<M1>::MIN_WAVES) col_fwd_kernel(ColArgs a) {
    typedef ColCfg<M1> Cfg;
    typedef typename Elem<DT>::type elem_t;
    typedef typename Pair<DT>::raw_t raw_t;
    constexpr int T = Cfg::T, C = Cfg::C, E = Cfg::E;
    HY_SMEM(smem);
    HY_LDS lc32* tlo = HY_LDS_CAST(lc32, smem);
    HY_LDS lc32* thi = tlo + 1024;
    HY_LDS float* plane = HY_LDS_CAST(float, thi + M1);

    const int tid = threadIdx.x;
    const int c = tid % C, r = tid / C;
    const int n2 = col_group(1024 / C) * C + c;
    const int row = blockIdx.y;
    const bool second = blockIdx.z != 0;
    const elem_t* xrow = reinterpret_cast<const elem_t*>(second ? a.x2 : a.x) + (long)(row / a.inner) * a.outer_stride +
                         (long)(row % a.inner) * a.inner_stride;
    c32* Wrow = (second ? a.W2 : a.W) + ((size_t)(row / a.inner) * a.w_bstride + (row % a.inner)) * M1 * 1024;
    const int nfull = a.L >> 1;                         // pairs n < nfull are complete; n == nfull is the odd tail
    // odd L on PITCHED rows (inner_stride > L, round 5): the tail sample comes in with the regular pair loads -- pair nfull = (x[L - 1], the
    // element behind the row's end: inside the row's pitch, loaded and discarded by a select, never part of any arithmetic) -- instead of
    // by a separate predicated scalar load behind them (col_fwd<160, bf16>: 215 vs 204 us at 159999 x 2 / 160000 x 2, profiles/r5v_*)
    const bool tail_pair = (a.L & 1) != 0 && a.inner_stride > (long)a.L;
    const int nload = nfull + (tail_pair ? 1 : 0);

    // all global loads of the thread back to back: E input pairs (+ the table slice)
    // Rows n1 >= M1/2 (s >= E/2) lie beyond L/2 <= M/2 for every supported L: the zero padding is never loaded.
    constexpr int EL = T > 1 ? E / 2 : Cfg::EH;
    raw_t raw[EL];
    if (nload > 0) {
        HY_UNROLL
        for (int s = 0; s < EL; ++s) {
            const int n = (r + T * s) * 1024 + n2;
            raw[s] = load_raw_pair<DT>(xrow, n < nload ? n : 0);
        }
    }
    stage_col_tables<M1, Cfg::THREADS>(tlo, a.tab.tw_lo, tid);
    c32 v[E];
    HY_UNROLL
    for (int s = 0; s < E; ++s) {
        const int n = (r + T * s) * 1024 + n2;
        v[s] = (s < EL && n < nload) ? Pair<DT>::cvt(raw[s < EL ? s : 0]) : mk(0.f, 0.f);
        if (tail_pair && n == nfull) v[s] = mk(v[s].x, 0.f);       // (a select: whatever lies behind the row's end -- NaN in the tests -- is dropped)
    }
    if ((a.L & 1) && !tail_pair) {                      // odd L on packed rows: the last sample is the real part of pair nfull
        HY_UNROLL
        for (int s = 0; s < E; ++s)
            if ((r + T * s) * 1024 + n2 == nfull) v[s] = mk(Elem<DT>::ld(xrow + a.L - 1), 0.f);
    }
    if constexpr (!is_pow2(E)) __syncthreads();        // the mixed-radix butterflies read their twiddles from the tables
    dft_any<E, false>(v, thi, 1);

    if constexpr (T == 1) {
        if constexpr (is_pow2(E)) __syncthreads();     // tables
        HY_UNROLL
        for (int q = 0; q < E; ++q) {
            const c32 w = outer_tw(tlo, thi, n2, q);
            stg(Wrow, (unsigned)(q * 1024 + n2), cmul(v[q], w));
        }
    } else {
        constexpr int NB = Cfg::NB;
        c32 x2[NB * T];
        col_exchange<T, C, NB>(v, x2, plane, c, r);     // its first barrier also covers the table staging
        HY_UNROLL
        for (int i = 0; i < NB; ++i) {
            const int jj = r + T * i;
            if (NB * T != 32 && jj >= 32) break;
            c32 y[T];
            y[0] = x2[i * T];
            HY_UNROLL
            for (int s = 1; s < T; ++s) y[s] = cmul(x2[i * T + s], lds_ld(thi + jj * s));
            dft_any<T, false>(y, thi, 32);
            HY_UNROLL
            for (int q = 0; q < T; ++q) {
                const int k1 = jj + 32 * q;
                const c32 w = outer_tw(tlo, thi, n2, k1);
                stg(Wrow, (unsigned)(k1 * 1024 + n2), cmul(y[q], w));
            }
        }
    }
}

template <int DT>
__device__ __forceinline__ void col_store(typename Elem<DT>::type* xrow, int n, int nfull, int L, c32 v, float* aux0, int row) {
    if (n < nfull) store_raw_pair<DT>(xrow, n, Pair<DT>::pack(v));
    else if (n == nfull && (L & 1)) Elem<DT>::st(xrow + L - 1, v.x);
    if (aux0 != nullptr && n == 0) aux0[row] = v.x;
}

template <int M1, int DT>
// <= 128 VGPRs (4 waves per SIMD): at 141 only ONE 512-thread workgroup fits a CU and the kernel ran at 2.5 TB/s
__global__ void __launch_bounds__(ColCfg<M1>::THREADS, ColCfg<M1>::MIN_WAVES) col_inv_kernel(ColArgs a) {
    typedef ColCfg<M1> Cfg;
    typedef typename Elem<DT>::type elem_t;
    constexpr int T = Cfg::T, C = Cfg::C, E = Cfg::E;
    HY_SMEM(smem);
    HY_LDS lc32* tlo = HY_LDS_CAST(lc32, smem);
    HY_LDS lc32* thi = tlo + 1024;
    HY_LDS float* plane = HY_LDS_CAST(float, thi + M1);

    const int tid = threadIdx.x;
    const int c = tid % C, r = tid / C;
    const int n2 = col_group(1024 / C) * C + c;
    const int row = blockIdx.y;
    elem_t* xrow = reinterpret_cast<elem_t*>(const_cast<void*>(a.x)) + (long)(row / a.inner) * a.outer_stride +
                   (long)(row % a.inner) * a.inner_stride;
    const c32* Wrow = a.W + ((size_t)(row / a.inner) * a.w_bstride + (row % a.inner)) * M1 * 1024;
    const int nfull = a.L >> 1;

    c32 v[E];
    HY_UNROLL
    for (int s = 0; s < E; ++s) v[s] = ldg(Wrow, (unsigned)((r + T * s) * 1024 + n2));   // all loads first
    stage_col_tables<M1, Cfg::THREADS>(tlo, a.tab.tw_lo, tid);
    __syncthreads();
    HY_UNROLL
    for (int s = 0; s < E; ++s) v[s] = cmulc(v[s], outer_tw(tlo, thi, n2, r + T * s));
    dft_any<E, true>(v, thi, 1);

    // Outputs n1 >= M1/2 lie beyond L/2 <= M/2 for every supported L: never stored, so never computed (the
    // compiler drops the butterflies that only feed them).
    if constexpr (T == 1) {
        constexpr int EO = Cfg::EH;
        HY_UNROLL
        for (int q = 0; q < EO; ++q) col_store<DT>(xrow, q * 1024 + n2, nfull, a.L, v[q], a.aux0, row);
    } else {
        constexpr int NB = Cfg::NB;
        c32 x2[NB * T];
        col_exchange<T, C, NB>(v, x2, plane, c, r);
        HY_UNROLL
        for (int i = 0; i < NB; ++i) {
            const int jj = r + T * i;
            if (NB * T != 32 && jj >= 32) break;
            c32 y[T];
            y[0] = x2[i * T];
            HY_UNROLL
            for (int s = 1; s < T; ++s) y[s] = cmulc(x2[i * T + s], lds_ld(thi + jj * s));
            dft_any<T, true>(y, thi, 32);
            // power-of-two T: outputs q >= T/2 are all >= M1/2; T = 5: q = 2 straddles it, col_store drops n >= L/2
            HY_UNROLL
            for (int q = 0; q < Cfg::TO; ++q) col_store<DT>(xrow, (jj + 32 * q) * 1024 + n2, nfull, a.L, y[q], a.aux0, row);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// row kernels.  One workgroup = one wavefront = the row pair (ra, rb) of one (b, channel):
//   SLOT0: ra = 0 (partner of k2 is (1024 - k2) mod 1024 in the same row), rb = M1/2 (partner 1023 - k2,
//          same row; exists only if M1 >= 2)
//   else : ra = slot, rb = M1 - slot, 0 < slot < M1/2 (partner 1023 - k2 in the other row = lane 63 - t,
//          register 31 - q)
// Lane t: half = t / 32 picks the row, j = t % 32; v[q] is element k2 = j + 32 q.
// ---------------------------------------------------------------------------------------------
struct RowArgs {
    c32* X;            // [B][inner][M1][1024]  column-transformed activation rows, transformed in place
    const c32* U;      // row_prod2: filter rows [inner][M1][1024];  row_bwd: u rows [B][inner][M1][1024]
    c32* S;            // row_bwd out: dk rows [inner][M1][1024]
    const float* bias; // [inner] or null
    const c32* K;      // row_bwd: filter rows [inner][M1][1024]
    c32* S0;           // row0_bwd: per-batch-item dk rows of the two self-paired rows, [B][inner][2][1024]
    c32* Y;            // row_prod2: where the product rows go (X itself when in place), [B][y_bstride..][M1][1024]
    int x_bstride;     // rows between batch items of X / U / Y: `inner` in a chunk workspace, D in the saved-spectrum
    int u_bstride;     //   buffer
    int y_bstride;
    Tables tab;
    int M1;
    int inner;
    int B;
    float scale;       // 1 / M
};

enum { MODE_CONV = 0, MODE_CORR = 1 };

// Coefficients of the packed-domain product for one element: Z'[k] = A Z[k] + Bc conj(Z[p]).
// h, hp = H[k], H[partner] (packed filter spectrum); w = w_M^k; bias already scaled.
template <int MODE>
__device__ __forceinline__ c32x2 packed_coeffs(c32 h, c32 hp, c32 w, float bias, float scale) {
    // He = (h + conj(hp))/2 ; Ho = (h - conj(hp))/(2i)
    const c32 He = mk(0.5f * (h.x + hp.x), 0.5f * (h.y - hp.y));
    const c32 Ho = mk(0.5f * (h.y + hp.y), -0.5f * (h.x - hp.x));
    c32 Ke, Ko;
    if (MODE == MODE_CONV) {
        Ke = mk(He.x + bias, He.y);
        Ko = Ho;
    } else {
        Ke = mk(He.x + bias, -He.y);
        Ko = cconj(cmul(w, Ho));
    }
    // A = Ke + (i/2)(1 - w) Ko ; Bc = (i/2)(1 + w) Ko
    const c32 t1 = cmul(mk(1.f - w.x, -w.y), Ko);
    const c32 t2 = cmul(mk(1.f + w.x, w.y), Ko);
    c32x2 r;
    r.a = mk((Ke.x - 0.5f * t1.y) * scale, (Ke.y + 0.5f * t1.x) * scale);
    r.b = mk(-0.5f * t2.y * scale, 0.5f * t2.x * scale);
    return r;
}
template <int MODE>
__device__ __forceinline__ c32 packed_product(c32 x, c32 xp, c32 h, c32 hp, c32 w, float bias) {
    const c32x2 c = packed_coeffs<MODE>(h, hp, w, bias, 1.0f);
    return cadd(cmul(c.a, x), cmulc(c.b, xp));
}

// w_M^k for k = k1 + M1 (j + 32 q):  (w_M^k1 * w_1024^j) * w_32^q, the last factor from a 32-entry constant table
// (a plain array, not a chain of conditionals: the loops calling this must stay under hipcc's unroll size limit;
// the index is a compile-time constant after unrolling, so the loads fold to immediates).
HY_CONST_TABLE float HY_COS32[32] = {1.0f, 0.98078528040323043058f, 0.92387953251128673848f, 0.83146961230254523567f, 0.70710678118654757274f, 0.55557023301960228867f, 0.38268343236508983729f, 0.19509032201612833135f, 0.0f, -0.19509032201612819257f, -0.38268343236508972627f, -0.5555702330196019556f, -0.70710678118654746172f, -0.83146961230254534669f, -0.92387953251128673848f, -0.98078528040323043058f, -1.0f, -0.98078528040323043058f, -0.92387953251128684951f, -0.83146961230254545772f, -0.70710678118654768376f, -0.55557023301960217765f, -0.38268343236509033689f, -0.19509032201612866442f, 0.0f, 0.19509032201612830359f, 0.38268343236509000382f, 0.55557023301960184458f, 0.70710678118654735069f, 0.83146961230254523567f, 0.92387953251128651644f, 0.98078528040323031956f};
HY_CONST_TABLE float HY_SIN32[32] = {0.0f, 0.19509032201612824808f, 0.38268343236508978178f, 0.55557023301960217765f, 0.70710678118654746172f, 0.83146961230254523567f, 0.92387953251128673848f, 0.98078528040323043058f, 1.0f, 0.98078528040323043058f, 0.92387953251128673848f, 0.83146961230254545772f, 0.70710678118654757274f, 0.55557023301960217765f, 0.3826834323650898928f, 0.19509032201612860891f, 0.0f, -0.19509032201612835911f, -0.38268343236508967076f, -0.5555702330196019556f, -0.70710678118654746172f, -0.83146961230254523567f, -0.92387953251128651644f, -0.98078528040323031956f, -1.0f, -0.98078528040323043058f, -0.92387953251128662746f, -0.83146961230254545772f, -0.70710678118654768376f, -0.55557023301960217765f, -0.3826834323650903924f, -0.19509032201612871993f};
__device__ __forceinline__ c32 pw_tw(c32 wkj, int q) {
    HY_OPAQUE(wkj.x); HY_OPAQUE(wkj.y);            // recompute per use instead of keeping 32 hoisted products alive
    const float c = HY_COS32[q], sn = HY_SIN32[q];
    return mk(wkj.x * c + wkj.y * sn, wkj.y * c - wkj.x * sn);
}

// ---------------------------------------------------------------------------------------------
// Two-operand row kernels: both operands arrive column-transformed only; the kernel row-transforms both, forms
// the packed-domain product in its pair form and inverse-transforms.  No spectrum is materialised in memory
// (row_spec + row_conv move 32 more bytes per point per row through the cache hierarchy), so these are the
// path for small batches, where the filter's row transform is not amortised anyway.
//
// Pair form for a real filter/signal pair (E, O = even/odd spectra of x; He, Ho of h; Ke = He + bias; w = w_M^k):
//     conv:  Ye = E Ke + w O Ho               Yo = E Ho + O Ke
//     corr:  Ye = E conj(Ke) + O conj(Ho)     Yo = conj(w) E conj(Ho) + O conj(Ke)
//     Z'[k] = Ye + i Yo                       Z'[partner] = conj(Ye) + i conj(Yo)
// ---------------------------------------------------------------------------------------------
template <int MODE>
__device__ __forceinline__ void prod_pair(c32 x, c32 xp, c32 h, c32 hp, c32 w, float bias, c32& zk, c32& zp) {
    const c32 E = mk(0.5f * (x.x + xp.x), 0.5f * (x.y - xp.y));
    const c32 O = mk(0.5f * (x.y + xp.y), -0.5f * (x.x - xp.x));
    const c32 Ke = mk(0.5f * (h.x + hp.x) + bias, 0.5f * (h.y - hp.y));
    const c32 Ho = mk(0.5f * (h.y + hp.y), -0.5f * (h.x - hp.x));
    c32 Ye, Yo;
    if (MODE == MODE_CONV) {
        Ye = cadd(cmul(E, Ke), cmul(w, cmul(O, Ho)));
        Yo = cadd(cmul(E, Ho), cmul(O, Ke));
    } else {
        Ye = cadd(cmulc(E, Ke), cmulc(O, Ho));
        Yo = cadd(cmulc(cmulc(E, Ho), w), cmulc(O, Ke));
    }
    zk = mk(Ye.x - Yo.y, Ye.y + Yo.x);
    zp = mk(Ye.x + Yo.y, Yo.x - Ye.y);
}

// One pair-form product pass of a general slot.  On entry v = spectrum of X (my 32 registers), h = spectrum of H.
// On exit v = spectrum of the product, scaled.  xl / pl: my / my partner lane's column in the LDS buffer.
template <int MODE>
__device__ __forceinline__ void pair_pass(c32 (&v)[32], const c32 (&h)[32], HY_LDS lc32* xl, const HY_LDS lc32* pl, c32 wkj, float bias,
                                          float scale) {
    HY_UNROLL
    for (int q = 0; q < 16; ++q) {                 // publish the upper register halves
        lds_st(xl + q * 32, v[16 + q]);
        lds_st(xl + 512 + q * 32, h[16 + q]);
    }
    __syncthreads();
    HY_UNROLL
    for (int q = 0; q < 16; ++q) {
        const c32 xp = lds_ld(pl + (15 - q) * 32);          // partner register 31 - q
        const c32 hp = lds_ld(pl + 512 + (15 - q) * 32);
        c32 zk, zp;
        prod_pair<MODE>(v[q], xp, h[q], hp, pw_tw(wkj, q), bias, zk, zp);
        v[q] = cscale(zk, scale);
        v[16 + q] = cscale(zp, scale);             // belongs to the partner's register 31 - q
    }
    __syncthreads();
    HY_UNROLL
    for (int q = 0; q < 16; ++q) lds_st(xl + q * 32, v[16 + q]);
    __syncthreads();
    HY_UNROLL
    for (int q = 0; q < 16; ++q) v[31 - q] = lds_ld(pl + q * 32);
    __syncthreads();
}

// X[b][ch] <- IFFT_row( product( FFT_row(X[b][ch]), FFT_row(H[ch]) ) ), in place, for b = 0 .. B-1, for the
// row pairs (slot, M1 - slot), slot = 1 .. M1/2 - 1.  grid (M1/2 - 1, inner); the filter rows are transformed once
// per workgroup and stay in registers over the batch loop.  (Rows 0 and M1/2: row0_prod2_kernel.)
template <int MODE>
__global__ void __launch_bounds__(64, 2) row_prod2_kernel(RowArgs a) {
    HY_SMEM(smem);
    const int lane = threadIdx.x, half = lane >> 5, j = lane & 31;
    HY_LDS lc32* const lds = HY_LDS_CAST(lc32, smem);
    const int M1 = a.M1;
    const int slot = blockIdx.x + 1, ch = blockIdx.y;
    const unsigned rowbytes = (unsigned)M1 * 1024u * 8u;
    const GBuf H = make_gbuf(a.U + (size_t)ch * M1 * 1024, rowbytes);
    const GBuf twT = make_gbuf(a.tab.tw_rowT, 8192), twR = make_gbuf(a.tab.tw_row, 8192);
    RowTw rtw;
    load_row_tw(rtw, twT, j);
    const c32 tj = gb_ldt(twR, (unsigned)j * 8u, 0u);
    const float bias = (a.bias != nullptr) ? a.bias[ch] : 0.f;

    HY_LDS lc32* xb = lds + half * ROW_LDS;
    HY_LDS lc32* xl = xb + j;
    const HY_LDS lc32* pl = lds + (1 - half) * ROW_LDS + (31 - j);
    HY_OPAQUE(xl);
    HY_OPAQUE(pl);
    const int myrow = half ? M1 - slot : slot;
    const unsigned vo = (unsigned)(myrow * 1024 + j) * 8u;
    c32 h[32], v[32];
    HY_UNROLL
    for (int s = 0; s < 32; ++s) h[s] = gb_ld(H, vo, (unsigned)s * 256u);
    const c32 wkj = cmul(a.tab.tw_lo[myrow], tj);
    row_fft1024<false>(h, xb, j, rtw);
    for (int b = 0; b < a.B; ++b) {
        const GBuf X = make_gbuf(a.X + ((size_t)b * a.x_bstride + ch) * M1 * 1024, rowbytes);
        const GBuf Y = make_gbuf(a.Y + ((size_t)b * a.y_bstride + ch) * M1 * 1024, rowbytes);
        HY_UNROLL
        for (int s = 0; s < 32; ++s) v[s] = gb_ld(X, vo, (unsigned)s * 256u);
        row_fft1024<false>(v, xb, j, rtw);
        pair_pass<MODE>(v, h, xl, pl, wkj, bias, a.scale);
        row_fft1024<true>(v, xb, j, rtw);
        HY_UNROLL
        for (int q = 0; q < 32; ++q) gb_st(Y, vo, (unsigned)q * 256u, v[q]);
    }
}

// Backward row kernel, row pairs (slot, M1 - slot), slot = 1 .. M1/2 - 1.  For b = 0 .. B-1: X[b] = dout rows
// (-> du rows, in place), U[b] = u rows; K = filter rows; dk rows (summed over b in a fixed order: deterministic) go
// to S.  One row transform of dout serves both gradients.  The batch sum is carried in S itself (the row's inverse
// transform is linear, so the partial sums are added after it: one 16 KB read-modify-write per row pair and batch
// item, served by the L2) rather than in 64 accumulator registers.  DO_DU = false skips the du half.
template <bool DO_DU>
__global__ void __launch_bounds__(64, 2) row_bwd_kernel(RowArgs a) {
    HY_SMEM(smem);
    const int lane = threadIdx.x, half = lane >> 5, j = lane & 31;
    HY_LDS lc32* const lds = HY_LDS_CAST(lc32, smem);
    const int M1 = a.M1;
    const int slot = blockIdx.x + 1, ch = blockIdx.y;
    const unsigned rowbytes = (unsigned)M1 * 1024u * 8u;
    const GBuf Kb = make_gbuf(a.K + (size_t)ch * M1 * 1024, rowbytes);
    const GBuf O = make_gbuf(a.S + (size_t)ch * M1 * 1024, rowbytes);
    const GBuf twT = make_gbuf(a.tab.tw_rowT, 8192), twR = make_gbuf(a.tab.tw_row, 8192);
    RowTw rtw;
    load_row_tw(rtw, twT, j);
    const c32 tj = gb_ldt(twR, (unsigned)j * 8u, 0u);
    const float bias = (a.bias != nullptr) ? a.bias[ch] : 0.f;

    HY_LDS lc32* xb = lds + half * ROW_LDS;
    HY_LDS lc32* xl = xb + j;
    const HY_LDS lc32* pl = lds + (1 - half) * ROW_LDS + (31 - j);
    HY_OPAQUE(xl);
    HY_OPAQUE(pl);
    const int myrow = half ? M1 - slot : slot;
    const unsigned vo = (unsigned)(myrow * 1024 + j) * 8u;
    const c32 wkj = cmul(a.tab.tw_lo[myrow], tj);
    for (int b = 0; b < a.B; ++b) {
        const GBuf X = make_gbuf(a.X + ((size_t)b * a.x_bstride + ch) * M1 * 1024, rowbytes);
        const GBuf U = make_gbuf(a.U + ((size_t)b * a.u_bstride + ch) * M1 * 1024, rowbytes);
        c32 h[32], v[32];
        HY_UNROLL
        for (int s = 0; s < 32; ++s) h[s] = gb_ld(U, vo, (unsigned)s * 256u);
        HY_UNROLL
        for (int s = 0; s < 32; ++s) v[s] = gb_ld(X, vo, (unsigned)s * 256u);
        row_fft1024<false>(h, xb, j, rtw);
        row_fft1024<false>(v, xb, j, rtw);
        {   // dk_b = corr(G, U): result into h (its registers are free once consumed), G stays in v
            HY_UNROLL
            for (int q = 0; q < 16; ++q) {
                lds_st(xl + q * 32, v[16 + q]);
                lds_st(xl + 512 + q * 32, h[16 + q]);
            }
            __syncthreads();
            HY_UNROLL
            for (int q = 0; q < 16; ++q) {
                const c32 gp = lds_ld(pl + (15 - q) * 32);
                const c32 up = lds_ld(pl + 512 + (15 - q) * 32);
                c32 zk, zp;
                prod_pair<MODE_CORR>(v[q], gp, h[q], up, pw_tw(wkj, q), 0.f, zk, zp);
                h[q] = cscale(zk, a.scale);
                h[16 + q] = cscale(zp, a.scale);
            }
            __syncthreads();
            HY_UNROLL
            for (int q = 0; q < 16; ++q) lds_st(xl + q * 32, h[16 + q]);
            __syncthreads();
            HY_UNROLL
            for (int q = 0; q < 16; ++q) h[31 - q] = lds_ld(pl + q * 32);
            __syncthreads();
        }
        row_fft1024<true>(h, xb, j, rtw);
        if (b > 0) {
            HY_UNROLL
            for (int q = 0; q < 32; ++q) h[q] = cadd(h[q], gb_ld(O, vo, (unsigned)q * 256u));
        }
        HY_UNROLL
        for (int q = 0; q < 32; ++q) gb_st(O, vo, (unsigned)q * 256u, h[q]);
        if (DO_DU) {   // du_b = corr(G, K) + bias
            HY_UNROLL
            for (int s = 0; s < 32; ++s) h[s] = gb_ld(Kb, vo, (unsigned)s * 256u);
            row_fft1024<false>(h, xb, j, rtw);
            pair_pass<MODE_CORR>(v, h, xl, pl, wkj, bias, a.scale);
            row_fft1024<true>(v, xb, j, rtw);
            HY_UNROLL
            for (int q = 0; q < 32; ++q) gb_st(X, vo, (unsigned)q * 256u, v[q]);
        }
    }
}

// dk rows only, for batches: the batch sum is carried in 64 accumulator registers in the frequency domain and
// inverse-transformed ONCE (row_bwd_kernel re-reads and re-transforms the filter row and read-modify-writes the dk
// rows per batch item -- right for B = 1, where it shares the transform of dout between du and dk; from B = 2 on
// this kernel + row_prod2_kernel<MODE_CORR> for du is faster: 4 instead of 5 row transforms per item, no RMW).
__global__ void __launch_bounds__(64, 2) row_dk_kernel(RowArgs a) {
    HY_SMEM(smem);
    const int lane = threadIdx.x, half = lane >> 5, j = lane & 31;
    HY_LDS lc32* const lds = HY_LDS_CAST(lc32, smem);
    const int M1 = a.M1;
    const int slot = blockIdx.x + 1, ch = blockIdx.y;
    const unsigned rowbytes = (unsigned)M1 * 1024u * 8u;
    const GBuf O = make_gbuf(a.S + (size_t)ch * M1 * 1024, rowbytes);
    const GBuf twT = make_gbuf(a.tab.tw_rowT, 8192), twR = make_gbuf(a.tab.tw_row, 8192);
    RowTw rtw;
    load_row_tw(rtw, twT, j);
    const c32 tj = gb_ldt(twR, (unsigned)j * 8u, 0u);
    HY_LDS lc32* xb = lds + half * ROW_LDS;
    HY_LDS lc32* xl = xb + j;
    const HY_LDS lc32* pl = lds + (1 - half) * ROW_LDS + (31 - j);
    HY_OPAQUE(xl);
    HY_OPAQUE(pl);
    const int myrow = half ? M1 - slot : slot;
    const unsigned vo = (unsigned)(myrow * 1024 + j) * 8u;
    const c32 wkj = cmul(a.tab.tw_lo[myrow], tj);
    c32 acc[32];
    HY_UNROLL
    for (int q = 0; q < 32; ++q) acc[q] = mk(0.f, 0.f);
    for (int b = 0; b < a.B; ++b) {
        const GBuf X = make_gbuf(a.X + ((size_t)b * a.x_bstride + ch) * M1 * 1024, rowbytes);
        const GBuf U = make_gbuf(a.U + ((size_t)b * a.u_bstride + ch) * M1 * 1024, rowbytes);
        c32 h[32], v[32];
        HY_UNROLL
        for (int s = 0; s < 32; ++s) h[s] = gb_ld(U, vo, (unsigned)s * 256u);
        HY_UNROLL
        for (int s = 0; s < 32; ++s) v[s] = gb_ld(X, vo, (unsigned)s * 256u);
        row_fft1024<false>(h, xb, j, rtw);
        row_fft1024<false>(v, xb, j, rtw);
        HY_UNROLL
        for (int q = 0; q < 16; ++q) {
            lds_st(xl + q * 32, v[16 + q]);
            lds_st(xl + 512 + q * 32, h[16 + q]);
        }
        __syncthreads();
        HY_UNROLL
        for (int q = 0; q < 16; ++q) {
            const c32 gp = lds_ld(pl + (15 - q) * 32);
            const c32 up = lds_ld(pl + 512 + (15 - q) * 32);
            c32 zk, zp;
            prod_pair<MODE_CORR>(v[q], gp, h[q], up, pw_tw(wkj, q), 0.f, zk, zp);
            acc[q] = cadd(acc[q], zk);
            acc[16 + q] = cadd(acc[16 + q], zp);
        }
        __syncthreads();
    }
    // partner values belong to the other row of the pair: hand them over, then one inverse transform for the whole batch
    HY_UNROLL
    for (int q = 0; q < 16; ++q) {
        acc[q] = cscale(acc[q], a.scale);
        lds_st(xl + q * 32, cscale(acc[16 + q], a.scale));
    }
    __syncthreads();
    HY_UNROLL
    for (int q = 0; q < 16; ++q) acc[31 - q] = lds_ld(pl + q * 32);
    __syncthreads();
    row_fft1024<true>(acc, xb, j, rtw);
    HY_UNROLL
    for (int q = 0; q < 32; ++q) gb_st(O, vo, (unsigned)q * 256u, acc[q]);
}

// ---------------------------------------------------------------------------------------------
// Rows 0 and M1/2 are their own partners (k2 <-> (1024 - k2) mod 1024, resp. 1023 - k2, inside the row) with an
// irregular lane map, so they take the per-element form of the product, partners read from natural-order LDS
// images.  A workgroup = one row (blockIdx.x: 0 -> row 0, 1 -> row M1/2) of TWO channels, one per half-wave
// (the second half idles on an odd channel count); 2 images x 2 halves = 33.8 KB of LDS.  These rows are 2 of M1,
// so this is a separate small launch and the main kernels above carry no divergent path (and no scratch).
// ---------------------------------------------------------------------------------------------
enum { ROW0_LDS = 4 * ROW_LDS };

template <int MODE>
__device__ __forceinline__ void self_pair_product(c32 (&v)[32], const c32 (&h)[32], HY_LDS lc32* imx, HY_LDS lc32* imh, int j,
                                                  int pk_base, c32 wkj, float bias, float scale) {
    HY_UNROLL
    for (int q = 0; q < 32; ++q) { lds_st(imh + j + 33 * q, h[q]); lds_st(imx + j + 33 * q, v[q]); }
    __syncthreads();
    HY_UNROLL
    for (int q = 0; q < 32; ++q) {
        const int pk2 = (pk_base - (j + 32 * q)) & 1023;
        v[q] = cscale(packed_product<MODE>(v[q], lds_ld(imx + row_idx(pk2)), h[q], lds_ld(imh + row_idx(pk2)), pw_tw(wkj, q), bias), scale);
    }
    __syncthreads();
}

template <int MODE>
__global__ void __launch_bounds__(64, 2) row0_prod2_kernel(RowArgs a) {
    HY_SMEM(smem);
    const int lane = threadIdx.x, half = lane >> 5, j = lane & 31;
    HY_LDS lc32* imx = HY_LDS_CAST(lc32, smem) + half * 2 * ROW_LDS;    // this half-wave's X image / exchange
    HY_LDS lc32* imh = imx + ROW_LDS;                                   // ... H image / exchange
    const int M1 = a.M1;
    const int myrow = blockIdx.x ? (M1 >> 1) : 0;
    const int pk_base = blockIdx.x ? 1023 : 1024;
    const int ch_raw = 2 * blockIdx.y + half;
    const bool valid = ch_raw < a.inner;
    const int ch = valid ? ch_raw : a.inner - 1;       // the idle half mirrors the last channel; its stores are dropped
    // the two half-waves work on different channels: the channel goes into the per-lane offset, the descriptor
    // covers the whole [inner][M1][1024] slab of one batch item (< 4 GB by the host's chunking)
    const unsigned slab = (unsigned)a.inner * (unsigned)M1 * 1024u * 8u;
    const GBuf H = make_gbuf(a.U, slab);
    const GBuf twT = make_gbuf(a.tab.tw_rowT, 8192), twR = make_gbuf(a.tab.tw_row, 8192);
    RowTw rtw;
    load_row_tw(rtw, twT, j);
    const c32 wkj = cmul(a.tab.tw_lo[myrow], gb_ldt(twR, (unsigned)j * 8u, 0u));
    const float bias = (a.bias != nullptr) ? a.bias[ch] : 0.f;
    const unsigned vo = ((unsigned)ch * (unsigned)M1 * 1024u + (unsigned)(myrow * 1024 + j)) * 8u;

    c32 h[32];
    HY_UNROLL
    for (int s = 0; s < 32; ++s) h[s] = gb_ld(H, vo, (unsigned)s * 256u);
    row_fft1024<false>(h, imh, j, rtw);
    {   // one batch item per workgroup (blockIdx.z): these two rows are 2/M1 of the work but, looped over the batch in
        // one wavefront, they were 22 % of the time at L = 32k and all of it at L <= 1k
        const int b = blockIdx.z;
        const GBuf X = make_gbuf(a.X + (size_t)b * a.x_bstride * M1 * 1024, slab);
        const GBuf Y = make_gbuf(a.Y + (size_t)b * a.y_bstride * M1 * 1024, slab);
        c32 v[32];
        HY_UNROLL
        for (int s = 0; s < 32; ++s) v[s] = gb_ld(X, vo, (unsigned)s * 256u);
        row_fft1024<false>(v, imx, j, rtw);
        self_pair_product<MODE>(v, h, imx, imh, j, pk_base, wkj, bias, a.scale);
        row_fft1024<true>(v, imx, j, rtw);
        if (valid) {
            HY_UNROLL
            for (int q = 0; q < 32; ++q) gb_st(Y, vo, (unsigned)q * 256u, v[q]);
        }
    }
}

template <bool DO_DU>
__global__ void __launch_bounds__(64, 2) row0_bwd_kernel(RowArgs a) {
    HY_SMEM(smem);
    const int lane = threadIdx.x, half = lane >> 5, j = lane & 31;
    HY_LDS lc32* imx = HY_LDS_CAST(lc32, smem) + half * 2 * ROW_LDS;    // G image
    HY_LDS lc32* imh = imx + ROW_LDS;                                   // U / K image, exchange
    const int M1 = a.M1;
    const int myrow = blockIdx.x ? (M1 >> 1) : 0;
    const int pk_base = blockIdx.x ? 1023 : 1024;
    const int ch_raw = 2 * blockIdx.y + half;
    const bool valid = ch_raw < a.inner;
    const int ch = valid ? ch_raw : a.inner - 1;
    const unsigned slab = (unsigned)a.inner * (unsigned)M1 * 1024u * 8u;
    const GBuf Kb = make_gbuf(a.K, slab);
    const GBuf twT = make_gbuf(a.tab.tw_rowT, 8192), twR = make_gbuf(a.tab.tw_row, 8192);
    RowTw rtw;
    load_row_tw(rtw, twT, j);
    const c32 wkj = cmul(a.tab.tw_lo[myrow], gb_ldt(twR, (unsigned)j * 8u, 0u));
    const float bias = (a.bias != nullptr) ? a.bias[ch] : 0.f;
    const unsigned vo = ((unsigned)ch * (unsigned)M1 * 1024u + (unsigned)(myrow * 1024 + j)) * 8u;

    {   // one batch item per workgroup; its dk rows go to S0[b] and row0_dk_reduce_kernel adds them up in batch order
        const int b = blockIdx.z;
        const GBuf X = make_gbuf(a.X + (size_t)b * a.x_bstride * M1 * 1024, slab);
        const GBuf U = make_gbuf(a.U + (size_t)b * a.u_bstride * M1 * 1024, slab);
        c32 h[32], v[32];
        HY_UNROLL
        for (int s = 0; s < 32; ++s) h[s] = gb_ld(U, vo, (unsigned)s * 256u);
        HY_UNROLL
        for (int s = 0; s < 32; ++s) v[s] = gb_ld(X, vo, (unsigned)s * 256u);
        row_fft1024<false>(h, imh, j, rtw);
        row_fft1024<false>(v, imx, j, rtw);
        // dk_b = corr(G, U) per element: the result replaces h, G stays in v and in its image
        HY_UNROLL
        for (int q = 0; q < 32; ++q) { lds_st(imh + j + 33 * q, h[q]); lds_st(imx + j + 33 * q, v[q]); }
        __syncthreads();
        HY_UNROLL
        for (int q = 0; q < 32; ++q) {
            const int pk2 = (pk_base - (j + 32 * q)) & 1023;
            h[q] = cscale(packed_product<MODE_CORR>(v[q], lds_ld(imx + row_idx(pk2)), h[q], lds_ld(imh + row_idx(pk2)), pw_tw(wkj, q), 0.f),
                          a.scale);
            if (DO_DU && (q & 3) == 3) HY_SCHED_FENCE();         // G stays live for du: four products at a time, not 32 partner reads + twiddles up front
        }
        __syncthreads();
        row_fft1024<true>(h, imh, j, rtw);
        if (valid) {
            c32* P = a.S0 + ((((size_t)b * a.inner + ch) * 2 + blockIdx.x) * 1024) + j;
            HY_UNROLL
            for (int q = 0; q < 32; ++q) P[32 * q] = h[q];
        }
        if (DO_DU) {
            HY_UNROLL
            for (int s = 0; s < 32; ++s) h[s] = gb_ld(Kb, vo, (unsigned)s * 256u);
            row_fft1024<false>(h, imh, j, rtw);
            HY_UNROLL
            for (int q = 0; q < 32; ++q) lds_st(imh + j + 33 * q, h[q]);       // K image; the G image is still in imx
            __syncthreads();
            // the partner indices and the 32 twiddles are recomputed for this second product (from copies the optimiser cannot identify
            // with the first product's): kept live across the three transforms in between they were 81 spilled registers
            int j2 = j;
            c32 wk2 = wkj;
            HY_OPAQUE(j2);
            HY_OPAQUE(wk2.x);
            HY_OPAQUE(wk2.y);
            // ... and this lane's own G values come back from the G image (they were 64 more registers live across those transforms)
            HY_UNROLL
            for (int q = 0; q < 32; ++q) {
                const int pk2 = (pk_base - (j2 + 32 * q)) & 1023;
                v[q] = cscale(packed_product<MODE_CORR>(lds_ld(imx + j2 + 33 * q), lds_ld(imx + row_idx(pk2)), h[q], lds_ld(imh + row_idx(pk2)),
                                                        pw_tw(wk2, q), bias),
                              a.scale);
                if ((q & 3) == 3) HY_SCHED_FENCE();
            }
            __syncthreads();
            row_fft1024<true>(v, imx, j, rtw);
            if (valid) {
                HY_UNROLL
                for (int q = 0; q < 32; ++q) gb_st(X, vo, (unsigned)q * 256u, v[q]);
            }
        }
    }
}

// dk rows 0 and M1/2: sum of the per-batch-item rows written by row0_bwd_kernel, in batch order (deterministic).
// grid (rows01, inner), 256 threads.
__global__ void __launch_bounds__(256) row0_dk_reduce_kernel(RowArgs a) {
    const int ch = blockIdx.y, r01 = blockIdx.x;
    const int myrow = r01 ? (a.M1 >> 1) : 0;
    c32* O = a.S + ((size_t)ch * a.M1 + myrow) * 1024;
    for (int e = threadIdx.x; e < 1024; e += 256) {
        c32 acc = mk(0.f, 0.f);
        for (int b = 0; b < a.B; ++b) acc = cadd(acc, a.S0[(((size_t)b * a.inner + ch) * 2 + r01) * 1024 + e]);
        O[e] = acc;
    }
}

#endif  // HY_HELPERS_ONLY

}  // namespace hyena
