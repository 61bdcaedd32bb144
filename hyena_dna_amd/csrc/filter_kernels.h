// filter_kernels.h -- the implicit Hyena filter (HyenaFilter.filter, src/models/sequence/hyena.py:229-238) and its
// gradients as fused fp32-MFMA kernels.
//
// Reference arithmetic per position l (hyena.py:109-131 positional embedding z_l, 199-215 the sine MLP, 96-106 Sin,
// 134-155 ExponentialModulation), HyenaDNA configuration (emb_dim 5, filter_order 64, two inner layers, order 2):
//     a0 = W0 z_l + b0            h0 = sin(f * a0)          W0 (64, E)
//     a1 = W1 h0  + b1            h1 = sin(f * a1)          W1 (64, 64)
//     a2 = W2 h1  + b2            h2 = sin(f * a2)          W2 (64, 64)      f (64,) shared by the three activations
//     y  = W3 h2                                             W3 (D, 64), no bias
//     k[d, l] = y[d] * (exp(-t_l |delta_d|) + shift)
// The reference runs this as 4 GEMMs + 3 (mul, sin) pairs + exp/mul and a transposing copy, ~25 kernels forward and
// backward, 12 of the 30 ms of a HyenaDNA layer at L = 2^20 (profiles/r1m).  Here:
//
//   filter_fwd_kernel      one pass, positions -> k (D, L) directly in the layout the long convolution reads.
//                          A wavefront owns 32 positions and carries the whole chain in registers: every layer is a
//                          v_mfma_f32_32x32x2_f32 GEMM  C[feature][position] += W[feature][k] * h[k][position]  whose
//                          C/D register layout (lane = position, registers = features 8(r/4) + 4(lane/32) + r%4) is
//                          fed back as the B operand of the next layer unchanged -- the contraction index is simply
//                          enumerated in that order, and the weights (A operand, from LDS) are read in the same
//                          order.  No LDS traffic for activations, no shuffles.  fp32 in, fp32 accumulate.
//   filter_layer_bwd_kernel<NO, NI, MODE>   one layer of the backward pass per launch (W3, W2, W1, W0):
//                          dh = W^T delta_out (contraction over features, same register trick), the activation's
//                          derivative from the SAVED pre-activation, and dW += delta_out h^T (contraction over
//                          POSITIONS: operands with the feature on the lane axis, delta read from global, h staged
//                          through LDS), bias / frequency gradients on the side.  Accumulators stay in registers
//                          over a persistent loop; per-workgroup partials are summed by filter_reduce_kernel in a
//                          fixed order (deterministic, no atomics).
//
// The forward saves the three pre-activations a0, a1, a2 (3 x 64 x L fp32) when gradients are needed -- less than the
// reference's autograd keeps (z W^T, f*a, sin(...) per layer, some twice under autocast).
//
// Precision: fp32 throughout (v_mfma_f32_32x32x2_f32 is exact fp32 FMA arithmetic) -- the graph the reference computes without
// autocast; it agrees with the fp32 reference to ~1e-6 (tests/test_filter_*.py state the tolerances).  Under torch.autocast the
// reference runs the four GEMMs in the 16-bit type: that graph is filter16_kernels.h's (this header's first-layer backward kernel
// serves it too, with its operands rounded on load: FilterBwdArgs::rdt).
#pragma once
#include "fftconv_kernels.h"

namespace hyena {

#ifdef HIPEMU
typedef hipemu::floatx16 f32x16;
#define HY_MFMA(a, b, c) hipemu::mfma_f32_32x32x2f32((a), (b), (c))
__device__ __forceinline__ float hy_exp2(float x) { return exp2f(x); }
#else
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define HY_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)
__device__ __forceinline__ float hy_exp2(float x) { return __builtin_amdgcn_exp2f(x); }
#endif

// fp32 buffer-addressed global access (see GBuf in fftconv_kernels.h): descriptor base + per-lane byte offset
// (+ wave-uniform byte offset for loads).  The hardware range check covers the per-lane offset only: lanes whose
// position is past the end pass FLT_OOB and their loads return 0 / their stores are dropped, with no branches.
#define FLT_OOB 0x80000000u
#ifdef HIPEMU
struct FBuf { char* p; unsigned n; };
__device__ __forceinline__ FBuf make_fbuf(const void* base, size_t bytes) { FBuf b; b.p = (char*)base; b.n = (unsigned)bytes; return b; }
__device__ __forceinline__ float fb_ld(FBuf b, unsigned voff, unsigned soff) {
    if ((size_t)voff + 4 > b.n) return 0.f;
    return *reinterpret_cast<const float*>(b.p + voff + soff);
}
__device__ __forceinline__ void fb_st(FBuf b, unsigned voff, float v) {
    if ((size_t)voff + 4 <= b.n) *reinterpret_cast<float*>(b.p + voff) = v;
}
__device__ __forceinline__ void fb_ld4(FBuf b, unsigned voff, float* v) {       // fully in range by contract
    if ((size_t)voff + 16 > b.n) abort();
    memcpy(v, b.p + voff, 16);
}
#else
struct FBuf { __amdgpu_buffer_rsrc_t r; };
__device__ __forceinline__ FBuf make_fbuf(const void* base, size_t bytes) {
    FBuf b;
    b.r = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr(const_cast<void*>(base)), 0, (unsigned)bytes, 0x00020000);
    return b;
}
__device__ __forceinline__ float fb_ld(FBuf b, unsigned voff, unsigned soff) {
    return u2f(__builtin_amdgcn_raw_buffer_load_b32(b.r, voff, soff, 0));
}
__device__ __forceinline__ void fb_st(FBuf b, unsigned voff, float v) {          // no scalar offset on stores (gb_st)
    __builtin_amdgcn_raw_buffer_store_b32(f2u(v), b.r, voff, 0, 0);
}
__device__ __forceinline__ void fb_ld4(FBuf b, unsigned voff, float* v) {
    const hy_u4 w = __builtin_amdgcn_raw_buffer_load_b128(b.r, voff, 0, 0);
    v[0] = u2f(w.x); v[1] = u2f(w.y); v[2] = u2f(w.z); v[3] = u2f(w.w);
}
#endif

// sin and cos of an fp32 argument: three-constant Cody-Waite reduction by pi/2 (fused multiply-adds, so the partial
// products are exact) + the classic degree-7 / degree-8 minimax polynomials on [-pi/4, pi/4].  Branch-free, ~1 ulp for
// |x| < 1e5; beyond that the argument f * a itself (one fp32 rounding) no longer determines the phase, so the slow
// exact-reduction path of libm's sinf would add registers, not information.
__device__ __forceinline__ void hy_sincos(float x, float* sn, float* cs) {
    const float n = rintf(x * 0.636619772367581343f);
    float r = fmaf(n, -1.57079601e+00f, x);
    r = fmaf(n, -3.13916473e-07f, r);
    r = fmaf(n, -5.39030253e-15f, r);
    const float r2 = r * r;
    const float ps = fmaf(fmaf(fmaf(-1.9515295891e-4f, r2, 8.3321608736e-3f), r2, -1.6666654611e-1f) * r2, r, r);
    const float pc = fmaf(fmaf(fmaf(2.443315711809948e-5f, r2, -1.388731625493765e-3f), r2, 4.166664568298827e-2f) * r2, r2,
                          fmaf(-0.5f, r2, 1.f));
    const int q = (int)n;
    const bool swap = (q & 1) != 0;
    const float s0 = swap ? pc : ps, c0 = swap ? ps : pc;
    *sn = (q & 2) ? -s0 : s0;
    *cs = ((q + 1) & 2) ? -c0 : c0;
}

enum {
    FLT_O = 64,                       // width of the sine MLP (filter_order) the kernels are built for
    FLT_E = 8,                        // embedding width after zero padding (emb_dim <= 8)
    FLT_WAVES = 8,
    FLT_THREADS = FLT_WAVES * 64,
    FLT_TP = 32,                      // positions per wavefront tile (one MFMA column block)
    FLT_WG_POS = FLT_WAVES * FLT_TP,  // positions per workgroup iteration of the backward kernels
    FLT_HS = FLT_WG_POS + 4,          // LDS row stride of the staged activations (floats; 16-byte aligned rows)
    FLT_MAX_WG = 256                  // persistent grid: one workgroup per CU
};

#define FLT_LOG2E 1.4426950408889634f

// row of the C/D operand held in register r of a lane in half-wave `half` (v_mfma_f32_32x32x2_f32)
__device__ __forceinline__ constexpr int crow(int r, int half) { return (r & 3) + 8 * (r >> 2) + 4 * half; }

struct FilterArgs {
    const float* z;       // (L, zs) positional embedding rows (first E columns used)
    const float* t;       // (L,)
    const float* w0;      // (64, E)
    const float* b0;      // (64,)
    const float* w1;      // (64, 64)
    const float* b1;
    const float* w2;
    const float* b2;
    const float* w3;      // (D, 64)
    const float* freq;    // (64,)
    const float* deltas;  // (D,)
    float* k;             // (D, L) out
    float* acts;          // (3, 64, L) pre-activations out, or nullptr
    float shift;
    int modulate;
    int L, E, zs, D;
    int ldk, lds;         // row pitch (floats, >= L) of k and of the saved pre-activations (packed: L)
};

// ---------------------------------------------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------------------------------------------
template <int D>
struct FltFwdLds {
    static constexpr int W0 = 0;                          // [64][9]
    static constexpr int W1 = W0 + FLT_O * (FLT_E + 1);   // [64][65]
    static constexpr int W2 = W1 + FLT_O * (FLT_O + 1);
    static constexpr int W3 = W2 + FLT_O * (FLT_O + 1);   // [D][65]
    static constexpr int CST = W3 + D * (FLT_O + 1);      // b0 | b1 | b2 | freq | |delta| log2(e)
    static constexpr int FLOATS = CST + 4 * FLT_O + D;
    static constexpr size_t BYTES = FLOATS * sizeof(float);
};

// one hidden layer on a 32-position tile: out[ob] = bias + W[32 ob .. 32 ob + 31][:] * in, then (optionally saved
// and) passed through sin(f .)
template <int KB>   // KB: 32-feature blocks of the input
__device__ __forceinline__ void flt_layer(const HY_LDS float* Ws, int ws, const HY_LDS float* bias, const f32x16 (&in)[KB],
                                          f32x16 (&out)[2], int n, int half) {
    HY_UNROLL
    for (int ob = 0; ob < 2; ++ob) {
        HY_UNROLL
        for (int r = 0; r < 16; ++r) out[ob][r] = bias[32 * ob + crow(r, half)];
    }
    HY_UNROLL
    for (int cb = 0; cb < KB; ++cb) {
        HY_UNROLL
        for (int r = 0; r < 16; ++r) {
            const int fin = 32 * cb + crow(r, half);
            HY_UNROLL
            for (int ob = 0; ob < 2; ++ob) out[ob] = HY_MFMA(Ws[(32 * ob + n) * ws + fin], in[cb][r], out[ob]);
        }
    }
}

template <bool SAVE>
__device__ __forceinline__ void flt_act(f32x16 (&x)[2], const HY_LDS float* freq, FBuf save, unsigned voff, unsigned L4, int half) {
    HY_UNROLL
    for (int cb = 0; cb < 2; ++cb) {
        HY_UNROLL
        for (int r = 0; r < 16; ++r) {
            const int f = 32 * cb + crow(r, half);
            const float a = x[cb][r];
            if (SAVE) fb_st(save, voff + (unsigned)f * L4, a);
            float sn, cs;
            hy_sincos(freq[f] * a, &sn, &cs);
            x[cb][r] = sn;
        }
    }
}

template <int D, bool SAVE>
__global__ void __launch_bounds__(FLT_THREADS, 2) filter_fwd_kernel(FilterArgs a) {
    typedef FltFwdLds<D> Lds;
    HY_SMEM(smem);
    HY_LDS float* sm = HY_LDS_CAST(float, smem);
    const int tid = threadIdx.x;
    for (int i = tid; i < FLT_O * (FLT_E + 1); i += FLT_THREADS) {
        const int o = i / (FLT_E + 1), e = i % (FLT_E + 1);
        sm[Lds::W0 + i] = e < a.E ? a.w0[o * a.E + e] : 0.f;
    }
    for (int i = tid; i < FLT_O * FLT_O; i += FLT_THREADS) {
        sm[Lds::W1 + (i >> 6) * (FLT_O + 1) + (i & 63)] = a.w1[i];
        sm[Lds::W2 + (i >> 6) * (FLT_O + 1) + (i & 63)] = a.w2[i];
    }
    for (int i = tid; i < D * FLT_O; i += FLT_THREADS) sm[Lds::W3 + (i >> 6) * (FLT_O + 1) + (i & 63)] = a.w3[i];
    for (int i = tid; i < FLT_O; i += FLT_THREADS) {
        sm[Lds::CST + i] = a.b0[i];
        sm[Lds::CST + FLT_O + i] = a.b1[i];
        sm[Lds::CST + 2 * FLT_O + i] = a.b2[i];
        sm[Lds::CST + 3 * FLT_O + i] = a.freq[i];
    }
    for (int i = tid; i < D; i += FLT_THREADS) sm[Lds::CST + 4 * FLT_O + i] = a.modulate ? fabsf(a.deltas[i]) * FLT_LOG2E : 0.f;
    __syncthreads();

    const int lane = tid & 63, wave = tid >> 6, half = lane >> 5, n = lane & 31;
    const int ntiles = (a.L + FLT_TP - 1) / FLT_TP;
    const HY_LDS float* freq = sm + Lds::CST + 3 * FLT_O;
    const HY_LDS float* cdec = sm + Lds::CST + 4 * FLT_O;
    const unsigned L4 = (unsigned)a.L * 4u, K4 = (unsigned)a.ldk * 4u, S4 = (unsigned)a.lds * 4u;   // bytes per row: t, k, saved
    const FBuf Kb = make_fbuf(a.k, (size_t)D * K4);
    const FBuf Ab = make_fbuf(a.acts, SAVE ? (size_t)3 * FLT_O * S4 : 0);
    const FBuf Zb = make_fbuf(a.z, (size_t)a.L * a.zs * 4u);
    const FBuf Tb = make_fbuf(a.t, L4);
    for (int tile = blockIdx.x * FLT_WAVES + wave; tile < ntiles; tile += gridDim.x * FLT_WAVES) {
        const int pos = tile * FLT_TP + n;
        const bool valid = pos < a.L;
        const unsigned vpos = valid ? (unsigned)pos * 4u : FLT_OOB;
        // layer 0: the contraction runs over the (zero-padded) embedding, k = 2 s + half
        f32x16 h[2], g[2];
        HY_UNROLL
        for (int ob = 0; ob < 2; ++ob) {
            HY_UNROLL
            for (int r = 0; r < 16; ++r) h[ob][r] = sm[Lds::CST + 32 * ob + crow(r, half)];
        }
        HY_UNROLL
        for (int s = 0; s < FLT_E / 2; ++s) {
            const int e = 2 * s + half;
            const float zv = fb_ld(Zb, valid && e < a.E ? (unsigned)(pos * a.zs + e) * 4u : FLT_OOB, 0);
            HY_UNROLL
            for (int ob = 0; ob < 2; ++ob) h[ob] = HY_MFMA(sm[Lds::W0 + (32 * ob + n) * (FLT_E + 1) + e], zv, h[ob]);
        }
        HY_SCHED_FENCE();
        flt_act<SAVE>(h, freq, Ab, vpos, S4, half);
        HY_SCHED_FENCE();
        flt_layer<2>(sm + Lds::W1, FLT_O + 1, sm + Lds::CST + FLT_O, h, g, n, half);
        HY_SCHED_FENCE();
        flt_act<SAVE>(g, freq, Ab, vpos + FLT_O * S4, S4, half);
        HY_SCHED_FENCE();
        flt_layer<2>(sm + Lds::W2, FLT_O + 1, sm + Lds::CST + 2 * FLT_O, g, h, n, half);
        HY_SCHED_FENCE();
        flt_act<SAVE>(h, freq, Ab, vpos + 2 * FLT_O * S4, S4, half);
        HY_SCHED_FENCE();
        // last layer + modulation, 32 output channels at a time
        const float tl = fb_ld(Tb, vpos, 0);
        for (int db = 0; db < D / 32; ++db) {
            f32x16 y;
            HY_UNROLL
            for (int r = 0; r < 16; ++r) y[r] = 0.f;
            HY_UNROLL
            for (int cb = 0; cb < 2; ++cb) {
                HY_UNROLL
                for (int r = 0; r < 16; ++r)
                    y = HY_MFMA(sm[Lds::W3 + (32 * db + n) * (FLT_O + 1) + 32 * cb + crow(r, half)], h[cb][r], y);
            }
            HY_UNROLL
            for (int r = 0; r < 16; ++r) {
                const int d = 32 * db + crow(r, half);
                const float m = a.modulate ? hy_exp2(-tl * cdec[d]) + a.shift : 1.f;
                fb_st(Kb, vpos + (unsigned)d * K4, y[r] * m);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// backward, one layer per launch
// ---------------------------------------------------------------------------------------------------------------
enum { FLT_ACT = 1, FLT_MOD = 2 };

struct FilterBwdArgs {
    const float* dout;     // (NO, L): gradient w.r.t. this layer's output (FLT_MOD: dk, modulation applied on load)
    const float* w;        // (NO, ni) this layer's weight, ni = 64 or E
    const float* aprev;    // FLT_ACT: saved pre-activation of the previous layer (64, L);  else z (L, zs)
    const float* freq;     // (64,)
    const float* t;        // FLT_MOD (L,)
    const float* deltas;   // FLT_MOD (NO,)
    float* dprev;          // out: gradient w.r.t. the previous layer's pre-activation (64, L) / w.r.t. z (E, L); may be null
    float* part_w;         // out: [slots][NO][NI] partial weight gradients
    float* part_b;         // out: [2 slots][NO] partial bias gradients, or nullptr (the last layer has no bias)
    float* part_f;         // out: [gridDim.x * 8][64] partial frequency gradients (FLT_ACT)
    float shift;
    int modulate;
    int L, ni, zs;
    int ldo, lda, ldp;     // row pitch (floats, >= L) of dout, of aprev (FLT_ACT) and of dprev (packed: L)
    int rdt;               // DT_BF16 / DT_F16: the 16-bit path's first layer (filter16_kernels.h) -- the weight and the embedding rows are
                           // rounded to that type on load, dz on store (what the reference's autocast Linear sees); 0 = plain fp32
};

// value of v after a round trip through the 16-bit type rdt (0: unchanged)
__device__ __forceinline__ float flt_rnd(float v, int rdt) {
    if (rdt == DT_BF16) return bf16_to_f32(f32_to_bf16(v));
    if (rdt == DT_F16) return f16_to_f32(f32_to_f16(v));
    return v;
}

template <int NO, int NI>
struct FltBwdCfg {
    static constexpr int NIB = (NI + 31) / 32;                      // 32-row blocks of the previous layer's features
    static constexpr int WS = NI + 1;                               // LDS row stride of the weight
    static constexpr int BLOCKS = (NO / 32) * NIB;                  // 32x32 blocks of dW
    static constexpr int BPW = BLOCKS >= FLT_WAVES ? BLOCKS / FLT_WAVES : 1;   // blocks per wavefront
    static constexpr int GROUPS = BLOCKS / BPW;                     // distinct block assignments
    static constexpr int KS = FLT_WAVES / GROUPS;                   // wavefronts sharing a block split the positions
    static constexpr int LDS_W = 0;
    static constexpr int LDS_H = LDS_W + NO * WS;                   // [NI][FLT_HS]
    static constexpr int LDS_C = LDS_H + NI * FLT_HS;               // freq[64] | cdec[NO] | t[FLT_WG_POS]
    static constexpr int FLOATS = LDS_C + FLT_O + NO + FLT_WG_POS;
    static constexpr size_t BYTES = FLOATS * sizeof(float);
    static_assert(NO % 64 == 0 && (NI == FLT_O || NI == FLT_E), "layer shapes of the HyenaDNA filter MLP");
    static_assert(BPW == 1 || (BPW == 2 && NIB == 2), "a wavefront's two blocks must share a row block");
    static_assert(NI < 32 || FLT_WAVES * FLT_O * FLT_TP <= NI * FLT_HS, "H doubles as reduction scratch");
};

// 16 consecutive floats of row `row` of a (rows, L) buffer with row pitch ld, starting at pos0 (a multiple of 16); positions >= L read as 0
__device__ __forceinline__ void flt_load16(FBuf b, int row, int pos0, int L, int ld, float (&v)[16]) {
    const unsigned base = ((unsigned)row * (unsigned)ld + (unsigned)pos0) * 4u;
    if ((ld & 3) == 0 && pos0 + 16 <= L) {
        HY_UNROLL
        for (int j = 0; j < 4; ++j) fb_ld4(b, base + 16u * j, &v[4 * j]);
    } else {
        HY_UNROLL
        for (int j = 0; j < 16; ++j) v[j] = fb_ld(b, pos0 + j < L ? base + 4u * j : FLT_OOB, 0);
    }
}

template <int NO, int NI, int MODE>
__global__ void __launch_bounds__(FLT_THREADS, 2) filter_layer_bwd_kernel(FilterBwdArgs a) {
    typedef FltBwdCfg<NO, NI> Cfg;
    constexpr bool ACT = (MODE & FLT_ACT) != 0, MOD = (MODE & FLT_MOD) != 0;
    static_assert(ACT == (NI == FLT_O), "only the first layer has no activation in front of it");
    HY_SMEM(smem);
    HY_LDS float* sm = HY_LDS_CAST(float, smem);
    HY_LDS float* Ws = sm + Cfg::LDS_W;
    HY_LDS float* Hs = sm + Cfg::LDS_H;
    HY_LDS float* freq = sm + Cfg::LDS_C;
    HY_LDS float* cdec = freq + FLT_O;
    HY_LDS float* Tt = cdec + NO;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, n = lane & 31;
    const int L = a.L;
    const bool modulate = MOD && a.modulate != 0;
    const float shift = a.shift;

    for (int i = tid; i < NO * NI; i += FLT_THREADS) {
        const int o = i / NI, c = i % NI;
        Ws[o * Cfg::WS + c] = flt_rnd(c < a.ni ? a.w[o * a.ni + c] : 0.f, a.rdt);
    }
    for (int i = tid; i < FLT_O; i += FLT_THREADS) freq[i] = ACT ? a.freq[i] : 0.f;
    for (int i = tid; i < NO; i += FLT_THREADS) cdec[i] = modulate ? fabsf(a.deltas[i]) * FLT_LOG2E : 0.f;
    __syncthreads();

    // this wavefront's share of dW: blocks (rb, cb0 .. cb0 + BPW - 1), positions of sub-range ks
    const int group = wave % Cfg::GROUPS, ks = wave / Cfg::GROUPS;
    const int blk0 = group * Cfg::BPW;
    const int rb = blk0 / Cfg::NIB, cb0 = blk0 % Cfg::NIB;
    // lane's row of the previous layer's features in feature-on-lane operands (NI = 8: lanes >= 8 carry zeros)
    const bool ivalid = NI >= 32 || n < NI;
    f32x16 accw[Cfg::BPW];
    HY_UNROLL
    for (int q = 0; q < Cfg::BPW; ++q) {
        HY_UNROLL
        for (int r = 0; r < 16; ++r) accw[q][r] = 0.f;
    }
    float accb = 0.f;
    f32x16 accf[Cfg::NIB];
    HY_UNROLL
    for (int q = 0; q < Cfg::NIB; ++q) {
        HY_UNROLL
        for (int r = 0; r < 16; ++r) accf[q][r] = 0.f;
    }

    const unsigned L4 = (unsigned)L * 4u, O4 = (unsigned)a.ldo * 4u, A4 = (unsigned)a.lda * 4u, P4 = (unsigned)a.ldp * 4u;
    const FBuf Db = make_fbuf(a.dout, (size_t)NO * O4);
    const FBuf Ab = make_fbuf(a.aprev, ACT ? (size_t)FLT_O * A4 : (size_t)L * a.zs * 4u);
    const FBuf Pb = make_fbuf(a.dprev, a.dprev == nullptr ? 0 : (ACT ? (size_t)FLT_O * P4 : (size_t)a.ni * P4));
    const FBuf Tb = make_fbuf(a.t, MOD ? L4 : 0);
    const int niter = (L + FLT_WG_POS - 1) / FLT_WG_POS;
    for (int it = blockIdx.x; it < niter; it += gridDim.x) {
        const int p0 = it * FLT_WG_POS;
        if (MOD && tid < FLT_WG_POS) Tt[tid] = fb_ld(Tb, p0 + tid < L ? (unsigned)(p0 + tid) * 4u : FLT_OOB, 0);
        // ---- contraction over features for this wavefront's 32 positions: dh = W^T delta
        {
            const int pos = p0 + FLT_TP * wave + n;
            const bool valid = pos < L;
            const unsigned vpos = valid ? (unsigned)pos * 4u : FLT_OOB;
            const float tl = MOD ? fb_ld(Tb, vpos, 0) : 0.f;
            f32x16 dh[Cfg::NIB];
            HY_UNROLL
            for (int q = 0; q < Cfg::NIB; ++q) {
                HY_UNROLL
                for (int r = 0; r < 16; ++r) dh[q][r] = 0.f;
            }
            // the saved pre-activations of this tile: all loads issued now, they land while the MFMAs below run (loaded one
            // by one next to their use they serialise behind the stores: vmcnt counts both)
            f32x16 apv[ACT ? Cfg::NIB : 1];
            if (ACT) {
                HY_UNROLL
                for (int q = 0; q < Cfg::NIB; ++q) {
                    HY_UNROLL
                    for (int r = 0; r < 16; ++r)
                        apv[q][r] = fb_ld(Ab, vpos + (unsigned)(4 * half) * A4, (unsigned)(32 * q + crow(r, 0)) * A4);
                }
            }
            HY_SCHED_FENCE();
            for (int s0 = 0; s0 < NO / 2; s0 += 16) {
                float dv[16];
                HY_UNROLL
                for (int j = 0; j < 16; ++j) {
                    const int o = s0 + j + (NO / 2) * half;
                    float v = fb_ld(Db, vpos + (unsigned)((NO / 2) * half) * O4, (unsigned)(s0 + j) * O4);
                    if (MOD) v *= modulate ? hy_exp2(-tl * cdec[o]) + shift : 1.f;
                    dv[j] = v;
                }
                HY_UNROLL
                for (int j = 0; j < 16; ++j) {
                    const int o = s0 + j + (NO / 2) * half;
                    HY_UNROLL
                    for (int q = 0; q < Cfg::NIB; ++q) {
                        const float wv = Ws[o * Cfg::WS + (NI >= 32 ? 32 * q + n : (ivalid ? n : 0))];
                        dh[q] = HY_MFMA(ivalid ? wv : 0.f, dv[j], dh[q]);
                    }
                }
            }
            if (ACT) {
                HY_UNROLL
                for (int q = 0; q < Cfg::NIB; ++q) {
                    HY_UNROLL
                    for (int r = 0; r < 16; ++r) {
                        const int f = 32 * q + crow(r, half);
                        const float ap = apv[q][r];
                        const float fr = freq[f];
                        float sn, cs;
                        hy_sincos(fr * ap, &sn, &cs);
                        const float gg = dh[q][r] * cs;
                        accf[q][r] += gg * ap;
                        fb_st(Pb, vpos + (unsigned)f * P4, gg * fr);
                        Hs[f * FLT_HS + FLT_TP * wave + n] = sn;
                    }
                }
            } else {
                // the layer below is the embedding itself: rows 0 .. E-1 of dh are dz, and h = z
                HY_UNROLL
                for (int r = 0; r < 4; ++r) {
                    const int e = crow(r, half);        // 0 .. 7
                    const bool ev = e < a.ni;
                    fb_st(Pb, ev ? vpos + (unsigned)e * P4 : FLT_OOB, flt_rnd(dh[0][r], a.rdt));
                    Hs[e * FLT_HS + FLT_TP * wave + n] = flt_rnd(fb_ld(Ab, valid && ev ? (unsigned)(pos * a.zs + e) * 4u : FLT_OOB, 0), a.rdt);
                }
            }
        }
        __syncthreads();
        // ---- contraction over positions: dW[rb, cb] += delta[rows of rb][positions] * h[rows of cb][positions]^T
        {
            constexpr int PT_PER = FLT_WAVES / Cfg::KS;
            const int o = 32 * rb + n;
            for (int pt = ks * PT_PER; pt < (ks + 1) * PT_PER; ++pt) {
                const int q0 = FLT_TP * pt + 16 * half;          // first of this lane's 16 positions within the tile
                float av[16];
                flt_load16(Db, o, p0 + q0, L, a.ldo, av);
                if (MOD) {
                    const float cd = cdec[o];
                    HY_UNROLL
                    for (int j = 0; j < 16; ++j) av[j] *= modulate ? hy_exp2(-Tt[q0 + j] * cd) + shift : 1.f;
                }
                if (cb0 == 0) {
                    HY_UNROLL
                    for (int j = 0; j < 16; ++j) accb += av[j];
                }
                HY_UNROLL
                for (int q = 0; q < Cfg::BPW; ++q) {
                    const int hrow = NI >= 32 ? 32 * (cb0 + q) + n : (ivalid ? n : 0);
                    float bv[16];
                    HY_UNROLL
                    for (int j = 0; j < 16; ++j) bv[j] = Hs[hrow * FLT_HS + q0 + j];
                    HY_UNROLL
                    for (int j = 0; j < 16; ++j) accw[q] = HY_MFMA(av[j], ivalid ? bv[j] : 0.f, accw[q]);
                }
            }
        }
        __syncthreads();
    }

    // ---- partial results of this workgroup
    const int slot = blockIdx.x * Cfg::KS + ks;
    HY_UNROLL
    for (int q = 0; q < Cfg::BPW; ++q) {
        HY_UNROLL
        for (int r = 0; r < 16; ++r) {
            const int o = 32 * rb + crow(r, half);
            const int c = NI >= 32 ? 32 * (cb0 + q) + n : n;
            if (ivalid) a.part_w[((size_t)slot * NO + o) * NI + c] = accw[q][r];
        }
    }
    if (a.part_b != nullptr && cb0 == 0) a.part_b[(size_t)(slot * 2 + half) * NO + 32 * rb + n] = accb;
    if (ACT) {
        // frequency gradient: sum the position-on-lane accumulators over the 32 lanes of either half-wave
        HY_UNROLL
        for (int q = 0; q < Cfg::NIB; ++q) {
            HY_UNROLL
            for (int r = 0; r < 16; ++r) Hs[(wave * FLT_O + 32 * q + crow(r, half)) * FLT_TP + n] = accf[q][r];
        }
        __syncthreads();
        float s = 0.f;
        for (int j = 0; j < FLT_TP; ++j) s += Hs[(wave * FLT_O + lane) * FLT_TP + ((j + lane) & (FLT_TP - 1))];
        a.part_f[(size_t)(blockIdx.x * FLT_WAVES + wave) * FLT_O + lane] = s;
    }
}

// (filter.hip defines FLT_DECLARE_ONLY: the non-template kernels below are defined once, in fftconv.hip's translation unit)
enum { FLT_RED_J = 16, FLT_RED_S = 16 };
// Several such reductions in ONE launch (round 4: a layer's weight, bias and frequency gradients, the weight and bias gradients of the add +
// LayerNorm backward): each is a handful of workgroups and a dependent launch of its own cost ~7 us of an otherwise idle GPU -- ten per filter
// backward, two per norm.  Job i owns blocks [first[i], first[i + 1]); the arithmetic and its order are filter_reduce_strided_kernel's.
struct RedJob { const float* part; float* out; int count, n, stride, accumulate; };
struct RedJobs { RedJob j[4]; int first[5]; int njobs; };
#ifdef FLT_DECLARE_ONLY
__global__ void filter_reduce_multi_kernel(RedJobs jobs);
__global__ void filter_reduce_kernel(const float* part, float* out, int count, int n, int accumulate);
__global__ void filter_reduce_strided_kernel(const float* part, float* out, int count, int n, int stride);
__global__ void filter_compact_kernel(const float* src, float* dst, int rows, int cols, int used);
#else
// out[j] (+)= sum_c part[c][j] in a fixed order: a workgroup owns 16 outputs, its 16 thread rows sum interleaved
// slices of c (independent loads in flight), thread row 0 adds the slices in order.
__global__ void __launch_bounds__(256) filter_reduce_kernel(const float* part, float* out, int count, int n, int accumulate) {
    HY_SMEM(smem);
    HY_LDS float* sm = HY_LDS_CAST(float, smem);            // [FLT_RED_S][FLT_RED_J]
    const int jj = threadIdx.x & (FLT_RED_J - 1), cs = threadIdx.x / FLT_RED_J;
    const int j = blockIdx.x * FLT_RED_J + jj;
    const int jc = j < n ? j : n - 1;
    float s = 0.f;
    int c = cs;
    for (; c + 7 * FLT_RED_S < count; c += 8 * FLT_RED_S) {
        float v[8];
        HY_UNROLL
        for (int u = 0; u < 8; ++u) v[u] = part[(size_t)(c + u * FLT_RED_S) * n + jc];
        HY_UNROLL
        for (int u = 0; u < 8; ++u) s += v[u];
    }
    for (; c < count; c += FLT_RED_S) s += part[(size_t)c * n + jc];
    sm[cs * FLT_RED_J + jj] = s;
    __syncthreads();
    if (cs == 0 && j < n) {
        float r = accumulate ? out[j] : 0.f;
        for (int q = 0; q < FLT_RED_S; ++q) r += sm[q * FLT_RED_J + jj];
        out[j] = r;
    }
}

// the same reduction over partial rows that are `stride` floats apart (n <= stride)
__global__ void __launch_bounds__(256) filter_reduce_strided_kernel(const float* part, float* out, int count, int n, int stride) {
    HY_SMEM(smem);
    HY_LDS float* sm = HY_LDS_CAST(float, smem);
    const int jj = threadIdx.x & (FLT_RED_J - 1), cs = threadIdx.x / FLT_RED_J;
    const int j = blockIdx.x * FLT_RED_J + jj;
    const int jc = j < n ? j : n - 1;
    float s = 0.f;
    int c = cs;
    for (; c + 7 * FLT_RED_S < count; c += 8 * FLT_RED_S) {          // eight independent loads in flight, added in the same order
        float v[8];
        HY_UNROLL
        for (int u = 0; u < 8; ++u) v[u] = part[(size_t)(c + u * FLT_RED_S) * stride + jc];
        HY_UNROLL
        for (int u = 0; u < 8; ++u) s += v[u];
    }
    for (; c < count; c += FLT_RED_S) s += part[(size_t)c * stride + jc];
    sm[cs * FLT_RED_J + jj] = s;
    __syncthreads();
    if (cs == 0 && j < n) {
        float r = 0.f;
        for (int q = 0; q < FLT_RED_S; ++q) r += sm[q * FLT_RED_J + jj];
        out[j] = r;
    }
}

__global__ void __launch_bounds__(256) filter_reduce_multi_kernel(RedJobs jobs) {
    HY_SMEM(smem);
    HY_LDS float* sm = HY_LDS_CAST(float, smem);
    int ji = 0;
    HY_UNROLL
    for (int i = 1; i < 4; ++i)
        if (i < jobs.njobs && (int)blockIdx.x >= jobs.first[i]) ji = i;
    const RedJob job = jobs.j[ji];
    const float* part = job.part;
    const int count = job.count, n = job.n, stride = job.stride;
    const int jj = threadIdx.x & (FLT_RED_J - 1), cs = threadIdx.x / FLT_RED_J;
    const int j = ((int)blockIdx.x - jobs.first[ji]) * FLT_RED_J + jj;
    const int jc = j < n ? j : n - 1;
    float s = 0.f;
    int c = cs;
    for (; c + 7 * FLT_RED_S < count; c += 8 * FLT_RED_S) {          // eight independent loads in flight, added in the same order
        float v[8];
        HY_UNROLL
        for (int u = 0; u < 8; ++u) v[u] = part[(size_t)(c + u * FLT_RED_S) * stride + jc];
        HY_UNROLL
        for (int u = 0; u < 8; ++u) s += v[u];
    }
    for (; c < count; c += FLT_RED_S) s += part[(size_t)c * stride + jc];
    sm[cs * FLT_RED_J + jj] = s;
    __syncthreads();
    if (cs == 0 && j < n) {
        float r = job.accumulate ? job.out[j] : 0.f;
        for (int q = 0; q < FLT_RED_S; ++q) r += sm[q * FLT_RED_J + jj];
        job.out[j] = r;
    }
}

// dst (rows, used) <- first `used` columns of src (rows, cols)
__global__ void __launch_bounds__(256) filter_compact_kernel(const float* src, float* dst, int rows, int cols, int used) {
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j < rows * used) dst[j] = src[(j / used) * cols + j % used];
}

#endif  // FLT_DECLARE_ONLY

// host side: collect up to four reductions, then one launch
struct RedBatch {
    RedJobs jobs;
    RedBatch() { jobs.njobs = 0; jobs.first[0] = 0; }
    void add(const float* part, float* out, int count, int n, int stride, int accumulate) {
        RedJob& j = jobs.j[jobs.njobs];
        j.part = part; j.out = out; j.count = count; j.n = n; j.stride = stride; j.accumulate = accumulate;
        jobs.first[jobs.njobs + 1] = jobs.first[jobs.njobs] + (n + FLT_RED_J - 1) / FLT_RED_J;
        ++jobs.njobs;
    }
    int blocks() const { return jobs.first[jobs.njobs]; }
};

}  // namespace hyena
