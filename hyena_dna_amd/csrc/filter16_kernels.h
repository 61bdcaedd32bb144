// filter16_kernels.h -- the implicit Hyena filter under 16-bit autocast: the same chain as filter_kernels.h with the four products on
// the 16-bit matrix cores (v_mfma_f32_32x32x16_bf16 / _f16) and every rounding the reference's autocast graph performs.
//
// Reference (src/models/sequence/hyena.py:199-215 the sine MLP, 96-106 Sin, 152-155 modulation) under torch.autocast(dtype T):
// every nn.Linear casts its input, weight and bias to T and returns T (fp32 accumulation, ONE rounding of acc + bias);
// Sin multiplies the fp32 parameter `freq` with the T tensor -- type promotion makes that product, and the sine, fp32;
// the next Linear rounds its input to T again; the modulation multiplies the T output of the last Linear with an fp32 tensor.
// With R() = round-to-nearest-even to T:
//     a0 = R(R(W0) R(z_l) + R(b0))       h0 = R(sin(f a0))
//     a1 = R(R(W1) h0 + R(b1))           h1 = R(sin(f a1))
//     a2 = R(R(W2) h1 + R(b2))           h2 = R(sin(f a2))
//     y  = R(R(W3) h2)                   k[d, l] = y[d] (exp(-t_l |delta_d|) + shift)                  (fp32)
// and autograd's backward of that graph:
//     d3 = R(dk m)        dh2 = R(R(W3)^T d3)      g2 = dh2 cos(f a2)     dfreq += sum g2 a2      d2 = R(g2 f)     ... down to d0
//     dW_i = sum_l d_i h_(i-1)^T,  db_i = sum_l d_i        (16-bit operands, fp32 sums; the reference rounds these sums to T once
//                                                            more before casting them to the fp32 parameters' .grad -- not repeated)
// scratch/filter16_model.py restates this in PyTorch ops: against the oracle under torch.autocast('cpu', bfloat16) the filter is
// bit-identical and the gradients agree to the reference's own final rounding (1.6e-3); the fp32 kernels of filter_kernels.h are
// 2e-1 away from that graph at the same weights (sin(10 a) amplifies the 2^-9 roundings of a) -- closer to the fp64 truth, but not
// what the reference computes under autocast.  tests/test_filter16_emu.py / test_gpu_filter.py hold these kernels to the autocast oracle.
//
// Kernels:
//   flt16_fwd_kernel        a wavefront owns 32 positions (lane = position) and carries the chain in registers, as in
//                           filter_kernels.h: the C/D layout of v_mfma_f32_32x32x16 (registers = features (r & 3) + 8 (r >> 2) +
//                           4 (lane / 32)) is rounded, passed through the sine and packed -- registers 8 j .. 8 j + 7 of a feature
//                           block ARE the eight contraction values a lane supplies to one MFMA of the next layer, so the weights
//                           (A operand, 16 bytes per lane from LDS) are simply stored in that order.  8 MFMAs per hidden layer and
//                           32-position tile where the fp32 kernel issues 64 of twice the length.  Pre-activations are saved as 16-bit
//                           PAIRS (features 2 p, 2 p + 1 in one 32-bit word, [32][L] words per layer): one 4-byte store per lane.
//   flt16_layer_bwd_kernel  one layer of the backward per launch (W3, W2, W1): dh = W^T delta (contraction over the output features,
//                           delta as B operand straight from global memory: fp32 dk with the modulation applied on load for W3, pair
//                           words for the inner layers), the activation derivative from the saved pre-activation, delta_prev written as
//                           pair words (fp32 rows for the layer that feeds the first one), sin(f a) staged to LDS as 16-bit; then
//                           dW += delta h^T (contraction over POSITIONS, 16 per MFMA: delta rows as 16/32-byte pieces from global / L2,
//                           h rows as ds_read_b128 from LDS).  Accumulators stay in registers over the persistent loop; partial sums
//                           leave per workgroup and are added in a fixed order by filter_reduce_kernel (bitwise reproducible).
//   The first layer (W0, contraction length E <= 8) runs filter_kernels.h's fp32 kernel on the rounded operands (FilterBwdArgs::rdt).
//
// Compiled by hipcc for gfx950 (product) and, with -DHIPEMU, by g++ against tests/hipemu (tests only).
#pragma once
#include "filter_kernels.h"

namespace hyena {
namespace f16k {

struct Frag { uint32_t w[4]; };                       // eight 16-bit values: one A or B operand of v_mfma_f32_32x32x16
#ifdef HIPEMU
template <int DT>
__device__ __forceinline__ f32x16 mfma16(const Frag& a, const Frag& b, f32x16 c) {
    hipemu::u32x4 x, y;
    __builtin_memcpy(x.w, a.w, 16);
    __builtin_memcpy(y.w, b.w, 16);
    return hipemu::mfma_f32_32x32x16_h<DT == DT_BF16>(x, y, c);
}
__device__ __forceinline__ Frag lds_ld16(const HY_LDS char* p) { Frag f; __builtin_memcpy(f.w, p, 16); return f; }
__device__ __forceinline__ uint32_t fb_ldu(FBuf b, unsigned voff, unsigned soff) {
    if ((size_t)voff + 4 > b.n) return 0u;
    uint32_t v; __builtin_memcpy(&v, b.p + voff + soff, 4); return v;
}
__device__ __forceinline__ void fb_stu(FBuf b, unsigned voff, uint32_t v) {
    if ((size_t)voff + 4 <= b.n) __builtin_memcpy(b.p + voff, &v, 4);
}
__device__ __forceinline__ void fb_stf(FBuf b, unsigned voff, float v) { fb_st(b, voff, v); }
__device__ __forceinline__ void fb_ld4u(FBuf b, unsigned voff, uint32_t* v) {      // fully in range by contract
    if ((size_t)voff + 16 > b.n) abort();
    __builtin_memcpy(v, b.p + voff, 16);
}
#else
template <int DT>
__device__ __forceinline__ f32x16 mfma16(const Frag& a, const Frag& b, f32x16 c) {
    if constexpr (DT == DT_BF16) {
        typedef __bf16 v8 __attribute__((ext_vector_type(8)));
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(v8, a), __builtin_bit_cast(v8, b), c, 0, 0, 0);
    } else {
        typedef _Float16 v8 __attribute__((ext_vector_type(8)));
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(v8, a), __builtin_bit_cast(v8, b), c, 0, 0, 0);
    }
}
typedef unsigned f16_lvec __attribute__((ext_vector_type(4)));
__device__ __forceinline__ Frag lds_ld16(const HY_LDS char* p) { return __builtin_bit_cast(Frag, *reinterpret_cast<const HY_LDS f16_lvec*>(p)); }
// F16_POL_ST: cache policy of this path's stores (k, the saved pre-activations, the layer gradients: written once, read by a LATER launch);
// 2 = non-temporal: forward 0.41 -> 0.35 ms, forward + backward 1.48 -> 1.40 ms at L = 2^20 (profiles/r3x_filter16_ab.txt)
#ifndef F16_POL_ST
#define F16_POL_ST 2
#endif
__device__ __forceinline__ uint32_t fb_ldu(FBuf b, unsigned voff, unsigned soff) { return __builtin_amdgcn_raw_buffer_load_b32(b.r, voff, soff, 0); }
__device__ __forceinline__ void fb_stu(FBuf b, unsigned voff, uint32_t v) { __builtin_amdgcn_raw_buffer_store_b32(v, b.r, voff, 0, F16_POL_ST); }
__device__ __forceinline__ void fb_stf(FBuf b, unsigned voff, float v) { __builtin_amdgcn_raw_buffer_store_b32(f2u(v), b.r, voff, 0, F16_POL_ST); }
__device__ __forceinline__ void fb_ld4u(FBuf b, unsigned voff, uint32_t* v) {
    const hy_u4 w = __builtin_amdgcn_raw_buffer_load_b128(b.r, voff, 0, 0);
    v[0] = w.x; v[1] = w.y; v[2] = w.z; v[3] = w.w;
}
#endif

template <int DT> __device__ __forceinline__ uint16_t cvt16(float v) { return Elem<DT>::cvt(v); }
template <int DT> __device__ __forceinline__ float dec16(uint16_t r) { return Elem<DT>::dec(r); }
template <int DT> __device__ __forceinline__ float rnd16(float v) { return dec16<DT>(cvt16<DT>(v)); }
template <int DT> __device__ __forceinline__ uint32_t pack16(float lo, float hi) { return (uint32_t)cvt16<DT>(lo) | ((uint32_t)cvt16<DT>(hi) << 16); }
template <int DT> __device__ __forceinline__ float lo16(uint32_t w) { return dec16<DT>((uint16_t)(w & 0xffffu)); }
template <int DT> __device__ __forceinline__ float hi16(uint32_t w) { return dec16<DT>((uint16_t)(w >> 16)); }

enum {
    F16_WROW = 2 * FLT_O + 16,            // bytes of a 64-wide weight row in LDS: +16 so that the 16-byte reads of 8 neighbouring lanes hit 32 banks
    F16_HROW = 2 * FLT_WG_POS + 16,       // bytes of a staged activation row (256 positions)
    F16_KSTEPS = FLT_WG_POS / 16          // MFMA steps over the positions of a workgroup tile
};

// byte offset, inside a forward weight row, of the value that multiplies input feature c (0 .. 63): MFMA step s = 2 (c / 32) + j,
// half-wave h, element i, where register r = 8 j + i of half h holds feature (r & 3) + 8 (r >> 2) + 4 h of the block
__device__ __forceinline__ constexpr int f16_wslot(int c) {
    const int cb = c >> 5, cp = c & 31, h = (cp >> 2) & 1, r = (cp & 3) + 4 * (cp >> 3), j = r >> 3, i = r & 7;
    return ((2 * (2 * cb + j) + h) * 8 + i) * 2;
}

// ---------------------------------------------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------------------------------------------
template <int D>
struct F16FwdLds {
    static constexpr int W0 = 0;                             // [64] x 16 B: the (zero-padded) embedding weights, one fragment per row
    static constexpr int W1 = W0 + FLT_O * 16;               // [64] x F16_WROW
    static constexpr int W2 = W1 + FLT_O * F16_WROW;
    static constexpr int W3 = W2 + FLT_O * F16_WROW;         // [D] x F16_WROW
    static constexpr int CST = W3 + D * F16_WROW;            // floats: R(b0) | R(b1) | R(b2) | freq | |delta| log2(e)
    static constexpr size_t BYTES = CST + (4 * FLT_O + D) * sizeof(float);
};

// rounds the pre-activations, saves them as pair words, and returns sin(f .) rounded and packed as the next layer's B fragments
template <bool SAVE, int DT>
__device__ __forceinline__ void f16_act(const f32x16 (&x)[2], Frag (&hb)[4], const HY_LDS float* freq, FBuf save, unsigned voff,
                                        unsigned L4, int half) {
    HY_UNROLL
    for (int cb = 0; cb < 2; ++cb) {
        HY_UNROLL
        for (int r = 0; r < 16; r += 2) {
            const int f = 32 * cb + crow(r, half);                   // even; register r + 1 holds feature f + 1
            const uint32_t aw = pack16<DT>(x[cb][r], x[cb][r + 1]);
            if (SAVE) fb_stu(save, voff + (unsigned)(f >> 1) * L4, aw);
            float s0, s1, c0, c1;
            hy_sincos(freq[f] * lo16<DT>(aw), &s0, &c0);
            hy_sincos(freq[f + 1] * hi16<DT>(aw), &s1, &c1);
            (void)c0; (void)c1;
            hb[2 * cb + (r >> 3)].w[(r & 7) >> 1] = pack16<DT>(s0, s1);
        }
    }
}

template <int D, bool SAVE, int DT>
__global__ void __launch_bounds__(FLT_THREADS, 2) flt16_fwd_kernel(FilterArgs a) {
    typedef F16FwdLds<D> Lds;
    HY_SMEM(smem);
    HY_LDS char* sm = HY_LDS_CAST(char, smem);
    HY_LDS float* cst = HY_LDS_CAST(float, sm + Lds::CST);
    const int tid = threadIdx.x;
    for (int i = tid; i < FLT_O * 8; i += FLT_THREADS) {
        const int o = i >> 3, e = i & 7;
        *HY_LDS_CAST(uint16_t, sm + Lds::W0 + o * 16 + e * 2) = cvt16<DT>(e < a.E ? a.w0[o * a.E + e] : 0.f);
    }
    for (int i = tid; i < FLT_O * FLT_O; i += FLT_THREADS) {
        const int o = i >> 6, slot = f16_wslot(i & 63);
        *HY_LDS_CAST(uint16_t, sm + Lds::W1 + o * F16_WROW + slot) = cvt16<DT>(a.w1[i]);
        *HY_LDS_CAST(uint16_t, sm + Lds::W2 + o * F16_WROW + slot) = cvt16<DT>(a.w2[i]);
    }
    for (int i = tid; i < D * FLT_O; i += FLT_THREADS)
        *HY_LDS_CAST(uint16_t, sm + Lds::W3 + (i >> 6) * F16_WROW + f16_wslot(i & 63)) = cvt16<DT>(a.w3[i]);
    for (int i = tid; i < FLT_O; i += FLT_THREADS) {
        cst[i] = rnd16<DT>(a.b0[i]);
        cst[FLT_O + i] = rnd16<DT>(a.b1[i]);
        cst[2 * FLT_O + i] = rnd16<DT>(a.b2[i]);
        cst[3 * FLT_O + i] = a.freq[i];
    }
    for (int i = tid; i < D; i += FLT_THREADS) cst[4 * FLT_O + i] = a.modulate ? fabsf(a.deltas[i]) * FLT_LOG2E : 0.f;
    __syncthreads();

    const int lane = tid & 63, wave = tid >> 6, half = lane >> 5, n = lane & 31;
    const int ntiles = (a.L + FLT_TP - 1) / FLT_TP;
    const HY_LDS float* freq = cst + 3 * FLT_O;
    const HY_LDS float* cdec = cst + 4 * FLT_O;                // zeros without modulation: exp2(-t 0) + 0 = 1, no branch per element
    const float shift = a.modulate ? a.shift : 0.f;
    const unsigned L4 = (unsigned)a.L * 4u, K4 = (unsigned)a.ldk * 4u, S4 = (unsigned)a.lds * 4u;      // bytes per row: t, k, saved
    const FBuf Kb = make_fbuf(a.k, (size_t)D * K4);
    const FBuf Ab = make_fbuf(a.acts, SAVE ? (size_t)3 * (FLT_O / 2) * S4 : 0);
    const FBuf Zb = make_fbuf(a.z, (size_t)a.L * a.zs * 4u);
    const FBuf Tb = make_fbuf(a.t, L4);
    const HY_LDS char* wrow = sm + n * F16_WROW + half * 16;            // + layer base + 32 ob F16_WROW + 32 s
    for (int tile = blockIdx.x * FLT_WAVES + wave; tile < ntiles; tile += gridDim.x * FLT_WAVES) {
        const int pos = tile * FLT_TP + n;
        const bool valid = pos < a.L;
        const unsigned vpos = valid ? (unsigned)pos * 4u : FLT_OOB;
        // layer 0: one MFMA step; the lanes of the lower half-wave supply the embedding (k = 0 .. 7), the upper half zeros
        Frag zb;
        {
            float zv[8];
            HY_UNROLL
            for (int e = 0; e < 8; ++e) zv[e] = fb_ld(Zb, valid && half == 0 && e < a.E ? (unsigned)(pos * a.zs + e) * 4u : FLT_OOB, 0);
            HY_UNROLL
            for (int e = 0; e < 4; ++e) zb.w[e] = pack16<DT>(zv[2 * e], zv[2 * e + 1]);
        }
        f32x16 x[2];
        Frag hb[4];
        HY_UNROLL
        for (int ob = 0; ob < 2; ++ob) {
            HY_UNROLL
            for (int r = 0; r < 16; ++r) x[ob][r] = cst[32 * ob + crow(r, half)];
            Frag wa = lds_ld16(sm + Lds::W0 + (32 * ob + n) * 16);
            HY_UNROLL
            for (int e = 0; e < 4; ++e) wa.w[e] = half ? 0u : wa.w[e];
            x[ob] = mfma16<DT>(wa, zb, x[ob]);
        }
        HY_SCHED_FENCE();
        f16_act<SAVE, DT>(x, hb, freq, Ab, vpos, S4, half);
        HY_SCHED_FENCE();
        HY_UNROLL
        for (int layer = 1; layer < 3; ++layer) {
            const int wbase = layer == 1 ? Lds::W1 : Lds::W2;
            HY_UNROLL
            for (int ob = 0; ob < 2; ++ob) {
                HY_UNROLL
                for (int r = 0; r < 16; ++r) x[ob][r] = cst[layer * FLT_O + 32 * ob + crow(r, half)];
                HY_UNROLL
                for (int s = 0; s < 4; ++s) x[ob] = mfma16<DT>(lds_ld16(wrow + wbase + 32 * ob * F16_WROW + 32 * s), hb[s], x[ob]);
            }
            HY_SCHED_FENCE();
            f16_act<SAVE, DT>(x, hb, freq, Ab, vpos + (unsigned)layer * (FLT_O / 2) * S4, S4, half);
            HY_SCHED_FENCE();
        }
        // last layer + modulation, 32 output channels at a time
        const float tl = fb_ld(Tb, vpos, 0);
        for (int db = 0; db < D / 32; ++db) {
            f32x16 y;
            HY_UNROLL
            for (int r = 0; r < 16; ++r) y[r] = 0.f;
            HY_UNROLL
            for (int s = 0; s < 4; ++s) y = mfma16<DT>(lds_ld16(wrow + Lds::W3 + 32 * db * F16_WROW + 32 * s), hb[s], y);
            HY_UNROLL
            for (int r = 0; r < 16; ++r) {
                const int d = 32 * db + crow(r, half);
                fb_stf(Kb, vpos + (unsigned)d * K4, rnd16<DT>(y[r]) * (hy_exp2(-tl * cdec[d]) + shift));
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// backward, one layer per launch (the layers with a 64-wide input: W3, W2, W1)
// ---------------------------------------------------------------------------------------------------------------
struct F16BwdArgs {
    const void* dout;        // MOD: dk (NO, L) fp32, modulation applied on load;  else (NO / 2, L) pair words
    const float* w;          // (NO, 64) fp32: this layer's weight
    const void* aprev;       // (32, L) pair words: saved pre-activation of the previous layer
    const float* freq;       // (64,)
    const float* t;          // MOD (L,)
    const float* deltas;     // MOD (NO,)
    void* dprev;             // out: (32, L) pair words, or (64, L) fp32 rows (OUTF32)
    float* part_w;           // out: [slots][NO][64]
    float* part_b;           // out: [2 slots][NO] or nullptr
    float* part_f;           // out: [gridDim.x * 8][64]
    float shift;
    int modulate;
    int L;
    int ldo, lda, ldp;       // row pitch (32-bit words, >= L) of dout, aprev, dprev (packed: L)
};

// wavefronts per SIMD the backward kernels are compiled for (4 = two workgroups per CU, at most 128 registers: measured slower,
// 1.52 -> 1.91 ms forward + backward at L = 2^20 with 20 - 75 spilled registers, profiles/r3v_filter16_ab.txt)
#ifndef F16_BWD_MINW
#define F16_BWD_MINW 2
#endif
// memory-level parallelism of the backward kernels (at 2 wavefronts per SIMD nothing else hides a round trip): MFMA steps of the
// feature contraction whose operand loads are in flight together, and MFMA steps of the position contraction per round of loads
// (2 / 1: 1.57 ms forward + backward at L = 2^20; 4 / 4: 1.48; 8 / 4 and 4 / 8: 1.51 -- profiles/r3x_filter16_ab.txt)
#ifndef F16_S1
#define F16_S1 4
#endif
#ifndef F16_S2
#define F16_S2 4
#endif

template <int NO>
struct F16BwdLds {
    static constexpr int WROW = 2 * NO + 16;                       // bytes of a row of W^T (input feature i: NO values)
    static constexpr int WT = 0;                                   // [64] x WROW
    static constexpr int HS = WT + FLT_O * WROW;                   // [64] x F16_HROW: sin(f a) of the tile, 16-bit
    static constexpr int CST = HS + FLT_O * F16_HROW;              // floats: freq[64] | cdec[NO] | t[FLT_WG_POS]
    static constexpr size_t BYTES = CST + (FLT_O + NO + FLT_WG_POS) * sizeof(float);
};

template <int NO, bool MOD, int DT, bool OUTF32>
__global__ void __launch_bounds__(FLT_THREADS, F16_BWD_MINW) flt16_layer_bwd_kernel(F16BwdArgs a) {
    typedef FltBwdCfg<NO, FLT_O> Cfg;
    typedef F16BwdLds<NO> Lds;
    constexpr int KS16 = NO / 16;                                  // MFMA steps over this layer's output features
    HY_SMEM(smem);
    HY_LDS char* sm = HY_LDS_CAST(char, smem);
    HY_LDS float* freq = HY_LDS_CAST(float, sm + Lds::CST);
    HY_LDS float* cdec = freq + FLT_O;
    HY_LDS float* Tt = cdec + NO;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, n = lane & 31;
    const int L = a.L;
    const bool modulate = MOD && a.modulate != 0;
    const float shift = modulate ? a.shift : 0.f;              // (cdec = 0 without modulation: the factor is exp2(0) + 0 = 1, no branch per element)

    for (int i = tid; i < NO * FLT_O; i += FLT_THREADS) {          // W^T, rounded: row = input feature, NO values in natural order
        const int o = i >> 6, c = i & 63;
        *HY_LDS_CAST(uint16_t, sm + Lds::WT + c * Lds::WROW + o * 2) = cvt16<DT>(a.w[i]);
    }
    for (int i = tid; i < FLT_O; i += FLT_THREADS) freq[i] = a.freq[i];
    for (int i = tid; i < NO; i += FLT_THREADS) cdec[i] = modulate ? fabsf(a.deltas[i]) * FLT_LOG2E : 0.f;
    __syncthreads();

    // this wavefront's share of dW: blocks (rb, cb0 .. cb0 + BPW - 1), MFMA steps [ks NKS, (ks + 1) NKS) of every tile
    const int group = wave % Cfg::GROUPS, ks = wave / Cfg::GROUPS;
    const int blk0 = group * Cfg::BPW;
    const int rb = blk0 / Cfg::NIB, cb0 = blk0 % Cfg::NIB;
    constexpr int NKS = F16_KSTEPS / Cfg::KS;
    f32x16 accw[Cfg::BPW];
    HY_UNROLL
    for (int q = 0; q < Cfg::BPW; ++q) {
        HY_UNROLL
        for (int r = 0; r < 16; ++r) accw[q][r] = 0.f;
    }
    float accb = 0.f;
    f32x16 accf[2];
    HY_UNROLL
    for (int q = 0; q < 2; ++q) {
        HY_UNROLL
        for (int r = 0; r < 16; ++r) accf[q][r] = 0.f;
    }

    const unsigned L4 = (unsigned)L * 4u, O4 = (unsigned)a.ldo * 4u, A4 = (unsigned)a.lda * 4u, P4 = (unsigned)a.ldp * 4u;
    const FBuf Db = make_fbuf(a.dout, MOD ? (size_t)NO * O4 : (size_t)(NO / 2) * O4);
    const FBuf Ab = make_fbuf(a.aprev, (size_t)(FLT_O / 2) * A4);
    const FBuf Pb = make_fbuf(a.dprev, OUTF32 ? (size_t)FLT_O * P4 : (size_t)(FLT_O / 2) * P4);
    const FBuf Tb = make_fbuf(a.t, MOD ? L4 : 0);
    const HY_LDS char* wtrow = sm + Lds::WT + n * Lds::WROW + half * 16;        // + 32 q WROW + 32 s
    const int niter = (L + FLT_WG_POS - 1) / FLT_WG_POS;
    for (int it = blockIdx.x; it < niter; it += gridDim.x) {
        const int p0 = it * FLT_WG_POS;
        if (MOD && tid < FLT_WG_POS) Tt[tid] = fb_ld(Tb, p0 + tid < L ? (unsigned)(p0 + tid) * 4u : FLT_OOB, 0);
        // ---- contraction over this layer's output features for the wavefront's 32 positions: dh = W^T delta
        {
            const int pos = p0 + FLT_TP * wave + n;
            const bool valid = pos < L;
            const unsigned vpos = valid ? (unsigned)pos * 4u : FLT_OOB;
            const float tl = MOD ? fb_ld(Tb, vpos, 0) : 0.f;
            // the saved pre-activations of the tile (pair words): issued now, they land while the MFMAs below run
            uint32_t apw[2][8];
            HY_UNROLL
            for (int q = 0; q < 2; ++q) {
                HY_UNROLL
                for (int r = 0; r < 16; r += 2)
                    apw[q][r >> 1] = fb_ldu(Ab, vpos + (unsigned)(2 * half) * A4, (unsigned)((32 * q + crow(r, 0)) >> 1) * A4);
            }
            f32x16 dh[2];
            HY_UNROLL
            for (int q = 0; q < 2; ++q) {
                HY_UNROLL
                for (int r = 0; r < 16; ++r) dh[q][r] = 0.f;
            }
            HY_SCHED_FENCE();
            // S1 MFMA steps (16 output features each) per round: 8 S1 / 4 S1 loads in flight per lane
            constexpr int S1 = KS16 < F16_S1 ? KS16 : F16_S1;
            for (int s0 = 0; s0 < KS16; s0 += S1) {
                Frag db[S1];
                if (MOD) {
                    float dv[8 * S1];
#ifdef F16_DBG_NO_P1LOAD        // (profiling builds only: what the first read of dk costs)
                    HY_UNROLL
                    for (int j = 0; j < 8 * S1; ++j) { dv[j] = 1e-3f * (float)(s0 + j + lane); HY_OPAQUE(dv[j]); }
#else
                    HY_UNROLL
                    for (int j = 0; j < 8 * S1; ++j)
                        dv[j] = fb_ld(Db, vpos + (unsigned)(8 * half) * O4, (unsigned)(16 * (s0 + (j >> 3)) + (j & 7)) * O4);
#endif
                    HY_UNROLL
                    for (int j = 0; j < 8 * S1; ++j) {
                        const int o = 16 * (s0 + (j >> 3)) + 8 * half + (j & 7);
                        dv[j] *= hy_exp2(-tl * cdec[o]) + shift;
                    }
                    HY_UNROLL
                    for (int j = 0; j < 4 * S1; ++j) db[j >> 2].w[j & 3] = pack16<DT>(dv[2 * j], dv[2 * j + 1]);
                } else {
                    HY_UNROLL
                    for (int j = 0; j < 4 * S1; ++j)
                        db[j >> 2].w[j & 3] = fb_ldu(Db, vpos + (unsigned)(4 * half) * O4, (unsigned)(8 * (s0 + (j >> 2)) + (j & 3)) * O4);
                }
                HY_UNROLL
                for (int u = 0; u < S1; ++u) {
                    HY_UNROLL
                    for (int q = 0; q < 2; ++q)
                        dh[q] = mfma16<DT>(lds_ld16(wtrow + 32 * q * Lds::WROW + 32 * (s0 + u)), db[u], dh[q]);
                }
            }
            HY_UNROLL
            for (int q = 0; q < 2; ++q) {
                HY_UNROLL
                for (int r = 0; r < 16; r += 2) {
                    const int f = 32 * q + crow(r, half);            // even; register r + 1: feature f + 1
                    const float ap0 = lo16<DT>(apw[q][r >> 1]), ap1 = hi16<DT>(apw[q][r >> 1]);
                    const float fr0 = freq[f], fr1 = freq[f + 1];
                    float sn0, cs0, sn1, cs1;
                    hy_sincos(fr0 * ap0, &sn0, &cs0);
                    hy_sincos(fr1 * ap1, &sn1, &cs1);
                    const float g0 = rnd16<DT>(dh[q][r]) * cs0, g1 = rnd16<DT>(dh[q][r + 1]) * cs1;
                    accf[q][r] += g0 * ap0;
                    accf[q][r + 1] += g1 * ap1;
                    if (OUTF32) {
                        fb_stf(Pb, vpos + (unsigned)f * P4, rnd16<DT>(g0 * fr0));
                        fb_stf(Pb, vpos + (unsigned)(f + 1) * P4, rnd16<DT>(g1 * fr1));
                    } else {
                        fb_stu(Pb, vpos + (unsigned)(f >> 1) * P4, pack16<DT>(g0 * fr0, g1 * fr1));
                    }
                    HY_LDS char* hp = sm + Lds::HS + f * F16_HROW + (FLT_TP * wave + n) * 2;
                    *HY_LDS_CAST(uint16_t, hp) = cvt16<DT>(sn0);
                    *HY_LDS_CAST(uint16_t, hp + F16_HROW) = cvt16<DT>(sn1);
                }
            }
        }
        __syncthreads();
        // ---- contraction over positions: dW[rb, cb] += delta[rows of rb][positions] h[rows of cb][positions]^T, 16 positions per MFMA
        {
            const int o = 32 * rb + n;
            constexpr int S2 = NKS < F16_S2 ? NKS : F16_S2;
            const unsigned rowbase = (MOD ? (unsigned)o : (unsigned)(o >> 1)) * (unsigned)a.ldo;
            const bool odd = (o & 1) != 0;                           // inner layers: this row's half of the pair words
            const float cd = MOD ? cdec[o] : 0.f;
            for (int kk0 = ks * NKS; kk0 < (ks + 1) * NKS; kk0 += S2) {
                // the raw operand words of S2 steps first (8 per step: fp32 values or pair words), then the arithmetic
                uint32_t raw[S2][8];
                const bool whole = (a.ldo & 3) == 0 && p0 + 16 * (kk0 + S2) <= L;
#ifdef F16_DBG_NO_P2LOAD        // (profiling builds only, scripts/build_variant.sh: wrong results by construction -- what the second read of dk costs)
                if (MOD) {
                    HY_UNROLL
                    for (int u = 0; u < S2; ++u) {
                        HY_UNROLL
                        for (int j = 0; j < 8; ++j) { raw[u][j] = 0x3c000000u + (unsigned)(kk0 + u + j + lane); HY_OPAQUE(raw[u][j]); }
                    }
                } else
#endif
                if (whole) {
                    HY_UNROLL
                    for (int u = 0; u < S2; ++u) {
                        const unsigned base = (rowbase + (unsigned)(p0 + 16 * (kk0 + u) + 8 * half)) * 4u;
                        fb_ld4u(Db, base, &raw[u][0]);
                        fb_ld4u(Db, base + 16u, &raw[u][4]);
                    }
                } else {
                    HY_UNROLL
                    for (int u = 0; u < S2; ++u) {
                        const int gp = p0 + 16 * (kk0 + u) + 8 * half;
                        const unsigned base = (rowbase + (unsigned)gp) * 4u;
                        HY_UNROLL
                        for (int j = 0; j < 8; ++j) raw[u][j] = fb_ldu(Db, gp + j < L ? base + 4u * j : FLT_OOB, 0);
                    }
                }
                HY_UNROLL
                for (int u = 0; u < S2; ++u) {
                    const int q0 = 16 * (kk0 + u) + 8 * half;        // first of this lane's 8 positions within the tile
                    Frag da;
                    if (MOD) {
                        float av[8];
                        HY_UNROLL
                        for (int j = 0; j < 8; ++j) av[j] = u2f(raw[u][j]) * (hy_exp2(-Tt[q0 + j] * cd) + shift);
                        HY_UNROLL
                        for (int j = 0; j < 4; ++j) da.w[j] = pack16<DT>(av[2 * j], av[2 * j + 1]);
                    } else {
                        HY_UNROLL
                        for (int j = 0; j < 4; ++j)
                            da.w[j] = odd ? (raw[u][2 * j] >> 16) | (raw[u][2 * j + 1] & 0xffff0000u)
                                          : (raw[u][2 * j] & 0xffffu) | (raw[u][2 * j + 1] << 16);
                    }
                    if (cb0 == 0) {
                        HY_UNROLL
                        for (int j = 0; j < 4; ++j) accb += lo16<DT>(da.w[j]) + hi16<DT>(da.w[j]);
                    }
                    HY_UNROLL
                    for (int q = 0; q < Cfg::BPW; ++q) {
                        const Frag hbv = lds_ld16(sm + Lds::HS + (32 * (cb0 + q) + n) * F16_HROW + q0 * 2);
                        accw[q] = mfma16<DT>(da, hbv, accw[q]);
                    }
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
            const int c = 32 * (cb0 + q) + n;
            a.part_w[((size_t)slot * NO + o) * FLT_O + c] = accw[q][r];
        }
    }
    if (a.part_b != nullptr && cb0 == 0) a.part_b[(size_t)(slot * 2 + half) * NO + 32 * rb + n] = accb;
    // frequency gradient: sum of the position-on-lane accumulators over the 32 lanes of either half-wave
    HY_UNROLL
    for (int q = 0; q < 2; ++q) {
        HY_UNROLL
        for (int r = 0; r < 16; ++r) {
            float v = accf[q][r];
            HY_UNROLL
            for (int m = 16; m >= 1; m >>= 1) v += u2f(HY_SHFL_U32(f2u(v), lane ^ m));
            if (n == 0) a.part_f[(size_t)(blockIdx.x * FLT_WAVES + wave) * FLT_O + 32 * q + crow(r, half)] = v;
        }
    }
}

}  // namespace f16k
}  // namespace hyena
