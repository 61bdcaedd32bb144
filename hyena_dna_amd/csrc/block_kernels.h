// block_kernels.h -- the residual glue of a HyenaDNA block around the mixer: (dropout ->) add -> LayerNorm, fused.
//
// Reference (prenorm block, src/models/sequence/simple_lm.py:267-271 / 280-284, long_conv_lm.py:381-396; the fused path
// the reference takes when flash_attn is installed is flash_attn.ops.layer_norm.dropout_add_layer_norm):
//     residual' = dropout(x0) + residual            (fp32 when residual_in_fp32)
//     out       = LayerNorm(residual'; weight, bias, eps)
// returning (out, residual') for prenorm blocks, out alone for the final norm.  HyenaDNA trains with residual dropout 0 and an
// embedding dropout of 0.1, which the reference applies as the FIRST block's dropout (long_conv_lm.py: resid_dropout1 = embed_dropout
// for layer 0).  Dropout (round 4) is part of the pass: the keep / drop decision of element i is a pure function of (seed, i) -- Philox
// 4x32-10, four consecutive elements per call -- so the backward regenerates it and no mask tensor exists (PyTorch's unfused dropout
// reads and writes the activation once more in each direction and keeps a byte per element: 0.86 ms per step at 2^20 x 256 fp32).
//
// One pass forward (read x0, residual; write out, residual'; + mean / rstd per row), one pass backward (read dout,
// residual', d residual'; write dx0 = d residual; per-workgroup partial weight / bias gradients, summed in a fixed
// order by filter_reduce_kernel).  A wavefront owns one row: lane l holds the E = D / 64 consecutive channels l E ..
// l E + E - 1 (one 16-byte access per tensor at D = 256), row statistics by a 6-step butterfly over the wavefront.
// Unfused, PyTorch spends an add, a cast, a LayerNorm and (under autocast) another cast on this: ~2.5x the traffic.
#pragma once
#include "fftconv_kernels.h"

namespace hyena {

enum { BLK_WAVES = 4, BLK_THREADS = BLK_WAVES * 64, BLK_MAX_GRID = 2048, BLK_VMAX = 16 /* token classes of the EMB kernels (the DNA vocabulary: 12 -> 16) */ };

struct AddNormArgs {
    const void* x;          // (rows, D) elements of XDT        fwd: x0          bwd: dout
    const float* res_in;    // (rows, D) fp32 or null           fwd: residual    bwd: gradient w.r.t. residual' (or null)
    const float* weight;    // (D,)
    const float* bias;      // (D,)  fwd only
    void* out;              // (rows, D) elements of ODT        fwd: out         bwd: dx0
    float* res_out;         // (rows, D) fp32                   fwd: residual'   bwd: gradient w.r.t. residual (or null)
    const float* saved;     // bwd: residual' as written by the forward
    float* mean;            // (rows,)  fwd: out, bwd: in
    float* rstd;            // (rows,)
    float* part;            // bwd: [gridDim.x][np][D] partial (dweight | dbias [| column sums of the dx0 values as stored])
    int np = 2;             // bwd: 2, or 3: also the column sums of dx0 -- the bias gradient of the linear layer that produced x0 (out_proj, fc2:
                            // round 6; their own streaming pass over dx0 cost 94 us each at 2^20 x 256)
    long rows;
    int D;
    float eps;
    const long long* ids;   // EMB kernels: (rows,) token ids; x0[row] = table[ids[row]] with table = `x` (V_MAX rows at most, fp32) -- the embedding
                            // is never materialised; the backward returns per-token-class sums of d x0 instead of d x0 (part_e)
    float* part_e;          // EMB bwd: [gridDim.x][BLK_VMAX][D] partial sums of the embedding gradient
    const unsigned long long* seed;   // device pointer to the 64-bit dropout seed, or null: no dropout
    unsigned drop_below;              // an element is dropped when its 32 random bits are < drop_below (= p * 2^32)
    float keep_scale;                 // 1 / (1 - p)
    int V;                            // EMB kernels: rows of the embedding table.  An id outside [0, V) reads NO memory: the forward poisons that
                                      // row of out / residual' with NaN (F.embedding, which this pass replaces, device-asserts there), the
                                      // backward leaves it out of every sum
};

// Philox 4x32-10 (Salmon et al., SC'11): counter (c0, c1, 0, 0), key (k0, k1) -> four 32-bit words
__device__ __forceinline__ void philox4x32_10(unsigned c0, unsigned c1, unsigned k0, unsigned k1, unsigned (&out)[4]) {
    unsigned x0 = c0, x1 = c1, x2 = 0u, x3 = 0u;
    HY_UNROLL
    for (int r = 0; r < 10; ++r) {
        const unsigned long long p0 = (unsigned long long)0xD2511F53u * x0, p1 = (unsigned long long)0xCD9E8D57u * x2;
        const unsigned n0 = (unsigned)(p1 >> 32) ^ x1 ^ k0, n1 = (unsigned)p1, n2 = (unsigned)(p0 >> 32) ^ x3 ^ k1, n3 = (unsigned)p0;
        x0 = n0; x1 = n1; x2 = n2; x3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = x0; out[1] = x1; out[2] = x2; out[3] = x3;
}
// v[e] -> dropout(v[e]) for the E consecutive elements starting at linear index `off` (a multiple of E; E a multiple of 4 or E < 4)
template <int E>
__device__ __forceinline__ void blk_dropout(float (&v)[E], size_t off, unsigned long long seed, unsigned drop_below, float keep_scale) {
    HY_UNROLL
    for (int q = 0; q < (E + 3) / 4; ++q) {
        const size_t idx = (off + 4 * q) >> 2;                  // one Philox call per aligned group of four elements
        unsigned rnd[4];
        philox4x32_10((unsigned)idx, (unsigned)(idx >> 32), (unsigned)seed, (unsigned)(seed >> 32), rnd);
        HY_UNROLL
        for (int i = 0; i < 4 && 4 * q + i < E; ++i) {
            const unsigned word = rnd[(off + 4 * q + i) & 3];   // E < 4: the lane's elements are a part of an aligned group
            v[4 * q + i] = word < drop_below ? 0.f : v[4 * q + i] * keep_scale;
        }
    }
}

__device__ __forceinline__ float wave_sum(float v) {
    HY_UNROLL
    for (int m = 32; m >= 1; m >>= 1) v += u2f(HY_SHFL_U32(f2u(v), (int)((threadIdx.x & 63) ^ m)));
    return v;
}

// E consecutive elements of a row as ONE access (rows start at multiples of D elements and c0 = lane * E, so the
// address is a multiple of E elements: 16 bytes for fp32 at D = 256, 8 bytes for 16-bit types)
template <int DT, int E>
__device__ __forceinline__ void blk_load(const void* base, size_t off, float (&v)[E]) {
    typedef typename Elem<DT>::type elem_t;
    struct __attribute__((aligned(sizeof(elem_t) * (E > 4 ? 4 : E)))) Raw { elem_t e[E]; } raw;
    raw = *reinterpret_cast<const Raw*>(reinterpret_cast<const elem_t*>(base) + off);
    HY_UNROLL
    for (int e = 0; e < E; ++e) v[e] = Elem<DT>::dec(raw.e[e]);
}
template <int DT, int E>
__device__ __forceinline__ void blk_store(void* base, size_t off, const float (&v)[E]) {
    typedef typename Elem<DT>::type elem_t;
    struct __attribute__((aligned(sizeof(elem_t) * (E > 4 ? 4 : E)))) Raw { elem_t e[E]; } raw;
    HY_UNROLL
    for (int e = 0; e < E; ++e) raw.e[e] = Elem<DT>::cvt(v[e]);
    *reinterpret_cast<Raw*>(reinterpret_cast<elem_t*>(base) + off) = raw;
}

template <int XDT, int ODT, int E, bool EMB = false>
__global__ void __launch_bounds__(BLK_THREADS) add_norm_fwd_kernel(AddNormArgs a) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c0 = lane * E;
    float w[E], b[E];
    HY_UNROLL
    for (int e = 0; e < E; ++e) { w[e] = a.weight[c0 + e]; b[e] = a.bias[c0 + e]; }
    const float inv_d = 1.f / (float)a.D;
    const unsigned long long seed = a.seed != nullptr ? *a.seed : 0ull;
    for (long row = (long)blockIdx.x * BLK_WAVES + wave; row < a.rows; row += (long)gridDim.x * BLK_WAVES) {
        const size_t off = (size_t)row * a.D + c0;
        float r[E];
        const long long id = EMB ? a.ids[row] : 0;
        const bool bad_id = EMB && (unsigned long long)id >= (unsigned long long)a.V;   // never index the table with it
        blk_load<XDT, E>(a.x, EMB ? (size_t)(bad_id ? 0 : id) * a.D + c0 : off, r);          // EMB: the row of the embedding table
        if (a.seed != nullptr) blk_dropout<E>(r, off, seed, a.drop_below, a.keep_scale);
        if (bad_id) {                                        // (after the dropout: the whole row, not only its kept elements)
            HY_UNROLL
            for (int e = 0; e < E; ++e) r[e] = __builtin_nanf("");
        }
        if (a.res_in != nullptr) {
            float q[E];
            blk_load<DT_F32, E>(a.res_in, off, q);
            HY_UNROLL
            for (int e = 0; e < E; ++e) r[e] += q[e];
        }
        float s = 0.f;
        HY_UNROLL
        for (int e = 0; e < E; ++e) s += r[e];
        const float mean = wave_sum(s) * inv_d;
        float v = 0.f;
        HY_UNROLL
        for (int e = 0; e < E; ++e) v += (r[e] - mean) * (r[e] - mean);
        const float rstd = 1.f / sqrtf(wave_sum(v) * inv_d + a.eps);
        float o[E];
        HY_UNROLL
        for (int e = 0; e < E; ++e) o[e] = (r[e] - mean) * rstd * w[e] + b[e];
        blk_store<ODT, E>(a.out, off, o);
        blk_store<DT_F32, E>(a.res_out, off, r);
        if (lane == 0) { a.mean[row] = mean; a.rstd[row] = rstd; }
    }
}

template <int GDT, int ODT, int E, bool EMB = false>
__global__ void __launch_bounds__(BLK_THREADS) add_norm_bwd_kernel(AddNormArgs a) {
    HY_SMEM(smem);
    HY_LDS float* red = HY_LDS_CAST(float, smem);            // [BLK_WAVES][np][64 * E]  (EMB: also [BLK_VMAX][64 * E], whichever is larger)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c0 = lane * E;
    float w[E], dw[E], db[E], dxs[E];
    HY_UNROLL
    for (int e = 0; e < E; ++e) { w[e] = a.weight[c0 + e]; dw[e] = 0.f; db[e] = 0.f; dxs[e] = 0.f; }
    const int np = EMB ? 2 : a.np;
    float eacc[EMB ? BLK_VMAX : 1][E];                       // EMB: this wavefront's sums of d x0 per token class (its rows, in order)
    HY_UNROLL
    for (int q = 0; q < (EMB ? BLK_VMAX : 1); ++q) {
        HY_UNROLL
        for (int e = 0; e < E; ++e) eacc[q][e] = 0.f;
    }
    const float inv_d = 1.f / (float)a.D;
    const unsigned long long seed = a.seed != nullptr ? *a.seed : 0ull;
    for (long row = (long)blockIdx.x * BLK_WAVES + wave; row < a.rows; row += (long)gridDim.x * BLK_WAVES) {
        const size_t off = (size_t)row * a.D + c0;
        float g[E], r[E];
        blk_load<GDT, E>(a.x, off, g);
        blk_load<DT_F32, E>(a.saved, off, r);
        const float mean = a.mean[row], rstd = a.rstd[row];
        float s1 = 0.f, s2 = 0.f, xh[E];
        HY_UNROLL
        for (int e = 0; e < E; ++e) {
            xh[e] = (r[e] - mean) * rstd;
            const float dxh = g[e] * w[e];
            s1 += dxh;
            s2 += dxh * xh[e];
            dw[e] += g[e] * xh[e];
            db[e] += g[e];
        }
        s1 = wave_sum(s1) * inv_d;
        s2 = wave_sum(s2) * inv_d;
        float dr[E];
        HY_UNROLL
        for (int e = 0; e < E; ++e) dr[e] = rstd * (g[e] * w[e] - s1 - xh[e] * s2);
        if (a.res_in != nullptr) {
            float h[E];
            blk_load<DT_F32, E>(a.res_in, off, h);
            HY_UNROLL
            for (int e = 0; e < E; ++e) dr[e] += h[e];
        }
        if (a.res_out != nullptr) blk_store<DT_F32, E>(a.res_out, off, dr);
        if (a.seed != nullptr) blk_dropout<E>(dr, off, seed, a.drop_below, a.keep_scale);      // d x0 = d residual' through the same mask
        if (EMB) {
            const long long id = a.ids[row];
            const int v = HY_SGPR((unsigned long long)id < (unsigned long long)a.V ? (int)id : -1);   // the row's token class: wave-uniform; -1: no class
            HY_UNROLL
            for (int q = 0; q < BLK_VMAX; ++q) {
                if (v == q) {
                    HY_UNROLL
                    for (int e = 0; e < E; ++e) eacc[EMB ? q : 0][e] += dr[e];
                }
            }
        } else {
            blk_store<ODT, E>(a.out, off, dr);
            if (np == 3) {                                   // (the values as stored: what a column sum over the dx0 tensor would read)
                HY_UNROLL
                for (int e = 0; e < E; ++e) dxs[e] += Elem<ODT>::dec(Elem<ODT>::cvt(dr[e]));
            }
        }
    }
    const int D = 64 * E;
    if (EMB) {
        // embedding-gradient partials of this workgroup: the 4 wavefronts add their sums into the LDS image one after the other
        HY_UNROLL
        for (int wv = 0; wv < BLK_WAVES; ++wv) {
            if (wave == wv) {
                HY_UNROLL
                for (int q = 0; q < BLK_VMAX; ++q) {
                    HY_UNROLL
                    for (int e = 0; e < E; ++e) {
                        HY_LDS float* slot = red + q * D + c0 + e;
                        *slot = wv == 0 ? eacc[EMB ? q : 0][e] : *slot + eacc[EMB ? q : 0][e];
                    }
                }
            }
            __syncthreads();
        }
        for (int i = threadIdx.x; i < BLK_VMAX * D; i += BLK_THREADS) a.part_e[(size_t)blockIdx.x * BLK_VMAX * D + i] = red[i];
        __syncthreads();
    }
    // weight / bias gradient partials of this workgroup: the 4 wavefronts are added in order
    HY_UNROLL
    for (int e = 0; e < E; ++e) {
        red[(wave * np + 0) * D + c0 + e] = dw[e];
        red[(wave * np + 1) * D + c0 + e] = db[e];
        if (np == 3) red[(wave * np + 2) * D + c0 + e] = dxs[e];
    }
    __syncthreads();
    for (int i = threadIdx.x; i < np * D; i += BLK_THREADS) {
        float s = 0.f;
        HY_UNROLL
        for (int q = 0; q < BLK_WAVES; ++q) s += red[q * np * D + i];
        a.part[(size_t)blockIdx.x * np * D + i] = s;
    }
}

}  // namespace hyena
