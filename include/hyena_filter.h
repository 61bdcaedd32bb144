/* hyena_filter.h -- C ABI of the fused implicit-filter kernels of libhyena_fftconv.so.
 *
 * Replaces, for the HyenaDNA filter configuration, what the reference computes in
 *   HyenaFilter.filter(L)                      src/models/sequence/hyena.py:229-238
 *     PositionalEmbedding.forward              hyena.py:130-131     z[:, :L], t[:, :L]
 *     implicit_filter (Linear/Sin x3, Linear)  hyena.py:199-215, Sin: 96-106
 *     ExponentialModulation.forward            hyena.py:152-155
 * followed by the `(l, d) -> (d, l)` rearrange of HyenaOperator.forward (hyena.py:405-412), and the autograd graph
 * PyTorch builds for it:
 *     a0 = W0 z_l + b0,  a1 = W1 sin(f a0) + b1,  a2 = W2 sin(f a1) + b2,  y = W3 sin(f a2)
 *     k[d, l] = y[d] * (exp(-t_l |delta_d|) + shift)            (modulate = 0:  k = y)
 *
 * Under torch.autocast the reference's four nn.Linear run in the 16-bit autocast type with fp32 accumulation (inputs, weights, biases
 * and outputs rounded to that type; the sine and the modulation stay fp32 by type promotion).  hyena_filter16_fwd / _bwd reproduce that
 * graph rounding by rounding on the 16-bit matrix cores (csrc/filter16_kernels.h); hyena_filter_fwd / _bwd are the fp32 graph.
 *
 * Supported shapes (hyena_filter_supported): filter order 64 with two inner layers, emb_dim <= 8, D in {64, 128, 256}.
 * Anything else is left to the caller's generic path.  All tensors fp32, row-major, device memory of the device that
 * is current for `stream`; nothing is allocated or synchronised inside the entry points.
 */
#ifndef HYENA_FILTER_H
#define HYENA_FILTER_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
    const float* z;        /* (L, z_stride): rows of pos_emb.z (hyena.py:128), first E columns are the embedding */
    const float* t;        /* (L,)          pos_emb.t (hyena.py:129) */
    const float* w0;       /* (64, E)       implicit_filter.0.weight */
    const float* b0;       /* (64,)         implicit_filter.0.bias */
    const float* w1;       /* (64, 64)      implicit_filter.2.weight */
    const float* b1;
    const float* w2;       /* (64, 64)      implicit_filter.4.weight */
    const float* b2;
    const float* w3;       /* (D, 64)       implicit_filter.6.weight (no bias, hyena.py:210) */
    const float* freq;     /* (64,)         the ONE Sin.freq shared by the three activations (hyena.py:199) */
    const float* deltas;   /* (D,)          modulation.deltas (hyena.py:149); may be NULL when modulate == 0 */
    float shift;           /* modulation.shift (hyena.py:144) */
    int modulate;          /* HyenaFilter.modulate (hyena.py:176) */
    int z_stride;
    int L, E, D;
} hyena_filter_params;

typedef struct {           /* gradients, same shapes as the parameters; every pointer required except dz */
    float* dw0;
    float* db0;
    float* dw1;
    float* db1;
    float* dw2;
    float* db2;
    float* dw3;
    float* dfreq;
    float* dz;             /* (E, L) gradient of the embedding rows, TRANSPOSED; NULL when pos_emb.z is a buffer */
} hyena_filter_grads;

/* 1 if the fused kernels cover this configuration. */
int hyena_filter_supported(int L, int E, int order, int D);

/* Bytes of the pre-activation buffer the forward fills for the backward (3 x 64 rows of hyena_filter_row_pitch(L) floats). */
size_t hyena_filter_saved_bytes(int L);

/* Bytes of scratch the backward needs (two (64, L) gradient buffers + per-workgroup partial sums). */
size_t hyena_filter_workspace_bytes(int L, int D);

/* k (D, L) <- filter.  `saved` may be NULL (inference). */
int hyena_filter_fwd(const hyena_filter_params* p, float* k, float* saved, void* stream);

/* Gradients of the parameters for upstream gradient dk (D, L).  `saved` is what hyena_filter_fwd filled for the same
 * parameters.  Reductions over L are deterministic (fixed-order partial sums, no atomics). */
int hyena_filter_bwd(const hyena_filter_params* p, const float* dk, const float* saved, const hyena_filter_grads* g,
                     void* workspace, size_t workspace_bytes, void* stream);

/* ---- the same filter as the reference computes it under torch.autocast(dtype), dtype = HYENA_BF16 or HYENA_F16 (hyena_fftconv.h) ----
 * Parameters are passed in fp32 exactly as above (autocast casts them per call: hyena.py:199-215 run under the trainer's autocast
 * context); k is fp32 (the modulation promotes to fp32, hyena.py:152-155).  `saved` holds the three pre-activations as 16-bit pairs. */

/* Bytes of the pre-activation buffer hyena_filter16_fwd fills for the backward (3 x 32 rows of hyena_filter_row_pitch(L) 16-bit pairs). */
size_t hyena_filter16_saved_bytes(int L);

/* k (D, L) fp32 <- filter.  `saved` may be NULL (inference). */
int hyena_filter16_fwd(const hyena_filter_params* p, int dtype, float* k, void* saved, void* stream);

/* Gradients (fp32: the sums over L are kept in fp32, the reference rounds them to `dtype` once more).  Workspace as for hyena_filter_bwd
 * (hyena_filter_workspace_bytes). */
int hyena_filter16_bwd(const hyena_filter_params* p, int dtype, const float* dk, const void* saved, const hyena_filter_grads* g,
                       void* workspace, size_t workspace_bytes, void* stream);

/* ---- pitched filter rows (round 5) -------------------------------------------------------------------------------------------------
 * The reference's trainer asks for L = max_length - 1 taps (hg38_dataset.py:220-223 -> hyena.py:389-394: l_filter = min(L, l_max)): odd, so the
 * rows of a packed (D, L) fp32 tensor are 4-byte but not 16-byte aligned.  `ldk` = floats between the starts of consecutive rows of k / dk
 * (ldk >= L; elements [L, ldk) of a row are never read or written) -- the same pitch hyena_fftconv_fwd_ld / _bwd_ld take for them.  The entry
 * points above are these with ldk = L.  The library-owned buffers (`saved`, the backward's workspace) are pitched internally to
 * hyena_filter_row_pitch(L) = L rounded up to 64 words: their sizes are what hyena_filter*_saved_bytes / _workspace_bytes return. */
int hyena_filter_row_pitch(int L);
int hyena_filter_fwd_ld(const hyena_filter_params* p, float* k, int ldk, float* saved, void* stream);
int hyena_filter_bwd_ld(const hyena_filter_params* p, const float* dk, int ldk, const float* saved, const hyena_filter_grads* g,
                        void* workspace, size_t workspace_bytes, void* stream);
int hyena_filter16_fwd_ld(const hyena_filter_params* p, int dtype, float* k, int ldk, void* saved, void* stream);
int hyena_filter16_bwd_ld(const hyena_filter_params* p, int dtype, const float* dk, int ldk, const void* saved, const hyena_filter_grads* g,
                          void* workspace, size_t workspace_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* HYENA_FILTER_H */
