/* hyena_block.h -- C ABI of the fused residual glue of a HyenaDNA block (libhyena_fftconv.so).
 *
 * Replaces what the reference's prenorm block does around the mixer and the MLP,
 *     dropped  = drop_path(dropout(hidden_states))                      src/models/sequence/simple_lm.py:267 / 280
 *     residual = dropped + residual                                     simple_lm.py:268 / 281
 *     hidden_states = norm(residual.to(norm.weight.dtype))              simple_lm.py:269 / 282
 *     residual = residual.to(torch.float32)   (residual_in_fp32)        simple_lm.py:270-271 / 283-284
 * and the final norm of the backbone (src/models/sequence/long_conv_lm.py:381-396), i.e. the operation the reference
 * delegates to flash_attn.ops.layer_norm.dropout_add_layer_norm when that CUDA-only package is installed
 * (long_conv_lm.py:31,387), for dropout probability 0 (HyenaDNA's resid_dropout; a caller-side dropout covers p > 0).
 *
 * Tensors: x0 / out / dout / dx0 (rows, D) elements of their dtype code (HYENA_F32 / HYENA_BF16 / HYENA_F16 of
 * hyena_fftconv.h); residual, residual', their gradients, weight, bias, mean, rstd fp32.  D a multiple of 64, <= 1024.
 * Nothing is allocated or synchronised inside; `stream` is a hipStream_t.
 */
#ifndef HYENA_BLOCK_H
#define HYENA_BLOCK_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 1 if (D, dtypes) are covered. */
int hyena_add_norm_supported(int D, int x_dtype, int out_dtype);

/* residual_out = x0 + residual_in (residual_in may be NULL: residual_out = x0);  out = LayerNorm(residual_out).
 * mean / rstd (rows,) are kept for the backward. */
int hyena_add_norm_fwd(const void* x0, int x_dtype, const float* residual_in, const float* weight, const float* bias, float eps,
                       void* out, int out_dtype, float* residual_out, float* mean, float* rstd, long rows, int D, void* stream);

/* Floats of scratch the backward needs for its per-workgroup partial weight / bias gradients. */
size_t hyena_add_norm_partial_floats(long rows, int D);

/* Given dout (gradient of out) and d_residual_out (gradient flowing into residual_out from downstream, may be NULL):
 *   dx0 (dtype dx_dtype) and d_residual_in (fp32, may be NULL) = the gradient of residual_out's sum,
 *   dweight, dbias (D,) reduced over the rows in a fixed order (deterministic). */
int hyena_add_norm_bwd(const void* dout, int dout_dtype, const float* d_residual_out, const float* residual_out,
                       const float* weight, const float* mean, const float* rstd, void* dx0, int dx_dtype,
                       float* d_residual_in, float* dweight, float* dbias, float* partial, long rows, int D, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* HYENA_BLOCK_H */
