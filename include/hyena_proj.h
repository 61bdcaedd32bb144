/* hyena_proj.h -- C ABI of the operator's input projection on the matrix cores with the front of the element-wise shell in its
 * epilogue (same library, libhyena_fftconv.so; kernels in hyena_dna_amd/csrc/proj_kernels.h).
 *
 * Replaces, for the HyenaDNA operator configuration (order 2, short_filter_order 3, d_model 128 or 256, 16-bit autocast), these
 * lines of HyenaOperator.forward (src/models/sequence/hyena.py) in ONE launch:
 *
 *     u  = self.in_proj(u)                                        hyena.py:391   nn.Linear(D, 3D)   [bias kept aside, see below]
 *     u  = rearrange(u, 'b l d -> b d l')                         hyena.py:392
 *     uc = self.short_filter(u)[..., :l_filter]                   hyena.py:394   nn.Conv1d(3D, 3D, 3, groups=3D, padding=2)
 *     *x, v = uc.split(d_model, dim=1);  v = v * x[1]             hyena.py:404, 420
 *
 * Tensors (row-major, contiguous; `dtype` HYENA_BF16 or HYENA_F16 as in hyena_fftconv.h):
 *   u   : (B, Lx, D)     W : (3D, D) same element type (nn.Linear weight, rows [0,D) = x0, [D,2D) = x1, [2D,3D) = v)
 *   bin : (3D,) fp32 in_proj bias or NULL;   w : (3D, 3) fp32 short-filter taps;   b : (3D,) fp32 short-filter bias
 *   xT  : (3D, B, Lx)  = W u^T WITHOUT the bias -- the tensor hyena_cm_post_fwd / hyena_cm_*_bwd (hyena_mixer.h) read
 *   vg  : (B, D, Lc)   = what hyena_cm_pre_fwd computes from that xT (bit-identical), Lc = min(Lx, l_max) <= Lx
 * Arithmetic: 16-bit operands, fp32 accumulation on v_mfma_f32_32x32x16_{bf16,f16}; xT rounded once; the short convolution and
 * the gate in fp32 on the rounded xT.  Asynchronous on `stream`; no workspace; no state.
 * Returns HYENA_OK or a HYENA_ERR_* code of hyena_fftconv.h (HYENA_ERR_BAD_ARG for shapes hyena_proj_supported rejects).
 */
#ifndef HYENA_PROJ_H
#define HYENA_PROJ_H
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif

/* 1 if hyena_inproj_pre_fwd serves this width / element type / size (D in {128, 256}, 16-bit, 8 <= Lx, B Lx < 2^31) */
int hyena_proj_supported(int B, int Lx, int D, int dtype);
int hyena_inproj_pre_fwd(const void* u, const void* W, const float* bin, const float* w, const float* b, void* xT, void* vg,
                         int B, int Lx, int Lc, int D, int dtype, void* stream);

#ifdef __cplusplus
}
#endif
#endif
