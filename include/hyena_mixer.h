/* hyena_mixer.h -- C ABI of the fused element-wise shell of the Hyena operator (same library, libhyena_fftconv.so).
 *
 * Replaces, for the HyenaDNA operator configuration (order 2, one head, one block, inner factor 1, dropout 0,
 * activation "id", short_filter_order 3), these lines of HyenaOperator.forward
 * (src/models/sequence/hyena.py:392-439) and their autograd:
 *
 *     u  = rearrange(in_proj(u), 'b l d -> b d l')                     hyena.py:391-392
 *     uc = self.short_filter(u)[..., :l_filter]                        hyena.py:394     nn.Conv1d(3D, 3D, 3, groups=3D, padding=2)
 *     *x, v = uc.split(d_model, dim=2)                                 hyena.py:396-404
 *     v = v * x[1]                                                     hyena.py:420     -> hyena_mixer_pre_fwd
 *     v = fftconv(v, k, bias)                                          hyena.py:423     (hyena_fftconv_*)
 *     y = rearrange(v * x[0], 'b h v z l -> b (z l) (h v)')            hyena.py:432-439 -> hyena_mixer_post_fwd
 *
 * Tensors (row-major, contiguous), `dtype` as in hyena_fftconv.h (HYENA_F32 / HYENA_BF16 / HYENA_F16):
 *   x   : (B, Lx, 3D)  output of in_proj; channels [0,D) = x0, [D,2D) = x1, [2D,3D) = v
 *   w   : (3D, 3) fp32 short-filter taps (Conv1d weight (3D,1,3));  b : (3D,) fp32
 *   vg, y, dy, dvg : (B, D, L);   z, dz : (B, L, D);   L = min(Lx, l_max) positions are processed
 *   dx  : (B, Lx, 3D); positions >= L are NOT written (zero-fill them when Lx > L)
 *   part: hyena_mixer_partial_floats(B, L, D) floats of scratch, laid out [B][runs][3D][4] = per-run partial sums of
 *         (dw[c][0], dw[c][1], dw[c][2], db[c]); summing over the first two axes gives the short filter's gradients
 *         (post_bwd fills channels [0,D), pre_bwd [D,3D)).  No atomics: results are deterministic.
 * Arithmetic is fp32; 16-bit tensors are rounded once on store.  All entry points are asynchronous on `stream`.
 */
#ifndef HYENA_MIXER_H
#define HYENA_MIXER_H
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif

/* vg[b,d,t] = xc[b,2D+d,t] * xc[b,D+d,t],  xc[c,t] = b[c] + w[c,0] x[t-2,c] + w[c,1] x[t-1,c] + w[c,2] x[t,c] */
int hyena_mixer_pre_fwd(const void* x, const float* w, const float* b, void* vg,
                        int B, int L, int Lx, int D, int dtype, void* stream);
/* z[b,t,d] = y[b,d,t] * xc[b,d,t] */
int hyena_mixer_post_fwd(const void* y, const void* x, const float* w, const float* b, void* z,
                         int B, int L, int Lx, int D, int dtype, void* stream);
size_t hyena_mixer_partial_floats(int B, int L, int D);
/* dy = dz^T * x0c;  dx[..., 0:D] and the partials of channels [0,D) from g = dz^T * y */
int hyena_mixer_post_bwd(const void* dz, const void* y, const void* x, const float* w, const float* b,
                         void* dy, void* dx, float* part, int B, int L, int Lx, int D, int dtype, void* stream);
/* dx[..., D:3D] and the partials of channels [D,3D) from dvg (gradient of vg) */
int hyena_mixer_pre_bwd(const void* dvg, const void* x, const float* w, const float* b,
                        void* dx, float* part, int B, int L, int Lx, int D, int dtype, void* stream);

/* ---- the same shell in CHANNEL-MAJOR layout (csrc/cm_kernels.h) -------------------------------------------------------
 * The projections produce / consume transposed tensors (the GEMM library takes transposed operands at no cost), so nothing
 * between in_proj and out_proj is ever rearranged ('b l d -> b d l' at hyena.py:392 and back at hyena.py:432-439 disappear):
 *   xT  : (3D, B, Lx)  = W_in u^T, WITHOUT the in_proj bias `bin` (3D, fp32; may be NULL) -- the kernels add it on load
 *   zT, dzT : (D, B, L);   dxT : (3D, B, Lx) (positions >= L are not written);   vg, y, dy, dvg : (B, D, L) as above
 *   part: hyena_cm_partial_floats(B, L, D) floats, [3D][B * tiles][8] = per-workgroup partial sums of
 *         (dw[c][0], dw[c][1], dw[c][2], db_sc[c], db_in[c]) in the first five of eight floats (the rest is padding, not
 *         written); summing axis 1 gives the gradients. */
size_t hyena_cm_partial_floats(int B, int L, int D);
int hyena_cm_pre_fwd(const void* xT, const float* bin, const float* w, const float* b, void* vg,
                     int B, int L, int Lx, int D, int dtype, void* stream);
int hyena_cm_post_fwd(const void* y, const void* xT, const float* bin, const float* w, const float* b, void* zT,
                      int B, int L, int Lx, int D, int dtype, void* stream);
int hyena_cm_post_bwd(const void* dzT, const void* y, const void* xT, const float* bin, const float* w, const float* b,
                      void* dy, void* dxT, float* part, int B, int L, int Lx, int D, int dtype, void* stream);
int hyena_cm_pre_bwd(const void* dvg, const void* xT, const float* bin, const float* w, const float* b,
                     void* dxT, float* part, int B, int L, int Lx, int D, int dtype, void* stream);

/* The same four kernels on STRIDED / PITCHED rows (round 5; see hyena_fftconv_fwd_ld in hyena_fftconv.h for why: the reference trainer's
 * L = max_length - 1 is odd, and rows of a packed tensor are then unaligned):
 *   csx, bsx : layout of xT / dxT -- row (c, b) starts at element c csx + b bsx  (bsx >= Lx, csx >= (B - 1) bsx + Lx)
 *   csz, bsz : layout of zT / dzT -- row (d, b) at d csz + b bsz                  (bsz >= L,  csz >= (B - 1) bsz + L)
 *              or BATCH-major rows: bsz >= (D - 1) csz + L, csz >= L -- the (B, D, L) layout of the convolution's tensors (csz = its row pitch,
 *              bsz = D csz): the gate between two long convolutions of an operator of order >= 3 (hyena.py:414-423) then writes the next
 *              convolution's input, and reads its gradient, in place (round 6)
 *   lda      : row pitch of the (B, D, L) tensors vg / y / dy / dvg -- row (b, d) at (b D + d) lda  (lda >= L)
 * Two layouts are in use (hyena_dna_amd/_lib.py): per-sequence pitch ld (cs = B ld, bs = ld: every row aligned; B = 1), and channel rows pitched
 * over the FLATTENED positions (cs = B L rounded up, bs = L: one library GEMM still sees one (C, B L) matrix; B > 1).  Elements outside a row's
 * length are never read or written.  The entry points above are these with the packed strides (cs = B len, bs = len, lda = L). */
int hyena_cm_pre_fwd_ld(const void* xT, const float* bin, const float* w, const float* b, void* vg,
                        int B, int L, int Lx, int D, long csx, int bsx, int lda, int dtype, void* stream);
int hyena_cm_post_fwd_ld(const void* y, const void* xT, const float* bin, const float* w, const float* b, void* zT,
                         int B, int L, int Lx, int D, long csx, int bsx, long csz, int bsz, int lda, int dtype, void* stream);
int hyena_cm_post_bwd_ld(const void* dzT, const void* y, const void* xT, const float* bin, const float* w, const float* b,
                         void* dy, void* dxT, float* part, int B, int L, int Lx, int D, long csx, int bsx, long csz, int bsz, int lda,
                         int dtype, void* stream);
int hyena_cm_pre_bwd_ld(const void* dvg, const void* xT, const float* bin, const float* w, const float* b,
                        void* dxT, float* part, int B, int L, int Lx, int D, long csx, int bsx, int lda, int dtype, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* HYENA_MIXER_H */
