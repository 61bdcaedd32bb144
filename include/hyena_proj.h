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
/* ... on STRIDED / PITCHED outputs (round 5; hyena_fftconv.h, hyena_fftconv_fwd_ld, says why; hyena_mixer.h, hyena_cm_*_ld, defines the layouts):
 * xT row (c, b) at element c csx + b bsx (bsx >= Lx, (B - 1) bsx + Lx <= csx < 2^31); vg row (b, d) at (b D + d) ldv (ldv >= Lc).  The kernel
 * walks the flattened positions in 64-position tiles: with csx a multiple of 8 and bsx = Lx every xT store is 16-byte aligned.  The entry
 * point above = the packed strides (csx = B Lx, bsx = Lx, ldv = Lc). */
int hyena_inproj_pre_fwd_ld(const void* u, const void* W, const float* bin, const float* w, const float* b, void* xT, void* vg,
                            int B, int Lx, int Lc, int D, long csx, int bsx, int ldv, int dtype, void* stream);


/* ---- out_proj with the second gate on its operand load (round 4) -----------------------------------------------------------------
 * Replaces, for the same operator configuration, these lines of HyenaOperator.forward in ONE launch (until round 4:
 * hyena_cm_post_fwd + a library GEMM):
 *     y = y * x[0]                                      hyena.py:432   (x[0] = the first third of short_filter(in_proj(u)))
 *     y = rearrange(y, 'b d l -> b l d')                hyena.py:439
 *     y = self.out_proj(y)                              hyena.py:440   nn.Linear(D, D)
 * Tensors: y (B, D, L) the long convolution's output; xT (3D, B, Lx) as above (rows [0, D) are read, bin / w / b likewise);
 * W (D, D) out_proj weight, bias (D,) fp32 [values already rounded to the element type] or NULL;
 * out (B, L, D);  zT (D, B, L) = y * x0 or NULL -- written only if the caller keeps it for the weight gradient (bit-identical to
 * hyena_cm_post_fwd).  64 <= L <= Lx (hyena_outproj_supported; any such L: a sequence's last tile is pulled back to end at L and rewrites up to
 * 63 positions identically; shorter sequences keep hyena_cm_post_fwd + the library GEMM).  Arithmetic: z = round(y * shortconv(xT + bin)) exactly as hyena_cm_post_fwd, then 16-bit operands with
 * fp32 accumulation on v_mfma_f32_32x32x16, out = one rounding of (sum + bias).  Asynchronous on `stream`; no workspace; no state. */
int hyena_outproj_supported(int B, int L, int D, int dtype);
/* Which generation of a kernel family the entry points launch.  family 1 = hyena_inproj_pre_fwd above: 1 = rounds 3 / 4's kernel (a wavefront holds 16
 * channels of each of x0 / x1 / v, D[channel][position] on v_mfma_f32_16x16x32, 48 two-byte LDS writes per tile; the default), 2 = round 6's (32 channels of
 * one group per wavefront on v_mfma_f32_32x32x16 with the operand tile as A -- a lane holds 16 consecutive positions of a channel: four 16-byte LDS
 * writes per tile --, 12 wavefronts per workgroup, three per SIMD, the previous tile's row phase issued between the matrix instructions; same values bit
 * for bit; measured 9 % faster at 32768 x 8 and 3 - 9 % SLOWER from 160000 x 2 up: selectable, not the default -- profiles/r6_inproj_gen2_not_kept.txt).  family 0 = the out_proj forward below: 1 = round 4's kernel (64 output channels per
 * wavefront held in 128 registers, two wavefronts per SIMD, the operand rows of a tile fetched in four exposed batches), 2 = round 6's (default: 16
 * output channels per wavefront, d_model / 16 wavefronts per workgroup, operand rows prefetched one tile ahead, whole-row epilogue; same values --
 * zT bit for bit, out to the summation order of the matrix cores; with the add + LayerNorm epilogue bit for bit what the two separate launches give).
 * generation <= 0 queries; returns the generation in use, -1 for an unknown family / generation.  Process-wide (A/B measurements, tests of both). */
int hyena_proj_kernel_generation(int family, int generation);
int hyena_outproj_gate_fwd(const void* y, const void* xT, const float* bin, const float* w, const float* b, const void* W,
                           const float* bias, void* out, void* zT, int B, int L, int Lx, int D, int dtype, void* stream);
/* ... on STRIDED / PITCHED operands: xT row (c, b) at c csx + b bsx, zT row (d, b) at d csz + b bsz, y row (b, d) at (b D + d) lda (layouts:
 * hyena_mixer.h, hyena_cm_*_ld).  The entry point above = the packed strides. */
int hyena_outproj_gate_fwd_ld(const void* y, const void* xT, const float* bin, const float* w, const float* b, const void* W,
                              const float* bias, void* out, void* zT, int B, int L, int Lx, int D, long csx, int bsx, long csz, int bsz,
                              int lda, int dtype, void* stream);
/* ... with the block's residual add + LayerNorm in its epilogue (round 5).  In a prenorm block (flash_attn Block =
 * src/models/sequence/simple_lm.py:262-284, long_conv_lm.py:381-396) the mixer's output goes straight into
 *     residual' = dropout(mixer_out) + residual;   hidden = LayerNorm(residual')          (dropout p = 0 in every HyenaDNA configuration)
 * A 64 x d_model output tile holds whole rows, so the kernel forms both from its accumulators and the out_proj output itself is never
 * written (nor read back by hyena_add_norm_fwd: 1 GB of traffic and one launch per layer at L = 2^20, d_model 256):
 *     out (B, L, D) 16-bit = LayerNorm(residual'; ln_weight, ln_bias, eps);  residual_out (B L, D) fp32 = round16(z W^T + bias) + residual_in;
 *     mean, rstd (B L,) fp32 for the backward (hyena_add_norm_bwd).
 * Values: exactly those of hyena_outproj_gate_fwd_ld followed by hyena_add_norm_fwd (same roundings, same summation order) -- bit for bit.
 * residual_in: (B L, D) fp32 or NULL (no residual yet); must not alias residual_out.  ln_weight == NULL: plain hyena_outproj_gate_fwd_ld. */
int hyena_outproj_gate_addnorm_fwd_ld(const void* y, const void* xT, const float* bin, const float* w, const float* b, const void* W,
                                      const float* bias, const float* residual_in, const float* ln_weight, const float* ln_bias, float eps,
                                      void* out, float* residual_out, float* mean, float* rstd, void* zT, int B, int L, int Lx, int D,
                                      long csx, int bsx, long csz, int bsz, int lda, int dtype, void* stream);


/* ---- out_proj's input gradient with the second gate's backward in its epilogue (round 5) ---------------------------------------------
 * The backward of the lines hyena_outproj_gate_fwd replaces (hyena.py:432-440) up to the long convolution, in ONE launch (until round 5: a
 * library GEMM writing dz^T = W_out^T dy^T, then hyena_cm_post_bwd reading it back):
 *     dz^T = round16(W_out^T dy^T)                     (never written)
 *     dyc[b, d, l] = dz * x0c                          gradient of the long convolution's output          (B, D, L), row pitch lda
 *     dxT[d, b, m] = w2 g[m] + w1 g[m + 1] + w0 g[m + 2],  g = dz * y      rows [0, D) of (3D, B, Lx), positions < L, xT's layout (csx, bsx)
 *     part[d][run][8] = per-run partial sums of (dw0, dw1, dw2, db_sc, db_in) of the x0 channels' short filter / in_proj bias;
 *                       hyena_outproj_dgrad_partial_floats(B, L, D) floats = D x runs x 8; summing axis 1 gives the gradients (fixed order).
 * dy (B L, D) 16-bit; Wt (D, D) = out_proj.weight TRANSPOSED, contiguous; y, xT, bin, w, b as for hyena_outproj_gate_fwd_ld.
 * Values: those of the unfused pair up to the summation order inside the product (dz^T within one 16-bit ulp of the library GEMM's, identical
 * almost everywhere); given the same dz^T, dyc and dxT are hyena_cm_post_bwd's bits.  D in {128, 256}, 16-bit types, any L >= 1. */
int hyena_outproj_dgrad_supported(int B, int L, int D, int dtype);
size_t hyena_outproj_dgrad_partial_floats(int B, int L, int D);
int hyena_outproj_dgrad_gate_bwd_ld(const void* dy, const void* Wt, const void* y, const void* xT, const float* bin, const float* w,
                                    const float* b, void* dyc, void* dxT, float* part, int B, int L, int Lx, int D, long csx, int bsx,
                                    int lda, int dtype, void* stream);


/* ---- the block's MLP (flash_attn.modules.mlp.Mlp = simple_lm.py:191-211; long_conv_lm.py:117-123: fc1 -> tanh-GELU -> fc2) ---------
 * The two products contracting over d_model, position-major, with the reference's element-wise passes in their epilogues:
 *   forward   a = x W1^T + b1,  h = gelu_tanh(a)          x (P, K), W1 (N, K), b1 (N,) fp32 or NULL [values already rounded to the
 *                                                          element type, as autocast rounds the bias] -> a, h (P, N), both stored
 *                                                          (a is what the backward needs, h is fc2's input)
 *   backward  da = (dy W2) * gelu_tanh'(a)                 dy (P, K), W2T (N, K) = fc2.weight^T, a (P, N) -> da (P, N) and
 *             part[runs][N] = per-run column sums of da    (summing axis 0 gives d b1; hyena_mlp_partial_floats(P, N) floats)
 * K = d_model in {128, 256}, N = d_inner a multiple of 256, 16-bit element types, P < 2^31 positions (B L).  fc2 itself, the
 * input gradient da W1 and the weight gradients contract over d_inner or over the positions and stay library GEMMs.
 * Numerics: fp32 accumulation; a = one rounding of (sum + b1); h = round(gelu(a)) from the rounded a; da = round(round(dy W2) gelu'(a))
 * -- the values the autocast graph of the reference produces, up to the summation order inside the products. */
int hyena_mlp_supported(long P, int K, int N, int dtype);
size_t hyena_mlp_partial_floats(long P, int N);
int hyena_mlp_fc1_gelu_fwd(const void* x, const void* W1, const float* b1, void* a, void* h, long P, int K, int N, int dtype, void* stream);
int hyena_mlp_dh_dgelu_bwd(const void* dy, const void* W2T, const void* a, void* da, float* part, long P, int K, int N, int dtype,
                           void* stream);

/* ---- bias gradient of a linear layer: out[n] = sum over p of x[p, n], fp32 ------------------------------------------------------
 * x (P, N) 16-bit position-major (the gradient arriving at out_proj / fc2: hyena.py:440, simple_lm.py:210); N a multiple of 8 with
 * N / 8 dividing 256 (128, 256, 512, 1024 ...); part = hyena_colsum_partial_floats(P, N) floats of scratch.  One pass at the memory
 * rate, two-stage fixed-order sums (bitwise reproducible). */
int hyena_colsum_supported(long P, int N, int dtype);
size_t hyena_colsum_partial_floats(long P, int N);
int hyena_colsum(const void* x, float* part, float* out, long P, int N, int dtype, void* stream);

#ifdef __cplusplus
}
#endif
#endif
