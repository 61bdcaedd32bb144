/* hyena_fftconv.h -- C ABI of the MI355X-native Hyena long-convolution library (libhyena_fftconv.so).
 *
 * This is the native FFI seam of the hot path.  It replaces, for every HyenaDNA configuration, what the
 * reference binds through its pybind11 module `fftconv`:
 *
 *   reference                                                   this library
 *   ---------------------------------------------------------   -----------------------------------
 *   fftconv_fwd(u, filter, D, v, head_dim, q, dropout_mask,      hyena_fftconv_fwd
 *       gelu, gelu_inp, gelu_q, fft_size, force_fp16_output,
 *       output_hbl_layout, fftfp16)
 *       csrc/fftconv/fftconv.cpp:53-132, called from
 *       src/ops/fftconv.py:84  (FFTConvFunc.forward)
 *   fftconv_bwd(dout, u, filter, D, v, head_dim, q, ...)         hyena_fftconv_bwd
 *       csrc/fftconv/fftconv.cpp:134-236, called from
 *       src/ops/fftconv.py:96  (FFTConvFunc.backward)
 *
 * and, semantically, the pure-PyTorch function every shipped config actually runs:
 *   fftconv_ref(u, k, D, dropout_mask=None, gelu=False)          src/models/sequence/hyena.py:59-88
 *       out = irfft(rfft(u, 2L) * rfft(k, 2L) / 2L, norm="forward")[..., :L] + u * D[..., None]
 *
 * Differences from the reference's native seam (all of them lifts of its restrictions):
 *   - the filter is passed in the TIME domain (`k`, fp32, (D, L)); the library owns the spectrum
 *     (the reference makes the caller run cuFFT first: src/ops/fftconv.py:65);
 *   - any L >= 1 (odd allowed; the reference requires even L and fft_size <= 16384, fftconv.cpp:114-115);
 *     the implementation supports L <= 1,048,576;
 *   - the backward returns dk in the time domain (the reference returns dk_f and the caller runs
 *     irfft: src/ops/fftconv.py:98) and reduces dk / dbias over the batch itself, deterministically
 *     (the reference returns per-batch partials and lets the caller .sum(): fftconv.cpp:209-210,235);
 *   - plain pointers, sizes and a hipStream_t: no torch types.  All device pointers must belong to the
 *     device that is current for `stream`.  Nothing is allocated, freed or synchronised inside the
 *     compute entry points (they are hipGraph-capturable); the caller provides the workspace and the
 *     twiddle tables.
 *   - options of the reference op that no HyenaDNA config enables (gelu, dropout_mask, head_dim = 8, q, v,
 *     k_rev / bidirectional) are not part of this ABI.
 *
 * Tensors (row-major, contiguous):
 *   u, out, dout, du : (B, D, L) elements of `dtype`
 *   k, dk            : (D, L)  fp32
 *   bias, dbias      : (D,)    fp32      (the reference calls this argument `D`)
 *
 * All FFT arithmetic is fp32 for every dtype, as in the reference (hyena.py:75: u.to(k.dtype)).
 */
#ifndef HYENA_FFTCONV_H
#define HYENA_FFTCONV_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* element types of u / out / dout / du */
enum { HYENA_F32 = 0, HYENA_BF16 = 1, HYENA_F16 = 2 };

/* status codes (0 = success) */
enum {
    HYENA_OK = 0,
    HYENA_ERR_BAD_ARG = 1,        /* null pointer, non-positive size, unknown dtype */
    HYENA_ERR_UNSUPPORTED_L = 2,  /* L > HYENA_MAX_L */
    HYENA_ERR_WORKSPACE = 3,      /* workspace too small for the requested chunking */
    HYENA_ERR_LAUNCH = 4          /* a kernel launch failed (hipGetLastError() != hipSuccess) */
};

#define HYENA_MAX_L 1048576

/* ABI version of this header (bumped on any signature change). */
int hyena_fftconv_abi_version(void);

/* Human-readable text for a status code. */
const char* hyena_fftconv_error_string(int status);

/* Two execution plans, chosen by L alone:
 *   L <= 32768   workspace-free: one workgroup per (b, d) row holds the whole transform in registers and LDS
 *                (M = 1024, 2048, ..., 32768, the smallest power of two >= L); HBM traffic is u, out and the filter
 *                spectrum H [D][M] (the workspace; the backward adds S D L floats of per-slice dk rows when D is below the
 *                number of CUs and the batch of a channel is split over S workgroups).  The reference's own fused kernel has
 *                this shape but stops at L = 8192 (csrc/fftconv/fftconv_cuda.cu:805, fftconv.cpp:114-115).  HYENA_FFTCONV_ONCHIP=0 in the environment
 *                routes these lengths to the two-level plan instead (testing / profiling).
 *   L >  32768   two-level (four-step) transform through a workspace, M = M1 x 1024.
 *
 * Complex transform length M used for sequence length L: the padded real transform has N = 2M >= 2L
 * points, M = M1 * 1024 with M1 the smallest supported column size >= L / 1024: the powers of two up to 1024 and
 * 2^a x {3, 5, 7} (3, 5, 6, 7, 10, 12, 14, 20, 24, 28, 96, 160, 192, 224, 320, 384, 448, 640, 768), so that the
 * zero padding beyond 2L stays below ~20 % for most lengths (hyenadna-medium-160k: L = 160000 -> M1 = 160, N = 327680;
 * -450k: L = 450560 -> M1 = 448, N = 917504); the reference uses N = 2L (hyena.py:61), which gives the same causal
 * result.  Returns 0 if L is unsupported. */
int hyena_fftconv_fft_size(int L);

/* Which plan serves sequence length L (tables built for one plan are not valid for the other). */
enum { HYENA_PLAN_NONE = 0, HYENA_PLAN_ONCHIP = 1, HYENA_PLAN_TWO_LEVEL = 2 };
int hyena_fftconv_plan(int L);

/* Bytes of device memory needed for the twiddle tables of sequence length L, and their one-time
 * initialisation: the host computes them in double precision and copies them on `stream` (the stream the compute
 * entry points will be given), then waits for that copy -- the ONE synchronising call of this ABI.  It must not
 * run inside a hipGraph capture: initialise the tables of every length you will use before capturing.
 * The same tables serve every call with the same hyena_fftconv_fft_size(L) and hyena_fftconv_plan(L). */
size_t hyena_fftconv_table_bytes(int L);
int hyena_fftconv_init_tables(void* d_tables, int L, void* stream);

/* Channels processed per pass through the kernel chain ("chunk").  Intermediate spectra of one chunk
 * live in the workspace; the default takes as many channels as an 8 GiB workspace holds (all of them at
 * every HyenaDNA size): launches are few and large, which measured faster on MI355X than Infinity-Cache-sized
 * chunks.  backward = 0 for hyena_fftconv_fwd, 1 for hyena_fftconv_bwd. */
int hyena_fftconv_default_chunk(int B, int D, int L, int backward);

/* Workspace bytes for the given problem and chunk (chunk <= 0 selects the default). */
size_t hyena_fftconv_workspace_bytes(int B, int D, int L, int backward, int chunk);

/* out[b,d,:] = causal_conv(u[b,d,:], k[d,:]) + bias[d] * u[b,d,:]          (hyena.py:59-88)
 * bias may be NULL (treated as 0).  `stream` is a hipStream_t. */
int hyena_fftconv_fwd(const void* u, const float* k, const float* bias, void* out,
                      int B, int D, int L, int dtype,
                      const void* d_tables, void* workspace, size_t workspace_bytes, int chunk,
                      void* stream);

/* Gradients of the above for upstream gradient dout:
 *   du[b,d,s]   = sum_{t>=s} dout[b,d,t] k[d,t-s] + bias[d] dout[b,d,s]
 *   dk[d,s]     = sum_b sum_{t>=s} dout[b,d,t] u[b,d,t-s]
 *   dbias[d]    = sum_b sum_t dout[b,d,t] u[b,d,t]
 * du / dk / dbias may each be NULL to skip that output (dbias requires dk's computation and is free with it). */
int hyena_fftconv_bwd(const void* dout, const void* u, const float* k, const float* bias,
                      void* du, float* dk, float* dbias,
                      int B, int D, int L, int dtype,
                      const void* d_tables, void* workspace, size_t workspace_bytes, int chunk,
                      void* stream);

/* Optional time-for-memory trade (what the reference's autograd does by keeping u_f, src/ops/fftconv.py:78 keeps
 * k_f): the forward can leave the column-transformed filter and activations in a caller-owned buffer
 *     saved = Wk [D][M] | Wu [B][D][M]   complex64,  hyena_fftconv_saved_bytes(B, D, L) bytes
 * and the backward then skips re-reading u and k and their column transforms (22 of ~150 MB of traffic per row at
 * L = 2^20).  Results are bitwise those of the plain entry points.  The buffer must stay untouched in between.
 * For L <= 32768 (the workspace-free path, below) the buffer holds the filter spectrum H [D][M] only and the backward
 * still needs `u` when dk is requested (it re-transforms u on chip); for longer sequences `u` may be NULL. */
size_t hyena_fftconv_saved_bytes(int B, int D, int L);
int hyena_fftconv_fwd_save(const void* u, const float* k, const float* bias, void* out,
                           int B, int D, int L, int dtype,
                           const void* d_tables, void* workspace, size_t workspace_bytes, int chunk,
                           void* saved, size_t saved_bytes, void* stream);
int hyena_fftconv_bwd_saved(const void* dout, const void* u, const float* bias, void* du, float* dk, float* dbias,
                            int B, int D, int L, int dtype,
                            const void* d_tables, void* workspace, size_t workspace_bytes, int chunk,
                            const void* saved, size_t saved_bytes, void* stream);

/* The same two operations on PITCHED rows (round 5).  The reference's trainer feeds the operator L = max_length - 1 positions
 * (src/dataloaders/datasets/hg38_dataset.py:220-223: `data = seq[:-1]`) -- 32 767, 159 999, 449 999, 999 999, 1 048 575: odd, so in a packed
 * (B, D, L) tensor every second row starts 2 bytes off a 4-byte boundary and no row but the first is 16-byte aligned.  Here the caller
 * states where rows start instead:
 *     ldx : elements between the starts of consecutive rows of u / out / dout / du -- row (b, d) starts at element (b D + d) ldx
 *     ldk : the same for k / dk (fp32 elements) -- row d starts at element d ldk
 * ldx, ldk >= L; elements [L, ld) of a row are never written and never enter any arithmetic (for odd L on the two-level plan the ONE element behind
 * a row's end is loaded together with the row's last sample and discarded -- it may hold anything, NaN included).  ldx = ldk = L is the packed layout of the entry points above (which
 * are these with that default).  hyena_dna_amd/_lib.py allocates the operator's channel-major tensors with ld = L rounded up to 64 elements
 * and hands PyTorch the [..., :L] views.
 * `saved` / `saved_bytes`: the optional spectrum buffer of hyena_fftconv_fwd_save / _bwd_saved, or NULL / 0.  With `saved` the backward does
 * not read k (and, for L > 32768, not u): they may be NULL then. */
int hyena_fftconv_fwd_ld(const void* u, const float* k, const float* bias, void* out,
                         int B, int D, int L, int ldx, int ldk, int dtype,
                         const void* d_tables, void* workspace, size_t workspace_bytes, int chunk,
                         void* saved, size_t saved_bytes, void* stream);
int hyena_fftconv_bwd_ld(const void* dout, const void* u, const float* k, const float* bias,
                         void* du, float* dk, float* dbias,
                         int B, int D, int L, int ldx, int ldk, int dtype,
                         const void* d_tables, void* workspace, size_t workspace_bytes, int chunk,
                         const void* saved, size_t saved_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* HYENA_FFTCONV_H */
