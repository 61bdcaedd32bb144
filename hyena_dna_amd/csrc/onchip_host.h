// onchip_host.h -- host-side interface of the workspace-free path (onchip.hip) used by the C ABI in fftconv.hip.
#pragma once
#include <stddef.h>

namespace hyena {
namespace oc {

enum { MAX_L = 32768 };

// Row pitch (elements between the starts of consecutive rows, >= L) of the activation tensors (u, out, dout, du: row (b, d) starts at
// element (b D + d) ldx) and of the filter-side tensors (k, dk: row d starts at element d ldk).  Packed tensors: ldx = ldk = L.
struct Pitch { int ldx, ldk; };

// R = M / 1024 (1, 2, 4, ..., 32) for a sequence length this path serves, 0 otherwise
int plan_r(int L);
// twiddle tables: number of complex64 entries, and their construction on the host (double precision)
size_t table_entries(int R);
size_t set_entries(int R);            // one table set [tw1 | tw2] of transform size 1024 R
void build_tables(int R, float* host_c32);
// device memory for the filter spectrum H [D][M] (the only intermediate of this path)
size_t spectrum_bytes(int D, int R);

// H = (FFT(k) + bias) / M into `H`
int launch_spec(int R, const float* k, const float* bias, void* H, const void* tab, int D, int L, Pitch ld, void* stream);
// out = conv(x, H) (conj = 0) or corr(x, H) (conj = 1)
int launch_conv(int R, const void* x, void* out, const void* H, const void* tab, int B, int D, int L, Pitch ld, int dtype, int conj,
                void* stream);
// dk (and dbias) from dout and u; `partials` = dk_partial_bytes(...) bytes of scratch (may be null when that is 0)
size_t dk_partial_bytes(int R, int B, int D, int L);
int dk_slices(int R, int B, int D, int* nb_out);
int launch_dk(int R, const void* dout, const void* u, float* dk, float* dbias, void* partials, const void* tab, int B, int D, int L,
              Pitch ld, int dtype, void* stream);

// du AND dk from one launch (round 6, M <= 16384, B >= 2): dk_kernel also multiplies its transform of dout by conj(H) and inverts it -- dout is
// transformed once instead of twice.  dkdu_ok: the calls it takes (default: M = 16384, where it wins 12 %; HYENA_FFTCONV_DUDK=1 / 0: every M <= 16384 /
// never; profiles/r6_dudk_ab.txt)
bool dkdu_ok(int R, int B);
int launch_dkdu(int R, const void* dout, const void* u, void* du, const void* H, float* dk, float* dbias, void* partials, const void* tab, int B,
                int D, int L, Pitch ld, int dtype, void* stream);

// dk at B = 1, M = 32768: spectrum of the u rows into `Uspec` (spectrum_bytes(D, R) bytes of scratch), then the convolution kernel with
// the conjugate and fp32 output rows -- nothing to accumulate over, so dk_kernel's shape (two spectra + an accumulator in registers) buys nothing
bool dk1_ok(int R, int B);
int launch_dk1(int R, const void* dout, const void* u, float* dk, float* dbias, void* Uspec, const void* tab, int D, int L, Pitch ld,
               int dtype, void* stream);

// Short rows with a small batch (R <= 2, B <= 2 row groups' worth): one launch per direction.  small_ok says whether the pair of
// fused kernels serves this call; launch_small_bwd needs the forward's filter spectrum H (the saved-spectrum buffer).
bool small_ok(int R, int B, int D, int L, Pitch ld, int dtype);
int launch_small_fwd(int R, const void* x, void* out, const float* k, const float* bias, void* Hout, const void* tab, int B, int D, int L,
                     Pitch ld, int dtype, void* stream);
int launch_small_bwd(int R, const void* dout, const void* u, void* du, float* dk, float* dbias, const void* H, const void* tab, int B, int D,
                     int L, Pitch ld, int dtype, void* stream);

}  // namespace oc
}  // namespace hyena
