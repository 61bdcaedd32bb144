// onchip.hip -- host side of the workspace-free long convolution for L <= 32768 (onchip_kernels.h).  Reached only
// through the C ABI of fftconv.hip; stateless like it (no allocation, no synchronisation, launches on the caller's stream).
#include "onchip_kernels.h"
#include "launch.h"
#include "onchip_host.h"
#include "../../include/hyena_fftconv.h"

#include <cmath>
#include <cstdlib>

#if defined(OC_PROFILE)
// profiling builds only: hand the device a buffer for conv_kernel's time stamps ([workgroup][wavefront][32] uint64)
extern "C" int hyena_oc_prof_set(void* buf) {
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(hyena::oc::oc_prof_buf), &buf, sizeof(buf));
}
#endif

namespace hyena {
namespace oc {

int plan_r(int L) {
    if (L < 1 || L > MAX_L) return 0;
    int r = 1;
    while (1024 * r < L) r *= 2;
    return r;
}

// one table set [tw1[11][T] | tw2[10][R]] of transform size 1024 R with input twist phi
void fill_set(int R, double phi, float* out) {
    const int T = 32 * R, M = 1024 * R;
    const double tau = 6.283185307179586476925286766559;
    for (int t = 0; t < T; ++t) {
        for (int b = 0; b < 8; ++b) {                        // tB[b] = w_M^(t (b + phi))
            const double a = -tau * ((double)t * ((double)b + phi)) / (double)M;
            out[2 * (b * T + t)] = (float)std::cos(a);
            out[2 * (b * T + t) + 1] = (float)std::sin(a);
        }
        for (int a3 = 1; a3 < 4; ++a3) {                     // tA[a] = w_M^(8 t a)
            const double a = -tau * (double)(((long)8 * t * a3) % M) / (double)M;
            out[2 * ((7 + a3) * T + t)] = (float)std::cos(a);
            out[2 * ((7 + a3) * T + t) + 1] = (float)std::sin(a);
        }
    }
    float* o2 = out + 2 * 11 * T;
    for (int tp = 0; tp < R; ++tp) {
        for (int b = 1; b < 8; ++b) {                        // tB[b] = w_T^(t' b)
            const double a = -tau * (double)(tp * b) / (double)T;
            o2[2 * ((b - 1) * R + tp)] = (float)std::cos(a);
            o2[2 * ((b - 1) * R + tp) + 1] = (float)std::sin(a);
        }
        for (int a3 = 1; a3 < 4; ++a3) {                     // tA[a] = w_T^(8 t' a)
            const double a = -tau * (double)((8 * tp * a3) % T) / (double)T;
            o2[2 * ((6 + a3) * R + tp)] = (float)std::cos(a);
            o2[2 * ((6 + a3) * R + tp) + 1] = (float)std::sin(a);
        }
    }
}
size_t set_entries(int R) { return (size_t)11 * 32 * R + (size_t)10 * R; }

// layout: the set of size R with phi = 1/4; for R = 32 additionally the two parity sets of size 16 (phi = 1/8, 5/8)
// that dk_kernel<16, 2> uses
size_t table_entries(int R) { return set_entries(R) + (R == 32 ? 2 * set_entries(16) : 0); }

void build_tables(int R, float* h) {
    fill_set(R, 0.25, h);
    if (R == 32) {
        fill_set(16, 0.125, h + 2 * set_entries(32));
        fill_set(16, 0.625, h + 2 * (set_entries(32) + set_entries(16)));
    }
}

size_t spectrum_bytes(int D, int R) { return (size_t)D * 1024 * R * sizeof(c32); }

// dk owns one workgroup (512 threads, the whole register file of a CU) per channel; with fewer channels than the MI355X has CUs
// the batch of a channel is cut into slices so that D S workgroups fill the chip.
static int dk_row_groups(int R) { const int bp = 512 / (32 * R); return R == 32 ? 1 : (bp > 16 ? 16 : (bp < 1 ? 1 : bp)); }
int dk_slices(int R, int B, int D, int* nb_out) {
    const int CUS = 256;
    const int iters = (B + dk_row_groups(R) - 1) / dk_row_groups(R);          // sequential steps of an unsliced workgroup
    int S = (CUS + D - 1) / D;
    if (S > iters) S = iters;
    if (S < 1) S = 1;
    int nb = (B + S - 1) / S;
    nb = (nb + dk_row_groups(R) - 1) / dk_row_groups(R) * dk_row_groups(R);    // whole steps per slice
    S = (B + nb - 1) / nb;
    if (nb_out) *nb_out = nb;
    return S;
}
size_t dk_partial_bytes(int R, int B, int D, int L) {
    const int S = dk_slices(R, B, D, nullptr);
    return S > 1 ? (size_t)S * D * L * sizeof(float) : 0;
}

template <int R, bool HALF>
static int spec_rh(const SpecArgs& a, void* stream) {
    typedef WgCfg<R> W;
    static thread_local int done = -1;
    hy_allow_lds(spec_kernel<R, HALF>, W::LDS, &done);
    HY_LAUNCH((spec_kernel<R, HALF>), dim3((a.D + W::RPW - 1) / W::RPW), dim3(W::WGT), W::LDS, stream, a);
    return hy_launch_error() ? HYENA_ERR_LAUNCH : HYENA_OK;
}
template <int R>
static int spec_r(const SpecArgs& a, void* stream) {
    return a.dtype == DT_F32 ? spec_rh<R, false>(a, stream) : spec_rh<R, true>(a, stream);
}
// dk at B = 1 (nothing to accumulate): dk = corr(dout, u) = the forward's two kernels -- spectrum of the u rows, then the convolution
// kernel with the conjugate and fp32 output rows.  Four wavefronts per SIMD instead of dk_kernel's two.
template <int R, bool HALF>
static int conv_f32out_rh(const ConvArgs& a, void* stream) {
    typedef WgCfg<R> W;
    static thread_local int done = -1;
    hy_allow_lds(conv_kernel<R, HALF, true>, W::LDS, &done);
    const int rows = a.B * a.D;
    HY_LAUNCH((conv_kernel<R, HALF, true>), dim3((rows + W::RPW - 1) / W::RPW), dim3(W::WGT), W::LDS, stream, a);
    return hy_launch_error() ? HYENA_ERR_LAUNCH : HYENA_OK;
}
template <int R>
static int conv_f32out_r(const ConvArgs& a, void* stream) {
    return a.dtype == DT_F32 ? conv_f32out_rh<R, false>(a, stream) : conv_f32out_rh<R, true>(a, stream);
}
template <int R, bool HALF>
static int conv_rh(const ConvArgs& a, void* stream) {
    typedef WgCfg<R> W;
    static thread_local int done = -1;
    hy_allow_lds(conv_kernel<R, HALF>, W::LDS, &done);
    const int rows = a.B * a.D;
    HY_LAUNCH((conv_kernel<R, HALF>), dim3((rows + W::RPW - 1) / W::RPW), dim3(W::WGT), W::LDS, stream, a);
    return hy_launch_error() ? HYENA_ERR_LAUNCH : HYENA_OK;
}
template <int R>
static int conv_r(const ConvArgs& a, void* stream) {
    return a.dtype == DT_F32 ? conv_rh<R, false>(a, stream) : conv_rh<R, true>(a, stream);
}
// ---- short rows, small batch: one launch per direction (small_fwd_kernel / small_bwd_kernel) --------------------------------
// HYENA_FFTCONV_SMALL=0 keeps the general kernels reachable at these sizes (A/B, tests).
static bool small_enabled() {
    const char* e = std::getenv("HYENA_FFTCONV_SMALL");
    return !(e != nullptr && e[0] == '0');
}
bool small_ok(int R, int B, int D, int L, Pitch ld, int dtype) {
    if (R > 2 || !small_enabled()) return false;
    const int G = 256 / (32 * R);
    // every row group takes at most two rows: beyond that the general kernels (one row per workgroup, the batch of a channel
    // sharing an XCD's L2) fill the chip better than D workgroups would
    if (B > 2 * G) return false;
    const size_t es = dtype == DT_F32 ? 4 : 2;
    return (size_t)B * D * ld.ldx * es < ((size_t)1 << 32) && (size_t)D * ld.ldk * 4 < ((size_t)1 << 32);       // 32-bit buffer offsets
}
#define HY_OC_SWITCH(R, call)                 \
    switch (R) {                              \
        case 1: return call(1);               \
        case 2: return call(2);               \
        case 4: return call(4);               \
        case 8: return call(8);               \
        case 16: return call(16);             \
        case 32: return call(32);             \
        default: return HYENA_ERR_UNSUPPORTED_L; \
    }

int launch_spec(int R, const float* k, const float* bias, void* H, const void* tab, int D, int L, Pitch ld, void* stream) {
    SpecArgs a;
    a.k = k; a.bias = bias; a.H = reinterpret_cast<c32*>(H); a.tab = reinterpret_cast<const c32*>(tab); a.D = D; a.L = L; a.dtype = DT_F32;
    a.ld = ld.ldk;
#define HY_CALL(r) spec_r<r>(a, stream)
    HY_OC_SWITCH(R, HY_CALL)
#undef HY_CALL
}

// Used at M = 32768 only, where dk_kernel needs its two parity launches: 80.0 -> 65.6 us (L = 32768, D = 256, bf16, MI355X); at
// M = 16384 / 8192 the single dk_kernel launch is the faster one (24.0 vs 38.7 us, 22.7 vs 25.6 us) -- profiles/r3ac_dk_batch1.txt.
// HYENA_FFTCONV_DK1=0 keeps dk_kernel reachable there (A/B, tests).
bool dk1_ok(int R, int B) {
    const char* e = std::getenv("HYENA_FFTCONV_DK1");
    return B == 1 && R == 32 && !(e != nullptr && e[0] == '0');
}
static int dk1_spec(int R, const SpecArgs& a, void* stream) {
#define HY_CALL(r) spec_r<r>(a, stream)
    HY_OC_SWITCH(R, HY_CALL)
#undef HY_CALL
}
static int dk1_conv(int R, const ConvArgs& a, void* stream) {
#define HY_CALL(r) conv_f32out_r<r>(a, stream)
    HY_OC_SWITCH(R, HY_CALL)
#undef HY_CALL
}
int launch_dk1(int R, const void* dout, const void* u, float* dk, float* dbias, void* Uspec, const void* tab, int D, int L, Pitch ld,
               int dtype, void* stream) {
    SpecArgs sa;
    sa.k = u; sa.bias = nullptr; sa.H = reinterpret_cast<c32*>(Uspec); sa.tab = reinterpret_cast<const c32*>(tab); sa.D = D; sa.L = L;
    sa.dtype = dtype; sa.ld = ld.ldx;
    int st = dk1_spec(R, sa, stream);
    if (st) return st;
    ConvArgs a;
    a.x = dout; a.out = dk; a.H = reinterpret_cast<const c32*>(Uspec); a.tab = reinterpret_cast<const c32*>(tab);
    a.B = 1; a.D = D; a.L = L; a.dtype = dtype; a.conj_sign = -1.0f; a.ldx = ld.ldx; a.ldo = ld.ldk;
    if ((st = dk1_conv(R, a, stream))) return st;
    if (dbias != nullptr) HY_LAUNCH((dk_bias_kernel<0>), dim3((D + 255) / 256), dim3(256), 0, stream, (const float*)dk, dbias, D, ld.ldk);
    return hy_launch_error() ? HYENA_ERR_LAUNCH : HYENA_OK;
}

int launch_conv(int R, const void* x, void* out, const void* H, const void* tab, int B, int D, int L, Pitch ld, int dtype, int conj,
                void* stream) {
    ConvArgs a;
    a.x = x; a.out = out; a.H = reinterpret_cast<const c32*>(H); a.tab = reinterpret_cast<const c32*>(tab);
    a.B = B; a.D = D; a.L = L; a.dtype = dtype; a.conj_sign = conj ? -1.0f : 1.0f; a.ldx = a.ldo = ld.ldx;
#define HY_CALL(r) conv_r<r>(a, stream)
    HY_OC_SWITCH(R, HY_CALL)
#undef HY_CALL
}

}  // namespace oc
}  // namespace hyena
