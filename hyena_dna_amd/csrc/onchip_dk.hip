// onchip_dk.hip -- host side of the workspace-free path's REGISTER-BOUND kernels: dk (512 threads x up to 256 VGPRs, two wavefronts per
// SIMD) and the one-launch kernels for short rows (one wavefront per SIMD).  At that occupancy a wavefront's own instruction stream is
// the bound, so this translation unit compiles the shared kernel sources with the complex arithmetic on the packed fp32 VALU
// (HY_PACKED_F32, fftconv_kernels.h: a complex add is one v_pk_add_f32, a multiply v_pk_mul_f32 + v_pk_fma_f32 -- 40 % fewer
// instructions); measured on MI355X (profiles/r3n_packed_ab.txt): dk 96 -> 69 us at 8192 x 8, 84 -> 53 us at 2048 x 32, the short-row
// pair 34.4 -> 29.9 us at 1024 x 8 x 128.  The conv / spectrum kernels of onchip.hip run at four wavefronts per SIMD, where packed
// instructions (half issue rate) change nothing, and stay unpacked.
#define HY_PACKED_F32 1
#ifndef OC_X1_HALVES
#define OC_X1_HALVES 0          // exchange 1 as complex halves (onchip_kernels.h) fits these kernels' 256-register budget -- and gains nothing: off
#endif
#include <cstdlib>
#include "onchip_kernels.h"
#include "launch.h"
#include "onchip_host.h"
#include "../../include/hyena_fftconv.h"

namespace hyena {
namespace oc {

template <int R, int NP, bool HALF>
static int dk_rh(const DkArgs& a, void* stream) {
    typedef DkCfg<R, NP> K;
    static thread_local int done = -1;
    hy_allow_lds(dk_kernel<R, NP, HALF, 0>, K::LDS, &done);
    HY_LAUNCH((dk_kernel<R, NP, HALF, 0>), dim3(a.D, a.S), dim3(K::WGT), K::LDS, stream, a);
    if constexpr (NP == 2) {       // the odd bins, a second launch (it adds to what the first one left in dk)
        static thread_local int done1 = -1;
        hy_allow_lds(dk_kernel<R, NP, HALF, 1>, K::LDS, &done1);
        HY_LAUNCH((dk_kernel<R, NP, HALF, 1>), dim3(a.D, a.S), dim3(K::WGT), K::LDS, stream, a);
    }
    if (a.S > 1) HY_LAUNCH((dk_sum_kernel<0>), dim3((a.L + 255) / 256, a.D), dim3(256), 0, stream, a);
    return hy_launch_error() ? HYENA_ERR_LAUNCH : HYENA_OK;
}
template <int R, int NP>
static int dk_r(const DkArgs& a, void* stream) {
    return a.dtype == DT_F32 ? dk_rh<R, NP, false>(a, stream) : dk_rh<R, NP, true>(a, stream);
}
// dk with du from the same transform of dout (round 6; M <= 16384)
template <int R, bool HALF>
static int dkdu_rh(const DkArgs& a, void* stream) {
    typedef DkCfg<R, 1> K;
    static thread_local int done = -1;
    hy_allow_lds((dk_kernel<R, 1, HALF, 0, true>), K::LDS, &done);
    HY_LAUNCH((dk_kernel<R, 1, HALF, 0, true>), dim3(a.D, a.S), dim3(K::WGT), K::LDS, stream, a);
    if (a.S > 1) HY_LAUNCH((dk_sum_kernel<0>), dim3((a.L + 255) / 256, a.D), dim3(256), 0, stream, a);
    return hy_launch_error() ? HYENA_ERR_LAUNCH : HYENA_OK;
}
template <int R>
static int dkdu_r(const DkArgs& a, void* stream) {
    return a.dtype == DT_F32 ? dkdu_rh<R, false>(a, stream) : dkdu_rh<R, true>(a, stream);
}

template <int R, bool HALF>
static int small_fwd_rh(const SmallFwdArgs& a, void* stream) {
    typedef SmallCfg<R> S;
    static thread_local int done = -1;
    hy_allow_lds(small_fwd_kernel<R, HALF>, S::LDS_FWD, &done);
    HY_LAUNCH((small_fwd_kernel<R, HALF>), dim3(a.D, 1), dim3(S::WGT_FWD), S::LDS_FWD, stream, a);
    return hy_launch_error() ? HYENA_ERR_LAUNCH : HYENA_OK;
}
template <int R, bool HALF>
static int small_bwd_rh(const SmallBwdArgs& a, void* stream) {
    typedef SmallCfg<R> S;
    static thread_local int done = -1;
    hy_allow_lds(small_bwd_kernel<R, HALF>, S::LDS_BWD, &done);
    HY_LAUNCH((small_bwd_kernel<R, HALF>), dim3(a.D), dim3(S::WGT_BWD), S::LDS_BWD, stream, a);
    return hy_launch_error() ? HYENA_ERR_LAUNCH : HYENA_OK;
}
int launch_small_fwd(int R, const void* x, void* out, const float* k, const float* bias, void* Hout, const void* tab, int B, int D, int L,
                     Pitch ld, int dtype, void* stream) {
    SmallFwdArgs a;
    a.x = x; a.out = out; a.k = k; a.bias = bias; a.Hout = reinterpret_cast<c32*>(Hout); a.tab = reinterpret_cast<const c32*>(tab);
    a.B = B; a.D = D; a.L = L; a.dtype = dtype; a.ldx = ld.ldx; a.ldk = ld.ldk;
    const bool half = dtype != DT_F32;
    if (R == 1) return half ? small_fwd_rh<1, true>(a, stream) : small_fwd_rh<1, false>(a, stream);
    if (R == 2) return half ? small_fwd_rh<2, true>(a, stream) : small_fwd_rh<2, false>(a, stream);
    return HYENA_ERR_UNSUPPORTED_L;
}
int launch_small_bwd(int R, const void* dout, const void* u, void* du, float* dk, float* dbias, const void* H, const void* tab, int B, int D,
                     int L, Pitch ld, int dtype, void* stream) {
    SmallBwdArgs a;
    a.dout = dout; a.u = u; a.du = du; a.dk = dk; a.dbias = dbias; a.H = reinterpret_cast<const c32*>(H);
    a.tab = reinterpret_cast<const c32*>(tab); a.B = B; a.D = D; a.L = L; a.dtype = dtype; a.ldx = ld.ldx; a.ldk = ld.ldk;
    const bool half = dtype != DT_F32;
    if (R == 1) return half ? small_bwd_rh<1, true>(a, stream) : small_bwd_rh<1, false>(a, stream);
    if (R == 2) return half ? small_bwd_rh<2, true>(a, stream) : small_bwd_rh<2, false>(a, stream);
    return HYENA_ERR_UNSUPPORTED_L;
}

// du and dk in one launch, at least two batch items.  Measured on the MI355X (profiles/r6_dudk_ab.txt, bf16, forward + backward): 16384 x 8 x 256
// 353.9 -> 310.2 us (- 12 %), 16383 x 8: 359.1 -> 309.0, 16384 x 2: 110.9 -> 103.7, 16384 x 8 x 128: 185.8 -> 167.4 -- and a LOSS below: 8192 x 8
// 163.9 -> 182.6, 4096 x 16: 157.8 -> 190.8, 2048 x 64 x 128: 144.9 -> 159.0 (the third transform at dk_kernel's two wavefronts per SIMD, 9 - 25
// spilled registers at R = 2 ... 8, costs more than the second transform of dout at conv_kernel's four).  Default: M = 16384 only;
// HYENA_FFTCONV_DUDK=1 takes it at every M <= 16384, =0 never.
bool dkdu_ok(int R, int B) {
    const char* e = std::getenv("HYENA_FFTCONV_DUDK");
    if (B < 2 || R < 1 || R > 16) return false;
    if (e != nullptr && e[0] == '0') return false;
    if (e != nullptr && e[0] == '1') return true;
    return R == 16;
}
int launch_dkdu(int R, const void* dout, const void* u, void* du, const void* H, float* dk, float* dbias, void* partials, const void* tab, int B,
                int D, int L, Pitch ld, int dtype, void* stream) {
    DkArgs a;
    a.dout = dout; a.u = u; a.dk = dk; a.dbias = dbias; a.tab = reinterpret_cast<const c32*>(tab);
    a.B = B; a.D = D; a.L = L; a.dtype = dtype; a.ldx = ld.ldx; a.ldk = ld.ldk;
    a.du = du; a.H = reinterpret_cast<const c32*>(H);
    a.S = dk_slices(R, B, D, &a.nb);
    a.part = reinterpret_cast<float*>(partials);
    if (a.S > 1 && partials == nullptr) return HYENA_ERR_WORKSPACE;
#define HY_CALL(r) dkdu_r<r>(a, stream)
    switch (R) {
        case 1: return HY_CALL(1);
        case 2: return HY_CALL(2);
        case 4: return HY_CALL(4);
        case 8: return HY_CALL(8);
        case 16: return HY_CALL(16);
        default: return HYENA_ERR_UNSUPPORTED_L;
    }
#undef HY_CALL
}

int launch_dk(int R, const void* dout, const void* u, float* dk, float* dbias, void* partials, const void* tab, int B, int D, int L,
              Pitch ld, int dtype, void* stream) {
    DkArgs a;
    a.du = nullptr; a.H = nullptr;
    a.dout = dout; a.u = u; a.dk = dk; a.dbias = dbias; a.tab = reinterpret_cast<const c32*>(tab);
    a.B = B; a.D = D; a.L = L; a.dtype = dtype; a.ldx = ld.ldx; a.ldk = ld.ldk;
    a.S = dk_slices(R, B, D, &a.nb);
    a.part = reinterpret_cast<float*>(partials);
    if (a.S > 1 && partials == nullptr) return HYENA_ERR_WORKSPACE;
    if (R == 32) {       // two 16384-point parity problems; their tables follow the size-32 set
        a.tab = reinterpret_cast<const c32*>(tab) + set_entries(32);
        return dk_r<16, 2>(a, stream);
    }
#define HY_CALL(r) dk_r<r, 1>(a, stream)
    switch (R) {
        case 1: return HY_CALL(1);
        case 2: return HY_CALL(2);
        case 4: return HY_CALL(4);
        case 8: return HY_CALL(8);
        case 16: return HY_CALL(16);
        default: return HYENA_ERR_UNSUPPORTED_L;
    }
#undef HY_CALL
}

}  // namespace oc
}  // namespace hyena
