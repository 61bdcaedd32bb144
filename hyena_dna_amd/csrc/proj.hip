// proj.hip -- host side of the matrix-core input projection (proj_kernels.h; C ABI in include/hyena_proj.h).
#include "proj_kernels.h"
#include "launch.h"
#include "../../include/hyena_fftconv.h"
#include "../../include/hyena_proj.h"

using namespace hyena;

namespace {
template <int K, int DT>
int launch_inproj(const pj::InProjArgs& a, int grid, void* stream) {
    typedef pj::PjCfg<K> C;
    static thread_local int done = -1;
    hy_allow_lds(pj::inproj_pre_fwd_kernel<K, DT>, C::LDS, &done);
    HY_LAUNCH((pj::inproj_pre_fwd_kernel<K, DT>), dim3(grid), dim3(pj::PJ_THREADS), C::LDS, stream, a);
    return hy_launch_error() ? HYENA_ERR_LAUNCH : HYENA_OK;
}
}  // namespace

extern "C" {

int hyena_proj_supported(int B, int Lx, int D, int dtype) {
    if (!(D == 128 || D == 256) || !(dtype == HYENA_BF16 || dtype == HYENA_F16)) return 0;
    if (B < 1 || Lx < 8) return 0;
    return (size_t)B * (size_t)Lx < ((size_t)1 << 31) ? 1 : 0;
}

int hyena_inproj_pre_fwd(const void* u, const void* W, const float* bin, const float* w, const float* b, void* xT, void* vg,
                         int B, int Lx, int Lc, int D, int dtype, void* stream) {
    if (u == nullptr || W == nullptr || w == nullptr || b == nullptr || xT == nullptr || vg == nullptr || Lc < 1 || Lc > Lx ||
        !hyena_proj_supported(B, Lx, D, dtype))
        return HYENA_ERR_BAD_ARG;
    pj::InProjArgs a;
    a.u = u; a.W = W; a.bin = bin; a.w = w; a.b = b; a.xT = xT; a.vg = vg; a.B = B; a.Lx = Lx; a.Lc = Lc; a.D = D;
    const size_t P = (size_t)B * Lx;
    a.tiles = (int)((P + pj::PJ_NT - 1) / pj::PJ_NT);
    // One workgroup per CU is resident (its wavefronts hold the weights in ~350 registers): a few runs per CU balance the tail,
    // long runs amortise the weight load and the warm-up tile.
    const int ncg = (D + pj::PJ_WAVES * pj::PJ_CB - 1) / (pj::PJ_WAVES * pj::PJ_CB);
    int runs = 256 * 4 / ncg;
    if (runs > a.tiles) runs = a.tiles;
    a.tiles_per_wg = (a.tiles + runs - 1) / runs;
    if (a.tiles_per_wg < 8 && a.tiles >= 8) a.tiles_per_wg = 8;
    runs = (a.tiles + a.tiles_per_wg - 1) / a.tiles_per_wg;
    const int grid = ((runs + 7) / 8) * 8 * ncg;
    if (D == 256) return dtype == HYENA_BF16 ? launch_inproj<256, DT_BF16>(a, grid, stream) : launch_inproj<256, DT_F16>(a, grid, stream);
    return dtype == HYENA_BF16 ? launch_inproj<128, DT_BF16>(a, grid, stream) : launch_inproj<128, DT_F16>(a, grid, stream);
}

}  // extern "C"
