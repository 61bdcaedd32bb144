// proj.hip -- host side of the matrix-core input projection (proj_kernels.h; C ABI in include/hyena_proj.h).
#include "proj_kernels.h"
#include "proj2_kernels.h"
#include "launch.h"
#include "../../include/hyena_fftconv.h"
#include "../../include/hyena_proj.h"

using namespace hyena;

namespace {
template <int K, int DT>
int launch_inproj(const pj::InProjArgs& a, int grid, void* stream) {
    typedef pj::IpCfg<K> C;
    static thread_local int done = -1;
    hy_allow_lds(pj::inproj_pre_fwd_kernel<K, DT>, C::LDS, &done);
    HY_LAUNCH((pj::inproj_pre_fwd_kernel<K, DT>), dim3(grid), dim3(pj::PJ_THREADS), C::LDS, stream, a);
    return hy_launch_error() ? HYENA_ERR_LAUNCH : HYENA_OK;
}
template <int K, int DT, int MODE>
int launch_mlp(const pj::MlpArgs& a, int grid, void* stream) {
    typedef pj::PmCfg<K> C;
    static thread_local int done = -1;
    hy_allow_lds(pj::mlp_kernel<K, DT, MODE>, C::lds(MODE), &done);
    HY_LAUNCH((pj::mlp_kernel<K, DT, MODE>), dim3(grid), dim3(pj::PJ_THREADS), C::lds(MODE), stream, a);
    return hy_launch_error() ? HYENA_ERR_LAUNCH : HYENA_OK;
}
// runs of tiles per unit group: a few per workgroup slot (2 per CU; tail balance), at least 8 tiles long (weight load amortised)
void mlp_schedule(size_t P, int ncg, pj::MlpArgs* a, int* runs_out, int* grid_out) {
    a->tiles = (int)((P + pj::PJ_NT - 1) / pj::PJ_NT);
    int runs = 256 * 8 / ncg;
    if (runs > a->tiles) runs = a->tiles;
    a->tiles_per_wg = (a->tiles + runs - 1) / runs;
    if (a->tiles_per_wg < 8 && a->tiles >= 8) a->tiles_per_wg = 8;
    runs = (a->tiles + a->tiles_per_wg - 1) / a->tiles_per_wg;
    *runs_out = runs;
    *grid_out = ((runs + 7) / 8) * 8 * ncg;
}
template <int K, int DT>
int launch_outproj(const pj::OutProjArgs& a, int grid, void* stream) {
    typedef pj::OpCfg<K> C;
    if (a.ln_w != nullptr) {                       // the residual add + LayerNorm in the epilogue
        static thread_local int done_ln = -1;
        hy_allow_lds(pj::outproj_gate_fwd_kernel<K, DT, true>, C::LDS_LN, &done_ln);
        HY_LAUNCH((pj::outproj_gate_fwd_kernel<K, DT, true>), dim3(grid), dim3(pj::PJ_THREADS), C::LDS_LN, stream, a);
        return hy_launch_error() ? HYENA_ERR_LAUNCH : HYENA_OK;
    }
    static thread_local int done = -1;
    hy_allow_lds(pj::outproj_gate_fwd_kernel<K, DT>, C::LDS, &done);
    HY_LAUNCH((pj::outproj_gate_fwd_kernel<K, DT>), dim3(grid), dim3(pj::PJ_THREADS), C::LDS, stream, a);
    return hy_launch_error() ? HYENA_ERR_LAUNCH : HYENA_OK;
}
// generation 2 (proj2_kernels.h): K / 16 wavefronts per workgroup, operand rows prefetched a tile ahead, whole-row epilogue
template <int K, int DT>
int launch_outproj2(const pj::OutProjArgs& a, int grid, void* stream) {
    typedef pj::Op2Cfg<K> C;
    if (a.ln_w != nullptr) {
        static thread_local int done_ln = -1;
        hy_allow_lds(pj::outproj_gate_fwd2_kernel<K, DT, true>, C::LDS_LN, &done_ln);
        HY_LAUNCH((pj::outproj_gate_fwd2_kernel<K, DT, true>), dim3(grid), dim3(C::THREADS), C::LDS_LN, stream, a);
        return hy_launch_error() ? HYENA_ERR_LAUNCH : HYENA_OK;
    }
    static thread_local int done = -1;
    hy_allow_lds(pj::outproj_gate_fwd2_kernel<K, DT, false>, C::LDS, &done);
    HY_LAUNCH((pj::outproj_gate_fwd2_kernel<K, DT, false>), dim3(grid), dim3(C::THREADS), C::LDS, stream, a);
    return hy_launch_error() ? HYENA_ERR_LAUNCH : HYENA_OK;
}
template <int K, int DT>
int launch_inproj2(const pj::InProjArgs& a, int grid, void* stream) {
    typedef pj::Ip2Cfg<K> C;
    static thread_local int done = -1;
    hy_allow_lds(pj::inproj_pre_fwd2_kernel<K, DT>, C::LDS, &done);
    HY_LAUNCH((pj::inproj_pre_fwd2_kernel<K, DT>), dim3(grid), dim3(pj::IP2_THREADS), C::LDS, stream, a);
    return hy_launch_error() ? HYENA_ERR_LAUNCH : HYENA_OK;
}
// which generation of a kernel family the entry points launch (hyena_proj_kernel_generation): A/B measurements and the tests of both
int g_outproj_generation = 2;
int g_inproj_generation = 1;        // generation 2 is built, parity-green and NOT faster at the long lengths (profiles/r6_inproj_gen2_not_kept.txt): selectable, not the default

template <int K, int DT>
int launch_dgrad(const pj::DgArgs& a, int grid, void* stream) {
    typedef pj::DgCfg<K> C;
    static thread_local int done = -1;
    hy_allow_lds(pj::outproj_dgrad_gate_bwd_kernel<K, DT>, C::LDS, &done);
    HY_LAUNCH((pj::outproj_dgrad_gate_bwd_kernel<K, DT>), dim3(grid), dim3(pj::PJ_THREADS), C::LDS, stream, a);
    return hy_launch_error() ? HYENA_ERR_LAUNCH : HYENA_OK;
}
// runs of tiles per channel group of the dgrad kernel: a few per workgroup slot, at least 16 tiles long (weight load + the warm-up tile amortised)
void dgrad_schedule(int B, int L, int D, pj::DgArgs* a, int* runs_out, int* grid_out) {
    const int ncg = D / (pj::PJ_WAVES * pj::DG_CB);
    a->tiles_per_seq = (L + pj::PJ_NT - 1) / pj::PJ_NT;
    a->tiles = B * a->tiles_per_seq;
    // ONE round of resident workgroups (DG_WGS per CU): a second, partly filled round costs up to half the launch (160 000 x 2: 640 workgroups
    // on 512 slots ran at 2.6 TB/s); runs of at least 16 tiles amortise the weight load and the warm-up tile
    int runs = 256 * DG_WGS / ncg;
    if (runs > a->tiles) runs = a->tiles;
    a->tiles_per_wg = (a->tiles + runs - 1) / runs;
    if (a->tiles_per_wg < 16 && a->tiles >= 16) a->tiles_per_wg = 16;
    runs = (a->tiles + a->tiles_per_wg - 1) / a->tiles_per_wg;
    a->nrec = runs;
    *runs_out = runs;
    *grid_out = ((runs + 7) / 8) * 8 * ncg;
}
template <int MODE>
int dispatch_mlp(const pj::MlpArgs& a, int K, int dtype, int grid, void* stream) {
    if (K == 256) return dtype == HYENA_BF16 ? launch_mlp<256, DT_BF16, MODE>(a, grid, stream) : launch_mlp<256, DT_F16, MODE>(a, grid, stream);
    return dtype == HYENA_BF16 ? launch_mlp<128, DT_BF16, MODE>(a, grid, stream) : launch_mlp<128, DT_F16, MODE>(a, grid, stream);
}
}  // namespace

#if defined(PJ_PROFILE)
// profiling builds only: hand the device a buffer for mlp_kernel's phase times ([workgroup][wavefront][16] uint64)
extern "C" int hyena_pj_prof_set(void* buf) {
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(hyena::pj::pj_prof_buf), &buf, sizeof(buf));
}
#endif

extern "C" {

static int colsum_groups(long P) {
    long g = (P + 511) / 512;                        // >= 512 rows per workgroup, at most two workgroups per CU
    return (int)(g < 1 ? 1 : (g > 512 ? 512 : g));
}

int hyena_colsum_supported(long P, int N, int dtype) {
    if (!(dtype == HYENA_BF16 || dtype == HYENA_F16) || N < 8 || N % 8 != 0 || pj::PJ_THREADS % (N / 8) != 0) return 0;
    return P >= 1 && (size_t)P < ((size_t)1 << 31) ? 1 : 0;
}

size_t hyena_colsum_partial_floats(long P, int N) { return P < 1 || N < 1 ? 0 : (size_t)colsum_groups(P) * N; }

int hyena_colsum(const void* x, float* part, float* out, long P, int N, int dtype, void* stream) {
    if (x == nullptr || part == nullptr || out == nullptr || !hyena_colsum_supported(P, N, dtype)) return HYENA_ERR_BAD_ARG;
    pj::ColsumArgs a;
    a.x = x; a.part = part; a.out = out; a.P = (unsigned)P; a.N = N; a.G = colsum_groups(P);
    a.rows_per_wg = (int)((P + a.G - 1) / a.G);
    a.G = (int)((P + a.rows_per_wg - 1) / a.rows_per_wg);
    const size_t lds = (size_t)pj::PJ_THREADS * 8 * sizeof(float);
    if (dtype == HYENA_BF16) HY_LAUNCH((pj::colsum_kernel<DT_BF16>), dim3(a.G), dim3(pj::PJ_THREADS), lds, stream, a);
    else HY_LAUNCH((pj::colsum_kernel<DT_F16>), dim3(a.G), dim3(pj::PJ_THREADS), lds, stream, a);
    HY_LAUNCH(pj::colsum_final_kernel, dim3((N + 63) / 64), dim3(pj::PJ_THREADS), pj::PJ_THREADS * sizeof(float), stream, a);
    return hy_launch_error() ? HYENA_ERR_LAUNCH : HYENA_OK;
}

int hyena_mlp_supported(long P, int K, int N, int dtype) {
    if (!(K == 128 || K == 256) || !(dtype == HYENA_BF16 || dtype == HYENA_F16)) return 0;
    if (N < 256 || N % 256 != 0 || P < 1) return 0;
    return (size_t)P < ((size_t)1 << 31) ? 1 : 0;
}

size_t hyena_mlp_partial_floats(long P, int N) {
    if (P < 1 || N < 256) return 0;
    pj::MlpArgs a;
    int runs, grid;
    mlp_schedule((size_t)P, N / 256, &a, &runs, &grid);
    return (size_t)runs * N;
}

int hyena_mlp_fc1_gelu_fwd(const void* x, const void* W1, const float* b1, void* a_out, void* h_out, long P, int K, int N, int dtype,
                           void* stream) {
    if (x == nullptr || W1 == nullptr || a_out == nullptr || h_out == nullptr || !hyena_mlp_supported(P, K, N, dtype)) return HYENA_ERR_BAD_ARG;
    pj::MlpArgs a;
    a.x = x; a.W = W1; a.bias = b1; a.a_in = nullptr; a.o0 = a_out; a.o1 = h_out; a.part = nullptr; a.P = (unsigned)P; a.N = N;
    int runs, grid;
    mlp_schedule((size_t)P, N / 256, &a, &runs, &grid);
    return dispatch_mlp<0>(a, K, dtype, grid, stream);
}

int hyena_mlp_dh_dgelu_bwd(const void* dy, const void* W2T, const void* a_in, void* da, float* part, long P, int K, int N, int dtype,
                           void* stream) {
    if (dy == nullptr || W2T == nullptr || a_in == nullptr || da == nullptr || part == nullptr || !hyena_mlp_supported(P, K, N, dtype))
        return HYENA_ERR_BAD_ARG;
    pj::MlpArgs a;
    a.x = dy; a.W = W2T; a.bias = nullptr; a.a_in = a_in; a.o0 = da; a.o1 = nullptr; a.part = part; a.P = (unsigned)P; a.N = N;
    int runs, grid;
    mlp_schedule((size_t)P, N / 256, &a, &runs, &grid);
    return dispatch_mlp<1>(a, K, dtype, grid, stream);
}

int hyena_proj_kernel_generation(int family, int generation) {
    // family 0: out_proj forward, family 1: in_proj forward (1 = rounds 3 / 4's kernels, 2 = round 6's).  generation <= 0 queries.  Returns the
    // generation in use afterwards, or -1 for an unknown family / generation.  Process-wide; not meant to be flipped while launches are in flight.
    if (family != 0 && family != 1) return -1;
    int& g = family == 0 ? g_outproj_generation : g_inproj_generation;
    if (generation == 1 || generation == 2) g = generation;
    else if (generation > 0) return -1;
    return g;
}

int hyena_outproj_supported(int B, int L, int D, int dtype) {
    if (!(D == 128 || D == 256) || !(dtype == HYENA_BF16 || dtype == HYENA_F16)) return 0;
    if (B < 1 || L < 64) return 0;                          // at least one whole 64-position tile per sequence
    return (size_t)B * (size_t)L < ((size_t)1 << 31) ? 1 : 0;
}

int hyena_outproj_gate_fwd(const void* y, const void* xT, const float* bin, const float* w, const float* b, const void* W,
                           const float* bias, void* out, void* zT, int B, int L, int Lx, int D, int dtype, void* stream) {
    return hyena_outproj_gate_fwd_ld(y, xT, bin, w, b, W, bias, out, zT, B, L, Lx, D, (long)B * Lx, Lx, (long)B * L, L, L, dtype, stream);
}

int hyena_outproj_gate_fwd_ld(const void* y, const void* xT, const float* bin, const float* w, const float* b, const void* W,
                              const float* bias, void* out, void* zT, int B, int L, int Lx, int D, long csx, int bsx, long csz, int bsz,
                              int lda, int dtype, void* stream) {
    return hyena_outproj_gate_addnorm_fwd_ld(y, xT, bin, w, b, W, bias, nullptr, nullptr, nullptr, 0.f, out, nullptr, nullptr, nullptr, zT, B, L,
                                             Lx, D, csx, bsx, csz, bsz, lda, dtype, stream);
}

static bool pj_layout_ok(long cs, int bs, int B, int len) { return bs >= len && cs >= (long)(B - 1) * bs + len; }

int hyena_outproj_gate_addnorm_fwd_ld(const void* y, const void* xT, const float* bin, const float* w, const float* b, const void* W,
                                      const float* bias, const float* residual_in, const float* ln_weight, const float* ln_bias, float eps,
                                      void* out, float* residual_out, float* mean, float* rstd, void* zT, int B, int L, int Lx, int D,
                                      long csx, int bsx, long csz, int bsz, int lda, int dtype, void* stream) {
    if (y == nullptr || xT == nullptr || w == nullptr || b == nullptr || W == nullptr || out == nullptr || L > Lx || lda < L ||
        !hyena_outproj_supported(B, L, D, dtype) || !pj_layout_ok(csx, bsx, B, Lx) || (zT != nullptr && !pj_layout_ok(csz, bsz, B, L)))
        return HYENA_ERR_BAD_ARG;
    if (ln_weight != nullptr && (ln_bias == nullptr || residual_out == nullptr || mean == nullptr || rstd == nullptr ||
                                 residual_out == residual_in))       // (a sequence's pulled-back last tile re-reads residual_in: not in place)
        return HYENA_ERR_BAD_ARG;
    pj::OutProjArgs a;
    a.y = y; a.xT = xT; a.bin = bin; a.w = w; a.b = b; a.W = W; a.bias = bias; a.out = out; a.zT = zT;
    a.res_in = residual_in; a.ln_w = ln_weight; a.ln_b = ln_bias; a.res_out = residual_out; a.mean = mean; a.rstd = rstd; a.eps = eps;
    a.B = B; a.L = L; a.Lx = Lx; a.D = D; a.csx = csx; a.bsx = bsx; a.csz = csz; a.bsz = bsz; a.lda = lda;
    a.tiles_per_seq = (L + pj::PJ_NT - 1) / pj::PJ_NT;
    a.tiles = B * a.tiles_per_seq;
    if (g_outproj_generation == 2) {
        // one (d_model 256) or two (128) workgroups per CU are resident; a few runs per slot balance the tail, runs of >= 8 tiles amortise the
        // weight load and the first tile's unhidden fetch
        int runs2 = 256 * (D == 256 ? 4 : 8);
        if (runs2 > a.tiles) runs2 = a.tiles;
        a.tiles_per_wg = (a.tiles + runs2 - 1) / runs2;
        if (a.tiles_per_wg < 8 && a.tiles >= 8) a.tiles_per_wg = 8;
        runs2 = (a.tiles + a.tiles_per_wg - 1) / a.tiles_per_wg;
        if (D == 256) return dtype == HYENA_BF16 ? launch_outproj2<256, DT_BF16>(a, runs2, stream) : launch_outproj2<256, DT_F16>(a, runs2, stream);
        return dtype == HYENA_BF16 ? launch_outproj2<128, DT_BF16>(a, runs2, stream) : launch_outproj2<128, DT_F16>(a, runs2, stream);
    }
    // generation 1: two workgroups per CU are resident; a few runs per slot balance the tail, runs of >= 8 tiles amortise the weight load
    int runs = 256 * 8;
    if (runs > a.tiles) runs = a.tiles;
    a.tiles_per_wg = (a.tiles + runs - 1) / runs;
    if (a.tiles_per_wg < 8 && a.tiles >= 8) a.tiles_per_wg = 8;
    runs = (a.tiles + a.tiles_per_wg - 1) / a.tiles_per_wg;
    if (D == 256) return dtype == HYENA_BF16 ? launch_outproj<256, DT_BF16>(a, runs, stream) : launch_outproj<256, DT_F16>(a, runs, stream);
    return dtype == HYENA_BF16 ? launch_outproj<128, DT_BF16>(a, runs, stream) : launch_outproj<128, DT_F16>(a, runs, stream);
}

int hyena_outproj_dgrad_supported(int B, int L, int D, int dtype) {
    if (!(D == 128 || D == 256) || !(dtype == HYENA_BF16 || dtype == HYENA_F16)) return 0;
    if (B < 1 || L < 1) return 0;
    return (size_t)B * (size_t)L < ((size_t)1 << 31) - 64 ? 1 : 0;
}

size_t hyena_outproj_dgrad_partial_floats(int B, int L, int D) {
    if (B < 1 || L < 1 || !(D == 128 || D == 256)) return 0;
    pj::DgArgs a;
    int runs, grid;
    dgrad_schedule(B, L, D, &a, &runs, &grid);
    return (size_t)D * runs * 8;
}

int hyena_outproj_dgrad_gate_bwd_ld(const void* dy, const void* Wt, const void* y, const void* xT, const float* bin, const float* w,
                                    const float* b, void* dyc, void* dxT, float* part, int B, int L, int Lx, int D, long csx, int bsx,
                                    int lda, int dtype, void* stream) {
    if (dy == nullptr || Wt == nullptr || y == nullptr || xT == nullptr || w == nullptr || b == nullptr || dyc == nullptr || dxT == nullptr ||
        part == nullptr || L > Lx || !pj_layout_ok(csx, bsx, B, Lx) || lda < L || !hyena_outproj_dgrad_supported(B, L, D, dtype))
        return HYENA_ERR_BAD_ARG;
    pj::DgArgs a;
    a.dy = dy; a.Wt = Wt; a.y = y; a.xT = xT; a.bin = bin; a.w = w; a.b = b; a.dyc = dyc; a.dxT = dxT; a.part = part;
    a.B = B; a.L = L; a.Lx = Lx; a.D = D; a.csx = csx; a.bsx = bsx; a.lda = lda;
    int runs, grid;
    dgrad_schedule(B, L, D, &a, &runs, &grid);
    if (D == 256) return dtype == HYENA_BF16 ? launch_dgrad<256, DT_BF16>(a, grid, stream) : launch_dgrad<256, DT_F16>(a, grid, stream);
    return dtype == HYENA_BF16 ? launch_dgrad<128, DT_BF16>(a, grid, stream) : launch_dgrad<128, DT_F16>(a, grid, stream);
}

int hyena_proj_supported(int B, int Lx, int D, int dtype) {
    if (!(D == 128 || D == 256) || !(dtype == HYENA_BF16 || dtype == HYENA_F16)) return 0;
    if (B < 1 || Lx < 8) return 0;
    return (size_t)B * (size_t)Lx < ((size_t)1 << 31) ? 1 : 0;
}

int hyena_inproj_pre_fwd(const void* u, const void* W, const float* bin, const float* w, const float* b, void* xT, void* vg,
                         int B, int Lx, int Lc, int D, int dtype, void* stream) {
    return hyena_inproj_pre_fwd_ld(u, W, bin, w, b, xT, vg, B, Lx, Lc, D, (long)B * Lx, Lx, Lc, dtype, stream);
}

int hyena_inproj_pre_fwd_ld(const void* u, const void* W, const float* bin, const float* w, const float* b, void* xT, void* vg,
                            int B, int Lx, int Lc, int D, long csx, int bsx, int ldv, int dtype, void* stream) {
    if (u == nullptr || W == nullptr || w == nullptr || b == nullptr || xT == nullptr || vg == nullptr || Lc < 1 || Lc > Lx ||
        ldv < Lc || !hyena_proj_supported(B, Lx, D, dtype) || bsx < Lx || csx < (long)(B - 1) * bsx + Lx || csx >= ((long)1 << 31))
        return HYENA_ERR_BAD_ARG;
    pj::InProjArgs a;
    a.u = u; a.W = W; a.bin = bin; a.w = w; a.b = b; a.xT = xT; a.vg = vg; a.B = B; a.Lx = Lx; a.Lc = Lc; a.D = D; a.csx = csx; a.bsx = bsx; a.ldv = ldv;
    const size_t P = (size_t)B * Lx;
    a.tiles = (int)((P + pj::PJ_NT - 1) / pj::PJ_NT);
    if (g_inproj_generation == 2) {
        // generation 2: one workgroup (12 wavefronts, 128 channels x 3 groups) per CU is resident; four per CU in total balance the tail, runs of
        // >= 16 tiles amortise the weight load and the warm-up tile
        const int ncg2 = D / pj::IP2_CH;
        int runs2 = 256 * 4 / ncg2;
        if (runs2 > a.tiles) runs2 = a.tiles;
        a.tiles_per_wg = (a.tiles + runs2 - 1) / runs2;
        if (a.tiles_per_wg < 16 && a.tiles >= 16) a.tiles_per_wg = 16;
        runs2 = (a.tiles + a.tiles_per_wg - 1) / a.tiles_per_wg;
        const int grid2 = ((runs2 + 7) / 8) * 8 * ncg2;
        if (D == 256) return dtype == HYENA_BF16 ? launch_inproj2<256, DT_BF16>(a, grid2, stream) : launch_inproj2<256, DT_F16>(a, grid2, stream);
        return dtype == HYENA_BF16 ? launch_inproj2<128, DT_BF16>(a, grid2, stream) : launch_inproj2<128, DT_F16>(a, grid2, stream);
    }
    // One workgroup per CU is resident (its wavefronts hold the weights in ~350 registers): a few runs per CU balance the tail,
    // long runs amortise the weight load and the warm-up tile.
    // Two workgroups per CU are resident (each wavefront holds its 48 weight rows in 96 registers): a few runs per slot balance
    // the tail, long runs amortise the weight load and the warm-up tile.
    const int ncg = D / (pj::PJ_WAVES * pj::IP_CB);
    int runs = 256 * 8 / ncg;
    if (runs > a.tiles) runs = a.tiles;
    a.tiles_per_wg = (a.tiles + runs - 1) / runs;
    if (a.tiles_per_wg < 8 && a.tiles >= 8) a.tiles_per_wg = 8;
#ifdef PJ_DBG_TPW
    if (const char* e = getenv("HYENA_PJ_TPW")) a.tiles_per_wg = atoi(e);        // experiments only: run length override
#endif
    runs = (a.tiles + a.tiles_per_wg - 1) / a.tiles_per_wg;
    const int grid = ((runs + 7) / 8) * 8 * ncg;
    if (D == 256) return dtype == HYENA_BF16 ? launch_inproj<256, DT_BF16>(a, grid, stream) : launch_inproj<256, DT_F16>(a, grid, stream);
    return dtype == HYENA_BF16 ? launch_inproj<128, DT_BF16>(a, grid, stream) : launch_inproj<128, DT_F16>(a, grid, stream);
}

}  // extern "C"
