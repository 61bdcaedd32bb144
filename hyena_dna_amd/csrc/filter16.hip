// filter16.hip -- host side of the implicit filter's 16-bit (autocast) kernels (filter16_kernels.h; C ABI in include/hyena_filter.h).
// Stateless like the rest of the library: no allocation, no synchronisation, every launch on the caller's stream.
#define HY_HELPERS_ONLY             // fftconv_kernels.h: element types, buffer helpers, no kernels
#define FLT_DECLARE_ONLY            // filter_reduce_kernel & co. are defined in fftconv.hip's translation unit
#include "filter16_kernels.h"
#include "launch.h"
#include "../../include/hyena_fftconv.h"
#include "../../include/hyena_filter.h"

using namespace hyena;
using namespace hyena::f16k;

namespace {
const size_t F16_RED_SMEM = FLT_RED_J * FLT_RED_S * sizeof(float);

int f16_grid(int L) {
    const int n = (L + FLT_WG_POS - 1) / FLT_WG_POS;
    return n < FLT_MAX_WG ? n : FLT_MAX_WG;
}

bool f16_params_ok(const hyena_filter_params* p, int dtype) {
    return p != nullptr && p->z && p->t && p->w0 && p->b0 && p->w1 && p->b1 && p->w2 && p->b2 && p->w3 && p->freq &&
           (p->deltas || !p->modulate) && p->z_stride >= p->E && hyena_filter_supported(p->L, p->E, FLT_O, p->D) &&
           (dtype == HYENA_BF16 || dtype == HYENA_F16);
}

template <int D, int DT>
void launch_fwd(const FilterArgs& a, void* stream) {
    const int ntiles = (a.L + FLT_TP - 1) / FLT_TP;
    int grid = (ntiles + FLT_WAVES - 1) / FLT_WAVES;
    if (grid > FLT_MAX_WG) grid = FLT_MAX_WG;
    static thread_local int done_save = -1, done_plain = -1;
    if (a.acts != nullptr) {
        hy_allow_lds(flt16_fwd_kernel<D, true, DT>, F16FwdLds<D>::BYTES, &done_save);
        HY_LAUNCH((flt16_fwd_kernel<D, true, DT>), dim3(grid), dim3(FLT_THREADS), F16FwdLds<D>::BYTES, stream, a);
    } else {
        hy_allow_lds(flt16_fwd_kernel<D, false, DT>, F16FwdLds<D>::BYTES, &done_plain);
        HY_LAUNCH((flt16_fwd_kernel<D, false, DT>), dim3(grid), dim3(FLT_THREADS), F16FwdLds<D>::BYTES, stream, a);
    }
}

template <int DT>
void launch_fwd_d(const FilterArgs& a, void* stream) {
    switch (a.D) {
        case 64: launch_fwd<64, DT>(a, stream); break;
        case 128: launch_fwd<128, DT>(a, stream); break;
        default: launch_fwd<256, DT>(a, stream); break;
    }
}

// layout of the partial-sum scratch: [slots][NO][64] weight | [2 slots][NO] bias | [grid * 8][64] frequency
template <int NO, bool MOD, int DT, bool OUTF32>
void launch_layer(F16BwdArgs a, float* part, float* dw, float* db, float* dfreq, bool first_freq, void* stream) {
    typedef FltBwdCfg<NO, FLT_O> Cfg;
    const int grid = f16_grid(a.L);
    const int slots = grid * Cfg::KS;
    a.part_w = part;
    a.part_b = db != nullptr ? part + (size_t)slots * NO * FLT_O : nullptr;
    a.part_f = part + (size_t)slots * NO * FLT_O + (size_t)2 * slots * NO;
    static thread_local int done = -1;
    hy_allow_lds(flt16_layer_bwd_kernel<NO, MOD, DT, OUTF32>, F16BwdLds<NO>::BYTES, &done);
    HY_LAUNCH((flt16_layer_bwd_kernel<NO, MOD, DT, OUTF32>), dim3(grid), dim3(FLT_THREADS), F16BwdLds<NO>::BYTES, stream, a);
    const int nw = NO * FLT_O;
    RedBatch red;                                   // the layer's three fixed-order reductions in one launch
    red.add(a.part_w, dw, slots, nw, nw, 0);
    if (db != nullptr) red.add(a.part_b, db, 2 * slots, NO, NO, 0);
    red.add(a.part_f, dfreq, grid * FLT_WAVES, FLT_O, FLT_O, first_freq ? 0 : 1);
    HY_LAUNCH(filter_reduce_multi_kernel, dim3(red.blocks()), dim3(256), F16_RED_SMEM, stream, red.jobs);
}

// the first layer (contraction length E <= 8) on filter_kernels.h's fp32 kernel, operands rounded on load (FilterBwdArgs::rdt)
void launch_layer0(FilterBwdArgs a, float* part, float* dw, float* db, void* stream) {
    typedef FltBwdCfg<FLT_O, FLT_E> Cfg;
    const int grid = f16_grid(a.L);
    const int slots = grid * Cfg::KS;
    a.part_w = part;
    a.part_b = part + (size_t)slots * FLT_O * FLT_E;
    a.part_f = part + (size_t)slots * FLT_O * FLT_E + (size_t)2 * slots * FLT_O;
    static thread_local int done = -1;
    hy_allow_lds(filter_layer_bwd_kernel<FLT_O, FLT_E, 0>, Cfg::BYTES, &done);
    HY_LAUNCH((filter_layer_bwd_kernel<FLT_O, FLT_E, 0>), dim3(grid), dim3(FLT_THREADS), Cfg::BYTES, stream, a);
    const int nw = FLT_O * a.ni;
    if (a.ni == FLT_E) {
        RedBatch red;
        red.add(a.part_w, dw, slots, nw, nw, 0);
        red.add(a.part_b, db, 2 * slots, (int)FLT_O, (int)FLT_O, 0);
        HY_LAUNCH(filter_reduce_multi_kernel, dim3(red.blocks()), dim3(256), F16_RED_SMEM, stream, red.jobs);
        return;
    } else {
        float* tmp = a.part_f + (size_t)grid * FLT_WAVES * FLT_O;
        HY_LAUNCH(filter_reduce_kernel, dim3((FLT_O * FLT_E + FLT_RED_J - 1) / FLT_RED_J), dim3(256), F16_RED_SMEM, stream, (const float*)a.part_w, tmp, slots,
                  FLT_O * FLT_E, 0);
        HY_LAUNCH(filter_compact_kernel, dim3((nw + 255) / 256), dim3(256), 0, stream, (const float*)tmp, dw, (int)FLT_O, (int)FLT_E, a.ni);
    }
    HY_LAUNCH(filter_reduce_kernel, dim3((FLT_O + FLT_RED_J - 1) / FLT_RED_J), dim3(256), F16_RED_SMEM, stream, (const float*)a.part_b, db, 2 * slots, (int)FLT_O, 0);
}

template <int DT>
void bwd_all(const hyena_filter_params* p, const float* dk, int ldk, const void* saved, const hyena_filter_grads* g, void* workspace, void* stream) {
    const int L = p->L, P = hyena_filter_row_pitch(L);       // library-owned rows are pitched to 64 words (fftconv.hip)
    float* dA = static_cast<float*>(workspace);              // (64, P) floats each: pair words use the first half
    float* dB = dA + (size_t)FLT_O * P;
    float* part = dB + (size_t)FLT_O * P;
    const uint32_t* sv = static_cast<const uint32_t*>(saved);
    const uint32_t* a0 = sv;
    const uint32_t* a1 = sv + (size_t)(FLT_O / 2) * P;
    const uint32_t* a2 = sv + (size_t)FLT_O * P;

    F16BwdArgs a;
    a.freq = p->freq; a.t = p->t; a.deltas = p->deltas; a.shift = p->shift; a.modulate = p->modulate; a.L = L;
    a.part_w = nullptr; a.part_b = nullptr; a.part_f = nullptr;
    // last layer: delta_3 = R(dk * modulation);  dW3, delta_2 -> dA
    a.dout = dk; a.w = p->w3; a.aprev = a2; a.dprev = dA; a.ldo = ldk; a.lda = P; a.ldp = P;
    switch (p->D) {
        case 64: launch_layer<64, true, DT, false>(a, part, g->dw3, nullptr, g->dfreq, true, stream); break;
        case 128: launch_layer<128, true, DT, false>(a, part, g->dw3, nullptr, g->dfreq, true, stream); break;
        default: launch_layer<256, true, DT, false>(a, part, g->dw3, nullptr, g->dfreq, true, stream); break;
    }
    a.modulate = 0; a.t = nullptr; a.deltas = nullptr; a.ldo = P;
    a.dout = dA; a.w = p->w2; a.aprev = a1; a.dprev = dB;
    launch_layer<FLT_O, false, DT, false>(a, part, g->dw2, g->db2, g->dfreq, false, stream);
    a.dout = dB; a.w = p->w1; a.aprev = a0; a.dprev = dA;                     // delta_0 leaves as fp32 rows for the fp32 kernel below
    launch_layer<FLT_O, false, DT, true>(a, part, g->dw1, g->db1, g->dfreq, false, stream);
    FilterBwdArgs b;
    b.dout = dA; b.w = p->w0; b.aprev = p->z; b.freq = p->freq; b.t = p->t; b.deltas = p->deltas; b.dprev = g->dz;
    b.part_w = nullptr; b.part_b = nullptr; b.part_f = nullptr; b.shift = p->shift; b.modulate = 0; b.L = L; b.ni = p->E; b.zs = p->z_stride;
    b.rdt = DT; b.ldo = P; b.lda = P; b.ldp = L;              // dz (E, L) is the caller's: packed
    launch_layer0(b, part, g->dw0, g->db0, stream);
}
}  // namespace

extern "C" {

size_t hyena_filter16_saved_bytes(int L) { return L >= 1 ? (size_t)3 * (FLT_O / 2) * hyena_filter_row_pitch(L) * sizeof(uint32_t) : 0; }

int hyena_filter16_fwd(const hyena_filter_params* p, int dtype, float* k, void* saved, void* stream) {
    return hyena_filter16_fwd_ld(p, dtype, k, p != nullptr ? p->L : 0, saved, stream);
}

int hyena_filter16_bwd(const hyena_filter_params* p, int dtype, const float* dk, const void* saved, const hyena_filter_grads* g,
                       void* workspace, size_t workspace_bytes, void* stream) {
    return hyena_filter16_bwd_ld(p, dtype, dk, p != nullptr ? p->L : 0, saved, g, workspace, workspace_bytes, stream);
}

int hyena_filter16_fwd_ld(const hyena_filter_params* p, int dtype, float* k, int ldk, void* saved, void* stream) {
    if (!f16_params_ok(p, dtype) || k == nullptr || ldk < p->L) return HYENA_ERR_BAD_ARG;
    FilterArgs a;
    a.z = p->z; a.t = p->t; a.w0 = p->w0; a.b0 = p->b0; a.w1 = p->w1; a.b1 = p->b1; a.w2 = p->w2; a.b2 = p->b2; a.w3 = p->w3;
    a.freq = p->freq; a.deltas = p->deltas; a.k = k; a.acts = static_cast<float*>(saved); a.shift = p->shift; a.modulate = p->modulate;
    a.L = p->L; a.E = p->E; a.zs = p->z_stride; a.D = p->D; a.ldk = ldk; a.lds = hyena_filter_row_pitch(p->L);
    if (dtype == HYENA_BF16) launch_fwd_d<DT_BF16>(a, stream);
    else launch_fwd_d<DT_F16>(a, stream);
    return hy_launch_error() ? HYENA_ERR_LAUNCH : HYENA_OK;
}

int hyena_filter16_bwd_ld(const hyena_filter_params* p, int dtype, const float* dk, int ldk, const void* saved, const hyena_filter_grads* g,
                          void* workspace, size_t workspace_bytes, void* stream) {
    if (!f16_params_ok(p, dtype) || dk == nullptr || saved == nullptr || g == nullptr || workspace == nullptr || ldk < p->L) return HYENA_ERR_BAD_ARG;
    if (!g->dw0 || !g->db0 || !g->dw1 || !g->db1 || !g->dw2 || !g->db2 || !g->dw3 || !g->dfreq) return HYENA_ERR_BAD_ARG;
    if (workspace_bytes < hyena_filter_workspace_bytes(p->L, p->D)) return HYENA_ERR_WORKSPACE;
    if (dtype == HYENA_BF16) bwd_all<DT_BF16>(p, dk, ldk, saved, g, workspace, stream);
    else bwd_all<DT_F16>(p, dk, ldk, saved, g, workspace, stream);
    return hy_launch_error() ? HYENA_ERR_LAUNCH : HYENA_OK;
}

}  // extern "C"
