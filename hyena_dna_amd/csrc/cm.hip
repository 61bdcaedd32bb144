// cm.hip -- host side of the channel-major operator shell (cm_kernels.h; C ABI in include/hyena_mixer.h).
#include "cm_kernels.h"
#include "launch.h"
#include "../../include/hyena_fftconv.h"
#include "../../include/hyena_mixer.h"

using namespace hyena;

namespace {
// a (C, B, len) layout: row (c, b) at c cs + b bs; rows must not overlap
bool cm_layout_ok(long cs, int bs, int B, int len) { return bs >= len && cs >= (long)(B - 1) * bs + len; }
// zT / dzT of the post kernels may also be BATCH-major rows -- row (d, b) at d cs + b bs with bs >= (D - 1) cs + len: the (B, D, len) layout the
// convolution takes, cs = its row pitch, bs = D cs -- so that the gate between two long convolutions (HyenaOperator at order >= 3, hyena.py:414-423)
// writes the next convolution's input, and reads its gradient, in place (round 6)
bool cm_zlayout_ok(long cs, int bs, int B, int D, int len) {
    return cm_layout_ok(cs, bs, B, len) || (cs >= len && (long)bs >= (long)(D - 1) * cs + len);
}
bool cm_ok(const void* xT, const float* w, const float* b, int B, int L, int Lx, int D, long csx, int bsx, int lda, int dtype) {
    return xT != nullptr && w != nullptr && b != nullptr && B >= 1 && L >= 1 && Lx >= L && D >= 1 && cm_layout_ok(csx, bsx, B, Lx) && lda >= L &&
           (dtype == HYENA_F32 || dtype == HYENA_BF16 || dtype == HYENA_F16);
}
int cm_tiles(int L) { return (L + CM_TILE - 1) / CM_TILE; }
// (channel, batch) rows per workgroup: sequences that leave at least half of a 2048-position tile empty share it, 2 / 4 / 8 batch items of one channel
int cm_rpw(int B, int L) {
    int r = 1;
    while (r < 8 && L * 2 * r <= CM_TILE && r * 2 <= B) r *= 2;
    return r;
}
dim3 cm_grid(int B, int L, int D) { const int r = cm_rpw(B, L); return dim3(cm_tiles(L), D, (B + r - 1) / r); }
const size_t CM_SMEM = 2 * 5 * 4 * sizeof(float);

#define HY_CM_DISPATCH(kernel, smem)                                                                                      \
    do {                                                                                                                  \
        switch (dtype) {                                                                                                  \
            case HYENA_F32: HY_LAUNCH((kernel<DT_F32>), cm_grid(B, L, D), dim3(CM_THREADS), smem, stream, a); break;      \
            case HYENA_BF16: HY_LAUNCH((kernel<DT_BF16>), cm_grid(B, L, D), dim3(CM_THREADS), smem, stream, a); break;    \
            default: HY_LAUNCH((kernel<DT_F16>), cm_grid(B, L, D), dim3(CM_THREADS), smem, stream, a); break;             \
        }                                                                                                                 \
    } while (0)
}  // namespace

extern "C" {

size_t hyena_cm_partial_floats(int B, int L, int D) {
    if (B < 1 || L < 1 || D < 1) return 0;
    const int r = cm_rpw(B, L);
    return (size_t)3 * D * ((B + r - 1) / r) * cm_tiles(L) * CM_NP;
}

int hyena_cm_pre_fwd_ld(const void* xT, const float* bin, const float* w, const float* b, void* vg, int B, int L, int Lx, int D, long csx,
                        int bsx, int lda, int dtype, void* stream) {
    const long csz = 0; const int bsz = 0;
    if (!cm_ok(xT, w, b, B, L, Lx, D, csx, bsx, lda, dtype) || vg == nullptr) return HYENA_ERR_BAD_ARG;
    CmArgs a;
    a.xT = xT; a.bin = bin; a.w = w; a.b = b; a.a0 = nullptr; a.a1 = nullptr; a.o0 = vg; a.dxT = nullptr; a.part = nullptr;
    a.B = B; a.L = L; a.D = D; a.Lx = Lx; a.csx = csx; a.bsx = bsx; a.csz = csz; a.bsz = bsz; a.lda = lda; a.rpw = cm_rpw(B, L);
    HY_CM_DISPATCH(cm_pre_fwd_kernel, 0);
    return hy_launch_error() ? HYENA_ERR_LAUNCH : HYENA_OK;
}

int hyena_cm_post_fwd_ld(const void* y, const void* xT, const float* bin, const float* w, const float* b, void* zT, int B, int L, int Lx,
                         int D, long csx, int bsx, long csz, int bsz, int lda, int dtype, void* stream) {
    if (!cm_ok(xT, w, b, B, L, Lx, D, csx, bsx, lda, dtype) || y == nullptr || zT == nullptr || !cm_zlayout_ok(csz, bsz, B, D, L)) return HYENA_ERR_BAD_ARG;
    CmArgs a;
    a.xT = xT; a.bin = bin; a.w = w; a.b = b; a.a0 = y; a.a1 = nullptr; a.o0 = zT; a.dxT = nullptr; a.part = nullptr;
    a.B = B; a.L = L; a.D = D; a.Lx = Lx; a.csx = csx; a.bsx = bsx; a.csz = csz; a.bsz = bsz; a.lda = lda; a.rpw = cm_rpw(B, L);
    HY_CM_DISPATCH(cm_post_fwd_kernel, 0);
    return hy_launch_error() ? HYENA_ERR_LAUNCH : HYENA_OK;
}

int hyena_cm_post_bwd_ld(const void* dzT, const void* y, const void* xT, const float* bin, const float* w, const float* b, void* dy,
                         void* dxT, float* part, int B, int L, int Lx, int D, long csx, int bsx, long csz, int bsz, int lda, int dtype,
                         void* stream) {
    if (!cm_ok(xT, w, b, B, L, Lx, D, csx, bsx, lda, dtype) || !cm_zlayout_ok(csz, bsz, B, D, L) || dzT == nullptr || y == nullptr || dy == nullptr || dxT == nullptr || part == nullptr)
        return HYENA_ERR_BAD_ARG;
    CmArgs a;
    a.xT = xT; a.bin = bin; a.w = w; a.b = b; a.a0 = y; a.a1 = dzT; a.o0 = dy; a.dxT = dxT; a.part = part;
    a.B = B; a.L = L; a.D = D; a.Lx = Lx; a.csx = csx; a.bsx = bsx; a.csz = csz; a.bsz = bsz; a.lda = lda; a.rpw = cm_rpw(B, L);
    HY_CM_DISPATCH(cm_post_bwd_kernel, CM_SMEM);
    return hy_launch_error() ? HYENA_ERR_LAUNCH : HYENA_OK;
}

int hyena_cm_pre_bwd_ld(const void* dvg, const void* xT, const float* bin, const float* w, const float* b, void* dxT, float* part, int B,
                        int L, int Lx, int D, long csx, int bsx, int lda, int dtype, void* stream) {
    const long csz = 0; const int bsz = 0;
    if (!cm_ok(xT, w, b, B, L, Lx, D, csx, bsx, lda, dtype) || dvg == nullptr || dxT == nullptr || part == nullptr) return HYENA_ERR_BAD_ARG;
    CmArgs a;
    a.xT = xT; a.bin = bin; a.w = w; a.b = b; a.a0 = dvg; a.a1 = nullptr; a.o0 = nullptr; a.dxT = dxT; a.part = part;
    a.B = B; a.L = L; a.D = D; a.Lx = Lx; a.csx = csx; a.bsx = bsx; a.csz = csz; a.bsz = bsz; a.lda = lda; a.rpw = cm_rpw(B, L);
    HY_CM_DISPATCH(cm_pre_bwd_kernel, CM_SMEM);
    return hy_launch_error() ? HYENA_ERR_LAUNCH : HYENA_OK;
}

// the packed layouts: ldx = Lx, lda = L
int hyena_cm_pre_fwd(const void* xT, const float* bin, const float* w, const float* b, void* vg, int B, int L, int Lx, int D, int dtype,
                     void* stream) {
    return hyena_cm_pre_fwd_ld(xT, bin, w, b, vg, B, L, Lx, D, (long)B * Lx, Lx, L, dtype, stream);
}
int hyena_cm_post_fwd(const void* y, const void* xT, const float* bin, const float* w, const float* b, void* zT, int B, int L, int Lx,
                      int D, int dtype, void* stream) {
    return hyena_cm_post_fwd_ld(y, xT, bin, w, b, zT, B, L, Lx, D, (long)B * Lx, Lx, (long)B * L, L, L, dtype, stream);
}
int hyena_cm_post_bwd(const void* dzT, const void* y, const void* xT, const float* bin, const float* w, const float* b, void* dy,
                      void* dxT, float* part, int B, int L, int Lx, int D, int dtype, void* stream) {
    return hyena_cm_post_bwd_ld(dzT, y, xT, bin, w, b, dy, dxT, part, B, L, Lx, D, (long)B * Lx, Lx, (long)B * L, L, L, dtype, stream);
}
int hyena_cm_pre_bwd(const void* dvg, const void* xT, const float* bin, const float* w, const float* b, void* dxT, float* part, int B,
                     int L, int Lx, int D, int dtype, void* stream) {
    return hyena_cm_pre_bwd_ld(dvg, xT, bin, w, b, dxT, part, B, L, Lx, D, (long)B * Lx, Lx, L, dtype, stream);
}

}  // extern "C"
