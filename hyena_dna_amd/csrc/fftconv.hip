// fftconv.hip -- host side of libhyena_fftconv.so: the C ABI declared in include/hyena_fftconv.h.
//
// Stateless: no allocation, no synchronisation, no globals; every launch goes to the caller's stream, so the
// entry points can be captured into a hipGraph and are safe to call from the autograd thread (the reference
// kernels launch on stream 0 without a device guard, csrc/fftconv/fftconv_cuda.cu:812 -- not repeated here).
//
// Built by hipcc --offload-arch=gfx950 (product).  With -DHIPEMU it builds against tests/hipemu with g++
// into a CPU emulation used only by the `not gpu` tests.
#include "fftconv_kernels.h"
#include "mixer_kernels.h"
#include "filter_kernels.h"
#include "block_kernels.h"
#include "../../include/hyena_fftconv.h"
#include "../../include/hyena_mixer.h"
#include "../../include/hyena_filter.h"
#include "../../include/hyena_block.h"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

using namespace hyena;

#include "launch.h"
#include "onchip_host.h"

namespace {

struct Plan {
    int L, M, M1;
    int R;          // > 0: the workspace-free path (onchip_kernels.h), M = 1024 R;  0: the two-level path below
};

// L <= 32768 runs on the workspace-free path unless HYENA_FFTCONV_ONCHIP=0 (read per call: a test / profiling knob that
// keeps the two-level kernels reachable at small sizes; it selects between two implementations of the same HIP library)
bool onchip_enabled() {
    const char* e = std::getenv("HYENA_FFTCONV_ONCHIP");
    return !(e != nullptr && e[0] == '0');
}

// Column sizes M1 the kernels are instantiated for (M = M1 x 1024 complex points serve L <= M): the powers of two, and
// 2^a x {3, 5, 7} where the second-stage operands still fit the register budget -- so that the zero padding beyond 2L
// is at most ~20 % instead of up to 100 % (hyenadna-medium-160k: M1 = 160; -450k: M1 = 448).
const int SUPPORTED_M1[] = {1, 2, 3, 4, 5, 6, 7, 8, 10, 12, 14, 16, 20, 24, 28, 32, 64, 96, 128, 160, 192, 224, 256, 320, 384, 448,
                            512, 640, 768, 1024};

bool make_plan(int L, Plan* p) {
    if (L < 1 || L > HYENA_MAX_L) return false;
    p->R = onchip_enabled() ? oc::plan_r(L) : 0;
    if (p->R) {
        p->L = L;
        p->M1 = p->R;
        p->M = 1024 * p->R;
        return true;
    }
    for (int m1 : SUPPORTED_M1) {
        if ((long)m1 * 1024 >= L) {
            p->L = L;
            p->M1 = m1;
            p->M = m1 * 1024;
            return true;
        }
    }
    return false;
}

// table layout inside d_tables (c32 units): tw_lo[1024] | tw_hi[1024 slots, M1 used] | tw_row[1024] | tw_rowT[1024]
Tables tables_from(const void* d_tables) {
    const c32* t = reinterpret_cast<const c32*>(d_tables);
    Tables tab;
    tab.tw_lo = t;
    tab.tw_hi = t + 1024;
    tab.tw_row = t + 2048;
    tab.tw_rowT = t + 3072;
    return tab;
}

template <int DT, bool INV>
int launch_col_dt(int M1, const ColArgs& a, int rows, void* stream) {
#define HY_COL_CASE(m1)                                                                              \
    case m1: {                                                                                       \
        typedef ColCfg<m1> Cfg;                                                                      \
        dim3 grid(1024 / Cfg::C, rows, (!INV && a.x2 != nullptr) ? 2 : 1), block(Cfg::THREADS);     \
        if (INV) HY_LAUNCH((col_inv_kernel<m1, DT>), grid, block, Cfg::LDS, stream, a);              \
        else HY_LAUNCH((col_fwd_kernel<m1, DT>), grid, block, Cfg::LDS, stream, a);                  \
        break;                                                                                       \
    }
    switch (M1) {
        HY_COL_CASE(1)
        HY_COL_CASE(2)
        HY_COL_CASE(3)
        HY_COL_CASE(4)
        HY_COL_CASE(5)
        HY_COL_CASE(6)
        HY_COL_CASE(7)
        HY_COL_CASE(8)
        HY_COL_CASE(10)
        HY_COL_CASE(12)
        HY_COL_CASE(14)
        HY_COL_CASE(16)
        HY_COL_CASE(20)
        HY_COL_CASE(24)
        HY_COL_CASE(28)
        HY_COL_CASE(32)
        HY_COL_CASE(64)
        HY_COL_CASE(96)
        HY_COL_CASE(128)
        HY_COL_CASE(160)
        HY_COL_CASE(192)
        HY_COL_CASE(224)
        HY_COL_CASE(256)
        HY_COL_CASE(320)
        HY_COL_CASE(384)
        HY_COL_CASE(448)
        HY_COL_CASE(512)
        HY_COL_CASE(640)
        HY_COL_CASE(768)
        HY_COL_CASE(1024)
        default: return HYENA_ERR_UNSUPPORTED_L;
    }
#undef HY_COL_CASE
    return hy_launch_error() ? HYENA_ERR_LAUNCH : HYENA_OK;
}

template <bool INV>
int launch_col(int dtype, int M1, const ColArgs& a, int rows, void* stream) {
    if (rows <= 0) return HYENA_OK;
    switch (dtype) {
        case HYENA_F32: return launch_col_dt<DT_F32, INV>(M1, a, rows, stream);
        case HYENA_BF16: return launch_col_dt<DT_BF16, INV>(M1, a, rows, stream);
        case HYENA_F16: return launch_col_dt<DT_F16, INV>(M1, a, rows, stream);
        default: return HYENA_ERR_BAD_ARG;
    }
}

const size_t ROW_SMEM = 2 * ROW_LDS * sizeof(c32);

const size_t ROW0_SMEM = ROW0_LDS * sizeof(c32);

// rows 0 and (for even M1) M1/2 are their own partners and go through the row0_* kernels; the regular pairs
// (k1, M1 - k1), k1 = 1 .. (M1 - 1) / 2, through row_*.
int self_paired_rows(int M1) { return (M1 >= 2 && M1 % 2 == 0) ? 2 : 1; }
int regular_pairs(int M1) { return (M1 - 1) / 2; }

// The self-paired rows (k1 = 0 and M1 / 2: 2 / M1 of the work) are dependent chains of ~25 us per launch on a grid that fills a fraction of
// the chip -- three serial launches of them were 3 - 4 % of a step at L = 160000 / 450560 (profiles/r3a_kernel_tables.md).  They touch rows
// no other row kernel touches, so they run BESIDE the regular pairs on a side stream: fork after the column transforms, join before the
// inverse column transforms.  The side stream and its two events are created once per calling thread and device (the only state this
// library keeps); inside a hipGraph capture the fork / join become graph edges.  HYENA_FFTCONV_SIDE=0 keeps everything on the caller's stream.
#ifndef HIPEMU
struct SideStream {
    bool tried = false;
    hipStream_t s = nullptr;
    hipEvent_t fork = nullptr, join = nullptr;
};
static SideStream* side_for(void* stream) {
    static const bool enabled = [] { const char* e = std::getenv("HYENA_FFTCONV_SIDE"); return !(e != nullptr && e[0] == '0'); }();
    if (!enabled) return nullptr;
    static thread_local SideStream tab[32];
    int cur = 0;
    if (hipGetDevice(&cur) != hipSuccess) return nullptr;
    int dev = cur;
    hipDevice_t sd;
    if (stream != nullptr && hipStreamGetDevice((hipStream_t)stream, &sd) == hipSuccess) dev = (int)sd;
    if (dev < 0 || dev >= 32) return nullptr;
    SideStream& t = tab[dev];
    if (!t.tried) {
        t.tried = true;
        if (dev != cur) (void)hipSetDevice(dev);
        const bool ok = hipStreamCreateWithFlags(&t.s, hipStreamNonBlocking) == hipSuccess &&
                        hipEventCreateWithFlags(&t.fork, hipEventDisableTiming) == hipSuccess &&
                        hipEventCreateWithFlags(&t.join, hipEventDisableTiming) == hipSuccess;
        if (dev != cur) (void)hipSetDevice(cur);
        if (!ok) { t.s = nullptr; (void)hipGetLastError(); }
    }
    return t.s != nullptr ? &t : nullptr;
}
// fork: the side stream waits for everything the caller's stream holds so far; returns the stream the self-paired rows go to
static void* side_fork(void* stream, SideStream** out) {
    SideStream* t = side_for(stream);
    *out = nullptr;
    if (t == nullptr) return stream;
    if (hipEventRecord(t->fork, (hipStream_t)stream) != hipSuccess || hipStreamWaitEvent(t->s, t->fork, 0) != hipSuccess) {
        (void)hipGetLastError();
        return stream;
    }
    *out = t;
    return t->s;
}
static void side_mark(SideStream* t) {                  // the side work is complete behind this point of the side stream
    if (t != nullptr) (void)hipEventRecord(t->join, t->s);
}
static void side_join(void* stream, SideStream* t) {    // the caller's stream continues only after it
    if (t != nullptr) (void)hipStreamWaitEvent((hipStream_t)stream, t->join, 0);
}
#else
struct SideStream {};
static void* side_fork(void* stream, SideStream** out) { *out = nullptr; return stream; }
static void side_mark(SideStream*) {}
static void side_join(void*, SideStream*) {}
#endif

template <int MODE>
int launch_row_prod2(const RowArgs& a, void* stream) {
    SideStream* sd;
    void* s0 = regular_pairs(a.M1) > 0 ? side_fork(stream, &sd) : (sd = nullptr, stream);
    HY_LAUNCH((row0_prod2_kernel<MODE>), dim3(self_paired_rows(a.M1), (a.inner + 1) / 2, a.B), dim3(64), ROW0_SMEM, s0, a);
    side_mark(sd);
    if (regular_pairs(a.M1) > 0)
        HY_LAUNCH((row_prod2_kernel<MODE>), dim3(regular_pairs(a.M1), a.inner), dim3(64), ROW_SMEM, stream, a);
    side_join(stream, sd);
    return hy_launch_error() ? HYENA_ERR_LAUNCH : HYENA_OK;
}

// From this batch size on the regular row pairs of the backward run as row_dk_kernel (+ row_prod2_kernel<MODE_CORR> for
// du) instead of the fused row_bwd_kernel (see row_dk_kernel).
const int ROW_BWD_SPLIT_BATCH = 2;     // measured: B = 1 fused 6.62 vs split 7.05 ms (L = 2^20); B = 2: 2.11 vs 2.09 (L = 160000);
                                       // B = 3: 0.690 vs 0.661, B = 8: 1.56 vs 1.44 ms (L = 32768)

template <bool DO_DU>
int launch_row_bwd(const RowArgs& a, void* stream) {
    SideStream* sd;
    void* s0 = regular_pairs(a.M1) > 0 ? side_fork(stream, &sd) : (sd = nullptr, stream);
    HY_LAUNCH((row0_bwd_kernel<DO_DU>), dim3(self_paired_rows(a.M1), (a.inner + 1) / 2, a.B), dim3(64), ROW0_SMEM, s0, a);
    HY_LAUNCH(row0_dk_reduce_kernel, dim3(self_paired_rows(a.M1), a.inner), dim3(256), 0, s0, a);
    side_mark(sd);
    if (regular_pairs(a.M1) > 0) {
        const dim3 grid(regular_pairs(a.M1), a.inner);
        if (a.B >= ROW_BWD_SPLIT_BATCH) {
            HY_LAUNCH(row_dk_kernel, grid, dim3(64), ROW_SMEM, stream, a);       // reads the dout rows before du replaces them
            if (DO_DU) {
                RowArgs c = a;                                                    // du = corr(dout, k) + bias, in place
                c.U = a.K; c.u_bstride = 0; c.Y = a.X; c.y_bstride = a.x_bstride;
                HY_LAUNCH((row_prod2_kernel<MODE_CORR>), grid, dim3(64), ROW_SMEM, stream, c);
            }
        } else {
            HY_LAUNCH((row_bwd_kernel<DO_DU>), grid, dim3(64), ROW_SMEM, stream, a);
        }
    }
    side_join(stream, sd);
    return hy_launch_error() ? HYENA_ERR_LAUNCH : HYENA_OK;
}

size_t elem_size(int dtype) { return dtype == HYENA_F32 ? 4 : 2; }

// Intermediates of one chunk.  Measured on MI355X (profiles/mall_bw_r1.txt, chunk sweeps in profiles/): keeping a
// chunk inside the 256 MiB Infinity Cache buys little (write-then-read runs at 6.0-6.4 TB/s there vs 5.0 TB/s from
// HBM), while every kernel launch pays a ~30 us ramp/tail at the ~35 us latency of one row workgroup -- so chunks
// are made as large as a generous workspace allows (all 256 channels at L = 2^20, B = 1: 8 GiB of 288).
const size_t CACHE_BUDGET = (size_t)8 << 30;

}  // namespace

extern "C" {

int hyena_fftconv_abi_version(void) { return 3; }

const char* hyena_fftconv_error_string(int status) {
    switch (status) {
        case HYENA_OK: return "ok";
        case HYENA_ERR_BAD_ARG: return "bad argument (null pointer, non-positive size or unknown dtype)";
        case HYENA_ERR_UNSUPPORTED_L: return "unsupported sequence length (1 <= L <= 1048576)";
        case HYENA_ERR_WORKSPACE: return "workspace too small";
        case HYENA_ERR_LAUNCH: return "kernel launch failed";
        default: return "unknown status";
    }
}

int hyena_fftconv_plan(int L) {
    Plan p;
    if (!make_plan(L, &p)) return HYENA_PLAN_NONE;
    return p.R ? HYENA_PLAN_ONCHIP : HYENA_PLAN_TWO_LEVEL;
}

int hyena_fftconv_fft_size(int L) {
    Plan p;
    return make_plan(L, &p) ? p.M : 0;
}

size_t hyena_fftconv_table_bytes(int L) {
    Plan p;
    if (!make_plan(L, &p)) return 0;
    if (p.R) return oc::table_entries(p.R) * sizeof(c32);
    return (size_t)4096 * sizeof(c32);
}

int hyena_fftconv_init_tables(void* d_tables, int L, void* stream) {
    (void)stream;
    Plan p;
    if (d_tables == nullptr) return HYENA_ERR_BAD_ARG;
    if (!make_plan(L, &p)) return HYENA_ERR_UNSUPPORTED_L;
    std::vector<c32> h(p.R ? oc::table_entries(p.R) : 4096);
    for (auto& e : h) { e.x = 1.0f; e.y = 0.0f; }
    if (p.R) {
        oc::build_tables(p.R, reinterpret_cast<float*>(h.data()));
#ifdef HIPEMU
        memcpy(d_tables, h.data(), h.size() * sizeof(c32));
#else
        // stream-ordered with the kernels that will read the table (pageable source: staged before the call returns)
        if (hipMemcpyAsync(d_tables, h.data(), h.size() * sizeof(c32), hipMemcpyHostToDevice, (hipStream_t)stream) != hipSuccess)
            return HYENA_ERR_LAUNCH;
        if (hipStreamSynchronize((hipStream_t)stream) != hipSuccess) return HYENA_ERR_LAUNCH;
#endif
        return HYENA_OK;
    }
    const double tau = 6.283185307179586476925286766559;
    for (int i = 0; i < 1024; ++i) {
        double a = -tau * (double)i / (double)p.M;
        h[i].x = (float)std::cos(a);
        h[i].y = (float)std::sin(a);
        double r = -tau * (double)i / 1024.0;
        h[2048 + i].x = (float)std::cos(r);
        h[2048 + i].y = (float)std::sin(r);
    }
    for (int s = 0; s < 32; ++s)
        for (int j = 0; j < 32; ++j) {
            double r = -tau * (double)(s * j) / 1024.0;
            h[3072 + s * 32 + j].x = (float)std::cos(r);
            h[3072 + s * 32 + j].y = (float)std::sin(r);
        }
    for (int i = 0; i < p.M1; ++i) {
        double a = -tau * (double)i / (double)p.M1;
        h[1024 + i].x = (float)std::cos(a);
        h[1024 + i].y = (float)std::sin(a);
    }
#ifdef HIPEMU
    memcpy(d_tables, h.data(), h.size() * sizeof(c32));
#else
    if (hipMemcpyAsync(d_tables, h.data(), h.size() * sizeof(c32), hipMemcpyHostToDevice, (hipStream_t)stream) != hipSuccess)
        return HYENA_ERR_LAUNCH;
    if (hipStreamSynchronize((hipStream_t)stream) != hipSuccess) return HYENA_ERR_LAUNCH;
#endif
    return HYENA_OK;
}

int hyena_fftconv_default_chunk(int B, int D, int L, int backward) {
    Plan p;
    if (!make_plan(L, &p) || B < 1 || D < 1) return 0;
    if (p.R) return D;
    const size_t per_channel = (size_t)(backward ? 2 * B + 2 : B + 1) * p.M * sizeof(c32);
    size_t c = CACHE_BUDGET / per_channel;
    const size_t cmax = ((size_t)1 << 28) / p.M;     // a chunk's [chunk][M] slab stays below 2 GiB (32-bit buffer offsets)
    if (c > cmax) c = cmax;
    if (c < 1) c = 1;
    if (c > (size_t)D) c = D;
    return (int)c;
}

size_t hyena_fftconv_workspace_bytes(int B, int D, int L, int backward, int chunk) {
    Plan p;
    if (!make_plan(L, &p) || B < 1 || D < 1) return 0;
    if (p.R)      // the filter spectrum is this path's only intermediate (+ dk's per-slice partial rows when D < number of CUs)
        return oc::spectrum_bytes(D, p.R) + (backward ? oc::dk_partial_bytes(p.R, B, D, L) + (oc::dk1_ok(p.R, B) ? oc::spectrum_bytes(D, p.R) : 0) : 0);
    if (chunk <= 0) chunk = hyena_fftconv_default_chunk(B, D, L, backward);
    if (chunk > D) chunk = D;
    if ((size_t)chunk * p.M > ((size_t)1 << 28)) chunk = (int)(((size_t)1 << 28) / p.M);
    return (size_t)(backward ? 2 * B + 2 : B + 1) * chunk * p.M * sizeof(c32) +
           (backward ? (size_t)B * chunk * 2 * 1024 * sizeof(c32) : 0);      // + per-batch dk rows of the self-paired rows
}

// saved-spectrum buffer (optional, persists from forward to backward): Wk [D][M] | Wu [B][D][M], complex64
static inline c32* saved_wk(void* saved) { return reinterpret_cast<c32*>(saved); }
static inline c32* saved_wu(void* saved, int D, int M) { return reinterpret_cast<c32*>(saved) + (size_t)D * M; }

size_t hyena_fftconv_saved_bytes(int B, int D, int L) {
    Plan p;
    if (!make_plan(L, &p) || B < 1 || D < 1) return 0;
    if (p.R) return oc::spectrum_bytes(D, p.R);              // H only: the backward re-transforms u on chip
    return (size_t)(B + 1) * D * p.M * sizeof(c32);
}

static int fwd_impl(const void* u, const float* k, const float* bias, void* out, int B, int D, int L, int ldx, int ldk, int dtype,
                    const void* d_tables, void* workspace, size_t workspace_bytes, int chunk, void* saved,
                    size_t saved_bytes, void* stream) {
    Plan p;
    if (u == nullptr || k == nullptr || out == nullptr || d_tables == nullptr || workspace == nullptr || B < 1 ||
        D < 1 || ldx < L || ldk < L || (dtype != HYENA_F32 && dtype != HYENA_BF16 && dtype != HYENA_F16))
        return HYENA_ERR_BAD_ARG;
    const oc::Pitch ld = {ldx, ldk};
    if (!make_plan(L, &p)) return HYENA_ERR_UNSUPPORTED_L;
    if (p.R) {
        if ((size_t)D * p.M * sizeof(c32) >= ((size_t)1 << 32)) return HYENA_ERR_BAD_ARG;       // 32-bit buffer offsets into H
        if (workspace_bytes < oc::spectrum_bytes(D, p.R)) return HYENA_ERR_WORKSPACE;
        if (saved != nullptr && saved_bytes < oc::spectrum_bytes(D, p.R)) return HYENA_ERR_WORKSPACE;
        if (oc::small_ok(p.R, B, D, L, ld, dtype))          // one launch: the filter transform rides in the convolution kernel
            return oc::launch_small_fwd(p.R, u, out, k, bias, saved, d_tables, B, D, L, ld, dtype, stream);
        void* H = saved ? saved : workspace;
        int st = oc::launch_spec(p.R, k, bias, H, d_tables, D, L, ld, stream);
        if (st) return st;
        return oc::launch_conv(p.R, u, out, H, d_tables, B, D, L, ld, dtype, 0, stream);
    }
    if (chunk <= 0) chunk = hyena_fftconv_default_chunk(B, D, L, 0);
    if (chunk > D) chunk = D;
    if ((size_t)chunk * p.M > ((size_t)1 << 28)) chunk = (int)(((size_t)1 << 28) / p.M);
    if (workspace_bytes < hyena_fftconv_workspace_bytes(B, D, L, 0, chunk)) return HYENA_ERR_WORKSPACE;
    if (saved != nullptr && saved_bytes < hyena_fftconv_saved_bytes(B, D, L)) return HYENA_ERR_WORKSPACE;

    const Tables tab = tables_from(d_tables);
    c32* wsWk = reinterpret_cast<c32*>(workspace);              // [chunk][M]      column-transformed filter
    c32* wsW = wsWk + (size_t)chunk * p.M;                      // [B][chunk][M]   column-transformed activations
    const size_t es = elem_size(dtype);
    int st;
    for (int d0 = 0; d0 < D; d0 += chunk) {
        const int cd = (D - d0 < chunk) ? D - d0 : chunk;
        // with a saved-spectrum buffer the column transforms land there (kept for the backward) and the row
        // kernel writes its product into the workspace instead of transforming in place
        c32* Wk = saved ? saved_wk(saved) + (size_t)d0 * p.M : wsWk;
        c32* Wu = saved ? saved_wu(saved, D, p.M) + (size_t)d0 * p.M : wsW;
        const int ub = saved ? D : cd;
        ColArgs ck;
        ck.x = k + (size_t)d0 * ldk; ck.W = Wk; ck.tab = tab; ck.L = L; ck.inner = cd; ck.w_bstride = cd;
        ck.outer_stride = 0; ck.inner_stride = ldk; ck.aux0 = nullptr; ck.x2 = nullptr; ck.W2 = nullptr;
        if ((st = launch_col<false>(HYENA_F32, p.M1, ck, cd, stream))) return st;
        ColArgs cu;
        cu.x = reinterpret_cast<const char*>(u) + (size_t)d0 * ldx * es; cu.W = Wu; cu.tab = tab; cu.L = L; cu.inner = cd;
        cu.w_bstride = ub;
        cu.outer_stride = (long)D * ldx; cu.inner_stride = ldx; cu.aux0 = nullptr; cu.x2 = nullptr; cu.W2 = nullptr;
        if ((st = launch_col<false>(dtype, p.M1, cu, B * cd, stream))) return st;
        RowArgs rc;
        rc.X = Wu; rc.U = Wk; rc.S = nullptr; rc.S0 = nullptr; rc.bias = bias ? bias + d0 : nullptr; rc.K = nullptr; rc.Y = wsW;
        rc.x_bstride = ub; rc.u_bstride = 0; rc.y_bstride = cd; rc.tab = tab;
        rc.M1 = p.M1; rc.inner = cd; rc.B = B; rc.scale = 1.0f / (float)p.M;
        if ((st = launch_row_prod2<MODE_CONV>(rc, stream))) return st;
        ColArgs co = cu;
        co.x = reinterpret_cast<char*>(out) + (size_t)d0 * ldx * es; co.W = wsW; co.w_bstride = cd;
        if ((st = launch_col<true>(dtype, p.M1, co, B * cd, stream))) return st;
    }
    return HYENA_OK;
}

static int bwd_impl(const void* dout, const void* u, const float* k, const float* bias, void* du, float* dk,
                    float* dbias, int B, int D, int L, int ldx, int ldk, int dtype, const void* d_tables, void* workspace,
                    size_t workspace_bytes, int chunk, const void* saved_c, size_t saved_bytes, void* stream) {
    Plan p;
    void* saved = const_cast<void*>(saved_c);
    const oc::Pitch ld = {ldx, ldk};
    if (dout == nullptr || (u == nullptr && saved == nullptr) || (k == nullptr && saved == nullptr) ||
        d_tables == nullptr || workspace == nullptr || B < 1 || D < 1 || ldx < L || ldk < L ||
        (dtype != HYENA_F32 && dtype != HYENA_BF16 && dtype != HYENA_F16))
        return HYENA_ERR_BAD_ARG;
    if (dbias != nullptr && dk == nullptr) return HYENA_ERR_BAD_ARG;
    if (!make_plan(L, &p)) return HYENA_ERR_UNSUPPORTED_L;
    if (p.R) {
        if (dk != nullptr && u == nullptr) return HYENA_ERR_BAD_ARG;        // this path keeps no spectrum of u: it re-reads u
        if (p.R == 1 && (size_t)B * D * ldx * elem_size(dtype) >= ((size_t)1 << 32)) return HYENA_ERR_BAD_ARG;   // 32-bit row offsets (dk, T = 32)
        if ((size_t)D * p.M * sizeof(c32) >= ((size_t)1 << 32)) return HYENA_ERR_BAD_ARG;
        if (workspace_bytes < hyena_fftconv_workspace_bytes(B, D, L, 1, chunk)) return HYENA_ERR_WORKSPACE;
        if (saved != nullptr && saved_bytes < oc::spectrum_bytes(D, p.R)) return HYENA_ERR_WORKSPACE;
        void* partials = reinterpret_cast<char*>(workspace) + oc::spectrum_bytes(D, p.R);
        int st;
        if ((du != nullptr || dk != nullptr) && oc::small_ok(p.R, B, D, L, ld, dtype)) {
            // du and dk from one launch.  Without the forward's spectrum the filter is transformed first by the forward kernel run
            // over an empty batch: the same code, hence the same bits, as the H the saved path reads.
            const void* H = saved;
            if (H == nullptr) {
                if ((st = oc::launch_small_fwd(p.R, nullptr, nullptr, k, bias, workspace, d_tables, 0, D, L, ld, dtype, stream))) return st;
                H = workspace;
            }
            return oc::launch_small_bwd(p.R, dout, u, du, dk, dbias, H, d_tables, B, D, L, ld, dtype, stream);
        }
        if (du != nullptr) {
            const void* H = saved;
            if (H == nullptr) {
                if ((st = oc::launch_spec(p.R, k, bias, workspace, d_tables, D, L, ld, stream))) return st;
                H = workspace;
            }
            if (dk != nullptr && oc::dkdu_ok(p.R, B))        // du rides on dk's transform of dout (M = 16384 by default: onchip_dk.hip)
                return oc::launch_dkdu(p.R, dout, u, du, H, dk, dbias, partials, d_tables, B, D, L, ld, dtype, stream);
            if ((st = oc::launch_conv(p.R, dout, du, H, d_tables, B, D, L, ld, dtype, 1, stream))) return st;
        }
        if (dk != nullptr && oc::dk1_ok(p.R, B)) {          // B = 1: the spectrum of u behind the partial rows, then conv with the conjugate
            void* uspec = reinterpret_cast<char*>(partials) + oc::dk_partial_bytes(p.R, B, D, L);
            return oc::launch_dk1(p.R, dout, u, dk, dbias, uspec, d_tables, D, L, ld, dtype, stream);
        }
        if (dk != nullptr && (st = oc::launch_dk(p.R, dout, u, dk, dbias, partials, d_tables, B, D, L, ld, dtype, stream))) return st;
        return HYENA_OK;
    }
    if (chunk <= 0) chunk = hyena_fftconv_default_chunk(B, D, L, 1);
    if (chunk > D) chunk = D;
    if ((size_t)chunk * p.M > ((size_t)1 << 28)) chunk = (int)(((size_t)1 << 28) / p.M);
    if (workspace_bytes < hyena_fftconv_workspace_bytes(B, D, L, 1, chunk)) return HYENA_ERR_WORKSPACE;
    if (saved != nullptr && saved_bytes < hyena_fftconv_saved_bytes(B, D, L)) return HYENA_ERR_WORKSPACE;

    const Tables tab = tables_from(d_tables);
    c32* wsWk = reinterpret_cast<c32*>(workspace);              // [chunk][M]      column-transformed filter
    c32* Sdk = wsWk + (size_t)chunk * p.M;                      // [chunk][M]      dk rows (batch sum) -> packed dk
    c32* Wg = Sdk + (size_t)chunk * p.M;                        // [B][chunk][M]   dout rows -> du rows
    c32* wsWu = Wg + (size_t)B * chunk * p.M;                   // [B][chunk][M]   u rows
    c32* S0 = wsWu + (size_t)B * chunk * p.M;                   // [B][chunk][2][1024] per-batch dk rows 0 and M1/2
    const size_t es = elem_size(dtype);
    int st;
    for (int d0 = 0; d0 < D; d0 += chunk) {
        const int cd = (D - d0 < chunk) ? D - d0 : chunk;
        c32* Wk = saved ? saved_wk(saved) + (size_t)d0 * p.M : wsWk;
        c32* Wu = saved ? saved_wu(saved, D, p.M) + (size_t)d0 * p.M : wsWu;
        const int ub = saved ? D : cd;
        ColArgs cg;
        cg.x = reinterpret_cast<const char*>(dout) + (size_t)d0 * ldx * es; cg.W = Wg; cg.tab = tab; cg.L = L; cg.inner = cd;
        cg.w_bstride = cd;
        cg.outer_stride = (long)D * ldx; cg.inner_stride = ldx; cg.aux0 = nullptr; cg.x2 = nullptr; cg.W2 = nullptr;
        ColArgs cgu = cg;                                        // dout (and u, when dk is wanted) in ONE launch
        if (dk != nullptr && !saved) { cgu.x2 = reinterpret_cast<const char*>(u) + (size_t)d0 * ldx * es; cgu.W2 = Wu; }
        if ((st = launch_col<false>(dtype, p.M1, cgu, B * cd, stream))) return st;
        ColArgs ck;
        ck.x = saved ? nullptr : k + (size_t)d0 * ldk; ck.W = Wk; ck.tab = tab; ck.L = L; ck.inner = cd; ck.w_bstride = cd;
        ck.outer_stride = 0; ck.inner_stride = ldk; ck.aux0 = nullptr; ck.x2 = nullptr; ck.W2 = nullptr;
        if (du != nullptr && !saved && (st = launch_col<false>(HYENA_F32, p.M1, ck, cd, stream))) return st;
        RowArgs rb;
        rb.X = Wg; rb.U = Wu; rb.S = Sdk; rb.S0 = S0; rb.bias = bias ? bias + d0 : nullptr; rb.K = Wk; rb.Y = Wg;
        rb.x_bstride = cd; rb.u_bstride = ub; rb.y_bstride = cd; rb.tab = tab;
        rb.M1 = p.M1; rb.inner = cd; rb.B = B; rb.scale = 1.0f / (float)p.M;
        if (dk != nullptr) {
            if (du != nullptr) st = launch_row_bwd<true>(rb, stream);
            else st = launch_row_bwd<false>(rb, stream);
            if (st) return st;
            ColArgs cdk;
            cdk.x = dk + (size_t)d0 * ldk; cdk.W = Sdk; cdk.tab = tab; cdk.L = L; cdk.inner = cd; cdk.w_bstride = cd;
            cdk.outer_stride = 0; cdk.inner_stride = ldk; cdk.aux0 = dbias ? dbias + d0 : nullptr; cdk.x2 = nullptr; cdk.W2 = nullptr;
            if ((st = launch_col<true>(HYENA_F32, p.M1, cdk, cd, stream))) return st;
        } else {
            RowArgs rc = rb;                                     // du only: X = dout rows, H = filter rows, corr
            rc.U = Wk; rc.K = nullptr; rc.S = nullptr;
            if ((st = launch_row_prod2<MODE_CORR>(rc, stream))) return st;
        }
        if (du != nullptr) {
            ColArgs co = cg;
            co.x = reinterpret_cast<char*>(du) + (size_t)d0 * ldx * es;
            if ((st = launch_col<true>(dtype, p.M1, co, B * cd, stream))) return st;
        }
    }
    return HYENA_OK;
}

int hyena_fftconv_fwd(const void* u, const float* k, const float* bias, void* out, int B, int D, int L, int dtype,
                      const void* d_tables, void* workspace, size_t workspace_bytes, int chunk, void* stream) {
    return fwd_impl(u, k, bias, out, B, D, L, L, L, dtype, d_tables, workspace, workspace_bytes, chunk, nullptr, 0, stream);
}

int hyena_fftconv_fwd_save(const void* u, const float* k, const float* bias, void* out, int B, int D, int L, int dtype,
                           const void* d_tables, void* workspace, size_t workspace_bytes, int chunk, void* saved,
                           size_t saved_bytes, void* stream) {
    if (saved == nullptr) return HYENA_ERR_BAD_ARG;
    return fwd_impl(u, k, bias, out, B, D, L, L, L, dtype, d_tables, workspace, workspace_bytes, chunk, saved, saved_bytes, stream);
}

int hyena_fftconv_fwd_ld(const void* u, const float* k, const float* bias, void* out, int B, int D, int L, int ldx, int ldk, int dtype,
                         const void* d_tables, void* workspace, size_t workspace_bytes, int chunk, void* saved, size_t saved_bytes,
                         void* stream) {
    return fwd_impl(u, k, bias, out, B, D, L, ldx, ldk, dtype, d_tables, workspace, workspace_bytes, chunk, saved, saved_bytes, stream);
}

int hyena_fftconv_bwd_ld(const void* dout, const void* u, const float* k, const float* bias, void* du, float* dk, float* dbias, int B,
                         int D, int L, int ldx, int ldk, int dtype, const void* d_tables, void* workspace, size_t workspace_bytes,
                         int chunk, const void* saved, size_t saved_bytes, void* stream) {
    return bwd_impl(dout, u, k, bias, du, dk, dbias, B, D, L, ldx, ldk, dtype, d_tables, workspace, workspace_bytes, chunk, saved,
                    saved_bytes, stream);
}

int hyena_fftconv_bwd(const void* dout, const void* u, const float* k, const float* bias, void* du, float* dk,
                      float* dbias, int B, int D, int L, int dtype, const void* d_tables, void* workspace,
                      size_t workspace_bytes, int chunk, void* stream) {
    return bwd_impl(dout, u, k, bias, du, dk, dbias, B, D, L, L, L, dtype, d_tables, workspace, workspace_bytes, chunk, nullptr, 0,
                    stream);
}

int hyena_fftconv_bwd_saved(const void* dout, const void* u, const float* bias, void* du, float* dk, float* dbias, int B, int D,
                            int L, int dtype, const void* d_tables, void* workspace, size_t workspace_bytes, int chunk,
                            const void* saved, size_t saved_bytes, void* stream) {
    if (saved == nullptr) return HYENA_ERR_BAD_ARG;
    return bwd_impl(dout, u, nullptr, bias, du, dk, dbias, B, D, L, L, L, dtype, d_tables, workspace, workspace_bytes, chunk,
                    saved, saved_bytes, stream);
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------------------------
// fused mixer shell (include/hyena_mixer.h)
// ---------------------------------------------------------------------------------------------------------------
namespace {
bool mix_ok(const void* x, const float* w, const float* b, int B, int L, int Lx, int D, int dtype) {
    return x != nullptr && w != nullptr && b != nullptr && B >= 1 && L >= 1 && Lx >= L && D >= 1 &&
           (dtype == HYENA_F32 || dtype == HYENA_BF16 || dtype == HYENA_F16);
}
dim3 mix_grid(int B, int L, int D) { return dim3((L + MIX_RUN - 1) / MIX_RUN, (D + 63) / 64, B); }
const size_t MIX_SMEM = 64 * 65 * sizeof(float);

// the wide-access kernels need full 64-channel tiles and 16-byte-aligned channel rows
bool mix_wide(int D) { return D % 64 == 0; }
const size_t MW_SMEM_PRE_FWD = ((MW_TP + 2) * 2 * MW_CS + MW_TC * MW_CS) * sizeof(float);
const size_t MW_SMEM_POST_FWD = ((MW_TP + 2) * MW_CS + 2 * MW_TC * MW_CS) * sizeof(float);
const size_t MW_SMEM_POST_BWD = ((MW_TP + 4) * MW_CS + (MW_TP + 2) * MW_CS + MW_TC * MW_YS) * sizeof(float);
const size_t MW_SMEM_PRE_BWD = (2 * (MW_TP + 4) * MW_CS + 2 * MW_TC * MW_YS) * sizeof(float);

#define HY_MIXW_DISPATCH(kernel, smem)                                                               \
    do {                                                                                             \
        switch (dtype) {                                                                             \
            case HYENA_F32: HY_LAUNCH((kernel<DT_F32>), mix_grid(B, L, D), dim3(MW_THREADS), smem, stream, a); break;   \
            case HYENA_BF16: HY_LAUNCH((kernel<DT_BF16>), mix_grid(B, L, D), dim3(MW_THREADS), smem, stream, a); break; \
            default: HY_LAUNCH((kernel<DT_F16>), mix_grid(B, L, D), dim3(MW_THREADS), smem, stream, a); break;          \
        }                                                                                            \
    } while (0)

#define HY_MIX_DISPATCH(kernel, smem)                                                                \
    do {                                                                                             \
        switch (dtype) {                                                                             \
            case HYENA_F32: HY_LAUNCH((kernel<DT_F32>), mix_grid(B, L, D), dim3(64), smem, stream, a); break;  \
            case HYENA_BF16: HY_LAUNCH((kernel<DT_BF16>), mix_grid(B, L, D), dim3(64), smem, stream, a); break; \
            default: HY_LAUNCH((kernel<DT_F16>), mix_grid(B, L, D), dim3(64), smem, stream, a); break;          \
        }                                                                                            \
    } while (0)
}  // namespace

extern "C" {
int hyena_mixer_pre_fwd(const void* x, const float* w, const float* b, void* vg, int B, int L, int Lx, int D, int dtype,
                        void* stream) {
    if (!mix_ok(x, w, b, B, L, Lx, D, dtype) || vg == nullptr) return HYENA_ERR_BAD_ARG;
    MixArgs a;
    a.x = x; a.w = w; a.b = b; a.a0 = vg; a.a1 = nullptr; a.a2 = nullptr; a.dx = nullptr; a.part = nullptr;
    a.B = B; a.L = L; a.D = D; a.Lx = Lx;
    if (mix_wide(D)) HY_MIXW_DISPATCH(mixer_pre_fwd_wide_kernel, MW_SMEM_PRE_FWD);
    else HY_MIX_DISPATCH(mixer_pre_fwd_kernel, MIX_SMEM);
    return hy_launch_error() ? HYENA_ERR_LAUNCH : HYENA_OK;
}

int hyena_mixer_post_fwd(const void* y, const void* x, const float* w, const float* b, void* z, int B, int L, int Lx, int D,
                         int dtype, void* stream) {
    if (!mix_ok(x, w, b, B, L, Lx, D, dtype) || y == nullptr || z == nullptr) return HYENA_ERR_BAD_ARG;
    MixArgs a;
    a.x = x; a.w = w; a.b = b; a.a0 = const_cast<void*>(y); a.a1 = z; a.a2 = nullptr; a.dx = nullptr; a.part = nullptr;
    a.B = B; a.L = L; a.D = D; a.Lx = Lx;
    if (mix_wide(D)) HY_MIXW_DISPATCH(mixer_post_fwd_wide_kernel, MW_SMEM_POST_FWD);
    else HY_MIX_DISPATCH(mixer_post_fwd_kernel, MIX_SMEM);
    return hy_launch_error() ? HYENA_ERR_LAUNCH : HYENA_OK;
}

size_t hyena_mixer_partial_floats(int B, int L, int D) {
    if (B < 1 || L < 1 || D < 1) return 0;
    return (size_t)B * ((L + MIX_RUN - 1) / MIX_RUN) * 3 * D * 4;
}

int hyena_mixer_post_bwd(const void* dz, const void* y, const void* x, const float* w, const float* b, void* dy, void* dx,
                         float* part, int B, int L, int Lx, int D, int dtype, void* stream) {
    if (!mix_ok(x, w, b, B, L, Lx, D, dtype) || dz == nullptr || y == nullptr || dy == nullptr || dx == nullptr ||
        part == nullptr)
        return HYENA_ERR_BAD_ARG;
    MixArgs a;
    a.x = x; a.w = w; a.b = b; a.a0 = const_cast<void*>(y); a.a1 = const_cast<void*>(dz); a.a2 = dy; a.dx = dx; a.part = part;
    a.B = B; a.L = L; a.D = D; a.Lx = Lx;
    if (mix_wide(D)) HY_MIXW_DISPATCH(mixer_post_bwd_wide_kernel, MW_SMEM_POST_BWD);
    else HY_MIX_DISPATCH(mixer_post_bwd_kernel, 2 * MIX_SMEM);
    return hy_launch_error() ? HYENA_ERR_LAUNCH : HYENA_OK;
}

int hyena_mixer_pre_bwd(const void* dvg, const void* x, const float* w, const float* b, void* dx, float* part, int B, int L,
                        int Lx, int D, int dtype, void* stream) {
    if (!mix_ok(x, w, b, B, L, Lx, D, dtype) || dvg == nullptr || dx == nullptr || part == nullptr) return HYENA_ERR_BAD_ARG;
    MixArgs a;
    a.x = x; a.w = w; a.b = b; a.a0 = const_cast<void*>(dvg); a.a1 = nullptr; a.a2 = nullptr; a.dx = dx; a.part = part;
    a.B = B; a.L = L; a.D = D; a.Lx = Lx;
    if (mix_wide(D)) HY_MIXW_DISPATCH(mixer_pre_bwd_wide_kernel, MW_SMEM_PRE_BWD);
    else HY_MIX_DISPATCH(mixer_pre_bwd_kernel, MIX_SMEM);
    return hy_launch_error() ? HYENA_ERR_LAUNCH : HYENA_OK;
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------------------------
// fused implicit filter (include/hyena_filter.h)
// ---------------------------------------------------------------------------------------------------------------
namespace {
// kernels that stage the whole weight set in LDS need more than the 64 KiB a launch may ask for by default
// (per device: the attribute is re-applied whenever the calling thread's current device changes)
template <typename K>
void flt_allow_lds(K kernel, size_t bytes, int* device_done) {
#ifndef HIPEMU
    int dev = -1;
    (void)hipGetDevice(&dev);
    if (*device_done == dev) return;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    *device_done = dev;
#else
    (void)kernel; (void)bytes; (void)device_done;
#endif
}

const size_t FLT_RED_SMEM = FLT_RED_J * FLT_RED_S * sizeof(float);

int flt_grid(int L) {
    const int n = (L + FLT_WG_POS - 1) / FLT_WG_POS;
    return n < FLT_MAX_WG ? n : FLT_MAX_WG;
}

size_t flt_part_floats(int D) {
    const int no = D > 128 ? D : 128;
    return (size_t)FLT_MAX_WG * ((size_t)no * FLT_O + 1024);
}

bool flt_params_ok(const hyena_filter_params* p) {
    return p != nullptr && p->z && p->t && p->w0 && p->b0 && p->w1 && p->b1 && p->w2 && p->b2 && p->w3 && p->freq &&
           (p->deltas || !p->modulate) && p->z_stride >= p->E && hyena_filter_supported(p->L, p->E, FLT_O, p->D);
}

template <int D>
void launch_filter_fwd(const FilterArgs& a, void* stream) {
    const int ntiles = (a.L + FLT_TP - 1) / FLT_TP;
    int grid = (ntiles + FLT_WAVES - 1) / FLT_WAVES;
    if (grid > FLT_MAX_WG) grid = FLT_MAX_WG;
    static thread_local int done_save = -1, done_plain = -1;
    if (a.acts != nullptr) {
        flt_allow_lds(filter_fwd_kernel<D, true>, FltFwdLds<D>::BYTES, &done_save);
        HY_LAUNCH((filter_fwd_kernel<D, true>), dim3(grid), dim3(FLT_THREADS), FltFwdLds<D>::BYTES, stream, a);
    } else {
        flt_allow_lds(filter_fwd_kernel<D, false>, FltFwdLds<D>::BYTES, &done_plain);
        HY_LAUNCH((filter_fwd_kernel<D, false>), dim3(grid), dim3(FLT_THREADS), FltFwdLds<D>::BYTES, stream, a);
    }
}

template <int NO, int NI, int MODE>
void launch_filter_layer_bwd(FilterBwdArgs a, float* part, float* dw, float* db, float* dfreq, bool first_freq, void* stream) {
    typedef FltBwdCfg<NO, NI> Cfg;
    const int grid = flt_grid(a.L);
    const int slots = grid * Cfg::KS;
    a.part_w = part;
    a.part_b = db != nullptr ? part + (size_t)slots * NO * NI : nullptr;
    a.part_f = part + (size_t)slots * NO * NI + (size_t)2 * slots * NO;
    static thread_local int done = -1;
    flt_allow_lds(filter_layer_bwd_kernel<NO, NI, MODE>, Cfg::BYTES, &done);
    HY_LAUNCH((filter_layer_bwd_kernel<NO, NI, MODE>), dim3(grid), dim3(FLT_THREADS), Cfg::BYTES, stream, a);
    const int nw = NO * a.ni;
    if (NI == a.ni) {
        HY_LAUNCH(filter_reduce_kernel, dim3((nw + FLT_RED_J - 1) / FLT_RED_J), dim3(256), FLT_RED_SMEM, stream, (const float*)a.part_w, dw, slots, nw, 0);
    } else {
        // layer 0 with E < 8: partial rows are 8 wide; reduce into a scratch tail of `part`, then the host-side
        // caller's dw0 gets the first E columns (done below by a strided second pass)
        float* tmp = a.part_f + (size_t)grid * FLT_WAVES * FLT_O;
        HY_LAUNCH(filter_reduce_kernel, dim3((NO * NI + FLT_RED_J - 1) / FLT_RED_J), dim3(256), FLT_RED_SMEM, stream, (const float*)a.part_w, tmp, slots, NO * NI, 0);
        HY_LAUNCH(filter_compact_kernel, dim3((nw + 255) / 256), dim3(256), 0, stream, (const float*)tmp, dw, NO, NI, a.ni);
    }
    RedBatch red;
    if (db != nullptr) red.add(a.part_b, db, 2 * slots, NO, NO, 0);
    if (MODE & FLT_ACT) red.add(a.part_f, dfreq, grid * FLT_WAVES, FLT_O, FLT_O, first_freq ? 0 : 1);
    if (red.jobs.njobs > 0) HY_LAUNCH(filter_reduce_multi_kernel, dim3(red.blocks()), dim3(256), FLT_RED_SMEM, stream, red.jobs);
}
}  // namespace

extern "C" {

int hyena_filter_supported(int L, int E, int order, int D) {
    return L >= 1 && L <= HYENA_MAX_L && E >= 1 && E <= FLT_E && order == FLT_O && (D == 64 || D == 128 || D == 256);
}

// Rows of the library-owned buffers (saved pre-activations, the backward's two gradient buffers) are pitched to 64 words whatever L is:
// the reference trainer's L = max_length - 1 is odd, and 16-byte accesses to packed rows of odd length are not aligned (round 5)
int hyena_filter_row_pitch(int L) { return L >= 1 ? (L + 63) & ~63 : 0; }

size_t hyena_filter_saved_bytes(int L) { return L >= 1 ? (size_t)3 * FLT_O * hyena_filter_row_pitch(L) * sizeof(float) : 0; }

size_t hyena_filter_workspace_bytes(int L, int D) {
    if (L < 1 || D < 1) return 0;
    return ((size_t)2 * FLT_O * hyena_filter_row_pitch(L) + flt_part_floats(D)) * sizeof(float);
}

int hyena_filter_fwd(const hyena_filter_params* p, float* k, float* saved, void* stream) {
    return hyena_filter_fwd_ld(p, k, p != nullptr ? p->L : 0, saved, stream);
}

int hyena_filter_bwd(const hyena_filter_params* p, const float* dk, const float* saved, const hyena_filter_grads* g,
                     void* workspace, size_t workspace_bytes, void* stream) {
    return hyena_filter_bwd_ld(p, dk, p != nullptr ? p->L : 0, saved, g, workspace, workspace_bytes, stream);
}

int hyena_filter_fwd_ld(const hyena_filter_params* p, float* k, int ldk, float* saved, void* stream) {
    if (!flt_params_ok(p) || k == nullptr || ldk < p->L) return HYENA_ERR_BAD_ARG;
    FilterArgs a;
    a.z = p->z; a.t = p->t; a.w0 = p->w0; a.b0 = p->b0; a.w1 = p->w1; a.b1 = p->b1; a.w2 = p->w2; a.b2 = p->b2; a.w3 = p->w3;
    a.freq = p->freq; a.deltas = p->deltas; a.k = k; a.acts = saved; a.shift = p->shift; a.modulate = p->modulate;
    a.L = p->L; a.E = p->E; a.zs = p->z_stride; a.D = p->D; a.ldk = ldk; a.lds = hyena_filter_row_pitch(p->L);
    switch (p->D) {
        case 64: launch_filter_fwd<64>(a, stream); break;
        case 128: launch_filter_fwd<128>(a, stream); break;
        default: launch_filter_fwd<256>(a, stream); break;
    }
    return hy_launch_error() ? HYENA_ERR_LAUNCH : HYENA_OK;
}

int hyena_filter_bwd_ld(const hyena_filter_params* p, const float* dk, int ldk, const float* saved, const hyena_filter_grads* g,
                        void* workspace, size_t workspace_bytes, void* stream) {
    if (!flt_params_ok(p) || dk == nullptr || saved == nullptr || g == nullptr || workspace == nullptr || ldk < p->L) return HYENA_ERR_BAD_ARG;
    if (!g->dw0 || !g->db0 || !g->dw1 || !g->db1 || !g->dw2 || !g->db2 || !g->dw3 || !g->dfreq) return HYENA_ERR_BAD_ARG;
    if (workspace_bytes < hyena_filter_workspace_bytes(p->L, p->D)) return HYENA_ERR_WORKSPACE;
    const int L = p->L, P = hyena_filter_row_pitch(L);
    float* dA = static_cast<float*>(workspace);
    float* dB = dA + (size_t)FLT_O * P;
    float* part = dB + (size_t)FLT_O * P;
    const float* a0 = saved;
    const float* a1 = saved + (size_t)FLT_O * P;
    const float* a2 = saved + (size_t)2 * FLT_O * P;

    FilterBwdArgs a;
    a.freq = p->freq; a.t = p->t; a.deltas = p->deltas; a.shift = p->shift; a.modulate = p->modulate; a.L = L; a.zs = p->z_stride; a.rdt = 0;
    // last layer: delta_out = dk * modulation;  dW3, and delta_2 -> dA
    a.dout = dk; a.w = p->w3; a.aprev = a2; a.dprev = dA; a.ni = FLT_O; a.ldo = ldk; a.lda = P; a.ldp = P;
    switch (p->D) {
        case 64: launch_filter_layer_bwd<64, FLT_O, FLT_ACT | FLT_MOD>(a, part, g->dw3, nullptr, g->dfreq, true, stream); break;
        case 128: launch_filter_layer_bwd<128, FLT_O, FLT_ACT | FLT_MOD>(a, part, g->dw3, nullptr, g->dfreq, true, stream); break;
        default: launch_filter_layer_bwd<256, FLT_O, FLT_ACT | FLT_MOD>(a, part, g->dw3, nullptr, g->dfreq, true, stream); break;
    }
    a.modulate = 0; a.ldo = P;
    a.dout = dA; a.w = p->w2; a.aprev = a1; a.dprev = dB;
    launch_filter_layer_bwd<FLT_O, FLT_O, FLT_ACT>(a, part, g->dw2, g->db2, g->dfreq, false, stream);
    a.dout = dB; a.w = p->w1; a.aprev = a0; a.dprev = dA;
    launch_filter_layer_bwd<FLT_O, FLT_O, FLT_ACT>(a, part, g->dw1, g->db1, g->dfreq, false, stream);
    a.dout = dA; a.w = p->w0; a.aprev = p->z; a.dprev = g->dz; a.ni = p->E; a.ldp = L;        // dz (E, L) is the caller's: packed
    launch_filter_layer_bwd<FLT_O, FLT_E, 0>(a, part, g->dw0, g->db0, g->dfreq, false, stream);
    return hy_launch_error() ? HYENA_ERR_LAUNCH : HYENA_OK;
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------------------------
// fused residual add + LayerNorm of a block (include/hyena_block.h)
// ---------------------------------------------------------------------------------------------------------------
namespace {
int blk_grid(long rows) {
    const long g = (rows + BLK_WAVES - 1) / BLK_WAVES;
    return (int)(g < BLK_MAX_GRID ? g : BLK_MAX_GRID);
}
bool blk_dtype_ok(int d) { return d == HYENA_F32 || d == HYENA_BF16 || d == HYENA_F16; }

template <int XDT, int ODT>
int blk_launch_e(bool fwd, const AddNormArgs& a, void* stream) {
    const int grid = blk_grid(a.rows);
#define HY_BLK_CASE(e)                                                                                                    \
    case e:                                                                                                               \
        if (fwd) HY_LAUNCH((add_norm_fwd_kernel<XDT, ODT, e>), dim3(grid), dim3(BLK_THREADS), 0, stream, a);              \
        else HY_LAUNCH((add_norm_bwd_kernel<XDT, ODT, e>), dim3(grid), dim3(BLK_THREADS),                                 \
                       (size_t)BLK_WAVES * 3 * 64 * e * sizeof(float), stream, a);                                        \
        break;
    switch (a.D / 64) {
        HY_BLK_CASE(1)
        HY_BLK_CASE(2)
        HY_BLK_CASE(4)
        HY_BLK_CASE(8)
        HY_BLK_CASE(16)
        default: return HYENA_ERR_BAD_ARG;
    }
#undef HY_BLK_CASE
    return hy_launch_error() ? HYENA_ERR_LAUNCH : HYENA_OK;
}

template <int XDT>
int blk_launch_o(bool fwd, int odt, const AddNormArgs& a, void* stream) {
    switch (odt) {
        case HYENA_F32: return blk_launch_e<XDT, DT_F32>(fwd, a, stream);
        case HYENA_BF16: return blk_launch_e<XDT, DT_BF16>(fwd, a, stream);
        default: return blk_launch_e<XDT, DT_F16>(fwd, a, stream);
    }
}
int blk_launch(bool fwd, int xdt, int odt, const AddNormArgs& a, void* stream) {
    switch (xdt) {
        case HYENA_F32: return blk_launch_o<DT_F32>(fwd, odt, a, stream);
        case HYENA_BF16: return blk_launch_o<DT_BF16>(fwd, odt, a, stream);
        default: return blk_launch_o<DT_F16>(fwd, odt, a, stream);
    }
}
}  // namespace

namespace {
// the embedding-fused forms: x0 = table[ids] (fp32 table, d_model 64 / 128 / 256), see block_kernels.h
template <int ODT>
int blk_launch_emb_fwd(const AddNormArgs& a, void* stream) {
    const int grid = blk_grid(a.rows);
    switch (a.D / 64) {
        case 1: HY_LAUNCH((add_norm_fwd_kernel<DT_F32, ODT, 1, true>), dim3(grid), dim3(BLK_THREADS), 0, stream, a); break;
        case 2: HY_LAUNCH((add_norm_fwd_kernel<DT_F32, ODT, 2, true>), dim3(grid), dim3(BLK_THREADS), 0, stream, a); break;
        case 4: HY_LAUNCH((add_norm_fwd_kernel<DT_F32, ODT, 4, true>), dim3(grid), dim3(BLK_THREADS), 0, stream, a); break;
        default: return HYENA_ERR_BAD_ARG;
    }
    return hy_launch_error() ? HYENA_ERR_LAUNCH : HYENA_OK;
}
template <int GDT>
int blk_launch_emb_bwd(const AddNormArgs& a, void* stream) {
    const int grid = blk_grid(a.rows);
    const size_t lds = (size_t)BLK_VMAX * a.D * sizeof(float);              // >= the [BLK_WAVES][2][D] of the weight / bias partials
    switch (a.D / 64) {
        case 1: HY_LAUNCH((add_norm_bwd_kernel<GDT, DT_F32, 1, true>), dim3(grid), dim3(BLK_THREADS), lds, stream, a); break;
        case 2: HY_LAUNCH((add_norm_bwd_kernel<GDT, DT_F32, 2, true>), dim3(grid), dim3(BLK_THREADS), lds, stream, a); break;
        case 4: HY_LAUNCH((add_norm_bwd_kernel<GDT, DT_F32, 4, true>), dim3(grid), dim3(BLK_THREADS), lds, stream, a); break;
        default: return HYENA_ERR_BAD_ARG;
    }
    return hy_launch_error() ? HYENA_ERR_LAUNCH : HYENA_OK;
}
}  // namespace

extern "C" {

int hyena_add_norm_supported(int D, int x_dtype, int out_dtype) {
    const int e = D / 64;
    return D >= 64 && D % 64 == 0 && (e == 1 || e == 2 || e == 4 || e == 8 || e == 16) && blk_dtype_ok(x_dtype) &&
           blk_dtype_ok(out_dtype);
}

namespace {
// dropout parameters of a launch; false if (p, seed) are inconsistent
bool blk_set_dropout(AddNormArgs& a, float p, const unsigned long long* seed) {
    a.seed = nullptr; a.drop_below = 0u; a.keep_scale = 1.f;
    if (p == 0.f) return true;
    if (!(p > 0.f) || !(p < 1.f) || seed == nullptr) return false;
    a.seed = seed;
    a.drop_below = (unsigned)((double)p * 4294967296.0);
    a.keep_scale = (float)(1.0 / (1.0 - (double)p));
    return true;
}
}  // namespace

int hyena_dropout_add_norm_fwd(const void* x0, int x_dtype, const float* residual_in, const float* weight, const float* bias, float eps,
                               float dropout_p, const unsigned long long* seed, void* out, int out_dtype, float* residual_out,
                               float* mean, float* rstd, long rows, int D, void* stream) {
    if (x0 == nullptr || weight == nullptr || bias == nullptr || out == nullptr || residual_out == nullptr || mean == nullptr ||
        rstd == nullptr || rows < 1 || !hyena_add_norm_supported(D, x_dtype, out_dtype))
        return HYENA_ERR_BAD_ARG;
    AddNormArgs a;
    a.x = x0; a.res_in = residual_in; a.weight = weight; a.bias = bias; a.out = out; a.res_out = residual_out; a.saved = nullptr;
    a.mean = mean; a.rstd = rstd; a.part = nullptr; a.rows = rows; a.D = D; a.eps = eps; a.ids = nullptr; a.part_e = nullptr; a.V = 0;
    if (!blk_set_dropout(a, dropout_p, seed)) return HYENA_ERR_BAD_ARG;
    return blk_launch(true, x_dtype, out_dtype, a, stream);
}

int hyena_add_norm_fwd(const void* x0, int x_dtype, const float* residual_in, const float* weight, const float* bias, float eps,
                       void* out, int out_dtype, float* residual_out, float* mean, float* rstd, long rows, int D, void* stream) {
    return hyena_dropout_add_norm_fwd(x0, x_dtype, residual_in, weight, bias, eps, 0.f, nullptr, out, out_dtype, residual_out, mean, rstd,
                                      rows, D, stream);
}

int hyena_embed_add_norm_supported(int V, int D, int out_dtype) {
    return V >= 1 && V <= BLK_VMAX && (D == 64 || D == 128 || D == 256) && blk_dtype_ok(out_dtype);
}

int hyena_embed_add_norm_fwd(const long long* ids, const float* table, int V, const float* weight, const float* bias, float eps,
                             float dropout_p, const unsigned long long* seed, void* out, int out_dtype, float* residual_out,
                             float* mean, float* rstd, long rows, int D, void* stream) {
    if (ids == nullptr || table == nullptr || weight == nullptr || bias == nullptr || out == nullptr || residual_out == nullptr ||
        mean == nullptr || rstd == nullptr || rows < 1 || !hyena_embed_add_norm_supported(V, D, out_dtype))
        return HYENA_ERR_BAD_ARG;
    AddNormArgs a;
    a.x = table; a.res_in = nullptr; a.weight = weight; a.bias = bias; a.out = out; a.res_out = residual_out; a.saved = nullptr;
    a.mean = mean; a.rstd = rstd; a.part = nullptr; a.rows = rows; a.D = D; a.eps = eps; a.ids = ids; a.part_e = nullptr; a.V = V;
    if (!blk_set_dropout(a, dropout_p, seed)) return HYENA_ERR_BAD_ARG;
    switch (out_dtype) {
        case HYENA_F32: return blk_launch_emb_fwd<DT_F32>(a, stream);
        case HYENA_BF16: return blk_launch_emb_fwd<DT_BF16>(a, stream);
        default: return blk_launch_emb_fwd<DT_F16>(a, stream);
    }
}

size_t hyena_embed_add_norm_partial_floats(long rows, int D) {
    if (rows < 1 || D < 1) return 0;
    return (size_t)blk_grid(rows) * (2 + BLK_VMAX) * D;
}

int hyena_embed_add_norm_bwd(const void* dout, int dout_dtype, const float* d_residual_out, const float* residual_out, const long long* ids,
                             const float* weight, const float* mean, const float* rstd, float dropout_p,
                             const unsigned long long* seed, float* d_table, int V, float* dweight, float* dbias, float* partial,
                             long rows, int D, void* stream) {
    if (dout == nullptr || residual_out == nullptr || ids == nullptr || weight == nullptr || mean == nullptr || rstd == nullptr ||
        d_table == nullptr || dweight == nullptr || dbias == nullptr || partial == nullptr || rows < 1 ||
        !hyena_embed_add_norm_supported(V, D, dout_dtype))
        return HYENA_ERR_BAD_ARG;
    const int grid = blk_grid(rows);
    AddNormArgs a;
    a.x = dout; a.res_in = d_residual_out; a.weight = weight; a.bias = nullptr; a.out = nullptr; a.res_out = nullptr;
    a.saved = residual_out; a.mean = const_cast<float*>(mean); a.rstd = const_cast<float*>(rstd); a.part = partial;
    a.rows = rows; a.D = D; a.eps = 0.f; a.ids = ids; a.part_e = partial + (size_t)grid * 2 * D; a.V = V;
    if (!blk_set_dropout(a, dropout_p, seed)) return HYENA_ERR_BAD_ARG;
    int st;
    switch (dout_dtype) {
        case HYENA_F32: st = blk_launch_emb_bwd<DT_F32>(a, stream); break;
        case HYENA_BF16: st = blk_launch_emb_bwd<DT_BF16>(a, stream); break;
        default: st = blk_launch_emb_bwd<DT_F16>(a, stream); break;
    }
    if (st) return st;
    RedBatch red;                                   // LayerNorm weight / bias gradients and the V x D embedding gradient, fixed order
    red.add(partial, dweight, grid, D, 2 * D, 0);
    red.add(partial + D, dbias, grid, D, 2 * D, 0);
    red.add(a.part_e, d_table, grid, V * D, BLK_VMAX * D, 0);
    HY_LAUNCH(filter_reduce_multi_kernel, dim3(red.blocks()), dim3(256), FLT_RED_SMEM, stream, red.jobs);
    return hy_launch_error() ? HYENA_ERR_LAUNCH : HYENA_OK;
}

size_t hyena_add_norm_partial_floats(long rows, int D) {
    if (rows < 1 || D < 1) return 0;
    return (size_t)blk_grid(rows) * 3 * D;          // (dweight | dbias | column sums of dx0: the third plane is used by the _colsum entry point only)
}

int hyena_add_norm_bwd(const void* dout, int dout_dtype, const float* d_residual_out, const float* residual_out,
                       const float* weight, const float* mean, const float* rstd, void* dx0, int dx_dtype,
                       float* d_residual_in, float* dweight, float* dbias, float* partial, long rows, int D, void* stream) {
    return hyena_dropout_add_norm_bwd(dout, dout_dtype, d_residual_out, residual_out, weight, mean, rstd, 0.f, nullptr, dx0, dx_dtype,
                                      d_residual_in, dweight, dbias, partial, rows, D, stream);
}

int hyena_dropout_add_norm_bwd(const void* dout, int dout_dtype, const float* d_residual_out, const float* residual_out,
                               const float* weight, const float* mean, const float* rstd, float dropout_p,
                               const unsigned long long* seed, void* dx0, int dx_dtype, float* d_residual_in, float* dweight,
                               float* dbias, float* partial, long rows, int D, void* stream) {
    return hyena_dropout_add_norm_bwd_colsum(dout, dout_dtype, d_residual_out, residual_out, weight, mean, rstd, dropout_p, seed, dx0, dx_dtype,
                                             d_residual_in, dweight, dbias, nullptr, partial, rows, D, stream);
}

int hyena_dropout_add_norm_bwd_colsum(const void* dout, int dout_dtype, const float* d_residual_out, const float* residual_out,
                                      const float* weight, const float* mean, const float* rstd, float dropout_p,
                                      const unsigned long long* seed, void* dx0, int dx_dtype, float* d_residual_in, float* dweight,
                                      float* dbias, float* dx0_colsum, float* partial, long rows, int D, void* stream) {
    if (dout == nullptr || residual_out == nullptr || weight == nullptr || mean == nullptr || rstd == nullptr || dx0 == nullptr ||
        dweight == nullptr || dbias == nullptr || partial == nullptr || rows < 1 || !hyena_add_norm_supported(D, dout_dtype, dx_dtype))
        return HYENA_ERR_BAD_ARG;
    AddNormArgs a;
    a.x = dout; a.res_in = d_residual_out; a.weight = weight; a.bias = nullptr; a.out = dx0; a.res_out = d_residual_in;
    a.saved = residual_out; a.mean = const_cast<float*>(mean); a.rstd = const_cast<float*>(rstd); a.part = partial;
    a.np = dx0_colsum != nullptr ? 3 : 2;
    a.rows = rows; a.D = D; a.eps = 0.f; a.ids = nullptr; a.part_e = nullptr; a.V = 0;
    if (!blk_set_dropout(a, dropout_p, seed)) return HYENA_ERR_BAD_ARG;
    const int st = blk_launch(false, dout_dtype, dx_dtype, a, stream);
    if (st) return st;
    const int grid = blk_grid(rows);
    // partial rows are [dweight (D) | dbias (D) (| column sums of dx0 (D))]: reductions with a row stride of np D
    RedBatch red;
    red.add(partial, dweight, grid, D, a.np * D, 0);
    red.add(partial + D, dbias, grid, D, a.np * D, 0);
    if (dx0_colsum != nullptr) red.add(partial + 2 * D, dx0_colsum, grid, D, a.np * D, 0);
    HY_LAUNCH(filter_reduce_multi_kernel, dim3(red.blocks()), dim3(256), FLT_RED_SMEM, stream, red.jobs);
    return hy_launch_error() ? HYENA_ERR_LAUNCH : HYENA_OK;
}

}  // extern "C"
