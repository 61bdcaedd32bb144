// proj2_kernels.h -- round 6: the second generation of the output projection's matrix-core kernel (hyena.py:432-440: y = out_proj((y * x[0])^T)).
//
// What round 5 measured about generation 1 (proj_kernels.h::outproj_gate_fwd_kernel): 585 - 685 us at L = 2^20, d = 256 for 2.0 GB = 3.0 - 3.4 TB/s,
// against 4.9 - 5.5 TB/s of the streaming passes -- "weights register-stationary -> 246 - 256 VGPRs -> 2 wavefronts per SIMD".  Read tile by tile it is
// worse than an occupancy problem: with 128 of a wavefront's 256 registers holding weights the operand rows of a 64-position tile are fetched in FOUR
// batches (all eight rounds in flight spilled 61 - 126 registers), each batch a full round trip to memory with nothing else of that workgroup to run
// under it -- 18 us per tile and workgroup where the tile's own arithmetic is ~3 us.  Generation 2 changes the shape, not the arithmetic:
//
//   * 16 output channels per wavefront instead of 64: the stationary weights are K / 32 fragments of v_mfma_f32_16x16x32 = 32 VGPRs at K = 256 (was
//     128), a workgroup is K / 16 wavefronts (1024 threads at K = 256: one workgroup per CU, four wavefronts per SIMD, <= 128 VGPRs each);
//   * a tile's operand rows (y and the x0 rows of xT: two rounds of 8 rows x 128 bytes per wavefront) are PREFETCHED into registers one whole tile ahead --
//     requested as soon as the previous tile's copies have been consumed, waited for one tile later -- there is room now;
//   * the product is taken with the weights as the A operand: D[channel][position], a lane holds four neighbouring channels of ONE position = 8 bytes
//     of a position-major row, parked with one ds_write_b64 per 16 x 16 block (generation 1: sixteen 2-byte writes per 32 x 32 block) into ONE
//     [64 positions][K] tile of the workgroup; after a barrier a wavefront owns 64 / WAVES WHOLE rows of it: the plain epilogue stores them as 512-byte
//     contiguous pieces, the LN = true epilogue runs the block's residual add + LayerNorm on them -- one code path, the layout and, operation for
//     operation, the arithmetic of block_kernels.h::add_norm_fwd_kernel.
//
// The z tile, its swizzle, the gate arithmetic (cm_post_fwd's, FMA by FMA: zT is its bits) and the pulled-back last tile of a sequence are generation 1's.
// Compiled by hipcc for gfx950 (product) and, with -DHIPEMU, by g++ against tests/hipemu (tests only).
#pragma once
#include "proj_kernels.h"

namespace hyena {
namespace pj {

template <int K> struct Op2Cfg {
    static_assert(K == 128 || K == 256, "d_model of the HyenaDNA models");
    static constexpr int WAVES = K / 16;                       // one 16-channel MFMA row block per wavefront (N = K)
    static constexpr int THREADS = WAVES * 64;
    static constexpr int KS = K / 32;                          // v_mfma_f32_16x16x32 steps over the contraction
    static constexpr int RND = K / (WAVES * 8);                // staging rounds per wavefront and tile: 8 channel rows each (= 2)
    static constexpr int ZBUF = PJ_NT * K * 2;                 // the z tile (generation 1's layout)
    static constexpr int TAPB = K * 8 * 4;                     // per channel 8 floats: w0 w1 w2 b_sc b_in ln_w ln_b -
    static constexpr int OROW = (K + 8) * 2;                   // a row of the output tile [position][channel], +16 bytes: b64 writes and 16-byte reads conflict-free
    static constexpr int OBUF = PJ_NT * OROW;
    static constexpr int ROWS = PJ_NT / WAVES;                 // whole output rows a wavefront owns in the row phase (4 / 8)
    static constexpr size_t LDS = (size_t)ZBUF + TAPB + OBUF;  // 73 KB at K = 256 (one workgroup per CU), 37 KB at K = 128 (two)
    // LN: + the tile's fp32 residual rows, brought in by LDS-direct loads a whole tile phase ahead (a wavefront fetches the rows it will normalise
    // into its own slice): 64 KB at K = 256 (137 KB in all), 32 KB at K = 128 (69 KB: still two workgroups per CU)
    static constexpr int RROW = K * 4;
    static constexpr int RBUF = PJ_NT * RROW;
    static constexpr size_t LDS_LN = LDS + RBUF;
};

template <int K, int DT, bool LN>
__global__ void __launch_bounds__(Op2Cfg<K>::THREADS, 4) outproj_gate_fwd2_kernel(OutProjArgs a) {
    typedef Op2Cfg<K> C;
    typedef typename Elem<DT>::type elem_t;
    static_assert(sizeof(elem_t) == 2, "16-bit element types only");
    HY_SMEM(smem);
    PJ_VMQ_DECL;
    const int tid = (int)threadIdx.x, wave = HY_SGPR(tid >> 6), lane = tid & 63, j = lane & 15, kq = lane >> 4;
    const int r = lane >> 3, c = lane & 7;                    // staging role: row of the round, 16-byte piece (8 positions)
    const int run = blockIdx.x;
    const int t_begin = run * a.tiles_per_wg;
    if (t_begin >= a.tiles) return;
    const int t_end = (t_begin + a.tiles_per_wg < a.tiles) ? t_begin + a.tiles_per_wg : a.tiles;
    const int n0 = wave * 16;                                 // first output channel of this wavefront
    constexpr int N = K;

    HY_LDS char* const zt = HY_LDS_CAST(char, smem);
    HY_LDS float* const taps = reinterpret_cast<HY_LDS float*>(HY_LDS_CAST(char, smem) + C::ZBUF);
    HY_LDS char* const ot = HY_LDS_CAST(char, smem) + C::ZBUF + C::TAPB;
    HY_LDS char* const rt = HY_LDS_CAST(char, smem) + C::LDS + wave * (C::ROWS * C::RROW);       // LN: this wavefront's residual rows

    for (int k = tid; k < K; k += C::THREADS) {
        taps[k * 8 + 0] = a.w[k * 3]; taps[k * 8 + 1] = a.w[k * 3 + 1]; taps[k * 8 + 2] = a.w[k * 3 + 2];
        taps[k * 8 + 3] = a.b[k];
        taps[k * 8 + 4] = a.bin != nullptr ? a.bin[k] : 0.f;
        if constexpr (LN) { taps[k * 8 + 5] = a.ln_w[k]; taps[k * 8 + 6] = a.ln_b[k]; }
    }
    // stationary operand: 16 weight rows as A fragments (row = output channel n0 + j, k = 32 ks + 8 kq ...)
    Frag wf[C::KS];
    {
        const char* row = reinterpret_cast<const char*>(a.W) + ((size_t)(n0 + j) * K + 8 * kq) * 2;
        HY_UNROLL
        for (int ks = 0; ks < C::KS; ++ks) wf[ks] = ld16(row + ks * 64);
    }
    float bias[4];                                            // D rows of this lane: channels n0 + 4 kq + i
    HY_UNROLL
    for (int i = 0; i < 4; ++i) bias[i] = a.bias != nullptr ? a.bias[n0 + 4 * kq + i] : 0.f;

    const elem_t* const yb = reinterpret_cast<const elem_t*>(a.y);
    const elem_t* const xb = reinterpret_cast<const elem_t*>(a.xT);
    elem_t* const zb = reinterpret_cast<elem_t*>(a.zT);
    elem_t* const ob = reinterpret_cast<elem_t*>(a.out);

    // a sequence's last tile is pulled back to end at L when L is not a multiple of 64 (generation 1): every tile is whole
    auto tile_origin = [&](int t, int& b, int& l0) {
        b = t / a.tiles_per_seq;
        const int l0raw = (t - b * a.tiles_per_seq) * PJ_NT;
        l0 = l0raw + PJ_NT <= a.L ? l0raw : a.L - PJ_NT;
    };
    // the operand rows of a tile, one tile ahead: round ii of this wavefront = channels 8 (ii WAVES + wave) + r, this lane the 8 positions l0 + 8 c ...
    Frag yr[C::RND], xr[C::RND];
    uint32_t halo[C::RND];
    auto fetch = [&](int t) {
        int b, l0;
        tile_origin(t, b, l0);
        const int lp = l0 + 8 * c;
        HY_UNROLL
        for (int ii = 0; ii < C::RND; ++ii) {
            const int k = 8 * (ii * C::WAVES + wave) + r;
            const elem_t* yrow = yb + ((size_t)b * a.D + k) * a.lda;
            const elem_t* xrow = xb + ((size_t)k * (size_t)a.csx + (size_t)b * a.bsx);
            yr[ii] = ld16(yrow + lp);                         // (gfx950 global memory takes under-aligned 16-byte accesses: rows may start at any even byte)
            xr[ii] = ld16(xrow + lp);
            uint32_t h2 = 0u;
            if (c == 0 && l0 >= 2) __builtin_memcpy(&h2, xrow + (l0 - 2), 4);
            else if (c == 0 && l0 == 1) { uint16_t h1; __builtin_memcpy(&h1, xrow, 2); h2 = (uint32_t)h1 << 16; }   // L = 65: the pulled-back tile starts at 1
            halo[ii] = h2;
        }
    };
    fetch(t_begin);
    PJ_BARRIER();                                             // the taps are in LDS

    for (int t = t_begin; t < t_end; ++t) {
        int b, l0;
        tile_origin(t, b, l0);
        const int lp = l0 + 8 * c;
        if constexpr (LN) {
            // this wavefront's ROWS residual rows of the tile (ROWS K 4 bytes, contiguous in memory) -> its LDS slice, 1 KB per instruction; they land
            // under phase A, the product and two barriers (in registers they spilled, and half of them had to be requested behind the park: exposed)
            if (a.res_in != nullptr) {
                const char* rb = reinterpret_cast<const char*>(a.res_in) + ((size_t)b * a.L + l0 + wave * C::ROWS) * (size_t)C::RROW;
                HY_UNROLL
                for (int i = 0; i < C::ROWS * C::RROW / 1024; ++i)
                    glds16(HY_UNIFORM_PTR(const char, rb), (uint32_t)(i * 1024 + lane * 16), rt + i * 1024, lane PJ_VMQ_ARG);
            }
        }
        // ---- phase A: z = round(y * shortconv(x0)) -> zT (global), pairs of channels -> the swizzled z tile ----
        HY_UNROLL
        for (int ii = 0; ii < C::RND; ++ii) {
            const int k = 8 * (ii * C::WAVES + wave) + r;
            const uint32_t prev = HY_SHFL_U32(xr[ii].w[3], lane - 1);        // positions lp - 2, lp - 1 (lane - 1 = same row, piece c - 1)
            const uint32_t hw = c == 0 ? halo[ii] : prev;
            elem_t xe[10], ye[8];
            __builtin_memcpy(xe, &hw, 4);
            __builtin_memcpy(xe + 2, xr[ii].w, 16);
            __builtin_memcpy(ye, yr[ii].w, 16);
            const float w0 = taps[k * 8], w1 = taps[k * 8 + 1], w2 = taps[k * 8 + 2], bsc = taps[k * 8 + 3], bin = taps[k * 8 + 4];
            elem_t ze[8];
            HY_UNROLL
            for (int e = 0; e < 8; ++e) {
                const int l = lp + e;
                const float x0 = l >= 2 ? Elem<DT>::dec(xe[e]) + bin : 0.f, x1 = l >= 1 ? Elem<DT>::dec(xe[e + 1]) + bin : 0.f,
                            x2 = Elem<DT>::dec(xe[e + 2]) + bin;
                const float c0 = __builtin_fmaf(w2, x2, __builtin_fmaf(w1, x1, __builtin_fmaf(w0, x0, bsc)));   // = cm_sc
                ze[e] = Elem<DT>::cvt(Elem<DT>::dec(ye[e]) * c0);                                                  // = cm_post_fwd
            }
            Frag zp;
            __builtin_memcpy(zp.w, ze, 16);
            if (zb != nullptr) st16(zb + ((size_t)k * (size_t)a.csz + (size_t)b * a.bsz) + lp, zp);
            // lanes r (even) and r + 1 (= lane ^ 8) hold channels k, k + 1 for the same 8 positions: the even one takes positions 0..3 of both,
            // the odd one positions 4..7, as 4-byte (k, k + 1) pairs
            const bool odd = (r & 1) != 0;
            const uint32_t s0 = odd ? zp.w[0] : zp.w[2], s1 = odd ? zp.w[1] : zp.w[3];
            const uint32_t g0 = HY_SHFL_U32(s0, lane ^ 8), g1 = HY_SHFL_U32(s1, lane ^ 8);
            const uint32_t m0 = odd ? zp.w[2] : zp.w[0], m1 = odd ? zp.w[3] : zp.w[1];
            const uint32_t lo0 = odd ? g0 : m0, lo1 = odd ? g1 : m1;
            const uint32_t hi0 = odd ? m0 : g0, hi1 = odd ? m1 : g1;
            const uint32_t pr[4] = {(lo0 & 0xffffu) | (hi0 << 16), (lo0 >> 16) | (hi0 & 0xffff0000u),
                                    (lo1 & 0xffffu) | (hi1 << 16), (lo1 >> 16) | (hi1 & 0xffff0000u)};
            const int kk = k & ~1, pp = 8 * c + (odd ? 4 : 0);
            HY_LDS char* const base = zt + (kk >> 3) * 1024 + (kk & 7) * 2;
            HY_UNROLL
            for (int e = 0; e < 4; ++e) *reinterpret_cast<HY_LDS uint32_t*>(base + op_slot(pp + e) * 16) = pr[e];
        }
        // the next tile's rows: in flight under this tile's product and row phase
        if (t + 1 < t_end) fetch(t + 1);
        constexpr int E = K / 64;
        PJ_BARRIER();                                         // the z tile is complete
        // ---- phase B: out^T tile = W z^T on the matrix cores: D[channel 4 kq + i][position 16 pt + j] ----
        acc4_t acc[PJ_NT / 16];
        HY_UNROLL
        for (int pt = 0; pt < PJ_NT / 16; ++pt) {
            HY_UNROLL
            for (int i = 0; i < 4; ++i) acc[pt][i] = 0.f;
        }
        HY_UNROLL
        for (int ks = 0; ks < C::KS; ++ks) {
            HY_UNROLL
            for (int pt = 0; pt < PJ_NT / 16; ++pt) {
                const Frag bz = lds_ld16(zt + (4 * ks + kq) * 1024 + op_slot(pt * 16 + j) * 16);      // channels 32 ks + 8 kq ..., position 16 pt + j
                acc[pt] = mfma16<DT>(wf[ks], bz, acc[pt]);
            }
        }
        // park: four neighbouring channels of one position = 8 bytes of the output tile's row
        HY_UNROLL
        for (int pt = 0; pt < PJ_NT / 16; ++pt) {
            elem_t o4[4];
            HY_UNROLL
            for (int i = 0; i < 4; ++i) o4[i] = Elem<DT>::cvt(acc[pt][i] + bias[i]);
            uint32_t p2[2];
            __builtin_memcpy(p2, o4, 8);
            HY_LDS uint32_t* dst = reinterpret_cast<HY_LDS uint32_t*>(ot + (pt * 16 + j) * C::OROW + (n0 + 4 * kq) * 2);
            dst[0] = p2[0];
            dst[1] = p2[1];
        }
        PJ_BARRIER();                                         // the output tile is complete; everybody is done with the z tile
        // ---- row phase: this wavefront's ROWS whole rows of the tile ----
        if constexpr (LN) {
            // the block's residual add + LayerNorm (simple_lm.py:280-284): lane l holds channels E l .. E l + E - 1 of a row -- add_norm_fwd_kernel's layout
            // and arithmetic; residual', out, mean and rstd are its bits
            const int c0 = lane * E;
            const float inv_d = 1.f / (float)K;
            PJ_VMWAIT(0);                                     // my residual rows have landed (requested a whole tile phase ago; with them the next tile's operand rows)
            HY_UNROLL
            for (int i = 0; i < C::ROWS; ++i) {
                const int pos = wave * C::ROWS + i;
                const size_t row = (size_t)b * a.L + l0 + pos, off = row * N + c0;
                elem_t xe[E];
                __builtin_memcpy(xe, ot + pos * C::OROW + c0 * 2, sizeof(xe));
                float v[E];
                HY_UNROLL
                for (int e = 0; e < E; ++e) v[e] = Elem<DT>::dec(xe[e]);
                if (a.res_in != nullptr) {
                    float q[E];
                    __builtin_memcpy(q, rt + i * C::RROW + c0 * 4, sizeof(q));
                    HY_UNROLL
                    for (int e = 0; e < E; ++e) v[e] += q[e];
                }
                float sm = 0.f;
                HY_UNROLL
                for (int e = 0; e < E; ++e) sm += v[e];
                const float mean = wave_sum(sm) * inv_d;
                float vs = 0.f;
                HY_UNROLL
                for (int e = 0; e < E; ++e) vs += (v[e] - mean) * (v[e] - mean);
                const float rstd = 1.f / sqrtf(wave_sum(vs) * inv_d + a.eps);
                float o[E];
                HY_UNROLL
                for (int e = 0; e < E; ++e) o[e] = (v[e] - mean) * rstd * taps[(c0 + e) * 8 + 5] + taps[(c0 + e) * 8 + 6];
                blk_store<DT, E>(a.out, off, o);
                blk_store<DT_F32, E>(a.res_out, off, v);
                if (lane == 0) { a.mean[row] = mean; a.rstd[row] = rstd; }
            }
        } else {
            constexpr int LPR = K * 2 / 16, RPI = 64 / LPR;   // lanes per row (32 / 16), rows per instruction (2 / 4)
            HY_UNROLL
            for (int m = 0; m < C::ROWS / RPI; ++m) {
                const int pos = wave * C::ROWS + m * RPI + lane / LPR, pc = lane % LPR;
                st16(ob + (((size_t)b * a.L + l0 + pos) * N + 8 * pc), lds_ld16(ot + pos * C::OROW + pc * 16));
            }
        }
        // (no barrier here: the next tile's phase A writes the z tile, which nobody reads any more; its park follows the next barrier)
    }
}

}  // namespace pj
}  // namespace hyena

namespace hyena {
namespace pj {

// =============================================================================================================================
// The input projection, generation 2 (round 6).  Same contract as proj_kernels.h::inproj_pre_fwd_kernel -- xT (3D, B, Lx) = W u^T without the bias,
// vg (B, D, Lc) = shortconv(xT + b_in)[v] * shortconv(xT + b_in)[x1], vg the bits cm_pre_fwd makes of the stored xT -- on another shape.
//
// What was measured first (profiles/r6b_rowpiece_probe.txt, r6e_proj_nost_nold_ab.txt; L = 2^20, d = 256): the kernel's memory traffic replayed by a
// kernel that does nothing else takes 0.52 - 0.54 ms; generation 1 with neither global loads nor stores runs 0.45 ms; together 0.72 - 0.77 ms: the
// instruction stream and the memory stream are each as long as the other and overlap badly.  Three steps, each measured:
//   (a) 32 channels of ONE group per wavefront on v_mfma_f32_32x32x16 with the operand tile as A -- D[position][channel], the tile's rows dealt to the MFMA
//       rows so that a lane holds SIXTEEN CONSECUTIVE positions of one channel: two ds_write_b128 per 32 x 32 block where generation 1 issues sixteen
//       2-byte writes per 16 x 16 block; 12 wavefronts = 128 channels x 3 groups per workgroup, 64 weight registers (was 96), three wavefronts per SIMD;
//       the [384 rows][64 positions] tile belongs to the workgroup: every wavefront stores its own 32 rows of xT and the sixteen 8-channel units of vg
//       are dealt to all twelve (the x0 wavefronts, which have no window arithmetic of their own, take two each).  ~300 instead of ~550 vector + LDS
//       instructions per tile and wavefront -- and NO faster: 474 us without memory traffic, 800 - 860 us with it.  With ONE workgroup per CU every
//       wavefront is in the same phase: matrix cores idle during the row phase, vector pipes and the memory queue idle during the product.
//   (b) the same in workgroups of 6 wavefronts, two per CU, hoping for generation 1's statistical overlap: 588 / 1000 us.  Dropped.
//   (c) (this kernel) the row phase of tile t - 1 is issued BETWEEN the matrix instructions of tile t, by every wavefront, in program order: the operand
//       tile is double-buffered (tile t + 1 is requested right after the barrier that releases tile t and has a whole tile to land), the parked tile is
//       single (a barrier between the last row-phase read and the park).
// Operand tile by LDS-direct loads with counted waits, the two-position halo carried in the tile's rows, a warm-up tile in front of a run, straight-line
// interior tiles and a predicated path for the others: generation 1's.
// =============================================================================================================================
enum { IP2_WAVES = 12, IP2_THREADS = IP2_WAVES * 64, IP2_CH = 128 /* channels of each group per workgroup */, IP2_CB = 32 /* channels per wavefront */ };
template <int K> struct Ip2Cfg {
    static_assert(K == 128 || K == 256, "d_model of the HyenaDNA models");
    static constexpr int KS = K / 16;                          // v_mfma_f32_32x32x16 steps over the contraction
    static constexpr int PCS = K / 8;                          // 16-byte pieces per row of the u tile
    static constexpr int UROWB = K * 2, UBUF = PJ_NT * UROWB;  // one u tile, unpadded (LDS-direct loads; source pieces permuted: piece c of position p at slot c ^ (p & 15))
    static constexpr int NCHUNK = UBUF / 1024;                 // LDS-direct loads (1 KB each) per tile: 32 / 16, dealt to the wavefronts round-robin
    static constexpr int EROW = PJ_EW * 2;                     // 144 bytes: [8 halo slots (the last two used)][64 positions]
    static constexpr int EBUF = 3 * IP2_CH * EROW;             // the workgroup's [group][channel][PJ_EW] tile: 54 KB
    static constexpr int TAPS = 2 * IP2_CH * 5 * 4;            // (w0, w1, w2, b_sc, b_in) of the x1 and v channels
    static constexpr size_t LDS = 2 * (size_t)UBUF + EBUF + TAPS;  // 123 KB at K = 256, 91 KB at K = 128: one workgroup per CU
};

// operand tile of generation 2: chunk = 1 KB = 64 sixteen-byte slots; slot S = 64 chunk + lane = (position S / PCS, slot S mod PCS) holds the source
// piece (slot ^ (position & 15)): the A-fragment reads of v_mfma_f32_32x32x16 with the row map below touch 16 different bank groups per lane group
template <int K, bool FULL>
__device__ __forceinline__ void issue_operand_tile2(const char* xbase, unsigned p0, unsigned P, HY_LDS char* ubuf, int wave, int lane PJ_VMQ_PARAM) {
    constexpr int PCS = K / 8, NCHUNK = PJ_NT * K * 2 / 1024;
    const char* const tb = HY_UNIFORM_PTR(const char, xbase + (size_t)p0 * K * 2);
    HY_OPAQUE(lane);
    HY_UNROLL
    for (int i = 0; i < (NCHUNK + IP2_WAVES - 1) / IP2_WAVES; ++i) {
        const int chunk = i * IP2_WAVES + wave;
        if (chunk < NCHUNK) {                                  // (wave-uniform)
            const int S = chunk * 64 + lane, pos = S / PCS, c = (S % PCS) ^ (pos & 15);
            if (FULL || p0 + (unsigned)pos < P) glds16(tb, (uint32_t)(pos * K * 2 + c * 16), ubuf + chunk * 1024, lane PJ_VMQ_ARG);
        }
    }
}

template <int K, int DT>
__global__ void __launch_bounds__(IP2_THREADS, 3) inproj_pre_fwd2_kernel(InProjArgs a) {
    typedef Ip2Cfg<K> C;
    typedef typename Elem<DT>::type elem_t;
    static_assert(sizeof(elem_t) == 2, "16-bit element types only");
    HY_SMEM(smem);
    PJ_VMQ_DECL;
    const int tid = (int)threadIdx.x, wave = HY_SGPR(tid >> 6), lane = tid & 63, j = lane & 31, hb = lane >> 5;
    const int D = a.D;
    const unsigned P = (unsigned)a.B * (unsigned)a.Lx;                       // flattened positions (< 2^31, checked by the host)
    // workgroup -> (channel group of 128, run of tiles); the channel groups of one run of positions get slots of ONE XCD (its L2 serves the re-read of u)
    const int ncg = D / IP2_CH;
    int cg, run;
    {
        const int wg = blockIdx.x, xcd = wg & 7, seq = wg >> 3;
        cg = seq % ncg;
        run = (seq / ncg) * 8 + xcd;
    }
    const int t_begin = run * a.tiles_per_wg;
    if (t_begin >= a.tiles) return;
    const int t_end = (t_begin + a.tiles_per_wg < a.tiles) ? t_begin + a.tiles_per_wg : a.tiles;
    const int grp = wave >> 2, cb = wave & 3;                               // this wavefront's group (0 = x0, 1 = x1, 2 = v) and 32-channel block
    const int c0 = grp * D + cg * IP2_CH + cb * IP2_CB;                     // its first row of W / xT
    const int lr0 = wave * IP2_CB;                                          // ... and of the workgroup's tile (rows ordered [group][channel])

    HY_LDS char* const ubuf = HY_LDS_CAST(char, smem);                       // two operand tiles
    HY_LDS char* const et = HY_LDS_CAST(char, smem) + 2 * C::UBUF;
    HY_LDS float* const taps = HY_LDS_CAST(float, smem + 2 * C::UBUF + C::EBUF);

    // stationary operand: 32 weight rows as B fragments (column = channel c0 + j, k = 16 ks + 8 hb ...)
    Frag wf[C::KS];
    {
        const char* row = reinterpret_cast<const char*>(a.W) + ((size_t)(c0 + j) * K + 8 * hb) * 2;
        HY_UNROLL
        for (int ks = 0; ks < C::KS; ++ks) wf[ks] = ld16(row + ks * 32);
    }
    // short-filter taps of the workgroup's x1 and v channels, (w0, w1, w2, b_sc, b_in) per channel
    for (int i = tid; i < 2 * IP2_CH; i += IP2_THREADS) {
        const int c = (1 + i / IP2_CH) * D + cg * IP2_CH + i % IP2_CH;
        HY_LDS float* t = taps + i * 5;
        t[0] = a.w[c * 3]; t[1] = a.w[c * 3 + 1]; t[2] = a.w[c * 3 + 2]; t[3] = a.b[c];
        t[4] = a.bin != nullptr ? a.bin[c] : 0.f;
    }
    const size_t CS = (size_t)a.csx;
    const char* const ubase = reinterpret_cast<const char*>(a.u);
    // row phase roles: every wavefront stores its own 32 rows of xT; vg unit q (8 channels) of the sixteen: x0 wavefront w takes 2 w and 2 w + 1, the others one
    const int nunits = wave < 4 ? 2 : 1, unit0 = wave < 4 ? 2 * wave : 4 + wave;
    const int r8 = lane >> 3, pc = lane & 7;

    const int t_first = t_begin > 0 ? t_begin - 1 : t_begin;                // warm-up tile: provides the halo of tile t_begin
    unsigned sb = ((unsigned)t_first * PJ_NT) / (unsigned)a.Lx;
    int sl0 = (int)((unsigned)t_first * PJ_NT - sb * (unsigned)a.Lx);
    const int t_whole = (int)(P / PJ_NT);                                    // tiles below this one are whole
    if (t_first < t_whole) issue_operand_tile2<K, true>(ubase, (unsigned)t_first * PJ_NT, P, ubuf + (t_first & 1) * C::UBUF, wave, lane PJ_VMQ_ARG);
    else issue_operand_tile2<K, false>(ubase, (unsigned)t_first * PJ_NT, P, ubuf + (t_first & 1) * C::UBUF, wave, lane PJ_VMQ_ARG);
    // A row m of position block pt <-> position 32 pt + 16 ((m >> 2) & 1) + 4 (m >> 3) + (m & 3): D row (r & 3) + 8 (r >> 2) + 4 hb is then position
    // 32 pt + 16 hb + r -- a lane's sixteen accumulators are sixteen consecutive positions of channel j
    const int arow = 16 * ((j >> 2) & 1) + 4 * (j >> 3) + (j & 3);
    const int ua = arow * C::UROWB, ux = (hb ^ (arow & 15)) * 16;            // slot of piece 2 ks + hb: (2 ks) ^ (hb ^ (position & 15)); 32 pt leaves position & 15 alone

    // the row phase of the PREVIOUS tile (prev_*), pending while this tile's product runs
    bool pend = false, pend_fast = false, counted = false;
    unsigned prev_p0 = 0, prev_sb = 0;
    int prev_sl0 = 0;

    // one unit of vg on an interior tile, in two halves (gi = 0: x1, gi = 1: v) so that they can sit in different gaps of the product
    auto vg_half = [&](int q, int gi, float (&prod)[8]) {
        const int ch = 8 * (unit0 + q) + r8;                                 // channel within the workgroup's 128
        const HY_LDS char* row = et + ((1 + gi) * IP2_CH + ch) * C::EROW + pc * 16;
        const Frag lo = lds_ld16(row), hi = lds_ld16(row + 16);
        elem_t pl[8], ph[8];
        __builtin_memcpy(pl, lo.w, 16);
        __builtin_memcpy(ph, hi.w, 16);
        float xs[10];
        const HY_LDS float* tp = taps + (gi * IP2_CH + ch) * 5;
        const float w0 = tp[0], w1 = tp[1], w2 = tp[2], bsc = tp[3], bin = tp[4];
        xs[0] = Elem<DT>::dec(pl[6]) + bin; xs[1] = Elem<DT>::dec(pl[7]) + bin;
        HY_UNROLL
        for (int i = 0; i < 8; ++i) xs[2 + i] = Elem<DT>::dec(ph[i]) + bin;
        HY_UNROLL
        for (int i = 0; i < 8; ++i) {
            const float c = __builtin_fmaf(w2, xs[i + 2], __builtin_fmaf(w1, xs[i + 1], __builtin_fmaf(w0, xs[i], bsc)));   // = cm_sc
            prod[i] = gi == 0 ? c : prod[i] * c;
        }
    };
    auto vg_store = [&](int q, const float (&prod)[8]) {
        const int ch = 8 * (unit0 + q) + r8;
        elem_t out[8];
        HY_UNROLL
        for (int i = 0; i < 8; ++i) out[i] = Elem<DT>::cvt(prod[i]);
        Frag vf;
        __builtin_memcpy(vf.w, out, 16);
        const size_t vbase = (size_t)prev_sb * D * a.ldv + (size_t)prev_sl0 + 8u * (unsigned)pc;
        PJ_ST16(reinterpret_cast<elem_t*>(a.vg) + (vbase + (size_t)(cg * IP2_CH + ch) * (size_t)a.ldv), vf);
    };
    // the predicated row phase: ragged tiles, sequence boundaries inside the tile, the first two positions of a sequence, positions beyond the convolved length
    auto row_phase_generic = [&]() {
        const unsigned p0 = prev_p0;
        HY_UNROLL
        for (int m = 0; m < 4; ++m) {
            const unsigned p = p0 + 8u * (unsigned)pc;
            const Frag v = lds_ld16(et + (lr0 + 8 * m + r8) * C::EROW + 16 + pc * 16);
            if (p >= P) continue;
            const unsigned xb_ = p / (unsigned)a.Lx;                         // the piece's first position: sequence, position within it
            const int xl = (int)(p - xb_ * (unsigned)a.Lx);
            elem_t* const crow = reinterpret_cast<elem_t*>(a.xT) + (size_t)(c0 + 8 * m + r8) * CS;
            if (xl + 8 <= a.Lx) PJ_ST16(crow + (size_t)xb_ * a.bsx + xl, v);
            else {                                                           // the piece runs into the next sequence's row (or past the last one)
                elem_t sv[8];
                __builtin_memcpy(sv, v.w, 16);
                for (int i = 0; i < 8; ++i) {
                    int li = xl + i;
                    unsigned bi = xb_;
                    if (li >= a.Lx) { li -= a.Lx; ++bi; }
                    if (p + i < P) crow[(size_t)bi * a.bsx + li] = sv[i];
                }
            }
        }
        HY_UNROLL
        for (int q = 0; q < 2; ++q) {
            if (q >= nunits) continue;
            const int ch = 8 * (unit0 + q) + r8;
            const unsigned p = p0 + 8u * (unsigned)pc;
            if (p >= P) continue;
            float prod[8];
            const unsigned b = p / (unsigned)a.Lx;
            const int l = (int)(p - b * (unsigned)a.Lx);
            HY_UNROLL
            for (int gi = 0; gi < 2; ++gi) {
                const HY_LDS char* row = et + ((1 + gi) * IP2_CH + ch) * C::EROW + pc * 16;
                const Frag lo = lds_ld16(row), hi = lds_ld16(row + 16);
                elem_t pl[8], ph[8];
                __builtin_memcpy(pl, lo.w, 16);
                __builtin_memcpy(ph, hi.w, 16);
                float xs[10];
                xs[0] = Elem<DT>::dec(pl[6]); xs[1] = Elem<DT>::dec(pl[7]);
                HY_UNROLL
                for (int i = 0; i < 8; ++i) xs[2 + i] = Elem<DT>::dec(ph[i]);
                const HY_LDS float* tp = taps + (gi * IP2_CH + ch) * 5;
                const float w0 = tp[0], w1 = tp[1], w2 = tp[2], bsc = tp[3], bin = tp[4];
                HY_UNROLL
                for (int i = 0; i < 8; ++i) {
                    int li = l + i;                                         // position within its sequence (the piece may cross into the next one)
                    if (li >= a.Lx) li -= a.Lx;
                    const float x0 = li >= 2 ? xs[i] + bin : 0.f, x1 = li >= 1 ? xs[i + 1] + bin : 0.f, x2 = xs[i + 2] + bin;
                    const float c = __builtin_fmaf(w2, x2, __builtin_fmaf(w1, x1, __builtin_fmaf(w0, x0, bsc)));   // = cm_sc
                    prod[i] = gi == 0 ? c : prod[i] * c;
                }
            }
            elem_t out[8];
            HY_UNROLL
            for (int i = 0; i < 8; ++i) out[i] = Elem<DT>::cvt(prod[i]);
            const int dch = cg * IP2_CH + ch;
            elem_t* vrow = reinterpret_cast<elem_t*>(a.vg) + ((size_t)b * D + dch) * a.ldv;
            if (l + 8 <= a.Lc) {
                Frag f;
                __builtin_memcpy(f.w, out, 16);
                PJ_ST16(vrow + l, f);
            } else {
                for (int i = 0; i < 8; ++i) {
                    int li = l + i;
                    unsigned bi = b;
                    if (li >= a.Lx) { li -= a.Lx; ++bi; }
                    if (p + i < P && li < a.Lc) reinterpret_cast<elem_t*>(a.vg)[((size_t)bi * D + dch) * a.ldv + li] = out[i];
                }
            }
        }
    };

#if defined(PJ_PROFILE) && !defined(HIPEMU)
    unsigned long long pj_d[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, pj_t, pj_start, pj_rt0, pj_rt1;
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(pj_rt0)::"memory");
    PJ_NOW(pj_t);
    pj_start = pj_t;
#endif
    // t = t_end is the drain iteration: no product, only the last tile's row phase
    for (int t = t_first; t <= t_end; ++t, sl0 += PJ_NT) {
        const bool have = t < t_end;
        while (sl0 >= a.Lx) { sl0 -= a.Lx; ++sb; }
        const unsigned p0 = (unsigned)t * PJ_NT;
        HY_LDS char* const ub = ubuf + (t & 1) * C::UBUF;
        // tile t has landed: behind its loads only the row-phase stores of the previous iteration were issued (an interior tile: 4 + nunits of them)
        if (counted) { if (wave < 4) PJ_VMWAIT(6); else PJ_VMWAIT(5); }
        else PJ_VMWAIT(0);
        PJ_MARK(0);
        PJ_BARRIER();                                                        // everybody's share of tile t has landed; the tile of t - 1 is parked; everybody has read the last fragment of t - 1
        PJ_MARK(1);
        if (t + 1 < t_end) {                                                 // tile t + 1 into the other buffer: a whole tile to land
            if (t + 1 < t_whole) issue_operand_tile2<K, true>(ubase, p0 + PJ_NT, P, ubuf + ((t + 1) & 1) * C::UBUF, wave, lane PJ_VMQ_ARG);
            else issue_operand_tile2<K, false>(ubase, p0 + PJ_NT, P, ubuf + ((t + 1) & 1) * C::UBUF, wave, lane PJ_VMQ_ARG);
        }
        counted = false;
        PJ_MARK(2);
        acc_t acc[2];
        HY_UNROLL
        for (int pt = 0; pt < 2; ++pt) {
            HY_UNROLL
            for (int r = 0; r < 16; ++r) acc[pt][r] = 0.f;
        }
        auto product = [&](int ks_lo, int ks_hi) {
            if (!have) return;
            HY_UNROLL
            for (int ks = ks_lo; ks < ks_hi; ++ks) {
                HY_UNROLL
                for (int pt = 0; pt < 2; ++pt) {
#ifdef IP2_DBG_NO_FRAG
                    const Frag af = wf[(ks + 1 + pt) % C::KS];            // (profiling builds only: no fragment reads -- results are wrong by construction)
#else
                    const Frag af = lds_ld16(ub + pt * 32 * C::UROWB + ua + (ux ^ (ks * 32)));
#endif
                    acc[pt] = mfma<DT>(af, wf[ks], acc[pt]);
                }
            }
        };
        constexpr int Q = C::KS / 8;                                         // matrix steps per gap: the product in eight slices (2 / 1 steps of 2 instructions)
        if (pend && pend_fast) {
            // interior tile -- whole, inside one sequence, at least two positions into it, inside the convolved length: no predicates, no divisions;
            // its pieces go between the slices of this tile's product
            counted = t + 1 < t_whole && t + 1 < t_end;
            const size_t xpos = (size_t)prev_sb * (size_t)a.bsx + (size_t)prev_sl0;
            Frag xr[4];
            float prod[8];
            auto xt_load = [&]() {
                HY_UNROLL
                for (int m = 0; m < 4; ++m) xr[m] = lds_ld16(et + (lr0 + 8 * m + r8) * C::EROW + 16 + pc * 16);
            };
            auto xt_store = [&](int m0, int m1) {
                HY_UNROLL
                for (int m = m0; m < m1; ++m)
                    PJ_ST16(reinterpret_cast<elem_t*>(a.xT) + ((size_t)(c0 + 8 * m + r8) * CS + xpos + 8u * (unsigned)pc), xr[m]);
            };
            // The three wavefronts of a SIMD are the x0, x1 and v wavefronts of one channel block, released together by the barrier: running the same
            // schedule they would all want the matrix pipe at once and then all the vector pipe at once.  So the three groups take the pieces in
            // different orders (IP2_STAGGER; 0 = everybody interleaved: measured 426 us without memory traffic where the matrix instructions alone are 200):
            // the x1 wavefronts do the previous tile's row phase FIRST, the v wavefronts LAST, the x0 wavefronts (twice the window arithmetic) in the gaps.
#ifndef IP2_STAGGER
#define IP2_STAGGER 1
#endif
            if (IP2_STAGGER && grp == 1) {
                xt_load();
                xt_store(0, 4);
                vg_half(0, 0, prod);
                vg_half(0, 1, prod);
                vg_store(0, prod);
                HY_SCHED_FENCE();
                product(0, C::KS);
            } else if (IP2_STAGGER && grp == 2) {
                product(0, C::KS);
                HY_SCHED_FENCE();
                xt_load();
                xt_store(0, 4);
                vg_half(0, 0, prod);
                vg_half(0, 1, prod);
                vg_store(0, prod);
            } else {
                product(0, Q);
                HY_SCHED_FENCE();
                xt_load();
                HY_SCHED_FENCE();
                product(Q, 2 * Q);
                HY_SCHED_FENCE();
                xt_store(0, 2);
                HY_SCHED_FENCE();
                product(2 * Q, 3 * Q);
                HY_SCHED_FENCE();
                xt_store(2, 4);
                HY_SCHED_FENCE();
                product(3 * Q, 4 * Q);
                HY_SCHED_FENCE();
                vg_half(0, 0, prod);
                HY_SCHED_FENCE();
                product(4 * Q, 5 * Q);
                HY_SCHED_FENCE();
                vg_half(0, 1, prod);
                vg_store(0, prod);
                HY_SCHED_FENCE();
                product(5 * Q, 6 * Q);
                HY_SCHED_FENCE();
                if (nunits > 1) vg_half(1, 0, prod);                         // (wave-uniform; the stores behind it are counted per role above)
                HY_SCHED_FENCE();
                product(6 * Q, 7 * Q);
                HY_SCHED_FENCE();
                if (nunits > 1) { vg_half(1, 1, prod); vg_store(1, prod); }
                HY_SCHED_FENCE();
                product(7 * Q, C::KS);
            }
        } else {
            if (pend) row_phase_generic();
            product(0, C::KS);
        }
        PJ_MARK(3);
        PJ_BARRIER();                                                        // everybody is done with the parked tile of t - 1
        PJ_MARK(4);
        if (have) {
            // the previous tile's last two positions become this tile's halo (slots 6, 7: one dword per row), each wavefront in its own rows
            if (lane < IP2_CB) {
                HY_LDS uint32_t* row = reinterpret_cast<HY_LDS uint32_t*>(et + (lr0 + lane) * C::EROW);
                row[3] = row[3 + PJ_NT / 2];
            }
            HY_WAVE_SYNC_PJ();
            // park: sixteen consecutive positions of channel j per block = two 16-byte pieces of the row
            HY_UNROLL
            for (int pt = 0; pt < 2; ++pt) {
                elem_t e16[16];
                HY_UNROLL
                for (int r = 0; r < 16; ++r) e16[r] = Elem<DT>::cvt(acc[pt][r]);
                Frag f0, f1;
                __builtin_memcpy(f0.w, e16, 16);
                __builtin_memcpy(f1.w, e16 + 8, 16);
                HY_LDS char* dst = et + (lr0 + j) * C::EROW + (8 + 32 * pt + 16 * hb) * 2;
                lds_st16(dst, f0);
                lds_st16(dst + 16, f1);
            }
        }
        // this tile's row phase runs under the next tile's product (the warm-up tile has none: it is parked for its last two positions only)
        pend = have && t >= t_begin;
        pend_fast = p0 + PJ_NT <= P && sl0 >= 2 && sl0 + PJ_NT <= a.Lc;
        prev_p0 = p0; prev_sb = sb; prev_sl0 = sl0;
        PJ_MARK(5);
    }
#if defined(PJ_PROFILE) && !defined(HIPEMU)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    PJ_MARK(9);
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(pj_rt1)::"memory");
    if (pj_prof_buf != nullptr && lane == 0) {
        unsigned long long* o = pj_prof_buf + ((size_t)blockIdx.x * IP2_WAVES + wave) * 16;
        HY_UNROLL
        for (int i = 0; i < 10; ++i) o[i] = pj_d[i];
        o[10] = (unsigned long long)(t_end - t_first + 1);
        o[11] = pj_t - pj_start;
        o[12] = pj_rt1 - pj_rt0;
    }
#endif
}

}  // namespace pj
}  // namespace hyena
