// tpl_kernels.hip - the source-based mode search of the TPL dispenser (tpl_mc_flow_dispenser_sb_generic,
// Source/Lib/Encoder/Codec/EbSourceBasedOperationsProcess.c:632-933): per block of the dispenser's grid the DC intra
// prediction from the source picture's own neighbours and every single-reference ME candidate at its clamped full-pel
// vector, costed by the sum of absolute differences or by the sum of the absolute coefficients of the residual's 16x16
// forward DCT, and the quantisation error of the candidate that wins (:935-982). One wave per block; a lane holds the
// block's pixels 4 at a time (dword idx = lane + 64 k of the block's rows), so a 16x16 block is one dword per lane and a
// 64x64 block 16. The transform runs on 16 lanes of the wave: a lane per column, then a lane per row, through LDS.
#include "search_common.h"

namespace svtme {

namespace {

constexpr int kTplPad = 32; // TPL_PADX, TPL_PADY: EbEncodeContext.h:43-44

// tpl_blk_idx_tab (:514-516): block i's index in the ME results
__constant__ const uint8_t kTplMeIndex[SVTME_TPL_BLOCKS] = {0, 1, 2, 3, 4, 5, 6, 9, 10, 7, 8, 11, 12, 13, 14, 17, 18, 15, 16, 19, 20};


// ---- the 16x16 forward DCT of svt_av1_wht_fwd_txfm (EbTransforms.c:3815-3830 -> av1_tranform_two_d_core_c, :2301-2370:
// input << 2, 16-point transform of every column with 13-bit cosines, rounded >> 2, 16-point transform of every row with
// 12-bit cosines; fwd_shift_16x16 and the cosine bits: EbTransforms.h:28, :46-57). The 16-point network is
// svt_av1_fdct16_new's (:183-336), written part by part: its only roundings are the two-term rotations
// (a ca + b cb + 2^(bit-1)) >> bit (half_btf, EbInvTransforms.h:289-300), sums and differences are exact. With 8-bit
// residuals every product stays below 2^30, so 32-bit arithmetic gives what the reference's 64-bit sums give.
template <int BIT>
__device__ __forceinline__ int32_t rot(int32_t a, int32_t ca, int32_t b, int32_t cb) {
    return (a * ca + b * cb + (1 << (BIT - 1))) >> BIT;
}

// c[k] = cos(4 k pi / 128) with BIT fractional bits
template <int BIT>
__device__ __forceinline__ void fdct16(const int32_t (&x)[16], int32_t (&X)[16], const int32_t *c) {
    int32_t s[8], d[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        s[i] = x[i] + x[15 - i];
        d[i] = x[7 - i] - x[8 + i];
    }
    // even half: the 8-point transform of s
    const int32_t p0 = s[0] + s[7], p1 = s[1] + s[6], p2 = s[2] + s[5], p3 = s[3] + s[4];
    const int32_t q0 = s[3] - s[4], q1 = s[2] - s[5], q2 = s[1] - s[6], q3 = s[0] - s[7];
    const int32_t a0 = p0 + p3, a1 = p1 + p2, a2 = p1 - p2, a3 = p0 - p3;
    X[0]  = rot<BIT>(a0, c[8], a1, c[8]);
    X[8]  = rot<BIT>(a0, c[8], a1, -c[8]);
    X[4]  = rot<BIT>(a2, c[12], a3, c[4]);
    X[12] = rot<BIT>(a3, c[12], a2, -c[4]);
    const int32_t m1 = rot<BIT>(q2, c[8], q1, -c[8]), m2 = rot<BIT>(q2, c[8], q1, c[8]);
    const int32_t e0 = q0 + m1, e1 = q0 - m1, e2 = q3 - m2, e3 = q3 + m2;
    X[2]  = rot<BIT>(e0, c[14], e3, c[2]);
    X[10] = rot<BIT>(e1, c[6], e2, c[10]);
    X[6]  = rot<BIT>(e2, c[6], e1, -c[10]);
    X[14] = rot<BIT>(e3, c[14], e0, -c[2]);
    // odd half
    const int32_t g2 = rot<BIT>(d[5], c[8], d[2], -c[8]), g5 = rot<BIT>(d[5], c[8], d[2], c[8]);
    const int32_t g3 = rot<BIT>(d[4], c[8], d[3], -c[8]), g4 = rot<BIT>(d[4], c[8], d[3], c[8]);
    const int32_t h0 = d[0] + g3, h3 = d[0] - g3, h1 = d[1] + g2, h2 = d[1] - g2;
    const int32_t h4 = d[7] - g4, h7 = d[7] + g4, h5 = d[6] - g5, h6 = d[6] + g5;
    const int32_t k1 = rot<BIT>(h6, c[12], h1, -c[4]), k6 = rot<BIT>(h6, c[4], h1, c[12]);
    const int32_t k2 = rot<BIT>(h2, -c[12], h5, -c[4]), k5 = rot<BIT>(h5, c[12], h2, -c[4]);
    const int32_t n0 = h0 + k1, n1 = h0 - k1, n2 = h3 - k2, n3 = h3 + k2;
    const int32_t n4 = h4 + k5, n5 = h4 - k5, n6 = h7 - k6, n7 = h7 + k6;
    X[1]  = rot<BIT>(n0, c[15], n7, c[1]);
    X[9]  = rot<BIT>(n1, c[7], n6, c[9]);
    X[5]  = rot<BIT>(n2, c[11], n5, c[5]);
    X[13] = rot<BIT>(n3, c[3], n4, c[13]);
    X[3]  = rot<BIT>(n4, c[3], n3, -c[13]);
    X[11] = rot<BIT>(n5, c[11], n2, -c[5]);
    X[7]  = rot<BIT>(n6, c[7], n1, -c[9]);
    X[15] = rot<BIT>(n7, c[15], n0, -c[1]);
}

constexpr int kTplPitch = 17; // LDS row pitch in dwords: the 16 lanes of a pass read a column without bank conflicts

// The residual src - prd of a 16x16 block (a lane holds 4 pixels of row lane >> 2) -> sum of the absolute coefficients of
// its forward DCT (svt_aom_satd, common_dsp_rtcd.c:47-54) and, when `quant`, get_quantize_error's recon_error
// (:167-193: svt_av1_quantize_fp_c without quantisation matrix, EbFullLoop.c:318-405, block error >> 2, at least 1), in
// every lane. pf_shape N2 / N4 keep the top-left 8x8 / 4x4 coefficients (av1_tranform_two_d_core_N2_c / _N4_c).
__device__ __forceinline__ void tpl_transform16(const TplParams &p, uint32_t src, uint32_t prd, int32_t *lds, bool quant, long long &satd,
                                                long long &recon_error) {
    const int lane = threadIdx.x & 63, r = lane >> 2, c4 = (lane & 3) * 4;
    wave_sync(); // the previous use of the buffer is over
#pragma unroll
    for (int j = 0; j < 4; ++j) lds[r * kTplPitch + c4 + j] = (int32_t)((src >> (8 * j)) & 0xff) - (int32_t)((prd >> (8 * j)) & 0xff);
    wave_sync();
    int32_t in[16], out[16];
    if (lane < 16) {
#pragma unroll
        for (int i = 0; i < 16; ++i) in[i] = lds[i * kTplPitch + lane] * 4;
        fdct16<13>(in, out, p.cos13);
    }
    wave_sync();
    if (lane < 16) {
#pragma unroll
        for (int i = 0; i < 16; ++i) lds[i * kTplPitch + lane] = (out[i] + 2) >> 2;
    }
    wave_sync();
    long long a_sum = 0, e_sum = 0;
    const int keep = p.pf_shape == 2 ? 4 : (p.pf_shape == 1 ? 8 : 16);
    if (lane < keep) {
#pragma unroll
        for (int i = 0; i < 16; ++i) in[i] = lds[lane * kTplPitch + i];
        fdct16<12>(in, out, p.cos12);
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int32_t v = i < keep ? out[i] : 0;
            const int32_t a = v < 0 ? -v : v;
            a_sum += a;
            if (quant) {
                const int ac = (lane | i) != 0;
                int32_t   dq = 0;
                if ((a << 1) >= p.dequant[ac]) {
                    const int32_t t     = min(a + p.round_fp[ac], 32767);
                    const int32_t level = (int32_t)(((long long)t * p.quant_fp[ac]) >> 16);
                    dq                  = level * p.dequant[ac];
                }
                const long long diff = (long long)a - dq; // |v - sign(v) dq|: same square
                e_sum += diff * diff;
            }
        }
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        a_sum += __shfl_xor(a_sum, m);
        e_sum += __shfl_xor(e_sum, m);
    }
    satd        = a_sum;
    e_sum >>= 2;
    recon_error = e_sum < 1 ? 1 : e_sum;
}

template <int S>
__device__ __forceinline__ void tpl_block(const TplParams &p, int b64, int bi, int x, int y, int32_t *lds) {
    constexpr int DW = S / 4, N = S * S / 256; // dwords per row, dwords per lane
    const int     lane = threadIdx.x & 63;
    const DevPlane cur = p.cur;
    uint32_t       src[N];
#pragma unroll
    for (int k = 0; k < N; ++k) {
        const int idx = lane + 64 * k, r = idx / DW, d = idx % DW;
        src[k]        = *reinterpret_cast<const uint32_t *>(cur.pix + (ptrdiff_t)(y + r) * cur.stride + x + 4 * d);
    }
    long long best_intra = INT64_MAX, best_inter = INT64_MAX, best_err = 0;
    int       best_rf = -1, best_mvx = 0, best_mvy = 0;

    if (!p.disable_intra_pred) {
        // the row above and the column left of the block; entries past the picture's right / bottom edge keep the fill
        // values 127 / 129 (update_neighbor_samples_array_open_loop_mb, EbEncIntraPrediction.c:1150-1235); which sides
        // count: dc_pred[x > 0][y > 0] (EbIntraPrediction.c:985-1035, :2705-2707)
        uint32_t sum = 0;
        if (lane < S) {
            if (y > 0) sum += x + lane < p.width ? cur.pix[(ptrdiff_t)(y - 1) * cur.stride + x + lane] : 127u;
            if (x > 0) sum += y + lane < p.height ? cur.pix[(ptrdiff_t)(y + lane) * cur.stride + x - 1] : 129u;
        }
        sum = wave_sum(sum);
        const uint32_t n  = (y > 0 ? S : 0) + (x > 0 ? S : 0);
        const uint32_t dc = n ? (sum + (n >> 1)) / n : 128u;
        uint32_t       acc = 0;
#pragma unroll
        for (int k = 0; k < N; ++k) acc = __builtin_amdgcn_sad_u8(src[k], dc * 0x01010101u, acc);
        best_intra = wave_sum(acc);
        if constexpr (S == 16) {
            if (!p.sad_intra) { // :790-813
                long long err;
                tpl_transform16(p, src[0], dc * 0x01010101u, lds, false, best_intra, err);
            }
        }
    }

    int me_off = kTplMeIndex[bi]; // :826-834
    if (!p.enable_me_16x16) me_off = (me_off - 1) / 4;
    const int      n_cand = p.i_slice ? 0 : p.cnt[(size_t)b64 * p.n_pu + me_off];
    const uint8_t *cand   = p.cand + ((size_t)b64 * p.n_pu + me_off) * p.max_cand;
    const uint32_t *mvs   = p.mv + ((size_t)b64 * p.n_pu + me_off) * p.max_refs;
    for (int ci = 0; ci < n_cand; ++ci) { // :836-930
        const uint32_t c = cand[ci];
        if ((c & 3) > 1) continue; // single references only
        const int list = c & 3, ref = list == 0 ? (c >> 2) & 3 : (c >> 4) & 3;
        if (ref >= p.n_ref[list]) { // a candidate that names a reference the caller did not hand over
            if (lane == 0) atomicOr(p.error, 1);
            continue;
        }
        if (p.ref_skip[list][ref]) continue;
        const uint32_t mv  = mvs[(list ? p.max_l0 : 0) + ref];
        int16_t        mvx = (int16_t)((int16_t)(mv & 0xffff) << 1), mvy = (int16_t)((int16_t)(mv >> 16) << 1);
        if (x + (mvx >> 3) < -kTplPad) mvx = (int16_t)((-kTplPad - x) << 3);
        if (x + S + (mvx >> 3) > kTplPad + p.width - 1) mvx = (int16_t)(((kTplPad + p.width - 1) - (x + S)) << 3);
        if (y + (mvy >> 3) < -kTplPad) mvy = (int16_t)((-kTplPad - y) << 3);
        if (y + S + (mvy >> 3) > kTplPad + p.height - 1) mvy = (int16_t)(((kTplPad + p.height - 1) - (y + S)) << 3);
        const DevPlane rp  = p.ref[list][ref];
        const uint8_t *org = rp.pix + (ptrdiff_t)(y + (mvy >> 3)) * rp.stride + x + (mvx >> 3);
        uint32_t       acc = 0;
#pragma unroll
        for (int k = 0; k < N; ++k) {
            const int idx = lane + 64 * k, r = idx / DW, d = idx % DW;
            uint32_t  f;
            __builtin_memcpy(&f, org + (ptrdiff_t)r * rp.stride + 4 * d, 4);
            acc = __builtin_amdgcn_sad_u8(src[k], f, acc);
        }
        long long sad = wave_sum(acc), err = 0;
        if constexpr (S == 16) {
            if (!p.sad_inter) { // :894-917
                uint32_t f;
                __builtin_memcpy(&f, org + (ptrdiff_t)(lane >> 2) * rp.stride + 4 * (lane & 3), 4);
                tpl_transform16(p, src[0], f, lds, p.src_dist != 0, sad, err);
            }
        }
        if (sad < best_inter) { // strictly smaller: the first of equal candidates stays
            best_inter = sad;
            best_err   = err;
            best_rf    = list == 0 ? ref : 4 + ref; // svt_get_ref_frame_type(list, ref) - 1
            best_mvx   = mvx;
            best_mvy   = mvy;
        }
    }
    const bool is_inter = best_inter < best_intra; // :932-933
    if constexpr (S == 16) {
        if (p.src_dist && p.sad_inter && is_inter) { // the transform of the chosen candidate (:938-970)
            const DevPlane rp = p.ref[best_rf >> 2][best_rf & 3];
            uint32_t       f;
            __builtin_memcpy(&f, rp.pix + (ptrdiff_t)(y + (best_mvy >> 3) + (lane >> 2)) * rp.stride + x + (best_mvx >> 3) + 4 * (lane & 3), 4);
            long long satd;
            tpl_transform16(p, src[0], f, lds, true, satd, best_err);
        }
    }
    if (lane == 0) {
        svtme_tpl_block o;
        o.best_intra_cost = best_intra;
        o.best_inter_cost = best_inter;
        o.mv_x            = (int16_t)best_mvx;
        o.mv_y            = (int16_t)best_mvy;
        o.best_rf_idx     = (int8_t)best_rf;
        o.best_is_inter   = is_inter;
        o.best_intra_mode = 0; // DC_PRED
        o.valid           = 1;
        o.src_recon_error = p.src_dist && is_inter ? best_err : 0;
        p.out[(size_t)b64 * SVTME_TPL_BLOCKS + bi] = o;
    }
}

// grid (16, n_b64), one wave each: slot s of a 64x64 block is its s-th 16x16 block (level 0), its s-th 32x32 block
// (level 1, s < 4) or the block itself (level 2, s == 0); blocks that are not complete run at level 0 (:2086-2088)
__global__ void __launch_bounds__(64) tpl_search_kernel(const TplParams p) {
    __shared__ int32_t lds[16 * kTplPitch];
    const int b64 = blockIdx.y, slot = blockIdx.x;
    const int sb_x = (b64 % p.b64_w) * 64, sb_y = (b64 / p.b64_w) * 64;
    const int level = (sb_x + 64 <= p.width && sb_y + 64 <= p.height) ? p.level : 0;
    if (level == 0) {
        const int q = slot >> 2, s = slot & 3; // z-order: 32x32 quadrant, 16x16 inside it
        const int x = sb_x + (q & 1) * 32 + (s & 1) * 16, y = sb_y + (q >> 1) * 32 + (s >> 1) * 16;
        if (x + 8 > p.width || y + 8 > p.height) return; // at least half of the block inside (:604-607)
        tpl_block<16>(p, b64, 5 + slot, x, y, lds);
    } else if (level == 1) {
        if (slot >= 4) return;
        tpl_block<32>(p, b64, 1 + slot, sb_x + (slot & 1) * 32, sb_y + (slot >> 1) * 32, lds);
    } else {
        if (slot >= 1) return;
        tpl_block<64>(p, b64, 0, sb_x, sb_y, lds);
    }
}

} // namespace

void launch_tpl_search(const TplParams &p, hipStream_t s) {
    hipLaunchKernelGGL(tpl_search_kernel, dim3(16, p.n_b64), dim3(64), 0, s, p);
}

} // namespace svtme
