// tpl_kernels.hip - the source-based mode search of the TPL dispenser (tpl_mc_flow_dispenser_sb_generic,
// Source/Lib/Encoder/Codec/EbSourceBasedOperationsProcess.c:632-933) for the controls whose costs are sums of absolute
// differences: per block of the dispenser's grid the DC intra prediction from the source picture's own neighbours and every
// single-reference ME candidate at its clamped full-pel vector. One wave per block; a lane holds the block's pixels
// 4 at a time (dword idx = lane + 64 k of the block's rows), so a 16x16 block is one dword per lane and a 64x64 block 16.
#include "search_common.h"

namespace svtme {

namespace {

constexpr int kTplPad = 32; // TPL_PADX, TPL_PADY: EbEncodeContext.h:43-44

// tpl_blk_idx_tab (:514-516): block i's index in the ME results
__constant__ const uint8_t kTplMeIndex[SVTME_TPL_BLOCKS] = {0, 1, 2, 3, 4, 5, 6, 9, 10, 7, 8, 11, 12, 13, 14, 17, 18, 15, 16, 19, 20};

template <int S>
__device__ __forceinline__ void tpl_block(const TplParams &p, int b64, int bi, int x, int y) {
    constexpr int DW = S / 4, N = S * S / 256; // dwords per row, dwords per lane
    const int     lane = threadIdx.x & 63;
    const DevPlane cur = p.cur;
    uint32_t       src[N];
#pragma unroll
    for (int k = 0; k < N; ++k) {
        const int idx = lane + 64 * k, r = idx / DW, d = idx % DW;
        src[k]        = *reinterpret_cast<const uint32_t *>(cur.pix + (ptrdiff_t)(y + r) * cur.stride + x + 4 * d);
    }
    long long best_intra = INT64_MAX, best_inter = INT64_MAX;
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
        const long long sad = wave_sum(acc);
        if (sad < best_inter) { // strictly smaller: the first of equal candidates stays
            best_inter = sad;
            best_rf    = list == 0 ? ref : 4 + ref; // svt_get_ref_frame_type(list, ref) - 1
            best_mvx   = mvx;
            best_mvy   = mvy;
        }
    }
    if (lane == 0) {
        svtme_tpl_block o;
        o.best_intra_cost = best_intra;
        o.best_inter_cost = best_inter;
        o.mv_x            = (int16_t)best_mvx;
        o.mv_y            = (int16_t)best_mvy;
        o.best_rf_idx     = (int8_t)best_rf;
        o.best_is_inter   = best_inter < best_intra; // :932-933
        o.best_intra_mode = 0; // DC_PRED
        o.valid           = 1;
        p.out[(size_t)b64 * SVTME_TPL_BLOCKS + bi] = o;
    }
}

// grid (16, n_b64), one wave each: slot s of a 64x64 block is its s-th 16x16 block (level 0), its s-th 32x32 block
// (level 1, s < 4) or the block itself (level 2, s == 0); blocks that are not complete run at level 0 (:2086-2088)
__global__ void __launch_bounds__(64) tpl_search_kernel(const TplParams p) {
    const int b64 = blockIdx.y, slot = blockIdx.x;
    const int sb_x = (b64 % p.b64_w) * 64, sb_y = (b64 / p.b64_w) * 64;
    const int level = (sb_x + 64 <= p.width && sb_y + 64 <= p.height) ? p.level : 0;
    if (level == 0) {
        const int q = slot >> 2, s = slot & 3; // z-order: 32x32 quadrant, 16x16 inside it
        const int x = sb_x + (q & 1) * 32 + (s & 1) * 16, y = sb_y + (q >> 1) * 32 + (s >> 1) * 16;
        if (x + 8 > p.width || y + 8 > p.height) return; // at least half of the block inside (:604-607)
        tpl_block<16>(p, b64, 5 + slot, x, y);
    } else if (level == 1) {
        if (slot >= 4) return;
        tpl_block<32>(p, b64, 1 + slot, sb_x + (slot & 1) * 32, sb_y + (slot >> 1) * 32);
    } else {
        if (slot >= 1) return;
        tpl_block<64>(p, b64, 0, sb_x, sb_y);
    }
}

} // namespace

void launch_tpl_search(const TplParams &p, hipStream_t s) {
    hipLaunchKernelGGL(tpl_search_kernel, dim3(16, p.n_b64), dim3(64), 0, s, p);
}

} // namespace svtme
