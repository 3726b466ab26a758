// hme_regions.hip - the level-1/2 HME kernel that does the four regions of a reference in one work item
// (hme_l12_regions_kernel), on hme_search.h's search functions.
#include "hme_search.h"

namespace svtme {

#ifdef SVTME_L2_PAIRS
constexpr int kL2AtOnce = 2;
#else
constexpr int kL2AtOnce = 1;
#endif
// The work item's LDS in dwords: without level 2 the level-1 block and small_search's window; with it 8 KB (five items per
// SIMD), of which level 1 takes its block and what is left as window, level 2 its larger block and the rest. 64 dwords of
// slack after a window as everywhere (pieces staged and rows read past its end).
template <bool L2>
constexpr int kRegionsLds = L2 ? (kL2AtOnce == 1 ? 2048 : 32 * kSrcPitch<false> + 2 * QuickGeo<16, 3>::WIN + 64) : 16 * kSrcPitch<false> + HI_WIN2_BYTES / 4 + 64;
template <bool L2>
constexpr int kRegionsWinBytes = L2 ? (kRegionsLds<true> - 32 * kSrcPitch<false> - 64) * 4 : HI_WIN2_BYTES; // (level 2's is the smaller one)
static_assert(kRegionsWinBytes<true> >= kL2AtOnce * QuickGeo<16, 3>::WIN * 4, "level 2's quick windows fit");
static_assert(kRegionsLds<true> - 16 * kSrcPitch<false> - 64 >= 4 * QuickGeo<8, 3>::WIN, "level 1's four quick windows fit");
static_assert(kRegionsWinBytes<false> >= 4 * QuickGeo<8, 3>::WIN * 4, "and without level 2");
static_assert(kRegionsWinBytes<true> / 4 / 63 >= 16 + 4, "small_search: a 64-pixel block's 63 window rows for four positions fit");

// Levels 1 and 2 of all four regions of reference (l, r) in one wave, for the shapes quick_search covers: what l12_region
// does per region, for windows of 8 x (1..3) positions (the pre-analysis ME's), with the four level-1 windows fetched in one
// memory round trip and the four level-2 windows in two, and the reference's record, the source blocks and the work item's
// set-up paid once instead of four times. A block or a window outside that shape (a picture's right / bottom edge) goes
// through small_search region by region. (The first pass's 8 x 8 windows keep a work item per region: with two trips per
// search the saving is smaller than what the larger item costs in resident waves.)
// L2 false: the controls have no level 2 (presets 7 and faster): that level's code, source block and windows are left out.
// The two levels use `lds` one after the other: [level 1's source block][its windows], then [level 2's source block][its
// window]; level 2's block was fetched with level 1's and waits in `blk2` (the launcher's work item holds as many bytes of
// LDS as the larger level needs, not both levels').
template <bool L2, class P>
__device__ __forceinline__ void l12_regions(const P &p, uint32_t *lds, const uint4 (&blk2)[2], int b64x, int b64y, int block_w, int block_h,
                                            int l, int r, HmeRefOut *rec) {
    const auto &k    = p.ctrl;
    const int   lane = threadIdx.x & 63;
    const bool  early = k.me_early_exit_th && rec->zz_sad < (k.me_early_exit_th >> 2);
    // region q = sh * 2 + sw lives at index i = sw * 2 + sh of the record's arrays
    int16_t  cx[4] = {0, 0, 0, 0}, cy[4] = {0, 0, 0, 0};
    uint32_t csad[4] = {0, 0, 0, 0};
    if (k.enable_hme_level0_flag && !early) {
        // the worst region (strict '>', sr_h outer / sr_w inner) gives way to the better pre-HME result (:1967-1999)
        uint32_t max_sad = 0;
        int      mi      = 0;
        for (int h_ = 0; h_ < 2; ++h_)
            for (int w_ = 0; w_ < 2; ++w_)
                if (rec->raw_sad[w_ * 2 + h_] > max_sad) {
                    max_sad = rec->raw_sad[w_ * 2 + h_];
                    mi      = w_ * 2 + h_;
                }
        const int s = rec->pre_sad[0] <= rec->pre_sad[1] ? 0 : 1;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int i = (q & 1) * 2 + (q >> 1);
            cx[q]   = rec->raw_x[i];
            cy[q]   = rec->raw_y[i];
            csad[q] = rec->raw_sad[i];
            if (k.prehme_enable && mi == i && rec->pre_sad[s] < csad[q]) {
                csad[q] = rec->pre_sad[s];
                cx[q]   = rec->pre_x[s];
                cy[q]   = rec->pre_y[s];
            }
            if (lane == 0) {
                rec->lx[0][i]   = cx[q];
                rec->ly[0][i]   = cy[q];
                rec->lsad[0][i] = csad[q];
            }
        }
    }
    for (int level = 1; level <= (L2 ? 2 : 1); ++level) {
        // a level that does not run, or level 1 on the early exit (:2032-2040; level 2 has none), leaves zeros
        if ((level == 1 ? !k.enable_hme_level1_flag : !k.enable_hme_level2_flag) || (level == 1 && early)) {
#pragma unroll
            for (int q = 0; q < 4; ++q) cx[q] = cy[q] = 0;
            continue;
        }
        const DevPlane ref   = load_plane(p.ref[l][r].lvl[level]);
        const int      shift = 2 - level;
        const int      ox = b64x >> shift, oy = b64y >> shift;
        const int      bw = block_w >> shift, bh = block_h >> shift;
        const int      rows = bh >> 1;
        const int      pad  = level == 2 ? 63 : ref.pad - 1;
        const int      sa_width = level == 1 ? k.hme_l1_sa.width : k.hme_l2_sa.width, sa_height = level == 1 ? k.hme_l1_sa.height : k.hme_l2_sa.height;
        int            wx[4], wy[4], hh[4];
        int16_t        ww[4], orgx[4], orgy[4];
        bool           quick = block_w == 64;
        unsigned long long ops = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            int16_t w = (int16_t)(((int16_t)sa_width + 7) & ~0x07), h = (int16_t)sa_height;
            if (level == 1) {
                cx[q] = (int16_t)(cx[q] >> 1);
                cy[q] = (int16_t)(cy[q] >> 1);
            }
            int16_t org_x = (int16_t)(-(w >> 1) + cx[q]), org_y = (int16_t)(-(h >> 1) + cy[q]);
            clip_axis(ox, org_x, w, -pad, ref.width);
            w = (int16_t)(w < 8 ? w : (w & ~0x07));
            clip_axis(oy, org_y, h, -pad, ref.height);
            ops += (unsigned long long)max((int)h, 0) * max((int)w, 0) * rows * bw;
            ww[q] = w, orgx[q] = org_x, orgy[q] = org_y, hh[q] = h;
            wx[q] = ox + org_x, wy[q] = oy + org_y;
            quick = quick && w == 8 && h >= 1 && h <= 3;
        }
        if (lane == 0) count_ops(p, 2 + level, ops);
        const uint32_t *src = lds;
        uint32_t       *win = lds + (level == 1 ? 16 : 32) * kSrcPitch<false>;
        u64             key[4];
        if (L2 && level == 2) {
            wave_sync(); // level 1's reads of its block and windows are done
            put_block<8, kSrcPitch<false>>(lds, blk2, block_w, block_h >> 1);
            wave_sync();
        }
        if (quick && level == 1) {
            wave_sync();
            quick_stage<8, 3, 4>(win, ref, rows, wx, wy, hh);
            wave_sync();
#pragma unroll
            for (int q = 0; q < 4; ++q) key[q] = quick_scan<8, 3>(src, win + q * QuickGeo<8, 3>::WIN, rows, hh[q]);
            wave_sync();
        } else if (quick && L2) { // the 64-pixel block's windows
#ifdef SVTME_L2_PAIRS // two at a time: half the round trips, 3.7 KB more LDS
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                const int wxp[2] = {wx[2 * half], wx[2 * half + 1]}, wyp[2] = {wy[2 * half], wy[2 * half + 1]};
                const int hp[2]  = {hh[2 * half], hh[2 * half + 1]};
                wave_sync();
                quick_stage<16, 3, 2>(win, ref, rows, wxp, wyp, hp);
                wave_sync();
                key[2 * half]     = quick_scan<16, 3>(src, win, rows, hp[0]);
                key[2 * half + 1] = quick_scan<16, 3>(src, win + QuickGeo<16, 3>::WIN, rows, hp[1]);
            }
#else // one at a time: the work item fits a quarter more often
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int wxp[1] = {wx[q]}, wyp[1] = {wy[q]}, hp[1] = {hh[q]};
                wave_sync();
                quick_stage<16, 3, 1>(win, ref, rows, wxp, wyp, hp);
                wave_sync();
                key[q] = quick_scan<16, 3>(src, win, rows, hp[0]);
            }
#endif
            wave_sync();
        } else {
#pragma unroll
            for (int q = 0; q < 4; ++q) key[q] = small_search<false, kRegionsWinBytes<L2>>(src, win, ref, bw, rows, wx[q], wy[q], ww[q], hh[q]);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int i = (q & 1) * 2 + (q >> 1);
            uint32_t  sad;
            decode_key(key[q], orgx[q], orgy[q], level == 1 ? 2 : 1, cx[q], cy[q], sad, true);
            if (lane == 0) {
                rec->lx[level][i]   = cx[q];
                rec->ly[level][i]   = cy[q];
                rec->lsad[level][i] = sad;
            }
        }
    }
}

// One wave per (b64, reference): the 2 x 2 regions' levels 1 and 2 (sub-sampled searches only). The kernel is a chain of
// memory round trips: its rate is the number of work items a SIMD holds, which the LDS sets (7.7 KB without level 2, 8 KB with).
template <class PICS, bool L2>
__global__ void __launch_bounds__(HI_THREADS) hme_l12_regions_kernel(PICS pp) {
    __shared__ __align__(16) uint32_t lds[kRegionsLds<L2>];
    const auto &p = pp[blockIdx.y];
    const auto &k = p.ctrl;
    const int   n_refs = p.n_ref[0] + (p.n_list == 2 ? p.n_ref[1] : 0);
    int         b64, item;
    if (!b64_band_item(blockIdx.x, p.b64_w * p.b64_h, n_refs, b64, item) || b64_left_out(p, b64)) return;
    int       b64x, b64y;
    b64_origin(p, b64, b64x, b64y);
    const int block_w = min(p.width - b64x, 64), block_h = min(p.height - b64y, 64);
    int       l = 0, r = item;
    if (r >= p.n_ref[0]) {
        r -= p.n_ref[0];
        l = 1;
    }
    if (!ref_searched(p, l) || !k.enable_hme_flag) return;
    // sampled rows of the quarter-resolution 32x32 block and of the full-resolution 64x64 block
    uint4 blk2[2] = {};
    if (k.enable_hme_level1_flag) stage_block<8, 2, kSrcPitch<false>>(lds, load_plane(p.cur.lvl[1]), b64x >> 1, b64y >> 1, block_w >> 1, block_h >> 2);
    if (L2 && k.enable_hme_level2_flag) fetch_block<8, 2>(blk2, load_plane(p.cur.lvl[2]), b64x, b64y, block_w, block_h >> 1);
    wave_sync();
    l12_regions<L2>(p, lds, blk2, b64x, b64y, block_w, block_h, l, r, &p.hme_out[(b64 * SVTME_MAX_LISTS + l) * SVTME_MAX_REFS + r]);
}

void launch_hme_l12_regions(const Batch &b, hipStream_t s) {
    const dim3 grid((b.n_b64() + 7) / 8 * 8 * b.max_refs(), b.n);
    const bool l2 = b.host[0].ctrl.enable_hme_level2_flag != 0; // (the controls are the batch's)
    if (b.n == 1) {
        if (l2) hipLaunchKernelGGL((hme_l12_regions_kernel<OnePicture, true>), grid, dim3(HI_THREADS), 0, s, OnePicture{b.host[0]});
        else hipLaunchKernelGGL((hme_l12_regions_kernel<OnePicture, false>), grid, dim3(HI_THREADS), 0, s, OnePicture{b.host[0]});
    } else {
        if (l2) hipLaunchKernelGGL((hme_l12_regions_kernel<ManyPictures, true>), grid, dim3(HI_THREADS), 0, s, ManyPictures{b.dev});
        else hipLaunchKernelGGL((hme_l12_regions_kernel<ManyPictures, false>), grid, dim3(HI_THREADS), 0, s, ManyPictures{b.dev});
    }
}
} // namespace svtme
