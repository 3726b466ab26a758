// hme_search.h - the search functions the kernels of hme_items.hip and hme_regions.hip share: pre-HME and HME level 0 of a
// work item (strip_units, l0_search, l0_item), the level-1 / level-2 searches of a region (small_search, strip_search_fs,
// quick_search and its window geometry) and the staging of source blocks. The two files are translation units of their own:
// with all kernels in one unit the compiler's inlining of these functions changed, and the per-region kernels lost a tenth
// of their speed.
//
// Same reference code as hme_kernel.hip (Source/Lib/Encoder/Codec/EbMotionEstimation.c: init_zz_sad :2318-2333,
// prehme_b64 / prehme_core :1773-1823 / :1621-1719, hme_level0_b64 / hme_level_0 / get_hme_l0_search_area :1899-2004 /
// :847-947 / :1827-1894, hme_level1_b64 / hme_level_1 :2009-2068 / :950-1049, hme_level2_b64 / hme_level_2 :2073-2112 /
// :1052-1140, svt_sad_loop_kernel EbComputeSAD_C.c:58-103), for the common case in which no reference depends on
// another one's result, so that every search of a picture can run at the same time:
//
//  hme_l0_kernel   one wave per (b64, reference, search), search = one of the 2 pre-HME shapes or one of the 2x2
//                  level-0 regions: ~15 000 items per 1080p picture (for the small level-0 areas of the fast presets:
//                  one item for the four regions of a reference, the two of a row searched as one window). The 16x16
//                  block of the sixteenth-resolution picture (8 sampled rows x 4 dwords) sits in scalar registers
//                  (scalar loads: it is the same for every lane); the reference window is staged in LDS. A lane
//                  owns 4 horizontally adjacent positions x T same-parity search rows and walks them together, staggered
//                  by one sampled source row, so each reference row is read from LDS once for 4 T positions: 32 T
//                  v_qsad_pk_u16_u8 per unit into T packed accumulators.
//  hme_l12_kernel  one wave per (b64, reference, region): replaces the worst level-0 region by a better pre-HME result,
//                  then searches level 1 around the level-0 vector and level 2 around the level-1 vector. Few positions
//                  (8x3) of a large block (32x32, 64x64): the block's rows are split over 8 lanes per position quad.
//
// The argmin is exact under the reference's rule (strict '<' in raster order): 64-bit keys (SAD << 32 | y << 16 | x).
//
// Both kernels exist in two forms: SUB_SAD_SEARCH (every other block row, SAD x2: every preset of the pre-analysis ME)
// and FULL_SAD_SEARCH (all block rows: the temporal filter's searches, EbTemporalFiltering.c:331). The template
// parameter FS selects the form; the sub-sampled form's code and resources do not depend on the other's.
#ifndef SVTME_HME_SEARCH_H
#define SVTME_HME_SEARCH_H
#include <stdlib.h>
#include <type_traits>
#include "search_common.h"

namespace svtme {

constexpr int HI_THREADS   = 64;
constexpr int HI_WIN_BYTES = 6144; // LDS budget of one staged reference window (with it five work items per SIMD are resident)
constexpr int HI_XCHUNK    = 96; // search positions per staged column chunk (level 0 / pre-HME); 128 for windows wider than that
constexpr int HI_WIN2_BYTES = 6144; // levels 1 and 2
constexpr int HI_WIN2_BYTES_FS = 8192; // the same with FULL_SAD_SEARCH: 16 + 63 rows of the 64x64 block's 16x16 window
constexpr int HI_THREADS_FS = 64; // threads of a level-1/2 work item with FULL_SAD_SEARCH

// Level-0 vector of region (0,0) of list 0 / reference 0 (record `o`) as hme_level0_b64 leaves it (:1967-1999): the
// region's own result unless it is the worst of the regions and a pre-HME result beats it; zero on the zero-MV early exit
// (:1916).
template <class P>
__device__ void first_ref_l0_centre(const P &p, const HmeRefOut &o, int &mvx, int &mvy) {
    const auto &k = p.ctrl;
    mvx = mvy = 0;
    if (!k.enable_hme_level0_flag || (k.me_early_exit_th && o.zz_sad < (k.me_early_exit_th >> 2))) return;
    mvx = o.raw_x[0];
    mvy = o.raw_y[0];
    if (k.prehme_enable) {
        uint32_t max_sad = 0;
        int      mi      = 0;
        for (int h_ = 0; h_ < k.num_hme_sa_h; ++h_)
            for (int w_ = 0; w_ < k.num_hme_sa_w; ++w_)
                if (o.raw_sad[w_ * 2 + h_] > max_sad) {
                    max_sad = o.raw_sad[w_ * 2 + h_];
                    mi      = w_ * 2 + h_;
                }
        const int s = o.pre_sad[0] <= o.pre_sad[1] ? 0 : 1;
        if (mi == 0 && o.pre_sad[s] < o.raw_sad[0]) {
            mvx = o.pre_x[s];
            mvy = o.pre_y[s];
        }
    }
}

// get_hme_l0_search_area :1827-1894. With p.dep_l0 every reference but the first reads the first one's level-0
// vector (`first`: the record of list 0 / reference 0, complete by then).
template <class P>
__device__ void l0_area(const P &p, const HmeRefOut &first, int l, int r, int16_t &sa_w, int16_t &sa_h) {
    const auto &k = p.ctrl;
    svtme_sa_minmax sa;
    sa.sa_min.width = k.hme_l0_sa.sa_min.width, sa.sa_min.height = k.hme_l0_sa.sa_min.height;
    sa.sa_max.width = k.hme_l0_sa.sa_max.width, sa.sa_max.height = k.hme_l0_sa.sa_max.height;
    if (k.enable_me_sr_adjustment && k.distance_based_hme_resizing) {
        int is_hor = 1, is_ver = 1, is_still = 0;
        if (k.reduce_hme_l0_sr_th_min && k.reduce_hme_l0_sr_th_max && (l || r)) {
            int mvx, mvy;
            first_ref_l0_centre(p, first, mvx, mvy);
            is_ver   = abs(mvx) < k.reduce_hme_l0_sr_th_min && abs(mvy) > k.reduce_hme_l0_sr_th_max;
            is_hor   = abs(mvx) > k.reduce_hme_l0_sr_th_max && abs(mvy) < k.reduce_hme_l0_sr_th_min;
            is_still = abs(mvx) < k.reduce_hme_l0_sr_th_min * 3 && abs(mvy) < k.reduce_hme_l0_sr_th_min * 3;
        }
        int x_off = is_hor ? 1 : 2, y_off = is_ver ? 1 : 2;
        if (k.enable_me_sr_adjustment == 2 && is_still) x_off = y_off = 4;
        // (divisors 1..7: through the table of magic numbers; operands below 2^16)
        const uint32_t dx = (uint32_t)(x_off + r), dy = (uint32_t)(y_off + r), mx = div_magic(dx), my = div_magic(dy);
        sa.sa_min.width  = (uint16_t)magic_div(sa.sa_min.width, dx, mx);
        sa.sa_min.height = (uint16_t)magic_div(sa.sa_min.height, dy, my);
        sa.sa_max.width  = (uint16_t)magic_div(sa.sa_max.width, dx, mx);
        sa.sa_max.height = (uint16_t)magic_div(sa.sa_max.height, dy, my);
    }
    const int factor = scaled_dist(p.dist[l][r]);
    // the number of regions per direction is 1 or 2: the divisions are shifts
    const int shw = k.num_hme_sa_w - 1, shh = k.num_hme_sa_h - 1;
    int16_t   w   = (int16_t)(sa.sa_min.width >> shw);
    w = (int16_t)min(((w * factor) + 15) & ~0x0F, ((sa.sa_max.width >> shw) + 15) & ~0x0F);
    int16_t h = (int16_t)(sa.sa_min.height >> shh);
    h         = (int16_t)min(h * factor, sa.sa_max.height >> shh);
    sa_w      = w;
    sa_h      = h;
}

// What the callers of svt_sad_loop_kernel make of its outputs: SAD x2 (rows were subsampled), position + window
// origin, times the level's scale (:938-944, :1040-1046, :1133-1137, :1710-1716).
// The low half of a key is y << 16 | x, the position inside the window: keys order like the reference's raster scan.
__device__ __forceinline__ void decode_key(u64 key, int org_x, int org_y, int scale, int16_t &mx, int16_t &my, uint32_t &sad,
                                           bool sub_sad = true) {
    if (key == ~0ull) { // no position evaluated: svt_sad_loop_kernel leaves 0xffffff and does not touch x / y
        sad = sub_sad ? 0xffffffu * 2 : 0xffffffu;
        mx  = (int16_t)(org_x * scale);
        my  = (int16_t)(org_y * scale);
        return;
    }
    const int y = (int)((uint32_t)key >> 16), x = (int)((uint32_t)key & 0xFFFFu);
    sad = (uint32_t)(key >> 32) * (sub_sad ? 2 : 1);
    mx  = (int16_t)((int16_t)(x + org_x) * scale);
    my  = (int16_t)((int16_t)(y + org_y) * scale);
}

// One position straight from global memory: any block width (multiple of 2) and row count. Used for the blocks of a
// picture's right edge, whose width is not the full 64 / 32 / 16 pixels.
__device__ uint32_t slow_sad(const DevPlane &cur, const DevPlane &ref, int ox, int oy, int rx, int ry, int bw, int rows, int step) {
    uint32_t sad = 0;
    for (int j = 0; j < rows; ++j) {
        const uint8_t *s = cur.pix + (ptrdiff_t)(oy + j * step) * cur.stride + ox;
        const uint8_t *f = ref.pix + (ptrdiff_t)(ry + j * step) * ref.stride + rx;
        int            i = 0;
        for (; i + 4 <= bw; i += 4) {
            uint32_t a, b;
            __builtin_memcpy(&a, s + i, 4);
            __builtin_memcpy(&b, f + i, 4);
            sad = __builtin_amdgcn_sad_u8(a, b, sad);
        }
        for (; i < bw; ++i) sad += (uint32_t)abs((int)s[i] - (int)f[i]);
    }
    return sad;
}

// The T search rows (STEP apart: same parity for a block sampled every other row, consecutive for a block sampled on
// every row) x 4 columns of every unit of a staged window band, for a 16-pixel-wide block of `rows` sampled rows held in
// src (wave-uniform); FULL: rows == NR. Folds each lane's best (SAD, y, x) into `best`.
// MODE 0: one window, one minimum (best[0]). The window was staged with one empty 16-byte piece in front of every
// (STEP T)-th row (stage_window's gap_shift): units STEP T rows apart start 4 banks apart instead of in the same one.
// MODE 1: the window is two regions of w_reg columns side by side (w_reg a multiple of 4, so a unit lies in one of them);
// each keeps its own minimum, with positions counted inside the region: best[0] the left one's, best[1] the right one's.
// MODE 2: two such rows of regions, h_reg rows each, one under the other in a window staged in one piece without the empty
// pieces (brows == 2 h_reg): the units are cut per row of regions, so none lies across the border; best[2], best[3] are
// the lower row's.
template <int T, bool FULL, int NR, int STEP, int MODE>
__device__ __forceinline__ void strip_units(const uint32_t *win, int pd, int n_xg, int brows, bool skip_even, const uint32_t (&src)[NR][4],
                                            int rows, int zero, int y_abs0, int x_abs0, int sa_w, u64 (&best)[4], int w_reg, int h_reg) {
    const int lane   = threadIdx.x & 63;
    const int hrows  = MODE == 2 ? h_reg : brows; // rows the units are cut from
    const int n_yg   = (hrows + STEP * T - 1) / (STEP * T);
    const int n_half = n_xg * n_yg * ((skip_even || STEP == 1) ? 1 : 2);
    const int n_unit = MODE == 2 ? 2 * n_half : n_half;
    const uint32_t xg_magic = div_magic((uint32_t)n_xg);
    for (int ub = 0; ub < n_unit; ub += HI_THREADS) {
        const int  u     = ub + lane;
        const bool valid = u < n_unit;
        const bool lower = MODE == 2 && u >= n_half;
        const int  uu    = lower ? u - n_half : u;
        const int  rest  = valid ? (int)magic_div((uint32_t)uu, (uint32_t)n_xg, xg_magic) : 0;
        const int  xg    = valid ? uu - rest * n_xg : 0;
        const int  par   = STEP == 1 ? 0 : (skip_even ? 1 : (rest & 1));
        const int  yg    = (skip_even || STEP == 1) ? rest : rest >> 1;
        const int  yrel  = yg * STEP * T + par; // first row, within the rows the units are cut from
        const int  n_pos = valid ? min(T, max(0, (hrows - yrel + STEP - 1) / STEP)) : 0;
        const int  yfirst = yrel + (lower ? h_reg : 0);
        // the four 8-byte operands of a step overlap by a dword; the odd pairs are read through a second pointer the
        // compiler cannot prove equal to the first, so every pair comes from LDS in its own aligned register pair
        const uint32_t *rr = win + yfirst * pd + (MODE == 2 ? 0 : 4 * yg) + xg; // (MODE < 2: yfirst / (STEP T) == yg)
        const uint32_t *rq = rr + 1 + zero;
        u64             acc[T];
#pragma unroll
        for (int t = 0; t < T; ++t) acc[t] = 0;
        // row loads run one step ahead of the arithmetic (sched_barrier keeps them there)
        uint32_t n01[2], n12[2], n23[2], n34[2];
        auto     fetch = [&](auto sn_c) { // the window row of step sn: STEP sn rows below the unit's first, past sn / T gaps
            constexpr int sn = decltype(sn_c)::value;
            if constexpr (MODE != 2 && sn > 0 && sn % T == 0) {
                rr += 4;
                rq += 4;
            }
            n01[0] = rr[0], n01[1] = rr[1], n23[0] = rr[2], n23[1] = rr[3];
            n12[0] = rq[0], n12[1] = rq[1], n34[0] = rq[2], n34[1] = rq[3];
            rr += STEP * pd;
            rq += STEP * pd;
        };
        fetch(std::integral_constant<int, 0>{});
        static_for<0, T + NR - 1>([&](auto s_c) {
            constexpr int s   = decltype(s_c)::value;
            const u64     p01 = ((u64)n01[1] << 32) | n01[0], p12 = ((u64)n12[1] << 32) | n12[0];
            const u64     p23 = ((u64)n23[1] << 32) | n23[0], p34 = ((u64)n34[1] << 32) | n34[0];
            if constexpr (s + 1 < T + NR - 1) fetch(std::integral_constant<int, s + 1>{});
            __builtin_amdgcn_sched_barrier(0);
            static_for<0, T>([&](auto t_c) {
                constexpr int t = decltype(t_c)::value;
                constexpr int m = s - t; // sampled source row of position t at this step
                if constexpr (m >= 0 && m < NR) {
                    if (FULL || m < rows) {
                        acc[t] = __builtin_amdgcn_qsad_pk_u16_u8(p01, src[m][0], acc[t]);
                        acc[t] = __builtin_amdgcn_qsad_pk_u16_u8(p12, src[m][1], acc[t]);
                        acc[t] = __builtin_amdgcn_qsad_pk_u16_u8(p23, src[m][2], acc[t]);
                        acc[t] = __builtin_amdgcn_qsad_pk_u16_u8(p34, src[m][3], acc[t]);
                    }
                }
            });
        });
        // per column: min over rows of SAD << 4 | t (SAD <= 16 * 16 * 255 < 65536: the packed sums never carried)
        uint32_t col[4] = {~0u, ~0u, ~0u, ~0u};
#pragma unroll
        for (int t = 0; t < T; ++t) {
            const uint32_t tk = t < n_pos ? (uint32_t)t : ~0u;
            const uint32_t lo = (uint32_t)acc[t], hi = (uint32_t)(acc[t] >> 32);
            col[0] = min(col[0], ((lo & 0xFFFFu) << 4) | tk);
            col[1] = min(col[1], ((lo >> 16) << 4) | tk);
            col[2] = min(col[2], ((hi & 0xFFFFu) << 4) | tk);
            col[3] = min(col[3], ((hi >> 16) << 4) | tk);
        }
        // raster order: row first, then column (positions counted inside the unit's region)
        u64        ubest  = ~0ull;
        const bool second = MODE >= 1 && x_abs0 + xg * 4 >= w_reg;
        const int  xo = second ? w_reg : 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int x = x_abs0 + xg * 4 + i;
            if (col[i] != ~0u && x < sa_w) {
                const int y = y_abs0 + yrel + STEP * (int)(col[i] & 15);
                const u64 k = ((u64)(col[i] >> 4) << 32) | ((uint32_t)y << 16) | (uint32_t)(x - xo);
                ubest       = k < ubest ? k : ubest;
            }
        }
        const int q = (second ? 1 : 0) + (lower ? 2 : 0);
#pragma unroll
        for (int j = 0; j < (MODE == 2 ? 4 : MODE + 1); ++j) best[j] = (q == j && ubest < best[j]) ? ubest : best[j];
    }
}

// Exhaustive search of the (bw x rows sampled rows, STEP = FS ? 1 : 2 plane rows apart) block of `cur` at (ox, oy) over
// the sa_w x sa_h window of `ref` whose first position is plane pixel (wx, wy). Leaves min (SAD << 32 | y << 16 | x) in
// key[0] (in every lane), ~0 when no position was evaluated (skip_even with a single row). Whole wave. skip_even needs
// FS == false.
// MODE 1 (bw == 16 only): the window is two regions of w_reg = sa_w / 2 columns side by side: key[0] the left region's
// minimum, key[1] the right one's, both with positions counted inside the region. MODE 2: two such rows of regions of
// h_reg = sa_h / 2 rows each, key[2], key[3] the lower one's; the caller has checked with l0_quad_fits that the window
// can be staged in one piece.
constexpr int kL0Under = 14; // rows under a band's last search row that a sub-sampled 16-row block reaches: 7 x 2
__device__ __forceinline__ bool l0_quad_fits(int sa_w, int sa_h) {
    const int n_xg = (sa_w + 3) >> 2, pd = (n_xg + 4 + 3) & ~3;
    return sa_w <= HI_XCHUNK && (sa_h + kL0Under + 16) * pd * 4 <= HI_WIN_BYTES;
}
template <bool FS, int MODE = 0>
__device__ __forceinline__ void l0_search(uint32_t *win, u64 (&key)[4], int zero, const DevPlane &cur, const DevPlane &ref, int ox, int oy, int bw,
                                          int rows, int wx, int wy, int sa_w, int sa_h, bool skip_even, int w_reg, int h_reg) {
    constexpr int NR = FS ? 16 : 8, STEP = FS ? 1 : 2, NKEY = MODE == 2 ? 4 : MODE + 1;
    const int lane = threadIdx.x & 63;
    u64       best[4] = {~0ull, ~0ull, ~0ull, ~0ull};
    key[0] = key[1] = key[2] = key[3] = ~0ull;
    if (sa_w <= 0 || sa_h <= 0 || rows <= 0) return;
    if (bw != 16) {
        for (int i = lane; i < sa_w * sa_h; i += HI_THREADS) {
            const int y = i / sa_w, x = i - y * sa_w;
            if (skip_even && (y & 1) == 0) continue;
            const u64 k = ((u64)slow_sad(cur, ref, ox, oy, wx + x, wy + y, bw, rows, STEP) << 32) | ((uint32_t)y << 16) | (uint32_t)x;
            best[0]     = k < best[0] ? k : best[0];
        }
    } else {
        // the block's sampled rows are the same for every lane: they come through the scalar (constant) path straight into
        // scalar registers, in flight together with the first window band's loads; nothing goes through LDS
        uint32_t src[NR][4];
        {
            typedef const uint32_t __attribute__((address_space(4))) ConstU32;
#pragma unroll
            for (int m = 0; m < NR; ++m) {
                ConstU32 *row = (ConstU32 *)(cur.pix + (ptrdiff_t)(oy + STEP * (m < rows ? m : 0)) * cur.stride + ox);
#pragma unroll
                for (int j = 0; j < 4; ++j) src[m][j] = row[j];
            }
        }
        // column chunks: a window of up to 96 positions in one piece; wider ones in chunks of 128 positions, whose 32
        // position quads x 2 row parities give each of the 64 lanes a unit per group of rows, whatever the rows per unit
        const int xchunk = sa_w > HI_XCHUNK ? 128 : HI_XCHUNK;
        for (int x0 = 0; x0 < sa_w; x0 += xchunk) {
            const int n_xg = (min(xchunk, sa_w - x0) + 3) >> 2;
            const int pd   = (n_xg + 4 + 3) & ~3; // dwords per staged row: the positions + 15 more pixels, in 16-byte pieces
            // rows the LDS budget holds: a band's search rows + the (NR - 1) STEP rows under the last one + the staggered
            // walk's overrun (rows that are addressed but never used)
            constexpr int under = (NR - 1) * STEP;
            static_assert(FS || under == kL0Under, "l0_quad_fits counts the rows under a band");
            // (a row costs up to 4 / (2 STEP) dwords more: the gap pieces, one per STEP T >= 2 STEP rows)
            const int     pdg   = pd + 2 / STEP;
            const int     cap   = (int)magic_div(HI_WIN_BYTES / 4, (uint32_t)pdg, div_magic((uint32_t)pdg));
            // The overrun (up to 16 rows) only exists under a band whose rows are not whole units, which only the window's
            // last band can be: the bands before it take all the rows the budget holds, in multiples of 16 (whole units for
            // every choice of rows per unit), the last one what fits with the overrun.
            const int     full = (cap - under) & ~15;
            int           last = cap - under - 16;
            last           = last >= 16 ? last & ~15 : max(last, 1);
            int band;
            for (int y0 = 0; y0 < sa_h; y0 += band) {
                const int left = sa_h - y0;
                band           = MODE == 2 ? sa_h /* (fits: l0_quad_fits) */ : (left <= last || full < 16 ? last : full);
                const int brows = min(band, left);
                // rows per unit (8, 4 or 2): the choice that costs the wave the fewest cycles - trips of 64 units, each
                // 16 QSADs per source row and search row (16 cycles apiece) and some 230 + 36 T cycles of bookkeeping
                int t = 2;
                {
                    const int npar = (skip_even || STEP == 1) ? 1 : 2;
                    const int cut = MODE == 2 ? h_reg : brows, sets = MODE == 2 ? 2 : 1;
                    int       best_cost = 0x7FFFFFFF;
#pragma unroll
                    for (int tt = 8; tt >= 2; tt >>= 1) {
                        const int n_yg = (cut + STEP * tt - 1) / (STEP * tt); // (a shift: STEP * tt is a constant power of two)
                        const int trips = (n_xg * n_yg * npar * sets + HI_THREADS - 1) / HI_THREADS;
                        const int cost = trips * (NR * 64 * tt + 230 + 36 * tt);
                        if (cost < best_cost) {
                            best_cost = cost;
                            t         = tt;
                        }
                    }
                }
                wave_sync();
                stage_window<5>(win, ref.pix + (ptrdiff_t)(wy + y0) * ref.stride + wx + x0, ref.stride, pd, brows + under,
                                MODE == 2 ? 31 : (STEP == 2 ? 1 : 0) + (t == 8 ? 3 : (t == 4 ? 2 : 1)));
                wave_sync();
                if (rows == NR) {
                    if (t == 8) strip_units<8, true, NR, STEP, MODE>(win, pd, n_xg, brows, skip_even, src, rows, zero, y0, x0, sa_w, best, w_reg, h_reg);
                    else if (t == 4) strip_units<4, true, NR, STEP, MODE>(win, pd, n_xg, brows, skip_even, src, rows, zero, y0, x0, sa_w, best, w_reg, h_reg);
                    else strip_units<2, true, NR, STEP, MODE>(win, pd, n_xg, brows, skip_even, src, rows, zero, y0, x0, sa_w, best, w_reg, h_reg);
                } else { // blocks of the picture's bottom edge
                    if (t == 8) strip_units<8, false, NR, STEP, MODE>(win, pd, n_xg, brows, skip_even, src, rows, zero, y0, x0, sa_w, best, w_reg, h_reg);
                    else if (t == 4) strip_units<4, false, NR, STEP, MODE>(win, pd, n_xg, brows, skip_even, src, rows, zero, y0, x0, sa_w, best, w_reg, h_reg);
                    else strip_units<2, false, NR, STEP, MODE>(win, pd, n_xg, brows, skip_even, src, rows, zero, y0, x0, sa_w, best, w_reg, h_reg);
                }
            }
        }
    }
    wave_sync(); // the window may be rewritten by this wave's next search
#pragma unroll
    for (int j = 0; j < NKEY; ++j) key[j] = wave_min_u64(best[j]);
}

template <bool FS, class P>
__device__ __forceinline__ void l0_item(const P &p, uint32_t *win, int b64x, int b64y, int block_w, int block_h, int l, int r,
                                        int sub, uint32_t zz, HmeRefOut *rec, const HmeRefOut *first, const HmeRefOut *same0, bool all_regions) {
    const auto    &k    = p.ctrl;
    const int      lane = threadIdx.x & 63;
    const DevPlane cur  = load_plane(p.cur.lvl[0]);
    const DevPlane ref  = load_plane(p.ref[l][r].lvl[0]);
    const int             ox = b64x >> 2, oy = b64y >> 2;
    const int             bw = block_w >> 2, bh = block_h >> 2;
    const int             rows = FS ? bh : bh >> 1;
    u64                   lbest[4]; // the minima of a search (up to four regions)
    // the search this item runs: window (origin relative to the block, size), row skipping, where the result goes
    int16_t   w = 0, h = 0, org_x = 0, org_y = 0;
    bool      skip = false;
    int16_t  *res_x, *res_y;
    uint32_t *res_sad;
    int       ops_k;
    if (sub < 2) {
        // pre-HME (prehme_b64 :1773-1823); check_prehme_early_exit :1746-1770 leaves a zero result
        if (!k.prehme_enable || (k.me_early_exit_th && zz < k.me_early_exit_th)) return;
        if (k.prehme_l1_early_exit && l == 1) {
            // a list-1 search is skipped when the list-0 search of the same index found a good or a short vector: it
            // mirrors that result (:1759-1768)
            const int16_t  x0 = same0->pre_x[sub], y0 = same0->pre_y[sub];
            const uint32_t s0 = same0->pre_sad[sub];
            if (s0 < 32 * 32 || (abs(x0) < 16 && abs(y0) < 16)) {
                if (lane == 0) {
                    rec->pre_x[sub]   = (int16_t)-x0;
                    rec->pre_y[sub]   = (int16_t)-y0;
                    rec->pre_sad[sub] = s0;
                }
                return;
            }
        }
        const auto            &cfg    = k.prehme_sa_cfg[sub];
        const int              factor = scaled_dist(p.dist[l][r]);
        w     = (int16_t)(uint16_t)min((uint32_t)cfg.sa_min.width * factor, (uint32_t)cfg.sa_max.width);
        h     = (int16_t)(uint16_t)min((uint32_t)cfg.sa_min.height * factor, (uint32_t)cfg.sa_max.height);
        org_x = (int16_t)-(int16_t)(w >> 1);
        org_y = (int16_t)-(int16_t)(h >> 1);
        clip_axis(ox, org_x, w, -(ref.pad - 1), ref.width);
        clip_axis(oy, org_y, h, -(ref.pad - 1), ref.height);
        // svt_sad_loop_kernel_c skips even rows only for 16-wide blocks of height <= 16
        skip    = k.prehme_skip_search_line && bw == 16 && rows <= 16;
        res_x   = &rec->pre_x[sub];
        res_y   = &rec->pre_y[sub];
        res_sad = &rec->pre_sad[sub];
        ops_k   = 1;
    } else {
        // level 0 (:1899-2004): a region per item; the early exit leaves zeros
        const int q = sub - 2, nw = k.num_hme_sa_w, nh = k.num_hme_sa_h;
        if (!k.enable_hme_flag || !k.enable_hme_level0_flag || q >= nw * nh) return;
        if (k.me_early_exit_th && zz < (k.me_early_exit_th >> 2)) return;
        int16_t   sa_w = 0, sa_h = 0;
        l0_area(p, *first, l, r, sa_w, sa_h);
        // the window of region q_
        auto region = [&](int q_, int16_t &w_, int16_t &h_, int16_t &x_, int16_t &y_) {
            const int sw = q_ % nw, sh = q_ / nw;
            w_ = (int16_t)((sa_w + 7) & ~0x07);
            h_ = sa_h;
            x_ = (int16_t)(-(int16_t)((w_ * nw) >> 1) + (int16_t)(w_ * sw));
            y_ = (int16_t)(-(int16_t)((h_ * nh) >> 1) + (int16_t)(h_ * sh));
            clip_axis(ox, x_, w_, -(ref.pad - 1), ref.width);
            w_ = (int16_t)(w_ < 8 ? w_ : (w_ & ~0x07));
            clip_axis(oy, y_, h_, -(ref.pad - 1), ref.height);
        };
        if (all_regions) {
            // This wave runs every region of the reference. Where the 2 x 2 regions tile one rectangle (no region was
            // clipped at the picture's edge) they are searched as one window - all four where it can be staged in one
            // piece, else the two of a row -, each region keeping its own minimum: the item's fixed part (geometry, the
            // block's rows, staging) is paid once and the lanes are two or four times as well filled, which is what counts
            // for the small windows of the fast presets.
            int16_t rw[4] = {0, 0, 0, 0}, rh[4] = {0, 0, 0, 0}, rx[4] = {0, 0, 0, 0}, ry[4] = {0, 0, 0, 0};
            for (int q_ = 0; q_ < nw * nh; ++q_) region(q_, rw[q_], rh[q_], rx[q_], ry[q_]);
            const bool tiled = nw == 2 && nh == 2 && bw == 16 && rw[0] >= 8 && rh[0] >= 1 && rw[1] == rw[0] && rw[2] == rw[0] && rw[3] == rw[0] &&
                rh[1] == rh[0] && rh[2] == rh[0] && rh[3] == rh[0] && rx[1] == rx[0] + rw[0] && rx[2] == rx[0] && rx[3] == rx[1] &&
                ry[1] == ry[0] && ry[2] == ry[0] + rh[0] && ry[3] == ry[2];
            unsigned long long ops = 0;
            for (int q_ = 0; q_ < nw * nh; ++q_) ops += (unsigned long long)max((int)rh[q_], 0) * max((int)rw[q_], 0) * rows * bw;
            if (lane == 0) count_ops(p, 2, ops);
            if (!FS && tiled && l0_quad_fits(2 * rw[0], 2 * rh[0])) {
                l0_search<FS, 2>(win, lbest, p.zero, cur, ref, ox, oy, bw, rows, ox + rx[0], oy + ry[0], 2 * rw[0], 2 * rh[0], false, rw[0], rh[0]);
                if (lane == 0)
                    for (int sh = 0; sh < 2; ++sh) {
                        decode_key(lbest[2 * sh], rx[0], ry[2 * sh], 4, rec->raw_x[sh], rec->raw_y[sh], rec->raw_sad[sh], !FS);
                        decode_key(lbest[2 * sh + 1], rx[1], ry[2 * sh], 4, rec->raw_x[2 + sh], rec->raw_y[2 + sh], rec->raw_sad[2 + sh], !FS);
                    }
            } else if (tiled) {
                for (int sh = 0; sh < 2; ++sh) {
                    l0_search<FS, 1>(win, lbest, p.zero, cur, ref, ox, oy, bw, rows, ox + rx[0], oy + ry[2 * sh], 2 * rw[0], rh[0], false, rw[0], 0);
                    if (lane == 0) {
                        decode_key(lbest[0], rx[0], ry[2 * sh], 4, rec->raw_x[sh], rec->raw_y[sh], rec->raw_sad[sh], !FS);
                        decode_key(lbest[1], rx[1], ry[2 * sh], 4, rec->raw_x[2 + sh], rec->raw_y[2 + sh], rec->raw_sad[2 + sh], !FS);
                    }
                }
            } else {
                for (int q_ = 0; q_ < nw * nh; ++q_) {
                    const int i   = (q_ % nw) * 2 + q_ / nw;
                    l0_search<FS>(win, lbest, p.zero, cur, ref, ox, oy, bw, rows, ox + rx[q_], oy + ry[q_], rw[q_], rh[q_], false, 0, 0);
                    if (lane == 0) decode_key(lbest[0], rx[q_], ry[q_], 4, rec->raw_x[i], rec->raw_y[i], rec->raw_sad[i], !FS);
                }
            }
            return;
        }
        region(q, w, h, org_x, org_y);
        const int i = (q % nw) * 2 + q / nw;
        res_x   = &rec->raw_x[i];
        res_y   = &rec->raw_y[i];
        res_sad = &rec->raw_sad[i];
        ops_k   = 2;
    }
    if (lane == 0) count_ops(p, ops_k, (unsigned long long)max(skip ? h >> 1 : (int)h, 0) * max((int)w, 0) * rows * bw);
    l0_search<FS>(win, lbest, p.zero, cur, ref, ox, oy, bw, rows, ox + org_x, oy + org_y, w, h, skip, 0, 0);
    if (lane == 0) decode_key(lbest[0], org_x, org_y, 4, *res_x, *res_y, *res_sad, !FS);
}

// ---------------------------------------------------------------------------------------------------------------------
// Dwords between the staged rows of a source block. 20 for the sub-sampled searches: quick_search gives neighbouring lanes
// neighbouring block rows, and 20 k mod 32 (k = 0..7) are eight different multiples of 4, so the 16-byte row reads of the
// eight lanes fall into different banks (with 16 they all start in bank 0).
template <bool FS>
constexpr int kSrcPitch = FS ? 16 : 20;
// The LDS of a level-1/2 work item (l12_region), in dwords. The levels use it one after the other: [the sampled rows of the
// quarter-resolution 32x32 block][level 1's window], then [those of the full-resolution 64x64 block][level 2's window, of
// kL12WinBytes], plus 64 dwords of slack after a window (pieces staged and rows read past its end). These kernels are chains
// of memory round trips: their rate is the number of work items a SIMD holds, which this size sets.
template <bool FS>
constexpr int kL12Src1 = (FS ? 32 : 16) * kSrcPitch<FS>;
template <bool FS>
constexpr int kL12Src2 = (FS ? 64 : 32) * kSrcPitch<FS>;
template <bool FS>
constexpr int kL12WinBytes = FS ? HI_WIN2_BYTES_FS : 5376; // (8 KB per work item without FULL_SAD_SEARCH: five per SIMD)
template <bool FS>
constexpr int kL12Lds = kL12Src2<FS> + kL12WinBytes<FS> / 4 + 64;
template <bool FS>
constexpr int kL12Blk2 = ((FS ? 64 : 32) + 15) / 16; // 16-byte pieces per lane of the 64x64 block's sampled rows

// Exhaustive search of a small window with a large block whose sampled rows are staged at `src` (16 dwords per row):
// position quads x block-row chunks over the lanes. bw: block width in pixels (multiple of 4 at these levels),
// rows: sampled rows, 2 plane rows apart.
// Sampled rows (STEP plane rows apart, at most 4 K of them) of the bw x rows block at (ox, oy) -> dst, PITCH dwords per
// row. Issues its loads and stores without waiting: the caller's next barrier orders them.
// (in two halves, for a caller that has the block's place in LDS only later: the pieces wait in registers)
template <int K, int STEP>
__device__ __forceinline__ void fetch_block(uint4 (&v)[(4 * K + 15) / 16], const DevPlane &cur, int ox, int oy, int bw, int rows) {
    const int lane = threadIdx.x & 63, c = lane & 3, pr = (bw + 15) >> 4; // one wave fetches
#pragma unroll
    for (int k = 0; k < (4 * K + 15) / 16; ++k) {
        const int r = (lane >> 2) + 16 * k;
        v[k]        = make_uint4(0, 0, 0, 0);
        if (c < pr && r < rows) __builtin_memcpy(&v[k], cur.pix + (ptrdiff_t)(oy + STEP * r) * cur.stride + ox + 16 * c, 16);
    }
}
template <int K, int PITCH>
__device__ __forceinline__ void put_block(uint32_t *dst, const uint4 (&v)[(4 * K + 15) / 16], int bw, int rows) {
    static_assert(PITCH % 4 == 0, "rows start on 16-byte boundaries");
    const int lane = threadIdx.x & 63, c = lane & 3, pr = (bw + 15) >> 4;
#pragma unroll
    for (int k = 0; k < (4 * K + 15) / 16; ++k) {
        const int r = (lane >> 2) + 16 * k;
        if (c < pr && r < rows) *reinterpret_cast<uint4 *>(dst + r * PITCH + 4 * c) = v[k];
    }
}
template <int K = 8, int STEP = 2, int PITCH = 16>
__device__ __forceinline__ void stage_block(uint32_t *dst, const DevPlane &cur, int ox, int oy, int bw, int rows) {
    // 16-byte pieces, a row's in neighbouring lanes: lane = (row of the trip: lane / 4, piece: lane % 4), 16 rows per trip
    // (PITCH is a multiple of 4 dwords and dst 16-byte aligned; a piece may reach past a narrower block's right edge,
    // into columns the searches do not read)
    constexpr int KT = (4 * K + 15) / 16; // trips: up to 4 K rows, 16 per trip
    uint4         v[KT];
    fetch_block<K, STEP>(v, cur, ox, oy, bw, rows);
    put_block<K, PITCH>(dst, v, bw, rows);
}

// One sampled block row of BWD dwords against 4 adjacent positions of a staged window row.
template <int BWD>
__device__ __forceinline__ u64 row_qsad(const uint32_t *rr, const uint32_t *ss, u64 a16) {
    uint32_t rv[BWD + 1], sv[BWD];
#pragma unroll
    for (int d = 0; d <= BWD; ++d) rv[d] = rr[d];
#pragma unroll
    for (int d = 0; d < BWD; ++d) sv[d] = ss[d];
#pragma unroll
    for (int d = 0; d < BWD; ++d) a16 = __builtin_amdgcn_qsad_pk_u16_u8(((u64)rv[d + 1] << 32) | rv[d], sv[d], a16);
    return a16;
}

// WINB: the bytes at `win`.
template <bool FS, int WINB = FS ? HI_WIN2_BYTES_FS : HI_WIN2_BYTES>
__device__ u64 small_search(const uint32_t *src, uint32_t *win, const DevPlane &ref, int bw, int rows, int wx, int wy, int sa_w, int sa_h) {
    constexpr int STEP = FS ? 1 : 2; // plane rows between sampled block rows
    constexpr int NT   = 64;
    const int lane = threadIdx.x & 63;
    u64       best = ~0ull;
    if (sa_w <= 0 || sa_h <= 0 || rows <= 0) return best;
    const int bwd = bw >> 2;
    const int span       = (rows - 1) * STEP + 1;
    const int chunk_log2 = rows >= 16 ? 3 : (rows >= 4 ? 2 : 0), chunks = 1 << chunk_log2;
    const int rpc        = (rows + chunks - 1) >> chunk_log2; // sampled block rows per lane
    // positions per staged column chunk: at most 64, and no more than leave room for one search row's `span` window rows
    const int xc = 4 * max(min(16, ((WINB / 4 / span) & ~3) - bwd), 1);
    for (int x0 = 0; x0 < sa_w; x0 += xc) {
        const int n_xg = (min(xc, sa_w - x0) + 3) >> 2, pd = (n_xg + bwd + 3) & ~3;
        const int band = max(WINB / (4 * pd) - (span - 1), 1);
        for (int y0 = 0; y0 < sa_h; y0 += band) {
            const int brows = min(band, sa_h - y0);
            wave_sync();
            stage_window<FS ? 2 : 6, NT>(win, ref.pix + (ptrdiff_t)(wy + y0) * ref.stride + wx + x0, ref.stride, pd, brows + span - 1);
            wave_sync();
            const int total = (n_xg * brows) << chunk_log2;
            for (int base = 0; base < total; base += NT) {
                const int  t     = base + lane;
                const bool valid = t < total;
                const int  chunk = t & (chunks - 1);
                const int  q     = valid ? t >> chunk_log2 : 0;
                const int  yl = q / n_xg, xg = q - yl * n_xg;
                uint32_t   a32[4] = {0, 0, 0, 0};
                if (valid) {
                    const int j0 = chunk * rpc, j1 = min(rows, j0 + rpc);
                    u64       a16 = 0;
                    int       px  = 0;
                    for (int j = j0; j < j1; ++j) {
                        const uint32_t *rr = win + (yl + STEP * j) * pd + xg;
                        const uint32_t *ss = &src[j * kSrcPitch<FS>];
                        if (px + bw > 256) { // keep the packed 16-bit sums below 65536
#pragma unroll
                            for (int i = 0; i < 4; ++i) a32[i] += (uint32_t)(a16 >> (16 * i)) & 0xFFFFu;
                            a16 = 0;
                            px  = 0;
                        }
                        if (FS && bwd == 16) a16 = row_qsad<16>(rr, ss, a16);
                        else if (FS && bwd == 8) a16 = row_qsad<8>(rr, ss, a16);
                        else {
                            uint32_t r0 = rr[0];
                            for (int d = 0; d < bwd; ++d) {
                                const uint32_t r1 = rr[d + 1];
                                a16 = __builtin_amdgcn_qsad_pk_u16_u8(((u64)r1 << 32) | r0, ss[d], a16);
                                r0  = r1;
                            }
                        }
                        px += bw;
                    }
#pragma unroll
                    for (int i = 0; i < 4; ++i) a32[i] += (uint32_t)(a16 >> (16 * i)) & 0xFFFFu;
                }
                for (int m = 1; m < chunks; m <<= 1) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) a32[i] += __shfl_xor(a32[i], m);
                }
                if (valid && chunk == 0) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int x = x0 + xg * 4 + i;
                        if (x < sa_w) {
                            const u64 kk = ((u64)a32[i] << 32) | ((uint32_t)(y0 + yl) << 16) | (uint32_t)x;
                            best         = kk < best ? kk : best;
                        }
                    }
                }
            }
        }
    }
    wave_sync();
    return wave_min_u64(best);
}

// FULL_SAD_SEARCH levels 1 / 2 in the temporal filter's shape: a window of at most 16 x 16 positions, a full-width block
// of NSTRIP 16-pixel column strips and up to NR rows (every row sampled), staged at `src`. Lane = (column strip) x (4
// adjacent positions) x (T consecutive search rows): the wave covers the whole window at once. The T rows are walked
// together, staggered by one block row, so each window row is read from LDS once per lane (four overlapping 8-byte
// operands) and each block row once (one 16-byte read): 4 T v_qsad_pk_u16_u8 per step. The packed 16-bit sums are
// folded into 32-bit sums every 16 block rows (16 x 16 x 255 < 65536); the strips' sums meet through DPP lane exchanges.
template <int NSTRIP, int NR, int T>
__device__ u64 strip_search_fs(const uint32_t *src, uint32_t *win, int zero, const DevPlane &ref, int rows, int wx, int wy, int sa_w, int sa_h) {
    constexpr int PD = 4 + 4 * NSTRIP; // dwords per staged row: 16 positions + the block width - 1 more pixels, rounded up
    static_assert(64 / NSTRIP / 4 * T >= 16, "the wave covers 16 search rows");
    const int lane = threadIdx.x & 63;
    const int n_xg = (sa_w + 3) >> 2, n_yg = (sa_h + T - 1) / T;
    wave_sync();
    stage_window<2>(win, ref.pix + (ptrdiff_t)wy * ref.stride + wx, ref.stride, PD, n_yg * T + NR - 1);
    wave_sync();
    const int  strip = lane % NSTRIP, unit = lane / NSTRIP, xg = unit & 3, yg = unit >> 2;
    const bool valid = xg < n_xg && yg < n_yg;
    const int  yfirst = valid ? yg * T : 0;
    // the four 8-byte operands of a step overlap by a dword; the odd pairs are read through a second pointer the
    // compiler cannot prove equal to the first, so every pair comes from LDS in its own aligned register pair
    const uint32_t *rr = win + yfirst * PD + (valid ? xg : 0) + strip * 4;
    const uint32_t *rq = rr + 1 + zero;
    const uint4    *sp = reinterpret_cast<const uint4 *>(src) + strip; // block row m: sp[4 m]
    u64             acc[T];
    uint32_t        a32[T][4];
#pragma unroll
    for (int t = 0; t < T; ++t) {
        acc[t] = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) a32[t][i] = 0;
    }
    uint32_t n01[2], n12[2], n23[2], n34[2];
    uint4    srow[T]; // block rows s, s - 1, .., s - T + 1 of the current step
    auto     fetch = [&](int s) {
        n01[0] = rr[0], n01[1] = rr[1], n23[0] = rr[2], n23[1] = rr[3];
        n12[0] = rq[0], n12[1] = rq[1], n34[0] = rq[2], n34[1] = rq[3];
        rr += PD;
        rq += PD;
        (void)s;
    };
    fetch(0);
    static_for<0, NR + T - 1>([&](auto s_c) {
        constexpr int s   = decltype(s_c)::value;
        const u64     p01 = ((u64)n01[1] << 32) | n01[0], p12 = ((u64)n12[1] << 32) | n12[0];
        const u64     p23 = ((u64)n23[1] << 32) | n23[0], p34 = ((u64)n34[1] << 32) | n34[0];
        // rotate the block rows: srow[t] = row s - t
#pragma unroll
        for (int t = T - 1; t > 0; --t) srow[t] = srow[t - 1];
        if constexpr (s < NR) srow[0] = sp[4 * s];
        if constexpr (s + 1 < NR + T - 1) fetch(s + 1);
        __builtin_amdgcn_sched_barrier(0);
        static_for<0, T>([&](auto t_c) {
            constexpr int t = decltype(t_c)::value;
            constexpr int m = s - t; // block row of position t at this step
            if constexpr (m >= 0 && m < NR) {
                if (m < rows) {
                    acc[t] = __builtin_amdgcn_qsad_pk_u16_u8(p01, srow[t].x, acc[t]);
                    acc[t] = __builtin_amdgcn_qsad_pk_u16_u8(p12, srow[t].y, acc[t]);
                    acc[t] = __builtin_amdgcn_qsad_pk_u16_u8(p23, srow[t].z, acc[t]);
                    acc[t] = __builtin_amdgcn_qsad_pk_u16_u8(p34, srow[t].w, acc[t]);
                }
                if constexpr ((m & 15) == 15 || m == NR - 1) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) a32[t][i] += (uint32_t)(acc[t] >> (16 * i)) & 0xFFFFu;
                    acc[t] = 0;
                }
            }
        });
    });
    // the strips of a position: adjacent lanes. A lane meets its positions in raster order (rows, then columns), so a
    // strictly smaller sum is all that replaces its best one.
    uint32_t bsad = ~0u, bpos = ~0u;
#pragma unroll
    for (int t = 0; t < T; ++t) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            uint32_t v = a32[t][i];
            v += quad_xor1(v);
            if constexpr (NSTRIP == 4) v += quad_xor2(v);
            const int x = xg * 4 + i, y = yfirst + t;
            if (valid && strip == 0 && x < sa_w && y < sa_h && v < bsad) {
                bsad = v;
                bpos = ((uint32_t)y << 16) | (uint32_t)x;
            }
        }
    }
    const u64 best = ((u64)bsad << 32) | bpos;
    wave_sync();
    return wave_min_u64(best);
}

template <int BWD, int MAXH>
struct QuickGeo {
    static constexpr int PD  = BWD + 2; // dwords per staged row: 8 positions + the block width - 1 more pixels, rounded up
    static constexpr int RPC = BWD / 4; // sampled block rows per lane (rows <= 8 RPC); RPC x 4 BWD x 255 fits the 16-bit sums
    static constexpr int PR  = (PD + 3) / 4; // 16-byte pieces per staged row; the last one carries PD - 4 (PR - 1) = 2 dwords
    static constexpr int NB  = ((MAXH + 2 * (8 * RPC - 1)) * PR + 63) / 64; // pieces per lane
    static constexpr int WIN = ((MAXH + 2 * (8 * RPC - 1)) * PD + 3) & ~3; // dwords of a staged window (16-byte multiple)
    static_assert(PD - 4 * (PR - 1) == 2, "the last piece of a row is half a piece");
    // piece index -> row: i / PR as a 24-bit multiplication and a shift, exact for every piece index of a window
    static constexpr uint32_t ROW_MAGIC = (2048 + PR - 1) / PR;
    static_assert((uint32_t)(64 * NB) * (ROW_MAGIC * PR - 2048) < 2048, "i * ROW_MAGIC >> 11 == i / PR for i < 64 NB");
    // (the pieces' addresses as 32-bit offsets from the window's first byte through 24-bit multiplications: full-rate
    // instructions where a division by a constant, a 64-bit multiply-add and a 32-bit multiplication run at a quarter)
    static __device__ __forceinline__ void piece(int i, int stride, uint32_t &from, int &at) {
        const uint32_t r = __umul24((uint32_t)i, ROW_MAGIC) >> 11, c = (uint32_t)i - __umul24(r, (uint32_t)PR);
        from = __umul24(r, (uint32_t)stride) + 16u * c;
        at   = (int)(__umul24(r, (uint32_t)PD) + 4u * c) | (c == PR - 1 ? (int)0x80000000 : 0);
    }
};

// The window of NQ searches at once: all loads of all windows are issued before the first store, so the NQ windows cost
// one memory round trip. Window q: (sa_h[q] + 2 (rows - 1)) rows from (wx[q], wy[q]) of `ref`, staged at win + q * WIN.
// piece i = lane + 64 k of a window, 16 bytes in one load (a third of the load instructions of dword loads, which is what
// these kernels wait for); indices past the end repeat the last piece (same value, same slot). The LDS rows stay PD dwords
// apart (the walk's bank pattern), so a row's last piece stores its first 8 bytes only.
template <int BWD, int MAXH, int NQ>
__device__ __forceinline__ void quick_stage(uint32_t *win, const DevPlane &ref, int rows, const int (&wx)[NQ], const int (&wy)[NQ],
                                            const int (&sa_h)[NQ]) {
    using G = QuickGeo<BWD, MAXH>;
    const int lane = threadIdx.x & 63;
    uint4     v[NQ][G::NB];
    int       at[NQ][G::NB];
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const int      n = (sa_h[q] + 2 * (rows - 1)) * G::PR;
        const uint8_t *g = ref.pix + (ptrdiff_t)wy[q] * ref.stride + wx[q];
#pragma unroll
        for (int k = 0; k < G::NB; ++k) {
            uint32_t from;
            G::piece(min(lane + 64 * k, n - 1), ref.stride, from, at[q][k]);
            __builtin_memcpy(&v[q][k], g + from, 16);
            at[q][k] += q * G::WIN; // (the flag bit stays: the sum of the low bits is below 2^31)
        }
    }
#pragma unroll
    for (int q = 0; q < NQ; ++q)
#pragma unroll
        for (int k = 0; k < G::NB; ++k) {
            uint2 *dst = reinterpret_cast<uint2 *>(win + (at[q][k] & 0x7FFFFFFF)); // (PD and WIN are even: 8-byte aligned)
            dst[0]     = make_uint2(v[q][k].x, v[q][k].y);
            if (at[q][k] >= 0) dst[1] = make_uint2(v[q][k].z, v[q][k].w);
        }
}

// small_search for the shapes the presets use at these levels - 8 x (1..MAXH) positions (3 rows in the pre-analysis ME, 8 in
// the first pass), full-width block of BWD dwords (8: the 32-pixel quarter-resolution block, 16: the 64-pixel
// full-resolution block) - over a window quick_stage left at `win`, with everything unrolled: lane = (position quad: 2 per
// row x 4 rows) x (8 chunks of BWD / 4 sampled block rows: rows chunk, chunk + 8, .., so that the eight lanes of a position
// quad read eight neighbouring rows of the block and of the window at a time - different banks, see kSrcPitch), 4 search
// rows per trip.
template <int BWD, int MAXH>
__device__ __forceinline__ u64 quick_scan(const uint32_t *src, const uint32_t *win, int rows, int sa_h) {
    using G = QuickGeo<BWD, MAXH>;
    constexpr int PD = G::PD, RPC = G::RPC;
    const int     lane = threadIdx.x & 63;
    const int     chunk = lane & 7, q = lane >> 3, xg = q & 1;
    uint32_t      best = ~0u;
    static_assert(MAXH <= 8 && BWD <= 16, "the 32-bit keys below");
#pragma unroll
    for (int y0 = 0; y0 < MAXH; y0 += 4) {
        if (y0 >= sa_h) break; // wave-uniform
        const int  yl     = y0 + (q >> 1);
        const bool active = yl < sa_h;
        u64        a16 = 0;
        if (active) {
#pragma unroll
            for (int jj = 0; jj < RPC; ++jj) {
                const int j = jj * 8 + chunk;
                if (j < rows) {
                    const uint32_t *rr = win + (yl + 2 * j) * PD + xg;
                    const uint32_t *ss = src + j * kSrcPitch<false>;
                    uint32_t        rv[BWD + 1], sv[BWD];
#pragma unroll
                    for (int d = 0; d <= BWD; ++d) rv[d] = rr[d];
#pragma unroll
                    for (int d = 0; d < BWD; ++d) sv[d] = ss[d];
#pragma unroll
                    for (int d = 0; d < BWD; ++d) a16 = __builtin_amdgcn_qsad_pk_u16_u8(((u64)rv[d + 1] << 32) | rv[d], sv[d], a16);
                }
            }
        }
        // the eight chunks' sums meet in every lane of the position quad (the last step through row_half_mirror: lane i
        // of a half row and lane 7 - i lie in different quads of lanes, which both hold their quad's sum by then)
        uint32_t a32[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            a32[i] = (uint32_t)(a16 >> (16 * i)) & 0xFFFFu;
            a32[i] += quad_xor1(a32[i]);
            a32[i] += quad_xor2(a32[i]);
            a32[i] += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)a32[i], 0x141, 0xF, 0xF, false);
        }
        // 32-bit keys SAD << 6 | y << 3 | x (SAD <= 32 rows x 64 pixels x 255 < 2^20, y and x < 8): they order like the 64-bit ones
        if (active && chunk == 0) {
#pragma unroll
            for (int i = 0; i < 4; ++i) best = min(best, (a32[i] << 6) | ((uint32_t)yl << 3) | (uint32_t)(xg * 4 + i));
        }
    }
    best = wave_min_u32(best);
    if (best == ~0u) return ~0ull;
    return ((u64)(best >> 6) << 32) | (((best >> 3) & 7u) << 16) | (best & 7u);
}


// One such search by itself (a work item per region): the same staging and scan in one piece.
template <int BWD, int MAXH>
__device__ u64 quick_search(const uint32_t *src, uint32_t *win, const DevPlane &ref, int rows, int wx, int wy, int sa_h) {
    const int wxs[1] = {wx}, wys[1] = {wy}, hs[1] = {sa_h};
    wave_sync();
    quick_stage<BWD, MAXH, 1>(win, ref, rows, wxs, wys, hs);
    wave_sync();
    const u64 key = quick_scan<BWD, MAXH>(src, win, rows, sa_h);
    wave_sync();
    return key;
}

// Levels 1 and 2 of region q of reference (l, r), whole wave: first the level-0 result of the region, which gives way
// to the better pre-HME result when it is the worst region (:1967-1999), then level 1 around the level-0 vector and
// level 2 around the level-1 vector. `rec`: the reference's record with zz_sad, pre_* and raw_* complete (global or LDS);
// lds: kL12Lds dwords with the quarter-resolution block's sampled rows staged at its start; blk2: the full-resolution
// block's, fetched by the caller (fetch_l12_blocks), which move into `lds` when level 1 is done with it.
template <bool FS, class P>
__device__ __forceinline__ void l12_region(const P &p, uint32_t *lds, const uint4 (&blk2)[kL12Blk2<FS>], int b64x, int b64y, int block_w,
                                           int block_h, int l, int r, int q, HmeRefOut *rec) {
    const auto &k    = p.ctrl;
    const int   lane = threadIdx.x & 63;
    const int   nw = k.num_hme_sa_w, nh = k.num_hme_sa_h;
    const int             sw = q % nw, sh = q / nw, i = sw * 2 + sh;
    const bool            early = k.me_early_exit_th && rec->zz_sad < (k.me_early_exit_th >> 2);

    // level-0 result of this region: the worst region (strict '>', sr_h outer / sr_w inner) gives way to the better
    // pre-HME result (:1967-1999)
    int16_t  cx = 0, cy = 0; // vector of the previous level (zero when that level did not run)
    uint32_t csad = 0;
    if (k.enable_hme_level0_flag && !early) {
        cx   = rec->raw_x[i];
        cy   = rec->raw_y[i];
        csad = rec->raw_sad[i];
        if (k.prehme_enable) {
            uint32_t max_sad = 0;
            int      mi      = 0;
            for (int h_ = 0; h_ < nh; ++h_)
                for (int w_ = 0; w_ < nw; ++w_)
                    if (rec->raw_sad[w_ * 2 + h_] > max_sad) {
                        max_sad = rec->raw_sad[w_ * 2 + h_];
                        mi      = w_ * 2 + h_;
                    }
            const int s = rec->pre_sad[0] <= rec->pre_sad[1] ? 0 : 1;
            if (mi == i && rec->pre_sad[s] < csad) {
                csad = rec->pre_sad[s];
                cx   = rec->pre_x[s];
                cy   = rec->pre_y[s];
            }
        }
        if (lane == 0) {
            rec->lx[0][i]   = cx;
            rec->ly[0][i]   = cy;
            rec->lsad[0][i] = csad;
        }
    }
    for (int level = 1; level <= 2; ++level) {
        // a level that does not run, or level 1 on the early exit (:2032-2040; level 2 has none), leaves zeros
        if ((level == 1 ? !k.enable_hme_level1_flag : !k.enable_hme_level2_flag) || (level == 1 && early)) {
            cx = cy = 0;
            continue;
        }
        const DevPlane  ref   = load_plane(p.ref[l][r].lvl[level]);
        const int       shift = 2 - level;
        const int       ox = b64x >> shift, oy = b64y >> shift;
        const int       bw = block_w >> shift, bh = block_h >> shift;
        const int       rows = FS ? bh : bh >> 1;
        const int       pad  = level == 2 ? 63 : ref.pad - 1;
        const int       sa_width = level == 1 ? k.hme_l1_sa.width : k.hme_l2_sa.width, sa_height = level == 1 ? k.hme_l1_sa.height : k.hme_l2_sa.height;
        int16_t         w = (int16_t)(((int16_t)sa_width + 7) & ~0x07), h = (int16_t)sa_height;
        if (level == 1) {
            cx = (int16_t)(cx >> 1);
            cy = (int16_t)(cy >> 1);
        }
        int16_t org_x = (int16_t)(-(w >> 1) + cx), org_y = (int16_t)(-(h >> 1) + cy);
        clip_axis(ox, org_x, w, -pad, ref.width);
        w = (int16_t)(w < 8 ? w : (w & ~0x07));
        clip_axis(oy, org_y, h, -pad, ref.height);
        if (lane == 0) count_ops(p, 2 + level, (unsigned long long)max((int)h, 0) * max((int)w, 0) * rows * bw);
        const uint32_t *src = lds;
        uint32_t       *win = lds + (level == 1 ? kL12Src1<FS> : kL12Src2<FS>);
        u64             key;
        if (level == 2) {
            wave_sync(); // level 1's reads of its block and window are done
            put_block<FS ? 16 : 8, kSrcPitch<FS>>(lds, blk2, block_w, FS ? block_h : block_h >> 1);
            wave_sync();
        }
        if constexpr (!FS) {
            if (block_w == 64 && w == 8 && h >= 1 && h <= 3)
                key = level == 1 ? quick_search<8, 3>(src, win, ref, rows, ox + org_x, oy + org_y, h)
                                 : quick_search<16, 3>(src, win, ref, rows, ox + org_x, oy + org_y, h);
            else if (block_w == 64 && w == 8 && h >= 4 && h <= 8) // the first pass's 8 x 8 windows
                key = level == 1 ? quick_search<8, 8>(src, win, ref, rows, ox + org_x, oy + org_y, h)
                                 : quick_search<16, 8>(src, win, ref, rows, ox + org_x, oy + org_y, h);
            else
                key = small_search<FS, kL12WinBytes<FS>>(src, win, ref, bw, rows, ox + org_x, oy + org_y, w, h);
        } else {
            if (block_w == 64 && w >= 1 && w <= 16 && h >= 1 && h <= 16)
                key = level == 1 ? strip_search_fs<2, 32, 2>(src, win, p.zero, ref, rows, ox + org_x, oy + org_y, w, h)
                                 : strip_search_fs<4, 64, 4>(src, win, p.zero, ref, rows, ox + org_x, oy + org_y, w, h);
            else
                key = small_search<FS, kL12WinBytes<FS>>(src, win, ref, bw, rows, ox + org_x, oy + org_y, w, h);
        }
        uint32_t  sad;
        decode_key(key, org_x, org_y, level == 1 ? 2 : 1, cx, cy, sad, !FS);
        if (lane == 0) {
            rec->lx[level][i]   = cx;
            rec->ly[level][i]   = cy;
            rec->lsad[level][i] = sad;
        }
    }
}

static_assert(QuickGeo<16, 8>::WIN * 4 <= kL12WinBytes<false> && QuickGeo<8, 8>::WIN <= kL12Lds<false> - kL12Src1<false> - 64,
              "the first pass's 8 x 8 windows fit");
static_assert((16 + 63) * (4 + 4 * 4) * 4 <= kL12WinBytes<true> && (16 + 31) * (4 + 4 * 2) <= kL12Lds<true> - kL12Src1<true> - 64,
              "the temporal filter's 16 x 16 windows fit");
static_assert(kL12WinBytes<false> / 4 / 63 >= 16 + 4 && kL12WinBytes<true> / 4 / 64 >= 16 + 4, "small_search: four positions of a 64-pixel block fit");

// Both levels' source blocks of a 64x64 block, one wave: the sampled rows of the quarter-resolution block to `lds`, those of
// the full-resolution block into registers.
template <bool FS, class P>
__device__ __forceinline__ void fetch_l12_blocks(const P &p, uint32_t *lds, uint4 (&blk2)[kL12Blk2<FS>], int b64x, int b64y, int block_w, int block_h) {
    const auto &k = p.ctrl;
#pragma unroll
    for (int i = 0; i < kL12Blk2<FS>; ++i) blk2[i] = make_uint4(0, 0, 0, 0);
    if constexpr (FS) {
        if (k.enable_hme_level1_flag) stage_block<8, 1>(lds, load_plane(p.cur.lvl[1]), b64x >> 1, b64y >> 1, block_w >> 1, block_h >> 1);
        if (k.enable_hme_level2_flag) fetch_block<16, 1>(blk2, load_plane(p.cur.lvl[2]), b64x, b64y, block_w, block_h);
    } else {
        if (k.enable_hme_level1_flag)
            stage_block<8, 2, kSrcPitch<false>>(lds, load_plane(p.cur.lvl[1]), b64x >> 1, b64y >> 1, block_w >> 1, block_h >> 2);
        if (k.enable_hme_level2_flag) fetch_block<8, 2>(blk2, load_plane(p.cur.lvl[2]), b64x, b64y, block_w, block_h >> 1);
    }
}

} // namespace svtme
#endif
