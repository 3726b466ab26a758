/*
 * tpl_oracle.c - TEST INFRASTRUCTURE, not part of the product: a plain-C restatement of the source-based mode search of
 * the reference's TPL dispenser (tpl_mc_flow_dispenser_sb_generic,
 * Source/Lib/Encoder/Codec/EbSourceBasedOperationsProcess.c:632-933) for the controls whose costs are sums of absolute
 * differences (tpl_opt_flag with use_pred_sad_in_intra_search / use_pred_sad_in_inter_search). Pinned by
 * tests/golden/tpl_*.npz, which hold what the real reference (oracle/_ref/ref_me, REF_MODE_TPL) computed for the same
 * pictures and ME results. Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it.
 *
 * Pictures come unpadded: the reference reads its padded planes (generate_padding replicates the border pixels), which is
 * the same as clamping the coordinates.
 */
#include "tpl_oracle.h"

#include <limits.h>
#include <string.h>

#define TPL_PAD 32 /* TPL_PADX, TPL_PADY: EbEncodeContext.h:43-44 */

typedef struct plane {
    const uint8_t *pix;
    int            stride, w, h;
} plane;

static int pel(const plane *p, int x, int y) {
    x = x < 0 ? 0 : (x > p->w - 1 ? p->w - 1 : x);
    y = y < 0 ? 0 : (y > p->h - 1 ? p->h - 1 : y);
    return p->pix[(size_t)y * p->stride + x];
}

/* tpl_blk_idx_tab (:514-516): position of block i in the 64x64 block and its index in the ME results */
static const uint8_t k_me_index[SVTME_TPL_BLOCKS] = {0, 1, 2, 3, 4, 5, 6, 9, 10, 7, 8, 11, 12, 13, 14, 17, 18, 15, 16, 19, 20};
static void blk_geometry(int i, int *x, int *y, int *size) {
    if (i == 0) {
        *x = *y = 0;
        *size   = 64;
    } else if (i < 5) {
        *x    = ((i - 1) & 1) * 32;
        *y    = ((i - 1) >> 1) * 32;
        *size = 32;
    } else { /* z-order inside each 32x32 */
        const int q = (i - 5) >> 2, s = (i - 5) & 3;
        *x    = (q & 1) * 32 + (s & 1) * 16;
        *y    = (q >> 1) * 32 + (s >> 1) * 16;
        *size = 16;
    }
}

/* The DC predictor over the neighbours update_neighbor_samples_array_open_loop_mb / get_neighbor_samples_dc collect
 * (EbEncIntraPrediction.c:1150-1235; :524-536): the row above and the column left of the block in the source picture;
 * entries past the picture's right / bottom edge keep their fill values 127 / 129; dc_pred[have_left][have_top]
 * (EbIntraPrediction.c:985-1035, :2705-2707). */
static int dc_value(const plane *cur, int x, int y, int size) {
    const int have_left = x > 0, have_top = y > 0;
    int       sum = 0, n = 0;
    if (have_top) {
        for (int i = 0; i < size; ++i) sum += x + i < cur->w ? pel(cur, x + i, y - 1) : 127;
        n += size;
    }
    if (have_left) {
        for (int i = 0; i < size; ++i) sum += y + i < cur->h ? pel(cur, x - 1, y + i) : 129;
        n += size;
    }
    return n ? (sum + (n >> 1)) / n : 128;
}

int orc_tpl_search(const uint8_t *cur_pix, int stride, const uint8_t *const ref_pix[SVTME_MAX_LISTS * SVTME_MAX_REFS],
                   const svtme_tpl_picture *p, svtme_tpl_block *out) {
    if (!p->tpl_opt_flag || !p->use_pred_sad_in_intra_search || !p->use_pred_sad_in_inter_search) return SVTME_ERR_UNSUPPORTED;
    const int   w = (int)p->width, h = (int)p->height, b64_w = (w + 63) / 64, b64_h = (h + 63) / 64;
    const plane cur = {cur_pix, stride, w, h};
    memset(out, 0, sizeof(*out) * (size_t)b64_w * b64_h * SVTME_TPL_BLOCKS);
    for (int b64 = 0; b64 < b64_w * b64_h; ++b64) {
        const int sb_x = (b64 % b64_w) * 64, sb_y = (b64 / b64_w) * 64;
        const int complete = sb_x + 64 <= w && sb_y + 64 <= h; /* sb_params->width == 64 && ->height == 64, :2086 */
        const int level = complete ? p->dispenser_search_level : 0;
        const int first = level == 0 ? 5 : (level == 1 ? 1 : 0), last = level == 0 ? 20 : (level == 1 ? 4 : 0); /* :540-542 */
        const uint8_t  *cnt  = p->total_me_candidate_index + (size_t)b64 * p->n_pu;
        const uint8_t  *cand = p->me_candidate_array + (size_t)b64 * p->n_pu * p->max_cand;
        const uint32_t *mvs  = p->me_mv_array + (size_t)b64 * p->n_pu * p->max_refs;
        for (int bi = first; bi <= last; ++bi) {
            int bx, by, size;
            blk_geometry(bi, &bx, &by, &size);
            const int x = sb_x + bx, y = sb_y + by;
            if (x + (size >> 1) > w || y + (size >> 1) > h) continue; /* :604-607 */
            svtme_tpl_block *o = &out[(size_t)b64 * SVTME_TPL_BLOCKS + bi];
            o->valid           = 1;
            o->best_rf_idx     = -1;
            o->best_intra_cost = INT64_MAX;
            o->best_inter_cost = INT64_MAX;
            if (!p->disable_intra_pred) { /* :637-692 */
                const int dc = dc_value(&cur, x, y, size);
                int64_t   sad = 0;
                for (int r = 0; r < size; ++r)
                    for (int c = 0; c < size; ++c) {
                        const int d = pel(&cur, x + c, y + r) - dc;
                        sad += d < 0 ? -d : d;
                    }
                o->best_intra_cost = sad;
            }
            int me_off = k_me_index[bi]; /* :826-834 */
            if (!p->enable_me_16x16) me_off = (me_off - 1) / 4;
            const int n_cand = p->i_slice ? 0 : cnt[me_off];
            for (int ci = 0; ci < n_cand; ++ci) { /* :836-930 */
                const uint8_t c = cand[(size_t)me_off * p->max_cand + ci];
                const int     direction = c & 3;
                if (direction > 1) continue;
                const int list = direction, ref = list == 0 ? (c >> 2) & 3 : (c >> 4) & 3;
                if (ref >= p->n_ref[list] || !ref_pix[list * SVTME_MAX_REFS + ref]) return SVTME_ERR_BAD_ARG;
                if (p->ref_skip[list][ref]) continue;
                const uint32_t mv = mvs[(size_t)me_off * p->max_refs + (list ? p->max_l0 : 0) + ref];
                int16_t mvx = (int16_t)((int16_t)(mv & 0xffff) << 1), mvy = (int16_t)((int16_t)(mv >> 16) << 1);
                /* max_width / max_height of the reference picture: the picture size */
                if (x + (mvx >> 3) < -TPL_PAD) mvx = (int16_t)((-TPL_PAD - x) << 3);
                if (x + size + (mvx >> 3) > TPL_PAD + w - 1) mvx = (int16_t)(((TPL_PAD + w - 1) - (x + size)) << 3);
                if (y + (mvy >> 3) < -TPL_PAD) mvy = (int16_t)((-TPL_PAD - y) << 3);
                if (y + size + (mvy >> 3) > TPL_PAD + h - 1) mvy = (int16_t)(((TPL_PAD + h - 1) - (y + size)) << 3);
                const plane rp = {ref_pix[list * SVTME_MAX_REFS + ref], stride, w, h};
                int64_t     sad = 0;
                for (int r = 0; r < size; ++r)
                    for (int cc = 0; cc < size; ++cc) {
                        const int d = pel(&cur, x + cc, y + r) - pel(&rp, x + (mvx >> 3) + cc, y + (mvy >> 3) + r);
                        sad += d < 0 ? -d : d;
                    }
                if (sad < o->best_inter_cost) {
                    o->best_inter_cost = sad;
                    o->best_rf_idx     = (int8_t)(list == 0 ? ref : 4 + ref); /* svt_get_ref_frame_type(list, ref) - 1 */
                    o->mv_x            = mvx;
                    o->mv_y            = mvy;
                }
            }
            o->best_is_inter = o->best_inter_cost < o->best_intra_cost; /* :932-933 */
        }
    }
    return SVTME_OK;
}
