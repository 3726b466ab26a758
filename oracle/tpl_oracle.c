/*
 * tpl_oracle.c - TEST INFRASTRUCTURE, not part of the product: a plain-C restatement of the source-based mode search of
 * the reference's TPL dispenser (tpl_mc_flow_dispenser_sb_generic,
 * Source/Lib/Encoder/Codec/EbSourceBasedOperationsProcess.c:632-933) for tpl_opt_flag = 1: costs as sums of absolute
 * differences or as sums of absolute 16x16 forward-DCT coefficients, and the quantisation error of the chosen inter
 * candidate (:935-982). Pinned by
 * tests/golden/tpl_*.npz, which hold what the real reference (oracle/_ref/ref_me, REF_MODE_TPL) computed for the same
 * pictures and ME results. Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it.
 *
 * Pictures come unpadded: the reference reads its padded planes (generate_padding replicates the border pixels), which is
 * the same as clamping the coordinates.
 */
#include "tpl_oracle.h"

#include <limits.h>
#include <math.h>
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


/* ---- the 16x16 forward DCT the dispenser costs residuals with: svt_av1_wht_fwd_txfm (EbTransforms.c:3815-3830) ->
 * svt_av1_fwd_txfm2d_16x16 (av1_tranform_two_d_core_c, :2301-2370, with fwd_shift_16x16 = {2, -2, 0} and cosine bits 13 for
 * the columns and 12 for the rows, EbTransforms.h:28, :46-57) over the 16-point butterfly network of svt_av1_fdct16_new
 * (:183-336). Every rounding of the network is a two-term rotation r(a, ca, b, cb) = (a ca + b cb + 2^(bit-1)) >> bit
 * (half_btf, EbInvTransforms.h:289-300); sums and differences are exact, so the network may be written part by part
 * (4-point core, odd half of the 8-point, odd half of the 16-point) instead of stage by stage. c[k] = cos(k pi / 128)
 * rounded to `bit` fractional bits (cospi_arr). The N2 / N4 shapes (:5392, :6998) keep the top-left 8x8 / 4x4
 * coefficients of the same transform and zero the rest. */
static int32_t rot(int32_t a, int32_t ca, int32_t b, int32_t cb, int bit) {
    return (int32_t)(((int64_t)a * ca + (int64_t)b * cb + ((int64_t)1 << (bit - 1))) >> bit);
}

static void fdct16(const int32_t *x, int32_t *X, int bit) {
    int32_t c[64];
    for (int k = 0; k < 64; ++k) c[k] = (int32_t)lround(cos(k * 3.14159265358979323846 / 128.0) * (double)(1 << bit));
    int32_t s[8], d[8];
    for (int i = 0; i < 8; ++i) {
        s[i] = x[i] + x[15 - i];
        d[i] = x[7 - i] - x[8 + i];
    }
    /* even half: the 8-point transform of s */
    int32_t p[4], q[4];
    for (int i = 0; i < 4; ++i) {
        p[i] = s[i] + s[7 - i];
        q[i] = s[3 - i] - s[4 + i];
    }
    const int32_t a0 = p[0] + p[3], a1 = p[1] + p[2], a2 = p[1] - p[2], a3 = p[0] - p[3];
    X[0]  = rot(a0, c[32], a1, c[32], bit);
    X[8]  = rot(a0, c[32], a1, -c[32], bit);
    X[4]  = rot(a2, c[48], a3, c[16], bit);
    X[12] = rot(a3, c[48], a2, -c[16], bit);
    const int32_t m1 = rot(q[2], c[32], q[1], -c[32], bit), m2 = rot(q[2], c[32], q[1], c[32], bit);
    const int32_t e0 = q[0] + m1, e1 = q[0] - m1, e2 = q[3] - m2, e3 = q[3] + m2;
    X[2]  = rot(e0, c[56], e3, c[8], bit);
    X[10] = rot(e1, c[24], e2, c[40], bit);
    X[6]  = rot(e2, c[24], e1, -c[40], bit);
    X[14] = rot(e3, c[56], e0, -c[8], bit);
    /* odd half */
    const int32_t g2 = rot(d[5], c[32], d[2], -c[32], bit), g5 = rot(d[5], c[32], d[2], c[32], bit);
    const int32_t g3 = rot(d[4], c[32], d[3], -c[32], bit), g4 = rot(d[4], c[32], d[3], c[32], bit);
    const int32_t h0 = d[0] + g3, h3 = d[0] - g3, h1 = d[1] + g2, h2 = d[1] - g2;
    const int32_t h4 = d[7] - g4, h7 = d[7] + g4, h5 = d[6] - g5, h6 = d[6] + g5;
    const int32_t k1 = rot(h6, c[48], h1, -c[16], bit), k6 = rot(h6, c[16], h1, c[48], bit);
    const int32_t k2 = rot(h2, -c[48], h5, -c[16], bit), k5 = rot(h5, c[48], h2, -c[16], bit);
    const int32_t n0 = h0 + k1, n1 = h0 - k1, n2 = h3 - k2, n3 = h3 + k2;
    const int32_t n4 = h4 + k5, n5 = h4 - k5, n6 = h7 - k6, n7 = h7 + k6;
    X[1]  = rot(n0, c[60], n7, c[4], bit);
    X[9]  = rot(n1, c[28], n6, c[36], bit);
    X[5]  = rot(n2, c[44], n5, c[20], bit);
    X[13] = rot(n3, c[12], n4, c[52], bit);
    X[3]  = rot(n4, c[12], n3, -c[52], bit);
    X[11] = rot(n5, c[44], n2, -c[20], bit);
    X[7]  = rot(n6, c[28], n1, -c[36], bit);
    X[15] = rot(n7, c[60], n0, -c[4], bit);
}

/* residual [16][16] (row-major) -> coefficients [16][16]; shape 0 / 1 / 2 = all / top-left 8x8 / top-left 4x4 */
void orc_tpl_fwd_dct16x16(const int16_t *residual, int32_t *coeff, int shape) {
    int32_t buf[256], in[16], out[16];
    for (int col = 0; col < 16; ++col) {
        for (int r = 0; r < 16; ++r) in[r] = residual[r * 16 + col] * 4; /* shift[0] = 2 */
        fdct16(in, out, 13);
        for (int r = 0; r < 16; ++r) buf[r * 16 + col] = (out[r] + 2) >> 2; /* shift[1] = -2 */
    }
    for (int r = 0; r < 16; ++r) fdct16(buf + r * 16, coeff + r * 16, 12); /* shift[2] = 0 */
    const int keep = shape == 2 ? 4 : (shape == 1 ? 8 : 16);
    for (int i = 0; i < 256; ++i)
        if ((i & 15) >= keep || (i >> 4) >= keep) coeff[i] = 0;
}

/* svt_aom_satd: common_dsp_rtcd.c:47-54 */
static int64_t satd256(const int32_t *coeff) {
    int64_t sum = 0;
    for (int i = 0; i < 256; ++i) sum += coeff[i] < 0 ? -(int64_t)coeff[i] : coeff[i];
    return sum;
}

/* get_quantize_error (:167-193) for TX_16X16: svt_av1_quantize_fp_c (EbFullLoop.c:318-405, no quantisation matrix, log
 * scale 0), svt_av1_block_error_c, >> 2, at least 1 */
int64_t orc_tpl_quantize_error16x16(const int32_t *coeff, const int16_t round_fp[2], const int16_t quant_fp[2], const int16_t dequant[2]) {
    int64_t err = 0;
    for (int i = 0; i < 256; ++i) {
        const int     ac = i != 0, v = coeff[i], neg = v < 0;
        int64_t       a = neg ? -(int64_t)v : v;
        int32_t       dq = 0;
        if ((a << 1) >= dequant[ac]) {
            a += round_fp[ac];
            a = a > INT16_MAX ? INT16_MAX : (a < INT16_MIN ? INT16_MIN : a);
            const int32_t level = (int32_t)((a * quant_fp[ac]) >> 16);
            if (level) dq = neg ? -(level * dequant[ac]) : level * dequant[ac];
        }
        const int64_t diff = (int64_t)v - dq;
        err += diff * diff;
    }
    err >>= 2;
    return err < 1 ? 1 : err;
}

int orc_tpl_search(const uint8_t *cur_pix, int stride, const uint8_t *const ref_pix[SVTME_MAX_LISTS * SVTME_MAX_REFS],
                   const svtme_tpl_picture *p, svtme_tpl_block *out) {
    if (!p->tpl_opt_flag || p->pf_shape > 2) return SVTME_ERR_UNSUPPORTED;
    const int sad_intra = p->use_pred_sad_in_intra_search, sad_inter = p->use_pred_sad_in_inter_search;
    /* transform-domain costs and the source distortion: 16x16 blocks, whole transforms */
    if ((!sad_intra || !sad_inter || p->src_dist) && (p->dispenser_search_level != 0 || p->subsample_tx != 0)) return SVTME_ERR_UNSUPPORTED;
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
                int16_t   res[256];
                int32_t   coeff[256];
                for (int r = 0; r < size; ++r)
                    for (int c = 0; c < size; ++c) {
                        const int d = pel(&cur, x + c, y + r) - dc;
                        sad += d < 0 ? -d : d;
                        if (!sad_intra) res[r * 16 + c] = (int16_t)d;
                    }
                if (!sad_intra) { /* :790-813 */
                    orc_tpl_fwd_dct16x16(res, coeff, p->pf_shape);
                    sad = satd256(coeff);
                }
                o->best_intra_cost = sad;
            }
            int64_t best_err = 0;
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
                int16_t     res[256];
                int32_t     coeff[256];
                for (int r = 0; r < size; ++r)
                    for (int cc = 0; cc < size; ++cc) {
                        const int d = pel(&cur, x + cc, y + r) - pel(&rp, x + (mvx >> 3) + cc, y + (mvy >> 3) + r);
                        sad += d < 0 ? -d : d;
                        if (size == 16) res[r * 16 + cc] = (int16_t)d;
                    }
                if (!sad_inter) { /* :894-917 */
                    orc_tpl_fwd_dct16x16(res, coeff, p->pf_shape);
                    sad = satd256(coeff);
                }
                if (sad < o->best_inter_cost) {
                    if (p->src_dist) { /* what :935-982 make of the best candidate */
                        if (sad_inter) orc_tpl_fwd_dct16x16(res, coeff, p->pf_shape);
                        best_err = orc_tpl_quantize_error16x16(coeff, p->y_round_fp, p->y_quant_fp, p->y_dequant);
                    }
                    o->best_inter_cost = sad;
                    o->best_rf_idx     = (int8_t)(list == 0 ? ref : 4 + ref); /* svt_get_ref_frame_type(list, ref) - 1 */
                    o->mv_x            = mvx;
                    o->mv_y            = mvy;
                }
            }
            o->best_is_inter = o->best_inter_cost < o->best_intra_cost; /* :932-933 */
            o->src_recon_error = o->best_is_inter ? best_err : 0;
        }
    }
    return SVTME_OK;
}
