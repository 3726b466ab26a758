/*
 * me_oracle.c - TEST INFRASTRUCTURE (see me_oracle.h). Scalar CPU restatement of the reference's open-loop ME:
 * reference store construction, preset -> control derivation, pre-HME, HME L0/L1/L2, search-centre selection,
 * HME pruning / search-range adjustment, the integer full-pel search with its 85 running minima, and the host
 * tail (ME pruning, candidate construction, distortion statistics, global-motion detection).
 * Every function names the reference lines it follows. Citations are relative to /root/reference/.
 */
#include "me_oracle.h"

#include <pthread.h>
#include <stdlib.h>
#include <string.h>

#define ORC_MAX(a, b) ((a) > (b) ? (a) : (b))
#define ORC_MIN(a, b) ((a) < (b) ? (a) : (b))
#define ORC_ABS(a) ((a) < 0 ? -(a) : (a))

/* ------------------------------------------------------------------------------------------ reference store */

/* generate_padding, Source/Lib/Common/Codec/EbMcp.c:112-164: left/right replication per picture row, then whole
 * padded rows copied upwards and downwards. */
static void orc_replicate(orc_plane *p) {
    uint8_t *row = p->buf + (size_t)p->pad * p->stride + p->pad;
    for (uint32_t y = 0; y < p->height; ++y, row += p->stride) {
        memset(row - p->pad, row[0], p->pad);
        memset(row + p->width, row[p->width - 1], p->pad);
    }
    uint8_t *top = p->buf + (size_t)p->pad * p->stride;
    uint8_t *bot = p->buf + (size_t)(p->pad + p->height - 1) * p->stride;
    for (uint32_t k = 1; k <= p->pad; ++k) {
        memcpy(top - (size_t)k * p->stride, top, p->stride);
        memcpy(bot + (size_t)k * p->stride, bot, p->stride);
    }
}

static int orc_plane_alloc(orc_plane *p, uint32_t w, uint32_t h, uint32_t pad) {
    p->width  = w;
    p->height = h;
    p->pad    = pad;
    p->stride = w + 2 * pad;
    p->buf    = (uint8_t *)calloc((size_t)p->stride * (h + 2 * pad) + 64, 1);
    return p->buf ? 0 : -1;
}

/* downsample_2d_c, Source/Lib/Encoder/Codec/EbPictureAnalysisProcess.c:187-219. step 2: 2x2 box with rounding;
 * step 4: the centre 2x2 of every 4x4 (rows/cols 1,2). */
static void orc_downsample(const orc_plane *in, orc_plane *out, uint32_t step) {
    const uint8_t *src  = in->buf + (size_t)in->pad * in->stride + in->pad;
    uint8_t       *dst  = out->buf + (size_t)out->pad * out->stride + out->pad;
    const uint32_t half = step >> 1;
    uint32_t       oy   = 0;
    for (uint32_t y = half; y < in->height; y += step, ++oy) {
        const uint8_t *r0 = src + (size_t)(y - 1) * in->stride;
        const uint8_t *r1 = src + (size_t)y * in->stride;
        uint32_t       ox = 0;
        for (uint32_t x = half; x < in->width; x += step, ++ox)
            dst[(size_t)oy * out->stride + ox] = (uint8_t)((r0[x - 1] + r0[x] + r1[x - 1] + r1[x] + 2) >> 2);
    }
}

/* Store geometry: EbEncHandle.c:1331-1355 (quarter pad 32, sixteenth pad 16), :3447-3460 (full pad 72).
 * Build order: pad_input_pictures (EbPictureAnalysisProcess.c:2754-2764) then
 * downsample_filtering_input_picture (:2688-2752): quarter from full; sixteenth from the *quarter* plane when
 * HME level 1 is on, else straight from full with step 4. */
int orc_planes_build(orc_planes *p, const uint8_t *luma, uint32_t stride, uint32_t width, uint32_t height,
                     int hme_level1_enabled) {
    memset(p, 0, sizeof(*p));
    if (orc_plane_alloc(&p->lvl[2], width, height, 72) || orc_plane_alloc(&p->lvl[1], width >> 1, height >> 1, 32) ||
        orc_plane_alloc(&p->lvl[0], width >> 2, height >> 2, 16)) {
        orc_planes_free(p);
        return -1;
    }
    orc_plane *f = &p->lvl[2];
    for (uint32_t y = 0; y < height; ++y)
        memcpy(f->buf + (size_t)(y + f->pad) * f->stride + f->pad, luma + (size_t)y * stride, width);
    orc_replicate(f);
    if (hme_level1_enabled) {
        orc_downsample(f, &p->lvl[1], 2);
        orc_replicate(&p->lvl[1]);
        orc_downsample(&p->lvl[1], &p->lvl[0], 2);
    } else {
        orc_downsample(f, &p->lvl[0], 4);
    }
    orc_replicate(&p->lvl[0]);
    return 0;
}

void orc_planes_free(orc_planes *p) {
    for (int i = 0; i < 3; ++i) {
        free(p->lvl[i].buf);
        p->lvl[i].buf = NULL;
    }
}

/* ------------------------------------------------------------------------------------------------ controls */

/* derive_input_resolution, Source/Lib/Encoder/Codec/EbSequenceControlSet.c:396-415; thresholds
 * Source/Lib/Common/Codec/EbDefinitions.h:2025-2030. */
uint8_t orc_input_resolution(uint32_t width, uint32_t height) {
    static const uint32_t th[6] = {0x28500, 0x4CE00, 0xA1400, 0x16DA00, 0x535200, 0x140A000};
    uint32_t              sz    = width * height;
    uint8_t               r     = 0;
    while (r < 6 && sz >= th[r]) ++r;
    return r;
}

static svtme_sa sa(uint16_t w, uint16_t h) {
    svtme_sa s = {w, h};
    return s;
}

enum { RES_480P = 2, RES_720P = 3, RES_1080P = 4, RES_4K = 5 };

/* signal_derivation_me_kernel_oq and its setters, Source/Lib/Encoder/Codec/EbMotionEstimationProcess.c:68-538;
 * HME level flags from signal_derivation_multi_processes_oq, EbPictureDecisionProcess.c:1797-1814. */
int orc_derive_controls(const svtme_preset_key *k, svtme_controls *c) {
    const int m = k->enc_mode, sc = k->sc_class1, res = k->input_resolution;
    if (m < -2 || m > 13) return -1;
    memset(c, 0, sizeof(*c));

    /* HME level flags (EbPictureDecisionProcess.c:1797-1814) */
    c->enable_hme_flag = c->enable_hme_level0_flag = 1;
    if (sc || m <= 6) c->enable_hme_level1_flag = c->enable_hme_level2_flag = 1;
    else c->enable_hme_level1_flag = 1; /* m <= 13 always holds */

    /* set_me_search_params :154-240 */
    svtme_sa mn, mx;
    if (sc) {
        if (m <= 2) mn = sa(175, 175), mx = sa(750, 750);
        else if (m <= 4) mn = sa(125, 125), mx = sa(500, 500);
        else if (m <= 6) mn = sa(75, 75), mx = sa(350, 350);
        else if (m <= 7) mn = sa(48, 48), mx = sa(224, 224);
        else if (m <= 8) mn = sa(32, 32), mx = sa(164, 164);
        else if (m <= 10) mn = sa(32, 32), mx = sa(96, 96);
        else if (m <= 12) mn = sa(16, 16), mx = sa(96, 96);
        else mn = sa(8, 8), mx = sa(32, 32);
    } else if (m <= 0) mn = sa(64, 64), mx = sa(256, 256);
    else if (m <= 2) mn = sa(64, 64), mx = sa(128, 128);
    else if (m <= 4) mn = sa(16, 16), mx = sa(64, 64);
    else if (m <= 5) mn = sa(16, 16), mx = sa(64, 32);
    else if (m <= 9) {
        if (res < RES_1080P) mn = sa(16, 16), mx = sa(32, 16);
        else mn = sa(16, 6), mx = sa(16, 9);
    } else if (m <= 11) {
        if (res < RES_4K) mn = sa(8, 5), mx = sa(16, 9);
        else mn = sa(8, 1), mx = sa(8, 1);
    } else if (m <= 12) {
        if (res < RES_720P) mn = sa(8, 3), mx = sa(16, 9);
        else if (res < RES_1080P) mn = sa(8, 1), mx = sa(16, 7);
        else if (res < RES_4K) mn = sa(8, 1), mx = sa(8, 7);
        else mn = sa(8, 1), mx = sa(8, 1);
    } else mn = sa(8, 3), mx = sa(8, 3);
    if ((k->frame_rate >> 16) < 50) {
        mn.width  = (uint16_t)((mn.width * 3) >> 1);
        mn.height = (uint16_t)((mn.height * 3) >> 1);
    }
    c->me_sa.sa_min = mn;
    c->me_sa.sa_max = mx;

    /* set_hme_search_params :68-149 */
    c->num_hme_sa_w = c->num_hme_sa_h = 2;
    if (m <= -2) mn = sa(240, 240), mx = sa(480, 480);
    else if (m <= 0) {
        if (res < RES_4K) mn = sa(32, 32), mx = sa(192, 192);
        else mn = sa(240, 240), mx = sa(480, 480);
    } else if (m <= 5) mn = sa(32, 32), mx = sa(192, 192);
    else if (m <= 6) {
        if (res <= RES_1080P && !sc) mn = sa(16, 16), mx = sa(192, 192);
        else mn = sa(32, 32), mx = sa(192, 192);
    } else if (m <= 11) {
        if (sc) mn = sa(32, 32), mx = sa(192, 192);
        else mn = sa(16, 16), mx = sa(192, 192);
    } else { /* m <= 13 */
        if (sc) mn = sa(32, 32), mx = sa(192, 192);
        else if (res < RES_4K) mn = sa(8, 8), mx = sa(96, 96);
        else mn = sa(16, 16), mx = sa(96, 96);
    }
    c->hme_l0_sa.sa_min = mn;
    c->hme_l0_sa.sa_max = mx;
    c->hme_l1_sa = c->hme_l2_sa = (m <= 1) ? sa(16, 16) : sa(8, 3);

    c->hme_search_method = 0; /* SUB_SAD_SEARCH :448-449 */
    c->me_search_method  = 0;
    c->stat_factor       = (sc || m <= 12) ? 100 : 80;
    if (!sc && m > 10) {
        c->reduce_hme_l0_sr_th_min = 8;
        c->reduce_hme_l0_sr_th_max = 200;
    }

    /* pre-HME :469-485, set_prehme_ctrls :385-424 */
    int prehme_level = sc ? 1 : (m <= 11 ? 1 : 3);
    if (!c->enable_hme_level1_flag) prehme_level = 0;
    c->prehme_enable = prehme_level != 0;
    if (prehme_level == 1) {
        c->prehme_sa_cfg[0].sa_min = sa(8, 100);
        c->prehme_sa_cfg[0].sa_max = sa(8, 400);
        c->prehme_sa_cfg[1].sa_min = sa(96, 3);
        c->prehme_sa_cfg[1].sa_max = sa(384, 3);
    } else if (prehme_level == 3) {
        c->prehme_sa_cfg[0].sa_min = sa(8, 100);
        c->prehme_sa_cfg[0].sa_max = sa(8, 350);
        c->prehme_sa_cfg[1].sa_min = sa(32, 7);
        c->prehme_sa_cfg[1].sa_max = sa(128, 7);
        c->prehme_skip_search_line = 1;
        c->prehme_l1_early_exit    = 1;
    }

    /* reference pruning :487-506, set_me_hme_ref_prune_ctrls :241-288 */
    int prune_level;
    if (m <= -2) prune_level = 0;
    else if (m <= 1) prune_level = 2;
    else if (!sc || m <= 9) prune_level = 5;
    else prune_level = 6;
    if (prune_level == 0) {
        c->prune_ref_if_hme_sad_dev_bigger_than_th = 0xFFFF;
        c->prune_ref_if_me_sad_dev_bigger_than_th  = 0xFFFF;
    } else {
        c->enable_me_hme_ref_pruning               = 1;
        c->prune_ref_if_hme_sad_dev_bigger_than_th = prune_level == 2 ? 80 : (prune_level == 5 ? 5 : 200);
        c->prune_ref_if_me_sad_dev_bigger_than_th  = 60;
        c->protect_closest_refs                    = prune_level != 6;
    }

    /* search-range adjustment :507-520, set_me_sr_adjustment_ctrls :290-359 */
    int sr_level;
    if (sc) sr_level = m <= 9 ? 4 : 5;
    else if (m <= -2) sr_level = 0;
    else if (m <= 3) sr_level = 1;
    else if (m <= 4) sr_level = 2;
    else sr_level = 3;
    if (sr_level) {
        c->enable_me_sr_adjustment            = sr_level >= 4 ? 2 : 1;
        c->reduce_me_sr_based_on_mv_length_th = sr_level == 4 ? 16 : (sr_level == 5 ? 20 : 4);
        c->stationary_hme_sad_abs_th          = sr_level == 4 ? 20000 : (sr_level == 5 ? 24000 : 12000);
        c->stationary_me_sr_divisor           = 8;
        c->reduce_me_sr_based_on_hme_sad_abs_th =
            sr_level == 4 ? 20000 : (sr_level == 5 ? 24000 : (sr_level == 3 ? 12000 : 6000));
        c->me_sr_divisor_for_low_hme_sad = 8;
        c->distance_based_hme_resizing   = sr_level >= 2;
        if (!c->enable_hme_level2_flag) {
            int d = c->enable_hme_level1_flag ? 4 : 16;
            c->stationary_hme_sad_abs_th            = (uint16_t)(c->stationary_hme_sad_abs_th / d);
            c->reduce_me_sr_based_on_hme_sad_abs_th = (uint16_t)(c->reduce_me_sr_based_on_hme_sad_abs_th / d);
        }
    }

    c->prune_me_candidates_th     = m <= 7 ? 0 : 65;
    c->use_best_unipred_cand_only = m <= 9 ? 0 : 1; /* EbPictureDecisionProcess.c:2058-2061 */
    c->me_early_exit_th           = m <= 6 ? 0 : (m <= 11 ? 64 * 64 * 8 : 64 * 64 * 9);
    return 0;
}

/* tf_set_me_hme_params_oq + tf_signal_derivation_me_kernel_oq, EbMotionEstimationProcess.c:554-629, with the HME level
 * flags signal_derivation_multi_processes_oq sets per hme_me_level (EbPictureDecisionProcess.c:1816-1841) and the search
 * method create_me_context_and_picture_control overrides for every block (EbTemporalFiltering.c:331). */
int orc_derive_tf_controls(uint8_t level, svtme_controls *c) {
    if (!c || level > 2) return -1;
    memset(c, 0, sizeof(*c));
    c->num_hme_sa_w = c->num_hme_sa_h = 2;
    c->hme_l1_sa = sa(16, 16);
    c->hme_l2_sa = sa(16, 16);
    if (level == 0) {
        c->hme_l0_sa.sa_min = sa(30, 30);
        c->hme_l0_sa.sa_max = sa(60, 60);
        c->me_sa.sa_min     = sa(60, 60);
        c->me_sa.sa_max     = sa(120, 120);
    } else if (level == 1) {
        c->hme_l0_sa.sa_min = sa(16, 16);
        c->hme_l0_sa.sa_max = sa(32, 32);
        c->me_sa.sa_min     = sa(16, 16);
        c->me_sa.sa_max     = sa(32, 32);
    } else {
        c->hme_l0_sa.sa_min = sa(8, 8);
        c->hme_l0_sa.sa_max = sa(16, 16);
        c->me_sa.sa_min     = sa(8, 4);
        c->me_sa.sa_max     = sa(16, 8);
    }
    c->enable_hme_flag        = 1;
    c->enable_hme_level0_flag = 1;
    c->enable_hme_level1_flag = level <= 1;
    c->enable_hme_level2_flag = level == 0;
    c->hme_search_method      = 1; /* FULL_SAD_SEARCH */
    c->me_search_method       = 0; /* SUB_SAD_SEARCH */
    c->stat_factor            = 100;
    /* set_me_hme_ref_prune_ctrls(0), :245-249 */
    c->prune_ref_if_hme_sad_dev_bigger_than_th = 0xFFFF;
    c->prune_ref_if_me_sad_dev_bigger_than_th  = 0xFFFF;
    return 0;
}

/* first_pass_signal_derivation_me_kernel, firstpass.c:674-716: set_first_pass_me_hme_params_oq (:627-662), every HME
 * level on (first_pass_signal_derivation_multi_processes :488-491) except level 2 with reduce_me_search, no pre-HME, no
 * pruning, no search-range adjustment. */
int orc_derive_first_pass_controls(uint8_t res, uint32_t frame_rate, uint8_t reduce_me_search, svtme_controls *c) {
    if (!c || res > 6) return -1;
    memset(c, 0, sizeof(*c));
    c->stat_factor  = 100;
    c->num_hme_sa_w = c->num_hme_sa_h = 2;
    c->me_sa.sa_min = reduce_me_search ? sa(8, 3) : sa(8, 8);
    c->me_sa.sa_max = reduce_me_search ? sa(8, 5) : sa(8, 8);
    c->hme_l0_sa.sa_min = res < 4 ? sa(4, 4) : sa(8, 8); /* INPUT_SIZE_1080p_RANGE = 4 */
    c->hme_l0_sa.sa_max = sa(96, 96);
    c->hme_l1_sa = sa(8, 8);
    c->hme_l2_sa = sa(8, 8);
    if ((frame_rate >> 16) < 50) {
        c->me_sa.sa_min.width  = (uint16_t)((c->me_sa.sa_min.width * 3) >> 1);
        c->me_sa.sa_min.height = (uint16_t)((c->me_sa.sa_min.height * 3) >> 1);
    }
    c->enable_hme_flag = c->enable_hme_level0_flag = c->enable_hme_level1_flag = 1;
    c->enable_hme_level2_flag = !reduce_me_search;
    c->prune_ref_if_hme_sad_dev_bigger_than_th = 0xFFFF;
    c->prune_ref_if_me_sad_dev_bigger_than_th  = 0xFFFF;
    return 0;
}

/* ----------------------------------------------------------------------------------------------- ME context */

typedef struct orc_search_info {
    int16_t  x, y;
    uint64_t sad;
} orc_search_info;

struct orc_ctx {
    /* persistent per-b64 state: MeContext fields, Source/Lib/Encoder/Codec/EbMotionEstimationContext.h:338-479 */
    uint32_t        best_sad[2][4][85];
    uint32_t        best_mv[2][4][85];
    int16_t         sc_x[2][4], sc_y[2][4];
    uint64_t        hme_sad[2][4];
    uint8_t         do_ref[2][4];
    uint32_t        zz_sad[2][4];
    uint32_t        sr_div[2][4];
    int16_t         lx[3][2][4][2][2], ly[3][2][4][2][2]; /* [level][list][ref][sr_w][sr_h] */
    uint64_t        lsad[3][2][4][2][2];
    orc_search_info pre[2][4][2];
    uint32_t        me_distortion[85];
    /* host-tail outputs per b64 (pa_me_data->me_results[b64], pcs stats) */
    uint8_t  total_cand[85];
    uint32_t mv_array[85 * REF_DUMP_MAX_REFS];
    uint8_t  cand_array[85 * REF_DUMP_MAX_CAND];
    uint32_t stat[6];
    uint8_t  stationary, allow_gm;
    uint8_t  tf_exit; /* tf_use_pred_64x64_only_th == ~0 after the tf_me_exit_th test */
    /* scratch */
    svtme_sa_minmax hme_l0_sa; /* working copy, rescaled per reference */
    uint32_t        block_w, block_h;
    uint64_t        ops[6];
};

orc_ctx *orc_ctx_new(void) { return (orc_ctx *)calloc(1, sizeof(orc_ctx)); }
void     orc_ctx_free(orc_ctx *c) { free(c); }
void     orc_ctx_ops(const orc_ctx *c, uint64_t ops[6]) { memcpy(ops, c->ops, sizeof(c->ops)); }

/* -------------------------------------------------------------------------------------------- SAD primitives */

/* svt_nxm_sad_kernel_helper_c, Source/Lib/Encoder/C_DEFAULT/EbComputeSAD_C.c:20-37,215-218 */
static uint32_t orc_sad(const uint8_t *src, uint32_t src_stride, const uint8_t *ref, uint32_t ref_stride,
                        uint32_t height, uint32_t width) {
    uint32_t sad = 0;
    for (uint32_t y = 0; y < height; ++y, src += src_stride, ref += ref_stride)
        for (uint32_t x = 0; x < width; ++x) sad += (uint32_t)ORC_ABS((int)src[x] - (int)ref[x]);
    return sad;
}

/* svt_sad_loop_kernel_c, EbComputeSAD_C.c:58-103: exhaustive search, first strictly smaller SAD in raster order
 * wins; even search rows are skipped only for 16-wide blocks of height <= 16 when skip_search_line is set; the
 * outputs keep their previous contents when nothing beats 0xffffff. */
static void orc_sad_loop(const uint8_t *src, uint32_t src_stride, const uint8_t *ref, uint32_t ref_stride,
                         uint32_t block_h, uint32_t block_w, uint64_t *best_sad, int16_t *best_x, int16_t *best_y,
                         uint32_t ref_stride_raw, int skip_search_line, int sa_w, int sa_h, uint64_t *ops) {
    *best_sad = 0xffffff;
    for (int y = 0; y < sa_h; ++y, ref += ref_stride_raw) {
        if (block_w == 16 && block_h <= 16 && skip_search_line && (y & 1) == 0) continue;
        for (int x = 0; x < sa_w; ++x) {
            uint32_t sad = orc_sad(src, src_stride, ref + x, ref_stride, block_h, block_w);
            if (sad < *best_sad) {
                *best_sad = sad;
                *best_x   = (int16_t)x;
                *best_y   = (int16_t)y;
            }
        }
        *ops += (uint64_t)(sa_w > 0 ? sa_w : 0) * block_h * block_w;
    }
}

/* get_scaled_picture_distance, EbMotionEstimation.c:1276-1280 */
static uint16_t orc_scaled_dist(uint16_t dist) { return (uint16_t)(((dist * 5) / 8) + ((dist % 8) ? 1 : 0)); }

/* dist as get_me_reference computes it, :1269-1271 */
static uint16_t orc_dist(const orc_pic *pic, int l, int r) {
    int64_t d = (int64_t)pic->picture_number - (int64_t)pic->ref_picture_number[l][r];
    return (uint16_t)(int16_t)ORC_ABS(d);
}

/* The window correction every stage applies per axis ("shift, don't shrink" on the low side):
 * :875-907 (HME L0), :976-1009 (L1), :1077-1105 (L2), :1441-1460 / :1502-1520 (ME), :1638-1682 (pre-HME).
 * `o` = block origin, `lo` = lowest allowed coordinate of the first search position, `dim` = picture size. */
static void orc_clip_axis(int o, int16_t *org, int16_t *size, int lo, int dim) {
    if (o + *org < lo) {
        *org  = (int16_t)(lo - o);
        *size = (int16_t)(*size - (lo - (o + *org))); /* the origin is already corrected: subtracts 0 */
    }
    if (o + *org > dim - 1) *org = (int16_t)(*org - ((o + *org) - (dim - 1)));
    if (o + *org + *size > dim) *size = (int16_t)ORC_MAX(1, *size - ((o + *org + *size) - dim));
}

static const uint8_t *orc_block_ptr(const orc_plane *p, int x, int y) {
    return p->buf + (size_t)((int)p->pad + y) * p->stride + ((int)p->pad + x);
}

/* One sad_loop search of stage level (0 = sixteenth ... 2 = full) at window origin (org_x, org_y) relative to
 * the block; shared by pre-HME and HME L0/L1/L2 (:909-944, :1011-1046, :1107-1137, :1684-1716). Applies the
 * SUB_SAD halving (every other row, result x2). */
static void orc_level_search(orc_ctx *c, const orc_pic *pic, const orc_plane *cur, const orc_plane *ref, int ox, int oy,
                             uint32_t bw, uint32_t bh, int org_x, int org_y, int sa_w, int sa_h, int skip,
                             uint64_t *sad, int16_t *mvx, int16_t *mvy, uint64_t *ops) {
    const int      full  = pic->ctrl.hme_search_method == 1;
    const uint8_t *src   = orc_block_ptr(cur, ox, oy);
    const uint8_t *rwin  = orc_block_ptr(ref, ox + org_x, oy + org_y);
    orc_sad_loop(src, full ? cur->stride : cur->stride * 2, rwin, full ? ref->stride : ref->stride * 2,
                 full ? bh : bh >> 1, bw, sad, mvx, mvy, ref->stride, skip, sa_w, sa_h, ops);
    if (!full) *sad *= 2;
    (void)c;
}

/* ------------------------------------------------------------------------------------------------ HME stages */

/* init_me_hme_data, EbMotionEstimation.c:2868-2924 */
static void orc_init_b64(orc_ctx *c, const orc_pic *pic) {
    if (pic->ctrl.enable_hme_flag) {
        memset(c->lx, 0, sizeof(c->lx));
        memset(c->ly, 0, sizeof(c->ly));
    }
    memset(c->best_mv, 0, sizeof(c->best_mv));
    for (int l = 0; l < 2; ++l)
        for (int r = 0; r < 4; ++r) {
            c->do_ref[l][r]  = 1;
            c->hme_sad[l][r] = 0xFFFFFFFFu;
            c->sr_div[l][r]  = 1;
            c->zz_sad[l][r]  = ~0u;
        }
}

static int orc_ref_searched(const orc_pic *pic, int l) { return pic->temporal_layer > 0 || l == 0; }

/* init_zz_sad / get_zz_sad, :2318-2333, :1720-1742 */
static void orc_zz_sad(orc_ctx *c, const orc_pic *pic, int ox, int oy) {
    const orc_plane *cur = &pic->cur->lvl[2];
    for (int l = 0; l < pic->n_list; ++l)
        for (int r = 0; r < pic->n_ref[l]; ++r) {
            if (!orc_ref_searched(pic, l)) continue;
            const orc_plane *ref = &pic->ref[l][r]->lvl[2];
            c->zz_sad[l][r] = orc_sad(orc_block_ptr(cur, ox, oy), cur->stride << 1, orc_block_ptr(ref, ox, oy),
                                      ref->stride << 1, c->block_h >> 1, c->block_w) << 1;
            c->ops[0] += (uint64_t)(c->block_h >> 1) * c->block_w;
        }
}

/* check_prehme_early_exit :1746-1770 */
static int orc_prehme_early_exit(orc_ctx *c, const orc_pic *pic, int l, int r, int s) {
    orc_search_info *p = &c->pre[l][r][s];
    if (pic->ctrl.me_early_exit_th && c->zz_sad[l][r] < pic->ctrl.me_early_exit_th) {
        p->x = p->y = 0;
        p->sad      = 0;
        return 1;
    }
    if (pic->ctrl.prehme_l1_early_exit && l == 1) {
        const orc_search_info *q = &c->pre[0][r][s];
        if (q->sad < 32 * 32 || (ORC_ABS(q->x) < 16 && ORC_ABS(q->y) < 16)) {
            p->x   = (int16_t)-q->x;
            p->y   = (int16_t)-q->y;
            p->sad = q->sad;
            return 1;
        }
    }
    return 0;
}

/* prehme_b64 :1773-1823 + prehme_core :1621-1719 */
static void orc_prehme(orc_ctx *c, const orc_pic *pic, int b64x, int b64y) {
    const orc_plane *cur = &pic->cur->lvl[0];
    const int        ox = (int16_t)b64x >> 2, oy = (int16_t)b64y >> 2;
    for (int l = 0; l < pic->n_list; ++l)
        for (int r = 0; r < pic->n_ref[l]; ++r) {
            if (!orc_ref_searched(pic, l)) {
                for (int s = 0; s < 2; ++s) {
                    c->pre[1][r][s].x   = (int16_t)-c->pre[0][r][s].x;
                    c->pre[1][r][s].y   = (int16_t)-c->pre[0][r][s].y;
                    c->pre[1][r][s].sad = c->pre[0][r][s].sad;
                }
                continue;
            }
            const orc_plane *ref    = &pic->ref[l][r]->lvl[0];
            const uint32_t   factor = orc_scaled_dist(orc_dist(pic, l, r));
            for (int s = 0; s < 2; ++s) {
                if (orc_prehme_early_exit(c, pic, l, r, s)) continue;
                const svtme_sa_minmax *cfg = &pic->ctrl.prehme_sa_cfg[s];
                int16_t w   = (int16_t)(uint16_t)ORC_MIN(cfg->sa_min.width * factor, (uint32_t)cfg->sa_max.width);
                int16_t h   = (int16_t)(uint16_t)ORC_MIN(cfg->sa_min.height * factor, (uint32_t)cfg->sa_max.height);
                int16_t org_x = (int16_t)-(int16_t)(w >> 1);
                int16_t org_y = (int16_t)-(int16_t)(h >> 1);
                orc_clip_axis(ox, &org_x, &w, -((int)ref->pad - 1), (int)ref->width);
                orc_clip_axis(oy, &org_y, &h, -((int)ref->pad - 1), (int)ref->height);
                orc_search_info *p = &c->pre[l][r][s];
                orc_level_search(c, pic, cur, ref, ox, oy, c->block_w >> 2, c->block_h >> 2, org_x, org_y, w, h,
                                 pic->ctrl.prehme_skip_search_line, &p->sad, &p->x, &p->y, &c->ops[1]);
                p->x = (int16_t)((p->x + org_x) * 4);
                p->y = (int16_t)((p->y + org_y) * 4);
            }
        }
}

/* get_hme_l0_search_area :1827-1894. Rescales c->hme_l0_sa in place, exactly like the reference rescales
 * context_ptr->hme_l0_sa; the caller restores it (:1960-1965). */
static void orc_hme_l0_area(orc_ctx *c, const orc_pic *pic, int l, int r, uint16_t dist, int16_t *sa_w, int16_t *sa_h) {
    const svtme_controls *k = &pic->ctrl;
    if (k->enable_me_sr_adjustment && k->distance_based_hme_resizing) {
        int is_hor = 1, is_ver = 1, is_still = 0;
        if (k->reduce_hme_l0_sr_th_min && k->reduce_hme_l0_sr_th_max && (l || r)) {
            const int mvx = c->lx[0][0][0][0][0], mvy = c->ly[0][0][0][0][0];
            is_ver   = ORC_ABS(mvx) < k->reduce_hme_l0_sr_th_min && ORC_ABS(mvy) > k->reduce_hme_l0_sr_th_max;
            is_hor   = ORC_ABS(mvx) > k->reduce_hme_l0_sr_th_max && ORC_ABS(mvy) < k->reduce_hme_l0_sr_th_min;
            is_still = ORC_ABS(mvx) < k->reduce_hme_l0_sr_th_min * 3 && ORC_ABS(mvy) < k->reduce_hme_l0_sr_th_min * 3;
        }
        uint8_t x_off = is_hor ? 1 : 2, y_off = is_ver ? 1 : 2;
        if (k->enable_me_sr_adjustment == 2 && is_still) x_off = y_off = 4;
        c->hme_l0_sa.sa_min.width  = (uint16_t)(c->hme_l0_sa.sa_min.width / (x_off + r));
        c->hme_l0_sa.sa_min.height = (uint16_t)(c->hme_l0_sa.sa_min.height / (y_off + r));
        c->hme_l0_sa.sa_max.width  = (uint16_t)(c->hme_l0_sa.sa_max.width / (x_off + r));
        c->hme_l0_sa.sa_max.height = (uint16_t)(c->hme_l0_sa.sa_max.height / (y_off + r));
    }
    const int32_t factor = orc_scaled_dist(dist);
    int16_t       w      = (int16_t)(c->hme_l0_sa.sa_min.width / k->num_hme_sa_w);
    w = (int16_t)ORC_MIN(((w * factor) + 15) & ~0x0F, ((c->hme_l0_sa.sa_max.width / k->num_hme_sa_w) + 15) & ~0x0F);
    int16_t h = (int16_t)(c->hme_l0_sa.sa_min.height / k->num_hme_sa_h);
    h         = (int16_t)ORC_MIN(h * factor, c->hme_l0_sa.sa_max.height / k->num_hme_sa_h);
    *sa_w     = w;
    *sa_h     = h;
}

/* hme_level_0 :847-947 / hme_level_1 :950-1049 / hme_level_2 :1052-1140: one region's search. `scale` is the
 * factor that brings the level's vectors to the next level's units (4, 2, 1). Level 0 places the window by region
 * index; levels 1 and 2 centre it on the previous level's vector. */
static void orc_hme_region(orc_ctx *c, const orc_pic *pic, int level, int l, int r, int sr_w, int sr_h, int b64x,
                           int b64y, int16_t sa_w, int16_t sa_h, int16_t cen_x, int16_t cen_y) {
    const orc_plane *cur   = &pic->cur->lvl[level];
    const orc_plane *ref   = &pic->ref[l][r]->lvl[level];
    const int        shift = 2 - level;
    const int        ox = (int16_t)b64x >> shift, oy = (int16_t)b64y >> shift;
    const int        pad = level == 2 ? 63 : (int)ref->pad - 1;
    int16_t          w   = (int16_t)((sa_w + 7) & ~0x07);
    int16_t          h   = sa_h;
    int16_t          org_x, org_y;
    if (level == 0) {
        org_x = (int16_t)(-(int16_t)((w * pic->ctrl.num_hme_sa_w) >> 1) + (int16_t)(w * sr_w));
        org_y = (int16_t)(-(int16_t)((h * pic->ctrl.num_hme_sa_h) >> 1) + (int16_t)(h * sr_h));
    } else {
        org_x = (int16_t)(-(w >> 1) + cen_x);
        org_y = (int16_t)(-(h >> 1) + cen_y);
    }
    orc_clip_axis(ox, &org_x, &w, -pad, (int)ref->width);
    w = (int16_t)(w < 8 ? w : (w & ~0x07));
    orc_clip_axis(oy, &org_y, &h, -pad, (int)ref->height);
    uint64_t *sad = &c->lsad[level][l][r][sr_w][sr_h];
    int16_t  *mx = &c->lx[level][l][r][sr_w][sr_h], *my = &c->ly[level][l][r][sr_w][sr_h];
    orc_level_search(c, pic, cur, ref, ox, oy, c->block_w >> shift, c->block_h >> shift, org_x, org_y, w, h, 0, sad,
                     mx, my, &c->ops[2 + level]);
    const int scale = level == 0 ? 4 : (level == 1 ? 2 : 1);
    *mx = (int16_t)((*mx + org_x) * scale);
    *my = (int16_t)((*my + org_y) * scale);
}

static void orc_zero_level(orc_ctx *c, const orc_pic *pic, int level, int l, int r) {
    for (int y = 0; y < pic->ctrl.num_hme_sa_h; ++y)
        for (int x = 0; x < pic->ctrl.num_hme_sa_w; ++x) {
            c->lx[level][l][r][x][y] = c->ly[level][l][r][x][y] = 0;
            c->lsad[level][l][r][x][y]                          = 0;
        }
}

/* hme_level0_b64 :1899-2004 */
static void orc_hme_l0(orc_ctx *c, const orc_pic *pic, int b64x, int b64y) {
    const svtme_controls *k = &pic->ctrl;
    c->hme_l0_sa            = k->hme_l0_sa;
    for (int l = 0; l < pic->n_list; ++l)
        for (int r = 0; r < pic->n_ref[l]; ++r) {
            if (k->me_early_exit_th && c->zz_sad[l][r] < (k->me_early_exit_th >> 2)) {
                orc_zero_level(c, pic, 0, l, r);
                continue;
            }
            if (!orc_ref_searched(pic, l)) continue;
            int16_t sa_w = 0, sa_h = 0;
            orc_hme_l0_area(c, pic, l, r, orc_dist(pic, l, r), &sa_w, &sa_h);
            for (int sh = 0; sh < k->num_hme_sa_h; ++sh)
                for (int sw = 0; sw < k->num_hme_sa_w; ++sw) orc_hme_region(c, pic, 0, l, r, sw, sh, b64x, b64y, sa_w, sa_h, 0, 0);
            if (k->enable_me_sr_adjustment && k->distance_based_hme_resizing) c->hme_l0_sa = k->hme_l0_sa;
            if (k->prehme_enable) {
                /* the worst region (strict >, h outer / w inner) gives way to the better pre-HME candidate */
                uint64_t max_sad = 0;
                int      mw = 0, mh = 0;
                for (int sh = 0; sh < k->num_hme_sa_h; ++sh)
                    for (int sw = 0; sw < k->num_hme_sa_w; ++sw)
                        if (c->lsad[0][l][r][sw][sh] > max_sad) {
                            max_sad = c->lsad[0][l][r][sw][sh];
                            mw = sw, mh = sh;
                        }
                const int s = c->pre[l][r][0].sad <= c->pre[l][r][1].sad ? 0 : 1;
                if (c->pre[l][r][s].sad < c->lsad[0][l][r][mw][mh]) {
                    c->lsad[0][l][r][mw][mh] = c->pre[l][r][s].sad;
                    c->lx[0][l][r][mw][mh]   = c->pre[l][r][s].x;
                    c->ly[0][l][r][mw][mh]   = c->pre[l][r][s].y;
                }
            }
        }
}

/* hme_level1_b64 :2009-2068 */
static void orc_hme_l1(orc_ctx *c, const orc_pic *pic, int b64x, int b64y) {
    const svtme_controls *k = &pic->ctrl;
    for (int l = 0; l < pic->n_list; ++l)
        for (int r = 0; r < pic->n_ref[l]; ++r) {
            if (!orc_ref_searched(pic, l)) continue;
            if (k->me_early_exit_th && c->zz_sad[l][r] < (k->me_early_exit_th >> 2)) {
                orc_zero_level(c, pic, 1, l, r);
                continue;
            }
            for (int sh = 0; sh < k->num_hme_sa_h; ++sh)
                for (int sw = 0; sw < k->num_hme_sa_w; ++sw)
                    orc_hme_region(c, pic, 1, l, r, sw, sh, b64x, b64y, (int16_t)k->hme_l1_sa.width,
                                   (int16_t)k->hme_l1_sa.height, (int16_t)(c->lx[0][l][r][sw][sh] >> 1),
                                   (int16_t)(c->ly[0][l][r][sw][sh] >> 1));
        }
}

/* hme_level2_b64 :2073-2112 (no early exit at this level) */
static void orc_hme_l2(orc_ctx *c, const orc_pic *pic, int b64x, int b64y) {
    const svtme_controls *k = &pic->ctrl;
    for (int l = 0; l < pic->n_list; ++l)
        for (int r = 0; r < pic->n_ref[l]; ++r) {
            if (!orc_ref_searched(pic, l)) continue;
            for (int sh = 0; sh < k->num_hme_sa_h; ++sh)
                for (int sw = 0; sw < k->num_hme_sa_w; ++sw)
                    orc_hme_region(c, pic, 2, l, r, sw, sh, b64x, b64y, (int16_t)k->hme_l2_sa.width,
                                   (int16_t)k->hme_l2_sa.height, c->lx[1][l][r][sw][sh], c->ly[1][l][r][sw][sh]);
        }
}

/* set_final_seach_centre_sb :2117-2316. The running centre/SAD are NOT reset per reference: a base-layer
 * list-1 reference (centre forced to 0,0) and any reference visited while HME is off inherit the previous
 * reference's SAD. */
static void orc_final_centre(orc_ctx *c, const orc_pic *pic) {
    const svtme_controls *k = &pic->ctrl;
    int16_t  hx = 0, hy = 0, cx = 0, cy = 0;
    uint64_t hsad = 0;
    for (int l = 0; l < pic->n_list; ++l)
        for (int r = 0; r < pic->n_ref[l]; ++r) {
            if (orc_ref_searched(pic, l)) {
                if (k->enable_hme_flag) {
                    int level = -1;
                    if (k->enable_hme_level0_flag && !k->enable_hme_level1_flag && !k->enable_hme_level2_flag) level = 0;
                    if (k->enable_hme_level1_flag && !k->enable_hme_level2_flag) level = 1;
                    if (k->enable_hme_level2_flag) level = 2;
                    if (level >= 0) {
                        hx   = c->lx[level][l][r][0][0];
                        hy   = c->ly[level][l][r][0][0];
                        hsad = c->lsad[level][l][r][0][0];
                        for (int sh = 0; sh < k->num_hme_sa_h; ++sh)
                            for (int sw = (sh == 0 ? 1 : 0); sw < k->num_hme_sa_w; ++sw)
                                if (c->lsad[level][l][r][sw][sh] < hsad) {
                                    hx   = c->lx[level][l][r][sw][sh];
                                    hy   = c->ly[level][l][r][sw][sh];
                                    hsad = c->lsad[level][l][r][sw][sh];
                                }
                    }
                    cx = hx;
                    cy = hy;
                }
            } else {
                cx = cy = 0;
            }
            c->sc_x[l][r]    = cx;
            c->sc_y[l][r]    = cy;
            c->hme_sad[l][r] = hsad;
            c->do_ref[l][r]  = 1;
        }
}

/* hme_prune_ref_and_adjust_sr :2373-2410: over all 2x4 entries, searched or not. */
static void orc_hme_prune(orc_ctx *c, const orc_pic *pic) {
    const svtme_controls *k    = &pic->ctrl;
    uint64_t              best = ~(uint64_t)0;
    for (int l = 0; l < 2; ++l)
        for (int r = 0; r < 4; ++r)
            if (c->hme_sad[l][r] < best) best = c->hme_sad[l][r];
    const uint16_t th = k->prune_ref_if_hme_sad_dev_bigger_than_th;
    for (int l = 0; l < 2; ++l)
        for (int r = 0; r < 4; ++r) {
            if (k->enable_me_hme_ref_pruning && (!k->protect_closest_refs || r > 0) && th != 0xFFFF &&
                (c->hme_sad[l][r] - best) * 100 > (uint64_t)th * best)
                c->do_ref[l][r] = 0;
            if (k->enable_me_sr_adjustment) {
                if (ORC_ABS(c->sc_x[l][r]) <= k->reduce_me_sr_based_on_mv_length_th &&
                    ORC_ABS(c->sc_y[l][r]) <= k->reduce_me_sr_based_on_mv_length_th &&
                    c->hme_sad[l][r] < k->stationary_hme_sad_abs_th)
                    c->sr_div[l][r] = k->stationary_me_sr_divisor;
                else if (c->hme_sad[l][r] < k->reduce_me_sr_based_on_hme_sad_abs_th)
                    c->sr_div[l][r] = k->me_sr_divisor_for_low_hme_sad;
            }
        }
}

/* ------------------------------------------------------------------------------------------- integer search */

/* z-order index of the 16x16 block at raster (row, col): `offsets` of svt_ext_all_sad_calculation_8x8_16x16_c,
 * EbMotionEstimation.c:366 (== tab16x16, EbMotionEstimation.h:108). */
static int orc_z16(int row, int col) { return ((row >> 1) << 3) | ((col >> 1) << 2) | ((row & 1) << 1) | (col & 1); }

/* One search position: the 64 8x8 SADs (rows 0,2,4,6 doubled when sub_sad), their 16x16 / 32x32 / 64x64 sums, and
 * the strict-< update of all 85 running minima. Restates svt_ext_all_sad_calculation_8x8_16x16_c :359-388 with
 * svt_ext_eight_sad_calculation_8x8_16x16 :234-357, svt_ext_eight_sad_calculation_32x32_64x64_c :395-451 and the
 * single-position forms :122-229; all of them reduce to this per position, visited in raster order. */
static void orc_me_position(uint32_t *best_sad, uint32_t *best_mv, const uint8_t *src, uint32_t src_stride,
                            const uint8_t *ref, uint32_t ref_stride, uint32_t mv, int sub_sad) {
    uint32_t s16[16], s32[4] = {0, 0, 0, 0}, s64 = 0;
    for (int by = 0; by < 4; ++by)
        for (int bx = 0; bx < 4; ++bx) {
            const int z = orc_z16(by, bx);
            uint32_t  sum = 0;
            for (int q = 0; q < 4; ++q) { /* 8x8 quadrants: TL, TR, BL, BR */
                const uint8_t *s = src + (size_t)(by * 16 + (q >> 1) * 8) * src_stride + bx * 16 + (q & 1) * 8;
                const uint8_t *f = ref + (size_t)(by * 16 + (q >> 1) * 8) * ref_stride + bx * 16 + (q & 1) * 8;
                uint32_t       sad;
                if (sub_sad) sad = orc_sad(s, src_stride * 2, f, ref_stride * 2, 4, 8) << 1;
                else sad = orc_sad(s, src_stride, f, ref_stride, 8, 8);
                const int idx = 21 + z * 4 + q;
                if (sad < best_sad[idx]) {
                    best_sad[idx] = sad;
                    best_mv[idx]  = mv;
                }
                sum += sad;
            }
            s16[z] = sum;
            if (sum < best_sad[5 + z]) {
                best_sad[5 + z] = sum;
                best_mv[5 + z]  = mv;
            }
        }
    for (int i = 0; i < 4; ++i) {
        s32[i] = s16[4 * i] + s16[4 * i + 1] + s16[4 * i + 2] + s16[4 * i + 3];
        if (s32[i] < best_sad[1 + i]) {
            best_sad[1 + i] = s32[i];
            best_mv[1 + i]  = mv;
        }
        s64 += s32[i];
    }
    if (s64 < best_sad[0]) {
        best_sad[0] = s64;
        best_mv[0]  = mv;
    }
}

/* check_00_center :1170-1243. Returns the SAD at the (clipped) HME centre and resets the centre to (0,0) when the zero
 * vector costs no more; for ME_MCTF a centre SAD above TF_HME_MV_SAD_TH (512) is weighted by TF_HME_MV_COST_WEIGHT
 * (125 %) first (:37-38, :1231-1241; costs are SAD << COST_PRECISION, 8 bits). */
static uint32_t orc_check_00(orc_ctx *c, const orc_pic *pic, int l, int r, int ox, int oy, int16_t *cx, int16_t *cy) {
    const orc_plane *cur = &pic->cur->lvl[2];
    const orc_plane *ref = &pic->ref[l][r]->lvl[2];
    const uint32_t   bh = c->block_h >> 1, bw = c->block_w;
    uint32_t         zero_sad;
    if (pic->ctrl.me_early_exit_th) zero_sad = c->zz_sad[l][r];
    else {
        zero_sad = orc_sad(orc_block_ptr(cur, ox, oy), cur->stride << 1, orc_block_ptr(ref, ox, oy), ref->stride << 1, bh, bw);
        c->ops[0] += (uint64_t)bh * bw;
    }
    zero_sad <<= 1;
    if (ox + *cx < -63) *cx = (int16_t)(-63 - ox);
    if (ox + *cx > (int)ref->width - 1) *cx = (int16_t)(*cx - ((ox + *cx) - ((int)ref->width - 1)));
    if (oy + *cy < -63) *cy = (int16_t)(-63 - oy);
    if (oy + *cy > (int)ref->height - 1) *cy = (int16_t)(*cy - ((oy + *cy) - ((int)ref->height - 1)));
    uint32_t hme_sad = orc_sad(orc_block_ptr(cur, ox, oy), cur->stride << 1, orc_block_ptr(ref, ox + *cx, oy + *cy),
                               ref->stride << 1, bh, bw) << 1;
    c->ops[0] += (uint64_t)bh * bw;
    uint64_t zero_cost = (uint64_t)zero_sad << 8, hme_cost = (uint64_t)hme_sad << 8;
    if (pic->me_type == 1 && hme_sad > 512) hme_cost = hme_cost * 125 / 100;
    if (zero_cost <= hme_cost) *cx = *cy = 0; /* MIN(zero_cost, hme_cost) == zero_cost */
    return hme_sad;
}

/* integer_search_b64 :1286-1570 with open_loop_me_fullpel_search_sblock :808-844 */
static void orc_integer_search(orc_ctx *c, const orc_pic *pic, uint32_t b64_index, int ox, int oy) {
    const svtme_controls *k   = &pic->ctrl;
    const orc_plane      *cur = &pic->cur->lvl[2];
    for (int l = 0; l < pic->n_list; ++l)
        for (int r = 0; r < pic->n_ref[l]; ++r) {
            if (!c->do_ref[l][r]) continue;
            const orc_plane *ref = &pic->ref[l][r]->lvl[2];
            int16_t cx = c->sc_x[l][r], cy = c->sc_y[l][r];
            /* the growth of the area with distance is slowed down except for ME_MCTF, :1338-1341 */
            const uint16_t dist = pic->me_type == 1 ? orc_dist(pic, l, r) : orc_scaled_dist(orc_dist(pic, l, r));
            int16_t w = (int16_t)ORC_MIN((int16_t)k->me_sa.sa_min.width * dist, k->me_sa.sa_max.width);
            int16_t h = (int16_t)ORC_MIN((int16_t)k->me_sa.sa_min.height * dist, k->me_sa.sa_max.height);
            w = (int16_t)((ORC_MAX(1, (int)(w / c->sr_div[l][r])) + 7) & ~0x07);
            h = (int16_t)ORC_MAX(3, (int)(h / c->sr_div[l][r]));
            const int16_t h0 = h, w0 = w;
            uint64_t      best_hme_sad = ~(uint64_t)0;
            if (k->me_early_exit_th) {
                if (c->zz_sad[l][r] < k->me_early_exit_th / 6) w = h = 1;
            } else {
                int accurate = 1;
                /* :1363-1364: the first pass may bypass the zero-vector check */
                if (!(pic->me_type == 4 && pic->bypass_zz_check) && (cx != 0 || cy != 0) && pic->is_ref) {
                    best_hme_sad = orc_check_00(c, pic, l, r, ox, oy, &cx, &cy);
                    if (cx == 0 && cy == 0) accurate = 0;
                }
                if (k->enable_me_sr_adjustment == 2) {
                    if ((accurate && best_hme_sad < 24 * 24) || (pic->is_ref && c->hme_sad[l][r] < 24 * 24)) h = (int16_t)(h / 2);
                    if ((l || r) && c->best_sad[0][0][0] < 5000 && h == h0 && w == w0) {
                        h = (int16_t)(h >> 1);
                        w = (int16_t)(w >> 1);
                    }
                }
            }
            int16_t org_x = (int16_t)(cx - (w >> 1)), org_y = (int16_t)(cy - (h >> 1));
            /* restricted motion vectors: the block's tile instead of the padded picture bounds the window, :1405-1498 */
            int lo_x = -63, lo_y = -63, hi_x = (int)pic->width, hi_y = (int)pic->height;
            if (pic->restricted_motion_vector) {
                lo_x = lo_y = 0;
                if (pic->tile_rects && (pic->me_type == 1 || pic->me_type == 4)) {
                    const svtme_tile_rect *t = &pic->tile_rects[b64_index];
                    lo_x = (int)t->start_x, lo_y = (int)t->start_y, hi_x = (int)t->end_x, hi_y = (int)t->end_y;
                }
            }
            orc_clip_axis(ox, &org_x, &w, lo_x, hi_x);
            w = (int16_t)(w < 8 ? w : (w & ~0x07));
            orc_clip_axis(oy, &org_y, &h, lo_y, hi_y);

            uint32_t *bs = c->best_sad[l][r], *bm = c->best_mv[l][r];
            for (int i = 0; i < 85; ++i) bs[i] = SVTME_MAX_SAD_VALUE; /* svt_initialize_buffer_32bits :1542 */
            const uint8_t *src = orc_block_ptr(cur, ox, oy);
            const int      sub = k->me_search_method == 0;
            for (int y = 0; y < h; ++y)
                for (int x = 0; x < w; ++x) {
                    const int      px = x + org_x, py = y + org_y;
                    const uint32_t mv = ((uint32_t)(uint16_t)py << 18) | (uint16_t)((uint16_t)px << 2); /* :474-476 */
                    orc_me_position(bs, bm, src, cur->stride, orc_block_ptr(ref, ox + px, oy + py), ref->stride, mv, sub);
                }
            c->ops[5] += (uint64_t)(w > 0 ? w : 0) * (h > 0 ? h : 0) * (sub ? 2048 : 4096);
        }
}

/* ------------------------------------------------------------------------------------------------ host tail */

static const uint8_t orc_z_to_raster[85] = {/* EbMotionEstimation.c:2412-2429 */
    0,  1,  2,  3,  4,  5,  6,  9,  10, 7,  8,  11, 12, 13, 14, 17, 18, 15, 16, 19, 20, 21, 22, 29, 30, 23, 24, 31, 32,
    37, 38, 45, 46, 39, 40, 47, 48, 25, 26, 33, 34, 27, 28, 35, 36, 41, 42, 49, 50, 43, 44, 51, 52, 53, 54, 61, 62, 55,
    56, 63, 64, 69, 70, 77, 78, 71, 72, 79, 80, 57, 58, 65, 66, 59, 60, 67, 68, 73, 74, 81, 82, 75, 76, 83, 84};

/* me_prune_ref :1576-1618 */
static void orc_me_prune(orc_ctx *c, const orc_pic *pic) {
    const svtme_controls *k = &pic->ctrl;
    for (int l = 0; l < pic->n_list; ++l)
        for (int r = 0; r < pic->n_ref[l]; ++r) {
            c->hme_sad[l][r] = 0;
            if (!c->do_ref[l][r]) {
                c->hme_sad[l][r] = (uint64_t)SVTME_MAX_SAD_VALUE * 64;
                continue;
            }
            for (int i = 0; i < 64; ++i) c->hme_sad[l][r] += c->best_sad[l][r][21 + i];
        }
    uint64_t best = ~(uint64_t)0;
    for (int l = 0; l < 2; ++l)
        for (int r = 0; r < 4; ++r)
            if (c->hme_sad[l][r] < best) best = c->hme_sad[l][r];
    const uint16_t th = k->prune_ref_if_me_sad_dev_bigger_than_th;
    for (int l = 0; l < 2; ++l)
        for (int r = 0; r < 4; ++r)
            if (k->enable_me_hme_ref_pruning && (!k->protect_closest_refs || r > 0) && th != 0xFFFF &&
                (c->hme_sad[l][r] - best) * 100 > (uint64_t)th * best)
                c->do_ref[l][r] = 0;
}

/* MeCandidate bit-field, EbMotionEstimationLcuResults.h:29-35 (gcc little-endian layout). The literal 24 the
 * reference assigns to the 1-bit list fields truncates to 0. */
static uint8_t orc_cand(int direction, int idx0, int idx1, int list0, int list1) {
    return (uint8_t)((direction & 3) | ((idx0 & 3) << 2) | ((idx1 & 3) << 4) | ((list0 & 1) << 6) | ((list1 & 1) << 7));
}

static int orc_use_me_pu(const orc_pic *pic, int n_idx) {
    return pic->enable_me_16x16 ? (pic->enable_me_8x8 || n_idx < 21) : n_idx < 5;
}

/* construct_me_candidate_array_mrp_off :2430-2543 */
static void orc_cands_mrp_off(orc_ctx *c, const orc_pic *pic) {
    uint32_t n_list = pic->n_list;
    uint8_t  org[2];
    org[0] = c->do_ref[0][0];
    org[1] = n_list == 1 ? 0 : c->do_ref[1][0];
    if (n_list < 2 || !c->do_ref[1][0]) n_list = 1;
    const uint32_t prune_th = (org[0] && org[1]) ? (uint32_t)pic->ctrl.prune_me_candidates_th : 0;
    const int      n_pus    = pic->enable_me_16x16 ? (pic->enable_me_8x8 ? 85 : 21) : 5;
    memset(c->total_cand, 1, (size_t)n_pus);
    for (int n = 0; n < 85; ++n) {
        const int pu  = orc_z_to_raster[n];
        const int use = orc_use_me_pu(pic, n);
        uint8_t  *ca  = use ? &c->cand_array[pu * pic->max_cand] : NULL;
        uint8_t   dr[2] = {org[0], org[1]};
        int       off   = 0;
        const uint32_t best = prune_th > 0 ? ORC_MIN(c->best_sad[0][0][n], c->best_sad[1][0][n]) : ~0u;
        int min_list = -1;
        if (pic->ctrl.use_best_unipred_cand_only && dr[0] && dr[1]) min_list = c->best_sad[0][0][n] < c->best_sad[1][0][n] ? 0 : 1;
        for (uint32_t l = 0; l < n_list && (use || off == 0); ++l) {
            if (!dr[l]) continue;
            if (prune_th > 0) {
                uint32_t d = (c->best_sad[l][0][n] - best) * 100;
                if (d > best * prune_th) {
                    dr[l] = 0;
                    continue;
                }
            }
            if (min_list != -1 && min_list != (int)l) {
                if (use) c->mv_array[pu * pic->max_refs + (l ? pic->max_l0 : 0)] = c->best_mv[l][0][n];
                continue;
            }
            if (off == 0) c->me_distortion[pu] = c->best_sad[l][0][n];
            if (use) {
                ca[off] = orc_cand((int)l, 0, 0, 0, l == 1 ? 1 : 0);
                c->mv_array[pu * pic->max_refs + (l ? pic->max_l0 : 0)] = c->best_mv[l][0][n];
            }
            off++;
        }
        if (dr[0] && dr[1] && use) {
            ca[off]           = orc_cand(2, 0, 0, 0, 1);
            c->total_cand[pu] = (uint8_t)(off + 1);
        }
    }
}

/* construct_me_candidate_array :2544-2685 */
static void orc_cands(orc_ctx *c, const orc_pic *pic) {
    const uint32_t n_list   = pic->n_list;
    const uint32_t prune_th = (uint32_t)pic->ctrl.prune_me_candidates_th;
    for (int n = 0; n < 85; ++n) {
        const int pu  = n > 4 ? orc_z_to_raster[n] : n;
        const int use = orc_use_me_pu(pic, n);
        uint8_t  *ca  = use ? &c->cand_array[pu * pic->max_cand] : NULL;
        uint8_t   dr[2][4];
        memset(dr, 0, sizeof(dr));
        int      off  = 0;
        uint32_t best = ~0u;
        if (prune_th > 0)
            for (uint32_t l = 0; l < n_list; ++l)
                for (int r = 0; r < pic->n_ref[l]; ++r)
                    if (c->do_ref[l][r] && c->best_sad[l][r][n] < best) best = c->best_sad[l][r][n];
        for (uint32_t l = 0; l < n_list && (use || off == 0); ++l)
            for (int r = 0; r < pic->n_ref[l] && (use || off == 0); ++r) {
                dr[l][r] = c->do_ref[l][r];
                if (!c->do_ref[l][r]) continue;
                if (prune_th > 0) {
                    uint32_t d = (c->best_sad[l][r][n] - best) * 100;
                    if (d > best * prune_th) {
                        dr[l][r] = 0;
                        continue;
                    }
                }
                if (off == 0) c->me_distortion[pu] = c->best_sad[l][r][n];
                if (use) {
                    ca[off] = orc_cand((int)l, r, r, 0, l == 1 ? 1 : 0);
                    c->mv_array[pu * pic->max_refs + (l ? pic->max_l0 : 0) + r] = c->best_mv[l][r][n];
                }
                off++;
            }
        if (n_list == 2 && use) {
            for (int a = 0; a < pic->n_ref[0]; ++a)
                for (int b = 0; b < pic->n_ref[1]; ++b)
                    if (dr[0][a] && dr[1][b]) ca[off++] = orc_cand(2, a, b, 0, 1);
            for (int a = 1; a < pic->n_ref[0]; ++a)
                if (dr[0][0] && dr[0][a]) ca[off++] = orc_cand(2, 0, a, 0, 0);
            if (pic->n_ref[1] == 3 && dr[1][0] && dr[1][2]) ca[off++] = orc_cand(2, 0, 2, 1, 1);
        }
        if (use) c->total_cand[pu] = (uint8_t)off;
    }
}

/* compute_distortion :2814-2865 */
static void orc_distortion(orc_ctx *c, const orc_pic *pic, uint32_t sb_w, uint32_t sb_h) {
    uint32_t d64 = c->me_distortion[0], d32 = 0, d16 = 0, d8 = 0;
    for (int i = 0; i < 4; ++i) d32 += c->me_distortion[1 + i];
    for (int i = 0; i < 16; ++i) d16 += c->me_distortion[5 + i];
    for (int i = 0; i < 64; ++i) d8 += c->me_distortion[21 + i];
    const uint64_t mean = d8 / 64;
    uint64_t       ssq  = 0;
    for (int i = 0; i < 64; ++i) {
        const int64_t diff = (int64_t)c->me_distortion[21 + i] - (int64_t)mean;
        ssq += (uint64_t)(diff * diff);
    }
    const uint32_t pix = sb_w * sb_h, sz = 64 * 64, f = pic->ctrl.stat_factor;
    c->stat[0] = pic->input_resolution <= 2 ? d8 : d16; /* rc_me_distortion */
    c->stat[1] = (((d64 * sz) / pix) * f) / 100;
    c->stat[2] = (((d32 * sz) / pix) * f) / 100;
    c->stat[3] = (((d16 * sz) / pix) * f) / 100;
    c->stat[4] = (((d8 * sz) / pix) * f) / 100;
    c->stat[5] = (uint32_t)(ssq / 64);
}

/* perform_gm_detection :2688-2811. n_idx is a raster PU index; the reference uses it as is to index the z-ordered
 * p_sb_best_mv array. */
static void orc_gm_detection(orc_ctx *c, const orc_pic *pic) {
    uint64_t cnt[2][4][2][2];
    uint64_t stationary = 0, tot = 0;
    memset(cnt, 0, sizeof(cnt));
    const int low_res = pic->input_resolution <= 2;
    const int n       = low_res ? 64 : 16;
    for (int i = 0; i < n; ++i) {
        int idx = (low_res ? 21 : 5) + i;
        if (low_res && !pic->enable_me_8x8) {
            if (idx >= 21) idx = 5 + (((idx - 21) >> 4) << 2) + (((idx - 21) & 7) >> 1); /* me_idx_85_8x8_to_16x16_conversion */
            if (!pic->enable_me_16x16 && idx >= 5) idx = 1 + (((idx - 5) >> 3) << 1) + (((idx - 5) & 3) >> 1);
        } else if (!low_res && !pic->enable_me_16x16 && idx >= 5) {
            idx = 1 + (((idx - 5) >> 3) << 1) + (((idx - 5) & 3) >> 1); /* me_idx_16x16_to_parent_32x32_conversion */
        }
        const uint8_t cand = c->cand_array[idx * pic->max_cand];
        const int     dir  = cand & 3;
        const int     l    = (dir == 0 || dir == 2) ? ((cand >> 6) & 1) : ((cand >> 7) & 1);
        const int     r    = (dir == 0 || dir == 2) ? ((cand >> 2) & 3) : ((cand >> 4) & 3);
        int           th;
        if (low_res) {
            uint64_t a = pic->picture_number, b = pic->ref_picture_number[l][r];
            uint16_t dist = (uint16_t)ORC_ABS((int16_t)(ORC_MAX(a, b) - ORC_MIN(a, b)));
            th = pic->gm_use_distance_based_active_th ? ORC_MAX(dist >> 1, 4) : 4;
        } else {
            uint16_t dist = (uint16_t)ORC_ABS((int16_t)(pic->picture_number - pic->ref_picture_number[l][r]));
            th = pic->gm_use_distance_based_active_th ? ORC_MAX(dist * 16, 32) : 32;
        }
        const int mx = (int16_t)(c->best_mv[l][r][idx] & 0xFFFF), my = (int16_t)(c->best_mv[l][r][idx] >> 16);
        if (mx < -th) cnt[l][r][0][0]++;
        else if (mx > th) cnt[l][r][0][1]++;
        if (my < -th) cnt[l][r][1][0]++;
        else if (my > th) cnt[l][r][1][1]++;
        const int sth = low_res ? 0 : 4;
        if (ORC_ABS(mx) <= sth && ORC_ABS(my) <= sth) stationary++;
        tot++;
    }
    if (stationary > (tot * 5) / 100) c->stationary = 1;
    for (int l = 0; l < 2; ++l)
        for (int r = 0; r < 4; ++r)
            for (int k = 0; k < 4; ++k)
                if (cnt[l][r][k >> 1][k & 1] > tot / 2) c->allow_gm = 1;
}

/* ------------------------------------------------------------------------------------------------ per b64 */

/* motion_estimation_b64 :2930-3002 (ME_OPEN_LOOP and ME_MCTF) */
static void orc_tail(orc_ctx *c, const orc_pic *pic);
static void orc_dump(const orc_ctx *c, const orc_pic *pic, ref_b64_dump *out);

void orc_me_b64(orc_ctx *c, const orc_pic *pic, uint32_t b64_index, ref_b64_dump *out, int run_tail) {
    const svtme_controls *k     = &pic->ctrl;
    const uint32_t        b64_w = (pic->width + 63) / 64;
    const int             ox = (int)(b64_index % b64_w) * 64, oy = (int)(b64_index / b64_w) * 64;
    c->block_w = ORC_MIN(pic->width - (uint32_t)ox, 64u);
    c->block_h = ORC_MIN(pic->height - (uint32_t)oy, 64u);

    orc_init_b64(c, pic);
    /* hme_b64 :2337-2371 */
    if (k->me_early_exit_th) orc_zz_sad(c, pic, ox, oy);
    if (k->prehme_enable) orc_prehme(c, pic, ox, oy);
    if (k->enable_hme_flag) {
        if (k->enable_hme_level0_flag) orc_hme_l0(c, pic, ox, oy);
        if (k->enable_hme_level1_flag) orc_hme_l1(c, pic, ox, oy);
        if (k->enable_hme_level2_flag) orc_hme_l2(c, pic, ox, oy);
    }
    orc_final_centre(c, pic);
    c->tf_exit = 0;
    if (pic->me_type == 1 && c->hme_sad[0][0] < pic->tf_me_exit_th) { /* :2958-2962 */
        c->tf_exit = 1;
        memset(c->best_sad, 0, sizeof(c->best_sad)); /* not written by this block: dumped as zeros */
        orc_dump(c, pic, out);
        return;
    }
    const int prune_ref = k->enable_hme_flag && pic->me_type != 1;
    if (prune_ref && (k->enable_me_sr_adjustment || k->enable_me_hme_ref_pruning)) orc_hme_prune(c, pic);
    orc_integer_search(c, pic, b64_index, ox, oy);

    if (run_tail && pic->me_type != 1) orc_tail(c, pic); /* :2977 */
    orc_dump(c, pic, out);
}

/* motion_estimation_b64 :2975-2999: what stays on the host when a device does the searches */
static void orc_tail(orc_ctx *c, const orc_pic *pic) {
    const svtme_controls *k = &pic->ctrl;
    /* pa_me_data->me_results[b64] is a per-b64 buffer (zero-filled by the drivers), not thread state */
    memset(c->total_cand, 0, sizeof(c->total_cand));
    memset(c->mv_array, 0, sizeof(c->mv_array));
    memset(c->cand_array, 0, sizeof(c->cand_array));
    if (k->enable_hme_flag && k->enable_me_hme_ref_pruning) orc_me_prune(c, pic);
    if (pic->n_ref[0] == 1 && pic->n_ref[1] == 1) orc_cands_mrp_off(c, pic);
    else orc_cands(c, pic);
    if (pic->me_type != 4) orc_distortion(c, pic, c->block_w, c->block_h); /* :2990 */
    c->stationary = c->allow_gm = 0;
    if (pic->gm_enabled) orc_gm_detection(c, pic);
}

/* The host tail alone, on search results handed over in `in` (best SAD / MV, search centres, HME SADs, do_ref,
 * zero-MV SADs, divisors): what the encoder's ME thread does after unpacking a device result. */
void orc_tail_b64(orc_ctx *c, const orc_pic *pic, uint32_t b64_index, const ref_b64_dump *in, ref_b64_dump *out) {
    const uint32_t b64_w = (pic->width + 63) / 64;
    const int      ox = (int)(b64_index % b64_w) * 64, oy = (int)(b64_index / b64_w) * 64;
    c->block_w = ORC_MIN(pic->width - (uint32_t)ox, 64u);
    c->block_h = ORC_MIN(pic->height - (uint32_t)oy, 64u);
    orc_init_b64(c, pic);
    memcpy(c->best_sad, in->best_sad, sizeof(in->best_sad));
    memcpy(c->best_mv, in->best_mv, sizeof(in->best_mv));
    memcpy(c->sc_x, in->hme_sc_x, sizeof(in->hme_sc_x));
    memcpy(c->sc_y, in->hme_sc_y, sizeof(in->hme_sc_y));
    memcpy(c->hme_sad, in->hme_sad, sizeof(in->hme_sad));
    memcpy(c->do_ref, in->do_ref, sizeof(in->do_ref));
    memcpy(c->zz_sad, in->zz_sad, sizeof(in->zz_sad));
    memcpy(c->sr_div, in->sr_divisor, sizeof(in->sr_divisor));
    orc_tail(c, pic);
    orc_dump(c, pic, out);
}

static void orc_dump(const orc_ctx *c, const orc_pic *pic, ref_b64_dump *out) {
    if (!out) return;
    memset(out, 0, sizeof(*out));
    memcpy(out->best_sad, c->best_sad, sizeof(out->best_sad));
    memcpy(out->best_mv, c->best_mv, sizeof(out->best_mv));
    memcpy(out->hme_sc_x, c->sc_x, sizeof(out->hme_sc_x));
    memcpy(out->hme_sc_y, c->sc_y, sizeof(out->hme_sc_y));
    memcpy(out->hme_sad, c->hme_sad, sizeof(out->hme_sad));
    memcpy(out->do_ref, c->do_ref, sizeof(out->do_ref));
    memcpy(out->zz_sad, c->zz_sad, sizeof(out->zz_sad));
    memcpy(out->sr_divisor, c->sr_div, sizeof(out->sr_divisor));
    memcpy(out->l0_x, c->lx[0], sizeof(out->l0_x));
    memcpy(out->l0_y, c->ly[0], sizeof(out->l0_y));
    memcpy(out->l0_sad, c->lsad[0], sizeof(out->l0_sad));
    memcpy(out->l1_x, c->lx[1], sizeof(out->l1_x));
    memcpy(out->l1_y, c->ly[1], sizeof(out->l1_y));
    memcpy(out->l1_sad, c->lsad[1], sizeof(out->l1_sad));
    memcpy(out->l2_x, c->lx[2], sizeof(out->l2_x));
    memcpy(out->l2_y, c->ly[2], sizeof(out->l2_y));
    memcpy(out->l2_sad, c->lsad[2], sizeof(out->l2_sad));
    for (int l = 0; l < 2; ++l)
        for (int r = 0; r < 4; ++r)
            for (int s = 0; s < 2; ++s) {
                out->pre_x[l][r][s]   = c->pre[l][r][s].x;
                out->pre_y[l][r][s]   = c->pre[l][r][s].y;
                out->pre_sad[l][r][s] = c->pre[l][r][s].sad;
            }
    memcpy(out->me_distortion, c->me_distortion, sizeof(out->me_distortion));
    memcpy(out->total_me_candidate_index, c->total_cand, sizeof(out->total_me_candidate_index));
    memcpy(out->me_mv_array, c->mv_array, (size_t)85 * pic->max_refs * 4);
    memcpy(out->me_candidate_array, c->cand_array, (size_t)85 * pic->max_cand);
    out->rc_me_distortion     = c->stat[0];
    out->me_64x64_distortion  = c->stat[1];
    out->me_32x32_distortion  = c->stat[2];
    out->me_16x16_distortion  = c->stat[3];
    out->me_8x8_distortion    = c->stat[4];
    out->me_8x8_cost_variance = c->stat[5];
    out->stationary_block_present_sb = c->stationary;
    out->rc_me_allow_gm              = c->allow_gm;
    out->tf_me_exit                  = c->tf_exit;
}

/* ---------------------------------------------------------------------------------------------- per picture */

typedef struct orc_job {
    const orc_pic *pic;
    ref_b64_dump  *out;
    int            run_tail;
    uint32_t       row0, row1, b64_w;
    uint64_t       ops[6];
} orc_job;

static void *orc_worker(void *arg) {
    orc_job *j = (orc_job *)arg;
    orc_ctx *c = orc_ctx_new();
    for (uint32_t by = j->row0; by < j->row1; ++by)
        for (uint32_t bx = 0; bx < j->b64_w; ++bx) {
            uint32_t b = by * j->b64_w + bx;
            if (j->pic->b64_step > 1 && (bx % j->pic->b64_step != 0 || by % j->pic->b64_step != 0)) continue;
            orc_me_b64(c, j->pic, b, j->out ? &j->out[b] : NULL, j->run_tail);
        }
    orc_ctx_ops(c, j->ops);
    orc_ctx_free(c);
    return NULL;
}

void orc_me_picture(const orc_pic *pic, ref_b64_dump *out, int run_tail, int threads, uint64_t ops[6]) {
    const uint32_t b64_w = (pic->width + 63) / 64, b64_h = (pic->height + 63) / 64;
    if (threads < 1) threads = 1;
    if ((uint32_t)threads > b64_h) threads = (int)b64_h;
    orc_job   *jobs = (orc_job *)calloc((size_t)threads, sizeof(orc_job));
    pthread_t *tid  = (pthread_t *)calloc((size_t)threads, sizeof(pthread_t));
    for (int t = 0; t < threads; ++t) {
        jobs[t].pic = pic;
        jobs[t].out = out;
        jobs[t].run_tail = run_tail;
        jobs[t].b64_w = b64_w;
        jobs[t].row0 = (uint32_t)((uint64_t)b64_h * (uint32_t)t / (uint32_t)threads);
        jobs[t].row1 = (uint32_t)((uint64_t)b64_h * (uint32_t)(t + 1) / (uint32_t)threads);
        if (threads == 1) orc_worker(&jobs[t]);
        else pthread_create(&tid[t], NULL, orc_worker, &jobs[t]);
    }
    if (ops) memset(ops, 0, 6 * sizeof(uint64_t));
    for (int t = 0; t < threads; ++t) {
        if (threads > 1) pthread_join(tid[t], NULL);
        if (ops)
            for (int i = 0; i < 6; ++i) ops[i] += jobs[t].ops[i];
    }
    free(jobs);
    free(tid);
}
