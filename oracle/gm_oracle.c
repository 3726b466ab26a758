/*
 * gm_oracle.c - TEST INFRASTRUCTURE, not part of the product.
 *
 * Plain-C restatement of the data-parallel stages of the reference's global-motion estimation, the ones the GPU engine
 * takes over behind svtme_gm_* (include/svtme.h). Each function names the reference lines it follows. RANSAC, the
 * refinement loop and the parameter costs stay host code of the encoder and are not restated.
 *
 * Pinned against the real reference (oracle/_ref/ref_gm, built by oracle/Makefile from the reference's own sources):
 * tests/golden/gm_*.npz hold the corners, correspondences and every svt_av1_warp_error / svt_av1_frame_error call of
 * compute_global_motion runs, and tests/test_gm_oracle.py checks this file against them.
 */
#include "gm_oracle.h"

#include <stdlib.h>
#include <string.h>

/* ---------------------------------------------------------------------------------------------- FAST-9 corners
 * svt_av1_fast_corner_detect (corner_detect.c:19-32) = svt_aom_fast9_detect_nonmax (third_party/fastfeat/fast.c:6-21) with
 * barrier 18, cut to the first max_points corners in raster order.
 *
 * fast_9.c's generated decision trees implement the segment test: a pixel p is a corner at threshold b when 9 contiguous
 * pixels of the 16-pixel circle of radius 3 are all > p + b or all < p - b (svt_aom_fast9_detect, fast_9.c:2976-5913, pixels
 * 3 .. size-4 in both directions). Its score (aom_fast9_corner_score, fast_9.c:8-2936) bisects for the largest threshold in [b, 254] at which
 * the pixel still is a corner, which in closed form is (largest over the 16 arcs of the smallest difference in the arc) - 1.
 * Non-maximum suppression (nonmax.c:8-119, Compare = ">=") keeps a corner when none of its 8 neighbours is a corner with
 * a score >= its own; the output keeps raster order.
 */
static const int8_t kRing[16][2] = {/* (dx, dy), make_offsets, fast_9.c:2939-2957 */
                                    {0, 3},  {1, 3},   {2, 2},   {3, 1},   {3, 0},  {3, -1}, {2, -2}, {1, -3},
                                    {0, -3}, {-1, -3}, {-2, -2}, {-3, -1}, {-3, 0}, {-3, 1}, {-2, 2}, {-1, 3}};

/* 0: no corner at the barrier, else the corner's score (18 .. 254) */
static int fast9_score(const uint8_t *p, int stride, int barrier) {
    int d[16];
    for (int i = 0; i < 16; ++i) d[i] = (int)p[kRing[i][1] * stride + kRing[i][0]] - (int)p[0];
    int best = 0; /* largest over arcs of the smallest |difference| of one sign */
    for (int s = 0; s < 16; ++s) {
        int lo_b = 256, lo_d = 256;
        for (int k = 0; k < 9; ++k) {
            const int v = d[(s + k) & 15];
            if (v < lo_b) lo_b = v; /* brighter arc: all of ring - p */
            if (-v < lo_d) lo_d = -v; /* darker arc: all of p - ring */
        }
        if (lo_b > best) best = lo_b;
        if (lo_d > best) best = lo_d;
    }
    return best > barrier ? best - 1 : 0;
}

int orc_gm_corners(const uint8_t *buf, int width, int height, int stride, int32_t *points, int max_points) {
    if (width < 7 || height < 7) return 0;
    uint8_t *score = (uint8_t *)calloc((size_t)width * height, 1);
    if (!score) return -1;
    for (int y = 3; y < height - 3; ++y)
        for (int x = 3; x < width - 3; ++x) score[(size_t)y * width + x] = (uint8_t)fast9_score(buf + (size_t)y * stride + x, stride, 18);
    int n = 0;
    for (int y = 3; y < height - 3 && n < max_points; ++y)
        for (int x = 3; x < width - 3 && n < max_points; ++x) {
            const int s = score[(size_t)y * width + x];
            if (!s) continue;
            int keep = 1;
            for (int dy = -1; dy <= 1 && keep; ++dy)
                for (int dx = -1; dx <= 1; ++dx)
                    if ((dx || dy) && score[(size_t)(y + dy) * width + x + dx] >= s) {
                        keep = 0;
                        break;
                    }
            if (keep) {
                points[2 * n]     = x;
                points[2 * n + 1] = y;
                ++n;
            }
        }
    free(score);
    return n;
}

/* ---------------------------------------------------------------------------------------------- correspondences
 * svt_av1_determine_correspondence (corner_match.c:153-214) and improve_correspondence (:80-151): 13x13 patches,
 * the match measure cov^2 / var2 in double (svt_av1_compute_cross_correlation_c, :42-66; cov and var2 are 32-bit
 * integers), points whose patch leaves the picture are skipped, matches further away than max(w, h) / 16 are skipped,
 * "first strictly larger" wins.
 */
#define PATCH 13
#define PATCH_BY2 6
#define PATCH_SQ 169

static double cross_correlation(const uint8_t *im1, int stride1, int x1, int y1, const uint8_t *im2, int stride2, int x2, int y2) {
    int sum1 = 0, sum2 = 0, sumsq2 = 0, cross = 0;
    for (int i = -PATCH_BY2; i <= PATCH_BY2; ++i)
        for (int j = -PATCH_BY2; j <= PATCH_BY2; ++j) {
            const int v1 = im1[(y1 + i) * stride1 + x1 + j], v2 = im2[(y2 + i) * stride2 + x2 + j];
            sum1 += v1;
            sum2 += v2;
            sumsq2 += v2 * v2;
            cross += v1 * v2;
        }
    const int var2 = sumsq2 * PATCH_SQ - sum2 * sum2, cov = cross * PATCH_SQ - sum1 * sum2;
    if (cov < 0) return 0;
    return ((double)cov * cov) / ((double)var2);
}

static int patch_inside(int x, int y, int w, int h) { return x >= PATCH_BY2 && y >= PATCH_BY2 && x + PATCH_BY2 < w && y + PATCH_BY2 < h; }

static int near_enough(int x1, int y1, int x2, int y2, int thresh_sq) {
    const int dx = x1 - x2, dy = y1 - y2;
    return dx * dx + dy * dy <= thresh_sq;
}

int orc_gm_correspondences(const uint8_t *frm, int frm_stride, const int32_t *frm_corners, int num_frm_corners, const uint8_t *ref,
                           int ref_stride, const int32_t *ref_corners, int num_ref_corners, int width, int height,
                           int32_t *correspondences) {
    const int thresh = (width < height ? height : width) >> 4, thresh_sq = thresh * thresh;
    int       n = 0;
    for (int i = 0; i < num_frm_corners; ++i) {
        const int fx = frm_corners[2 * i], fy = frm_corners[2 * i + 1];
        if (!patch_inside(fx, fy, width, height)) continue;
        double best = 0.0;
        int    best_j = -1;
        for (int j = 0; j < num_ref_corners; ++j) {
            const int rx = ref_corners[2 * j], ry = ref_corners[2 * j + 1];
            if (!patch_inside(rx, ry, width, height) || !near_enough(fx, fy, rx, ry, thresh_sq)) continue;
            const double ncc = cross_correlation(frm, frm_stride, fx, fy, ref, ref_stride, rx, ry);
            if (ncc > best) {
                best   = ncc;
                best_j = j;
            }
        }
        /* the template's variance x 169 (compute_variance, :23-36); the match must reach 0.75 of it, squared */
        int sum = 0, sumsq = 0;
        for (int a = -PATCH_BY2; a <= PATCH_BY2; ++a)
            for (int b = -PATCH_BY2; b <= PATCH_BY2; ++b) {
                const int v = frm[(fy + a) * frm_stride + fx + b];
                sum += v;
                sumsq += v * v;
            }
        const int template_norm = sumsq * PATCH_SQ - sum * sum;
        if (best > template_norm * 0.75 * 0.75) {
            correspondences[4 * n]     = fx;
            correspondences[4 * n + 1] = fy;
            correspondences[4 * n + 2] = ref_corners[2 * best_j];
            correspondences[4 * n + 3] = ref_corners[2 * best_j + 1];
            ++n;
        }
    }
    /* improve_correspondence: the reference point moves to the best of 9x9 positions around it, then the frame point */
    for (int i = 0; i < n; ++i) {
        int32_t *c = correspondences + 4 * i;
        double   best = 0.0;
        int      bx = 0, by = 0;
        for (int y = -4; y <= 4; ++y)
            for (int x = -4; x <= 4; ++x) {
                if (!patch_inside(c[2] + x, c[3] + y, width, height) || !near_enough(c[0], c[1], c[2] + x, c[3] + y, thresh_sq)) continue;
                const double ncc = cross_correlation(frm, frm_stride, c[0], c[1], ref, ref_stride, c[2] + x, c[3] + y);
                if (ncc > best) {
                    best = ncc;
                    by   = y;
                    bx   = x;
                }
            }
        c[2] += bx;
        c[3] += by;
    }
    for (int i = 0; i < n; ++i) {
        int32_t *c = correspondences + 4 * i;
        double   best = 0.0;
        int      bx = 0, by = 0;
        for (int y = -4; y <= 4; ++y)
            for (int x = -4; x <= 4; ++x) {
                if (!patch_inside(c[0] + x, c[1] + y, width, height) || !near_enough(c[0] + x, c[1] + y, c[2], c[3], thresh_sq)) continue;
                const double ncc = cross_correlation(ref, ref_stride, c[2], c[3], frm, frm_stride, c[0] + x, c[1] + y);
                if (ncc > best) {
                    best = ncc;
                    by   = y;
                    bx   = x;
                }
            }
        c[0] += bx;
        c[1] += by;
    }
    return n;
}

/* ---------------------------------------------------------------------------------------------- warp / frame error
 * svt_av1_warp_error for 8 bit (EbEncWarpedMotion.c:173-213, after the shear parameters have been found valid, :233-235):
 * the picture in 32x32 blocks, raster order; a block of the reference is warped (svt_av1_warp_affine_c,
 * Source/Lib/Common/Codec/EbWarpedMotion.c:577-694, not compound, 8 bit: 3 bits dropped after the horizontal and 11 after the
 * vertical 8-tap filter) and compared with the frame through the error table; the sum stops growing at the first block
 * that takes it above best_error.
 */
static int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
static int round_shift(int v, int n) { return (v + (1 << (n - 1))) >> n; }

static void warp_block(const orc_gm_tables *t, const orc_gm_warp *wm, const uint8_t *ref, int width, int height, int stride,
                       uint8_t *pred, int p_col, int p_row, int p_width, int p_height, int p_stride) {
    const int32_t *mat = wm->wmmat;
    for (int i = p_row; i < p_row + p_height; i += 8)
        for (int j = p_col; j < p_col + p_width; j += 8) {
            const int32_t src_x = j + 4, src_y = i + 4;
            const int32_t x4 = (int32_t)((int64_t)mat[2] * src_x + (int64_t)mat[3] * src_y + mat[0]);
            const int32_t y4 = (int32_t)((int64_t)mat[4] * src_x + (int64_t)mat[5] * src_y + mat[1]);
            const int32_t ix4 = x4 >> 16, iy4 = y4 >> 16;
            int32_t       sx4 = x4 & 0xffff, sy4 = y4 & 0xffff;
            sx4 += wm->alpha * -4 + wm->beta * -4;
            sy4 += wm->gamma * -4 + wm->delta * -4;
            sx4 &= ~63;
            sy4 &= ~63;
            int32_t tmp[15 * 8];
            for (int k = -7; k < 8; ++k) {
                const uint8_t *row = ref + (size_t)clampi(iy4 + k, 0, height - 1) * stride;
                for (int l = -4; l < 4; ++l) {
                    const int      sx = sx4 + wm->beta * (k + 4) + wm->alpha * (l + 4);
                    const int16_t *f  = t->warped_filter + 8 * (round_shift(sx, 10) + 64);
                    int32_t        sum = 1 << 14;
                    for (int m = 0; m < 8; ++m) sum += row[clampi(ix4 + l - 3 + m, 0, width - 1)] * f[m];
                    tmp[(k + 7) * 8 + l + 4] = round_shift(sum, 3);
                }
            }
            const int k_end = p_row + p_height - i - 4 < 4 ? p_row + p_height - i - 4 : 4;
            const int l_end = p_col + p_width - j - 4 < 4 ? p_col + p_width - j - 4 : 4;
            for (int k = -4; k < k_end; ++k)
                for (int l = -4; l < l_end; ++l) {
                    const int      sy = sy4 + wm->delta * (k + 4) + wm->gamma * (l + 4);
                    const int16_t *f  = t->warped_filter + 8 * (round_shift(sy, 10) + 64);
                    int32_t        sum = 1 << 19;
                    for (int m = 0; m < 8; ++m) sum += tmp[(k + m + 4) * 8 + l + 4] * f[m];
                    pred[(i - p_row + k + 4) * p_stride + (j - p_col + l + 4)] = (uint8_t)clampi(round_shift(sum, 11) - 128 - 256, 0, 255);
                }
        }
}

/* svt_av1_calc_frame_error_c, EbEncWarpedMotion.c:162-171 */
int64_t orc_gm_frame_error(const orc_gm_tables *t, const uint8_t *ref, int stride, const uint8_t *dst, int p_width, int p_height,
                           int p_stride) {
    int64_t sum = 0;
    for (int i = 0; i < p_height; ++i)
        for (int j = 0; j < p_width; ++j) sum += t->error_lut[255 + dst[j + i * p_stride] - ref[j + i * stride]];
    return sum;
}

int64_t orc_gm_warp_error(const orc_gm_tables *t, const orc_gm_warp *wm, const uint8_t *ref, int width, int height, int stride,
                          const uint8_t *dst, int p_width, int p_height, int p_stride, int64_t best_error) {
    int64_t   sum = 0;
    const int bw = p_width < 32 ? p_width : 32, bh = p_height < 32 ? p_height : 32;
    uint8_t   tmp[32 * 32];
    for (int i = 0; i < p_height; i += 32)
        for (int j = 0; j < p_width; j += 32) {
            const int ww = bw < p_width - j ? bw : p_width - j, wh = bh < p_height - i ? bh : p_height - i;
            warp_block(t, wm, ref, width, height, stride, tmp, j, i, ww, wh, 32);
            sum += orc_gm_frame_error(t, tmp, 32, dst + j + (size_t)i * p_stride, ww, wh, p_stride);
            if (sum > best_error) return sum;
        }
    return sum;
}

/* ------------------------------------------------------------------------------------------- parameter refinement
 * svt_av1_refine_integerized_param (Source/Lib/Encoder/Codec/global_motion.c:135-263) for 8 bit with everything it calls:
 * svt_av1_warp_error's prologue (EbEncWarpedMotion.c:233-235: the shear parameters, result 1 when they are invalid),
 * svt_get_shear_params (Source/Lib/Common/Codec/EbWarpedMotion.c:1115-1144, with resolve_divisor_32, :343-357, whose table
 * div_lut[f], :301-322, is round(2^22 / (256 + f))), svt_warp_plane's rotation-zoom fix-up (:701-704), force_wmtype,
 * add_param_offset and get_wmtype (global_motion.c:71-133).
 */
static int64_t round_pow2_signed_64(int64_t v, int n) {
    return v < 0 ? -((-v + (((int64_t)1 << n) >> 1)) >> n) : ((v + (((int64_t)1 << n) >> 1)) >> n);
}
static int round_pow2_signed(int v, int n) { return v < 0 ? -((-v + ((1 << n) >> 1)) >> n) : ((v + ((1 << n) >> 1)) >> n); }

int orc_gm_shear(orc_gm_model *wm) {
    const int32_t *mat = wm->wmmat;
    if (!(mat[2] > 0)) return 0;
    wm->alpha = (int16_t)clampi(mat[2] - (1 << 16), INT16_MIN, INT16_MAX);
    wm->beta  = (int16_t)clampi(mat[3], INT16_MIN, INT16_MAX);
    int       shift = 31 - __builtin_clz((unsigned)mat[2]);
    const int e = mat[2] - (1 << shift);
    const int f = shift > 8 ? (e + ((1 << (shift - 8)) >> 1)) >> (shift - 8) : e << (8 - shift);
    shift += 14;
    const int16_t y = (int16_t)(((1 << 22) + (256 + f) / 2) / (256 + f));
    int64_t       v = ((int64_t)mat[4] * (1 << 16)) * y;
    wm->gamma = (int16_t)clampi((int)round_pow2_signed_64(v, shift), INT16_MIN, INT16_MAX);
    v         = ((int64_t)mat[3] * mat[4]) * y;
    wm->delta = (int16_t)clampi(mat[5] - (int)round_pow2_signed_64(v, shift) - (1 << 16), INT16_MIN, INT16_MAX);
    wm->alpha = (int16_t)(round_pow2_signed(wm->alpha, 6) * (1 << 6));
    wm->beta  = (int16_t)(round_pow2_signed(wm->beta, 6) * (1 << 6));
    wm->gamma = (int16_t)(round_pow2_signed(wm->gamma, 6) * (1 << 6));
    wm->delta = (int16_t)(round_pow2_signed(wm->delta, 6) * (1 << 6));
    if (4 * abs(wm->alpha) + 7 * abs(wm->beta) >= (1 << 16) || 4 * abs(wm->gamma) + 4 * abs(wm->delta) >= (1 << 16)) return 0;
    return 1;
}

static int32_t add_param_offset(int index, int32_t value, int32_t offset) {
    const int scale = index < 2 ? 16 - 6 : 16 - 15; /* GM_TRANS_PREC_DIFF, GM_ALPHA_PREC_DIFF */
    const int one   = (index == 2 || index == 5) ? 1 : 0;
    value = (value - (one << 16)) >> scale;
    value += offset;
    value = clampi(value, -(1 << 12), 1 << 12); /* GM_TRANS_MAX, GM_ALPHA_MAX */
    value *= 1 << scale;
    return value + (one << 16);
}

static void force_wmtype(orc_gm_model *wm, int wmtype) {
    if (wmtype == 0) wm->wmmat[0] = wm->wmmat[1] = 0;
    if (wmtype <= 1) {
        wm->wmmat[2] = 1 << 16;
        wm->wmmat[3] = 0;
    }
    if (wmtype <= 2) {
        wm->wmmat[4] = -wm->wmmat[3];
        wm->wmmat[5] = wm->wmmat[2];
    }
    wm->wmtype = wmtype;
}

typedef struct refine_ctx {
    const orc_gm_tables *t;
    const uint8_t       *ref, *dst;
    int                  width, height, r_stride, d_stride;
    int                  evals;
} refine_ctx;

/* svt_av1_warp_error as the loop calls it: it changes the model (shear, rotation-zoom fix-up) */
static int64_t refine_error(refine_ctx *c, orc_gm_model *wm, int64_t best_error) {
    c->evals++;
    if (!orc_gm_shear(wm)) return 1;
    if (wm->wmtype == 2) {
        wm->wmmat[5] = wm->wmmat[2];
        wm->wmmat[4] = -wm->wmmat[3];
    }
    orc_gm_warp w;
    memcpy(w.wmmat, wm->wmmat, sizeof(w.wmmat));
    w.alpha = wm->alpha; w.beta = wm->beta; w.gamma = wm->gamma; w.delta = wm->delta;
    return orc_gm_warp_error(c->t, &w, c->ref, c->width, c->height, c->r_stride, c->dst, c->width, c->height, c->d_stride, best_error);
}

int64_t orc_gm_refine(const orc_gm_tables *t, orc_gm_model *wm, int wmtype, const uint8_t *ref, int r_stride, const uint8_t *dst,
                      int width, int height, int d_stride, int n_refinements, int64_t best_frame_error, int *evaluations) {
    refine_ctx c = {t, ref, dst, width, height, r_stride, d_stride, 0};
    const int  n_params = 2 * wmtype; /* max_trans_model_params */
    int32_t   *mat = wm->wmmat;
    force_wmtype(wm, wmtype);
    int64_t best_error = refine_error(&c, wm, best_frame_error);
    if (best_frame_error < best_error) best_error = best_frame_error;
    int step = 1 << (n_refinements - 1);
    for (int i = 0; i < n_refinements; i++, step >>= 1)
        for (int p = 0; p < n_params; ++p) {
            int           step_dir = 0;
            const int32_t curr = mat[p];
            int32_t       best = curr;
            for (int dir = -1; dir <= 1; dir += 2) { /* left, then right */
                mat[p] = add_param_offset(p, curr, dir * step);
                const int64_t err = refine_error(&c, wm, best_error);
                if (err < best_error) {
                    best_error = err;
                    best       = mat[p];
                    step_dir   = dir;
                }
            }
            mat[p] = best;
            while (step_dir) { /* on in that direction while the error falls */
                mat[p] = add_param_offset(p, best, step * step_dir);
                const int64_t err = refine_error(&c, wm, best_error);
                if (err < best_error) {
                    best_error = err;
                    best       = mat[p];
                } else {
                    mat[p]   = best;
                    step_dir = 0;
                }
            }
        }
    force_wmtype(wm, wmtype);
    if (mat[5] == (1 << 16) && !mat[4] && mat[2] == (1 << 16) && !mat[3]) wm->wmtype = (!mat[1] && !mat[0]) ? 0 : 1;
    else wm->wmtype = (mat[2] == mat[5] && mat[3] == -mat[4]) ? 2 : 3;
    if (evaluations) *evaluations = c.evals;
    return best_error;
}
