/*
 * tpl_oracle.c - TEST INFRASTRUCTURE, not part of the product: a plain-C restatement of the source-based mode search of
 * the reference's TPL dispenser (tpl_mc_flow_dispenser_sb_generic,
 * Source/Lib/Encoder/Codec/EbSourceBasedOperationsProcess.c:632-933): costs as sums of absolute
 * differences or as sums of absolute 16x16 forward-DCT coefficients, the quantisation error of the chosen inter
 * candidate (:935-982) and, for tpl_opt_flag = 0, the thirteen intra modes and the rate estimate. Pinned by
 * tests/golden/tpl_*.npz, which hold what the real reference (oracle/_ref/ref_me, REF_MODE_TPL) computed for the same
 * pictures and ME results. Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it.
 *
 * Pictures come unpadded: the reference reads its padded planes (generate_padding replicates the border pixels), which is
 * the same as clamping the coordinates.
 */
#include "tpl_oracle.h"

#include <limits.h>
#include <math.h>
#include <stdlib.h>
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


/* ---- the forward DCT the dispenser costs residuals with: svt_av1_wht_fwd_txfm (EbTransforms.c:3815-3830) ->
 * svt_av1_fwd_txfm2d_WxH (av1_tranform_two_d_core_c, :2301-2370): input << shift[0], an H-point transform of every column,
 * rounded >> -shift[1], a W-point transform of every row, x sqrt(2) where W : H is 2 : 1; shifts and cosine bits per size:
 * EbTransforms.h:26-57. The 1-D transforms are the butterfly networks of svt_av1_fdct4_new .. svt_av1_fdct32_new
 * (:183-677). Every rounding of a network is a two-term rotation r(a, ca, b, cb) = (a ca + b cb + 2^(bit-1)) >> bit
 * (half_btf, EbInvTransforms.h:289-300); sums and differences are exact, so a network may be written part by part - the
 * N-point transform is the N/2-point transform of the mirrored sums (even outputs) plus an "odd half" of the mirrored
 * differences - instead of stage by stage. c[k] = cos(k pi / 128) rounded to `bit` fractional bits (cospi_arr). The
 * N2 / N4 shapes (:5392, :6998) keep the top-left (W/2 x H/2) / (W/4 x H/4) coefficients of the same transform and zero
 * the rest. */
static int32_t rot(int32_t a, int32_t ca, int32_t b, int32_t cb, int bit) {
    return (int32_t)(((int64_t)a * ca + (int64_t)b * cb + ((int64_t)1 << (bit - 1))) >> bit);
}

static void cos_table(int32_t c[64], int bit) {
    for (int k = 0; k < 64; ++k) c[k] = (int32_t)lround(cos(k * 3.14159265358979323846 / 128.0) * (double)(1 << bit));
}

/* X[0], X[1], X[2], X[3] of the 4-point transform */
static void fdct4(const int32_t *x, int32_t *X, const int32_t *c, int bit) {
    const int32_t a0 = x[0] + x[3], a1 = x[1] + x[2], a2 = x[1] - x[2], a3 = x[0] - x[3];
    X[0] = rot(a0, c[32], a1, c[32], bit);
    X[2] = rot(a0, c[32], a1, -c[32], bit);
    X[1] = rot(a2, c[48], a3, c[16], bit);
    X[3] = rot(a3, c[48], a2, -c[16], bit);
}

static void fdct8(const int32_t *x, int32_t *X, const int32_t *c, int bit) {
    int32_t p[4], q[4], E[4];
    for (int i = 0; i < 4; ++i) {
        p[i] = x[i] + x[7 - i];
        q[i] = x[3 - i] - x[4 + i];
    }
    fdct4(p, E, c, bit);
    for (int i = 0; i < 4; ++i) X[2 * i] = E[i];
    const int32_t m1 = rot(q[2], c[32], q[1], -c[32], bit), m2 = rot(q[2], c[32], q[1], c[32], bit);
    const int32_t e0 = q[0] + m1, e1 = q[0] - m1, e2 = q[3] - m2, e3 = q[3] + m2;
    X[1] = rot(e0, c[56], e3, c[8], bit);
    X[5] = rot(e1, c[24], e2, c[40], bit);
    X[3] = rot(e2, c[24], e1, -c[40], bit);
    X[7] = rot(e3, c[56], e0, -c[8], bit);
}

static void fdct16(const int32_t *x, int32_t *X, const int32_t *c, int bit) {
    int32_t s[8], d[8], E[8];
    for (int i = 0; i < 8; ++i) {
        s[i] = x[i] + x[15 - i];
        d[i] = x[7 - i] - x[8 + i];
    }
    fdct8(s, E, c, bit);
    for (int i = 0; i < 8; ++i) X[2 * i] = E[i];
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

static void fdct32(const int32_t *x, int32_t *X, const int32_t *c, int bit) {
    int32_t s[16], o[16], t[16], E[16];
    for (int i = 0; i < 16; ++i) {
        s[i] = x[i] + x[31 - i];
        o[i] = x[15 - i] - x[16 + i];
    }
    fdct16(s, E, c, bit);
    for (int i = 0; i < 16; ++i) X[2 * i] = E[i];
    /* the odd half: rotations by pi/4 of the inner eight, sums, rotations by pi/8 multiples, sums, by pi/16 multiples,
     * sums, and the sixteen output rotations */
    for (int j = 4; j < 8; ++j) {
        const int32_t lo = o[j], hi = o[15 - j];
        o[j]      = rot(hi, c[32], lo, -c[32], bit);
        o[15 - j] = rot(hi, c[32], lo, c[32], bit);
    }
    for (int j = 0; j < 4; ++j) {
        t[j]      = o[j] + o[7 - j];
        t[7 - j]  = o[j] - o[7 - j];
        t[8 + j]  = o[15 - j] - o[8 + j];
        t[15 - j] = o[15 - j] + o[8 + j];
    }
    o[0] = t[0], o[1] = t[1], o[6] = t[6], o[7] = t[7], o[8] = t[8], o[9] = t[9], o[14] = t[14], o[15] = t[15];
    o[2]  = rot(t[13], c[48], t[2], -c[16], bit);
    o[3]  = rot(t[12], c[48], t[3], -c[16], bit);
    o[4]  = rot(t[4], -c[48], t[11], -c[16], bit);
    o[5]  = rot(t[5], -c[48], t[10], -c[16], bit);
    o[10] = rot(t[10], c[48], t[5], -c[16], bit);
    o[11] = rot(t[11], c[48], t[4], -c[16], bit);
    o[12] = rot(t[12], c[16], t[3], c[48], bit);
    o[13] = rot(t[13], c[16], t[2], c[48], bit);
    t[0] = o[0] + o[3], t[1] = o[1] + o[2], t[2] = o[1] - o[2], t[3] = o[0] - o[3];
    t[4] = o[7] - o[4], t[5] = o[6] - o[5], t[6] = o[6] + o[5], t[7] = o[7] + o[4];
    t[8] = o[8] + o[11], t[9] = o[9] + o[10], t[10] = o[9] - o[10], t[11] = o[8] - o[11];
    t[12] = o[15] - o[12], t[13] = o[14] - o[13], t[14] = o[14] + o[13], t[15] = o[15] + o[12];
    o[0] = t[0], o[3] = t[3], o[4] = t[4], o[7] = t[7], o[8] = t[8], o[11] = t[11], o[12] = t[12], o[15] = t[15];
    o[1]  = rot(t[14], c[56], t[1], -c[8], bit);
    o[2]  = rot(t[2], -c[56], t[13], -c[8], bit);
    o[5]  = rot(t[10], c[24], t[5], -c[40], bit);
    o[6]  = rot(t[6], -c[24], t[9], -c[40], bit);
    o[9]  = rot(t[9], c[24], t[6], -c[40], bit);
    o[10] = rot(t[10], c[40], t[5], c[24], bit);
    o[13] = rot(t[13], c[56], t[2], -c[8], bit);
    o[14] = rot(t[14], c[8], t[1], c[56], bit);
    t[0] = o[0] + o[1], t[1] = o[0] - o[1], t[2] = o[3] - o[2], t[3] = o[3] + o[2];
    t[4] = o[4] + o[5], t[5] = o[4] - o[5], t[6] = o[7] - o[6], t[7] = o[7] + o[6];
    t[8] = o[8] + o[9], t[9] = o[8] - o[9], t[10] = o[11] - o[10], t[11] = o[11] + o[10];
    t[12] = o[12] + o[13], t[13] = o[12] - o[13], t[14] = o[15] - o[14], t[15] = o[15] + o[14];
    /* output 2 m + 1 comes from the pair (t[j], t[15 - j]) with j the 4-bit reversal of m for m < 8, mirrored above */
    static const uint8_t angle[8] = {62, 30, 46, 14, 54, 22, 38, 6}; /* cosine index of t[j] in output rotation j */
    static const uint8_t where[8] = {1, 17, 9, 25, 5, 21, 13, 29};   /* output index of rotation j (bit reversal) */
    for (int j = 0; j < 8; ++j) {
        X[where[j]]      = rot(t[j], c[angle[j]], t[15 - j], c[64 - angle[j]], bit);
        X[32 - where[j]] = rot(t[15 - j], c[angle[j]], t[j], -c[64 - angle[j]], bit);
    }
}

static void fdct(int n, const int32_t *x, int32_t *X, const int32_t *c, int bit) {
    if (n == 4) fdct4(x, X, c, bit);
    else if (n == 8) fdct8(x, X, c, bit);
    else if (n == 16) fdct16(x, X, c, bit);
    else fdct32(x, X, c, bit);
}

/* residual [h][w] (row-major, pitch w) -> coefficients [h][w]; (w, h) one of (16, 16), (16, 8), (16, 4), (32, 8);
 * shape 0 / 1 / 2 = all / the top-left half / quarter in each direction. Returns 0, or -1 for another size. */
int orc_tpl_fwd_dct(const int16_t *residual, int w, int h, int32_t *coeff, int shape) {
    int shift1, bit_col = 13, bit_row; /* shift[0] = 2 and shift[2] = 0 for all four */
    if (w == 16 && h == 16) shift1 = 2, bit_row = 12;
    else if (w == 16 && h == 8) shift1 = 2, bit_row = 13;
    else if (w == 16 && h == 4) shift1 = 1, bit_row = 13;
    else if (w == 32 && h == 8) shift1 = 2, bit_row = 12;
    else return -1;
    int32_t cc[64], cr[64], buf[32 * 16], in[32], out[32];
    cos_table(cc, bit_col);
    cos_table(cr, bit_row);
    for (int col = 0; col < w; ++col) {
        for (int r = 0; r < h; ++r) in[r] = residual[r * w + col] * 4;
        fdct(h, in, out, cc, bit_col);
        for (int r = 0; r < h; ++r) buf[r * w + col] = (out[r] + (1 << (shift1 - 1))) >> shift1;
    }
    for (int r = 0; r < h; ++r) {
        fdct(w, buf + r * w, coeff + r * w, cr, bit_row);
        if (w == 2 * h) /* NewSqrt2 = 5793 with 12 fractional bits */
            for (int col = 0; col < w; ++col) coeff[r * w + col] = (int32_t)(((int64_t)coeff[r * w + col] * 5793 + 2048) >> 12);
    }
    const int kw = w >> shape, kh = h >> shape;
    for (int i = 0; i < w * h; ++i)
        if (i % w >= kw || i / w >= kh) coeff[i] = 0;
    return 0;
}

/* svt_aom_satd: common_dsp_rtcd.c:47-54 */
static int64_t satd256(const int32_t *coeff) {
    int64_t sum = 0;
    for (int i = 0; i < 256; ++i) sum += coeff[i] < 0 ? -(int64_t)coeff[i] : coeff[i];
    return sum;
}

/* get_quantize_error (:167-193) for a transform of n coefficients other than TX_32X32: svt_av1_quantize_fp_c
 * (EbFullLoop.c:318-405, no quantisation matrix, log scale 0), svt_av1_block_error_c, >> 2, at least 1 */
int64_t orc_tpl_quantize_error(const int32_t *coeff, int n, const int16_t round_fp[2], const int16_t quant_fp[2], const int16_t dequant[2]) {
    int64_t err = 0;
    for (int i = 0; i < n; ++i) {
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

/* ---- tpl_opt_flag = 0: every intra mode of a 16x16 block (:693-821).
 *
 * The neighbours, update_neighbor_samples_array_open_loop_mb(1, 1, ..) (EbEncIntraPrediction.c:1150-1235): above[-1 .. 31]
 * and left[-1 .. 31] around the block at (x, y) of the source picture. Fill values 127 / 129; the corner is the pixel at
 * (x - 1, y - 1) or 128 on the picture's first row / column; the column left of the block as far down as the picture goes
 * (at most 32 rows), its lower half then overwritten by entry 15; on the first column the pixel above the block stands in
 * for the corner and the column; the row above as far right as the picture goes, its right half overwritten by entry 15
 * unless the block lies on the first column; on the first row one entry of the left column stands in for the corner and
 * the row. */
static void tpl_neighbours(const plane *cur, int x, int y, uint8_t *above, uint8_t *left) {
    memset(above - 1, 127, 33);
    memset(left - 1, 129, 33);
    above[-1] = left[-1] = (uint8_t)((x && y) ? pel(cur, x - 1, y - 1) : 128);
    if (x) {
        if (!y) left[-1] = (uint8_t)pel(cur, x - 1, y);
        const int n = y + 32 > cur->h ? cur->h - y : 32;
        for (int i = 0; i < n; ++i) left[i] = (uint8_t)pel(cur, x - 1, y + i);
        for (int i = 16; i < 32; ++i) left[i] = left[15];
    } else if (y) {
        const int n = y + 32 > cur->h ? cur->h - y : 32;
        for (int i = -1; i < n; ++i) left[i] = (uint8_t)pel(cur, x, y - 1);
        above[-1] = (uint8_t)pel(cur, x, y - 1);
    }
    if (y) {
        const int n = x + 32 > cur->w ? cur->w - x : 32;
        for (int i = 0; i < n; ++i) above[i] = (uint8_t)pel(cur, x + i, y - 1);
        if (x)
            for (int i = 16; i < 32; ++i) above[i] = above[15];
    } else if (x) {
        const int     n = x + 32 > cur->w ? cur->w - x : 32;
        const uint8_t v = left[32 - n];
        for (int i = -1; i < n; ++i) above[i] = v;
    }
}

/* svt_av1_filter_intra_edge_c (EbIntraPrediction.c:105-128): 5-tap smoothing of p[1 .. sz - 1], ends replicated */
static void tpl_edge_filter(uint8_t *p, int sz, int strength) {
    static const int taps[3][5] = {{0, 4, 8, 4, 0}, {0, 5, 6, 5, 0}, {2, 4, 4, 4, 2}};
    if (!strength) return;
    uint8_t edge[40];
    memcpy(edge, p, (size_t)sz);
    for (int i = 1; i < sz; ++i) {
        int s = 0;
        for (int j = 0; j < 5; ++j) {
            int k = i - 2 + j;
            k = k < 0 ? 0 : (k > sz - 1 ? sz - 1 : k);
            s += edge[k] * taps[strength - 1][j];
        }
        p[i] = (uint8_t)((s + 8) >> 4);
    }
}

/* intra_edge_filter_strength(16, 16, delta, 0) (:130-165, block width + height = 32) */
static int tpl_edge_strength(int delta) {
    const int d = delta < 0 ? -delta : delta;
    return d >= 32 ? 3 : (d >= 4 ? 2 : (d >= 1 ? 1 : 0));
}

/* filter_intra_edge (:2633-2689) for a 16x16 block and a directional mode of angle p_angle: the corner when both edges are
 * read, then each edge that exists in the picture (16 + corner + the 16 extension entries zone 1 / zone 3 read). A 16x16
 * block never has its edges upsampled (use_intra_edge_upsample: width + height > 16). */
static void tpl_filter_edges(int p_angle, int x, int y, uint8_t *above, uint8_t *left) {
    if (p_angle == 90 || p_angle == 180) return;
    const int need_above = p_angle < 180, need_left = p_angle > 90;
    if (need_above && need_left) { /* filter_intra_edge_corner :2397-2404 */
        const int s = (left[0] * 5 + above[-1] * 6 + above[0] * 5 + 8) >> 4;
        above[-1] = left[-1] = (uint8_t)s;
    }
    if (need_above && y > 0) tpl_edge_filter(above - 1, 16 + 1 + (p_angle < 90 ? 16 : 0), tpl_edge_strength(p_angle - 90));
    if (need_left && x > 0) tpl_edge_filter(left - 1, 16 + 1 + (p_angle > 180 ? 16 : 0), tpl_edge_strength(p_angle - 180));
}

/* eb_dr_intra_derivative (:194-228) at the angles the nominal modes use */
static int tpl_derivative(int angle) { return angle == 23 ? 151 : (angle == 45 ? 64 : 27 /* 67 */); }

static const uint8_t k_sm_weights16[16] = {255, 225, 196, 170, 145, 123, 102, 84, 68, 54, 43, 33, 26, 20, 17, 16}; /* sm_weight_arrays + 16, :27-45 */
static const int     k_mode_angle[9]    = {0, 90, 180, 45, 135, 113, 157, 203, 67}; /* mode_to_angle_map, EbIntraPrediction.h:127-137 */

/* intra_prediction_open_loop_mb (:2691-2712) for TX_16X16: mode 0 DC_PRED (dc_pred[x > 0][y > 0]), 1 V, 2 H, 3 D45, 4 D135,
 * 5 D113, 6 D157, 7 D203, 8 D67 (dr_predictor :2373-2395 without upsampling: zones 1 / 2 / 3 :263-362), 9 SMOOTH,
 * 10 SMOOTH_V, 11 SMOOTH_H (:1056-1120), 12 PAETH (:1145-1165) */
static void tpl_intra_predict(int mode, int x, int y, const uint8_t *above, const uint8_t *left, uint8_t *dst /* [16][16] */) {
    if (mode == 0) {
        int sum = 0, n = 0;
        if (y > 0) {
            for (int i = 0; i < 16; ++i) sum += above[i];
            n += 16;
        }
        if (x > 0) {
            for (int i = 0; i < 16; ++i) sum += left[i];
            n += 16;
        }
        memset(dst, n ? (sum + (n >> 1)) / n : 128, 256);
        return;
    }
    const int angle = mode <= 8 ? k_mode_angle[mode] : 0;
    for (int r = 0; r < 16; ++r)
        for (int c = 0; c < 16; ++c) {
            int v;
            if (mode == 1) v = above[c];
            else if (mode == 2) v = left[r];
            else if (mode <= 8 && angle < 90) { /* zone 1 */
                const int dx = tpl_derivative(angle), xx = (r + 1) * dx, base = (xx >> 6) + c, sh = (xx & 0x3F) >> 1;
                v = base < 31 ? (above[base] * (32 - sh) + above[base + 1] * sh + 16) >> 5 : above[31];
            } else if (mode <= 8 && angle < 180) { /* zone 2 */
                const int dx = tpl_derivative(180 - angle), dy = tpl_derivative(angle - 90);
                const int xx = -(r + 1) * dx, base = (xx >> 6) + c;
                if (base >= -1) {
                    const int sh = (xx & 0x3F) >> 1;
                    v = (above[base] * (32 - sh) + above[base + 1] * sh + 16) >> 5;
                } else {
                    const int yy = (r << 6) - (c + 1) * dy, b2 = yy >> 6, sh = (yy & 0x3F) >> 1;
                    v = (left[b2] * (32 - sh) + left[b2 + 1] * sh + 16) >> 5;
                }
            } else if (mode <= 8) { /* zone 3 */
                const int dy = tpl_derivative(270 - angle), yy = (c + 1) * dy, base = (yy >> 6) + r, sh = (yy & 0x3F) >> 1;
                v = base < 31 ? (left[base] * (32 - sh) + left[base + 1] * sh + 16) >> 5 : left[31];
            } else if (mode == 9) {
                v = (k_sm_weights16[r] * above[c] + (256 - k_sm_weights16[r]) * left[15] + k_sm_weights16[c] * left[r] +
                     (256 - k_sm_weights16[c]) * above[15] + 256) >> 9;
            } else if (mode == 10) v = (k_sm_weights16[r] * above[c] + (256 - k_sm_weights16[r]) * left[15] + 128) >> 8;
            else if (mode == 11) v = (k_sm_weights16[c] * left[r] + (256 - k_sm_weights16[c]) * above[15] + 128) >> 8;
            else {
                const int tl = above[-1], base = above[c] + left[r] - tl;
                const int pl = abs(base - left[r]), pt = abs(base - above[c]), ptl = abs(base - tl);
                v = (pl <= pt && pl <= ptl) ? left[r] : (pt <= ptl ? above[c] : tl);
            }
            dst[r * 16 + c] = (uint8_t)v;
        }
}

/* Position of coefficient (r, c) in default_scan_16x16 (EbCoefficients.h:484-502): the anti-diagonals in turn, odd ones
 * from the top row down, even ones from the left column up */
static int tpl_scan_pos16(int r, int c) {
    const int d = r + c, before = d <= 15 ? d * (d + 1) / 2 : 256 - (31 - d) * (32 - d) / 2;
    const int r_min = d > 15 ? d - 15 : 0, r_max = d < 15 ? d : 15;
    return before + ((d & 1) ? r - r_min : r_max - r);
}

/* rate_estimator (:195-209) of the 16x16 coefficients quantised as in get_quantize_error: (eob + 1 + the sum of
 * floor(log2(|level| + 1)) over the coefficients in front of eob) << AV1_PROB_COST_SHIFT, eob = the scan position after the
 * last level that is not zero (svt_av1_quantize_fp_c, EbFullLoop.c:318-405) */
int64_t orc_tpl_rate16(const int32_t *coeff, const int16_t round_fp[2], const int16_t quant_fp[2], const int16_t dequant[2]) {
    int eob = 0, logs = 0;
    for (int i = 0; i < 256; ++i) {
        const int ac = i != 0;
        int64_t   a = coeff[i] < 0 ? -(int64_t)coeff[i] : coeff[i];
        if ((a << 1) < dequant[ac]) continue;
        a += round_fp[ac];
        a = a > INT16_MAX ? INT16_MAX : a;
        const int32_t level = (int32_t)((a * quant_fp[ac]) >> 16);
        if (!level) continue;
        const int pos = tpl_scan_pos16(i / 16, i % 16);
        if (pos + 1 > eob) eob = pos + 1;
        for (uint32_t v = (uint32_t)level + 1; v > 1; v >>= 1) ++logs; /* log2f_32, EbUtility.c:233-245 */
    }
    return (int64_t)(eob + 1 + logs) << 9;
}

/* The transform-domain cost of every intra mode 0 .. 12 of the 16x16 block at (x, y), as the loop at :735-820 computes them:
 * exposed so that tests can compare each one with the cost the reference recorded. */
void orc_tpl_intra_costs(const uint8_t *cur_pix, int stride, int w, int h, int x, int y, int64_t costs[13]) {
    const plane cur = {cur_pix, stride, w, h};
    uint8_t     above0[40], left0[40], above1[40], left1[40], prd[256];
    int16_t     res[256];
    int32_t     coeff[256];
    tpl_neighbours(&cur, x, y, above0 + 4, left0 + 4);
    for (int mode = 0; mode <= 12; ++mode) {
        const uint8_t *above = above0 + 4, *left = left0 + 4;
        if (mode >= 1 && mode <= 8) {
            memcpy(above1, above0, sizeof(above1));
            memcpy(left1, left0, sizeof(left1));
            tpl_filter_edges(k_mode_angle[mode], x, y, above1 + 4, left1 + 4);
            above = above1 + 4, left = left1 + 4;
        }
        tpl_intra_predict(mode, x, y, above, left, prd);
        for (int r = 0; r < 16; ++r)
            for (int c = 0; c < 16; ++c) res[r * 16 + c] = (int16_t)(pel(&cur, x + c, y + r) - prd[r * 16 + c]);
        orc_tpl_fwd_dct(res, 16, 16, coeff, 0);
        costs[mode] = satd256(coeff);
    }
}

int orc_tpl_search(const uint8_t *cur_pix, int stride, const uint8_t *const ref_pix[SVTME_MAX_LISTS * SVTME_MAX_REFS],
                   const svtme_tpl_picture *p, svtme_tpl_block *out) {
    if (p->pf_shape > 2) return SVTME_ERR_UNSUPPORTED;
    /* without tpl_opt_flag every cost is a transform-domain cost of the whole transform (:780-813, :878-917) */
    const int all_modes = !p->tpl_opt_flag, pf_shape = all_modes ? 0 : p->pf_shape;
    const int sad_intra = !all_modes && p->use_pred_sad_in_intra_search, sad_inter = !all_modes && p->use_pred_sad_in_inter_search;
    const int mode_end  = all_modes ? p->intra_mode_end : 0; /* :730-733 */
    if (all_modes && mode_end != 12 && mode_end != 11 && mode_end != 8) return SVTME_ERR_BAD_ARG;
    /* transform-domain costs: 16x16 blocks, whole transforms. The source distortion: the transforms the tpl levels use
     * (:543-545): 16x16, 16x8 and 16x4 (every 2nd / 4th row of a 16x16 block), 32x8 (every 4th row of a 32x32 block) */
    if ((!sad_intra || !sad_inter) && (p->dispenser_search_level != 0 || p->subsample_tx != 0)) return SVTME_ERR_UNSUPPORTED;
    if (p->src_dist && (p->subsample_tx > 2 || p->dispenser_search_level > 1 || (p->dispenser_search_level == 1 && p->subsample_tx != 2)))
        return SVTME_ERR_UNSUPPORTED;
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
            if (!p->disable_intra_pred && all_modes) { /* :693-821 */
                int64_t costs[13];
                orc_tpl_intra_costs(cur_pix, stride, w, h, x, y, costs);
                for (int mode = 0; mode <= mode_end; ++mode)
                    if (costs[mode] < o->best_intra_cost) { /* the first of the cheapest stays (:816-819) */
                        o->best_intra_cost = costs[mode];
                        o->best_intra_mode = (uint8_t)mode;
                    }
            } else if (!p->disable_intra_pred) { /* :637-692 */
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
                    orc_tpl_fwd_dct(res, 16, 16, coeff, pf_shape);
                    sad = satd256(coeff);
                }
                o->best_intra_cost = sad;
            }
            int64_t best_err = 0, best_rate = 0;
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
                int16_t     res[32 * 16];
                int32_t     coeff[32 * 16];
                const int   step = 1 << p->subsample_tx, tx_h = size >> p->subsample_tx; /* svt_aom_subtract_block, :950-957 */
                for (int r = 0; r < size; ++r)
                    for (int cc = 0; cc < size; ++cc) {
                        const int d = pel(&cur, x + cc, y + r) - pel(&rp, x + (mvx >> 3) + cc, y + (mvy >> 3) + r);
                        sad += d < 0 ? -d : d;
                        if (size <= 32 && r % step == 0 && r / step < 16) res[(r / step) * size + cc] = (int16_t)d;
                    }
                if (!sad_inter) { /* :894-917 */
                    orc_tpl_fwd_dct(res, 16, 16, coeff, pf_shape);
                    sad = satd256(coeff);
                }
                if (sad < o->best_inter_cost) {
                    if (p->src_dist) { /* what :935-982 make of the best candidate */
                        if (sad_inter && orc_tpl_fwd_dct(res, size, tx_h, coeff, pf_shape)) return SVTME_ERR_UNSUPPORTED;
                        best_err = orc_tpl_quantize_error(coeff, size * tx_h, p->y_round_fp, p->y_quant_fp, p->y_dequant);
                        if (all_modes) best_rate = orc_tpl_rate16(coeff, p->y_round_fp, p->y_quant_fp, p->y_dequant); /* :975-977 */
                    }
                    o->best_inter_cost = sad;
                    o->best_rf_idx     = (int8_t)(list == 0 ? ref : 4 + ref); /* svt_get_ref_frame_type(list, ref) - 1 */
                    o->mv_x            = mvx;
                    o->mv_y            = mvy;
                }
            }
            o->best_is_inter = o->best_inter_cost < o->best_intra_cost; /* :932-933 */
            o->src_recon_error = o->best_is_inter ? best_err : 0;
            o->src_rate        = o->best_is_inter ? best_rate : 0;
        }
    }
    return SVTME_OK;
}


/* ---- the reconstruction path (:1004-1197): prediction of the chosen mode from the reconstructed picture (intra) or the
 * reference picture (inter), residual, forward transform, quantisation, inverse transform into the reconstructed picture,
 * and the statistics result_model_store (:211-288) keeps.
 *
 * The inverse DCT: svt_av1_inv_txfm2d_add_WxH_c (inv_txfm2d_add_c, EbInvTransforms.c:2455-2533) - every row of coefficients
 * (x 1/sqrt(2) where W : H is 2 : 1), clamped to 16 bits, through a W-point transform, rounded >> -shift[0]; every column,
 * clamped to 16 bits, through an H-point transform, rounded >> 4, added to the prediction and clipped to a byte. The 1-D
 * transforms are the networks of svt_av1_idct4_new .. svt_av1_idct16_new (:75-355) at 12 cosine bits: rotations rounded as
 * in the forward transform, every sum and difference clamped to 16 bits (stage range 16 at 8 bits, :23-65). The N-point
 * network is the N/2-point network of the even inputs and an "odd half" of the odd ones, joined by N clamped sums and
 * differences; written that way here (the same values at every node). */
static int32_t clamp16(int32_t v) { return v > 32767 ? 32767 : (v < -32768 ? -32768 : v); }

static void idct4(const int32_t *x, int32_t *X, const int32_t *c) {
    const int32_t a0 = rot(x[0], c[32], x[2], c[32], 12), a1 = rot(x[0], c[32], x[2], -c[32], 12);
    const int32_t a2 = rot(x[1], c[48], x[3], -c[16], 12), a3 = rot(x[1], c[16], x[3], c[48], 12);
    X[0] = clamp16(a0 + a3), X[1] = clamp16(a1 + a2), X[2] = clamp16(a1 - a2), X[3] = clamp16(a0 - a3);
}

static void idct8(const int32_t *x, int32_t *X, const int32_t *c) {
    const int32_t ev[4] = {x[0], x[2], x[4], x[6]};
    int32_t       E[4];
    idct4(ev, E, c);
    const int32_t b4 = rot(x[1], c[56], x[7], -c[8], 12), b5 = rot(x[5], c[24], x[3], -c[40], 12);
    const int32_t b6 = rot(x[5], c[40], x[3], c[24], 12), b7 = rot(x[1], c[8], x[7], c[56], 12);
    const int32_t c4 = clamp16(b4 + b5), c5 = clamp16(b4 - b5), c6 = clamp16(-b6 + b7), c7 = clamp16(b6 + b7);
    const int32_t d[4] = {c4, rot(c5, -c[32], c6, c[32], 12), rot(c5, c[32], c6, c[32], 12), c7}; /* nodes 4 .. 7 */
    for (int i = 0; i < 4; ++i) {
        X[i]     = clamp16(E[i] + d[3 - i]);
        X[7 - i] = clamp16(E[i] - d[3 - i]);
    }
}

static void idct16(const int32_t *x, int32_t *X, const int32_t *c) {
    int32_t ev[8], E[8];
    for (int i = 0; i < 8; ++i) ev[i] = x[2 * i];
    idct8(ev, E, c);
    const int32_t b8 = rot(x[1], c[60], x[15], -c[4], 12), b9 = rot(x[9], c[28], x[7], -c[36], 12);
    const int32_t b10 = rot(x[5], c[44], x[11], -c[20], 12), b11 = rot(x[13], c[12], x[3], -c[52], 12);
    const int32_t b12 = rot(x[13], c[52], x[3], c[12], 12), b13 = rot(x[5], c[20], x[11], c[44], 12);
    const int32_t b14 = rot(x[9], c[36], x[7], c[28], 12), b15 = rot(x[1], c[4], x[15], c[60], 12);
    const int32_t c8 = clamp16(b8 + b9), c9 = clamp16(b8 - b9), c10 = clamp16(-b10 + b11), c11 = clamp16(b10 + b11);
    const int32_t c12 = clamp16(b12 + b13), c13 = clamp16(b12 - b13), c14 = clamp16(-b14 + b15), c15 = clamp16(b14 + b15);
    const int32_t d9 = rot(c9, -c[16], c14, c[48], 12), d10 = rot(c10, -c[48], c13, -c[16], 12);
    const int32_t d13 = rot(c10, -c[16], c13, c[48], 12), d14 = rot(c9, c[48], c14, c[16], 12);
    const int32_t e8 = clamp16(c8 + c11), e9 = clamp16(d9 + d10), e10 = clamp16(d9 - d10), e11 = clamp16(c8 - c11);
    const int32_t e12 = clamp16(-c12 + c15), e13 = clamp16(-d13 + d14), e14 = clamp16(d13 + d14), e15 = clamp16(c12 + c15);
    const int32_t f[8] = {e8, e9, rot(e10, -c[32], e13, c[32], 12), rot(e11, -c[32], e12, c[32], 12),
                          rot(e11, c[32], e12, c[32], 12), rot(e10, c[32], e13, c[32], 12), e14, e15}; /* nodes 8 .. 15 */
    for (int i = 0; i < 8; ++i) {
        X[i]      = clamp16(E[i] + f[7 - i]);
        X[15 - i] = clamp16(E[i] - f[7 - i]);
    }
}

/* svt_av1_idct32_new (:356-706): the odd half takes inputs 1, 17, 9, 25, 5, 21, 13, 29, 3, 19, 11, 27, 7, 23, 15, 31 as nodes
 * 16 .. 31 (arrays below are indexed node - 16) */
static void idct32(const int32_t *x, int32_t *X, const int32_t *c) {
    static const uint8_t in_of[16] = {1, 17, 9, 25, 5, 21, 13, 29, 3, 19, 11, 27, 7, 23, 15, 31};
    static const uint8_t k2[8] = {62, 30, 46, 14, 54, 22, 38, 6}; /* stage 2: node 16 + i with node 31 - i */
    int32_t ev[16], E[16], o[16], b[16], d[16], e[16], f[16], g[16], h[16];
    for (int i = 0; i < 16; ++i) ev[i] = x[2 * i], o[i] = x[in_of[i]];
    idct16(ev, E, c);
    for (int i = 0; i < 8; ++i) {
        b[i]      = rot(o[i], c[k2[i]], o[15 - i], -c[64 - k2[i]], 12);
        b[15 - i] = rot(o[i], c[64 - k2[i]], o[15 - i], c[k2[i]], 12);
    }
    for (int i = 0; i < 16; i += 4) { /* stage 3 */
        d[i] = clamp16(b[i] + b[i + 1]), d[i + 1] = clamp16(b[i] - b[i + 1]);
        d[i + 2] = clamp16(-b[i + 2] + b[i + 3]), d[i + 3] = clamp16(b[i + 2] + b[i + 3]);
    }
    /* stage 4 */
    b[0] = d[0], b[3] = d[3], b[4] = d[4], b[7] = d[7], b[8] = d[8], b[11] = d[11], b[12] = d[12], b[15] = d[15];
    b[1]  = rot(d[1], -c[8], d[14], c[56], 12), b[14] = rot(d[1], c[56], d[14], c[8], 12);
    b[2]  = rot(d[2], -c[56], d[13], -c[8], 12), b[13] = rot(d[2], -c[8], d[13], c[56], 12);
    b[5]  = rot(d[5], -c[40], d[10], c[24], 12), b[10] = rot(d[5], c[24], d[10], c[40], 12);
    b[6]  = rot(d[6], -c[24], d[9], -c[40], 12), b[9] = rot(d[6], -c[40], d[9], c[24], 12);
    /* stage 5 */
    e[0] = clamp16(b[0] + b[3]), e[1] = clamp16(b[1] + b[2]), e[2] = clamp16(b[1] - b[2]), e[3] = clamp16(b[0] - b[3]);
    e[4] = clamp16(-b[4] + b[7]), e[5] = clamp16(-b[5] + b[6]), e[6] = clamp16(b[5] + b[6]), e[7] = clamp16(b[4] + b[7]);
    e[8] = clamp16(b[8] + b[11]), e[9] = clamp16(b[9] + b[10]), e[10] = clamp16(b[9] - b[10]), e[11] = clamp16(b[8] - b[11]);
    e[12] = clamp16(-b[12] + b[15]), e[13] = clamp16(-b[13] + b[14]), e[14] = clamp16(b[13] + b[14]), e[15] = clamp16(b[12] + b[15]);
    /* stage 6 */
    f[0] = e[0], f[1] = e[1], f[6] = e[6], f[7] = e[7], f[8] = e[8], f[9] = e[9], f[14] = e[14], f[15] = e[15];
    f[2]  = rot(e[2], -c[16], e[13], c[48], 12), f[13] = rot(e[2], c[48], e[13], c[16], 12);
    f[3]  = rot(e[3], -c[16], e[12], c[48], 12), f[12] = rot(e[3], c[48], e[12], c[16], 12);
    f[4]  = rot(e[4], -c[48], e[11], -c[16], 12), f[11] = rot(e[4], -c[16], e[11], c[48], 12);
    f[5]  = rot(e[5], -c[48], e[10], -c[16], 12), f[10] = rot(e[5], -c[16], e[10], c[48], 12);
    /* stage 7 */
    for (int i = 0; i < 4; ++i) {
        g[i]      = clamp16(f[i] + f[7 - i]);
        g[7 - i]  = clamp16(f[i] - f[7 - i]);
        g[8 + i]  = clamp16(-f[8 + i] + f[15 - i]);
        g[15 - i] = clamp16(f[8 + i] + f[15 - i]);
    }
    /* stage 8 */
    for (int i = 0; i < 4; ++i) h[i] = g[i], h[12 + i] = g[12 + i];
    for (int i = 4; i < 8; ++i) {
        h[i]      = rot(g[i], -c[32], g[15 - i], c[32], 12);
        h[15 - i] = rot(g[i], c[32], g[15 - i], c[32], 12);
    }
    for (int i = 0; i < 16; ++i) { /* stage 9 */
        X[i]      = clamp16(E[i] + h[15 - i]);
        X[31 - i] = clamp16(E[i] - h[15 - i]);
    }
}

static void idct(int n, const int32_t *x, int32_t *X, const int32_t *c) {
    if (n == 4) idct4(x, X, c);
    else if (n == 8) idct8(x, X, c);
    else if (n == 16) idct16(x, X, c);
    else idct32(x, X, c);
}

/* dq [h][w] ((w, h) = (16, 16), (16, 8), (16, 4) or (32, 8)) added to the h rows at dst, dst_stride apart */
void orc_tpl_inv_dct_add(const int32_t *dq, int w, int h, uint8_t *dst, int dst_stride) {
    int32_t c[64], buf[32 * 16], in[32], out[32];
    cos_table(c, 12);
    const int shift0 = (w == 16 && h < 16) ? 1 : 2; /* inv_shift_16x16 / 32x8: 2; 16x8 / 16x4: 1, EbInvTransforms.h:53-67 */
    for (int r = 0; r < h; ++r) {
        for (int i = 0; i < w; ++i) {
            int32_t v = dq[r * w + i];
            if (w == 2 * h) v = (int32_t)(((int64_t)v * 2896 + 2048) >> 12); /* new_inv_sqrt2, 12 bits */
            in[i] = clamp16(v);
        }
        idct(w, in, out, c);
        for (int i = 0; i < w; ++i) buf[r * w + i] = (out[i] + (1 << (shift0 - 1))) >> shift0;
    }
    for (int col = 0; col < w; ++col) {
        for (int r = 0; r < h; ++r) in[r] = clamp16(buf[r * w + col]);
        idct(h, in, out, c);
        for (int r = 0; r < h; ++r) {
            const int v = dst[r * dst_stride + col] + ((out[r] + 8) >> 4);
            dst[r * dst_stride + col] = (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
        }
    }
}

/* get_quantize_error's dqcoeff and whether any level is left (eob != 0): svt_av1_quantize_fp_c as in orc_tpl_quantize_error */
static int tpl_dequantised(const int32_t *coeff, int n, const int16_t round_fp[2], const int16_t quant_fp[2], const int16_t dequant[2],
                           int32_t *dq) {
    int any = 0;
    for (int i = 0; i < n; ++i) {
        const int ac = i != 0, v = coeff[i], neg = v < 0;
        int64_t   a = neg ? -(int64_t)v : v;
        dq[i]       = 0;
        if ((a << 1) >= dequant[ac]) {
            a += round_fp[ac];
            a = a > INT16_MAX ? INT16_MAX : a;
            const int32_t level = (int32_t)((a * quant_fp[ac]) >> 16);
            if (level) {
                dq[i] = neg ? -(level * dequant[ac]) : level * dequant[ac];
                any   = 1;
            }
        }
    }
    return any;
}

int orc_tpl_recon(const uint8_t *cur_pix, int stride, const uint8_t *const ref_pix[SVTME_MAX_LISTS * SVTME_MAX_REFS],
                  const svtme_tpl_picture *p, const svtme_tpl_block *blocks, uint8_t *recon, orc_tpl_stats *stats) {
    if (p->pf_shape > 2 || p->subsample_tx > 2) return SVTME_ERR_UNSUPPORTED;
    /* 16x16 blocks with 16x16 / 16x8 / 16x4 transforms (tpl levels 1 .. 6), or 32x32 blocks with the 32x8 transform, whose
     * incomplete 64x64 blocks take 16x16 blocks with 16x4 (tpl level 7) */
    if (p->dispenser_search_level > 1 || (p->dispenser_search_level == 1 && (p->subsample_tx != 2 || !p->tpl_opt_flag))) return SVTME_ERR_UNSUPPORTED;
    const int   w = (int)p->width, h = (int)p->height, b64_w = (w + 63) / 64, b64_h = (h + 63) / 64;
    const plane cur = {cur_pix, stride, w, h}, rec = {recon, stride, w, h};
    const int   all_modes = !p->tpl_opt_flag, shape = all_modes ? 0 : p->pf_shape; /* :1131-1133 */
    const int   dc_path = p->tpl_opt_flag && p->use_pred_sad_in_intra_search; /* intra_dc_sad_path :594-595 */
    const int   sub = p->subsample_tx, step = 1 << sub;
    memset(stats, 0, sizeof(*stats) * (size_t)b64_w * b64_h * SVTME_TPL_BLOCKS);
    for (int b64 = 0; b64 < b64_w * b64_h; ++b64) {
        const int sb_x = (b64 % b64_w) * 64, sb_y = (b64 / b64_w) * 64;
        const int level = (sb_x + 64 <= w && sb_y + 64 <= h) ? p->dispenser_search_level : 0; /* :2086-2088 */
        for (int bi = level ? 1 : 5; bi <= (level ? 4 : 20); ++bi) {
            int bx, by, S;
            blk_geometry(bi, &bx, &by, &S);
            const int x = sb_x + bx, y = sb_y + by, tx_h = S >> sub;
            if (x + (S >> 1) > w || y + (S >> 1) > h) continue; /* :604-607 */
            const svtme_tpl_block *b = &blocks[(size_t)b64 * SVTME_TPL_BLOCKS + bi];
            uint8_t                prd[32 * 32];
            if (b->best_is_inter) { /* :1005-1030: the block the vector points at in the reference picture */
                const int list = b->best_rf_idx < 4 ? 0 : 1, ref = b->best_rf_idx & 3;
                if (!ref_pix[list * SVTME_MAX_REFS + ref]) return SVTME_ERR_BAD_ARG;
                const plane rp = {ref_pix[list * SVTME_MAX_REFS + ref], stride, w, h};
                for (int r = 0; r < S; ++r)
                    for (int c = 0; c < S; ++c) prd[r * S + c] = (uint8_t)pel(&rp, x + (b->mv_x >> 3) + c, y + (b->mv_y >> 3) + r);
            } else if (dc_path || S == 32) { /* :1044-1078 (a 32x32 block's mode is DC_PRED, over the same neighbours) */
                memset(prd, dc_value(&rec, x, y, S), (size_t)S * S);
            } else { /* :1080-1119: the neighbours in the reconstructed picture */
                uint8_t   above[40], left[40];
                const int mode = b->best_intra_mode;
                tpl_neighbours(&rec, x, y, above + 4, left + 4);
                if (mode >= 1 && mode <= 8) tpl_filter_edges(k_mode_angle[mode], x, y, above + 4, left + 4);
                tpl_intra_predict(mode, x, y, above + 4, left + 4, prd);
            }
            int16_t res[256];
            int32_t coeff[256], dq[256];
            for (int r = 0; r < tx_h; ++r) /* svt_aom_subtract_block over every step-th row, :1123-1130 */
                for (int c = 0; c < S; ++c) res[r * S + c] = (int16_t)(pel(&cur, x + c, y + r * step) - prd[r * step * S + c]);
            if (orc_tpl_fwd_dct(res, S, tx_h, coeff, shape)) return SVTME_ERR_UNSUPPORTED;
            const int64_t recon_error = orc_tpl_quantize_error(coeff, S * tx_h, p->y_round_fp, p->y_quant_fp, p->y_dequant);
            const int     eob = tpl_dequantised(coeff, S * tx_h, p->y_round_fp, p->y_quant_fp, p->y_dequant, dq);
            const int64_t rate_cost = p->tpl_opt_flag ? 0 : orc_tpl_rate16(coeff, p->y_round_fp, p->y_quant_fp, p->y_dequant);
            if (!p->disable_intra_pred && eob) { /* :1143-1177 (disable_intra_pred implies a picture that is not a reference) */
                orc_tpl_inv_dct_add(dq, S, tx_h, prd, S * step);
                for (int r = 0; r < S; ++r) /* the rows between take the row above them */
                    if (r % step) memcpy(prd + r * S, prd + (r - r % step) * S, (size_t)S);
            }
            for (int r = 0; r < S && y + r < h; ++r)
                for (int c = 0; c < S && x + c < w; ++c) recon[(size_t)(y + r) * stride + x + c] = prd[r * S + c];
            orc_tpl_stats *o = &stats[(size_t)b64 * SVTME_TPL_BLOCKS + bi];
            o->recrf_dist = (recon_error << 4) << sub; /* TPL_DEP_COST_SCALE_LOG2, :1179-1181 */
            o->recrf_rate = (rate_cost << 4) << sub;
            if (b->best_is_inter) { /* :980-981 */
                o->srcrf_dist = (b->src_recon_error << 4) << sub;
                o->srcrf_rate = (b->src_rate << 4) << sub;
            } else { /* :1182-1187 */
                o->srcrf_dist = o->recrf_dist;
                o->srcrf_rate = o->recrf_rate;
            }
            if (o->recrf_dist < o->srcrf_dist) o->recrf_dist = o->srcrf_dist; /* :1189-1190 */
            if (o->recrf_rate < o->srcrf_rate) o->recrf_rate = o->srcrf_rate;
            /* result_model_store :214-222 */
            o->srcrf_dist /= S, o->recrf_dist /= S, o->srcrf_rate /= S, o->recrf_rate /= S;
            if (o->srcrf_dist < 1) o->srcrf_dist = 1;
            if (o->recrf_dist < 1) o->recrf_dist = 1;
            if (o->srcrf_rate < 1) o->srcrf_rate = 1;
            if (o->recrf_rate < 1) o->recrf_rate = 1;
        }
    }
    return SVTME_OK;
}
