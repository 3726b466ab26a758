/* tpl_oracle.h - TEST INFRASTRUCTURE: the CPU restatement of the TPL dispenser's source-based mode search (tpl_oracle.c). */
#ifndef TPL_ORACLE_H
#define TPL_ORACLE_H
#include <stdint.h>
#include "../include/svtme.h"

/* cur_pix / ref_pix[list * SVTME_MAX_REFS + ref]: unpadded width x height luma pictures of one stride; the arrays `p`
 * points to are read as svtme_tpl_search reads them; out: [n_b64][SVTME_TPL_BLOCKS]. Returns an SVTME_* code. */
int orc_tpl_search(const uint8_t *cur_pix, int stride, const uint8_t *const ref_pix[SVTME_MAX_LISTS * SVTME_MAX_REFS],
                   const svtme_tpl_picture *p, svtme_tpl_block *out);
/* The forward DCT of a residual block [h][w], (w, h) one of (16, 16), (16, 8), (16, 4), (32, 8) (shape: 0 whole, 1 / 2 the
 * top-left half / quarter in each direction kept; -1 for another size), and the quantisation error of n coefficients: the
 * restatement's own pieces, exposed so that tests can pin them one by one. */
int     orc_tpl_fwd_dct(const int16_t *residual, int w, int h, int32_t *coeff, int shape);
int64_t orc_tpl_quantize_error(const int32_t *coeff, int n, const int16_t round_fp[2], const int16_t quant_fp[2], const int16_t dequant[2]);
/* rate_estimator's value for 16x16 coefficients quantised the same way (tpl_opt_flag = 0) */
int64_t orc_tpl_rate16(const int32_t *coeff, const int16_t round_fp[2], const int16_t quant_fp[2], const int16_t dequant[2]);
/* the transform-domain costs of intra modes DC_PRED .. PAETH_PRED of the 16x16 block at (x, y) (tpl_opt_flag = 0) */
void orc_tpl_intra_costs(const uint8_t *cur_pix, int stride, int w, int h, int x, int y, int64_t costs[13]);
/* The reconstruction path of the dispenser (:1004-1197) for 16x16 blocks (dispenser_search_level 0): `blocks` is what
 * orc_tpl_search found with src_dist set; recon: width x height bytes of `stride`, written block by block in the dispenser's
 * order; stats [n_b64][SVTME_TPL_BLOCKS]: TplStats as result_model_store leaves them, zero for blocks that do not exist. */
typedef struct orc_tpl_stats {
    int64_t srcrf_dist, recrf_dist, srcrf_rate, recrf_rate;
} orc_tpl_stats;
int  orc_tpl_recon(const uint8_t *cur_pix, int stride, const uint8_t *const ref_pix[SVTME_MAX_LISTS * SVTME_MAX_REFS],
                   const svtme_tpl_picture *p, const svtme_tpl_block *blocks, uint8_t *recon, orc_tpl_stats *stats);
void orc_tpl_inv_dct_add(const int32_t *dq, int w, int h, uint8_t *dst, int dst_stride);
#endif
