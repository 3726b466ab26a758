/* tpl_oracle.h - TEST INFRASTRUCTURE: the CPU restatement of the TPL dispenser's source-based mode search (tpl_oracle.c). */
#ifndef TPL_ORACLE_H
#define TPL_ORACLE_H
#include <stdint.h>
#include "../include/svtme.h"

/* cur_pix / ref_pix[list * SVTME_MAX_REFS + ref]: unpadded width x height luma pictures of one stride; the arrays `p`
 * points to are read as svtme_tpl_search reads them; out: [n_b64][SVTME_TPL_BLOCKS]. Returns an SVTME_* code. */
int orc_tpl_search(const uint8_t *cur_pix, int stride, const uint8_t *const ref_pix[SVTME_MAX_LISTS * SVTME_MAX_REFS],
                   const svtme_tpl_picture *p, svtme_tpl_block *out);
/* The 16x16 forward DCT of a residual block [16][16] (shape: 0 whole, 1 top-left 8x8, 2 top-left 4x4 kept) and the
 * quantisation error of its coefficients: the restatement's own pieces, exposed so that tests can pin them one by one. */
void    orc_tpl_fwd_dct16x16(const int16_t *residual, int32_t *coeff, int shape);
int64_t orc_tpl_quantize_error16x16(const int32_t *coeff, const int16_t round_fp[2], const int16_t quant_fp[2], const int16_t dequant[2]);
#endif
