/* tpl_oracle.h - TEST INFRASTRUCTURE: the CPU restatement of the TPL dispenser's source-based mode search (tpl_oracle.c). */
#ifndef TPL_ORACLE_H
#define TPL_ORACLE_H
#include <stdint.h>
#include "../include/svtme.h"

/* cur_pix / ref_pix[list * SVTME_MAX_REFS + ref]: unpadded width x height luma pictures of one stride; the arrays `p`
 * points to are read as svtme_tpl_search reads them; out: [n_b64][SVTME_TPL_BLOCKS]. Returns an SVTME_* code. */
int orc_tpl_search(const uint8_t *cur_pix, int stride, const uint8_t *const ref_pix[SVTME_MAX_LISTS * SVTME_MAX_REFS],
                   const svtme_tpl_picture *p, svtme_tpl_block *out);
#endif
