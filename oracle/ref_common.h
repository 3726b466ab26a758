/*
 * ref_common.h - TEST INFRASTRUCTURE: what the drivers around the real reference (ref_driver.c, gm_driver.c) share: the
 * reference's picture store of one source picture, built with the reference's own functions.
 */
#ifndef REF_COMMON_H
#define REF_COMMON_H
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "EbDefinitions.h"
#include "EbPictureBufferDesc.h"
#include "EbMcp.h"
#include "aom_dsp_rtcd.h"

#define FULL_PAD 72 /* scs->left_padding etc., Source/Lib/Encoder/Globals/EbEncHandle.c:3447-3460 */

typedef struct Planes {
    EbPictureBufferDesc full, quarter, sixteenth;
} Planes;

static void die(const char *msg) {
    fprintf(stderr, "ref driver: %s\n", msg);
    exit(2);
}

static void desc_init(EbPictureBufferDesc *d, uint32_t w, uint32_t h, uint32_t pad) {
    memset(d, 0, sizeof(*d));
    d->width     = (uint16_t)w;
    d->height    = (uint16_t)h;
    d->max_width = (uint16_t)w;
    d->max_height = (uint16_t)h;
    d->origin_x  = (uint16_t)pad;
    d->origin_y  = (uint16_t)pad;
    d->origin_bot_y = (uint16_t)pad;
    d->stride_y  = (uint16_t)(w + 2 * pad);
    d->bit_depth = EB_8BIT;
    d->luma_size = (uint32_t)d->stride_y * (h + 2 * pad);
    d->buffer_y  = (EbByte)calloc(d->luma_size + 64, 1);
    if (!d->buffer_y) die("out of memory");
}

/* The reference store of one picture, built with the reference's own functions in the order
 * picture_analysis_kernel uses them: pad_input_pictures -> generate_padding
 * (EbPictureAnalysisProcess.c:2754-2764), then downsample_filtering_input_picture (:2688-2752). */
static void build_planes(Planes *p, const uint8_t *luma, uint32_t w, uint32_t h) {
    desc_init(&p->full, w, h, FULL_PAD);
    desc_init(&p->quarter, w >> 1, h >> 1, 32);
    desc_init(&p->sixteenth, w >> 2, h >> 2, 16);
    for (uint32_t y = 0; y < h; ++y)
        memcpy(p->full.buffer_y + (size_t)(y + FULL_PAD) * p->full.stride_y + FULL_PAD, luma + (size_t)y * w, w);
    generate_padding(p->full.buffer_y, p->full.stride_y, w, h, FULL_PAD, FULL_PAD);
    downsample_2d(&p->full.buffer_y[p->full.origin_x + p->full.origin_y * p->full.stride_y],
                  p->full.stride_y, p->full.width, p->full.height,
                  &p->quarter.buffer_y[p->quarter.origin_x + p->quarter.origin_x * p->quarter.stride_y],
                  p->quarter.stride_y, 2);
    generate_padding(p->quarter.buffer_y, p->quarter.stride_y, p->quarter.width, p->quarter.height,
                     p->quarter.origin_x, p->quarter.origin_y);
    downsample_2d(&p->quarter.buffer_y[p->quarter.origin_x + p->quarter.origin_y * p->quarter.stride_y],
                  p->quarter.stride_y, p->quarter.width, p->quarter.height,
                  &p->sixteenth.buffer_y[p->sixteenth.origin_x + p->sixteenth.origin_x * p->sixteenth.stride_y],
                  p->sixteenth.stride_y, 2);
    generate_padding(p->sixteenth.buffer_y, p->sixteenth.stride_y, p->sixteenth.width, p->sixteenth.height,
                     p->sixteenth.origin_x, p->sixteenth.origin_y);
}
#endif
