/*
 * gm_oracle.h - TEST INFRASTRUCTURE, not part of the product: CPU restatement of the data-parallel stages of the
 * reference's global-motion estimation (compute_global_motion, Source/Lib/Encoder/Codec/EbGlobalMotionEstimation.c:155-301):
 * FAST-9 corners, corner correspondences, and the warp / frame error of a candidate model. Only tests, smoke() and the CPU
 * baseline may call it; see gm_oracle.c.
 */
#ifndef GM_ORACLE_H
#define GM_ORACLE_H
#include <stdint.h>

#define ORC_GM_MAX_CORNERS 4096 /* MAX_CORNERS, global_motion.h:35 */

/* The two tables the warp error needs. They belong to the AV1 specification and are not kept in this repository: the
 * caller hands them over (tests take them from the reference build at run time). */
typedef struct orc_gm_tables {
    const int16_t *warped_filter; /* [193][8], eb_warped_filter, EbWarpedMotion.h:44 */
    const int32_t *error_lut; /* [512], error_measure_lut, EbWarpedMotion.h:45 */
} orc_gm_tables;

typedef struct orc_gm_warp {
    int32_t wmmat[6];
    int16_t alpha, beta, gamma, delta;
} orc_gm_warp;

int orc_gm_corners(const uint8_t *buf, int width, int height, int stride, int32_t *points, int max_points);
int orc_gm_correspondences(const uint8_t *frm, int frm_stride, const int32_t *frm_corners, int num_frm_corners, const uint8_t *ref,
                           int ref_stride, const int32_t *ref_corners, int num_ref_corners, int width, int height,
                           int32_t *correspondences);
int64_t orc_gm_warp_error(const orc_gm_tables *t, const orc_gm_warp *wm, const uint8_t *ref, int width, int height, int stride,
                          const uint8_t *dst, int p_width, int p_height, int p_stride, int64_t best_error);
int64_t orc_gm_frame_error(const orc_gm_tables *t, const uint8_t *ref, int stride, const uint8_t *dst, int p_width, int p_height,
                           int p_stride);

/* the fields of EbWarpedMotionParams (EbDefinitions.h:1969-1975) the refinement loop touches */
typedef struct orc_gm_model {
    int32_t wmmat[6];
    int16_t alpha, beta, gamma, delta;
    int32_t wmtype;
} orc_gm_model;
int     orc_gm_shear(orc_gm_model *wm);
int64_t orc_gm_refine(const orc_gm_tables *t, orc_gm_model *wm, int wmtype, const uint8_t *ref, int r_stride, const uint8_t *dst,
                      int width, int height, int d_stride, int n_refinements, int64_t best_frame_error, int *evaluations);
#endif
