/*
 * me_oracle.h - TEST INFRASTRUCTURE. CPU restatement (plain scalar C) of SVT-AV1's open-loop motion
 * estimation, used only as the checker of the HIP engine and as bench.py's "port" CPU baseline.
 * The product (svt-av1_amd/) never includes, links or calls anything declared here.
 *
 * Parity status: PINNED. The reference publishes no golden vectors for ME (its tests are C-vs-SIMD equality,
 * test/SadTest.cc), so the pins are outputs of the reference itself: oracle/_ref/ref_me runs the real
 * motion_estimation_b64() on seeded inputs; tests/golden/ holds those outputs and tests/test_oracle_golden.py
 * requires byte equality of every dumped field.
 */
#ifndef ME_ORACLE_H
#define ME_ORACLE_H
#include <stdint.h>
#include "../include/svtme.h"
#include "ref_dump.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct orc_plane {
    uint8_t *buf; /* first byte of the padded allocation */
    uint32_t stride, width, height, pad;
} orc_plane;

/* lvl[0] = sixteenth, lvl[1] = quarter, lvl[2] = full resolution; same numbering as get_me_reference's
 * `level` (Source/Lib/Encoder/Codec/EbMotionEstimation.c:1247-1273). */
typedef struct orc_planes {
    orc_plane lvl[3];
} orc_planes;

int  orc_planes_build(orc_planes *p, const uint8_t *luma, uint32_t stride, uint32_t width, uint32_t height,
                      int hme_level1_enabled);
void orc_planes_free(orc_planes *p);

int     orc_derive_controls(const svtme_preset_key *key, svtme_controls *out);
/* tf_signal_derivation_me_kernel_oq (EbMotionEstimationProcess.c:554-629) for pcs->tf_ctrls.hme_me_level 0..2 */
int     orc_derive_tf_controls(uint8_t hme_me_level, svtme_controls *out);
/* first_pass_signal_derivation_me_kernel (firstpass.c:674-716) with set_first_pass_me_hme_params_oq (:627-662) */
int     orc_derive_first_pass_controls(uint8_t input_resolution, uint32_t frame_rate, uint8_t reduce_me_search, svtme_controls *out);
uint8_t orc_input_resolution(uint32_t width, uint32_t height);

typedef struct orc_pic {
    uint32_t          width, height;
    uint64_t          picture_number;
    uint8_t           temporal_layer, is_ref, n_list, n_ref[2];
    const orc_planes *cur;
    const orc_planes *ref[2][4];
    uint64_t          ref_picture_number[2][4];
    svtme_controls    ctrl;
    /* host-tail configuration (pcs / pa_me_data fields) */
    uint8_t enable_me_8x8, enable_me_16x16, max_cand, max_refs, max_l0, input_resolution;
    uint8_t gm_enabled, gm_use_distance_based_active_th;
    /* MeContext::me_type: 0 = ME_OPEN_LOOP, 1 = ME_MCTF (temporal-filter search of one neighbour frame: one list, one
     * reference, no pruning, no host tail), 4 = ME_FIRST_PASS (no compute_distortion; check_00_center skipped when
     * bypass_zz_check), and MeContext::tf_me_exit_th */
    uint8_t  me_type;
    uint8_t  bypass_zz_check; /* scs->ipp_pass_ctrls.bypass_zz_check, ME_FIRST_PASS only */
    uint8_t  restricted_motion_vector; /* scs->static_config.restricted_motion_vector: the integer search window stays
                                          inside the block's tile (EbMotionEstimation.c:1405-1437, :1467-1498) */
    uint16_t tf_me_exit_th;
    /* scs->sb_params_array's tile fields per 64x64 block, read by ME_MCTF and ME_FIRST_PASS searches (:1408-1410); NULL, or
     * an open-loop search (pcs->sb_params_array, EbPictureControlSet.c:1684-1688): the picture is the tile */
    const svtme_tile_rect *tile_rects;
    uint32_t b64_step; /* orc_me_picture: > 1 searches only the blocks whose column and row index are multiples of it
                          (MeContext::bypass_blk_step, firstpass.c:1282-1285); the other records are left alone */
} orc_pic;

typedef struct orc_ctx orc_ctx; /* per-"thread" MeContext equivalent; state persists between b64s */
orc_ctx *orc_ctx_new(void);
void     orc_ctx_free(orc_ctx *c);

/* Runs motion_estimation_b64 (EbMotionEstimation.c:2930-3002) for one b64 and writes the resulting state.
 * run_tail = 0 stops after integer_search_b64 (the part the GPU engine replaces). */
void orc_me_b64(orc_ctx *c, const orc_pic *pic, uint32_t b64_index, ref_b64_dump *out, int run_tail);
/* Host tail only (EbMotionEstimation.c:2975-2999) on handed-over search results. */
void orc_tail_b64(orc_ctx *c, const orc_pic *pic, uint32_t b64_index, const ref_b64_dump *in, ref_b64_dump *out);

/* Algorithmic abs-diff-accumulate operation counts since the context was created:
 * [0] zz/check_00 SADs, [1] pre-HME, [2] HME-L0, [3] HME-L1, [4] HME-L2, [5] integer search. */
void orc_ctx_ops(const orc_ctx *c, uint64_t ops[6]);

/* Whole picture over `threads` host threads (b64 rows split like oracle/ref_driver.c). `out` holds n_b64 records. */
void orc_me_picture(const orc_pic *pic, ref_b64_dump *out, int run_tail, int threads, uint64_t ops[6]);

#ifdef __cplusplus
}
#endif
#endif
