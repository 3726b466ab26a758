/*
 * ref_driver.c - TEST INFRASTRUCTURE, not part of the product.
 *
 * Drives the REAL reference open-loop ME (motion_estimation_b64 and everything below it) over
 * synthetic pictures and dumps what it computed. Compiled only where /root/reference exists, against the
 * reference's own headers and objects (see oracle/Makefile, target _ref); nothing of the reference is
 * copied into this repository. The binary lands in oracle/_ref/ and is used
 *   - to generate the golden fixtures under tests/golden/ (tests/golden/make_golden.py),
 *   - to validate the oracle restatement (oracle/me_oracle.c),
 *   - as the "reference" CPU baseline of bench.py (C control code of the reference + its AVX2 SAD
 *     kernels, re-pointed below exactly as setup_rtcd_internal would on an AVX2 machine,
 *     Source/Lib/Encoder/Codec/aom_dsp_rtcd.c:406,417-423).
 *
 * It mirrors what motion_estimation_kernel() does around the call
 * (Source/Lib/Encoder/Codec/EbMotionEstimationProcess.c:698-846): signal derivation, pointing MeContext at
 * the source block and at the reference planes, then motion_estimation_b64 per 64x64 block.
 *
 * Mode REF_MODE_SHIM is the reference-side binding of the GPU engine (include/svtme.h) end to end: the engine's results
 * are unpacked into a real MeContext per 64x64 block and the reference's own host tail runs on them (run_pic_shim).
 *
 * Usage: ref_me <job file> <output file> [libsvtme.so, REF_MODE_SHIM only]
 */
#include <dlfcn.h>
#include <pthread.h>
#include <stddef.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "EbDefinitions.h"
#include "EbPictureControlSet.h"
#include "EbSequenceControlSet.h"
#include "EbMotionEstimation.h"
#include "EbMotionEstimationProcess.h"
#include "EbMotionEstimationContext.h"
#include "EbPictureBufferDesc.h"
#include "EbComputeSAD.h"
#include "aom_dsp_rtcd.h"
#include "common_dsp_rtcd.h"
#include "EbMcp.h"
#include "EbEncodeContext.h"
#include "EbModeDecisionProcess.h" /* quantizer_to_qindex */
#include "EbUtility.h" /* get_coded_blk_stats */
#include "EbCoefficients.h" /* av1_scan_orders */
#include "EbInvTransforms.h" /* txsize_to_bsize */
#include "EbTransforms.h" /* svt_av1_wht_fwd_txfm */

#include "../include/svtme.h"
#include "ref_dump.h"

/* reference functions without a prototype in an installed header */
void    setup_common_rtcd_internal(CPU_FLAGS flags);
uint8_t get_disallow_below_16x16_picture_level(EbEncMode enc_mode, EbInputResolution resolution,
                                               EB_SLICE slice_type, uint8_t sc_class1,
                                               uint8_t is_used_as_reference_flag,
                                               uint8_t temporal_layer_index);
uint8_t get_enable_me_16x16(EbEncMode enc_mode);
EbErrorType tf_signal_derivation_me_kernel_oq(PictureParentControlSet *pcs_ptr, MotionEstimationContext_t *context_ptr);
EbErrorType first_pass_signal_derivation_me_kernel(SequenceControlSet *scs_ptr, PictureParentControlSet *pcs_ptr,
                                                   MotionEstimationContext_t *context_ptr);
/* the host tail of motion_estimation_b64 (EbMotionEstimation.c:1576, :2430, :2544, :2688, :2814) */
void me_prune_ref(MeContext *context_ptr);
void construct_me_candidate_array_mrp_off(PictureParentControlSet *pcs_ptr, MeContext *context_ptr,
                                          uint32_t num_of_list_to_search, uint32_t sb_index);
void construct_me_candidate_array(PictureParentControlSet *pcs_ptr, MeContext *context_ptr,
                                  uint32_t num_of_list_to_search, uint32_t sb_index);
void perform_gm_detection(PictureParentControlSet *pcs_ptr, uint32_t sb_index, MeContext *context_ptr);
void compute_distortion(PictureParentControlSet *pcs_ptr, uint32_t sb_index, MeContext *context_ptr);
void    get_max_allocated_me_refs(uint8_t ref_count_used_list0, uint8_t ref_count_used_list1,
                                  uint8_t *max_ref_to_alloc, uint8_t *max_cand_to_alloc);

/* the TPL dispenser and what it needs set up (EbSourceBasedOperationsProcess.c:547, :375; EbResourceCoordinationProcess.c:161;
 * EbIntraPrediction.c:1867, :1879, called by svt_av1_enc_init, EbEncHandle.c:1460-1464) */
void tpl_mc_flow_dispenser_sb_generic(EncodeContext *encode_context_ptr, SequenceControlSet *scs_ptr, PictureParentControlSet *pcs_ptr,
                                      int32_t frame_idx, uint32_t sb_index, int32_t qIndex, uint8_t dispenser_search_level);
void set_tpl_extended_controls(PictureParentControlSet *pcs_ptr, uint8_t tpl_level);
void svt_av1_build_quantizer(AomBitDepth bit_depth, int32_t y_dc_delta_q, int32_t u_dc_delta_q, int32_t u_ac_delta_q,
                             int32_t v_dc_delta_q, int32_t v_ac_delta_q, Quants *const quants, Dequants *const deq);
void init_intra_dc_predictors_c_internal(void);
void init_intra_predictors_internal(void);

/* AVX2 kernels of the reference, compiled stand-alone from
 * Source/Lib/Encoder/ASM_AVX2/EbComputeSAD_Intrinsic_AVX2.c (prototypes as in aom_dsp_rtcd.h). */
#ifdef REF_HAVE_AVX2
void svt_sad_loop_kernel_avx2_intrin(uint8_t *src, uint32_t src_stride, uint8_t *ref, uint32_t ref_stride,
                                     uint32_t block_height, uint32_t block_width, uint64_t *best_sad,
                                     int16_t *x_search_center, int16_t *y_search_center,
                                     uint32_t src_stride_raw, uint8_t skip_search_line,
                                     int16_t search_area_width, int16_t search_area_height);
void svt_ext_all_sad_calculation_8x8_16x16_avx2(uint8_t *src, uint32_t src_stride, uint8_t *ref,
                                                uint32_t ref_stride, uint32_t mv, uint8_t out_8x8,
                                                uint32_t *p_best_sad_8x8, uint32_t *p_best_sad_16x16,
                                                uint32_t *p_best_mv8x8, uint32_t *p_best_mv16x16,
                                                uint32_t p_eight_sad16x16[16][8],
                                                uint32_t p_eight_sad8x8[64][8], EbBool sub_sad);
void svt_ext_eight_sad_calculation_32x32_64x64_avx2(uint32_t p_sad16x16[16][8], uint32_t *p_best_sad_32x32,
                                                    uint32_t *p_best_sad_64x64, uint32_t *p_best_mv32x32,
                                                    uint32_t *p_best_mv64x64, uint32_t mv,
                                                    uint32_t p_sad32x32[4][8]);
void svt_ext_sad_calculation_8x8_16x16_avx2_intrin(uint8_t *src, uint32_t src_stride, uint8_t *ref,
                                                   uint32_t ref_stride, uint32_t *p_best_sad_8x8,
                                                   uint32_t *p_best_sad_16x16, uint32_t *p_best_mv8x8,
                                                   uint32_t *p_best_mv16x16, uint32_t mv,
                                                   uint32_t *p_sad16x16, uint32_t *p_sad8x8, EbBool sub_sad);
uint32_t svt_nxm_sad_kernel_helper_avx2(const uint8_t *src, uint32_t src_stride, const uint8_t *ref,
                                        uint32_t ref_stride, uint32_t height, uint32_t width);
#endif

#include "ref_common.h"

typedef struct Job {
    ref_job_header hdr;
    uint8_t       *frames;
    ref_job_pic   *pics;
    Planes        *planes;
} Job;

typedef struct PicState {
    SequenceControlSet      *scs;
    PictureParentControlSet *pcs;
    MotionEstimationData    *me_data;
    uint32_t                 n_b64, b64_w, b64_h;
    uint8_t                 *out; /* ref_pic_dump followed by n_b64 ref_b64_dump */
} PicState;

static void fill_controls(svtme_controls *c, const MeContext *m) {
    memset(c, 0, sizeof(*c));
    c->me_sa.sa_min.width  = m->me_sa.sa_min.width;  c->me_sa.sa_min.height = m->me_sa.sa_min.height;
    c->me_sa.sa_max.width  = m->me_sa.sa_max.width;  c->me_sa.sa_max.height = m->me_sa.sa_max.height;
    c->hme_l0_sa.sa_min.width = m->hme_l0_sa.sa_min.width; c->hme_l0_sa.sa_min.height = m->hme_l0_sa.sa_min.height;
    c->hme_l0_sa.sa_max.width = m->hme_l0_sa.sa_max.width; c->hme_l0_sa.sa_max.height = m->hme_l0_sa.sa_max.height;
    c->hme_l1_sa.width = m->hme_l1_sa.width; c->hme_l1_sa.height = m->hme_l1_sa.height;
    c->hme_l2_sa.width = m->hme_l2_sa.width; c->hme_l2_sa.height = m->hme_l2_sa.height;
    c->num_hme_sa_w = m->num_hme_sa_w; c->num_hme_sa_h = m->num_hme_sa_h;
    c->enable_hme_flag = m->enable_hme_flag;
    c->enable_hme_level0_flag = m->enable_hme_level0_flag;
    c->enable_hme_level1_flag = m->enable_hme_level1_flag;
    c->enable_hme_level2_flag = m->enable_hme_level2_flag;
    c->hme_search_method = m->hme_search_method;
    c->me_search_method  = m->me_search_method;
    c->prehme_enable = m->prehme_ctrl.enable;
    if (m->prehme_ctrl.enable) {
        c->prehme_skip_search_line = m->prehme_ctrl.skip_search_line;
        c->prehme_l1_early_exit = m->prehme_ctrl.l1_early_exit;
        for (int i = 0; i < 2; ++i) {
            c->prehme_sa_cfg[i].sa_min.width  = m->prehme_ctrl.prehme_sa_cfg[i].sa_min.width;
            c->prehme_sa_cfg[i].sa_min.height = m->prehme_ctrl.prehme_sa_cfg[i].sa_min.height;
            c->prehme_sa_cfg[i].sa_max.width  = m->prehme_ctrl.prehme_sa_cfg[i].sa_max.width;
            c->prehme_sa_cfg[i].sa_max.height = m->prehme_ctrl.prehme_sa_cfg[i].sa_max.height;
        }
    }
    c->enable_me_hme_ref_pruning = m->me_hme_prune_ctrls.enable_me_hme_ref_pruning;
    c->prune_ref_if_hme_sad_dev_bigger_than_th = m->me_hme_prune_ctrls.prune_ref_if_hme_sad_dev_bigger_than_th;
    c->prune_ref_if_me_sad_dev_bigger_than_th  = m->me_hme_prune_ctrls.prune_ref_if_me_sad_dev_bigger_than_th;
    if (c->enable_me_hme_ref_pruning) c->protect_closest_refs = m->me_hme_prune_ctrls.protect_closest_refs;
    c->enable_me_sr_adjustment = m->me_sr_adjustment_ctrls.enable_me_sr_adjustment;
    if (c->enable_me_sr_adjustment) {
        c->distance_based_hme_resizing = m->me_sr_adjustment_ctrls.distance_based_hme_resizing;
        c->reduce_me_sr_based_on_mv_length_th = m->me_sr_adjustment_ctrls.reduce_me_sr_based_on_mv_length_th;
        c->stationary_hme_sad_abs_th = m->me_sr_adjustment_ctrls.stationary_hme_sad_abs_th;
        c->stationary_me_sr_divisor = m->me_sr_adjustment_ctrls.stationary_me_sr_divisor;
        c->reduce_me_sr_based_on_hme_sad_abs_th = m->me_sr_adjustment_ctrls.reduce_me_sr_based_on_hme_sad_abs_th;
        c->me_sr_divisor_for_low_hme_sad = m->me_sr_adjustment_ctrls.me_sr_divisor_for_low_hme_sad;
    }
    c->reduce_hme_l0_sr_th_min = m->reduce_hme_l0_sr_th_min;
    c->reduce_hme_l0_sr_th_max = m->reduce_hme_l0_sr_th_max;
    c->stat_factor = m->stat_factor;
    c->use_best_unipred_cand_only = m->use_best_unipred_cand_only;
    c->prune_me_candidates_th = m->prune_me_candidates_th;
    c->me_early_exit_th = m->me_early_exit_th;
}

static void setup_pic(const Job *job, uint32_t pi, PicState *st) {
    const ref_job_pic *jp = &job->pics[pi];
    const uint32_t     w = job->hdr.width, h = job->hdr.height;
    SequenceControlSet      *scs = (SequenceControlSet *)calloc(1, sizeof(*scs));
    PictureParentControlSet *pcs = (PictureParentControlSet *)calloc(1, sizeof(*pcs));
    if (!scs || !pcs) die("out of memory");
    st->scs = scs;
    st->pcs = pcs;
    st->b64_w = (w + 63) / 64;
    st->b64_h = (h + 63) / 64;
    st->n_b64 = st->b64_w * st->b64_h;

    derive_input_resolution(&scs->input_resolution, w * h);
    scs->static_config.frame_rate               = jp->frame_rate;
    scs->static_config.restricted_motion_vector = jp->restricted_motion_vector;
    scs->static_config.enc_mode                 = (int8_t)jp->enc_mode;
    scs->sb_sz                                  = 64;
    scs->ipp_pass_ctrls.bypass_zz_check         = 0;
    if (jp->me_type == ME_FIRST_PASS) { /* what EbEncHandle.c:3040-3070 sets per first-pass level */
        scs->ipp_pass_ctrls.bypass_zz_check  = jp->tf_hme_me_level & 1;
        scs->ipp_pass_ctrls.reduce_me_search = (jp->tf_hme_me_level >> 1) & 1;
    }

    pcs->scs_ptr        = scs;
    pcs->picture_number = jp->picture_number;
    pcs->aligned_width  = (uint16_t)w;
    pcs->aligned_height = (uint16_t)h;
    pcs->enc_mode       = (EbEncMode)jp->enc_mode;
    pcs->sc_class1      = jp->sc_class1;
    pcs->temporal_layer_index      = jp->temporal_layer;
    pcs->is_used_as_reference_flag = jp->is_ref;
    pcs->slice_type     = jp->n_list == 2 ? B_SLICE : P_SLICE;
    pcs->ref_list0_count_try = jp->n_ref[0];
    pcs->ref_list1_count_try = jp->n_ref[1];
    pcs->sb_total_count = (uint16_t)st->n_b64;
    pcs->max_number_of_pus_per_sb = SQUARE_PU_COUNT; /* EbResourceCoordinationProcess.c:720 */
    /* HME level flags: signal_derivation_multi_processes_oq, EbPictureDecisionProcess.c:1797-1814 */
    pcs->enable_hme_flag        = 1;
    pcs->enable_hme_level0_flag = 1;
    if (pcs->sc_class1 || pcs->enc_mode <= ENC_M6) {
        pcs->enable_hme_level1_flag = 1;
        pcs->enable_hme_level2_flag = 1;
    } else if (pcs->enc_mode <= ENC_M13) {
        pcs->enable_hme_level1_flag = 1;
        pcs->enable_hme_level2_flag = 0;
    } else {
        pcs->enable_hme_level1_flag = 0;
        pcs->enable_hme_level2_flag = 0;
    }
    if (jp->me_type == ME_FIRST_PASS) /* first_pass_signal_derivation_multi_processes, firstpass.c:488-491 */
        pcs->enable_hme_level1_flag = pcs->enable_hme_level2_flag = 1;
    pcs->use_best_me_unipred_cand_only = pcs->enc_mode <= ENC_M9 ? 0 : 1; /* :2058-2061 */
    pcs->enable_me_8x8 = !get_disallow_below_16x16_picture_level(
        pcs->enc_mode, scs->input_resolution, B_SLICE, 0, 1, 0); /* EbPictureControlSet.c:1533 */
    pcs->enable_me_16x16 = get_enable_me_16x16(pcs->enc_mode);
    /* temporal-filter HME level flags, EbPictureDecisionProcess.c:1816-1841 */
    pcs->tf_ctrls.hme_me_level     = jp->tf_hme_me_level;
    pcs->tf_enable_hme_flag        = 1;
    pcs->tf_enable_hme_level0_flag = 1;
    pcs->tf_enable_hme_level1_flag = jp->tf_hme_me_level <= 1;
    pcs->tf_enable_hme_level2_flag = jp->tf_hme_me_level == 0;
    pcs->gm_ctrls.enabled = jp->gm_enabled;
    pcs->gm_ctrls.use_distance_based_active_th = jp->gm_dist_th;

    pcs->sb_params_array = (SbParams *)calloc(st->n_b64, sizeof(SbParams));
    for (uint32_t i = 0; i < st->n_b64; ++i) {
        SbParams *sp = &pcs->sb_params_array[i];
        uint32_t  bx = i % st->b64_w, by = i / st->b64_w;
        sp->horizontal_index = (uint8_t)bx;
        sp->vertical_index   = (uint8_t)by;
        sp->origin_x = (uint16_t)(bx * 64);
        sp->origin_y = (uint16_t)(by * 64);
        sp->width  = (uint8_t)((w - bx * 64) < 64 ? (w - bx * 64) : 64);
        sp->height = (uint8_t)((h - by * 64) < 64 ? (h - by * 64) : 64);
        sp->tile_start_x = 0; sp->tile_start_y = 0; sp->tile_end_x = w; sp->tile_end_y = h;
    }
    /* The sequence's own array: ME_MCTF and ME_FIRST_PASS read their tiles from it (EbMotionEstimation.c:1408-1410), the
     * open-loop search reads the picture's array above, whose one tile is the picture (EbPictureControlSet.c:1684-1688).
     * Tile fields as EbResourceCoordinationProcess.c:1304-1349 leaves them for uniform tiles of tw x th 64x64 blocks. */
    scs->sb_params_array = (SbParams *)calloc(st->n_b64, sizeof(SbParams));
    memcpy(scs->sb_params_array, pcs->sb_params_array, st->n_b64 * sizeof(SbParams));
    const uint32_t tw = jp->tile_b64 & 15, th = jp->tile_b64 >> 4;
    for (uint32_t i = 0; i < st->n_b64; ++i) {
        SbParams *sp = &scs->sb_params_array[i];
        uint32_t  bx = i % st->b64_w, by = i / st->b64_w;
        if (tw) {
            sp->tile_start_x = bx / tw * tw * 64;
            sp->tile_end_x   = (sp->tile_start_x + tw * 64) < w ? (sp->tile_start_x + tw * 64) : w;
        }
        if (th) {
            sp->tile_start_y = by / th * th * 64;
            sp->tile_end_y   = (sp->tile_start_y + th * 64) < h ? (sp->tile_start_y + th * 64) : h;
        }
    }

    pcs->rc_me_distortion            = (uint32_t *)calloc(st->n_b64, 4);
    pcs->me_64x64_distortion         = (uint32_t *)calloc(st->n_b64, 4);
    pcs->me_32x32_distortion         = (uint32_t *)calloc(st->n_b64, 4);
    pcs->me_16x16_distortion         = (uint32_t *)calloc(st->n_b64, 4);
    pcs->me_8x8_distortion           = (uint32_t *)calloc(st->n_b64, 4);
    pcs->me_8x8_cost_variance        = (uint32_t *)calloc(st->n_b64, 4);
    pcs->stationary_block_present_sb = (uint8_t *)calloc(st->n_b64, 1);
    pcs->rc_me_allow_gm              = (uint8_t *)calloc(st->n_b64, 1);

    MotionEstimationData *md = (MotionEstimationData *)calloc(1, sizeof(*md));
    uint8_t max_refs, max_cand;
    get_max_allocated_me_refs(job->hdr.ref_count_used_list0, job->hdr.ref_count_used_list1, &max_refs, &max_cand);
    md->max_cand = max_cand;
    md->max_refs = max_refs;
    md->max_l0   = (uint8_t)job->hdr.ref_count_used_list0;
    if (jp->me_type == ME_FIRST_PASS) { /* "assume max 2 references for first pass", firstpass.c:1273-1276 */
        md->max_cand = 3;
        md->max_refs = 2;
        md->max_l0   = 1;
    }
    md->me_results = (MeSbResults **)calloc(st->n_b64, sizeof(MeSbResults *));
    for (uint32_t i = 0; i < st->n_b64; ++i) {
        MeSbResults *r = (MeSbResults *)calloc(1, sizeof(*r));
        r->me_mv_array = (MvCandidate *)calloc((size_t)SQUARE_PU_COUNT * REF_DUMP_MAX_REFS, sizeof(MvCandidate));
        r->me_candidate_array = (MeCandidate *)calloc((size_t)SQUARE_PU_COUNT * REF_DUMP_MAX_CAND, sizeof(MeCandidate));
        r->total_me_candidate_index = (uint8_t *)calloc(SQUARE_PU_COUNT, 1);
        md->me_results[i] = r;
    }
    pcs->pa_me_data = md;
    st->me_data     = md;
    st->out = (uint8_t *)calloc(1, sizeof(ref_pic_dump) + (size_t)st->n_b64 * sizeof(ref_b64_dump));
    if (!st->out) die("out of memory");
}

static void free_pic(PicState *st) {
    PictureParentControlSet *pcs = st->pcs;
    for (uint32_t i = 0; i < st->n_b64; ++i) {
        free(st->me_data->me_results[i]->me_mv_array);
        free(st->me_data->me_results[i]->me_candidate_array);
        free(st->me_data->me_results[i]->total_me_candidate_index);
        free(st->me_data->me_results[i]);
    }
    free(st->me_data->me_results);
    free(st->me_data);
    free(pcs->sb_params_array);
    free(st->scs->sb_params_array);
    free(pcs->rc_me_distortion); free(pcs->me_64x64_distortion); free(pcs->me_32x32_distortion);
    free(pcs->me_16x16_distortion); free(pcs->me_8x8_distortion); free(pcs->me_8x8_cost_variance);
    free(pcs->stationary_block_present_sb); free(pcs->rc_me_allow_gm);
    free(pcs); free(st->scs); free(st->out);
    memset(st, 0, sizeof(*st));
}

static void dump_b64(ref_b64_dump *d, const MeContext *m, const PicState *st, uint32_t b64) {
    const PictureParentControlSet *pcs = st->pcs;
    memcpy(d->best_sad, m->p_sb_best_sad, sizeof(d->best_sad));
    memcpy(d->best_mv, m->p_sb_best_mv, sizeof(d->best_mv));
    for (int l = 0; l < 2; ++l)
        for (int r = 0; r < 4; ++r) {
            d->hme_sc_x[l][r] = m->search_results[l][r].hme_sc_x;
            d->hme_sc_y[l][r] = m->search_results[l][r].hme_sc_y;
            d->hme_sad[l][r]  = m->search_results[l][r].hme_sad;
            d->do_ref[l][r]   = m->search_results[l][r].do_ref;
            d->zz_sad[l][r]   = m->zz_sad[l][r];
            d->sr_divisor[l][r] = m->reduce_me_sr_divisor[l][r];
            for (int a = 0; a < 2; ++a) {
                for (int b = 0; b < 2; ++b) {
                    d->l0_x[l][r][a][b] = m->x_hme_level0_search_center[l][r][a][b];
                    d->l0_y[l][r][a][b] = m->y_hme_level0_search_center[l][r][a][b];
                    d->l0_sad[l][r][a][b] = m->hme_level0_sad[l][r][a][b];
                    d->l1_x[l][r][a][b] = m->x_hme_level1_search_center[l][r][a][b];
                    d->l1_y[l][r][a][b] = m->y_hme_level1_search_center[l][r][a][b];
                    d->l1_sad[l][r][a][b] = m->hme_level1_sad[l][r][a][b];
                    d->l2_x[l][r][a][b] = m->x_hme_level2_search_center[l][r][a][b];
                    d->l2_y[l][r][a][b] = m->y_hme_level2_search_center[l][r][a][b];
                    d->l2_sad[l][r][a][b] = m->hme_level2_sad[l][r][a][b];
                }
                d->pre_x[l][r][a]   = m->prehme_data[l][r][a].best_mv.as_mv.col;
                d->pre_y[l][r][a]   = m->prehme_data[l][r][a].best_mv.as_mv.row;
                d->pre_sad[l][r][a] = m->prehme_data[l][r][a].sad;
            }
        }
    memcpy(d->me_distortion, m->me_distortion, sizeof(d->me_distortion));
    const MeSbResults *r = pcs->pa_me_data->me_results[b64];
    memcpy(d->total_me_candidate_index, r->total_me_candidate_index, SQUARE_PU_COUNT);
    memset(d->me_mv_array, 0, sizeof(d->me_mv_array));
    memset(d->me_candidate_array, 0, sizeof(d->me_candidate_array));
    memcpy(d->me_mv_array, r->me_mv_array, (size_t)SQUARE_PU_COUNT * pcs->pa_me_data->max_refs * 4);
    memcpy(d->me_candidate_array, r->me_candidate_array, (size_t)SQUARE_PU_COUNT * pcs->pa_me_data->max_cand);
    d->rc_me_distortion = pcs->rc_me_distortion[b64];
    d->me_64x64_distortion = pcs->me_64x64_distortion[b64];
    d->me_32x32_distortion = pcs->me_32x32_distortion[b64];
    d->me_16x16_distortion = pcs->me_16x16_distortion[b64];
    d->me_8x8_distortion   = pcs->me_8x8_distortion[b64];
    d->me_8x8_cost_variance = pcs->me_8x8_cost_variance[b64];
    d->stationary_block_present_sb = pcs->stationary_block_present_sb[b64];
    d->rc_me_allow_gm = pcs->rc_me_allow_gm[b64];
    d->tf_me_exit = m->me_type == ME_MCTF && m->tf_use_pred_64x64_only_th == (uint8_t)~0;
    if (d->tf_me_exit) memset(d->best_sad, 0, sizeof(d->best_sad)); /* left over from the previous block */
}

typedef struct Worker {
    const Job *job;
    PicState  *st;
    uint32_t   pi;
    uint32_t   row_begin, row_end; /* b64 rows */
    int        dump;
} Worker;

/* One ME thread: own MeContext (motion_estimation_context_ctor, EbMotionEstimationProcess.c:642-655), then the
 * b64 loop of motion_estimation_kernel (:750-846). */
static void *worker_main(void *arg) {
    Worker                   *wk  = (Worker *)arg;
    const Job                *job = wk->job;
    PicState                 *st  = wk->st;
    const ref_job_pic        *jp  = &job->pics[wk->pi];
    PictureParentControlSet  *pcs = st->pcs;
    SequenceControlSet       *scs = st->scs;
    MotionEstimationContext_t mectx;
    memset(&mectx, 0, sizeof(mectx));
    MeContext *m = (MeContext *)calloc(1, sizeof(MeContext));
    if (!m) die("out of memory");
    me_context_ctor(m);
    mectx.me_context_ptr = m;
    if (jp->me_type == ME_MCTF) {
        /* the temporal filter's ME thread: tf_signal_derivation_me_kernel_oq (EbMotionEstimationProcess.c:693-694), then per
         * block create_me_context_and_picture_control (:302-331) and the MeContext set-up of :4844-4865 */
        m->me_type = ME_MCTF;
        tf_signal_derivation_me_kernel_oq(pcs, &mectx);
        m->hme_search_method = FULL_SAD_SEARCH;
    } else if (jp->me_type == ME_FIRST_PASS) {
        /* first_pass_me's thread: EbMotionEstimationProcess.c:687-689 / :696, then first_pass_setup_me_context per block
         * (firstpass.c:1166-1188) */
        m->me_type = ME_FIRST_PASS;
        first_pass_signal_derivation_me_kernel(scs, pcs, &mectx);
    } else {
        m->me_type = ME_OPEN_LOOP;
        signal_derivation_me_kernel_oq(scs, pcs, &mectx);
    }

    Planes *cur = &job->planes[jp->cur_frame];
    if (wk->dump && wk->row_begin == 0) {
        ref_pic_dump *pd = (ref_pic_dump *)st->out;
        fill_controls(&pd->ctrl, m);
        pd->enable_me_8x8   = pcs->enable_me_8x8;
        pd->enable_me_16x16 = pcs->enable_me_16x16;
        pd->max_cand = pcs->pa_me_data->max_cand;
        pd->max_refs = pcs->pa_me_data->max_refs;
        pd->max_l0   = pcs->pa_me_data->max_l0;
        pd->input_resolution = (uint8_t)scs->input_resolution;
        pd->n_b64 = st->n_b64;
    }
    for (uint32_t by = wk->row_begin; by < wk->row_end; ++by) {
        for (uint32_t bx = 0; bx < st->b64_w; ++bx) {
            uint32_t b64 = bx + by * st->b64_w;
            uint32_t ox = bx * 64, oy = by * 64;
            /* MeContext::bypass_blk_step, firstpass.c:1282-1285 */
            if (job->hdr.b64_step > 1 && (bx % job->hdr.b64_step != 0 || by % job->hdr.b64_step != 0)) continue;
            m->b64_src_ptr    = &cur->full.buffer_y[(cur->full.origin_y + oy) * cur->full.stride_y + cur->full.origin_x + ox];
            m->b64_src_stride = cur->full.stride_y;
            if (m->enable_hme_level1_flag) {
                m->quarter_b64_buffer = &cur->quarter.buffer_y[(cur->quarter.origin_y + (oy >> 1)) * cur->quarter.stride_y +
                                                               cur->quarter.origin_x + (ox >> 1)];
                m->quarter_b64_buffer_stride = cur->quarter.stride_y;
            }
            if (m->enable_hme_level0_flag) {
                m->sixteenth_b64_buffer = &cur->sixteenth.buffer_y[(cur->sixteenth.origin_y + (oy >> 2)) * cur->sixteenth.stride_y +
                                                                   cur->sixteenth.origin_x + (ox >> 2)];
                m->sixteenth_b64_buffer_stride = cur->sixteenth.stride_y;
            }
            m->me_type = jp->me_type == ME_MCTF ? ME_MCTF : (jp->me_type == ME_FIRST_PASS ? ME_FIRST_PASS : ME_OPEN_LOOP);
            m->tf_me_exit_th             = jp->tf_me_exit_th;
            m->tf_use_pred_64x64_only_th = 0;
            m->num_of_list_to_search       = jp->n_list;
            m->num_of_ref_pic_to_search[0] = jp->n_ref[0];
            if (jp->n_list == 2) m->num_of_ref_pic_to_search[1] = jp->n_ref[1];
            m->temporal_layer_index      = jp->temporal_layer;
            m->is_used_as_reference_flag = jp->is_ref;
            for (int l = 0; l < jp->n_list; ++l)
                for (int r = 0; r < jp->n_ref[l]; ++r) {
                    Planes *rp = &job->planes[jp->ref_frame[l][r]];
                    m->me_ds_ref_array[l][r].picture_ptr           = &rp->full;
                    m->me_ds_ref_array[l][r].quarter_picture_ptr   = &rp->quarter;
                    m->me_ds_ref_array[l][r].sixteenth_picture_ptr = &rp->sixteenth;
                    m->me_ds_ref_array[l][r].picture_number        = jp->ref_poc[l][r];
                }
            motion_estimation_b64(pcs, b64, ox, oy, m, &cur->full);
            if (wk->dump)
                dump_b64((ref_b64_dump *)(st->out + sizeof(ref_pic_dump)) + b64, m, st, b64);
        }
    }
    free(m->p_eight_pos_sad16x16);
    free(m);
    return NULL;
}

/* ---------------------------------------------------------------------------------------------------------------
 * REF_MODE_SHIM: what the patch to motion_estimation_kernel() does (EbMotionEstimationProcess.c:750-846 with the call at
 * :841 replaced): per picture store_put / submit / wait through the engine's C ABI, per 64x64 block svtme_unpack_b64
 * straight into the thread's MeContext, then the rest of motion_estimation_b64 (EbMotionEstimation.c:2972-3000) with the
 * reference's own functions. The library is loaded at run time so that the other modes need no GPU runtime. */

/* svtme_search_result is copied byte for byte into MeContext::search_results */
_Static_assert(sizeof(svtme_search_result) == sizeof(SearchResults), "svtme_search_result / SearchResults size");
_Static_assert(offsetof(svtme_search_result, list_i) == offsetof(SearchResults, list_i), "list_i");
_Static_assert(offsetof(svtme_search_result, ref_i) == offsetof(SearchResults, ref_i), "ref_i");
_Static_assert(offsetof(svtme_search_result, hme_sc_x) == offsetof(SearchResults, hme_sc_x), "hme_sc_x");
_Static_assert(offsetof(svtme_search_result, hme_sc_y) == offsetof(SearchResults, hme_sc_y), "hme_sc_y");
_Static_assert(offsetof(svtme_search_result, hme_sad) == offsetof(SearchResults, hme_sad), "hme_sad");
_Static_assert(offsetof(svtme_search_result, do_ref) == offsetof(SearchResults, do_ref), "do_ref");
_Static_assert(sizeof(((MeContext *)0)->search_results) == SVTME_MAX_LISTS * SVTME_MAX_REFS * sizeof(svtme_search_result),
               "MeContext::search_results shape");
_Static_assert(sizeof(((MeContext *)0)->p_sb_best_sad) == SVTME_MAX_LISTS * SVTME_MAX_REFS * SVTME_PU_COUNT * 4,
               "MeContext::p_sb_best_sad shape");
_Static_assert(sizeof(((MeContext *)0)->p_sb_best_mv) == SVTME_MAX_LISTS * SVTME_MAX_REFS * SVTME_PU_COUNT * 4,
               "MeContext::p_sb_best_mv shape");
_Static_assert(sizeof(((MeContext *)0)->zz_sad) == SVTME_MAX_LISTS * SVTME_MAX_REFS * 4, "MeContext::zz_sad shape");
_Static_assert(sizeof(((MeContext *)0)->reduce_me_sr_divisor) == SVTME_MAX_LISTS * SVTME_MAX_REFS * 4,
               "MeContext::reduce_me_sr_divisor shape");

typedef struct Shim {
    void         *dl;
    svtme_engine *engine;
    int (*create)(const svtme_config *, svtme_engine **);
    void (*destroy)(svtme_engine *);
    int (*store_put)(svtme_engine *, uint64_t, uint64_t, const uint8_t *, uint32_t, uint32_t, uint32_t);
    int (*submit)(svtme_engine *, const svtme_picture *, svtme_ticket *);
    int (*wait)(svtme_engine *, svtme_ticket, svtme_results *);
    int (*done)(svtme_engine *, svtme_ticket);
    int (*unpack)(const svtme_results *, uint32_t, uint32_t *, uint32_t *, svtme_search_result *, uint32_t *, uint32_t *);
    int (*tpl_search)(svtme_engine *, const svtme_tpl_picture *, svtme_tpl_block *);
    int (*set_tile_rects)(svtme_engine *, uint32_t, uint32_t, const svtme_tile_rect *, uint32_t);
    int tiles_set;
} Shim;

static void *shim_sym(Shim *s, const char *name) {
    void *f = dlsym(s->dl, name);
    if (!f) {
        fprintf(stderr, "ref_me: %s: %s\n", name, dlerror());
        exit(2);
    }
    return f;
}

static void shim_open(Shim *s, const char *lib, const Job *job) {
    memset(s, 0, sizeof(*s));
    s->dl = dlopen(lib, RTLD_NOW | RTLD_LOCAL);
    if (!s->dl) {
        fprintf(stderr, "ref_me: cannot load %s: %s\n", lib, dlerror());
        exit(2);
    }
    *(void **)&s->create    = shim_sym(s, "svtme_create");
    *(void **)&s->destroy   = shim_sym(s, "svtme_destroy");
    *(void **)&s->store_put = shim_sym(s, "svtme_store_put");
    *(void **)&s->submit    = shim_sym(s, "svtme_picture_submit");
    *(void **)&s->wait      = shim_sym(s, "svtme_picture_wait");
    *(void **)&s->done      = shim_sym(s, "svtme_picture_done");
    *(void **)&s->unpack    = shim_sym(s, "svtme_unpack_b64");
    *(void **)&s->tpl_search = shim_sym(s, "svtme_tpl_search");
    *(void **)&s->set_tile_rects = shim_sym(s, "svtme_set_tile_rects");
    svtme_config cfg;
    memset(&cfg, 0, sizeof(cfg));
    cfg.max_width     = job->hdr.width;
    cfg.max_height    = job->hdr.height;
    cfg.store_slots   = job->hdr.n_frames + 1;
    cfg.max_in_flight = 2;
    if (s->create(&cfg, &s->engine) != SVTME_OK) die("svtme_create failed");
    /* the device picture store: one put per source picture, keyed like the EbPaReferenceObject it mirrors */
    for (uint32_t i = 0; i < job->hdr.n_frames; ++i)
        if (s->store_put(s->engine, (uint64_t)(uintptr_t)&job->planes[i], i, job->frames + (size_t)job->hdr.width * job->hdr.height * i,
                         job->hdr.width, job->hdr.width, job->hdr.height) != SVTME_OK)
            die("svtme_store_put failed");
}

static void run_pic_shim(Shim *s, const Job *job, uint32_t pi, PicState *st) {
    const ref_job_pic        *jp  = &job->pics[pi];
    PictureParentControlSet  *pcs = st->pcs;
    SequenceControlSet       *scs = st->scs;
    MotionEstimationContext_t mectx;
    memset(&mectx, 0, sizeof(mectx));
    MeContext *m = (MeContext *)calloc(1, sizeof(MeContext));
    if (!m) die("out of memory");
    me_context_ctor(m);
    mectx.me_context_ptr = m;
    if (jp->me_type == ME_MCTF) {
        m->me_type = ME_MCTF;
        tf_signal_derivation_me_kernel_oq(pcs, &mectx);
        m->hme_search_method = FULL_SAD_SEARCH;
    } else if (jp->me_type == ME_FIRST_PASS) {
        m->me_type = ME_FIRST_PASS;
        first_pass_signal_derivation_me_kernel(scs, pcs, &mectx);
    } else {
        m->me_type = ME_OPEN_LOOP;
        signal_derivation_me_kernel_oq(scs, pcs, &mectx);
    }
    /* ---- once per picture: the job description from the pcs and from the controls the reference just derived */
    svtme_picture d;
    memset(&d, 0, sizeof(d));
    d.cur_key        = (uint64_t)(uintptr_t)&job->planes[jp->cur_frame];
    d.picture_number = pcs->picture_number;
    d.width          = pcs->aligned_width;
    d.height         = pcs->aligned_height;
    d.temporal_layer_index      = pcs->temporal_layer_index;
    d.is_used_as_reference_flag = pcs->is_used_as_reference_flag;
    d.num_of_list_to_search       = jp->n_list;
    d.num_of_ref_pic_to_search[0] = jp->n_ref[0];
    d.num_of_ref_pic_to_search[1] = jp->n_list == 2 ? jp->n_ref[1] : 0;
    d.me_type = (uint8_t)m->me_type;
    d.bypass_zz_check          = m->me_type == ME_FIRST_PASS ? scs->ipp_pass_ctrls.bypass_zz_check : 0;
    d.restricted_motion_vector = scs->static_config.restricted_motion_vector;
    d.tf_me_exit_th            = m->me_type == ME_MCTF ? jp->tf_me_exit_th : 0;
    d.b64_step                 = (uint8_t)job->hdr.b64_step; /* MeContext::bypass_blk_step */
    for (int l = 0; l < jp->n_list; ++l)
        for (int r = 0; r < jp->n_ref[l]; ++r) {
            d.ref_key[l][r]            = (uint64_t)(uintptr_t)&job->planes[jp->ref_frame[l][r]];
            d.ref_picture_number[l][r] = jp->ref_poc[l][r];
        }
    fill_controls(&d.ctrl, m);
    /* once per sequence, where resource coordination has filled scs->sb_params_array's tiles
     * (EbResourceCoordinationProcess.c:1304-1349): the four tile fields of every 64x64 block */
    if (scs->static_config.restricted_motion_vector && !s->tiles_set) {
        svtme_tile_rect *t = (svtme_tile_rect *)calloc(st->n_b64, sizeof(*t));
        if (!t) die("out of memory");
        for (uint32_t i = 0; i < st->n_b64; ++i) {
            const SbParams *sp = &scs->sb_params_array[i];
            t[i].start_x = sp->tile_start_x, t[i].start_y = sp->tile_start_y, t[i].end_x = sp->tile_end_x, t[i].end_y = sp->tile_end_y;
        }
        if (s->set_tile_rects(s->engine, pcs->aligned_width, pcs->aligned_height, t, st->n_b64) != SVTME_OK) die("svtme_set_tile_rects failed");
        free(t);
        s->tiles_set = 1;
    }
    svtme_ticket  ticket;
    svtme_results res;
    if (s->submit(s->engine, &d, &ticket) != SVTME_OK) die("svtme_picture_submit failed");
    if (s->wait(s->engine, ticket, &res) != SVTME_OK) die("svtme_picture_wait failed");

    ref_pic_dump *pd = (ref_pic_dump *)st->out;
    pd->ctrl            = d.ctrl;
    pd->enable_me_8x8   = pcs->enable_me_8x8;
    pd->enable_me_16x16 = pcs->enable_me_16x16;
    pd->max_cand = pcs->pa_me_data->max_cand;
    pd->max_refs = pcs->pa_me_data->max_refs;
    pd->max_l0   = pcs->pa_me_data->max_l0;
    pd->input_resolution = (uint8_t)scs->input_resolution;
    pd->n_b64 = st->n_b64;

    /* ---- per 64x64 block, where motion_estimation_kernel calls motion_estimation_b64 */
    Planes *cur = &job->planes[jp->cur_frame];
    for (uint32_t b64 = 0; b64 < st->n_b64; ++b64) {
        const uint32_t ox = (b64 % st->b64_w) * 64, oy = (b64 / st->b64_w) * 64;
        if (d.b64_step > 1 && ((b64 % st->b64_w) % d.b64_step != 0 || (b64 / st->b64_w) % d.b64_step != 0)) continue; /* firstpass.c:1282-1285 */
        m->me_type = (EbMeType)d.me_type;
        m->tf_me_exit_th             = jp->tf_me_exit_th;
        m->tf_use_pred_64x64_only_th = 0;
        m->num_of_list_to_search       = jp->n_list;
        m->num_of_ref_pic_to_search[0] = jp->n_ref[0];
        if (jp->n_list == 2) m->num_of_ref_pic_to_search[1] = jp->n_ref[1];
        m->temporal_layer_index      = jp->temporal_layer;
        m->is_used_as_reference_flag = jp->is_ref;
        for (int l = 0; l < jp->n_list; ++l)
            for (int r = 0; r < jp->n_ref[l]; ++r) {
                Planes *rp = &job->planes[jp->ref_frame[l][r]];
                m->me_ds_ref_array[l][r].picture_ptr           = &rp->full;
                m->me_ds_ref_array[l][r].quarter_picture_ptr   = &rp->quarter;
                m->me_ds_ref_array[l][r].sixteenth_picture_ptr = &rp->sixteenth;
                m->me_ds_ref_array[l][r].picture_number        = jp->ref_poc[l][r];
            }
        m->block_width  = (cur->full.width - ox) < BLOCK_SIZE_64 ? cur->full.width - ox : BLOCK_SIZE_64; /* :2944-2949 */
        m->block_height = (cur->full.height - oy) < BLOCK_SIZE_64 ? cur->full.height - oy : BLOCK_SIZE_64;
        /* in place of init_me_hme_data, hme_b64, hme_prune_ref_and_adjust_sr and integer_search_b64 (:2954-2970) */
        if (s->unpack(&res, b64, &m->p_sb_best_sad[0][0][0], &m->p_sb_best_mv[0][0][0],
                      (svtme_search_result *)&m->search_results[0][0], &m->zz_sad[0][0],
                      &m->reduce_me_sr_divisor[0][0]) != SVTME_OK)
            die("svtme_unpack_b64 failed");
        const uint8_t prune_ref = m->enable_hme_flag && m->me_type != ME_MCTF;
        if (m->me_type == ME_MCTF && ((const svtme_search_result *)&m->search_results[0][0])->tf_me_exit) {
            m->tf_use_pred_64x64_only_th = (uint8_t)~0; /* :2958-2962 */
        } else {
            /* the rest of motion_estimation_b64, :2972-3000 */
            if (prune_ref && m->me_hme_prune_ctrls.enable_me_hme_ref_pruning) me_prune_ref(m);
            if (m->me_type != ME_MCTF) {
                if (m->num_of_ref_pic_to_search[REF_LIST_0] == 1 && m->num_of_ref_pic_to_search[REF_LIST_1] == 1)
                    construct_me_candidate_array_mrp_off(pcs, m, m->num_of_list_to_search, b64);
                else
                    construct_me_candidate_array(pcs, m, m->num_of_list_to_search, b64);
                if (m->me_type != ME_FIRST_PASS) compute_distortion(pcs, b64, m);
                pcs->stationary_block_present_sb[b64] = 0;
                pcs->rc_me_allow_gm[b64]              = 0;
                if (pcs->gm_ctrls.enabled) perform_gm_detection(pcs, b64, m);
            }
        }
        dump_b64((ref_b64_dump *)(st->out + sizeof(ref_pic_dump)) + b64, m, st, b64);
    }
    if (s->done(s->engine, ticket) != SVTME_OK) die("svtme_picture_done failed");
    free(m->p_eight_pos_sad16x16);
    free(m);
}


/* ---------------------------------------------------------------------------------------------------------------
 * REF_MODE_TPL: the TPL dispenser of one picture (tpl_disp_kernel's loop over the 64x64 blocks,
 * EbSourceBasedOperationsProcess.c:2105-2116) on the ME results run_pic just left in pcs->pa_me_data, every reference
 * outside the sliding window (the source pictures are the references of the reconstruction path too), which makes a
 * picture's run independent of the others. The cost functions are reached through the reference's own function
 * pointers, which lets the driver keep every cost per block. */
static const uint8_t k_tpl_blk_cu[21] = {0, 1, 22, 43, 64, 2, 7, 12, 17, 23, 28, 33, 38, 44, 49, 54, 59, 65, 70, 75, 80}; /* tpl_blk_idx_tab[0], :514-516 */

typedef struct TplLog {
    const uint8_t *plane0; /* pixel (0,0) of the current picture's full plane */
    uint32_t       stride, w16;
    int32_t        cur; /* 16x16 cell of the block at work */
    uint8_t       *n;
    int32_t       *costs; /* [cell][REF_TPL_MAX_COSTS] */
    uint32_t (*sad)(const uint8_t *, uint32_t, const uint8_t *, uint32_t, uint32_t, uint32_t);
    int (*satd)(const TranLow *, int);
    void (*subtract)(int, int, int16_t *, ptrdiff_t, const uint8_t *, ptrdiff_t, const uint8_t *, ptrdiff_t);
} TplLog;
static TplLog g_tpl;

static void tpl_log_block(const uint8_t *src_mb) {
    const ptrdiff_t off = src_mb - g_tpl.plane0;
    g_tpl.cur = (int32_t)((off / g_tpl.stride) >> 4) * (int32_t)g_tpl.w16 + (int32_t)((off % g_tpl.stride) >> 4);
}
static void tpl_log_cost(int32_t c) {
    if (g_tpl.cur < 0) return;
    if (g_tpl.n[g_tpl.cur] >= REF_TPL_MAX_COSTS) die("more costs per block than REF_TPL_MAX_COSTS");
    g_tpl.costs[(size_t)g_tpl.cur * REF_TPL_MAX_COSTS + g_tpl.n[g_tpl.cur]++] = c;
}
static uint32_t tpl_rec_sad(const uint8_t *src, uint32_t src_stride, const uint8_t *ref, uint32_t ref_stride, uint32_t h, uint32_t w) {
    tpl_log_block(src);
    const uint32_t v = g_tpl.sad(src, src_stride, ref, ref_stride, h, w);
    tpl_log_cost((int32_t)v);
    return v;
}
static void tpl_rec_subtract(int rows, int cols, int16_t *diff, ptrdiff_t diff_stride, const uint8_t *src, ptrdiff_t src_stride,
                             const uint8_t *pred, ptrdiff_t pred_stride) {
    tpl_log_block(src);
    g_tpl.subtract(rows, cols, diff, diff_stride, src, src_stride, pred, pred_stride);
}
static int tpl_rec_satd(const TranLow *coeff, int length) {
    const int v = g_tpl.satd(coeff, length);
    tpl_log_cost(v);
    return v;
}

typedef struct TplState {
    EncodeContext      *enc;
    EbPictureBufferDesc recon;
    EbObjectWrapper     scs_wrapper;
    uint32_t            w16, h16, n_stats;
    int32_t             qindex;
} TplState;

static void tpl_setup(const Job *job, uint32_t pi, PicState *st, TplState *t) {
    const ref_job_pic       *jp  = &job->pics[pi];
    PictureParentControlSet *pcs = st->pcs;
    SequenceControlSet      *scs = st->scs;
    const uint32_t           w = job->hdr.width, h = job->hdr.height;
    memset(t, 0, sizeof(*t));
    t->w16 = (w + 15) >> 4;
    t->h16 = (h + 15) >> 4;
    t->scs_wrapper.object_ptr = scs;
    pcs->scs_wrapper_ptr      = &t->scs_wrapper;
    pcs->enhanced_picture_ptr = &job->planes[jp->cur_frame].full;
    pcs->hierarchical_levels  = 4;
    scs->static_config.hierarchical_levels = 4;
    scs->static_config.qp     = jp->tpl_qp;
    scs->lad_mg     = 1;
    scs->tpl_lad_mg = 1; /* keeps the source-based results in tpl_src_stats_buffer (:983-992) */
    scs->in_loop_ois = 1; /* EbEncHandle.c:3495 */
    scs->seq_header.max_frame_width  = (uint16_t)w;
    scs->seq_header.max_frame_height = (uint16_t)h;
    scs->seq_header.enable_intra_edge_filter = 1;
    scs->enable_paeth = scs->enable_smooth = DEFAULT;
    set_tpl_extended_controls(pcs, jp->tpl_level);
    svt_av1_build_quantizer(AOM_BITS_8, 0, 0, 0, 0, 0, &scs->quants_8bit, &scs->deq_8bit);
    t->qindex = quantizer_to_qindex[jp->tpl_qp]; /* tpl_mc_flow_dispenser :1355, enable_tpl_qps is 0 at every level */

    TPLData *td = &pcs->tpl_data;
    td->tpl_slice_type            = pcs->slice_type;
    td->tpl_temporal_layer_index  = jp->temporal_layer;
    td->is_used_as_reference_flag = jp->is_ref;
    td->base_pcs                  = pcs;
    td->tpl_ref0_count            = jp->n_ref[0]; /* tpl_regular_setup_me_refs, EbPictureDecisionProcess.c:4955-4996 */
    td->tpl_ref1_count            = jp->n_list == 2 ? jp->n_ref[1] : 0;
    for (int l = 0; l < jp->n_list; ++l)
        for (int r = 0; r < jp->n_ref[l]; ++r) {
            td->ref_in_slide_window[l][r] = EB_FALSE;
            td->ref_tpl_group_idx[l][r]   = -1;
            td->tpl_ref_ds_ptr_array[l][r].picture_ptr    = &job->planes[jp->ref_frame[l][r]].full;
            td->tpl_ref_ds_ptr_array[l][r].picture_number = jp->ref_poc[l][r];
        }
    pcs->tpl_src_data_ready = 0;

    MotionEstimationData *md = pcs->pa_me_data;
    md->tpl_src_stats_buffer = (TplSrcStats *)calloc((size_t)t->w16 * t->h16, sizeof(TplSrcStats));
    for (uint32_t i = 0; i < t->w16 * t->h16; ++i) md->tpl_src_stats_buffer[i].best_rf_idx = -99; /* "not written" */
    t->n_stats    = (t->w16 * 2) * (t->h16 * 2); /* the finest grid result_model_store writes (8x8, :253-287) */
    md->tpl_stats = (TplStats **)calloc(t->n_stats, sizeof(TplStats *)); /* pointers into one array: a block's entry is */
    TplStats *all = (TplStats *)calloc(t->n_stats + 4 * t->w16 * 2 + 4, sizeof(TplStats)); /* followed in memory by its */
    for (uint32_t i = 0; i < t->n_stats; ++i) md->tpl_stats[i] = all + i; /* neighbours' (result_model_store) */

    desc_init(&t->recon, w, h, TPL_PADX); /* :1730-1745 */
    t->enc = (EncodeContext *)calloc(1, sizeof(EncodeContext));
    if (!t->enc || !md->tpl_src_stats_buffer) die("out of memory");
    t->enc->mc_flow_rec_picture_buffer[0] = &t->recon;

    g_tpl.plane0 = pcs->enhanced_picture_ptr->buffer_y + pcs->enhanced_picture_ptr->origin_x +
        (size_t)pcs->enhanced_picture_ptr->origin_y * pcs->enhanced_picture_ptr->stride_y;
    g_tpl.stride = pcs->enhanced_picture_ptr->stride_y;
    g_tpl.w16    = t->w16;
    g_tpl.cur    = -1;
    g_tpl.n      = (uint8_t *)calloc((size_t)t->w16 * t->h16, 1);
    g_tpl.costs  = (int32_t *)calloc((size_t)t->w16 * t->h16 * REF_TPL_MAX_COSTS, sizeof(int32_t));
}

static void tpl_dump(const Job *job, uint32_t pi, PicState *st, TplState *t, FILE *o) {
    PictureParentControlSet *pcs = st->pcs;
    const TplControls       *tc  = &pcs->tpl_ctrls;
    const uint32_t           w = job->hdr.width, h = job->hdr.height;
    ref_tpl_pic_dump pd;
    memset(&pd, 0, sizeof(pd));
    pd.tpl_opt_flag = tc->tpl_opt_flag;
    pd.use_pred_sad_in_intra_search = tc->use_pred_sad_in_intra_search;
    pd.use_pred_sad_in_inter_search = tc->use_pred_sad_in_inter_search;
    pd.dispenser_search_level = tc->dispenser_search_level;
    pd.subsample_tx = tc->subsample_tx;
    pd.pf_shape     = (uint8_t)tc->pf_shape;
    pd.disable_intra_pred = tc->disable_intra_pred_nref && pcs->tpl_data.is_used_as_reference_flag == 0;
    pd.synth_blk_size = tc->synth_blk_size;
    pd.qindex = (uint32_t)t->qindex;
    uint64_t hash = 0xcbf29ce484222325ull;
    for (uint32_t y = 0; y < h; ++y) {
        const uint8_t *row = t->recon.buffer_y + (size_t)(y + t->recon.origin_y) * t->recon.stride_y + t->recon.origin_x;
        for (uint32_t x = 0; x < w; ++x) hash = (hash ^ row[x]) * 0x100000001b3ull;
    }
    pd.recon_hash = hash;
    for (int k = 0; k < 2; ++k) {
        pd.y_round_fp[k] = st->scs->quants_8bit.y_round_fp[t->qindex][k];
        pd.y_quant_fp[k] = st->scs->quants_8bit.y_quant_fp[t->qindex][k];
        pd.y_dequant[k]  = st->scs->deq_8bit.y_dequant_qtx[t->qindex][k];
    }
    fwrite(&pd, sizeof(pd), 1, o);
    for (uint32_t b64 = 0; b64 < st->n_b64; ++b64) {
        const SbParams *sp = &pcs->sb_params_array[b64];
        for (int bi = 0; bi < 21; ++bi) {
            const CodedBlockStats *bs = get_coded_blk_stats(k_tpl_blk_cu[bi]);
            const uint32_t x = sp->origin_x + bs->origin_x, y = sp->origin_y + bs->origin_y, cell = (y >> 4) * t->w16 + (x >> 4);
            ref_tpl_blk_dump d;
            memset(&d, 0, sizeof(d));
            const uint8_t level = (sp->width == 64 && sp->height == 64) ? tc->dispenser_search_level : 0;
            const int in_level  = level == 0 ? bi >= 5 : (level == 1 ? (bi >= 1 && bi <= 4) : bi == 0);
            const TplSrcStats *ss = &pcs->pa_me_data->tpl_src_stats_buffer[cell];
            if (in_level && x < w && y < h && ss->best_rf_idx != -99) {
                d.valid           = 1;
                d.best_mode       = ss->best_mode;
                d.best_intra_mode = (uint8_t)ss->best_intra_mode;
                d.best_rf_idx     = ss->best_rf_idx;
                d.mv_row          = ss->mv.row;
                d.mv_col          = ss->mv.col;
                d.ref_frame_poc   = ss->ref_frame_poc;
                d.src_srcrf_dist  = ss->srcrf_dist;
                d.src_srcrf_rate  = ss->srcrf_rate;
                d.n_costs         = g_tpl.n[cell];
                memcpy(d.costs, &g_tpl.costs[(size_t)cell * REF_TPL_MAX_COSTS], sizeof(d.costs));
                const TplStats *ts; /* result_model_store's grids, :223-287 */
                if (tc->synth_blk_size == 32) ts = pcs->pa_me_data->tpl_stats[(y >> 5) * ((pcs->aligned_width + 31) / 32) + (x >> 5)];
                else if (tc->synth_blk_size == 16) ts = pcs->pa_me_data->tpl_stats[(y >> 4) * ((pcs->aligned_width + 15) / 16) + (x >> 4)];
                else ts = pcs->pa_me_data->tpl_stats[(y >> 3) * (((pcs->aligned_width + 15) / 16) << 1) + (x >> 3)];
                d.srcrf_dist = ts->srcrf_dist; d.recrf_dist = ts->recrf_dist;
                d.srcrf_rate = ts->srcrf_rate; d.recrf_rate = ts->recrf_rate;
                d.stats_mv_row = ts->mv.row;   d.stats_mv_col = ts->mv.col;
                d.stats_ref_frame_poc = ts->ref_frame_poc;
            }
            fwrite(&d, sizeof(d), 1, o);
        }
    }
}

static void tpl_free(PicState *st, TplState *t) {
    MotionEstimationData *md = st->pcs->pa_me_data;
    free(md->tpl_stats[0]);
    free(md->tpl_stats);
    free(md->tpl_src_stats_buffer);
    free(t->recon.buffer_y);
    free(t->enc);
    free(g_tpl.n);
    free(g_tpl.costs);
    g_tpl.n = NULL; g_tpl.costs = NULL;
}


/* REF_MODE_TPL with the engine: what a patched dispenser does with svtme_tpl_search's answers. The search of :634-933 is
 * the engine's; the encoder keeps :935-992 (transform and quantisation of the best inter candidate for srcrf_dist) and
 * leaves the block's source-based results where the reference itself keeps them between TPL groups
 * (tpl_src_stats_buffer, :983-992), so that the unmodified dispenser, run with tpl_src_data_ready = 1, takes them from
 * there (:993-1002) and does its reconstruction path. */
static const uint8_t k_tpl_me_idx[21] = {0, 1, 2, 3, 4, 5, 6, 9, 10, 7, 8, 11, 12, 13, 14, 17, 18, 15, 16, 19, 20};
static const TxSize  k_tpl_tx[3][3]   = {{TX_16X16, TX_32X32, TX_64X64}, {TX_16X8, TX_32X16, TX_64X32}, {TX_16X4, TX_32X8, TX_64X16}}; /* :543-545 */

static void tpl_fill_from_engine(Shim *s, const Job *job, uint32_t pi, PicState *st, TplState *t) {
    const ref_job_pic       *jp  = &job->pics[pi];
    PictureParentControlSet *pcs = st->pcs;
    SequenceControlSet      *scs = st->scs;
    const TplControls       *tc  = &pcs->tpl_ctrls;
    MotionEstimationData    *md  = pcs->pa_me_data;
    const uint32_t           n_pu = SQUARE_PU_COUNT;

    /* the picture's ME results, one 64x64 block after the other, as svtme_tpl_search reads them */
    uint8_t  *cnt  = (uint8_t *)calloc((size_t)st->n_b64 * n_pu, 1);
    uint8_t  *cand = (uint8_t *)calloc((size_t)st->n_b64 * n_pu * md->max_cand, 1);
    uint32_t *mv   = (uint32_t *)calloc((size_t)st->n_b64 * n_pu * md->max_refs, 4);
    for (uint32_t b = 0; b < st->n_b64; ++b) {
        const MeSbResults *r = md->me_results[b];
        memcpy(cnt + (size_t)b * n_pu, r->total_me_candidate_index, n_pu);
        memcpy(cand + (size_t)b * n_pu * md->max_cand, r->me_candidate_array, (size_t)n_pu * md->max_cand);
        memcpy(mv + (size_t)b * n_pu * md->max_refs, r->me_mv_array, (size_t)n_pu * md->max_refs * 4);
    }
    svtme_tpl_picture d;
    memset(&d, 0, sizeof(d));
    d.cur_key = (uint64_t)(uintptr_t)&job->planes[jp->cur_frame];
    d.width   = pcs->enhanced_picture_ptr->width;
    d.height  = pcs->enhanced_picture_ptr->height;
    d.n_ref[0] = pcs->tpl_data.tpl_ref0_count;
    d.n_ref[1] = pcs->tpl_data.tpl_ref1_count;
    for (int l = 0; l < 2; ++l)
        for (int r = 0; r < d.n_ref[l]; ++r) {
            d.ref_key[l][r] = (uint64_t)(uintptr_t)&job->planes[jp->ref_frame[l][r]];
            const int32_t g = pcs->tpl_data.ref_tpl_group_idx[l][r];
            d.ref_skip[l][r] = g > 0 && pcs->tpl_data.base_pcs->tpl_valid_pic[g] == 0; /* :846-849 */
        }
    d.i_slice                      = pcs->slice_type == I_SLICE;
    d.dispenser_search_level       = tc->dispenser_search_level;
    d.disable_intra_pred           = tc->disable_intra_pred_nref && pcs->tpl_data.is_used_as_reference_flag == 0;
    d.tpl_opt_flag                 = tc->tpl_opt_flag;
    d.use_pred_sad_in_intra_search = tc->use_pred_sad_in_intra_search;
    d.use_pred_sad_in_inter_search = tc->use_pred_sad_in_inter_search;
    d.enable_me_16x16              = pcs->enable_me_16x16;
    d.subsample_tx                 = tc->subsample_tx;
    d.pf_shape                     = (uint8_t)tc->pf_shape;
    { /* the last intra mode the dispenser tries without tpl_opt_flag (:724-733) */
        const EbBool paeth  = scs->enable_paeth == DEFAULT ? EB_TRUE : (EbBool)scs->enable_paeth;
        const EbBool smooth = scs->enable_smooth == DEFAULT ? EB_TRUE : (EbBool)scs->enable_smooth;
        d.intra_mode_end    = paeth ? PAETH_PRED : (smooth ? SMOOTH_H_PRED : D67_PRED);
    }
    /* the engine also transforms and quantises the candidate it chose, for the transforms the tpl levels use (16x16, 16x8,
     * 16x4 on 16x16 blocks, 32x8 on 32x32 blocks); for another combination the lines below do it with the reference's own */
    d.src_dist = tc->dispenser_search_level == 0 || (tc->dispenser_search_level == 1 && tc->subsample_tx == 2);
    for (int k = 0; k < 2; ++k) {
        d.y_round_fp[k] = scs->quants_8bit.y_round_fp[t->qindex][k];
        d.y_quant_fp[k] = scs->quants_8bit.y_quant_fp[t->qindex][k];
        d.y_dequant[k]  = scs->deq_8bit.y_dequant_qtx[t->qindex][k];
    }
    d.n_pu     = (uint8_t)n_pu;
    d.max_cand = md->max_cand;
    d.max_refs = md->max_refs;
    d.max_l0   = md->max_l0;
    d.total_me_candidate_index = cnt;
    d.me_candidate_array       = cand;
    d.me_mv_array              = mv;
    svtme_tpl_block *blk = (svtme_tpl_block *)calloc((size_t)st->n_b64 * SVTME_TPL_BLOCKS, sizeof(*blk));
    if (s->tpl_search(s->engine, &d, blk) != SVTME_OK) die("svtme_tpl_search failed");

    MacroblockPlane mbp; /* :579-586 */
    mbp.quant_qtx = scs->quants_8bit.y_quant[t->qindex];       mbp.quant_fp_qtx = scs->quants_8bit.y_quant_fp[t->qindex];
    mbp.round_fp_qtx = scs->quants_8bit.y_round_fp[t->qindex]; mbp.quant_shift_qtx = scs->quants_8bit.y_quant_shift[t->qindex];
    mbp.zbin_qtx = scs->quants_8bit.y_zbin[t->qindex];         mbp.round_qtx = scs->quants_8bit.y_round[t->qindex];
    mbp.dequant_qtx = scs->deq_8bit.y_dequant_qtx[t->qindex];
    const EbPictureBufferDesc *in = pcs->enhanced_picture_ptr;
    static int16_t src_diff[64 * 64];
    static TranLow coeff[64 * 64], qcoeff[64 * 64], dqcoeff[64 * 64];
    for (uint32_t b = 0; b < st->n_b64; ++b) {
        const SbParams *sp    = &pcs->sb_params_array[b];
        const int       level = (sp->width == 64 && sp->height == 64) ? tc->dispenser_search_level : 0;
        const uint32_t  size  = 16u << level;
        const TxSize    tx    = k_tpl_tx[tc->subsample_tx][level];
        for (int bi = 0; bi < SVTME_TPL_BLOCKS; ++bi) {
            const svtme_tpl_block *e = &blk[(size_t)b * SVTME_TPL_BLOCKS + bi];
            if (!e->valid) continue;
            const CodedBlockStats *bs = get_coded_blk_stats(k_tpl_blk_cu[bi]);
            const uint32_t x = sp->origin_x + bs->origin_x, y = sp->origin_y + bs->origin_y;
            TplSrcStats   *ss = &md->tpl_src_stats_buffer[(y >> 4) * t->w16 + (x >> 4)];
            memset(ss, 0, sizeof(*ss));
            ss->best_rf_idx     = e->best_rf_idx;
            ss->mv.row          = e->mv_y;
            ss->mv.col          = e->mv_x;
            ss->best_mode       = e->best_is_inter ? NEWMV : DC_PRED;
            ss->best_intra_mode = (PredictionMode)e->best_intra_mode;
            if (e->best_rf_idx >= 0) {
                const int l = e->best_rf_idx < 4 ? 0 : 1, r = e->best_rf_idx >= 4 ? e->best_rf_idx - 4 : e->best_rf_idx;
                ss->ref_frame_poc = pcs->tpl_data.tpl_ref_ds_ptr_array[l][r].picture_number;
                if (e->best_is_inter && d.src_dist) {
                    /* rate_estimator's value without tpl_opt_flag, else no rate (:975-979) */
                    ss->srcrf_rate = tc->tpl_opt_flag ? 0 : (e->src_rate << TPL_DEP_COST_SCALE_LOG2) << tc->subsample_tx;
                    ss->srcrf_dist = (e->src_recon_error << TPL_DEP_COST_SCALE_LOG2) << tc->subsample_tx; /* :980-981 */
                } else if (e->best_is_inter) { /* :935-982 on the host */
                    const EbPictureBufferDesc *rp = pcs->tpl_data.tpl_ref_ds_ptr_array[l][r].picture_ptr;
                    const uint8_t *src_mb = in->buffer_y + in->origin_x + x + (size_t)(in->origin_y + y) * in->stride_y;
                    const int32_t  ro = (int32_t)rp->origin_x + ((int32_t)x + (e->mv_x >> 3)) +
                        ((int32_t)y + (e->mv_y >> 3) + (int32_t)rp->origin_y) * (int32_t)rp->stride_y;
                    svt_aom_subtract_block(size >> tc->subsample_tx, size, src_diff, size << tc->subsample_tx, src_mb,
                                           in->stride_y << tc->subsample_tx, rp->buffer_y + ro, rp->stride_y << tc->subsample_tx);
                    svt_av1_wht_fwd_txfm(src_diff, size << tc->subsample_tx, coeff, tx, tc->tpl_opt_flag ? tc->pf_shape : DEFAULT_SHAPE, 8, 0);
                    /* get_quantize_error, :167-193 */
                    const ScanOrder *so = &av1_scan_orders[tx][DCT_DCT];
                    const int pix_num = 1 << num_pels_log2_lookup[txsize_to_bsize[tx]], shift = tx == TX_32X32 ? 0 : 2;
                    uint16_t  eob = 0;
                    int64_t   sse;
                    svt_av1_quantize_fp(coeff, pix_num, mbp.zbin_qtx, mbp.round_fp_qtx, mbp.quant_fp_qtx, mbp.quant_shift_qtx, qcoeff, dqcoeff,
                                        mbp.dequant_qtx, &eob, so->scan, so->iscan);
                    int64_t recon_error = svt_av1_block_error(coeff, dqcoeff, pix_num, &sse) >> shift;
                    if (recon_error < 1) recon_error = 1;
                    ss->srcrf_rate = 0; /* tpl_opt_flag: no rate (:975-979) */
                    ss->srcrf_dist = (recon_error << TPL_DEP_COST_SCALE_LOG2) << tc->subsample_tx;
                }
            }
        }
    }
    pcs->tpl_src_data_ready = 1;
    free(blk); free(cnt); free(cand); free(mv);
}

static double now_s(void);
static double g_tpl_seconds[2]; /* the binding's part (engine call, transform and quantisation of the choices), the dispenser */

static void run_pic_tpl(const Job *job, uint32_t pi, PicState *st, FILE *o, Shim *shim) {
    TplState t;
    tpl_setup(job, pi, st, &t);
    const double t0 = now_s();
    if (shim) tpl_fill_from_engine(shim, job, pi, st, &t);
    const double t1 = now_s();
    g_tpl_seconds[0] += t1 - t0;
    for (uint32_t sb = 0; sb < st->n_b64; ++sb) {
        const SbParams *sp = &st->scs->sb_params_array[sb];
        tpl_mc_flow_dispenser_sb_generic(t.enc, st->scs, st->pcs, 0, sb, t.qindex,
                                         (sp->width == 64 && sp->height == 64) ? st->pcs->tpl_ctrls.dispenser_search_level : 0);
    }
    g_tpl_seconds[1] += now_s() - t1;
    tpl_dump(job, pi, st, &t, o);
    tpl_free(st, &t);
}

static double now_s(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec + ts.tv_nsec * 1e-9;
}

static void run_pic(const Job *job, uint32_t pi, PicState *st, uint32_t threads, int dump) {
    if (threads < 1) threads = 1;
    if (threads > st->b64_h) threads = st->b64_h;
    pthread_t *tid = (pthread_t *)calloc(threads, sizeof(pthread_t));
    Worker    *wk  = (Worker *)calloc(threads, sizeof(Worker));
    for (uint32_t t = 0; t < threads; ++t) {
        wk[t].job = job; wk[t].st = st; wk[t].pi = pi; wk[t].dump = dump;
        wk[t].row_begin = (uint32_t)((uint64_t)st->b64_h * t / threads);
        wk[t].row_end   = (uint32_t)((uint64_t)st->b64_h * (t + 1) / threads);
        if (threads == 1) worker_main(&wk[t]);
        else pthread_create(&tid[t], NULL, worker_main, &wk[t]);
    }
    if (threads > 1)
        for (uint32_t t = 0; t < threads; ++t) pthread_join(tid[t], NULL);
    free(tid); free(wk);
}

/* ref_me dct <w> <h> <in> <out>: the reference's forward transform as the TPL dispenser calls it (svt_av1_wht_fwd_txfm, 8 bit)
 * on every w x h residual block of <in> (int16 [n][h][w]), for DEFAULT_SHAPE, N2_SHAPE and N4_SHAPE: int32 [3][n][h][w].
 * Sizes: the dispenser's own (tx_size_array / sub2_tx_size_array / sub4_tx_size_array, :543-545) up to 32 wide. */
static int dct_main(int w, int h, const char *in_path, const char *out_path) {
    TxSize tx;
    if (w == 16 && h == 16) tx = TX_16X16;
    else if (w == 16 && h == 8) tx = TX_16X8;
    else if (w == 16 && h == 4) tx = TX_16X4;
    else if (w == 32 && h == 8) tx = TX_32X8;
    else die("dct: size not among the dispenser's");
    const size_t px = (size_t)w * h;
    FILE *f = fopen(in_path, "rb");
    if (!f) die("cannot open the block file");
    fseek(f, 0, SEEK_END);
    const long n = ftell(f) / (long)(px * 2);
    fseek(f, 0, SEEK_SET);
    int16_t *blocks = (int16_t *)malloc((size_t)n * px * 2);
    if (fread(blocks, px * 2, (size_t)n, f) != (size_t)n) die("short block file");
    fclose(f);
    setup_common_rtcd_internal(0);
    setup_rtcd_internal(0);
    FILE *o = fopen(out_path, "wb");
    if (!o) die("cannot open output file");
    const EB_TRANS_COEFF_SHAPE shapes[3] = {DEFAULT_SHAPE, N2_SHAPE, N4_SHAPE};
    for (int s = 0; s < 3; ++s)
        for (long i = 0; i < n; ++i) {
            int32_t coeff[32 * 32];
            memset(coeff, 0, sizeof(coeff));
            svt_av1_wht_fwd_txfm(blocks + i * px, w, coeff, tx, shapes[s], 8, 0);
            fwrite(coeff, sizeof(int32_t), px, o);
        }
    fclose(o);
    free(blocks);
    return 0;
}

int main(int argc, char **argv) {
    if (argc == 6 && !strcmp(argv[1], "dct")) return dct_main(atoi(argv[2]), atoi(argv[3]), argv[4], argv[5]);
    if (argc < 3) die("usage: ref_me <job> <out>");
    FILE *f = fopen(argv[1], "rb");
    if (!f) die("cannot open job file");
    Job job;
    memset(&job, 0, sizeof(job));
    if (fread(&job.hdr, sizeof(job.hdr), 1, f) != 1) die("short job header");
    if (memcmp(job.hdr.magic, REF_JOB_MAGIC, 8)) die("bad job magic");
    const uint32_t w = job.hdr.width, h = job.hdr.height;
    if (w % 8 || h % 8 || w < 64 || h < 64) die("width/height must be multiples of 8 and >= 64");
    size_t fsz = (size_t)w * h;
    job.frames = (uint8_t *)malloc(fsz * job.hdr.n_frames);
    job.pics   = (ref_job_pic *)calloc(job.hdr.n_pics ? job.hdr.n_pics : 1, sizeof(ref_job_pic));
    if (fread(job.frames, fsz, job.hdr.n_frames, f) != job.hdr.n_frames) die("short frame data");
    if (job.hdr.n_pics && fread(job.pics, sizeof(ref_job_pic), job.hdr.n_pics, f) != job.hdr.n_pics) die("short pic data");
    fclose(f);

    setup_common_rtcd_internal(0);
    setup_rtcd_internal(0);
    int use_avx2 = 0;
#ifdef REF_HAVE_AVX2
    if (job.hdr.use_avx2 && __builtin_cpu_supports("avx2")) {
        svt_sad_loop_kernel                       = svt_sad_loop_kernel_avx2_intrin;
        svt_ext_all_sad_calculation_8x8_16x16     = svt_ext_all_sad_calculation_8x8_16x16_avx2;
        svt_ext_eight_sad_calculation_32x32_64x64 = svt_ext_eight_sad_calculation_32x32_64x64_avx2;
        svt_ext_sad_calculation_8x8_16x16         = svt_ext_sad_calculation_8x8_16x16_avx2_intrin;
        svt_nxm_sad_kernel                        = svt_nxm_sad_kernel_helper_avx2;
        use_avx2 = 1;
    }
#endif

    job.planes = (Planes *)calloc(job.hdr.n_frames, sizeof(Planes));
    for (uint32_t i = 0; i < job.hdr.n_frames; ++i) build_planes(&job.planes[i], job.frames + fsz * i, w, h);

    FILE *o = fopen(argv[2], "wb");
    if (!o) die("cannot open output file");
    ref_out_header oh;
    memset(&oh, 0, sizeof(oh));
    memcpy(oh.magic, REF_OUT_MAGIC, 8);
    oh.n_pics = job.hdr.n_pics;
    oh.n_b64  = ((w + 63) / 64) * ((h + 63) / 64);
    oh.pic_dump_size = sizeof(ref_pic_dump);
    oh.b64_dump_size = sizeof(ref_b64_dump);
    oh.used_avx2 = (uint32_t)use_avx2;
    oh.mode = job.hdr.mode;

    if (job.hdr.mode == REF_MODE_PLANES) {
        /* dump the padded planes of every frame: sixteenth, quarter, full */
        oh.n_pics = job.hdr.n_frames;
        fwrite(&oh, sizeof(oh), 1, o);
        for (uint32_t i = 0; i < job.hdr.n_frames; ++i) {
            EbPictureBufferDesc *d[3] = {&job.planes[i].sixteenth, &job.planes[i].quarter, &job.planes[i].full};
            for (int k = 0; k < 3; ++k) {
                uint32_t geo[4] = {d[k]->stride_y, d[k]->width, d[k]->height, d[k]->origin_x};
                fwrite(geo, sizeof(geo), 1, o);
                fwrite(d[k]->buffer_y, 1, d[k]->luma_size, o);
            }
        }
    } else if (job.hdr.mode == REF_MODE_SHIM) {
        if (argc < 4) die("REF_MODE_SHIM needs the path of libsvtme.so as third argument");
        Shim shim;
        shim_open(&shim, argv[3], &job);
        fwrite(&oh, sizeof(oh), 1, o);
        for (uint32_t pi = 0; pi < job.hdr.n_pics; ++pi) {
            PicState st;
            memset(&st, 0, sizeof(st));
            setup_pic(&job, pi, &st);
            run_pic_shim(&shim, &job, pi, &st);
            fwrite(st.out, 1, sizeof(ref_pic_dump) + (size_t)st.n_b64 * sizeof(ref_b64_dump), o);
            free_pic(&st);
        }
        shim.destroy(shim.engine);
    } else if (job.hdr.mode == REF_MODE_TPL) {
        init_intra_dc_predictors_c_internal();
        init_intra_predictors_internal();
        g_tpl.sad      = svt_nxm_sad_kernel_sub_sampled;
        g_tpl.satd     = svt_aom_satd;
        g_tpl.subtract = svt_aom_subtract_block;
        Shim shim, *use_shim = NULL;
        if (argc >= 4) { /* the source-based search from the GPU engine */
            shim_open(&shim, argv[3], &job);
            use_shim = &shim;
        }
        fwrite(&oh, sizeof(oh), 1, o);
        for (uint32_t pi = 0; pi < job.hdr.n_pics; ++pi) {
            PicState st;
            memset(&st, 0, sizeof(st));
            setup_pic(&job, pi, &st);
            run_pic(&job, pi, &st, job.hdr.threads, 1);
            fwrite(st.out, 1, sizeof(ref_pic_dump) + (size_t)st.n_b64 * sizeof(ref_b64_dump), o);
            svt_nxm_sad_kernel_sub_sampled = tpl_rec_sad;
            svt_aom_satd                   = tpl_rec_satd;
            svt_aom_subtract_block         = tpl_rec_subtract;
            run_pic_tpl(&job, pi, &st, o, use_shim);
            svt_nxm_sad_kernel_sub_sampled = g_tpl.sad;
            svt_aom_satd                   = g_tpl.satd;
            svt_aom_subtract_block         = g_tpl.subtract;
            free_pic(&st);
        }
        if (use_shim) shim.destroy(shim.engine);
        printf("{\"pictures\": %u, \"binding_seconds\": %.6f, \"dispenser_seconds\": %.6f, \"engine\": %d}\n", job.hdr.n_pics,
               g_tpl_seconds[0], g_tpl_seconds[1], use_shim != NULL);
    } else if (job.hdr.mode == REF_MODE_DUMP) {
        fwrite(&oh, sizeof(oh), 1, o);
        for (uint32_t pi = 0; pi < job.hdr.n_pics; ++pi) {
            PicState st;
            memset(&st, 0, sizeof(st));
            setup_pic(&job, pi, &st);
            run_pic(&job, pi, &st, job.hdr.threads, 1);
            fwrite(st.out, 1, sizeof(ref_pic_dump) + (size_t)st.n_b64 * sizeof(ref_b64_dump), o);
            free_pic(&st);
        }
    } else { /* REF_MODE_BENCH: time `reps` passes over all pictures, nothing dumped */
        PicState *st = (PicState *)calloc(job.hdr.n_pics, sizeof(PicState));
        for (uint32_t pi = 0; pi < job.hdr.n_pics; ++pi) setup_pic(&job, pi, &st[pi]);
        for (uint32_t pi = 0; pi < job.hdr.n_pics && pi < 2; ++pi) run_pic(&job, pi, &st[pi], job.hdr.threads, 0); /* warm-up */
        double   t0 = now_s();
        uint32_t reps = job.hdr.reps ? job.hdr.reps : 1;
        for (uint32_t r = 0; r < reps; ++r)
            for (uint32_t pi = 0; pi < job.hdr.n_pics; ++pi) run_pic(&job, pi, &st[pi], job.hdr.threads, 0);
        double t1 = now_s();
        oh.bench_seconds  = t1 - t0;
        oh.bench_pictures = (uint64_t)reps * job.hdr.n_pics;
        fwrite(&oh, sizeof(oh), 1, o);
        printf("{\"pictures\": %llu, \"seconds\": %.6f, \"threads\": %u, \"avx2\": %d}\n",
               (unsigned long long)oh.bench_pictures, oh.bench_seconds, job.hdr.threads, use_avx2);
        for (uint32_t pi = 0; pi < job.hdr.n_pics; ++pi) free_pic(&st[pi]);
        free(st);
    }
    fclose(o);
    return 0;
}
