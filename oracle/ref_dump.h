/*
 * ref_dump.h - TEST INFRASTRUCTURE: binary job / dump formats shared by oracle/ref_driver.c (the driver
 * around the real reference), oracle/oracle_cli.c (the same job run through the restatement) and the Python
 * readers in tests/. All fields little-endian, structs packed.
 */
#ifndef REF_DUMP_H
#define REF_DUMP_H
#include <stdint.h>
#include "../include/svtme.h"

#define REF_JOB_MAGIC "SVTMEJ01"
#define REF_OUT_MAGIC "SVTMEO01"
#define REF_MODE_DUMP 0
#define REF_MODE_BENCH 1
#define REF_MODE_PLANES 2
#define REF_MODE_DUMP_NOTAIL 3 /* restatement only: state right after integer_search_b64 (what the GPU engine hands over) */
#define REF_MODE_TAIL 4 /* restatement only: the host tail (me_prune_ref .. perform_gm_detection) run on handed-over
                           search results: a third argument names a file of n_pics * n_b64 ref_b64_dump records */
#define REF_MODE_SHIM 5 /* ref_me only: the reference-side binding end to end. The searches run on the GPU engine through
                           its C ABI (a third argument names libsvtme.so), every block's results are unpacked into a real
                           MeContext, and the reference's own me_prune_ref .. perform_gm_detection run on it */
#define REF_MODE_TPL 6 /* ref_me only: after each picture's ME, the TPL dispenser (tpl_mc_flow_dispenser_sb_generic,
                          EbSourceBasedOperationsProcess.c:547-1199) over all its 64x64 blocks with the controls of
                          ref_job_pic::tpl_level and the quantiser of ::tpl_qp; every picture's ME dump is followed by a
                          ref_tpl_pic_dump and n_b64 * 21 ref_tpl_blk_dump records. With a third argument (libsvtme.so) the
                          source-based search of every block comes from the GPU engine (svtme_tpl_search) and the dispenser
                          runs with tpl_src_data_ready = 1, i.e. only its reconstruction path */
#define REF_TPL_MAX_COSTS 24
#define REF_DUMP_MAX_REFS 7 /* MAX_PA_ME_MV, EbMotionEstimationLcuResults.h:23 */
#define REF_DUMP_MAX_CAND 23 /* MAX_PA_ME_CAND, :24 */

#pragma pack(push, 1)
typedef struct ref_job_header {
    char     magic[8];
    uint32_t width, height; /* luma, multiples of 8 */
    uint32_t n_frames; /* followed by n_frames * width * height luma bytes */
    uint32_t n_pics; /* then n_pics ref_job_pic */
    uint32_t mode;
    uint32_t threads;
    uint32_t use_avx2;
    uint32_t reps;
    uint32_t ref_count_used_list0; /* MRP maxima used to size me_results, EbPictureDecisionProcess.c:5075-5092 */
    uint32_t ref_count_used_list1;
    uint32_t b64_step; /* > 1: only the 64x64 blocks whose column and row index are multiples of it are searched, as
                          first_pass_me does with MeContext::bypass_blk_step (firstpass.c:1282-1285); the other blocks'
                          records stay zero */
} ref_job_header;

typedef struct ref_job_pic {
    uint32_t cur_frame;
    uint32_t frame_rate; /* 16.16 */
    uint64_t picture_number;
    int32_t  enc_mode;
    uint8_t  temporal_layer;
    uint8_t  is_ref;
    uint8_t  n_list;
    uint8_t  n_ref[2];
    uint8_t  sc_class1;
    uint8_t  gm_enabled;
    uint8_t  gm_dist_th;
    uint32_t ref_frame[2][4];
    uint64_t ref_poc[2][4];
    /* me_type 0 = ME_OPEN_LOOP. 1 = ME_MCTF (EbMotionEstimationContext.h:46), the temporal filter's search: controls from
     * tf_signal_derivation_me_kernel_oq with pcs->tf_ctrls.hme_me_level = tf_hme_me_level, MeContext::tf_me_exit_th.
     * 4 = ME_FIRST_PASS (:49), the first pass of a two-pass encode: controls from first_pass_signal_derivation_me_kernel
     * with scs->ipp_pass_ctrls.bypass_zz_check = tf_hme_me_level & 1 and .reduce_me_search = (tf_hme_me_level >> 1) & 1 */
    uint8_t  me_type;
    uint8_t  tf_hme_me_level;
    uint16_t tf_me_exit_th;
    uint8_t  restricted_motion_vector; /* scs->static_config.restricted_motion_vector */
    uint8_t  tpl_level; /* REF_MODE_TPL: set_tpl_extended_controls' level (EbResourceCoordinationProcess.c:161-341) */
    uint8_t  tpl_qp; /* REF_MODE_TPL: scs->static_config.qp */
    /* Uniform tiles of (tile_b64 & 15) x (tile_b64 >> 4) 64x64 blocks, 0 = the picture is one tile in that direction.
     * The drivers fill scs->sb_params_array[].tile_{start,end}_{x,y} with what EbResourceCoordinationProcess.c:1304-1349
     * leaves there for such a grid (tile ends clipped to the picture). With restricted_motion_vector the temporal filter's
     * and the first pass's integer searches keep their windows inside a block's tile (EbMotionEstimation.c:1405-1498); the
     * open-loop search reads pcs->sb_params_array, whose one tile is the picture (EbPictureControlSet.c:1684-1688). */
    uint8_t  tile_b64;
} ref_job_pic;

typedef struct ref_out_header {
    char     magic[8];
    uint32_t n_pics;
    uint32_t n_b64;
    uint32_t pic_dump_size;
    uint32_t b64_dump_size;
    uint32_t used_avx2;
    uint32_t mode;
    double   bench_seconds;
    uint64_t bench_pictures;
} ref_out_header;

typedef struct ref_pic_dump {
    svtme_controls ctrl; /* MeContext controls after signal_derivation_me_kernel_oq */
    uint8_t        enable_me_8x8, enable_me_16x16, max_cand, max_refs, max_l0, input_resolution;
    uint8_t        pad[2];
    uint32_t       n_b64;
} ref_pic_dump;

/* State of MeContext / pcs after motion_estimation_b64 returned for one b64. */
typedef struct ref_b64_dump {
    uint32_t best_sad[2][4][85];
    uint32_t best_mv[2][4][85];
    int16_t  hme_sc_x[2][4], hme_sc_y[2][4];
    uint64_t hme_sad[2][4]; /* final, i.e. after me_prune_ref when that ran */
    uint8_t  do_ref[2][4]; /* final */
    uint32_t zz_sad[2][4];
    uint32_t sr_divisor[2][4];
    int16_t  l0_x[2][4][2][2], l0_y[2][4][2][2];
    uint64_t l0_sad[2][4][2][2];
    int16_t  l1_x[2][4][2][2], l1_y[2][4][2][2];
    uint64_t l1_sad[2][4][2][2];
    int16_t  l2_x[2][4][2][2], l2_y[2][4][2][2];
    uint64_t l2_sad[2][4][2][2];
    int16_t  pre_x[2][4][2], pre_y[2][4][2];
    uint64_t pre_sad[2][4][2];
    /* host tail */
    uint32_t me_distortion[85];
    uint8_t  total_me_candidate_index[85];
    uint32_t me_mv_array[85 * REF_DUMP_MAX_REFS];
    uint8_t  me_candidate_array[85 * REF_DUMP_MAX_CAND];
    uint32_t rc_me_distortion, me_64x64_distortion, me_32x32_distortion, me_16x16_distortion,
        me_8x8_distortion, me_8x8_cost_variance;
    uint8_t stationary_block_present_sb, rc_me_allow_gm;
    /* ME_MCTF only: 1 when motion_estimation_b64 returned at the tf_me_exit_th test (:2958-2961), i.e. it set
     * tf_use_pred_64x64_only_th to ~0 and searched nothing; best_sad is then dumped as zeros (the reference leaves
     * whatever the previous block wrote). */
    uint8_t tf_me_exit;
} ref_b64_dump;

/* REF_MODE_TPL */
typedef struct ref_tpl_pic_dump {
    uint8_t  tpl_opt_flag, use_pred_sad_in_intra_search, use_pred_sad_in_inter_search, dispenser_search_level;
    uint8_t  subsample_tx, pf_shape, disable_intra_pred, synth_blk_size;
    uint32_t qindex;
    uint64_t recon_hash; /* FNV-1a over the reconstructed luma, row by row, after the last block */
    int16_t  y_round_fp[2], y_quant_fp[2], y_dequant[2]; /* [dc, ac] of scs->quants_8bit / deq_8bit at qindex (:579-586) */
    uint8_t  pad[4];
} ref_tpl_pic_dump;

/* One of the 21 blocks (tpl_blk_idx_tab order: 64x64, 4 x 32x32, 16 x 16x16) of a 64x64 block. */
typedef struct ref_tpl_blk_dump {
    uint8_t  valid; /* the dispenser processed this block */
    uint8_t  best_mode; /* TplSrcStats (:985-991) */
    uint8_t  best_intra_mode;
    uint8_t  n_costs;
    int32_t  best_rf_idx;
    int16_t  mv_row, mv_col;
    uint64_t ref_frame_poc;
    int64_t  src_srcrf_dist, src_srcrf_rate;
    /* every svt_nxm_sad_kernel_sub_sampled / svt_aom_satd result of the block in call order: the intra modes, then the
     * ME candidates that passed the gates (:636-930) */
    int32_t  costs[REF_TPL_MAX_COSTS];
    /* TplStats as result_model_store left them (:211-288) */
    int64_t  srcrf_dist, recrf_dist, srcrf_rate, recrf_rate;
    int16_t  stats_mv_row, stats_mv_col;
    uint64_t stats_ref_frame_poc;
} ref_tpl_blk_dump;
#pragma pack(pop)
#endif
