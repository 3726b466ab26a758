/*
 * svtme.h - C ABI of the MI355X open-loop motion-estimation engine.
 *
 * The engine replaces, at picture granularity, the body of the 64x64-block loop of SVT-AV1's
 * motion_estimation_kernel() (Source/Lib/Encoder/Codec/EbMotionEstimationProcess.c:750-846): for
 * every 64x64 block (b64) and every searched reference it runs init_me_hme_data, hme_b64,
 * hme_prune_ref_and_adjust_sr and integer_search_b64 (EbMotionEstimation.c:2868, 2337, 2373, 1286)
 * on the GPU and hands back exactly the MeContext fields the unchanged host tail (me_prune_ref,
 * construct_me_candidate_array, compute_distortion, perform_gm_detection) reads.
 *
 * Everything in the signatures is plain C: pointers, sizes and POD structs. No HIP or torch types.
 * All entry points are thread-safe. Return value: 0 on success, a negative svtme_status otherwise.
 */
#ifndef SVTME_H
#define SVTME_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SVTME_MAX_LISTS 2 /* MAX_NUM_OF_REF_PIC_LIST, Source/Lib/Common/Codec/EbDefinitions.h:2431 */
#define SVTME_MAX_REFS 4 /* MAX_REF_IDX / REF_LIST_MAX_DEPTH, EbDefinitions.h:2448, Source/API/EbSvtAv1Enc.h:27 */
#define SVTME_PU_COUNT 85 /* SQUARE_PU_COUNT, Source/Lib/Encoder/Codec/EbMotionEstimationLcuResults.h:25 */
#define SVTME_MAX_SAD_VALUE (128 * 128 * 255) /* MAX_SAD_VALUE, EbMotionEstimation.h:93 */

typedef enum svtme_status {
    SVTME_OK               = 0,
    SVTME_ERR_BAD_ARG      = -1, /* maps to EB_ErrorBadParameter */
    SVTME_ERR_NO_RESOURCES = -2, /* maps to EB_ErrorInsufficientResources */
    SVTME_ERR_NOT_FOUND    = -3, /* unknown store key / ticket; maps to EB_ErrorUndefined */
    SVTME_ERR_DEVICE       = -4, /* a HIP call failed; maps to EB_ErrorUndefined */
    SVTME_ERR_UNSUPPORTED  = -5  /* a control combination the engine does not implement */
} svtme_status;

/* SearchArea / SearchAreaMinMax, EbMotionEstimationContext.h:304-311 */
typedef struct svtme_sa {
    uint16_t width;
    uint16_t height;
} svtme_sa;
typedef struct svtme_sa_minmax {
    svtme_sa sa_min;
    svtme_sa sa_max;
} svtme_sa_minmax;

/*
 * The per-picture ME controls: the fields of MeContext that signal_derivation_me_kernel_oq()
 * (EbMotionEstimationProcess.c:430-538) and its setters write, flattened. Field names follow the
 * reference (MeContext, PreHmeCtrls, MeHmeRefPruneCtrls, MeSrCtrls;
 * EbMotionEstimationContext.h:279-323, 379-403, 464-472).
 */
typedef struct svtme_controls {
    svtme_sa_minmax me_sa; /* integer search area */
    svtme_sa_minmax hme_l0_sa; /* total HME level-0 area */
    svtme_sa        hme_l1_sa;
    svtme_sa        hme_l2_sa;
    uint16_t        num_hme_sa_w;
    uint16_t        num_hme_sa_h;
    uint8_t         enable_hme_flag;
    uint8_t         enable_hme_level0_flag;
    uint8_t         enable_hme_level1_flag;
    uint8_t         enable_hme_level2_flag;
    uint8_t         hme_search_method; /* 0 = SUB_SAD_SEARCH, 1 = FULL_SAD_SEARCH (EbDefinitions.h:2044) */
    uint8_t         me_search_method;
    /* PreHmeCtrls */
    uint8_t         prehme_enable;
    uint8_t         prehme_skip_search_line;
    uint8_t         prehme_l1_early_exit;
    uint8_t         reserved0;
    svtme_sa_minmax prehme_sa_cfg[2];
    /* MeHmeRefPruneCtrls */
    uint8_t  enable_me_hme_ref_pruning;
    uint8_t  protect_closest_refs;
    uint16_t prune_ref_if_hme_sad_dev_bigger_than_th;
    uint16_t prune_ref_if_me_sad_dev_bigger_than_th;
    /* MeSrCtrls */
    uint8_t  enable_me_sr_adjustment;
    uint8_t  distance_based_hme_resizing;
    uint16_t reduce_me_sr_based_on_mv_length_th;
    uint16_t stationary_hme_sad_abs_th;
    uint16_t stationary_me_sr_divisor;
    uint16_t reduce_me_sr_based_on_hme_sad_abs_th;
    uint16_t me_sr_divisor_for_low_hme_sad;
    /* misc MeContext controls */
    uint8_t  reduce_hme_l0_sr_th_min;
    uint8_t  reduce_hme_l0_sr_th_max;
    uint8_t  stat_factor; /* host tail only */
    uint8_t  use_best_unipred_cand_only; /* host tail only */
    int32_t  prune_me_candidates_th; /* host tail only */
    uint32_t me_early_exit_th;
} svtme_controls;

/* Inputs of signal_derivation_me_kernel_oq(): what the preset tables key on. */
typedef struct svtme_preset_key {
    int8_t   enc_mode; /* EbEncMode: ENC_MRS (-2) .. ENC_M13 (13) */
    uint8_t  sc_class1; /* screen-content class */
    uint8_t  input_resolution; /* EbInputResolution 0..6; see svtme_input_resolution() */
    uint8_t  reserved;
    uint32_t frame_rate; /* 16.16 fixed point, scs->static_config.frame_rate */
} svtme_preset_key;

/*
 * Restates signal_derivation_me_kernel_oq() (EbMotionEstimationProcess.c:430-538) with
 * set_me_search_params :154, set_hme_search_params :68, set_prehme_ctrls :385,
 * set_me_hme_ref_prune_ctrls :241, set_me_sr_adjustment_ctrls :290, and the HME level flags of
 * signal_derivation_multi_processes_oq() (EbPictureDecisionProcess.c:1797-1814).
 */
int svtme_derive_controls(const svtme_preset_key *key, svtme_controls *out);

/*
 * Controls of the temporal filter's motion search: tf_signal_derivation_me_kernel_oq() with tf_set_me_hme_params_oq()
 * (EbMotionEstimationProcess.c:554-629) for pcs->tf_ctrls.hme_me_level 0..2, the HME level flags
 * signal_derivation_multi_processes_oq() sets for that level (EbPictureDecisionProcess.c:1816-1841) and the
 * FULL_SAD_SEARCH HME method create_me_context_and_picture_control() forces per block (EbTemporalFiltering.c:331).
 */
int svtme_derive_tf_controls(uint8_t hme_me_level, svtme_controls *out);

/*
 * Controls of the first pass of a two-pass encode: first_pass_signal_derivation_me_kernel() with
 * set_first_pass_me_hme_params_oq() (firstpass.c:627-716) and the HME level flags of
 * first_pass_signal_derivation_multi_processes() (:488-491). reduce_me_search = scs->ipp_pass_ctrls.reduce_me_search.
 */
int svtme_derive_first_pass_controls(uint8_t input_resolution, uint32_t frame_rate, uint8_t reduce_me_search,
                                     svtme_controls *out);

/* derive_input_resolution(), Source/Lib/Encoder/Codec/EbSequenceControlSet.c:396-415. */
uint8_t svtme_input_resolution(uint32_t width, uint32_t height);

typedef struct svtme_engine svtme_engine;

typedef struct svtme_config {
    int32_t  device; /* HIP device ordinal */
    uint32_t max_width; /* luma; multiples of 8 */
    uint32_t max_height;
    uint32_t store_slots; /* capacity of the device picture store (>= references + pictures in flight) */
    uint32_t max_in_flight; /* pictures submitted but not yet waited for */
    uint32_t search_streams; /* 0 = default (3), at most 3: pictures take turns on this many HIP streams, so that one
                                picture's kernel tails overlap the others' kernels; 1 runs pictures back to back */
    uint32_t kernel_timing; /* 1: record an event after every kernel so that svtme_picture_timing() can report
                               per-kernel durations (a few extra runtime calls per picture); 0: totals only */
    uint32_t serial_hme; /* diagnostic, 0 = off: run pre-HME / HME with the serial kernel (one workgroup per 64x64 block
                            walks the references in order, as the reference's loops do) instead of the kernels of
                            independent work items; same results, slower; kept as a cross-check */
} svtme_config;

/* Replaces nothing in the reference (the reference has no device): engine life cycle. Every call of this library makes
 * the engine's device current while it runs and leaves the calling thread's current HIP device as it found it. Calls
 * on one engine may come from several threads (an internal mutex orders them). */
int  svtme_create(const svtme_config *cfg, svtme_engine **out);
void svtme_destroy(svtme_engine *e);

/*
 * Device picture store = mirror of EbPaReferenceObject (Source/Lib/Encoder/Codec/EbReferenceObject.h:69-82):
 * full, quarter and sixteenth resolution 8-bit luma with replicated borders (pads 72 / 32 / 16,
 * Source/Lib/Encoder/Globals/EbEncHandle.c:1331-1355, 3447-3460).
 * svtme_store_put uploads the un-padded 8-bit luma (`stride` bytes per row) and builds all three padded
 * planes on the device: generate_padding (Source/Lib/Common/Codec/EbMcp.c:112-164) and
 * downsample_filtering_input_picture (EbPictureAnalysisProcess.c:2688-2752). `key` identifies the slot
 * (the integration uses the EbPaReferenceObject pointer); putting an existing key overwrites it (the
 * temporal filter rewrites planes, EbTemporalFiltering.c:5169-5217). The host buffer may be reused as
 * soon as the call returns. A picture that reads the slot can be submitted once the put has returned; puts of
 * different keys may run concurrently from several threads (the frame copy is outside the engine's lock).
 */
int svtme_store_put(svtme_engine *e, uint64_t key, uint64_t picture_number, const uint8_t *luma,
                    uint32_t stride, uint32_t width, uint32_t height);
/*
 * The same for the n source pictures that become final together - the pictures of a mini-GOP after their temporal
 * filtering, right before send_picture_out posts them (EbPictureDecisionProcess.c:6815-6827): one pass over the store's
 * bookkeeping, n uploads queued back to back and ONE launch of each store-building kernel for all of them, instead of
 * two launches per picture. All pictures have the size width x height and rows `stride` apart; keys[i] / lumas[i] /
 * picture_numbers[i] belong to picture i; n <= SVTME_MAX_PUT_BATCH. pinned != 0: the rows are fetched by DMA from the
 * callers' page-locked buffers as they lie (the rules of svtme_store_put_pinned apply to every buffer), else they are
 * packed into the engine's staging buffers first and the callers' buffers are free on return. Either all pictures are
 * put or none (an error leaves the store as it was, except that keys that existed before keep their slots).
 */
#define SVTME_MAX_PUT_BATCH 16
int svtme_store_put_many(svtme_engine *e, uint32_t n, const uint64_t *keys, const uint64_t *picture_numbers,
                         const uint8_t *const *lumas, uint32_t stride, uint32_t width, uint32_t height, int pinned);
/*
 * Copies the three built planes of `key` from another engine's store (another GPU of the node) into this one's, device
 * to device - over xGMI where the GPUs are peers - instead of a second upload over PCIe and a second run of the
 * store-building kernels. No counterpart in the reference (it has one address space). Both engines must have been
 * created with the same max_width / max_height. Synchronous: returns when the planes have arrived, so that the source
 * slot may be rewritten or released right after; SVTME_ERR_NOT_FOUND if `src` does not hold the key.
 */
int svtme_store_copy_peer(svtme_engine *dst, svtme_engine *src, uint64_t key);
/*
 * The same put without the host-side frame copy, for luma that already lies in page-locked memory: the rows are fetched
 * by DMA from the caller's buffer as they lie, `stride` apart (the encoder's y8b input buffers, the pool created at
 * Source/Lib/Encoder/Globals/EbEncHandle.c:1778-1787 by svt_input_y8b_creator, which the PA reference object's full
 * plane aliases, EbResourceCoordinationProcess.c:1294: allocate the pool with svtme_host_alloc or page-lock it once with
 * svtme_host_register). The call returns
 * at once; the buffer must stay unchanged until svtme_store_wait(key) has returned or a picture that reads the slot has
 * been waited for. That is the encoder's own rule for these buffers: they live as long as the PA reference object and are
 * only rewritten by the temporal filter, before the picture is put again. Memory that is not page-locked still works
 * (the runtime then stages the copy itself, slower); a stride that is not a multiple of 4 or exceeds width + 256 is
 * handled like svtme_store_put.
 */
int svtme_store_put_pinned(svtme_engine *e, uint64_t key, uint64_t picture_number, const uint8_t *luma,
                           uint32_t stride, uint32_t width, uint32_t height);
/* Blocks until the planes of `key` are built on the device (its last put has run). */
int svtme_store_wait(svtme_engine *e, uint64_t key);
/* Page-locked host memory for svtme_store_put_pinned behind plain-C signatures, so that the encoder-side shim needs no HIP
 * headers. Either the shim allocates the luma buffers of the y8b pool with svtme_host_alloc (in svt_input_y8b_creator, in
 * place of its EB_MALLOC of buffer_y), or it page-locks the buffers it has with svtme_host_register. The latter depends on
 * the driver being able to lock user pages and returns SVTME_ERR_UNSUPPORTED where it cannot (some virtual machines);
 * registering a range twice is not an error. */
int svtme_host_alloc(svtme_engine *e, size_t bytes, void **out);
int svtme_host_free(svtme_engine *e, void *ptr);
int svtme_host_register(svtme_engine *e, void *ptr, size_t bytes);
int svtme_host_unregister(svtme_engine *e, void *ptr);
/* Mirrors release_pa_reference_objects (EbReferenceObject.c:298): frees the slot once no submitted
 * picture still reads it. */
int svtme_store_release(svtme_engine *e, uint64_t key);
/* Test/diagnostic read-back of one padded plane (level 0 = sixteenth, 1 = quarter, 2 = full). `dst` must
 * hold stride*(height + 2*pad) bytes; geometry is returned through the out parameters. */
int svtme_store_read(svtme_engine *e, uint64_t key, int level, uint8_t *dst, size_t dst_size,
                     uint32_t *stride, uint32_t *width, uint32_t *height, uint32_t *pad);

/* Optional second stage on the device: the rest of motion_estimation_b64 after integer_search_b64
 * (EbMotionEstimation.c:2975-2999: me_prune_ref :1576-1618, construct_me_candidate_array[_mrp_off] :2430-2685,
 * compute_distortion :2814-2865, perform_gm_detection :2688-2811), so that the host only copies finished
 * MotionEstimationResults and per-b64 statistics. All zero = off (the host runs that code itself). */
typedef struct svtme_tail {
    uint8_t enable; /* 1: tail results in addition to the search arrays; 2: tail results only (p_sb_best_sad / p_sb_best_mv
                       stay on the device and come back NULL), which halves the result copy */
    uint8_t enable_me_8x8; /* pcs->enable_me_8x8 */
    uint8_t enable_me_16x16; /* pcs->enable_me_16x16 */
    uint8_t max_refs; /* pcs->pa_me_data->max_refs, <= SVTME_TAIL_MV_STRIDE */
    uint8_t max_cand; /* pcs->pa_me_data->max_cand, <= SVTME_TAIL_CAND_STRIDE */
    uint8_t max_l0; /* pcs->pa_me_data->max_l0 */
    uint8_t input_resolution; /* scs->input_resolution */
    uint8_t gm_enabled; /* pcs->gm_ctrls.enabled */
    uint8_t gm_use_distance_based_active_th; /* pcs->gm_ctrls.use_distance_based_active_th */
    uint8_t reserved[7];
} svtme_tail;
#define SVTME_TAIL_MV_STRIDE 7 /* MAX_PA_ME_MV, EbMotionEstimationLcuResults.h:23 */
#define SVTME_TAIL_CAND_STRIDE 23 /* MAX_PA_ME_CAND, :24 */

/* MeContext::me_type with the values of EbMeType (EbMotionEstimationContext.h:44-50), so that a shim can pass
 * context->me_type through. 0 (ME_CLOSE_LOOP, which never reaches motion_estimation_b64) is read as SVTME_ME_OPEN_LOOP, so
 * that a zero-initialised svtme_picture is an open-loop job. */
#define SVTME_ME_OPEN_LOOP 3 /* ME_OPEN_LOOP: the pre-analysis search of motion_estimation_kernel() */
#define SVTME_ME_MCTF 1 /* ME_MCTF: the temporal filter's search of one neighbour frame */
#define SVTME_ME_FIRST_PASS 4 /* ME_FIRST_PASS: the first pass of a two-pass encode (firstpass.c:1237-1300) */

/* One picture's ME job: what motion_estimation_kernel() sets on MeContext before the b64 loop
 * (EbMotionEstimationProcess.c:795-837) plus the pcs/scs fields integer_search_b64 reads.
 *
 * me_type = SVTME_ME_MCTF runs the temporal filter's form of motion_estimation_b64 for all blocks of one
 * (centre frame, neighbour frame) pair (EbTemporalFiltering.c:4835-4873 per block): cur_key = the centre frame, one
 * list with one reference = the neighbour, ctrl from svtme_derive_tf_controls(), temporal_layer_index /
 * is_used_as_reference_flag of the centre picture. The differences (EbMotionEstimation.c): the integer search area grows
 * with the plain picture distance (:1338-1341); check_00_center weights a search-centre SAD above 512 by 125 % before
 * comparing it with the zero vector (:1231-1241); no reference pruning or search-range adjustment (:2952); a block whose
 * HME SAD is below tf_me_exit_th is not searched at all (:2958-2962): its search_results[0][0].tf_me_exit is 1 (the
 * caller sets tf_use_pred_64x64_only_th = ~0) and its minima read as zero; tail must be off (:2977). The
 * tf_tot_horz_blks / tf_tot_vert_blks counters (:2365-2370) follow from search_results[0][0].hme_sc_x / hme_sc_y.
 *
 * me_type = SVTME_ME_FIRST_PASS is first_pass_me()'s call (firstpass.c:1237-1300): one list of up to two references (the
 * previous and the golden frame, :1177-1186), temporal layer 0, used as reference, ctrl from
 * svtme_derive_first_pass_controls(). Differences: check_00_center is skipped when bypass_zz_check
 * (scs->ipp_pass_ctrls.bypass_zz_check, EbMotionEstimation.c:1363-1364) and the host tail has no compute_distortion
 * (:2990; a device tail leaves those statistics zero). Skipped frames (MeContext::skip_frame) are the caller's business: it
 * does not submit them; skipped blocks (bypass_blk_step): b64_step. */
typedef struct svtme_picture {
    uint64_t cur_key; /* store key of the current picture */
    uint64_t picture_number; /* pcs->picture_number */
    uint32_t width; /* input_ptr->width  (== pcs->aligned_width here) */
    uint32_t height; /* input_ptr->height (== pcs->aligned_height here) */
    uint8_t  temporal_layer_index;
    uint8_t  is_used_as_reference_flag;
    uint8_t  num_of_list_to_search; /* 1 for P, 2 for B */
    uint8_t  num_of_ref_pic_to_search[SVTME_MAX_LISTS]; /* ref_list{0,1}_count_try */
    uint8_t  me_type; /* SVTME_ME_OPEN_LOOP (or 0), SVTME_ME_MCTF or SVTME_ME_FIRST_PASS */
    uint8_t  bypass_zz_check; /* SVTME_ME_FIRST_PASS only */
    uint8_t  restricted_motion_vector; /* scs->static_config.restricted_motion_vector: the integer search window stays inside
                                          the block's tile instead of the padded picture (EbMotionEstimation.c:1405-1437,
                                          :1467-1498). The open-loop ME always sees one tile = the picture
                                          (EbPictureControlSet.c:1684-1688); temporal-filter and first-pass searches take
                                          the tiles of svtme_set_tile_rects() (:1408-1410), the picture when none are set */
    uint64_t ref_key[SVTME_MAX_LISTS][SVTME_MAX_REFS]; /* store keys of pcs->ref_pa_pic_ptr_array */
    uint64_t ref_picture_number[SVTME_MAX_LISTS][SVTME_MAX_REFS];
    svtme_controls ctrl;
    svtme_tail     tail;
    uint32_t       tf_me_exit_th; /* MeContext::tf_me_exit_th = pcs->tf_ctrls.me_exit_th; SVTME_ME_MCTF only */
    uint8_t        b64_step; /* MeContext::bypass_blk_step (firstpass.c:1282-1285, :1313-1317): 0 or 1 = every 64x64 block; n > 1
                                (the first pass uses 2 from 360p on when ipp_pass_ctrls.bypass_blk_step is set) = only the blocks
                                whose column and row index are multiples of n are searched, nothing is computed for the others
                                and their results are undefined (first_pass_me does not call motion_estimation_b64 for them) */
    uint8_t        reserved[3];
} svtme_picture;

/* The tile of one 64x64 block: SbParams::tile_start_x, tile_start_y, tile_end_x, tile_end_y in the reference's order
 * (EbPictureControlSet.h:483-486), in luma pixels, end exclusive. */
typedef struct svtme_tile_rect {
    uint32_t start_x, start_y, end_x, end_y;
} svtme_tile_rect;

/* Hands the engine scs->sb_params_array's tile fields as EbResourceCoordinationProcess.c:1304-1349 fills them for a
 * sequence with restricted_motion_vector: rects[n_b64] in raster order for pictures of width x height. From then on the
 * integer search of SVTME_ME_MCTF and SVTME_ME_FIRST_PASS pictures of that size submitted with restricted_motion_vector
 * keeps each block's window inside its tile, as integer_search_b64 does with scs->sb_params_array
 * (EbMotionEstimation.c:1405-1498); SVTME_ME_OPEN_LOOP pictures keep the picture as their tile, as the reference does with
 * pcs->sb_params_array. A sequence-level call: it waits for the searches in flight, then replaces the table;
 * rects == NULL clears it. Every bound must lie inside 0..width / 0..height (SVTME_ERR_BAD_ARG otherwise); a restricted
 * temporal-filter / first-pass picture of another size than the table's is refused with SVTME_ERR_BAD_ARG. */
int svtme_set_tile_rects(svtme_engine *e, uint32_t width, uint32_t height, const svtme_tile_rect *rects, uint32_t n_b64);

/* SearchResults, EbMotionEstimationContext.h:330-337, as left by hme_prune_ref_and_adjust_sr: the same six fields at the
 * same offsets (0, 1, 2, 4, 8, 16) in the same 24 bytes on the LP64 / Win64 ABIs the encoder builds for, so an array of
 * these can be copied byte for byte into MeContext::search_results[2][4] (:399). The engine's one extra field lives in
 * the reference struct's tail padding, which nothing in the encoder reads. */
typedef struct svtme_search_result {
    uint8_t  list_i;
    uint8_t  ref_i;
    int16_t  hme_sc_x;
    int16_t  hme_sc_y;
    uint64_t hme_sad;
    uint8_t  do_ref;
    uint8_t  tf_me_exit; /* SVTME_ME_MCTF, entry [0][0] only: 1 = the block left at the tf_me_exit_th test */
    uint8_t  reserved[6];
} svtme_search_result;

/* Per-b64 outputs. Pointers are engine-owned pinned host memory, valid until svtme_picture_done().
 * n_b64 = ceil(w/64)*ceil(h/64), raster order. The small arrays are in the layout of the MeContext arrays they are copied
 * into; the two big ones hold one plane of [n_b64][85] entries per searched reference (list-0 references first, then
 * list 1: plane = r for list 0, n_ref[0] + r for list 1), so that only references the picture has cross PCIe.
 * svtme_unpack_b64() turns a block's entries back into MeContext's [2][4][85] arrays. */
typedef struct svtme_results {
    uint32_t                   n_b64;
    const uint32_t            *p_sb_best_sad; /* [n_ref[0] + n_ref[1]][n_b64][85], MeContext::p_sb_best_sad, EbMotionEstimationContext.h:367 */
    const uint32_t            *p_sb_best_mv; /* same shape, (y_qpel << 16) | (x_qpel & 0xffff), :368 */
    const svtme_search_result *search_results; /* [n_b64][2][4] */
    const uint32_t            *zz_sad; /* [n_b64][2][4], MeContext::zz_sad :471 */
    const uint32_t            *reduce_me_sr_divisor; /* [n_b64][2][4], :400 */
    /* Only for pictures submitted with tail.enable (else NULL): MotionEstimationResults of every b64
     * (EbMotionEstimationLcuResults.h:37-45; arrays indexed as the reference does, pu * max_refs + ... and
     * pu * max_cand + ..., unused entries zero), MeContext::me_distortion, the pcs statistics of compute_distortion /
     * perform_gm_detection and the final do_ref / hme_sad of me_prune_ref. */
    const uint8_t  *total_me_candidate_index; /* [n_b64][85] */
    const uint32_t *me_mv_array; /* [n_b64][85 * SVTME_TAIL_MV_STRIDE] */
    const uint8_t  *me_candidate_array; /* [n_b64][85 * SVTME_TAIL_CAND_STRIDE], MeCandidate bit-fields */
    const uint32_t *me_distortion; /* [n_b64][85] */
    const uint32_t *b64_stats; /* [n_b64][8]: rc_me_distortion, me_64x64 / 32x32 / 16x16 / 8x8_distortion,
                                  me_8x8_cost_variance, stationary_block_present_sb, rc_me_allow_gm */
    const uint8_t  *final_do_ref; /* [n_b64][2][4] */
    const uint64_t *final_hme_sad; /* [n_b64][2][4] */
    uint32_t        n_ref[SVTME_MAX_LISTS]; /* references per list of the picture = planes of the two big arrays */
} svtme_results;

typedef uint64_t svtme_ticket;

/* Queues the whole picture (all b64 x all references) on the engine's stream and returns at once.
 * Replaces the per-b64 calls motion_estimation_b64 -> init_me_hme_data / hme_b64 /
 * hme_prune_ref_and_adjust_sr / integer_search_b64 (EbMotionEstimation.c:2954-2970). */
int svtme_picture_submit(svtme_engine *e, const svtme_picture *pic, svtme_ticket *ticket);
/* The same for several pictures at once, which then share their kernel launches: the per-launch costs (launch latency,
 * the under-filled tail of every kernel) are paid once per batch instead of once per picture, which is what bounds the
 * fast presets. The natural batch is what send_picture_out() posts back to back for one mini-GOP
 * (EbPictureDecisionProcess.c:6842-6874). The pictures must have one size, one set of controls, one me_type and one
 * tail setting (SVTME_ERR_BAD_ARG otherwise; submit such pictures in batches of their own); they may differ in everything
 * else. tickets[i] belongs to pics[i]; each picture is waited for and handed back on its own, and the results of all of
 * them arrive together. Needs n free pictures in flight. */
#define SVTME_MAX_BATCH 32
int svtme_pictures_submit(svtme_engine *e, const svtme_picture *pics, uint32_t n, svtme_ticket *tickets);
/* Blocks until the picture's results are in host memory. */
int svtme_picture_wait(svtme_engine *e, svtme_ticket ticket, svtme_results *out);
/* Returns the result buffers of a waited picture to the engine. */
int svtme_picture_done(svtme_engine *e, svtme_ticket ticket);

/*
 * svtme_picture_submit + svtme_picture_wait in one synchronous call that allocates nothing. It has no counterpart in the
 * reference; it is for callers that cannot allocate device memory while they run: code under HIP graph capture, tight
 * serving loops, harnesses that forbid hipMalloc. Same kernels and same results as svtme_picture_submit, but
 *  - the kernels' intermediates (HME records, search windows, running minima, counters) live in a file-scope device array
 *    of the library, 13.1 MiB, which every GPU the library runs on holds until the process ends, instead of the buffers
 *    svtme_create allocated per picture in flight;
 *  - the results go into `result_block`, memory the caller owns and the device can write: device memory or pinned host
 *    memory, 64-byte aligned, at least SVTME_STATIC_RESULT_BYTES(n_b64, references of the picture) bytes; `out` receives
 *    pointers into it (the five arrays of a picture without device tail, laid out as svtme_picture_wait returns them);
 *  - the picture's source planes are store slots as usual (svtme_store_put before the call; it allocates nothing either).
 * Limits: at most SVTME_STATIC_MAX_B64 64x64 blocks (3840x2160; larger pictures get SVTME_ERR_BAD_ARG and a message on
 * stderr), tail.enable = 0 and serial_hme = 0 (SVTME_ERR_UNSUPPORTED otherwise). The call returns when the results are in
 * `result_block`. All calls on one GPU share that GPU's array: within a process they take turns (a lock), calls from
 * different processes each have their own copy; the call clears what has to start from zero, whatever the last call left.
 * No hipMalloc / hipFree / hipHostMalloc, stream or event creation happens inside. The call holds the engine's lock from
 * its first launch to the end of its last kernel: submits, puts and waits of other threads on the same engine stall for
 * that long, and it shares its stream with the svtme_gm_* / svtme_tpl_search calls (they take turns). A result_block that
 * is neither device memory nor page-locked host memory is refused with SVTME_ERR_BAD_ARG.
 */
#define SVTME_STATIC_MAX_B64 2040
#define SVTME_STATIC_RESULT_BYTES(n_b64, n_refs) \
    ((size_t)(n_b64) * 256 + 4096 + (size_t)(n_refs) * (size_t)(n_b64) * SVTME_PU_COUNT * 8)
int svtme_picture_search_static(svtme_engine *e, const svtme_picture *pic, void *result_block, size_t result_bytes,
                                svtme_results *out);

/* Device time of a waited picture's kernels in milliseconds (HIP events on the stream the kernels run on):
 * [0] HME pre-HME + level-0 kernel (or the serial HME kernel), [1] HME level-1/2 kernel, [2] integer-search plan,
 * [3] integer-search kernel, [4] integer-search finish, [5] first kernel start to last kernel end,
 * [6] the same up to the end of the result copy, [7] the number of pictures in the picture's batch. [0]..[4] are 0 unless
 * the engine was created with kernel_timing = 1. The pictures of a batch share their launches: each reports an equal
 * share (1 / [7]) of the batch's durations. Diagnostic only. */
int svtme_picture_timing(svtme_engine *e, svtme_ticket ticket, float ms[8]);

/* Switches the per-kernel events of the submissions that follow on (1) or off (0); svtme_config.kernel_timing is the
 * state an engine starts in. The events cost a few microseconds of device time per submission, which a launch-bound
 * workload notices (a tenth of the rate at 720p preset 12): a caller that samples kernel durations switches them on for
 * the submissions it will ask svtme_picture_timing about. No counterpart in the reference: a measurement aid. */
int svtme_kernel_timing(svtme_engine *e, int on);

/* Number of abs-diff-accumulate operations the waited picture's searches performed (after clipping, early exits
 * and pruning): [0] zero-MV / search-centre SADs, [1] pre-HME, [2] HME L0, [3] L1, [4] L2, [5] integer search.
 * Diagnostic: the numerator of the roofline figures. */
int svtme_picture_ops(svtme_engine *e, svtme_ticket ticket, uint64_t ops[6]);

/* Copies one b64's results into caller arrays shaped like MeContext's
 * (p_sb_best_sad[2][4][85], p_sb_best_mv[2][4][85], search_results[2][4], zz_sad[2][4],
 * reduce_me_sr_divisor[2][4]); this is what the shim in motion_estimation_kernel() calls per b64
 * instead of hme_b64 + integer_search_b64. */
int svtme_unpack_b64(const svtme_results *res, uint32_t b64_index, uint32_t *p_sb_best_sad,
                     uint32_t *p_sb_best_mv, svtme_search_result *search_results, uint32_t *zz_sad,
                     uint32_t *reduce_me_sr_divisor);

/*
 * Several engines, one per GPU of a node, behind one handle: the picture-sharding scheduler. It has no counterpart in the
 * reference (which has one address space); it sits where send_picture_out() posts the pictures of a mini-GOP back to back
 * (EbPictureDecisionProcess.c:5031-5125, :6842-6874). Open-loop ME reads source pictures only, so those pictures are
 * independent: svtme_group_pictures_submit takes a mini-GOP's pictures (the constraints of svtme_pictures_submit apply),
 * gives each to the engine where it ends earliest - the engines' accumulated estimated work plus the picture's own and
 * what it costs to bring the sources it lacks there - and submits every engine's share as one batch;
 * svtme_group_picture_submit is the same for one picture. svtme_group_store_put only announces a source (the
 * buffer must stay valid and unchanged until svtme_group_store_release: true of the encoder's y8b buffers, which live as
 * long as their PA reference object; `pinned` != 0 says it is page-locked, see svtme_store_put_pinned); the group brings
 * it to an engine the first time a picture that runs there reads it, so every source reaches exactly the GPUs that need
 * it: as a device-to-device copy of the built planes from an engine that holds them already (svtme_store_copy_peer: two
 * engines on one GPU, or two GPUs that are peers - over xGMI -; svtme_group_set_peer_copies(g, 0) turns that off), else as
 * an upload from the host, the sources an engine lacks for a batch in one svtme_store_put_many. Announcing an existing key
 * again (the temporal filter rewrote the planes) makes every copy stale; the release frees the key's slot on every engine
 * that ever took a copy. Results come from each GPU to its own pinned host buffers: no exchange between GPUs inside a
 * picture, no collective. `cfg` applies to every engine (its `device` is replaced by devices[i]; the same device may
 * appear more than once; at most 56 engines). Tickets of a group are only valid with the group's wait / done;
 * svtme_group_ticket_engine says which engine (index into devices[]) runs the picture. If a submit fails part of the way
 * (an engine out of resources), the tickets that are not 0 on return were submitted and are waited for and handed back
 * as usual. svtme_group_stats: pictures submitted, sources announced, uploads made ((uploads + peer copies) / sources = how
 * many GPUs a source travels to on average); svtme_group_peer_stats: device-to-device copies made, and how many
 * (engine, engine) directions can copy that way.
 */
typedef struct svtme_group svtme_group;
int      svtme_group_create(const int32_t *devices, uint32_t n_devices, const svtme_config *cfg, svtme_group **out);
void     svtme_group_destroy(svtme_group *g);
uint32_t svtme_group_size(const svtme_group *g);
int svtme_group_store_put(svtme_group *g, uint64_t key, uint64_t picture_number, const uint8_t *luma, uint32_t stride,
                          uint32_t width, uint32_t height, int pinned);
int svtme_group_store_release(svtme_group *g, uint64_t key);
int svtme_group_picture_submit(svtme_group *g, const svtme_picture *pic, svtme_ticket *ticket);
int svtme_group_pictures_submit(svtme_group *g, const svtme_picture *pics, uint32_t n, svtme_ticket *tickets);
uint32_t svtme_group_ticket_engine(svtme_ticket ticket);
int svtme_group_set_peer_copies(svtme_group *g, int enable);
int svtme_group_peer_stats(svtme_group *g, uint64_t *peer_copies, uint64_t *peer_links);
int svtme_group_picture_wait(svtme_group *g, svtme_ticket ticket, svtme_results *out);
int svtme_group_picture_done(svtme_group *g, svtme_ticket ticket);
int svtme_group_stats(svtme_group *g, uint64_t *pictures, uint64_t *sources, uint64_t *uploads);
/* For the calls that take an engine and store keys (svtme_tpl_search, svtme_gm_*) from a caller that holds a group: the
 * engine of the group on which all of keys[0 .. n_keys) are resident when the call returns - the one that held most of
 * them already (the engine that ran the picture's ME holds the picture and its references), the missing ones are uploaded
 * to it. The handle stays valid as long as the group; it must not be destroyed. No counterpart in the reference.
 * svtme_group_gm_set_tables: svtme_gm_set_tables on every engine of the group. */
int svtme_group_engine_for(svtme_group *g, const uint64_t *keys, uint32_t n_keys, svtme_engine **engine);
int svtme_group_gm_set_tables(svtme_group *g, const int16_t *warped_filter, const int32_t *error_measure_lut);
/* svtme_set_tile_rects on every engine of the group */
int svtme_group_set_tile_rects(svtme_group *g, uint32_t width, uint32_t height, const svtme_tile_rect *rects, uint32_t n_b64);

/* ---- Global-motion estimation: the data-parallel stages of compute_global_motion
 * (Source/Lib/Encoder/Codec/EbGlobalMotionEstimation.c:155-301) on the planes the store already holds. The encoder keeps
 * its own RANSAC (ransac.c, double precision, a few hundred points), the final shear derivation and the parameter costs; it
 * calls the engine where it called these functions (the refinement loop either as a whole, svtme_gm_refine, or - keeping
 * its own loop - error by error, svtme_gm_warp_error):
 *
 *   svtme_gm_corners          <- svt_av1_fast_corner_detect (corner_detect.c:19-32): FAST-9, barrier 18, non-maximum
 *                                suppression, the first max_points corners in raster order as (x, y) pairs
 *   svtme_gm_correspondences  <- svt_av1_fast_corner_detect of the reference picture + svt_av1_determine_correspondence
 *                                (corner_match.c:153-214; global_motion.c:294-309): corners of both pictures (at most
 *                                SVTME_GM_MAX_CORNERS each), best match by normalised cross-correlation of 13x13 patches
 *                                within max(w, h) / 16 pixels, then the two 9x9 refinements; (x, y, rx, ry) per match
 *   svtme_gm_warp_error       <- svt_av1_warp_error (EbEncWarpedMotion.c:229-267) for 8 bit after svt_get_shear_params
 *                                succeeded (the caller passes the shear it derived; when that function fails the
 *                                reference returns 1 without looking at the pictures, and so does the caller): the
 *                                reference picture warped in 32x32 blocks and compared with the current picture through
 *                                the error table; as in the reference the sum stops at the first block, in raster order,
 *                                that takes it above best_error
 *   svtme_gm_frame_error      <- svt_av1_frame_error (EbEncWarpedMotion.c:215-227) for 8 bit: the same measure, no warp
 *
 *   svtme_gm_refine           <- svt_av1_refine_integerized_param (global_motion.c:135-263) for 8 bit, with the
 *                                svt_av1_warp_error calls it makes (svt_get_shear_params, EbWarpedMotion.c:1115-1144,
 *                                included): the whole loop - every refinement step of every parameter, each an error over
 *                                the picture - as one chain of launches that the device advances by itself; the host
 *                                looks at it once it has ended. wm: the model as svt_av1_convert_model_to_params left it
 *                                (wmmat[0..5] and, for completeness, the shear the struct holds); on return the refined
 *                                model as the reference's function leaves it (wmmat[6] and [7], which this struct does not
 *                                carry, are 0 then), its type (get_wmtype) and the function's return value in *error.
 *                                *evaluations (may be NULL): how many errors the loop took. Replaces the encoder's loop
 *                                around svtme_gm_warp_error; RANSAC and the parameter costs stay with the encoder
 *
 * downsample_level is pcs->gm_ctrls.downsample_level (GM_FULL 0, GM_DOWN 1, GM_DOWN16 2, EbDefinitions.h:155-159) and
 * selects the full, quarter or sixteenth plane of both pictures (EbGlobalMotionEstimation.c:109-122). The calls are
 * synchronous and take turns on one engine; the keys must stay in the store and must not be put again while a call that
 * reads them runs.
 *
 * svtme_gm_set_tables: the warp filter (eb_warped_filter[193][8], EbWarpedMotion.h:44) and the error table
 * (error_measure_lut[512], EbWarpedMotion.h:45) of the AV1 specification are not part of this library; the encoder hands
 * over its own once per engine, before the first error call (SVTME_ERR_BAD_ARG until then).
 */
#define SVTME_GM_MAX_CORNERS 4096 /* MAX_CORNERS, global_motion.h:35 */
typedef struct svtme_gm_warp { /* the fields of EbWarpedMotionParams (EbDefinitions.h:1969-1975) the warp reads */
    int32_t wmmat[6];
    int16_t alpha, beta, gamma, delta;
} svtme_gm_warp;
int svtme_gm_set_tables(svtme_engine *e, const int16_t *warped_filter, const int32_t *error_measure_lut);
/* max_points: at most SVTME_GM_MAX_CORNERS (SVTME_ERR_BAD_ARG beyond it); the first max_points corners in raster order */
int svtme_gm_corners(svtme_engine *e, uint64_t key, int downsample_level, int32_t *points, int32_t max_points, int32_t *num_points);
int svtme_gm_correspondences(svtme_engine *e, uint64_t cur_key, uint64_t ref_key, int downsample_level,
                             int32_t *correspondences /* [4 * SVTME_GM_MAX_CORNERS] */, int32_t *num_correspondences);
int svtme_gm_warp_error(svtme_engine *e, uint64_t cur_key, uint64_t ref_key, int downsample_level, const svtme_gm_warp *wm,
                        int64_t best_error, int64_t *error);
int svtme_gm_frame_error(svtme_engine *e, uint64_t cur_key, uint64_t ref_key, int downsample_level, int64_t *error);
typedef struct svtme_gm_model { /* EbWarpedMotionParams (EbDefinitions.h:1969-1975) without wmmat[6], [7] and invalid */
    int32_t wmmat[6];
    int16_t alpha, beta, gamma, delta;
    int32_t wmtype; /* TransformationType: IDENTITY 0, TRANSLATION 1, ROTZOOM 2, AFFINE 3 */
} svtme_gm_model;
/* wmtype: the type to force (the reference passes the model's own, EbGlobalMotionEstimation.c:231-246); n_refinements:
 * gm_ctrls.params_refinement_steps, 1 .. 13 (the first step is 1 << (n_refinements - 1) units of a parameter whose range is
 * 1 << 12; the encoder uses 1 and 5; SVTME_ERR_BAD_ARG outside that, as for a wmtype outside 0 .. 3); best_frame_error:
 * the best error of the models tried before (INT64_MAX for the first) */
int svtme_gm_refine(svtme_engine *e, uint64_t cur_key, uint64_t ref_key, int downsample_level, svtme_gm_model *wm, int wmtype,
                    int n_refinements, int64_t best_frame_error, int64_t *error, uint32_t *evaluations);

/* ---- TPL dispenser: the source-based mode search of tpl_mc_flow_dispenser_sb_generic
 * (Source/Lib/Encoder/Codec/EbSourceBasedOperationsProcess.c:632-933): for every block of the dispenser's grid the cost of
 * the DC intra prediction from the source picture's own neighbours (:651-691) and the cost of every single-reference ME
 * candidate at its full-pel vector, clamped to TPL_PADX / TPL_PADY around the reference (:836-930), on the full-resolution
 * planes the store already holds (tpl_ref_ds_ptr_array[l][r].picture_ptr is the input_padded_picture_ptr of the same PA
 * reference objects ME reads, EbPictureDecisionProcess.c:4985-4988). These searches read source pictures only, so all
 * blocks of a picture are independent; what follows in the dispenser - transform, quantisation and reconstruction of the
 * chosen mode, in raster order because intra prediction then reads reconstructed neighbours (:935-1197) - stays in the
 * encoder, which runs it on the results exactly as it does when tpl_src_data_ready is set (:993-1002).
 *
 * The grid: dispenser_search_level 0 / 1 / 2 = 16x16 / 32x32 / 64x64 blocks (:540-542), 16x16 in every 64x64 block that
 * is not complete (:2086-2088, :2113-2115), a block counts when at least half of it lies inside the picture (:604-607).
 * Results are indexed [b64][i], i = 0 the 64x64 block, 1-4 the 32x32 blocks, 5-20 the 16x16 blocks in the order of
 * tpl_blk_idx_tab (:514-516); blocks outside the grid have valid = 0.
 *
 * Costs, for tpl_opt_flag = 1 (tpl levels 2-7, presets M6 and faster; EbResourceCoordinationProcess.c:203-330,
 * EbEncHandle.c:3098-3125; only DC intra prediction is tried then): the sum of absolute differences where
 * use_pred_sad_in_intra_search / use_pred_sad_in_inter_search say so (:690, :886), else the sum of the absolute
 * coefficients of the residual's forward 16x16 DCT (svt_av1_wht_fwd_txfm with pf_shape + svt_aom_satd, :790-813, :900-916):
 * on the device for dispenser_search_level 0 and subsample_tx 0, which is what the levels that use it have.
 * tpl_opt_flag = 0 (tpl level 1, presets M5 and slower, :186-198: 16x16 blocks, whole 16x16 transforms, DEFAULT_SHAPE):
 * every intra mode from DC_PRED to intra_mode_end is tried on the source picture's own neighbours
 * (update_neighbor_samples_array_open_loop_mb, EbEncIntraPrediction.c:1150-1235; the directional modes after
 * filter_intra_edge, EbIntraPrediction.c:2633-2689), each costed in the transform domain, the first of the cheapest wins
 * (:735-820, best_intra_mode); the inter candidates are costed in the transform domain too, and with src_dist the chosen
 * one also yields rate_estimator's value (:195-209, src_rate). Every combination not named here: SVTME_ERR_UNSUPPORTED.
 *
 * src_dist = 1 adds what :935-982 derive from the chosen inter candidate: its residual's forward transform, quantised with
 * the picture's quantiser (svt_av1_quantize_fp, no quantisation matrices) and compared with the coefficients
 * (get_quantize_error, :167-193); src_recon_error is that function's recon_error, from which the encoder forms
 * srcrf_dist = (recon_error << TPL_DEP_COST_SCALE_LOG2) << subsample_tx. Available for the transforms the tpl levels use
 * (tx_size_array / sub2_tx_size_array / sub4_tx_size_array, :543-545): 16x16, 16x8 and 16x4 on 16x16 blocks
 * (dispenser_search_level 0, subsample_tx 0 / 1 / 2) and 32x8 on 32x32 blocks (level 1 with subsample_tx 2, whose
 * incomplete 64x64 blocks take 16x4); SVTME_ERR_UNSUPPORTED otherwise - the encoder then transforms the chosen candidate
 * itself.
 *
 * The ME results come in the layout of MeSbResults (EbMotionEstimationLcuResults.h:29-52), n_pu entries per 64x64 block
 * (SQUARE_PU_COUNT 85, MAX_SB64_PU_COUNT_NO_8X8 21, or 5 when enable_me_16x16 is 0: EbPictureControlSet.c:112-117), one
 * 64x64 block after the other. A candidate count above max_cand, or a candidate that names a reference beyond n_ref, makes
 * the call fail with SVTME_ERR_BAD_ARG. Synchronous; takes turns with the global-motion calls. */
#define SVTME_TPL_BLOCKS 21
typedef struct svtme_tpl_picture {
    uint64_t cur_key; /* pcs->enhanced_picture_ptr: the picture's own store entry */
    uint64_t ref_key[SVTME_MAX_LISTS][SVTME_MAX_REFS]; /* tpl_data.tpl_ref_ds_ptr_array[l][r].picture_ptr */
    uint8_t  ref_skip[SVTME_MAX_LISTS][SVTME_MAX_REFS]; /* ref_tpl_group_idx > 0 && !base_pcs->tpl_valid_pic[..] (:846-849) */
    uint32_t width, height; /* enhanced_picture_ptr->width / height */
    uint8_t  i_slice; /* pcs->slice_type == I_SLICE: no ME candidates (:832-834) */
    uint8_t  dispenser_search_level; /* tpl_ctrls.dispenser_search_level */
    uint8_t  disable_intra_pred; /* tpl_ctrls.disable_intra_pred_nref && !tpl_data.is_used_as_reference_flag (:592-593) */
    uint8_t  tpl_opt_flag; /* tpl_ctrls.* */
    uint8_t  use_pred_sad_in_intra_search;
    uint8_t  use_pred_sad_in_inter_search;
    uint8_t  enable_me_16x16; /* pcs->enable_me_16x16 (:827-828) */
    uint8_t  n_pu; /* entries per 64x64 block in the three arrays below */
    uint8_t  max_cand, max_refs, max_l0; /* pa_me_data->max_cand / max_refs / max_l0 */
    uint8_t  n_ref[SVTME_MAX_LISTS]; /* tpl_data.tpl_ref0_count / tpl_ref1_count: ref_key[l][r] is set for r < n_ref[l]; a
                                        candidate that names another reference makes the call fail with SVTME_ERR_BAD_ARG */
    uint8_t  subsample_tx; /* tpl_ctrls.subsample_tx */
    uint8_t  pf_shape; /* tpl_ctrls.pf_shape: DEFAULT_SHAPE 0, N2_SHAPE 1, N4_SHAPE 2 (EbDefinitions.h) */
    uint8_t  src_dist; /* 1: fill svtme_tpl_block::src_recon_error */
    /* [0] dc, [1] ac of the luma quantiser at the dispenser's qindex: scs->quants_8bit.y_round_fp[q], .y_quant_fp[q],
     * scs->deq_8bit.y_dequant_qtx[q] (:579-586); read when src_dist is set */
    int16_t  y_round_fp[2], y_quant_fp[2], y_dequant[2];
    uint8_t  intra_mode_end; /* tpl_opt_flag 0 only: the last intra mode tried (:730-733): PAETH_PRED 12 by default,
                                SMOOTH_H_PRED 11 without scs->enable_paeth, D67_PRED 8 without enable_smooth either */
    uint8_t  reserved[3];
    const uint8_t  *total_me_candidate_index; /* [n_b64][n_pu] */
    const uint8_t  *me_candidate_array; /* [n_b64][n_pu * max_cand], MeCandidate bytes */
    const uint32_t *me_mv_array; /* [n_b64][n_pu * max_refs], MvCandidate::as_int */
} svtme_tpl_picture;

typedef struct svtme_tpl_block {
    int64_t best_intra_cost; /* INT64_MAX when intra prediction is disabled (:636) */
    int64_t best_inter_cost; /* INT64_MAX when no candidate passed the gates (:635) */
    int16_t mv_x, mv_y; /* final_best_mv.col / .row: 1/8 pel, clamped (:853-876); 0 without a candidate */
    int8_t  best_rf_idx; /* svt_get_ref_frame_type(list, ref) - 1, -1 without a candidate (:619, :850) */
    uint8_t best_is_inter; /* best_inter_cost < best_intra_cost: best_mode is NEWMV, else DC_PRED (:932-933) */
    uint8_t best_intra_mode; /* PredictionMode of best_intra_cost: DC_PRED (0) unless tpl_opt_flag is 0 */
    uint8_t valid; /* 1: the block belongs to the dispenser's grid */
    int64_t src_recon_error; /* src_dist and best_is_inter: get_quantize_error's recon_error of the chosen candidate; else 0 */
    int64_t src_rate; /* the same with tpl_opt_flag 0: rate_estimator(qcoeff, eob, tx_size) of the chosen candidate, from
                         which the encoder forms srcrf_rate = (rate << TPL_DEP_COST_SCALE_LOG2) << subsample_tx; else 0 */
} svtme_tpl_block;
int svtme_tpl_search(svtme_engine *e, const svtme_tpl_picture *pic, svtme_tpl_block *out /* [n_b64][SVTME_TPL_BLOCKS] */);

/*
 * The dispenser's reconstruction path, the second half of tpl_mc_flow_dispenser_sb_generic
 * (EbSourceBasedOperationsProcess.c:1004-1197), for the 16x16 blocks of dispenser_search_level 0 (tpl levels 1 to 6;
 * SVTME_ERR_UNSUPPORTED otherwise). `blocks` is what svtme_tpl_search returned for the picture with src_dist set - or what the
 * encoder kept of it between TPL groups (tpl_src_stats_buffer, :983-1002). Per block: the prediction - the reference
 * picture's block at the full-pel vector for NEWMV (:1005-1030), else DC from the reconstructed neighbours
 * (intra_dc_sad_path, :1044-1078) or best_intra_mode with update_neighbor_samples_array_open_loop_mb_recon,
 * filter_intra_edge and intra_prediction_open_loop_mb (:1080-1119) -, the residual over every (1 << subsample_tx)-th row,
 * svt_av1_wht_fwd_txfm with pf_shape, get_quantize_error, rate_estimator (:1123-1141), and, unless disable_intra_pred is
 * set, av1_inv_transform_recon8bit of the dequantised coefficients into the reconstructed picture with the rows between
 * copied from the row above (:1143-1177). ref_key names the pictures the predictions are copied from: source pictures, or
 * pictures reconstructed by earlier calls that the caller has put into the store under keys of their own
 * (encode_context_ptr->mc_flow_rec_picture_buffer for references inside the TPL group, :1011-1021).
 * stats: [n_b64][SVTME_TPL_BLOCKS], TplStats as result_model_store leaves them (:211-222, :1179-1197): srcrf / recrf
 * distortion and rate, divided by the block size, at least 1; zero for blocks outside the grid. recon: NULL, or width x
 * height bytes of recon_stride that receive the reconstructed picture. The blocks that depend on reconstructed neighbours
 * (the intra ones) run as a wavefront over the picture; the others all at once. Synchronous; shares its stream with the
 * other svtme_tpl_* / svtme_gm_* calls.
 */
typedef struct svtme_tpl_stats {
    int64_t srcrf_dist, recrf_dist, srcrf_rate, recrf_rate;
} svtme_tpl_stats;
int svtme_tpl_recon(svtme_engine *e, const svtme_tpl_picture *pic, const svtme_tpl_block *blocks /* [n_b64][SVTME_TPL_BLOCKS] */,
                    svtme_tpl_stats *stats /* [n_b64][SVTME_TPL_BLOCKS] */, uint8_t *recon, uint32_t recon_stride);
/*
 * Keeps the picture the engine's last successful svtme_tpl_recon reconstructed as a picture of the store under `key`, device
 * to device: what the dispenser's caller does with encode_context_ptr->mc_flow_rec_picture_buffer[frame_idx] for the later
 * pictures of the TPL group that reference it (tpl_mc_flow_dispenser pads it with generate_padding,
 * EbSourceBasedOperationsProcess.c:1408-1414; references inside the group are read from there, :1011-1017). The store builds
 * its padded planes from the reconstruction as svtme_store_put builds them from a source picture - no copy to the host and
 * back. SVTME_ERR_NOT_FOUND without a reconstructed picture; otherwise svtme_store_put's codes. Release the key like any other.
 */
int svtme_tpl_recon_keep(svtme_engine *e, uint64_t key, uint64_t picture_number);

const char *svtme_version(void);

#ifdef __cplusplus
}
#endif
#endif /* SVTME_H */
