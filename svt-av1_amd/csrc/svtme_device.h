// svtme_device.h - device-side data layout shared by the HIP kernels and the host engine.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/svtme.h"

namespace svtme {

// One padded 8-bit plane of the device picture store. `pix` points at pixel (0,0); rows -pad..height+pad-1 and
// columns -pad..width+pad-1 are valid. stride is a multiple of 64 bytes and `pix` is 4-byte aligned, so a row's
// byte alignment is the same for every row.
struct DevPlane {
    uint8_t *pix;
    int32_t  stride;
    int32_t  width;
    int32_t  height;
    int32_t  pad;
};

// lvl[0] = sixteenth, lvl[1] = quarter, lvl[2] = full: get_me_reference's level numbering
// (Source/Lib/Encoder/Codec/EbMotionEstimation.c:1247-1273).
struct DevPicture {
    DevPlane lvl[3];
};

// HME results of one (b64, list, ref). Region index = sr_w * 2 + sr_h, i.e. the memory order of the reference's
// [EB_HME_SEARCH_AREA_COLUMN_MAX_COUNT][EB_HME_SEARCH_AREA_ROW_MAX_COUNT] arrays (EbMotionEstimationContext.h:404-427).
struct HmeRefOut {
    int16_t  lx[3][4];
    int16_t  ly[3][4];
    uint32_t lsad[3][4];
    int16_t  pre_x[2];
    int16_t  pre_y[2];
    uint32_t pre_sad[2];
    uint32_t zz_sad;
    uint32_t valid; // bit0: HME ran for this reference (not a base-layer list-1 reference)
    // level-0 results before the worst region gives way to a better pre-HME result
    int16_t  raw_x[4];
    int16_t  raw_y[4];
    uint32_t raw_sad[4];
};

// Integer-search window of one (b64, list, ref), window origin relative to the b64 origin; w == 0: not searched.
struct MeJob {
    int16_t org_x, org_y, w, h;
};

// Work items of the integer search. A 64-position column chunk of a window (`wc` positions wide) has n_xg position quads;
// per group of 8 search rows it is cut into U units - a quad's 4 positions x 8 rows, or with at most 8 quads one row
// parity of them -, numbered group by group, and an item takes UPI consecutive units: 16, one per quad of lanes, where any
// 16 consecutive units lie in at most two groups of rows (U = 8, 12, 14, 16); else the units of one group (U = 10) or of
// two (chunks of one or two quads).
__host__ __device__ inline void me_item_units(int wc, int &n_xg, int &U, int &UPI) {
    n_xg = (wc + 3) >> 2;
    U    = n_xg <= 8 ? 2 * n_xg : n_xg;
    UPI  = (U >= 12 || U == 8) ? 16 : (U == 10 ? 10 : 2 * U);
}
// items of a w x h window: an item per group of rows for every 64-position chunk but the last, whose units are packed
__host__ __device__ inline int me_item_count(int w, int h) {
    const int G = (h + 7) >> 3, full = (w - 1) / 64;
    int       n_xg, U, UPI;
    me_item_units(w - full * 64, n_xg, U, UPI);
    return full * G + (G * U + UPI - 1) / UPI;
}

// Everything a picture's kernels need. A launch covers the pictures of one batch: their records lie next to each other in
// device memory and a workgroup picks its picture from its index (blockIdx.x / n_b64, or a grid dimension of its own).
// The pictures of a batch have one size, one set of controls and one me_type; they differ in planes, reference lists,
// distances, layer and buffers.
struct PicParams {
    DevPicture     cur;
    DevPicture     ref[SVTME_MAX_LISTS][SVTME_MAX_REFS];
    int32_t        dist[SVTME_MAX_LISTS][SVTME_MAX_REFS]; // |poc - ref poc| as uint16 (get_me_reference :1269)
    svtme_controls ctrl;
    int32_t        width, height; // input_ptr->width/height == pcs->aligned_width/height
    int32_t        b64_w, b64_h;
    uint32_t       b64_w_magic; // ceil(2^32 / b64_w): block row = (b64 * magic) >> 32 (b64_w > 1)
    int32_t        n_list;
    int32_t        n_ref[SVTME_MAX_LISTS];
    int32_t        temporal_layer;
    int32_t        is_ref;
    int32_t        serial_refs; // 1: one workgroup walks all references of a b64 in order (the serial HME kernel)
    int16_t        dep_pre; // list-1 pre-HME searches read the list-0 results (check_prehme_early_exit :1759-1768)
    int16_t        dep_l0; // the level-0 area of every reference but the first depends on the first one's vector (:1838-1850)
    int32_t        me_type; // SVTME_ME_OPEN_LOOP / SVTME_ME_MCTF / SVTME_ME_FIRST_PASS
    int16_t        bypass_zz_check; // first pass only
    int16_t        b64_step; // > 1: only the 64x64 blocks whose column and row index are multiples of it are searched
    int32_t        restricted; // integer search windows stay inside the block's tile (restricted_motion_vector)
    uint32_t       tf_me_exit_th;
    int32_t        me_max_items; // upper bound of search work items per (b64, ref) for this picture's controls
    int32_t        reserved_me; // (unused)
    int32_t        zero; // always 0, but the compiler cannot know: keeps it from merging loads that must stay apart
    const svtme_tile_rect *tiles; // [n_b64] the blocks' tiles for restricted windows; nullptr: the picture is the tile
    // intermediates
    MeJob              *me_jobs; // [n_b64][2][4]
    unsigned long long *me_keys; // [n_b64][2][4][85] running minima (SAD << 32 | y << 16 | x), window-relative
    // outputs
    HmeRefOut           *hme_out; // [n_b64][2][4]
    uint32_t            *best_sad; // [n_ref[0] + n_ref[1]][n_b64][85]: a plane per reference, list 0 first
    uint32_t            *best_mv; // same shape
    svtme_search_result *search_results; // [n_b64][2][4]
    uint32_t            *zz_sad; // [n_b64][2][4]
    uint32_t            *sr_divisor; // [n_b64][2][4]
    // abs-diff-accumulate operations actually performed, as SVTME_OPS_SLOTS partial sums of 8 counters each (spread so
    // that the workgroups of a picture do not all hit one address): [0] zero-MV / centre checks, [1] pre-HME,
    // [2..4] HME L0..L2, [5] integer search
    unsigned long long  *ops;
    unsigned long long  *ops_out; // where me_finish_kernel leaves a copy of the counters for the result transfer
    // optional host tail on the device (tail.enable)
    svtme_tail          tail;
    uint8_t            *t_total_cand; // [n_b64][85]
    uint32_t           *t_mv_array; // [n_b64][85 * 7]
    uint8_t            *t_cand_array; // [n_b64][85 * 23]
    uint32_t           *t_distortion; // [n_b64][85]
    uint32_t           *t_stats; // [n_b64][8]
    uint8_t            *t_do_ref; // [n_b64][8]
    unsigned long long *t_hme_sad; // [n_b64][8]
    uint8_t            *t_zero_begin; // the tail's output arrays as one block (a multiple of 64 bytes), cleared by
    size_t              t_zero_bytes; // me_finish_kernel before the tail kernel runs
};

constexpr int SVTME_OPS_SLOTS = 64;
constexpr int SVTME_OPS_BYTES = SVTME_OPS_SLOTS * 8 * 8;

// How a kernel receives its pictures' records. One picture: the record itself is the kernel argument (it arrives with
// the dispatch, in scalar registers' reach, nothing to fetch). A batch: a pointer to the records in device memory.
struct OnePicture {
    PicParams p;
    __device__ __forceinline__ const PicParams &operator[](int) const { return p; }
};
// The records of a batch were written by an earlier kernel and do not change while the search kernels run: they are read
// through the constant address space, so that the compiler may fetch a field again (as it does with kernel arguments)
// instead of keeping hundreds of bytes of plane descriptors alive in registers and spilling them.
typedef const PicParams __attribute__((address_space(4))) ConstPicParams;
struct ManyPictures {
    const PicParams *pp;
    __device__ __forceinline__ ConstPicParams &operator[](int i) const { return ((ConstPicParams *)pp)[i]; }
};
// A plane descriptor out of a record in either address space, as a value.
template <class T>
__device__ __forceinline__ DevPlane load_plane(const T &src) {
    DevPlane d;
    d.pix    = src.pix;
    d.stride = src.stride;
    d.width  = src.width;
    d.height = src.height;
    d.pad    = src.pad;
    return d;
}

// The pictures one set of launches covers.
struct Batch {
    const PicParams *host; // n records: the host's copies of what `dev` holds
    const PicParams *dev; // the same n records in device memory
    int              n;
    int              n_b64() const { return host[0].b64_w * host[0].b64_h; }
    int              max_refs() const {
        int m = 0;
        for (int i = 0; i < n; ++i) {
            const int r = host[i].n_ref[0] + (host[i].n_list == 2 ? host[i].n_ref[1] : 0);
            m           = r > m ? r : m;
        }
        return m;
    }
};

// store kernels (store_kernels.hip)
void launch_store_build(const uint8_t *luma, int luma_stride, const DevPicture &pic, int hme_level1, hipStream_t s);
// the same for n <= SVTME_MAX_PUT_BATCH pictures of one size in one launch of each kernel
struct StoreBatch {
    const uint8_t *luma[SVTME_MAX_PUT_BATCH]; // un-padded luma in device memory
    int32_t        stride[SVTME_MAX_PUT_BATCH];
    DevPicture     pic[SVTME_MAX_PUT_BATCH];
    unsigned long long m_side[3], m_pw[3]; // per plane: ceil(2^40 / (2 pad)), ceil(2^40 / (width + 2 pad)); filled by the launcher
};
void launch_store_build_many(StoreBatch &b, int n, int hme_level1, hipStream_t s);
// search kernels
// Each launcher records marks[i] (when not null) on the stream after its i-th kernel (timing only). The buffers hold
// zeroed hme_out and operation counters on entry (me_finish_kernel leaves them so).
void launch_params_write(const PicParams *host, PicParams *dev, int n, hipStream_t s); // n records -> device memory, by kernel
void launch_hme(const Batch &b, hipStream_t s, hipEvent_t marks[2]); // picks the serial or the itemised kernels
void launch_hme_items(const Batch &b, hipStream_t s, hipEvent_t marks[2]);
void launch_hme_l12_regions(const Batch &b, hipStream_t s); // hme_regions.hip: levels 1 / 2, a work item per (b64, reference)
void launch_me(const Batch &b, hipStream_t s, hipEvent_t marks[3]);
void launch_tail(const Batch &b, hipStream_t s); // after launch_me, when the pictures have a device tail

// global-motion kernels (gm_kernels.hip). Corner lists: [0] = count, (x, y) pairs from [2] on.
void launch_gm_corners(const DevPlane &pl, uint8_t *score, uint8_t *keep, int *row_count, int max_points, int *out, hipStream_t s);
// one workgroup per possible corner (n_frm_max); out[i] = {x, y, rx, ry, matched} for the corners the list holds
void launch_gm_match(const DevPlane &frm, const DevPlane &ref, const int *frm_corners, const int *ref_corners, int n_frm_max, int *out,
                     hipStream_t s);
// wm == nullptr: the frame error (no warp)
void launch_gm_error(const DevPlane &frm, const DevPlane &ref, const svtme_gm_warp *wm, const int16_t *filter, const int32_t *lut,
                     uint32_t *block_err, long long best_error, long long *result, hipStream_t s);

// svtme_gm_refine: where svt_av1_refine_integerized_param's loop (global_motion.c:135-263) stands, kept in device memory
// between the evaluations of a chain
struct GmRefineState {
    int32_t   mat[6]; // wm->wmmat[0..5] as the loop holds it (the next evaluation's model)
    int16_t   alpha, beta, gamma, delta; // wm's shear as the last svt_get_shear_params left it
    int32_t   wmtype; // the type the loop forces; the resulting type once done
    int32_t   n_refinements, n_params;
    int32_t   phase, i, p, step, step_dir, curr_param, best_param; // the loop's position
    int32_t   skip; // the next evaluation's result is 1 without a warp (the shear derivation failed)
    int32_t   done, evals;
    long long best_error;
};
void launch_gm_refine_begin(GmRefineState *state, const GmRefineState &init, hipStream_t s);
// `count` evaluations of the chain: two launches each (the blocks' errors; their sum and the loop's next move)
void launch_gm_refine_evaluations(const DevPlane &frm, const DevPlane &ref, GmRefineState *state, const int16_t *filter, const int32_t *lut,
                                  uint32_t *block_err, int count, hipStream_t s);

// TPL dispenser, source-based mode search (tpl_kernels.hip): the planes and the ME results of one picture
struct TplParams {
    DevPlane cur; // full-resolution planes
    DevPlane ref[SVTME_MAX_LISTS][SVTME_MAX_REFS];
    uint8_t  ref_skip[SVTME_MAX_LISTS][SVTME_MAX_REFS];
    int32_t  n_ref[SVTME_MAX_LISTS];
    int32_t  width, height, b64_w, n_b64;
    int32_t  level, disable_intra_pred, i_slice, enable_me_16x16; // svtme_tpl_picture's fields
    int32_t  n_pu, max_cand, max_refs, max_l0;
    int32_t  sad_intra, sad_inter, pf_shape, src_dist, subsample_tx;
    int32_t  mode_end; // the last intra mode tried: 0 (DC_PRED only) with tpl_opt_flag, else svtme_tpl_picture::intra_mode_end
    int32_t  rate; // also fill svtme_tpl_block::src_rate (no tpl_opt_flag, src_dist)
    int32_t  round_fp[2], quant_fp[2], dequant[2]; // [dc, ac]
    int32_t  cos13[32], cos12[32]; // cos(2 k pi / 128) with 13 / 12 fractional bits, k = 0..31
    const uint8_t  *cnt; // [n_b64][n_pu]
    const uint8_t  *cand; // [n_b64][n_pu * max_cand]
    const uint32_t *mv; // [n_b64][n_pu * max_refs]
    svtme_tpl_block *out; // [n_b64][SVTME_TPL_BLOCKS], zeroed
    int             *error; // bit 0: a candidate named a reference beyond n_ref
    // svtme_tpl_recon: `out` holds the caller's block records; the reconstructed picture and the statistics
    uint8_t         *recon;
    int32_t          recon_stride, dc_path; // dc_path: intra blocks are predicted from the DC of their neighbours (intra_dc_sad_path)
    svtme_tpl_stats *stats; // [n_b64][SVTME_TPL_BLOCKS], zeroed
};
// dev: the record in device memory; transform_costs: it asks for costs in the transform domain (every intra mode, or
// sad_intra / sad_inter not set): the kernel that takes them four at a time
void launch_tpl_search(const TplParams *dev, int n_b64, bool transform_costs, hipStream_t s);
// the reconstruction path of a picture: the inter blocks in one launch, then the intra blocks level by level (list: the
// blocks, bx | by << 16, sorted by level; level_count[l]: how many level l has)
void launch_tpl_recon(const TplParams *dev, int n_b64, const uint32_t *list, const int *level_count, int n_levels, hipStream_t s);

int  occupancy_me_search();
void occupancy_hme_items(int out[2]);

} // namespace svtme
