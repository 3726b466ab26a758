/*
 * gm_driver.c - TEST INFRASTRUCTURE, not part of the product.
 *
 * Drives the REAL reference global-motion estimation (compute_global_motion,
 * Source/Lib/Encoder/Codec/EbGlobalMotionEstimation.c:155-301) over pairs of synthetic pictures and records what it
 * computed: the final model and, in call order, every svt_av1_fast_corner_detect, svt_av1_determine_correspondence,
 * svt_av1_warp_error, svt_av1_frame_error and svt_av1_refine_integerized_param call with its inputs and result. Compiled only where /root/reference exists,
 * against the reference's own headers and objects (oracle/Makefile, target ref); the binary lands in oracle/_ref/.
 *
 * These functions are the seam of the GPU engine's svtme_gm_* entry points (include/svtme.h). They are reached through
 * the linker's --wrap: the reference's objects call __wrap_<name> below, which either passes the call on to the
 * reference's own function (mode 0) or sends it to the engine (mode 1, the reference-side binding end to end: the
 * reference's RANSAC, refinement loop, shear derivation and costs run unchanged around the engine's results; mode 1 + 4:
 * the refinement loop as a whole goes to the engine too, svtme_gm_refine, and the reference's own loop does not run).
 *
 * Usage: ref_gm <job file> <output file> [libsvtme.so, mode 1 only]
 *        ref_gm tables <output file>      the warp filter (193 x 8 int16) and the error table (512 int32) of the build,
 *                                         for tests that need to hand them to the oracle or to the engine
 */
#include <dlfcn.h>
#include <stddef.h>
#include <stdint.h>

#include "EbDefinitions.h"
#include "EbPictureControlSet.h"
#include "EbGlobalMotionEstimation.h"
#include "EbWarpedMotion.h"
#include "EbEncWarpedMotion.h"
#include "global_motion.h"
#include "corner_detect.h"
#include "corner_match.h"
#include "aom_dsp_rtcd.h"
#include "common_dsp_rtcd.h"

#include "../include/svtme.h"
#include "ref_common.h"

void setup_common_rtcd_internal(CPU_FLAGS flags);
/* AVX2 kernels of the reference, compiled stand-alone (oracle/Makefile); prototypes as in common_dsp_rtcd.h:2208 and
 * aom_dsp_rtcd.h:908, which show them on x86 builds only */
#ifdef REF_HAVE_AVX2
void    svt_av1_warp_affine_avx2(const int32_t *mat, const uint8_t *ref, int width, int height, int stride, uint8_t *pred, int p_col,
                                 int p_row, int p_width, int p_height, int p_stride, int subsampling_x, int subsampling_y,
                                 ConvolveParams *conv_params, int16_t alpha, int16_t beta, int16_t gamma, int16_t delta);
int64_t svt_av1_calc_frame_error_avx2(const uint8_t *const ref, int stride, const uint8_t *const dst, int p_width, int p_height,
                                      int p_stride);
#endif

#define GM_JOB_MAGIC "SVTGMJ02"
#define GM_OUT_MAGIC "SVTGMO01"

#pragma pack(push, 1)
typedef struct gm_job_header {
    char     magic[8];
    uint32_t width, height; /* full-resolution luma, multiples of 8 */
    uint32_t n_frames; /* followed by n_frames * width * height luma bytes */
    uint32_t n_jobs; /* then n_jobs gm_job */
    uint32_t mode; /* 0: the reference alone; 1: the four seam functions on the GPU engine; +2: the reference's AVX2 kernels
                      for the warp filter and the frame error (as setup_rtcd_internal binds them on an
                      AVX2 machine, aom_dsp_rtcd.c / common_dsp_rtcd.c), for the CPU side of a timing comparison;
                      +4 (with 1): svt_av1_refine_integerized_param as a whole on the engine */
} gm_job_header;
typedef struct gm_job {
    uint32_t cur_frame, ref_frame;
    uint32_t downsample_level; /* gm_ctrls.downsample_level */
    uint32_t rotzoom_model_only, params_refinement_steps, allow_high_precision_mv;
    /* direct != 0: not compute_global_motion but one svt_av1_refine_integerized_param call on the pair, with this model
     * (zero shear), forced type and error to beat - for models the estimation itself does not arrive at */
    uint32_t direct;
    int32_t  mat[6];
    int32_t  wmtype;
    int64_t  best_frame_error;
} gm_job;
#pragma pack(pop)

/* ---- the record of one run: int64 words. Each entry: kind, number of payload words, payload. */
enum { REC_CORNERS = 1, REC_CORRESPONDENCES = 2, REC_WARP_ERROR = 3, REC_FRAME_ERROR = 4, REC_REFINE = 5, REC_FINAL = 9 };
static FILE *g_out;
static void rec(int64_t kind, const int64_t *payload, int64_t n) {
    fwrite(&kind, 8, 1, g_out);
    fwrite(&n, 8, 1, g_out);
    if (n) fwrite(payload, 8, (size_t)n, g_out);
}

/* ---- the engine, when the seam functions go there */
typedef struct Shim {
    void         *lib;
    svtme_engine *engine;
    int (*create)(const svtme_config *, svtme_engine **);
    void (*destroy)(svtme_engine *);
    int (*store_put)(svtme_engine *, uint64_t, uint64_t, const uint8_t *, uint32_t, uint32_t, uint32_t);
    int (*store_release)(svtme_engine *, uint64_t);
    int (*gm_set_tables)(svtme_engine *, const int16_t *, const int32_t *);
    int (*gm_corners)(svtme_engine *, uint64_t, int, int32_t *, int32_t, int32_t *);
    int (*gm_correspondences)(svtme_engine *, uint64_t, uint64_t, int, int32_t *, int32_t *);
    int (*gm_warp_error)(svtme_engine *, uint64_t, uint64_t, int, const svtme_gm_warp *, int64_t, int64_t *);
    int (*gm_frame_error)(svtme_engine *, uint64_t, uint64_t, int, int64_t *);
    int (*gm_refine)(svtme_engine *, uint64_t, uint64_t, int, svtme_gm_model *, int, int, int64_t, int64_t *, uint32_t *);
} Shim;
static Shim g_shim;
static int  g_use_engine, g_engine_refine;
/* the pair compute_global_motion is working on */
static const uint8_t *g_frm_buf, *g_ref_buf;
static uint64_t       g_frm_key, g_ref_key;
static int            g_level;

static void *shim_sym(const char *name) {
    void *p = dlsym(g_shim.lib, name);
    if (!p) {
        fprintf(stderr, "ref_gm: %s missing from the engine library\n", name);
        exit(2);
    }
    return p;
}

/* ---- the seam ---------------------------------------------------------------------------------------------------- */
int __real_svt_av1_fast_corner_detect(unsigned char *buf, int width, int height, int stride, int *points, int max_points);
int __wrap_svt_av1_fast_corner_detect(unsigned char *buf, int width, int height, int stride, int *points, int max_points) {
    int n;
    if (g_use_engine) {
        const uint64_t key = buf == g_frm_buf ? g_frm_key : g_ref_key;
        if (buf != g_frm_buf && buf != g_ref_buf) die("corner detection on an unknown plane");
        int32_t got = 0;
        if (g_shim.gm_corners(g_shim.engine, key, g_level, points, max_points, &got) != SVTME_OK) die("svtme_gm_corners failed");
        n = got;
    } else
        n = __real_svt_av1_fast_corner_detect(buf, width, height, stride, points, max_points);
    int64_t *w = (int64_t *)malloc(sizeof(int64_t) * (2 + 2 * (size_t)n));
    w[0] = buf == g_frm_buf ? 0 : 1;
    w[1] = n;
    for (int i = 0; i < 2 * n; ++i) w[2 + i] = points[i];
    rec(REC_CORNERS, w, 2 + 2 * (int64_t)n);
    free(w);
    return n;
}

int __real_svt_av1_determine_correspondence(unsigned char *frm, int *frm_corners, int num_frm_corners, unsigned char *ref,
                                            int *ref_corners, int num_ref_corners, int width, int height, int frm_stride,
                                            int ref_stride, int *correspondence_pts);
int __wrap_svt_av1_determine_correspondence(unsigned char *frm, int *frm_corners, int num_frm_corners, unsigned char *ref,
                                            int *ref_corners, int num_ref_corners, int width, int height, int frm_stride,
                                            int ref_stride, int *correspondence_pts) {
    int n;
    if (g_use_engine) {
        /* the engine matches the corners it found itself for the two keys; they are the ones handed in here */
        static int32_t all[4 * SVTME_GM_MAX_CORNERS];
        int32_t        got = 0;
        if (frm != g_frm_buf || ref != g_ref_buf) die("correspondences on unknown planes");
        if (g_shim.gm_correspondences(g_shim.engine, g_frm_key, g_ref_key, g_level, all, &got) != SVTME_OK)
            die("svtme_gm_correspondences failed");
        if (got > num_frm_corners) die("more correspondences than frame corners");
        memcpy(correspondence_pts, all, sizeof(int32_t) * 4 * (size_t)got);
        n = got;
    } else
        n = __real_svt_av1_determine_correspondence(frm, frm_corners, num_frm_corners, ref, ref_corners, num_ref_corners, width,
                                                    height, frm_stride, ref_stride, correspondence_pts);
    int64_t *w = (int64_t *)malloc(sizeof(int64_t) * (1 + 4 * (size_t)n));
    w[0] = n;
    for (int i = 0; i < 4 * n; ++i) w[1 + i] = correspondence_pts[i];
    rec(REC_CORRESPONDENCES, w, 1 + 4 * (int64_t)n);
    free(w);
    return n;
}

int64_t __real_svt_av1_warp_error(EbWarpedMotionParams *wm, int use_hbd, int bd, const uint8_t *ref, const uint8_t *ref_2b,
                                  int width, int height, int stride, uint8_t *dst, int p_col, int p_row, int p_width,
                                  int p_height, int p_stride, int subsampling_x, int subsampling_y, int64_t best_error);
int64_t __wrap_svt_av1_warp_error(EbWarpedMotionParams *wm, int use_hbd, int bd, const uint8_t *ref, const uint8_t *ref_2b,
                                  int width, int height, int stride, uint8_t *dst, int p_col, int p_row, int p_width,
                                  int p_height, int p_stride, int subsampling_x, int subsampling_y, int64_t best_error) {
    int64_t err;
    if (g_use_engine) {
        if (use_hbd || p_col || p_row || subsampling_x || subsampling_y || ref != g_ref_buf || dst != g_frm_buf)
            die("warp error outside what the engine covers");
        /* EbEncWarpedMotion.c:233-235, the reference's own shear derivation */
        if (wm->wmtype <= AFFINE && !svt_get_shear_params(wm)) err = 1;
        else {
            svtme_gm_warp g;
            for (int i = 0; i < 6; ++i) g.wmmat[i] = wm->wmmat[i];
            if (wm->wmtype == ROTZOOM) { /* svt_warp_plane, EbWarpedMotion.c:701-704 */
                g.wmmat[5] = wm->wmmat[5] = wm->wmmat[2];
                g.wmmat[4] = wm->wmmat[4] = -wm->wmmat[3];
            }
            g.alpha = wm->alpha; g.beta = wm->beta; g.gamma = wm->gamma; g.delta = wm->delta;
            if (g_shim.gm_warp_error(g_shim.engine, g_frm_key, g_ref_key, g_level, &g, best_error, &err) != SVTME_OK)
                die("svtme_gm_warp_error failed");
        }
    } else
        err = __real_svt_av1_warp_error(wm, use_hbd, bd, ref, ref_2b, width, height, stride, dst, p_col, p_row, p_width,
                                        p_height, p_stride, subsampling_x, subsampling_y, best_error);
    /* what the call saw: the matrix and the shear as they stand after the call (both paths derive the shear first) */
    int64_t w[13] = {wm->wmmat[0], wm->wmmat[1], wm->wmmat[2], wm->wmmat[3], wm->wmmat[4], wm->wmmat[5],
                     wm->alpha,    wm->beta,     wm->gamma,    wm->delta,    wm->wmtype,   best_error,   err};
    rec(REC_WARP_ERROR, w, 13);
    return err;
}

int64_t __real_svt_av1_frame_error(int use_hbd, int bd, const uint8_t *ref, int stride, uint8_t *dst, int p_width, int p_height,
                                   int p_stride);
int64_t __wrap_svt_av1_frame_error(int use_hbd, int bd, const uint8_t *ref, int stride, uint8_t *dst, int p_width, int p_height,
                                   int p_stride) {
    int64_t err;
    if (g_use_engine) {
        if (use_hbd || ref != g_ref_buf || dst != g_frm_buf) die("frame error outside what the engine covers");
        if (g_shim.gm_frame_error(g_shim.engine, g_frm_key, g_ref_key, g_level, &err) != SVTME_OK) die("svtme_gm_frame_error failed");
    } else
        err = __real_svt_av1_frame_error(use_hbd, bd, ref, stride, dst, p_width, p_height, p_stride);
    rec(REC_FRAME_ERROR, &err, 1);
    return err;
}

int64_t __real_svt_av1_refine_integerized_param(EbWarpedMotionParams *wm, TransformationType wmtype, int use_hbd, int bd, uint8_t *ref,
                                                uint8_t *ref_2b, int r_width, int r_height, int r_stride, uint8_t *dst, int d_width,
                                                int d_height, int d_stride, int n_refinements, int64_t best_frame_error);
int64_t __wrap_svt_av1_refine_integerized_param(EbWarpedMotionParams *wm, TransformationType wmtype, int use_hbd, int bd, uint8_t *ref,
                                                uint8_t *ref_2b, int r_width, int r_height, int r_stride, uint8_t *dst, int d_width,
                                                int d_height, int d_stride, int n_refinements, int64_t best_frame_error) {
    /* in: the model, the type to force, the steps, the error to beat; out: the model, its type, the shear, the result */
    int64_t w[26] = {wm->wmmat[0], wm->wmmat[1], wm->wmmat[2], wm->wmmat[3], wm->wmmat[4], wm->wmmat[5], wm->alpha, wm->beta,
                     wm->gamma,    wm->delta,    wm->wmtype,   wmtype,       n_refinements, best_frame_error};
    int64_t err;
    if (g_engine_refine) {
        if (use_hbd || ref != g_ref_buf || dst != g_frm_buf) die("refinement outside what the engine covers");
        svtme_gm_model m;
        for (int i = 0; i < 6; ++i) m.wmmat[i] = wm->wmmat[i];
        m.alpha = wm->alpha; m.beta = wm->beta; m.gamma = wm->gamma; m.delta = wm->delta;
        m.wmtype = wm->wmtype;
        if (g_shim.gm_refine(g_shim.engine, g_frm_key, g_ref_key, g_level, &m, (int)wmtype, n_refinements, best_frame_error, &err, NULL) !=
            SVTME_OK)
            die("svtme_gm_refine failed");
        for (int i = 0; i < 6; ++i) wm->wmmat[i] = m.wmmat[i];
        wm->wmmat[6] = wm->wmmat[7] = 0; /* force_wmtype, global_motion.c:130 */
        wm->alpha = m.alpha; wm->beta = m.beta; wm->gamma = m.gamma; wm->delta = m.delta;
        wm->wmtype = (TransformationType)m.wmtype;
    } else
        err = __real_svt_av1_refine_integerized_param(wm, wmtype, use_hbd, bd, ref, ref_2b, r_width, r_height, r_stride, dst, d_width,
                                                      d_height, d_stride, n_refinements, best_frame_error);
    for (int i = 0; i < 6; ++i) w[14 + i] = wm->wmmat[i];
    w[20] = wm->alpha; w[21] = wm->beta; w[22] = wm->gamma; w[23] = wm->delta;
    w[24] = wm->wmtype;
    w[25] = err;
    rec(REC_REFINE, w, 26);
    return err;
}

/* ------------------------------------------------------------------------------------------------------------------ */
static uint8_t *plane_origin(EbPictureBufferDesc *d) { return d->buffer_y + d->origin_x + (size_t)d->origin_y * d->stride_y; }

int main(int argc, char **argv) {
    if (argc == 3 && !strcmp(argv[1], "tables")) {
        FILE *o = fopen(argv[2], "wb");
        if (!o) die("cannot open output file");
        fwrite(eb_warped_filter, sizeof(int16_t), 193 * 8, o);
        fwrite(error_measure_lut, sizeof(int32_t), 512, o);
        fclose(o);
        return 0;
    }
    if (argc < 3) die("usage: ref_gm <job> <out> [libsvtme.so]");
    FILE *f = fopen(argv[1], "rb");
    if (!f) die("cannot open job file");
    gm_job_header hdr;
    if (fread(&hdr, sizeof(hdr), 1, f) != 1 || memcmp(hdr.magic, GM_JOB_MAGIC, 8)) die("bad job header");
    const uint32_t w = hdr.width, h = hdr.height;
    if (w % 8 || h % 8 || w < 64 || h < 64) die("width/height must be multiples of 8 and >= 64");
    const size_t fsz = (size_t)w * h;
    uint8_t     *frames = (uint8_t *)malloc(fsz * hdr.n_frames);
    gm_job      *jobs = (gm_job *)calloc(hdr.n_jobs ? hdr.n_jobs : 1, sizeof(gm_job));
    if (fread(frames, fsz, hdr.n_frames, f) != hdr.n_frames) die("short frame data");
    if (hdr.n_jobs && fread(jobs, sizeof(gm_job), hdr.n_jobs, f) != hdr.n_jobs) die("short job data");
    fclose(f);

    setup_common_rtcd_internal(0);
    setup_rtcd_internal(0);
#ifdef REF_HAVE_AVX2
    if ((hdr.mode & 2) && __builtin_cpu_supports("avx2")) {
        svt_av1_warp_affine               = svt_av1_warp_affine_avx2;
        svt_av1_calc_frame_error          = svt_av1_calc_frame_error_avx2;
    }
#endif
    Planes *planes = (Planes *)calloc(hdr.n_frames, sizeof(Planes));
    for (uint32_t i = 0; i < hdr.n_frames; ++i) build_planes(&planes[i], frames + fsz * i, w, h);

    g_use_engine = hdr.mode & 1;
    if (g_use_engine) {
        if (argc < 4) die("mode 1 needs the path of libsvtme.so as third argument");
        g_shim.lib = dlopen(argv[3], RTLD_NOW | RTLD_LOCAL);
        if (!g_shim.lib) {
            fprintf(stderr, "ref_gm: %s\n", dlerror());
            return 2;
        }
        *(void **)&g_shim.create             = shim_sym("svtme_create");
        *(void **)&g_shim.destroy            = shim_sym("svtme_destroy");
        *(void **)&g_shim.store_put          = shim_sym("svtme_store_put");
        *(void **)&g_shim.store_release      = shim_sym("svtme_store_release");
        *(void **)&g_shim.gm_set_tables      = shim_sym("svtme_gm_set_tables");
        *(void **)&g_shim.gm_corners         = shim_sym("svtme_gm_corners");
        *(void **)&g_shim.gm_correspondences = shim_sym("svtme_gm_correspondences");
        *(void **)&g_shim.gm_warp_error      = shim_sym("svtme_gm_warp_error");
        *(void **)&g_shim.gm_frame_error     = shim_sym("svtme_gm_frame_error");
        *(void **)&g_shim.gm_refine          = shim_sym("svtme_gm_refine");
        g_engine_refine                      = (hdr.mode & 4) != 0;
        svtme_config cfg;
        memset(&cfg, 0, sizeof(cfg));
        cfg.max_width = w; cfg.max_height = h;
        cfg.store_slots = hdr.n_frames + 1; cfg.max_in_flight = 1;
        if (g_shim.create(&cfg, &g_shim.engine) != SVTME_OK) die("svtme_create failed");
        /* the encoder hands over its own tables */
        if (g_shim.gm_set_tables(g_shim.engine, &eb_warped_filter[0][0], (const int32_t *)error_measure_lut) != SVTME_OK)
            die("svtme_gm_set_tables failed");
        for (uint32_t i = 0; i < hdr.n_frames; ++i)
            if (g_shim.store_put(g_shim.engine, 1000 + i, i, frames + fsz * i, w, w, h) != SVTME_OK) die("svtme_store_put failed");
    }

    g_out = fopen(argv[2], "wb");
    if (!g_out) die("cannot open output file");
    fwrite(GM_OUT_MAGIC, 1, 8, g_out);
    PictureParentControlSet *pcs = (PictureParentControlSet *)calloc(1, sizeof(*pcs));
    for (uint32_t j = 0; j < hdr.n_jobs; ++j) {
        const gm_job *g = &jobs[j];
        if (g->cur_frame >= hdr.n_frames || g->ref_frame >= hdr.n_frames || g->downsample_level > 2) die("bad job");
        pcs->gm_ctrls.rotzoom_model_only      = (uint8_t)g->rotzoom_model_only;
        pcs->gm_ctrls.params_refinement_steps = (uint8_t)g->params_refinement_steps;
        pcs->gm_ctrls.downsample_level        = (uint8_t)g->downsample_level;
        /* the planes global_motion_estimation picks for the level, EbGlobalMotionEstimation.c:109-122 */
        Planes              *pc = &planes[g->cur_frame], *pr = &planes[g->ref_frame];
        EbPictureBufferDesc *in = g->downsample_level == GM_DOWN16 ? &pc->sixteenth : g->downsample_level == GM_DOWN ? &pc->quarter : &pc->full;
        EbPictureBufferDesc *rf = g->downsample_level == GM_DOWN16 ? &pr->sixteenth : g->downsample_level == GM_DOWN ? &pr->quarter : &pr->full;
        g_frm_buf = plane_origin(in);
        g_ref_buf = plane_origin(rf);
        g_frm_key = 1000 + g->cur_frame;
        g_ref_key = 1000 + g->ref_frame;
        g_level   = (int)g->downsample_level;
        EbWarpedMotionParams best;
        memset(&best, 0, sizeof(best));
        if (g->direct) {
            for (int i = 0; i < 6; ++i) best.wmmat[i] = g->mat[i];
            best.wmtype = (TransformationType)g->wmtype;
            svt_av1_refine_integerized_param(&best, (TransformationType)g->wmtype, EB_FALSE, EB_8BIT, g_ref_buf, NULL, rf->width, rf->height,
                                             rf->stride_y, (uint8_t *)g_frm_buf, in->width, in->height, in->stride_y,
                                             (int)g->params_refinement_steps, g->best_frame_error);
        } else
            compute_global_motion(pcs, in, rf, &best, (int)g->allow_high_precision_mv);
        int64_t fin[14] = {best.wmtype,   best.wmmat[0], best.wmmat[1], best.wmmat[2], best.wmmat[3], best.wmmat[4], best.wmmat[5],
                           best.wmmat[6], best.wmmat[7], best.alpha,    best.beta,     best.gamma,    best.delta,    best.invalid};
        rec(REC_FINAL, fin, 14);
    }
    fclose(g_out);
    if (g_use_engine) g_shim.destroy(g_shim.engine);
    return 0;
}
