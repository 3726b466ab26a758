/*
 * oracle_cli.c - TEST INFRASTRUCTURE. Runs a job file (oracle/ref_dump.h) through the CPU restatement
 * (me_oracle.c) and writes the same dump format oracle/_ref/ref_me writes, so the two can be compared byte for
 * byte. Also prints the algorithmic operation counts of the run as one JSON line.
 *
 * Usage: oracle_me <job file> <output file> [<handover file>, mode REF_MODE_TAIL only]
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "me_oracle.h"

static void die(const char *m) {
    fprintf(stderr, "oracle_me: %s\n", m);
    exit(2);
}

static double now_s(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec + ts.tv_nsec * 1e-9;
}

/* get_disallow_below_16x16_picture_level(enc_mode, resolution, B_SLICE, 0, 1, 0),
 * Source/Lib/Encoder/Codec/EbEncDecProcess.c:4489-4514, as called from EbPictureControlSet.c:1533 */
static uint8_t enable_me_8x8(int m, int res) {
    uint8_t disallow;
    if (m <= 8) disallow = 0;
    else if (m <= 9) disallow = res <= 4 ? 0 : 1;
    else if (m <= 11) disallow = res <= 2 ? 0 : 1;
    else disallow = 1;
    return !disallow;
}

/* scs->sb_params_array's tile fields for uniform tiles of (tile_b64 & 15) x (tile_b64 >> 4) 64x64 blocks, as
 * EbResourceCoordinationProcess.c:1304-1349 leaves them (see ref_job_pic::tile_b64) */
static svtme_tile_rect *make_tile_rects(const ref_job_header *h, uint8_t tile_b64) {
    const uint32_t   bw = (h->width + 63) / 64, bh = (h->height + 63) / 64, tw = tile_b64 & 15, th = tile_b64 >> 4;
    svtme_tile_rect *t = (svtme_tile_rect *)calloc((size_t)bw * bh, sizeof(*t));
    if (!t) die("out of memory");
    for (uint32_t i = 0; i < bw * bh; ++i) {
        const uint32_t bx = i % bw, by = i / bw;
        t[i].end_x = h->width;
        t[i].end_y = h->height;
        if (tw) {
            t[i].start_x = bx / tw * tw * 64;
            t[i].end_x   = t[i].start_x + tw * 64 < h->width ? t[i].start_x + tw * 64 : h->width;
        }
        if (th) {
            t[i].start_y = by / th * th * 64;
            t[i].end_y   = t[i].start_y + th * 64 < h->height ? t[i].start_y + th * 64 : h->height;
        }
    }
    return t;
}

static void fill_pic(orc_pic *p, const ref_job_header *h, const ref_job_pic *jp, const orc_planes *planes) {
    memset(p, 0, sizeof(*p));
    p->width = h->width;
    p->height = h->height;
    p->picture_number = jp->picture_number;
    p->temporal_layer = jp->temporal_layer;
    p->is_ref = jp->is_ref;
    p->n_list = jp->n_list;
    p->n_ref[0] = jp->n_ref[0];
    p->n_ref[1] = jp->n_list == 2 ? jp->n_ref[1] : 0;
    p->cur = &planes[jp->cur_frame];
    for (int l = 0; l < 2; ++l)
        for (int r = 0; r < 4; ++r) {
            p->ref[l][r] = &planes[jp->ref_frame[l][r]];
            p->ref_picture_number[l][r] = jp->ref_poc[l][r];
        }
    svtme_preset_key key;
    memset(&key, 0, sizeof(key));
    key.enc_mode = (int8_t)jp->enc_mode;
    key.sc_class1 = jp->sc_class1;
    key.input_resolution = orc_input_resolution(h->width, h->height);
    key.frame_rate = jp->frame_rate;
    if (jp->me_type == 1) {
        if (orc_derive_tf_controls(jp->tf_hme_me_level, &p->ctrl)) die("bad temporal-filter level");
        p->me_type = 1;
        p->tf_me_exit_th = jp->tf_me_exit_th;
    } else if (jp->me_type == 4) {
        if (orc_derive_first_pass_controls(key.input_resolution, jp->frame_rate, (jp->tf_hme_me_level >> 1) & 1, &p->ctrl))
            die("bad first-pass controls");
        p->me_type = 4;
        p->bypass_zz_check = jp->tf_hme_me_level & 1;
    } else if (orc_derive_controls(&key, &p->ctrl)) die("bad preset");
    p->input_resolution = key.input_resolution;
    p->enable_me_8x8 = enable_me_8x8(jp->enc_mode, key.input_resolution);
    p->enable_me_16x16 = jp->enc_mode <= 12; /* get_enable_me_16x16, EbPictureControlSet.c:90-99 */
    /* get_max_allocated_me_refs, EbPictureControlSet.c:83-89 */
    const uint32_t l0 = h->ref_count_used_list0, l1 = h->ref_count_used_list1;
    p->max_refs = (uint8_t)(l0 + l1);
    p->max_cand = (uint8_t)(l0 + l1 + l0 * l1 + (l0 - 1) + (l1 == 3 ? 1 : 0));
    p->max_l0 = (uint8_t)l0;
    p->gm_enabled = jp->gm_enabled;
    p->gm_use_distance_based_active_th = jp->gm_dist_th;
    p->restricted_motion_vector = jp->restricted_motion_vector;
    p->b64_step = h->b64_step;
    if (jp->tile_b64) p->tile_rects = make_tile_rects(h, jp->tile_b64); /* lives as long as the process */
    if (jp->me_type == 4) { /* firstpass.c:1273-1276 and set_gm_controls(pcs, 0), :702-703 */
        p->max_cand = 3;
        p->max_refs = 2;
        p->max_l0 = 1;
        p->gm_enabled = 0;
    }
}

int main(int argc, char **argv) {
    if (argc < 3) die("usage: oracle_me <job> <out>");
    FILE *f = fopen(argv[1], "rb");
    if (!f) die("cannot open job file");
    ref_job_header h;
    if (fread(&h, sizeof(h), 1, f) != 1 || memcmp(h.magic, REF_JOB_MAGIC, 8)) die("bad job header");
    if (h.width % 8 || h.height % 8 || h.width < 64 || h.height < 64) die("width/height must be multiples of 8 and >= 64");
    const size_t fsz = (size_t)h.width * h.height;
    uint8_t     *frames = (uint8_t *)malloc(fsz * h.n_frames);
    ref_job_pic *pics = (ref_job_pic *)calloc(h.n_pics ? h.n_pics : 1, sizeof(ref_job_pic));
    if (fread(frames, fsz, h.n_frames, f) != h.n_frames) die("short frame data");
    if (h.n_pics && fread(pics, sizeof(ref_job_pic), h.n_pics, f) != h.n_pics) die("short pic data");
    fclose(f);

    orc_planes *planes = (orc_planes *)calloc(h.n_frames, sizeof(orc_planes));
    for (uint32_t i = 0; i < h.n_frames; ++i)
        if (orc_planes_build(&planes[i], frames + fsz * i, h.width, h.width, h.height, 1)) die("out of memory");

    FILE *o = fopen(argv[2], "wb");
    if (!o) die("cannot open output file");
    ref_out_header oh;
    memset(&oh, 0, sizeof(oh));
    memcpy(oh.magic, REF_OUT_MAGIC, 8);
    oh.n_pics = h.n_pics;
    oh.n_b64 = ((h.width + 63) / 64) * ((h.height + 63) / 64);
    oh.pic_dump_size = sizeof(ref_pic_dump);
    oh.b64_dump_size = sizeof(ref_b64_dump);
    oh.mode = h.mode;
    uint64_t ops_total[6] = {0, 0, 0, 0, 0, 0};

    if (h.mode == REF_MODE_PLANES) {
        oh.n_pics = h.n_frames;
        fwrite(&oh, sizeof(oh), 1, o);
        for (uint32_t i = 0; i < h.n_frames; ++i)
            for (int k = 0; k < 3; ++k) {
                const orc_plane *p = &planes[i].lvl[k];
                uint32_t geo[4] = {p->stride, p->width, p->height, p->pad};
                fwrite(geo, sizeof(geo), 1, o);
                fwrite(p->buf, 1, (size_t)p->stride * (p->height + 2 * p->pad), o);
            }
    } else if (h.mode == REF_MODE_DUMP || h.mode == REF_MODE_DUMP_NOTAIL) {
        fwrite(&oh, sizeof(oh), 1, o);
        ref_b64_dump *rec = (ref_b64_dump *)calloc(oh.n_b64, sizeof(ref_b64_dump));
        for (uint32_t pi = 0; pi < h.n_pics; ++pi) {
            orc_pic pic;
            fill_pic(&pic, &h, &pics[pi], planes);
            uint64_t ops[6];
            orc_me_picture(&pic, rec, h.mode == REF_MODE_DUMP, (int)h.threads, ops);
            for (int i = 0; i < 6; ++i) ops_total[i] += ops[i];
            ref_pic_dump pd;
            memset(&pd, 0, sizeof(pd));
            pd.ctrl = pic.ctrl;
            pd.enable_me_8x8 = pic.enable_me_8x8;
            pd.enable_me_16x16 = pic.enable_me_16x16;
            pd.max_cand = pic.max_cand;
            pd.max_refs = pic.max_refs;
            pd.max_l0 = pic.max_l0;
            pd.input_resolution = pic.input_resolution;
            pd.n_b64 = oh.n_b64;
            fwrite(&pd, sizeof(pd), 1, o);
            fwrite(rec, sizeof(ref_b64_dump), oh.n_b64, o);
        }
        free(rec);
    } else if (h.mode == REF_MODE_TAIL) {
        if (argc < 4) die("REF_MODE_TAIL needs a handover file");
        FILE *hf = fopen(argv[3], "rb");
        if (!hf) die("cannot open handover file");
        fwrite(&oh, sizeof(oh), 1, o);
        ref_b64_dump *in = (ref_b64_dump *)calloc(oh.n_b64, sizeof(ref_b64_dump));
        ref_b64_dump *rec = (ref_b64_dump *)calloc(oh.n_b64, sizeof(ref_b64_dump));
        orc_ctx      *ctx = orc_ctx_new();
        for (uint32_t pi = 0; pi < h.n_pics; ++pi) {
            orc_pic pic;
            fill_pic(&pic, &h, &pics[pi], planes);
            if (fread(in, sizeof(ref_b64_dump), oh.n_b64, hf) != oh.n_b64) die("short handover data");
            for (uint32_t b = 0; b < oh.n_b64; ++b) orc_tail_b64(ctx, &pic, b, &in[b], &rec[b]);
            ref_pic_dump pd;
            memset(&pd, 0, sizeof(pd));
            pd.ctrl = pic.ctrl;
            pd.enable_me_8x8 = pic.enable_me_8x8;
            pd.enable_me_16x16 = pic.enable_me_16x16;
            pd.max_cand = pic.max_cand;
            pd.max_refs = pic.max_refs;
            pd.max_l0 = pic.max_l0;
            pd.input_resolution = pic.input_resolution;
            pd.n_b64 = oh.n_b64;
            fwrite(&pd, sizeof(pd), 1, o);
            fwrite(rec, sizeof(ref_b64_dump), oh.n_b64, o);
        }
        orc_ctx_free(ctx);
        fclose(hf);
        free(in);
        free(rec);
    } else {
        uint32_t reps = h.reps ? h.reps : 1;
        double   t0 = now_s();
        for (uint32_t r = 0; r < reps; ++r)
            for (uint32_t pi = 0; pi < h.n_pics; ++pi) {
                orc_pic pic;
                fill_pic(&pic, &h, &pics[pi], planes);
                uint64_t ops[6];
                orc_me_picture(&pic, NULL, 0, (int)h.threads, ops);
                for (int i = 0; i < 6; ++i) ops_total[i] += ops[i];
            }
        oh.bench_seconds = now_s() - t0;
        oh.bench_pictures = (uint64_t)reps * h.n_pics;
        fwrite(&oh, sizeof(oh), 1, o);
    }
    fclose(o);
    printf("{\"pictures\": %llu, \"seconds\": %.6f, \"threads\": %u, \"ops\": [%llu, %llu, %llu, %llu, %llu, %llu]}\n",
           (unsigned long long)(h.mode == REF_MODE_BENCH ? oh.bench_pictures : h.n_pics), oh.bench_seconds, h.threads,
           (unsigned long long)ops_total[0], (unsigned long long)ops_total[1], (unsigned long long)ops_total[2],
           (unsigned long long)ops_total[3], (unsigned long long)ops_total[4], (unsigned long long)ops_total[5]);
    for (uint32_t i = 0; i < h.n_frames; ++i) orc_planes_free(&planes[i]);
    free(planes);
    free(frames);
    free(pics);
    return 0;
}
