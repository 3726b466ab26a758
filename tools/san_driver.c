/*
 * san_driver.c - drives the C ABI through its host-side paths so that a host-sanitized build of the library
 * (tools/run_sanitized.sh: AddressSanitizer + UndefinedBehaviorSanitizer on the host code only) sees them: engines of
 * several sizes, pictures in flight on a small store, slot overwrite and release with readers in flight, every
 * me_type, the device tail in both forms, batches of pictures that share their launches, zero-copy puts from
 * page-locked buffers, batched puts and copies between engines, the allocation-free search, a group of two engines (single and
 * batched submits, with and without copies between its engines), the global-motion refinement, argument errors, and destroy with work
 * outstanding.
 * Prints "ok <checksum>"; the sanitizers abort on their own findings.
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#include "svtme.h"

static uint32_t lcg(uint32_t *s) {
    *s = *s * 1664525u + 1013904223u;
    return *s >> 8;
}

static uint64_t fnv(uint64_t h, const void *p, size_t n) {
    const uint8_t *b = (const uint8_t *)p;
    for (size_t i = 0; i < n; ++i) h = (h ^ b[i]) * 1099511628211ull;
    return h;
}

#define CHECK(call)                                                                      \
    do {                                                                                 \
        int rc_ = (call);                                                                \
        if (rc_ != SVTME_OK) {                                                           \
            fprintf(stderr, "%s:%d %s failed with status %d\n", __FILE__, __LINE__, #call, rc_); \
            exit(1);                                                                     \
        }                                                                                \
    } while (0)
#define EXPECT(call, want)                                                               \
    do {                                                                                 \
        int rc_ = (call);                                                                \
        if (rc_ != (want)) {                                                             \
            fprintf(stderr, "%s:%d %s gave %d, expected %d\n", __FILE__, __LINE__, #call, rc_, (int)(want)); \
            exit(1);                                                                     \
        }                                                                                \
    } while (0)

static uint8_t *make_luma(uint32_t h, uint32_t stride, int k, uint32_t *seed) {
    uint8_t *p = (uint8_t *)malloc((size_t)stride * h); /* exactly the bytes the ABI may read */
    for (uint32_t y = 0; y < h; ++y)
        for (uint32_t x = 0; x < stride; ++x) {
            const uint32_t sx = x + 40 + 3 * (uint32_t)k, sy = y + 40 - 2 * (uint32_t)k;
            p[(size_t)y * stride + x] = (uint8_t)((((sx >> 3) * 37 + (sy >> 3) * 91) & 0xFF) / 2 + ((sx * 5 + sy * 3) & 31) + (lcg(seed) & 3));
        }
    return p;
}

/* reads every byte a result hands out */
static uint64_t hash_results(const svtme_results *r, const svtme_picture *pic, uint64_t h) {
    const size_t n = r->n_b64, planes = (size_t)r->n_ref[0] + r->n_ref[1];
    if (r->p_sb_best_sad) h = fnv(h, r->p_sb_best_sad, planes * n * SVTME_PU_COUNT * 4);
    if (r->p_sb_best_mv) h = fnv(h, r->p_sb_best_mv, planes * n * SVTME_PU_COUNT * 4);
    h = fnv(h, r->search_results, n * 8 * sizeof(svtme_search_result));
    h = fnv(h, r->zz_sad, n * 8 * 4);
    h = fnv(h, r->reduce_me_sr_divisor, n * 8 * 4);
    if (pic->tail.enable) {
        h = fnv(h, r->total_me_candidate_index, n * SVTME_PU_COUNT);
        h = fnv(h, r->me_mv_array, n * SVTME_PU_COUNT * SVTME_TAIL_MV_STRIDE * 4);
        h = fnv(h, r->me_candidate_array, n * SVTME_PU_COUNT * SVTME_TAIL_CAND_STRIDE);
        h = fnv(h, r->me_distortion, n * SVTME_PU_COUNT * 4);
        h = fnv(h, r->b64_stats, n * 8 * 4);
        h = fnv(h, r->final_do_ref, n * 8);
        h = fnv(h, r->final_hme_sad, n * 8 * 8);
    }
    if (r->p_sb_best_sad) {
        static uint32_t            sad[2][4][85], mv[2][4][85], zz[2][4], dv[2][4];
        static svtme_search_result sr[2][4];
        for (uint32_t b = 0; b < r->n_b64; ++b) {
            CHECK(svtme_unpack_b64(r, b, &sad[0][0][0], &mv[0][0][0], &sr[0][0], &zz[0][0], &dv[0][0]));
            h = fnv(h, sad, sizeof(sad));
            h = fnv(h, mv, sizeof(mv));
        }
        EXPECT(svtme_unpack_b64(r, r->n_b64, &sad[0][0][0], &mv[0][0][0], &sr[0][0], &zz[0][0], &dv[0][0]), SVTME_ERR_BAD_ARG);
    }
    return h;
}

static void base_picture(svtme_picture *pic, uint32_t w, uint32_t h, uint64_t cur, uint64_t poc) {
    memset(pic, 0, sizeof(*pic));
    pic->cur_key = cur;
    pic->picture_number = poc;
    pic->width = w;
    pic->height = h;
}

static void add_ref(svtme_picture *pic, int list, uint64_t key, uint64_t poc) {
    const int r = pic->num_of_ref_pic_to_search[list]++;
    pic->ref_key[list][r] = key;
    pic->ref_picture_number[list][r] = poc;
    if (pic->num_of_list_to_search < list + 1) pic->num_of_list_to_search = (uint8_t)(list + 1);
}

static uint64_t run_size(uint32_t w, uint32_t h, uint32_t max_w, uint32_t max_h, int preset, uint32_t seed, uint64_t hsh) {
    enum { N = 6 };
    svtme_config cfg;
    memset(&cfg, 0, sizeof(cfg));
    cfg.max_width = max_w;
    cfg.max_height = max_h;
    cfg.store_slots = N; /* + one overwrite below: the store is full while pictures are in flight */
    cfg.max_in_flight = 3;
    cfg.kernel_timing = 1;
    svtme_engine *eng = NULL;
    CHECK(svtme_create(&cfg, &eng));
    uint8_t *luma[N];
    for (int k = 0; k < N; ++k) {
        const uint32_t stride = (k & 1) ? w + 24 : w; /* both upload paths */
        luma[k] = make_luma(h, stride, k, &seed);
        CHECK(svtme_store_put(eng, 10 + (uint64_t)k, (uint64_t)k, luma[k], stride, w, h));
    }
    EXPECT(svtme_store_put(eng, 99, 9, luma[0], w, w, h), SVTME_ERR_NO_RESOURCES); /* store full */
    EXPECT(svtme_store_put(eng, 10, 0, luma[0], w - 8, w, h), SVTME_ERR_BAD_ARG); /* stride < width */
    EXPECT(svtme_store_release(eng, 12345), SVTME_ERR_NOT_FOUND);

    svtme_preset_key key;
    memset(&key, 0, sizeof(key));
    key.enc_mode = (int8_t)preset;
    key.input_resolution = svtme_input_resolution(w, h);
    key.frame_rate = 30u << 16;
    svtme_picture pic[4];
    /* a B picture with 2 + 2 references, no tail */
    base_picture(&pic[0], w, h, 12, 2);
    pic[0].temporal_layer_index = 1;
    add_ref(&pic[0], 0, 11, 1), add_ref(&pic[0], 0, 10, 0), add_ref(&pic[0], 1, 13, 3), add_ref(&pic[0], 1, 14, 4);
    CHECK(svtme_derive_controls(&key, &pic[0].ctrl));
    /* a base-layer picture with the device tail */
    base_picture(&pic[1], w, h, 14, 4);
    pic[1].is_used_as_reference_flag = 1;
    add_ref(&pic[1], 0, 10, 0), add_ref(&pic[1], 0, 11, 1), add_ref(&pic[1], 1, 10, 0);
    CHECK(svtme_derive_controls(&key, &pic[1].ctrl));
    pic[1].tail.enable = 1;
    pic[1].tail.enable_me_8x8 = pic[1].tail.enable_me_16x16 = 1;
    pic[1].tail.max_refs = 7, pic[1].tail.max_cand = 23, pic[1].tail.max_l0 = 4;
    pic[1].tail.input_resolution = key.input_resolution;
    pic[1].tail.gm_enabled = pic[1].tail.gm_use_distance_based_active_th = 1;
    /* tail results only */
    pic[2] = pic[1];
    pic[2].cur_key = 13, pic[2].picture_number = 3, pic[2].tail.enable = 2;
    /* a P picture */
    base_picture(&pic[3], w, h, 15, 5);
    add_ref(&pic[3], 0, 14, 4);
    CHECK(svtme_derive_controls(&key, &pic[3].ctrl));

    svtme_ticket  t[4];
    svtme_results res;
    for (int i = 0; i < 3; ++i) CHECK(svtme_picture_submit(eng, &pic[i], &t[i]));
    EXPECT(svtme_picture_submit(eng, &pic[3], &t[3]), SVTME_ERR_NO_RESOURCES); /* max_in_flight */
    /* overwrite and release slots the pictures in flight still read */
    CHECK(svtme_store_put(eng, 10, 0, luma[1], w + 24, w, h));
    CHECK(svtme_store_release(eng, 11));
    for (int i = 0; i < 3; ++i) {
        float    ms[8];
        uint64_t ops[6];
        CHECK(svtme_picture_wait(eng, t[i], &res));
        hsh = hash_results(&res, &pic[i], hsh);
        CHECK(svtme_picture_timing(eng, t[i], ms));
        CHECK(svtme_picture_ops(eng, t[i], ops));
        hsh = fnv(hsh, ops, sizeof(ops));
        CHECK(svtme_picture_done(eng, t[i]));
        EXPECT(svtme_picture_done(eng, t[i]), SVTME_ERR_NOT_FOUND);
    }
    EXPECT(svtme_picture_submit(eng, &pic[0], &t[0]), SVTME_ERR_NOT_FOUND); /* reference 11 was released */
    CHECK(svtme_picture_submit(eng, &pic[3], &t[3]));
    CHECK(svtme_picture_wait(eng, t[3], &res));
    hsh = hash_results(&res, &pic[3], hsh);
    CHECK(svtme_picture_done(eng, t[3]));

    /* temporal filter pair and first-pass picture */
    svtme_picture tf;
    base_picture(&tf, w, h, 12, 2);
    tf.me_type = SVTME_ME_MCTF;
    tf.is_used_as_reference_flag = 1;
    add_ref(&tf, 0, 13, 3);
    tf.tf_me_exit_th = 256;
    for (uint8_t level = 0; level < 3; ++level) {
        CHECK(svtme_derive_tf_controls(level, &tf.ctrl));
        CHECK(svtme_picture_submit(eng, &tf, &t[0]));
        CHECK(svtme_picture_wait(eng, t[0], &res));
        hsh = hash_results(&res, &tf, hsh);
        CHECK(svtme_picture_done(eng, t[0]));
    }
    EXPECT(svtme_derive_tf_controls(3, &tf.ctrl), SVTME_ERR_BAD_ARG);
    /* the sequence's tiles for restricted temporal-filter searches: 64x64 tiles, bad shapes and bounds, clearing */
    {
        const uint32_t   bw = (w + 63) / 64, bh = (h + 63) / 64, n = bw * bh;
        svtme_tile_rect *tiles = (svtme_tile_rect *)calloc(n, sizeof(*tiles));
        for (uint32_t i = 0; i < n; ++i) {
            tiles[i].start_x = (i % bw) * 64, tiles[i].start_y = (i / bw) * 64;
            tiles[i].end_x = tiles[i].start_x + 64 < w ? tiles[i].start_x + 64 : w;
            tiles[i].end_y = tiles[i].start_y + 64 < h ? tiles[i].start_y + 64 : h;
        }
        EXPECT(svtme_set_tile_rects(eng, w, h, tiles, n - 1), SVTME_ERR_BAD_ARG);
        tiles[0].end_x = w + 8;
        EXPECT(svtme_set_tile_rects(eng, w, h, tiles, n), SVTME_ERR_BAD_ARG);
        tiles[0].end_x = 64;
        CHECK(svtme_set_tile_rects(eng, w, h, tiles, n));
        CHECK(svtme_derive_tf_controls(0, &tf.ctrl));
        for (uint8_t restricted = 0; restricted < 2; ++restricted) {
            tf.restricted_motion_vector = restricted;
            CHECK(svtme_picture_submit(eng, &tf, &t[0]));
            CHECK(svtme_picture_wait(eng, t[0], &res));
            hsh = hash_results(&res, &tf, hsh);
            CHECK(svtme_picture_done(eng, t[0]));
        }
        CHECK(svtme_set_tile_rects(eng, w, h, NULL, 0));
        CHECK(svtme_picture_submit(eng, &tf, &t[0])); /* restricted, the picture as the one tile */
        CHECK(svtme_picture_wait(eng, t[0], &res));
        hsh = hash_results(&res, &tf, hsh);
        CHECK(svtme_picture_done(eng, t[0]));
        tf.restricted_motion_vector = 0;
        free(tiles);
    }
    svtme_picture fp;
    base_picture(&fp, w, h, 15, 5);
    fp.me_type = SVTME_ME_FIRST_PASS;
    fp.is_used_as_reference_flag = 1;
    fp.bypass_zz_check = 1;
    add_ref(&fp, 0, 14, 4), add_ref(&fp, 0, 10, 0);
    CHECK(svtme_derive_first_pass_controls(key.input_resolution, key.frame_rate, 0, &fp.ctrl));
    fp.tail.enable = 1;
    fp.tail.enable_me_8x8 = fp.tail.enable_me_16x16 = 1;
    fp.tail.max_refs = 2, fp.tail.max_cand = 3, fp.tail.max_l0 = 1;
    CHECK(svtme_picture_submit(eng, &fp, &t[0]));
    CHECK(svtme_picture_wait(eng, t[0], &res));
    hsh = hash_results(&res, &fp, hsh);
    CHECK(svtme_picture_done(eng, t[0]));
    /* the first pass's block step: only every second block of every second block row; the others are not looked at */
    fp.b64_step = 2;
    CHECK(svtme_picture_submit(eng, &fp, &t[0]));
    CHECK(svtme_picture_wait(eng, t[0], &res));
    {
        const uint32_t bw = (w + 63) / 64;
        for (uint32_t b = 0; b < res.n_b64; ++b)
            if ((b % bw) % 2 == 0 && (b / bw) % 2 == 0) hsh = fnv(hsh, &res.search_results[b * 8], 8 * sizeof(svtme_search_result));
    }
    CHECK(svtme_picture_done(eng, t[0]));
    fp.b64_step = 9;
    EXPECT(svtme_picture_submit(eng, &fp, &t[0]), SVTME_ERR_BAD_ARG);
    fp.b64_step = 0;

    /* read back a plane, then leave with a picture still in flight */
    {
        uint32_t stride = 0, pw = 0, ph = 0, pad = 0;
        EXPECT(svtme_store_read(eng, 14, 2, NULL, 0, &stride, &pw, &ph, &pad), SVTME_ERR_BAD_ARG);
        const size_t bytes = (size_t)(w + 2 * 72) * (h + 2 * 72);
        uint8_t     *dst = (uint8_t *)malloc(bytes);
        CHECK(svtme_store_read(eng, 14, 2, dst, bytes, &stride, &pw, &ph, &pad));
        hsh = fnv(hsh, dst, (size_t)stride * (ph + 2 * pad));
        EXPECT(svtme_store_read(eng, 14, 2, dst, bytes - 1, &stride, &pw, &ph, &pad), SVTME_ERR_BAD_ARG);
        free(dst);
    }
    CHECK(svtme_picture_submit(eng, &pic[3], &t[3]));
    svtme_destroy(eng);
    for (int k = 0; k < N; ++k) free(luma[k]);
    return hsh;
}

/* batches, page-locked puts, the allocation-free search and the engine group */
static uint64_t run_extras(uint32_t w, uint32_t h, uint32_t seed, uint64_t hsh) {
    enum { N = 5 };
    const uint32_t stride = w + 144; /* the encoder's padded input buffers */
    svtme_config   cfg;
    memset(&cfg, 0, sizeof(cfg));
    cfg.max_width = w, cfg.max_height = h, cfg.store_slots = N + 1, cfg.max_in_flight = 4;
    svtme_engine *eng = NULL;
    CHECK(svtme_create(&cfg, &eng));
    /* page-locked source buffers; a pageable one goes the same way (the runtime stages it) */
    uint8_t *pool = NULL, *plain = (uint8_t *)malloc((size_t)stride * h);
    CHECK(svtme_host_alloc(eng, (size_t)N * stride * h, (void **)&pool));
    EXPECT(svtme_host_alloc(eng, 0, (void **)&pool), SVTME_ERR_BAD_ARG);
    for (int k = 0; k < N; ++k) {
        uint8_t *src = make_luma(h, stride, k, &seed);
        memcpy(pool + (size_t)k * stride * h, src, (size_t)stride * h);
        if (k == 1) memcpy(plain, src, (size_t)stride * h);
        free(src);
    }
    for (int k = 0; k < N; ++k)
        CHECK(svtme_store_put_pinned(eng, 100 + (uint64_t)k, (uint64_t)k, k == 1 ? plain : pool + (size_t)k * stride * h, stride, w, h));
    CHECK(svtme_store_wait(eng, 101));
    EXPECT(svtme_store_wait(eng, 999), SVTME_ERR_NOT_FOUND);
    {
        const int rc = svtme_host_register(eng, plain, (size_t)stride * h); /* works, or says this machine cannot lock user pages */
        if (rc == SVTME_OK) CHECK(svtme_host_unregister(eng, plain));
        else if (rc != SVTME_ERR_UNSUPPORTED) EXPECT(rc, SVTME_OK);
    }
    /* three pictures with different reference lists as one batch; the same one by one must give the same bytes */
    svtme_preset_key key = {8, 0, svtme_input_resolution(w, h), 0, 30u << 16};
    svtme_picture    pic[3];
    base_picture(&pic[0], w, h, 102, 2);
    add_ref(&pic[0], 0, 100, 0), add_ref(&pic[0], 0, 101, 1), add_ref(&pic[0], 1, 104, 4);
    base_picture(&pic[1], w, h, 101, 1);
    add_ref(&pic[1], 0, 100, 0), add_ref(&pic[1], 1, 102, 2);
    pic[1].temporal_layer_index = 2;
    base_picture(&pic[2], w, h, 103, 3);
    add_ref(&pic[2], 0, 102, 2);
    for (int i = 0; i < 3; ++i) CHECK(svtme_derive_controls(&key, &pic[i].ctrl));
    svtme_ticket  t[3];
    svtme_results res;
    uint64_t      one = 0, batch = 0;
    for (int i = 0; i < 3; ++i) {
        CHECK(svtme_picture_submit(eng, &pic[i], &t[0]));
        CHECK(svtme_picture_wait(eng, t[0], &res));
        one = hash_results(&res, &pic[i], one);
        CHECK(svtme_picture_done(eng, t[0]));
    }
    CHECK(svtme_pictures_submit(eng, pic, 3, t));
    for (int i = 0; i < 3; ++i) {
        float ms[8];
        CHECK(svtme_picture_wait(eng, t[i], &res));
        CHECK(svtme_picture_timing(eng, t[i], ms));
        if (ms[7] != 3.f) EXPECT(-100, SVTME_OK);
        batch = hash_results(&res, &pic[i], batch);
    }
    for (int i = 2; i >= 0; --i) CHECK(svtme_picture_done(eng, t[i]));
    if (one != batch) {
        fprintf(stderr, "a batch and its pictures one by one differ\n");
        exit(1);
    }
    hsh = fnv(hsh, &batch, sizeof(batch));
    EXPECT(svtme_pictures_submit(eng, pic, 0, t), SVTME_ERR_BAD_ARG);
    EXPECT(svtme_pictures_submit(eng, pic, SVTME_MAX_BATCH + 1, t), SVTME_ERR_BAD_ARG);
    {
        svtme_picture two[2] = {pic[0], pic[1]};
        two[1].ctrl.me_sa.sa_max.height++; /* other controls: not one batch */
        EXPECT(svtme_pictures_submit(eng, two, 2, t), SVTME_ERR_BAD_ARG);
        svtme_picture five[5] = {pic[0], pic[1], pic[2], pic[0], pic[1]}; /* more than the engine keeps in flight */
        EXPECT(svtme_pictures_submit(eng, five, 5, t), SVTME_ERR_NO_RESOURCES);
    }
    /* the per-kernel events switched on for a submission (this engine starts without them) and off again */
    for (int on = 1; on >= 0; --on) {
        float ms[8];
        EXPECT(svtme_kernel_timing(eng, 2), SVTME_ERR_BAD_ARG);
        CHECK(svtme_kernel_timing(eng, on));
        CHECK(svtme_pictures_submit(eng, pic, 3, t));
        for (int i = 0; i < 3; ++i) {
            CHECK(svtme_picture_wait(eng, t[i], &res));
            CHECK(svtme_picture_timing(eng, t[i], ms));
            if ((on ? ms[0] <= 0.f || ms[3] <= 0.f : ms[0] != 0.f || ms[3] != 0.f) || ms[5] <= 0.f) EXPECT(-102 - on, SVTME_OK);
            CHECK(svtme_picture_done(eng, t[i]));
        }
    }
    /* the same picture through the entry point that allocates nothing, results in page-locked memory of ours */
    {
        const size_t n_b64 = (size_t)((w + 63) / 64) * ((h + 63) / 64), bytes = SVTME_STATIC_RESULT_BYTES(n_b64, 3);
        uint8_t     *block = NULL;
        CHECK(svtme_host_alloc(eng, bytes, (void **)&block));
        svtme_results st;
        CHECK(svtme_picture_search_static(eng, &pic[0], block, bytes, &st));
        uint64_t a = hash_results(&st, &pic[0], 0), b = 0;
        CHECK(svtme_picture_submit(eng, &pic[0], &t[0]));
        CHECK(svtme_picture_wait(eng, t[0], &res));
        b = hash_results(&res, &pic[0], 0);
        CHECK(svtme_picture_done(eng, t[0]));
        if (a != b) {
            fprintf(stderr, "svtme_picture_search_static and submit / wait differ\n");
            exit(1);
        }
        EXPECT(svtme_picture_search_static(eng, &pic[0], block, bytes - 1, &st), SVTME_ERR_BAD_ARG);
        EXPECT(svtme_picture_search_static(eng, &pic[0], block + 8, bytes, &st), SVTME_ERR_BAD_ARG);
        CHECK(svtme_host_free(eng, block));
        hsh = fnv(hsh, &a, sizeof(a));
    }
    /* global-motion calls on the stored pictures. The warp filter and the error table are the caller's to give: this walk
       uses stand-ins of the right shape (an identity filter: tap 3 = 128, and |error| as the measure), with which the warp
       error of the identity model is the frame error */
    {
        static int32_t pts[2 * SVTME_GM_MAX_CORNERS], corr[4 * SVTME_GM_MAX_CORNERS];
        int32_t        n_pts = 0, n_corr = 0, n_few = 0;
        int64_t        e_frame = 0, e_warp = 0, e_stop = 0;
        EXPECT(svtme_gm_frame_error(eng, 102, 100, 0, &e_frame), SVTME_ERR_BAD_ARG); /* no tables yet */
        int16_t filter[193 * 8];
        int32_t lut[512];
        for (int i = 0; i < 193; ++i)
            for (int k = 0; k < 8; ++k) filter[i * 8 + k] = k == 3 ? 128 : 0;
        for (int i = 0; i < 512; ++i) lut[i] = i < 255 ? 255 - i : i - 255;
        EXPECT(svtme_gm_set_tables(eng, NULL, lut), SVTME_ERR_BAD_ARG);
        CHECK(svtme_gm_set_tables(eng, filter, lut));
        for (int level = 0; level < 3; ++level) {
            CHECK(svtme_gm_corners(eng, 102, level, pts, SVTME_GM_MAX_CORNERS, &n_pts));
            CHECK(svtme_gm_corners(eng, 102, level, pts, 5, &n_few));
            if (n_few != (n_pts < 5 ? n_pts : 5)) EXPECT(-110, SVTME_OK);
            CHECK(svtme_gm_correspondences(eng, 102, 100, level, corr, &n_corr));
            if (n_corr > n_pts) EXPECT(-111, SVTME_OK);
            hsh = fnv(hsh, &n_pts, sizeof(n_pts));
            hsh = fnv(hsh, corr, sizeof(int32_t) * 4 * (size_t)n_corr);
            svtme_gm_warp wm = {{0, 0, 1 << 16, 0, 0, 1 << 16}, 0, 0, 0, 0};
            CHECK(svtme_gm_frame_error(eng, 102, 100, level, &e_frame));
            CHECK(svtme_gm_warp_error(eng, 102, 100, level, &wm, INT64_MAX, &e_warp));
            CHECK(svtme_gm_warp_error(eng, 102, 100, level, &wm, 0, &e_stop)); /* stops after the first 32x32 block */
            if (e_warp != e_frame || e_stop > e_warp || e_frame <= 0) EXPECT(-112, SVTME_OK);
            hsh = fnv(hsh, &e_frame, sizeof(e_frame));
            hsh = fnv(hsh, &e_stop, sizeof(e_stop));
        }
        /* the refinement loop as one call: with the stand-in tables any model comes back valid, and a second run from the
           refined model cannot find a smaller error than the first one returned */
        {
            svtme_gm_model m = {{3 << 13, -(2 << 13), (1 << 16) + 40, 12, -12, (1 << 16) + 40}, 0, 0, 0, 0, 2};
            int64_t        e1 = 0, e2 = 0;
            uint32_t       n1 = 0, n2 = 0;
            CHECK(svtme_gm_refine(eng, 102, 100, 1, &m, 2, 3, INT64_MAX, &e1, &n1));
            if (n1 < 1 + 2 * 4 * 3 || m.wmtype < 0 || m.wmtype > 3 || e1 <= 0) EXPECT(-113, SVTME_OK);
            CHECK(svtme_gm_refine(eng, 102, 100, 1, &m, m.wmtype, 3, e1, &e2, &n2));
            if (e2 > e1) EXPECT(-114, SVTME_OK);
            CHECK(svtme_gm_refine(eng, 102, 100, 0, &m, 0, 1, INT64_MAX, &e2, NULL)); /* identity forced: one evaluation */
            if (m.wmtype != 0 || m.wmmat[0] || m.wmmat[2] != 1 << 16) EXPECT(-115, SVTME_OK);
            hsh = fnv(hsh, &e1, sizeof(e1));
            hsh = fnv(hsh, &n1, sizeof(n1));
            EXPECT(svtme_gm_refine(eng, 102, 100, 1, &m, 4, 3, INT64_MAX, &e1, NULL), SVTME_ERR_BAD_ARG);
            EXPECT(svtme_gm_refine(eng, 102, 100, 1, &m, 2, 0, INT64_MAX, &e1, NULL), SVTME_ERR_BAD_ARG);
            EXPECT(svtme_gm_refine(eng, 102, 999, 1, &m, 2, 3, INT64_MAX, &e1, NULL), SVTME_ERR_NOT_FOUND);
            EXPECT(svtme_gm_refine(eng, 102, 100, 1, NULL, 2, 3, INT64_MAX, &e1, NULL), SVTME_ERR_BAD_ARG);
        }
        EXPECT(svtme_gm_corners(eng, 999, 0, pts, 8, &n_pts), SVTME_ERR_NOT_FOUND);
        EXPECT(svtme_gm_corners(eng, 102, 3, pts, 8, &n_pts), SVTME_ERR_BAD_ARG);
        EXPECT(svtme_gm_warp_error(eng, 102, 100, 0, NULL, 0, &e_warp), SVTME_ERR_BAD_ARG);
    }
    /* the TPL dispenser's source-based search on made-up ME results: one single-reference candidate per list and block
       with vectors that reach the clamp, every level, with and without intra prediction; its refusals */
    {
        const uint32_t n_b64 = ((w + 63) / 64) * ((h + 63) / 64), n_pu = SVTME_PU_COUNT, max_cand = 3, max_refs = 2;
        uint8_t  *cnt  = (uint8_t *)calloc((size_t)n_b64 * n_pu, 1);
        uint8_t  *cand = (uint8_t *)calloc((size_t)n_b64 * n_pu * max_cand, 1);
        uint32_t *mv   = (uint32_t *)calloc((size_t)n_b64 * n_pu * max_refs, 4);
        svtme_tpl_block *blk = (svtme_tpl_block *)calloc((size_t)n_b64 * SVTME_TPL_BLOCKS, sizeof(*blk));
        for (uint32_t i = 0; i < n_b64 * n_pu; ++i) {
            cnt[i] = 3;
            cand[i * max_cand + 0] = 0; /* list 0, reference 0 */
            cand[i * max_cand + 1] = 1 | (1 << 7); /* list 1, reference 0 */
            cand[i * max_cand + 2] = 2 | (1 << 7); /* bi-directional: not looked at */
            mv[i * max_refs + 0] = (uint32_t)(uint16_t)(int16_t)(((int)(i * 37 % 801)) - 400) | ((uint32_t)(uint16_t)(int16_t)(((int)(i * 53 % 601)) - 300) << 16);
            mv[i * max_refs + 1] = (uint32_t)(uint16_t)(int16_t)(-(int)(i % 23)) | ((uint32_t)(uint16_t)(int16_t)(i % 17) << 16);
        }
        svtme_tpl_picture tp;
        memset(&tp, 0, sizeof(tp));
        tp.cur_key = 102; tp.ref_key[0][0] = 100; tp.ref_key[1][0] = 101;
        tp.n_ref[0] = tp.n_ref[1] = 1;
        tp.width = w; tp.height = h;
        tp.tpl_opt_flag = tp.use_pred_sad_in_intra_search = tp.use_pred_sad_in_inter_search = 1;
        tp.enable_me_16x16 = 1;
        tp.n_pu = (uint8_t)n_pu; tp.max_cand = (uint8_t)max_cand; tp.max_refs = (uint8_t)max_refs; tp.max_l0 = 1;
        tp.total_me_candidate_index = cnt; tp.me_candidate_array = cand; tp.me_mv_array = mv;
        for (int level = 0; level < 3; ++level)
            for (int no_intra = 0; no_intra < 2; ++no_intra) {
                tp.dispenser_search_level = (uint8_t)level;
                tp.disable_intra_pred     = (uint8_t)no_intra;
                CHECK(svtme_tpl_search(eng, &tp, blk));
                uint32_t n_valid = 0;
                for (uint32_t i = 0; i < n_b64 * SVTME_TPL_BLOCKS; ++i) {
                    if (!blk[i].valid) continue;
                    ++n_valid;
                    if (blk[i].best_rf_idx != 0 && blk[i].best_rf_idx != 4) EXPECT(-120, SVTME_OK);
                    if (no_intra && !blk[i].best_is_inter) EXPECT(-121, SVTME_OK);
                }
                if (!n_valid) EXPECT(-122, SVTME_OK);
                hsh = fnv(hsh, blk, sizeof(*blk) * (size_t)n_b64 * SVTME_TPL_BLOCKS);
            }
        /* transform-domain costs and the quantisation error of the chosen candidate: whole 16x16 transforms only */
        tp.dispenser_search_level = 0;
        tp.disable_intra_pred     = 0;
        tp.src_dist               = 1;
        tp.y_round_fp[0] = 20; tp.y_round_fp[1] = 24; tp.y_quant_fp[0] = 1638; tp.y_quant_fp[1] = 1365; tp.y_dequant[0] = 40; tp.y_dequant[1] = 48;
        for (int shape = 0; shape < 3; ++shape)
            for (int sad = 0; sad < 2; ++sad) {
                tp.pf_shape = (uint8_t)shape;
                tp.use_pred_sad_in_intra_search = tp.use_pred_sad_in_inter_search = (uint8_t)sad;
                CHECK(svtme_tpl_search(eng, &tp, blk));
                for (uint32_t i = 0; i < n_b64 * SVTME_TPL_BLOCKS; ++i)
                    if (blk[i].valid && (blk[i].src_recon_error > 0) != (blk[i].best_is_inter != 0)) EXPECT(-123, SVTME_OK);
                hsh = fnv(hsh, blk, sizeof(*blk) * (size_t)n_b64 * SVTME_TPL_BLOCKS);
            }
        /* the reconstruction path on those answers: every transform the tpl levels use, with and without the picture coming
           back; the reconstructed picture put back into the store as a reference of the next call */
        {
            svtme_tpl_stats *st  = (svtme_tpl_stats *)calloc((size_t)n_b64 * SVTME_TPL_BLOCKS, sizeof(*st));
            uint8_t         *rec = (uint8_t *)malloc((size_t)w * h);
            tp.pf_shape = 1;
            tp.use_pred_sad_in_intra_search = tp.use_pred_sad_in_inter_search = 1;
            for (int form = 0; form < 4; ++form) { /* 16x16, 16x8, 16x4 on 16x16 blocks; 32x8 on 32x32 blocks */
                tp.dispenser_search_level = form == 3;
                tp.subsample_tx           = (uint8_t)(form == 3 ? 2 : form);
                CHECK(svtme_tpl_search(eng, &tp, blk));
                CHECK(svtme_tpl_recon(eng, &tp, blk, st, form & 1 ? rec : NULL, w));
                for (uint32_t i = 0; i < n_b64 * SVTME_TPL_BLOCKS; ++i)
                    if ((st[i].recrf_dist > 0) != (blk[i].valid != 0) || st[i].recrf_dist < st[i].srcrf_dist) EXPECT(-126, SVTME_OK);
                hsh = fnv(hsh, st, sizeof(*st) * (size_t)n_b64 * SVTME_TPL_BLOCKS);
                if (form & 1) hsh = fnv(hsh, rec, (size_t)w * h);
            }
            CHECK(svtme_tpl_recon_keep(eng, 150, 7)); /* device to device */
            tp.ref_key[0][0] = 150;
            CHECK(svtme_tpl_recon(eng, &tp, blk, st, NULL, 0));
            hsh = fnv(hsh, st, sizeof(*st) * (size_t)n_b64 * SVTME_TPL_BLOCKS);
            tp.ref_key[0][0] = 100;
            CHECK(svtme_store_release(eng, 150));
            EXPECT(svtme_tpl_recon_keep(NULL, 150, 7), SVTME_ERR_BAD_ARG);
            EXPECT(svtme_tpl_recon(eng, &tp, blk, NULL, NULL, 0), SVTME_ERR_BAD_ARG);
            EXPECT(svtme_tpl_recon(eng, &tp, blk, st, rec, w - 8), SVTME_ERR_BAD_ARG);
            tp.subsample_tx = 1; /* 32x32 blocks come with the 32x8 transform only */
            EXPECT(svtme_tpl_recon(eng, &tp, blk, st, NULL, 0), SVTME_ERR_UNSUPPORTED);
            tp.dispenser_search_level = 0;
            tp.subsample_tx = 0;
            tp.pf_shape     = 0;
            blk[5].valid = 1, blk[5].best_is_inter = 1, blk[5].best_rf_idx = 6; /* a reference the picture does not have */
            EXPECT(svtme_tpl_recon(eng, &tp, blk, st, NULL, 0), SVTME_ERR_BAD_ARG);
            free(st); free(rec);
        }
        tp.dispenser_search_level = 1; /* 32x32 blocks have no 16x16 transform */
        tp.use_pred_sad_in_intra_search = tp.use_pred_sad_in_inter_search = 0;
        EXPECT(svtme_tpl_search(eng, &tp, blk), SVTME_ERR_UNSUPPORTED);
        tp.tpl_opt_flag = 0; /* every intra mode: 16x16 blocks only */
        tp.intra_mode_end = 12;
        EXPECT(svtme_tpl_search(eng, &tp, blk), SVTME_ERR_UNSUPPORTED);
        tp.dispenser_search_level = 0;
        tp.intra_mode_end = 5; /* not one of the loop's three ends */
        EXPECT(svtme_tpl_search(eng, &tp, blk), SVTME_ERR_BAD_ARG);
        for (int end = 0; end < 3; ++end) { /* PAETH_PRED, SMOOTH_H_PRED, D67_PRED */
            tp.intra_mode_end = (uint8_t)(end == 0 ? 12 : (end == 1 ? 11 : 8));
            CHECK(svtme_tpl_search(eng, &tp, blk));
            for (uint32_t i = 0; i < n_b64 * SVTME_TPL_BLOCKS; ++i) {
                if (!blk[i].valid) continue;
                if (blk[i].best_intra_mode > tp.intra_mode_end) EXPECT(-124, SVTME_OK);
                if ((blk[i].src_rate > 0) != (blk[i].best_is_inter != 0)) EXPECT(-125, SVTME_OK);
            }
            hsh = fnv(hsh, blk, sizeof(*blk) * (size_t)n_b64 * SVTME_TPL_BLOCKS);
        }
        tp.tpl_opt_flag = 1;
        tp.n_ref[1] = 0; /* the list-1 candidates now name a reference that was not handed over */
        EXPECT(svtme_tpl_search(eng, &tp, blk), SVTME_ERR_BAD_ARG);
        tp.n_ref[1] = 1;
        tp.ref_key[1][0] = 999;
        EXPECT(svtme_tpl_search(eng, &tp, blk), SVTME_ERR_NOT_FOUND);
        EXPECT(svtme_tpl_search(eng, &tp, NULL), SVTME_ERR_BAD_ARG);
        free(cnt); free(cand); free(mv); free(blk);
    }
    /* the same sources once more as ONE batched put under other keys, and from there device to device into a second engine:
       the three pictures searched on either give the bytes they gave above */
    {
        uint64_t       keys[N], pocs[N];
        const uint8_t *lumas[N];
        for (int k = 0; k < N; ++k) keys[k] = 200 + (uint64_t)k, pocs[k] = (uint64_t)k, lumas[k] = pool + (size_t)k * stride * h;
        svtme_config cfg2 = cfg;
        cfg2.store_slots  = 2 * N + 1;
        svtme_engine *big = NULL, *peer = NULL;
        CHECK(svtme_create(&cfg2, &big));
        CHECK(svtme_create(&cfg, &peer));
        CHECK(svtme_store_put_many(big, N, keys, pocs, lumas, stride, w, h, 1));
        EXPECT(svtme_store_put_many(big, 0, keys, pocs, lumas, stride, w, h, 1), SVTME_ERR_BAD_ARG);
        EXPECT(svtme_store_put_many(big, SVTME_MAX_PUT_BATCH + 1, keys, pocs, lumas, stride, w, h, 1), SVTME_ERR_BAD_ARG);
        EXPECT(svtme_store_put_many(big, N, keys, pocs, NULL, stride, w, h, 1), SVTME_ERR_BAD_ARG);
        for (int k = 0; k < N; ++k) CHECK(svtme_store_copy_peer(peer, big, keys[k]));
        EXPECT(svtme_store_copy_peer(peer, big, 999), SVTME_ERR_NOT_FOUND);
        EXPECT(svtme_store_copy_peer(peer, NULL, keys[0]), SVTME_ERR_BAD_ARG);
        svtme_engine *both[2] = {big, peer};
        for (int e = 0; e < 2; ++e) {
            uint64_t again = 0;
            for (int i = 0; i < 3; ++i) {
                svtme_picture q = pic[i];
                q.cur_key += 100;
                for (int l = 0; l < SVTME_MAX_LISTS; ++l)
                    for (int r = 0; r < SVTME_MAX_REFS; ++r)
                        if (q.ref_key[l][r]) q.ref_key[l][r] += 100;
                CHECK(svtme_picture_submit(both[e], &q, &t[0]));
                CHECK(svtme_picture_wait(both[e], t[0], &res));
                again = hash_results(&res, &pic[i], again);
                CHECK(svtme_picture_done(both[e], t[0]));
            }
            if (again != one) {
                fprintf(stderr, "a batched put (engine %d) and single puts differ\n", e);
                exit(1);
            }
        }
        svtme_destroy(peer);
        svtme_destroy(big);
    }
    svtme_destroy(eng);
    /* two engines on this device behind one handle: pictures round robin, sources where they are read */
    {
        const int32_t devices[2] = {0, 0};
        svtme_group  *grp = NULL;
        EXPECT(svtme_group_create(devices, 0, &cfg, &grp), SVTME_ERR_BAD_ARG);
        CHECK(svtme_group_create(devices, 2, &cfg, &grp));
        if (svtme_group_size(grp) != 2) EXPECT(-101, SVTME_OK);
        for (int k = 0; k < N; ++k)
            CHECK(svtme_group_store_put(grp, 100 + (uint64_t)k, (uint64_t)k, pool + (size_t)k * stride * h, stride, w, h, k != 1));
        uint64_t g = 0;
        svtme_ticket gt[3];
        for (int i = 0; i < 3; ++i) CHECK(svtme_group_picture_submit(grp, &pic[i], &gt[i]));
        for (int i = 0; i < 3; ++i) {
            CHECK(svtme_group_picture_wait(grp, gt[i], &res));
            g = hash_results(&res, &pic[i], g);
            CHECK(svtme_group_picture_done(grp, gt[i]));
        }
        /* (buffer 1 of the pool holds the same pixels as `plain`, which the single engine searched) */
        if (g != one) {
            fprintf(stderr, "the group and a single engine differ\n");
            exit(1);
        }
        uint64_t np = 0, ns = 0, nu = 0;
        CHECK(svtme_group_stats(grp, &np, &ns, &nu));
        if (np != 3 || ns != N || nu < 4 || nu > 2 * N) EXPECT(-102, SVTME_OK);
        /* the three pictures as one submit (the group splits them over its engines), with copies between the engines
           and, after the sources have been announced afresh, without */
        for (int round = 0; round < 2; ++round) {
            uint64_t copies = 0, links = 0;
            CHECK(svtme_group_set_peer_copies(grp, round == 0));
            if (round)
                for (int k = 0; k < N; ++k)
                    CHECK(svtme_group_store_put(grp, 100 + (uint64_t)k, (uint64_t)k, pool + (size_t)k * stride * h, stride, w, h, k != 1));
            CHECK(svtme_group_pictures_submit(grp, pic, 3, gt));
            g = 0;
            for (int i = 0; i < 3; ++i) {
                if (svtme_group_ticket_engine(gt[i]) > 1) EXPECT(-103, SVTME_OK);
                CHECK(svtme_group_picture_wait(grp, gt[i], &res));
                g = hash_results(&res, &pic[i], g);
                CHECK(svtme_group_picture_done(grp, gt[i]));
            }
            if (g != one) {
                fprintf(stderr, "the group's batched submit (round %d) and a single engine differ\n", round);
                exit(1);
            }
            CHECK(svtme_group_peer_stats(grp, &copies, &links));
        }
        EXPECT(svtme_group_pictures_submit(grp, pic, 0, gt), SVTME_ERR_BAD_ARG);
        EXPECT(svtme_group_picture_wait(grp, (svtme_ticket)7 << 56, &res), SVTME_ERR_NOT_FOUND);
        CHECK(svtme_group_store_release(grp, 100));
        EXPECT(svtme_group_store_release(grp, 100), SVTME_ERR_NOT_FOUND);
        EXPECT(svtme_group_picture_submit(grp, &pic[0], &gt[0]), SVTME_ERR_NOT_FOUND); /* its reference is gone */
        svtme_group_destroy(grp);
    }
    /* the pool outlives the engines that read from it */
    {
        svtme_engine *e2 = NULL;
        CHECK(svtme_create(&cfg, &e2));
        CHECK(svtme_host_free(e2, pool));
        svtme_destroy(e2);
    }
    free(plain);
    return hsh;
}

int main(void) {
    uint64_t h = 1469598103934665603ull;
    EXPECT(svtme_create(NULL, NULL), SVTME_ERR_BAD_ARG);
    h = run_extras(320, 192, 5, h);
    h = run_size(192, 136, 192, 136, 8, 1, h);
    h = run_size(128, 64, 256, 192, 3, 2, h); /* smaller pictures than the engine was sized for */
    h = run_size(328, 200, 328, 200, 12, 3, h);
    h = run_size(64, 64, 64, 64, 0, 4, h);
    printf("ok %016llx\n", (unsigned long long)h);
    fflush(stdout);
    /* leave without running the HIP runtime's exit handlers: under the sanitizer's allocator they fail a check of the
       sanitizer runtime itself (memory freed after its device hooks were torn down), which is not the library's doing */
    _exit(0);
}
