/*
 * me_shim_demo.c - the engine driven from plain C, the way the encoder's ME process would drive it
 * (Source/Lib/Encoder/Codec/EbMotionEstimationProcess.c:698-846): put the source pictures into the device store,
 * submit one picture's ME job, wait, and unpack every 64x64 block into arrays shaped like MeContext's.
 *
 * Usage: me_shim_demo <width> <height> <preset> <seed>
 * Pictures are generated from the seed by a 32-bit LCG (a smooth field shifted by (3, -2) per picture plus noise), so a
 * test can rebuild the same input. Prints one line: the number of blocks and FNV-1a hashes of everything unpacked.
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

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

#define CHECK(call)                                                        \
    do {                                                                   \
        int rc_ = (call);                                                  \
        if (rc_ != SVTME_OK) {                                             \
            fprintf(stderr, "%s failed with status %d\n", #call, rc_);     \
            return 1;                                                      \
        }                                                                  \
    } while (0)

int main(int argc, char **argv) {
    if (argc < 5) {
        fprintf(stderr, "usage: me_shim_demo <width> <height> <preset> <seed>\n");
        return 2;
    }
    const uint32_t w = (uint32_t)atoi(argv[1]), h = (uint32_t)atoi(argv[2]);
    const int      preset = atoi(argv[3]);
    uint32_t       seed = (uint32_t)atoi(argv[4]);
    enum { N_PIC = 3 };

    /* a coarse random field, bilinear-free: 16x16 cells, wider than the picture so that it can be shifted */
    const uint32_t cw = w / 16 + 8, ch = h / 16 + 8;
    uint8_t       *cells = (uint8_t *)malloc((size_t)cw * ch);
    for (uint32_t i = 0; i < cw * ch; ++i) cells[i] = (uint8_t)(lcg(&seed) & 0xFF);
    uint8_t *luma[N_PIC];
    for (int k = 0; k < N_PIC; ++k) {
        luma[k] = (uint8_t *)malloc((size_t)w * h);
        for (uint32_t y = 0; y < h; ++y)
            for (uint32_t x = 0; x < w; ++x) {
                const uint32_t sx = x + 32 + 3 * (uint32_t)k, sy = y + 32 - 2 * (uint32_t)k;
                const uint32_t base = cells[(sy / 16) * cw + sx / 16];
                const uint32_t fine = ((sx * 7 + sy * 13) ^ (sx * sy)) & 15;
                luma[k][(size_t)y * w + x] = (uint8_t)((base * 3 + fine * 4 + (lcg(&seed) & 3)) >> 2);
            }
    }

    svtme_config cfg;
    memset(&cfg, 0, sizeof(cfg));
    cfg.device = 0;
    cfg.max_width = w;
    cfg.max_height = h;
    cfg.store_slots = N_PIC + 1;
    cfg.max_in_flight = 2;
    svtme_engine *eng = NULL;
    CHECK(svtme_create(&cfg, &eng));
    for (int k = 0; k < N_PIC; ++k) CHECK(svtme_store_put(eng, 100 + (uint64_t)k, (uint64_t)k, luma[k], w, w, h));

    /* picture 1 is a B picture between 0 and 2 */
    svtme_picture pic;
    memset(&pic, 0, sizeof(pic));
    pic.cur_key = 101;
    pic.picture_number = 1;
    pic.width = w;
    pic.height = h;
    pic.temporal_layer_index = 1;
    pic.is_used_as_reference_flag = 0;
    pic.num_of_list_to_search = 2;
    pic.num_of_ref_pic_to_search[0] = 1;
    pic.num_of_ref_pic_to_search[1] = 1;
    pic.ref_key[0][0] = 100;
    pic.ref_picture_number[0][0] = 0;
    pic.ref_key[1][0] = 102;
    pic.ref_picture_number[1][0] = 2;
    svtme_preset_key key;
    memset(&key, 0, sizeof(key));
    key.enc_mode = (int8_t)preset;
    key.input_resolution = svtme_input_resolution(w, h);
    key.frame_rate = 30u << 16;
    CHECK(svtme_derive_controls(&key, &pic.ctrl));

    svtme_ticket  ticket = 0;
    svtme_results res;
    CHECK(svtme_picture_submit(eng, &pic, &ticket));
    CHECK(svtme_picture_wait(eng, ticket, &res));

    /* what the b64 loop of motion_estimation_kernel() would do instead of hme_b64 + integer_search_b64 */
    static uint32_t            best_sad[SVTME_MAX_LISTS][SVTME_MAX_REFS][SVTME_PU_COUNT], best_mv[SVTME_MAX_LISTS][SVTME_MAX_REFS][SVTME_PU_COUNT];
    static svtme_search_result sr[SVTME_MAX_LISTS][SVTME_MAX_REFS];
    static uint32_t            zz[SVTME_MAX_LISTS][SVTME_MAX_REFS], divisor[SVTME_MAX_LISTS][SVTME_MAX_REFS];
    uint64_t h_sad = 1469598103934665603ull, h_mv = h_sad, h_sr = h_sad;
    for (uint32_t b = 0; b < res.n_b64; ++b) {
        CHECK(svtme_unpack_b64(&res, b, &best_sad[0][0][0], &best_mv[0][0][0], &sr[0][0], &zz[0][0], &divisor[0][0]));
        h_sad = fnv(h_sad, best_sad, sizeof(best_sad));
        h_mv  = fnv(h_mv, best_mv, sizeof(best_mv));
        for (int l = 0; l < SVTME_MAX_LISTS; ++l)
            for (int r = 0; r < SVTME_MAX_REFS; ++r) {
                h_sr = fnv(h_sr, &sr[l][r].hme_sc_x, 2);
                h_sr = fnv(h_sr, &sr[l][r].hme_sc_y, 2);
                h_sr = fnv(h_sr, &sr[l][r].do_ref, 1);
                h_sr = fnv(h_sr, &sr[l][r].hme_sad, 8);
            }
        h_sr = fnv(h_sr, zz, sizeof(zz));
        h_sr = fnv(h_sr, divisor, sizeof(divisor));
    }
    printf("%u %016llx %016llx %016llx\n", res.n_b64, (unsigned long long)h_sad, (unsigned long long)h_mv, (unsigned long long)h_sr);
    CHECK(svtme_picture_done(eng, ticket));
    for (int k = 0; k < N_PIC; ++k) CHECK(svtme_store_release(eng, 100 + (uint64_t)k));
    svtme_destroy(eng);
    for (int k = 0; k < N_PIC; ++k) free(luma[k]);
    free(cells);
    return 0;
}
