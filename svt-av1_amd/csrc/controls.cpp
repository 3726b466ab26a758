// controls.cpp - host mirror of the reference's per-preset ME signal derivation.
//
// svtme_derive_controls restates signal_derivation_me_kernel_oq (Source/Lib/Encoder/Codec/EbMotionEstimationProcess.c:430-538)
// and the setters it calls (set_me_search_params :154, set_hme_search_params :68, set_prehme_ctrls :385,
// set_me_hme_ref_prune_ctrls :241, set_me_sr_adjustment_ctrls :290), plus the HME level flags that
// signal_derivation_multi_processes_oq sets per preset (EbPictureDecisionProcess.c:1797-1814). The encoder-side
// shim can either call this or copy the fields of its own MeContext into svtme_controls; both give the same bytes.
#include <string.h>

#include "../../include/svtme.h"

namespace {

struct SaRow {
    int      max_mode; // row applies while enc_mode <= max_mode
    uint16_t min_w, min_h, max_w, max_h;
};

inline svtme_sa_minmax mk(uint16_t a, uint16_t b, uint16_t c, uint16_t d) {
    svtme_sa_minmax r;
    r.sa_min.width = a, r.sa_min.height = b, r.sa_max.width = c, r.sa_max.height = d;
    return r;
}

inline svtme_sa_minmax lookup(const SaRow *rows, int n, int mode) {
    for (int i = 0; i < n; ++i)
        if (mode <= rows[i].max_mode) return mk(rows[i].min_w, rows[i].min_h, rows[i].max_w, rows[i].max_h);
    return mk(rows[n - 1].min_w, rows[n - 1].min_h, rows[n - 1].max_w, rows[n - 1].max_h);
}

enum { RES_480P = 2, RES_720P = 3, RES_1080P = 4, RES_4K = 5 };

// integer-search areas for screen content, set_me_search_params :157-183
const SaRow kMeSc[] = {{2, 175, 175, 750, 750}, {4, 125, 125, 500, 500}, {6, 75, 75, 350, 350}, {7, 48, 48, 224, 224},
                       {8, 32, 32, 164, 164},   {10, 32, 32, 96, 96},    {12, 16, 16, 96, 96},  {13, 8, 8, 32, 32}};
// non screen content, resolution independent part, :184-197
const SaRow kMeSlow[] = {{0, 64, 64, 256, 256}, {2, 64, 64, 128, 128}, {4, 16, 16, 64, 64}, {5, 16, 16, 64, 32}};

svtme_sa_minmax me_area(int m, int sc, int res) {
    if (sc) return lookup(kMeSc, 8, m);
    if (m <= 5) return lookup(kMeSlow, 4, m);
    if (m <= 9) return res < RES_1080P ? mk(16, 16, 32, 16) : mk(16, 6, 16, 9); // :198-205
    if (m <= 11) return res < RES_4K ? mk(8, 5, 16, 9) : mk(8, 1, 8, 1); // :206-214
    if (m <= 12) { // :215-228
        if (res < RES_720P) return mk(8, 3, 16, 9);
        if (res < RES_1080P) return mk(8, 1, 16, 7);
        if (res < RES_4K) return mk(8, 1, 8, 7);
        return mk(8, 1, 8, 1);
    }
    return mk(8, 3, 8, 3);
}

// set_hme_search_params :74-139
svtme_sa_minmax hme_l0_area(int m, int sc, int res) {
    const svtme_sa_minmax wide = mk(240, 240, 480, 480), std32 = mk(32, 32, 192, 192), std16 = mk(16, 16, 192, 192);
    if (m <= -2) return wide;
    if (m <= 0) return res < RES_4K ? std32 : wide;
    if (m <= 5) return std32;
    if (m <= 6) return (res <= RES_1080P && !sc) ? std16 : std32;
    if (m <= 11) return sc ? std32 : std16;
    if (sc) return std32; // <= M13
    return res < RES_4K ? mk(8, 8, 96, 96) : mk(16, 16, 96, 96);
}

} // namespace

extern "C" uint8_t svtme_input_resolution(uint32_t width, uint32_t height) {
    // derive_input_resolution, Source/Lib/Encoder/Codec/EbSequenceControlSet.c:396-415 (EbDefinitions.h:2025-2030)
    const uint32_t th[6] = {0x28500, 0x4CE00, 0xA1400, 0x16DA00, 0x535200, 0x140A000};
    const uint32_t sz    = width * height;
    uint8_t        r     = 0;
    while (r < 6 && sz >= th[r]) ++r;
    return r;
}

extern "C" int svtme_derive_tf_controls(uint8_t hme_me_level, svtme_controls *c) {
    if (!c || hme_me_level > 2) return SVTME_ERR_BAD_ARG;
    memset(c, 0, sizeof(*c));
    // tf_set_me_hme_params_oq, EbMotionEstimationProcess.c:557-591: {HME level-0 area, integer search area} per level
    static const svtme_sa_minmax l0_area[3] = {{{30, 30}, {60, 60}}, {{16, 16}, {32, 32}}, {{8, 8}, {16, 16}}};
    static const svtme_sa_minmax me_area_tf[3] = {{{60, 60}, {120, 120}}, {{16, 16}, {32, 32}}, {{8, 4}, {16, 8}}};
    c->num_hme_sa_w = c->num_hme_sa_h = 2;
    c->hme_l0_sa = l0_area[hme_me_level];
    c->me_sa     = me_area_tf[hme_me_level];
    c->hme_l1_sa = c->hme_l2_sa = svtme_sa{16, 16};
    // tf_enable_hme_*_flag, EbPictureDecisionProcess.c:1816-1841
    c->enable_hme_flag        = 1;
    c->enable_hme_level0_flag = 1;
    c->enable_hme_level1_flag = hme_me_level <= 1;
    c->enable_hme_level2_flag = hme_me_level == 0;
    c->hme_search_method = 1; // FULL_SAD_SEARCH, EbTemporalFiltering.c:331 (overrides :609)
    c->me_search_method  = 0; // SUB_SAD_SEARCH :611
    // no pre-HME, no pruning, no search-range adjustment, no early exit :613-627
    c->prune_ref_if_hme_sad_dev_bigger_than_th = 0xFFFF;
    c->prune_ref_if_me_sad_dev_bigger_than_th  = 0xFFFF;
    c->stat_factor = 100;
    return SVTME_OK;
}

extern "C" int svtme_derive_first_pass_controls(uint8_t input_resolution, uint32_t frame_rate, uint8_t reduce_me_search,
                                                svtme_controls *c) {
    if (!c || input_resolution > 6 || reduce_me_search > 1) return SVTME_ERR_BAD_ARG;
    memset(c, 0, sizeof(*c));
    // set_first_pass_me_hme_params_oq, firstpass.c:627-662
    c->stat_factor  = 100;
    c->num_hme_sa_w = c->num_hme_sa_h = 2;
    c->me_sa     = reduce_me_search ? mk(8, 3, 8, 5) : mk(8, 8, 8, 8);
    c->hme_l0_sa = input_resolution < RES_1080P ? mk(4, 4, 96, 96) : mk(8, 8, 96, 96);
    c->hme_l1_sa = c->hme_l2_sa = svtme_sa{8, 8};
    if ((frame_rate >> 16) < 50) { // low frame rate: minimum area x1.5
        c->me_sa.sa_min.width  = (uint16_t)((c->me_sa.sa_min.width * 3) >> 1);
        c->me_sa.sa_min.height = (uint16_t)((c->me_sa.sa_min.height * 3) >> 1);
    }
    // every HME level on (:488-491), level 2 off with the reduced search (:684-690); SUB_SAD_SEARCH; no pre-HME, no
    // pruning, no search-range adjustment (:692-711)
    c->enable_hme_flag = c->enable_hme_level0_flag = c->enable_hme_level1_flag = 1;
    c->enable_hme_level2_flag = !reduce_me_search;
    c->prune_ref_if_hme_sad_dev_bigger_than_th = 0xFFFF;
    c->prune_ref_if_me_sad_dev_bigger_than_th  = 0xFFFF;
    return SVTME_OK;
}

extern "C" int svtme_derive_controls(const svtme_preset_key *key, svtme_controls *c) {
    if (!key || !c) return SVTME_ERR_BAD_ARG;
    const int m = key->enc_mode, sc = key->sc_class1 != 0, res = key->input_resolution;
    if (m < -2 || m > 13 || res > 6) return SVTME_ERR_BAD_ARG;
    memset(c, 0, sizeof(*c));

    c->enable_hme_flag        = 1;
    c->enable_hme_level0_flag = 1;
    c->enable_hme_level1_flag = 1; // every preset up to M13 (EbPictureDecisionProcess.c:1799-1810)
    c->enable_hme_level2_flag = (sc || m <= 6) ? 1 : 0;

    c->me_sa = me_area(m, sc, res);
    if ((key->frame_rate >> 16) < 50) { // low frame rate: minimum area x1.5, :235-239
        c->me_sa.sa_min.width  = (uint16_t)((c->me_sa.sa_min.width * 3) >> 1);
        c->me_sa.sa_min.height = (uint16_t)((c->me_sa.sa_min.height * 3) >> 1);
    }
    c->num_hme_sa_w = c->num_hme_sa_h = 2;
    c->hme_l0_sa = hme_l0_area(m, sc, res);
    const svtme_sa refine = m <= 1 ? svtme_sa{16, 16} : svtme_sa{8, 3}; // :141-148
    c->hme_l1_sa = c->hme_l2_sa = refine;
    c->hme_search_method = c->me_search_method = 0; // SUB_SAD_SEARCH :448-449
    c->stat_factor = (sc || m <= 12) ? 100 : 80; // :450-455
    if (!sc && m > 10) { // :457-468
        c->reduce_hme_l0_sr_th_min = 8;
        c->reduce_hme_l0_sr_th_max = 200;
    }

    // pre-HME level :469-485 and its shapes :385-424 (level 2 is never selected)
    const int prehme = !c->enable_hme_level1_flag ? 0 : (sc || m <= 11 ? 1 : 3);
    if (prehme) {
        c->prehme_enable    = 1;
        c->prehme_sa_cfg[0] = prehme == 1 ? mk(8, 100, 8, 400) : mk(8, 100, 8, 350);
        c->prehme_sa_cfg[1] = prehme == 1 ? mk(96, 3, 384, 3) : mk(32, 7, 128, 7);
        c->prehme_skip_search_line = c->prehme_l1_early_exit = prehme == 3;
    }

    // reference pruning level :487-506 -> thresholds :241-288
    const int prune = m <= -2 ? 0 : (m <= 1 ? 2 : ((!sc || m <= 9) ? 5 : 6));
    if (prune == 0) {
        c->prune_ref_if_hme_sad_dev_bigger_than_th = c->prune_ref_if_me_sad_dev_bigger_than_th = 0xFFFF;
    } else {
        c->enable_me_hme_ref_pruning               = 1;
        c->prune_ref_if_hme_sad_dev_bigger_than_th = prune == 2 ? 80 : (prune == 5 ? 5 : 200);
        c->prune_ref_if_me_sad_dev_bigger_than_th  = 60;
        c->protect_closest_refs                    = prune == 6 ? 0 : 1;
    }

    // search-range adjustment level :507-520 -> thresholds :290-359
    const int sr = sc ? (m <= 9 ? 4 : 5) : (m <= -2 ? 0 : (m <= 3 ? 1 : (m <= 4 ? 2 : 3)));
    if (sr) {
        static const uint16_t mv_th[6] = {0, 4, 4, 4, 16, 20}, stat_th[6] = {0, 12000, 12000, 12000, 20000, 24000},
                              low_th[6] = {0, 6000, 6000, 12000, 20000, 24000};
        c->enable_me_sr_adjustment              = sr >= 4 ? 2 : 1;
        c->distance_based_hme_resizing          = sr >= 2;
        c->reduce_me_sr_based_on_mv_length_th   = mv_th[sr];
        c->stationary_hme_sad_abs_th            = stat_th[sr];
        c->reduce_me_sr_based_on_hme_sad_abs_th = low_th[sr];
        c->stationary_me_sr_divisor = c->me_sr_divisor_for_low_hme_sad = 8;
        if (!c->enable_hme_level2_flag) { // thresholds were tuned for level-2 SADs, :346-358
            const int d = c->enable_hme_level1_flag ? 4 : 16;
            c->stationary_hme_sad_abs_th            = (uint16_t)(c->stationary_hme_sad_abs_th / d);
            c->reduce_me_sr_based_on_hme_sad_abs_th = (uint16_t)(c->reduce_me_sr_based_on_hme_sad_abs_th / d);
        }
    }

    c->prune_me_candidates_th     = m <= 7 ? 0 : 65; // :522-525
    c->use_best_unipred_cand_only = m <= 9 ? 0 : 1; // EbPictureDecisionProcess.c:2058-2061
    c->me_early_exit_th           = m <= 6 ? 0 : (m <= 11 ? 64 * 64 * 8 : 64 * 64 * 9); // :529-534
    return SVTME_OK;
}
