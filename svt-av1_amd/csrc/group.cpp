// group.cpp - several engines (one per GPU of a node) behind one handle: the picture-sharding scheduler of the C ABI.
//
// Open-loop ME reads only source pictures, so the pictures send_picture_out() posts back to back
// (Source/Lib/Encoder/Codec/EbPictureDecisionProcess.c:5031-5125, :6842-6874) are independent of each other: picture i
// of the dispatch order runs on engine i mod N. A source picture is uploaded to an engine the first time a picture that
// runs there reads it (as the picture itself or as a reference), so every source reaches exactly the GPUs that need it;
// results go from each GPU to its own pinned host buffers. No exchange between GPUs, no collective.
//
// Plain host code on top of the engine's own C entry points; nothing here touches the device directly.
#include <stdint.h>
#include <string.h>

#include <mutex>
#include <new>
#include <unordered_map>
#include <vector>

#include "../../include/svtme.h"

namespace {

struct Source {
    uint64_t       picture_number = 0;
    const uint8_t *luma = nullptr;
    uint32_t       stride = 0, width = 0, height = 0;
    int            pinned = 0;
    uint64_t       resident = 0; // bit i: engine i holds the current planes of this key
};

constexpr unsigned kEngineShift = 56; // group ticket = engine index << 56 | the engine's ticket

} // namespace

struct svtme_group {
    std::vector<svtme_engine *>          engines;
    std::mutex                           mu;
    std::unordered_map<uint64_t, Source> sources;
    uint64_t                             next = 0; // pictures submitted so far: the next one's dispatch index
    uint64_t                             n_sources = 0, n_uploads = 0;
};

extern "C" int svtme_group_create(const int32_t *devices, uint32_t n_devices, const svtme_config *cfg, svtme_group **out) {
    if (!devices || !cfg || !out || n_devices < 1 || n_devices > 64) return SVTME_ERR_BAD_ARG;
    svtme_group *g = new (std::nothrow) svtme_group();
    if (!g) return SVTME_ERR_NO_RESOURCES;
    for (uint32_t i = 0; i < n_devices; ++i) {
        svtme_config c = *cfg;
        c.device       = devices[i];
        svtme_engine *e = nullptr;
        const int     rc = svtme_create(&c, &e);
        if (rc != SVTME_OK) {
            svtme_group_destroy(g);
            return rc;
        }
        g->engines.push_back(e);
    }
    *out = g;
    return SVTME_OK;
}

extern "C" void svtme_group_destroy(svtme_group *g) {
    if (!g) return;
    for (svtme_engine *e : g->engines) svtme_destroy(e);
    delete g;
}

extern "C" uint32_t svtme_group_size(const svtme_group *g) { return g ? (uint32_t)g->engines.size() : 0; }

extern "C" int svtme_group_store_put(svtme_group *g, uint64_t key, uint64_t picture_number, const uint8_t *luma, uint32_t stride,
                                     uint32_t width, uint32_t height, int pinned) {
    if (!g || !luma || stride < width) return SVTME_ERR_BAD_ARG;
    std::lock_guard<std::mutex> lock(g->mu);
    Source &s = g->sources[key]; // an existing key: its planes were rewritten (temporal filter); every copy is stale now
    s.picture_number = picture_number;
    s.luma           = luma;
    s.stride         = stride;
    s.width          = width;
    s.height         = height;
    s.pinned         = pinned;
    s.resident       = 0;
    g->n_sources++;
    return SVTME_OK;
}

extern "C" int svtme_group_store_release(svtme_group *g, uint64_t key) {
    if (!g) return SVTME_ERR_BAD_ARG;
    std::lock_guard<std::mutex> lock(g->mu);
    auto it = g->sources.find(key);
    if (it == g->sources.end()) return SVTME_ERR_NOT_FOUND;
    int rc = SVTME_OK;
    for (size_t i = 0; i < g->engines.size(); ++i)
        if (it->second.resident >> i & 1) {
            const int r = svtme_store_release(g->engines[i], key);
            if (r != SVTME_OK && r != SVTME_ERR_NOT_FOUND) rc = r;
        }
    g->sources.erase(it);
    return rc;
}

// Makes `key` resident on engine i (group lock held).
static int ensure_resident(svtme_group *g, size_t i, uint64_t key) {
    auto it = g->sources.find(key);
    if (it == g->sources.end()) return SVTME_ERR_NOT_FOUND;
    Source &s = it->second;
    if (s.resident >> i & 1) return SVTME_OK;
    const int rc = s.pinned ? svtme_store_put_pinned(g->engines[i], key, s.picture_number, s.luma, s.stride, s.width, s.height)
                            : svtme_store_put(g->engines[i], key, s.picture_number, s.luma, s.stride, s.width, s.height);
    if (rc != SVTME_OK) return rc;
    s.resident |= 1ull << i;
    g->n_uploads++;
    return SVTME_OK;
}

extern "C" int svtme_group_picture_submit(svtme_group *g, const svtme_picture *pic, svtme_ticket *ticket) {
    if (!g || !pic || !ticket) return SVTME_ERR_BAD_ARG;
    if (pic->num_of_list_to_search < 1 || pic->num_of_list_to_search > SVTME_MAX_LISTS || pic->num_of_ref_pic_to_search[0] > SVTME_MAX_REFS ||
        pic->num_of_ref_pic_to_search[1] > SVTME_MAX_REFS)
        return SVTME_ERR_BAD_ARG;
    std::lock_guard<std::mutex> lock(g->mu);
    const size_t i = (size_t)(g->next % g->engines.size());
    int          rc = ensure_resident(g, i, pic->cur_key);
    for (int l = 0; l < pic->num_of_list_to_search && rc == SVTME_OK; ++l)
        for (int r = 0; r < pic->num_of_ref_pic_to_search[l] && rc == SVTME_OK; ++r) rc = ensure_resident(g, i, pic->ref_key[l][r]);
    if (rc != SVTME_OK) return rc;
    svtme_ticket t = 0;
    rc             = svtme_picture_submit(g->engines[i], pic, &t);
    if (rc != SVTME_OK) return rc;
    if (t >> kEngineShift) return SVTME_ERR_NO_RESOURCES; // (2^56 pictures on one engine)
    *ticket = ((svtme_ticket)i << kEngineShift) | t;
    g->next++;
    return SVTME_OK;
}

extern "C" int svtme_group_picture_wait(svtme_group *g, svtme_ticket ticket, svtme_results *out) {
    if (!g) return SVTME_ERR_BAD_ARG;
    const size_t i = (size_t)(ticket >> kEngineShift);
    if (i >= g->engines.size()) return SVTME_ERR_NOT_FOUND;
    return svtme_picture_wait(g->engines[i], ticket & ((1ull << kEngineShift) - 1), out);
}

extern "C" int svtme_group_picture_done(svtme_group *g, svtme_ticket ticket) {
    if (!g) return SVTME_ERR_BAD_ARG;
    const size_t i = (size_t)(ticket >> kEngineShift);
    if (i >= g->engines.size()) return SVTME_ERR_NOT_FOUND;
    return svtme_picture_done(g->engines[i], ticket & ((1ull << kEngineShift) - 1));
}

// The engine that holds most of the listed sources already gets the rest (group lock held by this function only).
extern "C" int svtme_group_engine_for(svtme_group *g, const uint64_t *keys, uint32_t n_keys, svtme_engine **engine) {
    if (!g || !keys || !engine || n_keys < 1 || n_keys > 1 + SVTME_MAX_LISTS * SVTME_MAX_REFS) return SVTME_ERR_BAD_ARG;
    std::lock_guard<std::mutex> lock(g->mu);
    size_t best = 0;
    int    best_count = -1;
    for (size_t i = 0; i < g->engines.size(); ++i) {
        int count = 0;
        for (uint32_t k = 0; k < n_keys; ++k) {
            auto it = g->sources.find(keys[k]);
            if (it == g->sources.end()) return SVTME_ERR_NOT_FOUND;
            count += (int)(it->second.resident >> i & 1);
        }
        if (count > best_count) {
            best_count = count;
            best       = i;
        }
    }
    for (uint32_t k = 0; k < n_keys; ++k)
        if (const int rc = ensure_resident(g, best, keys[k])) return rc;
    *engine = g->engines[best];
    return SVTME_OK;
}

extern "C" int svtme_group_gm_set_tables(svtme_group *g, const int16_t *warped_filter, const int32_t *error_measure_lut) {
    if (!g) return SVTME_ERR_BAD_ARG;
    for (svtme_engine *e : g->engines)
        if (const int rc = svtme_gm_set_tables(e, warped_filter, error_measure_lut)) return rc;
    return SVTME_OK;
}

extern "C" int svtme_group_set_tile_rects(svtme_group *g, uint32_t width, uint32_t height, const svtme_tile_rect *rects, uint32_t n_b64) {
    if (!g) return SVTME_ERR_BAD_ARG;
    for (svtme_engine *e : g->engines)
        if (const int rc = svtme_set_tile_rects(e, width, height, rects, n_b64)) return rc;
    return SVTME_OK;
}

extern "C" int svtme_group_stats(svtme_group *g, uint64_t *pictures, uint64_t *sources, uint64_t *uploads) {
    if (!g) return SVTME_ERR_BAD_ARG;
    std::lock_guard<std::mutex> lock(g->mu);
    if (pictures) *pictures = g->next;
    if (sources) *sources = g->n_sources;
    if (uploads) *uploads = g->n_uploads;
    return SVTME_OK;
}
