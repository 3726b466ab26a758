// group.cpp - several engines (one per GPU of a node) behind one handle: the picture-sharding scheduler of the C ABI.
//
// Open-loop ME reads only source pictures, so the pictures send_picture_out() posts back to back
// (Source/Lib/Encoder/Codec/EbPictureDecisionProcess.c:5031-5125, :6842-6874) are independent of each other. A mini-GOP
// handed over at once (svtme_group_pictures_submit) is split over the engines by accumulated work - each picture goes
// to the engine where its estimated cost plus the cost of bringing its missing sources there ends earliest - and every
// engine gets its share as ONE batch. A source picture reaches an engine the first time a picture that runs there reads
// it: from an engine that already holds its built planes, device to device (xGMI between peer GPUs), else by an upload
// from the host; the sources an engine is missing for a batch go up together (svtme_store_put_many). Results go from
// each GPU to its own pinned host buffers. No exchange between GPUs inside a picture, no collective.
//
// Host code on top of the engine's own C entry points; the only HIP calls here ask which devices are peers.
#include <hip/hip_runtime_api.h>
#include <stdint.h>
#include <string.h>

#include <condition_variable>
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
    uint64_t       version = 0; // of the announcement: a copy that arrives after a re-announcement is stale
    uint64_t       resident = 0; // bit i: engine i holds the current planes of this key
    uint64_t       holders = 0; // bit i: engine i holds a slot for this key, current or stale (what a release has to free)
    uint64_t       arriving = 0; // bit i: a copy to engine i is under way (outside the group's lock)
};

constexpr unsigned kEngineShift = 56; // group ticket = engine index << 56 | the engine's ticket
constexpr int      kMaxKeys = 1 + SVTME_MAX_LISTS * SVTME_MAX_REFS;

// get_scaled_picture_distance, EbMotionEstimation.c:1276-1280
int scaled_dist(int dist) { return (uint16_t)(((dist * 5) / 8) + ((dist % 8) ? 1 : 0)); }

// A picture's search time in microseconds, roughly: the abs-diff-accumulate operations its windows ask for before
// clipping and early exits (pre-HME shapes, level-0 regions, levels 1 / 2, the integer window of every reference) at a
// typical rate, plus a fixed share per reference. Only ratios matter: it balances the engines.
double picture_cost_us(const svtme_picture &p) {
    const svtme_controls &k = p.ctrl;
    const double n_b64 = (double)((p.width + 63) / 64) * ((p.height + 63) / 64);
    double       ops = 0, refs = 0;
    for (int l = 0; l < p.num_of_list_to_search && l < SVTME_MAX_LISTS; ++l)
        for (int r = 0; r < p.num_of_ref_pic_to_search[l] && r < SVTME_MAX_REFS; ++r) {
            const int64_t d = (int64_t)p.picture_number - (int64_t)p.ref_picture_number[l][r];
            const int     dist = (uint16_t)(int16_t)(d < 0 ? -d : d);
            const int     f = p.me_type == SVTME_ME_MCTF ? dist : scaled_dist(dist);
            auto          area = [&](const svtme_sa_minmax &sa, int div_w, int div_h) {
                const double w = (double)(sa.sa_min.width / div_w) * f, h = (double)(sa.sa_min.height / div_h) * f;
                const double wm = sa.sa_max.width / div_w, hm = sa.sa_max.height / div_h;
                return (w < wm ? w : wm) * (h < hm ? h : hm);
            };
            ops += area(k.me_sa, 1, 1) * 2048.0;
            if (k.enable_hme_flag && (p.temporal_layer_index > 0 || l == 0)) {
                const int nw = k.num_hme_sa_w ? k.num_hme_sa_w : 1, nh = k.num_hme_sa_h ? k.num_hme_sa_h : 1;
                if (k.enable_hme_level0_flag) ops += nw * nh * area(k.hme_l0_sa, nw, nh) * 128.0;
                if (k.prehme_enable) ops += (area(k.prehme_sa_cfg[0], 1, 1) + area(k.prehme_sa_cfg[1], 1, 1)) * 128.0;
                if (k.enable_hme_level1_flag) ops += nw * nh * (double)k.hme_l1_sa.width * k.hme_l1_sa.height * 512.0;
                if (k.enable_hme_level2_flag) ops += nw * nh * (double)k.hme_l2_sa.width * k.hme_l2_sa.height * 2048.0;
            }
            refs += 1;
        }
    return n_b64 * (ops / 70e6 + refs * 0.004) + 8.0;
}

} // namespace

struct svtme_group {
    std::vector<svtme_engine *>          engines;
    std::vector<int>                     devices;
    std::vector<double>                  load_us; // accumulated estimated work per engine
    std::vector<uint64_t>                peers; // bit j of peers[i]: engine i can copy from engine j device to device
    std::mutex                           mu;
    std::condition_variable              cv; // a copy to an engine has ended
    std::unordered_map<uint64_t, Source> sources;
    uint64_t                             next = 0; // pictures submitted so far
    uint64_t                             version = 0;
    uint64_t                             n_sources = 0, n_uploads = 0, n_peer = 0;
    int                                  use_peers = 1;
};

extern "C" int svtme_group_create(const int32_t *devices, uint32_t n_devices, const svtme_config *cfg, svtme_group **out) {
    if (!devices || !cfg || !out || n_devices < 1 || n_devices > 56) return SVTME_ERR_BAD_ARG;
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
        g->devices.push_back(devices[i]);
    }
    g->load_us.assign(n_devices, 0.0);
    g->peers.assign(n_devices, 0);
    // which engines can hand planes to which: two engines on one GPU always, two GPUs when the runtime says they are peers
    // (access is switched on once per direction; "already enabled" is fine)
    int prev = -1;
    (void)hipGetDevice(&prev);
    for (uint32_t i = 0; i < n_devices; ++i)
        for (uint32_t j = 0; j < n_devices; ++j) {
            if (i == j) continue;
            int ok = devices[i] == devices[j];
            if (!ok && hipDeviceCanAccessPeer(&ok, devices[i], devices[j]) != hipSuccess) ok = 0;
            if (ok && devices[i] != devices[j]) {
                if (hipSetDevice(devices[i]) == hipSuccess) {
                    const hipError_t err = hipDeviceEnablePeerAccess(devices[j], 0);
                    if (err != hipSuccess && err != hipErrorPeerAccessAlreadyEnabled) ok = 0;
                }
                (void)hipGetLastError();
            }
            if (ok) g->peers[i] |= 1ull << j;
        }
    if (prev >= 0) (void)hipSetDevice(prev);
    (void)hipGetLastError();
    *out = g;
    return SVTME_OK;
}

extern "C" void svtme_group_destroy(svtme_group *g) {
    if (!g) return;
    for (svtme_engine *e : g->engines) svtme_destroy(e);
    delete g;
}

extern "C" uint32_t svtme_group_size(const svtme_group *g) { return g ? (uint32_t)g->engines.size() : 0; }

extern "C" int svtme_group_set_peer_copies(svtme_group *g, int enable) {
    if (!g) return SVTME_ERR_BAD_ARG;
    std::lock_guard<std::mutex> lock(g->mu);
    g->use_peers = enable != 0;
    return SVTME_OK;
}

extern "C" int svtme_group_store_put(svtme_group *g, uint64_t key, uint64_t picture_number, const uint8_t *luma, uint32_t stride,
                                     uint32_t width, uint32_t height, int pinned) {
    if (!g || !luma || stride < width) return SVTME_ERR_BAD_ARG;
    std::lock_guard<std::mutex> lock(g->mu);
    // an existing key: its planes were rewritten (temporal filter); every copy is stale now, but the engines that hold one
    // keep their slots (`holders`), which the next copy there overwrites and the release frees
    Source &s = g->sources[key];
    s.picture_number = picture_number;
    s.luma           = luma;
    s.stride         = stride;
    s.width          = width;
    s.height         = height;
    s.pinned         = pinned;
    s.resident       = 0;
    s.version        = ++g->version;
    g->n_sources++;
    return SVTME_OK;
}

extern "C" int svtme_group_store_release(svtme_group *g, uint64_t key) {
    if (!g) return SVTME_ERR_BAD_ARG;
    std::unique_lock<std::mutex> lock(g->mu);
    auto it = g->sources.find(key);
    if (it == g->sources.end()) return SVTME_ERR_NOT_FOUND;
    while (it->second.arriving) { // copies under way end first (their slots are among the holders then)
        g->cv.wait(lock);
        it = g->sources.find(key);
        if (it == g->sources.end()) return SVTME_ERR_NOT_FOUND;
    }
    const uint64_t holders = it->second.holders;
    g->sources.erase(it);
    lock.unlock();
    int rc = SVTME_OK;
    for (size_t i = 0; i < g->engines.size(); ++i)
        if (holders >> i & 1) {
            const int r = svtme_store_release(g->engines[i], key);
            if (r != SVTME_OK && r != SVTME_ERR_NOT_FOUND) rc = r;
        }
    return rc;
}

// Makes the keys (no duplicates) resident on engine i. Called without the group's lock; the copies run outside it, so
// that the other engines' submits go on meanwhile: the sources this engine lacks are marked as arriving, fetched - from
// a peer engine that holds them, else from the host, those in one batched put per size - and marked resident.
static int make_resident(svtme_group *g, size_t i, const uint64_t *keys, int n_keys) {
    struct Job {
        uint64_t key, version;
        Source   s;
        int      from; // engine to copy from, -1: upload
        int      rc;
    };
    std::vector<Job> jobs;
    {
        std::unique_lock<std::mutex> lock(g->mu);
        for (;;) {
            bool wait = false;
            for (int k = 0; k < n_keys; ++k) {
                auto it = g->sources.find(keys[k]);
                if (it == g->sources.end()) return SVTME_ERR_NOT_FOUND;
                if (it->second.arriving >> i & 1) wait = true; // another thread is bringing it here
            }
            if (!wait) break;
            g->cv.wait(lock);
        }
        for (int k = 0; k < n_keys; ++k) {
            Source &s = g->sources.find(keys[k])->second;
            if (s.resident >> i & 1) continue;
            Job j;
            j.key = keys[k], j.version = s.version, j.s = s, j.from = -1, j.rc = SVTME_OK;
            if (g->use_peers)
                for (size_t e = 0; e < g->engines.size() && j.from < 0; ++e)
                    if ((s.resident >> e & 1) && (g->peers[i] >> e & 1)) j.from = (int)e;
            s.arriving |= 1ull << i;
            jobs.push_back(j);
        }
    }
    if (jobs.empty()) return SVTME_OK;
    // ---- outside the lock
    for (Job &j : jobs)
        if (j.from >= 0) {
            j.rc = svtme_store_copy_peer(g->engines[i], g->engines[(size_t)j.from], j.key);
            if (j.rc == SVTME_ERR_NOT_FOUND) j.from = -1, j.rc = SVTME_OK; // (released there meanwhile: from the host then)
        }
    // uploads: runs of pictures with one geometry go up in one batched put
    std::vector<size_t> up;
    for (size_t k = 0; k < jobs.size(); ++k)
        if (jobs[k].from < 0) up.push_back(k);
    for (size_t a = 0; a < up.size();) {
        size_t         b = a;
        uint64_t       kk[SVTME_MAX_PUT_BATCH], pn[SVTME_MAX_PUT_BATCH];
        const uint8_t *lu[SVTME_MAX_PUT_BATCH];
        const Source  &s0 = jobs[up[a]].s;
        while (b < up.size() && b - a < SVTME_MAX_PUT_BATCH) {
            const Source &s = jobs[up[b]].s;
            if (s.width != s0.width || s.height != s0.height || s.stride != s0.stride || s.pinned != s0.pinned) break;
            kk[b - a] = jobs[up[b]].key, pn[b - a] = s.picture_number, lu[b - a] = s.luma;
            ++b;
        }
        const int rc = svtme_store_put_many(g->engines[i], (uint32_t)(b - a), kk, pn, lu, s0.stride, s0.width, s0.height, s0.pinned);
        for (size_t k = a; k < b; ++k) jobs[up[k]].rc = rc;
        a = b;
    }
    int rc = SVTME_OK;
    {
        std::lock_guard<std::mutex> lock(g->mu);
        for (const Job &j : jobs) {
            auto it = g->sources.find(j.key);
            if (it != g->sources.end()) {
                Source &s = it->second;
                s.arriving &= ~(1ull << i);
                if (j.rc == SVTME_OK) {
                    s.holders |= 1ull << i;
                    if (s.version == j.version) s.resident |= 1ull << i; // (else it was announced again meanwhile: stale)
                }
            }
            if (j.rc == SVTME_OK) {
                if (j.from >= 0) g->n_peer++;
                else g->n_uploads++;
            } else if (rc == SVTME_OK)
                rc = j.rc;
        }
    }
    g->cv.notify_all();
    if (rc != SVTME_OK) return rc;
    // a key that was announced again while its copy was under way is not resident yet: once more for those
    {
        std::lock_guard<std::mutex> lock(g->mu);
        for (int k = 0; k < n_keys; ++k) {
            auto it = g->sources.find(keys[k]);
            if (it == g->sources.end()) return SVTME_ERR_NOT_FOUND;
            if (!(it->second.resident >> i & 1)) rc = 1;
        }
    }
    return rc == 1 ? make_resident(g, i, keys, n_keys) : SVTME_OK;
}

static int keys_of(const svtme_picture &pic, uint64_t *keys) {
    int n = 0;
    keys[n++] = pic.cur_key;
    for (int l = 0; l < pic.num_of_list_to_search; ++l)
        for (int r = 0; r < pic.num_of_ref_pic_to_search[l]; ++r) {
            bool dup = false;
            for (int k = 0; k < n; ++k) dup = dup || keys[k] == pic.ref_key[l][r];
            if (!dup) keys[n++] = pic.ref_key[l][r];
        }
    return n;
}

extern "C" int svtme_group_pictures_submit(svtme_group *g, const svtme_picture *pics, uint32_t n, svtme_ticket *tickets) {
    if (!g || !pics || !tickets || n < 1 || n > SVTME_MAX_BATCH) return SVTME_ERR_BAD_ARG;
    for (uint32_t i = 0; i < n; ++i) {
        const svtme_picture &a = pics[0], &b = pics[i];
        if (b.num_of_list_to_search < 1 || b.num_of_list_to_search > SVTME_MAX_LISTS || b.num_of_ref_pic_to_search[0] > SVTME_MAX_REFS ||
            b.num_of_ref_pic_to_search[1] > SVTME_MAX_REFS)
            return SVTME_ERR_BAD_ARG;
        // what svtme_pictures_submit asks of a batch, checked before anything is queued anywhere
        if (a.width != b.width || a.height != b.height || memcmp(&a.ctrl, &b.ctrl, sizeof(a.ctrl)) || memcmp(&a.tail, &b.tail, sizeof(a.tail)) ||
            (a.me_type ? a.me_type : SVTME_ME_OPEN_LOOP) != (b.me_type ? b.me_type : SVTME_ME_OPEN_LOOP) ||
            a.bypass_zz_check != b.bypass_zz_check || a.restricted_motion_vector != b.restricted_motion_vector || a.b64_step != b.b64_step ||
            a.tf_me_exit_th != b.tf_me_exit_th)
            return SVTME_ERR_BAD_ARG;
        tickets[i] = 0;
    }
    const size_t n_eng = g->engines.size();
    // ---- who runs what: in dispatch order, each picture where it ends earliest
    std::vector<int> owner(n);
    {
        std::lock_guard<std::mutex>          lock(g->mu);
        std::unordered_map<uint64_t, uint64_t> planned; // key -> engines that will hold it once this call's copies are through
        for (uint32_t i = 0; i < n; ++i) {
            uint64_t  keys[kMaxKeys];
            const int nk = keys_of(pics[i], keys);
            uint64_t  have[kMaxKeys], anywhere[kMaxKeys];
            double    bytes = 0;
            for (int k = 0; k < nk; ++k) {
                auto it = g->sources.find(keys[k]);
                if (it == g->sources.end()) return SVTME_ERR_NOT_FOUND;
                auto pl = planned.find(keys[k]);
                have[k] = it->second.resident | it->second.arriving | (pl == planned.end() ? 0 : pl->second);
                anywhere[k] = have[k];
                bytes       = (double)it->second.width * it->second.height;
            }
            const double cost = picture_cost_us(pics[i]);
            size_t       best = 0;
            double       best_end = 0;
            for (size_t e = 0; e < n_eng; ++e) {
                double fetch = 0;
                for (int k = 0; k < nk; ++k)
                    if (!(have[k] >> e & 1)) // device to device where a peer has it (the three planes), else over PCIe (the luma)
                        fetch += (g->use_peers && (anywhere[k] & g->peers[e])) ? 1.32 * bytes / 60e3 : bytes / 25e3;
                const double end = g->load_us[e] + cost + fetch;
                if (e == 0 || end < best_end) best = e, best_end = end;
            }
            owner[i] = (int)best;
            g->load_us[best] = best_end;
            for (int k = 0; k < nk; ++k) planned[keys[k]] |= 1ull << best;
        }
    }
    // ---- per engine: its sources, then its pictures as one batch
    int first_error = SVTME_OK;
    for (size_t e = 0; e < n_eng; ++e) {
        svtme_picture mine[SVTME_MAX_BATCH];
        uint32_t      idx[SVTME_MAX_BATCH], m = 0;
        uint64_t      keys[SVTME_MAX_BATCH * kMaxKeys];
        int           nk = 0;
        for (uint32_t i = 0; i < n; ++i)
            if (owner[i] == (int)e) {
                mine[m] = pics[i];
                idx[m++] = i;
                uint64_t  kk[kMaxKeys];
                const int c = keys_of(pics[i], kk);
                for (int k = 0; k < c; ++k) {
                    bool dup = false;
                    for (int q = 0; q < nk; ++q) dup = dup || keys[q] == kk[k];
                    if (!dup) keys[nk++] = kk[k];
                }
            }
        if (!m) continue;
        int rc = make_resident(g, e, keys, nk);
        svtme_ticket t[SVTME_MAX_BATCH];
        if (rc == SVTME_OK) rc = svtme_pictures_submit(g->engines[e], mine, m, t);
        if (rc != SVTME_OK) {
            if (first_error == SVTME_OK) first_error = rc;
            continue;
        }
        for (uint32_t j = 0; j < m; ++j) {
            if (t[j] >> kEngineShift) return SVTME_ERR_NO_RESOURCES; // (2^56 pictures on one engine)
            tickets[idx[j]] = ((svtme_ticket)e << kEngineShift) | t[j];
        }
        std::lock_guard<std::mutex> lock(g->mu);
        g->next += m;
    }
    return first_error;
}

extern "C" int svtme_group_picture_submit(svtme_group *g, const svtme_picture *pic, svtme_ticket *ticket) {
    return svtme_group_pictures_submit(g, pic, 1, ticket);
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

extern "C" uint32_t svtme_group_ticket_engine(svtme_ticket ticket) { return (uint32_t)(ticket >> kEngineShift); }

// The engine that holds most of the listed sources already gets the rest.
extern "C" int svtme_group_engine_for(svtme_group *g, const uint64_t *keys, uint32_t n_keys, svtme_engine **engine) {
    if (!g || !keys || !engine || n_keys < 1 || n_keys > (uint32_t)kMaxKeys) return SVTME_ERR_BAD_ARG;
    size_t best = 0;
    {
        std::lock_guard<std::mutex> lock(g->mu);
        int best_count = -1;
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
    }
    uint64_t uniq[kMaxKeys];
    int      nu = 0;
    for (uint32_t k = 0; k < n_keys; ++k) {
        bool dup = false;
        for (int q = 0; q < nu; ++q) dup = dup || uniq[q] == keys[k];
        if (!dup) uniq[nu++] = keys[k];
    }
    if (const int rc = make_resident(g, best, uniq, nu)) return rc;
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

extern "C" int svtme_group_peer_stats(svtme_group *g, uint64_t *peer_copies, uint64_t *peer_links) {
    if (!g) return SVTME_ERR_BAD_ARG;
    std::lock_guard<std::mutex> lock(g->mu);
    if (peer_copies) *peer_copies = g->n_peer;
    if (peer_links) {
        uint64_t n = 0;
        for (uint64_t m : g->peers) n += (uint64_t)__builtin_popcountll(m);
        *peer_links = n;
    }
    return SVTME_OK;
}
