// engine.cpp - host side of the engine: device picture store, picture queue, C ABI (include/svtme.h).
//
// HIP streams: the `s_store` streams carry the uploads and the store-building kernels (puts take turns on them, each with
// its own upload buffer, so one put's DMA overlaps the previous put's kernels), the `s_me` streams the search kernels
// (pictures take turns on up to three of them, so one picture's kernel tails overlap the others' kernels), `s_copy`
// the result copies when fewer than three search streams are in use (with three, a picture's copy follows its kernels
// on its own stream). A picture waits (on the device, through events) for the store slots it reads; a slot that is
// rewritten waits for the last picture that read it. Nothing here blocks the host except svtme_picture_wait,
// svtme_store_wait and a put that reuses a slot whose previous upload is still in flight; the engine mutex is held for
// bookkeeping and stream enqueues only, never across a frame copy or a wait.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <math.h>

#include <condition_variable>
#include <mutex>
#include <new>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

#include "svtme_device.h"

using namespace svtme;

#define HIP_TRY(expr)                                                                                     \
    do {                                                                                                  \
        hipError_t err__ = (expr);                                                                        \
        if (err__ != hipSuccess) {                                                                        \
            fprintf(stderr, "svtme: %s failed: %s (%s:%d)\n", #expr, hipGetErrorString(err__), __FILE__, __LINE__); \
            return SVTME_ERR_DEVICE;                                                                      \
        }                                                                                                 \
    } while (0)

namespace {

constexpr int kPad[3] = {16, 32, 72}; // sixteenth, quarter, full: EbEncHandle.c:1331-1355, :3447-3460
constexpr uint32_t kMaxExtraStride = 256; // svtme_store_put_pinned: stride - width it copies in one piece (the encoder's is 144)

struct Slot {
    bool       used = false;
    bool       putting = false; // a put has reserved the slot and is copying outside the lock: not yet visible to submits
    bool       uploaded = false; // `ready` has been recorded at least once (an upload from host_stage may be in flight)
    bool       release_pending = false;
    uint64_t   key = 0, picture_number = 0;
    uint64_t   gen = 0; // number of the put that built the planes: tells a cached by-product (corner lists) from a stale one
    int        readers = 0; // submitted pictures that have not been waited for yet (and global-motion / TPL calls that are running)
    int        sync_readers = 0; // of those, the global-motion / TPL calls: no event orders a rewrite behind them, a put waits
    int        put_waiting = 0; // puts that wait for them: new global-motion / TPL readers queue up behind such a put
    uint32_t   width = 0, height = 0;
    DevPicture pic{};
    uint8_t   *mem = nullptr; // device
    uint8_t   *host_stage = nullptr; // pinned
    hipEvent_t ready = nullptr; // planes built (recorded on s_store)
    // per search stream: the end-of-kernels event of the last picture that read the planes (the event belongs to that
    // picture's Flight; read_recorded is cleared again once that picture has been waited for)
    hipEvent_t last_read[4] = {nullptr, nullptr, nullptr, nullptr};
    bool       read_recorded[4] = {false, false, false, false};
    bool       ready_seen[4] = {false, false, false, false}; // this search stream already waits behind `ready`
};

struct Flight {
    bool                 busy = false, waited = false, kernel_timing = false;
    svtme_ticket         ticket = 0;
    uint32_t             n_b64 = 0;
    std::vector<int>     slots; // store slots referenced
    // device
    HmeRefOut           *d_hme = nullptr;
    MeJob               *d_jobs = nullptr;
    unsigned long long  *d_keys = nullptr;
    uint8_t             *d_res = nullptr; // the result block, see layout()
    // pinned host mirror of d_res
    uint8_t             *h_res = nullptr;
    size_t               off_sad = 0, off_mv = 0, off_sr = 0, off_zz = 0, off_div = 0, off_ops = 0, res_bytes = 0;
    // One block, laid out per picture so that a single copy brings exactly what the host asked for:
    //   [search results | zz | divisors | counters] [device-tail outputs, when the picture has a tail]
    //   [best SAD planes] [best MV planes]: one plane of n_b64 x 85 entries per searched reference (list-0 references
    //   first, then list 1), so that references a picture does not have cost nothing on the way to the host.
    void layout(size_t nb, size_t n_refs, bool with_tail) {
        const size_t n = nb * SVTME_MAX_LISTS * SVTME_MAX_REFS;
        off_sr    = 0;
        off_zz    = off_sr + n * sizeof(svtme_search_result);
        off_div   = off_zz + n * 4;
        off_ops   = (off_div + n * 4 + 63) / 64 * 64;
        off_thme  = off_ops + SVTME_OPS_BYTES;
        tail_end  = off_thme;
        if (with_tail) {
            off_tmv   = off_thme + nb * 8 * 8;
            off_tdist = off_tmv + nb * SVTME_PU_COUNT * SVTME_TAIL_MV_STRIDE * 4;
            off_tstat = off_tdist + nb * SVTME_PU_COUNT * 4;
            off_tcand = off_tstat + nb * 8 * 4;
            off_ttot  = off_tcand + nb * SVTME_PU_COUNT * SVTME_TAIL_CAND_STRIDE;
            off_tdo   = off_ttot + nb * SVTME_PU_COUNT;
            tail_end  = (off_tdo + nb * 8 + 63) / 64 * 64;
        }
        off_sad   = tail_end;
        off_mv    = off_sad + n_refs * nb * SVTME_PU_COUNT * 4;
        res_bytes = off_mv + n_refs * nb * SVTME_PU_COUNT * 4;
    }
    size_t off_thme = 0, off_tmv = 0, off_tdist = 0, off_tstat = 0, off_tcand = 0, off_ttot = 0, off_tdo = 0, tail_end = 0;
    uint32_t n_ref[2] = {0, 0};
    bool   tail = false, skip_search_arrays = false;
    int    batch = -1; // BatchRec the picture was submitted with
    int    param = -1; // its record in the parameter ring
};

// The pictures of one svtme_pictures_submit call: they share their launches, hence their stream and events.
struct BatchRec {
    bool     used = false, kernel_timing = false;
    int      n = 0, members_left = 0; // pictures; pictures not yet handed back with svtme_picture_done
    unsigned stream_index = 0; // search stream the batch runs on
    // start, after each of the 2 HME and 3 ME kernels, after the result copies
    hipEvent_t ev[7] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
};

// Global-motion estimation: buffers shared by the svtme_gm_* calls of one engine (allocated by the first of them).
struct GmState {
    std::mutex mu; // one call at a time
    bool       ready = false, tables = false;
    uint8_t   *d_score = nullptr, *d_keep = nullptr; // FAST score map and the flags that survive non-maximum suppression
    int       *d_rows = nullptr; // corners kept per row
    int       *d_corners[2] = {nullptr, nullptr}; // two corner lists: count, pad, (x, y) pairs
    int       *d_match = nullptr; // [SVTME_GM_MAX_CORNERS][5]
    uint32_t  *d_block_err = nullptr; // per 32x32 block
    long long *d_result = nullptr;
    GmRefineState *d_refine = nullptr; // svtme_gm_refine's loop state
    int16_t   *d_filter = nullptr; // the caller's warp filter [193][8]
    int32_t   *d_lut = nullptr; // and error table [512]
    int       *h_buf = nullptr; // pinned: what a call brings back
    struct Cached {
        bool     valid = false;
        uint64_t key = 0, gen = 0;
        int      level = 0;
    } cached[2]; // whose corners d_corners[i] holds
    int last_used = 1;
    // svtme_tpl_search: the caller's ME results and the block records, device and pinned host copies (first call allocates)
    uint8_t *d_tpl = nullptr, *h_tpl = nullptr;
    size_t   tpl_bytes = 0;
    // svtme_tpl_recon: block records, statistics, the picture's record and the reconstructed picture (first call allocates)
    uint8_t *d_rec = nullptr, *h_rec = nullptr;
    size_t   rec_bytes = 0;
    // the picture the last svtme_tpl_recon left in d_rec (svtme_tpl_recon_keep builds store planes from it): where, and the
    // events "reconstructed" (on the calls' stream) and "read by the store kernels" (on a store stream)
    size_t     rec_pic_off = 0, rec_pic_stride = 0;
    uint32_t   rec_w = 0, rec_h = 0;
    hipEvent_t rec_done = nullptr, rec_read = nullptr;
    bool       rec_read_pending = false;
};
constexpr size_t kGmCornerInts = 2 + 2 * (size_t)SVTME_GM_MAX_CORNERS, kGmMatchInts = 5 * (size_t)SVTME_GM_MAX_CORNERS;

} // namespace

// ---- scratch of svtme_picture_search_static(): the search kernels' intermediates (HME records, operation counters,
// integer-search windows, running minima) for one picture of at most SVTME_STATIC_MAX_B64 blocks, in device memory that
// belongs to the loaded library instead of an allocation. Every GPU the library runs on holds its own copy until the
// process ends. In host code the array's name is only a handle for hipGetSymbolAddress: the device address is taken once
// per device (static_scratch_of) and that pointer is what kernels, memsets and copies get.
constexpr size_t kStaticHmeBytes  = (size_t)SVTME_STATIC_MAX_B64 * SVTME_MAX_LISTS * SVTME_MAX_REFS * sizeof(HmeRefOut);
constexpr size_t kStaticOpsOff    = (kStaticHmeBytes + 255) / 256 * 256;
constexpr size_t kStaticJobsOff   = kStaticOpsOff + SVTME_OPS_BYTES;
constexpr size_t kStaticKeysOff   = (kStaticJobsOff + (size_t)SVTME_STATIC_MAX_B64 * SVTME_MAX_LISTS * SVTME_MAX_REFS * sizeof(MeJob) + 255) / 256 * 256;
constexpr size_t kStaticBytes     = kStaticKeysOff + (size_t)SVTME_STATIC_MAX_B64 * SVTME_MAX_LISTS * SVTME_MAX_REFS * SVTME_PU_COUNT * sizeof(unsigned long long);
static_assert(kStaticBytes <= 14u << 20, "the static scratch is meant to stay modest (13.1 MiB for 3840x2160)");
static __device__ __align__(256) unsigned char scratch[kStaticBytes];

constexpr int      kMaxDevices = 64;
static std::mutex  g_static_mu; // guards g_static_ptr and orders the calls that share a device's scratch within this process
static uint8_t    *g_static_ptr[kMaxDevices];

// Device address of this device's copy of `scratch`; the device must be current. Resolved once per device.
static uint8_t *static_scratch_of(int device) {
    if (device < 0 || device >= kMaxDevices) return nullptr;
    std::lock_guard<std::mutex> lock(g_static_mu);
    if (!g_static_ptr[device]) {
        void      *ptr = nullptr;
        hipError_t err = hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch));
        if (err != hipSuccess || !ptr) {
            fprintf(stderr, "svtme: hipGetSymbolAddress(scratch) failed on device %d: %s\n", device, hipGetErrorString(err));
            return nullptr;
        }
        g_static_ptr[device] = static_cast<uint8_t *>(ptr);
    }
    return g_static_ptr[device];
}

struct svtme_engine {
    uint8_t            *static_scratch = nullptr; // this device's copy of `scratch` (device address), set by svtme_create
    svtme_config        cfg{};
    std::mutex          mu;
    std::condition_variable cv_sync_readers; // a slot's last global-motion / TPL reader has left, or a put has been enqueued (with mu)
    hipStream_t         s_store[2] = {nullptr, nullptr}, s_me[4] = {nullptr, nullptr, nullptr, nullptr}, s_copy = nullptr;
    unsigned            next_me = 0, next_upload = 0, n_upload = 1;
    bool                timing_on = false; // per-kernel events on the submissions that follow (svtme_kernel_timing)
    int                 copy_on_search_stream = -1; // -1: with three search streams only; 0 / 1: diagnostic override
    std::vector<Slot>   slots;
    std::vector<Flight> flights;
    // the flights' result blocks, device and pinned host: flight i's at i * res_stride of one allocation each, so that the
    // results of a batch on neighbouring flights can go to the host in one strided copy
    uint8_t            *d_res_arena = nullptr, *h_res_arena = nullptr;
    size_t              res_stride = 0;
    std::vector<BatchRec> batches;
    // the pictures' parameter records: a ring in device memory and its pinned host mirror; a batch takes a run of
    // neighbouring records; the record behind the ring belongs to svtme_picture_search_static
    PicParams          *d_params = nullptr, *h_params = nullptr;
    std::vector<char>   param_busy;
    unsigned            param_head = 0;
    svtme_ticket        next_ticket = 1;
    size_t              slot_bytes = 0, max_b64 = 0;
    uint8_t            *d_upload[2] = {nullptr, nullptr}; // un-padded luma of the pictures being put (a store stream orders the reuse of its buffer)
    size_t              upload_bytes = 0;
    uint8_t            *d_upload_many = nullptr; // the same for svtme_store_put_many: SVTME_MAX_PUT_BATCH pictures (its first call allocates)
    GmState             gm;
    uint64_t            put_gen = 0;
    // the sequence's tiles per 64x64 block (svtme_set_tile_rects): room for max_b64 records, valid for pictures of
    // tiles_w x tiles_h (0: none set)
    svtme_tile_rect    *d_tiles = nullptr;
    uint32_t            tiles_w = 0, tiles_h = 0;
};

namespace {

size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// Makes the engine's device current for the calling thread and puts the thread's previous device back on the way out,
// so that a caller that works on several devices (or a framework that tracks its own current device) is not disturbed.
struct DeviceGuard {
    int  prev = -1;
    bool ok   = false;
    explicit DeviceGuard(int device) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        ok = prev == device || hipSetDevice(device) == hipSuccess;
        if (prev == device) prev = -1; // nothing to put back
    }
    ~DeviceGuard() {
        if (prev >= 0) (void)hipSetDevice(prev);
    }
    DeviceGuard(const DeviceGuard &)            = delete;
    DeviceGuard &operator=(const DeviceGuard &) = delete;
};
#define DEVICE_OR_FAIL(device)                  \
    DeviceGuard device_guard__(device);         \
    if (!device_guard__.ok) return SVTME_ERR_DEVICE

// Lays the three planes of a picture of w x h into one slot allocation.
void layout_planes(uint8_t *mem, uint32_t w, uint32_t h, DevPicture *pic, size_t *bytes) {
    size_t off = 0;
    for (int lvl = 2; lvl >= 0; --lvl) {
        const int      shift = 2 - lvl;
        const uint32_t pw = w >> shift, ph = h >> shift, pad = kPad[lvl];
        const size_t   stride = align_up(pw + 2 * pad, 64);
        DevPlane      &p      = pic->lvl[lvl];
        p.stride = (int32_t)stride;
        p.width  = (int32_t)pw;
        p.height = (int32_t)ph;
        p.pad    = (int32_t)pad;
        // pixel (0,0) on an 8-byte boundary: pad rows, then `pad` rounded up to 8 columns
        const size_t x0 = align_up(pad, 8);
        p.pix = mem ? mem + off + (size_t)pad * stride + x0 : nullptr;
        off += align_up(stride * (ph + 2 * pad) + 256, 256); // slack: staging reads run a few bytes past a window
    }
    *bytes = off;
}

// with_putting: also slots a put has reserved but not yet enqueued (only the store calls themselves want those)
int find_slot(svtme_engine *e, uint64_t key, bool with_putting = false) {
    for (size_t i = 0; i < e->slots.size(); ++i)
        if (e->slots[i].used && e->slots[i].key == key && (with_putting || !e->slots[i].putting)) return (int)i;
    return -1;
}

void free_flight(Flight &f) {
    if (f.d_hme) (void)hipFree(f.d_hme);
    if (f.d_jobs) (void)hipFree(f.d_jobs);
    if (f.d_keys) (void)hipFree(f.d_keys);
    f = Flight(); // (the result blocks are parts of the engine's arenas)
}

} // namespace

extern "C" const char *svtme_version(void) { return "svtme 0.1 (gfx950)"; }

extern "C" int svtme_create(const svtme_config *cfg, svtme_engine **out) {
    if (!cfg || !out || cfg->max_width < 64 || cfg->max_height < 64 || cfg->max_width % 8 || cfg->max_height % 8 ||
        cfg->max_width > 16384 || cfg->max_height > 16384 || cfg->store_slots < 2 || cfg->max_in_flight < 1 || cfg->search_streams > 3 || cfg->kernel_timing > 1 || cfg->serial_hme > 1)
        return SVTME_ERR_BAD_ARG;
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || cfg->device < 0 || cfg->device >= n_dev) {
        fprintf(stderr, "svtme: no HIP device %d (found %d)\n", cfg->device, n_dev);
        return SVTME_ERR_DEVICE;
    }
    DEVICE_OR_FAIL(cfg->device);
    svtme_engine *e = new (std::nothrow) svtme_engine();
    if (!e) return SVTME_ERR_NO_RESOURCES;
    e->cfg = *cfg;
    e->timing_on = cfg->kernel_timing != 0;
    auto fail = [&](int code) {
        svtme_destroy(e);
        return code;
    };
    e->static_scratch = static_scratch_of(cfg->device);
    if (!e->static_scratch) return fail(SVTME_ERR_DEVICE);
    if (const char *v = getenv("SVTME_UPLOAD_RINGS")) e->n_upload = atoi(v) == 2 ? 2 : 1;
    // (s_me[0..2] carry the searches, s_me[3] the allocation-free search and the synchronous global-motion / TPL calls)
    // The order matters. HIP spreads a process's streams over a few hardware queues (GPU_MAX_HW_QUEUES, 4 by default) in
    // the order of their creation, and two streams on one queue take turns: when the copy stream shares a queue with the
    // search stream the result copies no longer run beside the next batch's kernels (measured: 5.1 -> 4.3 k pictures/s at
    // 1080p preset 3, 37 -> 24.5 k at preset 8, 100 -> 71 k at 720p preset 12). Which queue a stream gets depends on how
    // many streams the process created before, so the three streams that work at the same time - search, store, copy - are
    // created one after the other: neighbours in that order get different queues whatever came first (with the former
    // order - store, the four search streams, copy - search and copy were four streams apart and met on one queue as soon
    // as three streams existed before the engine's, e.g. RCCL's).
    {
        hipStream_t *const order[] = {&e->s_me[0], &e->s_store[0], &e->s_copy, &e->s_me[1], &e->s_me[2], &e->s_me[3]};
        for (hipStream_t *st : order)
            if (hipStreamCreateWithFlags(st, hipStreamNonBlocking) != hipSuccess) return fail(SVTME_ERR_DEVICE);
        if (e->n_upload == 2 && hipStreamCreateWithFlags(&e->s_store[1], hipStreamNonBlocking) != hipSuccess) return fail(SVTME_ERR_DEVICE);
    }
    DevPicture tmp;
    layout_planes(nullptr, cfg->max_width, cfg->max_height, &tmp, &e->slot_bytes);
    e->max_b64 = (size_t)((cfg->max_width + 63) / 64) * ((cfg->max_height + 63) / 64);
    // room for rows that arrive with the caller's stride (svtme_store_put_pinned copies them as they lie)
    e->upload_bytes = (size_t)(cfg->max_width + kMaxExtraStride) * cfg->max_height;
    // One upload stream by default: a second one (puts alternating, one DMA overlapping the other put's kernels) raises the
    // put-only rate at 4K from 5.6 k to 6.2 k pictures/s but lowers the rate of searches that run beside the puts (a fifth
    // active stream on four hardware queues). SVTME_UPLOAD_RINGS=2 selects it (read above, before the streams are made).
    if (const char *v = getenv("SVTME_COPY_ON_SEARCH_STREAM")) e->copy_on_search_stream = atoi(v) != 0; // diagnostic
    for (unsigned i = 0; i < e->n_upload; ++i)
        if (hipMalloc((void **)&e->d_upload[i], e->upload_bytes) != hipSuccess) return fail(SVTME_ERR_NO_RESOURCES);
    e->slots.resize(cfg->store_slots);
    for (auto &s : e->slots) {
        if (hipMalloc((void **)&s.mem, e->slot_bytes) != hipSuccess ||
            hipHostMalloc((void **)&s.host_stage, (size_t)cfg->max_width * cfg->max_height, hipHostMallocDefault) != hipSuccess ||
            hipEventCreateWithFlags(&s.ready, hipEventDisableTiming) != hipSuccess)
            return fail(SVTME_ERR_NO_RESOURCES);
    }
    e->flights.resize(cfg->max_in_flight);
    {
        Flight probe;
        probe.layout(e->max_b64, SVTME_MAX_LISTS * SVTME_MAX_REFS, true); // the largest a picture can ask for
        e->res_stride = align_up(probe.res_bytes, 256);
        if (hipMalloc((void **)&e->d_res_arena, e->res_stride * cfg->max_in_flight) != hipSuccess ||
            hipHostMalloc((void **)&e->h_res_arena, e->res_stride * cfg->max_in_flight, hipHostMallocDefault) != hipSuccess)
            return fail(SVTME_ERR_NO_RESOURCES);
    }
    for (size_t fi = 0; fi < e->flights.size(); ++fi) {
        Flight      &f = e->flights[fi];
        const size_t n = e->max_b64 * SVTME_MAX_LISTS * SVTME_MAX_REFS;
        f.layout(e->max_b64, SVTME_MAX_LISTS * SVTME_MAX_REFS, true);
        f.d_res = e->d_res_arena + fi * e->res_stride;
        f.h_res = e->h_res_arena + fi * e->res_stride;
        if (hipMalloc((void **)&f.d_hme, n * sizeof(HmeRefOut) + SVTME_OPS_BYTES) != hipSuccess ||
            hipMalloc((void **)&f.d_jobs, n * sizeof(MeJob)) != hipSuccess ||
            hipMalloc((void **)&f.d_keys, n * SVTME_PU_COUNT * sizeof(unsigned long long)) != hipSuccess)
            return fail(SVTME_ERR_NO_RESOURCES);
        if (hipMemset(f.d_hme, 0, n * sizeof(HmeRefOut) + SVTME_OPS_BYTES) != hipSuccess) return fail(SVTME_ERR_DEVICE);
    }
    e->batches.resize(cfg->max_in_flight);
    for (auto &b : e->batches)
        for (auto &ev : b.ev)
            if (hipEventCreate(&ev) != hipSuccess) return fail(SVTME_ERR_NO_RESOURCES);
    const size_t n_params = 2 * (size_t)cfg->max_in_flight + SVTME_MAX_BATCH;
    e->param_busy.assign(n_params, 0);
    if (hipMalloc((void **)&e->d_params, (n_params + 1) * sizeof(PicParams)) != hipSuccess ||
        hipHostMalloc((void **)&e->h_params, (n_params + 1) * sizeof(PicParams), hipHostMallocDefault) != hipSuccess ||
        hipMalloc((void **)&e->d_tiles, e->max_b64 * sizeof(svtme_tile_rect)) != hipSuccess)
        return fail(SVTME_ERR_NO_RESOURCES);

    *out = e;
    return SVTME_OK;
}

extern "C" void svtme_destroy(svtme_engine *e) {
    if (!e) return;
    DeviceGuard guard(e->cfg.device);
    for (auto sm : e->s_me)
        if (sm) (void)hipStreamSynchronize(sm);
    if (e->s_copy) (void)hipStreamSynchronize(e->s_copy);
    for (auto ss : e->s_store)
        if (ss) (void)hipStreamSynchronize(ss);
    for (auto &s : e->slots) {
        if (s.mem) (void)hipFree(s.mem);
        if (s.host_stage) (void)hipHostFree(s.host_stage);
        if (s.ready) (void)hipEventDestroy(s.ready);
    }
    for (auto d : e->d_upload)
        if (d) (void)hipFree(d);
    if (e->d_upload_many) (void)hipFree(e->d_upload_many);
    for (auto &f : e->flights) free_flight(f);
    if (e->d_res_arena) (void)hipFree(e->d_res_arena);
    if (e->h_res_arena) (void)hipHostFree(e->h_res_arena);
    for (auto &b : e->batches)
        for (auto &ev : b.ev)
            if (ev) (void)hipEventDestroy(ev);
    if (e->d_params) (void)hipFree(e->d_params);
    if (e->h_params) (void)hipHostFree(e->h_params);
    if (e->d_tiles) (void)hipFree(e->d_tiles);
    {
        GmState &g = e->gm;
        void    *dev[] = {g.d_score, g.d_keep, g.d_rows, g.d_corners[0], g.d_corners[1], g.d_match, g.d_block_err, g.d_result, g.d_filter, g.d_lut, g.d_tpl, g.d_rec};
        for (void *d : dev)
            if (d) (void)hipFree(d);
        if (g.h_buf) (void)hipHostFree(g.h_buf);
        if (g.h_tpl) (void)hipHostFree(g.h_tpl);
        if (g.h_rec) (void)hipHostFree(g.h_rec);
        if (g.rec_done) (void)hipEventDestroy(g.rec_done);
        if (g.rec_read) (void)hipEventDestroy(g.rec_read);
    }
    for (auto sm : e->s_me)
        if (sm) (void)hipStreamDestroy(sm);
    for (auto ss : e->s_store)
        if (ss) (void)hipStreamDestroy(ss);
    if (e->s_copy) (void)hipStreamDestroy(e->s_copy);
    delete e;
}

// The two forms of a put. `pinned`: the rows are copied from the caller's buffer as they lie (one linear DMA of
// (height - 1) * stride + width bytes), the store kernels pick the pixels out of the strided copy; the caller keeps the
// buffer unchanged until the planes are built. Otherwise the rows are packed into the slot's pinned staging buffer first
// and the caller's buffer is free on return.
// dev_src: the picture lies in device memory already (rows `stride` apart, complete once `after` has happened): no staging,
// no upload; `done` (optional) is recorded behind the store kernels that read it.
static int store_put_impl(svtme_engine *e, uint64_t key, uint64_t picture_number, const uint8_t *luma, uint32_t stride,
                          uint32_t width, uint32_t height, bool pinned, const uint8_t *dev_src = nullptr, hipEvent_t after = nullptr,
                          hipEvent_t done = nullptr) {
    if (dev_src) luma = dev_src, pinned = true; // (neither is read as host memory below)
    if (!e || !luma || width < 64 || height < 64 || width % 8 || height % 8 || width > e->cfg.max_width ||
        height > e->cfg.max_height || stride < width)
        return SVTME_ERR_BAD_ARG;
    // a strided copy needs dword-aligned rows and has to fit the upload buffer; anything else takes the packing path
    if (pinned && !dev_src && (stride % 4 || stride - width > kMaxExtraStride)) pinned = false;
    DEVICE_OR_FAIL(e->cfg.device);
    int      si;
    bool     fresh, wait_upload;
    unsigned ring;
    {
        std::unique_lock<std::mutex> lock(e->mu);
        // A rewrite is ordered behind the pictures that read the slot through their events. The synchronous readers
        // (svtme_gm_*, svtme_tpl_search) record none: the put waits here until they have returned.
        for (;;) {
            si = find_slot(e, key, true);
            if (si < 0 || e->slots[si].sync_readers == 0) break;
            Slot &w = e->slots[si];
            w.put_waiting++; // (readers that arrive from now on let this put go first)
            e->cv_sync_readers.wait(lock);
            w.put_waiting--;
        }
        fresh = si < 0;
        if (fresh) {
            for (size_t i = 0; i < e->slots.size(); ++i)
                if (!e->slots[i].used) {
                    si = (int)i;
                    break;
                }
            if (si < 0) return SVTME_ERR_NO_RESOURCES;
        }
        Slot &s = e->slots[si];
        if (s.putting) return SVTME_ERR_BAD_ARG; // two puts of one key at the same time
        s.used            = true;
        s.putting         = true; // reserved: other puts leave the slot alone, submits do not see it yet
        s.release_pending = false;
        s.key             = key;
        wait_upload       = s.uploaded;
        ring              = e->next_upload++ % e->n_upload;
    }
    Slot &s = e->slots[si]; // (the vector never grows after svtme_create)
    auto  give_up = [&](int code) {
        std::lock_guard<std::mutex> lock(e->mu);
        s.putting = false;
        if (fresh) s.used = false;
        e->cv_sync_readers.notify_all();
        return code;
    };
    // ---- outside the lock: the slot's staging buffer is free once its previous upload has run, whether or not the slot
    // was in use in between; then the frame copy
    const uint8_t *src        = luma;
    uint32_t       src_stride = stride;
    size_t         bytes      = (size_t)(height - 1) * stride + width;
    if (!pinned) {
        if (wait_upload && hipEventSynchronize(s.ready) != hipSuccess) return give_up(SVTME_ERR_DEVICE);
        if (stride == width) memcpy(s.host_stage, luma, (size_t)width * height); // one large copy: streaming stores
        else
            for (uint32_t y = 0; y < height; ++y) memcpy(s.host_stage + (size_t)y * width, luma + (size_t)y * stride, width);
        src        = s.host_stage;
        src_stride = width;
        bytes      = (size_t)width * height;
    }
    // ---- enqueue: one linear DMA (a pitched copy into the padded plane runs at a quarter of the link's rate), then the
    // store kernels place the rows into the padded planes while they decimate
    std::lock_guard<std::mutex> lock(e->mu);
    const hipStream_t ss = e->s_store[ring];
    hipError_t        err = hipSuccess;
    for (int m = 0; m < 4 && err == hipSuccess; ++m) // rewrite after the readers
        if (s.read_recorded[m]) err = hipStreamWaitEvent(ss, s.last_read[m], 0);
    if (err == hipSuccess && s.uploaded) err = hipStreamWaitEvent(ss, s.ready, 0); // and after the slot's previous build (other ring)
    s.picture_number = picture_number;
    s.width          = width;
    s.height         = height;
    s.gen = ++e->put_gen;
    size_t plane_bytes;
    layout_planes(s.mem, width, height, &s.pic, &plane_bytes);
    if (dev_src) {
        if (err == hipSuccess && after) err = hipStreamWaitEvent(ss, after, 0);
        if (err == hipSuccess) {
            launch_store_build(dev_src, (int)stride, s.pic, 1, ss);
            err = hipGetLastError();
        }
        if (err == hipSuccess && done) err = hipEventRecord(done, ss);
    } else {
        if (err == hipSuccess) err = hipMemcpyAsync(e->d_upload[ring], src, bytes, hipMemcpyHostToDevice, ss);
        if (err == hipSuccess) {
            launch_store_build(e->d_upload[ring], (int)src_stride, s.pic, 1, ss);
            err = hipGetLastError();
        }
    }
    if (err == hipSuccess) err = hipEventRecord(s.ready, ss);
    s.putting = false;
    e->cv_sync_readers.notify_all(); // global-motion / TPL calls that met the slot while it was being put
    if (err != hipSuccess) {
        fprintf(stderr, "svtme: svtme_store_put: %s\n", hipGetErrorString(err));
        (void)hipStreamSynchronize(ss); // nothing of a failed put stays in flight
        if (fresh) s.used = false;
        return SVTME_ERR_DEVICE;
    }
    s.uploaded = true;
    s.ready_seen[0] = s.ready_seen[1] = s.ready_seen[2] = s.ready_seen[3] = false;
    return SVTME_OK;
}

extern "C" int svtme_store_put(svtme_engine *e, uint64_t key, uint64_t picture_number, const uint8_t *luma, uint32_t stride,
                               uint32_t width, uint32_t height) {
    return store_put_impl(e, key, picture_number, luma, stride, width, height, false);
}

extern "C" int svtme_store_put_pinned(svtme_engine *e, uint64_t key, uint64_t picture_number, const uint8_t *luma, uint32_t stride,
                                      uint32_t width, uint32_t height) {
    return store_put_impl(e, key, picture_number, luma, stride, width, height, true);
}

// The n pictures of a mini-GOP in one go: the slots are reserved in one pass under the lock, the frame copies (when the
// rows are not page-locked) run outside it, then n uploads and one launch of each store-building kernel are queued.
extern "C" int svtme_store_put_many(svtme_engine *e, uint32_t n, const uint64_t *keys, const uint64_t *picture_numbers,
                                    const uint8_t *const *lumas, uint32_t stride, uint32_t width, uint32_t height, int pinned) {
    if (!e || !keys || !picture_numbers || !lumas || n < 1 || n > SVTME_MAX_PUT_BATCH || width < 64 || height < 64 || width % 8 ||
        height % 8 || width > e->cfg.max_width || height > e->cfg.max_height || stride < width)
        return SVTME_ERR_BAD_ARG;
    for (uint32_t i = 0; i < n; ++i) {
        if (!lumas[i]) return SVTME_ERR_BAD_ARG;
        for (uint32_t j = 0; j < i; ++j)
            if (keys[j] == keys[i]) return SVTME_ERR_BAD_ARG;
    }
    if (pinned && (stride % 4 || stride - width > kMaxExtraStride)) pinned = 0;
    DEVICE_OR_FAIL(e->cfg.device);
    int  si[SVTME_MAX_PUT_BATCH];
    bool fresh[SVTME_MAX_PUT_BATCH], wait_upload[SVTME_MAX_PUT_BATCH];
    {
        std::unique_lock<std::mutex> lock(e->mu);
        if (!e->d_upload_many && hipMalloc((void **)&e->d_upload_many, e->upload_bytes * SVTME_MAX_PUT_BATCH) != hipSuccess) {
            (void)hipGetLastError();
            return SVTME_ERR_NO_RESOURCES;
        }
        // existing keys first wait for their synchronous readers (see store_put_impl): until none of the keys has any
        for (bool again = true; again;) {
            again = false;
            for (uint32_t i = 0; i < n && !again; ++i) {
                const int s0 = find_slot(e, keys[i], true);
                if (s0 < 0 || e->slots[s0].sync_readers == 0) continue;
                Slot &w = e->slots[s0];
                w.put_waiting++;
                e->cv_sync_readers.wait(lock);
                w.put_waiting--;
                again = true;
            }
        }
        uint32_t got = 0;
        int      rc = SVTME_OK;
        for (; got < n && rc == SVTME_OK; ++got) {
            int s0 = find_slot(e, keys[got], true);
            fresh[got] = s0 < 0;
            if (s0 < 0)
                for (size_t k = 0; k < e->slots.size() && s0 < 0; ++k)
                    if (!e->slots[k].used) s0 = (int)k;
            if (s0 < 0) rc = SVTME_ERR_NO_RESOURCES;
            else if (e->slots[s0].putting) rc = SVTME_ERR_BAD_ARG; // two puts of one key at the same time
            if (rc != SVTME_OK) break;
            Slot &s = e->slots[s0];
            s.used = s.putting = true;
            s.release_pending  = false;
            s.key              = keys[got];
            wait_upload[got]   = s.uploaded;
            si[got]            = s0;
        }
        if (rc != SVTME_OK) {
            for (uint32_t i = 0; i < got; ++i) {
                e->slots[si[i]].putting = false;
                if (fresh[i]) e->slots[si[i]].used = false;
            }
            e->cv_sync_readers.notify_all();
            return rc;
        }
    }
    auto undo = [&](int code) {
        std::lock_guard<std::mutex> lock(e->mu);
        for (uint32_t i = 0; i < n; ++i) {
            e->slots[si[i]].putting = false;
            if (fresh[i]) e->slots[si[i]].used = false;
        }
        e->cv_sync_readers.notify_all();
        return code;
    };
    // ---- outside the lock: the frame copies
    const uint8_t *src[SVTME_MAX_PUT_BATCH];
    uint32_t       src_stride = stride;
    size_t         bytes      = (size_t)(height - 1) * stride + width;
    for (uint32_t i = 0; i < n; ++i) src[i] = lumas[i];
    if (!pinned) {
        for (uint32_t i = 0; i < n; ++i) {
            Slot &s = e->slots[si[i]];
            if (wait_upload[i] && hipEventSynchronize(s.ready) != hipSuccess) return undo(SVTME_ERR_DEVICE);
            if (stride == width) memcpy(s.host_stage, lumas[i], (size_t)width * height);
            else
                for (uint32_t y = 0; y < height; ++y) memcpy(s.host_stage + (size_t)y * width, lumas[i] + (size_t)y * stride, width);
            src[i] = s.host_stage;
        }
        src_stride = width;
        bytes      = (size_t)width * height;
    }
    // ---- enqueue
    std::lock_guard<std::mutex> lock(e->mu);
    const hipStream_t ss = e->s_store[0];
    hipError_t        err = hipSuccess;
    StoreBatch        b;
    memset(&b, 0, sizeof(b));
    for (uint32_t i = 0; i < n && err == hipSuccess; ++i) {
        Slot &s = e->slots[si[i]];
        for (int m = 0; m < 4 && err == hipSuccess; ++m) // rewrite after the readers
            if (s.read_recorded[m]) err = hipStreamWaitEvent(ss, s.last_read[m], 0);
        if (err == hipSuccess && s.uploaded) err = hipStreamWaitEvent(ss, s.ready, 0);
        s.picture_number = picture_numbers[i];
        s.width          = width;
        s.height         = height;
        s.gen            = ++e->put_gen;
        size_t plane_bytes;
        layout_planes(s.mem, width, height, &s.pic, &plane_bytes);
        uint8_t *up = e->d_upload_many + (size_t)i * e->upload_bytes;
        if (err == hipSuccess) err = hipMemcpyAsync(up, src[i], bytes, hipMemcpyHostToDevice, ss);
        b.luma[i]   = up;
        b.stride[i] = (int32_t)src_stride;
        b.pic[i]    = s.pic;
    }
    if (err == hipSuccess) {
        launch_store_build_many(b, (int)n, 1, ss);
        err = hipGetLastError();
    }
    // (one point of the stream marks all of them built; every slot keeps an event of its own)
    for (uint32_t i = 0; i < n && err == hipSuccess; ++i) err = hipEventRecord(e->slots[si[i]].ready, ss);
    for (uint32_t i = 0; i < n; ++i) e->slots[si[i]].putting = false;
    e->cv_sync_readers.notify_all();
    if (err != hipSuccess) {
        fprintf(stderr, "svtme: svtme_store_put_many: %s\n", hipGetErrorString(err));
        (void)hipStreamSynchronize(ss);
        for (uint32_t i = 0; i < n; ++i)
            if (fresh[i]) e->slots[si[i]].used = false;
        return SVTME_ERR_DEVICE;
    }
    for (uint32_t i = 0; i < n; ++i) {
        Slot &s = e->slots[si[i]];
        s.uploaded = true;
        s.ready_seen[0] = s.ready_seen[1] = s.ready_seen[2] = s.ready_seen[3] = false;
    }
    return SVTME_OK;
}

// Device-to-device replication of a built slot from another engine (another GPU): one copy of the slot's three planes.
extern "C" int svtme_store_copy_peer(svtme_engine *dst, svtme_engine *src, uint64_t key) {
    if (!dst || !src || dst == src || dst->slot_bytes != src->slot_bytes) return SVTME_ERR_BAD_ARG;
    // the source slot: held like a synchronous reader for the duration of the copy
    int        ssi;
    Slot       from;
    {
        std::unique_lock<std::mutex> lock(src->mu);
        for (;;) {
            ssi = find_slot(src, key, true);
            if (ssi < 0 || (!src->slots[ssi].putting && src->slots[ssi].put_waiting == 0)) break;
            src->cv_sync_readers.wait(lock);
        }
        if (ssi < 0) return SVTME_ERR_NOT_FOUND;
        Slot &s = src->slots[ssi];
        s.readers++;
        s.sync_readers++;
        from = s; // (descriptor and event handle: plain values)
    }
    auto release_src = [&]() {
        std::lock_guard<std::mutex> lock(src->mu);
        Slot &s = src->slots[ssi];
        if (--s.sync_readers == 0) src->cv_sync_readers.notify_all();
        if (--s.readers == 0 && s.release_pending) {
            s.used            = false;
            s.release_pending = false;
        }
    };
    int rc = SVTME_OK;
    {
        DeviceGuard guard(dst->cfg.device);
        if (!guard.ok) {
            release_src();
            return SVTME_ERR_DEVICE;
        }
        int  di;
        bool fresh;
        {
            std::unique_lock<std::mutex> lock(dst->mu);
            for (;;) {
                di = find_slot(dst, key, true);
                if (di < 0 || dst->slots[di].sync_readers == 0) break;
                Slot &w = dst->slots[di];
                w.put_waiting++;
                dst->cv_sync_readers.wait(lock);
                w.put_waiting--;
            }
            fresh = di < 0;
            if (fresh)
                for (size_t k = 0; k < dst->slots.size() && di < 0; ++k)
                    if (!dst->slots[k].used) di = (int)k;
            if (di < 0) rc = SVTME_ERR_NO_RESOURCES;
            else if (dst->slots[di].putting) rc = SVTME_ERR_BAD_ARG;
            if (rc == SVTME_OK) {
                Slot &d = dst->slots[di];
                d.used = d.putting = true;
                d.release_pending  = false;
                d.key              = key;
            }
        }
        if (rc == SVTME_OK) {
            Slot             &d  = dst->slots[di];
            const hipStream_t ss = dst->s_store[0];
            hipError_t        err = hipSuccess;
            {
                std::lock_guard<std::mutex> lock(dst->mu);
                for (int m = 0; m < 4 && err == hipSuccess; ++m)
                    if (d.read_recorded[m]) err = hipStreamWaitEvent(ss, d.last_read[m], 0);
                if (err == hipSuccess && d.uploaded) err = hipStreamWaitEvent(ss, d.ready, 0);
                if (err == hipSuccess) err = hipStreamWaitEvent(ss, from.ready, 0); // the source planes are built (an event of the other device)
                size_t plane_bytes = 0;
                d.picture_number = from.picture_number;
                d.width          = from.width;
                d.height         = from.height;
                d.gen            = ++dst->put_gen;
                layout_planes(d.mem, d.width, d.height, &d.pic, &plane_bytes);
                if (err == hipSuccess)
                    err = hipMemcpyPeerAsync(d.mem, dst->cfg.device, from.mem, src->cfg.device, plane_bytes, ss);
                if (err == hipSuccess) err = hipEventRecord(d.ready, ss);
            }
            if (err == hipSuccess) err = hipEventSynchronize(d.ready); // outside the lock: the source slot is free after this
            std::lock_guard<std::mutex> lock(dst->mu);
            d.putting = false;
            dst->cv_sync_readers.notify_all();
            if (err != hipSuccess) {
                fprintf(stderr, "svtme: svtme_store_copy_peer: %s\n", hipGetErrorString(err));
                (void)hipStreamSynchronize(ss);
                (void)hipGetLastError();
                if (fresh) d.used = false;
                rc = SVTME_ERR_DEVICE;
            } else {
                d.uploaded = true;
                d.ready_seen[0] = d.ready_seen[1] = d.ready_seen[2] = d.ready_seen[3] = false;
            }
        }
    }
    release_src();
    return rc;
}

extern "C" int svtme_store_wait(svtme_engine *e, uint64_t key) {
    if (!e) return SVTME_ERR_BAD_ARG;
    hipEvent_t ready;
    {
        std::lock_guard<std::mutex> lock(e->mu);
        const int si = find_slot(e, key);
        if (si < 0) return SVTME_ERR_NOT_FOUND;
        ready = e->slots[si].ready;
    }
    DEVICE_OR_FAIL(e->cfg.device);
    HIP_TRY(hipEventSynchronize(ready));
    return SVTME_OK;
}

extern "C" int svtme_host_register(svtme_engine *e, void *ptr, size_t bytes) {
    if (!e || !ptr || !bytes) return SVTME_ERR_BAD_ARG;
    DEVICE_OR_FAIL(e->cfg.device);
    const hipError_t err = hipHostRegister(ptr, bytes, hipHostRegisterDefault);
    if (err == hipErrorHostMemoryAlreadyRegistered) {
        (void)hipGetLastError();
        return SVTME_OK;
    }
    if (err != hipSuccess) { // e.g. a virtual machine whose driver cannot lock user pages: use svtme_host_alloc there
        (void)hipGetLastError();
        return SVTME_ERR_UNSUPPORTED;
    }
    return SVTME_OK;
}

extern "C" int svtme_host_alloc(svtme_engine *e, size_t bytes, void **out) {
    if (!e || !bytes || !out) return SVTME_ERR_BAD_ARG;
    DEVICE_OR_FAIL(e->cfg.device);
    void *ptr = nullptr;
    if (hipHostMalloc(&ptr, bytes, hipHostMallocPortable) != hipSuccess || !ptr) { // (page-locked for every GPU: a group's engines all read it)
        (void)hipGetLastError();
        return SVTME_ERR_NO_RESOURCES;
    }
    *out = ptr;
    return SVTME_OK;
}

extern "C" int svtme_host_free(svtme_engine *e, void *ptr) {
    if (!e || !ptr) return SVTME_ERR_BAD_ARG;
    DEVICE_OR_FAIL(e->cfg.device);
    HIP_TRY(hipHostFree(ptr));
    return SVTME_OK;
}

extern "C" int svtme_host_unregister(svtme_engine *e, void *ptr) {
    if (!e || !ptr) return SVTME_ERR_BAD_ARG;
    DEVICE_OR_FAIL(e->cfg.device);
    HIP_TRY(hipHostUnregister(ptr));
    return SVTME_OK;
}

extern "C" int svtme_store_release(svtme_engine *e, uint64_t key) {
    if (!e) return SVTME_ERR_BAD_ARG;
    std::lock_guard<std::mutex> lock(e->mu);
    const int si = find_slot(e, key);
    if (si < 0) return SVTME_ERR_NOT_FOUND;
    Slot &s = e->slots[si];
    if (s.readers > 0) s.release_pending = true;
    else s.used = false;
    return SVTME_OK;
}

extern "C" int svtme_store_read(svtme_engine *e, uint64_t key, int level, uint8_t *dst, size_t dst_size, uint32_t *stride,
                                uint32_t *width, uint32_t *height, uint32_t *pad) {
    if (!e || level < 0 || level > 2 || !dst) return SVTME_ERR_BAD_ARG;
    std::lock_guard<std::mutex> lock(e->mu);
    DEVICE_OR_FAIL(e->cfg.device);
    const int si = find_slot(e, key);
    if (si < 0) return SVTME_ERR_NOT_FOUND;
    const DevPlane &p  = e->slots[si].pic.lvl[level];
    const uint32_t  os = (uint32_t)(p.width + 2 * p.pad), rows = (uint32_t)(p.height + 2 * p.pad);
    if (dst_size < (size_t)os * rows) return SVTME_ERR_BAD_ARG;
    HIP_TRY(hipEventSynchronize(e->slots[si].ready));
    HIP_TRY(hipMemcpy2D(dst, os, p.pix - (ptrdiff_t)p.pad * p.stride - p.pad, p.stride, os, rows, hipMemcpyDeviceToHost));
    if (stride) *stride = os;
    if (width) *width = (uint32_t)p.width;
    if (height) *height = (uint32_t)p.height;
    if (pad) *pad = (uint32_t)p.pad;
    return SVTME_OK;
}

extern "C" int svtme_set_tile_rects(svtme_engine *e, uint32_t width, uint32_t height, const svtme_tile_rect *rects, uint32_t n_b64) {
    if (!e) return SVTME_ERR_BAD_ARG;
    if (rects) {
        if (width < 64 || height < 64 || width > e->cfg.max_width || height > e->cfg.max_height ||
            n_b64 != ((width + 63) / 64) * ((height + 63) / 64))
            return SVTME_ERR_BAD_ARG;
        // the bounds keep every window the plan kernel derives from them on the padded planes
        for (uint32_t i = 0; i < n_b64; ++i)
            if (rects[i].start_x > width || rects[i].end_x > width || rects[i].start_y > height || rects[i].end_y > height)
                return SVTME_ERR_BAD_ARG;
    }
    std::lock_guard<std::mutex> lock(e->mu);
    DEVICE_OR_FAIL(e->cfg.device);
    // pictures in flight may still read the table
    for (auto sm : e->s_me)
        if (sm) HIP_TRY(hipStreamSynchronize(sm));
    e->tiles_w = e->tiles_h = 0;
    if (!rects) return SVTME_OK;
    HIP_TRY(hipMemcpy(e->d_tiles, rects, (size_t)n_b64 * sizeof(svtme_tile_rect), hipMemcpyHostToDevice));
    e->tiles_w = width;
    e->tiles_h = height;
    return SVTME_OK;
}

// What both submit forms check of one picture description before anything is touched.
static int check_picture(const svtme_picture *pic) {
    const svtme_controls &k = pic->ctrl;
    if (pic->num_of_list_to_search < 1 || pic->num_of_list_to_search > 2 || pic->num_of_ref_pic_to_search[0] < 1 ||
        pic->num_of_ref_pic_to_search[0] > SVTME_MAX_REFS ||
        (pic->num_of_list_to_search == 2 && (pic->num_of_ref_pic_to_search[1] < 1 || pic->num_of_ref_pic_to_search[1] > SVTME_MAX_REFS)) ||
        k.num_hme_sa_w < 1 || k.num_hme_sa_w > 2 || k.num_hme_sa_h < 1 || k.num_hme_sa_h > 2)
        return SVTME_ERR_BAD_ARG;
    // not implemented on the device: full-SAD integer search (nothing in the encoder selects it)
    if (k.me_search_method != 0) return SVTME_ERR_UNSUPPORTED;
    if (pic->tail.enable > 2 ||
        (pic->me_type != 0 && pic->me_type != SVTME_ME_OPEN_LOOP && pic->me_type != SVTME_ME_MCTF && pic->me_type != SVTME_ME_FIRST_PASS) ||
        pic->bypass_zz_check > 1 || pic->restricted_motion_vector > 1 || pic->b64_step > 8)
        return SVTME_ERR_BAD_ARG;
    // the temporal filter searches one neighbour frame at a time and has no host tail (EbTemporalFiltering.c:4844-4847,
    // EbMotionEstimation.c:2977)
    if (pic->me_type == SVTME_ME_MCTF &&
        (pic->num_of_list_to_search != 1 || pic->num_of_ref_pic_to_search[0] != 1 || pic->tail.enable))
        return SVTME_ERR_BAD_ARG;
    if (pic->tail.enable &&
        (pic->tail.max_refs > SVTME_TAIL_MV_STRIDE || pic->tail.max_cand > SVTME_TAIL_CAND_STRIDE || pic->tail.max_cand < 1 ||
         // me_mv_array holds list 0 at [0, n_ref[0]) and list 1 at [max_l0, max_l0 + n_ref[1]) of every partition's max_refs
         // entries (the first pass runs two list-0 references with max_l0 = 1 and no list 1, firstpass.c:1273-1276)
         pic->tail.max_refs < pic->num_of_ref_pic_to_search[0] ||
         (pic->num_of_list_to_search == 2 &&
          (pic->tail.max_l0 < pic->num_of_ref_pic_to_search[0] || pic->tail.max_refs < pic->tail.max_l0 + pic->num_of_ref_pic_to_search[1]))))
        return SVTME_ERR_BAD_ARG;
    return SVTME_OK;
}

// The part of a picture's parameter record that follows from its description and the store: planes, reference lists,
// distances, controls. `used` receives the store slots the picture reads (at most 1 + 2 x 4).
static int fill_params(svtme_engine *e, const svtme_picture *pic, PicParams &p, int *used, int *n_used) {
    const svtme_controls &k = pic->ctrl;
    memset(&p, 0, sizeof(p));
    *n_used = 0;
    const int ci = find_slot(e, pic->cur_key);
    if (ci < 0) return SVTME_ERR_NOT_FOUND;
    const Slot &cs = e->slots[ci];
    if (cs.width != pic->width || cs.height != pic->height) return SVTME_ERR_BAD_ARG;
    p.cur = cs.pic;
    used[(*n_used)++] = ci;
    p.n_list   = pic->num_of_list_to_search;
    p.n_ref[0] = pic->num_of_ref_pic_to_search[0];
    p.n_ref[1] = p.n_list == 2 ? pic->num_of_ref_pic_to_search[1] : 0;
    for (int l = 0; l < p.n_list; ++l)
        for (int r = 0; r < p.n_ref[l]; ++r) {
            const int ri = find_slot(e, pic->ref_key[l][r]);
            if (ri < 0) return SVTME_ERR_NOT_FOUND;
            if (e->slots[ri].width != pic->width || e->slots[ri].height != pic->height) return SVTME_ERR_BAD_ARG;
            p.ref[l][r] = e->slots[ri].pic;
            const int64_t d = (int64_t)pic->picture_number - (int64_t)pic->ref_picture_number[l][r];
            p.dist[l][r] = (uint16_t)(int16_t)(d < 0 ? -d : d);
            used[(*n_used)++] = ri;
        }
    p.ctrl = k;
    p.width  = (int32_t)pic->width;
    p.height = (int32_t)pic->height;
    p.b64_w  = (int32_t)((pic->width + 63) / 64);
    p.b64_h  = (int32_t)((pic->height + 63) / 64);
    p.b64_w_magic = (uint32_t)((0x100000000ull + (uint32_t)p.b64_w - 1) / (uint32_t)p.b64_w);
    p.temporal_layer = pic->temporal_layer_index;
    p.is_ref = pic->is_used_as_reference_flag;
    p.me_type       = pic->me_type ? pic->me_type : SVTME_ME_OPEN_LOOP;
    p.bypass_zz_check = pic->me_type == SVTME_ME_FIRST_PASS && pic->bypass_zz_check;
    p.b64_step        = pic->b64_step;
    p.restricted      = pic->restricted_motion_vector;
    // the open-loop search reads the picture's own block parameters, whose one tile is the picture; the temporal filter
    // and the first pass read the sequence's (EbMotionEstimation.c:1408-1410)
    if (p.restricted && p.me_type != SVTME_ME_OPEN_LOOP && e->tiles_w) {
        if (e->tiles_w != pic->width || e->tiles_h != pic->height) return SVTME_ERR_BAD_ARG;
        p.tiles = e->d_tiles;
    }
    p.tf_me_exit_th = pic->tf_me_exit_th;
    // a reference depends on an earlier one: pre-HME list-1 early exit (:1758-1768) and the HME level-0 area
    // rescaling from the first reference's vector (:1837-1851): those items run in a second launch
    p.dep_pre = k.prehme_enable && k.prehme_l1_early_exit && p.n_list == 2;
    p.dep_l0  = k.enable_hme_flag && k.enable_hme_level0_flag && k.enable_me_sr_adjustment && k.distance_based_hme_resizing &&
        k.reduce_hme_l0_sr_th_min && k.reduce_hme_l0_sr_th_max && p.n_ref[0] + p.n_ref[1] > 1;
    p.serial_refs = e->cfg.serial_hme != 0;
    {
        // search work items per (b64, ref): the most any window the controls allow can need (integer_search_b64
        // :1319-1345 caps the window at sa_max, rounds the width up to 8 and keeps the height >= 3; clipping at the
        // picture's edge may leave any multiple of 8 below that, or fewer than 8 columns)
        const int w_max = (std::max<int>(k.me_sa.sa_max.width, 1) + 7) & ~7, h_max = std::max<int>(k.me_sa.sa_max.height, 3);
        int       items = 1;
        for (int w = 1; w <= w_max; w += (w < 8 ? 1 : 8)) items = std::max(items, me_item_count(w, h_max));
        p.me_max_items = items;
    }
    return SVTME_OK;
}

// Where a picture's kernels leave their results: `res` is the result block (device memory, or memory of the caller's /
// pinned host memory the device writes), laid out by `f`.
static void point_results(PicParams &p, const Flight &f, uint8_t *res, const svtme_tail &tail) {
    p.best_sad       = reinterpret_cast<uint32_t *>(res + f.off_sad);
    p.best_mv        = reinterpret_cast<uint32_t *>(res + f.off_mv);
    p.search_results = reinterpret_cast<svtme_search_result *>(res + f.off_sr);
    p.zz_sad         = reinterpret_cast<uint32_t *>(res + f.off_zz);
    p.sr_divisor     = reinterpret_cast<uint32_t *>(res + f.off_div);
    p.ops_out        = reinterpret_cast<unsigned long long *>(res + f.off_ops);
    if (tail.enable) {
        p.tail         = tail;
        p.t_hme_sad    = reinterpret_cast<unsigned long long *>(res + f.off_thme);
        p.t_mv_array   = reinterpret_cast<uint32_t *>(res + f.off_tmv);
        p.t_distortion = reinterpret_cast<uint32_t *>(res + f.off_tdist);
        p.t_stats      = reinterpret_cast<uint32_t *>(res + f.off_tstat);
        p.t_cand_array = res + f.off_tcand;
        p.t_total_cand = res + f.off_ttot;
        p.t_do_ref     = res + f.off_tdo;
        p.t_zero_begin = res + f.off_thme;
        p.t_zero_bytes = f.tail_end - f.off_thme;
    }
}

extern "C" int svtme_pictures_submit(svtme_engine *e, const svtme_picture *pics, uint32_t n, svtme_ticket *tickets) {
    if (!e || !pics || !tickets || n < 1 || n > SVTME_MAX_BATCH) return SVTME_ERR_BAD_ARG;
    for (uint32_t i = 0; i < n; ++i) {
        const int rc = check_picture(&pics[i]);
        if (rc != SVTME_OK) return rc;
        // one set of launches covers the batch: one size, one set of controls, one kind of search
        const svtme_picture &a = pics[0], &b = pics[i];
        if (a.width != b.width || a.height != b.height || memcmp(&a.ctrl, &b.ctrl, sizeof(a.ctrl)) || memcmp(&a.tail, &b.tail, sizeof(a.tail)) ||
            (a.me_type ? a.me_type : SVTME_ME_OPEN_LOOP) != (b.me_type ? b.me_type : SVTME_ME_OPEN_LOOP) ||
            a.bypass_zz_check != b.bypass_zz_check || a.restricted_motion_vector != b.restricted_motion_vector || a.b64_step != b.b64_step ||
            a.tf_me_exit_th != b.tf_me_exit_th)
            return SVTME_ERR_BAD_ARG;
    }
    std::lock_guard<std::mutex> lock(e->mu);
    DEVICE_OR_FAIL(e->cfg.device);
    // ---- resources: a flight per picture, a batch record, a run of parameter records
    Flight *fl[SVTME_MAX_BATCH];
    {
        // neighbouring flights when there are n free ones in a row (their result blocks then lie one stride apart), else
        // any n free ones
        uint32_t got = 0;
        for (size_t i = 0, run = 0; i < e->flights.size() && got < n; ++i) {
            run = e->flights[i].busy ? 0 : run + 1;
            if (run == n)
                for (uint32_t j = 0; j < n; ++j) fl[got++] = &e->flights[i + 1 - n + j];
        }
        if (got < n) {
            got = 0;
            for (auto &c : e->flights)
                if (!c.busy && got < n) fl[got++] = &c;
        }
        if (got < n) return SVTME_ERR_NO_RESOURCES;
    }
    int bi = -1;
    for (size_t i = 0; i < e->batches.size(); ++i)
        if (!e->batches[i].used) {
            bi = (int)i;
            break;
        }
    if (bi < 0) return SVTME_ERR_NO_RESOURCES;
    BatchRec &B = e->batches[bi];
    int       run = -1;
    {
        const unsigned cap = (unsigned)e->param_busy.size();
        for (unsigned tries = 0, at = e->param_head; tries < cap && run < 0; ++tries, at = (at + 1) % cap) {
            if (at + n > cap) continue; // a run does not wrap
            bool free_run = true;
            for (unsigned j = 0; j < n && free_run; ++j) free_run = !e->param_busy[at + j];
            if (free_run) run = (int)at;
        }
        if (run < 0) return SVTME_ERR_NO_RESOURCES;
    }
    // ---- the pictures' parameter records
    PicParams *hp = e->h_params + run; // pinned: the records go to the device from here
    int        used[SVTME_MAX_BATCH][1 + SVTME_MAX_LISTS * SVTME_MAX_REFS], n_used[SVTME_MAX_BATCH];
    const bool tail = pics[0].tail.enable != 0, skip_arrays = tail && pics[0].tail.enable == 2;
    for (uint32_t i = 0; i < n; ++i) {
        PicParams &p  = hp[i];
        const int  rc = fill_params(e, &pics[i], p, used[i], &n_used[i]);
        if (rc != SVTME_OK) return rc;
        Flight      *f     = fl[i];
        const size_t n_b64 = (size_t)p.b64_w * p.b64_h;
        if (n_b64 > e->max_b64) return SVTME_ERR_BAD_ARG;
        f->layout(n_b64, (size_t)(p.n_ref[0] + p.n_ref[1]), tail); // never larger than the allocation (n_b64 <= max_b64)
        f->n_ref[0] = (uint32_t)p.n_ref[0];
        f->n_ref[1] = (uint32_t)p.n_ref[1];
        f->n_b64    = (uint32_t)n_b64;
        p.hme_out   = f->d_hme;
        p.me_jobs   = f->d_jobs;
        p.me_keys   = f->d_keys;
        // the counters sit right behind this picture's HME records
        p.ops = reinterpret_cast<unsigned long long *>(reinterpret_cast<uint8_t *>(f->d_hme) + n_b64 * SVTME_MAX_LISTS * SVTME_MAX_REFS * sizeof(HmeRefOut));
        point_results(p, *f, f->d_res, pics[i].tail);
    }

    // search_streams 0 (the default): pictures submitted one at a time take turns on three streams, so that one picture's
    // kernel tails overlap the others' kernels; the pictures of a batch already fill the GPU between them and all batches
    // go down one stream (measured with batches of 4-16: 4-25 % faster than three streams, and without their two regimes)
    const unsigned    n_sm = e->cfg.search_streams ? e->cfg.search_streams : (n > 1 ? 1 : 3);
    const unsigned    mi = n_sm == 1 ? 0 : e->next_me++ % n_sm;
    const hipStream_t sm = e->s_me[mi];
    // results back to pinned host memory on the copy stream, so that the next picture's kernels do not wait for PCIe
    // (with three search streams the copy stays on the picture's own stream: the other two keep the GPU busy, and
    // the process stays within four hardware queues)
    const hipStream_t sc = (e->copy_on_search_stream < 0 ? n_sm >= 3 : e->copy_on_search_stream != 0) ? sm : e->s_copy;
    const bool  kt = e->timing_on;
    // ---- everything the batch puts on its streams. (The HME records and the counters are zero on entry: allocated so,
    // and every picture's last search kernel clears them again.)
    auto enqueue = [&]() -> hipError_t {
        hipError_t err;
        for (uint32_t i = 0; i < n; ++i)
            for (int j = 0; j < n_used[i]; ++j) {
                Slot &sl = e->slots[used[i][j]];
                if (!sl.ready_seen[mi]) { // later pictures on this stream are ordered behind this wait anyway
                    if ((err = hipStreamWaitEvent(sm, sl.ready, 0)) != hipSuccess) return err;
                    sl.ready_seen[mi] = true;
                }
            }
        if ((err = hipEventRecord(B.ev[0], sm)) != hipSuccess) return err;
        if (n > 1) launch_params_write(hp, e->d_params + run, (int)n, sm); // (a single picture's record travels as kernel argument)
        const Batch batch{hp, e->d_params + run, (int)n};
        hipEvent_t  hme_marks[2] = {kt ? B.ev[1] : nullptr, kt ? B.ev[2] : nullptr};
        // (the last ME mark is recorded after the optional tail kernel, so that it ends the batch's kernels either way)
        hipEvent_t me_marks[3] = {kt ? B.ev[3] : nullptr, kt ? B.ev[4] : nullptr, tail ? nullptr : B.ev[5]};
        launch_hme(batch, sm, hme_marks);
        launch_me(batch, sm, me_marks);
        if (tail) {
            launch_tail(batch, sm);
            if ((err = hipEventRecord(B.ev[5], sm)) != hipSuccess) return err;
        }
        if ((err = hipGetLastError()) != hipSuccess) return err;
        // (kernels that write their results straight into the pinned host block were measured against these copies: slower
        // at 1080p and 4K - the finishing kernel then runs at the link's pace -, faster only at 720p)
        if (sc != sm && (err = hipStreamWaitEvent(sc, B.ev[5], 0)) != hipSuccess) return err;
        // tail.enable == 2: the host wants finished results only, the best SAD / MV planes (the end of the block) stay on
        // the device
        auto bytes_of = [&](uint32_t i) { return skip_arrays ? fl[i]->tail_end : fl[i]->res_bytes; };
        // Pictures on neighbouring flights that bring back equally much (the same number of references: low-delay P
        // pictures, temporal-filter pairs, the first pass) go to the host in one strided copy, a row per picture: a copy
        // call costs the submitting thread several microseconds, which is what bounds the smallest pictures.
        static const bool no_strided = getenv("SVTME_NO_STRIDED_COPY") != nullptr; // diagnostic
        for (uint32_t i = 0; i < n;) {
            uint32_t j = i + 1;
            while (!no_strided && j < n && fl[j] == fl[j - 1] + 1 && bytes_of(j) == bytes_of(i)) ++j;
            if (j - i > 1)
                err = hipMemcpy2DAsync(fl[i]->h_res, e->res_stride, fl[i]->d_res, e->res_stride, bytes_of(i), j - i, hipMemcpyDeviceToHost, sc);
            else
                err = hipMemcpyAsync(fl[i]->h_res, fl[i]->d_res, bytes_of(i), hipMemcpyDeviceToHost, sc);
            if (err != hipSuccess) return err;
            i = j;
        }
        return hipEventRecord(B.ev[6], sc);
    };
    const hipError_t enq_err = enqueue();
    if (enq_err != hipSuccess) {
        // Part of the batch may be on its stream already. The flights and the slots are not accounted for yet, so nothing
        // may stay in flight: drain the streams, then restore what the next picture expects of these buffers (zeroed HME
        // records and counters), and report.
        fprintf(stderr, "svtme: svtme_pictures_submit: %s\n", hipGetErrorString(enq_err));
        (void)hipStreamSynchronize(sm);
        if (sc != sm) (void)hipStreamSynchronize(sc);
        for (uint32_t i = 0; i < n; ++i)
            (void)hipMemset(fl[i]->d_hme, 0, e->max_b64 * SVTME_MAX_LISTS * SVTME_MAX_REFS * sizeof(HmeRefOut) + SVTME_OPS_BYTES);
        (void)hipGetLastError();
        return SVTME_ERR_DEVICE;
    }
    B.used          = true;
    B.kernel_timing = kt;
    B.n = B.members_left = (int)n;
    B.stream_index  = mi;
    e->param_head   = (unsigned)(run + n) % (unsigned)e->param_busy.size();
    for (uint32_t i = 0; i < n; ++i) {
        Flight *f = fl[i];
        for (int j = 0; j < n_used[i]; ++j) { // ev[5] was recorded behind the batch's last kernel
            Slot &sl = e->slots[used[i][j]];
            sl.last_read[mi]     = B.ev[5];
            sl.read_recorded[mi] = true;
            sl.readers++;
        }
        f->slots.assign(used[i], used[i] + n_used[i]);
        f->tail               = tail;
        f->skip_search_arrays = skip_arrays;
        f->batch              = bi;
        f->param              = run + (int)i;
        e->param_busy[run + i] = 1;
        f->busy   = true;
        f->waited = false;
        f->ticket = e->next_ticket++;
        tickets[i] = f->ticket;
    }
    return SVTME_OK;
}

extern "C" int svtme_picture_submit(svtme_engine *e, const svtme_picture *pic, svtme_ticket *ticket) {
    return svtme_pictures_submit(e, pic, 1, ticket);
}

// The allocation-free form of svtme_picture_submit + svtme_picture_wait: same kernels, same results, but the kernels'
// intermediates live in this device's copy of the file-scope `scratch` array and the results go straight into memory
// the caller owns. Nothing here allocates or frees (no hipMalloc / hipFree / hipHostMalloc, no event or stream
// creation, no heap containers): the stream is one svtme_create made, the store planes are the slots svtme_create made.
extern "C" int svtme_picture_search_static(svtme_engine *e, const svtme_picture *pic, void *result_block, size_t result_bytes,
                                           svtme_results *out) {
    if (!e || !pic || !result_block || !out) return SVTME_ERR_BAD_ARG;
    const int rc0 = check_picture(pic);
    if (rc0 != SVTME_OK) return rc0;
    if (pic->tail.enable) return SVTME_ERR_UNSUPPORTED; // the device tail's outputs are not part of this form's result block
    if (e->cfg.serial_hme) return SVTME_ERR_UNSUPPORTED;
    if (reinterpret_cast<uintptr_t>(result_block) % 64) return SVTME_ERR_BAD_ARG;
    {
        // the kernels write into the block: it has to be device memory or page-locked host memory (pageable memory would
        // be a GPU memory fault, not an error code). hipPointerGetAttributes allocates nothing.
        DeviceGuard guard(e->cfg.device);
        if (!guard.ok) return SVTME_ERR_DEVICE;
        hipPointerAttribute_t attr;
        memset(&attr, 0, sizeof(attr));
        const hipError_t perr = hipPointerGetAttributes(&attr, result_block);
        if (perr != hipSuccess || (attr.type != hipMemoryTypeDevice && attr.type != hipMemoryTypeHost && attr.type != hipMemoryTypeManaged)) {
            (void)hipGetLastError();
            fprintf(stderr, "svtme: svtme_picture_search_static: result_block is neither device memory nor page-locked host memory\n");
            return SVTME_ERR_BAD_ARG;
        }
    }
    const size_t n_b64 = (size_t)((pic->width + 63) / 64) * ((pic->height + 63) / 64);
    if (n_b64 > SVTME_STATIC_MAX_B64) {
        fprintf(stderr, "svtme: svtme_picture_search_static takes pictures of at most %d 64x64 blocks (3840x2160); %ux%u has %zu\n",
                SVTME_STATIC_MAX_B64, pic->width, pic->height, n_b64);
        return SVTME_ERR_BAD_ARG;
    }
    const int    n_ref0 = pic->num_of_ref_pic_to_search[0], n_ref1 = pic->num_of_list_to_search == 2 ? pic->num_of_ref_pic_to_search[1] : 0;
    if (result_bytes < SVTME_STATIC_RESULT_BYTES(n_b64, (size_t)(n_ref0 + n_ref1))) return SVTME_ERR_BAD_ARG;

    std::lock_guard<std::mutex> lock(e->mu);
    DEVICE_OR_FAIL(e->cfg.device);
    // calls that share this device's scratch within the process take turns
    std::lock_guard<std::mutex> scratch_turn(g_static_mu);
    // the picture's parameter record: the one behind the ring, in pinned host memory svtme_create allocated
    const size_t pi = e->param_busy.size();
    PicParams   &p  = e->h_params[pi];
    int          used[1 + SVTME_MAX_LISTS * SVTME_MAX_REFS], n_used = 0;
    const int    rc = fill_params(e, pic, p, used, &n_used);
    if (rc != SVTME_OK) return rc;
    // intermediates: this device's copy of the static array, through the pointer taken with hipGetSymbolAddress
    uint8_t *sc = e->static_scratch;
    p.hme_out = reinterpret_cast<HmeRefOut *>(sc);
    p.ops     = reinterpret_cast<unsigned long long *>(sc + kStaticOpsOff);
    p.me_jobs = reinterpret_cast<MeJob *>(sc + kStaticJobsOff);
    p.me_keys = reinterpret_cast<unsigned long long *>(sc + kStaticKeysOff);
    // results: the caller's block, laid out like a picture without device tail
    Flight lay;
    lay.layout(n_b64, (size_t)(n_ref0 + n_ref1), false);
    uint8_t *res = static_cast<uint8_t *>(result_block);
    point_results(p, lay, res, pic->tail);

    // The array keeps the last call's contents, so what the kernels expect to start from zero (HME records, counters) is
    // cleared here whatever the last call left.
    // (created by svtme_create; the global-motion and TPL calls queue on the same stream under their own lock: a static
    // search and such a call take turns on it, in the order they were queued)
    const hipStream_t sm = e->s_me[3];
    for (int i = 0; i < n_used; ++i) HIP_TRY(hipStreamWaitEvent(sm, e->slots[used[i]].ready, 0));
    HIP_TRY(hipMemsetAsync(sc, 0, n_b64 * SVTME_MAX_LISTS * SVTME_MAX_REFS * sizeof(HmeRefOut), sm));
    HIP_TRY(hipMemsetAsync(sc + kStaticOpsOff, 0, SVTME_OPS_BYTES, sm));
    const Batch batch{&p, e->d_params + pi, 1};
    hipEvent_t  no_marks[3] = {nullptr, nullptr, nullptr};
    launch_hme(batch, sm, no_marks);
    launch_me(batch, sm, no_marks);
    const hipError_t launch_err = hipGetLastError();
    // the planes must not be rewritten and the scratch not reused before the kernels are through: always drain the stream
    const hipError_t sync_err = hipStreamSynchronize(sm);
    if (launch_err != hipSuccess || sync_err != hipSuccess) {
        fprintf(stderr, "svtme: svtme_picture_search_static: %s\n", hipGetErrorString(launch_err != hipSuccess ? launch_err : sync_err));
        return SVTME_ERR_DEVICE;
    }
    {
        // the search kernels flag a work-item grid smaller than a window needs (a sizing bug, never a data condition), as
        // svtme_picture_wait reports it: 8 bytes of the block's counter area, read back whatever memory the block is
        unsigned long long flag[SVTME_OPS_SLOTS];
        HIP_TRY(hipMemcpy2D(flag, 8, res + lay.off_ops + 7 * 8, 64, 8, SVTME_OPS_SLOTS, hipMemcpyDefault));
        for (int s = 0; s < SVTME_OPS_SLOTS; ++s)
            if (flag[s]) return SVTME_ERR_DEVICE;
    }
    out->n_b64                = (uint32_t)n_b64;
    out->p_sb_best_sad        = reinterpret_cast<const uint32_t *>(res + lay.off_sad);
    out->p_sb_best_mv         = reinterpret_cast<const uint32_t *>(res + lay.off_mv);
    out->search_results       = reinterpret_cast<const svtme_search_result *>(res + lay.off_sr);
    out->zz_sad               = reinterpret_cast<const uint32_t *>(res + lay.off_zz);
    out->reduce_me_sr_divisor = reinterpret_cast<const uint32_t *>(res + lay.off_div);
    out->total_me_candidate_index = nullptr;
    out->me_mv_array = nullptr;
    out->me_candidate_array = nullptr;
    out->me_distortion = nullptr;
    out->b64_stats = nullptr;
    out->final_do_ref = nullptr;
    out->final_hme_sad = nullptr;
    out->n_ref[0] = (uint32_t)n_ref0;
    out->n_ref[1] = (uint32_t)n_ref1;
    return SVTME_OK;
}

static Flight *find_flight(svtme_engine *e, svtme_ticket t) {
    for (auto &f : e->flights)
        if (f.busy && f.ticket == t) return &f;
    return nullptr;
}

extern "C" int svtme_picture_wait(svtme_engine *e, svtme_ticket ticket, svtme_results *out) {
    if (!e || !out) return SVTME_ERR_BAD_ARG;
    Flight *f;
    {
        std::lock_guard<std::mutex> lock(e->mu);
        f = find_flight(e, ticket);
        if (!f) return SVTME_ERR_NOT_FOUND;
    }
    DEVICE_OR_FAIL(e->cfg.device);
    BatchRec &B = e->batches[f->batch]; // (stays the picture's batch until svtme_picture_done)
    HIP_TRY(hipEventSynchronize(B.ev[6]));
    std::lock_guard<std::mutex> lock(e->mu);
    if (!f->waited) {
        f->waited = true;
        for (int si : f->slots) {
            Slot &s = e->slots[si];
            if (s.last_read[B.stream_index] == B.ev[5]) s.read_recorded[B.stream_index] = false; // that read is over
            if (--s.readers == 0 && s.release_pending) {
                s.used = false;
                s.release_pending = false;
            }
        }
    }
    {
        // the search kernels flag a work-item grid smaller than a window needs (a sizing bug, never a data condition)
        const uint64_t *slots = reinterpret_cast<const uint64_t *>(f->h_res + f->off_ops);
        for (int s = 0; s < SVTME_OPS_SLOTS; ++s)
            if (slots[s * 8 + 7]) return SVTME_ERR_DEVICE;
    }
    out->n_b64                = f->n_b64;
    out->p_sb_best_sad        = f->skip_search_arrays ? nullptr : reinterpret_cast<const uint32_t *>(f->h_res + f->off_sad);
    out->p_sb_best_mv         = f->skip_search_arrays ? nullptr : reinterpret_cast<const uint32_t *>(f->h_res + f->off_mv);
    out->n_ref[0]             = f->n_ref[0];
    out->n_ref[1]             = f->n_ref[1];
    out->search_results       = reinterpret_cast<const svtme_search_result *>(f->h_res + f->off_sr);
    out->zz_sad               = reinterpret_cast<const uint32_t *>(f->h_res + f->off_zz);
    out->reduce_me_sr_divisor = reinterpret_cast<const uint32_t *>(f->h_res + f->off_div);
    out->total_me_candidate_index = f->tail ? f->h_res + f->off_ttot : nullptr;
    out->me_mv_array              = f->tail ? reinterpret_cast<const uint32_t *>(f->h_res + f->off_tmv) : nullptr;
    out->me_candidate_array       = f->tail ? f->h_res + f->off_tcand : nullptr;
    out->me_distortion            = f->tail ? reinterpret_cast<const uint32_t *>(f->h_res + f->off_tdist) : nullptr;
    out->b64_stats                = f->tail ? reinterpret_cast<const uint32_t *>(f->h_res + f->off_tstat) : nullptr;
    out->final_do_ref             = f->tail ? f->h_res + f->off_tdo : nullptr;
    out->final_hme_sad            = f->tail ? reinterpret_cast<const uint64_t *>(f->h_res + f->off_thme) : nullptr;
    return SVTME_OK;
}

extern "C" int svtme_picture_done(svtme_engine *e, svtme_ticket ticket) {
    if (!e) return SVTME_ERR_BAD_ARG;
    std::lock_guard<std::mutex> lock(e->mu);
    Flight *f = find_flight(e, ticket);
    if (!f) return SVTME_ERR_NOT_FOUND;
    if (!f->waited) return SVTME_ERR_BAD_ARG;
    f->busy = false;
    e->param_busy[f->param] = 0;
    BatchRec &B = e->batches[f->batch];
    if (--B.members_left == 0) B.used = false;
    return SVTME_OK;
}

extern "C" int svtme_picture_timing(svtme_engine *e, svtme_ticket ticket, float ms[8]) {
    if (!e || !ms) return SVTME_ERR_BAD_ARG;
    std::lock_guard<std::mutex> lock(e->mu);
    Flight *f = find_flight(e, ticket);
    if (!f || !f->waited) return SVTME_ERR_NOT_FOUND;
    // the pictures of a batch share their launches: each reports an equal share of the batch's durations
    const BatchRec &B = e->batches[f->batch];
    for (int i = 0; i < 5; ++i) {
        ms[i] = 0.f;
        if (B.kernel_timing) HIP_TRY(hipEventElapsedTime(&ms[i], B.ev[i], B.ev[i + 1]));
    }
    HIP_TRY(hipEventElapsedTime(&ms[5], B.ev[0], B.ev[5]));
    HIP_TRY(hipEventElapsedTime(&ms[6], B.ev[0], B.ev[6]));
    for (int i = 0; i < 7; ++i) ms[i] /= (float)B.n;
    ms[7] = (float)B.n;
    return SVTME_OK;
}

extern "C" int svtme_kernel_timing(svtme_engine *e, int on) {
    if (!e || on < 0 || on > 1) return SVTME_ERR_BAD_ARG;
    std::lock_guard<std::mutex> lock(e->mu);
    e->timing_on = on != 0;
    return SVTME_OK;
}

extern "C" int svtme_picture_ops(svtme_engine *e, svtme_ticket ticket, uint64_t ops[6]) {
    if (!e || !ops) return SVTME_ERR_BAD_ARG;
    std::lock_guard<std::mutex> lock(e->mu);
    Flight *f = find_flight(e, ticket);
    if (!f || !f->waited) return SVTME_ERR_NOT_FOUND;
    const uint64_t *slots = reinterpret_cast<const uint64_t *>(f->h_res + f->off_ops);
    for (int k = 0; k < 6; ++k) {
        ops[k] = 0;
        for (int s = 0; s < SVTME_OPS_SLOTS; ++s) ops[k] += slots[s * 8 + k];
    }
    return SVTME_OK;
}

extern "C" int svtme_unpack_b64(const svtme_results *res, uint32_t b64_index, uint32_t *p_sb_best_sad, uint32_t *p_sb_best_mv,
                                svtme_search_result *search_results, uint32_t *zz_sad, uint32_t *reduce_me_sr_divisor) {
    if (!res || b64_index >= res->n_b64) return SVTME_ERR_BAD_ARG;
    const size_t n = SVTME_MAX_LISTS * SVTME_MAX_REFS;
    if (res->n_ref[0] > SVTME_MAX_REFS || res->n_ref[1] > SVTME_MAX_REFS) return SVTME_ERR_BAD_ARG;
    // the planes of the searched references -> MeContext's [list][ref][pu] arrays; references the picture does not have read as zero
    const uint32_t *src[2] = {res->p_sb_best_sad, res->p_sb_best_mv};
    uint32_t       *dst[2] = {p_sb_best_sad, p_sb_best_mv};
    for (int a = 0; a < 2; ++a) {
        if (!dst[a] || !src[a]) continue;
        memset(dst[a], 0, n * SVTME_PU_COUNT * 4);
        for (uint32_t l = 0; l < SVTME_MAX_LISTS; ++l)
            for (uint32_t r = 0; r < res->n_ref[l]; ++r) {
                const size_t plane = l ? res->n_ref[0] + r : r;
                memcpy(dst[a] + (l * SVTME_MAX_REFS + r) * SVTME_PU_COUNT, src[a] + (plane * res->n_b64 + b64_index) * SVTME_PU_COUNT,
                       SVTME_PU_COUNT * 4);
            }
    }
    if (search_results) memcpy(search_results, res->search_results + (size_t)b64_index * n, n * sizeof(svtme_search_result));
    if (zz_sad) memcpy(zz_sad, res->zz_sad + (size_t)b64_index * n, n * 4);
    if (reduce_me_sr_divisor) memcpy(reduce_me_sr_divisor, res->reduce_me_sr_divisor + (size_t)b64_index * n, n * 4);
    return SVTME_OK;
}

// ------------------------------------------------------------------------------------------------ global-motion estimation
namespace {

int gm_init(svtme_engine *e) { // e->gm.mu held, device current
    GmState &g = e->gm;
    if (g.ready) return SVTME_OK;
    const size_t px = (size_t)e->cfg.max_width * e->cfg.max_height;
    const size_t blocks = (size_t)((e->cfg.max_width + 31) / 32) * ((e->cfg.max_height + 31) / 32);
    if (hipMalloc((void **)&g.d_score, px) != hipSuccess || hipMalloc((void **)&g.d_keep, px) != hipSuccess ||
        hipMalloc((void **)&g.d_rows, (size_t)e->cfg.max_height * sizeof(int)) != hipSuccess ||
        hipMalloc((void **)&g.d_corners[0], kGmCornerInts * sizeof(int)) != hipSuccess ||
        hipMalloc((void **)&g.d_corners[1], kGmCornerInts * sizeof(int)) != hipSuccess ||
        hipMalloc((void **)&g.d_match, kGmMatchInts * sizeof(int)) != hipSuccess ||
        hipMalloc((void **)&g.d_block_err, blocks * sizeof(uint32_t)) != hipSuccess ||
        hipMalloc((void **)&g.d_result, sizeof(long long)) != hipSuccess || hipMalloc((void **)&g.d_refine, sizeof(GmRefineState)) != hipSuccess ||
        hipMalloc((void **)&g.d_filter, 193 * 8 * sizeof(int16_t)) != hipSuccess ||
        hipMalloc((void **)&g.d_lut, 512 * sizeof(int32_t)) != hipSuccess ||
        hipHostMalloc((void **)&g.h_buf, (kGmCornerInts + kGmMatchInts + 4) * sizeof(int), hipHostMallocDefault) != hipSuccess) {
        (void)hipGetLastError();
        return SVTME_ERR_NO_RESOURCES; // (what was allocated is freed with the engine)
    }
    g.ready = true;
    return SVTME_OK;
}

// The planes of `n` keys at a downsample level, held against release for the duration of a call.
struct GmPlanes {
    svtme_engine *e;
    static constexpr int kMax = 1 + SVTME_MAX_LISTS * SVTME_MAX_REFS;
    int           n = 0, slot[kMax];
    DevPlane      plane[kMax];
    hipEvent_t    ready[kMax];
    uint64_t      gen[kMax];
    explicit GmPlanes(svtme_engine *e_) : e(e_) {}
    int take(const uint64_t *keys, int count, int level) {
        std::unique_lock<std::mutex> lock(e->mu);
        for (int i = 0; i < count; ++i) {
            int si;
            // a key that is being put (its planes are on their way), or whose put waits for earlier readers: after that put
            for (;;) {
                si = find_slot(e, keys[i], true);
                if (si < 0 || (!e->slots[si].putting && e->slots[si].put_waiting == 0)) break;
                e->cv_sync_readers.wait(lock);
            }
            if (si < 0) {
                drop_locked();
                return SVTME_ERR_NOT_FOUND;
            }
            Slot &s = e->slots[si];
            s.readers++;
            s.sync_readers++;
            slot[n]  = si;
            plane[n] = s.pic.lvl[2 - level]; // GM_FULL 0 -> the full plane, GM_DOWN 1 -> quarter, GM_DOWN16 2 -> sixteenth
            ready[n] = s.ready;
            gen[n]   = s.gen;
            ++n;
        }
        return SVTME_OK;
    }
    void drop_locked() {
        bool wake = false;
        for (int i = 0; i < n; ++i) {
            Slot &s = e->slots[slot[i]];
            if (--s.sync_readers == 0) wake = true;
            if (--s.readers == 0 && s.release_pending) {
                s.used            = false;
                s.release_pending = false;
            }
        }
        n = 0;
        if (wake) e->cv_sync_readers.notify_all();
    }
    ~GmPlanes() {
        if (n) {
            std::lock_guard<std::mutex> lock(e->mu);
            drop_locked();
        }
    }
};

// Index of the corner list that holds the corners of (key, level, gen), computing them (queued on `st`) if neither does;
// `avoid`: a list that must survive (-1: none).
int gm_corners_of(svtme_engine *e, uint64_t key, int level, uint64_t gen, const DevPlane &plane, int avoid, hipStream_t st) {
    GmState &g = e->gm;
    for (int i = 0; i < 2; ++i)
        if (g.cached[i].valid && g.cached[i].key == key && g.cached[i].level == level && g.cached[i].gen == gen) {
            g.last_used = i;
            return i;
        }
    const int i = avoid >= 0 ? 1 - avoid : 1 - g.last_used;
    launch_gm_corners(plane, g.d_score, g.d_keep, g.d_rows, SVTME_GM_MAX_CORNERS, g.d_corners[i], st);
    g.cached[i].valid = true;
    g.cached[i].key   = key;
    g.cached[i].level = level;
    g.cached[i].gen   = gen;
    g.last_used       = i;
    return i;
}

#define GM_TRY(expr)                                                                                      \
    do {                                                                                                  \
        hipError_t err__ = (expr);                                                                        \
        if (err__ != hipSuccess) {                                                                        \
            fprintf(stderr, "svtme: %s failed: %s (%s:%d)\n", #expr, hipGetErrorString(err__), __FILE__, __LINE__); \
            e->gm.cached[0].valid = e->gm.cached[1].valid = false;                                        \
            (void)hipStreamSynchronize(st);                                                               \
            return SVTME_ERR_DEVICE;                                                                      \
        }                                                                                                 \
    } while (0)

} // namespace

extern "C" int svtme_gm_set_tables(svtme_engine *e, const int16_t *warped_filter, const int32_t *error_measure_lut) {
    if (!e || !warped_filter || !error_measure_lut) return SVTME_ERR_BAD_ARG;
    DEVICE_OR_FAIL(e->cfg.device);
    std::lock_guard<std::mutex> gm_lock(e->gm.mu);
    if (const int rc = gm_init(e)) return rc;
    HIP_TRY(hipMemcpy(e->gm.d_filter, warped_filter, 193 * 8 * sizeof(int16_t), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(e->gm.d_lut, error_measure_lut, 512 * sizeof(int32_t), hipMemcpyHostToDevice));
    e->gm.tables = true;
    return SVTME_OK;
}

extern "C" int svtme_gm_corners(svtme_engine *e, uint64_t key, int downsample_level, int32_t *points, int32_t max_points, int32_t *num_points) {
    if (!e || !points || !num_points || max_points < 0 || max_points > SVTME_GM_MAX_CORNERS || downsample_level < 0 || downsample_level > 2)
        return SVTME_ERR_BAD_ARG;
    DEVICE_OR_FAIL(e->cfg.device);
    std::lock_guard<std::mutex> gm_lock(e->gm.mu);
    if (const int rc = gm_init(e)) return rc;
    GmPlanes pl(e);
    if (const int rc = pl.take(&key, 1, downsample_level)) return rc;
    const hipStream_t st = e->s_me[3];
    GmState          &g = e->gm;
    GM_TRY(hipStreamWaitEvent(st, pl.ready[0], 0));
    const int i = gm_corners_of(e, key, downsample_level, pl.gen[0], pl.plane[0], -1, st);
    GM_TRY(hipGetLastError());
    GM_TRY(hipMemcpyAsync(g.h_buf, g.d_corners[i], kGmCornerInts * sizeof(int), hipMemcpyDeviceToHost, st));
    GM_TRY(hipStreamSynchronize(st));
    const int n = std::min<int>(g.h_buf[0], max_points); // the first max_points in raster order, as the reference cuts them
    memcpy(points, g.h_buf + 2, sizeof(int32_t) * 2 * (size_t)n);
    *num_points = n;
    return SVTME_OK;
}

extern "C" int svtme_gm_correspondences(svtme_engine *e, uint64_t cur_key, uint64_t ref_key, int downsample_level, int32_t *correspondences,
                                        int32_t *num_correspondences) {
    if (!e || !correspondences || !num_correspondences || downsample_level < 0 || downsample_level > 2) return SVTME_ERR_BAD_ARG;
    DEVICE_OR_FAIL(e->cfg.device);
    std::lock_guard<std::mutex> gm_lock(e->gm.mu);
    if (const int rc = gm_init(e)) return rc;
    GmPlanes       pl(e);
    const uint64_t keys[2] = {cur_key, ref_key};
    if (const int rc = pl.take(keys, 2, downsample_level)) return rc;
    if (pl.plane[0].width != pl.plane[1].width || pl.plane[0].height != pl.plane[1].height) return SVTME_ERR_BAD_ARG;
    const hipStream_t st = e->s_me[3];
    GmState          &g = e->gm;
    GM_TRY(hipStreamWaitEvent(st, pl.ready[0], 0));
    GM_TRY(hipStreamWaitEvent(st, pl.ready[1], 0));
    const int ic = gm_corners_of(e, cur_key, downsample_level, pl.gen[0], pl.plane[0], -1, st);
    const int ir = cur_key == ref_key ? ic : gm_corners_of(e, ref_key, downsample_level, pl.gen[1], pl.plane[1], ic, st);
    launch_gm_match(pl.plane[0], pl.plane[1], g.d_corners[ic], g.d_corners[ir], SVTME_GM_MAX_CORNERS, g.d_match, st);
    GM_TRY(hipGetLastError());
    GM_TRY(hipMemcpyAsync(g.h_buf, g.d_corners[ic], 2 * sizeof(int), hipMemcpyDeviceToHost, st));
    GM_TRY(hipMemcpyAsync(g.h_buf + 2, g.d_match, kGmMatchInts * sizeof(int), hipMemcpyDeviceToHost, st));
    GM_TRY(hipStreamSynchronize(st));
    const int n_frm = std::min<int>(g.h_buf[0], SVTME_GM_MAX_CORNERS);
    int       n = 0;
    for (int i = 0; i < n_frm; ++i) { // the matched corners, in the order of the current picture's corner list
        const int *m = g.h_buf + 2 + 5 * i;
        if (m[4]) memcpy(correspondences + 4 * n++, m, 4 * sizeof(int32_t));
    }
    *num_correspondences = n;
    return SVTME_OK;
}

static int gm_error(svtme_engine *e, uint64_t cur_key, uint64_t ref_key, int downsample_level, const svtme_gm_warp *wm, int64_t best_error,
                    int64_t *error) {
    if (!e || !error || downsample_level < 0 || downsample_level > 2) return SVTME_ERR_BAD_ARG;
    DEVICE_OR_FAIL(e->cfg.device);
    std::lock_guard<std::mutex> gm_lock(e->gm.mu);
    if (!e->gm.ready || !e->gm.tables) return SVTME_ERR_BAD_ARG; // svtme_gm_set_tables first
    GmPlanes       pl(e);
    const uint64_t keys[2] = {cur_key, ref_key};
    if (const int rc = pl.take(keys, 2, downsample_level)) return rc;
    if (pl.plane[0].width != pl.plane[1].width || pl.plane[0].height != pl.plane[1].height) return SVTME_ERR_BAD_ARG;
    const hipStream_t st = e->s_me[3];
    GmState          &g = e->gm;
    GM_TRY(hipStreamWaitEvent(st, pl.ready[0], 0));
    GM_TRY(hipStreamWaitEvent(st, pl.ready[1], 0));
    launch_gm_error(pl.plane[0], pl.plane[1], wm, g.d_filter, g.d_lut, g.d_block_err, (long long)best_error, g.d_result, st);
    GM_TRY(hipGetLastError());
    GM_TRY(hipMemcpyAsync(g.h_buf, g.d_result, sizeof(long long), hipMemcpyDeviceToHost, st));
    GM_TRY(hipStreamSynchronize(st));
    long long r;
    memcpy(&r, g.h_buf, sizeof(r));
    *error = (int64_t)r;
    return SVTME_OK;
}

extern "C" int svtme_gm_warp_error(svtme_engine *e, uint64_t cur_key, uint64_t ref_key, int downsample_level, const svtme_gm_warp *wm,
                                   int64_t best_error, int64_t *error) {
    if (!wm) return SVTME_ERR_BAD_ARG;
    return gm_error(e, cur_key, ref_key, downsample_level, wm, best_error, error);
}

extern "C" int svtme_gm_frame_error(svtme_engine *e, uint64_t cur_key, uint64_t ref_key, int downsample_level, int64_t *error) {
    return gm_error(e, cur_key, ref_key, downsample_level, nullptr, INT64_MAX, error);
}

extern "C" int svtme_gm_refine(svtme_engine *e, uint64_t cur_key, uint64_t ref_key, int downsample_level, svtme_gm_model *wm, int wmtype,
                               int n_refinements, int64_t best_frame_error, int64_t *error, uint32_t *evaluations) {
    // n_refinements: the loop's first step is 1 << (n_refinements - 1); the parameters' ranges (GM_TRANS_MAX, GM_ALPHA_MAX =
    // 1 << 12 units) bound what a step can be
    if (!e || !wm || !error || downsample_level < 0 || downsample_level > 2 || wmtype < 0 || wmtype > 3 || n_refinements < 1 || n_refinements > 13)
        return SVTME_ERR_BAD_ARG;
    DEVICE_OR_FAIL(e->cfg.device);
    std::lock_guard<std::mutex> gm_lock(e->gm.mu);
    if (!e->gm.ready || !e->gm.tables) return SVTME_ERR_BAD_ARG; // svtme_gm_set_tables first
    GmPlanes       pl(e);
    const uint64_t keys[2] = {cur_key, ref_key};
    if (const int rc = pl.take(keys, 2, downsample_level)) return rc;
    if (pl.plane[0].width != pl.plane[1].width || pl.plane[0].height != pl.plane[1].height) return SVTME_ERR_BAD_ARG;
    const hipStream_t st = e->s_me[3];
    GmState          &g = e->gm;
    GM_TRY(hipStreamWaitEvent(st, pl.ready[0], 0));
    GM_TRY(hipStreamWaitEvent(st, pl.ready[1], 0));
    GmRefineState init{};
    for (int i = 0; i < 6; ++i) init.mat[i] = wm->wmmat[i];
    init.alpha         = wm->alpha;
    init.beta          = wm->beta;
    init.gamma         = wm->gamma;
    init.delta         = wm->delta;
    init.wmtype        = wmtype;
    init.n_refinements = n_refinements;
    init.best_error    = (long long)best_frame_error;
    launch_gm_refine_begin(g.d_refine, init, st);
    // The loop's shortest run: the first evaluation, then left and right for every parameter of every step; a walk adds to it.
    // That many launches go out at once, with some to spare for walks (a launch after the loop's end costs next to
    // nothing), then the state comes back; while the loop has not ended, more of them.
    GmRefineState *back = reinterpret_cast<GmRefineState *>(g.h_buf);
    int            chunk = 1 + 2 * (2 * wmtype) * n_refinements + 8;
    for (int launched = 0;; chunk = 16) {
        launch_gm_refine_evaluations(pl.plane[0], pl.plane[1], g.d_refine, g.d_filter, g.d_lut, g.d_block_err, chunk, st);
        launched += chunk;
        GM_TRY(hipGetLastError());
        GM_TRY(hipMemcpyAsync(back, g.d_refine, sizeof(GmRefineState), hipMemcpyDeviceToHost, st));
        GM_TRY(hipStreamSynchronize(st));
        if (back->done) break;
        // every walk step moves a clamped parameter by at least one unit, so the loop ends; this guards the host against a
        // device that does not report it
        if (launched > 6 * 13 * 2 * 8192 + 64) return SVTME_ERR_DEVICE;
    }
    for (int i = 0; i < 6; ++i) wm->wmmat[i] = back->mat[i];
    wm->alpha  = back->alpha;
    wm->beta   = back->beta;
    wm->gamma  = back->gamma;
    wm->delta  = back->delta;
    wm->wmtype = back->wmtype;
    *error     = (int64_t)back->best_error;
    if (evaluations) *evaluations = (uint32_t)back->evals;
    return SVTME_OK;
}

// ---- TPL dispenser: source-based mode search
extern "C" int svtme_tpl_search(svtme_engine *e, const svtme_tpl_picture *pic, svtme_tpl_block *out) {
    if (!e || !pic || !out || !pic->total_me_candidate_index || !pic->me_candidate_array || !pic->me_mv_array) return SVTME_ERR_BAD_ARG;
    if (pic->width < 64 || pic->height < 64 || pic->width > e->cfg.max_width || pic->height > e->cfg.max_height || pic->width % 8 ||
        pic->height % 8 || pic->dispenser_search_level > 2 || pic->n_ref[0] > SVTME_MAX_REFS || pic->n_ref[1] > SVTME_MAX_REFS ||
        // MeSbResults holds 85, 21 (no 8x8 partitions) or 5 (no 16x16 either) entries per block (EbPictureControlSet.c:112-117);
        // the blocks' 16x16 entries are read unless enable_me_16x16 is 0, the 32x32 ones then (:827-828)
        (pic->n_pu != SVTME_PU_COUNT && pic->n_pu != 21 && !(pic->n_pu == 5 && !pic->enable_me_16x16)) || pic->max_cand == 0 ||
        pic->max_refs == 0 ||
        pic->max_l0 + (pic->n_ref[1] ? pic->n_ref[1] : 0) > pic->max_refs || pic->n_ref[0] > pic->max_l0)
        return SVTME_ERR_BAD_ARG;
    // transform-domain costs are there for whole 16x16 transforms, the source distortion for the transforms the tpl levels
    // use: 16x16, 16x8, 16x4 on 16x16 blocks, 32x8 on 32x32 blocks. Without tpl_opt_flag every cost is a transform-domain
    // cost of the whole transform and every intra mode up to intra_mode_end is tried
    // (EbSourceBasedOperationsProcess.c:730-733, :780-813, :878-917).
    if (pic->pf_shape > 2) return SVTME_ERR_UNSUPPORTED;
    const bool all_modes = !pic->tpl_opt_flag;
    if (all_modes && pic->intra_mode_end != 12 && pic->intra_mode_end != 11 && pic->intra_mode_end != 8) return SVTME_ERR_BAD_ARG;
    const bool sad_intra = !all_modes && pic->use_pred_sad_in_intra_search, sad_inter = !all_modes && pic->use_pred_sad_in_inter_search;
    if ((!sad_intra || !sad_inter) && (pic->dispenser_search_level != 0 || pic->subsample_tx != 0)) return SVTME_ERR_UNSUPPORTED;
    if (pic->src_dist && (pic->subsample_tx > 2 || pic->dispenser_search_level > 1 || (pic->dispenser_search_level == 1 && pic->subsample_tx != 2)))
        return SVTME_ERR_UNSUPPORTED;
    DEVICE_OR_FAIL(e->cfg.device);
    std::lock_guard<std::mutex> gm_lock(e->gm.mu);
    GmState  &g = e->gm;
    const int b64_w = (int)(pic->width + 63) / 64, n_b64 = b64_w * ((int)(pic->height + 63) / 64);
    const size_t cnt_bytes = (size_t)n_b64 * pic->n_pu, cand_bytes = cnt_bytes * pic->max_cand, mv_bytes = cnt_bytes * pic->max_refs * 4;
    const size_t mv_off = (cnt_bytes + cand_bytes + 63) / 64 * 64, out_off = (mv_off + mv_bytes + 63) / 64 * 64;
    const size_t out_bytes = (size_t)n_b64 * SVTME_TPL_BLOCKS * sizeof(svtme_tpl_block), err_off = out_off + out_bytes;
    const size_t par_off = err_off + 64, total = par_off + (sizeof(TplParams) + 63) / 64 * 64; // the record travels with the ME results
    if (g.tpl_bytes < total) {
        if (g.d_tpl) (void)hipFree(g.d_tpl);
        if (g.h_tpl) (void)hipHostFree(g.h_tpl);
        g.d_tpl = g.h_tpl = nullptr;
        g.tpl_bytes = 0;
        if (hipMalloc((void **)&g.d_tpl, total) != hipSuccess || hipHostMalloc((void **)&g.h_tpl, total, hipHostMallocDefault) != hipSuccess) {
            (void)hipGetLastError();
            return SVTME_ERR_NO_RESOURCES;
        }
        g.tpl_bytes = total;
    }
    GmPlanes pl(e);
    uint64_t keys[GmPlanes::kMax];
    int      n_keys = 0;
    keys[n_keys++]  = pic->cur_key;
    for (int l = 0; l < SVTME_MAX_LISTS; ++l)
        for (int r = 0; r < pic->n_ref[l]; ++r) keys[n_keys++] = pic->ref_key[l][r];
    if (const int rc = pl.take(keys, n_keys, 0)) return rc;
    TplParams p;
    memset(&p, 0, sizeof(p));
    p.cur = pl.plane[0];
    if (p.cur.width != (int)pic->width || p.cur.height != (int)pic->height) return SVTME_ERR_BAD_ARG;
    for (int l = 0, k = 1; l < SVTME_MAX_LISTS; ++l)
        for (int r = 0; r < pic->n_ref[l]; ++r, ++k) {
            if (pl.plane[k].width != p.cur.width || pl.plane[k].height != p.cur.height) return SVTME_ERR_BAD_ARG;
            p.ref[l][r]      = pl.plane[k];
            p.ref_skip[l][r] = pic->ref_skip[l][r];
        }
    p.n_ref[0] = pic->n_ref[0];
    p.n_ref[1] = pic->n_ref[1];
    p.width = (int)pic->width;
    p.height = (int)pic->height;
    p.b64_w = b64_w;
    p.n_b64 = n_b64;
    p.level = pic->dispenser_search_level;
    p.disable_intra_pred = pic->disable_intra_pred;
    p.i_slice = pic->i_slice;
    p.enable_me_16x16 = pic->enable_me_16x16;
    p.n_pu = pic->n_pu;
    p.max_cand = pic->max_cand;
    p.max_refs = pic->max_refs;
    p.max_l0 = pic->max_l0;
    p.sad_intra = sad_intra;
    p.sad_inter = sad_inter;
    p.pf_shape  = all_modes ? 0 : pic->pf_shape; // DEFAULT_SHAPE without tpl_opt_flag (:800-802, :909-911)
    p.mode_end  = all_modes ? pic->intra_mode_end : 0;
    p.rate      = all_modes && pic->src_dist;
    p.src_dist  = pic->src_dist != 0;
    p.subsample_tx = pic->subsample_tx;
    for (int k = 0; k < 2; ++k) {
        p.round_fp[k] = pic->y_round_fp[k];
        p.quant_fp[k] = pic->y_quant_fp[k];
        p.dequant[k]  = pic->y_dequant[k];
    }
    for (int k = 0; k < 32; ++k) { // cospi_arr(13) / (12) at every second angle (EbInvTransforms.h)
        p.cos13[k] = (int32_t)lround(cos(2 * k * 3.14159265358979323846 / 128.0) * 8192.0);
        p.cos12[k] = (int32_t)lround(cos(2 * k * 3.14159265358979323846 / 128.0) * 4096.0);
    }
    p.cnt   = g.d_tpl;
    p.cand  = g.d_tpl + cnt_bytes;
    p.mv    = reinterpret_cast<const uint32_t *>(g.d_tpl + mv_off);
    p.out   = reinterpret_cast<svtme_tpl_block *>(g.d_tpl + out_off);
    p.error = reinterpret_cast<int *>(g.d_tpl + err_off);
    memcpy(g.h_tpl, pic->total_me_candidate_index, cnt_bytes);
    memcpy(g.h_tpl + cnt_bytes, pic->me_candidate_array, cand_bytes);
    memcpy(g.h_tpl + mv_off, pic->me_mv_array, mv_bytes);
    memcpy(g.h_tpl + par_off, &p, sizeof(p));
    const hipStream_t st = e->s_me[3];
    for (int k = 0; k < n_keys; ++k) GM_TRY(hipStreamWaitEvent(st, pl.ready[k], 0));
    GM_TRY(hipMemcpyAsync(g.d_tpl, g.h_tpl, mv_off + mv_bytes, hipMemcpyHostToDevice, st));
    GM_TRY(hipMemcpyAsync(g.d_tpl + par_off, g.h_tpl + par_off, sizeof(p), hipMemcpyHostToDevice, st));
    GM_TRY(hipMemsetAsync(g.d_tpl + out_off, 0, out_bytes + 64, st));
    launch_tpl_search(reinterpret_cast<const TplParams *>(g.d_tpl + par_off), n_b64, p.mode_end > 0 || !p.sad_inter || !p.sad_intra, st);
    GM_TRY(hipGetLastError());
    GM_TRY(hipMemcpyAsync(g.h_tpl + out_off, g.d_tpl + out_off, out_bytes + 64, hipMemcpyDeviceToHost, st));
    GM_TRY(hipStreamSynchronize(st));
    int err;
    memcpy(&err, g.h_tpl + err_off, sizeof(err));
    if (err) return SVTME_ERR_BAD_ARG;
    memcpy(out, g.h_tpl + out_off, out_bytes);
    return SVTME_OK;
}

// ---- TPL dispenser: reconstruction path
extern "C" int svtme_tpl_recon(svtme_engine *e, const svtme_tpl_picture *pic, const svtme_tpl_block *blocks, svtme_tpl_stats *stats, uint8_t *recon,
                               uint32_t recon_stride) {
    if (!e || !pic || !blocks || !stats) return SVTME_ERR_BAD_ARG;
    if (pic->width < 64 || pic->height < 64 || pic->width > e->cfg.max_width || pic->height > e->cfg.max_height || pic->width % 8 ||
        pic->height % 8 || pic->n_ref[0] > SVTME_MAX_REFS || pic->n_ref[1] > SVTME_MAX_REFS || (recon && recon_stride < pic->width))
        return SVTME_ERR_BAD_ARG;
    // 16x16 blocks with the 16x16, 16x8 and 16x4 transforms (tpl levels 1 to 6); 32x32 blocks with the 32x8 transform, whose
    // incomplete 64x64 blocks take 16x16 blocks with 16x4 (tpl level 7)
    if (pic->dispenser_search_level > 1 || pic->subsample_tx > 2 || pic->pf_shape > 2) return SVTME_ERR_UNSUPPORTED;
    const bool all_modes = !pic->tpl_opt_flag;
    if (all_modes && pic->subsample_tx != 0) return SVTME_ERR_UNSUPPORTED; // rate_estimator is there for the 16x16 transform
    if (pic->dispenser_search_level == 1 && (pic->subsample_tx != 2 || all_modes)) return SVTME_ERR_UNSUPPORTED;
    const int b64_w = (int)(pic->width + 63) / 64, n_b64 = b64_w * ((int)(pic->height + 63) / 64);
    const size_t n_blk = (size_t)n_b64 * SVTME_TPL_BLOCKS;
    for (size_t i = 0; i < n_blk; ++i) { // what the kernels index with
        const svtme_tpl_block &b = blocks[i];
        if (!b.valid || !b.best_is_inter) {
            if (b.valid && b.best_intra_mode > 12) return SVTME_ERR_BAD_ARG;
            continue;
        }
        if (b.best_rf_idx < 0 || b.best_rf_idx > 7 || (b.best_rf_idx & 3) >= pic->n_ref[b.best_rf_idx >> 2]) return SVTME_ERR_BAD_ARG;
        // the vector keeps the predicted block within TPL_PADX / TPL_PADY pixels of the picture, as the search's clamp leaves it
        // (:853-876): the planes' borders are what the kernels may read
        const int bi = (int)(i % SVTME_TPL_BLOCKS), sb = (int)(i / SVTME_TPL_BLOCKS);
        if (bi == 0) return SVTME_ERR_BAD_ARG; // (64x64 blocks: dispenser_search_level 2)
        const int S = bi < 5 ? 32 : 16, q = bi < 5 ? bi - 1 : (bi - 5) >> 2, in_q = (bi - 5) & 3;
        const int x = (sb % b64_w) * 64 + (q & 1) * 32 + (bi < 5 ? 0 : (in_q & 1) * 16), y = (sb / b64_w) * 64 + (q >> 1) * 32 + (bi < 5 ? 0 : (in_q >> 1) * 16);
        const int px = x + (b.mv_x >> 3), py = y + (b.mv_y >> 3);
        if (px < -32 || px + S > 32 + (int)pic->width - 1 || py < -32 || py + S > 32 + (int)pic->height - 1) return SVTME_ERR_BAD_ARG;
    }
    DEVICE_OR_FAIL(e->cfg.device);
    std::lock_guard<std::mutex> gm_lock(e->gm.mu);
    GmState  &g = e->gm;
    const size_t blk_bytes = n_blk * sizeof(svtme_tpl_block), stat_bytes = n_blk * sizeof(svtme_tpl_stats);
    const size_t stat_off = (blk_bytes + 63) / 64 * 64, par_off = (stat_off + stat_bytes + 63) / 64 * 64;
    // The intra blocks in levels: a block reads the reconstruction of the blocks left of, above and above-left of it, and on the
    // picture's first column of the one above-right (EbEncIntraPrediction.c:1237-1319); those predicted from reference pictures
    // are all done first, so a block's level is one more than the highest level among its intra neighbours.
    // (Kept per 16x16 cell of the picture; a 32x32 block fills its four cells. The 64x64 blocks in raster order and the blocks
    // inside them in the dispenser's order: whatever a block reads comes earlier.)
    const int bw16 = (int)(pic->width + 15) / 16, bh16 = (int)(pic->height + 15) / 16;
    std::vector<int> level((size_t)bw16 * bh16, -1), level_count;
    std::vector<std::pair<int, uint32_t>> found; // (level, position)
    for (int sb = 0; sb < n_b64; ++sb) {
        const int  sx = (sb % b64_w) * 4, sy = (sb / b64_w) * 4;
        const bool big = pic->dispenser_search_level == 1 && sx * 16 + 64 <= (int)pic->width && sy * 16 + 64 <= (int)pic->height;
        for (int k = 0; k < (big ? 4 : 16); ++k) {
            const int S = big ? 2 : 1; // cells per side
            const int bx = big ? sx + (k & 1) * 2 : sx + ((k >> 2) & 1) * 2 + (k & 1), by = big ? sy + (k >> 1) * 2 : sy + (k >> 3) * 2 + ((k >> 1) & 1);
            if (bx * 16 + S * 8 > (int)pic->width || by * 16 + S * 8 > (int)pic->height) continue;
            const int bi = big ? 1 + k : 5 + k;
            const svtme_tpl_block &b = blocks[(size_t)sb * SVTME_TPL_BLOCKS + bi];
            if (!b.valid || b.best_is_inter) continue;
            auto at = [&](int x, int y) { return x >= 0 && y >= 0 && x < bw16 && y < bh16 ? level[(size_t)y * bw16 + x] : -1; };
            int  l  = -1;
            for (int i = -1; i < (bx == 0 ? 2 * S : S); ++i) l = std::max(l, at(bx + i, by - 1)); // the row above (and above-right on the first column)
            for (int i = 0; i < S; ++i) l = std::max(l, at(bx - 1, by + i)); // the column to the left
            ++l;
            for (int j = 0; j < S; ++j)
                for (int i = 0; i < S; ++i)
                    if (bx + i < bw16 && by + j < bh16) level[(size_t)(by + j) * bw16 + bx + i] = l;
            if ((int)level_count.size() <= l) level_count.resize(l + 1, 0);
            ++level_count[l];
            found.emplace_back(l, (uint32_t)bx | ((uint32_t)by << 16));
        }
    }
    std::vector<uint32_t> order(found.size());
    {
        std::vector<int> next(level_count.size() + 1, 0);
        for (size_t l = 0; l < level_count.size(); ++l) next[l + 1] = next[l] + level_count[l];
        for (const auto &f : found) order[next[f.first]++] = f.second;
    }
    const size_t rec_stride = ((size_t)pic->width + 63) / 64 * 64, list_off = par_off + (sizeof(TplParams) + 63) / 64 * 64;
    const size_t rec_off = (list_off + (size_t)bw16 * bh16 * 4 + 63) / 64 * 64, total = rec_off + rec_stride * pic->height;
    if (!g.rec_done && (hipEventCreateWithFlags(&g.rec_done, hipEventDisableTiming) != hipSuccess ||
                        hipEventCreateWithFlags(&g.rec_read, hipEventDisableTiming) != hipSuccess))
        return SVTME_ERR_NO_RESOURCES;
    g.rec_w = g.rec_h = 0; // nothing to keep until this call has succeeded
    if (g.rec_read_pending) { // store kernels of svtme_tpl_recon_keep may still read the last picture
        if (hipEventSynchronize(g.rec_read) != hipSuccess) return SVTME_ERR_DEVICE;
        g.rec_read_pending = false;
    }
    if (g.rec_bytes < total) {
        if (g.d_rec) (void)hipFree(g.d_rec);
        if (g.h_rec) (void)hipHostFree(g.h_rec);
        g.d_rec = g.h_rec = nullptr;
        g.rec_bytes = 0;
        if (hipMalloc((void **)&g.d_rec, total) != hipSuccess || hipHostMalloc((void **)&g.h_rec, total, hipHostMallocDefault) != hipSuccess) {
            (void)hipGetLastError();
            return SVTME_ERR_NO_RESOURCES;
        }
        g.rec_bytes = total;
    }
    GmPlanes pl(e);
    uint64_t keys[GmPlanes::kMax];
    int      n_keys = 0;
    keys[n_keys++]  = pic->cur_key;
    for (int l = 0; l < SVTME_MAX_LISTS; ++l)
        for (int r = 0; r < pic->n_ref[l]; ++r) keys[n_keys++] = pic->ref_key[l][r];
    if (const int rc = pl.take(keys, n_keys, 0)) return rc;
    TplParams p;
    memset(&p, 0, sizeof(p));
    p.cur = pl.plane[0];
    if (p.cur.width != (int)pic->width || p.cur.height != (int)pic->height) return SVTME_ERR_BAD_ARG;
    for (int l = 0, k = 1; l < SVTME_MAX_LISTS; ++l)
        for (int r = 0; r < pic->n_ref[l]; ++r, ++k) {
            if (pl.plane[k].width != p.cur.width || pl.plane[k].height != p.cur.height) return SVTME_ERR_BAD_ARG;
            p.ref[l][r] = pl.plane[k];
        }
    p.n_ref[0] = pic->n_ref[0];
    p.n_ref[1] = pic->n_ref[1];
    p.width  = (int)pic->width;
    p.height = (int)pic->height;
    p.b64_w  = b64_w;
    p.n_b64  = n_b64;
    p.level  = pic->dispenser_search_level;
    p.disable_intra_pred = pic->disable_intra_pred;
    p.pf_shape     = all_modes ? 0 : pic->pf_shape; // DEFAULT_SHAPE without tpl_opt_flag (:1131-1133)
    p.rate         = all_modes; // rate_estimator, else 0 (:1141)
    p.subsample_tx = pic->subsample_tx;
    p.dc_path      = pic->tpl_opt_flag && pic->use_pred_sad_in_intra_search; // intra_dc_sad_path :594-595
    for (int k = 0; k < 2; ++k) {
        p.round_fp[k] = pic->y_round_fp[k];
        p.quant_fp[k] = pic->y_quant_fp[k];
        p.dequant[k]  = pic->y_dequant[k];
    }
    for (int k = 0; k < 32; ++k) {
        p.cos13[k] = (int32_t)lround(cos(2 * k * 3.14159265358979323846 / 128.0) * 8192.0);
        p.cos12[k] = (int32_t)lround(cos(2 * k * 3.14159265358979323846 / 128.0) * 4096.0);
    }
    p.out          = reinterpret_cast<svtme_tpl_block *>(g.d_rec);
    p.stats        = reinterpret_cast<svtme_tpl_stats *>(g.d_rec + stat_off);
    p.recon        = g.d_rec + rec_off;
    p.recon_stride = (int32_t)rec_stride;
    memcpy(g.h_rec, blocks, blk_bytes);
    memcpy(g.h_rec + par_off, &p, sizeof(p));
    if (!order.empty()) memcpy(g.h_rec + list_off, order.data(), order.size() * 4);
    const hipStream_t st = e->s_me[3];
    for (int k = 0; k < n_keys; ++k) GM_TRY(hipStreamWaitEvent(st, pl.ready[k], 0));
    GM_TRY(hipMemcpyAsync(g.d_rec, g.h_rec, blk_bytes, hipMemcpyHostToDevice, st));
    GM_TRY(hipMemcpyAsync(g.d_rec + par_off, g.h_rec + par_off, list_off - par_off + order.size() * 4, hipMemcpyHostToDevice, st)); // record and list
    GM_TRY(hipMemsetAsync(g.d_rec + stat_off, 0, stat_bytes, st));
    launch_tpl_recon(reinterpret_cast<const TplParams *>(g.d_rec + par_off), n_b64, reinterpret_cast<const uint32_t *>(g.d_rec + list_off),
                     level_count.data(), (int)level_count.size(), st);
    GM_TRY(hipGetLastError());
    GM_TRY(hipMemcpyAsync(g.h_rec + stat_off, g.d_rec + stat_off, stat_bytes, hipMemcpyDeviceToHost, st));
    if (recon) GM_TRY(hipMemcpyAsync(g.h_rec + rec_off, g.d_rec + rec_off, rec_stride * pic->height, hipMemcpyDeviceToHost, st));
    GM_TRY(hipEventRecord(g.rec_done, st));
    GM_TRY(hipStreamSynchronize(st));
    g.rec_pic_off    = rec_off;
    g.rec_pic_stride = rec_stride;
    g.rec_w          = pic->width;
    g.rec_h          = pic->height;
    memcpy(stats, g.h_rec + stat_off, stat_bytes);
    if (recon)
        for (uint32_t y = 0; y < pic->height; ++y) memcpy(recon + (size_t)y * recon_stride, g.h_rec + rec_off + (size_t)y * rec_stride, pic->width);
    return SVTME_OK;
}

extern "C" int svtme_tpl_recon_keep(svtme_engine *e, uint64_t key, uint64_t picture_number) {
    if (!e) return SVTME_ERR_BAD_ARG;
    DEVICE_OR_FAIL(e->cfg.device);
    std::lock_guard<std::mutex> gm_lock(e->gm.mu);
    GmState &g = e->gm;
    if (!g.rec_w) return SVTME_ERR_NOT_FOUND; // no reconstructed picture: no call yet, or the last one failed
    const int rc = store_put_impl(e, key, picture_number, nullptr, (uint32_t)g.rec_pic_stride, g.rec_w, g.rec_h, true, g.d_rec + g.rec_pic_off,
                                  g.rec_done, g.rec_read);
    if (rc == SVTME_OK) g.rec_read_pending = true;
    return rc;
}

// Diagnostic, not part of the public header: resident workgroups per CU of {hme_l0, hme_l12, me_search}.
extern "C" int svtme_debug_occupancy(int device, int out[3]) {
    DEVICE_OR_FAIL(device);
    occupancy_hme_items(out);
    out[2] = occupancy_me_search();
    return SVTME_OK;
}
