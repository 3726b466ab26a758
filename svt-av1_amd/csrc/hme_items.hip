// hme_items.hip - zero-MV SAD, pre-HME and HME levels 0/1/2 of one picture as two kernels of single-wave work items.
//
// The search functions are hme_search.h's (shared with hme_regions.hip); this unit holds the kernels hme_zz_kernel,
// hme_l0_kernel and hme_l12_kernel and their launcher.
#include "hme_search.h"

namespace svtme {

// Zero-MV SAD of every (b64, reference) for the controls with me_early_exit_th (init_zz_sad :2318-2333, get_zz_sad
// :1720-1742: the 64x64 block on every other row, x2), once instead of in each of the six level-0 / pre-HME items that
// test it; also the per-reference header fields. Four blocks per wave, 16 lanes each: a lane takes one dword column of the
// block's 32 sampled rows, all 64 loads in flight together (the kernel is one memory round trip and a reduction: a quarter
// of the waves for the same bytes).
template <class PICS>
__global__ void __launch_bounds__(HI_THREADS) hme_zz_kernel(PICS pp) {
    const auto &p    = pp[blockIdx.z];
    const int   lane = threadIdx.x & 63, d = lane & 15;
    const int   b64  = (int)blockIdx.x * 4 + (lane >> 4);
    int         l = 0, r = (int)blockIdx.y;
    if (r >= p.n_ref[0] + (p.n_list == 2 ? p.n_ref[1] : 0)) return; // the grid covers the batch's longest reference list
    if (r >= p.n_ref[0]) {
        r -= p.n_ref[0];
        l = 1;
    }
    const bool item = b64 < p.b64_w * p.b64_h && !b64_left_out(p, min(b64, p.b64_w * p.b64_h - 1));
    int        b64x, b64y;
    b64_origin(p, item ? b64 : 0, b64x, b64y);
    const int  block_w = min(p.width - b64x, 64), block_h = min(p.height - b64y, 64), bwd = block_w >> 2, rows = block_h >> 1;
    const bool searched = ref_searched(p, l);
    uint32_t   zz = ~0u;
    if (searched) {
        const DevPlane cur = load_plane(p.cur.lvl[2]), ref = load_plane(p.ref[l][r].lvl[2]);
        uint32_t       sv[32], fv[32];
#pragma unroll
        for (int k = 0; k < 32; ++k) {
            const bool on = item && d < bwd && k < rows;
            const int  rc = on ? k : 0, dc = on ? d : 0; // idle lanes repeat a valid address
            sv[k] = *reinterpret_cast<const uint32_t *>(cur.pix + (ptrdiff_t)(b64y + 2 * rc) * cur.stride + b64x + 4 * dc);
            fv[k] = *reinterpret_cast<const uint32_t *>(ref.pix + (ptrdiff_t)(b64y + 2 * rc) * ref.stride + b64x + 4 * dc);
        }
        uint32_t acc = 0;
#pragma unroll
        for (int k = 0; k < 32; ++k) acc = __builtin_amdgcn_sad_u8(sv[k], (item && d < bwd && k < rows) ? fv[k] : sv[k], acc);
#pragma unroll
        for (int m = 1; m < 16; m <<= 1) acc += __shfl_xor(acc, m); // the block's 16 lanes
        zz = acc << 1;
    }
    if (item && d == 0) {
        HmeRefOut *out = &p.hme_out[(b64 * SVTME_MAX_LISTS + l) * SVTME_MAX_REFS + r];
        if (searched) count_ops(p, 0, (unsigned long long)rows * block_w);
        out->zz_sad = zz;
        out->valid  = searched ? 1u : 0u;
    }
}

// One pre-HME shape (sub 0, 1) or level-0 region (sub 2..5, in the reference's visiting order) of reference (l, r),
// whole wave. `rec` is the reference's record (global or LDS), `first` the record of list 0 / reference 0 and `same0`
// the record of list 0 / reference r: both are only read when the picture's controls make references depend on each
// other (p.dep_l0, p.dep_pre), and are complete by then. zz: the zero-MV SAD of the reference (~0 when not computed).
template <bool FS, class PICS>
__global__ void __launch_bounds__(HI_THREADS, 6) hme_l0_kernel(PICS pp, int grid_refs, uint32_t grid_refs_magic, int hme_phase, int all_regions) {
    __shared__ __align__(16) uint32_t win[HI_WIN_BYTES / 4 + 64];
    // grid: x = 64x64 block, y = search of the reference, z = (picture, reference): the work items of one picture are
    // dispatched together; the picture's index by a multiplication (grid_refs_magic = ceil(2^32 / grid_refs))
    const int   pic_i = grid_refs == 1 ? (int)blockIdx.z : (int)__umulhi(blockIdx.z, grid_refs_magic);
    const auto &p   = pp[pic_i];
    const auto &k   = p.ctrl;
    const int   b64 = blockIdx.x;
    if (b64_left_out(p, b64)) return;
    int                   b64x, b64y;
    b64_origin(p, b64, b64x, b64y);
    const int             block_w = min(p.width - b64x, 64), block_h = min(p.height - b64y, 64);
    // references in reverse order: the closest references (smallest areas, cheapest items) are dispatched last and keep
    // the kernel's tail short (the grid covers the batch's longest reference list)
    int                   l = 0, r = p.n_ref[0] + (p.n_list == 2 ? p.n_ref[1] : 0) - 1 - ((int)blockIdx.z - pic_i * grid_refs);
    if (r < 0) return;
    if (r >= p.n_ref[0]) {
        r -= p.n_ref[0];
        l = 1;
    }
    const int  sub      = blockIdx.y; // 0, 1: pre-HME shapes; 2..: level-0 regions in the reference's visiting order
    // hme_phase 0: every item; 1: the items others depend on; 2: the items that read another reference's result, which
    // run in a second launch
    if (hme_phase) {
        const bool dependent = sub < 2 ? (p.dep_pre && l == 1) : (p.dep_l0 && (l || r));
        if (dependent != (hme_phase == 2)) return;
    }
    const bool searched = ref_searched(p, l);
    HmeRefOut *b64_out  = &p.hme_out[b64 * SVTME_MAX_LISTS * SVTME_MAX_REFS];
    HmeRefOut *out      = &b64_out[l * SVTME_MAX_REFS + r];
    const bool header   = sub == 2; // this item also writes the per-reference fields

    // zero-MV SAD (init_zz_sad :2318-2333); stays ~0 (init_me_hme_data :2921) when not computed. With the early exit on,
    // hme_zz_kernel has computed it once for the six items of this (b64, reference) and has written the header fields.
    uint32_t zz = ~0u;
    if (k.me_early_exit_th) {
        if (searched) zz = out->zz_sad;
    } else if (header && threadIdx.x == 0) {
        out->zz_sad = zz;
        out->valid  = searched ? 1u : 0u;
    }
    if (!searched) return;
    l0_item<FS>(p, win, b64x, b64y, block_w, block_h, l, r, sub, zz, out, b64_out, &b64_out[r], all_regions != 0);
}

template <bool FS, class PICS>
__global__ void __launch_bounds__(FS ? HI_THREADS_FS : HI_THREADS, FS ? 4 : 5) hme_l12_kernel(PICS pp) {
    __shared__ __align__(16) uint32_t lds[kL12Lds<FS>];
    const auto &p   = pp[blockIdx.y];
    const auto &k   = p.ctrl;
    const int             nw = k.num_hme_sa_w, nh = k.num_hme_sa_h;
    const int             n_refs = p.n_ref[0] + (p.n_list == 2 ? p.n_ref[1] : 0);
    int                   b64, item;
    if (!b64_band_item(blockIdx.x, p.b64_w * p.b64_h, n_refs * nw * nh, b64, item) || b64_left_out(p, b64)) return;
    int                   b64x, b64y;
    b64_origin(p, b64, b64x, b64y);
    const int             block_w = min(p.width - b64x, 64), block_h = min(p.height - b64y, 64);
    int                   l = 0, r = item / (nw * nh);
    const int             q = item - r * (nw * nh);
    if (r >= p.n_ref[0]) {
        r -= p.n_ref[0];
        l = 1;
    }
    if (!ref_searched(p, l) || !k.enable_hme_flag || q >= nw * nh) return;
    // both levels' source blocks are fetched while the level-0 results arrive
    uint4 blk2[kL12Blk2<FS>];
    fetch_l12_blocks<FS>(p, lds, blk2, b64x, b64y, block_w, block_h);
    wave_sync();
    l12_region<FS>(p, lds, blk2, b64x, b64y, block_w, block_h, l, r, q, &p.hme_out[(b64 * SVTME_MAX_LISTS + l) * SVTME_MAX_REFS + r]);
}

template <class PICS>
static void launch_hme_items_as(const Batch &b, PICS pics, hipStream_t s, hipEvent_t marks[2]) {
    const PicParams &p0 = b.host[0]; // size, controls and me_type are the batch's
    const int n_refs = b.max_refs(), n_b64 = b.n_b64();
    const int n_reg  = p0.ctrl.num_hme_sa_w * p0.ctrl.num_hme_sa_h;
    const bool fs = p0.ctrl.hme_search_method == 1; // FULL_SAD_SEARCH
    // Small level-0 areas (the fast presets: 16 x 16 positions and up over the 2 x 2 regions): one work item runs all regions
    // of a reference, the two of a row as one window; large ones keep a work item per region (shorter items, shorter tail).
    static const int all_env = getenv("SVTME_L0_ALL_REGIONS") ? atoi(getenv("SVTME_L0_ALL_REGIONS")) : -1; // diagnostic override
    const bool all_regions = !fs && n_reg == 4 && (all_env >= 0 ? all_env != 0 : p0.ctrl.hme_l0_sa.sa_min.width <= 16);
    const dim3 g0(n_b64, all_regions ? 3 : 2 + max(n_reg, 1), n_refs * b.n), g12((n_b64 + 7) / 8 * 8 * n_refs * n_reg, b.n);
    const uint32_t refs_magic = (uint32_t)((0x100000000ull + (uint32_t)n_refs - 1) / (uint32_t)n_refs);
    bool       deps = false;
    for (int i = 0; i < b.n; ++i) deps = deps || b.host[i].dep_pre || b.host[i].dep_l0;
    const int n_phase = deps ? 2 : 1;
    if (p0.ctrl.me_early_exit_th) hipLaunchKernelGGL(hme_zz_kernel<PICS>, dim3((n_b64 + 3) / 4, n_refs, b.n), dim3(HI_THREADS), 0, s, pics);
    for (int phase = 1; phase <= n_phase; ++phase) {
        const int hme_phase = n_phase == 1 ? 0 : phase;
        if (fs) hipLaunchKernelGGL((hme_l0_kernel<true, PICS>), g0, dim3(HI_THREADS), 0, s, pics, n_refs, refs_magic, hme_phase, 0);
        else hipLaunchKernelGGL((hme_l0_kernel<false, PICS>), g0, dim3(HI_THREADS), 0, s, pics, n_refs, refs_magic, hme_phase, all_regions ? 1 : 0);
    }
    if (marks[0]) (void)hipEventRecord(marks[0], s);
    if (p0.ctrl.enable_hme_flag) {
        // sub-sampled searches over 2 x 2 regions: one work item does the four regions of a reference
        static const bool per_region = getenv("SVTME_L12_PER_REGION") != nullptr; // diagnostic: a work item per region
        if (fs) hipLaunchKernelGGL((hme_l12_kernel<true, PICS>), g12, dim3(HI_THREADS_FS), 0, s, pics);
        else if (n_reg == 4 && !per_region && p0.ctrl.hme_l1_sa.height <= 3 && p0.ctrl.hme_l2_sa.height <= 3) launch_hme_l12_regions(b, s);
        else hipLaunchKernelGGL((hme_l12_kernel<false, PICS>), g12, dim3(HI_THREADS), 0, s, pics);
    }
    if (marks[1]) (void)hipEventRecord(marks[1], s);
}

void launch_hme_items(const Batch &b, hipStream_t s, hipEvent_t marks[2]) {
    if (b.n == 1) launch_hme_items_as(b, OnePicture{b.host[0]}, s, marks);
    else launch_hme_items_as(b, ManyPictures{b.dev}, s, marks);
}

// resident workgroups per CU the runtime expects for the two kernels (diagnostic)
void occupancy_hme_items(int out[2]) {
    (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&out[0], hme_l0_kernel<false, OnePicture>, HI_THREADS, 0);
    (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&out[1], hme_l12_kernel<false, OnePicture>, HI_THREADS, 0);
}

} // namespace svtme
