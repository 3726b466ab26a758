// me_kernel.hip - search-centre selection, HME-based reference pruning / search-range adjustment and the integer
// full-pel search with its 85 per-partition minima, for all 64x64 blocks and references of one picture.
//
// Replaces, from Source/Lib/Encoder/Codec/EbMotionEstimation.c:
//   set_final_seach_centre_sb :2117-2316      hme_prune_ref_and_adjust_sr :2373-2410
//   integer_search_b64 :1286-1570 (check_00_center :1170-1243, open_loop_me_fullpel_search_sblock :808-844)
// and the SAD kernels under them: svt_ext_all_sad_calculation_8x8_16x16 :359-388,
// svt_ext_eight_sad_calculation_32x32_64x64 :395-451, the single-position forms :122-229,
// svt_initialize_buffer_32bits (EbMeSadCalculation.c:14).
//
// Three kernels per picture:
//  me_plan_kernel   one wave per (b64, reference): search centre, pruning, search-range divisor, the zero-vs-HME
//                   centre check and the clipped search window -> one MeJob; resets that reference's 85 keys.
//  me_search_kernel one wave per work item = up to 16 "runs" (4 adjacent positions x 8 rows) of one window. Windows
//                   range from 8x3 to 64x64 positions and more, so a picture's few thousand (b64, reference) pairs are
//                   cut into ~10^4 items of similar cost that fill the chip evenly; items of one window merge through
//                   64-bit atomic minima in global memory. The item's reference rows (<= 134 rows x 128 bytes) and the
//                   sampled source rows sit in LDS. A quad of lanes owns a run; each lane of the quad owns one
//                   16-pixel column of the block. The 4 same-parity rows of a pass are walked together, staggered by
//                   one sampled source row, so each reference row is read from LDS once for 16 positions:
//                   512 v_qsad_pk_u16_u8 (4 positions x 4 pixels each) per pass feed packed 16-bit accumulators = the
//                   8x8 SADs (rows 0,2,4,6 of each 8x8, as the reference's sub-sampled search does; doubled at the
//                   end). 16x16 SADs are summed in-lane, 32x32 and 64x64 across the quad with DPP lane exchanges.
//                   Running minima carry the row number in the low key bits, so the reference's tie rule (first
//                   strictly smaller SAD in raster order) is reproduced exactly: keys order by (SAD, y, x).
//  me_finish_kernel keys -> p_sb_best_sad / p_sb_best_mv.
// Where the controls allow no window more than one work item (the fast presets' 16 x 9 positions and smaller, the temporal
// filter's last level, the first pass), the three are one kernel: the item plans its own window, searches it and writes the
// final SADs / vectors itself - no job records, no keys, no atomics, two launches less.
#include "search_common.h"

namespace svtme {

constexpr int ME_THREADS   = 64;
constexpr int ME_PD        = 32; // dwords per staged window row: 64 positions + 63 more pixels, rounded up
constexpr int ME_TILE      = 64; // search positions per tile side
constexpr int ME_ITEM_GROUPS = 2; // a work item's units lie in at most 2 consecutive groups of 8 search rows
constexpr int ME_ROW_BITS    = 4; // keys carry the row within the item
constexpr int ME_WIN_ROWS  = 8 * ME_ITEM_GROUPS + 62; // the item's search rows and the 62 rows under the last one
constexpr int ME_WIN_BYTES = ME_WIN_ROWS * ME_PD * 4;

struct MeShared {
    u64                    hsad[8]; // per (list, ref): search centre, HME SAD, do_ref, search-range divisor
    uint32_t               div[8];
    int16_t                sc_x[8], sc_y[8];
    uint8_t                do_ref[8];
    uint8_t                cand[8]; // this entry has an HME result of its own
};

// z-order index of the 16x16 block in raster (row, col): svt_ext_all_sad_calculation_8x8_16x16_c's `offsets` (:366)
__device__ __forceinline__ int z16(int row, int col) { return ((row >> 1) << 3) | ((col >> 1) << 2) | ((row & 1) << 1) | (col & 1); }

typedef unsigned short us2 __attribute__((ext_vector_type(2)));
// v_pk_min_u16 / v_pk_add_u16 on two 16-bit fields
__device__ __forceinline__ uint32_t pk_min_u16(uint32_t a, uint32_t b) {
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_min(__builtin_bit_cast(us2, a), __builtin_bit_cast(us2, b)));
}
__device__ __forceinline__ uint32_t pk_add_u16(uint32_t a, uint32_t b) {
    return __builtin_bit_cast(uint32_t, (us2)(__builtin_bit_cast(us2, a) + __builtin_bit_cast(us2, b)));
}

// min over 4 keys with the index of the first minimum
__device__ __forceinline__ void min4(const uint32_t (&k)[4], uint32_t &best, uint32_t &idx) {
    best = k[0];
    idx  = 0;
#pragma unroll
    for (uint32_t i = 1; i < 4; ++i)
        if (k[i] < best) {
            best = k[i];
            idx  = i;
        }
}

// Minimum over the lanes of a row of 16 that share lane % 4 (the four quads of the row), for lanes 12..15 of the row.
__device__ __forceinline__ uint32_t row_quads_min(uint32_t v) {
    v = min(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x114 /* row_shr:4 */, 0xF, 0xF, false));
    v = min(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x118 /* row_shr:8 */, 0xF, 0xF, false));
    return v;
}

// x / d for the small divisors of the work-item geometry (below 256; x * d far below 2^32): a shift or a multiplication
__device__ __forceinline__ int div_small(int x, int d) { return d == 16 ? x >> 4 : (int)magic_div((uint32_t)x, (uint32_t)d, div_magic((uint32_t)d)); }
// The same for the units per group of rows (2, 4, 8, 10, 12, 14, 16) and x below 2^12, without the table: x * ceil(2^16 / d) >> 16
__device__ __forceinline__ uint32_t units_inverse(int d) {
    return (d & (d - 1)) == 0 ? 65536u >> (31 - __clz(d)) : (d == 10 ? 6554u : (d == 12 ? 5462u : 4682u));
}

// The plan of reference my_idx (list-0 references first, then list 1; the caller has checked that the picture has it) of
// block b64, whole wave: the window to search, w == 0 when the reference is not searched. KEYS: the job goes to p.me_jobs and
// the reference's 85 keys are reset (the search runs as a kernel of its own); else the caller searches the window itself.
template <bool KEYS, class P>
__device__ __forceinline__ MeJob me_plan_item(const P &p, MeShared &S, int b64, int my_idx) {
    const auto &k   = p.ctrl;
    const int             tid = threadIdx.x;
    int                   b64x, b64y;
    b64_origin(p, b64, b64x, b64y);
    const int             block_w = min(p.width - b64x, 64), block_h = min(p.height - b64y, 64);
    int       my_l = 0, my_r = my_idx;
    if (my_r >= p.n_ref[0]) {
        my_r -= p.n_ref[0];
        my_l = 1;
    }

    // ---- set_final_seach_centre_sb (:2117-2316). Lanes 0-7 fetch the best HME region of one (list, ref) each; lane 0
    // then walks them in order: the running centre / SAD are not reset between references (base-layer list-1
    // references inherit the previous SAD).
    const HmeRefOut *hme = &p.hme_out[b64 * SVTME_MAX_LISTS * SVTME_MAX_REFS];
    int              level = -1;
    if (k.enable_hme_level0_flag && !k.enable_hme_level1_flag && !k.enable_hme_level2_flag) level = 0;
    if (k.enable_hme_level1_flag && !k.enable_hme_level2_flag) level = 1;
    if (k.enable_hme_level2_flag) level = 2;
    if (tid < 8) {
        const int l = tid >> 2, r = tid & 3;
        int16_t   hx = 0, hy = 0;
        u64       hsad = 0;
        bool      cand = false;
        if (l < p.n_list && r < p.n_ref[l] && ref_searched(p, l) && k.enable_hme_flag && level >= 0) {
            const HmeRefOut &h = hme[tid];
            cand = true;
            hx   = h.lx[level][0];
            hy   = h.ly[level][0];
            hsad = h.lsad[level][0];
            for (int sh = 0; sh < k.num_hme_sa_h; ++sh)
                for (int sw = (sh == 0 ? 1 : 0); sw < k.num_hme_sa_w; ++sw)
                    if (h.lsad[level][sw * 2 + sh] < hsad) {
                        hx   = h.lx[level][sw * 2 + sh];
                        hy   = h.ly[level][sw * 2 + sh];
                        hsad = h.lsad[level][sw * 2 + sh];
                    }
        }
        S.sc_x[tid] = hx;
        S.sc_y[tid] = hy;
        S.hsad[tid] = hsad;
        S.cand[tid] = cand;
    }
    __syncthreads();
    if (tid == 0) {
        int16_t hx = 0, hy = 0, cx = 0, cy = 0;
        u64     hsad = 0;
        for (int i = 0; i < 8; ++i) {
            const int l = i >> 2, r = i & 3;
            if (l < p.n_list && r < p.n_ref[l]) {
                if (ref_searched(p, l)) {
                    if (k.enable_hme_flag) {
                        if (S.cand[i]) {
                            hx   = S.sc_x[i];
                            hy   = S.sc_y[i];
                            hsad = S.hsad[i];
                        }
                        cx = hx;
                        cy = hy;
                    }
                } else {
                    cx = cy = 0;
                }
                S.sc_x[i] = cx;
                S.sc_y[i] = cy;
                S.hsad[i] = hsad;
            } else { // init_me_hme_data :2913-2923
                S.sc_x[i] = S.sc_y[i] = 0;
                S.hsad[i]             = 0xFFFFFFFFull;
            }
            S.do_ref[i] = 1;
            S.div[i]    = 1;
        }
        // ---- hme_prune_ref_and_adjust_sr (:2373-2410), over all 2x4 entries
        if (k.enable_hme_flag && p.me_type != SVTME_ME_MCTF && (k.enable_me_sr_adjustment || k.enable_me_hme_ref_pruning)) {
            u64 best = ~0ull;
            for (int i = 0; i < 8; ++i) best = S.hsad[i] < best ? S.hsad[i] : best;
            const uint32_t th = k.prune_ref_if_hme_sad_dev_bigger_than_th;
            for (int i = 0; i < 8; ++i) {
                const int r = i & 3;
                if (k.enable_me_hme_ref_pruning && (!k.protect_closest_refs || r > 0) && th != 0xFFFF &&
                    (S.hsad[i] - best) * 100 > (u64)th * best)
                    S.do_ref[i] = 0;
                if (k.enable_me_sr_adjustment) {
                    if (abs(S.sc_x[i]) <= k.reduce_me_sr_based_on_mv_length_th && abs(S.sc_y[i]) <= k.reduce_me_sr_based_on_mv_length_th &&
                        S.hsad[i] < k.stationary_hme_sad_abs_th)
                        S.div[i] = k.stationary_me_sr_divisor;
                    else if (S.hsad[i] < k.reduce_me_sr_based_on_hme_sad_abs_th)
                        S.div[i] = k.me_sr_divisor_for_low_hme_sad;
                }
            }
        }
    }
    __syncthreads();
    // the workgroup of the first reference publishes the per-b64 tables
    if (my_idx == 0 && tid < 8) {
        const int            l = tid >> 2, r = tid & 3;
        // one SearchResults record (EbMotionEstimationContext.h:330-337) as three whole 8-byte words, so that the padding
        // bytes of the record are defined (zero) as well
        static_assert(sizeof(svtme_search_result) == 24 && offsetof(svtme_search_result, list_i) == 0 &&
                          offsetof(svtme_search_result, ref_i) == 1 && offsetof(svtme_search_result, hme_sc_x) == 2 &&
                          offsetof(svtme_search_result, hme_sc_y) == 4 && offsetof(svtme_search_result, hme_sad) == 8 &&
                          offsetof(svtme_search_result, do_ref) == 16 && offsetof(svtme_search_result, tf_me_exit) == 17,
                      "svtme_search_result must keep the layout of the encoder's SearchResults");
        // temporal filter: a block that is already well predicted is not searched (:2958-2962)
        const u64 tf_me_exit = p.me_type == SVTME_ME_MCTF && tid == 0 && S.hsad[0] < p.tf_me_exit_th;
        u64      *o = reinterpret_cast<u64 *>(&p.search_results[b64 * 8 + tid]);
        o[0] = (u64)l | ((u64)r << 8) | ((u64)(uint16_t)S.sc_x[tid] << 16) | ((u64)(uint16_t)S.sc_y[tid] << 32);
        o[1] = S.hsad[tid];
        o[2] = (u64)S.do_ref[tid] | (tf_me_exit << 8);
        p.sr_divisor[b64 * 8 + tid] = S.div[tid];
        const bool have = l < p.n_list && r < p.n_ref[l];
        p.zz_sad[b64 * 8 + tid] = have ? hme[tid].zz_sad : ~0u;
    }
    const int my = my_l * 4 + my_r;
    MeJob    *job = &p.me_jobs[b64 * 8 + my];
    if (KEYS)
        for (int pu = tid; pu < SVTME_PU_COUNT; pu += ME_THREADS) p.me_keys[(size_t)(b64 * 8 + my) * SVTME_PU_COUNT + pu] = ~0ull;
    if (!S.do_ref[my] || (p.me_type == SVTME_ME_MCTF && S.hsad[0] < p.tf_me_exit_th) || b64_left_out(p, b64)) {
        if (KEYS && tid == 0) *job = MeJob{0, 0, 0, 0};
        return MeJob{0, 0, 0, 0};
    }

    // ---- integer_search_b64 for (my_l, my_r)
    const DevPlane  cur = load_plane(p.cur.lvl[2]);
    const DevPlane  ref = load_plane(p.ref[my_l][my_r].lvl[2]);
    int16_t         cx = S.sc_x[my], cy = S.sc_y[my];
    // the growth of the area with the distance is slowed down, except for the temporal filter (:1338-1341)
    const int       dist = p.me_type == SVTME_ME_MCTF ? p.dist[my_l][my_r] : scaled_dist(p.dist[my_l][my_r]);
    int16_t         w = (int16_t)min((int)(int16_t)k.me_sa.sa_min.width * dist, (int)k.me_sa.sa_max.width);
    int16_t         h = (int16_t)min((int)(int16_t)k.me_sa.sa_min.height * dist, (int)k.me_sa.sa_max.height);
    w = (int16_t)((max(1, (int)((uint32_t)w / S.div[my])) + 7) & ~0x07);
    h = (int16_t)max(3, (int)((uint32_t)h / S.div[my]));
    const int16_t  w0 = w, h0 = h;
    const uint32_t zz = hme[my_l * SVTME_MAX_REFS + my_r].zz_sad;
    if (k.me_early_exit_th) {
        if (zz < k.me_early_exit_th / 6) w = h = 1;
    } else {
        u64 best_hme_sad = ~0ull;
        int accurate     = 1;
        // the first pass may bypass the zero-vector check (:1363-1364)
        if (!(p.me_type == SVTME_ME_FIRST_PASS && p.bypass_zz_check) && (cx != 0 || cy != 0) && p.is_ref) {
            // check_00_center :1170-1243
            if (b64x + cx < -63) cx = (int16_t)(-63 - b64x);
            if (b64x + cx > ref.width - 1) cx = (int16_t)(cx - ((b64x + cx) - (ref.width - 1)));
            if (b64y + cy < -63) cy = (int16_t)(-63 - b64y);
            if (b64y + cy > ref.height - 1) cy = (int16_t)(cy - ((b64y + cy) - (ref.height - 1)));
            // both SADs (every other row) in one pass of this single wave: lane = one dword column of 8 sampled rows, all
            // loads in flight together
            uint32_t zero_sad = 0, hme_sad = 0;
            {
                const int d = tid & 15, bwd = block_w >> 2, rows = block_h >> 1;
                uint32_t  sv[8], zv[8], hv[8];
#pragma unroll
                for (int kk = 0; kk < 8; ++kk) {
                    const int  rr = (tid >> 4) + 4 * kk;
                    const bool on = d < bwd && rr < rows;
                    const int  rc = on ? rr : 0, dc = on ? d : 0; // idle lanes repeat a valid address
                    const uint8_t *sp = cur.pix + (ptrdiff_t)(b64y + 2 * rc) * cur.stride + b64x + 4 * dc;
                    const uint8_t *fp = ref.pix + (ptrdiff_t)(b64y + 2 * rc) * ref.stride + b64x + 4 * dc;
                    sv[kk] = *reinterpret_cast<const uint32_t *>(sp);
                    zv[kk] = *reinterpret_cast<const uint32_t *>(fp);
                    __builtin_memcpy(&hv[kk], fp + (ptrdiff_t)cy * ref.stride + cx, 4);
                }
#pragma unroll
                for (int kk = 0; kk < 8; ++kk) {
                    const bool on = d < bwd && (tid >> 4) + 4 * kk < rows;
                    zero_sad = __builtin_amdgcn_sad_u8(sv[kk], on ? zv[kk] : sv[kk], zero_sad);
                    hme_sad  = __builtin_amdgcn_sad_u8(sv[kk], on ? hv[kk] : sv[kk], hme_sad);
                }
                zero_sad = wave_sum(zero_sad) << 1;
                hme_sad  = wave_sum(hme_sad) << 1;
            }
            best_hme_sad           = hme_sad;
            if (tid == 0) count_ops(p, 0, 2ull * (block_h >> 1) * block_w);
            // costs are SAD << COST_PRECISION; the temporal filter weights a poor centre by TF_HME_MV_COST_WEIGHT
            // (125 %) when its SAD exceeds TF_HME_MV_SAD_TH (512), :37-38, :1231-1241
            u64 zero_cost = (u64)zero_sad << 8, hme_cost = (u64)hme_sad << 8;
            if (p.me_type == SVTME_ME_MCTF && hme_sad > 512) hme_cost = hme_cost * 125 / 100;
            if (zero_cost <= hme_cost) cx = cy = 0;
            if (cx == 0 && cy == 0) accurate = 0;
        }
        if (k.enable_me_sr_adjustment == 2) {
            if ((accurate && best_hme_sad < 24 * 24) || (p.is_ref && S.hsad[my] < 24 * 24)) h = (int16_t)(h / 2);
            // :1390-1399: a good 64x64 match in the first reference halves the other references' areas. The host runs
            // the first reference's search to completion before this plan is launched for the others.
            if (my_idx != 0) {
                const uint32_t sad00 = (uint32_t)(p.me_keys[(size_t)(b64 * 8) * SVTME_PU_COUNT] >> 32) << 1;
                if (sad00 < 5000 && h == h0 && w == w0) {
                    h = (int16_t)(h >> 1);
                    w = (int16_t)(w >> 1);
                }
            }
        }
    }
    int16_t org_x = (int16_t)(cx - (w >> 1)), org_y = (int16_t)(cy - (h >> 1));
    // restricted motion vectors: the block's tile instead of the padded picture bounds the window (:1405-1498); the
    // open-loop search's one tile is the picture, the other searches may come with the sequence's tiles (:1408-1410)
    int lo_x = -63, lo_y = -63, hi_x = p.width, hi_y = p.height;
    if (p.restricted) {
        lo_x = lo_y = 0;
        if (p.tiles) {
            const uint4 t = *reinterpret_cast<const uint4 *>(&p.tiles[b64]);
            lo_x = (int)t.x, lo_y = (int)t.y, hi_x = (int)t.z, hi_y = (int)t.w;
        }
    }
    clip_axis(b64x, org_x, w, lo_x, hi_x);
    w = (int16_t)(w < 8 ? w : (w & ~0x07));
    clip_axis(b64y, org_y, h, lo_y, hi_y);

    if (tid == 0) {
        if (KEYS) *job = MeJob{org_x, org_y, w, h};
        count_ops(p, 5, (unsigned long long)w * h * 2048ull);
        if (me_item_count(w, h) > p.me_max_items) count_ops(p, 7, 1); // cannot happen: the host sized the grid from sa_max
    }
    return MeJob{org_x, org_y, w, h};
}

template <class PICS>
__global__ void __launch_bounds__(ME_THREADS) me_plan_kernel(PICS pp, int me_ref0) {
    __shared__ MeShared S;
    const auto &p = pp[blockIdx.z]; // grid: x = 64x64 block, y = reference, z = picture
    // me_ref0: first reference (list-0 references first, then list 1) this launch covers; the grid covers the batch's
    // longest reference list
    const int my_idx = me_ref0 + (int)blockIdx.y;
    if (my_idx >= p.n_ref[0] + (p.n_list == 2 ? p.n_ref[1] : 0)) return;
    (void)me_plan_item<true>(p, S, (int)blockIdx.x, my_idx);
}

// The walk and the merge of a work item of at most 16 / NS units (the small windows of the fast presets: 8 x 3, 8 x 6, 16 x 3
// positions: split units, one row parity of a position quad each). With one unit per quad of lanes most of the wave would
// idle, so the block's 32 sampled rows are cut into NS slices of R = 32 / NS rows and every unit is walked by NS quads of
// lanes, a slice each: R + 3 steps instead of 35. A slice holds whole 8x8 and 16x16 partitions (R = 8: one row of 16x16
// blocks; R = 16: two); the 16-row column sums of the slices meet through lane exchanges for the 32x32 and 64x64 sums, which
// every slice then ranks (the same keys in each). Leaves the item's minima in win[4][85] as me_search_kernel's own walk does.
template <int NS>
__device__ __forceinline__ void me_walk_sliced(uint32_t *win, const uint32_t *src_rows, int u0, int u_end, uint32_t u_inv, int U, int g0, int rows,
                                               int w, int tx0) {
    constexpr int ROW_BITS = ME_ROW_BITS;
    constexpr int R = 32 / NS, UQ = 16 / NS; // sampled rows per slice, units (quads of lanes) per slice
    const int     tid = threadIdx.x, quad = tid >> 2, col = tid & 3;
    const int     sl = quad / UQ, uq = quad & (UQ - 1);
    const int     u     = u0 + uq;
    const bool    valid = u < u_end;
    const int     ug    = valid ? (int)(((uint32_t)u * u_inv) >> 16) : g0, uv = valid ? u - ug * U : 0;
    const int     grp   = ug - g0;
    const int     xg = uv >> 1, par = uv & 1;
    const int     ya = 8 * grp, yb = min(rows, ya + 8);
    const int     yfirst = ya + par;
    const int     n_pos  = valid ? min(4, max(0, (yb - yfirst + 1) >> 1)) : 0; // same in the lanes of a unit
    const bool    odd = (col & 1) != 0;
    uint32_t      b8[R / 4][2][2], b16[R / 8][4], b32[2][2], b64k[2];
#pragma unroll
    for (int a = 0; a < R / 4; ++a)
#pragma unroll
        for (int c = 0; c < 2; ++c) b8[a][c][0] = b8[a][c][1] = 0xFFFFFFFFu;
#pragma unroll
    for (int a = 0; a < R / 8; ++a)
#pragma unroll
        for (int i = 0; i < 4; ++i) b16[a][i] = 0xFFFFFFFFu;
    b32[0][0] = b32[0][1] = b32[1][0] = b32[1][1] = b64k[0] = b64k[1] = 0xFFFFFFFFu;
    uint32_t ypk[4], yrk[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const uint32_t yr = (uint32_t)(par + 2 * t);
        ypk[t]            = t < n_pos ? yr * 0x10001u : 0xFFFFFFFFu;
        yrk[t]            = t < n_pos ? (yr + 8 * grp) << 6 : 0xFFFFFFFFu;
    }
    u64      acc0[4], acc1[4];
    uint32_t t16[4][2], half[4][2], sw[4][4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        acc0[t] = acc1[t] = 0;
        t16[t][0] = t16[t][1] = half[t][0] = half[t][1] = 0;
    }
    // position t meets the slice's sampled row m at step s = t + m; the slice's first window row lies 2 R sl rows under
    // the unit's first
    const uint32_t *sbase = &src_rows[col * 4 + R * sl * 16];
    int             ie = (yfirst + 2 * R * sl) * ME_PD + xg + col * 4, io = ie + 1;
    uint32_t        n01[2], n12[2], n23[2], n34[2], nsw[4];
    auto            fetch = [&](auto s_c) {
        constexpr int s = decltype(s_c)::value;
        if constexpr ((s & 3) == 0) {
            if constexpr (s > 0) {
                ie += 8 * ME_PD;
                io += 8 * ME_PD;
            }
            asm volatile("" : "+v"(ie));
            asm volatile("" : "+v"(io));
        }
        constexpr int c = (s & 3) * 2 * ME_PD;
        n01[0] = win[ie + c], n01[1] = win[ie + c + 1];
        n12[0] = win[io + c], n12[1] = win[io + c + 1];
        n23[0] = win[ie + c + 2], n23[1] = win[ie + c + 3];
        n34[0] = win[io + c + 2], n34[1] = win[io + c + 3];
        if constexpr (s < R) {
            const uint4 v = *reinterpret_cast<const uint4 *>(sbase + s * 16);
            nsw[0] = v.x, nsw[1] = v.y, nsw[2] = v.z, nsw[3] = v.w;
        }
    };
    fetch(std::integral_constant<int, 0>{});
    static_for<0, R + 3>([&](auto s_c) {
        constexpr int s = decltype(s_c)::value;
        const u64 p01 = ((u64)n01[1] << 32) | n01[0], p12 = ((u64)n12[1] << 32) | n12[0];
        const u64 p23 = ((u64)n23[1] << 32) | n23[0], p34 = ((u64)n34[1] << 32) | n34[0];
        if constexpr (s < R) sw[s & 3][0] = nsw[0], sw[s & 3][1] = nsw[1], sw[s & 3][2] = nsw[2], sw[s & 3][3] = nsw[3];
        if constexpr (s + 1 < R + 3) fetch(std::integral_constant<int, s + 1>{});
        __builtin_amdgcn_sched_barrier(0);
        static_for<0, 4>([&](auto t_c) {
            constexpr int t = decltype(t_c)::value;
            constexpr int m = s - t; // the slice's sampled row of position t at this step
            if constexpr (m >= 0 && m < R) {
                acc0[t] = __builtin_amdgcn_qsad_pk_u16_u8(p01, sw[m & 3][0], (m & 3) == 0 ? (u64)0 : acc0[t]);
                acc0[t] = __builtin_amdgcn_qsad_pk_u16_u8(p12, sw[m & 3][1], acc0[t]);
                acc1[t] = __builtin_amdgcn_qsad_pk_u16_u8(p23, sw[m & 3][2], (m & 3) == 0 ? (u64)0 : acc1[t]);
                acc1[t] = __builtin_amdgcn_qsad_pk_u16_u8(p34, sw[m & 3][3], acc1[t]);
                if constexpr ((m & 3) == 3) {
                    constexpr int  by = m >> 2, a = by >> 1; // 8x8 block row / 16x16 block row within the slice
                    const uint32_t l0 = (uint32_t)acc0[t], h0 = (uint32_t)(acc0[t] >> 32);
                    const uint32_t l1 = (uint32_t)acc1[t], h1 = (uint32_t)(acc1[t] >> 32);
                    b8[by][0][0] = pk_min_u16(b8[by][0][0], (l0 << 3) | ypk[t]);
                    b8[by][0][1] = pk_min_u16(b8[by][0][1], (h0 << 3) | ypk[t]);
                    b8[by][1][0] = pk_min_u16(b8[by][1][0], (l1 << 3) | ypk[t]);
                    b8[by][1][1] = pk_min_u16(b8[by][1][1], (h1 << 3) | ypk[t]);
                    if constexpr ((by & 1) == 0) {
                        t16[t][0] = pk_add_u16(l0, l1);
                        t16[t][1] = pk_add_u16(h0, h1);
                    } else {
                        const uint32_t d0 = pk_add_u16(t16[t][0], pk_add_u16(l0, l1));
                        const uint32_t d1 = pk_add_u16(t16[t][1], pk_add_u16(h0, h1));
                        b16[a][0] = min(b16[a][0], (d0 << 16) | yrk[t]);
                        b16[a][1] = min(b16[a][1], (d0 & 0xFFFF0000u) | yrk[t]);
                        b16[a][2] = min(b16[a][2], (d1 << 16) | yrk[t]);
                        b16[a][3] = min(b16[a][3], (d1 & 0xFFFF0000u) | yrk[t]);
                        if constexpr ((a & 1) == 0) {
                            half[t][0] = d0;
                            half[t][1] = d1;
                        } else {
                            half[t][0] = pk_add_u16(half[t][0], d0);
                            half[t][1] = pk_add_u16(half[t][1], d1);
                        }
                        if constexpr (a == R / 8 - 1) {
                            // the slice's column sums of position t are complete: 32-row sums (<= 65280) of the block's top
                            // and bottom half from the slices, the same in every slice of the unit
                            uint32_t q0 = half[t][0], q1 = half[t][1];
                            if constexpr (NS == 4) {
                                q0 = pk_add_u16(q0, (uint32_t)__shfl_xor((int)q0, 16));
                                q1 = pk_add_u16(q1, (uint32_t)__shfl_xor((int)q1, 16));
                            }
                            const uint32_t o0 = (uint32_t)__shfl_xor((int)q0, 32), o1 = (uint32_t)__shfl_xor((int)q1, 32);
                            const bool     low = tid >= 32; // this slice lies in the block's bottom half
                            const uint32_t top0 = low ? o0 : q0, top1 = low ? o1 : q1, bot0 = low ? q0 : o0, bot1 = low ? q1 : o1;
                            const uint32_t tk = odd ? top1 : top0, tr = quad_xor1(odd ? top0 : top1);
                            const uint32_t bk = odd ? bot1 : bot0, br = quad_xor1(odd ? bot0 : bot1);
#pragma unroll
                            for (int j = 0; j < 2; ++j) {
                                const uint32_t tp = j ? (tk >> 16) + (tr >> 16) : (tk & 0xFFFFu) + (tr & 0xFFFFu);
                                const uint32_t bt = j ? (bk >> 16) + (br >> 16) : (bk & 0xFFFFu) + (br & 0xFFFFu);
                                uint32_t       all = tp + bt;
                                all += quad_xor2(all);
                                b32[0][j] = min(b32[0][j], (tp << (6 + ROW_BITS)) | yrk[t]);
                                b32[1][j] = min(b32[1][j], (bt << (6 + ROW_BITS)) | yrk[t]);
                                b64k[j]   = min(b64k[j], (all << (6 + ROW_BITS)) | yrk[t]);
                            }
                        }
                    }
                }
            }
        });
    });
    // ---- merge, as in me_search_kernel: a row of 16 lanes holds four units of one slice; the rows' results meet in
    // win[4][85], where a row fills in the partitions of its slice and leaves the others all ones
    const uint32_t xbase = (uint32_t)(xg * 4);
    const bool     narrow = w < 8;
    uint32_t       xlab[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) xlab[i] = (narrow && (int)(tx0 + xbase + i) >= w) ? 0xFFFFFFFFu : xbase + i;
    const bool writer = (tid & 12) == 12;
    wave_sync();
    for (int i = tid; i < 4 * SVTME_PU_COUNT; i += ME_THREADS) win[i] = 0xFFFFFFFFu;
    wave_sync();
    uint32_t *part = win + (tid >> 4) * SVTME_PU_COUNT;
#pragma unroll
    for (int byl = 0; byl < R / 4; ++byl)
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const uint32_t d0 = b8[byl][c][0], d1 = b8[byl][c][1];
            const uint32_t k0 = ((d0 & 0xFFFFu) << 6) | xlab[0], k1 = ((d0 >> 16) << 6) | xlab[1];
            const uint32_t k2 = ((d1 & 0xFFFFu) << 6) | xlab[2], k3 = ((d1 >> 16) << 6) | xlab[3];
            const uint32_t km = min(min(k0, k1), min(k2, k3));
            const uint32_t k  = row_quads_min(((km & 0xFFFFFE00u) << (ROW_BITS - 3)) | ((km & 0x1FFu) | ((uint32_t)grp << 9)));
            const int      by = byl + (R / 4) * sl, bx8 = col * 2 + c;
            const int      pu = 21 + z16(by >> 1, bx8 >> 1) * 4 + ((by & 1) << 1) + (bx8 & 1);
            if (writer) part[pu] = k;
        }
#pragma unroll
    for (int al = 0; al < R / 8; ++al) {
        const uint32_t k = row_quads_min(min(min(b16[al][0] | xlab[0], b16[al][1] | xlab[1]), min(b16[al][2] | xlab[2], b16[al][3] | xlab[3])));
        if (writer) part[5 + z16(al + (R / 8) * sl, col)] = k;
    }
    const uint32_t xl0 = odd ? xlab[2] : xlab[0], xl1 = odd ? xlab[3] : xlab[1];
#pragma unroll
    for (int a = 0; a < 2; ++a) {
        uint32_t k = min(b32[a][0] | xl0, b32[a][1] | xl1);
        k          = row_quads_min(min(k, quad_xor1(k)));
        if (writer && !odd) part[1 + a * 2 + (col >> 1)] = k;
    }
    {
        uint32_t k = min(b64k[0] | xl0, b64k[1] | xl1);
        k          = row_quads_min(min(k, quad_xor1(k)));
        if (writer && col == 0) part[0] = k;
    }
}

// FUSED: every window of the batch is one work item (me_max_items == 1), which plans it, searches it and writes the final
// results; me_plan_kernel has not run and me_finish_kernel<false> only tidies up.
template <class PICS, bool FUSED>
__global__ void __launch_bounds__(ME_THREADS, 2) me_search_kernel(PICS pp, int grid_refs, uint32_t grid_refs_magic, int ref_first, int ref_last) {
    constexpr int ROW_BITS = ME_ROW_BITS;
    __shared__ MeShared               S; // (FUSED only)
    __shared__ __align__(16) uint32_t src_rows[32 * 16]; // rows 0,2,..,62 of the source block, 16 dwords each
    __shared__ __align__(16) uint32_t win[ME_WIN_BYTES / 4];
    // the grid's slowest dimension is (picture, reference): the work items of one picture are dispatched together, so that
    // the workgroups in flight read a picture or two's planes rather than the whole batch's (what L2 holds)
    // (the picture's index by a multiplication: grid_refs_magic = ceil(2^32 / grid_refs))
    const int   pic_i = grid_refs == 1 ? (int)blockIdx.z : (int)__umulhi(blockIdx.z, grid_refs_magic);
    const auto &p = pp[pic_i];
    const int tid = threadIdx.x;
    const int b64 = blockIdx.x;
    int       b64x, b64y;
    b64_origin(p, b64, b64x, b64y);
    // references in reverse order: the closest references (smallest windows, cheapest items) are dispatched last and
    // keep the kernel's tail short. The grid covers the batch's longest reference list; the launch covers references
    // ref_first .. ref_last (list-0 references first, then list 1).
    int       my_l = 0, my_r = p.n_ref[0] + (p.n_list == 2 ? p.n_ref[1] : 0) - 1 - ((int)blockIdx.z - pic_i * grid_refs);
    if (my_r < ref_first || my_r > ref_last) return;
    const int my_idx = my_r;
    if (my_r >= p.n_ref[0]) {
        my_r -= p.n_ref[0];
        my_l = 1;
    }
    const int   my  = my_l * 4 + my_r;
    MeJob       job;
    if constexpr (FUSED) job = me_plan_item<false>(p, S, b64, my_idx);
    else job = p.me_jobs[b64 * 8 + my];
    const int   w = job.w, h = job.h, org_x = job.org_x, org_y = job.org_y;
    // where this reference's results go: output plane = reference, list-0 references first
    uint32_t *const out_sad = FUSED ? p.best_sad + (size_t)(my_idx * p.b64_w * p.b64_h + b64) * SVTME_PU_COUNT : nullptr;
    uint32_t *const out_mv  = FUSED ? p.best_mv + (size_t)(my_idx * p.b64_w * p.b64_h + b64) * SVTME_PU_COUNT : nullptr;
    if (w <= 0) {
        if constexpr (FUSED) // not searched (pruned, the temporal filter's exit, a block the first pass leaves out): zeros
            for (int pu = tid; pu < SVTME_PU_COUNT; pu += ME_THREADS) out_sad[pu] = out_mv[pu] = 0;
        return;
    }
    // ---- this item (me_item_units in svtme_device.h): a 64-position column chunk of the window is cut into units - the
    // 4 positions x 8 rows of a position quad and a group of 8 rows, or one row parity of them where the chunk has few
    // quads - numbered group by group; an item takes the next 16 of them, one per quad of lanes, from up to two groups of
    // rows. A window of 48 x 48 positions fills its items' lanes that way, where an item per group would leave a quarter
    // of them idle.
    const int G = (h + 7) >> 3, full = (w - 1) / ME_TILE;
    int       chunk = full, item_i = (int)blockIdx.y - full * G;
    if (item_i < 0) { // one of the 64-position chunks before the last: an item per group of rows
        chunk = div_small((int)blockIdx.y, G);
        item_i = (int)blockIdx.y - chunk * G;
    }
    const int tx0 = chunk * ME_TILE;
    int       n_xg, U, UPI;
    me_item_units(min(ME_TILE, w - tx0), n_xg, U, UPI);
    const bool split = n_xg <= ME_THREADS / 8; // few quads: each row parity of a (quad, group) is a unit of its own
    const int  n_unit_all = G * U, u0 = item_i * UPI;
    if (u0 >= n_unit_all) return;
    const int u_end = min(u0 + UPI, n_unit_all);
    const uint32_t u_inv = units_inverse(U);
    const int      g0 = (int)(((uint32_t)u0 * u_inv) >> 16), g_last = (int)(((uint32_t)(u_end - 1) * u_inv) >> 16);
    const int ty0 = 8 * g0, rows = min(h - ty0, 8 * (g_last - g0 + 1)); // search rows staged for this item

    const DevPlane cur = load_plane(p.cur.lvl[2]);
    const DevPlane ref = load_plane(p.ref[my_l][my_r].lvl[2]);
    // ---- open_loop_me_fullpel_search_sblock
    {
        uint4 v[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int i = tid + ME_THREADS * t; // 16-byte piece: row i / 4, piece i % 4
            __builtin_memcpy(&v[t], cur.pix + (ptrdiff_t)(b64y + 2 * (i >> 2)) * cur.stride + b64x + 16 * (i & 3), 16);
        }
#pragma unroll
        for (int t = 0; t < 2; ++t) reinterpret_cast<uint4 *>(src_rows)[tid + ME_THREADS * t] = v[t];
    }
    const int       quad = tid >> 2, col = tid & 3;
    const uint32_t *sbase = &src_rows[col * 4];

    {
        {
            __syncthreads();
            {
                // stage the window: 8 rows per trip, lane = 16 bytes of a row (unaligned 16-byte loads)
                const int      nq = (n_xg + 16 + 3) >> 2, nrow = rows + 62, d = tid & 7;
                const uint8_t *g = ref.pix + (ptrdiff_t)(b64y + org_y + ty0 + (tid >> 3)) * ref.stride + (b64x + org_x + tx0 + 16 * d);
                uint4         *dst = reinterpret_cast<uint4 *>(win) + (tid >> 3) * (ME_PD / 4) + d;
                if (d < nq) {
                    // nrow <= 78: at most 10 trips. The 9 trips that an item of one group of rows needs are all issued before
                    // the first store; trips past the last row repeat it (same value into the same slot), so nothing is
                    // conditional. Items whose units lie in two groups take one more.
                    constexpr int TRIPS = (8 + 62 + 7) / 8, MORE = (ME_WIN_ROWS + 7) / 8 - TRIPS;
                    uint4         v[TRIPS];
#pragma unroll
                    for (int t = 0; t < TRIPS; ++t) {
                        const int dr = min(8 * t, nrow - 1 - (tid >> 3));
                        __builtin_memcpy(&v[t], g + (ptrdiff_t)dr * ref.stride, 16);
                    }
#pragma unroll
                    for (int t = 0; t < TRIPS; ++t) dst[min(8 * t, nrow - 1 - (tid >> 3)) * (ME_PD / 4)] = v[t];
                    if (nrow > 8 * TRIPS) {
                        uint4 e[MORE];
#pragma unroll
                        for (int t = 0; t < MORE; ++t) {
                            const int dr = min(8 * (TRIPS + t), nrow - 1 - (tid >> 3));
                            __builtin_memcpy(&e[t], g + (ptrdiff_t)dr * ref.stride, 16);
                        }
#pragma unroll
                        for (int t = 0; t < MORE; ++t) dst[min(8 * (TRIPS + t), nrow - 1 - (tid >> 3)) * (ME_PD / 4)] = e[t];
                    }
                }
            }
            __syncthreads();
            const int  n_it   = split ? 1 : 2;
            // the small windows of the fast presets: every unit walked by two or four quads of lanes (me_walk_sliced)
            if (FUSED && u_end - u0 <= 4) me_walk_sliced<4>(win, src_rows, u0, u_end, u_inv, U, g0, rows, w, tx0);
            else if (FUSED && u_end - u0 <= 8) me_walk_sliced<2>(win, src_rows, u0, u_end, u_inv, U, g0, rows, w, tx0);
            else {
                // this quad's unit: group of rows `grp` (counted from the item's first), position quad xg, row parity
                const int  u     = u0 + quad;
                const bool valid = quad < UPI && u < u_end;
                const int  ug    = valid ? (int)(((uint32_t)u * u_inv) >> 16) : g0, uv = valid ? u - ug * U : 0;
                const int  grp   = ug - g0;
                const int  xg    = split ? uv >> 1 : uv;
                const int  par0  = split ? (uv & 1) : 0;
                const int  ya = 8 * grp, yb = min(rows, ya + 8);
                // running minima of this lane. 8x8: packed 16-bit keys SAD * 8 + row-in-run, two positions per dword;
                // 16x16: SAD << 16 | row << 6; larger: SAD << 9 | row << 6. A position that does not exist carries an
                // all-ones row mask, which makes its keys all ones.
                uint32_t b8[8][2][2];
                uint32_t b16[4][4], b32[2][2], b64k[2]; // (32x32 / 64x64: this lane's two of the quad's four positions)
#pragma unroll
                for (int a = 0; a < 8; ++a)
#pragma unroll
                    for (int c = 0; c < 2; ++c) b8[a][c][0] = b8[a][c][1] = 0xFFFFFFFFu;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
#pragma unroll
                    for (int a = 0; a < 4; ++a) b16[a][i] = 0xFFFFFFFFu;
                    b32[i >> 1][i & 1] = 0xFFFFFFFFu;
                }
                b64k[0] = b64k[1] = 0xFFFFFFFFu;
                const bool odd = (col & 1) != 0;
                // The 4 same-parity rows of a pass are walked together, staggered by one sampled source row: position t
                // meets source row m at step s = t + m, while the lane holds reference row (first row of the pass + 2 s) in
                // registers. Every reference row is read from LDS once for 4 positions and every source row once (kept
                // for 4 steps).
                for (int it = 0; it < n_it; ++it) {
                    const int par    = par0 + it;
                    const int yfirst = ya + par;
                    const int n_pos  = valid ? min(4, max(0, (yb - yfirst + 1) >> 1)) : 0; // same in the 4 lanes of a quad
                    if (__ballot(n_pos > 0) == 0) continue;
                    const uint32_t *wrow = win + yfirst * ME_PD + xg + col * 4;
                    uint32_t        ypk[4], yrk[4];
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const uint32_t yr = (uint32_t)(par + 2 * t); // within the quad's 8 rows; the 8x8 keys get the group at the merge
                        ypk[t]            = t < n_pos ? yr * 0x10001u : 0xFFFFFFFFu;
                        yrk[t]            = t < n_pos ? (yr + 8 * grp) << 6 : 0xFFFFFFFFu; // (bits 0..5 take the column at the merge)
                    }
                    u64      acc0[4], acc1[4];
                    uint32_t t16[4][2], half[4][2], topv[4][2], sw[4][4];
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        acc0[t] = acc1[t] = 0;
                        t16[t][0] = t16[t][1] = half[t][0] = half[t][1] = topv[t][0] = topv[t][1] = 0;
                    }
                    // Row loads run one step ahead of the arithmetic (sched_barrier keeps them there), so the LDS latency of
                    // step s + 1 hides under the 16 QSADs of step s. The two dword indices are re-derived through an empty
                    // asm every 4 steps: the compiler then addresses 4 steps' rows as immediates off one register instead
                    // of one add per load, and cannot merge the overlapping even / odd pairs into copies.
                    int      ie = (int)(wrow - win), io = ie + 1;
                    uint32_t n01[2], n12[2], n23[2], n34[2], nsw[4];
                    auto     fetch = [&](auto s_c) {
                        constexpr int s = decltype(s_c)::value;
                        if constexpr ((s & 3) == 0) {
                            if constexpr (s > 0) {
                                ie += 8 * ME_PD;
                                io += 8 * ME_PD;
                            }
                            asm volatile("" : "+v"(ie));
                            asm volatile("" : "+v"(io));
                        }
                        constexpr int c = (s & 3) * 2 * ME_PD;
                        n01[0] = win[ie + c], n01[1] = win[ie + c + 1];
                        n12[0] = win[io + c], n12[1] = win[io + c + 1];
                        n23[0] = win[ie + c + 2], n23[1] = win[ie + c + 3];
                        n34[0] = win[io + c + 2], n34[1] = win[io + c + 3];
                        if constexpr (s < 32) {
                            const uint4 v = *reinterpret_cast<const uint4 *>(sbase + s * 16);
                            nsw[0] = v.x, nsw[1] = v.y, nsw[2] = v.z, nsw[3] = v.w;
                        }
                    };
                    fetch(std::integral_constant<int, 0>{});
                    static_for<0, 35>([&](auto s_c) {
                        constexpr int s = decltype(s_c)::value;
                        const u64 p01 = ((u64)n01[1] << 32) | n01[0], p12 = ((u64)n12[1] << 32) | n12[0];
                        const u64 p23 = ((u64)n23[1] << 32) | n23[0], p34 = ((u64)n34[1] << 32) | n34[0];
                        if constexpr (s < 32) sw[s & 3][0] = nsw[0], sw[s & 3][1] = nsw[1], sw[s & 3][2] = nsw[2], sw[s & 3][3] = nsw[3];
                        if constexpr (s + 1 < 35) fetch(std::integral_constant<int, s + 1>{});
                        __builtin_amdgcn_sched_barrier(0);
                        static_for<0, 4>([&](auto t_c) {
                            constexpr int t = decltype(t_c)::value;
                            constexpr int m = s - t; // sampled source row (source row 2m) of position t at this step
                            if constexpr (m >= 0 && m <= 31) {
                                // (the first row of an 8x8 block row starts its sums from the constant 0: no registers to clear)
                                acc0[t] = __builtin_amdgcn_qsad_pk_u16_u8(p01, sw[m & 3][0], (m & 3) == 0 ? (u64)0 : acc0[t]);
                                acc0[t] = __builtin_amdgcn_qsad_pk_u16_u8(p12, sw[m & 3][1], acc0[t]);
                                acc1[t] = __builtin_amdgcn_qsad_pk_u16_u8(p23, sw[m & 3][2], (m & 3) == 0 ? (u64)0 : acc1[t]);
                                acc1[t] = __builtin_amdgcn_qsad_pk_u16_u8(p34, sw[m & 3][3], acc1[t]);
                                if constexpr ((m & 3) == 3) {
                                    // an 8x8 block row of position t is complete (SAD <= 8160: the shifted key fits its 16 bits)
                                    constexpr int  by = m >> 2, a = by >> 1;
                                    const uint32_t l0 = (uint32_t)acc0[t], h0 = (uint32_t)(acc0[t] >> 32);
                                    const uint32_t l1 = (uint32_t)acc1[t], h1 = (uint32_t)(acc1[t] >> 32);
                                    b8[by][0][0] = pk_min_u16(b8[by][0][0], (l0 << 3) | ypk[t]);
                                    b8[by][0][1] = pk_min_u16(b8[by][0][1], (h0 << 3) | ypk[t]);
                                    b8[by][1][0] = pk_min_u16(b8[by][1][0], (l1 << 3) | ypk[t]);
                                    b8[by][1][1] = pk_min_u16(b8[by][1][1], (h1 << 3) | ypk[t]);
                                    if constexpr ((by & 1) == 0) {
                                        t16[t][0] = pk_add_u16(l0, l1);
                                        t16[t][1] = pk_add_u16(h0, h1);
                                    } else {
                                        // a row of 16x16 partitions is complete (SAD <= 32640)
                                        const uint32_t d0 = pk_add_u16(t16[t][0], pk_add_u16(l0, l1));
                                        const uint32_t d1 = pk_add_u16(t16[t][1], pk_add_u16(h0, h1));
                                        b16[a][0] = min(b16[a][0], (d0 << 16) | yrk[t]);
                                        b16[a][1] = min(b16[a][1], (d0 & 0xFFFF0000u) | yrk[t]);
                                        b16[a][2] = min(b16[a][2], (d1 << 16) | yrk[t]);
                                        b16[a][3] = min(b16[a][3], (d1 & 0xFFFF0000u) | yrk[t]);
                                        if constexpr ((a & 1) == 0) { // 32-row column sums still fit 16 bits (<= 65280)
                                            half[t][0] = d0;
                                            half[t][1] = d1;
                                        } else {
                                            half[t][0] = pk_add_u16(half[t][0], d0);
                                            half[t][1] = pk_add_u16(half[t][1], d1);
                                        }
                                        if constexpr (a == 1) {
                                            topv[t][0] = half[t][0];
                                            topv[t][1] = half[t][1];
                                        }
                                        if constexpr (a == 3) {
                                            // The whole block: 32x32 and 64x64 sums across the quad. The two lanes of a pair
                                            // (the two 16-pixel columns of a 32x32 partition) split the quad's four positions:
                                            // each hands the packed sums of the other's two positions over and adds what it
                                            // receives to its own, so a lane sums and ranks two positions, not four.
                                            const uint32_t tk = odd ? topv[t][1] : topv[t][0], tr = quad_xor1(odd ? topv[t][0] : topv[t][1]);
                                            const uint32_t bk = odd ? half[t][1] : half[t][0], br = quad_xor1(odd ? half[t][0] : half[t][1]);
#pragma unroll
                                            for (int j = 0; j < 2; ++j) {
                                                const uint32_t tp = j ? (tk >> 16) + (tr >> 16) : (tk & 0xFFFFu) + (tr & 0xFFFFu);
                                                const uint32_t bt = j ? (bk >> 16) + (br >> 16) : (bk & 0xFFFFu) + (br & 0xFFFFu);
                                                uint32_t       all = tp + bt;
                                                all += quad_xor2(all);
                                                b32[0][j] = min(b32[0][j], (tp << (6 + ROW_BITS)) | yrk[t]);
                                                b32[1][j] = min(b32[1][j], (bt << (6 + ROW_BITS)) | yrk[t]);
                                                b64k[j]   = min(b64k[j], (all << (6 + ROW_BITS)) | yrk[t]);
                                            }
                                        }
                                    }
                                }
                            }
                        });
                    });
                }
                // ---- merge. Every key gets the position's column within the tile (6 bits) under its row: (SAD, row, column)
                // then orders like the reference's raster scan. 8x8: SAD << 9 | row << 6 | column; 16x16: SAD << 16 | row << 6
                // | column; larger: SAD << 9 | row << 6 | column. In the lane the minimum over its 4 columns; over the four
                // quads of a row of 16 lanes through DPP (row_quads_min: no LDS, no atomics); the four rows' results meet in
                // LDS, where the lane that owns a partition reads them. Lanes without a unit and columns past the right edge
                // of a window narrower than 8 carry all-ones keys.
                {
                    const uint32_t xbase = (uint32_t)(xg * 4);
                    const bool     narrow = w < 8;
                    uint32_t       xlab[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) xlab[i] = (narrow && (int)(tx0 + xbase + i) >= w) ? 0xFFFFFFFFu : xbase + i;
                    const bool     writer = (tid & 12) == 12; // lanes 12..15 of a row hold the row's minima
                    // (the staged window has served: the four rows' keys, [4][85] dwords, take its place)
                    wave_sync();
                    uint32_t      *part   = win + (tid >> 4) * SVTME_PU_COUNT;
#pragma unroll
                    for (int by = 0; by < 8; ++by)
#pragma unroll
                        for (int c = 0; c < 2; ++c) {
                            const uint32_t d0 = b8[by][c][0], d1 = b8[by][c][1];
                            // a 16-bit key SAD << 3 | row of 8, first with the column under it: the lane's minimum over
                            // its four columns; then the group goes between SAD and row: SAD << (6 + ROW_BITS) | row of the
                            // item << 6 | column (all ones stay above every key a position can produce)
                            const uint32_t k0 = ((d0 & 0xFFFFu) << 6) | xlab[0], k1 = ((d0 >> 16) << 6) | xlab[1];
                            const uint32_t k2 = ((d1 & 0xFFFFu) << 6) | xlab[2], k3 = ((d1 >> 16) << 6) | xlab[3];
                            const uint32_t km = min(min(k0, k1), min(k2, k3));
                            const uint32_t k  = row_quads_min(((km & 0xFFFFFE00u) << (ROW_BITS - 3)) | ((km & 0x1FFu) | ((uint32_t)grp << 9)));
                            const int      bx8 = col * 2 + c; // 8x8 block column
                            const int      pu  = 21 + z16(by >> 1, bx8 >> 1) * 4 + ((by & 1) << 1) + (bx8 & 1);
                            if (writer) part[pu] = k;
                        }
#pragma unroll
                    for (int a = 0; a < 4; ++a) {
                        const uint32_t k = row_quads_min(min(min(b16[a][0] | xlab[0], b16[a][1] | xlab[1]), min(b16[a][2] | xlab[2], b16[a][3] | xlab[3])));
                        if (writer) part[5 + z16(a, col)] = k;
                    }
                    // (32x32 / 64x64: the lane holds positions 0, 1 or 2, 3 of the quad; its pair partner the other two)
                    const uint32_t xl0 = odd ? xlab[2] : xlab[0], xl1 = odd ? xlab[3] : xlab[1];
#pragma unroll
                    for (int a = 0; a < 2; ++a) {
                        uint32_t k = min(b32[a][0] | xl0, b32[a][1] | xl1);
                        k          = row_quads_min(min(k, quad_xor1(k)));
                        if (writer && !odd) part[1 + a * 2 + (col >> 1)] = k;
                    }
                    {
                        uint32_t k = min(b64k[0] | xl0, b64k[1] | xl1);
                        k          = row_quads_min(min(k, quad_xor1(k)));
                        if (writer && col == 0) part[0] = k;
                    }
                }
            }
        }
    }
    __syncthreads();
    // ---- this item's minima join the window's: (SAD << 32) | (y << 16) | x, window-relative
    // (keys only ever decrease, so a plain - possibly stale - read that is already no larger settles it without an atomic)
    for (int pu = tid; pu < SVTME_PU_COUNT; pu += ME_THREADS) {
        const uint32_t k = min(min(win[pu], win[SVTME_PU_COUNT + pu]), min(win[2 * SVTME_PU_COUNT + pu], win[3 * SVTME_PU_COUNT + pu]));
        const int      sh = pu >= 5 && pu < 21 ? 16 : 6 + ROW_BITS;
        const uint32_t sad = k >> sh;
        // a key that no position produced (cannot happen for an item the plan made): all-ones fields exceed the largest SAD
        const bool none = sad > (pu >= 21 ? 8u * 4u : pu >= 5 ? 16u * 8u : pu >= 1 ? 32u * 16u : 64u * 32u) * 255u;
        const int  y = ty0 + (int)((k >> 6) & ((1 << ROW_BITS) - 1)), x = tx0 + (int)(k & 63);
        if constexpr (FUSED) {
            // the window's only item: these are the window's minima (me_finish_kernel's conversion: SADs were taken on every
            // other row, x2 :248-305; the vector in quarter pels, y in the high half :474-476; a key nobody set reads as
            // all ones there)
            const int py = (none ? 0xFFFF : y) + org_y, px = (none ? 0xFFFF : x) + org_x;
            out_sad[pu] = none ? 0xFFFFFFFEu : sad << 1;
            out_mv[pu]  = ((uint32_t)(uint16_t)py << 18) | (uint16_t)((uint16_t)px << 2);
        } else {
            if (none) continue;
            const u64 key = ((u64)sad << 32) | ((uint32_t)y << 16) | (uint32_t)x;
            u64      *g   = &p.me_keys[(size_t)(b64 * 8 + my) * SVTME_PU_COUNT + pu];
            if (key < *g) atomicMin(g, key);
        }
    }
}

// ---- results: p_sb_best_sad / p_sb_best_mv, one plane per reference of the picture. SADs were taken on every other
// row: x2 (:248-305). References that were not searched (pruned) read as zero.
// KEYS == false: the search items have written the results themselves (me_search_kernel<.., true>); only the tidying up.
template <class PICS, bool KEYS>
__global__ void __launch_bounds__(256) me_finish_kernel(PICS pp) {
    const auto &p = pp[blockIdx.y];
    const int n_b64 = p.b64_w * p.b64_h, per_plane = n_b64 * SVTME_PU_COUNT;
    const int n = (p.n_ref[0] + p.n_ref[1]) * per_plane;
    const int i = blockIdx.x * 256 + threadIdx.x;
    // the last kernel that needs the HME records and the counters leaves both zeroed for the next picture that uses
    // this buffer (they are zero when the engine is created), which saves every picture a memset
    if (blockIdx.x == 0)
        for (int c = threadIdx.x; c < SVTME_OPS_SLOTS * 8; c += 256) {
            p.ops_out[c] = p.ops[c];
            p.ops[c]     = 0;
        }
    if (p.tail.enable) {
        // the device tail fills its candidate lists, MV arrays and counts sparsely: they start from zero
        uint32_t *tz = reinterpret_cast<uint32_t *>(p.t_zero_begin);
        const int nt = (int)(p.t_zero_bytes >> 2);
        for (int z = i; z < nt; z += (int)gridDim.x * 256) tz[z] = 0;
    }
    {
        uint32_t *hz = reinterpret_cast<uint32_t *>(p.hme_out);
        const int nz = n_b64 * SVTME_MAX_LISTS * SVTME_MAX_REFS * (int)(sizeof(HmeRefOut) / 4);
        for (int z = i; z < nz; z += (int)gridDim.x * 256) hz[z] = 0;
    }
    if (!KEYS || i >= n) return;
    // output plane q = reference (l, r), list-0 references first
    const int   q = i / per_plane, rest = i - q * per_plane, b64 = rest / SVTME_PU_COUNT, pu = rest - b64 * SVTME_PU_COUNT;
    const int   l = q >= p.n_ref[0] ? 1 : 0, r = l ? q - p.n_ref[0] : q;
    const int   j = b64 * 8 + l * 4 + r;
    const MeJob job = p.me_jobs[j];
    const u64   key = p.me_keys[(size_t)j * SVTME_PU_COUNT + pu];
    uint32_t    sad = 0, mv = 0;
    if (job.w > 0) {
        const int py = (int)((key >> 16) & 0xFFFF) + job.org_y, px = (int)(key & 0xFFFF) + job.org_x;
        sad = (uint32_t)(key >> 32) << 1;
        mv  = ((uint32_t)(uint16_t)py << 18) | (uint16_t)((uint16_t)px << 2); // :474-476
    }
    p.best_sad[i] = sad;
    p.best_mv[i]  = mv;
}

// The parameter records of a batch go to device memory through one small kernel that reads them from the engine's
// page-locked host copy (which the device can address): a copy engine in the middle of a stream of kernels costs the
// stream tens of microseconds of hand-over each way, a kernel in front of kernels costs a launch - one for the whole
// batch. The host leaves the records alone until the batch's pictures have been handed back.
__global__ void __launch_bounds__(256) params_write_kernel(const uint32_t *__restrict__ src, uint32_t *__restrict__ dst, int n_words) {
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n_words; i += gridDim.x * 256) dst[i] = src[i];
}

void launch_params_write(const PicParams *host, PicParams *dev, int n, hipStream_t s) {
    const int n_words = n * (int)(sizeof(PicParams) / 4);
    // (a word per thread: every read crosses the host link, so they all go out at once - one round trip, not four)
    hipLaunchKernelGGL(params_write_kernel, dim3((n_words + 255) / 256), dim3(256), 0, s, reinterpret_cast<const uint32_t *>(host),
                       reinterpret_cast<uint32_t *>(dev), n_words);
}

template <class PICS>
static void launch_me_as(const Batch &b, PICS pics, hipStream_t s, hipEvent_t marks[3]) {
    const PicParams &p = b.host[0]; // size, controls and me_type are the batch's
    const int n_refs = b.max_refs(), n_b64 = b.n_b64();
    // The screen-content search-range rule (enable_me_sr_adjustment == 2 without the zero-MV early exit) makes every
    // other reference's area depend on the first reference's 64x64 result (EbMotionEstimation.c:1390-1399): that
    // reference is planned and searched first, the others after it. (When list 0 / reference 0 is not searched at all
    // the reference code reads whatever the previous block left; protect_closest_refs rules that out for the presets
    // that use the rule.)
    const bool two_phase = p.ctrl.enable_me_sr_adjustment == 2 && p.ctrl.me_early_exit_th == 0 && n_refs > 1;
    int        me_ref0 = 0;
    const dim3     grid(n_b64, p.me_max_items, n_refs * b.n);
    const uint32_t refs_magic = (uint32_t)((0x100000000ull + (uint32_t)n_refs - 1) / (uint32_t)n_refs);
    // no window of these controls is more than one work item: the item plans, searches and finishes its window
    static const bool no_fusion = getenv("SVTME_ME_UNFUSED") != nullptr; // diagnostic: the three kernels in every case
    if (p.me_max_items == 1 && !two_phase && !no_fusion) {
        if (marks[0]) (void)hipEventRecord(marks[0], s);
        hipLaunchKernelGGL((me_search_kernel<PICS, true>), grid, dim3(ME_THREADS), 0, s, pics, n_refs, refs_magic, 0, 1 << 30);
        if (marks[1]) (void)hipEventRecord(marks[1], s);
        // (the tidying up needs few workgroups: a grid-stride loop over the HME records)
        hipLaunchKernelGGL((me_finish_kernel<PICS, false>), dim3(std::min((n_b64 * n_refs * SVTME_PU_COUNT + 255) / 256, 64), b.n), dim3(256), 0, s, pics);
        if (marks[2]) (void)hipEventRecord(marks[2], s);
        return;
    }
    auto       search = [&](int ref_first, int ref_last) {
        hipLaunchKernelGGL((me_search_kernel<PICS, false>), grid, dim3(ME_THREADS), 0, s, pics, n_refs, refs_magic, ref_first, ref_last);
    };
    if (two_phase) {
        hipLaunchKernelGGL(me_plan_kernel<PICS>, dim3(n_b64, 1, b.n), dim3(ME_THREADS), 0, s, pics, 0);
        search(0, 0);
        me_ref0 = 1;
    }
    const int n_now = n_refs - me_ref0;
    hipLaunchKernelGGL(me_plan_kernel<PICS>, dim3(n_b64, n_now, b.n), dim3(ME_THREADS), 0, s, pics, me_ref0);
    if (marks[0]) (void)hipEventRecord(marks[0], s);
    search(me_ref0, 1 << 30);
    if (marks[1]) (void)hipEventRecord(marks[1], s);
    hipLaunchKernelGGL((me_finish_kernel<PICS, true>), dim3((n_b64 * n_refs * SVTME_PU_COUNT + 255) / 256, b.n), dim3(256), 0, s, pics);
    if (marks[2]) (void)hipEventRecord(marks[2], s);
}

void launch_me(const Batch &b, hipStream_t s, hipEvent_t marks[3]) {
    if (b.n == 1) launch_me_as(b, OnePicture{b.host[0]}, s, marks);
    else launch_me_as(b, ManyPictures{b.dev}, s, marks);
}

// resident workgroups per CU the runtime expects for the search kernel (diagnostic)
int occupancy_me_search() {
    int n = 0;
    (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, me_search_kernel<OnePicture, false>, ME_THREADS, 0);
    return n;
}

} // namespace svtme
