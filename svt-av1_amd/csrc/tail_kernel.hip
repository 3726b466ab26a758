// tail_kernel.hip - optional second stage: what motion_estimation_b64 does after integer_search_b64, per 64x64 block,
// on the results the search kernels left in device memory.
//
// Replaces, from Source/Lib/Encoder/Codec/EbMotionEstimation.c:
//   me_prune_ref :1576-1618                              construct_me_candidate_array_mrp_off :2430-2543
//   construct_me_candidate_array :2544-2685              compute_distortion :2814-2865
//   perform_gm_detection :2688-2811
// One single-wave workgroup per b64; a lane owns a partition (85 partitions: two rounds), sums and counts go through
// wave shuffles and LDS. Integer bookkeeping, a few microseconds per picture: latency-bound, not worth a roofline.
#include "search_common.h"

namespace svtme {

constexpr int TAIL_THREADS = 64;

// z_to_raster, EbMotionEstimation.c:2412-2429
__constant__ uint8_t TAIL_Z_TO_RASTER[85] = {
    0,  1,  2,  3,  4,  5,  6,  9,  10, 7,  8,  11, 12, 13, 14, 17, 18, 15, 16, 19, 20, 21, 22, 29, 30, 23, 24, 31, 32,
    37, 38, 45, 46, 39, 40, 47, 48, 25, 26, 33, 34, 27, 28, 35, 36, 41, 42, 49, 50, 43, 44, 51, 52, 53, 54, 61, 62, 55,
    56, 63, 64, 69, 70, 77, 78, 71, 72, 79, 80, 57, 58, 65, 66, 59, 60, 67, 68, 73, 74, 81, 82, 75, 76, 83, 84};

struct TailShared {
    u64      hme_sad[8];
    uint32_t me_dist[SVTME_PU_COUNT];
    uint32_t cnt[32]; // perform_gm_detection's per (list, ref, axis, sign) counters
    uint32_t stationary;
    uint8_t  do_ref[8];
    uint8_t  first_cand[SVTME_PU_COUNT]; // candidate 0 of every partition
    uint32_t sad[8][SVTME_PU_COUNT], mv[8][SVTME_PU_COUNT]; // this block's minima per reference slot l * 4 + r
};

// MeCandidate bit-field, EbMotionEstimationLcuResults.h:29-35: direction:2, ref_idx_l0:2, ref_idx_l1:2, ref0_list:1,
// ref1_list:1 (the literal 24 the reference assigns to the 1-bit list fields truncates to 0)
__device__ __forceinline__ uint8_t me_cand(int direction, int idx0, int idx1, int list0, int list1) {
    return (uint8_t)((direction & 3) | ((idx0 & 3) << 2) | ((idx1 & 3) << 4) | ((list0 & 1) << 6) | ((list1 & 1) << 7));
}

__device__ __forceinline__ u64 wave_sum_u64(u64 v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += (u64)__shfl_xor((long long)v, m);
    return v;
}

template <class PICS>
__global__ void __launch_bounds__(TAIL_THREADS) me_tail_kernel(PICS pp, int n_b64) {
    __shared__ TailShared S;
    const auto &p    = pp[blockIdx.x / n_b64];
    const auto &k    = p.ctrl;
    const auto &t    = p.tail;
    const int             lane = threadIdx.x;
    const int             b64  = blockIdx.x % n_b64;
    if (b64_left_out(p, b64)) return;
    const int             b64x = (b64 % p.b64_w) * 64, b64y = (b64 / p.b64_w) * 64;
    const uint32_t        block_w = (uint32_t)min(p.width - b64x, 64), block_h = (uint32_t)min(p.height - b64y, 64);
    // best SAD / MV of reference (l, r), partition n: one plane per reference of the picture, list 0 first
    const size_t plane = (size_t)p.b64_w * p.b64_h * SVTME_PU_COUNT;
    auto         at = [&](const uint32_t *a, int l, int r, int n) -> uint32_t {
        return a[(size_t)(l ? p.n_ref[0] + r : r) * plane + (size_t)b64 * SVTME_PU_COUNT + n];
    };
    // after the hand-over to LDS below the rest of the kernel reads the minima there
    const int bs = 0, bm = 1;
    auto      lds_at = [&](int which, int l, int r, int n) -> uint32_t { return which ? S.mv[l * 4 + r][n] : S.sad[l * 4 + r][n]; };
    const svtme_search_result *sr = p.search_results + (size_t)b64 * 8;
    uint8_t  *o_total = p.t_total_cand + (size_t)b64 * SVTME_PU_COUNT;
    uint32_t *o_mv    = p.t_mv_array + (size_t)b64 * SVTME_PU_COUNT * SVTME_TAIL_MV_STRIDE;
    uint8_t  *o_cand  = p.t_cand_array + (size_t)b64 * SVTME_PU_COUNT * SVTME_TAIL_CAND_STRIDE;
    const int n_ref0 = p.n_ref[0], n_ref1 = p.n_ref[1];

    if (lane < 8) {
        S.do_ref[lane]  = sr[lane].do_ref;
        S.hme_sad[lane] = sr[lane].hme_sad;
    }
    if (lane < 32) S.cnt[lane] = 0;
    if (lane == 0) S.stationary = 0;
    for (int i = lane; i < SVTME_PU_COUNT; i += TAIL_THREADS) {
        S.me_dist[i]    = 0;
        S.first_cand[i] = 0;
    }
    // (the per-b64 result buffers start from zero: me_finish_kernel cleared them)
    // every reference's minima of this block -> LDS, all loads in flight together: slot j = l * 4 + r, partitions lane
    // and lane + 64
    {
        uint32_t r_sad[2][8], r_mv[2][8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int  l = j >> 2, r = j & 3;
            const bool have = l < p.n_list && r < p.n_ref[l];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int  n  = lane + 64 * h;
                const bool on = have && n < SVTME_PU_COUNT;
                r_sad[h][j] = on ? at(p.best_sad, l, r, n) : 0u;
                r_mv[h][j]  = on ? at(p.best_mv, l, r, n) : 0u;
            }
        }
#pragma unroll
        for (int j = 0; j < 8; ++j)
#pragma unroll
            for (int h = 0; h < 2; ++h)
                if (lane + 64 * h < SVTME_PU_COUNT) {
                    S.sad[j][lane + 64 * h] = r_sad[h][j];
                    S.mv[j][lane + 64 * h]  = r_mv[h][j];
                }
    }
    __syncthreads();

    // ---- me_prune_ref :1576-1618: HME SAD := sum of the 64 best 8x8 SADs; prune on its deviation from the best
    if (k.enable_hme_flag && k.enable_me_hme_ref_pruning) {
        for (int l = 0; l < p.n_list; ++l)
            for (int r = 0; r < p.n_ref[l]; ++r) {
                const int i = l * 4 + r;
                u64       v = (u64)SVTME_MAX_SAD_VALUE * 64;
                if (S.do_ref[i]) v = wave_sum(S.sad[i][21 + lane]);
                if (lane == 0) S.hme_sad[i] = v;
            }
        __syncthreads();
        if (lane == 0) {
            u64 best = ~0ull;
            for (int i = 0; i < 8; ++i) best = S.hme_sad[i] < best ? S.hme_sad[i] : best;
            const uint32_t th = k.prune_ref_if_me_sad_dev_bigger_than_th;
            for (int i = 0; i < 8; ++i)
                if ((!k.protect_closest_refs || (i & 3) > 0) && th != 0xFFFF && (S.hme_sad[i] - best) * 100 > (u64)th * best)
                    S.do_ref[i] = 0;
        }
        __syncthreads();
    }
    if (lane < 8) {
        p.t_do_ref[(size_t)b64 * 8 + lane]  = S.do_ref[lane];
        p.t_hme_sad[(size_t)b64 * 8 + lane] = S.hme_sad[lane];
    }

    // ---- candidates. Partitions the encoder does not use (8x8 / 16x16 disabled) still set me_distortion.
    const bool mrp_off = n_ref0 == 1 && n_ref1 == 1;
    const int  n_pus   = t.enable_me_16x16 ? (t.enable_me_8x8 ? 85 : 21) : 5;
    if (mrp_off) { // memset(total_me_candidate_index, 1, n_pus) :2445
        for (int i = lane; i < n_pus; i += TAIL_THREADS) o_total[i] = 1;
        __syncthreads();
    }
    for (int n = lane; n < SVTME_PU_COUNT; n += TAIL_THREADS) {
        const int  pu  = TAIL_Z_TO_RASTER[n];
        const bool use = t.enable_me_16x16 ? (t.enable_me_8x8 || n < 21) : n < 5;
        uint8_t   *ca  = o_cand + pu * t.max_cand;
        int        off = 0;
        if (mrp_off) {
            // construct_me_candidate_array_mrp_off :2430-2543
            uint32_t n_list = (uint32_t)p.n_list;
            uint8_t  org[2] = {S.do_ref[0], (uint8_t)(n_list == 1 ? 0 : S.do_ref[4])};
            if (n_list < 2 || !S.do_ref[4]) n_list = 1;
            const uint32_t prune_th = (org[0] && org[1]) ? (uint32_t)k.prune_me_candidates_th : 0;
            uint8_t        dr[2] = {org[0], org[1]};
            const uint32_t s0 = lds_at(bs, 0, 0, n), s1 = lds_at(bs, 1, 0, n);
            const uint32_t best = prune_th > 0 ? min(s0, s1) : ~0u;
            int            min_list = -1;
            if (k.use_best_unipred_cand_only && dr[0] && dr[1]) min_list = s0 < s1 ? 0 : 1;
            for (uint32_t l = 0; l < n_list && (use || off == 0); ++l) {
                if (!dr[l]) continue;
                const uint32_t sad = l ? s1 : s0, mv = lds_at(bm, (int)l, 0, n);
                if (prune_th > 0) {
                    const uint32_t d = (sad - best) * 100;
                    if (d > best * prune_th) {
                        dr[l] = 0;
                        continue;
                    }
                }
                if (min_list != -1 && min_list != (int)l) {
                    if (use) o_mv[pu * t.max_refs + (l ? t.max_l0 : 0)] = mv;
                    continue;
                }
                if (off == 0) S.me_dist[pu] = sad;
                if (use) {
                    ca[off] = me_cand((int)l, 0, 0, 0, l == 1 ? 1 : 0);
                    if (off == 0) S.first_cand[pu] = ca[0];
                    o_mv[pu * t.max_refs + (l ? t.max_l0 : 0)] = mv;
                }
                off++;
            }
            if (dr[0] && dr[1] && use) {
                ca[off] = me_cand(2, 0, 0, 0, 1);
                if (off == 0) S.first_cand[pu] = ca[0];
                o_total[pu] = (uint8_t)(off + 1);
            }
        } else {
            // construct_me_candidate_array :2544-2685
            const uint32_t prune_th = (uint32_t)k.prune_me_candidates_th;
            uint8_t        dr[2][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
            uint32_t       best = ~0u;
            if (prune_th > 0)
                for (int l = 0; l < p.n_list; ++l)
                    for (int r = 0; r < p.n_ref[l]; ++r)
                        if (S.do_ref[l * 4 + r]) best = min(best, lds_at(bs, l, r, n));
            for (int l = 0; l < p.n_list && (use || off == 0); ++l)
                for (int r = 0; r < p.n_ref[l] && (use || off == 0); ++r) {
                    dr[l][r] = S.do_ref[l * 4 + r];
                    if (!dr[l][r]) continue;
                    const uint32_t sad = lds_at(bs, l, r, n);
                    if (prune_th > 0) {
                        const uint32_t d = (sad - best) * 100;
                        if (d > best * prune_th) {
                            dr[l][r] = 0;
                            continue;
                        }
                    }
                    if (off == 0) S.me_dist[pu] = sad;
                    if (use) {
                        ca[off] = me_cand(l, r, r, 0, l == 1 ? 1 : 0);
                        if (off == 0) S.first_cand[pu] = ca[0];
                        o_mv[pu * t.max_refs + (l ? t.max_l0 : 0) + r] = lds_at(bm, l, r, n);
                    }
                    off++;
                }
            if (p.n_list == 2 && use) {
                for (int a = 0; a < n_ref0; ++a)
                    for (int b = 0; b < n_ref1; ++b)
                        if (dr[0][a] && dr[1][b]) {
                            ca[off] = me_cand(2, a, b, 0, 1);
                            if (off == 0) S.first_cand[pu] = ca[0];
                            off++;
                        }
                for (int a = 1; a < n_ref0; ++a)
                    if (dr[0][0] && dr[0][a]) {
                        ca[off] = me_cand(2, 0, a, 0, 0);
                        if (off == 0) S.first_cand[pu] = ca[0];
                        off++;
                    }
                if (n_ref1 == 3 && dr[1][0] && dr[1][2]) {
                    ca[off] = me_cand(2, 0, 2, 1, 1);
                    if (off == 0) S.first_cand[pu] = ca[0];
                    off++;
                }
            }
            if (use) o_total[pu] = (uint8_t)off;
        }
    }
    __syncthreads();
    for (int i = lane; i < SVTME_PU_COUNT; i += TAIL_THREADS) p.t_distortion[(size_t)b64 * SVTME_PU_COUNT + i] = S.me_dist[i];

    // ---- compute_distortion :2814-2865
    {
        const uint32_t d64 = S.me_dist[0];
        const uint32_t d32 = S.me_dist[1] + S.me_dist[2] + S.me_dist[3] + S.me_dist[4];
        const uint32_t d16 = wave_sum(lane < 16 ? S.me_dist[5 + lane] : 0u);
        const uint32_t v8  = S.me_dist[21 + lane];
        const uint32_t d8  = wave_sum(v8);
        const long long diff = (long long)v8 - (long long)(d8 / 64);
        const u64       ssq  = wave_sum_u64((u64)(diff * diff));
        if (lane == 0) {
            const uint32_t pix = block_w * block_h, sz = 64 * 64, f = k.stat_factor;
            uint32_t      *st  = p.t_stats + (size_t)b64 * 8;
            st[0] = t.input_resolution <= 2 ? d8 : d16; // rc_me_distortion
            st[1] = (((d64 * sz) / pix) * f) / 100;
            st[2] = (((d32 * sz) / pix) * f) / 100;
            st[3] = (((d16 * sz) / pix) * f) / 100;
            st[4] = (((d8 * sz) / pix) * f) / 100;
            st[5] = (uint32_t)(ssq / 64);
            if (p.me_type == SVTME_ME_FIRST_PASS) // the first pass does not call compute_distortion (:2990)
                for (int i = 0; i < 6; ++i) st[i] = 0;
        }
    }

    // ---- perform_gm_detection :2688-2811. The partition index is a raster index; the reference uses it as is to index
    // the z-ordered p_sb_best_mv array.
    uint32_t stationary_flag = 0, allow_gm = 0;
    if (t.gm_enabled) {
        const bool low_res = t.input_resolution <= 2;
        const int  n_blk   = low_res ? 64 : 16;
        if (lane < n_blk) {
            int idx = (low_res ? 21 : 5) + lane;
            if (low_res && !t.enable_me_8x8) {
                if (idx >= 21) idx = 5 + (((idx - 21) >> 4) << 2) + (((idx - 21) & 7) >> 1); // me_idx_85_8x8_to_16x16_conversion
                if (!t.enable_me_16x16 && idx >= 5) idx = 1 + (((idx - 5) >> 3) << 1) + (((idx - 5) & 3) >> 1);
            } else if (!low_res && !t.enable_me_16x16 && idx >= 5) {
                idx = 1 + (((idx - 5) >> 3) << 1) + (((idx - 5) & 3) >> 1); // me_idx_16x16_to_parent_32x32_conversion
            }
            const uint8_t cand = S.first_cand[idx];
            const int     dir  = cand & 3;
            const int     l = (dir == 0 || dir == 2) ? ((cand >> 6) & 1) : ((cand >> 7) & 1);
            const int     r = (dir == 0 || dir == 2) ? ((cand >> 2) & 3) : ((cand >> 4) & 3);
            const int     dist = (int)(uint16_t)p.dist[l][r];
            const int     th = low_res ? (t.gm_use_distance_based_active_th ? max(dist >> 1, 4) : 4)
                                       : (t.gm_use_distance_based_active_th ? max(dist * 16, 32) : 32);
            const uint32_t mv = lds_at(bm, l, r, idx);
            const int      mx = (int16_t)(mv & 0xFFFF), my = (int16_t)(mv >> 16);
            uint32_t      *c  = &S.cnt[(l * 4 + r) * 4];
            if (mx < -th) atomicAdd(&c[0], 1u);
            else if (mx > th) atomicAdd(&c[1], 1u);
            if (my < -th) atomicAdd(&c[2], 1u);
            else if (my > th) atomicAdd(&c[3], 1u);
            const int sth = low_res ? 0 : 4;
            if (abs(mx) <= sth && abs(my) <= sth) atomicAdd(&S.stationary, 1u);
        }
        __syncthreads();
        const uint32_t tot = (uint32_t)n_blk;
        if (S.stationary > (tot * 5) / 100) stationary_flag = 1;
        for (int i = 0; i < 32; ++i)
            if (S.cnt[i] > tot / 2) allow_gm = 1;
    }
    if (lane == 0) {
        p.t_stats[(size_t)b64 * 8 + 6] = stationary_flag;
        p.t_stats[(size_t)b64 * 8 + 7] = allow_gm;
    }
}

void launch_tail(const Batch &b, hipStream_t s) {
    if (b.n == 1) hipLaunchKernelGGL(me_tail_kernel<OnePicture>, dim3(b.n_b64()), dim3(TAIL_THREADS), 0, s, OnePicture{b.host[0]}, b.n_b64());
    else hipLaunchKernelGGL(me_tail_kernel<ManyPictures>, dim3(b.n_b64() * b.n), dim3(TAIL_THREADS), 0, s, ManyPictures{b.dev}, b.n_b64());
}

} // namespace svtme
