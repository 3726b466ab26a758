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
// Mapping: one 256-thread workgroup per (b64, reference). The search window ((w+63) x (h+63) bytes) and the
// 64x64 source block are staged in LDS. A quad of lanes owns 4 horizontally adjacent search positions; each lane
// of the quad owns one 16-pixel column of the block and walks a run of up to 8 search rows. Per row it issues
// 128 v_qsad_pk_u16_u8 (4 positions x 4 pixels each) into sixteen packed 4x16-bit accumulators = its 16 8x8 SADs
// (rows 0,2,4,6 of each 8x8, as the reference's sub-sampled search does; doubled at the end). 16x16 SADs are summed
// in-lane, 32x32 and 64x64 across the quad with DPP-style lane exchanges. Running minima carry the row number in the
// low key bits, so the reference's tie rule (first strictly smaller SAD in raster order) is reproduced exactly;
// at the end of a run every lane merges its minima into 85 workgroup-wide 64-bit keys (SAD, y, x) with LDS atomics.
#include <type_traits>

#include "search_common.h"

namespace svtme {

constexpr int ME_THREADS   = 256;
constexpr int ME_WIN_BYTES = 40 * 1024;
constexpr int ME_RUN       = 8; // search rows per lane run; the row-in-run number needs 3 key bits

struct MeShared {
    __align__(16) uint32_t src[32 * 16]; // rows 0,2,..,62 of the source block, 16 dwords each
    u64                    best[SVTME_PU_COUNT];
    u64                    hsad[8]; // per (list, ref): search centre, HME SAD, do_ref, search-range divisor
    uint32_t               div[8];
    int16_t                sc_x[8], sc_y[8];
    uint8_t                do_ref[8];
    uint32_t               red;
    int                    zero; // always 0, but only known at run time
};

// z-order index of the 16x16 block in raster (row, col): svt_ext_all_sad_calculation_8x8_16x16_c's `offsets` (:366)
__device__ __forceinline__ int z16(int row, int col) { return ((row >> 1) << 3) | ((col >> 1) << 2) | ((row & 1) << 1) | (col & 1); }

// compile-time loop: f(std::integral_constant<int, I>) for I in [B, E); indices stay constant expressions, so the
// per-position register arrays below never become runtime-indexed
template <int B, int E, class F>
__device__ __forceinline__ void static_for(F &&f) {
    if constexpr (B < E) {
        f(std::integral_constant<int, B>{});
        static_for<B + 1, E>(f);
    }
}

// lane ^ 1 and lane ^ 2 within a quad, as DPP quad_perm moves (no LDS traffic)
__device__ __forceinline__ uint32_t quad_xor1(uint32_t v) { return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0xB1, 0xF, 0xF, true); }
__device__ __forceinline__ uint32_t quad_xor2(uint32_t v) { return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x4E, 0xF, 0xF, true); }

typedef unsigned short us2 __attribute__((ext_vector_type(2)));
// v_pk_min_u16 / v_pk_add_u16 / v_pk_lshlrev_b16 on two 16-bit fields
__device__ __forceinline__ uint32_t pk_min_u16(uint32_t a, uint32_t b) {
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_min(__builtin_bit_cast(us2, a), __builtin_bit_cast(us2, b)));
}

__global__ void __launch_bounds__(ME_THREADS, 2) me_kernel(PicParams p) {
    __shared__ MeShared                   S;
    extern __shared__ __align__(16) uint8_t win[];
    const svtme_controls &k   = p.ctrl;
    const int             tid = threadIdx.x;
    const int             b64 = blockIdx.x;
    const int             b64x = (b64 % p.b64_w) * 64, b64y = (b64 / p.b64_w) * 64;
    const int             block_w = min(p.width - b64x, 64), block_h = min(p.height - b64y, 64);

    int my_l = 0, my_r = blockIdx.y;
    if (my_r >= p.n_ref[0]) {
        my_r -= p.n_ref[0];
        my_l = 1;
    }

    // ---- set_final_seach_centre_sb (:2117-2316): one thread walks the references in order; the running
    // centre / SAD are not reset between references (base-layer list-1 references inherit the previous SAD).
    const HmeRefOut *hme = &p.hme_out[b64 * SVTME_MAX_LISTS * SVTME_MAX_REFS];
    if (tid == 0) {
        for (int i = 0; i < 8; ++i) { // init_me_hme_data :2913-2923
            S.sc_x[i] = S.sc_y[i] = 0;
            S.hsad[i]             = 0xFFFFFFFFull;
            S.do_ref[i]           = 1;
            S.div[i]              = 1;
        }
        int16_t hx = 0, hy = 0, cx = 0, cy = 0;
        u64     hsad = 0;
        int     level = -1;
        if (k.enable_hme_level0_flag && !k.enable_hme_level1_flag && !k.enable_hme_level2_flag) level = 0;
        if (k.enable_hme_level1_flag && !k.enable_hme_level2_flag) level = 1;
        if (k.enable_hme_level2_flag) level = 2;
        for (int l = 0; l < p.n_list; ++l)
            for (int r = 0; r < p.n_ref[l]; ++r) {
                if (ref_searched(p, l)) {
                    if (k.enable_hme_flag) {
                        if (level >= 0) {
                            const HmeRefOut &h = hme[l * SVTME_MAX_REFS + r];
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
                        cx = hx;
                        cy = hy;
                    }
                } else {
                    cx = cy = 0;
                }
                S.sc_x[l * 4 + r] = cx;
                S.sc_y[l * 4 + r] = cy;
                S.hsad[l * 4 + r] = hsad;
            }
        // ---- hme_prune_ref_and_adjust_sr (:2373-2410), over all 2x4 entries
        if (k.enable_hme_flag && (k.enable_me_sr_adjustment || k.enable_me_hme_ref_pruning)) {
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
    if (blockIdx.y == 0 && tid < 8) {
        const int            l = tid >> 2, r = tid & 3;
        svtme_search_result &o = p.search_results[b64 * 8 + tid];
        o.hme_sc_x = S.sc_x[tid];
        o.hme_sc_y = S.sc_y[tid];
        o.do_ref   = S.do_ref[tid];
        o.list_i   = (uint8_t)l;
        o.ref_i    = (uint8_t)r;
        o.reserved = 0;
        o.hme_sad  = S.hsad[tid];
        p.sr_divisor[b64 * 8 + tid] = S.div[tid];
        const bool have = l < p.n_list && r < p.n_ref[l];
        p.zz_sad[b64 * 8 + tid] = have ? hme[tid].zz_sad : ~0u;
    }
    const int my = my_l * 4 + my_r;
    if (!S.do_ref[my]) return;

    // ---- integer_search_b64 for (my_l, my_r)
    const DevPlane &cur = p.cur.lvl[2];
    const DevPlane &ref = p.ref[my_l][my_r].lvl[2];
    int16_t         cx = S.sc_x[my], cy = S.sc_y[my];
    const int       dist = scaled_dist(p.dist[my_l][my_r]);
    int16_t         w = (int16_t)min((int)(int16_t)k.me_sa.sa_min.width * dist, (int)k.me_sa.sa_max.width);
    int16_t         h = (int16_t)min((int)(int16_t)k.me_sa.sa_min.height * dist, (int)k.me_sa.sa_max.height);
    w = (int16_t)((max(1, (int)((uint32_t)w / S.div[my])) + 7) & ~0x07);
    h = (int16_t)max(3, (int)((uint32_t)h / S.div[my]));
    const int16_t h0 = h, w0 = w;
    const uint32_t zz = hme[my_l * SVTME_MAX_REFS + my_r].zz_sad;
    if (k.me_early_exit_th) {
        if (zz < k.me_early_exit_th / 6) w = h = 1;
    } else {
        u64 best_hme_sad = ~0ull;
        int accurate     = 1;
        if ((cx != 0 || cy != 0) && p.is_ref) {
            // check_00_center :1170-1243
            const uint32_t zero_sad = wg_block_sad(cur, ref, b64x, b64y, 0, 0, block_w, block_h >> 1, 2, &S.red) << 1;
            if (b64x + cx < -63) cx = (int16_t)(-63 - b64x);
            if (b64x + cx > ref.width - 1) cx = (int16_t)(cx - ((b64x + cx) - (ref.width - 1)));
            if (b64y + cy < -63) cy = (int16_t)(-63 - b64y);
            if (b64y + cy > ref.height - 1) cy = (int16_t)(cy - ((b64y + cy) - (ref.height - 1)));
            const uint32_t hme_sad = wg_block_sad(cur, ref, b64x, b64y, cx, cy, block_w, block_h >> 1, 2, &S.red) << 1;
            best_hme_sad           = hme_sad;
            if (tid == 0) atomicAdd(&p.ops[0], 2ull * (block_h >> 1) * block_w);
            if (zero_sad <= hme_sad) cx = cy = 0;
            if (cx == 0 && cy == 0) accurate = 0;
        }
        if (k.enable_me_sr_adjustment == 2) {
            if ((accurate && best_hme_sad < 24 * 24) || (p.is_ref && S.hsad[my] < 24 * 24)) h = (int16_t)(h / 2);
            // (the p_sb_best_sad[0][0][0] < 5000 rule of :1390-1399 needs the first reference's result: serial mode,
            // rejected by the host for now)
        }
    }
    (void)h0;
    (void)w0;
    int16_t org_x = (int16_t)(cx - (w >> 1)), org_y = (int16_t)(cy - (h >> 1));
    clip_axis(b64x, org_x, w, -63, p.width);
    w = (int16_t)(w < 8 ? w : (w & ~0x07));
    clip_axis(b64y, org_y, h, -63, p.height);

    // ---- open_loop_me_fullpel_search_sblock
    if (tid == 0) atomicAdd(&p.ops[5], (unsigned long long)w * h * 2048ull);
    if (tid < SVTME_PU_COUNT) S.best[tid] = ~0ull;
    for (int i = tid; i < 32 * 16; i += ME_THREADS) {
        const int r = i >> 4, d = i & 15;
        S.src[i] = *reinterpret_cast<const uint32_t *>(cur.pix + (ptrdiff_t)(b64y + 2 * r) * cur.stride + b64x + 4 * d);
    }
    const int n_xg  = (w + 3) >> 2;
    const int pitch = n_xg * 4 + 64; // bytes per staged window row
    const int pd    = pitch >> 2;
    // search rows per LDS band: a position reads rows y .. y+62, and the staggered walk runs up to 8 rows past that
    // for positions that are then discarded
    int       band_rows = ME_WIN_BYTES / pitch - 70;
    if (band_rows < 1) band_rows = 1;
    const int quad = tid >> 2, col = tid & 3;
    const uint32_t *sbase = &S.src[col * 4];
    if (tid == 0) S.zero = 0;
    __syncthreads();
    const int opaque_zero = S.zero;

    for (int y0 = 0; y0 < h; y0 += band_rows) {
        const int rows = min(band_rows, h - y0);
        __syncthreads();
        {
            uint32_t *dst = reinterpret_cast<uint32_t *>(win);
            const int n   = pd * (rows + 62);
            for (int i = tid; i < n; i += ME_THREADS) {
                const int r = i / pd, d = i - r * pd;
                dst[i] = load_u32_any(ref.pix + (ptrdiff_t)(b64y + org_y + y0 + r) * ref.stride + b64x + org_x + 4 * d);
            }
        }
        __syncthreads();
        const int n_run   = (rows + ME_RUN - 1) / ME_RUN;
        const int run_len = (rows + n_run - 1) / n_run;
        const int n_unit  = n_xg * n_run;
        for (int ub = 0; ub < n_unit; ub += ME_THREADS / 4) {
            const int  u     = ub + quad;
            const bool valid = u < n_unit;
            const int  run   = valid ? u / n_xg : 0;
            const int  xg    = valid ? u - run * n_xg : 0;
            const int  ya    = run * run_len, yb = min(rows, ya + run_len);
            // running minima of this lane: key = SAD * 8 + row-in-run (8x8: packed 16-bit; larger: 32-bit)
            uint32_t b8[8][2][2];
            uint32_t b16[4][4], b32[2][4], b64k[4];
#pragma unroll
            for (int a = 0; a < 8; ++a)
#pragma unroll
                for (int c = 0; c < 2; ++c) b8[a][c][0] = b8[a][c][1] = 0xFFFFFFFFu;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
#pragma unroll
                for (int a = 0; a < 4; ++a) b16[a][i] = 0xFFFFFFFFu;
                b32[0][i] = b32[1][i] = b64k[i] = 0xFFFFFFFFu;
            }
            // Two passes per run: its even rows, then its odd rows. The (up to) 4 rows of a pass are walked together,
            // staggered by one sampled source row: position t meets source row m at step s = t + m, while the lane holds
            // reference row (first row of the pass + 2 s) in registers. Every reference row is read from LDS once for 4
            // positions and every source row once (kept for 4 steps).
            for (int par = 0; par < 2; ++par) {
                const int yfirst = ya + par;
                const int n_pos  = valid ? min(4, max(0, (yb - yfirst + 1) >> 1)) : 0; // same in the 4 lanes of a quad
                if (__ballot(n_pos > 0) == 0) continue;
                const uint32_t *wrow = reinterpret_cast<const uint32_t *>(win) + yfirst * pd + xg + col * 4;
                const uint32_t *wrow_odd = wrow + 1 + opaque_zero;
                u64             acc0[4], acc1[4], t16[4];
                uint32_t        half[4][4], topv[4][4], sw[4][4];
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    acc0[t] = acc1[t] = t16[t] = 0;
#pragma unroll
                    for (int i = 0; i < 4; ++i) half[t][i] = topv[t][i] = 0;
                }
                static_for<0, 35>([&](auto s_c) {
                    constexpr int   s  = decltype(s_c)::value;
                    // the four 8-byte operands overlap by a dword. The odd pairs are read through a second pointer the
                    // compiler cannot prove equal to the first (+ a zero it only sees at run time), so every pair lands in its
                    // own aligned register pair straight from LDS instead of being assembled with register copies.
                    const uint32_t *rr = wrow + (2 * s) * pd, *rq = wrow_odd + (2 * s) * pd;
                    const u64 p01 = ((u64)rr[1] << 32) | rr[0], p12 = ((u64)rq[1] << 32) | rq[0];
                    const u64 p23 = ((u64)rr[3] << 32) | rr[2], p34 = ((u64)rq[3] << 32) | rq[2];
                    if constexpr (s < 32) {
                        const uint4 v = *reinterpret_cast<const uint4 *>(sbase + s * 16);
                        sw[s & 3][0] = v.x, sw[s & 3][1] = v.y, sw[s & 3][2] = v.z, sw[s & 3][3] = v.w;
                    }
                    static_for<0, 4>([&](auto t_c) {
                        constexpr int t = decltype(t_c)::value;
                        constexpr int m = s - t; // sampled source row (source row 2m) of position t at this step
                        if constexpr (m >= 0 && m <= 31) {
                            acc0[t] = __builtin_amdgcn_qsad_pk_u16_u8(p01, sw[m & 3][0], acc0[t]);
                            acc0[t] = __builtin_amdgcn_qsad_pk_u16_u8(p12, sw[m & 3][1], acc0[t]);
                            acc1[t] = __builtin_amdgcn_qsad_pk_u16_u8(p23, sw[m & 3][2], acc1[t]);
                            acc1[t] = __builtin_amdgcn_qsad_pk_u16_u8(p34, sw[m & 3][3], acc1[t]);
                            if constexpr ((m & 3) == 3) {
                                // an 8x8 block row of position t is complete
                                constexpr int  by  = m >> 2, a = by >> 1;
                                const bool     on  = t < n_pos;
                                const uint32_t yr  = (uint32_t)(par + 2 * t);
                                const uint32_t ypk = yr | (yr << 16);
                                if (on) {
                                    // 8x8: key16 = sad * 8 + yr (sad <= 8160: fits 16 bits, the add never carries), two positions per dword
#pragma unroll
                                    for (int c = 0; c < 2; ++c)
#pragma unroll
                                        for (int hw = 0; hw < 2; ++hw) {
                                            const uint32_t v  = (uint32_t)((c ? acc1[t] : acc0[t]) >> (32 * hw));
                                            b8[by][c][hw]     = pk_min_u16(b8[by][c][hw], (v << 3) + ypk);
                                        }
                                }
                                t16[t] += acc0[t] + acc1[t]; // 16-bit fields stay <= 32640
                                acc0[t] = acc1[t] = 0;
                                if constexpr ((by & 1) == 1) {
                                    // a 16-row band (one row of 16x16 partitions) is complete: 32-bit sums per position from here on
#pragma unroll
                                    for (int i = 0; i < 4; ++i) {
                                        const uint32_t sv = (uint32_t)(t16[t] >> (16 * i)) & 0xFFFFu;
                                        if (on) b16[a][i] = min(b16[a][i], (sv << 3) | yr);
                                        half[t][i] += sv;
                                    }
                                    t16[t] = 0;
                                    if constexpr (a == 1) {
#pragma unroll
                                        for (int i = 0; i < 4; ++i) {
                                            topv[t][i] = half[t][i];
                                            half[t][i] = 0;
                                        }
                                    }
                                    if constexpr (a == 3) { // the whole block: 32x32 and 64x64 sums across the quad
#pragma unroll
                                        for (int i = 0; i < 4; ++i) {
                                            uint32_t tp = topv[t][i], bt = half[t][i];
                                            tp += quad_xor1(tp);
                                            bt += quad_xor1(bt);
                                            uint32_t all = tp + bt;
                                            all += quad_xor2(all);
                                            if (on) {
                                                b32[0][i] = min(b32[0][i], (tp << 3) | yr);
                                                b32[1][i] = min(b32[1][i], (bt << 3) | yr);
                                                b64k[i]   = min(b64k[i], (all << 3) | yr);
                                            }
                                            half[t][i] = 0;
                                        }
                                    }
                                }
                            }
                        }
                    });
                });
            }
            // merge this lane's minima into the workgroup keys: (SAD << 32) | (y << 16) | x, window-relative
            if (valid && ya < yb) {
                const int ybase = y0 + ya;
#pragma unroll
                for (int by = 0; by < 8; ++by)
#pragma unroll
                    for (int c = 0; c < 2; ++c) {
                        u64 best = ~0ull;
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const uint32_t k16 = (b8[by][c][i >> 1] >> (16 * (i & 1))) & 0xFFFFu;
                            const int      x   = xg * 4 + i;
                            if (x < w) {
                                const u64 kk = ((u64)(k16 >> 3) << 32) | ((uint32_t)(ybase + (k16 & 7)) << 16) | (uint32_t)x;
                                best         = kk < best ? kk : best;
                            }
                        }
                        const int bx8 = col * 2 + c; // 8x8 block column
                        const int pu  = 21 + z16(by >> 1, bx8 >> 1) * 4 + ((by & 1) << 1) + (bx8 & 1);
                        atomicMin(&S.best[pu], best);
                    }
#pragma unroll
                for (int a = 0; a < 4; ++a) {
                    u64 best = ~0ull;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int x = xg * 4 + i;
                        if (x < w) {
                            const u64 kk = ((u64)(b16[a][i] >> 3) << 32) | ((uint32_t)(ybase + (b16[a][i] & 7)) << 16) | (uint32_t)x;
                            best         = kk < best ? kk : best;
                        }
                    }
                    atomicMin(&S.best[5 + z16(a, col)], best);
                }
                if ((col & 1) == 0) {
#pragma unroll
                    for (int a = 0; a < 2; ++a) {
                        u64 best = ~0ull;
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const int x = xg * 4 + i;
                            if (x < w) {
                                const u64 kk = ((u64)(b32[a][i] >> 3) << 32) | ((uint32_t)(ybase + (b32[a][i] & 7)) << 16) | (uint32_t)x;
                                best         = kk < best ? kk : best;
                            }
                        }
                        atomicMin(&S.best[1 + a * 2 + (col >> 1)], best);
                    }
                }
                if (col == 0) {
                    u64 best = ~0ull;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int x = xg * 4 + i;
                        if (x < w) {
                            const u64 kk = ((u64)(b64k[i] >> 3) << 32) | ((uint32_t)(ybase + (b64k[i] & 7)) << 16) | (uint32_t)x;
                            best         = kk < best ? kk : best;
                        }
                    }
                    atomicMin(&S.best[0], best);
                }
            }
        }
    }
    __syncthreads();
    // ---- results: p_sb_best_sad / p_sb_best_mv of (my_l, my_r). SADs were taken on every other row: x2 (:248-305).
    if (tid < SVTME_PU_COUNT) {
        const u64      key = S.best[tid];
        const size_t   o   = ((size_t)(b64 * SVTME_MAX_LISTS + my_l) * SVTME_MAX_REFS + my_r) * SVTME_PU_COUNT + tid;
        const int      py = (int)((key >> 16) & 0xFFFF) + org_y, px = (int)(key & 0xFFFF) + org_x;
        const uint32_t sad = (uint32_t)(key >> 32);
        p.best_sad[o] = k.me_search_method == 0 ? sad << 1 : sad;
        p.best_mv[o]  = ((uint32_t)(uint16_t)py << 18) | (uint16_t)((uint16_t)px << 2); // :474-476
    }
}

void launch_me(const PicParams &p, hipStream_t s) {
    const int n_refs = p.n_ref[0] + (p.n_list == 2 ? p.n_ref[1] : 0);
    dim3      grid(p.b64_w * p.b64_h, n_refs);
    hipLaunchKernelGGL(me_kernel, grid, dim3(ME_THREADS), ME_WIN_BYTES + 64, s, p);
}

} // namespace svtme
