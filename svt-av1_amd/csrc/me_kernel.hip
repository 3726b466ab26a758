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
#include "search_common.h"

namespace svtme {

constexpr int ME_THREADS   = 256;
constexpr int ME_WIN_BYTES = 40 * 1024;
constexpr int ME_RUN       = 8; // search rows per lane run; the row-in-run number needs 3 key bits

struct MeShared {
    u64      best[SVTME_PU_COUNT];
    uint32_t src[32 * 16]; // rows 0,2,..,62 of the source block, 16 dwords each
    uint32_t red;
};

// z-order index of the 16x16 block in raster (row, col): svt_ext_all_sad_calculation_8x8_16x16_c's `offsets` (:366)
__device__ __forceinline__ int z16(int row, int col) { return ((row >> 1) << 3) | ((col >> 1) << 2) | ((row & 1) << 1) | (col & 1); }

__device__ __forceinline__ uint32_t quad_xor1(uint32_t v) { return __shfl_xor(v, 1); }
__device__ __forceinline__ uint32_t quad_xor2(uint32_t v) { return __shfl_xor(v, 2); }

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

    // ---- set_final_seach_centre_sb (:2117-2316): every thread walks the references in order; the running
    // centre / SAD are not reset between references (base-layer list-1 references inherit the previous SAD).
    const HmeRefOut *hme = &p.hme_out[b64 * SVTME_MAX_LISTS * SVTME_MAX_REFS];
    int16_t          sc_x[2][4], sc_y[2][4];
    u64              hsad_tab[2][4];
    uint8_t          do_ref[2][4];
    uint32_t         div[2][4];
#pragma unroll
    for (int l = 0; l < 2; ++l)
#pragma unroll
        for (int r = 0; r < 4; ++r) { // init_me_hme_data :2913-2923
            sc_x[l][r] = sc_y[l][r] = 0;
            hsad_tab[l][r]          = 0xFFFFFFFFull;
            do_ref[l][r]            = 1;
            div[l][r]               = 1;
        }
    {
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
                sc_x[l][r]     = cx;
                sc_y[l][r]     = cy;
                hsad_tab[l][r] = hsad;
            }
    }
    // ---- hme_prune_ref_and_adjust_sr (:2373-2410), over all 2x4 entries
    if (k.enable_hme_flag && (k.enable_me_sr_adjustment || k.enable_me_hme_ref_pruning)) {
        u64 best = ~0ull;
#pragma unroll
        for (int l = 0; l < 2; ++l)
#pragma unroll
            for (int r = 0; r < 4; ++r) best = hsad_tab[l][r] < best ? hsad_tab[l][r] : best;
        const uint32_t th = k.prune_ref_if_hme_sad_dev_bigger_than_th;
#pragma unroll
        for (int l = 0; l < 2; ++l)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (k.enable_me_hme_ref_pruning && (!k.protect_closest_refs || r > 0) && th != 0xFFFF &&
                    (hsad_tab[l][r] - best) * 100 > (u64)th * best)
                    do_ref[l][r] = 0;
                if (k.enable_me_sr_adjustment) {
                    if (abs(sc_x[l][r]) <= k.reduce_me_sr_based_on_mv_length_th && abs(sc_y[l][r]) <= k.reduce_me_sr_based_on_mv_length_th &&
                        hsad_tab[l][r] < k.stationary_hme_sad_abs_th)
                        div[l][r] = k.stationary_me_sr_divisor;
                    else if (hsad_tab[l][r] < k.reduce_me_sr_based_on_hme_sad_abs_th)
                        div[l][r] = k.me_sr_divisor_for_low_hme_sad;
                }
            }
    }
    // the workgroup of the first reference publishes the per-b64 tables
    if (blockIdx.y == 0 && tid < 8) {
        const int            l = tid >> 2, r = tid & 3;
        svtme_search_result &o = p.search_results[b64 * 8 + tid];
        o.hme_sc_x = sc_x[l][r];
        o.hme_sc_y = sc_y[l][r];
        o.do_ref   = do_ref[l][r];
        o.list_i   = (uint8_t)l;
        o.ref_i    = (uint8_t)r;
        o.reserved = 0;
        o.hme_sad  = hsad_tab[l][r];
        p.sr_divisor[b64 * 8 + tid] = div[l][r];
        const bool have = l < p.n_list && r < p.n_ref[l];
        p.zz_sad[b64 * 8 + tid] = have ? hme[tid].zz_sad : ~0u;
    }
    if (!do_ref[my_l][my_r]) return;

    // ---- integer_search_b64 for (my_l, my_r)
    const DevPlane &cur = p.cur.lvl[2];
    const DevPlane &ref = p.ref[my_l][my_r].lvl[2];
    int16_t         cx = sc_x[my_l][my_r], cy = sc_y[my_l][my_r];
    const int       dist = scaled_dist(p.dist[my_l][my_r]);
    int16_t         w = (int16_t)min((int)(int16_t)k.me_sa.sa_min.width * dist, (int)k.me_sa.sa_max.width);
    int16_t         h = (int16_t)min((int)(int16_t)k.me_sa.sa_min.height * dist, (int)k.me_sa.sa_max.height);
    w = (int16_t)((max(1, (int)((uint32_t)w / div[my_l][my_r])) + 7) & ~0x07);
    h = (int16_t)max(3, (int)((uint32_t)h / div[my_l][my_r]));
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
            if ((accurate && best_hme_sad < 24 * 24) || (p.is_ref && hsad_tab[my_l][my_r] < 24 * 24)) h = (int16_t)(h / 2);
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
    int       band_rows = ME_WIN_BYTES / pitch - 62; // search rows per LDS band (a position reads rows y .. y+62)
    if (band_rows < 1) band_rows = 1;
    const int quad = tid >> 2, col = tid & 3;
    const uint32_t *sbase = &S.src[col * 4];

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
            for (int yl = ya; yl < (valid ? yb : ya); ++yl) {
                const uint32_t  yr   = (uint32_t)(yl - ya);
                const uint32_t  ypk  = yr | (yr << 16);
                const uint32_t *wrow = reinterpret_cast<const uint32_t *>(win) + yl * pd + xg + col * 4;
                uint32_t        top[4] = {0, 0, 0, 0}, bot[4] = {0, 0, 0, 0};
#pragma unroll
                for (int a = 0; a < 4; ++a) { // 16-row bands of the block = rows of 16x16 partitions
                    u64 t16 = 0;
#pragma unroll
                    for (int hb = 0; hb < 2; ++hb) {
                        const int by = 2 * a + hb; // 8x8 block row
                        u64       acc0 = 0, acc1 = 0;
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const int       row = by * 8 + 2 * j;
                            const uint32_t *rr  = wrow + row * pd;
                            const uint32_t *ss  = sbase + (row >> 1) * 16;
                            const uint32_t  r0 = rr[0], r1 = rr[1], r2 = rr[2], r3 = rr[3], r4 = rr[4];
                            acc0 = __builtin_amdgcn_qsad_pk_u16_u8(((u64)r1 << 32) | r0, ss[0], acc0);
                            acc0 = __builtin_amdgcn_qsad_pk_u16_u8(((u64)r2 << 32) | r1, ss[1], acc0);
                            acc1 = __builtin_amdgcn_qsad_pk_u16_u8(((u64)r3 << 32) | r2, ss[2], acc1);
                            acc1 = __builtin_amdgcn_qsad_pk_u16_u8(((u64)r4 << 32) | r3, ss[3], acc1);
                        }
                        // 8x8: key16 = sad * 8 + yr (sad <= 8160 -> fits 16 bits), two positions per dword
#pragma unroll
                        for (int c = 0; c < 2; ++c)
#pragma unroll
                            for (int hw = 0; hw < 2; ++hw) {
                                const uint32_t v  = (uint32_t)((c ? acc1 : acc0) >> (32 * hw));
                                const uint32_t kk = (v << 3) + ypk; // per half: sad*8 <= 65280, + yr <= 7: no carry
                                const uint32_t o  = b8[by][c][hw];
                                const uint32_t lo = min(kk & 0xFFFFu, o & 0xFFFFu), hi = min(kk >> 16, o >> 16);
                                b8[by][c][hw]     = lo | (hi << 16);
                            }
                        t16 += acc0 + acc1; // 16-bit fields stay <= 32640
                    }
                    // 16x16 (in-lane): 32-bit key per position
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const uint32_t s = (uint32_t)(t16 >> (16 * i)) & 0xFFFFu;
                        b16[a][i]        = min(b16[a][i], (s << 3) | yr);
                        if (a < 2) top[i] += s;
                        else bot[i] += s;
                    }
                }
                // 32x32 and 64x64: across the quad
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    uint32_t t = top[i], b = bot[i];
                    t += quad_xor1(t);
                    b += quad_xor1(b);
                    b32[0][i] = min(b32[0][i], (t << 3) | yr);
                    b32[1][i] = min(b32[1][i], (b << 3) | yr);
                    uint32_t all = t + b;
                    all += quad_xor2(all);
                    b64k[i] = min(b64k[i], (all << 3) | yr);
                }
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
