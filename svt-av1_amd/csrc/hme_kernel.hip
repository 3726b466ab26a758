// hme_kernel.hip - zero-MV SAD, pre-HME and HME levels 0/1/2 of one picture, for all 64x64 blocks and references.
//
// Replaces hme_b64 up to (not including) set_final_seach_centre_sb, Source/Lib/Encoder/Codec/EbMotionEstimation.c:
//   init_zz_sad / get_zz_sad :2318-2333, :1720-1742     prehme_b64 / prehme_core :1773-1823, :1621-1719
//   hme_level0_b64 / hme_level_0 / get_hme_l0_search_area :1899-2004, :847-947, :1827-1894
//   hme_level1_b64 / hme_level_1 :2009-2068, :950-1049   hme_level2_b64 / hme_level_2 :2073-2112, :1052-1140
// and the kernel they all call, svt_sad_loop_kernel (Source/Lib/Encoder/C_DEFAULT/EbComputeSAD_C.c:58-103).
//
// Mapping: one 256-thread workgroup per (b64, reference) - or per b64, walking its references in order, when a
// preset makes a reference depend on an earlier one. Each search stage stages the reference windows of all its
// independent searches (the 2 pre-HME shapes, the 2x2 regions of a level) in LDS, in row bands that fit the LDS
// budget. A lane owns 4 horizontally adjacent search positions and accumulates them with v_qsad_pk_u16_u8 (one
// instruction = 4 positions x 4 pixels); tall blocks are split over 2-8 lanes by rows and summed with wave shuffles.
// The argmin is exact under the reference's rule (strict '<' in raster order): every lane folds its positions into
// a 64-bit key (SAD << 32 | raster index) and the workgroup takes the minimum key with an LDS atomic.
// VALU-bound; the window bytes come from L2/HBM once per band.
#include "search_common.h"

namespace svtme {

constexpr int HME_THREADS = 256;
constexpr int HME_WIN_BYTES = 24 * 1024; // LDS budget for staged reference windows
constexpr int HME_MAX_BANDS = 48;
constexpr int HME_SRC_PITCH = 16; // dwords per staged source row (64 bytes)

struct SubSearch {
    const uint8_t *pix; // reference plane, pixel (0,0)
    int            stride;
    int            wx, wy; // plane coordinates of search position (0,0)
    int            sa_w, sa_h;
    int            skip; // skip even search rows (svt_sad_loop_kernel's skip_search_line)
};

struct Band {
    int sub, y0, rows, pitch, n_xg, lds_off, task0, batch;
};

struct HmeShared {
    u64       key[4];
    SubSearch sub[4];
    Band      band[HME_MAX_BANDS];
    int       n_band, n_batch;
    uint32_t  src[HME_SRC_PITCH * 64]; // up to 64 rows: a full-resolution block with FULL_SAD_SEARCH
    uint32_t  red;
    HmeRefOut cur; // results of the reference being processed
    // state carried between references in serial mode
    int16_t  pre0_x[SVTME_MAX_REFS][2], pre0_y[SVTME_MAX_REFS][2];
    uint32_t pre0_sad[SVTME_MAX_REFS][2];
    int16_t  l0c00_x, l0c00_y; // level-0 centre of list 0 / ref 0 / region (0,0)
};

// Copies the (subsampled) source block of `plane` at (ox, oy) into S.src: staged row j = block row j * step.
__device__ void stage_src(HmeShared &S, const DevPlane &plane, int ox, int oy, int bw, int rows, int step) {
    const int bwd = (bw + 3) >> 2;
    for (int i = threadIdx.x; i < rows * bwd; i += HME_THREADS) {
        const int r = i / bwd, d = i - r * bwd;
        S.src[r * HME_SRC_PITCH + d] = load_u32_any(plane.pix + (ptrdiff_t)(oy + r * step) * plane.stride + ox + 4 * d);
    }
}

// Exhaustive searches S.sub[0..n_sub) of the staged source block (bw x bh_rows, reference rows `row_step` apart)
// -> S.key[i] = min over positions of (SAD << 32 | y * sa_w + x); ~0 when no position was evaluated.
__device__ void wg_search(HmeShared &S, uint8_t *win, int n_sub, int bw, int bh_rows, int row_step, int chunk_log2,
                          const PicParams &p, int ops_k) {
    const int tid  = threadIdx.x;
    const int bwd  = bw >> 2; // whole source dwords per row
    const int tail = bw & 3; // 0 or 2 leftover source bytes (sixteenth-resolution edge blocks only)
    const int rdw  = bwd + (tail ? 1 : 0);
    const int span = (bh_rows - 1) * row_step + 1;
    if (tid < 4) S.key[tid] = ~0ull;
    if (tid == 0) {
        int nb = 0, batch = 0, off = 0, task = 0;
        unsigned long long n_ops = 0;
        for (int s = 0; s < n_sub; ++s) {
            const int rows_eval = S.sub[s].skip ? S.sub[s].sa_h >> 1 : S.sub[s].sa_h;
            n_ops += (unsigned long long)max(rows_eval, 0) * max(S.sub[s].sa_w, 0) * bh_rows * bw;
            const int n_xg     = (S.sub[s].sa_w + 3) >> 2;
            const int pitch    = (n_xg + rdw) * 4;
            int       max_rows = HME_WIN_BYTES / pitch - (span - 1);
            if (max_rows < 1) max_rows = 1;
            for (int y0 = 0; y0 < S.sub[s].sa_h && nb < HME_MAX_BANDS; y0 += max_rows) {
                const int rows  = min(max_rows, S.sub[s].sa_h - y0);
                const int bytes = pitch * (rows + span - 1);
                if (off && off + bytes > HME_WIN_BYTES) {
                    ++batch;
                    off  = 0;
                    task = 0;
                }
                Band &b   = S.band[nb++];
                b.sub     = s;
                b.y0      = y0;
                b.rows    = rows;
                b.pitch   = pitch;
                b.n_xg    = n_xg;
                b.lds_off = off;
                b.task0   = task;
                b.batch   = batch;
                off += bytes;
                task += (n_xg * rows) << chunk_log2;
            }
        }
        S.n_band  = nb;
        S.n_batch = nb ? batch + 1 : 0;
        if (n_ops) count_ops(p, ops_k, n_ops);
    }
    __syncthreads();
    const int n_band = S.n_band, n_batch = S.n_batch;
    int       b0 = 0;
    for (int batch = 0; batch < n_batch; ++batch) {
        int b1 = b0, total = 0;
        while (b1 < n_band && S.band[b1].batch == batch) {
            total = S.band[b1].task0 + ((S.band[b1].n_xg * S.band[b1].rows) << chunk_log2);
            ++b1;
        }
        // stage the reference windows of this batch, byte-aligned so that position x = 0 sits on a dword
        for (int b = b0; b < b1; ++b) {
            const Band       bd  = S.band[b];
            const SubSearch &sb  = S.sub[bd.sub];
            const int        pd  = bd.pitch >> 2;
            const int        n   = pd * (bd.rows + span - 1);
            uint32_t        *dst = reinterpret_cast<uint32_t *>(win + bd.lds_off);
            for (int i = tid; i < n; i += HME_THREADS) {
                const int r = i / pd, d = i - r * pd;
                dst[i] = load_u32_any(sb.pix + (ptrdiff_t)(sb.wy + bd.y0 + r) * sb.stride + sb.wx + 4 * d);
            }
        }
        __syncthreads();
        const int chunks = 1 << chunk_log2;
        const int rpc    = (bh_rows + chunks - 1) >> chunk_log2; // block rows per lane
        for (int base = 0; base < total; base += HME_THREADS) {
            const int  t     = base + tid;
            const bool valid = t < total;
            int        b     = b0;
            if (valid)
                while (b + 1 < b1 && t >= S.band[b + 1].task0) ++b;
            const Band bd    = S.band[b];
            const int  tl    = valid ? t - bd.task0 : 0;
            const int  chunk = tl & (chunks - 1);
            const int  q     = tl >> chunk_log2;
            const int  yl    = q / bd.n_xg;
            const int  xg    = q - yl * bd.n_xg;
            const int  y     = bd.y0 + yl;
            const SubSearch &sb = S.sub[bd.sub];
            const bool live  = valid && !(sb.skip && (y & 1) == 0);
            uint32_t   a32[4] = {0, 0, 0, 0};
            if (live) {
                const uint32_t *wrow = reinterpret_cast<const uint32_t *>(win + bd.lds_off) + xg;
                const int       pd   = bd.pitch >> 2;
                const int       j0 = chunk * rpc, j1 = min(bh_rows, j0 + rpc);
                u64             a16 = 0;
                int             px  = 0;
                for (int j = j0; j < j1; ++j) {
                    const uint32_t *rr = wrow + (yl + j * row_step) * pd;
                    const uint32_t *ss = &S.src[j * HME_SRC_PITCH];
                    if (px + bw > 256) { // keep the packed 16-bit sums below 65536
#pragma unroll
                        for (int i = 0; i < 4; ++i) a32[i] += (uint32_t)(a16 >> (16 * i)) & 0xFFFFu;
                        a16 = 0;
                        px  = 0;
                    }
                    uint32_t r0 = rr[0];
                    for (int k = 0; k < bwd; ++k) {
                        const uint32_t r1 = rr[k + 1];
                        a16 = __builtin_amdgcn_qsad_pk_u16_u8(((u64)r1 << 32) | r0, ss[k], a16);
                        r0  = r1;
                    }
                    if (tail) {
                        const u64      w = ((u64)rr[bwd + 1] << 32) | r0;
                        const uint32_t s = ss[bwd] & 0xFFFFu;
#pragma unroll
                        for (int i = 0; i < 4; ++i) a32[i] = __builtin_amdgcn_sad_u8((uint32_t)(w >> (8 * i)) & 0xFFFFu, s, a32[i]);
                    }
                    px += bw;
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) a32[i] += (uint32_t)(a16 >> (16 * i)) & 0xFFFFu;
            }
            for (int m = 1; m < chunks; m <<= 1) {
#pragma unroll
                for (int i = 0; i < 4; ++i) a32[i] += __shfl_xor(a32[i], m);
            }
            if (live && chunk == 0) {
                u64 best = ~0ull;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int x = xg * 4 + i;
                    if (x < sb.sa_w) {
                        const u64 k = ((u64)a32[i] << 32) | (uint32_t)(y * sb.sa_w + x);
                        best        = k < best ? k : best;
                    }
                }
                atomicMin(&S.key[bd.sub], best);
            }
        }
        __syncthreads();
        b0 = b1;
    }
}

// Decodes S.key[i] the way the callers of svt_sad_loop_kernel post-process its outputs: SAD x2 when rows were
// subsampled, position + window origin, times the level's scale (:938-944, :1040-1046, :1133-1137, :1710-1716).
__device__ __forceinline__ void decode_result(const HmeShared &S, int i, int org_x, int org_y, int scale, bool sub_sad,
                                              int16_t &mx, int16_t &my, uint32_t &sad) {
    const u64 k = S.key[i];
    if (k == ~0ull) { // no position evaluated: svt_sad_loop_kernel leaves 0xffffff and does not touch x / y
        sad = sub_sad ? 0xffffffu * 2 : 0xffffffu;
        mx  = (int16_t)(org_x * scale);
        my  = (int16_t)(org_y * scale);
        return;
    }
    const int sa_w = S.sub[i].sa_w;
    const int idx  = (int)(uint32_t)k;
    const int y = idx / sa_w, x = idx - y * sa_w;
    sad = (uint32_t)(k >> 32);
    if (sub_sad) sad *= 2;
    mx = (int16_t)((int16_t)(x + org_x) * scale);
    my = (int16_t)((int16_t)(y + org_y) * scale);
}

// get_hme_l0_search_area :1827-1894 on a working copy of hme_l0_sa (the reference rescales the context's copy in
// place and restores it after the reference, :1960-1965).
__device__ void hme_l0_area(const PicParams &p, const HmeShared &S, int l, int r, int16_t &sa_w, int16_t &sa_h) {
    const svtme_controls &k  = p.ctrl;
    svtme_sa_minmax       sa = k.hme_l0_sa;
    if (k.enable_me_sr_adjustment && k.distance_based_hme_resizing) {
        int is_hor = 1, is_ver = 1, is_still = 0;
        if (k.reduce_hme_l0_sr_th_min && k.reduce_hme_l0_sr_th_max && (l || r)) {
            const int mvx = S.l0c00_x, mvy = S.l0c00_y;
            is_ver   = abs(mvx) < k.reduce_hme_l0_sr_th_min && abs(mvy) > k.reduce_hme_l0_sr_th_max;
            is_hor   = abs(mvx) > k.reduce_hme_l0_sr_th_max && abs(mvy) < k.reduce_hme_l0_sr_th_min;
            is_still = abs(mvx) < k.reduce_hme_l0_sr_th_min * 3 && abs(mvy) < k.reduce_hme_l0_sr_th_min * 3;
        }
        int x_off = is_hor ? 1 : 2, y_off = is_ver ? 1 : 2;
        if (k.enable_me_sr_adjustment == 2 && is_still) x_off = y_off = 4;
        sa.sa_min.width  = (uint16_t)(sa.sa_min.width / (x_off + r));
        sa.sa_min.height = (uint16_t)(sa.sa_min.height / (y_off + r));
        sa.sa_max.width  = (uint16_t)(sa.sa_max.width / (x_off + r));
        sa.sa_max.height = (uint16_t)(sa.sa_max.height / (y_off + r));
    }
    const int factor = scaled_dist(p.dist[l][r]);
    int16_t   w      = (int16_t)(sa.sa_min.width / k.num_hme_sa_w);
    w = (int16_t)min(((w * factor) + 15) & ~0x0F, ((sa.sa_max.width / k.num_hme_sa_w) + 15) & ~0x0F);
    int16_t h = (int16_t)(sa.sa_min.height / k.num_hme_sa_h);
    h         = (int16_t)min(h * factor, sa.sa_max.height / k.num_hme_sa_h);
    sa_w      = w;
    sa_h      = h;
}

// One HME level for one reference: the num_hme_sa_w x num_hme_sa_h regions searched together.
// level 0: windows placed by region index; levels 1, 2: centred on the previous level's vector (>> 1 for level 1).
__device__ void hme_level(const PicParams &p, HmeShared &S, uint8_t *win, int level, int l, int r, int b64x, int b64y,
                          int block_w, int block_h, int16_t sa_w_in, int16_t sa_h_in) {
    const svtme_controls &k     = p.ctrl;
    const DevPlane       &cur   = p.cur.lvl[level];
    const DevPlane       &ref   = p.ref[l][r].lvl[level];
    const int             shift = 2 - level;
    const int             ox = b64x >> shift, oy = b64y >> shift;
    const int             bw = block_w >> shift, bh = block_h >> shift;
    const bool            sub_sad = k.hme_search_method == 0;
    const int             rows = sub_sad ? bh >> 1 : bh, step = sub_sad ? 2 : 1;
    const int             pad  = level == 2 ? 63 : ref.pad - 1;
    const int             nw = k.num_hme_sa_w, nh = k.num_hme_sa_h;
    int16_t               org_x[4], org_y[4];
    // every thread derives the same windows; thread 0 publishes them
    for (int sh = 0; sh < nh; ++sh)
        for (int sw = 0; sw < nw; ++sw) {
            const int i = sw * 2 + sh;
            int16_t   w = (int16_t)((sa_w_in + 7) & ~0x07), h = sa_h_in;
            if (level == 0) {
                org_x[i] = (int16_t)(-(int16_t)((w * nw) >> 1) + (int16_t)(w * sw));
                org_y[i] = (int16_t)(-(int16_t)((h * nh) >> 1) + (int16_t)(h * sh));
            } else {
                int16_t cx = S.cur.lx[level - 1][i], cy = S.cur.ly[level - 1][i];
                if (level == 1) {
                    cx = (int16_t)(cx >> 1);
                    cy = (int16_t)(cy >> 1);
                }
                org_x[i] = (int16_t)(-(w >> 1) + cx);
                org_y[i] = (int16_t)(-(h >> 1) + cy);
            }
            clip_axis(ox, org_x[i], w, -pad, ref.width);
            w = (int16_t)(w < 8 ? w : (w & ~0x07));
            clip_axis(oy, org_y[i], h, -pad, ref.height);
            if (threadIdx.x == 0) {
                SubSearch &s = S.sub[sw + sh * nw]; // search slots in the reference's visiting order (sr_h outer)
                s.pix    = ref.pix;
                s.stride = ref.stride;
                s.wx     = ox + org_x[i];
                s.wy     = oy + org_y[i];
                s.sa_w   = w;
                s.sa_h   = h;
                s.skip   = 0;
            }
        }
    stage_src(S, cur, ox, oy, bw, rows, step);
    __syncthreads();
    const int chunk_log2 = rows >= 32 ? 3 : (rows >= 16 ? 2 : 0);
    wg_search(S, win, nw * nh, bw, rows, step, chunk_log2, p, 2 + level);
    if (threadIdx.x == 0) {
        const int scale = level == 0 ? 4 : (level == 1 ? 2 : 1);
        for (int sh = 0; sh < nh; ++sh)
            for (int sw = 0; sw < nw; ++sw) {
                const int i = sw * 2 + sh;
                decode_result(S, sw + sh * nw, org_x[i], org_y[i], scale, sub_sad, S.cur.lx[level][i], S.cur.ly[level][i],
                              S.cur.lsad[level][i]);
            }
    }
    __syncthreads();
}

__device__ void zero_level(const PicParams &p, HmeShared &S, int level) {
    if (threadIdx.x == 0)
        for (int sh = 0; sh < p.ctrl.num_hme_sa_h; ++sh)
            for (int sw = 0; sw < p.ctrl.num_hme_sa_w; ++sw) {
                S.cur.lx[level][sw * 2 + sh] = S.cur.ly[level][sw * 2 + sh] = 0;
                S.cur.lsad[level][sw * 2 + sh]                              = 0;
            }
    __syncthreads();
}

__device__ void hme_one_ref(const PicParams &p, HmeShared &S, uint8_t *win, int b64, int l, int r) {
    const svtme_controls &k       = p.ctrl;
    const int             b64x = (b64 % p.b64_w) * 64, b64y = (b64 / p.b64_w) * 64;
    const int             block_w = min(p.width - b64x, 64), block_h = min(p.height - b64y, 64);
    const bool            searched = ref_searched(p, l);
    const bool            sub_sad  = k.hme_search_method == 0;
    HmeRefOut            *out = &p.hme_out[(b64 * SVTME_MAX_LISTS + l) * SVTME_MAX_REFS + r];

    if (threadIdx.x < sizeof(HmeRefOut) / 4) reinterpret_cast<uint32_t *>(&S.cur)[threadIdx.x] = 0;
    __syncthreads();

    // zero-MV SAD (init_zz_sad :2318-2333); stays ~0 (init_me_hme_data :2921) when not computed
    uint32_t zz = ~0u;
    if (k.me_early_exit_th && searched)
        zz = wg_block_sad(p.cur.lvl[2], p.ref[l][r].lvl[2], b64x, b64y, 0, 0, block_w, block_h >> 1, 2, &S.red) << 1;
    if (threadIdx.x == 0) {
        if (k.me_early_exit_th && searched) count_ops(p, 0, (unsigned long long)(block_h >> 1) * block_w);
        S.cur.zz_sad = zz;
        S.cur.valid  = searched ? 1u : 0u;
    }

    // pre-HME (prehme_b64 :1773-1823)
    if (k.prehme_enable) {
        if (searched) {
            const DevPlane &cur = p.cur.lvl[0];
            const DevPlane &ref = p.ref[l][r].lvl[0];
            const int       ox = b64x >> 2, oy = b64y >> 2;
            const int       bw = block_w >> 2, bh = block_h >> 2;
            const int       rows = sub_sad ? bh >> 1 : bh, step = sub_sad ? 2 : 1;
            const int       factor = scaled_dist(p.dist[l][r]);
            int16_t         org_x[2], org_y[2];
            int             slot[2], n_sub = 0;
            for (int s = 0; s < 2; ++s) {
                slot[s] = -1;
                // check_prehme_early_exit :1746-1770
                if (k.me_early_exit_th && zz < k.me_early_exit_th) {
                    if (threadIdx.x == 0) {
                        S.cur.pre_x[s] = S.cur.pre_y[s] = 0;
                        S.cur.pre_sad[s]                = 0;
                    }
                    continue;
                }
                if (k.prehme_l1_early_exit && l == 1 &&
                    (S.pre0_sad[r][s] < 32 * 32 || (abs(S.pre0_x[r][s]) < 16 && abs(S.pre0_y[r][s]) < 16))) {
                    if (threadIdx.x == 0) {
                        S.cur.pre_x[s]   = (int16_t)-S.pre0_x[r][s];
                        S.cur.pre_y[s]   = (int16_t)-S.pre0_y[r][s];
                        S.cur.pre_sad[s] = S.pre0_sad[r][s];
                    }
                    continue;
                }
                const svtme_sa_minmax &cfg = k.prehme_sa_cfg[s];
                int16_t w = (int16_t)(uint16_t)min((uint32_t)cfg.sa_min.width * factor, (uint32_t)cfg.sa_max.width);
                int16_t h = (int16_t)(uint16_t)min((uint32_t)cfg.sa_min.height * factor, (uint32_t)cfg.sa_max.height);
                org_x[s]  = (int16_t)-(int16_t)(w >> 1);
                org_y[s]  = (int16_t)-(int16_t)(h >> 1);
                clip_axis(ox, org_x[s], w, -(ref.pad - 1), ref.width);
                clip_axis(oy, org_y[s], h, -(ref.pad - 1), ref.height);
                slot[s] = n_sub;
                if (threadIdx.x == 0) {
                    SubSearch &ss = S.sub[n_sub];
                    ss.pix    = ref.pix;
                    ss.stride = ref.stride;
                    ss.wx     = ox + org_x[s];
                    ss.wy     = oy + org_y[s];
                    ss.sa_w   = w;
                    ss.sa_h   = h;
                    // svt_sad_loop_kernel_c skips even rows only for 16-wide blocks of height <= 16
                    ss.skip   = k.prehme_skip_search_line && bw == 16 && rows <= 16;
                }
                ++n_sub;
            }
            if (n_sub) {
                stage_src(S, cur, ox, oy, bw, rows, step);
                __syncthreads();
                wg_search(S, win, n_sub, bw, rows, step, 0, p, 1);
                if (threadIdx.x == 0)
                    for (int s = 0; s < 2; ++s)
                        if (slot[s] >= 0)
                            decode_result(S, slot[s], org_x[s], org_y[s], 4, sub_sad, S.cur.pre_x[s], S.cur.pre_y[s], S.cur.pre_sad[s]);
            }
            __syncthreads();
            if (l == 0 && threadIdx.x == 0)
                for (int s = 0; s < 2; ++s) {
                    S.pre0_x[r][s]   = S.cur.pre_x[s];
                    S.pre0_y[r][s]   = S.cur.pre_y[s];
                    S.pre0_sad[r][s] = S.cur.pre_sad[s];
                }
        }
        // base-layer list 1 mirrors list 0 (:1811-1819); nothing downstream reads it (HME is skipped for it)
    }
    __syncthreads();

    if (k.enable_hme_flag) {
        // level 0 (:1899-2004): the early exit is tested before the base-layer list-1 exclusion
        if (k.enable_hme_level0_flag) {
            if (k.me_early_exit_th && zz < (k.me_early_exit_th >> 2)) zero_level(p, S, 0);
            else if (searched) {
                int16_t sa_w = 0, sa_h = 0;
                hme_l0_area(p, S, l, r, sa_w, sa_h);
                hme_level(p, S, win, 0, l, r, b64x, b64y, block_w, block_h, sa_w, sa_h);
                if (k.prehme_enable && threadIdx.x == 0) {
                    // the worst region (strict '>', sr_h outer / sr_w inner) gives way to the better pre-HME result (:1967-1999)
                    uint32_t max_sad = 0;
                    int      mi      = 0;
                    for (int sh = 0; sh < k.num_hme_sa_h; ++sh)
                        for (int sw = 0; sw < k.num_hme_sa_w; ++sw)
                            if (S.cur.lsad[0][sw * 2 + sh] > max_sad) {
                                max_sad = S.cur.lsad[0][sw * 2 + sh];
                                mi      = sw * 2 + sh;
                            }
                    const int s = S.cur.pre_sad[0] <= S.cur.pre_sad[1] ? 0 : 1;
                    if (S.cur.pre_sad[s] < S.cur.lsad[0][mi]) {
                        S.cur.lsad[0][mi] = S.cur.pre_sad[s];
                        S.cur.lx[0][mi]   = S.cur.pre_x[s];
                        S.cur.ly[0][mi]   = S.cur.pre_y[s];
                    }
                }
                __syncthreads();
                if (l == 0 && r == 0 && threadIdx.x == 0) {
                    S.l0c00_x = S.cur.lx[0][0];
                    S.l0c00_y = S.cur.ly[0][0];
                }
            }
        }
        if (k.enable_hme_level1_flag && searched) { // :2009-2068
            if (k.me_early_exit_th && zz < (k.me_early_exit_th >> 2)) zero_level(p, S, 1);
            else hme_level(p, S, win, 1, l, r, b64x, b64y, block_w, block_h, (int16_t)k.hme_l1_sa.width, (int16_t)k.hme_l1_sa.height);
        }
        if (k.enable_hme_level2_flag && searched) // :2073-2112, no early exit at this level
            hme_level(p, S, win, 2, l, r, b64x, b64y, block_w, block_h, (int16_t)k.hme_l2_sa.width, (int16_t)k.hme_l2_sa.height);
    }
    __syncthreads();
    if (threadIdx.x < sizeof(HmeRefOut) / 4)
        reinterpret_cast<uint32_t *>(out)[threadIdx.x] = reinterpret_cast<const uint32_t *>(&S.cur)[threadIdx.x];
    __syncthreads();
}

template <class PICS>
__global__ void __launch_bounds__(HME_THREADS) hme_kernel(PICS pp, int n_b64) {
    __shared__ HmeShared                  S;
    extern __shared__ __align__(16) uint8_t win[];
    const PicParams                      &p   = pp[blockIdx.x / n_b64];
    const int                             b64 = blockIdx.x % n_b64;
    if (b64_left_out(p, b64)) return;
    if (threadIdx.x == 0) {
        S.l0c00_x = S.l0c00_y = 0;
        for (int r = 0; r < SVTME_MAX_REFS; ++r)
            for (int s = 0; s < 2; ++s) {
                S.pre0_x[r][s] = S.pre0_y[r][s] = 0;
                S.pre0_sad[r][s]                = 0;
            }
    }
    __syncthreads();
    if (gridDim.y == 1) {
        for (int l = 0; l < p.n_list; ++l)
            for (int r = 0; r < p.n_ref[l]; ++r) hme_one_ref(p, S, win, b64, l, r);
    } else {
        int g = blockIdx.y, l = 0;
        if (g >= p.n_ref[0] + (p.n_list == 2 ? p.n_ref[1] : 0)) return; // the grid covers the batch's longest reference list
        if (g >= p.n_ref[0]) {
            g -= p.n_ref[0];
            l = 1;
        }
        hme_one_ref(p, S, win, b64, l, g);
    }
}

void launch_hme(const Batch &b, hipStream_t s, hipEvent_t marks[2]) {
    // Kernels of independent work items, unless the caller asked for the serial kernel (one workgroup per b64 walks the
    // references in order). (The itemised full-SAD form has no row skipping for pre-HME; no preset combines the two.)
    const PicParams &p = b.host[0];
    const bool fs_skip = p.ctrl.hme_search_method != 0 && p.ctrl.prehme_enable && p.ctrl.prehme_skip_search_line;
    if (!p.serial_refs && p.ctrl.hme_search_method <= 1 && !fs_skip) {
        launch_hme_items(b, s, marks);
        return;
    }
    bool dep = p.serial_refs != 0;
    for (int i = 0; i < b.n; ++i) dep = dep || b.host[i].dep_pre || b.host[i].dep_l0;
    dim3 grid(b.n_b64() * b.n, dep ? 1 : b.max_refs());
    // (the serial kernel is a diagnostic: it takes the pictures of a batch one launch at a time)
    (void)grid;
    for (int i = 0; i < b.n; ++i) {
        const PicParams &q = b.host[i];
        const bool       dep_i = q.serial_refs || q.dep_pre || q.dep_l0;
        hipLaunchKernelGGL(hme_kernel<OnePicture>, dim3(b.n_b64(), dep_i ? 1 : q.n_ref[0] + (q.n_list == 2 ? q.n_ref[1] : 0)), dim3(HME_THREADS),
                           HME_WIN_BYTES + 64, s, OnePicture{q}, b.n_b64());
    }
    if (marks[0]) (void)hipEventRecord(marks[0], s);
    if (marks[1]) (void)hipEventRecord(marks[1], s);
}

} // namespace svtme
