// gm_kernels.hip - the data-parallel stages of global-motion estimation on the store's planes:
//   FAST-9 corners           svt_av1_fast_corner_detect, Source/Lib/Encoder/Codec/corner_detect.c:19-32
//                            (third_party/fastfeat/fast.c:6-21, fast_9.c, nonmax.c)
//   corner correspondences   svt_av1_determine_correspondence + improve_correspondence, corner_match.c:80-214
//   warp / frame error       svt_av1_warp_error / svt_av1_frame_error for 8 bit, EbEncWarpedMotion.c:162-267, with
//                            svt_av1_warp_affine_c, Source/Lib/Common/Codec/EbWarpedMotion.c:577-694
// All integer arithmetic except the match measure, which is the reference's double expression (one multiplication and one
// division, both correctly rounded on the device as on the host), so every result is the reference's bit for bit.
// Byte work on pictures that sit in L2 / HBM: nothing here is shaped like a matrix product.
#include "svtme_device.h"
#include "search_common.h"

namespace svtme {

namespace {

// ---------------------------------------------------------------------------------------------------------- corners
// Score map: 0 = no corner at barrier 18, else the corner's score (18 .. 254). The segment test of fast_9.c in closed form:
// a pixel is a corner at threshold b when, for 9 contiguous pixels of the 16-pixel circle, every (ring - p) > b, or every
// (p - ring) > b; its score (the bisection of aom_fast9_corner_score, fast_9.c:8-2936) is the largest such b, i.e. the
// largest over the 16 arcs of the smallest difference in the arc, minus one.
__global__ __launch_bounds__(256) void gm_score_kernel(DevPlane pl, uint8_t *score) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= pl.width || y >= pl.height) return;
    int s = 0;
    if (x >= 3 && y >= 3 && x < pl.width - 3 && y < pl.height - 3) {
        const uint8_t *p = pl.pix + (ptrdiff_t)y * pl.stride + x;
        const int      c = p[0], st = pl.stride;
        int            d[16]; // make_offsets, fast_9.c:2939-2957
        d[0]  = p[3 * st] - c;      d[1]  = p[3 * st + 1] - c;  d[2]  = p[2 * st + 2] - c;  d[3]  = p[st + 3] - c;
        d[4]  = p[3] - c;           d[5]  = p[-st + 3] - c;     d[6]  = p[-2 * st + 2] - c; d[7]  = p[-3 * st + 1] - c;
        d[8]  = p[-3 * st] - c;     d[9]  = p[-3 * st - 1] - c; d[10] = p[-2 * st - 2] - c; d[11] = p[-st - 3] - c;
        d[12] = p[-3] - c;          d[13] = p[st - 3] - c;      d[14] = p[2 * st - 2] - c;  d[15] = p[3 * st - 1] - c;
        // minimum and maximum over every window of 9 of the circle, by doubling
        int lo2[16], hi2[16], lo4[16], hi4[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            lo2[i] = min(d[i], d[(i + 1) & 15]);
            hi2[i] = max(d[i], d[(i + 1) & 15]);
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            lo4[i] = min(lo2[i], lo2[(i + 2) & 15]);
            hi4[i] = max(hi2[i], hi2[(i + 2) & 15]);
        }
        int best = 0;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int lo9 = min(min(lo4[i], lo4[(i + 4) & 15]), d[(i + 8) & 15]);
            const int hi9 = max(max(hi4[i], hi4[(i + 4) & 15]), d[(i + 8) & 15]);
            best          = max(best, max(lo9, -hi9)); // brighter arc: all of ring - p; darker arc: all of p - ring
        }
        s = best > 18 ? best - 1 : 0;
    }
    score[(size_t)y * pl.width + x] = (uint8_t)s;
}

// Non-maximum suppression (nonmax.c:8-119, Compare = ">="): a corner stays when none of its 8 neighbours is a corner with
// a score >= its own. One workgroup per row; keep[] gets the flags, row_count[] the number of corners kept in the row.
__global__ __launch_bounds__(256) void gm_nonmax_kernel(const uint8_t *score, int w, int h, uint8_t *keep, int *row_count) {
    __shared__ int cnt;
    const int      y = blockIdx.x;
    if (threadIdx.x == 0) cnt = 0;
    __syncthreads();
    int mine = 0;
    for (int x = threadIdx.x; x < w; x += 256) {
        const uint8_t *r = score + (size_t)y * w + x;
        const int      s = r[0];
        int            k = 0;
        if (s) { // (corners lie at least 3 pixels inside the picture: the neighbours exist)
            const int m = max(max(max(r[-w - 1], r[-w]), max(r[-w + 1], r[-1])), max(max(r[1], r[w - 1]), max(r[w], r[w + 1])));
            k           = m < s;
        }
        keep[(size_t)y * w + x] = (uint8_t)k;
        mine += k;
    }
    if (mine) atomicAdd(&cnt, mine);
    __syncthreads();
    if (threadIdx.x == 0) row_count[y] = cnt;
}

// Raster-order compaction: the corners of row y go behind those of the rows above it; only the first max_points are
// written. out[0] = min(total, max_points), then (x, y) pairs from out[2] on.
__global__ __launch_bounds__(256) void gm_compact_kernel(const uint8_t *keep, int w, int h, const int *row_count, int max_points, int *out) {
    __shared__ int part[256];
    __shared__ int wave_cnt[4];
    const int      y = blockIdx.x, t = threadIdx.x, lane = t & 63, wv = t >> 6;
    int            before = 0;
    for (int r = t; r < y; r += 256) before += row_count[r];
    part[t] = before;
    __syncthreads();
    for (int s = 128; s >= 1; s >>= 1) {
        if (t < s) part[t] += part[t + s];
        __syncthreads();
    }
    int base = part[0];
    if (y == h - 1 && t == 0) out[0] = min(base + row_count[y], max_points);
    if (base >= max_points) return; // (the whole workgroup takes this branch together)
    for (int x0 = 0; x0 < w; x0 += 256) {
        const int                x = x0 + t;
        const int                k = x < w ? keep[(size_t)y * w + x] : 0;
        const unsigned long long b = __ballot(k);
        if (lane == 0) wave_cnt[wv] = __popcll(b);
        __syncthreads();
        int idx = base + __popcll(b & ((1ull << lane) - 1));
        for (int i = 0; i < wv; ++i) idx += wave_cnt[i];
        if (k && idx < max_points) {
            out[2 + 2 * idx]     = x;
            out[2 + 2 * idx + 1] = y;
        }
        base += wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------------- correspondences
constexpr int kPatch = 13, kPatchBy2 = 6, kPatchSq = 169;

__device__ __forceinline__ bool patch_inside(int x, int y, int w, int h) {
    return x >= kPatchBy2 && y >= kPatchBy2 && x + kPatchBy2 < w && y + kPatchBy2 < h;
}

// svt_av1_compute_cross_correlation_c (corner_match.c:42-66) of a 13x13 template held in LDS (`tpl`, its pixel sum `sum1`)
// against the patch of `im` centred at (x, y); as the ordered bits of the double: 0 for "not a candidate" (the reference's
// 0.0 and its NaN of a flat patch, neither of which ever wins a "strictly larger" test that starts from 0.0).
__device__ __forceinline__ unsigned long long match_bits(const uint8_t *tpl, int sum1, const DevPlane &im, int x, int y) {
    const uint8_t *g = im.pix + (ptrdiff_t)(y - kPatchBy2) * im.stride + (x - kPatchBy2);
    int            sum2 = 0, sumsq2 = 0, cross = 0;
    for (int i = 0; i < kPatch; ++i, g += im.stride)
#pragma unroll
        for (int j = 0; j < kPatch; ++j) {
            const int v2 = g[j];
            sum2 += v2;
            sumsq2 += v2 * v2;
            cross += tpl[i * kPatch + j] * v2;
        }
    const int var2 = sumsq2 * kPatchSq - sum2 * sum2, cov = cross * kPatchSq - sum1 * sum2;
    if (cov <= 0 || var2 == 0) return 0;
    const double ncc = ((double)cov * (double)cov) / (double)var2;
    return (unsigned long long)__double_as_longlong(ncc); // positive, finite: the bit pattern orders like the value
}

// lane-wise (value, index) pairs -> the largest value and, among equals, the smallest index, in every lane
__device__ __forceinline__ void wave_argmax(unsigned long long &v, int &idx) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        const unsigned long long ov = __shfl_xor(v, m);
        const int                oi = __shfl_xor(idx, m);
        if (ov > v || (ov == v && oi < idx)) {
            v   = ov;
            idx = oi;
        }
    }
}

// 13x13 patch of `im` around (x, y) -> LDS; returns its pixel sum and sum of squares in every lane
__device__ __forceinline__ void load_patch(uint8_t *tpl, const DevPlane &im, int x, int y, int lane, int &sum, int &sumsq) {
    int s = 0, q = 0;
    for (int i = lane; i < kPatchSq; i += 64) {
        const int r = i / kPatch, c = i - r * kPatch;
        const int v = im.pix[(ptrdiff_t)(y - kPatchBy2 + r) * im.stride + (x - kPatchBy2 + c)];
        tpl[i]      = (uint8_t)v;
        s += v;
        q += v * v;
    }
    sum   = (int)wave_sum((uint32_t)s);
    sumsq = (int)wave_sum((uint32_t)q);
    wave_sync();
}

// One wave per corner of the current picture: the eligible reference corners (patch inside the picture, within the
// distance threshold) are gathered into a queue in LDS, in index order, so that the 64 lanes each match one of them at a
// time; then the two 9x9 refinements of improve_correspondence for this corner (they depend on nothing but the corner's own
// match). out[i] = {x, y, rx, ry, matched}; the host keeps the matched ones in order.
__global__ __launch_bounds__(64) void gm_match_kernel(DevPlane frm, DevPlane ref, const int *frm_corners, const int *ref_corners, int *out) {
    __shared__ uint8_t  tpl[kPatchSq + 3];
    __shared__ uint16_t queue[SVTME_GM_MAX_CORNERS];
    const int           i = blockIdx.x, lane = threadIdx.x, n_frm = frm_corners[0], n_ref = ref_corners[0];
    if (i >= n_frm) return;
    const int w = frm.width, h = frm.height;
    const int thresh = max(w, h) >> 4, thresh_sq = thresh * thresh;
    int       fx = frm_corners[2 + 2 * i], fy = frm_corners[2 + 2 * i + 1];
    int      *o = out + 5 * i;
    if (!patch_inside(fx, fy, w, h)) {
        if (lane == 0) o[4] = 0;
        return;
    }
    int sum1, sumsq1;
    load_patch(tpl, frm, fx, fy, lane, sum1, sumsq1);
    // eligible reference corners, in index order
    int n_q = 0;
    for (int j0 = 0; j0 < n_ref; j0 += 64) {
        const int j = j0 + lane;
        bool      ok = false;
        if (j < n_ref) {
            const int rx = ref_corners[2 + 2 * j], ry = ref_corners[2 + 2 * j + 1], dx = fx - rx, dy = fy - ry;
            ok           = patch_inside(rx, ry, w, h) && dx * dx + dy * dy <= thresh_sq;
        }
        const unsigned long long b = __ballot(ok);
        if (ok) queue[n_q + __popcll(b & ((1ull << lane) - 1))] = (uint16_t)j;
        n_q += __popcll(b);
    }
    wave_sync();
    unsigned long long best = 0;
    int                best_j = 0x7fffffff;
    for (int q = lane; q < n_q; q += 64) {
        const int                j = queue[q];
        const unsigned long long v = match_bits(tpl, sum1, ref, ref_corners[2 + 2 * j], ref_corners[2 + 2 * j + 1]);
        if (v > best) { // a lane meets its candidates in index order: the first of equals stays
            best   = v;
            best_j = j;
        }
    }
    wave_argmax(best, best_j);
    // "best correlation >= 0.75", normalised as the measure is (corner_match.c:191-196)
    const int template_norm = sumsq1 * kPatchSq - sum1 * sum1;
    if (!(best && __longlong_as_double((long long)best) > (double)template_norm * 0.75 * 0.75)) {
        if (lane == 0) o[4] = 0;
        return;
    }
    int rx = ref_corners[2 + 2 * best_j], ry = ref_corners[2 + 2 * best_j + 1];
    // improve_correspondence, first loop (corner_match.c:86-118): the reference point moves within 9x9 positions
    {
        unsigned long long b = 0;
        int                bp = 0x7fffffff;
        for (int p = lane; p < 81; p += 64) {
            const int y = p / 9 - 4, x = p - (p / 9) * 9 - 4, dx = fx - (rx + x), dy = fy - (ry + y);
            if (!patch_inside(rx + x, ry + y, w, h) || dx * dx + dy * dy > thresh_sq) continue;
            const unsigned long long v = match_bits(tpl, sum1, ref, rx + x, ry + y);
            if (v > b) {
                b  = v;
                bp = p;
            }
        }
        wave_argmax(b, bp);
        if (b) {
            ry += bp / 9 - 4;
            rx += bp - (bp / 9) * 9 - 4;
        }
    }
    // second loop (:119-150): the frame point moves, the template is the reference patch now
    wave_sync();
    int sum2, sumsq2;
    load_patch(tpl, ref, rx, ry, lane, sum2, sumsq2);
    {
        unsigned long long b = 0;
        int                bp = 0x7fffffff;
        for (int p = lane; p < 81; p += 64) {
            const int y = p / 9 - 4, x = p - (p / 9) * 9 - 4, dx = fx + x - rx, dy = fy + y - ry;
            if (!patch_inside(fx + x, fy + y, w, h) || dx * dx + dy * dy > thresh_sq) continue;
            const unsigned long long v = match_bits(tpl, sum2, frm, fx + x, fy + y);
            if (v > b) {
                b  = v;
                bp = p;
            }
        }
        wave_argmax(b, bp);
        if (b) {
            fy += bp / 9 - 4;
            fx += bp - (bp / 9) * 9 - 4;
        }
    }
    if (lane == 0) {
        o[0] = fx;
        o[1] = fy;
        o[2] = rx;
        o[3] = ry;
        o[4] = 1;
    }
}

// -------------------------------------------------------------------------------------------------- warp / frame error
// One workgroup of four waves per 32x32 block of the error measure (WARP_ERROR_BLOCK, EbEncWarpedMotion.c:20), a wave per
// 16x16 quadrant: its 8x8 blocks one after the other, 15 x 8 horizontally filtered values through LDS, then a lane per output
// pixel. The waves share the filter and the error table in LDS; block_err[block] = the block's sum of
// error_measure(frame - warped). identity != 0: no warp, the frame error against the reference picture itself.
struct alignas(16) WarpTables {
    int16_t filter[193 * 8]; // a filter's eight coefficients: 16 bytes, read as one piece
    int32_t lut[512];
    int32_t tmp[4][15 * 8];
    uint32_t part[4];
};

__device__ __forceinline__ void stage_warp_tables(WarpTables &t, const int16_t *filter, const int32_t *lut, bool with_filter) {
    for (int k = threadIdx.x; k < 512; k += 256) t.lut[k] = lut[k];
    if (with_filter)
        for (int k = threadIdx.x; k < 193 * 8 / 2; k += 256) reinterpret_cast<uint32_t *>(t.filter)[k] = reinterpret_cast<const uint32_t *>(filter)[k];
    __syncthreads();
}

typedef short short2_t __attribute__((ext_vector_type(2)));
// acc + a.lo * b.lo + a.hi * b.hi on the signed 16-bit halves
__device__ __forceinline__ int32_t dot2(int32_t a, int32_t b, int32_t acc) {
    return __builtin_amdgcn_sdot2(__builtin_bit_cast(short2_t, a), __builtin_bit_cast(short2_t, b), acc, false);
}
// acc + the eight bytes of (lo, hi) times the eight 16-bit coefficients of f
__device__ __forceinline__ int32_t taps8(uint32_t lo, uint32_t hi, const int4 &f, int32_t acc) {
    acc = dot2((int32_t)__builtin_amdgcn_perm(0, lo, 0x0c010c00), f.x, acc);
    acc = dot2((int32_t)__builtin_amdgcn_perm(0, lo, 0x0c030c02), f.y, acc);
    acc = dot2((int32_t)__builtin_amdgcn_perm(0, hi, 0x0c010c00), f.z, acc);
    return dot2((int32_t)__builtin_amdgcn_perm(0, hi, 0x0c030c02), f.w, acc);
}

// The wave's share of a 32x32 block's error (svt_av1_warp_affine_c, EbWarpedMotion.c:577-694, on the 8x8 blocks of the
// quadrant; svt_av1_calc_frame_error_c, EbEncWarpedMotion.c:162-171). The same sum in every lane.
__device__ uint32_t warp_block_share(const DevPlane &frm, const DevPlane &ref, const svtme_gm_warp &wm, int identity, WarpTables &t, int block) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int w = frm.width, h = frm.height, bw = (w + 31) >> 5;
    const int bx = block % bw, by = block / bw;
    const int j0 = bx * 32, i0 = by * 32, ww = min(32, w - j0), wh = min(32, h - i0);
    uint32_t  acc = 0;
    if (identity) {
        for (int p = threadIdx.x; p < 32 * 32; p += 256) {
            const int y = p >> 5, x = p & 31;
            if (x < ww && y < wh)
                acc += (uint32_t)t.lut[255 + frm.pix[(ptrdiff_t)(i0 + y) * frm.stride + j0 + x] - ref.pix[(ptrdiff_t)(i0 + y) * ref.stride + j0 + x]];
        }
        return wave_sum(acc);
    }
    int32_t  *tmp = t.tmp[wave];
    const int iq = i0 + 16 * (wave >> 1), jq = j0 + 16 * (wave & 1);
    for (int i = iq; i < min(iq + 16, i0 + wh); i += 8)
        for (int j = jq; j < min(jq + 16, j0 + ww); j += 8) {
            const int32_t src_x = j + 4, src_y = i + 4;
            const int32_t x4 = (int32_t)((uint32_t)wm.wmmat[2] * (uint32_t)src_x + (uint32_t)wm.wmmat[3] * (uint32_t)src_y + (uint32_t)wm.wmmat[0]);
            const int32_t y4 = (int32_t)((uint32_t)wm.wmmat[4] * (uint32_t)src_x + (uint32_t)wm.wmmat[5] * (uint32_t)src_y + (uint32_t)wm.wmmat[1]);
            const int32_t ix4 = x4 >> 16, iy4 = y4 >> 16;
            int32_t       sx4 = x4 & 0xffff, sy4 = y4 & 0xffff;
            sx4 += wm.alpha * -4 + wm.beta * -4;
            sy4 += wm.gamma * -4 + wm.delta * -4;
            sx4 &= ~63;
            sy4 &= ~63;
            // horizontal filter: rows iy4 - 7 .. iy4 + 7, 8 columns each; value s = row * 8 + column
            if (ix4 >= 7 && ix4 + 7 < ref.width && iy4 >= 7 && iy4 + 7 < ref.height) {
                // all 15 x 15 pixels lie inside the picture (the usual block): a lane's eight taps are eight consecutive
                // bytes, one load; pairs of them against pairs of coefficients
                const uint8_t *base = ref.pix + (ptrdiff_t)(iy4 - 7) * ref.stride + (ix4 - 7);
                for (int s = lane; s < 15 * 8; s += 64) {
                    const int          r = s >> 3, c = s & 7;
                    unsigned long long v;
                    __builtin_memcpy(&v, base + (uint32_t)(r * ref.stride + c), 8);
                    const int  sx = sx4 + wm.beta * (r - 3) + wm.alpha * c;
                    const int4 f  = *reinterpret_cast<const int4 *>(t.filter + 8 * min(max(((sx + 512) >> 10) + 64, 0), 192));
                    tmp[s]        = (taps8((uint32_t)v, (uint32_t)(v >> 32), f, 1 << 14) + 4) >> 3;
                }
            } else
                for (int s = lane; s < 15 * 8; s += 64) {
                    const int      k = (s >> 3) - 7, l = (s & 7) - 4;
                    const uint8_t *row = ref.pix + (ptrdiff_t)min(max(iy4 + k, 0), ref.height - 1) * ref.stride;
                    const int      sx = sx4 + wm.beta * (k + 4) + wm.alpha * (l + 4);
                    const int16_t *f  = t.filter + 8 * min(max(((sx + 512) >> 10) + 64, 0), 192); // (0 .. 192 for a valid shear)
                    int32_t        sum = 1 << 14;
#pragma unroll
                    for (int m = 0; m < 8; ++m) sum += row[min(max(ix4 + l - 3 + m, 0), ref.width - 1)] * f[m];
                    tmp[s] = (sum + 4) >> 3;
                }
            wave_sync();
            {
                const int k = (lane >> 3) - 4, l = (lane & 7) - 4;
                if (k < min(4, i0 + wh - i - 4) && l < min(4, j0 + ww - j - 4)) {
                    const int  sy = sy4 + wm.delta * (k + 4) + wm.gamma * (l + 4);
                    const int4 f  = *reinterpret_cast<const int4 *>(t.filter + 8 * min(max(((sy + 512) >> 10) + 64, 0), 192));
                    // the filtered values are positive and below 2^16 (an offset of 2^14 went into them, :630-640)
                    const int32_t *col = tmp + lane;
                    int32_t        sum = 1 << 19;
                    sum = dot2(col[0] | (col[8] << 16), f.x, sum);
                    sum = dot2(col[16] | (col[24] << 16), f.y, sum);
                    sum = dot2(col[32] | (col[40] << 16), f.z, sum);
                    sum = dot2(col[48] | (col[56] << 16), f.w, sum);
                    const int pix = min(max(((sum + 1024) >> 11) - 128 - 256, 0), 255);
                    acc += (uint32_t)t.lut[255 + frm.pix[(ptrdiff_t)(i + k + 4) * frm.stride + j + l + 4] - pix];
                }
            }
            wave_sync();
        }
    return wave_sum(acc);
}

// the four waves' shares -> the block's sum, in thread 0
__device__ __forceinline__ uint32_t warp_block_total(WarpTables &t, uint32_t share) {
    if ((threadIdx.x & 63) == 0) t.part[threadIdx.x >> 6] = share;
    __syncthreads();
    return t.part[0] + t.part[1] + t.part[2] + t.part[3];
}

__global__ __launch_bounds__(256) void gm_warp_error_kernel(DevPlane frm, DevPlane ref, svtme_gm_warp wm, int identity, const int16_t *filter,
                                                            const int32_t *lut, uint32_t *block_err) {
    __shared__ WarpTables t;
    stage_warp_tables(t, filter, lut, !identity);
    const uint32_t total = warp_block_total(t, warp_block_share(frm, ref, wm, identity, t, blockIdx.x));
    if (threadIdx.x == 0) block_err[blockIdx.x] = total;
}

// ------------------------------------------------------------------------------------------- parameter refinement
// svt_av1_refine_integerized_param (Source/Lib/Encoder/Codec/global_motion.c:135-263) as a state machine that one lane
// advances after every error evaluation: the loop's position (refinement step, parameter, which of "look left", "look right",
// "keep walking" it is in) lives in GmRefineState, so that a chain of identical kernel launches, each one evaluation,
// replays the loop without the host in between.
//
// What decides the loop is only whether an evaluation's error is below the best one so far: svt_av1_warp_error's early
// return (EbEncWarpedMotion.c:206-209) hands back a partial sum above best_error, which fails that comparison exactly as the
// whole sum would (the per-pixel errors are not negative), and a best error is only ever taken from a sum that ran to the end.
// So every evaluation here is the whole picture's sum.
enum { GM_PH_FIRST = 0, GM_PH_LEFT = 1, GM_PH_RIGHT = 2, GM_PH_WALK = 3 };

// svt_get_shear_params (Source/Lib/Common/Codec/EbWarpedMotion.c:1115-1144) on the state's matrix. The divisor table
// (div_lut, :301-322) in closed form: entry f = 2^22 / (256 + f), rounded to nearest.
__device__ __forceinline__ bool gm_shear(GmRefineState &st) {
    const int32_t *mat = st.mat;
    if (mat[2] <= 0) return false; // is_affine_valid, :359-362
    auto clamp16 = [](int v) { return min(max(v, -32768), 32767); };
    auto round64 = [](long long v, int n) { // ROUND_POWER_OF_TWO_SIGNED_64
        return v < 0 ? -((-v + ((1ll << n) >> 1)) >> n) : ((v + ((1ll << n) >> 1)) >> n);
    };
    auto reduce = [](int v) { // ROUND_POWER_OF_TWO_SIGNED(v, WARP_PARAM_REDUCE_BITS) * (1 << WARP_PARAM_REDUCE_BITS), into an int16_t
        return (int16_t)((v < 0 ? -((-v + 32) >> 6) : ((v + 32) >> 6)) * 64);
    };
    int alpha = clamp16(mat[2] - (1 << 16)), beta = clamp16(mat[3]);
    // resolve_divisor_32(mat[2]), :343-357
    const uint32_t d     = (uint32_t)mat[2];
    int            shift = 31 - __clz((int)d);
    const int32_t  e     = (int32_t)(d - (1u << shift));
    const int32_t  f     = shift > 8 ? (e + ((1 << (shift - 8)) >> 1)) >> (shift - 8) : e << (8 - shift);
    shift += 14;
    const int y = (int)(((1u << 22) + (uint32_t)(256 + f) / 2) / (uint32_t)(256 + f));
    long long v = ((long long)mat[4] * (1 << 16)) * y;
    const int gamma = clamp16((int)round64(v, shift));
    v               = ((long long)mat[3] * mat[4]) * y;
    const int delta = clamp16(mat[5] - (int)round64(v, shift) - (1 << 16));
    st.alpha = reduce(alpha);
    st.beta  = reduce(beta);
    st.gamma = reduce(gamma);
    st.delta = reduce(delta);
    // is_affine_shear_allowed, :364-370
    return !(4 * abs((int)st.alpha) + 7 * abs((int)st.beta) >= (1 << 16) || 4 * abs((int)st.gamma) + 4 * abs((int)st.delta) >= (1 << 16));
}

// add_param_offset, global_motion.c:91-113
__device__ __forceinline__ int32_t gm_add_param_offset(int index, int32_t value, int32_t offset) {
    const int scale = index < 2 ? 10 : 1; // GM_TRANS_PREC_DIFF, GM_ALPHA_PREC_DIFF
    const int centre = (index == 2 || index == 5) ? (1 << 16) : 0;
    value = (value - centre) >> scale;
    value += offset;
    value = min(max(value, -4096), 4096); // GM_TRANS_MAX = GM_ALPHA_MAX = 1 << 12
    value *= 1 << scale;
    return value + centre;
}

// force_wmtype, global_motion.c:115-133 (wmmat[6] and [7] are zero for every type; the state does not hold them)
__device__ __forceinline__ void gm_force_wmtype(GmRefineState &st) {
    switch (st.wmtype) {
    case 0: st.mat[0] = st.mat[1] = 0; [[fallthrough]];
    case 1: st.mat[2] = 1 << 16; st.mat[3] = 0; [[fallthrough]];
    case 2: st.mat[4] = -st.mat[3]; st.mat[5] = st.mat[2]; [[fallthrough]];
    default: break;
    }
}

// What svt_av1_warp_error does before it looks at the pictures (EbEncWarpedMotion.c:233-235, svt_warp_plane,
// EbWarpedMotion.c:701-704): the shear from the matrix as it stands - an invalid one makes the evaluation's result 1 - and
// only then, for a rotation-zoom model, the matrix's second row from its first.
__device__ __forceinline__ void gm_prepare_evaluation(GmRefineState &st) {
    st.skip = gm_shear(st) ? 0 : 1;
    if (!st.skip && st.wmtype == 2) {
        st.mat[5] = st.mat[2];
        st.mat[4] = -st.mat[3];
    }
}

__device__ __forceinline__ void gm_finish(GmRefineState &st) {
    gm_force_wmtype(st);
    // get_wmtype, global_motion.c:71-80
    const int32_t *m = st.mat;
    if (m[5] == (1 << 16) && !m[4] && m[2] == (1 << 16) && !m[3]) st.wmtype = (!m[1] && !m[0]) ? 0 : 1;
    else st.wmtype = (m[2] == m[5] && m[3] == -m[4]) ? 2 : 3;
    st.done = 1;
}

// wmmat[p] for a p that is not known before run time, with the matrix in registers
__device__ __forceinline__ int32_t gm_mat(const GmRefineState &st, int p) {
    const int32_t a = p & 1 ? st.mat[1] : st.mat[0], b = p & 1 ? st.mat[3] : st.mat[2], c = p & 1 ? st.mat[5] : st.mat[4];
    return p < 2 ? a : (p < 4 ? b : c);
}
__device__ __forceinline__ void gm_set_mat(GmRefineState &st, int p, int32_t v) {
    st.mat[0] = p == 0 ? v : st.mat[0];
    st.mat[1] = p == 1 ? v : st.mat[1];
    st.mat[2] = p == 2 ? v : st.mat[2];
    st.mat[3] = p == 3 ? v : st.mat[3];
    st.mat[4] = p == 4 ? v : st.mat[4];
    st.mat[5] = p == 5 ? v : st.mat[5];
}

// One evaluation's error has arrived: the loop moves on to its next evaluation, or ends. (One copy of each piece of the
// loop, reached by flags rather than by calls from every phase: the code runs once per launch, on one lane, from a cold
// instruction cache, so its size is its cost.)
__device__ __forceinline__ void gm_advance(GmRefineState &st, long long err) {
    ++st.evals;
    const bool better = err < st.best_error;
    bool       begin = false, next = false;
    switch (st.phase) {
    case GM_PH_FIRST: // :152-171
        st.best_error = min(err, st.best_error);
        st.step       = 1 << (st.n_refinements - 1);
        st.i = st.p = 0;
        begin       = true;
        break;
    case GM_PH_LEFT: // :195-199, then "look to the right" :201-226
        if (better) {
            st.best_error = err;
            st.best_param = gm_mat(st, st.p);
            st.step_dir   = -1;
        }
        gm_set_mat(st, st.p, gm_add_param_offset(st.p, st.curr_param, st.step));
        st.phase = GM_PH_RIGHT;
        break;
    case GM_PH_RIGHT: // :221-226, then the walk in the chosen direction :228-257
        if (better) {
            st.best_error = err;
            st.best_param = gm_mat(st, st.p);
            st.step_dir   = 1;
        }
        st.phase = GM_PH_WALK;
        next     = st.step_dir == 0;
        gm_set_mat(st, st.p, next ? st.best_param : gm_add_param_offset(st.p, st.best_param, st.step * st.step_dir));
        break;
    default: // GM_PH_WALK
        if (better) {
            st.best_error = err;
            st.best_param = gm_mat(st, st.p);
        }
        next = !better;
        gm_set_mat(st, st.p, next ? st.best_param : gm_add_param_offset(st.p, st.best_param, st.step * st.step_dir));
        break;
    }
    if (next) { // the next parameter, after the last one the next refinement step
        if (++st.p == st.n_params) {
            st.p = 0;
            ++st.i;
            st.step >>= 1;
        }
        begin = true;
    }
    if (begin) { // the loop's head for parameter p of refinement step i: "look to the left" (:178-199)
        if (st.i >= st.n_refinements || st.n_params == 0) {
            gm_finish(st);
            return;
        }
        st.step_dir   = 0;
        st.curr_param = st.best_param = gm_mat(st, st.p);
        gm_set_mat(st, st.p, gm_add_param_offset(st.p, st.curr_param, -st.step));
        st.phase = GM_PH_LEFT;
    }
    gm_prepare_evaluation(st);
}

// `init`: the caller's model (mat, shear), the type to force, n_refinements, best_error = the caller's best_frame_error.
__global__ void gm_refine_begin_kernel(GmRefineState *state, GmRefineState init) {
    if (threadIdx.x) return;
    GmRefineState st = init;
    st.n_params      = 2 * st.wmtype; // max_trans_model_params {0, 2, 4, 6}, global_motion.c:140
    st.phase    = GM_PH_FIRST;
    st.i = st.p = st.step = st.step_dir = st.curr_param = st.best_param = 0;
    st.done = st.evals = 0;
    gm_force_wmtype(st);
    gm_prepare_evaluation(st);
    *state = st;
}

// One evaluation of the chain: every workgroup adds up its 32x32 block with the state's model, gm_refine_step_kernel then
// adds the blocks up and advances the loop. (One kernel that did both, the workgroups counting themselves off with a
// device-wide atomic each, took 1.6 times as long at 1080p.) After the loop has ended the launches that are still queued
// return at once.
__global__ __launch_bounds__(256) void gm_refine_eval_kernel(DevPlane frm, DevPlane ref, const GmRefineState *state, const int16_t *filter,
                                                             const int32_t *lut, uint32_t *block_err) {
    __shared__ WarpTables t;
    const GmRefineState st = *state;
    if (st.done || st.skip) return;
    svtme_gm_warp wm;
    for (int k = 0; k < 6; ++k) wm.wmmat[k] = st.mat[k];
    wm.alpha = st.alpha;
    wm.beta  = st.beta;
    wm.gamma = st.gamma;
    wm.delta = st.delta;
    stage_warp_tables(t, filter, lut, true);
    const uint32_t total = warp_block_total(t, warp_block_share(frm, ref, wm, 0, t, blockIdx.x));
    if (threadIdx.x == 0) block_err[blockIdx.x] = total;
}

__global__ __launch_bounds__(256) void gm_refine_step_kernel(GmRefineState *state, const uint32_t *block_err, int n) {
    __shared__ unsigned long long part[4];
    GmRefineState      st = *state; // (asked for together with the sums below: one trip to memory, not two)
    unsigned long long sum = 0;
    for (int b0 = 0; b0 < n; b0 += 8 * 256) { // eight loads in flight per thread
        uint32_t v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int b = b0 + k * 256 + (int)threadIdx.x;
            v[k]        = b < n ? block_err[b] : 0;
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) sum += v[k];
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) sum += __shfl_xor(sum, m);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = sum;
    __syncthreads();
    if (threadIdx.x || st.done) return;
    gm_advance(st, st.skip ? 1 : (long long)(part[0] + part[1] + part[2] + part[3]));
    *state = st;
}

// The reference adds the blocks up in raster order and returns at the first block that takes the sum above best_error
// (EbEncWarpedMotion.c:206-209): result = that partial sum, or the total. One workgroup; thread t owns a run of
// consecutive blocks.
__global__ __launch_bounds__(1024) void gm_error_scan_kernel(const uint32_t *block_err, int n, long long best_error, long long *result) {
    __shared__ unsigned long long part[1024];
    __shared__ int                first;
    const int                     t = threadIdx.x, per = (n + 1023) / 1024, b0 = t * per, b1 = min(n, b0 + per);
    unsigned long long            mine = 0;
    for (int b = b0; b < b1; ++b) mine += block_err[b];
    part[t] = mine;
    if (t == 0) first = 0x7fffffff;
    __syncthreads();
    for (int s = 1; s < 1024; s <<= 1) { // inclusive scan of the runs' sums
        const unsigned long long v = t >= s ? part[t - s] : 0;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    unsigned long long run = part[t] - mine; // sum of everything before this thread's run
    int                hit = 0x7fffffff;
    for (int b = b0; b < b1; ++b) {
        run += block_err[b];
        if ((long long)run > best_error) {
            hit = b;
            break;
        }
    }
    if (hit != 0x7fffffff) atomicMin(&first, hit);
    __syncthreads();
    if (first == 0x7fffffff) {
        if (t == 1023) *result = (long long)part[1023];
    } else if (hit == first)
        *result = (long long)run;
}

} // namespace

void launch_gm_corners(const DevPlane &pl, uint8_t *score, uint8_t *keep, int *row_count, int max_points, int *out, hipStream_t s) {
    gm_score_kernel<<<dim3((pl.width + 63) / 64, (pl.height + 3) / 4), 256, 0, s>>>(pl, score);
    gm_nonmax_kernel<<<pl.height, 256, 0, s>>>(score, pl.width, pl.height, keep, row_count);
    gm_compact_kernel<<<pl.height, 256, 0, s>>>(keep, pl.width, pl.height, row_count, max_points, out);
}

void launch_gm_match(const DevPlane &frm, const DevPlane &ref, const int *frm_corners, const int *ref_corners, int n_frm_max, int *out,
                     hipStream_t s) {
    gm_match_kernel<<<n_frm_max, 64, 0, s>>>(frm, ref, frm_corners, ref_corners, out);
}

void launch_gm_error(const DevPlane &frm, const DevPlane &ref, const svtme_gm_warp *wm, const int16_t *filter, const int32_t *lut,
                     uint32_t *block_err, long long best_error, long long *result, hipStream_t s) {
    const int     n = ((frm.width + 31) / 32) * ((frm.height + 31) / 32);
    svtme_gm_warp w{};
    if (wm) w = *wm;
    gm_warp_error_kernel<<<n, 256, 0, s>>>(frm, ref, w, wm ? 0 : 1, filter, lut, block_err);
    gm_error_scan_kernel<<<1, 1024, 0, s>>>(block_err, n, best_error, result);
}

void launch_gm_refine_begin(GmRefineState *state, const GmRefineState &init, hipStream_t s) { gm_refine_begin_kernel<<<1, 64, 0, s>>>(state, init); }

void launch_gm_refine_evaluations(const DevPlane &frm, const DevPlane &ref, GmRefineState *state, const int16_t *filter, const int32_t *lut,
                                  uint32_t *block_err, int count, hipStream_t s) {
    const int n = ((frm.width + 31) / 32) * ((frm.height + 31) / 32);
    for (int i = 0; i < count; ++i) {
        gm_refine_eval_kernel<<<n, 256, 0, s>>>(frm, ref, state, filter, lut, block_err);
        gm_refine_step_kernel<<<1, 256, 0, s>>>(state, block_err, n);
    }
}

} // namespace svtme
