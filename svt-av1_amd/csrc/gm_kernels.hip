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
// One wave per 32x32 block of the error measure (WARP_ERROR_BLOCK, EbEncWarpedMotion.c:20): its 8x8 blocks one after the
// other, 15 x 8 horizontally filtered values through LDS, then a lane per output pixel; block_err[block] = the block's sum
// of error_measure(frame - warped). identity != 0: no warp, the frame error against the reference picture itself.
__global__ __launch_bounds__(64) void gm_warp_error_kernel(DevPlane frm, DevPlane ref, svtme_gm_warp wm, int identity, const int16_t *filter,
                                                           const int32_t *lut, uint32_t *block_err) {
    __shared__ int16_t s_filter[193 * 8];
    __shared__ int32_t s_lut[512];
    __shared__ int32_t tmp[15 * 8];
    const int          lane = threadIdx.x;
    const int          w = frm.width, h = frm.height, bw = (w + 31) >> 5;
    const int          bx = blockIdx.x % bw, by = blockIdx.x / bw;
    const int          j0 = bx * 32, i0 = by * 32, ww = min(32, w - j0), wh = min(32, h - i0);
    for (int k = lane; k < 512; k += 64) s_lut[k] = lut[k];
    if (!identity)
        for (int k = lane; k < 193 * 8 / 2; k += 64) reinterpret_cast<uint32_t *>(s_filter)[k] = reinterpret_cast<const uint32_t *>(filter)[k];
    wave_sync();
    uint32_t acc = 0;
    if (identity) {
        for (int p = lane; p < 32 * 32; p += 64) {
            const int y = p >> 5, x = p & 31;
            if (x < ww && y < wh)
                acc += (uint32_t)s_lut[255 + frm.pix[(ptrdiff_t)(i0 + y) * frm.stride + j0 + x] - ref.pix[(ptrdiff_t)(i0 + y) * ref.stride + j0 + x]];
        }
    } else {
        for (int i = i0; i < i0 + wh; i += 8)
            for (int j = j0; j < j0 + ww; j += 8) {
                const int32_t src_x = j + 4, src_y = i + 4;
                const int32_t x4 = (int32_t)((uint32_t)wm.wmmat[2] * (uint32_t)src_x + (uint32_t)wm.wmmat[3] * (uint32_t)src_y + (uint32_t)wm.wmmat[0]);
                const int32_t y4 = (int32_t)((uint32_t)wm.wmmat[4] * (uint32_t)src_x + (uint32_t)wm.wmmat[5] * (uint32_t)src_y + (uint32_t)wm.wmmat[1]);
                const int32_t ix4 = x4 >> 16, iy4 = y4 >> 16;
                int32_t       sx4 = x4 & 0xffff, sy4 = y4 & 0xffff;
                sx4 += wm.alpha * -4 + wm.beta * -4;
                sy4 += wm.gamma * -4 + wm.delta * -4;
                sx4 &= ~63;
                sy4 &= ~63;
                // horizontal filter: rows iy4 - 7 .. iy4 + 7, 8 columns each
                for (int t = lane; t < 15 * 8; t += 64) {
                    const int      k = (t >> 3) - 7, l = (t & 7) - 4;
                    const uint8_t *row = ref.pix + (ptrdiff_t)min(max(iy4 + k, 0), ref.height - 1) * ref.stride;
                    const int      sx = sx4 + wm.beta * (k + 4) + wm.alpha * (l + 4);
                    const int16_t *f  = s_filter + 8 * min(max(((sx + 512) >> 10) + 64, 0), 192); // (0 .. 192 for a valid shear)
                    int32_t        sum = 1 << 14;
#pragma unroll
                    for (int m = 0; m < 8; ++m) sum += row[min(max(ix4 + l - 3 + m, 0), ref.width - 1)] * f[m];
                    tmp[t] = (sum + 4) >> 3;
                }
                wave_sync();
                {
                    const int k = (lane >> 3) - 4, l = (lane & 7) - 4;
                    if (k < min(4, i0 + wh - i - 4) && l < min(4, j0 + ww - j - 4)) {
                        const int      sy = sy4 + wm.delta * (k + 4) + wm.gamma * (l + 4);
                        const int16_t *f  = s_filter + 8 * min(max(((sy + 512) >> 10) + 64, 0), 192);
                        int32_t        sum = 1 << 19;
#pragma unroll
                        for (int m = 0; m < 8; ++m) sum += tmp[(k + m + 4) * 8 + l + 4] * f[m];
                        const int pix = min(max(((sum + 1024) >> 11) - 128 - 256, 0), 255);
                        acc += (uint32_t)s_lut[255 + frm.pix[(ptrdiff_t)(i + k + 4) * frm.stride + j + l + 4] - pix];
                    }
                }
                wave_sync();
            }
    }
    acc = wave_sum(acc);
    if (lane == 0) block_err[blockIdx.x] = acc;
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
    gm_warp_error_kernel<<<n, 64, 0, s>>>(frm, ref, w, wm ? 0 : 1, filter, lut, block_err);
    gm_error_scan_kernel<<<1, 1024, 0, s>>>(block_err, n, best_error, result);
}

} // namespace svtme
