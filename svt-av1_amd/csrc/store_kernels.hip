// store_kernels.hip - builds the device picture store of one picture from its un-padded 8-bit luma:
// full-resolution copy, quarter and sixteenth planes, and the replicated borders of all three.
//
// Replaces on the device what picture analysis does on the host per input picture:
//   pad_input_pictures -> generate_padding      Source/Lib/Encoder/Codec/EbPictureAnalysisProcess.c:2754-2764,
//                                               Source/Lib/Common/Codec/EbMcp.c:112-164
//   downsample_filtering_input_picture          EbPictureAnalysisProcess.c:2688-2752 (downsample_2d_c :187-219)
// HBM-bound: W*H bytes read, (1 + 1/4 + 1/16)*W*H + borders written.
#include "svtme_device.h"

namespace svtme {

// One thread per 4x4 block of the full-resolution picture: copies the block, writes its 2x2 quarter pixels
// ((a+b+c+d+2)>>2 each) and its sixteenth pixel. With HME level 1 on, the sixteenth pixel is the rounded box of
// the four *quarter* pixels (two roundings, as the reference downsamples the quarter plane); otherwise it is the
// rounded mean of the centre 2x2 of the 4x4 block (the reference's step-4 call).
__global__ void __launch_bounds__(256) store_interior_kernel(const uint8_t *__restrict__ luma, int luma_stride,
                                                             DevPicture pic, int hme_level1) {
    const int bx  = blockIdx.x * blockDim.x + threadIdx.x;
    const int by  = blockIdx.y;
    const int nbx = pic.lvl[2].width >> 2;
    if (bx >= nbx) return;
    uint32_t row[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) row[r] = *reinterpret_cast<const uint32_t *>(luma + (size_t)(by * 4 + r) * luma_stride + bx * 4);
    const DevPlane &f = pic.lvl[2];
#pragma unroll
    for (int r = 0; r < 4; ++r) *reinterpret_cast<uint32_t *>(f.pix + (size_t)(by * 4 + r) * f.stride + bx * 4) = row[r];
    uint32_t q[2][2];
#pragma unroll
    for (int qy = 0; qy < 2; ++qy)
#pragma unroll
        for (int qx = 0; qx < 2; ++qx) {
            const uint32_t a = row[2 * qy], b = row[2 * qy + 1];
            const int      s = 16 * qx;
            q[qy][qx] = (((a >> s) & 255u) + ((a >> (s + 8)) & 255u) + ((b >> s) & 255u) + ((b >> (s + 8)) & 255u) + 2u) >> 2;
        }
    const DevPlane &qp = pic.lvl[1];
    if (2 * by + 1 < qp.height + 1 && 2 * bx + 1 < qp.width + 1) {
#pragma unroll
        for (int qy = 0; qy < 2; ++qy)
            *reinterpret_cast<uint16_t *>(qp.pix + (size_t)(by * 2 + qy) * qp.stride + bx * 2) = (uint16_t)(q[qy][0] | (q[qy][1] << 8));
    }
    uint32_t s16;
    if (hme_level1) s16 = (q[0][0] + q[0][1] + q[1][0] + q[1][1] + 2u) >> 2;
    else s16 = (((row[1] >> 8) & 255u) + ((row[1] >> 16) & 255u) + ((row[2] >> 8) & 255u) + ((row[2] >> 16) & 255u) + 2u) >> 2;
    const DevPlane &sp = pic.lvl[0];
    sp.pix[(size_t)by * sp.stride + bx] = (uint8_t)s16;
}

// Edge replication of all three planes: every border byte takes the interior pixel at the clamped coordinate,
// which is what generate_padding's "rows sideways first, then whole rows up and down" produces.
__global__ void __launch_bounds__(256) store_border_kernel(DevPicture pic) {
    const DevPlane &p   = pic.lvl[blockIdx.y];
    const int       pw  = p.width + 2 * p.pad;
    const int       side = p.height * 2 * p.pad; // left + right strips of the picture rows
    const int       total = side + 2 * p.pad * pw; // + full-width top and bottom strips
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        int x, y;
        if (i < side) {
            y = i / (2 * p.pad);
            const int c = i - y * 2 * p.pad;
            x = c < p.pad ? c - p.pad : p.width + (c - p.pad);
        } else {
            const int j = i - side;
            const int r = j / pw;
            x = j - r * pw - p.pad;
            y = r < p.pad ? r - p.pad : p.height + (r - p.pad);
        }
        const int sx = min(max(x, 0), p.width - 1), sy = min(max(y, 0), p.height - 1);
        p.pix[(ptrdiff_t)y * p.stride + x] = p.pix[(ptrdiff_t)sy * p.stride + sx];
    }
}

void launch_store_build(const uint8_t *luma, int luma_stride, const DevPicture &pic, int hme_level1, hipStream_t s) {
    const int nbx = pic.lvl[2].width >> 2, nby = pic.lvl[2].height >> 2;
    dim3      grid((nbx + 255) / 256, nby);
    hipLaunchKernelGGL(store_interior_kernel, grid, dim3(256), 0, s, luma, luma_stride, pic, hme_level1);
    const int border_full = pic.lvl[2].height * 2 * pic.lvl[2].pad + 2 * pic.lvl[2].pad * (pic.lvl[2].width + 2 * pic.lvl[2].pad);
    dim3      bgrid((border_full + 255) / 256, 3);
    hipLaunchKernelGGL(store_border_kernel, bgrid, dim3(256), 0, s, pic);
}

// ---- the same for several pictures of one size in one launch each (a mini-GOP's sources become final together,
// EbPictureDecisionProcess.c:6815-6827): the picture is the grid's third dimension, the pictures' descriptors travel as
// kernel arguments. The interior kernel handles a 16x4 block per thread (16-byte loads and stores).
__global__ void __launch_bounds__(256) store_interior_many_kernel(StoreBatch b, int hme_level1) {
    const DevPicture &pic = b.pic[blockIdx.z];
    const uint8_t    *luma = b.luma[blockIdx.z];
    const int         luma_stride = b.stride[blockIdx.z];
    const int bx  = blockIdx.x * blockDim.x + threadIdx.x; // 16-pixel column group (the width is a multiple of 8)
    const int by  = blockIdx.y;
    const int w   = pic.lvl[2].width;
    if (bx * 16 >= w) return;
    const bool half = bx * 16 + 8 >= w; // the last group of a width that is not a multiple of 16: 8 pixels
    uint32_t   row[4][4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const uint8_t *src = luma + (size_t)(by * 4 + r) * luma_stride + bx * 16;
        if (half) {
            uint2 v;
            __builtin_memcpy(&v, src, 8);
            row[r][0] = v.x, row[r][1] = v.y, row[r][2] = row[r][3] = 0;
        } else {
            uint4 v;
            __builtin_memcpy(&v, src, 16);
            row[r][0] = v.x, row[r][1] = v.y, row[r][2] = v.z, row[r][3] = v.w;
        }
    }
    const DevPlane &f = pic.lvl[2];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        uint8_t *dst = f.pix + (size_t)(by * 4 + r) * f.stride + bx * 16; // (8-byte aligned: pixel (0,0) is, the stride too)
        *reinterpret_cast<uint2 *>(dst) = make_uint2(row[r][0], row[r][1]);
        if (!half) *reinterpret_cast<uint2 *>(dst + 8) = make_uint2(row[r][2], row[r][3]);
    }
    // quarter pixels: 8 per row pair, 2 row pairs; sixteenth pixels: 4
    uint32_t q[2][8];
#pragma unroll
    for (int qy = 0; qy < 2; ++qy)
#pragma unroll
        for (int qx = 0; qx < 8; ++qx) {
            const uint32_t a = row[2 * qy][qx >> 1], c = row[2 * qy + 1][qx >> 1];
            const int      s = 16 * (qx & 1);
            q[qy][qx] = (((a >> s) & 255u) + ((a >> (s + 8)) & 255u) + ((c >> s) & 255u) + ((c >> (s + 8)) & 255u) + 2u) >> 2;
        }
    const DevPlane &qp = pic.lvl[1];
#pragma unroll
    for (int qy = 0; qy < 2; ++qy) {
        uint8_t       *dst = qp.pix + (size_t)(by * 2 + qy) * qp.stride + bx * 8;
        const uint32_t lo = q[qy][0] | (q[qy][1] << 8) | (q[qy][2] << 16) | (q[qy][3] << 24);
        const uint32_t hi = q[qy][4] | (q[qy][5] << 8) | (q[qy][6] << 16) | (q[qy][7] << 24);
        *reinterpret_cast<uint32_t *>(dst) = lo;
        if (!half) *reinterpret_cast<uint32_t *>(dst + 4) = hi;
    }
    uint32_t s16 = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        uint32_t v;
        if (hme_level1) v = (q[0][2 * k] + q[0][2 * k + 1] + q[1][2 * k] + q[1][2 * k + 1] + 2u) >> 2;
        else v = (((row[1][k] >> 8) & 255u) + ((row[1][k] >> 16) & 255u) + ((row[2][k] >> 8) & 255u) + ((row[2][k] >> 16) & 255u) + 2u) >> 2;
        s16 |= v << (8 * k);
    }
    const DevPlane &sp  = pic.lvl[0];
    uint8_t        *dst = sp.pix + (size_t)by * sp.stride + bx * 4;
    if (half) *reinterpret_cast<uint16_t *>(dst) = (uint16_t)s16;
    else *reinterpret_cast<uint32_t *>(dst) = s16;
}

__global__ void __launch_bounds__(256) store_border_many_kernel(StoreBatch b) {
    const DevPlane &p   = b.pic[blockIdx.z].lvl[blockIdx.y];
    const int       pw  = p.width + 2 * p.pad;
    const int       side = p.height * 2 * p.pad; // left + right strips of the picture rows
    const int       total = side + 2 * p.pad * pw; // + full-width top and bottom strips
    // (i / d as (i * ceil(2^40 / d)) >> 40: exact for i * d < 2^40; the host computed the two multipliers of each plane)
    const unsigned long long m_side = b.m_side[blockIdx.y], m_pw = b.m_pw[blockIdx.y];
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        int x, y;
        if (i < side) {
            y = (int)(((unsigned long long)(uint32_t)i * m_side) >> 40);
            const int c = i - y * 2 * p.pad;
            x = c < p.pad ? c - p.pad : p.width + (c - p.pad);
        } else {
            const int j = i - side;
            const int r = (int)(((unsigned long long)(uint32_t)j * m_pw) >> 40);
            x = j - r * pw - p.pad;
            y = r < p.pad ? r - p.pad : p.height + (r - p.pad);
        }
        const int sx = min(max(x, 0), p.width - 1), sy = min(max(y, 0), p.height - 1);
        p.pix[(ptrdiff_t)y * p.stride + x] = p.pix[(ptrdiff_t)sy * p.stride + sx];
    }
}

void launch_store_build_many(StoreBatch &b, int n, int hme_level1, hipStream_t s) {
    const DevPicture &pic = b.pic[0]; // one size for all
    for (int l = 0; l < 3; ++l) {
        const unsigned long long d_side = 2ull * (unsigned long long)pic.lvl[l].pad, d_pw = (unsigned long long)(pic.lvl[l].width + 2 * pic.lvl[l].pad);
        b.m_side[l] = ((1ull << 40) + d_side - 1) / d_side;
        b.m_pw[l]   = ((1ull << 40) + d_pw - 1) / d_pw;
    }
    const int         ngx = (pic.lvl[2].width + 15) >> 4, nby = pic.lvl[2].height >> 2;
    hipLaunchKernelGGL(store_interior_many_kernel, dim3((ngx + 255) / 256, nby, n), dim3(256), 0, s, b, hme_level1);
    const int border_full = pic.lvl[2].height * 2 * pic.lvl[2].pad + 2 * pic.lvl[2].pad * (pic.lvl[2].width + 2 * pic.lvl[2].pad);
    hipLaunchKernelGGL(store_border_many_kernel, dim3((border_full + 255) / 256, 3, n), dim3(256), 0, s, b);
}

} // namespace svtme
