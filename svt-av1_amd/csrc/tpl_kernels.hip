// tpl_kernels.hip - the source-based mode search of the TPL dispenser (tpl_mc_flow_dispenser_sb_generic,
// Source/Lib/Encoder/Codec/EbSourceBasedOperationsProcess.c:632-933): per block of the dispenser's grid the DC intra
// prediction from the source picture's own neighbours and every single-reference ME candidate at its clamped full-pel
// vector, costed by the sum of absolute differences or by the sum of the absolute coefficients of the residual's 16x16
// forward DCT, and the quantisation error of the candidate that wins (:935-982). One wave per block; a lane holds the
// block's pixels 4 at a time (dword idx = lane + 64 k of the block's rows), so a 16x16 block is one dword per lane and a
// 64x64 block 16. A 16x16 transform takes 16 lanes - a lane per column, then a lane per row, through LDS - so the costs of
// up to four predictions of a block (four intra modes, four candidates, or three candidates and the intra mode the rounds of
// four left over) are taken together, one in each quarter of the wave.
#include <algorithm>
#include "search_common.h"

namespace svtme {

namespace {

constexpr int kTplPad = 32; // TPL_PADX, TPL_PADY: EbEncodeContext.h:43-44

// tpl_blk_idx_tab (:514-516): block i's index in the ME results
__constant__ const uint8_t kTplMeIndex[SVTME_TPL_BLOCKS] = {0, 1, 2, 3, 4, 5, 6, 9, 10, 7, 8, 11, 12, 13, 14, 17, 18, 15, 16, 19, 20};


// ---- the forward DCT of svt_av1_wht_fwd_txfm (EbTransforms.c:3815-3830 -> av1_tranform_two_d_core_c, :2301-2370: input
// << 2, an H-point transform of every column with 13-bit cosines, rounded >> 2 (>> 1 for 16x4), a W-point transform of every
// row, x sqrt(2) where W : H is 2 : 1; shifts and cosine bits per size: EbTransforms.h:26-57). The 1-D transforms are the
// butterfly networks of svt_av1_fdct4_new .. svt_av1_fdct32_new (:183-677), written part by part: the N-point transform is
// the N/2-point transform of the mirrored sums (even outputs) plus an odd half of the mirrored differences; a network's
// only roundings are its two-term rotations (a ca + b cb + 2^(bit-1)) >> bit (half_btf, EbInvTransforms.h:289-300), sums
// and differences are exact. With 8-bit residuals every product stays below 2^30, so 32-bit arithmetic gives what the
// reference's 64-bit sums give. c[k / 2] = cos(k pi / 128) with BIT fractional bits, k even.
template <int BIT>
__device__ __forceinline__ int32_t rot(int32_t a, int32_t ca, int32_t b, int32_t cb) {
    // (24-bit multiplications, a quarter of the cost of 32-bit ones: the terms stay below 2^18 with 8-bit residuals - at most
    // 32 inputs of 11 bits added up, rotations grow a value by less than sqrt(2) - and the cosines below 2^14)
    return (__mul24(a, ca) + __mul24(b, cb) + (1 << (BIT - 1))) >> BIT;
}
#define TPL_C(k) c[(k) >> 1]

typedef const int32_t __attribute__((address_space(4))) *CosTable; // a table of the picture's record (see ConstTplParams)

template <int BIT>
__device__ __forceinline__ void fdct_n(const int32_t (&x)[4], int32_t (&X)[4], CosTable c) {
    const int32_t a0 = x[0] + x[3], a1 = x[1] + x[2], a2 = x[1] - x[2], a3 = x[0] - x[3];
    X[0] = rot<BIT>(a0, TPL_C(32), a1, TPL_C(32));
    X[2] = rot<BIT>(a0, TPL_C(32), a1, -TPL_C(32));
    X[1] = rot<BIT>(a2, TPL_C(48), a3, TPL_C(16));
    X[3] = rot<BIT>(a3, TPL_C(48), a2, -TPL_C(16));
}

template <int BIT>
__device__ __forceinline__ void fdct_n(const int32_t (&x)[8], int32_t (&X)[8], CosTable c) {
    int32_t p[4], q[4], E[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        p[i] = x[i] + x[7 - i];
        q[i] = x[3 - i] - x[4 + i];
    }
    fdct_n<BIT>(p, E, c);
#pragma unroll
    for (int i = 0; i < 4; ++i) X[2 * i] = E[i];
    const int32_t m1 = rot<BIT>(q[2], TPL_C(32), q[1], -TPL_C(32)), m2 = rot<BIT>(q[2], TPL_C(32), q[1], TPL_C(32));
    const int32_t e0 = q[0] + m1, e1 = q[0] - m1, e2 = q[3] - m2, e3 = q[3] + m2;
    X[1] = rot<BIT>(e0, TPL_C(56), e3, TPL_C(8));
    X[5] = rot<BIT>(e1, TPL_C(24), e2, TPL_C(40));
    X[3] = rot<BIT>(e2, TPL_C(24), e1, -TPL_C(40));
    X[7] = rot<BIT>(e3, TPL_C(56), e0, -TPL_C(8));
}

template <int BIT>
__device__ __forceinline__ void fdct_n(const int32_t (&x)[16], int32_t (&X)[16], CosTable c) {
    int32_t s[8], d[8], E[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        s[i] = x[i] + x[15 - i];
        d[i] = x[7 - i] - x[8 + i];
    }
    fdct_n<BIT>(s, E, c);
#pragma unroll
    for (int i = 0; i < 8; ++i) X[2 * i] = E[i];
    const int32_t g2 = rot<BIT>(d[5], TPL_C(32), d[2], -TPL_C(32)), g5 = rot<BIT>(d[5], TPL_C(32), d[2], TPL_C(32));
    const int32_t g3 = rot<BIT>(d[4], TPL_C(32), d[3], -TPL_C(32)), g4 = rot<BIT>(d[4], TPL_C(32), d[3], TPL_C(32));
    const int32_t h0 = d[0] + g3, h3 = d[0] - g3, h1 = d[1] + g2, h2 = d[1] - g2;
    const int32_t h4 = d[7] - g4, h7 = d[7] + g4, h5 = d[6] - g5, h6 = d[6] + g5;
    const int32_t k1 = rot<BIT>(h6, TPL_C(48), h1, -TPL_C(16)), k6 = rot<BIT>(h6, TPL_C(16), h1, TPL_C(48));
    const int32_t k2 = rot<BIT>(h2, -TPL_C(48), h5, -TPL_C(16)), k5 = rot<BIT>(h5, TPL_C(48), h2, -TPL_C(16));
    const int32_t n0 = h0 + k1, n1 = h0 - k1, n2 = h3 - k2, n3 = h3 + k2;
    const int32_t n4 = h4 + k5, n5 = h4 - k5, n6 = h7 - k6, n7 = h7 + k6;
    X[1]  = rot<BIT>(n0, TPL_C(60), n7, TPL_C(4));
    X[9]  = rot<BIT>(n1, TPL_C(28), n6, TPL_C(36));
    X[5]  = rot<BIT>(n2, TPL_C(44), n5, TPL_C(20));
    X[13] = rot<BIT>(n3, TPL_C(12), n4, TPL_C(52));
    X[3]  = rot<BIT>(n4, TPL_C(12), n3, -TPL_C(52));
    X[11] = rot<BIT>(n5, TPL_C(44), n2, -TPL_C(20));
    X[7]  = rot<BIT>(n6, TPL_C(28), n1, -TPL_C(36));
    X[15] = rot<BIT>(n7, TPL_C(60), n0, -TPL_C(4));
}

template <int BIT>
__device__ __forceinline__ void fdct_n(const int32_t (&x)[32], int32_t (&X)[32], CosTable c) {
    int32_t s[16], o[16], t[16], E[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        s[i] = x[i] + x[31 - i];
        o[i] = x[15 - i] - x[16 + i];
    }
    fdct_n<BIT>(s, E, c);
#pragma unroll
    for (int i = 0; i < 16; ++i) X[2 * i] = E[i];
    // the odd half: rotations by pi/4 of the inner eight, sums, rotations by multiples of pi/8, sums, of pi/16, sums, and
    // the sixteen output rotations
#pragma unroll
    for (int j = 4; j < 8; ++j) {
        const int32_t lo = o[j], hi = o[15 - j];
        o[j]      = rot<BIT>(hi, TPL_C(32), lo, -TPL_C(32));
        o[15 - j] = rot<BIT>(hi, TPL_C(32), lo, TPL_C(32));
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        t[j]      = o[j] + o[7 - j];
        t[7 - j]  = o[j] - o[7 - j];
        t[8 + j]  = o[15 - j] - o[8 + j];
        t[15 - j] = o[15 - j] + o[8 + j];
    }
    o[0] = t[0], o[1] = t[1], o[6] = t[6], o[7] = t[7], o[8] = t[8], o[9] = t[9], o[14] = t[14], o[15] = t[15];
    o[2]  = rot<BIT>(t[13], TPL_C(48), t[2], -TPL_C(16));
    o[3]  = rot<BIT>(t[12], TPL_C(48), t[3], -TPL_C(16));
    o[4]  = rot<BIT>(t[4], -TPL_C(48), t[11], -TPL_C(16));
    o[5]  = rot<BIT>(t[5], -TPL_C(48), t[10], -TPL_C(16));
    o[10] = rot<BIT>(t[10], TPL_C(48), t[5], -TPL_C(16));
    o[11] = rot<BIT>(t[11], TPL_C(48), t[4], -TPL_C(16));
    o[12] = rot<BIT>(t[12], TPL_C(16), t[3], TPL_C(48));
    o[13] = rot<BIT>(t[13], TPL_C(16), t[2], TPL_C(48));
    t[0] = o[0] + o[3], t[1] = o[1] + o[2], t[2] = o[1] - o[2], t[3] = o[0] - o[3];
    t[4] = o[7] - o[4], t[5] = o[6] - o[5], t[6] = o[6] + o[5], t[7] = o[7] + o[4];
    t[8] = o[8] + o[11], t[9] = o[9] + o[10], t[10] = o[9] - o[10], t[11] = o[8] - o[11];
    t[12] = o[15] - o[12], t[13] = o[14] - o[13], t[14] = o[14] + o[13], t[15] = o[15] + o[12];
    o[0] = t[0], o[3] = t[3], o[4] = t[4], o[7] = t[7], o[8] = t[8], o[11] = t[11], o[12] = t[12], o[15] = t[15];
    o[1]  = rot<BIT>(t[14], TPL_C(56), t[1], -TPL_C(8));
    o[2]  = rot<BIT>(t[2], -TPL_C(56), t[13], -TPL_C(8));
    o[5]  = rot<BIT>(t[10], TPL_C(24), t[5], -TPL_C(40));
    o[6]  = rot<BIT>(t[6], -TPL_C(24), t[9], -TPL_C(40));
    o[9]  = rot<BIT>(t[9], TPL_C(24), t[6], -TPL_C(40));
    o[10] = rot<BIT>(t[10], TPL_C(40), t[5], TPL_C(24));
    o[13] = rot<BIT>(t[13], TPL_C(56), t[2], -TPL_C(8));
    o[14] = rot<BIT>(t[14], TPL_C(8), t[1], TPL_C(56));
    t[0] = o[0] + o[1], t[1] = o[0] - o[1], t[2] = o[3] - o[2], t[3] = o[3] + o[2];
    t[4] = o[4] + o[5], t[5] = o[4] - o[5], t[6] = o[7] - o[6], t[7] = o[7] + o[6];
    t[8] = o[8] + o[9], t[9] = o[8] - o[9], t[10] = o[11] - o[10], t[11] = o[11] + o[10];
    t[12] = o[12] + o[13], t[13] = o[12] - o[13], t[14] = o[15] - o[14], t[15] = o[15] + o[14];
    // rotation j pairs t[j] with t[15 - j]; the outputs are in bit-reversed order
#define TPL_OUT(j, angle, where)                                                          \
    X[where]      = rot<BIT>(t[j], TPL_C(angle), t[15 - j], TPL_C(64 - angle));            \
    X[32 - where] = rot<BIT>(t[15 - j], TPL_C(angle), t[j], -TPL_C(64 - angle))
    TPL_OUT(0, 62, 1);
    TPL_OUT(1, 30, 17);
    TPL_OUT(2, 46, 9);
    TPL_OUT(3, 14, 25);
    TPL_OUT(4, 54, 5);
    TPL_OUT(5, 22, 21);
    TPL_OUT(6, 38, 13);
    TPL_OUT(7, 6, 29);
#undef TPL_OUT
}
#undef TPL_C

// a 32x8 or 16x16 residual, or four 16x16 ones, with a pitch one dword over the width (no bank conflicts)
constexpr int kTplLdsInts = 16 * 33, kTplLdsIntsX4 = 4 * 16 * 17;

// The residual src - prd of a block of side W whose transform takes every (W / H)-th row (a lane holds N dwords of 4 pixels:
// dword lane + 64 k of the block's rows) -> sum of the absolute coefficients of its W x H forward DCT (svt_aom_satd,
// common_dsp_rtcd.c:47-54) and, when `quant`, get_quantize_error's recon_error (:167-193: svt_av1_quantize_fp_c without
// quantisation matrix, EbFullLoop.c:318-405, block error >> 2, at least 1), in every lane. pf_shape N2 / N4 keep the top-left
// half / quarter of the coefficients in each direction (av1_tranform_two_d_core_N2_c / _N4_c). A lane per column, then a
// lane per row, through LDS.
// The picture's record lies in device memory and is read through the constant address space: lanes index its arrays
// (reference planes, quantiser) with run-time values, which a kernel argument of this size would pay for with a private copy.
typedef const TplParams __attribute__((address_space(4))) ConstTplParams;

// Position of coefficient (r, c) in default_scan_16x16 (EbCoefficients.h:484-502): the anti-diagonals in turn, the odd ones
// from the top row down, the even ones from the left column up.
__device__ __forceinline__ int tpl_scan_pos16(int r, int c) {
    const int d = r + c, before = d <= 15 ? d * (d + 1) / 2 : 256 - (31 - d) * (32 - d) / 2;
    const int r_min = d > 15 ? d - 15 : 0, r_max = d < 15 ? d : 15;
    return before + ((d & 1) ? r - r_min : r_max - r);
}

// want_rate (16x16 only): also rate_estimator's value (:195-209) for the quantised coefficients: (eob + 1 + the sum of
// floor(log2(level + 1)) over the levels in front of eob) << AV1_PROB_COST_SHIFT, eob = the scan position after the last level
// that is not zero.
template <int W, int H, int N>
__device__ __forceinline__ void tpl_transform(ConstTplParams &p, const uint32_t (&src)[N], const uint32_t (&prd)[N], int32_t *lds, bool quant,
                                              long long &satd, long long &recon_error, bool want_rate = false, long long *rate = nullptr) {
    constexpr int PITCH = W + 1, DW = W / 4, STEP = W / H, SHIFT1 = (W == 16 && H == 4) ? 1 : 2;
    constexpr int BIT_ROW = ((W == 16 && H == 16) || W == 32) ? 12 : 13;
    const int     lane = threadIdx.x & 63;
    wave_sync(); // the previous use of the buffer is over
#pragma unroll
    for (int k = 0; k < N; ++k) {
        const int idx = lane + 64 * k, r = idx / DW, d = idx % DW;
        if (r % STEP == 0) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                lds[(r / STEP) * PITCH + 4 * d + j] = (int32_t)((src[k] >> (8 * j)) & 0xff) - (int32_t)((prd[k] >> (8 * j)) & 0xff);
        }
    }
    wave_sync();
    {
        int32_t in[H], out[H];
        if (lane < W) {
#pragma unroll
            for (int i = 0; i < H; ++i) in[i] = lds[i * PITCH + lane] * 4;
            fdct_n<13>(in, out, p.cos13);
        }
        wave_sync();
        if (lane < W) {
#pragma unroll
            for (int i = 0; i < H; ++i) lds[i * PITCH + lane] = (out[i] + (1 << (SHIFT1 - 1))) >> SHIFT1;
        }
    }
    wave_sync();
    long long a_sum = 0, e_sum = 0;
    int       eob = 0, logs = 0;
    const int kw = W >> p.pf_shape, kh = H >> p.pf_shape;
    if (lane < kh) {
        int32_t in[W], out[W];
#pragma unroll
        for (int i = 0; i < W; ++i) in[i] = lds[lane * PITCH + i];
        fdct_n<BIT_ROW>(in, out, BIT_ROW == 12 ? p.cos12 : p.cos13);
#pragma unroll
        for (int i = 0; i < W; ++i) {
            int32_t v = i < kw ? out[i] : 0;
            if (W == 2 * H) v = (int32_t)(((long long)v * 5793 + 2048) >> 12); // NewSqrt2, 12 fractional bits
            const int32_t a = v < 0 ? -v : v;
            a_sum += a;
            if (quant) {
                const int ac = (lane | i) != 0;
                int32_t   dq = 0;
                if ((a << 1) >= p.dequant[ac]) {
                    const int32_t t     = min(a + p.round_fp[ac], 32767);
                    const int32_t level = (int32_t)(((long long)t * p.quant_fp[ac]) >> 16);
                    dq                  = level * p.dequant[ac];
                    if constexpr (W == 16 && H == 16) {
                        if (want_rate && level) {
                            eob = max(eob, tpl_scan_pos16(lane, i) + 1);
                            logs += 31 - __clz(level + 1);
                        }
                    }
                }
                const long long diff = (long long)a - dq; // |v - sign(v) dq|: same square
                e_sum += diff * diff;
            }
        }
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        a_sum += __shfl_xor(a_sum, m);
        e_sum += __shfl_xor(e_sum, m);
    }
    satd = a_sum;
    e_sum >>= 2;
    recon_error = e_sum < 1 ? 1 : e_sum;
    if constexpr (W == 16 && H == 16) {
        if (want_rate) {
#pragma unroll
            for (int m = 32; m >= 1; m >>= 1) {
                eob = max(eob, __shfl_xor(eob, m));
                logs += __shfl_xor(logs, m);
            }
            *rate = (long long)(eob + 1 + logs) << 9;
        }
    }
}

// tpl_transform<16, 16, 1> for up to four predictions of one 16x16 block at a time (quarter g of the wave takes prd[g] when
// bit g of `mask` is set): the costs, errors and rates of all of them, in every lane.
__device__ __forceinline__ void tpl_transform_x4(ConstTplParams &p, uint32_t src, const uint32_t (&prd)[4], uint32_t mask, int32_t *lds, bool quant,
                                                 bool want_rate, long long (&satd)[4], long long (&recon_error)[4], long long (&rate)[4]) {
    constexpr int PITCH = 17, AREA = 16 * PITCH;
    const int     lane = threadIdx.x & 63, g = lane >> 4, c = lane & 15;
    wave_sync(); // the previous use of the buffer is over
    {
        const int r = lane >> 2, d = lane & 3;
#pragma unroll
        for (int q = 0; q < 4; ++q)
            if ((mask >> q) & 1) {
#pragma unroll
                for (int j = 0; j < 4; ++j) lds[q * AREA + r * PITCH + 4 * d + j] = (int32_t)((src >> (8 * j)) & 0xff) - (int32_t)((prd[q] >> (8 * j)) & 0xff);
            }
    }
    wave_sync();
    int32_t   *mine = lds + g * AREA;
    const bool on = (mask >> g) & 1;
    {
        int32_t in[16], out[16];
        if (on) {
#pragma unroll
            for (int i = 0; i < 16; ++i) in[i] = mine[i * PITCH + c] * 4;
            fdct_n<13>(in, out, p.cos13);
        }
        wave_sync();
        if (on) {
#pragma unroll
            for (int i = 0; i < 16; ++i) mine[i * PITCH + c] = (out[i] + 2) >> 2;
        }
    }
    wave_sync();
    long long a_sum = 0, e_sum = 0;
    int       eob = 0, logs = 0;
    const int kw = 16 >> p.pf_shape, kh = 16 >> p.pf_shape;
    if (on && c < kh) {
        int32_t in[16], out[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) in[i] = mine[c * PITCH + i];
        fdct_n<12>(in, out, p.cos12);
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int32_t v = i < kw ? out[i] : 0;
            const int32_t a = v < 0 ? -v : v;
            a_sum += a;
            if (quant) {
                const int ac = (c | i) != 0;
                int32_t   dq = 0;
                if ((a << 1) >= p.dequant[ac]) {
                    const int32_t t     = min(a + p.round_fp[ac], 32767);
                    const int32_t level = (int32_t)(((long long)t * p.quant_fp[ac]) >> 16);
                    dq                  = level * p.dequant[ac];
                    if (want_rate && level) {
                        eob = max(eob, tpl_scan_pos16(c, i) + 1);
                        logs += 31 - __clz(level + 1);
                    }
                }
                const long long diff = (long long)a - dq; // |v - sign(v) dq|: same square
                e_sum += diff * diff;
            }
        }
    }
#pragma unroll
    for (int m = 8; m >= 1; m >>= 1) { // within the quarter
        a_sum += __shfl_xor(a_sum, m);
        e_sum += __shfl_xor(e_sum, m);
        eob = max(eob, __shfl_xor(eob, m));
        logs += __shfl_xor(logs, m);
    }
    e_sum >>= 2;
    e_sum = e_sum < 1 ? 1 : e_sum;
    const long long r = (long long)(eob + 1 + logs) << 9;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        satd[q]        = __shfl(a_sum, 16 * q);
        recon_error[q] = __shfl(e_sum, 16 * q);
        rate[q]        = __shfl(r, 16 * q);
    }
}

// ---- without tpl_opt_flag: every intra mode of a 16x16 block (:693-821). The block's neighbours in the source picture lie in
// LDS as two byte arrays of 33 entries, entry i = -1 .. 31 at [4 + i]: the corner and the 32 pixels above / left of the
// block and of its right / lower neighbour.
constexpr int kEdge = 40; // bytes per edge array

// update_neighbor_samples_array_open_loop_mb(1, 1, ..) (EbEncIntraPrediction.c:1150-1235), an entry per lane: fill values
// 127 / 129; the corner is the pixel at (x - 1, y - 1), or 128 on the picture's first row and column; the column left of the
// block as far down as the picture goes, entries 16 .. 31 then repeat entry 15; on the first column the pixel above the
// block stands in for the corner and the column; the row above as far right as the picture goes, entries 16 .. 31 repeating
// entry 15 unless the block lies on the first column; on the first row one entry of the left column stands in for the corner
// and the row.
__device__ __forceinline__ void tpl_collect_edges(ConstTplParams &p, const DevPlane &cur, int x, int y, uint8_t *above, uint8_t *left) {
    const int lane = threadIdx.x & 63, i = lane - 1;
    auto      px   = [&](int xx, int yy) -> int { return cur.pix[(ptrdiff_t)yy * cur.stride + xx]; };
    const int nl = min(32, p.height - y), na = min(32, p.width - x);
    auto      left_at = [&](int k) -> int {
        if (x > 0) {
            if (k < 0) return y > 0 ? px(x - 1, y - 1) : px(x - 1, y);
            if (k >= 16) k = 15;
            return k < nl ? px(x - 1, y + k) : 129;
        }
        if (y > 0) return k < nl ? px(x, y - 1) : 129;
        return k < 0 ? 128 : 129;
    };
    if (lane < 33) {
        int a;
        if (y > 0) {
            if (i < 0) a = x > 0 ? px(x - 1, y - 1) : px(x, y - 1);
            else {
                const int k = (x > 0 && i >= 16) ? 15 : i;
                a           = k < na ? px(x + k, y - 1) : 127;
            }
        } else if (x > 0) a = i < na ? left_at(32 - na) : 127;
        else a = i < 0 ? 128 : 127;
        above[4 + i] = (uint8_t)a;
        left[4 + i]  = (uint8_t)left_at(i);
    }
}

// filter_intra_edge (EbIntraPrediction.c:2633-2689) for a 16x16 block and a directional mode of angle `angle` (not 90 or
// 180), from the collected arrays into the filtered ones, an entry per lane: the corner when both edges are read
// (filter_intra_edge_corner :2397-2404), then svt_av1_filter_intra_edge_c (:105-128) over each edge that exists in the
// picture - the corner, 16 entries and, for zone 1 / zone 3, the 16 of the extension - with the strength of
// intra_edge_filter_strength(16, 16, delta, 0) (:130-165): 3 from 32 degrees on, else 2. A 16x16 block's edges are never
// upsampled (use_intra_edge_upsample: width + height > 16).
__device__ __forceinline__ void tpl_filter_edges(int angle, int x, int y, const uint8_t *a0, const uint8_t *l0, uint8_t *a1, uint8_t *l1) {
    const int  lane = threadIdx.x & 63, idx = lane; // entry idx - 1
    const bool need_above = angle < 180, need_left = angle > 90;
    if (lane >= 33) return;
    int ca = a0[3], cl = l0[3];
    if (need_above && need_left) ca = cl = (l0[4] * 5 + a0[3] * 6 + a0[4] * 5 + 8) >> 4;
    auto filtered = [&](const uint8_t *e, int corner, int sz, int delta) -> int {
        auto at = [&](int k) -> int { return k <= 0 ? corner : e[3 + min(k, sz - 1)]; };
        if (idx < 1 || idx >= sz) return idx == 0 ? corner : e[3 + idx];
        const int d = delta < 0 ? -delta : delta;
        if (d >= 32) return (2 * at(idx - 2) + 4 * at(idx - 1) + 4 * at(idx) + 4 * at(idx + 1) + 2 * at(idx + 2) + 8) >> 4;
        return (5 * at(idx - 1) + 6 * at(idx) + 5 * at(idx + 1) + 8) >> 4;
    };
    a1[3 + idx] = (uint8_t)filtered(a0, ca, need_above && y > 0 ? 17 + (angle < 90 ? 16 : 0) : 0, angle - 90);
    l1[3 + idx] = (uint8_t)filtered(l0, cl, need_left && x > 0 ? 17 + (angle > 180 ? 16 : 0) : 0, angle - 180);
}

__constant__ const uint8_t kSmWeights16[16] = {255, 225, 196, 170, 145, 123, 102, 84, 68, 54, 43, 33, 26, 20, 17, 16}; // sm_weight_arrays + 16, :27-45

// intra_prediction_open_loop_mb (:2691-2712) for TX_16X16, this lane's four pixels (row lane / 4, columns 4 (lane % 4) ..):
// mode 1 V, 2 H, 3 D45, 4 D135, 5 D113, 6 D157, 7 D203, 8 D67 (dr_predictor :2373-2395 without upsampling: zones 1 / 2 / 3,
// :263-362, with eb_dr_intra_derivative 64 / 27 / 151 at 45 / 67 / 23 degrees), 9 SMOOTH, 10 SMOOTH_V, 11 SMOOTH_H
// (:1056-1120), 12 PAETH (:1145-1165). above / left point at entry 0.
__device__ __forceinline__ uint32_t tpl_intra_dword(int mode, const uint8_t *above, const uint8_t *left) {
    const int lane = threadIdx.x & 63, r = lane >> 2;
    // mode_to_angle_map (EbIntraPrediction.h:127-137) and the derivatives of the two edges
    const int angle = mode == 3 ? 45 : (mode == 4 ? 135 : (mode == 5 ? 113 : (mode == 6 ? 157 : (mode == 7 ? 203 : 67))));
    auto      deriv = [](int a) -> int { return a == 23 ? 151 : (a == 45 ? 64 : 27); };
    uint32_t  out   = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int c = 4 * (lane & 3) + j;
        int       v;
        if (mode == 1) v = above[c];
        else if (mode == 2) v = left[r];
        else if (mode <= 8 && angle < 90) {
            const int xx = (r + 1) * deriv(angle), base = (xx >> 6) + c, sh = (xx & 0x3F) >> 1;
            v            = base < 31 ? (above[base] * (32 - sh) + above[base + 1] * sh + 16) >> 5 : above[31];
        } else if (mode <= 8 && angle < 180) {
            const int xx = -(r + 1) * deriv(180 - angle), base = (xx >> 6) + c;
            if (base >= -1) {
                const int sh = (xx & 0x3F) >> 1;
                v            = (above[base] * (32 - sh) + above[base + 1] * sh + 16) >> 5;
            } else {
                const int yy = (r << 6) - (c + 1) * deriv(angle - 90), b2 = yy >> 6, sh = (yy & 0x3F) >> 1;
                v            = (left[b2] * (32 - sh) + left[b2 + 1] * sh + 16) >> 5;
            }
        } else if (mode <= 8) {
            const int yy = (c + 1) * deriv(270 - angle), base = (yy >> 6) + r, sh = (yy & 0x3F) >> 1;
            v            = base < 31 ? (left[base] * (32 - sh) + left[base + 1] * sh + 16) >> 5 : left[31];
        } else if (mode == 9) {
            v = (kSmWeights16[r] * above[c] + (256 - kSmWeights16[r]) * left[15] + kSmWeights16[c] * left[r] +
                 (256 - kSmWeights16[c]) * above[15] + 256) >> 9;
        } else if (mode == 10) v = (kSmWeights16[r] * above[c] + (256 - kSmWeights16[r]) * left[15] + 128) >> 8;
        else if (mode == 11) v = (kSmWeights16[c] * left[r] + (256 - kSmWeights16[c]) * above[15] + 128) >> 8;
        else {
            const int tl = above[-1], base = above[c] + left[r] - tl;
            const int pl = abs(base - left[r]), pt = abs(base - above[c]), ptl = abs(base - tl);
            v            = (pl <= pt && pl <= ptl) ? left[r] : (pt <= ptl ? above[c] : tl);
        }
        out |= (uint32_t)(v & 0xff) << (8 * j);
    }
    return out;
}

// G: how many predictions of a 16x16 block are costed together in the transform domain (4: the intra modes and the candidates
// four at a time; 1: one by one, for pictures whose costs are sums of absolute differences - less state, more waves per CU)
template <int S, int G>
__device__ __forceinline__ void tpl_block(ConstTplParams &p, int b64, int bi, int x, int y, int32_t *lds, uint8_t *edges) {
    constexpr int DW = S / 4, N = S * S / 256; // dwords per row, dwords per lane
    const int     lane = threadIdx.x & 63;
    const DevPlane cur = load_plane(p.cur);
    uint32_t       src[N];
#pragma unroll
    for (int k = 0; k < N; ++k) {
        const int idx = lane + 64 * k, r = idx / DW, d = idx % DW;
        src[k]        = *reinterpret_cast<const uint32_t *>(cur.pix + (ptrdiff_t)(y + r) * cur.stride + x + 4 * d);
    }
    long long best_intra = INT64_MAX, best_inter = INT64_MAX, best_err = 0, best_rate = 0;
    int       best_rf = -1, best_mvx = 0, best_mvy = 0, best_mode = 0;
    // an intra mode left over after the rounds of four (the thirteenth, or the ninth): its prediction, costed in the free
    // quarter of the first round of candidates
    uint32_t late_prd  = 0;
    int      late_mode = -1;

    if (S == 16 && G == 4 && !p.disable_intra_pred && p.mode_end > 0) {
        // every mode from DC_PRED to mode_end in the transform domain, the first of the cheapest stays (:735-820)
        if constexpr (S == 16 && G == 4) {
            uint8_t *a0 = edges, *l0 = edges + kEdge, *a1 = edges + 2 * kEdge, *l1 = edges + 3 * kEdge;
            wave_sync();
            tpl_collect_edges(p, cur, x, y, a0, l0);
            wave_sync();
            for (int mode0 = 0; mode0 <= p.mode_end; mode0 += 4) { // four modes' costs at a time
                uint32_t  prd[4] = {0, 0, 0, 0};
                const int n = min(4, p.mode_end + 1 - mode0);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int mode = mode0 + q;
                    if (q >= n) break;
                    if (mode == 0) { // dc_pred[x > 0][y > 0] (:2705-2707)
                        uint32_t sum = 0;
                        if (lane < 16) sum = (y > 0 ? a0[4 + lane] : 0u) + (x > 0 ? l0[4 + lane] : 0u);
                        sum               = wave_sum(sum);
                        const uint32_t nn = (y > 0 ? 16 : 0) + (x > 0 ? 16 : 0);
                        prd[q]            = (nn ? (sum + (nn >> 1)) / nn : 128u) * 0x01010101u;
                    } else if (mode >= 3 && mode <= 8) { // the directional modes read filtered copies of the edges (:743-760)
                        const int angle = mode == 3 ? 45 : (mode == 4 ? 135 : (mode == 5 ? 113 : (mode == 6 ? 157 : (mode == 7 ? 203 : 67))));
                        wave_sync();
                        tpl_filter_edges(angle, x, y, a0, l0, a1, l1);
                        wave_sync();
                        prd[q] = tpl_intra_dword(mode, a1 + 4, l1 + 4);
                    } else {
                        prd[q] = tpl_intra_dword(mode, a0 + 4, l0 + 4);
                    }
                }
                if (n == 1 && mode0 > 0) {
                    late_prd  = prd[0];
                    late_mode = mode0;
                    break;
                }
                long long cost[4], err[4], rate[4];
                tpl_transform_x4(p, src[0], prd, (1u << n) - 1, lds, false, false, cost, err, rate);
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if (q < n && cost[q] < best_intra) { // in the order of the modes: the first of the cheapest stays
                        best_intra = cost[q];
                        best_mode  = mode0 + q;
                    }
            }
        }
    } else if (!p.disable_intra_pred) {
        // the row above and the column left of the block; entries past the picture's right / bottom edge keep the fill
        // values 127 / 129 (update_neighbor_samples_array_open_loop_mb, EbEncIntraPrediction.c:1150-1235); which sides
        // count: dc_pred[x > 0][y > 0] (EbIntraPrediction.c:985-1035, :2705-2707)
        uint32_t sum = 0;
        if (lane < S) {
            if (y > 0) sum += x + lane < p.width ? cur.pix[(ptrdiff_t)(y - 1) * cur.stride + x + lane] : 127u;
            if (x > 0) sum += y + lane < p.height ? cur.pix[(ptrdiff_t)(y + lane) * cur.stride + x - 1] : 129u;
        }
        sum = wave_sum(sum);
        const uint32_t n  = (y > 0 ? S : 0) + (x > 0 ? S : 0);
        const uint32_t dc = n ? (sum + (n >> 1)) / n : 128u;
        uint32_t       acc = 0;
#pragma unroll
        for (int k = 0; k < N; ++k) acc = __builtin_amdgcn_sad_u8(src[k], dc * 0x01010101u, acc);
        best_intra = wave_sum(acc);
        if constexpr (S == 16) {
            if (!p.sad_intra) { // :790-813
                long long      err;
                const uint32_t prd[1] = {dc * 0x01010101u};
                tpl_transform<16, 16, 1>(p, src, prd, lds, false, best_intra, err);
            }
        }
    }

    int me_off = kTplMeIndex[bi]; // :826-834
    if (!p.enable_me_16x16) me_off = (me_off - 1) / 4;
    int            n_cand = p.i_slice ? 0 : p.cnt[(size_t)b64 * p.n_pu + me_off];
    if (n_cand > p.max_cand) { // a count the arrays cannot hold: flagged (the call returns SVTME_ERR_BAD_ARG), not followed
        if (lane == 0) atomicOr(p.error, 2);
        n_cand = p.max_cand;
    }
    const uint8_t *cand   = p.cand + ((size_t)b64 * p.n_pu + me_off) * p.max_cand;
    const uint32_t *mvs   = p.mv + ((size_t)b64 * p.n_pu + me_off) * p.max_refs;
    // :836-930, G candidates at a time: their vectors and SADs one after the other, their transform-domain costs together
    for (int ci = 0; ci < n_cand;) {
        int       n = 0, c_rf[G], c_mvx[G], c_mvy[G];
        long long c_cost[G], c_err[G] = {}, c_rate[G] = {};
        uint32_t  c_f[G] = {};
#pragma unroll
        for (int q = 0; q < G; ++q) {
            if (G == 4 && q == 3 && late_mode >= 0) break; // that quarter is taken
            int list = 0, ref = 0;
            bool found = false;
            while (ci < n_cand && !found) { // the next candidate that takes part
                const uint32_t c = cand[ci++];
                if ((c & 3) > 1) continue; // single references only
                list = c & 3;
                ref  = list == 0 ? (c >> 2) & 3 : (c >> 4) & 3;
                if (ref >= p.n_ref[list]) { // a candidate that names a reference the caller did not hand over
                    if (lane == 0) atomicOr(p.error, 1);
                    continue;
                }
                found = !p.ref_skip[list][ref];
            }
            if (!found) break;
            n                  = q + 1;
            const uint32_t mv  = mvs[(list ? p.max_l0 : 0) + ref];
            int16_t        mvx = (int16_t)((int16_t)(mv & 0xffff) << 1), mvy = (int16_t)((int16_t)(mv >> 16) << 1);
            if (x + (mvx >> 3) < -kTplPad) mvx = (int16_t)((-kTplPad - x) << 3);
            if (x + S + (mvx >> 3) > kTplPad + p.width - 1) mvx = (int16_t)(((kTplPad + p.width - 1) - (x + S)) << 3);
            if (y + (mvy >> 3) < -kTplPad) mvy = (int16_t)((-kTplPad - y) << 3);
            if (y + S + (mvy >> 3) > kTplPad + p.height - 1) mvy = (int16_t)(((kTplPad + p.height - 1) - (y + S)) << 3);
            const DevPlane rp  = load_plane(p.ref[list][ref]);
            const uint8_t *org = rp.pix + (ptrdiff_t)(y + (mvy >> 3)) * rp.stride + x + (mvx >> 3);
            uint32_t       acc = 0;
#pragma unroll
            for (int k = 0; k < N; ++k) {
                const int idx = lane + 64 * k, r = idx / DW, d = idx % DW;
                uint32_t  f;
                __builtin_memcpy(&f, org + (ptrdiff_t)r * rp.stride + 4 * d, 4);
                acc = __builtin_amdgcn_sad_u8(src[k], f, acc);
                if (k == 0) c_f[q] = f; // (a 16x16 block is this one dword per lane)
            }
            c_cost[q] = wave_sum(acc);
            c_rf[q]   = list == 0 ? ref : 4 + ref; // svt_get_ref_frame_type(list, ref) - 1
            c_mvx[q]  = mvx;
            c_mvy[q]  = mvy;
        }
        if constexpr (S == 16) {
            if (!p.sad_inter && n) { // :894-917
                if constexpr (G == 4) {
                    const bool late = late_mode >= 0;
                    if (late) c_f[3] = late_prd;
                    tpl_transform_x4(p, src[0], c_f, ((1u << n) - 1) | (late ? 8u : 0u), lds, p.src_dist != 0, p.rate != 0, c_cost, c_err, c_rate);
                    if (late && c_cost[3] < best_intra) { // the last of the modes: after all the others
                        best_intra = c_cost[3];
                        best_mode  = late_mode;
                    }
                    late_mode = -1;
                } else
                    tpl_transform<16, 16, 1>(p, src, c_f, lds, p.src_dist != 0, c_cost[0], c_err[0], p.rate != 0, &c_rate[0]);
            }
        }
#pragma unroll
        for (int q = 0; q < G; ++q)
            if (q < n && c_cost[q] < best_inter) { // strictly smaller, in the candidates' order: the first of equal ones stays
                best_inter = c_cost[q];
                best_err   = c_err[q];
                best_rate  = c_rate[q];
                best_rf    = c_rf[q];
                best_mvx   = c_mvx[q];
                best_mvy   = c_mvy[q];
            }
    }
    if constexpr (S == 16 && G == 4) {
        if (late_mode >= 0) { // no round of candidates took it along
            const uint32_t prd[4] = {late_prd, 0, 0, 0};
            long long      cost[4], err[4], rate[4];
            tpl_transform_x4(p, src[0], prd, 1, lds, false, false, cost, err, rate);
            if (cost[0] < best_intra) {
                best_intra = cost[0];
                best_mode  = late_mode;
            }
        }
    }
    const bool is_inter = best_inter < best_intra; // :932-933
    if constexpr (S <= 32) {
        if (p.src_dist && p.sad_inter && is_inter) { // the transform of the chosen candidate (:938-970)
            const DevPlane rp  = load_plane(p.ref[best_rf >> 2][best_rf & 3]);
            const uint8_t *org = rp.pix + (ptrdiff_t)(y + (best_mvy >> 3)) * rp.stride + x + (best_mvx >> 3);
            uint32_t       f[N];
#pragma unroll
            for (int k = 0; k < N; ++k) {
                const int idx = lane + 64 * k, r = idx / DW, d = idx % DW;
                __builtin_memcpy(&f[k], org + (ptrdiff_t)r * rp.stride + 4 * d, 4);
            }
            long long satd;
            if constexpr (S == 16) {
                if (p.subsample_tx == 0) tpl_transform<16, 16, N>(p, src, f, lds, true, satd, best_err);
                else if (p.subsample_tx == 1) tpl_transform<16, 8, N>(p, src, f, lds, true, satd, best_err);
                else tpl_transform<16, 4, N>(p, src, f, lds, true, satd, best_err);
            } else {
                tpl_transform<32, 8, N>(p, src, f, lds, true, satd, best_err);
            }
        }
    }
    if (lane == 0) {
        svtme_tpl_block o;
        o.best_intra_cost = best_intra;
        o.best_inter_cost = best_inter;
        o.mv_x            = (int16_t)best_mvx;
        o.mv_y            = (int16_t)best_mvy;
        o.best_rf_idx     = (int8_t)best_rf;
        o.best_is_inter   = is_inter;
        o.best_intra_mode = (uint8_t)best_mode; // DC_PRED (0) unless every mode was tried
        o.valid           = 1;
        o.src_recon_error = p.src_dist && is_inter ? best_err : 0;
        o.src_rate        = p.rate && is_inter ? best_rate : 0;
        p.out[(size_t)b64 * SVTME_TPL_BLOCKS + bi] = o;
    }
}

// grid (16, n_b64), one wave each: slot s of a 64x64 block is its s-th 16x16 block (level 0), its s-th 32x32 block
// (level 1, s < 4) or the block itself (level 2, s == 0); blocks that are not complete run at level 0 (:2086-2088)
template <int G>
__global__ void __launch_bounds__(64, 4) tpl_search_kernel(const TplParams *pp) {
    __shared__ int32_t lds[G == 4 ? kTplLdsIntsX4 : kTplLdsInts];
    __shared__ __align__(4) uint8_t edges[4 * kEdge];
    ConstTplParams    &p = *(ConstTplParams *)pp;
    const int b64 = blockIdx.y, slot = blockIdx.x;
    const int sb_x = (b64 % p.b64_w) * 64, sb_y = (b64 / p.b64_w) * 64;
    const int level = (sb_x + 64 <= p.width && sb_y + 64 <= p.height) ? p.level : 0;
    if (level == 0) {
        const int q = slot >> 2, s = slot & 3; // z-order: 32x32 quadrant, 16x16 inside it
        const int x = sb_x + (q & 1) * 32 + (s & 1) * 16, y = sb_y + (q >> 1) * 32 + (s >> 1) * 16;
        if (x + 8 > p.width || y + 8 > p.height) return; // at least half of the block inside (:604-607)
        tpl_block<16, G>(p, b64, 5 + slot, x, y, lds, edges);
    } else if (level == 1) {
        if (slot >= 4) return;
        tpl_block<32, G>(p, b64, 1 + slot, sb_x + (slot & 1) * 32, sb_y + (slot >> 1) * 32, lds, edges);
    } else {
        if (slot >= 1) return;
        tpl_block<64, G>(p, b64, 0, sb_x, sb_y, lds, edges);
    }
}


// ---- the reconstruction path (:1004-1197) of a 16x16 block: prediction, residual, forward transform, quantisation, inverse
// transform into the reconstructed picture, statistics.
//
// The inverse DCT of svt_av1_inv_txfm2d_add_16xH_c (inv_txfm2d_add_c, EbInvTransforms.c:2455-2533): every row of coefficients
// (x 1/sqrt(2) for 16x8), clamped to 16 bits, through the 16-point network, rounded >> 2 (16x16) or >> 1; every column, clamped
// to 16 bits, through the H-point network, rounded >> 4, added to the prediction, clipped. The networks are
// svt_av1_idct4_new .. svt_av1_idct16_new (:75-355) with 12-bit cosines and every sum / difference clamped to 16 bits (the
// stage range at 8 bits, :23-65), written as the N/2-point network of the even inputs plus the odd half. Inputs are clamped
// to 16 bits and cosines are below 2^13, so the 24-bit multiplications of rot<> are exact.
#define TPL_C(k) c[(k) >> 1]
__device__ __forceinline__ int32_t clamp16(int32_t v) { return min(max(v, -32768), 32767); }

__device__ __forceinline__ void idct_n(const int32_t (&x)[4], int32_t (&X)[4], CosTable c) {
    const int32_t a0 = rot<12>(x[0], TPL_C(32), x[2], TPL_C(32)), a1 = rot<12>(x[0], TPL_C(32), x[2], -TPL_C(32));
    const int32_t a2 = rot<12>(x[1], TPL_C(48), x[3], -TPL_C(16)), a3 = rot<12>(x[1], TPL_C(16), x[3], TPL_C(48));
    X[0] = clamp16(a0 + a3), X[1] = clamp16(a1 + a2), X[2] = clamp16(a1 - a2), X[3] = clamp16(a0 - a3);
}

__device__ __forceinline__ void idct_n(const int32_t (&x)[8], int32_t (&X)[8], CosTable c) {
    const int32_t ev[4] = {x[0], x[2], x[4], x[6]};
    int32_t       E[4];
    idct_n(ev, E, c);
    const int32_t b4 = rot<12>(x[1], TPL_C(56), x[7], -TPL_C(8)), b5 = rot<12>(x[5], TPL_C(24), x[3], -TPL_C(40));
    const int32_t b6 = rot<12>(x[5], TPL_C(40), x[3], TPL_C(24)), b7 = rot<12>(x[1], TPL_C(8), x[7], TPL_C(56));
    const int32_t c4 = clamp16(b4 + b5), c5 = clamp16(b4 - b5), c6 = clamp16(-b6 + b7), c7 = clamp16(b6 + b7);
    const int32_t d[4] = {c4, rot<12>(c5, -TPL_C(32), c6, TPL_C(32)), rot<12>(c5, TPL_C(32), c6, TPL_C(32)), c7};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        X[i]     = clamp16(E[i] + d[3 - i]);
        X[7 - i] = clamp16(E[i] - d[3 - i]);
    }
}

__device__ __forceinline__ void idct_n(const int32_t (&x)[16], int32_t (&X)[16], CosTable c) {
    int32_t ev[8], E[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) ev[i] = x[2 * i];
    idct_n(ev, E, c);
    const int32_t b8 = rot<12>(x[1], TPL_C(60), x[15], -TPL_C(4)), b9 = rot<12>(x[9], TPL_C(28), x[7], -TPL_C(36));
    const int32_t b10 = rot<12>(x[5], TPL_C(44), x[11], -TPL_C(20)), b11 = rot<12>(x[13], TPL_C(12), x[3], -TPL_C(52));
    const int32_t b12 = rot<12>(x[13], TPL_C(52), x[3], TPL_C(12)), b13 = rot<12>(x[5], TPL_C(20), x[11], TPL_C(44));
    const int32_t b14 = rot<12>(x[9], TPL_C(36), x[7], TPL_C(28)), b15 = rot<12>(x[1], TPL_C(4), x[15], TPL_C(60));
    const int32_t c8 = clamp16(b8 + b9), c9 = clamp16(b8 - b9), c10 = clamp16(-b10 + b11), c11 = clamp16(b10 + b11);
    const int32_t c12 = clamp16(b12 + b13), c13 = clamp16(b12 - b13), c14 = clamp16(-b14 + b15), c15 = clamp16(b14 + b15);
    const int32_t d9 = rot<12>(c9, -TPL_C(16), c14, TPL_C(48)), d10 = rot<12>(c10, -TPL_C(48), c13, -TPL_C(16));
    const int32_t d13 = rot<12>(c10, -TPL_C(16), c13, TPL_C(48)), d14 = rot<12>(c9, TPL_C(48), c14, TPL_C(16));
    const int32_t e8 = clamp16(c8 + c11), e9 = clamp16(d9 + d10), e10 = clamp16(d9 - d10), e11 = clamp16(c8 - c11);
    const int32_t e12 = clamp16(-c12 + c15), e13 = clamp16(-d13 + d14), e14 = clamp16(d13 + d14), e15 = clamp16(c12 + c15);
    const int32_t f[8] = {e8, e9, rot<12>(e10, -TPL_C(32), e13, TPL_C(32)), rot<12>(e11, -TPL_C(32), e12, TPL_C(32)),
                          rot<12>(e11, TPL_C(32), e12, TPL_C(32)), rot<12>(e10, TPL_C(32), e13, TPL_C(32)), e14, e15};
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        X[i]      = clamp16(E[i] + f[7 - i]);
        X[15 - i] = clamp16(E[i] - f[7 - i]);
    }
}

// svt_av1_idct32_new (:356-706): the odd half takes inputs 1, 17, 9, 25, 5, 21, 13, 29, 3, 19, 11, 27, 7, 23, 15, 31 as nodes
// 16 .. 31 (the arrays below are indexed node - 16)
__device__ __forceinline__ void idct_n(const int32_t (&x)[32], int32_t (&X)[32], CosTable c) {
    constexpr int in_of[16] = {1, 17, 9, 25, 5, 21, 13, 29, 3, 19, 11, 27, 7, 23, 15, 31};
    constexpr int k2[8]     = {62, 30, 46, 14, 54, 22, 38, 6}; // stage 2: node 16 + i with node 31 - i
    int32_t       ev[16], E[16], o[16], b[16], d[16], e[16], f[16], g[16], h[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) ev[i] = x[2 * i], o[i] = x[in_of[i]];
    idct_n(ev, E, c);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        b[i]      = rot<12>(o[i], TPL_C(k2[i]), o[15 - i], -TPL_C(64 - k2[i]));
        b[15 - i] = rot<12>(o[i], TPL_C(64 - k2[i]), o[15 - i], TPL_C(k2[i]));
    }
#pragma unroll
    for (int i = 0; i < 16; i += 4) { // stage 3
        d[i] = clamp16(b[i] + b[i + 1]), d[i + 1] = clamp16(b[i] - b[i + 1]);
        d[i + 2] = clamp16(-b[i + 2] + b[i + 3]), d[i + 3] = clamp16(b[i + 2] + b[i + 3]);
    }
    // stage 4
    b[0] = d[0], b[3] = d[3], b[4] = d[4], b[7] = d[7], b[8] = d[8], b[11] = d[11], b[12] = d[12], b[15] = d[15];
    b[1] = rot<12>(d[1], -TPL_C(8), d[14], TPL_C(56)), b[14] = rot<12>(d[1], TPL_C(56), d[14], TPL_C(8));
    b[2] = rot<12>(d[2], -TPL_C(56), d[13], -TPL_C(8)), b[13] = rot<12>(d[2], -TPL_C(8), d[13], TPL_C(56));
    b[5] = rot<12>(d[5], -TPL_C(40), d[10], TPL_C(24)), b[10] = rot<12>(d[5], TPL_C(24), d[10], TPL_C(40));
    b[6] = rot<12>(d[6], -TPL_C(24), d[9], -TPL_C(40)), b[9] = rot<12>(d[6], -TPL_C(40), d[9], TPL_C(24));
    // stage 5
    e[0] = clamp16(b[0] + b[3]), e[1] = clamp16(b[1] + b[2]), e[2] = clamp16(b[1] - b[2]), e[3] = clamp16(b[0] - b[3]);
    e[4] = clamp16(-b[4] + b[7]), e[5] = clamp16(-b[5] + b[6]), e[6] = clamp16(b[5] + b[6]), e[7] = clamp16(b[4] + b[7]);
    e[8] = clamp16(b[8] + b[11]), e[9] = clamp16(b[9] + b[10]), e[10] = clamp16(b[9] - b[10]), e[11] = clamp16(b[8] - b[11]);
    e[12] = clamp16(-b[12] + b[15]), e[13] = clamp16(-b[13] + b[14]), e[14] = clamp16(b[13] + b[14]), e[15] = clamp16(b[12] + b[15]);
    // stage 6
    f[0] = e[0], f[1] = e[1], f[6] = e[6], f[7] = e[7], f[8] = e[8], f[9] = e[9], f[14] = e[14], f[15] = e[15];
    f[2] = rot<12>(e[2], -TPL_C(16), e[13], TPL_C(48)), f[13] = rot<12>(e[2], TPL_C(48), e[13], TPL_C(16));
    f[3] = rot<12>(e[3], -TPL_C(16), e[12], TPL_C(48)), f[12] = rot<12>(e[3], TPL_C(48), e[12], TPL_C(16));
    f[4] = rot<12>(e[4], -TPL_C(48), e[11], -TPL_C(16)), f[11] = rot<12>(e[4], -TPL_C(16), e[11], TPL_C(48));
    f[5] = rot<12>(e[5], -TPL_C(48), e[10], -TPL_C(16)), f[10] = rot<12>(e[5], -TPL_C(16), e[10], TPL_C(48));
#pragma unroll
    for (int i = 0; i < 4; ++i) { // stage 7
        g[i]      = clamp16(f[i] + f[7 - i]);
        g[7 - i]  = clamp16(f[i] - f[7 - i]);
        g[8 + i]  = clamp16(-f[8 + i] + f[15 - i]);
        g[15 - i] = clamp16(f[8 + i] + f[15 - i]);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) h[i] = g[i], h[12 + i] = g[12 + i]; // stage 8
#pragma unroll
    for (int i = 4; i < 8; ++i) {
        h[i]      = rot<12>(g[i], -TPL_C(32), g[15 - i], TPL_C(32));
        h[15 - i] = rot<12>(g[i], TPL_C(32), g[15 - i], TPL_C(32));
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) { // stage 9
        X[i]      = clamp16(E[i] + h[15 - i]);
        X[31 - i] = clamp16(E[i] - h[15 - i]);
    }
}
#undef TPL_C

// The transforms of a block of side W whose lanes hold its source and predicted pixels (dword lane + 64 k of the block's rows,
// N = W W / 256 dwords per lane): the W x H forward DCT of every (W / H)-th row's residual and its quantisation as in
// tpl_transform; then, when `inverse` and a level is left, the inverse DCT of the dequantised coefficients added to the
// prediction of those rows, the rows between taking the row above them (:1145-1176). (W, H) = (16, 16), (16, 8), (16, 4),
// (32, 8). Leaves the lane's reconstructed pixels in out[]; recon_error and rate in every lane.
template <int W, int H, int N>
__device__ __forceinline__ void tpl_recon_transform(ConstTplParams &p, const uint32_t (&src)[N], const uint32_t (&prd)[N], uint32_t (&out)[N],
                                                    int32_t *lds, bool inverse, bool want_rate, long long &recon_error, long long &rate) {
    constexpr int  PITCH = W + 1, DW = W / 4, STEP = W / H, SHIFT1 = (W == 16 && H == 4) ? 1 : 2;
    constexpr int  BIT_ROW = ((W == 16 && H == 16) || W == 32) ? 12 : 13, SHIFT_INV = (W == 16 && H < 16) ? 1 : 2;
    constexpr bool RECT = W == 2 * H, RATE = W == 16 && H == 16;
    static_assert(H * PITCH <= 16 * 17, "the coefficients fit the buffer");
    const int lane = threadIdx.x & 63;
    wave_sync();
#pragma unroll
    for (int k = 0; k < N; ++k) {
        const int idx = lane + 64 * k, r = idx / DW, d = idx % DW;
        if (r % STEP == 0) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                lds[(r / STEP) * PITCH + 4 * d + j] = (int32_t)((src[k] >> (8 * j)) & 0xff) - (int32_t)((prd[k] >> (8 * j)) & 0xff);
        }
    }
    wave_sync();
    {
        int32_t in[H], o[H];
        if (lane < W) {
#pragma unroll
            for (int i = 0; i < H; ++i) in[i] = lds[i * PITCH + lane] * 4;
            fdct_n<13>(in, o, p.cos13);
        }
        wave_sync();
        if (lane < W) {
#pragma unroll
            for (int i = 0; i < H; ++i) lds[i * PITCH + lane] = (o[i] + (1 << (SHIFT1 - 1))) >> SHIFT1;
        }
    }
    wave_sync();
    long long e_sum = 0;
    int       eob = 0, logs = 0, any = 0;
    const int kw = W >> p.pf_shape, kh = H >> p.pf_shape;
    if (lane < H) {
        int32_t in[W], o[W];
#pragma unroll
        for (int i = 0; i < W; ++i) in[i] = lds[lane * PITCH + i];
        if (lane < kh) fdct_n<BIT_ROW>(in, o, BIT_ROW == 12 ? p.cos12 : p.cos13);
#pragma unroll
        for (int i = 0; i < W; ++i) {
            int32_t v = (lane < kh && i < kw) ? o[i] : 0;
            if (RECT) v = (int32_t)(((long long)v * 5793 + 2048) >> 12); // NewSqrt2, 12 fractional bits
            const int32_t a  = v < 0 ? -v : v;
            const int     ac = (lane | i) != 0;
            int32_t       dq = 0;
            if ((a << 1) >= p.dequant[ac]) {
                const int32_t t     = min(a + p.round_fp[ac], 32767);
                const int32_t level = (int32_t)(((long long)t * p.quant_fp[ac]) >> 16);
                dq                  = level * p.dequant[ac];
                if (level) {
                    any = 1;
                    if (RATE && want_rate) {
                        eob = max(eob, tpl_scan_pos16(lane, i) + 1);
                        logs += 31 - __clz(level + 1);
                    }
                }
            }
            const long long diff = (long long)a - dq;
            e_sum += diff * diff;
            lds[lane * PITCH + i] = v < 0 ? -dq : dq; // dqcoeff
        }
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        e_sum += __shfl_xor(e_sum, m);
        any |= __shfl_xor(any, m);
        eob = max(eob, __shfl_xor(eob, m));
        logs += __shfl_xor(logs, m);
    }
    e_sum >>= 2;
    recon_error = e_sum < 1 ? 1 : e_sum;
    rate        = RATE && want_rate ? (long long)(eob + 1 + logs) << 9 : 0;
#pragma unroll
    for (int k = 0; k < N; ++k) out[k] = prd[k];
    if (!inverse || !any) return;
    wave_sync();
    {
        int32_t in[W], o[W];
        if (lane < H) {
#pragma unroll
            for (int i = 0; i < W; ++i) {
                int32_t v = lds[lane * PITCH + i];
                if (RECT) v = (int32_t)(((long long)v * 2896 + 2048) >> 12); // new_inv_sqrt2, 12 fractional bits
                in[i] = clamp16(v);
            }
            idct_n(in, o, p.cos12);
#pragma unroll
            for (int i = 0; i < W; ++i) lds[lane * PITCH + i] = (o[i] + (1 << (SHIFT_INV - 1))) >> SHIFT_INV;
        }
    }
    wave_sync();
    {
        int32_t in[H], o[H];
        if (lane < W) {
#pragma unroll
            for (int i = 0; i < H; ++i) in[i] = clamp16(lds[i * PITCH + lane]);
            idct_n(in, o, p.cos12);
        }
        wave_sync();
        if (lane < W) {
#pragma unroll
            for (int i = 0; i < H; ++i) lds[i * PITCH + lane] = (o[i] + 8) >> 4;
        }
    }
    wave_sync();
    // a pixel's transformed row: its own, or the one above it that was transformed - whose dword the same k of another lane holds
    // (64 / DW rows per k, a multiple of STEP)
    static_assert((64 / DW) % STEP == 0, "a row and its transformed row share k");
#pragma unroll
    for (int k = 0; k < N; ++k) {
        const int      idx = lane + 64 * k, r = idx / DW, d = idx % DW, rt = r - r % STEP;
        const uint32_t pt  = (uint32_t)__shfl((int)prd[k], (rt * DW + d) & 63);
        uint32_t       o   = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int v = (int)((pt >> (8 * j)) & 0xff) + lds[(rt / STEP) * PITCH + 4 * d + j];
            o |= (uint32_t)min(max(v, 0), 255) << (8 * j);
        }
        out[k] = o;
    }
}

// One block of side S of the reconstruction path, whole wave; `b` is the search's record of the block.
template <int S>
__device__ __forceinline__ void tpl_recon_block(ConstTplParams &p, int b64, int bi, int x, int y, const svtme_tpl_block &b, int32_t *lds, uint8_t *edges) {
    constexpr int  DW = S / 4, N = S * S / 256;
    const int      lane = threadIdx.x & 63;
    const DevPlane cur = load_plane(p.cur);
    DevPlane       rec = cur; // the reconstructed picture: the picture's size, no border
    rec.pix            = p.recon;
    rec.stride         = p.recon_stride;
    uint32_t src[N], prd[N], out[N];
#pragma unroll
    for (int k = 0; k < N; ++k) {
        const int idx = lane + 64 * k, r = idx / DW, d = idx % DW;
        src[k]        = *reinterpret_cast<const uint32_t *>(cur.pix + (ptrdiff_t)(y + r) * cur.stride + x + 4 * d);
    }
    if (b.best_is_inter) { // :1005-1030
        const DevPlane rp  = load_plane(p.ref[b.best_rf_idx >> 2][b.best_rf_idx & 3]);
        const uint8_t *org = rp.pix + (ptrdiff_t)(y + (b.mv_y >> 3)) * rp.stride + x + (b.mv_x >> 3);
#pragma unroll
        for (int k = 0; k < N; ++k) {
            const int idx = lane + 64 * k, r = idx / DW, d = idx % DW;
            __builtin_memcpy(&prd[k], org + (ptrdiff_t)r * rp.stride + 4 * d, 4);
        }
    } else if (S == 32 || p.dc_path || b.best_intra_mode == 0) { // :1044-1078; DC_PRED of :1080-1119 sees the same neighbours
        uint32_t sum = 0;
        if (lane < S) {
            if (y > 0) sum += x + lane < p.width ? rec.pix[(ptrdiff_t)(y - 1) * rec.stride + x + lane] : 127u;
            if (x > 0) sum += y + lane < p.height ? rec.pix[(ptrdiff_t)(y + lane) * rec.stride + x - 1] : 129u;
        }
        sum              = wave_sum(sum);
        const uint32_t n = (y > 0 ? S : 0) + (x > 0 ? S : 0);
        const uint32_t v = (n ? (sum + (n >> 1)) / n : 128u) * 0x01010101u;
#pragma unroll
        for (int k = 0; k < N; ++k) prd[k] = v;
    } else { // :1080-1119
        if constexpr (S == 16) {
            uint8_t  *a0 = edges, *l0 = edges + kEdge, *a1 = edges + 2 * kEdge, *l1 = edges + 3 * kEdge;
            const int mode = b.best_intra_mode;
            wave_sync();
            tpl_collect_edges(p, rec, x, y, a0, l0);
            wave_sync();
            if (mode >= 3 && mode <= 8) {
                const int angle = mode == 3 ? 45 : (mode == 4 ? 135 : (mode == 5 ? 113 : (mode == 6 ? 157 : (mode == 7 ? 203 : 67))));
                tpl_filter_edges(angle, x, y, a0, l0, a1, l1);
                wave_sync();
                prd[0] = tpl_intra_dword(mode, a1 + 4, l1 + 4);
            } else
                prd[0] = tpl_intra_dword(mode, a0 + 4, l0 + 4);
        }
    }
    long long  recon_error, rate;
    const bool inverse = !p.disable_intra_pred, want_rate = p.rate != 0;
    if constexpr (S == 16) {
        if (p.subsample_tx == 0) tpl_recon_transform<16, 16, N>(p, src, prd, out, lds, inverse, want_rate, recon_error, rate);
        else if (p.subsample_tx == 1) tpl_recon_transform<16, 8, N>(p, src, prd, out, lds, inverse, false, recon_error, rate);
        else tpl_recon_transform<16, 4, N>(p, src, prd, out, lds, inverse, false, recon_error, rate);
    } else {
        tpl_recon_transform<32, 8, N>(p, src, prd, out, lds, inverse, false, recon_error, rate);
    }
#pragma unroll
    for (int k = 0; k < N; ++k) {
        const int idx = lane + 64 * k, r = idx / DW, d = idx % DW;
        if (y + r < p.height) {
            uint8_t *dst = p.recon + (ptrdiff_t)(y + r) * p.recon_stride + x + 4 * d;
            if (x + 4 * d + 4 <= p.width) *reinterpret_cast<uint32_t *>(dst) = out[k];
            else
                for (int j = 0; j < 4 && x + 4 * d + j < p.width; ++j) dst[j] = (uint8_t)(out[k] >> (8 * j));
        }
    }
    if (lane == 0) { // :1179-1197, result_model_store :214-222
        const int       sub = p.subsample_tx;
        svtme_tpl_stats o;
        o.recrf_dist = (recon_error << 4) << sub;
        o.recrf_rate = (rate << 4) << sub;
        if (b.best_is_inter) {
            o.srcrf_dist = (b.src_recon_error << 4) << sub;
            o.srcrf_rate = (b.src_rate << 4) << sub;
        } else {
            o.srcrf_dist = o.recrf_dist;
            o.srcrf_rate = o.recrf_rate;
        }
        o.recrf_dist = max(o.recrf_dist, o.srcrf_dist);
        o.recrf_rate = max(o.recrf_rate, o.srcrf_rate);
        o.srcrf_dist = max(o.srcrf_dist / S, 1ll);
        o.recrf_dist = max(o.recrf_dist / S, 1ll);
        o.srcrf_rate = max(o.srcrf_rate / S, 1ll);
        o.recrf_rate = max(o.recrf_rate / S, 1ll);
        p.stats[(size_t)b64 * SVTME_TPL_BLOCKS + bi] = o;
    }
}

// INTRA false: grid (16, n_b64) as the search (slot s of a 64x64 block: its s-th 16x16 block, or at level 1 its s-th 32x32
// block, s < 4), the blocks predicted from reference pictures, which depend on nothing.
// INTRA true: blocks predicted from their reconstructed neighbours - the blocks left of, above and above-left of them, and
// for the picture's first column the block above-right (update_neighbor_samples_array_open_loop_mb_recon replaces the
// upper-right and lower-left extensions by repeats everywhere else). The host has sorted them into levels, a block's level
// one more than the highest of its intra neighbours'; a launch runs one level: blocks list[first + blockIdx.x] (position in
// 16-pixel units, x | y << 16).
template <bool INTRA>
__global__ void __launch_bounds__(64) tpl_recon_kernel(const TplParams *pp, const uint32_t *list, int first) {
    __shared__ int32_t lds[16 * 17];
    __shared__ __align__(4) uint8_t edges[4 * kEdge];
    ConstTplParams    &p = *(ConstTplParams *)pp;
    int bx, by; // in 16-pixel units
    if (INTRA) {
        const uint32_t at = list[first + (int)blockIdx.x];
        bx = (int)(at & 0xFFFFu);
        by = (int)(at >> 16);
    } else {
        const int b64 = blockIdx.y, slot = blockIdx.x, sx = (b64 % p.b64_w) * 4, sy = (b64 / p.b64_w) * 4;
        if (p.level == 1 && sx * 16 + 64 <= p.width && sy * 16 + 64 <= p.height) {
            if (slot >= 4) return;
            bx = sx + (slot & 1) * 2;
            by = sy + (slot >> 1) * 2;
        } else {
            const int q = slot >> 2, s = slot & 3;
            bx = sx + (q & 1) * 2 + (s & 1);
            by = sy + (q >> 1) * 2 + (s >> 1);
        }
    }
    const int  x = bx * 16, y = by * 16, b64 = (by >> 2) * p.b64_w + (bx >> 2);
    const bool big = p.level == 1 && (bx & ~3) * 16 + 64 <= p.width && (by & ~3) * 16 + 64 <= p.height; // a 32x32 block of a complete 64x64 block
    const int  S = big ? 32 : 16;
    if (x + (S >> 1) > p.width || y + (S >> 1) > p.height) return; // at least half of the block inside (:604-607)
    const int bi = big ? 1 + ((by >> 1) & 1) * 2 + ((bx >> 1) & 1)
                       : 5 + (((by >> 1) & 1) * 2 + ((bx >> 1) & 1)) * 4 + (by & 1) * 2 + (bx & 1); // z-order inside the 64x64 block
    const svtme_tpl_block b = p.out[(size_t)b64 * SVTME_TPL_BLOCKS + bi];
    if (!b.valid || (b.best_is_inter != 0) == INTRA) return;
    if (big) tpl_recon_block<32>(p, b64, bi, x, y, b, lds, edges);
    else tpl_recon_block<16>(p, b64, bi, x, y, b, lds, edges);
}

} // namespace

void launch_tpl_search(const TplParams *dev, int n_b64, bool transform_costs, hipStream_t s) {
    if (transform_costs) hipLaunchKernelGGL(tpl_search_kernel<4>, dim3(16, n_b64), dim3(64), 0, s, dev);
    else hipLaunchKernelGGL(tpl_search_kernel<1>, dim3(16, n_b64), dim3(64), 0, s, dev);
}

void launch_tpl_recon(const TplParams *dev, int n_b64, const uint32_t *list, const int *level_count, int n_levels, hipStream_t s) {
    hipLaunchKernelGGL(tpl_recon_kernel<false>, dim3(16, n_b64), dim3(64), 0, s, dev, (const uint32_t *)nullptr, 0);
    for (int l = 0, first = 0; l < n_levels; first += level_count[l++])
        hipLaunchKernelGGL(tpl_recon_kernel<true>, dim3(level_count[l]), dim3(64), 0, s, dev, list, first);
}

} // namespace svtme
