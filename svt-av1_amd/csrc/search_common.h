// search_common.h - device helpers shared by the HME and integer-search kernels.
#pragma once
#include <type_traits>

#include "svtme_device.h"

namespace svtme {

typedef unsigned long long u64;

// compile-time loop: f(std::integral_constant<int, I>) for I in [B, E); indices stay constant expressions, so
// register arrays indexed by them never become runtime-indexed
template <int B, int E, class F>
__device__ __forceinline__ void static_for(F &&f) {
    if constexpr (B < E) {
        f(std::integral_constant<int, B>{});
        static_for<B + 1, E>(f);
    }
}

// get_scaled_picture_distance, Source/Lib/Encoder/Codec/EbMotionEstimation.c:1276-1280
__device__ __forceinline__ int scaled_dist(int dist) { return (uint16_t)(((dist * 5) / 8) + ((dist % 8) ? 1 : 0)); }

// A reference takes part in zz-SAD / pre-HME / HME unless it is a list-1 reference of a base-layer picture
// (`temporal_layer_index > 0 || list_index == 0`, :1786, :1936, :2024, :2087, :2148, :2325).
template <class P>
__device__ __forceinline__ bool ref_searched(const P &p, int l) { return p.temporal_layer > 0 || l == 0; }

// Pixel origin of 64x64 block `b64` of the picture: the block row through the picture's magic number (one scalar multiply
// instead of a division by a number only known at run time).
template <class P>
__device__ __forceinline__ void b64_origin(const P &p, int b64, int &b64x, int &b64y) {
    const int by = p.b64_w == 1 ? b64 : (int)__umulhi((uint32_t)b64, p.b64_w_magic);
    b64x = (b64 - by * p.b64_w) * 64;
    b64y = by * 64;
}

// The per-axis window correction all stages share (:875-907, :976-1009, :1077-1105, :1441-1460, :1502-1520,
// :1638-1682): a window starting below `lo` is shifted, not shrunk (the size correction is computed from the
// already corrected origin and is therefore zero); at the far side the origin is clamped to dim-1 and the size cut
// so that the window ends at dim, but never below 1. int16_t arithmetic as in the reference.
// svtme_picture::b64_step: the first pass's bypass_blk_step (firstpass.c:1282-1285) leaves out every block whose column or row
// index is not a multiple of the step; nothing is computed for those.
template <class P>
__device__ __forceinline__ bool b64_left_out(const P &p, int b64) {
    const int s = p.b64_step;
    return s > 1 && ((b64 % p.b64_w) % s != 0 || (b64 / p.b64_w) % s != 0);
}

__device__ __forceinline__ void clip_axis(int o, int16_t &org, int16_t &size, int lo, int dim) {
    if (o + org < lo) {
        org  = (int16_t)(lo - o);
        size = (int16_t)(size - (lo - (o + org)));
    }
    if (o + org > dim - 1) org = (int16_t)(org - ((o + org) - (dim - 1)));
    if (o + org + size > dim) size = (int16_t)max(1, size - ((o + org + size) - dim));
}

// Orders the LDS traffic of one wave: what its lanes wrote before is what its lanes read after. A wave's LDS instructions
// execute in order, so nothing has to be waited for; the fence and the scheduling barrier only keep the compiler from moving
// accesses across this point. For code that runs as one wave among several of a workgroup, where __syncthreads() would
// make every wave of the workgroup meet.
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// lane ^ 1 and lane ^ 2 within a quad, as DPP quad_perm moves (no LDS traffic)
__device__ __forceinline__ uint32_t quad_xor1(uint32_t v) { return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0xB1, 0xF, 0xF, true); }
__device__ __forceinline__ uint32_t quad_xor2(uint32_t v) { return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x4E, 0xF, 0xF, true); }

// Workgroup id -> (b64, item) such that all `items_per_b64` items of a b64 are dispatched next to each other and - with
// workgroups handed to the 8 XCDs round robin - on the same XCD, whose L2 then serves the b64's source block and
// overlapping reference windows to all of them. The grid holds ceil(n_b64 / 8) * 8 * items_per_b64 workgroups;
// returns false for the padding ones.
__device__ __forceinline__ bool b64_major_item(int wg, int n_b64, int items_per_b64, int &b64, int &item) {
    const int per_group = 8 * items_per_b64;
    const int group = wg / per_group, rem = wg - group * per_group;
    item = rem >> 3;
    b64  = group * 8 + (rem & 7);
    return b64 < n_b64;
}

// Workgroup id -> (b64, item) with one band of ceil(n_b64 / 8) consecutive b64s per XCD (workgroups go to the XCDs round
// robin, so id & 7 is the XCD): an XCD's L2 then only sees the rows of the reference planes its band needs, instead of
// all of every plane. For kernels whose items all cost the same. The grid holds 8 * ceil(n_b64 / 8) * items_per_b64
// workgroups; returns false for the padding ones.
__device__ __forceinline__ bool b64_band_item(int wg, int n_b64, int items_per_b64, int &b64, int &item) {
    const int band = (n_b64 + 7) >> 3, xcd = wg & 7, slot = wg >> 3;
    const int bl = slot / items_per_b64;
    item = slot - bl * items_per_b64;
    b64  = xcd * band + bl;
    return b64 < n_b64 && bl < band;
}

// Adds n to operation counter k of this workgroup's partial-sum slot.
template <class P>
__device__ __forceinline__ void count_ops(const P &p, int k, unsigned long long n) {
    atomicAdd(&p.ops[((blockIdx.x + 5 * blockIdx.y) & (SVTME_OPS_SLOTS - 1)) * 8 + k], n);
}

// 4 bytes at an arbitrary byte address, from two aligned dword loads.
__device__ __forceinline__ uint32_t load_u32_any(const uint8_t *p) {
    const uintptr_t a  = reinterpret_cast<uintptr_t>(p);
    const uint32_t *q  = reinterpret_cast<const uint32_t *>(a & ~(uintptr_t)3);
    const uint32_t  lo = q[0], hi = q[1];
    return __builtin_amdgcn_alignbyte(hi, lo, (uint32_t)(a & 3));
}

// floor(i / d) = magic_div(i, d, div_magic(d)) for i * d < 2^32 (here: indices below 2^16, divisors below 2^8);
// d == 1 has no 32-bit magic number
// (a 64-bit division costs about a hundred instructions on the device: the magic numbers of all divisors below 256
// come from a table in constant memory instead, one scalar load)
struct DivMagicTable {
    uint32_t m[256];
    constexpr DivMagicTable() : m() {
        m[0] = 0;
        for (uint32_t d = 1; d < 256; ++d) m[d] = (uint32_t)((0x100000000ull + d - 1) / d);
    }
};
__constant__ const DivMagicTable kDivMagic{};
__device__ __forceinline__ uint32_t div_magic(uint32_t d) {
    return d < 256 ? kDivMagic.m[d] : (uint32_t)((0x100000000ull + d - 1) / d);
}
__device__ __forceinline__ uint32_t magic_div(uint32_t i, uint32_t d, uint32_t magic) { return d == 1 ? i : __umulhi(i, magic); }

// Copies an (n_row x pd dwords) window from a plane (any byte alignment) into LDS rows of pd dwords, whole 64-lane wave,
// in 16-byte pieces (unaligned 16-byte loads); pd is a multiple of 4 and `win` 16-byte aligned. A row's pieces go to
// neighbouring lanes: 2^lg lanes per row (the power of two that holds the row's pd / 4 pieces; the lanes beyond them
// idle), 64 >> lg rows per trip, so a lane keeps its piece column and walks down the rows: per piece a row number, its
// clamp, and two 24-bit multiply-adds (no division, no 64-bit address arithmetic: a 32-bit offset from the uniform base).
// Rows past the end repeat the last row (same value into the same slot), so nothing branches. NB loads are in flight per
// lane before the first is stored. Reads pd * 4 bytes per row.
// gap_shift: one empty 16-byte piece goes in front of every 2^gap_shift-th row (row r starts at dword
// r * pd + 4 * (r >> gap_shift)), so that lanes which walk row groups 2^gap_shift rows apart do not all start in the same
// bank when 2^gap_shift * pd is a multiple of 32 dwords; 31: none.
template <int NB, int NT = 64>
__device__ __forceinline__ void stage_window(uint32_t *win, const uint8_t *g, int stride, int pd, int n_row, int gap_shift = 31) {
    static_assert(NT == 64, "one wave stages");
    const int      pr = pd >> 2, lg = pr <= 2 ? 1 : (pr <= 4 ? 2 : (pr <= 8 ? 3 : (pr <= 16 ? 4 : 5)));
    const int      lane = threadIdx.x & 63, c = lane & ((1 << lg) - 1), r0 = lane >> lg, per_trip = NT >> lg;
    uint4         *dst = reinterpret_cast<uint4 *>(win);
    if (c < pr) {
#pragma unroll 1
        for (int base = r0; base < n_row; base += per_trip * NB) {
            uint4    v[NB];
            uint32_t at[NB];
#pragma unroll
            for (int k = 0; k < NB; ++k) {
                const uint32_t r = (uint32_t)min(base + per_trip * k, n_row - 1);
                __builtin_memcpy(&v[k], g + (uint32_t)(__umul24(r, (uint32_t)stride) + 16u * (uint32_t)c), 16);
                at[k] = __umul24(r, (uint32_t)pr) + (uint32_t)c + (r >> gap_shift);
            }
#pragma unroll
            for (int k = 0; k < NB; ++k) dst[at[k]] = v[k];
        }
    }
}

__device__ __forceinline__ uint32_t wave_sum(uint32_t v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
    return v;
}

// Minimum of a 64-bit key over the wave, the same value in every lane: DPP lane exchanges only (no LDS round trips). The
// four exchanges inside a row of 16 lanes leave the row's minimum in each of its lanes; row_bcast:15 / row_bcast:31 carry
// it on to the following rows, so that lane 63 holds the wave's, which every lane then reads.
__device__ __forceinline__ u64 wave_min_u64(u64 v) {
    uint32_t lo = (uint32_t)v, hi = (uint32_t)(v >> 32);
#define SVTME_MIN_STEP(CTRL, ROWS)                                                                          \
    {                                                                                                       \
        const uint32_t olo = (uint32_t)__builtin_amdgcn_update_dpp((int)lo, (int)lo, CTRL, ROWS, 0xF, false); \
        const uint32_t ohi = (uint32_t)__builtin_amdgcn_update_dpp((int)hi, (int)hi, CTRL, ROWS, 0xF, false); \
        const bool     lt  = ohi < hi || (ohi == hi && olo < lo);                                           \
        lo = lt ? olo : lo;                                                                                 \
        hi = lt ? ohi : hi;                                                                                 \
    }
    SVTME_MIN_STEP(0xB1, 0xF) // quad_perm [1, 0, 3, 2]
    SVTME_MIN_STEP(0x4E, 0xF) // quad_perm [2, 3, 0, 1]
    SVTME_MIN_STEP(0x141, 0xF) // row_half_mirror
    SVTME_MIN_STEP(0x140, 0xF) // row_mirror
    SVTME_MIN_STEP(0x142, 0xA) // row_bcast:15 into rows 1 and 3
    SVTME_MIN_STEP(0x143, 0xC) // row_bcast:31 into rows 2 and 3
#undef SVTME_MIN_STEP
    lo = (uint32_t)__builtin_amdgcn_readlane((int)lo, 63);
    hi = (uint32_t)__builtin_amdgcn_readlane((int)hi, 63);
    return ((u64)hi << 32) | lo;
}

// The same for a 32-bit key: a v_min_u32 with a DPP operand per step.
__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v) {
#define SVTME_MIN_STEP(CTRL, ROWS) v = min(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, CTRL, ROWS, 0xF, false));
    SVTME_MIN_STEP(0xB1, 0xF) // quad_perm [1, 0, 3, 2]
    SVTME_MIN_STEP(0x4E, 0xF) // quad_perm [2, 3, 0, 1]
    SVTME_MIN_STEP(0x141, 0xF) // row_half_mirror
    SVTME_MIN_STEP(0x140, 0xF) // row_mirror
    SVTME_MIN_STEP(0x142, 0xA) // row_bcast:15 into rows 1 and 3
    SVTME_MIN_STEP(0x143, 0xC) // row_bcast:31 into rows 2 and 3
#undef SVTME_MIN_STEP
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}

// The same for a single wave and a block of at most 64 x 32 sampled rows, every other plane row (the zero-MV and
// search-centre checks of a 64x64 block): lane = one dword column of 8 sampled rows, all loads in flight together, no
// LDS. Returns the sum in every lane.
__device__ __forceinline__ uint32_t wave_block_sad_sub2(const DevPlane &cur, const DevPlane &ref, int ox, int oy, int dx, int dy,
                                                        int bw, int rows) {
    const int lane = threadIdx.x & 63, d = lane & 15, bwd = bw >> 2;
    uint32_t  sv[8], fv[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int  r  = (lane >> 4) + 4 * k;
        const bool on = d < bwd && r < rows;
        const int  rc = on ? r : 0, dc = on ? d : 0; // idle lanes repeat a valid address
        sv[k] = *reinterpret_cast<const uint32_t *>(cur.pix + (ptrdiff_t)(oy + 2 * rc) * cur.stride + ox + 4 * dc);
        __builtin_memcpy(&fv[k], ref.pix + (ptrdiff_t)(oy + dy + 2 * rc) * ref.stride + ox + dx + 4 * dc, 4);
    }
    uint32_t acc = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const bool on = d < bwd && (lane >> 4) + 4 * k < rows;
        acc = __builtin_amdgcn_sad_u8(sv[k], on ? fv[k] : sv[k], acc);
    }
    return wave_sum(acc);
}

// SAD of the (bw x rows) block of `cur` at (ox, oy) against `ref` at (ox + dx, oy + dy), every `step`-th row:
// svt_nxm_sad_kernel as get_zz_sad (:1720-1742) and check_00_center (:1170-1243) call it (stride << 1, height >> 1).
// Whole workgroup takes part; `red` is one LDS word; returns the same sum in every thread. bw is a multiple of 4.
__device__ uint32_t wg_block_sad(const DevPlane &cur, const DevPlane &ref, int ox, int oy, int dx, int dy, int bw,
                                 int rows, int step, uint32_t *red) {
    const int bwd = bw >> 2;
    const int n   = rows * bwd;
    if (threadIdx.x == 0) *red = 0;
    __syncthreads();
    uint32_t acc = 0;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const int      r = i / bwd, d = i - r * bwd;
        const uint32_t s = *reinterpret_cast<const uint32_t *>(cur.pix + (ptrdiff_t)(oy + r * step) * cur.stride + ox + 4 * d);
        const uint32_t f = load_u32_any(ref.pix + (ptrdiff_t)(oy + dy + r * step) * ref.stride + ox + dx + 4 * d);
        acc = __builtin_amdgcn_sad_u8(s, f, acc);
    }
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0 && acc) atomicAdd(red, acc);
    __syncthreads();
    const uint32_t total = *red;
    __syncthreads();
    return total;
}

} // namespace svtme
