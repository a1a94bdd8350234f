/*
 * Open-loop motion estimation of one picture: one 256-thread workgroup per 64x64 LCU.
 *
 * Replaces, for every LCU of the picture, MotionEstimateLcu
 * (reference Source/Lib/Codec/EbMotionEstimation.c:3671-4450) and the LCU staging
 * loop of MotionEstimationKernel (Codec/EbMotionEstimationProcess.c:706-780):
 *
 *   per list:  TestSearchAreaBounds (:3363)  -> HME level 0/1/2 (:2012,2194,2315)
 *              -> EbHevcCheckZeroZeroCenter (:2946) -> FullPelSearch_LCU (:586)
 *              -> SuPelEnable (:3035) -> EbHevcHalfPelSearch_LCU (:1036)
 *              -> QuarterPelSearch_LCU (:1623)
 *   then:      EbHevcBiPredictionSearch (:2870) and the candidate sort (:4321-4440).
 *
 * Mapping to CDNA4:
 *   - the source LCU (4 KiB), its even rows at 1/4 and 1/16 resolution and all
 *     running best-SAD/MV state live in LDS for the lifetime of the workgroup;
 *   - SADs are v_sad_u8 on packed dwords (4 samples per lane-op); reference
 *     samples are read with unaligned dword loads from the padded planes (the
 *     overlapping windows of neighbouring search positions hit in L1/L2);
 *   - "first minimum in raster order" argmins are packed (sad,index) keys
 *     reduced with LDS atomic min, so the tie rules of the C code hold without
 *     any serial scan (64x64 uses the '<=' rule of
 *     GetEightHorizontalSearchPointResults_32x32_64x64, C_DEFAULT/EbComputeSAD_C.c:439);
 *   - the half-pel planes b/h/j come from prep_kernels.hip (whole picture, once)
 *     instead of being re-interpolated per LCU and list.
 * All arithmetic is integer; results are bit-exact with the C_DEFAULT path.
 */
#include <mutex>
#include "svt_amd_internal.h"

#define NT 256
#define LCU 64
#ifndef ME_MIN_WAVES_PER_SIMD
#define ME_MIN_WAVES_PER_SIMD 3 /* LDS (static + windows ~ 53 KB at cfg2) admits 3 workgroups per CU */
#endif
#ifndef ME_HME_WAVES_PER_SIMD
/* 4 workgroups per CU = a 128-VGPR budget: the HME kernel needs 119 and then has NO private segment.  At 6 (80 VGPRs) it spilled
 * 6 VGPRs + 11 SGPRs into a 128 B/lane scratch frame, and the PMC pass showed the whole frame going to memory and back:
 * 8.4 GB written + 8.4 GB fetched per 64-picture 4K batch = 2/3 of the kernels' HBM traffic (profiles/r02_c). */
#define ME_HME_WAVES_PER_SIMD 4
#endif
#define MAX_SAD_VALUE (64 * 64 * 255)
#define COST_PRECISION 8
#define MD_SHIFT 23
#define MD_OFFSET (1u << 22)

typedef uint32_t __attribute__((aligned(1))) u32u;

__device__ __forceinline__ uint32_t ld4(const uint8_t *p) { return *(const u32u *)p; }
__device__ __forceinline__ uint32_t sad4(uint32_t a, uint32_t b, uint32_t acc) { return __builtin_amdgcn_sad_u8(a, b, acc); }
__device__ __forceinline__ uint32_t avg4(uint32_t a, uint32_t b) { return __builtin_amdgcn_lerp(a, b, 0x01010101u); }
__device__ __forceinline__ int imin(int a, int b) { return a < b ? a : b; }
__device__ __forceinline__ int imax(int a, int b) { return a > b ? a : b; }
__device__ __forceinline__ int mvx(uint32_t mv) { return (int)(int16_t)(mv & 0xffff); }
__device__ __forceinline__ int mvy(uint32_t mv) { return (int)(int16_t)(mv >> 16); }
__device__ __forceinline__ uint32_t mvpack(int x, int y) { return ((uint32_t)(uint16_t)y << 16) | (uint16_t)x; }

/* Cross-lane sums and minima on the DPP path (a modifier of a VALU instruction, a few cycles) instead of __shfl_xor, which the compiler turns into ds_bpermute_b32 - an LDS
 * crossbar round trip of ~100 cycles per step, six dependent steps per wave reduction.  quad_perm [1,0,3,2] / [2,3,0,1] = the xor-1 / xor-2 partners; row_half_mirror pairs
 * lane i with 7 - i of its group of eight, row_mirror with 15 - i of its row of sixteen: after the quad steps every lane of a quad holds the quad's value, so a mirror step
 * completes the next power of two; the four rows of a wave are combined through v_readlane. */
#define ME_DPP(v, ctrl) ((uint32_t)__builtin_amdgcn_update_dpp(0, (int)(v), ctrl, 0xF, 0xF, true))
template <int GROUP> /* sum over aligned groups of GROUP lanes (1, 2, 4, 8, 16), the total in every lane of the group */
__device__ __forceinline__ uint32_t group_sum(uint32_t v)
{
    if (GROUP >= 2)
        v += ME_DPP(v, 0xB1);
    if (GROUP >= 4)
        v += ME_DPP(v, 0x4E);
    if (GROUP >= 8)
        v += ME_DPP(v, 0x141); /* row_half_mirror */
    if (GROUP >= 16)
        v += ME_DPP(v, 0x140); /* row_mirror */
    return v;
}
/* the same for a group size known at run time (wave-uniform): 1 << lg lanes, lg in 0 .. 6; beyond a row of sixteen one crossbar step each */
__device__ __forceinline__ uint32_t group_sum_rt(uint32_t v, int lg)
{
    if (lg >= 1)
        v += ME_DPP(v, 0xB1);
    if (lg >= 2)
        v += ME_DPP(v, 0x4E);
    if (lg >= 3)
        v += ME_DPP(v, 0x141);
    if (lg >= 4)
        v += ME_DPP(v, 0x140);
    if (lg >= 5)
        v += __shfl_xor(v, 16);
    if (lg >= 6)
        v += __shfl_xor(v, 32);
    return v;
}
__device__ __forceinline__ uint32_t wave_sum(uint32_t v) /* wave-uniform result */
{
    v = group_sum<16>(v);
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 0) + (uint32_t)__builtin_amdgcn_readlane((int)v, 16) + (uint32_t)__builtin_amdgcn_readlane((int)v, 32) +
           (uint32_t)__builtin_amdgcn_readlane((int)v, 48);
}
__device__ __forceinline__ unsigned long long wave_min64(unsigned long long v) /* wave-uniform result */
{
    uint32_t lo = (uint32_t)v, hi = (uint32_t)(v >> 32);
#pragma unroll
    for (int st = 0; st < 4; st++) {
        const int ctrl = st == 0 ? 0xB1 : st == 1 ? 0x4E : st == 2 ? 0x141 : 0x140;
        uint32_t plo, phi;
        if (st == 0)
            plo = ME_DPP(lo, 0xB1), phi = ME_DPP(hi, 0xB1);
        else if (st == 1)
            plo = ME_DPP(lo, 0x4E), phi = ME_DPP(hi, 0x4E);
        else if (st == 2)
            plo = ME_DPP(lo, 0x141), phi = ME_DPP(hi, 0x141);
        else
            plo = ME_DPP(lo, 0x140), phi = ME_DPP(hi, 0x140);
        (void)ctrl;
        const bool less = phi < hi || (phi == hi && plo < lo);
        lo = less ? plo : lo, hi = less ? phi : hi;
    }
    unsigned long long m = ~0ull;
#pragma unroll
    for (int r = 0; r < 64; r += 16) {
        const unsigned long long w = ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)hi, r) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)lo, r);
        m = w < m ? w : m;
    }
    return m;
}

/* SAD of one row of w samples (w even; the tail dword is masked) */
__device__ __forceinline__ uint32_t row_sad(const uint8_t *a, const uint8_t *b, int w)
{
    uint32_t s = 0;
    int x = 0;
    for (; x + 4 <= w; x += 4)
        s = sad4(ld4(a + x), ld4(b + x), s);
    if (x < w) {
        const uint32_t m = (1u << (8 * (w - x))) - 1u;
        s = sad4(ld4(a + x) & m, ld4(b + x) & m, s);
    }
    return s;
}

/* N consecutive dwords from an UNALIGNED LDS byte address, fetched as N+1 aligned dwords and
 * funnel-shifted with v_alignbyte_b32.  (Unaligned ds_read_b32 works on gfx950 but stalls the
 * LDS pipe: SQ_LDS_UNALIGNED_STALL was 75 % of all LDS cycles before this - profiles/r01.) */
template <int N>
__device__ __forceinline__ void lds_ld_unaligned(const uint8_t *p, uint32_t (&out)[N])
{
    const uint32_t sh = (uint32_t)(uintptr_t)p & 3u;
    const uint32_t *q = (const uint32_t *)(p - sh);
    uint32_t prev = q[0];
#pragma unroll
    for (int i = 0; i < N; i++) {
        const uint32_t nx = q[i + 1];
        out[i] = __builtin_amdgcn_alignbyte(nx, prev, sh);
        prev = nx;
    }
}

/* row SAD, a = aligned LDS source row, b = unaligned LDS window row, w samples (w even) */
__device__ __forceinline__ uint32_t row_sad_lds(const uint8_t *a, const uint8_t *b, int w)
{
    uint32_t s = 0;
    int x = 0;
    for (; x + 16 <= w; x += 16) {
        uint32_t r[4];
        lds_ld_unaligned<4>(b + x, r);
        const uint4 sv = *(const uint4 *)(a + x);
        s = sad4(sv.x, r[0], s), s = sad4(sv.y, r[1], s), s = sad4(sv.z, r[2], s), s = sad4(sv.w, r[3], s);
    }
    for (; x + 4 <= w; x += 4) {
        uint32_t r[1];
        lds_ld_unaligned<1>(b + x, r);
        s = sad4(*(const uint32_t *)(a + x), r[0], s);
    }
    if (x < w) {
        uint32_t r[1];
        lds_ld_unaligned<1>(b + x, r);
        const uint32_t m = (1u << (8 * (w - x))) - 1u;
        s = sad4(*(const uint32_t *)(a + x) & m, r[0] & m, s);
    }
    return s;
}

__device__ __forceinline__ uint32_t ssd4(uint32_t a, uint32_t b)
{
    uint32_t s = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int d = (int)((a >> (8 * i)) & 255) - (int)((b >> (8 * i)) & 255);
        s += (uint32_t)(d * d);
    }
    return s;
}

/* A rectangle of a reference plane staged in LDS (dynamic pool).  Rows start on 16-byte
 * boundaries of the plane so the staging loads are aligned dwordx4 and coalesced. */
struct LWin {
    const uint8_t *p; /* LDS address of plane sample (x0, y0) */
    int x0, y0, stride;
};
extern __shared__ __attribute__((aligned(16))) uint8_t g_pool[];

/* row pitch of a staged window covering plane columns [x0, x1): rows start on a 16-byte boundary of the plane.
 * odd != 0 rounds the pitch to an ODD number of 16-byte units: rows then start on 8 different bank phases instead of
 * 1-2 (a pitch of 64 B puts every other row on the same LDS banks), used for the HME windows whose work items walk
 * down the rows. */
__device__ __forceinline__ int win_pitch(int x0, int x1, int odd)
{
    int wa = ((x1 - (x0 & ~15)) + 15) & ~15;
    if (odd && !((wa >> 4) & 1))
        wa += 16;
    return wa;
}
__device__ __forceinline__ const uint8_t *wat(const LWin &w, int x, int y)
{
    return w.p + (y - w.y0) * w.stride + (x - w.x0);
}
/* stage plane samples [x0,x1) x [y0,y1) at LDS offset `off` of the pool; returns the new offset */
__device__ __forceinline__ int load_window(LWin &w, int off, const uint8_t *plane, int pitch, int x0, int y0, int x1,
                                           int y1, int t, int odd = 0)
{
    const int xa = x0 & ~15, wa = win_pitch(x0, x1, odd), n16 = wa >> 4, rows = y1 - y0;
    uint8_t *dst = g_pool + off;
    w.p = dst, w.x0 = xa, w.y0 = y0, w.stride = wa;
    for (int i = t; i < rows * n16; i += NT) {
        const int r = i / n16, c = i - r * n16;
        *(uint4 *)(dst + r * wa + c * 16) = *(const uint4 *)(plane + (ptrdiff_t)(y0 + r) * pitch + xa + c * 16);
    }
    return off + rows * wa;
}

/* Window whose first column is the plane column x0 ITSELF (any alignment): unaligned 16-byte global loads, 16-byte LDS stores */
typedef uint4 __attribute__((aligned(1))) u128u_w;
__device__ __forceinline__ int load_window_at(LWin &w, int off, const uint8_t *plane, int pitch, int x0, int y0, int x1, int y1, int t,
                                              int odd = 0)
{
    int wa = ((x1 - x0) + 15) & ~15;
    if (odd && !((wa >> 4) & 1))
        wa += 16;
    const int n16 = wa >> 4, rows = y1 - y0;
    uint8_t *dst = g_pool + off;
    w.p = dst, w.x0 = x0, w.y0 = y0, w.stride = wa;
    for (int i = t; i < rows * n16; i += NT) {
        const int r = i / n16, c = i - r * n16;
        *(uint4 *)(dst + r * wa + c * 16) = *(const u128u_w *)(plane + (ptrdiff_t)(y0 + r) * pitch + x0 + c * 16);
    }
    return off + rows * wa;
}
__device__ __forceinline__ uint32_t umin32(uint32_t a, uint32_t b) { return a < b ? a : b; }
__device__ __forceinline__ uint4 ldu16(const uint8_t *p) /* one unaligned 16-byte global load */
{
    const uint4 v = *(const u128u_w *)p;
    return v;
}

/* Same staging through the LDS-DMA path (global_load_lds_dwordx4): the data never passes through VGPRs and the
 * issuing wave does not wait for it, so the loads overlap with whatever is computed next.  One wave instruction
 * moves 64 consecutive 16-byte items to 1 KiB of consecutive LDS (M0 = LDS base of the chunk, lane i lands at
 * base + 16 i), which is exactly the row-major window layout.  The caller must execute
 * `__builtin_amdgcn_s_waitcnt(0)` + `__syncthreads()` before anybody reads the window. */
__device__ __forceinline__ int load_window_async(LWin &w, int off, const uint8_t *plane, int pitch, int x0, int y0, int x1,
                                                 int y1, int t, int odd = 0)
{
    const int xa = x0 & ~15, wa = win_pitch(x0, x1, odd), n16 = wa >> 4, rows = y1 - y0, total = rows * n16;
    uint8_t *dst = g_pool + off;
    w.p = dst, w.x0 = xa, w.y0 = y0, w.stride = wa;
    const int lane = t & 63;
    for (int c0 = (t >> 6) * 64; c0 < total; c0 += NT) { /* chunk of 64 items per wave instruction */
        const int i = c0 + lane;
        if (i < total) {
            const int r = i / n16, c = i - r * n16;
            __builtin_amdgcn_global_load_lds(
                (const __attribute__((address_space(1))) void *)(plane + (ptrdiff_t)(y0 + r) * pitch + xa + c * 16),
                (__attribute__((address_space(3))) void *)(dst + c0 * 16), 16, 0, 0);
        }
    }
    return off + rows * wa;
}

/* The three half-pel windows of the search kernel share their geometry: one pass derives (row, 16-byte column) of an item once and issues the three LDS-DMA loads */
__device__ __forceinline__ int load_windows3_async(LWin &wa_, LWin &wb_, LWin &wc_, int off, const uint8_t *pa, const uint8_t *pb, const uint8_t *pc, int pitch, int x0, int y0,
                                                   int x1, int y1, int t)
{
    const int xa = x0 & ~15, wa = win_pitch(x0, x1, 1), n16 = wa >> 4, rows = y1 - y0, total = rows * n16, bytes = rows * wa;
    uint8_t *dst = g_pool + off;
    wa_.p = dst, wb_.p = dst + bytes, wc_.p = dst + 2 * bytes;
    wa_.x0 = wb_.x0 = wc_.x0 = xa, wa_.y0 = wb_.y0 = wc_.y0 = y0, wa_.stride = wb_.stride = wc_.stride = wa;
    const int lane = t & 63;
    const uint32_t rc = (1u << 20) / (uint32_t)n16 + 1u; /* items < 2^12, n16 <= 16: (i * rc) >> 20 == i / n16 */
    for (int c0 = (t >> 6) * 64; c0 < total; c0 += NT) {
        const int i = c0 + lane;
        if (i < total) {
            const int r = (int)(((uint32_t)i * rc) >> 20), c = i - r * n16;
            const ptrdiff_t o = (ptrdiff_t)(y0 + r) * pitch + xa + c * 16;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(pa + o), (__attribute__((address_space(3))) void *)(dst + c0 * 16), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(pb + o), (__attribute__((address_space(3))) void *)(dst + bytes + c0 * 16), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(pc + o), (__attribute__((address_space(3))) void *)(dst + 2 * bytes + c0 * 16), 16, 0, 0);
        }
    }
    return off + 3 * bytes;
}

/* Z-order <-> raster (tab32x32 / tab8x8, EbMotionEstimation.c:98-102) */
__constant__ uint8_t c_tab16[16] = {0, 1, 4, 5, 2, 3, 6, 7, 8, 9, 12, 13, 10, 11, 14, 15};
__constant__ uint8_t c_tab8[64] = {0,  1,  4,  5,  16, 17, 20, 21, 2,  3,  6,  7,  18, 19, 22, 23,
                                   8,  9,  12, 13, 24, 25, 28, 29, 10, 11, 14, 15, 26, 27, 30, 31,
                                   32, 33, 36, 37, 48, 49, 52, 53, 34, 35, 38, 39, 50, 51, 54, 55,
                                   40, 41, 44, 45, 56, 57, 60, 61, 42, 43, 46, 47, 58, 59, 62, 63};

/* quarter-pel source pairs: SetQuarterPelRefinementInputsOnTheFly (EbMotionEstimation.c:1532-1621)
 * as {plane(0 F,1 B,2 H,3 J), dx, dy} relative to the integer anchor; order L,R,T,B,TL,TR,BR,BL */
struct QSrc { int8_t plane, dx, dy; };
__constant__ QSrc c_qtab[4][8][2] = {
    {{{1, 0, 0}, {0, 0, 0}}, {{0, 0, 0}, {1, 1, 0}}, {{2, 0, 0}, {0, 0, 0}}, {{0, 0, 0}, {2, 0, 1}},
     {{1, 0, 0}, {2, 0, 0}}, {{2, 0, 0}, {1, 1, 0}}, {{2, 0, 1}, {1, 1, 0}}, {{1, 0, 0}, {2, 0, 1}}},
    {{{0, -1, 0}, {1, 0, 0}}, {{1, 0, 0}, {0, 0, 0}}, {{3, 0, 0}, {1, 0, 0}}, {{1, 0, 0}, {3, 0, 1}},
     {{2, -1, 0}, {1, 0, 0}}, {{1, 0, 0}, {2, 0, 0}}, {{1, 0, 0}, {2, 0, 1}}, {{2, -1, 1}, {1, 0, 0}}},
    {{{3, 0, 0}, {2, 0, 0}}, {{2, 0, 0}, {3, 1, 0}}, {{0, 0, -1}, {2, 0, 0}}, {{2, 0, 0}, {0, 0, 0}},
     {{1, 0, -1}, {2, 0, 0}}, {{2, 0, 0}, {1, 1, -1}}, {{2, 0, 0}, {1, 1, 0}}, {{1, 0, 0}, {2, 0, 0}}},
    {{{2, -1, 0}, {3, 0, 0}}, {{3, 0, 0}, {2, 0, 0}}, {{1, 0, -1}, {3, 0, 0}}, {{3, 0, 0}, {1, 0, 0}},
     {{2, -1, 0}, {1, 0, -1}}, {{1, 0, -1}, {2, 0, 0}}, {{1, 0, 0}, {2, 0, 0}}, {{2, -1, 0}, {1, 0, 0}}}};

/* half-pel direction codes (EbMotionEstimation.c:58-65) */
enum { D_TL = 0, D_T = 1, D_TR = 2, D_R = 3, D_BR = 4, D_B = 5, D_BL = 6, D_L = 7 };

/* geometry of internal (Z-order) PU index n */
__device__ __forceinline__ void pu_geom_z(int n, int &x, int &y, int &sz)
{
    if (n == 0) {
        x = 0, y = 0, sz = 64;
    } else if (n < 5) {
        const int k = n - 1;
        x = (k & 1) << 5, y = (k >> 1) << 5, sz = 32;
    } else if (n < 21) {
        const int z = n - 5;
        x = ((z & 1) | (((z >> 2) & 1) << 1)) << 4;
        y = (((z >> 1) & 1) | (((z >> 3) & 1) << 1)) << 4;
        sz = 16;
    } else {
        const int z = n - 21;
        x = ((z & 1) | (((z >> 2) & 1) << 1) | (((z >> 4) & 1) << 2)) << 3;
        y = (((z >> 1) & 1) | (((z >> 3) & 1) << 1) | (((z >> 5) & 1) << 2)) << 3;
        sz = 8;
    }
}

/* the clamp sequence shared by every search (e.g. EbMotionEstimation.c:2064-2101) */
__device__ __forceinline__ void clamp_area(int origin, int pad, int pic, int &o, int &size)
{
    if (origin + o < -pad)
        o = -pad - origin;
    if (origin + o > pic - 1)
        o = o - ((origin + o) - (pic - 1));
    if (origin + o + size > pic)
        size = imax(1, size - ((origin + o + size) - pic));
}
__device__ __forceinline__ int clamp_center(int origin, int c, int pad, int pic)
{
    if (origin + c < -pad)
        c = -pad - origin;
    if (origin + c > pic - 1)
        c = c - ((origin + c) - (pic - 1));
    return c;
}

/* MeEbHevcGetMvdFractionBits (Codec/EbMdRateEstimation.c:172-236) */
__device__ uint32_t mvd_fraction_bits(int mvdX, int mvdY, const uint32_t *bits)
{
    const uint32_t ax = (uint32_t)abs(mvdX), ay = (uint32_t)abs(mvdY);
    const uint32_t xn = mvdX != 0, yn = mvdY != 0, xg = ax > 1, yg = ay > 1;
    uint32_t n = bits[xn] + bits[yn + (2u << xn)];
    if (xn)
        n += bits[xg + 6];
    if (yn)
        n += bits[yg + 6 + (2u << xg)];
    for (int k = 0; k < 2; k++) {
        const uint32_t a = k ? ay : ax, nz = k ? yn : xn, gt = k ? yg : xg;
        if (!nz)
            continue;
        if (gt) {
            uint32_t symbol = a - 2, count = 1, bn = 0;
            while (symbol >= (1u << count)) {
                bn++;
                symbol -= 1u << count;
                count++;
            }
            n += (bn + 1 + count) * 32768u;
        }
        n += 32768u;
    }
    return n;
}

/* LDS state shared by both kernels of the split (static allocation) */
struct MeShared {
    uint8_t src[LCU * LCU + 16];   /* source LCU rows (padded-plane content)              */
    uint8_t qsrc[32 * 16 + 16];    /* 1/4 LCU, even rows                                   */
    uint8_t ssrc[16 * 8 + 16];     /* 1/16 LCU, even rows                                  */
    unsigned long long hkey[4];    /* HME per-quadrant minima                              */
    uint32_t acc[8];               /* LCU-level SAD accumulators                           */
    int16_t hx[3][2][2], hy[3][2][2]; /* HME centres per level [w][h]                      */
    unsigned long long hs[3][2][2];
    int qp[4][4];                  /* HME quadrant search areas {origin x, origin y, width, height}         */
    int cand[6][4];                /* LCU-SAD candidates {unclamped x, y, clamped x, y}                     */
    uint32_t mvd_bits[12];         /* P.mvd_bits (dynamically indexed by the rate function)                 */
};
/* LDS state of the search kernel only; lives at the start of the dynamic pool (the staged windows follow), so the
 * HME kernel does not pay for it and fits ~3x more workgroups per CU */
struct MeSearch {
    union {                        /* the full-pel SAD trees and the sub-pel accumulators are never live together */
        struct {
            uint32_t sad32[256][4]; /* 32x32 SADs of a chunk of search rows, [position][quadrant]: the 64x64 sums */
        };
        struct {
            uint32_t dist[85][8];  /* sub-pel distortions (search metric)                  */
            uint32_t dsad[85][8];  /* full SAD at the same positions (SSD search only)     */
        };
    };
    uint32_t key[85];              /* packed (sad,index) minima, PUs 1..84                 */
    unsigned long long key64;      /* 64x64 */
    uint32_t best_sad[2][85], best_mv[2][85], best_ssd[2][85];
    uint8_t dir[2][85];
    uint32_t bipred[85];
    int e32, e16, e8, eq;
    int sums[9];                   /* SuPelEnable: per tier {sum mvx, sum mvy, sum sad} */
};
#define ME_SEARCH_BYTES ((int)((sizeof(MeSearch) + 15) & ~(size_t)15))
/* what the HME kernel hands to the search kernel (and to itself for list 1: trap A21) per LCU */
struct MeCarry {
    int16_t hx[3][2][2], hy[3][2][2];
    unsigned long long hs[3][2][2];
    int32_t cx[2], cy[2];          /* search centre after EbHevcCheckZeroZeroCenter, per list */
    int32_t hme_init_done, pad;
};
static_assert(sizeof(MeCarry) <= 192, "context.hip reserves 192 bytes of carry per LCU");

/* ------------------------------------------------------------------------- */

/* Sub-sampled LCU SAD against the reference at the displacements S.cand[c][2..3], c in [first, ncand):
 * NxMSadKernel(lcuSrcPtr, stride<<1, ref, stride<<1, lcuHeight>>1, lcuWidth) per candidate.
 * The caller must have synchronised S.cand; S.acc[c] for c < first is left untouched. */
typedef uint4 __attribute__((aligned(1))) u128u;
__device__ void lcu_sads(MeShared &S, const uint8_t *ref, int pitch, int ox, int oy, int lw, int lh, int ncand, int t,
                         int first = 0)
{
    if (t < 8 && t >= first)
        S.acc[t] = 0;
    __syncthreads();
    const int rows = lh >> 1; /* <= 32 */
    if (lw == LCU) {
        /* item = (candidate, row, 16-sample quarter): one unaligned 16-byte global load + 4 v_sad_u8; the 128 items
         * of a candidate fill two whole waves, so the candidate sum is a wave reduction + one LDS atomic per wave */
        for (int i = first * 128 + t; i < ncand * 128; i += NT) {
            const int c = i >> 7, r = (i >> 2) & 31, qx = (i & 3) << 4;
            uint32_t s = 0;
            if (r < rows) {
                const uint4 a = *(const uint4 *)&S.src[(2 * r) * LCU + qx];
                const uint4 b = *(const u128u *)(ref + (ptrdiff_t)(oy + S.cand[c][3] + 2 * r) * pitch + ox + S.cand[c][2] + qx);
                s = sad4(a.x, b.x, s), s = sad4(a.y, b.y, s), s = sad4(a.z, b.z, s), s = sad4(a.w, b.w, s);
            }
            s = wave_sum(s);
            if ((t & 63) == 0)
                atomicAdd(&S.acc[c], s);
        }
    } else {
        for (int i = first * 32 + t; i < ncand * 32; i += NT) {
            const int c = i >> 5, r = i & 31;
            if (r < rows) {
                const uint32_t s = row_sad(&S.src[(2 * r) * LCU],
                                           ref + (ptrdiff_t)(oy + S.cand[c][3] + 2 * r) * pitch + ox + S.cand[c][2], lw);
                atomicAdd(&S.acc[c], s);
            }
        }
    }
    __syncthreads();
}

/* exact p / d for p < 40000, d < 300 with r = (1<<24)/d + 1 (checked exhaustively) */
__device__ __forceinline__ uint32_t fastdiv_recip(uint32_t d) { return (1u << 24) / d + 1u; }
__device__ __forceinline__ uint32_t fastdiv(uint32_t p, uint32_t r) { return (uint32_t)(((unsigned long long)p * r) >> 24); }

/* One HME pass over up to four quadrants of one pyramid level.  Each quadrant's search
 * window is first staged in LDS (coalesced 16-byte loads).  A search position is then
 * owned by a group of `rows` adjacent lanes (8 / 16 / 32 at level 0 / 1 / 2), one block
 * row per lane; the row SADs are summed with a butterfly over the group, so all four
 * waves stay busy even for the 32-position level-1 searches.
 * SadLoopKernel semantics (C_DEFAULT/EbComputeSAD_C.c:170): raster scan, strict '<'. */
__device__ void hme_pass(MeShared &S, int level, const uint8_t *refplane, int pitch, int bx0, int by0, int bw,
                         int rows, int nq, int t)
{
    if (t < 4)
        S.hkey[t] = ~0ull;
    {
        int off = 0;
        for (int q = 0; q < nq; q++) {
            const int qx = S.qp[q][0], qy = S.qp[q][1], qw = S.qp[q][2], qh = S.qp[q][3];
            LWin w;
            off = load_window(w, off, refplane, pitch, bx0 + qx, by0 + qy, bx0 + qx + qw + bw,
                              by0 + qy + qh + 2 * (rows - 1) + 1, t);
        }
    }
    __syncthreads();
    const uint8_t *src = level == 0 ? S.ssrc : level == 1 ? S.qsrc : S.src;
    const int sstride = level == 0 ? 16 : level == 1 ? 32 : 2 * LCU;
    const int lg = level == 0 ? 3 : level == 1 ? 4 : 5; /* log2(rows) */
    const int y = t & (rows - 1);
    const uint8_t *srow = src + y * sstride;
    int off = 0;
    for (int q = 0; q < nq; q++) {
        const int qx = S.qp[q][0], qw = S.qp[q][2], qh = S.qp[q][3];
        /* same geometry as load_window computed above */
        const int x0 = bx0 + qx, xa = x0 & ~15, wstride = ((x0 + qw + bw - xa) + 15) & ~15;
        const int wrows = qh + 2 * (rows - 1) + 1;
        const uint8_t *wbase = g_pool + off + (x0 - xa) + y * 2 * wstride;
        off += wrows * wstride;
        const int npos = qw * qh, items = npos << lg;
        const uint32_t rc = fastdiv_recip((uint32_t)(qw > 0 ? qw : 1));
        unsigned long long best = ~0ull;
        for (int i0 = 0; i0 < items; i0 += NT) { /* uniform trip count: every lane joins the butterfly */
            const int p = (i0 + t) >> lg;
            uint32_t sv = 0;
            if (p < npos) {
                const int sy = (int)fastdiv((uint32_t)p, rc), sx = p - sy * qw;
                sv = row_sad_lds(srow, wbase + sy * wstride + sx, bw);
            }
            for (int o = rows >> 1; o > 0; o >>= 1)
                sv += __shfl_xor(sv, o);
            const unsigned long long k = p < npos ? (((unsigned long long)sv << 32) | (uint32_t)p) : ~0ull;
            best = k < best ? k : best;
        }
        best = wave_min64(best);
        if ((t & 63) == 0)
            atomicMin(&S.hkey[q], best);
    }
    __syncthreads();
}

/* v_qsad_pk_u16_u8: four SADs of the 4 source bytes against the 4 sliding byte windows of a 64-bit reference
 * word, accumulated into four packed u16 lanes (semantics checked on gfx950 by tools/qsad_probe.hip). */
__device__ __forceinline__ unsigned long long qsad(uint32_t ref_lo, uint32_t ref_hi, uint32_t src, unsigned long long acc)
{
    return __builtin_amdgcn_qsad_pk_u16_u8(((unsigned long long)ref_hi << 32) | ref_lo, src, acc);
}

/* HME pass, quad-SAD form (the fast path; hme_pass above remains for block widths that are not a multiple of 4).
 * A work item is (search row sy, aligned window dword m) = FOUR adjacent search positions; the lane walks the
 * block rows of its chunk, reading G+1 aligned window dwords and the G source dwords per row and issuing G
 * v_qsad_pk_u16_u8 - no unaligned access, no funnel shifts, 4 positions per instruction.  ROWS block rows are split
 * into C chunks on adjacent lanes (more parallelism for the small level-1/2 searches; rows-per-chunk x width <= 256
 * keeps the packed u16 sums exact).  Wave w serves quadrant w % nq.  Tie rule as hme_pass. */
template <int G, int ROWS, int C>
__device__ void hme_pass_q(MeShared &S, const uint8_t *src, int sstride, const uint8_t *refplane, int pitch, int bx0,
                           int by0, int nq, int t)
{
    if (t < 4)
        S.hkey[t] = ~0ull;
    {
        int off = 0;
        for (int q = 0; q < nq; q++) {
            const int qx = S.qp[q][0], qy = S.qp[q][1], qw = S.qp[q][2], qh = S.qp[q][3];
            LWin w;
            off = load_window_async(w, off, refplane, pitch, bx0 + qx, by0 + qy, bx0 + qx + qw + 4 * G,
                                    by0 + qy + qh + 2 * (ROWS - 1) + 1, t, 1);
        }
    }
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    const int wave = t >> 6, lane = t & 63;
    const int q = wave % nq, slot = wave / nq, nslots = (4 + nq - 1 - q) / nq; /* waves serving quadrant q */
    int off = 0;
    for (int k = 0; k < q; k++) {
        const int x0k = bx0 + S.qp[k][0];
        off += (S.qp[k][3] + 2 * (ROWS - 1) + 1) * win_pitch(x0k, x0k + S.qp[k][2] + 4 * G, 1);
    }
    const int qw = S.qp[q][2], qh = S.qp[q][3];
    const int x0 = bx0 + S.qp[q][0], xa = x0 & ~15, wstride = win_pitch(x0, x0 + qw + 4 * G, 1);
    const int bo0 = x0 - xa;                          /* window byte offset of search position sx = 0 */
    const int m0 = bo0 >> 2, mcount = ((bo0 + qw - 1) >> 2) - m0 + 1;
    const int items = qh * mcount * C;
    const uint32_t rc = fastdiv_recip((uint32_t)(mcount > 0 ? mcount : 1));
    constexpr int RPC = ROWS / C;                     /* block rows per chunk */
    unsigned long long best = ~0ull;
    for (int i0 = slot * 64; i0 < items; i0 += nslots * 64) { /* uniform per wave */
        const int li = i0 + lane, it = li / C, ch = li - it * C;
        const bool live = li < items;
        uint32_t sd[4] = {0, 0, 0, 0};
        int sy = 0, mi = 0;
        if (live) {
            sy = (int)fastdiv((uint32_t)it, rc), mi = it - sy * mcount;
            const uint32_t *wr = (const uint32_t *)(g_pool + off + (sy + 2 * ch * RPC) * wstride) + m0 + mi;
            const uint8_t *sr = src + ch * RPC * sstride;
            unsigned long long acc = 0;
#pragma unroll
            for (int r = 0; r < RPC; r++) {
                uint32_t d0 = wr[0]; /* five window dwords live at a time (a whole-row array costs 17 VGPRs at level 2) */
#pragma unroll
                for (int g4 = 0; g4 < G; g4 += 4) {
                    const uint32_t d1 = wr[g4 + 1], d2 = wr[g4 + 2], d3 = wr[g4 + 3], d4 = wr[g4 + 4];
                    const uint4 sv = *(const uint4 *)(sr + 4 * g4);
                    acc = qsad(d0, d1, sv.x, acc);
                    acc = qsad(d1, d2, sv.y, acc);
                    acc = qsad(d2, d3, sv.z, acc);
                    acc = qsad(d3, d4, sv.w, acc);
                    d0 = d4;
                }
                wr += (2 * wstride) >> 2;
                sr += sstride;
            }
            sd[0] = (uint32_t)(acc & 0xffff), sd[1] = (uint32_t)(acc >> 16) & 0xffff;
            sd[2] = (uint32_t)(acc >> 32) & 0xffff, sd[3] = (uint32_t)(acc >> 48);
        }
        if (C > 1) /* the chunks of a position sit on C adjacent lanes */
            sd[0] = group_sum<C>(sd[0]), sd[1] = group_sum<C>(sd[1]), sd[2] = group_sum<C>(sd[2]), sd[3] = group_sum<C>(sd[3]);
        if (live && ch == 0) {
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int sx = 4 * (m0 + mi) + k - bo0;
                if (sx >= 0 && sx < qw) {
                    const unsigned long long key = ((unsigned long long)sd[k] << 32) | (uint32_t)(sy * qw + sx);
                    best = key < best ? key : best;
                }
            }
        }
    }
    best = wave_min64(best);
    if (lane == 0 && best != ~0ull)
        atomicMin(&S.hkey[q], best);
    __syncthreads();
}

__device__ __forceinline__ int hme_l12_width(int w) { return (w < 8) ? 8 : (w & 7) ? w + (w - ((w >> 3) << 3)) : w; }

/* sub-pel row distortion: metric by fractionalSearchMethod; optional full SAD alongside */
__device__ __forceinline__ void row_metric(int method, uint32_t a, uint32_t b, uint32_t &d, uint32_t &sad)
{
    /* written with value selects: the two-branch form (sad = sad4(a, b, sad) / d = sad4(a, b, d)) was merged by the compiler into ONE
     * v_sad_u8 through a selected POINTER, which kept d and sad in scratch memory (12 B/lane private segment, profiles/r02_c) */
    const bool ssd = method == SVT_AMD_SSD_SEARCH;
    const uint32_t r = sad4(a, b, ssd ? sad : d);
    const uint32_t q = ssd ? ssd4(a, b) : 0u;
    sad = ssd ? r : sad;
    d = ssd ? d + q : r;
}

/* workgroup barrier that orders LDS traffic only (ds reads / writes / atomics): unlike __syncthreads() it does not wait for
 * outstanding vector-memory operations, so LDS-DMA loads issued earlier stay in flight across it */
#ifndef ME_EXP
#define ME_EXP 0
#endif
#define ME_BI_ITEMS 3 /* bi-prediction items a thread has in flight */
#define ME_F_ITEMS 3 /* 16-byte items of the F window a thread requests in the first round trip: 3 x 256 covers search areas up to 48 x 48 (+67) */
#define LDS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

/* optional phase profile: when the job carries a debug buffer, thread 0 of every workgroup
 * stores the shader clock at each phase boundary (svt_amd_debug_me_phase_profile) */
#define STAMP(i)                                                                      \
    do {                                                                              \
        if (J.dbg_clock && t == 0)                                                    \
            J.dbg_clock[((size_t)blockIdx.y * J.lcu_count + (lcu - lcu_begin)) * 16 + (i)] = __builtin_readcyclecounter(); \
    } while (0)

/* tier = 0 (64x64), 1 (32x32), 2 (16x16), 3 (8x8): closed forms instead of small private arrays, which the
 * compiler would place in scratch memory when indexed dynamically (scratch = VMEM latency + HBM write traffic) */
__device__ __forceinline__ int tier_first_of(int tier) { return tier == 0 ? 0 : tier == 1 ? 1 : tier == 2 ? 5 : 21; }
__device__ __forceinline__ int tier_cnt_of(int tier) { return 1 << (2 * tier); }
__device__ __forceinline__ int tier_sz_of(int tier) { return 64 >> tier; }
__device__ __forceinline__ int tier_lc_of(int tier) { return tier == 0 ? 5 : tier == 1 ? 3 : tier == 2 ? 1 : 0; }
__device__ __forceinline__ int pick4(int i, int a, int b, int c, int d) { return i == 0 ? a : i == 1 ? b : i == 2 ? c : d; }

/* grid = (max LCUs of any job, jobs): one workgroup per (picture, LCU).
 * The per-LCU chain is split in two kernels per reference list (the launcher runs them back to back):
 *   PHASE 0  "hme"     TestSearchAreaBounds + HME L0/L1/L2 + CheckZeroZeroCenter -> search centre (MeCarry)
 *   PHASE 1  "search"  full-pel 85-PU search + sub-pel refinement of that list; after the last list also
 *                      bi-prediction and the candidate records
 * The HME part needs ~15 KB of LDS and few registers, the search part ~55 KB: as separate kernels the
 * latency-bound HME phases run at ~3x the occupancy instead of inheriting the search kernel's footprint. */
template <int PHASE>
__global__ __launch_bounds__(NT, PHASE == 0 ? ME_HME_WAVES_PER_SIMD : ME_MIN_WAVES_PER_SIMD) void k_me(const MeJobDev *__restrict__ jobs, int list)
{
    __shared__ MeShared S;
    const MeJobDev &J = jobs[blockIdx.y];
    /* XCD-aware placement: workgroups are dealt round-robin to the 8 XCDs (gridDim.x is a multiple of 8), so
     * XCD x gets blockIdx.x = x, x+8, ...; give it a CONTIGUOUS run of LCUs - neighbouring LCUs share most of
     * their HME / search-window bytes and then hit in that XCD's L2 instead of refetching across XCDs. */
    const int per_xcd = (J.lcu_count + 7) >> 3;
    const int blk = (int)(blockIdx.x & 7) * per_xcd + (int)(blockIdx.x >> 3);
    if ((int)(blockIdx.x >> 3) >= per_xcd || blk >= J.lcu_count)
        return;
    const SvtAmdMeParams &P = J.P; /* stays in global memory: uniform fields come in through scalar loads; a private copy
                                     * would live in scratch because of the indexed arrays inside */
    if (list >= P.num_lists)
        return;
    const PicView cur = J.cur, ref0 = J.ref0, ref1 = J.ref1;
    SvtAmdMeLcuResult *__restrict__ out = J.out;
    const int lcu_begin = J.lcu_begin;
    const int t = threadIdx.x;
    const int W = P.luma_width, H = P.luma_height;
    const int wl = (W + LCU - 1) / LCU;
    const int lcu = lcu_begin + blk;
    const int ox = (lcu % wl) * LCU, oy = (lcu / wl) * LCU;
    const int lw = imin(LCU, W - ox), lh = imin(LCU, H - oy);
    const int pf = cur.pitch_full;
    const int method = P.fractional_search_method;
    MeCarry *__restrict__ carry = &J.carry[lcu];
    SvtAmdMeLcuResult *o = &out[lcu];
    MeSearch &B = *(MeSearch *)g_pool; /* PHASE 1 only */
    const PicView &R = list ? ref1 : ref0;
    (void)method;

    STAMP(PHASE == 0 ? 0 : 5);
    /* ---- stage the source LCU (EbMotionEstimationProcess.c:714-779); the search kernel does it together with its first reference window below ---- */
    int cx = 0, cy = 0;
    /* TestSearchAreaBounds' candidate c (zero, A, B, C, D, direct = list 0's 64x64 vector mirrored): unclamped and clamped displacement (EbMotionEstimation.c:3363-3665) */
    const uint32_t mv64_l0 = (PHASE == 0 && list) ? o->best_mv[0][0] : 0u; /* list 0's final 64x64 MV (direct candidate) */
    auto tsab_cand = [&](int c, int &ux, int &uy, int &kx, int &ky) {
        ux = c == 1 ? -(int)P.hme_l0_total_w : c == 2 ? (int)P.hme_l0_total_w : c == 5 ? 0 - (mvx(mv64_l0) >> 2) : 0;
        uy = c == 3 ? -(int)P.hme_l0_total_h : c == 4 ? (int)P.hme_l0_total_h : c == 5 ? 0 - (mvy(mv64_l0) >> 2) : 0;
        ux = (int16_t)ux, uy = (int16_t)uy;
        kx = c ? clamp_center(ox, ux, LCU - 1, W) : 0, ky = c ? clamp_center(oy, uy, LCU - 1, H) : 0;
    };
    /* whole-width LCUs: the candidates' sub-sampled LCU SADs need 16-byte reference rows that depend on nothing but the picture's controls (and, for the direct candidate,
     * one scalar load): requested TOGETHER with the source LCU - one trip to memory instead of two */
    const bool tsab_early = PHASE == 0 && (P.temporal_layer_index > 0 || list == 0) && P.update_hme_search_center && lw == LCU;
    const int tsab_nc = list == 1 ? 6 : 5;
    uint4 tsab_b[3];
    if (PHASE == 0) {
        uint32_t sv[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int i = t + k * NT, y = i >> 4, x = (i & 15) << 2;
            sv[k] = *(const uint32_t *)(cur.full + (ptrdiff_t)(oy + y) * pf + ox + x);
        }
        if (tsab_early) {
#pragma unroll
            for (int k = 0; k < 3; k++) { /* item = (candidate, row, 16-sample quarter), as lcu_sads deals them: candidate (t >> 7) + 2 k */
                const int i = t + k * NT, c = i >> 7, r = (i >> 2) & 31, qx = (i & 3) << 4;
                tsab_b[k] = make_uint4(0, 0, 0, 0);
                if (c < tsab_nc && r < (lh >> 1)) {
                    int ux, uy, kx, ky;
                    tsab_cand(c, ux, uy, kx, ky);
                    tsab_b[k] = ldu16(R.full + (ptrdiff_t)(oy + ky + 2 * r) * R.pitch_full + ox + kx + qx);
                }
            }
        }
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int i = t + k * NT, y = i >> 4, x = (i & 15) << 2;
            *(uint32_t *)&S.src[y * LCU + x] = sv[k];
        }
        if (t < 8)
            S.acc[t] = 0;
        if (t < 128) { /* 1/4: 16 even rows x 32 */
            const int y = t >> 3, x = (t & 7) << 2;
            *(uint32_t *)&S.qsrc[y * 32 + x] =
                ld4(cur.quarter + (ptrdiff_t)((oy >> 1) + 2 * y) * cur.pitch_quarter + (ox >> 1) + x);
        } else if (t < 160) { /* 1/16: 8 even rows x 16 */
            const int u = t - 128, y = u >> 2, x = (u & 3) << 2;
            *(uint32_t *)&S.ssrc[y * 16 + x] =
                ld4(cur.sixteenth + (ptrdiff_t)((oy >> 2) + 2 * y) * cur.pitch_sixteenth + (ox >> 2) + x);
        }
        if (t >= 64 && t < 76)
            S.mvd_bits[t - 64] = P.mvd_bits[t - 64];
        if (t < 12) { /* per-quadrant HME centres survive from list 0 to list 1 (trap A21) */
            (&S.hx[0][0][0])[t] = list ? (&carry->hx[0][0][0])[t] : (int16_t)0;
            (&S.hy[0][0][0])[t] = list ? (&carry->hy[0][0][0])[t] : (int16_t)0;
            (&S.hs[0][0][0])[t] = list ? (&carry->hs[0][0][0])[t] : 0ull;
        }
        __syncthreads();
    }

    LWin wF, wB, wH, wJ; /* LDS windows of the current list's reference planes */

    if (PHASE == 0) {
        int hme_init_done = list ? carry->hme_init_done : 0;
        int zero_sad_valid = 0;
        if (P.temporal_layer_index > 0 || list == 0) {
            STAMP(1);
            /* ---- TestSearchAreaBounds (EbMotionEstimation.c:3363-3665) ---- */
            if (P.update_hme_search_center) {
                const int nc = tsab_nc;
                if (t < 6) { /* candidate t: zero, A, B, C, D, direct (list-0 64x64 MV mirrored) */
                    int ux, uy, kx, ky;
                    tsab_cand(t, ux, uy, kx, ky);
                    S.cand[t][0] = ux, S.cand[t][1] = uy;
                    S.cand[t][2] = kx, S.cand[t][3] = ky;
                }
                if (tsab_early) { /* the reference rows came with the source LCU */
#pragma unroll
                    for (int k = 0; k < 3; k++) {
                        const int i = t + k * NT, c = i >> 7, r = (i >> 2) & 31, qx = (i & 3) << 4;
                        if (i < nc * 128) { /* whole waves: 128 items a candidate */
                            uint32_t sd = 0;
                            if (r < (lh >> 1)) {
                                const uint4 a = *(const uint4 *)&S.src[(2 * r) * LCU + qx], b = tsab_b[k];
                                sd = sad4(a.x, b.x, sd), sd = sad4(a.y, b.y, sd), sd = sad4(a.z, b.z, sd), sd = sad4(a.w, b.w, sd);
                            }
                            sd = wave_sum(sd);
                            if ((t & 63) == 0)
                                atomicAdd(&S.acc[c], sd);
                        }
                    }
                    __syncthreads();
                } else {
                    __syncthreads();
                    lcu_sads(S, R.full, R.pitch_full, ox, oy, lw, lh, nc, t);
                }
                zero_sad_valid = 1; /* S.acc[0] = SAD at (0,0) of this list's reference: reused by CheckZeroZeroCenter */
                /* tie order: zero, A, B, C, direct, D (:3634-3658); costs are sad << 9 */
                const uint32_t a0 = S.acc[0], a1 = S.acc[1], a2 = S.acc[2], a3 = S.acc[3], a4 = S.acc[4],
                               a5 = nc == 6 ? S.acc[5] : 0xffffffffu;
                uint32_t best = a0;
                best = a1 < best ? a1 : best, best = a2 < best ? a2 : best, best = a3 < best ? a3 : best;
                best = a4 < best ? a4 : best, best = a5 < best ? a5 : best;
                const int pick = best == a0 ? 0 : best == a1 ? 1 : best == a2 ? 2 : best == a3 ? 3 : best == a5 ? 5 : 4;
                cx = S.cand[pick][0], cy = S.cand[pick][1];
                __syncthreads(); /* S.acc / S.cand are reused below */
            }

            STAMP(2);
            /* ---- HME (EbMotionEstimation.c:3800-4069) ---- */
            if (P.enable_hme_flag && lh == LCU) {
                const int nw = P.num_hme_regions_w, nh = P.num_hme_regions_h;
                const int one = (P.one_quadrant_hme && !P.enable_hme_level1 && !P.enable_hme_level2);
                const int nq = nw * nh;
                const int qh_ = t < 4 ? (nw == 2 ? (t >> 1) : t) : 0, qw_ = t < 4 ? (nw == 2 ? (t & 1) : 0) : 0; /* q = h*nw + w */
                if (!hme_init_done) {
                    if (t < nq) {
                        const int sh0 = P.update_hme_search_center ? 2 : 0, sh1 = P.update_hme_search_center ? 1 : 0;
                        S.hx[0][qw_][qh_] = (int16_t)(cx >> sh0), S.hy[0][qw_][qh_] = (int16_t)(cy >> sh0);
                        S.hx[1][qw_][qh_] = (int16_t)(cx >> sh1), S.hy[1][qw_][qh_] = (int16_t)(cy >> sh1);
                        S.hx[2][qw_][qh_] = (int16_t)cx, S.hy[2][qw_][qh_] = (int16_t)cy;
                    }
                    hme_init_done = 1;
                    __syncthreads();
                }
                const uint32_t mx = P.hme_l0_mult_x, my = P.hme_l0_mult_y;
                if (P.enable_hme_level0) {
                    const int px16 = ox >> 2, py16 = oy >> 2;
                    const int pw16 = W >> 2, ph16 = H >> 2, pad16 = SVT_AMD_PAD_SIXTEENTH - 1;
                    const int nq0 = one ? 1 : nq;
                    if (t < nq0) { /* thread q derives quadrant q = (qw_, qh_) */
                        int sw, sh, so_x, so_y;
                        if (one) {
                            /* EbHevcHmeOneQuadrantLevel0 (:1847-2010) */
                            sw = (int16_t)((P.hme_l0_total_w * mx) / 100), sh = (int16_t)((P.hme_l0_total_h * my) / 100);
                            so_x = -(int)(int16_t)(sw >> 1) + (cx >> 2), so_y = -(int)(int16_t)(sh >> 1) + (cy >> 2);
                            clamp_area(px16, pad16, pw16, so_x, sw);
                            clamp_area(py16, pad16, ph16, so_y, sh);
                            if (sw & 15)
                                sw = (sw >> 4) << 4;
                        } else {
                            sw = (int16_t)(((qw_ ? P.hme_l0_w[1] : P.hme_l0_w[0]) * mx) / 100), sh = (int16_t)(((qh_ ? P.hme_l0_h[1] : P.hme_l0_h[0]) * my) / 100);
                            int dx = cx >> 2, dy = cy >> 2;
                            if (qw_)
                                dx += (int16_t)((P.hme_l0_w[0] * mx) / 100);
                            if (qh_)
                                dy += (int16_t)((P.hme_l0_h[0] * my) / 100);
                            so_x = (int16_t)(-(int)(int16_t)(((P.hme_l0_total_w * mx) / 100) >> 1) + dx);
                            so_y = (int16_t)(-(int)(int16_t)(((P.hme_l0_total_h * my) / 100) >> 1) + dy);
                            clamp_area(px16, pad16, pw16, so_x, sw);
                            clamp_area(py16, pad16, ph16, so_y, sh);
                        }
                        S.qp[t][0] = so_x, S.qp[t][1] = so_y, S.qp[t][2] = sw, S.qp[t][3] = sh;
                    }
                    __syncthreads();
                    if ((lw >> 2) == 16)
                        hme_pass_q<4, 8, 1>(S, S.ssrc, 16, R.sixteenth, R.pitch_sixteenth, px16, py16, nq0, t);
                    else
                        hme_pass(S, 0, R.sixteenth, R.pitch_sixteenth, px16, py16, lw >> 2, 8, nq0, t);
                    if (t < nq0) {
                        const unsigned long long k = S.hkey[t];
                        const int qx = S.qp[t][0], qy = S.qp[t][1], qwv = S.qp[t][2];
                        if (k != ~0ull) { /* an empty search leaves the centre untouched */
                            const int p = (int)(uint32_t)k, sy = p / qwv, sx = p - sy * qwv;
                            S.hx[0][qw_][qh_] = (int16_t)((int16_t)(sx + qx) * 4);
                            S.hy[0][qw_][qh_] = (int16_t)((int16_t)(sy + qy) * 4);
                            S.hs[0][qw_][qh_] = (k >> 32) * 2;
                        } else {
                            S.hs[0][qw_][qh_] = 0xffffffull * 2;
                            S.hx[0][qw_][qh_] = (int16_t)((int16_t)(S.hx[0][qw_][qh_] + qx) * 4);
                            S.hy[0][qw_][qh_] = (int16_t)((int16_t)(S.hy[0][qw_][qh_] + qy) * 4);
                        }
                    }
                    /* the next level's parameters are derived by the same thread from its own quadrant's result and
                     * are published by that level's barrier: a barrier here is only needed after the last level */
                    if (!P.enable_hme_level1 && !P.enable_hme_level2)
                        __syncthreads();
                }
                for (int lvl = 1; lvl <= 2; lvl++) {
                    if (!(lvl == 1 ? P.enable_hme_level1 : P.enable_hme_level2))
                        continue;
                    const int shf = 2 - lvl;
                    const int bx0 = ox >> shf, by0 = oy >> shf, pwl = W >> shf, phl = H >> shf;
                    const int padl = lvl == 2 ? LCU - 1 : SVT_AMD_PAD_QUARTER - 1;
                    if (t < nq) {
                        int sw = hme_l12_width((int16_t)(lvl == 1 ? (qw_ ? P.hme_l1_w[1] : P.hme_l1_w[0]) : (qw_ ? P.hme_l2_w[1] : P.hme_l2_w[0])));
                        int sh = (int16_t)(lvl == 1 ? (qh_ ? P.hme_l1_h[1] : P.hme_l1_h[0]) : (qh_ ? P.hme_l2_h[1] : P.hme_l2_h[0]));
                        const int pcx = lvl == 1 ? (S.hx[0][qw_][qh_] >> 1) : S.hx[1][qw_][qh_];
                        const int pcy = lvl == 1 ? (S.hy[0][qw_][qh_] >> 1) : S.hy[1][qw_][qh_];
                        int so_x = (int16_t)(-(sw >> 1) + pcx), so_y = (int16_t)(-(sh >> 1) + pcy);
                        clamp_area(bx0, padl, pwl, so_x, sw);
                        clamp_area(by0, padl, phl, so_y, sh);
                        S.qp[t][0] = so_x, S.qp[t][1] = so_y, S.qp[t][2] = sw, S.qp[t][3] = sh;
                    }
                    __syncthreads();
                    if (lw == LCU && lvl == 1)
                        hme_pass_q<8, 16, 4>(S, S.qsrc, 32, R.quarter, R.pitch_quarter, bx0, by0, nq, t);
                    else if (lw == LCU)
                        hme_pass_q<16, 32, 8>(S, S.src, 2 * LCU, R.full, R.pitch_full, bx0, by0, nq, t);
                    else
                        hme_pass(S, lvl, lvl == 1 ? R.quarter : R.full, lvl == 1 ? R.pitch_quarter : R.pitch_full, bx0, by0,
                                 lw >> shf, lvl == 1 ? 16 : 32, nq, t);
                    if (t < nq) {
                        const unsigned long long k = S.hkey[t];
                        const int qx = S.qp[t][0], qy = S.qp[t][1], qwv = S.qp[t][2];
                        const int p = (int)(uint32_t)k, sy = p / qwv, sx = p - sy * qwv;
                        S.hx[lvl][qw_][qh_] = (int16_t)((int16_t)(sx + qx) * (1 << shf));
                        S.hy[lvl][qw_][qh_] = (int16_t)((int16_t)(sy + qy) * (1 << shf));
                        S.hs[lvl][qw_][qh_] = (k >> 32) * 2;
                    }
                    if (lvl == 2 || !P.enable_hme_level2)
                        __syncthreads();
                }
                /* centre selection (:3958-4069) - every thread evaluates the same scalars */
                {
                    int hx = 0, hy = 0;
                    unsigned long long hsv = 0;
                    const int l0only = P.enable_hme_level0 && !P.enable_hme_level1 && !P.enable_hme_level2;
                    const int sel = P.enable_hme_level2 ? 2 : (P.enable_hme_level1 ? 1 : (l0only ? 0 : -1));
                    if (sel >= 0) {
                        hx = S.hx[sel][0][0], hy = S.hy[sel][0][0], hsv = S.hs[sel][0][0];
                        if (!(sel == 0 && P.one_quadrant_hme))
                            for (int h = 0; h < nh; h++)
                                for (int w = (h == 0 ? 1 : 0); w < nw; w++)
                                    if (S.hs[sel][w][h] < hsv)
                                        hx = S.hx[sel][w][h], hy = S.hy[sel][w][h], hsv = S.hs[sel][w][h];
                    }
                    if (sel == 2 && P.ref_pocs_equal && list == 1 && nh * nw > 1) {
                        /* second-best L2 quadrant for list 1 (:4034-4064); the sort is
                         * in place and visible to later code, so thread 0 performs it */
                        __syncthreads();
                        if (t == 0) {
                            const int total = nh * nw;
                            for (int q = 0; q < total - 1; q++)
                                for (int r = q + 1; r < total; r++) {
                                    const int qa = q / nw, qb = q % nw, ra = r / nw, rb = r % nw;
                                    if (S.hs[2][qa][qb] > S.hs[2][ra][rb]) {
                                        const int16_t tx = S.hx[2][qa][qb], ty = S.hy[2][qa][qb];
                                        const unsigned long long ts = S.hs[2][qa][qb];
                                        S.hx[2][qa][qb] = S.hx[2][ra][rb], S.hy[2][qa][qb] = S.hy[2][ra][rb], S.hs[2][qa][qb] = S.hs[2][ra][rb];
                                        S.hx[2][ra][rb] = tx, S.hy[2][ra][rb] = ty, S.hs[2][ra][rb] = ts;
                                    }
                                }
                        }
                        __syncthreads();
                        hx = S.hx[2][0][1], hy = S.hy[2][0][1];
                    }
                    cx = hx, cy = hy;
                }
            }
        }
        const int hcx = cx, hcy = cy; /* the HME centre before the zero-centre check: reported in the record */

        STAMP(3);
        /* ---- EbHevcCheckZeroZeroCenter (:2946-3034) ---- */
        if (cx != 0 || cy != 0) {
            if (P.update_hme_search_center) {
                cx = clamp_center(ox, cx, LCU - 1, W);
                cy = clamp_center(oy, cy, LCU - 1, H);
            }
            if (t == 0) {
                S.cand[0][2] = 0, S.cand[0][3] = 0;
                S.cand[1][2] = cx, S.cand[1][3] = cy;
            }
            __syncthreads();
            lcu_sads(S, R.full, R.pitch_full, ox, oy, lw, lh, 2, t, zero_sad_valid);
            const uint32_t zeroSad = S.acc[0] << 1, hmeSad = S.acc[1] << 1;
            const unsigned long long zeroCost = (unsigned long long)zeroSad << COST_PRECISION;
            const uint32_t rate = mvd_fraction_bits(abs(cx << 2), abs(cy << 2), S.mvd_bits);
            const unsigned long long hmeCost = (unsigned long long)(uint32_t)(hmeSad << COST_PRECISION) +
                                               ((((unsigned long long)P.lambda * rate) + MD_OFFSET) >> MD_SHIFT);
            if (zeroCost <= hmeCost)
                cx = 0, cy = 0;
            __syncthreads();
        }

        /* hand-over to the search kernel (and to the HME kernel of list 1) */
        if (t == 0) {
            carry->cx[list] = cx, carry->cy[list] = cy;
            carry->hme_init_done = hme_init_done;
            o->hme_center_x[list] = (int16_t)hcx, o->hme_center_y[list] = (int16_t)hcy;
            if (P.num_lists == 1)
                o->hme_center_x[1] = 0, o->hme_center_y[1] = 0;
        }
        if (t < 12) {
            (&carry->hx[0][0][0])[t] = (&S.hx[0][0][0])[t];
            (&carry->hy[0][0][0])[t] = (&S.hy[0][0][0])[t];
            (&carry->hs[0][0][0])[t] = (&S.hs[0][0][0])[t];
        }
        STAMP(4);
    } else { /* PHASE 1 */
        /* ---- search area (:4072-4200); unrestricted MVs ---- */
        cx = carry->cx[list], cy = carry->cy[list];
        int saw = imin(P.search_area_width, 127), sah = imin(P.search_area_height, 127);
        int sox = cx - (saw >> 1), soy = cy - (sah >> 1);
        clamp_area(ox, LCU - 1, W, sox, saw);
        clamp_area(oy, LCU - 1, H, soy, sah);
        /* ONE round trip to memory for everything the search starts from: the source LCU (16 bytes a thread), list 0's results (list 1 only) and the F window
         * (x in [sox-4, sox+saw+63+2), y in [soy-2, soy+sah+63+2): the search region plus what the sub-pel stages reach; its first column FOUR samples left of search
         * position 0, not at a 16-byte boundary of the plane: every block's position 0 then sits on an LDS dword and a search row of saw positions is exactly
         * saw / 4 quad-SAD items) are requested back to back and land in LDS together.  (Staged one after the other - the window needs the search centre of the
         * carry record, which is a scalar load - they cost two.) */
        const int wx0 = ox + sox - 2, wy0 = oy + soy - 2, wx1 = ox + sox + saw + 65, wy1 = oy + soy + sah + 65;
        int win_off;
        {
            const uint4 sv = *(const u128u_w *)(cur.full + (ptrdiff_t)(oy + (t >> 2)) * pf + ox + ((t & 3) << 4));
            uint32_t i_sad = 0, i_mv = 0;
            if (list && t < 85)
                i_sad = o->best_sad[0][t], i_mv = o->best_mv[0][t];
            int wa = ((wx1 - (wx0 - 2)) + 15) & ~15;
            if (!((wa >> 4) & 1))
                wa += 16;
            const int n16 = wa >> 4, rows = wy1 - wy0, total = rows * n16;
            const uint32_t rc = (1u << 20) / (uint32_t)n16 + 1u;
            uint8_t *dst = g_pool + ME_SEARCH_BYTES;
            wF.p = dst, wF.x0 = wx0 - 2, wF.y0 = wy0, wF.stride = wa;
            uint4 fw[ME_F_ITEMS];
#pragma unroll
            for (int k = 0; k < ME_F_ITEMS; k++) {
                const int i = t + k * NT;
                if (i < total) {
                    const int r = (int)(((uint32_t)i * rc) >> 20), c = i - r * n16;
                    fw[k] = ldu16(R.full + (ptrdiff_t)(wy0 + r) * R.pitch_full + (wx0 - 2) + c * 16);
                }
            }
            *(uint4 *)&S.src[(t >> 2) * LCU + ((t & 3) << 4)] = sv;
            if (t < 85) {
                B.best_sad[list][t] = 0, B.best_mv[list][t] = 0, B.best_ssd[list][t] = 0, B.dir[list][t] = 0;
                /* the other list: zero for a one-list picture, list 0's final result for list 1 */
                B.best_sad[1 - list][t] = i_sad, B.best_mv[1 - list][t] = i_mv;
                B.best_ssd[1 - list][t] = 0, B.dir[1 - list][t] = 0;
                B.bipred[t] = 0;
                B.key[t] = 0xffffffffu;
            }
            if (t == 0)
                B.key64 = ~0ull;
#pragma unroll
            for (int k = 0; k < ME_F_ITEMS; k++) {
                const int i = t + k * NT;
                if (i < total)
                    *(uint4 *)(dst + i * 16) = fw[k];
            }
            for (int i = t + ME_F_ITEMS * NT; i < total; i += NT) { /* search areas beyond 48 x 48: the rest of the window the plain way */
                const int r = (int)(((uint32_t)i * rc) >> 20), c = i - r * n16;
                *(uint4 *)(dst + i * 16) = ldu16(R.full + (ptrdiff_t)(wy0 + r) * R.pitch_full + (wx0 - 2) + c * 16);
            }
            win_off = ME_SEARCH_BYTES + rows * wa;
        }
        LDS_BARRIER(); /* source, F window, search state are in LDS (their global loads are waited for by the stores that carry them) */

        STAMP(6);
        /* ---- FullPelSearch_LCU (:586-633) ---- */
        {
            const int mult8 = saw & ~7;
            /* A WAVE owns one 32x32 quadrant; lane = (position group g = lane >> 4, 8x8 block blk = lane & 15 in Z order inside the
             * quadrant).  The whole SAD tree of the quadrant then lives in the wave: the four 8x8 blocks of a 16x16 are the four
             * lanes of a quad, the four 16x16 of the 32x32 are the four quads of a 16-lane row, so 16x16 and 32x32 sums are DPP
             * adds on the packed quad-SAD accumulators and the running minima stay in registers - no per-position SADs in LDS, no
             * atomics, no barrier inside the search.  Only the 64x64 SAD needs the other three waves: every wave leaves its 32x32
             * SADs of a chunk of search rows in LDS (<= 256 positions) and one pass sums them. */
            const int q = t >> 6, lane = t & 63, grp = lane >> 4, blk = lane & 15;
            const int bx = ((q & 1) << 5) + (((blk & 1) | (((blk >> 2) & 1) << 1)) << 3);
            const int by = ((q >> 1) << 5) + ((((blk >> 1) & 1) | (((blk >> 3) & 1) << 1)) << 3);
            uint32_t s0[4], s1[4];
#pragma unroll
            for (int r = 0; r < 4; r++) {
                s0[r] = *(const uint32_t *)&S.src[(by + 2 * r) * LCU + bx];
                s1[r] = *(const uint32_t *)&S.src[(by + 2 * r) * LCU + bx + 4];
            }
            STAMP(13);
            /* the half-pel planes (same region less the two extra columns) are first read by the sub-pel stages: fetch them by LDS-DMA UNDER the full-pel search
             * (waited for at "sub-pel windows landed" below).  Every barrier between here and there must be LDS_BARRIER: a __syncthreads() carries a vmcnt(0)
             * and would park the workgroup until the three windows have landed, which is what the first version of this stage did */
            load_windows3_async(wB, wH, wJ, win_off, R.hp_b, R.hp_h, R.hp_j, R.pitch_full, wx0, wy0, wx1, wy1, t);
#if ME_EXP >= 10
            STAMP(15);
#endif
            const uint32_t *rb4 = (const uint32_t *)wat(wF, ox + bx + sox, oy + by + soy); /* dword-aligned by construction */
            const int fs4 = wF.stride >> 2;
            const int mcount = (saw + 3) >> 2;                          /* quad-SAD items per search row */
            const int rows_per_chunk = imax(1, imin(sah, 256 / saw));   /* <= 256 positions of 32x32 SADs in LDS at a time */
            const uint32_t rcm = fastdiv_recip((uint32_t)mcount), rcs = fastdiv_recip((uint32_t)saw);
            uint32_t best8 = 0xffffffffu, best16 = 0xffffffffu, best32 = 0xffffffffu;
            const int slot = blk & 3; /* the position of an item this lane speaks for in the 16x16 / 32x32 sums */
            const bool fast = (saw & 3) == 0;
            const int sh16 = (slot & 1) ? 0 : 16;
            const int q4 = 4 / mcount, r4 = 4 - q4 * mcount;
            uint32_t b8[4] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu}, b16 = 0xffffffffu;
            for (int row0 = 0; row0 < sah; row0 += rows_per_chunk) {
                const int nrows = imin(rows_per_chunk, sah - row0), items = nrows * mcount, pbase = row0 * saw;
                if (fast) {
                    /* search rows of a multiple of four positions (every row exactly mcount items, raster index of an item's first position = 4 x its index): the
                     * running minima keep (sad << 16 | item index) PER POSITION of the item - built from the packed halves of the quad-SAD result with one v_lshl_or /
                     * v_and_or each, no unpacking, no per-position validity test - and are turned into (sad, raster index) keys once after the search.  A dead
                     * lane group (item index past the end) carries all-ones halves: its keys lose against every real one. */
                    /* Item addresses advance incrementally (four items further = q4 rows and r4 columns, one wrap at most): no multiplication inside the loop.  The
                     * twelve window dwords of the NEXT item are fetched before the current one is evaluated - one LDS round trip per iteration, overlapped, instead of
                     * one per block row. */
                    int mi = (int)grp, ry = 0;
                    while (mi >= mcount)
                        mi -= mcount, ry++;
                    int ro = (row0 + ry) * fs4 + mi; /* dword offset of the item's first window dword from rb4 */
                    uint32_t itg = (uint32_t)(row0 * mcount) + (uint32_t)grp;
                    uint32_t cur[12], nxt[12];
#pragma unroll
                    for (int rr = 0; rr < 4; rr++)
                        cur[3 * rr] = rb4[ro + 2 * rr * fs4], cur[3 * rr + 1] = rb4[ro + 2 * rr * fs4 + 1], cur[3 * rr + 2] = rb4[ro + 2 * rr * fs4 + 2];
                    for (int it0 = 0; it0 < items; it0 += 4) {
                        const bool live = it0 + (int)grp < items;
                        { /* the next item of this lane group (its own again when there is none: a valid address) */
                            int nmi = mi + r4, nro = ro + q4 * fs4 + r4;
                            if (nmi >= mcount)
                                nmi -= mcount, nro += fs4 - mcount;
                            const bool more = it0 + 4 + (int)grp < items;
                            mi = more ? nmi : mi, ro = more ? nro : ro;
#pragma unroll
                            for (int rr = 0; rr < 4; rr++)
                                nxt[3 * rr] = rb4[ro + 2 * rr * fs4], nxt[3 * rr + 1] = rb4[ro + 2 * rr * fs4 + 1], nxt[3 * rr + 2] = rb4[ro + 2 * rr * fs4 + 2];
                        }
                        unsigned long long acc = 0;
#pragma unroll
                        for (int rr = 0; rr < 4; rr++) {
                            acc = qsad(cur[3 * rr], cur[3 * rr + 1], s0[rr], acc);
                            acc = qsad(cur[3 * rr + 1], cur[3 * rr + 2], s1[rr], acc);
                        }
#pragma unroll
                        for (int k = 0; k < 12; k++)
                            cur[k] = nxt[k];
                        uint32_t lo = live ? (uint32_t)acc : 0xffffffffu, hi = live ? (uint32_t)(acc >> 32) : 0xffffffffu;
                        b8[0] = umin32(b8[0], (lo << 16) | itg), b8[1] = umin32(b8[1], (lo & 0xffff0000u) | itg);
                        b8[2] = umin32(b8[2], (hi << 16) | itg), b8[3] = umin32(b8[3], (hi & 0xffff0000u) | itg);
                        lo += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)lo, 0xB1, 0xF, 0xF, true); /* quad_perm [1,0,3,2] */
                        hi += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)hi, 0xB1, 0xF, 0xF, true);
                        lo += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)lo, 0x4E, 0xF, 0xF, true); /* quad_perm [2,3,0,1] */
                        hi += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)hi, 0x4E, 0xF, 0xF, true);
                        const uint32_t sel = slot < 2 ? lo : hi;
                        const uint32_t k16 = live ? (((sel << sh16) & 0xffff0000u) | itg) : 0xffffffffu;
                        b16 = umin32(b16, k16);
                        uint32_t v = (sel >> (16 - sh16)) & 0xffffu;
                        v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x124, 0xF, 0xF, true); /* row_ror:4 */
                        v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x128, 0xF, 0xF, true); /* row_ror:8 */
                        const uint32_t pk = 4u * itg + (uint32_t)slot;
                        const uint32_t k32 = live ? ((v << 14) | pk) : 0xffffffffu;
                        best32 = umin32(best32, k32);
                        if (live && blk < 4)
                            B.sad32[(int)pk - pbase][q] = v;
                        itg += 4;
                    }
                } else
                for (int it0 = 0; it0 < items; it0 += 4) { /* uniform trip count: the DPP sums need whole rows of lanes */
                    const int it = it0 + grp;
                    const bool live = it < items;
                    const int ry = live ? (int)fastdiv((uint32_t)it, rcm) : 0, mi = live ? it - ry * mcount : 0, sy = row0 + ry;
                    const uint32_t *r = rb4 + sy * fs4 + mi;
                    unsigned long long acc = 0;
#pragma unroll
                    for (int rr = 0; rr < 4; rr++) {
                        const uint32_t d0 = r[0], d1 = r[1], d2 = r[2];
                        acc = qsad(d0, d1, s0[rr], acc);
                        acc = qsad(d1, d2, s1[rr], acc);
                        r += 2 * fs4;
                    }
                    const int p0 = sy * saw + 4 * mi; /* raster index of the item's first position */
                    uint32_t lo = (uint32_t)acc, hi = (uint32_t)(acc >> 32);
                    /* 8x8: this lane's block, four positions */
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        const uint32_t sv = ((k < 2 ? lo : hi) >> (16 * (k & 1))) & 0xffffu;
                        const uint32_t key = (live && 4 * mi + k < saw) ? ((sv << 14) | (uint32_t)(p0 + k)) : 0xffffffffu;
                        best8 = key < best8 ? key : best8;
                    }
                    /* 16x16: packed sums over the quad (4 x 16320 fits a u16 lane, no carry between the halves) */
                    lo += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)lo, 0xB1, 0xF, 0xF, true); /* quad_perm [1,0,3,2] */
                    hi += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)hi, 0xB1, 0xF, 0xF, true);
                    lo += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)lo, 0x4E, 0xF, 0xF, true); /* quad_perm [2,3,0,1] */
                    hi += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)hi, 0x4E, 0xF, 0xF, true);
                    uint32_t v = ((slot < 2 ? lo : hi) >> (16 * (slot & 1))) & 0xffffu; /* 16x16 SAD at position p0 + slot */
                    const bool valid = live && 4 * mi + slot < saw;
                    const uint32_t pk = (uint32_t)(p0 + slot);
                    {
                        const uint32_t key = valid ? ((v << 14) | pk) : 0xffffffffu;
                        best16 = key < best16 ? key : best16;
                    }
                    /* 32x32: the same position's 16x16 SADs of the four quads of the row */
                    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x124, 0xF, 0xF, true); /* row_ror:4 */
                    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x128, 0xF, 0xF, true); /* row_ror:8 */
                    {
                        const uint32_t key = valid ? ((v << 14) | pk) : 0xffffffffu;
                        best32 = key < best32 ? key : best32;
                    }
                    if (valid && blk < 4)
                        B.sad32[(int)pk - pbase][q] = v;
                }
#if ME_EXP >= 10
                STAMP(0);
#endif
                LDS_BARRIER();
                { /* 64x64: '<=' inside complete groups of 8, '<' in the tail */
                    const int p = pbase + t;
                    unsigned long long k64 = ~0ull;
                    if (t < nrows * saw) {
                        const uint4 sq = *(const uint4 *)&B.sad32[t][0];
                        const uint32_t s = sq.x + sq.y + sq.z + sq.w;
                        const int sy = (int)fastdiv((uint32_t)p, rcs), sx = p - sy * saw;
                        const uint32_t code = (sx < mult8) ? (uint32_t)(16383 - p) : (0x4000u | (uint32_t)p);
                        k64 = ((unsigned long long)s << 15) | code;
                    }
                    if (q * 64 < nrows * saw) { /* the wave's minimum, then ONE LDS atomic per wave (64 lanes on one address serialise) */
                        k64 = wave_min64(k64);
                        if (lane == 0)
                            atomicMin(&B.key64, k64);
                    }
                }
                LDS_BARRIER();
            }
#if ME_EXP >= 10
            STAMP(1);
#endif
            if (fast) { /* (sad << 16 | item index) -> (sad << 14 | raster index): the order the reference's strict '<' scan keeps */
#pragma unroll
                for (int k = 0; k < 4; k++)
                    if (b8[k] != 0xffffffffu)
                        best8 = umin32(best8, ((b8[k] >> 16) << 14) | (4u * (b8[k] & 0xffffu) + (uint32_t)k));
                if (b16 != 0xffffffffu)
                    best16 = ((b16 >> 16) << 14) | (4u * (b16 & 0xffffu) + (uint32_t)slot);
            }
            /* minima of the position groups (and, for 16x16 / 32x32, of the lanes that spoke for different positions) */
            best8 = umin32(best8, (uint32_t)__shfl_xor((int)best8, 16));
            best8 = umin32(best8, (uint32_t)__shfl_xor((int)best8, 32));
#pragma unroll
            for (int o = 1; o <= 2; o <<= 1)
                best16 = umin32(best16, (uint32_t)__shfl_xor((int)best16, o));
            best16 = umin32(best16, (uint32_t)__shfl_xor((int)best16, 16));
            best16 = umin32(best16, (uint32_t)__shfl_xor((int)best16, 32));
#pragma unroll
            for (int o = 1; o <= 32; o <<= 1)
                best32 = umin32(best32, (uint32_t)__shfl_xor((int)best32, o));
            if (grp == 0)
                B.key[21 + ((q << 4) | blk)] = best8;
            if (grp == 0 && (blk & 3) == 0)
                B.key[5 + ((q << 2) | (blk >> 2))] = best16;
            if (lane == 0)
                B.key[1 + q] = best32;
            LDS_BARRIER();
#if ME_EXP >= 10
            STAMP(2);
#endif
            if (t < 85) {
                uint32_t s;
                int p;
                if (t == 0) {
                    const unsigned long long k = B.key64;
                    const uint32_t code = (uint32_t)(k & 0x7fff);
                    s = (uint32_t)(k >> 15);
                    p = (code & 0x4000u) ? (int)(code & 0x3fff) : 16383 - (int)code;
                } else {
                    const uint32_t k = B.key[t];
                    s = k >> 14;
                    p = (int)(k & 0x3fff);
                }
                const int sy = (int)fastdiv((uint32_t)p, rcs), sx = p - sy * saw;
                B.best_sad[list][t] = 2 * s;
                B.best_mv[list][t] = mvpack((sx + sox) * 4, (sy + soy) * 4);
            }
            STAMP(14);
            __builtin_amdgcn_s_waitcnt(0); /* sub-pel windows landed (LDS-DMA issued before the search) */
            __syncthreads();
        }

        STAMP(7);
        /* ---- sub-pel (:4236-4318) ---- */
        /* SuPelEnable (:3035-3361): tier sums of MV components and SADs, one PU per thread */
        if (P.fractional_search_model == 1) {
            /* wave 0 speaks for the 64 8x8 PUs (21..84); wave 1: lanes 0..15 the 16x16 PUs (5..20), lanes 16..19 the 32x32 PUs (1..4), the rest of its first 32 lanes
             * zeros - xor-shuffle sums (steps 1, 2, 4, 8 inside 16 lanes; 16, 32 as well in wave 0) instead of 252 LDS atomics on 9 addresses */
            if (t < 128) {
                const int lane = t & 63, n = t < 64 ? 21 + lane : lane < 16 ? 5 + lane : lane < 20 ? lane - 15 : -1;
                int vx = 0, vy = 0, vs = 0;
                if (n >= 0)
                    vx = mvx(B.best_mv[list][n]), vy = mvy(B.best_mv[list][n]), vs = (int)B.best_sad[list][n];
                if (t < 64) {
                    vx = (int)wave_sum((uint32_t)vx), vy = (int)wave_sum((uint32_t)vy), vs = (int)wave_sum((uint32_t)vs);
                } else {
                    vx = (int)group_sum<16>((uint32_t)vx), vy = (int)group_sum<16>((uint32_t)vy), vs = (int)group_sum<16>((uint32_t)vs);
                }
                if (t == 0 || t == 64 || t == 80) {
                    const int tier = t == 0 ? 2 : t == 64 ? 1 : 0;
                    B.sums[tier * 3 + 0] = vx, B.sums[tier * 3 + 1] = vy, B.sums[tier * 3 + 2] = vs;
                }
            }
            __syncthreads();
        }
        if (t == 0) {
            int e32 = 0, e16 = 0, e8 = 0, eq = 0;
            if (P.fractional_search_model == 0) {
                e32 = e16 = e8 = eq = 1;
            } else if (P.fractional_search_model == 1) {
                const int shift[3] = {2, 4, 6};
                uint32_t mag[3], avgsad[3];
#pragma unroll
                for (int tt = 0; tt < 3; tt++) {
                    const uint32_t ux = (uint32_t)(B.sums[tt * 3 + 0] >> shift[tt]), uy = (uint32_t)(B.sums[tt * 3 + 1] >> shift[tt]);
                    mag[tt] = ux * ux + uy * uy;
                    avgsad[tt] = (uint32_t)B.sums[tt * 3 + 2] >> shift[tt];
                }
                const int tl = P.temporal_layer_index;
                const uint32_t th = tl == 0 ? 48 * 48 : tl == 1 ? 32 * 32 : tl == 2 ? 80 * 80 : 48 * 48;
                const int small32 = mag[0] < th, low32 = avgsad[0] < 32 * 32 * 6;
                e32 = (tl == 0 || tl == 2) ? low32 : (tl == 1 ? (small32 ? low32 : 1) : (small32 ? 1 : low32));
                e16 = !(avgsad[1] < 16 * 16 * 2);
                e8 = (tl <= 2) ? !(avgsad[2] < 8 * 8 * 2) : ((mag[2] < th) ? !(avgsad[2] < 8 * 8 * 2) : 0);
                eq = 1;
            }
            B.e32 = e32, B.e16 = e16 && P.cu16x16_mode == 0, B.e8 = e8 && P.cu8x8_mode != 1, B.eq = eq;
        }
        __syncthreads();
        const int any_sub = B.e32 || B.e16 || B.e8 || B.eq || 0;
        const int run_sub = (P.fractional_search_model != 2) && any_sub;
        if (run_sub) {
            const int f64 = P.fractional_search_64x64;
            const int en0 = f64, en1 = B.e32, en2 = B.e16, en3 = B.e8;
#define EN(tier_) pick4(tier_, en0, en1, en2, en3)
            const int rstep = (method == SVT_AMD_SUB_SAD_SEARCH) ? 2 : 1;
            STAMP(8);
            /* ===== half-pel: EbHevcHalfPelSearch_LCU / PU_HalfPelRefinement (:733-1187) ===== */
            for (int i = t; i < 85 * 8; i += NT) {
                (&B.dist[0][0])[i] = 0;
                (&B.dsad[0][0])[i] = 0;
            }
            __syncthreads();
            /* item = (PU, position k, row chunk): chunk counts {32, 8, 2, 1} per tier give every lane the same
             * ~16 dword SADs; the chunks of one (PU, k) sit on adjacent lanes and are summed with a segmented
             * shuffle, so B.dist is written once per (PU, k) - no LDS atomics. */
            for (int tier = 0; tier < 4; tier++) {
                if (!EN(tier))
                    continue;
                const int sz = tier_sz_of(tier), rows = sz >> (rstep - 1), lc = tier_lc_of(tier), rpc = rows >> lc;
                const int items = tier_cnt_of(tier) * 8 << lc; /* a multiple of NT: whole waves, no tail */
                for (int i = t; i < items; i += NT) {
                    const int ch = i & ((1 << lc) - 1), k = (i >> lc) & 7, n = tier_first_of(tier) + (i >> (lc + 3));
                    int px_, py_, psz;
                    pu_geom_z(n, px_, py_, psz);
                    const uint32_t mv = B.best_mv[list][n];
                    /* order L,R,T,B,TL,TR,BR,BL: planes b,b,h,h,j,j,j,j; offsets */
                    const LWin &pl = (k < 2) ? wB : (k < 4 ? wH : wJ);
                    const int ddx = (k == 1 || k == 5 || k == 6) ? 1 : 0, ddy = (k == 3 || k == 6 || k == 7) ? 1 : 0;
                    const int y0 = ch * rpc * rstep;
                    const uint8_t *r = wat(pl, ox + px_ + (mvx(mv) >> 2) + ddx, oy + py_ + (mvy(mv) >> 2) + y0 + ddy);
                    const uint8_t *sp = &S.src[(py_ + y0) * LCU + px_];
                    uint32_t d = 0, sd = 0;
                    /* the item's rows as 16-byte segments (8x8 PUs: 8-byte), four at a time: ALL their LDS reads first (source: one aligned read; window:
                     * aligned dwords + v_alignbyte), then the metric - one LDS round trip per four segments instead of one per eight samples */
                    const int lgspr = tier == 0 ? 2 : tier == 1 ? 1 : 0, nseg = rpc << lgspr, rs = pl.stride * rstep;
                    for (int s0_ = 0; s0_ < nseg; s0_ += 4) {
                        uint32_t sv[4][4], rv[4][4];
#pragma unroll
                        for (int q_ = 0; q_ < 4; q_++) {
                            const int sg = s0_ + q_, rr = sg >> lgspr, x = (sg & ((1 << lgspr) - 1)) << 4;
                            if (sz >= 16) {
                                const uint4 v4 = *(const uint4 *)(sp + rr * (LCU * rstep) + x);
                                sv[q_][0] = v4.x, sv[q_][1] = v4.y, sv[q_][2] = v4.z, sv[q_][3] = v4.w;
                                lds_ld_unaligned<4>(r + rr * rs + x, rv[q_]);
                            } else {
                                const uint2 v2 = *(const uint2 *)(sp + rr * (LCU * rstep));
                                uint32_t w2[2];
                                lds_ld_unaligned<2>(r + rr * rs, w2);
                                sv[q_][0] = v2.x, sv[q_][1] = v2.y, rv[q_][0] = w2[0], rv[q_][1] = w2[1];
                                sv[q_][2] = sv[q_][3] = rv[q_][2] = rv[q_][3] = 0; /* metric of equal words: nothing */
                            }
                        }
#pragma unroll
                        for (int q_ = 0; q_ < 4; q_++) {
                            row_metric(method, sv[q_][0], rv[q_][0], d, sd);
                            row_metric(method, sv[q_][1], rv[q_][1], d, sd);
                            if (sz >= 16) {
                                row_metric(method, sv[q_][2], rv[q_][2], d, sd);
                                row_metric(method, sv[q_][3], rv[q_][3], d, sd);
                            }
                        }
                    }
                    d = group_sum_rt(d, lc);
                    if (method == SVT_AMD_SSD_SEARCH)
                        sd = group_sum_rt(sd, lc);
                    if (ch == 0) {
                        B.dist[n][k] = d;
                        B.dsad[n][k] = sd;
                    }
                }
            }
            /* SSD search also needs the SSE of the full-pel winner (:798-806) */
            if (method == SVT_AMD_SSD_SEARCH) {
                for (int tier = 0; tier < 4; tier++) {
                    if (!EN(tier))
                        continue;
                    const int sz = tier_sz_of(tier), items = tier_cnt_of(tier) * sz;
                    for (int i = t; i < items; i += NT) {
                        const int row = i % sz, n = tier_first_of(tier) + i / sz;
                        int px_, py_, psz;
                        pu_geom_z(n, px_, py_, psz);
                        const uint32_t mv = B.best_mv[list][n];
                        const uint8_t *r = wat(wF, ox + px_ + (mvx(mv) >> 2), oy + py_ + (mvy(mv) >> 2) + row);
                        const uint8_t *s = &S.src[(py_ + row) * LCU + px_];
                        uint32_t d = 0;
                        for (int x = 0; x < sz; x += 8) {
                            uint32_t v[2];
                            lds_ld_unaligned<2>(r + x, v);
                            d += ssd4(*(const uint32_t *)(s + x), v[0]) + ssd4(*(const uint32_t *)(s + x + 4), v[1]);
                        }
                        atomicAdd(&B.best_ssd[list][n], d);
                    }
                }
            }
            __syncthreads();
            if (t < 85) {
                const int tier = t == 0 ? 0 : t < 5 ? 1 : t < 21 ? 2 : 3;
                if (EN(tier)) {
                    const int mdx[8] = {-2, 2, 0, 0, -2, 2, 2, -2}, mdy[8] = {0, 0, -2, 2, -2, -2, 2, 2};
                    const uint32_t mv0 = B.best_mv[list][t];
                    uint32_t bsad = B.best_sad[list][t], bmv = mv0, bssd = B.best_ssd[list][t];
                    uint32_t dmin = 0xffffffffu;
                    for (int k = 0; k < 8; k++) {
                        const uint32_t d = (method == SVT_AMD_SUB_SAD_SEARCH) ? (B.dist[t][k] << 1) : B.dist[t][k];
                        dmin = d < dmin ? d : dmin;
                        if (method == SVT_AMD_SSD_SEARCH) {
                            if (d < bssd)
                                bsad = B.dsad[t][k], bmv = mvpack(mvx(mv0) + mdx[k], mvy(mv0) + mdy[k]), bssd = d;
                        } else if (d < bsad) {
                            bsad = d, bmv = mvpack(mvx(mv0) + mdx[k], mvy(mv0) + mdy[k]);
                        }
                    }
                    /* first match in the order L,R,T,B,TL,TR,BL,BR (:1002-1025) */
                    const int chk[8] = {0, 1, 2, 3, 4, 5, 7, 6};
                    const uint8_t code[8] = {D_L, D_R, D_T, D_B, D_TL, D_TR, D_BR, D_BL};
                    uint8_t dirv = 0;
                    for (int i = 7; i >= 0; i--) {
                        const uint32_t d = (method == SVT_AMD_SUB_SAD_SEARCH) ? (B.dist[t][chk[i]] << 1) : B.dist[t][chk[i]];
                        if (d == dmin)
                            dirv = code[chk[i]];
                    }
                    B.best_sad[list][t] = bsad, B.best_mv[list][t] = bmv, B.best_ssd[list][t] = bssd;
                    B.dir[list][t] = dirv;
                }
            }
            __syncthreads();

            STAMP(9);
            /* ===== quarter-pel: QuarterPelSearch_LCU / PU_QuarterPelRefinementOnTheFly (:1226-1846) ===== */
            const int qen0 = f64, qen1 = B.eq && B.e32, qen2 = B.eq && B.e16, qen3 = B.eq && B.e8;
#define QEN(tier_) pick4(tier_, qen0, qen1, qen2, qen3)
            for (int i = t; i < 85 * 8; i += NT) {
                (&B.dist[0][0])[i] = 0;
                (&B.dsad[0][0])[i] = 0;
            }
            __syncthreads();
            /* only the three positions next to the half-pel winner are evaluated (:1252-1273): item =
             * (PU, j in 0..2, row chunk); position code = winner direction + j - 1 on the ring
             * TL,T,TR,R,BR,B,BL,L (mirrored when the MV already sits on a half-pel position). */
            for (int tier = 0; tier < 4; tier++) {
                if (!QEN(tier))
                    continue;
                const int sz = tier == 0 ? 32 : tier_sz_of(tier); /* the 64x64 call passes 32x32 (:1677) */
                const int lc = tier == 0 ? 3 : tier_lc_of(tier), rows = sz >> (rstep - 1), rpc = rows >> lc;
                const int items = tier_cnt_of(tier) * 3 << lc;
                for (int i0 = 0; i0 < items; i0 += NT) { /* uniform trips: whole waves join the shuffles */
                    const int i = i0 + t;
                    const bool live = i < items;
                    const int ii = live ? i : 0;
                    const int ch = ii & ((1 << lc) - 1), pj = ii >> lc, pidx = pj / 3, j = pj - pidx * 3;
                    const int n = tier_first_of(tier) + pidx;
                    int px_, py_, psz;
                    pu_geom_z(n, px_, py_, psz);
                    const uint32_t mv = B.best_mv[list][n];
                    const int xMv = mvx(mv), yMv = mvy(mv);
                    const int qm = (yMv & 2) + ((xMv & 2) >> 1);
                    const int code = (B.dir[list][n] + j - 1 + (qm ? 4 : 0)) & 7;
                    const int k = (int)((0x07361524u >> (4 * code)) & 7u); /* direction code -> position index */
                    const int y = ch * rpc * rstep;
                    const int ax = ox + px_ + ((xMv + 2) >> 2), ay = oy + py_ + ((yMv + 2) >> 2) + y;
                    const QSrc q0 = c_qtab[qm][k][0], q1 = c_qtab[qm][k][1];
                    const LWin &w1 = q0.plane == 0 ? wF : q0.plane == 1 ? wB : q0.plane == 2 ? wH : wJ;
                    const LWin &w2 = q1.plane == 0 ? wF : q1.plane == 1 ? wB : q1.plane == 2 ? wH : wJ;
                    const uint8_t *r1 = wat(w1, ax + q0.dx, ay + q0.dy);
                    const uint8_t *r2 = wat(w2, ax + q1.dx, ay + q1.dy);
                    const uint8_t *sp = &S.src[(py_ + y) * LCU + px_];
                    uint32_t d = 0, sdv = 0;
                    if (live) {
                        /* as in the half-pel stage: 16-byte segments, four at a time, every LDS read (source and the two planes) before the metric */
                        const int lgspr = sz == 32 ? 1 : 0, nseg = rpc << lgspr, rs1 = w1.stride * rstep, rs2 = w2.stride * rstep;
                        for (int s0_ = 0; s0_ < nseg; s0_ += 4) {
                            uint32_t sv[4][4], v1[4][4], v2[4][4];
#pragma unroll
                            for (int q_ = 0; q_ < 4; q_++) {
                                const int sg = s0_ + q_, rr = sg >> lgspr, x = (sg & ((1 << lgspr) - 1)) << 4;
                                /* source = MeContext_t.lcuBuffer: zero outside the picture (trap A19); lw is a multiple of 8 */
                                const bool inside_y = (py_ + y + rr * rstep) < lh, in0 = inside_y && (px_ + x) < lw, in1 = inside_y && (px_ + x + 8) < lw;
                                if (sz >= 16) {
                                    const uint4 v4 = *(const uint4 *)(sp + rr * (LCU * rstep) + x);
                                    sv[q_][0] = in0 ? v4.x : 0u, sv[q_][1] = in0 ? v4.y : 0u, sv[q_][2] = in1 ? v4.z : 0u, sv[q_][3] = in1 ? v4.w : 0u;
                                    lds_ld_unaligned<4>(r1 + rr * rs1 + x, v1[q_]);
                                    lds_ld_unaligned<4>(r2 + rr * rs2 + x, v2[q_]);
                                } else {
                                    const uint2 s2 = *(const uint2 *)(sp + rr * (LCU * rstep));
                                    uint32_t a2[2], b2[2];
                                    lds_ld_unaligned<2>(r1 + rr * rs1, a2);
                                    lds_ld_unaligned<2>(r2 + rr * rs2, b2);
                                    sv[q_][0] = in0 ? s2.x : 0u, sv[q_][1] = in0 ? s2.y : 0u, v1[q_][0] = a2[0], v1[q_][1] = a2[1], v2[q_][0] = b2[0], v2[q_][1] = b2[1];
                                    sv[q_][2] = sv[q_][3] = v1[q_][2] = v1[q_][3] = v2[q_][2] = v2[q_][3] = 0;
                                }
                            }
#pragma unroll
                            for (int q_ = 0; q_ < 4; q_++) {
                                row_metric(method, sv[q_][0], avg4(v1[q_][0], v2[q_][0]), d, sdv);
                                row_metric(method, sv[q_][1], avg4(v1[q_][1], v2[q_][1]), d, sdv);
                                if (sz >= 16) {
                                    row_metric(method, sv[q_][2], avg4(v1[q_][2], v2[q_][2]), d, sdv);
                                    row_metric(method, sv[q_][3], avg4(v1[q_][3], v2[q_][3]), d, sdv);
                                }
                            }
                        }
                    }
                    d = group_sum_rt(d, lc);
                    if (method == SVT_AMD_SSD_SEARCH)
                        sdv = group_sum_rt(sdv, lc);
                    if (live && ch == 0) {
                        B.dist[n][k] = d;
                        B.dsad[n][k] = sdv;
                    }
                }
            }
            __syncthreads();
            if (t < 85) {
                const int tier = t == 0 ? 0 : t < 5 ? 1 : t < 21 ? 2 : 3;
                if (QEN(tier)) {
                    const int mdx[8] = {-1, 1, 0, 0, -1, 1, 1, -1}, mdy[8] = {0, 0, -1, 1, -1, -1, 1, 1};
                    const int kcode[8] = {D_L, D_R, D_T, D_B, D_TL, D_TR, D_BR, D_BL};
                    const uint32_t mv0 = B.best_mv[list][t];
                    const int qm = (mvy(mv0) & 2) + ((mvx(mv0) & 2) >> 1);
                    const int sd = B.dir[list][t];
                    uint32_t bsad = B.best_sad[list][t], bmv = mv0, bssd = B.best_ssd[list][t];
                    for (int k = 0; k < 8; k++) {
                        const int target = qm ? ((kcode[k] + 4) & 7) : kcode[k];
                        const int diff = (sd - target) & 7;
                        if (!(diff == 0 || diff == 1 || diff == 7))
                            continue;
                        const uint32_t d = (method == SVT_AMD_SUB_SAD_SEARCH) ? (B.dist[t][k] << 1) : B.dist[t][k];
                        if (method == SVT_AMD_SSD_SEARCH) {
                            if (d < bssd)
                                bsad = B.dsad[t][k], bmv = mvpack(mvx(mv0) + mdx[k], mvy(mv0) + mdy[k]), bssd = d;
                        } else if (d < bsad) {
                            bsad = d, bmv = mvpack(mvx(mv0) + mdx[k], mvy(mv0) + mdy[k]);
                        }
                    }
                    B.best_sad[list][t] = bsad, B.best_mv[list][t] = bmv, B.best_ssd[list][t] = bssd;
                }
            }
            __syncthreads();
        }
        /* ---- this list's results (read back by the kernels of list 1 and by the parity tests) ---- */
        if (t < 85) {
            o->best_sad[list][t] = B.best_sad[list][t];
            o->best_mv[list][t] = B.best_mv[list][t];
            if (P.num_lists == 1)
                o->best_sad[1][t] = 0, o->best_mv[1][t] = 0;
        }
        if (t == 0) {
            o->search_origin_x[list] = (int16_t)sox, o->search_origin_y[list] = (int16_t)soy;
            o->search_w[list] = (uint8_t)saw, o->search_h[list] = (uint8_t)sah;
            if (P.num_lists == 1)
                o->search_origin_x[1] = 0, o->search_origin_y[1] = 0, o->search_w[1] = 0, o->search_h[1] = 0;
        }
        if (list != P.num_lists - 1) {
            STAMP(12);
            return;
        }

    STAMP(10);
    /* ---- bi-prediction (:2608-2917) ---- */
    if (P.num_lists == 2) {
        const int rstep = (method == SVT_AMD_SUB_SAD_SEARCH) ? 2 : 1;
        const int npu = (P.cu16x16_mode != 0) ? 5 : ((P.cu8x8_mode != 0) ? 21 : 85);
        /* items: (pu n, row, 16-sample segment), all tiers in ONE index space: 128 items each for the 64 / 32 / 16 tiers and 256
         * half-width ones for the 8x8 tier, so every thread gets two or three short items (a tier at a time left 32 threads
         * walking 64-sample rows of the 64x64 PU while 224 waited) */
        int cum[5];
        cum[0] = 0;
#pragma unroll
        for (int tier = 0; tier < 4; tier++) {
            const int sz = tier_sz_of(tier), rows = sz >> (rstep - 1), segs = sz >= 16 ? sz >> 4 : 1;
            cum[tier + 1] = cum[tier] + (tier_first_of(tier) < npu ? tier_cnt_of(tier) * rows * segs : 0);
        }
        /* A thread's items are taken three at a time: first the plane reads of all three are requested (one unaligned 16-byte load per plane: the address units take
         * a wave's 64 scattered rows once per plane; avg of identical pointers is the identity ((v+v+1)>>1 == v), so the second plane of a full / half-pel position is
         * not fetched), then they are evaluated - one trip to memory per three items instead of one per item. */
        for (int base = 0; base < cum[4]; base += ME_BI_ITEMS * NT) {
            uint4 va0[ME_BI_ITEMS], vb0[ME_BI_ITEMS], va1[ME_BI_ITEMS], vb1[ME_BI_ITEMS];
            int in_[ME_BI_ITEMS], ij[ME_BI_ITEMS], iG[ME_BI_ITEMS], iw[ME_BI_ITEMS];
            const uint8_t *is[ME_BI_ITEMS];
#pragma unroll
            for (int u = 0; u < ME_BI_ITEMS; u++) {
                const int i = base + u * NT + t;
                iG[u] = 0;
                if (i < cum[4]) {
            const int tier = i < cum[1] ? 0 : i < cum[2] ? 1 : i < cum[3] ? 2 : 3;
                    const int sz = tier_sz_of(tier), rows = sz >> (rstep - 1), lgsegs = tier == 0 ? 2 : tier == 1 ? 1 : 0, wseg = sz >= 16 ? 16 : sz;
                    const int j = i - pick4(tier, cum[0], cum[1], cum[2], cum[3]);
                    const int lgrows = 6 - tier - (rstep == 2 ? 1 : 0); /* rows = 1 << lgrows */
                    const int seg = j & ((1 << lgsegs) - 1), jr = j >> lgsegs, row = jr & (rows - 1), n = tier_first_of(tier) + (jr >> lgrows);
                    int px_, py_, psz;
                    pu_geom_z(n, px_, py_, psz);
                    const int y = row * rstep, xs = seg << 4;
                    /* SelectBuffer / QuarterPelCompensation (:2440-2600) as bit tables over the 16 fractional positions (a 16-way switch on pointers diverges per lane):
                     * planes 0 F, 1 b (+1 column), 2 h (+1 row), 3 j (+1 row, +1 column); the first source of position 15 and the second of 3, 7, 11 sit one column
                     * further right, the second source of 12 .. 15 one row further down.  List 0's samples come from memory (its windows left with its kernel);
                     * THIS list's are in the LDS windows of the search (the sub-pel stages read the same positions) - half of the scattered row fetches, which bound
                     * this stage in the texture path (64 lanes = 64 rows = 64+ cache lines per load instruction), never leave the CU. */
                    {
                        const uint32_t mv = B.best_mv[0][n];
                        const int xMv = mvx(mv), yMv = mvy(mv);
                        const int ax = ox + px_ + xs + (xMv >> 2), ay = oy + py_ + (yMv >> 2) + y;
                        const uint32_t frac = (uint32_t)((xMv & 3) + ((yMv & 3) << 2));
                        const ptrdiff_t pz = ref0.pitch_full;
                        const uint32_t pa = (0xBAFA5450u >> (2 * frac)) & 3u, pb = (0x54BEBA14u >> (2 * frac)) & 3u;
                        const ptrdiff_t at = (ptrdiff_t)ay * pz + ax;
                        const uint8_t *ba = pa == 0 ? ref0.full : pa == 1 ? ref0.hp_b : pa == 2 ? ref0.hp_h : ref0.hp_j;
                        const uint8_t *bb = pb == 0 ? ref0.full : pb == 1 ? ref0.hp_b : pb == 2 ? ref0.hp_h : ref0.hp_j;
                        const uint8_t *a0p = ba + at + (pa & 1u) + ((pa >> 1) ? pz : 0) + ((0x8000u >> frac) & 1u);
                        const uint8_t *b0p = bb + at + (pb & 1u) + ((pb >> 1) ? pz : 0) + ((0x0888u >> frac) & 1u) + (((0xF000u >> frac) & 1u) ? pz : 0);
                        va0[u] = ldu16(a0p);
                        vb0[u] = va0[u];
                        if (b0p != a0p)
                            vb0[u] = ldu16(b0p);
                    }
                    {
                        const uint32_t mv = B.best_mv[1][n];
                        const int xMv = mvx(mv), yMv = mvy(mv);
                        const int ax = ox + px_ + xs + (xMv >> 2), ay = oy + py_ + (yMv >> 2) + y;
                        const uint32_t frac = (uint32_t)((xMv & 3) + ((yMv & 3) << 2));
                        const uint32_t pa = (0xBAFA5450u >> (2 * frac)) & 3u, pb = (0x54BEBA14u >> (2 * frac)) & 3u;
                        const uint8_t *wpa = pa == 0 ? wF.p : pa == 1 ? wB.p : pa == 2 ? wH.p : wJ.p, *wpb = pb == 0 ? wF.p : pb == 1 ? wB.p : pb == 2 ? wH.p : wJ.p;
                        const int sa = pa == 0 ? wF.stride : wB.stride, sb = pb == 0 ? wF.stride : wB.stride; /* the three half-pel windows share their geometry */
                        const int xa0 = pa == 0 ? wF.x0 : wB.x0, xb0 = pb == 0 ? wF.x0 : wB.x0;
                        const uint8_t *a1p = wpa + (ay + (int)(pa >> 1) - wF.y0) * sa + (ax + (int)(pa & 1u) + (int)((0x8000u >> frac) & 1u) - xa0);
                        const uint8_t *b1p = wpb + (ay + (int)(pb >> 1) + (int)((0xF000u >> frac) & 1u) - wF.y0) * sb + (ax + (int)(pb & 1u) + (int)((0x0888u >> frac) & 1u) - xb0);
                        uint32_t w4[4];
                        lds_ld_unaligned<4>(a1p, w4);
                        va1[u] = make_uint4(w4[0], w4[1], w4[2], w4[3]);
                        vb1[u] = va1[u];
                        if (b1p != a1p) {
                            lds_ld_unaligned<4>(b1p, w4);
                            vb1[u] = make_uint4(w4[0], w4[1], w4[2], w4[3]);
                        }
                    }
                    const int per_pu = rows << lgsegs;
                    in_[u] = n, ij[u] = j, iG[u] = per_pu < 64 ? per_pu : 64, iw[u] = wseg;
                    is[u] = &S.src[(py_ + y) * LCU + px_ + xs];
                }
            }
#pragma unroll
            for (int u = 0; u < ME_BI_ITEMS; u++) {
                if (!iG[u]) /* past the end: whole waves (every tier's share of the index space is a multiple of 64) */
                    continue;
                const uint4 a0 = va0[u], b0 = vb0[u], a1 = va1[u], b1 = vb1[u];
                uint32_t d = 0;
                if (iw[u] == 16) {
                    const uint4 sv = *(const uint4 *)is[u];
                    d = sad4(sv.x, avg4(avg4(a0.x, b0.x), avg4(a1.x, b1.x)), d);
                    d = sad4(sv.y, avg4(avg4(a0.y, b0.y), avg4(a1.y, b1.y)), d);
                    d = sad4(sv.z, avg4(avg4(a0.z, b0.z), avg4(a1.z, b1.z)), d);
                    d = sad4(sv.w, avg4(avg4(a0.w, b0.w), avg4(a1.w, b1.w)), d);
                } else { /* 8x8 PUs: the first eight of the sixteen bytes */
                    const uint2 sv = *(const uint2 *)is[u];
                    d = sad4(sv.x, avg4(avg4(a0.x, b0.x), avg4(a1.x, b1.x)), d);
                    d = sad4(sv.y, avg4(avg4(a0.y, b0.y), avg4(a1.y, b1.y)), d);
                }
                /* the items of a PU sit on adjacent lanes (groups of 4 .. 64, aligned): summed by shuffles, one LDS atomic per group instead of one per item
                 * (64 lanes on one address serialise) */
                const int G = iG[u];
                d = group_sum_rt(d, 31 - __builtin_clz((unsigned)G));
                if ((ij[u] & (G - 1)) == 0)
                    atomicAdd(&B.bipred[in_[u]], d);
            }
        }
        __syncthreads();
    }

    STAMP(11);
    /* ---- candidate records (:4321-4440) ---- */
    if (t < 85) {
        const int pu = t;
        const int n = pu == 0 ? 0 : pu < 5 ? pu : pu < 21 ? c_tab16[pu - 5] + 5 : c_tab8[pu - 21] + 21;
        int total = P.num_lists;
        if (P.num_lists == 2 && (P.cu8x8_mode == 0 || pu < 21) && (P.cu16x16_mode == 0 || pu < 5))
            total = 3;
        const uint32_t bi = (method == SVT_AMD_SUB_SAD_SEARCH) ? (B.bipred[n] << 1) : B.bipred[n];
        const uint32_t v[3] = {B.best_sad[0][n], B.best_sad[1][n], bi};
        SvtAmdMeCuResult r;
        r.x_mv_l0 = (int16_t)mvx(B.best_mv[0][n]), r.y_mv_l0 = (int16_t)mvy(B.best_mv[0][n]);
        r.x_mv_l1 = (int16_t)mvx(B.best_mv[1][n]), r.y_mv_l1 = (int16_t)mvy(B.best_mv[1][n]);
        r.total_me_candidate_index = (uint8_t)total;
        for (int k = 0; k < 3; k++)
            r.distortion[k] = 0, r.direction[k] = 0;
        /* (selects instead of v[o3[k]]: a dynamically indexed private array lives in scratch memory) */
        const uint32_t a = v[0], b = v[1], c = v[2];
        if (total == 3) {
            /* Sort3Elements (:2919-2944) */
            int o0, o1, o2;
            if (a <= b && a <= c) {
                o0 = 0;
                if (b <= c) o1 = 1, o2 = 2; else o1 = 2, o2 = 1;
            } else if (b <= a && b <= c) {
                o0 = 1;
                if (a <= c) o1 = 0, o2 = 2; else o1 = 2, o2 = 0;
            } else if (a <= b) {
                o0 = 2, o1 = 0, o2 = 1;
            } else {
                o0 = 2, o1 = 1, o2 = 0;
            }
            r.distortion[0] = o0 == 0 ? a : o0 == 1 ? b : c, r.direction[0] = (uint8_t)o0;
            r.distortion[1] = o1 == 0 ? a : o1 == 1 ? b : c, r.direction[1] = (uint8_t)o1;
            r.distortion[2] = o2 == 0 ? a : o2 == 1 ? b : c, r.direction[2] = (uint8_t)o2;
        } else if (total == 2) {
            const int f = a <= b ? 0 : 1;
            r.distortion[0] = f ? b : a, r.direction[0] = (uint8_t)f;
            r.distortion[1] = f ? a : b, r.direction[1] = (uint8_t)(1 - f);
        } else {
            r.distortion[0] = a, r.direction[0] = SVT_AMD_UNI_PRED_LIST_0;
        }
        o->pu[pu] = r;
    }
    STAMP(12);
    } /* PHASE 1 */
}

/* upper bounds of the dynamic LDS pools a job needs: HME kernel = the largest per-level window set;
 * search kernel = MeSearch + the four staged search windows */
static void me_pool_bytes(const SvtAmdMeParams *p, size_t *hme_pool, size_t *search_pool)
{
    auto win = [](int w, int rows) { return (size_t)((w + 30) & ~15) * (size_t)rows; };
    size_t need = 0, v;
    const int nq = p->num_hme_regions_w * p->num_hme_regions_h;
    if (p->enable_hme_flag) {
        if (p->enable_hme_level0) {
            const int tw = (p->hme_l0_total_w * p->hme_l0_mult_x) / 100, th = (p->hme_l0_total_h * p->hme_l0_mult_y) / 100;
            int mw = tw, mh = th;
            for (int k = 0; k < 2; k++) {
                mw = mw > (p->hme_l0_w[k] * p->hme_l0_mult_x) / 100 ? mw : (p->hme_l0_w[k] * p->hme_l0_mult_x) / 100;
                mh = mh > (p->hme_l0_h[k] * p->hme_l0_mult_y) / 100 ? mh : (p->hme_l0_h[k] * p->hme_l0_mult_y) / 100;
            }
            v = (size_t)nq * win(mw + 16 + 16, mh + 15);
            need = v > need ? v : need;
        }
        for (int lvl = 1; lvl <= 2; lvl++) {
            if (!(lvl == 1 ? p->enable_hme_level1 : p->enable_hme_level2))
                continue;
            int mw = 8, mh = 1;
            for (int k = 0; k < 2; k++) {
                const int w = lvl == 1 ? p->hme_l1_w[k] : p->hme_l2_w[k], h = lvl == 1 ? p->hme_l1_h[k] : p->hme_l2_h[k];
                const int ww = (w < 8) ? 8 : (w & 7) ? w + (w - ((w >> 3) << 3)) : w;
                mw = ww > mw ? ww : mw;
                mh = h > mh ? h : mh;
            }
            v = (size_t)nq * win(mw + (lvl == 1 ? 32 : 64) + 16, mh + (lvl == 1 ? 31 : 63));
            need = v > need ? v : need;
        }
    }
    *hme_pool = (need + 64 + 255) & ~(size_t)255; /* +64: the aligned over-read of the last window row */
    const int saw = p->search_area_width > 127 ? 127 : p->search_area_width;
    const int sah = p->search_area_height > 127 ? 127 : p->search_area_height;
    /* the four search windows are staged with an odd 16-byte pitch (win_pitch(.., 1)): the widest a window of w columns gets is
     * w + 15 (alignment of its first column) rounded up to 16 and then to an odd multiple.  The bound is exact on purpose: at
     * BASELINE configs[2] (16 x 9 search) it is what lets a third workgroup share a CU's 160 KiB. */
    auto win_odd = [](int w, int rows) {
        int wa = (w + 15 + 15) & ~15;
        if (!((wa >> 4) & 1))
            wa += 16;
        return (size_t)wa * (size_t)rows;
    };
    *search_pool = ((size_t)ME_SEARCH_BYTES + 4 * win_odd(saw + 67, sah + 67) + 64 + 255) & ~(size_t)255;
}

int svt_amd_launch_me_batch(SvtAmdContext *ctx, const MeJobDev *host_jobs, int njobs, int max_lcus)
{
    if (njobs < 1 || njobs > SVT_AMD_MAX_BATCH)
        return SVT_AMD_ERR_BAD_PARAM;
    size_t pool0 = 0, pool1 = 0;
    int max_lists = 1;
    for (int i = 0; i < njobs; i++) {
        size_t a, b;
        me_pool_bytes(&host_jobs[i].P, &a, &b);
        pool0 = a > pool0 ? a : pool0;
        pool1 = b > pool1 ? b : pool1;
        max_lists = host_jobs[i].P.num_lists > max_lists ? host_jobs[i].P.num_lists : max_lists;
    }
    if (pool1 + sizeof(MeShared) > 160 * 1024 || pool0 + sizeof(MeShared) > 160 * 1024) {
        svt_amd_set_error("motion estimation: search windows need %zu B of LDS (> 160 KiB)", pool1 + sizeof(MeShared));
        return SVT_AMD_ERR_BAD_PARAM;
    }
    {   /* dynamic-LDS limits of the two kernels: high-water marks PER DEVICE (function attributes are per device) */
        static std::mutex mu;
        static size_t attr0[64], attr1[64];
        std::lock_guard<std::mutex> g(mu);
        const int dv = ctx->device & 63;
        if (pool0 > attr0[dv]) {
            HIP_TRY(hipFuncSetAttribute((const void *)k_me<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)pool0));
            attr0[dv] = pool0;
        }
        if (pool1 > attr1[dv]) {
            HIP_TRY(hipFuncSetAttribute((const void *)k_me<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)pool1));
            attr1[dv] = pool1;
        }
    }
    /* copied into a pinned ring before returning, so host_jobs may be reused */
    {
        const int rcd = svt_amd_upload_descriptors(ctx, ctx->d_jobs, host_jobs, sizeof(MeJobDev) * (size_t)njobs);
        if (rcd)
            return rcd;
    }
    int rc = svt_amd_stamp_begin(ctx, KC_ME_SEARCH);
    if (rc)
        return rc;
    const dim3 grid((unsigned)((max_lcus + 7) & ~7), (unsigned)njobs);
    for (int list = 0; list < max_lists; list++) { /* list 1 depends on list 0's result (direct candidate, bi-pred) */
        hipLaunchKernelGGL(k_me<0>, grid, dim3(NT), pool0, ctx->stream, ctx->d_jobs, list);
        hipLaunchKernelGGL(k_me<1>, grid, dim3(NT), pool1, ctx->stream, ctx->d_jobs, list);
    }
    HIP_TRY(hipGetLastError());
    return svt_amd_stamp_end(ctx);
}
