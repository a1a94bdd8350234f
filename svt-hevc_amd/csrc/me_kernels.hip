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
#include "svt_amd_internal.h"

#define NT 256
#define LCU 64
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

__device__ __forceinline__ uint32_t wave_sum(uint32_t v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1)
        v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ unsigned long long wave_min64(unsigned long long v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        unsigned long long w = __shfl_xor(v, o);
        v = w < v ? w : v;
    }
    return v;
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

struct MeShared {
    uint8_t src[LCU * LCU + 16];   /* source LCU rows (padded-plane content)              */
    uint8_t qsrc[32 * 16 + 16];    /* 1/4 LCU, even rows                                   */
    uint8_t ssrc[16 * 8 + 16];     /* 1/16 LCU, even rows                                  */
    uint16_t sad8[64][64];         /* per-position 8x8 even-row SADs of the current chunk  */
    uint16_t sad16[64][16];
    uint32_t sad32[64][4];
    uint32_t key[85];              /* packed (sad,index) minima, PUs 1..84                 */
    unsigned long long key64;      /* 64x64 */
    unsigned long long hkey[4];    /* HME per-quadrant minima                              */
    uint32_t best_sad[2][85], best_mv[2][85], best_ssd[2][85];
    uint8_t dir[2][85];
    uint32_t dist[85][8];          /* sub-pel distortions (search metric)                  */
    uint32_t dsad[85][8];          /* full SAD at the same positions (SSD search only)     */
    uint32_t bipred[85];
    uint32_t acc[8];               /* LCU-level SAD accumulators                           */
    int16_t hx[3][2][2], hy[3][2][2]; /* HME centres per level [w][h]                      */
    unsigned long long hs[3][2][2];
    int cx, cy;                    /* search centre of the current list                    */
    int e32, e16, e8, eq;
};

/* ------------------------------------------------------------------------- */

template <int DUMMY>
__device__ void lcu_sads(MeShared &S, const uint8_t *ref, int pitch, int ox, int oy, int lw, int lh, int ncand,
                         const int *cdx, const int *cdy, int t)
{
    /* NxMSadKernel(lcuSrcPtr, stride<<1, ref, stride<<1, lcuHeight>>1, lcuWidth) per candidate */
    if (t < 8)
        S.acc[t] = 0;
    __syncthreads();
    const int rows = lh >> 1;
    for (int i = t; i < ncand * rows; i += NT) {
        const int c = i / rows, r = i - c * rows;
        const uint32_t s = row_sad(&S.src[(2 * r) * LCU], ref + (ptrdiff_t)(oy + cdy[c] + 2 * r) * pitch + ox + cdx[c], lw);
        atomicAdd(&S.acc[c], s);
    }
    __syncthreads();
}

/* One HME pass over up to four quadrants of one pyramid level.
 * SadLoopKernel semantics (C_DEFAULT/EbComputeSAD_C.c:170): raster scan, strict '<'. */
__device__ void hme_pass(MeShared &S, int level, const uint8_t *refplane, int pitch, int bx0, int by0, int bw,
                         int rows, int nq, const int *qox, const int *qoy, const int *qw, const int *qh, int t)
{
    if (t < 4)
        S.hkey[t] = ~0ull;
    __syncthreads();
    const uint8_t *src = level == 0 ? S.ssrc : level == 1 ? S.qsrc : S.src;
    const int sstride = level == 0 ? 16 : level == 1 ? 32 : 2 * LCU;
    for (int q = 0; q < nq; q++) {
        const int npos = qw[q] * qh[q];
        unsigned long long best = ~0ull;
        for (int p = t; p < npos; p += NT) {
            const int sy = p / qw[q], sx = p - sy * qw[q];
            const uint8_t *r = refplane + (ptrdiff_t)(by0 + qoy[q] + sy) * pitch + bx0 + qox[q] + sx;
            uint32_t s = 0;
            for (int y = 0; y < rows; y++)
                s += row_sad(src + y * sstride, r + (ptrdiff_t)(2 * y) * pitch, bw);
            const unsigned long long k = ((unsigned long long)s << 32) | (uint32_t)p;
            best = k < best ? k : best;
        }
        best = wave_min64(best);
        if ((t & 63) == 0)
            atomicMin(&S.hkey[q], best);
    }
    __syncthreads();
}

__device__ __forceinline__ int hme_l12_width(int w) { return (w < 8) ? 8 : (w & 7) ? w + (w - ((w >> 3) << 3)) : w; }

/* sub-pel row distortion: metric by fractionalSearchMethod; optional full SAD alongside */
__device__ __forceinline__ void row_metric(int method, uint32_t a, uint32_t b, uint32_t &d, uint32_t &sad)
{
    if (method == SVT_AMD_SSD_SEARCH) {
        d += ssd4(a, b);
        sad = sad4(a, b, sad);
    } else {
        d = sad4(a, b, d);
    }
}

__global__ __launch_bounds__(NT) void k_me_picture(SvtAmdMeParams P, PicView cur, PicView ref0, PicView ref1,
                                                   SvtAmdMeLcuResult *__restrict__ out, int lcu_begin)
{
    __shared__ MeShared S;
    const int t = threadIdx.x;
    const int W = P.luma_width, H = P.luma_height;
    const int wl = (W + LCU - 1) / LCU;
    const int lcu = lcu_begin + (int)blockIdx.x;
    const int ox = (lcu % wl) * LCU, oy = (lcu / wl) * LCU;
    const int lw = imin(LCU, W - ox), lh = imin(LCU, H - oy);
    const int pf = cur.pitch_full;
    const int method = P.fractional_search_method;

    /* ---- stage the source LCU (EbMotionEstimationProcess.c:714-779) ---- */
    for (int i = t; i < LCU * LCU / 4; i += NT) {
        const int y = i >> 4, x = (i & 15) << 2;
        *(uint32_t *)&S.src[y * LCU + x] = *(const uint32_t *)(cur.full + (ptrdiff_t)(oy + y) * pf + ox + x);
    }
    if (t < 128) { /* 1/4: 16 even rows x 32 */
        const int y = t >> 3, x = (t & 7) << 2;
        *(uint32_t *)&S.qsrc[y * 32 + x] =
            ld4(cur.quarter + (ptrdiff_t)((oy >> 1) + 2 * y) * cur.pitch_quarter + (ox >> 1) + x);
    } else if (t < 160) { /* 1/16: 8 even rows x 16 */
        const int u = t - 128, y = u >> 2, x = (u & 3) << 2;
        *(uint32_t *)&S.ssrc[y * 16 + x] =
            ld4(cur.sixteenth + (ptrdiff_t)((oy >> 2) + 2 * y) * cur.pitch_sixteenth + (ox >> 2) + x);
    }
    for (int i = t; i < 2 * 85; i += NT) {
        (&S.best_sad[0][0])[i] = 0;
        (&S.best_mv[0][0])[i] = 0;
        (&S.best_ssd[0][0])[i] = 0;
        (&S.dir[0][0])[i] = 0;
    }
    if (t < 85)
        S.bipred[t] = 0;
    if (t < 12) {
        (&S.hx[0][0][0])[t] = 0;
        (&S.hy[0][0][0])[t] = 0;
        (&S.hs[0][0][0])[t] = 0;
    }
    __syncthreads();

    int hme_init_done = 0;
    int sa_x[2] = {0, 0}, sa_y[2] = {0, 0}, sa_w[2] = {0, 0}, sa_h[2] = {0, 0};
    int hcx[2] = {0, 0}, hcy[2] = {0, 0};

    for (int list = 0; list < P.num_lists; list++) {
        const PicView &R = list ? ref1 : ref0;
        int cx = 0, cy = 0;

        if (P.temporal_layer_index > 0 || list == 0) {
            /* ---- TestSearchAreaBounds (EbMotionEstimation.c:3363-3665) ---- */
            if (P.update_hme_search_center) {
                int cdx[6], cdy[6], ux[6], uy[6];
                ux[0] = 0, uy[0] = 0;
                ux[1] = -(int)P.hme_l0_total_w, uy[1] = 0;
                ux[2] = (int)P.hme_l0_total_w, uy[2] = 0;
                ux[3] = 0, uy[3] = -(int)P.hme_l0_total_h;
                ux[4] = 0, uy[4] = (int)P.hme_l0_total_h;
                ux[5] = 0 - (mvx(S.best_mv[0][0]) >> 2), uy[5] = 0 - (mvy(S.best_mv[0][0]) >> 2);
                const int nc = list == 1 ? 6 : 5;
                for (int k = 0; k < 6; k++) {
                    ux[k] = (int16_t)ux[k], uy[k] = (int16_t)uy[k];
                    cdx[k] = k ? clamp_center(ox, ux[k], LCU - 1, W) : 0;
                    cdy[k] = k ? clamp_center(oy, uy[k], LCU - 1, H) : 0;
                }
                lcu_sads<0>(S, R.full, R.pitch_full, ox, oy, lw, lh, nc, cdx, cdy, t);
                unsigned long long cost[6], best = ~0ull;
                for (int k = 0; k < 6; k++) {
                    cost[k] = k < nc ? ((unsigned long long)(S.acc[k] << 1) << COST_PRECISION) : 0xFFFFFFFFFFFFFull;
                    best = cost[k] < best ? cost[k] : best;
                }
                const int order[6] = {0, 1, 2, 3, 5, 4}; /* zero, A, B, C, direct, D */
                for (int i = 5; i >= 0; i--)
                    if (best == cost[order[i]])
                        cx = ux[order[i]], cy = uy[order[i]];
                __syncthreads(); /* S.acc is reused below */
            }

            /* ---- HME (EbMotionEstimation.c:3800-4069) ---- */
            if (P.enable_hme_flag && lh == LCU) {
                const int nw = P.num_hme_regions_w, nh = P.num_hme_regions_h;
                if (!hme_init_done) {
                    if (t == 0)
                        for (int h = 0; h < imin(nh, 2); h++)
                            for (int w = 0; w < imin(nw, 2); w++) {
                                const int sh0 = P.update_hme_search_center ? 2 : 0, sh1 = P.update_hme_search_center ? 1 : 0;
                                S.hx[0][w][h] = (int16_t)(cx >> sh0), S.hy[0][w][h] = (int16_t)(cy >> sh0);
                                S.hx[1][w][h] = (int16_t)(cx >> sh1), S.hy[1][w][h] = (int16_t)(cy >> sh1);
                                S.hx[2][w][h] = (int16_t)cx, S.hy[2][w][h] = (int16_t)cy;
                            }
                    hme_init_done = 1;
                    __syncthreads();
                }
                const uint32_t mx = P.hme_l0_mult_x, my = P.hme_l0_mult_y;
                int qox[4], qoy[4], qw[4], qh[4];
                if (P.enable_hme_level0) {
                    const int px16 = ox >> 2, py16 = oy >> 2;
                    const int pw16 = W >> 2, ph16 = H >> 2, pad16 = SVT_AMD_PAD_SIXTEENTH - 1;
                    int nq;
                    if (P.one_quadrant_hme && !P.enable_hme_level1 && !P.enable_hme_level2) {
                        /* EbHevcHmeOneQuadrantLevel0 (:1847-2010) */
                        int sw = (int16_t)((P.hme_l0_total_w * mx) / 100), sh = (int16_t)((P.hme_l0_total_h * my) / 100);
                        int so_x = -(int)(int16_t)(sw >> 1) + (cx >> 2), so_y = -(int)(int16_t)(sh >> 1) + (cy >> 2);
                        clamp_area(px16, pad16, pw16, so_x, sw);
                        clamp_area(py16, pad16, ph16, so_y, sh);
                        if (sw & 15)
                            sw = (sw >> 4) << 4;
                        qox[0] = so_x, qoy[0] = so_y, qw[0] = sw, qh[0] = sh;
                        nq = 1;
                    } else {
                        nq = 0;
                        for (int h = 0; h < nh; h++)
                            for (int w = 0; w < nw; w++) {
                                int sw = (int16_t)((P.hme_l0_w[w] * mx) / 100), sh = (int16_t)((P.hme_l0_h[h] * my) / 100);
                                int dx = cx >> 2, dy = cy >> 2;
                                for (int k = w; k > 0; k--)
                                    dx += (int16_t)((P.hme_l0_w[k - 1] * mx) / 100);
                                for (int k = h; k > 0; k--)
                                    dy += (int16_t)((P.hme_l0_h[k - 1] * my) / 100);
                                int so_x = (int16_t)(-(int)(int16_t)(((P.hme_l0_total_w * mx) / 100) >> 1) + dx);
                                int so_y = (int16_t)(-(int)(int16_t)(((P.hme_l0_total_h * my) / 100) >> 1) + dy);
                                clamp_area(px16, pad16, pw16, so_x, sw);
                                clamp_area(py16, pad16, ph16, so_y, sh);
                                qox[nq] = so_x, qoy[nq] = so_y, qw[nq] = sw, qh[nq] = sh;
                                nq++;
                            }
                    }
                    hme_pass(S, 0, R.sixteenth, R.pitch_sixteenth, px16, py16, lw >> 2, 8, nq, qox, qoy, qw, qh, t);
                    if (t == 0) {
                        int q = 0;
                        const int one = (P.one_quadrant_hme && !P.enable_hme_level1 && !P.enable_hme_level2);
                        for (int h = 0; h < (one ? 1 : nh); h++)
                            for (int w = 0; w < (one ? 1 : nw); w++, q++) {
                                const unsigned long long k = S.hkey[q];
                                if (k != ~0ull) { /* an empty search leaves the centre untouched */
                                    const int p = (int)(uint32_t)k, sy = p / qw[q], sx = p - sy * qw[q];
                                    S.hx[0][w][h] = (int16_t)((int16_t)(sx + qox[q]) * 4);
                                    S.hy[0][w][h] = (int16_t)((int16_t)(sy + qoy[q]) * 4);
                                    S.hs[0][w][h] = (k >> 32) * 2;
                                } else {
                                    S.hs[0][w][h] = 0xffffffull * 2;
                                    S.hx[0][w][h] = (int16_t)((int16_t)(S.hx[0][w][h] + qox[q]) * 4);
                                    S.hy[0][w][h] = (int16_t)((int16_t)(S.hy[0][w][h] + qoy[q]) * 4);
                                }
                            }
                    }
                    __syncthreads();
                }
                for (int lvl = 1; lvl <= 2; lvl++) {
                    if (!(lvl == 1 ? P.enable_hme_level1 : P.enable_hme_level2))
                        continue;
                    const int shf = 2 - lvl;
                    const int bx0 = ox >> shf, by0 = oy >> shf, pwl = W >> shf, phl = H >> shf;
                    const int padl = lvl == 2 ? LCU - 1 : SVT_AMD_PAD_QUARTER - 1;
                    int nq = 0;
                    for (int h = 0; h < nh; h++)
                        for (int w = 0; w < nw; w++) {
                            int sw = hme_l12_width((int16_t)(lvl == 1 ? P.hme_l1_w[w] : P.hme_l2_w[w]));
                            int sh = (int16_t)(lvl == 1 ? P.hme_l1_h[h] : P.hme_l2_h[h]);
                            const int pcx = lvl == 1 ? (S.hx[0][w][h] >> 1) : S.hx[1][w][h];
                            const int pcy = lvl == 1 ? (S.hy[0][w][h] >> 1) : S.hy[1][w][h];
                            int so_x = (int16_t)(-(sw >> 1) + pcx), so_y = (int16_t)(-(sh >> 1) + pcy);
                            clamp_area(bx0, padl, pwl, so_x, sw);
                            clamp_area(by0, padl, phl, so_y, sh);
                            qox[nq] = so_x, qoy[nq] = so_y, qw[nq] = sw, qh[nq] = sh;
                            nq++;
                        }
                    hme_pass(S, lvl, lvl == 1 ? R.quarter : R.full, lvl == 1 ? R.pitch_quarter : R.pitch_full, bx0, by0,
                             lw >> shf, lvl == 1 ? 16 : 32, nq, qox, qoy, qw, qh, t);
                    if (t == 0) {
                        int q = 0;
                        for (int h = 0; h < nh; h++)
                            for (int w = 0; w < nw; w++, q++) {
                                const unsigned long long k = S.hkey[q];
                                const int p = (int)(uint32_t)k, sy = p / qw[q], sx = p - sy * qw[q];
                                S.hx[lvl][w][h] = (int16_t)((int16_t)(sx + qox[q]) * (1 << shf));
                                S.hy[lvl][w][h] = (int16_t)((int16_t)(sy + qoy[q]) * (1 << shf));
                                S.hs[lvl][w][h] = (k >> 32) * 2;
                            }
                    }
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
        hcx[list] = cx, hcy[list] = cy;

        /* ---- EbHevcCheckZeroZeroCenter (:2946-3034) ---- */
        if (cx != 0 || cy != 0) {
            if (P.update_hme_search_center) {
                cx = clamp_center(ox, cx, LCU - 1, W);
                cy = clamp_center(oy, cy, LCU - 1, H);
            }
            int cdx[2] = {0, cx}, cdy[2] = {0, cy};
            lcu_sads<0>(S, R.full, R.pitch_full, ox, oy, lw, lh, 2, cdx, cdy, t);
            const uint32_t zeroSad = S.acc[0] << 1, hmeSad = S.acc[1] << 1;
            const unsigned long long zeroCost = (unsigned long long)zeroSad << COST_PRECISION;
            const uint32_t rate = mvd_fraction_bits(abs(cx << 2), abs(cy << 2), P.mvd_bits);
            const unsigned long long hmeCost = (unsigned long long)(uint32_t)(hmeSad << COST_PRECISION) +
                                               ((((unsigned long long)P.lambda * rate) + MD_OFFSET) >> MD_SHIFT);
            if (zeroCost <= hmeCost)
                cx = 0, cy = 0;
            __syncthreads();
        }

        /* ---- search area (:4072-4200); unrestricted MVs ---- */
        int saw = imin(P.search_area_width, 127), sah = imin(P.search_area_height, 127);
        int sox = cx - (saw >> 1), soy = cy - (sah >> 1);
        clamp_area(ox, LCU - 1, W, sox, saw);
        clamp_area(oy, LCU - 1, H, soy, sah);
        sa_x[list] = sox, sa_y[list] = soy, sa_w[list] = saw, sa_h[list] = sah;

        /* ---- FullPelSearch_LCU (:586-633) ---- */
        {
            if (t < 85)
                S.key[t] = 0xffffffffu;
            if (t == 0)
                S.key64 = ~0ull;
            const int npos = saw * sah, mult8 = saw & ~7;
            const int b = t & 63, sub = t >> 6;
            int bx, by, bsz;
            pu_geom_z(21 + b, bx, by, bsz);
            uint32_t s0[4], s1[4];
#pragma unroll
            for (int r = 0; r < 4; r++) {
                s0[r] = *(const uint32_t *)&S.src[(by + 2 * r) * LCU + bx];
                s1[r] = *(const uint32_t *)&S.src[(by + 2 * r) * LCU + bx + 4];
            }
            uint32_t best8 = 0xffffffffu;
            const uint8_t *rbase = R.full + (ptrdiff_t)(oy + by + soy) * R.pitch_full + ox + bx + sox;
            __syncthreads();
            for (int base = 0; base < npos; base += 64) {
                for (int i = 0; i < 16; i++) {
                    const int pl = sub + 4 * i, p = base + pl;
                    if (p < npos) {
                        const int sy = p / saw, sx = p - sy * saw;
                        const uint8_t *r = rbase + (ptrdiff_t)sy * R.pitch_full + sx;
                        uint32_t s = 0;
#pragma unroll
                        for (int rr = 0; rr < 4; rr++) {
                            s = sad4(s0[rr], ld4(r + (ptrdiff_t)(2 * rr) * R.pitch_full), s);
                            s = sad4(s1[rr], ld4(r + (ptrdiff_t)(2 * rr) * R.pitch_full + 4), s);
                        }
                        S.sad8[pl][b] = (uint16_t)s;
                        const uint32_t k = (s << 14) | (uint32_t)p;
                        best8 = k < best8 ? k : best8;
                    }
                }
                __syncthreads();
                { /* 16x16: thread owns k16 = t&15, positions (t>>4) + 16*i */
                    const int k16 = t & 15;
                    uint32_t bk = 0xffffffffu;
                    for (int i = 0; i < 4; i++) {
                        const int pl = (t >> 4) + 16 * i, p = base + pl;
                        if (p < npos) {
                            const uint16_t *q = &S.sad8[pl][4 * k16];
                            const uint32_t s = (uint32_t)q[0] + q[1] + q[2] + q[3];
                            S.sad16[pl][k16] = (uint16_t)s;
                            const uint32_t k = (s << 14) | (uint32_t)p;
                            bk = k < bk ? k : bk;
                        }
                    }
                    if (bk != 0xffffffffu)
                        atomicMin(&S.key[5 + k16], bk);
                }
                __syncthreads();
                { /* 32x32: one (position, quadrant) per thread */
                    const int pl = t >> 2, k32 = t & 3, p = base + pl;
                    if (p < npos) {
                        const uint16_t *q = &S.sad16[pl][4 * k32];
                        const uint32_t s = (uint32_t)q[0] + q[1] + q[2] + q[3];
                        S.sad32[pl][k32] = s;
                        atomicMin(&S.key[1 + k32], (s << 14) | (uint32_t)p);
                    }
                }
                __syncthreads();
                if (t < 64) { /* 64x64: '<=' inside complete groups of 8, '<' in the tail */
                    const int p = base + t;
                    if (p < npos) {
                        const uint32_t s = S.sad32[t][0] + S.sad32[t][1] + S.sad32[t][2] + S.sad32[t][3];
                        const int sy = p / saw, sx = p - sy * saw;
                        const uint32_t code = (sx < mult8) ? (uint32_t)(16383 - p) : (0x4000u | (uint32_t)p);
                        atomicMin(&S.key64, ((unsigned long long)s << 15) | code);
                    }
                }
                __syncthreads();
            }
            atomicMin(&S.key[21 + b], best8);
            __syncthreads();
            if (t < 85) {
                uint32_t s;
                int p;
                if (t == 0) {
                    const unsigned long long k = S.key64;
                    const uint32_t code = (uint32_t)(k & 0x7fff);
                    s = (uint32_t)(k >> 15);
                    p = (code & 0x4000u) ? (int)(code & 0x3fff) : 16383 - (int)code;
                } else {
                    const uint32_t k = S.key[t];
                    s = k >> 14;
                    p = (int)(k & 0x3fff);
                }
                const int sy = p / saw, sx = p - sy * saw;
                S.best_sad[list][t] = 2 * s;
                S.best_mv[list][t] = mvpack((sx + sox) * 4, (sy + soy) * 4);
            }
            __syncthreads();
        }

        /* ---- sub-pel (:4236-4318) ---- */
        if (t == 0) {
            int e32 = 0, e16 = 0, e8 = 0, eq = 0;
            if (P.fractional_search_model == 0) {
                e32 = e16 = e8 = eq = 1;
            } else if (P.fractional_search_model == 1) {
                /* SuPelEnable (:3035-3361) */
                const int first[3] = {1, 5, 21}, count[3] = {4, 16, 64}, shift[3] = {2, 4, 6};
                uint32_t mag[3], avgsad[3];
                for (int tt = 0; tt < 3; tt++) {
                    int sx = 0, sy = 0;
                    uint32_t ss = 0;
                    for (int k = 0; k < count[tt]; k++) {
                        sx += mvx(S.best_mv[list][first[tt] + k]);
                        sy += mvy(S.best_mv[list][first[tt] + k]);
                        ss += S.best_sad[list][first[tt] + k];
                    }
                    const uint32_t ux = (uint32_t)(sx >> shift[tt]), uy = (uint32_t)(sy >> shift[tt]);
                    mag[tt] = ux * ux + uy * uy;
                    avgsad[tt] = ss >> shift[tt];
                }
                const int tl = P.temporal_layer_index;
                const uint32_t th = tl == 0 ? 48 * 48 : tl == 1 ? 32 * 32 : tl == 2 ? 80 * 80 : 48 * 48;
                const int small32 = mag[0] < th, low32 = avgsad[0] < 32 * 32 * 6;
                e32 = (tl == 0 || tl == 2) ? low32 : (tl == 1 ? (small32 ? low32 : 1) : (small32 ? 1 : low32));
                e16 = !(avgsad[1] < 16 * 16 * 2);
                e8 = (tl <= 2) ? !(avgsad[2] < 8 * 8 * 2) : ((mag[2] < th) ? !(avgsad[2] < 8 * 8 * 2) : 0);
                eq = 1;
            }
            S.e32 = e32, S.e16 = e16 && P.cu16x16_mode == 0, S.e8 = e8 && P.cu8x8_mode != 1, S.eq = eq;
        }
        __syncthreads();
        const int any_sub = S.e32 || S.e16 || S.e8 || S.eq || 0;
        const int run_sub = (P.fractional_search_model != 2) && any_sub;
        if (run_sub) {
            const int f64 = P.fractional_search_64x64;
            const int en[4] = {f64, S.e32, S.e16, S.e8};
            const int tier_first[4] = {0, 1, 5, 21}, tier_cnt[4] = {1, 4, 16, 64}, tier_sz[4] = {64, 32, 16, 8};
            const int rstep = (method == SVT_AMD_SUB_SAD_SEARCH) ? 2 : 1;
            const PicView &RR = R;
            /* ===== half-pel: EbHevcHalfPelSearch_LCU / PU_HalfPelRefinement (:733-1187) ===== */
            for (int i = t; i < 85 * 8; i += NT) {
                (&S.dist[0][0])[i] = 0;
                (&S.dsad[0][0])[i] = 0;
            }
            __syncthreads();
            for (int tier = 0; tier < 4; tier++) {
                if (!en[tier])
                    continue;
                const int sz = tier_sz[tier], rows = sz / rstep, items = tier_cnt[tier] * 8 * rows;
                for (int i = t; i < items; i += NT) {
                    const int row = i % rows, k = (i / rows) & 7, n = tier_first[tier] + i / (rows * 8);
                    int px_, py_, psz;
                    pu_geom_z(n, px_, py_, psz);
                    const uint32_t mv = S.best_mv[list][n];
                    const int ax = ox + px_ + (mvx(mv) >> 2), ay = oy + py_ + (mvy(mv) >> 2) + row * rstep;
                    /* order L,R,T,B,TL,TR,BR,BL: planes b,b,h,h,j,j,j,j; offsets */
                    const uint8_t *pl = (k < 2) ? RR.hp_b : (k < 4 ? RR.hp_h : RR.hp_j);
                    const int ddx = (k == 1 || k == 5 || k == 6) ? 1 : 0, ddy = (k == 3 || k == 6 || k == 7) ? 1 : 0;
                    const uint8_t *r = pl + (ptrdiff_t)(ay + ddy) * RR.pitch_full + ax + ddx;
                    const uint8_t *s = &S.src[(py_ + row * rstep) * LCU + px_];
                    uint32_t d = 0, sd = 0;
                    for (int x = 0; x < sz; x += 4)
                        row_metric(method, *(const uint32_t *)(s + x), ld4(r + x), d, sd);
                    atomicAdd(&S.dist[n][k], d);
                    if (method == SVT_AMD_SSD_SEARCH)
                        atomicAdd(&S.dsad[n][k], sd);
                }
            }
            /* SSD search also needs the SSE of the full-pel winner (:798-806) */
            if (method == SVT_AMD_SSD_SEARCH) {
                for (int tier = 0; tier < 4; tier++) {
                    if (!en[tier])
                        continue;
                    const int sz = tier_sz[tier], items = tier_cnt[tier] * sz;
                    for (int i = t; i < items; i += NT) {
                        const int row = i % sz, n = tier_first[tier] + i / sz;
                        int px_, py_, psz;
                        pu_geom_z(n, px_, py_, psz);
                        const uint32_t mv = S.best_mv[list][n];
                        const uint8_t *r = RR.full + (ptrdiff_t)(oy + py_ + (mvy(mv) >> 2) + row) * RR.pitch_full + ox + px_ + (mvx(mv) >> 2);
                        const uint8_t *s = &S.src[(py_ + row) * LCU + px_];
                        uint32_t d = 0;
                        for (int x = 0; x < sz; x += 4)
                            d += ssd4(*(const uint32_t *)(s + x), ld4(r + x));
                        atomicAdd(&S.best_ssd[list][n], d);
                    }
                }
            }
            __syncthreads();
            if (t < 85) {
                const int tier = t == 0 ? 0 : t < 5 ? 1 : t < 21 ? 2 : 3;
                if (en[tier]) {
                    const int mdx[8] = {-2, 2, 0, 0, -2, 2, 2, -2}, mdy[8] = {0, 0, -2, 2, -2, -2, 2, 2};
                    const uint32_t mv0 = S.best_mv[list][t];
                    uint32_t bsad = S.best_sad[list][t], bmv = mv0, bssd = S.best_ssd[list][t];
                    uint32_t dmin = 0xffffffffu;
                    for (int k = 0; k < 8; k++) {
                        const uint32_t d = (method == SVT_AMD_SUB_SAD_SEARCH) ? (S.dist[t][k] << 1) : S.dist[t][k];
                        dmin = d < dmin ? d : dmin;
                        if (method == SVT_AMD_SSD_SEARCH) {
                            if (d < bssd)
                                bsad = S.dsad[t][k], bmv = mvpack(mvx(mv0) + mdx[k], mvy(mv0) + mdy[k]), bssd = d;
                        } else if (d < bsad) {
                            bsad = d, bmv = mvpack(mvx(mv0) + mdx[k], mvy(mv0) + mdy[k]);
                        }
                    }
                    /* first match in the order L,R,T,B,TL,TR,BL,BR (:1002-1025) */
                    const int chk[8] = {0, 1, 2, 3, 4, 5, 7, 6};
                    const uint8_t code[8] = {D_L, D_R, D_T, D_B, D_TL, D_TR, D_BR, D_BL};
                    uint8_t dirv = 0;
                    for (int i = 7; i >= 0; i--) {
                        const uint32_t d = (method == SVT_AMD_SUB_SAD_SEARCH) ? (S.dist[t][chk[i]] << 1) : S.dist[t][chk[i]];
                        if (d == dmin)
                            dirv = code[chk[i]];
                    }
                    S.best_sad[list][t] = bsad, S.best_mv[list][t] = bmv, S.best_ssd[list][t] = bssd;
                    S.dir[list][t] = dirv;
                }
            }
            __syncthreads();

            /* ===== quarter-pel: QuarterPelSearch_LCU / PU_QuarterPelRefinementOnTheFly (:1226-1846) ===== */
            const int qen[4] = {f64, S.eq && S.e32, S.eq && S.e16, S.eq && S.e8};
            for (int i = t; i < 85 * 8; i += NT) {
                (&S.dist[0][0])[i] = 0;
                (&S.dsad[0][0])[i] = 0;
            }
            __syncthreads();
            for (int tier = 0; tier < 4; tier++) {
                if (!qen[tier])
                    continue;
                const int sz = tier == 0 ? 32 : tier_sz[tier]; /* the 64x64 call passes 32x32 (:1677) */
                const int rows = sz / rstep, items = tier_cnt[tier] * 8 * rows;
                for (int i = t; i < items; i += NT) {
                    const int row = i % rows, k = (i / rows) & 7, n = tier_first[tier] + i / (rows * 8);
                    int px_, py_, psz;
                    pu_geom_z(n, px_, py_, psz);
                    const uint32_t mv = S.best_mv[list][n];
                    const int xMv = mvx(mv), yMv = mvy(mv);
                    const int qm = (yMv & 2) + ((xMv & 2) >> 1);
                    const int sd = S.dir[list][n];
                    /* validity: position k (L,R,T,B,TL,TR,BR,BL) is tested when the half-pel
                     * winner direction lies within +-1 step of it (mirrored when the MV is
                     * already on a half-pel position), :1252-1273 */
                    /* ring order of directions: TL,T,TR,R,BR,B,BL,L == codes 0..7 */
                    const int kcode[8] = {D_L, D_R, D_T, D_B, D_TL, D_TR, D_BR, D_BL};
                    const int target = qm ? ((kcode[k] + 4) & 7) : kcode[k];
                    const int diff = (sd - target) & 7;
                    if (!(diff == 0 || diff == 1 || diff == 7))
                        continue;
                    const int y = row * rstep;
                    const int ax = ox + px_ + ((xMv + 2) >> 2), ay = oy + py_ + ((yMv + 2) >> 2) + y;
                    const QSrc q0 = c_qtab[qm][k][0], q1 = c_qtab[qm][k][1];
                    const uint8_t *planes[4] = {RR.full, RR.hp_b, RR.hp_h, RR.hp_j};
                    const uint8_t *r1 = planes[q0.plane] + (ptrdiff_t)(ay + q0.dy) * RR.pitch_full + ax + q0.dx;
                    const uint8_t *r2 = planes[q1.plane] + (ptrdiff_t)(ay + q1.dy) * RR.pitch_full + ax + q1.dx;
                    /* source = MeContext_t.lcuBuffer: zero outside the picture (trap A19) */
                    const int inside_y = (py_ + y) < lh;
                    const uint8_t *s = &S.src[(py_ + y) * LCU + px_];
                    uint32_t d = 0, sdv = 0;
                    for (int x = 0; x < sz; x += 4) {
                        const uint32_t sv = (inside_y && (px_ + x) < lw) ? *(const uint32_t *)(s + x) : 0u;
                        row_metric(method, sv, avg4(ld4(r1 + x), ld4(r2 + x)), d, sdv);
                    }
                    atomicAdd(&S.dist[n][k], d);
                    if (method == SVT_AMD_SSD_SEARCH)
                        atomicAdd(&S.dsad[n][k], sdv);
                }
            }
            __syncthreads();
            if (t < 85) {
                const int tier = t == 0 ? 0 : t < 5 ? 1 : t < 21 ? 2 : 3;
                if (qen[tier]) {
                    const int mdx[8] = {-1, 1, 0, 0, -1, 1, 1, -1}, mdy[8] = {0, 0, -1, 1, -1, -1, 1, 1};
                    const int kcode[8] = {D_L, D_R, D_T, D_B, D_TL, D_TR, D_BR, D_BL};
                    const uint32_t mv0 = S.best_mv[list][t];
                    const int qm = (mvy(mv0) & 2) + ((mvx(mv0) & 2) >> 1);
                    const int sd = S.dir[list][t];
                    uint32_t bsad = S.best_sad[list][t], bmv = mv0, bssd = S.best_ssd[list][t];
                    for (int k = 0; k < 8; k++) {
                        const int target = qm ? ((kcode[k] + 4) & 7) : kcode[k];
                        const int diff = (sd - target) & 7;
                        if (!(diff == 0 || diff == 1 || diff == 7))
                            continue;
                        const uint32_t d = (method == SVT_AMD_SUB_SAD_SEARCH) ? (S.dist[t][k] << 1) : S.dist[t][k];
                        if (method == SVT_AMD_SSD_SEARCH) {
                            if (d < bssd)
                                bsad = S.dsad[t][k], bmv = mvpack(mvx(mv0) + mdx[k], mvy(mv0) + mdy[k]), bssd = d;
                        } else if (d < bsad) {
                            bsad = d, bmv = mvpack(mvx(mv0) + mdx[k], mvy(mv0) + mdy[k]);
                        }
                    }
                    S.best_sad[list][t] = bsad, S.best_mv[list][t] = bmv, S.best_ssd[list][t] = bssd;
                }
            }
            __syncthreads();
        }
    } /* lists */

    /* ---- bi-prediction (:2608-2917) ---- */
    if (P.num_lists == 2) {
        const int rstep = (method == SVT_AMD_SUB_SAD_SEARCH) ? 2 : 1;
        const int npu = (P.cu16x16_mode != 0) ? 5 : ((P.cu8x8_mode != 0) ? 21 : 85);
        /* items: (pu n, row) */
        int first = 0;
        const int tier_first[4] = {0, 1, 5, 21}, tier_cnt[4] = {1, 4, 16, 64}, tier_sz[4] = {64, 32, 16, 8};
        (void)first;
        for (int tier = 0; tier < 4; tier++) {
            if (tier_first[tier] >= npu)
                break;
            const int sz = tier_sz[tier], rows = sz / rstep, items = tier_cnt[tier] * rows;
            for (int i = t; i < items; i += NT) {
                const int row = i % rows, n = tier_first[tier] + i / rows;
                int px_, py_, psz;
                pu_geom_z(n, px_, py_, psz);
                const int y = row * rstep;
                const uint8_t *a[2], *b[2];
                for (int l = 0; l < 2; l++) {
                    const PicView &RR = l ? ref1 : ref0;
                    const uint32_t mv = S.best_mv[l][n];
                    const int xMv = mvx(mv), yMv = mvy(mv);
                    const int ax = ox + px_ + (xMv >> 2), ay = oy + py_ + (yMv >> 2) + y;
                    const int frac = (xMv & 3) + ((yMv & 3) << 2);
                    const ptrdiff_t pz = RR.pitch_full;
                    const uint8_t *F = RR.full + ay * pz + ax, *B = RR.hp_b + ay * pz + ax + 1;
                    const uint8_t *Hh = RR.hp_h + (ay + 1) * pz + ax, *J = RR.hp_j + (ay + 1) * pz + ax + 1;
                    /* SelectBuffer / QuarterPelCompensation (:2440-2600) */
                    switch (frac) {
                    case 0: a[l] = F, b[l] = F; break;
                    case 2: a[l] = B, b[l] = B; break;
                    case 8: a[l] = Hh, b[l] = Hh; break;
                    case 10: a[l] = J, b[l] = J; break;
                    case 1: a[l] = F, b[l] = B; break;
                    case 3: a[l] = B, b[l] = F + 1; break;
                    case 4: a[l] = F, b[l] = Hh; break;
                    case 5: a[l] = B, b[l] = Hh; break;
                    case 6: a[l] = B, b[l] = J; break;
                    case 7: a[l] = B, b[l] = Hh + 1; break;
                    case 9: a[l] = Hh, b[l] = J; break;
                    case 11: a[l] = J, b[l] = Hh + 1; break;
                    case 12: a[l] = Hh, b[l] = F + pz; break;
                    case 13: a[l] = Hh, b[l] = B + pz; break;
                    case 14: a[l] = J, b[l] = B + pz; break;
                    default: a[l] = Hh + 1, b[l] = B + pz; break;
                    }
                }
                const uint8_t *s = &S.src[(py_ + y) * LCU + px_];
                uint32_t d = 0;
                for (int x = 0; x < sz; x += 4) {
                    /* avg of identical pointers is the identity: (v+v+1)>>1 == v */
                    const uint32_t p0 = avg4(ld4(a[0] + x), ld4(b[0] + x));
                    const uint32_t p1 = avg4(ld4(a[1] + x), ld4(b[1] + x));
                    d = sad4(*(const uint32_t *)(s + x), avg4(p0, p1), d);
                }
                atomicAdd(&S.bipred[n], d);
            }
        }
        __syncthreads();
    }

    /* ---- candidate records (:4321-4440) ---- */
    SvtAmdMeLcuResult *o = &out[lcu];
    if (t < 85) {
        const int pu = t;
        const int n = pu == 0 ? 0 : pu < 5 ? pu : pu < 21 ? c_tab16[pu - 5] + 5 : c_tab8[pu - 21] + 21;
        int total = P.num_lists;
        if (P.num_lists == 2 && (P.cu8x8_mode == 0 || pu < 21) && (P.cu16x16_mode == 0 || pu < 5))
            total = 3;
        const uint32_t bi = (method == SVT_AMD_SUB_SAD_SEARCH) ? (S.bipred[n] << 1) : S.bipred[n];
        const uint32_t v[3] = {S.best_sad[0][n], S.best_sad[1][n], bi};
        SvtAmdMeCuResult r;
        r.x_mv_l0 = (int16_t)mvx(S.best_mv[0][n]), r.y_mv_l0 = (int16_t)mvy(S.best_mv[0][n]);
        r.x_mv_l1 = (int16_t)mvx(S.best_mv[1][n]), r.y_mv_l1 = (int16_t)mvy(S.best_mv[1][n]);
        r.total_me_candidate_index = (uint8_t)total;
        for (int k = 0; k < 3; k++)
            r.distortion[k] = 0, r.direction[k] = 0;
        if (total == 3) {
            /* Sort3Elements (:2919-2944) */
            int o3[3];
            const uint32_t a = v[0], b = v[1], c = v[2];
            if (a <= b && a <= c) {
                o3[0] = 0;
                if (b <= c) o3[1] = 1, o3[2] = 2; else o3[1] = 2, o3[2] = 1;
            } else if (b <= a && b <= c) {
                o3[0] = 1;
                if (a <= c) o3[1] = 0, o3[2] = 2; else o3[1] = 2, o3[2] = 0;
            } else if (a <= b) {
                o3[0] = 2, o3[1] = 0, o3[2] = 1;
            } else {
                o3[0] = 2, o3[1] = 1, o3[2] = 0;
            }
            for (int k = 0; k < 3; k++)
                r.distortion[k] = v[o3[k]], r.direction[k] = (uint8_t)o3[k];
        } else if (total == 2) {
            const int f = v[0] <= v[1] ? 0 : 1;
            r.distortion[0] = v[f], r.direction[0] = (uint8_t)f;
            r.distortion[1] = v[1 - f], r.direction[1] = (uint8_t)(1 - f);
        } else {
            r.distortion[0] = v[0], r.direction[0] = SVT_AMD_UNI_PRED_LIST_0;
        }
        o->pu[pu] = r;
        o->best_sad[0][t] = S.best_sad[0][t];
        o->best_sad[1][t] = S.best_sad[1][t];
        o->best_mv[0][t] = S.best_mv[0][t];
        o->best_mv[1][t] = S.best_mv[1][t];
    }
    if (t < 2) {
        o->hme_center_x[t] = (int16_t)hcx[t];
        o->hme_center_y[t] = (int16_t)hcy[t];
        o->search_origin_x[t] = (int16_t)sa_x[t];
        o->search_origin_y[t] = (int16_t)sa_y[t];
        o->search_w[t] = (uint8_t)sa_w[t];
        o->search_h[t] = (uint8_t)sa_h[t];
    }
}

int svt_amd_launch_me(SvtAmdContext *ctx, const SvtAmdMeParams *p, const DevPicture *cur, const DevPicture *ref0,
                      const DevPicture *ref1, SvtAmdMeLcuResult *d_out, int lcu_begin, int lcu_end)
{
    const int nlcu = lcu_end - lcu_begin;
    int rc = svt_amd_stamp_begin(ctx, KC_ME_SEARCH);
    if (rc)
        return rc;
    hipLaunchKernelGGL(k_me_picture, dim3((unsigned)nlcu), dim3(NT), 0, ctx->stream, *p, make_view(cur),
                       make_view(ref0), make_view(ref1), d_out, lcu_begin);
    HIP_TRY(hipGetLastError());
    return svt_amd_stamp_end(ctx);
}
