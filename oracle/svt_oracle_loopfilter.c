/*
 * oracle/svt_oracle_loopfilter.c - TEST INFRASTRUCTURE ONLY (see svt_oracle.h).
 * CPU restatement of the in-loop filter and bit-depth packing leaf kernels:
 *   deblocking edge cores   C_DEFAULT/EbDeblockingFilter_C.c:39-577
 *   SAO statistics + apply  C_DEFAULT/EbSampleAdaptiveOffset_C.c:23-861
 *   10-bit pack / unpack    C_DEFAULT/EbPackUnPack_C.c:12-251
 * bps = bytes per sample (1: 8-bit kernels, 2: the *16bit kernels, 10-bit ranges).
 */
#include <stdlib.h>
#include <string.h>
#include "svt_oracle.h"

#define GET(p, i) (bps == 1 ? (int)((const uint8_t *)(p))[i] : (int)((const uint16_t *)(p))[i])
#define PUT(p, i, v)                              \
    do {                                          \
        if (bps == 1)                             \
            ((uint8_t *)(p))[i] = (uint8_t)(v);   \
        else                                      \
            ((uint16_t *)(p))[i] = (uint16_t)(v); \
    } while (0)
static inline int clip3(int lo, int hi, int v) { return v < lo ? lo : v > hi ? hi : v; }
static inline int sgn(int a, int b) { return (a - b) < 0 ? -1 : (a - b) > 0 ? 1 : 0; } /* SIGN, EbUtility.h:102 */

/* Luma4SampleEdgeDLFCore(16bit), EbDeblockingFilter_C.c:39-238, 240-438.  `edge` points at q0 of the
 * first of the 4 edge samples; p side is at negative offsets along the filter direction. */
void svt_oracle_Luma4SampleEdgeDLFCore(int bps, void *edge, uint32_t stride, int isVerticalEdge, int32_t tc, int32_t beta)
{
    const int maxv = bps == 1 ? 255 : 1023;
    const int fs = isVerticalEdge ? 1 : (int)stride, ns = isVerticalEdge ? (int)stride : 1;
#define S(k, line) GET(edge, (k) * fs + (line) * ns) /* k >= 0: q_k, k < 0: p_{-k-1} */
    const int dp0 = abs(S(-3, 0) - 2 * S(-2, 0) + S(-1, 0)), dp3 = abs(S(-3, 3) - 2 * S(-2, 3) + S(-1, 3));
    const int dq0 = abs(S(2, 0) - 2 * S(1, 0) + S(0, 0)), dq3 = abs(S(2, 3) - 2 * S(1, 3) + S(0, 3));
    const int dp = dp0 + dp3, dq = dq0 + dq3, d0 = dp0 + dq0, d3 = dp3 + dq3, d = d0 + d3;
    if (d >= beta)
        return;
    int strong = 1;
    for (int line = 0; line < 4; line += 3) {
        const int dl = line ? d3 : d0;
        strong = strong && ((dl << 1) < (beta >> 2)) &&
                 (beta >> 3) > (abs(S(-4, line) - S(-1, line)) + abs(S(3, line) - S(0, line))) &&
                 ((5 * tc + 1) >> 1) > abs(S(-1, line) - S(0, line));
    }
    for (int c = 0; c < 4; c++) {
        const int q0 = S(0, c), q1 = S(1, c), q2 = S(2, c), q3 = S(3, c);
        const int p0 = S(-1, c), p1 = S(-2, c), p2 = S(-3, c), p3 = S(-4, c);
#define W(k, v) PUT(edge, (k) * fs + c * ns, v)
        if (strong) {
            W(0, clip3(q0 - 2 * tc, q0 + 2 * tc, (p1 + 2 * p0 + 2 * q0 + 2 * q1 + q2 + 4) >> 3));
            W(-1, clip3(p0 - 2 * tc, p0 + 2 * tc, (p2 + 2 * p1 + 2 * p0 + 2 * q0 + q1 + 4) >> 3));
            W(1, clip3(q1 - 2 * tc, q1 + 2 * tc, (p0 + q0 + q1 + q2 + 2) >> 2));
            W(-2, clip3(p1 - 2 * tc, p1 + 2 * tc, (p2 + p1 + p0 + q0 + 2) >> 2));
            W(2, clip3(q2 - 2 * tc, q2 + 2 * tc, (p0 + q0 + q1 + 3 * q2 + 2 * q3 + 4) >> 3));
            W(-3, clip3(p2 - 2 * tc, p2 + 2 * tc, (2 * p3 + 3 * p2 + p1 + p0 + q0 + 4) >> 3));
        } else {
            int delta = ((q0 - p0) * 9 - (q1 - p1) * 3 + 8) >> 4;
            if (abs(delta) < tc * 10) {
                delta = clip3(-tc, tc, delta);
                W(0, clip3(0, maxv, q0 - delta));
                W(-1, clip3(0, maxv, p0 + delta));
                const int side = (beta + (beta >> 1)) >> 3, tc2 = tc >> 1;
                if (side > dp)
                    W(-2, clip3(0, maxv, p1 + clip3(-tc2, tc2, ((((p0 + p2 + 1) >> 1) - p1 + delta) >> 1))));
                if (side > dq)
                    W(1, clip3(0, maxv, q1 + clip3(-tc2, tc2, ((((q0 + q2 + 1) >> 1) - q1 - delta) >> 1))));
            }
        }
#undef W
    }
#undef S
}

/* Chroma2SampleEdgeDLFCore(16bit), EbDeblockingFilter_C.c:469-577 */
void svt_oracle_Chroma2SampleEdgeDLFCore(int bps, void *cb, void *cr, uint32_t stride, int isVerticalEdge,
                                         uint8_t cbTc, uint8_t crTc)
{
    const int maxv = bps == 1 ? 255 : 1023;
    const int fs = isVerticalEdge ? 1 : (int)stride, ns = isVerticalEdge ? (int)stride : 1;
    for (int plane = 0; plane < 2; plane++) {
        void *e = plane ? cr : cb;
        const int tc = plane ? crTc : cbTc;
        for (int c = 0; c < 2; c++) {
            const int q0 = GET(e, c * ns), q1 = GET(e, c * ns + fs), p0 = GET(e, c * ns - fs), p1 = GET(e, c * ns - 2 * fs);
            const int16_t delta = (int16_t)clip3(-tc, tc, ((((q0 - p0) << 2) + p1 - q1 + 4) >> 3));
            PUT(e, c * ns - fs, clip3(0, maxv, p0 + delta));
            PUT(e, c * ns, clip3(0, maxv, q0 - delta));
        }
    }
}

/* GatherSaoStatisticsLcu_62x62_16bit (:23), GatherSaoStatisticsLcuLossy_62x62 (:125),
 * ..._OnlyEo_90_45_135_Lossy (:227), ..._62x62_OnlyEo_90_45_135_16bit (:311).
 * Statistics over the interior (W-2)x(H-2); eo index order 0:0deg 1:90 2:135 3:45; categories
 * compacted (slot 2 <- 3, 3 <- 4) at the end; 8-bit variants clip diff to [-128,127]. */
void svt_oracle_GatherSaoStatistics(int bps, int only_eo_90_45_135, const void *input, uint32_t inputStride,
                                    const void *recon, uint32_t reconStride, uint32_t lcuWidth, uint32_t lcuHeight,
                                    int32_t *boDiff, uint16_t *boCount, int32_t eoDiff[4][5], uint16_t eoCount[4][5])
{
    const int boShift = bps == 1 ? 3 : 5;
    if (!only_eo_90_45_135)
        for (int i = 0; i < 32; i++)
            boDiff[i] = 0, boCount[i] = 0;
    for (int t = 0; t < 4; t++)
        for (int k = 0; k < 5; k++)
            eoDiff[t][k] = 0, eoCount[t][k] = 0;
    const int rs = (int)reconStride;
    for (uint32_t j = 1; j + 1 < lcuHeight; j++)
        for (uint32_t i = 1; i + 1 < lcuWidth; i++) {
            const int at = (int)(j * reconStride + i), r = GET(recon, at);
            int diff = GET(input, j * inputStride + i) - r;
            if (bps == 1)
                diff = clip3(-128, 127, diff);
            if (!only_eo_90_45_135) {
                boDiff[r >> boShift] += diff;
                boCount[r >> boShift]++;
            }
            const int nb[4][2] = {{-1, 1}, {-rs, rs}, {-rs - 1, rs + 1}, {-rs + 1, rs - 1}};
            for (int t = only_eo_90_45_135 ? 1 : 0; t < 4; t++) {
                const int idx = sgn(r, GET(recon, at + nb[t][0])) + sgn(r, GET(recon, at + nb[t][1])) + 2;
                eoDiff[t][idx] += diff;
                eoCount[t][idx]++;
            }
        }
    for (int t = 0; t < 4; t++) {
        eoDiff[t][2] = eoDiff[t][3], eoDiff[t][3] = eoDiff[t][4];
        eoCount[t][2] = eoCount[t][3], eoCount[t][3] = eoCount[t][4];
    }
}

/* SAOApplyBO(16bit), EbSampleAdaptiveOffset_C.c:394-468 */
void svt_oracle_SAOApplyBO(int bps, void *recon, uint32_t stride, uint32_t bandPosition, const int8_t *offset,
                           uint32_t lcuHeight, uint32_t lcuWidth)
{
    const int maxv = bps == 1 ? 255 : 1023, shift = bps == 1 ? 3 : 5;
    for (uint32_t y = 0; y < lcuHeight; y++)
        for (uint32_t x = 0; x < lcuWidth; x++) {
            const int v = GET(recon, y * stride + x);
            const uint32_t bo = (uint32_t)v >> shift;
            if (!(bo < bandPosition || bo > bandPosition + 4 - 1)) /* SAO_BO_LEN = 4 */
                PUT(recon, y * stride + x, clip3(0, maxv, v + offset[bo - bandPosition]));
        }
}

/* SAOApplyEO_0 / _90 / _135 / _45 (+16bit), EbSampleAdaptiveOffset_C.c:470-861.  The reference
 * works in place and carries neighbour signs; equivalently every sample is classified against the
 * ORIGINAL neighbours, where column -1 comes from temporalBufferLeft[y], row -1 from
 * temporalBufferUpper[x] (x = -1..W), and row H / column W are the untouched picture samples. */
void svt_oracle_SAOApplyEO(int bps, int eoType, void *recon, uint32_t stride, const void *left, const void *upper,
                           const int8_t *offset, uint32_t lcuHeight, uint32_t lcuWidth)
{
    const int maxv = bps == 1 ? 255 : 1023;
    const int W = (int)lcuWidth, H = (int)lcuHeight, os = W + 2;
    int *orig = (int *)malloc(sizeof(int) * (size_t)(W + 2) * (size_t)(H + 2));
#define O(x, y) orig[((y) + 1) * os + (x) + 1]
    for (int y = -1; y <= H; y++)
        for (int x = -1; x <= W; x++) {
            int v = 0;
            if (y == -1)
                v = upper ? GET(upper, x) : 0;
            else if (x == -1)
                v = left ? GET(left, y) : 0;
            else
                v = GET(recon, y * (int)stride + x);
            O(x, y) = v;
        }
    static const int dx[4][2] = {{-1, 1}, {0, 0}, {-1, 1}, {1, -1}}, dy[4][2] = {{0, 0}, {-1, 1}, {-1, 1}, {-1, 1}};
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++) {
            const int c = O(x, y);
            const int idx = sgn(c, O(x + dx[eoType][0], y + dy[eoType][0])) + sgn(c, O(x + dx[eoType][1], y + dy[eoType][1])) + 2;
            PUT(recon, y * (int)stride + x, clip3(0, maxv, c + offset[idx]));
        }
#undef O
    free(orig);
}

/* EbPackUnPack_C.c */
void svt_oracle_msbPack2D(const uint8_t *in8, uint32_t in8Stride, const uint8_t *inn, uint16_t *out16, uint32_t innStride,
                          uint32_t outStride, uint32_t w, uint32_t h) /* EB_ENC_msbPack2D :12 */
{
    for (uint32_t j = 0; j < h; j++)
        for (uint32_t k = 0; k < w; k++)
            out16[k + j * outStride] = (uint16_t)((in8[k + j * in8Stride] << 2) | ((inn[k + j * innStride] >> 6) & 3));
}
void svt_oracle_CompressedPackmsb(const uint8_t *in8, uint32_t in8Stride, const uint8_t *inn, uint16_t *out16,
                                  uint32_t innStride, uint32_t outStride, uint32_t w, uint32_t h) /* :40 */
{
    for (uint32_t j = 0; j < h; j++)
        for (uint32_t k = 0; k < w / 4; k++) {
            const uint8_t four = inn[k + j * innStride];
            for (int i = 0; i < 4; i++)
                out16[k * 4 + i + j * outStride] = (uint16_t)((in8[k * 4 + i + j * in8Stride] << 2) | ((four >> (6 - 2 * i)) & 3));
        }
}
void svt_oracle_CPack(const uint8_t *inn, uint32_t innStride, uint8_t *out, uint32_t outStride, uint32_t w, uint32_t h) /* CPack_C :83 */
{
    for (uint32_t r = 0; r < h; r++)
        for (uint32_t c = 0; c < w; c += 4) {
            const uint32_t i = c + r * innStride;
            out[c / 4 + r * outStride] = (uint8_t)(((inn[i] >> 0) & 0xC0) | ((inn[i + 1] >> 2) & 0x30) |
                                                   ((inn[i + 2] >> 4) & 0x0C) | ((inn[i + 3] >> 6) & 0x03));
        }
}
void svt_oracle_msbUnPack2D(const uint16_t *in16, uint32_t inStride, uint8_t *out8, uint8_t *outn, uint32_t out8Stride,
                            uint32_t outnStride, uint32_t w, uint32_t h) /* EB_ENC_msbUnPack2D :116; out8 only: UnPack8BitData :143 */
{
    for (uint32_t j = 0; j < h; j++)
        for (uint32_t k = 0; k < w; k++) {
            const uint16_t p = in16[k + j * inStride];
            out8[k + j * out8Stride] = (uint8_t)(p >> 2);
            if (outn)
                outn[k + j * outnStride] = (uint8_t)(p << 6);
        }
}
void svt_oracle_UnpackAvg(const uint16_t *l0, uint32_t s0, const uint16_t *l1, uint32_t s1, uint8_t *dst,
                          uint32_t dstStride, uint32_t w, uint32_t h) /* UnpackAvg :165 */
{
    for (uint32_t j = 0; j < h; j++)
        for (uint32_t k = 0; k < w; k++)
            dst[k + j * dstStride] = (uint8_t)(((uint8_t)(l0[k + j * s0] >> 2) + (uint8_t)(l1[k + j * s1] >> 2) + 1) >> 1);
}

/* ------------------------------------------------------------------------------------------------------------------
 * Picture-level deblocking: what the three per-LCU drivers LCUInternalAreaDLFCore / LCUBoundaryDLFCore /
 * LCUPictureEdgeDLFCore (+16bit; Codec/EbDeblockingFilter.c:2222-4330, called per LCU from EbCodingLoop.c:4600-4631)
 * leave in the reconstructed picture once every LCU has been through them: all vertical edges of the 8x8 grid, then all
 * horizontal edges (H.265 8.7.2; the drivers' internal-area / boundary / picture-edge split only orders the same edge
 * set so that this holds LCU by LCU).  Edge strength from the per-LCU arrays (index = 4x4 block raster inside the LCU,
 * EbDeblockingFilter.h:19-22), tc / beta from the mean qp of the two sides (tables :30-38; qp per 8x8 block,
 * :23-24), chroma edges on the 8-sample chroma grid where bS > 1 with the mapped chroma qp (:21-25), 4:2:0.
 * Pinned by tests/test_oracle_dlf_golden.py against whole pictures of real encoder runs.
 * ------------------------------------------------------------------------------------------------------------------ */
static const uint8_t kTc[54] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 1, 1, 1, 1, 1,
                                2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 5, 5, 6, 6, 7, 8, 9, 10, 11, 13, 14, 16, 18, 20, 22, 24};
static const uint8_t kBeta[52] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15,
                                  16, 17, 18, 20, 22, 24, 26, 28, 30, 32, 34, 36, 38, 40, 42, 44, 46, 48, 50, 52, 54, 56, 58, 60, 62, 64};
static const uint8_t kChromaQpMap[58] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21, 22, 23, 24, 25, 26, 27, 28,
                                         29, 29, 30, 31, 32, 33, 33, 34, 34, 35, 35, 36, 36, 37, 37, 38, 39, 40, 41, 42, 43, 44, 45, 46, 47, 48, 49, 50, 51};

static int chroma_tc(int qpMean, int qpOffset, int tcOffset)
{
    const int q = qpMean + qpOffset;
    const uint8_t qc = (uint8_t)(q < 0 ? q : q > 57 ? q - 6 : kChromaQpMap[q]); /* convertToChromaQp into an EB_U8 */
    return kTc[clip3(0, 53, (int)qc + 2 + tcOffset)];
}

void svt_oracle_dlf_picture(int bps, void *y, uint32_t strideY, void *cb, void *cr, uint32_t strideC, uint32_t width,
                            uint32_t height, const uint8_t *bs_v, const uint8_t *bs_h, const uint8_t *qp, uint32_t qpStride,
                            int tcOffset, int betaOffset, int cbQpOffset, int crQpOffset)
{
    const uint32_t lcuCols = (width + 63) >> 6, scale = bps == 1 ? 0 : 2;
    uint8_t *Y = (uint8_t *)y, *CB = (uint8_t *)cb, *CR = (uint8_t *)cr;
    for (int dir = 0; dir < 2; dir++) { /* 0: vertical edges, 1: horizontal edges */
        const uint8_t *bs = dir ? bs_h : bs_v;
        /* luma: 4-sample segments of the 8x8 grid */
        for (uint32_t py = dir ? 8 : 0; py < height; py += dir ? 8 : 4)
            for (uint32_t px = dir ? 0 : 8; px < width; px += dir ? 4 : 8) {
                const uint32_t lcu = (py >> 6) * lcuCols + (px >> 6);
                const int b = bs[lcu * 256 + ((px & 63) >> 2) + (((py & 63) >> 2) << 4)];
                if (!b)
                    continue;
                const int qq = qp[(px >> 3) + (py >> 3) * qpStride];
                const int qpp = dir ? qp[(px >> 3) + ((py - 1) >> 3) * qpStride] : qp[((px - 1) >> 3) + (py >> 3) * qpStride];
                const int Q = (qq + qpp + 1) >> 1;
                const int tc = kTc[clip3(0, 53, Q + ((b > 1) << 1) + tcOffset)] << scale;
                const int beta = kBeta[clip3(0, 51, Q + betaOffset)] << scale;
                svt_oracle_Luma4SampleEdgeDLFCore(bps, Y + ((size_t)py * strideY + px) * bps, strideY, !dir, tc, beta);
            }
        /* chroma (4:2:0): 2-sample segments of the 8x8 chroma grid, bS of the co-located luma 4x4 block */
        const uint32_t cw = width >> 1, ch = height >> 1;
        for (uint32_t cy = dir ? 8 : 0; cy < ch; cy += dir ? 8 : 2)
            for (uint32_t cx = dir ? 0 : 8; cx < cw; cx += dir ? 2 : 8) {
                const uint32_t lcu = (cy >> 5) * lcuCols + (cx >> 5);
                const int b = bs[lcu * 256 + ((cx & 31) >> 1) + (((cy & 31) >> 1) << 4)];
                if (b <= 1)
                    continue;
                const int qq = qp[((2 * cx) >> 3) + ((2 * cy) >> 3) * qpStride];
                const int qpp = dir ? qp[((2 * cx) >> 3) + ((2 * (cy - 1)) >> 3) * qpStride]
                                    : qp[((2 * (cx - 1)) >> 3) + ((2 * cy) >> 3) * qpStride];
                const int Q = (qq + qpp + 1) >> 1;
                const int cbTc = (uint8_t)(chroma_tc(Q, cbQpOffset, tcOffset) << scale);
                const int crTc = (uint8_t)(chroma_tc(Q, crQpOffset, tcOffset) << scale);
                const size_t off = ((size_t)cy * strideC + cx) * bps;
                svt_oracle_Chroma2SampleEdgeDLFCore(bps, CB + off, CR + off, strideC, !dir, cbTc, crTc);
            }
    }
}

/* ------------------------------------------------------------------------------------------------------------------
 * Picture-level SAO application: what ApplySaoOffsetsPicture(16bit) -> ApplySaoOffsetsLcu(16bit)
 * (Codec/EbEncDecProcess.c:522-757, :215-517, 16-bit twins :762-1330) leave in the reconstructed picture.  The reference
 * filters LCU by LCU in place, keeping the previous LCU's last column and the previous LCU row's last row aside so every
 * sample is classified against UNFILTERED neighbours - i.e. an out-of-place filter, which is what this restates.
 * Per LCU and component: type 1..4 = edge offset 0 / 90 / 135 / 45 degrees with offsets {o0, o1, 0, o2, o3} indexed by
 * sign(c - a) + sign(c - b) + 2 (:263-270), type 5 = band offset for the 4 bands from saoBandPosition (:497-506);
 * samples on a tile edge in the direction the class looks at keep their value (:283-318 etc.); clip to the bit depth.
 * params: one record per LCU = {type[2] (luma, chroma), offset[3][4], band[3]}; edge_flags per LCU: 1 left, 2 right,
 * 4 top, 8 bottom tile edge.  Pinned by tests/test_oracle_dlf_golden.py on whole pictures of real encoder runs.
 * ------------------------------------------------------------------------------------------------------------------ */
void svt_oracle_sao_apply_picture(int bps, const void *const src[3], void *const dst[3], uint32_t strideY, uint32_t strideC,
                                  uint32_t width, uint32_t height, const SvtOracleSaoLcu *lcus, int lumaOn, int chromaOn)
{
    const int maxv = bps == 1 ? 255 : 1023, boShift = bps == 1 ? 3 : 5;
    const uint32_t lcuCols = (width + 63) >> 6;
    for (int comp = 0; comp < 3; comp++) {
        const uint32_t sh = comp ? 1 : 0, W = width >> sh, H = height >> sh, stride = comp ? strideC : strideY, L = 64 >> sh;
        const int on = comp ? chromaOn : lumaOn;
        for (uint32_t y = 0; y < H; y++)
            for (uint32_t x = 0; x < W; x++) {
                const size_t i = (size_t)y * stride + x;
                const int c = GET(src[comp], i);
                const SvtOracleSaoLcu *p = &lcus[(y / L) * lcuCols + (x / L)];
                const uint32_t type = on ? p->type[comp ? 1 : 0] : 0;
                int v = c;
                if (type == 5) {
                    const int band = c >> boShift, pos = (int)p->band[comp];
                    if (band >= pos && band <= pos + 3)
                        v = clip3(0, maxv, c + (int8_t)p->offset[comp][band - pos]);
                } else if (type >= 1 && type <= 4) {
                    const uint32_t lx = x % L, ly = y % L, lw = (W - (x - lx)) < L ? (W - (x - lx)) : L, lh = (H - (y - ly)) < L ? (H - (y - ly)) : L;
                    const int atL = lx == 0 && (p->edge_flags & 1), atR = lx == lw - 1 && (p->edge_flags & 2);
                    const int atT = ly == 0 && (p->edge_flags & 4), atB = ly == lh - 1 && (p->edge_flags & 8);
                    const int skip = type == 1 ? (atL || atR) : type == 2 ? (atT || atB) : (atL || atR || atT || atB);
                    if (!skip) {
                        const int dx = type == 2 ? 0 : (type == 4 ? 1 : -1), dy = type == 1 ? 0 : -1;
                        const int a = GET(src[comp], i + (ptrdiff_t)dy * stride + dx), b = GET(src[comp], i - (ptrdiff_t)dy * stride - dx);
                        const int8_t o[5] = {(int8_t)p->offset[comp][0], (int8_t)p->offset[comp][1], 0, (int8_t)p->offset[comp][2],
                                             (int8_t)p->offset[comp][3]};
                        v = clip3(0, maxv, c + o[sgn(c, a) + sgn(c, b) + 2]);
                    }
                }
                PUT(dst[comp], i, v);
            }
    }
}

/* ------------------------------------------------------------------------------------------------------------------
 * Picture-level boundary-strength derivation: what SetBSArrayBasedOnPUBoundary (Codec/EbDeblockingFilter.c:339-442) and
 * SetBSArrayBasedOnTUBoundary (:472-530) with CalculateBSForPUBoundary (:109-335) / CalculateBSForTUBoundaryInsidePU
 * (:444-468), called per coding unit from the encode pass, leave in the per-LCU arrays.  Inputs as picture-level maps: one
 * entry per 8x8 block (prediction mode 1 inter / 2 intra, inter direction, log2 of the coding-unit size, the two motion
 * vectors), the luma cbf per 4x4 block, slice type, the two reference POCs, per-LCU tile-edge flags (1 left, 2 top).
 * Only 2Nx2N prediction units exist; transform-unit edges inside a coding unit occur only in 64x64 units (four 32x32).
 * Output layout = the reference's: [lcu raster][4x4 block raster inside the LCU]; entries the reference never writes
 * stay 0.  Pinned by tests/test_oracle_dlf_golden.py against the arrays of real encoder runs.
 * ------------------------------------------------------------------------------------------------------------------ */
static int mv_far(const int16_t a[2], const int16_t b[2])
{
    return abs(a[0] - b[0]) >= 4 || abs(a[1] - b[1]) >= 4; /* CHECK_MV_COMPONENT_EQUAL_OR_GREATER_THAN_4 (EbDeblockingFilter.h:25) */
}

static int bs_pu_edge(const SvtOracleCuMapEntry *cur, const SvtOracleCuMapEntry *nb, int cbfAny, int sliceType, const uint64_t poc[2])
{
    if (cur->mode == 2 || nb->mode == 2)
        return 2;
    if (sliceType == 1) /* EB_P_PICTURE: one list, one reference */
        return mv_far(cur->mv[0], nb->mv[0]) | cbfAny;
    int c1;
    switch (cur->dir + nb->dir * 3) {
    case 0: c1 = mv_far(cur->mv[0], nb->mv[0]); break;                                   /* L0 / L0: same picture */
    case 1: c1 = poc[1] != poc[0] || mv_far(cur->mv[1], nb->mv[0]); break;
    case 3: c1 = poc[0] != poc[1] || mv_far(cur->mv[0], nb->mv[1]); break;
    case 4: c1 = mv_far(cur->mv[1], nb->mv[1]); break;
    case 8:
        if (poc[0] == poc[1]) /* all four equalities of :243-244 hold */
            c1 = (mv_far(cur->mv[0], nb->mv[0]) || mv_far(cur->mv[1], nb->mv[1])) && (mv_far(cur->mv[0], nb->mv[1]) || mv_far(cur->mv[1], nb->mv[0]));
        else                  /* list-wise equal (:267): the same two pictures on both sides */
            c1 = mv_far(cur->mv[0], nb->mv[0]) || mv_far(cur->mv[1], nb->mv[1]);
        break;
    default: c1 = 1; break;                                                              /* one side bi, the other uni */
    }
    return c1 | cbfAny;
}

void svt_oracle_bs_picture(const SvtOracleCuMapEntry *map, const uint8_t *cbf, uint32_t width, uint32_t height, int sliceType,
                           const uint64_t refPoc[2], const uint8_t *lcuEdge, uint8_t *bs_v, uint8_t *bs_h)
{
    const uint32_t bw = width >> 3, bh = height >> 3, cw = width >> 2, lcuCols = (width + 63) >> 6, nlcu = lcuCols * ((height + 63) >> 6);
    memset(bs_v, 0, (size_t)nlcu * 256);
    memset(bs_h, 0, (size_t)nlcu * 256);
    for (int dir = 0; dir < 2; dir++) /* 0: vertical edges (left neighbour), 1: horizontal edges (upper neighbour) */
        for (uint32_t by = 0; by < bh; by++)
            for (uint32_t bx = 0; bx < bw; bx++) {
                const SvtOracleCuMapEntry *cur = &map[by * bw + bx];
                const uint32_t x = bx << 3, y = by << 3, size = 1u << cur->size_log2, pos = dir ? y : x;
                const uint32_t lcu = (y >> 6) * lcuCols + (x >> 6);
                const int cuEdge = (pos & (size - 1)) == 0;                 /* the block's left / top side is its coding unit's */
                const int tuEdge = !cuEdge && size == 64 && (pos & 31) == 0; /* the 32x32 units of a 64x64 coding unit */
                if (!cuEdge && !tuEdge)
                    continue;
                if (cuEdge && (pos & 63) == 0 && (lcuEdge[lcu] & (dir ? 2 : 1)))
                    continue; /* tile (or picture) boundary: never filtered, the array keeps 0 */
                if (pos == 0)
                    continue;
                const SvtOracleCuMapEntry *nb = dir ? &map[(by - 1) * bw + bx] : &map[by * bw + bx - 1];
                for (uint32_t k = 0; k < 2; k++) { /* the two 4-sample segments of the 8-sample block side */
                    const uint32_t sx = dir ? x + 4 * k : x, sy = dir ? y : y + 4 * k;
                    const uint32_t nx = dir ? sx : sx - 4, ny = dir ? sy - 4 : sy;
                    const int cbfAny = cbf[(sy >> 2) * cw + (sx >> 2)] || cbf[(ny >> 2) * cw + (nx >> 2)];
                    int b;
                    if (tuEdge)
                        b = cur->mode == 2 ? 2 : cbfAny;
                    else
                        b = bs_pu_edge(cur, nb, cbfAny, sliceType, refPoc);
                    (dir ? bs_h : bs_v)[lcu * 256 + ((sx & 63) >> 2) + (((sy & 63) >> 2) << 4)] = (uint8_t)b;
                }
            }
}
