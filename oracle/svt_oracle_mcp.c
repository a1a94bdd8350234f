/*
 * oracle/svt_oracle_mcp.c - TEST INFRASTRUCTURE ONLY (see svt_oracle.h).
 * CPU restatement of the reference's HEVC motion-compensation interpolation leaf set
 * (/root/reference/Source/Lib/C_DEFAULT/EbMcp_C.c, ~150 functions; tables Codec/EbMcpTables.c:14-745):
 *   uni-prediction luma      LumaInterpolationCopy, LumaInterpolationFilterPos{a..r}New          (:221-730)
 *   bi-prediction luma (raw) LumaInterpolationCopyOutRaw, LumaInterpolationFilterPos{a..r}OutRaw (:731-1215, 3878)
 *   chroma                   ChromaInterpolationCopy/FilterOneD/FilterTwoD (+OutRaw)              (:3789, 3906, 4888-5099)
 *   averaging                BiPredClipping (:21), BiPredClipping16bit (:3265)
 *   and the 16-bit (10-bit sample) forms of all of them (:3121-4887).
 * Every one of them is the H.265 8.5.3.3.3 separable filter (horizontal first, then vertical) with the
 * reference's fixed-point conventions, which this file states once:
 *   s1   = 0 (8-bit) / 2 (10-bit)          first-pass shift            (Shift1 / SHIFT2D1_10BIT)
 *   B    = 8192, except 8-bit chroma: 0     bias that keeps raw values in int16 (MinusOffset1, ChromaMinusOffset1,
 *                                           OFFSET2D1_10BIT = -(8192 << 2))
 *   raw 1-D / first pass   (sum - (B << s1)) >> s1
 *   raw copy               (p << (6 - s1)) - B                          (Shift6/MinusOffset6, BI_SHIFT_10BIT)
 *   raw 2-D second pass    sum >> 6                                     (Shift2, BI_SHIFT2D2_10BIT)
 *   uni 1-D                clip((sum + 32) >> 6)                        (Shift3/Offset3, SHIFT1D_10BIT)
 *   uni 2-D second pass    clip((sum + (B << 6) + (1 << (11 - s1))) >> (12 - s1))   (Shift4/Offset4, SHIFT2D2_10BIT)
 * Pinned by tests/test_oracle_mcp.py against slot 0 (C_DEFAULT) of the reference's own tables.
 */
#include "svt_oracle.h"

static const int8_t LUMA_TAPS[4][8] = {{0, 0, 0, 64, 0, 0, 0, 0}, {-1, 4, -10, 58, 17, -5, 1, 0},
                                       {-1, 4, -11, 40, 40, -11, 4, -1}, {0, 1, -5, 17, 58, -10, 4, -1}};
static const int8_t CHROMA_TAPS[8][4] = {{0, 64, 0, 0}, {-2, 58, 10, -2}, {-4, 54, 16, -2}, {-6, 46, 28, -4},
                                         {-4, 36, 36, -4}, {-4, 28, 46, -6}, {-2, 16, 54, -4}, {-2, 10, 58, -2}};

static inline int clipm(int v, int maxv) { return v < 0 ? 0 : v > maxv ? maxv : v; }

void svt_oracle_mcp(int bps, int chroma, int out_raw, uint32_t fx, uint32_t fy, const void *ref, uint32_t srcStride,
                    void *dst, uint32_t dstStride, uint32_t w, uint32_t h)
{
    const int ntaps = chroma ? 4 : 8, first = chroma ? -1 : -3;
    const int8_t *tx = chroma ? CHROMA_TAPS[fx & 7] : LUMA_TAPS[fx & 3], *ty = chroma ? CHROMA_TAPS[fy & 7] : LUMA_TAPS[fy & 3];
    const int s1 = bps == 1 ? 0 : 2, B = (bps == 2 || !chroma) ? 8192 : 0, maxv = bps == 1 ? 255 : 1023;
    if (out_raw)
        dstStride = w;
#define REF(x, y) (bps == 1 ? (int)((const uint8_t *)ref)[(ptrdiff_t)(y) * srcStride + (x)] \
                            : (int)((const uint16_t *)ref)[(ptrdiff_t)(y) * srcStride + (x)])
    for (int y = 0; y < (int)h; y++)
        for (int x = 0; x < (int)w; x++) {
            int v;
            if (!fx && !fy) {
                v = out_raw ? (int16_t)((REF(x, y) << (6 - s1)) - B) : REF(x, y);
            } else if (!fy || !fx) {
                int sum = 0;
                for (int k = 0; k < ntaps; k++)
                    sum += (fx ? tx[k] : ty[k]) * (fx ? REF(x + first + k, y) : REF(x, y + first + k));
                v = out_raw ? (int16_t)((sum - (B << s1)) >> s1) : clipm((sum + 32) >> 6, maxv);
            } else {
                int sum = 0;
                for (int j = 0; j < ntaps; j++) {
                    int hs = 0;
                    for (int k = 0; k < ntaps; k++)
                        hs += tx[k] * REF(x + first + k, y + first + j);
                    sum += ty[j] * (int16_t)((hs - (B << s1)) >> s1);
                }
                v = out_raw ? (int16_t)(sum >> 6) : clipm((sum + (B << 6) + (1 << (11 - s1))) >> (12 - s1), maxv);
            }
            if (out_raw)
                ((int16_t *)dst)[(size_t)y * dstStride + x] = (int16_t)v;
            else if (bps == 1)
                ((uint8_t *)dst)[(size_t)y * dstStride + x] = (uint8_t)v;
            else
                ((uint16_t *)dst)[(size_t)y * dstStride + x] = (uint16_t)v;
        }
#undef REF
}

/* BiPredClipping (:21-49): clip((l0 + l1 + offset) >> 7); BiPredClipping16bit (:3265-3292): offset 16400, shift 5 */
void svt_oracle_BiPredClipping(int bps, uint32_t w, uint32_t h, const int16_t *l0, const int16_t *l1, void *dst,
                               uint32_t dstStride, int32_t offset)
{
    for (uint32_t y = 0; y < h; y++)
        for (uint32_t x = 0; x < w; x++) {
            const int s = l0[y * w + x] + l1[y * w + x];
            if (bps == 1)
                ((uint8_t *)dst)[(size_t)y * dstStride + x] = (uint8_t)clipm((s + offset) >> 7, 255);
            else
                ((uint16_t *)dst)[(size_t)y * dstStride + x] = (uint16_t)clipm((s + 16400) >> 5, 1023);
        }
}

/* ------------------------------------------------------------------------------------------------------------------
 * Encode-pass inter prediction of one prediction unit (4:2:0): EncodePassInterPrediction
 * (Codec/EbInterPrediction.c:761-926) -> EncodeUniPredInterpolation / EncodeBiPredInterpolation (Codec/EbMcp.c:175-250,
 * :562-760) composed from the leaf restatements above.  ref planes: pointers to the START of the padded buffers.
 * Pinned by tests/test_oracle_inter_golden.py on records of real calls.
 * ------------------------------------------------------------------------------------------------------------------ */
static int clamp_i(int lo, int hi, int v) { return v < lo ? lo : v > hi ? hi : v; }

static void inter_pu(int bps, const SvtAmdInterPuJob *J, const SvtAmdRefPicture *ref0, const SvtAmdRefPicture *ref1, void *pred_y,
                     uint32_t strideY, void *pred_cb, void *pred_cr, uint32_t strideC)
{
    const SvtAmdRefPicture *refs[2] = {ref0, ref1};
    const int bi = J->pred_dir == 2;
    int16_t raw[2][3][64 * 64];
    for (int l = 0; l < 2; l++) {
        if (!(bi || J->pred_dir == l))
            continue;
        const SvtAmdRefPicture *R = refs[l];
        /* position in quarter samples inside the padded buffer, clamped (:802-812) */
        const int px = clamp_i(((int)R->originX - 71) << 2, (int)(R->width + R->originX + 7) << 2, (((int)J->pu_x + (int)R->originX) << 2) + J->mv[l][0]);
        const int py = clamp_i(((int)R->originY - 71) << 2, (int)(R->height + R->originY + 7) << 2, (((int)J->pu_y + (int)R->originY) << 2) + J->mv[l][1]);
        for (int p = 0; p < 3; p++) {
            const int sh = p ? 1 : 0, w = J->pu_w >> sh, h = J->pu_h >> sh;
            const uint32_t stride = p ? R->strideC : R->strideY;
            const int ix = p ? px >> 3 : px >> 2, iy = p ? py >> 3 : py >> 2, fx = p ? px & 7 : px & 3, fy = p ? py & 7 : py & 3;
            const uint8_t *src = (const uint8_t *)(p == 0 ? R->d_y : p == 1 ? R->d_cb : R->d_cr) + ((size_t)iy * stride + ix) * bps;
            void *dst = p == 0 ? pred_y : p == 1 ? pred_cb : pred_cr;
            if (bi)
                svt_oracle_mcp(bps, p != 0, 1, (uint32_t)fx, (uint32_t)fy, src, stride, raw[l][p], 0, (uint32_t)w, (uint32_t)h);
            else
                svt_oracle_mcp(bps, p != 0, 0, (uint32_t)fx, (uint32_t)fy, src, stride, dst, p ? strideC : strideY, (uint32_t)w, (uint32_t)h);
        }
    }
    if (bi)
        for (int p = 0; p < 3; p++) /* Offset5 / ChromaOffset5 (Codec/EbDefinitions.h:1022-1030); the 16-bit clip has its own constant */
            svt_oracle_BiPredClipping(bps, (uint32_t)(J->pu_w >> (p ? 1 : 0)), (uint32_t)(J->pu_h >> (p ? 1 : 0)), raw[0][p], raw[1][p],
                                      p == 0 ? pred_y : p == 1 ? pred_cb : pred_cr, p ? strideC : strideY, p ? 64 : 16448);
}

void svt_oracle_inter_pu(const SvtAmdInterPuJob *J, const SvtAmdRefPicture *ref0, const SvtAmdRefPicture *ref1, uint8_t *pred_y,
                         uint32_t strideY, uint8_t *pred_cb, uint8_t *pred_cr, uint32_t strideC)
{
    inter_pu(1, J, ref0, ref1, pred_y, strideY, pred_cb, pred_cr, strideC);
}

/* EncodePassInterPrediction16bit (Codec/EbInterPrediction.c:928-1110) -> UniPredInterpolation16bit / BiPredInterpolation16bit
 * (Codec/EbMcp.c:249, :804): the same positions and split on 16-bit sample planes.  Pinned by tests/test_oracle_inter_golden.py
 * on records of real calls of 10-bit encodes. */
void svt_oracle_inter_pu16bit(const SvtAmdInterPuJob *J, const SvtAmdRefPicture *ref0, const SvtAmdRefPicture *ref1, uint16_t *pred_y,
                              uint32_t strideY, uint16_t *pred_cb, uint16_t *pred_cr, uint32_t strideC)
{
    inter_pu(2, J, ref0, ref1, pred_y, strideY, pred_cb, pred_cr, strideC);
}
